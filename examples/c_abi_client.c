/* A plain-C client of the engine's C ABI (include/pinn_hip.h): what a ccall / cgo / JNI binding does, without any host framework.
 * Problem: u''(x) = -pi^2 sin(pi x) on [0, 1], u(0) = u(1) = 0, a 1 -> 16 -> 16 -> 1 tanh network; descriptor "pinnir 2" (equations as
 * s-expressions), 64 interior points + the two boundary points, 300 resident Adam iterations, then the trial function on a grid.
 *   gcc -std=c99 -Iinclude examples/c_abi_client.c -o c_abi_client -L<dir of the library> -lpinn_hip   (or -l:libpinn_emu.so for the CPU emulation)
 * Exit status 0 when the loss fell and the ABI calls all succeeded. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "pinn_hip.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, pinn_last_error()); return 1; } } while (0)

int main(void) {
    const char* desc =
        "pinnir 2\n"
        "ntheta 337\n"                       /* 1*16+16 + 16*16+16 + 16+1 */
        "params 0 0 337\n"
        "defaults \n"
        "pnames \n"
        "nets 1\n"
        "net 0 tanh 0 4 1 16 16 1\n"
        "netvar 0 u 1 x\n"
        "terms 3\n"
        "sterm 0 1 x\n"
        "lhs (D x 2 (u x))\n"
        "rhs (* -1 (* (^ pi 2) (sin (* pi x))))\n"
        "sterm 1 1 x\n"
        "lhs (u 0)\n"
        "rhs 0\n"
        "sterm 2 1 x\n"
        "lhs (u 1)\n"
        "rhs 0\n"
        "hint 0 64\nhint 1 1\nhint 2 1\n";
    pinn_handle h = NULL;
    CHECK(pinn_create(desc, &h));
    printf("backend: %s\n", pinn_backend());
    float xs[64], x0 = 0.f, x1 = 1.f;
    for (int i = 0; i < 64; ++i) xs[i] = (i + 0.5f) / 64.f;
    CHECK(pinn_set_points(h, 0, xs, 64, 0));
    CHECK(pinn_set_points(h, 1, &x0, 1, 0));
    CHECK(pinn_set_points(h, 2, &x1, 1, 0));
    float theta[337];
    unsigned s = 12345u;
    for (int i = 0; i < 337; ++i) { s = s * 1664525u + 1013904223u; theta[i] = ((float)(s >> 8) / 16777216.f - 0.5f) * 0.8f; }
    double losses[3];
    float grad[337];
    CHECK(pinn_loss_grad(h, theta, 337, NULL, losses, grad));
    const double first = losses[0] + losses[1] + losses[2];
    printf("initial losses: %.4e %.4e %.4e\n", losses[0], losses[1], losses[2]);
    float w[3] = {1.f, 100.f, 100.f};
    double hist[300];
    CHECK(pinn_adam_init(h, theta, 337));
    CHECK(pinn_adam_steps(h, 300, 0.01f, 0.9f, 0.999f, 1e-8f, w, hist));
    CHECK(pinn_adam_get(h, theta, 337));
    char path[64];
    CHECK(pinn_get_option(h, "adam_path", path, (int64_t)sizeof path));      /* "persistent": all 300 iterations inside one launch (small problems) */
    printf("pinn_adam_steps ran: %s\n", path);
    CHECK(pinn_loss_grad(h, theta, 337, NULL, losses, NULL));
    const double last = losses[0] + losses[1] + losses[2];
    printf("after 300 Adam iterations: %.4e %.4e %.4e (weighted objective %.4e -> %.4e)\n", losses[0], losses[1], losses[2], hist[0], hist[299]);
    float grid[5] = {0.1f, 0.3f, 0.5f, 0.7f, 0.9f}, u[5];
    CHECK(pinn_phi(h, 0, theta, 337, grid, 5, u));
    for (int i = 0; i < 5; ++i) printf("u(%.1f) = %+.4f   (sin(pi x) = %+.4f)\n", grid[i], u[i], sin(3.14159265358979 * grid[i]));
    CHECK(pinn_destroy(h));
    return last < first ? 0 : 2;
}
