// plat.hpp — the few runtime calls the engine needs, for the two builds of the SAME host code:
//   default   : HIP runtime on a gfx950 device (the product, libpinn_hip.so)
//   PINN_EMU  : host memory + lock-step wave emulation (tests/emu only; never shipped)
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>

#ifdef PINN_EMU
typedef void* plat_stream;
struct plat_event { double t; };
inline const char* plat_name() { return "emu"; }
inline int plat_init(std::string&) { return 0; }
// poison fresh allocations (0xFF.. = NaN) so that reads of never-written device memory fail the CPU tests
inline void* plat_malloc(size_t n) { void* p = std::malloc(n ? n : 1); if (p) std::memset(p, 0xFF, n ? n : 1); return p; }
inline void plat_free(void* p) { std::free(p); }
inline void* plat_host_alloc(size_t n) { return plat_malloc(n); }
inline void plat_host_free(void* p) { std::free(p); }
inline int plat_h2d(void* d, const void* s, size_t n, plat_stream) { std::memcpy(d, s, n); return 0; }
inline int plat_d2h(void* d, const void* s, size_t n, plat_stream) { std::memcpy(d, s, n); return 0; }
inline int plat_d2d(void* d, const void* s, size_t n, plat_stream) { std::memcpy(d, s, n); return 0; }
inline int plat_memset(void* d, int v, size_t n, plat_stream) { std::memset(d, v, n); return 0; }
inline int plat_sync(plat_stream) { return 0; }
struct plat_graph { int unused = 0; };
inline bool plat_graph_capture_begin(plat_stream) { return false; }        // the emulation has no graphs: callers run the plain loop
inline bool plat_graph_capture_end(plat_stream, plat_graph&) { return false; }
inline bool plat_graph_launch(plat_graph&, plat_stream) { return false; }
inline void plat_graph_destroy(plat_graph&) {}
inline int plat_num_cus() { return 2; }
inline int plat_device_count() { return 8; }            // the emulation pretends to be an 8-device node (host memory: "devices" are labels)
inline int plat_get_device() { return 0; }
inline int plat_set_device(int) { return 0; }
inline plat_stream plat_stream_create() { return nullptr; }
inline void plat_stream_destroy(plat_stream) {}
inline void plat_event_create(plat_event&) {}
inline void plat_event_destroy(plat_event&) {}
inline void plat_event_record(plat_event&, plat_stream) {}
inline float plat_event_ms(plat_event&, plat_event&) { return 0.f; }
inline void plat_event_sync(plat_event&) {}
inline void plat_stream_wait_event(plat_stream, plat_event&) {}
inline const char* plat_last_error() { return ""; }
#else
#include <hip/hip_runtime.h>
typedef hipStream_t plat_stream;
struct plat_event { hipEvent_t e; };
inline const char* plat_name() { return "hip"; }
inline const char* plat_last_error() { return hipGetErrorString(hipGetLastError()); }
inline int plat_init(std::string& err) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        err = std::string("no HIP device available (") + hipGetErrorString(e) + "): the PINN engine has no CPU fallback";
        return 1;
    }
    return 0;
}
inline int plat_device_count() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
inline int plat_get_device() { int d = 0; (void)hipGetDevice(&d); return d; }
inline int plat_set_device(int d) { return hipSetDevice(d) != hipSuccess; }
inline void* plat_malloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n ? n : 4) != hipSuccess) return nullptr;
    return p;
}
inline void plat_free(void* p) { if (p) (void)hipFree(p); }
// pinned, device-mapped, coherent host memory: kernels write results straight into it (no copy command on the host path)
inline void* plat_host_alloc(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n ? n : 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
    return p;
}
inline void plat_host_free(void* p) { if (p) (void)hipHostFree(p); }
inline int plat_h2d(void* d, const void* s, size_t n, plat_stream st) { return hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st) != hipSuccess; }
inline int plat_d2h(void* d, const void* s, size_t n, plat_stream st) { return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st) != hipSuccess; }
inline int plat_d2d(void* d, const void* s, size_t n, plat_stream st) { return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st) != hipSuccess; }
inline int plat_memset(void* d, int v, size_t n, plat_stream st) { return hipMemsetAsync(d, v, n, st) != hipSuccess; }
inline int plat_sync(plat_stream st) { return hipStreamSynchronize(st) != hipSuccess; }
inline int plat_num_cus() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount;
}
inline plat_stream plat_stream_create() { hipStream_t s = nullptr; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return s; }
inline void plat_stream_destroy(plat_stream s) { if (s) (void)hipStreamDestroy(s); }
inline void plat_event_create(plat_event& e) { (void)hipEventCreate(&e.e); }
inline void plat_event_destroy(plat_event& e) { (void)hipEventDestroy(e.e); }
inline void plat_event_record(plat_event& e, plat_stream s) { (void)hipEventRecord(e.e, s); }
inline void plat_event_sync(plat_event& e) { (void)hipEventSynchronize(e.e); }
inline void plat_stream_wait_event(plat_stream s, plat_event& e) { (void)hipStreamWaitEvent(s, e.e, 0); }
inline float plat_event_ms(plat_event& a, plat_event& b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, a.e, b.e); return ms; }
// hipGraph of a launch sequence recorded from a stream (the resident optimiser loop replays one step's graph)
struct plat_graph { hipGraph_t g = nullptr; hipGraphExec_t x = nullptr; };
inline bool plat_graph_capture_begin(plat_stream s) { return hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess; }
inline bool plat_graph_capture_end(plat_stream s, plat_graph& G) {
    if (hipStreamEndCapture(s, &G.g) != hipSuccess || !G.g) { G.g = nullptr; (void)hipGetLastError(); return false; }
    if (hipGraphInstantiate(&G.x, G.g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(G.g); G.g = nullptr; G.x = nullptr; (void)hipGetLastError(); return false; }
    return true;
}
inline bool plat_graph_launch(plat_graph& G, plat_stream s) { return hipGraphLaunch(G.x, s) == hipSuccess; }
inline void plat_graph_destroy(plat_graph& G) {
    if (G.x) (void)hipGraphExecDestroy(G.x);
    if (G.g) (void)hipGraphDestroy(G.g);
    G.x = nullptr; G.g = nullptr;
}
#endif
