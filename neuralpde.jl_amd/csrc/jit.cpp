// jit.cpp — runtime specialisation of the residual kernels for network shapes / jet sets outside the ahead-of-time table.
//
// The reference accepts ANY Lux chain per dependent variable (src/pinn_types.jl:79-108).  The kernels are compile-time specialised
// (padded width, depth, input count, jet-channel set: csrc/inst*.hip list the shapes of the reference's tests and of the five
// BASELINE configs); for everything else `pinn_create` generates the one-line instantiation of the SAME kernel templates, compiles it
// with the toolchain that built the library (hipcc --offload-arch=gfx950 for the product; g++ -DPINN_EMU for the test-only emulation
// build), caches the shared object by its specialisation key, loads it and continues.  No second code path: a specialised kernel is the
// identical template, and unsupported combinations (e.g. tiles that exceed the 160 KB of LDS) fail with the compiler's static_assert text.
//
//   cache   : $PINN_JIT_DIR, default <kernel source dir>/jit_cache/<backend>/   (objects are reused across processes and runs)
//   sources : $PINN_SRC_DIR, default the directory of the library (csrc/) or the one compiled in
//   opt out : PINN_NO_JIT=1  ->  shapes outside the table fail at pinn_create with the line to add to csrc/inst_*.hip
#include "engine_types.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <fstream>
#include <set>

#ifndef PINN_SRC_DIR_DEFAULT
#define PINN_SRC_DIR_DEFAULT ""
#endif

namespace pe {

namespace {

bool file_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

std::string dir_of_this_library() {
    Dl_info info;
    if (dladdr((void*)&dir_of_this_library, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t s = p.find_last_of('/');
        return s == std::string::npos ? "." : p.substr(0, s);
    }
    return ".";
}

std::string source_dir() {
    if (const char* e = std::getenv("PINN_SRC_DIR")) return e;
    const std::string d = dir_of_this_library();
    for (const std::string& c : {d, d + "/../../neuralpde.jl_amd/csrc", std::string(PINN_SRC_DIR_DEFAULT)})
        if (!c.empty() && file_exists(c + "/spec_registry.hpp")) return c;
    return "";
}

std::string tail_of(const std::string& path, size_t n) {
    std::ifstream f(path);
    std::string all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    // prefer the first "error" line: the static_assert text says what does not fit
    const size_t e = all.find("error");
    if (e != std::string::npos) {
        const size_t b = all.rfind('\n', e);
        return all.substr(b == std::string::npos ? 0 : b + 1, n);
    }
    return all.size() > n ? all.substr(all.size() - n) : all;
}

std::set<std::string>& loaded_keys() { static std::set<std::string> s; return s; }

}  // namespace

int jit_round_hp(int h) {
    if (h <= 16) return 16;
    if (h <= 32) return 32;
    return ((h + 63) / 64) * 64;          // the neuron-split kernels need a multiple of 64 (16 x waves per workgroup)
}

// Compile (or load from the cache) the kernel family member with the given compile-time parameters and add it to the registry.
// variant: 0 tanh / sigmoid, 1 + sin, 2 + per-layer tanh / sigmoid (family 1 only).
int jit_spec(int HP, int NHH, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int variant) {
    static const bool off = std::getenv("PINN_NO_JIT") != nullptr;
    if (off) return fail("runtime specialisation is disabled (PINN_NO_JIT)");
    const int family = HP >= 64 ? 2 : 1;
    if (family == 2 && variant == 2) return fail("per-layer tanh/sigmoid chains are compiled for nets up to 32 wide (one-wave-per-tile kernels) only");
    if (family == 2 && NHH < 1) return fail("the neuron-split kernels need at least two hidden layers");
    // point groups per tile: about 4-5 column groups of 16 (jet channels x point groups), as in the ahead-of-time table
    int C = 1 + NPAIR + ((HI >> 24) ? 1 : 0);
    for (int a = 0; a < 8; ++a) C += ((D1MASK >> a) & 1) + (a < 6 && ((HI >> (4 * a)) & 0xF) >= 3) + (a < 6 && ((HI >> (4 * a)) & 0xF) >= 4);
    const int PG = C >= 3 ? 1 : (C == 2 ? 2 : 4);
    char key[200];
    std::snprintf(key, sizeof key, "f%d_hp%d_nhh%d_d%d_f%x_p%llx_n%d_pg%d_h%x_v%d", family, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI, variant);
    if (loaded_keys().count(key)) return fail(std::string("specialised kernel ") + key + " is loaded but does not satisfy the request (internal)");
    const std::string src = source_dir();
    if (src.empty()) return fail("runtime specialisation needs the kernel sources (spec_registry.hpp); set PINN_SRC_DIR");
    std::string cache = std::getenv("PINN_JIT_DIR") ? std::getenv("PINN_JIT_DIR") : src + "/jit_cache";
    ::mkdir(cache.c_str(), 0755);
    cache += std::string("/") + plat_name();
    ::mkdir(cache.c_str(), 0755);
    const std::string base = cache + "/" + key, so = base + ".so";
    if (!file_exists(so)) {
        const std::string tmp = base + "." + std::to_string((long)::getpid());
        {
            std::ofstream f(tmp + ".hip");
            if (!f) return fail("cannot write " + tmp + ".hip (set PINN_JIT_DIR to a writable directory)");
            const char* macro = family == 2 ? (variant == 1 ? "PINN_INSTANTIATE2_HI_SIN" : "PINN_INSTANTIATE2_HI")
                                            : (variant == 1 ? "PINN_INSTANTIATE_HI_SIN" : (variant == 2 ? "PINN_INSTANTIATE_HI_MIX" : "PINN_INSTANTIATE_HI"));
            f << "// generated by jit.cpp: " << key << "\n#include \"spec_registry.hpp\"\n"
              << "#ifdef PINN_EMU\nnamespace wv { thread_local void (*emu_barrier_hook)(void*) = nullptr; thread_local void* emu_barrier_ctx = nullptr; }\n#endif\n"
              << "namespace pk { std::deque<SpecInfo>& registry() { static std::deque<SpecInfo> r; return r; } }\n"
              << macro << "(jit, " << HP << ", " << NHH << ", " << D << ", 0x" << std::hex << D1MASK << "u, 0x" << PAIRS << "ull, " << std::dec << NPAIR
              << ", " << PG << ", 0x" << std::hex << HI << std::dec << "u)\n"
              << "extern \"C\" __attribute__((visibility(\"default\"))) const pk::SpecInfo* pinn_jit_specs(int* n) { static std::vector<pk::SpecInfo> v(pk::registry().begin(), pk::registry().end()); *n = (int)v.size(); return v.data(); }\n";
        }
#ifdef PINN_EMU
        const char* cxx = std::getenv("CXX") ? std::getenv("CXX") : "g++";
        const std::string cmd = std::string(cxx) + " -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -DPINN_EMU -Wno-unknown-pragmas -I'" + src + "' -x c++ '" + tmp +
                                ".hip' -o '" + tmp + ".so' -lpthread > '" + tmp + ".log' 2>&1";
#else
        const char* hipcc = std::getenv("HIPCC") ? std::getenv("HIPCC") : "/opt/rocm/bin/hipcc";
        const std::string cmd = std::string(hipcc) + " -O3 -std=c++17 -fPIC -shared -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -I'" +
                                src + "' '" + tmp + ".hip' -o '" + tmp + ".so' > '" + tmp + ".log' 2>&1";
#endif
        std::fprintf(stderr, "[pinn] specialising kernel %s (one-off; cached in %s)\n", key, cache.c_str());
        const int rc = std::system(cmd.c_str());
        if (rc != 0 || !file_exists(tmp + ".so")) {
            const std::string why = tail_of(tmp + ".log", 600);
            std::remove((tmp + ".hip").c_str());
            std::remove((tmp + ".so").c_str());
            std::rename((tmp + ".log").c_str(), (base + ".failed.log").c_str());
            return fail(std::string("runtime specialisation of ") + key + " failed: " + why);
        }
        std::rename((tmp + ".hip").c_str(), (base + ".hip").c_str());
        std::remove((tmp + ".log").c_str());
        std::rename((tmp + ".so").c_str(), so.c_str());           // atomic: concurrent processes (one per GPU) may race to the same key
    }
    void* lib = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(std::string("cannot load specialised kernel ") + so + ": " + dlerror());
    typedef const pk::SpecInfo* (*specs_fn)(int*);
    specs_fn fn = (specs_fn)dlsym(lib, "pinn_jit_specs");
    if (!fn) return fail("specialised kernel " + so + " lacks pinn_jit_specs");
    int n = 0;
    const pk::SpecInfo* sp = fn(&n);
    for (int i = 0; i < n; ++i) { pk::registry().push_back(sp[i]); pk::registry().back().jit = 1; }
    loaded_keys().insert(key);
    return 0;
}

}  // namespace pe
