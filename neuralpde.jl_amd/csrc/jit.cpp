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
#include <map>
#include <set>
#include <sstream>

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
static int jit_build(int HP, int NHH, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int variant, int C,
                     const std::string& preamble, const std::string& keytail, const std::string& instantiate = "");

int jit_spec(int HP, int NHH, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int variant) {
    int C = 1 + NPAIR + ((HI >> 24) ? 1 : 0);
    for (int a = 0; a < 8; ++a) C += ((D1MASK >> a) & 1) + (a < 6 && ((HI >> (4 * a)) & 0xF) >= 3) + (a < 6 && ((HI >> (4 * a)) & 0xF) >= 4);
    return jit_build(HP, NHH, D, D1MASK, PAIRS, NPAIR, HI, variant, C, "", "");
}

static int jit_build(int HP, int NHH, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int variant, int C,
                     const std::string& preamble, const std::string& keytail, const std::string& instantiate) {
    static const bool off = std::getenv("PINN_NO_JIT") != nullptr;
    if (off) return fail("runtime specialisation is disabled (PINN_NO_JIT)");
    const int family = !instantiate.empty() ? 3 : (HP >= 64 ? 2 : 1);
    if (family == 2 && variant == 2) return fail("per-layer tanh/sigmoid chains are compiled for nets up to 32 wide (one-wave-per-tile kernels) only");
    if (family == 2 && NHH < 1) return fail("the neuron-split kernels need at least two hidden layers");
    // point groups per tile: about 4-5 column groups of 16 (jet channels x point groups), as in the ahead-of-time table
    const int PG = C >= 3 ? 1 : (C == 2 ? 2 : 4);
    char key[260];
    std::snprintf(key, sizeof key, "f%d_hp%d_nhh%d_d%d_f%x_p%llx_n%d_pg%d_h%x_v%d%s", family, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI, variant, keytail.c_str());
    if (loaded_keys().count(key)) return fail(std::string("specialised kernel ") + key + " is loaded but does not satisfy the request (internal)");
    const std::string src = source_dir();
    if (src.empty()) return fail("runtime specialisation needs the kernel sources (spec_registry.hpp); set PINN_SRC_DIR");
    std::string cache = std::getenv("PINN_JIT_DIR") ? std::getenv("PINN_JIT_DIR") : src + "/jit_cache";
    ::mkdir(cache.c_str(), 0755);
    cache += std::string("/") + plat_name();
    ::mkdir(cache.c_str(), 0755);
    {
        // one sub-directory per BUILD of the library: every host source depends on every kernel header (Makefile), so a change to the
        // kernel templates or to SpecInfo recompiles this file and retires the objects specialised against the old sources
        unsigned h = 2166136261u;
        for (const char* c = __DATE__ " " __TIME__; *c; ++c) h = (h ^ (unsigned char)*c) * 16777619u;
        char b[16];
        std::snprintf(b, sizeof b, "/b%08x", h);
        cache += b;
        ::mkdir(cache.c_str(), 0755);
    }
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
              << preamble;
            if (!instantiate.empty()) f << instantiate;
            else f << macro << "(jit, " << HP << ", " << NHH << ", " << D << ", 0x" << std::hex << D1MASK << "u, 0x" << PAIRS << "ull, " << std::dec << NPAIR
              << ", " << PG << ", 0x" << std::hex << HI << std::dec << "u)\n";
            f << "extern \"C\" __attribute__((visibility(\"default\"))) const pk::SpecInfo* pinn_jit_specs(int* n) { static std::vector<pk::SpecInfo> v(pk::registry().begin(), pk::registry().end()); *n = (int)v.size(); return v.data(); }\n";
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


// ------------------------------------------------------------------------------------------------------------------------------
// General multi-index jet sets (mixed derivatives of order >= 3, orders 5 and 6): the reference's `numeric_derivative` recursion
// takes any list of axes (src/pinn_types.jl:454-460).  The fixed channel categories of JetSet (first, pairs, Laplacian, pure 3rd /
// 4th) do not cover them, so for such a request the Faa di Bruno rules of the channel set — closed under sub-multi-indices — are
// GENERATED here as an explicit specialisation of JetSet / jet_forward / jet_adjoint and compiled with the kernel:
//     D^alpha phi(z) = sum over set partitions pi of alpha:  phi^(|pi|)(z) * prod_{B in pi} D^B z
// and its reverse mode.  A multi-index is a sorted list of axes, encoded as nibble 0 = order, nibbles 1.. = axes.
// ------------------------------------------------------------------------------------------------------------------------------
typedef std::vector<int> MI;

unsigned mi_encode(const MI& m) {
    unsigned v = (unsigned)m.size();
    for (size_t i = 0; i < m.size(); ++i) v |= (unsigned)m[i] << (4 * (i + 1));
    return v;
}
MI mi_decode(unsigned v) {
    MI m((size_t)(v & 0xF));
    for (size_t i = 0; i < m.size(); ++i) m[i] = (int)((v >> (4 * (i + 1))) & 0xF);
    return m;
}

// requested multi-indices -> ordered, closed channel list: [value | firsts by axis | the rest by (order, axes)]
std::vector<unsigned> gen_close(const std::vector<unsigned>& want) {
    std::set<MI> all;
    for (unsigned w : want) {
        const MI m = mi_decode(w);
        const int n = (int)m.size();
        for (int mask = 1; mask < (1 << n); ++mask) {
            MI sub;
            for (int i = 0; i < n; ++i) if (mask & (1 << i)) sub.push_back(m[i]);
            all.insert(sub);
        }
    }
    std::vector<MI> v(all.begin(), all.end());
    std::sort(v.begin(), v.end(), [](const MI& a, const MI& b) { return a.size() != b.size() ? a.size() < b.size() : a < b; });
    std::vector<unsigned> out{0u};
    for (auto& m : v) out.push_back(mi_encode(m));
    return out;
}

namespace {
struct Mono { int k; std::vector<int> ch; long count; };

// monomials of D^alpha phi(z): set partitions of the positions of alpha, blocks mapped to channels, identical monomials merged
std::vector<Mono> monomials(const MI& alpha, const std::map<MI, int>& chan) {
    const int n = (int)alpha.size();
    std::map<std::pair<int, std::vector<int>>, long> acc;
    std::vector<int> rgs(n, 0);                       // restricted growth string
    for (;;) {
        int k = 0;
        for (int v : rgs) k = std::max(k, v + 1);
        std::vector<int> chs;
        for (int b = 0; b < k; ++b) {
            MI blk;
            for (int i = 0; i < n; ++i) if (rgs[i] == b) blk.push_back(alpha[i]);
            std::sort(blk.begin(), blk.end());
            chs.push_back(chan.at(blk));
        }
        std::sort(chs.begin(), chs.end());
        acc[{k, chs}] += 1;
        int i = n - 1;                                 // next restricted growth string
        for (; i > 0; --i) {
            int mx = 0;
            for (int j = 0; j < i; ++j) mx = std::max(mx, rgs[j]);
            if (rgs[i] <= mx) { ++rgs[i]; for (int j = i + 1; j < n; ++j) rgs[j] = 0; break; }
        }
        if (i == 0) break;
    }
    std::vector<Mono> out;
    for (auto& kv : acc) out.push_back(Mono{kv.first.first, kv.first.second, kv.second});
    return out;
}
std::string prod(const char* arr, const std::vector<int>& chs, int skip_one = -1) {
    std::string s;
    bool skipped = false;
    for (int c : chs) {
        if (c == skip_one && !skipped) { skipped = true; continue; }
        s += std::string(s.empty() ? "" : " * ") + arr + "[" + std::to_string(c) + "]";
    }
    return s.empty() ? "vfloat(1.0f)" : s;
}
}  // namespace

// the generated JetSet specialisation for a closed channel list; outputs its template arguments and a cache-key suffix
static int gen_preamble(int D, const std::vector<unsigned>& channels, std::string& text, unsigned& d1mask_out, unsigned& HI_out, std::string& keytail_out) {
    const int C = (int)channels.size();
    if (C > pk::MAX_GEN_CHANNELS) return fail("derivative set needs " + std::to_string(C) + " jet channels (limit " + std::to_string(pk::MAX_GEN_CHANNELS) + ")");
    std::map<MI, int> chan;
    unsigned d1mask = 0;
    int maxord = 0, nfirst = 0;
    unsigned hash = 2166136261u;
    std::string keytail = "_g";
    for (int c = 0; c < C; ++c) {
        const MI m = mi_decode(channels[c]);
        chan[m] = c;
        maxord = std::max(maxord, (int)m.size());
        if (m.size() == 1) { d1mask |= 1u << m[0]; ++nfirst; }
        for (int a : m) if (a >= D) return fail("derivative axis out of range");
        hash = (hash ^ channels[c]) * 16777619u;
    }
    {
        char b[40];
        std::snprintf(b, sizeof b, "%08x_c%d", hash, C);
        keytail += b;
    }
    if (maxord > 6) return fail("derivative order > 6 is not supported by the HIP engine");
    const unsigned HI = 0x80000000u | (hash & 0x00FFFFFFu);
    std::ostringstream o;
    char jt[96];
    std::snprintf(jt, sizeof jt, "JetSet<0x%xu, 0ull, 0, 0x%xu>", d1mask, HI);
    o << "namespace pk {\ntemplate <> struct " << jt << " {\n"
      << "    static constexpr bool GEN = true;\n"
      << "    static constexpr int NFIRST = " << nfirst << ", NPAIR = 0, C = " << C << ", NLAP = 0, N3 = 0, N4 = 0, NORD = " << maxord + 1 << ";\n"
      << "    static constexpr unsigned LAP = 0;\n"
      << "    static constexpr int CH_FIRST = 1, CH_PAIR = " << 1 + nfirst << ", CH_LAP = CH_PAIR, CH_3 = CH_PAIR, CH_4 = CH_PAIR;\n"
      << "    static constexpr int first_axis(int k) { constexpr int t[] = {";
    for (int c = 1; c <= nfirst; ++c) o << mi_decode(channels[c])[0] << ", ";
    o << "-1}; return t[k]; }\n"
      << "    static constexpr int first_rank(int axis) { int c = 0; for (int a = 0; a < axis; ++a) if (0x" << std::hex << d1mask << std::dec << "u & (1u << a)) ++c; return c; }\n"
      << "    static constexpr int pair_a(int) { return -1; }\n    static constexpr int pair_b(int) { return -1; }\n    static constexpr int pair_index(int, int) { return -1; }\n"
      << "    static constexpr unsigned gen_channel(int i) { constexpr unsigned t[] = {";
    for (int c = 0; c < C; ++c) o << "0x" << std::hex << channels[c] << std::dec << "u, ";
    o << "0u}; return t[i]; }\n    static constexpr unsigned channel_mi(int i) { return gen_channel(i); }\n};\nusing JG = " << jt << ";\n";
    // forward rule
    std::vector<std::vector<Mono>> mons(C);
    for (int c = 1; c < C; ++c) mons[c] = monomials(mi_decode(channels[c]), chan);
    o << "template <> DEV void jet_forward<JG>(vfloat (&z)[JG::C], const vfloat (&d)[ND]) {\n";
    for (int c = 1; c < C; ++c) {
        o << "    const vfloat o" << c << " = ";
        for (size_t i = 0; i < mons[c].size(); ++i) {
            const Mono& m = mons[c][i];
            o << (i ? " + " : "") << "vfloat(" << m.count << ".0f) * d[" << m.k << "] * " << prod("z", m.ch);
        }
        o << ";\n";
    }
    for (int c = 1; c < C; ++c) o << "    z[" << c << "] = o" << c << ";\n";
    o << "}\n";
    // adjoint: g = adjoints of the post-activation channels in, of the pre-activation channels out; s = record (s[0] = a)
    o << "template <> DEV void jet_adjoint<JG>(vfloat (&g)[JG::C], const vfloat (&s)[JG::C], const vfloat (&d)[ND]) {\n"
      << "    vfloat zv = d[1] * g[0];\n";
    for (int b = 1; b < C; ++b) o << "    vfloat n" << b << " = vfloat(0.f);\n";
    for (int c = 1; c < C; ++c)
        for (const Mono& m : mons[c]) {
            o << "    zv = vfma(vfloat(" << m.count << ".0f) * d[" << m.k + 1 << "] * " << prod("s", m.ch) << ", g[" << c << "], zv);\n";
            std::set<int> distinct(m.ch.begin(), m.ch.end());
            for (int b : distinct) {
                const long mult = (long)std::count(m.ch.begin(), m.ch.end(), b);
                o << "    n" << b << " = vfma(vfloat(" << m.count * mult << ".0f) * d[" << m.k << "] * " << prod("s", m.ch, b) << ", g[" << c << "], n" << b << ");\n";
            }
        }
    o << "    g[0] = zv;\n";
    for (int b = 1; b < C; ++b) o << "    g[" << b << "] = n" << b << ";\n";
    o << "}\n}  // namespace pk\n";
    text = o.str();
    d1mask_out = d1mask;
    HI_out = HI;
    keytail_out = keytail;
    return 0;
}

int jit_spec_gen(int HP, int NHH, int D, const std::vector<unsigned>& channels, int variant) {
    std::string text, keytail;
    unsigned d1mask = 0, HI = 0;
    if (gen_preamble(D, channels, text, d1mask, HI, keytail)) return 1;
    return jit_build(HP, NHH, D, d1mask, 0ull, 0, HI, variant, (int)channels.size(), text, keytail);
}

// DGM network (family 3, pinn_kernels3.hpp): always specialised at run time
int jit_spec_dgm(int MP, int L, int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, const std::vector<unsigned>* gen_channels,
                 int act1, int act2) {
    std::string text, keytail;
    if (gen_channels) {
        if (gen_preamble(D, *gen_channels, text, D1MASK, HI, keytail)) return 1;
        PAIRS = 0ull; NPAIR = 0;
    }
    char tail[64];
    std::snprintf(tail, sizeof tail, "%s_dgm_a%d_%d", keytail.c_str(), act1, act2);
    char line[256];
    std::snprintf(line, sizeof line, "PINN_INSTANTIATE_DGM(jit, %d, %d, %d, 0x%xu, 0x%llxull, %d, 0x%xu, %d, %d)\n", MP, L, D, D1MASK, PAIRS, NPAIR, HI, act1, act2);
    return jit_build(MP, L, D, D1MASK, PAIRS, NPAIR, HI, 0, 0, text, tail, line);
}


}  // namespace pe
