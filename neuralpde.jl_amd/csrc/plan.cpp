// plan.cpp — the kernel plan of a handle: which compiled kernel evaluates which term, launch groups, packed-weight and gradient-slab
// index maps, the fixed-order reduction tables, tile tables (retile).
#include <algorithm>
#include "engine_types.hpp"

namespace pe {

static thread_local int g_gemm = pk::GEMM_SPLIT;
int current_gemm() { return g_gemm; }
GemmScope::GemmScope(int mode) : prev(g_gemm) { g_gemm = mode; }
GemmScope::~GemmScope() { g_gemm = prev; }
// activation kind of the network whose kernel is being looked up (-1: none): a run-time specialised member compiles that kind's kernels first (jit.cpp: rtc_build)
static thread_local int g_act_hint = -1;
int current_act_hint() { return g_act_hint; }
namespace { struct ActScope { int prev; explicit ActScope(int a) : prev(g_act_hint) { g_act_hint = a; } ~ActScope() { g_act_hint = prev; } }; }
// a family-2 kernel serves the current GEMM mode when it is compiled in that mode, or when its shape has one form only
static bool gemm_ok(const pk::SpecInfo& s) { return s.family != 2 || !s.twin || s.gemm == g_gemm; }

int round_hp(int h) {
    if (h <= 16) return 16;
    if (h <= 32) return 32;
    if (h <= 64) return 64;
    if (h <= 128) return 128;
    return jit_round_hp(h);          // wider nets: multiples of 64, kernels specialised at run time (jit.cpp)
}

// DGM networks (family 3): one specialised kernel per (padded modes, layers, inputs, channel set, activations)
static const pk::SpecInfo* find_spec_dgm(int MP, int L, int D, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs, unsigned need_hi,
                                         int act1, int act2) {
    const pk::SpecInfo* best = nullptr;
    for (const pk::SpecInfo& s : pk::registry()) {
        if (s.family != 3 || s.HP != MP || s.NHH != L || s.D != D || s.act1 != act1 || s.act2 != act2) continue;
        bool ok = true;
        if (need_hi & GEN_FLAG) {
            if (s.ngen <= 0) continue;
            for (unsigned m : gen_set((int)(need_hi & ~GEN_FLAG))) {
                bool f = false;
                for (int c = 0; c < s.ngen; ++c) f = f || s.gen[c] == m;
                ok = ok && f;
            }
        } else {
            if (s.ngen > 0 || (s.D1MASK & need_first) != need_first) continue;
            for (int a = 0; a < 6; ++a)
                if (((need_hi >> (4 * a)) & 0xF) > ((s.HI >> (4 * a)) & 0xF)) ok = false;
            for (auto& pr : need_pairs) {
                bool f = false;
                for (int p = 0; p < s.NPAIR; ++p)
                    f = f || ((int)((s.PAIRS >> (8 * p)) & 0xF) == pr.first && (int)((s.PAIRS >> (8 * p + 4)) & 0xF) == pr.second);
                ok = ok && f;
            }
        }
        if (ok && (!best || s.C < best->C)) best = &s;
    }
    return best;
}
static const pk::SpecInfo* ensure_spec_dgm(const Net& N, unsigned need_first, std::vector<std::pair<int, int>> need_pairs, unsigned need_hi) {
    const int M = N.sizes[1], MP = M <= 16 ? 16 : (M <= 32 ? 32 : 64), L = N.dgm_layers, D = N.sizes[0];
    if ((need_hi >> 24) && !(need_hi & GEN_FLAG)) { fail("DGM networks carry no forward-Laplacian channel (internal)"); return nullptr; }
    const pk::SpecInfo* sp = find_spec_dgm(MP, L, D, need_first, need_pairs, need_hi, N.act, N.act2);
    int cneed = 1 + (int)need_pairs.size();
    for (int a = 0; a < 8; ++a) cneed += ((need_first >> a) & 1) + (a < 6 && ((need_hi >> (4 * a)) & 0xF) >= 3) + (a < 6 && ((need_hi >> (4 * a)) & 0xF) >= 4);
    if (need_hi & GEN_FLAG) cneed = (int)gen_set((int)(need_hi & ~GEN_FLAG)).size();
    if (sp && sp->C <= cneed) return sp;              // one kernel per channel set: a boundary term does not ride on the interior term's kernel
    if (need_hi & GEN_FLAG) {
        if (jit_spec_dgm(MP, L, D, 0, 0ull, 0, 0, &gen_set((int)(need_hi & ~GEN_FLAG)), N.act, N.act2)) return sp;
    } else {
        for (int a = 0; a < 6; ++a)
            if (((need_hi >> (4 * a)) & 0xF) >= 3 && std::find(need_pairs.begin(), need_pairs.end(), std::make_pair(a, a)) == need_pairs.end()) need_pairs.push_back({a, a});
        std::sort(need_pairs.begin(), need_pairs.end());
        unsigned long long PAIRS = 0;
        unsigned first = need_first;
        for (size_t p = 0; p < need_pairs.size(); ++p) {
            PAIRS |= ((unsigned long long)(need_pairs[p].first | (need_pairs[p].second << 4))) << (8 * p);
            first |= (1u << need_pairs[p].first) | (1u << need_pairs[p].second);
        }
        if (jit_spec_dgm(MP, L, D, first, PAIRS, (int)need_pairs.size(), need_hi, nullptr, N.act, N.act2)) return sp;     // (sp: a wider kernel, if any, still serves)
    }
    return find_spec_dgm(MP, L, D, need_first, need_pairs, need_hi, N.act, N.act2);
}

const pk::SpecInfo* ensure_spec(const Net& N, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs, unsigned need_hi, int need_family) {
    if (N.kind == 1) return ensure_spec_dgm(N, need_first, need_pairs, need_hi);
    ActScope as((N.act >= 0 && N.act < 4) ? N.act : -1);
    const int HP = round_hp(N.maxhidden()), NHH = (int)N.sizes.size() - 3, D = N.sizes[0], variant = variant_of(N.act);
    if (need_hi & GEN_FLAG) {                    // general multi-index channel set: always generated (jit.cpp: jit_spec_gen)
        const pk::SpecInfo* g = find_spec(HP, NHH, D, 0, {}, need_hi, nullptr, variant, need_family);
        if (g) return g;
        if (jit_spec_gen(HP, NHH, D, gen_set((int)(need_hi & ~GEN_FLAG)), variant)) return nullptr;
        return find_spec(HP, NHH, D, 0, {}, need_hi, nullptr, variant, need_family);
    }
    const pk::SpecInfo* sp = find_spec(HP, NHH, D, need_first, need_pairs, need_hi, nullptr, variant, need_family);
    int cneed = 1 + (int)need_pairs.size() + ((need_hi >> 24) ? 1 : 0);
    for (int a = 0; a < 8; ++a) cneed += ((need_first >> a) & 1) + (a < 6 && ((need_hi >> (4 * a)) & 0xF) >= 3) + (a < 6 && ((need_hi >> (4 * a)) & 0xF) >= 4);
    // a shape served only by run-time specialised kernels gets one per channel set (a boundary term does not ride on the interior
    // term's 4-5 channel kernel); inside the ahead-of-time table a wider set may serve (e.g. the full-Hessian kernels of the small nets)
    if (sp && !(sp->jit && sp->C > cneed)) return sp;
    // exactly the requested channel set: first-derivative axes, pairs (a pure 3rd / 4th derivative also needs pair (a, a)), HI nibbles, LAP
    unsigned long long PAIRS = 0;
    std::vector<std::pair<int, int>> pairs = need_pairs;
    for (int a = 0; a < 6; ++a)
        if (((need_hi >> (4 * a)) & 0xF) >= 3 && std::find(pairs.begin(), pairs.end(), std::make_pair(a, a)) == pairs.end()) pairs.push_back({a, a});
    if (pairs.size() > 8) { fail("more than 8 second-derivative channels in one kernel"); return nullptr; }
    std::sort(pairs.begin(), pairs.end());
    for (size_t p = 0; p < pairs.size(); ++p) PAIRS |= ((unsigned long long)(pairs[p].first | (pairs[p].second << 4))) << (8 * p);
    unsigned first = need_first;
    for (auto& pr : pairs) first |= (1u << pr.first) | (1u << pr.second);
    const std::string before = g_err;
    if (jit_spec(HP, NHH, D, first, PAIRS, (int)pairs.size(), need_hi, variant)) {
        if (sp) { g_err = before; return sp; }                     // the wider kernel still serves
        return nullptr;
    }
    g_err = before;
    return find_spec(HP, NHH, D, need_first, need_pairs, need_hi, nullptr, variant, need_family);
}

const pk::SpecInfo* find_spec(int HP, int NHH, int D, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs,
                              unsigned need_hi, std::vector<int>* pair_index, int need_variant, int need_family) {
    const pk::SpecInfo* best = nullptr;
    const bool want_gen = (need_hi & GEN_FLAG) != 0;
    for (const pk::SpecInfo& s : pk::registry()) {
        if (s.family == 3 || s.HP != HP || s.NHH != NHH || s.D != D || !gemm_ok(s)) continue;       // (family 3 = DGM networks: find_spec_dgm)
        if (need_variant && !(s.has_sin & need_variant)) continue;
        if (want_gen != (s.ngen > 0)) continue;
        if (want_gen) {                          // general multi-index set: every requested channel must be carried
            bool all = true;
            for (unsigned m : gen_set((int)(need_hi & ~GEN_FLAG))) {
                bool f = false;
                for (int c = 0; c < s.ngen; ++c) f = f || s.gen[c] == m;
                all = all && f;
            }
            if (!all) continue;
            if (need_family && s.family != need_family) continue;
            if (!best || s.C < best->C) best = &s;
            continue;
        }
        if ((s.D1MASK & need_first) != need_first) continue;
        bool ok = true;
        for (int a = 0; a < 6; ++a)
            if (((need_hi >> (4 * a)) & 0xF) > ((s.HI >> (4 * a)) & 0xF)) ok = false;
        if ((need_hi >> 24) && (need_hi >> 24) != s.LAP) ok = false;          // a Laplacian channel must cover exactly the requested axes
        for (auto& pr : need_pairs) {
            bool f = false;
            for (int p = 0; p < s.NPAIR; ++p) {
                int a = (int)((s.PAIRS >> (8 * p)) & 0xF), b = (int)((s.PAIRS >> (8 * p + 4)) & 0xF);
                if (a == pr.first && b == pr.second) f = true;
            }
            ok = ok && f;
        }
        if (!ok) continue;
        // PINN_KERNEL_FAMILY=1|2 restricts the choice (tests / A-B measurements); default: family 2 where compiled
        static const int want_family = [] { const char* e = std::getenv("PINN_KERNEL_FAMILY"); return e ? std::atoi(e) : 0; }();
        if (want_family && s.family != want_family) continue;
        if (need_family && s.family != need_family) continue;
        if (!best || s.C < best->C || (s.C == best->C && s.family > best->family) ||
            (s.C == best->C && s.family == best->family && s.PG > best->PG)) best = &s;
    }
    (void)pair_index;
    return best;
}

// does kernel `s` (fixed channel categories) carry every channel of the needs?  (the superset test of find_spec for ONE kernel)
static bool spec_covers(const pk::SpecInfo& s, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs, unsigned need_hi) {
    if (s.family == 3 || s.ngen > 0 || (need_hi & GEN_FLAG)) return false;
    if ((s.D1MASK & need_first) != need_first) return false;
    for (int a = 0; a < 6; ++a)
        if (((need_hi >> (4 * a)) & 0xF) > ((s.HI >> (4 * a)) & 0xF)) return false;
    if ((need_hi >> 24) && (need_hi >> 24) != s.LAP) return false;
    for (auto& pr : need_pairs) {
        bool f = false;
        for (int p = 0; p < s.NPAIR; ++p)
            f = f || ((int)((s.PAIRS >> (8 * p)) & 0xF) == pr.first && (int)((s.PAIRS >> (8 * p + 4)) & 0xF) == pr.second);
        if (!f) return false;
    }
    return true;
}

int first_rank(const pk::SpecInfo& s, int axis) {
    int c = 0;
    for (int a = 0; a < axis; ++a)
        if (s.D1MASK & (1u << a)) ++c;
    return c;
}

unsigned slot_mi(const Slot& s) {
    unsigned v = (unsigned)s.order;
    for (int a = 0; a < s.order; ++a) v |= (unsigned)s.axes[a] << (4 * (a + 1));
    return v;
}
bool slot_is_general(const Slot& s) {
    if (s.lap) return false;
    if (s.order >= 5) return true;
    for (int a = 1; a < s.order && s.order >= 3; ++a)
        if (s.axes[a] != s.axes[0]) return true;
    return false;
}
static std::vector<std::vector<unsigned>>& gen_sets() { static std::vector<std::vector<unsigned>> v; return v; }
int gen_set_id(const std::vector<unsigned>& want) {
    const std::vector<unsigned> closed = gen_close(want);
    auto& v = gen_sets();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == closed) return (int)i;
    v.push_back(closed);
    return (int)v.size() - 1;
}
const std::vector<unsigned>& gen_set(int id) { return gen_sets().at((size_t)id); }

int chan_of(const pk::SpecInfo& s, const Slot& sl) {
    if (s.ngen > 0) {
        if (sl.lap) return -1;
        const unsigned m = slot_mi(sl);
        for (int c = 0; c < s.ngen; ++c)
            if (s.gen[c] == m) return c;
        return -1;
    }
    if (sl.lap) return sl.lap == s.LAP ? 1 + s.NFIRST + s.NPAIR : -1;
    const int nlap = s.LAP ? 1 : 0;
    if (sl.order == 0) return 0;
    if (sl.order == 1) return 1 + first_rank(s, sl.axes[0]);
    if (sl.order >= 3) {             // pure third / fourth derivative: channels after the pairs, thirds first
        int n3 = 0, n3_before = 0, n4_before = 0;
        for (int a = 0; a < 6; ++a) {
            const int h = (int)((s.HI >> (4 * a)) & 0xF);
            if (h >= 3) { ++n3; if (a < sl.axes[0]) ++n3_before; }
            if (h >= 4 && a < sl.axes[0]) ++n4_before;
        }
        if (sl.axes[0] >= 6 || (int)((s.HI >> (4 * sl.axes[0])) & 0xF) < sl.order) return -1;
        return sl.order == 3 ? 1 + s.NFIRST + s.NPAIR + nlap + n3_before : 1 + s.NFIRST + s.NPAIR + nlap + n3 + n4_before;
    }
    for (int p = 0; p < s.NPAIR; ++p) {
        int a = (int)((s.PAIRS >> (8 * p)) & 0xF), b = (int)((s.PAIRS >> (8 * p + 4)) & 0xF);
        if (a == sl.axes[0] && b == sl.axes[1]) return 1 + s.NFIRST + p;
    }
    return -1;
}

std::string spec_name(const pk::SpecInfo& s) {
    char b[160];
    std::snprintf(b, sizeof b, "F%d_HP%d_NHH%d_D%d_F%x_P%llx_H%x_L%x_PG%d(C=%d)", s.family, s.HP, s.NHH, s.D, s.D1MASK, s.PAIRS, s.HI, s.LAP, s.PG, s.C);
    return b;
}

// theta offsets of the networks and of theta.p lie inside theta
static int plan_check_nets(pinn_engine& E) {
    // ---- nets ----
    E.netplans.resize(E.nets.size());
    for (size_t n = 0; n < E.nets.size(); ++n) {
        const Net& N = E.nets[n];
        if (N.theta_off < 0 || N.theta_off + N.nparams() > E.ntheta) return fail("descriptor: net parameters exceed ntheta");
    }
    if (E.ne > 0 && (E.p_theta_off < 0 || E.p_theta_off + E.ne > E.ntheta)) return fail("descriptor: theta.p exceeds ntheta");
    return 0;
}

// which compiled kernel evaluates which term: single-network terms join fused launch groups, equations that couple several
// networks (or are too long for the fused tape) get forward / k_expr / reverse launches per network
static int plan_assign_terms(pinn_engine& E) {
    // ---- terms -> groups ----
    auto needs_of = [&](const Term& T, int net, unsigned& need_first, std::vector<std::pair<int, int>>& need_pairs, unsigned& need_hi) -> int {
        bool general = (need_hi & GEN_FLAG) != 0;
        for (auto& s : T.slots) general = general || (s.net == net && slot_is_general(s));
        if (general) {
            // a derivative outside the fixed channel categories (mixed of order >= 3, order 5-6): ALL channels this network needs — the
            // ones accumulated so far and this term's — travel as one general multi-index set (closed under sub-multi-indices)
            std::vector<unsigned> want;
            if (need_hi & GEN_FLAG) want = gen_set((int)(need_hi & ~GEN_FLAG));
            else {
                if (need_hi >> 24) return fail("a forward-Laplacian channel cannot be combined with a general derivative set (internal)");
                for (int a = 0; a < 8; ++a) if (need_first & (1u << a)) want.push_back(1u | ((unsigned)a << 4));
                for (auto& pr : need_pairs) want.push_back(2u | ((unsigned)pr.first << 4) | ((unsigned)pr.second << 8));
                for (int a = 0; a < 6; ++a) {
                    const unsigned h = (need_hi >> (4 * a)) & 0xF;
                    if (h >= 3) { unsigned m = h; for (unsigned i = 0; i < h; ++i) m |= (unsigned)a << (4 * (i + 1)); want.push_back(m); }
                }
            }
            for (auto& s : T.slots) {
                if (s.net != net || s.order == 0) continue;
                if (s.lap) return fail("a forward-Laplacian channel cannot be combined with a general derivative set (internal)");
                for (int a = 0; a < s.order; ++a)
                    if (s.axes[a] < 0 || s.axes[a] >= E.nets[net].sizes[0]) return fail("descriptor: slot axis out of range");
                want.push_back(slot_mi(s));
            }
            need_first = 0;
            need_pairs.clear();
            need_hi = GEN_FLAG | (unsigned)gen_set_id(want);
            return 0;
        }
        for (auto& s : T.slots) {
            if (s.net != net) continue;
            if (s.lap) {
                if (s.lap >> E.nets[net].sizes[0]) return fail("descriptor: lap slot axis out of range");
                need_first |= s.lap;
                if ((need_hi >> 24) && (need_hi >> 24) != s.lap) return fail("two different Laplacian channels of one network in one kernel are not supported");
                need_hi |= s.lap << 24;
                continue;
            }
            for (int a = 0; a < s.order; ++a) {
                if (s.axes[a] < 0 || s.axes[a] >= E.nets[net].sizes[0]) return fail("descriptor: slot axis out of range");
                need_first |= 1u << s.axes[a];
            }
            if (s.order >= 2) {        // (orders 3, 4 are pure: they also need the pure second derivative of their axis)
                auto pr = std::make_pair(s.axes[0], s.axes[1]);
                if (std::find(need_pairs.begin(), need_pairs.end(), pr) == need_pairs.end()) need_pairs.push_back(pr);
            }
            if (s.order >= 3) {
                const unsigned cur = (need_hi >> (4 * s.axes[0])) & 0xF;
                if ((unsigned)s.order > cur) need_hi = (need_hi & ~(0xFu << (4 * s.axes[0]))) | ((unsigned)s.order << (4 * s.axes[0]));
            }
        }
        return 0;
    };
    auto spec_for = [&](size_t t, int net, int d, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs, unsigned need_hi,
                        const pk::SpecInfo*& sp) -> int {
        const Net& N = E.nets[net];
        (void)t;
        d = N.sizes[0];                                 // kernels are compiled per network input dimension
        const int LH = (int)N.sizes.size() - 2;
        const int HP = round_hp(N.maxhidden());
        const std::string prev = g_err;
        g_err.clear();
        sp = ensure_spec(N, need_first, need_pairs, need_hi);
        const std::string why = g_err;
        g_err = prev;
        if (!sp) {
            if (!why.empty() && why.find("PINN_NO_JIT") == std::string::npos) return fail("term " + std::to_string(t) + ": " + why);
            char b[256];
            std::snprintf(b, sizeof b,
                          "term %zu: no compiled kernel for hidden width %d (padded %d), %d hidden layers, d=%d, first-derivative axes mask 0x%x, %zu second derivatives, higher-order mask 0x%x%s; add a PINN_INSTANTIATE line in csrc/inst_*.hip",
                          t, N.maxhidden(), HP, LH, d, need_first, need_pairs.size(), need_hi, N.act == pk::ACT_SIN ? ", sin activation (PINN_INSTANTIATE*_SIN)" : (N.act == pk::ACT_MIXED ? ", per-layer tanh/sigmoid (PINN_INSTANTIATE_HI_MIX, family 1 only)" : ""));
            return fail(b);
        }
        return 0;
    };
    // pass 0: forward-Laplacian fusion (fuse_laplacian) wherever a compiled kernel carries the resulting channel set
    static const bool no_lap = std::getenv("PINN_NO_LAPLACIAN") != nullptr;
    // coupled equations: every (equation, network) pair runs the kernel of the channels THAT equation reads from the network (a network
    // entering an equation only by its value costs one channel there, not the union over all equations); PINN_COUPLED_UNION=1 restores
    // one kernel per network carrying the union (A/B measurements)
    static const bool coupled_union = std::getenv("PINN_COUPLED_UNION") != nullptr;
    auto spec_exists = [&](int net, unsigned nf, const std::vector<std::pair<int, int>>& npairs, unsigned nh) {
        const Net& N = E.nets[net];
        if (find_spec(round_hp(N.maxhidden()), (int)N.sizes.size() - 3, N.sizes[0], nf, npairs, nh, nullptr, variant_of(N.act)) != nullptr) return true;
        // shapes without ANY compiled kernel get the fused (fewer channels) form specialised at run time
        bool any_aot = false;
        for (const pk::SpecInfo& s : pk::registry())
            any_aot = any_aot || (s.family != 3 && gemm_ok(s) && s.HP == round_hp(N.maxhidden()) && s.NHH == (int)N.sizes.size() - 3 && s.D == N.sizes[0] && s.C > 1);
        if (any_aot) return false;
        const std::string prev = g_err;
        const bool ok = ensure_spec(N, nf, npairs, nh) != nullptr;
        g_err = prev;
        return ok;
    };
    bool any_general = false;
    for (auto& T : E.terms) for (auto& sl : T.slots) any_general = any_general || slot_is_general(sl);
    for (auto& N : E.nets) any_general = any_general || N.kind == 1;        // DGM kernels carry multi-index channels only
    if (!no_lap && !any_general) {
        std::vector<Term> fused(E.terms.size());
        std::vector<char> did(E.terms.size(), 0);
        std::map<int, std::pair<unsigned, std::vector<std::pair<int, int>>>> cn;      // coupled networks: union needs with fusion
        std::map<int, unsigned> ch;
        bool coupled_ok = true, any_coupled = false;
        for (size_t t = 0; t < E.terms.size(); ++t) {
            fused[t] = E.terms[t];
            did[t] = fuse_laplacian(fused[t], E.np);
            std::vector<int> nets;
            for (auto& s : fused[t].slots) if (std::find(nets.begin(), nets.end(), s.net) == nets.end()) nets.push_back(s.net);
            if (nets.size() == 1) {
                if (!did[t]) continue;
                unsigned nf = 0, nh = 0;
                std::vector<std::pair<int, int>> npairs;
                g_err.clear();
                if (needs_of(fused[t], nets[0], nf, npairs, nh) == 0 && spec_exists(nets[0], nf, npairs, nh)) E.terms[t] = fused[t];
                g_err.clear();
            } else if (nets.size() > 1) {
                any_coupled = any_coupled || did[t];
                for (int net : nets) {
                    if (needs_of(fused[t], net, cn[net].first, cn[net].second, ch[net])) { coupled_ok = false; g_err.clear(); }
                    if (!coupled_union && coupled_ok) {          // one kernel per (equation, network): check this equation's own channel set
                        unsigned nf = 0, nh = 0;
                        std::vector<std::pair<int, int>> npairs;
                        coupled_ok = needs_of(fused[t], net, nf, npairs, nh) == 0 && spec_exists(net, nf, npairs, nh);
                        g_err.clear();
                    }
                }
            }
        }
        if (any_coupled && coupled_ok) {
            for (auto& kv : cn) coupled_ok = coupled_ok && (!coupled_union || spec_exists(kv.first, kv.second.first, kv.second.second, ch[kv.first]));
            if (coupled_ok)
                for (size_t t = 0; t < E.terms.size(); ++t) {
                    std::vector<int> nets;
                    for (auto& s : fused[t].slots) if (std::find(nets.begin(), nets.end(), s.net) == nets.end()) nets.push_back(s.net);
                    if (nets.size() > 1 && did[t]) E.terms[t] = fused[t];
                }
        }
    }
    // pass 1: which terms couple several networks; union of the jet needs per network over all coupled terms
    std::vector<std::vector<int>> term_nets(E.terms.size());
    std::map<int, std::pair<unsigned, std::vector<std::pair<int, int>>>> coupled_needs;   // net -> needs
    std::map<int, unsigned> coupled_hi;
    std::vector<char> two_launch(E.terms.size(), 0);
    for (size_t t = 0; t < E.terms.size(); ++t) {
        Term& T = E.terms[t];
        for (auto& s : T.slots)
            if (std::find(term_nets[t].begin(), term_nets[t].end(), s.net) == term_nets[t].end()) term_nets[t].push_back(s.net);
        std::sort(term_nets[t].begin(), term_nets[t].end());
        if (term_nets[t].empty()) return fail("term " + std::to_string(t) + " does not reference any dependent variable");
        if (T.d > 4) return fail("term " + std::to_string(t) + ": more than 4 coordinates");
        for (int net : term_nets[t]) {                  // input maps: default = the term's coordinates in order
            const Net& N = E.nets[net];
            if (!T.inmap.count(net)) {
                if (N.sizes[0] != T.d)
                    return fail("term " + std::to_string(t) + ": network " + std::to_string(net) + " takes " + std::to_string(N.sizes[0]) +
                                " inputs but the term binds " + std::to_string(T.d) + " coordinates and the descriptor has no inmap line for it");
                std::vector<int> id(T.d);
                for (int i = 0; i < T.d; ++i) id[i] = i;
                T.inmap[net] = id;
            }
            if ((int)T.inmap[net].size() != N.sizes[0])
                return fail("term " + std::to_string(t) + ": inmap length differs from the input count of network " + std::to_string(net));
        }
        // a single-network residual too long for the fused kernel's 32-row tape takes the two-launch path (k_expr has 96 rows)
        two_launch[t] = term_nets[t].size() > 1;
        if (!two_launch[t]) {
            Term probe = T;
            analyse_static(probe, E.np);
            unsigned nf = 0, nh = 0;
            std::vector<std::pair<int, int>> npairs;
            if (needs_of(T, term_nets[t][0], nf, npairs, nh)) return 1;
            int cmin = 1 + (int)npairs.size() + ((nh >> 24) ? 1 : 0);
            for (int a = 0; a < 8; ++a) cmin += ((nf >> a) & 1) + (a < 6 && ((nh >> (4 * a)) & 0xF) >= 3) + (a < 6 && ((nh >> (4 * a)) & 0xF) >= 4);
            if (nh & GEN_FLAG) cmin = (int)gen_set((int)(nh & ~GEN_FLAG)).size();
            two_launch[t] = T.d + E.np + cmin + (int)probe.src_root.size() + (int)probe.tape_ops.size() > rp::MAX_ROWS_FUSED;
        }
        if (two_launch[t])
            for (int net : term_nets[t]) {
                auto& nd = coupled_needs[net];
                if (needs_of(T, net, nd.first, nd.second, coupled_hi[net])) return 1;
            }
    }
    std::map<std::pair<int, const pk::SpecInfo*>, int> coupled_group;        // (net, kernel) -> kind-1 group
    for (size_t t = 0; t < E.terms.size(); ++t) {
        Term& T = E.terms[t];
        if (!two_launch[t]) {
            const int net = term_nets[t][0];
            T.net = net;
            unsigned need_first = 0;
            std::vector<std::pair<int, int>> need_pairs;
            unsigned need_hi = 0;
            if (needs_of(T, net, need_first, need_pairs, need_hi)) return 1;
            const pk::SpecInfo* sp = nullptr;
            // A SMALL term (descriptor `hint`: a boundary condition at a handful of points) rides on a launch group this network already
            // has when that group's kernel carries its channels: a few extra tiles there cost less than a launch of its own (family 1,
            // one tile per wave: ~20 us per launch whatever the point count; measured on cfg1: 58 -> 37 us per evaluation).  Decided
            // before the term's own kernel is looked up, so a run-time specialised shape does not compile a kernel it will not use.
            static const bool no_ride = std::getenv("PINN_NO_RIDE") != nullptr;
            if (!no_ride && T.hint_n > 0 && !(need_hi & GEN_FLAG))
                for (size_t g = 0; g < E.groups.size() && !sp; ++g) {
                    const Group& H = E.groups[g];
                    if (H.kind != 0 || H.net != net || H.spec->family == 3 || (int)H.terms.size() >= pk::MAX_GROUP_TERMS) continue;
                    if (T.hint_n > (H.spec->family == 1 ? 256 : 64)) continue;
                    if (spec_covers(*H.spec, need_first, need_pairs, need_hi)) sp = H.spec;
                }
            if (!sp && spec_for(t, net, T.d, need_first, need_pairs, need_hi, sp)) return 1;
            T.chan_of_slot.clear();
            for (auto& s : T.slots) {
                int c = chan_of(*sp, s);
                if (c < 0) return fail("internal: slot has no channel");
                T.chan_of_slot.push_back(c);
            }
            analyse_static(T, E.np);
            for (int q : T.tape_ops)
                if (T.ops[q].code == rp::OP_DATA)
                    return fail("term " + std::to_string(t) + ": per-point data channels must be inputs of the residual (they are evaluated in the source pass), not its output");
            const int rows = T.d + E.np + sp->C + (int)T.src_root.size() + (int)T.tape_ops.size();
            if (rows > rp::MAX_ROWS_FUSED)
                return fail("term " + std::to_string(t) + ": residual expression too long for the fused kernel tape (" + std::to_string(rows) + " rows > 32)");
            if (!E.netplans[net].spec) E.netplans[net].spec = sp;
            int gi = -1;
            for (size_t g = 0; g < E.groups.size(); ++g)
                if (E.groups[g].kind == 0 && E.groups[g].net == net && E.groups[g].spec == sp && (int)E.groups[g].terms.size() < pk::MAX_GROUP_TERMS) gi = (int)g;
            if (gi < 0) {
                E.groups.emplace_back();
                gi = (int)E.groups.size() - 1;
                E.groups[gi].net = net;
                E.groups[gi].spec = sp;
            }
            T.group = gi;
            T.slot_in_group = (int)E.groups[gi].terms.size();
            E.groups[gi].terms.push_back((int)t);
            continue;
        }
        // ---- coupled term ----
        for (int net : term_nets[t])
            if (E.nets[net].kind == 1)
                return fail("term " + std::to_string(t) + ": DGM networks are supported in single-network equations (one dependent variable per equation, "
                            "residual within the fused tape)");
        if ((int)T.slots.size() > aux::EXPR_MAX_SLOTS || T.d + E.np + (int)T.slots.size() + (int)T.ops.size() > aux::EXPR_MAX_ROWS)
            return fail("term " + std::to_string(t) + ": coupled residual expression too long");
        E.coupled.emplace_back();
        Coupled& Cp = E.coupled.back();
        T.coupled = (int)E.coupled.size() - 1;
        Cp.term = (int)t;
        Cp.nets = term_nets[t];
        T.chan_of_slot.assign(T.slots.size(), -1);
        Cp.slot_net.assign(T.slots.size(), -1);
        // the kernels first: the tail decision needs every network's channel count
        std::vector<const pk::SpecInfo*> sps(Cp.nets.size(), nullptr);
        for (size_t i = 0; i < Cp.nets.size(); ++i) {
            const int net = Cp.nets[i];
            if (coupled_union) {
                if (spec_for(t, net, T.d, coupled_needs[net].first, coupled_needs[net].second, coupled_hi[net], sps[i])) return 1;
            } else {
                unsigned nf = 0, nh = 0;
                std::vector<std::pair<int, int>> npairs;
                if (needs_of(T, net, nf, npairs, nh) || spec_for(t, net, T.d, nf, npairs, nh, sps[i])) return 1;
            }
        }
        // TAIL launch: the family-2 kernel with the most channels evaluates the residual tape itself, the other networks' jets as source rows
        {
            const bool no_tail = std::getenv("PINN_NO_TAIL_FUSE") != nullptr;      // (read per plan: the tests switch it)
            int best = -1, nsrc_all = 0;
            bool data_ops = false;
            for (auto& I : T.ops) data_ops = data_ops || I.code == rp::OP_DATA;
            for (size_t i = 0; i < Cp.nets.size(); ++i) {
                nsrc_all += sps[i]->C;
                if (sps[i]->family == 2 && (best < 0 || sps[i]->C > sps[best]->C)) best = (int)i;
            }
            if (best >= 0 && !no_tail && !data_ops && !coupled_union) {
                const int nsrc = nsrc_all - sps[best]->C;
                if (T.d + E.np + sps[best]->C + nsrc + (int)T.ops.size() <= rp::MAX_ROWS_FUSED) {
                    Cp.tail = best;
                    Cp.nsrc = nsrc;
                    Cp.src_off.assign(Cp.nets.size(), -1);
                    int off = 0;
                    for (size_t i = 0; i < Cp.nets.size(); ++i)
                        if ((int)i != best) { Cp.src_off[i] = off; off += sps[i]->C; }
                }
            }
        }
        for (size_t i = 0; i < Cp.nets.size(); ++i) {
            const int net = Cp.nets[i];
            const pk::SpecInfo* sp = sps[i];
            if ((int)i == Cp.tail) {                     // a launch group of its own: one term, MODE_FUSED
                E.groups.emplace_back();
                Group& G = E.groups.back();
                G.kind = 2;
                G.net = net;
                G.spec = sp;
                if (!E.netplans[net].spec) E.netplans[net].spec = sp;
                Cp.groups.push_back((int)E.groups.size() - 1);
                G.terms.push_back((int)t);
                for (size_t si = 0; si < T.slots.size(); ++si)
                    if (T.slots[si].net == net) {
                        T.chan_of_slot[si] = chan_of(*G.spec, T.slots[si]);
                        Cp.slot_net[si] = (int)i;
                        if (T.chan_of_slot[si] < 0) return fail("internal: slot has no channel");
                    }
                continue;
            }
            const auto key = std::make_pair(net, sp);
            if (coupled_group.count(key) && (int)E.groups[coupled_group[key]].terms.size() >= pk::MAX_GROUP_TERMS) coupled_group.erase(key);
            if (!coupled_group.count(key)) {
                E.groups.emplace_back();
                Group& G = E.groups.back();
                G.kind = 1;
                G.net = net;
                G.spec = sp;
                coupled_group[key] = (int)E.groups.size() - 1;
                if (!E.netplans[net].spec) E.netplans[net].spec = sp;
            }
            Group& G = E.groups[coupled_group[key]];
            Cp.groups.push_back(coupled_group[key]);
            G.terms.push_back((int)t);
            for (size_t si = 0; si < T.slots.size(); ++si)
                if (T.slots[si].net == net) {
                    T.chan_of_slot[si] = chan_of(*G.spec, T.slots[si]);
                    Cp.slot_net[si] = (int)i;
                    if (T.chan_of_slot[si] < 0) return fail("internal: slot has no channel");
                }
        }
    }
    return 0;
}

// per network: index map theta -> packed weight image (fragment order of the kernels)
static int plan_pack_maps(pinn_engine& E) {
    // ---- per-net pack index map ----
    for (size_t n = 0; n < E.nets.size(); ++n) {
        NetPlan& NP = E.netplans[n];
        if (!NP.spec) continue;   // net unused by any term
        if (NP.spec->family == 3) continue;          // DGM kernels read theta directly
        const pk::SpecInfo& s = *NP.spec;
        const Net& N = E.nets[n];
        const int LH = s.LH, HP = s.HP, MT = s.MT, D = s.D;
        std::vector<int> loff(LH + 1);
        int o = N.theta_off;
        for (int j = 0; j <= LH; ++j) {
            loff[j] = o;
            o += N.sizes[j + 1] * N.sizes[j] + N.sizes[j + 1];
        }
        auto Widx = [&](int j, int out, int in) -> int {
            if (out >= N.sizes[j + 1] || in >= N.sizes[j]) return -1;
            return loff[j] + out + in * N.sizes[j + 1];
        };
        auto bidx = [&](int j, int out) -> int {
            if (out >= N.sizes[j + 1]) return -1;
            return loff[j] + N.sizes[j + 1] * N.sizes[j] + out;
        };
        std::vector<int> idx(s.PACKED, -1);
        for (int i = 0; i < D; ++i)
            for (int nn = 0; nn < HP; ++nn) idx[s.OFF_W1 + i * HP + nn] = Widx(0, nn, i);
        for (int l = 0; l < LH; ++l)
            for (int nn = 0; nn < HP; ++nn) idx[s.OFF_B + l * HP + nn] = bidx(l, nn);
        for (int nn = 0; nn < HP; ++nn) idx[s.OFF_WL + nn] = Widx(LH, 0, nn);
        idx[s.OFF_BL] = bidx(LH, 0);
        for (int hl = 0; hl < s.NHH && s.family == 2; ++hl)
            for (int ta = 0; ta < MT; ++ta)
                for (int tb = 0; tb < MT; ++tb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int rr = 0; rr < 4; ++rr) {
                            const int g = lane >> 4, c = lane & 15;
                            // forward [mo=ta][mi=tb][lane][rr] = W[16mo+c][16mi+4g+rr]; transposed [mi=ta][mo=tb][lane][rr] = W[16mo+4g+rr][16mi+c]
                            idx[s.OFF_WPK + ((hl * MT + ta) * MT + tb) * 256 + lane * 4 + rr] = Widx(hl + 1, 16 * ta + c, 16 * tb + 4 * g + rr);
                            idx[s.OFF_WTPK + ((hl * MT + ta) * MT + tb) * 256 + lane * 4 + rr] = Widx(hl + 1, 16 * tb + 4 * g + rr, 16 * ta + c);
                        }
        for (int hl = 0; hl < s.NHH && s.family == 1; ++hl)
            for (int m1 = 0; m1 < MT; ++m1)
                for (int rr = 0; rr < 4; ++rr)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int m2 = 0; m2 < MT; ++m2) {
                            const int g = lane >> 4, c = lane & 15;
                            // forward fragments: [mi=m1][rr][lane][mo=m2] = W[out=16mo+c][in=16mi+4g+rr]
                            idx[s.OFF_WPK + hl * HP * HP + ((m1 * 4 + rr) * 64 + lane) * MT + m2] = Widx(hl + 1, 16 * m2 + c, 16 * m1 + 4 * g + rr);
                            // transposed fragments: [mo=m1][rr][lane][mi=m2] = W[out=16mo+4g+rr][in=16mi+c]
                            idx[s.OFF_WTPK + hl * HP * HP + ((m1 * 4 + rr) * 64 + lane) * MT + m2] = Widx(hl + 1, 16 * m1 + 4 * g + rr, 16 * m2 + c);
                        }
        if (s.BFX && s.NHH > 16) return fail("networks with more than 17 hidden layers are not supported by the 64-wide split-operand kernels");
        NP.npacked = s.PACKED;
        NP.d_packed = (float*)plat_malloc(sizeof(float) * s.PACKED);
        NP.d_pack_idx = (int*)plat_malloc(sizeof(int) * s.PACKED);
        if (!NP.d_packed || !NP.d_pack_idx) return fail("device allocation failed (packed weights)");
        plat_h2d(NP.d_pack_idx, idx.data(), sizeof(int) * s.PACKED, E.stream);
        plat_sync(E.stream);
        NP.h_pack_idx = idx;
    }
    // inverse map (theta element -> its positions in the packed images): the resident optimiser's update kernel writes the new value
    // of every parameter straight into the images, so the next evaluation needs no pack launch (engine.cpp: pinn_adam_steps)
    E.inv_ok = E.nets.size() <= (size_t)aux::MAX_PACK_NETS;
    if (E.inv_ok) {
        std::vector<std::vector<int>> inv((size_t)E.ntheta);
        for (size_t n = 0; n < E.nets.size(); ++n) {
            const NetPlan& NP = E.netplans[n];
            if (!NP.spec || NP.spec->family == 3) continue;
            if (NP.npacked >= (1 << 24)) { E.inv_ok = false; break; }
            for (int q = 0; q < NP.npacked; ++q)
                if (NP.h_pack_idx[q] >= 0) inv[(size_t)NP.h_pack_idx[q]].push_back((int)((n << 24) | (unsigned)q));
        }
        if (E.inv_ok) {
            std::vector<int> ptr{0}, pos;
            E.max_inv_pos = 0;
            for (auto& v : inv) { pos.insert(pos.end(), v.begin(), v.end()); ptr.push_back((int)pos.size()); E.max_inv_pos = std::max(E.max_inv_pos, (int)v.size()); }
            E.d_inv_ptr = (int*)plat_malloc(sizeof(int) * ptr.size());
            E.d_inv_pos = (int*)plat_malloc(sizeof(int) * std::max<size_t>(pos.size(), 1));
            if (!E.d_inv_ptr || !E.d_inv_pos) return fail("device allocation failed (inverse pack map)");
            plat_h2d(E.d_inv_ptr, ptr.data(), sizeof(int) * ptr.size(), E.stream);
            if (!pos.empty()) plat_h2d(E.d_inv_pos, pos.data(), sizeof(int) * pos.size(), E.stream);
            plat_sync(E.stream);
        }
    }
    return 0;
}

// A fused-tape program (rows [coordinates dt | params np | jet channels C | sources nsrc | ops]) whose output is an AFFINE function of
// the jet channels and sources with constant coefficients needs no interpreter: the kernel evaluates it with C + nsrc FMAs and seeds
// the reverse sweep with the coefficients (TermDev::linear).  True for every Dirichlet term u - g(x), for lap u - f(x), for linear
// PDEs with fixed coefficients.  Anything that touches a coordinate or parameter row directly, or multiplies two non-constant rows,
// keeps the tape.  PINN_NO_LINEAR=1 disables the shortcut (A/B measurements).
static bool detect_linear(const std::vector<rp::Instr>& prog, int dt, int np, int C, int nsrc, int out_row, LinearForm& L) {
    static const bool off = std::getenv("PINN_NO_LINEAR") != nullptr;
    if (off || C > pk::LIN_MAX_C || nsrc > pk::LIN_MAX_SRC) return false;
    const int r0 = dt + np + C + nsrc, nb = C + nsrc;
    struct Aff { bool ok; std::vector<double> c; double k; };
    auto row_form = [&](int row, const std::vector<Aff>& ops) -> Aff {
        Aff a{true, std::vector<double>((size_t)nb, 0.0), 0.0};
        if (row < dt + np) { a.ok = false; return a; }                 // coordinates / parameters: not covered
        if (row < r0) { a.c[(size_t)(row - dt - np)] = 1.0; return a; }
        return ops[(size_t)(row - r0)];
    };
    std::vector<Aff> ops;
    for (const rp::Instr& I : prog) {
        Aff o{true, std::vector<double>((size_t)nb, 0.0), 0.0};
        const Aff a = rp::is_nullary(I.code) ? o : row_form(I.a, ops);
        const Aff b = rp::is_binary(I.code) ? row_form(I.b, ops) : o;
        auto is_const = [&](const Aff& f) { if (!f.ok) return false; for (double v : f.c) if (v != 0.0) return false; return true; };
        switch (I.code) {
            case rp::OP_CONST: o.k = I.imm; break;
            case rp::OP_ADD: o.ok = a.ok && b.ok; for (int i = 0; i < nb; ++i) o.c[i] = a.c[i] + b.c[i]; o.k = a.k + b.k; break;
            case rp::OP_SUB: o.ok = a.ok && b.ok; for (int i = 0; i < nb; ++i) o.c[i] = a.c[i] - b.c[i]; o.k = a.k - b.k; break;
            case rp::OP_NEG: o.ok = a.ok; for (int i = 0; i < nb; ++i) o.c[i] = -a.c[i]; o.k = -a.k; break;
            case rp::OP_ADDC: o = a; o.k += I.imm; break;
            case rp::OP_MULC: o.ok = a.ok; for (int i = 0; i < nb; ++i) o.c[i] = a.c[i] * I.imm; o.k = a.k * I.imm; break;
            case rp::OP_MUL:
                if (is_const(a) && b.ok) { o = b; for (double& v : o.c) v *= a.k; o.k *= a.k; }
                else if (is_const(b) && a.ok) { o = a; for (double& v : o.c) v *= b.k; o.k *= b.k; }
                else o.ok = false;
                break;
            default: o.ok = false;
        }
        ops.push_back(o);
    }
    const Aff f = row_form(out_row, ops);
    if (!f.ok) return false;
    L.k = (float)f.k;
    for (int c = 0; c < pk::LIN_MAX_C; ++c) L.a[c] = c < C ? (float)f.c[(size_t)c] : 0.f;
    for (int j = 0; j < pk::LIN_MAX_SRC; ++j) L.b[j] = j < nsrc ? (float)f.c[(size_t)(C + j)] : 0.f;
    return true;
}

// per launch group: slabs, scratch, loss partials, programs, slab -> theta reduce rows, kernel arguments
static int plan_group_buffers(pinn_engine& E) {
    // ---- per-group buffers and reduce maps ----
    int total_terms = (int)E.terms.size();
    for (auto& G : E.groups) {
        const pk::SpecInfo& s = *G.spec;
        const Net& N = E.nets[G.net];
        const int LH = s.LH, HP = s.HP, MT = s.MT, D = s.D;
        (void)HP;
        // (PINN_WG_PER_CU=1 limits the grid to one workgroup per CU: occupancy experiments)
        static const int wg_cap = [] { const char* e = std::getenv("PINN_WG_PER_CU"); return e ? std::atoi(e) : 0; }();
        G.max_blocks = E.ncu * ((wg_cap > 0 && wg_cap < s.WG_PER_CU) ? wg_cap : s.WG_PER_CU);
        plat_event_create(G.ev_a);
        plat_event_create(G.ev_b);
        const size_t nw = (size_t)E.ncu * std::max(s.WG_PER_CU, s.WG_FWD) * s.NW;      // (loss-only launches run more workgroups per CU)
        const int slab_floats = s.family == 3 ? ((N.nparams() + pk::MAX_PARAMS + 63) / 64) * 64 : s.SLAB;
        G.slab_floats = slab_floats;
        G.d_slabs = (float*)plat_malloc(sizeof(float) * (size_t)G.max_blocks * slab_floats);
        G.d_losspart = (double*)plat_malloc(sizeof(double) * nw * total_terms);
        // family 3: the scratch rows are sized by the point sets (retile)
        G.d_scratch = s.family == 3 ? (float*)plat_malloc(4) : (float*)plat_malloc(sizeof(float) * (s.family == 2 ? (size_t)G.max_blocks : nw) * s.SCR);
        if (!G.d_slabs || !G.d_losspart || !G.d_scratch) return fail("device allocation failed (group buffers)");
        // columns of terms this group does not own are never written by its kernel but are summed by the reduction
        plat_memset(G.d_losspart, 0, sizeof(double) * nw * total_terms, E.stream);
        // programs (rows remapped to the kernel's channel numbering)
        std::vector<rp::Instr> prog;
        for (int ti : G.terms) {
            Term& T = E.terms[ti];
            if (G.kind == 1) {           // coupled terms: the tape runs in k_expr (or in the equation's tail launch), not in this kernel
                G.prog_off.push_back(0); G.prog_n.push_back(0); G.out_row.push_back(0);
                continue;
            }
            if (G.kind == 2) {
                // tail launch of a coupled equation: the WHOLE tape (nothing is hoisted), rows
                // [coordinates d | params np | this kernel's channels C | the other kernels' channels, network after network | ops]
                const Coupled& Cp = E.coupled[T.coupled];
                const int S = (int)T.slots.size();
                const int rslot0 = T.d + E.np, rop0 = rslot0 + S;
                auto remap2 = [&](int row) -> int {
                    if (row < rslot0) return row;
                    if (row < rop0) {
                        const int si = row - rslot0, i = Cp.slot_net[si];
                        return i == Cp.tail ? T.d + E.np + T.chan_of_slot[si] : T.d + E.np + s.C + Cp.src_off[i] + T.chan_of_slot[si];
                    }
                    return T.d + E.np + s.C + Cp.nsrc + (row - rop0);
                };
                G.prog_off.push_back((int)prog.size());
                G.prog_n.push_back((int)T.ops.size());
                std::vector<rp::Instr> mine;
                for (size_t q = 0; q < T.ops.size(); ++q) {
                    rp::Instr I = T.ops[q];
                    const int lim = rop0 + (int)q;
                    if (!rp::is_nullary(I.code) && (I.a < 0 || I.a >= lim)) return fail("descriptor: op operand row out of range");
                    if (rp::is_binary(I.code) && (I.b < 0 || I.b >= lim)) return fail("descriptor: op operand row out of range");
                    I.a = rp::is_nullary(I.code) ? 0 : remap2(I.a);
                    I.b = rp::is_binary(I.code) ? remap2(I.b) : 0;
                    rp::finalize(I);
                    prog.push_back(I);
                    mine.push_back(I);
                }
                if (T.out_row < 0 || T.out_row >= rop0 + (int)T.ops.size()) return fail("descriptor: out row out of range");
                T.lin = LinearForm();
                T.linear = detect_linear(mine, T.d, E.np, s.C, Cp.nsrc, remap2(T.out_row), T.lin);
                G.out_row.push_back(remap2(T.out_row));
                continue;
            }
            const int S = (int)T.slots.size();
            const int rslot0 = T.d + E.np, rop0 = rslot0 + S;
            const int nsrc = (int)T.src_root.size();
            std::vector<int> tape_pos(T.ops.size(), -1);
            for (size_t i = 0; i < T.tape_ops.size(); ++i) tape_pos[T.tape_ops[i]] = (int)i;
            // fused tape rows: [coordinates d | params np | jet channels C | sources nsrc | tape ops]
            auto remap = [&](int row) -> int {
                if (row < rslot0) return row;
                if (row < rop0) return T.d + E.np + T.chan_of_slot[row - rslot0];
                const int q = row - rop0;
                if (T.src_of_op[q] >= 0) return T.d + E.np + s.C + T.src_of_op[q];
                return T.d + E.np + s.C + nsrc + tape_pos[q];
            };
            G.prog_off.push_back((int)prog.size());
            G.prog_n.push_back((int)T.tape_ops.size());
            std::vector<rp::Instr> mine;
            for (int q : T.tape_ops) {
                rp::Instr I = T.ops[q];
                I.a = rp::is_nullary(I.code) ? 0 : remap(I.a);
                I.b = rp::is_binary(I.code) ? remap(I.b) : 0;
                rp::finalize(I);
                prog.push_back(I);
                mine.push_back(I);
            }
            T.lin = LinearForm();
            T.linear = (s.family == 2 || s.family == 1) && detect_linear(mine, T.d, E.np, s.C, nsrc, remap(T.out_row), T.lin);
            if (nsrc > 0) {
                T.d_src_prog = (rp::Instr*)plat_malloc(sizeof(rp::Instr) * T.src_prog.size());
                if (!T.d_src_prog) return fail("device allocation failed (source programs)");
                plat_h2d(T.d_src_prog, T.src_prog.data(), sizeof(rp::Instr) * T.src_prog.size(), E.stream);
            }
            if (T.out_row < 0 || T.out_row >= rop0 + (int)T.ops.size()) return fail("descriptor: out row out of range");
            G.out_row.push_back(remap(T.out_row));
        }
        G.d_prog = (rp::Instr*)plat_malloc(sizeof(rp::Instr) * std::max<size_t>(prog.size(), 1));
        if (!G.d_prog) return fail("device allocation failed (programs)");
        if (!prog.empty()) plat_h2d(G.d_prog, prog.data(), sizeof(rp::Instr) * prog.size(), E.stream);
        // reduce map (CSR): theta element -> slab offsets that must be summed (1 for workgroup-shared sections,
        // 4 for per-wave sections), see Spec in pinn_kernels.hpp
        std::vector<int> row_theta, row_ptr{0}, ent;
        auto add_row = [&](int theta_idx, int off, bool shared) {
            row_theta.push_back(theta_idx);
            if (shared) ent.push_back(off);
            else
                for (int w = 0; w < 4; ++w) ent.push_back(s.SH + w * s.PW + off);
            row_ptr.push_back((int)ent.size());
        };
        const bool coop = s.COOP != 0;
        std::vector<int> loff(LH + 1);
        int o = N.theta_off;
        for (int j = 0; j <= LH && s.family != 3; ++j) {        // (Dense chains; a DGM net's `sizes` is {d, modes, 1} whatever its depth)
            loff[j] = o;
            o += N.sizes[j + 1] * N.sizes[j] + N.sizes[j + 1];
        }
        if (s.family == 3) {                                    // slab = the network's parameters in theta order, then the PDE-parameter sums
            for (int e = 0; e < N.nparams(); ++e) add_row(N.theta_off + e, e, true);
            for (int j = 0; j < E.ne; ++j) add_row(E.p_theta_off + j, N.nparams() + j, true);
        }
        if (s.family == 2) {                                    // every slab entry has exactly one writer wave
            for (int in = 0; in < D; ++in)
                for (int out = 0; out < N.sizes[1]; ++out) add_row(loff[0] + out + in * N.sizes[1], s.O_W1 + in * s.HP + out, true);
            for (int l = 0; l < LH; ++l)
                for (int out = 0; out < N.sizes[l + 1]; ++out)
                    add_row(loff[l] + N.sizes[l + 1] * N.sizes[l] + out, s.O_BFRH + l * s.HP + out, true);
            for (int hl = 0; hl < s.NHH; ++hl) {
                const int j = hl + 1;
                for (int in = 0; in < N.sizes[j]; ++in)
                    for (int out = 0; out < N.sizes[j + 1]; ++out) {
                        const int to = out / 16, i = out % 16, g = i / 4, r = i % 4;
                        // fp32 dW GEMM: the B operand is four consecutive inputs per lane, so tile ti holds inputs 64 (ti / 4) + 4 c + ti % 4;
                        // transpose-read dW GEMM (SpecInfo::BFX_DW): plain tiles, column c of tile ti = input 16 ti + c
                        const int ti = s.BFX_DW ? in / 16 : (in / 64) * 4 + (in % 4), c = s.BFX_DW ? in % 16 : (in % 64) / 4;
                        add_row(loff[j] + out + in * N.sizes[j + 1], s.O_WBAR + (((hl * MT + to) * MT + ti) * 64 + g * 16 + c) * 4 + r, true);
                    }
            }
            for (int in = 0; in < N.sizes[LH]; ++in) add_row(loff[LH] + in, s.O_WL + in, true);
            add_row(loff[LH] + N.sizes[LH], s.O_BL, true);
            for (int j = 0; j < E.ne; ++j) add_row(E.p_theta_off + j, s.O_P + j, true);
        }
        for (int in = 0; in < D && s.family == 1; ++in)          // layer 0: W (n1 x d)
            for (int out = 0; out < N.sizes[1]; ++out)
                add_row(loff[0] + out + in * N.sizes[1], s.O_W1 + (in * MT + out % MT) * 16 + out / MT, false);
        for (int out = 0; out < N.sizes[1] && s.family == 1; ++out)              // bias of hidden layer 0
            add_row(loff[0] + N.sizes[1] * N.sizes[0] + out, s.O_BFR0 + (out % MT) * 16 + out / MT, false);
        for (int hl = 0; hl < s.NHH && s.family == 1; ++hl) {
            const int j = hl + 1;
            for (int in = 0; in < N.sizes[j]; ++in)
                for (int out = 0; out < N.sizes[j + 1]; ++out) {
                    const int to = out % MT, i = out / MT, g = i / 4, r = i % 4;
                    const int ti = in % MT, c = in / MT;
                    add_row(loff[j] + out + in * N.sizes[j + 1], s.O_WBAR + hl * s.HP * s.HP + ((to * MT + ti) * 64 + g * 16 + c) * 4 + r, coop);
                }
            for (int out = 0; out < N.sizes[j + 1]; ++out)
                add_row(loff[j] + N.sizes[j + 1] * N.sizes[j] + out, s.O_BFRH + (hl * MT + out % MT) * 16 + out / MT, coop);
        }
        for (int in = 0; s.family == 1 && in < N.sizes[LH]; ++in)                // W_out (1 x nLH)   (family first: a DGM net's `sizes` has three entries)
            add_row(loff[LH] + in, s.O_WL + ((in / 16) * 4 + (in % 16) / 4) * 4 + in % 4, false);
        if (s.family == 1) {
            add_row(loff[LH] + N.sizes[LH], s.O_BL, false);
            for (int j = 0; j < E.ne; ++j) add_row(E.p_theta_off + j, s.O_P + j, false);
        }
        G.nent = slab_floats;            // stage 1 is dense over slab offsets
        G.row_theta = row_theta;
        G.row_ptr = row_ptr;
        G.row_off = ent;                 // slab offsets of the contributions
        if (s.family != 1 && ent.size() == row_theta.size()) {          // one slab entry per theta element: inverse map for aux::k_reduce_one
            std::vector<int> inv((size_t)slab_floats, -1);
            std::vector<char> seen((size_t)E.ntheta, 0);
            size_t covered = 0;
            for (size_t r = 0; r < row_theta.size(); ++r) {
                inv[(size_t)ent[r]] = row_theta[r];
                if (!seen[(size_t)row_theta[r]]) { seen[(size_t)row_theta[r]] = 1; ++covered; }
            }
            G.d_ent_theta = (int*)plat_malloc(sizeof(int) * inv.size());
            if (!G.d_ent_theta) return fail("device allocation failed (reduce map)");
            plat_h2d(G.d_ent_theta, inv.data(), sizeof(int) * inv.size(), E.stream);
            G.ent_covers_theta = covered == (size_t)E.ntheta && covered == row_theta.size();
        }
        G.d_tmp = (double*)plat_malloc(sizeof(double) * (size_t)REDUCE_SPLIT * (G.nent + total_terms));
        if (!G.d_tmp) return fail("device allocation failed (reduce map)");
        // static part of the launch arguments
        pk::GroupArgs& ga = G.ga;
        std::memset(&ga, 0, sizeof ga);
        ga.packed = E.netplans[G.net].d_packed;
        ga.params = E.d_params;
        ga.prog = G.d_prog;
        ga.slabs = G.d_slabs;
        ga.losspart = G.d_losspart;
        ga.scratch = G.d_scratch;
        ga.nterms_total = total_terms;
        ga.nterms = (int)G.terms.size();
        ga.nparams = E.np;
        ga.nparams_estim = E.ne;
        ga.act = N.act;
        ga.act_layers = N.act_layers;
        if (s.family == 3) {
            ga.dgm_modes = N.sizes[1];
            ga.dgm_slab = slab_floats;
            ga.dgm_nparams = N.nparams();
        }
    }
    return 0;
}

// coupled equations: tape in descriptor row numbering (slots are direct inputs of k_expr), pseudo-group reduce rows
static int plan_coupled_programs(pinn_engine& E) {
    const int total_terms = (int)E.terms.size();
    // ---- coupled equations: tape in descriptor row numbering (slots are direct inputs of k_expr) ----
    for (auto& Cp : E.coupled) {
        Term& T = E.terms[Cp.term];
        const int lim0 = T.d + E.np + (int)T.slots.size();
        std::vector<rp::Instr> prog = T.ops;
        for (size_t q = 0; q < prog.size(); ++q) {
            rp::Instr& I = prog[q];
            const int lim = lim0 + (int)q;
            if (!rp::is_nullary(I.code) && (I.a < 0 || I.a >= lim)) return fail("descriptor: op operand row out of range");
            if (rp::is_binary(I.code) && (I.b < 0 || I.b >= lim)) return fail("descriptor: op operand row out of range");
            if (rp::is_nullary(I.code)) I.a = 0;
            if (!rp::is_binary(I.code)) I.b = 0;
            rp::finalize(I);
        }
        if (T.out_row < 0 || T.out_row >= lim0 + (int)prog.size()) return fail("descriptor: out row out of range");
        Cp.d_prog = (rp::Instr*)plat_malloc(sizeof(rp::Instr) * std::max<size_t>(prog.size(), 1));
        Cp.d_tmp = (double*)plat_malloc(sizeof(double) * (size_t)REDUCE_SPLIT * (16 + total_terms));
        if (!Cp.d_prog || !Cp.d_tmp) return fail("device allocation failed (coupled term)");
        if (!prog.empty()) plat_h2d(Cp.d_prog, prog.data(), sizeof(rp::Instr) * prog.size(), E.stream);
        Cp.row_ptr = {0};
        for (int j = 0; j < E.ne; ++j) {                 // dL/dp partials: 4 per-wave entries per parameter
            Cp.row_theta.push_back(E.p_theta_off + j);
            for (int w = 0; w < 4; ++w) Cp.row_off.push_back(w * 4 + j);
            Cp.row_ptr.push_back((int)Cp.row_off.size());
        }
        plat_sync(E.stream);
    }
    return 0;
}

// theta element -> (group, slab entry) contributions, group order fixed: the deterministic stage-2 gather
static int plan_global_reduce_map(pinn_engine& E) {
    // ---- global reduce map: theta element -> (group, slab entry) contributions, group order fixed ----
    if ((int)(E.groups.size() + E.coupled.size()) > aux::MAX_GROUPS) return fail("too many kernel launch groups for one engine");
    {
        std::vector<std::vector<std::pair<int, int>>> contrib((size_t)E.ntheta);
        for (size_t g = 0; g < E.groups.size(); ++g) {
            const Group& G = E.groups[g];
            for (size_t r = 0; r < G.row_theta.size(); ++r)
                for (int e = G.row_ptr[r]; e < G.row_ptr[r + 1]; ++e) contrib[G.row_theta[r]].push_back({(int)g, G.row_off[e]});
        }
        for (size_t c = 0; c < E.coupled.size(); ++c) {          // pseudo-groups after the kernel groups
            const Coupled& Cp = E.coupled[c];
            for (size_t r = 0; r < Cp.row_theta.size(); ++r)
                for (int e = Cp.row_ptr[r]; e < Cp.row_ptr[r + 1]; ++e)
                    contrib[Cp.row_theta[r]].push_back({(int)(E.groups.size() + c), Cp.row_off[e]});
        }
        std::vector<int> ptr{0}, grp, ent;
        E.max_contrib = 0;
        for (auto& c : contrib) {
            E.max_contrib = std::max(E.max_contrib, (int)c.size());
            for (auto& pr : c) { grp.push_back(pr.first); ent.push_back(pr.second); }
            ptr.push_back((int)grp.size());
        }
        // order of the one-stage reduction's threads: by (group, slab entry) of an element's first contribution (aux::Reduce2Args::perm)
        {
            std::vector<int> perm((size_t)E.ntheta);
            for (int r = 0; r < (int)E.ntheta; ++r) perm[(size_t)r] = r;
            std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) {
                const auto kx = contrib[(size_t)x].empty() ? std::make_pair(1 << 30, 0) : contrib[(size_t)x][0];
                const auto ky = contrib[(size_t)y].empty() ? std::make_pair(1 << 30, 0) : contrib[(size_t)y][0];
                return kx < ky;
            });
            for (size_t k = 0; k < E.terms.size(); ++k) perm.push_back((int)(E.ntheta + (int64_t)k));
            E.d_red_perm = (int*)plat_malloc(sizeof(int) * perm.size());
            if (!E.d_red_perm) return fail("device allocation failed (reduction order)");
            plat_h2d(E.d_red_perm, perm.data(), sizeof(int) * perm.size(), E.stream);
        }
        E.d_gr_ptr = (int*)plat_malloc(sizeof(int) * ptr.size());
        E.d_gr_grp = (int*)plat_malloc(sizeof(int) * std::max<size_t>(grp.size(), 1));
        E.d_gr_ent = (int*)plat_malloc(sizeof(int) * std::max<size_t>(ent.size(), 1));
        if (!E.d_gr_ptr || !E.d_gr_grp || !E.d_gr_ent) return fail("device allocation failed (global reduce map)");
        plat_h2d(E.d_gr_ptr, ptr.data(), sizeof(int) * ptr.size(), E.stream);
        if (!grp.empty()) {
            plat_h2d(E.d_gr_grp, grp.data(), sizeof(int) * grp.size(), E.stream);
            plat_h2d(E.d_gr_ent, ent.data(), sizeof(int) * ent.size(), E.stream);
        }
        plat_sync(E.stream);
    }
    return 0;
}

// fused launch groups of the same network with the same slab layout (e.g. the interior group and the boundary group of one PINN)
// can share one slab set: the later group's workgroups add their sums onto what the earlier group stored (GroupArgs::chain), and the
// reduction reads one set instead of two.  Decided per evaluation in run_loss_grad (the head must have been launched, and must
// have at least as many workgroups); here: which earlier group is a compatible head.
static int plan_chain_groups(pinn_engine& E) {
    for (size_t g = 0; g < E.groups.size(); ++g) {
        Group& G = E.groups[g];
        G.chain_to = -1;
        if (G.kind != 0 || G.spec->family != 2) continue;
        for (size_t h = 0; h < g; ++h) {
            const Group& H = E.groups[h];
            if (H.kind != 0 || H.spec->family != 2 || H.chain_to >= 0 || H.net != G.net) continue;
            if (H.spec->SLAB != G.spec->SLAB || H.nent != G.nent || H.max_blocks < G.max_blocks) continue;
            if (H.row_theta != G.row_theta || H.row_ptr != G.row_ptr || H.row_off != G.row_off) continue;
            G.chain_to = (int)h;
            break;
        }
    }
    return 0;
}

// A chained pair (head, tail) for which a MERGED kernel is compiled (pk::pair_registry: both members' bodies in one persistent kernel)
// runs as ONE launch: the wave's weight-gradient accumulators stay in registers from the head's tiles into the tail's — no slab store
// + reload between two chained launches, one ramp, one epilogue.  PINN_NO_MERGE=1 keeps the two chained launches (A/B measurements).
static bool same_member(const pk::SpecInfo& x, const pk::SpecInfo& y) {
    return x.family == 2 && y.family == 2 && x.gemm == y.gemm && x.HP == y.HP && x.NHH == y.NHH && x.D == y.D && x.D1MASK == y.D1MASK && x.PAIRS == y.PAIRS &&
           x.NPAIR == y.NPAIR && x.PG == y.PG && x.HI == y.HI && x.LAP == y.LAP && x.ngen == 0 && y.ngen == 0 && x.NW == y.NW && x.SLAB == y.SLAB;
}
static int plan_merge_groups(pinn_engine& E) {
    for (size_t g = 0; g < E.groups.size(); ++g) {
        Group& G = E.groups[g];
        if (G.chain_to < 0 || G.merged >= 0) continue;
        Group& H = E.groups[G.chain_to];
        if (H.merged >= 0 || H.terms.size() + G.terms.size() > (size_t)pk::MAX_GROUP_TERMS) continue;
        if (E.nets[H.net].act != pk::ACT_TANH && E.nets[H.net].act != pk::ACT_SIGMOID) continue;
        for (const pk::PairInfo& p : pk::pair_registry()) {
            if (!same_member(p.a, *H.spec) || !same_member(p.b, *G.spec)) continue;
            MergedUnit M;
            M.head = G.chain_to; M.tail = (int)g; M.pair = &p;
            M.max_blocks = std::min(E.ncu * p.WG_PER_CU, std::min(H.max_blocks, G.max_blocks));
            M.d_scratch = (float*)plat_malloc(sizeof(float) * (size_t)M.max_blocks * std::max(p.SCR, 1));
            const size_t lp = sizeof(double) * (size_t)E.ncu * std::max(p.WG_PER_CU, p.WG_FWD) * p.NW * E.terms.size();
            M.d_losspart = (double*)plat_malloc(lp);
            if (!M.d_scratch || !M.d_losspart) return fail("device allocation failed (merged launch buffers)");
            plat_memset(M.d_losspart, 0, lp, E.stream);
            H.merged = G.merged = (int)E.merged.size();
            E.merged.push_back(M);
            break;
        }
    }
    return 0;
}

int build_plan(pinn_engine& E) {
    GemmScope gs(E.gemm);
    if (plan_check_nets(E) || plan_assign_terms(E) || plan_pack_maps(E) || plan_group_buffers(E) || plan_coupled_programs(E) ||
        plan_chain_groups(E) || plan_merge_groups(E) || plan_global_reduce_map(E))
        return 1;
    return 0;
}

void free_plan(pinn_engine& E) {
    for (auto& T : E.terms) {
        plat_free(T.d_src_prog); T.d_src_prog = nullptr;
        plat_free(T.d_src); T.d_src = nullptr; T.src_cap = 0;
        T.net = T.group = T.slot_in_group = T.coupled = -1;
    }
    for (auto& G : E.groups) {
        plat_free(G.d_prog); plat_free(G.d_slabs); plat_free(G.d_losspart); plat_free(G.d_scratch); plat_free(G.d_rec);
        plat_free(G.d_tmp); plat_free(G.d_ent_theta);
        plat_event_destroy(G.ev_a); plat_event_destroy(G.ev_b);
    }
    for (auto& M : E.merged) { plat_free(M.d_scratch); plat_free(M.d_losspart); }
    for (auto& Cp : E.coupled) {
        for (size_t i = 0; i < Cp.d_jets.size(); ++i)
            if (Cp.tail < 0 || (int)i == Cp.tail) { plat_free(Cp.d_jets[i]); plat_free(Cp.d_ubar[i]); }      // (the others point into the *_all arrays)
        plat_free(Cp.d_jets_all); plat_free(Cp.d_ubar_all);
        plat_free(Cp.d_prog); plat_free(Cp.d_losspart); plat_free(Cp.d_pslab); plat_free(Cp.d_tmp);
    }
    for (auto& N : E.netplans) { plat_free(N.d_packed); plat_free(N.d_pack_idx); }
    plat_free(E.d_gr_ptr); plat_free(E.d_gr_grp); plat_free(E.d_gr_ent); plat_free(E.d_inv_ptr); plat_free(E.d_inv_pos); plat_free(E.d_red_perm);
    E.d_red_perm = nullptr;
    E.own_blocks = 0;                 // (the training kernel's thread map belongs to the plan)
    E.d_gr_ptr = E.d_gr_grp = E.d_gr_ent = E.d_inv_ptr = E.d_inv_pos = nullptr;
    E.inv_ok = false;
    E.groups.clear(); E.merged.clear(); E.coupled.clear(); E.netplans.clear();
}

// refresh tile tables after a point set changed
void retile(pinn_engine& E, int gi) {
    Group& G = E.groups[gi];
    const pk::SpecInfo& s = *G.spec;
    int tile = 0;
    for (size_t j = 0; j < G.terms.size(); ++j) {
        Term& T = E.terms[G.terms[j]];
        pk::TermDev& td = G.ga.terms[j];
        td.pts = T.d_pts;
        td.N = (int)T.n;
        td.tile0 = tile;
        td.ntiles = (int)((T.n + s.TP - 1) / s.TP);
        td.prog_off = G.prog_off[j];
        td.nops = G.prog_n[j];
        td.out_row = G.out_row[j];
        td.term_id = G.terms[j];
        td.scale = 0.f;
        td.out = nullptr;
        td.in = nullptr;
        td.pw = (T.pw_n == T.n && T.pw_n > 0) ? T.d_pw : nullptr;
        td.src = (G.kind == 0) ? T.d_src : nullptr;
        td.nsrc = (G.kind == 0) ? (int)T.src_root.size() : 0;
        td.src_bar = nullptr;
        td.linear = ((G.kind == 0 || G.kind == 2) && T.linear) ? 1 : 0;
        if (G.kind == 2) {               // tail launch of a coupled equation: the other networks' jets are the source rows, their seeds come back
            const Coupled& Cp = E.coupled[T.coupled];
            td.src = Cp.d_jets_all;
            td.nsrc = Cp.nsrc;
            td.src_bar = Cp.d_ubar_all;
        }
        td.lin_k = T.lin.k;
        for (int c = 0; c < pk::LIN_MAX_C; ++c) td.lin_a[c] = T.lin.a[c];
        for (int j = 0; j < pk::LIN_MAX_SRC; ++j) td.lin_b[j] = T.lin.b[j];
        {
            const std::vector<int>& m = T.inmap.at(G.net);
            td.dt = T.d;
            td.hetero = ((int)m.size() != T.d);
            for (int i = 0; i < 4; ++i) {
                td.imap[i] = i < (int)m.size() ? m[i] : 0;
                if (i < (int)m.size() && m[i] != i) td.hetero = 1;
            }
        }
        if (G.kind == 1 || G.kind == 2) {               // coupled term: this network's jet / seed buffers
            const Coupled& Cp = E.coupled[T.coupled];
            for (size_t i = 0; i < Cp.groups.size(); ++i)
                if (Cp.groups[i] == gi && i < Cp.d_jets.size()) { td.out = Cp.d_jets[i]; td.in = Cp.d_ubar[i]; }
        }
        tile += td.ntiles;
    }
    G.ga.ntiles = tile;
    G.blocks = std::max(1, std::min(G.max_blocks, s.family == 1 ? (tile + 3) / 4 : tile));
    if (s.family == 3) {                 // point-major scratch rows [ROWS][tiles x 64]
        const size_t need = (size_t)s.dgm_rows * (size_t)tile * 64;
        if (need > G.scratch_cap) {
            plat_sync(E.stream);
            plat_free(G.d_scratch);
            G.d_scratch = (float*)plat_malloc(sizeof(float) * need);
            G.scratch_cap = G.d_scratch ? need : 0;
        }
        G.ga.scratch = G.d_scratch;
        G.ga.dgm_npad = tile * 64;
    }
    // coupled groups: keep the forward launch's records in HBM when they fit the budget (default 96 GB per handle, PINN_REC_GB)
    G.use_rec = false;
    // (PINN_REC_MIN_C: records only for kernels with at least this many jet channels — below, the reverse launch repeats the forward pass;
    // with the split-operand GEMMs a 1-2 channel forward pass is cheaper than 2-4 KB of records per point through HBM and back)
    static const int rec_min_c = [] { const char* e = std::getenv("PINN_REC_MIN_C"); return e ? std::atoi(e) : 1; }();
    if (G.kind == 1 && s.family == 2 && s.REC > 0 && s.C >= rec_min_c) {
        static const double budget_gb = [] { const char* e = std::getenv("PINN_REC_GB"); return e ? std::atof(e) : 96.0; }();
        const size_t niter = ((size_t)tile + G.blocks - 1) / G.blocks;
        const size_t slots = niter * (size_t)G.blocks;              // dummy tiles of the last round get slots of their own
        const double gb = (double)slots * s.REC * 4.0 / 1e9;
        double others = 0.0;
        for (auto& H : E.groups) if (&H != &G && H.d_rec) others += (double)H.rec_slots * H.spec->REC * 4.0 / 1e9;
        if (gb + others <= budget_gb) {
            if (slots > G.rec_slots) {
                plat_sync(E.stream);
                plat_free(G.d_rec);
                G.d_rec = (float*)plat_malloc(sizeof(float) * slots * (size_t)s.REC);
                G.rec_slots = G.d_rec ? slots : 0;
            }
            G.use_rec = G.d_rec != nullptr;
        }
    }
    G.ga.rec = G.d_rec;
}

}  // namespace pe
