// engine.cpp — host side of the C ABI: descriptor parsing, kernel-plan construction, launches.
//
// Mirrors what the reference does once per `discretize` call on the host
// (src/discretize.jl:413-767: build loss functions, merge with the strategy, build full_loss_function)
// and what it does per optimiser iteration (src/discretize.jl:567-598).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pinn_hip.h"
#include "aux_kernels.hpp"
#include "plat.hpp"
#include "spec_registry.hpp"

#ifdef PINN_EMU
namespace wv {
thread_local void (*emu_barrier_hook)(void*) = nullptr;
thread_local void* emu_barrier_ctx = nullptr;
}  // namespace wv
#endif

namespace pk {
std::vector<SpecInfo>& registry() {
    static std::vector<SpecInfo> r;
    return r;
}
}  // namespace pk

namespace {

constexpr int REDUCE_SPLIT = 32;     // stage-1 chunks of the fixed-order slab reduction

thread_local std::string g_err;
int fail(const std::string& m) {
    g_err = m;
    return 1;
}

struct Slot {
    int net;
    int order;
    int axes[4];
    unsigned lap = 0;        // != 0: the sum of the pure second derivatives over these axes (one "forward Laplacian" jet channel)
};
struct Net {
    int act;
    int theta_off;
    std::vector<int> sizes;          // n0 .. nL (nL == 1)
    int nparams() const {
        int n = 0;
        for (size_t i = 0; i + 1 < sizes.size(); ++i) n += sizes[i + 1] * sizes[i] + sizes[i + 1];
        return n;
    }
    int maxhidden() const {
        int m = 0;
        for (size_t i = 1; i + 1 < sizes.size(); ++i) m = std::max(m, sizes[i]);
        return m;
    }
};
struct Term {
    int d = 0;
    std::vector<Slot> slots;
    std::vector<rp::Instr> ops;      // descriptor row numbering
    int out_row = 0;
    // plan
    int net = -1;
    int group = -1;
    int slot_in_group = -1;
    int coupled = -1;                // >= 0: index into pinn_engine::coupled (equation couples several networks)
    std::vector<int> chan_of_slot;
    // per referenced network: which of the term's coordinates feed the network's inputs (descriptor `inmap` lines; default
    // identity) — dependent variables of one system may take different arguments (src/discretize.jl:111-131)
    std::map<int, std::vector<int>> inmap;
    // coordinate-only subexpressions hoisted out of the fused tape (analyse_static): evaluated by k_src per point set
    std::vector<rp::Instr> src_prog;     // compact numbering: rows [0,d) coordinates, row d+i = static op i
    std::vector<int> src_root;           // compact row of source j
    std::vector<int> src_of_op;          // per descriptor op: source index, or -1
    std::vector<int> tape_ops;           // descriptor ops that stay in the fused tape, in order
    rp::Instr* d_src_prog = nullptr;
    float* d_src = nullptr;              // [nsrc][n]
    int64_t src_cap = 0;
    // user-supplied per-point data channels (OP_DATA; pinn_set_point_data), valid for the current point set only
    int ndata = 0;
    float* d_data = nullptr;
    int64_t data_n = 0, data_cap = 0;
    // optional quadrature weights of the current point set, stored as sqrt(n_norm * w_i) (pinn_set_point_weights)
    float* d_pw = nullptr;
    int64_t pw_n = 0, pw_cap = 0;
    // data
    float* d_pts = nullptr;
    int64_t n = 0, n_norm = 0;
    float* d_resid = nullptr;
    int64_t resid_cap = 0;
    // on-device sampler: kind 0 = fixed set, 1 = uniform (StochasticTraining), 2 = Latin hypercube (QuasiRandomTraining default),
    // redrawn before every training step
    int sampler = 0;
    float* d_lb = nullptr;
    float* d_ub = nullptr;
    unsigned seed = 0, draws = 0;
};
struct Group {
    int kind = 0;                    // 0: fused (single-network terms); 1: per-network FWD/GRADIN launches of coupled terms
    int net = -1;
    const pk::SpecInfo* spec = nullptr;
    std::vector<int> terms;
    pk::GroupArgs ga;
    rp::Instr* d_prog = nullptr;
    std::vector<int> prog_off, prog_n, out_row;
    float* d_slabs = nullptr;
    double* d_losspart = nullptr;
    float* d_scratch = nullptr;
    float* d_rec = nullptr;          // kind 1, family 2: per-tile records of the forward launch, read back by the reverse launch
    size_t rec_slots = 0;            // (instead of running the forward pass twice; falls back to recomputation above REC_BUDGET)
    bool use_rec = false;
    double* d_tmp = nullptr;         // stage-1 partial sums [nsplit][nent + K]
    std::vector<int> row_theta, row_ptr, row_off;   // host CSR: theta element -> slab offsets of this group
    int nent = 0;
    int blocks = 0;
    int max_blocks = 0;
    bool active = false;
    plat_event ev_a, ev_b;
    bool timed = false;
};
// an equation that couples several networks (systems of PDEs, src/discretize.jl:58-80): forward launch per network ->
// k_expr (tape over all networks' jets) -> reverse launch per network
struct Coupled {
    int term = -1;
    std::vector<int> nets;           // networks referenced, increasing
    std::vector<int> groups;         // per network: the kind-1 group that runs it
    std::vector<int> slot_net;       // per slot: index into `nets`
    std::vector<float*> d_jets, d_ubar;   // per network: [C_n][N]
    int64_t cap = 0;
    rp::Instr* d_prog = nullptr;
    double* d_losspart = nullptr;    // pseudo-group for the reduction: [blocks*4][K]
    float* d_pslab = nullptr;        // [blocks][16]
    double* d_tmp = nullptr;
    int blocks = 0, cap_blocks = 0;
    std::vector<int> row_theta, row_ptr, row_off;
};
struct NetPlan {
    const pk::SpecInfo* spec = nullptr;   // any spec with the right (HP,NHH,D): packed layout is shared
    float* d_packed = nullptr;
    int* d_pack_idx = nullptr;
    int npacked = 0;
};

}  // namespace

struct pinn_engine {
    int64_t ntheta = 0;
    int np = 0, ne = 0, p_theta_off = 0;
    std::vector<float> p_defaults;
    std::vector<Net> nets;
    std::vector<Term> terms;
    std::vector<Group> groups;
    std::vector<Coupled> coupled;
    std::vector<NetPlan> netplans;
    int ncu = 0;
    plat_stream stream = nullptr;
    bool own_stream = true;
    float* d_theta = nullptr;
    float* d_params = nullptr;
    float* d_defaults = nullptr;
    double* d_lossraw = nullptr;
    int* d_gr_ptr = nullptr;         // global reduce CSR over theta: contributions (group, entry)
    int* d_gr_grp = nullptr;
    int* d_gr_ent = nullptr;
    float* d_out = nullptr;          // [P grad | K raw sums]
    std::vector<float> h_out;
    plat_event ev0, ev1, ev2, ev3;
    plat_stream aux_stream[2] = {nullptr, nullptr};     // under-filled launch groups run concurrently (fork/join by events)
    plat_event ev_fork, ev_join[aux::MAX_GROUPS];
    float last_kernel_ms = 0.f, last_total_ms = 0.f;
    bool timing_valid = false;
    int timing_level = 2;        // 0: no events, 1: per-launch-group kernel events, 2: + phase events (pinn_last_timing)
    int timing_group = -1;       // level >= 1: which launch group gets events (-1: all)
    // resident-theta Adam state
    float* d_opt_theta = nullptr;
    float* d_opt_m = nullptr;
    float* d_opt_v = nullptr;
    float* d_opt_out = nullptr;      // [P + K]
    float* d_w_over_n = nullptr;     // [K]
    double* d_hist = nullptr;
    int hist_cap = 0;
    long long opt_t = 0;
    // phi scratch
    float* d_phi_pts = nullptr;
    float* d_phi_out = nullptr;
    int64_t phi_cap = 0;
};

namespace {

// ---------------------------------------------------------------------------------------------
// descriptor parsing
// ---------------------------------------------------------------------------------------------
const char* OPNAMES[rp::OP_COUNT] = {"CONST", "ADD", "SUB", "MUL", "DIV", "NEG", "ADDC", "MULC", "POWI", "POW", "POWC",
                                     "SIN", "COS", "TAN", "EXP", "LOG", "SQRT", "ABS", "TANH", "SINH", "COSH", "SECH",
                                     "SINPI", "COSPI", "MAX", "MIN", "DATA"};

int parse_descriptor(const char* text, pinn_engine& E) {
    std::istringstream in(text);
    std::string tok;
    auto expect = [&](const char* w) -> bool {
        in >> tok;
        return (bool)in && tok == w;
    };
    int ver = 0;
    if (!expect("pinnir") || !(in >> ver) || ver != 1) return fail("descriptor: expected 'pinnir 1'");
    if (!expect("ntheta") || !(in >> E.ntheta)) return fail("descriptor: ntheta");
    if (!expect("params") || !(in >> E.np >> E.ne >> E.p_theta_off)) return fail("descriptor: params");
    if (E.np < 0 || E.np > pk::MAX_PARAMS || E.ne > E.np) return fail("descriptor: at most 4 PDE parameters are supported");
    if (!expect("defaults")) return fail("descriptor: defaults");
    E.p_defaults.assign(pk::MAX_PARAMS, 0.f);
    for (int i = 0; i < E.np; ++i)
        if (!(in >> E.p_defaults[i])) return fail("descriptor: defaults values");
    int nn = 0;
    if (!expect("nets") || !(in >> nn) || nn < 1) return fail("descriptor: nets");
    E.nets.resize(nn);
    for (int i = 0; i < nn; ++i) {
        int id, ns;
        std::string act;
        if (!expect("net") || !(in >> id >> act >> E.nets[i].theta_off >> ns) || id != i) return fail("descriptor: net line");
        if (act == "tanh") E.nets[i].act = pk::ACT_TANH;
        else if (act == "sigmoid") E.nets[i].act = pk::ACT_SIGMOID;
        else if (act == "sin") E.nets[i].act = pk::ACT_SIN;
        else return fail("descriptor: unsupported activation '" + act + "' (supported: tanh, sigmoid, sin)");
        E.nets[i].sizes.resize(ns);
        for (int j = 0; j < ns; ++j)
            if (!(in >> E.nets[i].sizes[j])) return fail("descriptor: net sizes");
        if (ns < 3) return fail("descriptor: a chain needs at least one hidden layer");
        if (E.nets[i].sizes.back() != 1) return fail("descriptor: only single-output chains (one per dependent variable) are supported, as in the reference (pinn_types.jl:106-108)");
    }
    int nt = 0;
    if (!expect("terms") || !(in >> nt) || nt < 1) return fail("descriptor: terms");
    E.terms.resize(nt);
    for (int i = 0; i < nt; ++i) {
        Term& T = E.terms[i];
        int id, ns, no;
        if (!expect("term") || !(in >> id >> T.d >> ns >> no >> T.out_row) || id != i) return fail("descriptor: term line");
        T.slots.resize(ns);
        for (int s = 0; s < ns; ++s) {
            Slot& S = T.slots[s];
            std::string ord;
            if (!expect("slot") || !(in >> S.net >> ord)) return fail("descriptor: slot line");
            if (S.net < 0 || S.net >= nn) return fail("descriptor: slot net id");
            if (ord == "lap") {                      // slot <net> lap <n> a0 a1 ... : sum of d2/dx_a^2 over the listed axes
                int n = 0;
                if (!(in >> n) || n < 1 || n > 8) return fail("descriptor: lap slot");
                S.order = 2; S.axes[0] = S.axes[1] = S.axes[2] = S.axes[3] = 0;
                for (int a = 0; a < n; ++a) {
                    int ax;
                    if (!(in >> ax) || ax < 0 || ax > 7) return fail("descriptor: lap slot axes");
                    S.lap |= 1u << ax;
                }
                continue;
            }
            S.order = std::atoi(ord.c_str());
            if (ord.empty() || ord.find_first_not_of("0123456789") != std::string::npos) return fail("descriptor: slot order");
            if (S.order < 0 || S.order > 4) return fail("derivative order > 4 is not supported by the HIP engine");
            for (int a = 0; a < S.order; ++a)
                if (!(in >> S.axes[a])) return fail("descriptor: slot axes");
            if (S.order == 2 && S.axes[0] > S.axes[1]) std::swap(S.axes[0], S.axes[1]);
            if (S.order >= 3)
                for (int a = 1; a < S.order; ++a)
                    if (S.axes[a] != S.axes[0]) return fail("mixed derivatives of order > 2 are not supported by the HIP engine (pure d^3/dx^3, d^4/dx^4 are)");
        }
        T.ops.resize(no);
        for (int q = 0; q < no; ++q) {
            std::string name;
            rp::Instr& I = T.ops[q];
            if (!expect("op") || !(in >> name >> I.a >> I.b >> I.imm)) return fail("descriptor: op line");
            I.code = -1;
            for (int c = 0; c < rp::OP_COUNT; ++c)
                if (name == OPNAMES[c]) I.code = c;
            if (I.code < 0) return fail("descriptor: unknown op '" + name + "'");
            const int lim = T.d + E.np + ns + q;          // operands may only reference earlier rows
            if (!rp::is_nullary(I.code) && (I.a < 0 || I.a >= lim)) return fail("descriptor: op operand row out of range");
            if (rp::is_binary(I.code) && (I.b < 0 || I.b >= lim)) return fail("descriptor: op operand row out of range");
            rp::finalize(I);
            if (I.code == rp::OP_DATA) {
                if (I.imm < 0.f || I.imm > 15.f || I.imm != (float)(int)I.imm) return fail("descriptor: DATA channel index");
                T.ndata = std::max(T.ndata, (int)I.imm + 1);
            }
        }
        if (T.out_row < 0 || T.out_row >= T.d + E.np + ns + no) return fail("descriptor: out row out of range");
        // optional: inmap <net> <n> <coordinate index of input 0> ... (one line per network whose inputs are not simply the
        // term's coordinates in order)
        for (;;) {
            const std::streampos pos = in.tellg();
            std::string tok;
            if (!(in >> tok)) { in.clear(); break; }
            if (tok != "inmap") { in.seekg(pos); break; }
            int net, n;
            if (!(in >> net >> n) || net < 0 || net >= nn || n < 1 || n > 4) return fail("descriptor: inmap line");
            std::vector<int> m(n);
            for (int i = 0; i < n; ++i)
                if (!(in >> m[i]) || m[i] < 0 || m[i] >= T.d) return fail("descriptor: inmap coordinate index out of range");
            T.inmap[net] = m;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// plan: pick a compiled kernel for every term, build pack / reduce index maps
// ---------------------------------------------------------------------------------------------
// Split a single-network term's program into the coordinate-only part (no dependence on the trial function or on PDE
// parameters) and the rest.  Coordinate-only ops that feed the rest become "sources": per-point input channels of the
// fused kernel's tape, evaluated once per point set instead of once per loss evaluation.
void analyse_static(Term& T, int np) {
    const int S = (int)T.slots.size(), nops = (int)T.ops.size();
    const int rslot0 = T.d + np, rop0 = rslot0 + S;
    std::vector<char> dyn(nops, 0), keep(nops, 0), used(nops, 0);
    auto row_dyn = [&](int row) { return row >= T.d && (row < rop0 || dyn[row - rop0]); };
    for (int q = 0; q < nops; ++q) {
        const rp::Instr& I = T.ops[q];
        dyn[q] = (!rp::is_nullary(I.code) && row_dyn(I.a)) || (rp::is_binary(I.code) && row_dyn(I.b));
    }
    T.src_prog.clear(); T.src_root.clear(); T.tape_ops.clear();
    T.src_of_op.assign(nops, -1);
    auto mark = [&](int row) {                 // operand of a tape op: a static op row must be provided to the tape
        if (row < rop0) return;
        const int q = row - rop0;
        if (dyn[q]) return;
        if (T.ops[q].code == rp::OP_CONST) keep[q] = 1;      // constants stay in the tape (no dispatch cost)
        else used[q] = 1;
    };
    for (int q = 0; q < nops; ++q)
        if (dyn[q]) {
            keep[q] = 1;
            if (!rp::is_nullary(T.ops[q].code)) mark(T.ops[q].a);
            if (rp::is_binary(T.ops[q].code)) mark(T.ops[q].b);
        }
    mark(T.out_row);
    int nsrc = 0, nstatic = 0;
    for (int q = 0; q < nops; ++q) { nsrc += used[q]; nstatic += !dyn[q]; }
    if (nsrc == 0 || nsrc > aux::SRC_MAX || T.d + nstatic > aux::EXPR_MAX_ROWS) {       // nothing to hoist / too many: keep everything
        for (int q = 0; q < nops; ++q) T.tape_ops.push_back(q);
        return;
    }
    std::vector<int> compact(nops, -1);
    for (int q = 0; q < nops; ++q) {
        if (dyn[q]) continue;
        rp::Instr I = T.ops[q];
        auto cmap = [&](int row) { return row < T.d ? row : T.d + compact[row - rop0]; };
        I.a = rp::is_nullary(I.code) ? 0 : cmap(I.a);
        I.b = rp::is_binary(I.code) ? cmap(I.b) : 0;
        rp::finalize(I);
        compact[q] = (int)T.src_prog.size();
        T.src_prog.push_back(I);
        if (used[q]) { T.src_of_op[q] = (int)T.src_root.size(); T.src_root.push_back(T.d + compact[q]); }
    }
    for (int q = 0; q < nops; ++q)
        if (keep[q]) T.tape_ops.push_back(q);
}


// "Forward Laplacian": when pure second derivatives u_aa, u_bb, ... of one network occur in a residual only as terms of one sum
// (each used once, as leaves of the same tree of ADD ops), they are replaced by ONE jet channel carrying sum_a u_aa through the
// layers (JetSet::LAP) — the 2-D Poisson interior term then needs 4 channels (u, u_x, u_y, lap u) instead of 5, the 3-D heat
// equation 6 instead of 8.  Works on the descriptor numbering (rows [coords | params | slots | ops]); returns false (term
// untouched) when nothing can be fused.
bool fuse_laplacian(Term& T, int np) {
    const int S = (int)T.slots.size(), nops = (int)T.ops.size();
    const int rslot0 = T.d + np, rop0 = rslot0 + S;
    std::vector<int> uses(rop0 + nops, 0);
    for (int q = 0; q < nops; ++q) {
        const rp::Instr& I = T.ops[q];
        if (!rp::is_nullary(I.code)) ++uses[I.a];
        if (rp::is_binary(I.code)) ++uses[I.b];
    }
    ++uses[T.out_row];
    auto is_add = [&](int row) { return row >= rop0 && T.ops[row - rop0].code == rp::OP_ADD; };
    auto inner = [&](int row) { return is_add(row) && uses[row] == 1; };         // ADD node that only feeds its parent ADD
    auto cand = [&](int row) {                                                     // pure second derivative, used exactly once
        if (row < rslot0 || row >= rop0 || uses[row] != 1) return false;
        const Slot& s = T.slots[row - rslot0];
        return s.lap == 0 && s.order == 2 && s.axes[0] == s.axes[1];
    };
    // roots: ADD ops that are not themselves inner nodes of a larger ADD tree
    std::vector<char> is_inner_child(nops, 0);
    for (int q = 0; q < nops; ++q)
        if (T.ops[q].code == rp::OP_ADD) {
            if (inner(T.ops[q].a)) is_inner_child[T.ops[q].a - rop0] = 1;
            if (inner(T.ops[q].b)) is_inner_child[T.ops[q].b - rop0] = 1;
        }
    // a leaf c * u_aa: the bare slot (c = 1), MULC(slot, c) or NEG(slot) — numeric factors are distributed over sums by the host's
    // algebra system, so nu * (u_xx + u_yy) usually arrives as nu * u_xx + nu * u_yy
    struct Leaf { int slot_row; float coef; int via_op; };
    auto leaf_of = [&](int row) -> Leaf {
        if (cand(row)) return Leaf{row, 1.0f, -1};
        if (row >= rop0 && uses[row] == 1) {
            const rp::Instr& I = T.ops[row - rop0];
            if (I.code == rp::OP_MULC && cand(I.a)) return Leaf{I.a, I.imm, row - rop0};
            if (I.code == rp::OP_NEG && cand(I.a)) return Leaf{I.a, -1.0f, row - rop0};
        }
        return Leaf{-1, 0.f, -1};
    };
    struct Tree { int root; std::vector<int> leaves, nodes; std::vector<int> fused; std::vector<int> fused_ops; unsigned mask; int net; float coef; };
    std::vector<Tree> trees;
    for (int q = 0; q < nops; ++q) {
        if (T.ops[q].code != rp::OP_ADD || is_inner_child[q]) continue;
        Tree tr; tr.root = q; tr.mask = 0; tr.net = -1;
        std::vector<int> stack{rop0 + q};
        while (!stack.empty()) {
            const int row = stack.back(); stack.pop_back();
            tr.nodes.push_back(row - rop0);
            for (int child : {T.ops[row - rop0].b, T.ops[row - rop0].a}) {
                if (inner(child)) stack.push_back(child);
                else tr.leaves.push_back(child);
            }
        }
        // candidate leaves of one network with one common coefficient and distinct axes
        std::map<std::pair<int, float>, std::vector<int>> by_key;
        for (int leaf : tr.leaves) {
            const Leaf lf = leaf_of(leaf);
            if (lf.slot_row >= 0) by_key[{T.slots[lf.slot_row - rslot0].net, lf.coef}].push_back(leaf);
        }
        for (auto& kv : by_key) {
            unsigned mask = 0; bool dup = false;
            for (int leaf : kv.second) { const unsigned b = 1u << T.slots[leaf_of(leaf).slot_row - rslot0].axes[0]; dup = dup || (mask & b); mask |= b; }
            if (kv.second.size() >= 2 && !dup && tr.fused.empty()) {
                tr.fused = kv.second; tr.mask = mask; tr.net = kv.first.first; tr.coef = kv.first.second;
                for (int leaf : kv.second) if (leaf_of(leaf).via_op >= 0) tr.fused_ops.push_back(leaf_of(leaf).via_op);
            }
        }
        if (!tr.fused.empty()) trees.push_back(tr);
    }
    if (trees.empty()) return false;
    // at most one Laplacian channel per network in the compiled kernels: all fused groups of a network must agree on the axes
    std::map<int, unsigned> net_mask;
    for (auto& tr : trees) {
        if (net_mask.count(tr.net) && net_mask[tr.net] != tr.mask) return false;
        net_mask[tr.net] = tr.mask;
    }
    // ---- rebuild: slots (drop fused ones, append one lap slot per fused tree), ops (fused trees become ADD chains over the
    // remaining leaves + the lap slot) ----
    std::vector<char> slot_dead(S, 0);
    for (auto& tr : trees) for (int leaf : tr.fused) slot_dead[leaf_of(leaf).slot_row - rslot0] = 1;
    std::vector<Slot> nslots;
    std::vector<int> slot_new(S, -1);
    for (int s = 0; s < S; ++s) if (!slot_dead[s]) { slot_new[s] = (int)nslots.size(); nslots.push_back(T.slots[s]); }
    std::vector<int> tree_slot(trees.size());
    for (size_t i = 0; i < trees.size(); ++i) {
        Slot L; L.net = trees[i].net; L.order = 2; L.axes[0] = L.axes[1] = L.axes[2] = L.axes[3] = 0; L.lap = trees[i].mask;
        tree_slot[i] = (int)nslots.size();
        nslots.push_back(L);
    }
    const int S2 = (int)nslots.size(), rop0n = rslot0 + S2;
    std::vector<int> op_new(nops, -1);                    // old op -> new ROW (may be a non-op row when a tree collapses to one leaf)
    std::vector<char> op_dropped(nops, 0);
    std::map<int, size_t> root_tree;
    for (size_t i = 0; i < trees.size(); ++i) {
        root_tree[trees[i].root] = i;
        for (int n : trees[i].nodes) if (n != trees[i].root) op_dropped[n] = 1;
        for (int n : trees[i].fused_ops) op_dropped[n] = 1;
    }
    std::vector<rp::Instr> nops_v;
    auto map_row = [&](int row) -> int {
        if (row < rslot0) return row;
        if (row < rop0) return rslot0 + slot_new[row - rslot0];
        return op_new[row - rop0];
    };
    for (int q = 0; q < nops; ++q) {
        if (op_dropped[q]) continue;
        auto it = root_tree.find(q);
        if (it == root_tree.end()) {
            rp::Instr I = T.ops[q];
            if (!rp::is_nullary(I.code)) I.a = map_row(I.a);
            if (rp::is_binary(I.code)) I.b = map_row(I.b);
            op_new[q] = rop0n + (int)nops_v.size();
            nops_v.push_back(I);
            continue;
        }
        const Tree& tr = trees[it->second];
        int lap_row = rslot0 + tree_slot[it->second];
        if (tr.coef != 1.0f) {                          // c * (sum of second derivatives)
            rp::Instr I{};
            I.code = rp::OP_MULC; I.a = lap_row; I.b = 0; I.imm = tr.coef;
            rp::finalize(I);
            lap_row = rop0n + (int)nops_v.size();
            nops_v.push_back(I);
        }
        std::vector<int> leaves{lap_row};
        for (int leaf : tr.leaves)
            if (std::find(tr.fused.begin(), tr.fused.end(), leaf) == tr.fused.end()) leaves.push_back(map_row(leaf));
        int acc = leaves[0];
        for (size_t i = 1; i < leaves.size(); ++i) {
            rp::Instr I{};
            I.code = rp::OP_ADD; I.a = acc; I.b = leaves[i]; I.imm = 0.f;
            rp::finalize(I);
            acc = rop0n + (int)nops_v.size();
            nops_v.push_back(I);
        }
        op_new[q] = acc;
    }
    const int out_new = map_row(T.out_row);
    T.slots = nslots;
    T.ops = nops_v;
    T.out_row = out_new;
    return true;
}

int round_hp(int h) {
    if (h <= 16) return 16;
    if (h <= 32) return 32;
    if (h <= 64) return 64;
    if (h <= 128) return 128;
    return ((h + 15) / 16) * 16;
}

const pk::SpecInfo* find_spec(int HP, int NHH, int D, unsigned need_first, const std::vector<std::pair<int, int>>& need_pairs,
                              unsigned need_hi, std::vector<int>* pair_index, bool need_sin = false) {
    const pk::SpecInfo* best = nullptr;
    for (const pk::SpecInfo& s : pk::registry()) {
        if (s.HP != HP || s.NHH != NHH || s.D != D) continue;
        if (need_sin && !s.has_sin) continue;
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
        if (!best || s.C < best->C || (s.C == best->C && s.family > best->family) ||
            (s.C == best->C && s.family == best->family && s.PG > best->PG)) best = &s;
    }
    (void)pair_index;
    return best;
}

int first_rank(const pk::SpecInfo& s, int axis) {
    int c = 0;
    for (int a = 0; a < axis; ++a)
        if (s.D1MASK & (1u << a)) ++c;
    return c;
}

int chan_of(const pk::SpecInfo& s, const Slot& sl) {
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

int build_plan(pinn_engine& E) {
    // ---- nets ----
    E.netplans.resize(E.nets.size());
    for (size_t n = 0; n < E.nets.size(); ++n) {
        const Net& N = E.nets[n];
        if (N.theta_off < 0 || N.theta_off + N.nparams() > E.ntheta) return fail("descriptor: net parameters exceed ntheta");
    }
    if (E.ne > 0 && (E.p_theta_off < 0 || E.p_theta_off + E.ne > E.ntheta)) return fail("descriptor: theta.p exceeds ntheta");

    // ---- terms -> groups ----
    auto needs_of = [&](const Term& T, int net, unsigned& need_first, std::vector<std::pair<int, int>>& need_pairs, unsigned& need_hi) -> int {
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
        sp = find_spec(HP, LH - 1, d, need_first, need_pairs, need_hi, nullptr, N.act == pk::ACT_SIN);
        if (!sp) {
            char b[256];
            std::snprintf(b, sizeof b,
                          "term %zu: no compiled kernel for hidden width %d (padded %d), %d hidden layers, d=%d, first-derivative axes mask 0x%x, %zu second derivatives, higher-order mask 0x%x%s; add a PINN_INSTANTIATE line in csrc/inst_*.hip",
                          t, N.maxhidden(), HP, LH, d, need_first, need_pairs.size(), need_hi, N.act == pk::ACT_SIN ? ", sin activation (PINN_INSTANTIATE*_SIN)" : "");
            return fail(b);
        }
        return 0;
    };
    // pass 0: forward-Laplacian fusion (fuse_laplacian) wherever a compiled kernel carries the resulting channel set
    static const bool no_lap = std::getenv("PINN_NO_LAPLACIAN") != nullptr;
    auto spec_exists = [&](int net, unsigned nf, const std::vector<std::pair<int, int>>& npairs, unsigned nh) {
        const Net& N = E.nets[net];
        return find_spec(round_hp(N.maxhidden()), (int)N.sizes.size() - 3, N.sizes[0], nf, npairs, nh, nullptr, N.act == pk::ACT_SIN) != nullptr;
    };
    if (!no_lap) {
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
                for (int net : nets)
                    if (needs_of(fused[t], net, cn[net].first, cn[net].second, ch[net])) { coupled_ok = false; g_err.clear(); }
            }
        }
        if (any_coupled && coupled_ok) {
            for (auto& kv : cn) coupled_ok = coupled_ok && spec_exists(kv.first, kv.second.first, kv.second.second, ch[kv.first]);
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
    std::map<int, int> coupled_dim;
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
            two_launch[t] = T.d + E.np + cmin + (int)probe.src_root.size() + (int)probe.tape_ops.size() > rp::MAX_ROWS_FUSED;
        }
        if (two_launch[t])
            for (int net : term_nets[t]) {
                auto& nd = coupled_needs[net];
                if (needs_of(T, net, nd.first, nd.second, coupled_hi[net])) return 1;
                coupled_dim[net] = T.d;
            }
    }
    std::map<int, int> coupled_group;        // net -> kind-1 group
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
            if (spec_for(t, net, T.d, need_first, need_pairs, need_hi, sp)) return 1;
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
        if ((int)T.slots.size() > aux::EXPR_MAX_SLOTS || T.d + E.np + (int)T.slots.size() + (int)T.ops.size() > aux::EXPR_MAX_ROWS)
            return fail("term " + std::to_string(t) + ": coupled residual expression too long");
        E.coupled.emplace_back();
        Coupled& Cp = E.coupled.back();
        T.coupled = (int)E.coupled.size() - 1;
        Cp.term = (int)t;
        Cp.nets = term_nets[t];
        T.chan_of_slot.assign(T.slots.size(), -1);
        Cp.slot_net.assign(T.slots.size(), -1);
        for (size_t i = 0; i < Cp.nets.size(); ++i) {
            const int net = Cp.nets[i];
            if (!coupled_group.count(net)) {
                const pk::SpecInfo* sp = nullptr;
                if (spec_for(t, net, T.d, coupled_needs[net].first, coupled_needs[net].second, coupled_hi[net], sp)) return 1;
                E.groups.emplace_back();
                Group& G = E.groups.back();
                G.kind = 1;
                G.net = net;
                G.spec = sp;
                coupled_group[net] = (int)E.groups.size() - 1;
                if (!E.netplans[net].spec) E.netplans[net].spec = sp;
            }
            Group& G = E.groups[coupled_group[net]];
            if ((int)G.terms.size() >= pk::MAX_GROUP_TERMS) return fail("too many coupled equations for one network");
            Cp.groups.push_back(coupled_group[net]);
            G.terms.push_back((int)t);
            for (size_t si = 0; si < T.slots.size(); ++si)
                if (T.slots[si].net == net) {
                    T.chan_of_slot[si] = chan_of(*G.spec, T.slots[si]);
                    Cp.slot_net[si] = (int)i;
                    if (T.chan_of_slot[si] < 0) return fail("internal: slot has no channel");
                }
        }
    }

    // ---- per-net pack index map ----
    for (size_t n = 0; n < E.nets.size(); ++n) {
        NetPlan& NP = E.netplans[n];
        if (!NP.spec) continue;   // net unused by any term
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
        NP.npacked = s.PACKED;
        NP.d_packed = (float*)plat_malloc(sizeof(float) * s.PACKED);
        NP.d_pack_idx = (int*)plat_malloc(sizeof(int) * s.PACKED);
        if (!NP.d_packed || !NP.d_pack_idx) return fail("device allocation failed (packed weights)");
        plat_h2d(NP.d_pack_idx, idx.data(), sizeof(int) * s.PACKED, E.stream);
        plat_sync(E.stream);
    }

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
        const size_t nw = (size_t)G.max_blocks * 4;
        G.d_slabs = (float*)plat_malloc(sizeof(float) * (size_t)G.max_blocks * s.SLAB);
        G.d_losspart = (double*)plat_malloc(sizeof(double) * nw * total_terms);
        G.d_scratch = (float*)plat_malloc(sizeof(float) * (s.family == 2 ? (size_t)G.max_blocks : nw) * s.SCR);
        if (!G.d_slabs || !G.d_losspart || !G.d_scratch) return fail("device allocation failed (group buffers)");
        // columns of terms this group does not own are never written by its kernel but are summed by the reduction
        plat_memset(G.d_losspart, 0, sizeof(double) * nw * total_terms, E.stream);
        // programs (rows remapped to the kernel's channel numbering)
        std::vector<rp::Instr> prog;
        for (int ti : G.terms) {
            Term& T = E.terms[ti];
            if (G.kind == 1) {           // coupled terms: the tape runs in k_expr, not in the wave kernel
                G.prog_off.push_back(0); G.prog_n.push_back(0); G.out_row.push_back(0);
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
            for (int q : T.tape_ops) {
                rp::Instr I = T.ops[q];
                I.a = rp::is_nullary(I.code) ? 0 : remap(I.a);
                I.b = rp::is_binary(I.code) ? remap(I.b) : 0;
                rp::finalize(I);
                prog.push_back(I);
            }
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
        for (int j = 0; j <= LH; ++j) {
            loff[j] = o;
            o += N.sizes[j + 1] * N.sizes[j] + N.sizes[j + 1];
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
                        const int ti = (in / 64) * 4 + (in % 4), c = (in % 64) / 4;
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
        for (int in = 0; in < N.sizes[LH] && s.family == 1; ++in)                // W_out (1 x nLH)
            add_row(loff[LH] + in, s.O_WL + ((in / 16) * 4 + (in % 16) / 4) * 4 + in % 4, false);
        if (s.family == 1) {
            add_row(loff[LH] + N.sizes[LH], s.O_BL, false);
            for (int j = 0; j < E.ne; ++j) add_row(E.p_theta_off + j, s.O_P + j, false);
        }
        G.nent = s.SLAB;                 // stage 1 is dense over slab offsets
        G.row_theta = row_theta;
        G.row_ptr = row_ptr;
        G.row_off = ent;                 // slab offsets of the contributions
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
    }
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
        for (auto& c : contrib) {
            for (auto& pr : c) { grp.push_back(pr.first); ent.push_back(pr.second); }
            ptr.push_back((int)grp.size());
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
        {
            const std::vector<int>& m = T.inmap.at(G.net);
            td.dt = T.d;
            td.hetero = ((int)m.size() != T.d);
            for (int i = 0; i < 4; ++i) {
                td.imap[i] = i < (int)m.size() ? m[i] : 0;
                if (i < (int)m.size() && m[i] != i) td.hetero = 1;
            }
        }
        if (G.kind == 1) {               // coupled term: this network's jet / seed buffers
            const Coupled& Cp = E.coupled[T.coupled];
            for (size_t i = 0; i < Cp.groups.size(); ++i)
                if (Cp.groups[i] == gi && i < Cp.d_jets.size()) { td.out = Cp.d_jets[i]; td.in = Cp.d_ubar[i]; }
        }
        tile += td.ntiles;
    }
    G.ga.ntiles = tile;
    G.blocks = std::max(1, std::min(G.max_blocks, s.family == 2 ? tile : (tile + 3) / 4));
    // coupled groups: keep the forward launch's records in HBM when they fit the budget (default 96 GB per handle, PINN_REC_GB)
    G.use_rec = false;
    if (G.kind == 1 && s.family == 2 && s.REC > 0) {
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

// (re-)evaluate a term's coordinate-only source channels for its current point set
void eval_sources(pinn_engine& E, Term& T) {
    if (T.src_root.empty() || T.coupled >= 0) return;
    aux::SrcArgs a;
    std::memset(&a, 0, sizeof a);
    a.pts = T.d_pts; a.N = (int)T.n; a.d = T.d; a.prog = T.d_src_prog; a.nops = (int)T.src_prog.size();
    a.nsrc = (int)T.src_root.size();
    for (int j = 0; j < a.nsrc; ++j) a.root[j] = T.src_root[j];
    a.out = T.d_src;
    a.data = T.d_data;
    if (T.ndata > 0 && T.data_n != T.n) return;          // data not installed yet for this point set (ensure_points reports it)
    aux::launch_src(a, E.stream);
}

int ensure_points(pinn_engine& E) {
    for (size_t t = 0; t < E.terms.size(); ++t) {
        if (!E.terms[t].d_pts || E.terms[t].n <= 0)
            return fail("term " + std::to_string(t) + " has no collocation points (call pinn_set_points first)");
        if (E.terms[t].ndata > 0 && E.terms[t].data_n != E.terms[t].n)
            return fail("term " + std::to_string(t) + " uses per-point data channels but none are installed for its current point set (call pinn_set_point_data after pinn_set_points)");
    }
    return 0;
}

void pack_all(pinn_engine& E, const float* d_theta = nullptr) {
    const float* th = d_theta ? d_theta : E.d_theta;
    for (size_t n = 0; n < E.nets.size(); ++n) {
        NetPlan& NP = E.netplans[n];
        if (!NP.spec) continue;
        aux::launch_pack(NP.d_packed, NP.d_pack_idx, th, NP.npacked, E.stream);
    }
    aux::launch_params(E.d_params, th, E.d_defaults, E.np, E.ne, E.p_theta_off, E.stream);
}

// the device section shared by all loss/grad entry points; theta must already be in E.d_theta
// launch arguments of k_expr for one coupled equation
aux::ExprArgs expr_args(pinn_engine& E, Coupled& Cp, float scale, float* resid) {
    Term& T = E.terms[Cp.term];
    aux::ExprArgs a;
    std::memset(&a, 0, sizeof a);
    a.pts = T.d_pts; a.N = (int)T.n; a.d = T.d; a.nparams = E.np; a.nparams_estim = E.ne; a.params = E.d_params;
    a.nslots = (int)T.slots.size();
    std::vector<std::vector<char>> used(Cp.nets.size());
    for (size_t i = 0; i < Cp.nets.size(); ++i) used[i].assign(E.groups[Cp.groups[i]].spec->C, 0);
    for (int si = 0; si < a.nslots; ++si) {
        const int i = Cp.slot_net[si], ch = T.chan_of_slot[si];
        a.jets[si] = Cp.d_jets[i] + (size_t)ch * T.n;
        a.ubar[si] = Cp.d_ubar[i] + (size_t)ch * T.n;
        used[i][ch] = 1;
    }
    for (size_t i = 0; i < Cp.nets.size(); ++i)
        for (size_t ch = 0; ch < used[i].size(); ++ch)
            if (!used[i][ch] && a.nzero < aux::EXPR_MAX_SLOTS) a.zero[a.nzero++] = Cp.d_ubar[i] + ch * T.n;
    a.prog = Cp.d_prog; a.nops = (int)T.ops.size(); a.out_row = T.out_row; a.scale = scale;
    a.losspart = Cp.d_losspart; a.pslab = Cp.d_pslab; a.K = (int)E.terms.size(); a.term_id = Cp.term; a.resid = resid;
    a.data = T.d_data;
    a.pw = (T.pw_n == T.n && T.pw_n > 0) ? T.d_pw : nullptr;
    return a;
}

// the device section shared by all loss/grad entry points.  d_theta: theta in device memory; d_out: [P + K] floats in
// device memory.  Launches per evaluation: pack (1 per net) -> fused residual kernel (1 per group) -> reduce1 -> reduce2.
int run_loss_grad(pinn_engine& E, const float* d_theta, float* d_out, const float* term_w, int only_term /* -1 = all */, bool timing) {
    if (ensure_points(E)) return 1;
    const int K = (int)E.terms.size();
    const bool phase_ev = timing && E.timing_level >= 2;
    auto group_ev = [&](size_t g) { return timing && E.timing_level >= 1 && (E.timing_group < 0 || E.timing_group == (int)g); };
    if (phase_ev) plat_event_record(E.ev0, E.stream);
    pack_all(E, d_theta);
    if (phase_ev) plat_event_record(E.ev1, E.stream);
    aux::Reduce1Args a1;
    aux::Reduce2Args a2;
    std::memset(&a1, 0, sizeof a1);
    std::memset(&a2, 0, sizeof a2);
    int max_n1 = 1, max_split = 1;
    auto scale_of = [&](int ti) -> float {
        const float w = term_w ? term_w[ti] : 1.0f;
        const bool on = (only_term < 0 || only_term == ti);
        return on ? (float)(2.0 * (double)w / (double)E.terms[ti].n_norm) : 0.f;
    };
    // Fused groups that cannot fill the chip on their own (at most ~1.5 rounds of workgroups, e.g. one rank's share in an
    // 8-GPU strong-scaling run) are launched concurrently on auxiliary streams; big groups run back to back.
    int nfused_active = 0;
    bool all_small = true;
    for (auto& G : E.groups) {
        bool on = false;
        for (int ti : G.terms) on = on || (only_term < 0 || only_term == ti);
        if (G.kind == 0 && on) {
            ++nfused_active;
            const int per_round = G.max_blocks * (G.spec->family == 2 ? 1 : 4);
            if (2 * G.ga.ntiles > 3 * per_round) all_small = false;
        }
    }
    // opt-in: on this stack a cross-stream event wait costs 15-20 us, more than the overlap buys (profiles/r01 timeline)
    static const bool want_concurrent = std::getenv("PINN_CONCURRENT_GROUPS") != nullptr;
    const bool concurrent = want_concurrent && nfused_active >= 2 && all_small;
    int nforked = 0;
    if (concurrent) plat_event_record(E.ev_fork, E.stream);
    for (size_t g = 0; g < E.groups.size(); ++g) {
        Group& G = E.groups[g];
        bool any = false;
        for (size_t j = 0; j < G.terms.size(); ++j) {
            G.ga.terms[j].scale = scale_of(G.terms[j]);
            any = any || (only_term < 0 || only_term == G.terms[j]);
        }
        G.active = any;
        const int nsplit = std::min(REDUCE_SPLIT, G.blocks);
        a1.tmp[g] = G.d_tmp; a1.slabs[g] = G.d_slabs; a1.losspart[g] = G.d_losspart;
        a1.slab[g] = G.spec->SLAB; a1.nblocks[g] = G.blocks; a1.nsplit[g] = nsplit; a1.nent[g] = G.nent; a1.active[g] = any;
        a2.tmp[g] = G.d_tmp; a2.stride[g] = G.nent + K; a2.nsplit[g] = nsplit; a2.nent[g] = G.nent; a2.active[g] = any;
        if (!any) continue;
        max_n1 = std::max(max_n1, G.nent / 4 + K);
        max_split = std::max(max_split, nsplit);
        if (G.kind == 1) {               // coupled: forward launch now, reverse launch after k_expr
            G.spec->launch(G.ga, G.use_rec ? pk::MODE_FWDREC : pk::MODE_FWD, G.blocks, E.stream);
            continue;
        }
        plat_stream st = E.stream;
        const bool forked = concurrent && nforked++ > 0;       // the first fused group stays on the caller's stream
        if (forked) {
            st = E.aux_stream[nforked % 2];
            plat_stream_wait_event(st, E.ev_fork);
        }
        if (group_ev(g)) plat_event_record(G.ev_a, st);
        G.spec->launch(G.ga, pk::MODE_FUSED, G.blocks, st);
        if (group_ev(g)) plat_event_record(G.ev_b, st);
        G.timed = group_ev(g);
        if (forked) {
            plat_event_record(E.ev_join[g], st);
            plat_stream_wait_event(E.stream, E.ev_join[g]);
        }
    }
    for (size_t c = 0; c < E.coupled.size(); ++c) {
        Coupled& Cp = E.coupled[c];
        const int g = (int)(E.groups.size() + c);
        const bool on = (only_term < 0 || only_term == Cp.term);
        const int nsplit = std::min(REDUCE_SPLIT, Cp.blocks);
        a1.tmp[g] = Cp.d_tmp; a1.slabs[g] = Cp.d_pslab; a1.losspart[g] = Cp.d_losspart;
        a1.slab[g] = 16; a1.nblocks[g] = Cp.blocks; a1.nsplit[g] = nsplit; a1.nent[g] = 16; a1.active[g] = on;
        a2.tmp[g] = Cp.d_tmp; a2.stride[g] = 16 + K; a2.nsplit[g] = nsplit; a2.nent[g] = 16; a2.active[g] = on;
        bool groups_active = false;
        for (int gi : Cp.groups) groups_active = groups_active || E.groups[gi].active;
        if (!groups_active) continue;
        max_n1 = std::max(max_n1, 4 + K);
        max_split = std::max(max_split, nsplit);
        aux::launch_expr(expr_args(E, Cp, scale_of(Cp.term), nullptr), Cp.blocks, E.stream);    // scale 0 => zero seeds
    }
    for (size_t g = 0; g < E.groups.size(); ++g) {
        Group& G = E.groups[g];
        if (G.kind != 1 || !G.active) continue;
        if (group_ev(g)) plat_event_record(G.ev_a, E.stream);
        G.spec->launch(G.ga, G.use_rec ? pk::MODE_GRADREC : pk::MODE_GRADIN, G.blocks, E.stream);
        if (group_ev(g)) plat_event_record(G.ev_b, E.stream);
        G.timed = group_ev(g);
    }
    if (phase_ev) plat_event_record(E.ev2, E.stream);
    a1.K = K;
    a2.out = d_out; a2.lossraw = E.d_lossraw; a2.row_ptr = E.d_gr_ptr; a2.row_grp = E.d_gr_grp; a2.row_ent = E.d_gr_ent;
    a2.ngroups = (int)(E.groups.size() + E.coupled.size()); a2.P = (int)E.ntheta; a2.K = K;
    aux::launch_reduce(a1, a2, max_n1, max_split, E.stream);
    if (phase_ev) plat_event_record(E.ev3, E.stream);
    return 0;
}

int upload_theta(pinn_engine& E, const float* theta, int64_t p) {
    if (p != E.ntheta) return fail("theta length " + std::to_string(p) + " != ntheta " + std::to_string(E.ntheta));
    if (plat_h2d(E.d_theta, theta, sizeof(float) * p, E.stream)) return fail(std::string("H2D copy of theta failed: ") + plat_last_error());
    return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* pinn_backend(void) { return plat_name(); }
int pinn_abi_version(void) { return 1; }
const char* pinn_last_error(void) { return g_err.c_str(); }

int pinn_create(const char* descriptor, pinn_handle* out) {
    if (!descriptor || !out) return fail("pinn_create: null argument");
    *out = nullptr;
    std::string err;
    if (plat_init(err)) return fail(err);
    std::unique_ptr<pinn_engine> E(new pinn_engine());
    if (parse_descriptor(descriptor, *E)) return 1;
    E->ncu = plat_num_cus();
    E->stream = plat_stream_create();
    const int K = (int)E->terms.size();
    E->d_theta = (float*)plat_malloc(sizeof(float) * E->ntheta);
    E->d_params = (float*)plat_malloc(sizeof(float) * pk::MAX_PARAMS);
    E->d_defaults = (float*)plat_malloc(sizeof(float) * pk::MAX_PARAMS);
    E->d_lossraw = (double*)plat_malloc(sizeof(double) * K);
    E->d_out = (float*)plat_malloc(sizeof(float) * (E->ntheta + K));
    plat_event_create(E->ev0); plat_event_create(E->ev1); plat_event_create(E->ev2); plat_event_create(E->ev3);
    plat_event_create(E->ev_fork);
    for (auto& e : E->ev_join) plat_event_create(e);
    for (auto& st : E->aux_stream) st = plat_stream_create();
    if (!E->d_theta || !E->d_params || !E->d_defaults || !E->d_lossraw || !E->d_out) {
        pinn_destroy(E.release());                   // releases whatever was allocated
        return fail("device allocation failed");
    }
    plat_h2d(E->d_defaults, E->p_defaults.data(), sizeof(float) * pk::MAX_PARAMS, E->stream);
    E->h_out.resize(E->ntheta + K);
    if (build_plan(*E)) {
        const std::string msg = g_err;               // pinn_destroy must not lose the reason
        pinn_destroy(E.release());
        g_err = msg;
        return 1;
    }
    *out = E.release();
    return 0;
}

int pinn_destroy(pinn_handle h) {
    if (!h) return 0;
    pinn_engine& E = *h;
    plat_sync(E.stream);
    for (auto& T : E.terms) { plat_free(T.d_pts); plat_free(T.d_resid); plat_free(T.d_lb); plat_free(T.d_ub); plat_free(T.d_src_prog); plat_free(T.d_src); plat_free(T.d_data); plat_free(T.d_pw); }
    plat_free(E.d_opt_theta); plat_free(E.d_opt_m); plat_free(E.d_opt_v); plat_free(E.d_opt_out); plat_free(E.d_w_over_n); plat_free(E.d_hist);
    for (auto& G : E.groups) {
        plat_free(G.d_prog); plat_free(G.d_slabs); plat_free(G.d_losspart); plat_free(G.d_scratch); plat_free(G.d_rec);
        plat_free(G.d_tmp);
        plat_event_destroy(G.ev_a); plat_event_destroy(G.ev_b);
    }
    for (auto& Cp : E.coupled) {
        for (float* q : Cp.d_jets) plat_free(q);
        for (float* q : Cp.d_ubar) plat_free(q);
        plat_free(Cp.d_prog); plat_free(Cp.d_losspart); plat_free(Cp.d_pslab); plat_free(Cp.d_tmp);
    }
    for (auto& N : E.netplans) { plat_free(N.d_packed); plat_free(N.d_pack_idx); }
    plat_free(E.d_theta); plat_free(E.d_params); plat_free(E.d_defaults); plat_free(E.d_lossraw); plat_free(E.d_gr_ptr); plat_free(E.d_gr_grp); plat_free(E.d_gr_ent);
    plat_free(E.d_out); plat_free(E.d_phi_pts); plat_free(E.d_phi_out);
    plat_event_destroy(E.ev0); plat_event_destroy(E.ev1); plat_event_destroy(E.ev2); plat_event_destroy(E.ev3);
    plat_event_destroy(E.ev_fork);
    for (auto& e : E.ev_join) plat_event_destroy(e);
    for (auto& st : E.aux_stream) plat_stream_destroy(st);
    if (E.own_stream) plat_stream_destroy(E.stream);
    delete h;
    return 0;
}

int pinn_num_terms(pinn_handle h) { return h ? (int)h->terms.size() : -1; }
int64_t pinn_num_theta(pinn_handle h) { return h ? h->ntheta : -1; }

static int set_points_impl(pinn_handle h, int term, const float* pts, int64_t n, int64_t n_norm, bool device) {
    if (!h) return fail("null handle");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_set_points: term index out of range");
    if (!pts || n <= 0) return fail("pinn_set_points: empty point set (the reference's mean(abs2, .) over an empty set is NaN; refusing)");
    Term& T = E.terms[term];
    if (n * T.d >= (int64_t)1 << 31) return fail("pinn_set_points: point set too large for 32-bit indexing; shard it");
    plat_sync(E.stream);
    if (T.n != n || !T.d_pts) {
        plat_free(T.d_pts);
        T.d_pts = (float*)plat_malloc(sizeof(float) * n * T.d);
        if (!T.d_pts) return fail("device allocation failed (points)");
    }
    int rc = device ? plat_d2d(T.d_pts, pts, sizeof(float) * n * T.d, E.stream) : plat_h2d(T.d_pts, pts, sizeof(float) * n * T.d, E.stream);
    if (rc) return fail(std::string("copy of points failed: ") + plat_last_error());
    plat_sync(E.stream);
    T.n = n;
    T.n_norm = n_norm > 0 ? n_norm : n;
    T.data_n = 0;                                        // per-point data and weights belong to the previous set
    T.pw_n = 0;
    if (T.coupled < 0) {
        if (!T.src_root.empty()) {
            if (T.src_cap < n) {
                plat_free(T.d_src);
                T.d_src = (float*)plat_malloc(sizeof(float) * T.src_root.size() * (size_t)n);
                if (!T.d_src) return fail("device allocation failed (source channels)");
                T.src_cap = n;
            }
            eval_sources(E, T);
            plat_sync(E.stream);
        }
        retile(E, T.group);
        return 0;
    }
    Coupled& Cp = E.coupled[T.coupled];
    const int K = (int)E.terms.size();
    if (Cp.cap < n) {
        for (float* q : Cp.d_jets) plat_free(q);
        for (float* q : Cp.d_ubar) plat_free(q);
        Cp.d_jets.assign(Cp.nets.size(), nullptr);
        Cp.d_ubar.assign(Cp.nets.size(), nullptr);
        for (size_t i = 0; i < Cp.nets.size(); ++i) {
            const int C = E.groups[Cp.groups[i]].spec->C;
            Cp.d_jets[i] = (float*)plat_malloc(sizeof(float) * (size_t)C * n);
            Cp.d_ubar[i] = (float*)plat_malloc(sizeof(float) * (size_t)C * n);
            if (!Cp.d_jets[i] || !Cp.d_ubar[i]) return fail("device allocation failed (coupled jets)");
        }
        Cp.cap = n;
    }
    Cp.blocks = (int)((n + 255) / 256);
    if (Cp.cap_blocks < Cp.blocks) {
        plat_free(Cp.d_losspart); plat_free(Cp.d_pslab);
        Cp.d_losspart = (double*)plat_malloc(sizeof(double) * (size_t)Cp.blocks * 4 * K);
        Cp.d_pslab = (float*)plat_malloc(sizeof(float) * (size_t)Cp.blocks * 16);
        if (!Cp.d_losspart || !Cp.d_pslab) return fail("device allocation failed (coupled partials)");
        plat_memset(Cp.d_losspart, 0, sizeof(double) * (size_t)Cp.blocks * 4 * K, E.stream);
        plat_memset(Cp.d_pslab, 0, sizeof(float) * (size_t)Cp.blocks * 16, E.stream);
        plat_sync(E.stream);
        Cp.cap_blocks = Cp.blocks;
    }
    for (int gi : Cp.groups) retile(E, gi);
    return 0;
}
int pinn_set_points(pinn_handle h, int term, const float* pts, int64_t n, int64_t n_norm) { return set_points_impl(h, term, pts, n, n_norm, false); }
int pinn_set_points_device(pinn_handle h, int term, const float* d_pts, int64_t n, int64_t n_norm) { return set_points_impl(h, term, d_pts, n, n_norm, true); }

int pinn_loss_grad(pinn_handle h, const float* theta, int64_t p, const float* term_w, double* term_losses, float* grad) {
    if (!h || !theta) return fail("pinn_loss_grad: null argument");
    pinn_engine& E = *h;
    const int K = (int)E.terms.size();
    if (upload_theta(E, theta, p)) return 1;
    if (run_loss_grad(E, E.d_theta, E.d_out, term_w, -1, true)) return 1;
    if (plat_d2h(E.h_out.data(), E.d_out, sizeof(float) * (E.ntheta + K), E.stream)) return fail("D2H copy failed");
    // exact double sums for the host path
    std::vector<double> raw(K);
    if (plat_d2h(raw.data(), E.d_lossraw, sizeof(double) * K, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    E.timing_valid = E.timing_level >= 2;
    if (term_losses)
        for (int k = 0; k < K; ++k) term_losses[k] = raw[k] / (double)E.terms[k].n_norm;
    if (grad) std::memcpy(grad, E.h_out.data(), sizeof(float) * E.ntheta);
    return 0;
}

int pinn_loss_grad_f64(pinn_handle h, const double* theta, int64_t p, const double* term_w, double* term_losses, double* grad) {
    if (!h || !theta) return fail("pinn_loss_grad_f64: null argument");
    const int K = (int)h->terms.size();
    std::vector<float> th(p), w(K, 1.0f), g(grad ? p : 0);
    for (int64_t i = 0; i < p; ++i) th[i] = (float)theta[i];
    if (term_w)
        for (int k = 0; k < K; ++k) w[k] = (float)term_w[k];
    int rc = pinn_loss_grad(h, th.data(), p, w.data(), term_losses, grad ? g.data() : nullptr);
    if (rc) return rc;
    if (grad)
        for (int64_t i = 0; i < p; ++i) grad[i] = (double)g[i];
    return 0;
}

int pinn_term_grads(pinn_handle h, const float* theta, int64_t p, double* term_losses, float* term_grads) {
    if (!h || !theta || !term_grads) return fail("pinn_term_grads: null argument");
    pinn_engine& E = *h;
    const int K = (int)E.terms.size();
    if (upload_theta(E, theta, p)) return 1;
    for (int k = 0; k < K; ++k) {
        if (run_loss_grad(E, E.d_theta, E.d_out, nullptr, k, false)) return 1;
        if (plat_d2h(term_grads + (size_t)k * p, E.d_out, sizeof(float) * p, E.stream)) return fail("D2H copy failed");
        double raw = 0;
        if (plat_d2h(&raw, E.d_lossraw + k, sizeof(double), E.stream)) return fail("D2H copy failed");
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        if (term_losses) term_losses[k] = raw / (double)E.terms[k].n_norm;
    }
    return 0;
}

int pinn_loss_grad_device(pinn_handle h, const float* d_theta, const float* term_w, float* d_out, void* stream) {
    if (!h || !d_theta || !d_out) return fail("pinn_loss_grad_device: null argument");
    pinn_engine& E = *h;
    const int K = (int)E.terms.size();
    plat_stream user = (plat_stream)stream;
    plat_stream saved = E.stream;
    E.stream = user;     // NULL is the (legacy) default stream — e.g. torch's current stream
    int rc = run_loss_grad(E, d_theta, d_out, term_w, -1, true);
    E.stream = saved;
    if (rc && g_err.empty()) return fail("pinn_loss_grad_device failed");
    E.timing_valid = (rc == 0) && E.timing_level >= 2;
    return rc;
}

int pinn_residual(pinn_handle h, int term, const float* theta, int64_t p, float* r) {
    if (!h || !theta || !r) return fail("pinn_residual: null argument");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_residual: term index out of range");
    Term& T = E.terms[term];
    if (!T.d_pts) return fail("pinn_residual: term has no points");
    if (upload_theta(E, theta, p)) return 1;
    pack_all(E);
    if (T.resid_cap < T.n) {
        plat_free(T.d_resid);
        T.d_resid = (float*)plat_malloc(sizeof(float) * T.n);
        T.resid_cap = T.n;
        if (!T.d_resid) return fail("device allocation failed (residual)");
    }
    if (T.coupled >= 0) {
        Coupled& Cp = E.coupled[T.coupled];
        for (int gi : Cp.groups) {
            Group& G = E.groups[gi];
            pk::GroupArgs ga = G.ga;
            int j = 0;
            for (size_t q = 0; q < G.terms.size(); ++q) if (G.terms[q] == term) j = (int)q;
            ga.nterms = 1;
            ga.terms[0] = G.ga.terms[j];
            ga.terms[0].tile0 = 0;
            ga.ntiles = ga.terms[0].ntiles;
            G.spec->launch(ga, pk::MODE_FWD, std::max(1, std::min(G.max_blocks, (ga.ntiles + 3) / 4)), E.stream);
        }
        aux::launch_expr(expr_args(E, Cp, 0.f, T.d_resid), Cp.blocks, E.stream);
        if (plat_d2h(r, T.d_resid, sizeof(float) * T.n, E.stream)) return fail("D2H copy failed");
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        return 0;
    }
    Group& G = E.groups[T.group];
    pk::GroupArgs ga = G.ga;
    ga.nterms = 1;
    ga.terms[0] = G.ga.terms[T.slot_in_group];
    ga.terms[0].tile0 = 0;
    ga.terms[0].out = T.d_resid;
    ga.ntiles = ga.terms[0].ntiles;
    const int blocks = std::max(1, std::min(G.max_blocks, (ga.ntiles + 3) / 4));
    G.spec->launch(ga, pk::MODE_RESID, blocks, E.stream);
    if (plat_d2h(r, T.d_resid, sizeof(float) * T.n, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_phi(pinn_handle h, int net, const float* theta, int64_t p, const float* pts, int64_t n, float* out) {
    if (!h || !theta || !pts || !out) return fail("pinn_phi: null argument");
    pinn_engine& E = *h;
    if (net < 0 || net >= (int)E.nets.size()) return fail("pinn_phi: net index out of range");
    if (n <= 0) return fail("pinn_phi: n must be positive");
    const Net& N = E.nets[net];
    const int LH = (int)N.sizes.size() - 2;
    const pk::SpecInfo* sp = find_spec(round_hp(N.maxhidden()), LH - 1, N.sizes[0], 0, {}, 0u, nullptr, N.act == pk::ACT_SIN);
    if (!sp) return fail("pinn_phi: no compiled value-only kernel for this network shape");
    if (!E.netplans[net].spec) return fail("pinn_phi: network is not used by any term");
    if (upload_theta(E, theta, p)) return 1;
    pack_all(E);
    if (E.phi_cap < n) {
        plat_free(E.d_phi_pts); plat_free(E.d_phi_out);
        E.d_phi_pts = (float*)plat_malloc(sizeof(float) * n * N.sizes[0]);
        E.d_phi_out = (float*)plat_malloc(sizeof(float) * n * sp->C);
        E.phi_cap = n;
        if (!E.d_phi_pts || !E.d_phi_out) return fail("device allocation failed (phi)");
    }
    plat_h2d(E.d_phi_pts, pts, sizeof(float) * n * N.sizes[0], E.stream);
    pk::GroupArgs ga;
    std::memset(&ga, 0, sizeof ga);
    ga.packed = E.netplans[net].d_packed;
    ga.params = E.d_params;
    ga.nterms = 1;
    ga.nterms_total = 1;
    ga.act = N.act;
    ga.terms[0].pts = E.d_phi_pts;
    ga.terms[0].dt = N.sizes[0];
    for (int i = 0; i < 4; ++i) ga.terms[0].imap[i] = i;
    ga.terms[0].N = (int)n;
    ga.terms[0].tile0 = 0;
    ga.terms[0].ntiles = (int)((n + sp->TP - 1) / sp->TP);
    ga.terms[0].out = E.d_phi_out;
    ga.ntiles = ga.terms[0].ntiles;
    const int blocks = std::max(1, std::min(E.ncu, (ga.ntiles + 3) / 4));
    sp->launch(ga, pk::MODE_FWD, blocks, E.stream);
    if (plat_d2h(out, E.d_phi_out, sizeof(float) * n, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_set_sampler(pinn_handle h, int term, int kind, const float* lb, const float* ub, int64_t n, uint64_t seed) {
    if (!h) return fail("null handle");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_set_sampler: term index out of range");
    Term& T = E.terms[term];
    if (kind < 0 || kind > 2) return fail("pinn_set_sampler: kind must be 0 (fixed set), 1 (uniform) or 2 (Latin hypercube)");
    T.sampler = kind;
    if (kind == 0) return 0;
    if (!lb || !ub || n <= 0) return fail("pinn_set_sampler: bounds and a positive point count are required");
    if (!T.d_lb) { T.d_lb = (float*)plat_malloc(sizeof(float) * 8); T.d_ub = (float*)plat_malloc(sizeof(float) * 8); }
    if (!T.d_lb || !T.d_ub) return fail("device allocation failed (sampler)");
    plat_h2d(T.d_lb, lb, sizeof(float) * T.d, E.stream);
    plat_h2d(T.d_ub, ub, sizeof(float) * T.d, E.stream);
    T.seed = (unsigned)(seed ^ (seed >> 32)) + 0x9E3779B9U * (unsigned)(term + 1);
    T.draws = 0;
    // allocate / size the term's point buffer through the normal path with a first draw
    std::vector<float> tmp((size_t)n * T.d, 0.f);
    if (set_points_impl(h, term, tmp.data(), n, 0, false)) return 1;
    aux::launch_sample(kind, T.d_pts, (int)(n * T.d), T.d, T.d_lb, T.d_ub, T.seed, T.draws++, E.stream);
    eval_sources(E, T);
    plat_sync(E.stream);
    return 0;
}

int pinn_set_point_data(pinn_handle h, int term, const float* data, int ndata, int64_t n) {
    if (!h || !data) return fail("pinn_set_point_data: null argument");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_set_point_data: term index out of range");
    Term& T = E.terms[term];
    if (T.ndata == 0) return fail("pinn_set_point_data: the term's residual has no DATA channels");
    if (ndata != T.ndata) return fail("pinn_set_point_data: the term's residual uses " + std::to_string(T.ndata) + " data channel(s)");
    if (!T.d_pts || n != T.n) return fail("pinn_set_point_data: the term holds " + std::to_string(T.n) + " points (install the point set first)");
    if (T.sampler != 0) return fail("pinn_set_point_data: per-point data cannot follow a resampled point set");
    plat_sync(E.stream);
    if (T.data_cap < n) {
        plat_free(T.d_data);
        T.d_data = (float*)plat_malloc(sizeof(float) * (size_t)T.ndata * n);
        if (!T.d_data) return fail("device allocation failed (point data)");
        T.data_cap = n;
    }
    if (plat_h2d(T.d_data, data, sizeof(float) * (size_t)T.ndata * n, E.stream)) return fail("H2D copy of point data failed");
    T.data_n = n;
    eval_sources(E, T);
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_set_point_weights(pinn_handle h, int term, const float* w, int64_t n) {
    if (!h) return fail("pinn_set_point_weights: null handle");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_set_point_weights: term index out of range");
    Term& T = E.terms[term];
    plat_sync(E.stream);
    if (!w) {                                            // back to the plain mean
        T.pw_n = 0;
    } else {
        if (!T.d_pts || n != T.n) return fail("pinn_set_point_weights: the term holds " + std::to_string(T.n) + " points (install the point set first)");
        if (T.sampler != 0) return fail("pinn_set_point_weights: weights cannot follow a resampled point set");
        std::vector<float> s((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            if (!(w[i] >= 0.f)) return fail("pinn_set_point_weights: weights must be non-negative");
            s[i] = (float)std::sqrt((double)w[i] * (double)T.n_norm);
        }
        if (T.pw_cap < n) {
            plat_free(T.d_pw);
            T.d_pw = (float*)plat_malloc(sizeof(float) * (size_t)n);
            if (!T.d_pw) return fail("device allocation failed (point weights)");
            T.pw_cap = n;
        }
        if (plat_h2d(T.d_pw, s.data(), sizeof(float) * (size_t)n, E.stream)) return fail("H2D copy of point weights failed");
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        T.pw_n = n;
    }
    if (T.coupled < 0) retile(E, T.group);
    else for (int gi : E.coupled[T.coupled].groups) retile(E, gi);
    return 0;
}

int pinn_get_points(pinn_handle h, int term, float* pts, int64_t n) {
    if (!h || !pts) return fail("pinn_get_points: null argument");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_get_points: term index out of range");
    Term& T = E.terms[term];
    if (!T.d_pts || n != T.n) return fail("pinn_get_points: the term holds " + std::to_string(T.n) + " points");
    if (plat_d2h(pts, T.d_pts, sizeof(float) * (size_t)n * T.d, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_adam_init(pinn_handle h, const float* theta, int64_t p) {
    if (!h || !theta) return fail("pinn_adam_init: null argument");
    pinn_engine& E = *h;
    if (p != E.ntheta) return fail("pinn_adam_init: theta length mismatch");
    const int K = (int)E.terms.size();
    if (!E.d_opt_theta) {
        E.d_opt_theta = (float*)plat_malloc(sizeof(float) * p);
        E.d_opt_m = (float*)plat_malloc(sizeof(float) * p);
        E.d_opt_v = (float*)plat_malloc(sizeof(float) * p);
        E.d_opt_out = (float*)plat_malloc(sizeof(float) * (p + K));
        E.d_w_over_n = (float*)plat_malloc(sizeof(float) * K);
        if (!E.d_opt_theta || !E.d_opt_m || !E.d_opt_v || !E.d_opt_out || !E.d_w_over_n) return fail("device allocation failed (optimiser state)");
    }
    plat_h2d(E.d_opt_theta, theta, sizeof(float) * p, E.stream);
    plat_memset(E.d_opt_m, 0, sizeof(float) * p, E.stream);
    plat_memset(E.d_opt_v, 0, sizeof(float) * p, E.stream);
    plat_sync(E.stream);
    E.opt_t = 0;
    return 0;
}

int pinn_adam_steps(pinn_handle h, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w, double* loss_history) {
    if (!h) return fail("null handle");
    pinn_engine& E = *h;
    if (!E.d_opt_theta) return fail("pinn_adam_steps: call pinn_adam_init first");
    if (nsteps <= 0) return fail("pinn_adam_steps: nsteps must be positive");
    if (ensure_points(E)) return 1;
    const int K = (int)E.terms.size(), P = (int)E.ntheta;
    if (E.hist_cap < nsteps) {
        plat_free(E.d_hist);
        E.d_hist = (double*)plat_malloc(sizeof(double) * nsteps);
        E.hist_cap = nsteps;
        if (!E.d_hist) return fail("device allocation failed (loss history)");
    }
    std::vector<float> wn(K);
    for (int k = 0; k < K; ++k) wn[k] = (term_w ? term_w[k] : 1.0f) / (float)E.terms[k].n_norm;
    plat_h2d(E.d_w_over_n, wn.data(), sizeof(float) * K, E.stream);
    for (int s = 0; s < nsteps; ++s) {
        for (size_t t = 0; t < E.terms.size(); ++t) {            // resampling strategies: fresh points every evaluation, on device
            Term& T = E.terms[t];
            if (T.sampler != 0) {
                aux::launch_sample(T.sampler, T.d_pts, (int)(T.n * T.d), T.d, T.d_lb, T.d_ub, T.seed, T.draws++, E.stream);
                eval_sources(E, T);
            }
        }
        if (run_loss_grad(E, E.d_opt_theta, E.d_opt_out, term_w, -1, false)) return 1;
        aux::launch_total_loss(E.d_hist, s, E.d_opt_out, P, K, E.d_w_over_n, E.stream);
        ++E.opt_t;
        const float c1 = (float)(1.0 / (1.0 - std::pow((double)beta1, (double)E.opt_t)));
        const float c2 = (float)(1.0 / (1.0 - std::pow((double)beta2, (double)E.opt_t)));
        aux::launch_adam(E.d_opt_theta, E.d_opt_m, E.d_opt_v, E.d_opt_out, P, lr, beta1, beta2, eps, c1, c2, E.stream);
    }
    if (loss_history && plat_d2h(loss_history, E.d_hist, sizeof(double) * nsteps, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_adam_get(pinn_handle h, float* theta, int64_t p) {
    if (!h || !theta) return fail("pinn_adam_get: null argument");
    pinn_engine& E = *h;
    if (!E.d_opt_theta || p != E.ntheta) return fail("pinn_adam_get: no optimiser state / length mismatch");
    if (plat_d2h(theta, E.d_opt_theta, sizeof(float) * p, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_set_timing(pinn_handle h, int level, int group) {
    if (!h) return fail("pinn_set_timing: null handle");
    if (level < 0 || level > 2) return fail("pinn_set_timing: level must be 0, 1 or 2");
    if (group < -1 || group >= (int)h->groups.size()) return fail("pinn_set_timing: group index out of range");
    h->timing_level = level; h->timing_group = group;
    h->timing_valid = false;
    for (auto& G : h->groups) G.timed = false;
    return 0;
}

int pinn_last_timing(pinn_handle h, float* kernel_ms, float* total_ms) {
    if (!h) return fail("null handle");
    if (!h->timing_valid) return fail("no timing available (call pinn_loss_grad first)");
    plat_event_sync(h->ev3);
    if (kernel_ms) *kernel_ms = plat_event_ms(h->ev1, h->ev2);
    if (total_ms) *total_ms = plat_event_ms(h->ev0, h->ev3);
    return 0;
}

int pinn_num_groups(pinn_handle h) { return h ? (int)h->groups.size() : -1; }

#ifdef PINN_STAMP
// profiling build only (tools/stamp_profile.sh): raw read-back of one workgroup's gradient slab incl. the stamp tail
int pinn_debug_slab(pinn_handle h, int group, int blk, float* dst, int64_t n) {
    if (!h || group < 0 || group >= (int)h->groups.size()) return -1;
    Group& G = h->groups[group];
    if (blk < 0) return G.blocks;
    if (blk == (1 << 30)) return G.spec->SLAB;
    if (blk >= G.blocks || n > G.spec->SLAB) return -1;
    plat_sync(h->stream);
    plat_d2h(dst, G.d_slabs + (size_t)blk * G.spec->SLAB + (G.spec->SLAB - n), sizeof(float) * n, h->stream);
    plat_sync(h->stream);
    return 0;
}
#endif

int pinn_group_timing(pinn_handle h, int group, float* ms, int64_t* points, int* channels, int* tiles) {
    if (!h) return fail("null handle");
    if (group < 0 || group >= (int)h->groups.size()) return fail("pinn_group_timing: group index out of range");
    Group& G = h->groups[group];
    if (ms) *ms = -1.f;                   // not timed in the last evaluation (pinn_set_timing)
    if (G.timed) {
        plat_event_sync(G.ev_b);
        if (ms) *ms = plat_event_ms(G.ev_a, G.ev_b);
    }
    int64_t n = 0;
    for (int t : G.terms) n += h->terms[t].n;
    if (points) *points = n;
    if (channels) *channels = G.spec->C;
    if (tiles) *tiles = G.ga.ntiles;
    return 0;
}

int pinn_describe(pinn_handle h, char* buf, int64_t buflen) {
    if (!h || !buf || buflen <= 0) return fail("pinn_describe: bad argument");
    std::ostringstream os;
    os << "backend=" << plat_name() << " cus=" << h->ncu << " ntheta=" << h->ntheta << " terms=" << h->terms.size() << "\n";
    for (size_t g = 0; g < h->groups.size(); ++g) {
        const Group& G = h->groups[g];
        os << "group " << g << (G.kind == 1 ? (G.use_rec ? " [coupled fwd/gradin, records in HBM]" : " [coupled fwd/gradin]") : "") << " net=" << G.net << " kernel=" << spec_name(*G.spec) << " tiles=" << G.ga.ntiles << " blocks=" << G.blocks << " terms=";
        for (int t : G.terms) os << t << ",";
        os << "\n";
    }
    for (size_t t = 0; t < h->terms.size(); ++t) {
        const Term& T = h->terms[t];
        if (T.coupled >= 0) continue;
        os << "term " << t << ": tape ops=" << T.tape_ops.size() << " of " << T.ops.size() << ", sources=" << T.src_root.size()
           << " (" << T.src_prog.size() << " coordinate-only ops evaluated per point set)\n";
    }
    std::string s = os.str();
    std::snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return 0;
}

}  // extern "C"
