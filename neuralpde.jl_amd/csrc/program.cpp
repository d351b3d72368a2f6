// program.cpp — passes over a term's residual program in descriptor numbering: coordinate-only subexpressions -> sources
// (analyse_static), forward-Laplacian fusion (fuse_laplacian).
#include "engine_types.hpp"

namespace pe {

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
    // sums are trees of ADD and SUB ops (u_t - u_xx - u_yy arrives as SUB(SUB(u_t, u_xx), u_yy)): every leaf carries the sign it enters with
    auto is_add = [&](int row) { return row >= rop0 && (T.ops[row - rop0].code == rp::OP_ADD || T.ops[row - rop0].code == rp::OP_SUB); };
    auto inner = [&](int row) { return is_add(row) && uses[row] == 1; };         // ADD / SUB node that only feeds its parent node
    auto cand = [&](int row) {                                                     // pure second derivative, used exactly once
        if (row < rslot0 || row >= rop0 || uses[row] != 1) return false;
        const Slot& s = T.slots[row - rslot0];
        return s.lap == 0 && s.order == 2 && s.axes[0] == s.axes[1];
    };
    // roots: ADD ops that are not themselves inner nodes of a larger ADD tree
    std::vector<char> is_inner_child(nops, 0);
    for (int q = 0; q < nops; ++q)
        if (is_add(rop0 + q)) {
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
    struct Tree { int root; std::vector<int> leaves, nodes; std::vector<float> sign; std::vector<int> fused; std::vector<int> fused_ops; unsigned mask; int net; float coef; };
    std::vector<Tree> trees;
    for (int q = 0; q < nops; ++q) {
        if (!is_add(rop0 + q) || is_inner_child[q]) continue;
        Tree tr; tr.root = q; tr.mask = 0; tr.net = -1;
        std::vector<std::pair<int, float>> stack{{rop0 + q, 1.0f}};
        while (!stack.empty()) {
            const int row = stack.back().first;
            const float sg = stack.back().second;
            stack.pop_back();
            tr.nodes.push_back(row - rop0);
            const rp::Instr& N = T.ops[row - rop0];
            const float sb = N.code == rp::OP_SUB ? -sg : sg;
            for (auto ch : {std::make_pair(N.b, sb), std::make_pair(N.a, sg)}) {
                if (inner(ch.first)) stack.push_back(ch);
                else { tr.leaves.push_back(ch.first); tr.sign.push_back(ch.second); }
            }
        }
        // candidate leaves of one network with one common coefficient and distinct axes
        std::map<std::pair<int, float>, std::vector<int>> by_key;
        for (size_t li = 0; li < tr.leaves.size(); ++li) {
            const Leaf lf = leaf_of(tr.leaves[li]);
            if (lf.slot_row >= 0) by_key[{T.slots[lf.slot_row - rslot0].net, lf.coef * tr.sign[li]}].push_back(tr.leaves[li]);
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
        Slot L; L.net = trees[i].net; L.order = 2; for (int a = 0; a < MAX_DERIV_ORDER; ++a) L.axes[a] = 0; L.lap = trees[i].mask;
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
        int acc = lap_row;
        for (size_t li = 0; li < tr.leaves.size(); ++li) {
            const int leaf = tr.leaves[li];
            if (std::find(tr.fused.begin(), tr.fused.end(), leaf) != tr.fused.end()) continue;
            rp::Instr I{};
            I.code = tr.sign[li] < 0.f ? rp::OP_SUB : rp::OP_ADD; I.a = acc; I.b = map_row(leaf); I.imm = 0.f;      // the leaf keeps its sign
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

}  // namespace pe
