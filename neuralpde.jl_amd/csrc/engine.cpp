// engine.cpp — host side of the C ABI: descriptor parsing, kernel-plan construction, launches.
//
// Mirrors what the reference does once per `discretize` call on the host
// (src/discretize.jl:413-767: build loss functions, merge with the strategy, build full_loss_function)
// and what it does per optimiser iteration (src/discretize.jl:567-598).
#include "engine_types.hpp"
#include "aux_kernels.hpp"

#ifdef PINN_EMU
namespace wv {
thread_local void (*emu_barrier_hook)(void*) = nullptr;
thread_local void* emu_barrier_ctx = nullptr;
thread_local void (*emu_grid_hook)(void*) = nullptr;
thread_local void* emu_grid_ctx = nullptr;
}  // namespace wv
#endif

namespace pk {
std::deque<SpecInfo>& registry() {
    static std::deque<SpecInfo> r;
    return r;
}
std::deque<PairInfo>& pair_registry() {
    static std::deque<PairInfo> r;
    return r;
}
}  // namespace pk

namespace pe {
thread_local std::string g_err;
int fail(const std::string& m) {
    g_err = m;
    return 1;
}
}  // namespace pe

using namespace pe;

namespace {

// (re-)evaluate a term's coordinate-only source channels for its current point set
// embedded terms (periodic input embedding): the installed point set lives in d_upts [n][d_user]; the device rows [n][d] = the
// coordinates followed by their sin / cos rows are rebuilt from it whenever it changes (set_points, every sampler draw)
inline float* user_pts(Term& T) { return T.emb_cols.empty() ? T.d_pts : T.d_upts; }
void embed_points(pinn_engine& E, Term& T) {
    if (T.emb_cols.empty() || !T.d_upts || T.n <= 0) return;
    aux::EmbedArgs a;
    std::memset(&a, 0, sizeof a);
    a.upts = T.d_upts; a.pts = T.d_pts; a.n = (int)T.n; a.du = T.d_user; a.dx = T.d;
    for (size_t k = 0; k < T.emb_cols.size() && k < 4; ++k) { a.src[k] = T.emb_cols[k].src; a.is_cos[k] = T.emb_cols[k].is_cos; a.omega[k] = T.emb_cols[k].omega; }
    aux::launch_embed(a, E.stream);
}
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

// the weights every network's kernels read in this evaluation: the packed image rebuilt by k_pack (DGM kernels read theta itself)
void pack_all(pinn_engine& E, const float* d_theta = nullptr, bool packed_fresh = false) {
    const float* th = d_theta ? d_theta : E.d_theta;
    for (size_t n = 0; n < E.nets.size(); ++n) {
        NetPlan& NP = E.netplans[n];
        if (!NP.spec) continue;
        if (NP.spec->family == 3) { NP.cur = th + E.nets[n].theta_off; continue; }          // DGM kernels read theta unpacked
        NP.cur = NP.d_packed;
        const bool gather = !packed_fresh;                         // (fresh: the optimiser's update kernel already wrote the fp32 image)
        if (gather && !NP.spec->BFX) aux::launch_pack(NP.d_packed, NP.d_pack_idx, th, NP.npacked, E.stream);
        if (NP.spec->BFX) {                                        // split-operand GEMMs: the three bf16 pieces of every hidden->hidden weight,
                                                                   // written by the same launch as the fp32 image
            const Net& N = E.nets[n];
            aux::PackBfArgs pa;
            std::memset(&pa, 0, sizeof pa);
            pa.theta = th; pa.out_fwd = (unsigned*)(NP.d_packed + NP.spec->OFF_WB); pa.out_tr = (unsigned*)(NP.d_packed + NP.spec->OFF_WTB);
            pa.nhh = NP.spec->NHH; pa.hp = NP.spec->HP;
            pa.packed = NP.d_packed; pa.idx = NP.d_pack_idx; pa.n_gather = gather ? NP.spec->OFF_WB : 0;      // (the fp32 part of the image ends where the bf16 pieces begin)
            if (pa.nhh > aux::PACK_BF_MAX_LAYERS) return;          // (build_plan refuses such nets)
            int o = N.theta_off;
            for (size_t j = 0; j + 1 < N.sizes.size(); ++j) {
                if (j >= 1 && j <= (size_t)pa.nhh) { pa.woff[j - 1] = o; pa.nin[j - 1] = N.sizes[j]; pa.nout[j - 1] = N.sizes[j + 1]; }
                o += N.sizes[j + 1] * N.sizes[j] + N.sizes[j + 1];
            }
            aux::launch_pack_bf16(pa, E.stream);
        }
    }
    for (auto& G : E.groups) G.ga.packed = E.netplans[G.net].cur;
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

}  // namespace

// the device section shared by all loss/grad entry points.  d_theta: theta in device memory; d_out: [P + K] floats in
// device memory.  Launches per evaluation: pack (1 per net) -> fused residual kernel (1 per group, or 1 per MERGED pair of groups)
// -> reduction (one kernel when a single slab set carries the gradient, else reduce1 -> reduce2).
// loss_only: MODE_LOSS launches (forward + tape + sums of squares; coupled equations: forward launches + k_expr without adjoints), only
// the K sums are reduced and d_out[0, P) is left untouched.
int pe::run_loss_grad(pinn_engine& E, const float* d_theta, float* d_out, const float* term_w, int only_term /* -1 = all */, bool timing,
                  double* lossraw /* K exact double sums; default: E.d_lossraw */, bool packed_fresh, bool loss_only, float* d_sums) {
    if (!d_sums) d_sums = d_out + E.ntheta;              // the K sums follow the gradient unless the caller has a buffer of their own (loss-only)
    if (ensure_points(E)) return 1;
    const int K = (int)E.terms.size();
    const bool phase_ev = timing && E.timing_level >= 2;
    auto group_ev = [&](size_t g) { return timing && E.timing_level >= 1 && (E.timing_group < 0 || E.timing_group == (int)g); };
    if (phase_ev) plat_event_record(E.ev0, E.stream);
    pack_all(E, d_theta, packed_fresh);
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
    static const bool no_chain = std::getenv("PINN_NO_CHAIN") != nullptr;      // A/B switches
    const bool no_merge = std::getenv("PINN_NO_MERGE") != nullptr;            // (read per call: the tests switch them)
    const bool may_merge = !no_merge && !no_chain && !concurrent && only_term < 0;
    int nforked = 0;
    int nslabsets = 0, slabset_group = -1;          // launch groups whose own slab set carries gradient sums in this evaluation
    if (concurrent) plat_event_record(E.ev_fork, E.stream);
    for (size_t g = 0; g < E.groups.size(); ++g) {
        Group& G = E.groups[g];
        bool any = false;
        for (size_t j = 0; j < G.terms.size(); ++j) {
            G.ga.terms[j].scale = scale_of(G.terms[j]);
            any = any || (only_term < 0 || only_term == G.terms[j]);
        }
        G.active = any;
        G.timed = false;
        const MergedUnit* mu = (may_merge && G.kind == 0 && G.merged >= 0) ? &E.merged[G.merged] : nullptr;
        a1.tmp[g] = G.d_tmp; a1.slabs[g] = G.d_slabs; a1.losspart[g] = G.d_losspart;
        a2.tmp[g] = G.d_tmp;
        if (mu && (int)g == mu->tail) {
            // this group's tiles ran in the head's merged launch: its gradient and its loss partials are in the head's buffers
            G.launched_by = mu->head;
            G.launched_blocks = E.groups[mu->head].launched_blocks;
            a1.active[g] = 0; a2.active[g] = 0; a2.ent_active[g] = 0;
            a1.slab[g] = G.slab_floats; a1.nblocks[g] = 0; a1.nsplit[g] = 1; a1.nent[g] = 0; a1.nwpb[g] = G.spec->NW;
            a2.stride[g] = K; a2.nsplit[g] = 1; a2.nent[g] = 0;
            continue;
        }
        // chained launch groups: this group's workgroups add their sums onto the slabs the head group (same network, launched
        // earlier on the same stream in this evaluation) has just written, so the reduction reads one slab set, not two
        const bool chained = any && !loss_only && !no_chain && !concurrent && G.kind == 0 && G.chain_to >= 0 && E.groups[G.chain_to].active &&
                             G.blocks <= E.groups[G.chain_to].launched_blocks;
        const int nent = (chained || loss_only) ? 0 : G.nent;
        G.ga.slabs = chained ? E.groups[G.chain_to].d_slabs : G.d_slabs;
        G.ga.chain = chained ? 1 : 0;
        // a single-term evaluation (pinn_term_grads) of a fused group launches ONLY that term's tiles: the K per-term gradients then cost
        // about one full evaluation in total instead of K
        pk::GroupArgs ga_one;
        const pk::GroupArgs* ga_launch = &G.ga;
        int blocks = G.blocks;
        if (any && only_term >= 0 && G.kind == 0 && G.terms.size() > 1) {
            ga_one = G.ga;
            for (size_t j = 0; j < G.terms.size(); ++j)
                if (G.terms[j] == only_term) ga_one.terms[0] = G.ga.terms[j];
            ga_one.nterms = 1;
            ga_one.terms[0].tile0 = 0;
            ga_one.ntiles = ga_one.terms[0].ntiles;
            blocks = std::max(1, std::min(G.max_blocks, G.spec->family == 1 ? (ga_one.ntiles + 3) / 4 : ga_one.ntiles));
            ga_launch = &ga_one;
        }
        const bool merged_head = mu && (int)g == mu->head;
        if (merged_head) {
            // ONE launch for this group's tiles and the tail group's: terms and tiles concatenated, the tail's after the head's
            const Group& T2 = E.groups[mu->tail];
            ga_one = G.ga;
            for (size_t j = 0; j < T2.terms.size(); ++j) {
                pk::TermDev td = T2.ga.terms[j];
                td.tile0 += G.ga.ntiles;
                td.scale = scale_of(T2.terms[j]);
                ga_one.terms[G.terms.size() + j] = td;
            }
            ga_one.nterms = (int)(G.terms.size() + T2.terms.size());
            ga_one.sub_terms0 = (int)G.terms.size();
            ga_one.sub_tiles0 = G.ga.ntiles;
            ga_one.ntiles = G.ga.ntiles + T2.ga.ntiles;
            ga_one.scratch = mu->d_scratch;
            ga_one.scr_stride = mu->pair->SCR;
            ga_one.losspart = mu->d_losspart;
            a1.losspart[g] = mu->d_losspart;
            ga_one.chain = 0;
            blocks = std::max(1, std::min(loss_only ? E.ncu * mu->pair->WG_FWD : mu->max_blocks, ga_one.ntiles));
            ga_launch = &ga_one;
        }
        G.launched_blocks = blocks;
        G.launched_by = (int)g;
        // stage-1 chunks: the serial part of the two-stage sum is (blocks / chunks) loads in stage 1 plus (chunks x slab entries per theta
        // element: 4 per-wave copies in family 1, 1 in the others) in stage 2 — shortest for chunks ~ sqrt(blocks / entries)
        const int epe = G.spec->family == 1 ? 4 : 1;
        const int nsplit_g = std::max(1, std::min(REDUCE_SPLIT, (int)std::ceil(std::sqrt((double)blocks / epe))));
        a1.slab[g] = G.slab_floats; a1.nblocks[g] = blocks; a1.nsplit[g] = nsplit_g; a1.nent[g] = nent; a1.active[g] = any; a1.nwpb[g] = G.spec->NW;
        a2.stride[g] = nent + K; a2.nsplit[g] = nsplit_g; a2.nent[g] = nent; a2.active[g] = any;
        a2.ent_active[g] = any && !chained && !loss_only;
        if (!any) continue;
        if (a2.ent_active[g]) { ++nslabsets; slabset_group = (int)g; }
        max_n1 = std::max(max_n1, nent / 4 + K);
        max_split = std::max(max_split, nsplit_g);
        if (G.kind == 1) {               // coupled: forward launch now, reverse launch after k_expr / the equation's tail launch
            G.spec->launch(G.ga, (G.use_rec && !loss_only) ? pk::MODE_FWDREC : pk::MODE_FWD,
                           std::max(1, std::min(E.ncu * G.spec->WG_FWD, G.spec->family == 1 ? (G.ga.ntiles + 3) / 4 : G.ga.ntiles)), E.stream);
            continue;
        }
        if (G.kind == 2) continue;       // tail launch of a coupled equation: after the other networks' forward launches (below)
        plat_stream st = E.stream;
        const bool forked = concurrent && nforked++ > 0;       // the first fused group stays on the caller's stream
        if (forked) {
            st = E.aux_stream[nforked % 2];
            plat_stream_wait_event(st, E.ev_fork);
        }
        if (group_ev(g)) plat_event_record(G.ev_a, st);
        if (merged_head) mu->pair->launch(*ga_launch, loss_only ? pk::MODE_LOSS : pk::MODE_FUSED, blocks, st);
        else if (loss_only) {
            const int tiles = ga_launch->ntiles;
            const int fb = std::max(1, std::min(E.ncu * G.spec->WG_FWD, G.spec->family == 1 ? (tiles + 3) / 4 : tiles));
            a1.nblocks[g] = fb;                            // rows of loss partials this launch writes
            G.launched_blocks = fb;
            G.spec->launch(*ga_launch, pk::MODE_LOSS, fb, st);
        } else G.spec->launch(*ga_launch, pk::MODE_FUSED, blocks, st);
        if (group_ev(g)) plat_event_record(G.ev_b, st);
        G.timed = group_ev(g);
        if (forked) {
            plat_event_record(E.ev_join[g], st);
            plat_stream_wait_event(E.stream, E.ev_join[g]);
        }
    }
    bool coupled_active = false;
    for (size_t c = 0; c < E.coupled.size(); ++c) {
        Coupled& Cp = E.coupled[c];
        const int g = (int)(E.groups.size() + c);
        const bool on = (only_term < 0 || only_term == Cp.term);
        const int nsplit = std::min(REDUCE_SPLIT, Cp.blocks);
        const int nent = loss_only ? 0 : 16;
        a1.tmp[g] = Cp.d_tmp; a1.slabs[g] = Cp.d_pslab; a1.losspart[g] = Cp.d_losspart;
        a1.slab[g] = 16; a1.nblocks[g] = Cp.blocks; a1.nsplit[g] = nsplit; a1.nent[g] = nent; a1.active[g] = on; a1.nwpb[g] = 4;
        a2.tmp[g] = Cp.d_tmp; a2.stride[g] = nent + K; a2.nsplit[g] = nsplit; a2.nent[g] = nent; a2.active[g] = on; a2.ent_active[g] = on && !loss_only;
        bool groups_active = false;
        for (int gi : Cp.groups) groups_active = groups_active || E.groups[gi].active;
        if (!groups_active) continue;
        coupled_active = true;
        if (Cp.tail >= 0) {
            // TAIL launch: forward pass of the widest network + the equation's tape (the other networks' jets as source rows) + its reverse
            // sweep in one kernel; it writes the other networks' seeds and carries the term's loss partials and parameter gradients
            // itself, so the k_expr pseudo-group stays out of the reduction
            Group& G = E.groups[Cp.groups[Cp.tail]];
            a1.active[g] = 0; a2.active[g] = 0; a2.ent_active[g] = 0;
            if (group_ev((size_t)Cp.groups[Cp.tail])) plat_event_record(G.ev_a, E.stream);
            if (loss_only) {
                const int fb = std::max(1, std::min(E.ncu * G.spec->WG_FWD, G.ga.ntiles));
                a1.nblocks[Cp.groups[Cp.tail]] = fb;
                G.launched_blocks = fb;
                G.spec->launch(G.ga, pk::MODE_LOSS, fb, E.stream);
            } else G.spec->launch(G.ga, pk::MODE_FUSED, G.blocks, E.stream);
            if (group_ev((size_t)Cp.groups[Cp.tail])) plat_event_record(G.ev_b, E.stream);
            G.timed = group_ev((size_t)Cp.groups[Cp.tail]);
            continue;
        }
        max_n1 = std::max(max_n1, nent / 4 + K);
        max_split = std::max(max_split, nsplit);
        aux::ExprArgs ea = expr_args(E, Cp, scale_of(Cp.term), nullptr);    // scale 0 => zero seeds
        ea.loss_only = loss_only ? 1 : 0;
        aux::launch_expr(ea, Cp.blocks, E.stream);
    }
    for (size_t g = 0; g < E.groups.size() && !loss_only; ++g) {
        Group& G = E.groups[g];
        if (G.kind != 1 || !G.active) continue;
        if (group_ev(g)) plat_event_record(G.ev_a, E.stream);
        G.spec->launch(G.ga, G.use_rec ? pk::MODE_GRADREC : pk::MODE_GRADIN, G.blocks, E.stream);
        if (group_ev(g)) plat_event_record(G.ev_b, E.stream);
        G.timed = group_ev(g);
    }
    if (phase_ev) plat_event_record(E.ev2, E.stream);
    a1.K = K;
    a2.out = d_out; a2.sums = d_sums; a2.lossraw = lossraw ? lossraw : E.d_lossraw; a2.row_ptr = E.d_gr_ptr; a2.row_grp = E.d_gr_grp; a2.row_ent = E.d_gr_ent;
    a2.ngroups = (int)(E.groups.size() + E.coupled.size()); a2.P = (int)E.ntheta; a2.K = K;
    a2.skip_grad = loss_only ? 1 : 0;
    a2.perm = std::getenv("PINN_NO_REDUCE_PERM") ? nullptr : E.d_red_perm;
    // one slab set carries the whole gradient (a single network whose launch groups are merged / chained): one reduction kernel
    const bool no_reduce_one = std::getenv("PINN_NO_REDUCE_ONE") != nullptr;
    const Group* RG = (nslabsets == 1 && !coupled_active && !loss_only && !no_reduce_one) ? &E.groups[slabset_group] : nullptr;
    const bool one_kernel = ((RG && RG->d_ent_theta && RG->ent_covers_theta) || (loss_only && !no_reduce_one)) && !aux::reduce_is_small(a1, a2);
    if (one_kernel) {
        aux::ReduceOneArgs ro;
        std::memset(&ro, 0, sizeof ro);
        if (RG) { ro.slabs = RG->d_slabs; ro.slab = RG->slab_floats; ro.nblocks = a1.nblocks[slabset_group]; ro.nent = RG->nent; ro.ent_theta = RG->d_ent_theta; }
        ro.out = d_out; ro.sums = d_sums; ro.lossraw = a2.lossraw; ro.P = (int)E.ntheta; ro.K = K;       // (loss-only: nent = 0, only the K loss blocks run)
        for (int g = 0; g < a2.ngroups; ++g)
            if (a1.active[g]) { ro.losspart[ro.nloss] = a1.losspart[g]; ro.nrows[ro.nloss] = a1.nblocks[g] * a1.nwpb[g]; ++ro.nloss; }
        aux::launch_reduce_one(ro, E.stream);
    } else {
        aux::launch_reduce(a1, a2, max_n1, max_split, E.stream);
    }
    if (phase_ev) plat_event_record(E.ev3, E.stream);
    // a run-time specialised kernel whose code object could not be produced / loaded at its first launch (jit.cpp: rtc_launch) was NOT launched
    { const std::string e = jit_take_launch_error(); if (!e.empty()) return fail(e); }
    return 0;
}

void pe::sums_from_double(float* d_out_sums, const double* d_raw, int K, plat_stream st) { aux::launch_sums_from_double(d_out_sums, d_raw, K, st); }

int pe::upload_theta(pinn_engine& E, const float* theta, int64_t p) {
    if (p != E.ntheta) return fail("theta length " + std::to_string(p) + " != ntheta " + std::to_string(E.ntheta));
    std::memcpy(E.hp_theta, theta, sizeof(float) * p);          // pinned staging: the copy below is a plain DMA, no pageable bounce
    if (plat_h2d(E.d_theta, E.hp_theta, sizeof(float) * p, E.stream)) return fail(std::string("H2D copy of theta failed: ") + plat_last_error());
    return 0;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* pinn_backend(void) { return plat_name(); }
int pinn_abi_version(void) { return 1; }
const char* pinn_last_error(void) { return g_err.c_str(); }

int pinn_create(const char* descriptor, pinn_handle* out) { return pinn_create_on(descriptor, -1, out); }

int pinn_create_on(const char* descriptor, int device, pinn_handle* out) {
    if (!descriptor || !out) return fail("pinn_create: null argument");
    *out = nullptr;
    std::string err;
    if (plat_init(err)) return fail(err);
    if (device >= plat_device_count()) return fail("pinn_create_on: device " + std::to_string(device) + " does not exist (" + std::to_string(plat_device_count()) + " visible)");
    const int dev = device < 0 ? plat_get_device() : device;
    DeviceScope scope(dev);
    std::unique_ptr<pinn_engine> E(new pinn_engine());
    E->device = dev;
    if (const char* gm = std::getenv("PINN_GEMM")) {             // default GEMM arithmetic of new handles (pinn_set_option switches a live one)
        if (std::string(gm) == "fp32") E->gemm = pk::GEMM_FP32;
        else if (std::string(gm) == "split") E->gemm = pk::GEMM_SPLIT;
        else return fail(std::string("PINN_GEMM must be \"split\" or \"fp32\", not \"") + gm + "\"");
    }
    if (parse_descriptor(descriptor, *E)) return 1;
    E->terms0 = E->terms;                              // (before the planner rewrites them: the float64 mode evaluates the terms as parsed)
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
    // one pinned block for the host entry points; the reduction writes its results straight into it
    E->hp_theta = (float*)plat_host_alloc(sizeof(float) * (2 * E->ntheta + K + 2) + sizeof(double) * K);
    if (!E->hp_theta) {
        pinn_destroy(E.release());
        return fail("pinned host allocation failed");
    }
    E->hp_out = E->hp_theta + E->ntheta;
    E->hp_raw = (double*)(E->hp_theta + ((2 * E->ntheta + K + 1) / 2) * 2);
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
    DeviceScope scope(E.device);
    pinn_comm_destroy(h);
    plat_sync(E.stream);
    f64_destroy(E);
    free_plan(E);
    for (auto& T : E.terms) { plat_free(T.d_pts); plat_free(T.d_upts); plat_free(T.d_resid); plat_free(T.d_lb); plat_free(T.d_ub); plat_free(T.d_data); plat_free(T.d_pw); }
    plat_free(E.d_opt_theta); plat_free(E.d_opt_m); plat_free(E.d_opt_v); plat_free(E.d_opt_out); plat_free(E.d_w_over_n); plat_free(E.d_hist); plat_free(E.d_step); plat_free(E.d_draws); plat_free(E.d_sampled); plat_free(E.d_c12); plat_free(E.d_bar); plat_free(E.d_sums2); plat_free(E.d_own_r); plat_free(E.d_train_samp); plat_free(E.d_opt_bak); plat_host_free(E.h_flag);
    plat_free(E.d_theta); plat_free(E.d_params); plat_free(E.d_defaults); plat_free(E.d_lossraw);
    plat_free(E.d_out); plat_free(E.d_phi_pts); plat_free(E.d_phi_out); plat_free(E.d_phi_scr);
    plat_host_free(E.hp_theta);
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

static int term_installed(pinn_engine& E, int term);
static int set_points_impl(pinn_handle h, int term, const float* pts, int64_t n, int64_t n_norm, bool device) {
    if (!h) return fail("null handle");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_set_points: term index out of range");
    if (!pts || n <= 0) return fail("pinn_set_points: empty point set (the reference's mean(abs2, .) over an empty set is NaN; refusing)");
    Term& T = E.terms[term];
    if (n * T.d >= (int64_t)1 << 31) return fail("pinn_set_points: point set too large for 32-bit indexing; shard it");
    if (n > T.pts_cap || !T.d_pts) {                     // grow only: a resampled set of another size reuses the buffer
        plat_sync(E.stream);                             // (an evaluation in flight may still read the old buffer)
        plat_free(T.d_pts);
        plat_free(T.d_upts);
        T.d_upts = nullptr;
        T.pts_cap = 0;
        T.d_pts = (float*)plat_malloc(sizeof(float) * n * T.d);
        if (!T.emb_cols.empty()) T.d_upts = (float*)plat_malloc(sizeof(float) * n * T.d_user);
        if (!T.d_pts || (!T.emb_cols.empty() && !T.d_upts)) return fail("device allocation failed (points)");
        T.pts_cap = n;
    }
    int rc = device ? plat_d2d(user_pts(T), pts, sizeof(float) * n * T.d_user, E.stream) : plat_h2d(user_pts(T), pts, sizeof(float) * n * T.d_user, E.stream);
    if (rc) return fail(std::string("copy of points failed: ") + plat_last_error());
    T.n = n;
    embed_points(E, T);
    plat_sync(E.stream);
    T.n_norm = n_norm > 0 ? n_norm : n;
    T.data_n = 0;                                        // per-point data and weights belong to the previous set
    T.pw_n = 0;
    if (term_installed(E, term)) return 1;
    return f64_points_changed(E, term);                  // (float64 mode: the double copy follows)
}

// what follows the installation of a term's point set (also after a re-plan, pinn_set_option): source channels, tile tables, the buffers of
// a coupled equation
static int term_installed(pinn_engine& E, int term) {
    Term& T = E.terms[term];
    const int64_t n = T.n;
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
        for (size_t i = 0; i < Cp.d_jets.size(); ++i)
            if (Cp.tail < 0 || (int)i == Cp.tail) { plat_free(Cp.d_jets[i]); plat_free(Cp.d_ubar[i]); }
        plat_free(Cp.d_jets_all); plat_free(Cp.d_ubar_all);
        Cp.d_jets_all = Cp.d_ubar_all = nullptr;
        Cp.d_jets.assign(Cp.nets.size(), nullptr);
        Cp.d_ubar.assign(Cp.nets.size(), nullptr);
        Cp.cap = 0;
        if (Cp.tail >= 0 && Cp.nsrc > 0) {               // the other networks' channels in ONE array each way: the tail launch's source rows
            Cp.d_jets_all = (float*)plat_malloc(sizeof(float) * (size_t)Cp.nsrc * n);
            Cp.d_ubar_all = (float*)plat_malloc(sizeof(float) * (size_t)Cp.nsrc * n);
            if (!Cp.d_jets_all || !Cp.d_ubar_all) return fail("device allocation failed (coupled jets)");
        }
        for (size_t i = 0; i < Cp.nets.size(); ++i) {
            if (Cp.tail >= 0 && (int)i != Cp.tail) continue;
            const int C = E.groups[Cp.groups[i]].spec->C;
            Cp.d_jets[i] = (float*)plat_malloc(sizeof(float) * (size_t)C * n);
            Cp.d_ubar[i] = (float*)plat_malloc(sizeof(float) * (size_t)C * n);
            if (!Cp.d_jets[i] || !Cp.d_ubar[i]) return fail("device allocation failed (coupled jets)");
        }
        Cp.cap = n;
    }
    if (Cp.tail >= 0)                                    // rows packed with stride n (the kernels address channel c at c * N)
        for (size_t i = 0; i < Cp.nets.size(); ++i)
            if ((int)i != Cp.tail) { Cp.d_jets[i] = Cp.d_jets_all + (size_t)Cp.src_off[i] * n; Cp.d_ubar[i] = Cp.d_ubar_all + (size_t)Cp.src_off[i] * n; }
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
int pinn_set_points_f64(pinn_handle h, int term, const double* pts, int64_t n, int64_t n_norm) {
    if (!h || !pts || n <= 0) return fail("pinn_set_points_f64: null argument / empty point set");
    if (term < 0 || term >= (int)h->terms.size()) return fail("pinn_set_points_f64: term index out of range");
    const int d = h->terms[term].d_user;
    std::vector<float> p32((size_t)n * d);
    for (size_t i = 0; i < p32.size(); ++i) p32[i] = (float)pts[i];
    if (set_points_impl(h, term, p32.data(), n, n_norm, false)) return 1;       // the fp32 kernels' copy (EltypeAdaptor, src/eltype_matching.jl:8-10)
    if (!h->f64) return 0;
    DeviceScope scope(h->device);
    return f64_set_points(*h, term, pts, n);                                    // the float64 mode reads the points as given
}

static void measure_grad_health(pinn_engine& E, const float* grad_and_sums, const float* term_w);
static int gemm_auto_check(pinn_engine& E, const float* d_theta, const float* term_w, long long now);
static bool eval_eligible(pinn_engine& E);
static int eval_fused(pinn_engine& E, const float* d_theta, float* d_out, const float* term_w, double* lossraw);
static bool train_timed_out(pinn_engine& E);
// loss + gradient for a host entry point that synchronises right after: small problems in ONE launch (eval_fused), everything else by
// run_loss_grad; results in d_out / lossraw either way.  Synchronises.
static int eval_and_sync(pinn_engine& E, float* d_out, const float* term_w, double* lossraw, bool timing) {
    E.eval_path = 1;
    if (ensure_points(E)) return 1;                  // (every path: the one-launch evaluation below does not go through run_loss_grad's check — ADVICE r04)
    if (eval_eligible(E)) {
        if (eval_fused(E, E.d_theta, d_out, term_w, lossraw)) return 1;
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        if (!train_timed_out(E)) { E.eval_path = 2; return 0; }           // (timed out: the stand-alone kernels repeat the evaluation)
    }
    if (run_loss_grad(E, E.d_theta, d_out, term_w, -1, timing, lossraw, false, false)) return 1;
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_loss_grad(pinn_handle h, const float* theta, int64_t p, const float* term_w, double* term_losses, float* grad) {
    if (!h || !theta) return fail("pinn_loss_grad: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    const int K = (int)E.terms.size();
    if (E.f64) {                                         // float64 mode: converted at the boundary, evaluated in double
        if (p != E.ntheta) return fail("theta length " + std::to_string(p) + " != ntheta " + std::to_string(E.ntheta));
        std::vector<double> th(theta, theta + p), w(K, 1.0), g(grad ? p : 0);
        if (term_w) for (int k = 0; k < K; ++k) w[k] = term_w[k];
        if (f64_eval(E, th.data(), w.data(), term_losses, grad ? g.data() : nullptr)) return 1;
        if (grad) for (int64_t i = 0; i < p; ++i) grad[i] = (float)g[i];
        return 0;
    }
    if (upload_theta(E, theta, p)) return 1;
    // grad == NULL: loss-only evaluation (no records, no reverse sweep, no gradient reduction) — what a callback, an adaptive-weight
    // rule or a rejected line-search trial needs (the reference's per-term closures are value-only unless differentiated,
    // src/training_strategies.jl:215-221)
    if (grad) {
        if (eval_and_sync(E, E.hp_out, term_w, E.hp_raw, true)) return 1;                                            // results land in pinned host memory
    } else {
        if (run_loss_grad(E, E.d_theta, E.hp_out, term_w, -1, true, E.hp_raw, false, true)) return 1;
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    E.timing_valid = E.timing_level >= 2;
    if (term_losses)
        for (int k = 0; k < K; ++k) term_losses[k] = E.hp_raw[k] / (double)E.terms[k].n_norm;   // exact double sums
    if (grad) std::memcpy(grad, E.hp_out, sizeof(float) * E.ntheta);
    if (grad && E.gemm_auto) {                           // "gemm" = "auto": this evaluation's gradient health decides the NEXT evaluation's GEMM mode
        std::vector<float> gs((size_t)E.ntheta + K);
        std::memcpy(gs.data(), E.hp_out, sizeof(float) * E.ntheta);
        for (int k = 0; k < K; ++k) gs[(size_t)E.ntheta + k] = (float)E.hp_raw[k];
        measure_grad_health(E, gs.data(), term_w);
        if (gemm_auto_check(E, E.d_theta, term_w, ++E.eval_count)) return 1;
    }
    return 0;
}

int pinn_loss_grad_f64(pinn_handle h, const double* theta, int64_t p, const double* term_w, double* term_losses, double* grad) {
    if (!h || !theta) return fail("pinn_loss_grad_f64: null argument");
    const int K = (int)h->terms.size();
    if (h->f64) {                                        // float64 mode: native
        if (p != h->ntheta) return fail("theta length " + std::to_string(p) + " != ntheta " + std::to_string(h->ntheta));
        DeviceScope scope(h->device);
        return f64_eval(*h, theta, term_w, term_losses, grad);
    }
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

// l = sum_k logpdf(MvNormal(r_k, sigma_k^2 I), 0) = sum_k [ -N_k/2 log(2 pi) - N_k log sigma_k - SSE_k / (2 sigma_k^2) ]
// (src/training_strategies.jl:113-127; "SSE not MSE", src/discretize.jl:681): an affine function of the per-term sums of squares,
// so grad_theta l = - grad_theta sum_k w_k L_k with w_k = N_k / (2 sigma_k^2): ONE weighted evaluation.  sse[k]: raw sums of squares.
static void loglik_from_sse(const pinn_engine& E, const double* stds, const double* sse, double* loglik, double* grad_std) {
    const int K = (int)E.terms.size();
    double ll = 0.0;
    for (int k = 0; k < K; ++k) {
        // sharded point sets (n < n_norm): every quantity returned here is THIS SHARD'S additive part — the constants count the shard's
        // own points — so that loglik, grad_theta and grad_std summed over the ranks are the global values
        const double N = (double)E.terms[k].n, sd = stds[k];
        ll += -0.5 * N * std::log(2.0 * 3.14159265358979323846) - N * std::log(sd) - sse[k] / (2.0 * sd * sd);
        if (grad_std) grad_std[k] = -N / sd + sse[k] / (sd * sd * sd);
    }
    *loglik = ll;
}
// the float64-mode evaluation behind both pinn_loglik_grad entry points: grad_theta in double
static int loglik_f64(pinn_engine& E, const double* theta, const double* stds, double* loglik, double* grad_theta, double* grad_std) {
    const int K = (int)E.terms.size();
    std::vector<double> w(K), L(K), g(grad_theta ? (size_t)E.ntheta : 0);
    for (int k = 0; k < K; ++k) {
        if (!(stds[k] > 0.0)) return fail("pinn_loglik_grad: standard deviations must be positive");
        w[k] = (double)E.terms[k].n_norm / (2.0 * stds[k] * stds[k]);
    }
    if (f64_eval(E, theta, w.data(), L.data(), grad_theta ? g.data() : nullptr)) return 1;
    for (int k = 0; k < K; ++k) L[k] *= (double)E.terms[k].n_norm;           // mean -> SSE
    loglik_from_sse(E, stds, L.data(), loglik, grad_std);
    if (grad_theta) for (int64_t i = 0; i < E.ntheta; ++i) grad_theta[i] = -g[(size_t)i];
    return 0;
}

int pinn_loglik_grad(pinn_handle h, const float* theta, int64_t p, const double* stds, double* loglik, float* grad_theta, double* grad_std) {
    if (!h || !theta || !stds || !loglik) return fail("pinn_loglik_grad: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    const int K = (int)E.terms.size();
    if (E.f64) {                                         // float64 mode: converted at the boundary, evaluated in double
        if (p != E.ntheta) return fail("pinn_loglik_grad: theta length mismatch");
        std::vector<double> th(theta, theta + p), g(grad_theta ? (size_t)p : 0);
        if (loglik_f64(E, th.data(), stds, loglik, grad_theta ? g.data() : nullptr, grad_std)) return 1;
        if (grad_theta) for (int64_t i = 0; i < p; ++i) grad_theta[i] = (float)g[(size_t)i];
        return 0;
    }
    std::vector<float> w(K);
    for (int k = 0; k < K; ++k) {
        if (!(stds[k] > 0.0)) return fail("pinn_loglik_grad: standard deviations must be positive");
        w[k] = (float)((double)E.terms[k].n_norm / (2.0 * stds[k] * stds[k]));
    }
    if (upload_theta(E, theta, p)) return 1;
    if (run_loss_grad(E, E.d_theta, E.hp_out, w.data(), -1, false, E.hp_raw)) return 1;
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    loglik_from_sse(E, stds, E.hp_raw, loglik, grad_std);
    if (grad_theta)
        for (int64_t i = 0; i < E.ntheta; ++i) grad_theta[i] = -E.hp_out[i];
    return 0;
}

int pinn_loglik_grad_f64(pinn_handle h, const double* theta, int64_t p, const double* stds, double* loglik, double* grad_theta, double* grad_std) {
    if (!h || !theta || !stds || !loglik) return fail("pinn_loglik_grad_f64: null argument");
    pinn_engine& E = *h;
    if (p != E.ntheta) return fail("pinn_loglik_grad_f64: theta length mismatch");
    if (E.f64) {
        DeviceScope scope(E.device);
        return loglik_f64(E, theta, stds, loglik, grad_theta, grad_std);
    }
    std::vector<float> th((size_t)p), g(grad_theta ? (size_t)p : 0);
    for (int64_t i = 0; i < p; ++i) th[(size_t)i] = (float)theta[i];
    if (pinn_loglik_grad(h, th.data(), p, stds, loglik, grad_theta ? g.data() : nullptr, grad_std)) return 1;
    if (grad_theta) for (int64_t i = 0; i < p; ++i) grad_theta[i] = (double)g[(size_t)i];
    return 0;
}

// float64 mode: K evaluations with one-hot weights (the double kernels evaluate whole problems; per-term statistics are not their hot path);
// row k of term_grads (K x P doubles) = d term_losses[k] / d theta
static int term_grads_f64(pinn_engine& E, const double* theta, double* term_losses, double* term_grads) {
    const int K = (int)E.terms.size();
    const int64_t p = E.ntheta;
    std::vector<double> w(K), L(K);
    for (int k = 0; k < K; ++k) {
        for (int j = 0; j < K; ++j) w[j] = (j == k) ? 1.0 : 0.0;
        if (f64_eval(E, theta, w.data(), L.data(), term_grads + (size_t)k * p)) return 1;
        if (term_losses) term_losses[k] = L[k];
    }
    return 0;
}

int pinn_term_grads(pinn_handle h, const float* theta, int64_t p, double* term_losses, float* term_grads) {
    if (!h || !theta || !term_grads) return fail("pinn_term_grads: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    const int K = (int)E.terms.size();
    if (E.f64) {                                         // float64 mode: converted at the boundary
        if (p != E.ntheta) return fail("pinn_term_grads: theta length mismatch");
        std::vector<double> th(theta, theta + p), g((size_t)K * p);
        if (term_grads_f64(E, th.data(), term_losses, g.data())) return 1;
        for (size_t i = 0; i < g.size(); ++i) term_grads[i] = (float)g[i];
        return 0;
    }
    if (upload_theta(E, theta, p)) return 1;
    for (int k = 0; k < K; ++k) {
        if (run_loss_grad(E, E.d_theta, E.hp_out, nullptr, k, false, E.hp_raw)) return 1;
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        std::memcpy(term_grads + (size_t)k * p, E.hp_out, sizeof(float) * p);
        if (term_losses) term_losses[k] = E.hp_raw[k] / (double)E.terms[k].n_norm;
    }
    return 0;
}

int pinn_term_grads_f64(pinn_handle h, const double* theta, int64_t p, double* term_losses, double* term_grads) {
    if (!h || !theta || !term_grads) return fail("pinn_term_grads_f64: null argument");
    pinn_engine& E = *h;
    if (p != E.ntheta) return fail("pinn_term_grads_f64: theta length mismatch");
    const int K = (int)E.terms.size();
    if (E.f64) {
        DeviceScope scope(E.device);
        return term_grads_f64(E, theta, term_losses, term_grads);
    }
    std::vector<float> th((size_t)p), g((size_t)K * p);
    for (int64_t i = 0; i < p; ++i) th[(size_t)i] = (float)theta[i];
    if (pinn_term_grads(h, th.data(), p, term_losses, g.data())) return 1;
    for (size_t i = 0; i < g.size(); ++i) term_grads[i] = (double)g[i];
    return 0;
}

int pinn_loss_grad_device(pinn_handle h, const float* d_theta, const float* term_w, float* d_out, void* stream) {
    if (!h || !d_theta || !d_out) return fail("pinn_loss_grad_device: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    const int K = (int)E.terms.size();
    plat_stream user = (plat_stream)stream;
    plat_stream saved = E.stream;
    E.stream = user;     // NULL is the (legacy) default stream — e.g. torch's current stream
    // float64 mode (r05): the double kernels between a device-side widening of theta and narrowing of [gradient | sums]
    int rc = E.f64 ? f64_eval_from_device_f32(E, d_theta, term_w, d_out, true) : run_loss_grad(E, d_theta, d_out, term_w, -1, true);
    E.stream = saved;
    if (rc && g_err.empty()) return fail("pinn_loss_grad_device failed");
    E.timing_valid = (rc == 0) && E.timing_level >= 2;
    return rc;
}

int pinn_loss_grad_device_f64(pinn_handle h, const double* d_theta, const float* term_w, double* d_out, void* stream) {
    if (!h || !d_theta || !d_out) return fail("pinn_loss_grad_device_f64: null argument");
    pinn_engine& E = *h;
    if (!E.f64) return fail("pinn_loss_grad_device_f64: the handle is not in the float64 evaluation mode (pinn_set_option(h, \"precision\", \"f64\"))");
    DeviceScope scope(E.device);
    plat_stream saved = E.stream;
    E.stream = (plat_stream)stream;
    const int rc = f64_eval_from_device_f64(E, d_theta, term_w, d_out);
    E.stream = saved;
    if (rc && g_err.empty()) return fail("pinn_loss_grad_device_f64 failed");
    return rc;
}

int pinn_loss_device(pinn_handle h, const float* d_theta, float* d_sums, void* stream) {
    if (!h || !d_theta || !d_sums) return fail("pinn_loss_device: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    plat_stream saved = E.stream;
    E.stream = (plat_stream)stream;
    int rc = E.f64 ? f64_eval_from_device_f32(E, d_theta, nullptr, d_sums, false)
                   : run_loss_grad(E, d_theta, nullptr, nullptr, -1, false, nullptr, false, true, d_sums);      // loss-only: no gradient vector at all
    E.stream = saved;
    if (rc && g_err.empty()) return fail("pinn_loss_device failed");
    return rc;
}

int pinn_group_launched_by(pinn_handle h, int group) {
    if (!h || group < 0 || group >= (int)h->groups.size()) return -1;
    return h->groups[group].launched_by;
}

int pinn_residual(pinn_handle h, int term, const float* theta, int64_t p, float* r) {
    if (!h || !theta || !r) return fail("pinn_residual: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_residual: term index out of range");
    Term& T = E.terms[term];
    if (!T.d_pts) return fail("pinn_residual: term has no points");
    if (E.f64) {                                         // float64 mode: converted at the boundary, evaluated in double
        if (p != E.ntheta) return fail("pinn_residual: theta length mismatch");
        std::vector<double> th(theta, theta + p), rd((size_t)T.n);
        if (f64_residual(E, term, th.data(), rd.data())) return 1;
        for (int64_t i = 0; i < T.n; ++i) r[i] = (float)rd[(size_t)i];
        return 0;
    }
    if (upload_theta(E, theta, p)) return 1;
    pack_all(E);
    jit_take_launch_error();                             // (a stale message of an earlier call must not fail this one)
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
            G.spec->launch(ga, pk::MODE_FWD, std::max(1, std::min(E.ncu * G.spec->WG_FWD, G.spec->family == 1 ? (ga.ntiles + 3) / 4 : ga.ntiles)), E.stream);
        }
        aux::launch_expr(expr_args(E, Cp, 0.f, T.d_resid), Cp.blocks, E.stream);
        if (plat_d2h(r, T.d_resid, sizeof(float) * T.n, E.stream)) return fail("D2H copy failed");
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        { const std::string le = jit_take_launch_error(); if (!le.empty()) return fail("pinn_residual: " + le); }
        return 0;
    }
    Group& G = E.groups[T.group];
    pk::GroupArgs ga = G.ga;
    ga.nterms = 1;
    ga.terms[0] = G.ga.terms[T.slot_in_group];
    ga.terms[0].tile0 = 0;
    ga.terms[0].out = T.d_resid;
    ga.ntiles = ga.terms[0].ntiles;
    const int blocks = std::max(1, std::min(E.ncu * G.spec->WG_FWD, G.spec->family == 1 ? (ga.ntiles + 3) / 4 : ga.ntiles));
    G.spec->launch(ga, pk::MODE_RESID, blocks, E.stream);
    if (plat_d2h(r, T.d_resid, sizeof(float) * T.n, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    // a run-time specialised kernel whose (mode, activation) member could not be compiled at this, its first, launch leaves r unwritten
    { const std::string le = jit_take_launch_error(); if (!le.empty()) return fail("pinn_residual: " + le); }
    return 0;
}

int pinn_residual_f64(pinn_handle h, int term, const double* theta, int64_t p, double* r) {
    if (!h || !theta || !r) return fail("pinn_residual_f64: null argument");
    pinn_engine& E = *h;
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_residual_f64: term index out of range");
    if (p != E.ntheta) return fail("pinn_residual_f64: theta length mismatch");
    if (E.f64) {
        DeviceScope scope(E.device);
        return f64_residual(E, term, theta, r);
    }
    const int64_t n = E.terms[term].n;
    std::vector<float> th((size_t)p), rf((size_t)std::max<int64_t>(n, 1));
    for (int64_t i = 0; i < p; ++i) th[(size_t)i] = (float)theta[i];
    if (pinn_residual(h, term, th.data(), p, rf.data())) return 1;
    for (int64_t i = 0; i < n; ++i) r[i] = (double)rf[(size_t)i];
    return 0;
}

// forward-only launch of one network on caller-supplied points with kernel `sp`: the jet channels land in E.d_phi_out as [C][n]
static int forward_jets(pinn_engine& E, int net, const pk::SpecInfo* sp, const float* theta, int64_t p, const float* pts, int64_t n) {
    const Net& N = E.nets[net];
    if (!E.netplans[net].spec) return fail("network is not used by any term");
    if (sp->family != E.netplans[net].spec->family || sp->PACKED != E.netplans[net].spec->PACKED)
        return fail("internal: forward kernel and the network's packed weight image disagree");
    if (upload_theta(E, theta, p)) return 1;
    pack_all(E);
    jit_take_launch_error();                             // (clear: the check behind the launch below reports THIS launch)
    if (E.phi_cap < n || E.phi_chan < sp->C) {
        plat_sync(E.stream);
        plat_free(E.d_phi_pts); plat_free(E.d_phi_out);
        E.phi_cap = std::max(E.phi_cap, n);
        E.phi_chan = std::max(E.phi_chan, sp->C);
        E.d_phi_pts = (float*)plat_malloc(sizeof(float) * E.phi_cap * 8);
        E.d_phi_out = (float*)plat_malloc(sizeof(float) * E.phi_cap * E.phi_chan);
        if (!E.d_phi_pts || !E.d_phi_out) { E.phi_cap = 0; E.phi_chan = 0; return fail("device allocation failed (phi)"); }
    }
    std::vector<float> feats;                                  // periodic embedding: [arguments] -> [features] on the host (this entry point takes host points)
    if (!N.emb_idx.empty()) {
        const int nin = N.n_inputs(), ne = (int)N.emb_idx.size(), F = N.sizes[0];
        feats.resize((size_t)n * F);
        for (int64_t q = 0; q < n; ++q) {
            int pass = 0;
            for (int a = 0; a < nin; ++a) {
                const auto it = std::find(N.emb_idx.begin(), N.emb_idx.end(), a);
                const double x = pts[q * nin + a];
                if (it == N.emb_idx.end()) { feats[q * F + pass++] = (float)x; continue; }
                const int k = (int)(it - N.emb_idx.begin());
                const double ph = 6.283185307179586476925286766559 / N.emb_period[k] * x;
                feats[q * F + nin - ne + k] = (float)std::sin(ph);
                feats[q * F + nin + k] = (float)std::cos(ph);
            }
        }
        plat_h2d(E.d_phi_pts, feats.data(), sizeof(float) * n * F, E.stream);
        plat_sync(E.stream);                                   // (feats is a pageable temporary)
    } else
    plat_h2d(E.d_phi_pts, pts, sizeof(float) * n * N.sizes[0], E.stream);
    pk::GroupArgs ga;
    std::memset(&ga, 0, sizeof ga);
    ga.packed = E.netplans[net].cur;
    ga.params = E.d_params;
    ga.nterms = 1;
    ga.nterms_total = 1;
    ga.act = N.act;
    ga.act_layers = N.act_layers;
    ga.terms[0].pts = E.d_phi_pts;
    ga.terms[0].dt = N.sizes[0];
    for (int i = 0; i < 4; ++i) ga.terms[0].imap[i] = i;
    ga.terms[0].N = (int)n;
    ga.terms[0].tile0 = 0;
    ga.terms[0].ntiles = (int)((n + sp->TP - 1) / sp->TP);
    ga.terms[0].out = E.d_phi_out;
    ga.ntiles = ga.terms[0].ntiles;
    if (sp->family == 3) {               // DGM: unpacked weights + point-major scratch rows for these points
        const size_t need = (size_t)sp->dgm_rows * (size_t)ga.ntiles * 64;
        if (need > E.phi_scr_cap) {
            plat_sync(E.stream);
            plat_free(E.d_phi_scr);
            E.d_phi_scr = (float*)plat_malloc(sizeof(float) * need);
            E.phi_scr_cap = E.d_phi_scr ? need : 0;
            if (!E.d_phi_scr) return fail("device allocation failed (phi scratch)");
        }
        ga.scratch = E.d_phi_scr;
        ga.dgm_modes = N.sizes[1];
        ga.dgm_npad = ga.ntiles * 64;
        ga.dgm_nparams = N.nparams();
    }
    const int blocks = std::max(1, std::min(E.ncu * sp->WG_FWD, sp->family == 1 ? (ga.ntiles + 3) / 4 : ga.ntiles));
    sp->launch(ga, pk::MODE_FWD, blocks, E.stream);
    { const std::string le = jit_take_launch_error(); if (!le.empty()) return fail("forward kernel: " + le); }
    return 0;
}

int pinn_phi(pinn_handle h, int net, const float* theta, int64_t p, const float* pts, int64_t n, float* out) {
    if (!h || !theta || !pts || !out) return fail("pinn_phi: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    GemmScope gs(E.gemm);
    if (net < 0 || net >= (int)E.nets.size()) return fail("pinn_phi: net index out of range");
    if (n <= 0) return fail("pinn_phi: n must be positive");
    const Net& N = E.nets[net];
    if (E.f64) {                                         // float64 mode: converted at the boundary, evaluated in double
        if (p != E.ntheta) return fail("pinn_phi: theta length mismatch");
        const size_t np_ = (size_t)n * N.sizes[0];
        std::vector<double> th(theta, theta + p), xd(pts, pts + np_), od((size_t)n);
        if (f64_net_eval(E, net, th.data(), xd.data(), n, 0, nullptr, od.data())) return 1;
        for (int64_t i = 0; i < n; ++i) out[i] = (float)od[(size_t)i];
        return 0;
    }
    if (!E.netplans[net].spec) return fail("pinn_phi: network is not used by any term");
    const int LH = (int)N.sizes.size() - 2;
    (void)LH;
    const pk::SpecInfo* sp = ensure_spec(N, 0, {}, 0u, E.netplans[net].spec->family);
    if (!sp) return fail("pinn_phi: no value-only kernel for this network shape (" + g_err + ")");
    if (forward_jets(E, net, sp, theta, p, pts, n)) return 1;
    if (plat_d2h(out, E.d_phi_out, sizeof(float) * n, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_derivative(pinn_handle h, int net, const float* theta, int64_t p, const float* pts, int64_t n, int order, const int* axes, float* out) {
    if (!h || !theta || !pts || !out) return fail("pinn_derivative: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    GemmScope gs(E.gemm);
    if (net < 0 || net >= (int)E.nets.size()) return fail("pinn_derivative: net index out of range");
    if (n <= 0) return fail("pinn_derivative: n must be positive");
    if (order < 0 || order > MAX_DERIV_ORDER || (order > 0 && !axes)) return fail("pinn_derivative: order must be 0..6 (with `order` axes)");
    const Net& N = E.nets[net];
    if (E.f64) {                                         // float64 mode: converted at the boundary, evaluated in double
        if (p != E.ntheta) return fail("pinn_derivative: theta length mismatch");
        const size_t np_ = (size_t)n * N.sizes[0];
        std::vector<double> th(theta, theta + p), xd(pts, pts + np_), od((size_t)n);
        if (f64_net_eval(E, net, th.data(), xd.data(), n, order, axes, od.data())) return 1;
        for (int64_t i = 0; i < n; ++i) out[i] = (float)od[(size_t)i];
        return 0;
    }
    if (!E.netplans[net].spec) return fail("pinn_derivative: network is not used by any term");
    if (!N.emb_idx.empty()) return fail("pinn_derivative: not available for a network behind a periodic input embedding (use pinn_residual on a term that carries the derivative)");
    Slot sl;
    sl.net = net; sl.order = order; sl.lap = 0;
    for (int a = 0; a < MAX_DERIV_ORDER; ++a) sl.axes[a] = a < order ? axes[a] : 0;
    std::sort(sl.axes, sl.axes + order);
    for (int a = 0; a < order; ++a)
        if (sl.axes[a] < 0 || sl.axes[a] >= N.sizes[0]) return fail("pinn_derivative: axis out of range");
    unsigned need_first = 0, need_hi = 0;
    std::vector<std::pair<int, int>> need_pairs;
    if (slot_is_general(sl)) need_hi = GEN_FLAG | (unsigned)gen_set_id({slot_mi(sl)});      // mixed of order >= 3, orders 5-6: generated jet set
    else {
        for (int a = 0; a < order; ++a) need_first |= 1u << sl.axes[a];
        if (order >= 2) need_pairs.push_back({sl.axes[0], sl.axes[1]});
        if (order >= 3) need_hi = (unsigned)order << (4 * sl.axes[0]);
    }
    const int LH = (int)N.sizes.size() - 2;
    (void)LH;
    const pk::SpecInfo* sp = ensure_spec(N, need_first, need_pairs, need_hi, E.netplans[net].spec->family);
    if (!sp) return fail("pinn_derivative: no kernel carries this derivative for this network shape (" + g_err + ")");
    const int ch = chan_of(*sp, sl);
    if (ch < 0) return fail("internal: derivative has no channel");
    if (forward_jets(E, net, sp, theta, p, pts, n)) return 1;
    if (plat_d2h(out, E.d_phi_out + (size_t)ch * n, sizeof(float) * n, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

// phi(x, theta) and numeric_derivative in double (r06): the reference's closures compute in eltype(theta) = Float64 by default
// (src/pinn_types.jl:88-90, 445-482; src/discretize.jl:432-449).  Native on a handle in float64 mode; on an fp32 handle they narrow at the boundary.
int pinn_phi_f64(pinn_handle h, int net, const double* theta, int64_t p, const double* pts, int64_t n, double* out) {
    return pinn_derivative_f64(h, net, theta, p, pts, n, 0, nullptr, out);
}
int pinn_derivative_f64(pinn_handle h, int net, const double* theta, int64_t p, const double* pts, int64_t n, int order, const int* axes, double* out) {
    if (!h || !theta || !pts || !out) return fail("pinn_derivative_f64: null argument");
    pinn_engine& E = *h;
    if (net < 0 || net >= (int)E.nets.size()) return fail("pinn_derivative_f64: net index out of range");
    if (n <= 0) return fail("pinn_derivative_f64: n must be positive");
    if (p != E.ntheta) return fail("pinn_derivative_f64: theta length mismatch");
    if (order < 0 || order > MAX_DERIV_ORDER || (order > 0 && !axes)) return fail("pinn_derivative_f64: order must be 0..6 (with `order` axes)");
    if (E.f64) {
        DeviceScope scope(E.device);
        return f64_net_eval(E, net, theta, pts, n, order, axes, out);
    }
    const size_t np_ = (size_t)n * E.nets[net].n_inputs();
    std::vector<float> th((size_t)p), xf(np_), of((size_t)n);
    for (int64_t i = 0; i < p; ++i) th[(size_t)i] = (float)theta[i];
    for (size_t i = 0; i < np_; ++i) xf[i] = (float)pts[i];
    const int rc = order == 0 ? pinn_phi(h, net, th.data(), p, xf.data(), n, of.data()) : pinn_derivative(h, net, th.data(), p, xf.data(), n, order, axes, of.data());
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) out[i] = (double)of[(size_t)i];
    return 0;
}

int pinn_set_sampler(pinn_handle h, int term, int kind, const float* lb, const float* ub, int64_t n, uint64_t seed) {
    if (!h) return fail("null handle");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_set_sampler: term index out of range");
    Term& T = E.terms[term];
    if (kind < 0 || kind > 3) return fail("pinn_set_sampler: kind must be 0 (fixed set), 1 (uniform), 2 (Latin hypercube) or 3 (Sobol)");
    if (kind == 3 && T.d_user > 8) return fail("pinn_set_sampler: the Sobol sampler covers up to 8 axes");
    if (kind == 0) { T.sampler = 0; return 0; }
    if (!lb || !ub || n <= 0) return fail("pinn_set_sampler: bounds and a positive point count are required");
    if (!T.d_lb) { T.d_lb = (float*)plat_malloc(sizeof(float) * 8); T.d_ub = (float*)plat_malloc(sizeof(float) * 8); }
    if (!T.d_lb || !T.d_ub) return fail("device allocation failed (sampler)");
    plat_h2d(T.d_lb, lb, sizeof(float) * T.d_user, E.stream);
    plat_h2d(T.d_ub, ub, sizeof(float) * T.d_user, E.stream);
    T.seed = (unsigned)(seed ^ (seed >> 32)) + 0x9E3779B9U * (unsigned)(term + 1);
    if (kind == 3 && seed == 0) T.seed = 0;              // un-randomised Sobol: the same design on every draw
    T.draws = 0;
    // allocate / size the term's point buffer through the normal path with a first draw
    std::vector<float> tmp((size_t)n * T.d_user, 0.f);
    if (set_points_impl(h, term, tmp.data(), n, 0, false)) return 1;
    T.sampler = kind;                                    // only now: every check and allocation above has succeeded
    aux::launch_sample(kind, user_pts(T), (int)(n * T.d_user), T.d_user, T.d_lb, T.d_ub, T.seed, T.draws++, E.stream);
    embed_points(E, T);
    eval_sources(E, T);
    if (E.f64 && f64_points_from_device(E, term)) return 1;       // float64 mode: the double copy follows every draw (r05)
    plat_sync(E.stream);
    return 0;
}

int pinn_set_point_data(pinn_handle h, int term, const float* data, int ndata, int64_t n) {
    if (!h || !data) return fail("pinn_set_point_data: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
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
    return f64_set_point_data(E, term, nullptr);         // (float64 mode: the double rows follow)
}
int pinn_set_point_data_f64(pinn_handle h, int term, const double* data, int ndata, int64_t n) {
    if (!h || !data || ndata <= 0 || n <= 0) return fail("pinn_set_point_data_f64: null argument / empty data");
    std::vector<float> d32((size_t)ndata * (size_t)n);
    for (size_t i = 0; i < d32.size(); ++i) d32[i] = (float)data[i];
    if (pinn_set_point_data(h, term, d32.data(), ndata, n)) return 1;           // the fp32 kernels' rows (and every check)
    if (!h->f64) return 0;
    DeviceScope scope(h->device);
    return f64_set_point_data(*h, term, data);                                  // the float64 mode reads the observations as given
}

int pinn_set_point_weights(pinn_handle h, int term, const float* w, int64_t n) {
    if (!h) return fail("pinn_set_point_weights: null handle");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
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
    DeviceScope scope(E.device);
    if (term < 0 || term >= (int)E.terms.size()) return fail("pinn_get_points: term index out of range");
    Term& T = E.terms[term];
    if (!T.d_pts || n != T.n) return fail("pinn_get_points: the term holds " + std::to_string(T.n) + " points");
    if (plat_d2h(pts, user_pts(T), sizeof(float) * (size_t)n * T.d_user, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

// switch the GEMM arithmetic of a live handle: the kernel plan is rebuilt for the other mode (packed weight images, gradient-slab maps and
// reduction tables differ), the installed point sets, samplers, per-point data / weights and the optimiser state stay
static int replan_gemm(pinn_engine& E, int mode) {
    if (mode == E.gemm) return 0;
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    const int prev = E.gemm;
    free_plan(E);
    E.gemm = mode;
    int rc = build_plan(E);
    if (rc) {                                            // e.g. the other mode's kernel of a run-time specialised shape failed to compile: go back
        const std::string why = g_err;
        free_plan(E);
        E.gemm = prev;
        if (build_plan(E)) return fail("pinn_set_option: the kernel plan could not be rebuilt (" + why + "); the handle is unusable");
        g_err = why;
    }
    for (size_t t = 0; t < E.terms.size(); ++t)
        if (E.terms[t].d_pts && E.terms[t].n > 0 && term_installed(E, (int)t)) return 1;
    for (size_t t = 0; t < E.terms.size(); ++t) {        // per-point data feeds the source channels: re-evaluate where installed
        Term& T = E.terms[t];
        if (T.ndata > 0 && T.data_n == T.n && T.n > 0) eval_sources(E, T);
    }
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return rc;
}

// ---- "gemm" = "auto": WHEN to leave the split-bf16 products ----
// The rule is a MEASUREMENT, not a heuristic: at the current parameters the gradient is evaluated with both GEMM arithmetics of the handle
// (the split-bf16 products and their fp32-MFMA twin: an fmaf chain per product) and compared,
//     delta = |grad(split) - grad(fp32)|_2 / |grad(fp32)|_2 ,
// which IS the split products' arithmetic error at this iterate up to the twin's own (2 - 7 x smaller, profiles/r05_theta_variants_ab.txt
// section F).  "split" is kept while delta <= 1e-5 (the north star's tolerance), "fp32" runs above; an fp32 handle goes back when delta has
// fallen under 3e-6.  At initialisation delta ~ 2e-7; on the trained fixtures (cfg2 after 2,000 / 6,000 Adam steps) 1e-5 / 4e-3: the policy leaves
// the fast products exactly where parity starts to depend on them.  Cost: two evaluations and at most two re-plans (milliseconds) per check; checks
// run at the end of a pinn_adam_steps call (loop path) and inside pinn_loss_grad with a gradient, at most once per GEMM_CHECK_EVERY optimiser
// steps / evaluations, never inside a resident loop.  Also reported: rho = |grad|_2 / sqrt(L) ("grad_health"): |grad| <= 2 sqrt(L) J_rms, so rho
// against its value at initialisation is the cancellation factor of the gradient sum — when it has fallen by ~1e3 no fp32 arithmetic holds 1e-3
// any more and "precision" = "f64" is the remedy (the glue's precision policy puts Float64 parameters there from the start).
constexpr double GEMM_DELTA_UP = 1e-5, GEMM_DELTA_DOWN = 3e-6;
constexpr long long GEMM_CHECK_EVERY = 1000;
static void measure_grad_health(pinn_engine& E, const float* grad_and_sums, const float* term_w) {
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    double g2 = 0.0, L = 0.0;
    for (int64_t i = 0; i < P; ++i) g2 += (double)grad_and_sums[i] * (double)grad_and_sums[i];
    for (int k = 0; k < K; ++k) L += (term_w ? (double)term_w[k] : 1.0) * (double)grad_and_sums[P + k] / (double)E.terms[k].n_norm;
    E.grad_health = L > 0.0 ? std::sqrt(g2) / std::sqrt(L) : -1.0;
}
static int replan_gemm(pinn_engine& E, int mode);
// the policy's check at the device-resident parameters d_theta (clock: `now` optimiser steps / evaluations)
static int gemm_auto_check(pinn_engine& E, const float* d_theta, const float* term_w, long long now) {
    if (!E.gemm_auto || E.f64 || E.comm) return 0;
    if (E.gemm_check_t >= 0 && now - E.gemm_check_t < GEMM_CHECK_EVERY && now >= E.gemm_check_t) return 0;
    bool has_twin = false;
    for (const NetPlan& NP : E.netplans) has_twin = has_twin || (NP.spec && NP.spec->family == 2 && NP.spec->twin);
    if (!has_twin) return 0;
    const int64_t P = E.ntheta;
    const int start = E.gemm;
    std::vector<float> ga((size_t)P), gb((size_t)P);
    if (run_loss_grad(E, d_theta, E.d_out, term_w, -1, false)) return 1;
    if (plat_d2h(ga.data(), E.d_out, sizeof(float) * P, E.stream) || plat_sync(E.stream)) return fail("D2H copy failed");
    if (replan_gemm(E, start == pk::GEMM_SPLIT ? pk::GEMM_FP32 : pk::GEMM_SPLIT)) return 1;
    if (E.gemm == start) return 0;                       // (the twin could not be built: the handle stays where it was)
    if (run_loss_grad(E, d_theta, E.d_out, term_w, -1, false)) return 1;
    if (plat_d2h(gb.data(), E.d_out, sizeof(float) * P, E.stream) || plat_sync(E.stream)) return fail("D2H copy failed");
    const std::vector<float>& gs = start == pk::GEMM_SPLIT ? ga : gb;      // split / fp32 gradients
    const std::vector<float>& gf = start == pk::GEMM_SPLIT ? gb : ga;
    double d2 = 0.0, n2 = 0.0;
    for (int64_t i = 0; i < P; ++i) { const double d = (double)gs[(size_t)i] - (double)gf[(size_t)i]; d2 += d * d; n2 += (double)gf[(size_t)i] * (double)gf[(size_t)i]; }
    E.gemm_delta = n2 > 0.0 ? std::sqrt(d2 / n2) : 0.0;
    E.gemm_check_t = now;
    const int want = (start == pk::GEMM_SPLIT) ? (E.gemm_delta > GEMM_DELTA_UP ? pk::GEMM_FP32 : pk::GEMM_SPLIT)
                                               : (E.gemm_delta < GEMM_DELTA_DOWN ? pk::GEMM_SPLIT : pk::GEMM_FP32);
    return replan_gemm(E, want);
}

int pinn_set_option(pinn_handle h, const char* name, const char* value) {
    if (!h || !name || !value) return fail("pinn_set_option: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    const std::string k = name, v = value;
    if (k == "gemm") {
        if (v == "split") { E.gemm_auto = false; return replan_gemm(E, pk::GEMM_SPLIT); }
        if (v == "fp32") { E.gemm_auto = false; return replan_gemm(E, pk::GEMM_FP32); }
        if (v == "auto") { E.gemm_auto = true; E.gemm_check_t = -1; return 0; }
        return fail("pinn_set_option: gemm must be \"split\", \"fp32\" or \"auto\"");
    }
    if (k == "precision") {
        if (v == "f64") return f64_enable(E);
        if (v == "f32") { plat_sync(E.stream); f64_destroy(E); return 0; }
        return fail("pinn_set_option: precision must be \"f32\" or \"f64\"");
    }
    if (k == "persistent") {
        if (v == "on" || v == "1") { E.persistent = true; return 0; }
        if (v == "off" || v == "0") { E.persistent = false; return 0; }
        return fail("pinn_set_option: persistent must be \"on\" or \"off\"");
    }
    if (k == "derivative") {
        if (v == "stencil") return f64_stencil_enable(E, true);
        if (v == "exact") return E.f64 ? f64_stencil_enable(E, false) : 0;
        return fail("pinn_set_option: derivative must be \"exact\" or \"stencil\"");
    }
    return fail("pinn_set_option: unknown option \"" + k + "\" (known: gemm, precision, persistent, derivative)");
}

int pinn_get_option(pinn_handle h, const char* name, char* buf, int64_t buflen) {
    if (!h || !name || !buf || buflen <= 0) return fail("pinn_get_option: bad argument");
    const std::string k = name;
    if (k == "gemm") { std::snprintf(buf, (size_t)buflen, h->gemm_auto ? "auto(%s)" : "%s", h->gemm == pk::GEMM_FP32 ? "fp32" : "split"); return 0; }
    if (k == "grad_health") { std::snprintf(buf, (size_t)buflen, "%.9g", h->grad_health); return 0; }
    if (k == "gemm_delta") { std::snprintf(buf, (size_t)buflen, "%.9g", h->gemm_delta); return 0; }
    if (k == "precision") { std::snprintf(buf, (size_t)buflen, "%s", h->f64 ? "f64" : "f32"); return 0; }
    if (k == "persistent") { std::snprintf(buf, (size_t)buflen, "%s", h->persistent ? "on" : "off"); return 0; }
    if (k == "derivative") { std::snprintf(buf, (size_t)buflen, "%s", pe::f64_stencil_on(*h) ? "stencil" : "exact"); return 0; }
    if (k == "eval_path") { std::snprintf(buf, (size_t)buflen, "%s", h->eval_path == 2 ? "one launch" : (h->eval_path == 1 ? "stand-alone kernels" : "none")); return 0; }
    if (k == "f64_path") { std::snprintf(buf, (size_t)buflen, "%s", pe::f64_path(*h)); return 0; }
    if (k == "f64_affine") { std::snprintf(buf, (size_t)buflen, "%d", pe::f64_affine_terms(*h)); return 0; }       // terms on the affine-residual fast path
    if (k == "f64_merged") { std::snprintf(buf, (size_t)buflen, "%d", pe::f64_merged(*h)); return 0; }          // merged launches of the last float64 evaluation
    if (k == "adam_path") { std::snprintf(buf, (size_t)buflen, "%s", h->adam_path == 2 ? "persistent" : (h->adam_path == 1 ? "loop" : "none")); return 0; }
    return fail("pinn_get_option: unknown option \"" + k + "\" (known: gemm, precision, persistent, derivative, grad_health, gemm_delta, adam_path, eval_path, f64_path, f64_merged, f64_affine)");
}

int pinn_adam_init(pinn_handle h, const float* theta, int64_t p) {
    if (!h || !theta) return fail("pinn_adam_init: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (p != E.ntheta) return fail("pinn_adam_init: theta length mismatch");
    if (E.f64) {                                         // float64 mode: the optimiser state lives in double (f64.cpp)
        std::vector<double> th(theta, theta + p);
        return f64_adam_init(E, th.data());
    }
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

// pinn_adam_steps through a hipGraph (PINN_GRAPH=1; see the note there).  Returns 0 when all nsteps ran; -1 when the device-side step
// state could not be set up (NOTHING has run: the caller takes the plain loop); 1 when a step failed after the optimiser state may
// already have advanced (g_err holds the reason: the caller reports the failure and does NOT run the steps again).
static int adam_steps_graph(pinn_engine& E, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w) {
    const int K = (int)E.terms.size(), P = (int)E.ntheta;
    if (!E.d_step) {
        E.d_step = (int*)plat_malloc(sizeof(int));
        E.d_draws = (unsigned*)plat_malloc(sizeof(unsigned) * K);
        E.d_sampled = (int*)plat_malloc(sizeof(int) * K);
        if (!E.d_step || !E.d_draws || !E.d_sampled) return -1;
    }
    if (E.c12_cap < nsteps) {
        plat_sync(E.stream);
        plat_free(E.d_c12);
        E.d_c12 = (float*)plat_malloc(sizeof(float) * 2 * (size_t)nsteps);
        E.c12_cap = E.d_c12 ? nsteps : 0;
        if (!E.d_c12) return -1;
    }
    std::vector<float> c12(2 * (size_t)nsteps);
    for (int s = 0; s < nsteps; ++s) {
        c12[2 * s] = (float)(1.0 / (1.0 - std::pow((double)beta1, (double)(E.opt_t + s + 1))));
        c12[2 * s + 1] = (float)(1.0 / (1.0 - std::pow((double)beta2, (double)(E.opt_t + s + 1))));
    }
    std::vector<unsigned> draws(K);
    std::vector<int> sampled(K);
    const int zero = 0;
    for (int t = 0; t < K; ++t) { draws[t] = E.terms[t].draws; sampled[t] = E.terms[t].sampler != 0; }
    plat_h2d(E.d_c12, c12.data(), sizeof(float) * c12.size(), E.stream);
    plat_h2d(E.d_draws, draws.data(), sizeof(unsigned) * K, E.stream);
    plat_h2d(E.d_sampled, sampled.data(), sizeof(int) * K, E.stream);
    plat_h2d(E.d_step, &zero, sizeof(int), E.stream);
    if (plat_sync(E.stream)) return -1;                          // (the host vectors above are pageable)
    auto one_step = [&]() -> int {
        for (size_t t = 0; t < E.terms.size(); ++t) {
            Term& T = E.terms[t];
            if (T.sampler != 0) {
                aux::launch_sample_dev(T.sampler, user_pts(T), (int)(T.n * T.d_user), T.d_user, T.d_lb, T.d_ub, T.seed, E.d_draws + t, E.stream);
                embed_points(E, T);
                eval_sources(E, T);
            }
        }
        if (run_loss_grad(E, E.d_opt_theta, E.d_opt_out, term_w, -1, false)) return 1;
        aux::launch_total_loss_dev(E.d_hist, E.d_step, E.d_opt_out, P, K, E.d_w_over_n, E.stream);
        aux::launch_adam_dev(E.d_opt_theta, E.d_opt_m, E.d_opt_v, E.d_opt_out, P, lr, beta1, beta2, eps, E.d_c12, E.d_step, E.stream);
        aux::launch_advance(E.d_step, E.d_draws, E.d_sampled, K, E.stream);
        return 0;
    };
    int done = 0;
    if (one_step()) return 1;                                    // first step outside the capture: lazy allocations and module loads happen here
    done = 1;
    plat_graph graph;
    bool ok = plat_graph_capture_begin(E.stream);
    if (ok) {
        const int rc = one_step();
        ok = plat_graph_capture_end(E.stream, graph) && rc == 0;
    }
    for (; ok && done < nsteps; ++done)
        if (!plat_graph_launch(graph, E.stream)) ok = false;
    plat_graph_destroy(graph);
    for (; done < nsteps; ++done)                                // no graph (emulation, or the capture failed): same kernels, launched one by one
        if (one_step()) return 1;
    return 0;
}

// seed of a term's device sampler on this rank: ranks of a communicator draw different points (their shards of one global draw)
static unsigned sampler_seed(const pinn_engine& E, const Term& T) {
    return E.comm_size > 1 ? T.seed + 0x85EBCA6BU * (unsigned)(E.comm_rank + 1) : T.seed;
}

// one redraw of a sampled term's float point set (the float64 optimiser loop widens it afterwards: f64.cpp)
static void redraw_term_f32(pinn_engine& E, Term& T) {
    aux::launch_sample(T.sampler, user_pts(T), (int)(T.n * T.d_user), T.d_user, T.d_lb, T.d_ub, sampler_seed(E, T), T.draws++, E.stream);
    embed_points(E, T);                                  // (the float feature rows of an embedded term stay in step with the draw: a later switch back to fp32 reads them)
}

// Device-side table of a handle's redrawn terms (aux::ResampleTerm), draw counters as they stand now.  Returns the number of terms in the
// table (0: no sampler), -1 when a redrawn term carries what the one-launch redraw does not cover (embeddings, per-point data / weights,
// coupled equations: the per-term kernels stay), -2 on a device error (g_err set).  max_n = the largest set.
static int upload_resample_table(pinn_engine& E, int* max_n) {
    std::vector<pk::TrainSampler> samp;
    *max_n = 0;
    for (auto& T : E.terms) {
        if (T.sampler == 0) continue;
        if (!T.emb_cols.empty() || T.ndata > 0 || T.pw_n > 0 || T.coupled >= 0 || T.src_root.size() > (size_t)aux::SRC_MAX) return -1;
        pk::TrainSampler S;
        std::memset(&S, 0, sizeof S);
        S.pts = T.d_pts; S.n = (int)T.n; S.d = T.d; S.kind = T.sampler; S.lb = T.d_lb; S.ub = T.d_ub;
        S.seed = sampler_seed(E, T); S.draw0 = T.draws;
        S.has_src = T.src_root.empty() ? 0 : 1;
        if (S.has_src) {
            S.src.pts = T.d_pts; S.src.N = (int)T.n; S.src.d = T.d; S.src.prog = T.d_src_prog; S.src.nops = (int)T.src_prog.size();
            S.src.nsrc = (int)T.src_root.size();
            for (int j = 0; j < S.src.nsrc; ++j) S.src.root[j] = T.src_root[(size_t)j];
            S.src.out = T.d_src; S.src.data = nullptr;
        }
        *max_n = std::max(*max_n, S.n);
        samp.push_back(S);
    }
    if (samp.empty()) return 0;
    if (plat_sync(E.stream)) { fail(std::string("device error: ") + plat_last_error()); return -2; }      // (an earlier launch may still read the table)
    if ((int)samp.size() > E.train_samp_cap) {
        plat_free(E.d_train_samp);
        E.d_train_samp = plat_malloc(sizeof(pk::TrainSampler) * samp.size());
        E.train_samp_cap = E.d_train_samp ? (int)samp.size() : 0;
        if (!E.d_train_samp) { fail("device allocation failed (table of redrawn terms)"); return -2; }
    }
    plat_h2d(E.d_train_samp, samp.data(), sizeof(pk::TrainSampler) * samp.size(), E.stream);
    if (plat_sync(E.stream)) { fail(std::string("device error: ") + plat_last_error()); return -2; }      // (samp is a pageable temporary)
    return (int)samp.size();
}

// the resident loop over ndev handles: one handle (plain, or a rank of a one-process-per-GPU communicator), or the handles of one
// pinn_comm_init_all communicator (single process, several devices).  Per iteration and device: redraw the sampled sets -> evaluate the
// local shards; then ONE all-reduce of [gradient | sums] (+ the K double sums) over the communicator, each rank's call on its own stream;
// then the fused update (Adam + total loss + weight-image scatter) on every device.  No host synchronisation inside the loop.
static int adam_loop(pinn_engine** es, int ndev, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w) {
    const int K = (int)es[0]->terms.size(), P = (int)es[0]->ntheta;
    const bool collective = ndev > 1 || (es[0]->comm != nullptr && es[0]->comm_per_process);
    std::vector<float*> vec(ndev);
    std::vector<double*> raw(ndev);
    for (int i = 0; i < ndev; ++i) { vec[i] = es[i]->d_opt_out; raw[i] = es[i]->d_lossraw; }
    // redrawn point sets: ONE launch per step redraws every such term and its source channels (aux::k_resample over a device-side table);
    // terms with embeddings / per-point data keep their own sampler / embed / source launches (PINN_NO_FUSED_RESAMPLE=1: always)
    std::vector<int> nsamp(ndev, -1), max_n(ndev, 0);
    for (int i = 0; i < ndev && std::getenv("PINN_NO_FUSED_RESAMPLE") == nullptr; ++i) {
        DeviceScope scope(es[i]->device);
        nsamp[i] = upload_resample_table(*es[i], &max_n[i]);
        if (nsamp[i] == -2) return 1;
    }
    for (int s = 0; s < nsteps; ++s) {
        for (int i = 0; i < ndev; ++i) {
            pinn_engine& E = *es[i];
            DeviceScope scope(E.device);
            if (nsamp[i] > 0) {
                aux::launch_resample((const pk::TrainSampler*)E.d_train_samp, nsamp[i], max_n[i], s, E.stream);
                for (auto& T : E.terms) if (T.sampler != 0) ++T.draws;
            }
            for (size_t t = 0; t < E.terms.size() && nsamp[i] < 0; ++t) {            // resampling strategies: fresh points every evaluation, on device
                Term& T = E.terms[t];
                if (T.sampler != 0) {
                    aux::launch_sample(T.sampler, user_pts(T), (int)(T.n * T.d_user), T.d_user, T.d_lb, T.d_ub, sampler_seed(E, T), T.draws++, E.stream);
                    embed_points(E, T);
                    eval_sources(E, T);
                }
            }
            // from the second step on the packed weight images are already those of the current theta: the update kernel below wrote them
            const bool fused = E.inv_ok && E.d_inv_ptr && std::getenv("PINN_NO_FUSED_ADAM") == nullptr;
            if (run_loss_grad(E, E.d_opt_theta, E.d_opt_out, term_w, -1, false, nullptr, fused && s > 0)) return 1;
        }
        if (collective) {
            if (comm_all_reduce(es, ndev, vec.data(), raw.data())) return 1;
            for (int i = 0; i < ndev; ++i) {
                DeviceScope scope(es[i]->device);
                sums_from_double(es[i]->d_opt_out + P, es[i]->d_lossraw, K, es[i]->stream);      // the exact (double) sums replace the float-summed ones
            }
        }
        for (int i = 0; i < ndev; ++i) {
            pinn_engine& E = *es[i];
            DeviceScope scope(E.device);
            ++E.opt_t;
            const float c1 = (float)(1.0 / (1.0 - std::pow((double)beta1, (double)E.opt_t)));
            const float c2 = (float)(1.0 / (1.0 - std::pow((double)beta2, (double)E.opt_t)));
            // one update kernel per step: Adam + the evaluation's total loss + the new parameters scattered into the packed weight images
            // (PINN_NO_FUSED_ADAM=1: total_loss, adam and next step's pack as three launches)
            const bool fused = E.inv_ok && E.d_inv_ptr && std::getenv("PINN_NO_FUSED_ADAM") == nullptr;
            if (fused) {
                aux::AdamFusedArgs fa;
                std::memset(&fa, 0, sizeof fa);
                fa.theta = E.d_opt_theta; fa.m = E.d_opt_m; fa.v = E.d_opt_v; fa.out = E.d_opt_out; fa.P = P; fa.K = K;
                fa.lr = lr; fa.b1 = beta1; fa.b2 = beta2; fa.eps = eps; fa.c1 = c1; fa.c2 = c2;
                fa.inv_ptr = E.d_inv_ptr; fa.inv_pos = E.d_inv_pos;
                for (size_t n = 0; n < E.nets.size(); ++n) fa.packed[n] = E.netplans[n].d_packed;
                fa.hist = E.d_hist; fa.step = s; fa.w_over_n = E.d_w_over_n;
                aux::launch_adam_fused(fa, E.stream);
            } else {
                aux::launch_total_loss(E.d_hist, s, E.d_opt_out, P, K, E.d_w_over_n, E.stream);
                aux::launch_adam(E.d_opt_theta, E.d_opt_m, E.d_opt_v, E.d_opt_out, P, lr, beta1, beta2, eps, c1, c2, E.stream);
            }
        }
    }
    return 0;
}

// ---- persistent training kernel (pinn_train.hpp): the iterations of pinn_adam_steps inside ONE launch ----
// Eligible: ONE fused family-1 launch group of ONE network whose spec carries the kernel (tanh / sigmoid), fixed or plainly redrawn point sets, no estimated
// PDE parameters, no communicator, float32, every workgroup resident (one per CU) and few enough for the one-stage reduction whose
// association the kernel repeats (aux::reduce_is_small) — i.e. the small problems of the reference's own test-suite.  PINN_PERSISTENT=0
// (read per call) keeps the stand-alone loop; results are bit-identical either way (tests/test_train_kernel.py).
static bool train_eligible(pinn_engine& E) {
    const char* e = std::getenv("PINN_PERSISTENT");
    if (!E.persistent || (e && std::atoi(e) == 0)) return false;
    if (std::getenv("PINN_GRAPH")) return false;               // (the graph-replay experiment of the loop)
    if (E.comm || E.f64 || E.ne != 0 || E.groups.size() != 1 || !E.coupled.empty() || E.nets.size() != 1) return false;
    if (!E.inv_ok || !E.d_inv_ptr) return false;
    const Group& G = E.groups[0];
    if (G.kind != 0 || !G.spec || G.spec->family != 1 || !G.spec->train) return false;
    if (G.ga.act != pk::ACT_TANH && G.ga.act != pk::ACT_SIGMOID) return false;
    for (auto& T : E.terms)          // redrawn point sets: drawn inside the kernel unless they carry embeddings or per-point data / weights
        if (T.sampler != 0 && (!T.emb_cols.empty() || T.ndata > 0 || T.pw_n > 0 || T.coupled >= 0 || T.src_root.size() > (size_t)aux::SRC_MAX)) return false;
    const char* lim = std::getenv("PINN_REDUCE_DIRECT_MAX");
    const int direct_max = lim ? std::atoi(lim) : aux::REDUCE_DIRECT_MAX;
    if (G.blocks > direct_max || G.blocks > 32 || G.blocks > E.ncu) return false;
    // every thread of the launch owns at most one element of [theta | K sums] and keeps its maps and optimiser state in registers; a launch
    // with fewer threads than parameters (a handful of points under a comparatively large net) has nothing to win from the kernel
    const int K = (int)E.terms.size(), P = (int)E.ntheta;
    if (G.blocks * 256 < P + K || E.max_contrib > pk::TRAIN_MAX_CONTRIB || E.max_inv_pos > pk::TRAIN_MAX_POS) return std::getenv("PINN_TRAIN_GENERAL") != nullptr;
    return true;
}
// what a launch of the training kernel needs besides the optimiser: barrier words, the reduction inputs of the (single) launch group, the
// thread -> element map, the seeds of the reverse sweep.  Returns 0 / 1 (g_err set).
static int train_common(pinn_engine& E, pk::TrainArgs& ta, const float* term_w) {
    const int K = (int)E.terms.size(), P = (int)E.ntheta;
    Group& G = E.groups[0];
    if (!E.d_bar) {
        E.d_bar = (unsigned*)plat_malloc(sizeof(unsigned) * 16);      // [0] arrivals, [1] time-out flag, [4..11] phase ticks of a PINN_STAMP build
        E.d_sums2 = (float*)plat_malloc(sizeof(float) * 2 * (size_t)std::max(K, 1));
        E.h_flag = (unsigned*)plat_host_alloc(sizeof(unsigned) * 4);
        if (!E.d_bar || !E.d_sums2 || !E.h_flag) return fail("device allocation failed (grid barrier words)");
        E.h_flag[0] = 0;
        E.bar_arrivals = 0;
        plat_memset(E.d_bar, 0, sizeof(unsigned) * 16, E.stream);     // (once: the arrival counter runs on from launch to launch)
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    for (size_t j = 0; j < G.terms.size(); ++j) {
        const int ti = G.terms[j];
        G.ga.terms[j].scale = (float)(2.0 * (double)(term_w ? term_w[ti] : 1.0f) / (double)E.terms[ti].n_norm);
    }
    G.ga.slabs = G.d_slabs;
    G.ga.chain = 0;
    G.active = true;
    G.timed = false;
    G.launched_blocks = G.blocks;
    G.launched_by = 0;
    std::memset(&ta, 0, sizeof ta);
    ta.slabs = G.d_slabs; ta.losspart = G.d_losspart; ta.row_ptr = E.d_gr_ptr; ta.row_ent = E.d_gr_ent;
    ta.slab_floats = G.slab_floats; ta.nblocks = G.blocks;
    ta.P = P; ta.K = K;
    ta.sums2 = E.d_sums2;
    ta.bar = E.d_bar;
    ta.hflag = E.h_flag;
    ta.cached = (G.blocks * 256 >= P + K && E.max_contrib <= pk::TRAIN_MAX_CONTRIB && E.max_inv_pos <= pk::TRAIN_MAX_POS &&
                 std::getenv("PINN_TRAIN_NO_CACHE") == nullptr) ? 1 : 0;
    if (ta.cached && (!E.d_own_r || E.own_blocks != G.blocks)) {
        // thread -> element: theta element r sits at gid = its FIRST slab entry when those are distinct and leave room for the K sums at the
        // end of the grid (lanes of a wave then read consecutive slab entries: coalesced), else at gid = r
        const int nthr = G.blocks * 256;
        std::vector<int> own((size_t)nthr, -1);
        bool by_entry = true;
        for (size_t i = 0; i < G.row_theta.size() && by_entry; ++i) {
            const int e0 = G.row_ptr[i] < G.row_ptr[i + 1] ? G.row_off[G.row_ptr[i]] : -1;
            if (e0 < 0 || e0 >= nthr - K || own[(size_t)e0] >= 0) by_entry = false;
            else own[(size_t)e0] = G.row_theta[i];
        }
        int placed = 0;
        for (int v : own) placed += v >= 0;
        if (!by_entry || placed != P || std::getenv("PINN_TRAIN_NO_COALESCE")) {
            std::fill(own.begin(), own.end(), -1);
            for (int r = 0; r < P; ++r) own[(size_t)r] = r;
        } else {
            // 64-entry groups dealt round-robin over the workgroups (group g -> workgroup g % blocks, wave g / blocks): the slab reads of an
            // update spread over every CU of the launch instead of the first ceil(PW / 256)
            int last = 0;
            for (int e = 0; e < nthr; ++e) if (own[(size_t)e] >= 0) last = e;
            const int ngroups = last / 64 + 1;
            if ((ngroups + G.blocks - 1) / G.blocks <= 4) {
                std::vector<int> dealt((size_t)nthr, -1);
                for (int g = 0; g < ngroups; ++g)
                    for (int l = 0; l < 64; ++l) dealt[(size_t)((g % G.blocks) * 256 + (g / G.blocks) * 64 + l)] = own[(size_t)(g * 64 + l)];
                own.swap(dealt);
            }
        }
        for (int k = 0, gid = nthr - 1; k < K && gid >= 0; --gid)           // the K sums: threads without an element, from the end of the grid
            if (own[(size_t)gid] < 0) own[(size_t)gid] = P + k++;
        E.hist_gid = 0;
        for (int gid = nthr - 1; gid >= 0; --gid) if (own[(size_t)gid] < 0) { E.hist_gid = gid; break; }
        plat_sync(E.stream);
        plat_free(E.d_own_r);
        E.d_own_r = (int*)plat_malloc(sizeof(int) * (size_t)nthr);
        if (!E.d_own_r) return fail("device allocation failed (training kernel: thread map)");
        plat_h2d(E.d_own_r, own.data(), sizeof(int) * (size_t)nthr, E.stream);
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        E.own_blocks = G.blocks;
    }
    ta.own_r = E.d_own_r;
    ta.hist_gid = ta.cached ? E.hist_gid : 0;
    return 0;
}
// after a synchronisation: did a launch of the training kernel run into its barrier's time-out?  Then the barrier words start afresh and the
// handle keeps the stand-alone kernels from now on.
static bool train_timed_out(pinn_engine& E) {
    if (E.h_flag[0] == 0 && std::getenv("PINN_TRAIN_FORCE_TIMEOUT") == nullptr) return false;
    E.h_flag[0] = 0;
    E.bar_arrivals = 0;
    plat_memset(E.d_bar, 0, sizeof(unsigned) * 16, E.stream);
    plat_sync(E.stream);
    E.persistent = false;
    std::fprintf(stderr, "[pinn] the persistent kernel's grid barrier timed out (its workgroups were not all resident: is the device shared?); "
                         "this handle continues with the stand-alone kernels\n");
    return true;
}

// ONE evaluation of a small problem in one launch (pinn_train.hpp, TrainArgs::eval_only): the residual kernel, a grid barrier and the
// fixed-order sums of aux::reduce_direct_body, thread per slab entry — instead of the residual kernel + the reduction kernel.  Only for
// callers that synchronise right after (the host entry points): a time-out is detected there and the evaluation repeated by the
// stand-alone kernels.  Same numbers bit for bit.  Returns 0 done, 1 failure, 2 timed out (the caller takes the normal path).
static int eval_fused(pinn_engine& E, const float* d_theta, float* d_out, const float* term_w, double* lossraw) {
    Group& G = E.groups[0];
    pk::TrainArgs ta;
    if (train_common(E, ta, term_w)) return 1;
    pack_all(E, d_theta, false);
    ta.out = d_out; ta.lossraw = lossraw ? lossraw : E.d_lossraw;
    ta.eval_only = 1; ta.nsteps = 1;
    ta.arrivals0 = E.bar_arrivals;
    E.bar_arrivals += (unsigned)G.blocks;
    G.spec->train(G.ga, ta, G.blocks, E.stream);
    return 0;
}

// the same launch shape as the training kernel's, without the optimiser's needs (estimated PDE parameters are fine here); callers that
// asked for HIP events around the kernels (pinn_set_timing) keep the stand-alone kernels those events bracket.  PINN_NO_FUSED_EVAL=1: never.
static bool eval_eligible(pinn_engine& E) {
    const char* e = std::getenv("PINN_PERSISTENT");
    if (!E.persistent || (e && std::atoi(e) == 0) || std::getenv("PINN_NO_FUSED_EVAL")) return false;
    if (E.timing_level != 0 || E.comm || E.f64 || E.groups.size() != 1 || !E.coupled.empty() || E.nets.size() != 1) return false;
    const Group& G = E.groups[0];
    if (G.kind != 0 || !G.spec || G.spec->family != 1 || !G.spec->train) return false;
    if (G.ga.act != pk::ACT_TANH && G.ga.act != pk::ACT_SIGMOID) return false;
    const char* lim = std::getenv("PINN_REDUCE_DIRECT_MAX");
    const int direct_max = lim ? std::atoi(lim) : aux::REDUCE_DIRECT_MAX;
    if (G.blocks > direct_max || G.blocks > 32 || G.blocks > E.ncu) return false;
    const int K = (int)E.terms.size(), P = (int)E.ntheta;
    return G.blocks * 256 >= P + K && E.max_contrib <= pk::TRAIN_MAX_CONTRIB;
}

// returns 0 when all nsteps ran inside the kernel, 1 on failure (g_err set), 2 when the kernel's grid barrier timed out: the optimiser state
// and the draw counters are back where the call started and the caller runs the loop
static int adam_steps_train(pinn_engine& E, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w) {
    const int P = (int)E.ntheta;
    Group& G = E.groups[0];
    pk::TrainArgs ta;
    if (train_common(E, ta, term_w)) return 1;
    if (E.c12_cap < nsteps) {
        plat_sync(E.stream);
        plat_free(E.d_c12);
        E.d_c12 = (float*)plat_malloc(sizeof(float) * 2 * (size_t)nsteps);
        E.c12_cap = E.d_c12 ? nsteps : 0;
        if (!E.d_c12) return fail("device allocation failed (bias-correction table)");
    }
    std::vector<float> c12(2 * (size_t)nsteps);
    for (int s = 0; s < nsteps; ++s) {
        c12[2 * s] = (float)(1.0 / (1.0 - std::pow((double)beta1, (double)(E.opt_t + s + 1))));
        c12[2 * s + 1] = (float)(1.0 / (1.0 - std::pow((double)beta2, (double)(E.opt_t + s + 1))));
    }
    // snapshot of the optimiser state: a launch whose workgroups were not all resident (another process filling the device) ends by its
    // barrier time-out with wrong numbers — then the state is restored and the caller runs the stand-alone loop instead
    if (!E.d_opt_bak) {
        E.d_opt_bak = (float*)plat_malloc(sizeof(float) * 3 * (size_t)P);
        if (!E.d_opt_bak) return fail("device allocation failed (optimiser snapshot)");
    }
    plat_d2d(E.d_opt_bak, E.d_opt_theta, sizeof(float) * (size_t)P, E.stream);
    plat_d2d(E.d_opt_bak + P, E.d_opt_m, sizeof(float) * (size_t)P, E.stream);
    plat_d2d(E.d_opt_bak + 2 * (size_t)P, E.d_opt_v, sizeof(float) * (size_t)P, E.stream);
    std::vector<unsigned> draws0(E.terms.size());
    for (size_t t = 0; t < E.terms.size(); ++t) draws0[t] = E.terms[t].draws;
    plat_h2d(E.d_c12, c12.data(), sizeof(float) * c12.size(), E.stream);
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());       // (c12 is a pageable temporary)
    pack_all(E, E.d_opt_theta, false);                        // the weight image of the current theta; the kernel keeps it current from here on
    ta.out = E.d_opt_out; ta.lossraw = E.d_lossraw;
    ta.theta = E.d_opt_theta; ta.m = E.d_opt_m; ta.v = E.d_opt_v;
    ta.lr = lr; ta.b1 = beta1; ta.b2 = beta2; ta.eps = eps;
    ta.inv_ptr = E.d_inv_ptr; ta.inv_pos = E.d_inv_pos; ta.packed = E.netplans[G.net].d_packed;
    ta.w_over_n = E.d_w_over_n;
    // redrawn point sets (device samplers): the host draws the set of a launch's FIRST step, exactly as the loop does before every step;
    // the kernel draws the sets of the following steps itself (pinn_train.hpp: train_resample)
    bool any_sampler = false;
    for (auto& T : E.terms) any_sampler = any_sampler || T.sampler != 0;
    ta.fenced = any_sampler ? 1 : 0;
    // launches of at most TRAIN_CHUNK iterations
    const char* chunk_env = std::getenv("PINN_TRAIN_CHUNK");                        // (tests: several launches per call)
    const int TRAIN_CHUNK = (chunk_env && std::atoi(chunk_env) > 0) ? std::atoi(chunk_env) : 4096;
    for (int s0 = 0; s0 < nsteps; s0 += TRAIN_CHUNK) {
        ta.nsteps = std::min(TRAIN_CHUNK, nsteps - s0);
        ta.c12 = E.d_c12 + 2 * (size_t)s0;
        ta.hist = E.d_hist + s0;
        if (any_sampler) {
            int max_n = 0;
            const int ns = upload_resample_table(E, &max_n);      // draw counters of this launch's first step
            if (ns < 0) return ns == -2 ? 1 : fail("pinn_adam_steps: a redrawn term of the persistent kernel carries embeddings or per-point data");
            aux::launch_resample((const pk::TrainSampler*)E.d_train_samp, ns, max_n, 0, E.stream);
            for (auto& T : E.terms) if (T.sampler != 0) T.draws += (unsigned)ta.nsteps;
            ta.nsamp = ns;
            ta.samp = (const pk::TrainSampler*)E.d_train_samp;
        }
        ta.arrivals0 = E.bar_arrivals;
        E.bar_arrivals += (unsigned)G.blocks * 2u * (unsigned)ta.nsteps;
        G.spec->train(G.ga, ta, G.blocks, E.stream);
    }
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    if (train_timed_out(E)) {
        // back to the state the call started from
        plat_d2d(E.d_opt_theta, E.d_opt_bak, sizeof(float) * (size_t)P, E.stream);
        plat_d2d(E.d_opt_m, E.d_opt_bak + P, sizeof(float) * (size_t)P, E.stream);
        plat_d2d(E.d_opt_v, E.d_opt_bak + 2 * (size_t)P, sizeof(float) * (size_t)P, E.stream);
        for (size_t t = 0; t < E.terms.size(); ++t) E.terms[t].draws = draws0[t];
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        return 2;
    }
    E.opt_t += nsteps;
    return 0;
}

// checks and per-call device state shared by the Adam entry points
static int adam_prepare(pinn_engine& E, int nsteps, const float* term_w, const char* who) {
    if (!E.d_opt_theta) return fail(std::string(who) + ": call pinn_adam_init first");
    if (nsteps <= 0) return fail(std::string(who) + ": nsteps must be positive");
    if (ensure_points(E)) return 1;
    const int K = (int)E.terms.size();
    for (auto& T : E.terms)
        if (T.sampler == 3 && T.seed == 0 && E.comm_size > 1)
            return fail(std::string(who) + ": an un-randomised Sobol design (seed 0) is the same on every rank and cannot be sharded; give the sampler a seed");
    if (E.hist_cap < nsteps) {
        plat_free(E.d_hist);
        E.d_hist = (double*)plat_malloc(sizeof(double) * nsteps);
        E.hist_cap = nsteps;
        if (!E.d_hist) return fail("device allocation failed (loss history)");
    }
    std::vector<float> wn(K);
    for (int k = 0; k < K; ++k) wn[k] = (term_w ? term_w[k] : 1.0f) / (float)E.terms[k].n_norm;
    plat_h2d(E.d_w_over_n, wn.data(), sizeof(float) * K, E.stream);
    return plat_sync(E.stream) ? fail(std::string("device error: ") + plat_last_error()) : 0;      // (wn is a pageable temporary)
}

int pinn_adam_init_f64(pinn_handle h, const double* theta, int64_t p) {
    if (!h || !theta) return fail("pinn_adam_init_f64: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (p != E.ntheta) return fail("pinn_adam_init_f64: theta length mismatch");
    if (E.f64) return f64_adam_init(E, theta);
    std::vector<float> th(theta, theta + p);             // fp32 mode: narrowed at the boundary
    return pinn_adam_init(h, th.data(), p);
}
int pinn_adam_get_f64(pinn_handle h, double* theta, int64_t p) {
    if (!h || !theta) return fail("pinn_adam_get_f64: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (p != E.ntheta) return fail("pinn_adam_get_f64: theta length mismatch");
    if (E.f64) return f64_adam_get(E, theta);
    std::vector<float> th((size_t)p);
    if (pinn_adam_get(h, th.data(), p)) return 1;
    for (int64_t i = 0; i < p; ++i) theta[i] = (double)th[(size_t)i];
    return 0;
}

int pinn_adam_steps(pinn_handle h, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w, double* loss_history) {
    if (!h) return fail("null handle");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (E.f64) {                                         // float64 mode: redraw -> evaluate -> Adam, all in double on the device (f64.cpp)
        if (E.comm && !E.comm_per_process && E.comm_size > 1)
            return fail("pinn_adam_steps: the handle belongs to a single-process communicator (pinn_comm_init_all): use pinn_adam_steps_sharded");
        if (nsteps <= 0) return fail("pinn_adam_steps: nsteps must be positive");
        E.adam_path = 1;
        return f64_adam_steps(E, nsteps, (double)lr, (double)beta1, (double)beta2, (double)eps, term_w, loss_history, &redraw_term_f32);
    }
    if (E.comm && !E.comm_per_process && E.comm_size > 1)
        return fail("pinn_adam_steps: the handle belongs to a single-process communicator (pinn_comm_init_all): use pinn_adam_steps_sharded");
    if (adam_prepare(E, nsteps, term_w, "pinn_adam_steps")) return 1;
    const int K = (int)E.terms.size();
    E.adam_path = 1;
    if (train_eligible(E)) {                     // small problems: every iteration inside one persistent launch (pinn_train.hpp)
        E.adam_path = 2;
        const int rc = adam_steps_train(E, nsteps, lr, beta1, beta2, eps, term_w);
        if (rc == 1) return 1;
        if (rc == 0) {
            if (loss_history && plat_d2h(loss_history, E.d_hist, sizeof(double) * nsteps, E.stream)) return fail("D2H copy failed");
            if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
            return 0;
        }
        E.adam_path = 1;                         // (rc == 2: timed out and restored — the loop below runs the same steps)
    }
    // Default: plain launches, the step index / bias corrections / draw counters as kernel arguments.
    // PINN_GRAPH=1 (experiment, kept for reproduction): everything that changes from step to step lives in device memory and is advanced
    // by a kernel, so every step issues the SAME launch sequence, which is recorded once from the stream and replayed as a hipGraph.
    // Measured on MI355X / ROCm 7.2 (tools/time_adam_loop.py): the replay is SLOWER than the plain launches — cfg1 (1,026 points, 8 small
    // kernels per step) 69 vs 62 us per iteration, cfg2 401 vs 390 us: the loop is bound by the dependent kernels' own latencies, not by
    // host launch cost, and the graph's kernel nodes do not start any closer together than stream launches do.
    const bool want_graph = std::getenv("PINN_GRAPH") != nullptr && !E.comm;       // (read per call: the tests switch it)
    const int graph_rc = (want_graph && nsteps >= 8 && K <= 256) ? adam_steps_graph(E, nsteps, lr, beta1, beta2, eps, term_w) : -1;
    if (graph_rc > 0) return g_err.empty() ? fail("pinn_adam_steps: a step of the graph path failed") : 1;      // state may have advanced: no re-run
    if (graph_rc == 0) {
        E.opt_t += nsteps;
        for (auto& T : E.terms) if (T.sampler != 0) T.draws += (unsigned)nsteps;
    } else {
        pinn_engine* es[1] = {&E};
        if (adam_loop(es, 1, nsteps, lr, beta1, beta2, eps, term_w)) return 1;
    }
    if (loss_history && plat_d2h(loss_history, E.d_hist, sizeof(double) * nsteps, E.stream)) return fail("D2H copy failed");
    std::vector<float> gs;
    if (E.gemm_auto && !E.comm) {                        // "gemm" = "auto": the last step's [gradient | raw sums] (one small copy per CALL)
        gs.resize((size_t)E.ntheta + K);
        if (plat_d2h(gs.data(), E.d_opt_out, sizeof(float) * gs.size(), E.stream)) return fail("D2H copy failed");
    }
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    if (!gs.empty()) {
        measure_grad_health(E, gs.data(), term_w);
        if (gemm_auto_check(E, E.d_opt_theta, term_w, E.opt_t)) return 1;      // (the optimiser state survives a re-plan: replan_gemm)
    }
    return 0;
}

int pinn_adam_steps_sharded(pinn_handle* hs, int ndev, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w,
                            double* loss_history) {
    if (!hs || ndev < 1) return fail("pinn_adam_steps_sharded: need at least one handle");
    for (int i = 0; i < ndev; ++i) {
        if (!hs[i] || !hs[i]->comm || hs[i]->comm_per_process || hs[i]->comm_size != ndev || hs[i]->comm_rank != i)
            return fail("pinn_adam_steps_sharded: pass the handles of one pinn_comm_init_all communicator, in rank order");
        if ((hs[i]->f64 != nullptr) != (hs[0]->f64 != nullptr)) return fail("pinn_adam_steps_sharded: the handles are in different precision modes");
        if (hs[i]->opt_t != hs[0]->opt_t) return fail("pinn_adam_steps_sharded: the handles' optimiser states are at different steps (pinn_adam_init every handle with the same theta)");
        if (hs[i]->f64) continue;
        DeviceScope scope(hs[i]->device);
        if (adam_prepare(*hs[i], nsteps, term_w, "pinn_adam_steps_sharded")) return 1;
    }
    if (hs[0]->f64) {                                    // float64 mode (r06): the double kernels, one all-reduce of [P + K] doubles per iteration
        if (nsteps <= 0) return fail("pinn_adam_steps_sharded: nsteps must be positive");
        return f64_adam_steps_comm(hs, ndev, nsteps, (double)lr, (double)beta1, (double)beta2, (double)eps, term_w, loss_history, &redraw_term_f32);
    }
    if (adam_loop(hs, ndev, nsteps, lr, beta1, beta2, eps, term_w)) return 1;
    {
        DeviceScope scope(hs[0]->device);
        if (loss_history && plat_d2h(loss_history, hs[0]->d_hist, sizeof(double) * nsteps, hs[0]->stream)) return fail("D2H copy failed");
    }
    for (int i = 0; i < ndev; ++i) {
        DeviceScope scope(hs[i]->device);
        if (plat_sync(hs[i]->stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    return 0;
}

int pinn_adam_apply(pinn_handle h, const float* grad_and_sums, int64_t n, float lr, float beta1, float beta2, float eps, const float* term_w,
                    double* loss) {
    if (!h || !grad_and_sums) return fail("pinn_adam_apply: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    const int K = (int)E.terms.size(), P = (int)E.ntheta;
    if (n != (int64_t)P + K) return fail("pinn_adam_apply: the vector must hold P + K floats ([gradient | raw per-term sums])");
    if (E.f64) {                                         // float64 mode: the optimiser state lives in double (f64.cpp)
        std::vector<double> v(grad_and_sums, grad_and_sums + n);
        return f64_adam_apply(E, v.data(), (double)lr, (double)beta1, (double)beta2, (double)eps, term_w, loss);
    }
    if (adam_prepare(E, 1, term_w, "pinn_adam_apply")) return 1;
    if (plat_h2d(E.d_opt_out, grad_and_sums, sizeof(float) * (size_t)(P + K), E.stream)) return fail("H2D copy failed");
    ++E.opt_t;
    const float c1 = (float)(1.0 / (1.0 - std::pow((double)beta1, (double)E.opt_t)));
    const float c2 = (float)(1.0 / (1.0 - std::pow((double)beta2, (double)E.opt_t)));
    aux::launch_total_loss(E.d_hist, 0, E.d_opt_out, P, K, E.d_w_over_n, E.stream);
    aux::launch_adam(E.d_opt_theta, E.d_opt_m, E.d_opt_v, E.d_opt_out, P, lr, beta1, beta2, eps, c1, c2, E.stream);
    if (loss && plat_d2h(loss, E.d_hist, sizeof(double), E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}

int pinn_lbfgs(pinn_handle h, double* theta, int64_t p, int maxiters, int history, double gtol, const float* term_w, double* loss_history,
               int* iters_done) {
    if (!h || !theta) return fail("pinn_lbfgs: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (p != E.ntheta) return fail("pinn_lbfgs: theta length " + std::to_string(p) + " != ntheta " + std::to_string(E.ntheta));
    if (maxiters <= 0 || history <= 0 || history > 64) return fail("pinn_lbfgs: maxiters must be positive and history in 1..64");
    for (auto& T : E.terms)
        if (T.sampler != 0) return fail("pinn_lbfgs: a term redraws its points on the device; L-BFGS needs a fixed objective (fixed point sets)");
    if (ensure_points(E)) return 1;
    const int K = (int)E.terms.size();
    const size_t P = (size_t)p;
    std::vector<float> th32(P);
    std::vector<double> g(P), x(theta, theta + P), xn(P), gn(P), d(P), q(P);
    // one device evaluation: fused loss + gradient, or (grad == nullptr) the loss-only launch — a rejected line-search trial needs the
    // objective only, at about a third of the cost
    auto eval = [&](const std::vector<double>& at, std::vector<double>* grad, double& f) -> int {
        if (E.f64) {                                     // float64 mode: the objective and its gradient in double, no fp32 noise floor
            std::vector<double> w(K, 1.0), L(K);
            if (term_w) for (int k = 0; k < K; ++k) w[k] = term_w[k];
            if (f64_eval(E, at.data(), w.data(), L.data(), grad ? grad->data() : nullptr)) return 1;
            f = 0.0;
            for (int k = 0; k < K; ++k) f += w[k] * L[k];
            return 0;
        }
        for (size_t i = 0; i < P; ++i) th32[i] = (float)at[i];
        if (upload_theta(E, th32.data(), p)) return 1;
        if (grad) {
            if (eval_and_sync(E, E.hp_out, term_w, E.hp_raw, false)) return 1;
        } else {
            if (run_loss_grad(E, E.d_theta, E.hp_out, term_w, -1, false, E.hp_raw, false, true)) return 1;
            if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        }
        f = 0.0;
        for (int k = 0; k < K; ++k) f += (double)(term_w ? term_w[k] : 1.0f) * E.hp_raw[k] / (double)E.terms[k].n_norm;
        if (grad)
            for (size_t i = 0; i < P; ++i) (*grad)[i] = (double)E.hp_out[i];
        return 0;
    };
    auto dot = [&](const std::vector<double>& a, const std::vector<double>& b) { double s = 0.0; for (size_t i = 0; i < P; ++i) s += a[i] * b[i]; return s; };
    double f = 0.0;
    if (eval(x, &g, f)) return 1;
    std::deque<std::vector<double>> S, Y;
    std::deque<double> RHO;
    bool switched = false;
    // the noise-floor switch below only changes arithmetic where a launch group runs a family-2 kernel that exists as a split / fp32 twin
    bool has_twin = false;
    for (const NetPlan& NP : E.netplans) has_twin = has_twin || (NP.spec && NP.spec->family == 2 && NP.spec->twin);
    struct Restore {                                     // every exit path hands the handle back in the mode it came in
        pinn_engine& E; bool armed = false;
        ~Restore() { if (armed && E.gemm != pk::GEMM_SPLIT) { const std::string keep = g_err; replan_gemm(E, pk::GEMM_SPLIT); if (!keep.empty()) g_err = keep; } }
    } restore{E};
    int it = 0;
    for (; it < maxiters; ++it) {
        double gmax = 0.0;
        for (size_t i = 0; i < P; ++i) gmax = std::max(gmax, std::fabs(g[i]));
        if (!(gmax > gtol)) break;
        // two-loop recursion: d = -H g
        q = g;
        std::vector<double> alpha(S.size());
        for (int i = (int)S.size() - 1; i >= 0; --i) {
            alpha[i] = RHO[i] * dot(S[i], q);
            for (size_t j = 0; j < P; ++j) q[j] -= alpha[i] * Y[i][j];
        }
        double gamma = 1.0;
        if (!S.empty()) gamma = dot(S.back(), Y.back()) / dot(Y.back(), Y.back());
        for (size_t j = 0; j < P; ++j) q[j] *= gamma;
        for (size_t i = 0; i < S.size(); ++i) {
            const double beta = RHO[i] * dot(Y[i], q);
            for (size_t j = 0; j < P; ++j) q[j] += (alpha[i] - beta) * S[i][j];
        }
        for (size_t j = 0; j < P; ++j) d[j] = -q[j];
        double gd = dot(g, d);
        if (!(gd < 0.0)) {                               // not a descent direction (stale curvature): restart from steepest descent
            S.clear(); Y.clear(); RHO.clear();
            for (size_t j = 0; j < P; ++j) d[j] = -g[j];
            gd = dot(g, d);
        }
        // backtracking line search (Armijo, c1 = 1e-4); first iteration: step 1 / |g|_1-ish scale
        double t = S.empty() ? std::min(1.0, 1.0 / std::sqrt(dot(g, g))) : 1.0;
        double fn = f;
        bool ok = false;
        // the first trial (accepted most of the time) is a full evaluation: its gradient is the next iterate's; after a rejection the
        // trials are loss-only evaluations and the accepted point gets its gradient from one more full evaluation
        for (int ls = 0; ls < 30; ++ls) {
            for (size_t j = 0; j < P; ++j) xn[j] = x[j] + t * d[j];
            if (eval(xn, ls == 0 ? &gn : nullptr, fn)) return 1;
            if (std::isfinite(fn) && fn <= f + 1e-4 * t * gd) {
                if (ls > 0 && eval(xn, &gn, fn)) return 1;
                ok = true;
                break;
            }
            t *= 0.5;
        }
        if (!ok) {
            // no decrease along a descent direction: the evaluation's noise floor.  The split-operand GEMMs' floor is 2-4 x that of the
            // fp32 MFMA kernels: switch the handle to them once and go on from the same iterate
            if (!E.f64 && E.gemm == pk::GEMM_SPLIT && !switched && has_twin && std::getenv("PINN_LBFGS_KEEP_GEMM") == nullptr) {
                // a failed switch (the fp32 twin of a run-time specialised shape did not compile) leaves the handle in split mode — replan_gemm
                // restored the plan: the iterate reached so far is the answer, not an error
                if (replan_gemm(E, pk::GEMM_FP32)) { if (E.gemm != pk::GEMM_SPLIT || E.groups.empty()) return 1; g_err.clear(); break; }
                switched = E.gemm == pk::GEMM_FP32;
                restore.armed = switched;
                if (switched) {
                    if (eval(x, &g, f)) return 1;
                    S.clear(); Y.clear(); RHO.clear();   // curvature pairs carry the other arithmetic's gradient noise
                    --it;
                    continue;
                }
            }
            break;
        }
        std::vector<double> sv(P), yv(P);
        for (size_t j = 0; j < P; ++j) { sv[j] = xn[j] - x[j]; yv[j] = gn[j] - g[j]; }
        const double sy = dot(sv, yv);
        if (sy > 1e-10 * std::sqrt(dot(sv, sv) * dot(yv, yv))) {
            S.push_back(std::move(sv)); Y.push_back(std::move(yv)); RHO.push_back(1.0 / sy);
            if ((int)S.size() > history) { S.pop_front(); Y.pop_front(); RHO.pop_front(); }
        }
        x.swap(xn); g.swap(gn); f = fn;
        if (loss_history) loss_history[it] = f;
    }
    if (loss_history) for (int i = it; i < maxiters; ++i) loss_history[i] = f;
    if (iters_done) *iters_done = it;
    for (size_t i = 0; i < P; ++i) theta[i] = x[i];
    if (switched) { restore.armed = false; if (replan_gemm(E, pk::GEMM_SPLIT)) return 1; }
    return 0;
}

int pinn_adam_get(pinn_handle h, float* theta, int64_t p) {
    if (!h || !theta) return fail("pinn_adam_get: null argument");
    pinn_engine& E = *h;
    DeviceScope scope(E.device);
    if (E.f64) {
        if (p != E.ntheta) return fail("pinn_adam_get: length mismatch");
        std::vector<double> th((size_t)p);
        if (f64_adam_get(E, th.data())) return 1;
        for (int64_t i = 0; i < p; ++i) theta[i] = (float)th[(size_t)i];
        return 0;
    }
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
    if (!h->timing_valid) return fail("no timing available: switch the phase events on (pinn_set_timing(h, 2, -1)), then call pinn_loss_grad");
    plat_event_sync(h->ev3);
    if (kernel_ms) *kernel_ms = plat_event_ms(h->ev1, h->ev2);
    if (total_ms) *total_ms = plat_event_ms(h->ev0, h->ev3);
    return 0;
}

int pinn_num_groups(pinn_handle h) { return h ? (int)h->groups.size() : -1; }

#ifdef PINN_STAMP
// profiling build only: the four phase-tick sums of the last persistent training launch (pinn_train.hpp: TRAIN_STAMP)
int pinn_debug_train_stamps(pinn_handle h, unsigned long long* out4) {
    if (!h || !h->d_bar || !out4) return -1;
    plat_sync(h->stream);
    plat_d2h(out4, h->d_bar + 4, sizeof(unsigned long long) * 4, h->stream);
    plat_sync(h->stream);
    return 0;
}
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
    os << "backend=" << plat_name() << " cus=" << h->ncu << " ntheta=" << h->ntheta << " terms=" << h->terms.size()
       << (h->f64 ? " precision=f64 (pinn_loss_grad*, pinn_lbfgs evaluate in double — matrix-pipe tile kernels where instantiated, one lane per point elsewhere; the plan below serves the fp32 entry points)" + pe::f64_describe(*h) : std::string()) << "\n";
    for (size_t n = 0; n < h->netplans.size(); ++n)
        if (h->netplans[n].spec && h->netplans[n].spec->family != 3)
            os << "net " << n << " weights=packed image (k_pack per evaluation)"
               << " gemm=" << (h->netplans[n].spec->BFX ? (h->netplans[n].spec->BFX_DW ? "split-bf16(fwd,dA,dW)" : "split-bf16(fwd,dA)") : "fp32") << "\n";
    for (size_t g = 0; g < h->groups.size(); ++g) {
        const Group& G = h->groups[g];
        os << "group " << g << (G.kind == 1 ? (G.use_rec ? " [coupled fwd/gradin, records in HBM]" : " [coupled fwd/gradin]") :
                                (G.kind == 2 ? " [coupled tail: forward + tape + reverse in one launch]" : "")) << " net=" << G.net << " kernel=" << spec_name(*G.spec) << " tiles=" << G.ga.ntiles << " blocks=" << G.blocks << " terms=";
        for (int t : G.terms) os << t << ",";
        if (G.chain_to >= 0) os << " slabs=" << (G.blocks <= h->groups[G.chain_to].blocks ? "chained onto group " : "own (more workgroups than group ") << G.chain_to << (G.blocks <= h->groups[G.chain_to].blocks ? "" : ")");
        if (G.merged >= 0 && h->merged[G.merged].tail == (int)g) os << " launch=merged into group " << h->merged[G.merged].head << "'s (one persistent kernel walks both tile lists)";
        os << "\n";
    }
    for (size_t t = 0; t < h->terms.size(); ++t) {
        const Term& T = h->terms[t];
        if (T.coupled >= 0) continue;
        os << "term " << t << ": tape ops=" << T.tape_ops.size() << " of " << T.ops.size() << ", sources=" << T.src_root.size()
           << " (" << T.src_prog.size() << " coordinate-only ops evaluated per point set)" << (T.linear ? ", affine residual: no tape interpreter" : "") << "\n";
    }
    std::string s = os.str();
    std::snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return 0;
}

}  // extern "C"

#ifdef PINN_EMU
// test hook of the emulation build only (never part of libpinn_hip.so): the device Sobol' bit generator
extern "C" unsigned pinn_emu_sobol_bits(unsigned index, int axis) { return aux::sobol_bits(index, axis); }
#endif
