// f64.cpp — host side of the float64 evaluation mode (pinn_set_option(h, "precision", "f64"); kernels: pinn_kernels4.hpp).
//
// The reference evaluates in Float64 by default (src/discretize.jl:432-449: `init_params` are converted to Float64 unless the chain is
// given Float32 parameters; the EltypeAdaptor then fixes the points' eltype, src/eltype_matching.jl:8-10).  This mode is that arithmetic
// on the device, for callers that need more than fp32 can give at trained parameters: `pinn_loss_grad_f64`
// evaluates natively, `pinn_loss_grad` converts at the boundary, `pinn_lbfgs` iterates on the float64 objective.
// Scope: Dense chains with tanh / sigmoid / sin, equations of one or SEVERAL dependent variables (systems: up to 6 networks per equation, all
// with the same number of inputs), derivative orders <= 2 in 1-3 inputs (1-D: <= 4; 4-D: first and pure second derivatives), PDE parameters
// (param_estim), quadrature weights, per-point DATA channels, device samplers; no periodic embeddings, no DGM networks.  Anything else fails at pinn_set_option with a message — the fp32 plan of the handle stays usable.
#include "engine_types.hpp"
#include "pinn_kernels5.hpp"

namespace pk {
std::deque<F64Kernel>& f64_registry() {
    static std::deque<F64Kernel> r;
    return r;
}
std::deque<F64MKernel>& f64m_registry() {
    static std::deque<F64MKernel> r;
    return r;
}
}  // namespace pk

namespace pe {

struct F64Term {
    const pk::F64Kernel* k = nullptr;
    const pk::F64MKernel* km = nullptr;  // the matrix-pipe kernels of the same jet set (family 4m), nullptr: one lane per point (family 4)
    std::vector<int> nets;               // networks the equation references, increasing
    std::vector<int> slot_net, slot_chan;      // per slot: index into `nets`, jet channel
    rp::Instr* d_prog = nullptr;
    double* d_imm = nullptr;
    int nops = 0, out_row = 0, nslots = 0;
    double* d_pts = nullptr;             // [n][d] double; converted from the float set unless pinn_set_points_f64 installed it
    int64_t cap = 0, n = 0;
    bool exact_pts = false;              // installed in double (not a conversion of the float set)
    int ndata = 0;                       // per-point DATA channels of the residual (OP_DATA), valid for the current point set only
    double* d_data = nullptr;            // [ndata][n]; converted from the float rows unless pinn_set_point_data_f64 installed them
    int64_t data_cap = 0, data_n = 0;
};
struct F64State {
    std::vector<F64Term> terms;
    double* d_theta = nullptr;
    double* d_grad = nullptr;            // [P]
    double* d_sumsq = nullptr;           // [K]
    double* d_scratch = nullptr;
    size_t scratch_cap = 0;
    double* d_slab = nullptr;
    size_t slab_cap = 0;
    double* d_tpart = nullptr;           // matrix-pipe kernels: per-tile partial sums of the first / last layer entries (F64Args::tpart)
    size_t tpart_cap = 0;
    std::vector<double> h_out;           // host staging [P + K]
    double* d_m = nullptr;               // optimiser moments of the float64 Adam loop (f64_adam_*), allocated on first use
    double* d_v = nullptr;
    double* d_w_over_n = nullptr;        // [K] w_k / N_k of the running call
    double* d_hist = nullptr;
    int hist_cap = 0;
    bool opt_ready = false;              // d_theta holds the optimiser's parameters
    int path = 0;                        // kernels of the last evaluation: bit 0 one lane per point (family 4), bit 1 matrix pipe (family 4m)
};

static void f64_free(F64State* S) {
    if (!S) return;
    for (auto& T : S->terms) { plat_free(T.d_prog); plat_free(T.d_imm); plat_free(T.d_pts); plat_free(T.d_data); }
    plat_free(S->d_theta); plat_free(S->d_grad); plat_free(S->d_sumsq); plat_free(S->d_scratch); plat_free(S->d_slab); plat_free(S->d_tpart);
    plat_free(S->d_m); plat_free(S->d_v); plat_free(S->d_w_over_n); plat_free(S->d_hist);
    delete S;
}
void f64_destroy(pinn_engine& E) {
    f64_free((F64State*)E.f64);
    E.f64 = nullptr;
}

// the smallest float64 kernel of `D` inputs that carries every derivative slot of the term
static const pk::F64Kernel* f64_find(int D, const std::vector<Slot>& slots, std::vector<int>& chan, std::string& why) {
    const pk::F64Kernel* best = nullptr;
    for (const pk::F64Kernel& k : pk::f64_registry()) {
        if (k.D != D) continue;
        pk::SpecInfo s;
        std::memset(&s, 0, sizeof s);
        s.D = D; s.D1MASK = k.D1MASK; s.PAIRS = k.PAIRS; s.NPAIR = k.NPAIR; s.HI = k.HI; s.NFIRST = k.NFIRST; s.LAP = 0; s.ngen = 0;
        bool ok = true;
        std::vector<int> ch;
        for (auto& sl : slots) {
            if (sl.lap || slot_is_general(sl)) { ok = false; break; }
            if (sl.order == 1 && !((k.D1MASK >> sl.axes[0]) & 1)) { ok = false; break; }     // (chan_of ranks first derivatives inside the mask, it does not test membership)
            const int c = chan_of(s, sl);
            if (c < 0) { ok = false; break; }
            ch.push_back(c);
        }
        if (ok && (!best || k.C < best->C)) { best = &k; chan = ch; }
    }
    if (!best) why = "no float64 kernel carries this term's derivatives (orders <= 2 in 1-3 inputs; 1-D: <= 4; 4-D: first and pure second derivatives)";
    return best;
}

static int f64_convert_points(pinn_engine& E, F64Term& F, const Term& T) {
    // the float set as installed -> double (exact conversion of the fp32 values the fp32 kernels read)
    const int64_t n = T.n;
    F.data_n = 0;                                        // (per-point data belong to the previous set)
    if (n <= 0 || !T.d_pts) { F.n = 0; return 0; }
    std::vector<float> h((size_t)n * T.d);
    if (plat_d2h(h.data(), T.d_pts, sizeof(float) * h.size(), E.stream) || plat_sync(E.stream)) return fail("D2H copy of points failed");
    std::vector<double> hd(h.begin(), h.end());
    if (F.cap < n) {
        plat_free(F.d_pts);
        F.d_pts = (double*)plat_malloc(sizeof(double) * hd.size());
        if (!F.d_pts) { F.cap = 0; return fail("device allocation failed (float64 points)"); }
        F.cap = n;
    }
    if (plat_h2d(F.d_pts, hd.data(), sizeof(double) * hd.size(), E.stream) || plat_sync(E.stream)) return fail("H2D copy of points failed");
    F.n = n;
    F.exact_pts = false;
    return 0;
}

// the term's DATA rows in double: `src` given in double (pinn_set_point_data_f64), or the conversion of the float rows as installed
static int f64_install_data(pinn_engine& E, F64Term& F, const Term& T, const double* src) {
    const int64_t n = T.n;
    if (F.ndata == 0 || n <= 0) return 0;
    std::vector<double> hd;
    if (!src) {
        if (!T.d_data || T.data_n != n) { F.data_n = 0; return 0; }
        std::vector<float> h((size_t)F.ndata * n);
        if (plat_d2h(h.data(), T.d_data, sizeof(float) * h.size(), E.stream) || plat_sync(E.stream)) return fail("D2H copy of point data failed");
        hd.assign(h.begin(), h.end());
        src = hd.data();
    }
    if (F.data_cap < n) {
        plat_sync(E.stream);
        plat_free(F.d_data);
        F.d_data = (double*)plat_malloc(sizeof(double) * (size_t)F.ndata * n);
        if (!F.d_data) { F.data_cap = 0; return fail("device allocation failed (float64 point data)"); }
        F.data_cap = n;
    }
    if (plat_h2d(F.d_data, src, sizeof(double) * (size_t)F.ndata * n, E.stream) || plat_sync(E.stream)) return fail("H2D copy of point data failed");
    F.data_n = n;
    return 0;
}

int f64_enable(pinn_engine& E) {
    if (E.f64) return 0;
    if (E.terms0.size() != E.terms.size()) return fail("precision f64: internal (no pristine copy of the terms)");
    std::unique_ptr<F64State, void (*)(F64State*)> S(new F64State(), &f64_free);
    S->terms.resize(E.terms0.size());
    for (size_t t = 0; t < E.terms0.size(); ++t) {
        const Term& T = E.terms0[t];
        F64Term& F = S->terms[t];
        const std::string who = "precision f64: term " + std::to_string(t) + ": ";
        std::vector<int> nets;
        for (auto& sl : T.slots) if (std::find(nets.begin(), nets.end(), sl.net) == nets.end()) nets.push_back(sl.net);
        std::sort(nets.begin(), nets.end());
        if (nets.empty()) return fail(who + "references no dependent variable");
        if ((int)nets.size() > pk::F64_MAX_NETS) return fail(who + "references more than 6 dependent variables");
        F.nets = nets;
        bool any_sin = false, all_sin = true;
        for (int net : nets) {
            const Net& N = E.nets[net];
            if (N.kind != 0) return fail(who + "DGM networks are not covered by the float64 mode");
            if (!N.emb_idx.empty()) return fail(who + "periodic input embeddings are not covered by the float64 mode");
            if (N.act != pk::ACT_TANH && N.act != pk::ACT_SIGMOID && N.act != pk::ACT_SIN) return fail(who + "per-layer activation mixes are not covered by the float64 mode");
            if ((int)N.sizes.size() - 1 > pk::F64_MAX_LAYERS) return fail(who + "more than 16 Dense layers");
            if (N.sizes[0] != E.nets[nets[0]].sizes[0]) return fail(who + "its dependent variables take different numbers of arguments (one jet set serves all networks of an equation in the float64 mode)");
            any_sin = any_sin || N.act == pk::ACT_SIN;
            all_sin = all_sin && N.act == pk::ACT_SIN;
        }
        if (any_sin && !all_sin) return fail(who + "mixes sin networks with tanh / sigmoid networks");
        if (!T.emb_cols.empty()) return fail(who + "periodic input embeddings are not covered by the float64 mode");
        if (T.d > 4 || E.nets[nets[0]].sizes[0] > 4) return fail(who + "more than 4 coordinates");
        if ((int)T.slots.size() > pk::F64_MAX_SLOTS || T.d + E.np + (int)T.slots.size() + (int)T.ops.size() > pk::F64_MAX_ROWS)
            return fail(who + "residual expression too long for the float64 tape (96 rows)");
        F.ndata = E.terms[t].ndata;                      // (OP_DATA rows stay in this mode's tape: the float path hoists them into source channels)
        std::string why;
        F.k = f64_find(E.nets[nets[0]].sizes[0], T.slots, F.slot_chan, why);
        if (!F.k) return fail(who + why);
        // family 4m (v_mfma_f64_16x16x4_f64): tanh / sigmoid networks with hidden layers no wider than an instantiated 16 * HT
        {
            int maxh = 0;
            bool ok = !any_sin;
            for (int net : nets) {
                const Net& N = E.nets[net];
                if (N.sizes.size() < 3) ok = false;
                for (size_t j = 1; j + 1 < N.sizes.size(); ++j) maxh = std::max(maxh, N.sizes[j]);
            }
            F.km = nullptr;
            if (ok)
                for (const pk::F64MKernel& km : pk::f64m_registry())
                    if (km.D == F.k->D && km.D1MASK == F.k->D1MASK && km.PAIRS == F.k->PAIRS && km.NPAIR == F.k->NPAIR && km.HI == F.k->HI &&
                        16 * km.HT >= maxh && (!F.km || km.HT < F.km->HT)) F.km = &km;
        }
        F.slot_net.clear();
        for (auto& sl : T.slots) F.slot_net.push_back((int)(std::find(nets.begin(), nets.end(), sl.net) - nets.begin()));
        F.nops = (int)T.ops.size(); F.out_row = T.out_row; F.nslots = (int)T.slots.size();
        std::vector<double> imm(T.ops.size());
        for (size_t q = 0; q < T.ops.size(); ++q) imm[q] = q < T.imm64.size() ? T.imm64[q] : (double)T.ops[q].imm;
        F.d_prog = (rp::Instr*)plat_malloc(sizeof(rp::Instr) * std::max<size_t>(T.ops.size(), 1));
        F.d_imm = (double*)plat_malloc(sizeof(double) * std::max<size_t>(T.ops.size(), 1));
        if (!F.d_prog || !F.d_imm) return fail("device allocation failed (float64 programs)");
        if (!T.ops.empty()) {
            plat_h2d(F.d_prog, T.ops.data(), sizeof(rp::Instr) * T.ops.size(), E.stream);
            plat_h2d(F.d_imm, imm.data(), sizeof(double) * imm.size(), E.stream);
            if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
        }
        if (f64_convert_points(E, F, E.terms[t])) return 1;
        if (f64_install_data(E, F, E.terms[t], nullptr)) return 1;
    }
    const int K = (int)E.terms.size();
    S->d_theta = (double*)plat_malloc(sizeof(double) * E.ntheta);
    S->d_grad = (double*)plat_malloc(sizeof(double) * E.ntheta);
    S->d_sumsq = (double*)plat_malloc(sizeof(double) * K);
    if (!S->d_theta || !S->d_grad || !S->d_sumsq) return fail("device allocation failed (float64 state)");
    S->h_out.resize((size_t)E.ntheta + K);
    E.f64 = S.release();
    return 0;
}

// a float set was (re)installed through pinn_set_points while the mode is on: keep the double copy in step
int f64_points_changed(pinn_engine& E, int term) {
    if (!E.f64) return 0;
    F64State& S = *(F64State*)E.f64;
    return f64_convert_points(E, S.terms[term], E.terms[term]);
}

int f64_set_points(pinn_engine& E, int term, const double* pts, int64_t n) {
    F64State& S = *(F64State*)E.f64;
    F64Term& F = S.terms[term];
    const Term& T = E.terms[term];
    if (F.cap < n) {
        plat_sync(E.stream);
        plat_free(F.d_pts);
        F.d_pts = (double*)plat_malloc(sizeof(double) * (size_t)n * T.d);
        if (!F.d_pts) { F.cap = 0; return fail("device allocation failed (float64 points)"); }
        F.cap = n;
    }
    if (plat_h2d(F.d_pts, pts, sizeof(double) * (size_t)n * T.d, E.stream) || plat_sync(E.stream)) return fail("H2D copy of points failed");
    F.n = n;
    F.exact_pts = true;
    return 0;
}

// pinn_set_point_data while the mode is on: the double rows follow (data == nullptr: converted from the float rows just installed)
int f64_set_point_data(pinn_engine& E, int term, const double* data) {
    if (!E.f64) return 0;
    F64State& S = *(F64State*)E.f64;
    return f64_install_data(E, S.terms[term], E.terms[term], data);
}

static int f64_eval_device(pinn_engine& E, const double* term_w, bool want_grad);

int f64_eval(pinn_engine& E, const double* theta, const double* term_w, double* term_losses, double* grad) {
    F64State& S = *(F64State*)E.f64;
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    if (plat_h2d(S.d_theta, theta, sizeof(double) * P, E.stream)) return fail("H2D copy of theta failed");
    S.opt_ready = false;                                 // (d_theta no longer holds the optimiser's iterate)
    if (f64_eval_device(E, term_w, grad != nullptr)) return 1;
    if (plat_d2h(S.h_out.data(), S.d_grad, sizeof(double) * P, E.stream)) return fail("D2H copy failed");
    if (plat_d2h(S.h_out.data() + P, S.d_sumsq, sizeof(double) * K, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    if (term_losses)
        for (int k = 0; k < K; ++k) term_losses[k] = S.h_out[(size_t)P + k] / (double)E.terms[k].n_norm;
    if (grad) std::memcpy(grad, S.h_out.data(), sizeof(double) * P);
    return 0;
}

// loss sums (S.d_sumsq) and gradient (S.d_grad) of the parameters in S.d_theta, everything on the device, nothing synchronised
static int f64_eval_device(pinn_engine& E, const double* term_w, bool want_grad) {
    F64State& S = *(F64State*)E.f64;
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    const double* grad = want_grad ? S.d_grad : nullptr;         // (non-null = evaluate the gradient)
    for (int t = 0; t < K; ++t)
        if (S.terms[t].n <= 0 || S.terms[t].n != E.terms[t].n) return fail("term " + std::to_string(t) + " has no collocation points (call pinn_set_points first)");
    for (int t = 0; t < K; ++t)
        if (S.terms[t].ndata > 0 && S.terms[t].data_n != S.terms[t].n)
            return fail("term " + std::to_string(t) + " uses per-point data channels but none are installed for its current point set (call pinn_set_point_data after pinn_set_points)");
    // the first reduction of the evaluation writes the gradient instead of adding to it when its term's entries cover all of theta (one network,
    // or every network in the first equation); otherwise one memset.  Every term's first chunk writes its own sum of squares.
    bool grad_started = false;
    if (want_grad) {
        int64_t covered = E.ne;
        for (int ni : S.terms[0].nets) covered += E.nets[ni].nparams();
        if (covered != P) { plat_memset(S.d_grad, 0, sizeof(double) * P, E.stream); grad_started = true; }
    }
    S.path = 0;
    for (int t = 0; t < K; ++t) {
        const Term& T = E.terms[t];
        const Term& T0 = E.terms0[t];
        F64Term& F = S.terms[t];
        pk::F64Args a;
        std::memset(&a, 0, sizeof a);
        a.data = F.ndata > 0 ? F.d_data : nullptr;
        a.theta = S.d_theta; a.pts = F.d_pts; a.pw = (T.pw_n == T.n && T.pw_n > 0) ? T.d_pw : nullptr;
        a.N = (int)F.n; a.dt = T0.d;
        a.nnets = (int)F.nets.size();
        a.C = F.k->C;
        for (int i = 0; i < 8; ++i) a.first_ch[i] = F.k->first_ch[i];
        int rows = 0, ent = 0;
        bool sin_act = false;
        const bool mfma_rows = F.km && std::getenv("PINN_F64_NO_MFMA") == nullptr;
        for (int ni = 0; ni < a.nnets; ++ni) {
            const Net& N = E.nets[F.nets[ni]];
            pk::F64Net& n = a.net[ni];
            n.d = N.sizes[0];
            std::vector<int> m;
            if (T0.inmap.count(F.nets[ni])) m = T0.inmap.at(F.nets[ni]);
            else for (int i = 0; i < n.d; ++i) m.push_back(i);
            if ((int)m.size() != n.d) return fail("precision f64: inmap length differs from the network's input count");
            for (int i = 0; i < 4; ++i) n.imap[i] = i < (int)m.size() ? m[i] : 0;
            n.nl = (int)N.sizes.size() - 1;
            int o = N.theta_off;
            for (int l = 0; l < n.nl; ++l) {
                n.sizes[l] = N.sizes[l];
                n.woff[l] = o;
                n.boff[l] = o + N.sizes[l + 1] * N.sizes[l];
                o = n.boff[l] + N.sizes[l + 1];
            }
            n.sizes[n.nl] = N.sizes[n.nl];
            n.act = N.act;
            sin_act = sin_act || N.act == pk::ACT_SIN;
            n.theta0 = N.theta_off; n.nparams = N.nparams(); n.ent0 = ent;
            ent += n.nparams;
            n.tp0 = a.tp_p;                                       // (running column count of the tile partial sums)
            a.tp_p += (n.d + 1) * N.sizes[1] + N.sizes[n.nl - 1] + 1;
            // scratch rows: per hidden layer record / post-activation jets / dZ, then this network's seeds
            const int L = n.nl - 1;
            for (int l = 0; l < L; ++l) { n.r_rec[l] = rows; rows += N.sizes[l + 1] * a.C; }
            // value-only terms of tanh / sigmoid networks: the record of an element IS its post-activation value (act_record), so the
            // matrix-pipe kernels keep ONE copy (a third of the tile kernel's stores less; 4 of the bench workload's 5 terms)
            const bool post_is_rec = mfma_rows && a.C == 1 && N.act != pk::ACT_SIN;
            a.post_alias = post_is_rec ? 1 : 0;
            for (int l = 0; l < L; ++l) { if (post_is_rec) n.r_post[l] = n.r_rec[l]; else { n.r_post[l] = rows; rows += N.sizes[l + 1] * a.C; } }
            for (int l = 0; l < L; ++l) { n.r_dz[l] = rows; rows += N.sizes[l + 1] * a.C; }
            n.r_ubar = rows; rows += a.C;
        }
        a.ent_p = ent;
        a.nent = ent + E.ne + 1;
        a.r_pbar = rows; rows += std::max(E.ne, 1);
        a.r_sq = rows; rows += 1;
        a.np = E.np; a.ne = E.ne; a.p_off = E.p_theta_off;
        for (int j = 0; j < pk::MAX_PARAMS; ++j) a.pdef[j] = j < (int)E.p_defaults.size() ? (double)E.p_defaults[j] : 0.0;
        a.prog = F.d_prog; a.imm = F.d_imm; a.nops = F.nops; a.out_row = F.out_row; a.nslots = F.nslots;
        for (int s = 0; s < F.nslots; ++s) { a.slot_chan[s] = F.slot_chan[s]; a.slot_net[s] = F.slot_net[s]; }

        const double w = term_w ? term_w[t] : 1.0;
        a.scale = 2.0 * w / (double)T.n_norm;
        a.mode = grad ? 0 : 1;
        const bool mfma = F.km && std::getenv("PINN_F64_NO_MFMA") == nullptr;
        S.path |= mfma ? 2 : 1;
        // chunks of points: the scratch stays below 256 MB (one lane per point) / 4 GB (matrix-pipe kernels: one wave per 16-32 points, a chunk
        // should hold several tiles per SIMD — the bench workload's 65,536-point terms are one chunk each: 3 kernels + a reduction per term); $PINN_F64_SCRATCH_MB overrides
        double mb = mfma ? 4096.0 : 256.0;
        if (const char* e = std::getenv("PINN_F64_SCRATCH_MB")) mb = std::max(1.0, std::atof(e));
        int64_t chunk = (int64_t)((mb * 1024 * 1024) / (8.0 * rows));
        chunk = std::max<int64_t>(pk::F64_BLOCK, (chunk / pk::F64_BLOCK) * pk::F64_BLOCK);
        chunk = std::min<int64_t>(chunk, ((F.n + pk::F64_BLOCK - 1) / pk::F64_BLOCK) * pk::F64_BLOCK);
        const size_t need = (size_t)rows * (size_t)chunk;
        if (need > S.scratch_cap) {
            plat_sync(E.stream);
            plat_free(S.d_scratch);
            S.d_scratch = (double*)plat_malloc(sizeof(double) * need);
            S.scratch_cap = S.d_scratch ? need : 0;
            if (!S.d_scratch) return fail("device allocation failed (float64 scratch)");
        }
        const int nbmax = (int)(chunk / pk::F64_BLOCK);
        const size_t sneed = (size_t)nbmax * (size_t)a.nent;
        if (sneed > S.slab_cap) {
            plat_sync(E.stream);
            plat_free(S.d_slab);
            S.d_slab = (double*)plat_malloc(sizeof(double) * sneed);
            S.slab_cap = S.d_slab ? sneed : 0;
            if (!S.d_slab) return fail("device allocation failed (float64 slabs)");
        }
        a.scratch = S.d_scratch; a.npad = (int)chunk; a.slab = S.d_slab; a.nrows = rows;
        if (mfma) {
            a.ntp = a.tp_p + E.ne + 1;
            a.tile_pts = 16 * F.km->PG;
            const size_t tneed = (size_t)(chunk / a.tile_pts + 1) * (size_t)a.ntp;
            if (tneed > S.tpart_cap) {
                plat_sync(E.stream);
                plat_free(S.d_tpart);
                S.d_tpart = (double*)plat_malloc(sizeof(double) * tneed);
                S.tpart_cap = S.d_tpart ? tneed : 0;
                if (!S.d_tpart) return fail("device allocation failed (float64 tile sums)");
            }
            a.tpart = S.d_tpart;
        }
        for (int64_t p0 = 0; p0 < F.n; p0 += chunk) {
            a.p0 = (int)p0;
            a.npts = (int)std::min<int64_t>(chunk, F.n - p0);
            if (mfma) F.km->launch_tile(a, E.stream);
            else F.k->launch_point(a, sin_act, E.stream);
            if (!mfma) pk::launch_f64_dw(a, E.stream);           // (matrix-pipe path: those entries come out of the tile kernel, summed in the dW launch)
            if (mfma) F.km->launch_dwt(a, E.stream);
            else pk::launch_f64_dwt(a, E.stream);
            pk::F64ReduceArgs r;
            std::memset(&r, 0, sizeof r);
            r.slab = S.d_slab; r.nblocks = (a.npts + pk::F64_BLOCK - 1) / pk::F64_BLOCK; r.nent = a.nent;
            r.grad = S.d_grad; r.ent_p = a.ent_p; r.p_off = E.p_theta_off; r.nnets = a.nnets;
            for (int ni = 0; ni < a.nnets; ++ni) { r.ent0[ni] = a.net[ni].ent0; r.theta0[ni] = a.net[ni].theta0; }
            r.sumsq = S.d_sumsq + t; r.with_grad = grad ? 1 : 0;
            r.init_sumsq = (p0 == 0) ? 1 : 0;
            r.init_grad = (grad && !grad_started) ? 1 : 0;
            if (grad) grad_started = true;
            if (mfma) pk::launch_f64m_reduce(r, E.stream);
            else pk::launch_f64_reduce(r, E.stream);
        }
    }
    return 0;
}

// ---- the resident optimiser loop in float64 (pinn_adam_* on a handle in float64 mode; r05): theta, moments, points and every kernel of the
// iteration in double on the device — redraw (the fp32 samplers' points, converted on the device: the reference's StochasticTraining /
// QuasiRandomTraining(resampling = true) with Float64 parameters, src/training_strategies.jl:271-282, 365-389) -> evaluate -> Adam ----
int f64_adam_init(pinn_engine& E, const double* theta) {
    F64State& S = *(F64State*)E.f64;
    const int64_t P = E.ntheta;
    const int K = (int)E.terms.size();
    if (!S.d_m) {
        S.d_m = (double*)plat_malloc(sizeof(double) * P);
        S.d_v = (double*)plat_malloc(sizeof(double) * P);
        S.d_w_over_n = (double*)plat_malloc(sizeof(double) * K);
        if (!S.d_m || !S.d_v || !S.d_w_over_n) return fail("device allocation failed (float64 optimiser state)");
    }
    if (plat_h2d(S.d_theta, theta, sizeof(double) * P, E.stream)) return fail("H2D copy of theta failed");
    plat_memset(S.d_m, 0, sizeof(double) * P, E.stream);
    plat_memset(S.d_v, 0, sizeof(double) * P, E.stream);
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    S.opt_ready = true;
    E.opt_t = 0;
    return 0;
}
int f64_adam_get(pinn_engine& E, double* theta) {
    F64State& S = *(F64State*)E.f64;
    if (!S.opt_ready) return fail("pinn_adam_get: no float64 optimiser state (pinn_adam_init after switching to precision f64; evaluations at other parameters reset it)");
    if (plat_d2h(theta, S.d_theta, sizeof(double) * E.ntheta, E.stream) || plat_sync(E.stream)) return fail("D2H copy failed");
    return 0;
}
// a sampled term's float points (just redrawn on the device) -> the double copy the float64 kernels read
int f64_points_from_device(pinn_engine& E, int term) {
    F64State& S = *(F64State*)E.f64;
    F64Term& F = S.terms[term];
    const Term& T = E.terms[term];
    if (F.cap < T.n) {
        plat_sync(E.stream);
        plat_free(F.d_pts);
        F.d_pts = (double*)plat_malloc(sizeof(double) * (size_t)T.n * T.d);
        if (!F.d_pts) { F.cap = 0; return fail("device allocation failed (float64 points)"); }
        F.cap = T.n;
    }
    pk::launch_f64_cvt(T.d_pts, F.d_pts, (int64_t)T.n * T.d, E.stream);
    F.n = T.n;
    F.exact_pts = false;
    return 0;
}
int f64_adam_steps(pinn_engine& E, int nsteps, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss_history, void (*redraw)(pinn_engine&, Term&)) {
    F64State& S = *(F64State*)E.f64;
    if (!S.opt_ready) return fail("pinn_adam_steps: call pinn_adam_init first (float64 mode keeps its own optimiser state)");
    const int K = (int)E.terms.size();
    const int P = (int)E.ntheta;
    std::vector<double> w(K), won(K);
    for (int k = 0; k < K; ++k) { w[k] = term_w ? (double)term_w[k] : 1.0; won[k] = w[k] / (double)E.terms[k].n_norm; }
    plat_h2d(S.d_w_over_n, won.data(), sizeof(double) * K, E.stream);
    if (S.hist_cap < nsteps) {
        plat_sync(E.stream);
        plat_free(S.d_hist);
        S.d_hist = (double*)plat_malloc(sizeof(double) * nsteps);
        S.hist_cap = S.d_hist ? nsteps : 0;
        if (!S.d_hist) return fail("device allocation failed (loss history)");
    }
    for (int s = 0; s < nsteps; ++s) {
        for (size_t t = 0; t < E.terms.size(); ++t) {
            Term& T = E.terms[t];
            if (T.sampler == 0) continue;
            redraw(E, T);                                    // (engine.cpp: the fp32 sampler kernels, draw counter advanced)
            if (f64_points_from_device(E, (int)t)) return 1;
        }
        if (f64_eval_device(E, w.data(), true)) return 1;
        ++E.opt_t;
        const double c1 = 1.0 / (1.0 - std::pow(beta1, (double)E.opt_t)), c2 = 1.0 / (1.0 - std::pow(beta2, (double)E.opt_t));
        pk::launch_f64_total(S.d_hist, s, S.d_sumsq, S.d_w_over_n, K, E.stream);
        pk::launch_f64_adam(S.d_theta, S.d_m, S.d_v, S.d_grad, P, lr, beta1, beta2, eps, c1, c2, E.stream);
    }
    if (loss_history && plat_d2h(loss_history, S.d_hist, sizeof(double) * nsteps, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}
// float64 evaluation at DEVICE-resident float parameters (pinn_loss_grad_device / pinn_loss_device in float64 mode): theta converted on the
// device, [gradient | term sums] converted back into the caller's float vector — the kernels in between are the double ones
int f64_eval_from_device_f32(pinn_engine& E, const float* d_theta, const float* term_w, float* d_out, bool want_grad) {
    F64State& S = *(F64State*)E.f64;
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    std::vector<double> w(K);
    for (int k = 0; k < K; ++k) w[k] = term_w ? (double)term_w[k] : 1.0;
    pk::launch_f64_cvt(d_theta, S.d_theta, P, E.stream);
    S.opt_ready = false;
    if (f64_eval_device(E, w.data(), want_grad)) return 1;
    if (want_grad) pk::launch_f64_narrow(S.d_grad, d_out, P, E.stream);
    pk::launch_f64_narrow(S.d_sumsq, d_out + (want_grad ? P : 0), K, E.stream);
    return 0;
}

// theta and [gradient | raw sums] as DOUBLE device pointers: nothing crosses the boundary in fp32, nothing crosses PCIe
int f64_eval_from_device_f64(pinn_engine& E, const double* d_theta, const float* term_w, double* d_out) {
    F64State& S = *(F64State*)E.f64;
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    std::vector<double> w(K);
    for (int k = 0; k < K; ++k) w[k] = term_w ? (double)term_w[k] : 1.0;
    // the kernels read the caller's theta and write the caller's [gradient | sums] directly: no copies (the handle's own buffers — the optimiser's
    // iterate among them — stay as they are)
    double* const th0 = S.d_theta; double* const g0 = S.d_grad; double* const s0 = S.d_sumsq;
    S.d_theta = const_cast<double*>(d_theta); S.d_grad = d_out; S.d_sumsq = d_out + P;
    const int rc = f64_eval_device(E, w.data(), true);
    S.d_theta = th0; S.d_grad = g0; S.d_sumsq = s0;
    return rc;
}

// " f64_channels=5,1,1,1,1 f64_kernels=mfma:HT4xPG1,mfma:HT4xPG4,..." for pinn_describe
std::string f64_describe(const pinn_engine& E) {
    if (!E.f64) return "";
    const F64State& S = *(const F64State*)E.f64;
    std::string c = " f64_channels=", k = " f64_kernels=";
    for (size_t t = 0; t < S.terms.size(); ++t) {
        const F64Term& F = S.terms[t];
        c += (t ? "," : "") + std::to_string(F.k ? F.k->C : 0);
        k += (t ? "," : "") + (F.km ? "mfma:HT" + std::to_string(F.km->HT) + "xPG" + std::to_string(F.km->PG) : std::string("lanes"));
    }
    return c + k;
}

const char* f64_path(const pinn_engine& E) {
    if (!E.f64) return "off";
    const int p = ((const F64State*)E.f64)->path;
    return p == 2 ? "mfma" : (p == 1 ? "lanes" : (p == 3 ? "mfma+lanes" : "none"));
}

}  // namespace pe
