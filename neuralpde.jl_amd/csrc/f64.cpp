// f64.cpp — host side of the float64 evaluation mode (pinn_set_option(h, "precision", "f64"); kernels: pinn_kernels4.hpp).
//
// The reference evaluates in Float64 by default (src/discretize.jl:432-449: `init_params` are converted to Float64 unless the chain is
// given Float32 parameters; the EltypeAdaptor then fixes the points' eltype, src/eltype_matching.jl:8-10).  This mode is that arithmetic
// on the device, for callers that need more than fp32 can give at trained parameters: `pinn_loss_grad_f64`
// evaluates natively, `pinn_loss_grad` converts at the boundary, `pinn_lbfgs` iterates on the float64 objective.
// Scope: Dense chains with tanh / sigmoid / sin, equations of one or SEVERAL dependent variables (systems: up to 6 networks per equation, all
// with the same number of inputs), derivative orders <= 2 in 1-3 inputs (1-D: <= 4; 4-D: first and pure second derivatives), PDE parameters
// (param_estim), quadrature weights, per-point DATA channels, device samplers, periodic input embeddings (r06: feature rows in double); no DGM networks.  Anything else fails at pinn_set_option with a message — the fp32 plan of the handle stays usable.
#include "engine_types.hpp"
#include "pinn_kernels6.hpp"

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
    // affine residual (r06, f64_affine): r = S(point) + sum_s a_s u_s — the matrix-pipe tile kernel then skips the tape interpreter (F64Sub::lin)
    bool lin = false;
    std::vector<double> lin_a;                           // [nslots]
    std::vector<std::pair<double, int>> lin_terms;       // S = lin_k + sum coef * (tape row that depends on coordinates / constants only)
    double lin_k = 0.0;
    double* d_lin = nullptr;             // [nslots | n]: coefficients, then S at the points of the current set
    int64_t lin_cap = 0, lin_n = 0;      // lin_n == n: S is valid for the installed set
};
// one (network, shift sequence) of a term's finite-difference stencils: the trial function of `net` at the term's points moved by the shifts in order
struct F64VNet { int net; std::vector<std::pair<int, double>> shifts; };
// the reference-semantics evaluation of one term (pinn_set_option(h, "derivative", "stencil"); pinn_kernels4.hpp: k_f64_stape)
struct F64Stencil {
    bool active = false;                 // the term has derivative slots (value-only terms evaluate as always: nothing to difference)
    std::vector<F64VNet> vnets;
    rp::Instr* d_prog = nullptr;         // stencil tape: the difference formulas, then the term's ops; rows [coordinates | params | vnets | ops]
    double* d_imm = nullptr;
    int nops = 0, out_row = 0;
};
struct F64State {
    std::vector<F64Term> terms;
    bool stencil = false;                // derivative slots by the reference's central differences instead of exact jets
    std::vector<F64Stencil> sten;
    double* d_uv = nullptr;              // [nv][n] values of the virtual networks
    double* d_seeds = nullptr;           // [nv][n][2 + ne]
    double* d_spts = nullptr;            // [nv][n][d] their shifted point sets
    size_t uv_cap = 0, seeds_cap = 0, spts_cap = 0;
    double* d_theta = nullptr;           // [P] parameters of the running EVALUATION (host-entry evaluations upload here; never the optimiser's iterate)
    double* d_opt_theta = nullptr;       // [P] the Adam loop's iterate (f64_adam_*): evaluations in between — adaptive reweighting, callbacks — leave it alone
    double* d_grad = nullptr;            // [P]
    double* d_sumsq = nullptr;           // [K]
    double* d_scratch = nullptr;
    size_t scratch_cap = 0;
    double* d_slab = nullptr;
    size_t slab_cap = 0;
    double* d_tpart = nullptr;           // matrix-pipe kernels: per-tile partial sums of the first / last layer entries (F64Args::tpart)
    size_t tpart_cap = 0;
    double* h_pin = nullptr;             // pinned host staging: [P] theta on its way in | [P + K] gradient and sums on their way out (one copy each way)
    double* d_m = nullptr;               // optimiser moments of the float64 Adam loop (f64_adam_*), allocated on first use
    double* d_v = nullptr;
    double* d_w_over_n = nullptr;        // [K] w_k / N_k of the running call
    double* d_hist = nullptr;
    int hist_cap = 0;
    bool opt_ready = false;              // d_opt_theta holds the optimiser's parameters
    double* d_aux_pts = nullptr;         // caller-supplied points of pinn_phi_f64 / pinn_derivative_f64 [n][d]
    double* d_aux_out = nullptr;         // per-point outputs (residuals, trial-function values / derivatives) [n]
    int64_t aux_pts_cap = 0, aux_out_cap = 0;
    int path = 0;                        // kernels of the last evaluation: bit 0 one lane per point (family 4), bit 1 matrix pipe (family 4m)
    // MERGED launches of small problems (r06, f64_make_groups): terms that share networks, input binding and an instantiated jet set in ONE tile / dW /
    // reduction launch sequence; rebuilt when a term's point count changes
    struct Group { std::vector<int> terms, nets; std::map<int, std::vector<int>> inmap; const pk::F64Kernel* k = nullptr; const pk::F64MKernel* km = nullptr; std::vector<std::vector<int>> slot_chan; };
    std::vector<Group> groups;
    std::vector<int64_t> groups_sig;     // the point counts the grouping was made for (empty: not made yet)
    int merged_launches = 0;             // of the last evaluation (pinn_get_option "f64_merged")
};

static void f64_free(F64State* S) {
    if (!S) return;
    for (auto& T : S->terms) { plat_free(T.d_prog); plat_free(T.d_imm); plat_free(T.d_pts); plat_free(T.d_data); plat_free(T.d_lin); }
    for (auto& X : S->sten) { plat_free(X.d_prog); plat_free(X.d_imm); }
    plat_free(S->d_uv); plat_free(S->d_seeds); plat_free(S->d_spts);
    plat_free(S->d_theta); plat_free(S->d_opt_theta); plat_free(S->d_aux_pts); plat_free(S->d_aux_out); plat_free(S->d_grad); plat_free(S->d_scratch); plat_free(S->d_slab); plat_free(S->d_tpart);
    plat_free(S->d_m); plat_free(S->d_v); plat_free(S->d_w_over_n); plat_free(S->d_hist);
    plat_host_free(S->h_pin);
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

// family 4m (v_mfma_f64_16x16x4_f64) for the jet set of `k`: tanh / sigmoid networks with at least one hidden layer, none wider than an instantiated
// 16 * HT; nullptr: family 4 (one lane per point)
static const pk::F64MKernel* f64_find_m(const pinn_engine& E, const pk::F64Kernel* k, const std::vector<int>& nets) {
    int maxh = 0;
    for (int net : nets) {
        const Net& N = E.nets[net];
        if (N.act == pk::ACT_SIN || N.sizes.size() < 3) return nullptr;
        for (size_t j = 1; j + 1 < N.sizes.size(); ++j) maxh = std::max(maxh, N.sizes[j]);
    }
    // a register-resident kernel (family 4m) before a channel-sliced one (family 4s: every layer through the scratch rows); the narrowest that covers the width
    const bool no_sliced = std::getenv("PINN_F64_NO_SLICED") != nullptr;
    const pk::F64MKernel* best = nullptr;
    for (const pk::F64MKernel& km : pk::f64m_registry()) {
        if (!(km.D == k->D && km.D1MASK == k->D1MASK && km.PAIRS == k->PAIRS && km.NPAIR == k->NPAIR && km.HI == k->HI && 16 * km.HT >= maxh)) continue;
        if (km.sliced && no_sliced) continue;
        if (!best || (km.sliced < best->sliced) || (km.sliced == best->sliced && km.HT < best->HT)) best = &km;
    }
    return best;
}

// ---- affine residuals: r = S(point) + sum_s a_s u_s ----
// Symbolic pass over the term's tape (descriptor rows [coordinates | parameters | slots | ops]): every row is either a constant-coefficient affine
// form in the slots plus coordinate-only rows, or the term is not affine (nonlinear in u, point-dependent coefficients, estimated parameters, DATA
// rows) and keeps the interpreter.  Boundary conditions, linear PDEs with forcing terms (Poisson, heat, wave) qualify; Burgers does not.
static void f64_affine(const pinn_engine& E, const Term& T, F64Term& F) {
    F.lin = false;
    if (std::getenv("PINN_F64_NO_LIN") || F.ndata > 0) return;
    struct Form { bool ok = true; std::vector<double> a; std::vector<std::pair<double, int>> t; double k = 0.0; };
    const int dt = T.d, np = E.np, ns = (int)T.slots.size(), R0 = dt + np + ns;
    std::vector<Form> row((size_t)R0 + T.ops.size());
    for (auto& f : row) f.a.assign((size_t)ns, 0.0);
    for (int i = 0; i < dt; ++i) row[i].t.push_back({1.0, i});
    for (int k = 0; k < np; ++k) { if (k < E.ne) row[dt + k].ok = false; else row[dt + k].k = k < (int)E.p_defaults.size() ? (double)E.p_defaults[k] : 0.0; }
    for (int q = 0; q < ns; ++q) row[dt + np + q].a[q] = 1.0;
    auto has_u = [&](const Form& f) { for (double x : f.a) if (x != 0.0) return true; return false; };
    auto is_const = [&](const Form& f) { return f.ok && !has_u(f) && f.t.empty(); };
    auto scaled = [&](const Form& f, double c) { Form g = f; for (double& x : g.a) x *= c; for (auto& y : g.t) y.first *= c; g.k *= c; return g; };
    auto add = [&](const Form& x, const Form& y, double sy) {
        Form g = x;
        for (int q = 0; q < ns; ++q) g.a[q] += sy * y.a[q];
        for (auto& z : y.t) g.t.push_back({sy * z.first, z.second});
        g.k += sy * y.k;
        return g;
    };
    for (size_t o = 0; o < T.ops.size(); ++o) {
        const rp::Instr& I = T.ops[o];
        const double imm = o < T.imm64.size() ? T.imm64[o] : (double)I.imm;
        Form& f = row[(size_t)R0 + o];
        const Form* A = rp::is_nullary(I.code) ? nullptr : &row[(size_t)I.a];
        const Form* B = rp::is_binary(I.code) ? &row[(size_t)I.b] : nullptr;
        if ((A && !A->ok) || (B && !B->ok) || I.code == rp::OP_DATA) { f.ok = false; continue; }
        const bool u_dep = (A && has_u(*A)) || (B && has_u(*B));
        auto self = [&]() { Form g; g.a.assign((size_t)ns, 0.0); g.t.push_back({1.0, R0 + (int)o}); return g; };     // a coordinate-only row, evaluated by the source pass
        switch (I.code) {
            case rp::OP_CONST: f.k = imm; break;
            case rp::OP_ADD: f = add(*A, *B, 1.0); break;
            case rp::OP_SUB: f = add(*A, *B, -1.0); break;
            case rp::OP_NEG: f = scaled(*A, -1.0); break;
            case rp::OP_ADDC: f = *A; f.k += imm; break;
            case rp::OP_MULC: f = scaled(*A, imm); break;
            case rp::OP_MUL:
                if (is_const(*A)) f = scaled(*B, A->k);
                else if (is_const(*B)) f = scaled(*A, B->k);
                else if (!u_dep) f = self();
                else f.ok = false;
                break;
            case rp::OP_DIV:
                if (is_const(*B) && B->k != 0.0) f = scaled(*A, 1.0 / B->k);
                else if (!u_dep) f = self();
                else f.ok = false;
                break;
            default:
                if (!u_dep) f = self(); else f.ok = false;
                break;
        }
        if (f.ok && f.a.size() != (size_t)ns) f.a.assign((size_t)ns, 0.0);
    }
    const Form& out = row[(size_t)T.out_row];
    if (!out.ok) return;
    // (a MULC / DIV by a constant is not bit-identical to the tape's order of operations: the results agree to rounding, which is the mode's contract)
    std::vector<std::pair<double, int>> terms;
    for (auto& z : out.t) {
        bool merged = false;
        for (auto& y : terms) if (y.second == z.second) { y.first += z.first; merged = true; break; }
        if (!merged) terms.push_back(z);
    }
    if ((int)terms.size() > pk::F64_LIN_MAX_TERMS) return;
    F.lin = true; F.lin_a = out.a; F.lin_terms = terms; F.lin_k = out.k;
}
// S at the points of the installed set (device pass over F.d_pts), behind the coefficients in d_lin
static int f64_lin_install(pinn_engine& E, F64Term& F, const Term& T0) {
    F.lin_n = 0;
    if (!F.lin || F.n <= 0 || !F.d_pts) return 0;
    const int ns = F.nslots;
    if (F.lin_cap < F.n) {
        plat_sync(E.stream);
        plat_free(F.d_lin);
        F.d_lin = (double*)plat_malloc(sizeof(double) * ((size_t)ns + (size_t)F.n));
        F.lin_cap = F.d_lin ? F.n : 0;
        if (!F.d_lin) return fail("device allocation failed (float64 affine residual)");
    }
    if (ns > 0 && plat_h2d(F.d_lin, F.lin_a.data(), sizeof(double) * ns, E.stream)) return fail("H2D copy failed (float64 affine residual)");
    pk::F64LinSrcArgs a;
    std::memset(&a, 0, sizeof a);
    a.pts = F.d_pts; a.N = (int)F.n; a.dt = T0.d; a.np = E.np; a.nslots = ns; a.nops = F.nops;
    for (int j = 0; j < pk::MAX_PARAMS; ++j) a.pdef[j] = j < (int)E.p_defaults.size() ? (double)E.p_defaults[j] : 0.0;
    a.prog = F.d_prog; a.imm = F.d_imm;
    a.nterms = (int)F.lin_terms.size();
    for (int j = 0; j < a.nterms; ++j) { a.coef[j] = F.lin_terms[j].first; a.row[j] = F.lin_terms[j].second; }
    a.k = F.lin_k;
    a.out = F.d_lin + ns;
    pk::launch_f64_lin_src(a, E.stream);
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());      // (lin_a is a host vector; installs are rare)
    F.lin_n = F.n;
    return 0;
}
static const double* f64_lin_ptr(const F64Term& F) { return (F.lin && F.lin_n == F.n && F.n > 0) ? F.d_lin : nullptr; }

// feature rows of an embedded term, in double: row d_user + k = sin / cos (omega_k x coordinate src_k) of every point of the [n][T.d] host image
static void f64_embed_host(const Term& T, std::vector<double>& pts, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        for (size_t k = 0; k < T.emb_cols.size(); ++k) {
            const double ph = T.emb_cols[k].omega * pts[(size_t)i * T.d + T.emb_cols[k].src];
            pts[(size_t)i * T.d + T.d_user + k] = T.emb_cols[k].is_cos ? std::cos(ph) : std::sin(ph);
        }
}
static int f64_convert_points(pinn_engine& E, F64Term& F, const Term& T) {
    // the float set as installed -> double (exact conversion of the fp32 values the fp32 kernels read)
    const int64_t n = T.n;
    F.data_n = 0;                                        // (per-point data belong to the previous set)
    if (n <= 0 || !T.d_pts) { F.n = 0; return 0; }
    std::vector<float> h((size_t)n * T.d);
    if (plat_d2h(h.data(), T.d_pts, sizeof(float) * h.size(), E.stream) || plat_sync(E.stream)) return fail("D2H copy of points failed");
    std::vector<double> hd(h.begin(), h.end());
    if (!T.emb_cols.empty()) f64_embed_host(T, hd, n);    // (the float rows hold float-rounded sin / cos: recomputed from the coordinates in double)
    if (F.cap < n) {
        plat_free(F.d_pts);
        F.d_pts = (double*)plat_malloc(sizeof(double) * hd.size());
        if (!F.d_pts) { F.cap = 0; return fail("device allocation failed (float64 points)"); }
        F.cap = n;
    }
    if (plat_h2d(F.d_pts, hd.data(), sizeof(double) * hd.size(), E.stream) || plat_sync(E.stream)) return fail("H2D copy of points failed");
    F.n = n;
    F.exact_pts = false;
    return f64_lin_install(E, F, T);
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
            if (N.act != pk::ACT_TANH && N.act != pk::ACT_SIGMOID && N.act != pk::ACT_SIN && N.act != pk::ACT_MIXED) return fail(who + "this activation is not covered by the float64 mode");
            if (N.act == pk::ACT_MIXED && (int)N.sizes.size() - 2 > 8) return fail(who + "per-layer tanh / sigmoid mixes of more than 8 hidden layers are not covered by the float64 mode");
            if ((int)N.sizes.size() - 1 > pk::F64_MAX_LAYERS) return fail(who + "more than 16 Dense layers");
            if (N.sizes[0] != E.nets[nets[0]].sizes[0]) return fail(who + "its dependent variables take different numbers of arguments (one jet set serves all networks of an equation in the float64 mode)");
            any_sin = any_sin || N.act == pk::ACT_SIN;
            all_sin = all_sin && N.act == pk::ACT_SIN;
        }
        if (any_sin && !all_sin) return fail(who + "mixes sin networks with tanh / sigmoid networks");
        // (periodic input embeddings, r06: the descriptor has rewritten the term already — sin / cos feature rows behind the user's coordinates, the
        // network mapped onto them, derivative slots as chain-rule combinations (descriptor.cpp: apply_embeddings); this mode fills the feature rows in DOUBLE)
        if (T.emb_cols.size() > 4) return fail(who + "more than 4 embedded feature rows");
        if (T.d > 4 || E.nets[nets[0]].sizes[0] > 4) return fail(who + "more than 4 coordinates");
        if ((int)T.slots.size() > pk::F64_MAX_SLOTS || T.d + E.np + (int)T.slots.size() + (int)T.ops.size() > pk::F64_MAX_ROWS)
            return fail(who + "residual expression too long for the float64 tape (96 rows)");
        F.ndata = E.terms[t].ndata;                      // (OP_DATA rows stay in this mode's tape: the float path hoists them into source channels)
        std::string why;
        F.k = f64_find(E.nets[nets[0]].sizes[0], T.slots, F.slot_chan, why);
        if (!F.k) return fail(who + why);
        F.km = f64_find_m(E, F.k, nets);
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
        f64_affine(E, T, F);
        if (f64_convert_points(E, F, E.terms[t])) return 1;
        if (f64_install_data(E, F, E.terms[t], nullptr)) return 1;
    }
    const int K = (int)E.terms.size();
    // (+ F64S_PAD_THETA zeroed doubles behind theta: the sliced matrix-pipe kernels read full 16 HT-wide fragments of a narrower layer's matrix)
    S->d_theta = (double*)plat_malloc(sizeof(double) * (E.ntheta + pk::F64S_PAD_THETA));
    if (S->d_theta) plat_memset(S->d_theta, 0, sizeof(double) * (E.ntheta + pk::F64S_PAD_THETA), E.stream);
    S->d_grad = (double*)plat_malloc(sizeof(double) * (E.ntheta + K));      // [gradient (P) | sums (K)] contiguous: one all-reduce over a communicator
    S->d_sumsq = S->d_grad ? S->d_grad + E.ntheta : nullptr;
    if (!S->d_theta || !S->d_grad) return fail("device allocation failed (float64 state)");
    S->h_pin = (double*)plat_host_alloc(sizeof(double) * (2 * (size_t)E.ntheta + K));
    if (!S->h_pin) return fail("pinned host allocation failed (float64 staging)");
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
    if (T.emb_cols.empty()) {
        if (plat_h2d(F.d_pts, pts, sizeof(double) * (size_t)n * T.d, E.stream) || plat_sync(E.stream)) return fail("H2D copy of points failed");
    } else {                                             // the caller's [n][d_user] coordinates + the feature rows
        std::vector<double> hd((size_t)n * T.d);
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < T.d_user; ++j) hd[(size_t)i * T.d + j] = pts[(size_t)i * T.d_user + j];
        f64_embed_host(T, hd, n);
        if (plat_h2d(F.d_pts, hd.data(), sizeof(double) * hd.size(), E.stream) || plat_sync(E.stream)) return fail("H2D copy of points failed");
    }
    F.n = n;
    F.exact_pts = true;
    return f64_lin_install(E, F, T);
}

// pinn_set_point_data while the mode is on: the double rows follow (data == nullptr: converted from the float rows just installed)
int f64_set_point_data(pinn_engine& E, int term, const double* data) {
    if (!E.f64) return 0;
    F64State& S = *(F64State*)E.f64;
    return f64_install_data(E, S.terms[term], E.terms[term], data);
}

// ---- one term (or pseudo-term: a network's trial function / derivative at caller-supplied points) as the kernels see it ----
struct F64Launch {
    pk::F64Args a;
    int rows = 0;                        // scratch rows per point
    bool sin_act = false, mfma = false;
    bool sliced = false;                 // family 4s: the tile kernel passes every layer through the scratch rows (needed in every mode)
};
// everything of F64Args that does not depend on the evaluation (theta, weights, mode, chunk): networks, scratch row numbering, tape, slots
static int f64_build(pinn_engine& E, const F64Term& F, int dt, const std::map<int, std::vector<int>>* inmap, F64Launch& L) {
    pk::F64Args& a = L.a;
    std::memset(&a, 0, sizeof a);
    a.data = F.ndata > 0 ? F.d_data : nullptr;
    a.lin = f64_lin_ptr(F);
    a.pts = F.d_pts;
    a.N = (int)F.n; a.dt = dt;
    a.nnets = (int)F.nets.size();
    a.C = F.k->C;
    for (int i = 0; i < 8; ++i) a.first_ch[i] = F.k->first_ch[i];
    int rows = 0, ent = 0;
    L.sin_act = false;
    L.mfma = F.km && std::getenv("PINN_F64_NO_MFMA") == nullptr;
    L.sliced = L.mfma && F.km->sliced != 0;
    a.use_ceff = L.sliced ? 1 : 0;
    for (int ni = 0; ni < a.nnets; ++ni) {
        const Net& N = E.nets[F.nets[ni]];
        pk::F64Net& n = a.net[ni];
        n.d = N.sizes[0];
        std::vector<int> m;
        if (inmap && inmap->count(F.nets[ni])) m = inmap->at(F.nets[ni]);
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
        n.act_layers = N.act_layers;
        n.ceff = 1;
        for (int sl = 0; sl < F.nslots; ++sl) if (F.slot_net[sl] == ni) n.ceff = std::max(n.ceff, F.slot_chan[sl] + 1);
        if (std::getenv("PINN_F64_FULL_CHANNELS")) n.ceff = a.C;              // (A/B, tests)
        L.sin_act = L.sin_act || N.act == pk::ACT_SIN;
        n.theta0 = N.theta_off; n.nparams = N.nparams(); n.ent0 = ent;
        ent += n.nparams;
        n.tp0 = a.tp_p;                                       // (running column count of the tile partial sums)
        a.tp_p += (n.d + 1) * N.sizes[1] + N.sizes[n.nl - 1] + 1;
        // scratch rows: per hidden layer record / post-activation jets / dZ, then this network's seeds
        const int LH = n.nl - 1;
        for (int l = 0; l < LH; ++l) { n.r_rec[l] = rows; rows += N.sizes[l + 1] * a.C; }
        // value-only terms of tanh / sigmoid networks: the record of an element IS its post-activation value (act_record), so the
        // matrix-pipe kernels keep ONE copy (a third of the tile kernel's stores less; 4 of the bench workload's 5 terms)
        const bool post_is_rec = L.mfma && a.C == 1 && N.act != pk::ACT_SIN;
        a.post_alias = post_is_rec ? 1 : 0;
        for (int l = 0; l < LH; ++l) { if (post_is_rec) n.r_post[l] = n.r_rec[l]; else { n.r_post[l] = rows; rows += N.sizes[l + 1] * a.C; } }
        for (int l = 0; l < LH; ++l) { n.r_dz[l] = rows; rows += N.sizes[l + 1] * a.C; }
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
    a.nrows = rows;
    L.rows = rows;
    if (L.mfma) {
        a.ntp = a.tp_p + E.ne + 1;
        a.tile_pts = 16 * F.km->PG;
    }
    return 0;
}
static bool grow(double*& buf, size_t& cap, size_t need, plat_stream st) {
    if (need <= cap) return true;
    plat_sync(st);
    plat_free(buf);
    buf = (double*)plat_malloc(sizeof(double) * need);
    cap = buf ? need : 0;
    return buf != nullptr;
}
// points per launch and the buffers of one: scratch rows (below 256 MB with one lane per point / 4 GB with the matrix-pipe kernels — one wave per
// 16-32 points there, a chunk should hold several tiles per SIMD: the bench workload's 65,536-point terms are one chunk each; $PINN_F64_SCRATCH_MB
// overrides), per-block slabs, per-tile partial sums.  An allocation that fails is retried with half the chunk down to one block.
// with_sums == false (values-only launches, a.mode == 2): no slabs / tile sums
static int f64_buffers(pinn_engine& E, F64State& S, F64Launch& L, int64_t n, bool with_sums, int64_t& chunk_out) {
    pk::F64Args& a = L.a;
    double mb = L.mfma ? 4096.0 : 256.0;
    if (const char* e = std::getenv("PINN_F64_SCRATCH_MB")) mb = std::max(1.0, std::atof(e));
    const bool need_scratch = !(L.mfma && a.mode != 0) || L.sliced;              // (the tile kernels keep everything in registers unless a reverse sweep / dW launch follows)
    for (;; mb *= 0.5) {
        int64_t chunk = (int64_t)((mb * 1024 * 1024) / (8.0 * L.rows));
        chunk = std::max<int64_t>(pk::F64_BLOCK, (chunk / pk::F64_BLOCK) * pk::F64_BLOCK);
        chunk = std::min<int64_t>(chunk, ((n + pk::F64_BLOCK - 1) / pk::F64_BLOCK) * pk::F64_BLOCK);
        const bool last_try = chunk <= pk::F64_BLOCK;
        bool ok = true;
        if (need_scratch) ok = grow(S.d_scratch, S.scratch_cap, (size_t)L.rows * (size_t)chunk + pk::F64S_PAD_SCRATCH, E.stream);
        if (ok && with_sums) ok = grow(S.d_slab, S.slab_cap, (size_t)(chunk / pk::F64_BLOCK) * (size_t)a.nent, E.stream);
        if (ok && with_sums && L.mfma) ok = grow(S.d_tpart, S.tpart_cap, (size_t)(chunk / a.tile_pts + 1) * (size_t)a.ntp, E.stream);
        if (ok) {
            a.scratch = S.d_scratch; a.npad = (int)chunk; a.slab = S.d_slab; a.tpart = S.d_tpart;
            chunk_out = chunk;
            return 0;
        }
        if (last_try) return fail("device allocation failed (float64 scratch / slabs, even for one 512-point block)");
    }
}

// Small launches of the register-resident matrix-pipe kernels: SHORT blocks for the dW kernel.  One 512-point block = one workgroup per layer, whose four
// waves walk the block's 16-point groups x channels one dependent load + MFMA step after the other: 15 us for 176 points x 5 channels.  With short
// blocks the same steps spread over many workgroups (the slab gets one row per block, the reduction adds a few more rows).  Returns the block count.
constexpr int F64_SMALL_BLOCK = 64;
static int f64_short_blocks(pinn_engine& E, F64State& S, F64Launch& L) {
    pk::F64Args& a = L.a;
    a.block_pts = 0;
    const int full = (a.npts + pk::F64_BLOCK - 1) / pk::F64_BLOCK;
    if (!L.mfma || L.sliced || a.tile_pts <= 0 || F64_SMALL_BLOCK % a.tile_pts != 0 || std::getenv("PINN_F64_NO_SHORT_BLOCKS")) return full;
    // the shortest power-of-two block (>= 64 points) that still gives the launch ~512 workgroups: blocks x (hidden-to-hidden layers + 1)
    int rows = 1;
    for (int ni = 0; ni < a.nnets; ++ni) rows += std::max(0, a.net[ni].nl - 2);
    int bp = pk::F64_BLOCK;
    static const int target = std::getenv("PINN_F64_DWT_WGS") ? std::max(1, std::atoi(std::getenv("PINN_F64_DWT_WGS"))) : 512;      // (A/B knob)
    while (bp > F64_SMALL_BLOCK && (int64_t)((a.npts + bp - 1) / bp) * rows < target) bp >>= 1;
    if (bp == pk::F64_BLOCK) return full;
    const int nb = (a.npts + bp - 1) / bp;
    if (!grow(S.d_slab, S.slab_cap, (size_t)nb * (size_t)a.nent, E.stream)) return full;
    a.slab = S.d_slab;
    a.block_pts = bp;
    return nb;
}

// ---- the reference-semantics ("stencil") validation mode ----
// get_eps (src/symbolic_utilities.jl:98-103): eps(Float64)^(1 / (2 + order)), the TOTAL order of the derivative on every axis (:185)
static double stencil_eps(int order) { return std::pow(2.220446049250313e-16, 1.0 / (2.0 + (double)order)); }
namespace {
struct StOperand { int kind, idx; };     // 0: virtual network idx, 1: stencil op idx, 2: an original row below the slots (coordinate / parameter)
struct StBuilder {
    std::vector<F64VNet> vnets;
    struct Op { int code; StOperand a, b; double imm; };
    std::vector<Op> ops;
    int vnet(int net, const std::vector<std::pair<int, double>>& sh) {
        for (size_t i = 0; i < vnets.size(); ++i) if (vnets[i].net == net && vnets[i].shifts == sh) return (int)i;
        vnets.push_back({net, sh});
        return (int)vnets.size() - 1;
    }
    StOperand op(int code, StOperand a, StOperand b, double imm) { ops.push_back({code, a, b, imm}); return {1, (int)ops.size() - 1}; }
    StOperand mulc(StOperand a, double c) { return op(rp::OP_MULC, a, {2, 0}, c); }
    StOperand u(int net, std::vector<std::pair<int, double>> sh, int axis, double delta) {
        sh.push_back({axis, delta});
        return {0, vnet(net, sh)};
    }
    // numeric_derivative(phi, u, x, eps_s, order, theta) (src/pinn_types.jl:445-482) at the points shifted by `sh`: axes[0 .. k) = the derivative
    // variables, eps = the step (one magnitude for all axes: the total order's); the formulas in the reference's own order of operations
    StOperand deriv(int net, const int* axes, int k, double eps, const std::vector<std::pair<int, double>>& sh) {
        if (k == 0) return {0, vnet(net, sh)};
        const double inv = 1.0 / eps;                    // _epsilon = inv(first(eps[eps .!= 0]))
        const int ax = axes[k - 1];                      // eps = eps_s[order]
        bool same = true;
        for (int i = 1; i < k; ++i) same = same && axes[i] == axes[0];
        if (k > 4 || !same) {                            // :454-460
            auto up = sh, dn = sh;
            up.push_back({ax, eps}); dn.push_back({ax, -eps});
            const StOperand a = deriv(net, axes, k - 1, eps, up), b = deriv(net, axes, k - 1, eps, dn);
            return mulc(mulc(op(rp::OP_SUB, a, b, 0.0), inv), 0.5);
        }
        if (k == 4) {                                    // :461-468: (u(x+2e) - 4u(x+e) + 6u(x) - 4u(x-e) + u(x-2e)) * _epsilon^4
            StOperand t = op(rp::OP_SUB, u(net, sh, ax, 2.0 * eps), mulc(u(net, sh, ax, eps), 4.0), 0.0);
            t = op(rp::OP_ADD, t, mulc({0, vnet(net, sh)}, 6.0), 0.0);
            t = op(rp::OP_SUB, t, mulc(u(net, sh, ax, -eps), 4.0), 0.0);
            t = op(rp::OP_ADD, t, u(net, sh, ax, -2.0 * eps), 0.0);
            return mulc(t, (inv * inv) * (inv * inv));
        }
        if (k == 3) {                                    // :469-474: (u(x+2e) - 2u(x+e) + 2u(x-e) - u(x-2e)) * _epsilon^3 / 2
            StOperand t = op(rp::OP_SUB, u(net, sh, ax, 2.0 * eps), mulc(u(net, sh, ax, eps), 2.0), 0.0);
            t = op(rp::OP_ADD, t, mulc(u(net, sh, ax, -eps), 2.0), 0.0);
            t = op(rp::OP_SUB, t, u(net, sh, ax, -2.0 * eps), 0.0);
            return mulc(mulc(t, inv * inv * inv), 0.5);
        }
        if (k == 2) {                                    // :475-476: (u(x+e) + u(x-e) - 2u(x)) * _epsilon^2
            StOperand t = op(rp::OP_ADD, u(net, sh, ax, eps), u(net, sh, ax, -eps), 0.0);
            t = op(rp::OP_SUB, t, mulc({0, vnet(net, sh)}, 2.0), 0.0);
            return mulc(t, inv * inv);
        }
        return mulc(mulc(op(rp::OP_SUB, u(net, sh, ax, eps), u(net, sh, ax, -eps), 0.0), inv), 0.5);      // :477-478
    }
};
}  // namespace

// build every term's stencil tape (pinn_set_option(h, "derivative", "stencil"); the float64 mode must be on)
int f64_stencil_enable(pinn_engine& E, bool on) {
    if (!E.f64) return fail("pinn_set_option: \"derivative\" = \"stencil\" is a validation mode of the float64 evaluation (set \"precision\" to \"f64\" first)");
    F64State& S = *(F64State*)E.f64;
    if (!on) { S.stencil = false; return 0; }
    if (!S.sten.empty()) { S.stencil = true; return 0; }
    std::vector<F64Stencil> sten(E.terms0.size());
    struct Guard {                                       // (a failure below must not leak the tapes already uploaded)
        std::vector<F64Stencil>& v; bool armed = true;
        ~Guard() { if (armed) for (auto& X : v) { plat_free(X.d_prog); plat_free(X.d_imm); } }
    } guard{sten};
    for (size_t t = 0; t < E.terms0.size(); ++t) {
        const Term& T = E.terms0[t];
        F64Stencil& X = sten[t];
        const std::string who = "derivative = stencil: term " + std::to_string(t) + ": ";
        for (auto& sl : T.slots) X.active = X.active || sl.order > 0 || sl.lap != 0;
        if (!X.active) continue;
        if (!T.emb_cols.empty()) return fail(who + "the term reads a network behind a periodic input embedding: its derivative slots are derivatives with respect to the sin / cos FEATURES (descriptor rewrite), central differences of which are not the reference's differences in the coordinates");
        StBuilder B;
        std::vector<StOperand> slot_val;
        for (auto& sl : T.slots) {
            if (sl.lap) return fail(who + "internal (a fused Laplacian slot in the pristine term)");
            slot_val.push_back(B.deriv(sl.net, sl.axes, sl.order, stencil_eps(sl.order), {}));
        }
        const int dt = T.d, np = E.np, nv = (int)B.vnets.size(), nold = (int)T.slots.size(), npre = (int)B.ops.size();
        if (nv > pk::F64_MAX_SLOTS || dt + np + nv + npre + (int)T.ops.size() > pk::F64_MAX_ROWS)
            return fail(who + "the stencil tape needs " + std::to_string(dt + np + nv + npre + (int)T.ops.size()) + " rows / " + std::to_string(nv) + " shifted evaluations (limits: 96 / 24)");
        for (auto& vn : B.vnets)
            if (vn.shifts.size() > 8) return fail(who + "derivative order above 8");
        auto row = [&](StOperand o) { return o.kind == 0 ? dt + np + o.idx : (o.kind == 1 ? dt + np + nv + o.idx : o.idx); };
        std::vector<rp::Instr> prog;
        std::vector<double> imm;
        for (auto& o : B.ops) {
            rp::Instr I;
            std::memset(&I, 0, sizeof I);
            I.code = o.code; I.a = row(o.a); I.b = rp::is_binary(o.code) ? row(o.b) : 0; I.imm = (float)o.imm;
            rp::finalize(I);
            prog.push_back(I); imm.push_back(o.imm);
        }
        auto remap = [&](int r) {                        // a row of the term's own numbering [coordinates | params | slots | ops]
            if (r < dt + np) return r;
            if (r < dt + np + nold) return row(slot_val[r - dt - np]);
            return dt + np + nv + npre + (r - dt - np - nold);
        };
        for (size_t q = 0; q < T.ops.size(); ++q) {
            rp::Instr I = T.ops[q];
            if (!rp::is_nullary(I.code)) I.a = remap(I.a);
            if (rp::is_binary(I.code)) I.b = remap(I.b);
            prog.push_back(I);
            imm.push_back(q < T.imm64.size() ? T.imm64[q] : (double)T.ops[q].imm);
        }
        X.vnets = B.vnets;
        X.nops = (int)prog.size();
        X.out_row = remap(T.out_row);
        X.d_prog = (rp::Instr*)plat_malloc(sizeof(rp::Instr) * std::max<size_t>(prog.size(), 1));
        X.d_imm = (double*)plat_malloc(sizeof(double) * std::max<size_t>(prog.size(), 1));
        if (!X.d_prog || !X.d_imm) return fail("device allocation failed (stencil tape)");
        plat_h2d(X.d_prog, prog.data(), sizeof(rp::Instr) * prog.size(), E.stream);
        plat_h2d(X.d_imm, imm.data(), sizeof(double) * imm.size(), E.stream);
        if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    guard.armed = false;
    S.sten.swap(sten);
    S.stencil = true;
    return 0;
}
bool f64_stencil_on(const pinn_engine& E) { return E.f64 && ((const F64State*)E.f64)->stencil; }

static int f64_values(pinn_engine& E, F64State& S, const F64Term& F, F64Launch& L, int64_t n, double* d_out);
// the value-only pseudo-term of virtual network `vn` of term t on its shifted set
static int stencil_vnet_term(pinn_engine& E, F64State& S, const F64VNet& vn, double* d_pts, int64_t n, F64Term& F) {
    const Net& N = E.nets[vn.net];
    std::string why;
    Slot sl;
    sl.net = vn.net; sl.order = 0; sl.lap = 0;
    for (int q = 0; q < MAX_DERIV_ORDER; ++q) sl.axes[q] = 0;
    F.k = f64_find(N.sizes[0], std::vector<Slot>{sl}, F.slot_chan, why);
    if (!F.k) return fail("derivative = stencil: " + why);
    F.km = nullptr;                                      // (one lane per point: a validation mode)
    F.nets = {vn.net};
    F.slot_net = {0};
    F.nops = 0; F.nslots = 1; F.out_row = N.sizes[0] + E.np;
    F.n = n;
    F.d_pts = d_pts;
    return 0;
}
// one term in stencil mode.  mode 0: loss + gradient (accumulated into `grad`, which the caller has zeroed), 1: loss only, 2: residuals into d_resid
static int f64_stencil_term(pinn_engine& E, int t, const double* theta, double* grad, double* sumsq_t, double w, int mode, double* d_resid) {
    F64State& S = *(F64State*)E.f64;
    const F64Stencil& X = S.sten[t];
    const F64Term& F0 = S.terms[t];
    const Term& T = E.terms[t];
    const Term& T0 = E.terms0[t];
    const int64_t n = F0.n;
    const int nv = (int)X.vnets.size(), stride = 2 + E.ne;
    int dmax = 1;
    for (auto& vn : X.vnets) dmax = std::max(dmax, E.nets[vn.net].sizes[0]);
    if (!grow(S.d_uv, S.uv_cap, (size_t)nv * n, E.stream) || !grow(S.d_seeds, S.seeds_cap, (size_t)nv * n * stride, E.stream) ||
        !grow(S.d_spts, S.spts_cap, (size_t)nv * n * dmax, E.stream)) return fail("device allocation failed (stencil evaluation)");
    // 1. shifted sets and the virtual networks' values
    for (int v = 0; v < nv; ++v) {
        const F64VNet& vn = X.vnets[v];
        const Net& N = E.nets[vn.net];
        pk::F64ShiftArgs sa;
        std::memset(&sa, 0, sizeof sa);
        sa.pts = F0.d_pts; sa.dt = T0.d; sa.out = S.d_spts + (size_t)v * n * dmax; sa.d = N.sizes[0]; sa.n = (int)n;
        std::vector<int> m;
        if (T0.inmap.count(vn.net)) m = T0.inmap.at(vn.net);
        else for (int i = 0; i < sa.d; ++i) m.push_back(i);
        for (int i = 0; i < 4; ++i) sa.imap[i] = i < (int)m.size() ? m[i] : 0;
        sa.nshift = (int)vn.shifts.size();
        for (int q = 0; q < sa.nshift; ++q) { sa.axis[q] = vn.shifts[q].first; sa.delta[q] = vn.shifts[q].second; }
        pk::launch_f64_shift(sa, E.stream);
        F64Term F;
        if (stencil_vnet_term(E, S, vn, sa.out, n, F)) return 1;
        F64Launch L;
        int rc = f64_build(E, F, sa.d, nullptr, L);
        L.a.theta = theta;
        if (!rc) rc = f64_values(E, S, F, L, n, S.d_uv + (size_t)v * n);
        F.d_pts = nullptr;
        if (rc) return rc;
    }
    // 2. the stencil tape: residuals / squared residuals / seeds
    pk::F64StapeArgs ta;
    std::memset(&ta, 0, sizeof ta);
    ta.pts = F0.d_pts; ta.dt = T0.d; ta.n = (int)n; ta.theta = theta;
    ta.np = E.np; ta.ne = E.ne; ta.p_off = E.p_theta_off;
    for (int j = 0; j < pk::MAX_PARAMS; ++j) ta.pdef[j] = j < (int)E.p_defaults.size() ? (double)E.p_defaults[j] : 0.0;
    ta.pw = (T.pw_n == T.n && T.pw_n > 0) ? T.d_pw : nullptr;
    ta.data = F0.ndata > 0 ? F0.d_data : nullptr;
    ta.uv = S.d_uv; ta.nv = nv;
    ta.prog = X.d_prog; ta.imm = X.d_imm; ta.nops = X.nops; ta.out_row = X.out_row;
    ta.scale = 2.0 * w / (double)T.n_norm;
    ta.mode = mode; ta.resid = d_resid; ta.seeds = S.d_seeds; ta.seed_stride = stride;
    pk::launch_f64_stape(ta, E.stream);
    if (mode == 2) return 0;
    // 3. seeded launches: the squared residuals ride in virtual network 0's records; loss only: that launch alone
    for (int v = 0; v < (mode == 1 ? 1 : nv); ++v) {
        const F64VNet& vn = X.vnets[v];
        F64Term F;
        if (stencil_vnet_term(E, S, vn, S.d_spts + (size_t)v * n * dmax, n, F)) return 1;
        F64Launch L;
        if (f64_build(E, F, E.nets[vn.net].sizes[0], nullptr, L)) { F.d_pts = nullptr; return 1; }
        pk::F64Args& a = L.a;
        a.theta = theta; a.mode = mode; a.scale = 0.0;
        a.seed = S.d_seeds + (size_t)v * n * stride; a.seed_stride = stride;
        int64_t chunk = 0;
        if (f64_buffers(E, S, L, n, true, chunk)) { F.d_pts = nullptr; return 1; }
        for (int64_t p0 = 0; p0 < n; p0 += chunk) {
            a.p0 = (int)p0;
            a.npts = (int)std::min<int64_t>(chunk, n - p0);
            F.k->launch_point(a, L.sin_act, E.stream);
            pk::launch_f64_dw(a, E.stream);
            pk::launch_f64_dwt(a, E.stream);
            pk::F64ReduceArgs r;
            std::memset(&r, 0, sizeof r);
            r.slab = S.d_slab; r.nblocks = (a.npts + pk::F64_BLOCK - 1) / pk::F64_BLOCK; r.nent = a.nent;
            r.grad = grad; r.ent_p = a.ent_p; r.p_off = E.p_theta_off; r.nnets = 1;
            r.ent0[0] = a.net[0].ent0; r.theta0[0] = a.net[0].theta0;
            r.sumsq = sumsq_t; r.nsq = 1; r.sq_off[0] = 0; r.with_grad = mode == 0 ? 1 : 0;
            r.init_sumsq = (v == 0 && p0 == 0) ? 1 : 0;
            r.init_grad = 0;
            pk::launch_f64_reduce(r, E.stream);
        }
        F.d_pts = nullptr;
    }
    return 0;
}

// the matrix-pipe tile kernel reads the per-term fields through F64Args::sub (pinn_kernels4.hpp: F64Sub): a plain launch mirrors its own fields into entry 0
static void f64_launch_tile(const pk::F64MKernel* km, pk::F64Args& a, plat_stream st) {
    if (a.nsub == 0) {
        pk::F64Sub& u = a.sub[0];
        u.pts = a.pts; u.pw = a.pw; u.data = a.data; u.prog = a.prog; u.imm = a.imm; u.lin = a.lin; u.scale = a.scale;
        u.N = a.N; u.nops = a.nops; u.out_row = a.out_row; u.nslots = a.nslots;
        for (int q = 0; q < pk::F64_MAX_SLOTS; ++q) { u.slot_net[q] = (unsigned char)a.slot_net[q]; u.slot_chan[q] = (unsigned char)a.slot_chan[q]; }
    }
    km->launch_tile(a, st);
}

// Which terms ride in one launch (see F64State::Group).  Small problems only: a merged launch evaluates every member on the UNION jet set (a value-only
// boundary term carries the interior term's derivative channels), which costs nothing while every tile of the problem is resident at once and the
// evaluation is bound by the latency of one tile per launch — and would multiply the boundary terms' work on a large problem.
constexpr int64_t F64_MERGE_MAX_POINTS = 8192;
static void f64_make_groups(pinn_engine& E, F64State& S) {
    const int K = (int)S.terms.size();
    std::vector<int64_t> sig(K);
    for (int t = 0; t < K; ++t) sig[t] = S.terms[t].n;
    if (sig == S.groups_sig) return;
    S.groups_sig = sig;
    S.groups.clear();
    if (std::getenv("PINN_F64_NO_MERGE") || std::getenv("PINN_F64_NO_MFMA")) return;
    std::vector<char> used(K, 0);
    for (int t = 0; t < K; ++t) {
        if (used[t]) continue;
        const Term& T0 = E.terms0[t];
        const F64Term& F = S.terms[t];
        if (!F.km || F.km->sliced || F.n <= 0) continue;
        F64State::Group G;
        std::vector<Slot> slots = T0.slots;
        int64_t pts = F.n;
        G.terms.push_back(t);
        G.nets = F.nets;
        G.inmap = T0.inmap;
        const int d = E.nets[F.nets[0]].sizes[0];
        bool free_merge = true;                          // every member so far runs its OWN kernel on its own networks
        for (int u = t + 1; u < K && (int)G.terms.size() < pk::F64_MAX_SUB; ++u) {
            const Term& U0 = E.terms0[u];
            const F64Term& Fu = S.terms[u];
            if (used[u] || !Fu.km || Fu.km->sliced || Fu.n <= 0) continue;
            if (U0.d != T0.d || E.nets[Fu.nets[0]].sizes[0] != d) continue;
            // the launch evaluates the UNION of the members' networks for every tile (a network a member does not read gets zero seeds from it):
            // systems of equations over different subsets of the dependent variables (the reference's Lorenz test) still ride in one sequence
            std::vector<int> nets = G.nets;
            for (int ni : Fu.nets) if (std::find(nets.begin(), nets.end(), ni) == nets.end()) nets.push_back(ni);
            std::sort(nets.begin(), nets.end());
            if ((int)nets.size() > pk::F64_MAX_NETS) continue;
            std::map<int, std::vector<int>> inmap = G.inmap;
            bool clash = false;
            for (int ni : nets) {                        // one input binding per network: explicit maps must agree, an identity binding must be one for both
                const bool a_has = G.inmap.count(ni) != 0, b_has = U0.inmap.count(ni) != 0;
                const bool a_uses = std::find(G.nets.begin(), G.nets.end(), ni) != G.nets.end(), b_uses = std::find(Fu.nets.begin(), Fu.nets.end(), ni) != Fu.nets.end();
                if (a_uses && b_uses && (a_has != b_has || (a_has && G.inmap.at(ni) != U0.inmap.at(ni)))) { clash = true; break; }
                if (b_uses && b_has) inmap[ni] = U0.inmap.at(ni);
            }
            if (clash) continue;
            std::vector<Slot> trial = slots;
            trial.insert(trial.end(), U0.slots.begin(), U0.slots.end());
            std::vector<int> ch;
            std::string why;
            const pk::F64Kernel* k = f64_find(d, trial, ch, why);
            const pk::F64MKernel* km = k ? f64_find_m(E, k, nets) : nullptr;
            if (!km || km->sliced) continue;
            // beyond the small-problem size a member joins only where it costs nothing: the same kernel and the same networks as every other member
            // (the four boundary terms of a 2-D problem: one dW / reduction launch instead of four)
            const bool same = free_merge && k == F.k && k == Fu.k && nets == F.nets && nets == Fu.nets;
            if (pts + Fu.n > F64_MERGE_MAX_POINTS && !same) continue;
            free_merge = same;
            slots.swap(trial);
            pts += Fu.n;
            G.terms.push_back(u);
            G.nets.swap(nets);
            G.inmap.swap(inmap);
        }
        if (G.terms.size() < 2 || (pts > F64_MERGE_MAX_POINTS && !free_merge)) continue;
        std::vector<int> ch;
        std::string why;
        G.k = f64_find(d, slots, ch, why);
        G.km = G.k ? f64_find_m(E, G.k, G.nets) : nullptr;
        if (!G.km || G.km->sliced) continue;
        size_t o = 0;
        for (int u : G.terms) {
            const size_t ns = E.terms0[u].slots.size();
            G.slot_chan.emplace_back(ch.begin() + o, ch.begin() + o + ns);
            o += ns;
            used[u] = 1;
        }
        S.groups.push_back(std::move(G));
    }
}

// loss sums (`sumsq`, K doubles) and gradient (`grad`, P doubles; nullptr: loss only) at the parameters `theta` — all three device pointers —
// everything on the device, nothing synchronised
static int f64_eval_device(pinn_engine& E, const double* theta, double* grad, double* sumsq, const double* term_w) {
    F64State& S = *(F64State*)E.f64;
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    if (theta != S.d_theta && theta != S.d_opt_theta) {
        // a CALLER's device buffer (pinn_loss_grad_device_f64): the matrix-pipe kernels read up to F64S_PAD_THETA doubles past a narrow layer's matrix —
        // evaluate from the handle's padded copy (P doubles device to device: microseconds)
        bool sliced = false;                             // (r06: family 4m reads unclamped weight fragments as well)
        for (auto& F : S.terms) sliced = sliced || F.km != nullptr;
        if (sliced) {
            if (plat_d2d(S.d_theta, theta, sizeof(double) * P, E.stream)) return fail("device copy of theta failed");
            theta = S.d_theta;
        }
    }
    for (int t = 0; t < K; ++t)
        if (S.terms[t].n <= 0 || S.terms[t].n != E.terms[t].n) return fail("term " + std::to_string(t) + " has no collocation points (call pinn_set_points first)");
    for (int t = 0; t < K; ++t)
        if (S.terms[t].ndata > 0 && S.terms[t].data_n != S.terms[t].n)
            return fail("term " + std::to_string(t) + " uses per-point data channels but none are installed for its current point set (call pinn_set_point_data after pinn_set_points)");
    // the first reduction of the evaluation writes the gradient instead of adding to it when its term's entries cover all of theta (one network,
    // or every network in the first equation); otherwise one memset.  Every term's first chunk writes its own sum of squares.
    bool grad_started = false;
    if (grad && S.stencil) { plat_memset(grad, 0, sizeof(double) * P, E.stream); grad_started = true; }
    // (called in front of the evaluation's FIRST reduction: 1 = that launch writes the gradient, 0 = the gradient was just zeroed and it adds)
    auto first_reduction_writes = [&](const std::vector<int>& nets) -> int {
        if (!grad || grad_started) return 0;
        grad_started = true;
        int64_t covered = E.ne;
        for (int ni : nets) covered += E.nets[ni].nparams();
        if (covered == P) return 1;
        plat_memset(grad, 0, sizeof(double) * P, E.stream);
        return 0;
    };
    S.path = 0;
    S.merged_launches = 0;
    std::vector<char> done(K, 0);
    if (!S.stencil) {
        f64_make_groups(E, S);
        for (const F64State::Group& G : S.groups) {
            // a pseudo-term on the union jet set; its networks, rows and slab entries are the members' (same networks)
            const int m = (int)G.terms.size(), t0 = G.terms[0];
            const Term& T0 = E.terms0[t0];
            F64Term Fg;
            Fg.k = G.k; Fg.km = G.km; Fg.nets = G.nets;
            Fg.nops = 0; Fg.nslots = 0; Fg.out_row = 0; Fg.ndata = 0;
            F64Launch L;
            if (f64_build(E, Fg, T0.d, &G.inmap, L)) return 1;
            if (!L.mfma || L.sliced) continue;
            pk::F64Args& a = L.a;
            const int tp = a.tile_pts;
            int tiles = 0;
            a.nsub = m;
            for (int q = 0; q < m; ++q) {
                const int t = G.terms[q];
                const Term& T = E.terms[t];
                const F64Term& F = S.terms[t];
                pk::F64Sub& u = a.sub[q];
                a.sub_tile0[q] = tiles;
                tiles += (int)((F.n + tp - 1) / tp);
                u.pts = F.d_pts; u.pw = (T.pw_n == T.n && T.pw_n > 0) ? T.d_pw : nullptr; u.data = F.ndata > 0 ? F.d_data : nullptr;
                u.prog = F.d_prog; u.imm = F.d_imm; u.lin = f64_lin_ptr(F);
                u.scale = 2.0 * (term_w ? term_w[t] : 1.0) / (double)T.n_norm;
                u.N = (int)F.n; u.nops = F.nops; u.out_row = F.out_row; u.nslots = F.nslots;
                for (int sl = 0; sl < F.nslots; ++sl) {          // (slot_net: position of the slot's network in the launch's — the union's — list)
                    u.slot_net[sl] = (unsigned char)(std::find(G.nets.begin(), G.nets.end(), F.nets[F.slot_net[sl]]) - G.nets.begin());
                    u.slot_chan[sl] = (unsigned char)G.slot_chan[q][sl];
                }
            }
            a.sub_tile0[m] = tiles;
            for (int q = m + 1; q <= pk::F64_MAX_SUB; ++q) a.sub_tile0[q] = tiles;
            a.nent += m - 1;                             // (the slab row ends in one sum of squares per member)
            a.theta = theta;
            a.mode = grad ? 0 : 1;
            a.p0 = 0;
            a.npts = tiles * tp;
            int64_t chunk = 0;
            if (f64_buffers(E, S, L, a.npts, true, chunk)) return 1;
            if (chunk < a.npts) continue;                // (does not fit one launch: the members go one by one below)
            S.path |= 2;
            ++S.merged_launches;
            const int nblocks = f64_short_blocks(E, S, L);
            f64_launch_tile(G.km, a, E.stream);
            G.km->launch_dwt(a, E.stream);
            pk::F64ReduceArgs r;
            std::memset(&r, 0, sizeof r);
            r.slab = S.d_slab; r.nblocks = nblocks; r.nent = a.nent;
            r.grad = grad; r.ent_p = a.ent_p; r.p_off = E.p_theta_off; r.nnets = a.nnets;
            for (int ni = 0; ni < a.nnets; ++ni) { r.ent0[ni] = a.net[ni].ent0; r.theta0[ni] = a.net[ni].theta0; }
            r.sumsq = sumsq; r.nsq = m;
            for (int q = 0; q < m; ++q) r.sq_off[q] = G.terms[q];
            r.with_grad = grad ? 1 : 0;
            r.init_sumsq = 1;
            r.init_grad = first_reduction_writes(Fg.nets);
            pk::launch_f64m_reduce(r, E.stream);
            for (int t : G.terms) done[t] = 1;
        }
    }
    for (int t = 0; t < K; ++t) {
        if (done[t]) continue;
        const Term& T = E.terms[t];
        const Term& T0 = E.terms0[t];
        F64Term& F = S.terms[t];
        if (S.stencil && S.sten[t].active) {             // reference semantics: central differences (a validation mode, one lane per point)
            S.path |= 1;
            if (f64_stencil_term(E, t, theta, grad, sumsq + t, term_w ? term_w[t] : 1.0, grad ? 0 : 1, nullptr)) return 1;
            continue;
        }
        F64Launch L;
        if (f64_build(E, F, T0.d, &T0.inmap, L)) return 1;
        pk::F64Args& a = L.a;
        a.theta = theta;
        a.pw = (T.pw_n == T.n && T.pw_n > 0) ? T.d_pw : nullptr;
        const double w = term_w ? term_w[t] : 1.0;
        a.scale = 2.0 * w / (double)T.n_norm;
        a.mode = grad ? 0 : 1;
        S.path |= L.mfma ? 2 : 1;
        int64_t chunk = 0;
        if (f64_buffers(E, S, L, F.n, true, chunk)) return 1;
        for (int64_t p0 = 0; p0 < F.n; p0 += chunk) {
            a.p0 = (int)p0;
            a.npts = (int)std::min<int64_t>(chunk, F.n - p0);
            const int nblocks = f64_short_blocks(E, S, L);
            if (L.mfma) f64_launch_tile(F.km, a, E.stream);
            else F.k->launch_point(a, L.sin_act, E.stream);
            if (!L.mfma) pk::launch_f64_dw(a, E.stream);         // (matrix-pipe path: those entries come out of the tile kernel, summed in the dW launch)
            if (L.mfma) F.km->launch_dwt(a, E.stream);
            else pk::launch_f64_dwt(a, E.stream);
            pk::F64ReduceArgs r;
            std::memset(&r, 0, sizeof r);
            r.slab = S.d_slab; r.nblocks = nblocks; r.nent = a.nent;
            r.grad = grad; r.ent_p = a.ent_p; r.p_off = E.p_theta_off; r.nnets = a.nnets;
            for (int ni = 0; ni < a.nnets; ++ni) { r.ent0[ni] = a.net[ni].ent0; r.theta0[ni] = a.net[ni].theta0; }
            r.sumsq = sumsq + t; r.nsq = 1; r.sq_off[0] = 0; r.with_grad = grad ? 1 : 0;
            r.init_sumsq = (p0 == 0) ? 1 : 0;
            r.init_grad = first_reduction_writes(F.nets);
            if (L.mfma) pk::launch_f64m_reduce(r, E.stream);
            else pk::launch_f64_reduce(r, E.stream);
        }
    }
    return 0;
}

int f64_eval(pinn_engine& E, const double* theta, const double* term_w, double* term_losses, double* grad) {
    F64State& S = *(F64State*)E.f64;
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    // (pinned staging both ways: a pageable source / destination makes every copy a synchronous, internally staged one — the host entry of a small problem
    // is mostly those copies; gradient and sums are one device vector, so one copy brings both)
    std::memcpy(S.h_pin, theta, sizeof(double) * P);
    if (plat_h2d(S.d_theta, S.h_pin, sizeof(double) * P, E.stream)) return fail("H2D copy of theta failed");
    if (f64_eval_device(E, S.d_theta, grad ? S.d_grad : nullptr, S.d_sumsq, term_w)) return 1;
    double* out = S.h_pin + P;
    if (grad ? plat_d2h(out, S.d_grad, sizeof(double) * (P + K), E.stream) : plat_d2h(out + P, S.d_sumsq, sizeof(double) * K, E.stream)) return fail("D2H copy failed");
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    if (term_losses)
        for (int k = 0; k < K; ++k) term_losses[k] = out[(size_t)P + k] / (double)E.terms[k].n_norm;
    if (grad) std::memcpy(grad, out, sizeof(double) * P);
    return 0;
}

// ---- the public closures that return PER-POINT values, in double (r06): the reference evaluates them in eltype(theta) = Float64 by default —
// datafree_pde_loss_functions[i](cord, theta) (src/pinn_types.jl:435-439, compared at rtol 1e-8 in test/Forward/forward__ode.jl:46-47),
// phi(x, theta) (src/pinn_types.jl:88-90), numeric_derivative (src/pinn_types.jl:445-482; atol 1e-8 in test/Forward/forward__derivatives.jl:29-30) ----
// values-only launches (F64Args::mode == 2) of `L` over n points: out[p] on the device
static int f64_values(pinn_engine& E, F64State& S, const F64Term& F, F64Launch& L, int64_t n, double* d_out) {
    pk::F64Args& a = L.a;
    a.mode = 2;
    a.resid = d_out;
    a.scale = 0.0;
    S.path |= L.mfma ? 2 : 1;
    int64_t chunk = 0;
    if (f64_buffers(E, S, L, n, false, chunk)) return 1;
    for (int64_t p0 = 0; p0 < n; p0 += chunk) {
        a.p0 = (int)p0;
        a.npts = (int)std::min<int64_t>(chunk, n - p0);
        if (L.mfma) f64_launch_tile(F.km, a, E.stream);
        else F.k->launch_point(a, L.sin_act, E.stream);
    }
    return 0;
}
// residual_k(set_k, theta) at every point of the installed set (pinn_residual_f64; pinn_residual in float64 mode)
int f64_residual(pinn_engine& E, int term, const double* theta, double* r) {
    F64State& S = *(F64State*)E.f64;
    F64Term& F = S.terms[term];
    if (F.n <= 0 || F.n != E.terms[term].n) return fail("pinn_residual: term has no points");
    if (F.ndata > 0 && F.data_n != F.n) return fail("pinn_residual: the term uses per-point data channels but none are installed for its current point set");
    if (plat_h2d(S.d_theta, theta, sizeof(double) * E.ntheta, E.stream)) return fail("H2D copy of theta failed");
    size_t cap = (size_t)S.aux_out_cap;
    if (!grow(S.d_aux_out, cap, (size_t)F.n, E.stream)) { S.aux_out_cap = 0; return fail("device allocation failed (float64 residuals)"); }
    S.aux_out_cap = (int64_t)cap;
    S.path = 0;
    if (S.stencil && S.sten[term].active) {
        S.path = 1;
        if (f64_stencil_term(E, term, S.d_theta, nullptr, nullptr, 1.0, 2, S.d_aux_out)) return 1;
    } else {
        F64Launch L;
        if (f64_build(E, F, E.terms0[term].d, &E.terms0[term].inmap, L)) return 1;
        L.a.theta = S.d_theta;
        if (f64_values(E, S, F, L, F.n, S.d_aux_out)) return 1;
    }
    if (plat_d2h(r, S.d_aux_out, sizeof(double) * (size_t)F.n, E.stream) || plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    return 0;
}
// d^order phi_net / dx_axes (order 0: the trial function itself) at n caller-supplied points (pinn_phi_f64, pinn_derivative_f64 and their float
// counterparts in float64 mode): a pseudo-term of one slot and an empty tape on the jet set that carries the derivative
int f64_net_eval(pinn_engine& E, int net, const double* theta, const double* pts, int64_t n, int order, const int* axes, double* out) {
    F64State& S = *(F64State*)E.f64;
    const Net& N = E.nets[net];
    const std::string who = "float64 trial-function evaluation: ";
    if (N.kind != 0) return fail(who + "DGM networks are not covered by the float64 mode");
    if (N.act != pk::ACT_TANH && N.act != pk::ACT_SIGMOID && N.act != pk::ACT_SIN && N.act != pk::ACT_MIXED) return fail(who + "this activation is not covered by the float64 mode");
    if ((int)N.sizes.size() - 1 > pk::F64_MAX_LAYERS || N.sizes[0] > 4) return fail(who + "more than 16 Dense layers / more than 4 inputs");
    const int d = N.sizes[0];
    // a network behind a periodic input embedding (r06): the caller's points are the dependent variable's ARGUMENTS [n][n_inputs]; the Dense chain takes
    // the features (pass-through arguments, then sin, then cos of the embedded ones — engine_types.hpp: Net::emb_idx), formed here in double
    std::vector<double> feats;
    if (!N.emb_idx.empty()) {
        if (order > 0) return fail("pinn_derivative: not available for a network behind a periodic input embedding (use pinn_residual on a term that carries the derivative)");
        const int nin = N.n_inputs(), ne = (int)N.emb_idx.size();
        feats.resize((size_t)n * d);
        for (int64_t q = 0; q < n; ++q) {
            int pass = 0;
            for (int a = 0; a < nin; ++a) {
                const double x = pts[(size_t)q * nin + a];
                const auto it = std::find(N.emb_idx.begin(), N.emb_idx.end(), a);
                if (it == N.emb_idx.end()) { feats[(size_t)q * d + pass++] = x; continue; }
                const int k = (int)(it - N.emb_idx.begin());
                const double ph = 6.283185307179586476925286766559 / N.emb_period[k] * x;
                feats[(size_t)q * d + nin - ne + k] = std::sin(ph);
                feats[(size_t)q * d + nin + k] = std::cos(ph);
            }
        }
        pts = feats.data();
    }
    Slot sl;
    sl.net = net; sl.order = order; sl.lap = 0;
    for (int q = 0; q < MAX_DERIV_ORDER; ++q) sl.axes[q] = q < order ? axes[q] : 0;
    std::sort(sl.axes, sl.axes + order);
    for (int q = 0; q < order; ++q)
        if (sl.axes[q] < 0 || sl.axes[q] >= d) return fail("pinn_derivative: axis out of range");
    F64Term F;
    std::string why;
    F.k = f64_find(d, std::vector<Slot>{sl}, F.slot_chan, why);
    if (!F.k) return fail(who + why);
    F.nets = {net};
    F.km = f64_find_m(E, F.k, F.nets);
    F.slot_net = {0};
    F.nops = 0; F.nslots = 1; F.out_row = d + E.np;
    F.n = n;
    size_t cp = (size_t)S.aux_pts_cap, co = (size_t)S.aux_out_cap;
    const bool okp = grow(S.d_aux_pts, cp, (size_t)n * d, E.stream), oko = grow(S.d_aux_out, co, (size_t)n, E.stream);
    S.aux_pts_cap = (int64_t)cp; S.aux_out_cap = (int64_t)co;
    if (!okp || !oko) return fail("device allocation failed (float64 trial-function points)");
    F.d_pts = S.d_aux_pts;
    if (plat_h2d(S.d_theta, theta, sizeof(double) * E.ntheta, E.stream) || plat_h2d(S.d_aux_pts, pts, sizeof(double) * (size_t)n * d, E.stream)) {
        F.d_pts = nullptr;
        return fail("H2D copy failed");
    }
    F64Launch L;
    int rc = f64_build(E, F, d, nullptr, L);
    L.a.theta = S.d_theta;
    S.path = 0;
    if (!rc) rc = f64_values(E, S, F, L, n, S.d_aux_out);
    if (!rc && (plat_d2h(out, S.d_aux_out, sizeof(double) * (size_t)n, E.stream) || plat_sync(E.stream))) rc = fail(std::string("device error: ") + plat_last_error());
    F.d_pts = nullptr;                                   // (borrowed: the pseudo-term owns nothing)
    return rc;
}

// ---- the resident optimiser loop in float64 (pinn_adam_* on a handle in float64 mode; r05): theta, moments, points and every kernel of the
// iteration in double on the device — redraw (the fp32 samplers' points, converted on the device: the reference's StochasticTraining /
// QuasiRandomTraining(resampling = true) with Float64 parameters, src/training_strategies.jl:271-282, 365-389) -> evaluate -> Adam.
// The iterate has its OWN buffer (d_opt_theta): evaluations between two pinn_adam_steps calls (adaptive reweighting, callbacks) do not touch it ----
int f64_adam_init(pinn_engine& E, const double* theta) {
    F64State& S = *(F64State*)E.f64;
    const int64_t P = E.ntheta;
    const int K = (int)E.terms.size();
    if (!S.d_m) {
        S.d_m = (double*)plat_malloc(sizeof(double) * P);
        S.d_v = (double*)plat_malloc(sizeof(double) * P);
        S.d_opt_theta = (double*)plat_malloc(sizeof(double) * (P + pk::F64S_PAD_THETA));
        if (S.d_opt_theta) plat_memset(S.d_opt_theta, 0, sizeof(double) * (P + pk::F64S_PAD_THETA), E.stream);
        S.d_w_over_n = (double*)plat_malloc(sizeof(double) * K);
        if (!S.d_m || !S.d_v || !S.d_w_over_n || !S.d_opt_theta) return fail("device allocation failed (float64 optimiser state)");
    }
    if (plat_h2d(S.d_opt_theta, theta, sizeof(double) * P, E.stream)) return fail("H2D copy of theta failed");
    plat_memset(S.d_m, 0, sizeof(double) * P, E.stream);
    plat_memset(S.d_v, 0, sizeof(double) * P, E.stream);
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    S.opt_ready = true;
    E.opt_t = 0;
    return 0;
}
int f64_adam_get(pinn_engine& E, double* theta) {
    F64State& S = *(F64State*)E.f64;
    if (!S.opt_ready) return fail("pinn_adam_get: no float64 optimiser state (call pinn_adam_init after switching to precision f64)");
    if (plat_d2h(theta, S.d_opt_theta, sizeof(double) * E.ntheta, E.stream) || plat_sync(E.stream)) return fail("D2H copy failed");
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
    if (T.emb_cols.empty()) pk::launch_f64_cvt(T.d_pts, F.d_pts, (int64_t)T.n * T.d, E.stream);
    else {                                               // the draw lives in d_upts [n][d_user] (the float feature rows are not refreshed by this loop): widened, features in double
        pk::F64EmbedArgs ea;
        ea.pts = F.d_pts; ea.upts = T.d_upts; ea.n = (int)T.n; ea.d = T.d; ea.du = T.d_user; ea.ncols = (int)T.emb_cols.size();
        for (int k = 0; k < ea.ncols; ++k) { ea.src[k] = T.emb_cols[k].src; ea.is_cos[k] = T.emb_cols[k].is_cos; ea.omega[k] = T.emb_cols[k].omega; }
        pk::launch_f64_embed(ea, E.stream);
    }
    F.n = T.n;
    F.exact_pts = false;
    F.lin_n = 0;                                         // (a redrawn set: the affine fast path's per-point part is not recomputed inside the loop — the tape runs)
    F.data_n = 0;                                        // (per-point data belong to the previous set: a sampled term with DATA channels fails its next evaluation with the message — ADVICE r05)
    return 0;
}
int f64_adam_steps(pinn_engine& E, int nsteps, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss_history, void (*redraw)(pinn_engine&, Term&)) {
    pinn_engine* es[1] = {&E};
    return f64_adam_steps_comm(es, 1, nsteps, lr, beta1, beta2, eps, term_w, loss_history, redraw);
}
// this device's part of a single-process multi-device evaluation (pinn_loss_grad_sharded_f64): theta from the host, [gradient | sums] stay on the device
int f64_eval_sharded_local(pinn_engine& E, const double* theta, const double* term_w, double** d_out) {
    F64State& S = *(F64State*)E.f64;
    if (plat_h2d(S.d_theta, theta, sizeof(double) * E.ntheta, E.stream)) return fail("H2D copy of theta failed");
    if (f64_eval_device(E, S.d_theta, S.d_grad, S.d_sumsq, term_w)) return 1;
    *d_out = S.d_grad;
    return 0;
}
// the resident float64 Adam loop over a communicator (r06): per iteration every rank / device evaluates its shards, ONE all-reduce of [P + K] doubles,
// the identical update on every rank.  ndev == 1: a one-process-per-GPU communicator (or none); ndev > 1: the handles of a pinn_comm_init_all communicator
int f64_adam_steps_comm(pinn_engine** es, int ndev, int nsteps, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss_history,
                        void (*redraw)(pinn_engine&, Term&)) {
    const int K = (int)es[0]->terms.size();
    const int P = (int)es[0]->ntheta;
    std::vector<double> w(K), won(K);
    for (int k = 0; k < K; ++k) { w[k] = term_w ? (double)term_w[k] : 1.0; won[k] = w[k] / (double)es[0]->terms[k].n_norm; }
    std::vector<double*> vec(ndev);
    for (int i = 0; i < ndev; ++i) {
        pinn_engine& E = *es[i];
        DeviceScope scope(E.device);
        F64State& S = *(F64State*)E.f64;
        if (!S.opt_ready) return fail("pinn_adam_steps: call pinn_adam_init first (float64 mode keeps its own optimiser state)");
        plat_h2d(S.d_w_over_n, won.data(), sizeof(double) * K, E.stream);
        if (S.hist_cap < nsteps) {
            plat_sync(E.stream);
            plat_free(S.d_hist);
            S.d_hist = (double*)plat_malloc(sizeof(double) * nsteps);
            S.hist_cap = S.d_hist ? nsteps : 0;
            if (!S.d_hist) return fail("device allocation failed (loss history)");
        }
        vec[i] = S.d_grad;
    }
    const bool collective = ndev > 1 || es[0]->comm != nullptr;
    for (int s = 0; s < nsteps; ++s) {
        for (int i = 0; i < ndev; ++i) {
            pinn_engine& E = *es[i];
            DeviceScope scope(E.device);
            F64State& S = *(F64State*)E.f64;
            for (size_t t = 0; t < E.terms.size(); ++t) {
                Term& T = E.terms[t];
                if (T.sampler == 0) continue;
                redraw(E, T);                                    // (engine.cpp: the fp32 sampler kernels, rank-specific seeds, draw counter advanced)
                if (f64_points_from_device(E, (int)t)) return 1;
            }
            if (f64_eval_device(E, S.d_opt_theta, S.d_grad, S.d_sumsq, w.data())) return 1;
        }
        if (collective && comm_all_reduce_f64(es, ndev, vec.data(), (int64_t)P + K)) return 1;
        for (int i = 0; i < ndev; ++i) {
            pinn_engine& E = *es[i];
            DeviceScope scope(E.device);
            F64State& S = *(F64State*)E.f64;
            ++E.opt_t;
            const double c1 = 1.0 / (1.0 - std::pow(beta1, (double)E.opt_t)), c2 = 1.0 / (1.0 - std::pow(beta2, (double)E.opt_t));
            pk::launch_f64_adam_total(S.d_opt_theta, S.d_m, S.d_v, S.d_grad, P, lr, beta1, beta2, eps, c1, c2, S.d_hist, s, S.d_sumsq, S.d_w_over_n, K, E.stream);
        }
    }
    {
        pinn_engine& E0 = *es[0];
        DeviceScope scope(E0.device);
        F64State& S0 = *(F64State*)E0.f64;
        if (loss_history && plat_d2h(loss_history, S0.d_hist, sizeof(double) * nsteps, E0.stream)) return fail("D2H copy failed");
    }
    for (int i = 0; i < ndev; ++i) {
        DeviceScope scope(es[i]->device);
        if (plat_sync(es[i]->stream)) return fail(std::string("device error: ") + plat_last_error());
    }
    return 0;
}
// one Adam update of the float64 state from a caller-supplied host vector [gradient (P) | raw per-term sums (K)] (pinn_adam_apply in float64 mode)
int f64_adam_apply(pinn_engine& E, const double* grad_and_sums, double lr, double beta1, double beta2, double eps, const float* term_w, double* loss) {
    F64State& S = *(F64State*)E.f64;
    if (!S.opt_ready) return fail("pinn_adam_apply: call pinn_adam_init first (float64 mode keeps its own optimiser state)");
    const int K = (int)E.terms.size();
    const int P = (int)E.ntheta;
    if (plat_h2d(S.d_grad, grad_and_sums, sizeof(double) * P, E.stream)) return fail("H2D copy failed");
    ++E.opt_t;
    const double c1 = 1.0 / (1.0 - std::pow(beta1, (double)E.opt_t)), c2 = 1.0 / (1.0 - std::pow(beta2, (double)E.opt_t));
    pk::launch_f64_adam(S.d_opt_theta, S.d_m, S.d_v, S.d_grad, P, lr, beta1, beta2, eps, c1, c2, E.stream);
    if (plat_sync(E.stream)) return fail(std::string("device error: ") + plat_last_error());
    if (loss) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += (term_w ? (double)term_w[k] : 1.0) * grad_and_sums[P + k] / (double)E.terms[k].n_norm;
        *loss = s;
    }
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
    if (f64_eval_device(E, S.d_theta, want_grad ? S.d_grad : nullptr, S.d_sumsq, w.data())) return 1;
    if (want_grad) pk::launch_f64_narrow(S.d_grad, d_out, P, E.stream);
    pk::launch_f64_narrow(S.d_sumsq, d_out + (want_grad ? P : 0), K, E.stream);
    return 0;
}

// theta and [gradient | raw sums] as DOUBLE device pointers: nothing crosses the boundary in fp32, nothing crosses PCIe; the kernels read the
// caller's theta and write the caller's [gradient | sums] directly: no copies
int f64_eval_from_device_f64(pinn_engine& E, const double* d_theta, const float* term_w, double* d_out) {
    const int K = (int)E.terms.size();
    const int64_t P = E.ntheta;
    std::vector<double> w(K);
    for (int k = 0; k < K; ++k) w[k] = term_w ? (double)term_w[k] : 1.0;
    return f64_eval_device(E, d_theta, d_out, d_out + P, w.data());
}

// " f64_channels=5,1,1,1,1 f64_kernels=mfma:HT4xPG1,mfma:HT4xPG4,..." for pinn_describe
std::string f64_describe(const pinn_engine& E) {
    if (!E.f64) return "";
    const F64State& S = *(const F64State*)E.f64;
    std::string c = " f64_channels=", k = " f64_kernels=";
    for (size_t t = 0; t < S.terms.size(); ++t) {
        const F64Term& F = S.terms[t];
        c += (t ? "," : "") + std::to_string(F.k ? F.k->C : 0);
        k += (t ? "," : "") + (F.km ? std::string(F.km->sliced ? "mfma-sliced:HT" : "mfma:HT") + std::to_string(F.km->HT) + "xPG" + std::to_string(F.km->PG) : std::string("lanes"));
    }
    return c + k;
}

int f64_merged(const pinn_engine& E) { return E.f64 ? ((const F64State*)E.f64)->merged_launches : 0; }
int f64_affine_terms(const pinn_engine& E) {
    if (!E.f64) return 0;
    int n = 0;
    for (auto& F : ((const F64State*)E.f64)->terms) n += f64_lin_ptr(F) != nullptr && F.km != nullptr;
    return n;
}
const char* f64_path(const pinn_engine& E) {
    if (!E.f64) return "off";
    const int p = ((const F64State*)E.f64)->path;
    return p == 2 ? "mfma" : (p == 1 ? "lanes" : (p == 3 ? "mfma+lanes" : "none"));
}

}  // namespace pe
