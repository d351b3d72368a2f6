// descriptor.cpp — parser of the "pinnir 1" problem descriptor (grammar: DESIGN.md §2).
#include "engine_types.hpp"
#include <sstream>

namespace pe {

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
    if (!expect("pinnir") || !(in >> ver) || (ver != 1 && ver != 2)) return fail("descriptor: expected 'pinnir 1' or 'pinnir 2'");
    SexprContext ctx;                          // pinnir 2: names of the parameters / dependent variables and their inputs
    if (!expect("ntheta") || !(in >> E.ntheta)) return fail("descriptor: ntheta");
    if (!expect("params") || !(in >> E.np >> E.ne >> E.p_theta_off)) return fail("descriptor: params");
    if (E.np < 0 || E.np > pk::MAX_PARAMS || E.ne > E.np) return fail("descriptor: at most 4 PDE parameters are supported");
    if (!expect("defaults")) return fail("descriptor: defaults");
    E.p_defaults.assign(pk::MAX_PARAMS, 0.f);
    for (int i = 0; i < E.np; ++i)
        if (!(in >> E.p_defaults[i])) return fail("descriptor: defaults values");
    if (ver == 2) {                            // pnames <np names>
        if (!expect("pnames")) return fail("descriptor: pnames");
        ctx.params.resize(E.np);
        for (int i = 0; i < E.np; ++i)
            if (!(in >> ctx.params[i])) return fail("descriptor: pnames values");
    }
    int nn = 0;
    if (!expect("nets") || !(in >> nn) || nn < 1) return fail("descriptor: nets");
    ctx.depvars.resize(ver == 2 ? nn : 0);
    ctx.depvar_inputs.resize(ver == 2 ? nn : 0);
    E.nets.resize(nn);
    for (int i = 0; i < nn; ++i) {
        int id, ns;
        std::string act;
        if (!expect("net") || !(in >> id >> act >> E.nets[i].theta_off >> ns) || id != i) return fail("descriptor: net line");
        // one activation for all hidden layers, or a comma-separated list with one entry per hidden layer: tanh and sigmoid may be mixed
        // (e.g. the reference's Dense(1, n, tanh), Dense(n, n, sigma), Dense(n, 1)); sin only on all layers
        std::vector<int> kinds;
        bool dgm = false;
        {
            std::stringstream as(act);
            std::string tok;
            while (std::getline(as, tok, ',')) {
                if (tok == "dgm" && kinds.empty() && !dgm) { dgm = true; continue; }      // dgm,<activation1>,<activation2>,<layers>
                if (dgm && kinds.size() == 2) {
                    E.nets[i].dgm_layers = std::atoi(tok.c_str());
                    if (E.nets[i].dgm_layers < 1 || E.nets[i].dgm_layers > 8) return fail("descriptor: a DGM network has 1..8 gated layers");
                    continue;
                }
                if (tok == "tanh") kinds.push_back(pk::ACT_TANH);
                else if (tok == "sigmoid") kinds.push_back(pk::ACT_SIGMOID);
                else if (tok == "sin") kinds.push_back(pk::ACT_SIN);
                else return fail("descriptor: unsupported activation '" + tok + "' (supported: tanh, sigmoid, sin)");
            }
        }
        E.nets[i].sizes.resize(ns);
        for (int j = 0; j < ns; ++j)
            if (!(in >> E.nets[i].sizes[j])) return fail("descriptor: net sizes");
        {                                      // optional: embed <i> <n> <input index> <period> ... (PeriodicEmbedding in front of the chain)
            const std::streampos pos = in.tellg();
            std::string t2;
            if ((in >> t2) && t2 == "embed") {
                int eid, ne;
                if (!(in >> eid >> ne) || eid != i || ne < 1 || ne > 4) return fail("descriptor: embed line");
                Net& N = E.nets[i];
                for (int k = 0; k < ne; ++k) {
                    int ix; double per;
                    if (!(in >> ix >> per) || !(per > 0.0)) return fail("descriptor: embed line (input index and a positive period per embedded input)");
                    if (ix < 0 || ix >= N.sizes[0] - ne) return fail("descriptor: embed input index out of range");
                    if (std::find(N.emb_idx.begin(), N.emb_idx.end(), ix) != N.emb_idx.end()) return fail("descriptor: embed lists an input twice");
                    N.emb_idx.push_back(ix); N.emb_period.push_back(per);
                }
                if (dgm) return fail("descriptor: a periodic embedding in front of a DGM network is not supported");
            } else {
                in.clear();
                in.seekg(pos);
            }
        }
        if (ver == 2) {                        // netvar <i> <depvar name> <#inputs> <input names...>  (dict_depvar_input, symbolic_utilities.jl:401-426)
            int vid, nin;
            if (!expect("netvar") || !(in >> vid >> ctx.depvars[i] >> nin) || vid != i || nin != E.nets[i].n_inputs())
                return fail("descriptor: netvar line (one per net: name and as many input names as the chain has inputs)");
            ctx.depvar_inputs[i].resize(nin);
            for (int j = 0; j < nin; ++j)
                if (!(in >> ctx.depvar_inputs[i][j])) return fail("descriptor: netvar input names");
        }
        if (dgm) {
            // the reference's DGM(in_dims, 1, modes, layers, activation1, activation2, identity) (src/dgm.jl:97-115): sizes = d modes 1
            if (kinds.size() != 2 || E.nets[i].dgm_layers < 1 || ns != 3 || E.nets[i].sizes[2] != 1)
                return fail("descriptor: a DGM net line reads `net <i> dgm,<activation1>,<activation2>,<layers> <theta_off> 3 <d> <modes> 1`");
            if (E.nets[i].sizes[1] > 64) return fail("descriptor: DGM networks are supported up to 64 modes");
            E.nets[i].kind = 1;
            E.nets[i].act = kinds[0];
            E.nets[i].act2 = kinds[1];
            E.nets[i].act_layers = 0;
            continue;
        }
        if (ns < 3) return fail("descriptor: a chain needs at least one hidden layer");
        if (E.nets[i].sizes.back() != 1) return fail("descriptor: only single-output chains (one per dependent variable) are supported, as in the reference (pinn_types.jl:106-108)");
        const int nhidden = ns - 2;
        if (kinds.size() == 1) kinds.assign((size_t)nhidden, kinds[0]);
        if ((int)kinds.size() != nhidden) return fail("descriptor: the activation list of net " + std::to_string(i) + " needs one entry per hidden layer");
        bool uniform = true;
        for (int l = 1; l < nhidden; ++l) uniform = uniform && kinds[l] == kinds[0];
        E.nets[i].act = kinds[0];
        E.nets[i].act_layers = 0;
        if (!uniform) {
            if (nhidden > 8) return fail("descriptor: per-layer activations cover at most 8 hidden layers");
            for (int l = 0; l < nhidden; ++l) {
                if (kinds[l] == pk::ACT_SIN) return fail("descriptor: sin cannot be mixed with other activations inside one chain");
                E.nets[i].act_layers |= kinds[l] << (4 * l);
            }
            E.nets[i].act = pk::ACT_MIXED;
        }
    }
    int nt = 0;
    if (!expect("terms") || !(in >> nt) || nt < 1) return fail("descriptor: terms");
    E.terms.resize(nt);
    for (int i = 0; i < nt; ++i) {
        Term& T = E.terms[i];
        int id, ns, no;
        if (!(in >> tok)) return fail("descriptor: term line");
        if (ver == 2 && tok == "sterm") {
            // sterm <k> <d> <coordinate names>  /  lhs <s-expression>  /  rhs <s-expression>: lowered by sexpr.cpp
            int d;
            if (!(in >> id >> d) || id != i || d < 1 || d > 4) return fail("descriptor: sterm line");
            std::vector<std::string> iv(d);
            for (int j = 0; j < d; ++j)
                if (!(in >> iv[j])) return fail("descriptor: sterm coordinate names");
            std::string lhs, rhs;
            if (!expect("lhs") || !std::getline(in, lhs)) return fail("descriptor: lhs line");
            if (!expect("rhs") || !std::getline(in, rhs)) return fail("descriptor: rhs line");
            if (lower_sexpr_term(ctx, iv, lhs, rhs, E.np, T)) return fail("term " + std::to_string(i) + ": " + g_err);
            for (auto& S : T.slots) {
                if (S.order == 2 && S.axes[0] > S.axes[1]) std::swap(S.axes[0], S.axes[1]);
            }
            continue;
        }
        if (tok != "term" || !(in >> id >> T.d >> ns >> no >> T.out_row) || id != i) return fail("descriptor: term line");
        T.slots.resize(ns);
        for (int s = 0; s < ns; ++s) {
            Slot& S = T.slots[s];
            std::string ord;
            if (!expect("slot") || !(in >> S.net >> ord)) return fail("descriptor: slot line");
            if (S.net < 0 || S.net >= nn) return fail("descriptor: slot net id");
            if (ord == "lap") {                      // slot <net> lap <n> a0 a1 ... : sum of d2/dx_a^2 over the listed axes
                int n = 0;
                if (!(in >> n) || n < 1 || n > 8) return fail("descriptor: lap slot");
                S.order = 2; for (int a = 0; a < MAX_DERIV_ORDER; ++a) S.axes[a] = 0;
                for (int a = 0; a < n; ++a) {
                    int ax;
                    if (!(in >> ax) || ax < 0 || ax > 7) return fail("descriptor: lap slot axes");
                    S.lap |= 1u << ax;
                }
                continue;
            }
            S.order = std::atoi(ord.c_str());
            if (ord.empty() || ord.find_first_not_of("0123456789") != std::string::npos) return fail("descriptor: slot order");
            if (S.order < 0 || S.order > MAX_DERIV_ORDER) return fail("derivative order > 6 is not supported by the HIP engine");
            for (int a = 0; a < S.order; ++a)
                if (!(in >> S.axes[a])) return fail("descriptor: slot axes");
            std::sort(S.axes, S.axes + S.order);
        }
        T.ops.resize(no);
        T.imm64.clear();
        for (int q = 0; q < no; ++q) {
            std::string name;
            rp::Instr& I = T.ops[q];
            double immd = 0.0;
            if (!expect("op") || !(in >> name >> I.a >> I.b >> immd)) return fail("descriptor: op line");
            I.imm = (float)immd;
            T.imm64.push_back(immd);
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
    // optional, after the terms: `hint <term> <points>` — how many points the caller is going to install for the term.  The planner uses it
    // to let SMALL value-only terms (a boundary condition at one or two points) ride on a launch the network already has instead of paying a
    // launch of their own (plan.cpp: plan_assign_terms); without hints every channel set gets its own launch group.
    for (;;) {
        std::string t2;
        if (!(in >> t2)) break;
        long long id = -1, n = 0;
        if (t2 != "hint" || !(in >> id >> n) || id < 0 || id >= nt || n < 0) return fail("descriptor: trailing text after the last term (expected `hint <term> <points>`)");
        E.terms[id].hint_n = n;
    }
    return apply_embeddings(E);
}

// ---------------------------------------------------------------------------------------------
// periodic input embeddings: u(x) = N(f(x)) with features f = [x_other..., sin(w x_p)..., cos(w x_p)...]
// ---------------------------------------------------------------------------------------------
// The kernels know chains of Dense layers over their inputs only.  A term that references an embedded network is rewritten so that it
// fits that model: the term's point rows grow by one sin and one cos row per embedded coordinate (filled on the device whenever the point
// set changes), the network's inputs are mapped onto those rows, and every derivative slot of u with respect to the ORIGINAL arguments
// becomes the chain-rule combination of derivatives of N with respect to its FEATURES:
//     d/dx_p    =  w c d_s - w s d_c
//     d2/dx_p2  =  w^2 (c^2 d_ss - 2 s c d_sc + s^2 d_cc - s d_s - c d_c)        (s = sin(w x_p), c = cos(w x_p), w = 2 pi / period)
// (operators of different coordinates commute, so mixed derivatives are products of these).  The coefficient products reference
// coordinate rows only: the planner hoists them into the per-point-set source pass like any other coordinate-only subexpression.
namespace {
struct ExpTerm { double k; std::vector<int> rows; std::vector<int> faxes; };      // k * prod(rows) * d^|faxes| N / d f_faxes
}
int apply_embeddings(pinn_engine& E) {
    bool any = false;
    for (auto& N : E.nets) any = any || !N.emb_idx.empty();
    for (auto& T : E.terms) T.d_user = T.d;
    if (!any) return 0;
    const double TWO_PI = 6.283185307179586476925286766559;
    for (size_t ti = 0; ti < E.terms.size(); ++ti) {
        Term& T = E.terms[ti];
        std::vector<int> nets;
        for (auto& s : T.slots)
            if (std::find(nets.begin(), nets.end(), s.net) == nets.end()) nets.push_back(s.net);
        bool touched = false;
        for (int n : nets) touched = touched || !E.nets[n].emb_idx.empty();
        if (!touched) continue;
        const int d0 = T.d, np = E.np, S0 = (int)T.slots.size();
        // user-space input maps (explicit for every referenced network from here on: the row count of the term changes)
        for (int n : nets) {
            const Net& N = E.nets[n];
            if (!T.inmap.count(n)) {
                if (N.n_inputs() != d0)
                    return fail("term " + std::to_string(ti) + ": network " + std::to_string(n) + " takes " + std::to_string(N.n_inputs()) +
                                " arguments but the term binds " + std::to_string(d0) + " coordinates and the descriptor has no inmap line for it");
                std::vector<int> id(d0);
                for (int i = 0; i < d0; ++i) id[i] = i;
                T.inmap[n] = id;
            }
            if ((int)T.inmap[n].size() != N.n_inputs())
                return fail("term " + std::to_string(ti) + ": inmap length differs from the argument count of network " + std::to_string(n));
        }
        auto col_row = [&](int src, double omega, int is_cos) -> int {
            for (size_t i = 0; i < T.emb_cols.size(); ++i)
                if (T.emb_cols[i].src == src && T.emb_cols[i].omega == omega && T.emb_cols[i].is_cos == is_cos) return d0 + (int)i;
            T.emb_cols.push_back({src, omega, is_cos});
            return d0 + (int)T.emb_cols.size() - 1;
        };
        // feature maps: per embedded network, feature index / rows of every argument
        struct ArgInfo { int feat = -1, fs = -1, fc = -1, rs = -1, rc = -1; double w = 0.0; };
        std::map<int, std::vector<ArgInfo>> arg;
        std::map<int, std::vector<int>> new_inmap;
        for (int n : nets) {
            const Net& N = E.nets[n];
            if (N.emb_idx.empty()) continue;
            const int nin = N.n_inputs(), ne = (int)N.emb_idx.size();
            std::vector<ArgInfo> A(nin);
            std::vector<int> m(N.sizes[0]);
            int pass = 0;
            for (int a = 0; a < nin; ++a) {
                const auto it = std::find(N.emb_idx.begin(), N.emb_idx.end(), a);
                if (it == N.emb_idx.end()) { A[a].feat = pass; m[pass] = T.inmap[n][a]; ++pass; continue; }
                const int k = (int)(it - N.emb_idx.begin());
                A[a].w = TWO_PI / N.emb_period[k];
                A[a].fs = nin - ne + k; A[a].fc = nin + k;
                A[a].rs = col_row(T.inmap[n][a], A[a].w, 0);
                A[a].rc = col_row(T.inmap[n][a], A[a].w, 1);
            }
            for (int a = 0; a < nin; ++a)
                if (A[a].feat < 0) { m[A[a].fs] = A[a].rs; m[A[a].fc] = A[a].rc; }
            arg[n] = A;
            new_inmap[n] = m;
        }
        const int dx = d0 + (int)T.emb_cols.size();
        if (dx > 4) return fail("term " + std::to_string(ti) + ": coordinates plus periodic-embedding rows exceed 4 (" + std::to_string(dx) + ")");
        // expansion of every slot
        std::vector<std::vector<ExpTerm>> expn(S0);
        std::vector<Slot> nslots;
        auto slot_index = [&](int net, std::vector<int> fa, unsigned lap) -> int {
            std::sort(fa.begin(), fa.end());
            for (size_t i = 0; i < nslots.size(); ++i) {
                const Slot& s = nslots[i];
                if (s.net != net || s.lap != lap || s.order != (int)fa.size()) continue;
                bool eq = true;
                for (int a = 0; a < s.order; ++a) eq = eq && s.axes[a] == fa[a];
                if (eq) return (int)i;
            }
            Slot s; s.net = net; s.lap = lap; s.order = lap ? 2 : (int)fa.size();
            for (int a = 0; a < MAX_DERIV_ORDER; ++a) s.axes[a] = (!lap && a < (int)fa.size()) ? fa[a] : 0;
            nslots.push_back(s);
            return (int)nslots.size() - 1;
        };
        auto expand_axes = [&](int net, const std::vector<int>& axes, std::vector<ExpTerm>& out) -> int {      // product of per-argument operators
            const auto& A = arg[net];
            std::vector<ExpTerm> cur{{1.0, {}, {}}};
            std::map<int, int> mult;
            for (int a : axes) {
                if (a < 0 || a >= (int)A.size()) return fail("descriptor: slot axis out of range");
                ++mult[a];
            }
            for (auto& kv : mult) {
                const ArgInfo& I = A[kv.first];
                std::vector<ExpTerm> op;
                if (I.feat >= 0) op.push_back({1.0, {}, std::vector<int>((size_t)kv.second, I.feat)});
                else if (kv.second == 1) { op.push_back({I.w, {I.rc}, {I.fs}}); op.push_back({-I.w, {I.rs}, {I.fc}}); }
                else if (kv.second == 2) {
                    const double w2 = I.w * I.w;
                    op.push_back({w2, {I.rc, I.rc}, {I.fs, I.fs}}); op.push_back({-2.0 * w2, {I.rs, I.rc}, {I.fs, I.fc}});
                    op.push_back({w2, {I.rs, I.rs}, {I.fc, I.fc}}); op.push_back({-w2, {I.rs}, {I.fs}}); op.push_back({-w2, {I.rc}, {I.fc}});
                } else return fail("term " + std::to_string(ti) + ": derivatives of order > 2 in a periodically embedded coordinate are not supported");
                std::vector<ExpTerm> nxt;
                for (auto& x : cur)
                    for (auto& y : op) {
                        ExpTerm z{x.k * y.k, x.rows, x.faxes};
                        z.rows.insert(z.rows.end(), y.rows.begin(), y.rows.end());
                        z.faxes.insert(z.faxes.end(), y.faxes.begin(), y.faxes.end());
                        nxt.push_back(z);
                    }
                cur.swap(nxt);
            }
            out.insert(out.end(), cur.begin(), cur.end());
            return 0;
        };
        for (int s = 0; s < S0; ++s) {
            const Slot& sl = T.slots[s];
            if (E.nets[sl.net].emb_idx.empty()) {
                std::vector<int> fa(sl.axes, sl.axes + (sl.lap ? 0 : sl.order));
                ExpTerm e{1.0, {}, fa};
                expn[s].push_back(e);
                continue;
            }
            if (sl.lap) {
                for (int a = 0; a < 8; ++a)
                    if ((sl.lap >> a) & 1u)
                        if (expand_axes(sl.net, {a, a}, expn[s])) return 1;
            } else if (expand_axes(sl.net, std::vector<int>(sl.axes, sl.axes + sl.order), expn[s])) return 1;
            for (auto& e : expn[s])
                if ((int)e.faxes.size() > MAX_DERIV_ORDER) return fail("derivative order > 6 is not supported by the HIP engine");
        }
        // new slot list (lap slots of plain networks keep their one-channel form)
        std::vector<std::vector<int>> eslot(S0);
        for (int s = 0; s < S0; ++s)
            for (auto& e : expn[s]) {
                const Slot& sl = T.slots[s];
                const bool plain_lap = E.nets[sl.net].emb_idx.empty() && sl.lap;
                eslot[s].push_back(slot_index(sl.net, e.faxes, plain_lap ? sl.lap : 0u));
            }
        const int S1 = (int)nslots.size();
        const int base_ops = dx + np + S1;
        std::vector<rp::Instr> nops;
        auto emit = [&](int code, int a, int b, float imm) -> int {
            rp::Instr I; I.code = code; I.a = a; I.b = b; I.imm = imm; rp::finalize(I);
            nops.push_back(I);
            return base_ops + (int)nops.size() - 1;
        };
        std::vector<int> slot_row(S0);
        for (int s = 0; s < S0; ++s) {
            int acc = -1;
            for (size_t i = 0; i < expn[s].size(); ++i) {
                const ExpTerm& e = expn[s][i];
                int r = dx + np + eslot[s][i];
                if (!e.rows.empty()) {
                    int c = e.rows[0];
                    for (size_t j = 1; j < e.rows.size(); ++j) c = emit(rp::OP_MUL, c, e.rows[j], 0.f);
                    if (e.k != 1.0) c = emit(rp::OP_MULC, c, 0, (float)e.k);
                    r = emit(rp::OP_MUL, c, r, 0.f);
                } else if (e.k != 1.0) r = emit(rp::OP_MULC, r, 0, (float)e.k);
                acc = acc < 0 ? r : emit(rp::OP_ADD, acc, r, 0.f);
            }
            slot_row[s] = acc;
        }
        const int nexp = (int)nops.size();
        auto remap = [&](int r) -> int {
            if (r < d0) return r;
            if (r < d0 + np) return r - d0 + dx;
            if (r < d0 + np + S0) return slot_row[r - d0 - np];
            return r - (d0 + np + S0) + base_ops + nexp;
        };
        for (auto I : T.ops) {
            if (!rp::is_nullary(I.code)) I.a = remap(I.a);
            if (rp::is_binary(I.code)) I.b = remap(I.b);
            nops.push_back(I);
        }
        T.out_row = remap(T.out_row);
        T.ops.swap(nops);
        T.slots.swap(nslots);
        T.d = dx;
        for (auto& kv : new_inmap) T.inmap[kv.first] = kv.second;
    }
    return 0;
}

}  // namespace pe
