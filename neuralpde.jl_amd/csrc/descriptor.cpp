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
        if (ver == 2) {                        // netvar <i> <depvar name> <#inputs> <input names...>  (dict_depvar_input, symbolic_utilities.jl:401-426)
            int vid, nin;
            if (!expect("netvar") || !(in >> vid >> ctx.depvars[i] >> nin) || vid != i || nin != E.nets[i].sizes[0])
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
    return 0;
}

}  // namespace pe
