// sexpr.cpp — the SYMBOLIC front end of the descriptor ("pinnir 2"): every equation / boundary condition arrives as the pair of
// expressions the reference itself walks — `toexpr(expand_derivatives(eq.lhs))` and `toexpr(expand_derivatives(eq.rhs))`
// (src/symbolic_utilities.jl:360-370) — printed as prefix s-expressions, and is lowered HERE to jet slots + the SSA tape of rprog.hpp.
// This is the engine's restatement of `_transform_expression` (src/symbolic_utilities.jl:132-331):
//   * a dependent-variable call `(u x y)` / `(u 0 y)` becomes the value slot of its network — the call arguments are DROPPED exactly
//     as the reference does (:145-160: boundary values come from the point set, not from the expression);
//   * (nested) `Differential`s `(D x 2 (u x y))`, `(D x 1 (D y 1 (u x y)))` collapse into one jet slot (network, sorted axes), the
//     `derivative(phi, u, cord, eps, order, theta)` call of :161-202 (evaluated exactly by Taylor jets instead of central differences);
//   * everything else is the closed op set of SURVEY.md App. B; residual = lhs - rhs (:365-369).
// A host binding therefore needs no lowering logic of its own: the Julia glue (julia/NeuralPDEHIP.jl) only PRINTS the reference's
// Expr trees, and the Python mirror prints sympy trees (neuralpde.jl_amd/sexpr.py); both are lowered by this one implementation.
//
// Grammar:  expr := number | pi | symbol | ( head expr* )      head := depvar | D | + - * / ^ | function name
//           (D <variable> <order> expr)   order a positive integer literal
#include "engine_types.hpp"

#include <cctype>
#include <cmath>
#include <map>

namespace pe {

namespace {

struct Node {
    bool atom = true;
    std::string text;                 // atom text, or head of a list
    std::vector<Node> args;
    std::string key() const {
        if (atom) return text;
        std::string s = "(" + text;
        for (auto& a : args) s += " " + a.key();
        return s + ")";
    }
};

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    int depth = 0;
    static constexpr int MAX_DEPTH = 200;            // nesting of the reference's expression trees is a few dozen at most
    explicit Parser(const std::string& src) : s(src) {}
    void skip() { while (i < s.size() && std::isspace((unsigned char)s[i])) ++i; }
    bool parse(Node& out) {
        struct Guard { int& d; explicit Guard(int& x) : d(x) { ++d; } ~Guard() { --d; } } guard(depth);
        if (depth > MAX_DEPTH) { err = "expression nested deeper than " + std::to_string(MAX_DEPTH) + " levels"; return false; }
        skip();
        if (i >= s.size()) { err = "unexpected end of expression"; return false; }
        if (s[i] == ')') { err = "unexpected ')'"; return false; }
        if (s[i] != '(') {
            size_t j = i;
            while (j < s.size() && !std::isspace((unsigned char)s[j]) && s[j] != '(' && s[j] != ')') ++j;
            out.atom = true;
            out.text = s.substr(i, j - i);
            i = j;
            return true;
        }
        ++i;
        skip();
        size_t j = i;
        while (j < s.size() && !std::isspace((unsigned char)s[j]) && s[j] != '(' && s[j] != ')') ++j;
        if (j == i) { err = "a list needs a head symbol"; return false; }
        out.atom = false;
        out.text = s.substr(i, j - i);
        i = j;
        for (;;) {
            skip();
            if (i >= s.size()) { err = "missing ')'"; return false; }
            if (s[i] == ')') { ++i; return true; }
            Node a;
            if (!parse(a)) return false;
            out.args.push_back(std::move(a));
        }
    }
};

// A decimal literal and nothing else: [+-] digits [. digits] [e|E|f [+-] digits] (Julia prints Float32 exponents with `f`).  strtod alone
// would also take "inf", "nan", "infinity" and hex floats, i.e. silently fold a variable or parameter of such a name into a constant.
static bool decimal_literal(const std::string& t, double& v) {
    size_t i = 0, nd = 0;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) ++i;
    while (i < t.size() && std::isdigit((unsigned char)t[i])) { ++i; ++nd; }
    if (i < t.size() && t[i] == '.') { ++i; while (i < t.size() && std::isdigit((unsigned char)t[i])) { ++i; ++nd; } }
    if (nd == 0) return false;
    std::string u = t;
    if (i < t.size() && (t[i] == 'e' || t[i] == 'E' || t[i] == 'f')) {
        u[i] = 'e';
        ++i;
        if (i < t.size() && (t[i] == '+' || t[i] == '-')) ++i;
        size_t ne = 0;
        while (i < t.size() && std::isdigit((unsigned char)t[i])) { ++i; ++ne; }
        if (ne == 0) return false;
    }
    if (i != t.size()) return false;
    char* e = nullptr;
    v = std::strtod(u.c_str(), &e);
    return e && *e == 0 && std::isfinite(v);
}
bool as_number(const Node& n, double& v) {
    if (!n.atom) return false;
    if (n.text == "pi" || n.text == "π") { v = 3.14159265358979323846; return true; }
    if (n.text == "ℯ") { v = 2.71828182845904523536; return true; }
    if (decimal_literal(n.text, v)) return true;
    // Julia rationals print as a//b
    const size_t p = n.text.find("//");
    if (p == std::string::npos) return false;
    double a = 0.0, d = 0.0;
    if (!decimal_literal(n.text.substr(0, p), a) || !decimal_literal(n.text.substr(p + 2), d) || d == 0.0) return false;
    v = a / d;
    return true;
}
bool is_pi(const Node& n) { return n.atom && (n.text == "pi" || n.text == "π"); }

struct Ref { char kind; int idx; };      // 'x' coordinate, 'p' parameter, 's' slot, 'o' op

struct Lowering {
    const SexprContext& C;
    const std::vector<std::string>& indvars;
    std::vector<Slot> slots;
    struct RawOp { int code; Ref a, b; bool has_a, has_b; float imm; double imm64; };
    std::vector<RawOp> ops;
    std::map<std::string, Ref> memo;
    std::string err;

    Lowering(const SexprContext& c, const std::vector<std::string>& iv) : C(c), indvars(iv) {}

    Ref emit(int code, const Ref* a, const Ref* b, double imm) {
        RawOp o;
        o.code = code; o.has_a = a != nullptr; o.has_b = b != nullptr; o.imm = (float)imm; o.imm64 = imm;
        o.a = a ? *a : Ref{'x', 0}; o.b = b ? *b : Ref{'x', 0};
        ops.push_back(o);
        return Ref{'o', (int)ops.size() - 1};
    }
    Ref konst(double v) { return emit(rp::OP_CONST, nullptr, nullptr, v); }
    Ref unary(int code, Ref a, double imm = 0.0) { return emit(code, &a, nullptr, imm); }
    Ref binary(int code, Ref a, Ref b) { return emit(code, &a, &b, 0.0); }

    int depvar_of(const std::string& name) const {
        for (size_t i = 0; i < C.depvars.size(); ++i)
            if (C.depvars[i] == name) return (int)i;
        return -1;
    }
    bool fail_(const std::string& m) { if (err.empty()) err = m; return false; }

    bool slot_ref(int net, std::vector<int> axes, Ref& out) {
        std::sort(axes.begin(), axes.end());
        if ((int)axes.size() > MAX_DERIV_ORDER) return fail_("derivative order " + std::to_string(axes.size()) + " > 6 of " + C.depvars[net] + " is not supported by the HIP engine");
        for (size_t i = 0; i < slots.size(); ++i) {
            bool same = slots[i].net == net && slots[i].order == (int)axes.size();
            for (size_t a = 0; same && a < axes.size(); ++a) same = slots[i].axes[a] == axes[a];
            if (same) { out = Ref{'s', (int)i}; return true; }
        }
        Slot s;
        s.net = net; s.order = (int)axes.size(); s.lap = 0;
        for (int a = 0; a < MAX_DERIV_ORDER; ++a) s.axes[a] = a < (int)axes.size() ? axes[a] : 0;
        slots.push_back(s);
        out = Ref{'s', (int)slots.size() - 1};
        return true;
    }

    // (D var order expr) chains down to a dependent-variable call
    bool derivative(const Node& n, Ref& out) {
        std::vector<std::pair<std::string, int>> by;
        const Node* cur = &n;
        while (!cur->atom && cur->text == "D") {
            if (cur->args.size() != 3 || !cur->args[0].atom || !cur->args[1].atom) return fail_("malformed (D variable order expr)");
            char* oe = nullptr;
            const long ord = std::strtol(cur->args[1].text.c_str(), &oe, 10);
            if (cur->args[1].text.empty() || *oe != 0 || ord < 1 || ord > MAX_DERIV_ORDER)
                return fail_("derivative order must be an integer 1.." + std::to_string(MAX_DERIV_ORDER) + ", got '" + cur->args[1].text + "'");
            by.push_back({cur->args[0].text, (int)ord});
            cur = &cur->args[2];
        }
        const int net = cur->atom ? -1 : depvar_of(cur->text);
        if (net < 0)
            return fail_("a Differential must act on a dependent variable (apply expand_derivatives first, as parse_equation does, "
                         "src/symbolic_utilities.jl:361-364); got " + cur->key());
        const std::vector<std::string>& inputs = C.depvar_inputs[net];
        std::vector<int> axes;
        for (auto& pr : by) {
            int ax = -1;
            for (size_t i = 0; i < inputs.size(); ++i)
                if (inputs[i] == pr.first) ax = (int)i;
            if (ax < 0) return fail_("derivative of " + C.depvars[net] + " w.r.t. " + pr.first + ", which is not one of its inputs");
            for (int k = 0; k < pr.second; ++k) axes.push_back(ax);
        }
        return slot_ref(net, axes, out);
    }

    bool lower(const Node& n, Ref& out) {
        const std::string k = n.key();
        auto it = memo.find(k);
        if (it != memo.end()) { out = it->second; return true; }
        if (!lower_(n, out)) return false;
        memo[k] = out;
        return true;
    }

    // product of the given factors with a numeric coefficient folded out
    bool product(const std::vector<const Node*>& fs, double coeff, Ref& out) {
        bool have = false;
        Ref acc{'x', 0};
        std::vector<Ref> den;
        for (const Node* f : fs) {
            double v;
            if (as_number(*f, v)) { coeff *= v; continue; }
            if (!f->atom && f->text == "/" && f->args.size() == 2) {          // a / b inside a product: numerator here, denominator later
                double nv;
                Ref r;
                if (as_number(f->args[0], nv)) coeff *= nv;
                else {
                    if (!lower(f->args[0], r)) return false;
                    acc = have ? binary(rp::OP_MUL, acc, r) : r;
                    have = true;
                }
                double dv;
                if (as_number(f->args[1], dv)) coeff /= dv;
                else { if (!lower(f->args[1], r)) return false; den.push_back(r); }
                continue;
            }
            if (!f->atom && f->text == "^" && f->args.size() == 2) {
                double pv;
                if (as_number(f->args[1], pv) && pv == -1.0) { Ref r; if (!lower(f->args[0], r)) return false; den.push_back(r); continue; }
            }
            Ref r;
            if (!lower(*f, r)) return false;
            acc = have ? binary(rp::OP_MUL, acc, r) : r;
            have = true;
        }
        if (!have) { acc = konst(coeff); coeff = 1.0; }
        for (Ref& d : den) acc = binary(rp::OP_DIV, acc, d);
        if (coeff == -1.0) acc = unary(rp::OP_NEG, acc);
        else if (coeff != 1.0) acc = unary(rp::OP_MULC, acc, coeff);
        out = acc;
        return true;
    }

    bool lower_(const Node& n, Ref& out) {
        double v;
        if (n.atom) {
            if (as_number(n, v)) { out = konst(v); return true; }
            for (size_t i = 0; i < indvars.size(); ++i)
                if (indvars[i] == n.text) { out = Ref{'x', (int)i}; return true; }
            for (size_t i = 0; i < C.params.size(); ++i)
                if (C.params[i] == n.text) { out = Ref{'p', (int)i}; return true; }
            return fail_("symbol " + n.text + " is neither an independent variable of this term nor a parameter");
        }
        const std::string& h = n.text;
        if (h == "D") return derivative(n, out);
        const int net = depvar_of(h);
        if (net >= 0) return slot_ref(net, {}, out);                      // call arguments dropped (symbolic_utilities.jl:145-160)
        const size_t na = n.args.size();
        if (h == "+") {
            double c = 0.0;
            bool have = false;
            Ref acc{'x', 0};
            std::vector<Ref> neg;
            for (const Node& a : n.args) {
                if (as_number(a, v)) { c += v; continue; }
                // -x and (-1) * x terms are subtracted
                if (!a.atom && a.text == "-" && a.args.size() == 1) { Ref r; if (!lower(a.args[0], r)) return false; neg.push_back(r); continue; }
                if (!a.atom && a.text == "*") {
                    double coeff = 1.0;
                    std::vector<const Node*> rest;
                    for (const Node& f : a.args) { double fv; if (as_number(f, fv)) coeff *= fv; else rest.push_back(&f); }
                    if (coeff == -1.0 && !rest.empty()) { Ref r; if (!product(rest, 1.0, r)) return false; neg.push_back(r); continue; }
                }
                Ref r;
                if (!lower(a, r)) return false;
                acc = have ? binary(rp::OP_ADD, acc, r) : r;
                have = true;
            }
            for (Ref& r : neg) { acc = have ? binary(rp::OP_SUB, acc, r) : unary(rp::OP_NEG, r); have = true; }
            if (!have) { out = konst(c); return true; }
            if (c != 0.0) acc = unary(rp::OP_ADDC, acc, c);
            out = acc;
            return true;
        }
        if (h == "-") {
            if (na == 1) {
                if (as_number(n.args[0], v)) { out = konst(-v); return true; }
                Ref r;
                if (!lower(n.args[0], r)) return false;
                out = unary(rp::OP_NEG, r);
                return true;
            }
            if (na != 2) return fail_("'-' takes one or two arguments");
            double va, vb;
            const bool ca = as_number(n.args[0], va), cb = as_number(n.args[1], vb);
            if (ca && cb) { out = konst(va - vb); return true; }
            Ref a, b;
            if (cb) { if (!lower(n.args[0], a)) return false; out = vb != 0.0 ? unary(rp::OP_ADDC, a, -vb) : a; return true; }
            if (ca) { if (!lower(n.args[1], b)) return false; b = unary(rp::OP_NEG, b); out = va != 0.0 ? unary(rp::OP_ADDC, b, va) : b; return true; }
            if (!lower(n.args[0], a) || !lower(n.args[1], b)) return false;
            out = binary(rp::OP_SUB, a, b);
            return true;
        }
        if (h == "*") {
            std::vector<const Node*> fs;
            for (const Node& a : n.args) fs.push_back(&a);
            return product(fs, 1.0, out);
        }
        if (h == "/") {
            if (na != 2) return fail_("'/' takes two arguments");
            std::vector<const Node*> fs{&n};
            return product(fs, 1.0, out);
        }
        if (h == "^") {
            if (na != 2) return fail_("'^' takes two arguments");
            Ref b;
            double pv;
            if (as_number(n.args[1], pv)) {
                double bv;
                if (as_number(n.args[0], bv)) { out = konst(std::pow(bv, pv)); return true; }
                if (!lower(n.args[0], b)) return false;
                if (pv == std::floor(pv) && std::fabs(pv) <= 64.0) out = unary(rp::OP_POWI, b, pv);
                else if (pv == 0.5) out = unary(rp::OP_SQRT, b);
                else out = unary(rp::OP_POWC, b, pv);
                return true;
            }
            Ref p;
            if (!lower(n.args[0], b) || !lower(n.args[1], p)) return false;
            out = binary(rp::OP_POW, b, p);
            return true;
        }
        if (h == "inv" && na == 1) { Ref b; if (!lower(n.args[0], b)) return false; out = unary(rp::OP_POWI, b, -1.0); return true; }
        if ((h == "max" || h == "min") && na >= 2) {
            Ref acc;
            if (!lower(n.args[0], acc)) return false;
            for (size_t i = 1; i < na; ++i) { Ref r; if (!lower(n.args[i], r)) return false; acc = binary(h == "max" ? rp::OP_MAX : rp::OP_MIN, acc, r); }
            out = acc;
            return true;
        }
        static const std::map<std::string, int> fn = {
            {"sin", rp::OP_SIN}, {"cos", rp::OP_COS}, {"tan", rp::OP_TAN}, {"exp", rp::OP_EXP}, {"log", rp::OP_LOG}, {"sqrt", rp::OP_SQRT},
            {"abs", rp::OP_ABS}, {"tanh", rp::OP_TANH}, {"sinh", rp::OP_SINH}, {"cosh", rp::OP_COSH}, {"sech", rp::OP_SECH},
            {"sinpi", rp::OP_SINPI}, {"cospi", rp::OP_COSPI}};
        auto f = fn.find(h);
        if (f != fn.end() && na == 1) {
            const Node& a = n.args[0];
            if (as_number(a, v)) {                                        // constant folding keeps e.g. sin(pi) exact
                const double pi = 3.14159265358979323846;
                switch (f->second) {
                    case rp::OP_SIN: out = konst(is_pi(a) ? 0.0 : std::sin(v)); return true;
                    case rp::OP_COS: out = konst(is_pi(a) ? -1.0 : std::cos(v)); return true;
                    case rp::OP_EXP: out = konst(std::exp(v)); return true;
                    case rp::OP_SQRT: out = konst(std::sqrt(v)); return true;
                    case rp::OP_SINPI: out = konst(std::sin(pi * v)); return true;
                    case rp::OP_COSPI: out = konst(std::cos(pi * v)); return true;
                    default: break;
                }
            }
            // sin(pi * z) -> SINPI(z): evaluated as sinpif on the device, closer to the reference's Float64 sin(pi z) than an fp32
            // product pi * z followed by sinf
            if ((f->second == rp::OP_SIN || f->second == rp::OP_COS) && !a.atom && a.text == "*") {
                int npi = 0;
                std::vector<const Node*> rest;
                for (const Node& q : a.args) { if (is_pi(q)) ++npi; else rest.push_back(&q); }
                if (npi == 1 && !rest.empty()) {
                    Ref z;
                    if (!product(rest, 1.0, z)) return false;
                    out = unary(f->second == rp::OP_SIN ? rp::OP_SINPI : rp::OP_COSPI, z);
                    return true;
                }
            }
            Ref r;
            if (!lower(a, r)) return false;
            out = unary(f->second, r);
            return true;
        }
        return fail_("function '" + h + "' with " + std::to_string(na) + " argument(s) is outside the engine's closed op set (SURVEY.md App. B)");
    }
};

}  // namespace

int lower_sexpr_term(const SexprContext& C, const std::vector<std::string>& indvars, const std::string& lhs, const std::string& rhs,
                     int np, Term& T) {
    Node nl, nr;
    Parser pl(lhs), pr(rhs);
    if (!pl.parse(nl)) return fail("sexpr (lhs): " + pl.err);
    pl.skip();
    if (pl.i != lhs.size()) return fail("sexpr (lhs): trailing characters");
    if (!pr.parse(nr)) return fail("sexpr (rhs): " + pr.err);
    pr.skip();
    if (pr.i != rhs.size()) return fail("sexpr (rhs): trailing characters");
    Lowering L(C, indvars);
    Node res;                                       // residual = lhs - rhs (symbolic_utilities.jl:365-369)
    res.atom = false;
    res.text = "-";
    res.args = {nl, nr};
    Ref out;
    if (!L.lower(res, out)) return fail("residual lowering: " + L.err);
    if (L.slots.empty()) {
        // the residual does not depend on any dependent variable after folding; bind the term to the first one named in the equation
        int net = -1;
        std::vector<const Node*> st{&nl, &nr};
        while (!st.empty() && net < 0) {
            const Node* q = st.back();
            st.pop_back();
            if (!q->atom) {
                for (size_t i = 0; i < C.depvars.size(); ++i) if (C.depvars[i] == q->text) net = (int)i;
                for (auto& a : q->args) st.push_back(&a);
            }
        }
        if (net < 0) return fail("equation does not contain a dependent variable");
        Ref dummy;
        L.slot_ref(net, {}, dummy);
    }
    if (out.kind != 'o') out = L.unary(rp::OP_ADDC, out, 0.0);             // residual is a bare input row: materialise it
    const int d = (int)indvars.size(), S = (int)L.slots.size();
    auto row = [&](const Ref& r) -> int {
        switch (r.kind) {
            case 'x': return r.idx;
            case 'p': return d + r.idx;
            case 's': return d + np + r.idx;
            default: return d + np + S + r.idx;
        }
    };
    T.d = d;
    T.slots = L.slots;
    T.ops.clear();
    T.imm64.clear();
    for (auto& o : L.ops) {
        rp::Instr I;
        I.code = o.code;
        I.a = o.has_a ? row(o.a) : 0;
        I.b = (o.has_b && rp::is_binary(o.code)) ? row(o.b) : 0;
        I.imm = o.imm;
        rp::finalize(I);
        T.ops.push_back(I);
        T.imm64.push_back(o.imm64);
    }
    T.out_row = row(out);
    T.ndata = 0;
    // every dependent variable reads its own rows of `cord` (src/discretize.jl:111-131)
    T.inmap.clear();
    for (auto& s : T.slots) {
        if (T.inmap.count(s.net)) continue;
        const std::vector<std::string>& in = C.depvar_inputs[s.net];
        std::vector<int> m;
        for (auto& name : in) {
            int ix = -1;
            for (int i = 0; i < d; ++i) if (indvars[i] == name) ix = i;
            if (ix < 0) return fail("input " + name + " of " + C.depvars[s.net] + " is not among the term's coordinates");
            m.push_back(ix);
        }
        bool ident = (int)m.size() == d;
        for (size_t i = 0; ident && i < m.size(); ++i) ident = m[i] == (int)i;
        if (!ident) T.inmap[s.net] = m;
    }
    if (d + np + S + (int)T.ops.size() > rp::MAX_ROWS) return fail("residual expression too long");
    return 0;
}

}  // namespace pe
