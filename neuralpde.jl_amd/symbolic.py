"""Problem statement objects and the symbolic -> residual-IR lowering.

Python (sympy) mirror of the part of the reference that stays on the host:
  PDESystem / Differential / Interval     [3P] ModelingToolkit / Symbolics / DomainSets
  get_vars, get_argument, get_variables   src/symbolic_utilities.jl:401-426, 456-468, 498-526
  parse_equation / _transform_expression  src/symbolic_utilities.jl:132-331, 360-370
The lowering emits IR (ir.py) instead of a Julia Expr:
  * a dependent-variable call `u(x, y)` / `u(0, y)` becomes the value slot of its network — the call
    arguments are DROPPED exactly as the reference does (symbolic_utilities.jl:145-160): boundary values
    come from the point set, not from the expression;
  * (nested) `Differential`s of a dependent variable collapse into one jet slot (net, sorted axes)
    (symbolic_utilities.jl:161-202).  Where the reference evaluates it with central finite differences
    (`numeric_derivative`, src/pinn_types.jl:445-482) the engine evaluates the exact derivative by
    Taylor-mode propagation; the two agree to ~1e-8 in the reference's default Float64 mode;
  * `expand_derivatives` first (symbolic_utilities.jl:361-364)  ->  sympy `.doit()`;
  * residual = lhs - rhs (symbolic_utilities.jl:365-369).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import sympy as sp

from .ir import BINARY, Instr, Slot, TermIR


# ------------------------------------------------------------------------------------------------
# problem statement
# ------------------------------------------------------------------------------------------------
def parameters(names: str):
    """`@parameters x y` -> sympy symbols."""
    s = sp.symbols(names, real=True)
    return s if isinstance(s, tuple) else (s,)


def variables(names: str):
    """`@variables u(..)` -> sympy undefined functions."""
    fs = [sp.Function(n) for n in names.replace(",", " ").split()]
    return tuple(fs)


class Differential:
    """`Differential(x)`; `Differential(x)^2` is `Differential(x)**2`."""

    def __init__(self, var, order: int = 1):
        self.var, self.order = var, order

    def __pow__(self, n: int):
        return Differential(self.var, self.order * int(n))

    def __call__(self, expr):
        return sp.Derivative(expr, (self.var, self.order))


@dataclass
class Equation:
    """`lhs ~ rhs` (a sympy Eq would auto-evaluate trivial equalities such as `u(0) ~ u(0)`)."""
    lhs: object
    rhs: object

    def __post_init__(self):
        self.lhs = sp.sympify(self.lhs)
        self.rhs = sp.sympify(self.rhs)


def Eq(lhs, rhs) -> Equation:
    return Equation(lhs, rhs)


@dataclass
class Interval:
    lo: float
    hi: float


@dataclass
class VarDomain:
    """`x ∈ Interval(a, b)`."""
    variable: object
    domain: Interval


def In(var, interval: Interval) -> VarDomain:
    return VarDomain(var, interval)


@dataclass
class PDESystem:
    """PDESystem(eqs, bcs, domains, ivs, dvs[, ps]; defaults) — [3P] ModelingToolkit; usage e.g.
    test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:59-82."""
    eqs: Sequence[Equation]
    bcs: Sequence[Equation]
    domain: Sequence[VarDomain]
    ivs: Sequence
    dvs: Sequence                     # applied functions, e.g. [u(x, y)]
    ps: Sequence = ()                 # parameter symbols
    defaults: Optional[Dict] = None   # {param symbol: value}

    def __post_init__(self):
        if isinstance(self.eqs, Equation):
            self.eqs = [self.eqs]
        if isinstance(self.bcs, Equation):
            self.bcs = [self.bcs]
        self.eqs, self.bcs = list(self.eqs), list(self.bcs)


# ------------------------------------------------------------------------------------------------
# get_vars / get_argument / get_variables
# ------------------------------------------------------------------------------------------------
@dataclass
class VarInfo:
    depvars: List[str]
    indvars: List[str]
    dict_indvars: Dict[str, int]
    dict_depvars: Dict[str, int]
    dict_depvar_input: Dict[str, List[str]]
    sym: Dict[str, object]            # name -> sympy symbol / function


def get_vars(ivs, dvs) -> VarInfo:
    """src/symbolic_utilities.jl:401-426."""
    indvars = [str(v) for v in ivs]
    depvars, ddi, sym = [], {}, {str(v): v for v in ivs}
    for d in dvs:
        if isinstance(d, sp.core.function.AppliedUndef):
            name = str(d.func)
            depvars.append(name)
            ddi[name] = [str(a) for a in d.args]
            sym[name] = d.func
        else:
            name = str(d)
            depvars.append(name)
            ddi[name] = list(indvars)          # default to all inputs if not given (:418)
            sym[name] = d
    return VarInfo(depvars, indvars, {n: i + 1 for i, n in enumerate(indvars)}, {n: i + 1 for i, n in enumerate(depvars)}, ddi, sym)


def _depvar_calls(expr, vi: VarInfo):
    """All applications of dependent variables in an expression, in first-seen (pre-order) order."""
    found = []
    for node in sp.preorder_traversal(expr):
        if isinstance(node, sp.core.function.AppliedUndef) and str(node.func) in vi.dict_depvars and node not in found:
            found.append(node)
    return found


def get_argument(eqs: Sequence[Equation], vi: VarInfo) -> List[List]:
    """Arguments of the dependent-variable calls of each equation: symbols once each, numbers kept
    (src/symbolic_utilities.jl:502-526).  The reference takes `first` of an unordered Set of call sites
    per dependent variable; we take the first call in pre-order of `lhs - rhs` (identical whenever all
    calls of a depvar in one equation have the same arguments, which holds for every reference test)."""
    out = []
    for eq in eqs:
        calls = []
        seen_dep = set()
        for node in _depvar_calls(sp.Add(eq.lhs, -eq.rhs, evaluate=False), vi):
            if str(node.func) not in seen_dep:
                seen_dep.add(str(node.func))
                calls.append(node)
        # order of dict_depvars keys
        calls.sort(key=lambda c: vi.dict_depvars[str(c.func)])
        args, syms = [], set()
        for c in calls:
            for a in c.args:
                if a.is_Symbol:
                    if str(a) in syms:
                        continue
                    syms.add(str(a))
                    args.append(a)
                else:
                    args.append(float(a))
        out.append(args)
    return out


def get_variables(eqs, vi: VarInfo) -> List[List]:
    """src/symbolic_utilities.jl:456-468."""
    return [[a for a in args if not isinstance(a, float)] for args in get_argument(eqs, vi)]


def term_indvars(eq: Equation, vi: VarInfo) -> List[str]:
    """`this_eq_indvars = unique(vcat(values(this_eq_pair)...))` — inputs of every depvar appearing in the
    equation (src/discretize.jl:41-43), in depvar order."""
    names = []
    present = {str(c.func) for c in _depvar_calls(sp.Add(eq.lhs, -eq.rhs, evaluate=False), vi)}
    for dv in vi.depvars:
        if dv in present:
            for n in vi.dict_depvar_input[dv]:
                if n not in names:
                    names.append(n)
    return names


# ------------------------------------------------------------------------------------------------
# lowering
# ------------------------------------------------------------------------------------------------
class LoweringError(ValueError):
    pass


_FUNCS = {"sin": "SIN", "cos": "COS", "tan": "TAN", "exp": "EXP", "log": "LOG", "tanh": "TANH", "sinh": "SINH",
          "cosh": "COSH", "Abs": "ABS", "sech": "SECH"}


class _Builder:
    def __init__(self, vi: VarInfo, indvars: List[str], params: List[str]):
        self.vi, self.indvars, self.params = vi, indvars, params
        self.slots: List[Slot] = []
        self.ops: List[Instr] = []
        self.memo: Dict[object, Tuple[str, int]] = {}

    # references are ('x', i) coordinate, ('p', i) parameter, ('s', i) slot, ('o', i) op
    def slot(self, s: Slot):
        if s not in self.slots:
            self.slots.append(s)
        return ("s", self.slots.index(s))

    def emit(self, op, a=None, b=None, imm=0.0):
        self.ops.append((op, a, b, float(imm)))
        return ("o", len(self.ops) - 1)

    def const(self, v):
        return self.emit("CONST", imm=float(v))

    def lower(self, e):
        if e in self.memo:
            return self.memo[e]
        r = self._lower(e)
        self.memo[e] = r
        return r

    def _depvar_slot(self, call, axes_syms):
        name = str(call.func)
        net = self.vi.dict_depvars[name] - 1
        inputs = self.vi.dict_depvar_input[name]
        axes = []
        for (v, n) in axes_syms:
            if str(v) not in inputs:
                raise LoweringError(f"derivative of {name} w.r.t. {v}, which is not one of its inputs {inputs}")
            axes += [inputs.index(str(v))] * int(n)
        if len(axes) > 6:
            raise LoweringError(f"derivative order {len(axes)} > 6 of {name} is not supported by the HIP engine")
        # orders <= 2 (incl. mixed) and pure 3rd / 4th derivatives ride the fixed jet-channel categories; any other multi-index (mixed
        # of order >= 3, orders 5-6: the reference's recursion takes them all, src/pinn_types.jl:454-460) gets a generated jet set
        return self.slot(Slot(net, tuple(sorted(axes))))

    def _lower(self, e):
        if e.is_Number or e.is_NumberSymbol:
            return self.const(float(e))
        if e.is_Symbol:
            n = str(e)
            if n in self.indvars:
                return ("x", self.indvars.index(n))
            if n in self.params:
                return ("p", self.params.index(n))
            raise LoweringError(f"symbol {n} is neither an independent variable of this term {self.indvars} nor a parameter")
        if isinstance(e, sp.Derivative):
            inner = e.expr
            if isinstance(inner, sp.core.function.AppliedUndef) and str(inner.func) in self.vi.dict_depvars:
                return self._depvar_slot(inner, e.variable_count)
            d = e.doit()
            if isinstance(d, sp.Derivative) and d == e:
                raise LoweringError(f"cannot expand derivative {e}")
            return self.lower(d)
        if isinstance(e, sp.core.function.AppliedUndef):
            if str(e.func) in self.vi.dict_depvars:
                return self._depvar_slot(e, [])
            raise LoweringError(f"unknown function {e.func}")
        if e.is_Add:
            const = 0.0
            pos, neg = [], []
            for a in e.args:
                if a.is_Number or a.is_NumberSymbol:
                    const += float(a)
                    continue
                c, rest = a.as_coeff_Mul()
                if c == -1:
                    neg.append(self.lower(rest))
                else:
                    pos.append(self.lower(a))
            acc = None
            for r in pos:
                acc = r if acc is None else self.emit("ADD", acc, r)
            for r in neg:
                acc = self.emit("NEG", r) if acc is None else self.emit("SUB", acc, r)
            if acc is None:
                return self.const(const)
            if const != 0.0:
                acc = self.emit("ADDC", acc, imm=const)
            return acc
        if e.is_Mul:
            coeff = 1.0
            num, den = [], []
            for a in e.args:
                if a.is_Number or a.is_NumberSymbol:
                    coeff *= float(a)
                elif a.is_Pow and a.args[1].is_Number and a.args[1] == -1:
                    den.append(self.lower(a.args[0]))
                else:
                    num.append(self.lower(a))
            acc = None
            for r in num:
                acc = r if acc is None else self.emit("MUL", acc, r)
            if acc is None:
                acc = self.const(coeff)
                coeff = 1.0
            for r in den:
                acc = self.emit("DIV", acc, r)
            if coeff == -1.0:
                acc = self.emit("NEG", acc)
            elif coeff != 1.0:
                acc = self.emit("MULC", acc, imm=coeff)
            return acc
        if e.is_Pow:
            b, p = e.args
            if p.is_Integer:
                return self.emit("POWI", self.lower(b), imm=int(p))
            if p.is_Number and p == sp.Rational(1, 2):
                return self.emit("SQRT", self.lower(b))
            if p.is_Number or p.is_NumberSymbol:
                return self.emit("POWC", self.lower(b), imm=float(p))
            return self.emit("POW", self.lower(b), self.lower(p))
        if isinstance(e, (sp.Max, sp.Min)):
            op = "MAX" if isinstance(e, sp.Max) else "MIN"
            acc = self.lower(e.args[0])
            for a in e.args[1:]:
                acc = self.emit(op, acc, self.lower(a))
            return acc
        if isinstance(e, sp.Function):
            name = type(e).__name__
            if name in ("sin", "cos") and len(e.args) == 1:
                # sin(pi*z) -> SINPI(z): evaluated as sinpif on the device, closer to the reference's Float64
                # sin(pi*z) than an fp32 product pi*z followed by sinf.
                arg = e.args[0]
                if arg.is_Mul and sp.pi in arg.args:
                    return self.emit("SINPI" if name == "sin" else "COSPI", self.lower(arg / sp.pi))
                if arg == sp.pi:
                    return self.const(0.0 if name == "sin" else -1.0)
            if name in _FUNCS:
                return self.emit(_FUNCS[name], self.lower(e.args[0]))
            raise LoweringError(f"function {name} is outside the engine's closed op set (SURVEY.md App. B)")
        raise LoweringError(f"cannot lower {type(e).__name__}: {e}")


def lower_equation(eq: Equation, vi: VarInfo, params: Sequence, kind: str) -> TermIR:
    """parse_equation (src/symbolic_utilities.jl:360-370) + build_symbolic_loss_function
    (src/discretize.jl:28-152) for one equation, to IR."""
    indvars = term_indvars(eq, vi)
    pnames = [str(p) for p in params]
    B = _Builder(vi, indvars, pnames)
    expr = (eq.lhs - eq.rhs)
    if not _depvar_calls(sp.Add(eq.lhs, -eq.rhs, evaluate=False), vi):
        raise LoweringError("equation does not contain a dependent variable")
    # an identically-zero residual (e.g. `u(0.0) ~ u(0.0)`, test/Forward/forward__ode.jl:12) is legal
    out = B.lower(sp.sympify(expr))
    if not B.slots:     # residual simplified to a constant: still bind the term to its network
        first = _depvar_calls(sp.Add(eq.lhs, -eq.rhs, evaluate=False), vi)[0]
        B.slot(Slot(vi.dict_depvars[str(first.func)] - 1, ()))
    d, NP, S = len(indvars), len(pnames), len(B.slots)

    def row(ref):
        k, i = ref
        return {"x": i, "p": d + i, "s": d + NP + i, "o": d + NP + S + i}[k]

    if out[0] != "o":        # residual is a bare input row: materialise it
        out = B.emit("ADDC", out, imm=0.0)
    ops = []
    for (op, a, b, imm) in B.ops:
        ops.append(Instr(op, row(a) if a is not None else 0, row(b) if (b is not None and op in BINARY) else 0, imm))
    # every dependent variable gets its own rows of `cord` (src/discretize.jl:111-131): inputs of net k = the term's
    # coordinates named by dict_depvar_input[depvar k]
    inmaps = {}
    for sl in B.slots:
        if sl.net in inmaps:
            continue
        name = vi.depvars[sl.net]
        m = tuple(indvars.index(v) for v in vi.dict_depvar_input[name])
        if m != tuple(range(d)):
            inmaps[sl.net] = m
    from .sexpr import SexprError, sexpr
    try:                                                   # the same equation for the symbolic front end ("pinnir 2")
        ls, rs = sexpr(eq.lhs), sexpr(eq.rhs)
    except SexprError:
        ls = rs = None
    return TermIR(dim=d, slots=list(B.slots), ops=ops, out_row=row(out), indvars=tuple(indvars), kind=kind,
                  source=str(expr), inmaps=inmaps, lhs_sexpr=ls, rhs_sexpr=rs)
