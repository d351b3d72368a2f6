"""Prefix s-expression printer for the engine's symbolic front end ("pinnir 2", csrc/sexpr.cpp).

The Julia glue (julia/NeuralPDEHIP.jl: `sexpr`) prints the Expr trees the reference itself walks — `toexpr(expand_derivatives(eq.lhs))`
and `toexpr(expand_derivatives(eq.rhs))` (src/symbolic_utilities.jl:360-370); this module prints the same shape from sympy trees, so
that the Python mirror exercises the ONE lowering implementation both hosts share (the C++ restatement of `_transform_expression`,
src/symbolic_utilities.jl:132-331).  Shape: `(head arg ...)` with heads `+ - * / ^`, function names, dependent-variable names
(`(u x y)`, `(u 0 y)`) and `(D <variable> <order> <expr>)` for Differentials (nested for mixed derivatives, as Symbolics nests them)."""
from __future__ import annotations

import sympy as sp

_FUNCS = {"sin": "sin", "cos": "cos", "tan": "tan", "exp": "exp", "log": "log", "tanh": "tanh", "sinh": "sinh", "cosh": "cosh",
          "Abs": "abs", "sech": "sech"}


class SexprError(ValueError):
    pass


def _num(v) -> str:
    f = float(v)
    if f == int(f) and abs(f) < 1e15:
        return str(int(f))
    return repr(f)


def sexpr(e) -> str:
    e = sp.sympify(e)
    if e is sp.pi:
        return "pi"
    if e.is_Number or e.is_NumberSymbol:
        return _num(e)
    if e.is_Symbol:
        return str(e)
    if isinstance(e, sp.Derivative):
        inner = e.expr
        if not isinstance(inner, sp.core.function.AppliedUndef):
            d = e.doit()                                   # expand_derivatives (symbolic_utilities.jl:361-364)
            if isinstance(d, sp.Derivative) and d == e:
                raise SexprError(f"cannot expand derivative {e}")
            return sexpr(d)
        s = sexpr(inner)
        for var, n in e.variable_count:                    # innermost first: Dy(Dxx(u)) prints as (D y 1 (D x 2 (u x y)))
            s = f"(D {var} {int(n)} {s})"
        return s
    if isinstance(e, sp.core.function.AppliedUndef):
        return "(" + " ".join([str(e.func)] + [sexpr(a) for a in e.args]) + ")"
    if e.is_Add:
        return "(+ " + " ".join(sexpr(a) for a in e.args) + ")"
    if e.is_Mul:
        return "(* " + " ".join(sexpr(a) for a in e.args) + ")"
    if e.is_Pow:
        b, p = e.args
        if p == -1:
            return f"(/ 1 {sexpr(b)})"
        return f"(^ {sexpr(b)} {sexpr(p)})"
    if isinstance(e, (sp.Max, sp.Min)):
        return "(" + ("max" if isinstance(e, sp.Max) else "min") + " " + " ".join(sexpr(a) for a in e.args) + ")"
    if isinstance(e, sp.Function):
        name = type(e).__name__
        if name in _FUNCS and len(e.args) == 1:
            return f"({_FUNCS[name]} {sexpr(e.args[0])})"
        raise SexprError(f"function {name} is outside the engine's closed op set (SURVEY.md App. B)")
    raise SexprError(f"cannot print {type(e).__name__}: {e}")
