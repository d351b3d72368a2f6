"""Residual IR ("pinnir 1"): the serialisable problem description handed to `pinn_create`.

It carries exactly the information `symbolic_discretize` extracts from a PDESystem in the reference
(src/discretize.jl:413-545): the chains, the flat-theta layout, and per equation / boundary
condition the residual expression with its `u(...)` / `derivative(...)` call sites — here as jet
slots plus a straight-line SSA tape instead of a Julia Expr (src/symbolic_utilities.jl:132-331).

Row numbering inside a term (descriptor side):
    [0, d)                 coordinates, in the positional order of `this_eq_indvars` (discretize.jl:126-131)
    [d, d+NP)              PDE parameters (theta.p when param_estim, else default_p; discretize.jl:83-109)
    [d+NP, d+NP+S)         jet slots
    [d+NP+S, ...)          ops
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

OPS = ["CONST", "ADD", "SUB", "MUL", "DIV", "NEG", "ADDC", "MULC", "POWI", "POW", "POWC", "SIN", "COS", "TAN", "EXP",
       "LOG", "SQRT", "ABS", "TANH", "SINH", "COSH", "SECH", "SINPI", "COSPI", "MAX", "MIN", "DATA"]
BINARY = {"ADD", "SUB", "MUL", "DIV", "POW", "MAX", "MIN"}
NULLARY = {"CONST", "DATA"}


@dataclass(frozen=True)
class Slot:
    net: int
    axes: Tuple[int, ...]        # () = value; (i,) = d/dx_i; (i, j) = d2/dx_i dx_j; ... any multi-index up to order 6 (net-input axes, sorted)

    @property
    def order(self) -> int:
        return len(self.axes)


@dataclass
class Instr:
    op: str
    a: int = 0
    b: int = 0
    imm: float = 0.0


@dataclass
class TermIR:
    dim: int
    slots: List[Slot]
    ops: List[Instr]
    out_row: int
    indvars: Tuple[str, ...] = ()       # names bound to the rows of `cord`
    kind: str = "pde"                   # "pde" | "bc"
    source: str = ""                    # printable form of lhs - rhs
    # net -> for each network input the index of the term coordinate that feeds it, only for networks whose inputs are
    # not simply the term's coordinates in order (dependent variables with different arguments, src/discretize.jl:111-131)
    inmaps: Dict[int, Tuple[int, ...]] = field(default_factory=dict)
    # the equation as the pair of prefix s-expressions the symbolic front end lowers ("pinnir 2", csrc/sexpr.cpp); None for terms that
    # exist only as tapes (DataLoss)
    lhs_sexpr: str = None
    rhs_sexpr: str = None


@dataclass
class NetIR:
    sizes: Tuple[int, ...]
    act: str
    theta_off: int
    embed: Tuple[Tuple[int, float], ...] = ()      # PeriodicEmbedding in front of the chain: (0-based input index, period) pairs

    def lines(self, i: int) -> list:
        out = [f"net {i} {self.act} {self.theta_off} {len(self.sizes)} " + " ".join(str(s) for s in self.sizes)]
        if self.embed:
            out.append(f"embed {i} {len(self.embed)} " + " ".join(f"{int(ix)} {float(p)!r}" for ix, p in self.embed))
        return out


@dataclass
class ProblemIR:
    ntheta: int
    nets: List[NetIR]
    terms: List[TermIR]
    nparams: int = 0
    nparams_estim: int = 0
    p_theta_off: int = 0
    p_defaults: Sequence[float] = ()
    # names for the symbolic front end ("pinnir 2"): PDE parameters, dependent variable of every net and its arguments
    param_names: Sequence[str] = ()
    depvar_names: Sequence[str] = ()
    depvar_inputs: Sequence[Sequence[str]] = ()

    @staticmethod
    def _hint_lines(hints) -> list:
        """`hint <term> <points>` lines (optional tail of a descriptor): the point counts the caller is about to install; lets the planner
        put a boundary condition of a few points onto a launch the network already has (csrc/plan.cpp)."""
        return [f"hint {k} {int(n)}" for k, n in enumerate(hints or []) if n and n > 0]

    def to_descriptor2(self, hints=None) -> str:
        """"pinnir 2": the equations travel as s-expressions and are lowered inside the library (csrc/sexpr.cpp) — the form the Julia glue
        emits (julia/NeuralPDEHIP.jl).  Terms without a symbolic form (DataLoss tapes) keep their pinnir-1 lines."""
        out = ["pinnir 2", f"ntheta {self.ntheta}",
               f"params {self.nparams} {self.nparams_estim} {self.p_theta_off}",
               "defaults " + " ".join(repr(float(v)) for v in list(self.p_defaults)[: self.nparams]),
               "pnames " + " ".join(self.param_names),
               f"nets {len(self.nets)}"]
        for i, n in enumerate(self.nets):
            out += n.lines(i)
            out.append(f"netvar {i} {self.depvar_names[i]} {len(self.depvar_inputs[i])} " + " ".join(self.depvar_inputs[i]))
        out.append(f"terms {len(self.terms)}")
        for i, t in enumerate(self.terms):
            if t.lhs_sexpr is None:
                out += self._term_lines(i, t)
                continue
            out.append(f"sterm {i} {t.dim} " + " ".join(t.indvars))
            out.append("lhs " + t.lhs_sexpr)
            out.append("rhs " + t.rhs_sexpr)
        return "\n".join(out + self._hint_lines(hints)) + "\n"

    def _term_lines(self, i, t):
        out = [f"term {i} {t.dim} {len(t.slots)} {len(t.ops)} {t.out_row}"]
        for s in t.slots:
            out.append(f"slot {s.net} {s.order} " + " ".join(str(a) for a in s.axes))
        for q in t.ops:
            out.append(f"op {q.op} {q.a} {q.b} {float(q.imm)!r}")
        for net, m in sorted(t.inmaps.items()):
            out.append(f"inmap {net} {len(m)} " + " ".join(str(i) for i in m))
        return out

    def to_descriptor(self, hints=None) -> str:
        out = ["pinnir 1", f"ntheta {self.ntheta}",
               f"params {self.nparams} {self.nparams_estim} {self.p_theta_off}",
               "defaults " + " ".join(repr(float(v)) for v in list(self.p_defaults)[: self.nparams]),
               f"nets {len(self.nets)}"]
        for i, n in enumerate(self.nets):
            out += n.lines(i)
        out.append(f"terms {len(self.terms)}")
        for i, t in enumerate(self.terms):
            out += self._term_lines(i, t)
        return "\n".join(out + self._hint_lines(hints)) + "\n"
