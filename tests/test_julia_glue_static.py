"""julia/NeuralPDEHIP.jl cannot be executed here (no Julia).  What can be checked statically is: every C symbol it `ccall`s is declared in
include/pinn_hip.h with the same number of arguments, its block structure and brackets balance, and the descriptor strings its printer is
documented to emit are the golden ones the Python mirror produces (tests/test_sexpr_frontend.py covers the latter)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "NeuralPDEHIP.jl")
HDR = os.path.join(ROOT, "include", "pinn_hip.h")


def _strip(s):
    out, i, n = [], 0, len(s)
    while i < n:
        if s.startswith('"""', i):
            j = s.find('"""', i + 3)
            i = (j + 3) if j >= 0 else n
            out.append('""')
        elif s[i] == '"':
            j = i + 1
            while j < n and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif s[i] == "#":
            j = s.find("\n", i)
            i = j if j >= 0 else n
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def _header_arity():
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    ar = {}
    for m in re.finditer(r"\b(pinn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        args = m.group(2).strip()
        ar[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return ar


def test_every_ccall_matches_the_header():
    src = _strip(open(JL).read())
    ar = _header_arity()
    calls = list(re.finditer(r"ccall\(sym\(:(pinn_[a-z0-9_]+)\),\s*[A-Za-z0-9_{}]+,\s*\(([^()]*(?:\([^()]*\)[^()]*)*)\)", src))
    assert len(calls) >= 10
    for m in calls:
        name, argt = m.group(1), m.group(2).strip()
        assert name in ar, f"{name} is ccall'ed by the Julia glue but not declared in include/pinn_hip.h"
        n = 0 if argt == "" else len([a for a in re.split(r",(?![^{]*})", argt) if a.strip()])
        assert n == ar[name], f"{name}: the Julia glue passes {n} arguments, the header declares {ar[name]}"


def test_block_structure_balances():
    t = _strip(open(JL).read())
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ch in t:
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            assert stack and stack.pop() == pairs[ch], "unbalanced brackets in julia/NeuralPDEHIP.jl"
    assert not stack
    opens = ends = depth = 0
    for m in re.finditer(r"[\[\]\(\)]|\b(function|if|for|while|let|begin|do|struct|module|try|quote|macro|end)\b", t):
        tok = m.group(0)
        if tok in "[(":
            depth += 1
        elif tok in "])":
            depth -= 1
        elif depth == 0:
            ends += tok == "end"
            opens += tok != "end"
    assert opens == ends, (opens, ends)
