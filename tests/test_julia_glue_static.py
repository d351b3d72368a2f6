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


def test_ccall_argument_types_match_the_header():
    """argument TYPE classes (32/64-bit integers, floats, pointers, C strings) of every ccall against the C declarations: a Cint where the
    header says int64_t would be a silent stack/register mismatch on the first real run."""
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)

    def cclass(t):
        t = t.strip()
        if "*" in t:
            return "str" if re.match(r"const\s+char\s*\*", t) else "ptr"
        t = re.sub(r"\b[a-z_0-9]+$", "", t).strip() if re.search(r"\s", t) else t
        return {"int": "i32", "int64_t": "i64", "float": "f32", "double": "f64", "uint64_t": "u64", "pinn_handle": "ptr", "unsigned": "u32"}.get(t, t)

    def jclass(t):
        t = t.strip()
        if t == "Cstring":
            return "str"
        if t.startswith("Ptr{") or t.startswith("Ref{"):
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Clonglong": "i64", "Cfloat": "f32", "Float32": "f32", "Cdouble": "f64",
                "Float64": "f64", "UInt64": "u64", "Culonglong": "u64", "Cuint": "u32"}.get(t, t)
    decl = {}
    for m in re.finditer(r"\b(pinn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        args = m.group(2).strip()
        decl[m.group(1)] = [] if args in ("", "void") else [cclass(a) for a in args.split(",")]
    src = _strip(open(JL).read())
    n = 0
    for m in re.finditer(r"ccall\(sym\(:(pinn_[a-z0-9_]+)\),\s*[A-Za-z0-9_{}]+,\s*\(([^()]*(?:\([^()]*\)[^()]*)*)\)", src):
        ja = [jclass(a) for a in re.split(r",(?![^{]*})", m.group(2).strip()) if a.strip()]
        assert ja == decl[m.group(1)], (m.group(1), ja, decl[m.group(1)])
        n += 1
    assert n >= 10
