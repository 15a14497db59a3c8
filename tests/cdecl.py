"""Test helper: a small parser of include/tortoise_hip.h (functions and structs) and of julia/TortoiseHIP.jl (ccall tuples and
struct fields), for the static cross-checks of the bindings (tests/test_julia_shim.py, tests/test_abi.py). No Julia toolchain
exists in the image, so the shim cannot be executed: what CAN be checked is that every `ccall` names an exported function with
the header's arity and C types, and that the Julia structs mirror the C structs field for field."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tortoise_hip.h")
JULIA = os.path.join(ROOT, "julia", "TortoiseHIP.jl")


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def _norm_ctype(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", "*", t)
    return t


def parse_header(path=HEADER):
    """-> (functions {name: (ret, [argtype, ...])}, structs {name: [(field, ctype), ...]})"""
    src = re.sub(r"^[ \t]*#.*$", " ", _strip_c_comments(open(path).read()), flags=re.M)      # preprocessor lines
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"([\w\s]+?)\s+([\w\s,]+)$", decl)
            ctype = _norm_ctype(mm.group(1))
            for name in mm.group(2).split(","):
                fields.append((name.strip(), ctype))
        structs[m.group(3)] = fields
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"enum\s*\{.*?\}\s*;", " ", body, flags=re.S)
    funcs = {}
    for m in re.finditer(r"([A-Za-z_][\w\s]*?[\s\*]+)(tsat_\w+)\s*\(([^()]*)\)\s*;", body):
        ret = _norm_ctype(m.group(1))
        args = []
        a = m.group(3).strip()
        if a and a != "void":
            for arg in a.split(","):
                arg = arg.strip()
                mm = re.match(r"(.*?[\s\*])(\w+)$", arg)          # drop the parameter name
                args.append(_norm_ctype(mm.group(1) if mm else arg))
        funcs[m.group(2)] = (ret, args)
    return funcs, structs


def _split_top(s):
    """split a Julia tuple body at top-level commas"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(s, i):
    """index just past the parenthesis group that opens at s[i] == '('"""
    depth = 0
    for j in range(i, len(s)):
        if s[j] == "(":
            depth += 1
        elif s[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced")


def parse_julia(path=JULIA):
    """-> (ccalls [(name, ret, [argtype, ...], n_values)], structs {name: [(field, jltype), ...]}, unbound [names])"""
    src = open(path).read()
    code = "\n".join(re.sub(r"(^|\s)#(?!=).*$", "", ln) for ln in src.splitlines())       # line comments
    calls = []
    for m in re.finditer(r"ccall\(\s*\(\s*:(\w+)\s*,\s*LIB\s*\)\s*,", code):
        start = code.rfind("ccall(", 0, m.end()) + len("ccall")
        end = _balanced(code, start)
        parts = _split_top(code[start + 1:end - 1])
        # parts: [(:name, LIB), Ret, (ArgTypes...), values...]
        ret = parts[1]
        tup = parts[2].strip()
        assert tup.startswith("(") and tup.endswith(")"), tup
        argt = [a for a in _split_top(tup[1:-1]) if a]
        calls.append((m.group(1), ret, argt, len(parts) - 3))
    structs = {}
    for m in re.finditer(r"(?:Base\.@kwdef\s+)?(?:mutable\s+)?struct\s+(\w+)\b(.*?)\n\s*end\b", code, flags=re.S):
        fields = []
        for piece in re.split(r"[;\n]", m.group(2)):
            mm = re.match(r"\s*(\w+)::([\w{},\s]+?)(?:\s*=.*)?$", piece)
            if mm:
                fields.append((mm.group(1), mm.group(2).strip()))
        structs[m.group(1)] = fields
    unbound = []
    mu = re.search(r"const\s+UNBOUND\s*=\s*\[(.*?)\]", code, flags=re.S)
    if mu:
        unbound = re.findall(r":(\w+)", mu.group(1))
    return calls, structs, unbound


# C type -> the Julia ccall types that are layout-compatible with it
def julia_ok(ctype, jl, struct_map):
    c = ctype.replace("const ", "").strip()
    scalars = {"int": {"Cint", "Int32"}, "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64", "Clonglong"}, "double": {"Float64", "Cdouble"},
               "float": {"Cfloat", "Float32"}, "uint64_t": {"UInt64"}}
    if c in scalars:
        return jl in scalars[c]
    if c == "void":
        return jl in ("Cvoid", "Nothing")
    if c == "char*":
        return jl in ("Cstring", "Ptr{UInt8}", "Ptr{Cchar}")
    if c == "tsat_handle**":
        return jl == "Ptr{Ptr{Cvoid}}"
    if c == "tsat_handle*":
        return jl == "Ptr{Cvoid}"
    if c == "void*":
        return bool(re.fullmatch(r"(Ptr|Ref)\{[\w{}]+\}", jl))                # any data pointer
    if c.endswith("*"):
        base = c[:-1]
        if base in scalars:
            return any(jl in (f"Ptr{{{t}}}", f"Ref{{{t}}}") for t in scalars[base])
        if base in struct_map:
            return jl in (f"Ptr{{{struct_map[base]}}}", f"Ref{{{struct_map[base]}}}")
    return False
