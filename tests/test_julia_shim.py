"""CPU tier: static verification of the bindings a maintainer of the reference would add.

julia/TortoiseHIP.jl replaces `TrajectoryOptimization.solve!(sat, solver)` (src/TortoiseSat.jl:190-203) and its neighbours by
`ccall`s into libtortoise_hip.so. There is no Julia toolchain in the image, so the shim cannot be run; what a wrong binding
would break — a missing symbol, an arity or type mismatch that corrupts the call frame, a struct whose fields are in another
order or of another width — is checked here against include/tortoise_hip.h by parsing both files (tests/cdecl.py). The Python
ctypes prototypes (tortoisesat.jl_amd/_abi.py) are held to the same header."""
import ctypes as C
import re

import cdecl

STRUCT_MAP = {"tsat_options": "Options", "tsat_stats": "Stats", "tsat_tvlqr_options": "TvlqrOptions",
              "tsat_tvlqr_stats": "TvlqrStats", "tsat_btable_options": "BtableOptions"}
JL_SIZE = {"Int32": 4, "Cint": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "Cdouble": 8, "Float32": 4, "Cfloat": 4}
C_SIZE = {"int32_t": 4, "int": 4, "int64_t": 8, "uint64_t": 8, "double": 8, "float": 4}


def test_header_parses_completely():
    funcs, structs = cdecl.parse_header()
    src = re.sub(r"/\*.*?\*/", " ", open(cdecl.HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(tsat_\w+)\s*\(", src))
    assert declared == set(funcs), declared ^ set(funcs)                 # the parser sees every declaration
    assert set(structs) == set(STRUCT_MAP)
    assert all(re.fullmatch(r"(const )?\w+\*{0,2}", t) for _, a in funcs.values() for t in a)


def test_every_ccall_matches_the_header():
    funcs, _ = cdecl.parse_header()
    calls, _, unbound = cdecl.parse_julia()
    assert len(calls) >= 30
    for name, ret, argt, nvals in calls:
        assert name in funcs, f"ccall of :{name}, which the header does not declare"
        cret, cargs = funcs[name]
        assert len(argt) == len(cargs), f":{name}: {len(argt)} argument types in the ccall tuple, {len(cargs)} in the header"
        assert nvals == len(cargs), f":{name}: {nvals} values passed for {len(cargs)} parameters"
        assert cdecl.julia_ok(cret, ret, STRUCT_MAP), f":{name}: return type {ret} for C `{cret}`"
        for i, (c, j) in enumerate(zip(cargs, argt)):
            assert cdecl.julia_ok(c, j, STRUCT_MAP), f":{name}: argument {i + 1} is `{c}` in the header, {j} in the ccall"
    bound = {c[0] for c in calls}
    missing = set(funcs) - bound - set(unbound)
    assert not missing, f"header functions with neither a binding nor an UNBOUND entry in TortoiseHIP.jl: {sorted(missing)}"
    assert not (set(unbound) & bound) and set(unbound) <= set(funcs)


def test_julia_structs_mirror_the_c_structs():
    _, cstructs = cdecl.parse_header()
    _, jstructs, _ = cdecl.parse_julia()
    for cname, jname in STRUCT_MAP.items():
        cf, jf = cstructs[cname], jstructs[jname]
        assert [n for n, _ in cf] == [n for n, _ in jf], f"{jname}: field names / order differ from {cname}"
        for (n, ct), (_, jt) in zip(cf, jf):
            assert cdecl.julia_ok(ct, jt, STRUCT_MAP), f"{jname}.{n}: {jt} for C `{ct}`"
        # same natural-alignment layout: offsets computed from the field widths agree, and so does the padded size
        def layout(widths):
            off, offs = 0, []
            for w in widths:
                off = (off + w - 1) // w * w
                offs.append(off); off += w
            return offs, (off + 7) // 8 * 8
        assert layout([C_SIZE[t] for _, t in cf]) == layout([JL_SIZE[t] for _, t in jf])


def test_python_prototypes_match_the_header(pkg):
    """the ctypes prototypes the product binds (tortoisesat.jl_amd/_abi.py) against the same header: names, arity, widths"""
    abi = pkg._abi
    funcs, cstructs = cdecl.parse_header()
    assert set(abi.PROTOTYPES) == set(funcs)
    width = {"int": 4, "int32_t": 4, "int64_t": 8, "double": 8, "float": 4, "uint64_t": 8}
    for name, (cret, cargs) in funcs.items():
        res, args = abi.PROTOTYPES[name]
        assert len(args) == len(cargs), name
        for c, a in zip(cargs, args):
            c = c.replace("const ", "")
            if c in width:
                assert C.sizeof(a) == width[c] and not hasattr(a, "contents"), (name, c, a)
            else:
                assert c.endswith("*") and C.sizeof(a) == C.sizeof(C.c_void_p), (name, c, a)      # a pointer type
        cr = cret.replace("const ", "")
        assert (res is None) == (cr == "void")
        if cr in width:
            assert C.sizeof(res) == width[cr], name
    for cname, cls in (("tsat_options", abi.Options), ("tsat_stats", abi.Stats), ("tsat_tvlqr_options", abi.TvlqrOptions),
                       ("tsat_btable_options", abi.BtableOptions)):
        assert [n for n, _ in cstructs[cname]] == [f[0] for f in cls._fields_], cname
        assert [width[t] for _, t in cstructs[cname]] == [C.sizeof(f[1]) for f in cls._fields_], cname
    assert [n for n, _ in cstructs["tsat_tvlqr_stats"]] == list(abi.TVLQR_STATS_DTYPE.names)
    assert [n for n, _ in cstructs["tsat_stats"]] == list(abi.STATS_DTYPE.names)


def test_the_checker_is_not_vacuous(tmp_path):
    """seeded faults in a copy of the shim are caught: a widened integer argument, a dropped argument, swapped struct fields"""
    funcs, cstructs = cdecl.parse_header()
    src = open(cdecl.JULIA).read()

    def faults(text):
        p = tmp_path / "mutant.jl"
        p.write_text(text)
        calls, jstructs, _ = cdecl.parse_julia(str(p))
        bad = []
        for name, ret, argt, nvals in calls:
            cret, cargs = funcs[name]
            if len(argt) != len(cargs) or nvals != len(cargs) or not cdecl.julia_ok(cret, ret, STRUCT_MAP) \
                    or not all(cdecl.julia_ok(c, j, STRUCT_MAP) for c, j in zip(cargs, argt)):
                bad.append(name)
        for cname, jname in STRUCT_MAP.items():
            if [n for n, _ in cstructs[cname]] != [n for n, _ in jstructs[jname]]:
                bad.append(jname)
        return bad

    assert faults(src) == []
    m1 = src.replace("(Ptr{Cvoid}, Int64, Int32, Int32, Int64, Int32)", "(Ptr{Cvoid}, Int64, Int64, Int32, Int64, Int32)", 1)
    assert m1 != src and "tsat_batch_reserve" in faults(m1)
    m2 = src.replace("(Ptr{Cvoid}, Ref{Options}, Ptr{Cfloat}), s.handle, o, C_NULL)", "(Ptr{Cvoid}, Ref{Options}), s.handle, o)", 1)
    assert m2 != src and "tsat_batch_run" in faults(m2)
    m3 = src.replace("    cost_tol::Float64 = 1e-4\n    grad_tol::Float64 = 1e-5\n", "    grad_tol::Float64 = 1e-5\n    cost_tol::Float64 = 1e-4\n", 1)
    assert m3 != src and "Options" in faults(m3)
