"""CPU tier: the C-ABI library loads and exports every symbol include/tortoise_hip.h declares; struct layouts
agree between C and ctypes; and without a GPU the product fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "tortoise_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsat_[a-z_]+)\s*\(", txt)))


def test_header_and_ctypes_prototypes_agree(pkg):
    assert header_functions() == sorted(pkg._abi.PROTOTYPES)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._abi.load()          # raises if libtortoise_hip.so is not built — there is no fallback
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.tsat_version() == 300


def test_struct_layouts_and_defaults(pkg, ol):
    abi = pkg._abi
    lib = abi.load()
    o = abi.Options()
    lib.tsat_default_options(C.byref(o))
    ref = ol.default_options()      # the oracle fills the same struct from its own C++ translation unit
    for name, _ in abi.Options._fields_:
        assert getattr(o, name) == getattr(ref, name), name
    assert (o.integrator, o.max_outer, o.max_inner) == (3, 20, 50)     # src/TortoiseSat.jl:146,195-196
    assert o.u_scale == 1e-2                                           # src/DerivFunction.jl:37
    assert C.sizeof(abi.Stats) == 64 and abi.STATS_DTYPE.itemsize == 64
    # the widened entry points: defaults are the reference's script constants, oracle and library agree field by field
    t, tr = abi.TvlqrOptions(), ol.tvlqr_default_options()
    lib.tsat_tvlqr_default_options(C.byref(t))
    for name, _ in abi.TvlqrOptions._fields_:
        assert getattr(t, name) == getattr(tr, name), name
    assert (t.linearize_dt_sq, t.min_steps, t.w_tol, t.angle_tol) == (1, 10, 0.05, 0.08727)   # src/monte_carlo.jl:70-71,251
    assert C.sizeof(abi.TvlqrOptions) == 80 and abi.TVLQR_STATS_DTYPE.itemsize == 32
    bt, br = abi.BtableOptions(), ol.BtableOptions()
    lib.tsat_btable_default_options(C.byref(bt))
    ol.load().orc_btable_default_options(C.byref(br))
    for name, _ in abi.BtableOptions._fields_:
        assert getattr(bt, name) == getattr(br, name), name
    assert (bt.n_half, bt.mjd, bt.r_igrf_km, bt.date) == (5000, 58155.0, 6771.0, 2019.0)      # src/TortoiseSat.jl:44,61
    assert bt.gm == 3.986004418e14 * 1e-9 and C.sizeof(abi.BtableOptions) == 40


def test_no_gpu_means_error_not_fallback(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = pkg._abi.load()
    h = C.c_void_p()
    rc = lib.tsat_create(C.byref(h), 0)
    assert rc < 0 and not h.value
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.trajopt.AugmentedLagrangianSolver(None, None)


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under the package (or the library sources) may reference it"""
    pdir = os.path.join(ROOT, "tortoisesat.jl_amd")
    for dp, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "orc_" not in txt, os.path.join(dp, f)
