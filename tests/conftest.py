"""Shared test plumbing. CPU tier: `pytest -m "not gpu"`; GPU tier: `pytest -m gpu` (needs an MI355X)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def pkg():
    from tsat_loader import load_package

    return load_package()


@pytest.fixture(scope="session")
def ol():
    import oracle_lib

    oracle_lib.build()
    oracle_lib.load()
    return oracle_lib


class Emu:
    """ctypes binding of tests/emu/libtsat_emu.so (the HIP kernel source run by 64 host threads per wave)."""

    def __init__(self, abi, name="libtsat_emu.so"):
        d = os.path.join(ROOT, "tests", "emu")
        subprocess.check_call(["make", "-C", d, name], stdout=subprocess.DEVNULL)
        self.lib = C.CDLL(os.path.join(d, name))
        self.abi = abi

    def solve(self, batch, opts, trace_rows=0):
        T, N = batch.T, batch.N
        o = opts.copy()
        o.n_knots, o.n_tab = N, batch.n_tab
        X = np.zeros((T, N, 7)); U = np.zeros((T, N - 1, 3)); K = np.zeros((T, N - 1, 7, 3))
        stats = np.zeros(T, dtype=self.abi.STATS_DTYPE)
        trace = np.zeros((T, max(trace_rows, 1), 8))
        d = self.abi.as_dp
        rc = self.lib.emu_solve_batch(
            C.byref(o), C.c_int64(T), C.c_int64(batch.Btab.shape[0]), d(batch.x0), d(batch.xf), d(batch.Btab),
            self.abi.as_ip(batch.btab_idx), d(batch.tau0), d(batch.dtau), d(batch.dt), d(batch.Jmat), d(batch.Qd),
            d(batch.Qfd), d(batch.Rd), d(batch.ulo), d(batch.uhi), d(batch.U0), d(X), d(U), d(K),
            stats.ctypes.data_as(C.c_void_p), d(trace) if trace_rows else None, C.c_int(trace_rows),
            None if batch.n_knots is None else self.abi.as_ip(np.ascontiguousarray(batch.n_knots, dtype=np.int32)))
        if rc != 0:
            raise RuntimeError(f"emu_solve_batch rc={rc}")
        return dict(X=X, U=U, K=K, stats=stats, trace=trace)


    def tvlqr(self, batch, X, U, Qd, Qfd, Rd, x0_sim, noise=None, opts=None, noise_ids=None):
        T, N = batch.T, batch.N
        if opts is None:
            opts = self.abi.TvlqrOptions(0, 0, 1, 10, 1e-2, 0.05, 0.08727, 0, 0, 0, (0.38 * np.pi / 180) ** 2, (np.pi / 180) ** 2, 1e-10)
        o = opts
        o.n_knots, o.n_tab = N, batch.n_tab
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        X, U, Qd, Qfd, Rd, x0_sim = c(X), c(U), c(Qd), c(Qfd), c(Rd), c(x0_sim)
        noise = None if noise is None else c(noise)
        Xs = np.zeros((T, N, 7)); Us = np.zeros((T, N - 1, 3)); K = np.zeros((T, N - 1, 6, 3))
        st = np.zeros(T, dtype=self.abi.TVLQR_STATS_DTYPE)
        d = self.abi.as_dp
        rc = self.lib.emu_tvlqr_batch(C.byref(o), C.c_int64(T), C.c_int64(batch.Btab.shape[0]), d(X), d(U), d(batch.xf),
                                      d(batch.Btab), self.abi.as_ip(batch.btab_idx), d(batch.tau0), d(batch.dtau), d(batch.dt),
                                      d(batch.Jmat), d(Qd), d(Qfd), d(Rd), d(x0_sim), d(noise), d(Xs), d(Us), d(K),
                                      st.ctypes.data_as(C.c_void_p),
                                      None if batch.n_knots is None else self.abi.as_ip(np.ascontiguousarray(batch.n_knots, dtype=np.int32)),
                                      None if noise_ids is None else np.ascontiguousarray(noise_ids, dtype=np.int64).ctypes.data_as(C.POINTER(C.c_int64)))
        if rc != 0:
            raise RuntimeError(f"emu_tvlqr_batch rc={rc}")
        return dict(X_sim=Xs, U_sim=Us, K=K, stats=st)


    def mpc(self, batch, opts, n_steps, plant_integrator=4):
        T, N = batch.T, batch.N
        o = opts.copy()
        o.n_knots, o.n_tab = N, batch.n_tab
        Xh = np.zeros((T, n_steps + 1, 7)); Uh = np.zeros((T, n_steps, 3))
        X = np.zeros((T, N, 7)); U = np.zeros((T, N - 1, 3))
        st = np.zeros(T, dtype=self.abi.STATS_DTYPE)
        d = self.abi.as_dp
        rc = self.lib.emu_mpc_batch(C.byref(o), C.c_int64(T), C.c_int64(batch.Btab.shape[0]), d(batch.x0), d(batch.xf), d(batch.Btab),
                                    self.abi.as_ip(batch.btab_idx), d(batch.tau0), d(batch.dtau), d(batch.dt), d(batch.Jmat),
                                    d(batch.Qd), d(batch.Qfd), d(batch.Rd), d(batch.ulo), d(batch.uhi), d(batch.U0),
                                    C.c_int32(n_steps), C.c_int32(plant_integrator), d(Xh), d(Uh), st.ctypes.data_as(C.c_void_p),
                                    d(X), d(U), None if batch.n_knots is None else self.abi.as_ip(np.ascontiguousarray(batch.n_knots, dtype=np.int32)))
        if rc != 0:
            raise RuntimeError(f"emu_mpc_batch rc={rc}")
        return dict(X_hist=Xh, U_hist=Uh, stats=st, X=X, U=U)

    def btable(self, kep, t0, tf, N, mjd=58155.0, gm=3.986004418e5, r_igrf_km=6771.0, date=2019.0):
        kep = np.ascontiguousarray(kep, dtype=np.float64)
        T = kep.shape[0]
        t0 = np.ascontiguousarray(np.broadcast_to(t0, (T,)), dtype=np.float64)
        tf = np.ascontiguousarray(np.broadcast_to(tf, (T,)), dtype=np.float64)
        o = self.abi.BtableOptions(int(N), 0, mjd, gm, r_igrf_km, date)
        B = np.zeros((T, 2 * N, 3)); pos = np.zeros((T, 2 * N + 1, 3))
        d = self.abi.as_dp
        self.lib.emu_btable_batch(C.byref(o), C.c_int64(T), d(kep), d(t0), d(tf), d(B), d(pos))
        return B, pos

    def horizon(self, Btab, dt_row, cutoff):
        Btab = np.ascontiguousarray(Btab, dtype=np.float64)
        T, n = Btab.shape[0], Btab.shape[1]
        dt_row = np.ascontiguousarray(np.broadcast_to(dt_row, (T,)), dtype=np.float64)
        cutoff = np.ascontiguousarray(np.broadcast_to(cutoff, (T,)), dtype=np.float64)
        idx = np.zeros(T, dtype=np.int32); cond = np.zeros(T)
        d = self.abi.as_dp
        self.lib.emu_horizon_batch(C.c_int64(T), C.c_int32(n), d(Btab), d(dt_row), d(cutoff), self.abi.as_ip(idx), d(cond))
        return idx, cond


@pytest.fixture(scope="session")
def emu(pkg):
    return Emu(pkg._abi)


@pytest.fixture(scope="session")
def emu_dense(pkg):
    """the dense build of the solve kernel (two wavefronts per SIMD on the GPU): same source, -DTSAT_DENSE"""
    return Emu(pkg._abi, "libtsat_emu_dense.so")


@pytest.fixture(scope="session")
def emu_packed(pkg):
    """the packed build (PK_G trajectories per wavefront share the forward sweeps and the Riccati recursion), -DTSAT_PACKED"""
    return Emu(pkg._abi, "libtsat_emu_packed.so")


@pytest.fixture(scope="session")
def emu_mixed(pkg):
    """the mixed-precision build of the solve kernel (options.precision = 32): float linearisation, everything else double, -DTSAT_JAC32"""
    return Emu(pkg._abi, "libtsat_emu_mixed.so")


@pytest.fixture(scope="session")
def emu_packed8(pkg):
    """the packed build with eight trajectories per wavefront (tsat_kernels_packed8.hip)"""
    return Emu(pkg._abi, "libtsat_emu_packed8.so")


@pytest.fixture(scope="session")
def emu_packed4w(pkg):
    """four trajectories per wavefront at one wavefront per SIMD (tsat_kernels_packed4w.hip)"""
    return Emu(pkg._abi, "libtsat_emu_packed4w.so")


@pytest.fixture(scope="session")
def emu_packed8w(pkg):
    """eight trajectories per wavefront at one wavefront per SIMD: twelve-knot record ring, double-buffered forward chunks (tsat_kernels_packed8w.hip)"""
    return Emu(pkg._abi, "libtsat_emu_packed8w.so")


@pytest.fixture(scope="session")
def emu_packed16w(pkg):
    """sixteen trajectories per wavefront, four line-search candidates each (tsat_kernels_packed16w.hip)"""
    return Emu(pkg._abi, "libtsat_emu_packed16w.so")


@pytest.fixture(scope="session")
def emu_packed16w_mixed(pkg):
    """the mixed-precision sixteen-per-wavefront build: all sixteen knots of a pass in the record ring (tsat_kernels_packed16w_mixed.hip)"""
    return Emu(pkg._abi, "libtsat_emu_packed16w_mixed.so")


@pytest.fixture(scope="session")
def emu_packed_mixed(pkg):
    """the mixed-precision packed build (tsat_kernels_packed_mixed.hip)"""
    return Emu(pkg._abi, "libtsat_emu_packed_mixed.so")


def oracle_options(ol, **kw):
    o = ol.default_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


COUNT_FIELDS = ("status", "outer_iters", "inner_iters", "ls_trials", "n_backward", "bp_restarts", "fp_fails")


def parity_errors(ref, got):
    """per-trajectory (|dX|_inf, |dU|_inf / max(1, |U_ref|_inf)) — the two numbers the fp64 parity bar is stated on"""
    ax = tuple(range(1, ref["X"].ndim))
    dX = np.max(np.abs(ref["X"] - got["X"]), axis=ax)
    scale = np.maximum(1.0, np.max(np.abs(ref["U"]), axis=ax)) if ref["U"].size else np.ones_like(dX)
    dU = (np.max(np.abs(ref["U"] - got["U"]), axis=ax) if ref["U"].size else np.zeros_like(dX)) / scale
    return dX, dU


def assert_same_solution(ref, got, tol=1e-9, counts=True):
    """fp64 parity bar: identical iteration / line-search / restart counts, and on EVERY
    trajectory |dX|_inf < 1e-9 (states are O(1): unit quaternion, rates << 1 rad/s) and |dU|_inf < 1e-9 max(1, |U|_inf)
    — the controls are stored in units of u_scale = 0.01 A m^2 and reach the box |u| <= 19 on the Monte-Carlo workloads
    (|u| <= 1 on the single slew), so the bar on U is 1e-9 of the trajectory's own control scale, never looser than 1e-9
    times the bound."""
    if counts:
        for f in COUNT_FIELDS:
            assert np.array_equal(ref["stats"][f], got["stats"][f]), f
    dX, dU = parity_errors(ref, got)
    assert np.max(dX) < tol, (float(np.max(dX)), int(np.argmax(dX)))
    assert np.max(dU) < tol, (float(np.max(dU)), int(np.argmax(dU)))
    np.testing.assert_allclose(got["stats"]["cost"], ref["stats"]["cost"], rtol=1e-9)
    np.testing.assert_allclose(got["stats"]["c_max"], ref["stats"]["c_max"], rtol=1e-7, atol=1e-10)
