"""Field tables on the GPU — the first step of the reference's per-run setup.

``B_ECI, pos, vel = magnetic_simulation(p, t0, tf, N, mag_field)`` (src/magnetic_toolbox.jl:33-106) propagates the orbit
from Keplerian elements with a fixed-step Euler integrator, rotates to ECEF with GMST, evaluates IGRF-12 at every
sample and rotates the field back to ECI; the result has 2N rows, the last one left zero. ``magnetic_simulation`` here
does that for T orbits at once through ``tsat_btable_batch``.
"""
import ctypes as C

import numpy as np

from . import _abi


def magnetic_simulation(solver, kep, t0, tf, N, mjd=58155.0, gm=3.986004418e5, alt=400.0, R_E=6371.0, date=2019.0,
                        want_pos=True, host=True):
    """kep (T,6) = [e, a (km), i, RAAN, argp, anomaly] in degrees (src/TortoiseSat.jl:35-42); t0, tf scalars or (T,).
    Returns (B_ECI (T, 2N, 3) Tesla, pos (T, 2N+1, 3) km or None). ``host=False``: the tables are not downloaded (B_ECI is
    None) — they stay on the device, where ``horizon.condition_based_time(solver, None, ...)`` and an upload of a batch with
    ``Btab = None`` pick them up (the Monte-Carlo's tables never cross PCIe unless the caller wants to keep them)."""
    lib = _abi.load()
    kep = np.ascontiguousarray(np.atleast_2d(kep), dtype=np.float64)
    if kep.shape[1] != 6:
        raise ValueError("kep must be (T, 6)")
    T = kep.shape[0]
    t0 = np.ascontiguousarray(np.broadcast_to(np.asarray(t0, dtype=np.float64), (T,)))
    tf = np.ascontiguousarray(np.broadcast_to(np.asarray(tf, dtype=np.float64), (T,)))
    o = _abi.BtableOptions()
    lib.tsat_btable_default_options(C.byref(o))
    o.n_half, o.mjd, o.gm, o.r_igrf_km, o.date = int(N), float(mjd), float(gm), float(alt + R_E), float(date)
    B = np.empty((T, 2 * N, 3)) if host else None
    pos = np.empty((T, 2 * N + 1, 3)) if want_pos else None
    rc = lib.tsat_btable_batch(solver._h, C.byref(o), T, _abi.as_dp(kep), _abi.as_dp(t0), _abi.as_dp(tf), _abi.as_dp(B),
                               _abi.as_dp(pos))
    solver._check(rc, "tsat_btable_batch")
    return B, pos


def attach_igrf_tables(solver, batch, kep=None, chunk=4096):
    """Replace the field tables of a workload batch by IGRF-12 tables generated on the GPU for its orbits — what the
    reference does before every solve (``magnetic_simulation(A[i,:], t0, t_final, N, ...)``, src/monte_carlo.jl:149): one
    table per row of ``kep`` (default ``batch.meta["kep"]``), sampled over the slew horizon N dt with one row per knot step
    (2N rows, the last one zero; the solve reads rows 0 .. N, so only the first N + 8 travel on). One orbit for the whole
    batch, or one per trajectory."""
    kep = np.atleast_2d(np.asarray(batch.meta["kep"] if kep is None else kep, dtype=np.float64))
    if kep.shape[0] not in (1, batch.T):
        raise ValueError("one orbit for the batch or one per trajectory")
    N, dt = batch.N, float(batch.dt[0])
    if not np.all(batch.dt == dt):
        raise ValueError("attach_igrf_tables needs a common step")
    rows = min(2 * N, N + 8)
    parts = [np.ascontiguousarray(magnetic_simulation(solver, kep[a:a + chunk], 0.0, N * dt, N, want_pos=False)[0][:, :rows])
             for a in range(0, kep.shape[0], chunk)]
    batch.Btab = np.ascontiguousarray(np.concatenate(parts)) if len(parts) > 1 else parts[0]
    batch.n_tab = rows
    batch.btab_idx = np.zeros(batch.T, np.int32) if kep.shape[0] == 1 else np.arange(batch.T, dtype=np.int32)
    batch.tau0[:] = 0.0
    batch.dtau[:] = 1.0
    batch.meta = dict(batch.meta, field="IGRF-12 (tsat_btable_batch)")
    return batch
