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
                        want_pos=True):
    """kep (T,6) = [e, a (km), i, RAAN, argp, anomaly] in degrees (src/TortoiseSat.jl:35-42); t0, tf scalars or (T,).
    Returns (B_ECI (T, 2N, 3) Tesla, pos (T, 2N+1, 3) km or None)."""
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
    B = np.empty((T, 2 * N, 3))
    pos = np.empty((T, 2 * N + 1, 3)) if want_pos else None
    rc = lib.tsat_btable_batch(solver._h, C.byref(o), T, _abi.as_dp(kep), _abi.as_dp(t0), _abi.as_dp(tf), _abi.as_dp(B),
                               _abi.as_dp(pos))
    solver._check(rc, "tsat_btable_batch")
    return B, pos
