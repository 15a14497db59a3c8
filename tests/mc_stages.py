"""The four numeric stages of tortoisesat.jl_amd/monte_carlo.py served by the CPU oracle (test infrastructure)."""
import numpy as np

from conftest import oracle_options


class OracleStages:
    def __init__(self, ol, nthreads=8):
        self.ol, self.nthreads = ol, nthreads

    def magnetic_simulation(self, kep, t0, tf, N, s):
        return self.ol.btable_batch(kep, t0, tf, N, mjd=s.mjd, gm=s.GM, r_igrf_km=s.alt + s.R_E, date=s.igrf_date, want_pos=False)[0]

    def condition_based_time(self, B, dt_row, cutoff):
        return self.ol.horizon_batch(B, dt_row, cutoff)[0]

    def solve(self, batch, s):
        o = oracle_options(self.ol, max_outer=s.outer, max_inner=s.inner, dj_counter_limit=s.dJ_counter_limit)
        o.error_state = 1
        return self.ol.solve_batch(batch, o, nthreads=self.nthreads, want_K=False)

    def attitude_simulation(self, batch, X, U, x0_sim, Qd, Qfd, Rd, noise_seed, noise_ids, s):
        o = self.ol.tvlqr_default_options()
        o.w_tol, o.angle_tol = s.w_tol, s.angle_tol
        o.noise_mode, o.noise_seed = 1, noise_seed
        o.rate_as_written = int(bool(getattr(s, "rate_as_written", False)))
        return self.ol.tvlqr_batch(batch, X, U, Qd, Qfd, Rd, x0_sim, opts=o, nthreads=self.nthreads, noise_ids=noise_ids)
