"""NumPy-free transcription of the reference's SECOND IGRF-12 algorithm, `igrf12syn` (src/igrf.jl:335-534, a Julia port
of the IAGA FORTRAN `igrf12syn`), for the tests only.

It is the one numerical statement the reference makes about its own field model: `igrf12` and `igrf12syn` agree to about
0.01 nT (src/igrf.jl:283-287). The two share neither the recursion (Schmidt quasi-normal p/q built in one flat loop here,
separate Legendre / derivative tables there) nor the coefficient table (the flat `gh` array of src/igrf12syn_coefs.jl,
committed as data in tests/golden/igrf12syn_gh.npz by tools/extract_igrf12syn_coeffs.py, against the G/H matrices of
src/igrf12_coefs.jl), so agreement pins the oracle and the GPU tables to something this repository did not write.
"""
import math
import os

import numpy as np

GH = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "igrf12syn_gh.npz"))["gh"]


def igrf12syn(isv, date, itype, alt, colat, elong):
    """(x north, y east, z down, f) in nT (nT/yr for isv = 1); itype 2: geocentric (alt = radius, km), colat / elong deg."""
    if date < 1900 or date > 2025:
        raise ValueError("IGRF-12 is valid in [1900, 2025)")
    gh = GH
    cl = [0.0] * 14
    sl = [0.0] * 14
    p = [0.0] * 106
    q = [0.0] * 106
    if date < 2015:
        t = 0.2 * (date - 1900)
        ll = int(math.floor(t))
        t = t - ll
        if date < 1995:
            nmx, nc, ll, kmx = 10, 120, 120 * ll, 66
        else:
            nmx, nc = 13, 195
            ll = 120 * 19 + nc * int(math.floor(0.2 * (date - 1995)))
            kmx = 105
        tc = 1 - t
        if isv == 1:
            t, tc = 0.2, -0.2
    else:
        t, tc = date - 2015, 1.0
        if isv == 1:
            t, tc = 1.0, 0.0
        ll, nmx, nc, kmx = 3060, 13, 195, 105
    r = alt
    ct, st = math.cos(colat * math.pi / 180), math.sin(colat * math.pi / 180)
    cl[1], sl[1] = math.cos(elong * math.pi / 180), math.sin(elong * math.pi / 180)
    cd, sd = 1.0, 0.0
    l, m, n = 1, 1, 0
    fn = gn = 0
    if itype != 2:                         # geodetic -> geocentric on the WGS84 spheroid
        a2, b2 = 40680631.6, 40408296.0
        one, two = a2 * st**2, b2 * ct**2
        three = one + two
        rho = math.sqrt(three)
        r = math.sqrt(alt * (alt + 2 * rho) + (a2 * one + b2 * two) / three)
        cd = (alt + rho) / r
        sd = (a2 - b2) / rho * ct * st / r
        one = ct
        ct = ct * cd - st * sd
        st = st * cd + one * sd
    ratio = 6371.2 / r
    rr = ratio**2
    p[1], p[3], q[1], q[3] = 1.0, st, 0.0, ct
    x = y = z = 0.0
    for k in range(2, kmx + 1):
        if n < m:
            m, n = 0, n + 1
            rr *= ratio
            fn, gn = n, n - 1
        fm = m
        if m == n:
            if k != 3:
                one = math.sqrt(1 - 0.5 / fm)
                j = k - n - 1
                p[k] = one * st * p[j]
                q[k] = one * (st * q[j] + ct * p[j])
                cl[m] = cl[m - 1] * cl[1] - sl[m - 1] * sl[1]
                sl[m] = sl[m - 1] * cl[1] + cl[m - 1] * sl[1]
        else:
            gmm = m * m
            one = math.sqrt(fn * fn - gmm)
            two = math.sqrt(gn * gn - gmm) / one
            three = (fn + gn) / one
            i = k - n
            j = i - n + 1
            p[k] = three * ct * p[i] - two * p[j]
            q[k] = three * (ct * q[i] - st * p[i]) - two * q[j]
        lm = ll + l                         # 1-based into gh
        one = (tc * gh[lm - 1] + t * gh[lm + nc - 1]) * rr
        if m != 0:
            two = (tc * gh[lm] + t * gh[lm + nc]) * rr
            three = one * cl[m] + two * sl[m]
            x += three * q[k]
            z -= (fn + 1) * three * p[k]
            if st != 0:
                y += (one * sl[m] - two * cl[m]) * fm * p[k] / st
            else:
                y += (one * sl[m] - two * cl[m]) * q[k] * ct
            l += 2
        else:
            x += one * q[k]
            z -= (fn + 1) * one * p[k]
            l += 1
        m += 1
    one = x
    x = x * cd + z * sd
    z = z * cd - one * sd
    return x, y, z, math.sqrt(x * x + y * y + z * z)


def igrf12_geocentric(date, r_m, lat, lon):
    """same call shape as refmath_igrf.igrf12 / the oracle: radius in m, latitude / longitude in rad -> NED field in nT"""
    x, y, z, _ = igrf12syn(0, date, 2, r_m / 1000.0, 90.0 - math.degrees(lat), math.degrees(lon))
    return np.array([x, y, z])
