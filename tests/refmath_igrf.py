"""Independent NumPy transcription of the reference's field-table generation, written from the Julia
text line by line: kep_ECI (src/kep_ECI.jl:1-49), OrbitPlotter (src/OrbitPlotter.jl:1-52), the Euler orbit + GMST +
lat/long + IGRF + frame chain of magnetic_simulation (src/magnetic_toolbox.jl:33-106), igrf12 geocentric
(src/igrf.jl:70-274), Schmidt Legendre functions (src/legendre.jl:254-292) and their derivatives
(src/dlegendre.jl:221-309). Coefficients: the generated data header include/igrf12_2015_coeffs.h."""
import os
import re

import numpy as np

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "igrf12_2015_coeffs.h")


def _arr(name):
    txt = open(_HDR).read()
    m = re.search(name + r"\[[A-Z0-9_]+\] = \{([^}]*)\}", txt)
    return np.array([float(x) for x in m.group(1).split(",")])


G2015, GSV, H2015, HSV = _arr("IGRF12_G2015"), _arr("IGRF12_GSV"), _arr("IGRF12_H2015"), _arr("IGRF12_HSV")


def legendre_schmidt(phi, n_max):
    # src/legendre.jl:254-292, ph_term = false
    P = np.zeros((n_max + 1, n_max + 1))
    c = np.cos(phi)
    s = np.sqrt(1 - c**2)
    P[0, 0] = 1
    P[1, 0] = c
    P[1, 1] = s
    for n in range(2, n_max + 1):
        for m in range(0, n):
            aux = (n - m) * (n + m)
            a_nm = np.sqrt(((2 * n - 1) * (2 * n - 1)) / aux)
            b_nm = np.sqrt(((n + m - 1) * (n - m - 1)) / aux)
            P[n, m] = a_nm * c * P[n - 1, m] - b_nm * P[n - 2, m]
        P[n, n] = s * np.sqrt((2 * n - 1) / (2 * n)) * P[n - 1, n - 1]
    return P


def dlegendre_schmidt(phi, P):
    # src/dlegendre.jl:221-309 (the Schmidt variant forwards to it, :384-404), ph_term = false
    rows = P.shape[0]
    dP = np.zeros_like(P)
    Pp = np.pad(P, ((0, 0), (0, 2)))
    fact = -1 if (phi % (2 * np.pi)) > np.pi else 1
    for n in range(1, rows):
        for m in range(0, n + 1):
            if m == 0:
                aux = np.sqrt(n * (n + 1) / 2)
                dP[n, 0] = -(0.5 * aux) * Pp[n, 1] + (-0.5 * aux) * Pp[n, 1]
            elif m == 1:
                a_nm = 0.5 * np.sqrt(2 * n * (n + 1))
                b_nm = -0.5 * np.sqrt((n + 2) * (n - 1))
                dP[n, 1] = a_nm * Pp[n, 0] + b_nm * Pp[n, 2]
            elif n != m:
                a_nm = 0.5 * np.sqrt((n + m) * (n - m + 1))
                b_nm = -0.5 * np.sqrt((n + m + 1) * (n - m))
                dP[n, m] = a_nm * Pp[n, m - 1] + b_nm * Pp[n, m + 1]
            else:
                a_nm = 0.5 * np.sqrt((n + m) * (n - m + 1))
                dP[n, m] = a_nm * Pp[n, m - 1]
            dP[n, m] *= fact
    return dP


def igrf12(date, r, lam, Om):
    # src/igrf.jl:70-274 for 2015 <= date < 2020 (idx = 24, epoch 2015, dt = date - 2015, n_max = 13)
    theta = np.pi / 2 - lam
    phi = Om if Om >= 0 else 2 * np.pi + Om
    r = r / 1000
    dt = date - 2015
    n_max = 13
    P = legendre_schmidt(theta, n_max)
    dP = dlegendre_schmidt(theta, P)
    a = 6371.2
    sin_p, cos_p = np.sin(phi), np.cos(phi)
    ratio = a / r
    fact = ratio
    dVr = dVt = dVp = 0.0
    kg = kh = 0
    for n in range(1, n_max + 1):
        aux_r = aux_t = aux_p = 0.0
        Gnm = G2015[kg] + GSV[kg] * dt
        kg += 1
        aux_r += -(n + 1) / r * Gnm * P[n, 0]
        aux_t += Gnm * dP[n, 0]
        sin_m1, sin_m2 = 0.0, -sin_p
        cos_m1, cos_m2 = 1.0, cos_p
        for m in range(1, n + 1):
            sin_m = 2 * cos_p * sin_m1 - sin_m2
            cos_m = 2 * cos_p * cos_m1 - cos_m2
            Gnm = G2015[kg] + GSV[kg] * dt
            Hnm = H2015[kh] + HSV[kh] * dt
            kg += 1
            kh += 1
            GcHs = Gnm * cos_m + Hnm * sin_m
            GsHc = Gnm * sin_m - Hnm * cos_m
            aux_r += -(n + 1) / r * GcHs * P[n, m]
            aux_t += GcHs * dP[n, m]
            aux_p += (-m * GsHc * dP[n, m]) if theta == 0 else (-m * GsHc * P[n, m])
            sin_m2, sin_m1 = sin_m1, sin_m
            cos_m2, cos_m1 = cos_m1, cos_m
        fact *= ratio
        dVr += aux_r * fact
        dVp += aux_p * fact
        dVt += aux_t * fact
    dVr *= a
    dVp *= a
    dVt *= a
    x = 1 / r * dVt
    y = (-1 / r * dVp) if theta == 0 else (-1 / (r * np.sin(theta)) * dVp)
    z = dVr
    return np.array([x, y, z])


def _Rz_deg(angle):
    c, s = np.cos(np.deg2rad(angle)), np.sin(np.deg2rad(angle))
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])


def _Rx_deg(angle):
    c, s = np.cos(np.deg2rad(angle)), np.sin(np.deg2rad(angle))
    return np.array([[1, 0, 0], [0, c, s], [0, -s, c]])


def kep_ECI(kep, t0, GM):
    # src/kep_ECI.jl:1-35 (degrees; element 6 is used as the mean anomaly; the t0 term mixes rad into deg as written)
    A = np.array(kep, dtype=float).copy()
    A[5] = np.fmod(kep[5] + t0 * np.sqrt(GM / kep[1] ** 3), 360)
    E = A[5] / 180 * np.pi
    for _ in range(100):
        E = E - (E - A[0] * np.sin(E) - A[5] / 180 * np.pi) / (1 - A[0] * np.cos(E))
    nu = 2 * np.rad2deg(np.arctan2(np.sqrt(1 + A[0]) * np.sin(E / 2), np.sqrt(1 - A[0]) * np.cos(E / 2)))
    r_c = A[1] * (1 - A[0] * np.cos(E))
    o = r_c * np.array([np.cos(np.deg2rad(nu)), np.sin(np.deg2rad(nu)), 0])
    o_dot = np.sqrt(GM * A[1]) / r_c * np.array([-np.sin(E), np.sqrt(1 - A[0] ** 2) * np.cos(E), 0])
    M = _Rz_deg(-A[3]) @ _Rx_deg(-A[2]) @ _Rz_deg(-A[4])
    return M @ o, M @ o_dot


def orbit_rhs(x):
    # src/OrbitPlotter.jl:1-48 (only the terms that reach the return value; the J2 term is as written)
    r, v = x[:3], x[3:]
    GM = 3.986004418e14 * (1 / 1000) ** 3
    nr = np.linalg.norm(r)
    f_grav = GM / nr**2 * -r / nr
    J2 = 0.0010826359
    f_J2 = np.array([J2 * r[0] / nr**7 * (6 * r[2] - 1.5 * (r[0] ** 2 + r[1] ** 2)),
                     J2 * r[1] / nr**7 * (6 * r[2] - 1.5 * (r[0] ** 2 + r[1] ** 2)),
                     J2 * r[2] / nr**7 * (3 * r[2] - 4.5 * (r[0] ** 2 + r[1] ** 2))])
    return np.concatenate([v, f_grav + f_J2])


def Rz(theta):
    # src/magnetic_toolbox.jl:136-140
    return np.array([[np.cos(theta), np.sin(theta), 0], [-np.sin(theta), np.cos(theta), 0], [0, 0, 1]])


def magnetic_simulation(kep, t0, tf, N, MJD, GM, r_igrf_km, date=2019):
    # src/magnetic_toolbox.jl:33-106
    r0, v0 = kep_ECI(kep, t0, GM)
    u = np.concatenate([r0, v0])
    dt = (tf - t0) / N
    pos = np.zeros((2 * N + 1, 3))
    for i in range(2 * N + 1):           # Euler(), adaptive = false, tspan = (t0, 2 tf)
        pos[i] = u[:3]
        u = u + dt * orbit_rhs(u)
    t = t0 + dt * np.arange(2 * N + 1)
    GMST = (280.4606 + 360.9856473 * (t / 24 / 60 / 60 + MJD) - 51544.5) / 180 * np.pi
    B = np.zeros((2 * N, 3))
    NED_to_ENU = np.array([[0, 1, 0], [1, 0, 0], [0, 0, -1]])
    for i in range(2 * N - 1):           # the last row stays zero
        pe = Rz(GMST[i]) @ pos[i]
        lat = np.arcsin(pe[2] / np.linalg.norm(pe))
        lon = np.arctan2(pe[1], pe[0])
        b = igrf12(date, r_igrf_km * 1000, lat, lon) / 1.0e9
        R = np.array([[-np.sin(lon), -np.sin(lat) * np.cos(lon), np.cos(lat) * np.cos(lon)],
                      [np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat) * np.sin(lon)],
                      [0, np.cos(lat), np.sin(lat)]])
        B[i] = Rz(GMST[i]).T @ R @ NED_to_ENU @ b
    return B, pos
