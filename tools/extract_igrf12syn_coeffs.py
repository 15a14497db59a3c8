#!/usr/bin/env python3
"""Extracts the flat IGRF-12 coefficient array `gh` used by the reference's SECOND field algorithm (`igrf12syn`) into a
data fixture for the tests.

The numbers are the IAGA/NOAA IGRF-12 constants as laid out in the FORTRAN `igrf12.f` (one block of g/h values per
5-year model, 1900 ... 2015, then the 2015-20 secular variation); the reference carries them in
src/igrf12syn_coefs.jl as `gh_igrf12`, a table that is independent of the `G_igrf12` / `H_igrf12` matrices its first
algorithm (`igrf12`) and this repository's kernels read. Only the numbers are written (tests/golden/igrf12syn_gh.npz).

    python tools/extract_igrf12syn_coeffs.py /root/reference/src/igrf12syn_coefs.jl
"""
import os
import re
import sys

import numpy as np


def main(path):
    text = open(path).read()
    body = text[text.index("const gh_igrf12 = [") + len("const gh_igrf12 = ["):]
    body = body[: body.index("\n         ]")]
    vals = []
    for line in body.splitlines():
        line = line.split("#")[0]
        for tok in line.split(","):
            tok = tok.strip()
            if not tok:
                continue
            m = re.fullmatch(r"zeros\((\d+)\)\.\.\.", tok)
            if m:
                vals.extend([0.0] * int(m.group(1)))
            else:
                vals.append(float(tok))
    gh = np.array(vals, dtype=np.float64)
    # 19 models of degree 10 (120 values), 5 of degree 13 (195) incl. 2015, + the secular-variation block (195)
    assert gh.size == 19 * 120 + 5 * 195 + 195, gh.size
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "igrf12syn_gh.npz")
    np.savez_compressed(out, gh=gh)
    print(f"{gh.size} coefficients -> {out}; g10(1900) = {gh[0]}, g10(2015) = {gh[3060]}, dg10/dt = {gh[3255]}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/igrf12syn_coefs.jl")
