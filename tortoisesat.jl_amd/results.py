"""Result files of the Monte-Carlo (src/monte_carlo.jl:334-343).

The script writes, per experiment of `n` trials, ``{n}_A.h5`` (dataset "A") and per trial ``{n}_states_{i}.h5``
("one_state" and "states" = the closed-loop states, the script writes the same array under both names),
``{n}_control_{i}.h5`` ("control"), ``{n}_B_N_{i}.h5`` ("B_ECI") and ``{n}_t_total_{i}.h5`` ("t_total"), i counted from 1.
`write_monte_carlo` writes exactly that file set as HDF5 through the system's libhdf5 (hdf5io.py; same dataset names, and the
same bytes HDF5.jl stores for the script's column-major arrays: a Julia 7 x n_i array is an (n_i, 7) dataset). Where no
libhdf5 can be loaded — or with fmt="npz" — the same names and arrays go into ``.npz`` archives instead.
Julia shapes: A n x 6, states 7 x n_i, control 3 x (n_i - 1), B_ECI 2N x 3, t_total n_i.
"""
import os

import numpy as np

from . import hdf5io


def write_monte_carlo(out_dir, res, tag=None, fmt=None):
    """`res` is what monte_carlo.run_trials returns (with keep_trajectories). Returns the list of files written."""
    os.makedirs(out_dir, exist_ok=True)
    n = tag if tag is not None else len(res["A"])
    fmt = fmt or ("h5" if hdf5io.available() else "npz")
    files = []

    def put(name, **arrays):
        path = os.path.join(out_dir, f"{name}.{fmt}")
        if fmt == "h5":
            if os.path.exists(path):
                os.remove(path)
            for k, a in arrays.items():
                # trial arrays are held Julia-shaped (rows = components); their transposes are the C-order datasets HDF5.jl writes
                hdf5io.h5write(path, k, np.ascontiguousarray(np.asarray(a, dtype=np.float64).T))
        else:
            np.savez(path, **arrays)
        files.append(path)

    put(f"{n}_A", A=np.asarray(res["A"]))
    for j, i in enumerate(res.get("selected", [])):
        put(f"{n}_states_{i + 1}", one_state=res["sim_states"][j], states=res["sim_states"][j])
        put(f"{n}_control_{i + 1}", control=res["sim_control_inputs"][j])
        put(f"{n}_B_N_{i + 1}", B_ECI=res["B_ECI_total"][j])
        put(f"{n}_t_total_{i + 1}", t_total=res["t_total"][j])
    return files


def read_monte_carlo(out_dir, tag):
    """Inverse of write_monte_carlo: dict(A, states{i}, control{i}, B_ECI{i}, t_total{i}) keyed by 1-based trial, arrays in
    the Julia shapes."""
    def load(path, key):
        if path.endswith(".h5"):
            return np.ascontiguousarray(hdf5io.h5read(path, key).T)
        return np.load(path)[key]

    ext = ".h5" if os.path.exists(os.path.join(out_dir, f"{tag}_A.h5")) else ".npz"
    out = dict(A=load(os.path.join(out_dir, f"{tag}_A{ext}"), "A"), states={}, control={}, B_ECI={}, t_total={})
    for f in sorted(os.listdir(out_dir)):
        for stem, key in (("states", "states"), ("control", "control"), ("B_N", "B_ECI"), ("t_total", "t_total")):
            pre = f"{tag}_{stem}_"
            if f.startswith(pre) and f.endswith(ext):
                out[key][int(f[len(pre):-len(ext)])] = load(os.path.join(out_dir, f), key)
    return out
