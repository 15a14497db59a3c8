"""Result files of the Monte-Carlo (src/monte_carlo.jl:334-343).

The script writes, per experiment of `n` trials, ``{n}_A.h5`` (dataset "A") and per trial ``{n}_states_{i}.h5``
("states" = the closed-loop states), ``{n}_control_{i}.h5`` ("control"), ``{n}_B_N_{i}.h5`` ("B_ECI") and
``{n}_t_total_{i}.h5`` ("t_total"), i counted from 1. No HDF5 library exists in this image, so the same file set is
written as ``.npz`` archives holding the same dataset names and the same array shapes (states 7 x n_i, control
3 x (n_i - 1), B_ECI 2N x 3, t_total n_i); julia/TortoiseHIP.jl writes the .h5 files where HDF5.jl exists.
"""
import os

import numpy as np


def write_monte_carlo(out_dir, res, tag=None):
    """`res` is what monte_carlo.run_trials returns (with keep_trajectories). Returns the list of files written."""
    os.makedirs(out_dir, exist_ok=True)
    n = tag if tag is not None else len(res["A"])
    files = []

    def put(name, **arrays):
        path = os.path.join(out_dir, name + ".npz")
        np.savez(path, **arrays)
        files.append(path)

    put(f"{n}_A", A=np.asarray(res["A"]))
    for j, i in enumerate(res.get("selected", [])):
        put(f"{n}_states_{i + 1}", states=res["sim_states"][j])
        put(f"{n}_control_{i + 1}", control=res["sim_control_inputs"][j])
        put(f"{n}_B_N_{i + 1}", B_ECI=res["B_ECI_total"][j])
        put(f"{n}_t_total_{i + 1}", t_total=res["t_total"][j])
    return files


def read_monte_carlo(out_dir, tag):
    """Inverse of write_monte_carlo: dict(A, states{i}, control{i}, B_ECI{i}, t_total{i}) keyed by 1-based trial."""
    out = dict(A=np.load(os.path.join(out_dir, f"{tag}_A.npz"))["A"], states={}, control={}, B_ECI={}, t_total={})
    for f in sorted(os.listdir(out_dir)):
        for stem, key in (("states", "states"), ("control", "control"), ("B_N", "B_ECI"), ("t_total", "t_total")):
            pre = f"{tag}_{stem}_"
            if f.startswith(pre) and f.endswith(".npz"):
                out[key][int(f[len(pre):-4])] = np.load(os.path.join(out_dir, f))[key]
    return out
