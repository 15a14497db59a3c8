"""Developer tool: turn what tools/collect_profiles.sh wrote into the committed summaries under profiles/.

  python tools/summarize_profiles.py gpurun_out/final profiles/r01
writes <dst>/bench_n1_final.json, kernel_stats_*.csv, phase/pipeline/Monte-Carlo text files and profiles/pmc_summary.json
(per launch of the solve kernel; FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; the read figure is bracketed
by the raw value and its double, see MI355X_MICROARCH.md on wide streaming reads)."""
import csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_hash   # the stamp bench.py checks before it reports these counters
os.makedirs(dst, exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    if not g:
        raise SystemExit(f"missing {pattern}")
    return g[0]


RAW = os.path.join(dst, "raw")
os.makedirs(RAW, exist_ok=True)


def keep_raw(d):
    """the raw rocprofv3 counter file of a pass, next to its summary (small: one row per dispatch and counter)"""
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        if os.path.getsize(f) < 4 << 20:
            shutil.copy(f, os.path.join(RAW, f"{d}_counter_collection.csv"))


def counters(d):
    """mean per LAUNCH: a packed launch with an endgame is two kernels (solve + resume, tsat_set_endgame), dispatched equally often"""
    acc, n = {}, {}
    kern = None
    keep_raw(d)
    with open(one(f"{d}/**/*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            kind = "solve" if "tsat_solve_kernel" in name else ("resume" if "tsat_resume_kernel" in name else None)
            # the timed configuration only — rk3, isotropic inertia, quaternion hooks on — not the hooks-off launches of `other_mode`
            if kind is None or "3, 2, 1>" not in name:
                continue
            k = (kind, r["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
            if kind == "solve":
                kern = name
    out = {}
    for (kind, c), v in acc.items():
        out[c] = out.get(c, 0.0) + v / n[(kind, c)]
    if any(kind == "resume" for kind, _ in acc):
        kern += " + its resume kernel"
    return out, kern


shutil.copy(os.path.join(src, "bench_n1.json"), os.path.join(dst, "bench_n1_final.json"))
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, "kernel_stats_bench_steps5_final.csv"))
shutil.copy(one("pipeline/**/*kernel_stats.csv"), os.path.join(dst, "kernel_stats_pipeline.csv"))
for a, b in (("phase_clocks.txt", "phase_clocks_final.txt"), ("pipeline.log", "pipeline_wall.txt"),
             ("monte_carlo.txt", "monte_carlo_wall.txt"), ("large_batch.txt", "large_batch.txt"), ("mpc_timing.txt", "mpc_timing.txt"), ("stats.json", "bench_under_rocprof.json"),
             ("build_by_batch_size.txt", "build_by_batch_size.txt"), ("fp32_eval.txt", "fp32_eval.txt"), ("phase_clocks_packed.txt", "phase_clocks_packed.txt"),
             ("phase_clocks_packed8.txt", "phase_clocks_packed8.txt"), ("phase_clocks_dense.txt", "phase_clocks_dense.txt"),
             ("bench_c2_fp64.json", "bench_c2_fp64.json"), ("bench_c2_mixed.json", "bench_c2_mixed.json"), ("bench_c4.json", "bench_c4.json"),
             ("phase_clocks_packed8w.txt", "phase_clocks_packed8w.txt"), ("phase_clocks_packed16w.txt", "phase_clocks_packed16w.txt"),
             ("phase_clocks_packed16w_mixed.txt", "phase_clocks_packed16w_mixed.txt"), ("valu_f64_ubench.txt", "valu_f64_ubench.txt"), ("endgame_sweep.txt", "endgame_sweep.txt")):
    if not os.path.exists(os.path.join(src, a)):
        continue
    with open(os.path.join(src, a)) as f:
        keep = [ln for ln in f if not ln.startswith(("W20", "E20", "I20"))]
    with open(os.path.join(dst, b), "w") as f:
        f.writelines(keep)
bench = json.loads(open(os.path.join(src, "bench_n1.json")).read().strip().splitlines()[-1])
fe, kern = counters("pmc_fetch")
wr, _ = counters("pmc_write")
sq, _ = counters("pmc_sq")
rd_raw, wr_b = fe["FETCH_SIZE"] * 1024.0, wr["WRITE_SIZE"] * 1024.0
out = {
    "kernel": kern, "workload": bench["config"]["workload"], "bench_config": bench["config"].get("bench_config", 1), "dtype": bench["dtype"],
    "traffic_over_algorithmic": (2 * rd_raw + wr_b) / bench["roofline"]["algorithmic_bytes_per_launch"],
    "kernel_source_sha256_16": kernel_source_hash(),
    "FETCH_SIZE_KB": fe["FETCH_SIZE"], "WRITE_SIZE_KB": wr["WRITE_SIZE"],
    "read_bytes_raw": rd_raw, "read_bytes_x2": 2 * rd_raw, "write_bytes": wr_b,
    "hbm_bytes_per_launch": 2 * rd_raw + wr_b, "hbm_bytes_per_launch_lower": rd_raw + wr_b,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), mean per launch of the solve kernel; FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (upper bracket; 8-byte accesses are uncalibrated, lower bracket = raw).",
    "sq": sq,
}
with open(os.path.join(root, "profiles", "pmc_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))

# the same counters for the other configurations, when collected: configs[2] in fp64 and as quoted (fp32), configs[3]
def other(tag, bench_file, cfg, stats_csv):
    if not glob.glob(os.path.join(src, f"pmc_fetch_{tag}", "**", "*counter_collection.csv"), recursive=True):
        return
    shutil.copy(one(f"stats_{tag}/**/*kernel_stats.csv"), os.path.join(dst, stats_csv))
    b2 = json.loads(open(os.path.join(src, bench_file)).read().strip().splitlines()[-1])
    if "config" not in b2:      # an `other_configs` entry of the default line on its own (tools/shard_line.py)
        b2["config"] = {"workload": b2["workload"]}
        b2["roofline"]["kernel_ms"] = b2["ms_per_step"]
    fe, kern = counters(f"pmc_fetch_{tag}")
    wr, _ = counters(f"pmc_write_{tag}")
    sq, _ = counters(f"pmc_sq_{tag}")
    rd_raw, wr_b = fe["FETCH_SIZE"] * 1024.0, wr["WRITE_SIZE"] * 1024.0
    ms = b2["roofline"]["kernel_ms"]
    out2 = {
        "kernel": kern, "workload": b2["config"]["workload"], "bench_config": cfg, "dtype": b2["dtype"], "kernel_source_sha256_16": kernel_source_hash(),
        "kernel_ms": ms, "FETCH_SIZE_KB": fe["FETCH_SIZE"], "WRITE_SIZE_KB": wr["WRITE_SIZE"],
        "read_bytes_raw": rd_raw, "read_bytes_x2": 2 * rd_raw, "write_bytes": wr_b,
        "hbm_bytes_per_launch": 2 * rd_raw + wr_b, "hbm_bytes_per_launch_lower": rd_raw + wr_b,
        "algorithmic_bytes_per_launch": b2["roofline"]["algorithmic_bytes_per_launch"],
        "traffic_over_algorithmic": (2 * rd_raw + wr_b) / b2["roofline"]["algorithmic_bytes_per_launch"],
        "hbm_GBs_upper": (2 * rd_raw + wr_b) / (ms * 1e-3) / 1e9,
        "valu_issue_frac_of_4_cycle_peak": sq["SQ_INSTS_VALU"] / (1024 * 2.4e9 * ms * 1e-3 / 4.0),      # at the nominal 2.4 GHz; bench.py prices it at the clock the device reports
        "note": "as pmc_summary.json; valu_issue_frac: SQ_INSTS_VALU against one VALU instruction per 4 cycles on each of the 1024 SIMDs at 2.4 GHz",
        "sq": sq,
    }
    with open(os.path.join(root, "profiles", f"pmc_summary_{tag}.json"), "w") as f:
        json.dump(out2, f, indent=1)
    print(json.dumps(out2, indent=1))


other("c2", "bench_c2_fp64.json", 2, "kernel_stats_c2_fp64.csv")
other("c2mixed", "bench_c2_mixed.json", 2, "kernel_stats_c2_mixed.csv")
other("c3", "bench_c3.json", 3, "kernel_stats_c3.csv")
other("c3shard", "bench_c3shard.json", 3, "kernel_stats_c3shard.csv")
if os.path.exists(os.path.join(src, "bench_c3shard.json")):
    shutil.copy(os.path.join(src, "bench_c3shard.json"), os.path.join(dst, "bench_c3shard.json"))
for a, b in (("bench_c3.json", "bench_c3.json"), ("straggler_stats.txt", "straggler_stats.txt"), ("straggler_timeline.txt", "straggler_timeline.txt")):
    if os.path.exists(os.path.join(src, a)):
        with open(os.path.join(src, a)) as f:
            keep = [ln for ln in f if not ln.startswith(("W20", "E20", "I20")) and "amdgpu.ids" not in ln]
        with open(os.path.join(dst, b), "w") as f:
            f.writelines(keep)
