"""Developer tool: fill the @NAME@ placeholders of a text (DESIGN.md section drafts, profiles/rNN/README.md) with the numbers of the
committed measurement files of a round, so that a document quotes what the files hold.

    python tools/fill_numbers.py profiles/r04 < draft.md > filled.md
"""
import json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
rdir = sys.argv[1]


def line(name):
    with open(os.path.join(rdir, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def pmc(name):
    with open(os.path.join(root, "profiles", name)) as f:
        return json.load(f)


def stats_row(csv_name, needle):
    import csv
    with open(os.path.join(rdir, csv_name)) as f:
        for r in csv.DictReader(f):
            if needle in r["Name"]:
                return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    return float("nan"), 0


def phase(fname):
    out = {}
    with open(os.path.join(rdir, fname)) as f:
        for ln in f:
            m = re.match(r"\s+(forward sweep|jacobian lanes|riccati|parallel passes)\s*:.*per knot-iteration\s+([0-9.]+) cycles", ln)
            if m:
                out[m.group(1)] = float(m.group(2))
    return out


from bench import kernel_source_hash

h = line("bench_n1_final.json")
oc = {o["bench_config"]: o for o in h.get("other_configs", [])}
c2, c2m, c3, c4 = line("bench_c2_fp64.json"), line("bench_c2_mixed.json"), line("bench_c3.json"), line("bench_c4.json")
p1, p2, p2m, p3 = pmc("pmc_summary.json"), pmc("pmc_summary_c2.json"), pmc("pmc_summary_c2mixed.json"), pmc("pmc_summary_c3.json")
ur = line("bench_under_rocprof.json")
ks, _ = stats_row("kernel_stats_bench_steps5_final.csv", "tsat_solve_kernel<double, 3, 2, 1>")
ks2a, _ = stats_row("kernel_stats_c2_mixed.csv", "tsat_solve_kernel_packed_mixed16w")
ks2b, _ = stats_row("kernel_stats_c2_mixed.csv", "tsat_resume_kernel_packed_mixed<")
ph = phase("phase_clocks_final.txt")
phw = phase("phase_clocks_packed8w.txt")
ph8 = phase("phase_clocks_packed8.txt")
lower = lambda p: p["hbm_bytes_per_launch_lower"] / p["algorithmic_bytes_per_launch"]
shard_ms = None
with open(os.path.join(rdir, "straggler_stats.txt")) as f:
    for ln in f:
        m = re.match(r"variant 0: ([0-9.]+) ms -> ([0-9]+) solves/s", ln)
        if m:
            shard_ms, shard_sps = float(m.group(1)), float(m.group(2))
V = {
    "STAMP": kernel_source_hash(),
    "H_MS": f"{h['ms_per_step']:.1f}", "H_SPS": f"{h['value'] / 1e3:.1f}", "H_FRAC": f"{h['roofline']['frac']:.3f}",
    "H_TR": f"{p1['traffic_over_algorithmic']:.2f}", "H_TRL": f"{lower(p1):.2f}",
    "H0_MS": f"{h['other_mode']['kernel_ms']:.1f}", "H0_SPS": f"{h['other_mode']['solves_per_s'] / 1e3:.1f}",
    "CPU": f"{h['cpu_baseline']['value']:.0f}",
    "O2_MS": f"{oc[2]['ms_per_step']:.0f}", "O2_SPS": f"{oc[2]['solves_per_s'] / 1e3:.1f}",
    "O3_MS": f"{oc[3]['ms_per_step']:.0f}", "O3_SPS": f"{oc[3]['solves_per_s'] / 1e3:.1f}",
    "O4_MS": f"{oc[4]['ms_per_control_step']:.2f}", "O4_SPS": f"{oc[4]['solves_per_s'] / 1e6:.2f}",
    "C2_MS": f"{c2['ms_per_step']:.0f}", "C2_SPS": f"{c2['value'] / 1e3:.1f}", "C2_FRAC": f"{c2['roofline']['frac']:.3f}",
    "C2_TR": f"{p2['traffic_over_algorithmic']:.2f}", "C2_TRL": f"{lower(p2):.2f}",
    "C2M_MS": f"{c2m['ms_per_step']:.0f}", "C2M_SPS": f"{c2m['value'] / 1e3:.1f}", "C2M_FRAC": f"{c2m['roofline']['frac']:.3f}",
    "C2M_TR": f"{p2m['traffic_over_algorithmic']:.2f}", "C2M_TRL": f"{lower(p2m):.2f}",
    "C3_MS": f"{c3['ms_per_step'] / 1e3:.2f}", "C3_SPS": f"{c3['value'] / 1e3:.1f}", "C3_FRAC": f"{c3['roofline']['frac']:.3f}",
    "C3_TR": f"{p3['traffic_over_algorithmic']:.2f}",
    "C3S_MS": f"{shard_ms:.0f}", "C3S_SPS": f"{shard_sps / 1e3:.1f}",
    "C4_MS": f"{c4['ms_per_step'] / 100:.2f}", "C4_SPS": f"{c4['value'] / 1e6:.2f}", "C4_FRAC": f"{c4['roofline']['frac']:.3f}",
    "KS_MS": f"{ks:.2f}", "KS_HIP": f"{ur['roofline']['kernel_ms']:.2f}", "KS_C2M": f"{ks2a:.1f} + {ks2b:.1f}",
    "W_FWD": f"{phw['forward sweep']:.0f}", "W_JAC": f"{phw['jacobian lanes']:.0f}", "W_RIC": f"{phw['riccati']:.0f}", "W_PAR": f"{phw['parallel passes']:.0f}",
    "P8H_FWD": f"{ph8['forward sweep'] / 2:.0f}", "P8H_JAC": f"{ph8['jacobian lanes'] / 2:.0f}", "P8H_RIC": f"{ph8['riccati'] / 2:.0f}", "P8H_PAR": f"{ph8['parallel passes'] / 2:.0f}",
    "P_FWD": f"{ph['forward sweep']:.0f}", "P_JAC": f"{ph['jacobian lanes']:.0f}", "P_RIC": f"{ph['riccati']:.0f}", "P_PAR": f"{ph['parallel passes']:.0f}",
}
text = sys.stdin.read()
missing = set(re.findall(r"@([A-Z0-9_]+)@", text)) - set(V)
if missing:
    raise SystemExit(f"no value for {sorted(missing)}")
sys.stdout.write(re.sub(r"@([A-Z0-9_]+)@", lambda m: V[m.group(1)], text))
