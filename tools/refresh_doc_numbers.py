"""Developer tool: after a new measurement set has been summarised into profiles/rNN, bring the numbers the documents quote in line
with it. DESIGN.md's results block is regenerated from tools/doc_templates/design_s5_results.md; every other quoted number is found
by its context (the words around it) with the value the PREVIOUS measurement files gave — a git worktree of HEAD — and replaced.

    python tools/refresh_doc_numbers.py profiles/r04        # run BEFORE committing the new measurement files (the PMC summaries included:
                                                            # a summary committed earlier leaves its figures to be patched by hand)
"""
import json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rdir = sys.argv[1]


def values(root):
    src = open(os.path.join(ROOT, "tools", "fill_numbers.py")).read()
    src = src.replace('root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', f'root = "{root}"')
    src = src.replace('text = sys.stdin.read()', 'import json as _j; print(_j.dumps(V)); raise SystemExit(0)')
    out = subprocess.run([sys.executable, "-c", src, os.path.join(root, rdir)], capture_output=True, text=True, cwd=root)
    return json.loads(out.stdout.strip().splitlines()[-1])


with tempfile.TemporaryDirectory() as tmp:
    old_root = os.path.join(tmp, "head")
    subprocess.check_call(["git", "-C", ROOT, "worktree", "add", "-q", "--detach", old_root, "HEAD"])
    try:
        old = values(old_root)
    finally:
        subprocess.check_call(["git", "-C", ROOT, "worktree", "remove", "--force", old_root])
new = values(ROOT)
print({k: (old[k], new[k]) for k in new if old.get(k) != new[k]})


def ctx(s, before, key, after="", where=""):
    o = before + old[key] + after
    assert o in s, (where, key, o)
    return s.replace(o, before + new[key] + after)


def fill(text):
    import re
    return re.sub(r"@([A-Z0-9_]+)@", lambda m: new[m.group(1)], text)


p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
a = s.index("* Round-4 results, one GPU (`profiles/r04/`; round 3 in brackets):")
b = s.index("* **What the sequential chains cost, instruction by instruction**")
s = s[:a] + fill(open(os.path.join(ROOT, "tools", "doc_templates", "design_s5_results.md")).read()) + s[b:]
for before, key, after in (("iteration (`phase_clocks_packed8w.txt`) forward sweep ", "W_FWD", ", Jacobian lanes "), (", Jacobian lanes ", "W_JAC", ", Riccati lanes "),
                           (", Riccati lanes ", "W_RIC", ", passes "), (", passes ", "W_PAR", " cycles — each"), ("the shard of `configs[3]` is ", "C3S_MS", " ms in\n   packed8w"),
                           ("fp64 **", "C2_MS", " ms = "), (" ms = ", "C2_SPS", " k solves/s** (31.9 k), `roofline.frac` "),
                           (" k solves/s** (31.9 k), `roofline.frac` ", "C2_FRAC", ", **traffic"), ("precision 32 (the mixed build, which is held to the oracle): **", "C2M_MS", " ms = "),
                           (" ms = ", "C2M_SPS", " k**, frac "), (" k**, frac ", "C2M_FRAC", ", traffic "),
                           ("**not reached for the one-trajectory mapping: ", "H_MS", " ms, "), ("the shard is at ", "C3S_SPS", " k ≥ 12.5 k")):
    s = ctx(s, before, key, after, "DESIGN.md")
open(p, "w").write(s)
p = os.path.join(ROOT, "README.md"); s = open(p).read()
for before, key, after in (("| ", "H_MS", " ms per batch = **"), (" ms per batch = **", "H_SPS", " k solves/s** (`profiles/r04/bench_n1_final.json`"), ("hooks off ", "H0_MS", " ms; CPU oracle"),
                           ("threads of the same box: ", "CPU", " solves/s. The kernel"), ("16384 × 1000 knots: **", "C2_SPS", " k solves/s** fp64"),
                           ("3.41×), **", "C2M_SPS", " k** with `precision = 32`"), ("30 … 150): **", "C3S_SPS", " k solves/s** per GPU"),
                           ("sweep on one GPU ", "C3_SPS", " k (`profiles/r04/`)"), ("| ", "C4_MS", " ms per step = "), (" ms per step = ", "C4_SPS", " M re-solves/s (CPU oracle"),
                           ("one of its eight shards on one GPU (", "C3S_MS", " ms for shard 3")):
    s = ctx(s, before, key, after, "README.md")
open(p, "w").write(s)
# profiles/rNN/README.md: every value sits in a row of its own kind; replace value by value inside the rows that quote measurement files
p = os.path.join(ROOT, rdir, "README.md"); s = open(p).read()
rows = s.split("\n")
keys_by_row = {"| `bench_n1_final.json` |": ("H_MS", "H_SPS", "H0_MS", "CPU", "O2_MS", "O2_SPS", "O3_MS", "O3_SPS", "O4_MS", "O4_SPS"),
               "| `kernel_stats_bench_steps5_final.csv`": ("KS_MS", "KS_HIP"), "| `../pmc_summary.json`": ("H_TR", "H_TRL"),
               "| `bench_c2_fp64.json`": ("C2_MS", "C2_SPS", "C2_FRAC", "C2_TR", "C2_TRL", "C2M_MS", "C2M_SPS", "C2M_FRAC", "C2M_TR", "C2M_TRL"),
               "| `bench_c3.json`": ("C3_MS", "C3_SPS", "C3_TR"), "| `straggler_stats.txt`": ("C3S_MS", "C3S_SPS"), "| `bench_c4.json`": ("C4_SPS", "C4_MS"),
               "| `phase_clocks_final.txt`": ("P_FWD", "P_JAC", "P_RIC", "P_PAR", "W_FWD", "W_JAC", "W_RIC", "W_PAR")}
for i, r in enumerate(rows):
    for k, keys in keys_by_row.items():
        if r.startswith(k):
            for key in keys:
                if old[key] != new[key]:
                    assert r.count(old[key]) >= 1, (k, key, old[key])
                    r = r.replace(old[key], new[key], 1)
            rows[i] = r
s = "\n".join(rows).replace(old["STAMP"], new["STAMP"])
open(p, "w").write(s)
print("documents refreshed")
