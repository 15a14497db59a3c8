"""Register / scratch / LDS budget of every build of the solve kernel, from the compiler itself (no GPU needed).

For each translation unit of libtortoise_hip.so: `hipcc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage` gives, per
KERNEL, VGPRs / AGPRs / spilled registers / scratch bytes / LDS / occupancy (the remarks), and the ISA listing gives, per
FUNCTION (the phase functions are not inlined: each hot loop has its own register allocation), the registers it uses, its
scratch size and how many scratch_* instructions it contains — i.e. whether the spills sit inside the hot loops or in the
driver around them.

    python tools/resource_usage.py [profiles/r03/resource_usage.txt]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tortoisesat.jl_amd", "csrc")
UNITS = ["tsat_kernels.hip", "tsat_kernels_dense.hip", "tsat_kernels_packed.hip", "tsat_kernels_packed8.hip", "tsat_kernels_packed4w.hip", "tsat_kernels_packed8w.hip", "tsat_kernels_packed16w.hip", "tsat_kernels_dense_mixed.hip",
         "tsat_kernels_packed_mixed.hip", "tsat_kernels_packed8_mixed.hip", "tsat_kernels_packed8w_mixed.hip", "tsat_kernels_packed16w_mixed.hip"]
# the variants the BASELINE configs run (rk3, isotropic inertia) first, then the worst of the others
BENCH = ("3, 2, 1>", "3, 2, 0>")


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def compile_unit(unit, tmp):
    s = os.path.join(tmp, unit + ".s")
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", unit, "-o", s], cwd=CSRC, capture_output=True, text=True)
    if p.returncode:
        raise RuntimeError(p.stderr[-2000:])
    return s, p.stderr


def kernel_remarks(err):
    out, cur = collections.OrderedDict(), None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(?:Function Name: (\S+)|\s*([\w \[\]/]+?):\s+(\S+))\s+\[-Rpass", line)
        if not m:
            continue
        if m.group(1):
            cur = out.setdefault(m.group(1), {})
        elif cur is not None:
            cur[m.group(2).strip()] = m.group(3)
    return out


def function_info(path):
    """per function: info block fields + scratch / instruction counts"""
    out, cur, ins = collections.OrderedDict(), None, None
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith((".L", "BB")):
            cur = m.group(1)
            out[cur] = dict(n=0, scratch_insts=0)
            continue
        if cur is None:
            continue
        m = re.match(r"^; (codeLenInByte|NumVgprs|NumAgprs|ScratchSize|TotalNumSgprs)[ =:]+(\d+)", line)
        if m:
            out[cur][m.group(1)] = int(m.group(2))
            continue
        t = line.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        out[cur]["n"] += 1
        if op.startswith("scratch_"):
            out[cur]["scratch_insts"] += 1
    return out


def own_name(full):
    """`tsat::name<args>` of a demangled signature (return type, parameter list and nested tsat:: qualifiers in the arguments
    set aside), or None for kernels and foreign functions"""
    sig = full.split("(")[0]
    depth, start = 0, None
    for i, ch in enumerate(sig):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif depth == 0 and sig.startswith("tsat::", i):
            start = i + 6
    return sig[start:].strip() if start is not None else None


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03", "resource_usage.txt")
    lines = ["Register / scratch / LDS budget of every build of the solve kernel (tools/resource_usage.py; hipcc "
             "-Rpass-analysis=kernel-resource-usage + the ISA listing's per-function info blocks; gfx950, -O3).",
             "Kernels: what the launch reserves. Functions: the non-inlined phase functions, each with its own register allocation;",
             "`scratch insts` = scratch_load / scratch_store instructions in the function's body (spills inside its loops).", ""]
    with tempfile.TemporaryDirectory() as tmp:
        for unit in UNITS:
            s, err = compile_unit(unit, tmp)
            rem, fi = kernel_remarks(err), function_info(s)
            dm = demangle(list(fi) + list(rem))
            lines.append(f"== {unit}")
            lines.append(f"  {'kernel':<62} {'VGPR':>5} {'AGPR':>5} {'spillV':>6} {'spillS':>6} {'scratch B/lane':>14} {'LDS B':>6} {'waves/SIMD':>10}")
            solve = [(k, v) for k, v in rem.items() if "solve_kernel" in dm.get(k, k) or "resume_kernel" in dm.get(k, k)]
            solve.sort(key=lambda kv: (not any(b in dm[kv[0]] for b in BENCH), dm[kv[0]]))
            for k, v in solve:
                name = re.sub(r"\(tsat::KArgs<\w+>\)|void ", "", dm[k])
                lines.append(f"  {name:<62} {v.get('VGPRs', '?'):>5} {v.get('AGPRs', '?'):>5} {v.get('VGPRs Spill', '?'):>6} {v.get('SGPRs Spill', '?'):>6} "
                             f"{v.get('ScratchSize [bytes/lane]', '?'):>14} {v.get('LDS Size [bytes/block]', '?'):>6} {v.get('Occupancy [waves/SIMD]', '?'):>10}")
            lines.append(f"  {'function (bench variant rk3 / isotropic / hooks on, then the largest scratch user of each name)':<100} {'VGPR':>5} {'AGPR':>5} {'scratch B':>9} {'scratch insts':>13} {'insts':>6}")
            by_name = collections.OrderedDict()
            for f, d in fi.items():
                full = dm.get(f, f)
                own = own_name(full)
                if own is None or "NumVgprs" not in d:
                    continue
                base = re.sub(r"<.*", "", own)
                by_name.setdefault(base, []).append((own, d))
            for base, lst in by_name.items():
                bench = [x for x in lst if any(b in x[0] for b in BENCH) or re.fullmatch(r"\w+<(double|float)(, 6(, tsat::\w+(<1>)?)?)?\s?>", x[0])]
                worst = max(lst, key=lambda x: x[1]["scratch_insts"])
                shown = []
                for full, d in (bench[:3] + [worst]):
                    if full in shown:
                        continue
                    shown.append(full)
                    nm = full[:100]
                    lines.append(f"  {nm:<100} {d['NumVgprs']:>5} {d.get('NumAgprs', 0):>5} {d.get('ScratchSize', 0):>9} {d['scratch_insts']:>13} {d['n']:>6}")
            lines.append("")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
