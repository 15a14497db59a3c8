"""Static check of the DPP data hazards in the ISA hipcc generates for the row-oriented Riccati recursion (no GPU needed).

On gfx9 a DPP instruction that reads, through its DPP operand (src0), a VGPR written by a VALU instruction needs two wait states
between the two; a DPP instruction after a VALU write of EXEC needs five. The compiler's hazard recogniser inserts them for its own
instructions — it does not look inside inline asm, and the recursion's `v_fmac_f64_dpp` / `v_mov_b64_dpp` blocks
(tortoisesat.jl_amd/csrc/tsat_riccati_dpp.inc) are inline asm: each block opens with an `s_nop` that covers whatever the compiler may
have scheduled in front of it, and inside a block no DPP source is written. This script holds the generated code to that: it
compiles the translation units that contain the recursion to ISA and walks every function, counting wait states (one per
instruction, N + 1 for `s_nop N`) between a VALU write of a register and a DPP read of it.

    python tools/check_dpp_hazards.py            # exit code 1 and a listing if a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tortoisesat.jl_amd", "csrc")
# the packed unit holds both users of the blocks: riccati_group (four trajectories per wavefront) and, through the hand-over of a
# wavefront's last live trajectory, riccati_rows (one); `--all` adds the wide and the mixed-precision units
UNITS = ["tsat_kernels_packed.hip"]
ALL_UNITS = ["tsat_kernels.hip", "tsat_kernels_packed.hip", "tsat_kernels_packed_mixed.hip"]
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(operand):
    """set of VGPR numbers an operand names"""
    out = set()
    for m in REG.finditer(operand):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def operands(rest):
    """split the operand list of an instruction (up to the first DPP / modifier keyword)"""
    rest = re.split(r"\s(?:row_|quad_perm|bank_mask|bound_ctrl|wave_|offset|glc|slc|off\b)", " " + rest, maxsplit=1)[0]
    return [o.strip() for o in rest.split(",") if o.strip()]


def check_listing(path):
    """-> (number of DPP instructions seen, list of violations)"""
    n_dpp, bad = 0, []
    hist = []          # recent instructions, newest last: (wait_states, valu_dst_regs, writes_exec, text)
    func = None
    with open(path) as f:
        for ln in f:
            s = ln.split(";")[0].strip() if not ln.lstrip().startswith(";;#") else ""
            if not s:
                continue
            if s.endswith(":"):
                if not s.startswith(".L"):
                    func = s[:-1]
                hist = []          # a label: what runs before it is unknown here (every DPP block opens with its own s_nop)
                continue
            if s.startswith("."):
                continue
            parts = s.split(None, 1)
            mn, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            if mn == "s_nop":
                hist.append((int(rest.strip() or "0", 0) + 1, set(), False, s))
                hist = hist[-12:]
                continue
            ops = operands(rest)
            is_valu = mn.startswith("v_") and not mn.startswith("v_readlane") and not mn.startswith("v_readfirstlane")
            dst = regs(ops[0]) if (is_valu and ops and not mn.startswith("v_cmp")) else set()
            writes_exec = mn.startswith("v_cmpx") or (is_valu and ops and ops[0].strip() == "exec")
            if mn.endswith("_dpp"):
                n_dpp += 1
                src0 = regs(ops[1]) if len(ops) > 1 else set()
                ws = 0
                for w, d, ex, txt in reversed(hist):
                    if ws < 2 and d & src0:
                        bad.append(f"{func}: `{s}` reads {sorted(d & src0)} {ws} wait state(s) after `{txt}`")
                    if ws < 5 and ex:
                        bad.append(f"{func}: `{s}` {ws} wait state(s) after a VALU write of EXEC `{txt}`")
                    ws += w
                    if ws >= 5:
                        break
            hist.append((1, dst, writes_exec, s))
            hist = hist[-12:]
    return n_dpp, bad


def main(units=None):
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for u in (units or (ALL_UNITS if "--all" in sys.argv else UNITS)):
            out = os.path.join(tmp, u + ".s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(CSRC, u), "-o", out],
                                  stderr=subprocess.DEVNULL)
            n, b = check_listing(out)
            print(f"{u}: {n} DPP instructions, {len(b)} hazard(s)")
            total += n
            bad += b
    for b in bad[:40]:
        print("  " + b)
    if total == 0:
        print("no DPP instruction found: the check saw nothing")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
