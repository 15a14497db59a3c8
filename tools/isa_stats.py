"""Developer tool: per-function instruction mix of a gfx950 ISA listing (hipcc -S --cuda-device-only).
   python tools/isa_stats.py file.s [substring-of-demangled-or-mangled-name ...]"""
import re, sys, subprocess, collections

def functions(path):
    cur, out = None, collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith((".L", "BB")):
            cur = m.group(1); out[cur] = []; continue
        if cur is None: continue
        if line.startswith("\t.") or line.startswith(".") : 
            if ".end_amdhsa_kernel" in line or line.startswith("\t.size") or line.startswith(".Lfunc_end"): 
                pass
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.endswith(":"): continue
        out[cur].append(t.split()[0])
    return out

def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))

if __name__ == "__main__":
    fns = functions(sys.argv[1]); dm = demangle(list(fns))
    pats = sys.argv[2:]
    print(f"{'function':<70} {'total':>6} {'valu':>6} {'fma64':>6} {'mul64':>6} {'add64':>6} {'f32':>6} {'ds_rd':>6} {'ds_wr':>6} {'glob':>5} {'scratch':>7} {'salu':>6} {'waitcnt':>7}")
    for n, ins in fns.items():
        d = dm.get(n, n)
        if pats and not any(p in d or p in n for p in pats): continue
        if len(ins) < 50: continue
        c = collections.Counter(ins)
        g = lambda f: sum(v for k, v in c.items() if f(k))
        short = re.sub(r"tsat::", "", d)[:70]
        print(f"{short:<70} {len(ins):>6} {g(lambda k: k.startswith('v_')):>6} {g(lambda k: k.startswith(('v_fma_f64','v_fmac_f64'))):>6} {g(lambda k: k.startswith('v_mul_f64')):>6} "
              f"{g(lambda k: k.startswith('v_add_f64')):>6} {g(lambda k: k.endswith('_f32') and k.startswith('v_')):>6} {g(lambda k: k.startswith('ds_read') or k.startswith('ds_load')):>6} {g(lambda k: k.startswith('ds_write') or k.startswith('ds_store')):>6} "
              f"{g(lambda k: k.startswith(('global_', 'flat_', 'buffer_'))):>5} {g(lambda k: k.startswith('scratch_')):>7} {g(lambda k: k.startswith('s_') and not k.startswith('s_waitcnt')):>6} {g(lambda k: k.startswith('s_waitcnt')):>7}")
