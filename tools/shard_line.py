"""Developer tool (GPU): the configs[3] shard entry of the default bench line on its own — shard 3 of 8 of the inclination sweep,
8192 trajectories in the build a shard takes — so that rocprofv3 passes see these launches only (tools/collect_profiles.sh, PART S;
profiles/pmc_summary_c3shard.json). Prints the entry as one JSON line.   python tools/shard_line.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tsat_loader import load_package
load_package()
from tortoisesat_jl_amd import magnetic as mg, slew_setup as ss, trajopt as to

real_stdout = os.dup(1)
os.dup2(2, 1)
entry = bench.other_configs(None, torch.device("cuda", 0), ss, mg, to, torch, only=3)[0]
sys.stdout.flush()
os.dup2(real_stdout, 1)
print(json.dumps(entry), flush=True)
