#!/bin/bash
# Developer tool, run on the GPU box from the repo root:  bash tools/collect_profiles.sh gpurun_out/final
# Final-round measurement set: bench line, rocprofv3 kernel statistics of the same command, three separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ counters; --kernel-trace only, as MI355X_MICROARCH.md prescribes), phase clocks of the
# stamped build, the whole pipeline and the Monte-Carlo experiment. Summarise with tools/summarize_profiles.py.
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/final}")
ROOT=$(pwd)
mkdir -p "$OUT"
PART="${PART:-AB}"     # A: configs[1], pipeline, experiment, batch sizes; B: configs[2] (fp64, mixed), [3], [4] — one gpurun call each fits 20 minutes
                       # S: the configs[3] shard entry of the default line on its own (bench entry, kernel statistics, PMC passes)
                       # L: the bench lines alone, once more (after the PMC summaries of A / P / B have been committed: the lines then carry `roofline.traffic`)
if [[ "$PART" == *A* || "$PART" == *L* ]]; then
python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
fi
if [[ "$PART" == *L* ]]; then
python3 bench.py --config 2 --precision 64 --steps 3 --warmup 1 --cpu-seconds 8 > "$OUT/bench_c2_fp64.json" 2> "$OUT/bench_c2_fp64.err"
python3 bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 8 > "$OUT/bench_c2_mixed.json" 2> "$OUT/bench_c2_mixed.err"
python3 bench.py --config 4 --steps 2 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config 3 --steps 2 --warmup 1 --cpu-seconds 8 --gather none > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
fi
if [[ "$PART" == *A* || "$PART" == *P* ]]; then      # P: the rocprofv3 passes of configs[1] alone
cd /tmp && export TMPDIR=/tmp
# (the profiled runs are the headline alone: --no-other-configs keeps the packed / mixed / receding-horizon launches of the default line out of them)
B="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $B > "$OUT/stats.json" 2> "$OUT/stats.err"
B2="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --gather none"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- $B2 > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- $B2 > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -o run -- $B2 > /dev/null 2> "$OUT/pmc_sq.err"
cd "$ROOT"
fi
if [[ "$PART" == *A* ]]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pipeline" -o run -- python3 $ROOT/tools/pipeline_timing.py > "$OUT/pipeline.log" 2> "$OUT/pipeline.err"
cd "$ROOT"
python3 tools/phase_profile.py 1024 1000 1 1 > "$OUT/phase_clocks.txt" 2> "$OUT/phase_clocks.err"
python3 tools/monte_carlo_timing.py 1024 1024 > "$OUT/monte_carlo.txt" 2> "$OUT/monte_carlo.err"
TSAT_VARIANTS=2,3,4,5,6 python3 tools/large_batch.py > "$OUT/large_batch.txt" 2> "$OUT/large_batch.err"
python3 tools/threshold_timing.py 1024 2048 3072 4096 6144 8192 12288 16384 > "$OUT/build_by_batch_size.txt" 2> "$OUT/build_by_batch_size.err"
python3 tools/fp32_eval.py 16384 512 > "$OUT/fp32_eval.txt" 2> "$OUT/fp32_eval.err"
{ python3 tools/mpc_timing.py 512 500; python3 tools/mpc_timing.py 4096 200; } > "$OUT/mpc_timing.txt" 2> "$OUT/mpc_timing.err"
fi
if [[ "$PART" == *B* ]]; then
# larger configurations on this one GPU: configs[2] shape in fp64 (packed build) and as quoted (fp32), a configs[3] shard of
# 8192 trajectories per GPU (what each of 8 GPUs gets), configs[4]; kernel statistics and PMC passes of the fp64 16384 run
python3 bench.py --config 2 --precision 64 --steps 3 --warmup 1 --cpu-seconds 8 > "$OUT/bench_c2_fp64.json" 2> "$OUT/bench_c2_fp64.err"
python3 bench.py --config 2 --steps 3 --warmup 1 --cpu-seconds 8 > "$OUT/bench_c2_mixed.json" 2> "$OUT/bench_c2_mixed.err"
python3 bench.py --config 4 --steps 2 --warmup 1 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
cd /tmp
C2="python3 $ROOT/bench.py --config 2 --precision 64 --steps 2 --warmup 1 --no-cpu-baseline --gather none"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o run -- $C2 > /dev/null 2> "$OUT/stats_c2.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c2" -o run -- $C2 > /dev/null 2> "$OUT/pmc_fetch_c2.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c2" -o run -- $C2 > /dev/null 2> "$OUT/pmc_write_c2.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq_c2" -o run -- $C2 > /dev/null 2> "$OUT/pmc_sq_c2.err"
# the same four passes for configs[2] as BASELINE.json quotes it (precision = 32: the mixed build) and for configs[3] (the whole 65536-trajectory sweep on this one GPU)
C2F="python3 $ROOT/bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --gather none"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2mixed" -o run -- $C2F > /dev/null 2> "$OUT/stats_c2mixed.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c2mixed" -o run -- $C2F > /dev/null 2> "$OUT/pmc_fetch_c2mixed.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c2mixed" -o run -- $C2F > /dev/null 2> "$OUT/pmc_write_c2mixed.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq_c2mixed" -o run -- $C2F > /dev/null 2> "$OUT/pmc_sq_c2mixed.err"
echo "configs[2] mixed passes done" > "$OUT/progress.txt"
cd "$ROOT"
python3 bench.py --config 3 --steps 2 --warmup 1 --cpu-seconds 8 --gather none > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
cd /tmp
C3="python3 $ROOT/bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline --gather none"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3" -o run -- $C3 > /dev/null 2> "$OUT/stats_c3.err"
echo "configs[3] stats pass done" >> "$OUT/progress.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c3" -o run -- $C3 > /dev/null 2> "$OUT/pmc_fetch_c3.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c3" -o run -- $C3 > /dev/null 2> "$OUT/pmc_write_c3.err"
echo "configs[3] fetch / write passes done" >> "$OUT/progress.txt"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq_c3" -o run -- $C3 > /dev/null 2> "$OUT/pmc_sq_c3.err"
echo "configs[3] passes done" >> "$OUT/progress.txt"
cd "$ROOT"
python3 tools/straggler_stats.py 8192 > "$OUT/straggler_stats.txt" 2> "$OUT/straggler_stats.err"
python3 tools/endgame_sweep.py > "$OUT/endgame_sweep.txt" 2> "$OUT/endgame_sweep.err"
python3 tools/straggler_timeline.py 8192 0 2048 > "$OUT/straggler_timeline.txt" 2> "$OUT/straggler_timeline.err"
TSAT_PK_G=4 python3 tools/phase_profile.py 16384 1000 3 1 > "$OUT/phase_clocks_packed.txt" 2> "$OUT/phase_clocks_packed.err"
TSAT_PK_G=8 python3 tools/phase_profile.py 16384 1000 4 1 > "$OUT/phase_clocks_packed8.txt" 2> "$OUT/phase_clocks_packed8.err"
TSAT_PK_G=8 python3 tools/phase_profile.py 8192 1000 5 1 > "$OUT/phase_clocks_packed8w.txt" 2> "$OUT/phase_clocks_packed8w.err"
TSAT_PK_G=16 python3 tools/phase_profile.py 16384 1000 6 1 > "$OUT/phase_clocks_packed16w.txt" 2> "$OUT/phase_clocks_packed16w.err"
TSAT_PK_G=16 python3 tools/phase_profile.py 16384 1000 6 1 32 > "$OUT/phase_clocks_packed16w_mixed.txt" 2> "$OUT/phase_clocks_packed16w_mixed.err"
python3 tools/phase_profile.py 16384 1000 2 1 > "$OUT/phase_clocks_dense.txt" 2> "$OUT/phase_clocks_dense.err"
fi
if [[ "$PART" == *S* ]]; then      # the configs[3] shard of the default line on its own: bench entry + kernel statistics + the three PMC passes
python3 tools/shard_line.py > "$OUT/bench_c3shard.json" 2> "$OUT/bench_c3shard.err"
cd /tmp && export TMPDIR=/tmp
SH="python3 $ROOT/tools/shard_line.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3shard" -o run -- $SH > /dev/null 2> "$OUT/stats_c3shard.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c3shard" -o run -- $SH > /dev/null 2> "$OUT/pmc_fetch_c3shard.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c3shard" -o run -- $SH > /dev/null 2> "$OUT/pmc_write_c3shard.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq_c3shard" -o run -- $SH > /dev/null 2> "$OUT/pmc_sq_c3shard.err"
cd "$ROOT"
fi
tools/ubench/valu_f64 > "$OUT/valu_f64_ubench.txt" 2>&1 || true
# gpurun copies back at most 64 MiB: report and drop anything large (the summaries need the small CSVs and text files only)
find "$OUT" -type f -size +4M -exec ls -la {} \; -delete
du -sh "$OUT"
echo done
