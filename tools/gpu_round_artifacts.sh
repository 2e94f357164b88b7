#!/bin/bash
# Regenerates the per-round measurement artifacts on the GPU box (run through gpurun from the repo root):
#   tools/gpu_round_artifacts.sh r01 [all|bench|pmc]
# -> gpurun_out/<tag>_bench_n1.json, <tag>_bench_n1_kernel_stats.csv, <tag>_pmc_traffic.json, ...
# EVERY step runs under its own `timeout`: in round 1 a counter-collection pass without one stalled and burnt the
# remaining 33 GPU-minutes of the round (the kernel-trace and bench steps before it had finished in < 2 min).
set -u
TAG=${1:-r00}
WHAT=${2:-all}   # all | bench | pmc
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  timeout 300 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
  tail -c 600 $OUT/${TAG}_bench_n1.json
  cd /tmp && export TMPDIR=/tmp
  # same command under the kernel trace (the CPU-baseline leg launches no kernels)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $REPO/bench.py --no-cpu-baseline \
     > $OUT/${TAG}_bench_n1_under_rocprof.json 2> /tmp/prof_kt.err
  f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_bench_n1_kernel_stats.csv
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  cd /tmp && export TMPDIR=/tmp
  # HBM traffic of the dominant kernel: one PMC pass per counter (never combined with other trace domains) over the
  # SEARCH of the bench workload only (tools/probe_knn.py: 10 000 query segments x 1 M rows x 1024, a few dozen
  # dispatches) -- counter collection over the whole bench (600 k torch dispatches of the synthetic-image factory)
  # is what stalled in round 1
  for c in FETCH_SIZE WRITE_SIZE; do
    NR=1000000 NQ=10000 REPS=2 PMC_CAL=1 timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -- \
       python $REPO/tools/probe_knn.py > /tmp/prof_$c.log 2> /tmp/prof_$c.err || echo "PMC pass $c: timeout or failure"
  done
  ff=$(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' 2>/dev/null | head -1)
  fw=$(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' 2>/dev/null | head -1)
  if [ -n "$ff" ] && [ -n "$fw" ]; then
    python $REPO/tools/pmc_summary.py $ff $fw q200x50_db1000000_d1024_k64_w1 > $OUT/${TAG}_pmc_traffic.json 2> $OUT/${TAG}_pmc.err
    head -c 1500 $OUT/${TAG}_pmc_traffic.json
  else
    echo "no counter_collection.csv: PMC summary skipped" | tee $OUT/${TAG}_pmc.err
  fi
fi
