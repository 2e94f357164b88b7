#!/bin/bash
# Regenerates the per-round measurement artifacts on the GPU box (run through gpurun from the repo root):
#   tools/gpu_round_artifacts.sh r01
# -> gpurun_out/<tag>_bench_n1.json, <tag>_bench_n1_kernel_stats.csv, <tag>_pmc_traffic.json, ...
set -u
TAG=${1:-r00}
WHAT=${2:-all}   # all | pmc
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
if [ "$WHAT" = all ]; then
python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
tail -c 600 $OUT/${TAG}_bench_n1.json
cd /tmp && export TMPDIR=/tmp
# same command under the kernel trace (the CPU-baseline leg launches no kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $REPO/bench.py --no-cpu-baseline \
   > $OUT/${TAG}_bench_n1_under_rocprof.json 2> /tmp/prof_kt.err
f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_bench_n1_kernel_stats.csv
fi
cd /tmp && export TMPDIR=/tmp
# HBM traffic: one PMC pass per counter (never combined with other trace domains)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -- python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 \
     --pmc-calibrate > /tmp/prof_$c.json 2> /tmp/prof_$c.err
done
ff=$(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' | head -1)
fw=$(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python $REPO/tools/pmc_summary.py $ff $fw q200x50_db1000000_d1024_k64_w1 > $OUT/${TAG}_pmc_traffic.json 2> $OUT/${TAG}_pmc.err
head -c 1500 $OUT/${TAG}_pmc_traffic.json; cat $OUT/${TAG}_pmc.err | tail -3
head -3 $ff | cut -c1-600; grep -c . $ff
