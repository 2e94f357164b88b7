#!/bin/bash
# Regenerates the per-round measurement artifacts on the GPU box (run through gpurun from the repo root):
#   tools/gpu_round_artifacts.sh r02 [all|bench|pmc]      (all = the counter passes first, then the bench that quotes them)
# -> gpurun_out/<tag>_bench_n1.json, <tag>_bench_n1_kernel_stats.csv, <tag>_pmc_traffic.json, ...
# EVERY step runs under its own `timeout`: in round 1 a counter-collection pass without one stalled and burnt the
# remaining 33 GPU-minutes of the round.
set -u
TAG=${1:-r00}
WHAT=${2:-all}   # all | bench | pmc
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  cd /tmp && export TMPDIR=/tmp
  # Round 6: the counter passes run over bench.py ITSELF -- the timed steps, with the options, the query-group
  # hint and the pca_path of the run -- not over a replay of its kernel shapes.  Counter collection is restricted to the library's
  # kernels + the marker (rocprofv3 --kernel-include-regex): the ~600 k torch dispatches of the synthetic-image factory, under which
  # collection stalled in round 1, are not instrumented.  `--pmc-calibrate` brackets the timed steps with two torch.sign dispatches
  # over 1 GiB: the byte calibration and the window tools/pmc_summary.py --window cuts out.  One pass per counter group, never
  # combined with other trace domains; every pass under its own timeout.
  KRE=$(grep -ho "void [A-Za-z0-9_]*kernel[A-Za-z0-9_]*" $REPO/revisit-anything_amd/csrc/*.hip | awk '{print $2}' | sort -u | paste -sd'|')
  KRE="$KRE|sign_kernel"
  STEPS=2
  BENCH_ARGS="--pmc-calibrate --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --steps $STEPS --warmup 1"
  export SEGVLAD_GUARD=0
  # (a) the same command under the kernel trace only: the step timeline the counter passes must reproduce launch for launch
  rm -rf /tmp/prof_TL
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_TL -- python $REPO/bench.py $BENCH_ARGS > /tmp/prof_TL.log 2> /tmp/prof_TL.err \
     || echo "timeline pass: timeout or failure"
  ftl=$(find /tmp/prof_TL -name '*kernel_trace.csv' 2>/dev/null | head -1)
  if [ -n "$ftl" ]; then
    python3 - "$ftl" > $OUT/${TAG}_step_timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sign_kernel" in r["Kernel_Name"]]
lo, hi = marks[0], marks[-1]
win = rows[lo + 1:hi]
votes = [i for i, r in enumerate(win) if "vote_kernel" in r["Kernel_Name"]]
print("# one timed step of bench.py (the LAST of the window between the two torch.sign markers): start, gap to the previous kernel's end, duration")
first = votes[-2] + 1 if len(votes) >= 2 else 0
t0 = int(win[first]["Start_Timestamp"]); prev = t0; gaps = 0.0
for r in win[first:votes[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - prev) / 1e3
    gaps += max(g, 0.0)
    print("%9.1f us  +%7.1f gap  %9.1f us  %s" % ((s - t0) / 1e3, g, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev = max(prev, e)
print("step %.1f us, gaps %.1f us; %d dispatches in the window of %d steps" % ((prev - t0) / 1e3, gaps, len(win), len(votes)))
PY
    tail -2 $OUT/${TAG}_step_timeline.txt
  fi
  # (b) the counter passes
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_$c
    timeout 500 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$KRE" --output-format csv -d /tmp/prof_$c -- \
       python $REPO/bench.py $BENCH_ARGS > /tmp/prof_$c.log 2> /tmp/prof_$c.err || echo "PMC pass $c: timeout or failure"
  done
  rm -rf /tmp/prof_SQ
  timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
       --kernel-trace --kernel-include-regex "$KRE" --output-format csv -d /tmp/prof_SQ -- python $REPO/bench.py $BENCH_ARGS > /tmp/prof_SQ.log 2> /tmp/prof_SQ.err \
       || echo "PMC pass SQ: timeout or failure"
  rm -rf /tmp/prof_LDS
  timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CYCLES \
       --kernel-trace --kernel-include-regex "$KRE" --output-format csv -d /tmp/prof_LDS -- python $REPO/bench.py $BENCH_ARGS > /tmp/prof_LDS.log 2> /tmp/prof_LDS.err \
       || echo "PMC pass LDS: timeout or failure"
  fl=$(find /tmp/prof_LDS -name '*counter_collection.csv' 2>/dev/null | head -1)
  ff=$(find /tmp/prof_FETCH_SIZE -name '*counter_collection.csv' 2>/dev/null | head -1)
  fw=$(find /tmp/prof_WRITE_SIZE -name '*counter_collection.csv' 2>/dev/null | head -1)
  fs=$(find /tmp/prof_SQ -name '*counter_collection.csv' 2>/dev/null | head -1)
  ft=$(find /tmp/prof_SQ -name '*kernel_trace.csv' 2>/dev/null | head -1)
  tail -c 300 /tmp/prof_SQ.log
  if [ -n "$ff" ] && [ -n "$fw" ]; then
    SQARGS=""; [ -n "$fs" ] && [ -n "$ft" ] && SQARGS="--sq $fs --sq-trace $ft"
    [ -n "$fl" ] && SQARGS="$SQARGS --lds $fl"
    [ -n "$ftl" ] && SQARGS="$SQARGS --timeline-trace $ftl"
    python $REPO/tools/pmc_summary.py --window --steps $STEPS --key q200x50_db1000000_d1024_k64_w1 --fetch $ff --write $fw $SQARGS \
       > $OUT/${TAG}_pmc_traffic.json 2> $OUT/${TAG}_pmc.err || echo "pmc_summary: exit $? (see ${TAG}_pmc.err)"
    head -c 1500 $OUT/${TAG}_pmc_traffic.json
    cat $OUT/${TAG}_pmc.err | tail -3
  else
    echo "no counter_collection.csv: PMC summary skipped" | tee $OUT/${TAG}_pmc.err
    tail -5 /tmp/prof_FETCH_SIZE.err
  fi
  cd $REPO
  # the bench below quotes a PMC summary only if it was collected from the CURRENT kernel sources: put this one in place
  [ -s $OUT/${TAG}_pmc_traffic.json ] && cp $OUT/${TAG}_pmc_traffic.json $REPO/profiles/${TAG}_pmc_traffic.json
  [ -s $OUT/${TAG}_step_timeline.txt ] && cp $OUT/${TAG}_step_timeline.txt $REPO/profiles/${TAG}_step_timeline.txt
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  timeout 420 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
  tail -c 600 $OUT/${TAG}_bench_n1.json
  cd /tmp && export TMPDIR=/tmp
  # same command under the kernel trace (the CPU-baseline leg launches no kernels)
  # (--no-ubench: the micro-benchmark is a child process, rocprofv3 would trace it into a second set of files)
  rm -rf /tmp/prof_kt
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $REPO/bench.py --no-cpu-baseline --no-ubench --shard-sim 0 \
     > $OUT/${TAG}_bench_n1_under_rocprof.json 2> /tmp/prof_kt.err
  f=$(ls -S $(find /tmp/prof_kt -name '*kernel_stats.csv') 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_bench_n1_kernel_stats.csv
  # the single-image streaming pass (roofline_knn_stream): per-kernel durations of 50 passes
  rm -rf /tmp/prof_st
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python $REPO/tools/probe_stream.py 50 \
     > $OUT/${TAG}_stream_pass.txt 2> /tmp/prof_st.err
  f=$(find /tmp/prof_st -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_stream_pass_kernel_stats.csv
  cd $REPO
  # micro-benchmarks behind the ceilings quoted in DESIGN.md (MFMA rate under the power cap; gather bandwidth vs bytes in flight)
  { timeout 120 ./tools/ubench/mfma_peak 20000; timeout 120 ./tools/ubench/gather_bw; timeout 120 ./tools/ubench/lds_dma_bw; } > $OUT/${TAG}_ubench.txt 2>&1
  # the plain same-shape GEMM yardstick (torch.matmul fp16, result written) on its own
  timeout 120 python tools/yardstick_gemm.py > $OUT/${TAG}_yardstick_gemm.json 2>/dev/null
fi
