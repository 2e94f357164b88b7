#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_config2_redundant.py -q -k "config2 or second_tier" > $OUT/r3c4_new.log 2>&1; echo "new tests rc=$?"
grep -n "\[config2\]\|passed\|failed\|Error" $OUT/r3c4_new.log | cut -c1-900
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "raw_descriptor" > $OUT/r3c4_regr.log 2>&1; echo "regression rc=$?"; tail -3 $OUT/r3c4_regr.log
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python $REPO/tools/probe_stream.py 50 > $REPO/$OUT/r3c4_stream.log 2>&1
f=$(find /tmp/prof_st -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$OUT/r3c4_stream_kernel_stats.csv
f=$(find /tmp/prof_st -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
# last search pass: print the kernels of the last ~20 dispatches with start offsets
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-22:]
t0=int(tail[0]['Start_Timestamp'])
for r in tail:
    print('%8.1f us +%6.1f  %s'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:60]))
PY
cd $REPO; grep -v amdgpu.ids $OUT/r3c4_stream.log | tail -6
