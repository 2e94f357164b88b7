#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_config2_redundant.py -q > $OUT/r3c6_new.log 2>&1; echo "new tests rc=$?"
grep -n "\[config2\]\|\[redundant_db\]\|passed\|failed\|Error" $OUT/r3c6_new.log | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q > $OUT/r3c6_regr.log 2>&1; echo "regression rc=$?"; tail -3 $OUT/r3c6_regr.log
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python $REPO/tools/probe_stream.py 50 > $REPO/$OUT/r3c6_stream.log 2>&1
f=$(find /tmp/prof_st -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-16:]
t0=int(tail[0]['Start_Timestamp'])
for r in tail:
    print('%8.1f us +%6.1f  %s'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:60]))
PY
cd $REPO; grep -v "amdgpu.ids\|rocprofv3\|Opened" $OUT/r3c6_stream.log | tail -6
