#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_config2_redundant.py -q > $OUT/r3c2_new.log 2>&1; echo "new tests rc=$?"
grep -n "\[config2\]\|\[redundant_db\]\|passed\|failed\|Error" $OUT/r3c2_new.log | cut -c1-700
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "raw_descriptor or overflow or dense_block" > $OUT/r3c2_regr.log 2>&1; echo "regression rc=$?"; tail -3 $OUT/r3c2_regr.log
# bisect the 45 ms/step the sub-records lose
for v in "" "--search-stats" "--group 31"; do
  timeout 300 python bench.py --no-cpu-baseline --no-sub-records --no-ubench --steps 3 --warmup 1 $v > $OUT/r3c2_b.json 2>/dev/null
  python - "$v" <<'PY'
import json,sys
j=json.loads(open('gpurun_out/r3c2_b.json').read().strip().splitlines()[-1])
print('variant [%s]'%sys.argv[1], {k:round(j[k],2) for k in ('value','ms_per_step','ms_per_step_hip_event_median')}, 'stage sum %.2f'%sum(j['stages_ms_per_step'].values()))
PY
done
SEGVLAD_F16_CFG=50 timeout 300 python bench.py --pipeline --no-sub-records --no-cpu-baseline --no-ubench > $OUT/r3c2_pipe50.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c2_pipe50.json').read().strip().splitlines()[-1])
print('pipeline + f16_cfg=50', {k:round(j[k],2) for k in ('value','ms_per_step','ms_per_step_hip_event_median')})
PY
SEGVLAD_F16_CFG=50 timeout 300 python bench.py --no-sub-records --no-cpu-baseline --no-ubench > $OUT/r3c2_ser50.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c2_ser50.json').read().strip().splitlines()[-1])
print('serial + f16_cfg=50', {k:round(j[k],2) for k in ('value','ms_per_step','ms_per_step_hip_event_median')})
PY
timeout 200 python tools/probe_stream.py 50 > $OUT/r3c2_stream.log 2>&1; cat $OUT/r3c2_stream.log
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_st -- python $REPO/tools/probe_stream.py 50 > /tmp/prof_st.log 2>&1
f=$(find /tmp/prof_st -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $REPO/$OUT/r3c2_stream_kernel_stats.csv
cd $REPO; python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3c2_stream_kernel_stats.csv')))
for r in rows:
    if int(r['Calls'])>=50 and 'at::' not in r['Name']:
        print(r['Name'][:70], r['Calls'], '%.1f us avg'%(float(r['AverageNs'])/1e3))
PY
