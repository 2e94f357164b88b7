#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
python -c "
from revisit_anything_amd import build as b
print(b.build())"
timeout 600 python -m pytest tests/test_gpu_config2_redundant.py -q > $OUT/r3c3_new.log 2>&1; echo "new tests rc=$?"
grep -n "\[config2\]\|\[redundant_db\]\|passed\|failed\|Error" $OUT/r3c3_new.log | cut -c1-900
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "adjacency or raw_descriptor or pipeline" > $OUT/r3c3_regr.log 2>&1; echo "regression rc=$?"; tail -3 $OUT/r3c3_regr.log
timeout 200 python tools/probe_stream.py 50 > $OUT/r3c3_stream.log 2>&1; cat $OUT/r3c3_stream.log
timeout 300 python tools/probe_cfg_ab.py 250 50 0 200 > $OUT/r3c3_ab.log 2>&1; cat $OUT/r3c3_ab.log
timeout 300 python bench.py --no-cpu-baseline --no-sub-records --no-ubench --steps 3 --warmup 1 --group 31 > $OUT/r3c3_b.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c3_b.json').read().strip().splitlines()[-1])
print('group 31', {k:round(j[k],2) for k in ('value','ms_per_step','ms_per_step_hip_event_median')}, 'stage sum %.2f'%sum(j['stages_ms_per_step'].values()))
PY
timeout 300 python bench.py --no-cpu-baseline --no-sub-records --no-ubench --steps 3 --warmup 1 --no-pca --db-images 1000 --search-stats > $OUT/r3c3_c2.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c3_c2.json').read().strip().splitlines()[-1])
print('config2', {k:round(j[k],2) for k in ('value','ms_per_step','ms_per_step_hip_event_median')}, j['stages_ms_per_step'], j['search_stats'])
PY
