#!/bin/bash
# round-3 GPU call 1: new kNN paths (config 2, redundant DB, tiers), regressions, bench with sub-records, pipeline mode
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_config2_redundant.py -q -x > $OUT/r3c1_new.log 2>&1; echo "new tests rc=$?"
tail -25 $OUT/r3c1_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q > $OUT/r3c1_regr.log 2>&1; echo "regression rc=$?"
tail -8 $OUT/r3c1_regr.log
timeout 600 python bench.py > $OUT/r3c1_bench.json 2> $OUT/r3c1_bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/r3c1_bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c1_bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','ms_per_step_hip_event_median','recall_at_1','mode')})
print('stages',j['stages_ms_per_step'])
print('roof',{k:j['roofline'].get(k) for k in ('frac','mfma_only_ceiling_ms','frac_of_mfma_only_ceiling','ubench','effective_clock_ghz')})
print('stream',j['roofline_knn_stream'])
print('config2',j.get('config2'))
print('redundant',j.get('redundant_db'))
print('oracle',j.get('oracle_check'))
PY
timeout 400 python bench.py --pipeline --no-sub-records --no-cpu-baseline --no-ubench > $OUT/r3c1_pipe.json 2> $OUT/r3c1_pipe.err; echo "pipeline rc=$?"
tail -c 600 $OUT/r3c1_pipe.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c1_pipe.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','ms_per_step_hip_event_median','recall_at_1','mode')})
print('stages',j['stages_ms_per_step'])
PY
