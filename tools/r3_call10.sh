#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/ -q -m gpu -x ) > $OUT/r3c10_all.log 2>&1; echo "all gpu tests rc=$?"; tail -12 $OUT/r3c10_all.log
timeout 600 python bench.py > $OUT/r3c10_bench.json 2> $OUT/r3c10_bench.err; echo "bench rc=$?"; tail -c 400 $OUT/r3c10_bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3c10_bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','ms_per_step_hip_event_median','recall_at_1','mode','pca_path')})
print('stages',j['stages_ms_per_step'])
print('roof',{k:j['roofline'].get(k) for k in ('frac','mfma_only_ceiling_ms','frac_of_mfma_only_ceiling')})
print('stream',{k:j['roofline_knn_stream'][k] for k in ('frac','filter_only_frac','pass_ms','filter_ms','select_refine_ms')})
c=j.get('config2'); print('config2',{k:c.get(k) for k in ('value','ms_per_step','search_stats','stages_ms_per_step','error')})
c=j.get('redundant_db'); print('redundant',{k:c.get(k) for k in ('value','ms_per_step','recall_at_1','recall_at_1_within_sibling_group','search_stats','error')})
print('oracle',j.get('oracle_check'))
print('cpu',j.get('cpu_baseline'))
PY
