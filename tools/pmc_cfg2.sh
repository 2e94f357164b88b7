#!/bin/bash
# L2 fills (FETCH_SIZE) of the deep-row filter launches of the config2 step (raw 98 304-d rows), per tile walk: tools/pmc_cfg2.sh (gpurun)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for w in ${WALKS:-3 0}; do
  rm -rf /tmp/prof_c$w
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "knn_f16_filter_kernel|sign_kernel" --output-format csv -d /tmp/prof_c$w -- \
    python $REPO/bench.py --no-pca --db-images 1000 --pmc-calibrate --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --steps 2 --warmup 1 --set f16_walk=$w > /tmp/prof_c$w.log 2> /tmp/prof_c$w.err || echo "pass $w failed"
  python3 -c "import sys,json; j=json.loads(open('/tmp/prof_c$w.log').read().strip().splitlines()[-1]); print('walk $w', j['value'], j['stages_ms_per_step']['knn_gemm'])"
  f=$(find /tmp/prof_c$w -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']=='FETCH_SIZE': d[r['Kernel_Name']].append(float(r['Counter_Value']))
cal=[v for k,v in d.items() if 'sign_kernel' in k]
scale=(1<<30)/(sum(cal[0])/len(cal[0])) if cal else 0.0
print('calibration scale', scale)
for k,v in d.items():
    if 'knn_f16_filter' in k: print(k[22:100], 'launches',len(v), 'GB per launch', [round(x*scale/1e9,1) for x in v][-3:])
PY
done
