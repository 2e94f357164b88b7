#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=$1
for w in 0 2 3; do
  rm -rf /tmp/prof_w$w
  OPTS=f16_walk=$w PARTS=knn,cal REPS=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_w$w -- python $REPO/tools/probe_counters.py > /tmp/prof_w$w.log 2> /tmp/prof_w$w.err || echo "pass $w failed"
  grep "^knn" /tmp/prof_w$w.log
  f=$(find /tmp/prof_w$w -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']=='FETCH_SIZE': d[r['Kernel_Name']].append(float(r['Counter_Value']))
cal=[v for k,v in d.items() if 'sign_kernel' in k]
scale=(1<<30)/(sum(cal[0])/len(cal[0])) if cal else 0.0
print('calibration: raw per GiB', cal[0] if cal else None, 'scale', scale)
for k,v in d.items():
    if 'knn_f16_filter' in k: print(k[:75], 'launches',len(v), 'raw', [round(x) for x in v], 'GB per launch', [round(x*scale/1e9,2) for x in v])
PY
done
