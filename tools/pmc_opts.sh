#!/bin/bash
# L2 fills (FETCH_SIZE, calibrated on a 1 GiB torch.sign) of the kNN filter launches for a list of option sets, one process each:
#   tools/pmc_opts.sh <repo> "f16_pol=0" "f16_pol=1" ...
cd /tmp && export TMPDIR=/tmp
REPO=$1; shift
for o in "$@"; do
  tag=$(echo "$o" | tr -c 'a-zA-Z0-9' '_')
  rm -rf /tmp/prof_o$tag
  OPTS="$o" PARTS=knn,cal REPS=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_o$tag -- python $REPO/tools/probe_counters.py > /tmp/prof_o$tag.log 2> /tmp/prof_o$tag.err || echo "pass $o failed"
  echo "== $o: $(grep '^knn' /tmp/prof_o$tag.log | cut -c1-110)"
  f=$(find /tmp/prof_o$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']=='FETCH_SIZE': d[r['Kernel_Name']].append(float(r['Counter_Value']))
cal=[v for k,v in d.items() if 'sign_kernel' in k]
scale=(1<<30)/(sum(cal[0])/len(cal[0])) if cal else 0.0
for k,v in d.items():
    if 'knn_f16_filter' in k and 'true' in k[:70]: print('   GB per launch', [round(x*scale/1e9,2) for x in v])
PY
done
