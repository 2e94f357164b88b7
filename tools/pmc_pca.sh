#!/bin/bash
# usage: pmc_pca.sh OPTS  -> FETCH_SIZE + SQ busy of gemm_f16x3 under given options
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for o in "$@"; do
  rm -rf /tmp/pp
  PARTS=vlad,cal REPS=1 OPTS=$o timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pp -- python $REPO/tools/probe_counters.py > /tmp/pp.log 2>&1
  f=$(find /tmp/pp -name '*counter_collection.csv' | head -1)
  python - "$f" "$o" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0][-70:]].append(float(r["Counter_Value"]))
cal = [sum(v)/len(v) for k, v in agg.items() if "sign_kernel" in k]
sc = (1 << 30) / cal[0] if cal else 2048.0
for k, v in agg.items():
    if "gemm_f16x3" in k or "aggregate" in k or "splitk" in k:
        print(f"{sys.argv[2]:12s} {k[-60:]:60s} n={len(v)} fetch/launch = {sum(v)/len(v)*sc/1e9:.2f} GB")
PY
done
