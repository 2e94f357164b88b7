#!/bin/bash
# PMC pass over tools/probe_knn.py; prints per-kernel counter averages.  usage: tools/pmc_probe.sh "CTR1 CTR2 ..." [kernel-substring]
CTRS=${1:-SQ_WAVES}
PAT=${2:-knn_f16_filter}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmcp -- python $REPO/tools/probe_knn.py > /tmp/pmcp.log 2>&1
f=$(find /tmp/pmcp -name '*counter_collection.csv' | head -1)
python - "$f" "$PAT" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"], r["Grid_Size"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, t) in sorted(agg.items()):
    print(f"{k[0]:60s} grid={k[2]:>10s} {k[1]:28s} n={n:3d} avg={t/n:.4g}")
PY
