# HBM-side traffic (L2 fills) of the exact refinement on raw 98 304-d rows, grouped against per row: rocprofv3 --pmc FETCH_SIZE over
# tools/probe_refine_group.py (10 000 queries x 200-deep bands; FETCH_SIZE unit = 2048 B on gfx950, tools/pmc_summary.py's calibration)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/probe_refine_group.py 98304 250 200 200 > /tmp/pm.log 2>/tmp/pm.err
grep "refine probe" /tmp/pm.log
f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
python - "$f" <<PY
import csv,sys,collections
import os
pats=os.environ.get("PMC_KERNELS","refine").split(",")    # PMC_KERNELS=refine,knn_f16_filter: the deep-row filter launches as well
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if not any(p in k for p in pats): continue
    k=k.split("(")[0][-48:]
    acc[k]+=float(r["Counter_Value"]); n[k]+=1
for k,v in acc.items():
    print("%-50s launches %3d  L2 fills per launch %.2f GB"%(k, n[k], v/n[k]*2048/1e9))
PY
