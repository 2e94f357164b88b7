cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SEGVLAD_GUARD=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -- python $REPO/tools/probe_stream.py 8 > /tmp/tr.log 2>&1
tail -3 /tmp/tr.log
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 3 passes: find the filter kernels
idx=[i for i,r in enumerate(rows) if 'knn_f16_filter' in r['Kernel_Name']]
last=idx[-2]
# print kernels from 8 before to 6 after
t0=None
for r in rows[last-8:last+8]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} dur {(e-s)/1e3:7.1f}  {r['Kernel_Name'][:60]}  grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')}")
PY
