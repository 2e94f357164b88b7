cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p2
cat > /tmp/ss.py <<PY
import sys, json
sys.argv=["bench.py"]
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
a=bench.parse()
r=bench.shard_sim(a, 8)
print(json.dumps({k:r[k] for k in ("per_rank_ms","stages_ms")}))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -- python /tmp/ss.py > /tmp/p2.json 2>/tmp/p2.err
tail -1 /tmp/p2.json
f=$(find /tmp/p2 -name "*kernel_trace.csv" | head -1)
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last vote kernel ends the last step; walk back to the previous vote kernel
ends=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("void vote_kernel") or "vote_kernel" in r["Kernel_Name"]]
i1=ends[-1]; i0=ends[-2]
t0=int(rows[i0]["End_Timestamp"])
prev=t0
tot_gap=0
for r in rows[i0+1:i1+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3
    tot_gap+=max(gap,0)
    print("%8.1f us  +%6.1f gap  %7.1f us  %s"%((s-t0)/1e3, gap, (e-s)/1e3, r["Kernel_Name"][:80]))
    prev=max(prev,e)
print("step %.1f us, gaps %.1f us"%((prev-t0)/1e3, tot_gap))
PY
