# per-dispatch durations of the refinement kernels of ONE config2 step (raw 98 304-d rows): tools/trace_refine_cfg2.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p2
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -- python $GRAFT_REPO_ROOT/bench.py --no-pca --db-images 1000 --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --steps 2 --warmup 1 > /tmp/p2.json 2>/tmp/p2.err
f=$(find /tmp/p2 -name "*kernel_trace.csv" | head -1)
python - "$f" <<PY
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
votes=[i for i,r in enumerate(rows) if "vote_kernel" in r["Kernel_Name"]]
lo=votes[-2]+1; hi=votes[-1]
t0=int(rows[lo]["Start_Timestamp"]); prev=t0
for r in rows[lo:hi+1]:
    n=r["Kernel_Name"]
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if (e-s)>20000 or "refine" in n or "select" in n:
        print("%10.1f us +%7.1f gap %10.1f us  %s grid %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,n[:80],r.get("Grid_Size")))
    prev=max(prev,e)
print("step %.1f us"%((prev-t0)/1e3))
PY
