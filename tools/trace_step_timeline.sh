# kernel timeline of ONE timed step of bench.py (between the last two vote kernels): start, gap to the previous kernel's end, duration
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p4
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p4 -- python $GRAFT_REPO_ROOT/bench.py --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --steps 3 --warmup 1 > /tmp/p4.json 2>/tmp/p4.err
f=$(find /tmp/p4 -name "*kernel_trace.csv" | head -1)
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ends=[i for i,r in enumerate(rows) if "vote_kernel" in r["Kernel_Name"]]
# the timed steps are the last 3 before the pipelined/other extras: take the pair with a full step in between
i1=ends[3]; i0=ends[2]
t0=int(rows[i0]["End_Timestamp"]); prev=t0; tot_gap=0
for r in rows[i0+1:i1+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3
    if gap>0: tot_gap+=gap
    print("%8.1f us  +%7.1f gap  %8.1f us  %s"%((s-t0)/1e3, gap, (e-s)/1e3, r["Kernel_Name"][:70]))
    prev=max(prev,e)
print("step %.1f us, gaps %.1f us"%((prev-t0)/1e3, tot_gap))
PY
