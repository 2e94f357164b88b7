cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py $BARGS --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --steps 3 --warmup 1 --search-stats > /tmp/p1.json 2>/tmp/p1.err
f=$(ls -S $(find /tmp/p1 -name "*kernel_stats.csv") | head -1)
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:40]:
    n=r["Name"]
    if "at::native" in n or "rocclr" in n: continue
    print(n[:70], r["Calls"], "%.3f ms tot"%(float(r["TotalDurationNs"])/1e6), "%.3f avg"%(float(r["AverageNs"])/1e6))
PY
python -c "
import json;r=json.loads(open('/tmp/p1.json').read().strip().splitlines()[-1]);print(r['value'],r['stages_ms_per_step'],r.get('search_stats'))"
