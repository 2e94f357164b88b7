#!/bin/bash
# GPU idle time inside the bench's steps: kernel trace of a short run, gaps between consecutive dispatches of the last steps
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_gap
SEGVLAD_GUARD=0 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_gap -- python $REPO/bench.py --steps 4 --warmup 2 --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --verify-images 0 > /tmp/gap.json 2>/tmp/gap.err
tail -c 300 /tmp/gap.json
f=$(find /tmp/prof_gap -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/prof_gap -name "*memory_copy_trace.csv" | head -1)
python3 - "$f" "$m" <<'PY'
import csv,sys
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
try:
    rows+=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),'MEMCPY '+r.get('Direction','')) for r in csv.DictReader(open(sys.argv[2]))]
except Exception as e: print("no memcpy trace", e)
rows.sort()
# a step = from one describe's first incidence launch to the next step's: two incidence launches per step (two describe
# chunks); keep the windows that hold a full-level filter launch (> 5 ms) -- the batch steps, not the one-image passes
starts=[i for i,r in enumerate(rows) if 'incidence_fused' in r[2]]
wins=[]
for a_,b_ in zip(starts, starts[1:]+[len(rows)]):
    seg=rows[a_:b_]
    if any('knn_f16_filter' in r[2] and r[1]-r[0] > 5e6 for r in seg): wins.append((a_,b_))
print("batch-step windows", len(wins))
for a_,b_ in wins[-3:]:
    # the step starts at the previous incidence launch if that one had no filter (first describe chunk of the step)
    k=starts.index(a_)
    if k>0 and not any('knn_f16_filter' in r[2] and r[1]-r[0] > 5e6 for r in rows[starts[k-1]:a_]): a_=starts[k-1]
    seg=rows[a_:b_]
    busy=sum(e-b for b,e,_ in seg); span=seg[-1][1]-seg[0][0]
    gaps=[(seg[k+1][0]-seg[k][1], seg[k][2][:46], seg[k+1][2][:46]) for k in range(len(seg)-1)]
    big=sorted(gaps,reverse=True)[:14]
    print(f"step: {len(seg)} dispatches, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms; gaps > 5 us: {sum(1 for g in gaps if g[0]>5e3)} totalling {sum(g[0] for g in gaps if g[0]>5e3)/1e6:.3f} ms")
    for g,x,y in big: print(f"    gap {g/1e3:7.1f} us  after {x}  before {y}")
PY
