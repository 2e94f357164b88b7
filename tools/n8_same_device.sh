# The N = 8 (and N = 4) line on ONE GPU: eight ranks share cuda:0, gloo collectives (RCCL refuses several ranks per device) -- every
# piece of the N > 1 path except the transport: image-aligned shards, ragged query slices, both all-gathers, merge, vote, the line's
# per-rank stage / collective records.  Predictions must equal the N = 1 run's.
cd $GRAFT_REPO_ROOT
export SEGVLAD_GUARD=0 MASTER_ADDR=127.0.0.1
COMMON="--steps 1 --warmup 1 --db-images 4000 --query-images 40 --no-cpu-baseline --no-ubench --no-sub-records --shard-sim 0"
timeout 600 python bench.py $COMMON --dump-preds /tmp/p1.npy 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): p=json.loads(l); print('N=1', p['value'], p['recall_at_1'], p['pred_sha1'])"
for n in 4 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --same-device --dist-backend gloo $COMMON --dump-preds /tmp/p$n.npy 2>/tmp/n$n.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        p=json.loads(l); print('N=$n', p['n_gpus'], p['value'], p['recall_at_1'], p['pred_sha1'], p['scaling']); print('  rank0', json.dumps(p['per_rank_stages_ms'][0])[:600]); print('  rank$((n-1))', json.dumps(p['per_rank_stages_ms'][-1])[:300])"
python3 -c "
import numpy as np; a=np.load('/tmp/p1.npy'); b=np.load('/tmp/p$n.npy'); print('  predictions equal to N=1:', np.array_equal(a,b))"
tail -2 /tmp/n$n.err | cut -c1-300
done
