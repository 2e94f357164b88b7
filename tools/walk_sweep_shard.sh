# tile walk / block height of the batch filter on a 125 k-row shard (bench.py shard_sim, W = 8): tools/walk_sweep_shard.sh (gpurun)
for o in "f16_walk=3" "f16_walk=0" "f16_walk=2" "f16_walk=1" "f16_gm=4" "f16_gm=16" "f16_walk=3"; do
  timeout 300 python bench.py --no-cpu-baseline --no-ubench --shard-sim-only --steps 3 --warmup 1 --shard-sim 8 --set $o 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['shard_sim']; print('$o', round(s['per_rank_ms'],3), s['stages_ms']['knn_gemm'], s['stages_ms']['knn_select'])"
done
