# the same sweep on the headline workload (10 000 x 1 M x 1024, level carry on): tools/walk_sweep_1024.sh (gpurun)
for w in 3 0 2 1 7 3 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-ubench --no-sub-records --shard-sim 0 --steps 10 --warmup 2 --set f16_walk=$w 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk $w', round(j['value'],1), j['stages_ms_per_step']['knn_gemm'], j['pred_sha1'])"
done
timeout 300 python bench.py --no-pca --db-images 1000 --no-cpu-baseline --no-ubench --no-sub-records --shard-sim 0 --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 default', round(j['value'],1), j['stages_ms_per_step']['knn_gemm'], j['pred_sha1'])"
