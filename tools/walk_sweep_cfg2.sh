for w in 3 0 1 2 4 5 6 7 3 0; do
  timeout 300 python bench.py --no-pca --db-images 1000 --no-cpu-baseline --no-ubench --no-sub-records --shard-sim 0 --steps 4 --warmup 1 --set f16_walk=$w 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk $w', round(j['value'],1), j['stages_ms_per_step']['knn_gemm'], j['pred_sha1'])"
done
for g in 4 16 2; do
  timeout 300 python bench.py --no-pca --db-images 1000 --no-cpu-baseline --no-ubench --no-sub-records --shard-sim 0 --steps 4 --warmup 1 --set f16_walk=0 --set f16_gm=$g 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk 0 gm $g', round(j['value'],1), j['stages_ms_per_step']['knn_gemm'], j['pred_sha1'])"
done
