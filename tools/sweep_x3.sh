# XCD block height / tile of the PCA split GEMM inside the bench step: tools/sweep_x3.sh (gpurun)
for o in "x3_gm=-1" "x3_gm=0" "x3_gm=2" "x3_gm=4" "x3_gm=8" "x3_gm=16" "x3_tile=128" "x3_gm=-1"; do
  timeout 300 python bench.py --no-cpu-baseline --no-ubench --no-sub-records --steps 10 --warmup 2 --shard-sim 0 --set $o 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stages_ms_per_step']; print('$o', round(j['value'],1), s['pca'], s['describe'], j['pred_sha1'])"
done
