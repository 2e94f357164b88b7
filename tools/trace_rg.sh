cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 4 6 7; do
rm -rf /tmp/p1
SEGVLAD_RG_ABL=$a timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --no-pca --db-images 1000 --no-sub-records --no-cpu-baseline --no-ubench --shard-sim 0 --steps 2 --warmup 1 > /tmp/p1.json 2>/tmp/p1.err
f=$(ls -S $(find /tmp/p1 -name "*kernel_stats.csv") | head -1)
echo "abl=$a: $(grep refine_group_gemm $f | cut -d, -f1-4 | cut -c1-60,150-)"
done
