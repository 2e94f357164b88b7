#!/bin/bash
set -u
for nw in 8 4 8 4; do
  PARTS=vlad REPS=5 OPTS="pj_nw=$nw" timeout 200 python tools/probe_counters.py 2>&1 | grep -v amdgpu.ids | sed "s/^/pj_nw=$nw: /"
done
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py -q -k "pca or fused or project" 2>&1 | tail -4
