#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_frows.py -q ) > $OUT/r3c11_frows.log 2>&1; echo "frows rc=$?"; tail -8 $OUT/r3c11_frows.log
timeout 600 python tools/probe_kmeans.py 654 10 2>&1 | grep -v amdgpu.ids | tee $OUT/r3c11_kmeans.json
timeout 900 python tools/e2e_throughput.py --dino giant --sam huge --ref 24 --query 8 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/r3c11_e2e.txt
