#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sharded.py -q > $OUT/r3c8_sh.log 2>&1; echo "sharded tests rc=$?"; tail -15 $OUT/r3c8_sh.log
timeout 200 python tools/probe_stream.py 50 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_bench_shapes.py -q -k "knn" > $OUT/r3c8_regr.log 2>&1; echo "regression rc=$?"; tail -3 $OUT/r3c8_regr.log
