#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_bench_step.py -q ) > $OUT/r3c12_step.log 2>&1; echo "bench step tests rc=$?"; tail -15 $OUT/r3c12_step.log
