#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_frows.py -q -k "pca_fit" > $OUT/r3c9_f2.log 2>&1; echo "f2 tests rc=$?"; tail -15 $OUT/r3c9_f2.log
timeout 600 python tools/probe_pca_fit.py 50000 49152 1024 8 2>&1 | grep -v amdgpu.ids | tee $OUT/r3c9_fit_49152.json | cut -c1-700
timeout 600 python tools/probe_pca_fit.py 50000 98304 1024 8 2>&1 | grep -v amdgpu.ids | tee $OUT/r3c9_fit_98304.json | cut -c1-700
