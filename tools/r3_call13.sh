#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 200 python tools/probe_stream.py 50 2>&1 | grep -v amdgpu.ids
( time timeout 2400 python -m pytest tests/ -q -m gpu ) > $OUT/r3c13_all.log 2>&1; echo "all gpu tests rc=$?"; tail -12 $OUT/r3c13_all.log
