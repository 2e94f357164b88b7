#!/bin/bash
set -u
timeout 500 python tools/probe_cfg_ab.py 250 7 9 2 5 1 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c14_ab.log
