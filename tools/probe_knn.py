#!/usr/bin/env python3
"""Search-only probe (for PMC runs): nq x n_r x d exact kNN, a few repetitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from revisit_anything_amd.engine import SegVLADEngine

nr = int(os.environ.get("NR", 200000)); nq = int(os.environ.get("NQ", 4096)); d = int(os.environ.get("D", 1024)); reps = int(os.environ.get("REPS", 3))
eng = SegVLADEngine(0)
g = torch.Generator(device=eng.device); g.manual_seed(0)
R = torch.nn.functional.normalize(torch.randn(nr, d, device=eng.device, generator=g), dim=1)
Q = torch.nn.functional.normalize(torch.randn(nq, d, device=eng.device, generator=g), dim=1)
eng.db_add(R)
eng.search(Q, 200); torch.cuda.synchronize()
eng.set_profiling(True); eng.profile_reset()
t0 = time.time()
for _ in range(reps):
    eng.search(Q, 200)
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
gm = eng.stage_ms("knn_gemm")[0] / reps
try:
    gm += eng.stage_ms("knn_level0")[0] / reps   # the sampled level's exact GEMM (only on the large-database path)
except Exception:
    pass
if os.environ.get("PMC_CAL"):   # known 1 GiB read + 1 GiB write for tools/pmc_summary.py (run under rocprofv3 --pmc)
    cal = torch.empty(1 << 28, dtype=torch.float32, device=eng.device).normal_()
    torch.cuda.synchronize()
    cal2 = cal.sign()
    torch.cuda.synchronize()
    del cal, cal2
print(f"search nq={nq} nr={nr} d={d}: wall {dt*1e3:.2f} ms, gemm {gm:.2f} ms -> {2*nq*nr*d/gm/1e9:.1f} TF (algorithmic), select {eng.stage_ms('knn_select')[0]/reps:.2f} ms")
