"""How much of the single-image filter launch is its last, partly filled round of workgroups?  The 64 x 128 streaming kernel runs 512
resident workgroups (2 per CU): 1 M rows = 7813 tiles = 15.26 rounds.  Same pass over 983 040 rows (15 rounds exactly), 1 000 000 and
1 048 576 (16 rounds): if the filter's time follows the ROUNDS (ceil) rather than the rows, a persistent form has that much to gain."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
g = torch.Generator(device=dev)
g.manual_seed(1)
d, k, reps = 1024, 200, 60
Rall = torch.nn.functional.normalize(torch.randn(1048576, d, device=dev, generator=g), dim=1)
Q = torch.nn.functional.normalize(Rall[torch.arange(50, device=dev) * 977] + 0.03 * torch.randn(50, d, device=dev, generator=g), dim=1)
for n in (983040, 1000000, 1048576, 917504, 983040):
    eng.db_reset()
    eng.db_add(Rall[:n])
    for _ in range(5):
        eng.search(Q, k)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.profile_reset()
    for _ in range(reps):
        eng.search(Q, k)
    torch.cuda.synchronize()
    ms, nl = eng.stage_ms("knn_gemm")
    eng.set_profiling(False)
    f = ms / reps
    print(f"n = {n}: {n / 128 / 512:.2f} rounds, filter {f * 1e3:.1f} us = {n * 2048 / (f * 1e-3) / 1e12:.2f} TB/s, {f * 1e3 / (n / 128 / 512):.2f} us per round")
