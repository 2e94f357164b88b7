"""Development probe: vote kernel time against the concentration of the matches on few reference images (long runs of one
image id used to be summed by ONE thread; a workgroup per query image, so the slowest image set the kernel time).

    gpurun -- 'python tools/probe_vote.py'
"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd.engine import SegVLADEngine
from revisit_anything_amd import _lib
eng = SegVLADEngine(0); dev = eng.device
g = torch.Generator(device=dev); g.manual_seed(0)
nQ, S, k, nR = 200, 50, 50, 20000
img_of_seg = torch.arange(nR, device=dev, dtype=torch.int32).repeat_interleave(S)
eng.db_add(torch.nn.functional.normalize(torch.randn(1000, 64, device=dev, generator=g), dim=1), img_of_seg[:1000])
# hack: vote needs img_of_seg for all ids -> use ids < 1000 rows? use concentrated ids
for conc in (0.0, 0.5, 0.9):
    idx = torch.randint(0, 1000, (nQ * S, k), device=dev, generator=g)
    hot = (torch.rand(nQ * S, k, device=dev, generator=g) < conc)
    idx = torch.where(hot, torch.randint(0, 50, (nQ * S, k), device=dev, generator=g), idx).to(torch.int64)
    sims = torch.rand(nQ * S, k, device=dev, generator=g)
    offs = (np.arange(nQ + 1) * S).astype(np.int32)
    for mode in (_lib.VOTE_WT_BORDA_IM,):
        for _ in range(3): eng.vote(idx, sims, offs, n_top=5, mode=mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): eng.vote(idx, sims, offs, n_top=5, mode=mode)
        torch.cuda.synchronize(); print(f"conc {conc}: vote {(time.perf_counter()-t0)/20*1e3:.3f} ms")
