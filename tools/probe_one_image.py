"""ONE query image through the whole path (describe -> search 200 -> keep 50 -> vote) on the bench's shapes: wall time per
image and per-stage times; under rocprofv3 --kernel-trace the kernel timeline of an image.
    python tools/probe_one_image.py [reps] [pca_path]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd import synth  # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402
from revisit_anything_amd.pipeline import SegVLADPipeline  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pca_path = sys.argv[2] if len(sys.argv) > 2 else "auto"
dev = torch.device("cuda:0")
S, K, D, P, H, W = 50, 64, 1536, 1024, 480, 640
N = (H // 14) * (W // 14)
eng = SegVLADEngine(0)
C_np = synth.make_vocab(K, D, seed=1000)
eng.set_vocab(C_np)
g = torch.Generator(device=dev)
g.manual_seed(5000)
comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
eng.pca_set(mean, comps, torch.logspace(-3, -6, P, device=dev), whiten=True)
del comps
eng.set_option("pca_path", pca_path)
pipe = SegVLADPipeline(eng, H, W, 14, order=3, use_pca=True)
n = 1_000_000
R = torch.nn.functional.normalize(torch.randn(n, P, device=dev, generator=g), dim=1)
eng.db_add(R, torch.arange(n, device=dev, dtype=torch.int32) // S)
tok = torch.from_numpy(synth.make_tokens(C_np, N, seed=2000)[None]).to(dev)
msk = torch.from_numpy(synth.make_masks(S, H // 2, W // 2, seed=2000)).to(dev)
off = np.array([0, S], dtype=np.int32)


def one():
    qd = pipe.describe(tok, msk, off)
    d2, idx = eng.search(qd, 200)
    sims, m = eng.sims_from_d2(d2, idx, 50)
    return eng.vote(m, sims, off, n_top=5)


for _ in range(5):
    one()
torch.cuda.synchronize()
eng.set_profiling(True)
eng.profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
print(f"one image end to end (pca_path {pca_path}): {dt:.3f} ms wall")
tot = 0.0
for s in ("incidence", "centroids", "adjacency", "assign", "prep", "aggregate", "pca", "knn_level0", "knn_gemm", "knn_select", "vote"):
    try:
        ms, nl = eng.stage_ms(s)
    except Exception:
        continue
    tot += ms / reps
    print(f"  {s}: {ms / reps * 1e3:.1f} us, {nl / reps:.1f} launches")
print(f"  stages sum {tot * 1e3:.1f} us")
