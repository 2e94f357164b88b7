#!/usr/bin/env python3
"""PCA-projection probe: n rows of K*D -> P (timing of the pca stage only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from revisit_anything_amd.engine import SegVLADEngine
from revisit_anything_amd import synth

n = int(os.environ.get("NROWS", 10000)); kd = int(os.environ.get("KD", 98304)); P = int(os.environ.get("P", 1024)); reps = int(os.environ.get("REPS", 5))
eng = SegVLADEngine(0)
g = torch.Generator(device=eng.device); g.manual_seed(0)
mean = torch.randn(kd, device=eng.device, generator=g) * 0.01
comps = torch.randn(P, kd, device=eng.device, generator=g) / kd ** 0.5
var = torch.rand(P, device=eng.device, generator=g) + 0.5
eng.pca_set(mean, comps, var, whiten=True)
X = torch.nn.functional.normalize(torch.randn(n, kd, device=eng.device, generator=g), dim=1)
eng.pca_apply(X, l2norm=True); torch.cuda.synchronize()
eng.set_profiling(True); eng.profile_reset()
for _ in range(reps):
    eng.pca_apply(X, l2norm=True)
torch.cuda.synchronize()
ms = eng.stage_ms("pca")[0] / reps
print(f"pca n={n} kd={kd} P={P}: {ms:.2f} ms -> {2*n*kd*P/ms/1e9:.1f} TF fp32-equivalent")
