"""PCA fit at the reference's scale (place_rec_pca.py:330-342, 380-411: <= 50 000 sampled segments x K*D -> 1024, whitened),
resident on the device (pca_fit.fit_pca_device).  Synthetic descriptors: unit rows with a decaying spectrum + noise.
    python tools/probe_pca_fit.py [n] [KD] [P] [n_iter]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd import pca_fit  # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
kd = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
p = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
n_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
g = torch.Generator(device=dev)
g.manual_seed(7)
r = 1536
spec = torch.logspace(0, -2.5, r, device=dev)
X = torch.empty(n, kd, device=dev)
Bm = torch.randn(r, kd, device=dev, generator=g) / kd ** 0.5
for a in range(0, n, 5000):
    b = min(n, a + 5000)
    X[a:b] = (torch.randn(b - a, r, device=dev, generator=g) * spec) @ Bm + 0.002 * torch.randn(b - a, kd, device=dev, generator=g) \
        + 0.01 * Bm[0]
    X[a:b] = torch.nn.functional.normalize(X[a:b], dim=1)
del Bm
torch.cuda.synchronize()
tm = {}
t0 = time.perf_counter()
mean, comps, var = pca_fit.fit_pca_device(eng, X, n_components=p, n_iter=n_iter, seed=1, timings=tm)
torch.cuda.synchronize()
tm["wall_s"] = time.perf_counter() - t0
# sanity: orthonormal rows; the fitted variances are those of the projected data
ct = torch.as_tensor(comps[:64]).to(dev).double()
orth = float((ct @ ct.t() - torch.eye(64, device=dev, dtype=torch.float64)).abs().max())
sub = X[:: max(1, n // 4096)].double() - torch.as_tensor(mean).to(dev).double()
proj_var = ((sub @ torch.as_tensor(comps[:16]).to(dev).double().t()) ** 2).mean(0).cpu().numpy()
tm.update(orthonormality_err=orth, var_top16=var[:16].tolist(), var_top16_of_a_row_subsample=proj_var.tolist(),
          var_last=float(var[-1]), peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)
print(json.dumps(tm))
