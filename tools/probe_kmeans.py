"""Vocabulary k-means at the reference's scale (vlad_c_centers_pt_gen.py: K = 32 clusters of D = 1536 over ~10^6 DINOv2
tokens): seconds per Lloyd iteration with the device half-step (vocabulary.DeviceBackend).   python tools/probe_kmeans.py [images] [iters]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd import synth, vocabulary as vq  # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 654          # 654 x 1530 = 1 000 620 tokens
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K, D, N = 32, 1536, 1530
dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
C0 = torch.from_numpy(synth.make_vocab(K, D, seed=3)).to(dev)
g = torch.Generator(device=dev)
g.manual_seed(4)
toks = torch.empty(B, D, N, device=dev)
for b in range(B):
    z = torch.randint(0, K, (N,), device=dev, generator=g)
    toks[b] = torch.nn.functional.normalize(C0[z] + 0.08 * torch.randn(N, D, device=dev, generator=g), dim=1).t()
db = vq.DeviceBackend(eng, toks, batch=64)
torch.cuda.synchronize()
t0 = time.perf_counter()
cen, labels, it = vq.cosine_kmeans(num_clusters=K, backend=db, seed=5, max_iter=iters, tol=0.0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"tokens": B * N, "K": K, "D": D, "iterations": it, "seconds": dt, "seconds_per_iteration": dt / it,
                  "tokens_per_second": B * N * it / dt, "centre_norm_min_max": [float(np.linalg.norm(cen, axis=1).min()),
                                                                                 float(np.linalg.norm(cen, axis=1).max())],
                  "cluster_size_min_max": [int(np.bincount(labels, minlength=K).min()), int(np.bincount(labels, minlength=K).max())]}))
