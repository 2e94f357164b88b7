#!/usr/bin/env python3
"""Bounded replay of the bench workload's three dominant kernels for rocprofv3 --pmc passes (a few dozen dispatches
instead of the 600 k torch dispatches of bench.py's synthetic-image factory, under which counter collection stalled in
round 1):

  * knn_f16_filter_kernel ... segvlad_search of 10 000 query segments x 1 M rows x 1024  (NQ, NR, D)
  * token_norms_kernel + gemm_f16x3_kernel (grouped) + project_aggregate_kernel ... segvlad_images_pca of B = 200 images, K = 64, D = 1536, N = 1530,
    S = 50, P = 1024 (the fused VLAD -> PCA call of the bench step)
  * one torch.sign over 1 GiB as the byte-count calibration of FETCH_SIZE / WRITE_SIZE (tools/pmc_summary.py)

Prints one line per part with HIP-event times, so the same script doubles as an A/B timing probe."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revisit_anything_amd import synth
from revisit_anything_amd.engine import SegVLADEngine

NR = int(os.environ.get("NR", 1000000))
NQ = int(os.environ.get("NQ", 10000))
D = int(os.environ.get("D", 1024))
B = int(os.environ.get("B", 200))
REPS = int(os.environ.get("REPS", 2))
PARTS = os.environ.get("PARTS", "knn,vlad,cal").split(",")

eng = SegVLADEngine(0)
dev = eng.device
g = torch.Generator(device=dev)
g.manual_seed(0)
for kv in os.environ.get("OPTS", "").split(","):
    if "=" in kv:
        eng.set_option(*kv.split("=", 1))

if "knn" in PARTS:
    R = torch.nn.functional.normalize(torch.randn(NR, D, device=dev, generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(NQ, D, device=dev, generator=g), dim=1)
    eng.db_add(R)
    eng.search(Q, 200)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.profile_reset()
    t0 = time.time()
    for _ in range(REPS):
        eng.search(Q, 200)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / REPS
    gm, nl = eng.stage_ms("knn_gemm")
    l0 = eng.stage_ms("knn_level0")[0] if NR > 32768 else 0.0
    sel = eng.stage_ms("knn_select")[0]
    eng.set_profiling(False)
    print(f"knn nq={NQ} nr={NR} d={D}: wall {dt * 1e3:.2f} ms; filter {gm / REPS:.2f} ms in {nl // REPS} launches "
          f"({2 * NQ * NR * D / (gm / REPS) / 1e9:.0f} TF algorithmic), level0 {l0 / REPS:.2f} ms, select+refine {sel / REPS:.2f} ms; "
          f"{eng.search_stats()}")
    del R, Q
    eng.db_reset()

if "vlad" in PARTS:
    K, Dd, N, S, P = 64, 1536, 1530, 50, 1024
    C = synth.make_vocab(K, Dd, seed=1000)
    eng.set_vocab(C)
    comps = torch.randn(P, K * Dd, device=dev, generator=g) / (K * Dd) ** 0.5
    mean = torch.randn(K * Dd, device=dev, generator=g) * (0.2 / (K * Dd) ** 0.5)
    var = torch.logspace(-3, -6, P, device=dev)
    eng.pca_set(mean, comps, var, whiten=True)
    del comps
    z = torch.randint(0, K, (B, N), device=dev, generator=g)
    Ct = torch.from_numpy(C).to(dev)
    x = torch.empty(B, Dd, N, device=dev)
    for b0 in range(0, B, 20):
        xb = Ct[z[b0:b0 + 20]] + 0.05 * torch.randn(z[b0:b0 + 20].shape[0], N, Dd, device=dev, generator=g)
        x[b0:b0 + 20] = torch.nn.functional.normalize(xb, dim=2).permute(0, 2, 1)
    masks = torch.from_numpy(np.stack([synth.make_masks(S, 240, 320, seed=2000 + b) for b in range(8)]).reshape(8 * S, 240, 320)
                             .astype(np.uint8)).to(dev).repeat((B + 7) // 8, 1, 1)[:B * S].contiguous()
    offs = (np.arange(B + 1) * S).astype(np.int32)
    bits, cent = eng.incidence_centroids(masks, 480, 640)
    adj = eng.adjacency(cent, offs, 3)
    eng.seg_vlad_pca(x, bits, offs, adj, l2norm=True)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.profile_reset()
    for _ in range(REPS):
        bits, cent = eng.incidence_centroids(masks, 480, 640)
        adj = eng.adjacency(cent, offs, 3)
        eng.seg_vlad_pca(x, bits, offs, adj, l2norm=True)
    torch.cuda.synchronize()
    st = {s: round(eng.stage_ms(s)[0] / REPS, 3) for s in ("incidence", "adjacency", "assign", "prep", "aggregate", "pca")}
    eng.set_profiling(False)
    print(f"vlad+pca B={B} K={K}: {st} ms; pca {3 * 2 * B * S * K * Dd * P / st['pca'] / 1e9:.0f} TF of fp16 products")

if "cal" in PARTS:   # known 1 GiB read + 1 GiB write for tools/pmc_summary.py
    cal = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
    torch.cuda.synchronize()
    cal2 = cal.sign()
    torch.cuda.synchronize()
    del cal, cal2
