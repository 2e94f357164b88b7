"""Development probe: the two forms of the fused segment-VLAD -> PCA call (pca_path = planes | project) against each
other and the fp64 oracle at the bench shape, and their stage times at a bench-sized batch.

    gpurun -- 'python tools/probe_project.py'
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import segvlad_oracle as O   # noqa: E402  (development tool: the checker)
from revisit_anything_amd import synth   # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine   # noqa: E402

dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
K, D, N, S, P = 64, 1536, 1530, 50, 1024
C = synth.make_vocab(K, D, seed=1000)
mean, comps, var = synth.make_pca_model(K * D, P, seed=5000)
eng.set_vocab(C)
eng.pca_set(mean, comps, var, whiten=True)

# ---- correctness on 6 images ---------------------------------------------------------------------------------------------
Bc = 6
toks, incs, adjs = [], [], []
for j in range(Bc):
    tok = synth.make_tokens(C, N, seed=1001 + j)
    masks = synth.make_masks(S, 240, 320, seed=1101 + j)
    toks.append(tok)
    incs.append(O.incidence(masks, 480, 640))
    adjs.append(O.nbr_masks_agg_fast_single([m for m in masks], 3))
offs = (np.arange(Bc + 1) * S).astype(np.int32)
bits = np.concatenate([O.pack_bits_u64(i) for i in incs]).view(np.int64)
adj = np.concatenate([a.astype(np.uint8).reshape(-1) for a in adjs])
tk = np.stack(toks)
ys = {}
for path in ("planes", "project"):
    eng.set_option("pca_path", path)
    ys[path] = [eng.seg_vlad_pca(tk, bits, offs, adj, l2norm=l2)["out"].cpu().numpy().astype(np.float64) for l2 in (False, True)]
compsd = comps.astype(np.float64)
for b in range(Bc):
    ref_desc = O.seg_vlad(toks[b], incs[b], C, adjs[b])
    raw = O.pca_transform(ref_desc, mean, compsd, var, True)
    ref = O.normalize_feat(raw)
    sl = slice(b * S, (b + 1) * S)
    for path in ys:
        e_raw = np.abs(ys[path][0][sl] - raw).max() / np.abs(raw).max()
        e_n = np.abs(ys[path][1][sl] - ref).max()
        print(f"image {b} {path:8s}: raw rel err {e_raw:.2e}   unit-row abs err {e_n:.2e}")
print("project vs planes, unit rows:", np.abs(ys["project"][1] - ys["planes"][1]).max())

# ---- timing at a bench-sized batch ------------------------------------------------------------------------------------------
B = int(os.environ.get("B", 198))
REPS = 5
g = torch.Generator(device=dev).manual_seed(7)
z = torch.randint(0, K, (B, N), device=dev, generator=g)
Ct = torch.from_numpy(C).to(dev)
x = torch.empty(B, D, N, device=dev)
for b0 in range(0, B, 20):
    xb = Ct[z[b0:b0 + 20]] + 0.05 * torch.randn(z[b0:b0 + 20].shape[0], N, D, device=dev, generator=g)
    x[b0:b0 + 20] = torch.nn.functional.normalize(xb, dim=2).permute(0, 2, 1)
masks = torch.from_numpy(np.stack([synth.make_masks(S, 240, 320, seed=2000 + b) for b in range(8)]).reshape(8 * S, 240, 320)
                         .astype(np.uint8)).to(dev).repeat((B + 7) // 8, 1, 1)[:B * S].contiguous()
offs = (np.arange(B + 1) * S).astype(np.int32)
bits, cent = eng.incidence_centroids(masks, 480, 640)
adj = eng.adjacency(cent, offs, 3)
outs = {}
for path in ("planes", "project"):
    eng.set_option("pca_path", path)
    outs[path] = eng.seg_vlad_pca(x, bits, offs, adj, l2norm=True)["out"].clone()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(REPS):
        eng.seg_vlad_pca(x, bits, offs, adj, l2norm=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS
    st = {s: round(eng.stage_ms(s)[0] / REPS, 3) for s in ("assign", "prep", "aggregate", "pca")}
    eng.set_profiling(False)
    print(f"{path:8s} B={B}: wall {dt * 1e3:.2f} ms  stages {st}")
print("batch: project vs planes max abs diff (unit rows):", (outs["project"] - outs["planes"]).abs().max().item())

# determinism / safe-wait comparison of the project path
eng.set_option("pca_path", "project")
a = eng.seg_vlad_pca(x, bits, offs, adj, l2norm=False)["out"].clone()
b2 = eng.seg_vlad_pca(x, bits, offs, adj, l2norm=False)["out"].clone()
eng.set_option("debug_search", "7")
c = eng.seg_vlad_pca(x, bits, offs, adj, l2norm=False)["out"].clone()
eng.set_option("debug_search", "0")
print("repeat equal:", torch.equal(a, b2), " safe-wait equal:", torch.equal(a, c), " max diff:", (a - c).abs().max().item(),
      " rows differing:", int(((a != c).any(1)).sum()), "of", a.shape[0])

# identity "PCA": y must equal the descriptor itself -- localises any wrong token / chunk / norm
K2, D2, N2 = 4, 128, 200
C2 = synth.make_vocab(K2, D2, seed=77)
eng.set_vocab(C2)
I = np.eye(K2 * D2, dtype=np.float32)
eng.pca_set(np.zeros(K2 * D2, np.float32), I, np.ones(K2 * D2, np.float32), whiten=False)
rng = np.random.Generator(np.random.PCG64(5))
tk2, inc2, adj2 = [], [], []
for b, S2 in enumerate([7, 40, 66, 3]):
    tk2.append(synth.make_tokens(C2, N2, seed=300 + b, noise=0.3))
    inc2.append(rng.random((S2, N2)) < 0.2)
    adj2.append(np.eye(S2, dtype=bool))
offs2 = np.concatenate([[0], np.cumsum([i.shape[0] for i in inc2])]).astype(np.int32)
bits2 = np.concatenate([O.pack_bits_u64(i) for i in inc2]).view(np.int64)
adjc = np.concatenate([a.astype(np.uint8).reshape(-1) for a in adj2])
for path in ("planes", "project"):
    eng.set_option("pca_path", path)
    y = eng.seg_vlad_pca(np.stack(tk2), bits2, offs2, adjc, l2norm=False)["out"].cpu().numpy().astype(np.float64)
    ref = np.concatenate([O.seg_vlad(tk2[b], inc2[b], C2, adj2[b]) for b in range(4)])
    print(f"identity PCA {path:8s}: max abs err {np.abs(y - ref).max():.2e}  (|ref| max {np.abs(ref).max():.2f})")
