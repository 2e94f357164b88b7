#!/usr/bin/env python3
"""Probe: does the describe stage of a 200-image batch get faster as TWO half batches on two contexts / HIP streams?  Its kernels
alternate between HBM-bound passes (assignment, block norms + planes) and the matrix-pipe-bound grouped GEMM; side by side the
halves could fill each other's idle unit.  Prints ms per 200 images for the one-call form and for the split form, and whether the
rows are the same bits.  Not part of the product; run it on a GPU box:

    python tools/probe_describe_split.py [n_images=200] [reps=10]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revisit_anything_amd import synth
from revisit_anything_amd.engine import SegVLADEngine
from revisit_anything_amd.pipeline import SegVLADPipeline

NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 200
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S, K, D, H, W, P = 50, 64, 1536, 480, 640, 1024
N, Hm, Wm = (H // 14) * (W // 14), H // 2, W // 2
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def make_engine():
    eng = SegVLADEngine(0)
    eng.set_vocab(synth.make_vocab(K, D, seed=1000))
    g = torch.Generator(device=dev); g.manual_seed(5000)
    comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
    mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
    eng.pca_set(mean, comps, torch.logspace(-3, -6, P, device=dev), whiten=True)
    return eng


C = torch.from_numpy(synth.make_vocab(K, D, seed=1000)).to(dev)


def images(n, seed):
    gg = torch.Generator(device=dev); gg.manual_seed(seed)
    z = torch.randint(0, K, (n, N), device=dev, generator=gg)
    x = torch.nn.functional.normalize(C[z] + 0.05 * torch.randn(n, N, D, device=dev, generator=gg), dim=2).permute(0, 2, 1).contiguous()
    m = (torch.rand(n * S, Hm, Wm, device=dev, generator=gg) < 0.02).to(torch.uint8)
    m[:, Hm // 2, Wm // 2] = 1
    return x, m


e0, e1, e2 = make_engine(), make_engine(), make_engine()
p0, p1, p2 = (SegVLADPipeline(e, H, W, 14, order=3, use_pca=True) for e in (e0, e1, e2))
x, m = images(NQ, 7)
h = NQ // 2
offs = (np.arange(NQ + 1) * S).astype(np.int32)
offs_a = (np.arange(h + 1) * S).astype(np.int32)
offs_b = (np.arange(NQ - h + 1) * S).astype(np.int32)
xa, xb = x[:h].contiguous(), x[h:].contiguous()
ma, mb = m[:h * S].contiguous(), m[h * S:].contiguous()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def whole():
    return p0.describe(x, m, offs)


def split():
    cur = torch.cuda.current_stream(dev)
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = p1.describe(xa, ma, offs_a)
    with torch.cuda.stream(s2):
        b = p2.describe(xb, mb, offs_b)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3, out


tw, ow = timed(whole)
ts, (oa, ob) = timed(split)
same = bool(torch.equal(ow[:h * S], oa) and torch.equal(ow[h * S:], ob))
diff = float(torch.max(torch.abs(ow - torch.cat([oa, ob]))))
print(f"[describe split] {NQ} images: one call {tw:.3f} ms, two half batches on two streams {ts:.3f} ms; rows bit-identical {same}, max |diff| {diff:.2e}")
