#!/usr/bin/env python3
"""Segment-VLAD stage probe: B images of the 17places geometry, K=64 -> per-stage HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from revisit_anything_amd import synth
from revisit_anything_amd.engine import SegVLADEngine

B = int(os.environ.get("B", 200)); K = int(os.environ.get("K", 64)); reps = int(os.environ.get("REPS", 5))
D, N, S = 1536, 1530, 50
eng = SegVLADEngine(0)
dev = eng.device
g = torch.Generator(device=dev); g.manual_seed(0)
C = synth.make_vocab(K, D, seed=1000)
eng.set_vocab(C)
z = torch.randint(0, K, (B, N), device=dev, generator=g)
Ct = torch.from_numpy(C).to(dev)
x = torch.empty(B, D, N, device=dev)
for b0 in range(0, B, 20):
    xb = Ct[z[b0:b0 + 20]] + 0.05 * torch.randn(z[b0:b0 + 20].shape[0], N, D, device=dev, generator=g)
    x[b0:b0 + 20] = torch.nn.functional.normalize(xb, dim=2).permute(0, 2, 1)
bits = torch.randint(-2**62, 2**62, (B * S, (N + 63) // 64), device=dev, generator=g, dtype=torch.int64) & \
       torch.randint(-2**62, 2**62, (B * S, (N + 63) // 64), device=dev, generator=g, dtype=torch.int64) & \
       torch.randint(-2**62, 2**62, (B * S, (N + 63) // 64), device=dev, generator=g, dtype=torch.int64)   # ~12 % coverage
offs = (np.arange(B + 1) * S).astype(np.int32)
adj = (torch.rand(B, S, S, device=dev, generator=g) < 0.15).to(torch.uint8)
adj = (adj | torch.eye(S, device=dev, dtype=torch.uint8)[None]).reshape(-1).contiguous()
out = torch.empty(B * S, K * D, device=dev)
eng.seg_vlad(x, bits, offs, adj, out=out); torch.cuda.synchronize()
eng.set_profiling(True); eng.profile_reset()
for _ in range(reps):
    eng.seg_vlad(x, bits, offs, adj, out=out)
torch.cuda.synchronize()
st = {s: round(eng.stage_ms(s)[0] / reps, 3) for s in ("assign", "prep", "aggregate")}
print(f"VLAD B={B} K={K}: {st}")
