"""A/B of the fp16 filter configurations inside ONE process (box-to-box and run-to-run spreads are +-3 %: decisions between
variants 2 % apart need interleaved measurements).   python tools/probe_cfg_ab.py 250 50 [...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

cfgs = [int(x) for x in sys.argv[1:]] or [250, 50]
dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
g = torch.Generator(device=dev)
g.manual_seed(1)
n, d, k, nq = 1_000_000, 1024, 200, 10_000
R = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
eng.db_add(R)
Q = torch.nn.functional.normalize(R[torch.arange(nq, device=dev) * 97] + 0.03 * torch.randn(nq, d, device=dev, generator=g), dim=1)
res = {c: [] for c in cfgs}
ref = None
for rnd in range(7):
    for c in cfgs:
        eng.set_option("f16_cfg", c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d2, idx = eng.search(Q, k)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = (d2.clone(), idx.clone())
        assert torch.equal(idx, ref[1]) and torch.equal(d2, ref[0])
        if rnd:
            res[c].append(e0.elapsed_time(e1))
for c in cfgs:
    v = sorted(res[c])
    print(f"f16_cfg {c}: median {v[len(v) // 2]:.2f} ms, min {v[0]:.2f}, max {v[-1]:.2f}   (search of {nq} x {n} x {d}, k = {k})")
