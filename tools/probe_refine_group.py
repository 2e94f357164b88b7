"""Stage times of the exact refinement, grouped (csrc/refine_group_kernels.hip) against per row, on planted data:
`places` groups of `per` near-duplicate rows, a query image = `seg` noisy copies of one place's rows.
  python tools/probe_refine_group.py [d] [places] [per] [k]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
places = int(sys.argv[2]) if len(sys.argv) > 2 else 250
per = int(sys.argv[3]) if len(sys.argv) > 3 else 200
k = int(sys.argv[4]) if len(sys.argv) > 4 else 200
seg, n_img = 50, 200
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(7)
R = torch.empty(places * per, d, device=dev)
for p in range(places):
    base = torch.randn(1, d, device=dev, generator=g)
    R[p * per:(p + 1) * per] = base + 0.7 * torch.randn(per, d, device=dev, generator=g)
R = torch.nn.functional.normalize(R, dim=1)
pick = torch.randint(0, places, (n_img,), device=dev, generator=g)
rows = (pick[:, None] * per + torch.arange(seg, device=dev)[None, :]).reshape(-1)
Q = torch.nn.functional.normalize(R[rows] + 0.3 * torch.randn(rows.numel(), d, device=dev, generator=g) / d ** 0.5, dim=1)
eng = SegVLADEngine(0)
eng.db_add(R)
eng.set_option("query_group", seg)
ref = None
for mode in (0, 1):
    eng.set_option("refine_group", mode)
    eng.set_option("search_stats", 1)
    out = eng.search(Q, k)
    st = eng.search_stats()
    eng.set_option("search_stats", 0)
    eng.search(Q, k)
    eng.set_profiling(True)
    eng.profile_reset()
    for _ in range(2):
        out = eng.search(Q, k)
    torch.cuda.synchronize()
    ms = {s: eng.stage_ms(s)[0] / 2 for s in ("knn_level0", "knn_gemm", "knn_select")}
    eng.set_profiling(False)
    same = None if ref is None else bool(torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1]))
    if ref is None:
        ref = out
    print(f"[refine probe] d={d} rows={places * per} k={k} refine_group={mode}: "
          f"{ {a: round(b, 3) for a, b in ms.items()} } refine_sum {st['refine_sum']} groups {st['grp_groups']} "
          f"union_sum {st['grp_union_sum']} identical_to_per_row {same}", flush=True)
