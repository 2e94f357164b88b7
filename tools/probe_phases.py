"""Phase timing of the default batch filter kernel (development build with the ablations: SEGVLAD_LIB_PATH must point at
lib/libsegvlad_hip_abl.so -- `python revisit-anything_amd/build.py --ablations`).  f16_cfg 91 / 92 = epilogue 0 / 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "abl" in os.environ.get("SEGVLAD_LIB_PATH", ""), "run with SEGVLAD_LIB_PATH=.../libsegvlad_hip_abl.so"
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
g = torch.Generator(device=dev)
g.manual_seed(1)
n, d, k, nq = 1_000_000, 1024, 200, 10_000
R = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
eng.db_add(R)
Q = torch.nn.functional.normalize(R[torch.arange(nq, device=dev) * 97] + 0.03 * torch.randn(nq, d, device=dev, generator=g), dim=1)
eng.search(Q, k)
for walk in (0, 3):
    eng.set_option("f16_walk", walk)
    for cfg in (91, 92, 93):
        eng.set_option("f16_cfg", cfg)
        print(f"--- f16_cfg {cfg} walk {walk}", file=sys.stderr, flush=True)
        eng.search(Q, k)
        torch.cuda.synchronize()
