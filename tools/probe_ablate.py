"""Timing ablations of the default batch filter kernel (development build: SEGVLAD_LIB_PATH=.../libsegvlad_hip_abl.so; WRONG
results by construction).  f16_cfg 94 = no epilogue, 95 = no epilogue and no DMA in the k-loop (MFMA + LDS fragment reads +
barriers only), default = the full kernel."""
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
eng.set_profiling(True)
for cfg in (-1, 94, 95, -1, 94, 95):
    eng.set_option("f16_cfg", cfg)
    eng.profile_reset()
    for _ in range(3):
        try:
            eng.search(Q, k)
        except Exception as e:   # (an ablated filter may starve the later stages: only the filter's time is of interest)
            print("search raised:", str(e)[:80])
    torch.cuda.synchronize()
    ms, nl = eng.stage_ms("knn_gemm")
    print(f"f16_cfg {cfg}: filter launches {ms / 3:.2f} ms per search ({nl // 3} launches) = {2.0 * nq * n * d / (ms / 3) / 1e9:.0f} TF algorithmic", flush=True)
