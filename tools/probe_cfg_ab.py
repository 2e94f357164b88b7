"""A/B of fp16 filter variants inside ONE process (box-to-box and run-to-run spreads are +-3 %: decisions between variants
2 % apart need interleaved measurements).  Every variant's result is asserted bit-identical to the first one's.
   python tools/probe_cfg_ab.py 250 50                      (f16_cfg values)
   python tools/probe_cfg_ab.py f16_walk=0,f16_epi=0 f16_walk=3,f16_epi=1 ...   (option sets; unnamed options keep their defaults)
Environment: NR / NQ / D / K (default 1 000 000 / 10 000 / 1024 / 200), ROUNDS (7), DATA=planted|random."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

DEFAULTS = {"f16_cfg": -1, "f16_gm": -1, "f16_walk": -1, "f16_epi": -1, "f16_mf": -1, "f16_deep_cfg": -1, "f16_pp": -1, "f16_small_mf": 0, "f16_buf": -1, "f16_dsplit": 0}


def parse(spec):
    if "=" not in spec:
        return {"f16_cfg": int(spec)}
    return {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",")}


specs = sys.argv[1:] or ["250", "50"]
variants = [parse(s) for s in specs]
dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
g = torch.Generator(device=dev)
g.manual_seed(1)
n, d, k, nq = (int(os.environ.get(a, b)) for a, b in (("NR", 1_000_000), ("D", 1024), ("K", 200), ("NQ", 10_000)))
R = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
eng.db_add(R)
if os.environ.get("DATA", "planted") == "planted":
    Q = torch.nn.functional.normalize(R[(torch.arange(nq, device=dev) * 97) % n] + 0.03 * torch.randn(nq, d, device=dev, generator=g), dim=1)
else:
    Q = torch.nn.functional.normalize(torch.randn(nq, d, device=dev, generator=g), dim=1)
res = {s: [] for s in specs}
gem = {s: [] for s in specs}
ref = None
eng.set_profiling(True)
for rnd in range(int(os.environ.get("ROUNDS", 7))):
    for s, v in zip(specs, variants):
        for key, val in {**DEFAULTS, **v}.items():
            eng.set_option(key, val)
        eng.profile_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d2, idx = eng.search(Q, k)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = (d2.clone(), idx.clone())
        assert torch.equal(idx, ref[1]) and torch.equal(d2, ref[0]), f"variant {s} changed the result"
        if rnd:
            res[s].append(e0.elapsed_time(e1))
            gem[s].append(eng.stage_ms("knn_gemm")[0])
for s in specs:
    v, gv = sorted(res[s]), sorted(gem[s])
    print(f"{s}: search median {v[len(v) // 2]:.2f} ms (min {v[0]:.2f}, max {v[-1]:.2f}); filter launches median {gv[len(gv) // 2]:.2f} ms "
          f"= {2.0 * nq * n * d / gv[len(gv) // 2] / 1e9:.0f} TF algorithmic   ({nq} x {n} x {d}, k = {k})", flush=True)
