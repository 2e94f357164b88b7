"""One 50-segment query image per pass over a 1 M x 1024 index: per-stage times (HIP events) and -- under
rocprofv3 --kernel-trace --stats -- the per-kernel durations of the pass.   python tools/probe_stream.py [reps] [key=value ...]
(key=value: segvlad_set_option switches, e.g. small_plan=0 for the deep level plan of the batches)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
eng = SegVLADEngine(0)
opts = [kv.split("=", 1) for kv in sys.argv[2:]]
g = torch.Generator(device=dev)
g.manual_seed(1)
n, d, k = 1_000_000, 1024, 200
R = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
eng.db_add(R)
Q = torch.nn.functional.normalize(R[torch.arange(50, device=dev) * 977] + 0.03 * torch.randn(50, d, device=dev, generator=g), dim=1)
ref = eng.search(Q, k)          # default options: the reference result of this run
for key, val in opts:
    eng.set_option(key, val)
for _ in range(5):
    got = eng.search(Q, k)
torch.cuda.synchronize()
if opts:
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), "options changed the result"
    print("options", opts, ": result bit-identical to the default's")
eng.set_profiling(True)
eng.profile_reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng.search(Q, k)
e1.record()
torch.cuda.synchronize()
tot = 0.0
for s in ("knn_level0", "knn_gemm", "knn_select"):
    ms, nl = eng.stage_ms(s)
    tot += ms / reps
    print(f"{s}: {ms / reps * 1e3:.1f} us per pass, {nl / reps:.1f} launches")
print(f"stages sum {tot * 1e3:.1f} us; wall per search {e0.elapsed_time(e1) / reps * 1e3:.1f} us; streamed 2.05 GB -> "
      f"{2.048e9 / (tot * 1e-3) / 1e12:.2f} TB/s of the stage sum = {2.048e9 / (tot * 1e-3) / 8e12:.3f} of HBM peak")
print(eng.search_stats())
# the call as a caller sees it: stage timers off, `reps` calls back to back, ONE synchronisation at the end (bench.py's call_ms_wall)
import time  # noqa: E402

eng.set_profiling(False)
for _ in range(5):
    eng.search(Q, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.search(Q, k)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
print(f"call_ms_wall {wall * 1e3:.4f} ms -> {2.048e9 / wall / 8e12:.3f} of HBM peak by the wall clock of the call")
