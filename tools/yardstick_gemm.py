"""Independent yardstick for the kNN filter's roofline fraction (measurement only -- never the product path): a PLAIN
fp16 GEMM of the filter's shape, 10 000 x 1 M x 1024 (queries x database rows x d), through torch.matmul (hipBLASLt /
rocBLAS), which WRITES its 10 000 x 1 M fp16 result (20 GB) -- the filter keeps ~800 of every query's 1 M distances.
Prints one JSON line {"yardstick_gemm_tflops": ..., "ms": ..., "shape": [...], "chunks": ...}; bench.py imports measure().
   python tools/yardstick_gemm.py [nq n d chunks reps]"""
import json
import sys

import torch


def measure(nq=10000, n=1000000, d=1024, chunks=4, reps=5):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    Q = torch.nn.functional.normalize(torch.randn(nq, d, device=dev, generator=g), dim=1).half()
    R = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1).half()
    step = (n + chunks - 1) // chunks
    outs = [torch.empty(nq, min(step, n - c * step), device=dev, dtype=torch.float16) for c in range(chunks)]

    def one():
        for c in range(chunks):
            torch.matmul(Q, R[c * step:(c + 1) * step].t(), out=outs[c])

    one()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    del outs, Q, R
    torch.cuda.empty_cache()
    return {"yardstick_gemm_tflops": round(2.0 * nq * n * d / ms / 1e9, 1), "ms": round(ms, 3), "shape": [nq, n, d], "chunks": chunks,
            "what": "torch.matmul fp16 (hipBLASLt / rocBLAS), fp16 result written to HBM"}


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[1:6]]
    print(json.dumps(measure(*args)))
