#!/usr/bin/env python3
"""Quick per-stage timing probe on the GPU box (HIP events inside the library)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from revisit_anything_amd import synth
from revisit_anything_amd.engine import SegVLADEngine

def main():
    eng = SegVLADEngine(0)
    eng.set_profiling(True)
    dev = eng.device
    g = torch.Generator(device=dev); g.manual_seed(0)
    for K in (32, 64):
        D, N, S, B = 1536, 1530, 50, 16
        C = synth.make_vocab(K, D, seed=1000)
        eng.set_vocab(C)
        z = torch.randint(0, K, (B, N), device=dev, generator=g)
        Ct = torch.from_numpy(C).to(dev)
        x = Ct[z] + 0.05 * torch.randn(B, N, D, device=dev, generator=g)
        x = torch.nn.functional.normalize(x, dim=2).permute(0, 2, 1).contiguous()      # [B,D,N]
        masks = torch.from_numpy(np.stack([synth.make_masks(S, 240, 320, seed=2000 + b) for b in range(B)]).reshape(B * S, 240, 320).astype(np.uint8)).to(dev)
        offs = (np.arange(B + 1) * S).astype(np.int32)
        adj = (torch.rand(B, S, S, device=dev, generator=g) < 0.25).to(torch.uint8)
        adj = (adj | torch.eye(S, device=dev, dtype=torch.uint8)[None]).reshape(-1).contiguous()
        out = torch.empty(B * S, K * D, device=dev)
        for it in range(3):
            bits = eng.incidence(masks, 480, 640)
            eng.seg_vlad(x, bits, offs, adj, out=out)
        torch.cuda.synchronize()
        t0 = time.time()
        for it in range(5):
            bits = eng.incidence(masks, 480, 640)
            eng.seg_vlad(x, bits, offs, adj, out=out)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        st = {s: eng.stage_ms(s)[0] for s in ("incidence", "assign", "prep", "aggregate")}
        alg = B * (4 * D * N + 4 * S * K * D + S * N / 8 + S * S + S * 240 * 320)
        print(f"VLAD K={K} B={B}: wall {dt*1e3:.3f} ms/batch ({B/dt:.0f} img/s), stages ms {st}, alg GB/s over kernels {alg/ (sum(st.values())*1e-3)/1e9:.0f}")
    # PCA
    K = 64; KD = K * 1536; P = 1024; n = 5000
    comps = torch.randn(P, KD, device=dev, generator=g) / KD ** 0.5
    mean = torch.randn(KD, device=dev, generator=g) * 1e-3
    var = torch.rand(P, device=dev, generator=g) * 1e-3 + 1e-6
    eng.pca_set(mean, comps, var, True)
    X = torch.randn(n, KD, device=dev, generator=g) / KD ** 0.5
    for it in range(2):
        Y = eng.pca_apply(X, l2norm=True)
    torch.cuda.synchronize()
    ms = eng.stage_ms("pca")[0]
    print(f"PCA {n}x{KD}->{P}: {ms:.2f} ms = {2*n*KD*P/ms/1e9:.1f} TFLOP/s")
    del comps, X
    # kNN
    for nr, nq, d in ((50000, 10000, 1024), (125000, 1600, 1024)):
        R = torch.nn.functional.normalize(torch.randn(nr, d, device=dev, generator=g), dim=1)
        Q = torch.nn.functional.normalize(torch.randn(nq, d, device=dev, generator=g), dim=1)
        eng.db_reset(); eng.db_add(R, torch.arange(nr, device=dev, dtype=torch.int32) // 50)
        for it in range(2):
            d2, idx = eng.search(Q, 200)
        torch.cuda.synchronize()
        t0 = time.time(); d2, idx = eng.search(Q, 200); torch.cuda.synchronize(); dt = time.time() - t0
        g_ms, s_ms = eng.stage_ms("knn_gemm")[0], eng.stage_ms("knn_select")[0]
        print(f"kNN nr={nr} nq={nq} d={d}: wall {dt*1e3:.2f} ms; last-chunk gemm {g_ms:.2f} ms select {s_ms:.2f} ms; overall {2*nq*nr*d/dt/1e12:.1f} TFLOP/s")
        sims, m50 = eng.sims_from_d2(d2, idx, 50)
        offs = (np.arange(nq // 50 + 1) * 50).astype(np.int32)
        for it in range(2):
            pred, sc = eng.vote(m50, sims, offs, n_top=5)
        torch.cuda.synchronize()
        print(f"   vote {nq//50} images: {eng.stage_ms('vote')[0]:.3f} ms")

if __name__ == "__main__":
    main()
