"""How many DISTINCT database rows do the 50 segments of one query image send to the exact refinement?

The refinement re-evaluates, per query ROW, the band {d~2 <= A_k + 2 eps} (>= k rows) with the fp32 chain: 10 000 x 210 row
gathers per batch on raw 98 304-d descriptors (827 GB).  If the segments of an image share their neighbours, the union per
IMAGE is what has to be read.  Prints, for the bench's config2 workload (raw K*D, 1000 reference images) and for a PCA'd
index, the distribution of |union of the top-k ids over an image's segments| for k = 200 and 50.

  python tools/probe_union.py [db_images_pca]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from revisit_anything_amd import synth  # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402
from revisit_anything_amd.pipeline import SegVLADPipeline  # noqa: E402


def one(no_pca: bool, n_ref: int, n_q: int = 200):
    dev = torch.device("cuda:0")
    S, K, D, H, W = 50, 64, 1536, 480, 640
    N = (H // 14) * (W // 14)
    P = K * D if no_pca else 1024
    eng = SegVLADEngine(0)
    C_np = synth.make_vocab(K, D, seed=1000)
    eng.set_vocab(C_np)
    C = torch.from_numpy(C_np).to(dev)
    if not no_pca:
        g = torch.Generator(device=dev)
        g.manual_seed(5000)
        comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
        mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
        eng.pca_set(mean, comps, torch.logspace(-3, -6, P, device=dev), whiten=True)
        del comps
        eng.set_option("pca_path", "project")
    pipe = SegVLADPipeline(eng, H, W, 14, order=3, use_pca=not no_pca)
    fac = bench.ImageFactory(dev, C, N, S, H // 2, W // 2, bench.QUERY_OWN_DEFAULT, 4)
    rows = torch.empty(n_ref * S, P, device=dev)
    bb = 100
    tok = torch.empty(bb, D, N, device=dev)
    msk = torch.empty(bb * S, H // 2, W // 2, dtype=torch.uint8, device=dev)
    for b0 in range(0, n_ref, bb):
        nb = min(bb, n_ref - b0)
        for j in range(nb):
            t, m = fac.reference(b0 + j)
            tok[j] = t
            msk[j * S:(j + 1) * S] = m
        offs = (np.arange(nb + 1) * S).astype(np.int32)
        rows[b0 * S:(b0 + nb) * S] = pipe.describe(tok[:nb], msk[:nb * S], offs)
    eng.db_add(rows, torch.arange(n_ref, device=dev, dtype=torch.int32).repeat_interleave(S))
    tau = np.random.Generator(np.random.PCG64(4000)).integers(0, n_ref, size=n_q)
    for j in range(0, n_q, bb):
        nb = min(bb, n_q - j)
        for i in range(nb):
            t, m = fac.query(int(tau[j + i]), j + i)
            tok[i] = t
            msk[i * S:(i + 1) * S] = m
        offs = (np.arange(nb + 1) * S).astype(np.int32)
        qd = pipe.describe(tok[:nb], msk[:nb * S], offs)
        d2, idx = eng.search(qd, 200)
        idx = idx.cpu().numpy().reshape(nb, S, 200)
        for kk in (200, 50):
            u = np.array([len(np.unique(idx[i, :, :kk])) for i in range(nb)])
            own = np.array([np.mean(idx[i, :, :kk] // S // 4 == tau[j + i] // 4) for i in range(nb)])
            print(f"[union] no_pca={no_pca} n_ref={n_ref} images {j}..{j + nb}: top-{kk} union per image "
                  f"min {u.min()} median {int(np.median(u))} mean {u.mean():.0f} max {u.max()} (of {S * kk} slots); "
                  f"share of slots in the query's own sibling group {own.mean():.3f}", flush=True)
    eng.close()


if __name__ == "__main__":
    one(True, 1000)
    one(False, int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
