#!/usr/bin/env python3
"""BASELINE configs[2] ("17places end-to-end: SAM ViT-H + DINOv2 -> HIP VLAD/kNN on 1 x MI355X"), THROUGHPUT ONLY: the
dataset and both backbones' weights are not in this image (no network), so the backbones are random-initialised networks
of the published geometry (transformers' Dinov2Model / SamModel on PyTorch-ROCm) and the images are synthetic.  What is
measured is the cost split of the whole chain, image -> tokens + masks -> segment descriptors -> index / retrieval:

    python tools/e2e_throughput.py [--dino large|giant|small] [--sam huge|base] [--ref 32] [--query 8]

Recall is meaningless with random weights; the HIP part of the chain is parity-tested separately on the same kind of
producer output (tests/test_gpu_e2e_producers.py)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisit_anything_amd import producers as pr, synth  # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402
from revisit_anything_amd.pipeline import SegVLADPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dino", default="large")
    ap.add_argument("--dino-layer", type=int, default=None)
    ap.add_argument("--sam", default="huge")
    ap.add_argument("--ref", type=int, default=32)
    ap.add_argument("--query", type=int, default=8)
    ap.add_argument("--segments", type=int, default=50, help="keep the S best-scored SAM masks per image (a random network proposes hundreds)")
    ap.add_argument("--clusters", type=int, default=64)
    ap.add_argument("--pca-dim", type=int, default=1024)
    ap.add_argument("--points-per-side", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W = 480, 640
    cfg = {"rmin": 0, "desired_width": W, "desired_height": H, "resize": True}
    n_layers = {"small": 12, "base": 12, "large": 24, "giant": 40}[a.dino]
    layer = a.dino_layer if a.dino_layer is not None else {"giant": 31}.get(a.dino, n_layers - 1)
    torch.manual_seed(0)
    dino = pr.DinoV2ValueFacet.from_config(a.dino, layer=layer, device=dev)
    # every filter off (a random network's predicted IoUs / stability scores mean nothing, and its masks overlap so much
    # that box NMS at 0.7 left ONE segment per image in round 2): the S best-scored proposals are kept, so that the HIP
    # describe stage is timed at S = 50 segments per image like the reference's data
    sam = pr.SamAutoMasks.from_config(a.sam, device=dev, points_per_side=a.points_per_side, pred_iou_thresh=-1.0,
                                      stability_score_thresh=0.0, box_nms_thresh=1.01)
    with torch.no_grad():   # give the random SAM non-degenerate outputs (its default initialiser is ~0)
        for p in sam.model.parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 0.02)
    D = dino.model.config.hidden_size
    K, S, P = a.clusters, a.segments, a.pca_dim
    eng = SegVLADEngine(0)
    eng.set_vocab(synth.make_vocab(K, D, seed=1000))
    g = torch.Generator(device=dev)
    g.manual_seed(5000)
    eng.pca_set(torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5),
                torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5, torch.logspace(-3, -6, P, device=dev), whiten=True)
    pipe = SegVLADPipeline(eng, H, W, 14, order=3, use_pca=True)
    pipe0 = SegVLADPipeline(eng, H, W, 14, order=0, use_pca=True)
    rng = np.random.Generator(np.random.PCG64(1))
    t = {"dino": 0.0, "sam": 0.0, "describe": 0.0, "retrieve": 0.0}

    def produce(img_bgr):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tok = pr.image_to_tokens(np.ascontiguousarray(img_bgr[:, :, ::-1]), dino, cfg)           # [1, D, 34, 45]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        segs, recs = pr.masks_given_image(sam, img_bgr, cfg)                                       # 240 x 320 masks, best first
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t["dino"] += t1 - t0; t["sam"] += t2 - t1
        m = np.stack(segs[:S]).astype(np.uint8) if segs else np.zeros((0, H // 2, W // 2), np.uint8)
        return tok.reshape(D, -1), m

    def describe(imgs):
        toks, masks, offs = [], [], [0]
        for im in imgs:
            tk, m = produce(im)
            toks.append(tk), masks.append(torch.from_numpy(m).to(dev)), offs.append(offs[-1] + m.shape[0])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            d = pipe.describe(torch.stack(toks).contiguous(), torch.cat(masks).contiguous(), np.asarray(offs, np.int32))
        except Exception as e:   # a random network's masks can coincide: Qhull refuses duplicate / collinear centroid sets
            print(f"[e2e] neighbourhood aggregation skipped for this batch ({type(e).__name__}): order 0", file=sys.stderr)
            d = pipe0.describe(torch.stack(toks).contiguous(), torch.cat(masks).contiguous(), np.asarray(offs, np.int32))
        torch.cuda.synchronize(); t["describe"] += time.perf_counter() - t0
        return d, np.asarray(offs, np.int32)

    refs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(a.ref)]
    produce(refs[0])                                   # warm-up (MIOpen / hipBLASLt plans)
    for k in t:
        t[k] = 0.0
    t_all = time.perf_counter()
    rd, roffs = describe(refs)
    eng.db_reset()
    eng.db_add(rd, np.repeat(np.arange(a.ref, dtype=np.int32), np.diff(roffs)))
    qs = [np.clip(refs[i % a.ref].astype(np.int16) + rng.integers(-8, 9, (H, W, 3)), 0, 255).astype(np.uint8) for i in range(a.query)]
    qd, qoffs = describe(qs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pred, _, _, _ = pipe.retrieve(qd, qoffs, k_search=min(200, int(roffs[-1])), k_vote=min(50, int(roffs[-1])), n_top=5)
    torch.cuda.synchronize(); t["retrieve"] = time.perf_counter() - t0
    wall = time.perf_counter() - t_all
    n = a.ref + a.query
    print(f"e2e (random-init DINOv2-{a.dino} layer {layer} D={D}, SAM-{a.sam}, {a.points_per_side}^2 points, S<={S}, K={K}, PCA {P}): "
          f"{n} images in {wall:.2f} s = {n / wall:.2f} images/s; per image: DINO {t['dino'] / n * 1e3:.1f} ms, SAM {t['sam'] / n * 1e3:.1f} ms, "
          f"HIP describe {t['describe'] / n * 1e3:.2f} ms; retrieve {a.query} queries {t['retrieve'] * 1e3:.2f} ms; "
          f"segments/image {float(np.diff(np.concatenate([roffs, qoffs[1:] + roffs[-1]])).mean()):.1f}")


if __name__ == "__main__":
    main()
