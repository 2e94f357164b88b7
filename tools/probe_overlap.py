#!/usr/bin/env python3
"""Probe for a pipelined throughput mode: does describing batch i+1 (segment-VLAD + PCA: HBM- and latency-bound)
overlap with searching batch i (kNN filter: matrix-pipe bound) when the two run on their own contexts, HIP streams and
host threads?  Prints steps/s of the sequential schedule and of the two-stage pipeline on the bench workload
(smaller database by default: NR images).  Not part of the product; run it on a GPU box:

    NR=20000 NQ=200 STEPS=8 python tools/probe_overlap.py
"""
import os
import queue
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revisit_anything_amd import synth
from revisit_anything_amd.engine import SegVLADEngine
from revisit_anything_amd.pipeline import SegVLADPipeline

NR = int(os.environ.get("NR", 4000)); NQ = int(os.environ.get("NQ", 200)); STEPS = int(os.environ.get("STEPS", 8))
S, K, D, H, W, P = 50, 64, 1536, 480, 640, 1024
N, Hm, Wm = (H // 14) * (W // 14), H // 2, W // 2
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def make_engine():
    eng = SegVLADEngine(0)
    eng.set_vocab(synth.make_vocab(K, D, seed=1000))
    g = torch.Generator(device=dev); g.manual_seed(5000)
    comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
    mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
    eng.pca_set(mean, comps, torch.logspace(-3, -6, P, device=dev), whiten=True)
    return eng


eng_d, eng_s = make_engine(), make_engine()          # describe context, search context (owns the database)
pipe = SegVLADPipeline(eng_d, H, W, 14, order=3, use_pca=True)
g = torch.Generator(device=dev); g.manual_seed(1)
C = torch.from_numpy(synth.make_vocab(K, D, seed=1000)).to(dev)


def images(n, seed):
    gg = torch.Generator(device=dev); gg.manual_seed(seed)
    z = torch.randint(0, K, (n, N), device=dev, generator=gg)
    x = torch.nn.functional.normalize(C[z] + 0.05 * torch.randn(n, N, D, device=dev, generator=gg), dim=2).permute(0, 2, 1).contiguous()
    m = (torch.rand(n * S, Hm, Wm, device=dev, generator=gg) < 0.02).to(torch.uint8)
    m[:, Hm // 2, Wm // 2] = 1
    return x, m


offs = (np.arange(NQ + 1) * S).astype(np.int32)
rows = []
for b0 in range(0, NR, 100):
    x, m = images(min(100, NR - b0), 100 + b0)
    rows.append(pipe.describe(x, m, (np.arange(x.shape[0] + 1) * S).astype(np.int32)))
rows = torch.cat(rows)
eng_s.db_add(rows, torch.arange(NR, device=dev, dtype=torch.int32).repeat_interleave(S))
q_tok, q_msk = images(NQ, 7)
pipe_s = SegVLADPipeline(eng_s, H, W, 14, order=3, use_pca=True)


def describe():
    return pipe.describe(q_tok, q_msk, offs)


def retrieve(qd):
    return pipe_s.retrieve(qd, offs, 200, 50, 5)


for _ in range(2):
    retrieve(describe())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    retrieve(describe())
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / STEPS

s_d, s_s = torch.cuda.Stream(), torch.cuda.Stream()
q = queue.Queue(maxsize=2)


def producer():
    with torch.cuda.stream(s_d):
        for _ in range(STEPS):
            qd = describe()
            ev = torch.cuda.Event()
            ev.record(s_d)
            q.put((qd, ev))
    q.put(None)


def consumer():
    with torch.cuda.stream(s_s):
        while True:
            item = q.get()
            if item is None:
                break
            qd, ev = item
            s_s.wait_event(ev)
            qd.record_stream(s_s)
            retrieve(qd)


torch.cuda.synchronize()
t0 = time.perf_counter()
tp, tc = threading.Thread(target=producer), threading.Thread(target=consumer)
tp.start(); tc.start(); tp.join(); tc.join()
torch.cuda.synchronize()
pipe_t = (time.perf_counter() - t0) / STEPS
print(f"db {NR * S} rows, {NQ} query images/step: sequential {seq * 1e3:.2f} ms/step ({NQ / seq:.0f} img/s), "
      f"two-stage pipeline {pipe_t * 1e3:.2f} ms/step ({NQ / pipe_t:.0f} img/s)")
