#!/usr/bin/env python3
"""bench.py -- SegVLAD retrieval hot path on MI355X (see DESIGN.md section "Measurement").

One STEP = one pass of the hot path over one batch of synthetic query images whose inputs (DINO
token blocks [D][N] fp32 and SAM masks [S][Hm][Wm] u8) are already resident in HBM:

    masks -> incidence -> centroids -> (host Qhull) adjacency^order -> segment-VLAD (K clusters)
          -> PCA-whiten + L2  -> exact kNN (search 200) against the row-sharded segment DB
          -> [N>1: all_gather of per-shard top-k + merge] -> keep 50, 2-d^2 -> weighted image vote

Workload (BASELINE.json north_star / configs[3] shape, fits one GPU once PCA'd): 200 query images
x 50 segments, 640x480 -> 34x45 = 1530 tokens, D=1536, K=64, PCA 1024, DB = 20 000 reference images
x 50 = 1 M segments.  The DB is built (untimed) by the same VLAD+PCA kernels from synthetic reference
images; queries are perturbed copies of reference images so Recall@1 is meaningful.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  Strong scaling: the 1 M-segment DB and the 200-image query batch
are fixed; rank r holds DB rows of reference images [r*n/N, (r+1)*n/N) and describes query images
[r*200/N, (r+1)*200/N) before an all_gather of the 1024-d query descriptors.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from revisit_anything_amd import synth  # noqa: E402
from revisit_anything_amd.engine import SegVLADEngine  # noqa: E402
from revisit_anything_amd.pipeline import SegVLADPipeline, recall_at  # noqa: E402
from revisit_anything_amd.sharded import ShardedSegmentIndex, shard_images  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0
# Difficulty of the synthetic retrieval task (SURVEY 8d asks for an oracle Recall@1 of about 0.8, so that the vote
# matters and near-duplicate sibling places are real distractors).  Calibrated on the device with --sweep-own.
QUERY_OWN_DEFAULT = 0.12   # device sweep (r02): 0.0 -> 0.20, 0.1 -> 0.73, 0.2 -> 0.965, >= 0.3 -> 1.0 Recall@1


FILTER_KIND = "f16"   # set from SegVLADEngine.search_stats() after the first search


def eng_filter_products() -> int:
    """MFMA products the kNN filter spends per algorithmic fp32 multiply-add (fp16 = 1, bf16x3 split = 3, fp32 = 1)."""
    return 3 if FILTER_KIND == "bf16x3" else 1


def eng_pca_products(kd: int, p_dim: int) -> int:
    """MFMA products of the PCA projection per algorithmic fp32 multiply-add: the default is three fp16 products of a
    two-term split (hi.hi + hi.lo + lo.hi); SEGVLAD_PCA_FP32=1 (read at context creation) selects the fp32 MFMA GEMM."""
    return 1 if (os.environ.get("SEGVLAD_PCA_FP32") or kd % 64 or p_dim % 64) else 3


def kernel_src_sha() -> str:
    """sha256 (first 16 hex) over the kernel sources: a PMC summary is only quoted if it was collected from THESE kernels."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "revisit-anything_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_counters(kernel_prefix: str, workload_key: str):
    """Counter evidence of a kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json, produced by
    tools/pmc_summary.py from separate --pmc passes over `bench.py --pmc-calibrate` itself since round 6 -- the dispatches of
    the TIMED steps, cut out between two marker dispatches; rounds 1-5: over a replay of the kernel shapes):
    (HBM bytes per launch, MFMA utilisation, source note).  A summary is quoted ONLY if it records the sha of
    the kernel sources it was collected from and that sha equals the current sources'; otherwise (None, None, why)."""
    import glob
    sha = kernel_src_sha()
    why = "no PMC summary committed for this workload"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("workload_key") != workload_key:
            continue
        if j.get("kernel_src_sha") != sha:
            why = f"{os.path.basename(f)} was collected from kernel sources {j.get('kernel_src_sha')}, current {sha}: not quoted"
            continue
        for k in j.get("kernels", []):
            if k["name"].startswith(kernel_prefix) or ("_Z" in k["name"][:2] and kernel_prefix in k["name"]):   # (some names stay mangled)
                return k.get("hbm_bytes_per_launch"), k.get("mfma_util"), os.path.basename(f) + f" (kernel sources {sha})"
    return None, None, why


def pmc_field(kernel_prefix: str, workload_key: str, field: str):
    """Another field of the same (sha-gated) PMC summary record, e.g. effective_clock_ghz = GRBM_GUI_ACTIVE / duration."""
    import glob
    sha = kernel_src_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("workload_key") != workload_key or j.get("kernel_src_sha") != sha:
            continue
        for k in j.get("kernels", []):
            if k["name"].startswith(kernel_prefix) or ("_Z" in k["name"][:2] and kernel_prefix in k["name"]):
                return k.get(field)
    return None


def mfma_ubench():
    """tools/ubench/mfma_peak (built by __graft_entry__.build()): sustained fp16 MFMA rate of this box, live."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=120)
    except Exception:
        return None
    out = {}
    for line in r.stdout.splitlines():
        m = re.match(r"\s*(.*?)\s+blocks=(\d+) iters=\d+: ([0-9.]+) ms -> ([0-9.]+) TFLOP/s", line)
        if not m or m.group(2) != "256":
            continue
        name, tf = m.group(1), float(m.group(4))
        if name.startswith("mfma only (smooth"):
            out["mfma_only_smooth_tflops"] = tf
        elif name.startswith("mfma only (random"):
            out["mfma_only_random_tflops"] = tf
        elif name.startswith("mfma 16x16x32 only (random"):
            out["mfma16_only_random_tflops"] = tf
        elif name.startswith("mfma + 6 ds_read"):
            out["mfma_plus_lds_reads_tflops"] = tf
        elif name.startswith("+ 8 DMA pieces / 32"):
            out["mfma_lds_barrier_dma_l2_tflops"] = tf
        elif name.startswith("+ 8 DMA pieces, streaming"):
            out["mfma_lds_barrier_dma_hbm_tflops"] = tf
    out["note"] = ("v_mfma_f32_32x32x16_f16, 256 workgroups x 8 waves, registers only / + 6 ds_read_b128 per 8 MFMA / + barrier + "
                   "global->LDS DMA; random operands (smooth operands draw less power and clock higher); mfma16_*: the same flops "
                   "through v_mfma_f32_16x16x32_f16, the shape the batch filter kernel uses since round 4 (less energy per flop)")
    return out or None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--query-images", type=int, default=200)
    p.add_argument("--db-images", type=int, default=20000)
    p.add_argument("--segments", type=int, default=50)
    p.add_argument("--clusters", type=int, default=64)
    p.add_argument("--dim", type=int, default=1536)
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--pca-dim", type=int, default=1024)
    p.add_argument("--order", type=int, default=3)
    p.add_argument("--no-pca", action="store_true", help="raw K*D descriptors (only with a small --db-images)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--build-batch", type=int, default=100)
    p.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI); 'gloo' + "
                   "--same-device lets several ranks share ONE GPU to exercise the N>1 code path on a 1-GPU box")
    p.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0")
    p.add_argument("--dump-preds", default=None, help="debug: rank 0 writes the last step's predictions to this .npy")
    p.add_argument("--pmc-calibrate", action="store_true", help="map a 1 GiB tensor through torch.sign (a known 1 GiB read + 1 GiB write) "
                   "right before and right after the timed steps: tools/pmc_summary.py calibrates FETCH_SIZE / WRITE_SIZE on them and "
                   "takes the dispatches BETWEEN them as the timed path's (rocprofv3 --pmc passes over this very process)")
    p.add_argument("--query-own", type=float, default=QUERY_OWN_DEFAULT, help="difficulty: correlation of a query image's private token "
                   "noise with its reference image's (1 = the reference image itself, 0 = indistinguishable from its 3 sibling images)")
    p.add_argument("--sweep-own", default=None, help="debug: comma-separated --query-own values; after the DB build print the device "
                   "Recall@1 for each (stderr) and exit")
    p.add_argument("--verify-images", type=int, default=20, help="query images re-computed by the CPU oracle (fp64) and compared "
                   "with the device's predictions (N=1 only; part of the cpu_baseline leg)")
    p.add_argument("--debug-timing", action="store_true", help="after the timed region, print a synchronised per-phase wall-clock breakdown of one step to stderr")
    p.add_argument("--group", type=int, default=4, help="reference images per 'same place' sibling group (4 = the headline workload; 31 = a "
                   "17places-like video sequence, gt.py:60-64: every frame has its +-15 neighbours as near-duplicates)")
    p.add_argument("--pipeline", action="store_true", help="describe batch i+1 (its own context and stream) under the search of batch i")
    p.add_argument("--vote-depth", action="store_true", help="search k_vote = 50 deep instead of the reference's 200 (same votes; the "
                   "'vote_depth' sub-records of the line: NOT the headline, whose step searches 200 like place_rec_main.py:56)")
    p.add_argument("--no-sub-records", action="store_true", help="skip the 'config2' (raw K*D search, BASELINE configs[1]) and "
                   "'redundant_db' (sibling groups of 31) sub-records of the N=1 line")
    p.add_argument("--search-stats", action="store_true", help="record candidate / refine list occupancies (one extra read-back per search)")
    p.add_argument("--native-comm", action="store_true", help="N>1: exchange through the C-ABI's own RCCL communicator "
                   "(segvlad_search_sharded / segvlad_allgather_rows) instead of torch.distributed collectives")
    p.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="segvlad_set_option switches of the search context "
                   "(tuning A/B: e.g. --set batch_plan=1); recorded in the line")
    p.add_argument("--shard-sim-only", action="store_true", help="of the sub-records, only `shard_sim` (tuning sweeps of the per-rank step)")
    p.add_argument("--no-ubench", action="store_true", help="skip the live MFMA ceiling micro-benchmark (tools/ubench/mfma_peak) and the "
                   "same-shape plain-GEMM yardstick (tools/yardstick_gemm.py)")
    p.add_argument("--dry-run-collectives", action="store_true", help="no benchmark: the multi-GPU exchange checked WITHOUT the other ranks "
                   "(sharded.dry_run_collectives: operand shapes / dtypes / contiguity of both collectives and the unpack + merge for world "
                   "sizes 2..8 against a single index, the real collectives + the C-ABI communicator at world size 1 on the nccl backend); "
                   "prints a JSON report and exits non-zero on a violation")
    p.add_argument("--shard-sim", type=int, default=8, help="N=1 line: emulate ONE rank of a row-sharded run over this many GPUs on the one "
                   "GPU (a 1/W shard of the database, ALL query segments searched k_vote deep, 1/W of the query images described, the "
                   "W-way merge and the vote; no collective) -> the 'shard_sim' sub-record; 0 = skip")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------
# synthetic images, generated on the GPU from per-batch seeds (torch) so that 20k reference images
# never touch the host.  Reference images come in groups of 4 "same place" siblings.
# ------------------------------------------------------------------------------------------------
class ImageFactory:
    def __init__(self, dev, C: torch.Tensor, N: int, S: int, Hm: int, Wm: int, query_own: float = 0.7, group_size: int = 4):
        self.dev, self.C, self.N, self.S, self.Hm, self.Wm = dev, C, N, S, Hm, Wm
        self.query_own = float(query_own)
        self.G = int(group_size)
        self.K, self.D = C.shape
        self.yy = torch.arange(Hm, device=dev).view(1, Hm, 1)
        self.xx = torch.arange(Wm, device=dev).view(1, 1, Wm)

    def _gen(self, seed):
        g = torch.Generator(device=self.dev)
        g.manual_seed(int(seed))
        return g

    def group(self, gid: int):
        """Shared content of a sibling group: cluster ids, base noise, masks."""
        g = self._gen(10_000_019 + gid)
        z = torch.randint(0, self.K, (self.N,), device=self.dev, generator=g)
        base = torch.randn(self.N, self.D, device=self.dev, generator=g)
        S, Hm, Wm = self.S, self.Hm, self.Wm
        h = torch.randint(8, 61, (S,), device=self.dev, generator=g)
        w = torch.randint(8, 81, (S,), device=self.dev, generator=g)
        y0 = (torch.rand(S, device=self.dev, generator=g) * (Hm - h + 1)).long()
        x0 = (torch.rand(S, device=self.dev, generator=g) * (Wm - w + 1)).long()
        m = (self.yy >= y0.view(S, 1, 1)) & (self.yy < (y0 + h).view(S, 1, 1)) & (self.xx >= x0.view(S, 1, 1)) & (self.xx < (x0 + w).view(S, 1, 1))
        return z, base, m.to(torch.uint8)

    def own_noise(self, img_id: int):
        return torch.randn(self.N, self.D, device=self.dev, generator=self._gen(20_000_003 + img_id))

    def tokens(self, z, base, own):
        x = self.C[z] + 0.05 * (0.8 * base + 0.6 * own)
        return torch.nn.functional.normalize(x, dim=1).t().contiguous()      # [D,N] as the reference stores it

    def reference(self, img_id: int):
        z, base, m = self.group(img_id // self.G)
        return self.tokens(z, base, self.own_noise(img_id)), m

    def query(self, tau: int, qid: int):
        z, base, m = self.group(tau // self.G)
        a = self.query_own
        own = a * self.own_noise(tau) + (1.0 - a * a) ** 0.5 * torch.randn(self.N, self.D, device=self.dev, generator=self._gen(30_000_001 + qid))
        return self.tokens(z, base, own), m


def run(a, top=True):
    """One workload: build the DB shard, time `a.steps` steps, return the result record (rank 0; None elsewhere).
    top=False: a sub-record of the N=1 line (own context, no process group, no CPU leg unless asked for)."""
    global FILTER_KIND
    world = int(os.environ.get("WORLD_SIZE", "1")) if top else 1
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1 and top:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)

    S, K, D = a.segments, a.clusters, a.dim
    H, W = a.height, a.width
    N = (H // 14) * (W // 14)
    Hm, Wm = H // 2, W // 2              # SAM masks at half resolution (place_rec_SAM_DINO.py:61)
    use_pca = not a.no_pca
    P = a.pca_dim if use_pca else K * D
    nQ, nR = a.query_images, a.db_images

    eng = SegVLADEngine(local)
    if a.search_stats:
        eng.set_option("search_stats", 1)
    for kv in a.set:
        eng.set_option(*kv.split("=", 1))
    C_np = synth.make_vocab(K, D, seed=1000)
    eng.set_vocab(C_np)
    C = torch.from_numpy(C_np).to(dev)
    if use_pca:
        g = torch.Generator(device=dev)
        g.manual_seed(5000)
        comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
        mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
        var = torch.logspace(-3, -6, P, device=dev)
        eng.pca_set(mean, comps, var, whiten=True)
        del comps
    # The fused VLAD -> PCA call picks its form ("project" / "planes") per call from the batch size; the two agree to ~1e-5
    # relative, not bit for bit.  Every describe of this run -- DB build, timed steps, the oracle check's 4-image batch --
    # uses the form the TIMED batch would get, so that the check covers the measured path.
    pca_path = None
    if use_pca:
        nql = int(shard_images(nQ, world)[rank + 1] - shard_images(nQ, world)[rank])
        pca_path = "project" if (eng_pca_products(K * D, P) == 3 and D % 32 == 0 and P % 4 == 0 and S * K >= 1.25 * N
                                 and nql * N >= 128 * K) else "planes"
        eng.set_option("pca_path", pca_path)
    pipe = SegVLADPipeline(eng, H, W, 14, order=a.order, use_pca=use_pca)
    fac = ImageFactory(dev, C, N, S, Hm, Wm, a.query_own, a.group)
    # --pipeline: the describe stage gets its own context (vocabulary + PCA model) and its own stream, so that batch
    # i+1 is described (HBM side) under the search of batch i (matrix pipe side); `eng` keeps the index
    eng_d, pipe_d, s_desc = eng, pipe, None
    # (also created for the "pipelined" sub-measurement of the default N=1 line, which times both modes on one index)
    also_pipelined = top and world == 1 and not a.pipeline and not a.no_sub_records and not a.sweep_own and not a.shard_sim_only
    if a.pipeline or also_pipelined:
        eng_d = SegVLADEngine(local)
        if pca_path:
            eng_d.set_option("pca_path", pca_path)
        eng_d.set_vocab(C_np)
        if use_pca:
            g = torch.Generator(device=dev)
            g.manual_seed(5000)
            comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
            mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
            eng_d.pca_set(mean, comps, torch.logspace(-3, -6, P, device=dev), whiten=True)
            del comps
        # (no per-batch empty-mask read-back: it is a host synchronisation in the middle of the describe stage)
        pipe_d = SegVLADPipeline(eng_d, H, W, 14, order=a.order, use_pca=use_pca, check_empty=False)
        s_desc = torch.cuda.Stream(device=dev)

    # ---- queries: tau = seeded choice of reference images ---------------------------------------------
    rq = np.random.Generator(np.random.PCG64(4000))
    tau = rq.integers(0, nR, size=nQ)
    qb = shard_images(nQ, world)
    q_lo, q_hi = int(qb[rank]), int(qb[rank + 1])
    nq_local = q_hi - q_lo
    q_tok = torch.empty(nq_local, D, N, device=dev)
    q_msk = torch.empty(nq_local * S, Hm, Wm, dtype=torch.uint8, device=dev)

    def make_queries():
        for j, qi in enumerate(range(q_lo, q_hi)):
            t, m = fac.query(int(tau[qi]), qi)
            q_tok[j] = t
            q_msk[j * S:(j + 1) * S] = m

    make_queries()
    q_off_local = (np.arange(nq_local + 1) * S).astype(np.int32)
    q_off_all = (np.arange(nQ + 1) * S).astype(np.int32)

    # ---- DB shard (untimed build through the same kernels) -----------------------------------------------
    t_build0 = time.time()
    ib = shard_images(nR, world)
    r_lo, r_hi = int(ib[rank]), int(ib[rank + 1])
    rows = torch.empty((r_hi - r_lo) * S, P, device=dev)
    bb = a.build_batch
    tok = torch.empty(bb, D, N, device=dev)
    msk = torch.empty(bb * S, Hm, Wm, dtype=torch.uint8, device=dev)
    for b0 in range(r_lo, r_hi, bb):
        nb = min(bb, r_hi - b0)
        for j in range(nb):
            t, m = fac.reference(b0 + j)
            tok[j] = t
            msk[j * S:(j + 1) * S] = m
        offs = (np.arange(nb + 1) * S).astype(np.int32)
        d = pipe.describe(tok[:nb], msk[:nb * S], offs)
        rows[(b0 - r_lo) * S:(b0 - r_lo + nb) * S] = d
    del tok, msk
    img_of_seg = torch.arange(r_lo, r_hi, device=dev, dtype=torch.int32).repeat_interleave(S)
    index = ShardedSegmentIndex(eng, rank=rank, world=world, device=dev, native_comm=a.native_comm and world > 1)
    index.build(rows, img_of_seg)
    rows_keep = rows if (world == 1 and not a.no_cpu_baseline) else None   # host copy source for the CPU-baseline leg
    del rows
    torch.cuda.synchronize()
    t_build = time.time() - t_build0

    # ---- one step ---------------------------------------------------------------------------------------------
    def step():
        qd = pipe.describe(q_tok, q_msk, q_off_local)
        if world > 1:   # every rank needs all query descriptors: one all_gather of the ragged slices
            qd = index.gather_rows(qd, [int(qb[r + 1] - qb[r]) * S for r in range(world)])
        return index.retrieve(qd, q_off_all, 200, 50, 5, vote_depth_only=a.vote_depth)

    # pipelined steps: K describes + K searches, describe(i+1) enqueued on its own stream BEFORE search(i) is issued (the
    # search synchronises with the host twice; everything it needs to overlap with must already be in the queue)
    def describe_async():
        s_desc.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s_desc):
            qd = pipe_d.describe(q_tok, q_msk, q_off_local)
            ev = torch.cuda.Event()
            ev.record(s_desc)
        return qd, ev

    def steps_pipelined(n_steps, marks=None):
        out_ = None
        nxt = describe_async()
        for i in range(n_steps):
            qd, ev = nxt
            if i + 1 < n_steps:
                nxt = describe_async()
            torch.cuda.current_stream(dev).wait_event(ev)
            qd.record_stream(torch.cuda.current_stream(dev))   # allocated on the describe stream, consumed on this one
            if world > 1:
                qd = index.gather_rows(qd, [int(qb[r + 1] - qb[r]) * S for r in range(world)])
            out_ = index.retrieve(qd, q_off_all, 200, 50, 5)
            if marks is not None:
                marks[i + 1].record()
        return out_

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gt = [[int(t)] for t in tau]
    # 17places-style ground truth (gt.py:60-64: every frame within the localisation radius counts): the whole sibling group
    gt_group = [list(range(int(t) // a.group * a.group, min(nR, (int(t) // a.group + 1) * a.group))) for t in tau]
    if a.sweep_own:   # difficulty calibration: device Recall@1 as a function of --query-own (debug; prints and exits)
        for v in [float(x) for x in a.sweep_own.split(",")]:
            fac.query_own = v
            make_queries()
            pr = step()[0].cpu().numpy()
            rc = recall_at(pr, gt, 5)
            grp = float(np.mean(pr[:, 0] // a.group == tau // a.group))
            if rank == 0:
                print(f"[sweep] query_own={v:.3f}: Recall@1 {rc[0]:.3f} Recall@5 {rc[4]:.3f}, right sibling group {grp:.3f}", file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        return None

    for _ in range(a.warmup):
        out = step()
    if a.pipeline and a.warmup:
        out = steps_pipelined(min(2, a.warmup))
    FILTER_KIND = eng.search_stats()["filter"]
    for e_ in ({eng, eng_d} if a.pipeline else {eng}):
        e_.set_profiling(True)
        e_.profile_reset()
    # HIP events on the stream the steps are issued on (hipEvent-timed steps, median), beside the wall clock
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    pmc_cal = None
    if a.pmc_calibrate and top:
        # counter passes over THIS process (tools/gpu_round_artifacts.sh pmc, rocprofv3 --pmc with --kernel-include-regex = the
        # library's kernels + this one): torch.sign over 1 GiB (2^30 B read, 2^30 B written) is dispatched right BEFORE and right
        # AFTER the timed steps, outside the fences -- the byte calibration of FETCH_SIZE / WRITE_SIZE and, by dispatch order,
        # the WINDOW tools/pmc_summary.py cuts the timed steps' dispatches out of (the counter evidence
        # comes from the timed path, not from a replay)
        pmc_cal = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        torch.cuda.synchronize()
        pmc_cal.sign()
        torch.cuda.synchronize()
    if world > 1:
        index.profile_collectives(True)
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    if a.pipeline:
        out = steps_pipelined(a.steps, marks)
    else:
        for i in range(a.steps):
            out = step()
            marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    coll_ms = index.collective_ms() if world > 1 else {}
    if world > 1:
        index.profile_collectives(False)
    if pmc_cal is not None:
        pmc_cal.sign()
        torch.cuda.synchronize()
        del pmc_cal
    for e_ in {eng, eng_d}:
        e_.set_profiling(False)
    step_ms_events = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    print(f"[bench] adjacency flag checks so far {getattr(pipe, 'n_flag_checks', 0)}, flagged images {getattr(pipe, 'n_flagged_images', 0)}, "
          f"eager countdown {pipe._eager_flags}", file=sys.stderr)
    pipelined = None
    if also_pipelined:   # the same K steps with describe(i+1) issued under search(i): same index, same inputs, same fences
        out_p = steps_pipelined(2)
        fence()
        tp0 = time.perf_counter()
        out_p = steps_pipelined(a.steps)
        fence()
        dtp = time.perf_counter() - tp0
        pipelined = {"value": nQ * a.steps / dtp, "ms_per_step": dtp / a.steps * 1e3, "vs_serial": dt / dtp,
                     "predictions_identical": bool(torch.equal(out_p[0], out[0])),
                     "note": "describe of batch i+1 on its own context + stream under the search of batch i (bench.py --pipeline "
                             "makes this the timed mode); the search's persistent filter kernel leaves the describe kernels few "
                             "CUs, and both draw on the same power budget"}
    sstats = eng.search_stats()   # (of the timed run's last search: before any sub-measurement searches again)
    fp32_rec = None
    if top and world == 1 and not a.no_sub_records and not a.sweep_own and FILTER_KIND != "fp32" and not a.shard_sim_only:
        # The SAME workload with the same-arithmetic filter (option knn_filter=fp32: fp32 MFMA distances in every level, no
        # 16-bit product anywhere): what the line's fp16 pruning buys, and the evidence that it changes nothing -- the
        # searches' (d2, idx) must be BIT-identical and the predictions identical.
        qd_chk = pipe.describe(q_tok, q_msk, q_off_local)
        d2_a, idx_a = eng.search(qd_chk, 200)
        eng.set_option("knn_filter", "fp32")
        out_f = step()
        fence()
        tf0 = time.perf_counter()
        n_f = 2
        for _ in range(n_f):
            out_f = step()
        fence()
        dtf = time.perf_counter() - tf0
        d2_b, idx_b = eng.search(qd_chk, 200)
        fp32_rec = {"ms_per_step": dtf / n_f * 1e3, "value": nQ * n_f / dtf, "steps": n_f, "filter": eng.search_stats()["filter"],
                    "predictions_identical": bool(torch.equal(out_f[0], out[0])),
                    "d2_idx_bit_identical": bool(torch.equal(d2_a, d2_b) and torch.equal(idx_a, idx_b)),
                    "speedup_of_the_timed_run": (dtf / n_f) / (dt / a.steps),
                    "note": "same steps, option knn_filter=fp32 (every filter level on fp32 MFMA, peak 157.3 TFLOP/s); the timed run's "
                            "fp16 product only prunes, under a proven error bound, and its survivors are re-evaluated in fp32"}
        eng.set_option("knn_filter", "auto")
        del qd_chk, d2_a, idx_a, d2_b, idx_b, out_f
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if a.debug_timing and rank == 0:
        def tick(label, t_prev):
            torch.cuda.synchronize()
            t = time.perf_counter()
            print(f"[debug-timing] {label}: {(t - t_prev) * 1e3:.2f} ms", file=sys.stderr)
            return t
        torch.cuda.synchronize()
        t = time.perf_counter()
        bits = eng.incidence(q_msk, H, W, 14); t = tick("incidence", t)
        cent = eng.mask_centroids(q_msk).cpu().numpy(); t = tick("centroids + D2H", t)
        from revisit_anything_amd.pipeline import adjacency_batch
        adj = adjacency_batch(cent, q_off_local, a.order, pipe.adj_workers); t = tick("host adjacency (Qhull)", t)
        desc = eng.seg_vlad(q_tok, bits, q_off_local, adj)["out"]; t = tick("seg_vlad (incl. adj H2D)", t)
        qd = eng.pca_apply(desc, l2norm=True) if use_pca else desc; t = tick("pca", t)
        d2, idx = eng.search(qd, 200); t = tick("search", t)
        sims, m = eng.sims_from_d2(d2, idx, 50); t = tick("sims", t)
        eng.vote(m, sims, q_off_all, n_top=5, img_of_seg=index.img_of_seg_global); t = tick("vote", t)
    pred = out[0].cpu().numpy()
    if a.dump_preds and rank == 0:
        np.save(a.dump_preds, pred)
    recalls = recall_at(pred, gt, 5)
    recalls_group = recall_at(pred, gt_group, 5)

    # ---- per-stage device time (HIP events on the engine stream, summed over the timed steps) ----------------
    stages = {}
    # ("describe": segvlad_describe as the stream sees it -- since round 5 the mask branch, incidence + adjacency, runs on the
    #  context's side stream beside the assignment pass, so the parts' times no longer add up to the stage's)
    for s in ("incidence", "adjacency", "assign", "prep", "aggregate", "pca", "describe", "knn_level0", "knn_gemm", "knn_select", "knn_redo",
              "knn_fallback", "vote"):
        for e_ in ({eng, eng_d} if a.pipeline else {eng}):
            try:
                ms, n = e_.stage_ms(s)
                stages[s] = {"ms_per_step": ms / a.steps, "launches_per_step": n / a.steps}
            except Exception:
                pass
    b2b = None
    if top and world == 1 and not a.sweep_own:
        # the same searches back to back, without a describe stage in between (a diagnostic: the filter kernel is power
        # limited, and what runs before it -- and what its operands look like -- moves its clock)
        qd_b = pipe.describe(q_tok, q_msk, q_off_local)
        eng.search(qd_b, 200)
        eng.set_profiling(True)
        eng.profile_reset()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for _ in range(5):
            eng.search(qd_b, 200)
        torch.cuda.synchronize()
        b2b = {"ms_per_search": (time.perf_counter() - tb0) / 5 * 1e3, "knn_gemm_ms": eng.stage_ms("knn_gemm")[0] / 5,
               "knn_select_ms": eng.stage_ms("knn_select")[0] / 5}
        eng.set_profiling(False)
        del qd_b
    per_rank = None
    if world > 1:   # every rank's stage times travel to rank 0 (the slowest rank sets the step time)
        mine = {k: round(v["ms_per_step"], 4) for k, v in stages.items()}
        mine["n_local_rows"] = int(index.n_local)
        # (round 6) the collectives as THIS rank's stream saw them (event pairs around each call: a rank that waits for a slower
        # peer shows it here) and the bytes it sent / received per step: the first real SCALE run is diagnosable from its one line
        mine["collective_ms_per_step"] = {k: round(v / a.steps, 4) for k, v in coll_ms.items()}
        mine["collective_bytes_per_step"] = ShardedSegmentIndex.collective_bytes(world, nQ * S, 50, P, [int(qb[r + 1] - qb[r]) * S for r in range(world)])
        try:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        except Exception as e:   # diagnostics only: never lose the line over them
            per_rank = [f"all_gather_object failed: {e}"]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None

    # ---- roofline of the dominant kernel ------------------------------------------------------------------------
    n_local_rows = index.n_local
    d_knn = P
    wl_key = f"q{nQ}x{S}_db{nR * S}_d{d_knn}_k{K}_w{world}"
    main_stages = {k: v for k, v in stages.items() if k not in ("knn_fallback", "knn_redo")}
    dom = max(main_stages, key=lambda k: main_stages[k]["ms_per_step"]) if main_stages else None
    roof = None
    if dom in ("knn_gemm", "knn_select", "pca"):
        # the exact-kNN stage (distance GEMM + selection) and the PCA projection are MFMA bound;
        # the selection kernels are accounted to the kNN stage's GEMM as overhead, the roofline is quoted on the GEMM
        key = "pca" if dom == "pca" else "knn_gemm"
        if key == "knn_gemm":
            flops_step = 2.0 * nQ * S * n_local_rows * d_knn          # 2 * B_q * N_r * d (algorithmic)
            kern = ("knn_f16_filter_kernel (Q.R^T as one fp16 MFMA product per fp32 fma, global->LDS DMA, fused d2 + "
                    "threshold-filter epilogue; exact fp32 refinement of the survivors)" if FILTER_KIND == "f16" else
                    "knn_bf16_filter_kernel (Q.R^T as 3 bf16 MFMA products hi.hi+hi.lo+lo.hi, fused d2 + threshold-filter "
                    "epilogue; exact fp32 refinement of the survivors)")
        else:
            flops_step = 2.0 * nq_local * S * (K * D) * P              # 2 * S * K*D * P per image
            kern = ("gemm_f16x3_kernel (PCA projection as 3 fp16 MFMA products of a two-term split, fused mean-subtract + "
                    "whitening scale)" if eng_pca_products(K * D, P) == 3 else
                    "gemm_nt_kernel<0> (PCA projection, fp32 MFMA, fused mean-subtract + whitening scale)")
        launches = stages[key]["launches_per_step"]
        avg_ms = stages[key]["ms_per_step"] / max(launches, 1)      # the named kernel's own launches (rocprofv3 "AverageNs")
        stage_ms_ = stages[key]["ms_per_step"]
        if key == "knn_gemm" and "knn_level0" in stages:             # + the sampled level's exact fp32 GEMM (another kernel):
            stage_ms_ += stages["knn_level0"]["ms_per_step"]         #   overhead of the pass, charged to `achieved`
        f32_equiv = flops_step / (stage_ms_ * 1e-3) / 1e12
        if key == "knn_gemm":
            # the filter GEMM runs on the 16-bit MFMA pipe: PRODUCTS_PER_FMA MFMA products per algorithmic fp32 fma
            prods = eng_filter_products()
            ach, peak, unit_note = f32_equiv * prods, PEAK_16BIT_MFMA_TFLOPS, f"{prods} x 16-bit MFMA product(s) per fp32 fma"
        elif eng_pca_products(K * D, P) == 3:
            ach, peak, unit_note = f32_equiv * 3, PEAK_16BIT_MFMA_TFLOPS, "3 x fp16 MFMA products per fp32 fma"
        else:
            ach, peak, unit_note = f32_equiv, PEAK_F32_MFMA_TFLOPS, "fp32 MFMA"
        traffic, mfma_util, traffic_src = pmc_counters(kern.split(" ")[0], wl_key)
        executed = None
        if key == "knn_gemm" and sstats.get("levels", 0) >= 1:
            # rows the filter launches of one step actually multiplied: the levels' strided samples + the last level (all rows, or --
            # level_carry -- the rows the stride-16 level has not seen); `achieved` stays ALGORITHMIC (every pair counted once)
            lv = int(sstats["levels"])
            rows_exec = sum(-(-n_local_rows // 16 ** j) for j in range(lv)) - int(sstats.get("carry_rows", 0))
            executed = {"filter_rows_per_step": rows_exec, "vs_algorithmic": rows_exec / max(n_local_rows, 1),
                        "carry_rows": int(sstats.get("carry_rows", 0)),
                        "executed_tflops": 2.0 * nQ * S * rows_exec * d_knn * eng_filter_products() / (stages[key]["ms_per_step"] * 1e-3) / 1e12}
        roof = {"kernel": kern, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "executed": executed,
                "traffic": traffic, "mfma_util": mfma_util, "traffic_source": traffic_src, "avg_launch_ms": avg_ms,
                "launches_per_step": launches, "dominant_stage": dom,
                "arithmetic": unit_note, "fp32_equivalent_tflops": f32_equiv,
                "fp32_equivalent_vs_fp32_mfma_peak": f32_equiv / PEAK_F32_MFMA_TFLOPS,
                "effective_clock_ghz": pmc_field(kern.split(" ")[0], wl_key, "effective_clock_ghz")}
        if top and not a.no_ubench and peak == PEAK_16BIT_MFMA_TFLOPS:
            # what the matrix pipe of THIS box delivers to a kernel that does nothing but v_mfma_f32_32x32x16_f16 on random
            # operands (the chip is power limited: it does not hold its nominal clock), and with the LDS fragment reads of a
            # 2 x 4 wave tile beside them -- measured now (tools/ubench/mfma_peak), so that `frac` can be read against it
            ub = mfma_ubench()
            if ub:
                roof["ubench"] = ub
                ceil_tf = ub.get("mfma16_only_random_tflops") if key == "knn_gemm" else None   # the filter's own MFMA shape
                ceil_tf = ceil_tf or ub.get("mfma_only_random_tflops")
                if ceil_tf:
                    roof["mfma_only_ceiling_ms"] = flops_step * eng_filter_products() / (ceil_tf * 1e12) * 1e3
                    roof["frac_of_mfma_only_ceiling"] = ach / ceil_tf
                if ub.get("mfma_plus_lds_reads_tflops"):
                    roof["mfma_plus_lds_reads_ceiling_ms"] = flops_step * eng_filter_products() / (ub["mfma_plus_lds_reads_tflops"] * 1e12) * 1e3
    elif dom is not None:
        bytes_img = 4 * D * N + 4 * S * K * D + S * N / 8 + S * S + S * Hm * Wm
        ms = sum(stages[s]["ms_per_step"] for s in ("incidence", "assign", "prep", "aggregate") if s in stages)
        ach = bytes_img * nq_local / (ms * 1e-3) / 1e9
        roof = {"kernel": "segment-VLAD kernels (incidence+assign+prep+aggregate)", "bound": "hbm", "achieved": ach,
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None, "dominant_stage": dom}
    # secondary: the HBM-bound VLAD stage, always reported.  With the fused projection the K*D-wide fp32 descriptor
    # never reaches HBM.  In its "planes" form the stage writes two fp16 planes of the same size instead (4 S K D bytes
    # either way); in its "project" form (see roofline_pca) it writes the fp16 planes of the token residuals (4 N D) and
    # the block norms (4 S K).
    pca_form = pca_path
    vlad_ms = sum(stages[s]["ms_per_step"] for s in ("incidence", "assign", "prep", "aggregate") if s in stages)
    out_bytes = 4 * N * D + 4 * S * K if pca_form == "project" else 4 * S * K * D
    bytes_img = 4 * D * N + out_bytes + S * N / 8 + S * S + S * Hm * Wm
    vlad_roof = {"bound": "hbm", "achieved": bytes_img * nq_local / (vlad_ms * 1e-3) / 1e9 if vlad_ms else None,
                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "alg_bytes_per_image": bytes_img, "stage_ms": vlad_ms,
                 "note": "incidence + assign + prep + aggregate; the 'project' form moves 31 % fewer bytes per image than the "
                         "'planes' form (22.7 vs 32.9 MB), so fractions are not comparable across the two -- stage_ms is"}
    if vlad_roof["achieved"]:
        vlad_roof["frac"] = vlad_roof["achieved"] / PEAK_HBM_GBS
        # (verdict r04) the same stage priced on the bytes NO implementation can avoid -- the tokens and the masks in, the
        # projected descriptors (or the K*D-wide ones) out -- instead of this implementation's own intermediate planes
        irr = 4 * D * N + S * Hm * Wm + (4 * S * P if use_pca else 4 * S * K * D)
        vlad_roof["irreducible_bytes_per_image"] = irr
        vlad_roof["frac_on_irreducible_bytes"] = irr * nq_local / (vlad_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        if "describe" in stages:   # the whole describe call as the stream sees it (mask branch beside the assignment pass; PCA included)
            dms = stages["describe"]["ms_per_step"]
            vlad_roof["describe_ms_wall"] = dms
            vlad_roof["describe_frac_on_irreducible_bytes"] = irr * nq_local / (dms * 1e-3) / 1e9 / PEAK_HBM_GBS
    if pca_form == "project":
        # round 4: the block norms + residual planes come from the Gram kernels (tasks of <= 32 / <= 64 tokens), the
        # block-sum kernel keeps the larger tasks: the quoted traffic is the sum over the three launches
        agg_kernel = "gram_norms_kernel<1> + gram_norms_kernel<2> + token_norms_kernel"
        parts = [pmc_counters(kn, wl_key)[0] for kn in ("gram_norms_kernelILi1E", "gram_norms_kernelILi2E", "token_norms_kernelILb0E")]
        agg_traffic = sum(p for p in parts if p) if any(parts) else None
    else:
        agg_kernel = "aggregate_kernel"
        agg_traffic, _, agg_src = pmc_counters(agg_kernel, wl_key)
    vlad_roof["aggregate_kernel"] = agg_kernel
    vlad_roof["aggregate_traffic"] = agg_traffic
    pca_roof = None
    if use_pca and "pca" in stages:
        # The fused call picks the smaller product (api.hip, images_impl): "project" = every token's residual times its
        # cluster's D x P slice of the components, segments aggregated afterwards in the P-d space (N D P per image);
        # "planes" = the finished descriptors times the components (S K D P per image).  `achieved` counts the flops
        # EXECUTED by the chosen form (fp16 products: 3 per multiply); the stage also holds the P-space aggregation /
        # the row normalisation.
        prods = eng_pca_products(K * D, P)
        form = pca_form
        rows, depth = (nq_local * N, D) if form == "project" else (nq_local * S, K * D)
        pf = 2.0 * rows * depth * P * prods / (stages["pca"]["ms_per_step"] * 1e-3) / 1e12
        pt, pu, psrc = pmc_counters("gemm_f16x3_kernel", wl_key)
        pca_roof = {"bound": "mfma", "form": form, "achieved": pf,
                    "peak": PEAK_16BIT_MFMA_TFLOPS if prods == 3 else PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "traffic": pt, "mfma_util": pu, "traffic_source": psrc,
                    "flops_vs_descriptor_projection": rows * depth / (nq_local * S * K * D),
                    "alg_bytes_per_launch": 2.0 * 2 * (rows * depth + P * K * D) + 4.0 * rows * P * (2 if form == "project" else 1)}
        pca_roof["frac"] = pca_roof["achieved"] / pca_roof["peak"]
        # the same stage time priced at the work the reference's formulation needs (S K D P per image, x products): what a
        # descriptor-projection GEMM would have to sustain to be as fast
        pca_roof["descriptor_projection_equivalent_tflops"] = pf / pca_roof["flops_vs_descriptor_projection"]
        pca_roof["descriptor_projection_equivalent_frac"] = pca_roof["descriptor_projection_equivalent_tflops"] / pca_roof["peak"]

    # secondary: the kNN stage in its HBM-bound regime (B_q <= 50, i.e. ONE query image per pass)
    stream_roof = None
    if use_pca and world == 1 and n_local_rows > 0:
        qd1 = pipe.describe(q_tok[:1], q_msk[:S], np.array([0, S], dtype=np.int32))
        for _ in range(3):
            eng.search(qd1, 200)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        eng.profile_reset()
        reps = 20
        for _ in range(reps):
            eng.search(qd1, 200)
        torch.cuda.synchronize()
        g_ms = eng.stage_ms("knn_gemm")[0] / reps
        try:
            g_ms += eng.stage_ms("knn_level0")[0] / reps   # the sampled level's exact fp32 GEMM belongs to the pass
        except Exception:
            pass
        s_ms = eng.stage_ms("knn_select")[0] / reps
        eng.set_profiling(False)
        # the whole call as the caller sees it (query preparation, flag read-back, host synchronisation), stage timers off
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.search(qd1, 200)
        torch.cuda.synchronize()
        call_ms = (time.perf_counter() - t0) / reps * 1e3
        # ... and ONE query image through the whole path (describe -> search 200 -> keep 50 -> vote), host to host: what an
        # online caller waits for (the pca_path of the run is the batch's; a one-image describe pads every cluster to a tile)
        off1 = np.array([0, S], dtype=np.int32)
        for _ in range(3):
            index.retrieve(pipe.describe(q_tok[:1], q_msk[:S], off1), off1, 200, 50, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            index.retrieve(pipe.describe(q_tok[:1], q_msk[:S], off1), off1, 200, 50, 5)
        torch.cuda.synchronize()
        image_ms = (time.perf_counter() - t0) / reps * 1e3
        alg = 4.0 * n_local_rows * d_knn + 4.0 * S * d_knn + 8.0 * S * 200     # fp32 DB rows read once
        moved = alg / 2 if FILTER_KIND == "f16" else alg                        # the fp16 filter streams a 2-byte plane
        pass_ms = g_ms + s_ms
        # headline = the bytes the pass ACTUALLY moves (the fp16 plane once + the refine gathers), over the WHOLE pass
        # (filter + select + exact refinement); the SURVEY's fp32 bytes are kept beside it for comparison only
        # round 6: `achieved` / `frac` are priced on the CALL'S WALL CLOCK -- the search as a caller sees it, `reps`
        # calls back to back with one synchronisation at the end (the call no longer synchronises itself: its overflow counters are
        # read on the device, small_pass_kernels.hip) -- not on the sum of the stage timers, which stays beside it (`frac_by_stage_sum`)
        stream_roof = {"bound": "hbm", "B_q": S, "unit": "GB/s", "peak": PEAK_HBM_GBS,
                       "achieved": moved / (call_ms * 1e-3) / 1e9, "frac": moved / (call_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                       "frac_by_call_wall": moved / (call_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                       "frac_by_stage_sum": moved / (pass_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                       "filter_only_gbs": moved / (g_ms * 1e-3) / 1e9, "filter_only_frac": moved / (g_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                       "survey_fp32_bytes_gbs": alg / (pass_ms * 1e-3) / 1e9,
                       "filter_ms": g_ms, "select_refine_ms": s_ms, "pass_ms": pass_ms, "call_ms_wall": call_ms,
                       "one_image_end_to_end_ms_wall": image_ms,
                       "search_stats": eng.search_stats(),
                       "note": "one 50-segment query image per pass over the whole shard; 'achieved' = bytes actually streamed "
                               "(2-byte fp16 plane when the filter is f16) / the wall clock of the call (20 calls back to back, one "
                               "synchronisation at the end); frac_by_stage_sum = the same bytes / (head + filter + select + refine + tail "
                               "stage timers, measured in a separate loop with the timers on)"}

    res = {
        "metric": "query_images_per_sec", "value": nQ * a.steps / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "ms_per_step_hip_event_median": float(np.median(step_ms_events)), "ms_per_step_hip_event_min": float(np.min(step_ms_events)),
        "timing_note": "value / ms_per_step: wall clock over the K steps between two barrier + synchronize fences (the contract); "
                       "the hip_event figures are per-step HIP events on the issuing stream (SURVEY 8d)",
        "mode": "pipelined (describe i+1 on its own context/stream under search i)" if a.pipeline else "serial",
        "options": a.set or None,
        "pipelined": pipelined, "fp32_filter": fp32_rec, "search_back_to_back": b2b,
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "filter_dtype": FILTER_KIND, "pca_gemm_dtype": "f16x3" if eng_pca_products(K * D, P) == 3 else "f32",
        "dtype_note": "every reported distance / similarity / descriptor is fp32-class: the fp16 MFMA product of the kNN stage only "
                      "FILTERS candidates with a rigorous error margin, the survivors are re-evaluated with an fp32 fma chain "
                      "(bit-identical to the all-fp32 path); the projection GEMM is a 3-product fp16 split with fp32 accumulation",
        "data": "synthetic",
        "config": {"workload": f"{nQ} query images x {S} seg vs {nR * S}-segment DB ({nR} ref images), {W}x{H} -> {N} tokens, "
                               f"D={D}, K={K}, {'PCA ' + str(P) if use_pca else 'raw K*D'}, order {a.order}, " + ("search 50 (the vote's depth) / vote 50" if a.vote_depth else "search 200 / vote 50"),
                   "query_images": nQ, "db_segments": nR * S, "segments_per_image": S, "clusters": K, "desc_dim": D,
                   "tokens": N, "pca_dim": P if use_pca else None, "order": a.order, "parallelism": f"db-row-shard x{world}" + (" (C-ABI RCCL communicator)" if a.native_comm and world > 1 else ""),
                   "query_own": a.query_own},
        "recall_at_1": recalls[0], "recall_at_5": recalls[4], "db_build_s": t_build,
        "pred_sha1": __import__("hashlib").sha1(np.ascontiguousarray(pred).astype(np.int64).tobytes()).hexdigest()[:16],
        "pca_path": pca_path, "sibling_group": a.group, "recall_at_1_within_sibling_group": recalls_group[0],
        "search_stats": sstats, "per_rank_stages_ms": per_rank,
        "stages_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in stages.items()},
        # The step as a sum of NON-overlapping parts (round 5).  "describe" is the describe stage as the stream sees it: its mask
        # branch (incidence, adjacency) runs on a second stream BESIDE the assignment pass, so those three timers overlap -- each is
        # inflated by the contention and their sum exceeds the stage (do not add them); the stage itself, the search's three timers
        # and the vote do not overlap.  gap = what the host leaves between them (launch latency, the search's two read-backs).
        "step_accounting_ms": (lambda d_, s_: {"describe": round(d_, 4), "search": round(s_, 4), "vote": round(stages.get("vote", {}).get("ms_per_step", 0.0), 4),
                                              "sum": round(d_ + s_ + stages.get("vote", {}).get("ms_per_step", 0.0), 4),
                                              "step": round(dt / a.steps * 1e3, 4),
                                              "gap": round(dt / a.steps * 1e3 - d_ - s_ - stages.get("vote", {}).get("ms_per_step", 0.0), 4)})(
            stages["describe"]["ms_per_step"] if "describe" in stages else sum(stages[x]["ms_per_step"] for x in ("incidence", "adjacency", "assign", "prep", "aggregate", "pca") if x in stages),
            sum(stages[x]["ms_per_step"] for x in ("knn_level0", "knn_gemm", "knn_select", "knn_redo", "knn_fallback") if x in stages)),
        "roofline": roof, "roofline_vlad": vlad_roof, "roofline_pca": pca_roof, "roofline_knn_stream": stream_roof,
    }

    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a, rows_keep, fac, tau, C_np, use_pca, P, N, S, K, D, H, W, pipe, index, q_tok, q_msk)
        res["oracle_check"] = res["cpu_baseline"].pop("oracle_check")
    else:
        res["cpu_baseline"] = None
    if world > 1 and top:
        dist.destroy_process_group()
    # release this workload's device memory before a sub-record builds its own
    del index, q_tok, q_msk, rows_keep, out
    eng.close()
    if eng_d is not eng:
        eng_d.close()
    torch.cuda.empty_cache()
    return res


def sub_record(a, **over):
    """Another workload measured by the same code path (its own context), trimmed to what a sub-record of the line needs."""
    import copy
    b = copy.copy(a)
    b.no_cpu_baseline, b.no_sub_records, b.no_ubench, b.pipeline = True, True, True, False
    b.dump_preds = b.sweep_own = None
    b.debug_timing = b.pmc_calibrate = False
    b.steps, b.warmup = 3, 1
    for k, v in over.items():
        setattr(b, k, v)
    try:
        r = run(b, top=False)
    except Exception as e:   # a sub-record never takes the headline line down with it
        return {"error": f"{type(e).__name__}: {e}"}
    keep = ("value", "unit", "ms_per_step", "ms_per_step_hip_event_median", "filter_dtype", "recall_at_1", "recall_at_5",
            "recall_at_1_within_sibling_group", "sibling_group", "search_stats", "stages_ms_per_step", "db_build_s", "steps", "pred_sha1")
    out = {k: r[k] for k in keep if k in r}
    out["workload"] = r["config"]["workload"]
    if r.get("oracle_check"):
        out["oracle_check"] = r["oracle_check"]
    return out


def dry_run_collectives_main(a):
    """bench.py --dry-run-collectives (also: tests/test_gpu_sharded.py): see sharded.dry_run_collectives.  One process; a world-1
    process group on the requested backend (nccl = RCCL) is created so that the REAL collectives and the C-ABI communicator run."""
    from revisit_anything_amd.sharded import dry_run_collectives

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    created = False
    if not dist.is_initialized():
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group(a.dist_backend, rank=0, world_size=1)
        created = True
    eng = SegVLADEngine(local)
    rep = dry_run_collectives(eng, dev, worlds=(2, 3, 4, 5, 6, 7, 8), nq=200, k=50, d=64)
    # the C-ABI's own communicator at world size 1: communicator id -> init -> sharded search == plain search, row gather == identity
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    R = torch.nn.functional.normalize(torch.randn(5000, 64, device=dev, generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(100, 64, device=dev, generator=g), dim=1)
    index = ShardedSegmentIndex(eng, rank=0, world=1, device=dev, native_comm=True)
    index.build(R, torch.arange(5000, dtype=torch.int32, device=dev) // 50)
    d2n, idn = index.search(Q, 50)
    d2p, idp = eng.search(Q, 50)
    ok_native = bool(torch.equal(torch.as_tensor(d2n), d2p) and torch.equal(torch.as_tensor(idn), idp))
    rows = eng.allgather_rows(Q.contiguous(), world=1)
    ok_native = ok_native and bool(torch.equal(rows, Q))
    rep["native_comm_world_1"] = {"ok": ok_native, "info": eng.comm_info() if hasattr(eng, "comm_info") else None}
    rep["ok"] = all(v["ok"] for v in rep["worlds"].values()) and ok_native and rep.get("world_1_process_group", {}).get("ok", False)
    eng.close()
    if created:
        dist.destroy_process_group()
    print(json.dumps({"dry_run_collectives": rep}))
    if not rep["ok"]:
        raise SystemExit(4)


def main():
    a = parse()
    if a.dry_run_collectives:
        dry_run_collectives_main(a)
        return
    res = run(a, top=True)
    if res is None:
        return
    world = res["n_gpus"]
    if world == 1 and not a.no_sub_records and not a.no_pca and a.group == 4 and not a.sweep_own and not a.shard_sim_only:
        # BASELINE configs[1] in its literal form (place_rec_main.py:49-60 with pca off): 1000 reference images x 50
        # segments of raw K*D = 98 304-d descriptors, 200 query images, search 200 -- the deep-row fp16 filter with
        # blocked accumulation + the coalesced exact refinement; which filter ran and the list occupancies are in search_stats
        res["config2"] = sub_record(a, no_pca=True, db_images=1000, search_stats=True)
        # a 17places-like temporally redundant database (gt.py:60-64): sibling groups of 31 near-duplicate frames
        res["redundant_db"] = sub_record(a, group=31, search_stats=True)
        # the same two workloads for a caller that does not keep the 200-wide lists (place_rec_main.py:61-75 pickles them only
        # under save_results; the prediction reads 50 columns, :77-85): the single index searches -- and refines -- 50 deep, as
        # every shard of a multi-GPU run already does.  Same predictions; NOT the headline, whose step searches 200.
        for key, over in (("vote_depth", {}), ("config2_vote_depth", {"no_pca": True, "db_images": 1000})):
            r = sub_record(a, vote_depth=True, search_stats=True, **over)
            full = res if key == "vote_depth" else res["config2"]
            if "error" not in r:
                r = {k: r[k] for k in ("value", "unit", "ms_per_step", "search_stats", "stages_ms_per_step", "workload", "pred_sha1") if k in r}
                r["predictions_identical_to_search_200"] = bool(r.get("pred_sha1")) and r.get("pred_sha1") == full.get("pred_sha1")
            res[key] = r
    if world == 1 and a.shard_sim > 1 and not a.no_sub_records and not a.no_pca and not a.sweep_own:
        try:
            res["shard_sim"] = shard_sim(a, a.shard_sim)
        except Exception as e:
            res["shard_sim"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not a.no_ubench and res.get("roofline") and res["roofline"].get("dominant_stage", "").startswith("knn"):
        # an independent yardstick for `frac`: a PLAIN fp16 GEMM of the filter's shape through torch.matmul (hipBLASLt /
        # rocBLAS), measured now on this box -- measurement only, never the product path
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("yardstick_gemm", os.path.join(ROOT, "tools", "yardstick_gemm.py"))
            yg = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(yg)
            y = yg.measure(a.query_images * a.segments, a.db_images * a.segments, a.pca_dim)
            res["roofline"]["yardstick_gemm_tflops"] = y["yardstick_gemm_tflops"]
            res["roofline"]["yardstick_gemm"] = y
            res["roofline"]["achieved_vs_yardstick"] = res["roofline"]["achieved"] / y["yardstick_gemm_tflops"]
        except Exception as e:
            res["roofline"]["yardstick_gemm"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(res))


def shard_sim(a, W):
    """ONE rank of a W-way row-sharded run, emulated on the one GPU (no node with W GPUs is available to the builder): the
    rank's 1/W shard of the database, its 1/W slice of the query images to describe, ALL query segments searched k_vote deep
    against the shard (sharded.py: retrieve), the W-way merge of the exchanged lists and the vote.  The two collectives --
    the query-descriptor all-gather (nQ*S*P*4 bytes in total) and the packed top-k record all-gather (nQ*S*k_vote*12 bytes per
    rank) -- are NOT executed: their byte counts and an xGMI estimate are reported beside the compute."""
    S, K, D, H, W_, P = a.segments, a.clusters, a.dim, a.height, a.width, a.pca_dim
    N, Hm, Wm = (H // 14) * (W_ // 14), H // 2, W_ // 2
    nQ, nR = a.query_images, a.db_images
    nR_l, nQ_l = nR // W, max(1, nQ // W)
    dev = torch.device("cuda:0")
    eng = SegVLADEngine(0)
    C_np = synth.make_vocab(K, D, seed=1000)
    eng.set_vocab(C_np)
    g = torch.Generator(device=dev)
    g.manual_seed(5000)
    comps = torch.randn(P, K * D, device=dev, generator=g) / (K * D) ** 0.5
    mean = torch.randn(K * D, device=dev, generator=g) * (0.2 / (K * D) ** 0.5)
    eng.pca_set(mean, comps, torch.logspace(-3, -6, P, device=dev), whiten=True)
    del comps
    eng.set_option("pca_path", "project")
    pipe = SegVLADPipeline(eng, H, W_, 14, order=a.order, use_pca=True)
    fac = ImageFactory(dev, torch.from_numpy(C_np).to(dev), N, S, Hm, Wm, a.query_own, a.group)
    bb = a.build_batch
    tok = torch.empty(bb, D, N, device=dev)
    msk = torch.empty(bb * S, Hm, Wm, dtype=torch.uint8, device=dev)

    def describe_images(ids, maker):
        out = torch.empty(len(ids) * S, P, device=dev)
        for b0 in range(0, len(ids), bb):
            nb = min(bb, len(ids) - b0)
            for j in range(nb):
                t, m = maker(ids[b0 + j])
                tok[j] = t
                msk[j * S:(j + 1) * S] = m
            out[b0 * S:(b0 + nb) * S] = pipe.describe(tok[:nb], msk[:nb * S], (np.arange(nb + 1) * S).astype(np.int32))
        return out

    rows = describe_images(list(range(nR_l)), fac.reference)                    # rank 0's shard: reference images [0, nR / W)
    tau = np.random.Generator(np.random.PCG64(4000)).integers(0, nR, size=nQ)
    qd_all = describe_images(list(range(nQ)), lambda qi: fac.query(int(tau[qi]), qi))   # stands for the gathered descriptors
    q_tok = torch.empty(nQ_l, D, N, device=dev)
    q_msk = torch.empty(nQ_l * S, Hm, Wm, dtype=torch.uint8, device=dev)
    for j in range(nQ_l):
        t, m = fac.query(int(tau[j]), j)
        q_tok[j] = t
        q_msk[j * S:(j + 1) * S] = m
    eng.db_add(rows, None)
    img_of_seg = torch.arange(nR, device=dev, dtype=torch.int32).repeat_interleave(S)   # the global map every rank holds
    q_off_l, q_off_all = (np.arange(nQ_l + 1) * S).astype(np.int32), (np.arange(nQ + 1) * S).astype(np.int32)
    kv = 50

    eng.hint_query_groups(q_off_all)   # (what ShardedSegmentIndex.retrieve tells the engine: an image's rows as one refinement group)
    for kv_ in a.set:
        eng.set_option(*kv_.split("=", 1))
    # the other ranks' lists: this rank's own, with ids moved into their shards (distinct ids, same merge work).  Three torch
    # kernels stand for the unpack of the gathered records (sharded.unpack_topk_records: two copies) -- round 5 built the lists with
    # ten (a `cat` of W shifted copies), 0.2 ms of emulation glue inside the figure
    id_shift = (torch.arange(W, device=dev, dtype=torch.int64) * (nR_l * S)).repeat_interleave(kv)[None, :]

    def rank_step():
        pipe.describe(q_tok, q_msk, q_off_l)                                     # this rank's slice of the query images
        d2, idx = eng.search(qd_all, kv)                                         # all query segments against the shard
        d2c = d2.repeat(1, W)
        idc = idx.repeat(1, W) + id_shift
        md, mi = eng.merge_topk(d2c, idc, W, kv)
        sims, m = eng.sims_from_d2(md, mi, kv)
        return eng.vote(m, sims, q_off_all, n_top=5, img_of_seg=img_of_seg)

    for _ in range(2):
        rank_step()
    eng.set_profiling(True)
    eng.profile_reset()
    torch.cuda.synchronize()
    n_it = 5
    t0 = time.perf_counter()
    for _ in range(n_it):
        rank_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_it * 1e3
    stages = {}
    for st in ("incidence", "adjacency", "assign", "prep", "aggregate", "pca", "knn_level0", "knn_gemm", "knn_select", "vote"):
        try:
            stages[st] = round(eng.stage_ms(st)[0] / n_it, 4)
        except Exception:
            pass
    gather_bytes, rec_bytes = nQ * S * P * 4, nQ * S * kv * 12
    # all-gather over xGMI (fully connected inside a node, 7 links per GPU, ~153 GB/s peak each: MI355X_MICROARCH.md): every
    # rank receives (W-1)/W of the total, one peer per link, all links at once; 50 GB/s per link is a conservative planning
    # figure for a collective of this size, the one-link serial figure is the pessimistic bound
    bytes_in = gather_bytes * (W - 1) / W + rec_bytes * (W - 1)
    est_comm_ms = bytes_in / (min(W - 1, 7) * 50e9) * 1e3
    est_comm_ms_one_link = bytes_in / 50e9 * 1e3
    eng.close()
    torch.cuda.empty_cache()
    return {"world": W, "per_rank_ms": ms, "stages_ms": stages, "shard_rows": nR_l * S, "query_images_described": nQ_l,
            "implied_upper_bound_images_per_s": nQ / (ms * 1e-3),
            "implied_upper_bound_with_comm_estimate_images_per_s": nQ / ((ms + est_comm_ms) * 1e-3),
            "collectives_not_executed": {"query_descriptor_allgather_bytes_total": gather_bytes, "topk_record_allgather_bytes_per_rank": rec_bytes,
                                         "estimate_ms_at_50GBs_per_link_all_links": est_comm_ms,
                                         "estimate_ms_at_50GBs_one_link": est_comm_ms_one_link},
            "note": "one rank's compute of a W-way run on ONE GPU (host-side torch glue of the emulation included in per_rank_ms): an "
                    "UPPER bound on the W-GPU rate -- collectives, their synchronisation and load imbalance come on top; "
                    "the strong-scaling efficiency this implies against the N=1 line is value_bound / (W x N=1 value)"}


def cpu_baseline(a, db_rows, fac, tau, C_np, use_pca, P, N, S, K, D, H, W, pipe, index, q_tok, q_msk):
    """The oracle (NumPy restatement of the reference, 'port') on this box's host cores over a BOUNDED sample of the
    same workload -- and the checker of the device path:

    * timed (the baseline): adjacency + seg-VLAD + PCA for the first `n_v` query images; ONE image's 50 segments
      against the full DB with an fp32 sgemm (what faiss IndexFlatL2 computes -- this is NumPy/OpenBLAS, NOT faiss);
      the vote for it;
    * checked (untimed): the same `n_v` images through the oracle's fp64 exact kNN + vote, against a device run over
      exactly these images (the vote's min/max are global over the batch, so both sides must see the same batch)."""
    from oracle import segvlad_oracle as O

    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    n_v = max(1, min(int(a.verify_images), q_tok.shape[0]))
    t_desc = 0.0
    descs, o_labels = [], []
    for qi in range(n_v):
        t, m = q_tok[qi].cpu().numpy(), q_msk[qi * S:(qi + 1) * S].cpu().numpy().astype(bool)
        t0 = time.perf_counter()
        inc = O.incidence(m, H, W)
        adj = O.nbr_masks_agg_fast_single([x for x in m], a.order) if a.order else None
        v, aux = O.seg_vlad(t, inc, C_np, adj, return_aux=True)
        t_desc += time.perf_counter() - t0
        descs.append(v)
        o_labels.append(aux["labels"])
    # PCA transform of the sampled descriptors with the same synthetic model (regenerated on the host)
    t_pca = 0.0
    if use_pca:
        g = torch.Generator(device=fac.dev)
        g.manual_seed(5000)
        comps = (torch.randn(P, K * D, device=fac.dev, generator=g) / (K * D) ** 0.5)
        mean = (torch.randn(K * D, device=fac.dev, generator=g) * (0.2 / (K * D) ** 0.5)).cpu().numpy()
        comps = comps.cpu().numpy().astype(np.float64)
        var = torch.logspace(-3, -6, P).numpy()
        t0 = time.perf_counter()
        ys = [O.normalize_feat(O.pca_transform(v, mean, comps, var, True)) for v in descs]
        t_pca = time.perf_counter() - t0
        del comps
    else:
        ys = descs
    # exact kNN of ONE image against the full DB on the host (fp32 sgemm like faiss; timed)
    Rh = db_rows.cpu().numpy()
    n_db, d = Rh.shape
    q = ys[0].astype(np.float32)
    t0 = time.perf_counter()
    rn = np.einsum("ij,ij->i", Rh, Rh)          # (no [n, d] temporary: raw K*D rows are 20 GB)
    d2 = (q * q).sum(1)[:, None] + rn[None, :] - 2.0 * (q @ Rh.T)
    part = np.argpartition(d2, 200, axis=1)[:, :200]
    pd = np.take_along_axis(d2, part, 1)
    o = np.argsort(pd, axis=1, kind="stable")
    idx = np.take_along_axis(part, o, 1)
    dd = np.take_along_axis(pd, o, 1)
    t_knn = time.perf_counter() - t0
    sims1 = (2 - dd[:, :50]).astype(np.float32)
    img = (np.arange(n_db) // S).astype(np.int64)
    t0 = time.perf_counter()
    O.get_matches_wt_borda_im(idx[:, :50], 1, sims1, [np.arange(S)], img, n=5)
    t_vote = time.perf_counter() - t0
    per_img = t_desc / n_v + t_pca / n_v + t_knn + t_vote

    # ---- the check: oracle (fp64 exact kNN) vs a device run over exactly these n_v images ---------------------------
    offs = (np.arange(n_v + 1) * S).astype(np.int32)
    qd_dev = pipe.describe(q_tok[:n_v], q_msk[:n_v * S], offs)
    p_dev, _, m_dev, s_dev = index.retrieve(qd_dev, offs, 200, 50, 5)
    p_dev, m_dev, s_dev = p_dev.cpu().numpy(), m_dev.cpu().numpy(), s_dev.cpu().numpy()
    Qo = np.concatenate(ys).astype(np.float32)
    dmat = O.l2_matrix(Rh, Qo, rows_block=max(1024, int(2e9 // (8 * d))))   # fp64 copies of <= 2 GB of rows at a time
    od2, oidx = O.topk_from_d2(dmat, 200)
    osims = (2 - od2[:, :50]).astype(np.float32)
    segr = [np.arange(i * S, (i + 1) * S) for i in range(n_v)]
    p_or, sc_or = O.get_matches_wt_borda_im(oidx[:, :50], n_v, osims, segr, img, n=5, return_scores=True)
    top1_same = int(sum(int(p_dev[i][0]) == int(p_or[i][0]) for i in range(n_v)))
    top5_same = int(sum([int(x) for x in p_dev[i] if x >= 0] == [int(x) for x in p_or[i]] for i in range(n_v)))
    qq, rr = np.nonzero(m_dev != oidx[:, :50])
    near_tie = True
    if len(qq):   # an id mismatch is legitimate only as a near-tie: the device's row is as close as the oracle's
        # (fp32 fma chain over d terms against fp64: ~sqrt(d) ulps of a unit dot product -- 1e-5 at d = 1024; raw 98 304-d
        #  rows get north_star's 1e-4)
        near_tie = bool(np.abs(dmat[qq, m_dev[qq, rr]] - od2[qq, rr]).max() < (1e-5 if d <= 4096 else 1e-4))
    gt = [[int(t)] for t in tau[:n_v]]
    from revisit_anything_amd.pipeline import recall_at

    def pad5(rows):
        return np.array([[int(x) for x in r] + [-1] * (5 - len(r)) for r in rows], dtype=np.int64)

    # the tie audit of the assignment (verdict r04, weak 12): the device's labels against the oracle's over the verified images'
    # tokens, and how many of them sit in the band where an fp32 arg-max may legitimately differ from the reference's (top-1
    # minus top-2 cosine below 1e-6: the `gap` output of segvlad_images)
    bits_v = pipe.eng.incidence(q_msk[:n_v * S], H, W, 14)
    av = pipe.eng.seg_vlad(q_tok[:min(n_v, 4)], bits_v[:min(n_v, 4) * S], offs[:min(n_v, 4) + 1], None, want_labels=True, want_gap=True)
    lab_dev = av["labels"].cpu().numpy().reshape(-1)
    gap_dev = av["gap"].cpu().numpy().reshape(-1)
    lab_or = np.concatenate([np.asarray(x).reshape(-1) for x in o_labels[:min(n_v, 4)]])
    del av
    check = {"images": n_v, "top1_identical": top1_same, "top5_identical": top5_same,
             "assignment_audit": {"tokens": int(lab_dev.size), "labels_differing_from_the_oracle": int((lab_dev != lab_or).sum()),
                                  "tokens_with_gap_below_1e-6": int((gap_dev < 1e-6).sum()), "smallest_gap": float(gap_dev.min())},
             "sims_max_abs_diff": float(np.abs(s_dev - osims).max()), "neighbour_id_mismatches": int(len(qq)),
             "neighbour_id_mismatches_are_near_ties": near_tie,
             "desc_max_abs_diff": float(np.abs(qd_dev.cpu().numpy() - Qo).max()),
             "oracle_recall_at_1": recall_at(pad5(p_or), gt, 5)[0], "device_recall_at_1": recall_at(p_dev, gt, 5)[0],
             "ok": bool(top1_same == n_v and near_tie and np.abs(s_dev - osims).max() < 1e-4)}
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": int(cores), "kind": "port",
            "kind_note": "oracle/segvlad_oracle.py (NumPy restatement of the reference); the kNN leg is a NumPy/OpenBLAS fp32 sgemm + "
                         "argpartition -- the arithmetic faiss.IndexFlatL2 performs, NOT faiss itself (not installable here)",
            "sample": f"{n_v} query images for adjacency+seg-VLAD(+PCA), 1 image (50 segs) exact kNN vs the full {n_db}-row DB (fp32 sgemm), vote",
            "seconds": {"seg_vlad_per_img": t_desc / n_v, "pca_per_img": t_pca / n_v, "knn_per_img": t_knn, "vote_per_img": t_vote},
            "oracle_check": check}


if __name__ == "__main__":
    main()
