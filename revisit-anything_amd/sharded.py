"""Row-sharded exact segment index over the GPUs of one node (one process per GPU).

The reference is single-process; sharding is the build's own layer:

* the reference-segment rows are split into contiguous blocks, rank r keeps rows
  [row_start[r], row_start[r+1]) and answers queries against its block only;
* every rank sees the full query batch, computes its local top-k (distance, GLOBAL segment id),
  then ONE all_gather (RCCL over xGMI when the backend is "nccl") exchanges the per-shard lists as
  packed 12-byte records {fp32 distance bits, int64 id}: nq * k * 12 B per rank;
* every rank merges the world*k candidates per query segment to the global top-k -- by distance,
  ties by lower global id, i.e. exactly what a single index over all rows returns -- and votes.

The compute is delegated to a backend object with the SegVLADEngine interface (db_reset, db_add,
search, merge_topk, sims_from_d2, vote); the distributed plumbing below is backend-agnostic so
that it can be exercised with gloo on CPU (tests/test_sharded_gloo.py) with a checker backend.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int) -> np.ndarray:
    """Contiguous, balanced row blocks: row_start[r] = floor(r * n / world)."""
    return np.array([(r * n_rows) // world for r in range(world + 1)], dtype=np.int64)


def shard_images(n_images: int, world: int) -> np.ndarray:
    """Contiguous image blocks (a reference image's segments never straddle two ranks)."""
    return np.array([(r * n_images) // world for r in range(world + 1)], dtype=np.int64)


class ShardedSegmentIndex:
    def __init__(self, backend, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                 device: Optional[torch.device] = None, native_comm: bool = False):
        """native_comm: exchange through the C-ABI's own RCCL communicator (segvlad_comm_init / segvlad_search_sharded /
        segvlad_allgather_rows: pack, all-gather, unpack and merge on the context's stream, no torch collective on the data
        path) instead of torch.distributed; the 128-byte communicator id still travels over torch.distributed's host
        channel.  What a C caller of include/segvlad.h gets."""
        self.be = backend
        self.native = bool(native_comm)
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if self.distributed else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if self.distributed else 1)
        self.device = device if device is not None else getattr(backend, "device", torch.device("cpu"))
        self.row_start = np.zeros(self.world + 1, dtype=np.int64)
        self.n_local = 0
        self.img_of_seg_global: Optional[torch.Tensor] = None
        # profile_collectives(True): every collective on the data path is bracketed by a pair of events on the current stream;
        # collective_ms() sums them per name (the N > 1 bench line carries them per rank, with the bytes of collective_bytes())
        self._coll_events = None

    # ---- what travels, and how long it takes --------------------------------------------------------------------------
    def profile_collectives(self, on: bool = True):
        self._coll_events = [] if on else None

    def _timed(self, name: str, fn):
        if self._coll_events is None or self.device.type != "cuda":
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self._coll_events.append((name, e0, e1))
        return out

    def collective_ms(self) -> dict:
        """{collective name: summed milliseconds since profile_collectives(True)} (synchronises on the recorded events)."""
        out: dict = {}
        for name, e0, e1 in self._coll_events or []:
            e1.synchronize()
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        if self._coll_events is not None:
            self._coll_events = []
        return out

    @staticmethod
    def collective_bytes(world: int, nq: int, k: int, d: int, rows_per_rank: Sequence[int]) -> dict:
        """Bytes every rank SENDS / RECEIVES per retrieve(): the all-gather of the query descriptors (fp32 rows; ragged slices are
        padded to the longest) and the all-gather of the per-shard top-k as packed 12-byte records."""
        mx = max(int(r) for r in rows_per_rank) if len(rows_per_rank) else 0
        return {"query_rows_allgather_send": mx * d * 4, "query_rows_allgather_recv": world * mx * d * 4,
                "topk_records_allgather_send": nq * k * 12, "topk_records_allgather_recv": world * nq * k * 12}

    # ---- build --------------------------------------------------------------------------------------
    def build(self, local_rows, local_img_of_seg):
        """local_rows [n_local, d] (this rank's block, in global row order), local_img_of_seg [n_local]
        with GLOBAL reference-image ids.  Exchanges the block sizes and the (small) segment->image map."""
        self.be.db_reset()
        n_local = int(local_rows.shape[0])
        counts = [n_local]
        if self.world > 1:
            t = torch.tensor([n_local], dtype=torch.int64, device=self.device)
            allc = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(allc, t, group=self.group)
            counts = [int(c.item()) for c in allc]
        self.row_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.n_local = n_local
        img = torch.as_tensor(np.asarray(local_img_of_seg) if not isinstance(local_img_of_seg, torch.Tensor) else local_img_of_seg)
        img = img.to(torch.int32).to(self.device)
        if self.world > 1:
            mx = max(counts)
            pad = torch.full((mx,), -1, dtype=torch.int32, device=self.device)
            pad[:n_local] = img
            parts = [torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(parts, pad, group=self.group)
            img = torch.cat([p[:c] for p, c in zip(parts, counts)])
        self.img_of_seg_global = img.contiguous()
        if n_local:
            self.be.db_add(local_rows, None)
        if self.native:
            uid = [self.be.comm_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(uid, src=0, group=self.group)
            self.be.comm_init(uid[0], self.rank, self.world)

    @property
    def n_total(self) -> int:
        return int(self.row_start[-1])

    # ---- query descriptors: every rank describes a slice of the query images, all ranks need all rows -----------
    def gather_rows(self, local_rows: torch.Tensor, rows_per_rank: Sequence[int]) -> torch.Tensor:
        """Concatenation, in rank order, of every rank's ``local_rows`` ([rows_per_rank[r], d]) -- one all_gather
        (RCCL over xGMI with the nccl backend).  Slices may be ragged: short ones are padded to the longest for the
        collective and trimmed afterwards."""
        rows_per_rank = [int(r) for r in rows_per_rank]
        if len(rows_per_rank) != self.world:
            raise ValueError("rows_per_rank must have one entry per rank")
        if int(local_rows.shape[0]) != rows_per_rank[self.rank]:
            raise ValueError(f"rank {self.rank} holds {int(local_rows.shape[0])} rows, rows_per_rank says {rows_per_rank[self.rank]}")
        if self.world == 1:
            return local_rows
        mx = max(rows_per_rank)
        x = local_rows.contiguous()
        if self.native and min(rows_per_rank) == mx and x.dim() == 2 and x.dtype == torch.float32:
            # (the C-ABI gathers fp32 rows; any other type keeps torch.distributed's collective and its dtype)
            return self._timed("query_rows_allgather", lambda: self.be.allgather_rows(x, world=self.world))
        if min(rows_per_rank) == mx:   # equal slices (the usual case): ONE all_gather_into_tensor, no padding, no trimming
            out = torch.empty((self.world * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            self._timed("query_rows_allgather", lambda: dist.all_gather_into_tensor(out, x, group=self.group))
            return out
        if x.shape[0] < mx:
            x = torch.cat([x, x.new_zeros((mx - x.shape[0],) + tuple(x.shape[1:]))])
        parts = [torch.empty_like(x) for _ in range(self.world)]
        self._timed("query_rows_allgather", lambda: dist.all_gather(parts, x, group=self.group))
        return torch.cat([p[:n] for p, n in zip(parts, rows_per_rank)])

    # ---- query --------------------------------------------------------------------------------------
    def search(self, Q, k: int, k_local: Optional[int] = None):
        """Global top-k over all shards: (d2 [nq,k] ascending, idx [nq,k] GLOBAL ids), identical on every rank.
        k_local >= k: search depth used on each shard before the exchange (only the first k columns travel:
        the global top-k is contained in the union of the per-shard top-k lists)."""
        nq = int(Q.shape[0])
        if self.native:   # local search, packed all-gather, merge: one C-ABI call on the context's stream
            # (its all-gather sits inside the C-ABI call: the event pair brackets local search + exchange + merge)
            return self._timed("native_search_sharded", lambda: self.be.search_sharded(Q, k, int(self.row_start[self.rank])))
        if self.world == 1 and self.n_local and (k_local is None or k_local <= k):
            # a single index: the engine's result as it is (no id offset, no slice: four small torch kernels and their launch
            # gaps per call otherwise -- 0.1 ms of a 26-ms step)
            d2, idx = self.be.search(Q, k)
            return torch.as_tensor(d2).to(self.device), torch.as_tensor(idx).to(self.device)
        if self.n_local:
            d2, idx = self.be.search(Q, max(k, k_local or k))
            d2 = torch.as_tensor(d2).to(self.device)[:, :k].contiguous()
            idx = torch.as_tensor(idx).to(self.device)[:, :k].contiguous()
            idx = torch.where(idx >= 0, idx + int(self.row_start[self.rank]), idx)
        else:
            d2 = torch.full((nq, k), float("inf"), dtype=torch.float32, device=self.device)
            idx = torch.full((nq, k), -1, dtype=torch.int64, device=self.device)
        if self.world == 1:
            return d2, idx
        d2c, idc = self.exchange_topk(d2, idx)
        md, mi = self.be.merge_topk(d2c, idc, self.world, k)
        return torch.as_tensor(md), torch.as_tensor(mi)

    def exchange_topk(self, d2: torch.Tensor, idx: torch.Tensor):
        """ONE collective for the per-shard lists: (d2 fp32, global id int64) travel as a packed 12-byte record
        {fp32 bits, id low, id high}; returns ([nq, world*k] d2, [nq, world*k] ids), shard-major within a row.
        (Also valid at world size 1, which is how the RCCL path is smoke-tested on a 1-GPU box.)"""
        nq, k = int(d2.shape[0]), int(d2.shape[1])
        rec = self.pack_topk_records(d2, idx)
        allrec = torch.empty((self.world * nq, k, 3), dtype=torch.int32, device=self.device)   # rank-major concatenation
        self._timed("topk_records_allgather", lambda: dist.all_gather_into_tensor(allrec, rec, group=self.group))
        return self.unpack_topk_records(allrec, self.world)

    @staticmethod
    def pack_topk_records(d2: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """[nq, k] (fp32 d2, int64 global id) -> [nq, k, 3] int32 records {fp32 bits, id low, id high}, contiguous: what a rank
        hands to the all-gather."""
        if d2.dtype != torch.float32 or idx.dtype != torch.int64 or d2.shape != idx.shape or d2.dim() != 2:
            raise ValueError(f"pack_topk_records: want fp32 / int64 [nq, k] pairs, got {d2.dtype}{tuple(d2.shape)} / {idx.dtype}{tuple(idx.shape)}")
        nq, k = int(d2.shape[0]), int(d2.shape[1])
        rec = torch.empty((nq, k, 3), dtype=torch.int32, device=d2.device)
        rec[:, :, 0] = d2.contiguous().view(torch.int32)
        rec[:, :, 1:] = idx.contiguous().view(torch.int32).view(nq, k, 2)
        return rec

    @staticmethod
    def unpack_topk_records(allrec: torch.Tensor, world: int):
        """The all-gather's output -- the ranks' record blocks concatenated in rank order, [world * nq, k, 3] int32 -- as
        ([nq, world * k] fp32 d2, [nq, world * k] int64 ids), shard-major within a row (what merge_topk takes)."""
        if allrec.dtype != torch.int32 or allrec.dim() != 3 or allrec.shape[2] != 3 or allrec.shape[0] % world or not allrec.is_contiguous():
            raise ValueError(f"unpack_topk_records: want a contiguous int32 [world * nq, k, 3] block, got {allrec.dtype}{tuple(allrec.shape)}")
        nq, k = int(allrec.shape[0]) // world, int(allrec.shape[1])
        r = allrec.view(world, nq, k, 3).permute(1, 0, 2, 3)                                    # [nq, world, k, 3]
        d2c = r[..., 0].contiguous().view(torch.float32).view(nq, world * k)
        idc = r[..., 1:].contiguous().view(torch.int64).view(nq, world * k)
        return d2c, idc

    def retrieve(self, Q, qseg_offsets: Sequence[int], k_search: int = 200, k_vote: int = 50, n_top: int = 5, mode: int = 0,
                 want_scores: bool = False, vote_depth_only: bool = False):
        """search -> keep k_vote, 2-d^2 -> vote with the global segment->image map.  The global min/max of the
        vote (func_vpr.py:212-213) is taken over the merged (global) similarities, so it needs no extra collective.
        vote_depth_only: a caller that does not keep the k_search-wide lists (place_rec_main.py:61-75 pickles them only under
        save_results) may have a single index search k_vote deep as well -- same first k_vote columns, same votes."""
        if hasattr(self.be, "hint_query_groups"):   # an image's rows as one group of the exact refinement (same results)
            self.be.hint_query_groups(qseg_offsets)
        if self.world > 1 or vote_depth_only:
            # The reference searches 200 and keeps 50 (place_rec_main.py:56,78).  The global top-50 is contained in the union
            # of the per-shard top-50 lists, and an exact search returns the same first 50 rows whatever depth it is asked
            # for (ascending by (distance, id)): every shard searches -- and refines -- k_vote deep, not k_search, and only
            # those columns travel.  (k_search only matters to a single index, which mirrors the reference's call.)
            d2, idx = self.search(Q, k_vote)
        else:
            d2, idx = self.search(Q, k_search)
        sims, m = self.be.sims_from_d2(d2, idx, k_vote)
        pred, sc = self.be.vote(m, sims, np.asarray(qseg_offsets, dtype=np.int32), n_top=n_top, mode=mode,
                                img_of_seg=self.img_of_seg_global, want_scores=want_scores)
        return pred, sc, m, sims


def dry_run_collectives(backend, device, worlds: Sequence[int] = (2, 3, 4, 8), nq: int = 96, k: int = 50, d: int = 64, seed: int = 0) -> dict:
    """Everything of the multi-GPU exchange that can be checked WITHOUT the other ranks (VERDICT r05 next #5b): the first real
    N > 1 run must not be the first time these shapes exist.

    For every world size W in `worlds`: a planted index is cut into W image-aligned shards, every shard is searched on `backend`
    (one after the other, on this one device), each rank's operands of the two collectives are built exactly as the ranks would
    build them -- `pack_topk_records`, the (padded) query-row slices -- and CHECKED against what `all_gather_into_tensor` demands
    (dtype, contiguity, output numel == W x input numel); the collective itself is replaced by the concatenation in rank order
    that it is defined to produce; `unpack_topk_records` + `merge_topk` must then return the single index's (d2, ids) bit for bit.
    At W = 1 the real collectives run when a process group is initialised (the nccl = RCCL backend on a GPU box).
    Returns a report; raises AssertionError / ValueError on the first violation."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    n_img, S = 64, 23
    R = torch.nn.functional.normalize(torch.randn(n_img * S, d, generator=g), dim=1).to(device)
    Q = torch.nn.functional.normalize(R[torch.randint(0, n_img * S, (nq,), generator=g).to(device)] + 0.05 * torch.randn(nq, d, generator=g).to(device), dim=1)
    backend.db_reset()
    backend.db_add(R, None)
    d2_ref, idx_ref = (torch.as_tensor(t).to(device) for t in backend.search(Q, k))
    report = {"worlds": {}, "nq": nq, "k": k, "d": d, "rows": n_img * S}
    for W in worlds:
        ib = shard_images(n_img, W)
        recs, row_slices = [], []
        qb = shard_images(nq, W)                                      # the ranks' slices of the query rows (ragged unless W | nq)
        rows_per_rank = [int(qb[r + 1] - qb[r]) for r in range(W)]
        mx = max(rows_per_rank)
        for r in range(W):
            lo, hi = int(ib[r]) * S, int(ib[r + 1]) * S
            backend.db_reset()
            if hi > lo:
                backend.db_add(R[lo:hi].contiguous(), None)
                dd, ii = (torch.as_tensor(t).to(device) for t in backend.search(Q, k))
                ii = torch.where(ii >= 0, ii + lo, ii)
            else:
                dd = torch.full((nq, k), float("inf"), dtype=torch.float32, device=device)
                ii = torch.full((nq, k), -1, dtype=torch.int64, device=device)
            rec = ShardedSegmentIndex.pack_topk_records(dd, ii)
            assert rec.dtype == torch.int32 and rec.is_contiguous() and tuple(rec.shape) == (nq, k, 3), (W, r, rec.shape)
            recs.append(rec)
            x = Q[int(qb[r]):int(qb[r + 1])].contiguous()
            if x.shape[0] < mx:
                x = torch.cat([x, x.new_zeros((mx - x.shape[0], d))])
            assert x.is_contiguous() and x.dtype == torch.float32 and tuple(x.shape) == (mx, d)
            row_slices.append(x)
        allrec = torch.cat(recs)                                      # = all_gather_into_tensor's output: rank-major concatenation
        assert allrec.numel() == W * recs[0].numel() and allrec.is_contiguous()
        d2c, idc = ShardedSegmentIndex.unpack_topk_records(allrec, W)
        assert tuple(d2c.shape) == (nq, W * k) and d2c.dtype == torch.float32 and idc.dtype == torch.int64 and d2c.is_contiguous() and idc.is_contiguous()
        md, mi = (torch.as_tensor(t).to(device) for t in backend.merge_topk(d2c, idc, W, k))
        assert torch.equal(mi, idx_ref) and torch.equal(md, d2_ref), f"world {W}: the merged shards differ from the single index"
        gathered = torch.cat([p[:n] for p, n in zip(row_slices, rows_per_rank)])
        assert torch.equal(gathered, Q), f"world {W}: the padded row gather does not reproduce the query rows"
        report["worlds"][str(W)] = {"ok": True, "rows_per_rank": rows_per_rank,
                                    "bytes": ShardedSegmentIndex.collective_bytes(W, nq, k, d, rows_per_rank)}
    backend.db_reset()
    # world 1 through the REAL collectives (whatever backend the process group has: nccl on the GPU box)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() == 1:
        index = ShardedSegmentIndex(backend, rank=0, world=1, device=device)
        index.build(R, torch.arange(n_img, dtype=torch.int32).repeat_interleave(S))
        d2c, idc = index.exchange_topk(d2_ref, idx_ref)
        assert torch.equal(d2c, d2_ref) and torch.equal(idc, idx_ref)
        out = torch.empty_like(Q)
        dist.all_gather_into_tensor(out, Q.contiguous())
        assert torch.equal(out, Q)
        t = torch.tensor([1.0], device=device)
        dist.all_reduce(t)
        report["world_1_process_group"] = {"backend": dist.get_backend(), "ok": True}
        backend.db_reset()
    return report
