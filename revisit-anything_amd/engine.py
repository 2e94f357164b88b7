"""SegVLADEngine: thin object wrapper over the C-ABI (include/segvlad.h) for PyTorch-ROCm host code.

PyTorch is plumbing here: device memory (tensors handed over as raw pointers), the current HIP
stream, and -- in sharded.py -- torch.distributed.  All arithmetic happens in libsegvlad_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import SegVLADDegenerateError, SegVLADError


def _ptr(x) -> int:
    """Raw address of a torch tensor (host or device) or a NumPy array; None -> NULL."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    raise TypeError(f"unsupported buffer type {type(x)}")


def _as(x, dtype_np, dtype_t):
    """Coerce to a contiguous array/tensor of the wanted dtype without moving it between host/device."""
    if isinstance(x, torch.Tensor):
        return x.to(dtype_t).contiguous()
    return np.ascontiguousarray(x, dtype=dtype_np)


class SegVLADEngine:
    """One context per (device, stream user).  Not thread-safe (the C context is not re-entrant)."""

    def __init__(self, device: int | str | torch.device = 0):
        if not torch.cuda.is_available():
            raise SegVLADError("SegVLADEngine needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.device = dev
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.segvlad_create(C.byref(h), dev.index or 0)
        if rc != 0:
            raise SegVLADError(f"segvlad_create(device={dev.index}) failed with {rc}")
        self._h = h
        self.K = self.D = 0
        self.P = self.KD = 0
        self.vocab_generation = 0   # bumped by every set_vocab / pca_set: lets callers cache "my model is resident"
        self.pca_generation = 0
        self._keep = []  # tensors that must outlive async kernels of the last call

    # ---- plumbing ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.segvlad_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.segvlad_last_error(self._h)
            raise SegVLADError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else ''}", code=int(rc))

    def _stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.lib.segvlad_set_stream(self._h, C.c_void_p(s))

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def synchronize(self):
        self._stream()
        self._check(self.lib.segvlad_synchronize(self._h), "synchronize")

    def set_profiling(self, on: bool):
        self.lib.segvlad_set_profiling(self._h, int(on))

    def profile_reset(self):
        self.lib.segvlad_profile_reset(self._h)

    def set_option(self, key: str, value):
        """segvlad_set_option: arithmetic / tuning switches of this context (see include/segvlad.h)."""
        self._check(self.lib.segvlad_set_option(self._h, str(key).encode(), str(value).encode()), f"set_option({key})")

    def hint_query_groups(self, qseg_offsets) -> int:
        """Tell the search how the query rows of the coming batches are grouped (option ``query_group``): when every query
        image brings the same number of rows (<= 64) the exact refinement takes an image's rows as one group -- the bands of
        an image's segments share most of their rows (csrc/refine_group_kernels.hip).  Never changes a result."""
        runs = np.diff(np.asarray(qseg_offsets, dtype=np.int64))
        hint = int(runs[0]) if runs.size and int(runs.min()) == int(runs.max()) and 1 <= int(runs[0]) <= 64 else 0
        if hint != getattr(self, "_group_hint", None):
            self.set_option("query_group", hint)
            self._group_hint = hint
        return hint

    def search_stats(self) -> dict:
        """Statistics of the last search(): levels, filter arithmetic, rows redone on the exact path, list occupancies."""
        v = (C.c_int64 * 13)()
        self._check(self.lib.segvlad_search_stats(self._h, v, 13), "search_stats")
        names = ("levels", "filter", "n_fallback", "cand_max", "cand_sum", "refine_max", "refine_sum", "n_queries", "n_redo",
                 "n_refine2", "grp_groups", "grp_union_sum", "carry_rows")
        d = dict(zip(names, [int(x) for x in v]))
        d["filter"] = {0: "none", 1: "f16", 2: "bf16x3", 3: "fp32"}[d["filter"]]
        return d

    def stage_ms(self, stage: str):
        """(total ms, kernel launches) of the stage since the last profile_reset()."""
        ms, n = C.c_float(), C.c_int()
        self._stream()
        self._check(self.lib.segvlad_stage_ms(self._h, stage.encode(), C.byref(ms), C.byref(n)), f"stage_ms({stage})")
        return ms.value, n.value

    # ---- row-sharded index over several GPUs (segvlad_comm_* / segvlad_search_sharded: RCCL bound at run time) ----------
    def comm_unique_id(self) -> bytes:
        """128-byte RCCL id drawn by ONE rank; hand it to every rank's comm_init over any host channel."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = self.lib.segvlad_comm_unique_id(buf)
        if rc != 0:
            raise SegVLADError(f"comm_unique_id failed ({rc}): RCCL could not be bound (librccl.so; SEGVLAD_RCCL_LIB)", code=int(rc))
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, world: int):
        """Collective: bind an RCCL communicator (world ranks, one GPU each) to this context."""
        if len(uid) != _lib.COMM_ID_BYTES:
            raise ValueError("uid must be the 128 bytes of comm_unique_id()")
        self._stream()
        self._check(self.lib.segvlad_comm_init(self._h, C.c_char_p(uid), int(rank), int(world)), "comm_init")

    def comm_destroy(self):
        self._check(self.lib.segvlad_comm_destroy(self._h), "comm_destroy")

    def comm_info(self) -> dict:
        r, w = C.c_int(), C.c_int()
        o = C.create_string_buffer(256)
        self._check(self.lib.segvlad_comm_info(self._h, C.byref(r), C.byref(w), o, 256), "comm_info")
        return {"rank": r.value, "world": w.value, "rccl": o.value.decode(errors="replace")}

    def allgather_rows(self, x, world: Optional[int] = None) -> torch.Tensor:
        """[n_local, d] fp32 of every rank -> [world * n_local, d] in rank order (equal slices), one RCCL all-gather.
        ``world``: the communicator's size if the caller knows it (saves a C call per gather)."""
        t = _as(x, np.float32, torch.float32)
        n, d = int(t.shape[0]), int(t.shape[1])
        out = self._empty(((int(world) if world else self.comm_info()["world"]) * n, d), torch.float32)
        self._stream()
        self._check(self.lib.segvlad_allgather_rows(self._h, _ptr(t), n, d, _ptr(out)), "allgather_rows")
        self._keep = [t]
        return out

    def search_sharded(self, Q, k: int, id_base: int):
        """Collective: global exact top-k over all ranks' shards, identical on every rank (d2 [nq,k], GLOBAL ids [nq,k])."""
        q = _as(Q, np.float32, torch.float32)
        nq = int(q.shape[0])
        d2 = self._empty((nq, k), torch.float32)
        idx = self._empty((nq, k), torch.int64)
        self._stream()
        self._check(self.lib.segvlad_search_sharded(self._h, _ptr(q), nq, int(k), int(id_base), _ptr(d2), _ptr(idx)), "search_sharded")
        self._keep = [q]
        return d2, idx

    # ---- vocabulary -------------------------------------------------------------------------------
    def set_vocab(self, c_centers):
        c = _as(c_centers, np.float32, torch.float32)
        K, D = c.shape
        self._stream()
        self._check(self.lib.segvlad_set_vocab(self._h, _ptr(c), K, D), "set_vocab")
        if isinstance(c, torch.Tensor) and c.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()
        self.K, self.D = int(K), int(D)
        self.vocab_generation += 1

    # ---- masks ------------------------------------------------------------------------------------
    def incidence(self, masks, H: int, W: int, patch: int = 14) -> torch.Tensor:
        """masks [S,Hm,Wm] bool/uint8 (tensor or ndarray) -> int64 tensor [S, ceil(N/64)] holding u64 bit rows."""
        m = masks.to(torch.uint8).contiguous() if isinstance(masks, torch.Tensor) else np.ascontiguousarray(masks).astype(np.uint8)
        S, Hm, Wm = m.shape
        N = (H // patch) * (W // patch)
        out = self._empty((S, (N + 63) // 64), torch.int64)
        self._stream()
        self._check(self.lib.segvlad_incidence(self._h, _ptr(m), S, Hm, Wm, H, W, patch, _ptr(out)), "incidence")
        self._keep = [m]
        return out

    def incidence_centroids(self, masks, H: int, W: int, patch: int = 14):
        """One pass over the masks: (incidence bit rows [S, ceil(N/64)] as int64, centroids [S,2] fp64 (x, y))."""
        m = masks.to(torch.uint8).contiguous() if isinstance(masks, torch.Tensor) else np.ascontiguousarray(masks).astype(np.uint8)
        S, Hm, Wm = m.shape
        N = (H // patch) * (W // patch)
        out = self._empty((S, (N + 63) // 64), torch.int64)
        cent = self._empty((S, 2), torch.float64)
        self._stream()
        self._check(self.lib.segvlad_incidence_centroids(self._h, _ptr(m), S, Hm, Wm, H, W, patch, _ptr(out), _ptr(cent)),
                    "incidence_centroids")
        self._keep = [m]
        return out, cent

    def mask_centroids(self, masks) -> torch.Tensor:
        m = masks.to(torch.uint8).contiguous() if isinstance(masks, torch.Tensor) else np.ascontiguousarray(masks).astype(np.uint8)
        S, Hm, Wm = m.shape
        out = self._empty((S, 2), torch.float64)
        self._stream()
        self._check(self.lib.segvlad_mask_centroids(self._h, _ptr(m), S, Hm, Wm, _ptr(out)), "mask_centroids")
        self._keep = [m]
        return out

    def adjacency(self, centroids, seg_offsets, order: int, check_empty: bool = False) -> torch.Tensor:
        """centroids [S_tot,2] fp64 -> uint8 buffer with the concatenated per-image [S_b,S_b] (A1^order > 0)
        blocks, computed on the device.  check_empty=True synchronises and raises ValueError on an empty mask
        (the reference's behaviour), and SegVLADDegenerateError when an image holds a non-generic centroid
        configuration (duplicate or exactly co-circular points: Qhull's triangulation is then a matter of its own
        tie-breaking; SegVLADPipeline catches this and uses the reference's Qhull path for that batch)."""
        c = _as(centroids, np.float64, torch.float64)
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        B = len(so) - 1
        total = int(((so[1:] - so[:-1]).astype(np.int64) ** 2).sum())
        out = self._empty((total,), torch.uint8)
        n_bad = (C.c_uint32 * 1)(0)
        self._stream()
        self._check(self.lib.segvlad_adjacency(self._h, _ptr(c), _ptr(so), B, int(order), _ptr(out),
                                               C.cast(n_bad, C.c_void_p) if check_empty else None), "adjacency")
        self._keep = [c]
        if check_empty and n_bad[0] & 0xFFFF:
            raise ValueError(f"{n_bad[0] & 0xFFFF} empty mask(s): centroid undefined")
        if check_empty and n_bad[0] >> 16:
            raise SegVLADDegenerateError(f"adjacency: {n_bad[0] >> 16} image(s) with a degenerate centroid configuration "
                                         "(duplicate or co-circular centroids): the Delaunay triangulation is not unique")
        return out

    def adjacency_flagged(self, centroids, seg_offsets, order: int, device_flags: bool = False):
        """adjacency() plus one flag byte per image: bit 0 = empty mask in the image, bit 1 = non-generic centroid
        configuration (segvlad_adjacency_flagged).  Never raises on either.  The flags come back as a NumPy uint8 [B] (ONE
        B-byte read-back, i.e. a host synchronisation) -- or, with ``device_flags``, as a device tensor the caller reads when it
        suits it (e.g. after it has enqueued the kernels that consume the adjacency)."""
        c = _as(centroids, np.float64, torch.float64)
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        B = len(so) - 1
        total = int(((so[1:] - so[:-1]).astype(np.int64) ** 2).sum())
        out = self._empty((total,), torch.uint8)
        flags = self._empty((max(B, 1),), torch.uint8) if device_flags else np.zeros(max(B, 1), np.uint8)
        self._stream()
        self._check(self.lib.segvlad_adjacency_flagged(self._h, _ptr(c), _ptr(so), B, int(order), _ptr(out), _ptr(flags)),
                    "adjacency_flagged")
        self._keep = [c]
        return out, flags[:B]

    # ---- segment VLAD -----------------------------------------------------------------------------
    def seg_vlad(self, tokens, inc_bits, seg_offsets: Sequence[int], adj=None, want_labels=False, want_gap=False,
                 want_block_norms=False, out: Optional[torch.Tensor] = None):
        """tokens [B,D,N] fp32; inc_bits [S_tot,nw] (int64 storage of u64); seg_offsets [B+1];
        adj: None | uint8 buffer with the concatenated per-image [S_b,S_b] matrices.
        Returns dict(out=[S_tot,K*D] fp32 device tensor, labels?, gap?, block_norms?)."""
        if self.K == 0:
            raise SegVLADError("seg_vlad: set_vocab first")
        t = _as(tokens, np.float32, torch.float32)
        if t.ndim == 2:
            t = t[None]
        B, D, N = t.shape
        if D != self.D:
            raise ValueError(f"tokens have D={D}, vocabulary has D={self.D}")
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        assert so.shape == (B + 1,)
        S_tot = int(so[-1])
        ib = inc_bits if isinstance(inc_bits, torch.Tensor) else np.ascontiguousarray(inc_bits)
        a = None
        if adj is not None:
            a = adj.to(torch.uint8).contiguous() if isinstance(adj, torch.Tensor) else np.ascontiguousarray(adj).astype(np.uint8)
        if out is None:
            out = self._empty((S_tot, self.K * self.D), torch.float32)
        res = {"out": out}
        lab = self._empty((B, N), torch.uint8) if want_labels else None
        gap = self._empty((B, N), torch.float32) if want_gap else None
        bn = self._empty((S_tot, self.K), torch.float32) if want_block_norms else None
        self._stream()
        self._check(self.lib.segvlad_images(self._h, _ptr(t), B, N, _ptr(ib), _ptr(so), _ptr(a), _ptr(out), _ptr(lab),
                                            _ptr(gap), _ptr(bn)), "images")
        self._keep = [t, ib, a]
        if want_labels:
            res["labels"] = lab
        if want_gap:
            res["gap"] = gap
        if want_block_norms:
            res["block_norms"] = bn
        return res

    def seg_vlad_pca(self, tokens, inc_bits, seg_offsets: Sequence[int], adj=None, l2norm: bool = True, want_desc=False,
                     want_labels=False, want_gap=False):
        """Fused seg_vlad + pca_apply (segvlad_images_pca): returns dict(out=[S_tot,P] projected (and normalised)
        descriptors, desc?=[S_tot,K*D], labels?, gap?).  Without want_desc the K*D-wide descriptor never reaches HBM."""
        if self.K == 0:
            raise SegVLADError("seg_vlad_pca: set_vocab first")
        if self.P == 0:
            raise SegVLADError("seg_vlad_pca: pca_set first")
        t = _as(tokens, np.float32, torch.float32)
        if t.ndim == 2:
            t = t[None]
        B, D, N = t.shape
        if D != self.D:
            raise ValueError(f"tokens have D={D}, vocabulary has D={self.D}")
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        assert so.shape == (B + 1,)
        S_tot = int(so[-1])
        ib = inc_bits if isinstance(inc_bits, torch.Tensor) else np.ascontiguousarray(inc_bits)
        a = None
        if adj is not None:
            a = adj.to(torch.uint8).contiguous() if isinstance(adj, torch.Tensor) else np.ascontiguousarray(adj).astype(np.uint8)
        y = self._empty((S_tot, self.P), torch.float32)
        desc = self._empty((S_tot, self.K * self.D), torch.float32) if want_desc else None
        lab = self._empty((B, N), torch.uint8) if want_labels else None
        gap = self._empty((B, N), torch.float32) if want_gap else None
        self._stream()
        self._check(self.lib.segvlad_images_pca(self._h, _ptr(t), B, N, _ptr(ib), _ptr(so), _ptr(a), _ptr(y), int(bool(l2norm)),
                                                _ptr(desc), _ptr(lab), _ptr(gap)), "images_pca")
        self._keep = [t, ib, a]
        res = {"out": y}
        if want_desc:
            res["desc"] = desc
        if want_labels:
            res["labels"] = lab
        if want_gap:
            res["gap"] = gap
        return res

    def describe(self, masks, tokens, seg_offsets: Sequence[int], H: int, W: int, patch: int = 14, order: int = 3, pca: bool = True,
                 l2norm: bool = True, want_desc: bool = False) -> dict:
        """segvlad_describe: the describe stage of a batch in one call -- incidence + centroids + device adjacency on the
        context's side stream beside the token-assignment pass, then seg-VLAD (+ PCA).  Device tensors only.  Returns
        dict(out=[S_tot,P] or [S_tot,K*D], bits, cent, adj, flags [B] uint8 on the device: bit 0 empty mask, bit 1 non-generic
        centroids -- pipeline.py patches flagged images with Qhull), desc? with pca and want_desc."""
        if self.K == 0:
            raise SegVLADError("describe: set_vocab first")
        if pca and self.P == 0:
            raise SegVLADError("describe: pca_set first")
        m = masks if masks.dtype == torch.uint8 else masks.to(torch.uint8)
        m = m.contiguous()
        t = tokens.to(torch.float32).contiguous()
        if t.ndim == 2:
            t = t[None]
        B, D, N = t.shape
        if D != self.D:
            raise ValueError(f"tokens have D={D}, vocabulary has D={self.D}")
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        assert so.shape == (B + 1,)
        S_tot = int(so[-1])
        assert m.shape[0] == S_tot
        sizes = (so[1:] - so[:-1]).astype(np.int64)
        bits = self._empty((S_tot, (N + 63) // 64), torch.int64)
        cent = self._empty((S_tot, 2), torch.float64)
        adj = self._empty((int((sizes * sizes).sum()),), torch.uint8)
        flags = self._empty((B,), torch.uint8)
        y = self._empty((S_tot, self.P), torch.float32) if pca else None
        desc = self._empty((S_tot, self.K * self.D), torch.float32) if (want_desc or not pca) else None
        self._stream()
        self._check(self.lib.segvlad_describe(self._h, _ptr(m), int(m.shape[1]), int(m.shape[2]), int(H), int(W), int(patch), _ptr(t), B, N,
                                              _ptr(so), int(order), _ptr(bits), _ptr(cent), _ptr(adj), _ptr(flags), _ptr(desc), _ptr(y),
                                              int(bool(l2norm))), "describe")
        self._keep = [m, t]
        res = {"out": y if pca else desc, "bits": bits, "cent": cent, "adj": adj, "flags": flags}
        if pca and want_desc:
            res["desc"] = desc
        return res

    # ---- the same stage as begin / flags / end: the caller patches flagged images with Qhull WHILE the assignment pass runs ----
    def describe_begin(self, masks, tokens, seg_offsets: Sequence[int], H: int, W: int, patch: int = 14, order: int = 3,
                       pca: bool = True) -> dict:
        """segvlad_describe_begin: enqueues the mask branch (side stream) and the assignment pass, returns the handle
        describe_flags / describe_end take (it owns the intermediates: bits, cent, adj, flags)."""
        if self.K == 0:
            raise SegVLADError("describe: set_vocab first")
        if pca and self.P == 0:
            raise SegVLADError("describe: pca_set first")
        m = (masks if masks.dtype == torch.uint8 else masks.to(torch.uint8)).contiguous()
        t = tokens.to(torch.float32).contiguous()
        if t.ndim == 2:
            t = t[None]
        B, D, N = t.shape
        if D != self.D:
            raise ValueError(f"tokens have D={D}, vocabulary has D={self.D}")
        so = np.ascontiguousarray(seg_offsets, dtype=np.int32)
        assert so.shape == (B + 1,) and m.shape[0] == int(so[-1])
        S_tot = int(so[-1])
        sizes = (so[1:] - so[:-1]).astype(np.int64)
        h = {"m": m, "t": t, "so": so, "B": B, "N": N, "S_tot": S_tot, "pca": bool(pca), "order": int(order),
             "bits": self._empty((S_tot, (N + 63) // 64), torch.int64), "cent": self._empty((S_tot, 2), torch.float64),
             "adj": self._empty((int((sizes * sizes).sum()),), torch.uint8), "flags": self._empty((B,), torch.uint8)}
        self._stream()
        self._check(self.lib.segvlad_describe_begin(self._h, _ptr(m), int(m.shape[1]), int(m.shape[2]), int(H), int(W), int(patch), _ptr(t),
                                                    B, N, _ptr(so), int(order), _ptr(h["bits"]), _ptr(h["cent"]), _ptr(h["adj"]),
                                                    _ptr(h["flags"]), int(bool(pca))), "describe_begin")
        return h

    def describe_flags(self, h: dict):
        """segvlad_describe_flags: (flags [B] uint8, centroids [S_tot,2] fp64) on the HOST; waits for the mask branch only."""
        flags = np.empty(h["B"], np.uint8)
        cent = np.empty((h["S_tot"], 2), np.float64)
        self._check(self.lib.segvlad_describe_flags(self._h, flags.ctypes.data_as(C.c_void_p), cent.ctypes.data_as(C.c_void_p)),
                    "describe_flags")
        return flags, cent

    def describe_cancel(self, h: dict):
        """Ends a describe_begin without results (the mask branch is joined, the context is usable again)."""
        self.lib.segvlad_describe_end(self._h, _ptr(h["t"]), h["B"], h["N"], _ptr(h["bits"]), _ptr(h["so"]), _ptr(h["adj"]), 0, None, None,
                                      None, None, 0)

    def describe_end(self, h: dict, patch_images=None, patch_blocks=None, l2norm: bool = True, want_desc: bool = False) -> dict:
        """segvlad_describe_end: patch_images (ascending image indices) / patch_blocks (their [S_b,S_b] uint8 adjacency matrices,
        host arrays) replace the device adjacency of those images; then prep -> aggregation (-> projection)."""
        pca = h["pca"]
        y = self._empty((h["S_tot"], self.P), torch.float32) if pca else None
        desc = self._empty((h["S_tot"], self.K * self.D), torch.float32) if (want_desc or not pca) else None
        n_patch = 0 if patch_images is None else len(patch_images)
        pi = pb = None
        if n_patch:
            pi = np.ascontiguousarray(patch_images, dtype=np.int32)
            pb = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.uint8).reshape(-1) for b in patch_blocks]))
        self._stream()
        self._check(self.lib.segvlad_describe_end(self._h, _ptr(h["t"]), h["B"], h["N"], _ptr(h["bits"]), _ptr(h["so"]), _ptr(h["adj"]),
                                                  n_patch, _ptr(pi), _ptr(pb), _ptr(desc), _ptr(y), int(bool(l2norm))), "describe_end")
        res = {"out": y if pca else desc, "bits": h["bits"], "cent": h["cent"], "adj": h["adj"], "flags": h["flags"]}
        if pca and want_desc:
            res["desc"] = desc
        return res

    def kmeans_step(self, tokens, sums: torch.Tensor, counts: torch.Tensor, want_labels: bool = False):
        """segvlad_kmeans_step: with the current centres in the context (set_vocab), ACCUMULATES the per-cluster sums of the normalised
        tokens into ``sums`` [K, D] fp64 and the assignment counts into ``counts`` [K] int64 (device tensors); tokens [B, D, N]."""
        t = _as(tokens, np.float32, torch.float32)
        if isinstance(t, torch.Tensor):
            t = t.contiguous()
        if t.ndim == 2:
            t = t[None]
        B, D, N = t.shape
        if D != self.D:
            raise ValueError(f"tokens have D={D}, vocabulary has D={self.D}")
        if not (isinstance(sums, torch.Tensor) and sums.is_cuda and sums.dtype == torch.float64 and tuple(sums.shape) == (self.K, self.D)
                and sums.is_contiguous() and isinstance(counts, torch.Tensor) and counts.is_cuda and counts.dtype == torch.int64
                and tuple(counts.shape) == (self.K,)):
            raise ValueError("kmeans_step: sums must be a contiguous device fp64 [K, D] tensor, counts a device int64 [K] tensor")
        lab = self._empty((B, N), torch.uint8) if want_labels else None
        self._stream()
        self._check(self.lib.segvlad_kmeans_step(self._h, _ptr(t), B, N, _ptr(sums), _ptr(counts), _ptr(lab)), "kmeans_step")
        self._keep = [t]
        return lab

    def cluster_aggregate(self, num_c: int, res, labels, inc_bits, adj=None) -> torch.Tensor:
        """vlad_matmuls_per_cluster surface: res [N,D] fp32 residuals, labels [N] (< num_c), inc_bits [S,nw]."""
        r = _as(res, np.float32, torch.float32)
        N, D = r.shape
        lab = _as(labels, np.uint8, torch.uint8)
        ib = inc_bits if isinstance(inc_bits, torch.Tensor) else np.ascontiguousarray(inc_bits)
        S = ib.shape[0]
        a = None
        if adj is not None:
            a = adj.to(torch.uint8).contiguous() if isinstance(adj, torch.Tensor) else np.ascontiguousarray(adj).astype(np.uint8)
        out = self._empty((S, num_c * D), torch.float32)
        self._stream()
        self._check(self.lib.segvlad_cluster_aggregate(self._h, num_c, _ptr(r), _ptr(lab), N, D, _ptr(ib), S, _ptr(a), _ptr(out)),
                    "cluster_aggregate")
        self._keep = [r, lab, ib, a]
        return out

    # ---- PCA --------------------------------------------------------------------------------------
    def pca_set(self, mean, components, explained_variance=None, whiten=True):
        comps = _as(components, np.float32, torch.float32)
        P, KD = comps.shape
        mean = None if mean is None else _as(mean, np.float32, torch.float32)
        var = None if explained_variance is None else _as(explained_variance, np.float32, torch.float32)
        self._stream()
        self._check(self.lib.segvlad_pca_set(self._h, _ptr(mean), _ptr(comps), _ptr(var), P, KD, int(bool(whiten))), "pca_set")
        self.P, self.KD = int(P), int(KD)
        self.pca_generation += 1

    def pca_apply(self, X, l2norm=False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _as(X, np.float32, torch.float32)
        n, KD = x.shape
        if KD != self.KD:
            raise ValueError(f"X has {KD} columns, PCA model expects {self.KD}")
        if out is None:
            out = self._empty((n, self.P), torch.float32)
        self._stream()
        self._check(self.lib.segvlad_pca_apply(self._h, _ptr(x), n, _ptr(out), int(l2norm)), "pca_apply")
        self._keep = [x]
        return out

    def normalize_rows(self, X) -> torch.Tensor:
        x = _as(X, np.float32, torch.float32)
        n, d = x.shape
        out = self._empty((n, d), torch.float32)
        self._stream()
        self._check(self.lib.segvlad_normalize_rows(self._h, _ptr(x), n, d, _ptr(out)), "normalize_rows")
        self._keep = [x]
        return out

    # ---- exact kNN --------------------------------------------------------------------------------
    def db_reset(self):
        self._check(self.lib.segvlad_db_reset(self._h), "db_reset")

    def db_add(self, R, img_of_seg=None):
        r = _as(R, np.float32, torch.float32)
        n, d = r.shape
        im = None if img_of_seg is None else _as(img_of_seg, np.int32, torch.int32)
        self._stream()
        self._check(self.lib.segvlad_db_add(self._h, _ptr(r), n, d, _ptr(im)), "db_add")
        if isinstance(r, torch.Tensor) and r.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()  # the index copied the rows: r may now be freed

    def db_size(self):
        n, d = C.c_int64(), C.c_int()
        self.lib.segvlad_db_size(self._h, C.byref(n), C.byref(d))
        return n.value, d.value

    def search(self, Q, k: int):
        q = _as(Q, np.float32, torch.float32)
        nq = q.shape[0]
        d2 = self._empty((nq, k), torch.float32)
        idx = self._empty((nq, k), torch.int64)
        self._stream()
        self._check(self.lib.segvlad_search(self._h, _ptr(q), nq, k, _ptr(d2), _ptr(idx)), "search")
        self._keep = [q]
        return d2, idx

    def merge_topk(self, d2_parts, idx_parts, parts: int, k: int):
        d = _as(d2_parts, np.float32, torch.float32)
        i = _as(idx_parts, np.int64, torch.int64)
        nq = d.shape[0]
        assert d.shape[1] == parts * k and i.shape == d.shape
        od = self._empty((nq, k), torch.float32)
        oi = self._empty((nq, k), torch.int64)
        self._stream()
        self._check(self.lib.segvlad_merge_topk(self._h, _ptr(d), _ptr(i), nq, parts, k, _ptr(od), _ptr(oi)), "merge_topk")
        self._keep = [d, i]
        return od, oi

    def sims_from_d2(self, d2, idx, k_keep: int):
        d = _as(d2, np.float32, torch.float32)
        i = _as(idx, np.int64, torch.int64)
        nq, k_in = d.shape
        s = self._empty((nq, k_keep), torch.float32)
        oi = self._empty((nq, k_keep), torch.int64)
        self._stream()
        self._check(self.lib.segvlad_sims_from_d2(self._h, _ptr(d), _ptr(i), nq, k_in, k_keep, _ptr(s), _ptr(oi)), "sims_from_d2")
        self._keep = [d, i]
        return s, oi

    def minmax(self, sims) -> torch.Tensor:
        s = _as(sims, np.float32, torch.float32)
        out = self._empty((2,), torch.float32)
        self._stream()
        self._check(self.lib.segvlad_minmax(self._h, _ptr(s), s.numel() if isinstance(s, torch.Tensor) else s.size, _ptr(out)), "minmax")
        self._keep = [s]
        return out

    def vote(self, idx, sims, qseg_offsets, n_top=5, mode=_lib.VOTE_WT_BORDA_IM, img_of_seg=None, smin=float("nan"),
             smax=float("nan"), want_scores=True):
        i = _as(idx, np.int64, torch.int64)
        s = None if sims is None else _as(sims, np.float32, torch.float32)
        qo = np.ascontiguousarray(qseg_offsets, dtype=np.int32)
        n_img = len(qo) - 1
        k = i.shape[1]
        im = None if img_of_seg is None else _as(img_of_seg, np.int32, torch.int32)
        n_ref = 0 if im is None else int(im.shape[0])
        pred = self._empty((n_img, n_top), torch.int32)
        sc = self._empty((n_img, n_top), torch.float64) if want_scores else None
        self._stream()
        self._check(self.lib.segvlad_vote(self._h, _ptr(i), _ptr(s), _ptr(im), n_ref, _ptr(qo), n_img, k, smin, smax, n_top,
                                          int(mode), _ptr(pred), _ptr(sc)), "vote")
        self._keep = [i, s, im]
        return pred, sc
