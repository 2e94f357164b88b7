#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own function bodies.  Runs ONLY in the build
container (needs /root/reference); the outputs are committed under tests/golden/ as data.

Recipe: ast-parse func_vpr.py / place_rec_main.py, keep only the named
top-level FunctionDefs, exec them in a namespace that holds numpy/torch/scipy, and reroute the
hard-coded 'cuda' device strings to 'cpu'.  No reference source is copied anywhere: the functions
are executed where they lie and only their inputs' seeds + outputs are stored.

Inputs are regenerated from seeds by revisit_anything_amd.synth in the tests, so fixtures hold
only parameters and expected outputs (plus small masks where needed).
"""
import ast
import os
import pickle
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F
from scipy.spatial import Delaunay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from revisit_anything_amd import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def extract(path, names, ns):
    body = [n for n in ast.parse(open(path).read()).body if isinstance(n, ast.FunctionDef) and n.name in names]
    # keep the LAST definition of duplicated names (python semantics), error if any is missing
    assert {n.name for n in body} == set(names), set(names) - {n.name for n in body}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)


def patch_cuda_to_cpu():
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = ["cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a]
        k = {kk: ("cpu" if kk == "device" and isinstance(v, str) and v.startswith("cuda") else v) for kk, v in k.items()}
        return _to(self, *a, **k)

    torch.Tensor.to = to
    _zeros = torch.zeros

    def zeros(*a, **k):
        k = {kk: ("cpu" if kk == "device" and isinstance(v, str) and v.startswith("cuda") else v) for kk, v in k.items()}
        return _zeros(*a, **k)

    torch.zeros = zeros


class NumpyIndexFlatL2:
    """faiss.IndexFlatL2 stand-in (faiss is not installable here): fp32 inputs, exact squared L2
    evaluated in fp64 and rounded to fp32, ascending, stable (ties -> lower id)."""

    def __init__(self, d):
        self.d = d
        self.x = None

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        self.x = x if self.x is None else np.concatenate([self.x, x])

    def search(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32).astype(np.float64)
        r = self.x.astype(np.float64)
        d2 = ((q * q).sum(1)[:, None] + (r * r).sum(1)[None, :] - 2 * q @ r.T).astype(np.float32)
        o = np.argsort(d2, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(d2, o, 1), o.astype(np.int64)


def load_ref():
    fv = types.ModuleType("func_vpr")
    fv.__dict__.update(np=np, torch=torch, F=F, time=time, Delaunay=Delaunay)
    extract(f"{REF}/func_vpr.py",
            {"first_k_unique_indices", "weighted_borda_count", "get_matches", "calc_recall", "normalizeFeat",
             "vlad_single", "vlad_matmuls_per_cluster", "seg_vlad_gpu_single_img", "seg_vlad_gpu_single",
             "getNbrsDelaunay", "nbrMasksAGGFastSingle", "getIdxSingleFast"}, fv.__dict__)
    prm = types.ModuleType("place_rec_main")
    prm.__dict__.update(np=np, torch=torch, func_vpr=fv, os=os, pickle=pickle,
                        faiss=types.SimpleNamespace(IndexFlatL2=NumpyIndexFlatL2))
    extract(f"{REF}/place_rec_main.py", {"recall_segloc"}, prm.__dict__)
    patch_cuda_to_cpu()
    return fv, prm


def ref_ind_matrix(H, W):
    """place_rec_main.py:187-194 executed literally (the double python loop) for small H, W."""
    dh, dw = H // 14, W // 14
    idx_matrix = np.empty((H, W, 2)).astype("int32")
    for i in range(H):
        for j in range(W):
            idx_matrix[i, j] = np.array([np.clip(i // 14, 0, dh - 1), np.clip(j // 14, 0, dw - 1)])
    ind_matrix = np.ravel_multi_index(idx_matrix.reshape(-1, 2).T, (dh, dw))
    return idx_matrix, ind_matrix


def capture_vlad(fv):
    """Wrap vlad_matmuls_per_cluster so the internal (masks, labels) of vlad_single are recorded."""
    rec = {}
    orig = fv.vlad_matmuls_per_cluster

    def wrapper(num_c, masks, res, clus_labels, adjMat=None, device="cuda"):
        rec["inc"] = masks.bool().numpy().copy()
        rec["labels"] = clus_labels.numpy().copy()
        rec["res"] = res.numpy().copy()
        return orig(num_c, masks, res, clus_labels, adjMat=adjMat, device="cpu")

    fv.vlad_matmuls_per_cluster = wrapper
    return rec


def run_seg_vlad(fv, rec, tokens_dn, masks, C, H, W, adj, D):
    dh, dw = H // 14, W // 14
    idx, ind = ref_ind_matrix(H, W)
    dino = torch.from_numpy(tokens_dn.reshape(1, D, dh, dw))
    cfg = {"desired_height": H, "desired_width": W}
    gd = fv.seg_vlad_gpu_single_img(torch.tensor(ind), idx, dino, "k", [m for m in masks], torch.from_numpy(C), cfg,
                                    desc_dim=D, adj_mat=None if adj is None else adj)
    return gd.numpy(), rec["inc"].copy(), rec["labels"].copy()


def main():
    os.makedirs(OUT, exist_ok=True)
    fv, prm = load_ref()
    rec = capture_vlad(fv)

    # ---- shipped vocabularies (data files of the reference) -------------------------------------
    voc = torch.load(f"{REF}/cache/vocabulary/dinov2_vitg14/l31_value_c32/indoor/c_centers.pt").numpy()
    np.save(f"{OUT}/vocab_indoor_k32_d1536.npy", voc.astype(np.float32))
    vnv = [d for d in sorted(os.listdir(f"{REF}/cache/vocabulary/dinov2_vitg14/l31_value_c32")) if "NV" in d][0]
    vocnv = torch.load(f"{REF}/cache/vocabulary/dinov2_vitg14/l31_value_c32/{vnv}/c_centers.pt").numpy()
    np.save(f"{OUT}/vocab_nv_k32_d768.npy", vocnv.astype(np.float32))
    print("vocab", voc.shape, vocnv.shape, vnv)

    # ---- pixel->token map ------------------------------------------------------------------------
    pm = {}
    for (H, W) in [(84, 112), (100, 130), (480, 640)]:
        idx, ind = ref_ind_matrix(H, W)
        pm[f"ind_{H}_{W}"] = ind.astype(np.int64)
    np.savez_compressed(f"{OUT}/pixel_map.npz", **pm)

    # ---- vlad_tiny: D=32 K=8 N=6x8 S=6, 2x upsample, orders 0/1/3 ----------------------------------
    D, K, H, W, S = 32, 8, 84, 112, 6
    C = synth.make_vocab(K, D, seed=1001)
    tok = synth.make_tokens(C, (H // 14) * (W // 14), seed=2001, noise=0.3)
    masks = synth.make_masks(S, H // 2, W // 2, seed=2101, hmin=4, hmax=20, wmin=4, wmax=30)
    tiny = dict(D=D, K=K, H=H, W=W, S=S, masks=masks)
    for order in (0, 1, 3):
        adj = fv.nbrMasksAGGFastSingle([m for m in masks], order) if order else None
        gd, inc, lab = run_seg_vlad(fv, rec, tok, masks, C, H, W, adj, D)
        assert gd.shape == (S, 32 * D) and np.all(gd[:, K * D:] == 0)
        tiny[f"vlad_o{order}"] = gd[:, :K * D]
        tiny[f"adj_o{order}"] = np.eye(S, dtype=bool) if adj is None else adj.numpy()
        tiny["inc"], tiny["labels"] = inc, lab
    np.savez_compressed(f"{OUT}/vlad_tiny.npz", **tiny)

    # ---- vlad_ref_shape: REF geometry, real indoor vocabulary, S=50, order 3 ----------------------
    D, K, H, W, S = 1536, 32, 480, 640, 50
    tok = synth.make_tokens(voc, 34 * 45, seed=2002)
    masks = synth.make_masks(S, 240, 320, seed=2102)
    adj = fv.nbrMasksAGGFastSingle([m for m in masks], 3)
    t0 = time.time()
    gd, inc, lab = run_seg_vlad(fv, rec, tok, masks, voc, H, W, adj, D)
    print("ref-shape seg_vlad (reference body, cpu) %.3fs" % (time.time() - t0), gd.shape, gd.dtype)
    G = np.random.Generator(np.random.PCG64(777)).standard_normal((K * D, 16))
    np.savez_compressed(f"{OUT}/vlad_ref_shape.npz", S=S, adj=adj.numpy(), labels=lab.astype(np.uint8),
                        inc=np.packbits(inc, axis=1), proj=gd @ G, sub=gd[:, ::61], head=gd[:, :256], tail=gd[:, -256:])
    # adversarial tokens (near-tied assignments) at reduced N: labels + descriptor subsample
    tokA = synth.make_tokens(voc, 34 * 45, seed=2003, adversarial=True)
    gdA, incA, labA = run_seg_vlad(fv, rec, tokA, masks, voc, H, W, None, D)
    np.savez_compressed(f"{OUT}/vlad_ref_shape_adv.npz", labels=labA.astype(np.uint8), sub=gdA[:, ::61], proj=gdA @ G)

    # ---- K=64 through the only K-parametric entry (vlad_matmuls_per_cluster) -----------------------
    D, K, S, N = 32, 64, 66, 20 * 15
    C64 = synth.make_vocab(K, D, seed=1003)
    tok = synth.make_tokens(C64, N, seed=2004, noise=0.2)
    r = np.random.Generator(np.random.PCG64(2104))
    inc64 = r.random((S, N)) < 0.15
    inc64[3] = False  # a segment that covers no token
    adj64 = (r.random((S, S)) < 0.05) | np.eye(S, dtype=bool)
    xn = F.normalize(torch.from_numpy(tok.T.copy()), dim=1)
    cn = F.normalize(torch.from_numpy(C64), dim=1)
    lab = torch.argmax(xn @ cn.T, dim=1)
    res = xn - torch.from_numpy(C64)[lab]
    out64, _ = fv.vlad_matmuls_per_cluster(K, torch.from_numpy(inc64).double(), res.double(), lab,
                                           adjMat=torch.from_numpy(adj64).double())
    np.savez_compressed(f"{OUT}/vlad_k64.npz", D=D, K=K, S=S, N=N, inc=inc64, adj=adj64, labels=lab.numpy(),
                        vlad=out64.numpy())

    # ---- bench shape: K=64, D=1536, N=1530, S=50, order 3 through the only K-parametric entry --------------
    # (vlad_single hard-codes 32 clusters, func_vpr.py:1142; the incidence is captured from a K=32 run of the
    #  reference's seg_vlad_gpu_single_img over the same masks)
    D, K, H, W, S = 1536, 64, 480, 640, 50
    C64b = synth.make_vocab(K, D, seed=1000)
    tokb = synth.make_tokens(C64b, 34 * 45, seed=2005)
    masksb = synth.make_masks(S, 240, 320, seed=2105)
    adjb = fv.nbrMasksAGGFastSingle([m for m in masksb], 3)
    _, incb, _ = run_seg_vlad(fv, rec, tokb, masksb, voc, H, W, None, D)
    xn = F.normalize(torch.from_numpy(tokb.T.copy()), dim=1)
    cn = F.normalize(torch.from_numpy(C64b), dim=1)
    labb = torch.argmax(xn @ cn.T, dim=1)
    resb = xn - torch.from_numpy(C64b)[labb]
    outb, _ = fv.vlad_matmuls_per_cluster(K, torch.from_numpy(incb).double(), resb.double(), labb, adjMat=adjb.double())
    outb = outb.numpy()
    Gb = np.random.Generator(np.random.PCG64(778)).standard_normal((K * D, 16))
    np.savez_compressed(f"{OUT}/vlad_bench_shape.npz", S=S, K=K, D=D, adj=adjb.numpy(), labels=labb.numpy().astype(np.uint8),
                        inc=np.packbits(incb, axis=1), proj=outb @ Gb, sub=outb[:, ::127], head=outb[:, :256], tail=outb[:, -256:])
    print("bench-shape K=64 D=1536 descriptor", outb.shape)

    # ---- VPAir geometry (place_rec_global_config.py:97-111): 800x600 image, masks 300x400, N = 42*57 = 2394 -------
    D, K, H, W, S = 1536, 32, 600, 800, 50
    tokv = synth.make_tokens(voc, 42 * 57, seed=2006)
    masksv = synth.make_masks(S, 300, 400, seed=2106, hmax=75, wmax=100)
    adjv = fv.nbrMasksAGGFastSingle([m for m in masksv], 3)
    gdv, incv, labv = run_seg_vlad(fv, rec, tokv, masksv, voc, H, W, adjv, D)
    Gv = np.random.Generator(np.random.PCG64(779)).standard_normal((K * D, 16))
    # PCA-whitened 512-d descriptors (BASELINE config 5): sklearn's transform given a (synthetic, seeded) model
    from sklearn.decomposition import PCA as _PCA
    pm_mean, pm_comps, pm_var = synth.make_pca_model(K * D, 512, seed=5001)
    pv = _PCA(n_components=512, whiten=True)
    pv.mean_, pv.components_, pv.explained_variance_ = pm_mean.astype(np.float64), pm_comps.astype(np.float64), pm_var.astype(np.float64)
    yv = pv.transform(gdv)
    np.savez_compressed(f"{OUT}/vlad_vpair_shape.npz", S=S, adj=adjv.numpy(), labels=labv.astype(np.uint8),
                        inc=np.packbits(incv, axis=1), proj=gdv @ Gv, sub=gdv[:, ::61], head=gdv[:, :256], tail=gdv[:, -256:],
                        pca512=yv)
    print("VPAir-shape descriptor", gdv.shape, "pca512", yv.shape)

    # ---- AnyLoc global VLAD (f5): the reference's VLAD.generate + get_recall bodies -------------------------------
    # VLAD.generate / generate_res_vec are extracted from utilities.py (class methods); its kmeans member is
    # fast_pytorch_kmeans (third party, absent): a stand-in with the published cosine `predict` (arg-max of the
    # similarity between normalised points and normalised centroids).
    import einops as ein
    ucls = [n for n in ast.parse(open(f"{REF}/utilities.py").read()).body if isinstance(n, ast.ClassDef) and n.name == "VLAD"][0]
    meths = [m for m in ucls.body if isinstance(m, ast.FunctionDef) and m.name in ("generate", "generate_res_vec", "can_use_cache_vlad")]
    uns = dict(np=np, torch=torch, F=F, ein=ein, os=os, Union=__import__("typing").Union, List=__import__("typing").List)
    exec(compile(ast.Module(body=meths, type_ignores=[]), f"{REF}/utilities.py", "exec"), uns)

    class _KM:
        def __init__(self, c):
            self.c = F.normalize(c, dim=1)

        def predict(self, x):
            return torch.argmax(F.normalize(x, dim=1) @ self.c.T, dim=1)

    vobj = types.SimpleNamespace(num_clusters=32, desc_dim=1536, intra_norm=True, norm_descs=True, vlad_mode="hard", cache_dir=None,
                                 c_centers=torch.from_numpy(voc), kmeans=_KM(torch.from_numpy(voc)))
    vobj.can_use_cache_vlad = lambda: False
    vobj.generate_res_vec = lambda q, cid=None: uns["generate_res_vec"](vobj, q, cid)
    av = {}
    for j, seed in enumerate((2010, 2011)):
        tk = synth.make_tokens(voc, 34 * 45, seed=seed, noise=0.2)
        xq = F.normalize(torch.from_numpy(tk.reshape(1, 1536, -1)), dim=1).permute(0, 2, 1).squeeze()     # aggFt :941-943
        gdv = uns["generate"](vobj, xq).numpy()
        av[f"vlad{j}_sub"] = gdv[::37].astype(np.float64)
        av[f"vlad{j}_proj"] = gdv.astype(np.float64) @ np.random.Generator(np.random.PCG64(780)).standard_normal((32 * 1536, 8))
    from sklearn.neighbors import KDTree
    fv.__dict__["KDTree"] = KDTree
    extract(f"{REF}/func_vpr.py", {"get_recall"}, fv.__dict__)
    rr = np.random.Generator(np.random.PCG64(781))
    dbv = rr.standard_normal((40, 64)); dbv /= np.linalg.norm(dbv, axis=1, keepdims=True)
    qv = dbv[rr.integers(0, 40, 15)] + 0.4 * rr.standard_normal((15, 64)); qv /= np.linalg.norm(qv, axis=1, keepdims=True)
    gta = [[int(i)] for i in rr.integers(0, 40, 15)]
    gta[3] = []
    for i in (0, 1, 2, 5, 8):
        gta[i] = [int(np.argmin(((dbv - qv[i]) ** 2).sum(1)))]
    rec_a, match_a = fv.get_recall(dbv.astype(np.float32), qv.astype(np.float32), gta, k=5)
    av.update(db=dbv.astype(np.float32), q=qv.astype(np.float32), gt=np.array([g[0] if g else -1 for g in gta]), recall=np.asarray(rec_a),
              ids=np.stack([m["img_id_r"] for m in match_a]))
    np.savez_compressed(f"{OUT}/anyloc_cases.npz", **av)
    print("anyloc recall", rec_a)

    # ---- incidence cases (captured mask_idx) --------------------------------------------------------
    ic = {}
    cases = [("same", 112, 140, 112, 140), ("x2", 60, 80, 120, 160), ("x2clip", 63, 77, 126, 154),
             ("nonint", 50, 70, 126, 150), ("down", 200, 260, 100, 130)]
    Cs = synth.make_vocab(4, 8, seed=1)
    for name, Hm, Wm, H, W in cases:
        m = synth.make_blob_masks(7, Hm, Wm, seed=hash(name) % 1000 if False else len(name) * 31 + Hm)
        tok = synth.make_tokens(Cs, (H // 14) * (W // 14), seed=5)
        _, inc, _ = run_seg_vlad(fv, rec, tok, m, Cs, H, W, None, 8)
        ic[f"{name}_masks"] = np.packbits(m.reshape(7, -1), axis=1)
        ic[f"{name}_shape"] = np.array([7, Hm, Wm, H, W])
        ic[f"{name}_inc"] = inc
    np.savez_compressed(f"{OUT}/incidence_cases.npz", **ic)

    # ---- adjacency cases ----------------------------------------------------------------------------
    ac = {}
    for S in (1, 2, 3, 4, 6, 12, 50):
        m = synth.make_masks(S, 60, 80, seed=300 + S, hmin=3, hmax=20, wmin=3, wmax=25)
        ac[f"S{S}_masks"] = np.packbits(m.reshape(S, -1), axis=1)
        for order in (1, 2, 3):
            ac[f"S{S}_o{order}"] = fv.nbrMasksAGGFastSingle([x for x in m], order).numpy()
    np.savez_compressed(f"{OUT}/adjacency_cases.npz", **ac)

    # ---- vote cases ----------------------------------------------------------------------------------
    vc = {}
    r = np.random.Generator(np.random.PCG64(600))
    n_ref_img, segs, n_q = 300, 20, 40
    imInds = np.repeat(np.arange(n_ref_img), segs)
    seg_per_q = r.integers(1, 30, size=n_q)
    off = np.concatenate([[0], np.cumsum(seg_per_q)])
    nq = int(off[-1])
    matches = r.integers(0, n_ref_img * segs, size=(nq, 50)).astype(np.int64)
    # make votes concentrate: half of each query's matches come from a small image pool
    for i in range(n_q):
        pool = r.integers(0, n_ref_img, size=4)
        rows = slice(off[i], off[i + 1])
        sel = r.random((off[i + 1] - off[i], 50)) < 0.5
        repl = pool[r.integers(0, 4, size=sel.shape)] * segs + r.integers(0, segs, size=sel.shape)
        matches[rows] = np.where(sel, repl, matches[rows])
    sims = np.sort(r.uniform(0.2, 1.9, size=(nq, 50)).astype(np.float32), axis=1)[:, ::-1].copy()
    segRange = [np.arange(off[i], off[i + 1]) for i in range(n_q)]
    gt = [[0]] * n_q
    for n in (1, 5):
        p = fv.get_matches(matches, gt, sims, segRange, imInds, n=n, method="max_seg_topk_wt_borda_Im")
        vc[f"wt_n{n}"] = np.array([list(x) + [-1] * (n - len(x)) for x in p], dtype=np.int64)
    p = fv.get_matches(matches, gt, sims, segRange, imInds, n=5, method="max_seg_topk")
    vc["cnt_n5"] = np.array([list(x) + [-1] * (5 - len(x)) for x in p], dtype=np.int64)
    vc.update(matches=matches, sims=sims, off=off, imInds=imInds)
    # hand-made tie cases: equal weights -> order must follow first appearance (rank-major, then segment)
    tm = np.array([[3, 1, 2, 0], [2, 3, 0, 1]], dtype=np.int64)          # 2 segments x 4 ranks, 4 ref segs
    ts = np.array([[1.0, 0.5, 0.5, 0.25], [1.0, 0.5, 0.5, 0.25]], dtype=np.float32)
    tim = np.array([2, 0, 3, 1])  # image ids must stay < len(imIndsRef): func_vpr.py:219 indexes imIndsRef with them
    p = fv.get_matches(tm, [[0]], ts, [np.arange(2)], tim, n=4, method="max_seg_topk_wt_borda_Im")
    vc.update(tie_matches=tm, tie_sims=ts, tie_imInds=tim, tie_pred=np.array(p[0], dtype=np.int64))
    np.savez_compressed(f"{OUT}/vote_cases.npz", **vc)

    # ---- recall cases ---------------------------------------------------------------------------------
    r = np.random.Generator(np.random.PCG64(700))
    preds = r.integers(0, 40, size=(60, 5))
    gts = [list(r.integers(0, 40, size=r.integers(0, 4))) for _ in range(60)]
    gts[7] = []
    rec5 = fv.calc_recall([list(p) for p in preds], gts, 5)
    gtpad = np.full((60, 4), -1, dtype=np.int64)
    for i, g in enumerate(gts):
        gtpad[i, :len(g)] = g
    np.savez_compressed(f"{OUT}/recall_cases.npz", preds=preds, gt=gtpad, recalls=np.array(rec5))

    # ---- pca_small (sklearn installed here; pins the affine map, not the fit) -----------------------------
    from sklearn.decomposition import PCA
    r = np.random.Generator(np.random.PCG64(800))
    X = (r.standard_normal((200, 64)) @ r.standard_normal((64, 64))).astype(np.float32)
    pca = PCA(n_components=8, whiten=True, svd_solver="arpack", random_state=0).fit(X)
    Xt = r.standard_normal((30, 64))
    np.savez_compressed(f"{OUT}/pca_small.npz", mean=pca.mean_, components=pca.components_,
                        explained_variance=pca.explained_variance_, X=Xt, Y=pca.transform(Xt))

    # ---- e2e_small: recall_segloc chain (kNN via the NumPy IndexFlatL2 stand-in) ---------------------------
    n_img, S, d, n_q = 260, 10, 1024, 30
    R, img = synth.make_planted_db(n_img, S, d, seed=3000)
    Q, tau, off = synth.make_planted_queries(R, n_img, S, n_q, seed=4000, sigma_q=3.0)
    gt = [[int(t)] for t in tau]
    gt[5] = []
    segRange2 = [np.arange(off[i], off[i + 1]) for i in range(n_q)]
    captured = {}
    gm = fv.get_matches

    def gm_wrap(matches, gt_, sims, srq, imr, n=1, method="max_sim"):
        captured.update(matches=matches.copy(), sims=sims.copy())
        p = gm(matches, gt_, sims, srq, imr, n=n, method=method)
        captured["preds"] = p
        return p

    fv.get_matches = gm_wrap
    # the reference scales descriptors by arbitrary norms before normalizeFeat; feed un-normalised rows
    scale_r = np.random.Generator(np.random.PCG64(1)).uniform(0.5, 2.0, size=(R.shape[0], 1))
    scale_q = np.random.Generator(np.random.PCG64(2)).uniform(0.5, 2.0, size=(Q.shape[0], 1))
    recalls = prm.recall_segloc("/tmp", "synthetic", {"pca": True, "results_pkl_suffix": "x"}, "e2e",
                                torch.from_numpy(R.astype(np.float64) * scale_r), torch.from_numpy(Q.astype(np.float64) * scale_q),
                                gt, segRange2, img.astype(np.int64), False, "indoor", save_results=False)
    fv.get_matches = gm
    np.savez_compressed(f"{OUT}/e2e_small.npz", n_img=n_img, S=S, d=d, n_q=n_q, recalls=np.array(recalls),
                        preds=np.array([list(p) + [-1] * (5 - len(p)) for p in captured["preds"]], dtype=np.int64),
                        matches_50=captured["matches"], sims_50=captured["sims"])
    print("e2e recalls", recalls)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot)


# ---------------------------------------------------------------------------------------------------------------------
# producers: the vendored SAM utilities and the reference's DINO token extraction, executed where they lie
# ---------------------------------------------------------------------------------------------------------------------
def gen_get_matches_top1(fv=None):
    """The per-segment-top-1 methods of get_matches (func_vpr.py:86-117; "max_sim" is the DEFAULT argument), executed from the
    reference's own body -> tests/golden/get_matches_top1.npz."""
    if fv is None:
        fv, _ = load_ref()
    # ---- the per-segment-top-1 methods of get_matches (func_vpr.py:86-117; "max_sim" is the DEFAULT argument) -----------
    # inputs are 1-D: one neighbour per query segment.  Similarities are distinct, vote counts are made distinct per
    # query image (the reference ranks both with numpy's unstable argsort: a tie's order is an accident of the sort)
    t1 = {}
    r = np.random.Generator(np.random.PCG64(650))
    n_ref_img, segs, n_q = 120, 10, 30
    imInds1 = np.repeat(np.arange(n_ref_img), segs)
    rows_per_q = r.integers(1, 70, size=n_q)          # images with more than 50 segments exercise max_sim's [-50:]
    off1 = np.concatenate([[0], np.cumsum(rows_per_q)])
    m1 = np.empty(int(off1[-1]), dtype=np.int64)
    for i in range(n_q):
        ns = int(rows_per_q[i])
        pool = r.choice(n_ref_img, size=min(8, n_ref_img), replace=False)
        # image j of the pool gets a distinct number of votes: ns split as descending distinct shares where possible
        shares = np.sort(r.choice(np.arange(1, ns + 8), size=len(pool), replace=False))[::-1].astype(np.float64)
        counts = np.floor(shares / shares.sum() * ns).astype(int)
        counts[0] += ns - counts.sum()
        img = np.repeat(pool, counts)
        r.shuffle(img)
        m1[off1[i]:off1[i + 1]] = img * segs + r.integers(0, segs, size=ns)
    s1 = r.permutation(int(off1[-1])).astype(np.float32) / np.float32(off1[-1]) * np.float32(1.7) + np.float32(0.2)
    segRange1 = [np.arange(off1[i], off1[i + 1]) for i in range(n_q)]
    gt1 = [[0]] * n_q
    for meth in ("max_sim", "max_seg", "max_seg_sim"):
        for n in (1, 5):
            p = fv.get_matches(m1, gt1, s1, segRange1, imInds1, n=n, method=meth)
            t1[f"{meth}_n{n}"] = np.array([list(x) + [-1] * (n - len(x)) for x in p], dtype=np.int64)
    # the default argument IS max_sim
    p = fv.get_matches(m1, gt1, s1, segRange1, imInds1, n=3)
    t1["default_n3"] = np.array([list(x) + [-1] * (3 - len(x)) for x in p], dtype=np.int64)
    # which counts tie inside a query image's top-n (such entries are compared as sets by the tests)
    t1.update(matches=m1, sims=s1, off=off1, imInds=imInds1)
    # 2-D (top-50) inputs: what the reference raises
    errs = []
    for meth in ("max_sim", "max_seg", "max_seg_sim"):
        try:
            m2 = r.integers(0, n_ref_img * segs, size=(40, 50)).astype(np.int64)
            s2 = np.sort(r.uniform(0.2, 1.9, size=(40, 50)).astype(np.float32), axis=1)[:, ::-1].copy()
            fv.get_matches(m2, [[0]], s2, [np.arange(7)], imInds1, n=2, method=meth)
            errs.append("none")
        except Exception as e:  # noqa: BLE001
            errs.append(type(e).__name__)
    t1["errors_2d"] = np.array(errs)
    np.savez_compressed(f"{OUT}/get_matches_top1.npz", **t1)
    print("get_matches_top1:", {k: v.shape for k, v in t1.items() if k.endswith("n5")}, list(t1["errors_2d"]))


def stub_extractor(seed: int, D: int):
    """Deterministic stand-in for the DINOv2 value-facet extractor (weights are not available): [1, 3, h, w] normalised
    image -> [1, (h/14)(w/14), D]: per-patch channel means and mean squares through a seeded linear map + tanh.  The tests
    rebuild the same function from the seed (tests/test_producers.py::_stub_extractor)."""
    g = torch.Generator().manual_seed(seed)
    Wm = torch.randn(6, D, generator=g)

    def f(x):
        p = torch.nn.functional.avg_pool2d(x, 14)                                  # [1, 3, h/14, w/14]
        p2 = torch.nn.functional.avg_pool2d(x * x, 14)
        t = torch.cat([p, p2], 1).flatten(2).transpose(1, 2)                       # [1, n, 6]
        return torch.tanh(t @ Wm) + 0.25 * (t @ Wm)

    return f


def gen_producers():
    """tests/golden/producers.npz:
    * sam/segment_anything/utils/amg.py: build_point_grid, calculate_stability_score, batched_mask_to_box,
      box_xyxy_to_xywh -- pure torch / numpy, executed unmodified;
    * func_vpr.py: getAnyLocFt + process_single_DINO, executed with (a) the stub extractor above, (b) cv2.cvtColor as the
      BGR->RGB channel flip it is (cfg['resize'] = False: cv2.resize is not restated here), (c) the three torchvision
      transforms getAnyLocFt composes (ToTensor, Normalize, CenterCrop), restated from torchvision's documented semantics
      because torchvision is not installable in this image -- so the DINO fixtures pin the reference's own flow (crop to
      patch multiples, extractor call, reshape / permute, channel L2 normalisation), not torchvision itself."""
    out = {}
    amg = types.ModuleType("amg")
    amg.__dict__.update(np=np, torch=torch, math=__import__("math"), deepcopy=__import__("copy").deepcopy)
    from typing import Any, Dict, Generator, ItemsView, List, Tuple
    amg.__dict__.update(Any=Any, Dict=Dict, Generator=Generator, ItemsView=ItemsView, List=List, Tuple=Tuple)
    extract(f"{REF}/sam/segment_anything/utils/amg.py",
            {"build_point_grid", "calculate_stability_score", "batched_mask_to_box", "box_xyxy_to_xywh"}, amg.__dict__)
    for n in (1, 3, 16, 32):
        out[f"grid_{n}"] = amg.build_point_grid(n)
    g = torch.Generator().manual_seed(910)
    for j, (shape, thr, off) in enumerate([((7, 24, 31), 0.0, 1.0), ((4, 3, 17, 20), 0.0, 1.0), ((5, 40, 40), 0.3, 0.7)]):
        logits = torch.randn(shape, generator=g) * 2.0
        out[f"stab_{j}_seed"] = np.array([910, j])
        out[f"stab_{j}_logits"] = logits.numpy()
        out[f"stab_{j}_args"] = np.array([thr, off])
        out[f"stab_{j}_out"] = amg.calculate_stability_score(logits, thr, off).numpy()
    # boxes: blobs, an empty mask, a full mask, a single pixel, extra leading dimensions
    m = torch.rand((9, 30, 44), generator=g) > 0.93
    m[2] = False
    m[3] = True
    m[4] = False
    m[4, 17, 5] = True
    m[5, :, :10] = False
    m[6, :12, :] = False
    out["box_masks"] = np.packbits(m.numpy(), axis=-1)
    out["box_masks_shape"] = np.array(m.shape)
    out["box_out"] = amg.batched_mask_to_box(m).numpy()
    out["box_out_nd"] = amg.batched_mask_to_box(m.reshape(3, 3, 30, 44)).numpy()
    out["box_out_2d"] = amg.batched_mask_to_box(m[0]).numpy()
    # (per box, as automatic_mask_generator.py:157 calls it: the function indexes [2] / [3] of a single box)
    out["box_xywh"] = np.stack([amg.box_xyxy_to_xywh(bx).numpy() for bx in torch.from_numpy(out["box_out"])])

    # ---- getAnyLocFt / process_single_DINO ------------------------------------------------------------------------
    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    def _to_tensor():
        return lambda img: torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255.0)

    def _normalize(mean, std):
        m_, s_ = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
        return lambda x: (x - m_) / s_

    def _center_crop(size):
        def f(x):
            h, w = x.shape[-2:]
            th, tw = size
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
            return x[..., top:top + th, left:left + tw]
        return f

    tvf = types.SimpleNamespace(Compose=_Compose, ToTensor=_to_tensor, Normalize=_normalize, CenterCrop=_center_crop)
    cv2 = types.SimpleNamespace(COLOR_BGR2RGB=4, cvtColor=lambda img, code: np.ascontiguousarray(img[:, :, ::-1]),
                                resize=None)
    fv = types.ModuleType("func_vpr_dino")
    fv.__dict__.update(np=np, torch=torch, tvf=tvf, cv2=cv2)
    extract(f"{REF}/func_vpr.py", {"getAnyLocFt", "process_single_DINO"}, fv.__dict__)
    rng = np.random.Generator(np.random.PCG64(920))
    for j, (H, W, D) in enumerate([(233, 317, 48), (480, 640, 32), (300, 400, 24)]):
        img_bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ext = stub_extractor(930 + j, D)
        cfg = {"resize": False, "dinov2": True}
        img_p, feat_norm = fv.process_single_DINO(cfg, img_bgr, ext, "cpu")
        raw = fv.getAnyLocFt(np.ascontiguousarray(img_bgr[:, :, ::-1]), ext, "cpu", upsample=False)
        up = fv.getAnyLocFt(np.ascontiguousarray(img_bgr[:, :, ::-1]), ext, "cpu", upsample=True) if j == 0 else None
        out[f"dino_{j}_args"] = np.array([920, H, W, D, 930 + j])
        out[f"dino_{j}_img"] = img_bgr
        out[f"dino_{j}_feat_norm"] = feat_norm.numpy()
        out[f"dino_{j}_raw"] = raw.numpy()
        assert np.array_equal(img_p, img_bgr[:, :, ::-1])
        if up is not None:
            out[f"dino_{j}_up_sample"] = up.numpy()[0, :4, ::37, ::41]                 # a thin slice of the up-sampled map
    np.savez_compressed(f"{OUT}/producers.npz", **out)
    print("producers.npz", {k: np.asarray(v).shape for k, v in out.items() if not k.endswith("_img")})


# ---------------------------------------------------------------------------------------------------------------------
# the flow of the vendored SamAutomaticMaskGenerator (round-3 verdict item 6)
# ---------------------------------------------------------------------------------------------------------------------
SAM_CASES = [  # (H, W, points_per_side, points_per_batch, generator keyword arguments, mask_threshold of the stub model, seed)
    (72, 96, 8, 24, {}, 0.0, 9401),                                                        # the reference's defaults
    (72, 96, 8, 64, {"pred_iou_thresh": 0.0, "stability_score_thresh": 0.0, "box_nms_thresh": 0.5, "stability_score_offset": 0.5}, 0.0, 9402),
    (50, 50, 5, 7, {"pred_iou_thresh": 0.8, "stability_score_thresh": 0.9}, 0.3, 9403),     # a model threshold off zero
]


def torchvision_nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms restated from its documented semantics (torchvision is not installable in this image): boxes in
    decreasing score order, each kept unless its IoU with an already kept box exceeds the threshold; returns the kept
    indices in that order.  (The fixture's scores are distinct, so torchvision's unspecified tie order does not enter.)"""
    order = sorted(range(len(scores)), key=lambda i: -float(scores[i]))
    keep = []
    b = boxes.tolist()

    def iou(p, q):
        iw, ih = min(p[2], q[2]) - max(p[0], q[0]), min(p[3], q[3]) - max(p[1], q[1])
        inter = max(iw, 0.0) * max(ih, 0.0)
        return inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter)

    for i in order:
        if all(not (iou(b[i], b[j]) > iou_threshold) for j in keep):
            keep.append(i)
    return torch.tensor(keep, dtype=torch.int64)


def gen_sam_generate():
    """tests/golden/sam_generate.npz: SamAutomaticMaskGenerator.generate -> _generate_masks -> _process_crop ->
    _process_batch (sam/segment_anything/automatic_mask_generator.py, the class body executed where it lies, together
    with the whole of utils/amg.py, imported as the module it is) on a STUB predictor whose logits / predicted IoUs are
    tests/sam_stub.stub_predict -- the mask decoder needs weights this image does not have.  Restated, and said so:
    torchvision's batched_nms / box_area (one category: batched_nms is nms) -- torchvision cannot be installed here."""
    import importlib.util
    from typing import Any, Dict, List, Optional, Tuple

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sam_stub import stub_predict

    spec = importlib.util.spec_from_file_location("ref_amg", f"{REF}/sam/segment_anything/utils/amg.py")
    amg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(amg)

    class StubPredictor:   # the five members the generator touches
        def __init__(self, model):
            self.model = model
            self.device = "cpu"
            self.transform = types.SimpleNamespace(apply_coords=lambda pts, im_size: np.asarray(pts, dtype=np.float64))
            self.im = None

        def set_image(self, im):
            self.im = im.shape[:2]

        def reset_image(self):
            self.im = None

        def predict_torch(self, pts, labels, multimask_output, return_logits):
            assert multimask_output and return_logits and pts.shape[1] == 1 and bool((labels == 1).all())
            lg, io = stub_predict(pts[:, 0, :].numpy(), self.im[0], self.im[1], self.model.seed)
            return torch.from_numpy(lg), torch.from_numpy(io), None

    ns = {k: getattr(amg, k) for k in dir(amg) if not k.startswith("__")}
    ns.update(np=np, torch=torch, Any=Any, Dict=Dict, List=List, Optional=Optional, Tuple=Tuple, Sam=object, SamPredictor=StubPredictor,
              batched_nms=lambda boxes, scores, idxs, iou_threshold: torchvision_nms(boxes, scores, iou_threshold),
              box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    path = f"{REF}/sam/segment_anything/automatic_mask_generator.py"
    body = [n for n in ast.parse(open(path).read()).body if isinstance(n, ast.ClassDef) and n.name == "SamAutomaticMaskGenerator"]
    assert len(body) == 1
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    out = {"n_cases": np.array(len(SAM_CASES))}
    for c, (H, W, pps, ppb, kw, thr, seed) in enumerate(SAM_CASES):
        model = types.SimpleNamespace(mask_threshold=thr, seed=seed)
        gen = ns["SamAutomaticMaskGenerator"](model, points_per_side=pps, points_per_batch=ppb, **kw)
        img = np.zeros((H, W, 3), dtype=np.uint8)
        recs = gen.generate(img)
        assert len(recs) >= 3, (c, len(recs))
        ious = [r["predicted_iou"] for r in recs]
        assert len(set(ious)) == len(ious)   # distinct scores: no tie order to pin
        out[f"c{c}_args"] = np.array([H, W, pps, ppb, seed], dtype=np.int64)
        out[f"c{c}_thr"] = np.array([thr, kw.get("pred_iou_thresh", 0.88), kw.get("stability_score_thresh", 0.95),
                                     kw.get("stability_score_offset", 1.0), kw.get("box_nms_thresh", 0.7)])
        out[f"c{c}_seg"] = np.packbits(np.stack([r["segmentation"] for r in recs]), axis=-1)
        out[f"c{c}_area"] = np.array([r["area"] for r in recs], dtype=np.int64)
        out[f"c{c}_bbox"] = np.array([r["bbox"] for r in recs], dtype=np.int64)
        out[f"c{c}_iou"] = np.array(ious, dtype=np.float64)
        out[f"c{c}_pts"] = np.array([r["point_coords"][0] for r in recs], dtype=np.float64)
        out[f"c{c}_stab"] = np.array([r["stability_score"] for r in recs], dtype=np.float64)
        out[f"c{c}_crop"] = np.array([r["crop_box"] for r in recs], dtype=np.int64)
        print(f"sam_generate case {c}: {len(recs)} records of {pps * pps * 3} proposals")
    np.savez_compressed(f"{OUT}/sam_generate.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "producers":
        os.makedirs(OUT, exist_ok=True)
        patch_cuda_to_cpu()
        gen_producers()
    elif len(sys.argv) > 1 and sys.argv[1] == "sam":
        os.makedirs(OUT, exist_ok=True)
        gen_sam_generate()
    elif len(sys.argv) > 1 and sys.argv[1] == "top1":
        os.makedirs(OUT, exist_ok=True)
        gen_get_matches_top1()
    else:
        main()
        gen_get_matches_top1()
        gen_producers()
        gen_sam_generate()
