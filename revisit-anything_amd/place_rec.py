"""Drop-in for the hot-path part of the reference's ``place_rec_main.py``: ``recall_segloc``
(place_rec_main.py:44-96), a ``faiss.IndexFlatL2``-compatible exact index, the dataset/experiment
config surface (place_rec_global_config.py) and the ground-truth rules that need no dataset images
(gt.py:60-64 17places, 66-69 AmsterTime, 71-73 VPAir's vpair_gt.npy)."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from . import func_vpr


class IndexFlatL2:
    """faiss.IndexFlatL2 surface used by the reference (place_rec_main.py:53-60): ``add(x)``,
    ``search(x, k) -> (D2 float32 [n,k] ascending, I int64 [n,k])``, ``ntotal``, ``d``.  Exact,
    brute force, on the MI355X (16-bit MFMA candidate filter + exact fp32 refinement, or the fp32 distance-matrix
    path for small indices).  Every index owns its own engine context (its own rows, planes and scratch), so several
    indices can be alive at once -- as with faiss -- and none of them disturbs the vocabulary / PCA state of the
    process-wide func_vpr engine."""

    def __init__(self, d: int):
        from .engine import SegVLADEngine

        self.d = int(d)
        self._eng = SegVLADEngine(func_vpr.engine().device)
        self.ntotal = 0

    def reset(self):
        self._eng.db_reset()
        self.ntotal = 0

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x
        if x.shape[1] != self.d:
            raise ValueError(f"IndexFlatL2(d={self.d}).add got vectors of dimension {x.shape[1]}")
        self._eng.db_add(x)
        self.ntotal += int(x.shape[0])

    def search(self, x, k: int):
        x = np.ascontiguousarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x
        if x.shape[1] != self.d:
            raise ValueError(f"IndexFlatL2(d={self.d}).search got vectors of dimension {x.shape[1]}")
        d2, idx = self._eng.search(x, int(k))
        return d2.cpu().numpy(), idx.cpu().numpy()


def _to_numpy(t):
    return t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def recall_segloc(workdir, dataset_name, experiment_config, experiment_name, segFtVLAD1, segFtVLAD2, gt, segRange2,
                  imInds1, map_calculate, domain, save_results=True):
    """place_rec_main.py:44-96, same arguments and return value (the list of Recall@1..5).
    ``d`` is derived from the data (the reference hard-codes 1024 / 49152, place_rec_main.py:49,52)."""
    R = _to_numpy(segFtVLAD1)
    Q = _to_numpy(segFtVLAD2)
    index = IndexFlatL2(R.shape[1])
    if experiment_config["pca"]:
        # (the reference's normalizeFeat returns host arrays that index.add / search upload again -- place_rec_main.py:53-55; here the
        #  normalised rows stay on the device between the two steps: one PCIe crossing for a 1 M x 1024 database instead of three.
        #  Same kernel, same bits as func_vpr.normalizeFeat)
        index.add(func_vpr._normalizeFeat_device(R))
        sims, matches = index.search(func_vpr._normalizeFeat_device(Q), 200)
    else:
        index.add(R)
        sims, matches = index.search(Q, 200)
    if save_results:
        out_folder = f"{workdir}/results/global/"
        os.makedirs(f"{out_folder}/{experiment_name}", exist_ok=True)
        pkl = f"{out_folder}/{experiment_name}/{dataset_name}_matches_sims_domain_{domain}__{experiment_config['results_pkl_suffix']}"
        with open(pkl, "wb") as file:
            pickle.dump({"sims": sims, "matches": matches}, file)
        print(f"Results saved to {pkl}")
    sims_50 = 2 - sims[:, :50]
    matches_50 = matches[:, :50]
    max_seg_preds = func_vpr.get_matches(matches_50, gt, sims_50, segRange2, imInds1, n=5, method="max_seg_topk_wt_borda_Im")
    max_seg_recalls = func_vpr.calc_recall(max_seg_preds, gt, 5)
    print("VLAD + PCA Results \n ")
    if map_calculate:
        queries_results = func_vpr.convert_to_queries_results_for_map(max_seg_preds, gt)
        print(f"Mean Average Precision (mAP): {func_vpr.calculate_map(queries_results)}")
    print("Max Seg Logs: ", max_seg_recalls)
    return max_seg_recalls


# ---- config surface (place_rec_global_config.py:8-232): same keys, so reference configs load unchanged ----
def load_global_config(path: str):
    """Execute a reference-style ``place_rec_global_config.py`` and return (datasets, experiments, workdir_data)."""
    ns: dict = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    return ns["datasets"], ns["experiments"], ns.get("workdir_data")


def default_experiment(order: int = 3, pca: bool = True) -> dict:
    """The keys the drivers read from ``experiments[name]`` (place_rec_main.py:197-226, 250, 255, 261)."""
    return {"results_pkl_suffix": f"_results_SegLoc_VLAD{'_PCA' if pca else ''}_o{order}.pkl", "global_method_name": "SegLoc",
            "minArea": 0, "order": order, "pca": pca, "pca_model_pkl": f"_r_fitted_pca_model_order{order}.pkl",
            "pca_model_pkl_map": f"_r_fitted_pca_model_order{order}_map.pkl"}


def default_dataset(name: str, width: int = 640, height: int = 480, mask_width: int = 320, vocab: str = "indoor") -> dict:
    """The keys the drivers read from ``datasets[name]`` (place_rec_main.py:119-166, 197-200)."""
    return {"masks_h5_filename_r": f"{name}_r_masks_{mask_width}.h5", "masks_h5_filename_q": f"{name}_q_masks_{mask_width}.h5",
            "dino_h5_filename_r": f"{name}_r_dino_{width}.h5", "dino_h5_filename_q": f"{name}_q_dino_{width}.h5",
            "data_subpath1_r": "ref", "data_subpath2_q": "query",
            "cfg": {"rmin": 0, "desired_width": width, "desired_height": height},
            "map_vlad_cluster": name, "domain_vlad_cluster": vocab}


def get_gt(dataset, cfg=None, workdir_data=None, ims1_r=None, ims2_q=None, func_vpr_module=None):
    """The ground-truth rules that need no dataset files: 17places (gt.py:60-64: query frame i matches reference frames
    i-15 .. i+15) and AmsterTime (gt.py:66-69: pair i matches pair i); VPAir reads its one metadata file.  Everything else
    needs the dataset's own metadata (out of scope)."""
    # (both rules only need the LENGTH of an image list; a missing list is the caller's error, as in gt.py)
    rules = {"17places": ("ims2_q", ims2_q, 15), "AmsterTime": ("ims1_r", ims1_r, 0)}
    if dataset in rules:
        arg_name, images, radius = rules[dataset]
        if images is None:
            raise ValueError(f"{arg_name} is required for the {dataset} ground truth (frame i matches frames i-{radius}..i+{radius})")
        window = np.arange(-radius, radius + 1)
        # 17places: a list of numpy integers per query, out-of-range frames included (calc_recall only tests membership);
        # AmsterTime: the plain index
        return [list(i + window) if radius else [i] for i in range(len(images))]
    if dataset == "VPAir":
        # gt.py:71-73 -> dataloaders/vpair_dataloader.py:93-98: `vpair_gt.npy` holds, per query, a pair whose second
        # entry lists the soft-positive reference indices
        if workdir_data is None:
            raise ValueError("workdir_data must be provided for the VPAir dataset.")
        gt_positives = np.load(os.path.join(workdir_data, "VPAir", "vpair_gt.npy"), allow_pickle=True)
        return [gt_positives[i][1] for i in range(len(gt_positives))]
    raise NotImplementedError(f"ground truth for dataset {dataset!r} needs its metadata files (gt.py)")
