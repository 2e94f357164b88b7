"""On-disk formats around the hot path without an h5py dependency.

The reference keeps its inputs in two HDF5 files per split (``func_vpr.py:661-662, 674-678``):

* DINO tokens   ``/{image_key}/ift_dino``                    float32 ``[1, D, h, w]`` (``chunks=True``)
* SAM masks     ``/{image_key}/masks/{j}/{field}``           ``segmentation`` bool ``[Hm, Wm]`` plus the SAM record
                                                            fields ``area, bbox, predicted_iou, point_coords,
                                                            stability_score, crop_box``

and reads them as ``f[key]['ift_dino'][()]`` (``func_vpr.py:1082``) and
``f[f"{key}/masks/"].keys()`` / ``f[f"{key}/masks/{j}"]['segmentation'][()]`` (``func_vpr.py:757-759``).
h5py is not part of this image, so the build's own store is a directory of ``.npz`` files (one per image key) that
presents exactly that nesting: a ``FeatureStore`` can be handed to ``func_vpr.preload_masks`` /
``func_vpr.seg_vlad_gpu_single`` wherever the reference passes an open ``h5py.File``.  ``convert_h5`` /
``export_h5`` move data between the two where h5py exists.

Also here: the small model files of the path -- the vocabulary (``c_centers.pt``, ``place_rec_main.py:149-154``),
the fitted PCA (sklearn pickle, ``func_vpr.py:1434-1438``) and the result pickles ``{sims, matches}``
(``place_rec_main.py:70-75``)."""
from __future__ import annotations

import os
import pickle
import re
import urllib.parse
from typing import Dict, Iterable, Iterator, List, Mapping, Optional, Sequence, Tuple

import numpy as np

MASK_FIELDS = ("segmentation", "area", "bbox", "predicted_iou", "point_coords", "stability_score", "crop_box")


def _natural_key(s: str):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", str(s))]


def _fname(key: str) -> str:
    return urllib.parse.quote(str(key), safe="") + ".npz"


def _key_of(fname: str) -> str:
    return urllib.parse.unquote(fname[:-4])


class Dataset:
    """The slice of ``h5py.Dataset`` the reference uses: ``ds[()]``, ``ds[...]``, ``shape``, ``dtype``."""

    def __init__(self, loader):
        self._loader = loader
        self._value = None

    def _get(self) -> np.ndarray:
        if self._value is None:
            self._value = np.asarray(self._loader())
        return self._value

    def __getitem__(self, item):
        v = self._get()
        return v if (isinstance(item, tuple) and len(item) == 0) else v[item]

    @property
    def shape(self):
        return self._get().shape

    @property
    def dtype(self):
        return self._get().dtype

    def __array__(self, dtype=None):
        v = self._get()
        return v if dtype is None else v.astype(dtype)


class Group(Mapping):
    """Read-only nested mapping with h5py's path semantics: ``g['a/b/']`` == ``g['a']['b']``."""

    def __init__(self, children: Dict[str, object]):
        self._children = children

    def __getitem__(self, path):
        parts = [p for p in str(path).split("/") if p]
        node: object = self
        for p in parts:
            if not isinstance(node, Group):
                raise KeyError(path)
            child = node._children[p]
            node = child() if callable(child) and not isinstance(child, (Group, Dataset)) else child
        return node

    def __iter__(self) -> Iterator[str]:
        return iter(self._children)

    def __len__(self) -> int:
        return len(self._children)

    def keys(self):
        return list(self._children.keys())


class FeatureStore(Group):
    """A directory of per-image ``.npz`` files seen through the reference's HDF5 layout.

    ``kind='dino'``:  ``store[key]['ift_dino'][()]``                          -> float32 ``[1, D, h, w]``
    ``kind='masks'``: ``store[f'{key}/masks/'].keys()``                       -> ``['0', '1', ...]``
                      ``store[f'{key}/masks/{j}']['segmentation'][()]``       -> bool ``[Hm, Wm]``
    Files are opened lazily, one image at a time."""

    def __init__(self, root: str, kind: str):
        if kind not in ("dino", "masks"):
            raise ValueError("kind must be 'dino' or 'masks'")
        self.root, self.kind = root, kind
        names = sorted((f for f in os.listdir(root) if f.endswith(".npz")), key=_natural_key)
        super().__init__({_key_of(f): (lambda f=f: self._open(f)) for f in names})

    def _open(self, fname: str) -> Group:
        path = os.path.join(self.root, fname)
        if self.kind == "dino":
            return Group({"ift_dino": Dataset(lambda: np.load(path)["ift_dino"])})
        z = np.load(path)
        seg = z["segmentation"]
        recs: Dict[str, object] = {}
        for j in range(seg.shape[0]):
            fields: Dict[str, object] = {"segmentation": Dataset(lambda j=j: seg[j])}
            for f in MASK_FIELDS[1:]:
                if f in z.files:
                    fields[f] = Dataset(lambda f=f, j=j: z[f][j])
            recs[str(j)] = Group(fields)
        return Group({"masks": Group(recs)})


def write_dino(root: str, key: str, ift_dino) -> str:
    """One image's token block, as the reference stores it (``func_vpr.py:661-662``): float32 ``[1, D, h, w]``."""
    a = np.asarray(ift_dino, dtype=np.float32)
    if a.ndim != 4 or a.shape[0] != 1:
        raise ValueError(f"ift_dino must be [1, D, h, w], got {a.shape}")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, _fname(key))
    np.savez(path, ift_dino=a)
    return path


def write_masks(root: str, key: str, masks: Sequence) -> str:
    """One image's SAM records (``func_vpr.py:674-678``).  ``masks``: a list of SAM dicts (``segmentation`` + the
    record fields), a list of 2-D arrays, or a ``[S, Hm, Wm]`` array."""
    os.makedirs(root, exist_ok=True)
    if len(masks) and isinstance(masks[0], Mapping):
        seg = np.stack([np.asarray(m["segmentation"]).astype(bool) for m in masks])
        extra = {}
        for f in MASK_FIELDS[1:]:
            if all(f in m for m in masks):
                extra[f] = np.stack([np.asarray(m[f]) for m in masks])
    else:
        seg = np.asarray(masks).astype(bool)
        if seg.ndim == 2:
            seg = seg[None]
        extra = {}
    if seg.ndim != 3:
        raise ValueError(f"masks must stack to [S, Hm, Wm], got {seg.shape}")
    path = os.path.join(root, _fname(key))
    np.savez_compressed(path, segmentation=seg, **extra)
    return path


def _h5py():
    try:
        import h5py  # noqa: WPS433 (optional dependency, absent from the build image)
    except ImportError as e:  # pragma: no cover - exercised only where h5py is missing
        raise ImportError("h5py is needed to read/write the reference's .h5 files; it is not part of this image. "
                          "Run the conversion where h5py exists, or write the store directly with write_dino/write_masks.") from e
    return h5py


def convert_h5(h5_path: str, root: str, kind: str, keys: Optional[Iterable[str]] = None) -> int:
    """Reference HDF5 -> store directory.  Returns the number of images converted."""
    h5py = _h5py()
    n = 0
    with h5py.File(h5_path, "r") as f:
        for key in (keys if keys is not None else f.keys()):
            if kind == "dino":
                write_dino(root, key, f[key]["ift_dino"][()])
            else:
                grp = f[f"{key}/masks/"]
                recs = []
                for j in sorted(grp.keys(), key=_natural_key):
                    recs.append({fld: grp[j][fld][()] for fld in grp[j].keys()})
                write_masks(root, key, recs)
            n += 1
    return n


def export_h5(store: FeatureStore, h5_path: str) -> int:
    """Store directory -> an HDF5 file with the reference's layout (so the reference scripts can read it back)."""
    h5py = _h5py()
    n = 0
    with h5py.File(h5_path, "w") as f:
        for key in store.keys():
            grp = f.create_group(str(key))
            if store.kind == "dino":
                grp.create_dataset("ift_dino", data=store[key]["ift_dino"][()], chunks=True)
            else:
                mg = grp.create_group("masks")
                for j in store[key]["masks"].keys():
                    for fld, ds in store[key]["masks"][j].items():
                        mg.create_dataset(f"{j}/{fld}", data=ds[()])
            n += 1
    return n


# ---- model files ----------------------------------------------------------------------------------------------------
def load_vocabulary(path: str) -> np.ndarray:
    """``c_centers.pt`` (a ``[K, D]`` tensor saved with torch.save, ``place_rec_main.py:149-154``) or ``.npy`` -> float32."""
    if path.endswith(".npy"):
        c = np.load(path)
    else:
        import torch

        c = torch.load(path, map_location="cpu")
        c = c.detach().cpu().numpy() if hasattr(c, "detach") else np.asarray(c)
    c = np.ascontiguousarray(c, dtype=np.float32)
    if c.ndim != 2:
        raise ValueError(f"vocabulary must be [K, D], got {c.shape}")
    return c


def load_pca(path: str) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray], bool]:
    """(mean, components, explained_variance, whiten) from the reference's sklearn pickle (``func_vpr.py:1434-1438``:
    ``pickle.load`` of a fitted ``sklearn.decomposition.PCA``) or from the ``.npz`` written by ``save_pca``."""
    if path.endswith(".npz"):
        z = np.load(path)
        var = z["explained_variance"] if "explained_variance" in z.files else None
        return z["mean"], z["components"], var, bool(z["whiten"]) if "whiten" in z.files else var is not None
    with open(path, "rb") as f:
        model = pickle.load(f)
    mean = np.asarray(model.mean_, dtype=np.float32)
    comps = np.asarray(model.components_, dtype=np.float32)
    var = np.asarray(model.explained_variance_, dtype=np.float32)
    return mean, comps, var, bool(getattr(model, "whiten", False))


def save_pca(path: str, mean, components, explained_variance=None, whiten: bool = True) -> None:
    extra = {} if explained_variance is None else {"explained_variance": np.asarray(explained_variance, dtype=np.float32)}
    np.savez(path, mean=np.asarray(mean, dtype=np.float32), components=np.asarray(components, dtype=np.float32),
             whiten=np.asarray(bool(whiten)), **extra)


def save_results(path: str, sims, matches) -> None:
    """The result pickle of ``recall_segloc`` (``place_rec_main.py:70-75``): ``{'sims': ..., 'matches': ...}``."""
    with open(path, "wb") as f:
        pickle.dump({"sims": np.asarray(sims), "matches": np.asarray(matches)}, f)


def experiment_pickle_paths(workdir: str, dataset_name: str, experiment_name: str, experiment_config: dict, domain) -> dict:
    """The three result files of a ``--save_results`` run, under the reference's own names
    (``place_rec_main.py:62-68`` ``{dataset}_matches_sims_domain_{domain}__{suffix}``, ``:292-300`` ``..._segFtVLAD1_...``,
    ``:357-365`` ``..._segFtVLAD2_...``, all in ``{workdir}/results/global/{experiment_name}/``)."""
    folder = f"{workdir}/results/global//{experiment_name}"      # (the reference's f"{out_folder}/{experiment_name}" with out_folder ending in "/")
    suffix = experiment_config["results_pkl_suffix"]
    return {kind: f"{folder}/{dataset_name}_{kind}_domain_{domain}__{suffix}" for kind in ("segFtVLAD1", "segFtVLAD2", "matches_sims")}


def save_experiment_pickles(workdir: str, dataset_name: str, experiment_name: str, experiment_config: dict, domain,
                            segFtVLAD1=None, segFtVLAD2=None, sims=None, matches=None) -> dict:
    """What ``place_rec_main.py`` pickles under ``save_results`` -- the reference descriptors (``:292-305``) and the query
    descriptors (``:357-370``) as the torch CPU tensors the reference dumps, and ``{'sims', 'matches'}`` = the 200-deep
    ``index.search`` output as NumPy arrays (``:61-75``; 'sims' holds the squared distances, as in the reference).  Only the
    given items are written; returns the paths."""
    import torch

    paths = experiment_pickle_paths(workdir, dataset_name, experiment_name, experiment_config, domain)
    os.makedirs(os.path.dirname(paths["matches_sims"]), exist_ok=True)
    for kind, t in (("segFtVLAD1", segFtVLAD1), ("segFtVLAD2", segFtVLAD2)):
        if t is None:
            continue
        t = t.detach().cpu() if isinstance(t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(t))
        with open(paths[kind], "wb") as f:
            pickle.dump(t, f)
        print(f"{kind} tensor saved to {paths[kind]}")
    if sims is not None and matches is not None:
        to_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)   # noqa: E731
        save_results(paths["matches_sims"], to_np(sims), to_np(matches))
        print(f"Results saved to {paths['matches_sims']}")
    return paths


def load_results(path: str) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as f:
        d = pickle.load(f)
    return np.asarray(d["sims"]), np.asarray(d["matches"])


def list_keys(store: FeatureStore) -> List[str]:
    """Image keys in natural order (the reference iterates the sorted file list of the dataset folder)."""
    return sorted(store.keys(), key=_natural_key)
