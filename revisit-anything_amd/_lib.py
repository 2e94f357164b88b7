"""ctypes binding of include/segvlad.h.  Loads the in-tree libsegvlad_hip.so and fails loudly when
it is missing or cannot be loaded -- there is NO CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_ctx_p = C.c_void_p
_f32p = C.c_void_p  # all bulk pointers are passed as raw addresses (host or device)

SEGVLAD_OK = 0
# error codes of include/segvlad.h (SegVLADError.code)
SEGVLAD_ERR_ARG, SEGVLAD_ERR_HIP, SEGVLAD_ERR_STATE, SEGVLAD_ERR_LIMIT, SEGVLAD_ERR_NOMEM, SEGVLAD_ERR_COMM = -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
VOTE_WT_BORDA_IM = 0
VOTE_COUNT = 1

# name -> (restype, argtypes); kept in one table so tests can check the exported symbols against the header
SIGNATURES = {
    "segvlad_version": (C.c_int, []),
    "segvlad_create": (C.c_int, [C.POINTER(c_ctx_p), C.c_int]),
    "segvlad_destroy": (C.c_int, [c_ctx_p]),
    "segvlad_last_error": (C.c_char_p, [c_ctx_p]),
    "segvlad_set_stream": (C.c_int, [c_ctx_p, C.c_void_p]),
    "segvlad_synchronize": (C.c_int, [c_ctx_p]),
    "segvlad_set_vocab": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int]),
    "segvlad_incidence": (C.c_int, [c_ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "segvlad_incidence_centroids": (C.c_int, [c_ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    "segvlad_mask_centroids": (C.c_int, [c_ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "segvlad_adjacency": (C.c_int, [c_ctx_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "segvlad_adjacency_flagged": (C.c_int, [c_ctx_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "segvlad_images": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _f32p, C.c_void_p,
                                  _f32p, _f32p]),
    "segvlad_images_pca": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _f32p, C.c_int,
                                      _f32p, C.c_void_p, _f32p]),
    "segvlad_describe": (C.c_int, [c_ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_int]),
    "segvlad_describe_begin": (C.c_int, [c_ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "segvlad_describe_flags": (C.c_int, [c_ctx_p, C.c_void_p, C.c_void_p]),
    "segvlad_describe_end": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, _f32p, _f32p, C.c_int]),
    "segvlad_kmeans_step": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "segvlad_cluster_aggregate": (C.c_int, [c_ctx_p, C.c_int, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                             C.c_void_p, _f32p]),
    "segvlad_pca_set": (C.c_int, [c_ctx_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]),
    "segvlad_pca_apply": (C.c_int, [c_ctx_p, _f32p, C.c_int, _f32p, C.c_int]),
    "segvlad_normalize_rows": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, _f32p]),
    "segvlad_db_reset": (C.c_int, [c_ctx_p]),
    "segvlad_db_add": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, C.c_void_p]),
    "segvlad_db_size": (C.c_int, [c_ctx_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "segvlad_search": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, _f32p, C.c_void_p]),
    "segvlad_merge_topk": (C.c_int, [c_ctx_p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p]),
    "segvlad_sims_from_d2": (C.c_int, [c_ctx_p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p]),
    "segvlad_minmax": (C.c_int, [c_ctx_p, _f32p, C.c_int64, _f32p]),
    "segvlad_vote": (C.c_int, [c_ctx_p, C.c_void_p, _f32p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "segvlad_set_profiling": (C.c_int, [c_ctx_p, C.c_int]),
    "segvlad_profile_reset": (C.c_int, [c_ctx_p]),
    "segvlad_stage_ms": (C.c_int, [c_ctx_p, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "segvlad_set_option": (C.c_int, [c_ctx_p, C.c_char_p, C.c_char_p]),
    "segvlad_search_stats": (C.c_int, [c_ctx_p, C.POINTER(C.c_int64), C.c_int]),
    "segvlad_comm_unique_id": (C.c_int, [C.c_void_p]),
    "segvlad_comm_init": (C.c_int, [c_ctx_p, C.c_void_p, C.c_int, C.c_int]),
    "segvlad_comm_destroy": (C.c_int, [c_ctx_p]),
    "segvlad_comm_info": (C.c_int, [c_ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "segvlad_allgather_rows": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, _f32p]),
    "segvlad_search_sharded": (C.c_int, [c_ctx_p, _f32p, C.c_int, C.c_int, C.c_int64, _f32p, C.c_void_p]),
}

_lib = None


class SegVLADError(RuntimeError):
    """A failed C-ABI call.  ``code`` is the library's return value (SEGVLAD_ERR_*; None when the failure happened on
    the Python side): callers branch on the CODE / the exception TYPE, never on the wording of the message."""

    def __init__(self, msg: str = "", code=None):
        super().__init__(msg)
        self.code = code


class SegVLADDegenerateError(SegVLADError):
    """The device Delaunay met a non-generic centroid configuration (duplicate or co-circular centroids): the
    triangulation is then a matter of Qhull's tie-breaking, and the caller should take the reference's Qhull path."""


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if the sources are newer and hipcc is present) the HIP library."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    alt = os.environ.get("SEGVLAD_LIB_PATH")   # development only: another build of the same library (e.g. the timing-ablation build)
    if alt:
        path, build_if_missing = alt, False
    if build_if_missing and _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # no hipcc on this box: use the shipped .so if there is one
            if not os.path.exists(path):
                raise SegVLADError(f"libsegvlad_hip.so is missing and could not be built: {e}") from e
    if not os.path.exists(path):
        raise SegVLADError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                           "There is no CPU fallback.")
    # torch ships its own libamdhip64: import it FIRST so that this library binds to the HIP runtime
    # PyTorch-ROCm already initialised (one runtime per process; streams/pointers are then shared)
    import torch  # noqa: F401

    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise SegVLADError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
