"""The C-ABI's own multi-GPU layer (csrc/comm.hip: segvlad_comm_init / segvlad_search_sharded / segvlad_allgather_rows) at
WORLD SIZE 2 on a one-GPU box.  RCCL refuses two ranks on one device, so the library is pointed (SEGVLAD_RCCL_LIB, its own
override of the RCCL it binds at run time) at a TEST-ONLY stand-in, tests/rccl_stub/: the six entry points comm.hip binds,
an all-gather over host shared memory between the two processes.  What runs unchanged is everything of ours around the
collective: the packed 12-byte records, the trailer record with a rank's status, the rank-major unpack, the merge by
(distance, global id), and the rule that a collective is entered by every rank or aborted."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "rccl_stub")
STUB = os.path.join(STUB_DIR, "librccl_stub.so")


def _build_stub():
    """Built lazily HERE (and, best effort, by __graft_entry__.build()): the product build never depends on it; a box that cannot
    build it skips these tests."""
    sys.path.insert(0, STUB_DIR)
    try:
        import build_stub

        build_stub.build()
    except Exception as e:   # noqa: BLE001
        pytest.skip(f"the RCCL test stub cannot be built here: {e}")
    finally:
        sys.path.remove(STUB_DIR)


def _problem():
    from revisit_anything_amd import synth

    n_img, S, d, n_q = 900, 40, 64, 12
    R, img = synth.make_planted_db(n_img, S, d, seed=3000)
    Q, tau, off = synth.make_planted_queries(R, n_img, S, n_q, seed=4000, sigma_q=2.0)
    return R, img, Q, tau, off, n_img, S


def _worker(rank, world, out_dir, mode):
    sys.path.insert(0, ROOT)
    os.environ["SEGVLAD_RCCL_LIB"] = STUB
    os.environ.setdefault("SVSTUB_TIMEOUT_S", "30")
    import torch

    from revisit_anything_amd._lib import SegVLADError
    from revisit_anything_amd.engine import SegVLADEngine
    from revisit_anything_amd.sharded import shard_images

    R, img, Q, tau, off, n_img, S = _problem()
    eng = SegVLADEngine(0)
    uid_file = os.path.join(out_dir, "uid.bin")
    if rank == 0:
        uid = eng.comm_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            assert time.time() - t0 < 60
            time.sleep(0.01)
        uid = open(uid_file, "rb").read()
    eng.comm_init(uid, rank, world)
    info = eng.comm_info()
    assert info["world"] == world and info["rank"] == rank and "rccl_stub" in info["rccl"], info
    ib = shard_images(n_img, world)
    lo, hi = int(ib[rank]) * S, int(ib[rank + 1]) * S
    if mode != "empty_shard" or rank == 0:
        eng.db_add(R[lo:hi] if mode != "empty_shard" else R, None)
    base = lo if mode != "empty_shard" else 0
    Qd = torch.from_numpy(Q).to(eng.device)
    res = {"mode": mode}
    if mode in ("ok", "empty_shard"):
        d2, ids = eng.search_sharded(Qd, 60, base)
        # the query rows split over the ranks and gathered back, like bench.py's descriptors
        half = Q.shape[0] // world
        rows = eng.allgather_rows(Qd[rank * half:(rank + 1) * half], world)
        res.update(d2=d2.cpu().numpy(), ids=ids.cpu().numpy(), rows=rows.cpu().numpy())
        d2b, idsb = eng.search_sharded(Qd, 7, base)       # a second collective on the same communicator
        res.update(d2b=d2b.cpu().numpy(), idsb=idsb.cpu().numpy())
    else:
        if rank == 1:
            eng.set_option("debug_fail_search", 1 if mode == "fail_local" else 2)
        try:
            eng.search_sharded(Qd, 60, base)
            res["error"] = "none"
        except SegVLADError as e:
            res["error"] = str(e)
            res["code"] = int(e.code)
        res["world_after"] = eng.comm_info()["world"]
        if mode == "fail_local":   # every rank left the collective: the communicator is still good
            eng.set_option("debug_fail_search", 0)
            d2, ids = eng.search_sharded(Qd, 60, base)
            res.update(d2=d2.cpu().numpy(), ids=ids.cpu().numpy())
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    eng.close()


def _run(tmp_path, mode):
    import torch.multiprocessing as mp

    _build_stub()
    mp.spawn(_worker, args=(2, str(tmp_path), mode), nprocs=2, join=True)
    return [np.load(tmp_path / f"r{r}.npz") for r in range(2)]


def _single():
    from revisit_anything_amd.engine import SegVLADEngine

    R, img, Q, tau, off, n_img, S = _problem()
    eng = SegVLADEngine(0)
    eng.db_add(R, img)
    out = [x.cpu().numpy() for x in eng.search(Q, 60)] + [x.cpu().numpy() for x in eng.search(Q, 7)]
    eng.close()
    return out, Q


def test_two_ranks_through_the_cabi_equal_a_single_index(tmp_path):
    (d2, ids, d2b, idsb), Q = _single()
    for z in _run(tmp_path, "ok"):
        assert np.array_equal(z["ids"], ids) and np.array_equal(z["d2"], d2)
        assert np.array_equal(z["idsb"], idsb) and np.array_equal(z["d2b"], d2b)
        half = Q.shape[0] // 2
        assert np.array_equal(z["rows"], Q[:2 * half])


def test_an_empty_shard_contributes_nothing(tmp_path):
    (d2, ids, d2b, idsb), _ = _single()
    for z in _run(tmp_path, "empty_shard"):      # rank 0 holds every row, rank 1 none: (inf, -1) records never win the merge
        assert np.array_equal(z["ids"], ids) and np.array_equal(z["d2"], d2)


def test_a_failed_local_search_is_an_error_on_every_rank_and_the_communicator_survives(tmp_path):
    from revisit_anything_amd import _lib

    (d2, ids, _, _), _ = _single()
    z0, z1 = _run(tmp_path, "fail_local")
    assert int(z1["code"]) == _lib.SEGVLAD_ERR_STATE and "local search failed" in str(z1["error"])
    assert int(z0["code"]) == _lib.SEGVLAD_ERR_COMM and "rank 1" in str(z0["error"])
    for z in (z0, z1):
        assert int(z["world_after"]) == 2
        assert np.array_equal(z["ids"], ids) and np.array_equal(z["d2"], d2)


def test_a_rank_that_cannot_join_aborts_the_communicator_and_nobody_hangs(tmp_path):
    from revisit_anything_amd import _lib

    t0 = time.time()
    z0, z1 = _run(tmp_path, "abort")
    assert time.time() - t0 < 120
    assert int(z1["code"]) == _lib.SEGVLAD_ERR_STATE and "aborted" in str(z1["error"]) and int(z1["world_after"]) == 0
    # the peer was inside (or entered) the all-gather: it fails there instead of waiting for ever, and aborts its side too
    assert int(z0["code"]) == _lib.SEGVLAD_ERR_COMM and int(z0["world_after"]) == 0, (str(z0["error"]), int(z0["world_after"]))
