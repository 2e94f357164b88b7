"""bench.py's own step on the GPU box (VERDICT r01 #1d, #8): the device predictions of the benchmarked workload must
equal the fp64 oracle's on >= 20 query images at a difficulty where the oracle's Recall@1 is well below 1 (the vote
matters), and the N = 2 row-sharded path (two ranks sharing the one GPU, gloo collectives) must give the N = 1
predictions."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SEGVLAD_GUARD="0")   # bench.py times: plain contexts (conftest switches the guard on)
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_step_predictions_equal_oracle_on_20_images():
    j = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--verify-images", "20"], 1500)
    oc = j["oracle_check"]
    assert oc["images"] == 20
    assert oc["top1_identical"] == 20, oc                     # north_star: identical top-1 image ids
    assert oc["sims_max_abs_diff"] < 1e-4, oc                 # north_star: cosine scores within 1e-4
    assert oc["neighbour_id_mismatches_are_near_ties"], oc
    assert oc["ok"]
    assert oc["device_recall_at_1"] == oc["oracle_recall_at_1"]
    assert j["filter_dtype"] == "f16" and j["dtype"] == "f32"
    assert j["search_stats"]["n_fallback"] == 0 and j["search_stats"]["levels"] == 3
    assert j["roofline"]["bound"] == "mfma" and j["cpu_baseline"]["kind"] == "port"
    # the sub-records of the N=1 line: both timing modes give the same predictions; BASELINE
    # configs[1] in its literal raw-descriptor form and the 31-near-duplicates database run the fp16 filter with nobody on
    # the distance-matrix path
    assert j["mode"] == "serial" and j["pipelined"]["predictions_identical"] is True
    c2, rd = j["config2"], j["redundant_db"]
    assert "error" not in c2 and "error" not in rd, (c2, rd)
    assert c2["search_stats"]["filter"] == "f16" and c2["search_stats"]["n_fallback"] == 0 and "raw K*D" in c2["workload"]
    assert c2["search_stats"]["refine_sum"] / c2["search_stats"]["n_queries"] < 400
    assert rd["sibling_group"] == 31 and rd["search_stats"]["n_fallback"] == 0 and rd["search_stats"]["n_redo"] <= 100
    assert rd["recall_at_1_within_sibling_group"] >= 0.95
    # round 4: the same two workloads with the single index searching only as deep as the vote reads (50 of 200 columns):
    # the predictions are those of the 200-deep runs, the refinement a fraction
    vd, c2v = j["vote_depth"], j["config2_vote_depth"]
    assert "error" not in vd and "error" not in c2v, (vd, c2v)
    assert vd["predictions_identical_to_search_200"] is True and c2v["predictions_identical_to_search_200"] is True
    assert "search 50" in vd["workload"] and "search 200" in j["config"]["workload"]
    # (every timing / rate / workload-difficulty assertion on this line lives in tests/test_gpu_perf.py under the marker
    #  `gpu_perf`: a slow box must never turn the parity suite red -- VERDICT r05 "weak" #2.  The line is left for that test.)
    with open(os.path.join(ROOT, "gpurun_out", "bench_step_line.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out"))
              else os.path.join(ROOT, ".bench_step_line.json"), "w") as f:
        json.dump(j, f)


def test_bench_two_ranks_one_gpu_equal_single_rank(tmp_path):
    """--gpus 2 with both ranks on cuda:0 and gloo collectives: the sharded path (query-descriptor gather, packed
    (d2, id) all-gather, merge, vote) must reproduce the single-rank predictions exactly."""
    p1, p2 = str(tmp_path / "p1.npy"), str(tmp_path / "p2.npy")
    common = ["--steps", "1", "--warmup", "1", "--db-images", "4000", "--query-images", "40", "--no-cpu-baseline"]
    j1 = _run([sys.executable, "bench.py", *common, "--dump-preds", p1], 900)
    port = 29500 + (os.getpid() % 2000)
    j2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2", "--same-device", "--dist-backend", "gloo", *common,
               "--dump-preds", p2], 900)
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert np.array_equal(np.load(p1), np.load(p2))
    assert j1["recall_at_1"] == j2["recall_at_1"]


def test_bench_four_ranks_one_gpu_ragged_slices_equal_single_rank(tmp_path):
    """The same at N = 4 with RAGGED slices (42 query images: 10 / 10 / 11 / 11 per rank; 3001 reference images: shards of 750 / 750 /
    750 / 751 images) -- the padded row gather and its trimming, shard bounds that are not multiples of anything -- and the round-6
    records of the N > 1 line: per rank the collectives' milliseconds and bytes (tools/n8_same_device.sh runs N = 8 the same way)."""
    p1, p4 = str(tmp_path / "p1.npy"), str(tmp_path / "p4.npy")
    common = ["--steps", "1", "--warmup", "1", "--db-images", "3001", "--query-images", "42", "--no-cpu-baseline", "--no-ubench",
              "--no-sub-records", "--shard-sim", "0"]
    j1 = _run([sys.executable, "bench.py", *common, "--dump-preds", p1], 900)
    port = 31500 + (os.getpid() % 2000)
    j4 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "4", "--same-device", "--dist-backend", "gloo", *common,
               "--dump-preds", p4], 1200)
    assert j4["n_gpus"] == 4 and np.array_equal(np.load(p1), np.load(p4)) and j1["recall_at_1"] == j4["recall_at_1"]
    ranks = j4["per_rank_stages_ms"]
    assert len(ranks) == 4 and sorted(r["n_local_rows"] for r in ranks) == [37500, 37500, 37500, 37550]
    for r in ranks:
        assert set(r["collective_ms_per_step"]) == {"query_rows_allgather", "topk_records_allgather"}
        assert r["collective_bytes_per_step"]["topk_records_allgather_send"] == 42 * 50 * 50 * 12
        assert r["collective_bytes_per_step"]["query_rows_allgather_recv"] == 4 * 11 * 50 * 1024 * 4     # padded to the longest slice


def test_bench_config2_raw_descriptors_end_to_end_equals_oracle():
    """BASELINE configs[1] literally (place_rec_main.py:49-60 with pca off): 1000 reference images x 50 segments of raw
    K*D = 98 304-d descriptors, 200 query images, search 200 / vote 50 -- one image's 50 query segments re-computed by the
    fp64 oracle against all 50 000 rows: identical top-1 image, sims within 1e-4, every id mismatch a near-tie."""
    j = _run([sys.executable, "bench.py", "--no-pca", "--db-images", "1000", "--steps", "1", "--warmup", "1", "--verify-images", "1",
              "--no-sub-records", "--no-ubench", "--search-stats"], 1500)
    oc = j["oracle_check"]
    assert oc["images"] == 1 and oc["top1_identical"] == 1 and oc["ok"], oc
    assert oc["sims_max_abs_diff"] < 1e-4 and oc["neighbour_id_mismatches_are_near_ties"], oc
    st = j["search_stats"]
    assert j["filter_dtype"] == "f16" and st["n_fallback"] == 0 and st["levels"] >= 1, st
    assert j["config"]["pca_dim"] is None and j["config"]["db_segments"] == 50000
