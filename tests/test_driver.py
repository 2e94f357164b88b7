"""Host logic of the experiment loop (place_rec_main.py:244-352) over stored inputs: batching, ragged segment counts,
imInds / segRange and their quirks.  The describing function is a NumPy checker here (no GPU)."""
import numpy as np
import pytest

from revisit_anything_amd import driver, store as st


def pooled(tokens, masks, offs):
    """Checker: per segment, [mean of the mask, segment index inside its image, image position in the batch, token sum]."""
    out = []
    for b in range(len(offs) - 1):
        for j in range(offs[b], offs[b + 1]):
            out.append([masks[j].mean(), j - offs[b], b, tokens[b].sum()])
    return np.array(out, np.float64).reshape(-1, 4)


def make_split(tmp_path, seg_counts, D=6, h=2, w=3, Hm=8, Wm=10, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    droot, mroot = str(tmp_path / f"dino{seed}"), str(tmp_path / f"masks{seed}")
    keys = [f"img_{i}.jpg" for i in range(len(seg_counts))]
    toks, masks = {}, {}
    for k, S in zip(keys, seg_counts):
        toks[k] = rng.standard_normal((1, D, h, w)).astype(np.float32)
        masks[k] = (rng.random((S, Hm, Wm)) < 0.3)
        st.write_dino(droot, k, toks[k])
        if S:
            st.write_masks(mroot, k, masks[k])
        else:   # an image without masks still has an (empty) group in the reference's file
            np.savez(f"{mroot}/{st._fname(k)}", segmentation=np.zeros((0, Hm, Wm), bool))
    return st.FeatureStore(droot, "dino"), st.FeatureStore(mroot, "masks"), keys, toks, masks


def test_describe_split_bookkeeping_and_batching(tmp_path):
    counts = [3, 0, 12, 1, 5]
    dino, msk, keys, toks, masks = make_split(tmp_path, counts)
    assert driver.natural_sorted(["img_10.jpg", "img_2.jpg", "img_1.jpg"]) == ["img_1.jpg", "img_2.jpg", "img_10.jpg"]
    for bs in (1, 2, 100):
        desc, im, seg_range = driver.describe_split(dino, msk, keys, pooled, batch_size=bs)
        assert desc.shape == (sum(counts), 4)
        assert im.tolist() == sum(([i] * c for i, c in enumerate(counts)), [])          # place_rec_main.py:250-252
        assert [r.tolist() for r in seg_range] == [np.where(im == i)[0].tolist() for i in range(5)]
        assert len(seg_range[1]) == 0                                                   # image without segments: empty range
        # every row was described with ITS image's tokens and ITS mask, whatever the batching
        row = 0
        for i, (k, c) in enumerate(zip(keys, counts)):
            for j in range(c):
                assert np.isclose(desc[row, 0], masks[k][j].mean()) and desc[row, 1] == j
                assert np.isclose(desc[row, 3], toks[k].sum(), rtol=1e-5)
                assert desc[row, 2] == i % bs if bs < 100 else desc[row, 2] == i
                row += 1


def test_trailing_images_without_segments_get_no_range(tmp_path):
    """place_rec_main.py:287-288 builds segRange for i <= imInds[-1] only."""
    dino, msk, keys, _, _ = make_split(tmp_path, [2, 3, 0], seed=1)
    _, im, seg_range = driver.describe_split(dino, msk, keys, pooled)
    assert im.tolist() == [0, 0, 1, 1, 1] and len(seg_range) == 2


def test_mixed_geometry_in_a_batch_is_an_error(tmp_path):
    dino, msk, keys, _, _ = make_split(tmp_path, [2, 2], seed=2)
    st.write_dino(dino.root, keys[1], np.zeros((1, 6, 3, 3), np.float32))
    with pytest.raises(ValueError, match="different shapes"):
        driver.describe_split(st.FeatureStore(dino.root, "dino"), msk, keys, pooled)
    tokens, m = driver.load_image_inputs(st.FeatureStore(dino.root, "dino"), msk, keys[0])
    assert tokens.shape == (6, 6) and m.shape == (2, 8, 10) and m.dtype == np.uint8


def test_run_segloc_save_results_writes_the_reference_pickles(tmp_path):
    """place_rec_main.py:62-75, 292-305, 357-370: the three `--save_results` files, under the reference's names, with the
    reference's contents (torch CPU tensors; {'sims': d2 [n_q, k_search], 'matches': ids}).  A NumPy stand-in pipeline."""
    import pickle

    import torch

    from revisit_anything_amd.place_rec import default_experiment

    dino_r, msk_r, keys_r, _, _ = make_split(tmp_path, [2, 3, 1], seed=3)
    dino_q, msk_q, keys_q, _, _ = make_split(tmp_path, [2, 1], seed=4)

    class FakeEng:
        device = "cpu"

    class FakePipe:
        eng = FakeEng()

        def describe(self, tokens, masks, offs, l2norm=True):
            return torch.from_numpy(pooled(tokens.numpy(), masks.numpy(), offs).astype(np.float32))

        def index_reset(self):
            self.rows = None

        def index_add(self, rows, img):
            self.rows, self.img = np.asarray(rows), np.asarray(img)

        def retrieve(self, q, q_off, k_search, k_vote, n_top):
            q = np.asarray(q)
            d2 = ((q[:, None, :] - self.rows[None]) ** 2).sum(-1).astype(np.float32)
            idx = np.argsort(d2, axis=1, kind="stable")[:, :k_search]
            d2 = np.take_along_axis(d2, idx, 1)
            self.last_search = (torch.from_numpy(d2), torch.from_numpy(idx))
            pred = np.stack([self.img[idx[q_off[i], :n_top]] for i in range(len(q_off) - 1)])
            return pred, None, idx[:, :k_vote], 2 - d2[:, :k_vote]

    save = {"workdir": str(tmp_path / "w"), "dataset_name": "VPAir", "experiment_name": "e", "domain": "aerial",
            "experiment_config": default_experiment(order=3, pca=True)}
    rec, pred, matches, sims = driver.run_segloc(dino_r, msk_r, keys_r, dino_q, msk_q, keys_q, [[0], [1]], FakePipe(), n_top=2,
                                                 k_search=4, k_vote=3, save_results=save)
    paths = st.experiment_pickle_paths(save["workdir"], "VPAir", "e", save["experiment_config"], "aerial")
    assert paths["segFtVLAD1"].endswith("/results/global//e/VPAir_segFtVLAD1_domain_aerial___results_SegLoc_VLAD_PCA_o3.pkl")
    ft1, ft2 = (pickle.load(open(paths[k], "rb")) for k in ("segFtVLAD1", "segFtVLAD2"))
    assert isinstance(ft1, torch.Tensor) and tuple(ft1.shape) == (6, 4) and tuple(ft2.shape) == (3, 4)
    d2, ids = st.load_results(paths["matches_sims"])
    assert d2.shape == (3, 4) and ids.shape == (3, 4) and np.array_equal(ids[:, :3], matches) and np.allclose(2 - d2[:, :3], sims)
    # without the switch nothing is written
    driver.run_segloc(dino_r, msk_r, keys_r, dino_q, msk_q, keys_q, [[0], [1]], FakePipe(), n_top=2, k_search=4, k_vote=3)
