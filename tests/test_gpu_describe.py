"""segvlad_describe (round 5): the describe stage of a batch in ONE call, with the mask branch (incidence + centroids ->
adjacency) on the context's side stream beside the token-assignment pass.  It issues the SAME kernels as the separate entry
points, so every output must equal theirs bit for bit -- ragged batches, PCA and raw descriptors, repeated calls (the side
stream is forked and joined every time), and the pipeline's flag handling on top of it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(K=32, D=256, P=64, seed=0):
    from revisit_anything_amd import synth
    from revisit_anything_amd.engine import SegVLADEngine

    eng = SegVLADEngine(0)
    C = synth.make_vocab(K, D, seed=1000 + seed)
    eng.set_vocab(C)
    mean, comps, var = synth.make_pca_model(K * D, P, seed=5000 + seed)
    eng.pca_set(mean, comps, var, whiten=True)
    return eng, C


def _batch(C, B, S_list, H, W, seed):
    from revisit_anything_amd import synth

    N = (H // 14) * (W // 14)
    tok = np.stack([synth.make_tokens(C, N, seed=seed + b, noise=0.2) for b in range(B)])
    masks = np.concatenate([synth.make_blob_masks(S_list[b], H // 2, W // 2, seed=seed + 100 + b) for b in range(B)])
    off = np.concatenate([[0], np.cumsum(S_list)]).astype(np.int32)
    return torch.from_numpy(tok).cuda(), torch.from_numpy(masks.astype(np.uint8)).cuda(), off


@pytest.mark.parametrize("pca", [True, False])
def test_describe_equals_the_separate_calls_bit_for_bit(pca):
    eng, C = _setup()
    H, W = 112, 140
    for rep, S_list in enumerate(([12, 7, 15, 9], [5, 5, 5, 5, 5, 5], [20])):
        tok, masks, off = _batch(C, len(S_list), S_list, H, W, seed=10 * rep)
        bits, cent = eng.incidence_centroids(masks, H, W, 14)
        adj, flags = eng.adjacency_flagged(cent, off, 3, device_flags=True)
        ref = (eng.seg_vlad_pca(tok, bits, off, adj, l2norm=True) if pca else eng.seg_vlad(tok, bits, off, adj))["out"]
        r = eng.describe(masks, tok, off, H, W, 14, 3, pca=pca, l2norm=True)
        assert torch.equal(r["bits"], bits) and torch.equal(r["adj"], adj) and torch.equal(r["flags"], flags)
        assert torch.equal(r["cent"].view(torch.int64), cent.view(torch.int64))
        assert torch.equal(r["out"], ref)
        r2 = eng.describe(masks, tok, off, H, W, 14, 3, pca=pca, l2norm=True)       # again: fork / join per call
        assert torch.equal(r2["out"], ref)
    eng.close()


def test_pipeline_describe_through_the_fused_call_equals_the_stepwise_pipeline():
    from revisit_anything_amd.pipeline import SegVLADPipeline

    eng, C = _setup(seed=1)
    H, W = 112, 140
    tok, masks, off = _batch(C, 5, [9, 14, 6, 11, 8], H, W, seed=77)
    fused = SegVLADPipeline(eng, H, W, 14, order=2, use_pca=True)
    step = SegVLADPipeline(eng, H, W, 14, order=2, use_pca=True, host_adjacency=True)   # Qhull for every image, stepwise calls
    a = fused.describe(tok, masks, off)
    b = step.describe(tok, masks, off)
    assert torch.equal(a, b)
    # an empty mask is reported like before
    masks2 = masks.clone()
    masks2[3] = 0
    with pytest.raises(ValueError):
        fused.describe(tok, masks2, off)
    eng.close()


def test_describe_argument_errors():
    from revisit_anything_amd._lib import SEGVLAD_ERR_ARG, SegVLADError

    eng, C = _setup(seed=2)
    H, W = 112, 140
    tok, masks, off = _batch(C, 2, [4, 6], H, W, seed=5)
    with pytest.raises(SegVLADError) as ei:
        eng.describe(masks, tok, off, H, W, 14, 0)              # order 0 has no mask branch: segvlad_images(adj = NULL)
    assert ei.value.code == SEGVLAD_ERR_ARG
    with pytest.raises(SegVLADError) as ei:
        eng.describe(masks, tok, off, H + 14, W, 14, 3)         # token grid does not match the image
    assert ei.value.code == SEGVLAD_ERR_ARG
    out = eng.describe(masks, tok, off, H, W, 14, 3)["out"]     # the context is still usable
    assert torch.isfinite(out).all()
    eng.close()


def test_split_describe_with_patched_adjacency_equals_images_pca_on_the_patched_adjacency():
    """begin / flags / end: the flags and centroids reach the host while the assignment pass runs; blocks handed to `end`
    replace the device adjacency of their images, and the result is segvlad_images_pca's on that adjacency."""
    from revisit_anything_amd.func_vpr import adjacency_from_centroids

    eng, C = _setup(seed=3)
    H, W = 112, 140
    S_list = [9, 13, 7, 10]
    tok, masks, off = _batch(C, 4, S_list, H, W, seed=31)
    bits, cent = eng.incidence_centroids(masks, H, W, 14)
    adj, flags = eng.adjacency_flagged(cent, off, 3, device_flags=True)
    h = eng.describe_begin(masks, tok, off, H, W, 14, 3, pca=True)
    f_host, c_host = eng.describe_flags(h)
    assert np.array_equal(f_host, flags.cpu().numpy()) and np.array_equal(c_host, cent.cpu().numpy())
    # patch images 1 and 3 with ANOTHER adjacency (order 1 instead of 3: visibly different descriptors)
    imgs = [1, 3]
    blocks = [adjacency_from_centroids(c_host[off[b]:off[b + 1]], 1).numpy().astype(np.uint8) for b in imgs]
    out = eng.describe_end(h, imgs, blocks, l2norm=True)["out"]
    adj2 = adj.clone()
    aoff = np.concatenate([[0], np.cumsum(np.asarray(S_list) ** 2)])
    for b, blk in zip(imgs, blocks):
        adj2[int(aoff[b]):int(aoff[b + 1])] = torch.from_numpy(blk.reshape(-1)).cuda()
    ref = eng.seg_vlad_pca(tok, bits, off, adj2, l2norm=True)["out"]
    assert torch.equal(out, ref)
    assert not torch.equal(out, eng.seg_vlad_pca(tok, bits, off, adj, l2norm=True)["out"])
    # a begin without an end is refused, a cancel frees the context
    h2 = eng.describe_begin(masks, tok, off, H, W, 14, 3, pca=True)
    from revisit_anything_amd._lib import SEGVLAD_ERR_STATE, SegVLADError

    with pytest.raises(SegVLADError) as ei:
        eng.describe_begin(masks, tok, off, H, W, 14, 3, pca=True)
    assert ei.value.code == SEGVLAD_ERR_STATE
    eng.describe_cancel(h2)
    assert torch.equal(eng.describe(masks, tok, off, H, W, 14, 3)["out"], eng.seg_vlad_pca(tok, bits, off, adj, l2norm=True)["out"])
    eng.close()


@pytest.mark.parametrize("case", ["pca_arith_fp32", "kd_not_a_multiple_of_32"])
def test_pipeline_describe_falls_back_when_the_split_call_cannot_take_the_projection(case):
    """segvlad_describe_begin needs the fp16x3 form of the PCA model; with option pca_arith=fp32, or with
    K*D not a multiple of 32 (no split planes at all), it reports SEGVLAD_ERR_LIMIT and pipeline.describe takes the separate
    entry points, whose images_pca has the plain-projection form -- as it did before the three-call path existed."""
    from revisit_anything_amd._lib import SEGVLAD_ERR_LIMIT, SegVLADError
    from revisit_anything_amd.pipeline import SegVLADPipeline

    if case == "pca_arith_fp32":
        eng, C = _setup(seed=4)
        eng.set_option("pca_arith", "fp32")
    else:
        eng, C = _setup(K=5, D=36, P=16, seed=5)        # K*D = 180: not a multiple of 32
    H, W = 112, 140
    tok, masks, off = _batch(C, 3, [6, 9, 5], H, W, seed=91)
    with pytest.raises(SegVLADError) as ei:
        eng.describe_begin(masks, tok, off, H, W, 14, 2, pca=True)
    assert ei.value.code == SEGVLAD_ERR_LIMIT
    fused = SegVLADPipeline(eng, H, W, 14, order=2, use_pca=True)
    step = SegVLADPipeline(eng, H, W, 14, order=2, use_pca=True, host_adjacency=True)
    a = fused.describe(tok, masks, off)
    b = step.describe(tok, masks, off)
    assert torch.equal(a, b) and torch.isfinite(a).all() and a.shape == (20, eng.P)
    eng.close()


def test_other_entry_points_refuse_while_a_describe_is_open():
    """between begin and end the per-batch scratch belongs to the open batch; segvlad_images / _images_pca /
    _cluster_aggregate / _incidence / _adjacency return SEGVLAD_ERR_STATE instead of describing another batch with its offsets,
    and the open describe still ends with the right rows."""
    from revisit_anything_amd._lib import SEGVLAD_ERR_STATE, SegVLADError

    eng, C = _setup(seed=6)
    H, W = 112, 140
    tok, masks, off = _batch(C, 3, [7, 5, 9], H, W, seed=13)
    tok2, masks2, off2 = _batch(C, 2, [4, 6], H, W, seed=14)
    bits, cent = eng.incidence_centroids(masks, H, W, 14)
    adj, _ = eng.adjacency_flagged(cent, off, 3, device_flags=True)
    ref = eng.seg_vlad_pca(tok, bits, off, adj, l2norm=True)["out"]
    bits2, cent2 = eng.incidence_centroids(masks2, H, W, 14)
    adj2, _ = eng.adjacency_flagged(cent2, off2, 3, device_flags=True)
    h = eng.describe_begin(masks, tok, off, H, W, 14, 3, pca=True)
    for call in (lambda: eng.seg_vlad(tok2, bits2, off2, adj2), lambda: eng.seg_vlad_pca(tok2, bits2, off2, adj2),
                 lambda: eng.incidence_centroids(masks2, H, W, 14), lambda: eng.adjacency_flagged(cent2, off2, 3, device_flags=True)):
        with pytest.raises(SegVLADError) as ei:
            call()
        assert ei.value.code == SEGVLAD_ERR_STATE
    assert torch.equal(eng.describe_end(h, None, None, l2norm=True)["out"], ref)
    assert torch.equal(eng.seg_vlad_pca(tok, bits, off, adj, l2norm=True)["out"], ref)     # and the context is free again
    eng.close()
