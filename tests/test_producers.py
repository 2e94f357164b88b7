"""DINOv2 value-facet producer on a tiny random-initialised backbone: the hook returns what the
reference's extractor returns (value facet of layer L, CLS dropped, utilities.py:219-288), the pre-processing follows
getAnyLocFt / process_single_DINO (func_vpr.py:489-506, 549-562), and the output has the ``ift_dino`` layout the hot
path consumes."""
import os

import numpy as np
import pytest
import torch

from revisit_anything_amd import producers as pr


@pytest.fixture(scope="module")
def tiny():
    torch.manual_seed(0)
    ex = pr.DinoV2ValueFacet.from_config("small", layer=1, hidden_size=48, num_hidden_layers=3, num_attention_heads=3,
                                         image_size=56)
    yield ex
    ex.close()


def test_value_facet_is_the_last_third_of_a_fused_qkv(tiny):
    """The reference hooks a fused qkv linear and keeps output[..., 2D:3D]; with separate q, k, v linears that is the
    output of the value linear on the same input (the layer-normed hidden state entering layer L)."""
    x = torch.randn(2, 3, 56, 70)
    tok = tiny(x)
    assert tok.shape == (2, 4 * 5, 48)
    m = tiny.model
    with torch.no_grad():
        hs = m(pixel_values=x, output_hidden_states=True).hidden_states[tiny.layer]      # input of layer L
        lay = m.encoder.layer[tiny.layer]
        z = lay.norm1(hs)
        att = lay.attention.attention
        w = torch.cat([att.query.weight, att.key.weight, att.value.weight])              # the fused layout of the hub model
        b = torch.cat([att.query.bias, att.key.bias, att.value.bias])
        qkv = torch.nn.functional.linear(z, w, b)
    assert torch.allclose(tok, qkv[:, 1:, 2 * 48:], atol=1e-6)
    assert not torch.allclose(tok.norm(dim=-1), torch.ones(2, 20), atol=1e-3)            # norm_descs=False


def test_image_to_tokens_layout_and_preprocessing(tiny):
    rng = np.random.Generator(np.random.PCG64(5))
    img = rng.integers(0, 256, (60, 75, 3), dtype=np.uint8)                             # not a multiple of 14
    out = pr.image_to_tokens(img, tiny)
    assert out.shape == (1, 48, 4, 5) and out.dtype == torch.float32                     # [1, D, h, w] = ift_dino
    # manual: ToTensor, ImageNet norm, CenterCrop((56, 70)) with torchvision's offsets (2, 2), tokens row-major
    x = torch.from_numpy(img).permute(2, 0, 1).float() / 255.0
    x = (x - torch.tensor(pr.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(pr.IMAGENET_STD).view(3, 1, 1)
    x = x[:, 2:58, 2:72][None]
    ref = tiny(x).reshape(1, 4, 5, 48).permute(0, 3, 1, 2)
    assert torch.allclose(pr.image_to_tokens(img, tiny, normalize=False), ref, atol=1e-6)   # getAnyLocFt(upsample=False)
    # process_single_DINO L2-normalises over the channel axis before the map is stored (func_vpr.py:561)
    assert torch.allclose(out, torch.nn.functional.normalize(ref, dim=1), atol=1e-6)
    assert torch.allclose(out.norm(dim=1), torch.ones(1, 4, 5), atol=1e-5)
    # the 17places geometry: 480 x 640 -> 476 x 630 -> 34 x 45 = 1530 tokens (bench.py's N)
    c = pr.center_crop_to_patches(torch.zeros(3, 480, 640))
    assert c.shape == (3, 476, 630) and (476 // 14) * (630 // 14) == 1530
    # resize branch of process_single_DINO
    out2 = pr.image_to_tokens(img, tiny, {"resize": True, "desired_width": 84, "desired_height": 56})
    assert out2.shape == (1, 48, 4, 6)


def test_dino_given_image_crops_rmin_rows_then_resizes(tiny):
    """func_vpr.py:626-644: the loader-side ``[rmin:, :, :]`` crop happens BEFORE the resize; BGR in, unit tokens out."""
    rng = np.random.Generator(np.random.PCG64(8))
    bgr = rng.integers(0, 256, (90, 100, 3), dtype=np.uint8)
    cfg = {"rmin": 20, "desired_width": 70, "desired_height": 56}
    t = pr.dino_given_image(tiny, bgr, cfg)
    assert t.shape == (1, 48, 4, 5) and t.device.type == "cpu"
    assert torch.allclose(t.norm(dim=1), torch.ones(1, 4, 5), atol=1e-5)
    rgb_crop = np.ascontiguousarray(bgr[20:, :, ::-1])
    ref = pr.image_to_tokens(rgb_crop, tiny, {"resize": True, "desired_width": 70, "desired_height": 56})
    assert torch.allclose(t, ref.cpu(), atol=1e-6)
    img_p, t2 = pr.process_single_DINO({"resize": False}, bgr[20:76, :70], tiny)
    assert img_p.shape == (56, 70, 3) and np.array_equal(img_p, bgr[20:76, :70, ::-1]) and t2.shape == (1, 48, 4, 5)


def test_input_validation(tiny):
    with pytest.raises(ValueError):
        tiny(torch.zeros(1, 3, 50, 70))
    with pytest.raises(ValueError):
        pr.image_to_tokens(np.zeros((60, 75, 3), np.float32), tiny)
    with pytest.raises(ValueError):
        pr.DinoV2ValueFacet.from_config("small", layer=7, hidden_size=48, num_hidden_layers=3, num_attention_heads=3)


def test_tokens_feed_the_store_and_the_driver(tiny, tmp_path):
    from revisit_anything_amd import driver, store as st

    rng = np.random.Generator(np.random.PCG64(6))
    img = rng.integers(0, 256, (56, 70, 3), dtype=np.uint8)
    tok = pr.image_to_tokens(img, tiny)
    st.write_dino(str(tmp_path / "d"), "a.jpg", tok.numpy())
    st.write_masks(str(tmp_path / "m"), "a.jpg", rng.random((3, 28, 35)) < 0.4)
    t, m = driver.load_image_inputs(st.FeatureStore(str(tmp_path / "d"), "dino"), st.FeatureStore(str(tmp_path / "m"), "masks"), "a.jpg")
    assert t.shape == (48, 20) and m.shape == (3, 28, 35)
    assert np.allclose(t, tok.numpy().reshape(48, 20))


# ---- SAM automatic masks (f3, second half) ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny_sam():
    from transformers import SamConfig, SamModel

    torch.manual_seed(7)
    cfg = SamConfig(vision_config=dict(hidden_size=32, output_channels=16, num_hidden_layers=2, num_attention_heads=2, image_size=128,
                                       patch_size=16, window_size=4, global_attn_indexes=[1], mlp_dim=64, num_pos_feats=8),
                    prompt_encoder_config=dict(hidden_size=16, image_size=128, patch_size=16, mask_input_channels=4),
                    mask_decoder_config=dict(hidden_size=16, mlp_dim=32, num_hidden_layers=2, num_attention_heads=2,
                                             iou_head_hidden_dim=16, iou_head_depth=2))
    m = SamModel(cfg)
    with torch.no_grad():                      # the default initialiser (std 1e-10) gives constant outputs
        for p in m.parameters():
            p.normal_(0.0, 0.35)
    return m


def test_sam_helpers_match_their_definitions():
    g = pr.build_point_grid(2)
    assert np.allclose(g, [[0.25, 0.25], [0.75, 0.25], [0.25, 0.75], [0.75, 0.75]])           # (x, y), x fastest
    assert pr.build_point_grid(32).shape == (1024, 2)
    rng = np.random.Generator(np.random.PCG64(3))
    lg = torch.from_numpy(rng.standard_normal((5, 12, 9)).astype(np.float32)) * 2
    st = pr.stability_score(lg, 0.0, 1.0).numpy()
    ref = [(lg[i] > 1).sum().item() / (lg[i] > -1).sum().item() for i in range(5)]
    assert np.allclose(st, ref)
    m = torch.zeros(3, 8, 10, dtype=torch.bool)
    m[0, 2:5, 3:9] = True
    m[1, 7, 0] = True
    assert pr.mask_boxes(m).tolist() == [[3, 2, 8, 4], [0, 7, 0, 7], [0, 0, 0, 0]]           # XYXY inclusive; empty -> zeros
    # NMS against the textbook loop
    b = torch.from_numpy(rng.uniform(0, 50, (60, 2)).astype(np.float32))
    boxes = torch.cat([b, b + torch.from_numpy(rng.uniform(5, 30, (60, 2)).astype(np.float32))], dim=1)
    scores = torch.from_numpy(rng.random(60).astype(np.float32))
    keep = pr.box_nms(boxes, scores, 0.3).tolist()

    def iou(p, q):
        iw = max(0.0, min(p[2], q[2]) - max(p[0], q[0]))
        ih = max(0.0, min(p[3], q[3]) - max(p[1], q[1]))
        inter = iw * ih
        return inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter)

    want = []
    for i in np.argsort(-scores.numpy(), kind="stable"):
        if all(iou(boxes[i].tolist(), boxes[j].tolist()) <= 0.3 for j in want):
            want.append(int(i))
    assert keep == want


def _stub_extractor(seed: int, D: int):
    """The stand-in extractor the DINO fixtures were generated with (tools/make_golden.py::stub_extractor)."""
    g = torch.Generator().manual_seed(seed)
    Wm = torch.randn(6, D, generator=g)

    def f(x):
        p = torch.nn.functional.avg_pool2d(x, 14)
        p2 = torch.nn.functional.avg_pool2d(x * x, 14)
        t = torch.cat([p, p2], 1).flatten(2).transpose(1, 2)
        return torch.tanh(t @ Wm) + 0.25 * (t @ Wm)

    return f


def test_sam_helpers_equal_the_vendored_amg_utilities(golden_dir):
    """The reference's OWN utilities (sam/segment_anything/utils/amg.py: build_point_grid :179-186,
    calculate_stability_score :156-176, batched_mask_to_box :303-338, box_xyxy_to_xywh :91-96), executed by
    tools/make_golden.py where they lie -- not expectations written by hand."""
    z = np.load(os.path.join(golden_dir, "producers.npz"))
    for n in (1, 3, 16, 32):
        assert np.array_equal(pr.build_point_grid(n), z[f"grid_{n}"])                          # bit-identical float64
    for j in range(3):
        thr, off = (float(v) for v in z[f"stab_{j}_args"])
        got = pr.stability_score(torch.from_numpy(z[f"stab_{j}_logits"]), thr, off).numpy()
        assert got.shape == z[f"stab_{j}_out"].shape and np.array_equal(got, z[f"stab_{j}_out"])
    shape = tuple(int(v) for v in z["box_masks_shape"])
    m = torch.from_numpy(np.unpackbits(z["box_masks"], axis=-1)[..., :shape[-1]].astype(bool))
    b = pr.mask_boxes(m)
    assert np.array_equal(b.numpy(), z["box_out"].astype(np.int64))                            # empty -> zeros, full, 1 pixel
    assert np.array_equal(pr.mask_boxes(m.reshape(3, 3, *shape[1:])).numpy(), z["box_out_nd"].astype(np.int64))
    assert np.array_equal(pr.mask_boxes(m[0][None]).numpy()[0], z["box_out_2d"].astype(np.int64))
    # the record's XYWH box (automatic_mask_generator.py:157: box_xyxy_to_xywh) as generate() forms it
    xywh = torch.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], dim=1).numpy()
    assert np.array_equal(xywh, z["box_xywh"].astype(np.int64))


def test_dino_tokens_equal_getAnyLocFt_and_process_single_DINO(golden_dir):
    """func_vpr.py:489-506 (getAnyLocFt) and :549-562 (process_single_DINO) executed with a seeded stub extractor on
    three image sizes (two of them not multiples of 14: CenterCrop offsets), against image_to_tokens / process_single_DINO
    here fed with the same extractor."""
    z = np.load(os.path.join(golden_dir, "producers.npz"))
    for j in range(3):
        _, H, W, D, seed = (int(v) for v in z[f"dino_{j}_args"])
        img_bgr = z[f"dino_{j}_img"]
        assert img_bgr.shape == (H, W, 3)
        ext = _stub_extractor(seed, D)
        img_p, tok = pr.process_single_DINO({"resize": False}, img_bgr, ext)
        assert np.array_equal(img_p, img_bgr[:, :, ::-1])
        assert tok.shape == z[f"dino_{j}_feat_norm"].shape == (1, D, H // 14, W // 14)
        assert np.abs(tok.numpy() - z[f"dino_{j}_feat_norm"]).max() < 2e-6                     # same ops; fp32 reduction orders
        raw = pr.image_to_tokens(np.ascontiguousarray(img_bgr[:, :, ::-1]), ext, None, normalize=False)
        assert np.abs(raw.numpy() - z[f"dino_{j}_raw"]).max() < 2e-5
        assert np.abs(np.linalg.norm(tok.numpy(), axis=1) - 1.0).max() < 1e-5


def test_sam_auto_masks_records_and_half_resolution(tiny_sam, tmp_path):
    from revisit_anything_amd import store as st

    gen = pr.SamAutoMasks(tiny_sam, points_per_side=6, points_per_batch=16, pred_iou_thresh=-1.0, stability_score_thresh=0.0,
                          box_nms_thresh=0.7)
    rng = np.random.Generator(np.random.PCG64(11))
    img = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    x, (nh, nw) = gen.preprocess(img)
    assert tuple(x.shape) == (1, 3, 128, 128) and (nh, nw) == (96, 128)                      # longest side -> 128, padded
    recs = gen.generate(img)
    assert len(recs) >= 1
    ious = [r["predicted_iou"] for r in recs]
    assert ious == sorted(ious, reverse=True)                                                 # batched_nms order
    for r in recs:
        seg = r["segmentation"]
        assert seg.dtype == bool and seg.shape == (96, 128) and r["area"] == int(seg.sum()) > 0
        ys, xs = np.nonzero(seg)
        assert r["bbox"] == [int(xs.min()), int(ys.min()), int(xs.max() - xs.min()), int(ys.max() - ys.min())]
        assert r["crop_box"] == [0, 0, 128, 96] and len(r["point_coords"]) == 1 and 0 <= r["stability_score"] <= 1
    # surviving boxes do not overlap by more than the NMS threshold
    bx = torch.tensor([[r["bbox"][0], r["bbox"][1], r["bbox"][0] + r["bbox"][2], r["bbox"][1] + r["bbox"][3]] for r in recs])
    assert len(pr.box_nms(bx, torch.tensor(ious), 0.7)) == len(recs)
    # thresholds act: the default 0.88 predicted-IoU cut removes everything a random network proposes
    strict = pr.SamAutoMasks(tiny_sam, points_per_side=4, pred_iou_thresh=1e9)
    assert strict.generate(img) == []
    # the reference runs SAM at HALF the configured resolution (place_rec_SAM_DINO.py:61) on the BGR frame
    cfg = {"rmin": 0, "desired_width": 128, "desired_height": 96}
    segs, recs2 = pr.masks_given_image(gen, img[:, :, ::-1], cfg)
    assert len(segs) == len(recs2) >= 1 and all(s.shape == (48, 64) for s in segs)
    full, _ = pr.masks_given_image(gen, img[:, :, ::-1], cfg, mask_full_resolution=True)
    assert all(s.shape == (96, 128) for s in full)
    # records go into the mask store with the reference's nesting and come back through preload_masks
    st.write_masks(str(tmp_path / "m"), "a.jpg", recs2)
    from revisit_anything_amd.func_vpr import preload_masks
    back = preload_masks(st.FeatureStore(str(tmp_path / "m"), "masks"), "a.jpg")
    assert len(back) == len(recs2) and all(np.array_equal(b, r["segmentation"]) for b, r in zip(back, recs2))
    assert np.array_equal(pr.resize_like_cv2(img, 128, 96), img)                              # identity size: untouched


def test_sam_generate_flow_equals_the_vendored_generator(golden_dir):
    """SamAutoMasks.generate (batching, predicted-IoU cut, stability cut, threshold, boxes, NMS by predicted IoU, records)
    against the reference's vendored SamAutomaticMaskGenerator.generate, both on the stub decoder of tests/sam_stub.py:
    same records, same order, masks bit for bit (tools/make_golden.py::gen_sam_generate)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sam_stub import stub_predict

    z = np.load(os.path.join(golden_dir, "sam_generate.npz"))

    class Stubbed(pr.SamAutoMasks):
        def __init__(self, seed, **kw):   # no network: the two model-facing members are replaced below
            self.device = torch.device("cpu")
            self.points_per_batch = int(kw.pop("points_per_batch"))
            self.pred_iou_thresh, self.stability_score_thresh = float(kw.pop("pred_iou_thresh")), float(kw.pop("stability_score_thresh"))
            self.stability_score_offset, self.box_nms_thresh = float(kw.pop("stability_score_offset")), float(kw.pop("box_nms_thresh"))
            self.min_mask_region_area, self.mask_threshold = 0, float(kw.pop("mask_threshold"))
            self.point_grid = pr.build_point_grid(kw.pop("points_per_side"))
            self.seed = seed
            assert not kw

        def _embed(self, img_rgb):
            return None

        def _predict(self, state, pts_img, out_hw):
            lg, io = stub_predict(pts_img, out_hw[0], out_hw[1], self.seed)
            return torch.from_numpy(lg), torch.from_numpy(io)

    for c in range(int(z["n_cases"])):
        H, W, pps, ppb, seed = (int(v) for v in z[f"c{c}_args"])
        thr, piou, stab, off, nms = (float(v) for v in z[f"c{c}_thr"])
        gen = Stubbed(seed, points_per_side=pps, points_per_batch=ppb, pred_iou_thresh=piou, stability_score_thresh=stab,
                      stability_score_offset=off, box_nms_thresh=nms, mask_threshold=thr)
        recs = gen.generate(np.zeros((H, W, 3), dtype=np.uint8))
        want_seg = np.unpackbits(z[f"c{c}_seg"], axis=-1)[..., :W].astype(bool)
        assert len(recs) == len(want_seg) >= 3, (c, len(recs), len(want_seg))
        assert np.array_equal(np.stack([r["segmentation"] for r in recs]), want_seg)           # same masks in the same ORDER
        assert [r["area"] for r in recs] == z[f"c{c}_area"].tolist()
        assert [r["bbox"] for r in recs] == z[f"c{c}_bbox"].tolist()
        assert [r["crop_box"] for r in recs] == z[f"c{c}_crop"].tolist()
        assert np.array_equal(np.array([r["predicted_iou"] for r in recs]), z[f"c{c}_iou"])
        assert np.array_equal(np.array([r["stability_score"] for r in recs]), z[f"c{c}_stab"])
        assert np.array_equal(np.array([r["point_coords"][0] for r in recs]), z[f"c{c}_pts"])
