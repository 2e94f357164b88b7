"""DINOv2 value-facet producer on a tiny random-initialised backbone: the hook returns what the
reference's extractor returns (value facet of layer L, CLS dropped, utilities.py:219-288), the pre-processing follows
getAnyLocFt / process_single_DINO (func_vpr.py:489-506, 549-562), and the output has the ``ift_dino`` layout the hot
path consumes."""
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
