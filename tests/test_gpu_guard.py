"""Guard mode (SEGVLAD_GUARD=1; csrc/ctx.h DevBuf): every device buffer of a context sits between poison fences, the back
fence of a per-call scratch buffer right behind the bytes of the CURRENT request, and every API call ends with a check of
all fences.  The library's scratch only ever grows, so without this a request that an earlier, larger one already covers
can never fail (round 3 found an out-of-bounds write that way, and only on a fresh context).  conftest.py switches the
mode on for the whole `-m gpu` run; here the guard itself is tested: it must trip on a buffer that is deliberately a few
bytes short, name it, stay quiet otherwise, and change no result."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(guard: bool):
    from revisit_anything_amd.engine import SegVLADEngine

    old = os.environ.get("SEGVLAD_GUARD")
    os.environ["SEGVLAD_GUARD"] = "1" if guard else "0"
    try:
        return SegVLADEngine(0)
    finally:
        if old is None:
            os.environ.pop("SEGVLAD_GUARD", None)
        else:
            os.environ["SEGVLAD_GUARD"] = old


def _data(n=40_000, d=128, nq=300, seed=5):
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    R = torch.nn.functional.normalize(torch.randn(n, d, device="cuda:0", generator=g), dim=1)
    Q = torch.nn.functional.normalize(R[:nq] + 0.05 * torch.randn(nq, d, device="cuda:0", generator=g), dim=1)
    return R, Q


def test_guard_is_quiet_and_changes_nothing():
    from revisit_anything_amd._lib import SegVLADError

    R, Q = _data()
    plain, guarded = _engine(False), _engine(True)
    with pytest.raises(SegVLADError):          # the test hook only exists on a guarded context
        plain.set_option("guard_undersize", "s_qnorm:4")
    res = []
    for eng in (plain, guarded):
        eng.db_add(R)
        for nq in (300, 50, 300, 7):           # shrinking and growing requests on the same (grow-only) buffers
            res.append(eng.search(Q[:nq], 10))
        eng.synchronize()                      # (guarded: checks every fence once more)
    half = len(res) // 2
    for (d2a, ia), (d2b, ib) in zip(res[:half], res[half:]):
        assert torch.equal(ia, ib) and torch.equal(d2a, d2b)


@pytest.mark.parametrize("victim", ["s_qnorm:4", "s_cand_cnt:4", "s_ref_cnt:4"])
def test_guard_trips_on_a_buffer_a_few_bytes_short(victim):
    """The fence of one scratch buffer is put a few bytes EARLY, so that a correct kernel's last words land in it: the call
    must fail with SEGVLAD_ERR_STATE and name the buffer -- first with a smaller request AFTER a larger one (the case a
    grow-only buffer hides), and the failure must stick."""
    from revisit_anything_amd import _lib

    R, Q = _data()
    eng = _engine(True)
    eng.db_add(R)
    eng.search(Q, 10)                          # 300 queries: every buffer now holds more than the next call asks for
    eng.set_option("guard_undersize", victim)
    with pytest.raises(_lib.SegVLADError) as ei:
        eng.search(Q[:50], 10)                 # 50 queries: inside the capacity, beyond the (shortened) request
    assert ei.value.code == _lib.SEGVLAD_ERR_STATE
    msg = str(ei.value)
    assert "guard" in msg and victim.split(":")[0] in msg, msg
    with pytest.raises(_lib.SegVLADError):     # sticky: the context stays failed
        eng.synchronize()


def test_guard_covers_the_describe_path():
    """A batch of images through incidence -> adjacency -> seg-VLAD on a guarded context (exact-size buffers, fences checked
    after every call) equals the same calls on a plain one."""
    from revisit_anything_amd import synth

    K, D, H, W, S, B = 16, 128, 112, 140, 9, 3
    N = (H // 14) * (W // 14)
    C = synth.make_vocab(K, D, seed=1)
    toks = np.stack([synth.make_tokens(C, N, seed=10 + b, noise=0.2) for b in range(B)])
    masks = np.concatenate([synth.make_blob_masks(S, H // 2, W // 2, seed=20 + b) for b in range(B)])
    seg_off = np.arange(B + 1, dtype=np.int32) * S
    outs = []
    for guard in (False, True):
        eng = _engine(guard)
        eng.set_vocab(C)
        bits = eng.incidence(masks, H, W)
        outs.append(eng.seg_vlad(toks, bits, seg_off, None)["out"].cpu().numpy())
        eng.synchronize()
    assert np.array_equal(outs[0], outs[1])
