"""Deep rows (d >= 4096) on the persistent 256 x 256 filter kernel with FLUSHED accumulation blocks (knn_f16_filter_kernel KFL,
option f16_deep_cfg -1 / 5; DESIGN.md 4 item 6 "Round 6b"): every 4096 elements a wave adds its accumulators into a global scratch
slice with non-returning L2 atomics.  The search must return what the register-blocked kernel (f16_deep_cfg = 4) and the all-fp32
filter return, bit for bit -- at a depth without a flush (d = 4096: one block), with one (8192) and with two and a ragged last block
(d = 12 288 + 64 is not allowed: d % 64 == 0 and whole blocks are not required -- 10 240 = 2.5 blocks).  Run with `-m gpu`."""
import pytest
from conftest import engine_scope

pytestmark = pytest.mark.gpu


@pytest.fixture(scope=engine_scope)
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a ROCm device (no CPU fallback exists)"
    from revisit_anything_amd.engine import SegVLADEngine

    e = SegVLADEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("d", [4096, 8192, 10240])
def test_flushed_blocks_equal_register_blocks_and_fp32_filter(eng, d):
    import torch

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(9400 + d)
    n, nq, k = 70000, 1024, 40           # 4 query tiles x 257 database tiles of the complement level: the persistent kernel's 1024
    base = torch.nn.functional.normalize(torch.randn(n // 4, 1, d, device=dev, generator=g), dim=2)
    R = torch.nn.functional.normalize(base + (0.05 / d ** 0.5) * torch.randn(n // 4, 4, d, device=dev, generator=g), dim=2).reshape(-1, d).contiguous()
    del base
    src = torch.randint(0, n, (nq,), device=dev, generator=g)
    Q = torch.nn.functional.normalize(R[src] + (1.0 / d ** 0.5) * torch.randn(nq, d, device=dev, generator=g), dim=1).contiguous()
    Q[nq // 2:] = torch.nn.functional.normalize(torch.randn(nq - nq // 2, d, device=dev, generator=g), dim=1)   # un-planted half
    eng.db_reset()
    eng.db_add(R)
    try:
        eng.set_option("search_stats", 1)
        d2a, ida = eng.search(Q, k)
        sta = eng.search_stats()
        eng.set_option("f16_deep_cfg", 4)
        d2b, idb = eng.search(Q, k)
        stb = eng.search_stats()
        eng.set_option("knn_filter", "fp32")
        d2f, idf = eng.search(Q, k)
    finally:
        eng.set_option("knn_filter", "auto")
        eng.set_option("f16_deep_cfg", -1)
        eng.set_option("search_stats", 0)
    print(f"d={d}: flushed {sta}\n          register-blocked {stb}")
    assert sta["filter"] == "f16" and sta["levels"] >= 2 and sta["carry_rows"] == (n + 15) // 16 and sta["n_fallback"] == 0, sta
    assert torch.equal(ida, idb) and torch.equal(d2a, d2b)
    assert torch.equal(ida, idf) and torch.equal(d2a, d2f)
    assert (ida[:nq // 2, :4] // 4 == (src[:nq // 2] // 4)[:, None]).float().mean() > 0.99
    # the flushed kernel's margin is the larger one (kb = 4096 against 1024): its lists can only be longer
    assert sta["cand_sum"] >= stb["cand_sum"] and sta["refine_sum"] >= stb["refine_sum"], (sta, stb)
