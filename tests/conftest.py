import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# Guard mode for the whole GPU run (csrc/ctx.h: DevBuf; tests/test_gpu_guard.py): fenced, exact-size device buffers whose
# fences are checked after every API call -- the driver's `pytest -m gpu` shares one context per module, and a grow-only
# scratch buffer sized by an earlier, larger call hides an under-sized request (commit 987adcb) unless the fence moves with
# the request.  SEGVLAD_GUARD=0 in the environment switches it off (timing runs; bench.py's own subprocesses do that).
os.environ.setdefault("SEGVLAD_GUARD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_perf: timing / rate assertions on a real MI355X (box dependent; never part of -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` (or `-m gpu_perf`) on a box without a GPU must fail loudly rather than silently skip: only auto-skip the
    # markers the user did NOT ask for explicitly (`-m "not gpu"` asks for neither).
    if _gpu_available():
        return
    import re

    expr = config.getoption("-m") or ""
    asked = {m for m in ("gpu", "gpu_perf") if re.search(r"(?<!not )\b%s\b" % m, expr)}
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        for m in ("gpu", "gpu_perf"):
            if m in it.keywords and m not in asked:
                it.add_marker(skip)


def engine_scope(fixture_name, config):
    """Scope of the GPU test modules' `eng` fixture: one context per module (fast) -- or, with SEGVLAD_FRESH_ENGINE=1, a
    FRESH context for every test: the library's scratch buffers only ever grow, so a context that an earlier, larger call
    has sized hides under-allocations (one was found that way: the exact level's distance block of a single-image search)."""
    return "function" if os.environ.get("SEGVLAD_FRESH_ENGINE") else "module"
