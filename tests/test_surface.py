"""Drop-in surface: every callable the reference driver uses must exist in the build's modules with
the same positional parameter names, order and default values.  The expected signatures were captured from the
reference's source with tools/make_signatures.py (names and default literals only) into tests/golden/signatures.json."""
import inspect
import json
import os

import pytest

G = os.path.join(os.path.dirname(__file__), "golden", "signatures.json")
EXPECTED = json.load(open(G))


def _module_for(ref_file):
    if ref_file == "func_vpr.py":
        from revisit_anything_amd import func_vpr
        return func_vpr
    from revisit_anything_amd import place_rec
    return place_rec


@pytest.mark.parametrize("key", sorted(EXPECTED))
def test_signature_matches_reference(key):
    ref_file, rest = key.split(":")
    name = rest.split("@")[0]
    exp = EXPECTED[key]
    fn = getattr(_module_for(ref_file), name, None)
    assert fn is not None, f"{name} (reference {key}) is missing from the drop-in module"
    sig = inspect.signature(fn)
    params = [p for p in sig.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    assert [p.name for p in params] == exp["params"], key
    n_req = sum(p.default is inspect.Parameter.empty for p in params)
    assert n_req == exp["n_required"], key
    for p in params:
        if p.name in exp["defaults"]:
            assert p.default == exp["defaults"][p.name], (key, p.name, p.default, exp["defaults"][p.name])
