#!/usr/bin/env python3
"""Captures the call signatures of the reference's hot-path surface from its source text with
`ast` -- parameter names, order and default literals, nothing else -- into tests/golden/signatures.json.  Runs only in
the build container (it reads /root/reference); the committed JSON is what tests/test_surface.py checks the drop-in
modules against."""
import ast
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "signatures.json")

WANT = {
    "func_vpr.py": ["preload_masks", "getIdxSingleFast", "nbrMasksAGGFastSingle", "seg_vlad_gpu_single",
                    "seg_vlad_gpu_single_img", "vlad_single", "vlad_matmuls_per_cluster", "apply_pca_transform_from_pkl",
                    "apply_pca_transform_from_pkl_numpy", "normalizeFeat", "get_matches", "calc_recall", "weighted_borda_count",
                    "aggFt", "get_recall"],
    "place_rec_main.py": ["recall_segloc"],
}


def sig(fn: ast.FunctionDef):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(pos) - len(a.defaults)) + [ast.literal_eval(d) if isinstance(d, ast.Constant) else ast.unparse(d)
                                                       for d in a.defaults]
    return {"params": pos, "defaults": {p: d for p, d in zip(pos, defaults) if d is not None or p in pos[len(pos) - len(a.defaults):]},
            "n_required": len(pos) - len(a.defaults), "vararg": a.vararg.arg if a.vararg else None,
            "kwarg": a.kwarg.arg if a.kwarg else None}


def main():
    out = {}
    for fname, names in WANT.items():
        tree = ast.parse(open(os.path.join(REF, fname)).read())
        found = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
        for n in names:
            if n not in found:
                raise SystemExit(f"{fname}: {n} not found")
            out[f"{fname}:{n}@{found[n].lineno}"] = sig(found[n])
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"wrote {OUT} ({len(out)} signatures)")


if __name__ == "__main__":
    main()
