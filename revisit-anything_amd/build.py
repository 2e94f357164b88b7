"""Builds libsegvlad_hip.so (HIP, gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsegvlad_hip.so")
SOURCES = ["api.hip", "vlad_kernels.hip", "gemm_kernels.hip", "select_kernels.hip", "vote_kernels.hip",
           "knn_filter_kernels.hip", "gemm_f16x3_kernels.hip", "project_kernels.hip", "comm.hip", "refine_group_kernels.hip", "small_pass_kernels.hip", "kmeans_kernels.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "segvlad.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, ablations: bool = False) -> str:
    """ablations=True (development only): the build with the timing ablations (WRONG results) and the phase timers, as a
    SECOND library next to the product's (lib/libsegvlad_hip_abl.so; load it with SEGVLAD_LIB_PATH)."""
    # (the environment switch only applies to an explicit `python build.py`: _lib.load() must always get the product library
    #  at LIB_PATH rebuilt, whatever the environment says -- ADVICE r04)
    lib_path = LIB_PATH.replace(".so", "_abl.so") if ablations else LIB_PATH
    if not ablations and not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", "_abl.o" if ablations else ".o"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result",
               "-c", os.path.join(CSRC, src), "-o", obj]
        if ablations:
            cmd.insert(1, "-DSEGVLAD_ABLATIONS")
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                ablations="--ablations" in sys.argv or bool(os.environ.get("SEGVLAD_BUILD_ABLATIONS"))))
