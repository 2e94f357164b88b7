"""Builds tests/rccl_stub/librccl_stub.so (test infrastructure only; see rccl_stub.cpp).  The ROCm root comes from ROCM_PATH or
from the hipcc on PATH, the compiler from CXX / g++ / hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_stub.cpp")
SO = os.path.join(HERE, "librccl_stub.so")


def rocm_root() -> str:
    if os.environ.get("ROCM_PATH"):
        return os.environ["ROCM_PATH"]
    hipcc = shutil.which("hipcc")
    if hipcc:
        return os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    return "/opt/rocm"


def build(force: bool = False) -> str:
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(SRC):
        return SO
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("hipcc") or os.path.join(rocm_root(), "bin", "hipcc")
    root = rocm_root()
    subprocess.run([cxx, "-O2", "-fPIC", "-shared", "-std=c++17", "-w", "-D__HIP_PLATFORM_AMD__", f"-I{root}/include", SRC, "-o", SO,
                    f"-L{root}/lib", "-lamdhip64", "-lrt"], check=True)
    return SO


if __name__ == "__main__":
    print(build(force=True))
