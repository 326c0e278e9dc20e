"""Builds libpetiga_amd.so (HIP kernels + C++ host + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libpetiga_amd.so")
SOURCES = ["engine.hip", "host.cpp"]
HEADERS = ["igx.hpp", "forms.hpp", "generic_kernel.hpp", "gram_mfma.hpp", "exchange.hpp", "fileio.hpp", os.path.join("..", "..", "include", "petiga_amd.h")]


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not (force or stale()):
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libpetiga_amd.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-o", SO] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
