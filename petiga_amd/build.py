"""Builds libpetiga_amd.so (HIP kernels + C++ host + C ABI) in-tree for gfx950 with hipcc.

engine.hip is compiled several times in parallel: once as the main unit (C ABI, set-up, drivers, pencil kernels) and,
with -DIGX_TU_DISPATCH -DIGX_TU_DIM=d -DIGX_TU_GROUP=g, as units that hold only the element-kernel instantiations of one
dimension / form group (see the top of engine.hip).  One translation unit took 6 minutes; the units take about 2."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
# IGX_BUILD_DEBUG=1: the experiment build (-DIGX_DEBUG: cycle stamps, phase switches) into its own objects / library
DEBUG = os.environ.get("IGX_BUILD_DEBUG", "") not in ("", "0")
OBJ = os.path.join(HERE, "_build_debug" if DEBUG else "_build")
SO = os.path.join(HERE, "libpetiga_amd_debug.so" if DEBUG else "libpetiga_amd.so")
HEADERS = ["igx.hpp", "forms.hpp", "generic_kernel.hpp", "feature_mfma.hpp", "first_touch.hpp", "gram_mfma.hpp", "exchange.hpp", "fileio.hpp",
           os.path.join("..", "..", "include", "petiga_amd.h")]
# (object name, source, extra flags)
UNITS = [("engine_main.o", "engine.hip", []),
         ("engine_d1.o", "engine.hip", ["-DIGX_TU_DISPATCH", "-DIGX_TU_DIM=1", "-DIGX_TU_GROUP=-1"]),
         ("engine_d2.o", "engine.hip", ["-DIGX_TU_DISPATCH", "-DIGX_TU_DIM=2", "-DIGX_TU_GROUP=-1"]),
         ("engine_d3g0.o", "engine.hip", ["-DIGX_TU_DISPATCH", "-DIGX_TU_DIM=3", "-DIGX_TU_GROUP=0"]),
         ("engine_d3g1.o", "engine.hip", ["-DIGX_TU_DISPATCH", "-DIGX_TU_DIM=3", "-DIGX_TU_GROUP=1"]),
         ("engine_d3g2.o", "engine.hip", ["-DIGX_TU_DISPATCH", "-DIGX_TU_DIM=3", "-DIGX_TU_GROUP=2"]),
         ("host.o", "host.cpp", [])]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"] + (["-DIGX_DEBUG"] if DEBUG else [])


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = sorted(set(u[1] for u in UNITS)) + HEADERS
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in srcs)


def build(force=False, verbose=False):
    if not (force or stale()):
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libpetiga_amd.so")
    os.makedirs(OBJ, exist_ok=True)
    procs = []
    for obj, src, extra in UNITS:
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(OBJ, obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((obj, out))
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(o for o, _ in failed) + "\n" + "\n".join(t[-4000:] for _, t in failed))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [os.path.join(OBJ, u[0]) for u in UNITS]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
