#!/usr/bin/env python3
"""bench.py -- element stiffness assemblies per second, 3-D p=3 Poisson on 256^3 elements
(BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  N>1: either way works -- `python bench.py --gpus N ...` starts its own N ranks (launch_ranks: children of a process that has
  not touched the GPU) and relays rank 0's line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
  the ranks are the launcher's.

A step = one complete IGAComputeSystem-equivalent: zero A and b, form every local element's K_e/F_e,
apply the Dirichlet fix-up, scatter into the device CSR, and (N>1) reduce the ghost rows to their
owners over RCCL.  Inputs (1-D tables, pattern) are resident in HBM before the timed region.
The 256^3 mesh is split over the N ranks with PetIGA's own partition rule (strong scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_PER_ELEM = {3: 1572864, 2: 118098}        # 2*nen^2*nqp*dim (SURVEY 8d / BASELINE.md 3)
BYTES_PER_ELEM = {3: 2785, 2: 1019}            # compulsory CSR bytes per element
# the other BASELINE configs ("Same figures, other configs"): algorithmic flop and compulsory bytes per element
ALG = {"elasticity": (4718592, 25400), "cahnhilliard": (790000, 1009), "nsvms": (8388608, 44000)}
NOMINAL_MHZ = 2400.0
FP64_PEAK_TFLOPS = 78.6                        # MI355X fp64 vector = matrix peak (256 CU * 4 SIMD * 32 flop/clk * 2.4 GHz)
HBM_PEAK_GBS = 8000.0
KERNEL_TAG = "r05"                             # profiles/traffic.json must describe this round's kernel to be quoted


def physical_cores():
    """Physical cores of this host (unique (package, core) pairs; SMT siblings count once)."""
    seen = set()
    try:
        pkg = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if pkg is not None and core is not None:
                    seen.add((pkg, core))
                pkg = core = None
        if pkg is not None and core is not None:
            seen.add((pkg, core))
    except OSError:
        pass
    n = len(seen) or (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:                       # a container's CPU quota (cgroup v2 "cpu.max": "<quota> <period>" or "max <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# single-core rates of the oracle (elements/s), only to size the bounded sample
CPU_RATE_GUESS = {"poisson": 930.0, "elasticity": 350.0, "cahnhilliard": 6600.0, "nsvms": 85.0}      # (measured: profiles/r04_*_line.json)


def cpu_baseline(form, degree, seconds_target=15.0, geometry=False):
    """Times the CPU oracle (port of the reference loop) on this box's host cores on a bounded sample of the same
    workload.  One worker per PHYSICAL core; every worker assembles its own box of m^3 elements into its own local
    matrix -- what a rank of `mpiexec -n cores` does with its ghosted box (the reference's MatSetValuesLocal works on the
    rank-local rows) -- so no worker pays for a global pattern.  Reported: the measured all-core rate, the single-core
    rate and cores x single-core (the no-loss bound); the speed-up quoted next to it uses the larger of the two."""
    import multiprocessing as mp
    cores = physical_cores()
    rate1 = (CPU_RATE_GUESS["poisson"] if degree == 3 else 16000.0) if form == "poisson" else CPU_RATE_GUESS[form]
    m = int(round((rate1 * seconds_target) ** (1.0 / 3.0)))
    m = max(8, min(m, 60))
    # one core alone first (also warms the page cache / builds nothing: the .so is prebuilt)
    e1, t1 = _cpu_worker((form, degree, m, geometry))
    t0 = time.time()
    if cores > 1:
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(form, degree, m, geometry)] * cores)
    else:
        res = [(e1, t1)]
    wall = max(r[1] for r in res)
    elems = sum(r[0] for r in res)
    single = e1 / t1
    what = {"poisson": "3-D p=%d Poisson System (Dirichlet on 6 faces%s)" % (degree, ", the bench's rational NURBS map" if geometry else ""),
            "elasticity": "3-D p=3 Elasticity System (orc_form_elasticity; clamped face, u_x = 1 on the opposite one)",
            "cahnhilliard": "3-D p=2 C1 CahnHilliard IFunction + IJacobian (orc_form_ch_residual / orc_form_ch_tangent)",
            "nsvms": "3-D p=3 NavierStokesVMS IFunction + IJacobian on the bench's rational NURBS map (orc_form_ns_residual / orc_form_ns_tangent; axes 0, 2 periodic, no-slip on axis 1)"}[form]
    return dict(value=elems / wall, unit="elements/s", cores=cores, kind="port", single_core_value=single,
                cores_x_single_core=cores * single,
                logical_cpus=os.cpu_count(),
                sample="%s, %d^3 elements per core into a core-local matrix, %d physical cores at once "
                       "(oracle/igaoracle.c + igaforms.c, the reference's loop: full-order tabulation, scalar callback, search-insert); slowest core %.1f s, "
                       "one core alone %.1f s, pool wall %.1f s" % (what, m, cores, wall, t1, time.time() - t0))


def live_traffic(child_args, dom_name, timeout=300):
    """HBM-side bytes per launch of the dominant kernel, measured now: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, each
    with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) around a one-step child run of this same script.
    The kernel instantiation is picked ONCE (from the FETCH pass: the one that fetched most in total) and the WRITE pass reads the
    same name.  Returns (dict or None, note): raw FETCH_SIZE / WRITE_SIZE in bytes per launch and the guide's corrected figure
    (2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies a 128-byte read request as 64 bytes)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    key = dom_name.split("<")[0].split("(")[0]          # gram_pencil, block_pencil, state_pencil, band_pt, form_pencil, feature_assemble
    mean, name = {}, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="igx_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + child_args
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT)
            per = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    # the symbol may carry a variant suffix the engine's kernel name does not (gram_pencil_w6<...>, state_pencil_geo<...>)
                    if r["Counter_Name"] == counter and re.search(r"igx::" + re.escape(key) + r"(_\w+)?<", r["Kernel_Name"]):
                        per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
            if not per:
                return None, "no %s samples of %s" % (counter, key)
            if name is None:
                name = max(per, key=lambda k: sum(per[k]))       # (two-assembly steps: the IJacobian's kernel carries the traffic)
            if name not in per:
                return None, "the %s pass has no samples of %s" % (counter, name)
            mean[counter] = sum(per[name]) / len(per[name])
        except Exception as e:                              # a missing counter, a time-out: the line falls back to the replayed figure
            return None, "rocprofv3 pass failed: %r" % (e,)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = mean["FETCH_SIZE"] * 1024.0, mean["WRITE_SIZE"] * 1024.0
    return dict(corrected=2.0 * fetch + write, fetch_raw=fetch, write_raw=write, kernel=name.replace("void igx::", "")), \
        "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each around a one-step child run of this command; " \
        "traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch (the guide's gfx950 correction: a 128-byte read request is tallied as 64 bytes -- it holds " \
        "for the kernels of this library that read whole 128-byte lines, band_pt's blocks among them: reads + point records = 2 x FETCH_SIZE to 2 % there; " \
        "traffic_raw_fetch / traffic_raw_write are the uncorrected counters)"


def _bench_geometry(p, size, periodic):
    """The smooth rational map of `--geometry` (config 5's premise): the same control net for the engine and the oracle."""
    import numpy as np
    from petiga_amd.geometry import greville       # no tests/ (and with it no oracle module) on the GPU leg
    gv = []
    for i in range(3):
        U = (np.arange(-p, size + p + 1) / size) if periodic[i] else np.concatenate([[0.0] * (p + 1), np.arange(1, size) / size, [1.0] * (p + 1)])
        gv.append(greville(U, p))
    mesh = np.meshgrid(*gv[::-1], indexing="ij")[::-1]
    X = np.stack([m.copy() for m in mesh], axis=-1)
    X[..., 0] += 0.05 * np.sin(2 * np.pi * mesh[1])
    X[..., 1] += 0.05 * np.sin(2 * np.pi * mesh[2])
    W = 1.0 + 0.1 * np.cos(2 * np.pi * mesh[0])
    return X.reshape(-1, 3), W.reshape(-1)


def _cpu_worker(args):
    form, degree, m, with_geo = args
    import numpy as np
    import oracle_api as O
    w = WORKLOADS[form]
    p = degree if form == "poisson" else w["p"]
    g = O.OracleIGA(3, w["dof"])
    for i in range(3):
        g.axis_uniform(i, p, m, periodic=bool(w["periodic"][i]))
    g.setup()
    ctx = None
    if form == "poisson":
        for d in range(3):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 1.0)
    elif form == "elasticity":
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
        ctx = O.ElasticityCtx(1.0, 1.0)
    elif form == "cahnhilliard":
        ctx = O.CahnHilliardCtx(1.5, 200.0, 0.63, 1.0, 1.0 / (3.0 * m * m), 1.0)
    elif form == "nsvms":
        for s_ in range(2):
            for f in range(3):
                g.set_boundary_value(1, s_, f, 0.0)
        ctx = O.NSVMSCtx(1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2)
    if form == "nsvms" or with_geo:
        X, W = _bench_geometry(p, m, w["periodic"])
        g.set_geometry(X, W)
    A = g.create_mat()
    if w["op"] == "system":
        t = time.time()
        g.compute_system("orc_form_" + form, ctx, A=A)
        dt = time.time() - t
    else:
        n = g.global_size()
        rng = np.random.default_rng(0)
        U = (0.63 if form == "cahnhilliard" else 0.1) + 0.05 * (2 * rng.random(n) - 1)
        V = 0.01 * (2 * rng.random(n) - 1)
        res, tan = ("orc_form_ch_residual", "orc_form_ch_tangent") if form == "cahnhilliard" else ("orc_form_ns_residual", "orc_form_ns_tangent")
        t = time.time()
        g.compute_ifunction(res, ctx, 1.0e3, V, 0.0, U)
        g.compute_ijacobian(tan, ctx, 1.0e3, V, 0.0, U, A=A)
        dt = time.time() - t
    return m ** 3, dt


# demo/Poisson3D.c:3-23 (System) as a run-time form (IGXSetFormSource): what a PetIGA user's callback looks like to the library
# when it is not one of the built-in structs.  bench.py --source assembles the metric configuration through it.
USER_POISSON_SOURCE = r"""
struct UserPoisson {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 0xEu, VEC_TEST_MASK = 0x1u;
  static constexpr bool MAT_SYMMETRIC = true;
  static __device__ void mat(const PtView &, const double *Na, const double *Nb, double *T) { T[0] = Na[1] * Nb[1] + Na[2] * Nb[2] + Na[3] * Nb[3]; }
  static __device__ void vec(const PtView &p, const double *Na, double *R) { R[0] = Na[0] * p.prm[0]; }
};
"""

WORKLOADS = {
    # name: (dim, dof, p, C, default size at N GPUs, periodic, form, op, description)
    "poisson": dict(dof=1, p=3, size=256, periodic=(0, 0, 0), form="poisson", op="system",
                    metric="element stiffness assemblies/sec (3D p=3 Poisson, 256^3 elems)", ref="IGAComputeSystem demo/Poisson3D.c"),
    "elasticity": dict(dof=3, p=3, size=128, periodic=(0, 0, 0), form="elasticity", op="system",
                       metric="element stiffness assemblies/sec (3D p=3 Elasticity, 128^3 elems, 3 DOF/node)", ref="IGAComputeSystem demo/Elasticity3D.c"),
    "cahnhilliard": dict(dof=1, p=2, size=256, periodic=(0, 0, 0), form="cahnhilliard", op="tangent",
                         metric="element residual+tangent assemblies/sec (3D p=2 CahnHilliard, 256^3 elems)", ref="IGAComputeIFunction + IGAComputeIJacobian demo/CahnHilliard3D.c"),
    "nsvms": dict(dof=4, p=3, size=192, periodic=(1, 0, 1), form="nsvms", op="tangent",
                  metric="element residual+tangent assemblies/sec (3D p=3 NavierStokesVMS, 192^3 elems, 4 DOF/node, NURBS geometry)", ref="IGAComputeIFunction + IGAComputeIJacobian demo/NavierStokesVMS.c"),
}


def _state(vec_like_mat, sizes, dof, amp, base):
    """A partition-independent synthetic state: a hash of the GLOBAL natural node index, laid out in this rank's row box."""
    import numpy as np
    nrow, _, maps = vec_like_mat.layout()
    ns = sizes["node_sizes"]
    n0 = maps[0][0].astype(np.int64)[None, None, :]
    n1 = maps[1][0].astype(np.int64)[None, :, None]
    n2 = maps[2][0].astype(np.int64)[:, None, None]
    idx = (n0 + ns[0] * (n1 + ns[1] * n2)).reshape(-1)
    out = np.empty((idx.size, dof))
    for c in range(dof):
        x = np.sin((idx * dof + c) * 12.9898 + 78.233) * 43758.5453
        out[:, c] = base + amp * (2 * (x - np.floor(x)) - 1)
    return out.reshape(-1)


def build_problem(P, name, size, degree, world, rank, kernel, geometry, source=False, body_force=False):
    import numpy as np
    w = WORKLOADS[name]
    dof, p = w["dof"], (degree if name == "poisson" else w["p"])
    g = P.IGX(3, dof)
    g.set_comm(world, rank)
    for i in range(3):
        g.axis_uniform(i, p, size, periodic=bool(w["periodic"][i]))
    g.setup()
    params = ()
    if name == "poisson":
        for d in range(3):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 1.0)
    elif name == "elasticity":      # demo/Elasticity3D.c:66-71: clamped face (0,0), u_x = 1 on face (0,1); lambda = mu = 1
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
        params = (1.0, 1.0)
    elif name == "cahnhilliard":    # demo/CahnHilliard3D.c:247-251: theta, alpha, cbar, L0, lambda = tau h^2, tau
        params = (1.5, 200.0, 0.63, 1.0, 1.0 / (3.0 * size * size), 1.0)
    elif name == "nsvms":           # demo/NavierStokesVMS.c:362-385: no-slip on axis 1; nu, f, dt
        for s_ in range(2):
            for f in range(3):
                g.set_boundary_value(1, s_, f, 0.0)
        params = (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2)
    if geometry:                    # a smooth rational map (config 5's premise), the same net on every rank
        X, W = _bench_geometry(p, size, [bool(x) for x in w["periodic"]])
        g.set_geometry(X, W)
    if source:
        g.set_form_source(USER_POISSON_SOURCE, "UserPoisson", (1.0,))
    else:
        if body_force and name == "elasticity":
            g.set_form("elasticity_f", tuple(params) + (0.3, -1.25, 2.0))
        else:
            g.set_form(w["form"], params)
    g.set_kernel(kernel)
    A, b = g.create_mat(), g.create_vec()
    U = V = None
    if w["op"] == "tangent":
        U = g.create_vec().set(_state(A, g.sizes(), dof, 0.05, 0.63 if name == "cahnhilliard" else 0.1))
        V = g.create_vec().set(_state(A, g.sizes(), dof, 0.01, 0.0))
    return g, A, b, U, V, p


def launch_ranks(n):
    """Starts the n ranks of `bench.py --gpus n` as children of this (GPU-free, torch-free) process: the same command line with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rank 0's stdout relayed, the others' kept on stderr.
    Returns the worst exit code; a rank that fails takes the others down after a grace period (exact PIDs), nothing is retried."""
    import socket
    import subprocess
    with socket.socket() as sk:               # a free rendezvous port on the loop-back interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool supports only dmabuf IPC; with the legacy mode (the runtime's
    # default) hipIpcGetMemHandle fails with "invalid argument", and with it RCCL's intra-node transport between processes and any
    # sharing of device memory across ranks (the RCCL double of tests/fake_rccl included).  The image exports it already; it is set
    # here only when the caller's environment lacks it, and never overrides a value the caller chose.
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, cwd=os.getcwd(), stdout=None if r == 0 else sys.stderr))
    worst, t_fail = 0, None
    live = list(procs)
    while live:
        for pr in list(live):
            rc = pr.poll()
            if rc is not None:
                live.remove(pr)
                if rc != 0:
                    worst = worst or (rc if rc > 0 else 128 - rc)
                    t_fail = t_fail or time.time()
        if t_fail and live and time.time() - t_fail > 30.0:      # the others may sit in a rendezvous / collective for ever
            for pr in live:
                pr.kill()
            t_fail = time.time() + 1e9
        time.sleep(0.05)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--form", default="poisson", choices=sorted(WORKLOADS), help="poisson = the BASELINE metric; the others are BASELINE configs 3, 4, 5")
    ap.add_argument("--size", type=int, default=0, help="elements per axis (0: the config's own size)")
    ap.add_argument("--degree", type=int, default=3, help="poisson only")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 MFMA pencil, 3 feature")
    ap.add_argument("--geometry", action="store_true", help="mapped rational geometry (default for nsvms)")
    ap.add_argument("--source", action="store_true", help="poisson only: the form is given as run-time source (IGXSetFormSource), not as the built-in struct")
    ap.add_argument("--body-force", action="store_true", help="elasticity only: the same K with F[a][i] = N_a f_i (IGX_FORM_ELASTICITY_F); the CPU baseline keeps the demo's form")
    ap.add_argument("--two-calls", action="store_true", help="tangent workloads: IFunction and IJacobian as two calls (two passes of the element loop) instead of IGXComputeIFunctionIJacobian")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the N>1 checksum against a single-rank assembly on rank 0")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not spawn the rocprofv3 --pmc passes that measure roofline.traffic (N = 1); replay profiles/traffic.json instead")
    args = ap.parse_args()
    wl = WORKLOADS[args.form]
    # (NavierStokesVMS at 192^3 is an 8-GPU configuration -- 313 GB of matrix values; on fewer than 4 GPUs the default mesh is
    #  one GPU's share of it, 96^3, and the workload string says so)
    size = args.size or (wl["size"] if (args.form != "nsvms" or args.gpus >= 4) else 96)
    geometry = args.geometry or args.form == "nsvms"

    # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself and relays rank 0's line.  It has
    # not imported torch and never touches the GPU (a process that has initialised the GPU must not exec or be replaced).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # Libraries write to file descriptor 1 on their own (RCCL's version banner at communicator creation, gloo's connection notes):
    # everything but the result line goes to stderr, so that stdout carries exactly ONE line.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    os.environ.setdefault("IGX_CLOCK_PROBE", "1")   # read at IGXCreate: three stores by one lane per launch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or through torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    # IGX_BENCH_BACKEND=gloo is a test transport (ranks may then share one GPU); the product transport is RCCL
    backend = os.environ.get("IGX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    if backend == "nccl" and ndev < world:    # every rank sees the same count and leaves before any rendezvous: rc != 0, no hang
        sys.exit("bench.py: --gpus %d over RCCL needs %d GPUs, this box shows %d (IGX_BENCH_BACKEND=gloo is the test transport for ranks sharing one GPU)" % (world, world, ndev))
    torch.cuda.set_device(local % max(ndev, 1) if backend != "nccl" else local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # A collective that never completes (a peer died, a link is down) must end the run with an exit code, not hang it: every
        # rank arms a watchdog that leaves the process when the whole measurement has not finished in time (IGX_BENCH_TIMEOUT_S).
        import threading
        limit = float(os.environ.get("IGX_BENCH_TIMEOUT_S", "1500"))

        def _expired():
            sys.stderr.write("bench.py: rank %d of %d did not finish within %.0f s (IGX_BENCH_TIMEOUT_S): leaving with exit code 124\n" % (rank, world, limit))
            sys.stderr.flush()
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)      # where every thread of this rank stands
            os._exit(124)
        wd = threading.Timer(limit, _expired)
        wd.daemon = True
        wd.start()
        dist.init_process_group(backend, rank=rank, world_size=world)

    import petiga_amd as P
    from petiga_amd import exchange
    assert not args.source or args.form == "poisson", "--source is the metric configuration's form given as source"
    g, A, b, U, V, p = build_problem(P, args.form, size, args.degree, world, rank, args.kernel, geometry, args.source, args.body_force)
    # the library's own exchange: RCCL (or the gloo test transport).  IGX_BENCH_TRANSPORT=rccl keeps the library on its grouped
    # ncclSend / ncclRecv path while torch.distributed stays on gloo: with IGX_RCCL_LIB naming tests/fake_rccl's double, N ranks
    # that share one GPU run the product transport's schedule (tests/test_gpu_bench_ranks.py).
    want = os.environ.get("IGX_BENCH_TRANSPORT") or None
    transport = exchange.init_comm(g, transport=want) if world > 1 else None
    if want and world > 1 and transport != want:
        sys.exit("bench.py: IGX_BENCH_TRANSPORT=%s asked for, the library bound %r" % (want, transport))
    # what the transport itself reports: ncclCommCount of the library's communicator ("did RCCL see N ranks")
    try:
        comm_kind, comm_ranks = g.comm_ranks() if world > 1 else (None, None)
    except Exception:          # (a librccl without ncclCommCount: the line says so instead of the run ending here)
        comm_kind, comm_ranks = transport, None
    tangent = wl["op"] == "tangent"
    shift = 1.0e3

    def step():
        if tangent:
            if world > 1:          # DMGlobalToLocal of the state (IGAGetLocalVecArray): owner values to the ghosts
                g.refresh_ghosts(U)
                g.refresh_ghosts(V)
            if args.two_calls:
                g.compute_ifunction(shift, V, 0.0, U, b)
                g.compute_ijacobian(shift, V, 0.0, U, A)
            else:                  # the pair of a Newton step at one state in one call: one pass of the walk where a fused kernel exists
                g.compute_ifunction_ijacobian(shift, V, 0.0, U, b, A)      # (F and J are those of the two drivers: tests/test_gpu_state_pencil.py)
        else:
            g.compute_system(A, b)
        if world > 1:              # MatAssemblyBegin/End + VecAssemblyBegin/End: ghost rows to their owners
            g.reduce_ghost_rows(A, b)

    def fence():
        g.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    try:
        g.clock_probe()            # clears the sums: the figure below covers the timed steps only
    except Exception:
        pass
    g.set_timing(True)
    fence()
    t0 = time.perf_counter()
    dom_ms, dom_launches, dom_elems, dom_name, dom_flop = 0.0, 0, 0, "none", 0.0
    step_ms, step_dom_ms, step_mhz, step_ticks = [], [], [], []
    t_prev = t0
    for _ in range(args.steps):
        step()
        # HIP events recorded on the engine's own stream around the launches of the dominant kernel of this step
        d = g.dominant_kernel()
        dom_ms += d["ms"]; dom_launches += d["launches"]; dom_elems += d["elements"]
        dom_name, dom_flop = d["name"], d["executed_flop_per_element"]
        # per step: the shader clock the pencil kernel saw in THIS step (the probe's sums are read and cleared behind the step's
        # last launch), and the host clock at that point -- box variance and clock dips show up as spread, a slower kernel as a
        # shift of min / median / max together
        try:
            mhz, ticks = g.clock_probe()
        except Exception:
            mhz, ticks = None, 0
        t_now = time.perf_counter()
        step_ms.append((t_now - t_prev) * 1e3); step_dom_ms.append(d["ms"]); step_mhz.append(mhz); step_ticks.append(ticks)
        t_prev = t_now
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    total_elems = size ** 3
    value = total_elems * args.steps / dt

    # checksums over the owned rows: partition-independent up to rounding.  N > 1: rank 0 also assembles the whole mesh as a
    # single rank (it fits next to its share on a 288 GB GPU for the headline) and the sums must agree: a lost or doubled
    # ghost row cannot print a rate.
    cs = g.checksum(A, b)
    check = None
    kernel_name, proc_sizes, local_elements = g.kernel_name(), g.sizes()["proc_sizes"], int(g.element_count())
    try:                       # ms by which the upper face of axis 2 was packed before the assembly's last launch finished
        overlap_ms = g.comm_overlap_ms() if world > 1 else None
    except Exception:
        overlap_ms = None
    try:                       # phases of the last reduction (upper faces of axes 2, 1, 0) that were packed behind a face mark of the assembly
        early_phases = g.comm_early_phases() if world > 1 else None
    except Exception:
        early_phases = None
    # what the face-first decision of the pencil walks rested on, on this rank: the link rate the communicator measured at its
    # creation (or $IGX_LINK_GBS, or the constant when nothing could be probed) and the passes the last assembly made
    link = g.comm_link_rate() if world > 1 else None
    face_passes = g.face_passes() if world > 1 else None

    def assemble(gg, AA, bb, UU, VV):
        if tangent:
            if world > 1 and gg.comm_size() > 1:
                gg.refresh_ghosts(UU)
                gg.refresh_ghosts(VV)
            gg.compute_ifunction(shift, VV, 0.0, UU, bb)
            gg.compute_ijacobian(shift, VV, 0.0, UU, AA)
        else:
            gg.compute_system(AA, bb)
        if gg.comm_size() > 1:
            gg.reduce_ghost_rows(AA, bb)

    def rel_diff(cs_n, ref, nrows):
        # the signed sums may cancel: they are measured against sum|A| and sqrt(n sum b^2) (their natural bounds)
        scale = [max(float(ref[1]), 1e-300), max(float(ref[1]), 1e-300), max((nrows * float(ref[3])) ** 0.5, 1e-300), max(float(ref[3]), 1e-300)]
        return [abs(float(x) - float(y)) / sc for x, y, sc in zip(cs_n, ref, scale)]

    if world > 1:
        dev = "cuda" if backend == "nccl" else "cpu"
        tcs = torch.tensor(cs, dtype=torch.float64, device=dev)
        dist.all_reduce(tcs, op=dist.ReduceOp.SUM)
        cs = tcs.cpu().numpy()
        if not args.no_check:
            # Reference: the same mesh assembled by ONE rank on rank 0's GPU.  Its matrix must fit next to rank 0's share: the
            # headline (70 GB) does on a 288 GB GPU, NavierStokesVMS at 192^3 (313 GB of values) does not.  Then the whole flow --
            # partition, ghost refresh, assembly, ghost-row reduction -- is verified on the largest mesh of the same kind that
            # fits, with the same ranks and the same transport, and the line says so: nothing is skipped silently and nothing
            # dies in IGXCreateMat.
            bs = wl["dof"]
            def single_rank_bytes(n):
                per = [(n if wl["periodic"][i] else n + p) for i in range(3)]
                rows = per[0] * per[1] * per[2]
                blocks = rows * (2 * p + 1) ** 3
                return blocks * (bs * bs * 8 + 4) + rows * (8 + 4 * bs * 8)
            free = torch.tensor([torch.cuda.mem_get_info()[0] if rank == 0 else 0], dtype=torch.float64, device=dev)
            dist.broadcast(free, src=0)
            budget = 0.85 * float(free.item())
            csize = size
            while csize > 16 and single_rank_bytes(csize) > budget:
                csize = max(16, (csize * 3 // 4) // world * world) if csize * 3 // 4 >= world else 16
            if csize == size:
                cs_n = cs
            else:           # every rank assembles its share of the reduced mesh through the same exchange
                A = b = None            # (their memory goes back before the reduced meshes are created)
                gk, Ak, bk, Uk, Vk, _ = build_problem(P, args.form, csize, args.degree, world, rank, args.kernel, geometry, args.source, args.body_force)
                exchange.init_comm(gk, transport="rccl" if transport == "rccl" else "host")
                assemble(gk, Ak, bk, Uk, Vk)
                gk.synchronize()
                tk = torch.tensor(gk.checksum(Ak, bk), dtype=torch.float64, device=dev)
                dist.all_reduce(tk, op=dist.ReduceOp.SUM)
                cs_n = tk.cpu().numpy()
                Ak = bk = gk = None
            if rank == 0:
                g1, A1, b1, U1, V1, _ = build_problem(P, args.form, csize, args.degree, 1, 0, args.kernel, geometry, args.source, args.body_force)
                assemble(g1, A1, b1, U1, V1)
                g1.synchronize()
                ref = g1.checksum(A1, b1)
                rel = rel_diff(cs_n, ref, float(b1.n))
                check = dict(reference="single-rank assembly of the %s mesh on rank 0's GPU" % ("same" if csize == size else "%d^3 (the largest that fits next to rank 0's share; the %d^3 matrix needs %.0f GB on one GPU)" % (csize, size, single_rank_bytes(size) / 1e9)),
                             size=csize, rel_diff=rel)
                A1 = b1 = g1 = None
                assert max(rel) < 1e-9, "N-rank checksums differ from the single-rank assembly: %s vs %s" % (list(cs_n), list(ref))

    # ---- roofline of the dominant kernel, on every rank (north_star: "achieved-vs-roofline HBM and MFMA fractions reported at
    # 1/2/4/8 GPUs"); rank 0's block is the line's `roofline`, the others travel in `roofline_per_rank`
    # shader clock the chip sustained while the pencil kernel ran (s_memtime ticks per 100 MHz s_memrealtime tick of the first
    # and last workgroup of every timed launch, IGXGetClockProbe): `peak` stays the nominal 2.4 GHz figure, this says how much of the gap is clock
    good = [(m, w) for m, w in zip(step_mhz, step_ticks) if m]
    clock_mhz = (sum(m * max(w, 1) for m, w in good) / sum(max(w, 1) for m, w in good)) if good else None
    if args.form == "poisson":
        flop, cbytes = FLOP_PER_ELEM.get(args.degree, 2 * (args.degree + 1) ** 9 * 3), BYTES_PER_ELEM.get(args.degree)
    else:
        flop, cbytes = ALG[args.form]
    avg_launch_s = (dom_ms / 1e3) / max(dom_launches, 1)
    elems_per_launch = dom_elems / max(dom_launches, 1)
    achieved = flop * elems_per_launch / avg_launch_s / 1e12 if (avg_launch_s > 0 and flop) else None
    executed = dom_flop * elems_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
    # HBM bytes per launch of the dominant kernel come from rocprofv3 --pmc passes of this same command
    # (scripts/profile_round.sh), committed as profiles/traffic.json: they are NOT measured inside this run, so the
    # line names the file; null when the file does not describe this configuration (form, size, degree, ranks, round).
    traffic, traffic_source, traffic_raw = None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            for ent in tj.get("configs", [tj]):
                if ent.get("form", "poisson") == args.form and ent.get("size") == size and ent.get("degree", p) == p and ent.get("n_gpus") == world and ent.get("kernel_tag") == KERNEL_TAG and bool(ent.get("geometry", args.form == "nsvms")) == bool(geometry) \
                        and dom_name.split("<")[0].split("(")[0] in ent.get("kernel", ""):      # (the entry must describe the kernel that ran: gram_pencil ~ gram_pencil_w6<...>)
                    # PMC passes run one step: bytes per launch of the dominant kernel of THIS step shape
                    traffic = ent.get("bytes_per_launch")
                    traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, round %s; replayed, not measured in this run)" % ent.get("round", tj.get("round"))
        except Exception:
            traffic = None
    if world == 1 and rank == 0 and not args.no_live_traffic and dom_launches > 0:
        child = ["--form", args.form, "--size", str(size), "--degree", str(args.degree), "--kernel", str(args.kernel), "--steps", "1", "--warmup", "0",
                 "--no-cpu-baseline", "--no-live-traffic"] + (["--geometry"] if args.geometry else []) + (["--source"] if args.source else [])
        lt, note = live_traffic(child, dom_name)
        if lt is not None:
            traffic, traffic_source, traffic_raw = lt["corrected"], note, lt
        elif traffic_source:
            traffic_source += " [live measurement unavailable: %s]" % note
    # Dominant kernel.  `achieved` / `frac` count the flops the kernel EXECUTES on the matrix cores (the headline kernel skips the
    # 6 mirror tiles of the symmetric K_e: 10 of 16), so frac <= 1 is the fp64 MFMA-pipe fraction; the ALGORITHMIC rate
    # (SURVEY 8d / BASELINE.md 3 flop per element) is kept next to it.  For the two-assembly steps (IFunction + IJacobian) the
    # dominant kernel is the IJacobian's.
    roof = {"bound": "mfma", "achieved": executed, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": executed / FP64_PEAK_TFLOPS,
            "achieved_algorithmic": achieved, "frac_algorithmic": (achieved / FP64_PEAK_TFLOPS) if achieved else None,
            "traffic": traffic, "traffic_source": traffic_source,
            "traffic_raw_fetch": traffic_raw["fetch_raw"] if traffic_raw else None, "traffic_raw_write": traffic_raw["write_raw"] if traffic_raw else None,
            "traffic_kernel": traffic_raw["kernel"] if traffic_raw else None,
            "hbm_frac": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_launch_s > 0) else None,
            "shader_clock_mhz": clock_mhz, "nominal_clock_mhz": NOMINAL_MHZ,
            "frac_at_measured_clock": (executed / (FP64_PEAK_TFLOPS * clock_mhz / NOMINAL_MHZ)) if clock_mhz else None,
            "kernel": dom_name, "launches_per_step": dom_launches // max(args.steps, 1),
            "avg_launch_ms": avg_launch_s * 1e3, "elements_per_launch": elems_per_launch,
            "flop_per_element": flop, "executed_flop_per_element": dom_flop,
            "algorithmic_bytes_per_element": cbytes,
            "algorithmic_hbm_gbs": (cbytes * elems_per_launch / avg_launch_s / 1e9) if avg_launch_s > 0 else None}
    roofs = None
    if world > 1:
        roofs = [None] * world
        dist.all_gather_object(roofs, dict(rank=rank, local_elements=local_elements, **{k: roof[k] for k in ("achieved", "frac", "achieved_algorithmic", "frac_algorithmic", "hbm_frac", "algorithmic_hbm_gbs", "shader_clock_mhz", "avg_launch_ms", "elements_per_launch", "launches_per_step", "kernel")}))
    if rank == 0:
        line = {
            "metric": wl["metric"],
            "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_min": min(step_ms), "ms_per_step_median": sorted(step_ms)[len(step_ms) // 2], "ms_per_step_max": max(step_ms),
            "per_step": {"ms": [round(x, 3) for x in step_ms], "dominant_kernel_ms": [round(x, 3) for x in step_dom_ms],
                         "shader_clock_mhz": [round(x, 1) if x else None for x in step_mhz]},
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: p=%d C%d, %d^3 elements, dof=%d, Gauss %d^3%s%s"
                                   % (wl["ref"], p, p - 1, size, wl["dof"], p + 1, ", Dirichlet u=1 on 6 faces" if args.form == "poisson" else "",
                                      ", rational NURBS geometry map" if geometry else "") +
                                   (" (one GPU's share of the 192^3 configuration)" if (args.form == "nsvms" and size == 96 and not args.size) else "") +
                                   (" -- the form given as run-time source (IGXSetFormSource)" if args.source else ""),
                       "kernels": kernel_name, "partition": proc_sizes,
                       "pair_call": (None if not tangent else ("IGXComputeIFunction + IGXComputeIJacobian (two passes)" if args.two_calls else "IGXComputeIFunctionIJacobian (F and J of one state in one call)")),
                       "transport": transport, "rccl_ranks": comm_ranks if comm_kind == "rccl" else None, "transport_ranks": comm_ranks,
                       "exchange_started_before_assembly_end_ms": overlap_ms, "exchange_early_phases": early_phases,
                       "exchange_link_gbs": round(link[0], 2) if link else None, "exchange_link_source": link[1] if link else None,
                       "exchange_link_probe": ("%.3f ms for %d face message(s) of %s MB per direction" % (link[2], link[3], os.environ.get("IGX_LINK_PROBE_MB", "64"))) if link and link[1] == "measured" else None,
                       "face_passes": face_passes, "checksum": [float(x) for x in cs], "checksum_check": check},
            "roofline": roof,
            "device": P.device_info(),
        }
        if roofs:
            line["roofline_per_rank"] = roofs
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.form, args.degree, geometry=geometry and args.form != "nsvms")
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu"] = value / max(cb["value"], cb["cores_x_single_core"])
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line))
        sys.stdout.flush()
        os.dup2(2, 1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
