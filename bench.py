#!/usr/bin/env python3
"""bench.py -- element stiffness assemblies per second, 3-D p=3 Poisson on 256^3 elements
(BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

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
FP64_PEAK_TFLOPS = 78.6                        # MI355X fp64 vector = matrix peak (256 CU * 4 SIMD * 32 flop/clk * 2.4 GHz)
HBM_PEAK_GBS = 8000.0
KERNEL_TAG = "r02"                             # profiles/traffic.json must describe this round's kernel to be quoted


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
    return max(1, n)


def cpu_baseline(degree, seconds_target=15.0):
    """Times the CPU oracle (port of the reference loop) on this box's host cores on a bounded sample of the same
    workload.  One worker per PHYSICAL core; every worker assembles its own box of m^3 elements into its own local
    matrix -- what a rank of `mpiexec -n cores` does with its ghosted box (the reference's MatSetValuesLocal works on the
    rank-local rows) -- so no worker pays for a global pattern.  Reported: the measured all-core rate, the single-core
    rate and cores x single-core (the no-loss bound); the speed-up quoted next to it uses the larger of the two."""
    import multiprocessing as mp
    cores = physical_cores()
    rate1 = 900.0 if degree == 3 else 9000.0          # rough single-core rate, only to size the sample
    m = int(round((rate1 * seconds_target) ** (1.0 / 3.0)))
    m = max(8, min(m, 40))
    # one core alone first (also warms the page cache / builds nothing: the .so is prebuilt)
    e1, t1 = _cpu_worker((degree, m))
    t0 = time.time()
    if cores > 1:
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(degree, m)] * cores)
    else:
        res = [(e1, t1)]
    wall = max(r[1] for r in res)
    elems = sum(r[0] for r in res)
    single = e1 / t1
    return dict(value=elems / wall, unit="elements/s", cores=cores, kind="port", single_core_value=single,
                cores_x_single_core=cores * single,
                logical_cpus=os.cpu_count(),
                sample="3-D p=%d Poisson System (Dirichlet on 6 faces), %d^3 elements per core into a core-local matrix, %d physical cores at once "
                       "(oracle/igaoracle.c, the reference's loop: order-%d tabulation, scalar callback, search-insert); slowest core %.1f s, "
                       "one core alone %.1f s, pool wall %.1f s" % (degree, m, cores, degree, wall, t1, time.time() - t0))


def _cpu_worker(args):
    degree, m = args
    import oracle_api as O
    g = O.OracleIGA(3, 1)
    for i in range(3):
        g.axis_uniform(i, degree, m)
    g.setup()
    for d in range(3):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    A = g.create_mat()
    t = time.time()
    g.compute_system("orc_form_poisson", A=A)
    dt = time.time() - t
    return m ** 3, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=256, help="elements per axis (metric config: 256)")
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # IGX_BENCH_BACKEND=gloo is a test transport (ranks may then share one GPU); the product transport is RCCL
    backend = os.environ.get("IGX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    import petiga_amd as P
    from petiga_amd import exchange
    g = P.IGX(3, 1)
    g.set_comm(world, rank)
    for i in range(3):
        g.axis_uniform(i, args.degree, args.size)
    g.setup()
    for d in range(3):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    g.set_form("poisson")
    g.set_kernel(args.kernel)
    A, b = g.create_mat(), g.create_vec()
    ex = exchange.GhostExchange(g, A, b) if world > 1 else None

    def step():
        g.compute_system(A, b)
        if ex is not None:
            ex.reduce()

    def fence():
        g.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    g.set_timing(True)
    fence()
    t0 = time.perf_counter()
    dom_ms, dom_launches, dom_elems, dom_name, dom_flop = 0.0, 0, 0, "none", 0.0
    for _ in range(args.steps):
        step()
        # HIP events recorded on the engine's own stream around the launches of the dominant kernel of this step
        d = g.dominant_kernel()
        dom_ms += d["ms"]; dom_launches += d["launches"]; dom_elems += d["elements"]
        dom_name, dom_flop = d["name"], d["executed_flop_per_element"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    total_elems = args.size ** 3
    value = total_elems * args.steps / dt

    if rank == 0:
        flop = FLOP_PER_ELEM.get(args.degree, 2 * (args.degree + 1) ** 9 * 3)
        avg_launch_s = (dom_ms / 1e3) / max(dom_launches, 1)
        elems_per_launch = dom_elems / max(dom_launches, 1)
        achieved = flop * elems_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        executed = dom_flop * elems_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel come from rocprofv3 --pmc passes of this same command
        # (scripts/profile_round.sh), committed as profiles/traffic.json: they are NOT measured inside this run, so the
        # line names the file and the commit that last touched it; null when the file does not describe this configuration.
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("size") == args.size and tj.get("degree") == args.degree and tj.get("n_gpus") == world and tj.get("kernel_tag") == KERNEL_TAG:
                    traffic = tj.get("bytes_per_launch")
                    traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, round %s; replayed, not measured in this run)" % tj.get("round")
            except Exception:
                traffic = None
        line = {
            "metric": "element stiffness assemblies/sec (3D p=3 Poisson, 256^3 elems)",
            "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "IGAComputeSystem demo/Poisson3D.c: p=%d C%d, %d^3 elements, dof=1, Dirichlet u=1 on 6 faces, Gauss %d^3"
                                   % (args.degree, args.degree - 1, args.size, args.degree + 1),
                       "kernels": g.kernel_name(), "partition": g.sizes()["proc_sizes"]},
            # Dominant kernel.  `achieved` / `frac` count the flops the kernel EXECUTES on the matrix cores (it skips the 6
            # mirror tiles of the symmetric K_e: 10 of 16), so frac <= 1 is the fp64 MFMA-pipe fraction; the ALGORITHMIC rate
            # (2*nen^2*nqp*dim flop per element, SURVEY 8d / BASELINE.md 3) is kept next to it.
            "roofline": {"bound": "mfma", "achieved": executed, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": executed / FP64_PEAK_TFLOPS,
                         "achieved_algorithmic": achieved, "frac_algorithmic": achieved / FP64_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "hbm_frac": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_launch_s > 0) else None,
                         "kernel": dom_name, "launches_per_step": dom_launches // max(args.steps, 1),
                         "avg_launch_ms": avg_launch_s * 1e3, "elements_per_launch": elems_per_launch,
                         "flop_per_element": flop, "executed_flop_per_element": dom_flop,
                         "algorithmic_bytes_per_element": BYTES_PER_ELEM.get(args.degree)},
            "device": P.device_info(),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.degree)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu"] = value / max(cb["value"], cb["cores_x_single_core"])
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
