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


def cpu_baseline(degree, seconds_target=20.0):
    """Times the CPU oracle (port of the reference loop) on this box's host cores on a bounded sample
    of the same workload: same discretisation, a smaller cube, every core assembling its own block
    of elements (emulates mpiexec -n cores; src/petigapart.c partition)."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 64))
    while cores > 1 and cores not in (2, 4, 8, 16, 32, 64):
        cores -= 1
    rate1 = 550.0 if degree == 3 else 6000.0          # rough single-core guess, only to size the sample
    n = int(round((rate1 * cores * seconds_target) ** (1.0 / 3.0)))
    n = max(8, min(n, 64))
    t0 = time.time()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(degree, n, cores, r) for r in range(cores)])
    wall = max(r[1] for r in res)
    elems = sum(r[0] for r in res)
    # one core alone (the reference is single-threaded per rank; SURVEY 8d asks for both figures)
    n1 = 16 if degree == 3 else 32
    e1, t1 = _cpu_worker((degree, n1, 1, 0))
    return dict(value=elems / wall, unit="elements/s", cores=cores, kind="port", single_core_value=e1 / t1,
                single_core_sample="%d^3 elements on one core, %.1f s" % (n1, t1),
                sample="3-D p=%d Poisson, %d^3 elements, %d ranks (one per core), oracle/igaoracle.c; slowest rank %.1f s, pool wall %.1f s"
                       % (degree, n, cores, wall, time.time() - t0))


def _cpu_worker(args):
    degree, n, size, rank = args
    import oracle_api as O
    g = O.OracleIGA(3, 1)
    for i in range(3):
        g.axis_uniform(i, degree, n)
    g.set_partition(size, rank)
    g.setup()
    for d in range(3):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    A = g.create_mat()
    t = time.time()
    g.compute_system("orc_form_poisson", A=A)
    dt = time.time() - t
    w = g.ranges()["elem_width"]
    return w[0] * w[1] * w[2], dt


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
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")      # HBM bytes per launch of the dominant kernel from rocprofv3 --pmc passes
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("size") == args.size and tj.get("degree") == args.degree and tj.get("n_gpus") == world:
                    traffic = tj.get("bytes_per_launch")
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
            # dominant kernel; achieved = ALGORITHMIC flops (2*nen^2*nqp*dim per element) / measured launch time.
            # The kernel exploits the symmetry of K_e (10 of 16 MFMA tiles), so it EXECUTES executed_flop_per_element < flop_per_element
            # and `frac` may exceed 1; mfma_busy_frac is the executed-flop fraction of the fp64 MFMA peak.
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": dom_name, "launches_per_step": dom_launches // max(args.steps, 1),
                         "avg_launch_ms": avg_launch_s * 1e3, "elements_per_launch": elems_per_launch,
                         "flop_per_element": flop, "executed_flop_per_element": dom_flop,
                         "mfma_busy_frac": executed / FP64_PEAK_TFLOPS,
                         "algorithmic_bytes_per_element": BYTES_PER_ELEM.get(args.degree)},
            "device": P.device_info(),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.degree)
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
