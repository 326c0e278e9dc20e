#!/usr/bin/env python3
"""Round 6: launch-bound small meshes (VERDICT r05 missing item 8).  REPS assemblies of IGAComputeSystem queued back to back, one
synchronize at the end: wall time per assembly, to set against the sum of the kernels' own durations (rocprofv3 --kernel-trace --stats of
this script).  IGX_GRAPH=1: the assembly's launches replayed as a HIP graph."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import petiga_amd as P

REPS = int(os.environ.get("REPS", "40"))
SIZES = os.environ.get("SIZES")
DEG = int(os.environ.get("DEG", "3"))
for p, n in ([(DEG, int(x)) for x in SIZES.split()] if SIZES else ((3, 16), (3, 24), (3, 32), (3, 48), (2, 32), (2, 48))):
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, n)
    g.setup()
    for d in range(3):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    g.set_form("poisson")
    A, b = g.create_mat(), g.create_vec()
    mat = os.environ.get("IGX_SMALL_OP", "") == "matrix"
    run = (lambda: g.compute_matrix(A)) if mat else (lambda: g.compute_system(A, b))
    for _ in range(3):
        run()
    g.synchronize()
    t = time.perf_counter()
    for _ in range(REPS):
        run()
    g.synchronize()
    dt = (time.perf_counter() - t) / REPS
    print("p=%d %d^3: %.1f us per assembly, %.2f M el/s, %d launches  %s" % (p, n, dt * 1e6, n ** 3 / dt / 1e6, g.last_timing()[2], g.kernel_name()[:60]), flush=True)
