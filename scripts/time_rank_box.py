#!/usr/bin/env python3
"""What the face-first passes cost: the assembly of ONE rank's box of the metric configuration at 8 ranks (rank 0 of [2,2,2]: 128^3
elements of the 256^3 mesh, upper neighbours on all three axes) on one GPU, with a do-nothing transport attached so that the
assembly makes its face passes; IGX_OVERLAP=0 (one pass), 2 (the upper half of axis 2 first), 1 (three faces), unset (the walk's own choice: cost against the size of the faces)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import petiga_amd as P  # noqa: E402

world, size = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = P.IGX(3, 1)
g.set_comm(world, 0)
for i in range(3):
    g.axis_uniform(i, 3, size)
g.setup()
for d in range(3):
    for s in range(2):
        g.set_boundary_value(d, s, 0, 1.0)
g.set_form("poisson")
g.comm_init_transport(lambda send, recv: None)
A, b = g.create_mat(), g.create_vec()
g.set_timing(True)
ts = []
for _ in range(6):
    g.synchronize()
    t = time.perf_counter()
    g.compute_system(A, b)
    g.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
d = g.dominant_kernel()
print("IGX_OVERLAP=%s ranks %d box %s: assembly %.2f ms (min of 5), %d launches" % (os.environ.get("IGX_OVERLAP", "unset"), world, g.sizes()["elem_width"], min(ts[1:]), d["launches"]))
