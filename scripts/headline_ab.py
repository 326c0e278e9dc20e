#!/usr/bin/env python3
"""A/B of the headline kernel inside one gpurun lease: the metric configuration (3-D p=3 Poisson, 256^3, System driver) assembled
`--steps` times through the package under --pkg (this tree, or a worktree of an earlier round) and, optionally, another build of
the library (IGX_LIB).  Prints one JSON line: per-step ms of the dominant kernel, the shader clock of each step (IGX_CLOCK_PROBE)
and their product, shader cycles per launch -- the figure that separates a slower kernel from a slower box."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--pkg", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap.add_argument("--label", default="")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--degree", type=int, default=3)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--warmup", type=int, default=2)
args = ap.parse_args()
sys.path.insert(0, os.path.abspath(args.pkg))
os.environ["IGX_CLOCK_PROBE"] = "1"
import torch  # noqa: F401,E402
import petiga_amd as P  # noqa: E402

g = P.IGX(3, 1)
for i in range(3):
    g.axis_uniform(i, args.degree, args.size)
g.setup()
for d in range(3):
    for s in range(2):
        g.set_boundary_value(d, s, 0, 1.0)
g.set_form("poisson")
A, b = g.create_mat(), g.create_vec()
for _ in range(args.warmup):
    g.compute_system(A, b)
g.synchronize()
g.clock_probe()
g.set_timing(True)
ms, dom, mhz, launches = [], [], [], 0
for _ in range(args.steps):
    t = time.perf_counter()
    g.compute_system(A, b)
    d = g.dominant_kernel()
    c = g.clock_probe()[0]
    ms.append((time.perf_counter() - t) * 1e3); dom.append(d["ms"]); mhz.append(c); launches = d["launches"]
cyc = [m * 1e-3 * c * 1e6 / max(launches, 1) for m, c in zip(dom, mhz)]
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps(dict(label=args.label or args.pkg, lib=os.environ.get("IGX_LIB"), kernel=d["name"], launches=launches,
                      step_ms_median=round(med(ms), 3), dom_ms_median=round(med(dom), 3), mhz_median=round(med(mhz), 1),
                      mcycles_per_launch_median=round(med(cyc) / 1e6, 4), mcycles_per_launch_min=round(min(cyc) / 1e6, 4),
                      melem_per_s_median=round(args.size ** 3 / med(ms) / 1e3, 2),
                      step_ms=[round(x, 2) for x in ms], mhz=[round(x) for x in mhz])))
