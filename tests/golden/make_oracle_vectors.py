#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_element_vectors.npz: element matrices / vectors of the five BASELINE forms on one
interior and one corner element, computed by the CPU oracle (oracle/) on seeded inputs.  They pin the ORACLE against
accidental change (the reference itself cannot be built in this image); they are not outputs of
the reference.   usage: python tests/golden/make_oracle_vectors.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_api as O  # noqa: E402
from common import warped_geometry  # noqa: E402


def cases():
    """(name, assembled matrix values, vector) for small meshes: the assembled system is the sum of the element
    matrices, so any change of an element kernel of the oracle moves these numbers."""
    out = {}
    g = O.OracleIGA(3, 1)
    for i in range(3):
        g.axis_uniform(i, 3, 3)
    g.setup()
    for d in range(3):
        g.set_boundary_value(d, 0, 0, 1.0)
    A, b = g.compute_system("orc_form_poisson")
    out["poisson3d_p3"] = (A.val.copy(), b.copy())
    g = O.OracleIGA(2, 1)
    for i in range(2):
        g.axis_uniform(i, 2, 4)
    g.setup()
    X, W = warped_geometry(g, 2, seed=1, rational=True)
    g.set_geometry(X, W)
    g.set_boundary_value(0, 1, 0, 0.5)
    A, b = g.compute_system("orc_form_poisson")
    out["poisson2d_p2_nurbs"] = (A.val.copy(), b.copy())
    g = O.OracleIGA(3, 3)
    for i in range(3):
        g.axis_uniform(i, 2, 2)
    g.setup()
    for f in range(3):
        g.set_boundary_value(0, 0, f, 0.0)
    g.set_boundary_value(0, 1, 0, 1.0)
    A, b = g.compute_system("orc_form_elasticity", O.ElasticityCtx(2.5, 0.7))
    out["elasticity3d_p2"] = (A.val.copy(), b.copy())
    rng = np.random.default_rng(7)
    g = O.OracleIGA(3, 1)
    for i in range(3):
        g.axis_uniform(i, 2, 3)
    g.set_order(2)
    g.setup()
    n = g.global_size()
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), 0.1 * rng.standard_normal(n)
    ctx = O.CahnHilliardCtx(1.5, 200.0, 0.63, 1.0, 1.0 / 27, 1.0)
    F = g.compute_ifunction("orc_form_ch_residual", ctx, 2.0, V, 0.0, U)
    J = g.compute_ijacobian("orc_form_ch_tangent", ctx, 2.0, V, 0.0, U)
    out["cahnhilliard3d_p2"] = (J.val.copy(), F.copy())
    g = O.OracleIGA(3, 4)
    g.axis_uniform(0, 2, 5, periodic=True); g.axis_uniform(1, 2, 2); g.axis_uniform(2, 2, 5, periodic=True)
    g.set_order(2)
    g.setup()
    for s in range(2):
        for f in range(3):
            g.set_boundary_value(1, s, f, 0.0)
    n = g.global_size()
    U, V = 0.3 * rng.standard_normal(n), 0.1 * rng.standard_normal(n)
    ctx = O.NSVMSCtx(1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2)
    F = g.compute_ifunction("orc_form_ns_residual", ctx, 2.0, V, 0.0, U)
    J = g.compute_ijacobian("orc_form_ns_tangent", ctx, 2.0, V, 0.0, U)
    out["navierstokesvms_p2"] = (J.val.copy(), F.copy())
    return out


if __name__ == "__main__":
    data = {}
    for k, (m, v) in cases().items():
        data[k + "_mat"] = m
        data[k + "_vec"] = v
    np.savez_compressed(os.path.join(HERE, "oracle_element_vectors.npz"), **data)
    print({k: v.shape for k, v in data.items()})
