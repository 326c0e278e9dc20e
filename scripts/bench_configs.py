#!/usr/bin/env python3
"""Times the secondary BASELINE configs (one assembly each) on the current kernels; not the headline bench."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import petiga_amd as P


def run(name, dim, dof, p, N, form, params=(), periodic=None, op="system", bc=None, C=-1, steps=2, geo=False):
    g = P.IGX(dim, dof)
    for i in range(dim):
        g.axis_uniform(i, p, N[i], C, periodic=bool(periodic[i]) if periodic else False)
    g.setup()
    if bc:
        bc(g)
    g.set_form(form, params)
    if geo:
        from petiga_amd.geometry import greville
        import numpy as _np
        gv = [greville(_np.concatenate([[0.0] * (p + 1), _np.arange(1, N[i]) / N[i], [1.0] * (p + 1)]), p) for i in range(dim)]
        mesh = _np.meshgrid(*gv[::-1], indexing="ij")[::-1]
        X = _np.stack([m.copy() for m in mesh], axis=-1)
        X[..., 0] += 0.05 * _np.sin(2 * _np.pi * mesh[1]); X[..., 1] += 0.05 * _np.sin(2 * _np.pi * mesh[dim - 1])
        W = 1.0 + 0.1 * _np.cos(2 * _np.pi * mesh[0])
        g.set_geometry(X.reshape(-1, dim), None if geo == "poly" else W.reshape(-1))
    A = g.create_mat() if op in ("system", "ijacobian", "matrix") else None
    b = g.create_vec()
    U = V = None
    if op in ("ifunction", "ijacobian"):
        rng = np.random.default_rng(0)
        U = g.create_vec().set(0.63 + 0.05 * (2 * rng.random(b.n) - 1))
        V = g.create_vec().set(np.zeros(b.n))
    times = []
    for _ in range(steps):
        g.synchronize()
        t = time.perf_counter()
        if op == "system":
            g.compute_system(A, b)
        elif op == "matrix":
            g.compute_matrix(A)
        elif op == "ijacobian":
            g.compute_ijacobian(1e3, V, 0.0, U, A)
        else:
            g.compute_ifunction(1e3, V, 0.0, U, b)
        g.synchronize()
        times.append(time.perf_counter() - t)
    nel = int(np.prod(N))
    if os.environ.get("BENCH_COMPACT"):
        print("%-44s %8.2f ms %7.2f M el/s  %s" % (name[:44], min(times) * 1e3, nel / min(times) / 1e6, g.kernel_name()))
    else:
        print(json.dumps(dict(config=name, kernel=g.kernel_name(), elements=nel, ms=min(times) * 1e3, elements_per_s=nel / min(times))))


def dirichlet_all(g, dim, v=1.0):
    for d in range(dim):
        for s in range(2):
            g.set_boundary_value(d, s, 0, v)


import os
which = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c4r"]
if "c1" in which:
    run("Poisson2D p=2 64^2", 2, 1, 2, (64, 64), "poisson", bc=lambda g: dirichlet_all(g, 2))
if "c2" in which:
    run("Poisson3D p=2 128^3", 3, 1, 2, (128,) * 3, "poisson", bc=lambda g: dirichlet_all(g, 3))
if "c3" in which:
    def bc3(g):
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
    run("Elasticity3D p=3 64^3 (config 3 is 128^3)", 3, 3, 3, (64,) * 3, "elasticity", (1.0, 1.0), bc=bc3)
if "c4" in which:
    h2 = 1.0 / (3 * 128 * 128)
    run("CahnHilliard3D p=2 128^3 tangent (config 4 is 256^3)", 3, 1, 2, (128,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ijacobian")
if "c4g" in which:
    h2 = 1.0 / (3 * 128 * 128)
    run("CahnHilliard3D p=2 128^3 tangent on a NURBS geometry", 3, 1, 2, (128,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ijacobian", geo=True)
    run("CahnHilliard3D p=2 128^3 tangent on a polynomial geometry", 3, 1, 2, (128,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ijacobian", geo="poly")
    run("CahnHilliard3D p=2 128^3 residual on a NURBS geometry", 3, 1, 2, (128,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ifunction", geo=True)
if "c4r" in which:
    h2 = 1.0 / (3 * 128 * 128)
    run("CahnHilliard3D p=2 128^3 residual", 3, 1, 2, (128,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ifunction")
if "c5" in which:
    def bc5(g):
        for s in range(2):
            for f in range(3):
                g.set_boundary_value(1, s, f, 0.0)
    run("NavierStokesVMS p=3 32^3 tangent (config 5 is 192^3 on 8 GPUs)", 3, 4, 3, (32,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), periodic=(1, 0, 1), op="ijacobian", bc=bc5)
if "c5g" in which:
    def bc5g(g):
        for s in range(2):
            for f in range(3):
                g.set_boundary_value(1, s, f, 0.0)
    run("NavierStokesVMS p=3 48^3 tangent on a NURBS geometry", 3, 4, 3, (48,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), op="ijacobian", bc=bc5g, geo=True)
    run("NavierStokesVMS p=3 48^3 residual on a NURBS geometry", 3, 4, 3, (48,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), op="ifunction", bc=bc5g, geo=True)
    run("NavierStokesVMS p=3 48^3 tangent, no geometry", 3, 4, 3, (48,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), op="ijacobian", bc=bc5g)
if "c6" in which:
    run("Poisson3D p=3 64^3 on a NURBS geometry", 3, 1, 3, (64,) * 3, "poisson", bc=lambda g: dirichlet_all(g, 3), geo=True)
if "c6b" in which:
    run("Poisson3D p=3 128^3 on a NURBS geometry", 3, 1, 3, (128,) * 3, "poisson", bc=lambda g: dirichlet_all(g, 3), geo=True)
if "c6m" in which:
    run("Poisson3D p=3 128^3 on a NURBS geometry, Matrix driver", 3, 1, 3, (128,) * 3, "poisson", op="matrix", geo=True)
if "c6p" in which:
    run("Poisson3D p=3 128^3 on a polynomial geometry", 3, 1, 3, (128,) * 3, "poisson", bc=lambda g: dirichlet_all(g, 3), geo="poly")
if "metric" in which:
    run("Poisson3D p=3 256^3 (the metric config)", 3, 1, 3, (256,) * 3, "poisson", bc=lambda g: dirichlet_all(g, 3))
if "c7" in which:
    run("Poisson3D p=2 96^3 on a NURBS geometry", 3, 1, 2, (96,) * 3, "poisson", bc=lambda g: dirichlet_all(g, 3), geo=True)
if "full3" in which:
    run("Elasticity3D p=3 128^3 (config 3, full size)", 3, 3, 3, (128,) * 3, "elasticity", (1.0, 1.0), bc=bc3 if "c3" in which else (lambda g: [g.set_boundary_value(0, 0, f, 0.0) for f in range(3)] and g.set_boundary_value(0, 1, 0, 1.0)))
if "full4" in which:
    h2 = 1.0 / (3 * 256 * 256)
    run("CahnHilliard3D p=2 256^3 tangent (config 4, full size, one GPU)", 3, 1, 2, (256,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ijacobian")
    run("CahnHilliard3D p=2 256^3 residual (config 4, full size, one GPU)", 3, 1, 2, (256,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ifunction")
if "full5" in which:
    def bc5f(g):
        for s in range(2):
            for f in range(3):
                g.set_boundary_value(1, s, f, 0.0)
    run("NavierStokesVMS p=3 96^3 tangent (config 5 is 192^3 on 8 GPUs: this is one GPU's share)", 3, 4, 3, (96,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), periodic=(1, 0, 1), op="ijacobian", bc=bc5f)
if "c5r" in which:
    def bc5r(g):
        for s in range(2):
            for f in range(3):
                g.set_boundary_value(1, s, f, 0.0)
    run("NavierStokesVMS p=3 48^3 residual", 3, 4, 3, (48,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), periodic=(1, 0, 1), op="ifunction", bc=bc5r)
if "e2" in which:
    run("Elasticity3D p=2 96^3", 3, 3, 2, (96,) * 3, "elasticity", (1.0, 1.0), bc=lambda g: [g.set_boundary_value(0, 0, f, 0.0) for f in range(3)])
if "n2" in which:
    run("NavierStokesVMS p=2 48^3 tangent", 3, 4, 2, (48,) * 3, "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), periodic=(1, 0, 1), op="ijacobian")
if "m4" in which:
    run("mass dof=4 p=2 64^3", 3, 4, 2, (64,) * 3, "mass")
if "c4p3" in which:
    h2 = 1.0 / (3 * 96 * 96)
    run("CahnHilliard3D p=3 96^3 tangent", 3, 1, 3, (96,) * 3, "cahnhilliard", (1.5, 200.0, 0.63, 1.0, h2, 1.0), op="ijacobian")
    run("Bratu p=3 96^3 Jacobian", 3, 1, 3, (96,) * 3, "bratu", (3.5,), op="ijacobian")
