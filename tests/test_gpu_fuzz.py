"""Seeded sweep over discretisations (dimension, mixed degrees, element counts down to one, reduced continuity, periodic
axes, quadrature sizes, Dirichlet sets, NURBS geometry): device result vs oracle for the automatic kernel choice and
for the generic kernel.  Ragged / minimal inputs the reference's tests touch (test/IGACreate.c loops over
dim, degree, continuity, periodicity) are the point here."""
import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, rel_err, warped_geometry

pytestmark = pytest.mark.gpu


def _random_case(rng):
    dim = int(rng.integers(1, 4))
    p = [int(rng.integers(1, 4 if dim == 3 else 5)) for _ in range(dim)]
    periodic = [bool(rng.random() < 0.25) for _ in range(dim)]
    N, C = [], []
    for i in range(dim):
        c = int(rng.integers(0, p[i])) if rng.random() < 0.4 else p[i] - 1
        nmin = 1
        if periodic[i]:
            c = p[i] - 1                                  # periodic + reduced continuity is not exercised by the reference either
            nmin = 2 * p[i] + 1                           # nnp >= 2p+1 (engine restriction, IGX_ERR_SUP otherwise)
        N.append(int(rng.integers(nmin, nmin + (5 if dim == 3 else 9))))
        C.append(c)
    nqp = [None if rng.random() < 0.6 else int(rng.integers(max(1, p[i]), p[i] + 3)) for i in range(dim)]
    form = rng.choice(["poisson", "mass", "elasticity"] if dim == 3 else ["poisson", "mass"])
    dof = 1 if form == "poisson" else (3 if form == "elasticity" else int(rng.integers(1, 4)))
    geo = rng.choice(["none", "poly", "nurbs"]) if not any(periodic) else "none"
    bcs = []
    for d in range(dim):
        for s in range(2):
            if not periodic[d] and rng.random() < 0.5:
                bcs.append((d, s, int(rng.integers(0, dof)), float(rng.normal())))
    return dict(dim=dim, p=p, N=N, C=C, periodic=periodic, nqp=nqp, form=form, dof=dof, geo=geo, bcs=bcs)


@pytest.mark.parametrize("seed", range(150))
@pytest.mark.parametrize("kernel", [0, 1])
def test_random_discretisation(seed, kernel, monkeypatch):
    monkeypatch.setenv("IGX_KERNEL", str(kernel))
    c = _random_case(np.random.default_rng(1000 + seed))
    orc, eng = make_pair(c["dim"], c["dof"], c["p"], c["N"], C=c["C"], periodic=c["periodic"], nqp=c["nqp"])
    if c["geo"] != "none":
        X, W = warped_geometry(orc, c["dim"], seed=seed, rational=(c["geo"] == "nurbs"), amp=0.08)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    for (d, s, f, v) in c["bcs"]:
        orc.set_boundary_value(d, s, f, v); eng.set_boundary_value(d, s, f, v)
    if c["form"] == "poisson":
        Ao, bo = orc.compute_system("orc_form_poisson"); eng.set_form("poisson")
    elif c["form"] == "mass":
        Ao, bo = orc.compute_system("orc_form_mass"); eng.set_form("mass")
    else:
        Ao, bo = orc.compute_system("orc_form_elasticity", O.ElasticityCtx(1.7, 0.6)); eng.set_form("elasticity", (1.7, 0.6))
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    tol = 1e-12 if c["geo"] == "none" else 2e-11
    compare_mats(A, Ao, tol)
    assert np.abs(b.get() - bo).max() <= tol * max(np.abs(bo).max(), 1.0), c
