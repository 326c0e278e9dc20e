"""Periodic axes with fewer than 2p+1 basis functions (p+1 <= nnp < 2p+1), held by one rank.  A row's stencil wraps onto itself
there; the reference's ghost-index pattern (ColumnIndices, src/petigamat.c:243-267) maps the duplicates to the same global
column through the LGMap and MatSetValuesLocal adds them.  The oracle restates that; the engine keeps the distinct columns of a
row, sends every duplicate slot to the same position and gives each element of the axis a colour of its own."""
import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    # dim, dof, p, N, periodic, form, params
    (1, 1, 2, (3,), (1,), "poisson", ()),
    (1, 1, 3, (5,), (1,), "mass", ()),
    (2, 1, 2, (3, 6), (1, 0), "poisson", ()),
    (2, 1, 2, (4, 4), (1, 1), "mass", ()),
    (2, 1, 3, (4, 7), (1, 0), "poisson", ()),
    (2, 2, 3, (5, 6), (1, 1), "mass", ()),
    (3, 1, 2, (3, 4, 5), (1, 1, 0), "poisson", ()),
    (3, 1, 3, (8, 4, 5), (0, 1, 1), "poisson", ()),          # the pencil kernel walks axis 0 next to two tiny periodic axes
    (3, 1, 3, (9, 6, 4), (0, 0, 1), "poisson", ()),
    (3, 3, 2, (3, 5, 4), (1, 0, 0), "elasticity", (1.3, 0.7)),
    (3, 3, 3, (4, 5, 4), (1, 0, 1), "elasticity", (1.3, 0.7)),
]


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-d%d-p%d-N%s" % (c[5], c[0], c[2], "x".join(map(str, c[3]))))
def test_system_on_tiny_periodic_axes(case, kernel):
    dim, dof, p, N, periodic, form, params = case
    periodic = [bool(x) for x in periodic]
    orc, eng = make_pair(dim, dof, p, list(N), periodic=periodic)
    eng.set_kernel(kernel)
    for g in (orc, eng):
        for d in range(dim):
            if not periodic[d]:
                for f in range(dof):
                    g.set_boundary_value(d, 0, f, 0.25 * (f + 1))
    eng.set_form(form, params)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    A_o, b_o = orc.compute_system("orc_form_" + form, O.ElasticityCtx(*params) if form == "elasticity" else None)
    # every row of a tiny periodic axis couples with all its nnp functions, once
    n0 = [orc.axis(i)["nnp"] for i in range(dim)]
    for d in range(dim):
        if periodic[d]:
            assert n0[d] < 2 * p + 1
    compare_mats(A, A_o, 1e-12)
    assert rel_err(b.get(), b_o) <= 1e-12 or np.abs(b.get() - b_o).max() <= 1e-13


@pytest.mark.parametrize("kernel", [0, 1])
def test_cahn_hilliard_on_a_fully_periodic_4x4x4_mesh(kernel):
    """demo/CahnHilliard3D.c is periodic on every axis; p = 2 with 4 elements per axis is below 2p+1 = 5 functions."""
    params = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)
    orc, eng = make_pair(3, 1, 2, [4, 4, 4], periodic=[True] * 3)
    eng.set_kernel(kernel)
    eng.set_form("cahnhilliard", params)
    n = orc.global_size()
    rng = np.random.default_rng(11)
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    Uv, Vv = eng.create_vec().set(U), eng.create_vec().set(V)
    A, F = eng.create_mat(), eng.create_vec()
    eng.compute_ifunction(7.5, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(7.5, Vv, 0.0, Uv, A)
    eng.synchronize()
    ctx = O.CahnHilliardCtx(*params)
    F_o = orc.compute_ifunction("orc_form_ch_residual", ctx, 7.5, V, 0.0, U)
    A_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, 7.5, V, 0.0, U)
    compare_mats(A, A_o, 1e-11)
    assert rel_err(F.get(), F_o) <= 1e-11


def test_fewer_than_p_plus_1_functions_is_refused():
    import petiga_amd as P
    g = P.IGX(2, 1)
    g.axis_uniform(0, 3, 3, periodic=True)     # 3 functions < p + 1: an element's own functions would alias each other
    g.axis_uniform(1, 3, 5)
    with pytest.raises(P.IGXError):
        g.setup()
        g.create_mat()
