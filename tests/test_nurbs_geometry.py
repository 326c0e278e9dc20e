"""Knot-refined NURBS geometry through the IGA file format into the engine, pinned by the closed forms
of test/IGAGeometryMap.c (quarter annulus: area 3*pi/4, volume 2*area)."""
import numpy as np
import pytest
from scipy.interpolate import BSpline

import petiga_amd as P
from petiga_amd import geometry as G
import oracle_api as O
from common import compare_mats, rel_err


def _eval(degrees, knots, Pw, pts):
    """Tensor-product NURBS evaluation with scipy (independent of the library): pts [n][dim] -> X [n][nsd]."""
    out = []
    for u in pts:
        c = Pw
        for axis in range(len(degrees)):        # contract axis 0 first = last net dimension
            B = BSpline(knots[axis], np.moveaxis(c, c.ndim - 2, 0), degrees[axis])
            c = B(u[axis])
        out.append(c[:-1] / c[-1])
    return np.array(out)


@pytest.mark.parametrize("dim", [2, 3])
def test_knot_refinement_keeps_the_map(dim):
    d, U, Pw = G.quarter_annulus(dim)
    U2, Pw2 = G.refine_uniform(d, U, Pw, [5, 7, 3][:dim])
    assert [len(u) for u in U2] == [6 + 4, 6 + 6, 4 + 2][:dim]
    pts = np.random.default_rng(0).random((40, dim)) * 0.999
    a, b = _eval(d, U, Pw, pts), _eval(d, U2, Pw2, pts)
    assert np.abs(a - b).max() < 1e-14
    r = np.hypot(b[:, 0], b[:, 1])
    assert np.abs(r - (1 + pts[:, 0])).max() < 1e-14          # circles of radius 1+u (test/IGAGeometryMap.c:47-52)


def test_file_round_trip_python_and_library(tmp_path):
    d, U, Pw = G.quarter_annulus(3)
    U, Pw = G.refine_uniform(d, U, Pw, [3, 4, 2])
    f = tmp_path / "annulus.dat"
    G.write_iga(f, d, U, Pw)
    d2, U2, Pw2 = G.read_iga(f)
    assert d2 == d and all(np.array_equal(a, b) for a, b in zip(U, U2)) and np.array_equal(Pw, Pw2)
    g = P.IGX(); g.set_dof(1); g.read(f); g.setup()
    assert g.sizes()["elem_sizes"] == [3, 4, 2]
    out = tmp_path / "again.dat"; g.write(out)
    d3, U3, Pw3 = G.read_iga(out)
    assert d3 == d and np.allclose(Pw3, Pw, rtol=4e-16, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_annulus_from_file_on_device(tmp_path, dim):
    d, U, Pw = G.quarter_annulus(dim)
    U, Pw = G.refine_uniform(d, U, Pw, [6, 8, 3][:dim])
    f = tmp_path / "annulus.dat"
    G.write_iga(f, d, U, Pw)
    eng = P.IGX(); eng.set_dof(1); eng.read(f)
    for i in range(dim):
        eng.set_quadrature(i, d[i] + 3)
    eng.setup()
    for a in range(dim):
        for s in range(2):
            eng.set_boundary_form(a, s, True)
    A = np.pi * (4 - 1) / 4
    Pm = 2 * (2 - 1) + np.pi * (2 + 1) / 2
    vol, area = eng.compute_scalar("volume")
    assert abs(vol - (A if dim == 2 else 2 * A)) < 1e-6       # test/IGAGeometryMap.c:545-568, its tolerance
    assert abs(area - (Pm if dim == 2 else 2 * A + 2 * Pm)) < 1e-6
    eng.clear_boundary()
    # the same discretisation in the oracle: Poisson parity on the rational geometry
    orc = O.OracleIGA(dim, 1)
    for i in range(dim):
        orc.axis_knots(i, d[i], U[i]); orc.set_quadrature(i, d[i] + 3)
    orc.setup()
    X, W = G.split_net(Pw)
    orc.set_geometry(X, W)
    assert abs(orc.compute_scalar("orc_scalar_volume", 2)[0] - vol) < 1e-12 * vol
    for g in (orc, eng):
        g.set_boundary_value(0, 0, 0, 1.0); g.set_boundary_value(0, 1, 0, 0.0)
    eng.set_form("poisson")
    K, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, b)
    Ko, bo = orc.compute_system("orc_form_poisson")
    compare_mats(K, Ko, 1e-11)
    assert rel_err(b.get(), bo) < 1e-11
