"""GPU parity of vec_sumfact (petiga_amd/csrc/vec_sumfact.hpp): the vector-only drivers (IGAComputeVector src/petigaksp.c:127-170,
IGAComputeFunction src/petigasnes.c:44-92, IGAComputeIFunction src/petigats.c:55-110) in three dimensions at nen, nqp <= 4 per
axis, with the state and the geometry interpolated by sum factorisation across the lanes of one wavefront and the test-function
sums taken the same way backwards.  Engine vs oracle on identical inputs to 1e-11 of max|F| (1e-12 for the scalar forms),
Dirichlet rows overwritten as IGAFormFixFunction does; the same call twice returns the same bits."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import make_pair, warped_geometry

pytestmark = pytest.mark.gpu

NU, FX, DT = 1.472e-4, 3.37204e-3, 1e-2
CH = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)


def _close(a, b, tol):
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * np.abs(b).max(), "vector differs: %g (scale %g)" % (np.abs(a - b).max(), np.abs(b).max())


def _geometry(orc, eng, geo, seed):
    if geo:
        X, W = warped_geometry(orc, 3, seed=seed, rational=(geo == "nurbs"), amp=0.08)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)


@pytest.mark.parametrize("N,periodic,geo,walls", [
    ((7, 3, 8), (True, False, True), None, True),          # config 5's topology (demo/NavierStokesVMS.c:362-385)
    ((7, 3, 8), (True, False, True), "nurbs", True),
    ((5, 4, 6), (False, False, False), "poly", False),
    ((4, 5, 4), (False, True, False), "nurbs", True),
    ((1, 1, 1), (False, False, False), "nurbs", False),    # one element: 64 of the 64 lanes are points, nothing to colour
    ((9, 2, 3), (True, False, False), None, False),
])
def test_ns_vms_residual(N, periodic, geo, walls):
    orc, eng = make_pair(3, 4, 3, list(N), periodic=list(periodic))
    _geometry(orc, eng, geo, 5 + sum(N))
    if walls:
        for g in (orc, eng):
            for d in range(3):
                if not periodic[d]:
                    for side in range(2):
                        for f in range(3):
                            g.set_boundary_value(d, side, f, 0.1 * f - 0.05 * side)
    ctx = O.NSVMSCtx(NU, FX, -0.4 * FX, 0.25 * FX, DT)
    rng = np.random.default_rng(23)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n) * 0.1
    shift = 2.0 / DT
    eng.set_form("nsvms", (NU, FX, -0.4 * FX, 0.25 * FX, DT))
    Uv, Vv, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec()
    eng.compute_ifunction(shift, Vv, 0.0, Uv, F)
    eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name(), eng.kernel_name()
    F1 = F.get().copy()
    _close(F1, orc.compute_ifunction("orc_form_ns_residual", ctx, shift, V, 0.0, U), 1e-11)
    F.set(np.full(n, np.nan))
    eng.compute_ifunction(shift, Vv, 0.0, Uv, F)
    assert np.array_equal(F.get(), F1)                      # every entry written, same bits


@pytest.mark.parametrize("p,N,C_,periodic,geo", [
    (2, (6, 5, 7), -1, (False, False, False), None),        # demo/CahnHilliard3D.c: p = 2, C1
    (2, (6, 5, 7), -1, (True, True, True), None),           # ... on the periodic box of the demo
    (3, (4, 5, 3), 1, (True, False, True), None),           # p = 3 at reduced continuity
    (2, (4, 4, 4), -1, (False, False, False), "nurbs"),     # second derivatives of the test functions on a NURBS map (round 4: this kernel too)
    (2, (6, 5, 4), -1, (False, False, False), "poly"),      # ... on a polynomial map
    (3, (4, 3, 5), -1, (False, False, False), "nurbs"),     # ... at p = 3 (one element per wavefront)
])
def test_cahn_hilliard_residual(p, N, C_, periodic, geo):
    orc, eng = make_pair(3, 1, p, list(N), C=C_, periodic=list(periodic))
    _geometry(orc, eng, geo, 41)
    ctx = O.CahnHilliardCtx(*CH)
    rng = np.random.default_rng(3)
    n = orc.global_size()
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    eng.set_form("cahnhilliard", CH)
    Uv, Vv, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec()
    eng.compute_ifunction(250.0, Vv, 0.0, Uv, F)
    eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name(), eng.kernel_name()
    _close(F.get(), orc.compute_ifunction("orc_form_ch_residual", ctx, 250.0, V, 0.0, U), 1e-11 if geo is None else 1e-10)


@pytest.mark.parametrize("p,N,nqp,geo", [
    ((1, 1, 1), (5, 6, 4), None, None),
    ((1, 2, 3), (5, 4, 3), None, "nurbs"),                  # a different degree on each axis
    ((3, 3, 3), (4, 3, 5), [3, 4, 2], "poly"),              # fewer Gauss points than p + 1
    ((2, 2, 2), (6, 6, 6), None, "nurbs"),
    ((3, 2, 1), (3, 7, 9), [4, 4, 4], None),                # more points than p + 1 on the low-degree axes
])
def test_bratu_function_and_ifunction(p, N, nqp, geo):
    orc, eng = make_pair(3, 1, list(p), list(N), nqp=nqp)
    _geometry(orc, eng, geo, 17)
    for g in (orc, eng):
        for d in range(3):
            g.set_boundary_value(d, 0, 0, 0.25 * d)         # FixFunction: F_a = U_a - value on these rows
    lam = C.c_double(3.5)
    rng = np.random.default_rng(8)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    eng.set_form("bratu", (3.5,))
    Uv, Vv, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec()
    eng.compute_function(Uv, F)
    eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name(), eng.kernel_name()
    _close(F.get(), orc.compute_function("orc_form_bratu_function", lam, U), 1e-12)
    eng.compute_ifunction(4.0, Vv, 0.0, Uv, F)
    eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name()
    _close(F.get(), orc.compute_ifunction("orc_form_bratu_ifunction", lam, 4.0, V, 0.0, U), 1e-12)


@pytest.mark.parametrize("form,oform,dof,geo", [("poisson", "orc_form_poisson", 1, "nurbs"), ("poisson_f", "orc_form_poisson_f", 1, None),
                                                ("l2proj_x2", "orc_form_l2proj_x2", 1, "poly")])
def test_vector_driver_of_linear_forms(form, oform, dof, geo):
    """IGAComputeVector (no boundary fix-up, src/petigaksp.c:127-170) on knot vectors with repeated interior knots"""
    knots = [np.r_[[0] * 3, 0.2, 0.5, 0.5, 0.7, [1] * 3], np.r_[[0] * 3, 0.4, 0.6, [1] * 3], np.r_[[0] * 3, 0.1, 0.3, 0.3, 0.9, [1] * 3]]
    orc, eng = make_pair(3, dof, 2, [0, 0, 0], knots=knots)
    _geometry(orc, eng, geo, 2)
    orc.clear_boundary()
    _, b_o = orc.compute_system(oform)
    eng.set_form(form)
    b = eng.create_vec()
    eng.compute_vector(b)
    eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name(), eng.kernel_name()
    _close(b.get(), b_o, 1e-12)


def test_switch_and_kernel_choice(monkeypatch):
    """IGX_VEC_SUMFACT=0 and IGXSetKernel(1..3) keep the earlier kernels; the results agree to rounding"""
    import petiga_amd as P
    res = {}
    for tag, env, kernel in (("sumfact", None, 0), ("off", "0", 0), ("generic", None, 1)):
        if env is None:
            monkeypatch.delenv("IGX_VEC_SUMFACT", raising=False)
        else:
            monkeypatch.setenv("IGX_VEC_SUMFACT", env)
        g = P.IGX(3, 1)
        for i in range(3):
            g.axis_uniform(i, 2, 5 + i)
        g.setup()
        g.set_kernel(kernel)
        g.set_form("cahnhilliard", CH)
        n = int(np.prod(g.sizes()["node_sizes"]))
        rng = np.random.default_rng(1)
        U, V, F = g.create_vec().set(0.63 + 0.05 * rng.standard_normal(n)), g.create_vec().set(rng.standard_normal(n)), g.create_vec()
        g.compute_ifunction(10.0, V, 0.0, U, F)
        g.synchronize()
        res[tag] = (F.get().copy(), g.kernel_name())
    assert "vec_sumfact" in res["sumfact"][1] and "vec_sumfact" not in res["off"][1] and "vec_sumfact" not in res["generic"][1]
    for tag in ("off", "generic"):
        _close(res[tag][0], res["sumfact"][0], 1e-12)


@pytest.mark.parametrize("geo", [None, "nurbs"])
def test_fix_table_values_in_function_and_jacobian(geo):
    """IGASetFixTable (src/petigaform.c:273-298): the value of a fixed dof comes from a row-indexed table; IGAElementFixValues puts
    it into the state, FixFunction subtracts it (src/petigaelem.c:1334-1358, :1449-1461).  The sum-factorised kernel reads the
    table, and so does the state walk when it gathers the coefficients of U."""
    from common import compare_mats
    orc, eng = make_pair(3, 1, 2, [9, 4, 5])
    _geometry(orc, eng, geo, 3)
    for g in (orc, eng):
        for d in range(3):
            g.set_boundary_value(d, 0, 0, 0.0); g.set_boundary_value(d, 1, 0, 0.0)      # the faces; the values come from the table
    rng = np.random.default_rng(4)
    n = orc.global_size()
    table, U = rng.standard_normal(n) * 0.2, rng.standard_normal(n) * 0.3
    orc.set_fixtable(table)
    eng.set_fixtable(eng.create_vec().set(table))
    lam = C.c_double(3.5)
    eng.set_form("bratu", (3.5,))
    Uv, F, J = eng.create_vec().set(U), eng.create_vec(), eng.create_mat()
    eng.compute_function(Uv, F); eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name()
    _close(F.get(), orc.compute_function("orc_form_bratu_function", lam, U), 1e-12)
    eng.compute_jacobian(Uv, J); eng.synchronize()
    assert "state_pencil" in eng.kernel_name(), eng.kernel_name()      # (round 4: on a mapped geometry too, at p = 2)
    compare_mats(J, orc.compute_jacobian("orc_form_bratu_jacobian", lam, U), 1e-12 if geo is None else 1e-11)
    orc.set_fixtable(None)
