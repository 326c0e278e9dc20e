"""p = 4 in 3-D (nen = 125 > 64): the matrix-core kernel with 8x8 tiles of 16x16 (feature_mfma.hpp, TA = 8: wave w owns tile
column w and all eight tile rows) against the oracle.  Before round 2 such discretisations ran on the point-form kernel alone
(0.13 M elements/s at 32^3)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, rel_err, warped_geometry

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("p,N,geo,form,dof", [((4, 4, 4), (3, 2, 3), "none", "poisson", 1), ((4, 4, 4), (2, 3, 2), "nurbs", "poisson", 1),
                                             ((4, 4, 3), (3, 3, 2), "poly", "poisson", 1), ((5, 4, 4), (2, 2, 3), "none", "poisson", 1),
                                             ((4, 4, 4), (2, 2, 3), "none", "mass", 2), ((4, 4, 4), (3, 2, 2), "nurbs", "mass", 3),
                                             ((4, 4, 4), (2, 3, 2), "none", "mass", 1),
                                             ((5, 5, 5), (2, 2, 2), "none", "poisson", 1), ((5, 5, 4), (2, 2, 2), "nurbs", "poisson", 1),
                                             ((5, 5, 5), (2, 1, 2), "none", "mass", 2), ((7, 4, 4), (1, 2, 2), "poly", "poisson", 1)])
def test_system_at_degree_four(p, N, geo, form, dof, kernel):
    orc, eng = make_pair(3, dof, list(p), list(N))
    eng.set_kernel(kernel)
    if geo != "none":
        X, W = warped_geometry(orc, 3, seed=3, rational=(geo == "nurbs"), amp=0.08)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):
        g.set_boundary_value(0, 0, 0, 0.5)
        g.set_boundary_value(2, 1, dof - 1, -1.0)
        g.set_boundary_load(1, 1, 0, 2.0)
    A_o, b_o = orc.compute_system("orc_form_" + form)
    eng.set_form(form)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    nen = int(np.prod([q + 1 for q in p]))
    if kernel == 0:      # 8x8 tiles up to nen = 128, 16 tile rows x two column panels up to 256 (one accumulator set: scalar forms, mass)
        assert ("tiles=8x8" in eng.kernel_name()) == (64 < nen <= 128), eng.kernel_name()
        assert ("tiles=16x16" in eng.kernel_name()) == (128 < nen <= 256), eng.kernel_name()
    tol = 1e-12 if geo == "none" else 2e-11
    compare_mats(A, A_o, tol)
    assert np.abs(b.get() - b_o).max() <= tol * max(np.abs(b_o).max(), 1.0)
    eng.compute_matrix(A)
    eng.compute_vector(b)
    eng.synchronize()
    orc.clear_boundary()
    A_o2, b_o2 = orc.compute_system("orc_form_" + form)
    compare_mats(A, A_o2, tol)
    assert np.abs(b.get() - b_o2).max() <= tol * max(np.abs(b_o2).max(), 1.0)


@pytest.mark.parametrize("kernel", [0, 1])
def test_bratu_at_degree_four(kernel):
    """a nonlinear scalar form through the Function / Jacobian and IFunction / IJacobian drivers"""
    orc, eng = make_pair(3, 1, 4, [2, 3, 2])
    eng.set_kernel(kernel)
    for g in (orc, eng):
        for d in range(3):
            g.set_boundary_value(d, 0, 0, 0.0)
    lam = C.c_double(3.5)
    rng = np.random.default_rng(8)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    eng.set_form("bratu", (3.5,))
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(4.0, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(4.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    if kernel == 0:
        assert "tiles=8x8" in eng.kernel_name()
    F_o = orc.compute_ifunction("orc_form_bratu_ifunction", lam, 4.0, V, 0.0, U)
    assert np.abs(F.get() - F_o).max() <= 1e-12 * np.abs(F_o).max()
    compare_mats(J, orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, 4.0, V, 0.0, U), 1e-12)


def test_cahn_hilliard_at_degree_four():
    """second derivatives of N at nen = 125 (PHI_MASK keeps 7 of 13 features)"""
    params = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)
    orc, eng = make_pair(3, 1, 4, [3, 2, 2])
    eng.set_form("cahnhilliard", params)
    n = orc.global_size()
    rng = np.random.default_rng(12)
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    Uv, Vv, A, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat(), eng.create_vec()
    eng.compute_ifunction(7.5, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(7.5, Vv, 0.0, Uv, A)
    eng.synchronize()
    assert "tiles=8x8" in eng.kernel_name()
    ctx = O.CahnHilliardCtx(*params)
    compare_mats(A, orc.compute_ijacobian("orc_form_ch_tangent", ctx, 7.5, V, 0.0, U), 1e-11)
    assert rel_err(F.get(), orc.compute_ifunction("orc_form_ch_residual", ctx, 7.5, V, 0.0, U)) <= 1e-11


def test_vector_only_drivers_at_degree_four_for_a_four_field_form():
    """NS-VMS at p = 4: the Tangent has too many accumulator sets for 8x8 tiles (point-form kernel), the Residual needs none"""
    orc, eng = make_pair(3, 4, 4, [5, 2, 2], periodic=[True, False, False])
    for g in (orc, eng):
        for side in range(2):
            for f in range(3):
                g.set_boundary_value(1, side, f, 0.0)
    nu, fx, dt = 1.472e-4, 3.37204e-3, 1e-2
    ctx = O.NSVMSCtx(nu, fx, 0.0, 0.0, dt)
    rng = np.random.default_rng(2)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n) * 0.1
    eng.set_form("nsvms", (nu, fx, 0.0, 0.0, dt))
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(200.0, Vv, 0.0, Uv, F)
    eng.synchronize()
    assert "feature_assemble(vector only" in eng.kernel_name()
    F_o = orc.compute_ifunction("orc_form_ns_residual", ctx, 200.0, V, 0.0, U)
    assert np.abs(F.get() - F_o).max() <= 1e-11 * np.abs(F_o).max()
    eng.compute_ijacobian(200.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "generic_assemble" in eng.kernel_name()
    compare_mats(J, orc.compute_ijacobian("orc_form_ns_tangent", ctx, 200.0, V, 0.0, U), 1e-11)


@pytest.mark.parametrize("seed", range(30))
def test_random_high_degree_discretisation(seed):
    """seeded sweep: degrees 3..5 per axis (nen up to 256), 1..3 elements per axis, Poisson / mass (1-2 fields), identity /
    polynomial / NURBS geometry, random Dirichlet values and loads"""
    rng = np.random.default_rng(9000 + seed)
    p = [int(rng.integers(3, 6)) for _ in range(3)]
    if np.prod([q + 1 for q in p]) > 256:
        p[int(rng.integers(0, 3))] = 3
    N = [int(rng.integers(1, 4)) for _ in range(3)]
    form = str(rng.choice(["poisson", "mass"]))
    dof = 1 if form == "poisson" else int(rng.integers(1, 3))
    geo = str(rng.choice(["none", "poly", "nurbs"]))
    orc, eng = make_pair(3, dof, p, N)
    if geo != "none":
        X, W = warped_geometry(orc, 3, seed=seed, rational=(geo == "nurbs"), amp=0.06)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for d in range(3):
        for s in range(2):
            r, f, v = rng.random(), int(rng.integers(0, dof)), float(rng.normal())
            for g in (orc, eng):
                if r < 0.4:
                    g.set_boundary_value(d, s, f, v)
                elif r < 0.6:
                    g.set_boundary_load(d, s, 0, v)
    A_o, b_o = orc.compute_system("orc_form_" + form)
    eng.set_form(form)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "feature_assemble" in eng.kernel_name()
    tol = 1e-12 if geo == "none" else 5e-11
    compare_mats(A, A_o, tol)
    assert np.abs(b.get() - b_o).max() <= tol * max(np.abs(b_o).max(), 1.0)
