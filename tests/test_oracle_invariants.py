"""Physics invariants of the ORACLE's restated callbacks that the reference itself never runs (Elasticity3D,
CahnHilliard, NavierStokesVMS, Bratu: built by demo/makefile, no numeric pin).  These are independent checks of the
restatement (last row): they follow from partition of unity (sum_a N_a = 1, sum_a grad N_a = 0,
sum_a lap N_a = 0), from B-splines reproducing linear fields through Greville abscissae, and from the tangent being the
derivative of the residual -- not from the code under test.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import greville, make_pair, warped_geometry


def _int_N(orc):
    """b_a = int N_a (test/IGACreate.c's vector)"""
    m = O.OracleIGA(orc.dim, 1)
    for i in range(orc.dim):
        ax = orc.axis(i)
        m.axis_knots(i, ax["p"], ax["U"], periodic=ax["periodic"])
    m.setup()
    if getattr(orc, "_keep", None) is not None:
        m.set_geometry(*orc._keep)
    return m.compute_system("orc_form_mass")[1]


@pytest.mark.parametrize("dim,geo", [(2, False), (3, False), (2, True)])
def test_cahn_hilliard_mass_conservation_and_tangent(dim, geo):
    # demo/CahnHilliard3D.c:55-179.  Periodic box (or natural boundaries): sum_a R_a = int c_t, and the column sums of the
    # tangent are shift * int N_b: every other term carries grad N_a or lap N_a, which sum to zero over a.
    N = 6 if dim == 2 else 4
    orc, _ = make_pair(dim, 1, 2, N, periodic=not geo, engine=False)
    if geo:
        X, W = warped_geometry(orc, dim, seed=2, rational=True, amp=0.1)
        orc.set_geometry(X, W)
    h = 1.0 / np.sqrt(dim * N * N)
    ctx = O.CahnHilliardCtx(1.5, 200.0, 0.63, 1.0 if dim == 3 else -1.0, h * h, 1.0)
    rng = np.random.default_rng(1)
    n = orc.global_size()
    U = 0.63 + 0.05 * (2 * rng.random(n) - 1)
    V = rng.standard_normal(n)
    shift = 250.0
    F = orc.compute_ifunction("orc_form_ch_residual", ctx, shift, V, 0.0, U)
    J = orc.compute_ijacobian("orc_form_ch_tangent", ctx, shift, V, 0.0, U).scipy()
    bN = _int_N(orc)
    scale = np.abs(F).sum()
    assert abs(F.sum() - V @ bN) < 1e-12 * scale
    colsum = np.asarray(J.sum(axis=0)).ravel()
    assert np.abs(colsum - shift * bN).max() < 1e-10 * np.abs(J.data).max()
    # Newton consistency: J d = d/de F(U + e d, V + shift e d) (central difference)
    d = rng.standard_normal(n)
    e = 1e-6
    Fp = orc.compute_ifunction("orc_form_ch_residual", ctx, shift, V + shift * e * d, 0.0, U + e * d)
    Fm = orc.compute_ifunction("orc_form_ch_residual", ctx, shift, V - shift * e * d, 0.0, U - e * d)
    fd = (Fp - Fm) / (2 * e)
    assert np.abs(J @ d - fd).max() < 2e-6 * np.abs(fd).max()


def test_elasticity_rigid_body_modes():
    # demo/Elasticity3D.c:13-46 with mu = 1 (the :37 quirk is invisible): translations and infinitesimal rotations are in
    # the null space of the un-constrained matrix, on a mapped NURBS geometry as well.
    for geo in (False, True):
        orc, _ = make_pair(3, 3, 2, 3, engine=False)
        if geo:
            X, W = warped_geometry(orc, 3, seed=4, rational=False, amp=0.1)
            orc.set_geometry(X, W)
        ctx = O.ElasticityCtx(2.5, 1.0)
        A, _ = orc.compute_system("orc_form_elasticity", ctx)
        K = A.scipy()
        if geo:
            P = X                                   # polynomial geometry: the control points reproduce x itself
        else:
            g = [greville(orc.axis(i)["U"], 2) for i in range(3)]
            m = np.meshgrid(*g[::-1], indexing="ij")[::-1]
            P = np.stack(m, axis=-1).reshape(-1, 3)
        scale = np.abs(K.data).max()
        for t in np.eye(3):
            assert np.abs(K @ np.tile(t, len(P))).max() < 1e-12 * scale
        for w in np.eye(3):
            assert np.abs(K @ np.cross(w, P).reshape(-1)).max() < 1e-12 * scale * np.abs(P).max()
        assert abs(K - K.T).max() < 1e-13 * scale


def test_navier_stokes_vms_invariants():
    # demo/NavierStokesVMS.c:78-164 in a fully periodic box: a uniform flow with no forcing is a steady solution (R = 0),
    # and for any state the pressure rows sum to int div u = 0 (all other terms of Rp carry grad N_a).
    orc, _ = make_pair(3, 4, 2, [5, 6, 5], periodic=True, engine=False)
    n = orc.global_size()
    ctx = O.NSVMSCtx(1e-3, 0.0, 0.0, 0.0, 1e-2)
    U = np.tile([0.3, -0.2, 0.5, 1.7], n // 4)
    F = orc.compute_ifunction("orc_form_ns_residual", ctx, 200.0, np.zeros(n), 0.0, U)
    assert np.abs(F).max() < 1e-13
    rng = np.random.default_rng(9)
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n) * 0.1
    ctx = O.NSVMSCtx(1e-3, 0.02, -0.01, 0.03, 1e-2)
    F = orc.compute_ifunction("orc_form_ns_residual", ctx, 200.0, V, 0.0, U)
    assert abs(F[3::4].sum()) < 1e-11 * np.abs(F[3::4]).sum()
    # momentum rows: sum_a R_a = int (u_t - f + (u + u') . grad u) -- with grad u = 0 (uniform velocity, any pressure
    # field p): int (u_t - f) exactly
    U2 = U.copy(); U2[0::4], U2[1::4], U2[2::4] = 0.3, -0.2, 0.5
    F = orc.compute_ifunction("orc_form_ns_residual", ctx, 200.0, V, 0.0, U2)
    bN = np.repeat(_int_N(orc), 1)
    for c, f in enumerate((0.02, -0.01, 0.03)):
        assert abs(F[c::4].sum() - (V[c::4] @ bN - f * bN.sum())) < 1e-11 * np.abs(F[c::4]).sum()


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_bratu_jacobian_is_the_derivative_of_the_function(dim):
    # demo/BratuFJ.F90:23-107
    orc, _ = make_pair(dim, 1, 2, 5, engine=False)
    for d in range(dim):
        for s in range(2):
            orc.set_boundary_value(d, s, 0, 0.0)
    lam = C.c_double(6.8)
    rng = np.random.default_rng(dim)
    n = orc.global_size()
    U, d = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    J = orc.compute_jacobian("orc_form_bratu_jacobian", lam, U).scipy()
    e = 1e-6
    fd = (orc.compute_function("orc_form_bratu_function", lam, U + e * d) - orc.compute_function("orc_form_bratu_function", lam, U - e * d)) / (2 * e)
    assert np.abs(J @ d - fd).max() < 1e-7 * np.abs(fd).max()
    assert abs(J - J.T).max() < 1e-13 * np.abs(J.data).max()


@pytest.mark.parametrize("dim,rational", [(2, False), (2, True), (3, False), (3, True)])
def test_boundary_load_total_is_the_face_area(dim, rational):
    # AddFlux (src/petigaelem.c:1191-1212) lumps load * BoundaryArea onto the face's basis functions: over the whole face the
    # vector gains load * (area of the mapped face), which IGAComputeScalar's boundary pass measures independently
    # (test/IGAGeometryMap.c:383-450).  With a mapped geometry the lumping is exact only in the sum.
    orc, _ = make_pair(dim, 1, 2, 4, engine=False)
    X, W = warped_geometry(orc, dim, seed=8, rational=rational, amp=0.12)
    orc.set_geometry(X, W)
    _, b0 = orc.compute_system("orc_form_poisson")
    for axis, side in ((0, 1), (dim - 1, 0)):
        orc.clear_boundary()
        orc.set_boundary_load(axis, side, 0, 2.0)
        _, b = orc.compute_system("orc_form_poisson")
        orc.clear_boundary()
        orc.set_boundary_form(axis, side, True)
        area = orc.compute_scalar("orc_scalar_volume", 2, full=True)[1]
        # BoundaryArea hands every face function of an element the same share A_e / nen_face (not int N_a): summed over the
        # nen_face functions of the element that is its face area
        assert abs((b - b0).sum() - 2.0 * area) < 1e-10 * area
