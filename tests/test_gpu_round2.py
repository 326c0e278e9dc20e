"""GPU parity, second batch: the cases the round-1 review found unpinned.

* NavierStokesVMS at p=3 (BASELINE config 5's degree) against the oracle, also on a rational NURBS geometry;
* CahnHilliard on a NURBS geometry;
* IGAComputeFunction / IGAComputeJacobian (src/petigasnes.c:23-139) through demo/Bratu.c's callbacks;
* IGASetBoundaryLoad (src/petigaform.c:340; AddFlux / BoundaryArea, src/petigaelem.c:1118-1212) in 1-3 D, on identity and
  mapped geometries (BoundaryArea's geometry branch, src/petiga2d.F90:276-346, src/petiga3d.F90:379-464).
Every case runs under both kernel families (automatic choice = MFMA kernels where they cover the case, and the generic
point-form kernel).  fp64 tolerance 1e-12 relative (1e-11 on mapped geometries and for the nonlinear tangents), scale =
max|K| over non-Dirichlet rows (tests/common.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, rel_err, warped_geometry

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "generic"])
def kernel_family(request, monkeypatch):
    monkeypatch.setenv("IGX_KERNEL", "0" if request.param == "auto" else "1")
    return request.param


def _vec_close(a, b, tol):
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300), "vector differs: %g (scale %g)" % (np.abs(a - b).max(), np.abs(b).max())


# ---------------------------------------------------------------- config 5: NavierStokesVMS p=3 (+ NURBS)
@pytest.mark.parametrize("geo", ["none", "nurbs"])
def test_navier_stokes_vms_p3(geo, kernel_family):
    # demo/NavierStokesVMS.c:78-244,362-385: p=3, axes 0 and 2 periodic, no-slip on axis 1, dof 4
    p, N = 3, [7, 3, 8]
    orc, eng = make_pair(3, 4, p, N, periodic=[True, False, True])
    if geo == "nurbs":
        X, W = warped_geometry(orc, 3, seed=5, rational=True, amp=0.08)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):
        for side in range(2):
            for f in range(3):
                g.set_boundary_value(1, side, f, 0.0)
    nu, fx, dt = 1.472e-4, 3.37204e-3, 1e-2
    ctx = O.NSVMSCtx(nu, fx, 0.0, 0.0, dt)
    params = (nu, fx, 0.0, 0.0, dt)
    rng = np.random.default_rng(23)
    n = orc.global_size()
    U = rng.standard_normal(n) * 0.3
    V = rng.standard_normal(n) * 0.1
    shift = 2.0 / dt
    F_o = orc.compute_ifunction("orc_form_ns_residual", ctx, shift, V, 0.0, U)
    J_o = orc.compute_ijacobian("orc_form_ns_tangent", ctx, shift, V, 0.0, U)
    eng.set_form("nsvms", params)
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(shift, Vv, 0.0, Uv, F)
    if kernel_family == "auto":
        assert "vec_sumfact" in eng.kernel_name()
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    if kernel_family == "auto":
        assert "feature_assemble" in eng.kernel_name() and "mfma" in eng.kernel_name()
    _vec_close(F.get(), F_o, 1e-11)
    compare_mats(J, J_o, 1e-11)


@pytest.mark.parametrize("dim,N", [(2, 7), (3, 4)])
def test_cahn_hilliard_on_nurbs_geometry(dim, N):
    orc, eng = make_pair(dim, 1, 2, N)
    X, W = warped_geometry(orc, dim, seed=31 + dim, rational=True, amp=0.1)
    orc.set_geometry(X, W)
    eng.set_geometry(X, W)
    h = 1.0 / np.sqrt(dim * N * N)
    prm = (1.5, 200.0, 0.63, 1.0 if dim == 3 else -1.0, h * h, 1.0)
    ctx = O.CahnHilliardCtx(*prm)
    rng = np.random.default_rng(3)
    n = orc.global_size()
    U = 0.63 + 0.05 * (2 * rng.random(n) - 1)
    V = rng.standard_normal(n)
    shift = 1.0e3
    F_o = orc.compute_ifunction("orc_form_ch_residual", ctx, shift, V, 0.0, U)
    J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, shift, V, 0.0, U)
    eng.set_form("cahnhilliard", prm)
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(shift, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    _vec_close(F.get(), F_o, 1e-10)
    compare_mats(J, J_o, 1e-10)


# ---------------------------------------------------------------- IGAComputeFunction / IGAComputeJacobian
@pytest.mark.parametrize("dim,p,N,geo", [(1, 2, 9, False), (2, 2, 6, False), (2, 3, 5, True), (3, 2, 4, False), (3, 3, 3, True)])
def test_function_and_jacobian_drivers_bratu(dim, p, N, geo):
    # demo/Bratu.c:50-67: u = 0 on every face, lambda = 6.80; Function/Jacobian, then IFunction/IJacobian
    orc, eng = make_pair(dim, 1, p, N)
    if geo:
        X, W = warped_geometry(orc, dim, seed=dim + p, rational=True, amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):
        for d in range(dim):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 0.0)
    lam = C.c_double(6.80)
    rng = np.random.default_rng(dim * 7 + p)
    n = orc.global_size()
    U = rng.standard_normal(n) * 0.4
    V = rng.standard_normal(n)
    tol = 1e-11 if geo else 1e-12
    eng.set_form("bratu", (lam.value,))
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    # steady
    F_o = orc.compute_function("orc_form_bratu_function", lam, U)
    J_o = orc.compute_jacobian("orc_form_bratu_jacobian", lam, U)
    eng.compute_function(Uv, F)
    eng.compute_jacobian(Uv, J)
    eng.synchronize()
    _vec_close(F.get(), F_o, tol)
    compare_mats(J, J_o, tol)
    # fixed rows of FixFunction hold U - value (src/petigaelem.c:1449-1461): with value 0 that is U itself
    # transient
    shift = 37.5
    F_o = orc.compute_ifunction("orc_form_bratu_ifunction", lam, shift, V, 0.25, U)
    J_o = orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, shift, V, 0.25, U)
    eng.compute_ifunction(shift, Vv, 0.25, Uv, F)
    eng.compute_ijacobian(shift, Vv, 0.25, Uv, J)
    eng.synchronize()
    _vec_close(F.get(), F_o, tol)
    compare_mats(J, J_o, tol)


# ---------------------------------------------------------------- IGASetBoundaryLoad
def _loads(objs, dim, dof):
    """Loads on a few faces (two fields where dof allows), values on others; an edge shared by a loaded and a fixed face."""
    for g in objs:
        g.set_boundary_load(0, 1, 0, 2.5)
        g.set_boundary_value(0, 0, 0, 0.25)
        if dof > 1:
            g.set_boundary_load(0, 1, dof - 1, -1.5)
            g.set_boundary_value(0, 0, dof - 1, -0.5)
        if dim > 1:
            g.set_boundary_load(1, 0, 0, -0.75)
            g.set_boundary_load(1, 1, 0, 1.25)
        if dim > 2:
            g.set_boundary_load(2, 1, dof - 1, 3.0)
            g.set_boundary_value(2, 0, 0, 1.0)


@pytest.mark.parametrize("geo", ["none", "poly", "nurbs"])
@pytest.mark.parametrize("dim,dof,p,N", [(1, 1, 2, 7), (1, 2, 3, 5), (2, 1, 2, 6), (2, 2, (3, 2), (4, 5)), (3, 1, 2, 4), (3, 3, 2, 3), (3, 1, 3, 3)])
def test_boundary_loads_system(dim, dof, p, N, geo):
    orc, eng = make_pair(dim, dof, list(p) if isinstance(p, tuple) else p, list(N) if isinstance(N, tuple) else N)
    if geo != "none":
        X, W = warped_geometry(orc, dim, seed=17 + dim, rational=(geo == "nurbs"), amp=0.12)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    _loads((orc, eng), dim, dof)
    tol = 1e-12 if geo == "none" else 1e-11
    if dof == 1:
        oform, eform, octx, prm = "orc_form_poisson", "poisson", None, ()
    elif dim == 3 and dof == 3:
        oform, eform, octx, prm = "orc_form_elasticity", "elasticity", O.ElasticityCtx(1.5, 0.8), (1.5, 0.8)
    else:
        oform, eform, octx, prm = "orc_form_mass", "mass", None, ()
    A_o, b_o = orc.compute_system(oform, octx)
    eng.set_form(eform, prm)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    compare_mats(A, A_o, tol)
    _vec_close(b.get(), b_o, tol)
    # the loads do reach the vector: without them it differs
    orc.clear_boundary()
    _, b_free = orc.compute_system(oform, octx)
    assert np.abs(b_free - b_o).max() > 1e-3


@pytest.mark.parametrize("dim,geo", [(1, False), (2, False), (2, True), (3, True)])
def test_boundary_loads_function_driver(dim, geo):
    # IGAElementFixFunction subtracts the flux (src/petigaelem.c:1449-1456)
    orc, eng = make_pair(dim, 1, 2, 5)
    if geo:
        X, W = warped_geometry(orc, dim, seed=41, rational=True, amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    _loads((orc, eng), dim, 1)
    lam = C.c_double(1.0)
    rng = np.random.default_rng(5)
    U = rng.standard_normal(orc.global_size()) * 0.3
    F_o = orc.compute_function("orc_form_bratu_function", lam, U)
    eng.set_form("bratu", (1.0,))
    Uv, F = eng.create_vec().set(U), eng.create_vec()
    eng.compute_function(Uv, F)
    eng.synchronize()
    _vec_close(F.get(), F_o, 1e-11)



@pytest.mark.parametrize("geo", ["none", "poly", "nurbs"])
@pytest.mark.parametrize("p,N,stretch", [(2, (9, 4, 5), False), (3, (8, 5, 4), False), (3, (9, 4, 6), True), (2, (10, 5, 4), True)])
def test_boundary_loads_on_the_pencil_kernel(p, N, stretch, geo, kernel_family):
    """Identity geometry, dof 1, a walkable axis 0: the pencil kernel assembles K and F, a per-face kernel adds the lumped loads
    (AddFlux / BoundaryArea, src/petigaelem.c:1118-1132,1191-1212) to the rows no Dirichlet value holds."""
    knots = None
    if stretch:      # non-uniform element sizes on the axes of the loaded faces (C^{p-1} interior knots)
        knots = []
        for n in N:
            x = np.linspace(0.0, 1.0, n + 1) ** 1.5
            knots.append(np.concatenate([[0.0] * p, x, [1.0] * p]))
    orc, eng = make_pair(3, 1, p, list(N), knots=knots)
    if geo != "none":      # BoundaryArea's geometry branch: per-element face areas on the device, then the per-node sums
        X, W = warped_geometry(orc, 3, seed=13, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    _loads((orc, eng), 3, 1)
    for g in (orc, eng):
        g.set_boundary_load(0, 0, 0, 0.6)          # a load on a face that also carries a Dirichlet value: discarded there
    A_o, b_o = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    if kernel_family == "auto":      # (p = 2 on the identity geometry: the patch walk since round 6; the per-face load kernel is the same)
        assert "gram_pencil" in eng.kernel_name() or "gram_patch" in eng.kernel_name()
    tol = 1e-12 if geo == "none" else 1e-11
    compare_mats(A, A_o, tol)
    _vec_close(b.get(), b_o, tol)
    orc.clear_boundary()
    _, b_free = orc.compute_system("orc_form_poisson")
    assert np.abs(b_free - b_o).max() > 1e-3


@pytest.mark.parametrize("geo", ["none", "poly", "nurbs"])
@pytest.mark.parametrize("p,N,faces", [(2, (9, 4, 5), "all"), (3, (8, 5, 4), "all"), (3, (9, 4, 5), "some"), (2, (10, 6, 4), "some")])
def test_fix_table_on_the_pencil_kernel(p, N, faces, geo, kernel_family):
    """IGASetFixTable (src/petigaform.c:273-298; test/IGAFixTable.c): Dirichlet values per node from a vector.  The axis-0 walk of
    the pencil kernel reads them from the table in its fix-up (lifting of F through the fixed columns, F of the fixed rows)."""
    orc, eng = make_pair(3, 1, p, list(N))
    if geo != "none":
        X, W = warped_geometry(orc, 3, seed=9, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    rng = np.random.default_rng(41)
    table = rng.standard_normal(orc.global_size())
    for g in (orc, eng):
        for d in range(3):
            for s in range(2):
                if faces == "all" or (d + s) % 2 == 0:
                    g.set_boundary_value(d, s, 0, 7.0)          # the constant is ignored: the table holds the values
    orc.set_fixtable(table)
    eng.set_fixtable(eng.create_vec().set(table))
    A_o, b_o = orc.compute_system("orc_form_poisson_f")
    eng.set_form("poisson_f")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    if kernel_family == "auto":      # (p = 2 on the identity geometry: the patch walk since round 6; the per-face load kernel is the same)
        assert "gram_pencil" in eng.kernel_name() or "gram_patch" in eng.kernel_name()
    tol = 1e-12 if geo == "none" else 1e-11
    compare_mats(A, A_o, tol)
    _vec_close(b.get(), b_o, tol)
    # the table is what fixes the values: with the constant instead, F differs
    orc.set_fixtable(None)
    _, b_const = orc.compute_system("orc_form_poisson_f")
    assert np.abs(b_const - b_o).max() > 1e-3
