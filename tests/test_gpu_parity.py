"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical inputs.
Tolerance: fp64, 1e-12 relative to max|K_e entry| (summation order differs: colour order vs
lexicographic element order, FMA contraction) -- BASELINE.json north_star "stated fp64 tolerance".
Indexing, sparsity pattern and colouring are compared exactly."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse.linalg as sla

import oracle_api as O
from common import compare_mats, make_pair, rel_err, warped_geometry

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(autouse=True, params=["auto", "generic"])
def kernel_family(request, monkeypatch):
    """Every parity case runs twice: with the automatic kernel choice (MFMA kernels wherever they cover the case)
    and with the generic point-form kernel preset (IGX_KERNEL is read when an IGX is created)."""
    monkeypatch.setenv("IGX_KERNEL", "0" if request.param == "auto" else "1")
    return request.param


def _poison(mat):
    """Overwrite the device values with NaN bit patterns: a first-touch store that misses an entry shows up."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _, _, val = mat.device_ptrs()
    assert hip.hipMemset(val, 0xFF, mat.nblocks * mat.bs * mat.bs * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


def dirichlet_all(objs, dim, value=1.0, field=0):
    for g in objs:
        for d in range(dim):
            for s in range(2):
                g.set_boundary_value(d, s, field, value)


def system_pair(orc, eng, oform, eform, octx=None, params=()):
    A_o, b_o = orc.compute_system(oform, octx)
    eng.set_form(eform, params)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    return A, b, A_o, b_o


@pytest.mark.parametrize("dim,p,N", [(1, 3, 7), (2, 2, 8), (2, 2, 64), (2, 3, 5), (3, 1, 4), (3, 2, 6), (3, 3, 5), (3, 4, 3), (3, (3, 2, 1), (4, 5, 6))])
def test_poisson_system(dim, p, N, kernel_family):
    orc, eng = make_pair(dim, 1, p, N)
    dirichlet_all((orc, eng), dim)
    nen = np.prod([q + 1 for q in (p if isinstance(p, tuple) else [p] * dim)])
    eng.set_kernel(3 if (kernel_family == "auto" and dim >= 2 and nen <= 64) else 1)   # 3: feature-GEMM MFMA kernel
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    compare_mats(A, A_o, TOL)
    assert rel_err(b.get(), b_o) < TOL


def test_tutorial_sizes_on_device():
    # docs/manual/TUTORIAL.rst:113-115: 3-D p=2 16^3 -> 5832 rows / 592704 non-zeros
    orc, eng = make_pair(3, 1, 2, 16)
    A = eng.create_mat()
    assert (A.nbrows, A.nblocks) == (5832, 592704)
    rp, ci, _ = A.host()
    Ao = orc.create_mat()
    assert np.array_equal(rp, Ao.rowptr) and np.array_equal(ci, Ao.colidx)


def test_poisson_16cube_p3_full(kernel_family):
    orc, eng = make_pair(3, 1, 3, 16)
    dirichlet_all((orc, eng), 3)
    eng.set_kernel(3 if kernel_family == "auto" else 1)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    compare_mats(A, A_o, TOL)
    assert rel_err(b.get(), b_o) < TOL
    # solve on the host from the device matrix: u == 1 + Poisson bump, min at the boundary
    x = sla.spsolve(A.to_scipy_global().tocsc(), b.get())
    assert abs(x.min() - 1.0) < 1e-10 and x.max() > 1.0


@pytest.mark.parametrize("dim,dof,periodic,p", [(1, 4, (0,), (2,)), (2, 2, (0, 0), (2, 2)), (3, 1, (0, 0, 0), (2, 2, 2)),
                                               (2, 3, (0, 1), (2, 3)), (2, 3, (1, 0), (2, 3)), (2, 3, (1, 1), (2, 3)), (3, 2, (1, 0, 1), (2, 2, 3))])
def test_mass_system_periodic(dim, dof, periodic, p):
    # test/IGACreate.c (test/makefile:25-33), incl. periodic axes with mixed degrees
    N = 9 if dim < 3 else 7
    orc, eng = make_pair(dim, dof, list(p), N, periodic=[bool(x) for x in periodic])
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_mass", "mass")
    compare_mats(A, A_o, TOL)
    assert rel_err(b.get(), b_o) < TOL
    x = sla.spsolve(A.to_scipy_global().tocsc(), b.get())
    assert x.max() - x.min() < 1e-9


@pytest.mark.parametrize("dim,N", [(1, 16), (2, 16), (3, 4)])
def test_fixtable_flow(dim, N):
    # test/IGAFixTable.c end to end on the device path, -check_error 1e-6
    orc, eng = make_pair(dim, 1, 2, N)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_l2proj_x2", "l2proj_x2")
    compare_mats(A, A_o, TOL)
    x = sla.spsolve(A.to_scipy_global().tocsc(), b.get())
    dirichlet_all((orc, eng), dim, 0.0)
    orc.set_fixtable(x)
    xv = eng.create_vec().set(x)
    eng.set_fixtable(xv)
    A2, b2, A2_o, b2_o = system_pair(orc, eng, "orc_form_poisson_f", "poisson_f")
    compare_mats(A2, A2_o, TOL)
    assert rel_err(b2.get(), b2_o) < TOL
    u = sla.spsolve(A2.to_scipy_global().tocsc(), b2.get())
    orc.set_fixtable(None)
    err = np.sqrt(orc.compute_scalar("orc_scalar_x2err", 1, U=u)[0])
    assert err < 1e-6


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_errnorm_projection(dim):
    # test/IGAErrNorm.c:134-146: the device-assembled L2 projection reproduces 1, Sx, Sx^2, Px
    N = 8 if dim < 3 else 4
    orc, eng = make_pair(dim, 4, 2, N, nqp=3, order=2)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_errnorm", "errnorm")
    compare_mats(A, A_o, TOL)
    assert rel_err(b.get(), b_o) < TOL
    x = sla.spsolve(A.to_scipy_global().tocsc(), b.get())
    for order in (0, 1, 2):
        S = np.sqrt(orc.compute_scalar("orc_scalar_errnorm", 4, U=x, ctx=C.c_int(order)))
        assert np.all(S < np.sqrt(np.finfo(float).eps))


@pytest.mark.parametrize("p,N", [(2, 4), (3, 3)])
def test_elasticity_system(p, N):
    # demo/Elasticity3D.c: lambda = mu = 1, clamped face (0,0), u_x = 1 on face (0,1)
    orc, eng = make_pair(3, 3, p, N)
    for g in (orc, eng):
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
    for lam, mu in ((1.0, 1.0), (2.5, 0.7)):       # the second pair exposes the reference's mu*mu quirk
        ctx = O.ElasticityCtx(lam, mu)
        A, b, A_o, b_o = system_pair(orc, eng, "orc_form_elasticity", "elasticity", ctx, (lam, mu))
        compare_mats(A, A_o, TOL)
        assert np.abs(b.get() - b_o).max() <= TOL * max(np.abs(b_o).max(), 1.0)


@pytest.mark.parametrize("geo", ["poly", "nurbs"])
@pytest.mark.parametrize("dim,p,N", [(2, 2, 5), (2, 3, 4), (3, 2, 4), (3, 3, 3)])
def test_poisson_on_mapped_geometry(dim, p, N, geo, kernel_family):
    orc, eng = make_pair(dim, 1, p, N)
    X, W = warped_geometry(orc, dim, seed=dim * 10 + p, rational=(geo == "nurbs"))
    orc.set_geometry(X, W)
    eng.set_geometry(X, W)
    dirichlet_all((orc, eng), dim, 0.5)
    eng.set_kernel(3 if kernel_family == "auto" else 1)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    compare_mats(A, A_o, 1e-11)
    assert rel_err(b.get(), b_o) < 1e-11


def test_nonuniform_knots_reduced_continuity():
    # repeated interior knots (C0 and C1 lines), non-uniform spacing
    U0 = np.array([0, 0, 0, 0, 0.2, 0.2, 0.5, 0.5, 0.5, 0.8, 1, 1, 1, 1.0])
    U1 = np.array([0, 0, 0, 0.1, 0.35, 0.35, 0.7, 1, 1, 1.0])
    orc, eng = make_pair(2, 1, [3, 2], [0, 0], knots=[U0, U1])
    dirichlet_all((orc, eng), 2, 2.0)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    compare_mats(A, A_o, TOL)
    assert rel_err(b.get(), b_o) < TOL


@pytest.mark.parametrize("dim,N,periodic", [(2, 8, False), (2, 9, True), (3, 5, False), (3, 6, True)])
def test_cahn_hilliard_residual_and_tangent(dim, N, periodic):
    # demo/CahnHilliard{2,3}D.c: p=2 C1, random c around cbar, V random, shift = 1/dt
    orc, eng = make_pair(dim, 1, 2, N, periodic=periodic)
    h = 1.0 / np.sqrt(dim * N * N)
    prm = dict(theta=1.5, alpha=200.0, cbar=0.63, L0=1.0 if dim == 3 else -1.0, lam=1.0 * h * h, tau=1.0)
    ctx = O.CahnHilliardCtx(prm["theta"], prm["alpha"], prm["cbar"], prm["L0"], prm["lam"], prm["tau"])
    params = (prm["theta"], prm["alpha"], prm["cbar"], prm["L0"], prm["lam"], prm["tau"])
    rng = np.random.default_rng(7)
    n = orc.global_size()
    U = 0.63 + 0.05 * (2 * rng.random(n) - 1)
    V = rng.standard_normal(n)
    shift, t = 1.0e3, 0.0
    F_o = orc.compute_ifunction("orc_form_ch_residual", ctx, shift, V, t, U)
    J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, shift, V, t, U)
    eng.set_form("cahnhilliard", params)
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(shift, Vv, t, Uv, F)
    eng.compute_ijacobian(shift, Vv, t, Uv, J)
    eng.synchronize()
    assert np.abs(F.get() - F_o).max() <= 1e-11 * np.abs(F_o).max()
    compare_mats(J, J_o, 1e-11)


def test_navier_stokes_vms_residual_and_tangent():
    # demo/NavierStokesVMS.c:362-385: axes 0,2 periodic, no-slip on axis 1; dof 4
    p, N = 2, [6, 3, 5]
    orc, eng = make_pair(3, 4, p, N, periodic=[True, False, True])
    for g in (orc, eng):
        for side in range(2):
            for f in range(3):
                g.set_boundary_value(1, side, f, 0.0)
    nu, fx, dt = 1.472e-4, 3.37204e-3, 1e-2
    ctx = O.NSVMSCtx(nu, fx, 0.0, 0.0, dt)
    params = (nu, fx, 0.0, 0.0, dt)
    rng = np.random.default_rng(11)
    n = orc.global_size()
    U = rng.standard_normal(n) * 0.3
    V = rng.standard_normal(n) * 0.1
    shift = 2.0 / dt
    F_o = orc.compute_ifunction("orc_form_ns_residual", ctx, shift, V, 0.0, U)
    J_o = orc.compute_ijacobian("orc_form_ns_tangent", ctx, shift, V, 0.0, U)
    eng.set_form("nsvms", params)
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(shift, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert np.abs(F.get() - F_o).max() <= 1e-11 * np.abs(F_o).max()
    compare_mats(J, J_o, 1e-11)


def test_matrix_and_vector_drivers_skip_bc():
    # IGAComputeMatrix / IGAComputeVector apply no BC fix-up (src/petigaksp.c:33-125)
    orc, eng = make_pair(2, 1, 2, 6)
    dirichlet_all((orc, eng), 2)
    orc.clear_boundary()
    A_o, b_o = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_matrix(A)
    eng.compute_vector(b)
    eng.synchronize()
    compare_mats(A, A_o, TOL)
    assert rel_err(b.get(), b_o) < TOL


def test_error_behaviour():
    import petiga_amd as P
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, 2, 4)
    g.setup()
    A, b = g.create_mat(), g.create_vec()
    with pytest.raises(P.IGXError) as e:        # IGACheckFormOp: PETSC_ERR_ARG_WRONGSTATE (73)
        g.compute_system(A, b)
    assert e.value.code == 73
    g2 = P.IGX(3, 1)
    with pytest.raises(P.IGXError) as e:        # IGACheckSetUp
        g2.create_mat()
    assert e.value.code == 73
    # tangled geometry -> non-positive Jacobian reported like a debug build of the reference (PETSC_ERR_USER)
    orc, eng = make_pair(2, 1, 2, 3)
    X, _ = warped_geometry(orc, 2, rational=False)
    X[:, 0] *= -1.0
    eng.set_geometry(X)
    eng.set_form("poisson")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    with pytest.raises(P.IGXError) as e:
        eng.synchronize()
    assert e.value.code == 83


# ---------------------------------------------------------------- the MFMA gradient-Gram kernel (metric path)
@pytest.mark.parametrize("N,bc", [((5, 5, 5), "all1"), ((4, 6, 9), "mixed"), ((1, 1, 1), "all1"), ((2, 3, 1), "mixed"), ((8, 8, 8), "none"), ((12, 5, 7), "partial")])
def test_mfma_poisson_p3(N, bc):
    orc, eng = make_pair(3, 1, 3, list(N))
    for g in (orc, eng):
        if bc == "all1":
            dirichlet_all((g,), 3, 1.0)
        elif bc == "mixed":      # a different value on every face: corner/edge nodes take the last face's value
            k = 0
            for d in range(3):
                for s in range(2):
                    g.set_boundary_value(d, s, 0, 0.5 + 0.25 * k)
                    k += 1
        elif bc == "partial":
            g.set_boundary_value(0, 0, 0, 2.0)
            g.set_boundary_value(2, 1, 0, -1.0)
    eng.set_kernel(2)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    assert "mfma" in eng.kernel_name()
    compare_mats(A, A_o, TOL)
    assert np.abs(b.get() - b_o).max() <= TOL * np.abs(b_o).max()
    # the generic kernel and the MFMA kernel agree with each other as well
    eng.set_kernel(1)
    A2, b2 = eng.create_mat(), eng.create_vec()
    eng.compute_system(A2, b2)
    eng.synchronize()
    assert np.abs(A.host(True) - A2.host(True)).max() <= TOL * np.abs(A_o.val).max()


def test_mfma_matrix_driver_and_default_selection():
    orc, eng = make_pair(3, 1, 3, 6)
    dirichlet_all((orc, eng), 3)
    orc.clear_boundary()
    A_o, _ = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A = eng.create_mat()
    eng.set_kernel(0)
    eng.compute_matrix(A)      # kernel 0 = automatic: picks the MFMA kernel for the metric configuration
    eng.synchronize()
    assert "mfma" in eng.kernel_name()
    compare_mats(A, A_o, TOL)


def test_mfma_full_size_properties():
    """Size-independent properties at a large mesh (no oracle): interior rows of the Poisson matrix sum to
    zero (partition of unity), the matrix is symmetric, Dirichlet diagonals equal element multiplicity."""
    import petiga_amd as P
    N = 48
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, 3, N)
    g.setup()
    for d in range(3):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    g.set_form("poisson")
    A, b = g.create_mat(), g.create_vec()
    g.compute_system(A, b)
    g.synchronize()
    M = A.to_scipy_global()
    n = N + 3
    idx = np.arange(n ** 3)
    i0, i1, i2 = idx % n, (idx // n) % n, idx // (n * n)
    onb = (i0 == 0) | (i0 == n - 1) | (i1 == 0) | (i1 == n - 1) | (i2 == 0) | (i2 == n - 1)
    scale = np.abs(M.data).max()
    assert abs(M - M.T).max() <= 1e-12 * scale
    near = (np.minimum(i0, n - 1 - i0) <= 3) | (np.minimum(i1, n - 1 - i1) <= 3) | (np.minimum(i2, n - 1 - i2) <= 3)
    rowsum = np.asarray(M.sum(axis=1)).ravel()
    assert np.abs(rowsum[~near]).max() <= 1e-12 * scale
    mult = lambda i: np.minimum(np.minimum(i + 1, 4), np.minimum(n - i, 4))
    d = M.diagonal()
    assert np.array_equal(d[onb], (mult(i0) * mult(i1) * mult(i2))[onb].astype(float))
    assert np.array_equal(b.get()[onb], d[onb])


# ---------------------------------------------------------------- pencil walk (combine-before-write) specifics
@pytest.mark.parametrize("N,bc,walk", [((70, 5, 5), "all1", None), ((66, 4, 5), "none", None), ((72, 4, 4), "axis0", None),
                                       ((9, 5, 70), "mixed", "2"), ((5, 68, 4), "all1", "1"), ((40, 6, 6), "mixed", None)])
def test_mfma_pencil_segments_and_walk_axes(N, bc, walk, monkeypatch):
    """Long pencils are cut into segments with 3 re-computed halo elements; every walk axis is exercised
    (axis 0: band-row flush with symmetric tiles; axes 1, 2: generic 7-tile flush)."""
    if walk is not None:
        monkeypatch.setenv("IGX_WALK_AXIS", walk)
    orc, eng = make_pair(3, 1, 3, list(N))
    for g in (orc, eng):
        if bc == "all1":
            dirichlet_all((g,), 3, 1.0)
        elif bc == "mixed":
            k = 0
            for d in range(3):
                for s in range(2):
                    g.set_boundary_value(d, s, 0, 0.5 + 0.25 * k)
                    k += 1
        elif bc == "axis0":
            g.set_boundary_value(0, 0, 0, 2.0)
            g.set_boundary_value(0, 1, 0, -1.0)
    eng.set_kernel(2)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    assert "pencil" in eng.kernel_name() and ("walk=%s" % (walk or "0")) in eng.kernel_name()
    compare_mats(A, A_o, TOL)
    assert np.abs(b.get() - b_o).max() <= TOL * max(np.abs(b_o).max(), 1e-300)


def test_mfma_falls_back_when_axis0_not_walkable():
    # axis 0 with C1 lines (two new basis functions per element) cannot be walked: the walk moves to axis 2
    orc, eng = make_pair(3, 1, 3, [10, 5, 12], C=[1, 2, 2])
    dirichlet_all((orc, eng), 3, 1.5)
    eng.set_kernel(2)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    assert "walk=2" in eng.kernel_name()
    compare_mats(A, A_o, TOL)
    assert np.abs(b.get() - b_o).max() <= TOL * np.abs(b_o).max()


def test_mfma_repeatable_bitwise():
    """Colour-ordered, conflict-free scatter: two assemblies of the same system are bit-identical."""
    orc, eng = make_pair(3, 1, 3, [34, 9, 8])
    dirichlet_all((eng,), 3, 0.75)
    eng.set_form("poisson")
    out = []
    for _ in range(2):
        A, b = eng.create_mat(), eng.create_vec()
        eng.compute_system(A, b)
        eng.synchronize()
        out.append((A.host(True), b.get()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


# ---------------------------------------------------------------- multi-rank assembly (all ranks emulated on one GPU)
def _rank_matrix_rows(eng, A, b):
    """(global rows, global cols, values, owned mask per entry), vector entries of owned rows."""
    rows, cols, vals = A.to_coo_global()
    rp, _, _ = A.host()
    nrow, _, _ = A.layout()
    r = np.arange(A.nbrows)
    own = np.array([eng.row_owned(int(a), int(b_), int(c)) for a, b_, c in zip(r % nrow[0], (r // nrow[0]) % nrow[1], r // (nrow[0] * nrow[1]))])
    own_entries = np.repeat(np.repeat(own, np.diff(rp)), A.bs * A.bs)
    return rows, cols, vals, own_entries, own


@pytest.mark.parametrize("size,dim,dof,p,N,periodic,form", [(2, 3, 1, 3, (9, 8, 10), (0, 0, 0), "poisson"), (8, 3, 1, 3, (10, 9, 8), (0, 0, 0), "poisson"),
                                                             (4, 2, 2, 2, (9, 10), (0, 0), "mass"), (4, 3, 1, 2, (8, 8, 8), (1, 0, 1), "poisson"),
                                                             (8, 3, 1, 3, (20, 18, 16), (0, 0, 0), "poisson"),
                                                             (4, 3, 3, 2, (6, 7, 5), (0, 0, 0), "elasticity+nurbs"), (8, 3, 1, 3, (7, 8, 9), (0, 0, 0), "poisson+nurbs"),
                                                             (3, 2, 1, 3, (11, 4), (0, 0), "poisson+nurbs"),
                                                             # axis 0 long enough for the pencil kernel's mapped-geometry variant on every rank
                                                             (2, 3, 1, 3, (9, 4, 8), (0, 0, 0), "poisson+nurbs"), (4, 3, 1, 2, (10, 6, 8), (0, 0, 0), "poisson+nurbs"),
                                                             # ranks thinner than p elements: a ghost layer reaches two (three) ranks up
                                                             (5, 1, 1, 3, (10,), (0,), "poisson"), (6, 2, 1, 3, (6, 8), (0, 0), "poisson"), (7, 1, 2, 3, (7,), (0,), "mass"),
                                                             (27, 3, 1, 2, (3, 4, 3), (0, 0, 0), "poisson"), (12, 3, 1, 3, (4, 6, 5), (0, 0, 0), "poisson+nurbs"),
                                                             # the property array's ghosted box on every rank (IGALoadProperty's scatters, src/petigaio.c:441-446)
                                                             (4, 3, 1, 2, (6, 7, 5), (0, 0, 0), "property+nurbs"), (3, 2, 1, 3, (11, 4), (0, 0), "property")])
def test_multirank_ghost_row_reduction(size, dim, dof, p, N, periodic, form):
    """Every rank assembles its own element box, ghost rows are packed / added through the C ABI exactly as
    petiga_amd/exchange.py does between processes; the owned rows of all ranks together must be the
    single-rank matrix of the oracle."""
    import torch
    import petiga_amd as P
    periodic = [bool(x) for x in periodic]
    orc, _ = make_pair(dim, dof, p, list(N), periodic=periodic, engine=False)
    geo = form.endswith("+nurbs")
    form = form.split("+")[0]
    if geo:
        X, W = warped_geometry(orc, dim, seed=9, rational=True, amp=0.1)
        orc.set_geometry(X, W)
    if form == "poisson":
        dirichlet_all((orc,), dim, 1.0)
    PA = None
    if form == "property":
        PA = 1.0 + np.random.default_rng(21).random((orc.global_size(), 2))
        orc.set_property(PA)
    octx, prm = (O.ElasticityCtx(1.3, 0.8), (1.3, 0.8)) if form == "elasticity" else (None, ())
    A_o, b_o = orc.compute_system("orc_form_" + form, octx)
    engs, mats, vecs, sendbufs = [], [], [], {}
    for r in range(size):
        g = P.IGX(dim, dof)
        for i in range(dim):
            g.axis_uniform(i, p, N[i], periodic=periodic[i])
        g.set_comm(size, r)
        g.setup()
        if geo:
            g.set_geometry(X, W)          # the global control net; the rank keeps its ghosted box
        if PA is not None:
            g.set_property(PA)
        if form == "poisson":
            dirichlet_all((g,), dim, 1.0)
        g.set_form(form, prm)
        A, b = g.create_mat(), g.create_vec()
        _poison(A)                       # stale values must not survive (first-touch stores + neighbour-row zeroing)
        g.compute_system(A, b)
        for k, (peer, m, v) in enumerate(g.neighbors(True)):
            buf = torch.empty(m + v, dtype=torch.float64, device="cuda")
            g.pack_ghost_rows(A, b, k, buf.data_ptr())
            sendbufs[(r, peer)] = buf
        g.synchronize()
        engs.append(g); mats.append(A); vecs.append(b)
    for r, g in enumerate(engs):
        for k, (peer, m, v) in enumerate(g.neighbors(False)):
            buf = sendbufs[(peer, r)]
            assert buf.numel() == m + v
            g.unpack_ghost_rows(mats[r], vecs[r], k, buf.data_ptr())
        g.synchronize()
    n = orc.global_size()
    import scipy.sparse as sp
    M = sp.csr_matrix((n, n))
    F = np.zeros(n)
    seen = np.zeros(n, dtype=int)
    for r, g in enumerate(engs):
        rows, cols, vals, own_e, own = _rank_matrix_rows(g, mats[r], vecs[r])
        M = M + sp.coo_matrix((vals[own_e], (rows[own_e], cols[own_e])), shape=(n, n)).tocsr()
        nrow, _, maps = mats[r].layout()
        ns = g.sizes()["node_sizes"]
        rr = np.arange(mats[r].nbrows)
        grow = maps[0][0][rr % nrow[0]].astype(np.int64) + ns[0] * (maps[1][0][(rr // nrow[0]) % nrow[1]].astype(np.int64) + ns[1] * maps[2][0][rr // (nrow[0] * nrow[1])].astype(np.int64))
        bv = vecs[r].get().reshape(-1, dof)
        for c in range(dof):
            F[grow[own] * dof + c] = bv[own, c]
            seen[grow[own] * dof + c] += 1
    assert np.all(seen == 1)
    Mo = A_o.scipy()
    D = abs(M - Mo)
    assert D.max() <= TOL * abs(Mo).max()
    assert np.abs(F - b_o).max() <= TOL * max(np.abs(b_o).max(), 1e-300)


# ---------------------------------------------------------------- the PetIGA-binding route: IGXCreateFromTables
def _tables_from_oracle(orc, dim, dof, keep):
    """Fill IGXTables from the oracle's IGA struct, field for field what a set-up PetIGA `IGA` holds
    (struct _p_IGA / _n_IGAAxis / _n_IGABasis, include/petiga.h:80-141,327-391)."""
    import petiga_amd as P
    t = P.IGXTables()
    s = orc.s
    t.dim, t.dof, t.order = dim, dof, s.order
    for i in range(dim):
        ax, bd, a = s.axis[i], s.basis[i], t.axis[i]
        a.p, a.m, a.periodic, a.nel, a.nnp = ax.p, ax.m, ax.periodic, ax.nel, ax.nnp
        a.U, a.span = ax.U, ax.span
        a.nqp, a.nen, a.offset, a.detJac, a.weight, a.point, a.value = bd.nqp, bd.nen, bd.offset, bd.detJac, bd.weight, bd.point, bd.value
    for name in ("proc_sizes", "proc_ranks", "elem_sizes", "elem_start", "elem_width", "node_sizes", "node_lstart", "node_lwidth", "node_gstart", "node_gwidth"):
        for i in range(3):
            getattr(t, name)[i] = getattr(s, name)[i]
    t.nsd, t.rational = s.nsd, s.rational
    t.geometryX, t.rationalW = s.geometryX, s.rationalW
    t.property, t.propertyA = s.property, s.propertyA
    keep.append(t)
    return t


@pytest.mark.parametrize("dim,dof,p,N,geo,form", [(3, 1, 3, (9, 6, 7), None, "poisson"), (3, 1, 2, (5, 6, 4), "nurbs", "poisson"),
                                                   (2, 2, 3, (7, 6), None, "mass"), (3, 3, 2, (4, 4, 3), None, "elasticity"),
                                                   (3, 1, 2, (4, 3, 5), "nurbs", "property")])      # iga->property / iga->propertyA handed over with the tables
def test_create_from_tables_matches_oracle(dim, dof, p, N, geo, form):
    import petiga_amd as P
    orc, _ = make_pair(dim, dof, p, list(N), engine=False)
    if geo:
        X, W = warped_geometry(orc, dim, seed=3, rational=True)
        orc.set_geometry(X, W)
    if form == "property":
        orc.set_property(1.0 + np.random.default_rng(4).random((orc.global_size(), 2)))
    keep = []
    eng = P.IGX.from_tables(_tables_from_oracle(orc, dim, dof, keep))
    ctx, params = None, ()
    if form == "poisson":
        dirichlet_all((orc, eng), dim, 1.25)
    if form == "elasticity":
        ctx, params = O.ElasticityCtx(1.0, 1.0), (1.0, 1.0)
        for g in (orc, eng):
            for f in range(3):
                g.set_boundary_value(0, 0, f, 0.0)
            g.set_boundary_value(0, 1, 0, 1.0)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_" + form, form, ctx, params)
    compare_mats(A, A_o, 1e-11 if geo else TOL)
    assert np.abs(b.get() - b_o).max() <= (1e-11 if geo else TOL) * max(np.abs(b_o).max(), 1.0)


@pytest.mark.parametrize("pack", [1, 0])
@pytest.mark.parametrize("N,bc", [((10, 5, 6), "all1"), ((70, 4, 5), "mixed"), ((9, 9, 9), "none"), ((8, 1, 1), "all1"), ((33, 7, 3), "axis0")])
def test_mfma_pencil_degree2(N, bc, pack, monkeypatch):
    """BASELINE config 2 family (demo/Poisson3D.c at p=2 C1), band width 5: the packed tiles of round 5 (27 functions in two MFMA
    tiles, band rows combined in an LDS window) and, with IGX_P2_PACK=0, the kernel they replaced (the 3x3x3 basis zero-padded into
    4x4 tile slots, one tile per pair of node layers) -- still the code of p = 2 on mapped geometries, so it stays tested."""
    monkeypatch.setenv("IGX_P2_PACK", str(pack))      # (read when an IGX is created)
    monkeypatch.setenv("IGX_PATCH", "0")              # (the pencil walk itself: since round 6 the patch walk takes these cases by default, tests/test_gpu_patch.py)
    orc, eng = make_pair(3, 1, 2, list(N))
    for g in (orc, eng):
        if bc == "all1":
            dirichlet_all((g,), 3, 1.0)
        elif bc == "mixed":
            k = 0
            for d in range(3):
                for s in range(2):
                    g.set_boundary_value(d, s, 0, 0.5 + 0.25 * k)
                    k += 1
        elif bc == "axis0":
            g.set_boundary_value(0, 0, 0, 2.0)
            g.set_boundary_value(0, 1, 0, -1.0)
    eng.set_kernel(2)
    A, b, A_o, b_o = system_pair(orc, eng, "orc_form_poisson", "poisson")
    assert "p=2" in eng.kernel_name() and "pencil" in eng.kernel_name() and (("packed tiles" in eng.kernel_name()) == bool(pack)), eng.kernel_name()
    compare_mats(A, A_o, TOL)
    assert np.abs(b.get() - b_o).max() <= TOL * max(np.abs(b_o).max(), 1e-300)
    eng.set_form("poisson")
    A2 = eng.create_mat()
    eng.compute_matrix(A2)          # IGAComputeMatrix: no BC fix-up
    eng.synchronize()
    orc.clear_boundary()
    A2_o, _ = orc.compute_system("orc_form_poisson")
    compare_mats(A2, A2_o, TOL)


@pytest.mark.parametrize("case", ["elasticity", "ch3d", "ns", "poisson_nurbs", "mass4_2d", "p1"])
def test_feature_mfma_kernel_is_selected_and_matches(case):
    """The automatic choice routes matrix-producing operations of every form to the MFMA feature-GEMM kernel
    (generic kernel: vectors only, dim 1, nen > 64); its result equals the generic kernel's to 1e-12."""
    import petiga_amd as P
    rng = np.random.default_rng(5)
    if case == "elasticity":
        _, eng = make_pair(3, 3, 3, 3); form, prm, op = "elasticity", (2.0, 0.5), "system"
        for f in range(3):
            eng.set_boundary_value(0, 0, f, 0.25 * f)
    elif case == "ch3d":
        _, eng = make_pair(3, 1, 2, 4, order=2); form, prm, op = "cahnhilliard", (1.5, 200.0, 0.63, 1.0, 1e-3, 1.0), "ijacobian"
    elif case == "ns":
        _, eng = make_pair(3, 4, 3, [7, 3, 8], periodic=[True, False, True], order=2); form, prm, op = "nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), "ijacobian"
        for s in range(2):
            for f in range(3):
                eng.set_boundary_value(1, s, f, 0.0)
    elif case == "poisson_nurbs":
        orc, eng = make_pair(3, 1, (3, 2, 3), (3, 4, 3)); form, prm, op = "poisson", (), "system"
        X, W = warped_geometry(orc, 3, seed=2, rational=True)
        eng.set_geometry(X, W)
        eng.set_boundary_value(2, 1, 0, 1.5)
    elif case == "mass4_2d":
        _, eng = make_pair(2, 4, 3, 5); form, prm, op = "mass", (), "system"
    else:
        _, eng = make_pair(3, 1, 1, 5); form, prm, op = "poisson", (), "system"
        eng.set_boundary_value(0, 0, 0, 1.0)
    eng.set_form(form, prm)
    n = eng.create_vec().n
    U = eng.create_vec().set(0.5 + 0.1 * rng.standard_normal(n)); V = eng.create_vec().set(0.1 * rng.standard_normal(n))
    out = {}
    for k in (0, 1):
        eng.set_kernel(k)
        A, b = eng.create_mat(), eng.create_vec()
        if op == "system":
            eng.compute_system(A, b)
        else:
            eng.compute_ijacobian(2.5, V, 0.1, U, A)
        eng.synchronize()
        out[k] = (A.host(True), b.get(), eng.kernel_name())
    assert "feature_assemble(mfma" in out[0][2] and "generic" in out[1][2]
    scale = np.abs(out[1][0]).max()
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-12 * scale
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-12 * max(np.abs(out[1][1]).max(), 1.0)
    # bitwise repeatable, and independent of what the matrix held before (first-touch stores, no MatZeroEntries)
    eng.set_kernel(0)
    A2 = eng.create_mat(); b2 = eng.create_vec()
    _poison(A2)
    if op == "system":
        eng.compute_system(A2, b2)
    else:
        eng.compute_ijacobian(2.5, V, 0.1, U, A2)
    assert np.array_equal(A2.host(True), out[0][0])


@pytest.mark.parametrize("p,N,size", [(3, (9, 10, 11), 1), (3, (16, 5, 4), 1), (2, (8, 9, 10), 1), (3, (12, 9, 10), 2), (3, (16, 9, 20), 2), (3, (16, 16, 16), 8), (2, (9, 3, 1), 1)])
def test_pencil_first_touch_needs_no_zeroing(p, N, size, monkeypatch):
    """The axis-0 pencil walk stores the first contribution of every entry (no MatZeroEntries): poisoned matrices
    must come out identical to freshly zeroed ones, on every rank of a partition."""
    import petiga_amd as P
    monkeypatch.setenv("IGX_PATCH", "0")      # (the bit-repeatable pencil walk; the patch walk's first touch: tests/test_gpu_patch.py)
    for r in range(size):
        g = P.IGX(3, 1)
        for i in range(3):
            g.axis_uniform(i, p, N[i])
        g.set_comm(size, r)
        g.setup()
        g.set_boundary_value(0, 0, 0, 1.0); g.set_boundary_value(1, 1, 0, -2.0); g.set_boundary_value(2, 0, 0, 0.5)
        g.set_form("poisson")
        g.set_kernel(2)
        A, b = g.create_mat(), g.create_vec()
        g.compute_system(A, b); g.synchronize()
        assert "mfma" in g.kernel_name() and (size > 1 or "walk=0" in g.kernel_name())   # other walks zero the matrix first
        ref = A.host(True)
        assert np.all(np.isfinite(ref))
        _poison(A)
        g.compute_system(A, b); g.synchronize()
        assert np.array_equal(A.host(True), ref)
        _poison(A)
        g.compute_matrix(A); g.synchronize()
        assert np.all(np.isfinite(A.host(True)))
        g.set_kernel(1)                      # and the zeroed read-modify-write path of the generic kernel agrees
        A2, b2 = g.create_mat(), g.create_vec()
        g.compute_system(A2, b2); g.synchronize()
        assert np.abs(A2.host(True) - ref).max() <= TOL * np.abs(ref).max()


@pytest.mark.parametrize("size,form,N,periodic", [(4, "ch", (6, 7, 8), (0, 0, 0)), (8, "ch", (8, 8, 8), (0, 0, 0)), (2, "ns", (9, 3, 8), (1, 0, 1)), (4, "ns", (9, 4, 8), (1, 0, 1)),
                                                  (2, "ch", (16, 4, 6), (0, 0, 0)), (4, "ch", (9, 6, 10), (1, 0, 0)),      # boxes the pencil walk takes: the fused pass on a rank's share
                                                  (27, "ch", (3, 4, 3), (0, 0, 0)),       # one element of degree 2 per rank and axis: ghost values come from two ranks up
                                                  (2, "chg", (24, 6, 6), (0, 0, 0)), (4, "chg", (20, 12, 5), (0, 0, 0))])      # Cahn-Hilliard on a NURBS patch: state_pencil_geo + vec_sumfact on a partition
def test_multirank_nonlinear_assembly_with_ghost_refresh(size, form, N, periodic, monkeypatch):
    """Nonlinear drivers on a partition (configs 4 and 5 are multi-GPU): every rank knows the state only on the nodes
    it owns, the owner -> ghost refresh (IGXPackOwnerValues / IGXUnpackGhostValues, the reverse of the ghost-row
    reduction) fills its ghost rows, then IFunction / IJacobian + ghost-row reduction reproduce the single-rank oracle."""
    import torch
    import scipy.sparse as sp
    import petiga_amd as P
    monkeypatch.setenv("IGX_FUSE_RESID", "1")      # (read at IGXCreate: the fused pass is checked on every rank's box below)
    periodic = [bool(x) for x in periodic]
    geo = form == "chg"
    form = "ch" if geo else form
    dof, p = (1, 2) if form == "ch" else (4, 2)
    orc, _ = make_pair(3, dof, p, list(N), periodic=periodic, order=2, engine=False)
    if geo:
        from common import warped_geometry
        Xg, Wg = warped_geometry(orc, 3, seed=12, rational=True, amp=0.05)
        orc.set_geometry(Xg, Wg)
    if form == "ns":
        for s_ in range(2):
            for f in range(3):
                orc.set_boundary_value(1, s_, f, 0.0)
    n = orc.global_size()
    rng = np.random.default_rng(11)
    Ug, Vg = (0.63 + 0.05 * (2 * rng.random(n) - 1), 0.1 * rng.standard_normal(n)) if form == "ch" else (0.3 * rng.standard_normal(n), 0.1 * rng.standard_normal(n))
    if form == "ch":
        ctx, prm, fres, ftan, ename = O.CahnHilliardCtx(1.5, 200.0, 0.63, 1.0, 1e-3, 1.0), (1.5, 200.0, 0.63, 1.0, 1e-3, 1.0), "orc_form_ch_residual", "orc_form_ch_tangent", "cahnhilliard"
    else:
        ctx, prm, fres, ftan, ename = O.NSVMSCtx(1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2), "orc_form_ns_residual", "orc_form_ns_tangent", "nsvms"
    F_o = orc.compute_ifunction(fres, ctx, 2.0, Vg, 0.1, Ug)
    J_o = orc.compute_ijacobian(ftan, ctx, 2.0, Vg, 0.1, Ug)
    engs, Us, Vs, grows, owns = [], [], [], [], []
    for r in range(size):
        g = P.IGX(3, dof)
        for i in range(3):
            g.axis_uniform(i, p, N[i], periodic=periodic[i])
        g.set_order(2); g.set_comm(size, r); g.setup()
        if form == "ns":
            for s_ in range(2):
                for f in range(3):
                    g.set_boundary_value(1, s_, f, 0.0)
        g.set_form(ename, prm)
        if geo:
            g.set_geometry(Xg, Wg)      # (the global net: every rank takes its part)
        A = g.create_mat()
        nrow, _, maps = A.layout()
        ns = g.sizes()["node_sizes"]
        rr = np.arange(A.nbrows)
        grow = maps[0][0][rr % nrow[0]].astype(np.int64) + ns[0] * (maps[1][0][(rr // nrow[0]) % nrow[1]].astype(np.int64) + ns[1] * maps[2][0][rr // (nrow[0] * nrow[1])].astype(np.int64))
        own = np.array([g.row_owned(int(a), int(b_), int(c)) for a, b_, c in zip(rr % nrow[0], (rr // nrow[0]) % nrow[1], rr // (nrow[0] * nrow[1]))])
        def local(vg):
            v = np.full((A.nbrows, dof), np.nan)
            v[own] = vg.reshape(-1, dof)[grow[own]]
            return g.create_vec().set(v.reshape(-1))
        engs.append(g); Us.append(local(Ug)); Vs.append(local(Vg)); grows.append(grow); owns.append(own)
    # owner -> ghost refresh, both state vectors
    for vecs in (Us, Vs):
        msgs = {}
        for r, g in enumerate(engs):
            for k, (peer, _, v) in enumerate(g.neighbors(False)):
                buf = torch.empty(max(v, 1), dtype=torch.float64, device="cuda")
                g.pack_owner_values(vecs[r], k, buf.data_ptr()); g.synchronize()
                assert (r, peer) not in msgs
                msgs[(r, peer)] = buf
        for r, g in enumerate(engs):
            for k, (peer, _, v) in enumerate(g.neighbors(True)):
                assert msgs[(peer, r)].numel() == max(v, 1)
                g.unpack_ghost_values(vecs[r], k, msgs[(peer, r)].data_ptr())
            g.synchronize()
    for r in range(size):
        assert np.array_equal(Us[r].get().reshape(-1, dof), Ug.reshape(-1, dof)[grows[r]])      # every ghost row holds its owner's value
    # assemble, reduce ghost rows, compare the owned rows with the single-rank oracle
    M = sp.csr_matrix((n, n)); F = np.zeros(n)
    mats, vecsF, send = [], [], {}
    for r, g in enumerate(engs):
        A, b = g.create_mat(), g.create_vec()
        _poison(A)
        g.compute_ifunction(2.0, Vs[r], 0.1, Us[r], b)
        if geo and os.environ.get("IGX_KERNEL") == "0":
            assert "vec_sumfact" in g.kernel_name(), g.kernel_name()
        g.compute_ijacobian(2.0, Vs[r], 0.1, Us[r], A)
        if geo and os.environ.get("IGX_KERNEL") == "0":
            assert "state_pencil<CahnHilliard>" in g.kernel_name() and "mapped geometry" in g.kernel_name(), g.kernel_name()
        if form == "ch" and not geo and os.environ.get("IGX_KERNEL") == "0":
            # the fused pass (IGXComputeIFunctionIJacobian: the Residual on the Tangent's MFMAs) on this rank's box -- ghost rows,
            # halo segments, rows zeroed for the neighbours' columns -- gives what the two drivers gave
            A2, b2 = g.create_mat(), g.create_vec()
            _poison(A2)
            g.compute_ifunction_ijacobian(2.0, Vs[r], 0.1, Us[r], b2, A2)
            g.synchronize()
            assert ("+Residual>" in g.kernel_name()) == (g.sizes()["elem_width"][0] >= 8), g.kernel_name()      # (a box the walk does not take: the two drivers, one after the other)
            assert np.abs(b2.get() - b.get()).max() <= 1e-12 * np.abs(b.get()).max()
            assert np.abs(A2.host(True) - A.host(True)).max() <= 1e-12 * np.abs(A.host(True)).max()
        for k, (peer, m, v) in enumerate(g.neighbors(True)):
            buf = torch.empty(m + v, dtype=torch.float64, device="cuda")
            g.pack_ghost_rows(A, b, k, buf.data_ptr()); send[(r, peer)] = buf
        g.synchronize(); mats.append(A); vecsF.append(b)
    for r, g in enumerate(engs):
        for k, (peer, m, v) in enumerate(g.neighbors(False)):
            g.unpack_ghost_rows(mats[r], vecsF[r], k, send[(peer, r)].data_ptr())
        g.synchronize()
        rows, cols, vals, own_e, own = _rank_matrix_rows(g, mats[r], vecsF[r])
        M = M + sp.coo_matrix((vals[own_e], (rows[own_e], cols[own_e])), shape=(n, n)).tocsr()
        bv = vecsF[r].get().reshape(-1, dof)
        for c in range(dof):
            F[grows[r][own] * dof + c] = bv[own, c]
    Mo = J_o.scipy()
    assert abs(M - Mo).max() <= (1e-10 if geo else 1e-11) * abs(Mo).max()
    assert np.abs(F - F_o).max() <= (1e-10 if geo else 1e-11) * np.abs(F_o).max()


def test_user_stream_and_event_timing(kernel_family):
    """IGXSetStream: the assembly runs on a caller-provided HIP stream (torch's), results unchanged; the HIP-event
    timings bench.py reads are positive and the dominant kernel is reported."""
    import torch
    orc, eng = make_pair(3, 1, 3, [9, 8, 8])
    dirichlet_all((orc, eng), 3)
    eng.set_form("poisson")
    A0, b0 = eng.create_mat(), eng.create_vec()
    eng.compute_system(A0, b0); eng.synchronize()
    st = torch.cuda.Stream()
    eng.set_stream(st.cuda_stream)
    eng.set_timing(True)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    st.synchronize()
    assert np.array_equal(A.host(True), A0.host(True)) and np.array_equal(b.get(), b0.get())
    total_ms, kernel_ms, launches = eng.last_timing()
    assert total_ms > 0 and kernel_ms > 0 and launches >= 1
    if kernel_family == "auto":
        d = eng.dominant_kernel()
        assert d["launches"] >= 1 and d["ms"] > 0 and d["elements"] == 9 * 8 * 8
    eng.set_stream(None)
