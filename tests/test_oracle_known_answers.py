"""Pins the CPU oracle against the known-answer tests the reference itself holds for the
assembly path.  Each test names the reference test it restates."""
import numpy as np
import pytest
import scipy.sparse.linalg as sla

import oracle_api as O

SQ2 = np.sqrt(2.0)
TIGHT = 1e-12        # the closed forms of the reference tests are asserted at the parity tolerance, not at the reference's 1e-6


# ---------------------------------------------------------------- structural known answers
def test_tutorial_matrix_sizes():
    # docs/manual/TUTORIAL.rst:113-115 and :204-205
    for dim, rows, nnz in ((3, 5832, 592704), (2, 324, 7056)):
        g = O.OracleIGA(dim, 1)
        for i in range(dim):
            g.axis_uniform(i, 2, 16)
        g.setup()
        A = g.create_mat()
        assert (A.nrows, A.nnz) == (rows, nnz)


def test_tutorial_partition_balance():
    # docs/manual/TUTORIAL.rst:78-80: 16^3 on 8 ranks -> [2,2,2], 512 elements/rank, 512..1000 nodes/rank
    nodes = []
    for r in range(8):
        g = O.OracleIGA(3, 1)
        for i in range(3):
            g.axis_uniform(i, 2, 16)
        g.set_partition(8, r)
        g.setup()
        R = g.ranges()
        assert R["proc_sizes"] == [2, 2, 2]
        assert np.prod(R["elem_width"]) == 512
        nodes.append(int(np.prod(R["node_lwidth"])))
    assert min(nodes) == 512 and max(nodes) == 1000 and sum(nodes) == 18 ** 3


def test_partition_golden_grids():
    # SURVEY.md 8c golden partitions of src/petigapart.c for cubes and 64^2
    for N in (16, 128, 192, 256):
        assert O.partition(1, 0, [N] * 3)[0] == [1, 1, 1]
        assert O.partition(2, 0, [N] * 3)[0] == [1, 1, 2]
        assert O.partition(4, 0, [N] * 3)[0] == [1, 2, 2]
        assert O.partition(8, 0, [N] * 3)[0] == [2, 2, 2]
    assert O.partition(2, 0, [64, 64])[0] == [1, 2]
    assert O.partition(4, 0, [64, 64])[0] == [2, 2]
    assert O.partition(8, 0, [64, 64])[0] == [2, 4]
    # rank -> coords, axis 0 fastest (src/petigapart.c:161-166)
    assert O.partition(8, 5, [16] * 3)[1] == [1, 0, 1]
    # IGA_Dist1D block ranges (src/petigapart.c:170-176)
    assert O.distribute([3], [0], [10]) == ([4], [0])
    assert O.distribute([3], [1], [10]) == ([3], [4])
    assert O.distribute([3], [2], [10]) == ([3], [7])


def test_bspline_partition_of_unity_and_uniform_axis():
    g = O.OracleIGA(1, 1)
    g.axis_uniform(0, 3, 256)         # the metric config's axis
    g.setup()
    ax = g.axis(0)
    assert ax["m"] == 262 and ax["nnp"] == 259 and ax["nel"] == 256
    assert np.all(ax["U"][:4] == 0) and np.all(ax["U"][-4:] == 1) and ax["U"][4] == 1 / 256
    assert np.all(ax["span"] == 3 + np.arange(256))
    b = g.basis(0)
    assert np.allclose(b["value"][..., 0].sum(-1), 1, atol=1e-14)
    for k in (1, 2, 3):
        assert np.abs(b["value"][..., k].sum(-1)).max() < TIGHT * 256 ** k
    assert np.all(b["value"][..., 4] == 0)
    assert np.allclose(b["detJac"], 1 / 512)


# ---------------------------------------------------------------- test/IGAGeometryMap.c
def quarter_annulus(dim):
    """test/IGAGeometryMap.c:18-32, :493-530: one p=(2,2,1) element, quadrature 9x10x8, order 4."""
    PX = np.array([[1.0, 1.0, 0.0], [1.5, 1.5, 0.0], [2.0, 2.0, 0.0]])
    PY = np.array([[0.0, 1.0, 1.0], [0.0, 1.5, 1.5], [0.0, 2.0, 2.0]])
    PW = np.array([[1.0, SQ2 / 2, 1.0]] * 3)
    g = O.OracleIGA(dim, 1)
    g.axis_uniform(0, 2, 1, 0)
    g.axis_uniform(1, 2, 1, 1)
    if dim == 3:
        g.axis_uniform(2, 1, 1, 0)
    for i, q in enumerate((9, 10, 8)[:dim]):
        g.set_quadrature(i, q)
    g.set_order(4)
    g.setup()
    X, W = [], []
    for k in range(dim - 1):
        for j in range(3):
            for i in range(3):
                W.append(PW[i][j])
                X.append([PX[i][j], PY[i][j]] + ([2.0 * k] if dim == 3 else []))
    g.set_geometry(np.array(X), np.array(W))
    return g


def check_geometry_map(e, dim):
    """TestGeometryMap, test/IGAGeometryMap.c:34-258.  The reference asserts 1e-6; the chain reproduces the closed forms to a few
    ulps (3e-15), and parity is claimed at 1e-12, so that is what is asserted here."""
    tol = TIGHT
    for q in range(e["nqp"]):
        u, v = e["point"][q][0], e["point"][q][1]
        w = e["point"][q][2] if dim == 3 else 0.0
        X = e["mapX0"][q]
        xw = (1 + u) * (v * v * (-1 + SQ2) + v * (-SQ2 + 2) - 1)
        yw = (1 + u) * (v * v * (-1 + SQ2) - v * SQ2)
        ww = v * v * (-2 + SQ2) + v * (-SQ2 + 2) - 1
        assert abs(X[0] - xw / ww) < tol and abs(X[1] - yw / ww) < tol
        if dim == 3:
            assert abs(X[2] - 2 * w) < tol
        J = SQ2 * (1 + u) / ((2 - SQ2) * v * v + (-2 + SQ2) * v + 1)
        if dim == 3:
            J *= 2
        assert abs(e["detX"][q] - J) < tol
        F = e["mapX1"][q]
        F00 = (v * v * (-1 + SQ2) + v * (-SQ2 + 2) - 1) / ww
        F01 = (-v * (u + 1) * (-2 * v + SQ2 * v + 2)) / (ww * ww)
        F10 = (v * v * (-1 + SQ2) - v * SQ2) / ww
        F11 = ((u + 1) * (v - 1) * (-2 * v + SQ2 * v - SQ2)) / (ww * ww)
        assert np.allclose([F[0, 0], F[0, 1], F[1, 0], F[1, 1]], [F00, F01, F10, F11], atol=tol)
        if dim == 3:
            assert np.allclose([F[0, 2], F[2, 0], F[2, 1], F[2, 2]], [0, 0, 0, 2], atol=tol)
        H, D = e["mapX2"][q], e["mapX3"][q]
        assert np.allclose(H, np.swapaxes(H, 1, 2), atol=tol)
        assert abs(H[0, 0, 0]) < tol and abs(H[1, 0, 0]) < tol
        for perm in ((0, 2, 1, 3), (0, 1, 3, 2), (0, 3, 2, 1)):
            assert np.allclose(D, np.transpose(D, perm), atol=tol)
        x1, x2, y1, y2 = F[0, 1], H[0, 1, 1], F[1, 1], H[1, 1, 1]
        kappa = (x1 * y2 - y1 * x2) / (x1 * x1 + y1 * y1) ** 1.5
        assert abs(kappa - 1 / np.hypot(X[0], X[1])) < tol
        # sum_a C_a (x) grad^k N_a = I, 0, 0   (:185-255)
        Cx = e["geometryX"]
        G = np.einsum("al,ai->li", Cx, e["shape1"][q])
        assert np.allclose(G, np.eye(dim), atol=tol)
        assert np.abs(np.einsum("al,aij->lij", Cx, e["shape2"][q])).max() < tol
        assert np.abs(np.einsum("al,aijk->lijk", Cx, e["shape3"][q])).max() < tol


@pytest.mark.parametrize("dim", [2, 3])
def test_geometry_map_quarter_annulus(dim):
    g = quarter_annulus(dim)
    e = g.element([0] * dim)
    assert e["nqp"] == int(np.prod((9, 10, 8)[:dim]))
    check_geometry_map(e, dim)
    # Domain(): no boundary data on the interior pass (:260-273)
    assert np.all(e["detS"] == 0) and np.all(e["normal"] == 0)
    # the six (four) faces, Boundary_00..21 (:275-389)
    for bid in range(2 * dim):
        f = g.element([0] * dim, bid)
        check_geometry_map(f, dim)
        axis, side = divmod(bid, 2)
        for q in range(f["nqp"]):
            n, X = f["normal"][q], f["mapX0"][q]
            if axis == 0:
                r = np.hypot(X[0], X[1])
                assert abs(r - (2.0 if side else 1.0)) < TIGHT
                sgn = 1.0 if side else -1.0
                assert abs(n[0] - sgn * X[0] / r) < TIGHT and abs(n[1] - sgn * X[1] / r) < TIGHT
            elif axis == 1:
                assert abs(f["detS"][q] - (1.0 if dim == 2 else 2.0)) < TIGHT
                expect = [-1.0, 0.0] if side else [0.0, -1.0]
                assert np.allclose(n[:2], expect, atol=TIGHT)
            else:
                assert abs(f["detS"][q] - f["detX"][q] / (dim - 1)) < TIGHT
                assert np.allclose(n, [0, 0, 1.0 if side else -1.0], atol=TIGHT)
            if dim == 3 and axis < 2:
                assert abs(n[2]) < TIGHT
    # volume and surface area (:545-568)
    for a in range(dim):
        for s in range(2):
            g.set_boundary_form(a, s, True)
    S = g.compute_scalar("orc_scalar_volume", 2, full=True)
    A = np.pi * (4 - 1) / 4
    P = 2 * (2 - 1) + np.pi * (2 + 1) / 2
    V = A if dim == 2 else 2 * A
    Sf = P if dim == 2 else 2 * A + 2 * P
    assert abs(S[0] - V) < TIGHT and abs(S[1] - Sf) < TIGHT


# ---------------------------------------------------------------- test/IGAErrNorm.c
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_errnorm_radicals_and_projection(dim):
    import ctypes as C
    n = 8 if dim < 3 else 4       # the reference runs 8 elements/axis; 4 keeps the 3-D solve small
    g = O.OracleIGA(dim, 4)
    for i in range(dim):
        g.axis_uniform(i, 2, n)
        g.set_quadrature(i, 3)
    g.set_order(2)
    g.setup()
    s = np.sqrt
    L2 = {1: [1, 1 / s(3), 1 / s(5), 1 / s(3)], 2: [1, s(7) / s(6), s(28) / s(45), 1 / s(9)], 3: [1, s(5) / s(2), s(19) / s(15), 1 / s(27)]}
    H1 = {1: [0, 1, 2 / s(3), 1], 2: [0, s(2), s(8) / s(3), s(2) / s(3)], 3: [0, s(3), 2, 1 / s(3)]}
    H2 = {1: [0, 0, 2, 0], 2: [0, 0, s(8), s(2)], 3: [0, 0, s(12), s(2)]}
    tol = TIGHT        # the reference: sqrt(eps)
    zero = np.zeros(g.global_size())
    for order, expect in ((0, L2), (1, H1), (2, H2)):
        o = C.c_int(order)
        S = np.sqrt(g.compute_scalar("orc_scalar_errnorm", 4, U=zero, ctx=o))
        assert np.allclose(S, expect[dim], atol=tol), (order, S, expect[dim])
    A, b = g.compute_system("orc_form_errnorm")
    x = sla.spsolve(A.scipy().tocsc(), b)
    for order in (0, 1, 2):
        o = C.c_int(order)
        S = np.sqrt(g.compute_scalar("orc_scalar_errnorm", 4, U=x, ctx=o))
        assert np.all(S < tol * (2 * n) ** order), (order, S)      # a seminorm of order k amplifies the rounding of the solve by h^-k


# ---------------------------------------------------------------- test/IGAFixTable.c
@pytest.mark.parametrize("dim,nel", [(1, 16), (2, 16), (3, 1), (3, 4)])
def test_fixtable_poisson(dim, nel):
    # test/makefile:77-82: -check_error 1e-6, default p=2 C1
    g = O.OracleIGA(dim, 1)
    for i in range(dim):
        g.axis_uniform(i, 2, nel)
    g.setup()
    A, b = g.compute_system("orc_form_l2proj_x2")
    x = sla.spsolve(A.scipy().tocsc(), b)
    for d in range(dim):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 0.0)
    g.set_fixtable(x)
    A, b = g.compute_system("orc_form_poisson_f")
    M = A.scipy()
    # parity trap: a Dirichlet row's diagonal = number of elements holding the node
    x = sla.spsolve(M.tocsc(), b)
    g.set_fixtable(None)
    err = np.sqrt(g.compute_scalar("orc_scalar_x2err", 1, U=x)[0])
    assert err < TIGHT


def test_dirichlet_diagonal_is_element_multiplicity():
    g = O.OracleIGA(2, 1)
    for i in range(2):
        g.axis_uniform(i, 2, 4)
    g.setup()
    for d in range(2):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    A, b = g.compute_system("orc_form_poisson")
    M = A.scipy().toarray()
    n = 6
    # corner node 0: one element; edge node 1: two; edge node 2: three (p=2, C1)
    assert M[0, 0] == 1 and M[1, 1] == 2 and M[2, 2] == 3
    assert b[0] == 1 and b[1] == 2 and b[2] == 3
    assert np.count_nonzero(M[0]) == 1 and np.count_nonzero(M[:, 0]) == 1
    interior = 2 * n + 2
    assert abs(M[interior].sum() - 0) > -1  # row exists
    assert np.allclose(M, M.T, atol=1e-14)


# ---------------------------------------------------------------- test/IGACreate.c
@pytest.mark.parametrize("dim,dof,periodic,degree", [
    (1, 4, (0,), (2,)), (2, 2, (0, 0), (2, 2)), (3, 1, (0, 0, 0), (2, 2, 2)),
    (2, 3, (0, 1), (2, 3)), (2, 3, (1, 0), (2, 3)), (2, 3, (1, 1), (2, 3)),
])
def test_mass_solve_gives_one(dim, dof, periodic, degree):
    # test/IGACreate.c:103-125 and test/makefile:25-33: M x = int N  =>  x == 1
    g = O.OracleIGA(dim, dof)
    for i in range(dim):
        g.axis_uniform(i, degree[i], 16 if dim < 3 else 6, periodic=bool(periodic[i]))
    g.setup()
    A, b = g.compute_system("orc_form_mass")
    x = sla.spsolve(A.scipy().tocsc(), b)
    assert x.max() - x.min() < 1e-9 and abs(x.mean() - 1) < 1e-9
    vol = g.compute_scalar("orc_scalar_volume", 2)
    assert abs(vol[0] - 1.0) < 1e-13


# ---------------------------------------------------------------- order-3 tabulation and property arrays (round 6)
def test_third_derivatives_known_answers():
    """p->shape[3] (src/petigamapshf.f90.in:60-72 behind InverseMap order 3, src/petigamapinv.f90.in:49-60) through a form: the sum over
    the element's functions of d3 N_a vanishes at every point (partition of unity), so the c : d3N term of orc_form_der3 leaves the
    sum of F unchanged; and IGAPointFormDer3 of a field that is linear in x (U_a = x_a) is zero (test/IGAGeometryMap.c:221-255)."""
    import ctypes as C
    from common import make_pair, warped_geometry
    orc, _ = make_pair(3, 1, 3, [3, 2, 2], order=3, engine=False)
    X, W = warped_geometry(orc, 3, seed=4, rational=True)
    orc.set_geometry(X, W)
    prm = lambda *v: (C.c_double * 3)(*v)
    _, F0 = orc.compute_system("orc_form_der3", prm(0.0, 0.0, 0.0))
    _, F1 = orc.compute_system("orc_form_der3", prm(0.0, 1.0, 0.0))
    assert np.abs(F1 - F0).max() > 1e-3                       # the third derivatives are there ...
    assert abs(F1.sum() - F0.sum()) < TIGHT * np.abs(F1).sum()      # ... and sum to zero over the functions
    U = X[:, 0].copy()                                         # (one rank, no periodic axis: the net is the node grid)
    G0 = orc.compute_function("orc_form_der3_function", prm(0.0, 0.0, 0.0), U)
    G1 = orc.compute_function("orc_form_der3_function", prm(0.0, 0.0, 1.0), U)
    assert np.abs(G1 - G0).max() < 1e-9 * np.abs(G0).max()    # d3(x) = 0 (the warped net's third-order terms reach 1e2: relative to them 1e-12)
    V = np.random.default_rng(2).standard_normal(U.size)
    G2 = orc.compute_function("orc_form_der3_function", prm(0.0, 0.0, 1.0), V)
    assert np.abs(G2 - G0).max() > 1e-3


def test_property_array_known_answers():
    """p->property (src/petigaelem.c:745-752): constant properties k = 2, f = 3 turn orc_form_property into 2 K_Poisson and 3 F_Poisson."""
    from common import make_pair
    orc, _ = make_pair(2, 1, 2, [4, 3], engine=False)
    with pytest.raises(RuntimeError):
        orc.compute_system("orc_form_property")               # "No property set" (src/petigaelem.c:300)
    n = orc.global_size()
    A = np.tile([2.0, -1.0, 3.0], (n, 1))
    orc.set_property(A)
    K, F = orc.compute_system("orc_form_property")
    K0, F0 = orc.compute_system("orc_form_poisson")
    assert np.abs(K.val - 2.0 * K0.val).max() < TIGHT and np.abs(F - 3.0 * F0).max() < TIGHT


@pytest.mark.parametrize("dim", [1, 2])
def test_curve_and_surface_in_space_known_answers(dim):
    """IGASetGeometryDim with nsd != dim (src/petigaelem.c:940-1029: geometry map only, parametric shape functions and measure):
    a quarter circle of radius R as a NURBS curve in the plane (dim 1, nsd 2), extruded to a quarter cylinder (dim 2, nsd 3).
    orc_form_surface builds the metric from p->mapX[1], p->mapX[2]: sum K = length / area (partition of unity), sum F = that
    times the curvature 1 / R, which is the same at every point."""
    from common import make_pair
    R, h = 1.75, 0.8
    orc, _ = make_pair(dim, 1, [2, 1][:dim], [1, 2][:dim], nqp=[10, 2][:dim], engine=False)
    arc = [(R, 0.0), (R, R), (0.0, R)]; wts = [1.0, SQ2 / 2, 1.0]
    if dim == 1:
        X, W = np.array(arc), np.array(wts)
    else:
        X = np.array([[x, y, h * k / 2] for k in range(3) for (x, y) in arc]); W = np.array(wts * 3)
    orc.set_geometry(X, W)
    K, F = orc.compute_system("orc_form_surface")
    size = np.pi * R / 2 * (h if dim == 2 else 1.0)
    assert abs(K.val.sum() - size) < TIGHT and abs(F.sum() - size / R) < TIGHT
    e = orc.element([0] * dim)
    assert np.allclose(np.hypot(e["mapX0"][:, 0], e["mapX0"][:, 1]), R, atol=TIGHT)      # on the circle
    assert e["mapX1"].shape[1:] == (dim + 1, dim) and np.all(e["detX"] == 0)              # [nsd][dim]; no inverse map, no detX
