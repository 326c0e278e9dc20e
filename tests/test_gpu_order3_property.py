"""Order-3 tabulation and property arrays on the general kernel (round 6).
 * IGASetOrder(iga,3): p->shape[3] -- K2 at order 3 (src/petiga3d.F90:32-233), Rationalize order 3 (src/petigarat.f90.in), GeometryMap,
   InverseMap order 3 (src/petigamapinv.f90.in:49-60), ShapeFunctions order 3 (src/petigamapshf.f90.in:60-72) -- and IGAPointFormDer3
   (include/petiga.h:731), read by IGX_FORM_DER3 and by a struct given as source; the oracle holds the restated chain, pinned by
   test/IGAGeometryMap.c's identities (tests/test_oracle_known_answers.py).
 * IGASetPropertyDim / iga->propertyA (include/petiga.h:350-353, gathered per element at src/petigaelem.c:745-752, p->property).
Engine vs oracle: pattern bit-exact, values to 1e-11 of the largest free entry (third derivatives reach 1e3-1e5 on these meshes)."""
import ctypes as C

import numpy as np
import pytest

import petiga_amd as P
from common import compare_mats, iga_file_bytes, make_pair, warped_geometry

pytestmark = pytest.mark.gpu
PRM = (0.01, 0.5, 0.25)


def _prm(v): return (C.c_double * 3)(*v)


def _bc(orc, eng, dim):
    for g in (orc, eng):
        g.set_boundary_value(0, 0, 0, 0.5)
        if dim > 1:
            g.set_boundary_value(1, 1, 0, -1.0)
        g.set_boundary_load(dim - 1, 0, 0, 1.5)


@pytest.mark.parametrize("dim,p,N,geo,bc", [(1, 3, [6], None, True), (2, 3, [4, 3], None, False), (2, 2, [4, 5], "nurbs", True), (3, 3, [3, 2, 2], "nurbs", True),
                                            (3, 2, [3, 3, 2], "poly", False), (3, 3, [2, 3, 2], None, True), (2, 4, [3, 3], "nurbs", False)])
def test_third_order_system_vs_oracle(dim, p, N, geo, bc):
    orc, eng = make_pair(dim, 1, p, N, order=3)
    if geo:
        X, W = warped_geometry(orc, dim, seed=5, rational=geo == "nurbs")
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    if bc:
        _bc(orc, eng, dim)
    A_o, b_o = orc.compute_system("orc_form_der3", _prm(PRM))
    eng.set_form("der3", PRM)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "generic" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, 1e-11)
    assert np.abs(b.get() - b_o).max() <= 1e-11 * max(np.abs(b_o).max(), 1.0)
    # the third derivatives carry the result: without them K is the mass matrix
    A0, _ = orc.compute_system("orc_form_der3", _prm((0.0, 0.0, 0.0)))
    assert np.abs(A_o.val - A0.val).max() > 1e-3 * np.abs(A0.val).max()


@pytest.mark.parametrize("dim,p,N,geo", [(2, 3, [4, 3], "nurbs"), (3, 2, [3, 2, 3], "nurbs"), (3, 3, [2, 2, 3], None)])
def test_third_derivatives_of_the_state_vs_oracle(dim, p, N, geo):
    """IGAPointFormDer3 through the Function driver; a field linear in x has none (test/IGAGeometryMap.c:221-255 through the engine)."""
    orc, eng = make_pair(dim, 1, p, N, order=3)
    X = None
    if geo:
        X, W = warped_geometry(orc, dim, seed=6, rational=True)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    _bc(orc, eng, dim)
    rng = np.random.default_rng(9)
    U = rng.standard_normal(orc.global_size())
    F_o = orc.compute_function("orc_form_der3_function", _prm(PRM), U)
    eng.set_form("der3", PRM)
    Uv, F = eng.create_vec().set(U), eng.create_vec()
    eng.compute_function(Uv, F)
    eng.synchronize()
    assert np.abs(F.get() - F_o).max() <= 1e-11 * np.abs(F_o).max()
    if X is not None:
        eng.clear_boundary()
        Ux = eng.create_vec().set(X[:, 0].copy())
        F1, F0 = eng.create_vec(), eng.create_vec()
        eng.compute_function(Ux, F1)
        eng.set_form("der3", (PRM[0], PRM[1], 0.0))
        eng.compute_function(Ux, F0)
        assert np.abs(F1.get() - F0.get()).max() <= 1e-9 * np.abs(F0.get()).max()


def test_order_and_kernel_rules():
    _, eng = make_pair(2, 1, 2, [4, 4])                     # default order = max degree = 2 (src/petiga.c:1472-1475)
    eng.set_form("der3", PRM)
    A, b = eng.create_mat(), eng.create_vec()
    with pytest.raises(P.IGXError) as e:
        eng.compute_system(A, b)
    assert e.value.code == 73 and "IGASetOrder" in str(e.value)      # PETSC_ERR_ARG_WRONGSTATE
    eng.set_order(3)
    eng.compute_system(A, b)
    eng.set_kernel(3)
    with pytest.raises(P.IGXError) as e:
        eng.compute_system(A, b)
    assert e.value.code == 56                                         # PETSC_ERR_SUP: the general kernel only
    eng.set_kernel(0)
    eng.set_form("property")
    with pytest.raises(P.IGXError) as e:
        eng.compute_system(A, b)
    assert e.value.code == 73 and "No property set" in str(e.value)   # src/petigaelem.c:300


@pytest.mark.parametrize("dim,p,N,geo,npd", [(2, 2, [5, 4], None, 2), (3, 2, [3, 4, 3], "nurbs", 3), (3, 3, [3, 2, 2], "poly", 1), (1, 3, [7], None, 2)])
def test_property_array_system_vs_oracle(dim, p, N, geo, npd, tmp_path):
    orc, eng = make_pair(dim, 1, p, N)
    X = W = None
    if geo:
        X, W = warped_geometry(orc, dim, seed=8, rational=geo == "nurbs")
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    rng = np.random.default_rng(12)
    A = 1.0 + rng.random((orc.global_size(), npd))          # (positive conductivity; one rank, no periodic axis: the net is the node grid)
    orc.set_property(A); eng.set_property(A)
    assert eng.property_dim() == npd
    _bc(orc, eng, dim)
    K_o, F_o = orc.compute_system("orc_form_property")
    eng.set_form("property")
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    eng.synchronize()
    assert "generic" in eng.kernel_name(), eng.kernel_name()
    compare_mats(K, K_o, 1e-12)
    assert np.abs(F.get() - F_o).max() <= 1e-12 * max(np.abs(F_o).max(), 1.0)
    # ... and read from a file (IGALoad, src/petigaio.c:65-70): the same system
    U = [np.array(orc.axis(i)["U"]) for i in range(dim)]
    f = tmp_path / "prop.dat"; f.write_bytes(iga_file_bytes([p] * dim, U, X, W, A))
    eng2 = P.IGX(); eng2.set_dof(1); eng2.read(f); eng2.setup()
    _bc(orc, eng2, dim)
    eng2.set_form("property")
    K2, F2 = eng2.create_mat(), eng2.create_vec()
    eng2.compute_system(K2, F2)
    compare_mats(K2, K_o, 1e-12)


USER_DER3 = r"""
// a struct of ORDER 3 given as source: the third derivatives behind the second ones in Na, IGAPointFormDer3 in p.d3u; and one that
// reads the property array (NEED_PROP)
template <int DIM> struct UserDer3 {
  static constexpr int DOF = 1, ORDER = 3; static constexpr unsigned NEED = NEED_X | NEED_U | NEED_D3U;
  static constexpr int O3 = 1 + DIM + DIM * DIM;
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double s = 0; for (int f = 0; f < DIM * DIM * DIM; ++f) s += Na[O3 + f] * Nb[O3 + f];
    T[0] = Na[0] * Nb[0] + p.prm[0] * s;
  }
  static __device__ void vec(const PtView &p, const double *Na, double *R) {
    double x2 = 0, s = 0, su = 0;
    for (int i = 0; i < DIM; ++i) x2 += p.x[i] * p.x[i];
    for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) for (int k = 0; k < DIM; ++k) {
      const double c = 1.0 / (1.0 + i + 2.0 * j + 3.0 * k); s += c * Na[O3 + (i * DIM + j) * DIM + k]; su += c * p.d3u[(i * DIM + j) * DIM + k]; }
    R[0] = Na[0] * (1.0 + x2) + p.prm[1] * s + p.prm[2] * Na[0] * su;
  }
};
template <int DIM> struct UserProperty {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_PROP;
  static __device__ double at(const PtView &p, int c) { double s = 0; for (int a = 0; a < p.nen; ++a) s += p.shape[a * p.nf] * p.property[a * p.npd + c]; return s; }
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) { double s = 0; for (int i = 0; i < DIM; ++i) s += Na[1 + i] * Nb[1 + i]; T[0] = at(p, 0) * s; }
  static __device__ void vec(const PtView &p, const double *Na, double *R) { R[0] = Na[0] * at(p, p.npd - 1); }
};
"""


@pytest.mark.parametrize("kind", ["der3", "property"])
def test_user_source_of_order_three_and_with_properties(kind):
    dim, p, N = 3, 2, [3, 2, 3]
    orc, eng = make_pair(dim, 1, p, N, order=3)
    X, W = warped_geometry(orc, dim, seed=3, rational=True)
    orc.set_geometry(X, W); eng.set_geometry(X, W)
    _bc(orc, eng, dim)
    if kind == "der3":
        K_o, F_o = orc.compute_system("orc_form_der3", _prm(PRM))
        eng.set_form_source(USER_DER3, "UserDer3<3>", PRM)
    else:
        A = 1.0 + np.random.default_rng(1).random((orc.global_size(), 2))
        orc.set_property(A); eng.set_property(A)
        K_o, F_o = orc.compute_system("orc_form_property")
        eng.set_form_source(USER_DER3, "UserProperty<3>")
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    eng.synchronize()
    assert "generic_assemble" in eng.kernel_name() and "hiprtc" in eng.kernel_name(), eng.kernel_name()
    compare_mats(K, K_o, 1e-11)
    assert np.abs(F.get() - F_o).max() <= 1e-11 * max(np.abs(F_o).max(), 1.0)


# ---------------------------------------------------------------- a geometry of another dimension than the parametric one (nsd != dim)
def _lifted(orc, dim, nsd, seed, rational=True):
    """A curve / surface in space: the warped net of the parametric dimension with nsd - dim smooth coordinates more."""
    X, W = warped_geometry(orc, dim, seed=seed, rational=rational)
    cols = [X]
    for k in range(nsd - dim):
        z = 0.3 * np.sin(2.0 * X[:, 0] + k) + (0.2 * np.cos(3.0 * X[:, 1]) if dim > 1 else 0.1 * X[:, 0] ** 2)
        cols.append(z[:, None])
    return np.concatenate(cols, axis=1), W


@pytest.mark.parametrize("dim,nsd,p,N,rational,bc,gs", [(1, 2, 2, [6], True, True, 0.0), (1, 3, 3, [5], False, False, 0.3), (2, 3, 2, [4, 5], True, True, 0.2),
                                                        (2, 3, 3, [3, 4], True, False, 0.0), (2, 3, 2, [5, 3], False, True, 0.1)])
def test_curve_and_surface_system_vs_oracle(dim, nsd, p, N, rational, bc, gs, tmp_path):
    """IGASetGeometryDim(nsd) with nsd != dim (demo/ClassicalShell.c:154): IGA_GeometryMap (src/petigaval.F90:10-43) tabulated, no inverse
    map, parametric shape functions and measure, axis normals (src/petigaelem.c:940-1029); the form reads p->mapX[1], p->mapX[2] and
    IGAPointFormInvGradGeomMap's pseudo-inverse (src/petigaval.F90:124-142); the boundary load goes through BoundaryArea's geometry
    branch with nsd columns (src/petigaelem.c:1133-1160)."""
    orc, eng = make_pair(dim, 1, p, N)
    X, W = _lifted(orc, dim, nsd, seed=13, rational=rational)
    orc.set_geometry(X, W); eng.set_geometry(X, W)
    if bc:
        _bc(orc, eng, dim)
    K_o, F_o = orc.compute_system("orc_form_surface", C.c_double(gs))
    eng.set_form("surface", (gs,))
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    eng.synchronize()
    assert "generic" in eng.kernel_name(), eng.kernel_name()
    compare_mats(K, K_o, 1e-12)
    assert np.abs(F.get() - F_o).max() <= 1e-12 * max(np.abs(F_o).max(), 1.0)
    # any other form on such a geometry: the general kernel, parametric gradients, no detX -- what the reference computes
    A_o, b_o = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    eng.compute_system(K, F)
    assert "generic" in eng.kernel_name(), eng.kernel_name()
    compare_mats(K, A_o, 1e-12)
    assert np.abs(F.get() - b_o).max() <= 1e-12 * max(np.abs(b_o).max(), 1.0)
    eng.set_kernel(3)
    with pytest.raises(P.IGXError) as e:
        eng.compute_system(K, F)
    assert e.value.code == 56
    # ... and read from a file whose geometry block says nsd (IGALoad, src/petigaio.c:58-63)
    U = [np.array(orc.axis(i)["U"]) for i in range(dim)]
    f = tmp_path / "surf.dat"; f.write_bytes(iga_file_bytes([p] * dim, U, X, W))
    eng2 = P.IGX(); eng2.set_dof(1); eng2.read(f); eng2.setup()
    if bc:
        _bc(orc, eng2, dim)
    eng2.set_form("surface", (gs,))
    K2, F2 = eng2.create_mat(), eng2.create_vec()
    eng2.compute_system(K2, F2)
    compare_mats(K2, K_o, 1e-11)      # (the file holds x w: one rounding more)


def test_quarter_cylinder_known_answers_through_the_engine():
    R, h = 1.75, 0.8
    orc, eng = make_pair(2, 1, [2, 1], [1, 2], nqp=[10, 2])
    arc = [(R, 0.0), (R, R), (0.0, R)]; wts = [1.0, np.sqrt(0.5), 1.0]
    X = np.array([[x, y, h * k / 2] for k in range(3) for (x, y) in arc]); W = np.array(wts * 3)
    eng.set_geometry(X, W)
    eng.set_form("surface")
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    area = np.pi * R / 2 * h
    assert abs(K.to_coo_global()[2].sum() - area) < 1e-12 and abs(F.get().sum() - area / R) < 1e-12
    with pytest.raises(P.IGXError):
        eng.set_geometry(X[:, :1], W)      # nsd below dim


USER_SHELL = r"""
// a struct that reads the geometry map's derivatives like demo/ClassicalShell.c:57-80 (NEED_MAPX): the Lame parameters A1, A2 of the
// parametric lines and the normal curvatures b1, b2 there, on the parametric basis
struct UserShell {
  static constexpr int DOF = 1, ORDER = 2; static constexpr unsigned NEED = NEED_MAPX;
  static __device__ void lame(const PtView &p, double &A1, double &A2, double &b1, double &b2) {
    const double (*g)[2] = (const double (*)[2])p.X1; const double (*h)[2][2] = (const double (*)[2][2])p.X2;
    double n[3] = {g[1][0] * g[2][1] - g[2][0] * g[1][1], -(g[0][0] * g[2][1] - g[2][0] * g[0][1]), g[0][0] * g[1][1] - g[1][0] * g[0][1]};
    const double r = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); n[0] *= r; n[1] *= r; n[2] *= r;
    A1 = sqrt(g[0][0] * g[0][0] + g[1][0] * g[1][0] + g[2][0] * g[2][0]); A2 = sqrt(g[0][1] * g[0][1] + g[1][1] * g[1][1] + g[2][1] * g[2][1]);
    b1 = -1.0 / (A1 * A1) * (n[0] * h[0][0][0] + n[1] * h[1][0][0] + n[2] * h[2][0][0]);
    b2 = -1.0 / (A2 * A2) * (n[0] * h[0][1][1] + n[1] * h[1][1][1] + n[2] * h[2][1][1]);
  }
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double A1, A2, b1, b2; lame(p, A1, A2, b1, b2);
    T[0] = (Na[1] * Nb[1] / (A1 * A1) + Na[2] * Nb[2] / (A2 * A2) + (b1 * b1 + b2 * b2) * Na[0] * Nb[0]) * A1 * A2;
  }
  static __device__ void vec(const PtView &p, const double *Na, double *R) { double A1, A2, b1, b2; lame(p, A1, A2, b1, b2); R[0] = Na[0] * (b1 + b2) * A1 * A2; }
};
"""


def test_user_shell_struct_on_the_cylinder():
    """p->mapX[1], p->mapX[2] in a struct given as source, read as demo/ClassicalShell.c:57-80 does: on the quarter cylinder the
    parametric lines are lines of curvature, A1 A2 is the area element, b1 = 1/R (outward normal: -1/R ... the sign of the demo's
    formula), b2 = 0: sum F = -+ area / R, sum K = the (b1^2 + b2^2)-weighted area."""
    R, h = 1.75, 0.8
    orc, eng = make_pair(2, 1, [2, 1], [1, 2], nqp=[10, 2])
    arc = [(R, 0.0), (R, R), (0.0, R)]; wts = [1.0, np.sqrt(0.5), 1.0]
    X = np.array([[x, y, h * k / 2] for k in range(3) for (x, y) in arc]); W = np.array(wts * 3)
    eng.set_geometry(X, W)
    eng.set_form_source(USER_SHELL, "UserShell")
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    assert "generic_assemble" in eng.kernel_name()
    area = np.pi * R / 2 * h
    assert abs(abs(F.get().sum()) - area / R) < 1e-12 and abs(K.to_coo_global()[2].sum() - area / R ** 2) < 1e-12


# ---------------------------------------------------------------- the additions next to the path's other features
def _net_size(orc, dim):
    """control points of the geometry grid (n + 1 per axis: a periodic axis keeps its wrapped copies, src/petigaio.c:187-199)"""
    n = 1
    for i in range(dim):
        ax = orc.axis(i)
        n *= len(ax["U"]) - 1 - ax["p"]
    return n


def test_property_array_on_a_periodic_axis_and_on_a_visited_face():
    """The property array lives on the geometry grid (the wrapped copies of a periodic axis included); a boundary-form pass sees the
    face element's nodal values like the interior pass does (src/petigaelem.c:745-752 is the closure of every element)."""
    orc, eng = make_pair(2, 1, 2, [6, 5], periodic=[True, False])
    A = 1.0 + np.random.default_rng(3).random((_net_size(orc, 2), 2))
    orc.set_property(A); eng.set_property(A)
    for g in (orc, eng):
        g.set_boundary_value(1, 0, 0, 0.5)
        g.set_boundary_form(1, 1, True)
    K_o, F_o = orc.compute_system("orc_form_property")
    eng.set_form("property")
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    compare_mats(K, K_o, 1e-12)
    assert np.abs(F.get() - F_o).max() <= 1e-12 * max(np.abs(F_o).max(), 1.0)


def test_third_order_form_over_a_visited_face():
    orc, eng = make_pair(2, 1, 3, [4, 3], order=3)
    X, W = warped_geometry(orc, 2, seed=8, rational=True)
    orc.set_geometry(X, W); eng.set_geometry(X, W)
    for g in (orc, eng):
        g.set_boundary_form(0, 1, True); g.set_boundary_form(1, 0, True)
    K_o, F_o = orc.compute_system("orc_form_der3", _prm(PRM))
    eng.set_form("der3", PRM)
    K, F = eng.create_mat(), eng.create_vec()
    eng.compute_system(K, F)
    compare_mats(K, K_o, 1e-11)
    assert np.abs(F.get() - F_o).max() <= 1e-11 * max(np.abs(F_o).max(), 1.0)


@pytest.mark.parametrize("size,dim,nsd,p,N,what", [(2, 2, 3, 2, (6, 5), "surface"), (3, 1, 2, 3, (9,), "surface"), (4, 3, 3, 2, (6, 5, 4), "reduced"), (2, 2, 2, 3, (7, 4), "der3")])
def test_round6_additions_on_a_partition(size, dim, nsd, p, N, what):
    """Every rank assembles its box (the ghosted box of the net for a surface in space; the GLOBAL first and last element of an axis keep
    the full rule under IGA_RULE_REDUCED, src/petigabasis.c:163-170; third-order tabulation per rank); owned rows of all ranks after
    the ghost-row exchange = the single-rank oracle."""
    import torch
    import scipy.sparse as sp
    from test_gpu_parity import _rank_matrix_rows
    orc, _ = make_pair(dim, 1, p, list(N), order=3 if what == "der3" else None, engine=False)
    if what == "reduced":
        for i in range(dim):
            orc.set_rule_type(i, "reduced")
        orc.setup()
    X = W = None
    if what != "reduced":
        X, W = _lifted(orc, dim, nsd, seed=17) if nsd != dim else warped_geometry(orc, dim, seed=17, rational=True)
        orc.set_geometry(X, W)
    form, octx, prm = {"surface": ("surface", C.c_double(0.2), (0.2,)), "reduced": ("poisson", None, ()), "der3": ("der3", _prm(PRM), PRM)}[what]
    orc.set_boundary_value(0, 0, 0, 0.5)
    A_o, b_o = orc.compute_system("orc_form_" + form, octx)
    engs, mats, vecs, sendbufs = [], [], [], {}
    for r in range(size):
        g = P.IGX(dim, 1)
        for i in range(dim):
            g.axis_uniform(i, p, N[i])
            if what == "reduced":
                g.set_rule_type(i, "reduced")
        if what == "der3":
            g.set_order(3)
        g.set_comm(size, r)
        g.setup()
        if X is not None:
            g.set_geometry(X, W)
        g.set_boundary_value(0, 0, 0, 0.5)
        g.set_form(form, prm)
        A, b = g.create_mat(), g.create_vec()
        g.compute_system(A, b)
        for k, (peer, m, v) in enumerate(g.neighbors(True)):
            buf = torch.empty(m + v, dtype=torch.float64, device="cuda")
            g.pack_ghost_rows(A, b, k, buf.data_ptr())
            sendbufs[(r, peer)] = buf
        g.synchronize()
        engs.append(g); mats.append(A); vecs.append(b)
    for r, g in enumerate(engs):
        for k, (peer, m, v) in enumerate(g.neighbors(False)):
            g.unpack_ghost_rows(mats[r], vecs[r], k, sendbufs[(peer, r)].data_ptr())
        g.synchronize()
    n = orc.global_size()
    M = sp.csr_matrix((n, n)); F = np.zeros(n); seen = np.zeros(n, dtype=int)
    for r, g in enumerate(engs):
        rows, cols, vals, own_e, own = _rank_matrix_rows(g, mats[r], vecs[r])
        M = M + sp.coo_matrix((vals[own_e], (rows[own_e], cols[own_e])), shape=(n, n)).tocsr()
        nrow, _, maps = mats[r].layout()
        ns = g.sizes()["node_sizes"]
        rr = np.arange(mats[r].nbrows)
        grow = maps[0][0][rr % nrow[0]].astype(np.int64) + ns[0] * (maps[1][0][(rr // nrow[0]) % nrow[1]].astype(np.int64) + ns[1] * maps[2][0][rr // (nrow[0] * nrow[1])].astype(np.int64))
        bv = vecs[r].get()
        F[grow[own]] = bv[own]; seen[grow[own]] += 1
    assert np.all(seen == 1)
    Mo = A_o.scipy()
    assert abs(M - Mo).max() <= 1e-11 * abs(Mo).max()
    assert np.abs(F - b_o).max() <= 1e-11 * max(np.abs(b_o).max(), 1e-300)
