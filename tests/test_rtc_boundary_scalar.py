"""Run-time forms to parity with the built-in ones: boundary-form passes (IGAElementNextForm,
src/petigaelem.c:427-447; `if (p->atboundary)` in the callback) and user functionals (IGAComputeScalar's `Scalar` argument,
src/petigacomp.c:35-98) for structs given as HIP source.  demo/NitscheMethod.c and demo/BoundaryIntegral.c as source against
the oracle's restatement of the same callbacks, on both device kernels; the functionals of test/IGAFixTable.c (ErrorSqr) and
test/IGAGeometryMap.c (volume / area) as source against the oracle and the built-in kinds; the vector-only drivers of a user struct
on the sum-factorised kernel (vec_sumfact.hpp compiled for the struct)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, rel_err, warped_geometry

NITSCHE = r"""
// demo/NitscheMethod.c:69-110 as user source: Poisson inside (f = -2 dim), u = sum x_i^2 imposed weakly on the visited faces;
// params = {k = max degree}
template <int DIM> struct UserNitsche {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_X | NEED_G;
  static constexpr bool HAS_BOUNDARY = true;
  static __device__ void mat(const PtView &, const double *Na, const double *Nb, double *T) {
    double s = 0; for (int i = 0; i < DIM; ++i) s += Na[1 + i] * Nb[1 + i];
    T[0] = s;
  }
  static __device__ void vec(const PtView &, const double *Na, double *R) { R[0] = Na[0] * (-2.0 * DIM); }
  static __device__ double alpha(const PtView &p) {      // C / h with h = NormalMeshSize (:57-66)
    double s = 0;
    for (int i = 0; i < DIM; ++i) { double Ni = 0; for (int j = 0; j < DIM; ++j) Ni += p.G[i * DIM + j] * p.normal[j]; s += Ni * Ni; }
    return 5 * (p.prm[0] + 1) / (2 / sqrt(s));
  }
  static __device__ void bmat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double dna = 0, dnb = 0;
    for (int i = 0; i < DIM; ++i) { dna += Na[1 + i] * p.normal[i]; dnb += Nb[1 + i] * p.normal[i]; }
    T[0] = -Na[0] * dnb - Nb[0] * dna + alpha(p) * Na[0] * Nb[0];
  }
  static __device__ void bvec(const PtView &p, const double *Na, double *R) {
    double g = 0, dna = 0;
    for (int i = 0; i < DIM; ++i) { g += p.x[i] * p.x[i]; dna += Na[1 + i] * p.normal[i]; }
    R[0] = -dna * g + alpha(p) * Na[0] * g;
  }
};
"""

BOUNDARY_INTEGRAL = r"""
// demo/BoundaryIntegral.c:26-56 as user source: Laplace inside (F = 0), Neumann data 1 on the visited faces
template <int DIM> struct UserBoundaryIntegral {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr bool HAS_BOUNDARY = true;
  static __device__ void mat(const PtView &, const double *Na, const double *Nb, double *T) { double s = 0; for (int i = 0; i < DIM; ++i) s += Na[1 + i] * Nb[1 + i]; T[0] = s; }
  static __device__ void vec(const PtView &, const double *, double *R) { R[0] = 0.0; }
  static __device__ void bmat(const PtView &, const double *, const double *, double *T) { T[0] = 0.0; }
  static __device__ void bvec(const PtView &, const double *Na, double *R) { R[0] = Na[0]; }
};
"""

PLAIN_MASS = r"""
// no atboundary branch: the ordinary integrand is integrated over the visited faces too (test/IGACreate.c:45-63 System)
struct UserMass {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static __device__ void mat(const PtView &, const double *Na, const double *Nb, double *T) { T[0] = Na[0] * Nb[0]; }
  static __device__ void vec(const PtView &, const double *Na, double *R) { R[0] = Na[0]; }
};
"""

FUNCTIONALS = r"""
// test/IGAFixTable.c:66 + src/petigacomp.c:102 (ErrorSqr): |sum x_i^2 - u|^2
template <int DIM> struct UserX2Err {
  static constexpr int DOF = 1, ORDER = 1, NSCALAR = 1; static constexpr unsigned NEED = NEED_X | NEED_U;
  static __device__ void scalar(const PtView &p, double *S) { double g = 0; for (int i = 0; i < DIM; ++i) g += p.x[i] * p.x[i]; const double e = fabs(g - p.u[0]); S[0] = e * e; }
};
// test/IGAGeometryMap.c:383-389: volume inside, area on the visited faces; a third number with the state and a parameter:
// int (prm0 u^2 + |grad u|^2) over the domain
template <int DIM> struct UserMeasures {
  static constexpr int DOF = 1, ORDER = 1, NSCALAR = 3; static constexpr unsigned NEED = NEED_U | NEED_GU;
  static __device__ void scalar(const PtView &p, double *S) {
    S[0] = p.atboundary ? 0.0 : 1.0; S[1] = p.atboundary ? 1.0 : 0.0;
    double g2 = 0; for (int i = 0; i < DIM; ++i) g2 += p.gu[i] * p.gu[i];
    S[2] = p.atboundary ? 0.0 : p.prm[0] * p.u[0] * p.u[0] + g2;
  }
};
"""


@pytest.mark.parametrize("dim", [2, 3])
def test_boundary_structs_compile_without_a_gpu(dim):
    """IGXSetFormSource + IGXCheckFormSource: the point-form and the matrix-core instantiations with the bmat / bvec branch"""
    import petiga_amd as P
    g = P.IGX(dim, 1)
    for i in range(dim):
        g.axis_uniform(i, 2, 4)
    g.set_form_source(NITSCHE, "UserNitsche<%d>" % dim, (2.0,))
    g.check_form_source(True, False)
    g.check_form_source(False, False)


def _faces(objs, dim, faces=None):
    for g in objs:
        for a in range(dim):
            for s in range(2):
                if faces is None or (a, s) in faces:
                    g.set_boundary_form(a, s, True)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("dim,p,N,geo", [(1, 2, 8, None), (2, 2, 9, None), (2, 3, 5, "nurbs"), (3, 2, 4, None), (3, 3, 3, "nurbs"), (3, 4, 2, "poly")])
def test_nitsche_as_source_matches_oracle(dim, p, N, geo, kernel):
    orc, eng = make_pair(dim, 1, p, N)
    eng.set_kernel(kernel)
    if geo:
        X, W = warped_geometry(orc, dim, seed=4, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    _faces((orc, eng), dim)
    Ao, bo = orc.compute_system("orc_form_nitsche", C.c_int(p))
    eng.set_form_source(NITSCHE, "UserNitsche<%d>" % dim, (float(p),))
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "hiprtc" in eng.kernel_name() and (("generic_assemble" in eng.kernel_name()) == (kernel == 1 or dim == 1))
    compare_mats(A, Ao, 1e-11)
    assert rel_err(b.get(), bo) < 1e-11
    # the drivers that apply no Dirichlet fix-up make the same passes
    eng.compute_matrix(A); eng.compute_vector(b); eng.synchronize()
    compare_mats(A, Ao, 1e-11)
    assert rel_err(b.get(), bo) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("dim,axis,side,geo", [(2, 0, 1, None), (2, 1, 0, "nurbs"), (3, 2, 1, None), (3, 0, 0, "poly")])
def test_boundary_integral_as_source_matches_oracle(dim, axis, side, geo, kernel):
    orc, eng = make_pair(dim, 1, 2, 6 if dim == 2 else 4)
    eng.set_kernel(kernel)
    if geo:
        X, W = warped_geometry(orc, dim, seed=9, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    for g in (orc, eng):
        g.set_boundary_value(axis, 1 - side, 0, 1.0)      # demo/BoundaryIntegral.c:172-176
        g.set_boundary_form(axis, side, True)
    Ao, bo = orc.compute_system("orc_form_boundary_integral")
    eng.set_form_source(BOUNDARY_INTEGRAL, "UserBoundaryIntegral<%d>" % dim)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    compare_mats(A, Ao, 1e-12)
    assert rel_err(b.get(), bo) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
def test_struct_without_a_boundary_branch_is_integrated_over_the_face(kernel):
    """src/petigaelem.c:427-447: the same callback is called on the face; only `atboundary` tells it"""
    orc, eng = make_pair(3, 1, 2, [3, 4, 3])
    eng.set_kernel(kernel)
    X, W = warped_geometry(orc, 3, seed=2, rational=True, amp=0.08)
    orc.set_geometry(X, W); eng.set_geometry(X, W)
    _faces((orc, eng), 3, {(0, 1), (2, 0)})
    Ao, bo = orc.compute_system("orc_form_mass")
    eng.set_form_source(PLAIN_MASS, "UserMass")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    compare_mats(A, Ao, 1e-12)
    assert rel_err(b.get(), bo) < 1e-12


@pytest.mark.gpu
def test_pencil_walk_declines_a_visited_face():
    """a run-time form that would take the pencil walk stays on the element kernels when a face is visited"""
    from test_rtc_forms import USER_POISSON
    orc, eng = make_pair(3, 1, 2, [8, 3, 3])
    _faces((orc, eng), 3, {(1, 0)})
    Ao, bo = orc.compute_system("orc_form_poisson")
    eng.set_form_source(USER_POISSON, "UserPoisson", (1.0,))
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "form_pencil" not in eng.kernel_name()
    compare_mats(A, Ao, 1e-12)
    assert rel_err(b.get(), bo) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dim,p,geo", [(1, 2, None), (2, 2, "nurbs"), (3, 3, "poly"), (3, 2, "nurbs")])
def test_user_functionals_match_oracle_and_builtin(dim, p, geo):
    orc, eng = make_pair(dim, 1, p, 5 if dim < 3 else 3)
    if geo:
        X, W = warped_geometry(orc, dim, seed=6, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    rng = np.random.default_rng(dim)
    U = rng.standard_normal(orc.global_size())
    Uv = eng.create_vec().set(U)
    s_o = orc.compute_scalar("orc_scalar_x2err", 1, U=U)
    s_b = eng.compute_scalar("x2err", Uv)
    s_u = eng.compute_scalar_source(FUNCTIONALS, "UserX2Err<%d>" % dim, 1, Uv)
    assert abs(s_u[0] - s_o[0]) <= 1e-12 * abs(s_o[0]) and abs(s_u[0] - s_b[0]) <= 1e-13 * abs(s_b[0])
    # volume / area with two visited faces, and a functional of u and grad u with a parameter
    faces = {(0, 1)} if dim == 1 else {(0, 1), (dim - 1, 0)}
    _faces((orc, eng), dim, faces)
    v_o = orc.compute_scalar("orc_scalar_volume", 2, full=True)
    v_b = eng.compute_scalar("volume")
    m = eng.compute_scalar_source(FUNCTIONALS, "UserMeasures<%d>" % dim, 3, Uv, (0.75,))
    assert np.abs(m[:2] - v_o).max() <= 1e-12 * np.abs(v_o).max() and np.abs(m[:2] - v_b).max() <= 1e-13 * np.abs(v_b).max()
    # int (0.75 u^2 + |grad u|^2) = U^T (0.75 M + K) U with the oracle's mass and stiffness matrices (no faces: interior only)
    orc.clear_boundary()
    M, _ = orc.compute_system("orc_form_mass")
    K, _ = orc.compute_system("orc_form_poisson")
    ref = 0.75 * (U @ (M.scipy() @ U)) + U @ (K.scipy() @ U)
    assert abs(m[2] - ref) <= 1e-11 * abs(ref)
    # the same functional twice: same bits; a functional struct is refused as a form and a form struct as a functional
    assert np.array_equal(m, eng.compute_scalar_source(FUNCTIONALS, "UserMeasures<%d>" % dim, 3, Uv, (0.75,)))
    import petiga_amd as P
    with pytest.raises(P.IGXError):
        eng.compute_scalar_source(PLAIN_MASS, "UserMass", 1, Uv)
    with pytest.raises(P.IGXError):
        eng.compute_scalar_source(FUNCTIONALS, "UserMeasures<%d>" % dim, 2, Uv, (0.75,))


BRATU3 = r"""
// demo/BratuFJ.F90:23-176 as user source in three dimensions; params = {lambda}
struct UserBratu3 {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_U | NEED_UT | NEED_GU;
  static __device__ void vec(const PtView &p, const double *Na, double *R) {
    double s = 0; for (int i = 0; i < 3; ++i) s += Na[1 + i] * p.gu[i];
    R[0] = Na[0] * p.ut[0] + s - Na[0] * p.prm[0] * exp(p.u[0]);
  }
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double s = 0; for (int i = 0; i < 3; ++i) s += Na[1 + i] * Nb[1 + i];
    T[0] = p.shift * Na[0] * Nb[0] + s - Na[0] * Nb[0] * p.prm[0] * exp(p.u[0]);
  }
};
"""


def test_vector_kernel_of_a_user_struct_compiles_without_a_gpu():
    """IGXCheckFormSource(gram = 3): vec_sumfact<UserStruct, GEO> for the current geometry kind"""
    import petiga_amd as P
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, 2, 4)
    g.set_form_source(BRATU3, "UserBratu3", (3.5,))
    g.check_form_source(False, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("p,N,geo", [(2, (5, 4, 6), None), (3, (3, 4, 3), "nurbs"), ((1, 2, 3), (4, 3, 3), "poly")])
def test_user_struct_residual_on_the_sum_factorised_kernel(p, N, geo, monkeypatch):
    orc, eng = make_pair(3, 1, list(p) if isinstance(p, tuple) else p, list(N))
    if geo:
        X, W = warped_geometry(orc, 3, seed=12, rational=(geo == "nurbs"), amp=0.08)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    for g in (orc, eng):
        for d in range(3):
            g.set_boundary_value(d, 1, 0, 0.1 * d)
    lam = C.c_double(3.5)
    rng = np.random.default_rng(8)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    eng.set_form_source(BRATU3, "UserBratu3", (3.5,))
    Uv, Vv, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec()
    eng.compute_function(Uv, F); eng.synchronize()
    assert "vec_sumfact<UserBratu3>(hiprtc" in eng.kernel_name(), eng.kernel_name()
    assert rel_err(F.get(), orc.compute_function("orc_form_bratu_function", lam, U)) < 1e-12
    eng.compute_ifunction(4.0, Vv, 0.0, Uv, F); eng.synchronize()
    assert "vec_sumfact" in eng.kernel_name()
    F1 = F.get().copy()
    assert rel_err(F1, orc.compute_ifunction("orc_form_bratu_ifunction", lam, 4.0, V, 0.0, U)) < 1e-12
    # the element kernels give the same numbers to rounding; the Jacobian of the same struct stays on them
    eng.set_kernel(3)
    eng.compute_ifunction(4.0, Vv, 0.0, Uv, F); eng.synchronize()
    assert "feature_assemble" in eng.kernel_name() and rel_err(F.get(), F1) < 1e-12
    eng.set_kernel(0)
    J = eng.create_mat()
    eng.compute_ijacobian(4.0, Vv, 0.0, Uv, J); eng.synchronize()
    compare_mats(J, orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, 4.0, V, 0.0, U), 1e-12)


BRATU3_WALK = BRATU3.replace("struct UserBratu3 {", """struct UserBratu3Walk {
  // the Jacobian on the pencil walk: A = (N, grad N), B = (JW (shift - lambda e^u) N, JW grad N)
  static constexpr int PENCIL_NFEAT = 4, PENCIL_NC = 2;
  static __device__ void pencil_coef(const PtView &p, double JW, double *c) { c[0] = JW * (p.shift - p.prm[0] * exp(p.u[0])); c[1] = JW; }
  static __device__ void pencil_trial(const double *c, double N, const double *g, double, double *B) { B[0] = c[0] * N; for (int i = 0; i < 3; ++i) B[1 + i] = c[1] * g[i]; }""")


@pytest.mark.parametrize("p", [2, 3])
def test_state_pencil_of_a_user_struct_compiles_without_a_gpu(p):
    """IGXCheckFormSource(gram = 4): state_pencil<p, UserStruct>"""
    import petiga_amd as P
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, 8)
    g.set_form_source(BRATU3_WALK, "UserBratu3Walk", (3.5,))
    g.check_form_source(True, 4)
    g.set_form_source(BRATU3, "UserBratu3", (3.5,))      # no hooks: the instantiation does not compile
    with pytest.raises(P.IGXError):
        g.check_form_source(True, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("p,N,periodic,driver", [(2, (9, 4, 5), (False, False, False), "jacobian"), (3, (8, 4, 4), (False, True, False), "ijacobian")])
def test_user_struct_tangent_on_the_pencil_walk(p, N, periodic, driver):
    orc, eng = make_pair(3, 1, p, list(N), periodic=list(periodic))
    for g in (orc, eng):
        for d in range(3):
            if not periodic[d]:
                g.set_boundary_value(d, 0, 0, 0.2 * d)
    lam = C.c_double(3.5)
    rng = np.random.default_rng(8)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    eng.set_form_source(BRATU3_WALK, "UserBratu3Walk", (3.5,))
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    if driver == "jacobian":
        eng.compute_jacobian(Uv, J)
        J_o = orc.compute_jacobian("orc_form_bratu_jacobian", lam, U)
    else:
        eng.compute_ijacobian(4.0, Vv, 0.0, Uv, J)
        J_o = orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, 4.0, V, 0.0, U)
    eng.synchronize()
    assert "state_pencil<UserBratu3Walk,hiprtc>" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-12)
    # without the hooks the same Jacobian comes from the element kernel
    eng.set_form_source(BRATU3, "UserBratu3", (3.5,))
    if driver == "jacobian":
        eng.compute_jacobian(Uv, J)
    else:
        eng.compute_ijacobian(4.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "feature_assemble<UserBratu3>" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-12)


def test_state_pencil_of_a_user_struct_on_a_mapped_geometry_compiles_without_a_gpu():
    """IGXCheckFormSource(gram = 4) with a geometry set at p = 2: state_pencil_geo<2, RAT, UserStruct>"""
    import petiga_amd as P
    from common import warped_geometry
    for rational in (False, True):
        orc, g = make_pair(3, 1, 2, [6, 5, 4])
        X, W = warped_geometry(orc, 3, seed=2, rational=rational, amp=0.05)
        g.set_geometry(X, W)
        g.set_form_source(BRATU3_WALK, "UserBratu3Walk", (3.5,))
        g.check_form_source(True, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("rational", [False, True])
def test_user_struct_tangent_on_the_pencil_walk_on_a_mapped_geometry(rational):
    from common import warped_geometry
    orc, eng = make_pair(3, 1, 2, [9, 4, 5])
    X, W = warped_geometry(orc, 3, seed=9, rational=rational, amp=0.06)
    orc.set_geometry(X, W); eng.set_geometry(X, W)
    for g in (orc, eng):
        for d in range(3):
            g.set_boundary_value(d, 0, 0, 0.2 * d)
    lam = C.c_double(3.5)
    rng = np.random.default_rng(8)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    eng.set_form_source(BRATU3_WALK, "UserBratu3Walk", (3.5,))
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    eng.compute_ijacobian(4.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "state_pencil<UserBratu3Walk,hiprtc>" in eng.kernel_name() and "mapped geometry" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, 4.0, V, 0.0, U), 1e-11)


USER_CH3 = r"""
// demo/CahnHilliard3D.c:39-179 as user source; params = {theta, alpha, cbar, L0, lambda, tau}.  Second-order test features (the
// Laplacian of N) and the hooks of the pencil walk.
struct UserCH3 {
  static constexpr unsigned MAT_TEST_MASK = 0xFu | (1u << 4) | (1u << 8) | (1u << 12);      // N, grad N, the diagonal of hess N
  static constexpr unsigned PHI_MASK = MAT_TEST_MASK, VEC_TEST_MASK = MAT_TEST_MASK;
  static constexpr int DOF = 1, ORDER = 2; static constexpr unsigned NEED = NEED_U | NEED_UT | NEED_GU | NEED_HU;
  struct K { double M, dM, d2M, dmu, d2mu, lap, t1; };
  static __device__ K coef(const PtView &p) {
    K k; const double c = p.u[0], theta = p.prm[0], L0 = p.prm[3], lambda = p.prm[4];
    const double scale = L0 * L0 / lambda;
    k.M = c * (1 - c); k.dM = 1 - 2 * c; k.d2M = -2;
    k.dmu = (0.5 / theta / (c * (1 - c)) - 2) * scale;
    k.d2mu = (-0.5 / theta * (1 - 2 * c) / (c * c * (1 - c) * (1 - c))) * scale;
    k.lap = p.hu[0] + p.hu[4] + p.hu[8];
    k.t1 = k.M * k.dmu + k.dM * k.lap;
    return k;
  }
  static __device__ double lapN(const double *N) { return N[4] + N[8] + N[12]; }
  static __device__ void vec(const PtView &p, const double *Na, double *R) {
    const K k = coef(p);
    double Ra = Na[0] * p.ut[0];
    for (int i = 0; i < 3; ++i) Ra += Na[1 + i] * k.t1 * p.gu[i];
    R[0] = Ra + lapN(Na) * k.M * k.lap;
  }
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const K k = coef(p);
    const double lapNa = lapN(Na), lapNb = lapN(Nb);
    double Kab = p.shift * Na[0] * Nb[0];
    for (int i = 0; i < 3; ++i) Kab += Na[1 + i] * k.t1 * Nb[1 + i];
    const double t2 = (k.dM * k.dmu + k.M * k.d2mu + k.d2M * k.lap) * Nb[0] + k.dM * lapNb;
    for (int i = 0; i < 3; ++i) Kab += Na[1 + i] * t2 * p.gu[i];
    T[0] = Kab + lapNa * (k.dM * k.lap * Nb[0] + k.M * lapNb);
  }
  static constexpr int PENCIL_NFEAT = 5, PENCIL_NC = 9;
  static __device__ void pencil_coef(const PtView &p, double JW, double *c) {
    const K k = coef(p);
    c[0] = JW * p.shift; c[1] = JW * k.t1; c[2] = JW * (k.dM * k.dmu + k.M * k.d2mu + k.d2M * k.lap); c[3] = JW * k.dM;
    c[4] = JW * (k.dM * k.lap); c[5] = JW * k.M;
    for (int i = 0; i < 3; ++i) c[6 + i] = p.gu[i];
  }
  static __device__ void pencil_trial(const double *c, double N, const double *g, double lap, double *B) {
    const double h = c[2] * N + c[3] * lap;
    B[0] = c[0] * N;
    for (int i = 0; i < 3; ++i) B[1 + i] = c[1] * g[i] + c[6 + i] * h;
    B[4] = c[4] * N + c[5] * lap;
  }
};
"""


@pytest.mark.gpu
@pytest.mark.parametrize("geo", [None, "poly", "nurbs"])
def test_user_cahn_hilliard_struct_on_a_mapped_geometry(geo):
    """A second-order struct given as source: its IFunction on vec_sumfact and its IJacobian on the state walk, on the identity
    geometry, a polynomial map and a NURBS map (the physical Laplacian of the test functions needs the map's second derivatives)"""
    CH = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)
    orc, eng = make_pair(3, 1, 2, [9, 4, 5])
    if geo:
        X, W = warped_geometry(orc, 3, seed=21, rational=(geo == "nurbs"), amp=0.06)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    ctx = O.CahnHilliardCtx(*CH)
    rng = np.random.default_rng(5)
    n = orc.global_size()
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    eng.set_form_source(USER_CH3, "UserCH3", CH)
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_ifunction(250.0, Vv, 0.0, Uv, F); eng.synchronize()
    assert "vec_sumfact<UserCH3>(hiprtc" in eng.kernel_name(), eng.kernel_name()
    assert rel_err(F.get(), orc.compute_ifunction("orc_form_ch_residual", ctx, 250.0, V, 0.0, U)) < 1e-10
    eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J); eng.synchronize()
    assert "state_pencil<UserCH3,hiprtc>" in eng.kernel_name() and (("mapped geometry" in eng.kernel_name()) == bool(geo)), eng.kernel_name()
    compare_mats(J, orc.compute_ijacobian("orc_form_ch_tangent", ctx, 250.0, V, 0.0, U), 1e-10)


ELASTICITY_BANDS = r"""
// demo/Elasticity3D.c:13-46 as user source (with its :37 quirk); params = {lambda, mu}.  The declarations say what the callback's
// shape is -- point-independent coefficients on gradient pairs (MAT_PAIR_MASK), which block entries a pair reaches
// (pair_block_mask), F = 0 (VEC_ZERO) -- and the struct takes the band-row kernel (block_pencil.hpp) like the built-in form.
struct UserElasticityBands {
  static constexpr int DOF = 3, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 0xEu;
  static constexpr unsigned long long MAT_PAIR_MASK = (0xEull << 8) | (0xEull << 16) | (0xEull << 24);
  static constexpr bool VEC_ZERO = true;
  static constexpr unsigned pair_block_mask(int f, int g) { return f == g ? 0x111u : ((1u << ((f - 1) * 3 + (g - 1))) | (1u << ((g - 1) * 3 + (f - 1)))); }
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const double l = p.prm[0], m = p.prm[1];
    const double ax = Na[1], ay = Na[2], az = Na[3], bx = Nb[1], by = Nb[2], bz = Nb[3];
    T[0] = ax * bx * (l + 2 * m) + m * (ay * by + az * bz); T[1] = ax * by * l + ay * bx * m; T[2] = ax * bz * l + az * bx * m;
    T[3] = ax * by * m + ay * bx * l; T[4] = ay * by * (l + 2 * m) + m * (az * bz + ax * bx * m); T[5] = ay * bz * l + az * by * m;
    T[6] = ax * bz * m + az * bx * l; T[7] = ay * bz * m + az * by * l; T[8] = m * (ax * bx + ay * by) + az * bz * (l + 2 * m);
  }
  static __device__ void vec(const PtView &, const double *, double *R) { R[0] = 0; R[1] = 0; R[2] = 0; }
};
"""


def test_band_row_kernel_of_a_user_struct_compiles_without_a_gpu():
    """IGXCheckFormSource(gram = 5): block_pencil<UserStruct, 3, SYSTEM> for the System and the Matrix driver"""
    import petiga_amd as P
    g = P.IGX(3, 3)
    for i in range(3):
        g.axis_uniform(i, 3, 8)
    g.set_form_source(ELASTICITY_BANDS, "UserElasticityBands", (1.5, 0.8))
    g.check_form_source(True, 5)


@pytest.mark.gpu
@pytest.mark.parametrize("N,bc,driver", [((9, 4, 4), "clamped", "system"), ((12, 3, 5), "mixed", "system"), ((8, 4, 3), "none", "matrix")])
def test_user_elasticity_on_the_band_row_kernel(N, bc, driver):
    orc, eng = make_pair(3, 3, 3, list(N))
    for g in (orc, eng):
        if bc == "clamped":       # demo/Elasticity3D.c:100-108: one face clamped, u_x = 1 on the opposite one
            for f in range(3):
                g.set_boundary_value(0, 0, f, 0.0)
            g.set_boundary_value(0, 1, 0, 1.0)
        elif bc == "mixed":
            g.set_boundary_value(1, 0, 2, 0.3); g.set_boundary_value(2, 1, 0, -0.2); g.set_boundary_load(0, 1, 1, 0.7)
    ctx = O.ElasticityCtx(1.5, 0.8)
    eng.set_form_source(ELASTICITY_BANDS, "UserElasticityBands", (1.5, 0.8))
    A, b = eng.create_mat(), eng.create_vec()
    if driver == "system":
        Ao, bo = orc.compute_system("orc_form_elasticity", ctx)
        eng.compute_system(A, b)
    else:
        orc.clear_boundary()
        Ao, bo = orc.compute_system("orc_form_elasticity", ctx)
        eng.compute_matrix(A)
    eng.synchronize()
    assert "block_pencil<UserElasticityBands>(hiprtc" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, Ao, 1e-12)
    if driver == "system":
        assert rel_err(b.get(), bo) < 1e-12
    # IGX_KERNEL-style choice 3 keeps the element kernel: same numbers to rounding
    eng.set_kernel(3)
    if driver == "system":
        eng.compute_system(A, b)
    else:
        eng.compute_matrix(A)
    eng.synchronize()
    assert "feature_assemble<UserElasticityBands>" in eng.kernel_name()
    compare_mats(A, Ao, 1e-12)
