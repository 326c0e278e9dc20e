"""Run-time compiled user forms (IGXSetFormSource: the open end of the IGASetForm* plugin API, include/petiga.h:153-197,
src/petigaform.c:388-833).  The compile (hiprtc, no GPU needed) and its error reporting are checked on the CPU; the launch is a
GPU test against the oracle's restatement of a reference demo that is NOT among the built-in forms (demo/AdvectionDiffusion.c)
and against the built-in Bratu form given once more as source."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, warped_geometry

ADVECTION_DIFFUSION = r"""
// demo/AdvectionDiffusion.c:26-47: K = grad Na . grad Nb + Na (w . grad Nb), F = 0; params = wind[3]
template <int DIM> struct AdvDiff {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double diffusion = 0, advection = 0;
    for (int i = 0; i < DIM; ++i) { diffusion += Na[1 + i] * Nb[1 + i]; advection += p.prm[i] * Nb[1 + i]; }
    T[0] = diffusion + Na[0] * advection;
  }
  static __device__ void vec(const PtView &, const double *, double *R) { R[0] = 0.0; }
};
"""

BRATU_AGAIN = r"""
// demo/BratuFJ.F90:23-176 as user source; params = {lambda}
struct UserBratu {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_U | NEED_UT | NEED_GU;
  static __device__ void vec(const PtView &p, const double *Na, double *R) {
    double s = 0; for (int i = 0; i < 2; ++i) s += Na[1 + i] * p.gu[i];
    R[0] = Na[0] * p.ut[0] + s - Na[0] * p.prm[0] * exp(p.u[0]);
  }
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double s = 0; for (int i = 0; i < 2; ++i) s += Na[1 + i] * Nb[1 + i];
    T[0] = p.shift * Na[0] * Nb[0] + s - Na[0] * Nb[0] * p.prm[0] * exp(p.u[0]);
  }
};
"""


USER_ELASTICITY = r"""
// demo/Elasticity3D.c:13-46 given as source (with its :37 quirk); params = {lambda, mu}.  GRAM = 1 adds the declaration that the
// coefficients are point-independent (MAT_PAIR_MASK): the matrix cores then accumulate the feature Gram matrices only.
template <int GRAM> struct UserElasticity;
template <> struct UserElasticity<0> {
  static constexpr int DOF = 3, ORDER = 1; static constexpr unsigned NEED = 0;
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const double l = p.prm[0], m = p.prm[1];
    const double ax = Na[1], ay = Na[2], az = Na[3], bx = Nb[1], by = Nb[2], bz = Nb[3];
    T[0] = ax * bx * (l + 2 * m) + m * (ay * by + az * bz); T[1] = ax * by * l + ay * bx * m; T[2] = ax * bz * l + az * bx * m;
    T[3] = ax * by * m + ay * bx * l; T[4] = ay * by * (l + 2 * m) + m * (az * bz + ax * bx * m); T[5] = ay * bz * l + az * by * m;
    T[6] = ax * bz * m + az * bx * l; T[7] = ay * bz * m + az * by * l; T[8] = m * (ax * bx + ay * by) + az * bz * (l + 2 * m);
  }
  static __device__ void vec(const PtView &, const double *, double *R) { R[0] = 0; R[1] = 0; R[2] = 0; }
};
template <> struct UserElasticity<1> : UserElasticity<0> {
  static constexpr unsigned MAT_TEST_MASK = 0xEu;
  static constexpr unsigned long long MAT_PAIR_MASK = (0xEull << 8) | (0xEull << 16) | (0xEull << 24);
};
"""


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_user_form_compiles_without_a_gpu(dim):
    import petiga_amd as P
    g = P.IGX(dim, 1)
    g.set_form_source(ADVECTION_DIFFUSION, "AdvDiff<%d>" % dim, (1.0, 0.5, 0.25))


@pytest.mark.parametrize("dim,p,dof,src,name,gram", [(2, 2, 1, ADVECTION_DIFFUSION, "AdvDiff<2>", False), (3, 3, 1, ADVECTION_DIFFUSION, "AdvDiff<3>", False),
                                                    (3, 5, 1, ADVECTION_DIFFUSION, "AdvDiff<3>", False),      # 16 tile rows x two column panels
                                                    (3, 2, 3, USER_ELASTICITY, "UserElasticity<0>", False), (3, 3, 3, USER_ELASTICITY, "UserElasticity<1>", True)])
def test_user_form_compiles_for_the_matrix_core_kernel_without_a_gpu(dim, p, dof, src, name, gram):
    """IGXCheckFormSource: the feature_assemble instantiations the drivers would launch (matrix and vector-only)."""
    import petiga_amd as P
    g = P.IGX(dim, dof)
    for i in range(dim):
        g.axis_uniform(i, p, 4)
    g.set_form_source(src, name, (1.0, 0.5, 0.25))
    g.check_form_source(True, gram)
    g.check_form_source(False, gram)


def test_compile_errors_come_back_with_the_log():
    import petiga_amd as P
    g = P.IGX(2, 1)
    with pytest.raises(P.IGXError) as e:
        g.set_form_source("struct Broken { static constexpr int DOF = 1, ORDER = 1; static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) { T[0] = undefined_symbol; } };", "Broken")
    assert e.value.code == 83 and "undefined_symbol" in str(e.value) and "user_form.hip" in str(e.value)
    with pytest.raises(P.IGXError):      # the struct lacks vec(): instantiating the element kernel fails
        g.set_form_source("struct NoVec { static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0; static __device__ void mat(const PtView &, const double *, const double *, double *T) { T[0] = 0; } };", "NoVec")


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("dim,p,N,geo", [(1, 2, 9, False), (2, 2, 7, False), (2, 3, 5, True), (3, 2, 4, False), (3, 3, 3, True), (3, 1, 5, True), (3, 4, 2, True)])
def test_advection_diffusion_source_form_matches_oracle(dim, p, N, geo, kernel):
    orc, eng = make_pair(dim, 1, p, N)
    eng.set_kernel(kernel)
    if geo:
        X, W = warped_geometry(orc, dim, seed=dim + 20, rational=True, amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):          # demo/AdvectionDiffusion.c:75-80: u = 1 on the inflow faces, 0 on the outflow faces
        for d in range(dim):
            g.set_boundary_value(d, 0, 0, 1.0)
            g.set_boundary_value(d, 1, 0, 0.0)
    wind = np.array([1.0, 0.6, -0.3]) * 10.0
    A_o, b_o = orc.compute_system("orc_form_advection_diffusion", (C.c_double * 3)(*wind))
    eng.set_form_source(ADVECTION_DIFFUSION, "AdvDiff<%d>" % dim, wind)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "hiprtc" in eng.kernel_name()
    # the matrix-core kernel when it covers the case (dim >= 2, nen <= 64), the point-form kernel otherwise / on request
    assert ("mfma" in eng.kernel_name()) == (kernel == 0 and dim >= 2)
    if p == 4 and kernel == 0:
        assert "tiles=8x8" in eng.kernel_name()
    tol = 1e-11 if geo else 1e-12
    compare_mats(A, A_o, tol)
    assert np.abs(b.get() - b_o).max() <= tol * max(np.abs(b_o).max(), 1.0)
    # IGAComputeMatrix (no fix-up) on the same module
    orc.clear_boundary()
    A_o2, _ = orc.compute_system("orc_form_advection_diffusion", (C.c_double * 3)(*wind))
    eng.compute_matrix(A)
    eng.synchronize()
    compare_mats(A, A_o2, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("gram", [0, 1])
@pytest.mark.parametrize("p,N,geo", [(2, (4, 3, 5), False), (3, (3, 4, 3), False), (3, (3, 3, 4), True)])
def test_vector_valued_source_form_on_the_matrix_cores(p, N, geo, gram):
    """dof = 3: all row fields in one launch; with MAT_PAIR_MASK the Gram path of the feature kernel."""
    orc, eng = make_pair(3, 3, p, list(N))
    if geo:
        X, W = warped_geometry(orc, 3, seed=31, rational=True, amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
        g.set_boundary_value(2, 1, 1, -0.5)
    eng.set_form_source(USER_ELASTICITY, "UserElasticity<%d>" % gram, (2.5, 0.7))
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "hiprtc,mfma" in eng.kernel_name()
    A_o, b_o = orc.compute_system("orc_form_elasticity", O.ElasticityCtx(2.5, 0.7))
    tol = 1e-11 if geo else 1e-12
    compare_mats(A, A_o, tol)
    assert np.abs(b.get() - b_o).max() <= tol * max(np.abs(b_o).max(), 1.0)
    eng.compute_vector(b)                  # the vector-only instantiation of the same source
    eng.synchronize()
    assert "vector only" in eng.kernel_name() and np.abs(b.get()).max() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
def test_nonlinear_source_form_through_function_and_jacobian_drivers(kernel):
    orc, eng = make_pair(2, 1, 2, 6)
    eng.set_kernel(kernel)
    for g in (orc, eng):
        for d in range(2):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 0.0)
    lam = C.c_double(6.8)
    rng = np.random.default_rng(2)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.4, rng.standard_normal(n)
    eng.set_form_source(BRATU_AGAIN, "UserBratu", (6.8,))
    Uv, Vv, F, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_vec(), eng.create_mat()
    eng.compute_function(Uv, F)
    eng.compute_jacobian(Uv, J)
    eng.synchronize()
    assert np.abs(F.get() - orc.compute_function("orc_form_bratu_function", lam, U)).max() <= 1e-12 * np.abs(U).max()
    compare_mats(J, orc.compute_jacobian("orc_form_bratu_jacobian", lam, U), 1e-12)
    eng.compute_ifunction(12.5, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(12.5, Vv, 0.0, Uv, J)
    eng.synchronize()
    F_o = orc.compute_ifunction("orc_form_bratu_ifunction", lam, 12.5, V, 0.0, U)
    assert np.abs(F.get() - F_o).max() <= 1e-12 * np.abs(F_o).max()
    compare_mats(J, orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, 12.5, V, 0.0, U), 1e-12)


USER_MASS4 = r"""
// four fields coupled through a constant matrix C (params: its diagonal and off-diagonal value) times the mass matrix, plus a load:
// K[(a,i),(b,j)] = C_ij Na Nb, F[(a,i)] = (i+1) Na.  No MAT_PAIR_MASK: the kernel takes it as a general (point-dependent) form.
struct UserMass4 {
  static constexpr int DOF = 4, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 0x1u;
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T[i * 4 + j] = (i == j ? p.prm[0] : p.prm[1]) * Na[0] * Nb[0];
  }
  static __device__ void vec(const PtView &, const double *Na, double *R) { for (int i = 0; i < 4; ++i) R[i] = (i + 1) * Na[0]; }
};
"""


@pytest.mark.gpu
def test_four_field_source_form_runs_as_fused_groups_on_the_matrix_cores():
    """dof = 4 at p = 3 (4x4 tiles): two groups of two row fields in one launch (feature_mfma.hpp, FUSE), here through hiprtc.
    The oracle's mass form gives M (dof = 1); the expected matrix is the Kronecker product M x C."""
    import scipy.sparse as sp
    orc1, _ = make_pair(3, 1, 3, [3, 4, 3], engine=False)
    M, m = orc1.compute_system("orc_form_mass")
    _, eng = make_pair(3, 4, 3, [3, 4, 3])
    d, o = 2.0, -0.25
    eng.set_form_source(USER_MASS4, "UserMass4", (d, o))
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "hiprtc,mfma" in eng.kernel_name() and "fused" in eng.kernel_name()
    Cm = np.full((4, 4), o) + (d - o) * np.eye(4)
    K_ref = sp.kron(M.scipy(), Cm).tocsr()
    rows, cols, vals = A.to_coo_global()
    K = sp.coo_matrix((vals, (rows, cols)), shape=K_ref.shape).tocsr()
    assert abs(K - K_ref).max() <= 1e-12 * abs(K_ref).max()
    # F[(a,i)] = (i+1) * integral of N_a = (i+1) * (M 1)_a
    rowsum = np.asarray(M.scipy().sum(axis=1)).ravel()
    assert np.abs(b.get().reshape(-1, 4) - rowsum[:, None] * np.arange(1, 5)[None, :]).max() <= 1e-12 * rowsum.max()


def test_code_object_cache_on_disk(tmp_path, monkeypatch):
    """IGX_RTC_CACHE_DIR: the second compile of the same program is a file read (and needs no hiprtc); a changed source misses."""
    import time
    import petiga_amd as P
    monkeypatch.setenv("IGX_RTC_CACHE_DIR", str(tmp_path))

    def compile_once(src):
        g = P.IGX(3, 1)
        for i in range(3):
            g.axis_uniform(i, 2, 4)
        t = time.perf_counter()
        g.set_form_source(src, "AdvDiff<3>", (1.0, 0.5, 0.25))
        g.check_form_source(True)
        return time.perf_counter() - t

    t_cold = compile_once(ADVECTION_DIFFUSION)
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len(files) == 2 and all(f.startswith("igx_") and f.endswith(".bin") for f in files)      # point-form kernel + matrix-core kernel
    t_warm = compile_once(ADVECTION_DIFFUSION)
    assert sorted(p.name for p in tmp_path.iterdir()) == files and t_warm < 0.5 * t_cold
    compile_once(ADVECTION_DIFFUSION + "\n// another program\n")
    assert len(list(tmp_path.iterdir())) == 4
    # a damaged entry is ignored and rewritten
    victim = tmp_path / files[0]
    victim.write_bytes(victim.read_bytes()[:100])
    compile_once(ADVECTION_DIFFUSION)
    assert victim.stat().st_size > 1000


# ---- run-time scalar forms on the pencil walk (form_pencil, gram_mfma.hpp): the enum gate is off the headline path
USER_POISSON = r"""
// demo/Poisson3D.c:3-23 (System) as a user struct; params = {forcing}.  The three optional declarations tell the library what the
// callback's shape is: gradients only, symmetric, load on N only.
struct UserPoisson {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 0xEu, VEC_TEST_MASK = 0x1u;
  static constexpr bool MAT_SYMMETRIC = true;
  static __device__ void mat(const PtView &, const double *Na, const double *Nb, double *T) { T[0] = Na[1] * Nb[1] + Na[2] * Nb[2] + Na[3] * Nb[3]; }
  static __device__ void vec(const PtView &p, const double *Na, double *R) { R[0] = Na[0] * p.prm[0]; }
};
// an anisotropic diffusion tensor and a load that both depend on the point: D(x) = diag(1 + x0, 2, 1 + x1 x2) + off-diagonal 0.3 x0
// on (0,1); f(x) = prm[0] (1 + x0 - x2): not a built-in form, same shape
struct UserDiffusion {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_X;
  static constexpr unsigned MAT_TEST_MASK = 0xEu, VEC_TEST_MASK = 0x1u;
  static constexpr bool MAT_SYMMETRIC = true;
  static __device__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const double d00 = 1.0 + p.x[0], d11 = 2.0, d22 = 1.0 + p.x[1] * p.x[2], d01 = 0.3 * p.x[0];
    T[0] = d00 * Na[1] * Nb[1] + d11 * Na[2] * Nb[2] + d22 * Na[3] * Nb[3] + d01 * (Na[1] * Nb[2] + Na[2] * Nb[1]);
  }
  static __device__ void vec(const PtView &p, const double *Na, double *R) { R[0] = Na[0] * p.prm[0] * (1.0 + p.x[0] - p.x[2]); }
};
"""


@pytest.mark.parametrize("p,geo", [(3, False), (2, False), (3, True)])
def test_pencil_walk_compiles_for_a_user_form_without_a_gpu(p, geo):
    """IGXCheckFormSource(gram = 2): form_pencil<System / Matrix, p, no geometry / rational geometry, UserForm>."""
    import petiga_amd as P
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, 8)
    if geo:
        orc, _ = make_pair(3, 1, p, 8, engine=False)
        g.setup()
        g.set_geometry(*warped_geometry(orc, 3, seed=1, rational=True, amp=0.1))
    g.set_form_source(USER_POISSON, "UserPoisson", (1.0,))
    g.check_form_source(True, 2)
    g.set_form_source(USER_POISSON, "UserDiffusion", (0.7,))
    g.check_form_source(True, 2)


def _dirichlet(objs, kind):
    for g in objs:
        if kind == "all":
            for d in range(3):
                for s in range(2):
                    g.set_boundary_value(d, s, 0, 0.5 + 0.25 * d + 0.125 * s)
        elif kind == "partial":
            g.set_boundary_value(0, 0, 0, 2.0)
            g.set_boundary_value(2, 1, 0, -1.0)
            g.set_boundary_load(1, 1, 0, 0.75)


@pytest.mark.gpu
@pytest.mark.parametrize("p,N,bc,geo", [(3, (9, 5, 6), "all", None), (3, (8, 4, 5), "partial", "nurbs"), (2, (8, 6, 5), "all", None), (2, (9, 4, 4), "partial", "poly"),
                                        (3, (12, 4, 4), "none", "nurbs")])
def test_user_poisson_on_the_pencil_walk_matches_oracle(p, N, bc, geo):
    """demo/Poisson3D.c's System given as source runs on form_pencil (not on the element mode) and reproduces the oracle."""
    orc, eng = make_pair(3, 1, p, list(N))
    if geo:
        X, W = warped_geometry(orc, 3, seed=sum(N), rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    _dirichlet((orc, eng), bc)
    A_o, b_o = orc.compute_system("orc_form_poisson")
    eng.set_form_source(USER_POISSON, "UserPoisson", (1.0,))
    eng.set_kernel(2)                       # insist on the pencil walk: an uncovered case would be an error
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "form_pencil<UserPoisson>" in eng.kernel_name(), eng.kernel_name()
    tol = 1e-11 if geo else 1e-12
    compare_mats(A, A_o, tol)
    assert np.abs(b.get() - b_o).max() <= tol * max(np.abs(b_o).max(), 1e-300)
    orc.clear_boundary()
    A_o2, _ = orc.compute_system("orc_form_poisson")
    eng.compute_matrix(A)
    eng.synchronize()
    assert "form_pencil" in eng.kernel_name()
    compare_mats(A, A_o2, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("p,geo", [(3, False), (3, True), (2, True)])
def test_variable_coefficient_user_form_pencil_walk_equals_element_mode(p, geo):
    """A form no built-in covers (x-dependent anisotropic diffusion, x-dependent load): the pencil walk against the element mode of the
    feature kernel and the point-form kernel on the same source -- three device paths, one matrix."""
    orc, eng = make_pair(3, 1, p, [10, 5, 4])
    if geo:
        X, W = warped_geometry(orc, 3, seed=9, rational=True, amp=0.1)
        eng.set_geometry(X, W)
    _dirichlet((eng,), "all")
    eng.set_form_source(USER_POISSON, "UserDiffusion", (0.7,))
    outs = {}
    for kernel in (2, 3, 1):
        eng.set_kernel(kernel)
        A, b = eng.create_mat(), eng.create_vec()
        eng.compute_system(A, b)
        eng.synchronize()
        outs[kernel] = (A.host(True), b.get(), eng.kernel_name())
    assert "form_pencil" in outs[2][2] and "feature_assemble" in outs[3][2] and "generic_assemble" in outs[1][2]
    scale = np.abs(outs[1][0]).max()
    for k in (2, 3):
        assert np.abs(outs[k][0] - outs[1][0]).max() <= 1e-11 * scale
        assert np.abs(outs[k][1] - outs[1][1]).max() <= 1e-11 * np.abs(outs[1][1]).max()
