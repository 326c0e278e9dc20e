// forms.hpp -- device point forms.  Each struct restates one of the reference's point callbacks
// (include/petiga.h:153-197) with the same contract: the UN-weighted integrand at one quadrature
// point, K block [dof][dof] for the basis pair (a,b) and F entries [dof] for basis a.
//
// Na / Nb point at the feature vector of one basis function at the point:
//   [0] = N, [1+i] = dN/dx_i, [1+DIM+i*DIM+j] = d2N/dx_i dx_j      (shape[0], shape[1], shape[2])
// Every `mat` below is linear in Nb, which the engine uses to apply Dirichlet lifting without
// storing K_e columns (see generic_kernel.hpp, phase "lift").
#pragma once
#include "igx.hpp"

namespace igx {

enum : unsigned { NEED_X = 1u, NEED_U = 2u, NEED_UT = 4u, NEED_GU = 8u, NEED_HU = 16u, NEED_G = 32u, NEED_D3U = 64u, NEED_PROP = 128u, NEED_MAPX = 256u };
// a form of ORDER 3, or one that reads the property array / the point's shape table, runs on the general kernel only
template <class Form> struct general_only_of { static constexpr bool v = Form::ORDER >= 3 || (Form::NEED & (NEED_PROP | NEED_D3U | NEED_MAPX)) != 0; };

struct PtView {
  const double *x;     // physical point [DIM] (parametric point when there is no geometry)
  const double *u;     // field values      [dof]
  const double *ut;    // time derivative   [dof]
  const double *gu;    // gradient          [dof][DIM]
  const double *hu;    // hessian           [dof][DIM][DIM]
  const double *G;     // IGAPointFormInvGradGeomMap [DIM][DIM] (src/petigapoint.c:269-294)
  // the general kernel only (forms of ORDER 3 or with NEED_PROP are routed there):
  const double *d3u = nullptr;       // third derivatives [dof][DIM][DIM][DIM] (IGAPointFormDer3, include/petiga.h:731; NEED_D3U)
  const double *property = nullptr;  // p->property: the property array of the element's nodes [nen][npd] (include/petiga.h:662); npd = 0 without one
  const double *shape = nullptr;     // the point's shape functions [nen][nf] (value, gradient, ... laid out as a form's Na), e.g. to interpolate the properties
  int npd = 0, nen = 0, nf = 0;
  // a geometry of another dimension than the parametric one (IGASetGeometryDim, nsd != dim: x, normal and G then have nsd columns, the
  // shape functions stay parametric) and forms with NEED_MAPX: the geometry map's derivatives at the point, p->mapX[1] / p->mapX[2]
  // (IGAPointFormGradGeomMap / HessGeomMap, src/petigapoint.c:243-263) [nsd][dim] / [nsd][dim][dim]; X2 for forms of ORDER >= 2
  int nsd = 0;
  const double *X1 = nullptr, *X2 = nullptr;
  const double *prm;   // form parameters (replaces ctx)
  double shift, t;
  const double *normal;  // unit outward normal [DIM] at a boundary-form point (p->normal), else null
  int atboundary, boundary_id;   // IGAPoint::atboundary / boundary_id (include/petiga.h:650-652)
};

// a form whose Tangent can take the pencil walk with the state summed inside the wavefront (gram_mfma.hpp: state_pencil) says so
// with PENCIL_NFEAT / PENCIL_NC / pencil_coef / pencil_trial (see FormCahnHilliard, FormBratu below)
template <class Form, class = void> struct pencil_state_of { static constexpr bool v = false; static constexpr int nfeat = 0, nc = 0; };
template <class Form> struct pencil_state_of<Form, decltype((void)Form::PENCIL_NFEAT)> { static constexpr bool v = true; static constexpr int nfeat = Form::PENCIL_NFEAT, nc = Form::PENCIL_NC; };
template <> struct pencil_state_of<void, void> { static constexpr bool v = false; static constexpr int nfeat = 0, nc = 0; };
// ... and its Residual can ride on the same walk (a fused IFunction + IJacobian pass: state_pencil_kr) when it is the same test
// features against point numbers, R_a = sum_q sum_f A_f(a, q) r_f(q): PENCIL_NCR more numbers per point and pencil_resid(c, r)
// for the PENCIL_NFEAT numbers r_f.  pencil_coef_r fills all PENCIL_NC + PENCIL_NCR from the state's point values WITHOUT u_t
// (p.ut is null there) and leaves JW in c[PENCIL_NC]; the engine evaluates u_t from V later -- at the head of the wave's own MFMA
// phase, where fp64 arithmetic does not wait for the partner's MFMAs -- and multiplies that slot by it: pencil_resid reads
// c[PENCIL_NC] = JW u_t.
template <class Form, class = void> struct pencil_resid_of { static constexpr bool v = false; static constexpr int ncr = 0; };
template <class Form> struct pencil_resid_of<Form, decltype((void)Form::PENCIL_NCR)> { static constexpr bool v = true; static constexpr int ncr = Form::PENCIL_NCR; };
template <> struct pencil_resid_of<void, void> { static constexpr bool v = false; static constexpr int ncr = 0; };

// demo/Poisson{1,2,3}D.c System (demo/Poisson3D.c:3-23)
template <int DIM> struct FormPoisson {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = ((1u << DIM) - 1u) << 1;   // gradients only
  static __device__ __forceinline__ void mat(const PtView &, const double *Na, const double *Nb, double *T) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) s += Na[1 + i] * Nb[1 + i];
    T[0] = s;
  }
  static __device__ __forceinline__ void vec(const PtView &, const double *Na, double *R) { R[0] = Na[0] * 1.0; }
};

// test/IGACreate.c:45-63 System (block-diagonal mass, F = N)
template <int DIM, int DOF_> struct FormMass {
  static constexpr int DOF = DOF_, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 1u;   // values only
  static constexpr unsigned long long MAT_PAIR_MASK = 1ull;   // N x N
  static constexpr unsigned pair_block_mask(int, int) { unsigned m = 0; for (int i = 0; i < DOF_; ++i) m |= 1u << (i * DOF_ + i); return m; }   // block diagonal
  static __device__ __forceinline__ void mat(const PtView &, const double *Na, const double *Nb, double *T) {
#pragma unroll
    for (int i = 0; i < DOF * DOF; ++i) T[i] = 0;
#pragma unroll
    for (int i = 0; i < DOF; ++i) T[i * DOF + i] = Na[0] * Nb[0];
  }
  static __device__ __forceinline__ void vec(const PtView &, const double *Na, double *R) {
#pragma unroll
    for (int i = 0; i < DOF; ++i) R[i] = Na[0];
  }
};

// test/IGAFixTable.c:25-43 System1 (L2 projection of sum x_i^2)
template <int DIM> struct FormL2ProjX2 {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_X;
  static constexpr unsigned MAT_TEST_MASK = 1u;
  static constexpr unsigned long long MAT_PAIR_MASK = 1ull;
  static __device__ __forceinline__ void mat(const PtView &, const double *Na, const double *Nb, double *T) { T[0] = Na[0] * Nb[0]; }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) {
    double g = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) g += p.x[i] * p.x[i];
    R[0] = Na[0] * g;
  }
};

// test/IGAFixTable.c:45-64 System2 (Poisson, f = -2 dim)
template <int DIM> struct FormPoissonF {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = FormPoisson<DIM>::MAT_TEST_MASK;
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) { FormPoisson<DIM>::mat(p, Na, Nb, T); }
  static __device__ __forceinline__ void vec(const PtView &, const double *Na, double *R) { R[0] = Na[0] * (-2.0 * DIM); }
};

// test/IGAErrNorm.c:26-75 System (4 fields: 1, sum x, sum x^2, prod x)
template <int DIM> struct FormErrNorm {
  static constexpr int DOF = 4, ORDER = 1; static constexpr unsigned NEED = NEED_X;
  static constexpr unsigned MAT_TEST_MASK = 1u;
  static constexpr unsigned long long MAT_PAIR_MASK = 1ull;
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) { FormMass<DIM, 4>::mat(p, Na, Nb, T); }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) {
    double s1 = 0, s2 = 0, pr = 1;
#pragma unroll
    for (int i = 0; i < DIM; ++i) { s1 += p.x[i]; s2 += p.x[i] * p.x[i]; pr *= p.x[i]; }
    R[0] = Na[0] * 1.0; R[1] = Na[0] * s1; R[2] = Na[0] * s2; R[3] = Na[0] * pr;
  }
};

// A form on the third derivatives (IGASetOrder(iga,3): p->shape[3] as test/IGAGeometryMap.c:179,221 reads it, IGAPointFormDer3 as
// demo/AutoDiff/CahnHilliardPrimalFAD.cxx:51 does).  params {k3, f3, u3}:
//   K_ab = N_a N_b + k3 sum_ijk d_ijk N_a d_ijk N_b,   F_a = N_a (1 + |x|^2) + f3 sum_ijk c_ijk d_ijk N_a + u3 N_a sum_ijk c_ijk d_ijk u,
// c_ijk = 1 / (1 + i + 2 j + 3 k); the last term through the Function / IFunction drivers (u = 0 under the System driver).
template <int DIM> struct FormDer3 {
  static constexpr int DOF = 1, ORDER = 3; static constexpr unsigned NEED = NEED_X | NEED_U | NEED_D3U;
  static constexpr int O3 = 1 + DIM + DIM * DIM;
  static __device__ __forceinline__ double c3(int i, int j, int k) { return 1.0 / (1.0 + i + 2.0 * j + 3.0 * k); }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double s = 0;
#pragma unroll
    for (int f = 0; f < DIM * DIM * DIM; ++f) s += Na[O3 + f] * Nb[O3 + f];
    T[0] = Na[0] * Nb[0] + p.prm[0] * s;
  }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) {
    double x2 = 0, s = 0, su = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) x2 += p.x[i] * p.x[i];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int k = 0; k < DIM; ++k) { const double c = c3(i, j, k); s += c * Na[O3 + (i * DIM + j) * DIM + k]; su += c * p.d3u[(i * DIM + j) * DIM + k]; }
    R[0] = Na[0] * (1.0 + x2) + p.prm[1] * s + p.prm[2] * Na[0] * su;
  }
};

// A form on a curve or a surface in space (IGASetGeometryDim with nsd != dim, demo/ClassicalShell.c:154): the shape functions are the
// parametric basis, the metric is built here from p->mapX[1], p->mapX[2] as demo/ClassicalShell.c:57-80 does:
//   g = F^T F, a = sqrt det g,  K_ab = (N_a N_b + d_al N_a g^{al be} d_be N_b) a,  F_a = N_a |H| a,
// H = g^{al be} (d_al d_be x - Gamma^ga_{al be} d_ga x) the mean-curvature vector (1/R on a circle, 2/R on a sphere).  DIM = 1, 2.
// params {gs}: K_ab += gs N_a N_b sum G^2, G of IGAPointFormInvGradGeomMap (the pseudo-inverse when nsd != dim, src/petigaval.F90:124-142).
template <int DIM> struct FormSurface {
  static constexpr int DOF = 1, ORDER = 2; static constexpr unsigned NEED = NEED_MAPX | NEED_G;
  struct Metric { double gi[4], ar, Hn; };
  static __device__ __forceinline__ Metric metric(const PtView &p) {
    constexpr int D2 = DIM * DIM;
    const int nsd = p.nsd; const double *X1 = p.X1, *X2 = p.X2;
    Metric m; double g[4] = {0, 0, 0, 0}, detg, H[3] = {0, 0, 0};
    for (int al = 0; al < DIM; ++al) for (int be = 0; be < DIM; ++be) { double t = 0; for (int i = 0; i < nsd; ++i) t += X1[i * DIM + al] * X1[i * DIM + be]; g[al * DIM + be] = t; }
    if (DIM == 1) { detg = g[0]; m.gi[0] = 1 / g[0]; }
    else { detg = g[0] * g[3] - g[1] * g[2]; m.gi[0] = g[3] / detg; m.gi[1] = -g[1] / detg; m.gi[2] = -g[2] / detg; m.gi[3] = g[0] / detg; }
    m.ar = sqrt(detg);
    for (int al = 0; al < DIM; ++al) for (int be = 0; be < DIM; ++be) {
      double Gam[2] = {0, 0};
      for (int ga = 0; ga < DIM; ++ga) for (int de = 0; de < DIM; ++de) { double t = 0; for (int i = 0; i < nsd; ++i) t += X1[i * DIM + de] * X2[i * D2 + al * DIM + be]; Gam[ga] += m.gi[ga * DIM + de] * t; }
      for (int i = 0; i < nsd; ++i) { double h = X2[i * D2 + al * DIM + be]; for (int ga = 0; ga < DIM; ++ga) h -= Gam[ga] * X1[i * DIM + ga]; H[i] += m.gi[al * DIM + be] * h; }
    }
    double hn = 0; for (int i = 0; i < nsd; ++i) hn += H[i] * H[i];
    m.Hn = sqrt(hn);
    return m;
  }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const Metric m = metric(p);
    double s = 0;
    for (int al = 0; al < DIM; ++al) for (int be = 0; be < DIM; ++be) s += Na[1 + al] * m.gi[al * DIM + be] * Nb[1 + be];
    double G2 = 0;
    for (int i = 0; i < DIM * p.nsd; ++i) G2 += p.G[i] * p.G[i];
    T[0] = (Na[0] * Nb[0] + s) * m.ar + p.prm[0] * Na[0] * Nb[0] * G2;
  }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) { const Metric m = metric(p); R[0] = Na[0] * m.Hn * m.ar; }
};

// A Poisson problem whose conductivity and source live on the control net as a property array (IGASetPropertyDim, include/petiga.h:350-353;
// the point sees its element's nodal values, p->property [nen][npd], and interpolates them with its own shape functions p->shape[0]):
//   k(q) = sum_a N_a(q) A_a[0],  f(q) = sum_a N_a(q) A_a[npd - 1],   K_ab = k grad N_a . grad N_b,   F_a = N_a f.
template <int DIM> struct FormProperty {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_PROP;
  static __device__ __forceinline__ double interp(const PtView &p, int c) { double s = 0; for (int a = 0; a < p.nen; ++a) s += p.shape[a * p.nf] * p.property[a * p.npd + c]; return s; }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) s += Na[1 + i] * Nb[1 + i];
    T[0] = interp(p, 0) * s;
  }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) { R[0] = Na[0] * interp(p, p.npd - 1); }
};

// demo/Elasticity3D.c:13-46 System; params {lambda, mu}.  The reference's [1][1] block carries an
// extra factor mu on its xx term (line 37); kept.
struct FormElasticity {
  static constexpr int DOF = 3, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 0xEu;   // gradients only
  static constexpr unsigned long long MAT_PAIR_MASK = (0xEull << 8) | (0xEull << 16) | (0xEull << 24);   // lambda, mu constant: all grad x grad pairs
  static constexpr bool VEC_ZERO = true;            // F = 0 (demo/Elasticity3D.c:43-45)
  // entries i*3+j of the block that the Gram pair (d_f N_a, d_g N_b) reaches: the diagonal for f = g, else (f,g) and (g,f)
  // (x * 0 does not fold under IEEE rules: without this the coefficient transform runs 81 multiply-adds per block instead of 21)
  static constexpr unsigned pair_block_mask(int f, int g) { return f == g ? 0x111u : ((1u << ((f - 1) * 3 + (g - 1))) | (1u << ((g - 1) * 3 + (f - 1)))); }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const double lambda = p.prm[0], mu = p.prm[1];
    const double Na_x = Na[1], Na_y = Na[2], Na_z = Na[3], Nb_x = Nb[1], Nb_y = Nb[2], Nb_z = Nb[3];
    T[0] = Na_x * Nb_x * (lambda + 2 * mu) + mu * (Na_y * Nb_y + Na_z * Nb_z);
    T[1] = Na_x * Nb_y * lambda + Na_y * Nb_x * mu;
    T[2] = Na_x * Nb_z * lambda + Na_z * Nb_x * mu;
    T[3] = Na_x * Nb_y * mu + Na_y * Nb_x * lambda;
    T[4] = Na_y * Nb_y * (lambda + 2 * mu) + mu * (Na_z * Nb_z + Na_x * Nb_x * mu);
    T[5] = Na_y * Nb_z * lambda + Na_z * Nb_y * mu;
    T[6] = Na_x * Nb_z * mu + Na_z * Nb_x * lambda;
    T[7] = Na_y * Nb_z * mu + Na_z * Nb_y * lambda;
    T[8] = mu * (Na_x * Nb_x + Na_y * Nb_y) + Na_z * Nb_z * (lambda + 2 * mu);
  }
  static __device__ __forceinline__ void vec(const PtView &, const double *, double *R) { R[0] = 0; R[1] = 0; R[2] = 0; }
};
// ... with a body force: the same K, F[a][i] = N_a f_i (the callback path of the reference is uniform in what a System callback puts
// into F: demo/Elasticity3D.c:43-45 writes zeros, src/petigapoint.c:427-450 adds whatever it finds); params {lambda, mu, fx, fy, fz}
struct FormElasticityF : FormElasticity {
  static constexpr bool VEC_ZERO = false;
  static constexpr unsigned VEC_TEST_MASK = 1u;      // vec() reads N only
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) { R[0] = Na[0] * p.prm[2]; R[1] = Na[0] * p.prm[3]; R[2] = Na[0] * p.prm[4]; }
};

// The VECTOR of IGAComputeSystem of a linear form without its K_e (the band-row kernels never hold one): per element
//   F_e = vec - sum_k K_e[:, k] v_k over the fixed dofs k,  F_e[k] = v_k          (IGAElementFixSystem, src/petigaelem.c:1377-1387)
// and mat() is linear in N_b, so sum_k K_e[a][k] v_k = sum_q JW mat(p, N_a, u_D) with u_D the field that holds v on the element's fixed
// dofs and 0 elsewhere -- a vector-only pass (vec_sumfact.hpp, OutDev::vec_mode 1: the state is u_D, a fixed row takes v).
template <class Form> struct SystemVectorOf {
  static constexpr int DOF = Form::DOF, ORDER = Form::ORDER;
  static constexpr unsigned NEED = Form::NEED | NEED_U | NEED_GU;
  static_assert(Form::ORDER < 2, "first-order forms (the lifting reads u_D and its gradient)");
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) {
    Form::vec(p, Na, R);
#pragma unroll
    for (int j = 0; j < DOF; ++j) {      // column j of the blocks, the trial function's features replaced by those of field j of u_D
      const double Nb[4] = {p.u[j], p.gu[j * 3 + 0], p.gu[j * 3 + 1], p.gu[j * 3 + 2]};
      double T[DOF * DOF];
      Form::mat(p, Na, Nb, T);
#pragma unroll
      for (int i = 0; i < DOF; ++i) R[i] -= T[i * DOF + j];
    }
  }
};

// demo/CahnHilliard3D.c:11-16,39-53,55-179 (Residual / Tangent); 2-D: demo/CahnHilliard2D.c.
// params {theta, alpha, cbar, L0, lambda, tau}; L0 <= 0 selects the 2-D demo's 3*alpha scaling.
template <int DIM> struct FormCahnHilliard {
  static constexpr unsigned hess_diag_mask() { unsigned m = 0; for (int i = 0; i < DIM; ++i) m |= 1u << (1 + DIM + i * (DIM + 1)); return m; }
  static constexpr unsigned MAT_TEST_MASK = 1u | (((1u << DIM) - 1u) << 1) | hess_diag_mask();   // N, grad N, diagonal of hess N (Laplacian)
  static constexpr unsigned PHI_MASK = MAT_TEST_MASK;   // Residual, Tangent and the field values read nothing else (hu: its diagonal)
  static constexpr unsigned VEC_TEST_MASK = MAT_TEST_MASK;   // the Residual reads the same test features
  static constexpr int DOF = 1, ORDER = 2; static constexpr unsigned NEED = NEED_U | NEED_UT | NEED_GU | NEED_HU;
  struct Coef { double M, dM, d2M, dmu, d2mu, lap, t1; };
  static __device__ __forceinline__ Coef coef(const PtView &p) {
    Coef k; const double c = p.u[0], theta = p.prm[0], alpha = p.prm[1], L0 = p.prm[3], lambda = p.prm[4];
    const double scale = (L0 > 0) ? L0 * L0 / lambda : 3 * alpha;
    k.M = c * (1 - c); k.dM = 1 - 2 * c; k.d2M = -2;
    k.dmu = (0.5 / theta * 1.0 / (c * (1 - c)) - 2) * scale;
    k.d2mu = (-0.5 / theta * (1 - 2 * c) / (c * c * (1 - c) * (1 - c))) * scale;
    k.lap = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) k.lap += p.hu[i * (DIM + 1)];
    k.t1 = k.M * k.dmu + k.dM * k.lap;
    return k;
  }
  static __device__ __forceinline__ double lapN(const double *N) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) s += N[1 + DIM + i * (DIM + 1)];
    return s;
  }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) {
    const Coef k = coef(p);
    double Ra = Na[0] * p.ut[0];
#pragma unroll
    for (int i = 0; i < DIM; ++i) Ra += Na[1 + i] * k.t1 * p.gu[i];
    Ra += lapN(Na) * k.M * k.lap;
    R[0] = Ra;
  }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    const Coef k = coef(p);
    const double lapNa = lapN(Na), lapNb = lapN(Nb);
    double Kab = p.shift * Na[0] * Nb[0];
#pragma unroll
    for (int i = 0; i < DIM; ++i) Kab += Na[1 + i] * k.t1 * Nb[1 + i];
    const double t2 = (k.dM * k.dmu + k.M * k.d2mu + k.d2M * k.lap) * Nb[0] + k.dM * lapNb;
#pragma unroll
    for (int i = 0; i < DIM; ++i) Kab += Na[1 + i] * t2 * p.gu[i];
    Kab += lapNa * (k.dM * k.lap * Nb[0] + k.M * lapNb);
    T[0] = Kab;
  }
  // The Tangent on the pencil walk (gram_mfma.hpp, state_pencil): k_q[a][b] = sum_f A_f(a) B_f(b) with the test-side features
  // A = (N, d0 N, d1 N, d2 N, lap N) -- plain tensor products of the 1-D rows -- and the trial side carrying the point's
  // coefficients: B_0 = JW shift N, B_{1+i} = JW t1 d_i N + d_i c (JW s N + JW dM lap N), B_4 = JW dM lap(c) N + JW M lap N (DIM = 3).
  static constexpr int PENCIL_NFEAT = 2 + DIM, PENCIL_NC = 6 + DIM;
  static __device__ __forceinline__ void pencil_coef(const PtView &p, double JW, double *c) {
    const Coef k = coef(p);
    c[0] = JW * p.shift; c[1] = JW * k.t1; c[2] = JW * (k.dM * k.dmu + k.M * k.d2mu + k.d2M * k.lap); c[3] = JW * k.dM;
    c[4] = JW * (k.dM * k.lap); c[5] = JW * k.M;
#pragma unroll
    for (int i = 0; i < DIM; ++i) c[6 + i] = p.gu[i];
  }
  static __device__ __forceinline__ void pencil_trial(const double *c, double N, const double *g, double lap, double *B) {
    const double h = c[2] * N + c[3] * lap;
    B[0] = c[0] * N;
#pragma unroll
    for (int i = 0; i < DIM; ++i) B[1 + i] = c[1] * g[i] + c[6 + i] * h;
    B[1 + DIM] = c[4] * N + c[5] * lap;
  }
  // The Residual (demo/CahnHilliard3D.c:55-109) on the same test features: R_a = JW (N_a c_t + t1 grad N_a . grad c + lap N_a M lap c)
  static constexpr int PENCIL_NCR = 2;
  static __device__ __forceinline__ void pencil_coef_r(const PtView &p, double JW, double *c) {
    pencil_coef(p, JW, c);
    const Coef k = coef(p);
    c[PENCIL_NC] = JW; c[PENCIL_NC + 1] = JW * (k.M * k.lap);      // (c[PENCIL_NC] becomes JW c_t)
  }
  static __device__ __forceinline__ void pencil_resid(const double *c, double *r) {
    r[0] = c[PENCIL_NC];
#pragma unroll
    for (int i = 0; i < DIM; ++i) r[1 + i] = c[1] * c[6 + i];
    r[1 + DIM] = c[PENCIL_NC + 1];
  }
};

// demo/Bratu.c + demo/BratuFJ.F90:23-176 (Function / Jacobian and IFunction / IJacobian, Galerkin branches); params {lambda}.
// Function / Jacobian are the IFunction / IJacobian with V absent (ut = 0) and shift = 0.
template <int DIM> struct FormBratu {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_U | NEED_UT | NEED_GU;
  static constexpr unsigned MAT_NEED = NEED_U;
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na, double *R) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) s += Na[1 + i] * p.gu[i];
    R[0] = Na[0] * p.ut[0] + s - Na[0] * p.prm[0] * exp(p.u[0]);
  }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) s += Na[1 + i] * Nb[1 + i];
    T[0] = p.shift * Na[0] * Nb[0] + s - Na[0] * Nb[0] * p.prm[0] * exp(p.u[0]);
  }
  // the Jacobian on the pencil walk (gram_mfma.hpp, state_pencil): A = (N, grad N), B = (JW (shift - lambda e^u) N, JW grad N)
  static constexpr int PENCIL_NFEAT = 1 + DIM, PENCIL_NC = 2;
  static __device__ __forceinline__ void pencil_coef(const PtView &p, double JW, double *c) { c[0] = JW * (p.shift - p.prm[0] * exp(p.u[0])); c[1] = JW; }
  static __device__ __forceinline__ void pencil_trial(const double *c, double N, const double *g, double, double *B) {
    B[0] = c[0] * N;
#pragma unroll
    for (int i = 0; i < DIM; ++i) B[1 + i] = c[1] * g[i];
  }
  // the Function (demo/BratuFJ.F90:23-118) on the same test features: R_a = JW (N_a (u_t - lambda e^u) + grad N_a . grad u)
  static constexpr int PENCIL_NCR = 2 + DIM;
  static __device__ __forceinline__ void pencil_coef_r(const PtView &p, double JW, double *c) {
    pencil_coef(p, JW, c);
    c[PENCIL_NC] = JW; c[PENCIL_NC + 1] = -JW * (p.prm[0] * exp(p.u[0]));      // (c[PENCIL_NC] becomes JW u_t)
#pragma unroll
    for (int i = 0; i < DIM; ++i) c[PENCIL_NC + 2 + i] = JW * p.gu[i];
  }
  static __device__ __forceinline__ void pencil_resid(const double *c, double *r) {
    r[0] = c[PENCIL_NC] + c[PENCIL_NC + 1];
#pragma unroll
    for (int i = 0; i < DIM; ++i) r[1 + i] = c[PENCIL_NC + 2 + i];
  }
};

// demo/NavierStokesVMS.c:9-244 (Tau, FineScale, Residual, Tangent); params {nu, fx, fy, fz, dt}
struct FormNSVMS {
  static constexpr int SHAPE_ORDER = 1;   // Residual/Tangent read N and grad N only; Hessians are needed of U alone
  static constexpr unsigned MAT_NEED = NEED_U | NEED_G;   // Tangent (:166-244) reads u and the metric tensor only
  // test-function features (bit 0 = N, bits 1..3 = grad N) each (i,j) block of the Tangent reads: 42 of the 64
  // (block, feature) combinations are non-zero, the others never reach the matrix cores
  static constexpr unsigned block_mask(int i, int j) {
    if (i < 3 && j < 3) return (i == j) ? 0xFu : ((1u << (1 + i)) | (1u << (1 + j)));   // Tii: all; nu Na_j Nb_i + tauC Na_i Nb_j
    if (i < 3 && j == 3) return 0xEu;            // -Na_i Nb + tauM adv(a) Nb_i
    if (i == 3 && j < 3) return 0x1u | (1u << (1 + j));   // Na Nb_j + tauM Na_j (...)
    return 0xEu;                                 // tauM grad Na . grad Nb
  }
  static constexpr int DOF = 4, ORDER = 2; static constexpr unsigned NEED = NEED_U | NEED_UT | NEED_GU | NEED_HU | NEED_G;
  static __device__ __forceinline__ void tau(const PtView &p, double &tauM, double &tauC) {
    const double *J = p.G; const double nu = p.prm[0], dt = p.prm[4], C_I = 1.0 / 12.0;
    double G[9], g[3] = {0, 0, 0}, G_G = 0, g_g = 0, u_G_u = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += J[i * 3 + k] * J[j * 3 + k]; G[i * 3 + j] = s; }
#pragma unroll
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) g[i] += J[i * 3 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) G_G += G[i] * G[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) g_g += g[i] * g[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) u_G_u += p.u[i] * G[i * 3 + j] * p.u[j];
    tauM = 4 / (dt * dt) + u_G_u + C_I * nu * nu * G_G;
    tauM = 1 / sqrt(tauM);
    tauC = tauM * g_g;
    tauC = 1 / tauC;
  }
  static __device__ __forceinline__ void vec(const PtView &p, const double *Na_, double *R) {
    const double nu = p.prm[0], fx = p.prm[1], fy = p.prm[2], fz = p.prm[3];
    double tauM, tauC; tau(p, tauM, tauC);
    const double ux = p.u[0], uy = p.u[1], uz = p.u[2], pr = p.u[3];
    const double ux_t = p.ut[0], uy_t = p.ut[1], uz_t = p.ut[2];
    const double ux_x = p.gu[0], ux_y = p.gu[1], ux_z = p.gu[2], uy_x = p.gu[3], uy_y = p.gu[4], uy_z = p.gu[5];
    const double uz_x = p.gu[6], uz_y = p.gu[7], uz_z = p.gu[8], p_x = p.gu[9], p_y = p.gu[10], p_z = p.gu[11];
    const double ux_l = p.hu[0] + p.hu[4] + p.hu[8], uy_l = p.hu[9] + p.hu[13] + p.hu[17], uz_l = p.hu[18] + p.hu[22] + p.hu[26];
    double ux_s = ux_t + (ux * ux_x + uy * ux_y + uz * ux_z) + p_x - nu * ux_l - fx;
    double uy_s = uy_t + (ux * uy_x + uy * uy_y + uz * uy_z) + p_y - nu * uy_l - fy;
    double uz_s = uz_t + (ux * uz_x + uy * uz_y + uz * uz_z) + p_z - nu * uz_l - fz;
    double p_s = ux_x + uy_y + uz_z;
    ux_s *= -tauM; uy_s *= -tauM; uz_s *= -tauM; p_s *= -tauC;
    const double Na = Na_[0], Na_x = Na_[1], Na_y = Na_[2], Na_z = Na_[3];
    double Rux = -Na * fx, Ruy = -Na * fy, Ruz = -Na * fz, Rp = 0.0;
    Rux += Na * ux_t - Na_x * pr + nu * (Na_x * (ux_x + ux_x) + Na_y * (ux_y + uy_x) + Na_z * (ux_z + uz_x));
    Ruy += Na * uy_t - Na_y * pr + nu * (Na_x * (uy_x + ux_y) + Na_y * (uy_y + uy_y) + Na_z * (uy_z + uz_y));
    Ruz += Na * uz_t - Na_z * pr + nu * (Na_x * (uz_x + ux_z) + Na_y * (uz_y + uy_z) + Na_z * (uz_z + uz_z));
    Rp += Na * (ux_x + uy_y + uz_z);
    Rux += -(Na_x * p_s); Ruy += -(Na_y * p_s); Ruz += -(Na_z * p_s);
    Rp += -(Na_x * ux_s + Na_y * uy_s + Na_z * uz_s);
    Rux += +Na * ((ux + ux_s) * ux_x + (uy + uy_s) * ux_y + (uz + uz_s) * ux_z);
    Ruy += +Na * ((ux + ux_s) * uy_x + (uy + uy_s) * uy_y + (uz + uz_s) * uy_z);
    Ruz += +Na * ((ux + ux_s) * uz_x + (uy + uy_s) * uz_y + (uz + uz_s) * uz_z);
    Rux += -(Na_x * ux_s * (ux + ux_s) + Na_y * ux_s * (uy + uy_s) + Na_z * ux_s * (uz + uz_s));
    Ruy += -(Na_x * uy_s * (ux + ux_s) + Na_y * uy_s * (uy + uy_s) + Na_z * uy_s * (uz + uz_s));
    Ruz += -(Na_x * uz_s * (ux + ux_s) + Na_y * uz_s * (uy + uy_s) + Na_z * uz_s * (uz + uz_s));
    R[0] = Rux; R[1] = Ruy; R[2] = Ruz; R[3] = Rp;
  }
  // Point coefficients apart from the basis functions (band_pt.hpp evaluates them once per Gauss point, ahead of the contraction):
  // mat(p, Na, Nb) == mat_c(point_coef(p), p, Na, Nb)
  static constexpr int NCOEF = 2;
  static constexpr bool HAS_MAT_UNIT = true;
  static __device__ __forceinline__ void point_coef(const PtView &p, double *c) { tau(p, c[0], c[1]); }
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na_, const double *Nb_, double *T) {
    double c[2]; point_coef(p, c);
    mat_c(c, p, Na_, Nb_, T);
  }
  // mat_c(c, p, e_F, Nb) for the unit test feature F (0: N, 1 + g: dN/dx_g), written out: only the entries block_mask(i,j) names
  // for F are set.  x * 0 does not fold under IEEE rules, so the generic call spends about four times these multiply-adds per
  // point on products with the zeros of e_F (band_pt.hpp: the fp64 VALU work shares the pipe with the MFMAs it feeds).
  template <int F>
  static __device__ __forceinline__ void mat_unit(const double *c, const PtView &p, const double *Nb_, double *T) {
    const double nu = p.prm[0], shift = p.shift, tauM = c[0], tauC = c[1];
    const double Nb = Nb_[0];
    const double S = shift * Nb + (p.u[0] * Nb_[1] + p.u[1] * Nb_[2] + p.u[2] * Nb_[3]);     // shift Nb + u . grad Nb
    if constexpr (F == 0) {
      T[0] = S; T[5] = S; T[10] = S;
      T[12] = Nb_[1]; T[13] = Nb_[2]; T[14] = Nb_[3];
    } else {
      constexpr int g = F - 1;
      const double a = tauM * p.u[g];
      const double Tii = nu * Nb_[1 + g] + a * S;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (i != j && i != g && j != g) continue;
          double t = (i == j) ? Tii : 0.0;
          if (j == g) t += nu * Nb_[1 + i];
          if (i == g) t += tauC * Nb_[1 + j];
          T[i * 4 + j] = t;
        }
#pragma unroll
      for (int i = 0; i < 3; ++i) T[i * 4 + 3] = (i == g) ? a * Nb_[1 + i] - Nb : a * Nb_[1 + i];
      T[12 + g] = tauM * S;
      T[15] = tauM * Nb_[1 + g];
    }
  }
  // band_pt.hpp: a fifth test feature, the advective derivative u . grad N_a, makes the momentum-pressure blocks rank 2 instead of 3:
  // T_i3 = -dN_a/dx_i N_b + tauM (u . grad N_a) dN_b/dx_i.  39 (block, feature) products per k-step instead of 42.
  // The diagonal momentum blocks share all but one product: T_ii = D + (nu + tauC) dN_a/dx_i dN_b/dx_i.  The rows come in two groups,
  // {0, 1} and {2, 3}; in the first one block (1,1) accumulates only T_11 - T_00 = (nu + tauC) (d_y N_a d_y N_b - d_x N_a d_x N_b) -- two
  // products instead of four -- and band_combine() adds block (0,0) to it when the sums are complete: 37 products per k-step.
  static constexpr int BAND_NFEAT = 5;
  static constexpr unsigned band_block_mask(int i, int j) {
    return (i < 3 && j == 3) ? ((1u << (1 + i)) | (1u << 4)) : ((i == 1 && j == 1) ? 0x6u : block_mask(i, j));
  }
  template <int F>
  static __device__ __forceinline__ void mat_unit5(const double *c, const PtView &p, const double *Nb_, double *T) {
    if constexpr (F == 4) {
#pragma unroll
      for (int i = 0; i < 3; ++i) T[i * 4 + 3] = c[0] * Nb_[1 + i];
    } else {
      mat_unit<F>(c, p, Nb_, T);
      if constexpr (F >= 1) T[(F - 1) * 4 + 3] = -Nb_[0];      // (the tauM u_g dN_b/dx_i part of every row went to feature 4)
      if constexpr (F == 1) T[5] = -(p.prm[0] + c[1]) * Nb_[1];
      if constexpr (F == 2) T[5] = (p.prm[0] + c[1]) * Nb_[2];
    }
  }
  // acc: the block entries (i - I0) * 4 + j of one band tile for the row fields I0, I0 + 1
  template <int I0, class V, int N>
  static __device__ __forceinline__ void band_combine(V (&acc)[N]) { if constexpr (I0 == 0) acc[1 * 4 + 1] += acc[0]; }
  // band_pt.hpp, round 4: a wave holds ALL 16 entries of a block, so the part the three diagonal momentum blocks share,
  // D = N_a S + grad N_a . (nu grad N_b) + tauM (u . grad N_a) S with S = shift N_b + u . grad N_b, is accumulated ONCE, in a 17th
  // accumulator, and added to them when the sums are complete (band_finish).  Products per k-step = the ranks of what is left:
  // D 4 (features N, d_g: nu d_g N_b + tauM u_g S), T_ii - D 1 each, T_ij 2 each, T_i3 2 each (d_i and the advective feature),
  // T_3j 2 each, T_33 3: 4 + 3 + 12 + 6 + 6 + 3 = 34 (round 3: 37 with the row fields in two wave groups).
  // What a constant or a sign can do is left to band_finish as well (fp64 VALU work shares the pipe with the MFMAs): the momentum
  // blocks are summed as T_ij / nu -- d_j N_a d_i N_b + (tauC / nu) d_i N_a d_j N_b -- and the momentum-pressure blocks as -T_i3, with
  // the advective feature negated: 52 instead of 64 multiply-adds per k-step.
  static constexpr int BAND_NACC = 17, BAND_NCOEF = 3;
  static constexpr bool BAND_NEG_FEAT5 = true;
  // (the scaling by 1 / nu; an inviscid flow stays on the feature kernel.  Host AND device: the launcher of a built-in struct calls
  //  it directly, a run-time struct's is evaluated by a one-lane kernel of its module -- rtc.hpp)
  __host__ __device__ static bool band_params_ok(const double *prm) { return prm[0] > 0.0; }
  static __device__ __forceinline__ void band_coef(const PtView &p, double *c) { double tM, tC; tau(p, tM, tC); c[0] = tM; c[1] = tC / p.prm[0]; c[2] = 1.0 + c[1]; }
  static constexpr unsigned band_acc_mask(int n) {
    if (n == 16) return 0xFu;
    const int i = n / 4, j = n % 4;
    if (i < 3 && j < 3) return (i == j) ? (1u << (1 + i)) : ((1u << (1 + i)) | (1u << (1 + j)));
    if (i < 3) return (1u << (1 + i)) | (1u << 4);
    if (j < 3) return 0x1u | (1u << (1 + j));
    return 0xEu;
  }
  // the trial-side value of accumulator n for the unit test feature F (0: N, 1 + g: d_g N, 4: -(u . grad N)); Nb_ carries JW
  template <int F>
  static __device__ __forceinline__ void mat_acc(const double *c, const PtView &p, const double *Nb_, double *T) {
    const double nu = p.prm[0], tauM = c[0], rC = c[1], rC1 = c[2];
    const double Nb = Nb_[0];
    const double S = p.shift * Nb + (p.u[0] * Nb_[1] + p.u[1] * Nb_[2] + p.u[2] * Nb_[3]);
    if constexpr (F == 0) {
      T[16] = S; T[12] = Nb_[1]; T[13] = Nb_[2]; T[14] = Nb_[3];
    } else if constexpr (F == 4) {
      T[3] = tauM * Nb_[1]; T[7] = tauM * Nb_[2]; T[11] = tauM * Nb_[3];
    } else {
      constexpr int g = F - 1;
      const double tS = tauM * S;
      T[16] = nu * Nb_[1 + g] + p.u[g] * tS;
      T[g * 4 + g] = rC1 * Nb_[1 + g];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k == g) continue;
        T[g * 4 + k] = rC * Nb_[1 + k];        // block (g, k), test feature d_g: (tauC / nu) d_g N_a d_k N_b
        T[k * 4 + g] = Nb_[1 + k];             // block (k, g), test feature d_g: d_g N_a d_k N_b
      }
      T[g * 4 + 3] = Nb;
      T[12 + g] = tS;
      T[15] = tauM * Nb_[1 + g];
    }
  }
  template <class V, int N>
  static __device__ __forceinline__ void band_finish(V (&acc)[N], const double *prm) {
    const double nu = prm[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i * 4 + j] *= nu;
      acc[i * 4 + i] += acc[16];
      acc[i * 4 + 3] = -acc[i * 4 + 3];
    }
  }
  static __device__ __forceinline__ void mat_c(const double *c, const PtView &p, const double *Na_, const double *Nb_, double *T) {
    const double nu = p.prm[0], shift = p.shift;
    const double tauM = c[0], tauC = c[1];
    const double ux = p.u[0], uy = p.u[1], uz = p.u[2];
    const double Na = Na_[0], Na_x = Na_[1], Na_y = Na_[2], Na_z = Na_[3];
    const double Nb = Nb_[0], Nb_x = Nb_[1], Nb_y = Nb_[2], Nb_z = Nb_[3];
    const double adva = ux * Na_x + uy * Na_y + uz * Na_z, advb = ux * Nb_x + uy * Nb_y + uz * Nb_z;
    const double Tii = (+shift * Na * Nb + Na * advb + nu * (Na_x * Nb_x + Na_y * Nb_y + Na_z * Nb_z) + tauM * adva * (shift * Nb + advb));
    T[0] = nu * Na_x * Nb_x + tauC * Na_x * Nb_x + Tii;
    T[1] = nu * Na_y * Nb_x + tauC * Na_x * Nb_y;
    T[2] = nu * Na_z * Nb_x + tauC * Na_x * Nb_z;
    T[4] = nu * Na_x * Nb_y + tauC * Na_y * Nb_x;
    T[5] = nu * Na_y * Nb_y + tauC * Na_y * Nb_y + Tii;
    T[6] = nu * Na_z * Nb_y + tauC * Na_y * Nb_z;
    T[8] = nu * Na_x * Nb_z + tauC * Na_z * Nb_x;
    T[9] = nu * Na_y * Nb_z + tauC * Na_z * Nb_y;
    T[10] = nu * Na_z * Nb_z + tauC * Na_z * Nb_z + Tii;
    T[3] = -Na_x * Nb + tauM * adva * Nb_x;
    T[7] = -Na_y * Nb + tauM * adva * Nb_y;
    T[11] = -Na_z * Nb + tauM * adva * Nb_z;
    T[12] = +Na * Nb_x + tauM * Na_x * (shift * Nb + advb);
    T[13] = +Na * Nb_y + tauM * Na_y * (shift * Nb + advb);
    T[14] = +Na * Nb_z + tauM * Na_z * (shift * Nb + advb);
    T[15] = +tauM * (Na_x * Nb_x + Na_y * Nb_y + Na_z * Nb_z);
  }
};


// ---- forms with a boundary branch (`if (p->atboundary)` in the reference's callback): bmat / bvec are the
// un-weighted integrands at a point of a visited face; JW there is detJac * weight * detS.
template <class F, class = void> struct has_boundary_of { static constexpr bool v = false; };
template <class F> struct has_boundary_of<F, decltype((void)F::HAS_BOUNDARY)> { static constexpr bool v = F::HAS_BOUNDARY; };

// demo/BoundaryIntegral.c:26-56: Laplace inside (F = 0), Neumann data 1 on the visited faces
template <int DIM> struct FormBoundaryIntegral {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr bool HAS_BOUNDARY = true;
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) { FormPoisson<DIM>::mat(p, Na, Nb, T); }
  static __device__ __forceinline__ void vec(const PtView &, const double *, double *R) { R[0] = 0.0; }
  static __device__ __forceinline__ void bmat(const PtView &, const double *, const double *, double *T) { T[0] = 0.0; }
  static __device__ __forceinline__ void bvec(const PtView &, const double *Na, double *R) { R[0] = Na[0] * 1.0; }
};

// demo/NitscheMethod.c:69-110: Poisson with u = sum x_i^2 imposed weakly on the visited faces; params {k = max degree}
template <int DIM> struct FormNitsche {
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = NEED_X | NEED_G;
  static constexpr bool HAS_BOUNDARY = true;
  static __device__ __forceinline__ void mat(const PtView &p, const double *Na, const double *Nb, double *T) { FormPoisson<DIM>::mat(p, Na, Nb, T); }
  static __device__ __forceinline__ void vec(const PtView &, const double *Na, double *R) { R[0] = Na[0] * (-2.0 * DIM); }
  static __device__ __forceinline__ double alpha(const PtView &p) {   // C/h, h = NormalMeshSize (:57-66)
    double s = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) { double Ni = 0; for (int j = 0; j < DIM; ++j) Ni += p.G[i * DIM + j] * p.normal[j]; s += Ni * Ni; }
    const double h = 2 / sqrt(s), C = 5 * (p.prm[0] + 1);
    return C / h;
  }
  static __device__ __forceinline__ void bmat(const PtView &p, const double *Na, const double *Nb, double *T) {
    double dna = 0, dnb = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) { dna += Na[1 + i] * p.normal[i]; dnb += Nb[1 + i] * p.normal[i]; }
    T[0] = -Na[0] * dnb - Nb[0] * dna + alpha(p) * Na[0] * Nb[0];
  }
  static __device__ __forceinline__ void bvec(const PtView &p, const double *Na, double *R) {
    double g = 0, dna = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) { g += p.x[i] * p.x[i]; dna += Na[1 + i] * p.normal[i]; }
    R[0] = -dna * g + alpha(p) * Na[0] * g;
  }
};

// ---- scalar functionals: the point callbacks handed to IGAComputeScalar (src/petigacomp.c:35-98).  scalar() returns
// the integrand values S[NSCALAR] at one point; the kernel multiplies by JW and sums (IGAPointAddArray, petigapoint.c:461).
template <class F, class = void> struct nscalar_of { static constexpr int v = 0; };
template <class F> struct nscalar_of<F, decltype((void)F::NSCALAR)> { static constexpr int v = F::NSCALAR; };

// src/petigacomp.c:102-120 (ErrorSqr) with test/IGAErrNorm.c:26-52 as Exact: fields 1, sum x, sum x^2, prod x;
// params {k}: k = 0 values, 1 gradients, 2 Hessians.  A null U gives the norms of the exact fields.
template <int DIM, bool SECOND_> struct ScalarErrNorm {
  static constexpr int DOF = 4, ORDER = SECOND_ ? 2 : 1, NSCALAR = 4;
  static constexpr unsigned NEED = NEED_X | NEED_U | NEED_GU | (SECOND_ ? NEED_HU : 0u);
  static __device__ __forceinline__ void scalar(const PtView &p, double *S) {
    const int order = (int)p.prm[0];
    double s1 = 0, s2 = 0, pr = 1;
#pragma unroll
    for (int i = 0; i < DIM; ++i) { s1 += p.x[i]; s2 += p.x[i] * p.x[i]; pr *= p.x[i]; }
    S[0] = S[1] = S[2] = S[3] = 0;
    if (order == 0) {
      const double ve[4] = {1, s1, s2, pr};
#pragma unroll
      for (int c = 0; c < 4; ++c) { const double e = fabs(ve[c] - p.u[c]); S[c] += e * e; }
    } else if (order == 1) {
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double ve[4] = {0, 1, 2 * p.x[i], pr / p.x[i]};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const double e = fabs(ve[c] - p.gu[c * DIM + i]); S[c] += e * e; }
      }
    } else if (SECOND_) {
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          const double ve[4] = {0, 0, (i == j) ? 2.0 : 0.0, (i == j) ? 0.0 : pr / (p.x[i] * p.x[j])};
#pragma unroll
          for (int c = 0; c < 4; ++c) { const double e = fabs(ve[c] - p.hu[c * DIM * DIM + i * DIM + j]); S[c] += e * e; }
        }
    }
  }
};

// test/IGAFixTable.c:66-72 (Exact = sum x_i^2) through ErrorSqr, one field
template <int DIM> struct ScalarX2Err {
  static constexpr int DOF = 1, ORDER = 1, NSCALAR = 1; static constexpr unsigned NEED = NEED_X | NEED_U;
  static __device__ __forceinline__ void scalar(const PtView &p, double *S) {
    double g = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) g += p.x[i] * p.x[i];
    const double e = fabs(g - p.u[0]); S[0] = e * e;
  }
};

// test/IGAGeometryMap.c:383-389 (Scalar): S[0] volume of the mapped domain (interior pass), S[1] area of the visited
// faces (boundary passes); any dof
template <int DIM> struct ScalarVolume {
  static constexpr int DOF = 1, ORDER = 1, NSCALAR = 2; static constexpr unsigned NEED = 0;
  static __device__ __forceinline__ void scalar(const PtView &p, double *S) { S[0] = p.atboundary ? 0.0 : 1.0; S[1] = p.atboundary ? 1.0 : 0.0; }
};

}  // namespace igx
