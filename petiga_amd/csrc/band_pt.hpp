// band_pt.hpp -- band rows by node layer for multi-field forms with POINT-DEPENDENT coefficients: the Tangent of
// demo/NavierStokesVMS.c:166-244 (BASELINE config 5: p = 3, 4 fields, rational NURBS geometry, axes 0 and 2 periodic).
//
// block_pencil.hpp turns the element loop inside out for constant-coefficient forms (Gram pairs, identity geometry).  The same
// walk for a form whose matrix integrand depends on the point -- through the state, the metric, the geometry -- needs two more
// things, and nothing else changes (layer-major, output stationary, LDS stage in matrix order, wave-coalesced read-add-write):
//
//   1. the operands are the PHYSICAL features of the basis functions at the Gauss point, built per k-step in registers from the
//      three 1-D rows (tensor product), the point's rational data and its inverse Jacobian (src/petigarat.f90.in,
//      petigamapshf.f90.in:30-58), and the B operand of a block (i,j) is the form's own mat_c() applied to the unit test feature
//      (as in feature_mfma.hpp: K_e^{ij} = A^T B^{ij}; block_mask(i,j) keeps structurally zero products off the matrix cores);
//   2. what depends on the point alone -- JW, F^-1, 1/W and dW/W, the state u, the form's point coefficients (tau_M, tau_C of
//      the VMS model: Form::point_coef) -- is tabulated ONCE per element by a small kernel ahead of the launch (band_points: one
//      wavefront per element, lane = Gauss point, sums over the control points factorised across the lanes as in
//      gram_mfma.hpp) and travels through a ring of five element records in LDS.  (fp64 VALU work next to another wave's MFMA
//      stream costs about one MFMA per instruction: a tabulation inside this kernel's flush phase would cost more than the
//      contraction it feeds.)
//
// Round 4 layout (dof = 4: a block is 128 bytes).  A wavefront holds ONE band tile at a time with ALL 16 entries of its blocks
// (16, or 17 with the part the diagonal momentum blocks share, accumulators of 8 registers): the features of a k-step are built
// once (round 3 built them in two wave groups, one per pair of row fields), the shared part of the diagonal blocks is summed once
// (34 instead of 37 products per k-step), and a lane ends up with whole 128-byte blocks -- its read-add-write goes from the
// accumulators straight to the matrix, one cache line per (lane, row slot): no LDS stage, no arrival counters.  Workgroup = one
// pencil, FOUR wavefronts (band tiles d = {0}, {+1,-3}, {-1,+3}, {+2,-2}: four tile products each), two workgroups per CU: the two
// wavefronts of a SIMD belong to different pencils and nothing synchronises them -- while one waits for its old values the
// other one streams MFMAs.  The next element's record enters the ring through global_load_lds (no registers, one barrier per
// layer).
// Periodic axes wrapped inside the rank (config 5 on one GPU) are taken on the walk axis (layers and elements modulo nel) and on
// axis 2 (its column positions come from the per-axis table); axis 1 must have consecutive positions.
#pragma once
#include "block_pencil.hpp"
#ifndef IGX_RTC
#include <climits>
#endif

namespace igx {

template <class F, class = void> struct has_mat_unit { static constexpr bool v = false; };
template <class F> struct has_mat_unit<F, decltype((void)F::HAS_MAT_UNIT)> { static constexpr bool v = F::HAS_MAT_UNIT; };
// BAND_NFEAT = 5: the form adds the advective derivative u . grad N as a fifth test feature (band_block_mask, mat_unit5: forms.hpp)
template <class F, class = void> struct band_nfeat_of { static constexpr int v = 4; };
template <class F> struct band_nfeat_of<F, decltype((void)F::BAND_NFEAT)> { static constexpr int v = F::BAND_NFEAT; };
template <class Form> __host__ __device__ constexpr unsigned bpt_mask(int i, int j) {
  if constexpr (band_nfeat_of<Form>::v == 5) return Form::band_block_mask(i, j); else return fm_block_mask<Form>(i, j);
}

// BAND_NACC / band_acc_mask / mat_acc / band_finish: the form names its accumulators itself (more than dof^2 when blocks share
// a part: forms.hpp, FormNSVMS); otherwise one accumulator per block entry with the masks above
template <class F, class = void> struct band_nacc_of { static constexpr int v = F::DOF * F::DOF; static constexpr bool own = false; };
template <class F> struct band_nacc_of<F, decltype((void)F::BAND_NACC)> { static constexpr int v = F::BAND_NACC; static constexpr bool own = true; };
template <class Form> __host__ __device__ constexpr unsigned bpt_acc_mask(int n) {
  if constexpr (band_nacc_of<Form>::own) return Form::band_acc_mask(n); else return bpt_mask<Form>(n / Form::DOF, n % Form::DOF);
}

// element record (doubles): 64 points x NPD, then the element's walk-axis rows [q][a][2], then its 64 NURBS weights; padded to
// whole KB (global_load_lds moves 64 lanes x 16 bytes per instruction).  Per point: JW | the map from the parametric value and
// gradient (n, d_0, d_1, d_2) of w_a N_a to the physical ones: R = g0 n, d_i R = h_i n + sum_b Gm[i][b] d_b, with g0 = 1/W,
// Gm[i][b] = du_b/dx_i / W, h_i = -sum_b Gm[i][b] dW_b/W (Rationalize + ShapeFunctions, src/petigarat.f90.in:3-57,
// petigamapshf.f90.in:30-58, in one 4 x 4 matrix) | u | the form's point coefficients
template <class F, class = void> struct band_ncoef_base { static constexpr int v = 0; };      // (a constant-coefficient form has none)
template <class F> struct band_ncoef_base<F, decltype((void)F::NCOEF)> { static constexpr int v = F::NCOEF; };
template <class F, class = void> struct band_ncoef_of { static constexpr int v = band_ncoef_base<F>::v; };
template <class F> struct band_ncoef_of<F, decltype((void)F::BAND_NCOEF)> { static constexpr int v = F::BAND_NCOEF; };      // (band_coef instead of point_coef)
// Round 6: constant-coefficient forms with 2 or 3 fields (MAT_PAIR_MASK: demo/Elasticity3D.c) on a MAPPED geometry.  block_pencil.hpp
// builds their Gram operands from the three 1-D rows -- identity geometry only; here the operands are the PHYSICAL features of the
// point records, the accumulators are the Gram pairs M_fg = sum_q JW d_f N_a d_g N_b (one MFMA per pair and k-step) and the
// coefficient transform K^{ij} = sum_fg C^{ij}_fg M_fg is block_pencil's own (bp_transform), lane-local at the end of a tile's
// products; a lane then holds whole blocks of dof^2 values and adds them to the matrix itself, 72 contiguous bytes at dof 3.
#ifndef BPT_PREFETCH
#define BPT_PREFETCH 1
#endif
template <class Form> constexpr bool bpt_pairs() { return mat_pair_mask_of<Form>::v != 0ull && !has_point_coef<Form>::v; }
template <class Form> constexpr int bpt_nacc() { if constexpr (bpt_pairs<Form>()) return bp_nacc<Form>(); else return band_nacc_of<Form>::v; }
template <class F, class = void> struct band_neg5_of { static constexpr bool v = false; };
template <class F> struct band_neg5_of<F, decltype((void)F::BAND_NEG_FEAT5)> { static constexpr bool v = F::BAND_NEG_FEAT5; };
template <class Form> constexpr int bpt_npd() { return 1 + 13 + Form::DOF + band_ncoef_of<Form>::v; }
template <class Form> constexpr int bpt_rec() { return (64 * bpt_npd<Form>() + 32 + 64 + 127) / 128 * 128; }

struct BandArgs {
  int ex_start, ex_step, ex_count, ey_start, ey_step, ey_count;
  int nel0, alias0;                  // elements on axis 0; axis 0 periodic and wrapped inside the rank
  int seg_len, nseg;                 // node layers per segment
  int first_touch, nelx, nely, fty_lo, fty_hi, fty_blocked;
  int wrap2;                         // axis 2 periodic and wrapped inside the rank (first-touch rule without clipping)
  double *pts;                       // element records of this launch: [pencil][element on axis 0][bpt_rec]
  int prio_layers, rmw_prio;         // schedule switches (IGX_BAND_PRIO, IGX_BAND_RMW_PRIO): s_setprio 2 through a workgroup's first layers; s_setprio 3 through a read-add-write
  int debug, dbg_block;
  long long *dbg_buf;                // -DIGX_DEBUG builds, IGX_DEBUG_TIMING: cycle stamps [workgroup][wave][layer][6]
};

// ---- the point tabulation: one wavefront per element
// (P = 2: the element's 27 functions and points sit in the 4 x 4 x 4 lane slots, the padding carries zeros -- like the headline kernel at p = 2)
template <class Form, int P = 3>
__global__ void __launch_bounds__(256)
band_points(SpaceDev S, ParamsDev prm, OutDev out, BandArgs pa) {
  constexpr int NB = P + 1, DOF = Form::DOF, NC = 4 + DOF, NPD = bpt_npd<Form>(), REC = bpt_rec<Form>();
  static_assert(P == 2 || P == 3, "degrees 2 and 3");
  __shared__ double sm_all[4][64 * NC + 128 + 192 + 96];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long elem = (long long)blockIdx.x * 4 + wave;
  const int npen = pa.ex_count * pa.ey_count;
  if (elem >= (long long)npen * pa.nel0) return;
  double *coef = sm_all[wave], *T1 = coef + 64 * NC, *T2 = T1 + 128, *uxr = T2 + 192, *vyr = uxr + 32, *ztg = vyr + 32;
  const int pencil = (int)(elem / pa.nel0), e0 = (int)(elem - (long long)pencil * pa.nel0);
  const int tx = pencil % pa.ex_count, ty = pencil / pa.ex_count;
  const int elx = pa.ex_start + tx * pa.ex_step, ely = pa.ey_start + ty * pa.ey_step;
  const AxisDev &AW = S.ax[0], &AX = S.ax[1], &AY = S.ax[2];
  const int off0 = AW.off[e0], offx = AX.off[elx], offy = AY.off[ely];
  const bool geo = S.nsd > 0, rat = S.rational != 0;
  {   // control points (homogeneous) and state of the lane's basis function (aw, ay, ax) = (lane >> 4, (lane >> 2) & 3, lane & 3)
    const int aw0 = lane >> 4, ay0 = (lane >> 2) & 3, ax0 = lane & 3;
    const bool vn = aw0 < NB && ay0 < NB && ax0 < NB;      // a basis function of the element (not a padding slot)
    const int aw = vn ? aw0 : 0, ay = vn ? ay0 : 0, ax = vn ? ax0 : 0;
    const size_t g = (size_t)(off0 + aw) + (size_t)AW.gwidth * ((size_t)(offx + ax) + (size_t)AX.gwidth * (size_t)(offy + ay));
    const double w = vn ? (rat ? S.W[g] : 1.0) : 0.0;
    double c[NC];
    c[0] = (geo && vn) ? S.X[g * 3 + 0] * w : 0.0; c[1] = (geo && vn) ? S.X[g * 3 + 1] * w : 0.0; c[2] = (geo && vn) ? S.X[g * 3 + 2] * w : 0.0; c[3] = w;
    const size_t row = (size_t)AW.rowmap[off0 + aw] + (size_t)AW.nrow * ((size_t)AX.rowmap[offx + ax] + (size_t)AX.nrow * (size_t)AY.rowmap[offy + ay]);
    // IGAElementFixValues (src/petigaelem.c:1327-1358): the state at a Dirichlet dof is the boundary value
    const int el3[3] = {e0, elx, ely}, aa[3] = {aw, ax, ay};
#pragma unroll
    for (int f = 0; f < DOF; ++f) {
      double u = out.U ? out.U[row * DOF + f] : 0.0;
      for (int d = 0; d < 3; ++d) {
        const AxisDev &A = S.ax[d];
        if (A.periodic) continue;
        for (int sd = 0; sd < 2; ++sd) {
          if (el3[d] + A.estart != (sd ? A.esizes - 1 : 0) || aa[d] != (sd ? P : 0)) continue;
          const BCDev &bv = S.bcv[d][sd];
          for (int k = 0; k < bv.count; ++k) if (bv.field[k] == f) u = S.fixtable ? S.fixtable[row * DOF + f] : bv.value[k];
        }
      }
      c[4 + f] = u * w;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) coef[lane * NC + k] = c[k];
    if (lane < 32) {
      const int q = lane >> 3, a = (lane >> 1) & 3, k = lane & 1;
      const bool ok = q < NB && a < NB;
      uxr[lane] = ok ? AX.tab[((size_t)elx * NB * NB + q * NB + a) * NDER + k] : 0.0;   // [q][a][2]
      ztg[lane] = ok ? AW.tab[((size_t)e0 * NB * NB + q * NB + a) * NDER + k] : 0.0;    // [q][a][2]
    } else {
      const int l2 = lane - 32, a = l2 >> 3, q = (l2 >> 1) & 3, k = l2 & 1;
      vyr[l2] = (q < NB && a < NB) ? AY.tab[((size_t)ely * NB * NB + q * NB + a) * NDER + k] : 0.0;     // [a][q][2]
    }
  }
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // H[c][k] = sum_a C_a[c] D_k N_a(q), k = (value, d/du0, d/du1, d/du2): sum factorisation across the lanes (gram_mfma.hpp)
  double H[NC][4];
  const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
  double zv[4], zd[4];
#pragma unroll
  for (int aw = 0; aw < 4; ++aw) { zv[aw] = ztg[(i2 * 4 + aw) * 2 + 0]; zd[aw] = ztg[(i2 * 4 + aw) * 2 + 1]; }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    {
      double tv = 0, td = 0;
#pragma unroll
      for (int ax = 0; ax < 4; ++ax) { const double C = coef[((i2 * 4 + i1) * 4 + ax) * NC + c]; tv += C * uxr[(i0 * 4 + ax) * 2 + 0]; td += C * uxr[(i0 * 4 + ax) * 2 + 1]; }
      T1[((0 * 4 + i1) * 4 + i2) * 4 + i0] = tv; T1[((1 * 4 + i1) * 4 + i2) * 4 + i0] = td;
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
      double m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
      for (int ay = 0; ay < 4; ++ay) {
        const double a = T1[((0 * 4 + ay) * 4 + i2) * 4 + i0], d = T1[((1 * 4 + ay) * 4 + i2) * 4 + i0];
        const double yv = vyr[(ay * 4 + i1) * 2 + 0], yd = vyr[(ay * 4 + i1) * 2 + 1];
        m0 += a * yv; m1 += d * yv; m2 += a * yd;
      }
      T2[((0 * 4 + i2) * 4 + i1) * 4 + i0] = m0; T2[((1 * 4 + i2) * 4 + i1) * 4 + i0] = m1; T2[((2 * 4 + i2) * 4 + i1) * 4 + i0] = m2;
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
      double h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
      for (int aw = 0; aw < 4; ++aw) {
        const double t0 = T2[((0 * 4 + aw) * 4 + i1) * 4 + i0], t1 = T2[((1 * 4 + aw) * 4 + i1) * 4 + i0], t2 = T2[((2 * 4 + aw) * 4 + i1) * 4 + i0];
        h0 += t0 * zv[aw]; h1 += t0 * zd[aw]; h2 += t1 * zv[aw]; h3 += t2 * zv[aw];
      }
      H[c][0] = h0; H[c][1] = h1; H[c][2] = h2; H[c][3] = h3;
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  // lane = Gauss point (qx, qy, qw) = (i0, i1, i2); record index = lane
  double rec[NPD];
  {
    const bool vq = i0 < NB && i1 < NB && i2 < NB;      // a Gauss point of the element (a padding slot keeps finite numbers and JW = 0)
    const double iw = vq ? 1.0 / H[3][0] : 1.0;
    double E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, det = 1.0;
    if (geo && vq) {
      double F[3][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double x = H[c][0] * iw;
#pragma unroll
        for (int b = 0; b < 3; ++b) F[c][b] = (H[c][1 + b] - x * H[3][1 + b]) * iw;
      }
      det = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) + F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
      if (!(det > 0.0)) atomicExch(out.errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
      const double id = 1.0 / det;
      E[0][0] = (F[1][1] * F[2][2] - F[1][2] * F[2][1]) * id; E[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * id; E[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * id;
      E[1][0] = (F[1][2] * F[2][0] - F[1][0] * F[2][2]) * id; E[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * id; E[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * id;
      E[2][0] = (F[1][0] * F[2][1] - F[1][1] * F[2][0]) * id; E[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * id; E[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * id;
    }
    const double Jw = AW.J[e0], Jx = AX.J[elx], Jy = AY.J[ely];
    rec[0] = vq ? det * (AW.w[e0 * NB + i2] * Jw) * (AX.w[elx * NB + i0] * Jx) * (AY.w[ely * NB + i1] * Jy) : 0.0;
    const double g0 = rat ? iw : 1.0;
    rec[1] = g0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double h = 0.0;
#pragma unroll
      for (int b = 0; b < 3; ++b) { const double gm = g0 * E[b][i]; rec[2 + i * 3 + b] = gm; h -= gm * (H[3][1 + b] * iw); }
      rec[11 + i] = (rat && vq) ? h : 0.0;
    }
    double u[DOF];
#pragma unroll
    for (int f = 0; f < DOF; ++f) { u[f] = H[4 + f][0] * iw; rec[14 + f] = u[f]; }
    // IGAPointFormInvGradGeomMap (src/petigapoint.c:269-294): G[a][i] = du_a/dx_i / (half length of the element on axis a)
    double G[9];
    const double L3[3] = {Jw, Jx, Jy};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < 3; ++i) G[a * 3 + i] = E[a][i] / L3[a];
    PtView p; p.x = nullptr; p.u = u; p.ut = nullptr; p.gu = nullptr; p.hu = nullptr; p.G = G; p.prm = prm.v; p.shift = out.shift; p.t = out.t;
    p.normal = nullptr; p.atboundary = 0; p.boundary_id = -1;
    if constexpr (band_nacc_of<Form>::own) Form::band_coef(p, rec + 14 + DOF); else if constexpr (has_point_coef<Form>::v) Form::point_coef(p, rec + 14 + DOF);
  }
  double *dst = pa.pts + (size_t)elem * REC;
#pragma unroll
  for (int k = 0; k < NPD; ++k) dst[lane * NPD + k] = rec[k];
  if (lane < 32) dst[64 * NPD + lane] = ztg[lane];
  dst[64 * NPD + 32 + lane] = coef[lane * NC + 3];      // weight of control point (aw, ay, ax) = lane
  if (64 * NPD + 96 + lane < REC) dst[64 * NPD + 96 + lane] = 0.0;      // (the padding travels into LDS: keep it defined)
}

// offsets (doubles) into the dynamic LDS block of band_pt
struct BptCarve { int ring, pre, cnt, rho, P, pen, uv, bc, total; };
__host__ __device__ static inline BptCarve bpt_carve(int rec, int seg_len) {
  BptCarve c; int pos = 0;
  auto take = [&](int n) { const int o = pos; pos += (n + 1) & ~1; return o; };
  c.ring = take(5 * rec);
  c.pre = take(seg_len); c.cnt = take((seg_len + 1) / 2); c.rho = take((seg_len + 1) / 2); c.P = take(seg_len * 4);
  c.pen = take(64); c.uv = take(64); c.bc = take(6 + 6 * 4);
  c.total = pos;
  return c;
}

// ring slot of the element with the (unwrapped) index eu = layer - slot: five slots, a window holds four
__device__ __forceinline__ int bpt_slot(int eu) { const int m = eu % 5; return m < 0 ? m + 5 : m; }

// the MFMAs of one test feature F at one k-step: the trial-side values of the accumulators whose mask names F
template <class Form, int F>
__device__ __forceinline__ void bpt_feature(d4_t (&acc)[bpt_nacc<Form>()], const double *cf, const PtView &p, const double (&na)[5], const double (&nb)[4]) {
  constexpr int NACC = bpt_nacc<Form>();
  double T[NACC];
  if constexpr (band_nacc_of<Form>::own) Form::template mat_acc<F>(cf, p, nb, T);
  else if constexpr (band_nfeat_of<Form>::v == 5) Form::template mat_unit5<F>(cf, p, nb, T);
  else if constexpr (has_mat_unit<Form>::v) Form::template mat_unit<F>(cf, p, nb, T);
  else {
    double ef[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ef[g] = (g == F) ? 1.0 : 0.0;
    Form::mat_c(cf, p, ef, nb, T);
  }
#pragma unroll
  for (int n = 0; n < NACC; ++n) {
    if (!((bpt_acc_mask<Form>(n) >> F) & 1u)) continue;
    acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[F], T[n], acc[n], 0, 0, 0);
  }
}

// one tile product: acc[n] += A_F(e, ta)^T B^n_F(e, tb) over the element's 64 points; k-step (qw, qy), k slot qx = lane >> 4
template <class Form, bool GEO, bool RAT, int NB>
__device__ __forceinline__ void bpt_product(d4_t (&acc)[bpt_nacc<Form>()], const double *rec, int ta, int tb, const double (&uxy)[4][3], int lane, const double *prm, double shift) {
  constexpr int DOF = Form::DOF, NPD = bpt_npd<Form>();
  const double *zt = rec + 64 * NPD, *wts = zt + 32;
  const int qx = lane >> 4;
  const double wa = RAT ? wts[ta * 16 + (lane & 15)] : 1.0, wb = RAT ? wts[tb * 16 + (lane & 15)] : 1.0;
#pragma unroll 1
  for (int qw = 0; qw < NB; ++qw) {
    // the walk-axis rows of the row (A) and column (B) basis function, with the NURBS weight of the control point
    const double zA0 = zt[(qw * 4 + ta) * 2 + 0] * wa, zA1 = zt[(qw * 4 + ta) * 2 + 1] * wa;
    const double zB0 = zt[(qw * 4 + tb) * 2 + 0] * wb, zB1 = zt[(qw * 4 + tb) * 2 + 1] * wb;
#pragma unroll
    for (int qy = 0; qy < NB; ++qy) {
      const double *pd = rec + ((qw * 4 + qy) * 4 + qx) * NPD;
      const double jw = pd[0];
      // value and gradient of the row (A) and column (B) basis function at the point (the trial side carries JW: mat() is linear
      // in it).  Parametric: n = z0 (u0 v0), d/du0 = z1 (u0 v0), d/du1 = z0 (u1 v0), d/du2 = z0 (u0 v1).  On a geometry the map to
      // the physical ones is the point's matrix (band_points): d_i = z1 P_i + z0 Q_i with P_i = Gm[i][0] u0v0 and
      // Q_i = Gm[i][1] u1v0 + Gm[i][2] u0v1 + h_i u0v0 -- P, Q serve both sides
      const double zj0 = zB0 * jw, zj1 = zB1 * jw;
      double na[5], nb[4];
      if (GEO) {
        const double n0 = RAT ? pd[1] * uxy[qy][0] : uxy[qy][0];
        na[0] = zA0 * n0; nb[0] = zj0 * n0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double Pi = pd[2 + i * 3 + 0] * uxy[qy][0];
          double Qi = pd[2 + i * 3 + 1] * uxy[qy][1] + pd[2 + i * 3 + 2] * uxy[qy][2];
          if (RAT) Qi += pd[11 + i] * uxy[qy][0];
          na[1 + i] = zA1 * Pi + zA0 * Qi; nb[1 + i] = zj1 * Pi + zj0 * Qi;
        }
      } else {
        na[0] = zA0 * uxy[qy][0]; na[1] = zA1 * uxy[qy][0]; na[2] = zA0 * uxy[qy][1]; na[3] = zA0 * uxy[qy][2];
        nb[0] = zj0 * uxy[qy][0]; nb[1] = zj1 * uxy[qy][0]; nb[2] = zj0 * uxy[qy][1]; nb[3] = zj0 * uxy[qy][2];
      }
      na[4] = 0.0;
      PtView p; p.x = nullptr; p.u = pd + 14; p.ut = nullptr; p.gu = nullptr; p.hu = nullptr; p.G = nullptr; p.prm = prm; p.shift = shift; p.t = 0.0;
      p.normal = nullptr; p.atboundary = 0; p.boundary_id = -1;
      if constexpr (bpt_pairs<Form>()) {      // the Gram pairs of a constant-coefficient form: one MFMA per (test feature f, trial feature g)
        constexpr unsigned long long PAIRS = mat_pair_mask_of<Form>::v;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (!((PAIRS >> (f * 8 + g)) & 1ull)) continue;
            acc[fm_pair_index(PAIRS, f, g)] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[f], nb[g], acc[fm_pair_index(PAIRS, f, g)], 0, 0, 0);
          }
      } else {
      const double *cf = pd + 14 + DOF;
      bpt_feature<Form, 0>(acc, cf, p, na, nb);
      bpt_feature<Form, 1>(acc, cf, p, na, nb);
      bpt_feature<Form, 2>(acc, cf, p, na, nb);
      bpt_feature<Form, 3>(acc, cf, p, na, nb);
      if constexpr (band_nfeat_of<Form>::v == 5) {      // the advective derivative of the test function (the state u is in the record)
        if constexpr (band_neg5_of<Form>::v) na[4] = (-pd[14]) * na[1] + (-pd[15]) * na[2] + (-pd[16]) * na[3];
        else na[4] = pd[14] * na[1] + pd[15] * na[2] + pd[16] * na[3];
        bpt_feature<Form, 4>(acc, cf, p, na, nb);
      }
      }
    }
  }
}

template <class Form, bool GEO, bool RAT, int P = 3>
__global__ void __launch_bounds__(256, 2)
band_pt(SpaceDev S, ParamsDev prm, OutDev out, BandArgs pa) {
  constexpr int NB = P + 1, BW = 2 * P + 1, DOF = Form::DOF, BS = DOF * DOF, NACC = bpt_nacc<Form>(), REC = bpt_rec<Form>();
  constexpr bool PAIRF = bpt_pairs<Form>();
  static_assert(P == 2 || P == 3, "degrees 2 and 3 (p = 2 in the 4 x 4 tile slots, zero padded)");
  static_assert((DOF == 4 && has_point_coef<Form>::v) || (PAIRF && (DOF == 2 || DOF == 3)),
                "128-byte blocks of a form that separates its point coefficients (NCOEF, point_coef, mat_c), or the Gram pairs of a constant-coefficient form with 2 or 3 fields");
  static_assert(NACC >= BS && NACC <= 20, "accumulators: the block entries first, shared parts behind them");
  extern __shared__ __attribute__((aligned(16))) double bpt_sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int role = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npen = pa.ex_count * pa.ey_count;
  const int seg = blockIdx.x / npen, pencil = blockIdx.x - seg * npen;
  const int tx = pencil % pa.ex_count, ty = pencil / pa.ex_count;
  const int elx = pa.ex_start + tx * pa.ex_step, ely = pa.ey_start + ty * pa.ey_step;
  const AxisDev &AW = S.ax[0], &AX = S.ax[1], &AY = S.ax[2];
  const bool alias0 = pa.alias0 != 0;
  const int NL = alias0 ? pa.nel0 : pa.nel0 + P;       // node layers (rows) of the pencil
  const int li_lo = seg * pa.seg_len, li_hi = min(li_lo + pa.seg_len, NL);
  const int nlay = li_hi - li_lo;
  const int lay_first = AW.off[0];
  const int offx = AX.off[elx], offy = AY.off[ely];
  const BptCarve cv = bpt_carve(REC, pa.seg_len);
  double *ring = bpt_sm + cv.ring, *uvs = bpt_sm + cv.uv;
  long long *Lpre = reinterpret_cast<long long *>(bpt_sm + cv.pre);
  int *Lcnt = reinterpret_cast<int *>(bpt_sm + cv.cnt), *LP = reinterpret_cast<int *>(bpt_sm + cv.P);
  BpPencil *pen = reinterpret_cast<BpPencil *>(bpt_sm + cv.pen);
  unsigned *bcm = reinterpret_cast<unsigned *>(bpt_sm + cv.bc);
  double *bcv = bpt_sm + cv.bc + 6;
  const double *recs = pa.pts + (size_t)pencil * pa.nel0 * REC;      // this pencil's element records
  const long long tw_wg = (kDebug && pa.dbg_buf) ? wall_clock64() : 0;

  // record of element eu (modulo nel on a wrapped axis) -> its ring slot: REC / 128 pieces of 1 KB, wave w moves the pieces
  // w, w + 4, ... (global_load_lds: 64 lanes x 16 bytes, no registers; done when vmcnt says so -- the barrier below)
  auto load_element = [&](int eu) {
    int e = eu;
    if (alias0) { e %= pa.nel0; if (e < 0) e += pa.nel0; } else if (e < 0 || e >= pa.nel0) return;
    const double *src = recs + (size_t)e * REC; double *dst = ring + bpt_slot(eu) * REC;
    for (int c = role; c < REC / 128; c += 4)
      __builtin_amdgcn_global_load_lds(src + c * 128 + lane * 2, (__attribute__((address_space(3))) void *)(dst + c * 128), 16, 0, 0);
  };
  {   // tables of the segment and of the pencil; the window of the first layer
    for (int i = tid; i < nlay; i += 256) {
      const int lay = lay_first + li_lo + i, rho = AW.rowmap[lay];
      Lcnt[i] = AW.rcnt[rho]; Lpre[i] = AW.prefix[rho];
      for (int d = 0; d < BW; ++d) LP[i * 8 + d] = AW.P[lay * BW + d];
    }
    if (tid < 4) {
      const int a = tid < NB ? tid : 0;      // (slot 3 at p = 2 is padding: a copy of slot 0, never written from)
      const int ixg = offx + a, rhox = AX.rowmap[ixg], iyg = offy + a, rhoy = AY.rowmap[iyg];
      pen->ps1[tid] = AX.prefix[rhox]; pen->c1[tid] = AX.rcnt[rhox]; pen->P1_0[tid] = AX.P[ixg * BW + (0 - a + P)]; pen->rmx[tid] = rhox;
      pen->ps2[tid] = AY.prefix[rhoy]; pen->c2[tid] = AY.rcnt[rhoy]; pen->rmy[tid] = rhoy;
      for (int b = 0; b < 4; ++b) pen->P2[tid * 4 + b] = AY.P[iyg * BW + ((b < NB ? b : 0) - a + P)];
    }
    if (tid == 64) {
      unsigned fx = 0, fy = 0;
      if (pa.first_touch)
        for (int a = 0; a < NB; ++a) for (int b = 0; b < NB; ++b) {
          if (first_touch_axis<P>(elx, a, b, pa.nelx)) fx |= 1u << (a * 4 + b);
          if (pa.wrap2 ? first_touch_axis_wrapped<P>(ely, a, b) : first_touch_axis<P>(ely, a, b, pa.nely, pa.fty_lo, pa.fty_hi, pa.fty_blocked)) fy |= 1u << (a * 4 + b);
        }
      pen->ftx = fx; pen->fty = fy;
    }
    if (tid >= 128 && tid < 192) {   // per (qx, ix): u0, u1 of axis 1; per (iy, qy): v0, v1 of axis 2 (raw rows)
      const int l2 = tid - 128;
      if (l2 < 32) { const int q = l2 >> 3, a = (l2 >> 1) & 3, k = l2 & 1; uvs[l2] = (q < NB && a < NB) ? AX.tab[((size_t)elx * NB * NB + q * NB + a) * NDER + k] : 0.0; }            // [q][a][2]
      else { const int l3 = l2 - 32, a = l3 >> 3, q = (l3 >> 1) & 3, k = l3 & 1; uvs[l2] = (q < NB && a < NB) ? AY.tab[((size_t)ely * NB * NB + q * NB + a) * NDER + k] : 0.0; }       // [a][q][2]
    }
    if (tid >= 192 && tid < 198) {   // Dirichlet faces this pencil can touch (IGAElementFixJacobian; not the Matrix driver)
      const int k = tid - 192, d = k >> 1, sd = k & 1;
      const AxisDev &A = S.ax[d];
      const int el = d == 0 ? 0 : (d == 1 ? elx : ely);
      bool on = out.op != OP_MATRIX && !A.periodic && S.bcv[d][sd].count > 0;
      if (on) on = (d == 0) ? (sd == 0 ? A.estart == 0 : A.estart + A.nel == A.esizes) : (sd == 0 ? el + A.estart == 0 : el + A.estart == A.esizes - 1);
      unsigned m = 0;
      for (int c = 0; c < 4; ++c) bcv[k * 4 + c] = 0.0;
      if (on) for (int c = 0; c < S.bcv[d][sd].count; ++c) { const int fld = S.bcv[d][sd].field[c]; if (fld < DOF) m |= 1u << fld; }
      bcm[k] = m;
    }
    // elements li_lo - P .. li_lo of the first layer's window (periodic: modulo nel)
    for (int t = 0; t <= P; ++t) load_element(li_lo - t);
  }
  __syncthreads();

  BpBC bc; bc.any = false; bc.v = bcv;
#pragma unroll
  for (int k = 0; k < 6; ++k) { bc.m[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)bcm[k]); bc.on[k] = bc.m[k] != 0u; bc.any = bc.any || bc.on[k]; }
  bc.wlo = bc.on[0] ? lay_first : -1000;
  bc.whi = bc.on[1] ? lay_first + NL - 1 : -1000;

  double uxy[4][3];     // per qy: u0 v0, u1 v0, u0 v1 of this lane's (qx, ix, iy)
  {
    const int qx = lane >> 4, ix = lane & 3, iy = (lane >> 2) & 3;
    const double u0 = uvs[(qx * 4 + ix) * 2 + 0], u1 = uvs[(qx * 4 + ix) * 2 + 1];
#pragma unroll
    for (int qy = 0; qy < 4; ++qy) {
      const double v0 = uvs[32 + (iy * 4 + qy) * 2 + 0], v1 = uvs[32 + (iy * 4 + qy) * 2 + 1];
      uxy[qy][0] = u0 * v0; uxy[qy][1] = u1 * v0; uxy[qy][2] = u0 * v1;
    }
  }
  const long long T0 = S.ax[0].tot, T10 = S.ax[1].tot * S.ax[0].tot;
  // block position of this lane's entry (row slots a1 = lane >> 4 on axis 1, a2 = r on axis 2; column slots b1, b2; column layer
  // lay + d): pos = base[r] + cc[r] * prefix0(lay) + pp[r] * count0(lay) + P0(lay, d)   (axis 1 has consecutive positions)
  const int a1 = lane >> 4, b1 = lane & 3, b2 = (lane >> 2) & 3;
  const bool lane_ok = a1 < NB && b1 < NB && b2 < NB;      // lanes of the zero padding (p = 2) hold no block
  long long pbase[4]; int pcc[4], ppp[4]; unsigned ftm = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long c1 = pen->c1[a1], c2 = pen->c2[r];
    pbase[r] = pen->ps2[r] * T10 + c2 * (pen->ps1[a1] * T0);
    pcc[r] = (int)(c2 * c1);
    ppp[r] = (int)(pen->P2[r * 4 + b2] * c1 + pen->P1_0[a1] + b1);
    if (((pen->ftx >> (a1 * 4 + b1)) & 1u) && ((pen->fty >> (r * 4 + b2)) & 1u)) ftm |= 1u << r;      // first touch: nothing to read
  }

  long long tk0 = 0, tw0 = 0;
  if (out.clk) { tk0 = __builtin_readcyclecounter(); tw0 = wall_clock64(); }
  const long long tw_loop = (kDebug && pa.dbg_buf) ? wall_clock64() : 0;
  for (int it = 0; it < nlay; ++it) {
    const int li = li_lo + it, lay = lay_first + li;
    // the element that enters the window with the next layer: its slot was last read in layer li - 1, which every wave has left
    if (it + 1 < nlay) load_element(li + 1);
    if (pa.prio_layers > 0) { if (it < pa.prio_layers) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    const int c0 = __builtin_amdgcn_readfirstlane(Lcnt[it]);
    const int held = alias0 ? P + 1 : (min(li, pa.nel0 - 1) - max(li - P, 0) + 1);
    const bool bcrow = bc.any && (bc.on[2] || bc.on[3] || bc.on[4] || bc.on[5] || (lay >= bc.wlo - P && lay <= bc.wlo + P) || (lay >= bc.whi - P && lay <= bc.whi + P));
    const long long ps0 = ((long long)__builtin_amdgcn_readfirstlane((int)(Lpre[it] >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)(Lpre[it] & 0xffffffffll));
    long long stamp[6] = {0, 0, 0, 0, 0, 0};
    if (kDebug && pa.dbg_buf) stamp[0] = __builtin_readcyclecounter();
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      // band tiles of this wave: d = {0}, {+1, -P}, {-1, +P} and, at p = 3, {+2, -2}: (P + 1) tile products per wave and layer
      constexpr int NONE = 99;
      const int d = half == 0 ? ((role == 0) ? 0 : (role == 1 ? 1 : (role == 2 ? -1 : (P == 3 ? 2 : NONE))))
                              : ((role == 1) ? -P : (role == 2 ? P : ((P == 3 && role == 3) ? -2 : NONE)));
      if (d == NONE) continue;
      const int p0d = __builtin_amdgcn_readfirstlane(LP[it * 8 + d + P]);
      if (p0d < 0) continue;          // the column layer does not exist (ends of a non-periodic axis)
      // The Gram pairs of a constant-coefficient form are 9 MFMAs per k-step, not 34: a layer's products (37k cycles of issue per
      // wave) are no longer than its two read-add-writes and its barrier, and the old values' round trip to memory would be a third of
      // the layer.  So the loads of a band tile's old blocks go out BEFORE its products (36 doubles per lane at dof 3: the accumulators
      // are 72 registers, not 136) and are consumed behind them.
      constexpr bool PREF = PAIRF && BPT_PREFETCH;
      double pre[PREF ? NB : 1][PREF ? BS : 1];
      if constexpr (PREF) {
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          const long long pos = pbase[r] + (long long)pcc[r] * ps0 + (long long)ppp[r] * c0 + p0d;
          const double *gp = out.val + pos * BS;
          const bool ld = lane_ok && !((ftm >> r) & 1u) && !(kDebug && (pa.debug & 1));
#pragma unroll
          for (int k = 0; k < BS / 2; ++k) { const d2u_t x = ld ? *reinterpret_cast<const d2u_t *>(gp + 2 * k) : (d2u_t){0.0, 0.0}; pre[r][2 * k] = x[0]; pre[r][2 * k + 1] = x[1]; }
          if constexpr (BS & 1) pre[r][BS - 1] = ld ? gp[BS - 1] : 0.0;
        }
      }
      d4_t acc[NACC];
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = (d4_t){0, 0, 0, 0};
      if (!(kDebug && (pa.debug & 2))) {
#pragma unroll 1
        for (int ta = 0; ta <= P; ++ta) {
          const int tb = ta + d;
          if (tb < 0 || tb > P) continue;
          const int eu = li - ta;        // (unwrapped: the ring slot follows it, the element itself is eu modulo nel on a wrapped axis)
          if (!alias0 && (eu < 0 || eu >= pa.nel0)) continue;
          bpt_product<Form, GEO, RAT, NB>(acc, ring + bpt_slot(eu) * REC, ta, tb, uxy, lane, prm.v, out.shift);
        }
      }
      if constexpr (PAIRF) {      // Gram sums -> blocks, in place (block_pencil.hpp: the form's constants C = mat(e_f, e_g))
        PtView p0; p0.x = nullptr; p0.u = nullptr; p0.ut = nullptr; p0.gu = nullptr; p0.hu = nullptr; p0.G = nullptr; p0.prm = prm.v; p0.shift = out.shift; p0.t = out.t;
        p0.normal = nullptr; p0.atboundary = 0; p0.boundary_id = -1;
        bp_transform<Form>(acc, p0);
      }
      else if constexpr (band_nacc_of<Form>::own) Form::band_finish(acc, prm.v);
      else if constexpr (band_nfeat_of<Form>::v == 5) Form::template band_combine<0>(acc);
      if (kDebug && pa.dbg_buf) stamp[1 + 2 * half] = __builtin_readcyclecounter();
      if (kDebug && (pa.debug & 1)) continue;
      // ---- read-add-write of the lane's four blocks (r = row slot on axis 2): a block is one 128-byte line
      if (pa.rmw_prio) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (!lane_ok) continue;
        const long long pos = pbase[r] + (long long)pcc[r] * ps0 + (long long)ppp[r] * c0 + p0d;
        double *gp = out.val + pos * BS;
        const bool first = (ftm >> r) & 1u;
        // (a block of 9 values starts on an 8-byte boundary: d2u_t is the pair type without the 16-byte promise)
        d2u_t oldv[BS / 2]; double oldl = 0.0;
        if constexpr (PREF) {      // (read ahead of the products, above)
#pragma unroll
          for (int k = 0; k < BS / 2; ++k) { oldv[k][0] = pre[r][2 * k]; oldv[k][1] = pre[r][2 * k + 1]; }
          if constexpr (BS & 1) oldl = pre[r][BS - 1];
        } else {
#pragma unroll
        for (int k = 0; k < BS / 2; ++k) oldv[k] = first ? (d2u_t){0.0, 0.0} : *reinterpret_cast<const d2u_t *>(gp + 2 * k);
        if constexpr (BS & 1) oldl = first ? 0.0 : gp[BS - 1];
        }
        double K[BS];
#pragma unroll
        for (int n = 0; n < BS; ++n) K[n] = acc[n][r];
        if (bcrow) {      // IGAElementFixJacobian on the combined blocks (src/petigaelem.c:1463-1490)
#pragma unroll
          for (int i = 0; i < DOF; ++i)
#pragma unroll
            for (int j = 0; j < DOF; ++j) {
              double va = 0, vb = 0;
              const bool fa = bp_fixed<P>(bc, a1, r, lay, i, va), fb = bp_fixed<P>(bc, b1, b2, lay + d, j, vb);
              if (fa || fb) K[i * DOF + j] = (d == 0 && a1 == b1 && r == b2 && i == j) ? (double)held : 0.0;
            }
        }
#pragma unroll
        for (int k = 0; k < BS / 2; ++k) { d2u_t w; w[0] = oldv[k][0] + K[2 * k]; w[1] = oldv[k][1] + K[2 * k + 1]; *reinterpret_cast<d2u_t *>(gp + 2 * k) = w; }
        if constexpr (BS & 1) gp[BS - 1] = oldl + K[BS - 1];
      }
      if (pa.rmw_prio) { if (it < pa.prio_layers) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
      if (kDebug && pa.dbg_buf) stamp[2 + 2 * half] = __builtin_readcyclecounter();
    }
    if (kDebug && pa.dbg_buf) stamp[5] = __builtin_readcyclecounter();
    __syncthreads();      // the next element's record is in the ring (vmcnt(0) before the barrier), nobody reads layer li's window any more
    if (kDebug && pa.dbg_buf && lane == 0 && it < 64) {
      long long *d = pa.dbg_buf + (((size_t)blockIdx.x * 4 + role) * 64 + it) * 8;
      for (int k = 0; k < 6; ++k) d[k] = stamp[k];
      d[6] = __builtin_readcyclecounter(); d[7] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4);      // after the barrier; HW_ID
    }
  }
  if (kDebug && pa.dbg_buf && tid == 0) {      // workgroup record behind the layer stamps: start / end on the 100 MHz clock, HW_ID, XCC_ID
    long long *w = pa.dbg_buf + (size_t)gridDim.x * 4 * 64 * 8 + (size_t)blockIdx.x * 4;
    w[0] = tw_wg; w[1] = wall_clock64(); pa.dbg_buf[(((size_t)blockIdx.x * 4 + 1) * 64 + 63) * 8] = tw_loop - tw_wg; w[2] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4); w[3] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
  }
  if (out.clk && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && tid == 0) {
    // IGX_CLOCK_PROBE: s_memtime against the 100 MHz s_memrealtime over the segment, first and last workgroup of every launch
    atomicAdd(reinterpret_cast<unsigned long long *>(out.clk), (unsigned long long)(__builtin_readcyclecounter() - tk0));
    atomicAdd(reinterpret_cast<unsigned long long *>(out.clk) + 1, (unsigned long long)(wall_clock64() - tw0));
    atomicAdd(reinterpret_cast<unsigned long long *>(out.clk) + 2, (unsigned long long)nlay);
  }
}

// ---- host side
// the form qualifies (also evaluated inside a run-time module: rtc.hpp reads it back)
template <class Form> constexpr bool bpt_form_ok() {
  if constexpr (bpt_pairs<Form>())      // (round 6) the Gram pairs of a constant-coefficient form with 2 or 3 fields, first order
    return (Form::DOF == 2 || Form::DOF == 3) && shape_order_of<Form>::v < 2 && Form::ORDER < 2 && !has_boundary_of<Form>::v && nscalar_of<Form>::v == 0 &&
           (mat_pair_mask_of<Form>::v >> 32) == 0ull && ((mat_pair_mask_of<Form>::v >> 4) & 0x0f0f0f0full) == 0ull && (mat_need_of<Form>::v & ~(NEED_U | NEED_G)) == 0u;
  else if constexpr (!has_point_coef<Form>::v) return false;
  else return Form::DOF == 4 && shape_order_of<Form>::v < 2 && !has_boundary_of<Form>::v && nscalar_of<Form>::v == 0 && mat_pair_mask_of<Form>::v == 0ull &&
              (mat_need_of<Form>::v & ~(NEED_U | NEED_G)) == 0u;
}
// MFMAs per k-step: one per (accumulator, test feature) product
template <class Form> constexpr int bpt_products() {
  if constexpr (bpt_pairs<Form>()) return fm_popcount(mat_pair_mask_of<Form>::v);
  int nm = 0;
  for (int n = 0; n < band_nacc_of<Form>::v; ++n) for (int f = 0; f < band_nfeat_of<Form>::v; ++f) if ((bpt_acc_mask<Form>(n) >> f) & 1u) nm++;
  return nm;
}

#ifndef IGX_RTC
// 3-D, p = 3 with 4 Gauss points per axis, 4 fields, matrix-only drivers (Matrix / Jacobian / IJacobian); axis 0: one new node layer
// per element (a periodic axis wrapped inside the rank is taken: layers and elements modulo nel, at least 2p+1 of them); axis 1 not
// wrapped inside the rank; axis 2 either way; any geometry (none / polynomial / NURBS) of dimension 3
// (dof 4: the matrix-only drivers; dof 2, 3 -- the Gram pairs of a constant-coefficient form -- also the System driver, whose vector comes from a
//  pass of its own: try_band_pt)
static bool band_pt_covers_space(const Space &s, const SpaceDev &S, const OutDev &out, int dof = 4, bool system_too = false) {
  if (s.env.block_pencil == 0) return false;
  if (out.op != OP_MATRIX && out.op != OP_JACOBIAN && out.op != OP_IJACOBIAN && !(system_too && out.op == OP_SYSTEM)) return false;
  if (s.dim != 3 || s.dof != dof || (s.nsd != 0 && s.nsd != 3) || S.fixtable) return false;
  const int p = s.axis[0].p;
  if (p != 2 && p != 3) return false;
  for (int d = 0; d < 3; ++d) {
    if (s.axis[d].p != p || s.basis[d].nqp != p + 1 || s.basis[d].nen != p + 1) return false;
    for (int sd = 0; sd < 2; ++sd) if (s.visit[d][sd]) return false;
    if (s.lay[d].alias && s.axis[d].nnp < 2 * p + 1) return false;
  }
  if (s.lay[1].alias) return false;
  if (s.elem_width[0] < 8) return false;
  for (int e = 0; e + 1 < s.elem_width[0]; ++e) if (s.basis[0].offset[s.elem_start[0] + e + 1] != s.basis[0].offset[s.elem_start[0] + e] + 1) return false;
  return true;
}
template <class Form>
static bool band_pt_covers(const Space &s, const SpaceDev &S, const OutDev &out) {
  if constexpr (!bpt_form_ok<Form>()) return false;
  else return band_pt_covers_space(s, S, out, Form::DOF, bpt_pairs<Form>());
}

// the launches of an assembly; `launch(points, grid, lds_bytes, geo, rat, degree, args)` starts band_points (points = true: 256 threads, four
// elements per workgroup) or band_pt (256 threads, lds_bytes of dynamic LDS) -- the compiled-in instantiations of a built-in form,
// or the module functions of a run-time struct (rtc.hpp); rec = bpt_rec<Form>(), products = bpt_products<Form>()
typedef std::function<void(bool, unsigned, size_t, bool, bool, int, const BandArgs &)> BandPtLaunch;
static int band_pt_run(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, std::string &kname, int &launches, std::string &err, bool &done, DomInfo &dom,
                       const std::function<void()> &zero_matrix, const std::function<void()> &slab_done, int rec, int products, const BandPtLaunch &launch) {
  const int P = s.axis[0].p;
  const bool alias0 = s.lay[0].alias != 0;
  // (the walk axis needs no rule: a pencil writes every block of its band rows exactly once, wrapped or not)
  const bool wrap2 = axis_first_touch_wrapped_ok(s, 2);
  const bool first_touch = !s.env.no_first_touch && out.val && axis_first_touch_ok(s, 1) && (axis_first_touch_ok(s, 2) || wrap2);
  if (!first_touch) { if (zero_matrix) zero_matrix(); }
  else if (s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] > 1) zero_neighbour_rows(s, out, stream);
  launches = 0;
  const int NL = alias0 ? s.elem_width[0] : s.elem_width[0] + P;
  const bool geo = s.nsd != 0, rat = s.rational != 0;
  int rc = 0;
  auto run = [&](const Box &bx, const int *fty) {
    for (int d = 1; d < 3; ++d) if (bx.hi[d] <= bx.lo[d]) return;
    for (int cy = 0; cy < s.lay[2].ncolors && !rc; ++cy) for (int cx = 0; cx < s.lay[1].ncolors && !rc; ++cx) {
      BandArgs pa; memset(&pa, 0, sizeof(pa));
      pa.first_touch = first_touch ? 1 : 0; pa.nelx = s.elem_width[1]; pa.nely = s.elem_width[2];
      pa.fty_lo = fty ? fty[0] : 0; pa.fty_hi = fty ? fty[1] : 0x7fffffff; pa.fty_blocked = fty ? fty[2] : 0x7fffffff;
      if (!color_range(s.lay[1], cx, bx.lo[1], bx.hi[1], pa.ex_start, pa.ex_step, pa.ex_count)) continue;
      if (!color_range(s.lay[2], cy, bx.lo[2], bx.hi[2], pa.ey_start, pa.ey_step, pa.ey_count)) continue;
      // (irregular colours of a wrapped axis hold single elements: color_range gives start / count with the regular step)
      pa.nel0 = s.elem_width[0]; pa.alias0 = alias0 ? 1 : 0; pa.wrap2 = wrap2 ? 1 : 0;
      const long long pencils = (long long)pa.ex_count * pa.ey_count;
      // Two four-wave workgroups per CU (registers, LDS).  A segment costs nothing but the window of element records at its start
      // (no halo is recomputed, and the SIMD gives its older wavefront priority: the partner workgroup keeps the pipe busy through a
      // newcomer's prologue), so short segments win: they even out the end of a launch.  Measured at 64^3 / 96^3 / 128^3 on a NURBS
      // map (IGX_NSEG sweeps, profiles/r04_nsvms_segments.txt): four layers per segment were the best length at every size.
      const int max_len = 64;
      int nseg = (NL + 3) / 4;
      if (s.env.nseg > 0) nseg = std::max((NL + max_len - 1) / max_len, std::min(s.env.nseg, std::max(1, NL / 2)));
      pa.seg_len = (NL + nseg - 1) / nseg; pa.nseg = (NL + pa.seg_len - 1) / pa.seg_len;
      pa.debug = s.env.debug_feature; pa.dbg_block = 7 + s.env.debug_noflush;
      pa.prio_layers = s.env.band_prio; pa.rmw_prio = s.env.band_rmw_prio;
      const size_t need = (size_t)pencils * pa.nel0 * rec * sizeof(double);
      if (pool_alloc(reinterpret_cast<void **>(&pa.pts), need, stream) != hipSuccess) { err = "device allocation of the point records failed"; rc = IGX_ERR_MEM; return; }
      const long long nelem = pencils * pa.nel0;
      launch(true, (unsigned)((nelem + 3) / 4), 0, geo, rat, P, pa);
      const size_t lds = (size_t)bpt_carve(rec, pa.seg_len).total * sizeof(double);
      static int dbg_done = 0;
      const bool dbg_t = kDebug && s.env.debug_timing && !dbg_done;
      const size_t dbg_n = (size_t)pencils * pa.nseg * 4 * 64 * 8 + (size_t)pencils * pa.nseg * 4;
      if (dbg_t) { (void)hipMalloc((void **)&pa.dbg_buf, dbg_n * 8); (void)hipMemset(pa.dbg_buf, 0, dbg_n * 8); }
      launch(false, (unsigned)(pencils * pa.nseg), lds, geo, rat, P, pa);
      if (dbg_t) {   // IGX_DEBUG_TIMING=1: where a layer's cycles go, per role, over the workgroups of the first launch (diagnostic only)
        dbg_done = 1;
        (void)hipStreamSynchronize(stream);
        std::vector<long long> h(dbg_n);
        (void)hipMemcpy(h.data(), pa.dbg_buf, dbg_n * 8, hipMemcpyDeviceToHost);
        double sum[4][6] = {{0}}; long long cnt[4] = {0}; long long odd = 0, tot = 0;
        for (size_t b = 0; b < (size_t)pencils * pa.nseg; ++b) for (int w = 0; w < 4; ++w) for (int l = 1; l + 1 < pa.seg_len && l < 63; ++l) {
          const long long *d = &h[((b * 4 + w) * 64 + l) * 8], *dn = d + 8;
          if (!d[0] || !dn[0]) continue;
          const long long d3 = d[3] ? d[3] : d[2], d4 = d[4] ? d[4] : d3;      // (role 0 has one band tile)
          sum[w][0] += (double)(d[1] - d[0]); sum[w][1] += (double)(d[2] - d[1]); sum[w][2] += (double)(d3 - d[2]); sum[w][3] += (double)(d4 - d3);
          sum[w][4] += (double)(d[6] - d[5]); sum[w][5] += (double)(dn[0] - d[0]); cnt[w]++;
          if (w == 0 && l == 1) { tot++; odd += d[7] & 1; }
        }
        for (int w = 0; w < 4; ++w) if (cnt[w]) fprintf(stderr, "[igx band_pt timing] role %d n=%lld cycles: tile A products %.0f | rmw %.0f | tile B products %.0f | rmw %.0f | barrier %.0f | layer %.0f\n",
                                                         w, cnt[w], sum[w][0] / cnt[w], sum[w][1] / cnt[w], sum[w][2] / cnt[w], sum[w][3] / cnt[w], sum[w][4] / cnt[w], sum[w][5] / cnt[w]);
        {   // the schedule: per XCC the span of its workgroups, the sum of their durations, the latest start
          const long long *wg = &h[(size_t)pencils * pa.nseg * 4 * 64 * 8]; const size_t nwg = (size_t)pencils * pa.nseg;
          long long t0 = LLONG_MAX, t1 = 0; for (size_t b = 0; b < nwg; ++b) { t0 = std::min(t0, wg[b * 4]); t1 = std::max(t1, wg[b * 4 + 1]); }
          double pro = 0; for (size_t b = 0; b < nwg; ++b) pro += (double)h[((b * 4 + 1) * 64 + 63) * 8];
          fprintf(stderr, "[igx band_pt schedule] %zu workgroups, launch span %.1f us, mean prologue (tables, window) %.1f us\n", nwg, (t1 - t0) / 100.0, pro / nwg / 100.0);
          for (int x = 0; x < 8; ++x) {
            long long n = 0, first_end = LLONG_MAX, last_end = 0, last_start = 0; double dur = 0; long long cus[64] = {0};
            for (size_t b = 0; b < nwg; ++b) if ((wg[b * 4 + 3] & 15) == x) {
              n++; dur += (double)(wg[b * 4 + 1] - wg[b * 4]); last_end = std::max(last_end, wg[b * 4 + 1]); last_start = std::max(last_start, wg[b * 4]);
              const long long hw = wg[b * 4 + 2]; cus[((hw >> 8) & 15) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 1)]++;      // CU_ID, SH_ID, SE_ID (low bit)
            }
            long long cmin = LLONG_MAX, cmax = 0; int ncus = 0; for (int c = 0; c < 64; ++c) if (cus[c]) { ncus++; cmin = std::min(cmin, cus[c]); cmax = std::max(cmax, cus[c]); }
            if (n) fprintf(stderr, "[igx band_pt schedule] xcc %d: %lld workgroups, mean duration %.1f us, last start %.1f us, last end %.1f us; %d CU ids, workgroups per id %lld..%lld\n",
                           x, n, dur / n / 100.0, (last_start - t0) / 100.0, (last_end - t0) / 100.0, ncus, cmin, cmax);
          }
        }
        {   // per layer index of a segment: cycles from the start of the layer to the start of the next one (role 0)
          fprintf(stderr, "[igx band_pt timing] layer periods by index:");
          for (int l = 0; l < pa.seg_len && l < 63; ++l) {
            double sm = 0; long long n = 0;
            for (size_t b = 0; b < (size_t)pencils * pa.nseg; ++b) { const long long *d = &h[((b * 4 + 0) * 64 + l) * 8]; if (d[0] && d[6]) { sm += (double)((l + 1 < pa.seg_len && d[8]) ? d[8] - d[0] : d[6] - d[0]); n++; } }
            if (n) fprintf(stderr, " %.0f", sm / n);
          }
          fprintf(stderr, "\n");
        }
        fprintf(stderr, "[igx band_pt timing] workgroups in an odd wave slot: %lld of %lld; seg_len %d nseg %d\n", odd, tot, pa.seg_len, pa.nseg);
        (void)hipFree(pa.dbg_buf);
      }
      (void)hipFreeAsync(pa.pts, stream);
      launches++;
    }
  };
  Box all; for (int d = 0; d < 3; ++d) { all.lo[d] = 0; all.hi[d] = s.elem_width[d]; }
  if (dom.ev0) (void)hipEventRecord(dom.ev0, stream);
  const int n2 = s.elem_width[2];
  const bool upper2 = s.proc_sizes[2] > 1 && (s.proc_ranks[2] < s.proc_sizes[2] - 1 || s.axis[2].periodic);
  if (slab_done && upper2 && n2 >= 2 * (P + 1)) {
    const int c2 = face_cut(n2, P);      // (a thick pass: pencil_common.hpp)
    Box top = all, rest = all; top.lo[2] = c2; rest.hi[2] = c2;
    const int ft_top[3] = {c2, n2, 0x7fffffff}, ft_rest[3] = {0, c2, c2};
    run(top, ft_top);
    if (!rc) slab_done();
    if (!rc) run(rest, ft_rest);
  } else run(all, nullptr);
  if (rc) return rc;
  if (dom.ev1) (void)hipEventRecord(dom.ev1, stream);
  if (hipGetLastError() != hipSuccess) { err = "band_pt kernel launch failed"; return IGX_ERR_LIB; }
  dom.name = std::string("band_pt<p=") + char('0' + P) + ">"; dom.launches = launches;
  dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2];
  dom.flop_per_element = 2048.0 * products * (P + 1) * (P + 1) * (P + 1) * (P + 1);      // (P+1)^2 tile products of (P+1)^2 k-steps per layer
  kname = std::string("band_pt(mfma_f64_16x16x4,p=") + char('0' + P) + ",dof=" + char('0' + s.dof) + std::string(",band rows by node layer,point records,whole blocks per lane") + (geo ? (rat ? ",NURBS geometry)" : ",mapped geometry)") : ")");
  done = true;
  return 0;
}

template <class Form>
static int try_band_pt(const Space &s, const SpaceDev &S, const ParamsDev &prm, const OutDev &out, hipStream_t stream, std::string &kname, int &launches,
                       std::string &err, bool &done, DomInfo &dom, const std::function<void()> &zero_matrix, const std::function<void()> &slab_done) {
  done = false;
  if constexpr (!bpt_form_ok<Form>()) return 0;
  else {
  if (!band_pt_covers<Form>(s, S, out)) return 0;
  if constexpr (band_nacc_of<Form>::own) { if (!Form::band_params_ok(prm.v)) return 0; }
  if constexpr (bpt_pairs<Form>()) {
    // The System driver of a constant-coefficient form: the matrix below (rows and columns of the fixed dofs emptied, the diagonal
    // counting the elements: IGAElementFixSystem's K part is IGAElementFixJacobian's), the VECTOR from a sum-factorised pass of its
    // own that never sees a K_e -- vec() minus the lifting of the Dirichlet values through the form's linearity in N_b, a fixed row
    // taking its value once per element (forms.hpp: SystemVectorOf; vec_sumfact.hpp, OutDev::vec_mode 1).  Boundary loads on a mapped
    // geometry stay with the feature kernel.
    if (out.op == OP_SYSTEM) {
      for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) if (s.load[d][sd].count) return 0;
#ifdef IGX_HAVE_VEC_SUMFACT
      OutDev ov = out; ov.vec_mode = 1; ov.op = OP_FUNCTION; ov.val = nullptr; ov.browptr = nullptr; ov.U = nullptr; ov.V = nullptr;
      bool vdone = false; int vl = 0; std::string vk;
      if (int rc = try_vec_sumfact<SystemVectorOf<Form>>(s, S, prm, ov, stream, vk, vl, vdone)) { err = "vec_sumfact kernel launch failed"; return rc; }
      if (!vdone) return 0;
#else
      return 0;
#endif
    }
  }
  return band_pt_run(s, S, out, stream, kname, launches, err, done, dom, zero_matrix, slab_done, bpt_rec<Form>(), bpt_products<Form>(),
                     [&](bool points, unsigned grid, size_t lds, bool geo, bool rat, int deg, const BandArgs &pa) {
                       if (points) { hipLaunchKernelGGL((deg == 2 ? band_points<Form, 2> : band_points<Form, 3>), dim3(grid), dim3(256), 0, stream, S, prm, out, pa); return; }
                       auto kern = deg == 2 ? (geo ? (rat ? band_pt<Form, true, true, 2> : band_pt<Form, true, false, 2>) : band_pt<Form, false, false, 2>)
                                            : (geo ? (rat ? band_pt<Form, true, true, 3> : band_pt<Form, true, false, 3>) : band_pt<Form, false, false, 3>);
                       (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                       hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, S, prm, out, pa);
                     });
  }
}
#endif   // !IGX_RTC

}  // namespace igx
