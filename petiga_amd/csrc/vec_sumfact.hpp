// vec_sumfact.hpp -- the vector-only drivers (IGAComputeVector / Function / IFunction, src/petigaksp.c:33, src/petigasnes.c:23,
// src/petigats.c:23) by sum factorisation in both directions.
//
// The reference's point callback fills F[a] for every basis function a of the element at every quadrature point
// (IGAPointAddVec, src/petigapoint.c:427-450): nen * nqp evaluations per element, and the feature kernel's vector-only mode keeps
// that shape (Phi for all (a,q) pairs, vec() per pair).  But every IGAFormVector / Function / IFunction is LINEAR in the test
// function, F_a^i = sum_q JW sum_f Phi_f(a,q) r^i_f(q) with r_f = vec(p, e_f), and on a tensor-product basis both halves factor over
// the axes:
//   forward   u, grad u, hess u (and x, dx/du, d2x/du2 of a mapped geometry) at the points from the nodal values: three short
//             contractions, one axis at a time (src/petigaval.F90:182-232 summed the long way);
//   point     the geometry chain K3-K6 (Rationalize, GeometryMap, InverseMap, src/petigarat/mapgeo/mapinv.f90.in), the form's
//             vec() on the unit test features -- nqp evaluations instead of nen * nqp;
//   backward  F_a = w_a sum_q sum_k C_k(q) D_k N_a(q) over the 10 parametric derivatives D_k up to order 2: the transpose of the
//             forward contractions.
// One wavefront per element (4 x 4 x 4 lanes = nodes = points), two elements per wavefront when no axis has more than three basis
// functions or points (p <= 2: 2 x 27 of the 64 lanes), no matrix cores (there is no
// dense contraction left: the work per element drops from ~nen * nqp * features to ~(nen + nqp) * (p+1) * components), coloured
// scatter (conflict-free, fixed order: bitwise repeatable).  dim 3, nen <= 4 and nqp <= 4 per axis; first-order test features on
// any geometry, second-order test features (Cahn-Hilliard's Laplacian) likewise (round 4); no boundary loads, no boundary
// passes -- everything else stays on the feature kernel.
#pragma once
#include "feature_mfma.hpp"

namespace igx {

// derivative k of the 10 (value, d0, d1, d2, d00, d01, d02, d11, d12, d22): orders on the axes, and the two-axis stage m it comes from
__device__ __forceinline__ constexpr int vs_v0(int k) { return (k == 1 || k == 5 || k == 6) ? 1 : (k == 4 ? 2 : 0); }
__device__ __forceinline__ constexpr int vs_v1(int k) { return (k == 2 || k == 5 || k == 8) ? 1 : (k == 7 ? 2 : 0); }
__device__ __forceinline__ constexpr int vs_v2(int k) { return (k == 3 || k == 6 || k == 8) ? 1 : (k == 9 ? 2 : 0); }
// m -> (v0, v1): 0 (0,0) 1 (1,0) 2 (0,1) 3 (2,0) 4 (1,1) 5 (0,2)
__device__ __forceinline__ constexpr int vs_m(int v0, int v1) { return v0 == 0 ? (v1 == 0 ? 0 : (v1 == 1 ? 2 : 5)) : (v0 == 1 ? (v1 == 0 ? 1 : 4) : 3); }
__device__ __forceinline__ constexpr int vs_mv0(int m) { return (m == 1 || m == 4) ? 1 : (m == 3 ? 2 : 0); }
__device__ __forceinline__ constexpr int vs_mv1(int m) { return (m == 2 || m == 4) ? 1 : (m == 5 ? 2 : 0); }

#define VS_SYNC() do { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); } while (0)

// forward: D[k] = sum_a coef_a D_k N_a(q) at this lane's point; ORD = highest derivative order wanted.  buf: 10 x 64 doubles of this
// wave; tab[d]: [q][a][3] (value, first, second derivative), zero padded to NS x NS.
// NS: lanes per axis -- 4: one element per wavefront; 3 (nen, nqp <= 3 on every axis): two elements per wavefront, lanes 0..26 and
// 27..53.  L: this lane's (i0, i1, i2) and the first lane of its element.
struct VsLane { int i0, i1, i2, base; };
template <int ORD, int NS>
__device__ __forceinline__ void vs_forward(double coef, double *buf, const double *tab0, const double *tab1, const double *tab2, int lane, const VsLane &L, double (&D)[10]) {
  const int i0 = L.i0, i1 = L.i1, i2 = L.i2;
  constexpr int S1 = NS, S2 = NS * NS;
  double *in = buf + L.base, *T1 = buf + 64 + L.base, *T2 = buf + 4 * 64 + L.base;
  lane -= L.base;
  VS_SYNC();
  in[lane] = coef;
  VS_SYNC();
  {   // axis 0: lane (q0, a1, a2)
    double t[3] = {0, 0, 0};
#pragma unroll
    for (int a0 = 0; a0 < NS; ++a0) {
      const double c = in[a0 + S1 * i1 + S2 * i2];
#pragma unroll
      for (int v = 0; v <= ORD; ++v) t[v] += c * tab0[(i0 * NS + a0) * 3 + v];
    }
#pragma unroll
    for (int v = 0; v <= ORD; ++v) T1[v * 64 + lane] = t[v];
  }
  VS_SYNC();
  {   // axis 1: lane (q0, q1, a2)
    double t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a1 = 0; a1 < NS; ++a1) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        if (vs_mv0(m) + vs_mv1(m) > ORD) continue;
        t[m] += T1[vs_mv0(m) * 64 + i0 + S1 * a1 + S2 * i2] * tab1[(i1 * NS + a1) * 3 + vs_mv1(m)];
      }
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) if (vs_mv0(m) + vs_mv1(m) <= ORD) T2[m * 64 + lane] = t[m];
  }
  VS_SYNC();
#pragma unroll
  for (int k = 0; k < 10; ++k) D[k] = 0.0;
#pragma unroll
  for (int a2 = 0; a2 < NS; ++a2) {
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      if (vs_v0(k) + vs_v1(k) + vs_v2(k) > ORD) continue;
      D[k] += T2[vs_m(vs_v0(k), vs_v1(k)) * 64 + i0 + S1 * i1 + S2 * a2] * tab2[(i2 * NS + a2) * 3 + vs_v2(k)];
    }
  }
}

// backward: F_a = sum_q sum_k C_k(q) D_k N_a(q) at this lane's node; buf: 16 x 64 doubles of this wave
// KMASK: the derivatives k whose coefficient can be non-zero (the others are neither stored nor summed); Cb holds them back to back
__device__ __forceinline__ constexpr int vs_popc(unsigned m) { int n = 0; for (; m; m &= m - 1) ++n; return n; }
__device__ __forceinline__ constexpr int vs_kc(unsigned kmask, int k) { return vs_popc(kmask & ((1u << k) - 1u)); }
template <int ORD, unsigned KMASK, int NS>
__device__ __forceinline__ double vs_backward(const double (&C)[10], double *buf, const double *tab0, const double *tab1, const double *tab2, int lane, const VsLane &L) {
  const int i0 = L.i0, i1 = L.i1, i2 = L.i2;
  constexpr int T1s = NS, T2s = NS * NS;
  // (round 6: S2 takes the place of Cb -- every lane holds its sums in registers until all have read Cb -- and S1 sits behind S2:
  //  max(popc, 9) buffers instead of popc + 6; with the forward stage's 10 that is 30 KB per workgroup, five workgroups per CU instead of four)
  constexpr int NS2 = ORD >= 2 ? 6 : 3;
  double *Cb = buf + L.base, *S2 = buf + L.base, *S1 = buf + NS2 * 64 + L.base;
  lane -= L.base;
  VS_SYNC();
#pragma unroll
  for (int k = 0; k < 10; ++k) if (((KMASK >> k) & 1u) && vs_v0(k) + vs_v1(k) + vs_v2(k) <= ORD) Cb[vs_kc(KMASK, k) * 64 + lane] = C[k];
  VS_SYNC();
  {   // axis 2: lane (q0, q1, a2)
    double t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q2 = 0; q2 < NS; ++q2) {
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        if (!((KMASK >> k) & 1u) || vs_v0(k) + vs_v1(k) + vs_v2(k) > ORD) continue;
        t[vs_m(vs_v0(k), vs_v1(k))] += Cb[vs_kc(KMASK, k) * 64 + i0 + T1s * i1 + T2s * q2] * tab2[(q2 * NS + i2) * 3 + vs_v2(k)];
      }
    }
    VS_SYNC();      // (every lane has read Cb)
#pragma unroll
    for (int m = 0; m < 6; ++m) if (vs_mv0(m) + vs_mv1(m) <= ORD) S2[m * 64 + lane] = t[m];
  }
  VS_SYNC();
  {   // axis 1: lane (q0, a1, a2)
    double t[3] = {0, 0, 0};
#pragma unroll
    for (int q1 = 0; q1 < NS; ++q1) {
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        if (vs_mv0(m) + vs_mv1(m) > ORD) continue;
        t[vs_mv0(m)] += S2[m * 64 + i0 + T1s * q1 + T2s * i2] * tab1[(q1 * NS + i1) * 3 + vs_mv1(m)];
      }
    }
#pragma unroll
    for (int v = 0; v <= ORD; ++v) S1[v * 64 + lane] = t[v];
  }
  VS_SYNC();
  double f = 0;
#pragma unroll
  for (int q0 = 0; q0 < NS; ++q0) {
#pragma unroll
    for (int v = 0; v <= ORD; ++v) f += S1[v * 64 + q0 + T1s * i1 + T2s * i2] * tab0[(q0 * NS + i0) * 3 + v];
  }
  return f;
}

// identity geometry: test feature tf (0 value, 1..3 gradient, 4 + 3a + b second derivative) -> derivative k of the 10
__device__ __forceinline__ constexpr unsigned vs_kmask_ident(unsigned vmask) {
  unsigned m = 0;
  for (int tf = 0; tf < 13; ++tf) {
    if (!((vmask >> tf) & 1u)) continue;
    if (tf < 4) m |= 1u << tf;
    else { const int a = (tf - 4) / 3, b = (tf - 4) % 3, lo = a < b ? a : b, hi = a < b ? b : a; m |= 1u << (4 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5))); }
  }
  return m | 0x7u;      // (at least three slots: S1 is laid over them)
}

// GEO: a mapped geometry and / or NURBS weights; without them the geometry chain is the identity at compile time (no E1 / E2
// products, no quotient rule) and the kernel needs half the registers
// the pipelined walk of round 6 (see the kernel): scalar forms without a geometry, where its registers fit four wavefronts per SIMD
template <class Form, bool GEO> __host__ __device__ constexpr bool vs_pipe() { return !GEO && Form::DOF == 1; }
template <class Form, bool GEO, int NS = 4>
__global__ void __launch_bounds__(256, (vs_pipe<Form, GEO>() ? 4 : 2))      // (two waves per SIMD: the geometry variants of NS-VMS and Cahn-Hilliard need 290-350 VGPRs uncapped, one wave per SIMD)
vec_sumfact(SpaceDev S, ParamsDev prm, OutDev out, ColorRange cr, long long nelem) {
  constexpr int EPW = NS == 3 ? 2 : 1, NL = NS * NS * NS;               // elements per wavefront, lanes per element
  constexpr int DOF = Form::DOF;
  constexpr unsigned VMASK = vec_test_mask_of<Form>::v;                  // test features vec() reads (bit f)
  constexpr bool SECOND_T = shape_order_of<Form>::v >= 2;                // second-order test features
  constexpr bool NEEDHU = (Form::NEED & NEED_HU) != 0, NEEDGU = (Form::NEED & (NEED_GU | NEED_HU)) != 0;
  constexpr int UORD = NEEDHU ? 2 : (NEEDGU ? 1 : 0);                    // derivative order of the state
  constexpr int NFS = SECOND_T ? 13 : 4;
  // derivatives of the test functions that can carry a coefficient: without a geometry feature tf maps to one of them; with one, the
  // inverse Jacobian mixes the three first derivatives and the rational correction reaches the value
  constexpr unsigned KMASK = GEO ? (SECOND_T ? 0x3FFu : 0xFu) : vs_kmask_ident(VMASK & ((1u << NFS) - 1u));
  constexpr int NBACK = vs_popc(KMASK) > (SECOND_T ? 9 : 5) ? vs_popc(KMASK) : (SECOND_T ? 9 : 5), NFWD = (UORD == 2 || GEO) ? 10 : 7, NBUF = NBACK > NFWD ? NBACK : NFWD;
  constexpr int TB = NS * NS * 3;                                        // doubles of one axis' rows [q][a][3], zero padded to NS x NS
  __shared__ double sm_all[4][NBUF * 64 + EPW * 3 * TB];      // Cahn-Hilliard without a geometry: 25 KB per workgroup, two elements per wavefront (a sixth workgroup per CU measured no gain over five: 12.8 ms either way)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int esub = (EPW == 2 && lane >= NL) ? 1 : 0;                    // which of the wavefront's elements this lane works on
  // Round 6 (PIPE: scalar forms without a geometry): every global load of the unit -- the 1-D rows, the node's state and old F, the
  // point's weights -- leaves at its start, behind closed forms for the element's offset and the row map.  A wavefront's life was a
  // chain of dependent round trips (offset -> row map -> state; the rows; the point's weights; F read-add-write at the end, each
  // behind a wavefront fence the compiler does not move loads across) around 2-3 us of arithmetic: 9.3 us with four wavefronts per
  // SIMD to hide it, 7.7 us now.  (The colour makes the early read of F safe: no other element of the launch touches these rows.)
  // Off for the geometry variants: they are at their 256 registers already.
  constexpr bool PIPE = vs_pipe<Form, GEO>();
  constexpr bool EF = GEO && Form::DOF == 1;      // ... scalar forms on a geometry: old F and the product of the point's weights (three registers) leave early too
  constexpr int TPL = EPW * 3;                                           // table entries per lane: lane j < TB holds entry j of each (element, axis) block
  const long long nunits = (nelem + EPW - 1) / EPW, ustride = (long long)gridDim.x * 4;
  long long unit = (long long)blockIdx.x * 4 + wave;
  if (unit >= nunits) return;
  double *buf = sm_all[wave], *tab0 = buf + NBUF * 64 + esub * 3 * TB, *tab1 = tab0 + TB, *tab2 = tab1 + TB;
  auto element_at = [&](long long ww, long long wfirst, int (&e3)[3]) {
    long long b = ww < nelem ? ww : wfirst;
    const int t0 = (int)(b % cr.count[0]); b /= cr.count[0];
    const int t1 = (int)(b % cr.count[1]); b /= cr.count[1];
    e3[0] = cr.start[0] + t0 * cr.step[0]; e3[1] = cr.start[1] + t1 * cr.step[1]; e3[2] = cr.start[2] + (int)b * cr.step[2];
  };
  const int ll = lane < EPW * NL ? lane - esub * NL : 0;                 // (the lanes beyond the last element idle on lane 0's indices)
  const int i0 = NS == 4 ? (ll & 3) : ll % 3, i1 = NS == 4 ? ((ll >> 2) & 3) : (ll / 3) % 3, i2 = NS == 4 ? (ll >> 4) : ll / 9;
  const VsLane VL = {i0, i1, i2, esub * NL};
  const int il[3] = {i0, i1, i2};
  const int op = out.op;
  const bool geo = GEO && S.nsd > 0, rat = GEO && S.rational != 0;
  const bool sysvec = out.vec_mode == 1, sysbody = out.vec_mode == 2;      // parts of a System assembly (igx.hpp: OutDev::vec_mode)
  const bool readU = out.U != nullptr && !sysvec && !sysbody, useU = readU || sysvec, useV = out.V != nullptr && !sysvec && !sysbody;
  int nb[3], nq[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) { nb[d] = S.ax[d].nen; nq[d] = S.ax[d].nqp; }
  const bool lanenode = i0 < nb[0] && i1 < nb[1] && i2 < nb[2], lanepoint = i0 < nq[0] && i1 < nq[1] && i2 < nq[2];
  // what an element needs from memory, in two steps: its offsets (A), then everything addressed through them (B)
  struct Pre { int el[3], off[3]; bool live; size_t row; double tab[TPL], U[DOF], V[DOF], Fo[DOF], fx[DOF], wgt, X[3], pt[3], wq[3], Jd[3]; };
  auto stageA = [&](long long un, Pre &P) {
    const long long w0n = un * EPW, wn = w0n + esub;
    P.live = lane < EPW * NL && wn < nelem;                             // (an odd count leaves the second half of the last wavefront idle)
    element_at(wn, w0n, P.el);
#pragma unroll
    for (int d = 0; d < 3; ++d) P.off[d] = S.ax[d].off_lin ? S.ax[d].off0 + P.el[d] : S.ax[d].off[P.el[d]];      // (closed form: the loads behind it leave with the first round trip)
  };
  auto stageB = [&](long long un, Pre &P) {
    const long long w0n = un * EPW;
    {      // 1-D rows of the element(s), [q][a][3], zero padded to 4 x 4 (the axis is a compile-time index: an index known only at run time sent nq / nb / the element to scratch and LDS)
      const int j = lane < TB ? lane : 0, q = j / (3 * NS), a = (j / 3) % NS, v = j % 3;
#pragma unroll
      for (int es = 0; es < EPW; ++es) {
        int e3[3]; element_at(w0n + es, w0n, e3);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          P.tab[es * 3 + d] = 0.0;
          if (lane < TB && q < nq[d] && a < nb[d]) P.tab[es * 3 + d] = S.ax[d].tab[((size_t)e3[d] * nq[d] * nb[d] + q * nb[d] + a) * NDER + v];
        }
      }
    }
    P.row = 0; P.wgt = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) P.X[c] = 0.0;
#pragma unroll
    for (int f = 0; f < DOF; ++f) { P.U[f] = 0; P.V[f] = 0; P.Fo[f] = 0; P.fx[f] = 0; }
    if (P.live && lanenode) {      // this lane's node: control point, state, old F (rowmap in closed form: AxisDev::rwrap)
      const int n0 = P.off[0] + i0, n1 = P.off[1] + i1, n2 = P.off[2] + i2;
      const size_t g = (size_t)n0 + (size_t)S.ax[0].gwidth * ((size_t)n1 + (size_t)S.ax[1].gwidth * (size_t)n2);
      const int r0 = n0 < S.ax[0].rwrap ? n0 : n0 - S.ax[0].rwrap, r1 = n1 < S.ax[1].rwrap ? n1 : n1 - S.ax[1].rwrap, r2 = n2 < S.ax[2].rwrap ? n2 : n2 - S.ax[2].rwrap;
      P.row = (size_t)r0 + (size_t)S.ax[0].nrow * ((size_t)r1 + (size_t)S.ax[1].nrow * (size_t)r2);
      P.wgt = rat ? S.W[g] : 1.0;
      if (geo) for (int c = 0; c < 3; ++c) P.X[c] = S.X[g * 3 + c];
#pragma unroll
      for (int f = 0; f < DOF; ++f) {
        if (readU) P.U[f] = out.U[P.row * DOF + f];
        if (useV) P.V[f] = out.V[P.row * DOF + f];
        if (S.fixtable) P.fx[f] = S.fixtable[P.row * DOF + f];
        if constexpr (PIPE || EF) P.Fo[f] = out.vec[P.row * DOF + f];
      }
    }
    if constexpr (PIPE) {      // (the geometry variants read these where they use them: no register to spare)
      const bool on = P.live && lanepoint;
      P.pt[0] = on ? S.ax[0].pt[P.el[0] * nq[0] + i0] : 0.0; P.pt[1] = on ? S.ax[1].pt[P.el[1] * nq[1] + i1] : 0.0; P.pt[2] = on ? S.ax[2].pt[P.el[2] * nq[2] + i2] : 0.0;
      P.wq[0] = on ? S.ax[0].w[P.el[0] * nq[0] + i0] : 0.0; P.wq[1] = on ? S.ax[1].w[P.el[1] * nq[1] + i1] : 0.0; P.wq[2] = on ? S.ax[2].w[P.el[2] * nq[2] + i2] : 0.0;
#pragma unroll
      for (int d = 0; d < 3; ++d) P.Jd[d] = S.ax[d].J[P.el[d]];
    } else if constexpr (EF) {
      const bool on = P.live && lanepoint;
      P.wq[0] = on ? (S.ax[0].w[P.el[0] * nq[0] + i0] * S.ax[0].J[P.el[0]]) * (S.ax[1].w[P.el[1] * nq[1] + i1] * S.ax[1].J[P.el[1]]) * (S.ax[2].w[P.el[2] * nq[2] + i2] * S.ax[2].J[P.el[2]]) : 0.0;
    }
  };
  Pre cur, nxt;
  if (PIPE) { stageA(unit, cur); stageB(unit, cur); }
  // One unit per wavefront.  Measured in round 6 (config 4's Residual, 256^3, 18.9 ms before): a wavefront walking a SEQUENCE of units with
  // the next one's loads in flight held 228 registers -- two wavefronts per SIMD -- and took 24.6 ms; TWO units as straight-line code
  // spilled 71 registers under the cap of 128 and took 32.6 ms; this, every load of the unit issued at its start, takes 15.6 ms.
  constexpr int UNITS = 1;
#pragma unroll
  for (int it = 0; it < UNITS; ++it) {
  const long long unext = unit + ustride; const bool more = PIPE && it + 1 < UNITS && unext < nunits;
  if (!PIPE) { stageA(unit, cur); stageB(unit, cur); }
  const bool live = cur.live;
  int el[3] = {cur.el[0], cur.el[1], cur.el[2]};
  VS_SYNC();      // (the previous element's readers of the rows are done)
#pragma unroll
  for (int k = 0; k < TPL; ++k) if (lane < TB) buf[NBUF * 64 + k * TB + lane] = cur.tab[k];
  if (PIPE && more) stageA(unext, nxt);
  // Dirichlet flags (IGAElementBuildFix / FixValues, src/petigaelem.c:1214-1358)
  const bool isnode = live && lanenode;
  const bool ispoint = live && lanepoint;
  const size_t row = cur.row; double Xw[3] = {cur.X[0] * cur.wgt, cur.X[1] * cur.wgt, cur.X[2] * cur.wgt}; const double wgt = cur.wgt; double Uv[DOF], Vv[DOF], ufix[DOF], Fold[DOF]; bool fixed[DOF];
#pragma unroll
  for (int f = 0; f < DOF; ++f) { Uv[f] = cur.U[f]; Vv[f] = cur.V[f]; Fold[f] = cur.Fo[f]; ufix[f] = 0; fixed[f] = false; }
  if (isnode) {
    if (op != OP_VECTOR || sysvec || sysbody) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {      // (unrolled: el[d] / il[d] stay registers -- indexed at run time the arrays went to scratch and LDS)
        const AxisDev &A = S.ax[d];
        if (A.periodic) continue;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
          if (el[d] + A.estart != (sd ? A.esizes - 1 : 0) || il[d] != (sd ? nb[d] - 1 : 0)) continue;
          const BCDev &bv = S.bcv[d][sd];
          for (int k = 0; k < bv.count; ++k) {
            const int f = bv.field[k];
#pragma unroll
            for (int ff = 0; ff < DOF; ++ff) if (ff == f) { fixed[ff] = true; ufix[ff] = S.fixtable ? cur.fx[ff] : bv.value[k]; }
          }
        }
      }
    }
  }
  VS_SYNC();
  // ---- forward: geometry (homogeneous coordinates), state
  double x[3], E1[9], E2[27], iw = 1.0, o1[3] = {0, 0, 0}, o2[9], detX = 1.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { E1[i] = (i % 4 == 0) ? 1.0 : 0.0; o2[i] = 0.0; }
#pragma unroll
  for (int i = 0; i < 27; ++i) E2[i] = 0.0;
  if constexpr (PIPE) { x[0] = cur.pt[0]; x[1] = cur.pt[1]; x[2] = cur.pt[2]; }
  else { x[0] = ispoint ? S.ax[0].pt[el[0] * nq[0] + i0] : 0.0; x[1] = ispoint ? S.ax[1].pt[el[1] * nq[1] + i1] : 0.0; x[2] = ispoint ? S.ax[2].pt[el[2] * nq[2] + i2] : 0.0; }
  // index of the second derivative (a, b) among the 10
  auto k2 = [](int a, int b) { const int lo = a < b ? a : b, hi = a < b ? b : a; return 4 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)); };
  if constexpr (GEO) if (geo || rat) {
    double Dw[10], Dx[3][10];
    if (UORD == 2) vs_forward<2, NS>(wgt, buf, tab0, tab1, tab2, lane, VL, Dw); else vs_forward<1, NS>(wgt, buf, tab0, tab1, tab2, lane, VL, Dw);
    const double W0 = ispoint ? Dw[0] : 1.0;
    iw = 1.0 / W0;
#pragma unroll
    for (int a = 0; a < 3; ++a) o1[a] = Dw[1 + a] * iw;
    if (UORD == 2)
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) o2[a * 3 + b] = Dw[k2(a, b)] * iw;
    if (geo) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { if (UORD == 2) vs_forward<2, NS>(Xw[c], buf, tab0, tab1, tab2, lane, VL, Dx[c]); else vs_forward<1, NS>(Xw[c], buf, tab0, tab1, tab2, lane, VL, Dx[c]); }
      double X1[9], X2[27];
#pragma unroll
      for (int c = 0; c < 3; ++c) {      // quotient rule on A = sum w X N, W = sum w N (src/petigarat.f90.in + petigamapgeo.f90.in)
        x[c] = Dx[c][0] * iw;
#pragma unroll
        for (int a = 0; a < 3; ++a) X1[c * 3 + a] = (Dx[c][1 + a] - x[c] * Dw[1 + a]) * iw;
        if (UORD == 2)
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) X2[c * 9 + a * 3 + b] = (Dx[c][k2(a, b)] - x[c] * Dw[k2(a, b)] - X1[c * 3 + a] * Dw[1 + b] - X1[c * 3 + b] * Dw[1 + a]) * iw;
      }
      if (!ispoint) { for (int i = 0; i < 9; ++i) X1[i] = (i % 4 == 0) ? 1.0 : 0.0; }
      detX = det3(X1, 3);
      inv3(X1, 3, detX, E1);
      if (ispoint && !(detX > 0.0)) atomicExch(out.errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
      if (UORD == 2) {   // InverseMap order 2 (src/petigamapinv.f90.in:32-45): E2[c][i][j] = -X2[k][a][b] E1[a][i] E1[b][j] E1[c][k]
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double U9[9], T9[9];
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) { double sm = 0; for (int b = 0; b < 3; ++b) sm += X2[k * 9 + a * 3 + b] * E1[b * 3 + j]; U9[a * 3 + j] = sm; }
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { double sm = 0; for (int a = 0; a < 3; ++a) sm += U9[a * 3 + j] * E1[a * 3 + i]; T9[i * 3 + j] = sm; }
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 9; ++i) E2[c * 9 + i] -= T9[i] * E1[c * 3 + k];
        }
      }
    }
  }
  double u[DOF], ut[DOF], gu[DOF * 3], hu[DOF * 9];
#pragma unroll
  for (int f = 0; f < DOF; ++f) {
    u[f] = 0; ut[f] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) gu[f * 3 + i] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) hu[f * 9 + i] = 0;
    if ((Form::NEED & (NEED_U | NEED_GU | NEED_HU)) && useU) {
      double D[10];
      vs_forward<UORD, NS>((fixed[f] ? ufix[f] : Uv[f]) * wgt, buf, tab0, tab1, tab2, lane, VL, D);
      u[f] = D[0] * iw;
      if constexpr (!GEO) {
        if (UORD >= 1) for (int a = 0; a < 3; ++a) gu[f * 3 + a] = D[1 + a];
        if (UORD == 2) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) hu[f * 9 + a * 3 + b] = D[k2(a, b)];
      } else
      if (UORD >= 1) {
        double u1[3], u2[9];
#pragma unroll
        for (int a = 0; a < 3; ++a) u1[a] = D[1 + a] * iw - u[f] * o1[a];      // u = A / W, du = (dA - u dW) / W (o1 = dW / W; 0 without weights)
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[f * 3 + i] = u1[0] * E1[0 * 3 + i] + u1[1] * E1[1 * 3 + i] + u1[2] * E1[2 * 3 + i];
        if (UORD == 2) {
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              double t = D[k2(a, b)] * iw;
              if (rat) t -= u[f] * o2[a * 3 + b] + u1[a] * o1[b] + u1[b] * o1[a];
              u2[a * 3 + b] = t;
            }
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {     // ShapeFunctions summed over a (src/petigamapshf.f90.in:30-58)
              double sm = 0;
#pragma unroll
              for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) sm += u2[a * 3 + b] * E1[a * 3 + i] * E1[b * 3 + j];
                sm += u1[a] * E2[a * 9 + i * 3 + j];
              }
              hu[f * 9 + i * 3 + j] = sm;
            }
        }
      }
    }
    if ((Form::NEED & NEED_UT) && useV) {
      double D[10];
      vs_forward<0, NS>((fixed[f] ? 0.0 : Vv[f]) * wgt, buf, tab0, tab1, tab2, lane, VL, D);
      ut[f] = D[0] * iw;
    }
  }
  // the next element's loads go out here: its offsets have arrived, and the point stage and the way back lie ahead
  if (PIPE && more) stageB(unext, nxt);
  // ---- point: JW and the form's vec() on the unit test features
  double JW = 0.0, G[9];
  if (ispoint) {
    JW = detX;
    if constexpr (EF && !PIPE) JW *= cur.wq[0];
    else {
#pragma unroll
    for (int d = 0; d < 3; ++d) JW *= PIPE ? cur.wq[d] * cur.Jd[d] : S.ax[d].w[el[d] * nq[d] + il[d]] * S.ax[d].J[el[d]];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < 3; ++i) G[a * 3 + i] = E1[a * 3 + i] / (PIPE ? cur.Jd[a] : S.ax[a].J[el[a]]);      // IGAPointFormInvGradGeomMap, src/petigapoint.c:269-294
  PtView p; p.x = x; p.u = u; p.ut = ut; p.gu = gu; p.hu = hu; p.G = G; p.prm = prm.v; p.shift = out.shift; p.t = out.t;
  p.normal = nullptr; p.atboundary = 0; p.boundary_id = -1;
  double Cq[DOF][10];
#pragma unroll
  for (int f = 0; f < DOF; ++f)
#pragma unroll
    for (int k = 0; k < 10; ++k) Cq[f][k] = 0.0;
#pragma unroll
  for (int tf = 0; tf < NFS; ++tf) {
    if (!((VMASK >> tf) & 1u)) continue;
    double e[NFS], R[DOF];
#pragma unroll
    for (int g = 0; g < NFS; ++g) e[g] = (g == tf) ? 1.0 : 0.0;
    Form::vec(p, e, R);
#pragma unroll
    for (int f = 0; f < DOF; ++f) {
      const double r = ispoint ? R[f] * JW : 0.0;      // (a padded lane evaluates vec() on zeros: its value may not even be finite)
      if (tf == 0) Cq[f][0] += r;
      else if (tf < 4) {      // physical gradient component tf-1 -> parametric: d_i R_a = sum_b E1[b][i] d_b R_a
        if constexpr (!GEO) Cq[f][tf] += r;
        else {
#pragma unroll
          for (int b = 0; b < 3; ++b) Cq[f][1 + b] += E1[b * 3 + (tf - 1)] * r;
        }
      } else {                // second-order test feature (i, j)
        const int i = (tf - 4) / 3, j = (tf - 4) % 3;
        if constexpr (!GEO) Cq[f][k2(i, j)] += r;      // (physical = parametric)
        else {
          // d_i d_j R_a = sum_ab E1[a][i] E1[b][j] d_a d_b R_a + sum_a E2[a][i][j] d_a R_a (ShapeFunctions order 2, src/petigamapshf.f90.in:30-58,
          // transposed: the slot of (a, b) takes both orders)
#pragma unroll
          for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) Cq[f][k2(a, b)] += E1[a * 3 + i] * E1[b * 3 + j] * r;
            Cq[f][1 + a] += E2[a * 9 + i * 3 + j] * r;
          }
        }
      }
    }
  }
  // rational test functions: R_a = w_a N_a / W, d_b R_a = (w_a / W) (d_b N_a - N_a W_b / W), d_b d_c R_a = (w_a / W) (d_b d_c N_a -
  // d_b N_a o_c - d_c N_a o_b - N_a (W_bc / W - 2 o_b o_c)) (src/petigarat.f90.in, order 2): the coefficients of the polynomial basis
  if (rat) {
#pragma unroll
    for (int f = 0; f < DOF; ++f) {
      double c0 = Cq[f][0] - (Cq[f][1] * o1[0] + Cq[f][2] * o1[1] + Cq[f][3] * o1[2]);
      if constexpr (GEO && SECOND_T) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = a; b < 3; ++b) {
            const double sk = Cq[f][k2(a, b)];      // (the slot holds both orders of (a, b))
            c0 -= sk * (o2[a * 3 + b] - 2.0 * o1[a] * o1[b]);
            if (a == b) Cq[f][1 + a] -= 2.0 * sk * o1[a];
            else { Cq[f][1 + a] -= sk * o1[b]; Cq[f][1 + b] -= sk * o1[a]; }
            Cq[f][k2(a, b)] = sk * iw;
          }
      }
      Cq[f][0] = c0 * iw; Cq[f][1] *= iw; Cq[f][2] *= iw; Cq[f][3] *= iw;
    }
  }
  // ---- backward, IGAElementFixFunction (src/petigaelem.c:1449-1461), IGAElementAssembleVec
#pragma unroll
  for (int f = 0; f < DOF; ++f) {
    double F = SECOND_T ? vs_backward<2, KMASK, NS>(Cq[f], buf, tab0, tab1, tab2, lane, VL) : vs_backward<1, KMASK, NS>(Cq[f], buf, tab0, tab1, tab2, lane, VL);
    F *= wgt;
    if (isnode) {
      if (fixed[f] && sysvec) F = ufix[f];                       // IGAElementFixSystem: F_e[k] = v
      else if (fixed[f] && sysbody) F = 0.0;                     // (the band-row kernel that follows adds v per element itself)
      else if (fixed[f] && (op == OP_FUNCTION || op == OP_IFUNCTION)) F = Uv[f] - ufix[f];
      if (F != 0.0) { if constexpr (PIPE || EF) out.vec[row * DOF + f] = Fold[f] + F; else out.vec[row * DOF + f] += F; }
    }
  }
  if (!more) break;
  if (PIPE) cur = nxt;
  unit = unext;
  }      // the wavefront's next element
}
#undef VS_SYNC

#ifndef IGX_RTC
// dim 3, at most 4 basis functions and 4 points per axis; vector-only drivers; no boundary loads (Function / IFunction subtract
// their lumped flux per element), no boundary-form passes
template <class Form>
static bool vec_sumfact_covers(const Space &s, const OutDev &out) {
  if constexpr (nscalar_of<Form>::v > 0 || has_boundary_of<Form>::v) return false;
  else {
    if (s.env.vec_sumfact == 0) return false;
    if (out.op != OP_VECTOR && out.op != OP_FUNCTION && out.op != OP_IFUNCTION && !out.vec_mode) return false;
    if (s.dim != 3 || s.dof != Form::DOF || (s.nsd != 0 && s.nsd != 3)) return false;
    for (int d = 0; d < 3; ++d) {
      if (s.basis[d].nen > 4 || s.basis[d].nqp > 4) return false;
      for (int sd = 0; sd < 2; ++sd) { if (s.visit[d][sd]) return false; if (out.op != OP_VECTOR && !out.vec_mode && s.load[d][sd].count) return false; }      // (vec_mode: the band-row launchers add the loads themselves)
    }
    return true;
  }
}

template <class Form>
static int try_vec_sumfact(const Space &s, const SpaceDev &S, const ParamsDev &prm, const OutDev &out, hipStream_t stream, std::string &kname, int &launches, bool &done) {
  done = false;
  if constexpr (nscalar_of<Form>::v > 0 || has_boundary_of<Form>::v) return 0;
  else {
  if (!vec_sumfact_covers<Form>(s, out)) return 0;
  launches = 0;
  const int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
      int first = -1, count = 0;
      for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
      if (count == 0) { empty = true; break; }
      cr.start[d] = first; cr.step[d] = L.p + 1; cr.count[d] = count;
    }
    if (empty) continue;
    const long long nelem = (long long)cr.count[0] * cr.count[1] * cr.count[2];
    // at most three basis functions and points per axis (p <= 2): two elements per wavefront (54 of 64 lanes instead of 27)
    bool three = !s.env.no_vec_pairs;
    for (int d = 0; d < 3; ++d) three = three && s.basis[d].nen <= 3 && s.basis[d].nqp <= 3;
    const bool g1 = s.nsd > 0 || s.rational;
    const bool onepass = true;      // (UNITS = 1 in the kernel)      // (one pass per wavefront: the launch has a wavefront per unit)
    if (three) {
      const unsigned grid = onepass ? (unsigned)((nelem + 7) / 8) : (unsigned)((nelem + 15) / 16);
      if (g1) hipLaunchKernelGGL((vec_sumfact<Form, true, 3>), dim3(grid), dim3(256), 0, stream, S, prm, out, cr, nelem);
      else hipLaunchKernelGGL((vec_sumfact<Form, false, 3>), dim3(grid), dim3(256), 0, stream, S, prm, out, cr, nelem);
    } else {
      const unsigned grid = onepass ? (unsigned)((nelem + 3) / 4) : (unsigned)((nelem + 7) / 8);
      if (g1) hipLaunchKernelGGL((vec_sumfact<Form, true, 4>), dim3(grid), dim3(256), 0, stream, S, prm, out, cr, nelem);
      else hipLaunchKernelGGL((vec_sumfact<Form, false, 4>), dim3(grid), dim3(256), 0, stream, S, prm, out, cr, nelem);
    }
    launches++;
  }
  if (hipGetLastError() != hipSuccess) return IGX_ERR_LIB;
  {
    bool three = !s.env.no_vec_pairs;
    for (int d = 0; d < 3; ++d) three = three && s.basis[d].nen <= 3 && s.basis[d].nqp <= 3;
    kname = three ? "vec_sumfact(vector only: sum factorisation forward and backward, two elements per wavefront)"
                  : "vec_sumfact(vector only: sum factorisation forward and backward, one wavefront per element)";
  }
  done = true;
  return 0;
  }
}

#endif   // !IGX_RTC

}  // namespace igx
