// gram_mfma.hpp -- MFMA kernels for "gradient-Gram" scalar forms: K_e = B^T diag(JW) B with
// B[(q,alpha),a] = d_alpha N_a(x_q)  (rows 11+13 of the reference fused).
// Covers demo/Poisson3D.c System (and test/IGAFixTable.c System2) on 3-D, p=3, 4x4x4 Gauss points,
// identity geometry: the BASELINE metric configuration.
//
// One wavefront owns one element.  K_e (64x64 fp64) lives in 16 accumulator tiles of
// v_mfma_f64_16x16x4_f64 (128 VGPRs); MFMA operands are never staged anywhere: each lane builds them
// from three 1-D table rows (tensor-product structure of K2, src/petiga3d.F90:32-233):
//   tile row/col slot   i = a1 + 4*a2   (lane&15),   tile index = a3 (wave-uniform)
//   k slot              q1              (lane>>4),   k-step = (q3,q2,alpha)
//   operand(q2,q3,alpha,t) = n0[q1][a1][alpha==0] * n1[q2][a2][alpha==1] * n2[q3][t][alpha==2]
// so a k-step costs ~11 v_mul_f64 for 16 MFMAs.  Result layout (measured, scripts/mfma_probe.hip):
// lane l, reg r of tile (ta,tb) holds K_e[a=(l>>4, r, ta)][b=(l&3, (l>>2)&3, tb)].
// IGX_RTC: the device half of this header is also compiled at run time for a user form (rtc.hpp: form_pencil below).
#pragma once
#ifndef IGX_RTC
#include <functional>
#include "first_touch.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <climits>
#include <cstdio>
#include <map>
#include <cstdlib>
#include <string>
#include <vector>
#endif
#include "igx.hpp"
#include "pencil_common.hpp"
#include "forms.hpp"

namespace igx {

constexpr int HOLD_LD = 66;   // padded row stride (doubles) of the pencil kernel's per-wavefront hold area [6 slots][4 r][HOLD_LD]

// the launches of the dominant kernel of one assembly, for the roofline line of bench.py
struct GramArgs {
  double forcing;      // F_a = forcing * int N_a   (1.0 for Poisson3D, -2*dim for IGAFixTable System2)
  int nwaves;          // elements in this launch
};

template <bool SYSTEM>
__global__ void __launch_bounds__(256, 2)
gram_p3_element(SpaceDev S, OutDev out, ColorRange cr, GramArgs ga) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = blockIdx.x * 4 + wave;
  if (w >= ga.nwaves) return;
  int el[3], ID[3], off[3];
  {
    int b = w;
    const int t0 = b % cr.count[0]; b /= cr.count[0];
    const int t1 = b % cr.count[1]; b /= cr.count[1];
    const int tt[3] = {t0, t1, b};
#pragma unroll
    for (int d = 0; d < 3; ++d) { el[d] = cr.start[d] + tt[d] * cr.step[d]; ID[d] = el[d] + S.ax[d].estart; off[d] = S.ax[d].off[el[d]]; }
  }
  const double *__restrict__ T0 = S.ax[0].tab + (size_t)el[0] * (4 * 4 * NDER);
  const double *__restrict__ T1 = S.ax[1].tab + (size_t)el[1] * (4 * 4 * NDER);
  const double *__restrict__ T2 = S.ax[2].tab + (size_t)el[2] * (4 * 4 * NDER);
  const double *__restrict__ Wq0 = S.ax[0].w + el[0] * 4;
  const double *__restrict__ Wq1 = S.ax[1].w + el[1] * 4;
  const double *__restrict__ Wq2 = S.ax[2].w + el[2] * 4;
  const double Jel = S.ax[0].J[el[0]] * S.ax[1].J[el[1]] * S.ax[2].J[el[2]];

  const int q1 = lane >> 4, i1 = lane & 3, i2 = (lane >> 2) & 3;   // k slot; tile row/col slot (a1|b1, a2|b2)
  // axis 0: value / derivative of basis i1 at point q1
  const double u0 = T0[(q1 * 4 + i1) * NDER + 0], u1 = T0[(q1 * 4 + i1) * NDER + 1];
  const double wq1 = Wq0[q1];
  // axis 1: basis i2 at the four points q2
  double v0[4], v1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { v0[q] = T1[(q * 4 + i2) * NDER + 0]; v1[q] = T1[(q * 4 + i2) * NDER + 1]; }

  d4_t acc[4][4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = (d4_t){0, 0, 0, 0};

#pragma unroll
  for (int q3 = 0; q3 < 4; ++q3) {
    double z0[4], z1[4];   // axis 2 (wave-uniform): basis t at point q3
#pragma unroll
    for (int t = 0; t < 4; ++t) { z0[t] = T2[(q3 * 4 + t) * NDER + 0]; z1[t] = T2[(q3 * 4 + t) * NDER + 1]; }
    const double s3 = Jel * Wq2[q3];
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      const double jw = wq1 * (Wq1[q2] * s3);
#pragma unroll
      for (int al = 0; al < 3; ++al) {
        const double uv = (al == 0 ? u1 : u0) * (al == 1 ? v1[q2] : v0[q2]);
        const double uvj = uv * jw;
        double opA[4], opB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const double z = (al == 2) ? z1[t] : z0[t]; opA[t] = uvj * z; opB[t] = uv * z; }
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[ta], opB[tb], acc[ta][tb], 0, 0, 0);
      }
    }
  }

  // ---- result coordinates of this lane
  const int a1 = lane >> 4, b1 = lane & 3, b2 = (lane >> 2) & 3;
  // Dirichlet data per face (IGAElementBuildFix, src/petigaelem.c:1214-1283); dof = 1 so field 0 only
  bool lo[3], hi[3]; double vlo[3], vhi[3]; bool anyfix = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = !S.ax[d].periodic && ID[d] == 0 && S.bcv[d][0].count > 0;
    hi[d] = !S.ax[d].periodic && ID[d] == S.ax[d].esizes - 1 && S.bcv[d][1].count > 0;
    vlo[d] = S.bcv[d][0].value[0]; vhi[d] = S.bcv[d][1].value[0];
    anyfix |= lo[d] | hi[d];
  }
  if (out.op == OP_MATRIX) anyfix = false;
  const int nr0 = S.ax[0].nrow, nr1 = S.ax[1].nrow;
  auto rowof = [&](int x, int y, int z) -> size_t {
    return (size_t)S.ax[0].rowmap[off[0] + x] + (size_t)nr0 * ((size_t)S.ax[1].rowmap[off[1] + y] + (size_t)nr1 * (size_t)S.ax[2].rowmap[off[2] + z]);
  };
  // fixed flag / value of a local basis function (x,y,z); later faces override earlier ones
  auto fixinfo = [&](int x, int y, int z, double &val) -> bool {
    bool f = false; const int c[3] = {x, y, z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (lo[d] && c[d] == 0) { f = true; val = vlo[d]; }
      if (hi[d] && c[d] == 3) { f = true; val = vhi[d]; }
    }
    return f;
  };

  __shared__ double s_corr[4][64];
  if (SYSTEM) {
    // F_a = forcing * J * prod_d sum_q w_d[q] N_d[q][a_d]   (tensor-product form of int N_a)
    const int fa0 = lane & 3, fa1 = (lane >> 2) & 3, fa2 = lane >> 4;
    double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s0 += Wq0[q] * T0[(q * 4 + fa0) * NDER];
      s1 += Wq1[q] * T1[(q * 4 + fa1) * NDER];
      s2 += Wq2[q] * T2[(q * 4 + fa2) * NDER];
    }
    double F = ga.forcing * (Jel * (s0 * s1 * s2));
    if (anyfix) {
      // F_a -= sum_{b fixed} K_ab v_b  (IGAElementFixSystem, src/petigaelem.c:1377-1387)
      double fv[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) { double v = 0; fv[tb] = fixinfo(b1, b2, tb, v) ? v : 0.0; }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double part = 0;
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) part += acc[ta][tb][r] * fv[tb];
          part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4); part += __shfl_xor(part, 8);
          if ((lane & 15) == 0) s_corr[wave][a1 + 4 * r + 16 * ta] = part;
        }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double v = 0;
      const bool fx = fixinfo(fa0, fa1, fa2, v);
      F = fx ? v : F - s_corr[wave][lane];
    }
    out.vec[rowof(fa0, fa1, fa2)] += F;
  }
  if (anyfix) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double va; const bool fa = fixinfo(a1, r, ta, va);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          double vb; const bool fb = fixinfo(b1, b2, tb, vb);
          if (fa || fb) acc[ta][tb][r] = (a1 == b1 && r == b2 && ta == tb) ? 1.0 : 0.0;
        }
      }
  }

  // ---- scatter (IGAElementAssembleMat, src/petigaelem.c:1542): conflict-free inside a colour
  const int i0 = off[0] + a1;
  const int c0 = S.ax[0].rcnt[S.ax[0].rowmap[i0]];
  const int P0 = S.ax[0].P[i0 * 7 + (b1 - a1 + 3)];
  // 16 loads in flight per batch (the compiler cannot hoist a load above an earlier store to an unknown
  // address, so a naive `*dst += x` chain is one HBM round trip per entry)
  int P2t[4][4];
  size_t k1c[4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta) {
    k1c[ta] = (size_t)S.ax[2].rowmap[off[2] + ta];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) P2t[ta][tb] = S.ax[2].P[(off[2] + ta) * 7 + (tb - ta + 3)];
  }
  const size_t r0c = (size_t)S.ax[0].rowmap[i0];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j1 = off[1] + r;
    const size_t r1c = (size_t)S.ax[1].rowmap[j1];
    const int c1 = S.ax[1].rcnt[r1c];
    const int P1 = S.ax[1].P[j1 * 7 + (b2 - r + 3)];
    double *dst[4][4]; double old[4][4];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
      const size_t base = (size_t)out.browptr[r0c + (size_t)nr0 * (r1c + (size_t)nr1 * k1c[ta])];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) dst[ta][tb] = out.val + base + ((size_t)P2t[ta][tb] * c1 + P1) * c0 + P0;
    }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) old[ta][tb] = *dst[ta][tb];
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) *dst[ta][tb] = old[ta][tb] + acc[ta][tb][r];
  }
}


// ======================================================================================================
// Pencil kernel: one wavefront walks a pencil of elements along mesh axis 0 and lets the MFMA accumulators
// combine the contributions of consecutive elements before anything is written.
//
// Roles of the mesh axes here:  W = axis 0 (walked; accumulator tile index, wave-uniform),
//                               X = axis 1 (tile row slot lane>>4 / column slot lane&3 / k slot lane>>4),
//                               Y = axis 2 (tile row = accumulator register r / column (lane>>2)&3).
// A tile (ta,tb) belongs to a pair of axis-0 node layers.  An element covers four consecutive layers; its
// local basis a0 sits in tile slot a0.  After an element, its first layer leaves the window: the 7 tiles
// with row- or column-slot 0 are complete for this pencil, are added to the CSR (row layer l x col layers
// l..l+3, and row layers l+1..l+3 x col layer l), and the window slides (tile (ta,tb) <- (ta+1,tb+1)).
// Per element 7*256 = 1792 entries reach memory instead of 4096, and two pencils conflict only if they
// share axis-1/axis-2 nodes: 16 colours instead of 64.  Walking axis 0 keeps a wavefront on the same 16
// CSR row neighbourhoods for the whole pencil (consecutive layers are consecutive CSR rows), which is what
// makes the read-modify-write TLB- and DRAM-page-friendly; walking axis 2 (rows 184 MB apart at 256^3)
// measured 45k cycles per flush.
//
// Long pencils are cut into segments.  A segment owns the row layers of its own elements; it first
// re-computes (without writing) the <=3 elements before its start so that its first rows are complete,
// and discards rows it does not own.  No colouring along the walk.
// Elements that touch a Dirichlet face are not walked here: they go to gram_p3_element (BC logic there).
// ======================================================================================================
struct PencilArgs {
  double forcing;
  int ex_start, ex_step, ex_count;   // pencils of this colour: local element indices on the two non-walked axes
  int ey_start, ey_step, ey_count;
  int w_lo, w_hi;          // walked range of local elements along the walk axis: [w_lo, w_hi)
  int seg_len, nseg;
  int blocks_per_seg;      // ceil(pencils / 8): a workgroup (8 wavefronts = 8 pencils) never straddles segments
  int ne_max;              // LDS capacity: elements (seg_len + 3 halo) ; layers = ne_max + 3
  int alias0;              // the walk axis is periodic and wrapped inside the rank: elements and node layers modulo its length; every
                           // segment re-computes the P elements before its start (the first one those at the end of the axis) and
                           // nobody owns rows beyond its own elements: each band row is still written exactly once per pencil
  int first_touch;         // 1: the matrix was NOT zeroed; the first colour that reaches an entry stores it (walk axis 0 only)
  int nelx, nely;          // local element counts on the two non-walked axes (for the first-touch rule)
  int fty_lo, fty_hi, fty_blocked;   // first-touch rule on the Y axis when the launches of an assembly come in several passes (see launch_pencils)
  int ftx_lo, ftx_hi, ftx_blocked;   // ... and on the X axis
  int w_halo_lo;           // lowest element a segment may re-compute as halo (= w_lo unless the walked range is the upper part of the axis: a face pass)
  int open_hi;             // 1: elements beyond w_hi exist and belong to another pass: the last segment owns no rows past its elements
  int free_run;            // 1: no s_barrier ping-pong between the two wave groups: the SIMD's own arbitration interleaves MFMA and flush phases (IGX_FREE_RUN; default: p = 2 on the identity geometry)
  int wpb;                 // wavefronts (= pencils) per workgroup: 8; 6 for the free-running p = 2 walk on the identity geometry (gram_pencil_w6: three waves per SIMD)
  int debug_noflush;       // experiment switch: 1 = skip the read-modify-write (timing of the MFMA walk alone)
  long long *debug_buf;    // experiment: cycle stamps [block][wave 0 and 4][64 steps][4]
};

struct PencilLane {        // per-lane constants of a pencil
  double u0, u1;           // X-axis basis value / derivative of this lane's (qx, ix), scaled by sqrt(w_qx * J_x)
  const double *vy;        // LDS: this lane's Y-axis basis row [4 q][2] (value, derivative)
  long long psx; int cx, px;            // X axis: row prefix / row length of this lane's row slot, column position of its column slot
  long long psy[4]; int cy[4], py[4];   // Y axis, per accumulator register r
  double sxy;              // forcing * sum_q wx Nx[fx] * sum_q wy Ny[fy] for the F lane
  long long frowxy;        // F row without the walk-axis part
  int fslot;
  unsigned stmask;         // bit r: this lane's entries of register r are stored without a read (first touch)
};

// LDS-staged axis-0 data of one segment (the knot-span tables of the walk): per element the 1-D basis rows
// (value, derivative), Gauss weights and half-length; per node layer the row prefix, row length and the 7
// column positions of the structured CSR.
struct PencilLds {
  double *zt;        // [ne][4 q][4 a][2]
  double *wq;        // [ne][4]
  double *Jz;        // [ne]
  long long *pre;    // [nl]  prefix0[rho0]
  int *cnt;          // [nl]  rcnt0[rho0]  (-1: layer does not exist)
  int *rho;          // [nl]
  int *P;            // [nl][8]
  int lay0;          // first layer held
};

// geo: the mapped-geometry variant reads the (unscaled) walk-axis rows from global memory instead: no zt
__device__ __forceinline__ PencilLds pencil_lds_carve(double *sm, int ne_max, bool geo = false) {
  PencilLds t; const int nl = ne_max + 3;
  t.zt = sm; t.wq = t.zt + (geo ? 0 : ne_max * 32); t.Jz = t.wq + ne_max * 4;
  t.pre = reinterpret_cast<long long *>(t.Jz + ((ne_max + 1) & ~1));
  t.cnt = reinterpret_cast<int *>(t.pre + nl); t.rho = t.cnt + nl; t.P = t.rho + nl; t.lay0 = 0;
  return t;
}
__host__ __device__ static inline size_t pencil_lds_bytes(int ne_max, bool geo = false, int wpb = 8) {
  const int nl = ne_max + 3;
  const size_t tables = (size_t)((geo ? 0 : ne_max * 32) + ne_max * 4 + ((ne_max + 1) & ~1)) * 8 + (size_t)nl * 8 + (size_t)nl * 4 * 10 + 64;
  return ((tables + 15) & ~(size_t)15) + (size_t)2 * wpb * 32 * 8;   // + per-wavefront Y-axis and X-axis basis rows [wpb waves][4 a][4 q][2]
}
// mapped geometry: per-wavefront [64 points][7] (JW * F^-1 F^-T: 00,01,02,11,12,22; forcing * JW / W) + [64][4] (1/W, dW/W);
// the element's control points (homogeneous, [aw][ay][ax][4]) are staged at its start and overwritten by the results
constexpr int GEO_M = 7, GEO_Z = 64 * GEO_M + 64 * 4, GEO_DOUBLES = GEO_Z + 32;   // + the element's walk-axis rows [q][a][2] (zero padded)
__host__ __device__ static inline size_t pencil_geo_bytes() { return (size_t)8 * GEO_DOUBLES * 8; }
// walk along axis 0 only: per-wavefront hold area for the lower-band entries [P(P+1)/2 slots][4 r][HOLD_LD lanes]
__host__ __device__ static inline size_t pencil_hold_bytes(int P) { return (size_t)8 * (P * (P + 1) / 2) * 4 * HOLD_LD * 8; }

// 768 MFMAs of one element: k-step (qw, qy, alpha), k slot = qx (lane>>4).  K_e = sum_q (sqrt(JW) grad N_a).(sqrt(JW) grad N_b):
// the quadrature weight is split as sqrt(JW_q) on both sides, and sqrt(JW_q) itself factorises over the axes, so the
// three 1-D rows are pre-scaled once (u: per pencil, vy: per pencil in LDS, zt: when the segment is staged) and the
// A and B operands of a tile pair are the same registers: 5 v_mul_f64 per 16 MFMAs.
template <int W, bool SYM, int NB>
__device__ __forceinline__ void pencil_mfma(d4_t (&acc)[4][4], const PencilLane &L, const double *zt /*LDS [4][4][2], pre-scaled, zero padded*/) {
#pragma unroll
  for (int qw = 0; qw < NB; ++qw) {
    double z0[NB], z1[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { z0[t] = zt[(qw * 4 + t) * 2 + 0]; z1[t] = zt[(qw * 4 + t) * 2 + 1]; }
#pragma unroll
    for (int qy = 0; qy < NB; ++qy) {
      const double vy0 = L.vy[qy * 2 + 0], vy1 = L.vy[qy * 2 + 1];
#pragma unroll
      for (int al = 0; al < 3; ++al) {   // al = 0: d/dw, 1: d/dx, 2: d/dy (the Gram sum runs over all three)
        const double uv = (al == 1 ? L.u1 : L.u0) * (al == 2 ? vy1 : vy0);
        double op[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) op[t] = uv * ((al == 0) ? z1[t] : z0[t]);
#pragma unroll
        for (int ta = 0; ta < NB; ++ta)
#pragma unroll
          for (int tb = SYM ? ta : 0; tb < NB; ++tb)   // SYM: K_e is symmetric, tile (tb,ta) is the transpose of (ta,tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[ta], op[tb], acc[ta][tb], 0, 0, 0);
      }
    }
  }
}

// p = 2: the k-step (qw, qy, alpha) with k slot qx leaves one slot in four empty (27 x 6 = 162 MFMAs per element); numbering the 27
// points 4 per k-step -- point = 4 * step + (lane>>4), every row value the lane's own LDS read -- takes 7 x 3 x 6 = 126
__device__ __forceinline__ void pencil_mfma_p2(d4_t (&acc)[4][4], const double *uxs /*LDS [q][a][2], pre-scaled*/, const double *vy /*this lane's [q][2]*/,
                                               const double *zt /*LDS [q][a][2], pre-scaled*/, int lane) {
  const int ks = lane >> 4, ix = lane & 3;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = rem - 3 * qy;
    const double u0 = on ? uxs[(qx * 4 + ix) * 2 + 0] : 0.0, u1 = on ? uxs[(qx * 4 + ix) * 2 + 1] : 0.0;
    const double v0 = vy[qy * 2 + 0], v1 = vy[qy * 2 + 1];
    const double cw = u0 * v0, cx = u1 * v0, cy = u0 * v1;
    double op[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { const double z0 = zt[(qw * 4 + t) * 2 + 0], z1 = zt[(qw * 4 + t) * 2 + 1]; op[0][t] = cw * z1; op[1][t] = cx * z0; op[2][t] = cy * z0; }
#pragma unroll
    for (int al = 0; al < 3; ++al)
#pragma unroll
      for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int tb = ta; tb < 3; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[al][ta], op[al][tb], acc[ta][tb], 0, 0, 0);
  }
}

// ---- p = 2 on the identity geometry, PACKED (gram_pencil_w6).  The layout above puts a 3 x 3 block of (a_x, a_y) into the 4 x 4 row
// slots of a tile and one tile per pair of node layers: 9 x 9 of 16 x 16 entries of every MFMA are real, 2.2 x the algorithmic
// flops.  Here the element's 27 basis functions are numbered a = 9 a_w + 3 a_y + a_x and packed into TWO tiles of 16 rows (27 of 32
// rows real), K_e is 2 x 2 tiles of which the 3 with Ta <= Tb are computed: 3 MFMAs per k-step instead of 6, 63 per element
// instead of 126.  The price: a tile no longer IS a pair of node layers, so the sum over the elements of the pencil (the combine
// before the write) cannot stay in the accumulators; it moves into a per-wavefront LDS window of band rows,
//   win[3 node layers (ring)][9 (a_y, a_x)][5 column layers][9 (b_y, b_x)],
// to which every lane adds its 16 entries with ds_add_f64 (both (a, b) and (b, a) for the off-diagonal tile), and from which the
// leaving layer's band row -- all five column layers: no transposed half to park -- is read by the flush (pencil0_leave<LDSWIN>).
constexpr int WIN_ROW = 45, WIN_LAYER = 9 * WIN_ROW, WIN_DOUBLES = 3 * WIN_LAYER + 1;      // 1216 doubles per wavefront
__host__ __device__ static inline size_t pencil_win_bytes(int wpb) { return (size_t)wpb * WIN_DOUBLES * 8; }

// PK = false: the indices as 30 ints (config 2 and the Tangent on the identity geometry: registers to spare, and the decode's integer
// work in the MFMA phase cost them 8-9 %); PK = true: packed into 10 (the fused pass and the Tangents on mapped geometries, which
// spilled them: xy | aw << 4, negative for a function of the padding) and decoded at the adds.
template <bool PK> struct P2kLaneT;
template <> struct P2kLaneT<false> {
  int ua[2], va[2], za[2];     // this lane's operand row of tile T = function 16 T + (lane & 15): offsets (doubles) into the X rows [q][a][2], Y rows [a][q][2], walk rows [q][a][2]
  int rs_[2][4], cs_[2][4], aw_[2][4];   // the lane's result rows a = 16 T + 4 i + (lane >> 4): window row part 45 xy - 9 aw, column part 9 aw + xy + 18, layer slot (-1: padding)
  int rsc_[2], csc_[2], awc_[2];  // ... and its result column b = 16 T + (lane & 15)
  int o;
  __device__ __forceinline__ void set(int T, int i, int xy, int aw, bool ok) { aw_[T][i] = ok ? aw : -1; rs_[T][i] = xy * 45 - 9 * aw; cs_[T][i] = 9 * aw + xy + 18; }
  __device__ __forceinline__ void setc(int T, int xy, int aw, bool ok) { awc_[T] = ok ? aw : -1; rsc_[T] = xy * 45 - 9 * aw; csc_[T] = 9 * aw + xy + 18; }
  __device__ __forceinline__ int aw(int T, int i) const { return aw_[T][i]; }
  __device__ __forceinline__ int rs(int T, int i) const { return rs_[T][i]; }
  __device__ __forceinline__ int cs(int T, int i) const { return cs_[T][i]; }
  __device__ __forceinline__ int xy(int T, int i) const { return cs_[T][i] - 18 - 9 * (aw_[T][i] < 0 ? 0 : aw_[T][i]); }
  __device__ __forceinline__ int awc(int T) const { return awc_[T]; }
  __device__ __forceinline__ int rsc(int T) const { return rsc_[T]; }
  __device__ __forceinline__ int csc(int T) const { return csc_[T]; }
};
template <> struct P2kLaneT<true> {
  int ua[2], va[2], za[2];
  // one int per result row: (rs + 18) | xy << 9 | aw << 13 (aw = 7: padding), decoded with one or two bit-field instructions where it
  // is used; the result columns keep plain ints (four registers)
  int code[2][4], csc_[2], awc_[2];
  // `o`: a zero the compiler cannot see through, made once per element of the walk: the decode then stays inside the walk loop --
  // hoisted out of it, it IS the registers it saves -- while its instructions remain free to be scheduled.
  int o;
  __device__ __forceinline__ void set(int T, int i, int xy, int aw, bool ok) { code[T][i] = (xy * 45 - 9 * aw + 18) | (xy << 9) | ((ok ? aw : 7) << 13); }
  __device__ __forceinline__ void setc(int T, int xy, int aw, bool ok) { awc_[T] = ok ? aw : -1; csc_[T] = 9 * aw + xy + 18; }
  __device__ __forceinline__ int aw(int T, int i) const { const int a = (code[T][i] | o) >> 13; return a == 7 ? -1 : a; }
  __device__ __forceinline__ int rs(int T, int i) const { return ((code[T][i] | o) & 511) - 18; }
  __device__ __forceinline__ int xy(int T, int i) const { return ((code[T][i] | o) >> 9) & 15; }
  __device__ __forceinline__ int cs(int T, int i) const { const int c = code[T][i] | o, a = c >> 13; return 9 * (a == 7 ? 0 : a) + ((c >> 9) & 15) + 18; }
  __device__ __forceinline__ int awc(int T) const { return awc_[T]; }
  __device__ __forceinline__ int rsc(int T) const { return (csc_[T] - 18 - 9 * (awc_[T] < 0 ? 0 : awc_[T])) * 45 - 9 * (awc_[T] < 0 ? 0 : awc_[T]); }
  __device__ __forceinline__ int csc(int T) const { return csc_[T]; }
};
template <bool PK>
__device__ __forceinline__ P2kLaneT<PK> pencil_p2k_lane(int lane) {
  static_assert(WIN_ROW == 45, "the window's row stride is part of the packed indices");
  P2kLaneT<PK> K; K.o = 0;
  auto split = [](int f, int &aw, int &ay, int &ax) { aw = f / 9; const int r = f - 9 * aw; ay = r / 3; ax = r - 3 * ay; };
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int f = 16 * T + (lane & 15);
    int aw, ay, ax; split(f < 27 ? f : 0, aw, ay, ax);
    K.setc(T, 3 * ay + ax, aw, f < 27);
    if (f >= 27) ax = 3;                       // the zero-padded slot of the X rows: a padding row contributes nothing
    K.ua[T] = ax * 2; K.va[T] = ay * 8; K.za[T] = aw * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = 16 * T + 4 * i + (lane >> 4);
      int bw, by, bx; split(g < 27 ? g : 0, bw, by, bx);
      K.set(T, i, 3 * by + bx, bw, g < 27);
    }
  }
  return K;
}

template <class KL>
__device__ __forceinline__ void pencil_mfma_p2k(d4_t (&pk)[3], const double *uxs /*LDS [q][a][2], pre-scaled, zero padded*/, const double *vys /*LDS [a][q][2]*/,
                                                const double *zt /*LDS [q][a][2]*/, const KL &K, int lane) {
  const int ks = lane >> 4;
  pk[0] = pk[1] = pk[2] = (d4_t){0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = on ? rem - 3 * qy : 3;      // (q_x = 3: the zero-padded point of the X rows)
    double op[2][3];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      const d2u_t u = *reinterpret_cast<const d2u_t *>(uxs + qx * 8 + K.ua[T]);
      const d2u_t v = *reinterpret_cast<const d2u_t *>(vys + K.va[T] + qy * 2);
      const d2u_t z = *reinterpret_cast<const d2u_t *>(zt + qw * 8 + K.za[T]);
      const double cw = u[0] * v[0], cx = u[1] * v[0], cy = u[0] * v[1];
      op[T][0] = cw * z[1]; op[T][1] = cx * z[0]; op[T][2] = cy * z[0];
    }
#pragma unroll
    for (int al = 0; al < 3; ++al) {
      pk[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[0][al], op[0][al], pk[0], 0, 0, 0);
      pk[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[0][al], op[1][al], pk[1], 0, 0, 0);
      pk[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[1][al], op[1][al], pk[2], 0, 0, 0);
    }
  }
}

// the element's entries into the window: lane holds K[16 Ta + 4 i + (lane >> 4)][16 Tb + (lane & 15)] in pk[tile][i]; li = the element's
// first node layer (segment-local): the row of layer li + aw lives in ring slot (li + aw) % 3
template <class KL>
__device__ __forceinline__ void pencil_win_add(double *win, const d4_t (&pk)[3], const KL &K, int li) {
  const int e3 = li % 3;
  const int ro0 = e3 * WIN_LAYER, ro1 = (e3 == 2 ? 0 : e3 + 1) * WIN_LAYER, ro2 = (e3 == 0 ? 2 : e3 - 1) * WIN_LAYER;
  auto ring = [&](int aw) { return aw == 0 ? ro0 : (aw == 1 ? ro1 : ro2); };
  auto add = [&](int idx, double v) { (void)__hip_atomic_fetch_add(win + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int Ta = t == 2 ? 1 : 0, Tb = t == 0 ? 0 : 1;
    if (K.awc(Tb) < 0) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (K.aw(Ta, i) < 0) continue;
      add(ring(K.aw(Ta, i)) + K.rs(Ta, i) + K.csc(Tb), pk[t][i]);
      if (t == 1) add(ring(K.awc(Tb)) + K.rsc(Tb) + K.cs(Ta, i), pk[t][i]);      // the mirror entry of the off-diagonal tile
    }
  }
}

// ... of a Tangent (not symmetric: all four tiles pk[2 Ta + Tb], no mirror entries)
template <class KL>
__device__ __forceinline__ void pencil_win_add_ns(double *win, const d4_t (&pk)[4], const KL &K, int li) {
  const int e3 = li % 3;
  const int ro0 = e3 * WIN_LAYER, ro1 = (e3 == 2 ? 0 : e3 + 1) * WIN_LAYER, ro2 = (e3 == 0 ? 2 : e3 - 1) * WIN_LAYER;
  auto ring = [&](int aw) { return aw == 0 ? ro0 : (aw == 1 ? ro1 : ro2); };
#pragma unroll
  for (int Ta = 0; Ta < 2; ++Ta)
#pragma unroll
    for (int Tb = 0; Tb < 2; ++Tb) {
      if (K.awc(Tb) < 0) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (K.aw(Ta, i) < 0) continue;
        (void)__hip_atomic_fetch_add(win + ring(K.aw(Ta, i)) + K.rs(Ta, i) + K.csc(Tb), pk[Ta * 2 + Tb][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
}

// The Residual of a fused pass (state_pencil_kr): column 27 of tiles (0, 1) and (1, 1) is R_a of the element (pencil_mfma_state_p2k<RESID>),
// held by the lanes with (lane & 15) == 11: row a = 16 Ta + 4 i + (lane >> 4).  It is summed over the pencil's elements in a ring of
// its own, rwin[3 node layers][9 (a_y, a_x)] (+ padding: 32 doubles per wavefront), like the band rows in theirs.
template <class KL>
__device__ __forceinline__ void pencil_rwin_add(double *rwin, const d4_t (&pk)[4], const KL &K, int li, int lane) {
  if ((lane & 15) != 11) return;
  const int e3 = li % 3;
  const int ro0 = e3 * 9, ro1 = (e3 == 2 ? 0 : e3 + 1) * 9, ro2 = (e3 == 0 ? 2 : e3 - 1) * 9;
#pragma unroll
  for (int Ta = 0; Ta < 2; ++Ta)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (Ta == 1 && i == 3) continue;               // rows 28..31: padding in every lane
      const int aw = K.aw(Ta, i);                     // (-1 only at Ta = 1, i = 2, lane >> 4 == 3: row 27; it adds 0 to the pad slot 27)
      const int xy = K.xy(Ta, i);
      const int idx = aw < 0 ? 27 : (aw == 0 ? ro0 : (aw == 1 ? ro1 : ro2)) + xy;
      (void)__hip_atomic_fetch_add(rwin + idx, aw < 0 ? 0.0 : pk[Ta * 2 + 1][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}
// The COMPACT window (12 blocks of 9 x 9 instead of 15), for kernels whose LDS is short (state_pencil_geo_k).  Between the leave of layer
// e - 1 and the leave of layer e the live (row layer, column layer) pairs are: row e with d = 0..4, row e + 1 with d = 0..3, row e + 2
// with d = 0..2 (d = column layer - row layer + 2).  So d <= 2 gets a private block per ring slot of the row, the two live d = 3 blocks
// (rows e, e + 1) alternate between two blocks by the parity of the row layer, and the one live d = 4 block has its own:
//   block(row layer r, d) = d <= 2 ? 3 (r % 3) + d : (d == 3 ? 9 + (r & 1) : 11),   offset = 81 block + 9 xy_row + xy_col.
constexpr int WINC_DOUBLES = 12 * 81 + 4;      // 976 doubles per wavefront
__host__ __device__ static inline size_t pencil_winc_bytes(int wpb) { return (size_t)wpb * WINC_DOUBLES * 8; }
__device__ __forceinline__ int pencil_winc_block(int r, int d) { return d <= 2 ? 3 * (r % 3) + d : (d == 3 ? 9 + (r & 1) : 11); }
// a Tangent's four tiles into the compact window: li = the element's first node layer (segment-local)
template <class KL>
__device__ __forceinline__ void pencil_winc_add_ns(double *win, const d4_t (&pk)[4], const KL &K, int li) {
  // (aw, xy) of a function from its window parts: rs = 45 xy - 9 aw, cs = 9 aw + xy + 18
#pragma unroll
  for (int Ta = 0; Ta < 2; ++Ta)
#pragma unroll
    for (int Tb = 0; Tb < 2; ++Tb) {
      if (K.awc(Tb) < 0) continue;
      const int xyb = K.csc(Tb) - 18 - 9 * K.awc(Tb);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int awa = K.aw(Ta, i);
        if (awa < 0) continue;
        const int xya = K.cs(Ta, i) - 18 - 9 * awa;
        const int off = 81 * pencil_winc_block(li + awa, K.awc(Tb) - awa + 2) + 9 * xya + xyb;
        (void)__hip_atomic_fetch_add(win + off, pk[Ta * 2 + Tb][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
}

// the 7 tiles that leave with layer `lay`: k = 0..3 -> (row lay, col lay+k), k = 4..6 -> (row lay+k-3, col lay);
// position of entry (row layer rl, this lane's (a,r) ; col layer cl, this lane's (b1,b2)):
//   pos = L.A[r] + L.C[r]*prefix0[rl] + L.B[r]*cnt0[rl] + P0[rl][cl-rl+3]
struct FlushPlan { bool on[7]; long long pre[7]; int cnt[7], p0[7]; };

__device__ __forceinline__ FlushPlan pencil_plan(const PencilLds &T, int nl, int lay, int own_lo, int own_hi) {
  FlushPlan f;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int rl = (k < 4) ? lay : lay + (k - 3), cl = (k < 4) ? lay + k : lay;
    const int ri = rl - T.lay0, ci = cl - T.lay0;
    bool ok = ri >= 0 && ri < nl && ci >= 0 && ci < nl && rl >= own_lo && rl < own_hi;
    int p0 = -1, c0 = 0; long long pre = 0;
    if (ok) { c0 = T.cnt[ri]; ok = c0 > 0 && T.cnt[ci] > 0; }
    if (ok) { p0 = T.P[ri * 8 + (cl - rl + 3)]; pre = T.pre[ri]; ok = p0 >= 0; }
    // everything here is wave-uniform: pin it to SGPRs (LDS reads come back in VGPRs)
    const int lo = __builtin_amdgcn_readfirstlane((int)(pre & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(pre >> 32));
    f.on[k] = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    f.pre[k] = ((long long)hi << 32) | (unsigned int)lo;
    f.cnt[k] = __builtin_amdgcn_readfirstlane(c0); f.p0[k] = __builtin_amdgcn_readfirstlane(p0);
  }
  return f;
}

// block-CSR position of an entry from the per-axis (row prefix, row length, column position) triples:
//   pos = ps2*T1*T0 + c2*(ps1*T0 + c1*ps0) + (P2*c1 + P1)*c0 + P0         (see k_browptr / IGXCreateMat)
template <int W>
__device__ __forceinline__ long long pencil_pos(const PencilLane &L, int r, long long psw, int cw, int pw, long long T0, long long T10) {
  long long ps[3]; int c[3], P[3];
  constexpr int X = (W == 0) ? 1 : 0, Y = (W == 2) ? 1 : 2;
  ps[W] = psw; c[W] = cw; P[W] = pw;
  ps[X] = L.psx; c[X] = L.cx; P[X] = L.px;
  ps[Y] = L.psy[r]; c[Y] = L.cy[r]; P[Y] = L.py[r];
  return ps[2] * T10 + (long long)c[2] * (ps[1] * T0 + (long long)c[1] * ps[0]) + ((long long)P[2] * c[1] + P[1]) * c[0] + P[0];
}

// read-modify-write of the 7 leaving tiles (k = 0..3 is tile (0,k), k = 4..6 is tile (k-3,0)) and of the F entry:
// all 29 loads are in flight together (one memory round trip), then the adds and the stores.  The partner
// wavefront on this SIMD issues its 768 MFMAs meanwhile (ping-pong schedule in the kernel).
template <bool SYSTEM, int W>
__device__ __forceinline__ void pencil_flush(const d4_t (&acc)[4][4], const PencilLane &L, const FlushPlan &f, const OutDev &out,
                                             long long T0, long long T10, bool fdo, long long frow, double Facc) {
  double old[7][4], Fold = 0;
  if (SYSTEM && fdo) Fold = out.vec[frow];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if (!f.on[k]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) old[k][r] = out.val[pencil_pos<W>(L, r, f.pre[k], f.cnt[k], f.p0[k], T0, T10)];
  }
  if (SYSTEM && fdo) out.vec[frow] = Fold + Facc;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int ta = (k < 4) ? 0 : k - 3, tb = (k < 4) ? k : 0;
    if (!f.on[k]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) out.val[pencil_pos<W>(L, r, f.pre[k], f.cnt[k], f.p0[k], T0, T10)] = old[k][r] + acc[ta][tb][r];
  }
}

// the window slides by one layer: tile (ta,tb) <- tile (ta+1,tb+1); the new last row / column start at zero
template <bool SYM, int NB>
__device__ __forceinline__ void pencil_shift(d4_t (&acc)[4][4], double &Facc, int fslot) {
#pragma unroll
  for (int ta = 0; ta < NB - 1; ++ta)
#pragma unroll
    for (int tb = SYM ? ta : 0; tb < NB - 1; ++tb) acc[ta][tb] = acc[ta + 1][tb + 1];
#pragma unroll
  for (int t = 0; t < NB; ++t) { acc[t][NB - 1] = (d4_t){0, 0, 0, 0}; if (!SYM) acc[NB - 1][t] = (d4_t){0, 0, 0, 0}; }
  // F lanes hold (fx, fy, slot = lane>>4): slot t takes over slot t+1's partial sum
  const double up = __shfl_down(Facc, 16);
  Facc = (fslot >= NB - 1) ? 0.0 : up;
}



// Dirichlet data seen by one pencil of the axis-0 walk (all wave-uniform).  A node is fixed by position only:
// first / last basis function of the first / last element of a non-periodic axis with boundary values
// (IGAElementBuildFix, src/petigaelem.c:1214-1283); later faces override earlier ones (axis 0, 1, 2; side 0, 1).
struct PencilBC {
  bool any;                 // this pencil / rank touches a face with boundary values
  bool xlo, xhi, ylo, yhi;  // the pencil's element is the first / last one on axis 1 (X) / axis 2 (Y)
  int wlo, whi;             // fixed node layers on axis 0 (ghost-local), or -1000
  double vwlo, vwhi, vxlo, vxhi, vylo, vyhi;
};
// IGASetFixTable (src/petigaform.c:273-298): the value of a fixed dof is read from a row-indexed table instead of the face's
// constant.  FIXT variants of the kernel carry what it takes to find the row of node (ix, iy, layer) of the pencil.
template <bool FIXT> struct PencilFix {};
template <> struct PencilFix<true> {
  const double *table;       // [rows]
  const int *rmx, *rmy;      // rowmap of axes X, Y at the pencil's first basis function
  long long sx, sy;          // row strides of axes X, Y
  const int *rho; int lay0, nl;   // row index of the node layers the segment holds (LDS)
};
template <int P, bool FIXT = false>
__device__ __forceinline__ bool pencil_fixed(const PencilBC &b, int ix, int iy, int lay, double &val, const PencilFix<FIXT> &fx = PencilFix<FIXT>()) {
  bool f = false;
  if (lay == b.wlo) { f = true; val = b.vwlo; }
  if (lay == b.whi) { f = true; val = b.vwhi; }
  if (b.xlo && ix == 0) { f = true; val = b.vxlo; }
  if (b.xhi && ix == P) { f = true; val = b.vxhi; }
  if (b.ylo && iy == 0) { f = true; val = b.vylo; }
  if (b.yhi && iy == P) { f = true; val = b.vyhi; }
  if constexpr (FIXT) {
    if (f) {
      const int li = lay - fx.lay0;
      // (a layer outside the segment's window holds no node of this rank: its entries are zero, any finite value does)
      val = (li >= 0 && li < fx.nl && ix <= P && iy <= P) ? fx.table[(long long)fx.rho[li] + fx.sx * fx.rmx[ix] + fx.sy * fx.rmy[iy]] : 0.0;
    }
  }
  return f;
}

// The Residual of a fused pass (pencil_rwin_add) leaves with its node layer: lanes 0..8 = (a_y, a_x) add the layer's nine sums to F.  A
// Dirichlet row holds nelem (u - v) instead -- IGAElementFixFunction sets F_e[k] = u - v in each of the nelem elements of the pencil
// that hold the node (src/petigaelem.c:1441-1462) -- with u the vector's own entry, not the value the state was evaluated with.
// The add is a memory-side atomic without a return: a colour's pencils share no row, so every row takes ONE add per launch and the
// result does not depend on an order -- and nothing waits for it (a load / add / store chain of its own in every flush cost 35 % of a
// launch; split around the band row's leave it held six registers across the walk's tightest spot).  rwin: [27 sums][5 pad][9 rows],
// the rows being this pencil's F rows without the walk-axis part (64-bit, written once per pencil).
constexpr int RWIN_DOUBLES = 48;
template <int P>
__device__ __forceinline__ void pencil_rwin_leave(double *rwin, int lane, const PencilLds &T, int nl, const OutDev &out, const double *fixtable, int lay, int own_lo, int own_hi,
                                                  const PencilBC &bc, int nelem, long long rsw) {
  const int li = lay - T.lay0;
  const bool exists = li >= 0 && li < nl && T.cnt[li] > 0;
  const bool owned = exists && lay >= own_lo && lay < own_hi;
  if (lane >= 9) return;
  const int slot = (((li % 3) + 3) % 3) * 9 + lane;
  double r = rwin[slot];
  rwin[slot] = 0.0;
  if (!owned) return;
  const long long row = (long long)T.rho[li] * rsw + reinterpret_cast<const long long *>(rwin + 32)[lane];
  double fv = 0;
  if (bc.any && pencil_fixed<P, false>(bc, lane % 3, lane / 3, lay, fv)) r = (double)nelem * (out.U[row] - (fixtable ? fixtable[row] : fv));
  (void)__hip_atomic_fetch_add(out.vec + row, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Walk along axis 0: leaving layer `lay`, band-row variant (degree P, NB = P+1 basis functions per axis,
// band width BW = 2P+1).  The CSR keeps the BW axis-0 neighbours of a row contiguous, so the entries
// (row lay ; cols lay-P..lay+P) of one (a, r, b1, b2) are 8*BW contiguous bytes and NB neighbouring lanes cover
// NB of those runs back to back (224 bytes at P=3).  The upper half (cols lay..lay+P) is in the accumulator
// tiles (0,0..P).  The lower half (cols lay-P..lay-1) is the transpose of the upper halves of the rows that
// left 1..P steps ago (K_e is symmetric, only tiles ta <= tb are computed): each leaving row parks its tiles
// (0,1..P) transposed in a per-wavefront LDS area, [P(P+1)/2 slots][4 r][HOLD_LD lanes], until the partner
// row leaves (distance d uses d slots, keyed by the column layer mod d).
// NONSYM (state_pencil: a Tangent is not symmetric): all (P+1)^2 tiles are computed; the lower half is parked as it is -- tile (dd,0),
// rows of layer lay+dd x columns of layer lay, is final when layer lay leaves and belongs to the same lane that will write the row.
// BCMAT: the Dirichlet fix-up of the band row without a right-hand side (IGAElementFixJacobian, src/petigaelem.c:1425-1447).
// RB: rows (accumulator registers r) per read-add-write batch of an interior band row: all P+1 at once (one memory round trip), or 2
// where the registers of the old values are what the kernel spills (System driver on a mapped geometry at p = 3)
// LDSWIN (p = 2 packed, see pencil_mfma_p2k): `hold` is the wavefront's window of band rows; the leaving layer's row is read from it
// (and its slots zeroed for the layer that takes the ring slot next), nothing is parked and the accumulators are not a window.
template <bool SYSTEM, int P, bool FIXT = false, bool NONSYM = false, bool BCMAT = SYSTEM, int RB = P + 1, int LDSWIN = 0>      // LDSWIN: 0 none, 1 the window, 2 the compact window
__device__ __forceinline__ void pencil0_leave(d4_t (&acc)[4][4], double &Facc, double *hold, int lane,
                                              const PencilLane &L, const PencilLds &T, int nl, const OutDev &out,
                                              int lay, int own_lo, int own_hi, long long T0, long long T10, const PencilBC &bc, int nelem,
                                              const PencilFix<FIXT> &fxt = PencilFix<FIXT>()) {
  constexpr int NB = P + 1, BW = 2 * P + 1;
  const int li = lay - T.lay0;
  const bool exists = li >= 0 && li < nl && T.cnt[li] > 0;
  const bool owned = __builtin_amdgcn_readfirstlane((int)(exists && lay >= own_lo && lay < own_hi)) != 0;
  int p0[BW]; bool full = owned; long long ps0 = 0; int c0 = 0;
  if (owned) {
#pragma unroll
    for (int d = 0; d < BW; ++d) { p0[d] = __builtin_amdgcn_readfirstlane(T.P[li * 8 + d]); full = full && (p0[d] == d); }
    const long long pv = T.pre[li];
    ps0 = ((long long)__builtin_amdgcn_readfirstlane((int)(pv >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)(pv & 0xffffffffll));
    c0 = __builtin_amdgcn_readfirstlane(T.cnt[li]);
  }
  const int a = lane >> 4, b1 = lane & 3, b2 = (lane >> 2) & 3;
  const bool lane_ok = a < NB && b1 < NB && b2 < NB;      // lanes of the zero padding (P < 3) hold no entry
  // lower half: distance dd was parked by layer lay-dd; read before this step re-uses the slots
  const int ls = lane ^ (lane >> 4);   // swizzled lane slot: keeps the transposed writes below off a single bank
  double v[NB][BW];
  if constexpr (LDSWIN == 2) {
    static_assert(LDSWIN != 2 || P == 2, "the LDS window: p = 2");
    const int lp = li < 0 ? 0 : li;      // (the walk leaves layers li >= 0 only)
#pragma unroll
    for (int d = 0; d < BW; ++d) {
      const int wb = 81 * pencil_winc_block(lp, d) + 9 * a + b2 * 3 + b1;      // + 27 r
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        v[r][d] = lane_ok ? hold[wb + 27 * r] : 0.0;
        if (lane_ok) hold[wb + 27 * r] = 0.0;
      }
    }
  } else if constexpr (LDSWIN == 1) {
    static_assert(LDSWIN != 1 || P == 2, "the LDS window: p = 2");
    const int wb = (((li % 3) + 3) % 3) * WIN_LAYER + a * WIN_ROW + b2 * 3 + b1;      // + (3 r) * WIN_ROW + 9 d
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int d = 0; d < BW; ++d) {
        v[r][d] = lane_ok ? hold[wb + 3 * r * WIN_ROW + 9 * d] : 0.0;
        if (lane_ok) hold[wb + 3 * r * WIN_ROW + 9 * d] = 0.0;
      }
  } else {
#pragma unroll
  for (int dd = 1; dd <= P; ++dd) {
    const int slot = dd * (dd - 1) / 2 + (((lay - dd) % dd) + dd) % dd;
#pragma unroll
    for (int r = 0; r < NB; ++r) v[r][P - dd] = hold[(slot * 4 + r) * HOLD_LD + ls];
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int k = 0; k <= P; ++k) v[r][P + k] = acc[0][k][r];
  }
  if (owned) {
    double Fold = 0; long long frow = 0; const bool fdo = SYSTEM && L.fslot == 0 && (lane & 3) < NB && ((lane >> 2) & 3) < NB;
    if (fdo) { frow = L.frowxy + T.rho[li]; Fold = out.vec[frow]; }
    double Fnew = Facc;
    // IGAElementFixSystem (src/petigaelem.c:1377-1387) on the combined band row: F_i -= sum_k K_ik v_k over fixed
    // columns (linear, so it commutes with the sum over elements), fixed rows / columns become 0, the diagonal of
    // a fixed row becomes the number of elements of this pencil that hold the node (each sets K_kk = 1, F_k = v)
    const bool bcrow = BCMAT && bc.any && (bc.xlo || bc.xhi || bc.ylo || bc.yhi || (lay >= bc.wlo - P && lay <= bc.wlo + P) || (lay >= bc.whi - P && lay <= bc.whi + P));
    if (bcrow) {
      double corr[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        double rv = 0; const bool rf = pencil_fixed<P, FIXT>(bc, a, r, lay, rv, fxt);
        double c = 0;
#pragma unroll
        for (int d = 0; d < BW; ++d) {
          double cv = 0; const bool cf = pencil_fixed<P, FIXT>(bc, b1, b2, lay + d - P, cv, fxt);
          if (cf) c += v[r][d] * cv;
          if (rf || cf) v[r][d] = (d == P && b1 == a && b2 == r && rf) ? (double)nelem : 0.0;
        }
        c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4); c += __shfl_xor(c, 8);
        corr[r] = c;
      }
      // F lanes are (fx = lane&3, fy = (lane>>2)&3) of slot 0: the row (a = fx, r = fy) lives in lane group fx
      const int fx = lane & 3, fy = (lane >> 2) & 3;
      double t = 0;
#pragma unroll
      for (int r = 0; r < NB; ++r) { const double cr = __shfl(corr[r], fx * 16); if (r == fy) t = cr; }
      double fv = 0; const bool ff = pencil_fixed<P, FIXT>(bc, fx, fy, lay, fv, fxt);
      Fnew = ff ? (double)nelem * fv : Facc - t;
    }
    if (full) {   // interior row: one BW-entry run per (lane, r)
#pragma unroll
      for (int r0 = 0; r0 < NB; r0 += RB) {
        double o[RB][BW];
        if (lane_ok) {
#pragma unroll
          for (int rr = 0; rr < RB; ++rr) {
            const int r = r0 + rr;
            if (r >= NB) continue;
            if ((L.stmask >> r) & 1u) {   // first touch: nothing to read
#pragma unroll
              for (int k = 0; k < BW; ++k) o[rr][k] = 0.0;
              continue;
            }
            const double *p = out.val + pencil_pos<0>(L, r, ps0, c0, 0, T0, T10);
#pragma unroll
            for (int k = 0; k < BW / 2; ++k) { const d2u_t x = *reinterpret_cast<const d2u_t *>(p + 2 * k); o[rr][2 * k] = x[0]; o[rr][2 * k + 1] = x[1]; }
            o[rr][BW - 1] = p[BW - 1];
          }
        }
        // F behind the first batch's loads: its store needs Fold, and a wait for that load ahead of the band row's loads would put
        // one more memory round trip into every flush (round 3 had it there: +1.7 % shader cycles per launch of the headline kernel)
        if (r0 == 0 && fdo) out.vec[frow] = Fold + Fnew;
        if (lane_ok) {
#pragma unroll
          for (int rr = 0; rr < RB; ++rr) {
            const int r = r0 + rr;
            if (r >= NB) continue;
            double *p = out.val + pencil_pos<0>(L, r, ps0, c0, 0, T0, T10);
#pragma unroll
            for (int k = 0; k < BW / 2; ++k) { d2u_t x; x[0] = o[rr][2 * k] + v[r][2 * k]; x[1] = o[rr][2 * k + 1] + v[r][2 * k + 1]; *reinterpret_cast<d2u_t *>(p + 2 * k) = x; }
            p[BW - 1] = o[rr][BW - 1] + v[r][BW - 1];
          }
        }
      }
    } else {      // rows next to the mesh ends: some columns do not exist
      if (fdo) out.vec[frow] = Fold + Fnew;
      if (lane_ok) {
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
          for (int d = 0; d < BW; ++d) if (p0[d] >= 0) {
            double *q = out.val + pencil_pos<0>(L, r, ps0, c0, p0[d], T0, T10);
            *q = ((L.stmask >> r) & 1u) ? v[r][d] : *q + v[r][d];
          }
      }
    }
  }
  // park the transposes of tiles (0,1..P): entry (row lay ; a, r') x (col lay+dd ; b1, b2) of this lane is entry
  // (row lay+dd ; b1, b2) x (col lay ; a, r') of the consumer lane (a_c = b1, b1_c = a, b2_c = r') register r_c = b2
  if constexpr (LDSWIN != 0) {      // (only F slides: slot t takes over slot t + 1's partial sum)
    const double up = __shfl_down(Facc, 16);
    Facc = (L.fslot >= NB - 1) ? 0.0 : up;
    return;
  } else
  if constexpr (NONSYM) {
#pragma unroll
    for (int dd = 1; dd <= P; ++dd) {
      const int slot = dd * (dd - 1) / 2 + ((lay % dd) + dd) % dd;
#pragma unroll
      for (int r = 0; r < NB; ++r) hold[(slot * 4 + r) * HOLD_LD + ls] = acc[dd][0][r];
    }
  } else if (b2 < NB) {
#pragma unroll
    for (int dd = 1; dd <= P; ++dd) {
      const int slot = dd * (dd - 1) / 2 + ((lay % dd) + dd) % dd;
#pragma unroll
      for (int rp = 0; rp < NB; ++rp) {
        const int lc = (b1 * 16 + rp * 4 + a) ^ b1;     // consumer lane, swizzled like its read (lc>>4 == b1)
        hold[(slot * 4 + b2) * HOLD_LD + lc] = acc[0][dd][rp];
      }
    }
  }
  pencil_shift<!NONSYM, NB>(acc, Facc, L.fslot);
}

// leaving layer `lay` (always tile slot 0): add its 7 tiles and its F entries to the global arrays.
template <bool SYSTEM, int W>
__device__ __forceinline__ void pencil_leave(d4_t (&acc)[4][4], double &Facc, const PencilLane &L, const PencilLds &T, int nl, const OutDev &out,
                                             int lay, int own_lo, int own_hi, long long T0, long long T10, long long fstride) {
  const FlushPlan f = pencil_plan(T, nl, lay, own_lo, own_hi);
  long long frow = 0; bool fdo = false;
  if (SYSTEM && L.fslot == 0) {
    const int li = lay - T.lay0;
    fdo = li >= 0 && li < nl && T.cnt[li] > 0 && lay >= own_lo && lay < own_hi;
    if (fdo) frow = L.frowxy + fstride * T.rho[li];
  }
  pencil_flush<SYSTEM, W>(acc, L, f, out, T0, T10, fdo, frow, Facc);
  pencil_shift<false, 4>(acc, Facc, L.fslot);
}

// ---- mapped geometry (K3 Rationalize, K4 GeometryMap, K5 InverseMap folded into the contraction).
// K_e[a][b] = sum_q JW_q grad_x R_a . grad_x R_b = sum_q sum_{beta,gamma} d_beta R_a (JW F^-1 F^-T)_{beta gamma} d_gamma R_b with PARAMETRIC
// derivatives of R, so the A operand keeps its tensor-product form (three 1-D rows, as on the identity geometry) and the B
// operand is the same three products combined with the 6 metric coefficients of the lane's Gauss point.  For a NURBS
// R_a = w_a N_a / W:  d_beta R_a = (w_a / W) (d_beta N_a - N_a W_beta / W): one scale per basis function and four per point.
// The metric is evaluated per element by the wavefront itself, lane = Gauss point, on homogeneous coordinates
// (x = A / W, dx/du = (dA - x dW) / W: src/petigarat.f90.in + petigamapgeo.f90.in), closed-form inverse (src/petigainv.f90.in).

// stage the element's 4 x 4 x 4 control points: lane = (aw, ay, ax) = (lane>>4, (lane>>2)&3, lane&3)
template <int P>
__device__ __forceinline__ void pencil_geo_ctrl(double *geo, const SpaceDev &S, int lane, int off0, int offx, int offy, double (&wt)[4]) {
  constexpr int NB = P + 1;
  const int aw = lane >> 4, ay = (lane >> 2) & 3, ax = lane & 3;
  double c[4] = {0, 0, 0, 0};
  if (aw < NB && ay < NB && ax < NB) {
    const size_t g = (size_t)(off0 + aw) + (size_t)S.ax[0].gwidth * ((size_t)(offx + ax) + (size_t)S.ax[1].gwidth * (size_t)(offy + ay));
    const double w = S.W ? S.W[g] : 1.0;
    c[0] = S.X[g * 3 + 0] * w; c[1] = S.X[g * 3 + 1] * w; c[2] = S.X[g * 3 + 2] * w; c[3] = w;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the previous element's readers of this area are done
#pragma unroll
  for (int k = 0; k < 4; ++k) geo[lane * 4 + k] = c[k];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // NURBS weights of the MFMA lane's basis functions (t, ix = lane&3, iy = (lane>>2)&3)
#pragma unroll
  for (int t = 0; t < 4; ++t) wt[t] = geo[((t * 4 + ((lane >> 2) & 3)) * 4 + (lane & 3)) * 4 + 3];
}

// lane = Gauss point (qx, qy, qw) = (lane&3, (lane>>2)&3, lane>>4); uxr [q][a][2], vyr [a][q][2], ztg [q][a][2] (walk axis) raw
// rows in LDS; wj = w_q J of the three axes at this lane's point
template <int P>
__device__ __forceinline__ void pencil_geo_eval(double *geo, int lane, const double *uxr, const double *vyr, const double *ztg,
                                                double wj, double forcing, bool rational, int *errflag) {
  constexpr int NB = P + 1;
  const int qx = lane & 3, qy = (lane >> 2) & 3, qw = lane >> 4;
  const bool valid = qx < NB && qy < NB && qw < NB;
  // H[c][k] = sum_a C_a[c] D_k N_a(q), c = (wX, wY, wZ, w), k = (value, d/du0, d/du1, d/du2), by sum factorisation shared across
  // the wavefront (one component at a time; scratch behind the control points): lanes (qx, ay, aw) contract axis X, lanes
  // (qx, qy, aw) axis Y, lanes (qx, qy, qw) the walk axis: 36 fused multiply-adds per lane and component instead of 64 x 22.
  // (fp64 VALU work shares the pipe with the partner wavefront's MFMAs: the straightforward loop took 36k cycles per element.)
  double H[4][4];
  double *T1 = geo + 256, *T2 = geo + 384;
  const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
  double zv[4] = {0, 0, 0, 0}, zd[4] = {0, 0, 0, 0};
  if (i2 < NB) {
#pragma unroll
    for (int aw = 0; aw < NB; ++aw) { zv[aw] = ztg[(i2 * 4 + aw) * 2 + 0]; zd[aw] = ztg[(i2 * 4 + aw) * 2 + 1]; }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    {   // axis X: lane (qx = i0, ay = i1, aw = i2)
      double tv = 0, td = 0;
#pragma unroll
      for (int ax = 0; ax < 4; ++ax) { const double C = geo[((i2 * 4 + i1) * 4 + ax) * 4 + c]; tv += C * uxr[(i0 * 4 + ax) * 2 + 0]; td += C * uxr[(i0 * 4 + ax) * 2 + 1]; }
      T1[((0 * 4 + i1) * 4 + i2) * 4 + i0] = tv; T1[((1 * 4 + i1) * 4 + i2) * 4 + i0] = td;
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {   // axis Y: lane (qx = i0, qy = i1, aw = i2); m = 0 (value, value), 1 (d/dX, value), 2 (value, d/dY)
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
    {   // walk axis: lane = point (qx = i0, qy = i1, qw = i2)
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
  double M[GEO_M] = {0, 0, 0, 0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
  if (valid) {
    const double iw = 1.0 / H[3][0];
    double F[3][3];     // F[c][beta] = dx_c / du_beta
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double x = H[c][0] * iw;
#pragma unroll
      for (int b = 0; b < 3; ++b) F[c][b] = (H[c][1 + b] - x * H[3][1 + b]) * iw;
    }
    const double det = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) + F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
    if (!(det > 0.0)) atomicExch(errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
    const double id = 1.0 / det;
    double E[3][3];     // E[beta][c] = du_beta / dx_c
    E[0][0] = (F[1][1] * F[2][2] - F[1][2] * F[2][1]) * id; E[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * id; E[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * id;
    E[1][0] = (F[1][2] * F[2][0] - F[1][0] * F[2][2]) * id; E[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * id; E[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * id;
    E[2][0] = (F[1][0] * F[2][1] - F[1][1] * F[2][0]) * id; E[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * id; E[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * id;
    const double JW = det * wj;
    int k = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int g = b; g < 3; ++g) M[k++] = JW * (E[b][0] * E[g][0] + E[b][1] * E[g][1] + E[b][2] * E[g][2]);
    M[6] = forcing * JW * (rational ? iw : 1.0);
    R[0] = iw; R[1] = H[3][1] * iw; R[2] = H[3][2] * iw; R[3] = H[3][3] * iw;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // every lane has read the control points
#pragma unroll
  for (int k = 0; k < GEO_M; ++k) geo[lane * GEO_M + k] = M[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) geo[64 * GEO_M + lane * 4 + k] = R[k];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---- run-time scalar forms on this kernel (include/petiga.h:153-197: the point callback is an arbitrary user function; here a
// struct compiled at run time, rtc.hpp).  A dof-1, first-order form whose matrix integrand reads the gradients only and is
// symmetric, k_q[a][b] = sum_ij d_i N_a C_ij(q) d_j N_b with C = mat(p, e_i, e_j) -- Poisson (demo/Poisson3D.c:3-23) is C = I; an
// anisotropic or x-dependent diffusion is the same shape -- has the metric JW F^-1 C F^-T in the place of JW F^-1 F^-T above, and
// its load term vec(p, e_N) in the place of the constant forcing: the rest of the walk (operands, symmetric tiles, band-row flush,
// first touch, Dirichlet fix-up) does not know the difference.  IDENT: no geometry, x = the parametric point, F = I.
template <class Form> struct is_builtin_gram { static constexpr bool v = false; };
template <> struct is_builtin_gram<void> { static constexpr bool v = true; };

template <int P, class Form, bool IDENT>
__device__ __forceinline__ void pencil_form_eval(double *geo, int lane, const double *uxr, const double *vyr, const double *ztg,
                                                 double wj, bool rational, int *errflag, const double *prm, double shift, double tt, const double (&xpar)[3]) {
  constexpr int NB = P + 1;
  const int qx = lane & 3, qy = (lane >> 2) & 3, qw = lane >> 4;
  const bool valid = qx < NB && qy < NB && qw < NB;
  double H[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) H[c][k] = 0.0;
  if constexpr (!IDENT) {      // the sums over the control points, as in pencil_geo_eval
    double *T1 = geo + 256, *T2 = geo + 384;
    const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
    double zv[4] = {0, 0, 0, 0}, zd[4] = {0, 0, 0, 0};
    if (i2 < NB) {
#pragma unroll
      for (int aw = 0; aw < NB; ++aw) { zv[aw] = ztg[(i2 * 4 + aw) * 2 + 0]; zd[aw] = ztg[(i2 * 4 + aw) * 2 + 1]; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      {
        double tv = 0, td = 0;
#pragma unroll
        for (int ax = 0; ax < 4; ++ax) { const double C = geo[((i2 * 4 + i1) * 4 + ax) * 4 + c]; tv += C * uxr[(i0 * 4 + ax) * 2 + 0]; td += C * uxr[(i0 * 4 + ax) * 2 + 1]; }
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
  }
  double M[GEO_M] = {0, 0, 0, 0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
  if (valid) {
    double E[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, x[3] = {xpar[0], xpar[1], xpar[2]}, det = 1.0, iw = 1.0;
    if constexpr (!IDENT) {
      iw = 1.0 / H[3][0];
      double F[3][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x[c] = H[c][0] * iw;
#pragma unroll
        for (int b = 0; b < 3; ++b) F[c][b] = (H[c][1 + b] - x[c] * H[3][1 + b]) * iw;
      }
      det = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) + F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
      if (!(det > 0.0)) atomicExch(errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
      const double id = 1.0 / det;
      E[0][0] = (F[1][1] * F[2][2] - F[1][2] * F[2][1]) * id; E[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * id; E[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * id;
      E[1][0] = (F[1][2] * F[2][0] - F[1][0] * F[2][2]) * id; E[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * id; E[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * id;
      E[2][0] = (F[1][0] * F[2][1] - F[1][1] * F[2][0]) * id; E[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * id; E[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * id;
      R[0] = iw; R[1] = H[3][1] * iw; R[2] = H[3][2] * iw; R[3] = H[3][3] * iw;
    }
    // the user's integrand at this point: C_ij = mat(p, e_i, e_j), load = vec(p, e_N)
    PtView p; p.x = x; p.u = nullptr; p.ut = nullptr; p.gu = nullptr; p.hu = nullptr; p.G = nullptr; p.prm = prm; p.shift = shift; p.t = tt;
    p.normal = nullptr; p.atboundary = 0; p.boundary_id = -1;
    double C[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) {
        double ei[4] = {0, 0, 0, 0}, ej[4] = {0, 0, 0, 0}, T[1];
        ei[1 + i] = 1.0; ej[1 + j] = 1.0;
        Form::mat(p, ei, ej, T);
        C[i][j] = T[0]; C[j][i] = T[0];     // (declared symmetric: MAT_SYMMETRIC)
      }
    double e0[4] = {1, 0, 0, 0}, Fv[1];
    Form::vec(p, e0, Fv);
    const double JW = det * wj;
    int k = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double EC[3];      // (E C)[b][j]
#pragma unroll
      for (int j = 0; j < 3; ++j) EC[j] = E[b][0] * C[0][j] + E[b][1] * C[1][j] + E[b][2] * C[2][j];
#pragma unroll
      for (int g = b; g < 3; ++g) M[k++] = JW * (EC[0] * E[g][0] + EC[1] * E[g][1] + EC[2] * E[g][2]);
    }
    M[6] = Fv[0] * JW * (rational ? iw : 1.0);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int k = 0; k < GEO_M; ++k) geo[lane * GEO_M + k] = M[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) geo[64 * GEO_M + lane * 4 + k] = R[k];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// the MFMAs of one element on a mapped geometry: k-step (qw, qy, beta), k slot = qx (lane>>4); 10 tiles (K_e is symmetric)
// (the walk-axis point loop stays rolled and the rational branch is a template parameter: fully unrolled with both branches
// the kernel was 60 KB of code and its MFMA phase took 55k cycles against 31k of MFMA issue: instruction-cache bound)
// Round 4: what does not depend on the tile slot t is computed once per point.  With the parametric gradient of the lane's basis
// function g_gamma(t) = w_t / W (z'_t P_gamma + z_t Q_gamma), P = (a_n, 0, 0), Q = (-a_n o_0, a_x - a_n o_1, a_y - a_n o_2) (o = dW / W;
// a polynomial map: Q = (0, a_x, a_y)), the trial operand B_beta(t) = sum_gamma M_beta,gamma g_gamma(t) is the SAME two-term form
// with PB_beta = M_beta0 P_0 and QB_beta = sum_gamma M_beta,gamma Q_gamma: 20 point-level + 10 per-slot multiply-adds per (qw, qy) instead
// of 3 + 20 per slot (NURBS: 83 -> 60 per 30 MFMAs; the fp64 VALU work shares the pipe with them).
// Round 4 (late): pipelined by hand.  Left to the compiler every point's loads and its 46 operand instructions stood in front of its
// first MFMA (~550 cycles without an MFMA in flight per point: 8.8k of the phase's 43.7k cycles at p = 3).  Now the point-level half
// of the NEXT point (P, Q, PB, QB: 10 doubles) is prepared in the last third of a point's MFMAs, and each third opens with only
// its own 8-12 operand instructions; __builtin_amdgcn_sched_barrier(0) keeps the thirds apart.
template <int NB, bool RAT>
__device__ __forceinline__ void pencil_mfma_geo(d4_t (&acc)[4][4], double u0, double u1, const double *vy, const double *ztg,
                                                const double *geo, int lane, const double (&wt)[4]) {
  const int qx = lane >> 4;
  static_assert(NB % 2 == 0, "two walk-axis points per trip of the rolled loop");
  struct Pt { double P0, Q0, Q1, Q2, PB[3], QB[3]; };
  auto prep = [&](int qw, int qy, Pt &s) {
    const double vy0 = vy[qy * 2 + 0], vy1 = vy[qy * 2 + 1];
    const int p = (qw * 4 + qy) * 4 + qx;
    const double *Mp = geo + p * GEO_M, *Rp = geo + 64 * GEO_M + p * 4;
    const double m00 = Mp[0], m01 = Mp[1], m02 = Mp[2], m11 = Mp[3], m12 = Mp[4], m22 = Mp[5];
    double P0 = u0 * vy0, Q0 = 0.0, Q1 = u1 * vy0, Q2 = u0 * vy1;
    if (RAT) {
      const double rinv = Rp[0];
      Q0 = -P0 * Rp[1]; Q1 -= P0 * Rp[2]; Q2 -= P0 * Rp[3];
      P0 *= rinv; Q0 *= rinv; Q1 *= rinv; Q2 *= rinv;
    }
    s.P0 = P0; s.Q0 = Q0; s.Q1 = Q1; s.Q2 = Q2;
    s.PB[0] = m00 * P0; s.PB[1] = m01 * P0; s.PB[2] = m02 * P0;
    s.QB[0] = m01 * Q1 + m02 * Q2; s.QB[1] = m11 * Q1 + m12 * Q2; s.QB[2] = m12 * Q1 + m22 * Q2;
    if (RAT) { s.QB[0] += m00 * Q0; s.QB[1] += m01 * Q0; s.QB[2] += m02 * Q0; }
  };
  Pt cur;
  prep(0, 0, cur);
#pragma unroll 1
  for (int qw2 = 0; qw2 < NB; qw2 += 2) {
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
    const int qw = qw2 + qh;
    double z0[NB], z1[NB];      // the walk-axis row of tile slot t, with the NURBS weight of its control point
#pragma unroll
    for (int t = 0; t < NB; ++t) { z0[t] = ztg[(qw * 4 + t) * 2 + 0]; z1[t] = ztg[(qw * 4 + t) * 2 + 1]; if (RAT) { z0[t] *= wt[t]; z1[t] *= wt[t]; } }
#pragma unroll
    for (int qy = 0; qy < NB; ++qy) {
      Pt nxt;
#pragma unroll
      for (int be = 0; be < 3; ++be) {
        double g[NB], B[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          if (be == 0) { g[t] = z1[t] * cur.P0; if (RAT) g[t] += z0[t] * cur.Q0; }
          else g[t] = z0[t] * (be == 1 ? cur.Q1 : cur.Q2);
          B[t] = z1[t] * cur.PB[be] + z0[t] * cur.QB[be];
        }
        if (be == 2) prep(qy + 1 < NB ? qw : min(qw + 1, NB - 1), qy + 1 < NB ? qy + 1 : 0, nxt);      // (the last one is not used)
#pragma unroll
        for (int ta = 0; ta < NB; ++ta)
#pragma unroll
          for (int tb = ta; tb < NB; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[ta], B[tb], acc[ta][tb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      cur = nxt;
    }
    }
  }
}

// p = 2 on a mapped geometry: 27 points in 7 k-steps (see pencil_mfma_p2): 126 MFMAs per element instead of 162.  Pipelined by
// hand like pencil_mfma_geo: the next step's point-level operands and walk-axis rows are prepared under the second third of a step's
// MFMAs (the compiler's order had a step's ~45 operand instructions and their loads in front of its first MFMA: 16.9k cycles per
// element for 8.1k of MFMA issue).
template <bool RAT>
__device__ __forceinline__ void pencil_mfma_geo_p2(d4_t (&acc)[4][4], const double *uxr, const double *vy, const double *ztg,
                                                   const double *geo, int lane, const double (&wt)[4]) {
  const int ks = lane >> 4, ix = lane & 3;
  struct Pt { double P0, Q0, Q1, Q2, PB[3], QB[3], z0[3], z1[3]; };
  struct Raw { double u0, u1, vy0, vy1, m[6], r[4], z0[3], z1[3]; };
  auto loads = [&](int j, Raw &w) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = rem - 3 * qy;
    w.u0 = on ? uxr[(qx * 4 + ix) * 2 + 0] : 0.0; w.u1 = on ? uxr[(qx * 4 + ix) * 2 + 1] : 0.0;
    w.vy0 = vy[qy * 2 + 0]; w.vy1 = vy[qy * 2 + 1];
    const int p = (qw * 4 + qy) * 4 + qx;
    const double *Mp = geo + p * GEO_M, *Rp = geo + 64 * GEO_M + p * 4;
#pragma unroll
    for (int k = 0; k < 6; ++k) w.m[k] = Mp[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) w.r[k] = RAT ? Rp[k] : 0.0;
#pragma unroll
    for (int t = 0; t < 3; ++t) { w.z0[t] = ztg[(qw * 4 + t) * 2 + 0]; w.z1[t] = ztg[(qw * 4 + t) * 2 + 1]; }
  };
  auto point = [&](const Raw &w, Pt &s) {      // (see pencil_mfma_geo: the point-level half of both operands)
    double P0 = w.u0 * w.vy0, Q0 = 0.0, Q1 = w.u1 * w.vy0, Q2 = w.u0 * w.vy1;
    if (RAT) {
      Q0 = -P0 * w.r[1]; Q1 -= P0 * w.r[2]; Q2 -= P0 * w.r[3];
      P0 *= w.r[0]; Q0 *= w.r[0]; Q1 *= w.r[0]; Q2 *= w.r[0];
    }
    s.P0 = P0; s.Q0 = Q0; s.Q1 = Q1; s.Q2 = Q2;
  };
  auto trial = [&](const Raw &w, Pt &s) {
    const double m00 = w.m[0], m01 = w.m[1], m02 = w.m[2], m11 = w.m[3], m12 = w.m[4], m22 = w.m[5];
    s.PB[0] = m00 * s.P0; s.PB[1] = m01 * s.P0; s.PB[2] = m02 * s.P0;
    s.QB[0] = m01 * s.Q1 + m02 * s.Q2; s.QB[1] = m11 * s.Q1 + m12 * s.Q2; s.QB[2] = m12 * s.Q1 + m22 * s.Q2;
    if (RAT) { s.QB[0] += m00 * s.Q0; s.QB[1] += m01 * s.Q0; s.QB[2] += m02 * s.Q0; }
  };
  auto rows = [&](const Raw &w, Pt &s) {
#pragma unroll
    for (int t = 0; t < 3; ++t) { s.z0[t] = RAT ? w.z0[t] * wt[t] : w.z0[t]; s.z1[t] = RAT ? w.z1[t] * wt[t] : w.z1[t]; }
  };
  Pt cur;
  { Raw w; loads(0, w); point(w, cur); trial(w, cur); rows(w, cur); }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    Pt nxt = cur; Raw w;
#pragma unroll
    for (int be = 0; be < 3; ++be) {
      double g[3], B[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (be == 0) { g[t] = cur.z1[t] * cur.P0; if (RAT) g[t] += cur.z0[t] * cur.Q0; }
        else g[t] = cur.z0[t] * (be == 1 ? cur.Q1 : cur.Q2);
        B[t] = cur.z1[t] * cur.PB[be] + cur.z0[t] * cur.QB[be];
      }
      // six MFMAs in three pairs, a slice of the next step's preparation behind each of the first pairs
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[0], B[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[0], B[1], acc[0][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j < 6) { if (be == 0) loads(j + 1, w); else if (be == 1) point(w, nxt); else trial(w, nxt); }
      acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[0], B[2], acc[0][2], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[1], B[1], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j < 6 && be == 2) rows(w, nxt);
      acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[1], B[2], acc[1][2], 0, 0, 0);
      acc[2][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[2], B[2], acc[2][2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = nxt;
  }
}

// F_a += w_a sum_q (forcing JW / W)_q N_a(q) for the F lane (fx, fy, slot): the sum factorises over the axes
template <int NB>
__device__ __forceinline__ double pencil_f_geo(const double *geo, int lane, const double *uxr, const double *vyr, const double *ztg) {
  const int fx = lane & 3, fy = (lane >> 2) & 3, aw = lane >> 4;
  if (fx >= NB || fy >= NB || aw >= NB) return 0.0;
  double s = 0;
#pragma unroll 1
  for (int qw = 0; qw < NB; ++qw) {      // (rolled: unrolled, the 64 point values stay live and spill; one qw slice -- 16 LDS reads in flight -- fits)
    double sy = 0;
#pragma unroll
    for (int qy = 0; qy < NB; ++qy) {
      double sx = 0;
#pragma unroll
      for (int qx = 0; qx < NB; ++qx) sx += uxr[(qx * 4 + fx) * 2] * geo[((qw * 4 + qy) * 4 + qx) * GEO_M + 6];
      sy += vyr[(fy * 4 + qy) * 2] * sx;
    }
    s += ztg[(qw * 4 + aw) * 2] * sy;
  }
  return s;
}

// ---- Tangents of nonlinear scalar forms on this walk (state_pencil below; SURVEY 8a rows 15, 17: IGAPointFormValue/Grad/Hess
// fused with the contraction).  A form opts in with PENCIL_NFEAT / PENCIL_NC / pencil_coef / pencil_trial (forms.hpp): its Tangent
// at a point is sum_f A_f(a) B_f(b) with A = (N, d_w N, d_x N, d_y N[, lap N]) -- tensor products of the three 1-D rows, built in
// registers like the Gram operands -- and B = pencil_trial(c_q, features of b), c_q = pencil_coef(state at the point, JW).
// No geometry (x = the parametric point: physical derivatives are the rows' own).  Per element the wavefront gathers the (P+1)^3
// coefficients of U, sums u, grad u and the diagonal of hess u at its (P+1)^3 points by sum factorisation across the lanes
// (three LDS exchanges) and leaves PENCIL_NC numbers per point where the metric of a mapped geometry would be.
// (pencil_state_of<Form>: forms.hpp)
constexpr int STATE_D2 = 48;      // per-wavefront second derivatives of the 1-D rows: X [q][a], Y [a][q], walk axis [q][a] (zero padded 4 x 4)
__host__ __device__ static inline size_t pencil_state_bytes() { return (size_t)8 * STATE_D2 * 8; }

// lane = (aw, ay, ax) holds ucoef, the coefficient of U at its node (the Dirichlet value there: IGAElementFixValues,
// src/petigaelem.c:1334-1358); on return geo[point * NC + k] holds the point coefficients (zeros on the padding)
// RESID (the fused IFunction + IJacobian pass): pencil_coef_r leaves the Residual's point numbers behind the Tangent's, u_t apart
// (pencil_resid_ut below).
template <int P, class Form, bool RESID = false>
__device__ __forceinline__ void pencil_state_eval(double *geo, const double *d2w, int lane, const double *uxr, const double *vyr, const double *ztg,
                                                  double wj, double ucoef, const double *prm, double shift, double tt, const double (&xpar)[3]) {
  constexpr int NB = P + 1, NC = Form::PENCIL_NC + (RESID ? pencil_resid_of<Form>::ncr : 0);
  static_assert(NC * 64 <= (RESID ? GEO_Z : 64 + 192 + 320), "point coefficients fit the scratch they replace");
  double *C0 = geo, *T1 = geo + 64, *T2 = geo + 256;
  const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the previous element's readers are done
  C0[lane] = ucoef;
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  {   // axis X: lane (qx = i0, ay = i1, aw = i2)
    double tv = 0, td = 0, t2 = 0;
#pragma unroll
    for (int ax = 0; ax < NB; ++ax) { const double C = C0[(i2 * 4 + i1) * 4 + ax]; tv += C * uxr[(i0 * 4 + ax) * 2 + 0]; td += C * uxr[(i0 * 4 + ax) * 2 + 1]; t2 += C * d2w[i0 * 4 + ax]; }
    T1[((0 * 4 + i1) * 4 + i2) * 4 + i0] = tv; T1[((1 * 4 + i1) * 4 + i2) * 4 + i0] = td; T1[((2 * 4 + i1) * 4 + i2) * 4 + i0] = t2;
  }
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  {   // axis Y: lane (qx = i0, qy = i1, aw = i2): (value, value), (d/dX, value), (value, d/dY), (d2/dX2, value), (value, d2/dY2)
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
#pragma unroll
    for (int ay = 0; ay < NB; ++ay) {
      const double a = T1[((0 * 4 + ay) * 4 + i2) * 4 + i0], d = T1[((1 * 4 + ay) * 4 + i2) * 4 + i0], e = T1[((2 * 4 + ay) * 4 + i2) * 4 + i0];
      const double yv = vyr[(ay * 4 + i1) * 2 + 0], yd = vyr[(ay * 4 + i1) * 2 + 1], y2 = d2w[16 + ay * 4 + i1];
      m0 += a * yv; m1 += d * yv; m2 += a * yd; m3 += e * yv; m4 += a * y2;
    }
    T2[((0 * 4 + i2) * 4 + i1) * 4 + i0] = m0; T2[((1 * 4 + i2) * 4 + i1) * 4 + i0] = m1; T2[((2 * 4 + i2) * 4 + i1) * 4 + i0] = m2;
    T2[((3 * 4 + i2) * 4 + i1) * 4 + i0] = m3; T2[((4 * 4 + i2) * 4 + i1) * 4 + i0] = m4;
  }
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double c[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) c[k] = 0.0;
  {   // walk axis: lane = point (qx = i0, qy = i1, qw = i2)
    double u = 0, gu[3] = {0, 0, 0}, hu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int aw = 0; aw < NB; ++aw) {
      const double t0 = T2[((0 * 4 + aw) * 4 + i1) * 4 + i0], t1 = T2[((1 * 4 + aw) * 4 + i1) * 4 + i0], t2 = T2[((2 * 4 + aw) * 4 + i1) * 4 + i0];
      const double t3 = T2[((3 * 4 + aw) * 4 + i1) * 4 + i0], t4 = T2[((4 * 4 + aw) * 4 + i1) * 4 + i0];
      const double zv = ztg[(i2 * 4 + aw) * 2 + 0], zd = ztg[(i2 * 4 + aw) * 2 + 1], z2 = d2w[32 + i2 * 4 + aw];
      u += t0 * zv; gu[0] += t0 * zd; gu[1] += t1 * zv; gu[2] += t2 * zv; hu[0] += t0 * z2; hu[4] += t3 * zv; hu[8] += t4 * zv;
    }
    if (i0 < NB && i1 < NB && i2 < NB) {
      PtView p; p.x = xpar; p.u = &u; p.ut = nullptr; p.gu = gu; p.hu = hu; p.G = nullptr; p.prm = prm; p.shift = shift; p.t = tt;
      p.normal = nullptr; p.atboundary = 0; p.boundary_id = -1;
      if constexpr (RESID) Form::pencil_coef_r(p, wj, c); else Form::pencil_coef(p, wj, c);
    }
  }
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // every lane has read the partial sums
#pragma unroll
  for (int k = 0; k < NC; ++k) geo[lane * NC + k] = c[k];
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// The Residual's u_t (fused pass): V's value at the element's points by the same three stages, at the HEAD of the wave's own MFMA phase
// -- in the flush phase every fp64 instruction waits for a gap in the partner's MFMA stream (about one MFMA each: the 30 of this
// evaluation cost 3k cycles there).  scr: 128 doubles of the wavefront's own, the first 64 holding the lanes' coefficients of V at
// their nodes (aw, ay, ax), gathered in the flush phase before (0 at a Dirichlet node: IGAElementDelValues,
// src/petigaelem.c:1327-1341); the slot PENCIL_NC of the point's numbers (JW from pencil_coef_r) becomes JW u_t.
template <int P, class Form>
__device__ __forceinline__ void pencil_resid_ut(double *geo, double *scr, int lane, const double *uxr, const double *vyr, const double *ztg) {
  constexpr int NB = P + 1, NC0 = Form::PENCIL_NC, NC = NC0 + pencil_resid_of<Form>::ncr;
  double *C1 = scr, *T1v = scr + 64, *T2v = scr;      // (C1: the lanes' coefficients of V, left there by the flush phase before; T2v takes its place once stage X has read it)
  const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double sv = 0;
#pragma unroll
  for (int ax = 0; ax < NB; ++ax) sv += C1[(i2 * 4 + i1) * 4 + ax] * uxr[(i0 * 4 + ax) * 2 + 0];
  T1v[(i1 * 4 + i2) * 4 + i0] = sv;
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double mv = 0;
#pragma unroll
  for (int ay = 0; ay < NB; ++ay) mv += T1v[(ay * 4 + i2) * 4 + i0] * vyr[(ay * 4 + i1) * 2 + 0];
  T2v[(i2 * 4 + i1) * 4 + i0] = mv;
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double ut = 0;
#pragma unroll
  for (int aw = 0; aw < NB; ++aw) ut += T2v[(aw * 4 + i1) * 4 + i0] * ztg[(i2 * 4 + aw) * 2 + 0];
  geo[lane * NC + NC0] *= ut;      // (padding lanes hold zeros)
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// the MFMAs of one element of a Tangent: k-step (qw, qy, feature), k slot = qx (lane>>4); all (P+1)^2 tiles
template <int NB, class Form>
__device__ __forceinline__ void pencil_mfma_state(d4_t (&acc)[4][4], double u0, double u1, double u2, const double *vy, const double *vy2,
                                                  const double *ztg, const double *zt2, const double *coef, int lane) {
  constexpr int NF = Form::PENCIL_NFEAT, NC = Form::PENCIL_NC;
  constexpr bool LAP = NF > 4;
  const int qx = lane >> 4;
#pragma unroll 1
  for (int qw = 0; qw < NB; ++qw) {
    double z0[NB], z1[NB], z2[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { z0[t] = ztg[(qw * 4 + t) * 2 + 0]; z1[t] = ztg[(qw * 4 + t) * 2 + 1]; z2[t] = LAP ? zt2[qw * 4 + t] : 0.0; }
#pragma unroll
    for (int qy = 0; qy < NB; ++qy) {
      const double vy0 = vy[qy * 2 + 0], vy1 = vy[qy * 2 + 1];
      const double *cp = coef + ((qw * 4 + qy) * 4 + qx) * NC;
      double c[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) c[k] = cp[k];
      const double a_n = u0 * vy0, a_x = u1 * vy0, a_y = u0 * vy1, a_l = LAP ? u2 * vy0 + u0 * vy2[qy] : 0.0;
      double A[NF][NB], B[NB][NF];
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        A[0][t] = a_n * z0[t]; A[1][t] = a_n * z1[t]; A[2][t] = a_x * z0[t]; A[3][t] = a_y * z0[t];
        double lap = 0.0;
        if constexpr (LAP) { lap = a_n * z2[t] + a_l * z0[t]; A[NF - 1][t] = lap; }
        const double g[3] = {A[1][t], A[2][t], A[3][t]};
        Form::pencil_trial(c, A[0][t], g, lap, B[t]);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int ta = 0; ta < NB; ++ta)
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f][ta], B[tb][f], acc[ta][tb], 0, 0, 0);
    }
  }
}

// p = 2: the 27 points of an element fill 7 k-steps of 4 (the loop above leaves the fourth k slot of each of its 9 steps empty:
// 405 MFMAs); point = 4 * step + (lane>>4), so every row is this lane's own LDS read
template <class Form>
__device__ __forceinline__ void pencil_mfma_state_p2(d4_t (&acc)[4][4], const double *uxr, const double *vyr, const double *ztg, const double *d2w,
                                                     const double *coef, int lane) {
  constexpr int NB = 3, NF = Form::PENCIL_NFEAT, NC = Form::PENCIL_NC;
  constexpr bool LAP = NF > 4;
  const int ks = lane >> 4, ix = lane & 3, iy = (lane >> 2) & 3;
  // Round 4 (late): the order is set by hand.  A step's MFMAs go out in threes; the next step's loads follow late (their registers are the dying operands'), its test
  // operands A the threes of the second-to-last feature, its trial operands B (pencil_trial, one tile slot at a time) those of the last
  // one -- where this step's other operands are dead.  The compiler's own order left ~25 operand instructions in front of several steps.
  struct Raw { double u0, u1, u2, v0, v1, v2, c[NC], z0[NB], z1[NB], z2[NB]; };
  auto loads = [&](int j, Raw &w) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = rem - 3 * qy;
    w.u0 = on ? uxr[(qx * 4 + ix) * 2 + 0] : 0.0; w.u1 = on ? uxr[(qx * 4 + ix) * 2 + 1] : 0.0; w.u2 = (on && LAP) ? d2w[qx * 4 + ix] : 0.0;
    w.v0 = vyr[(iy * 4 + qy) * 2 + 0]; w.v1 = vyr[(iy * 4 + qy) * 2 + 1]; w.v2 = LAP ? d2w[16 + iy * 4 + qy] : 0.0;
#pragma unroll
    for (int t = 0; t < NB; ++t) { w.z0[t] = ztg[(qw * 4 + t) * 2 + 0]; w.z1[t] = ztg[(qw * 4 + t) * 2 + 1]; w.z2[t] = LAP ? d2w[32 + qw * 4 + t] : 0.0; }
  };
  auto load_coef = [&](int j, Raw &w) {
    const int pt = 4 * j + ks;
    const int pc = pt < 27 ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = rem - 3 * qy;
    const double *cp = coef + ((qw * 4 + qy) * 4 + qx) * NC;
#pragma unroll
    for (int k = 0; k < NC; ++k) w.c[k] = cp[k];
  };
  auto test_ops = [&](const Raw &w, double (&A)[NF][NB]) {
    const double a_n = w.u0 * w.v0, a_x = w.u1 * w.v0, a_y = w.u0 * w.v1, a_l = LAP ? w.u2 * w.v0 + w.u0 * w.v2 : 0.0;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      A[0][t] = a_n * w.z0[t]; A[1][t] = a_n * w.z1[t]; A[2][t] = a_x * w.z0[t]; A[3][t] = a_y * w.z0[t];
      if constexpr (LAP) A[NF - 1][t] = a_n * w.z2[t] + a_l * w.z0[t];
    }
  };
  auto trial_ops = [&](const Raw &w, const double (&A)[NF][NB], int t, double (&B)[NB][NF]) {
    const double g[3] = {A[1][t], A[2][t], A[3][t]};
    Form::pencil_trial(w.c, A[0][t], g, LAP ? A[NF - 1][t] : 0.0, B[t]);
  };
  double A[NF][NB], B[NB][NF];
  { Raw w; loads(0, w); load_coef(0, w); test_ops(w, A);
#pragma unroll
    for (int t = 0; t < NB; ++t) trial_ops(w, A, t, B); }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double A2[NF][NB], B2[NB][NF]; Raw w;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int ta = 0; ta < NB; ++ta) {
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f][ta], B[tb][f], acc[ta][tb], 0, 0, 0);
        if (j < 6) {
          if (f == NF - 3 && ta == 1) loads(j + 1, w);
          if (f == NF - 2 && ta == 0) test_ops(w, A2);
          if (f == NF - 2 && ta == 1) load_coef(j + 1, w);
          if (f == NF - 1) trial_ops(w, A2, ta, B2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    if (j < 6) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < NB; ++t) { A[f][t] = A2[f][t]; B[t][f] = B2[t][f]; }
    }
  }
}

// p = 2 PACKED (see pencil_mfma_p2k): the 27 functions in two tiles of 16 rows, all four tiles of the (non-symmetric) Tangent: 4 MFMAs
// per feature and k-step instead of 9 -- CahnHilliard 140 per element instead of 315 -- and two operand columns per lane instead of
// three.  The sum over the pencil's elements moves into the LDS window (pencil_win_add_ns).
// RESID: the Residual on the same MFMAs.  Tile 1 has five padding columns (functions 27..31); column 27 -- the operand of the lanes
// with (lane & 15) == 11 -- carries the Residual's point numbers r_f(q) in the place of a trial function's B_f(b, q), so that
// K[a][27] = sum_q sum_f A_f(a, q) r_f(q) = R_a comes out of tiles (0, 1) and (1, 1) with the Tangent, for nothing on the pipe.
template <class Form, bool RESID = false, class KL = void>
__device__ __forceinline__ void pencil_mfma_state_p2k(d4_t (&pk)[4], const double *uxr, const double *vyr, const double *ztg, const double *d2w,
                                                      const double *coef, const KL &K, int lane) {
  constexpr int NF = Form::PENCIL_NFEAT, NC = Form::PENCIL_NC + (RESID ? pencil_resid_of<Form>::ncr : 0);
  const bool rcol = RESID && (lane & 15) == 11;
  constexpr bool LAP = NF > 4;
  const int ks = lane >> 4;
  pk[0] = pk[1] = pk[2] = pk[3] = (d4_t){0, 0, 0, 0};
  // The order is set by hand, as in pencil_mfma_state_p2: a step's MFMAs go out in pairs (one operand column against both trial
  // columns); the next step's rows are read behind the pairs of the third-to-last feature, its test operands built behind the first
  // pair of the second-to-last one, its coefficients read behind the second, its trial operands (pencil_trial, one column at a time)
  // behind the pairs of the last feature -- where this step's other operands are dead.
  struct Raw { d2u_t u[2], v[2], z[2]; double u2[2], v2[2], z2[2], c[NC]; };
  auto loads = [&](int j, Raw &w) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = on ? rem - 3 * qy : 3;      // (q_x = 3: the zero-padded point of the X rows)
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      w.u[T] = *reinterpret_cast<const d2u_t *>(uxr + qx * 8 + K.ua[T]);
      w.v[T] = *reinterpret_cast<const d2u_t *>(vyr + K.va[T] + qy * 2);
      w.z[T] = *reinterpret_cast<const d2u_t *>(ztg + qw * 8 + K.za[T]);
      w.u2[T] = LAP ? d2w[qx * 4 + (K.ua[T] >> 1)] : 0.0; w.v2[T] = LAP ? d2w[16 + (K.va[T] >> 1) + qy] : 0.0; w.z2[T] = LAP ? d2w[32 + qw * 4 + (K.za[T] >> 1)] : 0.0;
    }
  };
  auto load_coef = [&](int j, Raw &w) {
    const int pt = 4 * j + ks;
    const int pc = pt < 27 ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = rem - 3 * qy;
    const double *cp = coef + ((qw * 4 + qy) * 4 + qx) * NC;
#pragma unroll
    for (int k = 0; k < NC; ++k) w.c[k] = cp[k];
  };
  auto test_ops = [&](const Raw &w, double (&A)[NF][2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      const double a_n = w.u[T][0] * w.v[T][0], a_x = w.u[T][1] * w.v[T][0], a_y = w.u[T][0] * w.v[T][1];
      A[0][T] = a_n * w.z[T][0]; A[1][T] = a_n * w.z[T][1]; A[2][T] = a_x * w.z[T][0]; A[3][T] = a_y * w.z[T][0];
      if constexpr (LAP) A[NF - 1][T] = a_n * w.z2[T] + (w.u2[T] * w.v[T][0] + w.u[T][0] * w.v2[T]) * w.z[T][0];
    }
  };
  auto trial_ops = [&](const Raw &w, const double (&A)[NF][2], int T, double (&B)[2][NF]) {
    const double g[3] = {A[1][T], A[2][T], A[3][T]};
    Form::pencil_trial(w.c, A[0][T], g, LAP ? A[NF - 1][T] : 0.0, B[T]);
    if constexpr (RESID) {
      if (T == 1) {      // (the 28th point of the last step is padding: its TEST operands are zero, whatever r holds)
        double r[NF]; Form::pencil_resid(w.c, r);
#pragma unroll
        for (int f = 0; f < NF; ++f) B[1][f] = rcol ? r[f] : B[1][f];
      }
    }
  };
  double A[NF][2], B[2][NF];
  { Raw w; loads(0, w); load_coef(0, w); test_ops(w, A); trial_ops(w, A, 0, B); trial_ops(w, A, 1, B); }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double A2[NF][2], B2[2][NF]; Raw w;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int Ta = 0; Ta < 2; ++Ta) {
        pk[Ta * 2 + 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f][Ta], B[0][f], pk[Ta * 2 + 0], 0, 0, 0);
        pk[Ta * 2 + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f][Ta], B[1][f], pk[Ta * 2 + 1], 0, 0, 0);
        if (j < 6) {
          if (f == NF - 3) { if (Ta == 0) loads(j + 1, w); }
          if (f == NF - 2 && Ta == 0) test_ops(w, A2);
          if (f == NF - 2 && Ta == 1) load_coef(j + 1, w);
          if (f == NF - 1) trial_ops(w, A2, Ta, B2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    if (j < 6) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int T = 0; T < 2; ++T) { A[f][T] = A2[f][T]; B[T][f] = B2[T][f]; }
    }
  }
}

// ---- Tangents on a mapped geometry (p = 2; state_pencil_geo below).  The physical features of a basis function R_a = w_a N_a / W at
// a Gauss point are LINEAR in the parametric derivatives of the polynomial N_a up to the form's order:
//   R = n / W,   d_i R = sum_b E_bi (n_b - n o_b) / W                    (E = du/dx, o = dW / W: K3 Rationalize + K5 InverseMap)
//   lap R = [ sum_bg G_bg n_bg + sum_b (b_b - 2 (G o)_b) n_b + (2 o.G o - b.o - G : d2W / W) n ] / W
// with G = E E^T and b_b = -sum_c E_bc (G : d2 x_c) the Laplacian of the parametric coordinate u_b (the second-order term of the
// inverse map: src/petigageo.f90.in GeometryMap order 2, petigabsp/petigarat second derivatives); the factor w_a is the lane's.  So the
// wavefront leaves, per Gauss point, the form's coefficients c (pencil_coef of the PHYSICAL u, grad u, hess u there) and this map
// (1 + 9 + 3 [+ 6 + 3 + 1] numbers); the MFMA phase folds the map into the x-y part of the lane's tensor product once per point
// (23 multiply-adds) and finishes each tile slot with the walk-axis row (10), then pencil_trial as on the identity geometry.
// The sums over the (p+1)^3 control points -- x, y, z, W and w U, ten derivatives each -- are the sum factorisation of
// pencil_state_eval with the mixed second derivatives added.  Records are packed [27 points][SGEO_NPD] (stride 34 doubles: the four
// k slots of an MFMA step read four different records, 16-byte aligned and on different banks).
constexpr int SGEO_NPD = 34, SGEO_Z = 1216, SGEO_DOUBLES = SGEO_Z + 32 + 32;      // records / scratch | walk-axis rows [32] | NURBS weights of the element's nodes [27]
__host__ __device__ static inline size_t pencil_sgeo_bytes() { return (size_t)8 * SGEO_DOUBLES * 8; }

// stage the element's control points in homogeneous form with the state: lane = (aw, ay, ax) -> [node][5] = (wX, wY, wZ, w, wU),
// node = (aw (P+1) + ay)(P+1) + ax (packed: the scratch of the sums and the parked second derivatives share the area, see below)
template <int P>
__device__ __forceinline__ void pencil_sgeo_ctrl(double *geo, const SpaceDev &S, int lane, int off0, int offx, int offy, double ucoef) {
  constexpr int NB = P + 1;
  const int aw = lane >> 4, ay = (lane >> 2) & 3, ax = lane & 3;
  const bool valid = aw < NB && ay < NB && ax < NB;
  double c[5] = {0, 0, 0, 0, 0};
  if (valid) {
    const size_t g = (size_t)(off0 + aw) + (size_t)S.ax[0].gwidth * ((size_t)(offx + ax) + (size_t)S.ax[1].gwidth * (size_t)(offy + ay));
    const double w = S.W ? S.W[g] : 1.0;
    c[0] = S.X[g * 3 + 0] * w; c[1] = S.X[g * 3 + 1] * w; c[2] = S.X[g * 3 + 2] * w; c[3] = w; c[4] = ucoef * w;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the previous element's readers of this area are done
  if (valid) {
#pragma unroll
    for (int k = 0; k < 5; ++k) geo[((aw * NB + ay) * NB + ax) * 5 + k] = c[k];
    geo[SGEO_Z + 32 + (aw * NB + ay) * NB + ax] = c[3];      // (the MFMA phase reads the weight of its lane's basis functions from here)
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// The sums (flush phase of the previous element: LDS exchanges, 57 multiply-adds per component).  lane = Gauss point (qx, qy, qw) =
// (lane&3, (lane>>2)&3, lane>>4); rows as in pencil_state_eval.  Scratch, packed by the (P+1)^3 valid lanes: control points [node][5] |
// T1 [3][ay][aw][qx] | T2 [6][aw][qy][qx] | the raw second derivatives parked per point [5][6] (in registers the 50 sums of a point
// next to the walk's accumulators spilled 219 VGPRs).  Hx[component][value, d_w, d_x, d_y] stays in registers.
template <int P, bool LAP> struct SgeoLayout {
  static constexpr int NB = P + 1, N3 = NB * NB * NB;
  static constexpr int OT1 = (N3 * 5 + 1) & ~1, OT2 = (OT1 + 3 * N3 + 1) & ~1, OPK = (OT2 + 6 * N3 + 1) & ~1;
  static_assert(N3 * SGEO_NPD <= SGEO_Z && OPK + 30 * N3 <= SGEO_Z, "the point records and the scratch fit the area");
};
template <int P, bool RAT, class Form>
__device__ __forceinline__ void pencil_sgeo_sums(double *geo, const double *d2w_, int lane, const double *uxr_, const double *vyr_, const double *ztg_, double (&Hx)[5][4]) {
  constexpr int NB = P + 1;
  constexpr bool LAP = Form::PENCIL_NFEAT > 4;
  constexpr int ND = LAP ? 10 : 4;
  using LY = SgeoLayout<P, LAP>;
  double *C0 = geo, *T1 = geo + LY::OT1, *T2 = geo + LY::OT2, *PK = geo + LY::OPK;
  const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
  const bool valid = i0 < NB && i1 < NB && i2 < NB;
  const int j0 = min(i0, NB - 1), j1 = min(i1, NB - 1), j2 = min(i2, NB - 1);      // (padding lanes read in bounds, write nothing)
  const int pt = (j2 * NB + j1) * NB + j0;
  // derivative slots: value, w, x, y, ww, wx, wy, xx, xy, yy (w = the walk axis = parametric axis 0); the six second derivatives of the
  // five components are parked in LDS [point][component][6]
#pragma unroll
  for (int k = 0; k < 4; ++k) Hx[3][k] = k == 0 ? 1.0 : 0.0;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    if (!RAT && c == 3) continue;
    // (the 1-D rows are re-read per component: hoisted out of this loop they hold 54 VGPRs next to the walk's accumulators)
    int keep = 0; asm volatile("" : "+v"(keep));
    const double *uxr = uxr_ + keep, *vyr = vyr_ + keep, *ztg = ztg_ + keep, *d2w = d2w_ + keep;
    {   // axis X: lane (qx = i0, ay = i1, aw = i2)
      double tv = 0, td = 0, t2 = 0;
#pragma unroll
      for (int ax = 0; ax < NB; ++ax) {
        const double C = C0[((j2 * NB + j1) * NB + ax) * 5 + c];
        tv += C * uxr[(i0 * 4 + ax) * 2 + 0]; td += C * uxr[(i0 * 4 + ax) * 2 + 1];
        if (LAP) t2 += C * d2w[i0 * 4 + ax];
      }
      if (valid) {
        T1[((0 * NB + j1) * NB + j2) * NB + j0] = tv; T1[((1 * NB + j1) * NB + j2) * NB + j0] = td;
        if (LAP) T1[((2 * NB + j1) * NB + j2) * NB + j0] = t2;
      }
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {   // axis Y: lane (qx = i0, qy = i1, aw = i2): (v,v), (d,v), (v,d), (d2,v), (v,d2), (d,d)
      double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
#pragma unroll
      for (int ay = 0; ay < NB; ++ay) {
        const double a = T1[((0 * NB + ay) * NB + j2) * NB + j0], d = T1[((1 * NB + ay) * NB + j2) * NB + j0];
        const double yv = vyr[(ay * 4 + i1) * 2 + 0], yd = vyr[(ay * 4 + i1) * 2 + 1];
        m0 += a * yv; m1 += d * yv; m2 += a * yd;
        if (LAP) { const double e = T1[((2 * NB + ay) * NB + j2) * NB + j0], y2 = d2w[16 + ay * 4 + i1]; m3 += e * yv; m4 += a * y2; m5 += d * yd; }
      }
      if (valid) {
        T2[((0 * NB + j2) * NB + j1) * NB + j0] = m0; T2[((1 * NB + j2) * NB + j1) * NB + j0] = m1; T2[((2 * NB + j2) * NB + j1) * NB + j0] = m2;
        if (LAP) { T2[((3 * NB + j2) * NB + j1) * NB + j0] = m3; T2[((4 * NB + j2) * NB + j1) * NB + j0] = m4; T2[((5 * NB + j2) * NB + j1) * NB + j0] = m5; }
      }
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {   // walk axis: lane = point (qx = i0, qy = i1, qw = i2)
      double h[ND];
#pragma unroll
      for (int k = 0; k < ND; ++k) h[k] = 0.0;
#pragma unroll
      for (int aw = 0; aw < NB; ++aw) {
        const double zv = ztg[(i2 * 4 + aw) * 2 + 0], zd = ztg[(i2 * 4 + aw) * 2 + 1];
        const double t0 = T2[((0 * NB + aw) * NB + j1) * NB + j0], t1 = T2[((1 * NB + aw) * NB + j1) * NB + j0], t2 = T2[((2 * NB + aw) * NB + j1) * NB + j0];
        h[0] += t0 * zv; h[1] += t0 * zd; h[2] += t1 * zv; h[3] += t2 * zv;
        if constexpr (LAP) {
          const double z2 = d2w[32 + i2 * 4 + aw];
          const double t3 = T2[((3 * NB + aw) * NB + j1) * NB + j0], t4 = T2[((4 * NB + aw) * NB + j1) * NB + j0], t5 = T2[((5 * NB + aw) * NB + j1) * NB + j0];
          h[4] += t0 * z2; h[5] += t1 * zd; h[6] += t2 * zd; h[7] += t3 * zv; h[8] += t5 * zv; h[9] += t4 * zv;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) Hx[c][k] = h[k];
      if constexpr (LAP) {
        if (valid) {
#pragma unroll
          for (int k = 0; k < 6; ++k) PK[pt * 30 + c * 6 + k] = h[4 + k];      // (read back by this lane only)
        }
      }
    }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// The point's record from its sums (at the start of the element's own MFMA phase: 450 fp64 operations per lane, which next to the
// partner wavefront's MFMA stream -- in the flush phase -- wait for the shared FP64 pipe, ~25 cycles each).  On return
// geo[point * SGEO_NPD + k], point = (qw (P+1) + qy)(P+1) + qx, holds c[0..NC) | 1/W | E_bi / W [b][i] | -(E^T o)_i / W | (LAP:) m_k G_k / W
// (ww, wx, wy, xx, xy, yy; m = 2 off the diagonal) | (b - 2 G o)_b / W | the coefficient of n.  The scratch of the sums is dead but for
// the parked values, which their own lane reads before it stores the part of its record that follows from them (program order
// within the wavefront).
template <int P, bool RAT, class Form>
__device__ __forceinline__ void pencil_sgeo_point(double *geo, int lane, const double (&Hx)[5][4], double wj, const double *prm, double shift, double tt, int *errflag) {
  constexpr int NB = P + 1, NC = Form::PENCIL_NC;
  constexpr bool LAP = Form::PENCIL_NFEAT > 4;
  static_assert(NC + 13 + (LAP ? 10 : 0) <= SGEO_NPD, "the point record fits its stride");
  const double *PK = geo + SgeoLayout<P, LAP>::OPK;
  const int i0 = lane & 3, i1 = (lane >> 2) & 3, i2 = lane >> 4;
  const bool valid = i0 < NB && i1 < NB && i2 < NB;
  const int pt = (min(i2, NB - 1) * NB + min(i1, NB - 1)) * NB + min(i0, NB - 1);
  if (valid) {
    constexpr int PB[6] = {0, 0, 0, 1, 1, 2}, PG[6] = {0, 1, 2, 1, 2, 2};
    double *rec = geo + pt * SGEO_NPD, *Lp = rec + NC;
    const double (&Hw)[4] = Hx[3], (&Hu)[4] = Hx[4];
    const double iw = 1.0 / Hw[0];
    double o[3], Wh[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 3; ++b) o[b] = Hw[1 + b] * iw;
    if constexpr (LAP) {
#pragma unroll
      for (int k = 0; k < 6; ++k) Wh[k] = RAT ? PK[pt * 30 + 3 * 6 + k] * iw : 0.0;
    }
    // the quotient rule on the homogeneous sums: x_c (c = 0..2) and u
    double xv[3], F[3][3], u, du[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xv[c] = Hx[c][0] * iw;
#pragma unroll
      for (int b = 0; b < 3; ++b) F[c][b] = Hx[c][1 + b] * iw - xv[c] * o[b];      // F[c][b] = dx_c / du_b
    }
    u = Hu[0] * iw;
#pragma unroll
    for (int b = 0; b < 3; ++b) du[b] = Hu[1 + b] * iw - u * o[b];
    const double det = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) + F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
    if (!(det > 0.0)) atomicExch(errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
    const double id = 1.0 / det;
    double E[3][3];     // E[b][c] = du_b / dx_c
    E[0][0] = (F[1][1] * F[2][2] - F[1][2] * F[2][1]) * id; E[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * id; E[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * id;
    E[1][0] = (F[1][2] * F[2][0] - F[1][0] * F[2][2]) * id; E[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * id; E[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * id;
    E[2][0] = (F[1][0] * F[2][1] - F[1][1] * F[2][0]) * id; E[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * id; E[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * id;
    // the state in physical derivatives
    double gu[3], hu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) gu[i] = E[0][i] * du[0] + E[1][i] * du[1] + E[2][i] * du[2];
    double G[6], Go[3] = {0, 0, 0}, bb[3] = {0, 0, 0}, gw = 0;
    if constexpr (LAP) {
      double v[6];      // G = E E^T (m_k G_k below); v_k = u_,k - sum_c (d_c u) x_c,k: hess u = E^T v E
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        G[k] = (PB[k] == PG[k] ? 1.0 : 2.0) * (E[PB[k]][0] * E[PG[k]][0] + E[PB[k]][1] * E[PG[k]][1] + E[PB[k]][2] * E[PG[k]][2]);
        v[k] = PK[pt * 30 + 4 * 6 + k] * iw - du[PB[k]] * o[PG[k]] - du[PG[k]] * o[PB[k]] - u * Wh[k];
        gw += G[k] * Wh[k];
      }
      double q[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double xk = PK[pt * 30 + c * 6 + k] * iw - F[c][PB[k]] * o[PG[k]] - F[c][PG[k]] * o[PB[k]] - xv[c] * Wh[k];      // x_c,k
          s += G[k] * xk; v[k] -= gu[c] * xk;
        }
        q[c] = s;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += (PB[k] == PG[k] ? 1.0 : 2.0) * E[PB[k]][i] * E[PG[k]][i] * v[k];
        hu[i * 4] = s;
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) bb[b] = -(E[b][0] * q[0] + E[b][1] * q[1] + E[b][2] * q[2]);
      // (G o)_b with the doubled off-diagonal entries halved again
      Go[0] = G[0] * o[0] + 0.5 * (G[1] * o[1] + G[2] * o[2]); Go[1] = 0.5 * G[1] * o[0] + G[3] * o[1] + 0.5 * G[4] * o[2]; Go[2] = 0.5 * (G[2] * o[0] + G[4] * o[1]) + G[5] * o[2];
    }
    // ---- the record (the parked values have been read)
    {
      double c[NC];
      PtView p; p.x = xv; p.u = &u; p.ut = nullptr; p.gu = gu; p.hu = hu; p.G = nullptr; p.prm = prm; p.shift = shift; p.t = tt;
      p.normal = nullptr; p.atboundary = 0; p.boundary_id = -1;
      Form::pencil_coef(p, det * wj, c);
#pragma unroll
      for (int k = 0; k < NC; ++k) rec[k] = c[k];
    }
    Lp[0] = iw;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) Lp[1 + b * 3 + i] = iw * E[b][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Lp[10 + i] = -iw * (E[0][i] * o[0] + E[1][i] * o[1] + E[2][i] * o[2]);
    if constexpr (LAP) {
#pragma unroll
      for (int k = 0; k < 6; ++k) Lp[13 + k] = iw * G[k];
#pragma unroll
      for (int b = 0; b < 3; ++b) Lp[19 + b] = iw * (bb[b] - 2.0 * Go[b]);
      Lp[22] = iw * (2.0 * (o[0] * Go[0] + o[1] * Go[1] + o[2] * Go[2]) - (bb[0] * o[0] + bb[1] * o[1] + bb[2] * o[2]) - gw);
    }
  }
  __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// the MFMAs of one element of a Tangent on a mapped geometry, p = 2: 7 k-steps of 4 points as in pencil_mfma_state_p2.
// The loop stays rolled (unrolled, 84 VGPRs spill around it) and is pipelined by hand: a step's point-level operands (P, Q, R and
// the coefficients c: 19 doubles) are prepared under the MFMAs of the step before -- in program order behind the first feature's nine
// MFMAs, held there by the scheduling groups below; left to itself the compiler puts the 130 instructions of a step's preparation in
// front of its first MFMA, where the matrix pipe idles for ~1000 of the step's 3900 cycles.
template <int M, int N>
__device__ __forceinline__ void sgeo_sched_groups() {      // one MFMA, then its share of the other instructions (constant arguments only)
  if constexpr (M < N) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (M < 9) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    else if constexpr (M < 33) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
    sgeo_sched_groups<M + 1, N>();
  }
}
// ... PACKED (see pencil_mfma_state_p2k): two operand columns per lane (its functions 16 T + (lane & 15)), each with its own fold of the
// point's map, four tiles: 4 MFMAs per feature and k-step instead of 9.  The NURBS weight of function f sits at ztg[32 + f].
template <bool RAT, class Form, class KL = void>
__device__ __forceinline__ void pencil_mfma_state_geo_p2k(d4_t (&pk)[4], const double *uxr, const double *vyr, const double *ztg, const double *d2w,
                                                          const double *geo, const KL &K, int lane) {
  constexpr int NF = Form::PENCIL_NFEAT, NC = Form::PENCIL_NC;
  constexpr bool LAP = NF > 4;
  const int ks = lane >> 4;
  double wf[2] = {1.0, 1.0};
  if constexpr (RAT) {
#pragma unroll
    for (int T = 0; T < 2; ++T) { const int f = 16 * T + (lane & 15); wf[T] = ztg[32 + (f < 27 ? f : 26)]; }
  }
  pk[0] = pk[1] = pk[2] = pk[3] = (d4_t){0, 0, 0, 0};
#pragma unroll 1
  for (int j = 0; j < 7; ++j) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qp = rem - 3 * qy, qx = on ? qp : 3;      // (q_x = 3: the zero-padded point of the X rows)
    const double *rp = geo + pc * SGEO_NPD, *Lp = rp + NC;
    double c[NC], Lm[23];
#pragma unroll
    for (int k = 0; k < NC; ++k) c[k] = rp[k];
#pragma unroll
    for (int k = 0; k < 23; ++k) Lm[k] = (LAP || k < 13) ? Lp[k] : 0.0;
    double A[NF][2], B[2][NF];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      const d2u_t u = *reinterpret_cast<const d2u_t *>(uxr + qx * 8 + K.ua[T]);
      const d2u_t v = *reinterpret_cast<const d2u_t *>(vyr + K.va[T] + qy * 2);
      const d2u_t z = *reinterpret_cast<const d2u_t *>(ztg + qw * 8 + K.za[T]);
      const double z0 = z[0] * wf[T], z1 = z[1] * wf[T];
      const double a_n = u[0] * v[0], a_x = u[1] * v[0], a_y = u[0] * v[1];
      A[0][T] = z0 * (Lm[0] * a_n);
#pragma unroll
      for (int i = 0; i < 3; ++i) A[1 + i][T] = z0 * (Lm[10 + i] * a_n + Lm[4 + i] * a_x + Lm[7 + i] * a_y) + z1 * (Lm[1 + i] * a_n);
      double lap = 0.0;
      if constexpr (LAP) {
        const double u2 = d2w[qx * 4 + (K.ua[T] >> 1)], v2 = d2w[16 + (K.va[T] >> 1) + qy], z2 = d2w[32 + qw * 4 + (K.za[T] >> 1)] * wf[T];
        const double a_xx = u2 * v[0], a_xy = u[1] * v[1], a_yy = u[0] * v2;
        const double PL = Lm[22] * a_n + Lm[20] * a_x + Lm[21] * a_y + Lm[16] * a_xx + Lm[17] * a_xy + Lm[18] * a_yy;
        const double QL = Lm[19] * a_n + Lm[14] * a_x + Lm[15] * a_y, RL = Lm[13] * a_n;
        lap = z0 * PL + z1 * QL + z2 * RL;
        A[NF - 1][T] = lap;
      }
      const double g[3] = {A[1][T], A[2][T], A[3][T]};
      Form::pencil_trial(c, A[0][T], g, lap, B[T]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int Ta = 0; Ta < 2; ++Ta)
#pragma unroll
        for (int Tb = 0; Tb < 2; ++Tb) pk[Ta * 2 + Tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f][Ta], B[Tb][f], pk[Ta * 2 + Tb], 0, 0, 0);
  }
}

template <bool RAT, class Form>
__device__ __forceinline__ void pencil_mfma_state_geo_p2(d4_t (&acc)[4][4], const double *uxr, const double *vyr, const double *ztg, const double *d2w,
                                                         const double *geo, int lane) {
  constexpr int NB = 3, NF = Form::PENCIL_NFEAT, NC = Form::PENCIL_NC;
  constexpr bool LAP = NF > 4;
  const int ks = lane >> 4, ix = lane & 3, iy = (lane >> 2) & 3;
  const double *wl = ztg + 32 + min(iy, NB - 1) * NB + min(ix, NB - 1);      // NURBS weights [t][iy][ix] (a padding lane's operands are zero: any finite weight)
  struct Step { double c[NC], PN, PG[3], QG[3], PL, QL, RL; int qw; };
  auto prep = [&](int j, Step &s) {
    const int pt = 4 * j + ks;
    const bool on = pt < 27;
    const int pc = on ? pt : 0, qw = pc / 9, rem = pc - 9 * qw, qy = rem / 3, qx = rem - 3 * qy;
    const double u0 = on ? uxr[(qx * 4 + ix) * 2 + 0] : 0.0, u1 = on ? uxr[(qx * 4 + ix) * 2 + 1] : 0.0, u2 = (on && LAP) ? d2w[qx * 4 + ix] : 0.0;
    const double vy0 = vyr[(iy * 4 + qy) * 2 + 0], vy1 = vyr[(iy * 4 + qy) * 2 + 1], vy2 = LAP ? d2w[16 + iy * 4 + qy] : 0.0;
    const double *rp = geo + pc * SGEO_NPD, *Lp = rp + NC;
#pragma unroll
    for (int k = 0; k < NC; ++k) s.c[k] = rp[k];
    const double a_n = u0 * vy0, a_x = u1 * vy0, a_y = u0 * vy1;
    s.PN = Lp[0] * a_n;
#pragma unroll
    for (int i = 0; i < 3; ++i) { s.PG[i] = Lp[10 + i] * a_n + Lp[4 + i] * a_x + Lp[7 + i] * a_y; s.QG[i] = Lp[1 + i] * a_n; }
    s.PL = 0; s.QL = 0; s.RL = 0;
    if constexpr (LAP) {
      const double a_xx = u2 * vy0, a_xy = u1 * vy1, a_yy = u0 * vy2;
      s.PL = Lp[22] * a_n + Lp[20] * a_x + Lp[21] * a_y + Lp[16] * a_xx + Lp[17] * a_xy + Lp[18] * a_yy;
      s.QL = Lp[19] * a_n + Lp[14] * a_x + Lp[15] * a_y;
      s.RL = Lp[13] * a_n;
    }
    s.qw = qw;
  };
  Step cur;
  prep(0, cur);
#pragma unroll 1
  for (int j = 0; j < 7; ++j) {
    double A[NF][NB], B[NB][NF];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      double z0 = ztg[(cur.qw * 4 + t) * 2 + 0], z1 = ztg[(cur.qw * 4 + t) * 2 + 1], z2 = LAP ? d2w[32 + cur.qw * 4 + t] : 0.0;
      if (RAT) { const double w = wl[t * NB * NB]; z0 *= w; z1 *= w; z2 *= w; }
      A[0][t] = z0 * cur.PN;
#pragma unroll
      for (int i = 0; i < 3; ++i) A[1 + i][t] = z0 * cur.PG[i] + z1 * cur.QG[i];
      double lap = 0.0;
      if constexpr (LAP) { lap = z0 * cur.PL + z1 * cur.QL + z2 * cur.RL; A[NF - 1][t] = lap; }
      const double g[3] = {A[1][t], A[2][t], A[3][t]};
      Form::pencil_trial(cur.c, A[0][t], g, lap, B[t]);
    }
    Step nxt;
    prep(min(j + 1, 6), nxt);      // (the last one is not used: no branch inside the scheduling region)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int ta = 0; ta < NB; ++ta)
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f][ta], B[tb][f], acc[ta][tb], 0, 0, 0);
    cur = nxt;
    // the order of the region: the rows of this step, its operands, then the MFMAs with the next step's loads and arithmetic between them
    if constexpr (RAT) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
    sgeo_sched_groups<0, NF * NB * NB>();
  }
}

// ALIAS: the wrapped walk axis (PencilArgs::alias0) known at compile time -- 0: not wrapped, 1: wrapped, -1: read from the arguments.
// The identity-geometry Gram instantiations come in both fixed flavours, so the headline kernel carries none of the modulo logic.
template <bool SYSTEM, int W, int P, bool GEO, bool RAT, bool FIXT, class Form, bool IDENT, int ALIAS = -1, bool PACK = false, bool RESID = false>
__device__ __forceinline__ void gram_pencil_body(const SpaceDev &S, const OutDev &out, const PencilArgs &pa, const double *prm) {
  static_assert(!RESID || (PACK && IDENT && pencil_state_of<Form>::v && pencil_resid_of<Form>::v), "the fused Residual: a packed Tangent on the identity geometry of a form with the PENCIL_NCR hooks");
  static_assert(!GEO || W == 0, "the mapped-geometry variant walks axis 0");
  static_assert(!PACK || (P == 2 && W == 0 && !FIXT && ((!GEO && is_builtin_gram<Form>::v) || pencil_state_of<Form>::v)),
                "packed tiles: p = 2, the Gram matrix on the identity geometry (pencil_mfma_p2k) or a Tangent (pencil_mfma_state_p2k / pencil_mfma_state_geo_p2k)");
  constexpr int WINMODE = !PACK ? 0 : ((pencil_state_of<Form>::v && !IDENT) ? 2 : 1);      // the window of band rows: none / 15 blocks / compact (12 blocks: a Tangent on a mapped geometry is short of LDS)
  static_assert(is_builtin_gram<Form>::v || (GEO && !FIXT), "a run-time form takes the metric path");
  static_assert(!IDENT || (GEO && !RAT && !is_builtin_gram<Form>::v), "IDENT: a run-time form without a geometry");
  static_assert(!FIXT || (SYSTEM && W == 0), "fix tables: System driver, axis-0 walk");
  constexpr int NB = P + 1, BW = 2 * P + 1;
  constexpr int X = (W == 0) ? 1 : 0, Y = (W == 2) ? 1 : 2;   // the two non-walked mesh axes, X the faster one
  static_assert(P == 3 || W == 0, "degrees below 3 are only instantiated for the axis-0 walk");
  constexpr bool STATE = pencil_state_of<Form>::v;             // a Tangent: all tiles, point coefficients from the state (state_pencil)
  constexpr bool SGEO = STATE && !IDENT;                       // ... on a mapped geometry (p = 2: pencil_sgeo_sums / pencil_sgeo_point)
  static_assert(!STATE || !SYSTEM, "a Tangent on the walk: matrix only");
  static_assert(!SGEO || P == 2, "a Tangent on a mapped geometry: p = 2");
  constexpr int GZ = SGEO ? SGEO_Z : GEO_Z, GD = SGEO ? SGEO_DOUBLES : GEO_DOUBLES;      // the per-wavefront metric area and the walk-axis rows behind it
  extern __shared__ __attribute__((aligned(16))) double pencil_sm[];
  long long tw_entry = 0, tw_staged = 0, tw_loop = 0, tw_loopend = 0;      // -DIGX_DEBUG: the life of a workgroup (entry | tables staged | first element | last element | end)
  long long tw_wall = 0;                                                   // ... and its entry on the 100 MHz clock all XCCs share
  if (kDebug && pa.debug_buf) { tw_entry = __builtin_readcyclecounter(); tw_wall = wall_clock64(); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = blockIdx.x / pa.blocks_per_seg;
  const int pencil = (blockIdx.x - seg * pa.blocks_per_seg) * pa.wpb + wave;
  const int ws = pa.w_lo + seg * pa.seg_len;
  const int we = min(ws + pa.seg_len, pa.w_hi);
  static_assert(ALIAS <= 0 || W == 0, "only the axis-0 walk wraps");
  const bool alias0 = ALIAS >= 0 ? ALIAS == 1 : (W == 0 && pa.alias0 != 0);
  const int nelw = pa.w_hi - pa.w_lo;                        // (alias0: the whole axis, w_lo = 0)
  const int wh = alias0 ? ws - P : max(ws - P, pa.w_halo_lo);   // first element walked (alias0: may be negative = from the end of the axis)
  const int ne = we - wh, nl = ne + P;
  const AxisDev &AW = S.ax[W], &AX = S.ax[X], &AY = S.ax[Y];
  auto ew = [&](int ei) -> int { int e = wh + ei; if (alias0) { e %= nelw; if (e < 0) e += nelw; } return e; };   // element of walk step ei

  // ---- stage the segment's walk-axis tables in LDS (all 512 threads); tables are zero padded to 4 x 4
  PencilLds T = pencil_lds_carve(pencil_sm, pa.ne_max, GEO);
  T.lay0 = alias0 ? AW.off[0] + wh : AW.off[wh];             // (alias0: a virtual layer number; rows and tables are taken modulo the axis)
  {
    const int tid = threadIdx.x, nthr = pa.wpb * 64;
    if (!GEO) for (int i = tid; i < ne * 32; i += nthr) {   // i = e*32 + (q*4 + a)*2 + k ; rows scaled by sqrt(w_q * J_e)
      const int e = i >> 5, j = i & 31, q = j >> 3, aa = (j >> 1) & 3, k = j & 1, eg = ew(e);
      T.zt[i] = (q < NB && aa < NB) ? AW.tab[((size_t)eg * NB * NB + q * NB + aa) * NDER + k] * sqrt(AW.w[eg * NB + q] * AW.J[eg]) : 0.0;
    }
    // (GEO: the weight itself, not its root: the metric carries the whole JW)
    for (int i = tid; i < ne * 4; i += nthr) { const int e = i >> 2, q = i & 3, eg = ew(e); const double wj = (q < NB) ? AW.w[eg * NB + q] * AW.J[eg] : 0.0; T.wq[i] = GEO ? wj : sqrt(wj); }
    for (int i = tid; i < ne; i += nthr) T.Jz[i] = AW.J[ew(i)];
    for (int i = tid; i < nl; i += nthr) {
      int lay = T.lay0 + i;
      if (alias0) { lay = (lay - AW.off[0]) % nelw; if (lay < 0) lay += nelw; lay += AW.off[0]; }
      if (lay < AW.gwidth) {
        const int rho = AW.rowmap[lay];
        T.rho[i] = rho; T.cnt[i] = AW.rcnt[rho]; T.pre[i] = AW.prefix[rho];
        for (int d = 0; d < BW; ++d) T.P[i * 8 + d] = AW.P[lay * BW + d];
      } else { T.rho[i] = 0; T.cnt[i] = -1; T.pre[i] = 0; }
    }
  }
  __syncthreads();
  if (kDebug && pa.debug_buf) tw_staged = __builtin_readcyclecounter();
  // an idle wavefront (grid padding) keeps running the barrier schedule below on pencil 0 with writes disabled
  const bool valid = pencil < pa.ex_count * pa.ey_count;

  const int pc = valid ? pencil : 0;
  const int tx = pc % pa.ex_count, ty = pc / pa.ex_count;
  const int elx = pa.ex_start + tx * pa.ex_step, ely = pa.ey_start + ty * pa.ey_step;
  int own_lo = (seg == 0 && pa.w_halo_lo == pa.w_lo) ? -1 : AW.off[ws];
  int own_hi = (seg == pa.nseg - 1 && !pa.open_hi) ? (1 << 30) : AW.off[we];
  if (alias0) { own_lo = AW.off[0] + ws; own_hi = AW.off[0] + we; }
  if ((kDebug && pa.debug_noflush == 1) || !valid) { own_lo = 1 << 30; own_hi = 1 << 30; }
  const int offx = AX.off[elx], offy = AY.off[ely];
  const long long T0 = S.ax[0].tot, T10 = S.ax[1].tot * S.ax[0].tot;
  // row index = rho0 + nrow0*(rho1 + nrow1*rho2): stride of each axis' rho
  const long long rs[3] = {1, (long long)S.ax[0].nrow, (long long)S.ax[0].nrow * S.ax[1].nrow};

  PencilLane L;
  double *d2w = nullptr; double u2 = 0;      // STATE: second derivatives of the 1-D rows (LDS, per wavefront); this lane's X-row entry
  double *rwin = nullptr, *vscr = nullptr;   // RESID: the Residual's ring of node layers and the scratch of its u_t sums (LDS, per wavefront)
  {
    const double *__restrict__ TX = AX.tab + (size_t)elx * (NB * NB * NDER);
    const double *__restrict__ TY = AY.tab + (size_t)ely * (NB * NB * NDER);
    const double *__restrict__ WX = AX.w + elx * NB;
    const double *__restrict__ WYq = AY.w + ely * NB;
    const int qx = lane >> 4, ix = lane & 3, iy = (lane >> 2) & 3;
    L.u0 = 0; L.u1 = 0;
    if (qx < NB && ix < NB) { const double sx = GEO ? 1.0 : sqrt(WX[qx] * AX.J[elx]); L.u0 = TX[(qx * NB + ix) * NDER + 0] * sx; L.u1 = TX[(qx * NB + ix) * NDER + 1] * sx; }
    {   // Y-axis rows of this pencil -> LDS [a][q][2] (zero padded); a lane later reads its own row (a = iy) one q at a time
      double *vyw = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + ((pencil_lds_bytes(pa.ne_max, GEO, pa.wpb) - (size_t)2 * pa.wpb * 32 * 8))) + wave * 32;
      if (lane < 32) {
        const int aa = lane >> 3, qq = (lane >> 1) & 3, kk = lane & 1;
        vyw[lane] = (aa < NB && qq < NB) ? TY[(qq * NB + aa) * NDER + kk] * (GEO ? 1.0 : sqrt(WYq[qq] * AY.J[ely])) : 0.0;
      }
      L.vy = vyw + iy * 8;
      if (lane >= 32) {   // X-axis rows [q][a][2]: GEO unscaled, for the geometry evaluation and F; else scaled like u0 / u1 (p = 2: packed k-steps)
        const int l2 = lane - 32, qq = l2 >> 3, aa = (l2 >> 1) & 3, kk = l2 & 1;
        vyw[pa.wpb * 32 + l2] = (aa < NB && qq < NB) ? TX[(qq * NB + aa) * NDER + kk] * (GEO ? 1.0 : sqrt(WX[qq] * AX.J[elx])) : 0.0;
      }
    }
    if constexpr (STATE) {   // second derivatives of the X rows [q][a] and of the Y rows [a][q]
      d2w = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + pencil_lds_bytes(pa.ne_max, true) + (WINMODE == 2 ? pencil_winc_bytes(8) : (WINMODE == 1 ? pencil_win_bytes(8) : pencil_hold_bytes(P))) + (size_t)8 * GD * 8) + wave * STATE_D2;
      if (lane < 16) { const int qq = lane >> 2, aa = lane & 3; d2w[lane] = (qq < NB && aa < NB) ? TX[(qq * NB + aa) * NDER + 2] : 0.0; }
      else if (lane < 32) { const int aa = (lane - 16) >> 2, qq = lane & 3; d2w[lane] = (qq < NB && aa < NB) ? TY[(qq * NB + aa) * NDER + 2] : 0.0; }
      if (qx < NB && ix < NB) u2 = TX[(qx * NB + ix) * NDER + 2];
      if constexpr (RESID) {      // the Residual's ring behind the eight waves' second-derivative rows (PencilModule::extra_lds has its 2 KB)
        rwin = d2w - wave * STATE_D2 + 8 * STATE_D2 + wave * RWIN_DOUBLES;
        if (lane < 32) rwin[lane] = 0.0;
        vscr = d2w - wave * STATE_D2 + 8 * STATE_D2 + 8 * RWIN_DOUBLES + wave * 128;
      }
    }
    // scatter constants: this lane's result rows are (X: a = lane>>4, Y: r), columns (X: b1 = lane&3, Y: b2 = (lane>>2)&3)
    const int a = lane >> 4, b1 = lane & 3, b2 = (lane >> 2) & 3;
    const bool okx = a < NB && b1 < NB;
    const int ixg = offx + (a < NB ? a : 0), rhox = AX.rowmap[ixg];
    L.psx = AX.prefix[rhox]; L.cx = AX.rcnt[rhox]; L.px = okx ? AX.P[ixg * BW + (b1 - a + P)] : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int iyg = offy + (r < NB ? r : 0), rhoy = AY.rowmap[iyg];
      const long long psv = AY.prefix[rhoy];   // wave-uniform (depends on r only): pin to SGPRs
      L.psy[r] = ((long long)__builtin_amdgcn_readfirstlane((int)(psv >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)(psv & 0xffffffffll));
      L.cy[r] = __builtin_amdgcn_readfirstlane(AY.rcnt[rhoy]); L.py[r] = (r < NB && b2 < NB) ? AY.P[iyg * BW + (b2 - r + P)] : 0;
    }
    // F lane: (fx = lane&3, fy = (lane>>2)&3, slot = lane>>4)
    const int fx = lane & 3, fy = (lane >> 2) & 3;
    double sx = 0, sy = 0;
    if (fx < NB && fy < NB) {
#pragma unroll
      for (int q = 0; q < NB; ++q) { sx += WX[q] * TX[(q * NB + fx) * NDER]; sy += WYq[q] * TY[(q * NB + fy) * NDER]; }
    }
    L.sxy = pa.forcing * (sx * sy) * (AX.J[elx] * AY.J[ely]);
    L.frowxy = rs[X] * AX.rowmap[offx + (fx < NB ? fx : 0)] + rs[Y] * AY.rowmap[offy + (fy < NB ? fy : 0)];
    L.fslot = lane >> 4;
    L.stmask = 0;
    if (W == 0 && pa.first_touch) {
      const bool fx1 = first_touch_axis<P>(elx, a < NB ? a : 0, b1 < NB ? b1 : 0, pa.nelx, pa.ftx_lo, pa.ftx_hi, pa.ftx_blocked);
#pragma unroll
      for (int r = 0; r < NB; ++r) if (fx1 && first_touch_axis<P>(ely, r, b2 < NB ? b2 : 0, pa.nely, pa.fty_lo, pa.fty_hi, pa.fty_blocked)) L.stmask |= 1u << r;
    }
  }

  d4_t acc[4][4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = (d4_t){0, 0, 0, 0};
  double Facc = 0;
  double *hold = nullptr;
  if (W == 0) {
    constexpr int HS = WINMODE == 2 ? WINC_DOUBLES : (WINMODE == 1 ? WIN_DOUBLES : (P * (P + 1) / 2) * 4 * HOLD_LD);      // (PACK: the window of band rows takes the place of the hold area)
    hold = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + pencil_lds_bytes(pa.ne_max, GEO, pa.wpb)) + wave * HS;
    for (int i = lane; i < HS; i += 64) hold[i] = 0.0;
  }
  constexpr bool PKIDX = PACK && (RESID || SGEO);      // the window indices packed into 10 registers where 30 are what spills (P2kLaneT)
  P2kLaneT<PKIDX> K2; K2.o = 0;
  if constexpr (PACK) K2 = pencil_p2k_lane<PKIDX>(lane);
  // mapped geometry: this wavefront's metric area, the raw basis rows, the Gauss weights of this lane's point on axes X, Y
  double *geo = nullptr; const double *uxr = nullptr, *vyr = nullptr; double wjxy = 0, wt[4] = {1, 1, 1, 1};
  constexpr bool rational = GEO && RAT;
  if constexpr (GEO) {
    geo = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + pencil_lds_bytes(pa.ne_max, true) + (WINMODE == 2 ? pencil_winc_bytes(8) : (WINMODE == 1 ? pencil_win_bytes(8) : pencil_hold_bytes(P)))) + wave * GD;
    vyr = L.vy - ((lane >> 2) & 3) * 8; uxr = vyr + pa.wpb * 32;
    const int gqx = lane & 3, gqy = (lane >> 2) & 3;
    if (gqx < NB && gqy < NB) wjxy = (AX.w[elx * NB + gqx] * AX.J[elx]) * (AY.w[ely * NB + gqy] * AY.J[ely]);
  }
  PencilBC bc; bc.any = false; bc.xlo = bc.xhi = bc.ylo = bc.yhi = false; bc.wlo = bc.whi = -1000;
  bc.vwlo = bc.vwhi = bc.vxlo = bc.vxhi = bc.vylo = bc.vyhi = 0;
  if (W == 0 && (SYSTEM || STATE)) {
    bc.xlo = !AX.periodic && S.bcv[X][0].count > 0 && elx + AX.estart == 0;              bc.vxlo = S.bcv[X][0].value[0];
    bc.xhi = !AX.periodic && S.bcv[X][1].count > 0 && elx + AX.estart == AX.esizes - 1;  bc.vxhi = S.bcv[X][1].value[0];
    bc.ylo = !AY.periodic && S.bcv[Y][0].count > 0 && ely + AY.estart == 0;              bc.vylo = S.bcv[Y][0].value[0];
    bc.yhi = !AY.periodic && S.bcv[Y][1].count > 0 && ely + AY.estart == AY.esizes - 1;  bc.vyhi = S.bcv[Y][1].value[0];
    if (!AW.periodic && S.bcv[W][0].count > 0 && AW.estart == 0) { bc.wlo = AW.off[0]; bc.vwlo = S.bcv[W][0].value[0]; }
    if (!AW.periodic && S.bcv[W][1].count > 0 && AW.estart + AW.nel == AW.esizes) { bc.whi = AW.off[AW.nel - 1] + P; bc.vwhi = S.bcv[W][1].value[0]; }
    bc.any = bc.xlo || bc.xhi || bc.ylo || bc.yhi || bc.wlo > -1000 || bc.whi > -1000;
  }
  if constexpr (RESID) {             // the F rows of (a_y, a_x) = (lane / 3, lane % 3) without the walk-axis part, behind the ring (lanes 0..8)
    if (lane < 9) reinterpret_cast<long long *>(rwin + 32)[lane] = rs[X] * AX.rowmap[offx + lane % 3] + rs[Y] * AY.rowmap[offy + lane / 3];
  }
  double Hsum[SGEO ? 5 : 1][4];      // SGEO: the first-order sums of the next element's points, from its flush-phase half to its MFMA-phase half
  auto geometry = [&](int ei) {   // control points, NURBS weights of the lane's basis functions and the metric of element wh + ei
    if constexpr (GEO) {
      // the element's walk-axis rows, unscaled, where the MFMA phase reads them without a trip to memory (a global load
      // there stalls the in-order MFMA issue for its whole latency)
      if (lane < 32) { const int q = lane >> 3, a = (lane >> 1) & 3, k = lane & 1; geo[GZ + lane] = (q < NB && a < NB) ? AW.tab[((size_t)ew(ei) * NB * NB + q * NB + a) * NDER + k] : 0.0; }
      if constexpr (STATE) { if (lane >= 32 && lane < 48) { const int l2 = lane - 32, q = l2 >> 2, a = l2 & 3; d2w[lane] = (q < NB && a < NB) ? AW.tab[((size_t)ew(ei) * NB * NB + q * NB + a) * NDER + 2] : 0.0; } }
      if constexpr (!IDENT && !SGEO) pencil_geo_ctrl<P>(geo, S, lane, AW.off[ew(ei)], offx, offy, wt);
      else if constexpr (IDENT) { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }     // (the walk-axis rows above)
      const int gqw = lane >> 4;
      if constexpr (STATE) {
        // this lane's node (aw, ay, ax) = (lane>>4, (lane>>2)&3, lane&3): coefficient of U, or the Dirichlet value there
        const int aw = lane >> 4, ay = (lane >> 2) & 3, ax = lane & 3;
        double uc = 0.0, vc = 0.0;
        if (aw < NB && ay < NB && ax < NB) {
          const int li = ei + aw;                 // (one new node layer per element: AW.off[wh + ei] - T.lay0 = ei)
          const long long urow = (long long)T.rho[li] * rs[W] + rs[X] * AX.rowmap[offx + ax] + rs[Y] * AY.rowmap[offy + ay];
          uc = out.U[urow];
          if constexpr (RESID) vc = out.V ? out.V[urow] : 0.0;
          double fv = 0;
          if (bc.any && pencil_fixed<P, false>(bc, ax, ay, T.lay0 + li, fv)) { uc = S.fixtable ? S.fixtable[urow] : fv; vc = 0.0; }     // (IGASetFixTable: the value by row; IGAElementDelValues for V)
        }
        if constexpr (SGEO) {
          pencil_sgeo_ctrl<P>(geo, S, lane, AW.off[ew(ei)], offx, offy, uc);
          pencil_sgeo_sums<P, RAT, Form>(geo, d2w, lane, uxr, vyr, geo + GZ, Hsum);
        } else {
        double xpar[3] = {0, 0, 0};
        const int gqx = lane & 3, gqy = (lane >> 2) & 3;
        if (gqx < NB && gqy < NB && gqw < NB) { xpar[0] = AW.pt[ew(ei) * NB + gqw]; xpar[1] = AX.pt[elx * NB + gqx]; xpar[2] = AY.pt[ely * NB + gqy]; }
        pencil_state_eval<P, Form, RESID>(geo, d2w, lane, uxr, vyr, geo + GZ, wjxy * (gqw < NB ? T.wq[ei * 4 + gqw] : 0.0), uc, prm, out.shift, out.t, xpar);
        if constexpr (RESID) vscr[lane] = vc;      // (u_t of this element is summed at the head of its MFMA phase: pencil_resid_ut)
        }
      } else
      if constexpr (is_builtin_gram<Form>::v) pencil_geo_eval<P>(geo, lane, uxr, vyr, geo + GEO_Z, wjxy * (gqw < NB ? T.wq[ei * 4 + gqw] : 0.0), pa.forcing, rational, out.errflag);
      else {
        double xpar[3] = {0, 0, 0};
        if constexpr (IDENT) {
          const int gqx = lane & 3, gqy = (lane >> 2) & 3;
          if (gqx < NB && gqy < NB && gqw < NB) { xpar[0] = AW.pt[ew(ei) * NB + gqw]; xpar[1] = AX.pt[elx * NB + gqx]; xpar[2] = AY.pt[ely * NB + gqy]; }
        }
        pencil_form_eval<P, Form, IDENT>(geo, lane, uxr, vyr, geo + GEO_Z, wjxy * (gqw < NB ? T.wq[ei * 4 + gqw] : 0.0), rational, out.errflag, prm, out.shift, out.t, xpar);
      }
    }
  };
  // Round 6 (a Tangent on the identity geometry): the global loads of the NEXT element's state -- its walk-axis rows, the lane's
  // coefficient of U, the point's walk coordinate -- leave at the START of the flush phase, ahead of the band row's read-add-write,
  // instead of behind it: the two round trips overlap (each sat behind a wavefront fence the compiler does not move loads across).
  constexpr bool SPLIT = STATE && IDENT && !SGEO;
  struct StatePre { double zv, uc, vc, xw; };
  const int sp_aw = lane >> 4, sp_ay = (lane >> 2) & 3, sp_ax = lane & 3;
  const bool sp_node = sp_aw < NB && sp_ay < NB && sp_ax < NB;
  long long sp_rowxy = 0; double sp_x1 = 0, sp_x2 = 0;
  if constexpr (SPLIT) {
    if (sp_node) sp_rowxy = rs[X] * AX.rowmap[offx + sp_ax] + rs[Y] * AY.rowmap[offy + sp_ay];
    if (sp_ax < NB && sp_ay < NB) { sp_x1 = AX.pt[elx * NB + sp_ax]; sp_x2 = AY.pt[ely * NB + sp_ay]; }      // (this lane's point (q_x, q_y) = (lane & 3, (lane >> 2) & 3): constant along the walk)
  }
  auto state_load = [&](int ei, StatePre &sp) {
    sp.zv = 0.0; sp.uc = 0.0; sp.vc = 0.0; sp.xw = 0.0;
    if constexpr (SPLIT) {
    if (lane < 32) { const int q = lane >> 3, a = (lane >> 1) & 3, k = lane & 1; if (q < NB && a < NB) sp.zv = AW.tab[((size_t)ew(ei) * NB * NB + q * NB + a) * NDER + k]; }
    else if (lane < 48) { const int l2 = lane - 32, q = l2 >> 2, a = l2 & 3; if (q < NB && a < NB) sp.zv = AW.tab[((size_t)ew(ei) * NB * NB + q * NB + a) * NDER + 2]; }
    if (sp_node) {
      const int li = ei + sp_aw;
      const long long urow = (long long)T.rho[li] * rs[W] + sp_rowxy;
      sp.uc = out.U[urow];
      if constexpr (RESID) sp.vc = out.V ? out.V[urow] : 0.0;
      double fv = 0;
      if (bc.any && pencil_fixed<P, false>(bc, sp_ax, sp_ay, T.lay0 + li, fv)) { sp.uc = S.fixtable ? S.fixtable[urow] : fv; sp.vc = 0.0; }
    }
    if (sp_aw < NB) sp.xw = AW.pt[ew(ei) * NB + sp_aw];
    }
  };
  auto state_eval = [&](int ei, const StatePre &sp) {
    if constexpr (SPLIT) {
    if (lane < 32) geo[GZ + lane] = sp.zv;
    else if (lane < 48) d2w[lane] = sp.zv;
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double xpar[3] = {0, 0, 0};
    const int gqw = lane >> 4;
    if (sp_ax < NB && sp_ay < NB && gqw < NB) { xpar[0] = sp.xw; xpar[1] = sp_x1; xpar[2] = sp_x2; }
    pencil_state_eval<P, Form, RESID>(geo, d2w, lane, uxr, vyr, geo + GZ, wjxy * (gqw < NB ? T.wq[ei * 4 + gqw] : 0.0), sp.uc, prm, out.shift, out.t, xpar);
    if constexpr (RESID) vscr[lane] = sp.vc;
    }
  };
  if constexpr (SPLIT) { StatePre sp0; state_load(0, sp0); state_eval(0, sp0); }
  else geometry(0);

  PencilFix<FIXT> fxt;
  if constexpr (FIXT) { fxt.table = S.fixtable; fxt.rmx = AX.rowmap + offx; fxt.rmy = AY.rowmap + offy; fxt.sx = rs[X]; fxt.sy = rs[Y]; fxt.rho = T.rho; fxt.lay0 = T.lay0; fxt.nl = nl; }
  int held[4] = {0, 0, 0, 0};   // elements walked so far that hold the layer in window slot t

  // Ping-pong schedule.  Wavefronts w and w+4 of this 512-thread workgroup share a SIMD; group 0 (waves 0-3)
  // and group 1 (waves 4-7) run half a period apart, separated by s_barrier, so that on every SIMD one
  // wavefront issues its MFMAs while the other one does its read-modify-write:
  //   group 0:  mfma(0) | flush(0) | mfma(1) | flush(1) | ...
  //   group 1:          | mfma(0)  | flush(0)| mfma(1)  | ...
  const int grp = wave >> 2;
  long long tk0 = 0, tw0 = 0;
  if (out.clk) { tk0 = __builtin_readcyclecounter(); tw0 = wall_clock64(); }
  const bool pingpong = pa.free_run == 0;
  if (grp == 1 && pingpong) __builtin_amdgcn_s_barrier();
  int lay = T.lay0;
  if (kDebug && pa.debug_buf) tw_loop = __builtin_readcyclecounter();
  for (int ei = 0; ei < ne; ++ei) {
    lay = T.lay0 + ei;          // walk condition: one new layer per element, local basis a_w sits in tile slot a_w
    if constexpr (PKIDX) { int z = 0; asm volatile("" : "+s"(z)); K2.o = z; }      // (P2kLaneT<true>: the window indices are decoded per element)
    const double *zt = T.zt + ei * 32, *wqs = T.wq + ei * 4;
    long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
    if (kDebug && pa.debug_buf) tq0 = __builtin_readcyclecounter();
    const double *ztg = geo + GZ;   // GEO: the element's raw walk-axis rows [q][a][2] (staged with its metric)
    if (SYSTEM && GEO) {   // F of this element (its point values are in the metric area).  Here, ahead of this wave's own MFMAs and
      // while the partner wavefront is in its memory-bound flush: fp64 VALU work issued next to the PARTNER's MFMA stream waits
      // about one MFMA (64 cycles) per instruction -- in the flush phase these 64 multiply-adds per lane kept the System driver's
      // flush (46k cycles) longer than the MFMA phase (41k).
      const int fs = lane >> 4;
      const double wa = rational ? (fs == 0 ? wt[0] : (fs == 1 ? wt[1] : (fs == 2 ? wt[2] : wt[3]))) : 1.0;
      Facc += wa * pencil_f_geo<NB>(geo, lane, uxr, vyr, ztg);
    }
    if constexpr (SGEO) {
      const int gqw = lane >> 4;
      pencil_sgeo_point<P, RAT, Form>(geo, lane, Hsum, wjxy * (gqw < NB ? T.wq[ei * 4 + gqw] : 0.0), prm, out.shift, out.t, out.errflag);
      if constexpr (PACK) {
        d4_t pk[4];
        pencil_mfma_state_geo_p2k<RAT, Form, P2kLaneT<PKIDX>>(pk, uxr, vyr, ztg, d2w, geo, K2, lane);
        pencil_winc_add_ns(hold, pk, K2, ei);
      } else
      pencil_mfma_state_geo_p2<RAT, Form>(acc, uxr, vyr, ztg, d2w, geo, lane);
    }
    else if constexpr (STATE && PACK) {
      d4_t pk[4];
      if constexpr (RESID) pencil_resid_ut<P, Form>(geo, vscr, lane, uxr, vyr, ztg);
      pencil_mfma_state_p2k<Form, RESID, P2kLaneT<PKIDX>>(pk, uxr, vyr, ztg, d2w, geo, K2, lane);
      pencil_win_add_ns(hold, pk, K2, ei);      // (measured: in the flush phase instead, behind the barrier, 130.4 -> 125.8 M el/s at 128^3)
      if constexpr (RESID) pencil_rwin_add(rwin, pk, K2, ei, lane);
    }
    else if constexpr (STATE && P == 2) pencil_mfma_state_p2<Form>(acc, uxr, vyr, ztg, d2w, geo, lane);
    else if constexpr (STATE) pencil_mfma_state<NB, Form>(acc, L.u0, L.u1, u2, L.vy, d2w + 16 + ((lane >> 2) & 3) * 4, ztg, d2w + 32, geo, lane);
    else if constexpr (GEO && P == 2) pencil_mfma_geo_p2<RAT>(acc, uxr, L.vy, ztg, geo, lane, wt);
    else if constexpr (GEO) pencil_mfma_geo<NB, RAT>(acc, L.u0, L.u1, L.vy, ztg, geo, lane, wt);
    else if constexpr (PACK) {
      d4_t pk[3];
      const double *vys = L.vy - ((lane >> 2) & 3) * 8;
      pencil_mfma_p2k(pk, vys + pa.wpb * 32, vys, zt, K2, lane);
      pencil_win_add(hold, pk, K2, ei);
    }
    else if constexpr (P == 2 && W == 0) pencil_mfma_p2(acc, L.vy - ((lane >> 2) & 3) * 8 + pa.wpb * 32, L.vy, zt, lane);
    else pencil_mfma<W, W == 0, NB>(acc, L, zt);
    if (kDebug && pa.debug_buf) tq1 = __builtin_readcyclecounter();
    if (SYSTEM && !GEO) {   // F_a += f * J * prod_d sum_q w N : the walk-axis factor is sum_q sqrt(wJ) * (sqrt(wJ) N)
      double sw = 0;
      const int fs = L.fslot < NB ? L.fslot : 0;
#pragma unroll
      for (int q = 0; q < NB; ++q) sw += wqs[q] * zt[(q * 4 + fs) * 2];
      if (L.fslot < NB) Facc += L.sxy * sw;
    }
    if (pingpong) __builtin_amdgcn_s_barrier();
    if (kDebug && pa.debug_buf) tq2 = __builtin_readcyclecounter();
    // the partner wavefront on this SIMD now streams MFMAs (one issue slot per 64 cycles); without priority
    // the younger wavefront's address arithmetic only gets the left-over VALU slots (measured: 12k vs 60k cycles)
    __builtin_amdgcn_s_setprio(3);
    StatePre spn;
    if constexpr (SPLIT) { if (ei + 1 < ne) state_load(ei + 1, spn); }
#pragma unroll
    for (int t = 0; t < NB; ++t) held[t]++;
    if constexpr (RESID) pencil_rwin_leave<P>(rwin, lane, T, nl, out, S.fixtable, lay, own_lo, own_hi, bc, held[0], rs[W]);
    if (kDebug && pa.debug_buf && pa.debug_noflush == 2) tq0 = __builtin_readcyclecounter();      // (IGX_DEBUG_NOFLUSH=2: the stamps split the flush phase: Residual's leave | band row's leave | next element's state)
    if constexpr (W == 0) pencil0_leave<SYSTEM, P, FIXT, STATE, SYSTEM || STATE, (GEO && SYSTEM && P == 3) ? 2 : P + 1, WINMODE>(acc, Facc, hold, lane, L, T, nl, out, lay, own_lo, own_hi, T0, T10, bc, held[0], fxt);
    else pencil_leave<SYSTEM, W>(acc, Facc, L, T, nl, out, lay, own_lo, own_hi, T0, T10, rs[W]);
    if (kDebug && pa.debug_buf && pa.debug_noflush == 2) tq1 = __builtin_readcyclecounter();
#pragma unroll
    for (int t = 0; t < NB - 1; ++t) held[t] = held[t + 1];
    held[NB - 1] = 0;
    if constexpr (SPLIT) { if (ei + 1 < ne) state_eval(ei + 1, spn); }
    else if (GEO && ei + 1 < ne) geometry(ei + 1);   // the next element's metric, while the partner wavefront streams its MFMAs
    __builtin_amdgcn_s_setprio(0);
    if (kDebug && pa.debug_buf) { tq3 = __builtin_readcyclecounter(); if (pa.debug_noflush == 2) { const long long a = tq0, b = tq1; tq0 = tq2; tq1 = a; tq2 = b; }      // (columns: "mfma" = the Residual's leave, "wait" = the band row's leave, "flush" = the next element's state)
      if ((wave & 3) == 0 && wave < 8 && lane == 0 && ei < 62) { long long *d = pa.debug_buf + (((size_t)blockIdx.x * 2 + (wave >> 2)) * 64 + ei) * 4; d[0] = tq0; d[1] = tq1; d[2] = tq2; d[3] = tq3; } }
    if (pingpong) __builtin_amdgcn_s_barrier();
  }
  if (kDebug && pa.debug_buf) tw_loopend = __builtin_readcyclecounter();
  if (out.clk && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && wave == 0 && lane == 0) {
    // IGX_CLOCK_PROBE: s_memtime against the 100 MHz s_memrealtime over the walk, first and last workgroup of every launch
    atomicAdd(reinterpret_cast<unsigned long long *>(out.clk), (unsigned long long)(__builtin_readcyclecounter() - tk0));
    atomicAdd(reinterpret_cast<unsigned long long *>(out.clk) + 1, (unsigned long long)(wall_clock64() - tw0));
    atomicAdd(reinterpret_cast<unsigned long long *>(out.clk) + 2, (unsigned long long)ne);
  }
  if (grp == 0 && pingpong) __builtin_amdgcn_s_barrier();
  if (seg == pa.nseg - 1 && !alias0 && !pa.open_hi)       // the last segment also owns what is still in the window (wrapped axis: the first segment does)
    for (int k = 1; k <= P; ++k) {
      if constexpr (W == 0) {
        if constexpr (RESID) pencil_rwin_leave<P>(rwin, lane, T, nl, out, S.fixtable, lay + k, own_lo, own_hi, bc, held[0], rs[W]);
        pencil0_leave<SYSTEM, P, FIXT, STATE, SYSTEM || STATE, (GEO && SYSTEM && P == 3) ? 2 : P + 1, WINMODE>(acc, Facc, hold, lane, L, T, nl, out, lay + k, own_lo, own_hi, T0, T10, bc, held[0], fxt);
#pragma unroll
        for (int t = 0; t < NB - 1; ++t) held[t] = held[t + 1];
        held[NB - 1] = 0;
      } else pencil_leave<SYSTEM, W>(acc, Facc, L, T, nl, out, lay + k, own_lo, own_hi, T0, T10, rs[W]);
    }
  if (kDebug && pa.debug_buf && wave == 0 && lane == 0) {      // (slots 62, 63 of the workgroup's stamp area: segments are shorter than 62 elements where this is read)
    long long *d = pa.debug_buf + (((size_t)blockIdx.x * 2) * 64 + 62) * 4;
    d[0] = tw_entry; d[1] = tw_staged; d[2] = tw_loop; d[3] = tw_loopend; d[4] = __builtin_readcyclecounter(); d[5] = ne;
    // the workgroup's record behind the stamps: entry / exit on the shared 100 MHz clock, HW_ID (wave, SIMD, CU, SH, SE), XCC_ID
    long long *w = pa.debug_buf + (size_t)gridDim.x * 2 * 64 * 4 + (size_t)blockIdx.x * 4;
    w[0] = tw_wall; w[1] = wall_clock64(); w[2] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4); w[3] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
  }
}

template <bool SYSTEM, int W, int P, bool GEO = false, bool RAT = false, bool FIXT = false, int ALIAS = -1>
__global__ void __launch_bounds__(512, 2)
gram_pencil(SpaceDev S, OutDev out, PencilArgs pa) {
  gram_pencil_body<SYSTEM, W, P, GEO, RAT, FIXT, void, false, ALIAS>(S, out, pa, nullptr);
}

// Twelve-wave workgroups for the free-running p = 2 walk on the identity geometry (config 2): without the ping-pong nothing ties the
// waves to pairs, the kernel needs 154-176 VGPRs -- two waves per SIMD with eight-wave workgroups, a third of the register file idle --
// and a twelve-wave workgroup puts three waves on every SIMD (168 VGPRs each): the MFMA pipe finds a ready wave more often.
// (Two six-wave workgroups per CU do not do it: a workgroup's waves 4 and 5 land on SIMDs 0 and 1 again, a second workgroup does not fit
// there and the CU runs six waves: 140 M el/s against 175.)
// PACK: the element's 27 basis functions in two tiles of 16 rows, 63 MFMAs per element instead of 126, the band rows combined in an LDS
// window (pencil_mfma_p2k); IGX_P2_PACK=0 selects the layer-pair tiles above.
template <bool SYSTEM, int P, int ALIAS, bool PACK = false>
__global__ void __launch_bounds__(768, 3)
gram_pencil_w6(SpaceDev S, OutDev out, PencilArgs pa) {
  gram_pencil_body<SYSTEM, 0, P, false, false, false, void, false, ALIAS, PACK>(S, out, pa, nullptr);
}

// the same walk for a run-time scalar form (rtc.hpp compiles this instantiation with hiprtc; IDENT: no geometry)
template <bool SYSTEM, int P, bool IDENT, bool RAT, class Form>
__global__ void __launch_bounds__(512, 2)
form_pencil(SpaceDev S, OutDev out, PencilArgs pa, ParamsDev prm) {
  gram_pencil_body<SYSTEM, 0, P, true, RAT, false, Form, IDENT>(S, out, pa, prm.v);
}

// the walk for the Tangent of a nonlinear scalar form without a geometry (pencil_state_eval / pencil_mfma_state above)
template <int P, class Form>
__global__ void __launch_bounds__(512, 2)
state_pencil(SpaceDev S, OutDev out, PencilArgs pa, ParamsDev prm) {
  static_assert(pencil_state_of<Form>::v, "state_pencil: the form declares PENCIL_NFEAT, PENCIL_NC, pencil_coef and pencil_trial");
  gram_pencil_body<false, 0, P, true, false, false, Form, true>(S, out, pa, prm.v);
}

// ... at p = 2 with packed tiles and the LDS window of band rows (pencil_mfma_state_p2k; IGX_P2_PACK=0: state_pencil<2, Form>)
template <class Form>
__global__ void __launch_bounds__(512, 2)
state_pencil_k(SpaceDev S, OutDev out, PencilArgs pa, ParamsDev prm) {
  static_assert(pencil_state_of<Form>::v, "state_pencil_k: the form declares PENCIL_NFEAT, PENCIL_NC, pencil_coef and pencil_trial");
  gram_pencil_body<false, 0, 2, true, false, false, Form, true, -1, true>(S, out, pa, prm.v);
}

// ... and on a mapped geometry (p = 2: pencil_sgeo_sums / pencil_sgeo_point / pencil_mfma_state_geo_p2)
// ... with the Residual on the same MFMAs (one pass for IFunction + IJacobian: IGXComputeIFunctionIJacobian)
template <class Form>
__global__ void __launch_bounds__(512, 2)
state_pencil_kr(SpaceDev S, OutDev out, PencilArgs pa, ParamsDev prm) {
  static_assert(pencil_resid_of<Form>::v, "state_pencil_kr: the form declares PENCIL_NCR, pencil_coef_r and pencil_resid");
  gram_pencil_body<false, 0, 2, true, false, false, Form, true, -1, true, true>(S, out, pa, prm.v);
}

template <bool RAT, class Form>
__global__ void __launch_bounds__(512, 2)
state_pencil_geo_k(SpaceDev S, OutDev out, PencilArgs pa, ParamsDev prm) {      // packed tiles, the compact window of band rows
  static_assert(pencil_state_of<Form>::v, "state_pencil_geo_k: the form declares PENCIL_NFEAT, PENCIL_NC, pencil_coef and pencil_trial");
  gram_pencil_body<false, 0, 2, true, RAT, false, Form, false, -1, true>(S, out, pa, prm.v);
}
template <int P, bool RAT, class Form>
__global__ void __launch_bounds__(512, 2)
state_pencil_geo(SpaceDev S, OutDev out, PencilArgs pa, ParamsDev prm) {
  static_assert(pencil_state_of<Form>::v, "state_pencil_geo: the form declares PENCIL_NFEAT, PENCIL_NC, pencil_coef and pencil_trial");
  gram_pencil_body<false, 0, P, true, RAT, false, Form, false>(S, out, pa, prm.v);
}

#ifndef IGX_RTC
// ------------------------------------------------------------------ dispatch

template <bool SYSTEM>
static void launch_elements(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, const Box &bx, const GramArgs &g0, int &launches) {
  for (int d = 0; d < 3; ++d) if (bx.hi[d] <= bx.lo[d]) return;
  for (int c2 = 0; c2 < s.lay[2].ncolors; ++c2) for (int c1 = 0; c1 < s.lay[1].ncolors; ++c1) for (int c0 = 0; c0 < s.lay[0].ncolors; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool ok = true;
    for (int d = 0; d < 3 && ok; ++d) ok = color_range(s.lay[d], cc[d], bx.lo[d], bx.hi[d], cr.start[d], cr.step[d], cr.count[d]);
    if (!ok) continue;
    GramArgs ga = g0; ga.nwaves = cr.count[0] * cr.count[1] * cr.count[2];
    hipLaunchKernelGGL((gram_p3_element<SYSTEM>), dim3((unsigned)((ga.nwaves + 3) / 4)), dim3(256), 0, stream, S, out, cr, ga);
    launches++;
  }
}

// a run-time form's instantiation of the walk (rtc.hpp): the module function for this driver / degree / geometry, its parameters
// ... or a compiled-in instantiation for a built-in form (state_pencil: kfn, its extra LDS, flops per element for the roofline line)
typedef void (*PencilKernel)(SpaceDev, OutDev, PencilArgs, ParamsDev);
// one pass of an assembly that comes in several (the elements next to the upper faces first, a mark for the exchange after each):
// the first-touch rule of axis X (like fty for axis Y), and, when the pass is a part of the walk axis, how its ends join the others
struct PencilPass { int ftx[3] = {0, 0x7fffffff, 0x7fffffff}; int halo_lo = -1; bool open_hi = false; };
struct PencilModule { hipFunction_t fn = nullptr; ParamsDev prm; std::string name; PencilKernel kfn = nullptr; size_t extra_lds = 0; bool state = false; double flop_per_element = 0;
                      bool state_geo = false;         // state_geo: the instantiation is state_pencil_geo (evaluates the geometry itself)
                      int pack = 0;
                      const void *patch_kfn = nullptr; };      // gram_patch.hpp: state_patch_p2<Form> of the same form (p = 2, no geometry)                // pack: packed tiles, the window of band rows in the place of the hold areas (1: state_pencil_k; 2: the compact window, state_pencil_geo_k)

static inline int nseg_min_lds(int nw) { return std::max(1, (nw + 159) / 160); }
// a launch that could not be made (the LDS of the chosen segments beyond the device's, a module launch refused): try_gram_mfma
// reports it instead of the generic "kernel launch failed" of a later call
inline const char *&pencil_launch_error() { static thread_local const char *e = nullptr; return e; }

// Segments of one launch (the pencils of one colour, nw walked elements): as few as possible -- every segment re-computes P halo
// elements -- but enough workgroups for the CUs.  One 8-pencil workgroup per CU at a time (LDS; two where the tables are small), so a
// launch takes ceil(workgroups / slots) rounds of (segment length + halo): the count with the least rounds x length.  Returns the
// count; *cost = that product (element-steps of the launch's critical path).
static inline int pencil_cus() {
  static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  return ncu;
}
static inline int pencil_segments(long long pencils, int nw, int P, bool geo, bool walk0, size_t extra_lds, bool halo_always, long long *cost_out = nullptr, int wpb = 8) {
  const int ncu = pencil_cus();
  const long long bps = (pencils + wpb - 1) / wpb;
  int nseg = std::max(1, (nw + 159) / 160);
  long long best = -1; int best_n = nseg;
  // (round 6: segments down to two elements.  A pencil's elements are sequential -- 12.8 us of MFMA issue each at p = 3 -- so on a small
  //  mesh, where the CUs outnumber the workgroups, the walk's LENGTH is the launch's duration: 32^3 took 365 us per launch with the old
  //  floor of eight elements per segment, 250 us with four.  The "+ 1" is a workgroup's set-up in units of a step.)
  for (int n = nseg; n <= std::max(nseg, nw / 2); ++n) {
    const int len = (nw + n - 1) / n, ns = (nw + len - 1) / len;
    const size_t lds_n = pencil_lds_bytes(len + 3, geo, wpb) + (walk0 ? pencil_hold_bytes(P) * wpb / 8 : 0) + (geo ? pencil_geo_bytes() : 0) + extra_lds;
    // (next to the metric areas of a mapped geometry the tables of 128 + 3 elements no longer fit: 256^3 takes three segments there)
    if (lds_n > (size_t)160 * 1024) { if (best < 0) best_n = n + 1; continue; }
    const long long slots = (long long)ncu * std::max<long long>(1, std::min<long long>(2, (long long)(160 * 1024) / (long long)lds_n));   // resident workgroups
    const long long cost = ((bps * ns + slots - 1) / slots) * (len + ((ns > 1 || halo_always) ? P : 0) + 1);
    if (best < 0 || cost < best) { best = cost; best_n = n; }
  }
  if (cost_out) *cost_out = best < 0 ? 0 : best;
  return best_n;
}
// the same summed over the colour launches of a box of elements (axis-0 walk): what a pass of an assembly costs
// waves per workgroup of the axis-0 walk: twelve for the free-running p = 2 kernel on the identity geometry (gram_pencil_w6), else eight
static inline int pencil_wpb(const Space &s, int P, bool geo, bool fixt, bool has_mod) {
  static const int wpb_env = [] { const char *e = getenv("IGX_WPB"); return e ? atoi(e) : 0; }();
  const int free_run = s.env.free_run >= 0 ? s.env.free_run : ((P == 2 && !geo && !has_mod) ? 1 : 0);
  return (P == 2 && !geo && !fixt && !has_mod && free_run && wpb_env != 8) ? 12 : 8;
}
// ... and whether that kernel packs the element's 27 functions into two tiles (pencil_mfma_p2k; IGX_P2_PACK=0: the layer-pair tiles)
static inline bool pencil_p2_pack(const Space &s, int P, bool geo, bool fixt, bool has_mod) {
  return s.env.p2_pack != 0 && pencil_wpb(s, P, geo, fixt, has_mod) == 12;
}
// LDS the packed p = 2 walks take on top of what pencil_segments counts itself: the window of band rows in the place of the hold
// areas (about + 28 KB at eight waves, + 43 KB at twelve).  One function for the launcher and for the pass-cost model, so that the
// model's resident-workgroup count and LDS-limited segment lengths are those of the launch.
static size_t pencil_win_extra(const Space &s, int P, bool geo, bool fixt, const PencilModule *mod, int wpb) {
  const bool pack = pencil_p2_pack(s, P, geo, fixt, mod != nullptr) || (mod && mod->pack);
  if (!pack) return 0;
  const size_t win_bytes = (mod && mod->pack == 2) ? pencil_winc_bytes(wpb) : pencil_win_bytes(wpb);
  return win_bytes - pencil_hold_bytes(P) * wpb / 8;
}

static long long pencil_box_cost(const Space &s, int P, bool geo, size_t extra_lds, const Box &bx, bool halo_always, int wpb = 8) {
  long long total = 0;
  for (int d = 0; d < 3; ++d) if (bx.hi[d] <= bx.lo[d]) return 0;
  for (int cy = 0; cy < s.lay[2].ncolors; ++cy) for (int cx = 0; cx < s.lay[1].ncolors; ++cx) {
    int st, sp, nx, ny;
    if (!color_range(s.lay[1], cx, bx.lo[1], bx.hi[1], st, sp, nx) || !color_range(s.lay[2], cy, bx.lo[2], bx.hi[2], st, sp, ny)) continue;
    long long c = 0;
    (void)pencil_segments((long long)nx * ny, bx.hi[0] - bx.lo[0], P, geo, true, extra_lds, halo_always || s.lay[0].alias, &c, wpb);
    total += c;
  }
  return total;
}

template <bool SYSTEM, int W, int P, bool GEO = false, bool RAT = false, bool FIXT = false>
// fty (axis-0 walk): {lo, hi, blocked} of the first-touch rule on axis Y when the assembly comes in two passes over that axis
// (first_touch_axis); null: one pass over the whole axis
static void launch_pencils(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, const Box &bx, double forcing, int &launches, bool first_touch = false, const int *fty = nullptr,
                           const PencilModule *mod = nullptr, const PencilPass *pass = nullptr) {
  constexpr int X = (W == 0) ? 1 : 0, Y = (W == 2) ? 1 : 2;
  for (int d = 0; d < 3; ++d) if (bx.hi[d] <= bx.lo[d]) return;
  const int nw = bx.hi[W] - bx.lo[W];
  for (int cy = 0; cy < s.lay[Y].ncolors; ++cy) for (int cx = 0; cx < s.lay[X].ncolors; ++cx) {
    PencilArgs pa; pa.forcing = forcing;
    pa.first_touch = (W == 0 && first_touch) ? 1 : 0; pa.nelx = s.elem_width[X]; pa.nely = s.elem_width[Y];
    pa.alias0 = (W == 0 && s.lay[0].alias) ? 1 : 0;
    pa.fty_lo = fty ? fty[0] : 0; pa.fty_hi = fty ? fty[1] : 0x7fffffff; pa.fty_blocked = fty ? fty[2] : 0x7fffffff;
    pa.ftx_lo = pass ? pass->ftx[0] : 0; pa.ftx_hi = pass ? pass->ftx[1] : 0x7fffffff; pa.ftx_blocked = pass ? pass->ftx[2] : 0x7fffffff;
    pa.w_halo_lo = (pass && pass->halo_lo >= 0) ? pass->halo_lo : bx.lo[W]; pa.open_hi = (pass && pass->open_hi) ? 1 : 0;
    if (!color_range(s.lay[X], cx, bx.lo[X], bx.hi[X], pa.ex_start, pa.ex_step, pa.ex_count)) continue;
    if (!color_range(s.lay[Y], cy, bx.lo[Y], bx.hi[Y], pa.ey_start, pa.ey_step, pa.ey_count)) continue;
    const long long pencils = (long long)pa.ex_count * pa.ey_count;
    // p = 2 on the identity geometry: the flush is the longer phase and the rigid ping-pong makes the MFMA wave wait for it; left to
    // the SIMD's own arbitration the walk gains 5 % (128^3: 166.7 -> 175.3 M el/s).  Everything at p = 3, mapped geometries and
    // Tangents lose 6-8 % without the barriers (256^3: 64.8 -> 59.5).  IGX_FREE_RUN=0/1 overrides.
    pa.free_run = s.env.free_run >= 0 ? s.env.free_run : ((P == 2 && !GEO && !(mod && mod->state)) ? 1 : 0);
    // ... and free of the pairing, six-wave workgroups put three waves on every SIMD (gram_pencil_w6; IGX_WPB=8: the eight-wave kernel)
    const bool w6 = W == 0 && pencil_wpb(s, P, GEO, FIXT, mod != nullptr) == 12;
    pa.wpb = w6 ? 12 : 8;
    const bool pack = W == 0 && (pencil_p2_pack(s, P, GEO, FIXT, mod != nullptr) || (mod && mod->pack));      // (p = 2 on the identity geometry: packed tiles, the band rows combined in an LDS window instead of the hold areas)
    const size_t win_bytes = (mod && mod->pack == 2) ? pencil_winc_bytes(pa.wpb) : pencil_win_bytes(pa.wpb);
    const size_t win_extra = W == 0 ? pencil_win_extra(s, P, GEO, FIXT, mod, pa.wpb) : 0;
    int nseg = pencil_segments(pencils, nw, P, GEO, W == 0, (mod ? mod->extra_lds : 0) + win_extra, (W == 0 && s.lay[0].alias) || (pass && pass->halo_lo >= 0 && pass->halo_lo < bx.lo[W]), nullptr, pa.wpb);
    if (s.env.nseg > 0) nseg = std::max(nseg_min_lds(nw), std::min(s.env.nseg, std::max(1, nw / 2)));   // experiment switch
    pa.seg_len = (nw + nseg - 1) / nseg; pa.nseg = (nw + pa.seg_len - 1) / pa.seg_len;
    pa.w_lo = bx.lo[W]; pa.w_hi = bx.hi[W];
    pa.blocks_per_seg = (int)((pencils + pa.wpb - 1) / pa.wpb);
    // Launches that do not fill the chip (small and medium meshes), the plain Gram walk: segments AND workgroup size from a model of the
    // step -- measured at p = 3 (profiles/r06_small_meshes.txt): a step of the walk takes 20 us with one wavefront on the SIMD (its
    // MFMAs, then its flush), 31 us with two (the ping-pong pair), 14 us per wavefront beyond; four-wavefront workgroups (one per SIMD,
    // free-running) spread the pencils over twice the CUs.  time = rounds x (length + halo + 1) x step(wavefronts per SIMD on the
    // fullest CU).  Large meshes come out as before (eight wavefronts, the fewest segments that fit the LDS).  IGX_SMALL_WPB=0: off;
    // IGX_NSEG forces the segments and keeps the eight.
    if (W == 0 && !GEO && !mod && pa.wpb == 8 && s.env.small_wpb && s.env.nseg <= 0) {
      const int ncu = pencil_cus();
      const bool halo_always = (W == 0 && s.lay[0].alias) || (pass && pass->halo_lo >= 0 && pass->halo_lo < bx.lo[W]);
      double best = 1e300; int bn = nseg, bw = 8;
      for (int wc = 8; wc >= 4; wc -= 4)
        for (int n = nseg_min_lds(nw); n <= std::max(nseg_min_lds(nw), nw / 2); ++n) {
          const int len = (nw + n - 1) / n, ns = (nw + len - 1) / len;
          const size_t lds_n = pencil_lds_bytes(len + 3, false, wc) + pencil_hold_bytes(P) * wc / 8;
          if (lds_n > (size_t)160 * 1024) continue;
          const long long rmax = std::max<long long>(1, std::min<long long>(wc == 8 ? 2 : 4, (long long)(160 * 1024) / (long long)lds_n));
          const long long wgs = ((pencils + wc - 1) / wc) * ns, m = (wgs + ncu - 1) / ncu, rounds = (m + rmax - 1) / rmax;
          const long long wsimd = std::min(m, rmax) * wc / 4;
          double step = wsimd <= 1 ? 20.0 : (wsimd == 2 ? 31.0 : 14.0 * (double)wsimd);
          if (wc == 4 && wsimd >= 2) step *= 1.08;      // (without the barriers the p = 3 walk loses 6-8 % once wavefronts share a SIMD)
          const double cost = (double)rounds * (double)(len + ((ns > 1 || halo_always) ? P : 0) + 1) * step;
          if (cost < best * (1.0 - 1e-9)) { best = cost; bn = n; bw = wc; }
        }
      nseg = bn;
      pa.seg_len = (nw + nseg - 1) / nseg; pa.nseg = (nw + pa.seg_len - 1) / pa.seg_len;
      if (bw == 4) { pa.wpb = 4; pa.free_run = 1; }
      pa.blocks_per_seg = (int)((pencils + pa.wpb - 1) / pa.wpb);
    }
    pa.ne_max = pa.seg_len + 3;
    pa.debug_noflush = s.env.debug_noflush;
    pa.debug_buf = nullptr;
    static int dbg_done = 0, dbg_seen = 0;
    // IGX_DEBUG_TIMING=n: the n-th pencil launch of the process is the one that is stamped (1: the first)
    const bool dbg_t = kDebug && s.env.debug_timing && !dbg_done && ++dbg_seen >= std::max(1, atoi(getenv("IGX_DEBUG_TIMING") ? getenv("IGX_DEBUG_TIMING") : "1"));
    const size_t dbg_blocks = (size_t)pa.blocks_per_seg * pa.nseg;
    const size_t dbg_n = dbg_blocks * 2 * 64 * 4 + dbg_blocks * 4;
    if (dbg_t) { (void)hipMalloc((void **)&pa.debug_buf, dbg_n * 8); (void)hipMemset(pa.debug_buf, 0, dbg_n * 8); }
    const size_t lds = pencil_lds_bytes(pa.ne_max, GEO, pa.wpb) + (W == 0 ? (pack ? win_bytes : pencil_hold_bytes(P) * pa.wpb / 8) : 0) + (GEO ? pencil_geo_bytes() : 0) + (mod ? mod->extra_lds : 0);      // (the hold areas are per wavefront and come last when there is no metric area)
    if (lds > (size_t)160 * 1024) { pencil_launch_error() = "the pencil walk's tables do not fit the 160 KB of LDS for any segment length"; return; }
    if (mod && mod->kfn) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mod->kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(mod->kfn, dim3((unsigned)(pa.blocks_per_seg * pa.nseg)), dim3(512), lds, stream, S, out, pa, mod->prm);
    } else if (mod) {      // form_pencil<SYSTEM, P, IDENT, RAT, UserForm> of a run-time form: a module function (rtc.hpp)
      struct { SpaceDev S; OutDev out; PencilArgs pa; ParamsDev prm; } args;
      memset(&args, 0, sizeof(args));
      args.S = S; args.out = out; args.pa = pa; args.prm = mod->prm;
      size_t asz = sizeof(args);
      void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
      if (hipModuleLaunchKernel(mod->fn, (unsigned)(pa.blocks_per_seg * pa.nseg), 1, 1, 512, 1, 1, (unsigned)lds, stream, nullptr, cfg) != hipSuccess) { pencil_launch_error() = "launch of the run-time instantiation of the pencil walk failed"; return; }
    } else {
    void (*kern)(SpaceDev, OutDev, PencilArgs) = gram_pencil<SYSTEM, W, P, GEO, RAT, FIXT>;
    if constexpr (W == 0 && !GEO) kern = pa.alias0 ? gram_pencil<SYSTEM, W, P, GEO, RAT, FIXT, 1> : gram_pencil<SYSTEM, W, P, GEO, RAT, FIXT, 0>;
    if constexpr (W == 0 && P == 2 && !GEO && !FIXT) {
      if (pack) kern = pa.alias0 ? gram_pencil_w6<SYSTEM, P, 1, true> : gram_pencil_w6<SYSTEM, P, 0, true>;
      else if (w6) kern = pa.alias0 ? gram_pencil_w6<SYSTEM, P, 1> : gram_pencil_w6<SYSTEM, P, 0>;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(pa.blocks_per_seg * pa.nseg)), dim3((unsigned)(pa.wpb * 64)), lds, stream, S, out, pa);
    }
    if (dbg_t) {   // IGX_DEBUG_TIMING=1: cycle stamps of the ping-pong phases of the first launch (diagnostic only)
      dbg_done = 1;
      (void)hipStreamSynchronize(stream);
      std::vector<long long> h(dbg_n);
      (void)hipMemcpy(h.data(), pa.debug_buf, dbg_n * 8, hipMemcpyDeviceToHost);
      double sm = 0, sw = 0, sf = 0, sp = 0; long long cnt = 0; long long hist[16] = {0};
      for (size_t b = 0; b < dbg_blocks * 2; ++b) for (int e = 4; e < 60; ++e) {
        const long long *d = &h[(b * 64 + e) * 4], *dn = &h[(b * 64 + e + 1) * 4];
        if (!d[3] || !dn[0]) continue;
        sm += (double)(d[1] - d[0]); sw += (double)(d[2] - d[1]); sf += (double)(d[3] - d[2]); sp += (double)(dn[0] - d[0]); cnt++;
        long long fb = (d[3] - d[2]) / 8192; if (fb > 15) fb = 15; hist[fb]++;
      }
      fprintf(stderr, "[igx pencil timing] blocks=%d seg_len=%d n=%lld cycles: mfma %.0f | wait@barrier %.0f | flush %.0f | period %.0f\n   flush histogram (8192-cycle bins):",
              pa.blocks_per_seg * pa.nseg, pa.seg_len, cnt, sm / cnt, sw / cnt, sf / cnt, sp / cnt);
      for (int i = 0; i < 16; ++i) fprintf(stderr, " %lld", hist[i]);
      fprintf(stderr, "\n");
      {   // the life of a workgroup (wave 0): entry -> tables staged -> first element -> last element -> end
        double a = 0, b2 = 0, c = 0, d2 = 0, nel = 0; long long nb = 0;
        for (size_t b = 0; b < dbg_blocks; ++b) {
          const long long *d = &h[((b * 2) * 64 + 62) * 4];
          if (!d[0] || !d[4]) continue;
          a += (double)(d[1] - d[0]); b2 += (double)(d[2] - d[1]); c += (double)(d[3] - d[2]); d2 += (double)(d[4] - d[3]); nel += (double)d[5]; nb++;
        }
        if (nb) fprintf(stderr, "[igx pencil timing] workgroup life (wave 0, mean of %lld): staging %.0f | lane set-up %.0f | walk %.0f (%.1f elements) | trailing leaves %.0f cycles\n", nb, a / nb, b2 / nb, c / nb, nel / nb, d2 / nb);
        // ... its spread, per segment of the pencils, and the span of the launch (first entry to last end) on the same counter
        for (int sg = 0; sg < pa.nseg; ++sg) {
          double mn = 1e30, mx = 0, sm2 = 0; long long n2 = 0;
          for (int bb = 0; bb < pa.blocks_per_seg; ++bb) {
            const long long *d = &h[(((size_t)sg * pa.blocks_per_seg + bb) * 2 * 64 + 62) * 4];
            if (!d[0] || !d[4]) continue;
            const double life = (double)(d[4] - d[0]); mn = std::min(mn, life); mx = std::max(mx, life); sm2 += life; n2++;
          }
          if (n2) fprintf(stderr, "[igx pencil timing]   segment %d: %lld workgroups, life min %.0f mean %.0f max %.0f\n", sg, n2, mn, sm2 / n2, mx);
        }
      }
      {   // The launch as a timeline on the 100 MHz clock the XCCs share: where the time between the first entry and the last exit
          // goes.  Per CU (XCC, SE, SH, CU of HW_ID) the workgroups in the order they ran: how late the first one starts (dispatch
          // ramp), the gap between one workgroup's exit and the next one's entry, the tail after its last exit.
        const long long *wr = &h[dbg_blocks * 2 * 64 * 4];
        long long t0 = LLONG_MAX, t1 = 0;
        std::map<long long, std::vector<std::pair<long long, long long>>> cu;
        for (size_t b = 0; b < dbg_blocks; ++b) {
          const long long *w = wr + b * 4;
          if (!w[0] || !w[1]) continue;
          t0 = std::min(t0, w[0]); t1 = std::max(t1, w[1]);
          const long long hw = w[2], key = ((w[3] & 15) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
          cu[key].push_back({w[0], w[1]});
        }
        if (t1 > t0 && !cu.empty()) {
          const double span = (double)(t1 - t0);
          double busy = 0, ramp = 0, gap = 0, tail = 0, life1 = 0, life2 = 0, ramp_max = 0; long long n1 = 0, n2 = 0, ngap = 0; int rounds_max = 0;
          for (auto &kv : cu) {
            auto &v = kv.second; std::sort(v.begin(), v.end());
            ramp += (double)(v.front().first - t0); ramp_max = std::max(ramp_max, (double)(v.front().first - t0));
            tail += (double)(t1 - v.back().second);
            rounds_max = std::max(rounds_max, (int)v.size());
            for (size_t i = 0; i < v.size(); ++i) {
              busy += (double)(v[i].second - v[i].first);
              if (i == 0) { life1 += (double)(v[i].second - v[i].first); n1++; } else { life2 += (double)(v[i].second - v[i].first); n2++; gap += (double)(v[i].first - v[i - 1].second); ngap++; }
            }
          }
          const double ncu = (double)cu.size();
          fprintf(stderr, "[igx pencil timing]   timeline (100 MHz ticks = 10 ns): span %.0f = %.1f us over %d CUs that ran a workgroup, at most %d workgroups on one CU\n", span, span * 0.01, (int)cu.size(), rounds_max);
          fprintf(stderr, "[igx pencil timing]   per CU, mean ticks: first entry after the launch's first %.0f (max %.0f) | busy %.0f | between workgroups %.1f (x %.2f per CU) | idle after its last exit %.0f\n",
                  ramp / ncu, ramp_max, busy / ncu, ngap ? gap / ngap : 0.0, (double)ngap / ncu, tail / ncu);
          fprintf(stderr, "[igx pencil timing]   workgroup life, ticks: first on its CU %.0f (%lld) | later ones %.0f (%lld); CU-time in a workgroup's life %.3f of span x CUs (x %.3f of 256 CUs)\n",
                  n1 ? life1 / n1 : 0.0, n1, n2 ? life2 / n2 : 0.0, n2, busy / (span * ncu), busy / (span * 256.0));
        }
      }
      (void)hipFree(pa.debug_buf);
    }
    launches++;
  }
}

// ---- boundary loads on the identity geometry (IGAElementBuildFix: AddFlux with BoundaryArea's no-geometry branch,
// src/petigaelem.c:1118-1132,1191-1212; IGAElementFixSystem adds them to F_e before the fixed rows are set, :1371-1376).
// A boundary element lumps load * 4 * prod_{i != d} (J_i / nen_i) onto each of its basis functions on the face, so a face node
// receives load * 4 * (sum over its elements on axis t of J_t / nen_t) * (the same on axis u): the two 1-D sums come from the
// host, one thread per face node adds the product to its F row -- unless a Dirichlet value holds that dof (FixSystem discards
// the flux of a fixed row).  Sequential launches per face: one writer per row and launch, fixed order.
static __global__ void k_boundary_loads(FluxArgs F, int nr0, int nr1, double *vec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.nt * F.nu) return;
  int r[3]; r[F.d] = F.rd; r[F.t] = i % F.nt; r[F.u] = i / F.nt;
  for (int a = 0; a < 3; ++a) {        // a row on a Dirichlet face keeps the fixed value
    const int gi = F.gfirst[a] + r[a];
    if ((F.fixlo[a] && gi == 0) || (F.fixhi[a] && gi == F.glast[a])) return;
  }
  const double v = F.value * F.st[r[F.t]] * F.su[r[F.u]];
  if (v != 0.0) vec[(size_t)r[0] + (size_t)nr0 * ((size_t)r[1] + (size_t)nr1 * (size_t)r[2])] += v;
}

// ---- the same on a mapped geometry (BoundaryArea's geometry branch, src/petigaelem.c:1133-1165): the lumped value of a face
// element is load * prod_{i != d} (J_i / nen_i) * dS with dS = sum_q w_t w_u |x_,t x x_,u| over its Gauss points on the face
// (basis of axis d at the end knot: only the first / last layer of the element's control points counts; the routine's own
// Rationalize for NURBS).  One thread per face element fills area[e_u][e_t]; one thread per face node then adds the areas of the
// elements that hold it, in a fixed order (sums over <= (p+1)^2 elements).
struct FaceAreaArgs {
  int d, t, u, side;
  int nt, nu;                    // face elements of the rank on axes t, u
  double *area;                  // [nu][nt]
};
static __global__ void k_face_areas(SpaceDev S, FaceAreaArgs F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.nt * F.nu) return;
  int el[3]; el[F.t] = i % F.nt; el[F.u] = i / F.nt; el[F.d] = F.side ? S.ax[F.d].nel - 1 : 0;
  const AxisDev &At = S.ax[F.t], &Au = S.ax[F.u], &Ad = S.ax[F.d];
  const int n0 = At.nen, n1 = Au.nen, q0n = At.nqp, q1n = Au.nqp;
  const double *t0 = At.tab + (size_t)el[F.t] * q0n * n0 * NDER, *t1 = Au.tab + (size_t)el[F.u] * q1n * n1 * NDER;
  const double *w0 = At.w + (size_t)el[F.t] * q0n, *w1 = Au.w + (size_t)el[F.u] * q1n;
  const bool rat = S.rational != 0;
  int off[3]; for (int a = 0; a < 3; ++a) off[a] = S.ax[a].off[el[a]];
  const int ld = off[F.d] + (F.side ? Ad.nen - 1 : 0);      // the face layer of the control points
  auto node = [&](int a0, int a1) { int g[3]; g[F.d] = ld; g[F.t] = off[F.t] + a0; g[F.u] = off[F.u] + a1; return (size_t)g[0] + (size_t)S.ax[0].gwidth * ((size_t)g[1] + (size_t)S.ax[1].gwidth * (size_t)g[2]); };
  double dS = 0;
  for (int q1 = 0; q1 < q1n; ++q1) for (int q0 = 0; q0 < q0n; ++q0) {
    double W0 = 1, S1[2] = {0, 0};
    if (rat) {
      W0 = 0;
      for (int a1 = 0; a1 < n1; ++a1) for (int a0 = 0; a0 < n0; ++a0) {
        const double *r0 = t0 + (q0 * n0 + a0) * NDER, *r1 = t1 + (q1 * n1 + a1) * NDER;
        const double w = S.W[node(a0, a1)];
        W0 += w * r0[0] * r1[0]; S1[0] += w * r0[1] * r1[0]; S1[1] += w * r0[0] * r1[1];
      }
    }
    double G[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int a1 = 0; a1 < n1; ++a1) for (int a0 = 0; a0 < n0; ++a0) {
      const double *r0 = t0 + (q0 * n0 + a0) * NDER, *r1 = t1 + (q1 * n1 + a1) * NDER;
      const size_t g = node(a0, a1);
      double N0 = r0[0] * r1[0], N1[2] = {r0[1] * r1[0], r0[0] * r1[1]};
      if (rat) { const double w = S.W[g]; N0 = w * N0 / W0; for (int r = 0; r < 2; ++r) N1[r] = (w * N1[r] - N0 * S1[r]) / W0; }
      for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) G[r][c] += N1[r] * S.X[g * 3 + c];
    }
    double M[2][2] = {{0, 0}, {0, 0}};
    for (int r = 0; r < 2; ++r) for (int c2 = 0; c2 < 2; ++c2) for (int c = 0; c < 3; ++c) M[r][c2] += G[r][c] * G[c2][c];
    dS += sqrt(fabs(M[0][0] * M[1][1] - M[0][1] * M[1][0])) * w0[q0] * w1[q1];
  }
  F.area[i] = dS * (At.J[el[F.t]] / (double)n0) * (Au.J[el[F.u]] / (double)n1);
}
static __global__ void k_boundary_loads_mapped(SpaceDev S, FluxArgs F, FaceAreaArgs A, int nr0, int nr1, double *vec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.nt * F.nu) return;
  int r[3]; r[F.d] = F.rd; r[F.t] = i % F.nt; r[F.u] = i / F.nt;
  for (int a = 0; a < 3; ++a) {
    const int gi = F.gfirst[a] + r[a];
    if ((F.fixlo[a] && gi == 0) || (F.fixhi[a] && gi == F.glast[a])) return;
  }
  const AxisDev &At = S.ax[F.t], &Au = S.ax[F.u];
  double sum = 0;
  for (int eu = 0; eu < A.nu; ++eu) {
    const int ou = Au.off[eu];
    if (r[F.u] < ou || r[F.u] >= ou + Au.nen) continue;
    for (int et = 0; et < A.nt; ++et) {
      const int ot = At.off[et];
      if (r[F.t] < ot || r[F.t] >= ot + At.nen) continue;
      sum += A.area[eu * A.nt + et];
    }
  }
  const double v = F.value * sum;      // (F.value = the load itself here)
  if (v != 0.0) vec[(size_t)r[0] + (size_t)nr0 * ((size_t)r[1] + (size_t)nr1 * (size_t)r[2])] += v;
}

// dof = 1, dim = 3, no geometry, no axis wrapped inside the rank on the loaded faces' axes (rows = ghosted nodes)
static bool boundary_loads_supported(const Space &s) {
  for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) {
    if (!s.load[d][sd].count) continue;
    for (int a = 0; a < 3; ++a) if (s.lay[a].alias) return false;
  }
  return true;
}

static int launch_boundary_loads(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, std::string &err) {
  for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) {
    const BC &bl = s.load[d][sd];
    if (!bl.count || s.axis[d].periodic) continue;
    // the rank holds elements on this face?
    if (sd == 0 ? s.elem_start[d] != 0 : s.elem_start[d] + s.elem_width[d] != s.elem_sizes[d]) continue;
    double load = 0; bool any = false;
    for (int k = 0; k < bl.count; ++k) if (bl.field[k] == 0) { load += bl.value[k]; any = true; }     // (dof = 1: field 0)
    if (!any) continue;
    const int t = (d + 1) % 3, u = (d + 2) % 3;
    FluxArgs F; F.d = d; F.t = t; F.u = u; F.value = load * 4.0;
    F.rd = sd == 0 ? 0 : s.axis[d].nnp - 1 - s.lay[d].gstart;
    std::vector<double> sum[2];
    const int ax2[2] = {t, u};
    for (int k = 0; k < 2; ++k) {
      const int a = ax2[k]; const AxisLayout &L = s.lay[a]; const Basis1D &b = s.basis[a];
      sum[k].assign((size_t)L.nrow, 0.0);
      for (int e = 0; e < s.elem_width[a]; ++e) {
        const int ge = s.elem_start[a] + e, first = b.offset[ge] - L.gstart;   // ghost-local index of the element's first basis function
        for (int j = 0; j < b.nen; ++j) { const int i = first + j; if (i >= 0 && i < L.nrow) sum[k][(size_t)i] += b.detJac[ge] / (double)b.nen; }
      }
    }
    F.nt = s.lay[t].nrow; F.nu = s.lay[u].nrow;
    for (int a = 0; a < 3; ++a) {
      F.gfirst[a] = s.lay[a].gstart; F.glast[a] = s.axis[a].nnp - 1;
      auto holds = [&](const BC &bv) { for (int k = 0; k < bv.count; ++k) if (bv.field[k] == 0) return 1; return 0; };
      F.fixlo[a] = s.axis[a].periodic ? 0 : holds(s.value[a][0]); F.fixhi[a] = s.axis[a].periodic ? 0 : holds(s.value[a][1]);
    }
    if (s.nsd) {      // mapped geometry: face areas per element on the device, then the per-node sums
      FaceAreaArgs A; A.d = d; A.t = t; A.u = u; A.side = sd; A.nt = s.elem_width[t]; A.nu = s.elem_width[u];
      const int ne = A.nt * A.nu, nn = F.nt * F.nu;
      if (pool_alloc(reinterpret_cast<void **>(&A.area), (size_t)ne * sizeof(double), stream) != hipSuccess) { err = "device allocation of the face areas failed"; return IGX_ERR_MEM; }
      F.value = load; F.st = F.su = nullptr;
      hipLaunchKernelGGL(k_face_areas, dim3((unsigned)((ne + 63) / 64)), dim3(64), 0, stream, S, A);
      hipLaunchKernelGGL(k_boundary_loads_mapped, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, stream, S, F, A, s.lay[0].nrow, s.lay[1].nrow, out.vec);
      (void)hipFreeAsync(A.area, stream);
      continue;
    }
    // the two short tables live for this launch only: stream-ordered allocation, copy, kernel, free
    double *dt = nullptr;
    const size_t bytes = (sum[0].size() + sum[1].size()) * sizeof(double);
    if (pool_alloc(reinterpret_cast<void **>(&dt), bytes, stream) != hipSuccess) { err = "device allocation of the boundary-load sums failed"; return IGX_ERR_MEM; }
    (void)hipMemcpyAsync(dt, sum[0].data(), sum[0].size() * sizeof(double), hipMemcpyHostToDevice, stream);
    (void)hipMemcpyAsync(dt + sum[0].size(), sum[1].data(), sum[1].size() * sizeof(double), hipMemcpyHostToDevice, stream);
    (void)hipStreamSynchronize(stream);      // (pageable host memory: the vectors go out of scope below; a face with loads is rare and small)
    F.st = dt; F.su = dt + sum[0].size();
    const int n = F.nt * F.nu;
    hipLaunchKernelGGL(k_boundary_loads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, F, s.lay[0].nrow, s.lay[1].nrow, out.vec);
    (void)hipFreeAsync(dt, stream);
  }
  return 0;
}


// zero_matrix: MatZeroEntries of the caller; called before the first launch unless the axis-0 walk stores first touches
// slab_done (may be empty): called between the two passes of an assembly that forms the elements next to the upper face of axis
// 2 first -- the ghost rows of that face are complete then and their exchange can run under the rest of the launches
// gram_patch.hpp (round 6): the p = 2 walk of patches of pencils
static void launch_patches_p2(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, int &launches, double forcing, bool first_touch);
static void launch_state_patches_p2(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, int &launches, bool first_touch, const void *kernel, const ParamsDev &prm);
// one rank, no axis wrapped inside it, one new node per element on axes 1 and 2 (a patch's nodes are consecutive)
static bool patch_walk_covers(const Space &s) {
  if (s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] != 1) return false;
  for (int d = 0; d < 3; ++d) if (s.lay[d].alias) return false;
  for (int d = 1; d < 3; ++d)
    for (int e = 0; e + 1 < s.elem_width[d]; ++e) if (s.basis[d].offset[s.elem_start[d] + e + 1] != s.basis[d].offset[s.elem_start[d] + e] + 1) return false;
  return true;
}

static int try_gram_mfma(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, bool forced,
                         std::string &kname, int &launches, std::string &err, bool &done, DomInfo &dom, const std::function<void()> &zero_matrix,
                         const std::function<void()> &slab_done = std::function<void()>(), const PencilModule *mod = nullptr,
                         const std::function<void(int)> &face_done = std::function<void(int)>()) {
  done = false;
  auto no = [&](const char *why) { if (forced) { err = std::string("MFMA kernel does not cover this configuration: ") + why; return (int)IGX_ERR_SUP; } return 0; };
  if (!mod && s.form != IGX_FORM_POISSON && s.form != IGX_FORM_POISSON_F) return no("form is not a scalar gradient-Gram form");
  const bool state = mod && mod->state;       // a Tangent (state_pencil): Jacobian / IJacobian drivers, no geometry
  if (state ? (out.op != OP_JACOBIAN && out.op != OP_IJACOBIAN) : (out.op != OP_SYSTEM && out.op != OP_MATRIX)) return no("only System / Matrix drivers (a Tangent: Jacobian / IJacobian)");
  if (s.dim != 3 || s.dof != 1) return no("needs dim=3, dof=1");
  const bool geo = s.nsd != 0;
  if (geo && s.nsd != 3) return no("mapped geometry of another dimension");
  if (state && geo != mod->state_geo) return no("a Tangent on a mapped geometry takes the walk at p = 2 only");
  for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) if (s.visit[a][sd]) return no("boundary-form passes");
  const int deg = s.axis[0].p;
  if (deg != 2 && deg != 3) return no("needs p=2 or p=3");
  for (int d = 0; d < 3; ++d) {
    if (s.axis[d].p != deg || s.basis[d].nqp != deg + 1) return no("needs the same degree p and p+1 Gauss points on every axis");
    for (int sd = 0; sd < 2; ++sd) if (s.load[d][sd].count && out.op == OP_SYSTEM && !boundary_loads_supported(s)) return no("boundary loads next to an axis wrapped inside the rank");
  }
  GramArgs ga; ga.forcing = (s.form == IGX_FORM_POISSON) ? 1.0 : -6.0; ga.nwaves = 0;
  launches = 0;
  const bool sys = out.op == OP_SYSTEM;
  // walk axis: the slowest-varying mesh axis that qualifies keeps axis 0 (contiguous CSR columns) on the lanes
  int walk_axis = -1;
  { const int pref[3] = {s.env.walk_axis, 2, 1};
    for (int k = 0; k < 3 && walk_axis < 0; ++k) if (pref[k] >= 0 && pref[k] < 3 && axis_walkable(s, pref[k], pref[k] == 0)) walk_axis = pref[k]; }
  const bool walk = walk_axis >= 0;
  if (deg == 2 && walk_axis != 0) return no("p=2 needs a walkable axis 0");
  if ((geo || mod) && walk_axis != 0) return no("a mapped geometry / a run-time form needs a walkable axis 0");
  const bool fixt = S.fixtable != nullptr && sys;       // IGASetFixTable: Dirichlet values per node (the Matrix driver applies none)
  if (fixt && walk_axis != 0) return no("fix table without a walkable axis 0");
  if (fixt && mod) return no("fix table with a run-time form");
  Box all; for (int d = 0; d < 3; ++d) { all.lo[d] = 0; all.hi[d] = s.elem_width[d]; }
  if (!walk && deg != 3) return no("p=2 needs a walkable axis 0");
  const bool first_touch = walk_axis == 0 && !s.env.no_first_touch && out.val && axis_first_touch_ok(s, 1) && axis_first_touch_ok(s, 2);
  if (!first_touch) zero_matrix();
  else if (s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] > 1) zero_neighbour_rows(s, out, stream);   // the only entries no local element reaches
  // boundary loads first: F is zeroed and only added to, so the order is free -- and the ghost rows of the upper face of axis 2
  // must be complete when the first pass of a multi-rank assembly ends (slab_done below)
  if (sys) { if (int rc = launch_boundary_loads(s, S, out, stream, err)) return rc; }
  // (round 6: IGX_PATCH=1) p = 2 on the identity geometry, one rank: the walk of 4 x 3 patches of pencils whose band rows leave through the whole
  // workgroup (gram_patch.hpp): 4 colours, 190 instead of 405 entries per element read-add-written
  if (deg == 2 && walk_axis == 0 && s.env.patch && !geo && !mod && !fixt && (out.op == OP_MATRIX || out.op == OP_SYSTEM) && patch_walk_covers(s)) {
    if (dom.ev0) (void)hipEventRecord(dom.ev0, stream);
    launch_patches_p2(s, S, out, stream, launches, ga.forcing, first_touch);
    if (dom.ev1) (void)hipEventRecord(dom.ev1, stream);
    if (pencil_launch_error()) { err = pencil_launch_error(); pencil_launch_error() = nullptr; (void)hipGetLastError(); return IGX_ERR_LIB; }
    if (hipGetLastError() != hipSuccess) { err = "gram_patch kernel launch failed"; return IGX_ERR_LIB; }
    dom.name = "gram_patch<p=2>"; dom.launches = launches; dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2]; dom.flop_per_element = 2048.0 * 21 * 3;
    kname = "gram_patch(mfma_f64_16x16x4,p=2,walk=0,packed tiles,4x3 pencils per workgroup,one window)";
    done = true;
    return 0;
  }
  // ... and of a Tangent (IGX_PATCH_STATE=1): 4 x 2 pencils per workgroup, state_patch_p2
  if (deg == 2 && walk_axis == 0 && s.env.patch_state && state && !geo && mod->patch_kfn && patch_walk_covers(s)) {
    if (dom.ev0) (void)hipEventRecord(dom.ev0, stream);
    launch_state_patches_p2(s, S, out, stream, launches, first_touch, mod->patch_kfn, mod->prm);
    if (dom.ev1) (void)hipEventRecord(dom.ev1, stream);
    if (pencil_launch_error()) { err = pencil_launch_error(); pencil_launch_error() = nullptr; (void)hipGetLastError(); return IGX_ERR_LIB; }
    if (hipGetLastError() != hipSuccess) { err = "state_patch kernel launch failed"; return IGX_ERR_LIB; }
    dom.name = "state_patch<p=2>"; dom.launches = launches; dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2]; dom.flop_per_element = mod->flop_per_element;
    kname = "state_patch<" + mod->name + ">(mfma_f64_16x16x4,p=2,walk=0,packed tiles,4x2 pencils per workgroup,one window)";
    done = true;
    return 0;
  }
  if (!walk) {
    if (dom.ev0) (void)hipEventRecord(dom.ev0, stream);
    if (sys) launch_elements<true>(s, S, out, stream, all, ga, launches); else launch_elements<false>(s, S, out, stream, all, ga, launches);
    if (dom.ev1) (void)hipEventRecord(dom.ev1, stream);
    dom.name = "gram_p3_element"; dom.launches = launches; dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2]; dom.flop_per_element = 2048.0 * 48 * 16;
    kname = "gram_p3_element(mfma_f64_16x16x4)";
  } else {
    // P = elements without a Dirichlet face (pencil kernel), E = the rest (element kernel, which owns the BC logic)
    Box P = all;
    // (the axis-0 walk applies the Dirichlet fix-up itself; the other walks leave face elements to gram_p3_element)
    if (sys && walk_axis != 0) for (int d = 0; d < 3; ++d) {
      if (s.axis[d].periodic) continue;
      if (s.value[d][0].count && s.elem_start[d] == 0) P.lo[d] = 1;
      if (s.value[d][1].count && s.elem_start[d] + s.elem_width[d] == s.elem_sizes[d]) P.hi[d] = s.elem_width[d] - 1;
    }
    const int l0 = launches;
    if (dom.ev0) (void)hipEventRecord(dom.ev0, stream);
    auto run = [&](const Box &P, const int *fty, const PencilPass *pp = nullptr) {
    if (geo && fixt) {
      switch ((deg == 2 ? 0 : 2) + (s.rational ? 1 : 0)) {
      case 0: launch_pencils<true, 0, 2, true, false, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); break;
      case 1: launch_pencils<true, 0, 2, true, true, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); break;
      case 2: launch_pencils<true, 0, 3, true, false, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); break;
      default: launch_pencils<true, 0, 3, true, true, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); break;
      }
    } else if (geo || mod) {   // metric tensor per Gauss point from the wavefront's own geometry evaluation (a run-time form: its own coefficients)
      const int v = (deg == 2 ? 0 : 4) + (sys ? 2 : 0) + (s.rational ? 1 : 0);
      switch (v) {
      case 0: launch_pencils<false, 0, 2, true, false>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      case 1: launch_pencils<false, 0, 2, true, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      case 2: launch_pencils<true, 0, 2, true, false>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      case 3: launch_pencils<true, 0, 2, true, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      case 4: launch_pencils<false, 0, 3, true, false>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      case 5: launch_pencils<false, 0, 3, true, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      case 6: launch_pencils<true, 0, 3, true, false>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      default: launch_pencils<true, 0, 3, true, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, mod, pp); break;
      }
    } else if (fixt) {
      if (deg == 2) launch_pencils<true, 0, 2, false, false, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp);
      else launch_pencils<true, 0, 3, false, false, true>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp);
    } else if (deg == 2) {
      if (sys) launch_pencils<true, 0, 2>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); else launch_pencils<false, 0, 2>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp);
    } else switch (walk_axis * 2 + (sys ? 1 : 0)) {
    case 0: launch_pencils<false, 0, 3>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); break;
    case 1: launch_pencils<true, 0, 3>(s, S, out, stream, P, ga.forcing, launches, first_touch, fty, nullptr, pp); break;
    case 2: launch_pencils<false, 1, 3>(s, S, out, stream, P, ga.forcing, launches); break;
    case 3: launch_pencils<true, 1, 3>(s, S, out, stream, P, ga.forcing, launches); break;
    case 4: launch_pencils<false, 2, 3>(s, S, out, stream, P, ga.forcing, launches); break;
    default: launch_pencils<true, 2, 3>(s, S, out, stream, P, ga.forcing, launches); break;
    }
    };
    // Upper faces first (multi-rank, axis-0 walk): the upper half of axis 2 in every colour, a mark for the exchange (slab_done); of
    // what is left, the upper half of axis 1, a mark (face_done(1)); then the upper half of every remaining pencil -- the upper face
    // of axis 0: segments of their own, the first of which re-computes p halo elements and owns the rows from its first layer on --
    // and a mark (face_done(0)); then the rest, whose last segment owns no row past its elements.  The ghost rows of a face get nothing from the passes after its mark: its messages travel under them (comm.hpp).
    // First touch makes the launch order part of the result: per pass the rule applies to the elements of that pass alone, and an
    // entry an earlier pass reaches as well is only added to (fty / ftx: {lo, hi, blocked}).
    const int n0 = s.elem_width[0], n1 = s.elem_width[1], n2 = s.elem_width[2];
    auto upper = [&](int d) { return s.proc_sizes[d] > 1 && (s.proc_ranks[d] < s.proc_sizes[d] - 1 || s.axis[d].periodic); };
    const bool whole = walk_axis == 0 && P.lo[0] == 0 && P.hi[0] == n0 && P.lo[1] == 0 && P.hi[1] == n1 && P.lo[2] == 0 && P.hi[2] == n2;
    const bool can2 = slab_done && whole && upper(2) && n2 >= 2 * (deg + 1);
    const bool can1 = face_done && whole && upper(1) && n1 >= 2 * (deg + 1) && !s.lay[1].alias;
    const bool can0 = face_done && whole && upper(0) && n0 >= 16 && !s.lay[0].alias;
    bool faces_first = can2 || can1 || can0;
    const int c2 = face_cut(n2, deg), c1 = face_cut(n1, deg), c0 = std::max(8, face_cut(n0, deg));      // (thick passes: pencil_common.hpp)
    if (faces_first && s.env.overlap < 0) {
      // IGX_OVERLAP unset: make the passes only when they cost less than what they hide.  Cost: the launcher's own measure (rounds x
      // segment length, summed over the launches) of the passes against one pass, times the time of an element-step.  Gain: the
      // largest face message at the link's rate -- the faces travel on different links, at once (IGX_LINK_GBS, default 60 GB/s per
      // direction).  At 8 ranks of the metric configuration (128^3 per rank) three passes cost 6 ms of a 33 ms assembly and a face
      // is 141 MB = 2.4 ms: one pass; at 2 ranks (256 x 256 x 128) the pass costs 2.3 ms and the face is 552 MB = 9 ms: faces first
      // (scripts/time_rank_box.py, profiles/r04_face_passes.txt).
      const int wpb = pencil_wpb(s, deg, geo, fixt, mod != nullptr);      // (the workgroup count the launches will really have)
      const size_t xl = (mod ? mod->extra_lds : 0) + pencil_win_extra(s, deg, geo, fixt, mod, wpb);      // (and their LDS: the window of the packed walks)
      Box R = P; long long multi = 0;
      if (can2) { Box A = R; A.lo[2] = c2; R.hi[2] = c2; multi += pencil_box_cost(s, deg, geo || mod, xl, A, false, wpb); }
      if (can1) { Box B = R; B.lo[1] = c1; R.hi[1] = c1; multi += pencil_box_cost(s, deg, geo || mod, xl, B, false, wpb); }
      if (can0) { Box C = R; C.lo[0] = c0; R.hi[0] = c0; multi += pencil_box_cost(s, deg, geo || mod, xl, C, true, wpb); }
      multi += pencil_box_cost(s, deg, geo || mod, xl, R, false, wpb);
      const long long single = pencil_box_cost(s, deg, geo || mod, xl, P, false, wpb);
      const double t_step = (deg == 3 ? 30e-6 : 9e-6) * ((geo || mod) ? 1.4 : 1.0);      // s per element-step of a launch (256^3: 16 ms / (4 rounds x 131))
      const double cost_s = (double)(multi - single) * t_step;
      // (the communicator's figure: $IGX_LINK_GBS, else what IGXCommInitRCCL measured on this rank's own links, else 60)
      const char *lr = getenv("IGX_LINK_GBS"); const double rate = (lr && atof(lr) > 0 ? atof(lr) : (s.link_gbs > 0 ? s.link_gbs : 60.0)) * 1e9;
      double face = 0; const double rowb = 8.0 * (2 * deg + 1) * (2 * deg + 1) * (2 * deg + 1);
      const double nr[3] = {(double)s.lay[0].nrow, (double)s.lay[1].nrow, (double)s.lay[2].nrow};
      if (can2) face = std::max(face, deg * nr[0] * nr[1] * rowb);
      if (can1) face = std::max(face, deg * nr[0] * nr[2] * rowb);
      if (can0) face = std::max(face, deg * nr[1] * nr[2] * rowb);
      faces_first = face / rate > cost_s;
    }
    dom.passes = faces_first ? 1 + (can2 ? 1 : 0) + (can1 ? 1 : 0) + (can0 ? 1 : 0) : 1;
    if (faces_first) {
      Box R = P;
      auto pass_of = [&](const Box &b, bool face0, bool rest0, int *fty, PencilPass &pp) {
        fty[0] = b.lo[2]; fty[1] = b.hi[2]; fty[2] = (can2 && b.hi[2] <= c2) ? c2 : 0x7fffffff;
        pp.ftx[0] = b.lo[1]; pp.ftx[1] = b.hi[1]; pp.ftx[2] = (can1 && b.hi[1] <= c1) ? c1 : 0x7fffffff;
        pp.halo_lo = face0 ? 0 : -1; pp.open_hi = rest0;
      };
      if (can2) { Box A = R; A.lo[2] = c2; R.hi[2] = c2; int fty[3]; PencilPass pp; pass_of(A, false, false, fty, pp); run(A, fty, &pp); slab_done(); }
      if (can1) { Box B = R; B.lo[1] = c1; R.hi[1] = c1; int fty[3]; PencilPass pp; pass_of(B, false, false, fty, pp); run(B, fty, &pp); face_done(1); }
      if (can0) { Box C = R; C.lo[0] = c0; R.hi[0] = c0; int fty[3]; PencilPass pp; pass_of(C, true, false, fty, pp); run(C, fty, &pp); face_done(0); }
      { int fty[3]; PencilPass pp; pass_of(R, false, can0, fty, pp); run(R, fty, &pp); }
    } else run(P, nullptr);
    if (dom.ev1) (void)hipEventRecord(dom.ev1, stream);
    dom.name = std::string(state ? "state_pencil<walk=" : mod ? "form_pencil<hiprtc,walk=" : "gram_pencil<walk=") + char('0' + walk_axis) + ",p=" + char('0' + deg) + (geo ? ",geometry" : "") + ">"; dom.launches = launches - l0;
    dom.elements = (long long)std::max(0, P.hi[0] - P.lo[0]) * std::max(0, P.hi[1] - P.lo[1]) * std::max(0, P.hi[2] - P.lo[2]);
    // executed MFMA flops per element: 2*16*16*4 per v_mfma_f64_16x16x4, 48 k-steps, 10 (symmetric, walk 0) or 16 tiles
    // (p = 2: 21 k-steps x 6 layer-pair tiles, or x 3 packed tiles)
    dom.flop_per_element = 2048.0 * (deg == 3 ? 48 * (walk_axis == 0 ? 10 : 16) : 21 * ((walk_axis == 0 && pencil_p2_pack(s, deg, geo, fixt, mod != nullptr)) ? 3 : 6));
    if (state) dom.flop_per_element = mod->flop_per_element;
    // E as disjoint slabs: axis 0 faces (full), axis 1 faces (inside P along 0), axis 2 faces (inside P along 0,1)
    for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
      Box b = all;
      for (int k = 0; k < d; ++k) { b.lo[k] = P.lo[k]; b.hi[k] = P.hi[k]; }
      if (side == 0) { b.lo[d] = all.lo[d]; b.hi[d] = std::min(P.lo[d], P.hi[d]); }
      else { b.lo[d] = std::max(P.hi[d], P.lo[d]); b.hi[d] = all.hi[d]; }
      if (sys) launch_elements<true>(s, S, out, stream, b, ga, launches); else launch_elements<false>(s, S, out, stream, b, ga, launches);
    }
    kname = (state ? std::string("state_pencil<") + mod->name + ">(mfma_f64_16x16x4,p=" : mod ? std::string("form_pencil<") + mod->name + ">(hiprtc,mfma_f64_16x16x4,p=" : std::string("gram_pencil(mfma_f64_16x16x4,p=")) + char('0' + deg) + ",walk=" + char('0' + walk_axis) + (geo ? ",mapped geometry" : "") + ((deg == 2 && walk_axis == 0 && (pencil_p2_pack(s, deg, geo, fixt, mod != nullptr) || (mod && mod->pack))) ? ",packed tiles" : "") + (walk_axis == 0 ? ")" : ")+gram_p3_element(faces)");
  }
  if (pencil_launch_error()) { err = pencil_launch_error(); pencil_launch_error() = nullptr; (void)hipGetLastError(); return IGX_ERR_LIB; }
  if (hipGetLastError() != hipSuccess) { err = "gram MFMA kernel launch failed"; return IGX_ERR_LIB; }
  done = true;
  return 0;
}

#endif   // !IGX_RTC

}  // namespace igx
