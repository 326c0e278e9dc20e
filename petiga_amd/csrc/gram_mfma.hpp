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
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "igx.hpp"

namespace igx {

typedef double d4_t __attribute__((ext_vector_type(4)));

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
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j1 = off[1] + r;
    const int c1 = S.ax[1].rcnt[S.ax[1].rowmap[j1]];
    const int P1 = S.ax[1].P[j1 * 7 + (b2 - r + 3)];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
      const int k1 = off[2] + ta;
      const size_t base = (size_t)out.browptr[rowof(a1, r, ta)];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const int P2 = S.ax[2].P[k1 * 7 + (tb - ta + 3)];
        double *dst = out.val + base + ((size_t)P2 * c1 + P1) * c0 + P0;
        *dst += acc[ta][tb][r];
      }
    }
  }
}

static int try_gram_mfma(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, bool forced,
                         std::string &kname, int &launches, std::string &err, bool &done) {
  done = false;
  auto no = [&](const char *why) { if (forced) { err = std::string("MFMA kernel does not cover this configuration: ") + why; return (int)IGX_ERR_SUP; } return 0; };
  if (s.form != IGX_FORM_POISSON && s.form != IGX_FORM_POISSON_F) return no("form is not a scalar gradient-Gram form");
  if (out.op != OP_SYSTEM && out.op != OP_MATRIX) return no("only System / Matrix drivers");
  if (s.dim != 3 || s.dof != 1) return no("needs dim=3, dof=1");
  if (s.nsd) return no("mapped geometry");
  if (S.fixtable) return no("fix table");
  for (int d = 0; d < 3; ++d) {
    if (s.axis[d].p != 3 || s.basis[d].nqp != 4) return no("needs p=3 and 4 Gauss points per axis");
    for (int sd = 0; sd < 2; ++sd) if (s.load[d][sd].count) return no("boundary loads");
  }
  GramArgs ga; ga.forcing = (s.form == IGX_FORM_POISSON) ? 1.0 : -6.0;
  launches = 0;
  const int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
      int first = -1, count = 0;
      for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
      if (!count) { empty = true; break; }
      cr.start[d] = first; cr.step[d] = L.p + 1; cr.count[d] = count;
    }
    if (empty) continue;
    ga.nwaves = cr.count[0] * cr.count[1] * cr.count[2];
    const unsigned nblocks = (unsigned)((ga.nwaves + 3) / 4);
    if (out.op == OP_SYSTEM) hipLaunchKernelGGL(gram_p3_element<true>, dim3(nblocks), dim3(256), 0, stream, S, out, cr, ga);
    else hipLaunchKernelGGL(gram_p3_element<false>, dim3(nblocks), dim3(256), 0, stream, S, out, cr, ga);
    launches++;
  }
  if (hipGetLastError() != hipSuccess) { err = "gram_p3_element launch failed"; return IGX_ERR_LIB; }
  kname = "gram_p3_element(mfma_f64_16x16x4)";
  done = true;
  return 0;
}

}  // namespace igx
