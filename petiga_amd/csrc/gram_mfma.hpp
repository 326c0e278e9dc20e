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
#include <cstdlib>
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
// Pencil kernel: one wavefront walks a pencil of elements along axis 2 and lets the MFMA accumulators
// combine the contributions of consecutive elements before anything is written.
//
// The accumulator tile index is the axis-2 basis index, so a tile (ta,tb) belongs to a pair of axis-2
// node layers (i3,j3).  Layers are bound to tile slots by  slot = layer & 3 ; an element covers four
// consecutive layers, i.e. all four slots.  When the walk leaves layer l (after element off2 = l), the
// 7 tiles with row- or column-slot l&3 are complete for this pencil: they are added to the CSR
// (row layer l x col layers l..l+3, and row layers l+1..l+3 x col layer l) and zeroed, then the slot is
// reused by layer l+4.  Per element 7*256 = 1792 entries reach memory instead of 4096, and two pencils
// conflict only if they share axis-0/axis-1 nodes: 16 colours instead of 64.
// The loads of the read-modify-write are issued before the element's 768 MFMAs and consumed after them.
//
// Long pencils are cut into segments.  A segment owns the row layers of its own elements; it first
// re-computes (without writing) the <=3 elements before its start so that its first rows are complete,
// and discards rows it does not own.  No axis-2 colouring is needed.
// Elements that touch a Dirichlet face are not walked here: they go to gram_p3_element (BC logic there).
// ======================================================================================================
struct PencilArgs {
  double forcing;
  int e0_start, e0_step, e0_count;
  int e1_start, e1_step, e1_count;
  int z_lo, z_hi;          // walked range of local elements along axis 2: [z_lo, z_hi)
  int seg_len, nseg;
  int blocks_per_seg;      // ceil(pencils / 4): a workgroup (4 wavefronts = 4 pencils) never straddles segments
  int ne_max;              // LDS capacity: elements (seg_len + 3 halo) ; layers = ne_max + 3
  int debug_noflush;       // experiment switch: 1 = skip the read-modify-write (timing of the MFMA walk alone)
};

struct PencilLane {        // per-lane constants of a pencil
  double u0, u1, wq1, v0[4], v1[4];
  long long A[4];          // prefix1[rho1(r)]*tot0 + cnt1(r)*prefix0[rho0]
  int B[4], C[4];          // P1(r)*cnt0 + P0 ; cnt1(r)*cnt0
  double s01;              // forcing * sum_q w0 N0[a1] * sum_q w1 N1[a2] for the F lane
  long long frow01;        // rowmap0 + nrow0*rowmap1 of the F lane
  int fslot;
};

// LDS-staged axis-2 data of one segment (the "knot-span tables" of the walk): per element the 1-D basis
// rows (value, derivative), Gauss weights and half-length; per node layer the row prefix, row length and
// the 7 column positions of the structured CSR.
struct PencilLds {
  double *zt;        // [ne][4 q][4 a][2]
  double *wq;        // [ne][4]
  double *Jz;        // [ne]
  long long *pre;    // [nl]  prefix2[rho2] * tot1 * tot0
  int *cnt;          // [nl]  rcnt2[rho2]  (-1: layer does not exist)
  int *rho;          // [nl]
  int *P;            // [nl][8]
  int lay0;          // first layer held
};

__device__ __forceinline__ PencilLds pencil_lds_carve(double *sm, int ne_max) {
  PencilLds t; const int nl = ne_max + 3;
  t.zt = sm; t.wq = t.zt + ne_max * 32; t.Jz = t.wq + ne_max * 4;
  t.pre = reinterpret_cast<long long *>(t.Jz + ((ne_max + 1) & ~1));
  t.cnt = reinterpret_cast<int *>(t.pre + nl); t.rho = t.cnt + nl; t.P = t.rho + nl; t.lay0 = 0;
  return t;
}
static inline size_t pencil_lds_bytes(int ne_max) {
  const int nl = ne_max + 3;
  return (size_t)(ne_max * 32 + ne_max * 4 + ((ne_max + 1) & ~1)) * 8 + (size_t)nl * 8 + (size_t)nl * 4 * 10 + 64;
}

__device__ __forceinline__ void pencil_mfma(d4_t (&acc)[4][4], const PencilLane &L, const double *zt /*LDS [4][4][2]*/,
                                            const double *__restrict__ Wq1, const double *wq2 /*LDS*/, double Jel) {
#pragma unroll
  for (int q3 = 0; q3 < 4; ++q3) {
    double z0[4], z1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { z0[t] = zt[(q3 * 4 + t) * 2 + 0]; z1[t] = zt[(q3 * 4 + t) * 2 + 1]; }
    const double s3 = Jel * wq2[q3];
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      const double jw = L.wq1 * (Wq1[q2] * s3);
#pragma unroll
      for (int al = 0; al < 3; ++al) {
        const double uv = (al == 0 ? L.u1 : L.u0) * (al == 1 ? L.v1[q2] : L.v0[q2]);
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
}

// the 7 tiles that leave with layer `lay`: k = 0..3 -> (row lay, col lay+k), k = 4..6 -> (row lay+k-3, col lay)
struct FlushPlan { bool on[7]; long long base[7]; int cnt2[7], p2[7]; };

__device__ __forceinline__ FlushPlan pencil_plan(const PencilLds &T, int nl, int lay, int own_lo, int own_hi) {
  FlushPlan f;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int rl = (k < 4) ? lay : lay + (k - 3), cl = (k < 4) ? lay + k : lay;
    const int ri = rl - T.lay0, ci = cl - T.lay0;
    bool ok = ri >= 0 && ri < nl && ci >= 0 && ci < nl && rl >= own_lo && rl < own_hi;
    int p2 = -1, c2 = 0; long long pre = 0;
    if (ok) { c2 = T.cnt[ri]; ok = c2 > 0 && T.cnt[ci] > 0; }
    if (ok) { p2 = T.P[ri * 8 + (cl - rl + 3)]; pre = T.pre[ri]; ok = p2 >= 0; }
    // everything here is wave-uniform: pin it to SGPRs (LDS reads come back in VGPRs)
    const int lo = __builtin_amdgcn_readfirstlane((int)(pre & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(pre >> 32));
    f.on[k] = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    f.base[k] = ((long long)hi << 32) | (unsigned int)lo;
    f.cnt2[k] = __builtin_amdgcn_readfirstlane(c2); f.p2[k] = __builtin_amdgcn_readfirstlane(p2);
  }
  return f;
}

// read-modify-write of tiles [K0,K1) of the plan: k = 0..3 is tile (0,k), k = 4..6 is tile (k-3,0).  The loads
// of one call are all in flight together; a second wavefront on the SIMD runs its MFMAs meanwhile.
template <int K0, int K1>
__device__ __forceinline__ void pencil_flush(const d4_t (&acc)[4][4], const PencilLane &L, const FlushPlan &f, double *__restrict__ val) {
  double old[K1 - K0][4];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    if (!f.on[k]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long pos = f.base[k] + (long long)f.cnt2[k] * L.A[r] + (long long)f.p2[k] * L.C[r] + L.B[r];
      old[k - K0][r] = val[pos];
    }
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int ta = (k < 4) ? 0 : k - 3, tb = (k < 4) ? k : 0;
    if (!f.on[k]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long pos = f.base[k] + (long long)f.cnt2[k] * L.A[r] + (long long)f.p2[k] * L.C[r] + L.B[r];
      val[pos] = old[k - K0][r] + acc[ta][tb][r];
    }
  }
}

// the window slides by one layer: tile (ta,tb) <- tile (ta+1,tb+1); the new last row / column start at zero
__device__ __forceinline__ void pencil_shift(d4_t (&acc)[4][4], double &Facc, int fslot) {
#pragma unroll
  for (int ta = 0; ta < 3; ++ta)
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) acc[ta][tb] = acc[ta + 1][tb + 1];
#pragma unroll
  for (int t = 0; t < 4; ++t) { acc[3][t] = (d4_t){0, 0, 0, 0}; acc[t][3] = (d4_t){0, 0, 0, 0}; }
  // F lanes hold (a1, a2, slot = lane>>4): slot t takes over slot t+1's partial sum
  const double up = __shfl_down(Facc, 16);
  Facc = (fslot == 3) ? 0.0 : up;
}

// leaving layer `lay` (always tile slot 0): add its 7 tiles and its F entries to the global arrays
template <bool SYSTEM>
__device__ __forceinline__ void pencil_leave(d4_t (&acc)[4][4], double &Facc, const PencilLane &L, const PencilLds &T, int nl, const OutDev &out,
                                             int lay, int own_lo, int own_hi, long long nr01) {
  if (SYSTEM && L.fslot == 0) {
    const int li = lay - T.lay0;
    if (li >= 0 && li < nl && T.cnt[li] > 0 && lay >= own_lo && lay < own_hi) out.vec[L.frow01 + nr01 * T.rho[li]] += Facc;
  }
  const FlushPlan f = pencil_plan(T, nl, lay, own_lo, own_hi);
  pencil_flush<0, 4>(acc, L, f, out.val);
  pencil_flush<4, 7>(acc, L, f, out.val);
  pencil_shift(acc, Facc, L.fslot);
}

template <bool SYSTEM>
__global__ void __launch_bounds__(256, 2)
gram_p3_pencil(SpaceDev S, OutDev out, PencilArgs pa) {
  extern __shared__ __attribute__((aligned(16))) double pencil_sm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = blockIdx.x / pa.blocks_per_seg;
  const int pencil = (blockIdx.x - seg * pa.blocks_per_seg) * 4 + wave;
  const int zs = pa.z_lo + seg * pa.seg_len;
  const int ze = min(zs + pa.seg_len, pa.z_hi);
  const int zh = max(zs - 3, pa.z_lo);
  const int ne = ze - zh, nl = ne + 3;

  // ---- stage the segment's axis-2 tables in LDS (all 256 threads)
  PencilLds T = pencil_lds_carve(pencil_sm, pa.ne_max);
  T.lay0 = S.ax[2].off[zh];
  {
    const int tid = threadIdx.x;
    const double *__restrict__ tab2 = S.ax[2].tab + (size_t)zh * (4 * 4 * NDER);
    for (int i = tid; i < ne * 32; i += 256) { const int e = i >> 5, j = i & 31; T.zt[i] = tab2[(size_t)e * 64 + (j >> 1) * NDER + (j & 1)]; }
    for (int i = tid; i < ne * 4; i += 256) T.wq[i] = S.ax[2].w[zh * 4 + i];
    for (int i = tid; i < ne; i += 256) T.Jz[i] = S.ax[2].J[zh + i];
    const long long T10 = S.ax[1].tot * S.ax[0].tot;
    for (int i = tid; i < nl; i += 256) {
      const int lay = T.lay0 + i;
      if (lay < S.ax[2].gwidth) {
        const int rho = S.ax[2].rowmap[lay];
        T.rho[i] = rho; T.cnt[i] = S.ax[2].rcnt[rho]; T.pre[i] = S.ax[2].prefix[rho] * T10;
        for (int d = 0; d < 7; ++d) T.P[i * 8 + d] = S.ax[2].P[lay * 7 + d];
      } else { T.rho[i] = 0; T.cnt[i] = -1; T.pre[i] = 0; }
    }
  }
  __syncthreads();
  if (pencil >= pa.e0_count * pa.e1_count) return;

  const int t0 = pencil % pa.e0_count, t1 = pencil / pa.e0_count;
  const int el0 = pa.e0_start + t0 * pa.e0_step, el1 = pa.e1_start + t1 * pa.e1_step;
  int own_lo = (seg == 0) ? -1 : S.ax[2].off[zs];
  int own_hi = (seg == pa.nseg - 1) ? (1 << 30) : S.ax[2].off[ze];
  if (pa.debug_noflush) { own_lo = 1 << 30; own_hi = 1 << 30; }
  const int off0 = S.ax[0].off[el0], off1 = S.ax[1].off[el1];
  const long long nr01 = (long long)S.ax[0].nrow * S.ax[1].nrow;

  PencilLane L;
  {
    const double *__restrict__ T0 = S.ax[0].tab + (size_t)el0 * (4 * 4 * NDER);
    const double *__restrict__ T1 = S.ax[1].tab + (size_t)el1 * (4 * 4 * NDER);
    const double *__restrict__ Wq0 = S.ax[0].w + el0 * 4;
    const double *__restrict__ Wq1 = S.ax[1].w + el1 * 4;
    const int q1 = lane >> 4, i1 = lane & 3, i2 = (lane >> 2) & 3;
    L.u0 = T0[(q1 * 4 + i1) * NDER + 0]; L.u1 = T0[(q1 * 4 + i1) * NDER + 1]; L.wq1 = Wq0[q1];
#pragma unroll
    for (int q = 0; q < 4; ++q) { L.v0[q] = T1[(q * 4 + i2) * NDER + 0]; L.v1[q] = T1[(q * 4 + i2) * NDER + 1]; }
    // scatter constants: this lane's result rows are (a1 = lane>>4, a2 = r), columns (b1 = lane&3, b2 = (lane>>2)&3)
    const int a1 = lane >> 4, b1 = lane & 3, b2 = (lane >> 2) & 3;
    const int i0 = off0 + a1, rho0 = S.ax[0].rowmap[i0], c0 = S.ax[0].rcnt[rho0];
    const long long PS0 = S.ax[0].prefix[rho0];
    const int P0 = S.ax[0].P[i0 * 7 + (b1 - a1 + 3)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j1 = off1 + r, rho1 = S.ax[1].rowmap[j1], c1 = S.ax[1].rcnt[rho1];
      L.A[r] = S.ax[1].prefix[rho1] * S.ax[0].tot + (long long)c1 * PS0;
      L.B[r] = S.ax[1].P[j1 * 7 + (b2 - r + 3)] * c0 + P0;
      L.C[r] = c1 * c0;
    }
    // F lane: (fa0 = lane&3, fa1 = (lane>>2)&3, slot = lane>>4)
    const int fa0 = lane & 3, fa1 = (lane >> 2) & 3;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { s0 += Wq0[q] * T0[(q * 4 + fa0) * NDER]; s1 += Wq1[q] * T1[(q * 4 + fa1) * NDER]; }
    L.s01 = pa.forcing * (s0 * s1);
    L.frow01 = (long long)S.ax[0].rowmap[off0 + fa0] + (long long)S.ax[0].nrow * S.ax[1].rowmap[off1 + fa1];
    L.fslot = lane >> 4;
  }
  const double *__restrict__ Wq1 = S.ax[1].w + el1 * 4;
  const double J01 = S.ax[0].J[el0] * S.ax[1].J[el1];

  d4_t acc[4][4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = (d4_t){0, 0, 0, 0};
  double Facc = 0;

  int lay = T.lay0;
  for (int ei = 0; ei < ne; ++ei) {
    lay = T.lay0 + ei;          // walk condition: one new layer per element, local basis a3 sits in tile slot a3
    const double *zt = T.zt + ei * 32, *wq2 = T.wq + ei * 4;
    const double Jel = J01 * T.Jz[ei];
    pencil_mfma(acc, L, zt, Wq1, wq2, Jel);
    if (SYSTEM) {
      double s2 = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) s2 += wq2[q] * zt[(q * 4 + L.fslot) * 2];
      Facc += Jel * (L.s01 * s2);
    }
    pencil_leave<SYSTEM>(acc, Facc, L, T, nl, out, lay, own_lo, own_hi, nr01);
  }
  if (seg == pa.nseg - 1)       // the last segment also owns what is still in the window
    for (int k = 1; k <= 3; ++k) pencil_leave<SYSTEM>(acc, Facc, L, T, nl, out, lay + k, own_lo, own_hi, nr01);
}

// ------------------------------------------------------------------ dispatch
struct Box { int lo[3], hi[3]; };   // local element box [lo,hi)

// colour c of axis d restricted to [lo,hi): arithmetic sequence (regular colours) or a single element
static bool color_range(const AxisLayout &L, int c, int lo, int hi, int &start, int &step, int &count) {
  start = -1; count = 0; step = L.p + 1;
  for (int e = lo; e < hi; ++e) if (L.color[e] == c) { if (start < 0) start = e; count++; }
  return count > 0;
}

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

template <bool SYSTEM>
static void launch_pencils(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, const Box &bx, double forcing, int &launches) {
  for (int d = 0; d < 3; ++d) if (bx.hi[d] <= bx.lo[d]) return;
  const int nz = bx.hi[2] - bx.lo[2];
  for (int c1 = 0; c1 < s.lay[1].ncolors; ++c1) for (int c0 = 0; c0 < s.lay[0].ncolors; ++c0) {
    PencilArgs pa; pa.forcing = forcing;
    if (!color_range(s.lay[0], c0, bx.lo[0], bx.hi[0], pa.e0_start, pa.e0_step, pa.e0_count)) continue;
    if (!color_range(s.lay[1], c1, bx.lo[1], bx.hi[1], pa.e1_start, pa.e1_step, pa.e1_count)) continue;
    const long long pencils = (long long)pa.e0_count * pa.e1_count;
    // enough wavefronts for ~4 rounds of the 2048 resident ones (2 per SIMD), segments no shorter than 32 elements
    int nseg = (int)std::max<long long>(1, std::min<long long>((16384 + pencils - 1) / pencils, std::max(1, nz / 32)));
    pa.seg_len = (nz + nseg - 1) / nseg; pa.nseg = (nz + pa.seg_len - 1) / pa.seg_len;
    pa.z_lo = bx.lo[2]; pa.z_hi = bx.hi[2];
    pa.blocks_per_seg = (int)((pencils + 3) / 4);
    pa.ne_max = pa.seg_len + 3;
    { const char *dbg = getenv("IGX_DEBUG_NOFLUSH"); pa.debug_noflush = (dbg && dbg[0] == '1') ? 1 : 0; }
    const size_t lds = pencil_lds_bytes(pa.ne_max);
    auto kern = gram_p3_pencil<SYSTEM>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(pa.blocks_per_seg * pa.nseg)), dim3(256), lds, stream, S, out, pa);
    launches++;
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
  GramArgs ga; ga.forcing = (s.form == IGX_FORM_POISSON) ? 1.0 : -6.0; ga.nwaves = 0;
  launches = 0;
  const bool sys = out.op == OP_SYSTEM;
  // can axis 2 be walked?  (one new node layer per element, no wrap inside the rank)
  bool walk = !s.lay[2].alias && s.elem_width[2] >= 8;
  for (int e = 0; e + 1 < s.elem_width[2] && walk; ++e)
    if (s.basis[2].offset[s.elem_start[2] + e + 1] != s.basis[2].offset[s.elem_start[2] + e] + 1) walk = false;
  Box all; for (int d = 0; d < 3; ++d) { all.lo[d] = 0; all.hi[d] = s.elem_width[d]; }
  if (!walk) {
    if (sys) launch_elements<true>(s, S, out, stream, all, ga, launches); else launch_elements<false>(s, S, out, stream, all, ga, launches);
    kname = "gram_p3_element(mfma_f64_16x16x4)";
  } else {
    // P = elements without a Dirichlet face (pencil kernel), E = the rest (element kernel, which owns the BC logic)
    Box P = all;
    if (sys) for (int d = 0; d < 3; ++d) {
      if (s.axis[d].periodic) continue;
      if (s.value[d][0].count && s.elem_start[d] == 0) P.lo[d] = 1;
      if (s.value[d][1].count && s.elem_start[d] + s.elem_width[d] == s.elem_sizes[d]) P.hi[d] = s.elem_width[d] - 1;
    }
    if (sys) launch_pencils<true>(s, S, out, stream, P, ga.forcing, launches); else launch_pencils<false>(s, S, out, stream, P, ga.forcing, launches);
    // E as disjoint slabs: axis 0 faces (full), axis 1 faces (inside P along 0), axis 2 faces (inside P along 0,1)
    for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
      Box b = all;
      for (int k = 0; k < d; ++k) { b.lo[k] = P.lo[k]; b.hi[k] = P.hi[k]; }
      if (side == 0) { b.lo[d] = all.lo[d]; b.hi[d] = std::min(P.lo[d], P.hi[d]); }
      else { b.lo[d] = std::max(P.hi[d], P.lo[d]); b.hi[d] = all.hi[d]; }
      if (sys) launch_elements<true>(s, S, out, stream, b, ga, launches); else launch_elements<false>(s, S, out, stream, b, ga, launches);
    }
    kname = "gram_p3_pencil(mfma_f64_16x16x4)+gram_p3_element(faces)";
  }
  if (hipGetLastError() != hipSuccess) { err = "gram MFMA kernel launch failed"; return IGX_ERR_LIB; }
  done = true;
  return 0;
}

}  // namespace igx
