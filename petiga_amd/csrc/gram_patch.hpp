// gram_patch.hpp -- the p = 2 Gram walk with the band rows combined across ALL THREE axes before they reach memory (round 6).
//
// gram_pencil_w6 (gram_mfma.hpp) gives each of a workgroup's twelve wavefronts a pencil of its own and a window of its own: a band
// row is combined along the walk and then read-add-written once per pencil -- 405 entries per element, nine colours.  Here the
// twelve wavefronts walk a PATCH of 4 x 3 adjacent pencils in step and add into ONE window in LDS,
//   win[4 node layers (ring)][y pair][x pair][5 walk-axis offsets],
// a "pair" being two nodes of the patch on an axis that share one of its elements (24 on axis 1, 19 on axis 2).  A layer of the
// patch is complete when every wavefront has added its element: 2280 entries for 12 elements -- 190 per element instead of 405 --
// in runs of 5 x 8 bytes that lie 25 to a 200-byte stretch of a CSR row, and patches conflict only with their neighbours: 4 colours.
// Every thread of the workgroup owns ONE run (y pair, x pair) for the whole walk: its matrix address is a constant plus the
// layer's part, its old values are requested before the element's MFMAs and consumed behind them.
//
// Degree 2, identity geometry, axis-0 walk, packed tiles (pencil_mfma_p2k: 63 MFMAs per element).
#pragma once
#include "gram_mfma.hpp"

namespace igx {

constexpr int PATCH_MX = 4, PATCH_MY = 3, PATCH_W = PATCH_MX * PATCH_MY;      // pencils (= wavefronts) of a workgroup
constexpr int PATCH_NX = PATCH_MX + 2, PATCH_NY = PATCH_MY + 2;                // nodes of a patch on axes 1, 2 (p = 2)
constexpr int PATCH_NXP = 5 * PATCH_MX + 4, PATCH_NYP = 5 * PATCH_MY + 4;      // node pairs that share an element of the patch
constexpr int PATCH_LAYER = PATCH_NYP * PATCH_NXP * 5;                          // doubles per node layer of the window
constexpr int PATCH_SLOTS = 4;                                                  // ring: three layers being added to, one being flushed

struct PatchArgs {
  PencilArgs pa;                         // walk-axis range and segments, forcing, first touch (as for the pencil walk)
  int px_start, px_step, px_count;       // patches of this colour: patch indices on axis 1 ...
  int py_start, py_step, py_count;       // ... and on axis 2
};

// LDS behind the walk's tables: the window, the pair tables (ints), the F ring
__host__ __device__ static inline size_t patch_lds_bytes(int ne_max) {
  return pencil_lds_bytes(ne_max, false, PATCH_W) + (size_t)PATCH_SLOTS * PATCH_LAYER * 8 + (size_t)(PATCH_NX * 5 + PATCH_NY * 5 + PATCH_NXP + PATCH_NYP + 8) * 4 + 64;
}

template <bool SYSTEM>
__global__ void __launch_bounds__(768, 3)
gram_patch_p2(SpaceDev S, OutDev out, PatchArgs A) {
  static_assert(!SYSTEM, "step 1: the Matrix driver");
  constexpr int P = 2, NB = 3, BW = 5;
  extern __shared__ __attribute__((aligned(16))) double pencil_sm[];
  const PencilArgs &pa = A.pa;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg = blockIdx.x / pa.blocks_per_seg, patch = blockIdx.x - seg * pa.blocks_per_seg;
  const int tx = patch % A.px_count, ty = patch / A.px_count;
  const int ex0 = (A.px_start + tx * A.px_step) * PATCH_MX, ey0 = (A.py_start + ty * A.py_step) * PATCH_MY;
  const int mxv = min(PATCH_MX, pa.nelx - ex0), myv = min(PATCH_MY, pa.nely - ey0);      // elements of the patch inside the mesh
  const int wi = wave % PATCH_MX, wj = wave / PATCH_MX;
  const bool valid = wi < mxv && wj < myv;
  const int elx = ex0 + (valid ? wi : 0), ely = ey0 + (valid ? wj : 0);
  const AxisDev &AW = S.ax[0], &AX = S.ax[1], &AY = S.ax[2];
  const int ws = pa.w_lo + seg * pa.seg_len, we = min(ws + pa.seg_len, pa.w_hi);
  const int wh = max(ws - P, pa.w_halo_lo);
  const int ne = we - wh, nl = ne + P;

  PencilLds T = pencil_lds_carve(pencil_sm, pa.ne_max, false);
  T.lay0 = AW.off[wh];
  double *win = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + pencil_lds_bytes(pa.ne_max, false, PATCH_W));
  int *XP = reinterpret_cast<int *>(win + PATCH_SLOTS * PATCH_LAYER), *YP = XP + PATCH_NX * 5, *XI = YP + PATCH_NY * 5, *YI = XI + PATCH_NXP, *cntp = YI + PATCH_NYP;
  {   // the walk-axis tables of the segment (as gram_pencil_body stages them), the window, the pair tables
    const int nthr = PATCH_W * 64;
    for (int i = tid; i < ne * 32; i += nthr) {
      const int e = i >> 5, j = i & 31, q = j >> 3, aa = (j >> 1) & 3, k = j & 1, eg = wh + e;
      T.zt[i] = (q < NB && aa < NB) ? AW.tab[((size_t)eg * NB * NB + q * NB + aa) * NDER + k] * sqrt(AW.w[eg * NB + q] * AW.J[eg]) : 0.0;
    }
    for (int i = tid; i < ne * 4; i += nthr) { const int e = i >> 2, q = i & 3, eg = wh + e; T.wq[i] = (q < NB) ? sqrt(AW.w[eg * NB + q] * AW.J[eg]) : 0.0; }
    for (int i = tid; i < ne; i += nthr) T.Jz[i] = AW.J[wh + i];
    for (int i = tid; i < nl; i += nthr) {
      const int lay = T.lay0 + i;
      if (lay < AW.gwidth) {
        const int rho = AW.rowmap[lay];
        T.rho[i] = rho; T.cnt[i] = AW.rcnt[rho]; T.pre[i] = AW.prefix[rho];
        for (int d = 0; d < BW; ++d) T.P[i * 8 + d] = AW.P[lay * BW + d];
      } else { T.rho[i] = 0; T.cnt[i] = -1; T.pre[i] = 0; }
    }
    for (int i = tid; i < PATCH_SLOTS * PATCH_LAYER; i += nthr) win[i] = 0.0;
    if (tid < 2) {      // pairs (r, r + d) of an axis that share one of the patch's mv elements k: max(r, c) - 2 <= k <= min(r, c), 0 <= k < mv
      const int nn = tid == 0 ? PATCH_NX : PATCH_NY, mv = tid == 0 ? mxv : myv;
      int *PT = tid == 0 ? XP : YP, *PI = tid == 0 ? XI : YI;
      int n = 0;
      for (int r = 0; r < nn; ++r) for (int d = -2; d <= 2; ++d) {
        const int c = r + d, hi = r > c ? r : c, lo = r < c ? r : c;
        const bool ok = c >= 0 && c < nn && max(hi - 2, 0) <= min(lo, mv - 1);
        PT[r * 5 + d + 2] = ok ? n : -1;
        if (ok) PI[n++] = r | ((d + 2) << 8);
      }
      cntp[tid] = n;
    }
  }
  __syncthreads();
  const int nxp = cntp[0], nyp = cntp[1];

  // ---- this wavefront's pencil: the 1-D rows of axes 1, 2 in LDS (scaled by sqrt(w J)), the operand offsets of its lanes
  double *rows = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + (pencil_lds_bytes(pa.ne_max, false, PATCH_W) - (size_t)2 * PATCH_W * 32 * 8));
  double *vys = rows + wave * 32, *uxs = rows + PATCH_W * 32 + wave * 32;
  {
    const double *__restrict__ TX = AX.tab + (size_t)elx * (NB * NB * NDER);
    const double *__restrict__ TY = AY.tab + (size_t)ely * (NB * NB * NDER);
    if (lane < 32) { const int aa = lane >> 3, qq = (lane >> 1) & 3, kk = lane & 1; vys[lane] = (aa < NB && qq < NB) ? TY[(qq * NB + aa) * NDER + kk] * sqrt(AY.w[ely * NB + qq] * AY.J[ely]) : 0.0; }
    else { const int l2 = lane - 32, qq = l2 >> 3, aa = (l2 >> 1) & 3, kk = l2 & 1; uxs[l2] = (aa < NB && qq < NB) ? TX[(qq * NB + aa) * NDER + kk] * sqrt(AX.w[elx * NB + qq] * AX.J[elx]) : 0.0; }
  }
  const P2kLaneT<false> K = pencil_p2k_lane<false>(lane);
  // where the lane's results go in the window: (y pair, x pair, walk offset) of (row a, column b); code = offset * 4 + the ring slot's
  // layer part (the row's a_w), -1 for the padding.  Tiles (0,0), (0,1), (1,1) and the mirror entries of (0,1).
  int code[3][4], codem[4];
  {
    auto split = [](int f, int &aw, int &ay, int &ax) { aw = f / 9; const int r = f - 9 * aw; ay = r / 3; ax = r - 3 * ay; };
    auto entry = [&](int a, int b) -> int {
      if (a >= 27 || b >= 27 || !valid) return -1;
      int aw, ay, ax, bw, by, bx; split(a, aw, ay, ax); split(b, bw, by, bx);
      const int yp = YP[(wj + ay) * 5 + (by - ay + 2)], xp = XP[(wi + ax) * 5 + (bx - ax + 2)];
      return (((yp * nxp + xp) * 5 + (bw - aw + 2)) << 2) | aw;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int a0 = 4 * i + (lane >> 4), a1 = 16 + a0, b0 = lane & 15, b1 = 16 + b0;
      code[0][i] = entry(a0, b0); code[1][i] = entry(a0, b1); code[2][i] = entry(a1, b1);
      codem[i] = entry(b1, a0);
    }
  }
  // ---- this thread's run of the band rows: (y pair, x pair) for the whole walk; pos = RA + RB prefix0(layer) + RC count0(layer) + P0(layer, d)
  const bool unit = tid < nxp * nyp;
  long long RA = 0; int RB = 0, RC = 0, woff = 0;
  if (unit) {
    const int yp = tid / nxp, xp = tid - yp * nxp;
    const int yr = YI[yp] & 255, dy = (YI[yp] >> 8) - 2, xr = XI[xp] & 255, dx = (XI[xp] >> 8) - 2;
    const int ixg = AX.off[ex0] + xr, iyg = AY.off[ey0] + yr;
    const int rhox = AX.rowmap[ixg], rhoy = AY.rowmap[iyg];
    const long long ps1 = AX.prefix[rhox], ps2 = AY.prefix[rhoy];
    const int c1 = AX.rcnt[rhox], c2 = AY.rcnt[rhoy], P1 = AX.P[ixg * BW + dx + P], P2 = AY.P[iyg * BW + dy + P];
    const long long T0 = S.ax[0].tot, T10 = S.ax[1].tot * S.ax[0].tot;
    RA = ps2 * T10 + (long long)c2 * (ps1 * T0); RB = c2 * c1; RC = P2 * c1 + P1;
    woff = tid * 5;
  }
  int own_lo = (seg == 0 && pa.w_halo_lo == pa.w_lo) ? -1 : AW.off[ws];
  int own_hi = (seg == pa.nseg - 1 && !pa.open_hi) ? (1 << 30) : AW.off[we];

  // the run of layer li: its five old values are requested ahead of the element's MFMAs (fetch) and consumed behind the barrier (leave)
  struct Run { double *p[5]; double o[5]; bool on; };
  auto fetch = [&](int li, Run &r) {
    const int lay = T.lay0 + li;
    r.on = unit && li >= 0 && li < nl && T.cnt[li] > 0 && lay >= own_lo && lay < own_hi;
    if (!r.on) return;
    const long long base = RA + (long long)RB * T.pre[li] + (long long)RC * T.cnt[li];
#pragma unroll
    for (int d = 0; d < 5; ++d) { const int p0 = T.P[li * 8 + d]; r.p[d] = p0 >= 0 ? out.val + base + p0 : nullptr; r.o[d] = p0 >= 0 ? *r.p[d] : 0.0; }
  };
  auto leave = [&](int li, const Run &r) {
    if (!unit) return;
    double *w = win + (li & (PATCH_SLOTS - 1)) * PATCH_LAYER + woff;
    double v[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) { v[d] = w[d]; w[d] = 0.0; }
    if (!r.on) return;
#pragma unroll
    for (int d = 0; d < 5; ++d) if (r.p[d]) *r.p[d] = r.o[d] + v[d];
  };

  for (int ei = 0; ei < ne; ++ei) {
    Run run; fetch(ei, run);
    if (valid) {
      d4_t pk[3];
      pencil_mfma_p2k(pk, uxs, vys, T.zt + ei * 32, K, lane);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = code[t][i];
          if (c >= 0) (void)__hip_atomic_fetch_add(win + ((ei + (c & 3)) & (PATCH_SLOTS - 1)) * PATCH_LAYER + (c >> 2), pk[t][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (t == 1) { const int m = codem[i]; if (m >= 0) (void)__hip_atomic_fetch_add(win + ((ei + (m & 3)) & (PATCH_SLOTS - 1)) * PATCH_LAYER + (m >> 2), pk[t][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
    }
    __syncthreads();                  // every wavefront's element ei is in the window: layer ei is complete
    leave(ei, run);
  }
  if (seg == pa.nseg - 1 && !pa.open_hi)
    for (int k = 0; k < P; ++k) { Run run; fetch(ne + k, run); leave(ne + k, run); }
}

#ifndef IGX_RTC
// the launches of a Matrix assembly on a zeroed matrix: 2 x 2 colours of patches
static void launch_patches_p2(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, int &launches) {
  const int nx = s.elem_width[1], ny = s.elem_width[2], nw = s.elem_width[0];
  const int npx = (nx + PATCH_MX - 1) / PATCH_MX, npy = (ny + PATCH_MY - 1) / PATCH_MY;
  static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
    PatchArgs A; memset(&A, 0, sizeof(A));
    A.px_start = cx; A.px_step = 2; A.px_count = (npx - cx + 1) / 2;
    A.py_start = cy; A.py_step = 2; A.py_count = (npy - cy + 1) / 2;
    if (A.px_count <= 0 || A.py_count <= 0) continue;
    PencilArgs &pa = A.pa;
    pa.nelx = nx; pa.nely = ny; pa.w_lo = 0; pa.w_hi = nw; pa.w_halo_lo = 0; pa.open_hi = 0; pa.wpb = PATCH_W;
    const long long patches = (long long)A.px_count * A.py_count;
    // segments: the count with the fewest rounds x (length + halo), one workgroup per CU; the LDS holds <= 160 elements of tables
    int best = 1; long long bc = -1;
    for (int n = std::max(1, (nw + 159) / 160); n <= std::max(1, nw / 8); ++n) {
      const int len = (nw + n - 1) / n, ns = (nw + len - 1) / len;
      if (patch_lds_bytes(len + 3) > (size_t)160 * 1024) continue;
      const long long cost = ((patches * ns + ncu - 1) / ncu) * (len + (ns > 1 ? 2 : 0));
      if (bc < 0 || cost < bc) { bc = cost; best = n; }
    }
    if (s.env.nseg > 0) best = std::min(s.env.nseg, std::max(1, nw / 4));
    pa.seg_len = (nw + best - 1) / best; pa.nseg = (nw + pa.seg_len - 1) / pa.seg_len;
    pa.blocks_per_seg = (int)patches; pa.ne_max = pa.seg_len + 3;
    const size_t lds = patch_lds_bytes(pa.ne_max);
    if (lds > (size_t)160 * 1024) { pencil_launch_error() = "the patch walk's tables do not fit the LDS"; return; }
    auto kern = gram_patch_p2<false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(patches * pa.nseg)), dim3(PATCH_W * 64), lds, stream, S, out, A);
    launches++;
  }
}
#endif

}  // namespace igx
