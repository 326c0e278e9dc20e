// gram_patch.hpp -- the p = 2 Gram walk with the band rows combined across ALL THREE axes before they reach memory (round 6).
//
// gram_pencil_w6 (gram_mfma.hpp) gives each of a workgroup's twelve wavefronts a pencil of its own and a window of its own: a band
// row is combined along the walk and then read-add-written once per pencil -- 405 entries per element, nine colours of pencils.
// Here the twelve wavefronts walk a PATCH of 4 x 3 adjacent pencils in step and add into ONE window in LDS,
//   win[4 node layers (ring)][y pair][x pair][5 walk-axis offsets],
// a "pair" being two nodes of the patch on an axis that share one of its elements (24 on axis 1, 19 on axis 2).  A layer of the
// patch is complete when every wavefront has added its element: 2280 entries for 12 elements -- 190 per element instead of 405 --
// in runs that lie 25 to a 200-byte stretch of a CSR row, and patches conflict only with their neighbours: 4 colours instead of 9.
// Every thread of the workgroup owns ONE run (y pair, x pair) for the whole walk: its matrix address is a per-thread constant plus
// the layer's part, its old values are requested a step ahead -- before the element's MFMAs -- and consumed behind them.  One barrier
// per element.
//
// The price: up to nine wavefronts add to one window entry with ds_add_f64 and the ORDER of those adds is not fixed -- the assembly is
// correct to rounding (1e-12 against the oracle like every other path) but not bit-repeatable, which every other kernel of this library
// is.  A deterministic variant (each wavefront its own window, the runs gathered from the windows in a fixed order) was built, passed
// the bit-repeat tests and was slower than the pencil walk (profiles/r06_patch_walk.txt).  The shared window is + 23 % on config 2 (System
// driver, 128^3: 326 against 265 M el/s; Matrix driver 383 against 324) and the default where it applies; IGX_PATCH=0 keeps the
// bit-repeatable pencil walk.
//
// Degree 2, identity geometry, one rank, axis-0 walk; System and Matrix drivers, Dirichlet values on any face, first touch.
#pragma once
#include "gram_mfma.hpp"

namespace igx {

constexpr int PATCH_MX = 4, PATCH_MY = 3, PATCH_W = PATCH_MX * PATCH_MY;      // pencils (= wavefronts) of a workgroup
constexpr int PATCH_NX = PATCH_MX + 2, PATCH_NY = PATCH_MY + 2;                // nodes of a patch on axes 1, 2 (p = 2)
constexpr int PATCH_NXP = 5 * PATCH_MX + 4, PATCH_NYP = 5 * PATCH_MY + 4;      // node pairs that share an element of the patch
constexpr int PATCH_NODES = PATCH_NX * PATCH_NY;
constexpr int PATCH_LAYER = PATCH_NYP * PATCH_NXP * 5;                          // doubles per node layer of the window
constexpr int PATCH_SLOTS = 4;                                                  // ring: three layers being added to, one leaving

struct PatchArgs {
  PencilArgs pa;                         // walk-axis range and segments, forcing, first touch (as for the pencil walk)
  int px_start, px_step, px_count;       // patches of this colour: patch indices on axis 1 ...
  int py_start, py_step, py_count;       // ... and on axis 2
  int npx, npy;                          // patches per axis (first-touch rule: which neighbours exist)
  int dbg;                               // IGX_PATCH_DBG (timing experiments, wrong results): 1 no F sums, 2 no F stage, 4 no F leave, 
};

// LDS behind the walk's tables: the window, the pair tables (ints), the System driver's F and lifting stages (three deep: a layer's F leaves a
// step behind its band rows)
__host__ __device__ static inline size_t patch_lds_bytes(int ne_max) {
  return pencil_lds_bytes(ne_max, false, PATCH_W) + (size_t)PATCH_SLOTS * PATCH_LAYER * 8 + (size_t)(PATCH_NX * 5 + PATCH_NY * 5 + PATCH_NXP + PATCH_NYP + 8) * 4 +
         (size_t)3 * (PATCH_W * 9 + PATCH_NODES) * 8 + 64;
}

template <bool SYSTEM>
__global__ void __launch_bounds__(768, 3)
gram_patch_p2(SpaceDev S, OutDev out, PatchArgs A) {
  constexpr int P = 2, NB = 3, BW = 5, X = 1, Y = 2;
  extern __shared__ __attribute__((aligned(16))) double pencil_sm[];
  const PencilArgs &pa = A.pa;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg = blockIdx.x / pa.blocks_per_seg, patch = blockIdx.x - seg * pa.blocks_per_seg;
  const int tx = patch % A.px_count, ty = patch / A.px_count;
  const int ppx = A.px_start + tx * A.px_step, ppy = A.py_start + ty * A.py_step;
  const int ex0 = ppx * PATCH_MX, ey0 = ppy * PATCH_MY;
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
  double *Fp = reinterpret_cast<double *>(cntp + 8);            // [layer % 3][wave][9]: the leaving layer's F sums of each pencil
  double *corrp = Fp + 3 * PATCH_W * 9;                         // [layer % 3][node]: the lifting sum_k K_ik v_k of the row, summed over its runs
  {   // the walk-axis tables of the segment (as gram_pencil_body stages them), the windows, the pair tables
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
    for (int i = tid; i < 3 * PATCH_NODES; i += nthr) corrp[i] = 0.0;
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

  // ---- this wavefront's pencil: the 1-D rows of axes 1, 2 in LDS (scaled by sqrt(w J)), its window, the F factor of its F lanes
  double *rows = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + (pencil_lds_bytes(pa.ne_max, false, PATCH_W) - (size_t)2 * PATCH_W * 32 * 8));
  double *vys = rows + wave * 32, *uxs = rows + PATCH_W * 32 + wave * 32;
  double sxy = 0;
  {
    const double *__restrict__ TX = AX.tab + (size_t)elx * (NB * NB * NDER);
    const double *__restrict__ TY = AY.tab + (size_t)ely * (NB * NB * NDER);
    const double *__restrict__ WX = AX.w + elx * NB, *__restrict__ WYq = AY.w + ely * NB;
    if (lane < 32) { const int aa = lane >> 3, qq = (lane >> 1) & 3, kk = lane & 1; vys[lane] = (aa < NB && qq < NB) ? TY[(qq * NB + aa) * NDER + kk] * sqrt(WYq[qq] * AY.J[ely]) : 0.0; }
    else { const int l2 = lane - 32, qq = l2 >> 3, aa = (l2 >> 1) & 3, kk = l2 & 1; uxs[l2] = (aa < NB && qq < NB) ? TX[(qq * NB + aa) * NDER + kk] * sqrt(WX[qq] * AX.J[elx]) : 0.0; }
    if constexpr (SYSTEM) {      // F lane (fx = lane & 3, fy = (lane >> 2) & 3, slot = lane >> 4): forcing * sum_q w N on axes 1, 2 (as PencilLane::sxy)
      const int fx = lane & 3, fy = (lane >> 2) & 3;
      double sx = 0, sy = 0;
      if (fx < NB && fy < NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) { sx += WX[q] * TX[(q * NB + fx) * NDER]; sy += WYq[q] * TY[(q * NB + fy) * NDER]; }
      }
      sxy = valid ? pa.forcing * (sx * sy) * (AX.J[elx] * AY.J[ely]) : 0.0;
    }
  }
  const P2kLaneT<false> K = pencil_p2k_lane<false>(lane);      // (its operand offsets; the window offsets below replace the rest)
  // where the lane's results go in the window: (y pair, x pair, walk offset) of (row a, column b); code = offset * 4 + the row's a_w (the
  // ring slot's layer part), -1 for the padding.  Tiles (0,0), (0,1), (1,1) and the mirror entries of (0,1).
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
  const int fslot = lane >> 4;
  double Facc = 0;

  // ---- Dirichlet data of the patch (IGAElementBuildFix, src/petigaelem.c:1214-1283: a node is fixed by position, later faces override
  // earlier ones: axis 0, 1, 2; side 0, 1)
  bool bany = false, bxlo = false, bxhi = false, bylo = false, byhi = false; int bwlo = -1000, bwhi = -1000;
  double vwlo = 0, vwhi = 0, vxlo = 0, vxhi = 0, vylo = 0, vyhi = 0;
  if constexpr (SYSTEM) {
    bxlo = !AX.periodic && S.bcv[X][0].count > 0 && ex0 + AX.estart == 0;                       vxlo = S.bcv[X][0].value[0];
    bxhi = !AX.periodic && S.bcv[X][1].count > 0 && ex0 + mxv + AX.estart == AX.esizes;         vxhi = S.bcv[X][1].value[0];
    bylo = !AY.periodic && S.bcv[Y][0].count > 0 && ey0 + AY.estart == 0;                       vylo = S.bcv[Y][0].value[0];
    byhi = !AY.periodic && S.bcv[Y][1].count > 0 && ey0 + myv + AY.estart == AY.esizes;         vyhi = S.bcv[Y][1].value[0];
    if (!AW.periodic && S.bcv[0][0].count > 0 && AW.estart == 0) { bwlo = AW.off[0]; vwlo = S.bcv[0][0].value[0]; }
    if (!AW.periodic && S.bcv[0][1].count > 0 && AW.estart + AW.nel == AW.esizes) { bwhi = AW.off[AW.nel - 1] + P; vwhi = S.bcv[0][1].value[0]; }
    bany = bxlo || bxhi || bylo || byhi || bwlo > -1000 || bwhi > -1000;
  }
  auto fixed = [&](int xr, int yr, int lay, double &val) -> bool {      // node (layer lay, patch nodes yr, xr)
    bool f = false;
    if (lay == bwlo) { f = true; val = vwlo; }
    if (lay == bwhi) { f = true; val = vwhi; }
    if (bxlo && xr == 0) { f = true; val = vxlo; }
    if (bxhi && xr == mxv + 1) { f = true; val = vxhi; }
    if (bylo && yr == 0) { f = true; val = vylo; }
    if (byhi && yr == myv + 1) { f = true; val = vyhi; }
    return f;
  };

  // ---- this thread's run of the band rows: (y pair, x pair) for the whole walk; pos = RA + RB prefix0(layer) + RC count0(layer) + P0(layer, d)
  const bool unit = tid < nxp * nyp;
  long long RA = 0; int RB = 0, RC = 0;
  int uyr = 0, udy = 0, uxr = 0, udx = 0;
  bool ufirst = false;
  if (unit) {
    const int yp = tid / nxp, xp = tid - yp * nxp;
    uyr = YI[yp] & 255; udy = (YI[yp] >> 8) - 2; uxr = XI[xp] & 255; udx = (XI[xp] >> 8) - 2;
    const int ixg = AX.off[ex0] + uxr, iyg = AY.off[ey0] + uyr;
    const int rhox = AX.rowmap[ixg], rhoy = AY.rowmap[iyg];
    const long long ps1 = AX.prefix[rhox], ps2 = AY.prefix[rhoy];
    const int c1 = AX.rcnt[rhox], c2 = AY.rcnt[rhoy], P1 = AX.P[ixg * BW + udx + P], P2 = AY.P[iyg * BW + udy + P];
    const long long T0 = S.ax[0].tot, T10 = S.ax[1].tot * S.ax[0].tot;
    RA = ps2 * T10 + (long long)c2 * (ps1 * T0); RB = c2 * c1; RC = P2 * c1 + P1;
    // First touch: colours are launched (0,0), (1,0), (0,1), (1,1); a pair is also held by the neighbour patch below (above) when both its
    // nodes are among the two the patches share.  This patch stores -- no read -- when no patch of an EARLIER colour holds the pair:
    // per axis, an even patch always, an odd one unless a neighbour holds the pair too.
    if (pa.first_touch) {
      auto axis_first = [](int pp, int np, int mv, int r, int c) {
        if ((pp & 1) == 0) return true;
        const bool below = pp > 0 && r <= 1 && c <= 1, above = pp + 1 < np && r >= mv && c >= mv;
        return !(below || above);
      };
      ufirst = axis_first(ppx, A.npx, mxv, uxr, uxr + udx) && axis_first(ppy, A.npy, myv, uyr, uyr + udy);
    }
  }
  const int own_lo = (seg == 0 && pa.w_halo_lo == pa.w_lo) ? -1 : AW.off[ws];
  const int own_hi = (seg == pa.nseg - 1 && !pa.open_hi) ? (1 << 30) : AW.off[we];
  // rows of the F stage: thread t < 30 is patch node (yr, xr) = (t / 6, t % 6); its F row without the walk-axis part (read ONCE: two dependent
  // global loads per element in the F leave made wavefront 0 late at every barrier -- 1.2 of 7.9 ms at 128^3)
  long long frowxy = 0;
  constexpr int FT0 = (PATCH_W - 1) * 64;
  static_assert(PATCH_NYP * PATCH_NXP <= FT0 && PATCH_NODES <= 64, "the runs leave the last wavefront free for the F rows");
  if (SYSTEM && tid >= FT0 && tid < FT0 + PATCH_NODES) {
    const int yr = (tid - FT0) / PATCH_NX, xr = (tid - FT0) - yr * PATCH_NX;
    if (xr <= mxv + 1 && yr <= myv + 1) frowxy = (long long)S.ax[0].nrow * AX.rowmap[AX.off[ex0] + xr] + (long long)S.ax[0].nrow * S.ax[1].nrow * AY.rowmap[AY.off[ey0] + yr];
  }

  // the run of layer li: its five old values are requested a step ahead (fetch) and consumed when the layer is complete (leave)
  struct Run { long long base; double o[5]; int p0[5]; bool on, full; };
  auto fetch = [&](int li, Run &r) {
    const int lay = T.lay0 + li;
    r.on = unit && li >= 0 && li < nl && T.cnt[li] > 0 && lay >= own_lo && lay < own_hi;
    if (!r.on) return;
    r.base = RA + (long long)RB * T.pre[li] + (long long)RC * T.cnt[li];
    r.full = true;
#pragma unroll
    for (int d = 0; d < 5; ++d) { r.p0[d] = T.P[li * 8 + d]; r.full = r.full && r.p0[d] == d; }
    if (r.full) {      // an interior layer: the run is 40 contiguous bytes
      const double *p = out.val + r.base;
      if (ufirst) { r.o[0] = r.o[1] = r.o[2] = r.o[3] = r.o[4] = 0.0; }
      else { const d2u_t a = *reinterpret_cast<const d2u_t *>(p), b = *reinterpret_cast<const d2u_t *>(p + 2); r.o[0] = a[0]; r.o[1] = a[1]; r.o[2] = b[0]; r.o[3] = b[1]; r.o[4] = p[4]; }
    } else {
#pragma unroll
      for (int d = 0; d < 5; ++d) r.o[d] = (r.p0[d] >= 0 && !ufirst) ? out.val[r.base + r.p0[d]] : 0.0;
    }
  };
  // elements of the walk that hold layer li (the diagonal of a fixed row counts them: each sets K_kk = 1)
  auto held_w = [&](int li) { return min(li, ne - 1) - max(li - P, 0) + 1; };
  auto leave = [&](int li, const Run &r) {
    if (!unit) return;
    double *w = win + (li & (PATCH_SLOTS - 1)) * PATCH_LAYER + tid * 5;
    double v[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) { v[d] = w[d]; w[d] = 0.0; }
    if (!r.on) return;
    const int lay = T.lay0 + li;
    if constexpr (SYSTEM) {      // IGAElementFixSystem (src/petigaelem.c:1377-1387) on the combined run; the lifting of its row goes to the F stage
      double corr = 0;
      // (only the rows the fix-up can reach: a patch on a face of axis 1 or 2, or within p layers of a fixed layer of the walk axis)
      if (bany && (bxlo || bxhi || bylo || byhi || (lay >= bwlo - P && lay <= bwlo + P) || (lay >= bwhi - P && lay <= bwhi + P))) {
        double rv = 0; const bool rf = fixed(uxr, uyr, lay, rv);
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          double cv = 0; const bool cf = fixed(uxr + udx, uyr + udy, lay + d - P, cv);
          if (cf) corr += v[d] * cv;
          if (rf || cf) v[d] = (d == P && udx == 0 && udy == 0 && rf) ? (double)(held_w(li) * (min(uxr, mxv - 1) - max(uxr - P, 0) + 1) * (min(uyr, myv - 1) - max(uyr - P, 0) + 1)) : 0.0;
        }
      }
      // (this path is the one that is not bit-repeatable anyway: the row's lifting is summed with LDS atomics too, by the runs that have any)
      if (corr != 0.0) (void)__hip_atomic_fetch_add(corrp + (li % 3) * PATCH_NODES + uyr * PATCH_NX + uxr, corr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (r.full) {
      double *p = out.val + r.base;
      d2u_t a, b; a[0] = r.o[0] + v[0]; a[1] = r.o[1] + v[1]; b[0] = r.o[2] + v[2]; b[1] = r.o[3] + v[3];
      *reinterpret_cast<d2u_t *>(p) = a; *reinterpret_cast<d2u_t *>(p + 2) = b; p[4] = r.o[4] + v[4];
    } else {
#pragma unroll
      for (int d = 0; d < 5; ++d) if (r.p0[d] >= 0) out.val[r.base + r.p0[d]] = r.o[d] + v[d];
    }
  };
  // F of layer li: thread t < 30 = patch node (yr, xr): the pencils' sums (Fp) and its runs' liftings (corrp), each in a fixed order
  auto leave_f = [&](int li) {
    if constexpr (SYSTEM) {
      const int lay = T.lay0 + li;
      const bool on = li >= 0 && li < nl && T.cnt[li] > 0 && lay >= own_lo && lay < own_hi;
      const int ft = tid - FT0;      // (the F rows belong to threads of the last wavefront, which own no run)
      if (!on || ft < 0 || ft >= PATCH_NODES) return;
      const int yr = ft / PATCH_NX, xr = ft - yr * PATCH_NX;
      if (xr > mxv + 1 || yr > myv + 1) return;
      double f = 0;
      const double *Fl = Fp + (li % 3) * (PATCH_W * 9); double *cl = corrp + (li % 3) * PATCH_NODES;
      for (int j = max(0, yr - 2); j <= min(yr, myv - 1); ++j)
        for (int i = max(0, xr - 2); i <= min(xr, mxv - 1); ++i) f += Fl[(j * PATCH_MX + i) * 9 + 3 * (yr - j) + (xr - i)];
      double fv = 0;
      if (bany && (bxlo || bxhi || bylo || byhi || (lay >= bwlo - P && lay <= bwlo + P) || (lay >= bwhi - P && lay <= bwhi + P))) {
        if (fixed(xr, yr, lay, fv)) f = fv * (double)(held_w(li) * (min(xr, mxv - 1) - max(xr - P, 0) + 1) * (min(yr, myv - 1) - max(yr - P, 0) + 1));
        else f -= cl[ft];
      }
      cl[ft] = 0.0;
      const long long row = (long long)T.rho[li] + frowxy;
      // one memory-side add per row and launch (patches of a colour share no node): order-free, and nothing waits for it -- a load / add /
      // store chain here makes its wavefront late at every barrier
      (void)__hip_atomic_fetch_add(out.vec + row, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  auto f_stage = [&](int li) {      // the leaving layer's F sums of this pencil (slot 0 of the F lanes), then the slots slide
    if constexpr (SYSTEM) {
      if (fslot == 0 && (lane & 3) < NB && ((lane >> 2) & 3) < NB) Fp[(li % 3) * (PATCH_W * 9) + wave * 9 + 3 * ((lane >> 2) & 3) + (lane & 3)] = Facc;
      const double up = __shfl_down(Facc, 16);
      Facc = (fslot >= NB - 1) ? 0.0 : up;
    }
  };
  Run run; run.on = false;
  fetch(0, run);
  long long st_mfma = 0, st_wait = 0, st_leave = 0, st_t0 = 0; const long long st_wall0 = (kDebug && pa.debug_buf) ? wall_clock64() : 0;
  for (int ei = 0; ei < ne; ++ei) {
    if (kDebug && pa.debug_buf) st_t0 = __builtin_readcyclecounter();
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
    if constexpr (SYSTEM) {      // F_a += f J prod_d sum_q w N: the walk-axis factor is sum_q sqrt(wJ) (sqrt(wJ) N)
      double sw = 0;
      const int fs = fslot < NB ? fslot : 0;
#pragma unroll
      for (int q = 0; q < NB; ++q) sw += T.wq[ei * 4 + q] * T.zt[ei * 32 + (q * 4 + fs) * 2];
      if (fslot < NB && !(A.dbg & 1)) Facc += sxy * sw;
      if (!(A.dbg & 2)) f_stage(ei);
    }
    long long st_t1 = 0, st_t2 = 0;
    if (kDebug && pa.debug_buf) st_t1 = __builtin_readcyclecounter();
    __syncthreads();      // every wavefront's element ei is in the window: layer ei is complete (and its F sums are staged)
    if (kDebug && pa.debug_buf) st_t2 = __builtin_readcyclecounter();
    leave(ei, run);
    if (ei >= 1 && !(A.dbg & 4)) leave_f(ei - 1);      // (a step behind: the liftings of its runs were staged before this barrier)
    fetch(ei + 1, run);
    if (kDebug && pa.debug_buf) { const long long t3 = __builtin_readcyclecounter(); st_mfma += st_t1 - st_t0; st_wait += st_t2 - st_t1; st_leave += t3 - st_t2; }
  }
  if (kDebug && pa.debug_buf && lane == 0) {      // -DIGX_DEBUG: per wavefront [MFMAs + window adds | wait at the barrier | leave + next fetch] summed over the walk, elements; workgroup record
    long long *d = pa.debug_buf + ((size_t)blockIdx.x * PATCH_W + wave) * 4;
    d[0] = st_mfma; d[1] = st_wait; d[2] = st_leave; d[3] = ne;
    if (wave == 0) { long long *w = pa.debug_buf + (size_t)gridDim.x * PATCH_W * 4 + (size_t)blockIdx.x * 4; w[0] = st_wall0; w[1] = wall_clock64(); w[2] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4); w[3] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20); }
  }
  if constexpr (SYSTEM) { __syncthreads(); leave_f(ne - 1); }
  if (seg == pa.nseg - 1 && !pa.open_hi)
    for (int k = 0; k < P; ++k) {
      f_stage(ne + k);
      leave(ne + k, run);
      if constexpr (SYSTEM) { __syncthreads(); leave_f(ne + k); }
      fetch(ne + k + 1, run);
    }
}

// ---- the same walk for the Tangent of a nonlinear scalar form on the identity geometry (config 4: demo/CahnHilliard3D.c:111-179 through
// IGAComputeIJacobian).  state_pencil_k (gram_mfma.hpp) gives each of eight wavefronts a pencil, a window and a flush of its own; here the
// eight walk a patch of 4 x 2 pencils in step: the state's point values and the 140 MFMAs of an element are the pencil walk's
// (pencil_state_eval, pencil_mfma_state_p2k), the four tiles -- a Tangent is not symmetric -- go to ONE window, and every thread owns
// one run of the patch's band rows (14 x 24 pairs of nodes: 210 entries per element instead of 405, 4 colours instead of 9).
// IGAElementFixJacobian on the combined run (src/petigaelem.c:1483-1500).  Not bit-repeatable, like gram_patch_p2.
constexpr int SPATCH_MX = 4, SPATCH_MY = 2, SPATCH_W = SPATCH_MX * SPATCH_MY;
constexpr int SPATCH_NX = SPATCH_MX + 2, SPATCH_NY = SPATCH_MY + 2;
constexpr int SPATCH_NXP = 5 * SPATCH_MX + 4, SPATCH_NYP = 5 * SPATCH_MY + 4;
constexpr int SPATCH_LAYER = SPATCH_NYP * SPATCH_NXP * 5;
__host__ __device__ static inline size_t spatch_lds_bytes(int ne_max) {
  return pencil_lds_bytes(ne_max, true, SPATCH_W) + (size_t)PATCH_SLOTS * SPATCH_LAYER * 8 + (size_t)SPATCH_W * GEO_DOUBLES * 8 + (size_t)SPATCH_W * STATE_D2 * 8 +
         (size_t)(SPATCH_NX * 5 + SPATCH_NY * 5 + SPATCH_NXP + SPATCH_NYP + 8) * 4 + 64;
}

template <class Form>
__global__ void __launch_bounds__(512, 2)
state_patch_p2(SpaceDev S, OutDev out, PatchArgs A, ParamsDev prm) {
  static_assert(pencil_state_of<Form>::v, "state_patch_p2: the form declares PENCIL_NFEAT, PENCIL_NC, pencil_coef and pencil_trial");
  constexpr int P = 2, NB = 3, BW = 5, X = 1, Y = 2, GZ = GEO_Z, GD = GEO_DOUBLES;
  extern __shared__ __attribute__((aligned(16))) double pencil_sm[];
  const PencilArgs &pa = A.pa;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg = blockIdx.x / pa.blocks_per_seg, patch = blockIdx.x - seg * pa.blocks_per_seg;
  const int tx = patch % A.px_count, ty = patch / A.px_count;
  const int ppx = A.px_start + tx * A.px_step, ppy = A.py_start + ty * A.py_step;
  const int ex0 = ppx * SPATCH_MX, ey0 = ppy * SPATCH_MY;
  const int mxv = min(SPATCH_MX, pa.nelx - ex0), myv = min(SPATCH_MY, pa.nely - ey0);
  const int wi = wave % SPATCH_MX, wj = wave / SPATCH_MX;
  const bool valid = wi < mxv && wj < myv;
  const int elx = ex0 + (valid ? wi : 0), ely = ey0 + (valid ? wj : 0);
  const AxisDev &AW = S.ax[0], &AX = S.ax[1], &AY = S.ax[2];
  const int ws = pa.w_lo + seg * pa.seg_len, we = min(ws + pa.seg_len, pa.w_hi);
  const int wh = max(ws - P, pa.w_halo_lo);
  const int ne = we - wh, nl = ne + P;

  PencilLds T = pencil_lds_carve(pencil_sm, pa.ne_max, true);      // (no scaled walk-axis rows: an element's raw rows travel with its point numbers)
  T.lay0 = AW.off[wh];
  char *base = reinterpret_cast<char *>(pencil_sm) + pencil_lds_bytes(pa.ne_max, true, SPATCH_W);
  double *win = reinterpret_cast<double *>(base);
  double *geo = win + PATCH_SLOTS * SPATCH_LAYER + wave * GD;
  double *d2w = win + PATCH_SLOTS * SPATCH_LAYER + SPATCH_W * GD + wave * STATE_D2;
  int *XP = reinterpret_cast<int *>(win + PATCH_SLOTS * SPATCH_LAYER + SPATCH_W * GD + SPATCH_W * STATE_D2), *YP = XP + SPATCH_NX * 5, *XI = YP + SPATCH_NY * 5, *YI = XI + SPATCH_NXP, *cntp = YI + SPATCH_NYP;
  {
    const int nthr = SPATCH_W * 64;
    for (int i = tid; i < ne * 4; i += nthr) { const int e = i >> 2, q = i & 3, eg = wh + e; T.wq[i] = (q < NB) ? AW.w[eg * NB + q] * AW.J[eg] : 0.0; }      // (the weight itself: pencil_coef takes the whole JW)
    for (int i = tid; i < ne; i += nthr) T.Jz[i] = AW.J[wh + i];
    for (int i = tid; i < nl; i += nthr) {
      const int lay = T.lay0 + i;
      if (lay < AW.gwidth) {
        const int rho = AW.rowmap[lay];
        T.rho[i] = rho; T.cnt[i] = AW.rcnt[rho]; T.pre[i] = AW.prefix[rho];
        for (int d = 0; d < BW; ++d) T.P[i * 8 + d] = AW.P[lay * BW + d];
      } else { T.rho[i] = 0; T.cnt[i] = -1; T.pre[i] = 0; }
    }
    for (int i = tid; i < PATCH_SLOTS * SPATCH_LAYER; i += nthr) win[i] = 0.0;
    if (tid < 2) {
      const int nn = tid == 0 ? SPATCH_NX : SPATCH_NY, mv = tid == 0 ? mxv : myv;
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

  // ---- this wavefront's pencil: raw 1-D rows of axes 1, 2 and their second derivatives (LDS), the Gauss weights of the lane's point
  double *rows = reinterpret_cast<double *>(reinterpret_cast<char *>(pencil_sm) + (pencil_lds_bytes(pa.ne_max, true, SPATCH_W) - (size_t)2 * SPATCH_W * 32 * 8));
  double *vyr = rows + wave * 32, *uxr = rows + SPATCH_W * 32 + wave * 32;
  double wjxy = 0;
  {
    const double *__restrict__ TX = AX.tab + (size_t)elx * (NB * NB * NDER);
    const double *__restrict__ TY = AY.tab + (size_t)ely * (NB * NB * NDER);
    if (lane < 32) { const int aa = lane >> 3, qq = (lane >> 1) & 3, kk = lane & 1; vyr[lane] = (aa < NB && qq < NB) ? TY[(qq * NB + aa) * NDER + kk] : 0.0; }
    else { const int l2 = lane - 32, qq = l2 >> 3, aa = (l2 >> 1) & 3, kk = l2 & 1; uxr[l2] = (aa < NB && qq < NB) ? TX[(qq * NB + aa) * NDER + kk] : 0.0; }
    if (lane < 16) { const int qq = lane >> 2, aa = lane & 3; d2w[lane] = (qq < NB && aa < NB) ? TX[(qq * NB + aa) * NDER + 2] : 0.0; }
    else if (lane < 32) { const int aa = (lane - 16) >> 2, qq = lane & 3; d2w[lane] = (qq < NB && aa < NB) ? TY[(qq * NB + aa) * NDER + 2] : 0.0; }
    const int gqx = lane & 3, gqy = (lane >> 2) & 3;
    if (gqx < NB && gqy < NB) wjxy = (AX.w[elx * NB + gqx] * AX.J[elx]) * (AY.w[ely * NB + gqy] * AY.J[ely]);
  }
  const P2kLaneT<false> K = pencil_p2k_lane<false>(lane);      // (its operand offsets)
  int code[4][2];      // tile (Ta, Tb) = code[2 Ta + Tb], rows i = 0..3 in 16-bit halves: offset * 4 + the row's a_w, 0xffff for the padding
  {
    auto split = [](int f, int &aw, int &ay, int &ax) { aw = f / 9; const int r = f - 9 * aw; ay = r / 3; ax = r - 3 * ay; };
    auto entry = [&](int a, int b) -> int {
      if (a >= 27 || b >= 27 || !valid) return 0xffff;
      int aw, ay, ax, bw, by, bx; split(a, aw, ay, ax); split(b, bw, by, bx);
      const int yp = YP[(wj + ay) * 5 + (by - ay + 2)], xp = XP[(wi + ax) * 5 + (bx - ax + 2)];
      return (((yp * nxp + xp) * 5 + (bw - aw + 2)) << 2) | aw;
    };
    static_assert(SPATCH_LAYER * 4 < 0xffff, "a window offset and its slot fit 16 bits");
#pragma unroll
    for (int Ta = 0; Ta < 2; ++Ta)
#pragma unroll
      for (int Tb = 0; Tb < 2; ++Tb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          code[Ta * 2 + Tb][h] = entry(16 * Ta + 4 * (2 * h) + (lane >> 4), 16 * Tb + (lane & 15)) | (entry(16 * Ta + 4 * (2 * h + 1) + (lane >> 4), 16 * Tb + (lane & 15)) << 16);
  }
  // ---- Dirichlet data of the patch (a node is fixed by position; later faces override earlier ones)
  const bool bxlo = !AX.periodic && S.bcv[X][0].count > 0 && ex0 + AX.estart == 0, bxhi = !AX.periodic && S.bcv[X][1].count > 0 && ex0 + mxv + AX.estart == AX.esizes;
  const bool bylo = !AY.periodic && S.bcv[Y][0].count > 0 && ey0 + AY.estart == 0, byhi = !AY.periodic && S.bcv[Y][1].count > 0 && ey0 + myv + AY.estart == AY.esizes;
  const double vxlo = S.bcv[X][0].value[0], vxhi = S.bcv[X][1].value[0], vylo = S.bcv[Y][0].value[0], vyhi = S.bcv[Y][1].value[0];
  int bwlo = -1000, bwhi = -1000; double vwlo = 0, vwhi = 0;
  if (!AW.periodic && S.bcv[0][0].count > 0 && AW.estart == 0) { bwlo = AW.off[0]; vwlo = S.bcv[0][0].value[0]; }
  if (!AW.periodic && S.bcv[0][1].count > 0 && AW.estart + AW.nel == AW.esizes) { bwhi = AW.off[AW.nel - 1] + P; vwhi = S.bcv[0][1].value[0]; }
  const bool bany = bxlo || bxhi || bylo || byhi || bwlo > -1000 || bwhi > -1000;
  auto fixed = [&](int xr, int yr, int lay, double &val) -> bool {      // node (layer lay, patch nodes yr, xr)
    bool f = false;
    if (lay == bwlo) { f = true; val = vwlo; }
    if (lay == bwhi) { f = true; val = vwhi; }
    if (bxlo && xr == 0) { f = true; val = vxlo; }
    if (bxhi && xr == mxv + 1) { f = true; val = vxhi; }
    if (bylo && yr == 0) { f = true; val = vylo; }
    if (byhi && yr == myv + 1) { f = true; val = vyhi; }
    return f;
  };
  // ---- this thread's run
  // Wavefronts w and w + 4 share a SIMD; group 0 (wavefronts 0-3) and group 1 (4-7) run half a period apart as in the pencil walk -- one
  // issues its MFMAs while the other leaves its runs and evaluates its next state -- and each group owns half of the patch's runs.
  const int grp = wave >> 2;
  const int nruns = nxp * nyp, half = (nruns + 1) >> 1;
  const int rid = (tid & 255) + grp * half;
  const bool unit = (tid & 255) < half && rid < nruns;
  long long RA = 0; int RB = 0, RC = 0, uyr = 0, udy = 0, uxr_ = 0, udx = 0;
  bool ufirst = false;
  if (unit) {
    const int yp = rid / nxp, xp = rid - yp * nxp;
    uyr = YI[yp] & 255; udy = (YI[yp] >> 8) - 2; uxr_ = XI[xp] & 255; udx = (XI[xp] >> 8) - 2;
    const int ixg = AX.off[ex0] + uxr_, iyg = AY.off[ey0] + uyr;
    const int rhox = AX.rowmap[ixg], rhoy = AY.rowmap[iyg];
    const long long ps1 = AX.prefix[rhox], ps2 = AY.prefix[rhoy];
    const int c1 = AX.rcnt[rhox], c2 = AY.rcnt[rhoy], P1 = AX.P[ixg * BW + udx + P], P2 = AY.P[iyg * BW + udy + P];
    const long long T0 = S.ax[0].tot, T10 = S.ax[1].tot * S.ax[0].tot;
    RA = ps2 * T10 + (long long)c2 * (ps1 * T0); RB = c2 * c1; RC = P2 * c1 + P1;
    if (pa.first_touch) {
      auto axis_first = [](int pp, int np, int mv, int r, int c) {
        if ((pp & 1) == 0) return true;
        const bool below = pp > 0 && r <= 1 && c <= 1, above = pp + 1 < np && r >= mv && c >= mv;
        return !(below || above);
      };
      ufirst = axis_first(ppx, A.npx, mxv, uxr_, uxr_ + udx) && axis_first(ppy, A.npy, myv, uyr, uyr + udy);
    }
  }
  const int own_lo = (seg == 0 && pa.w_halo_lo == pa.w_lo) ? -1 : AW.off[ws];
  const int own_hi = (seg == pa.nseg - 1 && !pa.open_hi) ? (1 << 30) : AW.off[we];
  const long long rsx = S.ax[0].nrow, rsy = (long long)S.ax[0].nrow * S.ax[1].nrow;
  // U rows of this lane's node (aw, ay, ax) without the walk-axis part
  const int naw = lane >> 4, nay = (lane >> 2) & 3, nax = lane & 3;
  const bool isnode = valid && naw < NB && nay < NB && nax < NB;
  const long long urowxy = isnode ? rsx * AX.rowmap[AX.off[elx] + nax] + rsy * AY.rowmap[AY.off[ely] + nay] : 0;

  struct Run { long long base; double o[5]; bool on, full; };
  auto fetch = [&](int li, Run &r) {
    const int lay = T.lay0 + li;
    r.on = unit && li >= 0 && li < nl && T.cnt[li] > 0 && lay >= own_lo && lay < own_hi;
    if (!r.on) return;
    r.base = RA + (long long)RB * T.pre[li] + (long long)RC * T.cnt[li];
    r.full = true;
#pragma unroll
    for (int d = 0; d < 5; ++d) r.full = r.full && T.P[li * 8 + d] == d;
    if (r.full) {
      const double *p = out.val + r.base;
      if (ufirst) { r.o[0] = r.o[1] = r.o[2] = r.o[3] = r.o[4] = 0.0; }
      else { const d2u_t a = *reinterpret_cast<const d2u_t *>(p), b = *reinterpret_cast<const d2u_t *>(p + 2); r.o[0] = a[0]; r.o[1] = a[1]; r.o[2] = b[0]; r.o[3] = b[1]; r.o[4] = p[4]; }
    } else {
#pragma unroll
      for (int d = 0; d < 5; ++d) { const int p0 = T.P[li * 8 + d]; r.o[d] = (p0 >= 0 && !ufirst) ? out.val[r.base + p0] : 0.0; }
    }
  };
  auto held_w = [&](int li) { return min(li, ne - 1) - max(li - P, 0) + 1; };
  auto leave = [&](int li, const Run &r) {
    if (!unit || li < 0) return;
    double *w = win + (li & (PATCH_SLOTS - 1)) * SPATCH_LAYER + rid * 5;
    double v[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) { v[d] = w[d]; w[d] = 0.0; }
    if (!r.on) return;
    const int lay = T.lay0 + li;
    // IGAElementFixJacobian (src/petigaelem.c:1483-1500) on the combined run: rows and columns of fixed nodes emptied, the diagonal counting the elements
    if (bany && (bxlo || bxhi || bylo || byhi || (lay >= bwlo - P && lay <= bwlo + P) || (lay >= bwhi - P && lay <= bwhi + P))) {
      double rv = 0; const bool rf = fixed(uxr_, uyr, lay, rv);
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        double cv = 0; const bool cf = fixed(uxr_ + udx, uyr + udy, lay + d - P, cv);
        if (rf || cf) v[d] = (d == P && udx == 0 && udy == 0 && rf) ? (double)(held_w(li) * (min(uxr_, mxv - 1) - max(uxr_ - P, 0) + 1) * (min(uyr, myv - 1) - max(uyr - P, 0) + 1)) : 0.0;
      }
    }
    if (r.full) {
      double *p = out.val + r.base;
      d2u_t a, b; a[0] = r.o[0] + v[0]; a[1] = r.o[1] + v[1]; b[0] = r.o[2] + v[2]; b[1] = r.o[3] + v[3];
      *reinterpret_cast<d2u_t *>(p) = a; *reinterpret_cast<d2u_t *>(p + 2) = b; p[4] = r.o[4] + v[4];
    } else {
#pragma unroll
      for (int d = 0; d < 5; ++d) { const int p0 = T.P[li * 8 + d]; if (p0 >= 0) out.val[r.base + p0] = r.o[d] + v[d]; }
    }
  };
  // the state of element wh + ei at its Gauss points -> the form's point numbers in this wavefront's geo area (as gram_pencil_body's geometry())
  auto state = [&](int ei) {
    if (lane < 32) { const int q = lane >> 3, a = (lane >> 1) & 3, k = lane & 1; geo[GZ + lane] = (q < NB && a < NB) ? AW.tab[((size_t)(wh + ei) * NB * NB + q * NB + a) * NDER + k] : 0.0; }
    if (lane >= 32 && lane < 48) { const int l2 = lane - 32, q = l2 >> 2, a = l2 & 3; d2w[lane] = (q < NB && a < NB) ? AW.tab[((size_t)(wh + ei) * NB * NB + q * NB + a) * NDER + 2] : 0.0; }
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double uc = 0.0;
    if (isnode) {
      const int li = ei + naw;
      const long long urow = (long long)T.rho[li] + urowxy;
      uc = out.U[urow];
      double fv = 0;
      if (bany && fixed(wi + nax, wj + nay, T.lay0 + li, fv)) uc = S.fixtable ? S.fixtable[urow] : fv;
    }
    double xpar[3] = {0, 0, 0};
    const int gqx = lane & 3, gqy = (lane >> 2) & 3, gqw = lane >> 4;
    if (gqx < NB && gqy < NB && gqw < NB && valid) { xpar[0] = AW.pt[(wh + ei) * NB + gqw]; xpar[1] = AX.pt[elx * NB + gqx]; xpar[2] = AY.pt[ely * NB + gqy]; }
    pencil_state_eval<P, Form>(geo, d2w, lane, uxr, vyr, geo + GZ, wjxy * (gqw < NB ? T.wq[ei * 4 + gqw] : 0.0), uc, prm.v, out.shift, out.t, xpar);
  };

  // Schedule (a column = the span between two barriers):
  //   group 0:  mfma(0) | flush(0) | mfma(1) | flush(1) | ...          flush(e) of group 0: layer e - 1 leaves, state of element e + 1
  //   group 1:          | mfma(0)  | flush(0)| mfma(1)  | ...          flush(e) of group 1: layer e leaves,     state of element e + 1
  // Layer e is complete when both groups' mfma(e) are: group 1 leaves its runs of it right away, group 0 a column later, both while the
  // other group adds to the layers above only; with four slots, slot (e & 3) is empty again before mfma(e + 2) of group 0 reaches layer e + 4.
  const int lag = grp == 0 ? 1 : 0;
  Run run; run.on = false;
  fetch(-lag, run);
  if (valid) state(0);
  if (grp == 1) __syncthreads();
  for (int ei = 0; ei < ne; ++ei) {
    if (valid) {
      d4_t pk[4];
      if (!(A.dbg & 8)) pencil_mfma_state_p2k<Form, false, P2kLaneT<false>>(pk, uxr, vyr, geo + GZ, d2w, geo, K, lane);
      else pk[0] = pk[1] = pk[2] = pk[3] = (d4_t){1, 1, 1, 1};
      if (!(A.dbg & 4))
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = (code[t][i >> 1] >> (16 * (i & 1))) & 0xffff;
          if (c != 0xffff) (void)__hip_atomic_fetch_add(win + ((ei + (c & 3)) & (PATCH_SLOTS - 1)) * SPATCH_LAYER + (c >> 2), pk[t][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);      // (the partner wavefront streams MFMAs now: without priority this one gets the left-over issue slots)
    if (!(A.dbg & 2)) { leave(ei - lag, run); fetch(ei - lag + 1, run); }
    if (valid && ei + 1 < ne && !(A.dbg & 1)) state(ei + 1);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  if (grp == 0) __syncthreads();
  // every element is in the window: what is left of it (group 0's runs of the last layer; the last segment's two layers beyond)
  const int last = (seg == pa.nseg - 1 && !pa.open_hi) ? ne + P - 1 : ne - 1;
  for (int li = ne - lag; li <= last; ++li) { leave(li, run); fetch(li + 1, run); }
}

#ifndef IGX_RTC
// the launches of an assembly: 2 x 2 colours of patches, (0,0) first (the first-touch rule of the kernel knows the order)
static void launch_patches_p2(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, int &launches, double forcing, bool first_touch) {
  const int nx = s.elem_width[1], ny = s.elem_width[2], nw = s.elem_width[0];
  const int npx = (nx + PATCH_MX - 1) / PATCH_MX, npy = (ny + PATCH_MY - 1) / PATCH_MY;
  static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  const bool sys = out.op == OP_SYSTEM;
  for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
    PatchArgs A; memset(&A, 0, sizeof(A));
    A.px_start = cx; A.px_step = 2; A.px_count = (npx - cx + 1) / 2;
    A.py_start = cy; A.py_step = 2; A.py_count = (npy - cy + 1) / 2;
    A.npx = npx; A.npy = npy; { const char *d = getenv("IGX_PATCH_DBG"); A.dbg = d ? atoi(d) : 0; }
    if (A.px_count <= 0 || A.py_count <= 0) continue;
    PencilArgs &pa = A.pa;
    pa.forcing = forcing; pa.first_touch = first_touch ? 1 : 0;
    pa.nelx = nx; pa.nely = ny; pa.w_lo = 0; pa.w_hi = nw; pa.w_halo_lo = 0; pa.open_hi = 0; pa.wpb = PATCH_W;
    const long long patches = (long long)A.px_count * A.py_count;
    // segments: the count with the fewest rounds x (length + halo), one workgroup per CU
    int best = 1; long long bc = -1;
    for (int n = 1; n <= std::max(1, nw / 2); ++n) {      // (down to two elements per segment: pencil_segments)
      const int len = (nw + n - 1) / n, ns = (nw + len - 1) / len;
      if (patch_lds_bytes(len + 3) > (size_t)160 * 1024) continue;
      const long long cost = ((patches * ns + ncu - 1) / ncu) * (len + (ns > 1 ? 2 : 0) + 1);
      if (bc < 0 || cost < bc) { bc = cost; best = n; }
    }
    if (s.env.nseg > 0) best = std::min(s.env.nseg, std::max(1, nw / 2));
    pa.seg_len = (nw + best - 1) / best; pa.nseg = (nw + pa.seg_len - 1) / pa.seg_len;
    pa.blocks_per_seg = (int)patches; pa.ne_max = pa.seg_len + 3;
    const size_t lds = patch_lds_bytes(pa.ne_max);
    if (lds > (size_t)160 * 1024) { pencil_launch_error() = "the patch walk's tables do not fit the LDS"; return; }
    auto kern = sys ? gram_patch_p2<true> : gram_patch_p2<false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static int dbg_done = 0, dbg_seen = 0;      // -DIGX_DEBUG, IGX_DEBUG_TIMING=n: the n-th patch launch of the process is stamped
    const bool dbg_t = kDebug && s.env.debug_timing && !dbg_done && ++dbg_seen >= std::max(1, atoi(getenv("IGX_DEBUG_TIMING") ? getenv("IGX_DEBUG_TIMING") : "1"));
    const size_t nwg = (size_t)(patches * pa.nseg), dbg_n = nwg * PATCH_W * 4 + nwg * 4;
    if (dbg_t) { (void)hipMalloc((void **)&pa.debug_buf, dbg_n * 8); (void)hipMemset(pa.debug_buf, 0, dbg_n * 8); }
    hipLaunchKernelGGL(kern, dim3((unsigned)(patches * pa.nseg)), dim3(PATCH_W * 64), lds, stream, S, out, A);
    if (dbg_t) {
      dbg_done = 1;
      (void)hipStreamSynchronize(stream);
      std::vector<long long> h(dbg_n);
      (void)hipMemcpy(h.data(), pa.debug_buf, dbg_n * 8, hipMemcpyDeviceToHost);
      double sm[3][3] = {{0}}; long long cnt[3] = {0}; double el = 0;
      for (size_t b = 0; b < nwg; ++b) for (int w = 0; w < PATCH_W; ++w) {
        const long long *d = &h[(b * PATCH_W + w) * 4];
        if (!d[3]) continue;
        const int k = w < 7 ? 0 : (w < PATCH_W - 1 ? 1 : 2);      // wavefronts whose every lane owns a run | partly or none | the one with the F rows
        for (int c = 0; c < 3; ++c) sm[k][c] += (double)d[c] / (double)d[3];
        cnt[k]++; el += (double)d[3];
      }
      const char *nm[3] = {"wavefronts 0-6 (64 runs each)", "wavefronts 7-10 (8 runs / none)", "wavefront 11 (the F rows)"};
      fprintf(stderr, "[igx patch timing] colour (%d,%d): %zu workgroups, seg_len %d nseg %d; cycles per element and wavefront:\n", cx, cy, nwg, pa.seg_len, pa.nseg);
      for (int k = 0; k < 3; ++k) if (cnt[k]) fprintf(stderr, "[igx patch timing]   %s: MFMAs + window adds %.0f | wait at the barrier %.0f | leave + next fetch %.0f | period %.0f\n", nm[k], sm[k][0] / cnt[k], sm[k][1] / cnt[k], sm[k][2] / cnt[k], (sm[k][0] + sm[k][1] + sm[k][2]) / cnt[k]);
      {   // the launch as a per-CU timeline on the 100 MHz clock (as launch_pencils prints it)
        const long long *wr = &h[nwg * PATCH_W * 4];
        long long t0 = LLONG_MAX, t1 = 0; std::map<long long, std::vector<std::pair<long long, long long>>> cu;
        for (size_t b = 0; b < nwg; ++b) { const long long *w = wr + b * 4; if (!w[0] || !w[1]) continue; t0 = std::min(t0, w[0]); t1 = std::max(t1, w[1]);
          const long long hw = w[2]; cu[((w[3] & 15) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back({w[0], w[1]}); }
        if (t1 > t0 && !cu.empty()) {
          double busy = 0; int rmax = 0;
          for (auto &kv : cu) { rmax = std::max(rmax, (int)kv.second.size()); for (auto &iv : kv.second) busy += (double)(iv.second - iv.first); }
          fprintf(stderr, "[igx patch timing]   timeline: span %.1f us (walks only) over %d CUs, at most %d workgroups on one CU; CU-time inside a walk %.3f of span x 256 CUs\n",
                  (t1 - t0) * 0.01, (int)cu.size(), rmax, busy / ((double)(t1 - t0) * 256.0));
        }
      }
      (void)hipFree(pa.debug_buf); pa.debug_buf = nullptr;
    }
    launches++;
  }
}
// ... of a Tangent (Jacobian / IJacobian): `kern` = state_patch_p2<Form>
typedef void (*StatePatchKernel)(SpaceDev, OutDev, PatchArgs, ParamsDev);
static void launch_state_patches_p2(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, int &launches, bool first_touch, const void *kernel, const ParamsDev &prm) {
  StatePatchKernel kern = reinterpret_cast<StatePatchKernel>(const_cast<void *>(kernel));
  const int nx = s.elem_width[1], ny = s.elem_width[2], nw = s.elem_width[0];
  const int npx = (nx + SPATCH_MX - 1) / SPATCH_MX, npy = (ny + SPATCH_MY - 1) / SPATCH_MY;
  static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
    PatchArgs A; memset(&A, 0, sizeof(A));
    A.px_start = cx; A.px_step = 2; A.px_count = (npx - cx + 1) / 2;
    A.py_start = cy; A.py_step = 2; A.py_count = (npy - cy + 1) / 2;
    A.npx = npx; A.npy = npy; { const char *d = getenv("IGX_PATCH_DBG"); A.dbg = d ? atoi(d) : 0; }      // (timing experiments, wrong results: 1 no state, 2 no leave, 4 no window adds, 8 no MFMAs)
    if (A.px_count <= 0 || A.py_count <= 0) continue;
    PencilArgs &pa = A.pa;
    pa.first_touch = first_touch ? 1 : 0;
    pa.nelx = nx; pa.nely = ny; pa.w_lo = 0; pa.w_hi = nw; pa.w_halo_lo = 0; pa.open_hi = 0; pa.wpb = SPATCH_W;
    const long long patches = (long long)A.px_count * A.py_count;
    int best = 1; long long bc = -1;
    for (int n = 1; n <= std::max(1, nw / 2); ++n) {      // (down to two elements per segment: pencil_segments)
      const int len = (nw + n - 1) / n, ns = (nw + len - 1) / len;
      if (spatch_lds_bytes(len + 3) > (size_t)160 * 1024) continue;
      const long long cost = ((patches * ns + ncu - 1) / ncu) * (len + (ns > 1 ? 2 : 0) + 1);
      if (bc < 0 || cost < bc) { bc = cost; best = n; }
    }
    if (s.env.nseg > 0) best = std::min(s.env.nseg, std::max(1, nw / 2));
    pa.seg_len = (nw + best - 1) / best; pa.nseg = (nw + pa.seg_len - 1) / pa.seg_len;
    pa.blocks_per_seg = (int)patches; pa.ne_max = pa.seg_len + 3;
    const size_t lds = spatch_lds_bytes(pa.ne_max);
    if (lds > (size_t)160 * 1024) { pencil_launch_error() = "the patch walk's tables do not fit the LDS"; return; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(patches * pa.nseg)), dim3(SPATCH_W * 64), lds, stream, S, out, A, prm);
    launches++;
  }
}
#endif

}  // namespace igx
