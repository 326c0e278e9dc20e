// block_pencil.hpp -- band rows of constant-coefficient multi-field forms (demo/Elasticity3D.c:13-46) on the matrix cores,
// one node layer at a time.
//
// The reference adds every K_e into the matrix entry by entry (IGAElementAssembleMat, src/petigaelem.c:1542-1559); for a form with
// dof fields at p = 3 that is (64 dof)^2 values per element, 288 KiB for Elasticity3D, against 25 KiB of matrix per element.  The
// feature kernel's pencil mode (feature_mfma.hpp) combines the elements of a pencil in its accumulators and writes 7/16 of that in
// lane-private 72-byte pieces; its write-out and the drain of its stores were two thirds of an element's time.
//
// This kernel turns the loop inside out.  For a form with point-independent coefficients K^{ij}_{ab} = sum_{fg} C^{ij}_{fg} M_fg[a][b],
// M_fg = sum_q JW d_f N_a d_g N_b (MAT_PAIR_MASK forms), and on the identity geometry the operands of M_fg are products of three
// 1-D rows built in registers (gram_mfma.hpp).  A pencil of elements along mesh axis 0 is processed by NODE LAYER: the band row of
// layer l -- rows (l, a1, a2), columns (l + d, b1, b2), d = -3..3 -- is the sum over the at most four elements of the pencil that
// hold both layers.  Output stationary: 16 tile products per Gram pair and layer (the sliding window of gram_pencil needs 10, but
// keeps 45 tiles alive per pencil for Elasticity and parks the lower band in 108 KB of LDS), 2304 MFMAs = exactly the
// 2 (nen dof)^2 nqp flops of SURVEY 8d per element, nothing carried from layer to layer: any cut of a pencil into segments is free.
//
// Workgroup = one pencil, 8 wavefronts in two groups that alternate layers.  Wave `role` of a group owns the band tiles
// d = {0}, {+1,-3}, {-1,+3}, {+2,-2}: four tile products each, all Gram pairs of a tile in one wave, so the coefficient transform
// C^{ij}_{fg} is lane-local.  While one group streams its MFMAs the other one
//   1. forms its 3x3 blocks, applies IGAElementFixSystem (src/petigaelem.c:1360-1389) to the combined values and deposits them in an
//      LDS stage laid out like the matrix itself: [row a][b2][b1][d][dof^2] -- for a row the 4 x 7 blocks of one b2 are contiguous
//      in the block CSR, 2016 bytes at dof 3;
//   2. adds the stage to the matrix with wave-coalesced 16-byte accesses, 126 consecutive lanes per run (first touch: plain stores).
// Colouring over the pencils' axes 1, 2 as before (16 colours); every entry of a band row is written once per pencil.
#pragma once
#ifndef IGX_RTC
#include <functional>
#include <string>
#include <vector>
#endif
#include "pencil_common.hpp"
#include "feature_mfma.hpp"

namespace igx {

struct BlockPencilArgs {
  int ex_start, ex_step, ex_count;   // pencils of this colour: local element indices on axes 1 (X) and 2 (Y)
  int ey_start, ey_step, ey_count;
  int nel0;                          // local elements on axis 0: all of them are walked
  int seg_len, nseg;                 // node layers per segment (a workgroup forms the band rows of one segment of one pencil)
  int first_touch;                   // 1: the matrix was not zeroed; the first colour to reach a block stores it
  int nelx, nely;                    // local element counts on axes 1, 2 (first-touch rule)
  int fty_lo, fty_hi, fty_blocked;   // first-touch rule on axis 2 when the assembly comes in two passes over it (gram_mfma.hpp)
  int debug;                         // -DIGX_DEBUG builds: 1 no read-add-write, 2 no MFMA, 16 no priority for the flush, 32 priority for the MFMA phase
  int dbg_block;                     // -DIGX_DEBUG builds: the workgroup whose wave 0 stamps its phases (IGX_DEBUG_FEATURE & 8)
};

typedef double bp_d2_t __attribute__((ext_vector_type(2)));   // 16-byte aligned pair (LDS side of the read-add-write)
constexpr unsigned bp_row_features(unsigned long long pairs) { unsigned m = 0; for (int f = 0; f < 8; ++f) if ((pairs >> (f * 8)) & 0xffull) m |= 1u << f; return m; }
constexpr unsigned bp_col_features(unsigned long long pairs) { unsigned m = 0; for (int f = 0; f < 8; ++f) for (int g = 0; g < 8; ++g) if ((pairs >> (f * 8 + g)) & 1ull) m |= 1u << g; return m; }

// offsets (in doubles) into the dynamic LDS block
struct BpCarve { int stage, zt, pre, cnt, rho, P, pen, vy, fcorr, bc, arrive, total; };
__host__ __device__ static inline BpCarve bp_carve(int seg_len, int dof) {
  BpCarve c; int pos = 0;
  auto take = [&](int n) { const int o = pos; pos += (n + 1) & ~1; return o; };
  c.stage = take(16 * 4 * 4 * 7 * dof * dof);
  c.zt = take((seg_len + 3) * 32);
  c.pre = take(seg_len); c.cnt = take((seg_len + 1) / 2); c.rho = take((seg_len + 1) / 2); c.P = take(seg_len * 4);
  c.pen = take(64);             // per-pencil tables of axes 1, 2 (see BpPencil)
  c.vy = take(32);
  c.fcorr = take(2 * 4 * 16 * dof);
  c.bc = take(6 + 6 * 4);
  c.arrive = take(2);            // per group: waves that have deposited so far (monotonic)
  c.total = pos;
  return c;
}

// per-pencil tables of the two non-walked axes (LDS): row prefix / row length / first column position per row slot, column
// positions of axis 2, first-touch flags per slot pair, row indices
struct BpPencil {
  long long ps1[4], ps2[4];
  int c1[4], c2[4], P1_0[4], P2[16], rmx[4], rmy[4];
  unsigned ftx, fty;           // bit a*4+b
};

// Dirichlet data of a pencil: faces in the order (axis 0 lo, hi, axis 1 lo, hi, axis 2 lo, hi); later faces override earlier
// ones (IGAElementBuildFix, src/petigaelem.c:1214-1283).  on[k] = this pencil / rank touches face k and it has values.
struct BpBC {
  bool any; int wlo, whi; bool on[6]; unsigned m[6];
  const double *v;      // LDS [6][4]
  // IGASetFixTable (src/petigaform.c:273-298): the value of a fixed dof comes from a row-indexed table [rows][dof] instead of the face's
  // constant; row of node (ix, iy, lay) = rowmap0[lay] + s1 rmx[ix] + s2 rmy[iy] (the pencil's row maps: LDS)
  const double *table; const int *rowmap0; int gw0, dof; long long s1, s2; const int *rmx, *rmy;
};
template <int P>
__device__ __forceinline__ bool bp_fixed(const BpBC &b, int ix, int iy, int lay, int i, double &val) {
  bool f = false;
  if (lay == b.wlo && ((b.m[0] >> i) & 1u)) { f = true; val = b.v[0 * 4 + i]; }
  if (lay == b.whi && ((b.m[1] >> i) & 1u)) { f = true; val = b.v[1 * 4 + i]; }
  if (b.on[2] && ix == 0 && ((b.m[2] >> i) & 1u)) { f = true; val = b.v[2 * 4 + i]; }
  if (b.on[3] && ix == P && ((b.m[3] >> i) & 1u)) { f = true; val = b.v[3 * 4 + i]; }
  if (b.on[4] && iy == 0 && ((b.m[4] >> i) & 1u)) { f = true; val = b.v[4 * 4 + i]; }
  if (b.on[5] && iy == P && ((b.m[5] >> i) & 1u)) { f = true; val = b.v[5 * 4 + i]; }
  // (a layer this rank does not hold has no node here: its entries are zero, any finite value does)
  if (f && b.table) val = (lay >= 0 && lay < b.gw0 && ix <= P && iy <= P) ? b.table[((long long)b.rowmap0[lay] + b.s1 * b.rmx[ix] + b.s2 * b.rmy[iy]) * b.dof + i] : 0.0;
  return f;
}

// one tile product: acc[pair] += A_f(element e, row slot ta)^T B_g(element e, column slot tb) over the element's 64 points;
// k-step (qw, qy), k slot = qx (lane >> 4).  zt: the element's walk-axis rows [q][a][2] (value, derivative), scaled by sqrt(w J).
// uv[qy][c]: u0 vy0, u1 vy0, u0 vy1 of this lane (axes 1, 2; scaled likewise): sqrt(JW) sits on both operands.
// accumulator sets of a band tile: the Gram pairs, and room for the dof^2 entries of the blocks they are turned into in place
template <class Form> constexpr int bp_nacc() { return fm_popcount(mat_pair_mask_of<Form>::v) > Form::DOF * Form::DOF ? fm_popcount(mat_pair_mask_of<Form>::v) : Form::DOF * Form::DOF; }

template <class Form>
__device__ __forceinline__ void bp_product(d4_t (&acc)[bp_nacc<Form>()], const double *zt, int ta, int tb, const double (&uv)[4][3]) {
  constexpr unsigned long long PAIRS = mat_pair_mask_of<Form>::v;
  constexpr unsigned FR = bp_row_features(PAIRS), FC = bp_col_features(PAIRS);
#pragma unroll
  for (int qw = 0; qw < 4; ++qw) {
    const double zA0 = zt[(qw * 4 + ta) * 2 + 0], zA1 = zt[(qw * 4 + ta) * 2 + 1];
    const double zB0 = zt[(qw * 4 + tb) * 2 + 0], zB1 = zt[(qw * 4 + tb) * 2 + 1];
#pragma unroll
    for (int qy = 0; qy < 4; ++qy) {
      double A[4], B[4];
      // feature 0: N, 1: d/dx0 (walk axis), 2: d/dx1, 3: d/dx2
      if (FR & 1u) A[0] = zA0 * uv[qy][0];
      if (FR & 2u) A[1] = zA1 * uv[qy][0];
      if (FR & 4u) A[2] = zA0 * uv[qy][1];
      if (FR & 8u) A[3] = zA0 * uv[qy][2];
      if (FC & 1u) B[0] = zB0 * uv[qy][0];
      if (FC & 2u) B[1] = zB1 * uv[qy][0];
      if (FC & 4u) B[2] = zB0 * uv[qy][1];
      if (FC & 8u) B[3] = zB0 * uv[qy][2];
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (!((PAIRS >> (f * 8 + g)) & 1ull)) continue;
          acc[fm_pair_index(PAIRS, f, g)] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[f], B[g], acc[fm_pair_index(PAIRS, f, g)], 0, 0, 0);
        }
    }
  }
}

// Gram sums -> blocks, in place: K^{ij} = sum_{fg} C^{ij}_{fg} M_fg with C = mat(e_f, e_g) (the form's constants).  Done by the
// wave that holds the accumulators, at the end of its own MFMA phase: fp64 VALU work of the OTHER wave of the SIMD waits for a
// gap in this wave's MFMA stream, about one MFMA (64 cycles) per instruction -- in the flush phase these 21 multiply-adds per
// block took 11k cycles per layer and kept the MFMA group waiting at the deposit barrier.
template <class Form>
__device__ __forceinline__ void bp_transform(d4_t (&acc)[bp_nacc<Form>()], const PtView &p0) {
  constexpr unsigned long long PAIRS = mat_pair_mask_of<Form>::v;
  constexpr int NP = fm_popcount(PAIRS), BS = Form::DOF * Form::DOF;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double M[NP], K[BS];
#pragma unroll
    for (int n = 0; n < NP; ++n) M[n] = acc[n][r];
#pragma unroll
    for (int n = 0; n < BS; ++n) K[n] = 0.0;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (!((PAIRS >> (f * 8 + g)) & 1ull)) continue;
        double ef[4], eg[4], T[BS];
#pragma unroll
        for (int n = 0; n < 4; ++n) { ef[n] = (n == f) ? 1.0 : 0.0; eg[n] = (n == g) ? 1.0 : 0.0; }
        Form::mat(p0, ef, eg, T);
#pragma unroll
        for (int n = 0; n < BS; ++n) if ((fm_pair_block_mask<Form>(f, g) >> n) & 1u) K[n] += T[n] * M[fm_pair_index(PAIRS, f, g)];
      }
#pragma unroll
    for (int n = 0; n < BS; ++n) {
      // (pinned here: left to itself the compiler sinks these multiply-adds past the barrier into the deposit, their only user)
      asm volatile("" : "+v"(K[n]));
      acc[n][r] = K[n];
    }
  }
}

template <class Form, int P, bool SYSTEM>
__global__ void __launch_bounds__(512, 2)
block_pencil(SpaceDev S, ParamsDev prm, OutDev out, BlockPencilArgs pa) {
  static_assert(P == 3, "4 x 4 x 4 basis functions: the band tiles of a layer split evenly over four wavefronts");
  constexpr int NB = P + 1, BW = 2 * P + 1, DOF = Form::DOF, BS = DOF * DOF;
  constexpr unsigned long long PAIRS = mat_pair_mask_of<Form>::v;
  constexpr int NACC = bp_nacc<Form>();
  static_assert(PAIRS != 0ull && (PAIRS >> 32) == 0ull && ((PAIRS >> 4) & 0x0f0f0f0full) == 0ull, "Gram pairs over N and grad N");
  static_assert(DOF <= 3, "the LDS stage holds one band row: 16 x 16 x 7 blocks of dof^2 values");
  extern __shared__ __attribute__((aligned(16))) double bp_sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, role = wave & 3;
  const int npen = pa.ex_count * pa.ey_count;
  const int seg = blockIdx.x / npen, pencil = blockIdx.x - seg * npen;
  const int tx = pencil % pa.ex_count, ty = pencil / pa.ex_count;
  const int elx = pa.ex_start + tx * pa.ex_step, ely = pa.ey_start + ty * pa.ey_step;
  const AxisDev &AW = S.ax[0], &AX = S.ax[1], &AY = S.ax[2];
  const int NL = pa.nel0 + P;                       // node layers the rank's elements touch, li = 0 .. NL-1
  const int li_lo = seg * pa.seg_len, li_hi = min(li_lo + pa.seg_len, NL);
  const int nlay = li_hi - li_lo;
  const int e0 = max(li_lo - P, 0), e1 = min(li_hi - 1, pa.nel0 - 1);   // elements whose rows this segment reads
  const int ne = e1 - e0 + 1;
  const int lay_first = AW.off[0];                  // ghost-local index of layer li = 0
  const int offx = AX.off[elx], offy = AY.off[ely];

  const BpCarve cv = bp_carve(pa.seg_len, DOF);
  double *stage = bp_sm + cv.stage, *zts = bp_sm + cv.zt, *vys = bp_sm + cv.vy, *fcorr = bp_sm + cv.fcorr + grp * (4 * 16 * DOF);
  long long *Lpre = reinterpret_cast<long long *>(bp_sm + cv.pre);
  int *Lcnt = reinterpret_cast<int *>(bp_sm + cv.cnt), *Lrho = reinterpret_cast<int *>(bp_sm + cv.rho), *LP = reinterpret_cast<int *>(bp_sm + cv.P);
  BpPencil *pen = reinterpret_cast<BpPencil *>(bp_sm + cv.pen);
  unsigned *bcm = reinterpret_cast<unsigned *>(bp_sm + cv.bc);      // [6] field masks (0: face not touched), then 6 x 4 values
  double *bcv = bp_sm + cv.bc + 6;
  int *arrive = reinterpret_cast<int *>(bp_sm + cv.arrive) + grp * 2;
  if (tid < 4) reinterpret_cast<int *>(bp_sm + cv.arrive)[tid] = 0;

  // ---- stage the tables of the segment and of the pencil
  {
    const double *__restrict__ tabw = AW.tab + (size_t)e0 * (NB * NB * NDER);
    for (int i = tid; i < ne * 32; i += 512) {      // i = e*32 + (q*4 + a)*2 + k; rows scaled by sqrt(w_q J_e)
      const int e = i >> 5, j = i & 31, q = j >> 3, aa = (j >> 1) & 3, k = j & 1;
      zts[i] = tabw[((size_t)e * NB * NB + q * NB + aa) * NDER + k] * sqrt(AW.w[(e0 + e) * NB + q] * AW.J[e0 + e]);
    }
    for (int i = tid; i < nlay; i += 512) {
      const int lay = lay_first + li_lo + i, rho = AW.rowmap[lay];
      Lrho[i] = rho; Lcnt[i] = AW.rcnt[rho]; Lpre[i] = AW.prefix[rho];
      for (int d = 0; d < BW; ++d) LP[i * 8 + d] = AW.P[lay * BW + d];
    }
    if (tid < 4) {
      const int a = tid;
      const int ixg = offx + a, rhox = AX.rowmap[ixg], iyg = offy + a, rhoy = AY.rowmap[iyg];
      pen->ps1[a] = AX.prefix[rhox]; pen->c1[a] = AX.rcnt[rhox]; pen->P1_0[a] = AX.P[ixg * BW + (0 - a + P)]; pen->rmx[a] = rhox;
      pen->ps2[a] = AY.prefix[rhoy]; pen->c2[a] = AY.rcnt[rhoy]; pen->rmy[a] = rhoy;
      for (int b = 0; b < 4; ++b) pen->P2[a * 4 + b] = AY.P[iyg * BW + (b - a + P)];
    }
    if (tid == 64) {
      unsigned fx = 0, fy = 0;
      if (pa.first_touch)
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
          if (first_touch_axis<P>(elx, a, b, pa.nelx)) fx |= 1u << (a * 4 + b);
          if (first_touch_axis<P>(ely, a, b, pa.nely, pa.fty_lo, pa.fty_hi, pa.fty_blocked)) fy |= 1u << (a * 4 + b);
        }
      pen->ftx = fx; pen->fty = fy;
    }
    if (tid >= 128 && tid < 160) {   // axis-2 rows of this pencil [a][q][2], scaled by sqrt(w J)
      const int l2 = tid - 128, aa = l2 >> 3, qq = (l2 >> 1) & 3, kk = l2 & 1;
      vys[l2] = AY.tab[((size_t)ely * NB * NB + qq * NB + aa) * NDER + kk] * sqrt(AY.w[ely * NB + qq] * AY.J[ely]);
    }
    if (tid >= 192 && tid < 198) {   // Dirichlet faces this pencil can touch (System driver only)
      const int k = tid - 192, d = k >> 1, sd = k & 1;
      const AxisDev &A = S.ax[d];
      const int el = d == 0 ? 0 : (d == 1 ? elx : ely);   // (axis 0: the rank's first / last element decides)
      bool on = SYSTEM && !A.periodic && S.bcv[d][sd].count > 0;
      if (on) on = (d == 0) ? (sd == 0 ? A.estart == 0 : A.estart + A.nel == A.esizes) : (sd == 0 ? el + A.estart == 0 : el + A.estart == A.esizes - 1);
      unsigned m = 0;
      for (int c = 0; c < 4; ++c) bcv[k * 4 + c] = 0.0;
      if (on) for (int c = 0; c < S.bcv[d][sd].count; ++c) { const int fld = S.bcv[d][sd].field[c]; if (fld < DOF) { m |= 1u << fld; bcv[k * 4 + fld] = S.bcv[d][sd].value[c]; } }
      bcm[k] = m;
    }
  }
  __syncthreads();

  BpBC bc; bc.any = false; bc.v = bcv;
#pragma unroll
  for (int k = 0; k < 6; ++k) { bc.m[k] = SYSTEM ? (unsigned)__builtin_amdgcn_readfirstlane((int)bcm[k]) : 0u; bc.on[k] = bc.m[k] != 0u; bc.any = bc.any || bc.on[k]; }
  bc.wlo = bc.on[0] ? lay_first : -1000;
  bc.whi = bc.on[1] ? lay_first + NL - 1 : -1000;
  bc.table = SYSTEM ? S.fixtable : nullptr; bc.rowmap0 = AW.rowmap; bc.gw0 = AW.gwidth; bc.dof = DOF;
  bc.s1 = (long long)S.ax[0].nrow; bc.s2 = (long long)S.ax[0].nrow * S.ax[1].nrow; bc.rmx = pen->rmx; bc.rmy = pen->rmy;

  // ---- per-lane operand factors of axes 1, 2: k slot qx = lane >> 4, tile slot (ix, iy) = (lane & 3, (lane >> 2) & 3)
  double uv[4][3];
  {
    const int qx = lane >> 4, ix = lane & 3, iy = (lane >> 2) & 3;
    const double sx = sqrt(AX.w[elx * NB + qx] * AX.J[elx]);
    const double u0 = AX.tab[((size_t)elx * NB * NB + qx * NB + ix) * NDER + 0] * sx, u1 = AX.tab[((size_t)elx * NB * NB + qx * NB + ix) * NDER + 1] * sx;
#pragma unroll
    for (int qy = 0; qy < 4; ++qy) {
      const double vy0 = vys[(iy * 4 + qy) * 2 + 0], vy1 = vys[(iy * 4 + qy) * 2 + 1];
      uv[qy][0] = u0 * vy0; uv[qy][1] = u1 * vy0; uv[qy][2] = u0 * vy1;
    }
  }
  // band tiles of this wave: slot A and slot B (role 0 has one)
  const int dA = (role == 0) ? 0 : (role == 1 ? 1 : (role == 2 ? -1 : 2));
  const int dB = (role == 1) ? -3 : (role == 2 ? 3 : -2);
  const bool hasB = role != 0;
  const long long T0 = S.ax[0].tot, T10 = S.ax[1].tot * S.ax[0].tot;
  const long long rsx = (long long)S.ax[0].nrow, rsy = (long long)S.ax[0].nrow * S.ax[1].nrow;
  // read-add-write runs of this wave: rows a2 = role, run ru = (a1, b2); lane ru holds what does not depend on the layer:
  //   pos = ps2 T1 T0 + c2 (ps1 T0 + c1 ps0) + (P2 c1 + P1) c0      (block position of the run's first block)
  long long run_base, run_cc; int run_pp;
  {
    const int ru = lane & 15, a1 = ru >> 2, b2 = ru & 3;
    const long long c1 = pen->c1[a1], c2 = pen->c2[role];
    run_base = pen->ps2[role] * T10 + c2 * (pen->ps1[a1] * T0);
    run_cc = c2 * c1;
    run_pp = (int)(pen->P2[role * 4 + b2] * c1 + pen->P1_0[a1]);
  }
  const unsigned ftx = (unsigned)__builtin_amdgcn_readfirstlane((int)pen->ftx), fty = (unsigned)__builtin_amdgcn_readfirstlane((int)pen->fty);
  const int rmy_role = __builtin_amdgcn_readfirstlane(pen->rmy[role]);
  PtView p0; p0.x = nullptr; p0.u = nullptr; p0.ut = nullptr; p0.gu = nullptr; p0.hu = nullptr; p0.G = nullptr; p0.prm = prm.v; p0.shift = out.shift; p0.t = out.t; p0.normal = nullptr; p0.atboundary = 0; p0.boundary_id = -1;

  const int nit = (nlay + 1) >> 1;
  // -DIGX_DEBUG, IGX_DEBUG_FEATURE & 8: cycles per phase of wave 0 of one workgroup, summed over its iterations
  const bool stamp = kDebug && out.dbg && (int)blockIdx.x == pa.dbg_block && tid == 0;
  long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define BP_STAMP(k) do { if (stamp) { const long long t_ = (long long)__builtin_readcyclecounter(); ph[k] += t_ - tlast; tlast = t_; } } while (0)
  if (grp == 1) __builtin_amdgcn_s_barrier();
  if (stamp) tlast = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < nit; ++it) {
    const int k = 2 * it + grp;
    const bool act = k < nlay;
    const int li = li_lo + (act ? k : 0);       // layer of this group in this iteration
    const int lay = lay_first + li;
    d4_t accA[NACC], accB[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) { accA[n] = (d4_t){0, 0, 0, 0}; accB[n] = (d4_t){0, 0, 0, 0}; }

    // ---- MFMA phase: the tile products of this wave's band tiles; elements [max(l, l+d) - P, min(l, l+d)] of the pencil
    if (kDebug && (pa.debug & 32)) __builtin_amdgcn_s_setprio(2);
    {
      int eloA = max(li, li + dA) - P, ehiA = min(li, li + dA);
      if (eloA < 0) eloA = 0;
      if (ehiA > pa.nel0 - 1) ehiA = pa.nel0 - 1;
      if (!act || (kDebug && (pa.debug & 2))) ehiA = eloA - 1;
      for (int e = eloA; e <= ehiA; ++e) bp_product<Form>(accA, zts + (e - e0) * 32, li - e, li + dA - e, uv);
      BP_STAMP(0); BP_STAMP(1);
      bp_transform<Form>(accA, p0);
      if (hasB) {
        int eloB = max(li, li + dB) - P, ehiB = min(li, li + dB);
        if (eloB < 0) eloB = 0;
        if (ehiB > pa.nel0 - 1) ehiB = pa.nel0 - 1;
        if (!act || (kDebug && (pa.debug & 2))) ehiB = eloB - 1;
        for (int e = eloB; e <= ehiB; ++e) bp_product<Form>(accB, zts + (e - e0) * 32, li - e, li + dB - e, uv);
        bp_transform<Form>(accB, p0);
      }
    }
    BP_STAMP(2);
    __builtin_amdgcn_s_barrier();
    BP_STAMP(3);

    // ---- deposit: blocks K^{ij} = sum_{fg} C^{ij}_{fg} M_fg of this lane's (row slot, column slot) pairs, IGAElementFixSystem on
    // the combined values, into the stage in matrix order
    if (!(kDebug && (pa.debug & 16))) __builtin_amdgcn_s_setprio(3);
    const int kk = act ? k : 0;
    const int c0 = __builtin_amdgcn_readfirstlane(Lcnt[kk]);
    const int held = min(li, pa.nel0 - 1) - max(li - P, 0) + 1;      // elements of this pencil that hold the layer
    const bool bcrow = SYSTEM && bc.any && (bc.on[2] || bc.on[3] || bc.on[4] || bc.on[5] || (lay >= bc.wlo - P && lay <= bc.wlo + P) || (lay >= bc.whi - P && lay <= bc.whi + P));
    {
      const int a1 = lane >> 4, b1 = lane & 3, b2 = (lane >> 2) & 3;
      double corr[4][DOF];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < DOF; ++i) corr[r][i] = 0.0;
      auto deposit = [&](const d4_t (&acc)[NACC], int d) {
        const int p0d = __builtin_amdgcn_readfirstlane(LP[kk * 8 + d + P]);
        if (p0d < 0 || !act) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {       // a2 = r
          double K[BS];
#pragma unroll
          for (int n = 0; n < BS; ++n) K[n] = acc[n][r];
          if (bcrow) {
            bool fa[DOF], fb[DOF]; double vb[DOF];
#pragma unroll
            for (int i = 0; i < DOF; ++i) { double va = 0; fa[i] = bp_fixed<P>(bc, a1, r, lay, i, va); vb[i] = 0; fb[i] = bp_fixed<P>(bc, b1, b2, lay + d, i, vb[i]); }
#pragma unroll
            for (int i = 0; i < DOF; ++i)
#pragma unroll
              for (int j = 0; j < DOF; ++j) {
                if (fb[j]) corr[r][i] += K[i * DOF + j] * vb[j];
                if (fa[i] || fb[j]) K[i * DOF + j] = (d == 0 && a1 == b1 && r == b2 && i == j) ? (double)held : 0.0;
              }
          }
          double *sp = stage + ((size_t)(((a1 + 4 * r) * 4 + b2) * 4 + b1) * c0 + p0d) * BS;
#pragma unroll
          for (int n = 0; n < BS; ++n) sp[n] = K[n];
        }
      };
      BP_STAMP(8);       // (set-up of the phase: layer data, Dirichlet flags)
      deposit(accA, dA);
      if (hasB) deposit(accB, dB);
      BP_STAMP(9);       // (the deposits)
      if (bcrow) {       // F_i -= sum over fixed columns of K_ik v_k: sum over the 16 column slots of this wave's tiles, then over the waves
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < DOF; ++i) {
            double c = corr[r][i];
            c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4); c += __shfl_xor(c, 8);
            if ((lane & 15) == 0) fcorr[(role * 16 + a1 + 4 * r) * DOF + i] = c;
          }
      }
    }

    // This wave's part of the band row is in the stage: tell the other three waves of the group.  Not a barrier: s_barrier is
    // workgroup wide, and the other group's MFMA waves would stand at it while this group's loads below push through the memory
    // pipeline (measured: 7k-11k cycles of every 45k-cycle layer).  A counter in LDS per group, polled after the loads are out.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    BP_STAMP(4);
    // ---- read-add-write of the band row: wave `role` takes the rows a2 = role; run (a1, b2) = 4 c0 blocks, contiguous in the
    // stage and in the matrix.  With 121 KB of reads per layer the memory pipeline pushes back on the issue itself (measured: 11k
    // cycles to get the 32 loads of a lane out): they are requested before the wave waits for the other deposits.
    const int runlen = 4 * c0 * BS;                 // doubles per run (even)
    double *gp[16]; d2u_t oldv[16][2]; bool ldm[16];
    const long long ps0 = ((long long)__builtin_amdgcn_readfirstlane((int)(Lpre[kk] >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)(Lpre[kk] & 0xffffffffll));
    const bool dowrite = act && !(kDebug && (pa.debug & 1));
    const long long mypos = run_base + run_cc * ps0 + (long long)run_pp * c0;
    BP_STAMP(10);        // (the Dirichlet sums)
#pragma unroll
    for (int ru = 0; ru < 16; ++ru) {
      const int a1 = ru >> 2, b2 = ru & 3, a2 = role;
      const long long pos = ((long long)__builtin_amdgcn_readlane((int)(mypos >> 32), ru) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)(mypos & 0xffffffffll), ru);
      gp[ru] = out.val + pos * BS;
      // first touch: every block of the run is a first touch -> nothing to read
      const bool ally = (fty >> (a2 * 4 + b2)) & 1u, allx = ((ftx >> (a1 * 4)) & 0xfu) == 0xfu;
      ldm[ru] = dowrite && !(ally && allx);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = 2 * (lane + 64 * h);
        oldv[ru][h] = (d2u_t){0.0, 0.0};
        if (ldm[ru] && o < runlen) oldv[ru][h] = *reinterpret_cast<const d2u_t *>(gp[ru] + o);
      }
    }
    {   // every wave of this group has deposited
      const int target = 4 * (it + 1);
      while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
      asm volatile("" ::: "memory");
    }
    BP_STAMP(5);
    if (dowrite) {
      const int blk = c0 * BS;        // doubles per b1 group
#pragma unroll
      for (int ru = 0; ru < 16; ++ru) {
        const int a1 = ru >> 2, b2 = ru & 3, a2 = role;
        const double *sp = stage + (size_t)((a1 + 4 * a2) * 4 + b2) * runlen;
        const bool fy1 = (fty >> (a2 * 4 + b2)) & 1u;
        const unsigned fxm = fy1 ? ((ftx >> (a1 * 4)) & 0xfu) : 0u;      // b1 groups that are stored, not added to
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int o = 2 * (lane + 64 * h);
          if (o < runlen) {
            const bp_d2_t nv = *reinterpret_cast<const bp_d2_t *>(sp + o);      // (16-byte aligned: run lengths are even)
            d2u_t ov = oldv[ru][h];
            if (fxm) {       // per value: the b1 group it belongs to
              const int g0 = (o >= blk) + (o >= 2 * blk) + (o >= 3 * blk), g1 = (o + 1 >= blk) + (o + 1 >= 2 * blk) + (o + 1 >= 3 * blk);
              if ((fxm >> g0) & 1u) ov[0] = 0.0;
              if ((fxm >> g1) & 1u) ov[1] = 0.0;
            }
            d2u_t w; w[0] = ov[0] + nv[0]; w[1] = ov[1] + nv[1];
            *reinterpret_cast<d2u_t *>(gp[ru] + o) = w;
          }
        }
      }
      if (bcrow && lane < 4 * DOF) {     // F of the rows (a1, a2 = role): fixed rows hold value x multiplicity, the others the lifting
        const int a1 = lane / DOF, i = lane - a1 * DOF;
        double v = 0; const bool fx = bp_fixed<P>(bc, a1, role, lay, i, v);
        double c = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) c += fcorr[(w * 16 + a1 + 4 * role) * DOF + i];
        const double Fv = fx ? v * (double)held : -c;
        if (Fv != 0.0) {
          const long long frow = (long long)Lrho[kk] + rsx * pen->rmx[a1] + rsy * rmy_role;
          out.vec[frow * DOF + i] += Fv;
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    BP_STAMP(6);
    __builtin_amdgcn_s_barrier();
    BP_STAMP(7);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  if (stamp) {     // cumulative, in the layout engine.hip prints ("[feature stamps]": differences of consecutive entries)
    long long c = 0; out.dbg[0] = 0;
    for (int k = 0; k < 12; ++k) { c += ph[k] / (nit > 0 ? nit : 1); out.dbg[k + 1] = c; }
    out.dbg[31] = 13;
  }
#undef BP_STAMP
}

#ifndef IGX_RTC
// ---- boundary loads of a multi-field form on the identity geometry: gram_mfma.hpp's k_boundary_loads per field
// (IGAElementBuildFix: AddFlux, src/petigaelem.c:1191-1212; a Dirichlet value on the same dof discards the flux, :1371-1387)
static __global__ void k_boundary_loads_field(FluxArgs F, int nr0, int nr1, int dof, int field, double *vec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F.nt * F.nu) return;
  int r[3]; r[F.d] = F.rd; r[F.t] = i % F.nt; r[F.u] = i / F.nt;
  for (int a = 0; a < 3; ++a) {
    const int gi = F.gfirst[a] + r[a];
    if ((F.fixlo[a] && gi == 0) || (F.fixhi[a] && gi == F.glast[a])) return;
  }
  const double v = F.value * F.st[r[F.t]] * F.su[r[F.u]];
  if (v != 0.0) vec[((size_t)r[0] + (size_t)nr0 * ((size_t)r[1] + (size_t)nr1 * (size_t)r[2])) * dof + field] += v;
}

static int launch_boundary_loads_fields(const Space &s, const OutDev &out, hipStream_t stream, std::string &err) {
  for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) {
    const BC &bl = s.load[d][sd];
    if (!bl.count || s.axis[d].periodic) continue;
    if (sd == 0 ? s.elem_start[d] != 0 : s.elem_start[d] + s.elem_width[d] != s.elem_sizes[d]) continue;
    const int t = (d + 1) % 3, u = (d + 2) % 3;
    std::vector<double> sum[2];
    const int ax2[2] = {t, u};
    for (int k = 0; k < 2; ++k) {
      const int a = ax2[k]; const AxisLayout &L = s.lay[a]; const Basis1D &b = s.basis[a];
      sum[k].assign((size_t)L.nrow, 0.0);
      for (int e = 0; e < s.elem_width[a]; ++e) {
        const int ge = s.elem_start[a] + e, first = b.offset[ge] - L.gstart;
        for (int j = 0; j < b.nen; ++j) { const int i = first + j; if (i >= 0 && i < L.nrow) sum[k][(size_t)i] += b.detJac[ge] / (double)b.nen; }
      }
    }
    double *dt = nullptr;
    const size_t bytes = (sum[0].size() + sum[1].size()) * sizeof(double);
    if (pool_alloc(reinterpret_cast<void **>(&dt), bytes, stream) != hipSuccess) { err = "device allocation of the boundary-load sums failed"; return IGX_ERR_MEM; }
    (void)hipMemcpyAsync(dt, sum[0].data(), sum[0].size() * sizeof(double), hipMemcpyHostToDevice, stream);
    (void)hipMemcpyAsync(dt + sum[0].size(), sum[1].data(), sum[1].size() * sizeof(double), hipMemcpyHostToDevice, stream);
    (void)hipStreamSynchronize(stream);
    for (int field = 0; field < s.dof; ++field) {
      double load = 0; bool any = false;
      for (int k = 0; k < bl.count; ++k) if (bl.field[k] == field) { load += bl.value[k]; any = true; }
      if (!any) continue;
      FluxArgs F; F.d = d; F.t = t; F.u = u; F.value = load * 4.0;
      F.rd = sd == 0 ? 0 : s.axis[d].nnp - 1 - s.lay[d].gstart;
      F.nt = s.lay[t].nrow; F.nu = s.lay[u].nrow;
      for (int a = 0; a < 3; ++a) {
        F.gfirst[a] = s.lay[a].gstart; F.glast[a] = s.axis[a].nnp - 1;
        auto holds = [&](const BC &bv) { for (int k = 0; k < bv.count; ++k) if (bv.field[k] == field) return 1; return 0; };
        F.fixlo[a] = s.axis[a].periodic ? 0 : holds(s.value[a][0]); F.fixhi[a] = s.axis[a].periodic ? 0 : holds(s.value[a][1]);
      }
      F.st = dt; F.su = dt + sum[0].size();
      const int n = F.nt * F.nu;
      hipLaunchKernelGGL(k_boundary_loads_field, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, F, s.lay[0].nrow, s.lay[1].nrow, s.dof, field, out.vec);
    }
    (void)hipFreeAsync(dt, stream);
  }
  return 0;
}

// Does this kernel cover the case?  3-D, p = 3 with 4 Gauss points on every axis, identity geometry, System / Matrix driver of a
// constant-coefficient form with F = 0, axis 0 walkable (one new node layer per element, not wrapped inside the rank), axes 1, 2
// with consecutive column positions (not wrapped inside the rank).
template <class Form> constexpr bool bp_form_ok() {
  return mat_pair_mask_of<Form>::v != 0ull && Form::DOF <= 3 && Form::DOF >= 2 && shape_order_of<Form>::v < 2 && Form::ORDER < 2 &&
         !has_boundary_of<Form>::v && nscalar_of<Form>::v == 0;
}
static bool block_pencil_covers_space(const Space &s, const SpaceDev &S, const OutDev &out, int dof) {
  if (s.env.block_pencil == 0 || s.env.block_pencil == 2) return false;      // (2: band_pt only -- an experiment switch: the identity geometry through the point-record kernel)
  if (out.op != OP_SYSTEM && out.op != OP_MATRIX) return false;
  if (s.dim != 3 || s.dof != dof || s.nsd != 0) return false;      // (a fix table is read in the fix-up: bp_fixed)
  for (int d = 0; d < 3; ++d) {
    if (s.axis[d].p != 3 || s.basis[d].nqp != 4 || s.basis[d].nen != 4 || s.lay[d].alias) return false;
    for (int sd = 0; sd < 2; ++sd) if (s.visit[d][sd]) return false;
  }
  if (!axis_walkable(s, 0)) return false;
  return true;
}
template <class Form>
static bool block_pencil_covers(const Space &s, const SpaceDev &S, const OutDev &out) {
  if constexpr (!bp_form_ok<Form>()) return false;
  else return block_pencil_covers_space(s, S, out, Form::DOF);
}

// the launches of an assembly; `launch(sys, grid, lds_bytes, args)` starts the kernel -- the compiled-in instantiation of a built-in
// form, or the module function of a run-time struct (rtc.hpp)
typedef std::function<void(bool, unsigned, size_t, const BlockPencilArgs &)> BlockPencilLaunch;
static int block_pencil_run(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, std::string &kname, int &launches, std::string &err, bool &done, DomInfo &dom,
                            const std::function<void()> &zero_matrix, const std::function<void()> &slab_done, int DOF, int NP, const BlockPencilLaunch &launch);

template <class Form>
static int try_block_pencil(const Space &s, const SpaceDev &S, const ParamsDev &prm, const OutDev &out, hipStream_t stream, std::string &kname, int &launches,
                            std::string &err, bool &done, DomInfo &dom, const std::function<void()> &zero_matrix, const std::function<void()> &slab_done) {
  done = false;
  if constexpr (!bp_form_ok<Form>()) return 0;
  else {
  if (!block_pencil_covers<Form>(s, S, out)) return 0;
  if constexpr (!vec_zero_of<Form>::v) {
    // F != 0 (a body force): the form's own vec() comes from a vector-only pass ahead of the band rows -- sum factorisation, 64 evaluations
    // of vec() per element -- with the fixed rows left at zero; the band-row kernel then ADDS the Dirichlet lifting and value x
    // multiplicity of IGAElementFixSystem as it does for F = 0 (src/petigaelem.c:1377-1387 is linear in F_e).
    if (out.op == OP_SYSTEM) {
#ifdef IGX_HAVE_VEC_SUMFACT
      OutDev ov = out; ov.vec_mode = 2; ov.op = OP_VECTOR; ov.val = nullptr; ov.browptr = nullptr;
      bool vdone = false; int vl = 0; std::string vk;
      if (int rc = try_vec_sumfact<Form>(s, S, prm, ov, stream, vk, vl, vdone)) { err = "vec_sumfact kernel launch failed"; return rc; }
      if (!vdone) return 0;
#else
      return 0;
#endif
    }
  }
  return block_pencil_run(s, S, out, stream, kname, launches, err, done, dom, zero_matrix, slab_done, Form::DOF, fm_popcount(mat_pair_mask_of<Form>::v),
                          [&](bool sys, unsigned grid, size_t lds, const BlockPencilArgs &pa) {
                            auto kern = sys ? block_pencil<Form, 3, true> : block_pencil<Form, 3, false>;
                            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, S, prm, out, pa);
                          });
  }
}

static int block_pencil_run(const Space &s, const SpaceDev &S, const OutDev &out, hipStream_t stream, std::string &kname, int &launches, std::string &err, bool &done, DomInfo &dom,
                            const std::function<void()> &zero_matrix, const std::function<void()> &slab_done, int DOF, int NP, const BlockPencilLaunch &launch) {
  {
  constexpr int P = 3;
  const bool sys = out.op == OP_SYSTEM;
  const bool first_touch = !s.env.no_first_touch && out.val && axis_first_touch_ok(s, 1) && axis_first_touch_ok(s, 2);
  if (!first_touch) { if (zero_matrix) zero_matrix(); }
  else if (s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] > 1) zero_neighbour_rows(s, out, stream);
  if (sys) { if (int rc = launch_boundary_loads_fields(s, out, stream, err)) return rc; }
  launches = 0;
  static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  const int NL = s.elem_width[0] + P;
  auto run = [&](const Box &bx, const int *fty) {
    for (int d = 1; d < 3; ++d) if (bx.hi[d] <= bx.lo[d]) return;
    for (int cy = 0; cy < s.lay[2].ncolors; ++cy) for (int cx = 0; cx < s.lay[1].ncolors; ++cx) {
      BlockPencilArgs pa; memset(&pa, 0, sizeof(pa));
      pa.first_touch = first_touch ? 1 : 0; pa.nelx = s.elem_width[1]; pa.nely = s.elem_width[2];
      pa.fty_lo = fty ? fty[0] : 0; pa.fty_hi = fty ? fty[1] : 0x7fffffff; pa.fty_blocked = fty ? fty[2] : 0x7fffffff;
      if (!color_range(s.lay[1], cx, bx.lo[1], bx.hi[1], pa.ex_start, pa.ex_step, pa.ex_count)) continue;
      if (!color_range(s.lay[2], cy, bx.lo[2], bx.hi[2], pa.ey_start, pa.ey_step, pa.ey_count)) continue;
      pa.nel0 = s.elem_width[0];
      const long long pencils = (long long)pa.ex_count * pa.ey_count;
      // Segments cost nothing but their table staging and one idle half period: enough of them to fill the CUs a few times
      // over, at least 8 layers each, at most what the LDS holds next to the stage (zt: 256 bytes per element)
      const int max_len = 64;
      int nseg = (NL + max_len - 1) / max_len;
      {   // one workgroup per CU: the count with the fewest rounds x (layers + 1)
        long long best = -1; int best_n = nseg;
        for (int n = nseg; n <= std::max(nseg, NL / 8); ++n) {
          const int len = (NL + n - 1) / n, ns = (NL + len - 1) / len;
          const long long cost = ((pencils * ns + ncu - 1) / ncu) * (len + 1);
          if (best < 0 || cost < best) { best = cost; best_n = n; }
        }
        nseg = best_n;
      }
      if (s.env.nseg > 0) nseg = std::max((NL + max_len - 1) / max_len, std::min(s.env.nseg, std::max(1, NL / 2)));
      pa.seg_len = (NL + nseg - 1) / nseg; pa.nseg = (NL + pa.seg_len - 1) / pa.seg_len;
      pa.debug = s.env.debug_feature; pa.dbg_block = 7 + s.env.debug_noflush;
      const size_t lds = (size_t)bp_carve(pa.seg_len, DOF).total * sizeof(double);
      launch(sys, (unsigned)(pencils * pa.nseg), lds, pa);
      launches++;
    }
  };
  Box all; for (int d = 0; d < 3; ++d) { all.lo[d] = 0; all.hi[d] = s.elem_width[d]; }
  if (dom.ev0) (void)hipEventRecord(dom.ev0, stream);
  // several ranks on axis 2: the pencils within p elements of its upper face first, then the mark for the exchange, then the rest
  const int n2 = s.elem_width[2];
  const bool upper2 = s.proc_sizes[2] > 1 && (s.proc_ranks[2] < s.proc_sizes[2] - 1 || s.axis[2].periodic);
  if (slab_done && upper2 && n2 >= 2 * (P + 1)) {
    const int c2 = face_cut(n2, P);      // (a thick pass: pencil_common.hpp)
    Box top = all, rest = all; top.lo[2] = c2; rest.hi[2] = c2;
    const int ft_top[3] = {c2, n2, 0x7fffffff}, ft_rest[3] = {0, c2, c2};
    run(top, ft_top);
    slab_done();
    run(rest, ft_rest);
  } else run(all, nullptr);
  if (dom.ev1) (void)hipEventRecord(dom.ev1, stream);
  if (hipGetLastError() != hipSuccess) { err = "block pencil kernel launch failed"; return IGX_ERR_LIB; }
  dom.name = "block_pencil<p=3>"; dom.launches = launches;
  dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2];
  dom.flop_per_element = 2048.0 * NP * 16 * 16;      // 16 tile products of 16 k-steps per Gram pair and layer
  kname = std::string("block_pencil(mfma_f64_16x16x4,p=3,dof=") + char('0' + DOF) + ",band rows by node layer)";
  done = true;
  return 0;
  }
}

#endif   // !IGX_RTC

}  // namespace igx
