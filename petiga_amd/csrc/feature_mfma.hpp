// feature_mfma.hpp -- the general element kernel with the dense K_e contraction on the matrix cores.
//
// Every point callback of the reference that fills a matrix (System / Matrix / Jacobian / IJacobian,
// include/petiga.h:153-197) is BILINEAR in the test function a and the trial function b: with the feature vector
// Phi(a,q) = (N, dN/dx_i, d2N/dx_i dx_j) of a basis function at point q,
//     k_q[(a,i),(b,j)] = sum_f Phi_f(a,q) * mat(e_f, Phi(b,q))[i][j]                         (e_f = unit feature),
// so the reduction over points of rows 11-13 of SURVEY 8(a) (IGAPointAddMat, src/petigapoint.c:451-492) is the GEMM
//     K_e[(.,i),(.,j)] = A^T B^{ij},   A[(q,f)][a] = Phi_f(a,q),   B^{ij}[(q,f)][b] = JW_q * mat(e_f, Phi(b,q))[i][j],
// K dimension nqp * (features the form's test side uses).  A lives in LDS (written once by the tabulation phases,
// which are those of generic_kernel.hpp: closure, K1..K6, field values); every lane builds its B operand in registers
// with ONE call of the form's own mat() per k-step, which yields all dof*dof blocks at once, and feeds
// v_mfma_f64_16x16x4_f64.  Any geometry (none / polynomial / NURBS), any device form, dof <= 4, nen <= 64.
//
// One workgroup per element of the current colour.  Output tiles (16 test x 16 trial functions):
//   nen <= 64: 4x4 tiles per (i,j) block, 8 wavefronts: wave w owns trial-function tile column w&3 and the test tile
//   rows {2(w>>2), 2(w>>2)+1} (B built once per k-step, used by 2 MFMAs x dof^2 blocks); the two waves of a SIMD
//   cover each other's operand building;  nen <= 32: 2x2 tiles, 4 waves, one tile each;  nen <= 16: one tile, wave 0.
// Accumulators: (row fields per launch) x dof x tiles-per-wave x 4 f64 <= 144 VGPRs.  dof = 4 at nen = 64 takes two
// launches per colour (row fields {0,1} and {2,3}); everything else one.
// The quadrature points are processed in chunks so that Phi fits the LDS next to the other element arrays.
#pragma once
#include <hip/hip_runtime.h>
#include "forms.hpp"
#include "generic_kernel.hpp"

namespace igx {

typedef double fm_d4_t __attribute__((ext_vector_type(4)));

template <class F, class = void> struct shape_order_of { static constexpr int v = F::ORDER; };
template <class F> struct shape_order_of<F, decltype((void)F::SHAPE_ORDER)> { static constexpr int v = F::SHAPE_ORDER; };
// bit f set: mat(e_f, .) is not identically zero (features of the TEST function the matrix form reads)
template <class F, class = void> struct mat_test_mask_of { static constexpr unsigned v = 0xffffffffu; };
template <class F> struct mat_test_mask_of<F, decltype((void)F::MAT_TEST_MASK)> { static constexpr unsigned v = F::MAT_TEST_MASK; };

// bit f set: feature f of Phi is read by the form at all (by mat / vec on either side or through the field values); the others are
// neither tabulated into LDS nor summed into field Hessians.  Cahn-Hilliard reads the Laplacian only: 7 of 13 features, half the
// chunks of points at the same LDS footprint.
template <class F, class = void> struct phi_mask_of { static constexpr unsigned v = 0xffffffffu; };
template <class F> struct phi_mask_of<F, decltype((void)F::PHI_MASK)> { static constexpr unsigned v = F::PHI_MASK; };

// MAT_SYMMETRIC: mat(p, Na, Nb) == mat(p, Nb, Na) for every point (the form's own promise, like the masks); VEC_TEST_MASK: features
// of Na that vec() reads.  A scalar first-order form with MAT_TEST_MASK = gradients, MAT_SYMMETRIC and VEC_TEST_MASK = N runs
// on the pencil walk of gram_mfma.hpp (form_pencil): 480 MFMAs per element and combined band rows instead of the element mode.
template <class F, class = void> struct mat_symmetric_of { static constexpr bool v = false; };
template <class F> struct mat_symmetric_of<F, decltype((void)F::MAT_SYMMETRIC)> { static constexpr bool v = F::MAT_SYMMETRIC; };
template <class F, class = void> struct vec_test_mask_of { static constexpr unsigned v = 0xffffffffu; };
template <class F> struct vec_test_mask_of<F, decltype((void)F::VEC_TEST_MASK)> { static constexpr unsigned v = F::VEC_TEST_MASK; };

// VEC_ZERO: vec() returns zeros (Elasticity3D's F = 0): the vector phase runs only where Dirichlet values are lifted
template <class F, class = void> struct vec_zero_of { static constexpr bool v = false; };
template <class F> struct vec_zero_of<F, decltype((void)F::VEC_ZERO)> { static constexpr bool v = F::VEC_ZERO; };

// Forms whose matrix integrand has point-independent coefficients in the physical-space features (Poisson, mass,
// linear elasticity): K_e[(a,i),(b,j)] = sum_{f,g} C^{ij}_{fg} M_fg[a][b] with the feature Gram matrices
// M_fg = sum_q JW Phi_f(a,q) Phi_g(b,q).  Only the M_fg go through the matrix cores (K dimension nqp instead of
// nqp * features, and one accumulator set per (f,g) pair instead of per (i,j) block); C = mat(e_f, e_g) is applied
// once, at the scatter.  MAT_PAIR_MASK: bit f*8+g set when C_{fg} is not identically zero (f, g < 8).
template <class F, class = void> struct mat_pair_mask_of { static constexpr unsigned long long v = 0ull; };
template <class F> struct mat_pair_mask_of<F, decltype((void)F::MAT_PAIR_MASK)> { static constexpr unsigned long long v = F::MAT_PAIR_MASK; };
constexpr int fm_popcount(unsigned long long m) { int n = 0; while (m) { n += (int)(m & 1ull); m >>= 1; } return n; }
constexpr int fm_phi_slot(unsigned mask, int f) { return fm_popcount((unsigned long long)(mask & ((1u << f) - 1u))); }
constexpr int fm_pair_index(unsigned long long mask, int f, int g) { return fm_popcount(mask & ((1ull << (f * 8 + g)) - 1ull)); }
// Gram matrices are transposes of each other, M_gf[a][b] = M_fg[b][a]: the pairs with f <= g carry everything
constexpr unsigned long long fm_pairs_upper(unsigned long long mask) {
  unsigned long long u = 0ull;
  for (int f = 0; f < 8; ++f) for (int g = 0; g < 8; ++g) if ((mask >> (f * 8 + g)) & 1ull) u |= 1ull << ((f < g ? f : g) * 8 + (f < g ? g : f));
  return u;
}

// optional refinement of MAT_PAIR_MASK: Form::pair_block_mask(f, g) = bit i*DOF+j set when C^{ij}_{fg} is not identically zero
template <class F, class = void> struct has_pair_block_mask { static constexpr bool v = false; };
template <class F> struct has_pair_block_mask<F, decltype((void)F::pair_block_mask(0, 0))> { static constexpr bool v = true; };
template <class F> constexpr unsigned fm_pair_block_mask(int f, int g) { if constexpr (has_pair_block_mask<F>::v) return F::pair_block_mask(f, g); else return 0xffffffffu; }

// optional per-block refinement of MAT_TEST_MASK: Form::block_mask(i,j) = test features block (i,j) of mat() reads
template <class F, class = void> struct has_block_mask { static constexpr bool v = false; };
template <class F> struct has_block_mask<F, decltype((void)F::block_mask(0, 0))> { static constexpr bool v = true; };
template <class F> constexpr unsigned fm_block_mask(int i, int j) { if constexpr (has_block_mask<F>::v) return F::block_mask(i, j); else return 0xffffffffu; }

// point data the MATRIX callback reads (default: everything the form reads anywhere); matrix-only drivers
// (Jacobian / IJacobian / Matrix) skip the rest, e.g. the field Hessians NS-VMS needs for its residual alone
template <class F, class = void> struct mat_need_of { static constexpr unsigned v = F::NEED; };
template <class F> struct mat_need_of<F, decltype((void)F::MAT_NEED)> { static constexpr unsigned v = F::MAT_NEED; };
// band_pt.hpp: the form separates what depends on the point alone (NCOEF, point_coef) from what depends on the basis functions (mat_c)
template <class F, class = void> struct has_point_coef { static constexpr bool v = false; };
template <class F> struct has_point_coef<F, decltype((void)F::NCOEF)> { static constexpr bool v = true; };

struct FCarve {            // offsets in doubles into the dynamic LDS block
  int t1d[3], w1d[3];
  int gX, gW, Ue, Ve, ufix, fixval, fixflag, flux;
  int JW, xq, E1, E2, W0, W1, W2, G;
  int u, ut, gu, hu, hpart, lift, phi;
  int rowbase, rowid, cc, pax, adec, qdec, nrm, boff;
  int sfb;                 // field Hessians by sum factorisation: components per batch (DOF; 1 when the sums of all would not fit)
  int total;
  int QC, nchunk, NEP;     // points per chunk (multiple of 4), chunks, padded nen (16 * tiles)
};

// Phi(a,q): the full chain K2 -> K3 -> K6 for one (point, basis function) pair; o[NF]
template <int DIM, bool SECOND>
__device__ __forceinline__ void shape_features(const double *const t1d[3], const int na[3], const int *qdec, const int *adec, int q, int a,
                                               bool rat, bool geo, const double *gW, const double *W0, const double *W1, const double *W2,
                                               const double *E1, const double *E2, double *o) {
  constexpr int D2 = DIM * DIM;
  const int qp = qdec[q], ap = adec[a];       // packed per-axis indices (no integer division in the hot loops)
  const int qq[3] = {qp & 255, (qp >> 8) & 255, qp >> 16};
  const int aq[3] = {ap & 255, (ap >> 8) & 255, ap >> 16};
  double b0, b1[3], b2[9];
  tensor_basis<DIM, SECOND>(t1d, na, aq, qq, b0, b1, b2);
  if (rat) {   // Rationalize, src/petigarat.f90.in:3-57
    const double w = gW[a], iw0 = 1.0 / W0[q];   // one reciprocal instead of 1 + DIM (+ DIM^2) divisions
    const double r0 = w * b0 * iw0;
    double r1[3];
    for (int i = 0; i < DIM; ++i) r1[i] = (w * b1[i] - r0 * W1[q * DIM + i]) * iw0;
    if (SECOND)
      for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j)
        b2[i * DIM + j] = (w * b2[i * DIM + j] - r0 * W2[q * D2 + i * DIM + j] - r1[i] * W1[q * DIM + j] - r1[j] * W1[q * DIM + i]) * iw0;
    b0 = r0; for (int i = 0; i < DIM; ++i) b1[i] = r1[i];
  }
  o[0] = b0;
  if (!geo) {
    for (int i = 0; i < DIM; ++i) o[1 + i] = b1[i];
    if (SECOND) for (int i = 0; i < D2; ++i) o[1 + DIM + i] = b2[i];
  } else {     // ShapeFunctions, src/petigamapshf.f90.in:30-58
    const double *e1 = E1 + q * D2;
    for (int i = 0; i < DIM; ++i) { double s = 0; for (int al = 0; al < DIM; ++al) s += b1[al] * e1[al * DIM + i]; o[1 + i] = s; }
    if (SECOND) {
      const double *e2 = E2 + (size_t)q * DIM * D2;
      for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) {
        double s = 0;
        for (int al = 0; al < DIM; ++al) {
          for (int be = 0; be < DIM; ++be) s += b2[al * DIM + be] * e1[al * DIM + i] * e1[be * DIM + j];
          s += b1[al] * e2[al * D2 + i * DIM + j];
        }
        o[1 + DIM + i * DIM + j] = s;
      }
    }
  }
}

// First touch (as in the pencil kernel, gram_mfma.hpp): along one axis the entry (row slot a, column slot b) of element e
// is shared by the elements [e + max(a,b) - p, e + min(a,b)]; colours are e mod (p+1), launched in ascending order.
// (two passes over the axis, as in gram_mfma.hpp: this pass covers the elements [rlo, rhi), [blocked, nel) were assembled before)
__device__ __forceinline__ bool fm_first_touch_axis(int e, int a, int b, int nel, int p, int rlo = 0, int rhi = 0x7fffffff, int blocked = 0x7fffffff) {
  const int nb = p + 1;
  int lo = e + (a > b ? a : b) - p, hi = e + (a < b ? a : b);
  if (hi > nel - 1) hi = nel - 1;
  if (hi >= blocked) return false;
  if (lo < rlo) lo = rlo;
  if (hi > rhi - 1) hi = rhi - 1;
  const int c0 = ((lo + nb - 1) / nb) * nb;
  return (c0 <= hi) ? (e % nb == 0) : (e == lo);
}

typedef double fm_d2u_t __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access at 8-byte alignment
template <int N> __device__ __forceinline__ void load_run(const double *p, double *v) {
#pragma unroll
  for (int k = 0; k + 1 < N; k += 2) { const fm_d2u_t t = *reinterpret_cast<const fm_d2u_t *>(p + k); v[k] = t.x; v[k + 1] = t.y; }
  if (N & 1) v[N - 1] = p[N - 1];
}
template <int N> __device__ __forceinline__ void store_run(double *p, const double *v) {
#pragma unroll
  for (int k = 0; k + 1 < N; k += 2) { fm_d2u_t t; t.x = v[k]; t.y = v[k + 1]; *reinterpret_cast<fm_d2u_t *>(p + k) = t; }
  if (N & 1) p[N - 1] = v[N - 1];
}
__device__ __forceinline__ int pow2_floor(int x) { return x < 1 ? 1 : (1 << (31 - __clz(x))); }
// sum over the `np` (power of two) adjacent lanes of a group; every lane of the group ends with the same value
__device__ __forceinline__ double group_sum(double v, int np) {
  for (int m = 1; m < np; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// TA: tiles per side of an (i,j) block (1, 2 or 4).  Row fields [I0, I0+DOFI) are formed by this launch.

// executed v_mfma_f64_16x16x4 instructions per k-step and tile of the matrix phase (roofline accounting of the launchers)
template <class Form> constexpr int fm_mfma_per_kstep(int nfs) {
  const unsigned long long pairs = mat_pair_mask_of<Form>::v;
  if (pairs) return fm_popcount(pairs);
  int n = 0;
  for (int f = 0; f < nfs; ++f) {
    if (!((mat_test_mask_of<Form>::v >> f) & 1u)) continue;
    for (int i = 0; i < Form::DOF; ++i) for (int j = 0; j < Form::DOF; ++j) if ((fm_block_mask<Form>(i, j) >> f) & 1u) n++;
  }
  return n;
}

// Waves per SIMD a 4-wave kernel is compiled for (= workgroups per CU it is sized for): the tabulation phases are latency
// bound, so more resident workgroups win as long as the accumulators leave registers to work with
// (CahnHilliard p=2 tangent: 12.8 / 17.6 / 19.7 / 21.6 M elements/s at 1 / 2 / 3 / 4; NS-VMS p=2 with 16 tiles loses at 4).
template <class Form, int TA, int NW, int DOFI, bool HASM, bool PENCIL = false>
constexpr int fm_min_waves() {
  const unsigned long long pairs = mat_pair_mask_of<Form>::v;
  const int nacc = !HASM ? 0 : (pairs ? fm_popcount(PENCIL ? fm_pairs_upper(pairs) : pairs) : DOFI * Form::DOF);   // accumulator sets per wave
  if (NW != 4) return 1;      // 8 waves per workgroup: 256 VGPRs each (measured: a 128-VGPR cap gains nothing for scalar
                              // forms and costs the NS-VMS residual 20 % in spills)
  const int tiles = nacc * ((TA >= 8) ? TA : ((TA == 4) ? 4 : 1));   // 16x16 accumulator tiles per wave, 8 VGPRs each
  // pencil mode keeps the tiles live through every phase of an element: above 16 tiles (128 registers) a wave gets a SIMD
  // to itself, i.e. the unified 512-entry file with the accumulators in AGPRs
  if (PENCIL) return tiles > 16 ? 1 : (tiles <= 4 ? 3 : 2);   // (a scalar form's 4 tiles stay live next to ~110 registers of tabulation: 3 waves)
  return tiles <= 4 ? 4 : (tiles <= 9 ? 3 : 2);
}

// NW wavefronts per workgroup: 4, or 8 (TA == 4 only: two waves per SIMD share the tile column work).
// HASM = false: vector-only operations (Vector / Function / IFunction) -- same tabulation, no matrix phases.
//
// PENCIL (nen = 4x4x4, matrix-producing drivers, interior pass): combine before write.  The workgroup walks a pencil of
// elements along mesh axis 0 and keeps its accumulator tiles across elements.  Basis functions are numbered a = 16 a0 +
// a1 + 4 a2, so a 16x16 tile pairs two axis-0 node layers; a layer lives in tile slot (layer mod 4) for as long as the walk
// holds it, which makes the operand addresses of an element a rotation of the tile index and never moves an accumulator.
// After an element its first layer is complete for this pencil: the 7 tiles of an (i,j) block that touch that slot are
// fixed up (IGAElementFixSystem on the combined values: the diagonal of a fixed row is the number of walked elements that
// hold the node), written and cleared -- 7/16 of the entries per element reach memory, contributions of the elements of a
// pencil never meet in memory, and only pencils sharing axis-1/2 nodes conflict: 16 colours instead of 64.
// FUSE (element mode, DOFI < DOF): the groups of row fields are formed one after the other inside ONE launch -- the tabulation of
// the element (phases 0-4) is done once instead of once per group (NS-VMS p=3: two groups of two row fields).
template <class Form, int DIM, int TA, int NW, int I0, int DOFI, bool HASM, bool PENCIL = false, bool FUSE = false>
__global__ void __launch_bounds__(64 * NW, (fm_min_waves<Form, TA, NW, DOFI, HASM, PENCIL>()))
feature_assemble(SpaceDev S, ParamsDev prm, OutDev out, ColorRange cr, FCarve cv) {
  static_assert(!PENCIL || (TA == 4 && HASM && DIM == 3), "pencil mode: 4x4 tiles, matrix drivers, dim 3");
  constexpr int DOF = Form::DOF;
  static_assert(!FUSE || (HASM && !PENCIL && I0 == 0 && DOF % DOFI == 0), "fused groups: element mode, matrix drivers, all groups");
  constexpr int NGROUP = FUSE ? DOF / DOFI : 1;
  constexpr int NPASS = NGROUP * ((TA == 16 && HASM) ? 2 : 1);     // (TA = 16: two column panels per group)
  constexpr bool SECOND = Form::ORDER >= 2;                    // tabulation order (fields may need Hessians)
  constexpr bool SECOND_S = shape_order_of<Form>::v >= 2;      // order of the shape-function features mat()/vec() read
  constexpr int D2 = DIM * DIM;
  constexpr int NF = SECOND ? 1 + DIM + D2 : 1 + DIM;
  constexpr int NFS = SECOND_S ? 1 + DIM + D2 : 1 + DIM;       // features of a basis function
  constexpr unsigned PM = phi_mask_of<Form>::v & ((1u << NFS) - 1u);   // those kept in LDS: feature f lives in slot PS(f)
  static_assert(PM & 1u, "the value feature is always kept");
#define PS(f) fm_phi_slot(PM, (f))
#define PHAS(f) (((PM >> (f)) & 1u) != 0)
  constexpr unsigned FMASK = mat_test_mask_of<Form>::v;
  static_assert(NW == 4 || (NW == 8 && (TA == 4 || TA == 8 || TA == 16)), "wave layout");
  static_assert(TA < 8 || (NW == 8 && !PENCIL), "8x8 tiles (nen <= 128: p = 4 in 3-D): wave w owns tile column w and all 8 tile rows");
  // TA = 16 (nen <= 256: p = 5 in 3-D): 16 tile rows, the 16 tile columns in two panels of 8 formed one after the other (the
  // passes of the group loop below); one accumulator set only
  constexpr int NPANEL = (TA == 16 && HASM) ? 2 : 1;
  constexpr int NTA = (TA >= 8) ? TA : ((TA == 4) ? 16 / NW : 1);   // tiles per wave and (i,j) block
  constexpr int NEP = 16 * TA;                                 // padded nen
  constexpr bool HU_FLY = SECOND && !SECOND_S && (Form::NEED & NEED_HU);
  constexpr unsigned long long PAIRS = mat_pair_mask_of<Form>::v;
  constexpr int NS = nscalar_of<Form>::v;                       // > 0: a functional (IGAComputeScalar), no matrix / vector phases
  static_assert(NS == 0 || !HASM, "functionals have no matrix part");
  constexpr bool HASB = has_boundary_of<Form>::v;                // the callback has an `atboundary` branch (bmat / bvec)
  static_assert(!(HASB && PAIRS != 0ull), "forms with a boundary branch do not take the Gram path");
  constexpr bool GRAM = PAIRS != 0ull;                          // constant-coefficient form: accumulate feature Gram matrices
  static_assert(!GRAM || (DOFI == DOF && I0 == 0 && NFS <= 8), "Gram path forms all row fields in one launch");
  // pencil mode accumulates the pairs f <= g only: its write-out reads the mirror tile for the others
  constexpr unsigned long long PACC = PENCIL ? fm_pairs_upper(PAIRS) : PAIRS;
  constexpr int NACC = !HASM ? 1 : (GRAM ? fm_popcount(PACC) : DOFI * DOF);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid_ = threadIdx.x, nthr_ = blockDim.x;
  const int wave_ = __builtin_amdgcn_readfirstlane(tid_ >> 6);

  int el[3], ID[3], off[3], nq_[3], na_[3];
  {
    int b = blockIdx.x;
    int t0 = 0;
    if (!PENCIL) { t0 = b % cr.count[0]; b /= cr.count[0]; }   // PENCIL: the workgroup walks [start0, start0 + count0) itself
    const int t1 = b % cr.count[1]; b /= cr.count[1];
    const int tt[3] = {t0, t1, b};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      el[d] = cr.start[d] + tt[d] * cr.step[d];
      ID[d] = el[d] + S.ax[d].estart;
      off[d] = S.ax[d].off[el[d]];
      nq_[d] = S.ax[d].nqp; na_[d] = S.ax[d].nen;
    }
  }
  // boundary-form pass: one point on the face axis, basis from the end-of-axis table (src/petigaelem.c:796-823)
  const int bid = out.bid; const bool bpass = bid >= 0;
  const int baxis = bpass ? (bid >> 1) : -1, bside = bid & 1;
  if (bpass) nq_[baxis] = 1;
  const int NQ = nq_[0] * nq_[1] * nq_[2], NE = na_[0] * na_[1] * na_[2];
  const int QC = cv.QC, NQP = cv.QC * cv.nchunk;
  const int op = out.op;
  const bool hasV = (NS == 0) && (I0 == 0) && (op == OP_SYSTEM || op == OP_VECTOR || op == OP_FUNCTION || op == OP_IFUNCTION);
  const bool useU = out.U != nullptr, useV = out.V != nullptr;
  const bool geo = S.nsd > 0, rat = S.rational != 0;
  const unsigned need = (hasV || NS > 0) ? Form::NEED : mat_need_of<Form>::v;

  double *t1d[3] = {smem + cv.t1d[0], smem + cv.t1d[1], smem + cv.t1d[2]};
  double *w1d[3] = {smem + cv.w1d[0], smem + cv.w1d[1], smem + cv.w1d[2]};
  double *gX = smem + cv.gX, *gW = smem + cv.gW, *Ue = smem + cv.Ue, *Ve = smem + cv.Ve;
  double *ufix = smem + cv.ufix, *fixval = smem + cv.fixval, *flux = smem + cv.flux;
  int *fixflag = reinterpret_cast<int *>(smem + cv.fixflag);
  double *JW = smem + cv.JW, *xq = smem + cv.xq, *E1 = smem + cv.E1, *E2 = smem + cv.E2;
  double *W0 = smem + cv.W0, *W1 = smem + cv.W1, *W2 = smem + cv.W2, *Gq = smem + cv.G, *nrm = smem + cv.nrm;
  double *fu = smem + cv.u, *fut = smem + cv.ut, *fgu = smem + cv.gu, *fhu = smem + cv.hu, *lift = smem + cv.lift;
  double *phi = smem + cv.phi;                                  // [NFS][QC][NEP]
  long long *rowbase = reinterpret_cast<long long *>(smem + cv.rowbase);   // [NE] browptr of the row of basis function a
  long long *rowid = reinterpret_cast<long long *>(smem + cv.rowid);       // [NE] the row itself
  int *cc = reinterpret_cast<int *>(smem + cv.cc);              // [2][NE] rcnt0, rcnt1 of that row
  int *pax = reinterpret_cast<int *>(smem + cv.pax);            // [3][8][8] column position of b_d in the row of a_d
  int *adec = reinterpret_cast<int *>(smem + cv.adec);          // [NEP] a -> a0 | a1<<8 | a2<<16
  int *qdec = reinterpret_cast<int *>(smem + cv.qdec);          // [NQP] q -> q0 | q1<<8 | q2<<16
  __shared__ int s_anyfix;
  if (tid_ == 0) s_anyfix = 0;
  __syncthreads();
  const bool stamp = kDebug && out.dbg && blockIdx.x == 7 && tid_ == 0;
  int nst = 0;
#define FM_STAMP() do { if (stamp && nst < 31) out.dbg[nst++] = (long long)__builtin_readcyclecounter(); } while (0)
  FM_STAMP();

  // ---- accumulators (PENCIL: they live across the elements of the walk)
  const bool wave_active = (TA >= 2) || (wave_ == 0);
  const int tb_ = (TA >= 8) ? wave_ : ((TA == 4) ? (wave_ & 3) : (TA == 2 ? (wave_ & 1) : 0));     // (TA = 16: within the panel)
  const int ta0 = (TA >= 8) ? 0 : ((TA == 4) ? NTA * (wave_ >> 2) : (TA == 2 ? (wave_ >> 1) : 0));   // first row tile of this wave
  fm_d4_t acc[NACC][NTA];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int t = 0; t < NTA; ++t) acc[k][t] = (fm_d4_t){0, 0, 0, 0};

  const int nwalk = PENCIL ? cr.count[0] : 1;
  for (int ei = 0; ei < nwalk; ++ei) {
  if (PENCIL) {
    el[0] = cr.start[0] + ei; ID[0] = el[0] + S.ax[0].estart; off[0] = S.ax[0].off[el[0]];
    // Everything that depends on the pencil's position on axes 1, 2 only is invariant in this loop; hoisted, it stays live
    // across all phases of every element (measured: hundreds of spilled registers).  Hide the invariance from the optimiser.
    asm volatile("" : "+s"(el[1]), "+s"(el[2]), "+s"(off[1]), "+s"(off[2]), "+s"(ID[1]), "+s"(ID[2]));
  }
  // (likewise for what depends on the thread index and the element shape only: index decodes, division reciprocals, table
  // addresses.  An opaque zero per iteration keeps them from being hoisted: scratch 892 -> 504 bytes per lane.)
  int zv_ = 0, zs_ = 0;
  if (PENCIL) { asm volatile("" : "+v"(zv_)); asm volatile("" : "+s"(zs_)); }
  const int tid = tid_ + zv_, lane = tid & 63, wave = wave_ + zs_, nthr = nthr_ + zs_;
  const int nq[3] = {nq_[0] + zs_, nq_[1] + zs_, nq_[2] + zs_}, na[3] = {na_[0] + zs_, na_[1] + zs_, na_[2] + zs_};
  // tile slot of a local axis-0 index: the layer's position mod 4 (PENCIL); rslot undoes it for the tile index T
  const int rot = PENCIL ? (off[0] & 3) : 0;
  auto rslot = [&](int T) { return PENCIL ? ((T - rot) & 3) : T; };

  // ---- phase 0: 1-D rows, index tables, closure gathers, BC flags, CSR position tables
  for (int a = tid; a < NEP; a += nthr) {
    int aa[3]; slot_decode<PENCIL>(a, na, aa);
    adec[a] = (a < NE) ? (aa[0] | (aa[1] << 8) | (aa[2] << 16)) : 0;
  }
  for (int q = tid; q < NQP; q += nthr) {
    const int q0 = q % nq[0], q1 = (q / nq[0]) % nq[1], q2 = q / (nq[0] * nq[1]);
    qdec[q] = (q < NQ) ? (q0 | (q1 << 8) | (q2 << 16)) : 0;
  }
  // Stage A: every global load that depends on the element position only is issued before anything waits on one
  // (the loads of a workgroup are latency bound: one round trip instead of one per table).  All tables have at most
  // 8 * 8 * NDER = 256 <= blockDim entries, so one entry per thread.
  double tv[3] = {0, 0, 0}, wv[3] = {1, 1, 1}, ptv[3] = {0, 0, 0}, Jax[3]; int pv[3] = {0, 0, 0};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int n = nq[d] * na[d] * NDER;
    const bool face = (d == baxis);
    const double *src = face ? S.ax[d].bnd + (size_t)bside * n : S.ax[d].tab + (size_t)el[d] * n;
    if (tid < n) tv[d] = src[tid];
    if (tid < nq[d]) { if (face) ptv[d] = S.ax[d].bndpt[bside]; else { wv[d] = S.ax[d].w[el[d] * nq[d] + tid]; ptv[d] = S.ax[d].pt[el[d] * nq[d] + tid]; } }
    Jax[d] = S.ax[d].J[el[d]];
    const int Wd = 2 * S.ax[d].p + 1;
    if (tid < na[d] * na[d]) {   // column position, with the first-touch flag of the slot pair in bit 30
      const int ad = tid / na[d], bd = tid - ad * na[d];
      // (PENCIL: along the walk everything is combined on chip, so every entry is written once per pencil)
      const bool two = (d == 2) && out.ft2_hi > 0;
      pv[d] = S.ax[d].P[(off[d] + ad) * Wd + (bd - ad + S.ax[d].p)] |
              ((out.first_touch && ((PENCIL && d == 0) || fm_first_touch_axis(el[d], ad, bd, S.ax[d].nel, S.ax[d].p, two ? out.ft2_lo : 0, two ? out.ft2_hi : 0x7fffffff, two ? out.ft2_blocked : 0x7fffffff))) ? (1 << 30) : 0);
    }
  }
  const int gw0 = S.ax[0].gwidth, gw1 = S.ax[1].gwidth;
  const int nr0 = S.ax[0].nrow, nr1 = S.ax[1].nrow;
  const bool isa = tid < NE;                     // nen <= 64 <= blockDim: one basis function per thread
  const int a = isa ? tid : 0;
  int a012[3]; slot_decode<PENCIL>(a, na, a012);
  const int a0 = a012[0], a1 = a012[1], a2 = a012[2];
  const int i0 = off[0] + a0, i1 = off[1] + a1, i2 = off[2] + a2;
  const size_t g = (size_t)i0 + (size_t)gw0 * ((size_t)i1 + (size_t)gw1 * (size_t)i2);
  // rows in closed form (no dependent load): what hangs on the row index joins the same round trip
  const int r0 = i0 < S.ax[0].rwrap ? i0 : i0 - S.ax[0].rwrap, r1 = i1 < S.ax[1].rwrap ? i1 : i1 - S.ax[1].rwrap, r2 = i2 < S.ax[2].rwrap ? i2 : i2 - S.ax[2].rwrap;
  const size_t row = (size_t)r0 + (size_t)nr0 * ((size_t)r1 + (size_t)nr1 * (size_t)r2);
  double xg[DIM], wg = 1, Uv[DOF], Vv[DOF];
  int c0r = 0, c1r = 0, c2r = 0; long long pre0 = 0, pre1 = 0, pre2 = 0;
#pragma unroll
  for (int c = 0; c < DIM; ++c) xg[c] = 0;
#pragma unroll
  for (int c = 0; c < DOF; ++c) { Uv[c] = 0; Vv[c] = 0; }
  if (isa) {
    if (geo) for (int c = 0; c < DIM; ++c) xg[c] = S.X[g * DIM + c];
    if (rat) wg = S.W[g];
    c0r = S.ax[0].rcnt[r0]; c1r = S.ax[1].rcnt[r1];
    if (HASM) { c2r = S.ax[2].rcnt[r2]; pre0 = S.ax[0].prefix[r0]; pre1 = S.ax[1].prefix[r1]; pre2 = S.ax[2].prefix[r2]; }
    if (useU) for (int c = 0; c < DOF; ++c) Uv[c] = out.U[row * DOF + c];
    if (useV) for (int c = 0; c < DOF; ++c) Vv[c] = out.V[row * DOF + c];
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (tid < nq[d] * na[d] * NDER) t1d[d][tid] = tv[d];
    if (tid < nq[d]) { w1d[d][tid] = wv[d]; w1d[d][nq[d] + tid] = ptv[d]; }   // weights, then point coordinates
    if (tid < na[d] * na[d]) { const int ad = tid / na[d], bd = tid - ad * na[d]; pax[d * 64 + ad * 8 + bd] = pv[d]; }
  }
  FM_STAMP();
  // Stage B: the row tables and the closure values into LDS, BC flags
  if (isa) {
    rowid[a] = (long long)row;
    // browptr[row] in closed form from the per-axis prefix tables (k_browptr): small, cache-resident tables instead of a
    // dependent load from the 8-byte-per-row array in HBM
    if (HASM) rowbase[a] = pre2 * S.ax[1].tot * S.ax[0].tot + (long long)c2r * (pre1 * S.ax[0].tot + (long long)c1r * pre0);
    cc[a] = c0r; cc[NE + a] = c1r;
    if (geo) for (int c = 0; c < DIM; ++c) gX[a * DIM + c] = xg[c];
    if (rat) gW[a] = wg;
    if (useU) for (int c = 0; c < DOF; ++c) Ue[a * DOF + c] = Uv[c];
    if (useV) for (int c = 0; c < DOF; ++c) Ve[a * DOF + c] = Vv[c];
    const int aa[3] = {a0, a1, a2};
    for (int c = 0; c < DOF; ++c) { fixflag[a * DOF + c] = 0; fixval[a * DOF + c] = 0; flux[a * DOF + c] = 0; }
    if (op != OP_MATRIX && op != OP_VECTOR && op != OP_SCALAR) {   // IGAElementBuildFix (src/petigaelem.c:1214-1283); IGAComputeScalar reads U as it is
      for (int d = 0; d < DIM; ++d) {
        if (S.ax[d].periodic) continue;
        for (int side = 0; side < 2; ++side) {
          if (ID[d] != (side ? S.ax[d].esizes - 1 : 0)) continue;
          if (aa[d] != (side ? na[d] - 1 : 0)) continue;
          const BCDev &bv = S.bcv[d][side];
          for (int k = 0; k < bv.count; ++k) {
            const int c = bv.field[k];
            if (c >= DOF) continue;
            fixflag[a * DOF + c] = 1;
            fixval[a * DOF + c] = S.fixtable ? S.fixtable[row * DOF + c] : bv.value[k];
            s_anyfix = 1;
          }
          const BCDev &bl = S.bcl[d][side];
          if (bl.count && !geo) {   // BoundaryArea, no-geometry branch (src/petigaelem.c:1118-1132); mapped: add_mapped_flux below
            double A = 1;
            if (DIM > 1) {
              for (int i = 0; i < DIM; ++i) if (i != d) A *= Jax[i] / (double)na[i];
              A *= (DIM == 2) ? 2 : 4;
            }
            for (int k = 0; k < bl.count; ++k) { const int c = bl.field[k]; if (c < DOF) flux[a * DOF + c] += bl.value[k] * A; }
          }
        }
      }
    }
  }
  FM_STAMP();
  __syncthreads();
  const bool anyfix = s_anyfix != 0;
  if (geo && !bpass && op != OP_MATRIX && op != OP_VECTOR && op != OP_SCALAR) add_mapped_flux<DIM, DOF, PENCIL>(S, ID, el, t1d, w1d, nq, na, gX, gW, rat, flux, tid, nthr);
  if (anyfix && (useU || useV)) {   // IGAElementFixValues / DelValues (src/petigaelem.c:1327-1358)
    for (int k = tid; k < NE * DOF; k += nthr)
      if (fixflag[k]) { if (useU) { ufix[k] = Ue[k]; Ue[k] = fixval[k]; } if (useV) Ve[k] = 0.0; }
    __syncthreads();
  }

  FM_STAMP();
  // ---- phase 1: per-point geometry for all points (K1, K3 sums, K4, K5).  np1 adjacent lanes share one point and
  // split the sum over basis functions; partial sums meet in a fixed butterfly order (repeatable).
  double Jel = 1;
#pragma unroll
  for (int d = 0; d < 3; ++d) if (d != baxis) Jel *= Jax[d];   // bnd_detJac = 1
  // First-order tabulation on a mapped geometry: the sums over the nen control points factorise over the axes
  // (sum factorisation, three short contractions through LDS instead of nqp * nen * 16 products per element).  They
  // run on homogeneous coordinates: A_c = sum_a (w_a X_a[c]) N_a and W = sum_a w_a N_a with their first derivatives;
  // x = A / W and dx/du = (dA - x dW) / W reproduce Rationalize + GeometryMap (src/petigarat.f90.in, petigamapgeo.f90.in).
  // Second-order tabulations (Cahn-Hilliard, NS-VMS) take the same route with the second derivatives: 3 derivative orders per
  // axis, 6 pairs after two axes, 10 sums per component and point; x, dx/du, d2x/du2 follow by the quotient rule.  The
  // per-point loop over the 64 basis functions with its 39 accumulators cost 540k cycles per element of NS-VMS on a NURBS.
  const bool sumfact = geo || rat;
  // second-order geometry only when something reads it: second derivatives of N (SECOND_S) or the Hessians of the fields this
  // driver's callback needs (NS-VMS: the Residual does, the Tangent does not -- its tabulation stays first order)
  const bool need2 = SECOND && (SECOND_S || (need & NEED_HU) != 0);
  constexpr int SF_NV = SECOND ? 3 : 2, SF_NM = SECOND ? 6 : 3, SF_NK = SECOND ? 10 : 4;
  double *SF = phi;                                   // [NC][SF_NK][NQ], in the (not yet used) Phi region
  // sum_a COEF(a, c) * (tensor-product basis function a and its parametric derivatives) at every point, NC components:
  //   T1[c][v][q0][a1][a2] = sum_a0 C[a][c] n0[q0][a0][v], v = derivative order on axis 0
  //   T2[c][m][q0][q1][a2], m -> orders (v0, v1): (0,0) (1,0) (0,1) | (2,0) (1,1) (0,2)
  //   SF[c][k][q], k: 0 value, 1..3 d/du_i, 4..9 d2/du_i du_j for (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
  // (a macro, not a lambda: the matrix kernels hold one copy of these loops, laid out as before the field sums existed)
#define FM_SUM_FACTORISE(NCEXPR, COEF) do { const int NC = (NCEXPR); \
    const int n1 = NC * SF_NV * nq[0] * na[1] * na[2], n2 = NC * SF_NM * nq[0] * nq[1] * na[2], n3 = NC * SF_NK * NQ; \
    double *T1 = phi + n3, *T2 = T1 + n1; \
    for (int i = tid; i < n1; i += nthr) { \
      int r = i; const int a2 = r % na[2]; r /= na[2]; const int a1 = r % na[1]; r /= na[1]; const int q0 = r % nq[0]; r /= nq[0]; const int v = r % SF_NV, c = r / SF_NV; \
      if (v == 2 && !need2) continue; \
      double sm = 0; \
      for (int a0 = 0; a0 < na[0]; ++a0) { \
        const int a = slot_of<PENCIL>(a0, a1, a2, na); \
        sm += (COEF) * t1d[0][(q0 * na[0] + a0) * NDER + v]; \
      } \
      T1[i] = sm; \
    } \
    __syncthreads(); \
    for (int i = tid; i < n2; i += nthr) { \
      int r = i; const int a2 = r % na[2]; r /= na[2]; const int q1 = r % nq[1]; r /= nq[1]; const int q0 = r % nq[0]; r /= nq[0]; const int m = r % SF_NM, c = r / SF_NM; \
      const int v0 = (m == 1 || m == 4) ? 1 : (m == 3 ? 2 : 0), v1 = (m == 2 || m == 4) ? 1 : (m == 5 ? 2 : 0); \
      if (m >= 3 && !need2) continue; \
      double sm = 0; \
      for (int a1 = 0; a1 < na[1]; ++a1) sm += T1[(((c * SF_NV + v0) * nq[0] + q0) * na[1] + a1) * na[2] + a2] * t1d[1][(q1 * na[1] + a1) * NDER + v1]; \
      T2[i] = sm; \
    } \
    __syncthreads(); \
    for (int i = tid; i < n3; i += nthr) { \
      const int q = i % NQ, k = (i / NQ) % SF_NK, c = i / (SF_NK * NQ); \
      const int qp = qdec[q]; const int q0 = qp & 255, q1 = (qp >> 8) & 255, q2 = qp >> 16; \
 \
      const int m = (k == 1 || k == 6) ? 1 : ((k == 2 || k == 8) ? 2 : (k == 4 ? 3 : (k == 5 ? 4 : (k == 7 ? 5 : 0)))); \
      const int v2 = (k == 3 || k == 6 || k == 8) ? 1 : (k == 9 ? 2 : 0); \
      if (k >= 4 && !need2) continue; \
      double sm = 0; \
      for (int a2 = 0; a2 < na[2]; ++a2) sm += T2[(((c * SF_NM + m) * nq[0] + q0) * nq[1] + q1) * na[2] + a2] * t1d[2][(q2 * na[2] + a2) * NDER + v2]; \
      SF[i] = sm; \
    } \
    __syncthreads(); \
  } while (0)
  // components of the geometry sums: X (times w) and w
  if (sumfact) FM_SUM_FACTORISE(DIM + 1, ((c < DIM) ? (geo ? gX[a * DIM + c] * (rat ? gW[a] : 1.0) : 0.0) : (rat ? gW[a] : 1.0)));
  {
    const int np1 = 1;                 // one lane per point: the sums over the basis functions are done (sum factorisation above)
    const int qstep = nthr / np1;
    for (int qb = 0; qb < NQP; qb += qstep) {
      const int q = qb + tid / np1, part = tid & (np1 - 1);
      const bool valid = q < NQ;
      const int qs = valid ? q : 0;
      const int qp = qdec[qs];
      const int qq[3] = {qp & 255, (qp >> 8) & 255, qp >> 16};
      double detX = 1.0;
      double w0 = 1, w1[3] = {0, 0, 0}, w2[9] = {0};
      double x0[3], X1[9], X2[27];
#pragma unroll
      for (int d = 0; d < DIM; ++d) x0[d] = w1d[d][nq[d] + qq[d]];
      if (sumfact) {
        if (valid) {
          // index of the second derivative (i, j) among the 10 sums
          auto k2 = [](int i, int j) { const int lo = i < j ? i : j, hi = i < j ? j : i; return 4 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)); };
          if (rat) {
            w0 = SF[(DIM * SF_NK + 0) * NQ + q];
            for (int i = 0; i < DIM; ++i) w1[i] = SF[(DIM * SF_NK + 1 + i) * NQ + q];
            if (SECOND && need2) for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) w2[i * DIM + j] = SF[(DIM * SF_NK + k2(i, j)) * NQ + q];
          }
          if (geo) {   // quotient rule on the homogeneous sums A = sum w X N, W = sum w N
            const double iw = 1.0 / w0;
            for (int c = 0; c < DIM; ++c) {
              x0[c] = SF[(c * SF_NK + 0) * NQ + q] * iw;
              for (int al = 0; al < DIM; ++al) X1[c * DIM + al] = (SF[(c * SF_NK + 1 + al) * NQ + q] - x0[c] * w1[al]) * iw;
              if (SECOND && need2) for (int al = 0; al < DIM; ++al) for (int be = 0; be < DIM; ++be)
                X2[c * D2 + al * DIM + be] = (SF[(c * SF_NK + k2(al, be)) * NQ + q] - x0[c] * w2[al * DIM + be] - X1[c * DIM + al] * w1[be] - X1[c * DIM + be] * w1[al]) * iw;
            }
          }
        }
      }
      if (part != 0 || q >= NQP) continue;
      if (!valid) { JW[q] = 0; for (int i = 0; i < DIM; ++i) xq[q * DIM + i] = 0; continue; }   // padded point
      if (rat) {
        W0[q] = w0;
        for (int i = 0; i < DIM; ++i) W1[q * DIM + i] = w1[i];
        if (SECOND && need2) for (int i = 0; i < D2; ++i) W2[q * D2 + i] = w2[i];
      }
      if (geo) {
        detX = det3(X1, DIM);
        double e1[9];
        inv3(X1, DIM, detX, e1);
        for (int i = 0; i < D2; ++i) E1[q * D2 + i] = e1[i];
        if (SECOND && need2) {   // InverseMap order 2, src/petigamapinv.f90.in:32-45: E2[c][i][j] = -X2[k][a][b] e1[a][i] e1[b][j] e1[c][k],
          double e2[27];  // contracted one index at a time (3 x 81 products instead of 729)
          for (int i = 0; i < D2 * DIM; ++i) e2[i] = 0;
          for (int k = 0; k < DIM; ++k) {
            double U[9], T[9];
            for (int a = 0; a < DIM; ++a) for (int j = 0; j < DIM; ++j) { double sm = 0; for (int b = 0; b < DIM; ++b) sm += X2[k * D2 + a * DIM + b] * e1[b * DIM + j]; U[a * DIM + j] = sm; }
            for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) { double sm = 0; for (int a = 0; a < DIM; ++a) sm += U[a * DIM + j] * e1[a * DIM + i]; T[i * DIM + j] = sm; }
            for (int c = 0; c < DIM; ++c) for (int i = 0; i < D2; ++i) e2[c * D2 + i] -= T[i] * e1[c * DIM + k];
          }
          for (int i = 0; i < D2 * DIM; ++i) E2[(size_t)q * DIM * D2 + i] = e2[i];
        }
        if (!(detX > 0.0)) atomicExch(out.errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
      }
      if (bpass) {   // K7: IGA_GetNormal, src/petigaval.F90:45-99; detJac *= detS instead of detX (src/petigaelem.c:1012-1029)
        double n[3] = {0, 0, 0}, dS = 1;
        if (!geo) n[baxis] = 1.0;
        else if (DIM == 3) {
          const int r1 = (baxis + 1) % 3, r2 = (baxis + 2) % 3;
          const double s0 = X1[0 * DIM + r1], s1 = X1[1 * DIM + r1], s2 = X1[2 * DIM + r1];
          const double t0 = X1[0 * DIM + r2], t1 = X1[1 * DIM + r2], t2 = X1[2 * DIM + r2];
          n[0] = s1 * t2 - s2 * t1; n[1] = s2 * t0 - s0 * t2; n[2] = s0 * t1 - s1 * t0;
          dS = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
          n[0] /= dS; n[1] /= dS; n[2] /= dS;
        } else if (DIM == 2) {
          double t0, t1;
          if (baxis == 0) { t0 = +X1[0 * DIM + 1]; t1 = +X1[1 * DIM + 1]; } else { t0 = -X1[0 * DIM + 0]; t1 = -X1[1 * DIM + 0]; }
          n[0] = +t1; n[1] = -t0;
          dS = sqrt(n[0] * n[0] + n[1] * n[1]);
          n[0] /= dS; n[1] /= dS;
        }
        for (int i = 0; i < DIM; ++i) nrm[q * DIM + i] = bside ? n[i] : -n[i];
        detX = dS;
      }
      double w = 1;
#pragma unroll
      for (int d = 0; d < 3; ++d) w *= w1d[d][qq[d]];
      JW[q] = (Jel * detX) * w;
      for (int i = 0; i < DIM; ++i) xq[q * DIM + i] = x0[i];
      if (Form::NEED & NEED_G) {           // IGAPointFormInvGradGeomMap, src/petigapoint.c:269-294
        for (int a = 0; a < DIM; ++a) for (int i = 0; i < DIM; ++i) {
          const double L = S.ax[a].J[el[a]];
          Gq[q * D2 + a * DIM + i] = geo ? E1[q * D2 + a * DIM + i] / L : ((a == i) ? 1 / L : 0.0);
        }
      }
    }
  }
  __syncthreads();

  // (a matrix-producing kernel of a form whose matrix callback never reads the field Hessians carries no code for them: the
  // launcher sends the one driver that would need both, IGAComputeSystem, to the point-form kernel for such a form)
  constexpr bool HU_HERE = HU_FLY && (!HASM || (mat_need_of<Form>::v & NEED_HU) != 0);
  if constexpr (HU_HERE) if (need & NEED_HU) {
    // Hessians of the fields when the form itself never reads second derivatives of N (NS-VMS): no second-derivative feature
    // is ever formed.  The homogeneous sums A_c = sum_a (w_a U_a,c) N_a with their first and second PARAMETRIC derivatives come
    // from the same sum factorisation as the geometry (three short contractions through LDS); u, du, d2u follow by the
    // quotient rule with W, dW, d2W of the point (Rationalize summed over a, src/petigarat.f90.in), and the physical Hessian is
    // H_ij = d2u_ab E1_ai E1_bj + du_a E2_aij (ShapeFunctions summed over a, src/petigamapshf.f90.in:30-58): one lane per
    // (point, field) instead of nen second-order shape functions per point -- 50k of the 145k cycles of an NS-VMS residual
    // element on a NURBS.
    auto k2 = [](int i, int j) { const int lo = i < j ? i : j, hi = i < j ? j : i; return 4 + (lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)); };
    const int sfb = (cv.sfb > 0 && cv.sfb < DOF) ? cv.sfb : DOF;      // components per batch
    for (int c0 = 0; c0 < DOF; c0 += sfb) {
    FM_SUM_FACTORISE(sfb, (useU ? (rat ? gW[a] : 1.0) * Ue[a * DOF + c0 + c] : 0.0));
    for (int idx = tid; idx < NQP * sfb; idx += nthr) {
      const int q = idx / sfb, cl = idx - q * sfb, c = c0 + cl;
      double H[D2];
#pragma unroll
      for (int i = 0; i < D2; ++i) H[i] = 0;
      if (q < NQ) {
        const double w0 = rat ? W0[q] : 1.0, iw = 1.0 / w0;
        const double u = SF[(cl * SF_NK + 0) * NQ + q] * iw;
        double u1[DIM], u2[D2];
#pragma unroll
        for (int al = 0; al < DIM; ++al) u1[al] = (SF[(cl * SF_NK + 1 + al) * NQ + q] - (rat ? u * W1[q * DIM + al] : 0.0)) * iw;
#pragma unroll
        for (int al = 0; al < DIM; ++al)
#pragma unroll
          for (int be = 0; be < DIM; ++be) {
            double t = SF[(cl * SF_NK + k2(al, be)) * NQ + q];
            if (rat) t -= u * W2[q * D2 + al * DIM + be] + u1[al] * W1[q * DIM + be] + u1[be] * W1[q * DIM + al];
            u2[al * DIM + be] = t * iw;
          }
        if (geo) {
          const double *e1 = E1 + q * D2, *e2 = E2 + (size_t)q * DIM * D2;
#pragma unroll
          for (int i = 0; i < DIM; ++i)
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
              double sm = 0;
#pragma unroll
              for (int al = 0; al < DIM; ++al) {
#pragma unroll
                for (int be = 0; be < DIM; ++be) sm += u2[al * DIM + be] * e1[al * DIM + i] * e1[be * DIM + j];
                sm += u1[al] * e2[al * D2 + i * DIM + j];
              }
              H[i * DIM + j] = sm;
            }
        } else {
#pragma unroll
          for (int i = 0; i < D2; ++i) H[i] = u2[i];
        }
      }
#pragma unroll
      for (int i = 0; i < D2; ++i) fhu[(q * DOF + c) * D2 + i] = H[i];
    }
    __syncthreads();
    }   // batches of components
  }

  FM_STAMP();
  double Facc[DOF];
#pragma unroll
  for (int i = 0; i < DOF; ++i) Facc[i] = 0;
  const bool dolift = anyfix && op == OP_SYSTEM;
  int npv = pow2_floor(nthr / NE); if (npv > 16) npv = 16;

#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {     // (fully unrolled: I0p is a constant in every copy)
  const int I0p = I0 + (pass / NPANEL) * DOFI;   // first row field of this group
  const int tb = tb_ + (pass % NPANEL) * 8;      // tile column of this wave (TA = 16: in this pass's panel)
  const bool retab = (pass == 0) || (cv.nchunk > 1);   // a later pass finds Phi and the field values of a single chunk still in LDS
  if (pass > 0) {
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
      for (int t = 0; t < NTA; ++t) acc[k][t] = (fm_d4_t){0, 0, 0, 0};
    if (retab) __syncthreads();          // the last chunk's readers of the group before are done with phi
  }
  for (int ch = 0; ch < cv.nchunk; ++ch) {
    const int qc0 = ch * QC;
    if (ch) __syncthreads();             // the previous chunk's readers are done with phi / the field arrays

    // ---- phase 2: Phi of this chunk, [f][ql][a] with zero padding (a >= nen, q >= nqp)
    if (retab)
    for (int idx = tid; idx < QC * NEP; idx += nthr) {
      const int ql = idx / NEP, a = idx - ql * NEP, q = qc0 + ql;
      double o[NFS];
      if (a < NE && q < NQ) shape_features<DIM, SECOND_S>(t1d, na, qdec, adec, q, a, rat, geo, gW, W0, W1, W2, E1, E2, o);
      else {
#pragma unroll
        for (int f = 0; f < NFS; ++f) o[f] = 0;
      }
#pragma unroll
      for (int f = 0; f < NFS; ++f) if (PHAS(f)) phi[(PS(f) * QC + ql) * NEP + a] = o[f];
    }
    if (retab) __syncthreads();

    FM_STAMP();
    // ---- phase 3: field values at the chunk's points (src/petigaval.F90:182-232); np3 lanes per (point, field)
    if (retab && (need & (NEED_U | NEED_UT | NEED_GU | NEED_HU))) {
      {
        int np3 = pow2_floor(nthr / (QC * DOF)); if (np3 > 16) np3 = 16;
        const int istep = nthr / np3;
        for (int ib = 0; ib < QC * DOF; ib += istep) {
          const int idx = ib + tid / np3, part = tid & (np3 - 1);
          const bool valid = idx < QC * DOF;
          const int ids = valid ? idx : 0;
          const int ql = ids / DOF, c = ids - ql * DOF;
          double u = 0, ut = 0, g[3] = {0, 0, 0}, h[9] = {0};
          if (valid) for (int a = part; a < NE; a += np3) {
            const double Ua = useU ? Ue[a * DOF + c] : 0.0;
            const double f0 = phi[(0 * QC + ql) * NEP + a];
            u += f0 * Ua;
            if (useV) ut += f0 * Ve[a * DOF + c];
            if ((Form::NEED & NEED_GU) && (need & NEED_GU)) for (int i = 0; i < DIM; ++i) if (PHAS(1 + i)) g[i] += phi[(PS(1 + i) * QC + ql) * NEP + a] * Ua;
            if (SECOND_S && (Form::NEED & NEED_HU) && (need & NEED_HU)) for (int i = 0; i < D2; ++i) if (PHAS(1 + DIM + i)) h[i] += phi[(PS(1 + DIM + i) * QC + ql) * NEP + a] * Ua;
          }
          u = group_sum(u, np3); ut = group_sum(ut, np3);
          if ((Form::NEED & NEED_GU) && (need & NEED_GU)) for (int i = 0; i < DIM; ++i) if (PHAS(1 + i)) g[i] = group_sum(g[i], np3);
          if (SECOND_S && (Form::NEED & NEED_HU) && (need & NEED_HU)) for (int i = 0; i < D2; ++i) if (PHAS(1 + DIM + i)) h[i] = group_sum(h[i], np3);
          if (valid && part == 0) {
            fu[idx] = u; fut[idx] = ut;
            if ((Form::NEED & NEED_GU) && (need & NEED_GU)) for (int i = 0; i < DIM; ++i) fgu[idx * DIM + i] = g[i];   // regions exist only when needed
            if (SECOND_S && (Form::NEED & NEED_HU) && (need & NEED_HU)) for (int i = 0; i < D2; ++i) fhu[idx * D2 + i] = h[i];
          }
        }
      }
    }
    // ---- phase 4: Dirichlet lifting features of the chunk: lift[ql][j][:] = sum_b fixed(b,j) v_bj Phi[q][b][:]
    if (retab && dolift && hasV) {
      for (int idx = tid; idx < QC * DOF * NFS; idx += nthr) {
        const int f = idx % NFS, j = (idx / NFS) % DOF, ql = idx / (NFS * DOF);
        double sm = 0;
        const int fs = __popc(PM & ((1u << f) - 1u));
        if ((PM >> f) & 1u) for (int b = 0; b < NE; ++b) if (fixflag[b * DOF + j]) sm += fixval[b * DOF + j] * phi[(fs * QC + ql) * NEP + b];
        lift[idx] = sm;
      }
    }
    __syncthreads();

    auto point = [&](int q, int ql) {
      PtView p;
      p.x = xq + q * DIM; p.u = fu + ql * DOF; p.ut = fut + ql * DOF; p.gu = fgu + ql * DOF * DIM; p.hu = fhu + (HU_FLY ? q : ql) * DOF * D2;
      p.G = Gq + q * D2; p.prm = prm.v; p.shift = out.shift; p.t = out.t;
      p.normal = bpass ? nrm + q * DIM : nullptr; p.atboundary = bpass ? 1 : 0; p.boundary_id = bid;
      return p;
    };

    FM_STAMP();
    // ---- phase 5: K_e += A^T B on the matrix cores
    if constexpr (HASM) if (wave_active && !(kDebug && (out.debug & 4))) {
      const int kq = lane >> 4, col = rslot(tb) * 16 + (lane & 15);
      for (int s = 0; s < QC / 4; ++s) {
        const int ql = 4 * s + kq, q = qc0 + ql;
        const bool live = q < NQ;
        const PtView p = point(live ? q : 0, live ? ql : 0);
        const double jw = JW[q];
        double nb[NFS];
#pragma unroll
        for (int g = 0; g < NFS; ++g) nb[g] = PHAS(g) ? phi[(PS(g) * QC + ql) * NEP + col] : 0.0;
        if constexpr (GRAM) {
#pragma unroll
          for (int f = 0; f < NFS; ++f) {
            if (!((PACC >> (f * 8)) & 0xffull)) continue;
            double A[NTA];
#pragma unroll
            for (int t = 0; t < NTA; ++t) A[t] = PHAS(f) ? phi[(PS(f) * QC + ql) * NEP + rslot(ta0 + t) * 16 + (lane & 15)] : 0.0;
#pragma unroll
            for (int g = 0; g < NFS; ++g) {
              if (!((PACC >> (f * 8 + g)) & 1ull)) continue;
              const double B = live ? nb[g] * jw : 0.0;
#pragma unroll
              for (int t = 0; t < NTA; ++t)
                acc[fm_pair_index(PACC, f, g)][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t], B, acc[fm_pair_index(PACC, f, g)][t], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
        for (int f = 0; f < NFS; ++f) {
          if (!((FMASK >> f) & 1u) && !(HASB && bpass)) continue;
          double ef[NFS];
#pragma unroll
          for (int g = 0; g < NFS; ++g) ef[g] = (g == f) ? 1.0 : 0.0;
          double T[DOF * DOF];
          if constexpr (HASB) { if (bpass) Form::bmat(p, ef, nb, T); else Form::mat(p, ef, nb, T); }
          else Form::mat(p, ef, nb, T);
          double A[NTA];
#pragma unroll
          for (int t = 0; t < NTA; ++t) A[t] = PHAS(f) ? phi[(PS(f) * QC + ql) * NEP + rslot(ta0 + t) * 16 + (lane & 15)] : 0.0;
#pragma unroll
          for (int i = 0; i < DOFI; ++i)
#pragma unroll
            for (int j = 0; j < DOF; ++j) {
              if (!((fm_block_mask<Form>(I0p + i, j) >> f) & 1u) && !(HASB && bpass)) continue;   // structurally zero for this block
              const double B = live ? T[(I0p + i) * DOF + j] * jw : 0.0;
#pragma unroll
              for (int t = 0; t < NTA; ++t)
                acc[i * DOF + j][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t], B, acc[i * DOF + j][t], 0, 0, 0);
            }
        }
        }
      }
    }

    FM_STAMP();
    // ---- phase 6: F_e (vector part of IGAFormSystem), Dirichlet lifting through linearity of mat in Nb;
    // npv adjacent lanes share one basis function and split the points
    if constexpr (NS > 0) {
      // ---- IGAComputeScalar (src/petigacomp.c:35-98): per point JW * scalar(p), summed per element in a fixed order
      for (int ql = tid; ql < QC; ql += nthr) {
        const int q = qc0 + ql;
        double Sq[NS];
        if (q < NQ) { const PtView p = point(q, ql); Form::scalar(p, Sq); }
#pragma unroll
        for (int i = 0; i < NS; ++i) lift[ql * NS + i] = (q < NQ) ? Sq[i] * JW[q] : 0.0;
      }
      __syncthreads();
      if (tid < NS) for (int ql = 0; ql < QC; ++ql) Facc[0] += lift[ql * NS + tid];
    }
    if constexpr (NS == 0) if (hasV && pass == 0 && (!vec_zero_of<Form>::v || dolift) && tid < NE * npv) {
      const int a = tid / npv, part = tid & (npv - 1);
      const int qn = (qc0 + QC <= NQ) ? QC : (NQ - qc0 > 0 ? NQ - qc0 : 0);
      for (int ql = part; ql < qn; ql += npv) {
        const PtView p = point(qc0 + ql, ql);
        double Na[NFS];
#pragma unroll
        for (int f = 0; f < NFS; ++f) Na[f] = PHAS(f) ? phi[(PS(f) * QC + ql) * NEP + a] : 0.0;
        double R[DOF];
        if constexpr (HASB) { if (bpass) Form::bvec(p, Na, R); else Form::vec(p, Na, R); }
        else Form::vec(p, Na, R);
        if (dolift) {
#pragma unroll
          for (int j = 0; j < DOF; ++j) {
            double T[DOF * DOF];
            if constexpr (HASB) { if (bpass) Form::bmat(p, Na, lift + ((size_t)ql * DOF + j) * NFS, T); else Form::mat(p, Na, lift + ((size_t)ql * DOF + j) * NFS, T); }
            else Form::mat(p, Na, lift + ((size_t)ql * DOF + j) * NFS, T);
#pragma unroll
            for (int i = 0; i < DOF; ++i) R[i] -= T[i * DOF + j];
          }
        }
        const double jw = JW[qc0 + ql];
#pragma unroll
        for (int i = 0; i < DOF; ++i) Facc[i] += R[i] * jw;
      }
    }
  }

  FM_STAMP();
  // ---- IGAElementFixSystem / FixJacobian on the tiles, then IGAElementAssembleMat (coloured, conflict-free).
  // Per tile the four row groups are read together, then written: 4 x DOFI*DOF loads in flight per lane.
  if constexpr (HASM && !PENCIL) if (wave_active && !(kDebug && (out.debug & 1))) {
    const int b = tb * 16 + (lane & 15);
    const int bp = adec[b];
    const int b0 = bp & 255, b1 = (bp >> 8) & 255, b2 = bp >> 16;
#pragma unroll
    for (int t = 0; t < NTA; ++t) {
      const int arow0 = (ta0 + t) * 16;
      constexpr int RB = 4;   // row groups per batch: their loads are in flight together
#pragma unroll
      for (int rb = 0; rb < 4; rb += RB) {
      double v[RB][DOFI * DOF]; double *dst[RB]; bool ok[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int a = arow0 + (lane >> 4) + 4 * (rb + r);
        ok[r] = (a < NE) && (b < NE);
        const int as = ok[r] ? a : 0;
        const int ap = adec[as];
        const int a0 = ap & 255, a1 = (ap >> 8) & 255, a2 = ap >> 16;
        const int v0 = pax[0 * 64 + a0 * 8 + b0], v1 = pax[1 * 64 + a1 * 8 + b1], v2 = pax[2 * 64 + a2 * 8 + b2];
        const int P0 = v0 & 0x3fffffff, P1 = v1 & 0x3fffffff, P2 = v2 & 0x3fffffff;
        const size_t pos = (size_t)rowbase[as] + ((size_t)P2 * cc[NE + as] + P1) * cc[as] + P0;
        dst[r] = out.val + pos * (DOF * DOF) + I0p * DOF;
        // the first colour to reach a block stores it (no MatZeroEntries, no read)
        const bool first = ((v0 & v1 & v2) >> 30) & 1;
        if (ok[r] && !first && !(kDebug && (out.debug & 16))) load_run<DOFI * DOF>(dst[r], v[r]);
        else { for (int k = 0; k < DOFI * DOF; ++k) v[r][k] = 0; }
      }
      double kij[GRAM ? RB : 1][GRAM ? DOF * DOF : 1];
      if constexpr (GRAM) {   // K^{ij} = sum_{fg} C^{ij}_{fg} M_fg; C = mat(e_f, e_g) folds to the form's constants
        PtView p0; p0.x = xq; p0.u = fu; p0.ut = fut; p0.gu = fgu; p0.hu = fhu; p0.G = Gq; p0.prm = prm.v; p0.shift = out.shift; p0.t = out.t; p0.normal = nullptr; p0.atboundary = 0; p0.boundary_id = -1;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int k = 0; k < DOF * DOF; ++k) kij[r][k] = 0;
#pragma unroll
        for (int f = 0; f < NFS; ++f)
#pragma unroll
          for (int g = 0; g < NFS; ++g) {
            if (!((PAIRS >> (f * 8 + g)) & 1ull)) continue;
            double ef[NFS], eg[NFS];
#pragma unroll
            for (int k = 0; k < NFS; ++k) { ef[k] = (k == f) ? 1.0 : 0.0; eg[k] = (k == g) ? 1.0 : 0.0; }
            double T[DOF * DOF];
            Form::mat(p0, ef, eg, T);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
              for (int k = 0; k < DOF * DOF; ++k) kij[r][k] += T[k] * acc[fm_pair_index(PAIRS, f, g)][t][rb + r];
          }
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if (!ok[r]) continue;
        const int a = arow0 + (lane >> 4) + 4 * (rb + r);
#pragma unroll
        for (int i = 0; i < DOFI; ++i)
#pragma unroll
          for (int j = 0; j < DOF; ++j) {
            double x;
            if constexpr (GRAM) x = kij[r][i * DOF + j]; else x = acc[i * DOF + j][t][rb + r];
            if (anyfix && (fixflag[a * DOF + I0p + i] || fixflag[b * DOF + j])) x = (a == b && I0p + i == j && !bpass) ? 1.0 : 0.0;   // the unit diagonal comes from the interior pass only
            v[r][i * DOF + j] += x;
          }
        if (!(kDebug && (out.debug & 32))) store_run<DOFI * DOF>(dst[r], v[r]);
      }
      }   // row-group batches
    }
  }
  }   // groups of row fields (FUSE)
  FM_STAMP();
  if constexpr (NS > 0) { if (tid < NS) out.vec[(out.elem_base + blockIdx.x) * NS + tid] = Facc[0]; }
  if (hasV && tid < NE * npv) {   // IGAElementFixSystem / FixFunction on F_e, IGAElementAssembleVec
    const int a = tid / npv, part = tid & (npv - 1);
#pragma unroll
    for (int i = 0; i < DOF; ++i) Facc[i] = group_sum(Facc[i], npv);
    if (part == 0) {
      const size_t row = (size_t)rowid[a];
      double vnew[DOF]; bool any = false;
#pragma unroll
      for (int i = 0; i < DOF; ++i) {
        const int k = a * DOF + i;
        double v = Facc[i];
        // FixSystem / FixFunction act once on the element sums; with the passes in separate launches the interior
        // pass carries the constant parts (flux, fixed value) and a boundary pass only drops its fixed rows
        if (op == OP_SYSTEM) { if (!bpass) v += flux[k]; if (fixflag[k]) v = bpass ? 0.0 : fixval[k]; }                                         // src/petigaelem.c:1371-1387
        else if (op == OP_FUNCTION || op == OP_IFUNCTION) { if (!bpass) v -= flux[k]; if (fixflag[k]) v = bpass ? 0.0 : ufix[k] - fixval[k]; }  // :1449-1461
        vnew[i] = v; any = any || (v != 0.0);
      }
      if (any) {   // (a form without a load term, e.g. Elasticity3D's F = 0, adds nothing away from the Dirichlet faces: no round trip)
        double vold[DOF];
#pragma unroll
        for (int i = 0; i < DOF; ++i) vold[i] = out.vec[row * DOF + i];   // one round trip for the row's fields
#pragma unroll
        for (int i = 0; i < DOF; ++i) out.vec[row * DOF + i] = vold[i] + vnew[i];
      }
    }
  }
  // ---- PENCIL: the tiles of the leaving layer (local axis-0 index lv = 0; the last element of the walk flushes lv = 0..3 in
  // turn) are complete for this pencil.  Two steps through LDS (the Phi region is free by now):
  //   1. every wave copies its leaving accumulator tiles to LDS, [accumulator set][tile slot][16 rows x 17]: tile slots
  //      0..ncolb-1 = "row part" (row layer lv x column layers lv..3), then the "column part" (row layers lv+1..3 x column
  //      layer lv);
  //   2. all threads write them out: one thread per (row slot, column slot) pair and part.  It forms its blocks -- for a
  //      constant-coefficient form K^{ij} = sum_{fg} C^{ij}_{fg} M_fg with M_fg for f > g read from the mirror tile (also
  //      leaving) -- applies IGAElementFixSystem / FixJacobian to the combined values and adds them to the block CSR: the
  //      blocks of a thread in one CSR row are consecutive (the axis-0 neighbours of a row are contiguous), the loads of a
  //      batch of blocks in flight together.
  // A wave's own tiles are 2 of the 7 leaving ones at best: writing from the accumulators left most waves idle at the
  // barrier while two of them paid 4 dependent round trips per tile (measured 45-100k cycles per element, 58 % of the time),
  // and keeping the coefficient transform with the accumulators cost hundreds of spilled registers.  Measured and rejected:
  // a third pass that puts the finished blocks back into LDS in memory order for a lane-coalesced streaming read-add-write
  // (4 loads in flight per lane: 3.9, every load of a part in flight: 4.2 against 5.65 M elements/s on Elasticity3D: more
  // barriers, more spills, and 36 of 64 lanes active per instruction).
  if constexpr (HASM && PENCIL) {
    constexpr int NBK = DOFI * DOF, TS = 16 * 17;
    double *stage = phi;
    const bool last = ei == nwalk - 1;
    const int nlv = last ? 4 : 1;
    for (int lv = 0; lv < nlv; ++lv) {
      __syncthreads();                     // Phi (MFMA / vector phases of this element) or the previous pass's staging is free
      const int ncolb = 4 - lv;            // column layers of the row part = its tile slots
      if (wave_active && !(kDebug && (out.debug & 1))) {
        const int lb = rslot(tb_);
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
          const int la = rslot(ta0 + t);
          const bool rowp = (la == lv) && (lb >= lv), colp = (lb == lv) && (la > lv);
          if (!rowp && !colp) continue;
          const int ts = rowp ? lb - lv : ncolb + (la - lv - 1);
          double *sp = stage + ts * TS + (lane >> 4) * 17 + (lane & 15);
#pragma unroll
          for (int k = 0; k < NACC; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[k * 7 * TS + 4 * 17 * r] = acc[k][t][r];
            acc[k][t] = (fm_d4_t){0, 0, 0, 0};   // the slot is re-used by the layer that enters next
          }
        }
      }
      __syncthreads();
      if (!(kDebug && (out.debug & 1)))
      for (int pr = tid; pr < 512; pr += nthr) {
        const bool rowp = pr < 256;
        const int u = pr & 255, ra = u >> 4, cb = u & 15;
        const int nblk = rowp ? ncolb : 3 - lv;
        const int a1 = ra & 3, a2 = ra >> 2, b1 = cb & 3, b2 = cb >> 2;
        const int v1 = pax[1 * 64 + a1 * 8 + b1], v2 = pax[2 * 64 + a2 * 8 + b2];
        constexpr int KB = (NBK > 4) ? 2 : 4;   // blocks of a thread in flight together (all four: 5.37 vs 5.65 M elements/s on Elasticity3D, spills)
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += KB) {
          double old[KB][NBK]; double *dst[KB];
#pragma unroll
          for (int kk = 0; kk < KB; ++kk) {
            const int kb = k0 + kk;
            if (kb >= nblk) continue;
            const int la = rowp ? lv : lv + 1 + kb, lb = rowp ? lv + kb : lv;
            const int a = la * 16 + ra;
            const int v0 = pax[0 * 64 + la * 8 + lb];
            const int P0 = v0 & 0x3fffffff, P1 = v1 & 0x3fffffff, P2 = v2 & 0x3fffffff;
            const size_t pos = (size_t)rowbase[a] + ((size_t)P2 * cc[NE + a] + P1) * cc[a] + P0;
            dst[kk] = out.val + pos * (DOF * DOF) + I0 * DOF;
            const bool first = ((v0 & v1 & v2) >> 30) & 1;   // the first colour to reach a block stores it (no MatZeroEntries, no read)
            if (!first && !(kDebug && (out.debug & 16))) load_run<NBK>(dst[kk], old[kk]);
            else { for (int k = 0; k < NBK; ++k) old[kk][k] = 0; }
          }
#pragma unroll
          for (int kk = 0; kk < KB; ++kk) {
            const int kb = k0 + kk;
            if (kb >= nblk) continue;
            const int la = rowp ? lv : lv + 1 + kb, lb = rowp ? lv + kb : lv;
            const int a = la * 16 + ra, b = lb * 16 + cb;
            const int ts = rowp ? kb : ncolb + kb;
            // mirror tile (column layer x row layer): the diagonal tile is its own mirror
            const int tm = rowp ? (kb == 0 ? 0 : ncolb + kb - 1) : kb + 1;
            const double *sd = stage + ts * TS + ra * 17 + cb, *sm = stage + tm * TS + cb * 17 + ra;
            double kx[NBK];
            if constexpr (GRAM) {   // K^{ij} = sum_{fg} C^{ij}_{fg} M_fg; C = mat(e_f, e_g) folds to the form's constants
              PtView p0; p0.x = xq; p0.u = fu; p0.ut = fut; p0.gu = fgu; p0.hu = fhu; p0.G = Gq; p0.prm = prm.v; p0.shift = out.shift; p0.t = out.t; p0.normal = nullptr; p0.atboundary = 0; p0.boundary_id = -1;
#pragma unroll
              for (int k = 0; k < NBK; ++k) kx[k] = 0;
#pragma unroll
              for (int f = 0; f < NFS; ++f)
#pragma unroll
                for (int g = 0; g < NFS; ++g) {
                  if (!((PAIRS >> (f * 8 + g)) & 1ull)) continue;
                  double ef[NFS], eg[NFS];
#pragma unroll
                  for (int k = 0; k < NFS; ++k) { ef[k] = (k == f) ? 1.0 : 0.0; eg[k] = (k == g) ? 1.0 : 0.0; }
                  double T[DOF * DOF];
                  Form::mat(p0, ef, eg, T);
                  const double m = (f <= g) ? sd[fm_pair_index(PACC, f, g) * 7 * TS] : sm[fm_pair_index(PACC, g, f) * 7 * TS];
#pragma unroll
                  for (int k = 0; k < NBK; ++k) kx[k] += T[I0 * DOF + k] * m;
                }
            } else {
#pragma unroll
              for (int k = 0; k < NBK; ++k) kx[k] = sd[k * 7 * TS];
            }
            // IGAElementFixSystem / FixJacobian on the combined values: the diagonal of a fixed row is the number of walked
            // elements that hold its node (each sets K_kk = 1)
            if (anyfix) {
              const double held = (double)(min(ei, 3 - la) + 1);
#pragma unroll
              for (int i = 0; i < DOFI; ++i)
#pragma unroll
                for (int j = 0; j < DOF; ++j)
                  if (fixflag[a * DOF + I0 + i] || fixflag[b * DOF + j]) kx[i * DOF + j] = (a == b && I0 + i == j) ? held : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NBK; ++k) old[kk][k] += kx[k];
            if (!(kDebug && (out.debug & 32))) store_run<NBK>(dst[kk], old[kk]);
          }
        }
      }
    }
  }
  FM_STAMP();
  if (PENCIL) {   // next element of the walk: LDS is re-used, the vector rows written above are read by other threads
    if (tid == 0) s_anyfix = 0;
    __syncthreads();
  }
  }   // walk
  if (stamp) out.dbg[31] = nst;
#undef FM_STAMP
#undef FM_SUM_FACTORISE
#undef PS
#undef PHAS
}

}  // namespace igx
