// generic_kernel.hpp -- the general element kernel: any device form, dim 1..3, any degree / continuity,
// dof <= 8, optional NURBS geometry.  One workgroup (256 threads = 4 wavefronts) per element of the
// current colour.  It fuses the reference's per-element stages
//   IGAElementBuildClosure (src/petigaelem.c:693)   -> phase 0  (gathers of X, W, U, V; BC flags)
//   IGAElementBuildTabulation (src/petigaelem.c:794) -> phases 1-2 (K1..K6, straight into LDS)
//   IGAPointFormValue/Grad/Hess (src/petigapoint.c:327) -> phase 3
//   point callback + IGAPointAddMat/AddVec (src/petigapoint.c:451) -> phases 5-6 (register accumulators)
//   IGAElementFixSystem/Function/Jacobian (src/petigaelem.c:1360-1500) -> phases 4,5,6
//   IGAElementAssembleMat/Vec (src/petigaelem.c:1525) -> scatter, conflict-free inside a colour
// The shape-function table Phi[q][a][NF] lives in LDS when it fits (160 KiB/CU) and in a per-workgroup
// HBM scratch slice otherwise (flat addressing serves both).
#pragma once
#include <hip/hip_runtime.h>
#include "forms.hpp"

namespace igx {

struct Carve {          // offsets (in doubles) into the dynamic LDS block; -1 = absent
  int t1d[3], w1d[3];
  int gX, gW, Ue, Ve, ufix, fixval, fixflag, flux;
  int JW, xq, E1, E2, W0, W1, W2, G;
  int X1m, X2m;         // NEED_MAPX: the geometry map's first / second derivatives at the points [nqp][nsd][dim] / [nqp][nsd][dim][dim] (p->mapX[1], [2])
  int E3, W3, d3u;      // order 3 (Form::ORDER >= 3): third-order inverse map, third derivatives of the NURBS denominator and of the fields
  int gA;               // property array of the element's nodes [nen][npd] (IGAElementBuildClosure, src/petigaelem.c:745-752)
  int u, ut, gu, hu, lift, phi;
  int nrm;              // [nqp][DIM] outward unit normals of a boundary-form pass
  int total;            // doubles
};

template <int DIM> __host__ __device__ __forceinline__ constexpr int nfeat(int order) { return order >= 3 ? 1 + DIM + DIM * DIM + DIM * DIM * DIM : (order >= 2 ? 1 + DIM + DIM * DIM : 1 + DIM); }

__device__ __forceinline__ double det3(const double *A, int d) {   // A row-major [d][d]
  if (d == 1) return A[0];
  if (d == 2) return A[0] * A[3] - A[1] * A[2];
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
__device__ __forceinline__ void inv3(const double *A, int d, double det, double *B) {   // closed forms of src/petigainv.f90.in
  if (d == 1) { B[0] = 1 / det; return; }
  if (d == 2) { B[0] = A[3] / det; B[1] = -A[1] / det; B[2] = -A[2] / det; B[3] = A[0] / det; return; }
  B[0] = (A[4] * A[8] - A[5] * A[7]) / det; B[1] = (A[2] * A[7] - A[1] * A[8]) / det; B[2] = (A[1] * A[5] - A[2] * A[4]) / det;
  B[3] = (A[5] * A[6] - A[3] * A[8]) / det; B[4] = (A[0] * A[8] - A[2] * A[6]) / det; B[5] = (A[2] * A[3] - A[0] * A[5]) / det;
  B[6] = (A[3] * A[7] - A[4] * A[6]) / det; B[7] = (A[1] * A[6] - A[0] * A[7]) / det; B[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}

// parametric tensor-product basis value + derivatives of basis function a=(a0,a1,a2) at point q=(q0,q1,q2)
// (K2, src/petiga3d.F90:32-233) read from the LDS copies of the three 1-D rows.
template <int DIM, bool SECOND, bool THIRD = false>
__device__ __forceinline__ void tensor_basis(const double *const t[3], const int na[3], const int aq[3], const int qq[3],
                                             double &b0, double *b1, double *b2, double *b3 = nullptr) {
  double n[3][4];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d < DIM) {
      const double *r = t[d] + (qq[d] * na[d] + aq[d]) * NDER;
      n[d][0] = r[0]; n[d][1] = r[1]; n[d][2] = SECOND ? r[2] : 0.0; n[d][3] = THIRD ? r[3] : 0.0;
    } else { n[d][0] = 1; n[d][1] = 0; n[d][2] = 0; n[d][3] = 0; }
  }
  b0 = n[0][0] * n[1][0] * n[2][0];
#pragma unroll
  for (int i = 0; i < DIM; ++i) b1[i] = n[0][i == 0] * n[1][i == 1] * n[2][i == 2];
  if (SECOND) {
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j)
        b2[i * DIM + j] = n[0][(i == 0) + (j == 0)] * n[1][(i == 1) + (j == 1)] * n[2][(i == 2) + (j == 2)];
  }
  if constexpr (THIRD) {      // K2 at order 3 (src/petiga3d.F90:32-233): N3(c,b,a) = the product of the 1-D derivatives of the orders the three indices count
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int k = 0; k < DIM; ++k)
          b3[(i * DIM + j) * DIM + k] = n[0][(i == 0) + (j == 0) + (k == 0)] * n[1][(i == 1) + (j == 1) + (k == 1)] * n[2][(i == 2) + (j == 2) + (k == 2)];
  }
}

// Rationalize (src/petigarat.f90.in:3-57) of one basis function at one point: b_k in, R_k out (in place); w its weight, W_k the
// derivatives of the denominator there.  Orders 2 and 3 use the rationalised lower orders.
template <int DIM, bool SECOND, bool THIRD>
__device__ __forceinline__ void rationalize(double w, double w0, const double *W1, const double *W2, const double *W3, double &b0, double *b1, double *b2, double *b3) {
  const double r0 = w * b0 / w0;
  double r1[3];
  for (int i = 0; i < DIM; ++i) r1[i] = (w * b1[i] - r0 * W1[i]) / w0;
  if (SECOND)
    for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j)
      b2[i * DIM + j] = (w * b2[i * DIM + j] - r0 * W2[i * DIM + j] - r1[i] * W1[j] - r1[j] * W1[i]) / w0;
  if constexpr (THIRD)
    for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) for (int k = 0; k < DIM; ++k)
      b3[(i * DIM + j) * DIM + k] = (w * b3[(i * DIM + j) * DIM + k] - r0 * W3[(i * DIM + j) * DIM + k]
                                     - r1[i] * W2[j * DIM + k] - r1[j] * W2[i * DIM + k] - r1[k] * W2[i * DIM + j]
                                     - b2[j * DIM + k] * W1[i] - b2[i * DIM + k] * W1[j] - b2[i * DIM + j] * W1[k]) / w0;
  b0 = r0; for (int i = 0; i < DIM; ++i) b1[i] = r1[i];
}

// Local numbering of an element's basis functions in the kernels' LDS arrays.  Natural: a = a0 + na0*(a1 + na1*a2)
// (IGAElementBuildClosure, src/petigaelem.c:711-719).  WALK (feature kernel's pencil mode, 4x4x4 functions): the axis-0 index
// is the 16x16 MFMA tile, a = 16*a0 + a1 + 4*a2, so that a tile pairs two axis-0 node layers.
template <bool WALK> __device__ __forceinline__ int slot_of(int a0, int a1, int a2, const int na[3]) { return WALK ? (a0 * 16 + a1 + 4 * a2) : (a0 + na[0] * (a1 + na[1] * a2)); }
template <bool WALK> __device__ __forceinline__ void slot_decode(int a, const int na[3], int aa[3]) {
  if (WALK) { aa[0] = a >> 4; aa[1] = a & 3; aa[2] = (a >> 2) & 3; }
  else { aa[0] = a % na[0]; aa[1] = (a / na[0]) % na[1]; aa[2] = a / (na[0] * na[1]); }
}

// BoundaryArea, geometry branch (src/petigaelem.c:1133-1160 -> IGA_BoundaryArea_2D/3D, src/petiga2d.F90:276-346,
// src/petiga3d.F90:379-464): dS = sum over the Gauss points of the element's face of w * sqrt|det(F F^T)|, F = d(face map)/du
// from the first / last layer of the element's control points; raw Gauss weights (sum 2 per axis), derivatives w.r.t. the
// knot coordinate, the routine's own Rationalize.  Called by the few threads of a face element that carry a load.
template <int DIM, bool WALK = false>
__device__ inline double face_dS(const double *const t1d[3], const double *const w1d[3], const int nq[3], const int na[3],
                                 const double *gX, const double *gW, bool rat, int dir, int side, int nsd = DIM) {
  if constexpr (DIM == 1) return 1.0;
  else {
    constexpr int FD = DIM - 1;
    int ax[2] = {0, 0};
    { int k = 0; for (int i = 0; i < DIM; ++i) if (i != dir) ax[k++] = i; }
    const int n0 = na[ax[0]], n1 = (FD == 2) ? na[ax[1]] : 1, q0n = nq[ax[0]], q1n = (FD == 2) ? nq[ax[1]] : 1;
    auto ctrl = [&](int a0, int a1) { int loc[3] = {0, 0, 0}; loc[dir] = side ? na[dir] - 1 : 0; loc[ax[0]] = a0; if (FD == 2) loc[ax[1]] = a1; return slot_of<WALK>(loc[0], loc[1], loc[2], na); };
    double dS = 0;
    for (int q1 = 0; q1 < q1n; ++q1) for (int q0 = 0; q0 < q0n; ++q0) {
      double W0 = 1, S1[2] = {0, 0};
      if (rat) {
        W0 = 0;
        for (int a1 = 0; a1 < n1; ++a1) for (int a0 = 0; a0 < n0; ++a0) {
          const double *r0 = t1d[ax[0]] + (q0 * n0 + a0) * NDER; const double v1 = (FD == 2) ? t1d[ax[1]][(q1 * n1 + a1) * NDER] : 1.0, d1 = (FD == 2) ? t1d[ax[1]][(q1 * n1 + a1) * NDER + 1] : 0.0;
          const double w = gW[ctrl(a0, a1)];
          W0 += w * r0[0] * v1; S1[0] += w * r0[1] * v1; S1[1] += w * r0[0] * d1;
        }
      }
      double F[2][3] = {{0, 0, 0}, {0, 0, 0}};
      for (int a1 = 0; a1 < n1; ++a1) for (int a0 = 0; a0 < n0; ++a0) {
        const double *r0 = t1d[ax[0]] + (q0 * n0 + a0) * NDER; const double v1 = (FD == 2) ? t1d[ax[1]][(q1 * n1 + a1) * NDER] : 1.0, d1 = (FD == 2) ? t1d[ax[1]][(q1 * n1 + a1) * NDER + 1] : 0.0;
        const int la = ctrl(a0, a1);
        double N0 = r0[0] * v1, N1[2] = {r0[1] * v1, r0[0] * d1};
        if (rat) { const double w = gW[la]; N0 = w * N0 / W0; for (int r = 0; r < FD; ++r) N1[r] = (w * N1[r] - N0 * S1[r]) / W0; }
        for (int r = 0; r < FD; ++r) for (int c = 0; c < nsd; ++c) F[r][c] += N1[r] * gX[la * nsd + c];
      }
      double M[2][2] = {{0, 0}, {0, 0}};
      for (int r = 0; r < FD; ++r) for (int s = 0; s < FD; ++s) for (int c = 0; c < nsd; ++c) M[r][s] += F[r][c] * F[s][c];
      const double det = (FD == 1) ? M[0][0] : M[0][0] * M[1][1] - M[0][1] * M[1][0];
      dS += sqrt(fabs(det)) * w1d[ax[0]][q0] * ((FD == 2) ? w1d[ax[1]][q1] : 1.0);
    }
    return dS;
  }
}

// IGAElementBuildFix's flux part on a mapped geometry (AddFlux with BoundaryArea's geometry branch): run after the closure
// gathers are visible.  aa-decoding as in the kernels: a = a0 + na0*(a1 + na1*a2).
template <int DIM, int DOF, bool WALK = false>
__device__ inline void add_mapped_flux(const SpaceDev &S, const int ID[3], const int el[3], const double *const t1d[3], const double *const w1d[3],
                                       const int nq[3], const int na[3], const double *gX, const double *gW, bool rat, double *flux, int tid, int nthr, int nsd = DIM) {
  const int NE = na[0] * na[1] * na[2];
  for (int d = 0; d < DIM; ++d) {
    if (S.ax[d].periodic) continue;
    for (int side = 0; side < 2; ++side) {
      const BCDev &bl = S.bcl[d][side];
      if (!bl.count || ID[d] != (side ? S.ax[d].esizes - 1 : 0)) continue;
      for (int a = tid; a < NE; a += nthr) {
        int aa[3]; slot_decode<WALK>(a, na, aa);
        if (aa[d] != (side ? na[d] - 1 : 0)) continue;
        double A = 1;
        for (int i = 0; i < DIM; ++i) if (i != d) A *= S.ax[i].J[el[i]] / (double)na[i];
        A *= face_dS<DIM, WALK>(t1d, w1d, nq, na, gX, gW, rat, d, side, nsd);
        for (int k = 0; k < bl.count; ++k) { const int c = bl.field[k]; if (c < DOF) flux[a * DOF + c] += bl.value[k] * A; }
      }
    }
  }
}

template <class Form, int DIM>
__global__ void __launch_bounds__(256)
generic_assemble(SpaceDev S, ParamsDev prm, OutDev out, ColorRange cr, Carve cv, double *phi_global, size_t phi_stride) {
  constexpr int DOF = Form::DOF;
  constexpr bool SECOND = Form::ORDER >= 2, THIRD = Form::ORDER >= 3;      // (IGASetOrder: the tabulation goes as far as the form says it reads)
  constexpr int NF = nfeat<DIM>(Form::ORDER);
  constexpr int D2 = DIM * DIM, D3 = D2 * DIM;
  constexpr int NS = nscalar_of<Form>::v;   // > 0: a scalar functional (OP_SCALAR), no matrix / vector phases
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;

  // ---- which element
  int el[3], ID[3], off[3], nq[3], na[3];
  {
    int b = blockIdx.x;
    const int t0 = b % cr.count[0]; b /= cr.count[0];
    const int t1 = b % cr.count[1]; b /= cr.count[1];
    const int tt[3] = {t0, t1, b};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      el[d] = cr.start[d] + tt[d] * cr.step[d];
      ID[d] = el[d] + S.ax[d].estart;
      off[d] = S.ax[d].off[el[d]];
      nq[d] = S.ax[d].nqp; na[d] = S.ax[d].nen;
    }
  }
  // boundary-form pass (IGAElementNextForm, src/petigaelem.c:427-447): one point on the face axis, basis from the end-of-axis
  // table (src/petigaelem.c:796-823), weight 1, bnd_detJac 1
  const int bid = out.bid; const bool bpass = bid >= 0;
  const int baxis = bpass ? (bid >> 1) : -1, bside = bid & 1;
  if (bpass) nq[baxis] = 1;
  constexpr bool HASB = has_boundary_of<Form>::v;
  const int NQ = nq[0] * nq[1] * nq[2], NE = na[0] * na[1] * na[2];
  const int op = out.op;
  const bool hasM = (op == OP_SYSTEM || op == OP_MATRIX || op == OP_JACOBIAN || op == OP_IJACOBIAN);
  const bool hasV = (op == OP_SYSTEM || op == OP_VECTOR || op == OP_FUNCTION || op == OP_IFUNCTION);
  const bool useU = out.U != nullptr, useV = out.V != nullptr;
  const bool geo = S.nsd > 0, rat = S.rational != 0;
  // dim != nsd (a curve or a surface in space, IGASetGeometryDim: demo/ClassicalShell.c:154): the geometry map is tabulated
  // (IGA_GeometryMap, src/petigaval.F90:10-43), the inverse map and the physical shape functions are not -- p->shape stays the
  // parametric basis, detJac keeps the parametric measure, a face's normal is the axis (src/petigaelem.c:966-1029) -- and the form
  // builds its metric from p->mapX[1], p->mapX[2] (NEED_MAPX: p.X1, p.X2)
  const int nsd = geo ? S.nsd : DIM;
  const bool emb = geo && nsd != DIM;

  double *t1d[3] = {smem + cv.t1d[0], smem + cv.t1d[1], smem + cv.t1d[2]};
  double *w1d[3] = {smem + cv.w1d[0], smem + cv.w1d[1], smem + cv.w1d[2]};
  double *gX = smem + cv.gX, *gW = smem + cv.gW, *Ue = smem + cv.Ue, *Ve = smem + cv.Ve;
  double *ufix = smem + cv.ufix, *fixval = smem + cv.fixval, *flux = smem + cv.flux;
  int *fixflag = reinterpret_cast<int *>(smem + cv.fixflag);
  double *JW = smem + cv.JW, *xq = smem + cv.xq, *E1 = smem + cv.E1, *E2 = smem + cv.E2;
  double *W0 = smem + cv.W0, *W1 = smem + cv.W1, *W2 = smem + cv.W2, *Gq = smem + cv.G;
  double *X1m = smem + cv.X1m, *X2m = smem + cv.X2m;
  double *E3 = smem + cv.E3, *W3 = smem + cv.W3, *fd3u = smem + cv.d3u, *gA = smem + cv.gA;
  const int npd = S.npd;
  double *fu = smem + cv.u, *fut = smem + cv.ut, *fgu = smem + cv.gu, *fhu = smem + cv.hu, *lift = smem + cv.lift, *nrm = smem + cv.nrm;
  double *phi = (cv.phi >= 0) ? smem + cv.phi : phi_global + (size_t)blockIdx.x * phi_stride;
  __shared__ int s_anyfix;
  if (tid == 0) s_anyfix = 0;
  __syncthreads();

  // ---- phase 0: 1-D rows of this element (LDS-staged knot-span tables), closure gathers, BC flags
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int n = nq[d] * na[d] * NDER;
    const bool face = (d == baxis);
    const double *src = face ? S.ax[d].bnd + (size_t)bside * n : S.ax[d].tab + (size_t)el[d] * n;
    for (int i = tid; i < n; i += nthr) t1d[d][i] = src[i];
    for (int i = tid; i < nq[d]; i += nthr) w1d[d][i] = face ? 1.0 : S.ax[d].w[el[d] * nq[d] + i];
  }
  const int gw0 = S.ax[0].gwidth, gw1 = S.ax[1].gwidth;
  const int nr0 = S.ax[0].nrow, nr1 = S.ax[1].nrow;
  for (int a = tid; a < NE; a += nthr) {
    const int a0 = a % na[0], a1 = (a / na[0]) % na[1], a2 = a / (na[0] * na[1]);
    const int i0 = off[0] + a0, i1 = off[1] + a1, i2 = off[2] + a2;
    const size_t g = (size_t)i0 + (size_t)gw0 * ((size_t)i1 + (size_t)gw1 * (size_t)i2);
    const size_t row = (size_t)S.ax[0].rowmap[i0] + (size_t)nr0 * ((size_t)S.ax[1].rowmap[i1] + (size_t)nr1 * (size_t)S.ax[2].rowmap[i2]);
    if (geo) for (int c = 0; c < nsd; ++c) gX[a * nsd + c] = S.X[g * nsd + c];
    if (rat) gW[a] = S.W[g];
    for (int c = 0; c < npd; ++c) gA[a * npd + c] = S.A[g * npd + c];
    if (useU) for (int c = 0; c < DOF; ++c) Ue[a * DOF + c] = out.U[row * DOF + c];
    if (useV) for (int c = 0; c < DOF; ++c) Ve[a * DOF + c] = out.V[row * DOF + c];
    // IGAElementBuildFix (src/petigaelem.c:1214-1283): faces in (dir, side) order, later faces override
    const int aa[3] = {a0, a1, a2};
    for (int c = 0; c < DOF; ++c) { fixflag[a * DOF + c] = 0; fixval[a * DOF + c] = 0; flux[a * DOF + c] = 0; }
    if (op != OP_MATRIX && op != OP_VECTOR && op != OP_SCALAR) {   // IGAComputeScalar reads U as it is (no FixValues)
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
          if (bl.count && !(geo && DIM > 1) && !bpass) {   // BoundaryArea, no-geometry branch (src/petigaelem.c:1118-1132); mapped: add_mapped_flux below
            double A = 1;
            if (DIM > 1) {
              for (int i = 0; i < DIM; ++i) if (i != d) A *= S.ax[i].J[el[i]] / (double)na[i];
              A *= (DIM == 2) ? 2 : 4;
            }
            for (int k = 0; k < bl.count; ++k) { const int c = bl.field[k]; if (c < DOF) flux[a * DOF + c] += bl.value[k] * A; }
          }
        }
      }
    }
  }
  __syncthreads();
  const bool anyfix = s_anyfix != 0;
  if (geo && DIM > 1 && !bpass && op != OP_MATRIX && op != OP_VECTOR && op != OP_SCALAR) add_mapped_flux<DIM, DOF>(S, ID, el, t1d, w1d, nq, na, gX, gW, rat, flux, tid, nthr, nsd);
  // IGAElementFixValues / DelValues (src/petigaelem.c:1327-1358)
  if (anyfix && (useU || useV)) {
    for (int k = tid; k < NE * DOF; k += nthr)
      if (fixflag[k]) { if (useU) { ufix[k] = Ue[k]; Ue[k] = fixval[k]; } if (useV) Ve[k] = 0.0; }
    __syncthreads();
  }

  // ---- phase 1: per-point geometry (K1, K3 sums, K4, K5)
  double Jel = 1;
#pragma unroll
  for (int d = 0; d < 3; ++d) if (d != baxis) Jel *= S.ax[d].J[el[d]];
  for (int q = tid; q < NQ; q += nthr) {
    const int qq[3] = {q % nq[0], (q / nq[0]) % nq[1], q / (nq[0] * nq[1])};
    double detX = 1.0;
    double w0 = 1, w1[3] = {0, 0, 0}, w2[9] = {0}, w3[THIRD ? 27 : 1] = {0};
    double x0[3], X1[9], X2[27], X3[THIRD ? 81 : 1];
#pragma unroll
    for (int d = 0; d < DIM; ++d) x0[d] = (d == baxis) ? S.ax[d].bndpt[bside] : S.ax[d].pt[el[d] * nq[d] + qq[d]];
    if (rat) {
      w0 = 0;
      for (int a = 0; a < NE; ++a) {
        const int aq[3] = {a % na[0], (a / na[0]) % na[1], a / (na[0] * na[1])};
        double b0, b1[3], b2[9], b3[THIRD ? 27 : 1];
        tensor_basis<DIM, SECOND, THIRD>(t1d, na, aq, qq, b0, b1, b2, b3);
        const double w = gW[a];
        w0 += w * b0;
        for (int i = 0; i < DIM; ++i) w1[i] += w * b1[i];
        if (SECOND) for (int i = 0; i < D2; ++i) w2[i] += w * b2[i];
        if constexpr (THIRD) for (int i = 0; i < D3; ++i) w3[i] += w * b3[i];
      }
      W0[q] = w0;
      for (int i = 0; i < DIM; ++i) W1[q * DIM + i] = w1[i];
      if (SECOND) for (int i = 0; i < D2; ++i) W2[q * D2 + i] = w2[i];
      if constexpr (THIRD) for (int i = 0; i < D3; ++i) W3[q * D3 + i] = w3[i];
    }
    if (geo) {
      for (int i = 0; i < 3; ++i) x0[i] = 0;
      for (int i = 0; i < 9; ++i) X1[i] = 0;
      if (SECOND) for (int i = 0; i < 27; ++i) X2[i] = 0;
      if constexpr (THIRD) for (int i = 0; i < 81; ++i) X3[i] = 0;
      for (int a = 0; a < NE; ++a) {
        const int aq[3] = {a % na[0], (a / na[0]) % na[1], a / (na[0] * na[1])};
        double b0, b1[3], b2[9], b3[THIRD ? 27 : 1];
        tensor_basis<DIM, SECOND, THIRD>(t1d, na, aq, qq, b0, b1, b2, b3);
        if (rat) rationalize<DIM, SECOND, THIRD>(gW[a], w0, w1, w2, w3, b0, b1, b2, b3);   // Rationalize, src/petigarat.f90.in:3-57
        for (int i = 0; i < nsd; ++i) {      // GeometryMap, src/petigamapgeo.f90.in:3-71: X_k(i, :) = sum_a X(i, a) M_k(:, a), i < nsd
          const double x = gX[a * nsd + i];
          x0[i] += x * b0;
          for (int al = 0; al < DIM; ++al) X1[i * DIM + al] += x * b1[al];
          if (SECOND) for (int f = 0; f < D2; ++f) X2[i * D2 + f] += x * b2[f];
          if constexpr (THIRD) for (int f = 0; f < D3; ++f) X3[i * D3 + f] += x * b3[f];
        }
      }
      if (Form::NEED & NEED_MAPX) {
        for (int i = 0; i < nsd * DIM; ++i) X1m[q * nsd * DIM + i] = X1[i];
        if (SECOND) for (int i = 0; i < nsd; ++i) for (int f = 0; f < D2; ++f) X2m[(q * nsd + i) * D2 + f] = X2[i * D2 + f];
      }
      double e1[9] = {0};
      if (!emb) {
      detX = det3(X1, DIM);
      inv3(X1, DIM, detX, e1);           // e1[al][i] = du_al/dx_i
      for (int i = 0; i < D2; ++i) E1[q * D2 + i] = e1[i];
      if (SECOND) {   // InverseMap order 2, src/petigamapinv.f90.in:32-45: E2[c][i][j] = -X2[k][a][b] e1[a][i] e1[b][j] e1[c][k]
        for (int c = 0; c < DIM; ++c) for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) {
          double s = 0;
          for (int k = 0; k < DIM; ++k) for (int a = 0; a < DIM; ++a) for (int b = 0; b < DIM; ++b)
            s -= X2[k * D2 + a * DIM + b] * e1[a * DIM + i] * e1[b * DIM + j] * e1[c * DIM + k];
          E2[(q * DIM + c) * D2 + i * DIM + j] = s;
        }
      }
      if constexpr (THIRD) {   // InverseMap order 3, src/petigamapinv.f90.in:49-60: the third derivatives of the parametric coordinates
        const double *e2 = E2 + (size_t)q * DIM * D2;      // [c][i][j], written by this thread just above
        for (int dd = 0; dd < DIM; ++dd) for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) for (int k = 0; k < DIM; ++k) {
          double s = 0;
          for (int a = 0; a < DIM; ++a) for (int b = 0; b < DIM; ++b) for (int l = 0; l < DIM; ++l) {
            double t = 0;
            for (int c = 0; c < DIM; ++c) t += X3[l * D3 + (a * DIM + b) * DIM + c] * e1[a * DIM + i] * e1[b * DIM + j] * e1[c * DIM + k];
            t += X2[l * D2 + a * DIM + b] * (e1[a * DIM + i] * e2[b * D2 + j * DIM + k] + e1[b * DIM + j] * e2[a * D2 + i * DIM + k] + e1[b * DIM + k] * e2[a * D2 + i * DIM + j]);
            s -= t * e1[dd * DIM + l];
          }
          E3[((size_t)q * DIM + dd) * D3 + (i * DIM + j) * DIM + k] = s;
        }
      }
      if (!(detX > 0.0)) atomicExch(out.errflag, IGX_ERR_USER);   // src/petigaelem.c:989-993
      }      // !emb
    }
    if (bpass) {   // K7: IGA_GetNormal, src/petigaval.F90:45-99; detJac *= detS instead of detX (src/petigaelem.c:1012-1029)
      double n[3] = {0, 0, 0}, dS = 1;
      if (!geo || DIM == 1 || emb) n[baxis] = 1.0;      // (dim != nsd: src/petigaelem.c:1017-1020)
      else if (DIM == 3) {
        const int r1 = (baxis + 1) % 3, r2 = (baxis + 2) % 3;
        const double s0 = X1[0 * DIM + r1], s1 = X1[1 * DIM + r1], s2 = X1[2 * DIM + r1];
        const double t0 = X1[0 * DIM + r2], t1 = X1[1 * DIM + r2], t2 = X1[2 * DIM + r2];
        n[0] = s1 * t2 - s2 * t1; n[1] = s2 * t0 - s0 * t2; n[2] = s0 * t1 - s1 * t0;
        dS = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] /= dS; n[1] /= dS; n[2] /= dS;
      } else {
        double t0, t1;
        if (baxis == 0) { t0 = +X1[0 * DIM + 1]; t1 = +X1[1 * DIM + 1]; } else { t0 = -X1[0 * DIM + 0]; t1 = -X1[1 * DIM + 0]; }
        n[0] = +t1; n[1] = -t0;
        dS = sqrt(n[0] * n[0] + n[1] * n[1]);
        n[0] /= dS; n[1] /= dS;
      }
      for (int i = 0; i < nsd; ++i) nrm[q * nsd + i] = bside ? n[i] : -n[i];
      detX = dS;
    }
    double w = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) w *= w1d[d][qq[d]];
    JW[q] = (Jel * detX) * w;          // detJac[q] *= detX[q]; JW = detJac*weight (src/petigaelem.c:1024, petigapoint.c:461)
    for (int i = 0; i < nsd; ++i) xq[q * nsd + i] = x0[i];
    if (Form::NEED & NEED_G) {           // IGAPointFormInvGradGeomMap, src/petigapoint.c:269-294
      if (emb) {      // IGA_GetInvGradGeomMap, src/petigaval.F90:124-142: G = ((F^T F)^-1 F^T)^T with F = X1 [nsd][dim]
        double M[9] = {0}, Mi[9];
        for (int a = 0; a < DIM; ++a) for (int b = 0; b < DIM; ++b) { double t = 0; for (int i = 0; i < nsd; ++i) t += X1[i * DIM + a] * X1[i * DIM + b]; M[a * DIM + b] = t; }
        inv3(M, DIM, det3(M, DIM), Mi);
        for (int a = 0; a < DIM; ++a) for (int i = 0; i < nsd; ++i) { double t = 0; for (int b = 0; b < DIM; ++b) t += Mi[a * DIM + b] * X1[i * DIM + b]; Gq[(q * DIM + a) * nsd + i] = t / S.ax[a].J[el[a]]; }
      } else
      for (int a = 0; a < DIM; ++a) for (int i = 0; i < DIM; ++i) {
        const double L = S.ax[a].J[el[a]];
        Gq[q * D2 + a * DIM + i] = geo ? E1[q * D2 + a * DIM + i] / L : ((a == i) ? 1 / L : 0.0);
      }
    }
  }
  __syncthreads();

  // ---- phase 2: shape functions Phi[q][a][:] (K2 -> K3 -> K6)
  for (int idx = tid; idx < NQ * NE; idx += nthr) {
    const int q = idx / NE, a = idx - q * NE;
    const int qq[3] = {q % nq[0], (q / nq[0]) % nq[1], q / (nq[0] * nq[1])};
    const int aq[3] = {a % na[0], (a / na[0]) % na[1], a / (na[0] * na[1])};
    double b0, b1[3], b2[9], b3[THIRD ? 27 : 1];
    tensor_basis<DIM, SECOND, THIRD>(t1d, na, aq, qq, b0, b1, b2, b3);
    if (rat) rationalize<DIM, SECOND, THIRD>(gW[a], W0[q], W1 + q * DIM, W2 + q * D2, W3 + q * D3, b0, b1, b2, b3);
    double *o = phi + (size_t)idx * NF;
    o[0] = b0;
    if (!geo || emb) {
      for (int i = 0; i < DIM; ++i) o[1 + i] = b1[i];
      if (SECOND) for (int i = 0; i < D2; ++i) o[1 + DIM + i] = b2[i];
      if constexpr (THIRD) for (int i = 0; i < D3; ++i) o[1 + DIM + D2 + i] = b3[i];
    } else {   // ShapeFunctions, src/petigamapshf.f90.in:30-58
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
      if constexpr (THIRD) {   // ShapeFunctions order 3, src/petigamapshf.f90.in:60-72
        const double *e2 = E2 + (size_t)q * DIM * D2, *e3 = E3 + (size_t)q * DIM * D3;
        for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) for (int k = 0; k < DIM; ++k) {
          double s = 0;
          for (int al = 0; al < DIM; ++al) {
            for (int be = 0; be < DIM; ++be) {
              for (int ga = 0; ga < DIM; ++ga) s += b3[(al * DIM + be) * DIM + ga] * e1[al * DIM + i] * e1[be * DIM + j] * e1[ga * DIM + k];
              s += b2[al * DIM + be] * (e1[al * DIM + i] * e2[be * D2 + j * DIM + k] + e1[be * DIM + j] * e2[al * D2 + i * DIM + k] + e1[be * DIM + k] * e2[al * D2 + i * DIM + j]);
            }
            s += b1[al] * e3[al * D3 + (i * DIM + j) * DIM + k];
          }
          o[1 + DIM + D2 + (i * DIM + j) * DIM + k] = s;
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 3: field values at the points (src/petigaval.F90:182-232)
  if (Form::NEED & (NEED_U | NEED_UT | NEED_GU | NEED_HU | NEED_D3U)) {
    for (int idx = tid; idx < NQ * DOF; idx += nthr) {
      const int q = idx / DOF, c = idx - q * DOF;
      double u = 0, ut = 0, g[3] = {0, 0, 0}, h[9] = {0}, t3[THIRD ? 27 : 1] = {0};
      for (int a = 0; a < NE; ++a) {
        const double *f = phi + ((size_t)q * NE + a) * NF;
        const double Ua = useU ? Ue[a * DOF + c] : 0.0;
        u += f[0] * Ua;
        if (useV) ut += f[0] * Ve[a * DOF + c];
        if (Form::NEED & NEED_GU) for (int i = 0; i < DIM; ++i) g[i] += f[1 + i] * Ua;
        if (SECOND && (Form::NEED & NEED_HU)) for (int i = 0; i < D2; ++i) h[i] += f[1 + DIM + i] * Ua;
        if constexpr (THIRD) if (Form::NEED & NEED_D3U) for (int i = 0; i < D3; ++i) t3[i] += f[1 + DIM + D2 + i] * Ua;      // IGAPointFormDer3, src/petigapoint.c (IGA_GetDer3)
      }
      fu[idx] = u; fut[idx] = ut;
      if (Form::NEED & NEED_GU) for (int i = 0; i < DIM; ++i) fgu[idx * DIM + i] = g[i];
      if (SECOND && (Form::NEED & NEED_HU)) for (int i = 0; i < D2; ++i) fhu[idx * D2 + i] = h[i];
      if constexpr (THIRD) if (Form::NEED & NEED_D3U) for (int i = 0; i < D3; ++i) fd3u[idx * D3 + i] = t3[i];
    }
  }
  if constexpr (NS > 0) {
    // ---- IGAComputeScalar (src/petigacomp.c:35-98): S_e = sum_q JW[q] * scalar(q), one partial row per element
    __syncthreads();
    for (int q = tid; q < NQ; q += nthr) {
      PtView p;
      p.x = xq + q * nsd; p.u = fu + q * DOF; p.ut = fut + q * DOF; p.gu = fgu + q * DOF * DIM; p.hu = fhu + q * DOF * D2;
      p.G = Gq + q * DIM * nsd; p.nsd = nsd; p.X1 = X1m + q * nsd * DIM; p.X2 = X2m + q * nsd * D2; p.d3u = fd3u + q * DOF * D3; p.property = gA; p.npd = npd; p.shape = phi + (size_t)q * NE * NF; p.nen = NE; p.nf = NF; p.prm = prm.v; p.shift = out.shift; p.t = out.t; p.normal = bpass ? nrm + q * nsd : nullptr; p.atboundary = bpass ? 1 : 0; p.boundary_id = bid;
      double Sq[NS];
      Form::scalar(p, Sq);
      const double jw = JW[q];
#pragma unroll
      for (int i = 0; i < NS; ++i) lift[q * NS + i] = Sq[i] * jw;
    }
    __syncthreads();
    if (tid < NS) {   // fixed summation order: bitwise repeatable
      double s = 0;
      for (int q = 0; q < NQ; ++q) s += lift[q * NS + tid];
      out.vec[(out.elem_base + blockIdx.x) * NS + tid] = s;
    }
  } else {
  // ---- phase 4: Dirichlet lifting features: lift[q][j][:] = sum_b fixed(b,j) v_bj Phi[q][b][:]
  const bool dolift = anyfix && op == OP_SYSTEM;
  if (dolift) {
    for (int idx = tid; idx < NQ * DOF * NF; idx += nthr) {
      const int f = idx % NF, j = (idx / NF) % DOF, q = idx / (NF * DOF);
      double s = 0;
      for (int b = 0; b < NE; ++b) if (fixflag[b * DOF + j]) s += fixval[b * DOF + j] * phi[((size_t)q * NE + b) * NF + f];
      lift[idx] = s;
    }
  }
  __syncthreads();

  auto point = [&](int q) {
    PtView p;
    p.x = xq + q * nsd; p.u = fu + q * DOF; p.ut = fut + q * DOF; p.gu = fgu + q * DOF * DIM; p.hu = fhu + q * DOF * D2;
    p.G = Gq + q * DIM * nsd; p.nsd = nsd; p.X1 = X1m + q * nsd * DIM; p.X2 = X2m + q * nsd * D2; p.d3u = fd3u + q * DOF * D3; p.property = gA; p.npd = npd; p.shape = phi + (size_t)q * NE * NF; p.nen = NE; p.nf = NF; p.prm = prm.v; p.shift = out.shift; p.t = out.t; p.normal = bpass ? nrm + q * nsd : nullptr; p.atboundary = bpass ? 1 : 0; p.boundary_id = bid;
    return p;
  };
  const int W0s = 2 * S.ax[0].p + 1, W1s = 2 * S.ax[1].p + 1, W2s = 2 * S.ax[2].p + 1;

  // ---- phase 5: K_e = sum_q JW k_q  (register accumulators), BC rows/cols, coloured scatter
  if (hasM) {
    constexpr int PP = (DOF * DOF <= 4) ? 4 : (DOF * DOF <= 9 ? 2 : 1);   // pairs per thread per pass
    const int npairs = NE * NE;
    for (int base = 0; base < npairs; base += nthr * PP) {
      double acc[PP][DOF * DOF];
#pragma unroll
      for (int s = 0; s < PP; ++s)
#pragma unroll
        for (int i = 0; i < DOF * DOF; ++i) acc[s][i] = 0;
      for (int q = 0; q < NQ; ++q) {
        const PtView p = point(q);
        const double jw = JW[q];
#pragma unroll
        for (int s = 0; s < PP; ++s) {
          const int pr = base + s * nthr + tid;
          if (pr < npairs) {
            const int a = pr / NE, b = pr - a * NE;
            double T[DOF * DOF];
            if constexpr (HASB) { if (bpass) Form::bmat(p, phi + ((size_t)q * NE + a) * NF, phi + ((size_t)q * NE + b) * NF, T); else Form::mat(p, phi + ((size_t)q * NE + a) * NF, phi + ((size_t)q * NE + b) * NF, T); }
            else Form::mat(p, phi + ((size_t)q * NE + a) * NF, phi + ((size_t)q * NE + b) * NF, T);
#pragma unroll
            for (int i = 0; i < DOF * DOF; ++i) acc[s][i] += T[i] * jw;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < PP; ++s) {
        const int pr = base + s * nthr + tid;
        if (pr >= npairs) continue;
        const int a = pr / NE, b = pr - a * NE;
        if (anyfix && op != OP_MATRIX) {   // zero fixed rows / columns, unit diagonal (src/petigaelem.c:1377-1387, :1493-1499)
#pragma unroll
          for (int i = 0; i < DOF; ++i)
#pragma unroll
            for (int j = 0; j < DOF; ++j)
              if (fixflag[a * DOF + i] || fixflag[b * DOF + j]) acc[s][i * DOF + j] = (a == b && i == j && !bpass) ? 1.0 : 0.0;   // the unit diagonal comes from the interior pass only
        }
        const int a0 = a % na[0], a1 = (a / na[0]) % na[1], a2 = a / (na[0] * na[1]);
        const int b0 = b % na[0], b1 = (b / na[0]) % na[1], b2 = b / (na[0] * na[1]);
        const int i0 = off[0] + a0, i1 = off[1] + a1, i2 = off[2] + a2;
        const int r0 = S.ax[0].rowmap[i0], r1 = S.ax[1].rowmap[i1], r2 = S.ax[2].rowmap[i2];
        const int c0 = S.ax[0].rcnt[r0], c1 = S.ax[1].rcnt[r1];
        const int P0 = S.ax[0].P[i0 * W0s + (b0 - a0 + S.ax[0].p)];
        const int P1 = S.ax[1].P[i1 * W1s + (b1 - a1 + S.ax[1].p)];
        const int P2 = S.ax[2].P[i2 * W2s + (b2 - a2 + S.ax[2].p)];
        const size_t row = (size_t)r0 + (size_t)nr0 * ((size_t)r1 + (size_t)nr1 * (size_t)r2);
        const size_t pos = (size_t)out.browptr[row] + ((size_t)P2 * c1 + P1) * c0 + P0;
        double *dst = out.val + pos * (DOF * DOF);
#pragma unroll
        for (int i = 0; i < DOF * DOF; ++i) dst[i] += acc[s][i];
      }
    }
  }

  // ---- phase 6: F_e, BC fix-up, coloured scatter
  if (hasV) {
    for (int a = tid; a < NE; a += nthr) {
      double F[DOF];
#pragma unroll
      for (int i = 0; i < DOF; ++i) F[i] = 0;
      for (int q = 0; q < NQ; ++q) {
        const PtView p = point(q);
        const double *Na = phi + ((size_t)q * NE + a) * NF;
        double R[DOF];
        if constexpr (HASB) { if (bpass) Form::bvec(p, Na, R); else Form::vec(p, Na, R); }
        else Form::vec(p, Na, R);
        if (dolift) {   // F[i] -= K[i][k] v_k, summed through the lifting features (mat is linear in Nb)
#pragma unroll
          for (int j = 0; j < DOF; ++j) {
            double T[DOF * DOF];
            if constexpr (HASB) { if (bpass) Form::bmat(p, Na, lift + ((size_t)q * DOF + j) * NF, T); else Form::mat(p, Na, lift + ((size_t)q * DOF + j) * NF, T); }
            else Form::mat(p, Na, lift + ((size_t)q * DOF + j) * NF, T);
#pragma unroll
            for (int i = 0; i < DOF; ++i) R[i] -= T[i * DOF + j];
          }
        }
        const double jw = JW[q];
#pragma unroll
        for (int i = 0; i < DOF; ++i) F[i] += R[i] * jw;
      }
      const int a0 = a % na[0], a1 = (a / na[0]) % na[1], a2 = a / (na[0] * na[1]);
      const size_t row = (size_t)S.ax[0].rowmap[off[0] + a0] + (size_t)nr0 * ((size_t)S.ax[1].rowmap[off[1] + a1] + (size_t)nr1 * (size_t)S.ax[2].rowmap[off[2] + a2]);
#pragma unroll
      for (int i = 0; i < DOF; ++i) {
        const int k = a * DOF + i;
        double v = F[i];
        // FixSystem / FixFunction act once on the sum of the passes; in separate launches the interior pass carries the constant
        // parts (flux, fixed value) and a boundary pass only drops its fixed rows
        if (op == OP_SYSTEM) { if (!bpass) v += flux[k]; if (fixflag[k]) v = bpass ? 0.0 : fixval[k]; }                        // src/petigaelem.c:1371-1387
        else if (op == OP_FUNCTION || op == OP_IFUNCTION) { if (!bpass) v -= flux[k]; if (fixflag[k]) v = bpass ? 0.0 : ufix[k] - fixval[k]; }   // :1449-1461
        out.vec[row * DOF + i] += v;
      }
    }
  }
  }   // NS == 0
}

// deterministic two-stage sum of the per-element partial rows part[n][NS] -> res[NS]
static __global__ void __launch_bounds__(256) k_sum_partials(const double *part, int64_t n, int ns, double *res, int64_t chunk) {
  __shared__ double red[256];
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  for (int c = 0; c < ns; ++c) {
    double s = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += part[i * ns + c];
    red[threadIdx.x] = s; __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) res[(int64_t)blockIdx.x * ns + c] = red[0];
    __syncthreads();
  }
}

}  // namespace igx
