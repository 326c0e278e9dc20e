// coo.hpp -- hand-back of the device matrix / vectors to a PETSc-side adapter (adapter/petiga_amd_petsc.c);
// included by engine.hip.
//
// PETSc assembles device matrices from coordinate lists: MatSetPreallocationCOO(A, n, coo_i, coo_j) once, then
// MatSetValuesCOO(A, v, ADD_VALUES) per assembly with v on the device.  The engine's value array IS such a v (block by
// block, row-major inside a block, src/petigapoint.c:451-462 contract), so the hand-back is the index pair of every stored
// scalar in that order: natural numbering (node * dof + field, axis 0 fastest: what IGA_Grid_LocalIndices lists,
// src/petigagrid.c) or PETSc's own (PetIGA's AO = AOCreateMemoryScalable over the ranks' owned boxes in rank order,
// src/petigagrid.c:185-199: index = first index of the owner rank + position in the owner's box, i fastest).  PETSc then does
// what MatAssemblyBegin/End always did (rows of not-owned nodes travel to their owners), or -- after IGXReduceGhostRows -- the
// not-owned rows are masked with -1, which MatSetPreallocationCOO ignores.
struct NumDev {
  int dof, bs2, numbering, owned_only;
  int nrow[3], ncol[3], nsz[3], P[3];
  const int *rownode[3], *colnode[3], *rowowned[3];   // per axis: row / column index -> global node; row owned on this axis
  const int *oc[3], *lo[3], *lw[3];                   // per axis: node -> owner coordinate, offset inside the owner's box; coordinate -> width
  const int64_t *rstart;                              // rank -> first PETSc node index
};

__device__ __forceinline__ int64_t igx_node_index(const NumDev &N, int n0, int n1, int n2) {
  if (N.numbering == 0) return (int64_t)n0 + (int64_t)N.nsz[0] * ((int64_t)n1 + (int64_t)N.nsz[1] * n2);
  const int c0 = N.oc[0][n0], c1 = N.oc[1][n1], c2 = N.oc[2][n2];
  const int rank = c0 + N.P[0] * (c1 + N.P[1] * c2);
  return N.rstart[rank] + (int64_t)N.lo[0][n0] + (int64_t)N.lw[0][c0] * ((int64_t)N.lo[1][n1] + (int64_t)N.lw[1][c1] * N.lo[2][n2]);
}

// one wavefront per block row
__global__ void k_mat_coo(NumDev N, int64_t nbrows, const int64_t *browptr, const int32_t *bcolidx, int64_t *ci, int64_t *cj) {
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= nbrows) return;
  const int lane = threadIdx.x & 63;
  const int r0 = (int)(r % N.nrow[0]), r1 = (int)((r / N.nrow[0]) % N.nrow[1]), r2 = (int)(r / ((int64_t)N.nrow[0] * N.nrow[1]));
  const bool owned = N.rowowned[0][r0] && N.rowowned[1][r1] && N.rowowned[2][r2];
  const int64_t grow = igx_node_index(N, N.rownode[0][r0], N.rownode[1][r1], N.rownode[2][r2]);
  const int64_t lo = browptr[r], hi = browptr[r + 1];
  const int dof = N.dof;
  for (int64_t e = lo * N.bs2 + lane; e < hi * N.bs2; e += 64) {
    const int64_t blk = e / N.bs2; const int k = (int)(e - blk * N.bs2), i = k / dof, j = k - i * dof;
    const int c = bcolidx[blk];
    const int c0 = c % N.ncol[0], c1 = (c / N.ncol[0]) % N.ncol[1], c2 = c / (N.ncol[0] * N.ncol[1]);
    const int64_t gcol = igx_node_index(N, N.colnode[0][c0], N.colnode[1][c1], N.colnode[2][c2]);
    const bool keep = owned || !N.owned_only;
    ci[e] = keep ? grow * dof + i : -1;
    cj[e] = keep ? gcol * dof + j : -1;
  }
}

__global__ void k_vec_idx(NumDev N, int64_t nbrows, int64_t *idx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbrows * N.dof) return;
  const int64_t r = t / N.dof; const int c = (int)(t - r * N.dof);
  const int r0 = (int)(r % N.nrow[0]), r1 = (int)((r / N.nrow[0]) % N.nrow[1]), r2 = (int)(r / ((int64_t)N.nrow[0] * N.nrow[1]));
  const bool owned = N.rowowned[0][r0] && N.rowowned[1][r1] && N.rowowned[2][r2];
  idx[t] = (owned || !N.owned_only) ? igx_node_index(N, N.rownode[0][r0], N.rownode[1][r1], N.rownode[2][r2]) * N.dof + c : -1;
}

// ghosted local array [gw2][gw1][gw0][dof] (IGAGetLocalVecArray, src/petigavec.c:256-269) -> row-indexed vector
__global__ void k_from_ghosted(int gw0, int gw1, int gw2, int nr0, int nr1, int dof, const int *rm0, const int *rm1, const int *rm2, const double *src, double *dst, int to_ghosted) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)gw0 * gw1 * gw2 * dof;
  if (t >= n) return;
  const int c = (int)(t % dof); int64_t g = t / dof;
  const int i0 = (int)(g % gw0); g /= gw0; const int i1 = (int)(g % gw1), i2 = (int)(g / gw1);
  const int64_t row = (int64_t)rm0[i0] + (int64_t)nr0 * ((int64_t)rm1[i1] + (int64_t)nr1 * rm2[i2]);
  if (to_ghosted) const_cast<double *>(src)[t] = dst[row * dof + c]; else dst[row * dof + c] = src[t];
}

struct NumTables { DevBuf rownode[3], colnode[3], rowowned[3], oc[3], lo[3], lw[3], rstart; };

static int make_numdev(IGX g, int numbering, int owned_only, NumTables &T, NumDev &N) {
  const Space &s = g->s;
  if (numbering != 0 && numbering != 1) return fail(IGX_ERR_ARG_OUTOFRANGE, "numbering must be 0 (natural) or 1 (PETSc)");
  memset(&N, 0, sizeof(N));
  N.dof = s.dof; N.bs2 = s.dof * s.dof; N.numbering = numbering; N.owned_only = owned_only;
  std::vector<std::vector<int>> lwv(3);
  for (int d = 0; d < 3; ++d) {
    const AxisLayout &L = s.lay[d];
    N.nrow[d] = L.nrow; N.ncol[d] = L.ncol; N.nsz[d] = s.node_sizes[d]; N.P[d] = s.proc_sizes[d];
    // node -> owner coordinate / offset (space_setup's ranges: lstart = span[efirst] - p, the last rank owns to the end)
    std::vector<int> oc(s.node_sizes[d], 0), lo(s.node_sizes[d], 0); lwv[d].assign(s.proc_sizes[d], 1);
    if (d < s.dim) {
      const Axis &ax = s.axis[d]; const int np = s.proc_sizes[d], nel = s.elem_sizes[d], p = ax.p;
      for (int c = 0; c < np; ++c) {
        const int q = nel / np, r = nel % np, ew = q + (r > c ? 1 : 0), es = c * q + std::min(c, r), el = es + ew - 1;
        const int lstart = ax.span[es] - p, lend = (c == np - 1) ? ax.nnp : ((el < nel - 1) ? ax.span[el + 1] - p : ax.span[el] + 1);
        lwv[d][c] = lend - lstart;
        for (int n = std::max(lstart, 0); n < lend && n < s.node_sizes[d]; ++n) { oc[n] = c; lo[n] = n - lstart; }
      }
    }
    if (T.rownode[d].upload(L.rownode) || T.colnode[d].upload(L.colnode) || T.rowowned[d].upload(L.owned) || T.oc[d].upload(oc) || T.lo[d].upload(lo) || T.lw[d].upload(lwv[d]))
      return fail(IGX_ERR_MEM, "device allocation failed");
    N.rownode[d] = T.rownode[d].as<int>(); N.colnode[d] = T.colnode[d].as<int>(); N.rowowned[d] = T.rowowned[d].as<int>();
    N.oc[d] = T.oc[d].as<int>(); N.lo[d] = T.lo[d].as<int>(); N.lw[d] = T.lw[d].as<int>();
  }
  std::vector<int64_t> rstart((size_t)s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] + 1, 0);
  { size_t r = 0; for (int c2 = 0; c2 < s.proc_sizes[2]; ++c2) for (int c1 = 0; c1 < s.proc_sizes[1]; ++c1) for (int c0 = 0; c0 < s.proc_sizes[0]; ++c0, ++r) rstart[r + 1] = rstart[r] + (int64_t)lwv[0][c0] * lwv[1][c1] * lwv[2][c2]; }
  if (T.rstart.upload(rstart)) return fail(IGX_ERR_MEM, "device allocation failed");
  N.rstart = T.rstart.as<int64_t>();
  return 0;
}

extern "C" int IGXMatGetCOO(IGXMat A, int numbering, int owned_only, int64_t *coo_i, int64_t *coo_j, int on_device) {
  if (!A || !coo_i || !coo_j) return fail(IGX_ERR_ARG_WRONG, "null argument");
  IGX g = A->iga; NumTables T; NumDev N;
  if (int rc = make_numdev(g, numbering, owned_only, T, N)) return rc;
  const size_t n = (size_t)A->nblocks * N.bs2;
  DevBuf di, dj; int64_t *pi = coo_i, *pj = coo_j;
  if (!on_device) { if (di.alloc(n * 8) || dj.alloc(n * 8)) return fail(IGX_ERR_MEM, "device allocation failed"); pi = di.as<int64_t>(); pj = dj.as<int64_t>(); }
  hipLaunchKernelGGL(k_mat_coo, dim3((unsigned)((A->nbrows + 3) / 4)), dim3(256), 0, g->stream, N, A->nbrows, A->browptr.as<int64_t>(), A->bcolidx.as<int32_t>(), pi, pj);
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(g->stream));
  if (!on_device) { HIPCK(hipMemcpy(coo_i, pi, n * 8, hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(coo_j, pj, n * 8, hipMemcpyDeviceToHost)); }
  return 0;
}

// The same lists, kept on the device by the matrix itself (for MatSetPreallocationCOO of a device Mat type: no host staging of
// 2 x 8 bytes per stored scalar -- 93 GB for the metric configuration -- on this side), in the caller's index width: 8 bytes, or
// 4 when PETSc is built with 32-bit PetscInt (refused when an index does not fit).  Freed by IGXMatFreeCOO / IGXMatDestroy.
__global__ void k_narrow_i64(const int64_t *src, int32_t *dst, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = (int32_t)src[t];
}
extern "C" int IGXMatGetCOODevice(IGXMat A, int numbering, int owned_only, int index_bytes, void **coo_i, void **coo_j) {
  if (!A || !coo_i || !coo_j) return fail(IGX_ERR_ARG_WRONG, "null argument");
  if (index_bytes != 4 && index_bytes != 8) return fail(IGX_ERR_ARG_OUTOFRANGE, "index width must be 4 or 8 bytes");
  IGX g = A->iga; const Space &s = g->s;
  const int64_t nglobal = (int64_t)s.node_sizes[0] * s.node_sizes[1] * s.node_sizes[2] * s.dof;
  if (index_bytes == 4 && nglobal > 0x7fffffffll) return fail(IGX_ERR_ARG_OUTOFRANGE, "row indices do not fit 32 bits: PETSc needs --with-64-bit-indices for this problem");
  const size_t n = (size_t)A->nblocks * s.dof * s.dof;
  DevBuf wi, wj;
  if (wi.alloc(n * 8) || wj.alloc(n * 8)) return fail(IGX_ERR_MEM, "device allocation of the coordinate lists failed");
  if (int rc = IGXMatGetCOO(A, numbering, owned_only, wi.as<int64_t>(), wj.as<int64_t>(), 1)) return rc;
  if (index_bytes == 8) { std::swap(A->coo_i.p, wi.p); std::swap(A->coo_i.bytes, wi.bytes); std::swap(A->coo_j.p, wj.p); std::swap(A->coo_j.bytes, wj.bytes); }
  else {
    if (A->coo_i.alloc(n * 4) || A->coo_j.alloc(n * 4)) return fail(IGX_ERR_MEM, "device allocation of the coordinate lists failed");
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_narrow_i64, dim3(nb), dim3(256), 0, g->stream, wi.as<int64_t>(), A->coo_i.as<int32_t>(), (int64_t)n);
    hipLaunchKernelGGL(k_narrow_i64, dim3(nb), dim3(256), 0, g->stream, wj.as<int64_t>(), A->coo_j.as<int32_t>(), (int64_t)n);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(g->stream));
  }
  *coo_i = A->coo_i.p; *coo_j = A->coo_j.p;
  return 0;
}
// a caller without the HIP runtime headers (the PETSc adapter is plain C) fetches pieces of a device array through this
extern "C" int IGXDeviceToHost(void *host, const void *dev, size_t bytes) {
  if (!host || !dev) return fail(IGX_ERR_ARG_WRONG, "null argument");
  HIPCK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int IGXMatFreeCOO(IGXMat A) { if (!A) return fail(IGX_ERR_ARG_WRONG, "null matrix"); A->coo_i.alloc(0); A->coo_j.alloc(0); return 0; }

extern "C" int IGXVecGetIndices(IGXVec v, int numbering, int owned_only, int64_t *idx, int on_device) {
  if (!v || !idx) return fail(IGX_ERR_ARG_WRONG, "null argument");
  IGX g = v->iga; NumTables T; NumDev N;
  if (int rc = make_numdev(g, numbering, owned_only, T, N)) return rc;
  DevBuf di; int64_t *pi = idx;
  if (!on_device) { if (di.alloc((size_t)v->n * 8)) return fail(IGX_ERR_MEM, "device allocation failed"); pi = di.as<int64_t>(); }
  hipLaunchKernelGGL(k_vec_idx, dim3((unsigned)((v->n + 255) / 256)), dim3(256), 0, g->stream, N, g->nbrows, pi);
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(g->stream));
  if (!on_device) HIPCK(hipMemcpy(idx, pi, (size_t)v->n * 8, hipMemcpyDeviceToHost));
  return 0;
}

// on_device != 0: the copy is a kernel on the IGX's stream and returns without waiting for it -- a caller on another stream (a
// device Vec's array belongs to PETSc's stream) calls IGXSynchronize before it touches the array again, or shares the stream
// (IGXSetStream).  Host arrays are staged and the call returns when the data is in place.
static int ghosted_copy(IGXVec v, double *array, int on_device, int to_ghosted) {
  if (!v || !array) return fail(IGX_ERR_ARG_WRONG, "null argument");
  IGX g = v->iga; const Space &s = g->s;
  if (!to_ghosted) g->slab_valid = false;      // the vector is written after the assembly's face mark (comm.hpp)
  const int64_t n = (int64_t)s.lay[0].gwidth * s.lay[1].gwidth * s.lay[2].gwidth * s.dof;
  DevBuf tmp; double *p = array;
  if (!on_device) { if (tmp.alloc((size_t)n * 8)) return fail(IGX_ERR_MEM, "device allocation failed"); p = tmp.as<double>(); if (!to_ghosted) HIPCK(hipMemcpy(p, array, (size_t)n * 8, hipMemcpyHostToDevice)); }
  hipLaunchKernelGGL(k_from_ghosted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g->stream, s.lay[0].gwidth, s.lay[1].gwidth, s.lay[2].gwidth, s.lay[0].nrow, s.lay[1].nrow, s.dof,
                     g->ab[0].rowmap.as<int>(), g->ab[1].rowmap.as<int>(), g->ab[2].rowmap.as<int>(), p, v->a.as<double>(), to_ghosted);
  HIPCK(hipGetLastError());
  if (!on_device) { HIPCK(hipStreamSynchronize(g->stream)); if (to_ghosted) HIPCK(hipMemcpy(array, p, (size_t)n * 8, hipMemcpyDeviceToHost)); }
  return 0;
}
extern "C" int IGXVecCopyFromGhosted(IGXVec v, const double *array, int on_device) { return ghosted_copy(v, const_cast<double *>(array), on_device, 0); }
extern "C" int IGXVecCopyToGhosted(IGXVec v, double *array, int on_device) { return ghosted_copy(v, array, on_device, 1); }
extern "C" int IGXVecGetGhostedSize(IGXVec v, int64_t *n) { if (!v || !n) return fail(IGX_ERR_ARG_WRONG, "null argument"); const Space &s = v->iga->s; *n = (int64_t)s.lay[0].gwidth * s.lay[1].gwidth * s.lay[2].gwidth * s.dof; return 0; }
