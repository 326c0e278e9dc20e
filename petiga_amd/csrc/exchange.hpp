// exchange.hpp -- ghost-row pack/unpack for the multi-GPU path (one process per GPU); included by engine.hip.
//
// A rank assembles its own elements, so rows of nodes it holds but does not own (the <= p node layers on the
// high side of each axis, src/petiga.c:1172-1208) are partial sums that belong to an "upper" neighbour: the
// reference moves them through PETSc's stash in MatAssemblyBegin/End (src/petigaksp.c:197-198).  Here a row
// of the local matrix always carries the node's full stencil in ascending column order, which is the same
// set and order on sender and receiver, so a ghost row travels as a plain run of doubles and is ADDED to the
// owner's row.  One message per neighbour offset o in {0,1}^3 \ {0}: rows whose axis-d index is in the ghost
// part where o_d = 1 and in the owned part where o_d = 0.  Transport (RCCL send/recv over xGMI) is the
// caller's (petiga_amd/exchange.py).

struct NbrPlan {
  int rank;              // peer
  int off[3];            // neighbour offset: ranks above (send) / below (receive) on every axis; 0 or 1 unless ranks are thinner than p elements
  int start[3], count[3];  // local row-index box on this rank
  int64_t mat_doubles, vec_doubles;
};

// node ranges of the rank with processor coordinates c on axis d (same formulas as space_setup)
static void axis_ranges_of(const Space &s, int d, int c, int &lstart, int &lwidth, int &gstart, int &gwidth) {
  const Axis &ax = s.axis[d];
  const int np = s.proc_sizes[d], N = s.elem_sizes[d], q = N / np, r = N % np;
  const int ew = q + (r > c ? 1 : 0), es = c * q + std::min(c, r);
  const int ef = es, el = es + ew - 1, p = ax.p;
  lstart = ax.span[ef] - p; gstart = lstart;
  const int gend = ax.span[el] + 1;
  const int lend = (el < N - 1) ? ax.span[el + 1] - p : ax.span[el] + 1;
  lwidth = lend - lstart; gwidth = gend - gstart;
  if (c == np - 1) lwidth = ax.nnp - lstart;
}

static int rank_of(const Space &s, const int c[3]) { return c[0] + s.proc_sizes[0] * (c[1] + s.proc_sizes[1] * c[2]); }

// build the send list (send=true: to upper neighbours) or the receive list (from lower neighbours).
// Per axis the ghost layer of a rank (p node layers on a C^{p-1} axis) belongs to the ranks above it: to the next one alone when
// that owns at least as many nodes, to several when ranks are thinner than p elements (src/petiga.c:1172-1208 makes no such
// restriction: PETSc's stash routes every row to its true owner).  So an axis contributes pieces (k, start, count): k = 0 the owned
// part, k >= 1 the part of the ghost layer that rank me + k owns (send) / the part of rank me - k's ghost layer that I own (receive);
// a message per offset vector (k0, k1, k2) != 0.  Ranks are taken unwrapped on a periodic axis (rank C + np = rank C, its nodes
// shifted by the period), so the formulas are the same with and without the wrap.
static std::vector<NbrPlan> neighbour_plans(const Space &s, bool send) {
  std::vector<NbrPlan> out;
  if (!s.setup) return out;
  struct Piece { int k, peer, start, count; };
  std::vector<Piece> pieces[3];
  for (int d = 0; d < 3; ++d) {
    if (d >= s.dim) { pieces[d].push_back({0, 0, 0, 1}); continue; }
    const int np = s.proc_sizes[d], me = s.proc_ranks[d];
    const bool per = s.axis[d].periodic != 0;
    const int period = s.axis[d].nnp;      // distinct nodes of the axis (a periodic axis: its basis functions wrap after this many)
    pieces[d].push_back({0, me, 0, std::min(s.node_lwidth[d], s.node_gwidth[d])});      // owned part (same on both sides)
    if (np == 1) continue;                 // single rank on this axis: nothing to exchange (periodic wraps locally)
    auto mod = [&](int c) { return ((c % np) + np) % np; };
    auto fdiv = [&](int c) { return (c - mod(c)) / np; };
    auto unwrapped = [&](int C, int &LS, int &lw, int &gw) { int ls, gs; axis_ranges_of(s, d, mod(C), ls, lw, gs, gw); LS = ls + fdiv(C) * period; };
    for (int k = 1; k < np; ++k) {
      const int S = send ? me : me - k, R = send ? me + k : me;      // sender and owner, unwrapped
      if (!per && (S < 0 || R >= np)) break;
      int LSs, lws, gws, LSr, lwr, gwr;
      unwrapped(S, LSs, lws, gws); unwrapped(R, LSr, lwr, gwr);
      const int a = std::max(LSs + lws, LSr), b = std::min(LSs + gws, LSr + lwr);      // sender's ghost nodes that the owner owns
      if (LSr >= LSs + gws) break;           // beyond the ghost layer: no further rank holds any of it
      if (b <= a) continue;
      pieces[d].push_back({k, mod(send ? R : S), a - (send ? LSs : LSr), b - a});
    }
  }
  for (const Piece &p2 : pieces[2]) for (const Piece &p1 : pieces[1]) for (const Piece &p0 : pieces[0]) {
    if (p0.k == 0 && p1.k == 0 && p2.k == 0) continue;
    const Piece *pc[3] = {&p0, &p1, &p2};
    NbrPlan pl; int coords[3];
    for (int d = 0; d < 3; ++d) { pl.off[d] = pc[d]->k; pl.start[d] = pc[d]->start; pl.count[d] = pc[d]->count; coords[d] = pc[d]->peer; }
    pl.rank = rank_of(s, coords);
    int64_t t[3];
    for (int d = 0; d < 3; ++d) { t[d] = 0; for (int k = 0; k < pl.count[d]; ++k) t[d] += s.lay[d].rcnt[pl.start[d] + k]; }
    pl.mat_doubles = t[0] * t[1] * t[2] * s.dof * s.dof;
    pl.vec_doubles = (int64_t)pl.count[0] * pl.count[1] * pl.count[2] * s.dof;
    out.push_back(pl);
  }
  return out;
}

struct PackDev { int start[3], count[3], nrow[3]; const int *rcnt[3]; const int64_t *prefix[3]; int64_t tot[3], sub0[3]; int bs2, dof; };

// one workgroup per row of the sub-box.  MODE 0: copy the row's blocks / vector entries into the message; 1: add the
// message to them (ghost-row reduction); 2: overwrite them with the message (ghost-value refresh, vectors)
template <int MODE>
__global__ void k_ghost_rows(PackDev P, const int64_t *browptr, double *val, double *vec, double *buf, int64_t mat_doubles) {
  const int64_t r = blockIdx.x;
  const int k0 = (int)(r % P.count[0]), k1 = (int)((r / P.count[0]) % P.count[1]), k2 = (int)(r / ((int64_t)P.count[0] * P.count[1]));
  const int r0 = P.start[0] + k0, r1 = P.start[1] + k1, r2 = P.start[2] + k2;
  const int64_t row = (int64_t)r0 + (int64_t)P.nrow[0] * ((int64_t)r1 + (int64_t)P.nrow[1] * r2);
  if (val) {
    const int64_t c1 = P.rcnt[1][r1], c2 = P.rcnt[2][r2], c0 = P.rcnt[0][r0];
    // offset inside the message: same closed form as browptr, with prefixes relative to the sub-box
    const int64_t p0 = P.prefix[0][r0] - P.sub0[0], p1 = P.prefix[1][r1] - P.sub0[1], p2 = P.prefix[2][r2] - P.sub0[2];
    const int64_t off = (p2 * P.tot[1] * P.tot[0] + c2 * (p1 * P.tot[0] + c1 * p0)) * P.bs2;
    const int64_t n = c0 * c1 * c2 * P.bs2;
    double *m = val + browptr[row] * P.bs2, *b = buf + off;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { if (MODE == 0) b[i] = m[i]; else if (MODE == 1) m[i] += b[i]; else m[i] = b[i]; }
  }
  if (vec && threadIdx.x < P.dof) {
    double *v = vec + row * P.dof + threadIdx.x, *b = buf + mat_doubles + r * P.dof + threadIdx.x;
    if (MODE == 0) *b = *v; else if (MODE == 1) *v += *b; else *v = *b;
  }
}

// plan list `send_list` (true: upper neighbours / my ghost part, false: lower neighbours / my first owned nodes), entry k
static int ghost_rows(IGX g, IGXMat A, IGXVec b, int k, double *devbuf, bool send_list, int mode, hipStream_t stream_or_null = nullptr) {
  NEEDIGA(g);
  if (int rc = ensure_device(g)) return rc;
  const Space &s = g->s;
  { std::string e; if (int rc = exchange_supported(s, e)) return fail(rc, e); }
  const std::vector<NbrPlan> plans = neighbour_plans(s, send_list);
  if (k < 0 || k >= (int)plans.size()) return fail(IGX_ERR_ARG_OUTOFRANGE, "no such neighbour");
  if (!devbuf) return fail(IGX_ERR_ARG_WRONG, "null buffer");
  if (A && A->iga != g) return fail(IGX_ERR_ARG_WRONG, "matrix created by another IGX");
  if (b && b->iga != g) return fail(IGX_ERR_ARG_WRONG, "vector created by another IGX");
  const NbrPlan &pl = plans[k];
  PackDev P;
  for (int d = 0; d < 3; ++d) {
    P.start[d] = pl.start[d]; P.count[d] = pl.count[d]; P.nrow[d] = s.lay[d].nrow;
    P.rcnt[d] = g->ab[d].rcnt.as<int>(); P.prefix[d] = g->ab[d].prefix.as<int64_t>();
    P.tot[d] = 0; P.sub0[d] = 0;
    for (int r = 0; r < pl.start[d]; ++r) P.sub0[d] += s.lay[d].rcnt[r];
    for (int r = 0; r < pl.count[d]; ++r) P.tot[d] += s.lay[d].rcnt[pl.start[d] + r];
  }
  P.bs2 = s.dof * s.dof; P.dof = s.dof;
  const int64_t nrows = (int64_t)pl.count[0] * pl.count[1] * pl.count[2];
  if (nrows == 0) return 0;
  const int64_t matd = A ? pl.mat_doubles : 0;
  const hipStream_t st = stream_or_null ? stream_or_null : g->stream;
  const int64_t *bp = A ? A->browptr.as<int64_t>() : nullptr; double *vp = A ? A->val.as<double>() : nullptr, *xp = b ? b->a.as<double>() : nullptr;
  if (mode == 0) hipLaunchKernelGGL(k_ghost_rows<0>, dim3((unsigned)nrows), dim3(256), 0, st, P, bp, vp, xp, devbuf, matd);
  else if (mode == 1) hipLaunchKernelGGL(k_ghost_rows<1>, dim3((unsigned)nrows), dim3(256), 0, st, P, bp, vp, xp, devbuf, matd);
  else hipLaunchKernelGGL(k_ghost_rows<2>, dim3((unsigned)nrows), dim3(256), 0, st, P, bp, vp, xp, devbuf, matd);
  if (hipError_t e_ = hipGetLastError(); e_ != hipSuccess)
    return fail(IGX_ERR_LIB, std::string("ghost-row kernel (mode ") + std::to_string(mode) + ", entry " + std::to_string(k) + " of the " + (send_list ? "send" : "receive") + " list, peer " + std::to_string(pl.rank) +
                ", rows " + std::to_string(pl.count[0]) + " x " + std::to_string(pl.count[1]) + " x " + std::to_string(pl.count[2]) + "): " + hipGetErrorString(e_));
  return 0;
}

extern "C" int IGXGetNeighborCount(IGX g, int *nsend, int *nrecv) {
  NEEDIGA(g);
  if (!g->s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  { std::string e; if (int rc = exchange_supported(g->s, e)) return fail(rc, e); }
  if (nsend) *nsend = (int)neighbour_plans(g->s, true).size();
  if (nrecv) *nrecv = (int)neighbour_plans(g->s, false).size();
  return 0;
}
extern "C" int IGXGetNeighborInfo(IGX g, int send, int k, int *rank, int64_t *mat_doubles, int64_t *vec_doubles) {
  NEEDIGA(g);
  { std::string e; if (int rc = exchange_supported(g->s, e)) return fail(rc, e); }
  const std::vector<NbrPlan> plans = neighbour_plans(g->s, send != 0);
  if (k < 0 || k >= (int)plans.size()) return fail(IGX_ERR_ARG_OUTOFRANGE, "no such neighbour");
  if (rank) *rank = plans[k].rank; if (mat_doubles) *mat_doubles = plans[k].mat_doubles; if (vec_doubles) *vec_doubles = plans[k].vec_doubles;
  return 0;
}
extern "C" int IGXPackGhostRows(IGX g, IGXMat A, IGXVec b, int k, double *devbuf) { return ghost_rows(g, A, b, k, devbuf, true, 0); }
extern "C" int IGXUnpackGhostRows(IGX g, IGXMat A, IGXVec b, int k, const double *devbuf) { if (g) g->slab_valid = false; return ghost_rows(g, A, b, k, const_cast<double *>(devbuf), false, 1); }
// the reverse direction, before a nonlinear assembly: the owner's values travel to the ranks that hold the node as a ghost
// (IGAGetLocalVecArray = DMGlobalToLocal, src/petigavec.c:256-269).  Pack: entry k of the RECEIVE list (a lower neighbour,
// whose ghosts are my first owned nodes); unpack: entry k of the SEND list (an upper neighbour, owner of my ghost part).
extern "C" int IGXPackOwnerValues(IGX g, IGXVec v, int k, double *devbuf) { if (!v) return fail(IGX_ERR_ARG_WRONG, "null vector"); return ghost_rows(g, nullptr, v, k, devbuf, false, 0); }
extern "C" int IGXUnpackGhostValues(IGX g, IGXVec v, int k, const double *devbuf) { if (!v) return fail(IGX_ERR_ARG_WRONG, "null vector"); if (g) g->slab_valid = false; return ghost_rows(g, nullptr, v, k, const_cast<double *>(devbuf), true, 2); }
// 1 if this rank owns the row node with local row indices (r0,r1,r2) -- after the exchange only owned rows are final
extern "C" int IGXRowOwned(IGX g, int r0, int r1, int r2) {
  if (!g || !g->s.setup) return 0;
  const int r[3] = {r0, r1, r2};
  for (int d = 0; d < 3; ++d) { if (r[d] < 0 || r[d] >= g->s.lay[d].nrow) return 0; if (!g->s.lay[d].owned[r[d]]) return 0; }
  return 1;
}
