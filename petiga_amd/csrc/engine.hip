// engine.hip -- device side of libpetiga_amd: table upload, sparsity pattern, kernel dispatch, C ABI.
// Reference citations are file:line of dalcinl/PetIGA @ 2025-04-04.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>
#include "igx.hpp"
#include "generic_kernel.hpp"
#include "feature_mfma.hpp"
#include "first_touch.hpp"
#if defined(IGX_TU_DISPATCH) && IGX_TU_DIM == 3
#include "vec_sumfact.hpp"
#define IGX_HAVE_VEC_SUMFACT 1
#endif
#ifndef IGX_TU_DISPATCH
#include "gram_mfma.hpp"
#include "gram_patch.hpp"
#include "block_pencil.hpp"      // (the host launcher, for run-time forms: rtc.hpp; no kernel of it is instantiated in this unit)
#include "band_pt.hpp"           // (likewise: band_pt_run)
#elif IGX_TU_DIM == 3 && (IGX_TU_GROUP < 0 || IGX_TU_GROUP == 1)
#include "band_pt.hpp"           // (includes block_pencil.hpp; the constant-coefficient multi-field forms take band_pt on a mapped geometry)
#define IGX_HAVE_BLOCK_PENCIL 1
#define IGX_HAVE_BAND_PT 1
#elif IGX_TU_DIM == 3 && IGX_TU_GROUP == 2
#include "band_pt.hpp"
#define IGX_HAVE_BAND_PT 1
#endif

using namespace igx;

// The library is built from several translation units of this one source (see petiga_amd/build.py): the main unit
// (C ABI, set-up, drivers) and, compiled with -DIGX_TU_DISPATCH -DIGX_TU_DIM=d -DIGX_TU_GROUP=g, units that hold only
// the kernel instantiations of one dimension / form group.  The last-error text is one object for all of them.
inline std::string &igx_err_slot() { static thread_local std::string e; return e; }
#define g_err igx_err_slot()
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
constexpr int IGX_NOT_MINE = -12345;    // a dispatch unit's answer for a form of another group
#ifndef IGX_TU_GROUP
#define IGX_TU_GROUP -1
#endif
#define HIPCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(IGX_ERR_LIB, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------ objects
struct DevBuf {
  void *p = nullptr; size_t bytes = 0;
  DevBuf() = default;
  // owns its allocation: movable, not copyable (a std::vector<DevBuf> that grows must MOVE its elements -- copied, the old elements'
  // destructors freed the memory the new ones pointed to: the exchange buffers of a rank whose refresh and reduction lists differ in
  // length, i.e. 4 and 8 ranks with a nonlinear form, before round 4)
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { if (p) (void)hipFree(p); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { if (p) { (void)hipFree(p); p = nullptr; } bytes = n; if (!n) return 0; return hipMalloc(&p, n) == hipSuccess ? 0 : 1; }
  template <class T> int upload(const std::vector<T> &v) {
    if (alloc(v.size() * sizeof(T))) return 1;
    if (v.empty()) return 0;
    return hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

struct AxisBufs { DevBuf tab, w, J, pt, off, rowmap, rcnt, P, rcol, prefix, bnd; };


struct IgxComm;
struct RtcForm;
struct _p_IGX {
  Space s;
  bool on_device = false;
  AxisBufs ab[3];
  DevBuf X, W, A, fixtable, errflag, scratch;
  hipStream_t stream = nullptr;
  int kernel_choice = 0;
  _p_IGX() { s.env = read_env_switches(); kernel_choice = s.env.kernel; }   // environment switches are read here, once per IGX
  std::string last_kernel = "none";
  std::string rtc_note;           // why a run-time form was kept off a kernel it asked for (appended to the kernel name)
  bool timing = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // step begin, kernels begin/end, step end, dominant kernel begin/end
  double last_total_ms = 0, last_kernel_ms = 0; int last_launches = 0;
  std::function<void()> zero_matrix;   // MatZeroEntries of the running IGXCompute*, called by the kernel path that needs it
  std::function<void()> slab_done;     // set during an assembly with a communicator: marks "upper face of axis 2 assembled" on the engine stream
  DevBuf partials, dbgbuf, clkbuf;   // IGXComputeScalar: per-element partial sums + reduction stages
  DomInfo dom;
  int64_t nbrows = 0, nblocks = 0;
  std::shared_ptr<IgxComm> comm;   // transport of the ghost-row exchange (comm.hpp)
  // recorded on the engine stream when the elements next to the upper face of axis 2 have been assembled (pencil kernel, several
  // ranks): IGXReduceGhostRows starts the messages of that face behind it, under the launches of the other elements
  // slab_valid: bit 0 = the mark of axis 2 (slab_ev), bit 1 = axis 1, bit 2 = axis 0 (face_ev[1], face_ev[0]: the pencil walk's three passes)
  hipEvent_t slab_ev = nullptr; int slab_valid = 0; IGXMat slab_A = nullptr; IGXVec slab_b = nullptr;
  hipEvent_t face_ev[2] = {nullptr, nullptr};
  std::function<void(int)> face_done;  // marks "upper face of axis 1 / 0 assembled" (gram_mfma.hpp)
  std::shared_ptr<RtcForm> rtc; std::string rtc_source, rtc_name;   // run-time compiled user form (rtc.hpp)
  std::shared_ptr<RtcForm> rtc_scalar;                              // ... and the last user functional (IGXComputeScalarSource)
};

struct _p_IGXMat {
  IGX iga; int bs; int64_t nbrows, nblocks;
  DevBuf browptr, bcolidx, val;
  DevBuf coo_i, coo_j;     // coordinate lists kept on the device for the hand-back (IGXMatGetCOODevice), or empty
};
struct _p_IGXVec { IGX iga; int64_t n; DevBuf a; };

#ifndef IGX_TU_DISPATCH
extern "C" const char *IGXGetLastError(void) { return g_err.c_str(); }
#endif

#ifndef IGX_TU_DISPATCH
// ------------------------------------------------------------------ set-up mirror
extern "C" int IGXCreate(IGX *iga) { if (!iga) return fail(IGX_ERR_ARG_WRONG, "null pointer"); *iga = new _p_IGX(); igx_pool_acquire(); return 0; }
extern "C" int IGXDestroy(IGX *iga) {
  if (!iga || !*iga) return 0;
  (*iga)->comm.reset();       // (synchronises its exchange stream first)
  for (auto &e : (*iga)->ev) if (e) (void)hipEventDestroy(e);
  if ((*iga)->slab_ev) (void)hipEventDestroy((*iga)->slab_ev);
  for (int k = 0; k < 2; ++k) if ((*iga)->face_ev[k]) (void)hipEventDestroy((*iga)->face_ev[k]);
  delete *iga; *iga = nullptr;
  igx_pool_release();          // the last one out hands the launch-scoped pool back to the driver
  return 0;
}
#define NEEDIGA(g) do { if (!(g)) return fail(IGX_ERR_ARG_WRONG, "null IGX"); } while (0)
#define AXISCK(g, i) do { NEEDIGA(g); if ((i) < 0 || (i) >= 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "Index must be in range [0,2]"); } while (0)
static void touch(IGX g) { g->s.setup = false; g->on_device = false; }
static void drop_net(IGX g) { g->s.netX.clear(); g->s.netW.clear(); g->s.net_nsd = 0; g->s.netA.clear(); g->s.npd = 0; }

extern "C" int IGXSetDim(IGX g, int dim) { NEEDIGA(g); if (dim < 1 || dim > 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of parametric dimensions must be in range [1,3]"); g->s.dim = dim; touch(g); return 0; }
extern "C" int IGXSetDof(IGX g, int dof) { NEEDIGA(g); if (dof < 1) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of DOFs per node must be greater than one"); if (dof > MAXBC) return fail(IGX_ERR_SUP, "device path supports dof <= 8"); g->s.dof = dof; touch(g); return 0; }
extern "C" int IGXSetOrder(IGX g, int order) { NEEDIGA(g); if (order < 0) return fail(IGX_ERR_ARG_OUTOFRANGE, "Order must be nonnegative"); g->s.order = order < 1 ? 1 : (order > 4 ? 4 : order); return 0; }
extern "C" int IGXSetQuadrature(IGX g, int i, int q) { AXISCK(g, i); if (q == IGX_DECIDE && g->s.axis[i].p > 0) q = g->s.axis[i].p + 1; if (q <= 0) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of quadrature points must be positive"); g->s.rule_nqp[i] = q; touch(g); return 0; }
extern "C" int IGXSetRuleType(IGX g, int i, IGXRuleType type) {
  AXISCK(g, i);
  if ((int)type < 0 || (int)type > 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "unknown rule type");
  if (g->s.rule[i].type == (int)type) return 0;
  if (type == IGX_RULE_USER) return fail(IGX_ERR_ARG_WRONGSTATE, "a user-defined rule is set with IGXSetRule");
  g->s.rule[i].type = (int)type; g->s.rule[i].x.clear(); g->s.rule[i].w.clear(); touch(g); return 0;
}
extern "C" int IGXSetRuleSize(IGX g, int i, int nqp) {
  AXISCK(g, i);
  if (nqp < 1) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of quadrature points must be greater than zero");
  if (g->s.rule[i].type == IGX_RULE_USER && (int)g->s.rule[i].x.size() != nqp) return fail(IGX_ERR_ARG_WRONGSTATE, "the size of a user-defined rule is set with IGXSetRule");
  g->s.rule_nqp[i] = nqp; touch(g); return 0;
}
extern "C" int IGXSetRule(IGX g, int i, int q, const double x[], const double w[]) {
  AXISCK(g, i);
  if (q < 1) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of quadrature points must be greater than zero");
  if (!x || !w) return fail(IGX_ERR_ARG_WRONG, "null rule");
  g->s.rule[i].type = IGX_RULE_USER; g->s.rule[i].x.assign(x, x + q); g->s.rule[i].w.assign(w, w + q); g->s.rule_nqp[i] = q; touch(g); return 0;
}
extern "C" int IGXGetRule(IGX g, int i, int *q, double x[], double w[]) {
  AXISCK(g, i);
  const int n = g->s.rule_nqp[i] > 0 ? g->s.rule_nqp[i] : g->s.axis[i].p + 1;      // src/petigabasis.c:103
  if (q) *q = n;
  if (!x && !w) return 0;
  std::vector<double> X(std::max(n, 10)), W(std::max(n, 10));
  std::string e;
  if (int rc = rule_setup(g->s.rule[i], n, X.data(), W.data(), e)) return fail(rc, e);
  for (int k = 0; k < n; ++k) { if (x) x[k] = X[k]; if (w) w[k] = W[k]; }
  return 0;
}
extern "C" int IGXSetProcessors(IGX g, int i, int n) { AXISCK(g, i); g->s.proc_req[i] = n; touch(g); return 0; }
extern "C" int IGXSetComm(IGX g, int size, int rank) { NEEDIGA(g); if (size < 1 || rank < 0 || rank >= size) return fail(IGX_ERR_ARG_OUTOFRANGE, "bad communicator size/rank"); g->s.comm_size = size; g->s.comm_rank = rank; touch(g); return 0; }
extern "C" int IGXAxisSetDegree(IGX g, int i, int p) { AXISCK(g, i); if (p < 1) return fail(IGX_ERR_ARG_OUTOFRANGE, "Polynomial degree must be greater than zero"); if (p > 7) return fail(IGX_ERR_SUP, "degree > 7 not supported"); g->s.axis[i].p = p; touch(g); drop_net(g); return 0; }
extern "C" int IGXAxisSetPeriodic(IGX g, int i, int flag) { AXISCK(g, i); g->s.axis[i].periodic = flag ? 1 : 0; touch(g); return 0; }
extern "C" int IGXAxisInitUniform(IGX g, int i, int N, double Ui, double Uf, int C) { AXISCK(g, i); std::string e; int rc = axis_init_uniform(g->s.axis[i], N, Ui, Uf, C, e); touch(g); drop_net(g); return rc ? fail(rc, e) : 0; }
extern "C" int IGXAxisSetKnots(IGX g, int i, int m, const double U[]) { AXISCK(g, i); if (!U) return fail(IGX_ERR_ARG_WRONG, "null knots"); std::string e; int rc = axis_set_knots(g->s.axis[i], m, U, e); touch(g); drop_net(g); return rc ? fail(rc, e) : 0; }
static int apply_geometry(IGX g);
static int apply_property(IGX g);
extern "C" int IGXSetUp(IGX g) {
  NEEDIGA(g); std::string e; int rc = space_setup(g->s, e); g->on_device = false;
  if (rc) return fail(rc, e);
  if (int rp = apply_property(g)) return rp;
  return apply_geometry(g);   // a control net given by IGXRead / an earlier IGXSetGeometry survives re-partitioning
}

// copy the ghosted-local box of the global control net (what IGALoadGeometry's scatters produce, src/petigaio.c:240-275)
static int apply_geometry(IGX g) {
  Space &s = g->s;
  const int nsd = s.net_nsd;
  if (!nsd) return 0;
  if (nsd < s.dim || nsd > 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of space dimensions must be in range [dim,3]");
  int gs[3] = {1, 1, 1};
  for (int i = 0; i < s.dim; ++i) gs[i] = s.axis[i].span[s.axis[i].nel - 1] + 1;
  const size_t nnet = (size_t)gs[0] * gs[1] * gs[2];
  if (s.netX.size() != nnet * nsd) return fail(IGX_ERR_ARG_WRONG, "control net does not match the knot vectors");
  const bool hasW = !s.netW.empty();
  const int *g0 = s.node_gstart, *gw = s.node_gwidth;
  s.geomX.assign((size_t)gw[0] * gw[1] * gw[2] * nsd, 0.0);
  s.geomW.clear(); if (hasW) s.geomW.assign((size_t)gw[0] * gw[1] * gw[2], 0.0);
  size_t pos = 0;
  for (int k = g0[2]; k < g0[2] + gw[2]; ++k) for (int j = g0[1]; j < g0[1] + gw[1]; ++j) for (int i = g0[0]; i < g0[0] + gw[0]; ++i, ++pos) {
    const size_t gi = (size_t)i + (size_t)gs[0] * ((size_t)j + (size_t)gs[1] * (size_t)k);
    for (int c = 0; c < nsd; ++c) s.geomX[pos * nsd + c] = s.netX[gi * nsd + c];
    if (hasW) s.geomW[pos] = s.netW[gi];
  }
  s.nsd = nsd; s.rational = hasW ? 1 : 0;
  g->on_device = false;
  return 0;
}

// ... and of the property array (IGALoadProperty's scatters, src/petigaio.c:393-458)
static int apply_property(IGX g) {
  Space &s = g->s;
  s.propA.clear();
  if (!s.npd) return 0;
  int gs[3] = {1, 1, 1};
  for (int i = 0; i < s.dim; ++i) gs[i] = s.axis[i].span[s.axis[i].nel - 1] + 1;
  const size_t nnet = (size_t)gs[0] * gs[1] * gs[2], npd = (size_t)s.npd;
  if (s.netA.size() != nnet * npd) return fail(IGX_ERR_ARG_WRONG, "property array does not match the knot vectors");
  const int *g0 = s.node_gstart, *gw = s.node_gwidth;
  s.propA.assign((size_t)gw[0] * gw[1] * gw[2] * npd, 0.0);
  size_t pos = 0;
  for (int k = g0[2]; k < g0[2] + gw[2]; ++k) for (int j = g0[1]; j < g0[1] + gw[1]; ++j) for (int i = g0[0]; i < g0[0] + gw[0]; ++i, ++pos) {
    const size_t gi = (size_t)i + (size_t)gs[0] * ((size_t)j + (size_t)gs[1] * (size_t)k);
    for (size_t c = 0; c < npd; ++c) s.propA[pos * npd + c] = s.netA[gi * npd + c];
  }
  g->on_device = false;
  return 0;
}
// IGASetPropertyDim + the array IGALoadProperty fills (src/petigaio.c:359-458): npd numbers per node of the geometry grid, natural order; npd = 0 drops it
extern "C" int IGXSetProperty(IGX g, int npd, const double A[]) {
  NEEDIGA(g); Space &s = g->s;
  if (!s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  if (npd < 0) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of properties must be nonnegative");
  if (npd > 64) return fail(IGX_ERR_SUP, "device path supports at most 64 properties per node");
  if (npd == 0) { s.npd = 0; s.netA.clear(); return apply_property(g); }
  if (!A) return fail(IGX_ERR_ARG_WRONG, "null property array");
  size_t nnet = 1;
  for (int i = 0; i < s.dim; ++i) nnet *= (size_t)(s.axis[i].span[s.axis[i].nel - 1] + 1);
  s.netA.assign(A, A + nnet * (size_t)npd); s.npd = npd;
  return apply_property(g);
}
extern "C" int IGXGetPropertyDim(IGX g, int *npd) { NEEDIGA(g); if (!npd) return fail(IGX_ERR_ARG_WRONG, "null pointer"); *npd = g->s.npd; return 0; }

extern "C" int IGXSetGeometry(IGX g, int nsd, const double X[], const double W[]) {
  NEEDIGA(g); Space &s = g->s;
  if (!s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  if (nsd < s.dim || nsd > 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of space dimensions must be in range [dim,3]");      // (IGASetGeometryDim, src/petigaio.c:187: [1,3]; below dim there is no map)
  if (!X) return fail(IGX_ERR_ARG_WRONG, "null control points");
  size_t nnet = 1;
  for (int i = 0; i < s.dim; ++i) nnet *= (size_t)(s.axis[i].span[s.axis[i].nel - 1] + 1);
  s.netX.assign(X, X + nnet * nsd);
  s.netW.clear(); if (W) s.netW.assign(W, W + nnet);
  s.net_nsd = nsd;
  return apply_geometry(g);
}

static int bc_set(BC &bc, int field, double value) {   // src/petigaform.c:100-110
  int k; for (k = 0; k < bc.count; ++k) if (bc.field[k] == field) break;
  if (k == bc.count) bc.count++;
  bc.field[k] = field; bc.value[k] = value; return 0;
}
static int bc_args(IGX g, int axis, int side, int field) {
  if (axis < 0 || axis >= (g->s.dim > 0 ? g->s.dim : 3)) return fail(IGX_ERR_ARG_OUTOFRANGE, "Expecting 0<=axis<dim");
  if (side < 0 || side >= 2) return fail(IGX_ERR_ARG_OUTOFRANGE, "Expecting 0<=side<2");
  if (field < 0 || field >= (g->s.dof > 0 ? g->s.dof : 64)) return fail(IGX_ERR_ARG_OUTOFRANGE, "Expecting 0<=field<dof");
  return 0;
}
extern "C" int IGXSetBoundaryValue(IGX g, int axis, int side, int field, double v) { NEEDIGA(g); if (int rc = bc_args(g, axis, side, field)) return rc; return bc_set(g->s.value[axis][side], field, v); }
extern "C" int IGXSetBoundaryLoad(IGX g, int axis, int side, int field, double v) { NEEDIGA(g); if (int rc = bc_args(g, axis, side, field)) return rc; return bc_set(g->s.load[axis][side], field, v); }
extern "C" int IGXClearBoundary(IGX g) { NEEDIGA(g); for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) { g->s.value[a][s].count = 0; g->s.load[a][s].count = 0; g->s.visit[a][s] = false; } return 0; }
extern "C" int IGXSetBoundaryForm(IGX g, int axis, int side, int flag) {   // IGASetBoundaryForm -> IGAFormSetBoundaryForm, src/petigaform.c:134
  NEEDIGA(g);
  if (axis < 0 || axis >= 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "Index must be in range [0,2]");
  if (side < 0 || side >= 2) return fail(IGX_ERR_ARG_OUTOFRANGE, "Index must be in range [0,1]");
  g->s.visit[axis][side] = flag != 0; return 0;
}
extern "C" int IGXSetForm(IGX g, IGXFormKind kind, const double params[], int nparams) {
  NEEDIGA(g);
  if (nparams < 0 || nparams > MAXPARAM) return fail(IGX_ERR_ARG_OUTOFRANGE, "too many form parameters");
  g->s.form = kind; g->s.params.assign(params ? params : nullptr, params ? params + nparams : nullptr);
  return 0;
}
extern "C" int IGXGetSizes(IGX g, int es[3], int est[3], int ew[3], int ns[3], int nls[3], int nlw[3], int ngs[3], int ngw[3]) {
  NEEDIGA(g); const Space &s = g->s;
  if (!s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  for (int i = 0; i < 3; ++i) {
    if (es) es[i] = s.elem_sizes[i]; if (est) est[i] = s.elem_start[i]; if (ew) ew[i] = s.elem_width[i];
    if (ns) ns[i] = s.node_sizes[i]; if (nls) nls[i] = s.node_lstart[i]; if (nlw) nlw[i] = s.node_lwidth[i];
    if (ngs) ngs[i] = s.node_gstart[i]; if (ngw) ngw[i] = s.node_gwidth[i];
  }
  return 0;
}
extern "C" int IGXGetBasis(IGX g, int i, int *nel, int *nqp, int *nen, int offset[], double detJac[], double weight[], double point[], double value[]) {
  AXISCK(g, i); const Space &s = g->s;
  if (!s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  const Basis1D &b = s.basis[i];
  if (nel) *nel = b.nel; if (nqp) *nqp = b.nqp; if (nen) *nen = b.nen;
  if (offset) std::copy(b.offset.begin(), b.offset.end(), offset);
  if (detJac) std::copy(b.detJac.begin(), b.detJac.end(), detJac);
  if (weight) std::copy(b.weight.begin(), b.weight.end(), weight);
  if (point) std::copy(b.point.begin(), b.point.end(), point);
  if (value) std::copy(b.value.begin(), b.value.end(), value);
  return 0;
}
extern "C" int IGXGetProcessors(IGX g, int ps[3], int pr[3]) { NEEDIGA(g); for (int i = 0; i < 3; ++i) { if (ps) ps[i] = g->s.proc_sizes[i]; if (pr) pr[i] = g->s.proc_ranks[i]; } return 0; }
extern "C" int64_t IGXGetElementCount(IGX g) { if (!g || !g->s.setup) return 0; return (int64_t)g->s.elem_width[0] * g->s.elem_width[1] * g->s.elem_width[2]; }
extern "C" int IGXGetColoring(IGX g, int nc[3]) { NEEDIGA(g); if (!g->s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first"); for (int i = 0; i < 3; ++i) nc[i] = g->s.lay[i].ncolors; return 0; }
extern "C" int IGXGetElementColor(IGX g, int axis, int e) { if (!g || !g->s.setup || axis < 0 || axis > 2 || e < 0 || e >= (int)g->s.lay[axis].color.size()) return -1; return g->s.lay[axis].color[e]; }

extern "C" int IGXCreateFromTables(const IGXTables *t, IGX *out) {
  if (!t || !out) return fail(IGX_ERR_ARG_WRONG, "null pointer");
  std::unique_ptr<_p_IGX> g(new _p_IGX());
  Space &s = g->s;
  if (t->dim < 1 || t->dim > 3 || t->dof < 1 || t->dof > MAXBC) return fail(IGX_ERR_ARG_OUTOFRANGE, "bad dim/dof");
  s.dim = t->dim; s.dof = t->dof; s.order = t->order < 1 ? 1 : (t->order > 4 ? 4 : t->order);
  s.comm_size = 1; s.comm_rank = 0;
  for (int i = 0; i < 3; ++i) {
    s.proc_sizes[i] = i < s.dim ? t->proc_sizes[i] : 1; s.proc_ranks[i] = i < s.dim ? t->proc_ranks[i] : 0;
    s.comm_size *= s.proc_sizes[i];
  }
  for (int i = s.dim - 1; i >= 0; --i) s.comm_rank = s.comm_rank * s.proc_sizes[i] + s.proc_ranks[i];
  for (int i = 0; i < s.dim; ++i) {
    const IGXAxisTables &a = t->axis[i];
    if (!a.U || !a.span || !a.offset || !a.detJac || !a.weight || !a.point || !a.value) return fail(IGX_ERR_ARG_WRONG, "null axis table");
    Axis &ax = s.axis[i]; ax.p = a.p; ax.m = a.m; ax.periodic = a.periodic; ax.nel = a.nel; ax.nnp = a.nnp;
    if (a.p > 7) return fail(IGX_ERR_SUP, "degree > 7 not supported");
    ax.U.assign(a.U, a.U + a.m + 1); ax.span.assign(a.span, a.span + a.nel);
    Basis1D &b = s.basis[i]; b.nel = a.nel; b.nqp = a.nqp; b.nen = a.nen;
    b.offset.assign(a.offset, a.offset + a.nel); b.detJac.assign(a.detJac, a.detJac + a.nel);
    b.weight.assign(a.weight, a.weight + (size_t)a.nel * a.nqp); b.point.assign(a.point, a.point + (size_t)a.nel * a.nqp);
    b.value.assign(a.value, a.value + (size_t)a.nel * a.nqp * a.nen * 5);
    s.rule_nqp[i] = a.nqp;
  }
  for (int i = s.dim; i < 3; ++i) {
    Basis1D &b = s.basis[i]; b.nel = 1; b.nqp = 1; b.nen = 1;
    b.offset.assign(1, 0); b.detJac.assign(1, 1.0); b.weight.assign(1, 1.0); b.point.assign(1, 0.0); b.value.assign(5, 0.0); b.value[0] = 1.0;
  }
  for (int i = 0; i < 3; ++i) {
    const bool in = i < s.dim;
    s.elem_sizes[i] = in ? t->elem_sizes[i] : 1; s.elem_start[i] = in ? t->elem_start[i] : 0; s.elem_width[i] = in ? t->elem_width[i] : 1;
    s.node_sizes[i] = in ? t->node_sizes[i] : 1; s.node_lstart[i] = in ? t->node_lstart[i] : 0; s.node_lwidth[i] = in ? t->node_lwidth[i] : 1;
    s.node_gstart[i] = in ? t->node_gstart[i] : 0; s.node_gwidth[i] = in ? t->node_gwidth[i] : 1;
  }
  std::string e;
  if (int rc = space_layout(s, e)) return fail(rc, e);
  s.setup = true;
  if (t->nsd) {
    if (t->nsd < s.dim || t->nsd > 3) return fail(IGX_ERR_ARG_OUTOFRANGE, "Number of space dimensions must be in range [dim,3]");
    if (!t->geometryX) return fail(IGX_ERR_ARG_WRONGSTATE, "No geometry set");
    const size_t n = (size_t)s.node_gwidth[0] * s.node_gwidth[1] * s.node_gwidth[2];
    s.geomX.assign(t->geometryX, t->geometryX + n * t->nsd); s.nsd = t->nsd;
    if (t->rational) { if (!t->rationalW) return fail(IGX_ERR_ARG_WRONGSTATE, "No geometry set"); s.geomW.assign(t->rationalW, t->rationalW + n); s.rational = 1; }
  }
  if (t->property) {      // iga->property / iga->propertyA
    if (!t->propertyA) return fail(IGX_ERR_ARG_WRONGSTATE, "No property set");
    if (t->property < 0 || t->property > 64) return fail(IGX_ERR_SUP, "device path supports at most 64 properties per node");
    const size_t n = (size_t)s.node_gwidth[0] * s.node_gwidth[1] * s.node_gwidth[2];
    s.npd = t->property; s.propA.assign(t->propertyA, t->propertyA + n * (size_t)t->property);
  }
  *out = g.release(); igx_pool_acquire();
  return 0;
}

// ------------------------------------------------------------------ device upload
static int ensure_device(IGX g) {
  Space &s = g->s;
  if (!s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  if (g->on_device) return 0;
  for (int d = 0; d < 3; ++d) {
    const Basis1D &b = s.basis[d]; const AxisLayout &L = s.lay[d];
    const int e0 = s.elem_start[d], ne = s.elem_width[d], nq = b.nqp, na = b.nen;
    std::vector<double> tab((size_t)ne * nq * na * NDER), w((size_t)ne * nq), pt((size_t)ne * nq), J(ne);
    std::vector<int> off(ne);
    for (int e = 0; e < ne; ++e) {
      J[e] = b.detJac[e0 + e]; off[e] = b.offset[e0 + e] - L.gstart;
      for (int q = 0; q < nq; ++q) {
        w[(size_t)e * nq + q] = b.weight[(size_t)(e0 + e) * nq + q]; pt[(size_t)e * nq + q] = b.point[(size_t)(e0 + e) * nq + q];
        for (int a = 0; a < na; ++a) for (int k = 0; k < NDER; ++k)
          tab[(((size_t)e * nq + q) * na + a) * NDER + k] = b.value[(((size_t)(e0 + e) * nq + q) * na + a) * 5 + k];
      }
    }
    std::vector<int64_t> prefix(L.nrow + 1, 0);
    for (int r = 0; r < L.nrow; ++r) prefix[r + 1] = prefix[r] + L.rcnt[r];
    AxisBufs &B = g->ab[d];
    std::vector<double> bnd((size_t)2 * na * NDER, 0.0);
    for (int sd = 0; sd < 2; ++sd) {
      std::vector<double> bv = b.bnd_value[sd];
      if (bv.size() < (size_t)na * 5) {   // axis beyond dim, or tables handed over without the end-of-axis rows: evaluate here
        bv.assign((size_t)na * 5, 0.0);
        const Axis &ax = s.axis[d];
        if (d < s.dim && !ax.U.empty() && !ax.span.empty()) {
          const int k = sd ? ax.span[ax.nel - 1] : ax.span[0];
          bspline_ders(k, sd ? ax.U[k + 1] : ax.U[k], ax.p, std::min(ax.p, 4), ax.U.data(), bv.data());
        } else bv[0] = 1.0;
      }
      for (int a = 0; a < na; ++a) for (int k = 0; k < NDER; ++k) bnd[((size_t)sd * na + a) * NDER + k] = bv[(size_t)a * 5 + k];
    }
    {   // the element kernels take rowmap in closed form (AxisDev::rwrap): i, or i - nnp past the wrap of a rank-local periodic axis
      const int rw = L.alias ? s.axis[d].nnp : 0x7fffffff;
      for (int i = 0; i < L.gwidth; ++i) if (L.rowmap[i] != (i < rw ? i : i - rw)) return fail(IGX_ERR_PLIB, "row map of an axis is not of the closed form the kernels assume");
    }
    if (B.bnd.upload(bnd) || B.tab.upload(tab) || B.w.upload(w) || B.J.upload(J) || B.pt.upload(pt) || B.off.upload(off) || B.rowmap.upload(L.rowmap) ||
        B.rcnt.upload(L.rcnt) || B.P.upload(L.P) || B.rcol.upload(L.rcol) || B.prefix.upload(prefix))
      return fail(IGX_ERR_MEM, "device allocation of axis tables failed");
  }
  if (g->X.upload(s.geomX) || g->W.upload(s.geomW) || g->A.upload(s.propA)) return fail(IGX_ERR_MEM, "device allocation of geometry failed");
  if (!g->errflag.p) { if (g->errflag.alloc(sizeof(int))) return fail(IGX_ERR_MEM, "device allocation failed"); HIPCK(hipMemset(g->errflag.p, 0, sizeof(int))); }
  g->nbrows = (int64_t)s.lay[0].nrow * s.lay[1].nrow * s.lay[2].nrow;
  {
    int64_t t[3];
    for (int d = 0; d < 3; ++d) { t[d] = 0; for (int r = 0; r < s.lay[d].nrow; ++r) t[d] += s.lay[d].rcnt[r]; }
    g->nblocks = t[0] * t[1] * t[2];
  }
  g->on_device = true;
  return 0;
}

static SpaceDev make_spacedev(IGX g) {
  const Space &s = g->s; SpaceDev S;
  memset(&S, 0, sizeof(S));
  S.dim = s.dim; S.dof = s.dof; S.order = s.order; S.nsd = s.nsd; S.rational = s.rational;
  for (int d = 0; d < 3; ++d) {
    AxisDev &A = S.ax[d]; const AxisBufs &B = g->ab[d]; const AxisLayout &L = s.lay[d];
    A.nel = s.elem_width[d]; A.nqp = s.basis[d].nqp; A.nen = s.basis[d].nen; A.p = L.p;
    A.estart = s.elem_start[d]; A.esizes = s.elem_sizes[d]; A.periodic = d < s.dim ? s.axis[d].periodic : 0;
    A.gwidth = L.gwidth; A.nrow = L.nrow; A.ncol = L.ncol;
    A.tab = B.tab.as<double>(); A.w = B.w.as<double>(); A.J = B.J.as<double>(); A.pt = B.pt.as<double>();
    A.bnd = B.bnd.as<double>();
    if (d < s.dim && !s.axis[d].U.empty() && !s.axis[d].span.empty()) { const Axis &ax = s.axis[d]; A.bndpt[0] = ax.U[ax.span[0]]; A.bndpt[1] = ax.U[ax.span[ax.nel - 1] + 1]; }
    {
      const Basis1D &bd = s.basis[d]; const int e0 = s.elem_start[d], ne = s.elem_width[d];
      A.off_lin = 1; A.off0 = ne > 0 ? bd.offset[e0] - L.gstart : 0;
      for (int e = 0; e < ne; ++e) if (bd.offset[e0 + e] - L.gstart != A.off0 + e) { A.off_lin = 0; break; }
    }
    A.off = B.off.as<int>(); A.rowmap = B.rowmap.as<int>(); A.rwrap = L.alias ? s.axis[d].nnp : 0x7fffffff; A.rcnt = B.rcnt.as<int>(); A.P = B.P.as<int>();
    A.prefix = B.prefix.as<int64_t>(); A.tot = 0; for (int r = 0; r < L.nrow; ++r) A.tot += L.rcnt[r];
  }
  S.X = s.nsd ? g->X.as<double>() : nullptr; S.W = s.rational ? g->W.as<double>() : nullptr;
  S.npd = s.propA.empty() ? 0 : s.npd; S.A = S.npd ? g->A.as<double>() : nullptr;
  for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) {
    auto cp = [&](const BC &h, BCDev &dv) { dv.count = 0; for (int k = 0; k < h.count && dv.count < MAXBC; ++k) if (h.field[k] < s.dof) { dv.field[dv.count] = h.field[k]; dv.value[dv.count] = h.value[k]; dv.count++; } };
    cp(s.value[a][sd], S.bcv[a][sd]); cp(s.load[a][sd], S.bcl[a][sd]);
  }
  S.fixtable = g->fixtable.as<double>();
  return S;
}

// ------------------------------------------------------------------ sparsity pattern kernels (IGACreateMat, src/petigamat.c:345-549)
struct PatDev { int nrow[3], ncol[3], W[3]; const int *rcnt[3]; const int *rcol[3]; const int64_t *prefix[3]; int64_t tot[3]; };

__global__ void k_browptr(PatDev P, int64_t nbrows, int64_t *browptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nbrows) return;
  if (r == nbrows) { browptr[r] = P.tot[0] * P.tot[1] * P.tot[2]; return; }
  const int r0 = (int)(r % P.nrow[0]), r1 = (int)((r / P.nrow[0]) % P.nrow[1]), r2 = (int)(r / ((int64_t)P.nrow[0] * P.nrow[1]));
  const int64_t c1 = P.rcnt[1][r1], c2 = P.rcnt[2][r2];
  browptr[r] = P.prefix[2][r2] * P.tot[1] * P.tot[0] + c2 * (P.prefix[1][r1] * P.tot[0] + c1 * P.prefix[0][r0]);
}
// one wavefront per row, lanes stride over the row's entries: coalesced 256-byte stores
__global__ void k_bcolidx(PatDev P, int64_t nbrows, const int64_t *browptr, int32_t *bcolidx) {
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= nbrows) return;
  const int lane = threadIdx.x & 63;
  const int r0 = (int)(r % P.nrow[0]), r1 = (int)((r / P.nrow[0]) % P.nrow[1]), r2 = (int)(r / ((int64_t)P.nrow[0] * P.nrow[1]));
  const int c0 = P.rcnt[0][r0], c1 = P.rcnt[1][r1], c2 = P.rcnt[2][r2];
  int32_t *dst = bcolidx + browptr[r];
  const int n = c0 * c1 * c2;
  for (int k = lane; k < n; k += 64) {
    const int k0 = k % c0, k1 = (k / c0) % c1, k2 = k / (c0 * c1);
    dst[k] = P.rcol[0][r0 * P.W[0] + k0] + P.ncol[0] * (P.rcol[1][r1 * P.W[1] + k1] + P.ncol[1] * P.rcol[2][r2 * P.W[2] + k2]);
  }
}

extern "C" int IGXCreateMat(IGX g, IGXMat *mat) {
  NEEDIGA(g); if (!mat) return fail(IGX_ERR_ARG_WRONG, "null pointer");
  if (int rc = ensure_device(g)) return rc;
  const Space &s = g->s;
  if ((int64_t)s.lay[0].ncol * s.lay[1].ncol * s.lay[2].ncol > INT32_MAX) return fail(IGX_ERR_SUP, "more than 2^31 column nodes per rank");
  std::unique_ptr<_p_IGXMat> A(new _p_IGXMat());
  A->iga = g; A->bs = s.dof; A->nbrows = g->nbrows; A->nblocks = g->nblocks;
  if (A->browptr.alloc((size_t)(A->nbrows + 1) * sizeof(int64_t)) || A->bcolidx.alloc((size_t)A->nblocks * sizeof(int32_t)) ||
      A->val.alloc((size_t)A->nblocks * s.dof * s.dof * sizeof(double)))
    return fail(IGX_ERR_MEM, "device allocation of the matrix failed");
  PatDev P;
  for (int d = 0; d < 3; ++d) {
    P.nrow[d] = s.lay[d].nrow; P.ncol[d] = s.lay[d].ncol; P.W[d] = 2 * s.lay[d].p + 1;
    P.rcnt[d] = g->ab[d].rcnt.as<int>(); P.rcol[d] = g->ab[d].rcol.as<int>(); P.prefix[d] = g->ab[d].prefix.as<int64_t>();
    P.tot[d] = 0; for (int r = 0; r < s.lay[d].nrow; ++r) P.tot[d] += s.lay[d].rcnt[r];
  }
  const int64_t n1 = A->nbrows + 1;
  hipLaunchKernelGGL(k_browptr, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, g->stream, P, A->nbrows, A->browptr.as<int64_t>());
  hipLaunchKernelGGL(k_bcolidx, dim3((unsigned)((A->nbrows + 3) / 4)), dim3(256), 0, g->stream, P, A->nbrows, A->browptr.as<int64_t>(), A->bcolidx.as<int32_t>());
  HIPCK(hipMemsetAsync(A->val.p, 0, A->val.bytes, g->stream));
  HIPCK(hipGetLastError());
  HIPCK(hipStreamSynchronize(g->stream));
  *mat = A.release();
  return 0;
}
extern "C" int IGXMatDestroy(IGXMat *m) { if (m && *m) { delete *m; *m = nullptr; } return 0; }
extern "C" int IGXMatGetInfo(IGXMat m, int64_t *nbrows, int64_t *nblocks, int *bs) { if (!m) return fail(IGX_ERR_ARG_WRONG, "null matrix"); if (nbrows) *nbrows = m->nbrows; if (nblocks) *nblocks = m->nblocks; if (bs) *bs = m->bs; return 0; }
extern "C" int IGXMatGetDeviceArrays(IGXMat m, const int64_t **rp, const int32_t **ci, double **v) { if (!m) return fail(IGX_ERR_ARG_WRONG, "null matrix"); if (rp) *rp = m->browptr.as<int64_t>(); if (ci) *ci = m->bcolidx.as<int32_t>(); if (v) *v = m->val.as<double>(); return 0; }
extern "C" int IGXMatCopyToHost(IGXMat m, int64_t *rp, int32_t *ci, double *v) {
  if (!m) return fail(IGX_ERR_ARG_WRONG, "null matrix");
  HIPCK(hipStreamSynchronize(m->iga->stream));
  if (rp) HIPCK(hipMemcpy(rp, m->browptr.p, m->browptr.bytes, hipMemcpyDeviceToHost));
  if (ci) HIPCK(hipMemcpy(ci, m->bcolidx.p, m->bcolidx.bytes, hipMemcpyDeviceToHost));
  if (v) HIPCK(hipMemcpy(v, m->val.p, m->val.bytes, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int IGXMatGetLayout(IGXMat m, int nrow[3], int ncol[3]) { if (!m) return fail(IGX_ERR_ARG_WRONG, "null matrix"); for (int d = 0; d < 3; ++d) { nrow[d] = m->iga->s.lay[d].nrow; ncol[d] = m->iga->s.lay[d].ncol; } return 0; }
extern "C" int IGXMatGetAxisMaps(IGXMat m, int axis, int *rownode, int *colnode) {
  if (!m || axis < 0 || axis > 2) return fail(IGX_ERR_ARG_WRONG, "bad argument");
  const AxisLayout &L = m->iga->s.lay[axis];
  if (rownode) memcpy(rownode, L.rownode.data(), L.rownode.size() * sizeof(int));
  if (colnode) memcpy(colnode, L.colnode.data(), L.colnode.size() * sizeof(int));
  return 0;
}

extern "C" int IGXCreateVec(IGX g, IGXVec *vec) {
  NEEDIGA(g); if (!vec) return fail(IGX_ERR_ARG_WRONG, "null pointer");
  if (int rc = ensure_device(g)) return rc;
  std::unique_ptr<_p_IGXVec> v(new _p_IGXVec());
  v->iga = g; v->n = g->nbrows * g->s.dof;
  if (v->a.alloc((size_t)v->n * sizeof(double))) return fail(IGX_ERR_MEM, "device allocation of the vector failed");
  HIPCK(hipMemset(v->a.p, 0, v->a.bytes));
  *vec = v.release();
  return 0;
}
extern "C" int IGXVecDestroy(IGXVec *v) { if (v && *v) { delete *v; *v = nullptr; } return 0; }
extern "C" int IGXVecGetSize(IGXVec v, int64_t *n) { if (!v) return fail(IGX_ERR_ARG_WRONG, "null vector"); *n = v->n; return 0; }
extern "C" int IGXVecGetDeviceArray(IGXVec v, double **a) { if (!v) return fail(IGX_ERR_ARG_WRONG, "null vector"); *a = v->a.as<double>(); return 0; }
extern "C" int IGXVecCopyToHost(IGXVec v, double *h) { if (!v || !h) return fail(IGX_ERR_ARG_WRONG, "null argument"); HIPCK(hipStreamSynchronize(v->iga->stream)); HIPCK(hipMemcpy(h, v->a.p, v->a.bytes, hipMemcpyDeviceToHost)); return 0; }
extern "C" int IGXVecCopyFromHost(IGXVec v, const double *h) { if (!v || !h) return fail(IGX_ERR_ARG_WRONG, "null argument"); v->iga->slab_valid = false; /* a write after the assembly's face mark (comm.hpp) */ HIPCK(hipStreamSynchronize(v->iga->stream)); HIPCK(hipMemcpy(v->a.p, h, v->a.bytes, hipMemcpyHostToDevice)); return 0; }

extern "C" int IGXSetFixTable(IGX g, IGXVec U) {   // src/petigaform.c:273-298
  NEEDIGA(g);
  if (!U) { g->fixtable.alloc(0); return 0; }
  if (U->iga != g) return fail(IGX_ERR_ARG_WRONG, "vector belongs to another IGX");
  if (g->fixtable.alloc(U->a.bytes)) return fail(IGX_ERR_MEM, "device allocation failed");
  HIPCK(hipStreamSynchronize(g->stream));
  HIPCK(hipMemcpy(g->fixtable.p, U->a.p, U->a.bytes, hipMemcpyDeviceToDevice));
  return 0;
}

// ------------------------------------------------------------------ engine controls
extern "C" int IGXSetStream(IGX g, void *stream) { NEEDIGA(g); g->stream = (hipStream_t)stream; return 0; }
extern "C" int IGXSynchronize(IGX g) {
  NEEDIGA(g);
  HIPCK(hipStreamSynchronize(g->stream));
  if (g->errflag.p) {
    int flag = 0; HIPCK(hipMemcpy(&flag, g->errflag.p, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) { HIPCK(hipMemset(g->errflag.p, 0, sizeof(int))); return fail(flag, "Non-positive det(Jacobian)"); }
  }
  return 0;
}
extern "C" int IGXSetKernel(IGX g, int which) { NEEDIGA(g); if (which < 0 || which > 4) return fail(IGX_ERR_ARG_OUTOFRANGE, "kernel choice must be 0, 1, 2, 3 or 4"); g->kernel_choice = which; return 0; }
extern "C" int IGXGetKernelName(IGX g, char *buf, int len) { NEEDIGA(g); if (!buf || len < 1) return fail(IGX_ERR_ARG_WRONG, "bad buffer"); snprintf(buf, (size_t)len, "%s%s%s%s", g->last_kernel.c_str(), g->rtc_note.empty() ? "" : " [", g->rtc_note.c_str(), g->rtc_note.empty() ? "" : "]"); return 0; }
extern "C" int IGXGetFacePasses(IGX g, int *passes) { NEEDIGA(g); if (passes) *passes = g->dom.passes; return 0; }
extern "C" int IGXSetTiming(IGX g, int flag) {
  NEEDIGA(g); g->timing = flag != 0;
  if (g->timing) for (auto &e : g->ev) if (!e) HIPCK(hipEventCreate(&e));
  return 0;
}
extern "C" int IGXGetLastTiming(IGX g, double *total_ms, double *kernel_ms, int *launches) {
  NEEDIGA(g);
  if (g->timing && g->ev[0]) {
    HIPCK(hipEventSynchronize(g->ev[3]));
    float a = 0, b = 0;
    HIPCK(hipEventElapsedTime(&a, g->ev[0], g->ev[3]));
    HIPCK(hipEventElapsedTime(&b, g->ev[1], g->ev[2]));
    g->last_total_ms = a; g->last_kernel_ms = b;
  }
  if (total_ms) *total_ms = g->last_total_ms; if (kernel_ms) *kernel_ms = g->last_kernel_ms; if (launches) *launches = g->last_launches;
  return 0;
}
extern "C" int IGXGetDominantKernelTiming(IGX g, char *name, int len, double *ms, int *launches, int64_t *elements, double *flop_per_element) {
  NEEDIGA(g);
  double t = 0;
  if (g->timing && g->ev[4] && g->dom.launches > 0) {
    HIPCK(hipEventSynchronize(g->ev[5]));
    float a = 0; HIPCK(hipEventElapsedTime(&a, g->ev[4], g->ev[5])); t = a;
  }
  if (name && len > 0) snprintf(name, (size_t)len, "%s", g->dom.name.c_str());
  if (ms) *ms = t; if (launches) *launches = g->dom.launches; if (elements) *elements = g->dom.elements;
  if (flop_per_element) *flop_per_element = g->dom.flop_per_element;
  return 0;
}
extern "C" int IGXGetDeviceInfo(char *buf, int len) {
  int dev = 0; hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) { snprintf(buf, (size_t)len, "no HIP device"); return IGX_ERR_LIB; }
  snprintf(buf, (size_t)len, "%s arch=%s CUs=%d clock=%dMHz mem=%.1fGB", pr.name, pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000, pr.totalGlobalMem / 1e9);
  return 0;
}

#endif   // !IGX_TU_DISPATCH

#ifdef IGX_TU_DISPATCH
// ------------------------------------------------------------------ feature-GEMM kernel dispatch (feature_mfma.hpp)
// one launch per group of DOFI row fields: I0 = 0, DOFI, 2*DOFI, ...
template <class Form, int DIM, int TA, int NW, int DOFI, int I0, bool HASM, bool PENCIL = false>
static void launch_feature_passes(IGX g, const SpaceDev &S, const ParamsDev &prm, const OutDev &out, const ColorRange &cr, const FCarve &cv, size_t nblocks, size_t lds_bytes, bool first, int &launches) {
  if constexpr (HASM && !PENCIL && I0 == 0 && DOFI < Form::DOF) {
    if (g->s.env.fuse_groups) {   // all groups of row fields in one launch: the element is tabulated once (feature_mfma.hpp, FUSE)
      auto fused = feature_assemble<Form, DIM, TA, NW, 0, DOFI, true, false, true>;
      if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      hipLaunchKernelGGL(fused, dim3((unsigned)nblocks), dim3(64 * NW), lds_bytes, g->stream, S, prm, out, cr, cv);
      launches++;
      return;
    }
  }
  auto kern = feature_assemble<Form, DIM, TA, NW, I0, DOFI, HASM, PENCIL>;
  if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * NW), lds_bytes, g->stream, S, prm, out, cr, cv);
  launches++;
  if constexpr (HASM && I0 + DOFI < Form::DOF) launch_feature_passes<Form, DIM, TA, NW, DOFI, I0 + DOFI, HASM, PENCIL>(g, S, prm, out, cr, cv, nblocks, lds_bytes, first, launches);
}

// executed v_mfma_f64_16x16x4 flops per element of the feature kernel's matrix phase (roofline accounting)
template <class Form, int TA>
static double feature_mfma_flop(int nq_padded) {
  constexpr int NFS = (shape_order_of<Form>::v >= 2) ? 13 : 4;
  return 2048.0 * fm_mfma_per_kstep<Form>(NFS) * TA * TA * (nq_padded / 4);
}

// Pencil mode of the feature kernel (combine before write along axis 0): 4x4x4 basis functions, one new node layer per element
// on axis 0 (a periodic axis 0 wrapped inside the rank is walked too: the last elements add to the rows the first ones wrote,
// inside the same workgroup; such an axis never has first-touch stores), no boundary-form passes, and enough pencils per
// colour of axes 1, 2 to fill the CUs (a workgroup walks the whole local axis-0 range).  IGX_COMBINE=0 / 1 overrides the count.
static bool feature_pencil_wanted(IGX g, int wgs_per_cu, bool pays) {
  const Space &s = g->s;
  if (s.dim != 3 || s.env.combine == 0) return false;
  for (int d = 0; d < 3; ++d) if (s.basis[d].nen != 4) return false;
  if (s.elem_width[0] < 4) return false;
  for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) if (s.visit[a][sd]) return false;
  for (int e = 0; e + 1 < s.elem_width[0]; ++e) if (s.basis[0].offset[s.elem_start[0] + e + 1] != s.basis[0].offset[s.elem_start[0] + e] + 1) return false;
  if (s.env.combine > 0) return true;
  if (!pays) return false;
  static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
  const int nc1 = std::max(1, s.lay[1].ncolors), nc2 = std::max(1, s.lay[2].ncolors);
  const long long per_colour = (long long)((s.elem_width[1] + nc1 - 1) / nc1) * ((s.elem_width[2] + nc2 - 1) / nc2);
  return per_colour * 2 >= (long long)ncu * wgs_per_cu;   // too few pencils: one element per workgroup fills the chip better
}

// returns 0 and sets done when the feature kernel ran; done stays false when the case is not covered
template <class Form, int DIM, int TA, int NW, int DOFI, bool HASM, bool PEN = false>
static int launch_feature_plan(IGX g, const SpaceDev &S, const OutDev &out, bool &done) {
  const Space &s = g->s;
  constexpr int DOF = Form::DOF;
  constexpr bool SECOND = Form::ORDER >= 2, SECOND_S = shape_order_of<Form>::v >= 2;
  constexpr int D2 = DIM * DIM, NFS = SECOND_S ? 1 + DIM + D2 : 1 + DIM;
  constexpr int NTA = (TA >= 8) ? TA : ((TA == 4) ? 16 / NW : 1);
  constexpr int SCALN = nscalar_of<Form>::v;
  int nq[3], na[3]; int NQ = 1, NE = 1;
  for (int d = 0; d < 3; ++d) { nq[d] = s.basis[d].nqp; na[d] = s.basis[d].nen; NQ *= nq[d]; NE *= na[d]; }
  const int NEP = 16 * TA, NQ4 = (NQ + 3) & ~3;
  const bool vec_op = (out.op == OP_SYSTEM || out.op == OP_VECTOR || out.op == OP_FUNCTION || out.op == OP_IFUNCTION);
  const unsigned need = (vec_op || SCALN > 0) ? Form::NEED : mat_need_of<Form>::v;   // as in the kernel: matrix-only drivers skip residual-only point data
  const bool fields = (need & (NEED_U | NEED_UT | NEED_GU | NEED_HU)) != 0;
  const size_t lds_limit = 160 * 1024 - 512;
  // LDS budget per workgroup.  4-wave kernels (nen <= 32) are compiled for 2-4 waves per SIMD (fm_min_waves) and sized
  // so that as many workgroups fit a CU: their latency-bound tabulation phases then overlap (CahnHilliard p=2 tangent
  // 12.8 -> 21.6 M elements/s, residual 16.7 -> 28.9).  8-wave kernels (nen = 64) hold one workgroup per CU unless the
  // form is scalar (16 accumulator VGPRs: Poisson p=3 on a NURBS geometry 8.2 vs 6.4 M/s); for the others the fewest
  // chunks of points win (NS-VMS 0.92 vs 0.84, Elasticity 2.67 vs 2.24 M/s).
  const int lds_kb_env = s.env.feature_lds_kb;   // experiment switch
  constexpr int WGS = fm_min_waves<Form, TA, NW, DOFI, HASM, PEN>();
  const size_t lds_auto = (NW == 4) ? (size_t)(160 * 1024 / WGS - 1024) : ((HASM && TA == 4 && DOF == 1) ? (size_t)78 * 1024 : lds_limit);
  const size_t lds_target = lds_kb_env > 0 ? (size_t)lds_kb_env * 1024 : lds_auto;
  FCarve cv; size_t lds_bytes = 0; bool fits = false;
  for (int pass = 0; pass < 2 && !fits; ++pass) {
    const size_t cap = pass == 0 ? lds_target : lds_limit;
    for (int nchunk = 1; nchunk <= NQ4 / 4 && !fits; ++nchunk) {
      const int QC = (((NQ4 + nchunk - 1) / nchunk) + 3) & ~3, NQP = QC * nchunk;
      int pos = 0;
      auto take = [&](int n) { int o = pos; pos += (n + 1) & ~1; return o; };
      for (int d = 0; d < 3; ++d) { cv.t1d[d] = take(nq[d] * na[d] * NDER); cv.w1d[d] = take(2 * nq[d]); }
      cv.gX = take(NE * DIM); cv.gW = take(NE); cv.Ue = take(NE * DOF); cv.Ve = take(NE * DOF);
      cv.ufix = take(NE * DOF); cv.fixval = take(NE * DOF); cv.fixflag = take(NE * DOF); cv.flux = take(NE * DOF);
      cv.JW = take(NQP); cv.xq = take(NQP * DIM); cv.E1 = take(s.nsd ? NQP * D2 : 0); cv.E2 = take((s.nsd && SECOND) ? NQP * DIM * D2 : 0);
      cv.W0 = take(s.rational ? NQP : 0); cv.W1 = take(s.rational ? NQP * DIM : 0); cv.W2 = take((s.rational && SECOND) ? NQP * D2 : 0);
      cv.G = take((Form::NEED & NEED_G) ? NQP * D2 : 0);
      cv.u = take(fields ? QC * DOF : 0); cv.ut = take(fields ? QC * DOF : 0);
      cv.gu = take((need & NEED_GU) ? QC * DOF * DIM : 0);
      cv.hu = take((need & NEED_HU) ? ((SECOND && !SECOND_S) ? NQP : QC) * DOF * D2 : 0);
      cv.hpart = 0;
      cv.lift = take(SCALN > 0 ? QC * SCALN : (out.op == OP_SYSTEM ? QC * DOF * NFS : 0));
      cv.rowbase = take(HASM ? NE : 0); cv.rowid = take(NE); cv.cc = take(NE); cv.pax = take(96); cv.adec = take(NEP / 2); cv.qdec = take((NQP + 1) / 2); cv.nrm = take(NQP * DIM);
      // the sum-factorised geometry sums of phase 1 borrow the Phi region before Phi exists
      // (and so do the sums behind the field Hessians of a form that reads them without second derivatives of N: DOF components)
      constexpr bool HU_FLY = SECOND && !SECOND_S && (Form::NEED & NEED_HU) != 0;
      cv.sfb = (NE > 64) ? 1 : DOF;      // large elements: one field at a time
      const int sf_nc = std::max((s.nsd || s.rational) ? DIM + 1 : 0, (HU_FLY && (need & NEED_HU)) ? cv.sfb : 0);
      const int sf_need = sf_nc * ((SECOND ? 3 : 2) * nq[0] * na[1] * na[2] + (SECOND ? 6 : 3) * nq[0] * nq[1] * na[2] + (SECOND ? 10 : 4) * NQ);
      // pencil mode parks the 7 leaving tiles of every (i,j) block of a launch in the Phi region before they are written out
      constexpr unsigned long long PR = mat_pair_mask_of<Form>::v;
      const int stage_need = PEN ? 7 * 16 * 17 * (PR ? fm_popcount(fm_pairs_upper(PR)) : DOFI * DOF) : 0;   // the leaving tiles (feature_mfma.hpp)
      cv.boff = 0;
      constexpr int NPS = fm_popcount((unsigned long long)(phi_mask_of<Form>::v & ((1u << NFS) - 1u)));   // features kept in LDS
      cv.phi = take(std::max(std::max(NPS * QC * NEP, sf_need), stage_need));
      cv.total = pos; cv.QC = QC; cv.nchunk = nchunk; cv.NEP = NEP;
      lds_bytes = (size_t)pos * sizeof(double);
      if (lds_bytes <= cap) fits = true;
    }
  }
  if (!fits) return 0;   // not covered: the generic kernel takes it
  ParamsDev prm; memset(&prm, 0, sizeof(prm));
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];
  int launches = 0; bool first = true;
  constexpr bool SCAL = nscalar_of<Form>::v > 0;
  // first-touch stores instead of MatZeroEntries + read (same rule as the pencil kernel): regular e mod (p+1) colours on
  // every axis; a rank with neighbours zeroes only the rows that keep neighbour-owned columns
  bool first_touch = HASM && !s.env.no_first_touch;
  for (int d = 0; d < DIM && first_touch; ++d) first_touch = axis_first_touch_ok(s, d);
  OutDev out_ft = out; out_ft.first_touch = first_touch ? 1 : 0;
  if (HASM) {
    if (!first_touch) { if (g->zero_matrix) g->zero_matrix(); }
    else if (s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] > 1) zero_neighbour_rows(s, out, g->stream);
  }     // nothing is scattered: every element in one sweep, one partial row each
  int64_t elem_base = 0;
  int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  if (SCAL) nc[0] = nc[1] = nc[2] = 1;
  if (g->timing && g->dom.ev0 && g->dom.launches == 0) (void)hipEventRecord(g->dom.ev0, g->stream);
  constexpr bool pencil = PEN;
  // Several ranks on axis 2 and a communicator: the elements within p layers of the upper face of axis 2 first (every colour),
  // a mark for the exchange (g->slab_done), then the rest -- as the pencil kernel does (gram_mfma.hpp, try_gram_mfma).
  const int n2 = s.elem_width[2], p2 = s.axis[2].p;
  bool split = g->slab_done && !SCAL && DIM == 3 && s.proc_sizes[2] > 1 && (s.proc_ranks[2] < s.proc_sizes[2] - 1 || s.axis[2].periodic) && n2 >= 2 * (p2 + 1) && axis_first_touch_ok(s, 2);
  for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) if (s.visit[a][sd]) split = false;
  const int npass = split ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int lo2 = (split && pass == 1) ? 0 : (split ? n2 - p2 : 0), hi2 = (split && pass == 1) ? n2 - p2 : n2;
  if (split) { out_ft.ft2_lo = lo2; out_ft.ft2_hi = hi2; out_ft.ft2_blocked = pass == 1 ? n2 - p2 : 0x7fffffff; }
  if (split && pass == 1) g->slab_done();
  if constexpr (PEN) {
    {
      for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) {
        const int cc[3] = {0, c1, c2};
        ColorRange cr; bool empty = false;
        cr.start[0] = 0; cr.step[0] = 1; cr.count[0] = s.elem_width[0];
        for (int d = 1; d < 3; ++d) {
          const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
          int firstel = -1, count = 0;
          for (int e = (d == 2 ? lo2 : 0); e < (d == 2 ? hi2 : nel); ++e) if (L.color[e] == cc[d]) { if (firstel < 0) firstel = e; count++; }
          if (count == 0) { empty = true; break; }
          cr.start[d] = firstel; cr.step[d] = L.p + 1; cr.count[d] = count;
        }
        if (empty) continue;
        const size_t nblocks = (size_t)cr.count[1] * cr.count[2];
        launch_feature_passes<Form, DIM, TA, NW, DOFI, 0, HASM, true>(g, S, prm, out_ft, cr, cv, nblocks, lds_bytes, first, launches);
        first = false;
      }
    }
  }
  if constexpr (!PEN) for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d], stride = L.p + 1;
      int firstel = -1, count = 0;
      for (int e = (d == 2 ? lo2 : 0); e < (d == 2 ? hi2 : nel); ++e) if (L.color[e] == cc[d]) { if (firstel < 0) firstel = e; count++; }
      if (SCAL) { firstel = 0; count = nel; }
      if (count == 0) { empty = true; break; }
      cr.start[d] = firstel; cr.step[d] = SCAL ? 1 : stride; cr.count[d] = count;
    }
    if (empty) continue;
    const size_t per2 = (size_t)cr.count[0] * cr.count[1];
    const int chunk2 = (int)std::max<size_t>(1, std::min<size_t>((size_t)cr.count[2], ((size_t)1 << 30) / std::max<size_t>(per2, 1)));
    for (int k0 = 0; k0 < cr.count[2]; k0 += chunk2) {
      ColorRange sub = cr; sub.start[2] = cr.start[2] + k0 * cr.step[2]; sub.count[2] = std::min(chunk2, cr.count[2] - k0);
      const size_t nblocks = per2 * sub.count[2];
      OutDev o2 = out_ft; o2.elem_base = elem_base; elem_base += (int64_t)nblocks;
      launch_feature_passes<Form, DIM, TA, NW, DOFI, 0, HASM>(g, S, prm, o2, sub, cv, nblocks, lds_bytes, first, launches);
      first = false;
    }
  }
  }   // passes over axis 2
  // boundary-form passes (IGAElementNextForm, src/petigaelem.c:427-447): the elements of this rank on a visited face,
  // one point layer at the face; their K_e / F_e add to what the interior pass left (stream order)
  if constexpr (!PEN) for (int bid = 0; bid < 2 * DIM; ++bid) {   // (the pencil plan is not chosen when a face is visited)
    const int ax = bid / 2, sd = bid % 2;
    if (!s.visit[ax][sd]) continue;
    const int eface = sd ? s.elem_sizes[ax] - 1 : 0;
    if (eface < s.elem_start[ax] || eface >= s.elem_start[ax] + s.elem_width[ax]) continue;   // face is on another rank
    OutDev ob = out; ob.bid = bid;
    int nc2[3] = {nc[0], nc[1], nc[2]}; nc2[ax] = 1;
    for (int c2 = 0; c2 < nc2[2]; ++c2) for (int c1 = 0; c1 < nc2[1]; ++c1) for (int c0 = 0; c0 < nc2[0]; ++c0) {
      const int cc[3] = {c0, c1, c2};
      ColorRange cr; bool empty = false;
      for (int d = 0; d < 3; ++d) {
        if (d == ax) { cr.start[d] = eface - s.elem_start[ax]; cr.step[d] = 1; cr.count[d] = 1; continue; }
        const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d], stride = L.p + 1;
        int firstel = -1, count = 0;
        for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (firstel < 0) firstel = e; count++; }
        if (SCAL) { firstel = 0; count = nel; }
        if (count == 0) { empty = true; break; }
        cr.start[d] = firstel; cr.step[d] = SCAL ? 1 : stride; cr.count[d] = count;
      }
      if (empty) continue;
      const size_t nblocks = (size_t)cr.count[0] * cr.count[1] * cr.count[2];
      ob.elem_base = elem_base; elem_base += (int64_t)nblocks;
      launch_feature_passes<Form, DIM, TA, NW, DOFI, 0, HASM>(g, S, prm, ob, cr, cv, nblocks, lds_bytes, first, launches);
      first = false;
    }
  }
  HIPCK(hipGetLastError());
  g->last_launches = launches;
  if (g->dom.launches == 0) {   // dominant kernel of this assembly, for bench.py's roofline block
    if (g->timing && g->dom.ev1) (void)hipEventRecord(g->dom.ev1, g->stream);
    g->dom.name = std::string("feature_assemble<") + (pencil ? "pencil" : "element") + ">"; g->dom.launches = launches;
    const bool fused = HASM && !pencil && DOFI < DOF && s.env.fuse_groups;   // one launch forms all groups of row fields of its elements
    g->dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2] * ((HASM && !fused) ? (DOF / DOFI) : 1);
    g->dom.flop_per_element = HASM ? feature_mfma_flop<Form, TA>(cv.QC * cv.nchunk) * (fused ? DOF : DOFI) / DOF : 0.0;
  }
  if (HASM) g->last_kernel = std::string("feature_assemble(mfma_f64_16x16x4,tiles=") + std::to_string(TA) + "x" + std::to_string(TA) + ",waves=" + char('0' + NW) +
                   ",rowfields/launch=" + char('0' + DOFI) + ((!pencil && DOFI < DOF && s.env.fuse_groups) ? std::string("x") + char('0' + DOF / DOFI) + " fused" : std::string()) +
                   ",chunks=" + std::to_string(cv.nchunk) + (pencil ? ",pencil walk axis 0" : "") + ")";
  else g->last_kernel = std::string("feature_assemble(vector only,waves=") + char('0' + NW) + ",chunks=" + std::to_string(cv.nchunk) + ")";
  done = true;
  return 0;
}

template <class Form, int DIM, int TA>
static int launch_feature_ta(IGX g, const SpaceDev &S, const OutDev &out, bool &done) {
  constexpr int DOF = Form::DOF;
  constexpr int NW = (TA >= 4) ? 8 : 4;
  // 4x4 tiles: 8 waves, <= 144 accumulator VGPRs per wave (dof 4: two launches of two row fields).  Measured on
  // Elasticity3D p=3: one launch of all row fields with 8 waves 2.67 M elements/s (3.2 with Gram accumulators); three
  // launches of one row field with 4-wave workgroups, two per CU, 1.63 M elements/s (tabulation repeated per launch).
  if constexpr (nscalar_of<Form>::v > 0) return launch_feature_plan<Form, DIM, TA, NW, DOF, false>(g, S, out, done);
  else {
    const bool hasM = (out.op == OP_SYSTEM || out.op == OP_MATRIX || out.op == OP_JACOBIAN || out.op == OP_IJACOBIAN);
    if (!hasM) return launch_feature_plan<Form, DIM, TA, NW, DOF, false>(g, S, out, done);
    // field Hessians on the fly live in the vector-only kernels unless the matrix callback reads them too (feature_mfma.hpp,
    // HU_HERE): IGAComputeSystem of such a form (NS-VMS: nobody calls it) is left to the point-form kernel
    constexpr bool HU_SPLIT = Form::ORDER >= 2 && shape_order_of<Form>::v < 2 && (Form::NEED & NEED_HU) != 0 && (mat_need_of<Form>::v & NEED_HU) == 0;
    if (HU_SPLIT && out.op == OP_SYSTEM) return 0;
    constexpr bool GRAM = mat_pair_mask_of<Form>::v != 0ull;   // all row fields from one set of Gram accumulators
    // a scalar form at nen = 64 has 4 tiles per wave with 4-wave workgroups: small enough for 4 workgroups per CU
    // (Poisson p=3 on a NURBS geometry: 11.9 vs 10.4 M elements/s with the 8-wave layout)
    // 8x8 tiles (nen <= 128): a wave holds 8 tiles per accumulator set, so one row field per group (the groups are fused)
    constexpr int DOFI = (TA >= 8 && !GRAM) ? 1 : ((TA == 4 && DOF == 4 && !GRAM) ? 2 : DOF);
    if constexpr (TA == 4 && DIM == 3) {
      // pencil mode: the same wave layouts as the element mode (scalar forms: 4 waves with 4 tiles each, 4 workgroups per CU)
      constexpr int NWP = (DOF == 1) ? 4 : 8;
      constexpr int WG = fm_min_waves<Form, 4, NWP, (DOF == 1 ? 1 : DOFI), true, true>();
      // Measured (MI355X): Elasticity3D 128^3 3.35 -> 4.8 M elements/s; NavierStokesVMS 96^3 (two launches of 8 accumulator
      // sets, 128 registers live through a heavy tabulation) 1.41 -> 0.78: the automatic choice keeps the walk for the
      // constant-coefficient (Gram) multi-field forms.  Scalar forms lose as well (Poisson p=3 on a NURBS geometry, 128^3:
      // 8.5 vs 12.5 M elements/s: their scatter is small next to the tabulation, and the walk runs fewer workgroups per CU).
      constexpr bool PAYS = GRAM && DOF > 1;
      if (feature_pencil_wanted(g, WG, PAYS)) {   // (not covered when the staging buffers do not fit the LDS next to a mapped geometry's arrays)
        if (int rc = launch_feature_plan<Form, DIM, 4, NWP, (DOF == 1 ? 1 : DOFI), true, true>(g, S, out, done)) return rc;
        if (done) return 0;
      }
    }
    if constexpr (TA == 4 && DOF == 1) return launch_feature_plan<Form, DIM, 4, 4, 1, true>(g, S, out, done);
    else
    return launch_feature_plan<Form, DIM, TA, NW, DOFI, true>(g, S, out, done);
  }
}

template <class Form, int DIM>
static int launch_feature(IGX g, const SpaceDev &S, const OutDev &out, bool &done) {
  done = false;
  if constexpr (DIM < 2) return 0;
  else {
    const Space &s = g->s;
    constexpr bool fields = (Form::NEED & (NEED_U | NEED_UT | NEED_GU | NEED_HU)) != 0;
    if (s.dof != Form::DOF && (nscalar_of<Form>::v == 0 || fields)) return 0;
    int NE = 1; for (int d = 0; d < 3; ++d) NE *= s.basis[d].nen;
    if (NE > 256) return 0;
    if (NE > 128) {  // 16 tile rows x two panels of 8 tile columns (p = 5 in 3-D: nen = 216): one accumulator set of 16 tiles per wave
      constexpr unsigned long long PR = mat_pair_mask_of<Form>::v;
      constexpr int NACC16 = PR ? fm_popcount(PR) : Form::DOF;
      if constexpr (DIM == 3 && nscalar_of<Form>::v == 0 && !has_boundary_of<Form>::v) {
        for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) if (s.visit[a][sd]) return 0;
        if constexpr (NACC16 == 1) return launch_feature_ta<Form, DIM, 16>(g, S, out, done);
        else {
          const bool hasM = (out.op == OP_SYSTEM || out.op == OP_MATRIX || out.op == OP_JACOBIAN || out.op == OP_IJACOBIAN);
          if (hasM) return 0;
          return launch_feature_plan<Form, DIM, 16, 8, Form::DOF, false>(g, S, out, done);
        }
      } else return 0;
    }
    if (NE > 64) {   // 8x8 tiles (p = 4 in 3-D: nen = 125) while the accumulators fit: <= 2 sets of 8 tiles per wave
      constexpr unsigned long long PR = mat_pair_mask_of<Form>::v;
      constexpr int NACC8 = PR ? fm_popcount(PR) : Form::DOF;      // Gram pairs, or the dof blocks of one row field
      if constexpr (DIM == 3 && nscalar_of<Form>::v == 0 && !has_boundary_of<Form>::v) {
        for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) if (s.visit[a][sd]) return 0;
        if constexpr (NACC8 <= 2) return launch_feature_ta<Form, DIM, 8>(g, S, out, done);
        else {   // more accumulator sets than a wave holds at 8 tiles each: the vector-only drivers still take the kernel
          const bool hasM = (out.op == OP_SYSTEM || out.op == OP_MATRIX || out.op == OP_JACOBIAN || out.op == OP_IJACOBIAN);
          if (hasM) return 0;
          return launch_feature_plan<Form, DIM, 8, 8, Form::DOF, false>(g, S, out, done);
        }
      } else return 0;
    }
    if (NE <= 16) return launch_feature_ta<Form, DIM, 1>(g, S, out, done);
    if (NE <= 32) return launch_feature_ta<Form, DIM, 2>(g, S, out, done);
    return launch_feature_ta<Form, DIM, 4>(g, S, out, done);
  }
}

// ------------------------------------------------------------------ generic kernel dispatch
template <class Form, int DIM>
static int launch_generic(IGX g, const SpaceDev &S, const OutDev &out) {
  const Space &s = g->s;
  constexpr int DOF = Form::DOF;
  constexpr bool SECOND = Form::ORDER >= 2, THIRD = Form::ORDER >= 3;
  constexpr int NF = nfeat<DIM>(Form::ORDER), D2 = DIM * DIM, D3 = D2 * DIM;
  constexpr int NS = nscalar_of<Form>::v;
  // third-order tabulation, the property array and the point's shape table exist in the general kernel only
  constexpr bool GENERAL = general_only_of<Form>::v;
  if (GENERAL && g->kernel_choice != 0 && g->kernel_choice != 1) return fail(IGX_ERR_SUP, "a form of order 3 or one that reads the property array runs on the general kernel only");
  if ((Form::NEED & NEED_PROP) && !S.npd) return fail(IGX_ERR_ARG_WRONGSTATE, "No property set");      // src/petigaelem.c:300
  // a geometry of another dimension than the parametric one: tabulated by the general kernel only (no inverse map: src/petigaelem.c:966)
  const bool emb = s.nsd && s.nsd != DIM;
  if (emb && g->kernel_choice != 0 && g->kernel_choice != 1) return fail(IGX_ERR_SUP, "a geometry with nsd != dim runs on the general kernel only");
  if ((Form::NEED & NEED_MAPX) && !s.nsd) return fail(IGX_ERR_ARG_WRONGSTATE, "No geometry set");
  if (THIRD && s.order < 3) return fail(IGX_ERR_ARG_WRONGSTATE, "the form reads third derivatives (p->shape[3]): call IGASetOrder(iga,3) first");
#ifdef IGX_HAVE_VEC_SUMFACT
  if constexpr (DIM == 3 && !GENERAL) if (g->kernel_choice == 0 && !emb) {   // vector-only drivers: sum factorisation both ways (vec_sumfact.hpp)
    bool done = false;
    ParamsDev prm; memset(&prm, 0, sizeof(prm));
    for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];
    if (int rc = try_vec_sumfact<Form>(s, S, prm, out, g->stream, g->last_kernel, g->last_launches, done)) return fail(rc, "vec_sumfact kernel launch failed");
    if (done) return 0;
  }
#endif
#ifdef IGX_HAVE_BLOCK_PENCIL
  if constexpr (DIM == 3 && !GENERAL) if ((g->kernel_choice == 0 || g->kernel_choice == 4) && !emb) {   // band rows by node layer (block_pencil.hpp)
    bool done = false;
    ParamsDev prm; memset(&prm, 0, sizeof(prm));
    for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];
    if (int rc = try_block_pencil<Form>(s, S, prm, out, g->stream, g->last_kernel, g->last_launches, g_err, done, g->dom, g->zero_matrix, g->slab_done)) return rc;
    if (done) return 0;
  }
#endif
#ifdef IGX_HAVE_BAND_PT
  // (p = 2 only on request: the 27-function element in 4 x 4 x 4 tile slots wastes two thirds of the MFMAs there -- NS-VMS 48^3: 7.7 M el/s, the
  //  feature kernel's 2 x 2 tiles are as fast -- so the automatic choice keeps the feature kernel at p = 2)
  if constexpr (DIM == 3 && !GENERAL) if ((g->kernel_choice == 0 && s.axis[0].p == 3) || g->kernel_choice == 4) {   // band rows by node layer, point-dependent coefficients (band_pt.hpp)
    bool done = false;
    ParamsDev prm; memset(&prm, 0, sizeof(prm));
    for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];
    if (int rc = try_band_pt<Form>(s, S, prm, out, g->stream, g->last_kernel, g->last_launches, g_err, done, g->dom, g->zero_matrix, g->slab_done)) return rc;
    if (done) return 0;
  }
#endif
  if (g->kernel_choice == 4) return fail(IGX_ERR_SUP, "the band-row kernels do not cover this case (block_pencil: 3-D, p = 3, identity geometry, System / Matrix driver of a constant-coefficient form with 2 or 3 fields; band_pt: 3-D, p = 2 or 3, any geometry: matrix-only driver of a 4-field form with separated point coefficients, System / Matrix driver of a constant-coefficient form with 2 or 3 fields without boundary loads)");
  if constexpr (!GENERAL) if (g->kernel_choice != 1 && !emb) {   // matrix-producing ops: the dense contraction goes to the matrix cores when covered
    bool done = false;
    if (int rc = launch_feature<Form, DIM>(g, S, out, done)) return rc;
    if (done) return 0;
    if (g->kernel_choice == 3) return fail(IGX_ERR_SUP, "the feature-GEMM kernel does not cover this case (needs dim >= 2 and nen <= 64; in 3-D nen <= 128 / 256 for forms with at most two / one accumulator set)");
  }
  if (g->zero_matrix) g->zero_matrix();
  const bool fields = (Form::NEED & (NEED_U | NEED_UT | NEED_GU | NEED_HU | NEED_D3U)) != 0;
  if (s.dof != DOF && (NS == 0 || fields)) return fail(IGX_ERR_ARG_WRONG, "form does not match the number of fields (dof)");
  int nq[3], na[3]; int NQ = 1, NE = 1;
  for (int d = 0; d < 3; ++d) { nq[d] = s.basis[d].nqp; na[d] = s.basis[d].nen; NQ *= nq[d]; NE *= na[d]; }
  Carve cv; int pos = 0;
  auto take = [&](int n) { int o = pos; pos += (n + 1) & ~1; return o; };   // keep 16-byte alignment
  for (int d = 0; d < 3; ++d) { cv.t1d[d] = take(nq[d] * na[d] * NDER); cv.w1d[d] = take(nq[d]); }
  const int nsd = s.nsd ? s.nsd : DIM;
  cv.gX = take(NE * nsd); cv.gW = take(NE); cv.Ue = take(NE * DOF); cv.Ve = take(NE * DOF);
  cv.ufix = take(NE * DOF); cv.fixval = take(NE * DOF); cv.fixflag = take(NE * DOF); cv.flux = take(NE * DOF);
  cv.JW = take(NQ); cv.xq = take(NQ * nsd); cv.E1 = take((s.nsd && !emb) ? NQ * D2 : 0); cv.E2 = take((s.nsd && !emb && SECOND) ? NQ * DIM * D2 : 0);
  cv.W0 = take(s.rational ? NQ : 0); cv.W1 = take(s.rational ? NQ * DIM : 0); cv.W2 = take((s.rational && SECOND) ? NQ * D2 : 0);
  cv.G = take((Form::NEED & NEED_G) ? NQ * DIM * nsd : 0);
  cv.X1m = take((Form::NEED & NEED_MAPX) ? NQ * nsd * DIM : 0); cv.X2m = take(((Form::NEED & NEED_MAPX) && SECOND) ? NQ * nsd * D2 : 0);
  cv.E3 = take((s.nsd && !emb && THIRD) ? NQ * DIM * D3 : 0); cv.W3 = take((s.rational && THIRD) ? NQ * D3 : 0);
  cv.d3u = take((THIRD && (Form::NEED & NEED_D3U)) ? NQ * DOF * D3 : 0); cv.gA = take(NE * S.npd);
  cv.u = take(fields ? NQ * DOF : 0); cv.ut = take(fields ? NQ * DOF : 0);
  cv.gu = take((Form::NEED & NEED_GU) ? NQ * DOF * DIM : 0); cv.hu = take((Form::NEED & NEED_HU) ? NQ * DOF * D2 : 0);
  cv.lift = take(NS > 0 ? NQ * NS : (out.op == OP_SYSTEM ? NQ * DOF * NF : 0));
  cv.nrm = take(NQ * nsd);
  const size_t phi_doubles = (size_t)NQ * NE * NF;
  // Phi in LDS up to 64 KiB per workgroup, in an HBM slice beyond: a 131 KiB Phi (p = 3) in LDS leaves one workgroup per CU and
  // measured slower than the HBM slice with two (Poisson p=3 48^3: 0.91 vs 1.71 M elements/s, Elasticity 0.80 vs 1.28)
  const size_t lds_limit = 160 * 1024 - 512, phi_limit = 64 * 1024;
  bool phi_in_lds = ((size_t)pos + phi_doubles) * sizeof(double) <= phi_limit;
  if (phi_in_lds) cv.phi = take((int)phi_doubles); else cv.phi = -1;
  cv.total = pos;
  const size_t lds_bytes = (size_t)pos * sizeof(double);
  if (lds_bytes > lds_limit) return fail(IGX_ERR_SUP, "element work set exceeds the 160 KiB LDS");
  auto kern = generic_assemble<Form, DIM>;
  HIPCK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  ParamsDev prm; memset(&prm, 0, sizeof(prm));
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];

  const size_t scratch_cap = (size_t)2 << 30;   // HBM slice for Phi when it does not fit in LDS
  size_t max_blocks = phi_in_lds ? ((size_t)1 << 30) : scratch_cap / (phi_doubles * sizeof(double));
  if (max_blocks < 1) max_blocks = 1;
  int launches = 0;
  int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  if (NS > 0) nc[0] = nc[1] = nc[2] = 1;   // nothing is scattered: every element in one sweep
  int64_t elem_base = 0;
  for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d], stride = L.p + 1;
      int first = -1, count = 0;
      for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
      if (NS > 0) { first = 0; count = nel; }
      if (count == 0) { empty = true; break; }
      cr.start[d] = first; cr.step[d] = (NS > 0) ? 1 : stride; cr.count[d] = count;
    }
    if (empty) continue;
    // split along axis 2 so that a launch never needs more scratch than the cap
    const size_t per2 = (size_t)cr.count[0] * cr.count[1];
    int chunk2 = (int)std::max<size_t>(1, std::min<size_t>((size_t)cr.count[2], max_blocks / std::max<size_t>(per2, 1)));
    if (!phi_in_lds && per2 > max_blocks) return fail(IGX_ERR_SUP, "scratch too small for one element layer");
    for (int k0 = 0; k0 < cr.count[2]; k0 += chunk2) {
      ColorRange sub = cr; sub.start[2] = cr.start[2] + k0 * cr.step[2]; sub.count[2] = std::min(chunk2, cr.count[2] - k0);
      const size_t nblocks = per2 * sub.count[2];
      if (!phi_in_lds) {
        const size_t need = nblocks * phi_doubles * sizeof(double);
        if (g->scratch.bytes < need) { HIPCK(hipStreamSynchronize(g->stream)); if (g->scratch.alloc(need)) return fail(IGX_ERR_MEM, "scratch allocation failed"); }
      }
      OutDev o2 = out; o2.elem_base = elem_base; elem_base += (int64_t)nblocks;
      hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), lds_bytes, g->stream, S, prm, o2, sub, cv, g->scratch.as<double>(), phi_doubles);
      launches++;
    }
  }
  // boundary-form passes (IGAElementNextForm, src/petigaelem.c:427-447): the elements of this rank on a visited face, one point
  // layer at the face; their K_e / F_e add to what the interior pass left (stream order)
  for (int bid = 0; bid < 2 * DIM; ++bid) {
    const int ax = bid / 2, sd = bid % 2;
    if (!s.visit[ax][sd]) continue;
    const int eface = sd ? s.elem_sizes[ax] - 1 : 0;
    if (eface < s.elem_start[ax] || eface >= s.elem_start[ax] + s.elem_width[ax]) continue;   // the face is on another rank
    OutDev ob = out; ob.bid = bid;
    int nc2[3] = {nc[0], nc[1], nc[2]}; nc2[ax] = 1;
    for (int c2 = 0; c2 < nc2[2]; ++c2) for (int c1 = 0; c1 < nc2[1]; ++c1) for (int c0 = 0; c0 < nc2[0]; ++c0) {
      const int cc[3] = {c0, c1, c2};
      ColorRange cr; bool empty = false;
      for (int d = 0; d < 3; ++d) {
        if (d == ax) { cr.start[d] = eface - s.elem_start[ax]; cr.step[d] = 1; cr.count[d] = 1; continue; }
        const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d], stride = L.p + 1;
        int first = -1, count = 0;
        for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
        if (NS > 0) { first = 0; count = nel; }
        if (count == 0) { empty = true; break; }
        cr.start[d] = first; cr.step[d] = (NS > 0) ? 1 : stride; cr.count[d] = count;
      }
      if (empty) continue;
      const size_t nblocks = (size_t)cr.count[0] * cr.count[1] * cr.count[2];
      if (!phi_in_lds) {
        const size_t need = nblocks * phi_doubles * sizeof(double);
        if (nblocks > max_blocks) return fail(IGX_ERR_SUP, "scratch too small for one face layer");
        if (g->scratch.bytes < need) { HIPCK(hipStreamSynchronize(g->stream)); if (g->scratch.alloc(need)) return fail(IGX_ERR_MEM, "scratch allocation failed"); }
      }
      ob.elem_base = elem_base; elem_base += (int64_t)nblocks;
      hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), lds_bytes, g->stream, S, prm, ob, cr, cv, g->scratch.as<double>(), phi_doubles);
      launches++;
    }
  }
  HIPCK(hipGetLastError());
  g->last_launches = launches;
  g->last_kernel = std::string("generic_assemble(") + (phi_in_lds ? "phi=LDS" : "phi=HBM") + ")";
  return 0;
}

// GROUP < 0: every form; else only the forms of that group (0 scalar second-order-free forms, 1 multi-field
// constant-coefficient forms, 2 the nonlinear demos), IGX_NOT_MINE for the others
template <int DIM, int GROUP>
static int dispatch_dim(IGX g, const SpaceDev &S, const OutDev &out) {
  const Space &s = g->s;
#define IGX_GROUP(gid, call) do { if constexpr (GROUP < 0 || GROUP == (gid)) return call; else return IGX_NOT_MINE; } while (0)
  switch (s.form) {
  case IGX_FORM_POISSON:   IGX_GROUP(0, (launch_generic<FormPoisson<DIM>, DIM>(g, S, out)));
  case IGX_FORM_POISSON_F: IGX_GROUP(0, (launch_generic<FormPoissonF<DIM>, DIM>(g, S, out)));
  case IGX_FORM_L2PROJ_X2: IGX_GROUP(0, (launch_generic<FormL2ProjX2<DIM>, DIM>(g, S, out)));
  case IGX_FORM_BOUNDARYINTEGRAL: IGX_GROUP(0, (launch_generic<FormBoundaryIntegral<DIM>, DIM>(g, S, out)));
  case IGX_FORM_NITSCHE:          IGX_GROUP(0, (launch_generic<FormNitsche<DIM>, DIM>(g, S, out)));
  case IGX_FORM_DER3:      IGX_GROUP(0, (launch_generic<FormDer3<DIM>, DIM>(g, S, out)));
  case IGX_FORM_PROPERTY:  IGX_GROUP(0, (launch_generic<FormProperty<DIM>, DIM>(g, S, out)));
  case IGX_FORM_SURFACE:
    if constexpr (GROUP < 0 || GROUP == 0) {
      if constexpr (DIM <= 2) return launch_generic<FormSurface<DIM>, DIM>(g, S, out);
      else return fail(IGX_ERR_ARG_WRONG, "the surface form needs dim = 1 or 2 (a curve or a surface in space)");
    } else return IGX_NOT_MINE;
  case IGX_FORM_ERRNORM:   IGX_GROUP(1, (launch_generic<FormErrNorm<DIM>, DIM>(g, S, out)));
  case IGX_FORM_MASS:
    if constexpr (GROUP < 0 || GROUP == 1) {
      switch (s.dof) {
      case 1: return launch_generic<FormMass<DIM, 1>, DIM>(g, S, out);
      case 2: return launch_generic<FormMass<DIM, 2>, DIM>(g, S, out);
      case 3: return launch_generic<FormMass<DIM, 3>, DIM>(g, S, out);
      case 4: return launch_generic<FormMass<DIM, 4>, DIM>(g, S, out);
      default: return fail(IGX_ERR_SUP, "mass form is instantiated for dof 1..4");
      }
    } else return IGX_NOT_MINE;
  case IGX_FORM_ELASTICITY:
    if constexpr (GROUP < 0 || GROUP == 1) {
      if constexpr (DIM == 3) return launch_generic<FormElasticity, 3>(g, S, out);
      else return fail(IGX_ERR_ARG_WRONG, "Elasticity3D form needs dim = 3");
    } else return IGX_NOT_MINE;
  case IGX_FORM_ELASTICITY_F:
    if constexpr (GROUP < 0 || GROUP == 1) {
      if constexpr (DIM == 3) return launch_generic<FormElasticityF, 3>(g, S, out);
      else return fail(IGX_ERR_ARG_WRONG, "Elasticity3D form needs dim = 3");
    } else return IGX_NOT_MINE;
  case IGX_FORM_CAHNHILLIARD:
    if constexpr (GROUP < 0 || GROUP == 2) {
      if constexpr (DIM >= 2) return launch_generic<FormCahnHilliard<DIM>, DIM>(g, S, out);
      else return fail(IGX_ERR_ARG_WRONG, "Cahn-Hilliard form needs dim = 2 or 3");
    } else return IGX_NOT_MINE;
  case IGX_FORM_BRATU:     IGX_GROUP(2, (launch_generic<FormBratu<DIM>, DIM>(g, S, out)));
  case IGX_FORM_NSVMS:
    if constexpr (GROUP < 0 || GROUP == 2) {
      if constexpr (DIM == 3) return launch_generic<FormNSVMS, 3>(g, S, out);
      else return fail(IGX_ERR_ARG_WRONG, "NavierStokesVMS form needs dim = 3");
    } else return IGX_NOT_MINE;
  default: return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetForm...() first");   // IGACheckFormOp, include/petiga.h:925-936
  }
#undef IGX_GROUP
}

// IGAComputeScalar (src/petigacomp.c:35-98): the functionals of one dimension
template <int DIM>
static int dispatch_scalar(IGX g, int kind, const SpaceDev &S, const OutDev &out, int order) {
  switch (kind) {
  case IGX_SCALAR_VOLUME:  return launch_generic<ScalarVolume<DIM>, DIM>(g, S, out);
  case IGX_SCALAR_X2ERR:   return launch_generic<ScalarX2Err<DIM>, DIM>(g, S, out);
  case IGX_SCALAR_ERRNORM: return order >= 2 ? launch_generic<ScalarErrNorm<DIM, true>, DIM>(g, S, out) : launch_generic<ScalarErrNorm<DIM, false>, DIM>(g, S, out);
  default: return fail(IGX_ERR_ARG_OUTOFRANGE, "unknown scalar functional");
  }
}

// the entry points of this unit
int igx_tu_dispatch(std::integral_constant<int, IGX_TU_DIM>, std::integral_constant<int, IGX_TU_GROUP>, IGX g, const SpaceDev &S, const OutDev &out) {
  return dispatch_dim<IGX_TU_DIM, IGX_TU_GROUP>(g, S, out);
}
#if IGX_TU_GROUP < 0 || IGX_TU_GROUP == 2
int igx_tu_scalar(std::integral_constant<int, IGX_TU_DIM>, IGX g, int kind, const SpaceDev &S, const OutDev &out, int order) {
  return dispatch_scalar<IGX_TU_DIM>(g, kind, S, out, order);
}
#endif
#endif   // IGX_TU_DISPATCH

#ifndef IGX_TU_DISPATCH
// kernel instantiations live in the dispatch units (one per dimension; three form groups for dim 3)
int igx_tu_dispatch(std::integral_constant<int, 1>, std::integral_constant<int, -1>, IGX g, const SpaceDev &S, const OutDev &out);
int igx_tu_dispatch(std::integral_constant<int, 2>, std::integral_constant<int, -1>, IGX g, const SpaceDev &S, const OutDev &out);
int igx_tu_dispatch(std::integral_constant<int, 3>, std::integral_constant<int, 0>, IGX g, const SpaceDev &S, const OutDev &out);
int igx_tu_dispatch(std::integral_constant<int, 3>, std::integral_constant<int, 1>, IGX g, const SpaceDev &S, const OutDev &out);
int igx_tu_dispatch(std::integral_constant<int, 3>, std::integral_constant<int, 2>, IGX g, const SpaceDev &S, const OutDev &out);
int igx_tu_scalar(std::integral_constant<int, 1>, IGX g, int kind, const SpaceDev &S, const OutDev &out, int order);
int igx_tu_scalar(std::integral_constant<int, 2>, IGX g, int kind, const SpaceDev &S, const OutDev &out, int order);
int igx_tu_scalar(std::integral_constant<int, 3>, IGX g, int kind, const SpaceDev &S, const OutDev &out, int order);
static int launch_generic_rtc(IGX g, const SpaceDev &S, const OutDev &out);
static int dispatch_by_dim(IGX g, const SpaceDev &S, const OutDev &out) {
  using std::integral_constant;
  switch (g->s.dim) {
  case 1: return igx_tu_dispatch(integral_constant<int, 1>(), integral_constant<int, -1>(), g, S, out);
  case 2: return igx_tu_dispatch(integral_constant<int, 2>(), integral_constant<int, -1>(), g, S, out);
  default: {
    int rc = igx_tu_dispatch(integral_constant<int, 3>(), integral_constant<int, 0>(), g, S, out);
    if (rc == IGX_NOT_MINE) rc = igx_tu_dispatch(integral_constant<int, 3>(), integral_constant<int, 1>(), g, S, out);
    if (rc == IGX_NOT_MINE) rc = igx_tu_dispatch(integral_constant<int, 3>(), integral_constant<int, 2>(), g, S, out);
    return rc;
  }
  }
}

// ------------------------------------------------------------------ the drivers
// fused (IGXComputeFunctionJacobian / IGXComputeIFunctionIJacobian): op is the Jacobian's, b the Function's vector; one pass of the
// walk forms both (state_pencil_kr).  *fused_done = false and nothing written when no fused kernel covers the case: the caller
// then runs the two drivers one after the other.
static int compute(IGX g, int op, IGXMat A, IGXVec b, IGXVec U, IGXVec V, double shift, double t, bool *fused_done = nullptr) {
  NEEDIGA(g);
  if (int rc = ensure_device(g)) return rc;
  const Space &s = g->s;
  if (s.form == IGX_FORM_NONE) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetForm...() first");
  const bool fused = fused_done != nullptr;
  if (fused) {
    *fused_done = false;
    bool ok = s.dim == 3 && s.nsd == 0 && s.axis[0].p == 2 && s.env.state_pencil && s.env.p2_pack != 0 && s.env.fuse_resid != 0 && (g->kernel_choice == 0 || g->kernel_choice == 2) &&
              (s.form == IGX_FORM_CAHNHILLIARD || s.form == IGX_FORM_BRATU);
    for (int d = 0; d < 3 && ok; ++d) for (int sd = 0; sd < 2; ++sd) if (s.load[d][sd].count || s.visit[d][sd]) ok = false;      // (loads: F_e[k] -= flux, src/petigaelem.c:1449-1454 -- the element kernels' business)
    if (!ok) return 0;
  }
  const bool hasM = (op == OP_SYSTEM || op == OP_MATRIX || op == OP_JACOBIAN || op == OP_IJACOBIAN);
  const bool hasV = fused || (op == OP_SYSTEM || op == OP_VECTOR || op == OP_FUNCTION || op == OP_IFUNCTION);
  if (hasM && (!A || A->iga != g)) return fail(IGX_ERR_ARG_WRONG, "matrix missing or created by another IGX");
  if (hasV && (!b || b->iga != g)) return fail(IGX_ERR_ARG_WRONG, "vector missing or created by another IGX");
  if (U && U->iga != g) return fail(IGX_ERR_ARG_WRONG, "state vector created by another IGX");
  if (V && V->iga != g) return fail(IGX_ERR_ARG_WRONG, "state vector created by another IGX");
  OutDev out; memset(&out, 0, sizeof(out));
  out.op = op; out.shift = shift; out.t = t; out.errflag = g->errflag.as<int>(); out.bid = -1;
  out.debug = s.env.debug_feature;
  if (kDebug && (out.debug & 8)) { if (!g->dbgbuf.p) g->dbgbuf.alloc(32 * sizeof(long long)); HIPCK(hipMemsetAsync(g->dbgbuf.p, 0, 32 * sizeof(long long), g->stream)); out.dbg = g->dbgbuf.as<long long>(); }
  if (s.env.clock_probe) { if (!g->clkbuf.p) { if (g->clkbuf.alloc(4 * sizeof(long long))) return fail(IGX_ERR_MEM, "clock probe buffer"); HIPCK(hipMemsetAsync(g->clkbuf.p, 0, 4 * sizeof(long long), g->stream)); } out.clk = g->clkbuf.as<long long>(); }
  if (hasM) { out.browptr = A->browptr.as<int64_t>(); out.val = A->val.as<double>(); }
  if (hasV) out.vec = b->a.as<double>();
  out.U = U ? U->a.as<double>() : nullptr; out.V = V ? V->a.as<double>() : nullptr;
  if (g->timing) HIPCK(hipEventRecord(g->ev[0], g->stream));
  // MatZeroEntries / VecZeroEntries (src/petigaksp.c:166-167)
  // (the matrix is zeroed lazily: the axis-0 pencil walk stores first touches and needs no MatZeroEntries at all)
  bool zeroed = !hasM;
  auto zero_matrix = [&]() { if (!zeroed) { (void)hipMemsetAsync(A->val.p, 0, A->val.bytes, g->stream); zeroed = true; } };
  if (hasV) HIPCK(hipMemsetAsync(b->a.p, 0, b->a.bytes, g->stream));
  if (g->timing) HIPCK(hipEventRecord(g->ev[1], g->stream));
  const SpaceDev S = make_spacedev(g);
  int rc;
  bool done = false;
  g->dom = DomInfo(); g->dom.ev0 = g->timing ? g->ev[4] : nullptr; g->dom.ev1 = g->timing ? g->ev[5] : nullptr;
  if (g->kernel_choice != 1 && g->kernel_choice != 3 && s.form != IGX_FORM_SOURCE) {   // (a run-time form reaches the pencil walk through rtc.hpp)
    std::function<void()> slab_done;
    std::function<void(int)> face_done;
    g->slab_valid = 0;
    if (g->comm && s.env.overlap) {
      slab_done = [&]() {
        if (!g->slab_ev && hipEventCreateWithFlags(&g->slab_ev, hipEventDisableTiming) != hipSuccess) return;
        if (hipEventRecord(g->slab_ev, g->stream) == hipSuccess) { g->slab_valid |= 1; g->slab_A = A; g->slab_b = b; }
      };
      if (s.env.overlap != 2) face_done = [&](int axis) {      // (IGX_OVERLAP=2: the mark of axis 2 alone, as in round 3)
        if (axis < 0 || axis > 1) return;
        if (!g->face_ev[axis] && hipEventCreateWithFlags(&g->face_ev[axis], hipEventDisableTiming) != hipSuccess) return;
        if (hipEventRecord(g->face_ev[axis], g->stream) == hipSuccess) { g->slab_valid |= (axis == 1 ? 2 : 4); g->slab_A = A; g->slab_b = b; }
      };
    }
    g->face_done = face_done;
    // the Tangent of a nonlinear scalar form without a geometry walks the same pencils (gram_mfma.hpp: state_pencil)
    PencilModule st; memset(&st.prm, 0, sizeof(st.prm));
    if ((op == OP_JACOBIAN || op == OP_IJACOBIAN) && s.dim == 3 && s.nsd == 3 && s.axis[0].p == 2 && s.env.state_pencil && (g->kernel_choice == 0 || g->kernel_choice == 2)) {
      // ... and on a mapped geometry at p = 2 (state_pencil_geo: the map's second derivatives and the state's in one sum factorisation)
      if (s.form == IGX_FORM_CAHNHILLIARD) {
        st.kfn = s.rational ? state_pencil_geo<2, true, FormCahnHilliard<3>> : state_pencil_geo<2, false, FormCahnHilliard<3>>; st.name = "CahnHilliard";
        st.flop_per_element = 2048.0 * FormCahnHilliard<3>::PENCIL_NFEAT * 7 * 9;
      }
      if (s.form == IGX_FORM_BRATU) {
        st.kfn = s.rational ? state_pencil_geo<2, true, FormBratu<3>> : state_pencil_geo<2, false, FormBratu<3>>; st.name = "Bratu";
        st.flop_per_element = 2048.0 * FormBratu<3>::PENCIL_NFEAT * 7 * 9;
      }
      if (st.kfn && s.env.p2_pack != 0) {      // packed tiles (state_pencil_geo_k): 4 MFMAs per feature and k-step instead of 9
        st.pack = 2;
        if (s.form == IGX_FORM_CAHNHILLIARD) { st.kfn = s.rational ? state_pencil_geo_k<true, FormCahnHilliard<3>> : state_pencil_geo_k<false, FormCahnHilliard<3>>; st.flop_per_element = 2048.0 * FormCahnHilliard<3>::PENCIL_NFEAT * 7 * 4; }
        else { st.kfn = s.rational ? state_pencil_geo_k<true, FormBratu<3>> : state_pencil_geo_k<false, FormBratu<3>>; st.flop_per_element = 2048.0 * FormBratu<3>::PENCIL_NFEAT * 7 * 4; }
      }
      if (st.kfn) {
        st.state = true; st.state_geo = true; st.extra_lds = pencil_state_bytes() + pencil_sgeo_bytes() - pencil_geo_bytes();
        for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) st.prm.v[i] = s.params[i];
      }
    }
    // (asked for by name, a Tangent on a mapped geometry at another degree is refused with the reason; the automatic choice goes on to the feature kernel)
    if ((op == OP_JACOBIAN || op == OP_IJACOBIAN) && s.dim == 3 && s.nsd == 3 && s.axis[0].p != 2 && g->kernel_choice == 2 && (s.form == IGX_FORM_CAHNHILLIARD || s.form == IGX_FORM_BRATU))
      return fail(IGX_ERR_SUP, "a Tangent on a mapped geometry takes the walk at p = 2 only (state_pencil_geo); the feature kernel covers the other degrees");
    if ((op == OP_JACOBIAN || op == OP_IJACOBIAN) && s.dim == 3 && s.nsd == 0 && s.env.state_pencil && (g->kernel_choice == 0 || g->kernel_choice == 2)) {
      const int deg = s.axis[0].p;
      if (s.form == IGX_FORM_CAHNHILLIARD && (deg == 2 || deg == 3)) {
        st.kfn = deg == 2 ? state_pencil<2, FormCahnHilliard<3>> : state_pencil<3, FormCahnHilliard<3>>; st.name = "CahnHilliard";
        st.flop_per_element = 2048.0 * FormCahnHilliard<3>::PENCIL_NFEAT * (deg == 2 ? 7 * 9 : 16 * 16);     // executed: k-steps x tiles x features
      }
      if (s.form == IGX_FORM_BRATU && (deg == 2 || deg == 3)) {
        st.kfn = deg == 2 ? state_pencil<2, FormBratu<3>> : state_pencil<3, FormBratu<3>>; st.name = "Bratu";
        st.flop_per_element = 2048.0 * FormBratu<3>::PENCIL_NFEAT * (deg == 2 ? 7 * 9 : 16 * 16);
      }
      // p = 2: packed tiles (state_pencil_k: 4 MFMAs per feature and k-step instead of 9; IGX_P2_PACK=0: the layer-pair tiles)
      if (st.kfn && deg == 2 && s.env.p2_pack != 0) {
        st.pack = 1;
        if (s.form == IGX_FORM_CAHNHILLIARD) { st.kfn = fused ? state_pencil_kr<FormCahnHilliard<3>> : state_pencil_k<FormCahnHilliard<3>>; st.flop_per_element = 2048.0 * FormCahnHilliard<3>::PENCIL_NFEAT * 7 * 4; }
        else { st.kfn = fused ? state_pencil_kr<FormBratu<3>> : state_pencil_k<FormBratu<3>>; st.flop_per_element = 2048.0 * FormBratu<3>::PENCIL_NFEAT * 7 * 4; }
        if (fused) st.name += "+Residual";
        else st.patch_kfn = s.form == IGX_FORM_CAHNHILLIARD ? reinterpret_cast<const void *>(state_patch_p2<FormCahnHilliard<3>>) : reinterpret_cast<const void *>(state_patch_p2<FormBratu<3>>);
      }
      if (st.kfn) { st.state = true; st.extra_lds = pencil_state_bytes() + (fused ? (size_t)8 * (RWIN_DOUBLES + 128) * 8 : 0) /* fused: the Residual's ring and the scratch of its u_t sums, per wavefront */; for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) st.prm.v[i] = s.params[i]; }
    }
    rc = try_gram_mfma(g->s, S, out, g->stream, g->kernel_choice == 2, g->last_kernel, g->last_launches, g_err, done, g->dom, zero_matrix, slab_done, st.kfn ? &st : nullptr, face_done);
    g->face_done = nullptr;
    if (rc) return rc;
  }
  if (fused) {      // (no walk took it -- a non-walkable axis 0, a reduced continuity: the caller falls back to the two drivers)
    *fused_done = done;
    if (g->timing) { HIPCK(hipEventRecord(g->ev[2], g->stream)); HIPCK(hipEventRecord(g->ev[3], g->stream)); }
    return 0;
  }
  if (!done) {
    g->zero_matrix = zero_matrix;      // the feature kernel stores first touches and skips it; everything else zeroes first
    g->slab_valid = 0;
    if (g->comm && s.env.overlap) g->slab_done = [&]() {
      if (!g->slab_ev && hipEventCreateWithFlags(&g->slab_ev, hipEventDisableTiming) != hipSuccess) return;
      if (hipEventRecord(g->slab_ev, g->stream) == hipSuccess) { g->slab_valid |= 1; g->slab_A = A; g->slab_b = b; }
    };
    if (g->comm && s.env.overlap && s.env.overlap != 2) g->face_done = [&](int axis) {      // (run-time forms on the pencil walk: rtc.hpp; IGX_OVERLAP=2: the mark of axis 2 alone, as for the built-in forms)
      if (axis < 0 || axis > 1) return;
      if (!g->face_ev[axis] && hipEventCreateWithFlags(&g->face_ev[axis], hipEventDisableTiming) != hipSuccess) return;
      if (hipEventRecord(g->face_ev[axis], g->stream) == hipSuccess) { g->slab_valid |= (axis == 1 ? 2 : 4); g->slab_A = A; g->slab_b = b; }
    };
    rc = (s.form == IGX_FORM_SOURCE) ? launch_generic_rtc(g, S, out) : dispatch_by_dim(g, S, out);
    g->zero_matrix = nullptr; g->slab_done = nullptr; g->face_done = nullptr;
    if (rc) return rc;
  }
  if (g->timing) { HIPCK(hipEventRecord(g->ev[2], g->stream)); HIPCK(hipEventRecord(g->ev[3], g->stream)); }
  if (kDebug && out.dbg) {
    long long h[32]; HIPCK(hipStreamSynchronize(g->stream)); HIPCK(hipMemcpy(h, g->dbgbuf.p, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[feature stamps]"); for (int i = 1; i < (int)h[31] && i < 31; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]); fprintf(stderr, "\n");
  }
  return 0;
}

extern "C" int IGXComputeSystem(IGX g, IGXMat A, IGXVec b) { return compute(g, OP_SYSTEM, A, b, nullptr, nullptr, 0, 0); }
extern "C" int IGXComputeMatrix(IGX g, IGXMat A) { return compute(g, OP_MATRIX, A, nullptr, nullptr, nullptr, 0, 0); }
extern "C" int IGXComputeVector(IGX g, IGXVec b) { return compute(g, OP_VECTOR, nullptr, b, nullptr, nullptr, 0, 0); }
extern "C" int IGXComputeFunction(IGX g, IGXVec U, IGXVec F) { if (!U) return fail(IGX_ERR_ARG_WRONG, "null state vector"); return compute(g, OP_FUNCTION, nullptr, F, U, nullptr, 0, 0); }
extern "C" int IGXComputeJacobian(IGX g, IGXVec U, IGXMat J) { if (!U) return fail(IGX_ERR_ARG_WRONG, "null state vector"); return compute(g, OP_JACOBIAN, J, nullptr, U, nullptr, 0, 0); }
extern "C" int IGXComputeIFunction(IGX g, double a, IGXVec V, double t, IGXVec U, IGXVec F) { if (!U || !V) return fail(IGX_ERR_ARG_WRONG, "null state vector"); return compute(g, OP_IFUNCTION, nullptr, F, U, V, a, t); }
extern "C" int IGXComputeIJacobian(IGX g, double a, IGXVec V, double t, IGXVec U, IGXMat J) { if (!U || !V) return fail(IGX_ERR_ARG_WRONG, "null state vector"); return compute(g, OP_IJACOBIAN, J, nullptr, U, V, a, t); }
// One pass for the pair a Newton step asks for at the same state (SNESComputeFunction + SNESComputeJacobian; src/petigats.c:23-159,
// src/petigasnes.c:23-139): the results are those of the two drivers, in one walk where a fused kernel exists, else in two calls.
extern "C" int IGXComputeIFunctionIJacobian(IGX g, double a, IGXVec V, double t, IGXVec U, IGXVec F, IGXMat J) {
  if (!U || !V) return fail(IGX_ERR_ARG_WRONG, "null state vector");
  bool done = false;
  if (int rc = compute(g, OP_IJACOBIAN, J, F, U, V, a, t, &done)) return rc;
  if (done) return 0;
  if (int rc = compute(g, OP_IFUNCTION, nullptr, F, U, V, a, t)) return rc;
  const std::string kf = g->last_kernel;
  if (int rc = compute(g, OP_IJACOBIAN, J, nullptr, U, V, a, t)) return rc;
  g->last_kernel = kf + " | " + g->last_kernel;
  return 0;
}
extern "C" int IGXComputeFunctionJacobian(IGX g, IGXVec U, IGXVec F, IGXMat J) {
  if (!U) return fail(IGX_ERR_ARG_WRONG, "null state vector");
  bool done = false;
  if (int rc = compute(g, OP_JACOBIAN, J, F, U, nullptr, 0, 0, &done)) return rc;
  if (done) return 0;
  if (int rc = compute(g, OP_FUNCTION, nullptr, F, U, nullptr, 0, 0)) return rc;
  const std::string kf = g->last_kernel;
  if (int rc = compute(g, OP_JACOBIAN, J, nullptr, U, nullptr, 0, 0)) return rc;
  g->last_kernel = kf + " | " + g->last_kernel;
  return 0;
}

// ------------------------------------------------------------------ IGAComputeScalar (src/petigacomp.c:35-98)
extern "C" int IGXComputeScalar(IGX g, IGXVec U, int kind, const double params[], int nparams, int n, double S[]) {
  NEEDIGA(g);
  if (int rc = ensure_device(g)) return rc;
  Space &s = g->s;
  if (U && U->iga != g) return fail(IGX_ERR_ARG_WRONG, "state vector created by another IGX");
  if (!S || n < 1) return fail(IGX_ERR_ARG_WRONG, "null result array");
  const int ns = (kind == IGX_SCALAR_ERRNORM) ? 4 : (kind == IGX_SCALAR_VOLUME ? 2 : 1);
  if (n != ns) return fail(IGX_ERR_ARG_WRONG, "this functional returns " + std::to_string(ns) + " scalars");
  if (nparams < 0 || nparams > MAXPARAM || (nparams && !params)) return fail(IGX_ERR_ARG_OUTOFRANGE, "bad parameter list");
  const int order = (kind == IGX_SCALAR_ERRNORM && nparams > 0) ? (int)params[0] : 0;
  if (order < 0 || order > 2) return fail(IGX_ERR_ARG_OUTOFRANGE, "derivative order must be in range [0,2]");
  int64_t nel = (int64_t)s.elem_width[0] * s.elem_width[1] * s.elem_width[2];
  for (int a = 0; a < s.dim; ++a) for (int sd = 0; sd < 2; ++sd) {   // one more partial row per element of a visited face
    const int eface = sd ? s.elem_sizes[a] - 1 : 0;
    if (s.visit[a][sd] && eface >= s.elem_start[a] && eface < s.elem_start[a] + s.elem_width[a]) nel += (int64_t)s.elem_width[0] * s.elem_width[1] * s.elem_width[2] / s.elem_width[a];
  }
  const int nblk = (int)std::min<int64_t>(1024, (nel + 255) / 256);
  const int64_t chunk = (nel + nblk - 1) / nblk;
  const size_t need = ((size_t)nel + nblk + 1) * ns * sizeof(double);
  if (g->partials.bytes < need) { HIPCK(hipStreamSynchronize(g->stream)); if (g->partials.alloc(need)) return fail(IGX_ERR_MEM, "partial-sum buffer allocation failed"); }
  double *part = g->partials.as<double>(), *stage = part + (size_t)nel * ns, *res = stage + (size_t)nblk * ns;
  HIPCK(hipMemsetAsync(part, 0, (size_t)nel * ns * sizeof(double), g->stream));
  OutDev out; memset(&out, 0, sizeof(out));
  out.op = OP_SCALAR; out.bid = -1; out.errflag = g->errflag.as<int>(); out.vec = part; out.U = U ? U->a.as<double>() : nullptr;
  const SpaceDev Sd = make_spacedev(g);
  const std::vector<double> keep = s.params;            // the functional's parameters travel like a form's
  s.params.assign(params, params + nparams);
  int rc;
  switch (s.dim) {
  case 1: rc = igx_tu_scalar(std::integral_constant<int, 1>(), g, kind, Sd, out, order); break;
  case 2: rc = igx_tu_scalar(std::integral_constant<int, 2>(), g, kind, Sd, out, order); break;
  default: rc = igx_tu_scalar(std::integral_constant<int, 3>(), g, kind, Sd, out, order); break;
  }
  s.params = keep;
  if (rc) return rc;
  hipLaunchKernelGGL(k_sum_partials, dim3(nblk), dim3(256), 0, g->stream, part, nel, ns, stage, chunk);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, g->stream, stage, (int64_t)nblk, ns, res, (int64_t)nblk);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(S, res, ns * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  HIPCK(hipStreamSynchronize(g->stream));
  int flag = 0; HIPCK(hipMemcpy(&flag, g->errflag.p, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) { HIPCK(hipMemset(g->errflag.p, 0, sizeof(int))); return fail(flag, "Non-positive det(Jacobian) of the geometry mapping"); }
  return 0;
}

// ------------------------------------------------------------------ checksums over the owned rows
// one workgroup per stride of rows; fixed grid and a fixed in-block tree: bitwise repeatable
__global__ void __launch_bounds__(256) k_checksum(int nown0, int nown1, int nown2, int nrow0, int nrow1, int bs, const int64_t *browptr, const double *val, const double *vec, double *part) {
  __shared__ double red[256];
  const int64_t nown = (int64_t)nown0 * nown1 * nown2;
  double s[4] = {0, 0, 0, 0};
  for (int64_t k = blockIdx.x; k < nown; k += gridDim.x) {
    const int r0 = (int)(k % nown0), r1 = (int)((k / nown0) % nown1), r2 = (int)(k / ((int64_t)nown0 * nown1));
    const int64_t row = (int64_t)r0 + (int64_t)nrow0 * ((int64_t)r1 + (int64_t)nrow1 * r2);
    if (val) {
      const int64_t lo = browptr[row] * bs * bs, hi = browptr[row + 1] * bs * bs;
      for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const double v = val[i]; s[0] += v; s[1] += fabs(v); }
    }
    if (vec && (int)threadIdx.x < bs) { const double v = vec[row * bs + threadIdx.x]; s[2] += v; s[3] += v * v; }
  }
  for (int c = 0; c < 4; ++c) {
    red[threadIdx.x] = s[c]; __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * 4 + c] = red[0];
    __syncthreads();
  }
}

extern "C" int IGXChecksum(IGX g, IGXMat A, IGXVec b, double S[4]) {
  NEEDIGA(g);
  if (int rc = ensure_device(g)) return rc;
  if (!S) return fail(IGX_ERR_ARG_WRONG, "null result array");
  if ((A && A->iga != g) || (b && b->iga != g)) return fail(IGX_ERR_ARG_WRONG, "matrix / vector created by another IGX");
  const Space &s = g->s;
  int nown[3];
  for (int d = 0; d < 3; ++d) {   // owned rows are a prefix of the row box on every axis (ghosts sit on the high side, src/petiga.c:1172-1208)
    const AxisLayout &L = s.lay[d]; nown[d] = 0;
    for (int r = 0; r < L.nrow; ++r) { if (L.owned[r]) { if (nown[d] != r) return fail(IGX_ERR_PLIB, "owned rows are not a prefix of the row box"); nown[d]++; } }
  }
  const int nblk = 2048;
  const size_t need = ((size_t)nblk + 1) * 4 * sizeof(double);
  if (g->partials.bytes < need) { HIPCK(hipStreamSynchronize(g->stream)); if (g->partials.alloc(need)) return fail(IGX_ERR_MEM, "partial-sum buffer allocation failed"); }
  double *part = g->partials.as<double>(), *res = part + (size_t)nblk * 4;
  hipLaunchKernelGGL(k_checksum, dim3(nblk), dim3(256), 0, g->stream, nown[0], nown[1], nown[2], s.lay[0].nrow, s.lay[1].nrow, s.dof,
                     A ? A->browptr.as<int64_t>() : nullptr, A ? A->val.as<double>() : nullptr, b ? b->a.as<double>() : nullptr, part);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, g->stream, part, (int64_t)nblk, 4, res, (int64_t)nblk);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(S, res, 4 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
  HIPCK(hipStreamSynchronize(g->stream));
  return 0;
}

extern "C" int IGXGetClockProbe(IGX g, double *shader_mhz, int64_t *elements) {
  NEEDIGA(g);
  if (!shader_mhz) return fail(IGX_ERR_ARG_WRONG, "null result");
  if (!g->s.env.clock_probe || !g->clkbuf.p) return fail(IGX_ERR_ARG_WRONGSTATE, "set IGX_CLOCK_PROBE=1 before IGXCreate and run an assembly on the pencil kernel first");
  long long h[4];
  HIPCK(hipStreamSynchronize(g->stream));
  HIPCK(hipMemcpy(h, g->clkbuf.p, sizeof(h), hipMemcpyDeviceToHost));
  if (h[1] <= 0) return fail(IGX_ERR_ARG_WRONGSTATE, "no pencil-kernel launch since the last call");
  *shader_mhz = 100.0 * (double)h[0] / (double)h[1];
  if (elements) *elements = h[2];
  HIPCK(hipMemsetAsync(g->clkbuf.p, 0, 4 * sizeof(long long), g->stream));   // the sums start again
  return 0;
}

// ------------------------------------------------------------------ multi-GPU ghost rows (filled in by exchange.hpp)
#include "exchange.hpp"
#include "comm.hpp"
#include "coo.hpp"
#include "fileio.hpp"
#include "rtc.hpp"
#endif   // !IGX_TU_DISPATCH
