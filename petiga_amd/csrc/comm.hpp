// comm.hpp -- the transport of the ghost-row exchange, inside the library (included by engine.hip after exchange.hpp).
//
// Reference: the only communication on the assembly path is MatAssemblyBegin/End + VecAssemblyBegin/End
// (src/petigaksp.c:197-200; PETSc's stash moves the rows of not-owned nodes to their owners) and, before a nonlinear
// assembly, DMGlobalToLocal of the state vectors (IGAGetLocalVecArray, src/petigavec.c:256-269).  Here a rank's ghost rows
// go point-to-point to its <= 7 upper neighbours: grouped ncclSend / ncclRecv on RCCL (one xGMI link per neighbour pair of
// a 2x2x2 grid, all messages concurrent), on the library's own exchange stream.  The engine stream and the exchange stream
// meet through events only: IGXReduceGhostRows / IGXRefreshGhosts never block the host.
//
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, a copy that is already loaded in the
// process (e.g. the one PyTorch ships) is reused.  A second transport takes a host callback that is handed the packed device
// buffers: the test transport (gloo in tests / bench.py with ranks sharing one GPU) and the hook for an MPI-based caller.
#include <dlfcn.h>

namespace {

struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(void *) = nullptr;                                           // ncclGetUniqueId(ncclUniqueId*)
  int (*CommInitRank)(void **, int, IGXUniqueId, int) = nullptr;                  // ncclCommInitRank(comm*, nranks, id (by value), rank)
  int (*CommDestroy)(void *) = nullptr;
  int (*CommCount)(void *, int *) = nullptr;                                      // ncclCommCount(comm, int*)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;     // ncclSend(buf, count, datatype, peer, comm, stream)
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclDouble = 8;   // ncclFloat64 (rccl.h)

static int load_rccl(RcclApi &api, const char *path, std::string &err) {
  if (api.handle) return 0;
  // A library the caller names (argument, then $IGX_RCCL_LIB) is the one that is bound, whatever else the process has loaded
  // (PyTorch maps its own librccl.so: a test double named here must still win).  Otherwise a copy that is already loaded is
  // reused, and only then a fresh one.  RTLD_LOCAL: the symbols are taken with dlsym, nobody else should resolve against them.
  const char *env = getenv("IGX_RCCL_LIB");
  for (const char *n : {path, env}) {
    if (!n || !*n) continue;
    api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) {     // dlerror() hands its message out once and clears it: read it into a local
      const char *why = dlerror();
      err = std::string("cannot load ") + n + ": " + (why ? why : "not found");
      return IGX_ERR_LIB;
    }
    break;
  }
  const char *names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !api.handle; ++pass)     // first a copy that is already loaded, then a fresh one
    for (const char *n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (api.handle) break;
    }
  if (!api.handle) {
    const char *why = dlerror();
    err = std::string("cannot load librccl.so: ") + (why ? why : "not found");
    return IGX_ERR_LIB;
  }
  auto sym = [&](const char *n) { return dlsym(api.handle, n); };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv) {
    err = "librccl.so lacks the point-to-point API (ncclSend / ncclRecv)"; dlclose(api.handle); api = RcclApi(); return IGX_ERR_LIB;
  }
  return 0;
}

static RcclApi &rccl_api() { static RcclApi a; return a; }

}  // namespace

struct IgxComm {
  int kind = 0;                       // 1 RCCL, 2 host callback
  void *nccl = nullptr;               // ncclComm_t
  IGXTransportFn fn = nullptr; void *fnctx = nullptr;
  hipStream_t xs = nullptr;           // exchange stream
  hipEvent_t ready = nullptr, done = nullptr;
  hipEvent_t packed1 = nullptr; bool packed1_valid = false;   // exchange stream, after the first phase's messages were packed (IGXCommGetOverlap)
  std::vector<DevBuf> sbuf, rbuf;     // one per neighbour of the larger of the two lists
  int64_t last_bytes = 0;
  int early_phases = 0;               // phases of the last reduction that started on a face mark of the assembly (IGXCommGetEarlyPhases)
  double link_gbs = 0; int link_source = 0;      // IGXCommGetLinkRate: GB/s per direction of a face message; 0 assumed constant, 1 measured at init, 2 $IGX_LINK_GBS
  double link_probe_ms = 0; int link_faces = 0;
  ~IgxComm() {
    if (xs) (void)hipStreamSynchronize(xs);      // nothing of this communicator is in flight when it goes
    if (nccl && rccl_api().CommDestroy) (void)rccl_api().CommDestroy(nccl);
    if (ready) (void)hipEventDestroy(ready);
    if (done) (void)hipEventDestroy(done);
    if (packed1) (void)hipEventDestroy(packed1);
    if (xs) (void)hipStreamDestroy(xs);
  }
};

#define NCCLCK(call) do { int r_ = (call); if (r_ != 0) return fail(IGX_ERR_LIB, std::string(#call) + ": " + (rccl_api().GetErrorString ? rccl_api().GetErrorString(r_) : "RCCL error")); } while (0)

static int comm_common_init(IGX g, std::unique_ptr<IgxComm> &c) {
  HIPCK(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
  HIPCK(hipEventCreate(&c->ready));      // (with time stamps: IGXCommGetOverlap)
  HIPCK(hipEventCreate(&c->packed1));
  HIPCK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  g->comm = std::move(c);
  return 0;
}

// The rate a face message travels at, measured once per communicator: the face-first decision of the pencil walks weighs the cost
// of its extra passes against (largest face) / (this rate) (gram_mfma.hpp), and a constant (60 GB/s per direction was the guess
// of rounds 3-5) would make the first run on real links a matter of luck.  Every rank exchanges IGX_LINK_PROBE_MB (default 64) MB
// with each FACE neighbour of its ghost-row reduction -- the same grouped ncclSend / ncclRecv, the same stream, the same
// direction as the real exchange, all faces at once as they travel in the real exchange -- twice: the first group also pays
// RCCL's connection set-up, the second one is timed with events.  $IGX_LINK_GBS, when set, is taken instead (no probe).
// Collective over the communicator (every rank calls IGXCommInitRCCL); a rank without a face neighbour keeps the default.
static int comm_probe_links(IGX g) {
  IgxComm &c = *g->comm;
  const char *lr = getenv("IGX_LINK_GBS");
  if (lr && atof(lr) > 0) { c.link_gbs = atof(lr); c.link_source = 2; g->s.link_gbs = c.link_gbs; return 0; }
  c.link_gbs = 60.0; c.link_source = 0; g->s.link_gbs = 0;
  const char *pm = getenv("IGX_LINK_PROBE_MB");
  const double mb = pm ? atof(pm) : 64.0;
  if (g->s.comm_size < 2 || mb <= 0) return 0;
  auto face = [](const NbrPlan &p) { return (p.off[0] != 0) + (p.off[1] != 0) + (p.off[2] != 0) == 1; };
  std::vector<int> to, from;
  for (const NbrPlan &p : neighbour_plans(g->s, true)) if (face(p)) to.push_back(p.rank);
  for (const NbrPlan &p : neighbour_plans(g->s, false)) if (face(p)) from.push_back(p.rank);
  if (to.empty() && from.empty()) return 0;
  const size_t n = (size_t)(mb * 1e6 / 8);
  std::vector<DevBuf> sb(to.size()), rb(from.size());
  for (DevBuf &b : sb) { if (b.alloc(n * 8)) return fail(IGX_ERR_MEM, "link probe: buffer allocation failed"); HIPCK(hipMemsetAsync(b.p, 0, n * 8, c.xs)); }
  for (DevBuf &b : rb) if (b.alloc(n * 8)) return fail(IGX_ERR_MEM, "link probe: buffer allocation failed");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
  int rc = 0;
  for (int rep = 0; rep < 2 && !rc; ++rep) {
    if (hipEventRecord(e0, c.xs) != hipSuccess) { rc = IGX_ERR_LIB; break; }
    int r1 = rccl_api().GroupStart(), r2 = 0;
    for (size_t k = 0; k < from.size() && !r1 && !r2; ++k) r2 = rccl_api().Recv(rb[k].p, n, kNcclDouble, from[k], c.nccl, c.xs);
    for (size_t k = 0; k < to.size() && !r1 && !r2; ++k) r2 = rccl_api().Send(sb[k].p, n, kNcclDouble, to[k], c.nccl, c.xs);
    const int r3 = r1 ? 0 : rccl_api().GroupEnd();
    if (r1 || r2 || r3) { rc = IGX_ERR_LIB; g_err = std::string("link probe: ") + (rccl_api().GetErrorString ? rccl_api().GetErrorString(r1 ? r1 : (r2 ? r2 : r3)) : "RCCL error"); break; }
    if (hipEventRecord(e1, c.xs) != hipSuccess) { rc = IGX_ERR_LIB; break; }
  }
  if (!rc && hipStreamSynchronize(c.xs) != hipSuccess) { rc = IGX_ERR_LIB; g_err = "link probe: the exchange stream failed"; }
  float ms = 0;
  if (!rc && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) {
    c.link_probe_ms = ms; c.link_faces = (int)std::max(to.size(), from.size());
    c.link_gbs = (double)n * 8 / (ms * 1e-3) / 1e9; c.link_source = 1; g->s.link_gbs = c.link_gbs;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

// gbs: the rate the face-first decision uses; source: 0 the constant (no probe: one rank, no face neighbour, IGX_LINK_PROBE_MB=0),
// 1 measured at IGXCommInitRCCL, 2 $IGX_LINK_GBS; probe_ms / faces: the timed group and the face messages per direction it held
extern "C" int IGXCommGetLinkRate(IGX g, double *gbs, int *source, double *probe_ms, int *faces) {
  NEEDIGA(g); if (!g->comm) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGXCommInitRCCL() / IGXCommInitTransport() first");
  if (gbs) *gbs = g->comm->link_gbs > 0 ? g->comm->link_gbs : 60.0;
  if (source) *source = g->comm->link_source;
  if (probe_ms) *probe_ms = g->comm->link_probe_ms;
  if (faces) *faces = g->comm->link_faces;
  return 0;
}

extern "C" int IGXCommGetUniqueId(IGXUniqueId *id, const char *librccl_path) {
  if (!id) return fail(IGX_ERR_ARG_WRONG, "null id");
  std::string e; if (int rc = load_rccl(rccl_api(), librccl_path, e)) return fail(rc, e);
  NCCLCK(rccl_api().GetUniqueId(id));
  return 0;
}

extern "C" int IGXCommInitRCCL(IGX g, const IGXUniqueId *id, const char *librccl_path) {
  NEEDIGA(g); if (!id) return fail(IGX_ERR_ARG_WRONG, "null id");
  if (!g->s.setup) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetUp() first");
  std::string e; if (int rc = load_rccl(rccl_api(), librccl_path, e)) return fail(rc, e);
  std::unique_ptr<IgxComm> c(new IgxComm());
  c->kind = 1;
  NCCLCK(rccl_api().CommInitRank(&c->nccl, g->s.comm_size, *id, g->s.comm_rank));
  if (int rc = comm_common_init(g, c)) return rc;
  return comm_probe_links(g);
}

extern "C" int IGXCommInitTransport(IGX g, IGXTransportFn fn, void *ctx) {
  NEEDIGA(g); if (!fn) return fail(IGX_ERR_ARG_WRONG, "null transport");
  std::unique_ptr<IgxComm> c(new IgxComm());
  c->kind = 2; c->fn = fn; c->fnctx = ctx;
  const char *lr = getenv("IGX_LINK_GBS");
  if (lr && atof(lr) > 0) { c->link_gbs = atof(lr); c->link_source = 2; g->s.link_gbs = c->link_gbs; } else { c->link_gbs = 60.0; g->s.link_gbs = 0; }
  return comm_common_init(g, c);
}

extern "C" int IGXCommDestroy(IGX g) { NEEDIGA(g); if (g->comm) { (void)hipStreamSynchronize(g->comm->xs); g->comm.reset(); } g->s.link_gbs = 0; return 0; }

// One exchange: pack `npack` messages (list `pack_send_list`), move them, unpack the `nunp` received ones.  reduce = true:
// ghost rows to their owners (added); false: owner values to the ghosts (assigned; vectors only).
static int comm_exchange(IGX g, IGXMat A, IGXVec b, bool reduce) {
  NEEDIGA(g);
  if (!g->comm) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGXCommInitRCCL() / IGXCommInitTransport() first");
  if (int rc = ensure_device(g)) return rc;
  { std::string e; if (int rc = exchange_supported(g->s, e)) return fail(rc, e); }
  IgxComm &c = *g->comm;
  // reduce: I pack my SEND list (upper neighbours) and unpack my RECEIVE list; refresh: the other way round
  const std::vector<NbrPlan> out_plans = neighbour_plans(g->s, reduce), in_plans = neighbour_plans(g->s, !reduce);
  auto doubles = [&](const NbrPlan &p) { return (A ? p.mat_doubles : 0) + (b ? p.vec_doubles : 0); };
  if (c.sbuf.size() < out_plans.size()) c.sbuf.resize(out_plans.size());
  if (c.rbuf.size() < in_plans.size()) c.rbuf.resize(in_plans.size());
  bool grew = false;
  for (size_t k = 0; k < out_plans.size(); ++k) if (c.sbuf[k].bytes < (size_t)doubles(out_plans[k]) * 8) { if (!grew) { HIPCK(hipStreamSynchronize(c.xs)); grew = true; } if (c.sbuf[k].alloc((size_t)doubles(out_plans[k]) * 8)) return fail(IGX_ERR_MEM, "exchange buffer allocation failed"); }
  for (size_t k = 0; k < in_plans.size(); ++k) if (c.rbuf[k].bytes < (size_t)doubles(in_plans[k]) * 8) { if (!grew) { HIPCK(hipStreamSynchronize(c.xs)); grew = true; } if (c.rbuf[k].alloc((size_t)doubles(in_plans[k]) * 8)) return fail(IGX_ERR_MEM, "exchange buffer allocation failed"); }
  // Ghost-row reduction in up to three phases: the messages of the upper face of axis 2 (offsets with o2 >= 1) first, then those of
  // axis 1 (o2 = 0, o1 >= 1), then those of axis 0 alone.  Every rank issues the groups in this order; a rank whose assembly marked
  // the moment the rows of a face were complete (slab_ev / face_ev, engine.hip) starts that face's group there, under its remaining
  // launches -- pack and wire time leave the critical path.  The unpack adds into rows the receiver's own launches store into, so it
  // waits for the end of the assembly either way.
  const bool phased = reduce && g->s.env.overlap;   // (the environment is the same on every rank)
  // What the marked assembly wrote is face-complete at its marks, and a matrix / vector an EARLIER call wrote was complete before
  // them (one engine stream).  A mark is only good while it is the last write: every entry point that writes to an IGXMat /
  // IGXVec afterwards clears it (IGXCompute*, IGXVecCopyFromHost, IGXVecCopyFromGhosted, IGXReadVec, IGXUnpackGhost*, every
  // exchange), and the reduction then packs where the engine stream stands.
  const int marks = phased ? g->slab_valid : 0;      // bit 0: upper face of axis 2, bit 1: axis 1, bit 2: axis 0 (engine.hip)
  g->slab_valid = 0;
  HIPCK(hipEventRecord(c.ready, g->stream));
  c.last_bytes = 0; c.packed1_valid = false; c.early_phases = 0;
  // phase 1: the messages with o2 >= 1; phase 2: o2 = 0 and o1 >= 1; phase 3: o2 = o1 = 0 (o0 >= 1); phase 0: everything
  auto in_phase = [&](const NbrPlan &p, int phase) { return phase == 0 || (phase == 1 ? p.off[2] >= 1 : (phase == 2 ? (p.off[2] == 0 && p.off[1] >= 1) : (p.off[2] == 0 && p.off[1] == 0))); };
  auto pack = [&](int phase) -> int {
    for (size_t k = 0; k < out_plans.size(); ++k) {
      if (doubles(out_plans[k]) == 0 || !in_phase(out_plans[k], phase)) continue;
      if (int rc = ghost_rows(g, A, b, (int)k, c.sbuf[k].as<double>(), reduce, 0, c.xs)) return rc;
      c.last_bytes += doubles(out_plans[k]) * 8;
    }
    return 0;
  };
  auto move = [&](int phase) -> int {      // the messages of a phase: one RCCL group, or one call of the host transport
    if (c.kind == 1) {
      bool any = false;
      for (size_t k = 0; k < in_plans.size() && !any; ++k) any = doubles(in_plans[k]) && in_phase(in_plans[k], phase);
      for (size_t k = 0; k < out_plans.size() && !any; ++k) any = doubles(out_plans[k]) && in_phase(out_plans[k], phase);
      if (!any) return 0;
      NCCLCK(rccl_api().GroupStart());
      int rc_ = 0;      // (an error inside the group still closes it: RCCL must not be left with an open group)
      for (size_t k = 0; k < in_plans.size() && !rc_; ++k) if (doubles(in_plans[k]) && in_phase(in_plans[k], phase)) rc_ = rccl_api().Recv(c.rbuf[k].p, (size_t)doubles(in_plans[k]), kNcclDouble, in_plans[k].rank, c.nccl, c.xs);
      for (size_t k = 0; k < out_plans.size() && !rc_; ++k) if (doubles(out_plans[k]) && in_phase(out_plans[k], phase)) rc_ = rccl_api().Send(c.sbuf[k].p, (size_t)doubles(out_plans[k]), kNcclDouble, out_plans[k].rank, c.nccl, c.xs);
      const int re_ = rccl_api().GroupEnd();
      NCCLCK(rc_);
      NCCLCK(re_);
      return 0;
    }
    std::vector<int> sp, rp; std::vector<double *> sb, rb; std::vector<int64_t> sn, rn;
    for (size_t k = 0; k < out_plans.size(); ++k) if (doubles(out_plans[k]) && in_phase(out_plans[k], phase)) { sp.push_back(out_plans[k].rank); sb.push_back(c.sbuf[k].as<double>()); sn.push_back(doubles(out_plans[k])); }
    for (size_t k = 0; k < in_plans.size(); ++k) if (doubles(in_plans[k]) && in_phase(in_plans[k], phase)) { rp.push_back(in_plans[k].rank); rb.push_back(c.rbuf[k].as<double>()); rn.push_back(doubles(in_plans[k])); }
    if (sp.empty() && rp.empty()) return 0;
    HIPCK(hipStreamSynchronize(c.xs));      // a host transport reads the packed buffers (the engine stream keeps running)
    if (int rc = c.fn(c.fnctx, (int)sp.size(), sp.data(), sb.data(), sn.data(), (int)rp.size(), rp.data(), rb.data(), rn.data())) return fail(IGX_ERR_LIB, "transport callback failed with code " + std::to_string(rc));
    return 0;
  };
  // What the exchange stream waits for.  Only a PACK reads what the engine stream writes; a receive lands in a buffer of the
  // exchange itself.  So a phase in which this rank sends nothing is issued at once -- its receives are posted while the rank
  // still assembles, and the sender's early face message moves under the receiver's remaining launches (in a non-periodic
  // [1,1,2] / [2,2,2] grid the receiver of a face is exactly the rank without an upper neighbour there).  A phase with
  // sends waits for what it packs: the face mark of its axis (a marked assembly) or the assembly's last launch.  Sends and
  // receives of a phase stay in ONE group: split into two groups on one stream, two ranks that both receive first would wait
  // for each other's sends for ever.  Every rank issues the phases in the same order.  The unpack adds into rows the rank's own
  // launches store into: it always waits for c.ready.
  auto sends_in = [&](int phase) { for (size_t k = 0; k < out_plans.size(); ++k) if (doubles(out_plans[k]) && in_phase(out_plans[k], phase)) return true; return false; };
  if (phased) {
    for (int phase = 1; phase <= 3; ++phase) {
      const int bit = phase == 1 ? 1 : (phase == 2 ? 2 : 4);
      hipEvent_t mark = phase == 1 ? g->slab_ev : g->face_ev[phase == 2 ? 1 : 0];
      const bool early = (marks & bit) != 0 && mark != nullptr;
      if (sends_in(phase)) { HIPCK(hipStreamWaitEvent(c.xs, early ? mark : c.ready, 0)); if (early) c.early_phases++; }
      if (int rc = pack(phase)) return rc;
      if (phase == 1) c.packed1_valid = early && sends_in(1) && hipEventRecord(c.packed1, c.xs) == hipSuccess;
      if (int rc = move(phase)) return rc;
    }
  } else {
    // the exchange stream picks up where the engine stream stands (the assembly's last launch)
    if (sends_in(0)) HIPCK(hipStreamWaitEvent(c.xs, c.ready, 0));
    if (int rc = pack(0)) return rc;
    if (int rc = move(0)) return rc;
  }
  HIPCK(hipStreamWaitEvent(c.xs, c.ready, 0));        // the unpack (and a refresh's assignment) behind everything the engine stream has enqueued
  for (size_t k = 0; k < in_plans.size(); ++k) {
    if (doubles(in_plans[k]) == 0) continue;
    if (int rc = ghost_rows(g, A, b, (int)k, c.rbuf[k].as<double>(), !reduce, reduce ? 1 : 2, c.xs)) return rc;
  }
  // whatever the engine stream does next sees the exchanged rows
  HIPCK(hipEventRecord(c.done, c.xs));
  HIPCK(hipStreamWaitEvent(g->stream, c.done, 0));
  return 0;
}

extern "C" int IGXReduceGhostRows(IGX g, IGXMat A, IGXVec b) {
  if (!A && !b) return fail(IGX_ERR_ARG_WRONG, "nothing to reduce");
  if ((A && A->iga != g) || (b && b->iga != g)) return fail(IGX_ERR_ARG_WRONG, "matrix / vector created by another IGX");
  return comm_exchange(g, A, b, true);
}
extern "C" int IGXRefreshGhosts(IGX g, IGXVec v) {
  if (!v || v->iga != g) return fail(IGX_ERR_ARG_WRONG, "vector missing or created by another IGX");
  return comm_exchange(g, nullptr, v, false);
}
// How far ahead of the end of the assembly the upper face of axis 2 was packed in the last IGXReduceGhostRows: time from "the
// first phase's messages are packed" (exchange stream) to "the assembly's last launch has finished" (engine stream), in ms.
extern "C" int IGXCommGetOverlap(IGX g, double *ms) {
  NEEDIGA(g);
  if (!ms) return fail(IGX_ERR_ARG_WRONG, "null result");
  if (!g->comm || !g->comm->packed1_valid) return fail(IGX_ERR_ARG_WRONGSTATE, "the last ghost-row reduction did not start ahead of the end of its assembly");
  IgxComm &c = *g->comm;
  HIPCK(hipEventSynchronize(c.done));
  float t = 0;
  HIPCK(hipEventElapsedTime(&t, c.packed1, c.ready));
  *ms = t;
  return 0;
}
extern "C" int IGXCommGetRanks(IGX g, int *kind, int *ranks) {
  NEEDIGA(g);
  if (kind) *kind = g->comm ? g->comm->kind : 0;
  if (ranks) {
    *ranks = g->comm ? g->s.comm_size : 0;
    if (g->comm && g->comm->kind == 1) {
      if (!rccl_api().CommCount) return fail(IGX_ERR_LIB, "librccl.so lacks ncclCommCount");
      NCCLCK(rccl_api().CommCount(g->comm->nccl, ranks));
    }
  }
  return 0;
}
// how many phases of the last IGXReduceGhostRows were packed behind a face mark of the assembly instead of its last launch (0..3)
extern "C" int IGXCommGetEarlyPhases(IGX g, int *n) { NEEDIGA(g); if (!n) return fail(IGX_ERR_ARG_WRONG, "null result"); *n = g->comm ? g->comm->early_phases : 0; return 0; }
extern "C" int IGXCommGetLastBytes(IGX g, int64_t *bytes) { NEEDIGA(g); if (bytes) *bytes = g->comm ? g->comm->last_bytes : 0; return 0; }

// RCCL on one rank: n doubles travel to this rank itself through a grouped ncclSend / ncclRecv on the exchange stream
// (binding, communicator, stream and event plumbing on a single-GPU box); returns the largest difference
extern "C" int IGXCommLoopbackTest(IGX g, int64_t n, double *maxdiff) {
  NEEDIGA(g);
  if (!g->comm || g->comm->kind != 1) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGXCommInitRCCL() first");
  if (n < 1 || !maxdiff) return fail(IGX_ERR_ARG_WRONG, "bad arguments");
  IgxComm &c = *g->comm;
  DevBuf a, r; std::vector<double> h((size_t)n), back((size_t)n, -1.0);
  for (int64_t i = 0; i < n; ++i) h[(size_t)i] = 0.5 * (double)i - 3.0;
  if (a.upload(h) || r.alloc((size_t)n * 8)) return fail(IGX_ERR_MEM, "device allocation failed");
  HIPCK(hipMemset(r.p, 0, (size_t)n * 8));
  HIPCK(hipEventRecord(c.ready, g->stream));
  HIPCK(hipStreamWaitEvent(c.xs, c.ready, 0));
  NCCLCK(rccl_api().GroupStart());
  NCCLCK(rccl_api().Recv(r.p, (size_t)n, kNcclDouble, g->s.comm_rank, c.nccl, c.xs));
  NCCLCK(rccl_api().Send(a.p, (size_t)n, kNcclDouble, g->s.comm_rank, c.nccl, c.xs));
  NCCLCK(rccl_api().GroupEnd());
  HIPCK(hipEventRecord(c.done, c.xs));
  HIPCK(hipStreamWaitEvent(g->stream, c.done, 0));
  HIPCK(hipStreamSynchronize(g->stream));
  HIPCK(hipStreamSynchronize(c.xs));
  HIPCK(hipMemcpy(back.data(), r.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  double d = 0; for (int64_t i = 0; i < n; ++i) d = std::max(d, std::fabs(back[(size_t)i] - h[(size_t)i]));
  *maxdiff = d;
  return 0;
}
