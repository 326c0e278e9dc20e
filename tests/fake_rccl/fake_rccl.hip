// fake_rccl.hip -- a TEST DOUBLE for librccl.so's point-to-point API (test infrastructure, not product code).
//
// Why: the product transport of the ghost-row exchange (petiga_amd/csrc/comm.hpp, kind == 1) is grouped ncclSend / ncclRecv on a
// stream, in up to three stream-ordered phases with receives posted ahead of the local assembly.  A gpurun box has ONE GPU, and
// real RCCL refuses several ranks on one device, so that branch could only ever run as a one-rank loopback.  This library
// exports the same symbols and moves the data between PROCESSES THAT SHARE ONE GPU, keeping the semantics a schedule can
// depend on (and deadlock on):
//   * ncclSend / ncclRecv are ENQUEUED on the caller's stream: a group is ONE KERNEL on that stream (as in RCCL), so a send reads
//     its buffer only after the work that precedes it on the stream, everything the stream is given afterwards waits for the
//     whole group, and the host call returns at once -- the host takes no part in the transfer (no proxy thread: a host that
//     sits in a blocking HIP call cannot stall the exchange, exactly as with RCCL's intra-node kernels);
//   * the operations of one group progress CONCURRENTLY (one set of workgroups per operation); groups on one stream are ORDERED;
//   * a receive completes only when the matching send has been issued AND its stream has reached it; a send completes only when
//     its receive has taken the data (rendezvous -- the strictest behaviour RCCL may show, so a schedule that survives here does
//     not rely on eager buffering);
//   * messages between one (sender, receiver) pair match in issue order; a count mismatch is an error (ncclInvalidUsage);
//   * ncclCommInitRank is a blocking collective over all ranks of the id.
// FAKE_RCCL_BREAK=recv_first makes every group finish its receives before it starts its sends (what splitting a phase into a
// receive group followed by a send group on one stream would do): the tests use it to show that the double DOES hang on a
// schedule that real RCCL would hang on.  FAKE_RCCL_TIMEOUT_S (default 120): a watchdog thread (host memory only, no HIP calls)
// turns a group that stops making progress into a loud abort of the rank (exit code 86) that names the pending operations.
//
// Mechanics: a POSIX shared-memory control block per communicator id (one mailbox ring per ordered rank pair), mapped into the
// device's address space by every rank (hipHostRegister); one shared-memory segment per message, created by whichever side issues
// first and registered by both at ncclGroupEnd; the group kernel copies device -> segment, publishes `posted`, and waits for
// `taken` (send) / waits for `posted`, copies segment -> device, publishes `taken` (receive), all with system-scope atomics.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
constexpr int kRing = 16;                 // messages in flight per ordered pair
constexpr int kParts = 4;                 // workgroups per operation
constexpr int kCounters = 4096;           // ring of arrival counters (device memory)
constexpr uint64_t kMagic = 0x46414b4552434c32ull;   // "FAKERCL2"

struct Slot {                             // one message of a pair's ring; seq = index of the message on that pair (1-based)
  uint64_t posted;                        // seq once the sender's kernel has staged the data
  uint64_t taken;                         // seq once the receiver's kernel has copied it out
  uint64_t bytes;
  uint64_t pad;
};
struct Control {
  uint64_t magic;
  int nranks, arrived, departed, aborted;
  Slot slot[1];                           // [nranks * nranks * kRing], pair (src, dst) at (src * nranks + dst) * kRing
};

struct DevOp {                            // what the kernel reads (host-coherent memory, written before the launch)
  int send, counter;                      // counter: index into the arrival-counter ring
  char *buf;                              // device buffer
  char *seg;                              // the message segment, device view
  uint64_t bytes, seq;
  Slot *slot;                             // device view of the pair's slot
};
struct DevGroup { unsigned id, nops, counter, done; DevOp op[1]; };      // done: written by the kernel when the last operation has finished

struct Seg { void *host = nullptr; size_t bytes = 0; std::string name; bool unlink_after = false; };
struct OpRec { bool send; int peer; uint64_t seq; size_t bytes; };
struct GroupRec { unsigned id; std::vector<OpRec> ops; std::vector<Seg> segs; DevGroup *dev = nullptr; };

struct Comm {
  int nranks = 0, rank = 0, device = 0;
  std::string name;                       // shm name of the control block
  Control *ctl = nullptr, *ctl_dev = nullptr; size_t ctl_bytes = 0;
  std::vector<uint64_t> next_send, next_recv;     // per peer: messages issued so far
  unsigned *flags = nullptr;              // host-coherent: [0] last group whose kernel started, [2] error
  unsigned *counters = nullptr;           // device: arrival counters
  unsigned next_counter = 0;
  std::atomic<unsigned> issued{0};
  std::mutex mu; std::deque<GroupRec> live;       // groups not yet cleaned up (watchdog reads the op lists)
  std::thread watchdog; std::atomic<bool> stop{false};
  bool break_recv_first = false;
  double timeout_s = 120.0;
};

struct PendingOp { Comm *c; hipStream_t stream; bool send; void *buf; size_t bytes; int peer; uint64_t seq; };
thread_local int t_group_depth = 0;
thread_local std::vector<PendingOp> t_pending;

// every shared-memory name this process created or opened and has not yet unlinked: a rank that leaves through die() takes them
// along (several test processes may use the double at once: nobody may clean /dev/shm by prefix)
static std::mutex g_names_mu;
static std::vector<std::string> g_names;
static void remember(const std::string &n) { std::lock_guard<std::mutex> lk(g_names_mu); g_names.push_back(n); }
static void forget(const std::string &n) {
  std::lock_guard<std::mutex> lk(g_names_mu);
  for (size_t i = 0; i < g_names.size(); ++i) if (g_names[i] == n) { g_names.erase(g_names.begin() + (long)i); break; }
}

static void die(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt);
  fprintf(stderr, "fake_rccl: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); fflush(stderr);
  va_end(ap);
  for (const std::string &n : g_names) shm_unlink(n.c_str());      // (no lock: this thread is the last thing the process does)
  _exit(86);
}

__device__ inline uint64_t ld_sys(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void st_sys(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ void copy_part(char *dst, const char *src, uint64_t bytes, int part) {
  if ((bytes & 7) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 7) == 0) {
    const uint64_t n = bytes >> 3; uint64_t *d = reinterpret_cast<uint64_t *>(dst); const uint64_t *s = reinterpret_cast<const uint64_t *>(src);
    for (uint64_t i = (uint64_t)part * blockDim.x + threadIdx.x; i < n; i += (uint64_t)kParts * blockDim.x) d[i] = s[i];
  } else {
    for (uint64_t i = (uint64_t)part * blockDim.x + threadIdx.x; i < bytes; i += (uint64_t)kParts * blockDim.x) dst[i] = src[i];
  }
}

// one group: kParts workgroups per operation
__global__ void k_group(DevGroup *g, unsigned *flags, unsigned *counters, int *aborted) {
  const unsigned o = blockIdx.x / kParts; const int part = (int)(blockIdx.x % kParts);
  const DevOp op = g->op[o];
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&flags[0], g->id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ int last;
  if (op.send) {
    copy_part(op.seg, op.buf, op.bytes, part);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&counters[op.counter], 1u) == (unsigned)kParts - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) {
      counters[op.counter] = 0;
      st_sys(&op.slot->bytes, op.bytes);
      st_sys(&op.slot->posted, op.seq);
      while (ld_sys(&op.slot->taken) < op.seq) __builtin_amdgcn_s_sleep(64);      // rendezvous
    }
  } else {
    if (threadIdx.x == 0) {
      while (ld_sys(&op.slot->posted) < op.seq) __builtin_amdgcn_s_sleep(64);
      if (ld_sys(&op.slot->bytes) != op.bytes) {      // count mismatch: flag it and stay (the watchdog reports and ends the rank)
        __hip_atomic_store(&flags[2], 1u + o, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(aborted, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        while (true) __builtin_amdgcn_s_sleep(127);
      }
    }
    __syncthreads();
    copy_part(op.buf, op.seg, op.bytes, part);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&counters[op.counter], 1u) == (unsigned)kParts - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) { counters[op.counter] = 0; st_sys(&op.slot->taken, op.seq); }
  }
  if (threadIdx.x == 0) {      // the last operation of the group to finish counts the group as finished
    if (atomicAdd(&counters[g->counter], 1u) == g->nops - 1) {
      counters[g->counter] = 0;
      __hip_atomic_store(&g->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static size_t slot_index(const Comm *c, int src, int dst, uint64_t seq) { return ((size_t)src * c->nranks + dst) * kRing + (seq % kRing); }
static std::string msg_name(Comm *c, int src, int dst, uint64_t seq) { return c->name + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq); }

// create-or-open the segment of a message (whichever side issues first creates it); nullptr + rc on failure
static int open_segment(Comm *c, const std::string &nm, size_t bytes, Seg &out) {
  int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd >= 0) { if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return kSystemError; } }
  else {
    fd = shm_open(nm.c_str(), O_RDWR, 0600);
    if (fd < 0) return kSystemError;
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (true) {      // the creator sizes it right after creating it
      if (fstat(fd, &st) != 0) { close(fd); return kSystemError; }
      if (st.st_size > 0) break;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) { close(fd); return kSystemError; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if ((size_t)st.st_size != bytes) {
      fprintf(stderr, "fake_rccl: rank %d: message %s is %zu bytes on this side and %lld on the other (count mismatch)\n", c->rank, nm.c_str(), bytes, (long long)st.st_size);
      close(fd); return kInvalidUsage;
    }
  }
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return kSystemError;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { munmap(p, bytes); return kUnhandledCudaError; }
  out.host = p; out.bytes = bytes; out.name = nm;
  remember(nm);
  return kSuccess;
}

static void release(GroupRec &g) {
  for (Seg &s : g.segs) {
    if (s.host) { (void)hipHostUnregister(s.host); munmap(s.host, s.bytes); }
    if (s.unlink_after) shm_unlink(s.name.c_str());
    forget(s.name);
  }
  g.segs.clear();
  if (g.dev) { (void)hipHostFree(g.dev); g.dev = nullptr; }
}

static bool group_done(const GroupRec &g) { return g.dev && __atomic_load_n(&g.dev->done, __ATOMIC_ACQUIRE) != 0; }

// finished groups give their segments back (main thread only)
static void reap(Comm *c, bool all) {
  std::vector<GroupRec> gone;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto it = c->live.begin(); it != c->live.end();) {
      if (all || group_done(*it)) { gone.push_back(std::move(*it)); it = c->live.erase(it); } else ++it;
    }
  }
  for (GroupRec &g : gone) release(g);
}

static std::string describe(Comm *c, const GroupRec &g) {
  std::string w;
  for (const OpRec &op : g.ops) {
    const Slot &s = c->ctl->slot[op.send ? slot_index(c, c->rank, op.peer, op.seq) : slot_index(c, op.peer, c->rank, op.seq)];
    const uint64_t posted = __atomic_load_n(&s.posted, __ATOMIC_ACQUIRE), taken = __atomic_load_n(&s.taken, __ATOMIC_ACQUIRE);
    if (op.send && taken >= op.seq) continue;
    if (!op.send && taken >= op.seq) continue;
    w += std::string(op.send ? (posted >= op.seq ? " send(staged)->" : " send->") : (posted >= op.seq ? " recv(posted)<-" : " recv<-")) + std::to_string(op.peer) + "#" + std::to_string(op.seq);
  }
  return w.empty() ? " (none)" : w;
}

static void watchdog_main(Comm *c) {
  using clk = std::chrono::steady_clock;
  unsigned seen_started = 0, seen_issued = 0; size_t seen_open = 0;
  auto t_change = clk::now();
  while (!c->stop) {
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    const unsigned started = __atomic_load_n(&c->flags[0], __ATOMIC_ACQUIRE), issued = c->issued.load();
    const unsigned err = __atomic_load_n(&c->flags[2], __ATOMIC_ACQUIRE);
    size_t open = 0;
    std::string pending;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      for (const GroupRec &g : c->live) if (!group_done(g)) { ++open; pending += " [group " + std::to_string(g.id) + ":" + describe(c, g) + "]"; }
    }
    if (pending.empty()) pending = " nothing";
    if (err) die("rank %d: a receive's count does not match its send (operation %u of the running group); pending:%s", c->rank, err - 1, pending.c_str());
    if (__atomic_load_n(&c->ctl->aborted, __ATOMIC_ACQUIRE))
      die("rank %d: another rank aborted; here %u groups issued, the stream has started group %u, %zu unfinished; pending:%s", c->rank, issued, started, open, pending.c_str());
    if (started != seen_started || open != seen_open || issued != seen_issued) { seen_started = started; seen_open = open; seen_issued = issued; t_change = clk::now(); continue; }
    if (open == 0) { t_change = clk::now(); continue; }      // idle
    if (std::chrono::duration<double>(clk::now() - t_change).count() > c->timeout_s) {
      __atomic_store_n(&c->ctl->aborted, 1, __ATOMIC_RELEASE);
      die("rank %d: no progress for %.0f s: %u groups issued, the stream has started group %u, %zu unfinished; pending:%s  -- DEADLOCK in the schedule",
          c->rank, c->timeout_s, issued, started, open, pending.c_str());
    }
  }
}

static int launch_group(Comm *c, hipStream_t stream, const std::vector<PendingOp> &ops) {
  if (ops.empty()) return kSuccess;
  GroupRec rec; rec.id = c->issued.load() + 1;
  const size_t dbytes = sizeof(DevGroup) + sizeof(DevOp) * ops.size();
  if (hipHostMalloc(reinterpret_cast<void **>(&rec.dev), dbytes, hipHostMallocCoherent) != hipSuccess) return kUnhandledCudaError;
  rec.dev->id = rec.id; rec.dev->nops = (unsigned)ops.size(); rec.dev->counter = c->next_counter++ % kCounters; rec.dev->done = 0;
  int rc = kSuccess;
  for (size_t i = 0; i < ops.size() && rc == kSuccess; ++i) {
    const PendingOp &p = ops[i];
    DevOp &d = rec.dev->op[i];
    d.send = p.send ? 1 : 0; d.counter = (int)(c->next_counter++ % kCounters); d.buf = static_cast<char *>(p.buf); d.bytes = p.bytes; d.seq = p.seq; d.seg = nullptr;
    const int src = p.send ? c->rank : p.peer, dst = p.send ? p.peer : c->rank;
    d.slot = &c->ctl_dev->slot[slot_index(c, src, dst, p.seq)];
    if (p.bytes) {
      Seg s;
      rc = open_segment(c, msg_name(c, src, dst, p.seq), p.bytes, s);
      if (rc != kSuccess) break;
      s.unlink_after = !p.send;           // the receiver removes the name once its group has finished (the sender has it open by then)
      void *dp = nullptr;
      if (hipHostGetDevicePointer(&dp, s.host, 0) != hipSuccess) { rec.segs.push_back(s); rc = kUnhandledCudaError; break; }
      d.seg = static_cast<char *>(dp);
      rec.segs.push_back(s);
    }
    rec.ops.push_back({p.send, p.peer, p.seq, p.bytes});
  }
  if (rc != kSuccess) {      // (a count mismatch seen here: the peers that wait for this rank must not wait for ever)
    if (rc == kInvalidUsage) __atomic_store_n(&c->ctl->aborted, 1, __ATOMIC_RELEASE);
    release(rec); return rc;
  }
  DevGroup *dev = rec.dev;
  const unsigned nops = (unsigned)ops.size();
  { std::lock_guard<std::mutex> lk(c->mu); c->live.push_back(std::move(rec)); }
  c->issued.fetch_add(1);
  hipLaunchKernelGGL(k_group, dim3(nops * kParts), dim3(256), 0, stream, dev, c->flags, c->counters, &c->ctl_dev->aborted);
  return hipGetLastError() == hipSuccess ? kSuccess : kUnhandledCudaError;
}

static int enqueue_group(Comm *c, hipStream_t stream, std::vector<PendingOp> &&ops) {
  reap(c, false);
  if (!c->break_recv_first) return launch_group(c, stream, ops);
  std::vector<PendingOp> r, s;
  for (const PendingOp &p : ops) (p.send ? s : r).push_back(p);
  if (int rc = launch_group(c, stream, r)) return rc;
  return launch_group(c, stream, s);
}

static size_t type_bytes(int datatype) {
  switch (datatype) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}

static int p2p(bool send, void *buf, size_t count, int datatype, int peer, Comm *c, hipStream_t stream) {
  if (!c || !c->ctl) return kInvalidArgument;
  if (peer < 0 || peer >= c->nranks) return kInvalidArgument;
  const size_t tb = type_bytes(datatype);
  if (!tb) return kInvalidArgument;
  if (count && !buf) return kInvalidArgument;
  PendingOp op{c, stream, send, buf, count * tb, peer, send ? ++c->next_send[peer] : ++c->next_recv[peer]};
  if (t_group_depth > 0) { t_pending.push_back(op); return kSuccess; }
  std::vector<PendingOp> one{op};
  return enqueue_group(c, stream, std::move(one));
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;
typedef Comm *ncclComm_t;

int ncclGetVersion(int *v) { if (v) *v = 22606; return kSuccess; }
const char *ncclGetErrorString(int r) {
  switch (r) { case 0: return "no error"; case 1: return "unhandled cuda error"; case 2: return "unhandled system error"; case 3: return "internal error";
    case 4: return "invalid argument"; case 5: return "invalid usage"; default: return "unknown result code"; }
}
const char *ncclGetLastError(ncclComm_t) { return ""; }

int ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return kInvalidArgument;
  memset(id, 0, sizeof *id);
  unsigned long long r = 0;
  FILE *f = fopen("/dev/urandom", "rb");
  if (f) { if (fread(&r, sizeof r, 1, f) != 1) r = 0; fclose(f); }
  if (!r) r = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 32);
  snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%016llx", (int)getpid(), r);
  return kSuccess;
}

int ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
  id.internal[sizeof id.internal - 1] = 0;
  if (strncmp(id.internal, "/fake_rccl_", 11) != 0) return kInvalidArgument;
  std::unique_ptr<Comm> c(new Comm());
  c->nranks = nranks; c->rank = rank; c->name = id.internal;
  if (hipGetDevice(&c->device) != hipSuccess) return kUnhandledCudaError;
  if (const char *e = getenv("FAKE_RCCL_BREAK")) c->break_recv_first = strcmp(e, "recv_first") == 0;
  if (const char *e = getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_s = atof(e) > 0 ? atof(e) : c->timeout_s;
  c->ctl_bytes = sizeof(Control) + sizeof(Slot) * (size_t)nranks * nranks * kRing;
  // whoever comes first creates and sizes the control block; the others wait for its magic
  int fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  const bool creator = fd >= 0;
  const auto t0 = std::chrono::steady_clock::now();
  auto late = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s; };
  if (creator) { if (ftruncate(fd, (off_t)c->ctl_bytes) != 0) { close(fd); shm_unlink(c->name.c_str()); return kSystemError; } }
  else {
    while (true) {
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->ctl_bytes) break;
      if (fd >= 0) close(fd);
      if (late()) return kSystemError;
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  void *p = mmap(nullptr, c->ctl_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return kSystemError;
  c->ctl = static_cast<Control *>(p);       // (a fresh segment is zero-filled: every counter starts at 0)
  remember(c->name);
  if (creator) { c->ctl->nranks = nranks; __atomic_store_n(&c->ctl->magic, kMagic, __ATOMIC_RELEASE); }
  while (__atomic_load_n(&c->ctl->magic, __ATOMIC_ACQUIRE) != kMagic) { if (late()) return kSystemError; std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
  if (c->ctl->nranks != nranks) { fprintf(stderr, "fake_rccl: rank %d says %d ranks, the communicator has %d\n", rank, nranks, c->ctl->nranks); return kInvalidUsage; }
  if (hipHostRegister(c->ctl, c->ctl_bytes, hipHostRegisterDefault) != hipSuccess) return kUnhandledCudaError;
  void *dp = nullptr;
  if (hipHostGetDevicePointer(&dp, c->ctl, 0) != hipSuccess) return kUnhandledCudaError;
  c->ctl_dev = static_cast<Control *>(dp);
  if (hipHostMalloc(reinterpret_cast<void **>(&c->flags), 64, hipHostMallocCoherent) != hipSuccess) return kUnhandledCudaError;
  memset(c->flags, 0, 64);
  if (hipMalloc(reinterpret_cast<void **>(&c->counters), kCounters * sizeof(unsigned)) != hipSuccess) return kUnhandledCudaError;
  if (hipMemset(c->counters, 0, kCounters * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return kUnhandledCudaError;
  c->next_send.assign((size_t)nranks, 0); c->next_recv.assign((size_t)nranks, 0);
  // blocking collective: nobody returns before everybody is here
  __atomic_fetch_add(&c->ctl->arrived, 1, __ATOMIC_ACQ_REL);
  while (__atomic_load_n(&c->ctl->arrived, __ATOMIC_ACQUIRE) < nranks) {
    if (late()) { fprintf(stderr, "fake_rccl: rank %d: only %d of %d ranks reached ncclCommInitRank\n", rank, c->ctl->arrived, nranks); return kSystemError; }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  Comm *raw = c.release();
  raw->watchdog = std::thread(watchdog_main, raw);
  *comm = raw;
  return kSuccess;
}

int ncclCommCount(ncclComm_t c, int *n) { if (!c || !n) return kInvalidArgument; *n = c->nranks; return kSuccess; }
int ncclCommUserRank(ncclComm_t c, int *r) { if (!c || !r) return kInvalidArgument; *r = c->rank; return kSuccess; }
int ncclCommCuDevice(ncclComm_t c, int *d) { if (!c || !d) return kInvalidArgument; *d = c->device; return kSuccess; }

// (the caller has synchronised the streams it used, as ncclCommDestroy expects)
int ncclCommDestroy(ncclComm_t c) {
  if (!c) return kInvalidArgument;
  c->stop = true;
  if (c->watchdog.joinable()) c->watchdog.join();
  reap(c, true);
  if (c->counters) (void)hipFree(c->counters);
  if (c->flags) (void)hipHostFree(c->flags);
  if (c->ctl) {
    const bool last = __atomic_fetch_add(&c->ctl->departed, 1, __ATOMIC_ACQ_REL) + 1 == c->nranks;
    (void)hipHostUnregister(c->ctl);
    munmap(c->ctl, c->ctl_bytes);
    if (last) shm_unlink(c->name.c_str());
    forget(c->name);
  }
  delete c;
  return kSuccess;
}
int ncclCommAbort(ncclComm_t c) { if (c && c->ctl) __atomic_store_n(&c->ctl->aborted, 1, __ATOMIC_RELEASE); return ncclCommDestroy(c); }

int ncclGroupStart() { ++t_group_depth; return kSuccess; }
int ncclGroupEnd() {
  if (t_group_depth <= 0) return kInvalidUsage;
  if (--t_group_depth > 0) return kSuccess;
  // one group per (communicator, stream) that took part, in first-use order
  int rc = kSuccess;
  while (!t_pending.empty()) {
    Comm *c = t_pending.front().c; hipStream_t st = t_pending.front().stream;
    std::vector<PendingOp> ops;
    for (size_t i = 0; i < t_pending.size();) {
      if (t_pending[i].c == c && t_pending[i].stream == st) { ops.push_back(t_pending[i]); t_pending.erase(t_pending.begin() + (long)i); }
      else ++i;
    }
    const int r = enqueue_group(c, st, std::move(ops));
    if (r != kSuccess) rc = r;
  }
  return rc;
}

int ncclSend(const void *buf, size_t count, int datatype, int peer, ncclComm_t c, hipStream_t stream) { return p2p(true, const_cast<void *>(buf), count, datatype, peer, c, stream); }
int ncclRecv(void *buf, size_t count, int datatype, int peer, ncclComm_t c, hipStream_t stream) { return p2p(false, buf, count, datatype, peer, c, stream); }

}  // extern "C"
