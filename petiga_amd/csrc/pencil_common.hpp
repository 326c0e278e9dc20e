// pencil_common.hpp -- what the pencil kernels share (gram_mfma.hpp: scalar gradient-Gram forms, sliding window;
// block_pencil.hpp: constant-coefficient multi-field forms, band rows by node layer): vector types, the first-touch rule,
// colour ranges of a box of elements, the walkability test of an axis.
// IGX_RTC: the device half is also compiled at run time (rtc.hpp); hiprtc has neither the standard library nor files to include.
#pragma once
#ifndef IGX_RTC
#include <hip/hip_runtime.h>
#include <mutex>
#include "first_touch.hpp"
#endif
#include "igx.hpp"

namespace igx {

typedef double d4_t __attribute__((ext_vector_type(4)));

typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));

// First touch.  An entry (row slot a, column slot b of element e on one axis) receives contributions from the elements
// [e + max(a,b) - P, e + min(a,b)] (clipped to the rank's elements); colours are e mod (P+1) and launch in ascending
// order, so the first launch to reach the entry is colour 0 if the range holds a multiple of P+1, else the colour of
// its lowest element.  Along the walk axis a pencil combines everything in registers: one write per entry and pencil.
// Two passes (the elements [blocked, nel) of the axis were assembled by earlier launches, this pass covers [rlo, rhi)): an entry
// that the earlier pass reaches as well is never a first touch here, and the rule applies to the elements of this pass alone.
template <int P>
__device__ __forceinline__ bool first_touch_axis(int e, int a, int b, int nel, int rlo = 0, int rhi = 0x7fffffff, int blocked = 0x7fffffff) {
  constexpr int NB = P + 1;
  int lo = e + (a > b ? a : b) - P, hi = e + (a < b ? a : b);
  if (hi > nel - 1) hi = nel - 1;
  if (hi >= blocked) return false;
  if (lo < rlo) lo = rlo;
  if (hi > rhi - 1) hi = rhi - 1;
  const int c0 = ((lo + NB - 1) / NB) * NB;       // smallest multiple of NB >= lo
  return (c0 <= hi) ? (e % NB == 0) : (e == lo);
}

// The same rule on a periodic axis wrapped inside the rank whose element count is a multiple of P+1 (colours stay e mod (P+1) across
// the seam): the range is not clipped, its elements are taken modulo the axis.
template <int P>
__device__ __forceinline__ bool first_touch_axis_wrapped(int e, int a, int b) {
  constexpr int NB = P + 1;
  const int lo = e + (a > b ? a : b) - P, hi = e + (a < b ? a : b);
  const int c0 = ((lo + NB + NB - 1) / NB) * NB - NB;       // smallest multiple of NB >= lo (lo >= -P)
  return (c0 <= hi) ? (e % NB == 0) : (e == lo);
}

#ifndef IGX_RTC
struct Box { int lo[3], hi[3]; };   // local element box [lo,hi)

// Where a face-first pass cuts an axis of n elements: the elements [cut, n) are assembled first.  The p elements next to the face
// would do, but a pass that thin fills a quarter of the CUs with one short segment per workgroup (128^3 per rank: +5 ms per face
// on a 33 ms assembly, scripts/time_rank_box.py); the upper HALF of the axis costs only its segments' halo and still leaves the
// face's messages half (a quarter, an eighth) of the assembly to travel.  A multiple of p+1, at least p+1 from both ends.
static inline int face_cut(int n, int p) { const int c = (n / 2) / (p + 1) * (p + 1); return std::max(p + 1, std::min(c, n - (p + 1))); }

// Launch-scoped device arrays (the point records of band_pt, the boundary-load sums of the pencil kernels) live in a
// stream-ordered pool of the library's own: what one launch frees is what the next one takes, and nothing goes back to the
// driver at a synchronisation point.  (The device's default pool has a release threshold of 0: after every host synchronisation
// the next assembly paid a fresh device allocation for each of its launches, on the host, in the middle of its launch sequence.)
// One pool per device (inline: the translation units of the library share the table); the pool of the device that is current at
// the call is the one used, so an IGX created after the application switched devices allocates where its stream lives.  The pool
// keeps what it was given while any IGX is alive (release threshold: everything) and hands it back to the driver when the last
// one is destroyed (igx_pool_release) or when an allocation fails (trim, then one more try) -- torch / PETSc share the device.
constexpr int IGX_MAX_DEVICES = 64;
struct IgxPools { std::mutex mu; hipMemPool_t pool[IGX_MAX_DEVICES] = {}; bool tried[IGX_MAX_DEVICES] = {}; int live = 0; };
inline IgxPools &igx_pools() { static IgxPools p; return p; }
inline hipMemPool_t igx_pool() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IGX_MAX_DEVICES) return nullptr;
  IgxPools &t = igx_pools();
  std::lock_guard<std::mutex> lock(t.mu);
  if (!t.tried[dev]) {
    t.tried[dev] = true;
    hipMemPool_t p = nullptr;
    hipMemPoolProps props; memset(&props, 0, sizeof(props));
    props.allocType = hipMemAllocationTypePinned; props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice; props.location.id = dev;
    if (hipMemPoolCreate(&p, &props) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    if (p) { uint64_t keep = ~0ull; (void)hipMemPoolSetAttribute(p, hipMemPoolAttrReleaseThreshold, &keep); }
    t.pool[dev] = p;
  }
  return t.pool[dev];
}
inline void igx_pool_trim() {            // every device's pool: give unused memory back to the driver
  IgxPools &t = igx_pools();
  std::lock_guard<std::mutex> lock(t.mu);
  for (hipMemPool_t p : t.pool) if (p) (void)hipMemPoolTrimTo(p, 0);
}
inline void igx_pool_acquire() { IgxPools &t = igx_pools(); std::lock_guard<std::mutex> lock(t.mu); ++t.live; }
inline void igx_pool_release() {
  bool last;
  { IgxPools &t = igx_pools(); std::lock_guard<std::mutex> lock(t.mu); last = --t.live <= 0; if (last) t.live = 0; }
  if (last) igx_pool_trim();
}
inline hipError_t pool_alloc(void **ptr, size_t bytes, hipStream_t stream) {
  hipMemPool_t p = igx_pool();
  if (!p) return hipMallocAsync(ptr, bytes, stream);
  hipError_t rc = hipMallocFromPoolAsync(ptr, bytes, p, stream);
  if (rc != hipSuccess) {                // the pool may sit on memory that other launches freed in other sizes
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream);
    igx_pool_trim();
    rc = hipMallocFromPoolAsync(ptr, bytes, p, stream);
  }
  return rc;
}

// colour c of axis d restricted to [lo,hi): arithmetic sequence (regular colours) or a single element
static bool color_range(const AxisLayout &L, int c, int lo, int hi, int &start, int &step, int &count) {
  start = -1; count = 0; step = L.p + 1;
  for (int e = lo; e < hi; ++e) if (L.color[e] == c) { if (start < 0) start = e; count++; }
  return count > 0;
}

#endif   // !IGX_RTC

// boundary loads on the identity geometry: see k_boundary_loads (gram_mfma.hpp)
struct FluxArgs {
  int d, t, u;                 // face axis and the two axes of the face
  int rd;                      // row index of the face nodes on axis d
  int nt, nu;                  // face nodes (rank-local rows) on axes t, u
  const double *st, *su;       // [nt], [nu]: sum over the rank's elements holding the node of J / nen
  double value;                // load * 4
  int gfirst[3], glast[3];     // global node index of row 0 on every axis; last global node index (nnp - 1)
  int fixlo[3], fixhi[3];      // a Dirichlet value holds field 0 on the lower / upper face of the axis
};

#ifndef IGX_RTC
// one new node layer per element; a periodic axis wrapped inside the rank only where the walk takes it modulo its length (wrap_ok:
// the axis-0 walk of gram_mfma.hpp)
static bool axis_walkable(const Space &s, int d, bool wrap_ok = false) {
  if ((s.lay[d].alias && !wrap_ok) || s.elem_width[d] < 8) return false;
  for (int e = 0; e + 1 < s.elem_width[d]; ++e)
    if (s.basis[d].offset[s.elem_start[d] + e + 1] != s.basis[d].offset[s.elem_start[d] + e] + 1) return false;
  return true;
}

#endif   // !IGX_RTC

}  // namespace igx
