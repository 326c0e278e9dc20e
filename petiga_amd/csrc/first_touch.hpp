// first_touch.hpp -- host helpers shared by the kernels that store the first contribution of a matrix entry instead of
// zeroing the matrix first (gram_mfma.hpp, feature_mfma.hpp launchers).
#pragma once
#include <algorithm>
#include <hip/hip_runtime.h>
#include "igx.hpp"

#include <string>

namespace igx {
// the dominant kernel of the last assembly, for bench.py's roofline block (filled by the pencil launcher)
struct DomInfo {
  std::string name = "none"; int launches = 0; long long elements = 0; double flop_per_element = 0;
  int passes = 1;      // face-first passes of the last pencil-walk assembly (1: one pass over the rank's box)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // recorded around those launches when timing is on
};

// zero the values of the rows of a box of row indices (rows near a face shared with another rank: they keep columns
// of nodes this rank holds no element of, which only the ghost-row exchange fills)
static __global__ void k_zero_row_box(int s0, int s1, int s2, int c0, int c1, int nr0, int nr1, const int64_t *browptr, double *val, int bs2) {
  const int64_t r = blockIdx.x;
  const int k0 = (int)(r % c0), k1 = (int)((r / c0) % c1), k2 = (int)(r / ((int64_t)c0 * c1));
  const int64_t row = (int64_t)(s0 + k0) + (int64_t)nr0 * ((int64_t)(s1 + k1) + (int64_t)nr1 * (s2 + k2));
  double *p = val + browptr[row] * bs2; const int64_t n = (browptr[row + 1] - browptr[row]) * bs2;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.0;
}

// rows with stencil columns outside the rank's element box: within p node layers of a face that has a neighbour rank
static void zero_neighbour_rows(const Space &s, const OutDev &out, hipStream_t stream) {
  for (int d = 0; d < s.dim; ++d) {
    const bool per = s.axis[d].periodic != 0;
    const bool lower = s.proc_sizes[d] > 1 && (s.proc_ranks[d] > 0 || per), upper = s.proc_sizes[d] > 1 && (s.proc_ranks[d] < s.proc_sizes[d] - 1 || per);
    const int nr = s.lay[d].nrow, p = s.axis[d].p;
    for (int side = 0; side < 2; ++side) {
      if (!(side ? upper : lower)) continue;
      int st[3] = {0, 0, 0}, ct[3] = {s.lay[0].nrow, s.lay[1].nrow, s.lay[2].nrow};
      ct[d] = std::min(p, nr); st[d] = side ? nr - ct[d] : 0;
      const int64_t rows = (int64_t)ct[0] * ct[1] * ct[2];
      if (rows > 0) hipLaunchKernelGGL(k_zero_row_box, dim3((unsigned)rows), dim3(256), 0, stream, st[0], st[1], st[2], ct[0], ct[1], s.lay[0].nrow, s.lay[1].nrow, out.browptr, out.val, s.dof * s.dof);
    }
  }
}

// true when every element pair on axis d follows the e mod (p+1) colouring with one new node layer per element
static bool axis_first_touch_ok(const Space &s, int d) {
  if (s.lay[d].alias) return false;
  for (int e = 0; e + 1 < s.elem_width[d]; ++e)
    if (s.basis[d].offset[s.elem_start[d] + e + 1] != s.basis[d].offset[s.elem_start[d] + e] + 1) return false;
  for (int e = 0; e < s.elem_width[d]; ++e) if (s.lay[d].color[e] != e % (s.axis[d].p + 1)) return false;
  return true;
}

// a periodic axis wrapped inside the rank: the rule holds without clipping when the colours run e mod (p+1) across the seam
static bool axis_first_touch_wrapped_ok(const Space &s, int d) {
  const int nb = s.axis[d].p + 1;
  if (!s.lay[d].alias || s.elem_width[d] % nb != 0 || s.elem_width[d] < 2 * nb || s.axis[d].nnp < 2 * s.axis[d].p + 1) return false;
  for (int e = 0; e + 1 < s.elem_width[d]; ++e)
    if (s.basis[d].offset[s.elem_start[d] + e + 1] != s.basis[d].offset[s.elem_start[d] + e] + 1) return false;
  for (int e = 0; e < s.elem_width[d]; ++e) if (s.lay[d].color[e] != e % nb) return false;
  return true;
}

}  // namespace igx
