// host.cpp -- O(N) discretisation set-up that stays on the host: knot vectors, Gauss rules, 1-D basis
// tables, processor grid, element/node ranges and the per-axis index layout of the device matrix.
// Restates L0 of the reference (petigaaxis.c, petigarule.c, petigabasis.c, petigapart.c,
// petiga.c:1111-1310, petigamat.c:197-267); results are bit-exact for every integer quantity.
#include "igx.hpp"
#include <algorithm>
#include <climits>
#include <cmath>
#include <string>
#include <cstdlib>

namespace igx {

EnvSwitches read_env_switches() {
  EnvSwitches e;
  auto num = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
  e.kernel = num("IGX_KERNEL", 0); if (e.kernel < 0 || e.kernel > 4) e.kernel = 0;
  e.walk_axis = num("IGX_WALK_AXIS", 0);
  e.nseg = num("IGX_NSEG", 0);
  e.no_first_touch = num("IGX_NO_FIRST_TOUCH", 0) != 0;
  e.block_pencil = num("IGX_BLOCK_PENCIL", 1);
  e.vec_sumfact = num("IGX_VEC_SUMFACT", 1);
  e.state_pencil = num("IGX_STATE_PENCIL", 1);
  e.no_vec_pairs = num("IGX_NO_VEC_PAIRS", 0);
  e.fuse_groups = num("IGX_FUSE_GROUPS", 1);
  e.overlap = num("IGX_OVERLAP", -1);
  e.clock_probe = num("IGX_CLOCK_PROBE", 0) != 0;
  e.feature_lds_kb = num("IGX_FEATURE_LDS_KB", 0);
  e.combine = num("IGX_COMBINE", -1);
  e.free_run = num("IGX_FREE_RUN", -1);
  e.p2_pack = num("IGX_P2_PACK", 1);
  e.fuse_resid = num("IGX_FUSE_RESID", 0);
  e.patch = num("IGX_PATCH", 1);
  e.patch_state = num("IGX_PATCH_STATE", 0);
  e.small_wpb = num("IGX_SMALL_WPB", 1);
  e.band_prio = num("IGX_BAND_PRIO", 0); e.band_rmw_prio = num("IGX_BAND_RMW_PRIO", 0);
  if (kDebug) { e.debug_feature = num("IGX_DEBUG_FEATURE", 0); e.debug_noflush = num("IGX_DEBUG_NOFLUSH", 0); e.debug_timing = getenv("IGX_DEBUG_TIMING") != nullptr; }
  return e;
}

// ------------------------------------------------------------------ quadrature rules
// The reference tabulates Gauss-Legendre q = 1..10 (src/petigarule.c:182-319) and Gauss-Lobatto q = 2..10 (:321-459) as
// 36-digit constants.  Here the nodes are computed: Newton's method on P_q (Legendre) or P'_{q-1} (Lobatto) in long double
// from a cosine guess, polished in quadruple precision so that the value cast to double is the correctly rounded one -- the
// double the reference's literal denotes (tests/test_golden.py holds both rules to tests/golden/gauss_rules.json, bit for bit).
namespace {
using quad = __float128;
struct Leg { quad P, dP, ddP; };        // P_n(x), P_n'(x), P_n''(x)
Leg legendre(int n, quad x) {
  quad a = 1, b = x;
  if (n == 0) return {1, 0, 0};
  for (int k = 2; k <= n; ++k) { quad c = ((2 * k - 1) * x * b - (k - 1) * a) / k; a = b; b = c; }
  const quad dP = n * (x * b - a) / (x * x - 1);
  return {b, dP, (2 * x * dP - n * (n + 1) * b) / (1 - x * x)};      // Legendre's equation gives the second derivative
}
template <bool LOBATTO> quad rule_root(int n, long double guess) {
  long double x = guess;
  for (int it = 0; it < 64; ++it) {
    const Leg l = legendre(n, (quad)x);
    const long double dx = LOBATTO ? (long double)(l.dP / l.ddP) : (long double)(l.P / l.dP);
    x -= dx;
    if (fabsl(dx) < 1e-19L) break;
  }
  quad xq = x;
  for (int it = 0; it < 3; ++it) { const Leg l = legendre(n, xq); xq -= LOBATTO ? l.dP / l.ddP : l.P / l.dP; }
  return xq;
}
const long double kPi = 3.14159265358979323846264338327950288L;
}  // namespace

int gauss_legendre(int q, double *X, double *W) {
  if (q < 1 || q > 10) return IGX_ERR_ARG_OUTOFRANGE;
  for (int i = 0; i < (q + 1) / 2; ++i) {
    const quad x = rule_root<false>(q, cosl(kPi * (i + 0.75L) / (q + 0.5L)));
    const Leg l = legendre(q, x);
    const quad w = 2 / ((1 - x * x) * l.dP * l.dP);
    X[i] = -(double)x; X[q - 1 - i] = (double)x;
    W[i] = (double)w;  W[q - 1 - i] = (double)w;
  }
  if (q & 1) X[q / 2] = 0.0;
  return 0;
}

// Gauss-Lobatto: the end points and the roots of P'_{q-1}; weights 2 / (q (q-1) P_{q-1}(x)^2).
int gauss_lobatto(int q, double *X, double *W) {
  if (q < 2 || q > 10) return IGX_ERR_ARG_OUTOFRANGE;
  const int n = q - 1;
  for (int i = 0; i < (q + 1) / 2; ++i) {
    const quad x = i == 0 ? (quad)1 : rule_root<true>(n, cosl(kPi * i / n));
    const Leg l = legendre(n, x);
    const quad w = 2 / ((quad)(q * n) * l.P * l.P);
    X[i] = -(double)x; X[q - 1 - i] = (double)x;
    W[i] = (double)w;  W[q - 1 - i] = (double)w;
  }
  if (q & 1) X[q / 2] = 0.0;
  return 0;
}

// IGARuleSetUp (src/petigarule.c:116-143): the rule of an axis by type.
int rule_setup(const Rule1D &r, int nqp, double *X, double *W, std::string &err) {
  int rc = 0;
  switch (r.type) {
    case IGX_RULE_LEGENDRE: rc = gauss_legendre(nqp, X, W); break;
    case IGX_RULE_LOBATTO:  rc = gauss_lobatto(nqp, X, W); break;
    case IGX_RULE_USER:
      if ((int)r.x.size() != nqp) { err = "user-defined rule has a different number of points"; return IGX_ERR_ARG_WRONGSTATE; }
      for (int q = 0; q < nqp; ++q) { X[q] = r.x[q]; W[q] = r.w[q]; }
      break;
    default:   // IGA_RULE_REDUCED: Gauss-Legendre; basis_init gives the interior elements one point less (src/petigabasis.c:144-171)
      rc = gauss_legendre(nqp, X, W); break;
  }
  if (rc) { err = "Number of quadrature points not implemented"; return IGX_ERR_ARG_OUTOFRANGE; }
  return 0;
}

// ------------------------------------------------------------------ B-spline basis and derivatives
// Piegl & Tiller A2.3 (the algorithm of src/petigabsb.f90.in:3-63); output [a][0..4] like
// src/petigabsp.F90:3-16 with unused derivative slots zero.
void bspline_ders(int span, double u, int p, int nders, const double *U, double *out) {
  constexpr int MP = 8;
  double ndu[MP][MP], left[MP], right[MP], a[2][MP];
  ndu[0][0] = 1.0;
  for (int j = 1; j <= p; ++j) {
    left[j] = u - U[span + 1 - j];
    right[j] = U[span + j] - u;
    double saved = 0.0;
    for (int r = 0; r < j; ++r) {
      ndu[j][r] = right[r + 1] + left[j - r];
      double tmp = ndu[r][j - 1] / ndu[j][r];
      ndu[r][j] = saved + right[r + 1] * tmp;
      saved = left[j - r] * tmp;
    }
    ndu[j][j] = saved;
  }
  for (int j = 0; j <= p; ++j) {
    for (int k = 0; k < 5; ++k) out[j * 5 + k] = 0.0;
    out[j * 5] = ndu[j][p];
  }
  for (int r = 0; r <= p; ++r) {
    int s1 = 0, s2 = 1;
    a[0][0] = 1.0;
    for (int k = 1; k <= nders; ++k) {
      double d = 0.0;
      const int rk = r - k, pk = p - k;
      if (r >= k) { a[s2][0] = a[s1][0] / ndu[pk + 1][rk]; d = a[s2][0] * ndu[rk][pk]; }
      const int j1 = (rk >= -1) ? 1 : -rk;
      const int j2 = (r - 1 <= pk) ? k - 1 : p - r;
      for (int j = j1; j <= j2; ++j) {
        a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j];
        d += a[s2][j] * ndu[rk + j][pk];
      }
      if (r <= pk) { a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r]; d += a[s2][k] * ndu[r][pk]; }
      out[r * 5 + k] = d;
      std::swap(s1, s2);
    }
  }
  int fac = p;
  for (int k = 1; k <= nders; ++k) {
    for (int j = 0; j <= p; ++j) out[j * 5 + k] *= (double)fac;
    fac *= (p - k);
  }
}

// ------------------------------------------------------------------ knot vectors
static int next_knot(int m, const double *U, int k, int dir) {   // src/petigaaxis.c:482-494
  if (dir >= 0) {
    if (k < 0) return 0;
    for (int j = k + 1; j < m; ++j) if (U[j] > U[k]) return j;
    return m;
  }
  if (k > m) return m;
  for (int j = k - 1; j > 0; --j) if (U[j] < U[k]) return j;
  return 0;
}

void axis_finish(Axis &ax) {   // src/petigaaxis.c:286-312, :513-523
  const int p = ax.p, m = ax.m, n = m - p - 1;
  ax.span.clear();
  for (int k = next_knot(m, ax.U.data(), p, 1); k <= n + 1; k = next_knot(m, ax.U.data(), k, 1)) ax.span.push_back(k - 1);
  ax.nel = (int)ax.span.size();
  if (ax.periodic) {
    const int k = n + 1, j = next_knot(m, ax.U.data(), k, 1), s = j - k, C = p - s;
    ax.nnp = n - C;
  } else ax.nnp = n + 1;
}

int axis_init_uniform(Axis &ax, int N, double Ui, double Uf, int C, std::string &err) {   // src/petigaaxis.c:401-456
  const int p = ax.p;
  if (p < 1) { err = "Must call IGAAxisSetDegree() first"; return IGX_ERR_ORDER; }
  if (C == IGX_DECIDE) C = p - 1;
  if (N < 1) { err = "Number of elements must be greater than zero"; return IGX_ERR_ARG_WRONG; }
  if (Ui >= Uf) { err = "Initial value must be less than final value"; return IGX_ERR_ARG_WRONG; }
  if (C < 0 || C >= p) { err = "Continuity must be in range [0,p-1]"; return IGX_ERR_ARG_WRONG; }
  const int s = p - C, m = 2 * (p + 1) + (N - 1) * s - 1, n = m - p - 1;
  ax.m = m;
  ax.U.assign((size_t)m + 1, 0.0);
  int k = 0;
  for (; k <= p; ++k) { ax.U[k] = Ui; ax.U[m - k] = Uf; }
  for (int b = 1; b < N; ++b)
    for (int j = 0; j < s; ++j) ax.U[k++] = Ui + (double)b / (double)N * (Uf - Ui);
  if (ax.periodic)
    for (k = 0; k <= C; ++k) {
      ax.U[C - k] = ax.U[p] - ax.U[m - p] + ax.U[n - k];
      ax.U[m - C + k] = ax.U[m - p] - ax.U[p] + ax.U[p + 1 + k];
    }
  ax.nel = N;
  ax.span.resize(N);
  for (int e = 0; e < N; ++e) ax.span[e] = p + e * s;
  ax.nnp = ax.periodic ? n - C : n + 1;
  return 0;
}

int axis_set_knots(Axis &ax, int m, const double *U, std::string &err) {   // src/petigaaxis.c:202-253
  const int p = ax.p;
  if (p < 1) { err = "Must call IGAAxisSetDegree() first"; return IGX_ERR_ORDER; }
  if (m < 2 * p + 1) { err = "Number of knots must be at least 2*(p+1)"; return IGX_ERR_ARG_OUTOFRANGE; }
  for (int k = 1; k <= m; ++k) if (U[k - 1] > U[k]) { err = "Knot sequence must be increasing"; return IGX_ERR_ARG_OUTOFRANGE; }
  for (int k = 1, j = m; k < m; k = j) {
    j = next_knot(m, U, k, 1);
    if (j - k > p) { err = "Knot multiplicity greater than degree"; return IGX_ERR_ARG_OUTOFRANGE; }
  }
  ax.m = m;
  ax.U.assign(U, U + m + 1);
  axis_finish(ax);
  return 0;
}

// ------------------------------------------------------------------ 1-D tables (src/petigabasis.c:83-219)
int basis_init(Basis1D &b, const Axis &ax, const Rule1D &rule, int nqp, std::string &err) {
  std::vector<double> Xv(std::max(nqp, 10)), Wv(std::max(nqp, 10));
  double *X = Xv.data(), *W = Wv.data();
  if (int rc = rule_setup(rule, nqp, X, W, err)) return rc;
  if (ax.p > 7) { err = "degree > 7 not supported"; return IGX_ERR_SUP; }
  const int p = ax.p, nel = ax.nel, nen = p + 1, d = std::min(p, 4);
  // IGA_RULE_REDUCED (src/petigabasis.c:144-171): the first and the last element keep the nqp points, the others take the rule with
  // nqp - 1.  The reference pads the last slot with weight 0 and trims it per element (IGA_Quadrature_SIZE, src/petigaelem.c:764-776);
  // the kernels here keep one point count per axis, so the padded slot stays in the tables with weight 0 -- it adds an exact zero --
  // and sits at the element's midpoint, where every form is finite (the reference leaves PETSC_MAX_REAL and no basis values there).
  std::vector<double> Xr, Wr;
  if (rule.type == IGX_RULE_REDUCED && nel > 2 && nqp > 1) {
    Xr.assign(std::max(nqp, 10), 0.0); Wr.assign(std::max(nqp, 10), 0.0);
    if (gauss_legendre(nqp - 1, Xr.data(), Wr.data())) { err = "Number of quadrature points not implemented"; return IGX_ERR_ARG_OUTOFRANGE; }
  }
  b.nel = nel; b.nqp = nqp; b.nen = nen;
  b.offset.assign(nel, 0); b.detJac.assign(nel, 0.0);
  b.weight.assign((size_t)nel * nqp, 0.0); b.point.assign((size_t)nel * nqp, 0.0);
  b.value.assign((size_t)nel * nqp * nen * 5, 0.0);
  for (int e = 0; e < nel; ++e) {
    const int k = ax.span[e];
    const double u0 = ax.U[k], u1 = ax.U[k + 1], J = (u1 - u0) / 2;
    b.detJac[e] = J;
    b.offset[e] = k - p;
    const bool reduced = !Xr.empty() && e > 0 && e < nel - 1;
    for (int q = 0; q < nqp; ++q) {
      const bool pad = reduced && q == nqp - 1;
      b.weight[(size_t)e * nqp + q] = reduced ? (pad ? 0.0 : Wr[q]) : W[q];
      const double u = ((reduced ? (pad ? 0.0 : Xr[q]) : X[q]) + 1) * J + u0;
      b.point[(size_t)e * nqp + q] = u;
      bspline_ders(k, u, p, d, ax.U.data(), &b.value[((size_t)e * nqp + q) * nen * 5]);
    }
  }
  {   // basis at the two ends of the axis for boundary-form passes (src/petigabasis.c:196-216)
    const int k0 = ax.span[0], k1 = ax.span[nel - 1];
    b.bnd_point[0] = ax.U[k0]; b.bnd_point[1] = ax.U[k1 + 1];
    for (int sd = 0; sd < 2; ++sd) b.bnd_value[sd].assign((size_t)nen * 5, 0.0);
    bspline_ders(k0, b.bnd_point[0], p, d, ax.U.data(), b.bnd_value[0].data());
    bspline_ders(k1, b.bnd_point[1], p, d, ax.U.data(), b.bnd_value[1].data());
  }
  return 0;
}

// ------------------------------------------------------------------ processor grid (src/petigapart.c:11-168)
// The search below follows the reference step for step (initial guesses from the real-valued optimum,
// three descending sweeps, cube tie-breaks) because the resulting grid is part of the bit-exact contract.
namespace {
struct Grid2 { int m, n, cut; };
struct Grid3 { int m, n, p, cut; };
inline int cut2(int M, int N, int m, int n) { return M * (n - 1) + N * (m - 1); }
inline int cut3(int M, int N, int P, int m, int n, int p) { return N * P * (m - 1) + M * P * (n - 1) + M * N * (p - 1); }
inline int snap(int size, int m) { if (m == 0) m = 1; while (m > 0 && size % m) --m; return m; }

Grid2 guess2(int size, int M, int N) {
  Grid2 g;
  g.m = snap(size, (int)(0.5 + std::sqrt(((double)M) / ((double)N) * ((double)size))));
  g.n = size / g.m;
  g.cut = cut2(M, N, g.m, g.n);
  return g;
}
Grid2 best2(int size, int M, int N) {
  Grid2 a = guess2(size, M, N), bt = guess2(size, N, M), r;
  if (a.cut < bt.cut) { r.m = a.m; r.n = a.n; } else { r.m = bt.n; r.n = bt.m; }
  if (M == N && r.n < r.m) std::swap(r.m, r.n);
  r.cut = cut2(M, N, r.m, r.n);
  return r;
}
Grid3 sweep3(int size, int M, int N, int P) {
  Grid3 g;
  g.m = snap(size, (int)(0.5 + std::pow(((double)M * (double)M) / ((double)N * (double)P) * (double)size, 1. / 3.)));
  { Grid2 h = best2(size / g.m, N, P); g.n = h.m; g.p = h.n; }
  int C = cut3(M, N, P, g.m, g.n, g.p);
  for (int mm = g.m; mm >= 1; --mm) {
    if (size % mm) continue;
    Grid2 h = best2(size / mm, N, P);
    int CC = cut3(M, N, P, mm, h.m, h.n);
    if (CC < C) { g.m = mm; g.n = h.m; g.p = h.n; C = CC; }
  }
  for (int nn = g.n; nn >= 1; --nn) {
    if (size % nn) continue;
    Grid2 h = best2(size / nn, M, P);
    int CC = cut3(M, N, P, h.m, nn, h.n);
    if (CC < C) { g.m = h.m; g.n = nn; g.p = h.n; C = CC; }
  }
  for (int pp = g.p; pp >= 1; --pp) {
    if (size % pp) continue;
    Grid2 h = best2(size / pp, M, N);
    int CC = cut3(M, N, P, h.m, h.n, pp);
    if (CC < C) { g.m = h.m; g.n = h.n; g.p = pp; C = CC; }
  }
  g.cut = cut3(M, N, P, g.m, g.n, g.p);
  return g;
}
void best3(int size, int M, int N, int P, int &m, int &n, int &p) {
  Grid3 a = sweep3(size, M, N, P), b = sweep3(size, N, M, P), c = sweep3(size, P, M, N);
  int mm[3] = {a.m, b.n, c.n}, nn[3] = {a.n, b.m, c.p}, pp[3] = {a.p, b.p, c.m}, C[3] = {a.cut, b.cut, c.cut};
  int best = 0, Cmin = INT_MAX;
  for (int k = 0; k < 3; ++k) if (C[k] < Cmin) { Cmin = C[k]; best = k; }
  m = mm[best]; n = nn[best]; p = pp[best];
  if (M == N && n < m) std::swap(m, n);
  if (M == P && p < m) std::swap(m, p);
  if (N == P && p < n) std::swap(n, p);
}
}  // namespace

int partition(int size, int rank, int dim, const int N[3], int n[3], int coords[3]) {
  if (size < 1 || rank < 0 || rank >= size) return IGX_ERR_ARG_OUTOFRANGE;
  if (dim == 3) {
    int &m = n[0], &q = n[1], &p = n[2];
    if (m < 1 && q < 1 && p < 1) best3(size, N[0], N[1], N[2], m, q, p);
    else if (m < 1 && q < 1) { Grid2 g = best2(size / p, N[0], N[1]); m = g.m; q = g.n; }
    else if (m < 1 && p < 1) { Grid2 g = best2(size / q, N[0], N[2]); m = g.m; p = g.n; }
    else if (q < 1 && p < 1) { Grid2 g = best2(size / m, N[1], N[2]); q = g.m; p = g.n; }
    else if (m < 1) m = size / (q * p);
    else if (q < 1) q = size / (m * p);
    else if (p < 1) p = size / (m * q);
  } else if (dim == 2) {
    if (n[0] < 1 && n[1] < 1) { Grid2 g = best2(size, N[0], N[1]); n[0] = g.m; n[1] = g.n; }
    else if (n[0] < 1) n[0] = size / n[1];
    else if (n[1] < 1) n[1] = size / n[0];
  } else if (dim == 1) {
    if (n[0] < 1) n[0] = size;
  } else return IGX_ERR_ARG_OUTOFRANGE;
  int prod = 1;
  for (int k = 0; k < dim; ++k) prod *= n[k];
  if (prod != size) return IGX_ERR_ARG_OUTOFRANGE;
  for (int k = 0; k < dim; ++k) if (N[k] < n[k]) return IGX_ERR_ARG_OUTOFRANGE;
  for (int k = 0; k < dim; ++k) { coords[k] = rank % n[k]; rank = (rank - coords[k]) / n[k]; }
  return 0;
}

void distribute(int dim, const int size[3], const int rank[3], const int N[3], int n[3], int s[3]) {   // src/petigapart.c:170-202
  for (int k = 0; k < dim; ++k) {
    const int q = N[k] / size[k], r = N[k] % size[k];
    n[k] = q + (r > rank[k] ? 1 : 0);
    s[k] = rank[k] * q + std::min(rank[k], r);
  }
}

// ------------------------------------------------------------------ stencil of a basis function (src/petigamat.c:197-233)
void stencil(const Axis &ax, int i, int *first, int *last) {
  const int p = ax.p, m = ax.m, n = m - p - 1;
  const double *U = ax.U.data();
  *first = next_knot(m, U, i, +1) - p - 1;
  *last = next_knot(m, U, i + p + 1, -1);
  if (!ax.periodic) {
    if (i <= p) *first = 0;
    if (i >= n - p) *last = n;
  } else if (i == 0) {
    const int k = n + 1, j = next_knot(m, U, k, +1), s = j - k, C = p - s, nnp = n - C;
    *first = next_knot(m, U, nnp, +1) - nnp - p - 1;
  }
}

// ------------------------------------------------------------------ IGASetUp
// The ghost-row exchange (exchange.hpp) moves a rank's ghost rows to its +1 neighbour on each axis.  That is every owner
// only while no rank's ghost layer reaches past its neighbour's owned nodes, i.e. while every rank of a split axis owns at
// least as many nodes as its lower neighbour has ghosts (fewer than p elements per rank breaks it: the reference's PETSc
// stash would route such rows two ranks up).  Refused by the exchange entry points rather than summed into the wrong rows.
int exchange_supported(const Space &s, std::string &err) {
  // Ghost rows go to their true owners however thin the ranks are (exchange.hpp: one message per rank the ghost layer reaches).
  // What cannot be served: a periodic axis whose ghost layer reaches around to the sender itself (fewer nodes on the other
  // ranks together than one ghost layer).
  for (int i = 0; i < s.dim; ++i) {
    const Axis &ax = s.axis[i];
    const int np = s.proc_sizes[i], nel = s.elem_sizes[i], p = ax.p;
    if (np == 1 || !ax.periodic) continue;
    auto ranges = [&](int c, int &lw, int &gw) {   // node ranges of the rank with coordinate c on this axis (space_setup)
      const int q = nel / np, r = nel % np, ew = q + (r > c ? 1 : 0), es = c * q + std::min(c, r), el = es + ew - 1;
      const int lstart = ax.span[es] - p, gend = ax.span[el] + 1;
      const int lend = (el < nel - 1) ? ax.span[el + 1] - p : ax.span[el] + 1;
      lw = (c == np - 1) ? ax.nnp - lstart : lend - lstart; gw = gend - lstart;
    };
    for (int c = 0; c < np; ++c) {
      int lw, gw; ranges(c, lw, gw);
      int others = 0;
      for (int k = 1; k < np; ++k) { int lwn, gwn; ranges((c + k) % np, lwn, gwn); others += lwn; }
      if (gw - lw > others) {
        err = "axis " + std::to_string(i) + ": the ghost layer of a rank reaches around the periodic axis to the rank itself; use fewer processors on this axis";
        return IGX_ERR_SUP;
      }
    }
  }
  return 0;
}

int space_setup(Space &s, std::string &err) {
  const int dim = s.dim;
  if (dim < 1 || dim > 3) { err = "Must call IGASetDim() first"; return IGX_ERR_ARG_WRONGSTATE; }
  if (s.dof < 1) { err = "Must call IGASetDof() first"; return IGX_ERR_ARG_WRONGSTATE; }
  for (int i = 0; i < dim; ++i) {
    if (s.axis[i].p < 1) { err = "Must call IGAAxisSetDegree() first"; return IGX_ERR_ORDER; }
    if (s.axis[i].m < 2 * s.axis[i].p + 1) { err = "Must call IGAAxisSetKnots() first"; return IGX_ERR_ORDER; }
    if (s.axis[i].span.empty()) axis_finish(s.axis[i]);
  }
  int N[3] = {1, 1, 1};
  for (int i = 0; i < dim; ++i) N[i] = s.axis[i].nel;
  int n[3] = {s.proc_req[0], s.proc_req[1], s.proc_req[2]}, c[3] = {0, 0, 0};
  if (partition(s.comm_size, s.comm_rank, dim, N, n, c)) { err = "Bad partition"; return IGX_ERR_ARG_OUTOFRANGE; }
  for (int i = 0; i < 3; ++i) { s.proc_sizes[i] = i < dim ? n[i] : 1; s.proc_ranks[i] = i < dim ? c[i] : 0; }
  for (int i = 0; i < dim; ++i) s.elem_sizes[i] = N[i];
  distribute(dim, s.proc_sizes, s.proc_ranks, s.elem_sizes, s.elem_width, s.elem_start);
  for (int i = 0; i < dim; ++i) {   // src/petiga.c:1160-1209
    const Axis &ax = s.axis[i];
    const int nel = N[i], ef = s.elem_start[i], el = ef + s.elem_width[i] - 1, p = ax.p;
    const int lstart = ax.span[ef] - p, gstart = lstart, gend = ax.span[el] + 1;
    const int lend = (el < nel - 1) ? ax.span[el + 1] - p : ax.span[el] + 1;
    s.node_sizes[i] = ax.nnp;
    s.node_lstart[i] = lstart; s.node_lwidth[i] = lend - lstart;
    s.node_gstart[i] = gstart; s.node_gwidth[i] = gend - gstart;
    if (s.proc_ranks[i] == s.proc_sizes[i] - 1) s.node_lwidth[i] = s.node_sizes[i] - s.node_lstart[i];
  }
  for (int i = dim; i < 3; ++i) {
    s.elem_sizes[i] = 1; s.elem_start[i] = 0; s.elem_width[i] = 1;
    s.node_sizes[i] = 1; s.node_lstart[i] = 0; s.node_lwidth[i] = 1; s.node_gstart[i] = 0; s.node_gwidth[i] = 1;
  }
  s.nsd = 0; s.rational = 0; s.geomX.clear(); s.geomW.clear(); s.propA.clear();   // src/petiga.c:1290-1299
  if (s.order < 0) { int o = 0; for (int i = 0; i < dim; ++i) o = std::max(o, s.axis[i].p); s.order = std::min(std::max(o, 1), 4); }
  for (int i = 0; i < dim; ++i) {
    const int q = s.rule_nqp[i] > 0 ? s.rule_nqp[i] : s.axis[i].p + 1;   // src/petigabasis.c:103
    if (int rc = basis_init(s.basis[i], s.axis[i], s.rule[i], q, err)) return rc;
  }
  for (int i = dim; i < 3; ++i) {   // a collapsed axis: one element, one point, the constant 1
    Basis1D &b = s.basis[i];
    b.nel = 1; b.nqp = 1; b.nen = 1;
    b.offset.assign(1, 0); b.detJac.assign(1, 1.0); b.weight.assign(1, 1.0); b.point.assign(1, 0.0);
    b.value.assign(5, 0.0); b.value[0] = 1.0;
    s.axis[i] = Axis();
  }
  if (int rc = space_layout(s, err)) return rc;
  s.setup = true;
  return 0;
}

// ------------------------------------------------------------------ per-axis index layout of the local matrix
// Rows: the ghosted node box [gstart, gstart+gwidth) (src/petiga.c:1172-1208); a periodic axis held by a
// single rank wraps onto [0,nnp) (src/petigagrid.c:158-163).  Columns: the union of the rows' stencils
// (src/petigamat.c:243-267), sorted as PETSc stores an AIJ/BAIJ row (ascending column index).
int space_layout(Space &s, std::string &err) {
  for (int d = 0; d < 3; ++d) {
    AxisLayout &L = s.lay[d];
    L = AxisLayout();
    if (d >= s.dim) {
      L.p = 0; L.gstart = 0; L.gwidth = 1; L.nrow = 1; L.ncol = 1;
      L.rowmap = {0}; L.rownode = {0}; L.colnode = {0}; L.rcnt = {1}; L.rcol = {0}; L.P = {0}; L.owned = {1};
      L.ncolors = 1; L.color = {0};
      continue;
    }
    const Axis &ax = s.axis[d];
    const int p = ax.p, nnp = ax.nnp, W = 2 * p + 1;
    L.p = p; L.gstart = s.node_gstart[d]; L.gwidth = s.node_gwidth[d];
    L.alias = (ax.periodic && s.proc_sizes[d] == 1) ? 1 : 0;
    // A periodic axis with fewer than 2p+1 basis functions: the stencil of a row wraps onto itself and the reference's Mat adds
    // the duplicate columns (ghost indices through the LGMap, src/petigamat.c:243-267).  Held by one rank the row keeps its
    // distinct columns and P sends every duplicate to the same position; the elements of such an axis get one colour each
    // (below).  Still refused: fewer than p+1 functions (an element's own basis functions would alias each other) and the
    // axis split over ranks (a rank could not even hold its ghost layer, exchange_supported()).
    if (ax.periodic && nnp < (L.alias ? p + 1 : W)) { err = L.alias ? "periodic axis needs at least p+1 basis functions" : "periodic axis split over ranks needs at least 2p+1 basis functions"; return IGX_ERR_SUP; }
    auto wrapn = [&](int i) { int r = i % nnp; return r < 0 ? r + nnp : r; };
    auto st = [&](int I, int &f, int &l) {   // stencil of the (possibly unwrapped) node I, unwrapped
      if (!ax.periodic) { stencil(ax, I, &f, &l); return; }
      const int w = wrapn(I), shift = I - w;
      stencil(ax, w, &f, &l); f += shift; l += shift;
    };
    L.rowmap.resize(L.gwidth);
    if (L.alias) {
      L.nrow = nnp; L.ncol = nnp; L.cstart = 0;
      for (int i = 0; i < L.gwidth; ++i) L.rowmap[i] = wrapn(L.gstart + i);
      L.rownode.resize(nnp); L.colnode.resize(nnp);
      for (int i = 0; i < nnp; ++i) { L.rownode[i] = i; L.colnode[i] = i; }
    } else {
      L.nrow = L.gwidth;
      int f0, l0, f1, l1;
      st(L.gstart, f0, l0); st(L.gstart + L.gwidth - 1, f1, l1);
      L.cstart = f0; L.ncol = l1 - f0 + 1;
      L.rownode.resize(L.nrow); L.colnode.resize(L.ncol);
      for (int i = 0; i < L.gwidth; ++i) { L.rowmap[i] = i; L.rownode[i] = ax.periodic ? wrapn(L.gstart + i) : L.gstart + i; }
      for (int j = 0; j < L.ncol; ++j) L.colnode[j] = ax.periodic ? wrapn(L.cstart + j) : L.cstart + j;
    }
    L.rcnt.assign(L.nrow, 0); L.rcol.assign((size_t)L.nrow * W, -1); L.owned.assign(L.nrow, 0);
    L.P.assign((size_t)L.gwidth * W, -1);
    for (int i = 0; i < L.gwidth; ++i) {
      const int I = L.gstart + i, r = L.rowmap[i];
      int f, l; st(I, f, l);
      if (l - f + 1 > W) { err = "stencil wider than 2p+1"; return IGX_ERR_PLIB; }
      std::vector<int> cols;
      for (int Jn = f; Jn <= l; ++Jn) cols.push_back(L.alias ? wrapn(Jn) : Jn - L.cstart);
      std::vector<int> sorted = cols;
      std::sort(sorted.begin(), sorted.end());
      sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
      L.rcnt[r] = (int)sorted.size();
      for (size_t k = 0; k < sorted.size(); ++k) L.rcol[(size_t)r * W + k] = sorted[k];
      for (int dlt = 0; dlt < W; ++dlt) {
        const int Jn = I + dlt - p;
        if (Jn < f || Jn > l) continue;
        const int c = L.alias ? wrapn(Jn) : Jn - L.cstart;
        L.P[(size_t)i * W + dlt] = (int)(std::lower_bound(sorted.begin(), sorted.end(), c) - sorted.begin());
      }
      // ownership: [lstart, lstart+lwidth) on this axis (src/petiga.c:1172-1208); aliased rows are all owned
      const int ls = s.node_lstart[d], lw = s.node_lwidth[d];
      L.owned[r] = L.alias ? 1 : ((I >= ls && I < ls + lw) ? 1 : 0);
    }
    // element colours: e mod (p+1); elements e, e' with e'-e >= p+1 never share a basis function because
    // offset[e] strictly increases.  On an aliased periodic axis the trailing N mod (p+1) elements would
    // meet element 0's colour through the wrap, so each gets a colour of its own.
    const int nel = s.elem_width[d], stride = p + 1;
    L.color.resize(nel);
    int tail = L.alias ? nel % stride : 0;
    if (L.alias && nel < 2 * stride) tail = nel;   // tiny periodic axes: one colour per element
    const int regular = nel - tail;
    for (int e = 0; e < nel; ++e) L.color[e] = e < regular ? e % stride : std::min(regular, stride) + (e - regular);
    L.ncolors = std::min(regular, stride) + tail;
    if (L.ncolors < 1) L.ncolors = 1;
  }
  return 0;
}

}  // namespace igx
