// igx.hpp -- internal types of libpetiga_amd (host discretisation + device descriptors).
// Reference citations are file:line of dalcinl/PetIGA @ 2025-04-04.
#pragma once
// IGX_RTC: this header is also the prelude of run-time compiled user forms (rtc.hpp): hiprtc has no standard library and no
// hip_runtime.h to include, so the host part is skipped and the few fixed-width types are spelled out.
#ifndef IGX_RTC
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/petiga_amd.h"
#else
typedef long long int64_t;
typedef int int32_t;
typedef unsigned long size_t;
#define IGX_ERR_USER 83
#endif

namespace igx {

// Experiment hooks (cycle stamps, scatter / MFMA phase switches) are compiled into the kernels only with -DIGX_DEBUG.
#ifdef IGX_DEBUG
constexpr bool kDebug = true;
#else
constexpr bool kDebug = false;
#endif

#ifndef IGX_RTC
// Environment switches, read once when an IGX is created (IGXCreate / IGXCreateFromTables)
struct EnvSwitches {
  int kernel = 0;            // IGX_KERNEL=0..4 presets IGXSetKernel (the parity suite runs every case under two kernel families)
  int walk_axis = 0;         // IGX_WALK_AXIS: preferred walk axis of the pencil kernel
  int nseg = 0;              // IGX_NSEG: segments per pencil (0 = model)
  int clock_probe = 0;       // IGX_CLOCK_PROBE: the pencil kernel records shader-clock ticks against the 100 MHz wall clock (IGXGetClockProbe)
  int overlap = -1;          // IGX_OVERLAP: 0 = no face-first passes, the ghost-row exchange starts after the last launch; 1 = always; 2 = the face of axis 2
                             // alone (round 3); unset (-1): the pencil walk decides by cost against the size of the faces, the other kernels make their pass
  int fuse_groups = 1;       // IGX_FUSE_GROUPS=0: one launch per group of row fields again (NS-VMS p=3; experiment switch)
  int no_first_touch = 0;    // IGX_NO_FIRST_TOUCH: MatZeroEntries + read-modify-write everywhere
  int feature_lds_kb = 0;    // IGX_FEATURE_LDS_KB: LDS target of the feature kernel
  int block_pencil = 1;      // IGX_BLOCK_PENCIL=0: constant-coefficient multi-field forms stay on the feature kernel (block_pencil.hpp)
  int no_vec_pairs = 0;      // IGX_NO_VEC_PAIRS=1: vec_sumfact keeps one element per wavefront at p <= 2
  int state_pencil = 1;      // IGX_STATE_PENCIL=0: Tangents of scalar forms stay on the feature kernel (gram_mfma.hpp: state_pencil)
  int vec_sumfact = 1;       // IGX_VEC_SUMFACT=0: the vector-only drivers stay on the feature kernel (vec_sumfact.hpp)
  int free_run = -1;         // IGX_FREE_RUN=0/1: the pencil walk with / without its s_barrier ping-pong (-1: the launcher's choice)
  int p2_pack = 1;           // IGX_P2_PACK=0: the p = 2 walks keep one tile per pair of node layers (round 4) instead of the packed tiles
  int band_prio = 0, band_rmw_prio = 0;      // IGX_BAND_PRIO=k: band_pt raises the priority of a workgroup's first k layers; IGX_BAND_RMW_PRIO=1: ... of its read-add-writes
  int patch = 1;             // IGX_PATCH=0: the p = 2 Gram walk keeps one pencil and one window per wavefront (bit-repeatable) instead of the patches of
                             // 4 x 3 pencils with one shared window (gram_patch.hpp, round 6: + 23 % on config 2, the order of its LDS adds is not fixed)
  int small_wpb = 1;         // IGX_SMALL_WPB=0: launches that fill less than half the CUs keep eight-wavefront workgroups (gram_mfma.hpp, round 6)
  int patch_state = 0;       // IGX_PATCH_STATE=1: the Tangent of a p = 2 state form walks patches of 4 x 2 pencils (state_patch_p2) instead of state_pencil_k
  int fuse_resid = 0;        // IGX_FUSE_RESID=1: IGXComputeIFunctionIJacobian takes the fused walk (state_pencil_kr) where it exists; default: the two
                             // drivers one after the other -- measured in round 6: the fused launch costs 2.5 ms more than the Tangent's, the
                             // Residual's own pass 2.1 ms per launch
  int combine = -1;          // IGX_COMBINE: element bricks of the feature kernel (-1 = automatic, 0 = one element per workgroup)
  int debug_feature = 0, debug_noflush = 0, debug_timing = 0;   // only honoured by -DIGX_DEBUG builds
};
EnvSwitches read_env_switches();

// ------------------------------------------------------------------ host discretisation
struct Axis {                 // struct _n_IGAAxis, include/petiga.h:80-96
  int p = 0, m = 0, periodic = 0, nel = 0, nnp = 0;
  std::vector<double> U;
  std::vector<int> span;
};

struct Basis1D {              // struct _n_IGABasis, include/petiga.h:122-141
  int nel = 0, nqp = 0, nen = 0;
  std::vector<int> offset;
  std::vector<double> detJac, weight, point, value;   // value: [nel][nqp][nen][5]
  std::vector<double> bnd_value[2];                   // [nen][5] at the first / last knot (bnd_value, include/petiga.h:136)
  double bnd_point[2] = {0, 0};
};

struct BC {                   // struct _IGAFormBC, include/petiga.h:220-225
  int count = 0;
  int field[64];
  double value[64];
};

struct Rule1D {               // struct _n_IGARule, include/petiga.h:91-99 (the size lives in Space::rule_nqp)
  int type = 0;               // IGXRuleType
  std::vector<double> x, w;   // IGX_RULE_USER: the rule on [-1, 1]
};

int  gauss_legendre(int q, double *X, double *W);
int  gauss_lobatto(int q, double *X, double *W);
int  rule_setup(const Rule1D &r, int nqp, double *X, double *W, std::string &err);
void bspline_ders(int span, double u, int p, int nders, const double *U, double *out /*[p+1][5]*/);
int  axis_init_uniform(Axis &ax, int N, double Ui, double Uf, int C, std::string &err);
int  axis_set_knots(Axis &ax, int m, const double *U, std::string &err);
void axis_finish(Axis &ax);   // spans + nnp from U
int  basis_init(Basis1D &b, const Axis &ax, const Rule1D &rule, int nqp, std::string &err);
int  partition(int size, int rank, int dim, const int N[3], int n[3], int coords[3]);
void distribute(int dim, const int size[3], const int rank[3], const int N[3], int n[3], int s[3]);
void stencil(const Axis &ax, int i, int *first, int *last);   // src/petigamat.c:197-233

// Per-axis index layout of the local matrix (row box, column box, position tables, colours)
struct AxisLayout {
  int p = 0, gstart = 0, gwidth = 1;
  int alias = 0;                    // periodic axis wrapped inside one rank
  int nrow = 1, ncol = 1;
  int cstart = 0;                   // unwrapped node index of local column 0 (non-alias)
  std::vector<int> rowmap;          // [gwidth] ghost index -> row index
  std::vector<int> rownode;         // [nrow]   global node (wrapped)
  std::vector<int> colnode;         // [ncol]   global node (wrapped)
  std::vector<int> rcnt;            // [nrow]   columns in the row's per-axis stencil
  std::vector<int> rcol;            // [nrow][2p+1] sorted local column indices (padded -1)
  std::vector<int> P;               // [gwidth][2p+1] position of column (row + d - p) in the row's list, or -1
  std::vector<int> owned;           // [nrow] 1 if this rank owns the row node on this axis
  int ncolors = 1;
  std::vector<int> color;           // [nel_local]
};

struct Space {
  int dim = 0, dof = 0, order = -1;
  Axis axis[3];
  int rule_nqp[3] = {-1, -1, -1};
  Rule1D rule[3];
  Basis1D basis[3];
  int comm_size = 1, comm_rank = 0;
  int proc_req[3] = {-1, -1, -1};
  int proc_sizes[3] = {1, 1, 1}, proc_ranks[3] = {0, 0, 0};
  int elem_sizes[3] = {1, 1, 1}, elem_start[3] = {0, 0, 0}, elem_width[3] = {1, 1, 1};
  int node_sizes[3] = {1, 1, 1}, node_lstart[3] = {0, 0, 0}, node_lwidth[3] = {1, 1, 1};
  int node_gstart[3] = {0, 0, 0}, node_gwidth[3] = {1, 1, 1};
  int nsd = 0, rational = 0;
  std::vector<double> geomX, geomW;       // ghosted local
  std::vector<double> netX, netW;         // global control net (geometry grid, natural order) kept for IGXWrite / re-partitioning
  int npd = 0;                            // iga->property: numbers per node of the property array (0: none)
  std::vector<double> netA, propA;        // ... on the global net [node][npd] (natural order) / ghosted local (iga->propertyA, include/petiga.h:350-353)
  int net_nsd = 0;
  BC value[3][2], load[3][2];
  bool visit[3][2] = {{false, false}, {false, false}, {false, false}};   // IGAFormSetBoundaryForm, src/petigaform.c:134
  IGXFormKind form = IGX_FORM_NONE;
  std::vector<double> params;
  bool setup = false;
  double link_gbs = 0;        // GB/s per direction of a face message as the communicator measured it (comm.hpp); 0: not measured
  AxisLayout lay[3];
  EnvSwitches env;
};

int  space_setup(Space &s, std::string &err);          // IGASetUp stages 1+3 (src/petiga.c:1111-1310,1450-1493)
int  space_layout(Space &s, std::string &err);         // AxisLayout for the three axes
int  exchange_supported(const Space &s, std::string &err);   // 0, or IGX_ERR_SUP when a periodic ghost layer would wrap onto its own rank

#endif   // !IGX_RTC

// ------------------------------------------------------------------ device descriptors (POD, passed by value)
constexpr int MAXBC = 8;      // fields per face the device tables hold (dof <= 8 on the device path)
constexpr int NDER = 4;       // 1-D derivative slots kept on the device: orders 0..3

struct BCDev { int count; int field[MAXBC]; double value[MAXBC]; };

struct AxisDev {
  int nel, nqp, nen, p;
  int estart, esizes, periodic;
  int gwidth, nrow, ncol;
  const double *tab;   // [nel][nqp][nen][NDER]
  const double *w;     // [nel][nqp]
  const double *J;     // [nel]
  const double *pt;    // [nel][nqp]
  const double *bnd;   // [2 sides][nen][NDER] basis at the first / last knot of the axis
  double bndpt[2];
  const int *off;      // [nel]  ghost-local index of the element's first basis function
  const int *rowmap;   // [gwidth]
  int rwrap;           // rowmap in closed form: rowmap[i] = i < rwrap ? i : i - rwrap (a periodic axis wrapped inside the rank: nnp)
  const int *rcnt;     // [nrow]
  const int *P;        // [gwidth][2p+1]
  const int64_t *prefix;  // [nrow+1] exclusive prefix sums of rcnt
  int64_t tot;         // sum of rcnt
  int off_lin, off0;   // off in closed form: off[e] = off0 + e for every element of the rank (one new basis function per element: maximal continuity); else off_lin = 0
};

struct SpaceDev {
  int dim, dof, order, nsd, rational;
  AxisDev ax[3];
  const double *X;     // ghosted local [.][nsd] or null
  const double *W;     // ghosted local or null
  BCDev bcv[3][2], bcl[3][2];
  const double *fixtable;  // row-indexed [nrows][dof] or null
  const double *A;         // property array, ghosted local [.][npd], or null
  int npd;
};

struct ColorRange { int start[3], step[3], count[3]; };

enum Op { OP_SYSTEM = 0, OP_MATRIX, OP_VECTOR, OP_FUNCTION, OP_JACOBIAN, OP_IFUNCTION, OP_IJACOBIAN, OP_SCALAR };

struct OutDev {
  const int64_t *browptr;  // null when no matrix output
  double *val;
  double *vec;             // null when no vector output
  const double *U, *V;     // row-indexed state vectors or null
  double shift, t;
  int op;
  int *errflag;
  int bid;                 // boundary-form pass: 2*axis+side (IGAElementNextForm, src/petigaelem.c:427); -1 = interior pass
  int first_touch;         // 1: the matrix was not zeroed: the first colour that reaches an entry stores it (feature kernel)
  int debug;               // experiment switches (IGX_DEBUG_FEATURE): 1 no scatter, 2 atomic scatter, 4 no MFMA phase
  long long *clk;          // IGX_CLOCK_PROBE: [ticks, wall ticks, elements] of workgroup 0 of the last pencil launch, or null
  long long *dbg;          // experiment: s_memtime stamps of workgroup 0 per phase (IGX_DEBUG_FEATURE & 8)
  int ft2_lo, ft2_hi, ft2_blocked;   // feature kernel, assembly in two passes over axis 2 (upper face first): first-touch rule of the pass; ft2_hi = 0: one pass
  int64_t elem_base;       // OP_SCALAR: index of this launch's first element in the per-element partial sums (vec)
  int vec_mode;            // vec_sumfact as a part of IGAComputeSystem next to a band-row kernel: 1 the whole vector (lifting of the Dirichlet values through
                           // SystemVectorOf<Form>, a fixed row takes its value), 2 the form's vec() alone with the fixed rows left at 0 (block_pencil lifts itself)
};

constexpr int MAXPARAM = 8;
struct ParamsDev { double v[MAXPARAM]; };

}  // namespace igx
