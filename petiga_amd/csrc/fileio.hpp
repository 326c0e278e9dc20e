// fileio.hpp -- the on-disk formats either side of the path; included by engine.hip.
// IGARead/IGAWrite (src/petigaio.c:11-139): PETSc binary, i.e. big-endian; PetscInt = int32, PetscReal = double:
//   int classid (IGA_FILE_CLASSID 1211299, include/petiga.h:394) ; int info (bit 0 geometry, bit 1 property) ; int dim ;
//   per axis { int p ; int m+1 ; double U[m+1] } ;
//   if geometry: int nsd ; Vec { int VEC_FILE_CLASSID 1211214 ; int n ; double[n] } with n = prod(n_i+1)*(nsd+1),
//   natural order (axis 0 fastest), per control point (x*w, y*w, z*w, w)  (src/petigaio.c:268-275, :333-343).
// IGAWriteVec/IGAReadVec (src/petigaio.c:640-736): one PETSc Vec in natural order.
#include <cfloat>
#include <cstdio>
#include <exception>

namespace {
const int IGA_FILE_CLASSID_ = 1211299, VEC_FILE_CLASSID_ = 1211214;
inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
inline uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }
bool rd_int(FILE *f, int &v) { uint32_t u; if (fread(&u, 4, 1, f) != 1) return false; u = bswap32(u); memcpy(&v, &u, 4); return true; }
bool rd_dbl(FILE *f, double *a, size_t n) { for (size_t i = 0; i < n; ++i) { uint64_t u; if (fread(&u, 8, 1, f) != 1) return false; u = bswap64(u); memcpy(&a[i], &u, 8); } return true; }
bool wr_int(FILE *f, int v) { uint32_t u; memcpy(&u, &v, 4); u = bswap32(u); return fwrite(&u, 4, 1, f) == 1; }
bool wr_dbl(FILE *f, const double *a, size_t n) { for (size_t i = 0; i < n; ++i) { uint64_t u; memcpy(&u, &a[i], 8); u = bswap64(u); if (fwrite(&u, 8, 1, f) != 1) return false; } return true; }
struct FileCloser { FILE *f; ~FileCloser() { if (f) fclose(f); } };
}  // namespace

// bytes left in the file from the current position (record sizes in a header are checked against it before anything is
// allocated: a corrupt or hostile header must come back as PETSC_ERR_FILE_READ, not as std::bad_alloc through the C ABI)
static long long bytes_left(FILE *f) {
  const long pos = ftell(f);
  if (pos < 0 || fseek(f, 0, SEEK_END) != 0) return -1;
  const long end = ftell(f);
  if (fseek(f, pos, SEEK_SET) != 0) return -1;
  return (long long)end - pos;
}

static int igx_read_into(Space &s, FILE *f, const Space &keep) {
  int classid = 0, info = 0, dim = 0;
  if (!rd_int(f, classid) || classid != IGA_FILE_CLASSID_) return fail(IGX_ERR_ARG_WRONG, "Not an IGA in file");
  if (!rd_int(f, info) || !rd_int(f, dim) || dim < 1 || dim > 3) return fail(66 /*PETSC_ERR_FILE_READ*/, "bad IGA header");
  // IGAReset (src/petiga.c:225) drops the discretisation, not the options the caller set
  s.dof = keep.dof; s.order = keep.order; s.comm_size = keep.comm_size; s.comm_rank = keep.comm_rank; s.env = keep.env;
  for (int i = 0; i < 3; ++i) { s.proc_req[i] = keep.proc_req[i]; s.rule_nqp[i] = keep.rule_nqp[i]; s.rule[i] = keep.rule[i]; s.axis[i].periodic = 0; }
  s.form = keep.form; s.params = keep.params; s.dim = dim;
  for (int i = 0; i < dim; ++i) {
    int p = 0, nk = 0;
    if (!rd_int(f, p) || !rd_int(f, nk) || p < 1 || p > 7 || nk < 2 * (p + 1)) return fail(66, "bad axis record");
    if ((long long)nk * 8 > bytes_left(f)) return fail(66, "truncated knot vector");
    std::vector<double> U((size_t)nk);
    if (!rd_dbl(f, U.data(), U.size())) return fail(66, "truncated knot vector");
    s.axis[i].p = p;
    std::string e;
    if (int rc = axis_set_knots(s.axis[i], nk - 1, U.data(), e)) return fail(rc, e);   // IGAAxisInit, src/petigaaxis.c:314
  }
  if (info & 0x1) {
    int nsd = 0, vid = 0, n = 0;
    if (!rd_int(f, nsd) || nsd < 1 || nsd > 3) return fail(66, "bad geometry dimension");
    if (!rd_int(f, vid) || vid != VEC_FILE_CLASSID_ || !rd_int(f, n) || n < 0) return fail(66, "bad geometry Vec header");
    size_t nnet = 1;
    for (int i = 0; i < dim; ++i) nnet *= (size_t)(s.axis[i].m - s.axis[i].p);   // n_i + 1 control points
    if ((size_t)n != nnet * (nsd + 1)) return fail(IGX_ERR_ARG_WRONG, "geometry Vec size does not match the knot vectors");
    if ((long long)n * 8 > bytes_left(f)) return fail(66, "truncated geometry");
    std::vector<double> xw((size_t)n);
    if (!rd_dbl(f, xw.data(), xw.size())) return fail(66, "truncated geometry");
    s.netX.assign(nnet * nsd, 0.0); s.netW.assign(nnet, 1.0);
    double wmin = DBL_MAX, wmax = -DBL_MAX;
    for (size_t a = 0; a < nnet; ++a) {
      const double w = xw[a * (nsd + 1) + nsd];
      s.netW[a] = w; wmin = std::min(wmin, w); wmax = std::max(wmax, w);
      for (int c = 0; c < nsd; ++c) s.netX[a * nsd + c] = (std::fabs(w) > 0) ? xw[a * (nsd + 1) + c] / w : xw[a * (nsd + 1) + c];
    }
    if (!((wmax - wmin) > 100 * DBL_EPSILON)) s.netW.clear();   // iga->rational, src/petigaio.c:253-255
    s.net_nsd = nsd;
  }
  if (info & 0x2) {      // IGALoad, src/petigaio.c:65-70: the property dimension, then the Vec in natural order [node][npd] (IGALoadProperty :393-458)
    int npd = 0, vid = 0, n = 0;
    if (!rd_int(f, npd) || npd < 1 || npd > 64) return fail(66, "bad property dimension");
    if (!rd_int(f, vid) || vid != VEC_FILE_CLASSID_ || !rd_int(f, n) || n < 0) return fail(66, "bad property Vec header");
    size_t nnet = 1;
    for (int i = 0; i < dim; ++i) nnet *= (size_t)(s.axis[i].m - s.axis[i].p);
    if ((size_t)n != nnet * (size_t)npd) return fail(IGX_ERR_ARG_WRONG, "property Vec size does not match the knot vectors");
    if ((long long)n * 8 > bytes_left(f)) return fail(66, "truncated property array");
    s.netA.assign((size_t)n, 0.0);
    if (!rd_dbl(f, s.netA.data(), s.netA.size())) return fail(66, "truncated property array");
    s.npd = npd;
  }
  return 0;
}

extern "C" int IGXRead(IGX g, const char filename[]) {   // IGARead -> IGALoad, src/petigaio.c:141,11
  NEEDIGA(g); if (!filename) return fail(IGX_ERR_ARG_WRONG, "null file name");
  FileCloser fc{fopen(filename, "rb")};
  if (!fc.f) return fail(65 /*PETSC_ERR_FILE_OPEN*/, std::string("cannot open ") + filename);
  try {
    Space fresh;                       // parsed aside: a failed read leaves the IGX as it was
    if (int rc = igx_read_into(fresh, fc.f, g->s)) return rc;
    g->s = fresh;
  } catch (const std::exception &e) {  // nothing may unwind through the C boundary
    return fail(66, std::string("reading the IGA file failed: ") + e.what());
  }
  touch(g);
  return 0;
}

extern "C" int IGXWrite(IGX g, const char filename[]) {   // IGAWrite -> IGASave, src/petigaio.c:171,75
  NEEDIGA(g); if (!filename) return fail(IGX_ERR_ARG_WRONG, "null file name");
  const Space &s = g->s;
  if (s.dim < 1) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetDim() first");
  FileCloser fc{fopen(filename, "wb")};
  if (!fc.f) return fail(65, std::string("cannot open ") + filename);
  const bool prop = s.npd > 0 && !s.netA.empty();
  bool ok = wr_int(fc.f, IGA_FILE_CLASSID_) && wr_int(fc.f, (s.net_nsd ? 1 : 0) | (prop ? 2 : 0)) && wr_int(fc.f, s.dim);
  for (int i = 0; i < s.dim && ok; ++i) ok = wr_int(fc.f, s.axis[i].p) && wr_int(fc.f, s.axis[i].m + 1) && wr_dbl(fc.f, s.axis[i].U.data(), s.axis[i].U.size());
  if (ok && s.net_nsd) {
    const int nsd = s.net_nsd; const size_t nnet = s.netX.size() / nsd;
    std::vector<double> xw(nnet * (nsd + 1));
    for (size_t a = 0; a < nnet; ++a) {
      const double w = (!s.netW.empty() && std::fabs(s.netW[a]) > 0) ? s.netW[a] : 1.0;
      for (int c = 0; c < nsd; ++c) xw[a * (nsd + 1) + c] = s.netX[a * nsd + c] * w;
      xw[a * (nsd + 1) + nsd] = s.netW.empty() ? 1.0 : s.netW[a];
    }
    ok = wr_int(fc.f, nsd) && wr_int(fc.f, VEC_FILE_CLASSID_) && wr_int(fc.f, (int)xw.size()) && wr_dbl(fc.f, xw.data(), xw.size());
  }
  if (ok && prop) ok = wr_int(fc.f, s.npd) && wr_int(fc.f, VEC_FILE_CLASSID_) && wr_int(fc.f, (int)s.netA.size()) && wr_dbl(fc.f, s.netA.data(), s.netA.size());      // IGASave, src/petigaio.c:130-135
  return ok ? 0 : fail(67 /*PETSC_ERR_FILE_WRITE*/, "write failed");
}

// IGAWriteVec / IGAReadVec (src/petigaio.c:640-736): the Vec in natural order.  One rank only: with several ranks the
// reference scatters to natural order through PETSc, which is outside this library.
extern "C" int IGXWriteVec(IGX g, IGXVec v, const char filename[]) {
  NEEDIGA(g); if (!v || !filename) return fail(IGX_ERR_ARG_WRONG, "null argument");
  if (g->s.comm_size > 1) return fail(IGX_ERR_SUP, "IGXWriteVec is single-rank");
  std::vector<double> h((size_t)v->n);
  if (int rc = IGXVecCopyToHost(v, h.data())) return rc;
  FileCloser fc{fopen(filename, "wb")};
  if (!fc.f) return fail(65, std::string("cannot open ") + filename);
  return (wr_int(fc.f, VEC_FILE_CLASSID_) && wr_int(fc.f, (int)h.size()) && wr_dbl(fc.f, h.data(), h.size())) ? 0 : fail(67, "write failed");
}
extern "C" int IGXReadVec(IGX g, IGXVec v, const char filename[]) {
  NEEDIGA(g); if (!v || !filename) return fail(IGX_ERR_ARG_WRONG, "null argument");
  if (g->s.comm_size > 1) return fail(IGX_ERR_SUP, "IGXReadVec is single-rank");
  FileCloser fc{fopen(filename, "rb")};
  if (!fc.f) return fail(65, std::string("cannot open ") + filename);
  int vid = 0, n = 0;
  if (!rd_int(fc.f, vid) || vid != VEC_FILE_CLASSID_ || !rd_int(fc.f, n)) return fail(66, "not a PETSc Vec file");
  if (n != v->n) return fail(IGX_ERR_ARG_WRONG, "Vec size in file does not match");
  if ((long long)n * 8 > bytes_left(fc.f)) return fail(66, "truncated Vec");
  std::vector<double> h((size_t)n);
  if (!rd_dbl(fc.f, h.data(), h.size())) return fail(66, "truncated Vec");
  return IGXVecCopyFromHost(v, h.data());
}
