// rtc.hpp -- run-time compiled user forms (included by engine.hip, main unit only).
//
// PetIGA's point callbacks are arbitrary user functions (IGAFormSystem ... IGAFormIJacobian, include/petiga.h:153-197,
// registered through IGASetForm*, src/petigaform.c:388-833).  Host function pointers cannot run on the GPU and the built-in
// forms are a closed list, so the open end of the plugin API is source: the user hands over a HIP struct with the contract of
// forms.hpp (DOF, ORDER, NEED; mat() = un-weighted K block of a basis pair, vec() = un-weighted F entries of a basis
// function; zeroed outputs, row-major, src/petigapoint.c:427-462), IGXSetFormSource compiles
// generic_assemble<UserForm, dim> with hiprtc against the library's own headers (embedded at build time, rtc_sources.inc) and
// the seven drivers launch it like any built-in form: same closure, tabulation, boundary fix-up and coloured scatter.
// hiprtc is bound with dlopen, as RCCL is.  The compile itself needs no GPU (tests/test_rtc_forms.py checks the compile and
// its error log on the CPU; the launch is a GPU test against the oracle).
#include <dlfcn.h>
#include <unistd.h>
#include <map>
#include "rtc_sources.inc"

namespace {

struct HiprtcApi {
  void *h = nullptr;
  int (*Create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*AddName)(void *, const char *) = nullptr;
  int (*Compile)(void *, int, const char **) = nullptr;
  int (*LogSize)(void *, size_t *) = nullptr;
  int (*Log)(void *, char *) = nullptr;
  int (*Lowered)(void *, const char *, const char **) = nullptr;
  int (*CodeSize)(void *, size_t *) = nullptr;
  int (*Code)(void *, char *) = nullptr;
  int (*Destroy)(void **) = nullptr;
  int (*Version)(int *, int *) = nullptr;      // hiprtcVersion(major, minor): part of the cache key
};
static HiprtcApi &hiprtc_api() { static HiprtcApi a; return a; }

static int load_hiprtc(std::string &err) {
  HiprtcApi &a = hiprtc_api();
  if (a.h) return 0;
  const char *env = getenv("IGX_HIPRTC_LIB");
  const char *names[] = {env, "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
  for (int pass = 0; pass < 2 && !a.h; ++pass)
    for (const char *n : names) { if (!n || !*n) continue; a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)); if (a.h) break; }
  if (!a.h) {
    const char *why = dlerror();     // one call: the second would return NULL
    err = std::string("cannot load libhiprtc.so: ") + (why ? why : "not found");
    return IGX_ERR_LIB;
  }
  auto sym = [&](const char *n) { return dlsym(a.h, n); };
  a.Create = reinterpret_cast<decltype(a.Create)>(sym("hiprtcCreateProgram"));
  a.AddName = reinterpret_cast<decltype(a.AddName)>(sym("hiprtcAddNameExpression"));
  a.Compile = reinterpret_cast<decltype(a.Compile)>(sym("hiprtcCompileProgram"));
  a.LogSize = reinterpret_cast<decltype(a.LogSize)>(sym("hiprtcGetProgramLogSize"));
  a.Log = reinterpret_cast<decltype(a.Log)>(sym("hiprtcGetProgramLog"));
  a.Lowered = reinterpret_cast<decltype(a.Lowered)>(sym("hiprtcGetLoweredName"));
  a.CodeSize = reinterpret_cast<decltype(a.CodeSize)>(sym("hiprtcGetCodeSize"));
  a.Code = reinterpret_cast<decltype(a.Code)>(sym("hiprtcGetCode"));
  a.Destroy = reinterpret_cast<decltype(a.Destroy)>(sym("hiprtcDestroyProgram"));
  a.Version = reinterpret_cast<decltype(a.Version)>(sym("hiprtcVersion"));
  if (!a.Create || !a.AddName || !a.Compile || !a.LogSize || !a.Log || !a.Lowered || !a.CodeSize || !a.Code || !a.Destroy) { err = "libhiprtc.so lacks the expected API"; a = HiprtcApi(); return IGX_ERR_LIB; }
  return 0;
}

}  // namespace

// feature_assemble<UserForm, dim, TA, NW, I0, DOFI, HASM> for one wave layout: one function per group of row fields
struct RtcFeature {
  bool guard_known = false, guard_ok = true; std::vector<double> guard_prm;     // band_pt: the struct's band_params_ok at the parameters it was last asked about
  std::vector<char> code;
  std::vector<std::string> lowered;
  hipModule_t module = nullptr; std::vector<hipFunction_t> func;
  int meta[4] = {0, 0, 0, 0};          // workgroups per CU the kernel is compiled for, features kept in LDS, executed MFMAs per k-step, 0
  bool failed = false; std::string why;     // band_pt under the automatic choice: the instantiation did not compile; the form stays on the feature kernel
  ~RtcFeature() { if (module) (void)hipModuleUnload(module); }
};

// one compiled user form for one dimension: code object + what the host-side launcher must know about the struct
struct RtcForm {
  std::string name, source, lowered;
  int dim = 0;
  std::vector<char> code;
  // DOF, ORDER, NEED, NSCALAR, SHAPE_ORDER, MAT_NEED, MAT_PAIR_MASK != 0, has an atboundary branch, MAT_TEST_MASK, MAT_SYMMETRIC,
  // VEC_TEST_MASK, PENCIL_NFEAT or 0 (read from the module)
  // [12] number of Gram pairs (bits of MAT_PAIR_MASK), [13] VEC_ZERO, [14] NCOEF / point_coef declared (band_pt.hpp), [15] 0
  int meta[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  hipModule_t module = nullptr; hipFunction_t func = nullptr;
  std::map<int, std::shared_ptr<RtcFeature>> feature;   // key: TA | NW << 4 | DOFI << 8 | HASM << 12
  std::map<int, std::shared_ptr<RtcFeature>> pencil;    // form_pencil instantiations; key: SYSTEM | P << 1 | IDENT << 4 | RAT << 5
  std::map<int, std::shared_ptr<RtcFeature>> vecsf;     // vec_sumfact instantiations; key: GEO
  std::map<int, std::shared_ptr<RtcFeature>> state;     // state_pencil instantiations; key: P (+ 10 + rational on a mapped geometry, + 100 packed tiles)
  std::map<int, std::shared_ptr<RtcFeature>> block;     // block_pencil instantiations; key: SYSTEM
  std::map<int, std::shared_ptr<RtcFeature>> band;      // band_points + band_pt instantiations; key: GEO | RAT << 1 | degree << 2
  ~RtcForm() { if (module) (void)hipModuleUnload(module); }
};

// Code-object cache on disk (IGX_RTC_CACHE_DIR): key = FNV-1a of the whole program text -- the library's own headers are part of
// it, so another library build never finds a stale object -- and of the name expressions.  File: [count][len, lowered name]...[code].
static unsigned long long rtc_fnv(const std::string &t, unsigned long long h = 1469598103934665603ull) {
  for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; }
  return h;
}
// the options every program is compiled with: also part of the cache key
static const char *kRtcOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
constexpr int kRtcNOpts = 4;
// key of a program: its whole text, the name expressions, the options above, the HIP this library was built against AND the version
// the loaded libhiprtc.so reports (it is bound at run time: another ROCm's compiler never finds this one's objects)
static unsigned long long rtc_program_key(const std::string &src, const std::vector<std::string> &exprs, unsigned long long seed) {
  std::string tool = "hip " + std::to_string(HIP_VERSION);
  for (int i = 0; i < kRtcNOpts; ++i) { tool += ' '; tool += kRtcOpts[i]; }
  { std::string e; int maj = 0, min = 0;
    if (load_hiprtc(e) == 0 && hiprtc_api().Version && hiprtc_api().Version(&maj, &min) == 0) tool += " hiprtc " + std::to_string(maj) + "." + std::to_string(min);
    else tool += " hiprtc ?"; }
  unsigned long long h = rtc_fnv(src, rtc_fnv(tool, seed));
  for (const std::string &x : exprs) h = rtc_fnv(x, h ^ 0x9e3779b97f4a7c15ull);
  return h;
}
static std::string rtc_cache_path(const std::string &src, const std::vector<std::string> &exprs) {
  const char *dir = getenv("IGX_RTC_CACHE_DIR");
  if (!dir || !*dir) return std::string();
  char name[64]; snprintf(name, sizeof(name), "/igx_%016llx.bin", rtc_program_key(src, exprs, 1469598103934665603ull));
  return std::string(dir) + name;
}
static bool rtc_cache_load(const std::string &path, size_t nexpr, std::vector<char> &code, std::vector<std::string> &lowered, unsigned long long check) {
  FILE *f = path.empty() ? nullptr : fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  unsigned n = 0; unsigned long long stored = 0;
  // (the file starts with a second hash of the program -- another seed than the one in its name: a stale or foreign file is refused)
  if (fread(&stored, sizeof(stored), 1, f) == 1 && stored == check && fread(&n, sizeof(n), 1, f) == 1 && n == nexpr) {
    lowered.clear(); ok = true;
    for (unsigned i = 0; i < n && ok; ++i) {
      unsigned len = 0;
      ok = fread(&len, sizeof(len), 1, f) == 1 && len < (1u << 16);
      if (ok) { std::string t(len, '\0'); ok = len == 0 || fread(&t[0], 1, len, f) == len; lowered.push_back(t); }
    }
    unsigned long long cs = 0;
    ok = ok && fread(&cs, sizeof(cs), 1, f) == 1 && cs > 0 && cs < (1ull << 31);
    if (ok) { code.resize((size_t)cs); ok = fread(code.data(), 1, (size_t)cs, f) == cs; }
  }
  fclose(f);
  return ok;
}
static void rtc_cache_store(const std::string &path, const std::vector<char> &code, const std::vector<std::string> &lowered, unsigned long long check) {
  if (path.empty()) return;
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const unsigned n = (unsigned)lowered.size(); bool ok = fwrite(&check, sizeof(check), 1, f) == 1 && fwrite(&n, sizeof(n), 1, f) == 1;
  for (const std::string &t : lowered) { const unsigned len = (unsigned)t.size(); ok = ok && fwrite(&len, sizeof(len), 1, f) == 1 && (len == 0 || fwrite(t.data(), 1, len, f) == len); }
  const unsigned long long cs = code.size(); ok = ok && fwrite(&cs, sizeof(cs), 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());     // (rename: concurrent ranks write the same object)
}

// compiles `tail` behind the library headers and the user's source; returns the code object and the lowered names of `exprs`
static int rtc_build(const std::string &source, bool with_feature, const std::string &tail, const std::vector<std::string> &exprs,
                     std::vector<char> &code, std::vector<std::string> &lowered, bool with_pencil = false, bool with_vecsf = false, bool with_block = false, bool with_band = false) {
  std::string src;
  src.reserve(source.size() + 400000);
  src += "#define IGX_RTC 1\n";
  src += kRtcSrc_igx; src += "\n"; src += kRtcSrc_forms; src += "\n"; src += kRtcSrc_generic; src += "\n";
  if (with_feature) { src += kRtcSrc_feature; src += "\n"; }
  if (with_pencil) { src += kRtcSrc_pencil; src += "\n"; src += kRtcSrc_gram; src += "\n"; }
  if (with_vecsf) { src += kRtcSrc_vecsf; src += "\n"; }
  if (with_block || with_band) { src += kRtcSrc_pencil; src += "\n"; src += kRtcSrc_block; src += "\n"; }
  if (with_band) { src += kRtcSrc_band; src += "\n"; }
  src += "using namespace igx;\n#line 1 \"user_form.hip\"\n";
  src += source;
  src += "\n";
  src += tail;
  const std::string cache = rtc_cache_path(src, exprs);
  const unsigned long long check = cache.empty() ? 0ull : rtc_program_key(src, exprs, 0x84222325cbf29ce4ull);
  if (rtc_cache_load(cache, exprs.size(), code, lowered, check)) return 0;
  std::string e; if (int rc = load_hiprtc(e)) return fail(rc, e);
  HiprtcApi &a = hiprtc_api();
  void *prog = nullptr;
  if (a.Create(&prog, src.c_str(), "igx_user_form.hip", 0, nullptr, nullptr) != 0) return fail(IGX_ERR_LIB, "hiprtcCreateProgram failed");
  for (const std::string &x : exprs) (void)a.AddName(prog, x.c_str());
  const int rc = a.Compile(prog, kRtcNOpts, kRtcOpts);
  if (rc != 0) {
    size_t n = 0; (void)a.LogSize(prog, &n); std::string log(n, '\0'); if (n) (void)a.Log(prog, &log[0]);
    (void)a.Destroy(&prog);
    return fail(IGX_ERR_USER, "the form source does not compile:\n" + log);
  }
  lowered.clear();
  for (const std::string &x : exprs) {
    const char *low = nullptr;
    if (a.Lowered(prog, x.c_str(), &low) != 0 || !low) { (void)a.Destroy(&prog); return fail(IGX_ERR_LIB, "hiprtcGetLoweredName failed"); }
    lowered.push_back(low);
  }
  size_t cs = 0;
  if (a.CodeSize(prog, &cs) != 0 || cs == 0) { (void)a.Destroy(&prog); return fail(IGX_ERR_LIB, "hiprtcGetCodeSize failed"); }
  code.resize(cs);
  if (a.Code(prog, code.data()) != 0) { (void)a.Destroy(&prog); return fail(IGX_ERR_LIB, "hiprtcGetCode failed"); }
  (void)a.Destroy(&prog);
  rtc_cache_store(cache, code, lowered, check);
  return 0;
}

static int rtc_compile(IGX g, const std::string &source, const std::string &name, int dim, std::shared_ptr<RtcForm> &out) {
  const std::string expr = "igx::generic_assemble<" + name + ", " + std::to_string(dim) + ">";
  std::string tail = "// what the host-side launcher reads back\n__device__ int igx_user_meta[16] = {" + name + "::DOF, " + name + "::ORDER, (int)" + name + "::NEED, igx::nscalar_of<" + name +
                     ">::v, igx::shape_order_of<" + name + ">::v, (int)igx::mat_need_of<" + name + ">::v, igx::mat_pair_mask_of<" + name + ">::v != 0ull, igx::has_boundary_of<" + name + ">::v, (int)igx::mat_test_mask_of<" +
                     name + ">::v, igx::mat_symmetric_of<" + name + ">::v, (int)igx::vec_test_mask_of<" + name + ">::v, igx::pencil_state_of<" + name + ">::nfeat, igx::fm_popcount(igx::mat_pair_mask_of<" + name + ">::v), igx::vec_zero_of<" + name + ">::v, igx::has_point_coef<" + name + ">::v, 0};\n";
  tail += "template __global__ void " + expr + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::ColorRange, igx::Carve, double *, size_t);\n";
  std::shared_ptr<RtcForm> f(new RtcForm());
  std::vector<std::string> low;
  if (int rc = rtc_build(source, true, tail, {expr}, f->code, low)) return rc;
  f->lowered = low[0];
  f->name = name; f->source = source; f->dim = dim;
  out = f;
  return 0;
}

static int rtc_load(IGX g, RtcForm &f) {
  if (f.func) return 0;
  HIPCK(hipModuleLoadData(&f.module, f.code.data()));
  HIPCK(hipModuleGetFunction(&f.func, f.module, f.lowered.c_str()));
  hipDeviceptr_t p = nullptr; size_t n = 0;
  HIPCK(hipModuleGetGlobal(&p, &n, f.module, "igx_user_meta"));
  if (n != sizeof(f.meta)) return fail(IGX_ERR_LIB, "unexpected igx_user_meta size");
  HIPCK(hipMemcpy(f.meta, p, sizeof(f.meta), hipMemcpyDeviceToHost));
  return 0;
}

// the kernel arguments of generic_assemble, laid out as the kernarg segment is (natural alignment, in order)
struct RtcArgs { SpaceDev S; ParamsDev prm; OutDev out; ColorRange cr; Carve cv; double *phi_global; size_t phi_stride; };


// ---- the feature-GEMM kernel (feature_mfma.hpp) for a run-time form: launch_feature / launch_feature_ta / launch_feature_plan of
// engine.hip with the form's constants read from the module.  Element mode only (no pencil walk), matrix / vector drivers.
struct RtcFeatArgs { SpaceDev S; ParamsDev prm; OutDev out; ColorRange cr; FCarve cv; };

// (DOF is passed in: it is read from the module on a GPU, and given by the caller for the compile-only check)
static int rtc_feature_module(IGX g, RtcForm &F, int DIM, int DOF, int TA, int NW, int DOFI, bool HASM, bool load, std::shared_ptr<RtcFeature> &out) {
  const int key = TA | (NW << 4) | (DOFI << 8) | ((HASM ? 1 : 0) << 12);
  auto it = F.feature.find(key);
  if (it != F.feature.end() && (it->second->module || !load)) { out = it->second; return 0; }
  std::shared_ptr<RtcFeature> f(new RtcFeature());
  std::vector<std::string> exprs;
  const std::string common = F.name + ", " + std::to_string(DIM) + ", " + std::to_string(TA) + ", " + std::to_string(NW) + ", ";
  std::string tail;
  const bool fuse = HASM && DOFI < DOF;   // all groups of row fields in one launch (feature_mfma.hpp, FUSE)
  for (int I0 = 0; I0 < ((HASM && !fuse) ? DOF : 1); I0 += DOFI) {
    const std::string x = "igx::feature_assemble<" + common + std::to_string(I0) + ", " + std::to_string(DOFI) + ", " + (HASM ? "true" : "false") + ", false, " + (fuse ? "true" : "false") + ">";
    exprs.push_back(x);
    tail += "template __global__ void " + x + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::ColorRange, igx::FCarve);\n";
  }
  const std::string nfs = "((igx::shape_order_of<" + F.name + ">::v >= 2) ? 1 + " + std::to_string(DIM) + " + " + std::to_string(DIM * DIM) + " : 1 + " + std::to_string(DIM) + ")";
  tail += "__device__ int igx_feature_meta[4] = {igx::fm_min_waves<" + F.name + ", " + std::to_string(TA) + ", " + std::to_string(NW) + ", " + std::to_string(DOFI) + ", " + (HASM ? "true" : "false") +
          ", false>(), igx::fm_popcount((unsigned long long)(igx::phi_mask_of<" + F.name + ">::v & ((1u << " + nfs + ") - 1u))), igx::fm_mfma_per_kstep<" + F.name + ">(" + nfs + "), 0};\n";
  if (int rc = rtc_build(F.source, true, tail, exprs, f->code, f->lowered)) return rc;
  if (!load) { F.feature[key] = f; out = f; return 0; }
  HIPCK(hipModuleLoadData(&f->module, f->code.data()));
  for (const std::string &l : f->lowered) { hipFunction_t fn = nullptr; HIPCK(hipModuleGetFunction(&fn, f->module, l.c_str())); f->func.push_back(fn); }
  hipDeviceptr_t p = nullptr; size_t n = 0;
  HIPCK(hipModuleGetGlobal(&p, &n, f->module, "igx_feature_meta"));
  if (n != sizeof(f->meta)) return fail(IGX_ERR_LIB, "unexpected igx_feature_meta size");
  HIPCK(hipMemcpy(f->meta, p, sizeof(f->meta), hipMemcpyDeviceToHost));
  F.feature[key] = f; out = f;
  return 0;
}

// wave layouts of launch_feature_ta (engine.hip): 8 waves at 4x4 tiles, except scalar matrix forms (4 waves with 4 tiles each);
// dof 4 at 4x4 tiles takes two launches of two row fields
static void rtc_feature_layout(int NE, int DOF, bool GRAM, bool HASM, int &TA, int &NW, int &DOFI) {
  TA = NE <= 16 ? 1 : (NE <= 32 ? 2 : (NE <= 64 ? 4 : (NE <= 128 ? 8 : 16)));
  NW = (TA >= 4) ? 8 : 4; DOFI = DOF;
  if (HASM) { if (TA == 8 && !GRAM) DOFI = 1; if (TA == 4 && DOF == 4 && !GRAM) DOFI = 2; if (TA == 4 && DOF == 1) NW = 4; }
}

static int launch_feature_rtc(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out, bool &done) {
  done = false;
  const Space &s = g->s;
  const int DIM = s.dim, DOF = F.meta[0];
  if (DIM < 2 || DOF > 4) return 0;
  int nq[3], na[3]; int NQ = 1, NE = 1;
  for (int d = 0; d < 3; ++d) { nq[d] = s.basis[d].nqp; na[d] = s.basis[d].nen; NQ *= nq[d]; NE *= na[d]; }
  if (NE > 256) return 0;
  const bool SECOND = F.meta[1] >= 2, SECOND_S = F.meta[4] >= 2, GRAM = F.meta[6] != 0;
  if (NE > 64 && (DIM != 3 || GRAM || DOF > 2)) return 0;      // 8x8 tiles: p = 4 in 3-D, at most two accumulator sets per wave
  if (NE > 128 && DOF > 1) return 0;                           // 16 tile rows x two column panels: p = 5 in 3-D, one set
  if (GRAM && DOF * DOF > 16) return 0;
  const bool HASM = (out.op == OP_SYSTEM || out.op == OP_MATRIX || out.op == OP_JACOBIAN || out.op == OP_IJACOBIAN);
  if (HASM && out.op == OP_SYSTEM && SECOND && !SECOND_S && ((unsigned)F.meta[2] & NEED_HU) && !((unsigned)F.meta[5] & NEED_HU)) return 0;   // (HU_HERE, feature_mfma.hpp)
  int TA, NW, DOFI; rtc_feature_layout(NE, DOF, GRAM, HASM, TA, NW, DOFI);
  std::shared_ptr<RtcFeature> K;
  if (int rc = rtc_feature_module(g, F, DIM, DOF, TA, NW, DOFI, HASM, true, K)) return rc;
  const int WGS = K->meta[0], NPS = K->meta[1];
  const int D2 = DIM * DIM, NFS = SECOND_S ? 1 + DIM + D2 : 1 + DIM;
  const int NEP = 16 * TA, NQ4 = (NQ + 3) & ~3;
  const bool vec_op = (out.op == OP_SYSTEM || out.op == OP_VECTOR || out.op == OP_FUNCTION || out.op == OP_IFUNCTION);
  const unsigned need = vec_op ? (unsigned)F.meta[2] : (unsigned)F.meta[5];
  const bool fields = (need & (NEED_U | NEED_UT | NEED_GU | NEED_HU)) != 0;
  // LDS budget as in launch_feature_plan (a module kernel takes more than 64 KiB of dynamic LDS like any other: measured)
  const size_t lds_limit = 160 * 1024 - 512;
  const size_t lds_auto = s.env.feature_lds_kb > 0 ? (size_t)s.env.feature_lds_kb * 1024
                        : ((NW == 4) ? (size_t)(160 * 1024 / std::max(WGS, 1) - 1024) : ((HASM && TA == 4 && DOF == 1) ? (size_t)78 * 1024 : lds_limit));
  FCarve cv; size_t lds_bytes = 0; bool fits = false;
  for (int pass = 0; pass < 2 && !fits; ++pass) {
    const size_t cap = pass == 0 ? lds_auto : lds_limit;
    for (int nchunk = 1; nchunk <= NQ4 / 4 && !fits; ++nchunk) {
      const int QC = (((NQ4 + nchunk - 1) / nchunk) + 3) & ~3, NQP = QC * nchunk;
      int pos = 0;
      auto take = [&](int n) { int o = pos; pos += (n + 1) & ~1; return o; };
      memset(&cv, 0, sizeof(cv));
      for (int d = 0; d < 3; ++d) { cv.t1d[d] = take(nq[d] * na[d] * NDER); cv.w1d[d] = take(2 * nq[d]); }
      cv.gX = take(NE * DIM); cv.gW = take(NE); cv.Ue = take(NE * DOF); cv.Ve = take(NE * DOF);
      cv.ufix = take(NE * DOF); cv.fixval = take(NE * DOF); cv.fixflag = take(NE * DOF); cv.flux = take(NE * DOF);
      cv.JW = take(NQP); cv.xq = take(NQP * DIM); cv.E1 = take(s.nsd ? NQP * D2 : 0); cv.E2 = take((s.nsd && SECOND) ? NQP * DIM * D2 : 0);
      cv.W0 = take(s.rational ? NQP : 0); cv.W1 = take(s.rational ? NQP * DIM : 0); cv.W2 = take((s.rational && SECOND) ? NQP * D2 : 0);
      cv.G = take(((unsigned)F.meta[2] & NEED_G) ? NQP * D2 : 0);
      cv.u = take(fields ? QC * DOF : 0); cv.ut = take(fields ? QC * DOF : 0);
      cv.gu = take((need & NEED_GU) ? QC * DOF * DIM : 0);
      cv.hu = take((need & NEED_HU) ? ((SECOND && !SECOND_S) ? NQP : QC) * DOF * D2 : 0);
      cv.hpart = 0;
      cv.lift = take(out.op == OP_SYSTEM ? QC * DOF * NFS : 0);
      cv.rowbase = take(HASM ? NE : 0); cv.rowid = take(NE); cv.cc = take(NE); cv.pax = take(96); cv.adec = take(NEP / 2); cv.qdec = take((NQP + 1) / 2); cv.nrm = take(NQP * DIM);
      const bool hu_fly = SECOND && !SECOND_S && ((unsigned)F.meta[2] & NEED_HU) != 0;
      cv.sfb = (NE > 64) ? 1 : DOF;
      const int sf_nc = std::max((s.nsd || s.rational) ? DIM + 1 : 0, (hu_fly && (need & NEED_HU)) ? cv.sfb : 0);
      const int sf_need = sf_nc * ((SECOND ? 3 : 2) * nq[0] * na[1] * na[2] + (SECOND ? 6 : 3) * nq[0] * nq[1] * na[2] + (SECOND ? 10 : 4) * NQ);
      cv.boff = 0;
      cv.phi = take(std::max(NPS * QC * NEP, sf_need));
      cv.total = pos; cv.QC = QC; cv.nchunk = nchunk; cv.NEP = NEP;
      lds_bytes = (size_t)pos * sizeof(double);
      if (lds_bytes <= cap) fits = true;
    }
  }
  if (!fits) return 0;   // the generic kernel takes it
  RtcFeatArgs args; memset(&args, 0, sizeof(args));
  args.S = S; args.out = out; args.cv = cv;
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) args.prm.v[i] = s.params[i];
  bool first_touch = HASM && !s.env.no_first_touch;
  for (int d = 0; d < DIM && first_touch; ++d) first_touch = axis_first_touch_ok(s, d);
  args.out.first_touch = first_touch ? 1 : 0;
  if (HASM) {
    if (!first_touch) { if (g->zero_matrix) g->zero_matrix(); }
    else if (s.proc_sizes[0] * s.proc_sizes[1] * s.proc_sizes[2] > 1) zero_neighbour_rows(s, out, g->stream);
  }
  int launches = 0;
  const int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  if (g->timing && g->dom.ev0 && g->dom.launches == 0) (void)hipEventRecord(g->dom.ev0, g->stream);
  for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
      int firstel = -1, count = 0;
      for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (firstel < 0) firstel = e; count++; }
      if (count == 0) { empty = true; break; }
      cr.start[d] = firstel; cr.step[d] = L.p + 1; cr.count[d] = count;
    }
    if (empty) continue;
    const size_t nblocks = (size_t)cr.count[0] * cr.count[1] * cr.count[2];
    args.cr = cr;
    for (hipFunction_t fn : K->func) {
      size_t asz = sizeof(args);
      void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
      HIPCK(hipModuleLaunchKernel(fn, (unsigned)nblocks, 1, 1, (unsigned)(64 * NW), 1, 1, (unsigned)lds_bytes, g->stream, nullptr, cfg));
      launches++;
    }
  }
  // boundary-form passes (IGAElementNextForm, src/petigaelem.c:427-447), as launch_feature_plan makes them for a built-in form: the
  // elements of this rank on a visited face, one point layer at the face, added to what the interior pass left (no first touch)
  for (int bid = 0; bid < 2 * DIM; ++bid) {
    const int ax = bid / 2, sd = bid % 2;
    if (!s.visit[ax][sd]) continue;
    const int eface = sd ? s.elem_sizes[ax] - 1 : 0;
    if (eface < s.elem_start[ax] || eface >= s.elem_start[ax] + s.elem_width[ax]) continue;   // the face is on another rank
    RtcFeatArgs fa = args; fa.out.bid = bid; fa.out.first_touch = 0;
    int nc2[3] = {nc[0], nc[1], nc[2]}; nc2[ax] = 1;
    for (int c2 = 0; c2 < nc2[2]; ++c2) for (int c1 = 0; c1 < nc2[1]; ++c1) for (int c0 = 0; c0 < nc2[0]; ++c0) {
      const int cc[3] = {c0, c1, c2};
      ColorRange cr; bool empty = false;
      for (int d = 0; d < 3; ++d) {
        if (d == ax) { cr.start[d] = eface - s.elem_start[ax]; cr.step[d] = 1; cr.count[d] = 1; continue; }
        const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
        int firstel = -1, count = 0;
        for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (firstel < 0) firstel = e; count++; }
        if (count == 0) { empty = true; break; }
        cr.start[d] = firstel; cr.step[d] = L.p + 1; cr.count[d] = count;
      }
      if (empty) continue;
      fa.cr = cr;
      const size_t nblocks = (size_t)cr.count[0] * cr.count[1] * cr.count[2];
      for (hipFunction_t fn : K->func) {
        size_t asz = sizeof(fa);
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &fa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        HIPCK(hipModuleLaunchKernel(fn, (unsigned)nblocks, 1, 1, (unsigned)(64 * NW), 1, 1, (unsigned)lds_bytes, g->stream, nullptr, cfg));
        launches++;
      }
    }
  }
  g->last_launches = launches;
  if (g->dom.launches == 0) {
    if (g->timing && g->dom.ev1) (void)hipEventRecord(g->dom.ev1, g->stream);
    g->dom.name = "feature_assemble<element, hiprtc>"; g->dom.launches = launches;
    g->dom.elements = (long long)s.elem_width[0] * s.elem_width[1] * s.elem_width[2];
    g->dom.flop_per_element = HASM ? 2048.0 * K->meta[2] * TA * TA * (cv.QC * cv.nchunk / 4) : 0.0;
  }
  if (HASM) g->last_kernel = std::string("feature_assemble<") + F.name + ">(hiprtc,mfma_f64_16x16x4,tiles=" + std::to_string(TA) + "x" + std::to_string(TA) + ",waves=" + char('0' + NW) +
                             ",rowfields/launch=" + char('0' + DOFI) + (DOFI < DOF ? std::string("x") + char('0' + DOF / DOFI) + " fused" : std::string()) + ",chunks=" + std::to_string(cv.nchunk) + ")";
  else g->last_kernel = std::string("feature_assemble<") + F.name + ">(hiprtc,vector only,waves=" + char('0' + NW) + ",chunks=" + std::to_string(cv.nchunk) + ")";
  done = true;
  return 0;
}

// ---- the pencil walk of gram_mfma.hpp for a run-time form (form_pencil): dof 1, first order, matrix integrand on the gradients only
// and symmetric (MAT_TEST_MASK = gradients, MAT_SYMMETRIC), load term on N only (VEC_TEST_MASK = 1), point data = x at most;
// dim 3, uniform degree 2 or 3 with p+1 Gauss points, System / Matrix drivers, axis 0 walkable, with or without a geometry.
// Everything around the kernel -- colours, segments, first touch, the Dirichlet fix-up inside the walk, boundary loads, the
// upper-face-first pass of a multi-rank assembly -- is try_gram_mfma itself.
static bool rtc_pencil_eligible(const Space &s, const RtcForm &F, const OutDev &out) {
  if (s.dim != 3 || F.meta[0] != 1 || F.meta[1] >= 2 || F.meta[4] >= 2 || F.meta[3] > 0 || F.meta[7]) return false;
  if (((unsigned)F.meta[2] & ~NEED_X) != 0u) return false;
  if (((unsigned)F.meta[8] & 0xfu) != 0xeu || !F.meta[9] || ((unsigned)F.meta[10] & 0xfu) != 0x1u) return false;
  if (out.op != OP_SYSTEM && out.op != OP_MATRIX) return false;
  if (s.nsd != 0 && s.nsd != 3) return false;
  const int deg = s.axis[0].p;
  if (deg != 2 && deg != 3) return false;
  for (int d = 0; d < 3; ++d) if (s.axis[d].p != deg || s.basis[d].nqp != deg + 1) return false;
  return true;
}

static int launch_pencil_rtc(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out, bool &done, bool compile_only = false, int sys_only = -1) {
  done = false;
  const Space &s = g->s;
  if (!compile_only && !rtc_pencil_eligible(s, F, out)) return 0;
  const int deg = s.axis[0].p;
  const bool sys = compile_only ? sys_only != 0 : out.op == OP_SYSTEM, ident = s.nsd == 0, rat = s.rational != 0;
  const int key = (sys ? 1 : 0) | (deg << 1) | ((ident ? 1 : 0) << 4) | ((rat ? 1 : 0) << 5);
  std::shared_ptr<RtcFeature> K;
  auto it = F.pencil.find(key);
  if (it != F.pencil.end() && (it->second->module || compile_only)) K = it->second;
  else {
    K.reset(new RtcFeature());
    const std::string x = std::string("igx::form_pencil<") + (sys ? "true" : "false") + ", " + std::to_string(deg) + ", " + (ident ? "true" : "false") + ", " + (rat ? "true" : "false") + ", " + F.name + ">";
    const std::string tail = "template __global__ void " + x + "(igx::SpaceDev, igx::OutDev, igx::PencilArgs, igx::ParamsDev);\n";
    if (int rc = rtc_build(F.source, true, tail, {x}, K->code, K->lowered, true)) return rc;
    if (!compile_only) {
      HIPCK(hipModuleLoadData(&K->module, K->code.data()));
      hipFunction_t fn = nullptr; HIPCK(hipModuleGetFunction(&fn, K->module, K->lowered[0].c_str())); K->func.push_back(fn);
    }
    F.pencil[key] = K;
  }
  if (compile_only) { done = true; return 0; }
  PencilModule mod; memset(&mod.prm, 0, sizeof(mod.prm));
  mod.fn = K->func[0]; mod.name = F.name;
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) mod.prm.v[i] = s.params[i];
  std::function<void()> zero = g->zero_matrix ? g->zero_matrix : std::function<void()>([] {});
  return try_gram_mfma(s, S, out, g->stream, false, g->last_kernel, g->last_launches, g_err, done, g->dom, zero, g->slab_done, &mod, g->face_done);
}

// ---- the sum-factorised vector kernel (vec_sumfact.hpp) for a run-time form: Vector / Function / IFunction in 3-D at nen, nqp <= 4
// per axis; the conditions of vec_sumfact_covers with the struct's constants read from the module
struct RtcVecArgs { SpaceDev S; ParamsDev prm; OutDev out; ColorRange cr; long long nelem; };
static bool rtc_vecsf_eligible(const Space &s, const RtcForm &F, const OutDev &out) {
  if (!s.env.vec_sumfact || s.dim != 3 || F.meta[3] > 0 || F.meta[7]) return false;      // no functionals, no atboundary branch
  if (out.op != OP_VECTOR && out.op != OP_FUNCTION && out.op != OP_IFUNCTION) return false;
  if (s.dof != F.meta[0] || (s.nsd != 0 && s.nsd != 3)) return false;
  for (int d = 0; d < 3; ++d) {
    if (s.basis[d].nen > 4 || s.basis[d].nqp > 4) return false;
    for (int sd = 0; sd < 2; ++sd) { if (s.visit[d][sd]) return false; if (out.op != OP_VECTOR && s.load[d][sd].count) return false; }
  }
  return true;
}
static int launch_vecsf_rtc(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out, bool &done, bool compile_only = false) {
  done = false;
  const Space &s = g->s;
  if (!compile_only && !rtc_vecsf_eligible(s, F, out)) return 0;
  const bool geo = s.nsd > 0 || s.rational;
  bool three = !s.env.no_vec_pairs;      // p <= 2: two elements per wavefront
  for (int d = 0; d < 3; ++d) three = three && s.axis[d].p >= 1 && s.axis[d].p <= 2 && (compile_only || (s.basis[d].nen <= 3 && s.basis[d].nqp <= 3));
  const int key = (geo ? 1 : 0) | (three ? 2 : 0);
  std::shared_ptr<RtcFeature> K;
  auto it = F.vecsf.find(key);
  if (it != F.vecsf.end() && (it->second->module || compile_only)) K = it->second;
  else {
    K.reset(new RtcFeature());
    const std::string x = std::string("igx::vec_sumfact<") + F.name + ", " + (geo ? "true" : "false") + ", " + (three ? "3" : "4") + ">";
    const std::string tail = "template __global__ void " + x + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::ColorRange, long long);\n";
    if (int rc = rtc_build(F.source, true, tail, {x}, K->code, K->lowered, false, true)) return rc;
    if (!compile_only) {
      HIPCK(hipModuleLoadData(&K->module, K->code.data()));
      hipFunction_t fn = nullptr; HIPCK(hipModuleGetFunction(&fn, K->module, K->lowered[0].c_str())); K->func.push_back(fn);
    }
    F.vecsf[key] = K;
  }
  if (compile_only) { done = true; return 0; }
  RtcVecArgs args; memset(&args, 0, sizeof(args));
  args.S = S; args.out = out;
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) args.prm.v[i] = s.params[i];
  int launches = 0;
  const int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
      int first = -1, count = 0;
      for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
      if (count == 0) { empty = true; break; }
      cr.start[d] = first; cr.step[d] = L.p + 1; cr.count[d] = count;
    }
    if (empty) continue;
    args.cr = cr; args.nelem = (long long)cr.count[0] * cr.count[1] * cr.count[2];
    size_t asz = sizeof(args);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    const unsigned want = (unsigned)((args.nelem + (three ? 7 : 3)) / (three ? 8 : 4));      // (a wavefront walks a sequence of elements: vec_sumfact.hpp, round 6)
    HIPCK(hipModuleLaunchKernel(K->func[0], want, 1, 1, 256, 1, 1, 0, g->stream, nullptr, cfg));
    launches++;
  }
  g->last_launches = launches;
  g->last_kernel = std::string("vec_sumfact<") + F.name + ">(hiprtc,vector only: sum factorisation forward and backward, " + (three ? "two elements per wavefront)" : "one wavefront per element)");
  done = true;
  return 0;
}

// ---- the Tangent of a nonlinear scalar struct on the pencil walk (gram_mfma.hpp: state_pencil<P, UserStruct>): the struct declares
// PENCIL_NFEAT / PENCIL_NC / pencil_coef / pencil_trial like FormCahnHilliard and FormBratu (forms.hpp); dof 1, dim 3, uniform degree
// 2 or 3 without a geometry, degree 2 on a mapped one (state_pencil_geo<2, RAT, UserStruct>), Jacobian / IJacobian drivers.
// Everything around the kernel is try_gram_mfma, as for form_pencil.
static int launch_state_rtc(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out, bool &done, bool compile_only = false) {
  done = false;
  const Space &s = g->s;
  const int deg = s.axis[0].p;
  if (s.dim != 3 || (deg != 2 && deg != 3)) return 0;
  // (the struct's constants are read from the loaded module: a compile-only check on a machine without a GPU takes the caller's word)
  if (!compile_only && (F.meta[11] <= 0 || F.meta[0] != 1 || F.meta[3] > 0 || F.meta[7] || !s.env.state_pencil || (out.op != OP_JACOBIAN && out.op != OP_IJACOBIAN))) return 0;
  const bool geo = s.nsd != 0;
  if (geo && (s.nsd != 3 || deg != 2)) return 0;
  const bool pack = deg == 2 && s.env.p2_pack != 0;
  const int key = deg + (geo ? 10 + (s.rational ? 1 : 0) : 0) + (pack ? 100 : 0);      // (12: p = 2 on a polynomial map, 13: on a NURBS map; + 100: packed tiles)
  std::shared_ptr<RtcFeature> K;
  auto it = F.state.find(key);
  if (it != F.state.end() && (it->second->module || compile_only)) K = it->second;
  else {
    K.reset(new RtcFeature());
    // p = 2: the packed-tile kernels, as for the built-in forms (IGX_P2_PACK=0: the layer-pair tiles; the key below tells them apart)
    const std::string x = (deg == 2 && pack) ? (geo ? std::string("igx::state_pencil_geo_k<") + (s.rational ? "true, " : "false, ") + F.name + ">"
                                                    : std::string("igx::state_pencil_k<") + F.name + ">")
                        : geo ? std::string("igx::state_pencil_geo<2, ") + (s.rational ? "true, " : "false, ") + F.name + ">"
                              : std::string("igx::state_pencil<") + std::to_string(deg) + ", " + F.name + ">";
    const std::string tail = "template __global__ void " + x + "(igx::SpaceDev, igx::OutDev, igx::PencilArgs, igx::ParamsDev);\n";
    if (int rc = rtc_build(F.source, true, tail, {x}, K->code, K->lowered, true)) return rc;
    if (!compile_only) {
      HIPCK(hipModuleLoadData(&K->module, K->code.data()));
      hipFunction_t fn = nullptr; HIPCK(hipModuleGetFunction(&fn, K->module, K->lowered[0].c_str())); K->func.push_back(fn);
    }
    F.state[key] = K;
  }
  if (compile_only) { done = true; return 0; }
  PencilModule mod; memset(&mod.prm, 0, sizeof(mod.prm));
  mod.fn = K->func[0]; mod.name = F.name + ",hiprtc"; mod.state = true; mod.state_geo = geo;
  mod.extra_lds = pencil_state_bytes() + (geo ? pencil_sgeo_bytes() - pencil_geo_bytes() : 0);
  mod.pack = pack ? (geo ? 2 : 1) : 0;
  mod.flop_per_element = 2048.0 * F.meta[11] * (deg == 2 ? 7 * (pack ? 4 : 9) : 16 * 16);
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) mod.prm.v[i] = s.params[i];
  std::function<void()> zero = g->zero_matrix ? g->zero_matrix : std::function<void()>([] {});
  return try_gram_mfma(s, S, out, g->stream, false, g->last_kernel, g->last_launches, g_err, done, g->dom, zero, g->slab_done, &mod, g->face_done);
}

// ---- band rows by node layer (block_pencil.hpp) for a run-time struct: constant-coefficient multi-field forms (MAT_PAIR_MASK, 2 or 3
// fields, VEC_ZERO, first order, no atboundary branch), the conditions of bp_form_ok / block_pencil_covers with the struct's
// constants read from the module; colours, segments, first touch, boundary loads and the two-pass assembly are block_pencil_run
struct RtcBlockArgs { SpaceDev S; ParamsDev prm; OutDev out; BlockPencilArgs pa; };
static int launch_block_rtc(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out, bool &done, bool compile_only = false, int sys_only = -1) {
  done = false;
  const Space &s = g->s;
  if (s.dim != 3) return 0;
  if (!compile_only) {
    const int DOF = F.meta[0];
    if (F.meta[12] <= 0 || !F.meta[13] || DOF < 2 || DOF > 3 || F.meta[1] >= 2 || F.meta[4] >= 2 || F.meta[7] || F.meta[3] > 0) return 0;
    if (!block_pencil_covers_space(s, S, out, DOF)) return 0;
  }
  const bool sysk = compile_only ? sys_only != 0 : out.op == OP_SYSTEM;
  auto module_of = [&](bool sys, std::shared_ptr<RtcFeature> &K) -> int {
    auto it = F.block.find(sys ? 1 : 0);
    if (it != F.block.end() && (it->second->module || compile_only)) { K = it->second; return 0; }
    K.reset(new RtcFeature());
    const std::string x = std::string("igx::block_pencil<") + F.name + ", 3, " + (sys ? "true" : "false") + ">";
    const std::string tail = "template __global__ void " + x + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::BlockPencilArgs);\n";
    if (int rc = rtc_build(F.source, true, tail, {x}, K->code, K->lowered, false, false, true)) return rc;
    if (!compile_only) {
      HIPCK(hipModuleLoadData(&K->module, K->code.data()));
      hipFunction_t fn = nullptr; HIPCK(hipModuleGetFunction(&fn, K->module, K->lowered[0].c_str())); K->func.push_back(fn);
    }
    F.block[sys ? 1 : 0] = K;
    return 0;
  };
  std::shared_ptr<RtcFeature> K;
  if (int rc = module_of(sysk, K)) return rc;
  if (compile_only) { done = true; return 0; }
  ParamsDev prm; memset(&prm, 0, sizeof(prm));
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];
  hipFunction_t fn = K->func[0];
  hipStream_t stream = g->stream;
  std::function<void()> zero = g->zero_matrix ? g->zero_matrix : std::function<void()>([] {});
  int lrc = 0;
  const int rc = block_pencil_run(s, S, out, stream, g->last_kernel, g->last_launches, g_err, done, g->dom, zero, g->slab_done, F.meta[0], F.meta[12],
                                  [&](bool, unsigned grid, size_t lds, const BlockPencilArgs &pa) {
                                    RtcBlockArgs a; memset(&a, 0, sizeof(a));
                                    a.S = S; a.prm = prm; a.out = out; a.pa = pa;
                                    size_t asz = sizeof(a);
                                    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
                                    if (lds > (size_t)160 * 1024 || hipModuleLaunchKernel(fn, grid, 1, 1, 512, 1, 1, (unsigned)lds, stream, nullptr, cfg) != hipSuccess) lrc = IGX_ERR_LIB;
                                  });
  if (rc == 0 && lrc) return fail(lrc, "block_pencil: launch of the run-time instantiation failed");
  if (rc == 0 && done) g->last_kernel = std::string("block_pencil<") + F.name + ">(hiprtc,mfma_f64_16x16x4,p=3,dof=" + char('0' + F.meta[0]) + ",band rows by node layer)";
  return rc;
}

static int rtc_generic_launch(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out);
// launch_generic (engine.hip) with the form's constants read from the module instead of from a template parameter
// ---- band rows by node layer with point-dependent coefficients (band_pt.hpp) for a run-time struct: four fields, first order, the
// point coefficients separated from the basis functions (NCOEF, point_coef, mat_c; optionally mat_unit / BAND_NFEAT / BAND_NACC with
// their hooks, forms.hpp: FormNSVMS is the model); Matrix / Jacobian / IJacobian drivers.  The conditions of bpt_form_ok are checked
// on the constants read from the module; colours, segments, first touch and the two-pass assembly are band_pt_run.
struct RtcBandArgs { SpaceDev S; ParamsDev prm; OutDev out; BandArgs pa; };
static int launch_band_rtc(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out, bool &done, bool compile_only = false, int geo_only = -1) {
  done = false;
  const Space &s = g->s;
  if (s.dim != 3) return 0;
  if (!compile_only) {
    if (!F.meta[14] || F.meta[0] != 4 || F.meta[4] >= 2 || F.meta[7] || F.meta[3] > 0 || F.meta[6] || (F.meta[5] & ~(int)(NEED_U | NEED_G))) return 0;
    if (!band_pt_covers_space(s, S, out)) return 0;
  }
  const bool geo = compile_only ? (geo_only & 1) != 0 : s.nsd != 0, rat = compile_only ? (geo_only & 2) != 0 : s.rational != 0;
  const int deg = s.axis[0].p;
  if (deg != 2 && deg != 3) { if (compile_only) return fail(IGX_ERR_SUP, "band_pt needs degree 2 or 3"); return 0; }
  const int key = (geo ? 1 : 0) | (rat ? 2 : 0) | (deg << 2);
  std::shared_ptr<RtcFeature> K;
  auto it = F.band.find(key);
  if (it != F.band.end() && it->second->failed && !compile_only && g->kernel_choice == 0) { g->rtc_note = it->second->why; return 0; }
  if (it != F.band.end() && (it->second->module || compile_only)) K = it->second;
  else {
    K.reset(new RtcFeature());
    const std::string xp = std::string("igx::band_points<") + F.name + ", " + std::to_string(deg) + ">";
    const std::string xb = std::string("igx::band_pt<") + F.name + ", " + (geo ? "true" : "false") + ", " + (rat ? "true" : "false") + ", " + std::to_string(deg) + ">";
    const std::string tail = "static_assert(igx::bpt_form_ok<" + F.name + ">(), \"band_pt: four fields, first order, NCOEF / point_coef / mat_c, no atboundary branch\");\n"
                             "__device__ int igx_band_meta[4] = {igx::bpt_rec<" + F.name + ">(), igx::bpt_products<" + F.name + ">(), igx::band_nacc_of<" + F.name + ">::own ? 1 : 0, 0};\n"
                             // the struct's own guard on the parameters (a struct with BAND_NACC hooks declares band_params_ok __host__ __device__)
                             "__device__ double igx_band_prm[igx::MAXPARAM]; __device__ int igx_band_ok;\n"
                             "template <class F> __device__ int igx_band_guard_of(const double *prm) { if constexpr (igx::band_nacc_of<F>::own) return F::band_params_ok(prm) ? 1 : 0; else return 1; }\n"
                             "extern \"C\" __global__ void igx_band_guard() { igx_band_ok = igx_band_guard_of<" + F.name + ">(igx_band_prm); }\n"
                             "template __global__ void " + xp + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::BandArgs);\n"
                             "template __global__ void " + xb + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::BandArgs);\n";
    if (int rc = rtc_build(F.source, true, tail, {xp, xb}, K->code, K->lowered, false, false, false, true)) {
      // Under the automatic choice a struct whose band-row instantiation does not compile (the usual reason: a band_params_ok
      // that is not __host__ __device__, include/petiga_amd.h) is a struct the band-row kernel does not cover: the feature kernel
      // takes the assembly and the kernel name carries the reason.  Asked for by name (IGXSetKernel(4), IGXCheckFormSource) it fails.
      if (compile_only || g->kernel_choice != 0) return rc;
      K->failed = true;
      K->why = "band_pt<" + F.name + "> did not compile (is band_params_ok declared __host__ __device__?): " + g_err.substr(0, 400);
      F.band[key] = K; g->rtc_note = K->why;
      return 0;
    }
    if (!compile_only) {
      HIPCK(hipModuleLoadData(&K->module, K->code.data()));
      for (int k = 0; k < 2; ++k) { hipFunction_t fn = nullptr; HIPCK(hipModuleGetFunction(&fn, K->module, K->lowered[k].c_str())); K->func.push_back(fn); }
      hipDeviceptr_t p = nullptr; size_t n = 0;
      HIPCK(hipModuleGetGlobal(&p, &n, K->module, "igx_band_meta"));
      if (n != sizeof(K->meta)) return fail(IGX_ERR_LIB, "unexpected igx_band_meta size");
      HIPCK(hipMemcpy(K->meta, p, sizeof(K->meta), hipMemcpyDeviceToHost));
    }
    F.band[key] = K;
  }
  if (compile_only) { done = true; return 0; }
  ParamsDev prm; memset(&prm, 0, sizeof(prm));
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) prm.v[i] = s.params[i];
  hipStream_t stream = g->stream;
  // The struct's guard on its parameters, as the built-in path has it (band_pt.hpp: Form::band_params_ok): band_coef / band_finish
  // may divide by a parameter (FormNSVMS: 1 / nu), and a struct that says no stays on the feature kernel.  Evaluated by the
  // module's one-lane kernel whenever the parameters differ from the ones it was last asked about.
  if (K->meta[2]) {
    const std::vector<double> now(prm.v, prm.v + MAXPARAM);
    if (!K->guard_known || now != K->guard_prm) {
      hipFunction_t gf = nullptr; hipDeviceptr_t pp = nullptr, po = nullptr; size_t n = 0; int ok = 0;
      HIPCK(hipModuleGetFunction(&gf, K->module, "igx_band_guard"));
      HIPCK(hipModuleGetGlobal(&pp, &n, K->module, "igx_band_prm"));
      if (n != sizeof(prm.v)) return fail(IGX_ERR_LIB, "unexpected igx_band_prm size");
      HIPCK(hipModuleGetGlobal(&po, &n, K->module, "igx_band_ok"));
      HIPCK(hipMemcpyAsync(pp, prm.v, sizeof(prm.v), hipMemcpyHostToDevice, stream));
      HIPCK(hipModuleLaunchKernel(gf, 1, 1, 1, 1, 1, 1, 0, stream, nullptr, nullptr));
      HIPCK(hipMemcpyAsync(&ok, po, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIPCK(hipStreamSynchronize(stream));
      K->guard_known = true; K->guard_ok = ok != 0; K->guard_prm = now;
    }
    if (!K->guard_ok) return 0;       // done stays false: the caller goes on to the feature kernel
  }
  int lrc = 0;
  std::function<void()> zero = g->zero_matrix ? g->zero_matrix : std::function<void()>([] {});
  const int rc = band_pt_run(s, S, out, stream, g->last_kernel, g->last_launches, g_err, done, g->dom, zero, g->slab_done, K->meta[0], K->meta[1],
                             [&](bool points, unsigned grid, size_t lds, bool, bool, int, const BandArgs &pa) {
                               RtcBandArgs a; memset(&a, 0, sizeof(a));
                               a.S = S; a.prm = prm; a.out = out; a.pa = pa;
                               size_t asz = sizeof(a);
                               void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
                               if (hipModuleLaunchKernel(K->func[points ? 0 : 1], grid, 1, 1, 256, 1, 1, (unsigned)lds, stream, nullptr, cfg) != hipSuccess) lrc = IGX_ERR_LIB;
                             });
  if (rc == 0 && lrc) return fail(lrc, "band_pt: launch of the run-time instantiation failed");
  if (rc == 0 && done) g->last_kernel = std::string("band_pt<") + F.name + ">(hiprtc,mfma_f64_16x16x4,p=" + std::to_string(deg) + ",dof=4,band rows by node layer,point records)";
  return rc;
}

static int launch_generic_rtc(IGX g, const SpaceDev &S, const OutDev &out) {
  Space &s = g->s;
  g->rtc_note.clear();
  if (!g->rtc || g->rtc->dim != s.dim) {
    std::shared_ptr<RtcForm> f;
    if (int rc = rtc_compile(g, g->rtc_source, g->rtc_name, s.dim, f)) return rc;
    g->rtc = f;
  }
  RtcForm &F = *g->rtc;
  if (int rc = rtc_load(g, F)) return rc;
  const int DOF = F.meta[0];
  if (F.meta[3] > 0) return fail(IGX_ERR_SUP, "a struct with NSCALAR is a functional: IGXComputeScalarSource");
  if (s.dof != DOF) return fail(IGX_ERR_ARG_WRONG, "form does not match the number of fields (dof)");
  // a struct of ORDER 3, or one that reads the property array / the point's shape table / third derivatives of the state: the general kernel (as launch_generic)
  if (F.meta[1] >= 3 || ((unsigned)F.meta[2] & (NEED_PROP | NEED_D3U | NEED_MAPX)) || (s.nsd && s.nsd != s.dim)) {      // (... or a geometry with nsd != dim)
    if (g->kernel_choice != 0 && g->kernel_choice != 1) return fail(IGX_ERR_SUP, "a form of order 3 or one that reads the property array runs on the general kernel only");
    if (((unsigned)F.meta[2] & NEED_PROP) && !S.npd) return fail(IGX_ERR_ARG_WRONGSTATE, "No property set");
    if (((unsigned)F.meta[2] & NEED_MAPX) && !s.nsd) return fail(IGX_ERR_ARG_WRONGSTATE, "No geometry set");
    if (F.meta[1] >= 3 && s.order < 3) return fail(IGX_ERR_ARG_WRONGSTATE, "the form reads third derivatives (p->shape[3]): call IGASetOrder(iga,3) first");
    if (g->zero_matrix) g->zero_matrix();
    return rtc_generic_launch(g, F, S, out);
  }
  if (g->kernel_choice == 0) {   // vector-only drivers in 3-D: sum factorisation both ways (vec_sumfact.hpp)
    bool done = false;
    if (int rc = launch_vecsf_rtc(g, F, S, out, done)) return rc;
    if (done) return 0;
  }
  if ((g->kernel_choice == 0 && s.axis[0].p == 3) || g->kernel_choice == 4) {   // four-field structs with separated point coefficients: band rows by node layer (p = 2 on request)
    bool done = false;
    if (int rc = launch_band_rtc(g, F, S, out, done)) return rc;
    if (done) return 0;
  }
  if (g->kernel_choice == 0 || g->kernel_choice == 4) {   // constant-coefficient multi-field structs: band rows by node layer
    bool done = false;
    if (int rc = launch_block_rtc(g, F, S, out, done)) return rc;
    if (done) return 0;
    if (g->kernel_choice == 4) return fail(IGX_ERR_SUP, "the band-row kernels do not cover this run-time form / configuration (block_pencil: MAT_PAIR_MASK, 2 or 3 fields, VEC_ZERO, 3-D, p = 3, identity geometry, System / Matrix driver; band_pt: 4 fields, NCOEF / point_coef / mat_c, 3-D, p = 3, matrix-only driver)");
  }
  if (g->kernel_choice == 0 || g->kernel_choice == 2) {   // Tangents of scalar structs that opted in: the pencil walk with the state
    bool done = false;
    if (int rc = launch_state_rtc(g, F, S, out, done)) return rc;
    if (done) return 0;
  }
  if (g->kernel_choice == 0 || g->kernel_choice == 2) {   // scalar symmetric gradient forms: the pencil walk (combine before write)
    bool done = false;
    if (int rc = launch_pencil_rtc(g, F, S, out, done)) return rc;
    if (done) return 0;
    if (g->kernel_choice == 2) return fail(IGX_ERR_SUP, "the pencil kernel does not cover this run-time form / configuration (dof 1, gradients only, MAT_SYMMETRIC, VEC_TEST_MASK = 1, dim 3, p = 2 or 3)");
  }
  if (g->kernel_choice != 1) {   // the dense contraction on the matrix cores when the case is covered (as launch_generic does)
    bool done = false;
    if (int rc = launch_feature_rtc(g, F, S, out, done)) return rc;
    if (done) return 0;
    if (g->kernel_choice == 3) return fail(IGX_ERR_SUP, "the feature-GEMM kernel does not cover this case (needs dim >= 2 and nen <= 64; nen <= 128 with dof <= 2 and nen <= 256 with dof 1 in 3-D, no MAT_PAIR_MASK there)");
  }
  if (g->zero_matrix) g->zero_matrix();
  return rtc_generic_launch(g, F, S, out);
}

// the point-form kernel of a run-time struct: matrix / vector forms (coloured sweeps + boundary-form passes) and functionals
// (NSCALAR > 0: one sweep, a row of partial sums per element, as launch_generic does for the built-in ones)
static int rtc_generic_launch(IGX g, RtcForm &F, const SpaceDev &S, const OutDev &out) {
  Space &s = g->s;
  const int DOF = F.meta[0], DIM = s.dim, NS = F.meta[3]; const bool SECOND = F.meta[1] >= 2, THIRD = F.meta[1] >= 3; const unsigned NEED = (unsigned)F.meta[2];
  const int D2 = DIM * DIM, D3 = D2 * DIM, NF = 1 + DIM + (SECOND ? D2 : 0) + (THIRD ? D3 : 0);
  const bool fields = (NEED & (NEED_U | NEED_UT | NEED_GU | NEED_HU | NEED_D3U)) != 0;
  int nq[3], na[3]; int NQ = 1, NE = 1;
  for (int d = 0; d < 3; ++d) { nq[d] = s.basis[d].nqp; na[d] = s.basis[d].nen; NQ *= nq[d]; NE *= na[d]; }
  Carve cv; int pos = 0;
  auto take = [&](int n) { int o = pos; pos += (n + 1) & ~1; return o; };
  for (int d = 0; d < 3; ++d) { cv.t1d[d] = take(nq[d] * na[d] * NDER); cv.w1d[d] = take(nq[d]); }
  const int nsd = s.nsd ? s.nsd : DIM; const bool emb = s.nsd && s.nsd != DIM;
  cv.gX = take(NE * nsd); cv.gW = take(NE); cv.Ue = take(NE * DOF); cv.Ve = take(NE * DOF);
  cv.ufix = take(NE * DOF); cv.fixval = take(NE * DOF); cv.fixflag = take(NE * DOF); cv.flux = take(NE * DOF);
  cv.JW = take(NQ); cv.xq = take(NQ * nsd); cv.E1 = take((s.nsd && !emb) ? NQ * D2 : 0); cv.E2 = take((s.nsd && !emb && SECOND) ? NQ * DIM * D2 : 0);
  cv.W0 = take(s.rational ? NQ : 0); cv.W1 = take(s.rational ? NQ * DIM : 0); cv.W2 = take((s.rational && SECOND) ? NQ * D2 : 0);
  cv.G = take((NEED & NEED_G) ? NQ * DIM * nsd : 0);
  cv.X1m = take((NEED & NEED_MAPX) ? NQ * nsd * DIM : 0); cv.X2m = take(((NEED & NEED_MAPX) && SECOND) ? NQ * nsd * D2 : 0);
  cv.E3 = take((s.nsd && !emb && THIRD) ? NQ * DIM * D3 : 0); cv.W3 = take((s.rational && THIRD) ? NQ * D3 : 0);
  cv.d3u = take((THIRD && (NEED & NEED_D3U)) ? NQ * DOF * D3 : 0); cv.gA = take(NE * S.npd);
  cv.u = take(fields ? NQ * DOF : 0); cv.ut = take(fields ? NQ * DOF : 0);
  cv.gu = take((NEED & NEED_GU) ? NQ * DOF * DIM : 0); cv.hu = take((NEED & NEED_HU) ? NQ * DOF * D2 : 0);
  cv.lift = take(NS > 0 ? NQ * NS : (out.op == OP_SYSTEM ? NQ * DOF * NF : 0));
  cv.nrm = take(NQ * nsd);
  const size_t phi_doubles = (size_t)NQ * NE * NF;
  const size_t lds_limit = 64 * 1024;   // as launch_generic: beyond it Phi goes to an HBM slice and two workgroups share a CU
  const bool phi_in_lds = ((size_t)pos + phi_doubles) * sizeof(double) <= lds_limit;
  if (phi_in_lds) cv.phi = take((int)phi_doubles); else cv.phi = -1;
  cv.total = pos;
  const size_t lds_bytes = (size_t)pos * sizeof(double);
  if (lds_bytes > lds_limit) return fail(IGX_ERR_SUP, "element work set of the run-time form exceeds 64 KiB of LDS");
  RtcArgs args; memset(&args, 0, sizeof(args));
  args.S = S; args.out = out; args.cv = cv; args.phi_stride = phi_doubles;
  for (size_t i = 0; i < s.params.size() && i < MAXPARAM; ++i) args.prm.v[i] = s.params[i];
  const size_t scratch_cap = (size_t)2 << 30;
  size_t max_blocks = phi_in_lds ? ((size_t)1 << 30) : scratch_cap / (phi_doubles * sizeof(double));
  if (max_blocks < 1) max_blocks = 1;
  int launches = 0;
  int nc[3] = {s.lay[0].ncolors, s.lay[1].ncolors, s.lay[2].ncolors};
  if (NS > 0) nc[0] = nc[1] = nc[2] = 1;   // nothing is scattered: every element in one sweep
  int64_t elem_base = 0;
  auto sweep = [&](ColorRange cr, int bid) -> int {      // one colour (or face layer), split along axis 2 under the scratch cap
    const size_t per2 = (size_t)cr.count[0] * cr.count[1];
    const int chunk2 = (int)std::max<size_t>(1, std::min<size_t>((size_t)cr.count[2], max_blocks / std::max<size_t>(per2, 1)));
    if (!phi_in_lds && per2 > max_blocks) return fail(IGX_ERR_SUP, "scratch too small for one element layer");
    for (int k0 = 0; k0 < cr.count[2]; k0 += chunk2) {
      ColorRange sub = cr; sub.start[2] = cr.start[2] + k0 * cr.step[2]; sub.count[2] = std::min(chunk2, cr.count[2] - k0);
      const size_t nblocks = per2 * sub.count[2];
      if (!phi_in_lds) {
        const size_t need = nblocks * phi_doubles * sizeof(double);
        if (g->scratch.bytes < need) { HIPCK(hipStreamSynchronize(g->stream)); if (g->scratch.alloc(need)) return fail(IGX_ERR_MEM, "scratch allocation failed"); }
      }
      args.cr = sub; args.phi_global = g->scratch.as<double>(); args.out.bid = bid; args.out.elem_base = elem_base; elem_base += (int64_t)nblocks;
      size_t asz = sizeof(args);
      void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
      HIPCK(hipModuleLaunchKernel(F.func, (unsigned)nblocks, 1, 1, 256, 1, 1, (unsigned)lds_bytes, g->stream, nullptr, cfg));
      launches++;
    }
    return 0;
  };
  for (int c2 = 0; c2 < nc[2]; ++c2) for (int c1 = 0; c1 < nc[1]; ++c1) for (int c0 = 0; c0 < nc[0]; ++c0) {
    const int cc[3] = {c0, c1, c2};
    ColorRange cr; bool empty = false;
    for (int d = 0; d < 3; ++d) {
      const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
      int first = -1, count = 0;
      for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
      if (NS > 0) { first = 0; count = nel; }
      if (count == 0) { empty = true; break; }
      cr.start[d] = first; cr.step[d] = (NS > 0) ? 1 : L.p + 1; cr.count[d] = count;
    }
    if (empty) continue;
    if (int rc = sweep(cr, -1)) return rc;
  }
  // boundary-form passes (IGAElementNextForm, src/petigaelem.c:427-447): the elements of this rank on a visited face, one point
  // layer at the face; a struct with an atboundary branch supplies bmat / bvec (HAS_BOUNDARY), any other is integrated over the face
  for (int bid = 0; bid < 2 * DIM; ++bid) {
    const int ax = bid / 2, sd = bid % 2;
    if (!s.visit[ax][sd]) continue;
    const int eface = sd ? s.elem_sizes[ax] - 1 : 0;
    if (eface < s.elem_start[ax] || eface >= s.elem_start[ax] + s.elem_width[ax]) continue;   // the face is on another rank
    int nc2[3] = {nc[0], nc[1], nc[2]}; nc2[ax] = 1;
    for (int c2 = 0; c2 < nc2[2]; ++c2) for (int c1 = 0; c1 < nc2[1]; ++c1) for (int c0 = 0; c0 < nc2[0]; ++c0) {
      const int cc[3] = {c0, c1, c2};
      ColorRange cr; bool empty = false;
      for (int d = 0; d < 3; ++d) {
        if (d == ax) { cr.start[d] = eface - s.elem_start[ax]; cr.step[d] = 1; cr.count[d] = 1; continue; }
        const AxisLayout &L = s.lay[d]; const int nel = s.elem_width[d];
        int first = -1, count = 0;
        for (int e = 0; e < nel; ++e) if (L.color[e] == cc[d]) { if (first < 0) first = e; count++; }
        if (NS > 0) { first = 0; count = nel; }
        if (count == 0) { empty = true; break; }
        cr.start[d] = first; cr.step[d] = (NS > 0) ? 1 : L.p + 1; cr.count[d] = count;
      }
      if (empty) continue;
      if (int rc = sweep(cr, bid)) return rc;
    }
  }
  g->last_launches = launches;
  g->last_kernel = std::string("generic_assemble<") + F.name + "> (hiprtc," + (phi_in_lds ? "phi=LDS" : "phi=HBM") + ")";
  return 0;
}

extern "C" int IGXSetFormSource(IGX g, const char *source, const char *struct_name, const double params[], int nparams) {
  NEEDIGA(g);
  if (!source || !struct_name || !*struct_name) return fail(IGX_ERR_ARG_WRONG, "null source / struct name");
  if (nparams < 0 || nparams > MAXPARAM || (nparams && !params)) return fail(IGX_ERR_ARG_OUTOFRANGE, "bad parameter list");
  if (g->s.dim < 1) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGASetDim() first");
  std::shared_ptr<RtcForm> f;
  if (int rc = rtc_compile(g, source, struct_name, g->s.dim, f)) return rc;     // compile errors surface here, with the log
  g->rtc = f; g->rtc_source = source; g->rtc_name = struct_name;
  g->s.form = IGX_FORM_SOURCE; g->s.params.assign(params ? params : nullptr, params ? params + nparams : nullptr);
  return 0;
}

// IGAComputeScalar (src/petigacomp.c:35-98) with the user's point functional given as source: a struct with DOF, ORDER, NEED,
// NSCALAR = n and scalar(p, S) -- the un-weighted integrand values at a point (IGAFormScalar, include/petiga.h:188-191); the
// engine multiplies by JW and sums over the rank's elements and, for visited faces, over their boundary passes (p.atboundary).
extern "C" int IGXComputeScalarSource(IGX g, IGXVec U, const char *source, const char *struct_name, const double params[], int nparams, int n, double S[]) {
  NEEDIGA(g);
  if (!source || !struct_name || !*struct_name) return fail(IGX_ERR_ARG_WRONG, "null source / struct name");
  if (!S || n < 1) return fail(IGX_ERR_ARG_WRONG, "null result array");
  if (nparams < 0 || nparams > MAXPARAM || (nparams && !params)) return fail(IGX_ERR_ARG_OUTOFRANGE, "bad parameter list");
  if (int rc = ensure_device(g)) return rc;
  Space &s = g->s;
  if (U && U->iga != g) return fail(IGX_ERR_ARG_WRONG, "state vector created by another IGX");
  // (one compiled functional is kept per IGX: the same source and struct are not compiled twice)
  if (!g->rtc_scalar || g->rtc_scalar->source != source || g->rtc_scalar->name != struct_name || g->rtc_scalar->dim != s.dim) {
    std::shared_ptr<RtcForm> f;
    if (int rc = rtc_compile(g, source, struct_name, s.dim, f)) return rc;
    g->rtc_scalar = f;
  }
  RtcForm &F = *g->rtc_scalar;
  if (int rc = rtc_load(g, F)) return rc;
  const int ns = F.meta[3];
  if (ns < 1) return fail(IGX_ERR_ARG_WRONG, "the struct is not a functional (no NSCALAR)");
  if (n != ns) return fail(IGX_ERR_ARG_WRONG, "this functional returns " + std::to_string(ns) + " scalars");
  if (F.meta[0] != s.dof) return fail(IGX_ERR_ARG_WRONG, "functional does not match the number of fields (dof)");
  if (((unsigned)F.meta[2] & (NEED_U | NEED_GU | NEED_HU)) && !U) return fail(IGX_ERR_ARG_WRONG, "the functional reads the state: null vector");
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
  const std::vector<double> keep = s.params;
  s.params.assign(params ? params : nullptr, params ? params + nparams : nullptr);
  const int rc = rtc_generic_launch(g, F, Sd, out);
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

// Compile-only check of a run-time form against the kernels the drivers would launch for the axes set so far (no GPU needed):
// the point-form kernel was compiled by IGXSetFormSource; this adds the matrix-core kernel (feature_mfma.hpp) in the wave
// layout of the current degree, for the matrix drivers (with_matrix != 0) or the vector-only ones.
extern "C" int IGXCheckFormSource(IGX g, int with_matrix, int gram) {
  NEEDIGA(g);
  if (g->s.form != IGX_FORM_SOURCE || !g->rtc) return fail(IGX_ERR_ARG_WRONGSTATE, "Must call IGXSetFormSource() first");
  const Space &s = g->s;
  if (gram == 6) {           // band_points + band_pt of a four-field struct with separated point coefficients: compile only (no geometry and NURBS)
    bool done = false; OutDev o; memset(&o, 0, sizeof(o)); SpaceDev Sd; memset(&Sd, 0, sizeof(Sd));
    if (int rc = launch_band_rtc(g, *g->rtc, Sd, o, done, true, 0)) return rc;
    if (!done) return fail(IGX_ERR_SUP, "band_pt needs dim 3");
    return launch_band_rtc(g, *g->rtc, Sd, o, done, true, 3);
  }
  if (gram == 5) {           // block_pencil of a constant-coefficient multi-field struct (System and Matrix driver): compile only
    bool done = false; OutDev o; memset(&o, 0, sizeof(o)); SpaceDev Sd; memset(&Sd, 0, sizeof(Sd));
    if (int rc = launch_block_rtc(g, *g->rtc, Sd, o, done, true, 1)) return rc;
    if (!done) return fail(IGX_ERR_SUP, "block_pencil needs dim 3");
    return launch_block_rtc(g, *g->rtc, Sd, o, done, true, 0);
  }
  if (gram == 4) {           // state_pencil of a struct with the PENCIL_* hooks, for the current degree: compile only
    bool done = false; OutDev o; memset(&o, 0, sizeof(o)); SpaceDev Sd; memset(&Sd, 0, sizeof(Sd));
    if (int rc = launch_state_rtc(g, *g->rtc, Sd, o, done, true)) return rc;
    return done ? 0 : fail(IGX_ERR_SUP, "state_pencil needs dim 3, dof 1, degree 2 or 3 (2 on a mapped geometry) and a struct with PENCIL_NFEAT / PENCIL_NC / pencil_coef / pencil_trial");
  }
  if (gram == 3) {           // the sum-factorised vector kernel (vec_sumfact) of the struct, with and without a geometry: compile only
    if (s.dim != 3) return fail(IGX_ERR_SUP, "the sum-factorised vector kernel needs dim 3");
    bool done = false; OutDev o; memset(&o, 0, sizeof(o)); SpaceDev Sd; memset(&Sd, 0, sizeof(Sd));
    return launch_vecsf_rtc(g, *g->rtc, Sd, o, done, true);
  }
  if (gram == 2) {           // the pencil walk's instantiation (form_pencil) for the current degree / geometry: compile only
    if (s.dim != 3 || (s.axis[0].p != 2 && s.axis[0].p != 3)) return fail(IGX_ERR_SUP, "the pencil walk needs dim 3 and degree 2 or 3");
    bool done = false; OutDev o; memset(&o, 0, sizeof(o));
    SpaceDev Sd; memset(&Sd, 0, sizeof(Sd));
    if (int rc = launch_pencil_rtc(g, *g->rtc, Sd, o, done, true, 1)) return rc;      // System driver
    return launch_pencil_rtc(g, *g->rtc, Sd, o, done, true, 0);                        // Matrix driver
  }
  if (s.dim < 2) return 0;   // dim 1: the point-form kernel only
  int NE = 1;
  for (int d = 0; d < s.dim; ++d) { if (s.axis[d].p < 1) return fail(IGX_ERR_ARG_WRONGSTATE, "set the axes (degrees) first"); NE *= s.axis[d].p + 1; }
  if (NE > 256 || (NE > 64 && (s.dim != 3 || gram || s.dof > 2)) || (NE > 128 && s.dof > 1)) return 0;
  int TA, NW, DOFI; rtc_feature_layout(NE, s.dof, gram != 0, with_matrix != 0, TA, NW, DOFI);
  std::shared_ptr<RtcFeature> K;
  return rtc_feature_module(g, *g->rtc, s.dim, s.dof, TA, NW, DOFI, with_matrix != 0, false, K);
}
