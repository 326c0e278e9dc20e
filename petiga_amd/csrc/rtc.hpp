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
};
static HiprtcApi &hiprtc_api() { static HiprtcApi a; return a; }

static int load_hiprtc(std::string &err) {
  HiprtcApi &a = hiprtc_api();
  if (a.h) return 0;
  const char *env = getenv("IGX_HIPRTC_LIB");
  const char *names[] = {env, "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
  for (int pass = 0; pass < 2 && !a.h; ++pass)
    for (const char *n : names) { if (!n || !*n) continue; a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)); if (a.h) break; }
  if (!a.h) { err = std::string("cannot load libhiprtc.so: ") + (dlerror() ? dlerror() : "not found"); return IGX_ERR_LIB; }
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
  if (!a.Create || !a.AddName || !a.Compile || !a.LogSize || !a.Log || !a.Lowered || !a.CodeSize || !a.Code || !a.Destroy) { err = "libhiprtc.so lacks the expected API"; a = HiprtcApi(); return IGX_ERR_LIB; }
  return 0;
}

}  // namespace

// one compiled user form for one dimension: code object + what the host-side launcher must know about the struct
struct RtcForm {
  std::string name, source, lowered;
  int dim = 0;
  std::vector<char> code;
  int meta[4] = {0, 0, 0, 0};          // DOF, ORDER, NEED, NSCALAR (read from the module)
  hipModule_t module = nullptr; hipFunction_t func = nullptr;
  ~RtcForm() { if (module) (void)hipModuleUnload(module); }
};

static int rtc_compile(IGX g, const std::string &source, const std::string &name, int dim, std::shared_ptr<RtcForm> &out) {
  std::string e; if (int rc = load_hiprtc(e)) return fail(rc, e);
  HiprtcApi &a = hiprtc_api();
  std::string src;
  src.reserve(source.size() + 200000);
  src += "#define IGX_RTC 1\n";
  src += kRtcSrc_igx; src += "\n"; src += kRtcSrc_forms; src += "\n"; src += kRtcSrc_generic; src += "\n";
  src += "using namespace igx;\n#line 1 \"user_form.hip\"\n";
  src += source;
  src += "\n// what the host-side launcher reads back\n__device__ int igx_user_meta[4] = {" + name + "::DOF, " + name + "::ORDER, (int)" + name + "::NEED, igx::nscalar_of<" + name + ">::v};\n";
  const std::string expr = "igx::generic_assemble<" + name + ", " + std::to_string(dim) + ">";
  src += "template __global__ void " + expr + "(igx::SpaceDev, igx::ParamsDev, igx::OutDev, igx::ColorRange, igx::Carve, double *, size_t);\n";
  void *prog = nullptr;
  if (a.Create(&prog, src.c_str(), "igx_user_form.hip", 0, nullptr, nullptr) != 0) return fail(IGX_ERR_LIB, "hiprtcCreateProgram failed");
  (void)a.AddName(prog, expr.c_str());
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
  const int rc = a.Compile(prog, 4, opts);
  if (rc != 0) {
    size_t n = 0; (void)a.LogSize(prog, &n); std::string log(n, '\0'); if (n) (void)a.Log(prog, &log[0]);
    (void)a.Destroy(&prog);
    return fail(IGX_ERR_USER, "the form source does not compile:\n" + log);
  }
  std::shared_ptr<RtcForm> f(new RtcForm());
  const char *low = nullptr;
  if (a.Lowered(prog, expr.c_str(), &low) != 0 || !low) { (void)a.Destroy(&prog); return fail(IGX_ERR_LIB, "hiprtcGetLoweredName failed"); }
  f->lowered = low;
  size_t cs = 0; (void)a.CodeSize(prog, &cs); f->code.resize(cs); (void)a.Code(prog, f->code.data());
  (void)a.Destroy(&prog);
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

// launch_generic (engine.hip) with the form's constants read from the module instead of from a template parameter
static int launch_generic_rtc(IGX g, const SpaceDev &S, const OutDev &out) {
  Space &s = g->s;
  if (!g->rtc || g->rtc->dim != s.dim) {
    std::shared_ptr<RtcForm> f;
    if (int rc = rtc_compile(g, g->rtc_source, g->rtc_name, s.dim, f)) return rc;
    g->rtc = f;
  }
  RtcForm &F = *g->rtc;
  if (int rc = rtc_load(g, F)) return rc;
  const int DOF = F.meta[0], DIM = s.dim; const bool SECOND = F.meta[1] >= 2; const unsigned NEED = (unsigned)F.meta[2];
  if (F.meta[3] > 0) return fail(IGX_ERR_SUP, "run-time forms are matrix / vector forms (no scalar functionals)");
  if (s.dof != DOF) return fail(IGX_ERR_ARG_WRONG, "form does not match the number of fields (dof)");
  for (int a = 0; a < s.dim; ++a) for (int sd = 0; sd < 2; ++sd) if (s.visit[a][sd]) return fail(IGX_ERR_SUP, "boundary-form passes are not available for run-time forms");
  const int NF = SECOND ? 1 + DIM + DIM * DIM : 1 + DIM, D2 = DIM * DIM;
  if (g->zero_matrix) g->zero_matrix();
  const bool fields = (NEED & (NEED_U | NEED_UT | NEED_GU | NEED_HU)) != 0;
  int nq[3], na[3]; int NQ = 1, NE = 1;
  for (int d = 0; d < 3; ++d) { nq[d] = s.basis[d].nqp; na[d] = s.basis[d].nen; NQ *= nq[d]; NE *= na[d]; }
  Carve cv; int pos = 0;
  auto take = [&](int n) { int o = pos; pos += (n + 1) & ~1; return o; };
  for (int d = 0; d < 3; ++d) { cv.t1d[d] = take(nq[d] * na[d] * NDER); cv.w1d[d] = take(nq[d]); }
  cv.gX = take(NE * DIM); cv.gW = take(NE); cv.Ue = take(NE * DOF); cv.Ve = take(NE * DOF);
  cv.ufix = take(NE * DOF); cv.fixval = take(NE * DOF); cv.fixflag = take(NE * DOF); cv.flux = take(NE * DOF);
  cv.JW = take(NQ); cv.xq = take(NQ * DIM); cv.E1 = take(s.nsd ? NQ * D2 : 0); cv.E2 = take((s.nsd && SECOND) ? NQ * DIM * D2 : 0);
  cv.W0 = take(s.rational ? NQ : 0); cv.W1 = take(s.rational ? NQ * DIM : 0); cv.W2 = take((s.rational && SECOND) ? NQ * D2 : 0);
  cv.G = take((NEED & NEED_G) ? NQ * D2 : 0);
  cv.u = take(fields ? NQ * DOF : 0); cv.ut = take(fields ? NQ * DOF : 0);
  cv.gu = take((NEED & NEED_GU) ? NQ * DOF * DIM : 0); cv.hu = take((NEED & NEED_HU) ? NQ * DOF * D2 : 0);
  cv.lift = take(out.op == OP_SYSTEM ? NQ * DOF * NF : 0);
  cv.nrm = take(NQ * DIM);
  const size_t phi_doubles = (size_t)NQ * NE * NF;
  const size_t lds_limit = 64 * 1024;      // module kernels keep to the default dynamic-LDS limit; Phi spills to HBM beyond it
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
      args.cr = sub; args.phi_global = g->scratch.as<double>();
      size_t asz = sizeof(args);
      void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
      HIPCK(hipModuleLaunchKernel(F.func, (unsigned)nblocks, 1, 1, 256, 1, 1, (unsigned)lds_bytes, g->stream, nullptr, cfg));
      launches++;
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
