/*
 * petiga_amd.h -- C ABI of libpetiga_amd.so: the MI355X-native IGA element-assembly engine.
 *
 * This is the drop-in boundary for ONE path of dalcinl/PetIGA: the element loop behind
 *   IGAComputeSystem / IGAComputeMatrix / IGAComputeVector      (src/petigaksp.c:33,79,149)
 *   IGAComputeFunction / IGAComputeJacobian                     (src/petigasnes.c:23,82)
 *   IGAComputeIFunction / IGAComputeIJacobian                   (src/petigats.c:23,92)
 * i.e. IGANextElement -> IGAElementBuildTabulation -> point callback -> IGAPointAddMat ->
 * IGAElementFixSystem -> IGAElementAssembleMat, re-implemented as HIP kernels for gfx950.
 *
 * Everything is plain C: opaque handles, int/double scalars, raw pointers and sizes.  No PETSc,
 * torch or C++ types cross this boundary.  All citations are file:line of the reference tree.
 *
 * Two ways in:
 *   (1) IGXCreate + IGXAxis* + IGXSetUp ...   mirrors the PetIGA user API (include/petiga.h),
 *       so a test written against PetIGA reads the same against this library;
 *   (2) IGXCreateFromTables                    takes the tables a set-up PetIGA `IGA` already
 *       holds (struct _p_IGA, include/petiga.h:327-391) -- what IGAComputeSystem inside
 *       libpetiga would bind.
 *
 * Error convention: every function returns an int error code, 0 = success, non-zero values are
 * PETSc's own PetscErrorCode numbers for the same condition (include/petscerror.h) so the
 * adapter can `CHKERRQ` them unchanged.  IGXGetLastError() returns the message SETERRQ would print.
 */
#ifndef PETIGA_AMD_H
#define PETIGA_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* PetscErrorCode values reused verbatim */
#define IGX_ERR_SUP             56   /* PETSC_ERR_SUP             */
#define IGX_ERR_ORDER           58   /* PETSC_ERR_ORDER           */
#define IGX_ERR_ARG_WRONG       62   /* PETSC_ERR_ARG_WRONG       */
#define IGX_ERR_ARG_OUTOFRANGE  63   /* PETSC_ERR_ARG_OUTOFRANGE  */
#define IGX_ERR_ARG_WRONGSTATE  73   /* PETSC_ERR_ARG_WRONGSTATE  */
#define IGX_ERR_LIB             76   /* PETSC_ERR_LIB (HIP runtime failure) */
#define IGX_ERR_PLIB            77   /* PETSC_ERR_PLIB            */
#define IGX_ERR_USER            83   /* PETSC_ERR_USER (e.g. non-positive Jacobian, src/petigaelem.c:989-993) */
#define IGX_ERR_MEM             55   /* PETSC_ERR_MEM             */

#define IGX_DECIDE (-1)              /* PETSC_DECIDE */

typedef struct _p_IGX    *IGX;       /* replaces `IGA`  (include/petiga.h:327)                  */
typedef struct _p_IGXMat *IGXMat;    /* replaces the `Mat` of IGACreateMat (src/petigamat.c:345) */
typedef struct _p_IGXVec *IGXVec;    /* replaces the `Vec` of IGACreateVec (src/petigavec.c:78)  */

const char *IGXGetLastError(void);

/* ------------------------------------------------------------------------------------------
 * Device point forms.  PetIGA's point callbacks are host function pointers
 * (IGAFormSystem etc., include/petiga.h:153-197) and cannot run on the GPU; the engine keeps
 * their contract (point-wise, un-weighted integrand, K[a][i][b][j] / F[a][i] row-major,
 * src/petigapoint.c:451-462) and binds the function at compile time.  `params` replaces `ctx`.
 * ------------------------------------------------------------------------------------------ */
typedef enum {
  IGX_FORM_NONE        = 0,
  IGX_FORM_POISSON     = 1, /* System: demo/Poisson{1,2,3}D.c:3-23        K=grad.grad, F=N*1          params: none */
  IGX_FORM_MASS        = 2, /* System: test/IGACreate.c:45-63             K=Na*Nb (per field), F=N    params: none */
  IGX_FORM_L2PROJ_X2   = 3, /* System: test/IGAFixTable.c:25-43           K=Na*Nb, F=N*sum x^2        params: none */
  IGX_FORM_POISSON_F   = 4, /* System: test/IGAFixTable.c:45-64           K=grad.grad, F=N*(-2 dim)   params: none */
  IGX_FORM_ERRNORM     = 5, /* System: test/IGAErrNorm.c:54-75 (dof=4)    L2 projection of 1,Sx,Sx2,Px              */
  IGX_FORM_ELASTICITY  = 6, /* System: demo/Elasticity3D.c:13-46 (dof=3)  params: {lambda, mu}                      */
  IGX_FORM_CAHNHILLIARD= 7, /* IFunction/IJacobian: demo/CahnHilliard3D.c:55-179 (and the 2-D demo)
                               params: {theta, alpha, cbar, L0, lambda, tau}; L0<=0 selects the 2-D demo's 3*alpha scaling */
  IGX_FORM_NSVMS       = 8, /* IFunction/IJacobian: demo/NavierStokesVMS.c:78-244 (dof=4)
                               params: {nu, fx, fy, fz, dt}  (dt: the reference reads TSGetTimeStep at :85,:173) */
  IGX_FORM_BOUNDARYINTEGRAL = 9, /* System: demo/BoundaryIntegral.c:26-56  interior Laplace, F=N*1 on visited faces (Neumann) */
  IGX_FORM_NITSCHE     = 10,/* System: demo/NitscheMethod.c:69-110  Poisson with Nitsche terms on visited faces (normals,
                               normal mesh size through IGAPointFormInvGradGeomMap); params: {max degree k} */
  IGX_FORM_BRATU       = 11,/* Function/Jacobian and IFunction/IJacobian: demo/Bratu.c, demo/BratuFJ.F90:23-176  params: {lambda} */
  IGX_FORM_ELASTICITY_F = 12,/* System: demo/Elasticity3D.c's K with a body force, F[a][i] = N_a f_i   params: {lambda, mu, fx, fy, fz} */
  IGX_FORM_DER3 = 13,        /* System / Function on third derivatives (p->shape[3], IGAPointFormDer3: test/IGAGeometryMap.c:179,221): K = N N + k3 d3N : d3N,
                              * F = N (1 + |x|^2) + f3 c : d3N + u3 N c : d3u; params {k3, f3, u3}; the general kernel (order-3 tabulation) */
  IGX_FORM_SURFACE = 15,     /* System on a curve / surface in space (IGXSetGeometry with nsd > dim; demo/ClassicalShell.c:57-80 reads p->mapX the same way):
                              * Laplace-Beltrami + mass, load |H| (mean-curvature vector from p->mapX[1], p->mapX[2]); dim 1, 2; the general kernel */
  IGX_FORM_PROPERTY = 14,    /* System: Poisson with conductivity A[..][0] and source A[..][npd-1] of the property array (IGXSetProperty), interpolated
                              * at the point from p->property; the general kernel */
  IGX_FORM_SOURCE      = 100 /* a user form compiled at run time: IGXSetFormSource */
} IGXFormKind;

/* ------------------------------------------------------------------------------------------
 * (1) Mirror of the PetIGA set-up API
 * ------------------------------------------------------------------------------------------ */
int IGXCreate(IGX *iga);                                   /* IGACreate          src/petiga.c:32   */
int IGXDestroy(IGX *iga);                                  /* IGADestroy         src/petiga.c:79   */
int IGXSetDim(IGX iga,int dim);                            /* IGASetDim          src/petiga.c:281  */
int IGXSetDof(IGX iga,int dof);                            /* IGASetDof          src/petiga.c:334  */
int IGXSetOrder(IGX iga,int order);                        /* IGASetOrder        src/petiga.c:463  (clipped to [1,4]) */
int IGXSetQuadrature(IGX iga,int i,int q);                 /* IGASetQuadrature   src/petiga.c:530  */
/* Quadrature rule of an axis (IGARuleType, include/petiga.h:82-87).  LEGENDRE (q = 1..10) and LOBATTO (q = 2..10) carry the
 * doubles of the reference's tables (src/petigarule.c:182-319, :321-459); USER takes any rule on [-1,1]; REDUCED is Gauss-Legendre
 * with q points on the first and the last element of the axis and q - 1 on the others (src/petigabasis.c:144-171): the engine keeps
 * q slots per element and gives the last one of an interior element weight 0 (the reference trims it, src/petigaelem.c:764-776),
 * so the results are the reference's and the cost is that of the full rule. */
typedef enum { IGX_RULE_LEGENDRE = 0, IGX_RULE_LOBATTO = 1, IGX_RULE_REDUCED = 2, IGX_RULE_USER = 3 } IGXRuleType;
int IGXSetRuleType(IGX iga,int i,IGXRuleType type);        /* IGASetRuleType     src/petiga.c:500  */
int IGXSetRuleSize(IGX iga,int i,int nqp);                 /* IGASetRuleSize     src/petiga.c:515  */
int IGXSetRule(IGX iga,int i,int q,const double x[],const double w[]); /* IGAGetRule + IGARuleSetRule  src/petigarule.c:145 */
int IGXGetRule(IGX iga,int i,int *q,double x[],double w[]); /* IGAGetRule + IGARuleGetRule  src/petigarule.c:160: the rule IGXSetUp
                                                              will use (x, w may be NULL; room for *q entries: ask for q first) */
int IGXSetProcessors(IGX iga,int i,int processors);        /* IGASetProcessors   src/petiga.c:547  */
int IGXSetComm(IGX iga,int size,int rank);                 /* the (size,rank) of the MPI_Comm given to IGACreate */
int IGXAxisSetDegree(IGX iga,int i,int p);                 /* IGAAxisSetDegree   src/petigaaxis.c:168 */
int IGXAxisSetPeriodic(IGX iga,int i,int flag);            /* IGAAxisSetPeriodic src/petigaaxis.c:150 */
int IGXAxisInitUniform(IGX iga,int i,int N,double Ui,double Uf,int C); /* IGAAxisInitUniform src/petigaaxis.c:401 */
int IGXAxisSetKnots(IGX iga,int i,int m,const double U[]); /* IGAAxisSetKnots    src/petigaaxis.c:202 */
int IGXSetUp(IGX iga);                                     /* IGASetUp           src/petiga.c:1450 */

/* Geometry: control net on the geometry grid (n+1 points per axis, i0 fastest), Cartesian X[...][nsd]
 * and optional NURBS weights W (NULL = polynomial).  Stands for IGASetGeometryDim + the arrays
 * iga->geometryX / iga->rationalW that IGALoadGeometry fills (src/petigaio.c:201-356).
 * dim <= nsd <= 3.  nsd > dim (a curve or a surface in space, demo/ClassicalShell.c:154): the geometry map is tabulated, the inverse
 * map is not -- shape functions and measure stay parametric, a face's normal is its axis (src/petigaelem.c:966-1029) -- and the form
 * reads p->mapX[1], p->mapX[2] (NEED_MAPX: p.X1 [nsd][dim], p.X2 [nsd][dim][dim]); such assemblies run on the general kernel. */
int IGXSetGeometry(IGX iga,int nsd,const double X[],const double W[]);
/* Derivative order (IGASetOrder, src/petiga.c:463): forms that read third derivatives -- p->shape[3] ([nen][dim][dim][dim], include/petiga.h:657)
 * as the dim^3 numbers behind the Hessian in Na / Nb, IGAPointFormDer3 (include/petiga.h:731) as p.d3u with NEED_D3U -- declare ORDER = 3
 * and need IGXSetOrder(iga,3) (the default order is the largest degree, src/petiga.c:1472-1475); they run on the point-form kernel, which
 * then tabulates K2, Rationalize, GeometryMap, InverseMap and ShapeFunctions at order 3 (src/petigamapinv.f90.in:49-60,
 * src/petigamapshf.f90.in:60-72).  Every other form is tabulated as far as it reads, whatever the order set here. */
/* Property array: npd numbers per node of the geometry grid (natural order, [node][npd]); npd = 0 drops it.  Stands for
 * IGASetPropertyDim + iga->propertyA as IGALoadProperty fills it (src/petigaio.c:359-458).  A point's form sees the values of its
 * element's nodes as p->property [nen][npd] (IGAElementBuildClosure, src/petigaelem.c:745-752; include/petiga.h:662): forms that
 * read them (NEED_PROP) run on the general kernel.  IGXRead / IGXWrite carry the array (info bit 1 of the file, src/petigaio.c:38,105). */
int IGXSetProperty(IGX iga,int npd,const double A[]);
int IGXGetPropertyDim(IGX iga,int *npd);   /* IGAGetPropertyDim src/petigaio.c:384 */

int IGXSetBoundaryValue(IGX iga,int axis,int side,int field,double value); /* IGASetBoundaryValue src/petigaform.c:324 */
int IGXSetBoundaryLoad (IGX iga,int axis,int side,int field,double value); /* IGASetBoundaryLoad  src/petigaform.c:340 */
int IGXClearBoundary(IGX iga);                                             /* IGAFormClearBoundary src/petigaform.c:143 (all faces) */
/* boundary-form pass of face (axis,side): the form is also integrated over that face, one point layer at the face, with
 * p->atboundary / p->normal semantics (IGAElementNextForm src/petigaelem.c:427; normals src/petigaval.F90:45-99) */
int IGXSetBoundaryForm(IGX iga,int axis,int side,int flag);                /* IGASetBoundaryForm  src/petigaform.c:356 */
int IGXSetFixTable(IGX iga,IGXVec U);                                      /* IGASetFixTable      src/petigaform.c:273 (NULL clears) */

/* IGASetFormSystem / IGASetFormMatrix / ... (src/petigaform.c:388-833): kind + params replace (fn,ctx).
 * The one kind serves System/Matrix/Vector or Function/Jacobian/IFunction/IJacobian as the reference demo does. */
int IGXSetForm(IGX iga,IGXFormKind kind,const double params[],int nparams);

/* The open end of the plugin API (IGASetFormSystem(iga,fn,ctx) with an arbitrary user fn, src/petigaform.c:388-833): a point
 * form given as HIP source and compiled at run time (hiprtc) into the general element kernel.  `source` defines a struct
 * `struct_name` with the contract of the built-in forms (petiga_amd/csrc/forms.hpp, the device restatement of
 * include/petiga.h:153-197 + src/petigapoint.c:427-462):
 *     struct MyForm {
 *       static constexpr int DOF = 1, ORDER = 1;          // fields per node; 2 if second derivatives of N or of U are read, 3 for third
 *                                                         // ones (p->shape[3]: dim^3 numbers behind the Hessian in Na / Nb; IGXSetOrder(3))
 *       static constexpr unsigned NEED = NEED_X | NEED_U; // point data read: NEED_X x, NEED_U u, NEED_UT du/dt, NEED_GU grad u,
 *                                                         // NEED_HU hess u, NEED_G IGAPointFormInvGradGeomMap; on the point-form kernel
 *                                                         // only: NEED_D3U p.d3u (IGAPointFormDer3), NEED_PROP p.property [nen][npd] with
 *                                                         // the point's shape table p.shape [nen][nf], NEED_MAPX p.X1 / p.X2 (p->mapX[1], [2])
 *       static __device__ void mat(const PtView &p,const double *Na,const double *Nb,double *T); // T[i*DOF+j]: K block of (a,b)
 *       static __device__ void vec(const PtView &p,const double *Na,double *R);                  // R[i]: F entries of a
 *     };
 * Na / Nb: [0] N, [1+i] dN/dx_i, [1+dim+i*dim+j] d2N/dx_i dx_j; p.prm[] = params (the callback's ctx); the integrand is
 * un-weighted and mat() must be linear in Na and in Nb, as every IGAFormSystem/Jacobian is.  Compile errors come back as
 * PETSC_ERR_USER with the compiler log in IGXGetLastError().  The seven drivers then work as with a built-in form: on the
 * matrix cores (feature_assemble<MyForm,...>, compiled on first use for the wave layout of the degree, about half a second)
 * for dim >= 2 and (p+1)^dim <= 64 (in 3-D also p = 4, 5 for forms with at most two / one accumulator set), on the point-form
 * kernel otherwise or with IGXSetKernel(1).  IGX_RTC_CACHE_DIR in the
 * environment keeps the compiled code objects on disk for later processes.  Optional declarations that
 * speed the matrix-core kernel up, all bit masks over the feature index of Na / Nb:
 *       static constexpr unsigned MAT_TEST_MASK = ...;         // features of Na that mat() reads (others never enter the GEMM)
 *       static constexpr unsigned PHI_MASK = ...;              // features anything reads (others are not tabulated at all)
 *       static constexpr unsigned long long MAT_PAIR_MASK = ...; // bit 8f+g: mat() has a point-INDEPENDENT coefficient on
 *                                                              // Na[f]*Nb[g] and nothing else: Gram matrices on the matrix cores
 *       static constexpr unsigned pair_block_mask(int f,int g); // with MAT_PAIR_MASK: bit i*DOF+j set when the pair (f,g) reaches
 *                                                              // block entry (i,j) (x*0 does not fold under IEEE rules)
 *       static constexpr unsigned MAT_NEED = ...;              // subset of NEED that mat() reads (matrix-only drivers skip the rest)
 *       static constexpr bool MAT_SYMMETRIC = true;            // mat(p,Na,Nb) == mat(p,Nb,Na) at every point (the form's promise)
 *       static constexpr unsigned VEC_TEST_MASK = ...;         // features of Na that vec() reads
 * A scalar (DOF 1) first-order form with MAT_TEST_MASK = the gradients, MAT_SYMMETRIC and VEC_TEST_MASK = 1 (N) whose NEED is at
 * most NEED_X -- demo/Poisson3D.c's System, with any x-dependent or anisotropic diffusion tensor and load -- takes the pencil
 * walk of the headline kernel in 3-D at p = 2, 3 (form_pencil<MyForm>: combined band rows, first-touch stores, the Dirichlet
 * fix-up inside the walk, identity or mapped / NURBS geometry) instead of the element mode: IGXSetKernel(2) insists on it.
 * A struct with MAT_PAIR_MASK (and pair_block_mask), 2 or 3 fields and VEC_ZERO -- demo/Elasticity3D.c's System -- takes the band-row
 * kernel in 3-D at p = 3 on the identity geometry (block_pencil<MyForm>: the element loop turned inside out, each band row written
 * once per pencil), like the built-in form; IGXSetKernel(4) insists on it.
 * The vector-only drivers (Vector / Function / IFunction) of ANY struct without an atboundary branch run on the sum-factorised
 * kernel in 3-D at p <= 3 (vec_sumfact<MyForm>: nqp evaluations of vec() on unit test features instead of nen x nqp).
 * The Tangent (Jacobian / IJacobian) of a nonlinear scalar struct takes the same pencil walk when the struct splits it as
 * sum_f A_f(a) B_f(b) with the test side A = (N, dN/dx_0, dN/dx_1, dN/dx_2[, laplacian N]) and the trial side B carrying the point:
 *       static constexpr int PENCIL_NFEAT = 4 or 5, PENCIL_NC = <numbers per Gauss point, at most 9>;
 *       static __device__ void pencil_coef(const PtView &p,double JW,double *c);        // from u, grad u, diag hess u, shift, prm
 *       static __device__ void pencil_trial(const double *c,double N,const double *g,double lap,double *B);   // B[0..PENCIL_NFEAT)
 * (FormCahnHilliard / FormBratu in petiga_amd/csrc/forms.hpp are written this way; 3-D, p = 2 or 3, no geometry, dof 1.)
 * A four-field first-order struct that separates its point coefficients from the basis functions -- demo/NavierStokesVMS.c's
 * Tangent -- takes the band-row kernel with point records (band_points + band_pt<MyForm>, 3-D, p = 3; p = 2 with IGXSetKernel(4)):
 *       static constexpr int NCOEF = <coefficients per Gauss point>;
 *       static __device__ void point_coef(const PtView &p,double *c);                   // from the state and prm at the point
 *       static __device__ void mat_c(const double *c,const double *Na,const double *Nb,double *T);
 * and optionally mat_unit / BAND_NFEAT / BAND_NACC with band_coef / band_finish (FormNSVMS in petiga_amd/csrc/forms.hpp is the
 * model).  A struct with the BAND_NACC hooks also declares a guard on its parameters, evaluated on the DEVICE by a one-lane
 * kernel whenever the parameters change, so it must be callable there:
 *       __host__ __device__ static bool band_params_ok(const double *prm);      // false: stay on the feature kernel (e.g. nu = 0)
 * A guard that does not compile for the device (a plain host function) does not fail the assembly: the automatic choice falls
 * through to the feature kernel and IGXGetKernelName says why; IGXSetKernel(4) and IGXCheckFormSource(…, 6) report the compiler's log.
 *       static constexpr int SHAPE_ORDER = 1;                  // ORDER = 2 only for hess u: second derivatives of N are not kept
 *       static constexpr bool VEC_ZERO = true;                 // vec() returns zeros: the vector phase runs for the Dirichlet lifting only
 * Boundary-form passes (IGXSetBoundaryForm; `if (p->atboundary)` in the reference's callback, e.g. demo/NitscheMethod.c:69-110): a
 * struct that declares
 *       static constexpr bool HAS_BOUNDARY = true;
 *       static __device__ void bmat(const PtView &p,const double *Na,const double *Nb,double *T);   // the integrands at a point of a
 *       static __device__ void bvec(const PtView &p,const double *Na,double *R);                    // visited face (p.normal, p.boundary_id)
 * is integrated with bmat / bvec over the visited faces; a struct without them with its ordinary mat / vec, as the reference would. */
int IGXSetFormSource(IGX iga,const char *source,const char *struct_name,const double params[],int nparams);

/* On-disk formats (PETSc binary, big-endian): the discretisation + NURBS control net written by IGAWrite / igakit,
 * and a Vec in natural order.  IGXRead replaces dim, axes and geometry of `iga` (dof is kept); call IGXSetUp next. */
int IGXRead (IGX iga,const char filename[]);              /* IGARead     src/petigaio.c:141 -> IGALoad :11  */
int IGXWrite(IGX iga,const char filename[]);              /* IGAWrite    src/petigaio.c:171 -> IGASave :75  */
int IGXWriteVec(IGX iga,IGXVec vec,const char filename[]); /* IGAWriteVec src/petigaio.c:640 */
int IGXReadVec (IGX iga,IGXVec vec,const char filename[]); /* IGAReadVec  src/petigaio.c:690 */

/* sizes after IGXSetUp */
int IGXGetSizes(IGX iga,int elem_sizes[3],int elem_start[3],int elem_width[3],
                int node_sizes[3],int node_lstart[3],int node_lwidth[3],int node_gstart[3],int node_gwidth[3]);
int IGXGetProcessors(IGX iga,int proc_sizes[3],int proc_ranks[3]);
/* IGAGetBasis + struct _n_IGABasis (include/petiga.h:122-141): the 1-D tables IGXSetUp built for axis i, as the element loop
 * reads them (src/petigabasis.c:83-219): offset[nel], detJac[nel], weight[nel][nqp], point[nel][nqp], value[nel][nqp][nen][5].
 * Any array may be NULL; ask for the three sizes first. */
int IGXGetBasis(IGX iga,int i,int *nel,int *nqp,int *nen,int offset[],double detJac[],double weight[],double point[],double value[]);
int64_t IGXGetElementCount(IGX iga);   /* local elements */

/* ------------------------------------------------------------------------------------------
 * (2) Building the same object from a set-up PetIGA `IGA` (what libpetiga would pass)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  /* per axis: struct _n_IGAAxis (include/petiga.h:80-96) */
  int           p, m, periodic, nel, nnp;
  const double *U;        /* [m+1] */
  const int    *span;     /* [nel] */
  /* per axis: struct _n_IGABasis (include/petiga.h:122-141) */
  int           nqp, nen;
  const int    *offset;   /* [nel]            */
  const double *detJac;   /* [nel]            */
  const double *weight;   /* [nel][nqp]       */
  const double *point;    /* [nel][nqp]       */
  const double *value;    /* [nel][nqp][nen][5] */
} IGXAxisTables;

typedef struct {
  int dim, dof, order;                       /* iga->dim, dof, order                              */
  IGXAxisTables axis[3];
  int proc_sizes[3], proc_ranks[3];          /* iga->proc_sizes/ranks                             */
  int elem_sizes[3], elem_start[3], elem_width[3];
  int node_sizes[3], node_lstart[3], node_lwidth[3], node_gstart[3], node_gwidth[3];
  int nsd;                                   /* iga->geometry (0 = none)                          */
  int rational;                              /* iga->rational                                     */
  const double *geometryX;                   /* ghosted local [gw2][gw1][gw0][nsd]  (iga->geometryX) */
  const double *rationalW;                   /* ghosted local [gw2][gw1][gw0]       (iga->rationalW) */
  /* appended in round 6 -- zero the whole struct before filling it (the adapter does: PetscMemzero), so that a field this header
   * grows later reads as "absent" */
  int property;                              /* iga->property: numbers per node (0 = none), include/petiga.h:350 */
  const double *propertyA;                   /* ghosted local [gw2][gw1][gw0][npd]  (iga->propertyA, include/petiga.h:353) */
} IGXTables;

int IGXCreateFromTables(const IGXTables *t,IGX *iga);

/* ------------------------------------------------------------------------------------------
 * Matrices and vectors (device resident)
 * ------------------------------------------------------------------------------------------ */
/* IGACreateMat (src/petigamat.c:345-549): exact pattern of the reference, block CSR with
 * dof x dof row-major blocks (AIJ for dof=1, BAIJ for dof>1, src/petiga.c:1328).  Rows are the
 * nodes of the local row box, columns are local column indices; see IGXMatGetLayout. */
int IGXCreateMat(IGX iga,IGXMat *mat);
int IGXMatDestroy(IGXMat *mat);
int IGXMatGetInfo(IGXMat mat,int64_t *nbrows,int64_t *nblocks,int *bs);
/* device pointers: browptr int64[nbrows+1], bcolidx int32[nblocks], val double[nblocks*bs*bs] */
int IGXMatGetDeviceArrays(IGXMat mat,const int64_t **browptr,const int32_t **bcolidx,double **val);
int IGXMatCopyToHost(IGXMat mat,int64_t *browptr,int32_t *bcolidx,double *val); /* any may be NULL */
/* per axis: number of row / column indices and the global node index each maps to
 * (rows: node index after periodic wrap; cols likewise) -- the LGMap of src/petigagrid.c:213 */
int IGXMatGetLayout(IGXMat mat,int nrow[3],int ncol[3]);
int IGXMatGetAxisMaps(IGXMat mat,int axis,int *rownode /*[nrow]*/,int *colnode /*[ncol]*/);

/* IGACreateVec (src/petigavec.c:78): one value per (row node, field), same row numbering as the matrix */
int IGXCreateVec(IGX iga,IGXVec *vec);
int IGXVecDestroy(IGXVec *vec);
int IGXVecGetSize(IGXVec vec,int64_t *n);
int IGXVecGetDeviceArray(IGXVec vec,double **array);
int IGXVecCopyToHost(IGXVec vec,double *host);
int IGXVecCopyFromHost(IGXVec vec,const double *host);

/* ------------------------------------------------------------------------------------------
 * The hot path.  Same meaning and argument order as the reference drivers; outputs are zeroed
 * first, then every local element's K_e/F_e is formed, BC-fixed and added (colour-ordered,
 * conflict-free, deterministic).  All work is enqueued on the engine's stream
 * (IGXSetStream); call IGXSynchronize before reading results through raw device pointers.
 * ------------------------------------------------------------------------------------------ */
int IGXComputeSystem   (IGX iga,IGXMat A,IGXVec b);                               /* src/petigaksp.c:149 */
int IGXComputeMatrix   (IGX iga,IGXMat A);                                        /* src/petigaksp.c:79  */
int IGXComputeVector   (IGX iga,IGXVec b);                                        /* src/petigaksp.c:33  */
int IGXComputeFunction (IGX iga,IGXVec U,IGXVec F);                               /* src/petigasnes.c:23 */
int IGXComputeJacobian (IGX iga,IGXVec U,IGXMat J);                               /* src/petigasnes.c:82 */
int IGXComputeIFunction(IGX iga,double a,IGXVec V,double t,IGXVec U,IGXVec F);    /* src/petigats.c:23   */
int IGXComputeIJacobian(IGX iga,double a,IGXVec V,double t,IGXVec U,IGXMat J);    /* src/petigats.c:92   */
/* The pair a Newton step asks for at one state -- SNESComputeFunction then SNESComputeJacobian on the same U (TS: the same a, V, t)
 * -- in ONE pass of the element loop (R_e and K_e share the tabulation and the state's point values).  F and J are
 * those of IGXComputeIFunction + IGXComputeIJacobian (IGXComputeFunction + IGXComputeJacobian).  With IGX_FUSE_RESID=1 in the
 * environment and where a fused kernel exists (state_pencil_kr: Cahn-Hilliard / Bratu, 3-D, p = 2, no geometry, no boundary
 * loads: config 4) the Residual rides on the Tangent's MFMAs as one more operand column; otherwise -- the default: the fused walk
 * measured 3 % slower than the two passes, DESIGN.md 3.1 -- the two drivers run one after the other.  IGXGetKernelName tells. */
int IGXComputeIFunctionIJacobian(IGX iga,double a,IGXVec V,double t,IGXVec U,IGXVec F,IGXMat J);
int IGXComputeFunctionJacobian(IGX iga,IGXVec U,IGXVec F,IGXMat J);

/* Functionals of a discrete field: S[k] = sum over this rank's elements and points of JW * scalar_k(point)
 * (IGAComputeScalar, src/petigacomp.c:35-98, before its MPI_Allreduce: with several ranks the caller sums S over the
 * ranks, and U must hold the ghost rows' values).  The point callbacks are the ones the reference's tests use:
 *   IGX_SCALAR_VOLUME  n=2  test/IGAGeometryMap.c:383: S[0] volume of the mapped domain, S[1] area of the faces marked with
 *                           IGXSetBoundaryForm (boundary passes: normals, detS); U may be NULL
 *   IGX_SCALAR_X2ERR   n=1  src/petigacomp.c:102 (ErrorSqr) with Exact = sum x_i^2 (test/IGAFixTable.c:66), dof 1
 *   IGX_SCALAR_ERRNORM n=4  ErrorSqr with test/IGAErrNorm.c:26 as Exact, dof 4; params {k}: k=0 values, 1 gradients,
 *                           2 Hessians; U NULL gives the norms of the exact fields (squared, like IGAComputeErrorNorm
 *                           before its sqrt, src/petigacomp.c:160).
 * Sums are formed in a fixed order: bitwise repeatable. */
typedef enum { IGX_SCALAR_VOLUME = 1, IGX_SCALAR_X2ERR = 2, IGX_SCALAR_ERRNORM = 3 } IGXScalarKind;
int IGXComputeScalar(IGX iga,IGXVec U,int kind,const double params[],int nparams,int n,double S[]);
/* ... with the user's own point functional (IGAFormScalar, include/petiga.h:188-191; IGAComputeScalar's `Scalar` argument,
 * src/petigacomp.c:35) given as HIP source and compiled at run time like a form of IGXSetFormSource: a struct with
 *   static constexpr int DOF, ORDER, NSCALAR (= n); static constexpr unsigned NEED;
 *   static __device__ void scalar(const igx::PtView &p, double *S);     // un-weighted integrand values at the point
 * p carries x, u, grad u, hess u as NEED asks, prm = params, and atboundary / normal / boundary_id on the passes over faces marked
 * with IGXSetBoundaryForm (an error norm is this with |Exact(x) - u|^2: IGAComputeErrorNorm, src/petigacomp.c:122-186). */
int IGXComputeScalarSource(IGX iga,IGXVec U,const char *source,const char *struct_name,const double params[],int nparams,int n,double S[]);

/* engine controls */
int IGXSetStream(IGX iga,void *hipStream);      /* hipStream_t; NULL = default stream            */
int IGXSynchronize(IGX iga);
/* 0 = automatic; 1 = generic point-form kernel (no MFMA); 2 = MFMA gradient-Gram pencil kernel (Poisson, uniform
 * degree 2/3, no geometry); 3 = feature-GEMM element kernel (any form/geometry, dim >= 2, nen <= 64;
 * K_e on MFMA, vector-only operations share its tabulation); 4 = block pencil kernel (band rows by node layer:
 * constant-coefficient forms with 2 or 3 fields and F = 0, dim 3, p = 3, identity geometry, System / Matrix drivers;
 * the automatic choice takes it where it applies).  A forced kernel that does not cover the case answers PETSC_ERR_SUP. */
int IGXSetKernel(IGX iga,int which);
int IGXGetKernelName(IGX iga,char *buf,int len);
/* name of the kernel the last IGXCompute* used   */
/* time of the dominant kernel of the last IGXCompute* call, measured with HIP events on the
 * engine's stream (ms); IGXSetTiming(1) enables the events */
int IGXSetTiming(IGX iga,int flag);
int IGXGetLastTiming(IGX iga,double *total_ms,double *kernel_ms,int *launches);
/* the dominant kernel of the last IGXCompute* call: its name, the time of its launches (HIP events on the
 * engine's stream, ms), how many launches, how many elements they processed and the MFMA flops it
 * executes per element (for roofline accounting) */
int IGXGetDominantKernelTiming(IGX iga,char *name,int len,double *ms,int *launches,int64_t *elements,double *flop_per_element);

/* element colouring used by the scatter: colour of local element (i,j,k) and the number of
 * colours (bit-exact contract, tests/test_coloring.py) */
int IGXGetColoring(IGX iga,int ncolors[3]);
int IGXGetElementColor(IGX iga,int axis,int e);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (one process per GPU): ghost-row exchange buffers.  Rank r's ghost rows (nodes it
 * holds but does not own) are packed per upper neighbour (send list, k < nsend); the owner adds
 * them (receive list, k < nrecv).  A message is mat_doubles values (if A given) followed by
 * vec_doubles values (if b given).  Replaces the stash traffic of MatAssemblyBegin/End and
 * VecAssemblyBegin/End (src/petigaksp.c:197-200).  These are the building blocks; IGXReduceGhostRows / IGXRefreshGhosts
 * below run the whole exchange inside the library (RCCL or a caller-provided transport).
 * ------------------------------------------------------------------------------------------ */
int IGXGetNeighborCount(IGX iga,int *nsend,int *nrecv);
int IGXGetNeighborInfo(IGX iga,int send /*1=send list,0=recv list*/,int k,int *rank,int64_t *mat_doubles,int64_t *vec_doubles);
int IGXPackGhostRows  (IGX iga,IGXMat A,IGXVec b,int k,double *devbuf);   /* A or b may be NULL */
int IGXUnpackGhostRows(IGX iga,IGXMat A,IGXVec b,int k,const double *devbuf);
/* Reverse direction, before a nonlinear assembly (IGAGetLocalVecArray / DMGlobalToLocal, src/petigavec.c:256-269): the
 * owner's values of a state vector travel to the ranks holding the node as a ghost.  Pack entry k of the RECEIVE list
 * (vec_doubles of that entry), send it to that rank, which unpacks it as entry k' of its SEND list (assignment). */
int IGXPackOwnerValues  (IGX iga,IGXVec v,int k,double *devbuf);
int IGXUnpackGhostValues(IGX iga,IGXVec v,int k,const double *devbuf);
/* 1 if this rank owns the node of local row (r0,r1,r2): after the exchange only owned rows are final */
int IGXRowOwned(IGX iga,int r0,int r1,int r2);

/* The transport, inside the library.  IGXReduceGhostRows = pack every send-list message, move them, add the received ones
 * (the whole of MatAssemblyBegin/End + VecAssemblyBegin/End, src/petigaksp.c:197-200); IGXRefreshGhosts = the owner's values
 * of a state vector to the ranks that hold the node as a ghost (DMGlobalToLocal in IGAGetLocalVecArray,
 * src/petigavec.c:256-269).  Both are enqueued: packs, transfers and unpacks run on the library's exchange stream, which
 * waits for the engine stream's last launch and is waited for by its next one (events); the host never blocks.
 *   IGXCommInitRCCL: grouped ncclSend / ncclRecv (RCCL over xGMI).  librccl.so is bound with dlopen (an already loaded copy
 *     is reused; `librccl_path` or $IGX_RCCL_LIB name another).  Rank 0 calls IGXCommGetUniqueId and the caller broadcasts
 *     the 128 bytes (MPI_Bcast in PetIGA, torch.distributed in bench.py); size and rank are those of IGXSetComm.
 *   IGXCommInitTransport: a host callback that moves the packed DEVICE buffers (test transport: gloo; an MPI caller).
 *     It is called after the exchange stream has been synchronised and must return with the receive buffers filled. */
typedef struct { char internal[128]; } IGXUniqueId;       /* = ncclUniqueId */
typedef int (*IGXTransportFn)(void *ctx,int nsend,const int send_peer[],double *const send_buf[],const int64_t send_count[],
                              int nrecv,const int recv_peer[],double *const recv_buf[],const int64_t recv_count[]);
int IGXCommGetUniqueId(IGXUniqueId *id,const char *librccl_path /* may be NULL */);
int IGXCommInitRCCL(IGX iga,const IGXUniqueId *id,const char *librccl_path /* may be NULL */);
int IGXCommInitTransport(IGX iga,IGXTransportFn fn,void *ctx);
int IGXCommDestroy(IGX iga);
int IGXReduceGhostRows(IGX iga,IGXMat A,IGXVec b);        /* A or b may be NULL */
int IGXRefreshGhosts(IGX iga,IGXVec v);
int IGXCommGetLastBytes(IGX iga,int64_t *bytes_sent);      /* of the last exchange */
/* The rate (GB/s per direction) a face message of the ghost-row reduction travels at, as the face-first decision of the pencil
 * walks uses it: IGXCommInitRCCL times one grouped ncclSend / ncclRecv of IGX_LINK_PROBE_MB (default 64) MB with
 * every face neighbour, all faces at once, after an untimed one that pays the connection set-up; $IGX_LINK_GBS overrides it.
 * source: 0 the constant 60 (no probe: one rank, no face neighbour, a host transport), 1 measured, 2 $IGX_LINK_GBS;
 * probe_ms / faces: the timed group and the face messages per direction it held.  Any pointer may be NULL. */
int IGXCommGetLinkRate(IGX iga,double *gbs,int *source,double *probe_ms,int *faces);
/* Passes of the last pencil-walk assembly over the rank's box: 1, or 1 + the upper faces (axes 2, 1, 0) that were assembled first
 * so that their messages travel under the rest (the decision: extra passes' cost against largest face / link rate). */
int IGXGetFacePasses(IGX iga,int *passes);
/* one rank sends n doubles to itself through the RCCL path (binding, communicator, streams and events on a single GPU) */
int IGXCommLoopbackTest(IGX iga,int64_t n,double *maxdiff);
/* What the bound transport itself reports: kind 1 = RCCL, 2 = host callback (0: none bound); ranks = ncclCommCount of the RCCL
 * communicator (the number of ranks RCCL actually connected; MPI_Comm_size of the reference's communicator, src/petiga.c:1111),
 * the size given to IGXSetComm for a host transport. */
int IGXCommGetRanks(IGX iga,int *kind,int *ranks);

/* ------------------------------------------------------------------------------------------
 * Hand-back to PETSc (used by adapter/petiga_amd_petsc.c).  PETSc assembles device matrices from coordinate
 * lists: MatSetPreallocationCOO(A,n,coo_i,coo_j) once, MatSetValuesCOO(A,v,ADD_VALUES) per assembly with v on the device.
 * The engine's value array (IGXMatGetDeviceArrays) is that v; IGXMatGetCOO gives the index pair of every stored scalar in
 * the same order (block by block, row-major inside a block).  numbering 0: natural, node*dof+field with axis 0 fastest
 * (IGA_Grid_LocalIndices, src/petigagrid.c); 1: PETSc's, i.e. PetIGA's AO (AOCreateMemoryScalable over the ranks' owned boxes,
 * src/petigagrid.c:185-199).  owned_only: entries of rows this rank does not own get -1, which MatSetPreallocationCOO ignores
 * (use after IGXReduceGhostRows; without it PETSc itself moves the not-owned rows, as MatAssemblyBegin/End always did).
 * Arrays have nblocks*bs*bs (matrix) / vector-size entries, on the device (on_device != 0) or on the host.
 * IGXVecCopyFromGhosted / ToGhosted: the rank's ghosted local array [gw2][gw1][gw0][dof] (IGAGetLocalVecArray,
 * src/petigavec.c:256-269) to / from an IGXVec: identical unless a periodic axis is wrapped inside the rank.
 * IGXMatGetCOODevice: the same lists kept on the device by the matrix (freed by IGXMatFreeCOO / IGXMatDestroy), in the caller's
 * index width (index_bytes 8, or 4 for a PETSc with 32-bit PetscInt; PETSC_ERR_ARG_OUTOFRANGE when an index does not fit):
 * what a device Mat type's MatSetPreallocationCOO takes without 2 x 8 bytes per stored scalar of host staging on this side.
 * on_device copies (IGXVecCopyFromGhosted / ToGhosted, IGXMatGetCOO, IGXVecGetIndices) are ordered on the IGX's own stream:
 * a caller on another stream calls IGXSynchronize first (or shares the stream, IGXSetStream). */
int IGXMatGetCOO(IGXMat A,int numbering,int owned_only,int64_t *coo_i,int64_t *coo_j,int on_device);
int IGXMatGetCOODevice(IGXMat A,int numbering,int owned_only,int index_bytes,void **coo_i,void **coo_j);
int IGXMatFreeCOO(IGXMat A);
int IGXDeviceToHost(void *host,const void *dev,size_t bytes);   /* hipMemcpy for a caller without the HIP headers (the adapter) */
int IGXVecGetIndices(IGXVec v,int numbering,int owned_only,int64_t *idx,int on_device);
int IGXVecGetGhostedSize(IGXVec v,int64_t *n);
int IGXVecCopyFromGhosted(IGXVec v,const double *array,int on_device);
int IGXVecCopyToGhosted(IGXVec v,double *array,int on_device);

/* Checksums of an assembled system over the rows this rank OWNS (after the ghost-row reduction only those are final):
 * S[0] = sum A_ij, S[1] = sum |A_ij|, S[2] = sum b_i, S[3] = sum b_i^2; A or b may be NULL.  Added over the ranks of any
 * partition they equal the single-rank values up to rounding: bench.py asserts exactly that on its N > 1 path, the PetIGA
 * adapter can use it as a MatNorm-free sanity check.  Fixed summation order (bitwise repeatable). */
int IGXChecksum(IGX iga,IGXMat A,IGXVec b,double S[4]);

/* Compile-only check of a run-time form (no GPU needed): IGXSetFormSource compiles the point-form kernel; this compiles the
 * matrix-core kernel the drivers would launch for the degrees set so far (dim >= 2, (p+1)^dim <= 64; 3-D: <= 256), for the matrix drivers
 * (with_matrix != 0) or the vector-only ones.  gram != 0 when the struct declares MAT_PAIR_MASK (it decides the wave layout at
 * dof = 4; on a GPU the flag is read from the compiled module).  gram == 2: the pencil walk's instantiations instead (form_pencil,
 * System and Matrix driver, for the current degree and geometry; dim 3, p = 2 or 3).  gram == 3: the sum-factorised vector kernel
 * (vec_sumfact<MyForm>: Vector / Function / IFunction in 3-D at p <= 3) for the current geometry kind.  gram == 4: state_pencil<p, MyForm>
 * (the Tangent of a scalar struct with the PENCIL_* hooks, below).  gram == 5: block_pencil<MyForm> (System and Matrix driver) of a
 * constant-coefficient multi-field struct.  gram == 6: band_points + band_pt<MyForm> (Matrix / Jacobian / IJacobian of a four-field
 * struct that separates its point coefficients: NCOEF, point_coef, mat_c), without a geometry and on a NURBS map.
 * Returns 0 or IGX_ERR_USER with the compiler's log. */
int IGXCheckFormSource(IGX iga,int with_matrix,int gram);

/* Evidence of the overlap of the ghost-row exchange with the assembly: after IGXReduceGhostRows of an assembly
 * that formed the upper face of axis 2 first, the time in ms by which that face's messages were packed BEFORE the assembly's
 * last launch finished (positive: they travelled under the remaining launches).  PETSC_ERR_ARG_WRONGSTATE when the last
 * reduction did not start early (one rank on axis 2, IGX_OVERLAP=0, another kernel path). */
int IGXCommGetOverlap(IGX iga,double *ms);
/* ... and how many of the three phases of that reduction (upper faces of axes 2, 1, 0) were packed behind a face mark of the
 * assembly instead of its last launch: the pencil walk of a rank with upper neighbours on three axes marks all three. */
int IGXCommGetEarlyPhases(IGX iga,int *n);

/* Shader clock under load.  With IGX_CLOCK_PROBE=1 in the environment at IGXCreate, the first and the last workgroup of every
 * pencil-kernel launch add their s_memtime ticks and the ticks of the constant 100 MHz s_memrealtime counter over their walk to
 * two sums; this returns the ratio (MHz) and the elements those wavefronts walked since the previous call, and clears the sums.  The MFMA roofline in bench.py is quoted against the
 * nominal 2400 MHz peak; this figure says what the chip actually sustained while the kernel ran. */
int IGXGetClockProbe(IGX iga,double *shader_mhz,int64_t *elements);

/* library / device info for logs */
int IGXGetDeviceInfo(char *buf,int len);

#ifdef __cplusplus
}
#endif
#endif /* PETIGA_AMD_H */
