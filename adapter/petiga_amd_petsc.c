/*
 * petiga_amd_petsc.c -- the PETSc hand-back adapter: PetIGA's seven assembly drivers
 *   IGAComputeVector / IGAComputeMatrix / IGAComputeSystem        (src/petigaksp.c:33,79,149)
 *   IGAComputeFunction / IGAComputeJacobian                       (src/petigasnes.c:23,82)
 *   IGAComputeIFunction / IGAComputeIJacobian                     (src/petigats.c:23,92)
 * with unchanged signatures (include/petiga.h:837-877), their bodies replaced by calls into libpetiga_amd.so, so that
 * IGAKSPFormOperators (src/petigaksp.c:219), IGASNESFormFunction/Jacobian (src/petigasnes.c:141,156) and
 * IGATSFormIFunction/IJacobian (src/petigats.c:479,506) -- i.e. KSP / SNES / TS -- drive the solve as before.
 *
 * Build: compiled INTO libpetiga in place of the bodies in src/petigaksp.c, src/petigasnes.c, src/petigats.c, only when
 * PETSC_DIR is set (it needs petsc.h and petiga.h; this repository's image has neither, so the file is not compiled here --
 * no stand-in headers).  Every IGX* call below is exercised, in this order, by tests/test_gpu_handback.py through ctypes.
 *     make -C $PETIGA_DIR CFLAGS+="-DPETIGA_HAVE_AMD -I<repo>/include" LDLIBS+="-L<repo>/petiga_amd -lpetiga_amd"
 * PETSc >= 3.17 (MatSetPreallocationCOO / VecSetPreallocationCOO), a device matrix type (-iga_mat_type aijhipsparse or
 * aijkokkos) and, above 2^31 stored scalars per rank, --with-64-bit-indices.
 *
 * Data flow per assembly (nothing is copied through the host when the Mat / Vec are device types: state arrays are taken with
 * VecGetArrayReadAndMemType and handed over as device pointers, the coordinate lists are built on the device by the library --
 * IGXMatGetCOODevice, in PetscInt's width -- and the values are the engine's own device array):
 *   U,V (global Vec) --IGAGetLocalVecArray--> ghosted local array --IGXVecCopyFromGhosted--> IGXVec        (state)
 *   IGXCompute*  ->  IGXMat values / IGXVec on the device
 *   MatSetValuesCOO(A, engine value array, INSERT_VALUES)  with the coordinate list set once (IGXMatGetCOO, PETSc numbering):
 *   PETSc adds the duplicate (i,j) of different ranks and routes rows of not-owned nodes to their owners -- the stash
 *   traffic of MatAssemblyBegin/End (src/petigaksp.c:197-198), now on the device.  (Alternative: IGXReduceGhostRows over RCCL
 *   first, coordinate list with owned_only = 1.)
 *
 * The point callback: host function pointers cannot run on the GPU.  The user program registers the device form next to
 * its host callback,  IGASetFormAMD(iga, IGX_FORM_ELASTICITY, (PetscReal[]){lambda,mu}, 2);  or, for a callback that is not one
 * of the built-in forms, its HIP source:  IGASetFormSourceAMD(iga, source, "MyForm", params, n)  (IGXSetFormSource: compiled at
 * run time against the library's own headers, include/petiga_amd.h).  When none is registered, or the engine answers
 * PETSC_ERR_SUP (IGX_ERR_SUP: a case no device kernel covers), the driver falls through to PetIGA's own CPU loop
 * (IGACompute*_CPU, the original bodies), so every program keeps working; any other engine error is raised.
 *
 * Memory of the coordinate lists: MatSetPreallocationCOO wants all n = nnz index pairs at once (2 x sizeof(PetscInt) x n: 93 GB
 * for the metric configuration's 5.84e9 non-zeros with 64-bit indices, which that configuration needs on one rank).  They live
 * on the device inside the IGXMat until the preallocation is done and are freed right after (IGXMatFreeCOO); PETSc's own
 * permutation arrays are PETSc's business.  A host Mat type gets host copies, in chunks of at most 2^28 entries.
 */
#if defined(PETIGA_HAVE_AMD)
#include <petiga.h>
#include <petiga_amd.h>

typedef struct {
  IGX       igx;
  IGXMat    A;        /* device matrix with PetIGA's pattern (IGACreateMat) */
  IGXVec    b,U,V;
  PetscBool coo_mat,coo_vec;   /* coordinate lists handed to the Mat / Vec */
  IGXFormKind kind; PetscReal params[8]; PetscInt nparams;
  char     *source,*struct_name;   /* run-time form (IGASetFormSourceAMD) or NULL */
  int      *span32[3],*offset32[3];  /* narrowed copies of ax->span / bd->offset when PetscInt is 64 bits wide */
} IGAAmdCtx;

static PetscErrorCode IGAAmdCtxDestroy(void *p)
{
  IGAAmdCtx *c = (IGAAmdCtx*)p;
  PetscFunctionBegin;
  if (c) {
    PetscInt i;
    IGXVecDestroy(&c->V); IGXVecDestroy(&c->U); IGXVecDestroy(&c->b); IGXMatDestroy(&c->A); IGXDestroy(&c->igx);
    for (i=0; i<3; i++) { PetscCall(PetscFree(c->span32[i])); PetscCall(PetscFree(c->offset32[i])); }
    PetscCall(PetscFree(c->source)); PetscCall(PetscFree(c->struct_name));
    PetscCall(PetscFree(c));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

#define IGXCHK(comm,call) do { int rc_ = (call); if (PetscUnlikely(rc_)) SETERRQ(comm,(PetscErrorCode)rc_,"%s",IGXGetLastError()); } while (0)
/* a driver's IGXCompute* call: PETSC_ERR_SUP from the engine means "no device kernel covers this case" -> PetIGA's own loop */
/* (the hand-over is not silent: -info prints which driver left the GPU and the engine's reason) */
#define IGXTRY(comm,call,fallback) do { int rc_ = (call); \
    if (rc_ == IGX_ERR_SUP) { PetscCall(PetscInfo(NULL,"petiga_amd: %s is not covered by a device kernel (%s); assembling with PetIGA's CPU loop\n",#call,IGXGetLastError())); PetscFunctionReturn(fallback); } \
    if (PetscUnlikely(rc_)) SETERRQ(comm,(PetscErrorCode)rc_,"%s",IGXGetLastError()); } while (0)

static PetscErrorCode IGAAmdCtxGet(IGA iga,IGAAmdCtx **out)
{
  IGAAmdCtx *c; PetscContainer box;
  PetscFunctionBegin;
  PetscCall(PetscObjectQuery((PetscObject)iga,"IGAAmdCtx",(PetscObject*)&box));
  if (!box) {
    PetscCall(PetscNew(&c));
    PetscCall(PetscContainerCreate(PetscObjectComm((PetscObject)iga),&box));
    PetscCall(PetscContainerSetPointer(box,c));
    PetscCall(PetscContainerSetUserDestroy(box,IGAAmdCtxDestroy));
    PetscCall(PetscObjectCompose((PetscObject)iga,"IGAAmdCtx",(PetscObject)box));
    PetscCall(PetscContainerDestroy(&box));
  } else PetscCall(PetscContainerGetPointer(box,(void**)&c));
  *out = c;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* user-facing: the HIP source of a device form (struct `struct_name` with the contract of include/petiga_amd.h, IGXSetFormSource)
   stands for the host callback of this IGA: the open end of the plugin API reaches PetIGA programs */
PetscErrorCode IGASetFormSourceAMD(IGA iga,const char source[],const char struct_name[],const PetscReal params[],PetscInt nparams)
{
  IGAAmdCtx *c; PetscInt i;
  PetscFunctionBegin;
  PetscValidHeaderSpecific(iga,IGA_CLASSID,1);
  PetscAssertPointer(source,2); PetscAssertPointer(struct_name,3);
  PetscCheck(nparams >= 0 && nparams <= 8,PetscObjectComm((PetscObject)iga),PETSC_ERR_ARG_OUTOFRANGE,"at most 8 form parameters");
  PetscCall(IGAAmdCtxGet(iga,&c));
  PetscCall(PetscFree(c->source)); PetscCall(PetscFree(c->struct_name));
  PetscCall(PetscStrallocpy(source,&c->source));
  PetscCall(PetscStrallocpy(struct_name,&c->struct_name));
  c->kind = IGX_FORM_SOURCE; c->nparams = nparams;
  for (i=0; i<nparams; i++) c->params[i] = params[i];
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* user-facing: which built-in device form stands for the host callback of this IGA */
PetscErrorCode IGASetFormAMD(IGA iga,IGXFormKind kind,const PetscReal params[],PetscInt nparams)
{
  IGAAmdCtx *c; PetscInt i;
  PetscFunctionBegin;
  PetscValidHeaderSpecific(iga,IGA_CLASSID,1);
  PetscCheck(nparams >= 0 && nparams <= 8,PetscObjectComm((PetscObject)iga),PETSC_ERR_ARG_OUTOFRANGE,"at most 8 form parameters");
  PetscCall(IGAAmdCtxGet(iga,&c));
  PetscCall(PetscFree(c->source)); PetscCall(PetscFree(c->struct_name));
  c->kind = kind; c->nparams = nparams;
  for (i=0; i<nparams; i++) c->params[i] = params[i];
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the engine object of an IGA, built once from the tables IGASetUp computed (struct _p_IGA, include/petiga.h:327-391) */
static PetscErrorCode IGAGetAmd(IGA iga,IGAAmdCtx **out)
{
  IGAAmdCtx *c = NULL; PetscContainer box; MPI_Comm comm = PetscObjectComm((PetscObject)iga);
  PetscFunctionBegin;
  *out = NULL;
  PetscCall(PetscObjectQuery((PetscObject)iga,"IGAAmdCtx",(PetscObject*)&box));
  if (!box) PetscFunctionReturn(PETSC_SUCCESS);            /* no device form registered: CPU path */
  PetscCall(PetscContainerGetPointer(box,(void**)&c));
  if (!c->igx) {
    IGXTables t; PetscInt i;
    PetscCall(PetscMemzero(&t,sizeof(t)));
    t.dim = (int)iga->dim; t.dof = (int)iga->dof; t.order = (int)iga->order;
    for (i=0; i<iga->dim; i++) {
      IGAAxis ax = iga->axis[i]; IGABasis bd = iga->basis[i];
      t.axis[i].p = (int)ax->p; t.axis[i].m = (int)ax->m; t.axis[i].periodic = (int)ax->periodic;
      t.axis[i].nel = (int)ax->nel; t.axis[i].nnp = (int)ax->nnp; t.axis[i].U = ax->U;
      t.axis[i].nqp = (int)bd->nqp; t.axis[i].nen = (int)bd->nen;
      /* IGXTables carries C ints.  With --with-64-bit-indices (which the metric configuration needs on one rank) the span and
         offset tables are narrowed into copies the context owns: their values are node indices of ONE axis, far below 2^31 */
      if (sizeof(PetscInt) == sizeof(int)) { t.axis[i].span = (const int*)ax->span; t.axis[i].offset = (const int*)bd->offset; }
      else {
        PetscInt e;
        PetscCall(PetscFree(c->span32[i])); PetscCall(PetscFree(c->offset32[i]));
        PetscCall(PetscMalloc1(ax->nel,&c->span32[i])); PetscCall(PetscMalloc1(bd->nel,&c->offset32[i]));
        for (e=0; e<ax->nel; e++) c->span32[i][e] = (int)ax->span[e];
        for (e=0; e<bd->nel; e++) c->offset32[i][e] = (int)bd->offset[e];
        t.axis[i].span = c->span32[i]; t.axis[i].offset = c->offset32[i];
      }
      t.axis[i].detJac = bd->detJac; t.axis[i].weight = bd->weight; t.axis[i].point = bd->point;
      t.axis[i].value = bd->value;                    /* [nel][nqp][nen][5], src/petigabasis.c:192-196 */
    }
    for (i=0; i<3; i++) {
      t.proc_sizes[i] = (int)iga->proc_sizes[i];   t.proc_ranks[i] = (int)iga->proc_ranks[i];
      t.elem_sizes[i] = (int)iga->elem_sizes[i];   t.elem_start[i] = (int)iga->elem_start[i];   t.elem_width[i] = (int)iga->elem_width[i];
      t.node_sizes[i] = (int)iga->node_sizes[i];   t.node_lstart[i] = (int)iga->node_lstart[i]; t.node_lwidth[i] = (int)iga->node_lwidth[i];
      t.node_gstart[i] = (int)iga->node_gstart[i]; t.node_gwidth[i] = (int)iga->node_gwidth[i];
    }
    t.nsd = (int)iga->geometry; t.rational = (int)iga->rational;
    t.geometryX = iga->geometryX; t.rationalW = iga->rationalW;      /* ghosted local arrays, src/petigaelem.c:733-747 */
    t.property = (int)iga->property; t.propertyA = iga->propertyA;   /* include/petiga.h:350-353 (PetscScalar = double: real builds) */
    IGXCHK(comm,IGXCreateFromTables(&t,&c->igx));
    IGXCHK(comm,IGXCreateMat(c->igx,&c->A));
    IGXCHK(comm,IGXCreateVec(c->igx,&c->b));
    IGXCHK(comm,IGXCreateVec(c->igx,&c->U));
    IGXCHK(comm,IGXCreateVec(c->igx,&c->V));
  }
  /* the boundary tables of the IGAForm (include/petiga.h:220-225) and the form, every call: the user may change them */
  {
    PetscInt d,s,k;
    IGXCHK(comm,IGXClearBoundary(c->igx));
    for (d=0; d<iga->dim; d++) for (s=0; s<2; s++) {
      IGAFormBC bv = iga->form->value[d][s], bl = iga->form->load[d][s];
      for (k=0; k<bv->count; k++) IGXCHK(comm,IGXSetBoundaryValue(c->igx,(int)d,(int)s,(int)bv->field[k],(double)bv->value[k]));
      for (k=0; k<bl->count; k++) IGXCHK(comm,IGXSetBoundaryLoad (c->igx,(int)d,(int)s,(int)bl->field[k],(double)bl->value[k]));
      IGXCHK(comm,IGXSetBoundaryForm(c->igx,(int)d,(int)s,(int)iga->form->visit[d][s]));
    }
    if (iga->fixtable) {   /* IGASetFixTable (src/petigaform.c:273): the ghosted local array of the table's Vec */
      IGXCHK(comm,IGXVecCopyFromGhosted(c->U,iga->fixtableU,0));
      IGXCHK(comm,IGXSetFixTable(c->igx,c->U));
    } else IGXCHK(comm,IGXSetFixTable(c->igx,NULL));
    if (c->source) IGXCHK(comm,IGXSetFormSource(c->igx,c->source,c->struct_name,c->params,(int)c->nparams));   /* compiled once per source text */
    else           IGXCHK(comm,IGXSetForm(c->igx,c->kind,c->params,(int)c->nparams));
  }
  *out = c;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* state vector -> engine: the ghosted local form IS the engine's row numbering (IGAGetLocalVecArray, src/petigavec.c:256-269) */
static PetscErrorCode IGAAmdSetState(IGA iga,IGAAmdCtx *c,Vec vecU,IGXVec U)
{
  Vec localU; const PetscScalar *arrayU; PetscMemType mtype;
  PetscFunctionBegin;
  /* IGAGetLocalVecArray = IGAGetLocalVec + DMGlobalToLocal + VecGetArrayRead (src/petigavec.c:256-269); the array is taken with
     its memory type instead, so that a device Vec's values never visit the host */
  PetscCall(IGAGetLocalVec(iga,&localU));
  PetscCall(IGAGlobalToLocal(iga,vecU,localU,INSERT_VALUES));      /* (src/petigavec.c:266) */
  PetscCall(VecGetArrayReadAndMemType(localU,&arrayU,&mtype));
  IGXCHK(PetscObjectComm((PetscObject)iga),IGXVecCopyFromGhosted(U,(const double*)arrayU,PetscMemTypeDevice(mtype) ? 1 : 0));
  /* a device copy is a kernel on the engine's stream: done before PETSc gets its array back (include/petiga_amd.h) */
  if (PetscMemTypeDevice(mtype)) IGXCHK(PetscObjectComm((PetscObject)iga),IGXSynchronize(c->igx));
  PetscCall(VecRestoreArrayReadAndMemType(localU,&arrayU));
  PetscCall(IGARestoreLocalVec(iga,&localU));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* engine -> Mat: coordinate list once, values from the device pointer every time */
static PetscErrorCode IGAAmdHandBackMat(IGA iga,IGAAmdCtx *c,Mat mat)
{
  MPI_Comm comm = PetscObjectComm((PetscObject)iga); double *val; int64_t nb,nblk; int bs; PetscContainer tag;
  PetscFunctionBegin;
  IGXCHK(comm,IGXMatGetInfo(c->A,&nb,&nblk,&bs));
  PetscCall(PetscObjectQuery((PetscObject)mat,"IGAAmdCOO",(PetscObject*)&tag));
  if (!tag) {
    PetscCount n = (PetscCount)nblk*bs*bs; void *di,*dj; PetscBool device_mat; MatType mtype;
    PetscCall(MatGetType(mat,&mtype));
    PetscCall(PetscStrendswith(mtype,"hipsparse",&device_mat));
    if (!device_mat) PetscCall(PetscStrendswith(mtype,"kokkos",&device_mat));
    if (device_mat) {
      /* the lists stay on the device, in PetscInt's width (a 32-bit PetscInt whose range the problem exceeds is refused by the
         library with PETSC_ERR_ARG_OUTOFRANGE); MatSetPreallocationCOO of the device types takes device pointers (PETSc >= 3.18) */
      IGXCHK(comm,IGXMatGetCOODevice(c->A,1 /* PETSc numbering = iga->ao */,0 /* PETSc moves the not-owned rows */,(int)sizeof(PetscInt),&di,&dj));
      PetscCall(MatSetPreallocationCOO(mat,n,(PetscInt*)di,(PetscInt*)dj));
      IGXCHK(comm,IGXMatFreeCOO(c->A));
    } else {
      /* host Mat type: host lists, fetched from the library in chunks through one bounded staging pair */
      PetscInt *pi,*pj; int64_t *ci,*cj; PetscCount e,e0; const PetscCount chunk = (PetscCount)1 << 28;
      PetscCall(PetscMalloc2(n,&pi,n,&pj));
      IGXCHK(comm,IGXMatGetCOODevice(c->A,1,0,8,&di,&dj));
      PetscCall(PetscMalloc2(PetscMin(n,chunk),&ci,PetscMin(n,chunk),&cj));
      for (e0=0; e0<n; e0+=chunk) {
        const PetscCount m = PetscMin(chunk,n-e0);
        IGXCHK(comm,IGXDeviceToHost(ci,(const int64_t*)di+e0,(size_t)m*8));
        IGXCHK(comm,IGXDeviceToHost(cj,(const int64_t*)dj+e0,(size_t)m*8));
        for (e=0; e<m; e++) { pi[e0+e] = (PetscInt)ci[e]; pj[e0+e] = (PetscInt)cj[e]; }
      }
      PetscCall(PetscFree2(ci,cj));
      IGXCHK(comm,IGXMatFreeCOO(c->A));
      PetscCall(MatSetPreallocationCOO(mat,n,pi,pj));
      PetscCall(PetscFree2(pi,pj));
    }
    PetscCall(PetscContainerCreate(comm,&tag));
    PetscCall(PetscObjectCompose((PetscObject)mat,"IGAAmdCOO",(PetscObject)tag));
    PetscCall(PetscContainerDestroy(&tag));
  }
  IGXCHK(comm,IGXSynchronize(c->igx));                    /* also reports a non-positive Jacobian (PETSC_ERR_USER, src/petigaelem.c:989) */
  IGXCHK(comm,IGXMatGetDeviceArrays(c->A,NULL,NULL,&val));
  PetscCall(MatSetValuesCOO(mat,(const PetscScalar*)val,INSERT_VALUES));   /* device pointer: no host copy with a device Mat type */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode IGAAmdHandBackVec(IGA iga,IGAAmdCtx *c,Vec vec)
{
  MPI_Comm comm = PetscObjectComm((PetscObject)iga); double *a; int64_t n; PetscContainer tag;
  PetscFunctionBegin;
  IGXCHK(comm,IGXVecGetSize(c->b,&n));
  PetscCall(PetscObjectQuery((PetscObject)vec,"IGAAmdCOO",(PetscObject*)&tag));
  if (!tag) {
    int64_t *idx; PetscInt *pidx; PetscCount e;
    PetscCall(PetscMalloc1(n,&idx));
    IGXCHK(comm,IGXVecGetIndices(c->b,1,0,idx,0));
    if (sizeof(PetscInt) == sizeof(int64_t)) pidx = (PetscInt*)idx;
    else { PetscCall(PetscMalloc1(n,&pidx)); for (e=0; e<n; e++) pidx[e] = (PetscInt)idx[e]; }
    PetscCall(VecSetPreallocationCOO(vec,(PetscCount)n,pidx));
    if ((void*)pidx != (void*)idx) PetscCall(PetscFree(pidx));
    PetscCall(PetscFree(idx));
    PetscCall(PetscContainerCreate(comm,&tag));
    PetscCall(PetscObjectCompose((PetscObject)vec,"IGAAmdCOO",(PetscObject)tag));
    PetscCall(PetscContainerDestroy(&tag));
  }
  IGXCHK(comm,IGXSynchronize(c->igx));
  IGXCHK(comm,IGXVecGetDeviceArray(c->b,&a));
  PetscCall(VecSetValuesCOO(vec,(const PetscScalar*)a,INSERT_VALUES));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- the seven drivers.  IGXTRY: PETSC_ERR_SUP from the engine (no device kernel covers the case: a forced kernel, a boundary-
 *      form pass of a run-time form, ...) hands the call to the original body, kept as IGACompute*_CPU; other errors are raised. ---- */
#define IGAAMD_BEGIN(iga) \
  IGAAmdCtx *c; MPI_Comm comm = PetscObjectComm((PetscObject)(iga)); \
  PetscFunctionBegin; \
  PetscValidHeaderSpecific(iga,IGA_CLASSID,1); IGACheckSetUp(iga,1); \
  PetscCall(IGAGetAmd(iga,&c))

PetscErrorCode IGAComputeSystem(IGA iga,Mat matA,Vec vecB)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeSystem_CPU(iga,matA,vecB));
  IGACheckFormOp(iga,1,System);
  IGXTRY(comm,IGXComputeSystem(c->igx,c->A,c->b),IGAComputeSystem_CPU(iga,matA,vecB));
  PetscCall(IGAAmdHandBackMat(iga,c,matA));
  PetscCall(IGAAmdHandBackVec(iga,c,vecB));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeMatrix(IGA iga,Mat matA)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeMatrix_CPU(iga,matA));
  IGXTRY(comm,IGXComputeMatrix(c->igx,c->A),IGAComputeMatrix_CPU(iga,matA));
  PetscCall(IGAAmdHandBackMat(iga,c,matA));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeVector(IGA iga,Vec vecB)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeVector_CPU(iga,vecB));
  IGXTRY(comm,IGXComputeVector(c->igx,c->b),IGAComputeVector_CPU(iga,vecB));
  PetscCall(IGAAmdHandBackVec(iga,c,vecB));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeFunction(IGA iga,Vec vecU,Vec vecF)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeFunction_CPU(iga,vecU,vecF));
  PetscCall(IGAAmdSetState(iga,c,vecU,c->U));
  IGXTRY(comm,IGXComputeFunction(c->igx,c->U,c->b),IGAComputeFunction_CPU(iga,vecU,vecF));
  PetscCall(IGAAmdHandBackVec(iga,c,vecF));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeJacobian(IGA iga,Vec vecU,Mat matJ)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeJacobian_CPU(iga,vecU,matJ));
  PetscCall(IGAAmdSetState(iga,c,vecU,c->U));
  IGXTRY(comm,IGXComputeJacobian(c->igx,c->U,c->A),IGAComputeJacobian_CPU(iga,vecU,matJ));
  PetscCall(IGAAmdHandBackMat(iga,c,matJ));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeIFunction(IGA iga,PetscReal a,Vec vecV,PetscReal t,Vec vecU,Vec vecF)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeIFunction_CPU(iga,a,vecV,t,vecU,vecF));
  PetscCall(IGAAmdSetState(iga,c,vecV,c->V));
  PetscCall(IGAAmdSetState(iga,c,vecU,c->U));
  IGXTRY(comm,IGXComputeIFunction(c->igx,(double)a,c->V,(double)t,c->U,c->b),IGAComputeIFunction_CPU(iga,a,vecV,t,vecU,vecF));
  PetscCall(IGAAmdHandBackVec(iga,c,vecF));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeIJacobian(IGA iga,PetscReal a,Vec vecV,PetscReal t,Vec vecU,Mat matJ)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeIJacobian_CPU(iga,a,vecV,t,vecU,matJ));
  PetscCall(IGAAmdSetState(iga,c,vecV,c->V));
  PetscCall(IGAAmdSetState(iga,c,vecU,c->U));
  IGXTRY(comm,IGXComputeIJacobian(c->igx,(double)a,c->V,(double)t,c->U,c->A),IGAComputeIJacobian_CPU(iga,a,vecV,t,vecU,matJ));
  PetscCall(IGAAmdHandBackMat(iga,c,matJ));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* IGAComputeScalar (src/petigacomp.c:35-98) with the point functional given as device source: the program passes the struct next
 * to (or instead of) its host `Scalar` callback; the rank-local sums come from the engine, the reduction over the ranks is the
 * reference's own (:93). */
PetscErrorCode IGAComputeScalarSourceAMD(IGA iga,Vec vecU,PetscInt n,PetscScalar S[],const char source[],const char struct_name[],const PetscReal params[],PetscInt nparams)
{
  double local[64];
  IGAAMD_BEGIN(iga);
  PetscCheck(c,comm,PETSC_ERR_ARG_WRONGSTATE,"Must call IGASetFormAMD() or IGASetFormSourceAMD() first");
  PetscCheck(n >= 1 && n <= 64,comm,PETSC_ERR_ARG_OUTOFRANGE,"Number of scalars must be in range [1,64], got %" PetscInt_FMT,n);
  if (vecU) PetscCall(IGAAmdSetState(iga,c,vecU,c->U));
  IGXCHK(comm,IGXComputeScalarSource(c->igx,vecU ? c->U : NULL,source,struct_name,(const double*)params,(int)nparams,(int)n,local));
  PetscCallMPI(MPI_Allreduce(local,S,(PetscMPIInt)n,MPIU_SCALAR,MPIU_SUM,comm));      /* as src/petigacomp.c:84 (MPIU_Allreduce returns a PetscErrorCode before PETSc 3.22) */
  PetscFunctionReturn(PETSC_SUCCESS);
}
#endif /* PETIGA_HAVE_AMD */
