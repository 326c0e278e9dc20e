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
 * Data flow per assembly (nothing is copied through the host when the Mat / Vec are device types):
 *   U,V (global Vec) --IGAGetLocalVecArray--> ghosted local array --IGXVecCopyFromGhosted--> IGXVec        (state)
 *   IGXCompute*  ->  IGXMat values / IGXVec on the device
 *   MatSetValuesCOO(A, engine value array, INSERT_VALUES)  with the coordinate list set once (IGXMatGetCOO, PETSc numbering):
 *   PETSc adds the duplicate (i,j) of different ranks and routes rows of not-owned nodes to their owners -- the stash
 *   traffic of MatAssemblyBegin/End (src/petigaksp.c:197-198), now on the device.  (Alternative: IGXReduceGhostRows over RCCL
 *   first, coordinate list with owned_only = 1.)
 *
 * The point callback: host function pointers cannot run on the GPU.  The user program registers the device form next to
 * its host callback,  IGASetFormAMD(iga, IGX_FORM_ELASTICITY, (PetscReal[]){lambda,mu}, 2);  when none is registered (or the
 * engine answers PETSC_ERR_SUP) the driver falls through to PetIGA's own CPU loop, so every program keeps working.
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
} IGAAmdCtx;

static PetscErrorCode IGAAmdCtxDestroy(void *p)
{
  IGAAmdCtx *c = (IGAAmdCtx*)p;
  PetscFunctionBegin;
  if (c) { IGXVecDestroy(&c->V); IGXVecDestroy(&c->U); IGXVecDestroy(&c->b); IGXMatDestroy(&c->A); IGXDestroy(&c->igx); PetscCall(PetscFree(c)); }
  PetscFunctionReturn(PETSC_SUCCESS);
}

#define IGXCHK(comm,call) do { int rc_ = (call); if (PetscUnlikely(rc_)) SETERRQ(comm,(PetscErrorCode)rc_,"%s",IGXGetLastError()); } while (0)

/* user-facing: which device form stands for the host callback of this IGA */
PetscErrorCode IGASetFormAMD(IGA iga,IGXFormKind kind,const PetscReal params[],PetscInt nparams)
{
  IGAAmdCtx *c; PetscContainer box; PetscInt i;
  PetscFunctionBegin;
  PetscValidHeaderSpecific(iga,IGA_CLASSID,1);
  PetscCheck(nparams >= 0 && nparams <= 8,PetscObjectComm((PetscObject)iga),PETSC_ERR_ARG_OUTOFRANGE,"at most 8 form parameters");
  PetscCall(PetscObjectQuery((PetscObject)iga,"IGAAmdCtx",(PetscObject*)&box));
  if (!box) {
    PetscCall(PetscNew(&c));
    PetscCall(PetscContainerCreate(PetscObjectComm((PetscObject)iga),&box));
    PetscCall(PetscContainerSetPointer(box,c));
    PetscCall(PetscContainerSetUserDestroy(box,IGAAmdCtxDestroy));
    PetscCall(PetscObjectCompose((PetscObject)iga,"IGAAmdCtx",(PetscObject)box));
    PetscCall(PetscContainerDestroy(&box));
  } else PetscCall(PetscContainerGetPointer(box,(void**)&c));
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
      t.axis[i].nel = (int)ax->nel; t.axis[i].nnp = (int)ax->nnp; t.axis[i].U = ax->U; t.axis[i].span = (const int*)ax->span; /* 32-bit PetscInt; widen otherwise */
      t.axis[i].nqp = (int)bd->nqp; t.axis[i].nen = (int)bd->nen; t.axis[i].offset = (const int*)bd->offset;
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
    IGXCHK(comm,IGXSetForm(c->igx,c->kind,c->params,(int)c->nparams));
  }
  *out = c;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* state vector -> engine: the ghosted local form IS the engine's row numbering (IGAGetLocalVecArray, src/petigavec.c:256-269) */
static PetscErrorCode IGAAmdSetState(IGA iga,Vec vecU,IGXVec U)
{
  Vec localU; const PetscScalar *arrayU;
  PetscFunctionBegin;
  PetscCall(IGAGetLocalVecArray(iga,vecU,&localU,&arrayU));
  IGXCHK(PetscObjectComm((PetscObject)iga),IGXVecCopyFromGhosted(U,(const double*)arrayU,0));
  PetscCall(IGARestoreLocalVecArray(iga,vecU,&localU,&arrayU));
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
    PetscCount n = (PetscCount)nblk*bs*bs; int64_t *ci,*cj; PetscInt *pi,*pj; PetscCount e;
    PetscCall(PetscMalloc2(n,&ci,n,&cj));
    IGXCHK(comm,IGXMatGetCOO(c->A,1 /* PETSc numbering = iga->ao */,0 /* PETSc moves the not-owned rows */,ci,cj,0));
    if (sizeof(PetscInt) == sizeof(int64_t)) { pi = (PetscInt*)ci; pj = (PetscInt*)cj; }
    else { PetscCall(PetscMalloc2(n,&pi,n,&pj)); for (e=0; e<n; e++) { pi[e] = (PetscInt)ci[e]; pj[e] = (PetscInt)cj[e]; } }
    PetscCall(MatSetPreallocationCOO(mat,n,pi,pj));
    if ((void*)pi != (void*)ci) PetscCall(PetscFree2(pi,pj));
    PetscCall(PetscFree2(ci,cj));
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

/* ---- the seven drivers.  Each returns PETSC_ERR_SUP untouched from the engine when the case is not covered (the caller,
 *      the original body kept as IGACompute*_CPU, then runs PetIGA's own loop). ---- */
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
  IGXCHK(comm,IGXComputeSystem(c->igx,c->A,c->b));
  PetscCall(IGAAmdHandBackMat(iga,c,matA));
  PetscCall(IGAAmdHandBackVec(iga,c,vecB));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeMatrix(IGA iga,Mat matA)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeMatrix_CPU(iga,matA));
  IGXCHK(comm,IGXComputeMatrix(c->igx,c->A));
  PetscCall(IGAAmdHandBackMat(iga,c,matA));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeVector(IGA iga,Vec vecB)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeVector_CPU(iga,vecB));
  IGXCHK(comm,IGXComputeVector(c->igx,c->b));
  PetscCall(IGAAmdHandBackVec(iga,c,vecB));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeFunction(IGA iga,Vec vecU,Vec vecF)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeFunction_CPU(iga,vecU,vecF));
  PetscCall(IGAAmdSetState(iga,vecU,c->U));
  IGXCHK(comm,IGXComputeFunction(c->igx,c->U,c->b));
  PetscCall(IGAAmdHandBackVec(iga,c,vecF));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeJacobian(IGA iga,Vec vecU,Mat matJ)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeJacobian_CPU(iga,vecU,matJ));
  PetscCall(IGAAmdSetState(iga,vecU,c->U));
  IGXCHK(comm,IGXComputeJacobian(c->igx,c->U,c->A));
  PetscCall(IGAAmdHandBackMat(iga,c,matJ));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeIFunction(IGA iga,PetscReal a,Vec vecV,PetscReal t,Vec vecU,Vec vecF)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeIFunction_CPU(iga,a,vecV,t,vecU,vecF));
  PetscCall(IGAAmdSetState(iga,vecV,c->V));
  PetscCall(IGAAmdSetState(iga,vecU,c->U));
  IGXCHK(comm,IGXComputeIFunction(c->igx,(double)a,c->V,(double)t,c->U,c->b));
  PetscCall(IGAAmdHandBackVec(iga,c,vecF));
  PetscFunctionReturn(PETSC_SUCCESS);
}
PetscErrorCode IGAComputeIJacobian(IGA iga,PetscReal a,Vec vecV,PetscReal t,Vec vecU,Mat matJ)
{
  IGAAMD_BEGIN(iga);
  if (!c) PetscFunctionReturn(IGAComputeIJacobian_CPU(iga,a,vecV,t,vecU,matJ));
  PetscCall(IGAAmdSetState(iga,vecV,c->V));
  PetscCall(IGAAmdSetState(iga,vecU,c->U));
  IGXCHK(comm,IGXComputeIJacobian(c->igx,(double)a,c->V,(double)t,c->U,c->A));
  PetscCall(IGAAmdHandBackMat(iga,c,matJ));
  PetscFunctionReturn(PETSC_SUCCESS);
}
#endif /* PETIGA_HAVE_AMD */
