/*
 * igaoracle.h -- CPU restatement of PetIGA's element-assembly path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under petiga_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the timed
 * CPU baseline.
 *
 * Parity pin: the reference (dalcinl/PetIGA @ 2025-04-04) cannot be built in
 * this image (it needs PETSc headers/libraries, and its Fortran kernels need
 * PETSc's generated petscconf.h), so this restatement is pinned against the
 * known-answer tests the reference itself holds for this path
 * (test/IGAGeometryMap.c, test/IGAErrNorm.c, test/IGAFixTable.c,
 * test/IGACreate.c, docs/manual/TUTORIAL.rst sizes) -- see tests/test_oracle_*.py.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference root).
 */
#ifndef IGAORACLE_H
#define IGAORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int     p;        /* polynomial degree                         */
  int     m;        /* last knot index                           */
  double *U;        /* knots U[0..m]                             */
  int     periodic;
  int     nel;      /* non-empty spans                           */
  int     nnp;      /* basis functions (wrapped when periodic)   */
  int    *span;     /* span[e] = knot index of element e         */
} OrcAxis;          /* include/petiga.h:80-96 (struct _n_IGAAxis) */

typedef struct {
  int     nel, nqp, nen;
  int    *offset;   /* [nel]  first basis function of element    */
  double *detJac;   /* [nel]                                     */
  double *weight;   /* [nel][nqp]                                */
  double *point;    /* [nel][nqp]                                */
  double *value;    /* [nel][nqp][nen][5]                        */
  double  bnd_point[2], bnd_weight, bnd_detJac;
  double *bnd_value[2]; /* [nen][5] */
} OrcBasis;         /* include/petiga.h:122-141 (struct _n_IGABasis) */

typedef struct {
  int    count;
  int    field[64];
  double value[64];
} OrcBC;            /* include/petiga.h:220-225 (struct _IGAFormBC) */

typedef struct OrcPoint OrcPoint;
typedef struct OrcIGA   OrcIGA;

/* Point callbacks: include/petiga.h:153-197 */
typedef int (*OrcFormSystem)   (OrcPoint*,double*K,double*F,void*ctx);
typedef int (*OrcFormMatrix)   (OrcPoint*,double*K,void*ctx);
typedef int (*OrcFormVector)   (OrcPoint*,double*F,void*ctx);
typedef int (*OrcFormFunction) (OrcPoint*,const double*U,double*F,void*ctx);
typedef int (*OrcFormJacobian) (OrcPoint*,const double*U,double*J,void*ctx);
typedef int (*OrcFormIFunction)(OrcPoint*,double a,const double*V,double t,const double*U,double*F,void*ctx);
typedef int (*OrcFormIJacobian)(OrcPoint*,double a,const double*V,double t,const double*U,double*J,void*ctx);
typedef int (*OrcFormScalar)   (OrcPoint*,const double*U,int n,double*S,void*ctx);

/* A quadrature-point view: include/petiga.h:644-703 (struct _n_IGAPoint) */
struct OrcPoint {
  OrcIGA *iga;
  int     atboundary, boundary_id;
  int     count, index;
  int     neq, nen, dof, dim, nsd;
  const double *rational, *geometry;     /* element W[nen], X[nen][nsd] or NULL */
  double *weight, *detJac;
  double *point, *normal;
  double *basis[5], *shape[5];
  double *mapU[5], *mapX[5];
  double *detX, *detS;
  int     ID[3];
  const double *property;                /* element A[nen][npd] or NULL (include/petiga.h:662) */
  int     npd;
};

struct OrcIGA {
  int dim, dof, order;
  int nsd;          /* geometry dimension, 0 = no geometry */
  int rational;
  OrcAxis  axis[3];
  int      rule_nqp[3];
  OrcBasis basis[3];
  int proc_sizes[3], proc_ranks[3];
  int elem_sizes[3], elem_start[3], elem_width[3];
  int node_sizes[3], node_lstart[3], node_lwidth[3], node_gstart[3], node_gwidth[3];
  double *geometryX;   /* ghosted local [gw2][gw1][gw0][nsd] */
  double *rationalW;   /* ghosted local [gw2][gw1][gw0]      */
  OrcBC value[3][2], load[3][2];
  int   visit[3][2];
  int   fixtable; double *fixtableU;  /* ghosted local [..][dof] */
  int   setup;
  int   rule_type[3];      /* IGARuleType: 0 Legendre, 1 Lobatto, 3 user (include/petiga.h:82-87) */
  int   rule_user_n[3];
  double *rule_x[3], *rule_w[3];   /* user-defined rule on [-1,1] */
  int    property;         /* iga->property: numbers per node, 0 = none (include/petiga.h:350) */
  double *propertyA;       /* ghosted local [gw2][gw1][gw0][npd] (include/petiga.h:353) */
};

/* global (natural-order) CSR: rows = node*dof + c, i0 fastest */
typedef struct {
  int64_t  nrows;
  int64_t *rowptr;
  int32_t *colidx;
  double  *val;
} OrcMat;

/* ---- discretisation ---- */
OrcIGA *orc_create(int dim,int dof);
void    orc_destroy(OrcIGA*);
int     orc_axis_set_degree(OrcIGA*,int i,int p);
int     orc_axis_set_periodic(OrcIGA*,int i,int flag);
int     orc_axis_init_uniform(OrcIGA*,int i,int N,double Ui,double Uf,int C);
int     orc_axis_set_knots(OrcIGA*,int i,int m,const double *U);
int     orc_set_quadrature(OrcIGA*,int i,int q);
int     orc_set_rule_type(OrcIGA*,int i,int type);
int     orc_set_rule(OrcIGA*,int i,int q,const double *x,const double *w);
int     orc_set_order(OrcIGA*,int order);
int     orc_set_partition(OrcIGA*,int size,int rank);
int     orc_setup(OrcIGA*);
int     orc_set_geometry(OrcIGA*,int nsd,const double *Xglobal,const double *Wglobal);
int     orc_set_property(OrcIGA*,int npd,const double *Aglobal);   /* IGASetPropertyDim + the array IGALoadProperty fills (src/petigaio.c:359-458) */
int     orc_set_boundary_value(OrcIGA*,int axis,int side,int field,double v);
int     orc_set_boundary_load (OrcIGA*,int axis,int side,int field,double v);
int     orc_set_boundary_form (OrcIGA*,int axis,int side,int flag);
int     orc_clear_boundary(OrcIGA*);
int     orc_set_fixtable(OrcIGA*,const double *Uglobal);

/* ---- building blocks exposed for known-answer tests ---- */
int     orc_gauss_legendre(int q,double *X,double *W);
int     orc_gauss_lobatto(int q,double *X,double *W);
void    orc_bspline_ders(int k,double u,int p,int d,const double *U,double *B /*[p+1][5]*/);
int     orc_partition(int size,int rank,int dim,const int N[],int n[],int i[]);
void    orc_distribute(int dim,const int size[],const int rank[],const int N[],int n[],int s[]);
int64_t orc_global_size(const OrcIGA*);

/* ---- sparsity ---- */
OrcMat *orc_mat_create(const OrcIGA*);
void    orc_mat_destroy(OrcMat*);
void    orc_mat_zero(OrcMat*);

/* ---- assembly drivers (this rank's elements into the global natural CSR / vector) ---- */
int orc_compute_system   (OrcIGA*,OrcFormSystem,void*ctx,OrcMat*A,double*B);
int orc_compute_matrix   (OrcIGA*,OrcFormMatrix,void*ctx,OrcMat*A);
int orc_compute_vector   (OrcIGA*,OrcFormVector,void*ctx,double*B);
int orc_compute_function (OrcIGA*,OrcFormFunction,void*ctx,const double*U,double*F);
int orc_compute_jacobian (OrcIGA*,OrcFormJacobian,void*ctx,const double*U,OrcMat*J);
int orc_compute_ifunction(OrcIGA*,OrcFormIFunction,void*ctx,double a,const double*V,double t,const double*U,double*F);
int orc_compute_ijacobian(OrcIGA*,OrcFormIJacobian,void*ctx,double a,const double*V,double t,const double*U,OrcMat*J);
int orc_compute_scalar   (OrcIGA*,const double*U,int n,double*S,OrcFormScalar,void*ctx,int full);

/* element-level probe: tabulate one element (interior pass or a boundary face)
 * and copy out its arrays; K_e/F_e of a named form before/after BC fix-up. */
typedef struct {
  int nqp, nen, dim, nsd;
  double *weight,*detJac,*point,*normal,*detX,*detS;
  double *basis[5],*shape[5],*mapU[5],*mapX[5];
  int    *mapping;
  const double *geometryX,*rationalW;
} OrcElemView;
int orc_element_tabulate(OrcIGA*,const int ID[3],int boundary_id,OrcElemView*out);

/* ---- restated demo / test forms (oracle/igaforms.c) ---- */
typedef struct { double lambda,mu; } OrcElasticityCtx;
typedef struct { double theta,alpha,cbar,L0,lambda,tau; } OrcCahnHilliardCtx;
typedef struct { double nu,fx,fy,fz,dt; } OrcNSVMSCtx;
int orc_form_poisson   (OrcPoint*,double*,double*,void*);  /* demo/Poisson{1,2,3}D.c System */
int orc_form_mass      (OrcPoint*,double*,double*,void*);  /* test/IGACreate.c System: M, int N */
int orc_form_l2proj_x2 (OrcPoint*,double*,double*,void*);  /* test/IGAFixTable.c System1 */
int orc_form_poisson_f (OrcPoint*,double*,double*,void*);  /* test/IGAFixTable.c System2 */
int orc_form_boundary_integral(OrcPoint*,double*,double*,void*);  /* demo/BoundaryIntegral.c System */
int orc_form_nitsche(OrcPoint*,double*,double*,void*);            /* demo/NitscheMethod.c System; ctx = int* degree */
int orc_form_errnorm   (OrcPoint*,double*,double*,void*);  /* test/IGAErrNorm.c System (dof=4) */
int orc_form_elasticity(OrcPoint*,double*,double*,void*);  /* demo/Elasticity3D.c System */
int orc_form_der3      (OrcPoint*,double*,double*,void*);  /* third derivatives p->shape[3] (test/IGAGeometryMap.c:179,221); ctx: double[3] = {k3, f3, u3} */
int orc_form_der3_function(OrcPoint*,const double*,double*,void*);  /* ... with IGAPointFormDer3 of U (demo/AutoDiff/CahnHilliardPrimalFAD.cxx:51) */
int orc_form_surface   (OrcPoint*,double*,double*,void*);  /* Laplace-Beltrami + mass and a curvature load from p->mapX[1], p->mapX[2] (as demo/ClassicalShell.c:57-80 reads them), nsd != dim */
int orc_form_property  (OrcPoint*,double*,double*,void*);  /* Poisson with conductivity / source from p->property (include/petiga.h:662) */
int orc_form_elasticity_f(OrcPoint*,double*,double*,void*);  /* the same K with a body force; ctx: double[5] = {lambda, mu, fx, fy, fz} */
int orc_form_ch_residual(OrcPoint*,double,const double*,double,const double*,double*,void*);
int orc_form_ch_tangent (OrcPoint*,double,const double*,double,const double*,double*,void*);
int orc_form_ns_residual(OrcPoint*,double,const double*,double,const double*,double*,void*);
int orc_form_ns_tangent (OrcPoint*,double,const double*,double,const double*,double*,void*);
int orc_form_advection_diffusion(OrcPoint*,double*,double*,void*);     /* demo/AdvectionDiffusion.c System; ctx = double wind[3] */
int orc_form_bratu_function (OrcPoint*,const double*,double*,void*);   /* demo/BratuFJ.F90 Bratu_Function; ctx = double* lambda */
int orc_form_bratu_jacobian (OrcPoint*,const double*,double*,void*);   /* demo/BratuFJ.F90 Bratu_Jacobian */
int orc_form_bratu_ifunction(OrcPoint*,double,const double*,double,const double*,double*,void*);
int orc_form_bratu_ijacobian(OrcPoint*,double,const double*,double,const double*,double*,void*);
int orc_scalar_errnorm  (OrcPoint*,const double*,int,double*,void*); /* ctx = int* order; test/IGAErrNorm.c Exact */
int orc_scalar_x2err    (OrcPoint*,const double*,int,double*,void*); /* test/IGAFixTable.c Exact, L2 */
int orc_scalar_volume   (OrcPoint*,const double*,int,double*,void*); /* test/IGAGeometryMap.c Scalar */

/* field interpolation at a point: src/petigaval.F90:182-251 */
void orc_point_value(const OrcPoint*,const double*U,double*u);
void orc_point_grad (const OrcPoint*,const double*U,double*u);
void orc_point_hess (const OrcPoint*,const double*U,double*u);
void orc_point_der3 (const OrcPoint*,const double*U,double*u);   /* IGAPointFormDer3, include/petiga.h:731 */
void orc_point_del2 (const OrcPoint*,const double*U,double*u);
void orc_point_geommap(const OrcPoint*,double*x);
void orc_point_invgradgeommap(const OrcPoint*,double*G);

#ifdef __cplusplus
}
#endif
#endif
