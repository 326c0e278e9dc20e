/*
 * igaforms.c -- point callbacks of the reference's demos/tests, restated for the oracle.
 * TEST INFRASTRUCTURE ONLY (see igaoracle.h).
 *
 * Contract (include/petiga.h:153-197, src/petigapoint.c:427-462): K is [nen][dof][nen][dof]
 * row-major, F is [nen][dof]; both arrive zeroed; the callback returns the UN-weighted
 * integrand, the driver multiplies by detJac*weight.
 */
#include "igaoracle.h"
#include <math.h>
#include <string.h>

/* demo/Poisson3D.c:3-23, demo/Poisson2D.c:3-21, demo/Poisson1D.c: K = grad Na . grad Nb, F = Na*1 */
int orc_form_poisson(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,nen=p->nen,dim=p->dim;
  const double *N0 = p->shape[0], *N1 = p->shape[1];
  (void)ctx;
  for (a=0; a<nen; a++) {
    for (b=0; b<nen; b++) {
      double s = 0;
      for (i=0; i<dim; i++) s += N1[a*dim+i]*N1[b*dim+i];
      K[a*nen+b] = s;
    }
    F[a] = N0[a]*1.0;
  }
  return 0;
}

/* test/IGACreate.c:33-60 (System): block-diagonal mass matrix and F = N (so M x = F gives x == 1) */
int orc_form_mass(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,nen=p->nen,dof=p->dof,N=nen*dof;
  const double *N0 = p->shape[0];
  (void)ctx;
  for (a=0; a<nen; a++) {
    for (b=0; b<nen; b++) for (i=0; i<dof; i++) K[(a*dof+i)*N + b*dof+i] = N0[a]*N0[b];
    for (i=0; i<dof; i++) F[a*dof+i] = N0[a];
  }
  return 0;
}

static double sum_sq(int dim,const double *x) { int i; double u=0; for (i=0;i<dim;i++) u += x[i]*x[i]; return u; }

/* test/IGAFixTable.c:25-43 (System1): L2 projection of g = sum x_i^2 */
int orc_form_l2proj_x2(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,nen=p->nen; const double *N0 = p->shape[0]; double x[3]={0,0,0},g;
  (void)ctx;
  orc_point_geommap(p,x); g = sum_sq(p->dim,x);
  for (a=0; a<nen; a++) { for (b=0; b<nen; b++) K[a*nen+b] = N0[a]*N0[b]; F[a] = N0[a]*g; }
  return 0;
}

/* test/IGAFixTable.c:45-64 (System2): Poisson with f = -2*dim */
int orc_form_poisson_f(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,nen=p->nen,dim=p->dim; const double *N0=p->shape[0],*N1=p->shape[1]; double f = -2.0*dim;
  (void)ctx;
  for (a=0; a<nen; a++) {
    for (b=0; b<nen; b++) { double s=0; for (i=0;i<dim;i++) s += N1[a*dim+i]*N1[b*dim+i]; K[a*nen+b] = s; }
    F[a] = N0[a]*f;
  }
  return 0;
}

/* A form on the third derivatives p->shape[3] ([nen][dim][dim][dim], include/petiga.h:657; read as test/IGAGeometryMap.c:179,221 does) and
 * IGAPointFormDer3 (include/petiga.h:731; demo/AutoDiff/CahnHilliardPrimalFAD.cxx:51): the engine's IGX_FORM_DER3.  ctx = {k3, f3, u3}:
 * K_ab = N_a N_b + k3 d3N_a : d3N_b,  F_a = N_a (1 + |x|^2) + f3 c : d3N_a [+ u3 N_a c : d3u],  c_ijk = 1/(1 + i + 2j + 3k). */
static int der3_body(OrcPoint *p,const double *U,double *K,double *F,void *ctx)
{
  const double *prm = (const double*)ctx;
  int a,b,i,j,k,f,nen=p->nen,dim=p->dim,d3=dim*dim*dim; const double *N0=p->shape[0],*N3=p->shape[3];
  double x[3]={0,0,0},x2=0,su=0,u3[27];
  orc_point_geommap(p,x); for (i=0;i<dim;i++) x2 += x[i]*x[i];
  if (U) { orc_point_der3(p,U,u3); for (i=0;i<dim;i++) for (j=0;j<dim;j++) for (k=0;k<dim;k++) su += u3[(i*dim+j)*dim+k]/(1.0+i+2.0*j+3.0*k); }
  for (a=0; a<nen; a++) {
    double s = 0;
    if (K) for (b=0; b<nen; b++) { double t=0; for (f=0;f<d3;f++) t += N3[a*d3+f]*N3[b*d3+f]; K[a*nen+b] = N0[a]*N0[b] + prm[0]*t; }
    for (i=0;i<dim;i++) for (j=0;j<dim;j++) for (k=0;k<dim;k++) s += N3[a*d3+(i*dim+j)*dim+k]/(1.0+i+2.0*j+3.0*k);
    F[a] = N0[a]*(1.0+x2) + prm[1]*s + prm[2]*N0[a]*su;
  }
  return 0;
}
int orc_form_der3(OrcPoint *p,double *K,double *F,void *ctx) { return der3_body(p,NULL,K,F,ctx); }
int orc_form_der3_function(OrcPoint *p,const double *U,double *F,void *ctx) { return der3_body(p,U,NULL,F,ctx); }

/* A form on a curve / surface in space (IGASetGeometryDim with nsd != dim, demo/ClassicalShell.c:154): the metric is built in the callback
 * from p->mapX[1] [nsd][dim] and p->mapX[2] [nsd][dim][dim] (demo/ClassicalShell.c:57-80), the shape functions are the parametric basis:
 *   g = F^T F,  a = sqrt det g,  K_ab = (N_a N_b + d_alpha N_a g^{alpha beta} d_beta N_b) a,  F_a = N_a |H| a,
 * H = g^{alpha beta} (d_alpha d_beta x - Gamma^gamma_{alpha beta} d_gamma x) the mean-curvature vector (1/R on a circle, 2/R on a sphere):
 * the engine's IGX_FORM_SURFACE.  ctx (optional) = double gs: K_ab += gs N_a N_b sum_{alpha i} G_{alpha i}^2 with G of IGAPointFormInvGradGeomMap
 * (src/petigapoint.c:269-294; nsd != dim: the pseudo-inverse of IGA_GetInvGradGeomMap, src/petigaval.F90:124-142). */
int orc_form_surface(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,al,be,ga,de,nen=p->nen,dim=p->dim,nsd=p->nsd,d2=dim*dim;
  const double *N0=p->basis[0],*N1=p->basis[1],*X1=p->mapX[1],*X2=p->mapX[2];
  double g[4]={0,0,0,0},gi[4],detg,ar,H[3]={0,0,0},Hn=0,gs = ctx ? *(const double*)ctx : 0.0,G[9],G2=0;
  if (!p->geometry) return 73;   /* "No geometry set" */
  orc_point_invgradgeommap(p,G); for (i=0;i<dim*nsd;i++) G2 += G[i]*G[i];
  for (al=0;al<dim;al++) for (be=0;be<dim;be++) { double t=0; for (i=0;i<nsd;i++) t += X1[i*dim+al]*X1[i*dim+be]; g[al*dim+be] = t; }
  if (dim == 1) { detg = g[0]; gi[0] = 1/g[0]; }
  else { detg = g[0]*g[3]-g[1]*g[2]; gi[0] = g[3]/detg; gi[1] = -g[1]/detg; gi[2] = -g[2]/detg; gi[3] = g[0]/detg; }
  ar = sqrt(detg);
  for (al=0;al<dim;al++) for (be=0;be<dim;be++) {
    double Gam[2]={0,0};
    for (ga=0;ga<dim;ga++) for (de=0;de<dim;de++) { double t=0; for (i=0;i<nsd;i++) t += X1[i*dim+de]*X2[i*d2+al*dim+be]; Gam[ga] += gi[ga*dim+de]*t; }
    for (i=0;i<nsd;i++) { double h = X2[i*d2+al*dim+be]; for (ga=0;ga<dim;ga++) h -= Gam[ga]*X1[i*dim+ga]; H[i] += gi[al*dim+be]*h; }
  }
  for (i=0;i<nsd;i++) Hn += H[i]*H[i];
  Hn = sqrt(Hn);
  for (a=0; a<nen; a++) {
    for (b=0; b<nen; b++) {
      double s = 0;
      for (al=0;al<dim;al++) for (be=0;be<dim;be++) s += N1[a*dim+al]*gi[al*dim+be]*N1[b*dim+be];
      K[a*nen+b] = (N0[a]*N0[b] + s)*ar + gs*N0[a]*N0[b]*G2;
    }
    F[a] = N0[a]*Hn*ar;
  }
  return 0;
}

/* Poisson with the conductivity A[.][0] and the source A[.][npd-1] of the property array, interpolated at the point from the
 * element's nodal values p->property [nen][npd] (IGAElementBuildClosure, src/petigaelem.c:745-752): the engine's IGX_FORM_PROPERTY */
int orc_form_property(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,nen=p->nen,dim=p->dim,npd=p->npd; const double *N0=p->shape[0],*N1=p->shape[1],*A=p->property; double kq=0,fq=0;
  (void)ctx;
  if (!A || npd < 1) return 73;   /* PETSC_ERR_ARG_WRONGSTATE: "No property set" (src/petigaelem.c:300) */
  for (a=0; a<nen; a++) { kq += N0[a]*A[a*npd]; fq += N0[a]*A[a*npd+npd-1]; }
  for (a=0; a<nen; a++) {
    for (b=0; b<nen; b++) { double s=0; for (i=0;i<dim;i++) s += N1[a*dim+i]*N1[b*dim+i]; K[a*nen+b] = kq*s; }
    F[a] = N0[a]*fq;
  }
  return 0;
}

/* test/IGAErrNorm.c:26-52 (Exact) for the four fields 1, sum x, sum x^2, prod x */
static void errnorm_exact(const OrcPoint *p,int order,double *v)
{
  int i,j,dim=p->dim; double x[3]={0,0,0},s1=0,s2=0,pr=1;
  orc_point_geommap(p,x);
  for (i=0;i<dim;i++) { s1 += x[i]; s2 += x[i]*x[i]; pr *= x[i]; }
  if (order == 0) { v[0]=1; v[1]=s1; v[2]=s2; v[3]=pr; }
  else if (order == 1) {
    for (i=0;i<dim;i++) { v[0*dim+i]=0; v[1*dim+i]=1; v[2*dim+i]=2*x[i]; v[3*dim+i]=pr/x[i]; }
  } else {
    for (i=0;i<dim;i++) for (j=0;j<dim;j++) {
      v[0*dim*dim+i*dim+j]=0; v[1*dim*dim+i*dim+j]=0;
      v[2*dim*dim+i*dim+j]=(i==j)?2:0;
      v[3*dim*dim+i*dim+j]=(i==j)?0:(pr/(x[i]*x[j]));
    }
  }
}

/* test/IGAErrNorm.c:54-75 (System): 4-field L2 projection, K[a][i][b][i] = Na*Nb */
int orc_form_errnorm(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,nen=p->nen,dof=p->dof,N=nen*dof; const double *N0=p->shape[0]; double u[4];
  (void)ctx;
  errnorm_exact(p,0,u);
  for (a=0; a<nen; a++) for (b=0; b<nen; b++) for (i=0; i<dof; i++) K[(a*dof+i)*N + b*dof+i] = N0[a]*N0[b];
  for (a=0; a<nen; a++) for (i=0; i<dof; i++) F[a*dof+i] = N0[a]*u[i];
  return 0;
}

/* demo/BoundaryIntegral.c:26-56 (System): Laplace inside, Neumann data 1 on the visited faces */
int orc_form_boundary_integral(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,nen=p->nen,dim=p->dim; const double *N0=p->shape[0],*N1=p->shape[1];
  (void)ctx;
  if (!p->atboundary) {
    for (a=0; a<nen; a++) {
      for (b=0; b<nen; b++) { double s=0; for (i=0;i<dim;i++) s += N1[a*dim+i]*N1[b*dim+i]; K[a*nen+b] = s; }
      F[a] = 0.0;
    }
  } else {
    for (a=0; a<nen; a++) F[a] = N0[a]*1.0;
  }
  return 0;
}

/* demo/NitscheMethod.c:48-110 (Degree, NormalMeshSize, System); ctx = int* maximum degree (Degree(p) in the demo) */
int orc_form_nitsche(OrcPoint *p,double *K,double *F,void *ctx)
{
  int a,b,i,j,nen=p->nen,dim=p->dim; const double *N0=p->shape[0],*N1=p->shape[1];
  double x[3]={0,0,0};
  orc_point_geommap(p,x);
  if (!p->atboundary) {
    double f = -2.0*dim;
    for (a=0; a<nen; a++) {
      for (b=0; b<nen; b++) { double s=0; for (i=0;i<dim;i++) s += N1[a*dim+i]*N1[b*dim+i]; K[a*nen+b] = s; }
      F[a] = N0[a]*f;
    }
  } else {
    double g = sum_sq(dim,x), G[9], N[3], s=0, k = (double)*(int*)ctx, Cc, h, alpha;
    const double *n = p->normal;
    orc_point_invgradgeommap(p,G);
    for (i=0;i<dim;i++) { N[i]=0; for (j=0;j<dim;j++) N[i] += G[i*dim+j]*n[j]; s += N[i]*N[i]; }
    h = 2/sqrt(s); Cc = 5*(k+1); alpha = Cc/h;
    for (a=0; a<nen; a++) {
      double dna=0; for (i=0;i<dim;i++) dna += N1[a*dim+i]*n[i];
      for (b=0; b<nen; b++) {
        double dnb=0; for (i=0;i<dim;i++) dnb += N1[b*dim+i]*n[i];
        K[a*nen+b] += - N0[a]*dnb;
        K[a*nen+b] += - N0[b]*dna;
        K[a*nen+b] += + alpha*N0[a]*N0[b];
      }
      F[a] += - dna*g;
      F[a] += + alpha*N0[a]*g;
    }
  }
  return 0;
}

/* src/petigacomp.c:102-120 (ErrorSqr) specialised to test/IGAErrNorm.c's Exact; ctx = int* order.
 * U == all-zero vector reproduces "norm of the exact solution" (vecU NULL in the reference). */
int orc_scalar_errnorm(OrcPoint *p,const double *U,int n,double *S,void *ctx)
{
  int order = *(int*)ctx, dim=p->dim, nc=1, i,j; double va[4*9],ve[4*9];
  for (i=0;i<order;i++) nc *= dim;
  if (order == 0) orc_point_value(p,U,va); else if (order == 1) orc_point_grad(p,U,va); else orc_point_hess(p,U,va);
  errnorm_exact(p,order,ve);
  for (i=0; i<n; i++) for (j=0; j<nc; j++) { double e = fabs(ve[i*nc+j]-va[i*nc+j]); S[i] += e*e; }
  return 0;
}

/* test/IGAFixTable.c:66-72 (Exact) through ErrorSqr, order 0, dof 1 */
int orc_scalar_x2err(OrcPoint *p,const double *U,int n,double *S,void *ctx)
{
  double x[3]={0,0,0},uh,e; (void)ctx; (void)n;
  orc_point_geommap(p,x); orc_point_value(p,U,&uh);
  e = fabs(sum_sq(p->dim,x) - uh); S[0] += e*e;
  return 0;
}

/* test/IGAGeometryMap.c:383-389 (Scalar): S[0] volume, S[1] boundary area */
int orc_scalar_volume(OrcPoint *p,const double *U,int n,double *S,void *ctx)
{
  (void)U; (void)n; (void)ctx;
  if (p->atboundary) S[1] = 1.0; else S[0] = 1.0;
  return 0;
}

/* demo/Elasticity3D.c:13-46 (System).  NB line 37 of the reference multiplies the xx term of the
 * [1][1] block by mu a second time; reproduced as is. */
int orc_form_elasticity(OrcPoint *p,double *K,double *F,void *ctx)
{
  const OrcElasticityCtx *user = (const OrcElasticityCtx*)ctx;
  double lambda = user->lambda, mu = user->mu;
  const double *N1 = p->shape[1];
  int a,b,nen=p->nen,N=nen*3;
#define KL(a,i,b,j) K[((a)*3+(i))*N + (b)*3+(j)]
  for (a=0; a<nen; a++) {
    double Na_x=N1[a*3+0],Na_y=N1[a*3+1],Na_z=N1[a*3+2];
    for (b=0; b<nen; b++) {
      double Nb_x=N1[b*3+0],Nb_y=N1[b*3+1],Nb_z=N1[b*3+2];
      KL(a,0,b,0) = Na_x*Nb_x*(lambda + 2*mu) + mu*(Na_y*Nb_y + Na_z*Nb_z);
      KL(a,0,b,1) = Na_x*Nb_y*lambda + Na_y*Nb_x*mu;
      KL(a,0,b,2) = Na_x*Nb_z*lambda + Na_z*Nb_x*mu;
      KL(a,1,b,0) = Na_x*Nb_y*mu + Na_y*Nb_x*lambda;
      KL(a,1,b,1) = Na_y*Nb_y*(lambda + 2*mu) + mu*(Na_z*Nb_z + Na_x*Nb_x*mu);
      KL(a,1,b,2) = Na_y*Nb_z*lambda + Na_z*Nb_y*mu;
      KL(a,2,b,0) = Na_x*Nb_z*mu + Na_z*Nb_x*lambda;
      KL(a,2,b,1) = Na_y*Nb_z*mu + Na_z*Nb_y*lambda;
      KL(a,2,b,2) = mu*(Na_x*Nb_x + Na_y*Nb_y) + Na_z*Nb_z*(lambda + 2*mu);
    }
    F[a] = 0.0;   /* only the first nen of 3*nen entries, as in the reference */
  }
#undef KL
  return 0;
}

/* The same K with a body force f: F[a][i] = N_a f_i.  Not a reference demo (demo/Elasticity3D.c:43-45 writes F = 0); it exercises
 * what the reference's System path does with ANY callback's F (src/petigapoint.c:427-450 AddVec, src/petigaelem.c:1377-1387
 * FixSystem: F -= K[:,k] v, F[k] = v) for a multi-field form.  ctx: {lambda, mu, f[3]} */
int orc_form_elasticity_f(OrcPoint *p,double *K,double *F,void *ctx)
{
  const double *c = (const double*)ctx;
  OrcElasticityCtx user; int a,i,nen=p->nen;
  const double *N0 = p->shape[0];
  user.lambda = c[0]; user.mu = c[1];
  orc_form_elasticity(p,K,F,&user);
  for (a=0; a<nen; a++) for (i=0; i<3; i++) F[a*3+i] = N0[a]*c[2+i];
  return 0;
}

/* demo/CahnHilliard3D.c:11-16 (Mobility), :39-53 (ChemicalPotential); dim-generic over the
 * diagonal second derivatives (the 2-D demo, demo/CahnHilliard2D.c, uses 3*alpha scaling instead:
 * selected by L0 <= 0) */
static void ch_mobility(double c,double *M,double *dM,double *d2M) { *M = c*(1-c); *dM = 1-2*c; *d2M = -2; }
static void ch_chempot(const OrcCahnHilliardCtx *u,double c,double *dmu,double *d2mu)
{
  double scale = (u->L0 > 0) ? u->L0*u->L0/u->lambda : 3*u->alpha;
  *dmu  = (0.5/u->theta*1.0/(c*(1-c)) - 2)*scale;
  *d2mu = (-0.5/u->theta*(1-2*c)/(c*c*(1-c)*(1-c)))*scale;
}

/* demo/CahnHilliard3D.c:55-109 (Residual) */
int orc_form_ch_residual(OrcPoint *p,double shift,const double *V,double t,const double *U,double *R,void *ctx)
{
  const OrcCahnHilliardCtx *user = (const OrcCahnHilliardCtx*)ctx;
  int a,i,nen=p->nen,dim=p->dim,d2=dim*dim;
  double c,c_t,M,dM,d2M,dmu,d2mu,c1[3],c2[9],lap=0,t1;
  const double *N0=p->shape[0],*N1=p->shape[1],*N2=p->shape[2];
  (void)shift; (void)t;
  orc_point_value(p,V,&c_t); orc_point_value(p,U,&c);
  ch_mobility(c,&M,&dM,&d2M); ch_chempot(user,c,&dmu,&d2mu);
  orc_point_grad(p,U,c1); orc_point_hess(p,U,c2);
  for (i=0;i<dim;i++) lap += c2[i*(dim+1)];
  t1 = M*dmu + dM*lap;
  for (a=0; a<nen; a++) {
    double Ra = 0, lapN = 0;
    for (i=0;i<dim;i++) lapN += N2[a*d2+i*(dim+1)];
    Ra += N0[a]*c_t;
    for (i=0;i<dim;i++) Ra += N1[a*dim+i]*t1*c1[i];
    Ra += lapN*M*lap;
    R[a] = Ra;
  }
  return 0;
}

/* demo/CahnHilliard3D.c:111-179 (Tangent) */
int orc_form_ch_tangent(OrcPoint *p,double shift,const double *V,double t,const double *U,double *K,void *ctx)
{
  const OrcCahnHilliardCtx *user = (const OrcCahnHilliardCtx*)ctx;
  int a,b,i,nen=p->nen,dim=p->dim,d2=dim*dim;
  double c,M,dM,d2M,dmu,d2mu,c1[3],c2[9],lap=0,t1;
  const double *N0=p->shape[0],*N1=p->shape[1],*N2=p->shape[2];
  (void)V; (void)t;
  orc_point_value(p,U,&c);
  ch_mobility(c,&M,&dM,&d2M); ch_chempot(user,c,&dmu,&d2mu);
  orc_point_grad(p,U,c1); orc_point_hess(p,U,c2);
  for (i=0;i<dim;i++) lap += c2[i*(dim+1)];
  t1 = M*dmu + dM*lap;
  for (a=0; a<nen; a++) {
    double lapNa = 0;
    for (i=0;i<dim;i++) lapNa += N2[a*d2+i*(dim+1)];
    for (b=0; b<nen; b++) {
      double Kab = 0, lapNb = 0, t2;
      for (i=0;i<dim;i++) lapNb += N2[b*d2+i*(dim+1)];
      Kab += shift*N0[a]*N0[b];
      for (i=0;i<dim;i++) Kab += N1[a*dim+i]*t1*N1[b*dim+i];
      t2 = (dM*dmu + M*d2mu + d2M*lap)*N0[b] + dM*lapNb;
      for (i=0;i<dim;i++) Kab += N1[a*dim+i]*t2*c1[i];
      Kab += lapNa*(dM*lap*N0[b] + M*lapNb);
      K[a*nen+b] = Kab;
    }
  }
  return 0;
}

/* demo/NavierStokesVMS.c:9-46 (Tau) */
static void ns_tau(const double J[9],double dt,const double u[],double nu,double *tauM,double *tauC)
{
  double C_I = 1.0/12.0, G[9], g[3]={0,0,0}, G_G=0, g_g=0, u_G_u=0;
  int i,j,k;
  memset(G,0,sizeof(G));
  for (i=0;i<3;i++) for (j=0;j<3;j++) for (k=0;k<3;k++) G[i*3+j] += J[i*3+k]*J[j*3+k];
  for (i=0;i<3;i++) for (j=0;j<3;j++) g[i] += J[i*3+j];
  for (i=0;i<3;i++) for (j=0;j<3;j++) G_G += G[i*3+j]*G[i*3+j];
  for (i=0;i<3;i++) g_g += g[i]*g[i];
  for (i=0;i<3;i++) for (j=0;j<3;j++) u_G_u += u[i]*G[i*3+j]*u[j];
  *tauM = 4/(dt*dt) + u_G_u + C_I*nu*nu*G_G;
  *tauM = 1/sqrt(*tauM);
  *tauC = (*tauM)*g_g;
  *tauC = 1/(*tauC);
}

/* demo/NavierStokesVMS.c:78-164 (Residual); dt comes from ctx (the reference reads it with
 * TSGetTimeStep inside the point function, :85) */
int orc_form_ns_residual(OrcPoint *pnt,double shift,const double *V,double t,const double *U,double *Re,void *ctx)
{
  const OrcNSVMSCtx *user = (const OrcNSVMSCtx*)ctx;
  double nu = user->nu, dt = user->dt;
  double u_t[4],u[4],grad_u[12],der2_u[36],G[9],tauM,tauC;
  double ux,uy,uz,pr,ux_t,uy_t,uz_t;
  double ux_x,ux_y,ux_z,uy_x,uy_y,uy_z,uz_x,uz_y,uz_z,p_x,p_y,p_z;
  double ux_s,uy_s,uz_s,p_s;
  const double *N0 = pnt->shape[0], *N1 = pnt->shape[1];
  int a,nen=pnt->nen;
  (void)shift; (void)t;
  orc_point_value(pnt,V,u_t); orc_point_value(pnt,U,u);
  orc_point_grad(pnt,U,grad_u); orc_point_hess(pnt,U,der2_u);
  ux=u[0]; uy=u[1]; uz=u[2]; pr=u[3]; ux_t=u_t[0]; uy_t=u_t[1]; uz_t=u_t[2];
  ux_x=grad_u[0]; ux_y=grad_u[1]; ux_z=grad_u[2];
  uy_x=grad_u[3]; uy_y=grad_u[4]; uy_z=grad_u[5];
  uz_x=grad_u[6]; uz_y=grad_u[7]; uz_z=grad_u[8];
  p_x=grad_u[9]; p_y=grad_u[10]; p_z=grad_u[11];
  orc_point_invgradgeommap(pnt,G);
  ns_tau(G,dt,u,nu,&tauM,&tauC);
  { /* FineScale, :48-75 */
    double ux_l = der2_u[0*9+0]+der2_u[0*9+4]+der2_u[0*9+8];
    double uy_l = der2_u[1*9+0]+der2_u[1*9+4]+der2_u[1*9+8];
    double uz_l = der2_u[2*9+0]+der2_u[2*9+4]+der2_u[2*9+8];
    ux_s = ux_t + (ux*ux_x + uy*ux_y + uz*ux_z) + p_x - nu*ux_l - user->fx;
    uy_s = uy_t + (ux*uy_x + uy*uy_y + uz*uy_z) + p_y - nu*uy_l - user->fy;
    uz_s = uz_t + (ux*uz_x + uy*uz_y + uz*uz_z) + p_z - nu*uz_l - user->fz;
    p_s  = ux_x + uy_y + uz_z;
    ux_s *= -tauM; uy_s *= -tauM; uz_s *= -tauM; p_s *= -tauC;
  }
  for (a=0; a<nen; a++) {
    double Na=N0[a],Na_x=N1[a*3+0],Na_y=N1[a*3+1],Na_z=N1[a*3+2];
    double Rux,Ruy,Ruz,Rp;
    Rux = -Na*user->fx; Ruy = -Na*user->fy; Ruz = -Na*user->fz; Rp = 0.0;
    Rux += Na*ux_t - Na_x*pr + nu*( Na_x*( ux_x + ux_x ) + Na_y*( ux_y + uy_x ) + Na_z*( ux_z + uz_x ) );
    Ruy += Na*uy_t - Na_y*pr + nu*( Na_x*( uy_x + ux_y ) + Na_y*( uy_y + uy_y ) + Na_z*( uy_z + uz_y ) );
    Ruz += Na*uz_t - Na_z*pr + nu*( Na_x*( uz_x + ux_z ) + Na_y*( uz_y + uy_z ) + Na_z*( uz_z + uz_z ) );
    Rp  += Na*( ux_x + uy_y + uz_z );
    Rux += - ( Na_x*p_s ); Ruy += - ( Na_y*p_s ); Ruz += - ( Na_z*p_s );
    Rp  += - ( Na_x*ux_s + Na_y*uy_s + Na_z*uz_s );
    Rux += + Na * ( (ux+ux_s)*ux_x + (uy+uy_s)*ux_y + (uz+uz_s)*ux_z );
    Ruy += + Na * ( (ux+ux_s)*uy_x + (uy+uy_s)*uy_y + (uz+uz_s)*uy_z );
    Ruz += + Na * ( (ux+ux_s)*uz_x + (uy+uy_s)*uz_y + (uz+uz_s)*uz_z );
    Rux += - ( Na_x*ux_s*(ux+ux_s) + Na_y*ux_s*(uy+uy_s) + Na_z*ux_s*(uz+uz_s) );
    Ruy += - ( Na_x*uy_s*(ux+ux_s) + Na_y*uy_s*(uy+uy_s) + Na_z*uy_s*(uz+uz_s) );
    Ruz += - ( Na_x*uz_s*(ux+ux_s) + Na_y*uz_s*(uy+uy_s) + Na_z*uz_s*(uz+uz_s) );
    Re[a*4+0] = Rux; Re[a*4+1] = Ruy; Re[a*4+2] = Ruz; Re[a*4+3] = Rp;
  }
  return 0;
}

/* demo/NavierStokesVMS.c:166-244 (Tangent); accumulates (+=) into the zeroed K like the reference */
int orc_form_ns_tangent(OrcPoint *pnt,double shift,const double *V,double t,const double *U,double *Ke,void *ctx)
{
  const OrcNSVMSCtx *user = (const OrcNSVMSCtx*)ctx;
  double nu = user->nu, dt = user->dt, u[4], G[9], tauM, tauC, ux,uy,uz;
  const double *N0 = pnt->shape[0], *N1 = pnt->shape[1];
  int a,b,i,j,nen=pnt->nen,N=nen*4;
  (void)V; (void)t;
  orc_point_value(pnt,U,u); ux=u[0]; uy=u[1]; uz=u[2];
  orc_point_invgradgeommap(pnt,G);
  ns_tau(G,dt,u,nu,&tauM,&tauC);
  for (a=0; a<nen; a++) {
    double Na=N0[a],Na_x=N1[a*3+0],Na_y=N1[a*3+1],Na_z=N1[a*3+2];
    for (b=0; b<nen; b++) {
      double Nb=N0[b],Nb_x=N1[b*3+0],Nb_y=N1[b*3+1],Nb_z=N1[b*3+2];
      double T[4][4];
      double Tii =
        (+ shift * Na * Nb
         + Na * (ux * Nb_x + uy * Nb_y + uz * Nb_z)
         + nu * (Na_x * Nb_x + Na_y * Nb_y + Na_z * Nb_z)
         + tauM * (ux * Na_x + uy * Na_y + uz * Na_z) *
                  (shift * Nb + (ux * Nb_x + uy * Nb_y + uz * Nb_z)));
      T[0][0] = nu * Na_x * Nb_x  +  tauC * Na_x * Nb_x;
      T[0][1] = nu * Na_y * Nb_x  +  tauC * Na_x * Nb_y;
      T[0][2] = nu * Na_z * Nb_x  +  tauC * Na_x * Nb_z;
      T[1][0] = nu * Na_x * Nb_y  +  tauC * Na_y * Nb_x;
      T[1][1] = nu * Na_y * Nb_y  +  tauC * Na_y * Nb_y;
      T[1][2] = nu * Na_z * Nb_y  +  tauC * Na_y * Nb_z;
      T[2][0] = nu * Na_x * Nb_z  +  tauC * Na_z * Nb_x;
      T[2][1] = nu * Na_y * Nb_z  +  tauC * Na_z * Nb_y;
      T[2][2] = nu * Na_z * Nb_z  +  tauC * Na_z * Nb_z;
      T[0][0] += Tii; T[1][1] += Tii; T[2][2] += Tii;
      T[0][3] = - Na_x * Nb  +  tauM * (ux * Na_x + uy * Na_y + uz * Na_z) * Nb_x;
      T[1][3] = - Na_y * Nb  +  tauM * (ux * Na_x + uy * Na_y + uz * Na_z) * Nb_y;
      T[2][3] = - Na_z * Nb  +  tauM * (ux * Na_x + uy * Na_y + uz * Na_z) * Nb_z;
      T[3][0] = + Na * Nb_x  +  tauM * Na_x * (shift * Nb + (ux * Nb_x + uy * Nb_y + uz * Nb_z));
      T[3][1] = + Na * Nb_y  +  tauM * Na_y * (shift * Nb + (ux * Nb_x + uy * Nb_y + uz * Nb_z));
      T[3][2] = + Na * Nb_z  +  tauM * Na_z * (shift * Nb + (ux * Nb_x + uy * Nb_y + uz * Nb_z));
      T[3][3] = + tauM * (Na_x * Nb_x + Na_y * Nb_y + Na_z * Nb_z);
      for (i=0;i<4;i++) for (j=0;j<4;j++) Ke[(a*4+i)*N + b*4+j] += T[i][j];
    }
  }
  return 0;
}

/* demo/BratuFJ.F90:23-58 (Bratu_Function, Galerkin branch): F_a = grad N_a . grad u - N_a lambda exp(u); ctx = double* lambda */
int orc_form_bratu_function(OrcPoint *p,const double *U,double *F,void *ctx)
{
  int a,i,nen=p->nen,dim=p->dim; double u,gu[3],lambda = *(const double*)ctx;
  const double *N0=p->shape[0],*N1=p->shape[1];
  orc_point_value(p,U,&u); orc_point_grad(p,U,gu);
  for (a=0; a<nen; a++) {
    double s = 0;
    for (i=0;i<dim;i++) s += N1[a*dim+i]*gu[i];
    F[a] = + s - N0[a]*lambda*exp(u);
  }
  return 0;
}

/* demo/BratuFJ.F90:60-107 (Bratu_Jacobian, Galerkin branch): J(b,a) = grad N_a . grad N_b - N_a N_b lambda exp(u) */
int orc_form_bratu_jacobian(OrcPoint *p,const double *U,double *J,void *ctx)
{
  int a,b,i,nen=p->nen,dim=p->dim; double u,lambda = *(const double*)ctx;
  const double *N0=p->shape[0],*N1=p->shape[1];
  orc_point_value(p,U,&u);
  for (a=0; a<nen; a++) for (b=0; b<nen; b++) {
    double s = 0;
    for (i=0;i<dim;i++) s += N1[a*dim+i]*N1[b*dim+i];
    J[a*nen+b] = + s - N0[a]*N0[b]*lambda*exp(u);
  }
  return 0;
}

/* demo/BratuFJ.F90:111-141 (Bratu_IFunction): F_a = N_a v + grad N_a . grad u - N_a lambda exp(u) */
int orc_form_bratu_ifunction(OrcPoint *p,double shift,const double *V,double t,const double *U,double *F,void *ctx)
{
  int a,i,nen=p->nen,dim=p->dim; double v,u,gu[3],lambda = *(const double*)ctx;
  const double *N0=p->shape[0],*N1=p->shape[1];
  (void)shift; (void)t;
  orc_point_value(p,V,&v); orc_point_value(p,U,&u); orc_point_grad(p,U,gu);
  for (a=0; a<nen; a++) {
    double s = 0;
    for (i=0;i<dim;i++) s += N1[a*dim+i]*gu[i];
    F[a] = + N0[a]*v + s - N0[a]*lambda*exp(u);
  }
  return 0;
}

/* demo/BratuFJ.F90:143-176 (Bratu_IJacobian): J(b,a) = shift N_a N_b + grad N_a . grad N_b - N_a N_b lambda exp(u) */
int orc_form_bratu_ijacobian(OrcPoint *p,double shift,const double *V,double t,const double *U,double *J,void *ctx)
{
  int a,b,i,nen=p->nen,dim=p->dim; double u,lambda = *(const double*)ctx;
  const double *N0=p->shape[0],*N1=p->shape[1];
  (void)V; (void)t;
  orc_point_value(p,U,&u);
  for (a=0; a<nen; a++) for (b=0; b<nen; b++) {
    double s = 0;
    for (i=0;i<dim;i++) s += N1[a*dim+i]*N1[b*dim+i];
    J[a*nen+b] = + shift*N0[a]*N0[b] + s - N0[a]*N0[b]*lambda*exp(u);
  }
  return 0;
}

/* demo/AdvectionDiffusion.c:26-47 (System): K = grad Na . grad Nb + Na (w . grad Nb), F = 0; ctx = double wind[3] */
int orc_form_advection_diffusion(OrcPoint *p,double *K,double *F,void *ctx)
{
  const double *w = (const double*)ctx;
  int a,b,i,nen=p->nen,dim=p->dim; const double *N0=p->shape[0],*N1=p->shape[1];
  for (a=0; a<nen; a++) {
    for (b=0; b<nen; b++) {
      double diffusion = 0, advection = 0;
      for (i=0;i<dim;i++) { diffusion += N1[a*dim+i]*N1[b*dim+i]; advection += w[i]*N1[b*dim+i]; }
      K[a*nen+b] = diffusion + N0[a]*advection;
    }
    F[a] = 0.0;
  }
  return 0;
}
