/*
 * igaoracle.c -- CPU restatement of PetIGA's element-assembly path (see igaoracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into the product library.
 *
 * Loop structure, index conventions and arithmetic follow the reference
 * (dalcinl/PetIGA @ 2025-04-04); citations are file:line of that tree.
 * Array conventions: every multi-index array is laid out exactly as the
 * reference lays it out in memory (C row-major == the Fortran kernels'
 * column-major view), so `basis[k]` is [nqp][nen][dim^k], `mapX[1]` is
 * [nqp][nsd][dim], `mapU[1]` is [nqp][dim][nsd], K_e is [nen][dof][nen][dof].
 */
#include "igaoracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <limits.h>

#define ORC_ERR(msg) do { fprintf(stderr,"igaoracle: %s (%s:%d)\n",msg,__FILE__,__LINE__); return 1; } while (0)

static void *xcalloc(size_t n,size_t sz) { void *p = calloc(n?n:1,sz); if(!p){perror("calloc");abort();} return p; }

static int ipow(int b,int e) { int r=1; while (e-- > 0) r*=b; return r; }

/* ------------------------------------------------------------------ */
/* Quadrature rules.  src/petigarule.c:182-319 tabulates Gauss-Legendre */
/* nodes/weights for q=1..10 and :321-459 Gauss-Lobatto for q=2..10 as  */
/* 36-digit constants, ascending nodes, exactly symmetric               */
/* (X[q-1-i] = -X[i]).  The constants are not copied: the same numbers  */
/* are computed (Newton in long double, two more steps in __float128 so */
/* that the cast to double rounds correctly) and held, bit for bit, to  */
/* the doubles of the reference's table kept in                         */
/* tests/golden/gauss_rules.json (tests/test_golden.py).                */
/* ------------------------------------------------------------------ */
typedef __float128 orcq;

/* P_n(x) and P_{n-1}(x) by the three-term recurrence */
static void orc_legendre_pair(int n,orcq x,orcq *Pn,orcq *Pm)
{
  orcq p0 = 1, p1 = x; int k;
  if (n == 0) { *Pn = 1; *Pm = 0; return; }
  for (k=2; k<=n; k++) { orcq p2 = ((2*k-1)*x*p1 - (k-1)*p0)/k; p0=p1; p1=p2; }
  *Pn = p1; *Pm = p0;
}

int orc_gauss_legendre(int q,double *X,double *W)
{
  int i,it;
  if (q < 1 || q > 10) return 1;   /* the reference implements 1..10 only */
  for (i=0; i<(q+1)/2; i++) {
    long double xl = cosl(3.14159265358979323846264338327950288L*(i+0.75L)/(q+0.5L));
    orcq x,p,pm,dp,w;
    for (it=0; it<100; it++) {                 /* long double Newton */
      long double dx;
      orc_legendre_pair(q,(orcq)xl,&p,&pm);
      dp = q*((orcq)xl*p - pm)/((orcq)xl*xl-1);
      dx = (long double)(p/dp); xl -= dx;
      if (fabsl(dx) < 1e-19L) break;
    }
    x = xl;
    for (it=0; it<2; it++) {                   /* quadruple-precision polish */
      orc_legendre_pair(q,x,&p,&pm);
      dp = q*(x*p - pm)/(x*x-1);
      x -= p/dp;
    }
    orc_legendre_pair(q,x,&p,&pm);
    dp = q*(x*p - pm)/(x*x-1);
    w = 2/((1-x*x)*dp*dp);
    /* x is the i-th largest root; store ascending and mirrored */
    X[q-1-i] = (double)x;  W[q-1-i] = (double)w;
    X[i]     = -(double)x; W[i]     = (double)w;
  }
  if (q % 2) X[q/2] = 0.0;
  return 0;
}

/* Gauss-Lobatto (src/petigarule.c:321-459 tabulates q=2..10): x = +-1 and the roots of P'_{q-1};
   w = 2/(q(q-1) P_{q-1}(x)^2).  With n = q-1: (1-x^2) P_n' = n (P_{n-1} - x P_n) and
   (1-x^2) P_n'' = 2x P_n' - n(n+1) P_n. */
int orc_gauss_lobatto(int q,double *X,double *W)
{
  int i,it,n=q-1;
  if (q < 2 || q > 10) return 1;
  for (i=0; i<(q+1)/2; i++) {
    orcq x = 1,p,pm,w,d1,d2;
    if (i > 0) {
      long double xl = cosl(3.14159265358979323846264338327950288L*i/n);
      for (it=0; it<100; it++) {               /* long double Newton on P_n' */
        long double dx;
        x = xl;
        orc_legendre_pair(n,x,&p,&pm);
        d1 = n*(pm - x*p)/(1-x*x);
        d2 = (2*x*d1 - n*(n+1)*p)/(1-x*x);
        dx = (long double)(d1/d2); xl -= dx;
        if (fabsl(dx) < 1e-19L) break;
      }
      x = xl;
      for (it=0; it<2; it++) {                 /* quadruple-precision polish */
        orc_legendre_pair(n,x,&p,&pm);
        d1 = n*(pm - x*p)/(1-x*x);
        d2 = (2*x*d1 - n*(n+1)*p)/(1-x*x);
        x -= d1/d2;
      }
    }
    orc_legendre_pair(n,x,&p,&pm);
    w = 2/((orcq)(q*n)*p*p);
    X[q-1-i] = (double)x;  W[q-1-i] = (double)w;
    X[i]     = -(double)x; W[i]     = (double)w;
  }
  if (q % 2) X[q/2] = 0.0;
  return 0;
}

/* ------------------------------------------------------------------ */
/* Knot vectors                                                        */
/* ------------------------------------------------------------------ */

/* src/petigaaxis.c:482-494 (IGA_NextKnot) */
static int next_knot(int m,const double *U,int k,int direction)
{
  int j;
  if (direction >= 0) {
    if (k < 0) return 0;
    for (j=k+1; j<m; j++) if (U[j] > U[k]) return j;
    return m;
  } else {
    if (k > m) return m;
    for (j=k-1; j>0; j--) if (U[j] < U[k]) return j;
    return 0;
  }
}

static void axis_free(OrcAxis *ax) { free(ax->U); free(ax->span); ax->U=NULL; ax->span=NULL; ax->m=0; ax->nel=0; ax->nnp=0; }

/* src/petigaaxis.c:286-312 (IGAAxisGetSpans) + :513-523 (nnp in IGAAxisSetUp) */
static void axis_spans(OrcAxis *ax)
{
  int p=ax->p, m=ax->m, n=m-p-1, k, count;
  free(ax->span);
  k = p; count = 0;
  while ((k = next_knot(m,ax->U,k,1)) <= n+1) count++;
  ax->span = (int*)xcalloc((size_t)count,sizeof(int));
  k = p; count = 0;
  while ((k = next_knot(m,ax->U,k,1)) <= n+1) ax->span[count++] = k-1;
  ax->nel = count;
  if (ax->periodic) {
    int kk = n+1, j = next_knot(m,ax->U,kk,1), s = j-kk, C = p-s;
    ax->nnp = n-C;
  } else ax->nnp = n+1;
}

int orc_axis_set_degree(OrcIGA *iga,int i,int p)
{
  if (p < 1) ORC_ERR("degree must be >= 1");
  iga->axis[i].p = p; iga->setup = 0;
  return 0;
}
int orc_axis_set_periodic(OrcIGA *iga,int i,int flag) { iga->axis[i].periodic = flag?1:0; iga->setup=0; return 0; }

/* src/petigaaxis.c:401-456 (IGAAxisInitUniform) */
int orc_axis_init_uniform(OrcIGA *iga,int i,int N,double Ui,double Uf,int C)
{
  OrcAxis *ax = &iga->axis[i];
  int p=ax->p,s,n,m,r,j,k,b;
  if (p < 1) ORC_ERR("set degree first");
  if (C < 0) C = p-1;                       /* PETSC_DECIDE */
  if (N < 1 || Ui >= Uf || C >= p) ORC_ERR("bad uniform axis arguments");
  s = p - C; r = N; m = 2*(p+1) + (N-1)*s - 1; n = m - p - 1;
  free(ax->U); ax->U = (double*)xcalloc((size_t)m+1,sizeof(double)); ax->m = m;
  for (k=0; k<=p; k++) { ax->U[k] = Ui; ax->U[m-k] = Uf; }
  for (b=1; b<=r-1; b++)
    for (j=1; j<=s; j++)
      ax->U[k++] = Ui + (double)b/(double)N * (Uf-Ui);
  if (ax->periodic)
    for (k=0; k<=C; k++) {
      ax->U[C-k]   = ax->U[p]   - ax->U[m-p] + ax->U[n-k];
      ax->U[m-C+k] = ax->U[m-p] - ax->U[p]   + ax->U[p+1+k];
    }
  free(ax->span); ax->nel = r;
  ax->span = (int*)xcalloc((size_t)r,sizeof(int));
  for (j=0; j<r; j++) ax->span[j] = p + j*s;
  ax->nnp = ax->periodic ? n-C : n+1;
  iga->setup = 0;
  return 0;
}

/* src/petigaaxis.c:202-253 (IGAAxisSetKnots) */
int orc_axis_set_knots(OrcIGA *iga,int i,int m,const double *U)
{
  OrcAxis *ax = &iga->axis[i];
  int p = ax->p,k,j;
  if (p < 1) ORC_ERR("set degree first");
  if (m < 2*p+1) ORC_ERR("too few knots");
  for (k=1; k<=m; k++) if (U[k-1] > U[k]) ORC_ERR("knots must be non-decreasing");
  for (k=1,j=m; k<m; k=j) { j = next_knot(m,U,k,1); if (j-k > p) ORC_ERR("knot multiplicity > degree"); }
  free(ax->U); ax->U = (double*)xcalloc((size_t)m+1,sizeof(double)); ax->m = m;
  memcpy(ax->U,U,((size_t)m+1)*sizeof(double));
  free(ax->span); ax->span = NULL;
  axis_spans(ax);
  iga->setup = 0;
  return 0;
}

/* ------------------------------------------------------------------ */
/* 1-D B-spline basis and derivatives: Piegl & Tiller, The NURBS Book, */
/* algorithm A2.3, as in src/petigabsb.f90.in:3-63; output layout      */
/* B[a][0..4] as src/petigabsp.F90:3-16 (unused derivative slots = 0). */
/* ------------------------------------------------------------------ */
void orc_bspline_ders(int i,double uu,int p,int n,const double *U,double *B)
{
  double ndu[10][10], a[2][10], left[10], right[10], ders[10][5];   /* p <= 9 */
  int j,k,r;
  /* p <= 9 (the reference allows any p; the engine takes p <= 7) */
  ndu[0][0] = 1;
  for (j=1; j<=p; j++) {
    double saved = 0;
    left[j]  = uu - U[i+1-j];
    right[j] = U[i+j] - uu;
    for (r=0; r<j; r++) {
      double temp;
      ndu[j][r] = right[r+1] + left[j-r];
      temp = ndu[r][j-1] / ndu[j][r];
      ndu[r][j] = saved + right[r+1]*temp;
      saved = left[j-r]*temp;
    }
    ndu[j][j] = saved;
  }
  for (j=0; j<=p; j++) ders[j][0] = ndu[j][p];
  for (r=0; r<=p; r++) {
    int s1 = 0, s2 = 1;
    a[0][0] = 1;
    for (k=1; k<=n; k++) {
      double d = 0;
      int rk = r-k, pk = p-k, j1, j2, t;
      if (r >= k) { a[s2][0] = a[s1][0]/ndu[pk+1][rk]; d = a[s2][0]*ndu[rk][pk]; }
      j1 = (rk > -1) ? 1 : -rk;
      j2 = (r-1 <= pk) ? k-1 : p-r;
      for (j=j1; j<=j2; j++) {
        a[s2][j] = (a[s1][j] - a[s1][j-1])/ndu[pk+1][rk+j];
        d += a[s2][j]*ndu[rk+j][pk];
      }
      if (r <= pk) { a[s2][k] = -a[s1][k-1]/ndu[pk+1][r]; d += a[s2][k]*ndu[r][pk]; }
      ders[r][k] = d;
      t = s1; s1 = s2; s2 = t;
    }
  }
  r = p;
  for (k=1; k<=n; k++) {
    for (j=0; j<=p; j++) ders[j][k] *= (double)r;
    r *= (p-k);
  }
  for (j=0; j<=p; j++) {
    for (k=0; k<5; k++) B[j*5+k] = 0;
    for (k=0; k<=n; k++) B[j*5+k] = ders[j][k];
  }
}

static void basis_free(OrcBasis *b)
{
  free(b->offset); free(b->detJac); free(b->weight); free(b->point); free(b->value);
  free(b->bnd_value[0]); free(b->bnd_value[1]);
  memset(b,0,sizeof(*b));
}

/* src/petigabasis.c:83-219 (IGABasisInitQuadrature) with the rule of IGARuleSetUp (src/petigarule.c:116-143):
   type 0 Gauss-Legendre, 1 Gauss-Lobatto, 3 user-defined (ux, uw); the reduced rule (type 2, :144-171) is not restated */
static int basis_init_quadrature(OrcBasis *bs,const OrcAxis *ax,int nqp,int type,const double *ux,const double *uw)
{
  int p=ax->p, m=ax->m, n=m-p-1, nel=ax->nel, nen=p+1, d=(p<4)?p:4, e,q;
  double X[64],W[64],Xr[64],Wr[64]; int nr = nqp;
  const double *U = ax->U;
  if (nqp > 64) ORC_ERR("rule size not implemented");
  if (type == 0)      { if (orc_gauss_legendre(nqp,X,W)) ORC_ERR("rule size not implemented"); }
  else if (type == 1) { if (orc_gauss_lobatto(nqp,X,W))  ORC_ERR("rule size not implemented"); }
  else if (type == 3) { if (!ux || !uw) ORC_ERR("user rule not set"); for (q=0; q<nqp; q++) { X[q]=ux[q]; W[q]=uw[q]; } }
  else {   /* IGA_RULE_REDUCED, src/petigabasis.c:144-171: Gauss-Legendre with nqp points on the first and the last element, with
            * nqp-1 on the others (the last slot: weight 0, point PETSC_MAX_REAL, no basis values; IGA_Quadrature_SIZE trims it) */
    if (orc_gauss_legendre(nqp,X,W)) ORC_ERR("rule size not implemented");
    if (nel > 2 && nqp > 1) { nr = nqp-1; if (orc_gauss_legendre(nr,Xr,Wr)) ORC_ERR("rule size not implemented"); }
  }
  basis_free(bs);
  bs->nel=nel; bs->nqp=nqp; bs->nen=nen;
  bs->offset = (int*)   xcalloc((size_t)nel,sizeof(int));
  bs->detJac = (double*)xcalloc((size_t)nel,sizeof(double));
  bs->weight = (double*)xcalloc((size_t)nel*nqp,sizeof(double));
  bs->point  = (double*)xcalloc((size_t)nel*nqp,sizeof(double));
  bs->value  = (double*)xcalloc((size_t)nel*nqp*nen*5,sizeof(double));
  for (e=0; e<nel; e++) {
    int k = ax->span[e];
    double u0 = U[k], u1 = U[k+1], J = (u1-u0)/2;
    bs->detJac[e] = J;
    bs->offset[e] = k - p;
    if (type == 2 && nr < nqp && e > 0 && e < nel-1) {
      for (q=0; q<nr; q++) { bs->weight[e*nqp+q] = Wr[q]; bs->point[e*nqp+q] = (Xr[q] + 1)*J + u0; }
      for (q=nr; q<nqp; q++) { bs->weight[e*nqp+q] = 0; bs->point[e*nqp+q] = 1.797693134862315708e308; }
    } else
    for (q=0; q<nqp; q++) {
      bs->weight[e*nqp+q] = W[q];
      bs->point [e*nqp+q] = (X[q] + 1)*J + u0;
    }
    for (q=0; q<nqp && bs->weight[e*nqp+q] > 0; q++)
      orc_bspline_ders(k,bs->point[e*nqp+q],p,d,U,&bs->value[((size_t)e*nqp+q)*nen*5]);
  }
  {
    int k0 = p, k1 = n;
    double u0 = U[k0], u1 = U[k1+1];
    bs->bnd_value[0] = (double*)xcalloc((size_t)nen*5,sizeof(double));
    bs->bnd_value[1] = (double*)xcalloc((size_t)nen*5,sizeof(double));
    bs->bnd_point[0] = u0; bs->bnd_point[1] = u1; bs->bnd_detJac = 1.0; bs->bnd_weight = 1.0;
    orc_bspline_ders(k0,u0,p,d,U,bs->bnd_value[0]);
    orc_bspline_ders(k1,u1,p,d,U,bs->bnd_value[1]);
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Partitioning: src/petigapart.c:11-202.  Integer-only; the search    */
/* (cut-minimising processor grid, tie-breaks included) is reproduced  */
/* decision for decision so grids are bit-exact.                       */
/* ------------------------------------------------------------------ */
static int cut2(int M,int N,int m,int n) { return M*(n-1) + N*(m-1); }
static int cut3(int M,int N,int P,int m,int n,int p) { return N*P*(m-1) + M*P*(n-1) + M*N*(p-1); }

static int largest_divisor_le(int size,int m) { if (m == 0) m = 1; while (m > 0 && size % m) m--; return m; }

static int part2_try(int size,int M,int N,int *m,int *n)
{
  int mm = (int)(0.5 + sqrt(((double)M)/((double)N)*((double)size)));
  mm = largest_divisor_le(size,mm);
  *m = mm; *n = size/mm;
  return cut2(M,N,*m,*n);
}
static void part2(int size,int M,int N,int *m,int *n)
{
  int m1,n1,m2,n2,a,b;
  a = part2_try(size,M,N,&m1,&n1);
  b = part2_try(size,N,M,&n2,&m2);
  if (a < b) { *m = m1; *n = n1; } else { *m = m2; *n = n2; }
  if (M == N && *n < *m) { int t = *m; *m = *n; *n = t; }
}
static int part3_try(int size,int M,int N,int P,int *_m,int *_n,int *_p)
{
  int m,n,p,C,mm,nn,pp,CC;
  m = (int)(0.5 + pow(((double)M*(double)M)/((double)N*(double)P)*(double)size,1./3.));
  m = largest_divisor_le(size,m);
  part2(size/m,N,P,&n,&p);
  C = cut3(M,N,P,m,n,p);
  for (mm=m; mm>=1; mm--) { if (size % mm) continue; part2(size/mm,N,P,&nn,&pp); CC = cut3(M,N,P,mm,nn,pp); if (CC < C) { m=mm; n=nn; p=pp; C=CC; } }
  for (nn=n; nn>=1; nn--) { if (size % nn) continue; part2(size/nn,M,P,&mm,&pp); CC = cut3(M,N,P,mm,nn,pp); if (CC < C) { m=mm; n=nn; p=pp; C=CC; } }
  for (pp=p; pp>=1; pp--) { if (size % pp) continue; part2(size/pp,M,N,&mm,&nn); CC = cut3(M,N,P,mm,nn,pp); if (CC < C) { m=mm; n=nn; p=pp; C=CC; } }
  *_m=m; *_n=n; *_p=p;
  return cut3(M,N,P,m,n,p);
}
static void part3(int size,int M,int N,int P,int *_m,int *_n,int *_p)
{
  int m[3],n[3],p[3],C[3],k,i=0,Cmin=INT_MAX,t;
  C[0] = part3_try(size,M,N,P,&m[0],&n[0],&p[0]);
  C[1] = part3_try(size,N,M,P,&n[1],&m[1],&p[1]);
  C[2] = part3_try(size,P,M,N,&p[2],&m[2],&n[2]);
  for (k=0; k<3; k++) if (C[k] < Cmin) { Cmin = C[k]; i = k; }
  if (M == N && n[i] < m[i]) { t=m[i]; m[i]=n[i]; n[i]=t; }
  if (M == P && p[i] < m[i]) { t=m[i]; m[i]=p[i]; p[i]=t; }
  if (N == P && p[i] < n[i]) { t=n[i]; n[i]=p[i]; p[i]=t; }
  *_m=m[i]; *_n=n[i]; *_p=p[i];
}

/* src/petigapart.c:136-168 (IGA_Partition); n[] entries < 1 mean "decide" */
int orc_partition(int size,int rank,int dim,const int N[],int n[],int i[])
{
  int k,prod=1;
  if (size < 1) return 1;
  if (i && (rank < 0 || rank >= size)) return 1;
  if (dim == 3) {
    int m=n[0],nn=n[1],p=n[2];
    if (m<1 && nn<1 && p<1) part3(size,N[0],N[1],N[2],&m,&nn,&p);
    else if (m<1 && nn<1) part2(size/p,N[0],N[1],&m,&nn);
    else if (m<1 && p<1)  part2(size/nn,N[0],N[2],&m,&p);
    else if (nn<1 && p<1) part2(size/m,N[1],N[2],&nn,&p);
    else if (m<1)  m  = size/(nn*p);
    else if (nn<1) nn = size/(m*p);
    else if (p<1)  p  = size/(m*nn);
    n[0]=m; n[1]=nn; n[2]=p;
  } else if (dim == 2) {
    int m=n[0],nn=n[1];
    if (m<1 && nn<1) part2(size,N[0],N[1],&m,&nn);
    else if (m<1) m = size/nn;
    else if (nn<1) nn = size/m;
    n[0]=m; n[1]=nn;
  } else if (dim == 1) {
    if (n[0] < 1) n[0] = size;
  } else return 1;
  for (k=0; k<dim; k++) prod *= n[k];
  if (prod != size) return 1;
  for (k=0; k<dim; k++) if (N[k] < n[k]) return 1;
  if (i) for (k=0; k<dim; k++) { i[k] = rank % n[k]; rank -= i[k]; rank /= n[k]; }
  return 0;
}

/* src/petigapart.c:170-202 (IGA_Dist1D / IGA_Distribute) */
void orc_distribute(int dim,const int size[],const int rank[],const int N[],int n[],int s[])
{
  int k;
  for (k=0; k<dim; k++) {
    int q = N[k]/size[k], r = N[k]%size[k];
    n[k] = q + (r > rank[k]);
    s[k] = rank[k]*q + ((r > rank[k]) ? rank[k] : r);
  }
}

/* ------------------------------------------------------------------ */
/* IGA object                                                          */
/* ------------------------------------------------------------------ */
OrcIGA *orc_create(int dim,int dof)
{
  OrcIGA *iga = (OrcIGA*)xcalloc(1,sizeof(OrcIGA));
  int i;
  iga->dim = dim; iga->dof = dof; iga->order = -1;
  for (i=0; i<3; i++) { iga->proc_sizes[i] = 1; iga->proc_ranks[i] = 0; iga->rule_nqp[i] = -1; }
  return iga;
}

void orc_destroy(OrcIGA *iga)
{
  int i;
  if (!iga) return;
  for (i=0; i<3; i++) { axis_free(&iga->axis[i]); basis_free(&iga->basis[i]); }
  for (i=0; i<3; i++) { free(iga->rule_x[i]); free(iga->rule_w[i]); }
  free(iga->geometryX); free(iga->rationalW); free(iga->fixtableU); free(iga->propertyA);
  free(iga);
}

int orc_set_quadrature(OrcIGA *iga,int i,int q) { if (q < 1) ORC_ERR("nqp must be positive"); iga->rule_nqp[i] = q; iga->setup = 0; return 0; }
/* src/petiga.c:500-513 (IGASetRuleType) -> src/petigarule.c:89-98 */
int orc_set_rule_type(OrcIGA *iga,int i,int type) { if (type < 0 || type > 2) ORC_ERR("rule type"); iga->rule_type[i] = type; iga->setup = 0; return 0; }
/* src/petigarule.c:145-158 (IGARuleSetRule): a user-defined rule on [-1,1] */
int orc_set_rule(OrcIGA *iga,int i,int q,const double *x,const double *w)
{
  if (q < 1) ORC_ERR("nqp must be positive");
  free(iga->rule_x[i]); free(iga->rule_w[i]);
  iga->rule_x[i] = (double*)xcalloc((size_t)q,sizeof(double)); iga->rule_w[i] = (double*)xcalloc((size_t)q,sizeof(double));
  memcpy(iga->rule_x[i],x,(size_t)q*sizeof(double)); memcpy(iga->rule_w[i],w,(size_t)q*sizeof(double));
  iga->rule_type[i] = 3; iga->rule_user_n[i] = q; iga->rule_nqp[i] = q; iga->setup = 0; return 0;
}
/* src/petiga.c:463-472 (IGASetOrder): clipped to [1,4] */
int orc_set_order(OrcIGA *iga,int order) { if (order < 0) ORC_ERR("order must be >= 0"); iga->order = order<1?1:(order>4?4:order); return 0; }

int orc_set_partition(OrcIGA *iga,int size,int rank)
{
  int N[3]={1,1,1},n[3]={-1,-1,-1},r[3]={0,0,0},i;
  for (i=0; i<iga->dim; i++) { if (!iga->axis[i].span) ORC_ERR("axes first"); N[i] = iga->axis[i].nel; }
  if (orc_partition(size,rank,iga->dim,N,n,r)) ORC_ERR("bad partition");
  for (i=0; i<3; i++) { iga->proc_sizes[i] = (i<iga->dim)?n[i]:1; iga->proc_ranks[i] = (i<iga->dim)?r[i]:0; }
  iga->setup = 0;
  return 0;
}

/* src/petiga.c:1111-1310 (IGASetUp_Stage1, Galerkin branch) + :1450-1493 (IGASetUp stage 3) */
int orc_setup(OrcIGA *iga)
{
  int i,dim = iga->dim;
  if (dim < 1 || dim > 3) ORC_ERR("dim must be 1..3");
  for (i=0; i<dim; i++) { if (!iga->axis[i].U) ORC_ERR("axis not initialised"); if (!iga->axis[i].span) axis_spans(&iga->axis[i]); }
  for (i=0; i<dim; i++) iga->elem_sizes[i] = iga->axis[i].nel;
  orc_distribute(dim,iga->proc_sizes,iga->proc_ranks,iga->elem_sizes,iga->elem_width,iga->elem_start);
  for (i=0; i<dim; i++) {
    const OrcAxis *ax = &iga->axis[i];
    int nel = iga->elem_sizes[i], efirst = iga->elem_start[i], elast = efirst + iga->elem_width[i] - 1, p = ax->p;
    int gstart = ax->span[efirst] - p, gend = ax->span[elast] + 1, lstart = ax->span[efirst] - p, lend;
    lend = (elast < nel-1) ? ax->span[elast+1] - p : ax->span[elast] + 1;
    iga->node_sizes[i]  = ax->nnp;
    iga->node_lstart[i] = lstart; iga->node_lwidth[i] = lend - lstart;
    iga->node_gstart[i] = gstart; iga->node_gwidth[i] = gend - gstart;
    if (iga->proc_ranks[i] == iga->proc_sizes[i]-1) iga->node_lwidth[i] = iga->node_sizes[i] - iga->node_lstart[i];
  }
  for (i=dim; i<3; i++) {
    iga->elem_sizes[i]=1; iga->elem_start[i]=0; iga->elem_width[i]=1;
    iga->node_sizes[i]=1; iga->node_lstart[i]=0; iga->node_lwidth[i]=1; iga->node_gstart[i]=0; iga->node_gwidth[i]=1;
  }
  /* Stage 1 drops geometry / fix table (src/petiga.c:1290-1298) */
  iga->nsd = 0; iga->rational = 0;
  free(iga->geometryX); iga->geometryX = NULL;
  free(iga->rationalW); iga->rationalW = NULL;
  free(iga->propertyA); iga->propertyA = NULL; iga->property = 0;   /* src/petiga.c:1296-1299 */
  free(iga->fixtableU); iga->fixtableU = NULL; iga->fixtable = 0;
  if (iga->order < 0) { int o = 0; for (i=0; i<dim; i++) if (iga->axis[i].p > o) o = iga->axis[i].p; orc_set_order(iga,o); }
  for (i=0; i<dim; i++) {
    int q = iga->rule_nqp[i] > 0 ? iga->rule_nqp[i] : iga->axis[i].p + 1;   /* src/petigabasis.c:103 */
    if (iga->rule_type[i] == 3) q = iga->rule_user_n[i];
    if (basis_init_quadrature(&iga->basis[i],&iga->axis[i],q,iga->rule_type[i],iga->rule_x[i],iga->rule_w[i])) return 1;
  }
  for (i=dim; i<3; i++) {   /* unused axes: one "element", one point, N=1 (what IGAAxisReset+rule reset give) */
    OrcBasis *b = &iga->basis[i];
    basis_free(b);
    b->nel=1; b->nqp=1; b->nen=1;
    b->offset=(int*)xcalloc(1,sizeof(int)); b->detJac=(double*)xcalloc(1,sizeof(double));
    b->weight=(double*)xcalloc(1,sizeof(double)); b->point=(double*)xcalloc(1,sizeof(double));
    b->value=(double*)xcalloc(5,sizeof(double));
    b->detJac[0]=1; b->weight[0]=1; b->point[0]=0; b->value[0]=1;
  }
  iga->setup = 1;
  return 0;
}

int64_t orc_global_size(const OrcIGA *iga)
{ return (int64_t)iga->node_sizes[0]*iga->node_sizes[1]*iga->node_sizes[2]*iga->dof; }

/* Control net in natural order over the *geometry* grid (n+1 points per axis, not wrapped),
 * i0 fastest; copies the ghosted-local box like IGASetGeometryDim+IGALoadGeometry would
 * (src/petigaio.c:268-275 stores (x*w..,w); here X is already Cartesian and W separate). */
int orc_set_geometry(OrcIGA *iga,int nsd,const double *X,const double *W)
{
  int gs[3]={1,1,1},i,j,k,c,dim=iga->dim;
  const int *g0 = iga->node_gstart,*gw = iga->node_gwidth;
  size_t pos = 0;
  if (!iga->setup) ORC_ERR("setup first");
  if (nsd < dim || nsd > 3) ORC_ERR("Number of space dimensions must be in range [dim,3]");   /* IGASetGeometryDim, src/petigaio.c:187 */
  for (i=0; i<dim; i++) gs[i] = iga->axis[i].span[iga->axis[i].nel-1] + 1;
  free(iga->geometryX); free(iga->rationalW);
  iga->geometryX = (double*)xcalloc((size_t)gw[0]*gw[1]*gw[2]*nsd,sizeof(double));
  iga->rationalW = W ? (double*)xcalloc((size_t)gw[0]*gw[1]*gw[2],sizeof(double)) : NULL;
  for (k=g0[2]; k<g0[2]+gw[2]; k++)
    for (j=g0[1]; j<g0[1]+gw[1]; j++)
      for (i=g0[0]; i<g0[0]+gw[0]; i++,pos++) {
        size_t g = (size_t)i + (size_t)gs[0]*((size_t)j + (size_t)gs[1]*(size_t)k);
        for (c=0; c<nsd; c++) iga->geometryX[pos*nsd+c] = X[g*nsd+c];
        if (W) iga->rationalW[pos] = W[g];
      }
  iga->nsd = nsd; iga->rational = W ? 1 : 0;
  return 0;
}

/* IGASetPropertyDim + IGALoadProperty (src/petigaio.c:359-458): the ghosted-local box of the natural-order array [node][npd] */
int orc_set_property(OrcIGA *iga,int npd,const double *A)
{
  int gs[3]={1,1,1},i,j,k,c,dim=iga->dim;
  const int *g0 = iga->node_gstart,*gw = iga->node_gwidth;
  size_t pos = 0;
  if (!iga->setup) ORC_ERR("setup first");
  if (npd < 0) ORC_ERR("Number of properties must be nonnegative");
  free(iga->propertyA); iga->propertyA = NULL; iga->property = npd;
  if (!npd) return 0;
  for (i=0; i<dim; i++) gs[i] = iga->axis[i].span[iga->axis[i].nel-1] + 1;
  iga->propertyA = (double*)xcalloc((size_t)gw[0]*gw[1]*gw[2]*npd,sizeof(double));
  for (k=g0[2]; k<g0[2]+gw[2]; k++)
    for (j=g0[1]; j<g0[1]+gw[1]; j++)
      for (i=g0[0]; i<g0[0]+gw[0]; i++,pos++) {
        size_t g = (size_t)i + (size_t)gs[0]*((size_t)j + (size_t)gs[1]*(size_t)k);
        for (c=0; c<npd; c++) iga->propertyA[pos*npd+c] = A[g*npd+c];
      }
  return 0;
}

/* src/petigaform.c:100-141 */
static void bc_set(OrcBC *bc,int field,double value)
{
  int k;
  for (k=0; k<bc->count; k++) if (bc->field[k] == field) break;
  if (k == bc->count) bc->count++;
  bc->field[k] = field; bc->value[k] = value;
}
int orc_set_boundary_value(OrcIGA *iga,int axis,int side,int field,double v)
{ if (axis<0||axis>=3||side<0||side>=2||field<0||field>=64) ORC_ERR("bad BC argument"); bc_set(&iga->value[axis][side],field,v); return 0; }
int orc_set_boundary_load(OrcIGA *iga,int axis,int side,int field,double v)
{ if (axis<0||axis>=3||side<0||side>=2||field<0||field>=64) ORC_ERR("bad BC argument"); bc_set(&iga->load[axis][side],field,v); return 0; }
int orc_set_boundary_form(OrcIGA *iga,int axis,int side,int flag)
{ if (axis<0||axis>=3||side<0||side>=2) ORC_ERR("bad BC argument"); iga->visit[axis][side] = flag?1:0; return 0; }
int orc_clear_boundary(OrcIGA *iga)
{ int a,s; for (a=0;a<3;a++) for (s=0;s<2;s++) { iga->value[a][s].count=0; iga->load[a][s].count=0; iga->visit[a][s]=0; } return 0; }

/* wrap a ghost index onto the global node grid: src/petigagrid.c:158-163 */
static int wrap(int i,int n) { if (i < 0) return n + i; if (i >= n) return i % n; return i; }

/* src/petigaform.c:273-298 (IGASetFixTable): ghosted-local copy of a global vector */
int orc_set_fixtable(OrcIGA *iga,const double *U)
{
  const int *g0 = iga->node_gstart,*gw = iga->node_gwidth,*ns = iga->node_sizes;
  int i,j,k,c,dof=iga->dof; size_t pos=0;
  free(iga->fixtableU); iga->fixtableU = NULL; iga->fixtable = 0;
  if (!U) return 0;
  iga->fixtableU = (double*)xcalloc((size_t)gw[0]*gw[1]*gw[2]*dof,sizeof(double));
  for (k=g0[2]; k<g0[2]+gw[2]; k++)
    for (j=g0[1]; j<g0[1]+gw[1]; j++)
      for (i=g0[0]; i<g0[0]+gw[0]; i++,pos++) {
        size_t g = (size_t)wrap(i,ns[0]) + (size_t)ns[0]*((size_t)wrap(j,ns[1]) + (size_t)ns[1]*(size_t)wrap(k,ns[2]));
        for (c=0; c<dof; c++) iga->fixtableU[pos*dof+c] = U[g*dof+c];
      }
  iga->fixtable = 1;
  return 0;
}

/* ------------------------------------------------------------------ */
/* Sparsity pattern: src/petigamat.c:197-267 (Stencil, ColumnIndices)  */
/* and :440-535 (rows in natural order, tensor product of per-axis     */
/* ranges, columns sorted as PETSc AIJ stores them).                   */
/* ------------------------------------------------------------------ */
static void stencil(const OrcAxis *ax,int i,int *first,int *last)
{
  int p=ax->p, m=ax->m, n=m-p-1, k;
  const double *U = ax->U;
  k = next_knot(m,U,i,+1);        *first = k - p - 1;
  k = next_knot(m,U,i+p+1,-1);    *last  = k;
  if (!ax->periodic) {
    if (i <= p)   *first = 0;
    if (i >= n-p) *last  = n;
  } else if (i == 0) {
    int kk = n+1, j = next_knot(m,U,kk,+1), s = j-kk, C = p-s, nnp = n-C;
    k = next_knot(m,U,nnp,+1) - nnp;
    *first = k - p - 1;
  }
}

static int cmp_i32(const void *a,const void *b) { int32_t x=*(const int32_t*)a,y=*(const int32_t*)b; return (x>y)-(x<y); }

OrcMat *orc_mat_create(const OrcIGA *iga)
{
  const int *ns = iga->node_sizes; int dof = iga->dof, dim = iga->dim;
  int64_t nnodes = (int64_t)ns[0]*ns[1]*ns[2], nrows = nnodes*dof, r;
  OrcMat *A = (OrcMat*)xcalloc(1,sizeof(OrcMat));
  int first[3][2],i,j,k,c,d;
  int *f[3],*l[3];
  A->nrows = nrows;
  A->rowptr = (int64_t*)xcalloc((size_t)nrows+1,sizeof(int64_t));
  for (d=0; d<3; d++) {
    f[d] = (int*)xcalloc((size_t)ns[d],sizeof(int)); l[d] = (int*)xcalloc((size_t)ns[d],sizeof(int));
    for (i=0; i<ns[d]; i++) {
      if (d < dim) stencil(&iga->axis[d],i,&f[d][i],&l[d][i]); else { f[d][i]=0; l[d][i]=0; }
    }
  }
  (void)first;
  /* pass 1: counts (after wrapping, duplicates cannot occur unless the axis is tiny; dedupe anyway) */
  for (k=0; k<ns[2]; k++) for (j=0; j<ns[1]; j++) for (i=0; i<ns[0]; i++) {
    int64_t node = (int64_t)i + (int64_t)ns[0]*((int64_t)j + (int64_t)ns[1]*k);
    int cnt[3]; int a;
    int idx[3] = {i,j,k};
    for (a=0; a<3; a++) { cnt[a] = l[a][idx[a]] - f[a][idx[a]] + 1; if (cnt[a] > ns[a]) cnt[a] = ns[a]; }
    for (c=0; c<dof; c++) A->rowptr[node*dof+c+1] = (int64_t)cnt[0]*cnt[1]*cnt[2]*dof;
  }
  for (r=0; r<nrows; r++) A->rowptr[r+1] += A->rowptr[r];
  A->colidx = (int32_t*)xcalloc((size_t)A->rowptr[nrows],sizeof(int32_t));
  A->val    = (double*) xcalloc((size_t)A->rowptr[nrows],sizeof(double));
  for (k=0; k<ns[2]; k++) for (j=0; j<ns[1]; j++) for (i=0; i<ns[0]; i++) {
    int64_t node = (int64_t)i + (int64_t)ns[0]*((int64_t)j + (int64_t)ns[1]*k);
    int kk,jj,ii,cc,n=0;
    int32_t *cols = A->colidx + A->rowptr[node*dof];
    int cnt;
    int k1 = l[2][k], j1 = l[1][j], i1 = l[0][i];
    if (k1 - f[2][k] + 1 > ns[2]) k1 = f[2][k] + ns[2] - 1;
    if (j1 - f[1][j] + 1 > ns[1]) j1 = f[1][j] + ns[1] - 1;
    if (i1 - f[0][i] + 1 > ns[0]) i1 = f[0][i] + ns[0] - 1;
    for (kk=f[2][k]; kk<=k1; kk++) for (jj=f[1][j]; jj<=j1; jj++) for (ii=f[0][i]; ii<=i1; ii++) {
      int64_t cn = (int64_t)wrap(ii,ns[0]) + (int64_t)ns[0]*((int64_t)wrap(jj,ns[1]) + (int64_t)ns[1]*wrap(kk,ns[2]));
      for (cc=0; cc<dof; cc++) cols[n++] = (int32_t)(cn*dof+cc);
    }
    cnt = n;
    qsort(cols,(size_t)cnt,sizeof(int32_t),cmp_i32);
    for (c=1; c<dof; c++) memcpy(A->colidx + A->rowptr[node*dof+c],cols,(size_t)cnt*sizeof(int32_t));
  }
  for (d=0; d<3; d++) { free(f[d]); free(l[d]); }
  return A;
}
void orc_mat_destroy(OrcMat *A) { if (!A) return; free(A->rowptr); free(A->colidx); free(A->val); free(A); }
void orc_mat_zero(OrcMat *A) { memset(A->val,0,(size_t)A->rowptr[A->nrows]*sizeof(double)); }

/* MatSetValues(ADD_VALUES) semantics: search the (sorted) row for the column */
static int mat_add(OrcMat *A,int64_t row,int32_t col,double v)
{
  int64_t lo = A->rowptr[row], hi = A->rowptr[row+1]-1;
  while (lo <= hi) {
    int64_t mid = (lo+hi)/2; int32_t c = A->colidx[mid];
    if (c == col) { A->val[mid] += v; return 0; }
    if (c < col) lo = mid+1; else hi = mid-1;
  }
  return 1; /* MAT_NEW_NONZERO_LOCATION_ERR */
}

/* ------------------------------------------------------------------ */
/* Element iterator state (src/petigaelem.c:140-263 IGAElementInit)    */
/* ------------------------------------------------------------------ */
typedef struct {
  OrcIGA *iga;
  int dim,nsd,dof,nen,nqp_max,order;
  int count,index,ID[3],sqp[3];
  int atboundary,boundary_id;
  int *mapping;            /* ghosted-local node index per a */
  double *rationalW,*geometryX,*propertyA; int npd;
  double *weight,*detJac,*normal,*detX,*detS;
  double *basis[5],*shape[5],*mapU[5],*mapX[5];
  int nfix; int *ifix; double *vfix,*ufix;
  int nflux; int *iflux; double *vflux;
  double *wvec[4],*wmat[2],*wval[2];
  double *pvec,*pmat;
  OrcPoint point;
} Elem;

static Elem *elem_create(OrcIGA *iga)
{
  Elem *e = (Elem*)xcalloc(1,sizeof(Elem));
  int k,i,dim=iga->dim,nsd=iga->nsd?iga->nsd:dim,nen=1,nqp=1,dof=iga->dof;
  e->iga=iga; e->dim=dim; e->nsd=nsd; e->dof=dof; e->order=iga->order;
  for (i=0; i<3; i++) { nen *= iga->basis[i].nen; nqp *= iga->basis[i].nqp; }
  e->nen=nen; e->nqp_max=nqp;
  e->count = iga->elem_width[0]*iga->elem_width[1]*iga->elem_width[2];
  e->index = -1; e->boundary_id = -1;
  e->mapping   = (int*)xcalloc((size_t)nen,sizeof(int));
  e->rationalW = (double*)xcalloc((size_t)nen,sizeof(double));
  e->geometryX = (double*)xcalloc((size_t)nen*nsd,sizeof(double));
  e->npd = iga->property; e->propertyA = (double*)xcalloc((size_t)nen*(iga->property ? iga->property : 1),sizeof(double));   /* src/petigaelem.c:155,199 */
  e->weight = (double*)xcalloc((size_t)nqp,sizeof(double));
  e->detJac = (double*)xcalloc((size_t)nqp,sizeof(double));
  e->normal = (double*)xcalloc((size_t)nqp*nsd,sizeof(double));
  e->detX   = (double*)xcalloc((size_t)nqp,sizeof(double));
  e->detS   = (double*)xcalloc((size_t)nqp,sizeof(double));
  for (k=0; k<5; k++) {
    e->basis[k] = (double*)xcalloc((size_t)nqp*nen*ipow(dim,k),sizeof(double));
    e->shape[k] = (double*)xcalloc((size_t)nqp*nen*ipow(nsd,k),sizeof(double));
    e->mapU[k]  = (double*)xcalloc((size_t)nqp*dim*ipow(nsd,k),sizeof(double));
    e->mapX[k]  = (double*)xcalloc((size_t)nqp*nsd*ipow(dim,k),sizeof(double));
  }
  /* identity maps when there is no geometry: src/petigaelem.c:349-358 */
  if (!iga->nsd) {
    int q;
    for (q=0; q<nqp; q++) {
      e->detX[q] = 1.0;
      for (i=0; i<dim; i++) { e->mapX[1][q*nsd*dim + i*(dim+1)] = 1.0; e->mapU[1][q*dim*nsd + i*(dim+1)] = 1.0; }
    }
  }
  e->ifix  = (int*)xcalloc((size_t)nen*dof,sizeof(int));
  e->vfix  = (double*)xcalloc((size_t)nen*dof,sizeof(double));
  e->ufix  = (double*)xcalloc((size_t)nen*dof,sizeof(double));
  e->iflux = (int*)xcalloc((size_t)nen*dof,sizeof(int));
  e->vflux = (double*)xcalloc((size_t)nen*dof,sizeof(double));
  for (k=0; k<4; k++) e->wvec[k] = (double*)xcalloc((size_t)nen*dof,sizeof(double));
  for (k=0; k<2; k++) e->wmat[k] = (double*)xcalloc((size_t)nen*dof*nen*dof,sizeof(double));
  for (k=0; k<2; k++) e->wval[k] = (double*)xcalloc((size_t)nen*dof,sizeof(double));
  e->pvec = (double*)xcalloc((size_t)nen*dof,sizeof(double));
  e->pmat = (double*)xcalloc((size_t)nen*dof*nen*dof,sizeof(double));
  return e;
}

static void elem_destroy(Elem *e)
{
  int k;
  free(e->mapping); free(e->rationalW); free(e->geometryX); free(e->propertyA);
  free(e->weight); free(e->detJac); free(e->normal); free(e->detX); free(e->detS);
  for (k=0; k<5; k++) { free(e->basis[k]); free(e->shape[k]); free(e->mapU[k]); free(e->mapX[k]); }
  free(e->ifix); free(e->vfix); free(e->ufix); free(e->iflux); free(e->vflux);
  for (k=0; k<4; k++) free(e->wvec[k]);
  for (k=0; k<2; k++) { free(e->wmat[k]); free(e->wval[k]); }
  free(e->pvec); free(e->pmat);
  free(e);
}

/* src/petigaelem.c:693-755 (IGAElementBuildClosure) */
static void elem_closure(Elem *e)
{
  OrcIGA *iga = e->iga;
  const OrcBasis *BD = iga->basis;
  const int *ID = e->ID,*start = iga->node_gstart,*width = iga->node_gwidth;
  int ia,ja,ka,a=0;
  int inen=BD[0].nen, ioff=BD[0].offset[ID[0]];
  int jnen=BD[1].nen, joff=BD[1].offset[ID[1]];
  int knen=BD[2].nen, koff=BD[2].offset[ID[2]];
  int jstride = width[0], kstride = width[0]*width[1];
  for (ka=0; ka<knen; ka++) for (ja=0; ja<jnen; ja++) for (ia=0; ia<inen; ia++) {
    int iA = (ioff+ia) - start[0], jA = (joff+ja) - start[1], kA = (koff+ka) - start[2];
    e->mapping[a++] = iA + jA*jstride + kA*kstride;
  }
  if (iga->rational) for (a=0; a<e->nen; a++) e->rationalW[a] = iga->rationalW[e->mapping[a]];
  if (iga->nsd) { int i,nsd=e->nsd; for (a=0; a<e->nen; a++) for (i=0; i<nsd; i++) e->geometryX[i+a*nsd] = iga->geometryX[(size_t)e->mapping[a]*nsd+i]; }
  if (iga->property && iga->propertyA) { int i,npd=e->npd; for (a=0; a<e->nen; a++) for (i=0; i<npd; i++) e->propertyA[i+a*npd] = iga->propertyA[(size_t)e->mapping[a]*npd+i]; }   /* src/petigaelem.c:745-752 */
}

/* ------------------------------------------------------------------ */
/* Numeric kernels K1..K7                                              */
/* ------------------------------------------------------------------ */

/* src/petigaelem.c:763-775 (IGA_Quadrature_SIZE): trailing non-positive weights are dropped */
static int quad_size(const OrcBasis *BD,const int *ID,int NQ[3])
{
  int i;
  for (i=0; i<3; i++) {
    int q = BD[i].nqp - 1; const double *w = BD[i].weight + ID[i]*BD[i].nqp;
    NQ[i] = 1; while (q >= 0 && w[q] <= 0) q--; NQ[i] += q;
  }
  return NQ[0]*NQ[1]*NQ[2];
}

/* 1-D slices an element pass reads: interior (src/petigaelem.c:777-786) or a face (:788-792) */
typedef struct { int nq,na; const double *X,*W,*N; double J; } Slice;

/* K1: src/petiga{1,2,3}d.F90 IGA_Quadrature_*D -- i fastest; J is one scalar broadcast */
static void k_quadrature(int dim,const Slice s[3],double *X,double *W,double *J)
{
  int iq,jq,kq,q=0; double JJ = s[0].J*s[1].J*s[2].J;
  for (kq=0; kq<s[2].nq; kq++) for (jq=0; jq<s[1].nq; jq++) for (iq=0; iq<s[0].nq; iq++,q++) {
    X[q*dim+0] = s[0].X[iq];
    if (dim > 1) X[q*dim+1] = s[1].X[jq];
    if (dim > 2) X[q*dim+2] = s[2].X[kq];
    W[q] = s[0].W[iq]*s[1].W[jq]*s[2].W[kq];
    J[q] = JJ;
  }
}

/* K2: src/petiga3d.F90:32-233 (TensorBasisFuns) and the 1-D/2-D siblings.  The component of
 * N_k with derivative multi-index (d_1..d_k) is the product over axes of the 1-D derivative of
 * order (#times that axis appears); flat component index = d_1 + dim*d_2 + dim^2*d_3 ... */
static void k_basisfuns(int order,int dim,const Slice s[3],double *N[5])
{
  int iq,jq,kq,ia,ja,ka,k,q=0,nen = s[0].na*s[1].na*s[2].na;
  for (kq=0; kq<s[2].nq; kq++) for (jq=0; jq<s[1].nq; jq++) for (iq=0; iq<s[0].nq; iq++,q++) {
    const double *iN = s[0].N + (size_t)iq*s[0].na*5;
    const double *jN = s[1].N + (size_t)jq*s[1].na*5;
    const double *kN = s[2].N + (size_t)kq*s[2].na*5;
    int a = 0;
    for (ka=0; ka<s[2].na; ka++) for (ja=0; ja<s[1].na; ja++) for (ia=0; ia<s[0].na; ia++,a++) {
      int nc = 1;
      for (k=0; k<=order; k++) {
        double *out = N[k] + ((size_t)q*nen + a)*nc;
        int f;
        for (f=0; f<nc; f++) {
          int cnt[3] = {0,0,0}, t = f, r;
          for (r=0; r<k; r++) { cnt[t%dim]++; t /= dim; }
          out[f] = iN[ia*5+cnt[0]] * jN[ja*5+cnt[1]] * kN[ka*5+cnt[2]];
        }
        nc *= dim;
      }
    }
  }
}

/* K3: src/petigarat.f90.in:3-57 (Rationalize), one quadrature point.  R_k are [nen][dim^k]. */
static void k_rationalize(int order,int dim,int nen,const double *W,double *R0,double *R1,double *R2,double *R3,double *R4)
{
  int a,i,j,k; double W0=0,W1[3],W2[9],W3[27];
  int d1=dim,d2=dim*dim,d3=dim*dim*dim;
  for (a=0; a<nen; a++) R0[a] = W[a]*R0[a];
  for (a=0; a<nen; a++) W0 += R0[a];
  for (a=0; a<nen; a++) R0[a] /= W0;
  if (order < 1) return;
  for (i=0; i<dim; i++) {
    double s=0; for (a=0; a<nen; a++) s += W[a]*R1[a*d1+i];
    W1[i] = s;
    for (a=0; a<nen; a++) R1[a*d1+i] = W[a]*R1[a*d1+i] - R0[a]*W1[i];
  }
  for (a=0; a<nen*d1; a++) R1[a] /= W0;
  if (order < 2) return;
  for (j=0; j<dim; j++) for (i=0; i<dim; i++) {
    double s=0; int f=i+dim*j;
    for (a=0; a<nen; a++) s += W[a]*R2[a*d2+f];
    W2[f] = s;
    for (a=0; a<nen; a++)
      R2[a*d2+f] = W[a]*R2[a*d2+f] - R0[a]*W2[f] - R1[a*d1+i]*W1[j] - R1[a*d1+j]*W1[i];
  }
  for (a=0; a<nen*d2; a++) R2[a] /= W0;
  if (order < 3) return;
  for (k=0; k<dim; k++) for (j=0; j<dim; j++) for (i=0; i<dim; i++) {
    double s=0; int f=i+dim*(j+dim*k);
    for (a=0; a<nen; a++) s += W[a]*R3[a*d3+f];
    W3[f] = s;
    for (a=0; a<nen; a++)
      R3[a*d3+f] = W[a]*R3[a*d3+f] - R0[a]*W3[f]
        - R1[a*d1+i]*W2[j+dim*k] - R1[a*d1+j]*W2[i+dim*k] - R1[a*d1+k]*W2[i+dim*j]
        - R2[a*d2+j+dim*k]*W1[i] - R2[a*d2+i+dim*k]*W1[j] - R2[a*d2+i+dim*j]*W1[k];
  }
  for (a=0; a<nen*d3; a++) R3[a] /= W0;
  if (order < 4) return;
  memset(R4,0,sizeof(double)*(size_t)nen*d3*dim);   /* "XXX Implement" in the reference: zero */
}

/* K4: src/petigamapgeo.f90.in:3-71 (GeometryMap), one point: X_k[i][f] = sum_a X[a][i]*M_k[a][f] */
static void k_geometrymap(int order,int dim,int nsd,int nen,const double *X,double *const M[5],double *Xk[5],size_t q)
{
  int k,a,i,f,nc=1;
  for (k=0; k<=order; k++) {
    const double *Mk = M[k] + q*nen*nc; double *out = Xk[k] + q*nsd*nc;
    for (i=0; i<nsd*nc; i++) out[i] = 0;
    for (a=0; a<nen; a++) for (i=0; i<nsd; i++) { double x = X[a*nsd+i]; for (f=0; f<nc; f++) out[i*nc+f] += x*Mk[a*nc+f]; }
    nc *= dim;
  }
}

/* src/petigadet.f90.in:3-20 / src/petigainv.f90.in:3-32; A is column-major A(r,c) = A[r+d*c] */
static double k_det(int d,const double *A)
{
#define A_(r,c) A[(r)+d*(c)]
  if (d == 1) return A_(0,0);
  if (d == 2) return A_(0,0)*A_(1,1) - A_(1,0)*A_(0,1);
  return + A_(0,0)*(A_(1,1)*A_(2,2) - A_(2,1)*A_(1,2))
         - A_(1,0)*(A_(0,1)*A_(2,2) - A_(2,1)*A_(0,2))
         + A_(2,0)*(A_(0,1)*A_(1,2) - A_(1,1)*A_(0,2));
}
static void k_inv(int d,double det,const double *A,double *B)
{
#define B_(r,c) B[(r)+d*(c)]
  int i;
  if (d == 1) { B[0] = 1/det; return; }
  if (d == 2) {
    B_(0,0) = + A_(1,1); B_(1,0) = - A_(1,0); B_(0,1) = - A_(0,1); B_(1,1) = + A_(0,0);
    for (i=0; i<4; i++) B[i] /= det;
    return;
  }
  B_(0,0) = + A_(1,1)*A_(2,2) - A_(1,2)*A_(2,1);
  B_(1,0) = - A_(1,0)*A_(2,2) + A_(1,2)*A_(2,0);
  B_(2,0) = + A_(1,0)*A_(2,1) - A_(1,1)*A_(2,0);
  B_(0,1) = - A_(0,1)*A_(2,2) + A_(0,2)*A_(2,1);
  B_(1,1) = + A_(0,0)*A_(2,2) - A_(0,2)*A_(2,0);
  B_(2,1) = - A_(0,0)*A_(2,1) + A_(0,1)*A_(2,0);
  B_(0,2) = + A_(0,1)*A_(1,2) - A_(0,2)*A_(1,1);
  B_(1,2) = - A_(0,0)*A_(1,2) + A_(0,2)*A_(1,0);
  B_(2,2) = + A_(0,0)*A_(1,1) - A_(0,1)*A_(1,0);
  for (i=0; i<9; i++) B[i] /= det;
#undef A_
#undef B_
}

/* K5: src/petigamapinv.f90.in:3-74 (InverseMap), one point, dim == nsd == d.
 * Fortran views: X1(a,i) X2(b,a,k) X3(c,b,a,l) ; E1(i,a) E2(j,i,c) E3(k,j,i,dd) */
static void k_inversemap(int order,int d,const double *X1,const double *X2,const double *X3,double *dX,double *E1,double *E2,double *E3,double *E4)
{
  int i,j,k,l,a,b,c,dd,d2=d*d,d3=d*d*d;
#define X2_(b,a,k)   X2[(b)+d*(a)+d2*(k)]
#define X3_(c,b,a,l) X3[(c)+d*(b)+d2*(a)+d3*(l)]
#define E1_(i,a)     E1[(i)+d*(a)]
#define E2_(j,i,c)   E2[(j)+d*(i)+d2*(c)]
#define E3_(k,j,i,e) E3[(k)+d*(j)+d2*(i)+d3*(e)]
  if (order < 1) return;
  *dX = k_det(d,X1);
  k_inv(d,*dX,X1,E1);
  if (order < 2) return;
  memset(E2,0,sizeof(double)*(size_t)d3);
  for (i=0;i<d;i++) for (j=0;j<d;j++) for (k=0;k<d;k++) for (a=0;a<d;a++) for (b=0;b<d;b++) for (c=0;c<d;c++)
    E2_(j,i,c) -= X2_(b,a,k)*E1_(i,a)*E1_(j,b)*E1_(k,c);
  if (order < 3) return;
  memset(E3,0,sizeof(double)*(size_t)d3*d);
  for (dd=0;dd<d;dd++) for (i=0;i<d;i++) for (j=0;j<d;j++) for (k=0;k<d;k++) for (a=0;a<d;a++) for (b=0;b<d;b++) for (l=0;l<d;l++) {
    for (c=0;c<d;c++) E3_(k,j,i,dd) -= X3_(c,b,a,l)*E1_(i,a)*E1_(j,b)*E1_(k,c)*E1_(l,dd);
    E3_(k,j,i,dd) -= X2_(b,a,l)*(E1_(i,a)*E2_(k,j,b) + E1_(j,b)*E2_(k,i,a) + E1_(k,b)*E2_(j,i,a))*E1_(l,dd);
  }
  if (order < 4) return;
  memset(E4,0,sizeof(double)*(size_t)d3*d*d);
}

/* K6: src/petigamapshf.f90.in:3-83 (ShapeFunctions), one point.
 * N1(a,node) N2(b,a,node) N3(c,b,a,node) ; R1(i,node) R2(j,i,node) R3(k,j,i,node) */
static void k_shapefuns(int order,int d,int nen,const double *E1,const double *E2,const double *E3,
                        const double *N1,const double *N2,const double *N3,double *R1,double *R2,double *R3,double *R4)
{
  int n,i,j,k,a,b,c,d2=d*d,d3=d*d*d;
  if (order < 1) return;
  for (n=0; n<nen; n++) for (i=0; i<d; i++) {
    double s = 0; for (a=0; a<d; a++) s += N1[n*d+a]*E1_(i,a);
    R1[n*d+i] = s;
  }
  if (order < 2) return;
  for (n=0; n<nen; n++) for (i=0; i<d; i++) for (j=0; j<d; j++) {
    double s = 0;
    for (a=0; a<d; a++) {
      for (b=0; b<d; b++) s += N2[n*d2+b+d*a]*E1_(i,a)*E1_(j,b);
      s += N1[n*d+a]*E2_(j,i,a);
    }
    R2[n*d2+j+d*i] = s;
  }
  if (order < 3) return;
  for (n=0; n<nen; n++) for (i=0; i<d; i++) for (j=0; j<d; j++) for (k=0; k<d; k++) {
    double s = 0;
    for (a=0; a<d; a++) {
      for (b=0; b<d; b++) {
        for (c=0; c<d; c++) s += N3[n*d3+c+d*b+d2*a]*E1_(i,a)*E1_(j,b)*E1_(k,c);
        s += N2[n*d2+b+d*a]*(E1_(i,a)*E2_(k,j,b) + E1_(j,b)*E2_(k,i,a) + E1_(k,b)*E2_(j,i,a));
      }
      s += N1[n*d+a]*E3_(k,j,i,a);
    }
    R3[n*d3+k+d*j+d2*i] = s;
  }
  if (order < 4) return;
  memset(R4,0,sizeof(double)*(size_t)nen*d3*d);
#undef X2_
#undef X3_
#undef E1_
#undef E2_
#undef E3_
}

/* K7: src/petigaval.F90:45-99 (IGA_GetNormal); F(a,i) column-major = C mapX[1][i][a] */
static void k_normal(int dim,int axis,int side,const double *F,double *dS,double *N)
{
#define F_(r,c) F[(r)+dim*(c)]
  int i;
  if (dim == 3) {
    double s[3],t[3];
    int r1 = (axis+1)%3, r2 = (axis+2)%3;
    for (i=0;i<3;i++) { s[i] = F_(r1,i); t[i] = F_(r2,i); }
    N[0] = s[1]*t[2] - s[2]*t[1]; N[1] = s[2]*t[0] - s[0]*t[2]; N[2] = s[0]*t[1] - s[1]*t[0];
    *dS = sqrt(N[0]*N[0]+N[1]*N[1]+N[2]*N[2]);
    for (i=0;i<3;i++) N[i] /= *dS;
  } else if (dim == 2) {
    double t[2];
    if (axis == 0) { t[0] = +F_(1,0); t[1] = +F_(1,1); } else { t[0] = -F_(0,0); t[1] = -F_(0,1); }
    N[0] = +t[1]; N[1] = -t[0];
    *dS = sqrt(N[0]*N[0]+N[1]*N[1]);
    N[0] /= *dS; N[1] /= *dS;
  } else { *dS = 1; N[0] = 1; }
  if (side == 0) for (i=0;i<dim;i++) N[i] = -N[i];
#undef F_
}

/* src/petigaelem.c:794-1033 (IGAElementBuildTabulation) */
static int elem_tabulate(Elem *e)
{
  OrcIGA *iga = e->iga;
  const OrcBasis *BD = iga->basis;
  const int *ID = e->ID; int *NQ = e->sqp;
  int ord=e->order, dim=e->dim, nsd=e->nsd, nen=e->nen, nqp, q, i, axis=-1, side=-1;
  Slice s[3];
  nqp = quad_size(BD,ID,NQ);
  if (e->atboundary) { axis = e->boundary_id/2; side = e->boundary_id%2; nqp /= NQ[axis]; NQ[axis] = 1; }
  e->point.count = nqp;
  for (i=0; i<3; i++) {
    if (e->atboundary && i == axis) {
      s[i].nq = 1; s[i].na = BD[i].nen; s[i].X = &BD[i].bnd_point[side]; s[i].W = &BD[i].bnd_weight; s[i].J = BD[i].bnd_detJac; s[i].N = BD[i].bnd_value[side];
    } else {
      s[i].nq = NQ[i]; s[i].na = BD[i].nen;
      s[i].X = BD[i].point + ID[i]*BD[i].nqp; s[i].W = BD[i].weight + ID[i]*BD[i].nqp; s[i].J = BD[i].detJac[ID[i]];
      s[i].N = BD[i].value + (size_t)ID[i]*BD[i].nqp*BD[i].nen*5;
    }
  }
  k_quadrature(dim,s,e->mapU[0],e->weight,e->detJac);
  k_basisfuns(ord,dim,s,e->basis);
  if (iga->rational) {
    int d1=dim,d2=dim*dim,d3=d2*dim,d4=d3*dim;
    for (q=0; q<nqp; q++)
      k_rationalize(ord,dim,nen,e->rationalW,e->basis[0]+(size_t)q*nen,e->basis[1]+(size_t)q*nen*d1,e->basis[2]+(size_t)q*nen*d2,e->basis[3]+(size_t)q*nen*d3,e->basis[4]+(size_t)q*nen*d4);
  }
  if (iga->nsd) {
    /* src/petigaelem.c:940-964: the same sums for any nsd (IGA_GeometryMap, src/petigaval.F90:10-43) */
    for (q=0; q<nqp; q++) k_geometrymap(ord,dim,nsd,nen,e->geometryX,e->basis,e->mapX,(size_t)q);
    if (dim == nsd) {   /* src/petigaelem.c:966: inverse map and physical shape functions only when the map is square */
      int d1=dim,d2=dim*dim,d3=d2*dim,d4=d3*dim;
      for (q=0; q<nqp; q++)
        k_inversemap(ord,dim,e->mapX[1]+(size_t)q*d2,e->mapX[2]+(size_t)q*d3,e->mapX[3]+(size_t)q*d4,&e->detX[q],
                     e->mapU[1]+(size_t)q*d2,e->mapU[2]+(size_t)q*d3,e->mapU[3]+(size_t)q*d4,e->mapU[4]+(size_t)q*d4*dim);
      memcpy(e->shape[0],e->basis[0],sizeof(double)*(size_t)nqp*nen);
      for (q=0; q<nqp; q++)
        k_shapefuns(ord,dim,nen,e->mapU[1]+(size_t)q*d2,e->mapU[2]+(size_t)q*d3,e->mapU[3]+(size_t)q*d4,
                    e->basis[1]+(size_t)q*nen*d1,e->basis[2]+(size_t)q*nen*d2,e->basis[3]+(size_t)q*nen*d3,
                    e->shape[1]+(size_t)q*nen*d1,e->shape[2]+(size_t)q*nen*d2,e->shape[3]+(size_t)q*nen*d3,e->shape[4]+(size_t)q*nen*d4);
    }
  }
  if (e->atboundary) {
    if (iga->nsd && dim == nsd) for (q=0; q<nqp; q++) k_normal(dim,axis,side,e->mapX[1]+(size_t)q*nsd*dim,&e->detS[q],e->normal+(size_t)q*nsd);   /* src/petigaelem.c:1016-1021 */
    else { memset(e->normal,0,sizeof(double)*(size_t)nqp*nsd); for (q=0; q<nqp; q++) { e->detS[q] = 1.0; e->normal[q*nsd+axis] = side ? 1.0 : -1.0; } }
  }
  if (iga->nsd && dim == nsd) {   /* src/petigaelem.c:1024 */
    if (!e->atboundary) for (q=0; q<nqp; q++) e->detJac[q] *= e->detX[q];
    else                for (q=0; q<nqp; q++) e->detJac[q] *= e->detS[q];
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Iteration: elements, forms, points                                  */
/* ------------------------------------------------------------------ */

/* src/petiga2d.F90:276-346 / src/petiga3d.F90:379-464 (IGA_BoundaryArea_2D/3D): dS = sum_q w_q sqrt|det(F F^T)| over the
 * face of the element, F = d(face map)/d(knot coordinates) from the first / last layer of the element's control points,
 * raw Gauss weights (sum 2 per axis), basis derivatives w.r.t. the knot coordinate, local Rationalize. */
static double boundary_area_dS(const Elem *e,int dir,int side)
{
  const OrcIGA *iga = e->iga; int dim=e->dim, nsd=e->nsd, fd=dim-1;
  int ax[2]={0,0}, nq[2]={1,1}, na[2]={1,1}, m[3]={1,1,1}, i,k,q0,q1,a0,a1,c,r,s2;
  const double *W[2]={NULL,NULL}, *N[2]={NULL,NULL}; double dS=0;
  for (i=0; i<dim; i++) m[i] = iga->basis[i].nen;
  int qs[3]; quad_size(iga->basis,e->ID,qs);   /* IGA_Quadrature_SIZE(BD,ID,qshape), src/petigaelem.c:1137: the trimmed counts of a reduced rule */
  for (k=0,i=0; i<dim; i++) { if (i==dir) continue; ax[k]=i; nq[k]=qs[i]; na[k]=iga->basis[i].nen;
    W[k] = iga->basis[i].weight + (size_t)e->ID[i]*iga->basis[i].nqp; N[k] = iga->basis[i].value + (size_t)e->ID[i]*iga->basis[i].nqp*na[k]*5; k++; }
  for (q1=0; q1<nq[1]; q1++) for (q0=0; q0<nq[0]; q0++) {
    double N0[64], N1[2][64], Xw[64], F[2][3], M[2][2]={{0,0},{0,0}}, detJ=1, W0=0, S1[2]={0,0}; int nen = na[0]*na[1];
    for (a1=0; a1<na[1]; a1++) for (a0=0; a0<na[0]; a0++) {
      int a = a0 + na[0]*a1, loc[3], la; double n0 = N[0][(q0*na[0]+a0)*5+0], d0 = N[0][(q0*na[0]+a0)*5+1], n1 = 1, d1 = 0;
      if (fd == 2) { n1 = N[1][(q1*na[1]+a1)*5+0]; d1 = N[1][(q1*na[1]+a1)*5+1]; }
      N0[a] = n0*n1; N1[0][a] = d0*n1; N1[1][a] = n0*d1;
      loc[0]=loc[1]=loc[2]=0; loc[dir] = side ? m[dir]-1 : 0; loc[ax[0]] = a0; if (fd == 2) loc[ax[1]] = a1;
      la = loc[0] + m[0]*(loc[1] + m[1]*loc[2]);
      Xw[a] = iga->rational ? e->rationalW[la] : 1.0;
      (void)la;
    }
    if (iga->rational) {   /* local Rationalize */
      for (k=0; k<nen; k++) { N0[k] *= Xw[k]; W0 += N0[k]; }
      for (k=0; k<nen; k++) N0[k] /= W0;
      for (r=0; r<fd; r++) { for (k=0; k<nen; k++) S1[r] += Xw[k]*N1[r][k]; for (k=0; k<nen; k++) N1[r][k] = (Xw[k]*N1[r][k] - N0[k]*S1[r])/W0; }
    }
    if (iga->nsd) {        /* local Jacobian: F = N1 X^T, M = F F^T, J = sqrt|det M| */
      for (r=0; r<fd; r++) for (c=0; c<nsd; c++) F[r][c] = 0;
      for (a1=0; a1<na[1]; a1++) for (a0=0; a0<na[0]; a0++) {
        int a = a0 + na[0]*a1, loc[3]={0,0,0}, la;
        loc[dir] = side ? m[dir]-1 : 0; loc[ax[0]] = a0; if (fd == 2) loc[ax[1]] = a1;
        la = loc[0] + m[0]*(loc[1] + m[1]*loc[2]);
        for (r=0; r<fd; r++) for (c=0; c<nsd; c++) F[r][c] += N1[r][a]*e->geometryX[(size_t)la*nsd+c];
      }
      for (r=0; r<fd; r++) for (s2=0; s2<fd; s2++) { M[r][s2]=0; for (c=0; c<nsd; c++) M[r][s2] += F[r][c]*F[s2][c]; }
      detJ = (fd == 1) ? M[0][0] : M[0][0]*M[1][1]-M[0][1]*M[1][0];
      detJ = sqrt(fabs(detJ));
    }
    dS += detJ * W[0][q0]*(fd == 2 ? W[1][q1] : 1.0);
  }
  return dS;
}

/* src/petigaelem.c:1118-1164 (BoundaryArea) */
static int boundary_area(const Elem *e,int dir,int side,double *A)
{
  int i,dim=e->dim; double a=1;
  if (dim == 1) { *A = 1; return 0; }
  for (i=0; i<dim; i++) if (i != dir) a *= e->iga->basis[i].detJac[e->ID[i]]/(double)e->iga->basis[i].nen;
  if (!e->iga->nsd) a *= (dim==2) ? 2 : 4;   /* sum(W) = 2 */
  else {
    if (e->nen > 64*8) ORC_ERR("boundary_area: element too large");
    for (i=0; i<dim; i++) if (i != dir && e->iga->basis[i].nen > 8) ORC_ERR("boundary_area: degree too large");
    a *= boundary_area_dS(e,dir,side);
  }
  *A = a;
  return 0;
}

/* src/petigaelem.c:1166-1212 (AddFixa / AddFlux) */
static void add_fixa(Elem *e,const OrcBC *bc,int a)
{
  int j,k,count=e->nfix,dof=e->dof;
  for (k=0; k<bc->count; k++) {
    int c = bc->field[k], idx = a*dof + c; double val = bc->value[k];
    if (c >= dof) continue;
    if (e->iga->fixtable) val = e->iga->fixtableU[c + (size_t)e->mapping[a]*dof];
    for (j=0; j<count; j++) if (e->ifix[j] == idx) break;
    if (j == count) count++;
    e->ifix[j] = idx; e->vfix[j] = val;
  }
  e->nfix = count;
}
static void add_flux(Elem *e,const OrcBC *bc,int a,double A)
{
  int j,k,count=e->nflux,dof=e->dof;
  for (k=0; k<bc->count; k++) {
    int c = bc->field[k], idx = a*dof + c; double val = bc->value[k];
    if (c >= dof) continue;
    for (j=0; j<count; j++) if (e->iflux[j] == idx) break;
    if (j == count) e->vflux[count++] = 0.0;
    e->iflux[j] = idx; e->vflux[j] += val*A;
  }
  e->nflux = count;
}

/* src/petigaelem.c:1214-1283 (BuildFix / IGAElementBuildFix) */
static int elem_buildfix(Elem *e)
{
  OrcIGA *iga = e->iga; int i,dim=e->dim;
  e->nfix = 0; e->nflux = 0;
  for (i=0; i<dim; i++) {
    int side, last = iga->elem_sizes[i]-1;
    if (iga->axis[i].periodic) continue;
    for (side=0; side<2; side++) {
      const OrcBC *bcv = &iga->value[i][side], *bcl = &iga->load[i][side];
      int S[3]={0,0,0},E[3]={1,1,1},ia,ja,ka,d; double Area = 1;
      if (e->ID[i] != (side ? last : 0)) continue;
      if (!bcv->count && !bcl->count) continue;
      if (bcl->count && boundary_area(e,i,side,&Area)) return 1;
      for (d=0; d<dim; d++) E[d] = iga->basis[d].nen;
      { int jstride = E[0], kstride = E[0]*E[1];
        if (side) S[i] = E[i]-1; else E[i] = S[i]+1;
        for (ka=S[2]; ka<E[2]; ka++) for (ja=S[1]; ja<E[1]; ja++) for (ia=S[0]; ia<E[0]; ia++) {
          int a = ia + ja*jstride + ka*kstride;
          add_fixa(e,bcv,a); add_flux(e,bcl,a,Area);
        } }
    }
  }
  return 0;
}

/* src/petigaelem.c:375-410 (IGANextElement): i fastest */
static int elem_next(Elem *e)
{
  int i,index,coord;
  index = ++e->index;
  if (index >= e->count) { e->index = -1; return 0; }
  for (i=0; i<e->dim; i++) {
    coord = index % e->iga->elem_width[i];
    index = (index - coord)/e->iga->elem_width[i];
    e->ID[i] = coord + e->iga->elem_start[i];
  }
  for (i=e->dim; i<3; i++) e->ID[i] = 0;
  elem_closure(e);
  if (elem_buildfix(e)) return -1;
  e->boundary_id = -1; e->atboundary = 0;
  return 1;
}

/* src/petigaelem.c:427-447 (IGAElementNextForm) */
static int elem_next_form(Elem *e)
{
  int dim = e->dim;
  while (++e->boundary_id < 2*dim) {
    int i = e->boundary_id/2, s = e->boundary_id%2;
    int el = s ? e->iga->elem_sizes[i]-1 : 0;
    if (e->ID[i] != el) continue;
    if (!e->iga->visit[i][s]) continue;
    e->atboundary = 1;
    return 1;
  }
  if (e->boundary_id++ == 2*dim) { e->atboundary = 0; return 1; }
  e->atboundary = 0; e->boundary_id = -1;
  return 0;
}

/* src/petigaelem.c:458-475 (IGAElementBeginPoint) */
static int elem_begin_point(Elem *e)
{
  OrcPoint *p = &e->point;
  memset(p,0,sizeof(*p));
  p->iga = e->iga; p->index = -1;
  p->atboundary = e->atboundary; p->boundary_id = e->boundary_id;
  p->neq = e->nen; p->nen = e->nen; p->dof = e->dof; p->dim = e->dim; p->nsd = e->nsd;
  p->rational = e->iga->rational ? e->rationalW : NULL;
  p->geometry = e->iga->nsd ? e->geometryX : NULL;
  p->property = (e->iga->property && e->iga->propertyA) ? e->propertyA : NULL; p->npd = p->property ? e->npd : 0;   /* src/petigaelem.c:370 */
  p->ID[0]=e->ID[0]; p->ID[1]=e->ID[1]; p->ID[2]=e->ID[2];
  return elem_tabulate(e);
}

/* src/petigaelem.c:477-592 (IGAElementNextPoint): pointer bumps; shape aliases basis when no geometry */
static int elem_next_point(Elem *e)
{
  OrcPoint *p = &e->point;
  int k,nen=p->nen,dim=p->dim,nsd=p->nsd,index;
  int geo = e->iga->nsd ? 1 : 0;
  index = ++p->index;
  if (index == 0) {
    p->weight = e->weight; p->detJac = e->detJac; p->point = e->mapU[0]; p->normal = e->normal;
    for (k=0; k<5; k++) {
      p->basis[k] = e->basis[k];
      p->shape[k] = (geo && dim==nsd) ? e->shape[k] : e->basis[k];
      p->mapU[k]  = e->mapU[k];
      p->mapX[k]  = geo ? e->mapX[k] : e->mapU[k];
    }
    p->detX = e->detX; p->detS = e->detS;
    return 1;
  }
  if (index >= p->count) { p->index = -1; return 0; }
  p->weight += 1; p->detJac += 1; p->point += dim; p->normal += nsd;
  for (k=0; k<5; k++) {
    p->basis[k] += nen*ipow(dim,k);
    p->shape[k] += nen*ipow((geo && dim==nsd)?nsd:dim,k);
    p->mapU[k]  += dim*ipow(nsd,k);
    p->mapX[k]  += geo ? nsd*ipow(dim,k) : dim*ipow(nsd,k);
  }
  p->detX += 1; p->detS += 1;
  return 1;
}

/* src/petigapoint.c:451-465 (IGAPointAddArray) */
static void point_add(const OrcPoint *p,size_t n,const double *a,double *A)
{
  size_t i; double JW = p->detJac[0]*p->weight[0];
  for (i=0; i<n; i++) A[i] += a[i]*JW;
}

/* src/petigaelem.c:1074-1100 (IGAElementGetValues) from a *global natural* vector via the
 * ghost wrap (stands in for IGAGetLocalVecArray's g2l scatter, src/petigavec.c:256-269) */
static void elem_get_values(const Elem *e,const double *Uglobal,double *U)
{
  const OrcIGA *iga = e->iga; const int *g0=iga->node_gstart,*gw=iga->node_gwidth,*ns=iga->node_sizes;
  int a,i,dof=e->dof;
  if (!Uglobal) { memset(U,0,sizeof(double)*(size_t)e->nen*dof); return; }
  for (a=0; a<e->nen; a++) {
    int m = e->mapping[a];
    int iA = m % gw[0] + g0[0], jA = (m/gw[0]) % gw[1] + g0[1], kA = m/(gw[0]*gw[1]) + g0[2];
    size_t g = (size_t)wrap(iA,ns[0]) + (size_t)ns[0]*((size_t)wrap(jA,ns[1]) + (size_t)ns[1]*(size_t)wrap(kA,ns[2]));
    for (i=0; i<dof; i++) U[a*dof+i] = Uglobal[g*dof+i];
  }
}
static int64_t elem_global_node(const Elem *e,int a)
{
  const OrcIGA *iga = e->iga; const int *g0=iga->node_gstart,*gw=iga->node_gwidth,*ns=iga->node_sizes;
  int m = e->mapping[a];
  int iA = m % gw[0] + g0[0], jA = (m/gw[0]) % gw[1] + g0[1], kA = m/(gw[0]*gw[1]) + g0[2];
  return (int64_t)wrap(iA,ns[0]) + (int64_t)ns[0]*((int64_t)wrap(jA,ns[1]) + (int64_t)ns[1]*wrap(kA,ns[2]));
}

/* src/petigaelem.c:1327-1358 */
static void elem_del_values(const Elem *e,double *V) { int f; for (f=0; f<e->nfix; f++) V[e->ifix[f]] = 0.0; }
static void elem_fix_values(Elem *e,double *U) { int f; for (f=0; f<e->nfix; f++) { int k=e->ifix[f]; e->ufix[f] = U[k]; U[k] = e->vfix[f]; } }

/* src/petigaelem.c:1360-1389 (IGAElementFixSystem, Galerkin branch) */
static void elem_fix_system(const Elem *e,double *K,double *F)
{
  int M = e->nen*e->dof, N = M, f,i,j;
  for (f=0; f<e->nflux; f++) F[e->iflux[f]] += e->vflux[f];
  for (f=0; f<e->nfix; f++) {
    int k = e->ifix[f]; double v = e->vfix[f];
    for (i=0; i<M; i++) F[i] -= K[i*N+k]*v;
    for (i=0; i<M; i++) K[i*N+k] = 0.0;
    for (j=0; j<N; j++) K[k*N+j] = 0.0;
    K[k*N+k] = 1.0;
    F[k] = v;
  }
}
/* src/petigaelem.c:1441-1462 */
static void elem_fix_function(const Elem *e,double *F)
{
  int f;
  for (f=0; f<e->nflux; f++) F[e->iflux[f]] -= e->vflux[f];
  for (f=0; f<e->nfix; f++) F[e->ifix[f]] = e->ufix[f] - e->vfix[f];
}
/* src/petigaelem.c:1483-1500 */
static void elem_fix_jacobian(const Elem *e,double *J)
{
  int M = e->nen*e->dof, N = M, f,i,j;
  for (f=0; f<e->nfix; f++) {
    int k = e->ifix[f];
    for (i=0; i<M; i++) J[i*N+k] = 0.0;
    for (j=0; j<N; j++) J[k*N+j] = 0.0;
    J[k*N+k] = 1.0;
  }
}

/* src/petigaelem.c:1525-1559 (IGAElementAssembleVec/Mat): ADD_VALUES through the local-to-global map */
static void elem_assemble_vec(const Elem *e,const double *F,double *vec)
{
  int a,i,dof=e->dof;
  for (a=0; a<e->nen; a++) { int64_t g = elem_global_node(e,a); for (i=0; i<dof; i++) vec[g*dof+i] += F[a*dof+i]; }
}
static int elem_assemble_mat(const Elem *e,const double *K,OrcMat *A)
{
  int a,b,i,j,dof=e->dof,nen=e->nen,N=nen*dof;
  for (a=0; a<nen; a++) { int64_t ga = elem_global_node(e,a);
    for (i=0; i<dof; i++) for (b=0; b<nen; b++) { int64_t gb = elem_global_node(e,b);
      for (j=0; j<dof; j++)
        if (mat_add(A,ga*dof+i,(int32_t)(gb*dof+j),K[(a*dof+i)*N + b*dof+j])) ORC_ERR("new nonzero location");
    } }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Drivers: one template, as in the reference (15 near-identical copies) */
/* ------------------------------------------------------------------ */
enum { OP_SYSTEM, OP_MATRIX, OP_VECTOR, OP_FUNCTION, OP_JACOBIAN, OP_IFUNCTION, OP_IJACOBIAN };

static int run(OrcIGA *iga,int op,void *fn,void *ctx,double a,const double *Vg,double t,const double *Ug,OrcMat *A,double *B)
{
  Elem *e; int rc, ierr = 0;
  size_t nv, nm;
  int wantU = (op==OP_FUNCTION||op==OP_JACOBIAN||op==OP_IFUNCTION||op==OP_IJACOBIAN);
  int wantV = (op==OP_IFUNCTION||op==OP_IJACOBIAN);
  int hasM  = (op==OP_SYSTEM||op==OP_MATRIX||op==OP_JACOBIAN||op==OP_IJACOBIAN);
  int hasV  = (op==OP_SYSTEM||op==OP_VECTOR||op==OP_FUNCTION||op==OP_IFUNCTION);
  if (!iga->setup) ORC_ERR("setup first");
  if (iga->rational && !iga->rationalW) ORC_ERR("no geometry set");
  e = elem_create(iga);
  nv = (size_t)e->nen*e->dof; nm = nv*nv;
  /* src/petigaksp.c:166-167: zero the outputs */
  if (hasM) orc_mat_zero(A);
  if (hasV) memset(B,0,sizeof(double)*(size_t)orc_global_size(iga));
  while ((rc = elem_next(e)) > 0) {
    double *Ke = e->wmat[0], *Fe = e->wvec[0], *U = e->wval[0], *V = e->wval[1];
    if (hasM) memset(Ke,0,sizeof(double)*nm);
    if (hasV) memset(Fe,0,sizeof(double)*nv);
    if (wantV) { elem_get_values(e,Vg,V); }
    if (wantU) { elem_get_values(e,Ug,U); }
    if (wantV) elem_del_values(e,V);
    if (wantU) elem_fix_values(e,U);
    while (elem_next_form(e)) {
      if (elem_begin_point(e)) { ierr = 1; goto done; }
      while (elem_next_point(e)) {
        OrcPoint *p = &e->point;
        double *Kq = e->pmat, *Fq = e->pvec;
        if (hasM) memset(Kq,0,sizeof(double)*nm);
        if (hasV) memset(Fq,0,sizeof(double)*nv);
        switch (op) {
        case OP_SYSTEM:    ierr = ((OrcFormSystem)fn)(p,Kq,Fq,ctx); break;
        case OP_MATRIX:    ierr = ((OrcFormMatrix)fn)(p,Kq,ctx); break;
        case OP_VECTOR:    ierr = ((OrcFormVector)fn)(p,Fq,ctx); break;
        case OP_FUNCTION:  ierr = ((OrcFormFunction)fn)(p,U,Fq,ctx); break;
        case OP_JACOBIAN:  ierr = ((OrcFormJacobian)fn)(p,U,Kq,ctx); break;
        case OP_IFUNCTION: ierr = ((OrcFormIFunction)fn)(p,a,V,t,U,Fq,ctx); break;
        case OP_IJACOBIAN: ierr = ((OrcFormIJacobian)fn)(p,a,V,t,U,Kq,ctx); break;
        }
        if (ierr) goto done;
        if (hasM) point_add(p,nm,Kq,Ke);
        if (hasV) point_add(p,nv,Fq,Fe);
      }
    }
    switch (op) {
    case OP_SYSTEM: elem_fix_system(e,Ke,Fe); break;
    case OP_FUNCTION: case OP_IFUNCTION: elem_fix_function(e,Fe); break;
    case OP_JACOBIAN: case OP_IJACOBIAN: elem_fix_jacobian(e,Ke); break;
    default: break;   /* Matrix / Vector: no BC fix-up (src/petigaksp.c:33-125) */
    }
    if (hasM && elem_assemble_mat(e,Ke,A)) { ierr = 1; goto done; }
    if (hasV) elem_assemble_vec(e,Fe,B);
  }
  if (rc < 0) ierr = 1;
done:
  elem_destroy(e);
  return ierr;
}

int orc_compute_system(OrcIGA *iga,OrcFormSystem f,void *ctx,OrcMat *A,double *B) { return run(iga,OP_SYSTEM,(void*)f,ctx,0,NULL,0,NULL,A,B); }
int orc_compute_matrix(OrcIGA *iga,OrcFormMatrix f,void *ctx,OrcMat *A) { return run(iga,OP_MATRIX,(void*)f,ctx,0,NULL,0,NULL,A,NULL); }
int orc_compute_vector(OrcIGA *iga,OrcFormVector f,void *ctx,double *B) { return run(iga,OP_VECTOR,(void*)f,ctx,0,NULL,0,NULL,NULL,B); }
int orc_compute_function(OrcIGA *iga,OrcFormFunction f,void *ctx,const double *U,double *F) { return run(iga,OP_FUNCTION,(void*)f,ctx,0,NULL,0,U,NULL,F); }
int orc_compute_jacobian(OrcIGA *iga,OrcFormJacobian f,void *ctx,const double *U,OrcMat *J) { return run(iga,OP_JACOBIAN,(void*)f,ctx,0,NULL,0,U,J,NULL); }
int orc_compute_ifunction(OrcIGA *iga,OrcFormIFunction f,void *ctx,double a,const double *V,double t,const double *U,double *F) { return run(iga,OP_IFUNCTION,(void*)f,ctx,a,V,t,U,NULL,F); }
int orc_compute_ijacobian(OrcIGA *iga,OrcFormIJacobian f,void *ctx,double a,const double *V,double t,const double *U,OrcMat *J) { return run(iga,OP_IJACOBIAN,(void*)f,ctx,a,V,t,U,J,NULL); }

/* src/petigacomp.c:35-98 (IGAComputeScalar); full!=0 also walks the boundary-form passes like
 * test/IGAGeometryMap.c:391-450 (IGAComputeScalarFull) */
int orc_compute_scalar(OrcIGA *iga,const double *Ug,int n,double *S,OrcFormScalar fn,void *ctx,int full)
{
  Elem *e; int rc, ierr=0, i; double *work;
  if (!iga->setup) ORC_ERR("setup first");
  e = elem_create(iga);
  work = (double*)xcalloc((size_t)n,sizeof(double));
  for (i=0; i<n; i++) S[i] = 0;
  while ((rc = elem_next(e)) > 0) {
    double *U = e->wval[0];
    elem_get_values(e,Ug,U);
    if (!full) { e->boundary_id = 2*e->dim - 1; } /* only the interior pass */
    while (elem_next_form(e)) {
      if (elem_begin_point(e)) { ierr = 1; goto done; }
      while (elem_next_point(e)) {
        memset(work,0,sizeof(double)*(size_t)n);
        if ((ierr = fn(&e->point,U,n,work,ctx))) goto done;
        point_add(&e->point,(size_t)n,work,S);
      }
    }
  }
  if (rc < 0) ierr = 1;
done:
  free(work); elem_destroy(e);
  return ierr;
}

/* Tabulate a single element for inspection.  The returned pointers stay valid until the next
 * call (one static iterator, like iga->iterator in the reference). */
int orc_element_tabulate(OrcIGA *iga,const int ID[3],int boundary_id,OrcElemView *out)
{
  static Elem *e = NULL;
  int k;
  if (e) { elem_destroy(e); e = NULL; }
  if (!iga->setup) ORC_ERR("setup first");
  e = elem_create(iga);
  e->index = 0;
  for (k=0; k<3; k++) e->ID[k] = (k < e->dim) ? ID[k] : 0;
  elem_closure(e);
  if (elem_buildfix(e)) return 1;
  if (boundary_id >= 0 && boundary_id < 2*e->dim) { e->atboundary = 1; e->boundary_id = boundary_id; }
  else { e->atboundary = 0; e->boundary_id = 2*e->dim; }
  if (elem_begin_point(e)) return 1;
  out->nqp = e->point.count; out->nen = e->nen; out->dim = e->dim; out->nsd = e->nsd;
  out->weight=e->weight; out->detJac=e->detJac; out->point=e->mapU[0]; out->normal=e->normal; out->detX=e->detX; out->detS=e->detS;
  for (k=0; k<5; k++) {
    out->basis[k]=e->basis[k];
    out->shape[k]=(iga->nsd && e->dim==e->nsd)?e->shape[k]:e->basis[k];
    out->mapU[k]=e->mapU[k];
    out->mapX[k]=(iga->nsd)?e->mapX[k]:e->mapU[k];
  }
  out->mapping = e->mapping; out->geometryX = e->geometryX; out->rationalW = e->rationalW;
  return 0;
}

/* ------------------------------------------------------------------ */
/* Field interpolation at a point: src/petigaval.F90:182-251 and       */
/* src/petigapoint.c:214-294                                           */
/* ------------------------------------------------------------------ */
static void eval(int nen,int dof,int nc,const double *N,const double *U,double *V)
{
  int a,i,f;
  for (i=0; i<dof*nc; i++) V[i] = 0;
  for (a=0; a<nen; a++) for (i=0; i<dof; i++) { double u = U[a*dof+i]; for (f=0; f<nc; f++) V[i*nc+f] += N[a*nc+f]*u; }
}
void orc_point_value(const OrcPoint *p,const double *U,double *u) { eval(p->nen,p->dof,1,p->shape[0],U,u); }
void orc_point_grad (const OrcPoint *p,const double *U,double *u) { eval(p->nen,p->dof,p->dim,p->shape[1],U,u); }
void orc_point_hess (const OrcPoint *p,const double *U,double *u) { eval(p->nen,p->dof,p->dim*p->dim,p->shape[2],U,u); }
void orc_point_der3 (const OrcPoint *p,const double *U,double *u) { eval(p->nen,p->dof,p->dim*p->dim*p->dim,p->shape[3],U,u); }
void orc_point_del2 (const OrcPoint *p,const double *U,double *u)
{
  int a,c,i,dim=p->dim,dof=p->dof,d2=dim*dim;
  for (c=0; c<dof; c++) u[c] = 0;
  for (a=0; a<p->nen; a++) for (c=0; c<dof; c++) for (i=0; i<dim; i++) u[c] += p->shape[2][a*d2+i*(dim+1)]*U[a*dof+c];
}
void orc_point_geommap(const OrcPoint *p,double *x)
{
  int i; const double *X = p->geometry ? p->mapX[0] : p->mapU[0];
  for (i=0; i<(p->geometry?p->nsd:p->dim); i++) x[i] = X[i];
}
/* src/petigapoint.c:269-294 (IGAPointFormInvGradGeomMap): rows scaled by 1/L_a, L = element half-lengths */
void orc_point_invgradgeommap(const OrcPoint *p,double *G)
{
  int a,i,dim=p->dim,nsd=p->nsd; double L[3]={1,1,1};
  for (i=0; i<dim; i++) L[i] = p->iga->basis[i].detJac[p->ID[i]];
  if (p->geometry && dim != nsd) {   /* IGA_GetInvGradGeomMap, src/petigaval.F90:124-142: G = ((F^T F)^-1 F^T)^T, F = mapX[1] [nsd][dim] */
    const double *F = p->mapX[1]; double M[9]={0,0,0,0,0,0,0,0,0},Mi[9]; int b;
    for (a=0;a<dim;a++) for (b=0;b<dim;b++) { double t=0; for (i=0;i<nsd;i++) t += F[i*dim+a]*F[i*dim+b]; M[a+dim*b] = t; }
    k_inv(dim,k_det(dim,M),M,Mi);
    for (a=0;a<dim;a++) for (i=0;i<nsd;i++) { double t=0; for (b=0;b<dim;b++) t += Mi[a+dim*b]*F[i*dim+b]; G[a*nsd+i] = t/L[a]; }
  }
  else if (p->geometry) { memcpy(G,p->mapU[1],sizeof(double)*(size_t)dim*nsd); for (a=0;a<dim;a++) for (i=0;i<nsd;i++) G[a*nsd+i] /= L[a]; }
  else { memset(G,0,sizeof(double)*(size_t)dim*dim); for (i=0;i<dim;i++) G[i*(dim+1)] = 1/L[i]; }
}
