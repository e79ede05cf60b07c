/* oracle_lsodes.h -- internal to oracle/ (TEST INFRASTRUCTURE ONLY). */
#ifndef RAC2D_ORACLE_LSODES_H
#define RAC2D_ORACLE_LSODES_H
#include "oracle_sparse.h"
typedef void (*orc_f_fn)(void *ctx, const double *y, double *ydot);
typedef void (*orc_jac_fn)(void *ctx, const double *y, double *vals /* on S->IAN/JAN */);
typedef struct {
  int n; const orc_symbolic *S; orc_f_fn f; orc_jac_fn jac; void *ctx;
  const double *rtol, *atol;         /* [n], owned by caller; may change between calls (ISTATE=3) */
  double tcrit, hmax; int mxstep;    /* RWORK(1), RWORK(6), IWORK(6) */
  long lrw_ref, lenrw_ref;           /* the reference's RWORK length and DLSODES' IWORK(17) (0: unknown, P survives ISTATE=3) */
  double *yh, *ewt, *savf, *acor, *P, *L, *U, *Dinv, *w, *z;
  /* COMMON /DLS001/ */
  double conit, crate, el[14], elco[6][14], hold, rmax, tesco[6][4], ccmax, el0, h, hmin, hmxi, hu, rc, tn, uround;
  int ialth, ipup, lmax, nslp, icf, ierpj, iersl, jcur, jstart, kflag, l, maxord, maxcor, msbp, mxncf, nq, nst, nfe, nje, nqu;
  /* COMMON /DLSS01/ */
  double con0, conmin, ccmxj, psmall, rbig; int iplost, msbj, nslj, nlu;
  /* driver SAVEd locals */
  int init, nhnil, nslast, imxer; double h0;
} orc_lsodes;
orc_lsodes *orc_lsodes_create(int n, const orc_symbolic *S, orc_f_fn f, orc_jac_fn jac, void *ctx);
void orc_lsodes_free(orc_lsodes *);
void orc_lsodes_call(orc_lsodes *, double *y, double *t, double tout, int *istate);
#endif
