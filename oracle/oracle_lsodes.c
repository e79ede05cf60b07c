/* oracle_lsodes.c -- variable-order BDF integrator, ODEPACK-DLSODES-faithful.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, for the one configuration rac-2d uses (MF=21: BDF, user sparsity, analytic Jacobian;
 * ITOL=4, ITASK=4, IOPT=1 with MAXORD=5, MXSTEP=mxstep_per_interval, MXHNIL=1, H0=0, HMAX=TCRIT=t_max,
 * HMIN=0; src/chemistry.f90:190-201,425-426,1955-1961):
 *   DLSODES driver  src/opkdmain.f:3069-3588   (blocks A-H)         -> lsodes_call
 *   DSTODE          src/opkda1.f:630-1126      (one BDF step)       -> stode
 *   DCFODE          src/opkda1.f:146-171       (BDF coefficients)   -> cfode_bdf
 *   DPRJS           src/opkda1.f:1735-1838     (P = I - h*el0*J, reuse/rescale rule, LU) -> prjs
 *   DSOLSS          src/opkda1.f:1919-1924     (solve)              -> via orc_lu_solve
 *   DINTDY          src/opkda1.f:236-263       (k = 0 interpolation) -> intdy0
 *   DEWSET/DVNORM   src/opkda1.f:1170-1171, 1203-1206
 * The control flow is re-expressed with loops and small functions; every numerical expression keeps
 * the reference's operand order so that the trajectory is the same up to the linear solver's rounding.
 */
#include "oracle_lsodes.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double vnorm(int n, const double *v, const double *w) {
  double sum = 0.0;
  for (int i = 0; i < n; i++) { double q = v[i] * w[i]; sum = sum + q * q; }
  return sqrt(sum / n);
}

static void cfode_bdf(orc_lsodes *s) { /* METH = 2 branch */
  double pc[13]; pc[1] = 1.0; double rq1fac = 1.0;
  for (int nq = 1; nq <= 5; nq++) {
    double fnq = nq; int nqp1 = nq + 1;
    pc[nqp1] = 0.0;
    for (int ib = 1; ib <= nq; ib++) { int i = nq + 2 - ib; pc[i] = pc[i - 1] + fnq * pc[i]; }
    pc[1] = fnq * pc[1];
    for (int i = 1; i <= nqp1; i++) s->elco[nq][i] = pc[i] / pc[2];
    s->elco[nq][2] = 1.0;
    s->tesco[nq][1] = rq1fac;
    s->tesco[nq][2] = nqp1 / s->elco[nq][1];
    s->tesco[nq][3] = (nq + 2) / s->elco[nq][1];
    rq1fac = rq1fac / fnq;
  }
}

static void set_order_coeffs(orc_lsodes *s) { /* DSTODE label 150 */
  for (int i = 1; i <= s->l; i++) s->el[i] = s->elco[s->nq][i];
  s->rc = s->rc * s->el[1] / s->el0;
  s->el0 = s->el[1];
  s->conit = 0.5 / (s->nq + 2);
}

static void ewset(orc_lsodes *s, const double *ycur) {
  for (int i = 0; i < s->n; i++) s->ewt[i] = s->rtol[i] * fabs(ycur[i]) + s->atol[i];
}

/* DPRJS, MITER = 1.  y = predicted values. */
static void prjs(orc_lsodes *s, const double *y) {
  const orc_symbolic *S = s->S; const int n = s->n;
  double hl0 = s->h * s->el0, con = -hl0;
  int jok = 1;
  if (s->nst == 0 || s->nst >= s->nslj + s->msbj) jok = 0;
  if (s->icf == 1 && fabs(s->rc - 1.0) < s->ccmxj) jok = 0;
  if (s->icf == 2) jok = 0;
  if (jok == 1) {
    s->jcur = 0;
    double rcon = con / s->con0, rcont = fabs(con) / s->conmin;
    if (rcont > s->rbig && s->iplost == 1) jok = 0;
    else {
      for (int j = 0; j < n; j++)
        for (int k = S->IAN[j] - 1; k < S->IAN[j + 1] - 1; k++) {
          int i = S->JAN[k] - 1; double pij = s->P[k];
          if (i == j) {
            pij = pij - 1.0;
            if (fabs(pij) < s->psmall) { s->iplost = 1; s->conmin = fmin(fabs(s->con0), s->conmin); }
          }
          pij = pij * rcon;
          if (i == j) pij = pij + 1.0;
          s->P[k] = pij;
        }
    }
  }
  if (jok == 0) {
    s->jcur = 1; s->nje++; s->nslj = s->nst; s->iplost = 0; s->conmin = fabs(con);
    s->jac(s->ctx, y, s->P); /* J on the IAN/JAN pattern */
    for (int j = 0; j < n; j++)
      for (int k = S->IAN[j] - 1; k < S->IAN[j + 1] - 1; k++) {
        s->P[k] = s->P[k] * con;
        if (S->JAN[k] - 1 == j) s->P[k] = s->P[k] + 1.0;
      }
  }
  s->nlu++; s->con0 = con; s->ierpj = 0;
  if (orc_numeric_lu(S, s->P, s->L, s->U, s->Dinv, s->w) != 0) s->ierpj = 1; /* zero pivot */
}

static void retract(orc_lsodes *s) { /* inverse Pascal, DSTODE :956-962 */
  const int n = s->n;
  for (int jb = 1; jb <= s->nq; jb++)
    for (int j = s->nq - jb; j < s->nq; j++)
      for (int i = 0; i < n; i++) s->yh[j * n + i] = s->yh[j * n + i] - s->yh[(j + 1) * n + i];
}

/* rescale by rh (labels 170/175); returns nothing, caller routes on iredo */
static void rescale(orc_lsodes *s, double rh, int apply_hmin) {
  const int n = s->n;
  if (apply_hmin) rh = fmax(rh, s->hmin / fabs(s->h));
  rh = fmin(rh, s->rmax);
  rh = rh / fmax(1.0, fabs(s->h) * s->hmxi * rh);
  double r = 1.0;
  for (int j = 2; j <= s->l; j++) {
    r = r * rh;
    for (int i = 0; i < n; i++) s->yh[(j - 1) * n + i] = s->yh[(j - 1) * n + i] * r;
  }
  s->h = s->h * rh; s->rc = s->rc * rh; s->ialth = s->l;
}

static int stode(orc_lsodes *s, double *y) {
  const int n = s->n; double *yh = s->yh, *ewt = s->ewt, *savf = s->savf, *acor = s->acor;
  double told = s->tn, delp = 0.0, del = 0.0, dsm = 0.0, rh = 0.0;
  int ncf = 0, m = 0, iredo = 0;
  s->kflag = 0; s->ierpj = 0; s->iersl = 0; s->jcur = 0; s->icf = 0;

  if (s->jstart == 0) {
    s->lmax = s->maxord + 1; s->nq = 1; s->l = 2; s->ialth = 2; s->rmax = 10000.0; s->rc = 0.0;
    s->el0 = 1.0; s->crate = 0.7; s->hold = s->h; s->nslp = 0; s->ipup = 1;
    cfode_bdf(s); set_order_coeffs(s);
  } else if (s->jstart < 0) {
    if (s->jstart == -1) { s->ipup = 1; s->lmax = s->maxord + 1; if (s->ialth == 1) s->ialth = 2; }
    if (s->h != s->hold) { /* label 160 */
      rh = s->h / s->hold; s->h = s->hold; iredo = 3;
      rescale(s, rh, 0);
    }
  }

  for (;;) { /* ---- label 200: one attempt at the step ---- */
    if (fabs(s->rc - 1.0) > s->ccmax) s->ipup = 1;
    if (s->nst >= s->nslp + s->msbp) s->ipup = 1;
    s->tn = s->tn + s->h;
    for (int jb = 1; jb <= s->nq; jb++)
      for (int j = s->nq - jb; j < s->nq; j++)
        for (int i = 0; i < n; i++) yh[j * n + i] = yh[j * n + i] + yh[(j + 1) * n + i];

    int converged = 0, fatal = 0;
    for (;;) { /* ---- label 220: corrector with current or refreshed P ---- */
      m = 0;
      for (int i = 0; i < n; i++) y[i] = yh[i];
      s->f(s->ctx, y, savf); s->nfe++;
      if (s->ipup > 0) {
        prjs(s, y);
        s->ipup = 0; s->rc = 1.0; s->nslp = s->nst; s->crate = 0.7;
        if (s->ierpj != 0) break; /* -> 430 */
      }
      for (int i = 0; i < n; i++) acor[i] = 0.0;
      int fail410 = 0;
      for (;;) { /* label 270/350 */
        for (int i = 0; i < n; i++) y[i] = s->h * savf[i] - (yh[n + i] + acor[i]);
        orc_lu_solve(s->S, s->L, s->U, s->Dinv, y, s->z);
        del = vnorm(n, y, ewt);
        for (int i = 0; i < n; i++) { acor[i] = acor[i] + y[i]; y[i] = yh[i] + s->el[1] * acor[i]; }
        if (m != 0) s->crate = fmax(0.2 * s->crate, del / delp);
        double dcon = del * fmin(1.0, 1.5 * s->crate) / (s->tesco[s->nq][2] * s->conit);
        if (dcon <= 1.0) { converged = 1; break; }
        m++;
        if (m == s->maxcor) { fail410 = 1; break; }
        if (m >= 2 && del > 2.0 * delp) { fail410 = 1; break; }
        delp = del;
        s->f(s->ctx, y, savf); s->nfe++;
      }
      if (converged) break;
      if (fail410 && s->jcur != 1) { s->icf = 1; s->ipup = 1; continue; } /* label 410 -> 220 */
      break; /* -> 430 */
    }

    if (!converged) { /* ---- label 430 ---- */
      s->icf = 2; ncf++; s->rmax = 2.0; s->tn = told;
      retract(s);
      if (s->ierpj < 0 || s->iersl < 0) { s->kflag = -3; fatal = 1; }
      else if (fabs(s->h) <= s->hmin * 1.00001) { s->kflag = -2; fatal = 1; }
      else if (ncf == s->mxncf) { s->kflag = -2; fatal = 1; }
      if (fatal) break;
      rh = 0.25; s->ipup = 1; iredo = 1;
      rescale(s, rh, 1);
      continue;
    }

    /* ---- label 450: corrector converged, local error test ---- */
    s->jcur = 0;
    if (m == 0) dsm = del / s->tesco[s->nq][2];
    else dsm = vnorm(n, acor, ewt) / s->tesco[s->nq][2];

    int consider = 0; /* 1 -> compute rhdn/rhsm/rhup and choose */
    double rhup = 0.0;
    if (dsm > 1.0) { /* label 500 */
      s->kflag = s->kflag - 1; s->tn = told;
      retract(s);
      s->rmax = 2.0;
      if (fabs(s->h) <= s->hmin * 1.00001) { s->kflag = -1; break; }
      if (s->kflag <= -3) { /* label 640 */
        if (s->kflag == -10) { s->kflag = -1; break; }
        rh = 0.1; rh = fmax(s->hmin / fabs(s->h), rh);
        s->h = s->h * rh;
        for (int i = 0; i < n; i++) y[i] = yh[i];
        s->f(s->ctx, y, savf); s->nfe++;
        for (int i = 0; i < n; i++) yh[n + i] = s->h * savf[i];
        s->ipup = 1; s->ialth = 5;
        if (s->nq != 1) { s->nq = 1; s->l = 2; set_order_coeffs(s); }
        continue;
      }
      iredo = 2; rhup = 0.0; consider = 1;
    } else { /* success */
      s->kflag = 0; iredo = 0; s->nst++; s->hu = s->h; s->nqu = s->nq;
      for (int j = 1; j <= s->l; j++)
        for (int i = 0; i < n; i++) yh[(j - 1) * n + i] = yh[(j - 1) * n + i] + s->el[j] * acor[i];
      s->ialth--;
      if (s->ialth == 0) { /* label 520 */
        rhup = 0.0;
        if (s->l != s->lmax) {
          for (int i = 0; i < n; i++) savf[i] = acor[i] - yh[(s->lmax - 1) * n + i];
          double dup = vnorm(n, savf, ewt) / s->tesco[s->nq][3];
          double exup = 1.0 / (s->l + 1);
          rhup = 1.0 / (1.4 * pow(dup, exup) + 0.0000014);
        }
        consider = 1;
      } else {
        if (s->ialth <= 1 && s->l != s->lmax)
          for (int i = 0; i < n; i++) yh[(s->lmax - 1) * n + i] = acor[i];
        goto done700;
      }
    }

    if (consider) { /* labels 540-630 */
      double exsm = 1.0 / s->l;
      double rhsm = 1.0 / (1.2 * pow(dsm, exsm) + 0.0000012);
      double rhdn = 0.0;
      if (s->nq != 1) {
        double ddn = vnorm(n, yh + (s->l - 1) * n, ewt) / s->tesco[s->nq][1];
        double exdn = 1.0 / s->nq;
        rhdn = 1.0 / (1.3 * pow(ddn, exdn) + 0.0000013);
      }
      int newq; int sel; /* 0: same order, 1: down, 2: up */
      if (rhsm >= rhup) sel = (rhsm < rhdn) ? 1 : 0;
      else sel = (rhup > rhdn) ? 2 : 1;
      if (sel == 2) { /* label 590 */
        newq = s->l; rh = rhup;
        if (rh < 1.1) { s->ialth = 3; goto done700; }
        double r = s->el[s->l] / s->l;
        for (int i = 0; i < n; i++) yh[newq * n + i] = acor[i] * r;
      } else {
        if (sel == 0) { newq = s->nq; rh = rhsm; }
        else { newq = s->nq - 1; rh = rhdn; if (s->kflag < 0 && rh > 1.0) rh = 1.0; }
        if (s->kflag == 0 && rh < 1.1) { s->ialth = 3; goto done700; } /* label 610 */
        if (s->kflag <= -2) rh = fmin(rh, 0.2);
      }
      if (newq != s->nq) { s->nq = newq; s->l = s->nq + 1; set_order_coeffs(s); } /* label 630 -> 150 */
      rescale(s, rh, 1); /* label 170 */
      if (iredo == 0) { s->rmax = 10.0; goto done700; } /* label 690 */
      continue; /* redo the step */
    }
  }
  /* failure exits (labels 660/670/680 -> 720) */
  s->hold = s->h; s->jstart = 1;
  return s->kflag;

done700: {
    double r = 1.0 / s->tesco[s->nqu][2];
    for (int i = 0; i < n; i++) acor[i] = acor[i] * r;
  }
  s->hold = s->h; s->jstart = 1;
  return s->kflag;
}

static void intdy0(orc_lsodes *s, double t, double *dky) { /* DINTDY with K = 0 */
  const int n = s->n; double sfac = (t - s->tn) / s->h;
  for (int i = 0; i < n; i++) dky[i] = s->yh[(s->l - 1) * n + i];
  for (int j = s->nq - 1; j >= 0; j--)
    for (int i = 0; i < n; i++) dky[i] = s->yh[j * n + i] + sfac * dky[i];
}

orc_lsodes *orc_lsodes_create(int n, const orc_symbolic *S, orc_f_fn f, orc_jac_fn jac, void *ctx) {
  orc_lsodes *s = calloc(1, sizeof *s);
  s->n = n; s->S = S; s->f = f; s->jac = jac; s->ctx = ctx;
  s->yh = calloc((size_t)6 * n, sizeof(double));
  s->ewt = calloc((size_t)n, sizeof(double)); s->savf = calloc((size_t)n, sizeof(double));
  s->acor = calloc((size_t)n, sizeof(double)); s->w = calloc((size_t)n, sizeof(double)); s->z = calloc((size_t)n, sizeof(double));
  s->P = calloc((size_t)S->nnz, sizeof(double));
  s->L = calloc((size_t)S->nzl + 1, sizeof(double)); s->U = calloc((size_t)S->nzu + 1, sizeof(double));
  s->Dinv = calloc((size_t)n, sizeof(double));
  return s;
}
void orc_lsodes_free(orc_lsodes *s) {
  if (!s) return;
  free(s->yh); free(s->ewt); free(s->savf); free(s->acor); free(s->w); free(s->z); free(s->P); free(s->L);
  free(s->U); free(s->Dinv); free(s);
}

static void finish(orc_lsodes *s, double *y, double *t) { /* label 580 */
  for (int i = 0; i < s->n; i++) y[i] = s->yh[i];
  *t = s->tn;
}

/* One DLSODES call with ITASK = 4.  istate in: 1/2/3, out: 2 or negative (ODEPACK codes). */
void orc_lsodes_call(orc_lsodes *s, double *y, double *t, double tout, int *istate) {
  const int n = s->n; const double u = 2.220446049250313e-16; /* DUMACH() */
  int ihit = 0;
  if (*istate < 1 || *istate > 3) { *istate = -3; return; }
  if (*istate != 1 && s->init == 0) { *istate = -3; return; }
  if (*istate == 1) { s->init = 0; if (tout == *t) return; }
  if (*istate == 1 || *istate == 3) { /* Block B */
    s->maxord = 5; if (s->mxstep <= 0) s->mxstep = 500;
    if (*istate == 1) s->h0 = 0.0; /* RWORK(5) = 0 */
    if (s->hmax < 0.0) { *istate = -3; return; }
    s->hmxi = 0.0; if (s->hmax > 0.0) s->hmxi = 1.0 / s->hmax;
    s->hmin = 0.0;
    for (int i = 0; i < n; i++) if (s->rtol[i] < 0.0 || s->atol[i] < 0.0) { *istate = -3; return; }
    if (*istate == 3) {
      /* DIPREP/DPREP rerun: same ordering, same symbolic LU.  DPREP zeroes NNZ words at a TEMPORARY location
       * at the far end of the work array (IPA = LENWK+1-NNZ, src/opkda1.f:1487-1494) and then moves IPA back to
       * LREQ+1-NNZ (:1511), where the previous P still sits because the layout is the same as before; so the
       * saved P survives an ISTATE=3 call and DPRJS may legitimately rescale it (JOK = 1). */
      s->jstart = -1;
      /* ... except where the temporary area overlaps it: the zeroed words are RWORK(LRW - NCOLM*N - NNZ + 1 .. LRW - NCOLM*N)
       * (LENWK = LRW - 20 - NCOLM*N at this point, src/opkdmain.f:3185-3196, NCOLM = min(NQ+1, MAXORD+2)), the saved P is
       * RWORK(20 + LREQ - NNZ + 1 .. 20 + LREQ) with LREQ = LENRW - 20 - 9 N; with the LRW the reference allocates
       * (20 + 4 NNZ0 + 28 N, src/chemistry.f90:1945) the last Z = 20 + LREQ - (LRW - NCOLM*N - NNZ) entries of P are zeroed.
       * Checked against the reference: with it the output times of an mxstep = 6 run agree to 7 digits over the first
       * intervals, without it they are off by factors from the first ISTATE = 3 call on (tests/golden/policy_grain.npz). */
      if (s->lenrw_ref > 0) {
        const long nnz = s->S->nnz, ncolm = (s->nq + 1 < 7) ? s->nq + 1 : 7;
        long Z = 20 + (s->lenrw_ref - 20 - 9L * n) - (s->lrw_ref - ncolm * n - nnz);
        if (Z > nnz) Z = nnz;
        for (long k = nnz - Z; k < nnz; k++) if (k >= 0) s->P[k] = 0.0;
      }
    }
  }
  if (*istate == 1) { /* Block C */
    s->uround = u; s->tn = *t; s->nst = 0; s->h = 1.0;
    for (int i = 0; i < n; i++) s->yh[i] = y[i];
    s->f(s->ctx, y, s->yh + n); s->nfe = 1;
    ewset(s, s->yh);
    for (int i = 0; i < n; i++) { if (s->ewt[i] <= 0.0) { *istate = -3; return; } s->ewt[i] = 1.0 / s->ewt[i]; }
    memset(s->P, 0, (size_t)s->S->nnz * sizeof(double));
    if ((s->tcrit - tout) * (tout - *t) < 0.0) { *istate = -3; return; }
    s->jstart = 0; s->msbj = 50; s->nslj = 0; s->ccmxj = 0.2; s->psmall = 1000.0 * u; s->rbig = 0.01 / s->psmall;
    s->nhnil = 0; s->nje = 0; s->nlu = 0; s->nslast = 0; s->hu = 0.0; s->nqu = 0; s->ccmax = 0.3;
    s->maxcor = 3; s->msbp = 20; s->mxncf = 10;
    if (s->h0 == 0.0) {
      double tdist = fabs(tout - *t), w0 = fmax(fabs(*t), fabs(tout));
      if (tdist < 2.0 * u * w0) { *istate = -3; return; }
      double tol = s->rtol[0];
      for (int i = 0; i < n; i++) tol = fmax(tol, s->rtol[i]);
      if (tol <= 0.0) {
        for (int i = 0; i < n; i++) { double ayi = fabs(y[i]); if (ayi != 0.0) tol = fmax(tol, s->atol[i] / ayi); }
      }
      tol = fmax(tol, 100.0 * u); tol = fmin(tol, 0.001);
      double sum = vnorm(n, s->yh + n, s->ewt);
      sum = 1.0 / (tol * w0 * w0) + tol * sum * sum;
      s->h0 = 1.0 / sqrt(sum);
      s->h0 = fmin(s->h0, tdist);
      s->h0 = copysign(s->h0, tout - *t);
    }
    double rh = fabs(s->h0) * s->hmxi;
    if (rh > 1.0) s->h0 = s->h0 / rh;
    s->h = s->h0;
    for (int i = 0; i < n; i++) s->yh[n + i] = s->h0 * s->yh[n + i];
  } else { /* Block D, ITASK = 4 */
    s->nslast = s->nst;
    if ((s->tn - s->tcrit) * s->h > 0.0) { *istate = -3; return; }
    if ((s->tcrit - tout) * s->h < 0.0) { *istate = -3; return; }
    if ((s->tn - tout) * s->h >= 0.0) { intdy0(s, tout, y); *t = tout; *istate = 2; return; }
    double hmx = fabs(s->tn) + fabs(s->h);
    ihit = fabs(s->tn - s->tcrit) <= 100.0 * u * hmx;
    if (ihit) { finish(s, y, t); *t = s->tcrit; *istate = 2; return; }
    double tnext = s->tn + s->h * (1.0 + 4.0 * u);
    if ((tnext - s->tcrit) * s->h > 0.0) {
      s->h = (s->tcrit - s->tn) * (1.0 - 4.0 * u);
      if (*istate == 2) s->jstart = -2;
    }
  }
  int first = (*istate == 1);
  for (;;) { /* Block E */
    if (!first) {
      if (s->nst - s->nslast >= s->mxstep) { *istate = -1; finish(s, y, t); return; }
      ewset(s, s->yh);
      for (int i = 0; i < n; i++) { if (s->ewt[i] <= 0.0) { *istate = -6; finish(s, y, t); return; } s->ewt[i] = 1.0 / s->ewt[i]; }
    }
    first = 0;
    double tolsf = u * vnorm(n, s->yh, s->ewt);
    if (tolsf > 1.0) {
      if (s->nst == 0) { *istate = -3; return; }
      *istate = -2; finish(s, y, t); return;
    }
    if (s->tn + s->h == s->tn) s->nhnil++;
    int kflag = stode(s, y);
    if (getenv("ORC_TRACE")) fprintf(stderr, "[oracle trace] tn=%.6e h=%.6e hu=%.6e nq=%d kflag=%d nst=%d nfe=%d nje/nlu=%d\n", s->tn, s->h, s->hu, s->nq, kflag, s->nst, s->nfe, s->nje * 10000 + s->nlu);
    if (kflag != 0) {
      if (kflag == -3) { *istate = -7; finish(s, y, t); return; }
      *istate = (kflag == -1) ? -4 : -5;
      double big = 0.0; s->imxer = 1;
      for (int i = 0; i < n; i++) { double size = fabs(s->acor[i] * s->ewt[i]); if (big < size) { big = size; s->imxer = i + 1; } }
      finish(s, y, t); return;
    }
    s->init = 1;
    if ((s->tn - tout) * s->h >= 0.0) { intdy0(s, tout, y); *t = tout; *istate = 2; return; }
    double hmx = fabs(s->tn) + fabs(s->h);
    ihit = fabs(s->tn - s->tcrit) <= 100.0 * u * hmx;
    if (ihit) { finish(s, y, t); *t = s->tcrit; *istate = 2; return; }
    double tnext = s->tn + s->h * (1.0 + 4.0 * u);
    if ((tnext - s->tcrit) * s->h > 0.0) { s->h = (s->tcrit - s->tn) * (1.0 - 4.0 * u); s->jstart = -2; }
  }
}
