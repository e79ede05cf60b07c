/* oracle_solve.c -- output-time loop, error policy and cell hand-off.  TEST INFRASTRUCTURE ONLY.
 * Restates src/chemistry.f90:391-588 chem_evol_solve and :272-387 ode_solver_error_handling for the
 * fixed-T mode (evolT = maySwitchT = .false., update_gH_params_realtime = .false.), plus the cell
 * initial condition of src/disk.f90:2055-2066 and the call order of src/disk.f90:1671-1686.
 * The reference's CPU-time guards (:438, :480-491) depend on wall-clock time; they are restated on MODELLED
 * time: per-call costs of the reference measured on one core (f 47 us, full Jacobian
 * 10.4 ms, ~1.0 ms of LU+solves per factorisation) times the call counters.  With the template's
 * max_runtime_allowed = 60 s they do not fire on any fixture cell.
 */
#include "oracle.h"
#include "oracle_lsodes.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const orc_network *net; const orc_params *p; const double *cell, *rates; const orc_symbolic *S; } cb_ctx;
static void cb_f(void *c, const double *y, double *ydot) { cb_ctx *x = c; orc_ode_f(x->net, x->p, x->cell, x->rates, y, ydot); }
static void cb_jac(void *c, const double *y, double *vals) {
  cb_ctx *x = c; orc_jac_on_pattern(x->net, x->p, x->cell, x->rates, y, x->S->IAN, x->S->JAN, vals);
}

/* The symbolic factorisation is cell independent: cache one per network (not thread safe by design). */
static const orc_network *g_net; static orc_symbolic *g_sym;
static const orc_symbolic *symbolic_for(const orc_network *net) {
  if (g_net != net) { orc_symbolic_free(g_sym); g_sym = orc_symbolic_build(net->NEQ, net->IA, net->JA); g_net = net; }
  return g_sym;
}

/* ode_solver_error_handling: only the tolerance loosening has an effect on the computation */
static void loosen(const orc_network *net, int istate, int imxer, double *rtol, double *atol) {
  if (istate != -4 && istate != -5) return;
  int idx = imxer - 1;
  if (imxer <= net->nS) { rtol[idx] = fmin(rtol[idx] * 10.0, 1e-3); atol[idx] = fmin(atol[idx] * 100.0, 1e-20); }
  else { rtol[idx] = fmin(rtol[idx] * 10.0, 1e-2); atol[idx] = fmin(atol[idx] * 100.0, 1.0); }
}

int orc_evol_solve(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
                   double *rtol, double *atol, double *y, double t0, double t_max,
                   double *t_final, int *quality, int *nerr_out, int *n_record_real,
                   double *record, double *touts, orc_stats *st) {
  /* p->dt_first_step is chemsol_params%dt_first_step of THIS run (the caller's continue rule has been applied to it) */
  const int NEQ = net->NEQ;
  const orc_symbolic *S = symbolic_for(net);
  cb_ctx ctx = {net, p, cell, rates, S};
  orc_lsodes *s = orc_lsodes_create(NEQ, S, cb_f, cb_jac, &ctx);
  s->rtol = rtol; s->atol = atol; s->tcrit = t_max; s->hmax = t_max; s->mxstep = p->mxstep_per_interval;
  s->lrw_ref = 20 + 4L * net->NNZ + 28L * NEQ; /* chem_prepare_solver_storage, src/chemistry.f90:1945 */
  {
    /* IWORK(17) as the reference's DLSODES reports it (ref_driver section '# workspace'); YSMP-dependent, known for the shipped networks */
    static const struct { int nS, nR; long lenrw; } known[] = {{464, 4767, 59430}, {467, 4801, 60324}, {484, 5830, 65578}, {524, 6425, 82134}};
    s->lenrw_ref = 0;
    for (unsigned k = 0; k < sizeof known / sizeof known[0]; k++) if (known[k].nS == net->nS && known[k].nR == net->nR) s->lenrw_ref = known[k].lenrw;
  }
  int n_record = orc_n_record(p, t0, t_max);
  int istate = 1, nerr = 0, nerr_c = 0, qual = 0, nrr = 1, ret = 0;
  double t = t0, t_step = p->dt_first_step, tout = t + t_step;
  orc_stats acc; memset(&acc, 0, sizeof acc);
  double rt_total = 0.0, rt_last = 1e300;
  const double rt_max = p->max_runtime_allowed, rt_per_step = 5.0 / (double)n_record * rt_max;
  if (touts) touts[0] = t;
  if (record) memcpy(record, y, (size_t)NEQ * sizeof(double));
  for (int i = 2; i <= n_record; i++) {
    if (tout >= t_max) tout = t_max;
    int was_restart = (istate == 1);
    long nst0 = was_restart ? 0 : s->nst, nfe0 = was_restart ? 0 : s->nfe, nje0 = was_restart ? 0 : s->nje, nlu0 = was_restart ? 0 : s->nlu;
    orc_lsodes_call(s, y, &t, tout, &istate);
    acc.nst += s->nst - nst0; acc.nfe += s->nfe - nfe0; acc.nje += s->nje - nje0; acc.nlu += s->nlu - nlu0;
    const double rt_this = p->rt_cost_f * (double)(s->nfe - nfe0) + p->rt_cost_jac * (double)(s->nje - nje0) + p->rt_cost_lu * (double)(s->nlu - nlu0);
    rt_total += rt_this;
    if (touts) touts[i - 1] = t;
    if (record) memcpy(record + (size_t)(i - 1) * NEQ, y, (size_t)NEQ * sizeof(double));
    nrr = i;
    if (rt_max > 0.0) {
      if (rt_this > fmax(10.0 * rt_last, 0.5 * rt_max) || rt_total > rt_max) break;
      if (rt_this > rt_per_step) istate = 1;
      rt_last = rt_this;
    }
    if (t >= t_max) break;
    if (istate < 0) {
      nerr++; nerr_c++;
      if (istate == -7) { ret = -7; break; } /* error_stop */
      loosen(net, istate, s->imxer, rtol, atol);
      if (istate == -3) { qual += 256; break; }
      if (nerr_c < 3) istate = 3; else { istate = 1; nerr_c = 0; }
    }
    {
      double yT = y[NEQ - 1]; int bad = isnan(yT) || yT <= 0.0;
      /* the reference indexes y(i_gH2) etc. even when the index is 0; only meaningful when present */
      if (net->i_gH2 > 0 && fabs(y[net->i_gH2 - 1]) > 1.0) bad = 1;
      if (net->i_gH2O > 0 && fabs(y[net->i_gH2O - 1]) > 1.0) bad = 1;
      if (net->i_gH > 0 && fabs(y[net->i_gH - 1]) > 1.0) bad = 1;
      if (net->idx10[1] > 0 && fabs(y[net->idx10[1] - 1]) > 2.0) bad = 1;
      if (net->idx10[2] > 0 && fabs(y[net->idx10[2] - 1]) > 1.0) bad = 1;
      if (bad) { qual += 512; break; }
    }
    if (p->steps_reset_solver > 0 && i % p->steps_reset_solver == 0) istate = 1;
    t_step = t_step * p->ratio_tstep;
    tout = t + t_step;
  }
  for (int i = nrr + 1; i <= n_record; i++) {
    if (touts) touts[i - 1] = t;
    if (record) memcpy(record + (size_t)(i - 1) * NEQ, y, (size_t)NEQ * sizeof(double));
  }
  if (nerr > (int)(0.1f * (float)n_record)) qual += 1;
  if (t <= 0.5 * t_max) qual += 2;
  if (t_final) *t_final = t;
  if (quality) *quality = qual;
  if (nerr_out) *nerr_out = nerr;
  if (n_record_real) *n_record_real = nrr;
  if (st) {
    *st = acc; st->nnz = S->nnz; st->nzl = S->nzl; st->nzu = S->nzu;
    st->nst_last = s->nst; st->nfe_last = s->nfe; st->nje_last = s->nje; st->nlu_last = s->nlu;
  }
  orc_lsodes_free(s);
  return ret;
}

int orc_solve_cell(const orc_network *net, const orc_params *p, const double *cell, const double *y0,
                   double *y_out, double *t_final, int *quality, int *nerr, orc_stats *stats) {
  const int NEQ = net->NEQ, nS = net->nS;
  double *rates = malloc((size_t)net->nR * sizeof(double));
  double *rtol = malloc((size_t)NEQ * sizeof(double)), *atol = malloc((size_t)NEQ * sizeof(double));
  memcpy(y_out, y0, (size_t)nS * sizeof(double));
  if (net->i_Grain0 > 0) y_out[net->i_Grain0 - 1] = cell[ORC_P_D2H];
  y_out[nS] = cell[ORC_P_TGAS];
  double t_max = cell[ORC_P_TMAX] > 0.0 ? cell[ORC_P_TMAX] : p->t_max;
  orc_set_tolerances(net, p, 1, cell[ORC_P_D2H], rtol, atol);
  int rc = orc_cal_rates(net, p, cell, rates, NULL);
  if (rc == 0) rc = orc_evol_solve(net, p, cell, rates, rtol, atol, y_out, 0.0, t_max, t_final, quality, nerr, NULL, NULL, NULL, stats);
  free(rates); free(rtol); free(atol);
  return rc;
}

void orc_rectify_abundances(const orc_network *net, double *y) { /* src/chemistry.f90:2192-2194 */
  double q = 0.0;
  for (int i = 0; i < net->nS; i++) q += y[i] * (double)net->elements[i * ORC_NELEM + 0];
  y[net->idx10[2] - 1] = y[net->idx10[2] - 1] + q;
}

int orc_calc_cell(const orc_network *net, const orc_params *p0, const double *cell, const double *y_init, int nlocal_iter,
                  double *abund_out, double *t_final_out, int *quality_out, orc_iter_info *info, orc_stats *stats) {
  const int NEQ = net->NEQ, nS = net->nS;
  double *rates = malloc((size_t)net->nR * sizeof(double));
  double *rtol = malloc((size_t)NEQ * sizeof(double)), *atol = malloc((size_t)NEQ * sizeof(double));
  double *y = malloc((size_t)NEQ * sizeof(double)), *abund = malloc((size_t)nS * sizeof(double));
  const double t_max = cell[ORC_P_TMAX] > 0.0 ? cell[ORC_P_TMAX] : p0->t_max;
  double t_final = 0.0;
  int qual_cell = 0, niter = 0, rc = 0;
  orc_stats acc; memset(&acc, 0, sizeof acc);
  memcpy(abund, y_init, (size_t)nS * sizeof(double));
  for (int j = 1; j <= nlocal_iter; j++) {
    orc_params p = *p0;
    const double t0 = (j > 1) ? t_final : 0.0;
    memcpy(y, abund, (size_t)nS * sizeof(double));
    y[nS] = cell[ORC_P_TGAS];
    if (j > 1) { /* set_initial_condition_4solver_continue, src/disk.f90:2121-2130 */
      orc_rectify_abundances(net, y);
      p.dt_first_step = fmax(p0->dt_first_step, t0 * 1e-3);
    }
    const int n_record = orc_n_record(&p, t0, t_max);
    double *record = malloc((size_t)n_record * NEQ * sizeof(double)), *touts = malloc((size_t)n_record * sizeof(double));
    orc_set_tolerances(net, &p, j, cell[ORC_P_D2H], rtol, atol);
    rc = orc_cal_rates(net, &p, cell, rates, NULL);
    double t_end = t0; int q = 0, nerr = 0, nrr = 1; orc_stats st;
    if (rc == 0) rc = orc_evol_solve(net, &p, cell, rates, rtol, atol, y, t0, t_max, &t_end, &q, &nerr, &nrr, record, touts, &st);
    if (rc != 0) { free(record); free(touts); break; }
    acc.nst += st.nst; acc.nfe += st.nfe; acc.nje += st.nje; acc.nlu += st.nlu; acc.nnz = st.nnz; acc.nzl = st.nzl; acc.nzu = st.nzu;
    niter = j;
    orc_iter_info *I = info ? &info[j - 1] : NULL;
    if (I) { I->t0 = t0; I->dt_first = p.dt_first_step; I->n_record = n_record; I->t_end = touts[nrr - 1]; I->nerr = nerr; }
    if (j > 1 && touts[nrr - 1] <= t_final) { /* "Local iteration does not proceed" (src/disk.f90:1706-1714) */
      if (I) { I->quality = qual_cell; I->isav = 0; I->t_final = t_final; I->proceeds = 0; I->n_mol_on_grain = NAN; }
      free(record); free(touts);
      break;
    }
    int isav;
    for (isav = nrr; isav >= 1; isav--) { /* src/disk.f90:1716-1721 */
      const double *r = record + (size_t)(isav - 1) * NEQ;
      if (!isnan(r[nS]) && !(net->idx10[0] > 0 && isnan(r[net->idx10[0] - 1]))) break;
    }
    qual_cell = q;
    double nmol = NAN;
    if (isav > 1) {
      memcpy(abund, record + (size_t)(isav - 1) * NEQ, (size_t)nS * sizeof(double));
      t_final = touts[isav - 1];
      nmol = 0.0; /* get_ice_coverage's side effect, src/chemistry.f90:995-1001 */
      for (int g = 0; g < net->nGrain; g++) nmol += abund[net->idxGrain[g] - 1];
      nmol = nmol / cell[ORC_P_D2H];
    }
    if (I) { I->quality = qual_cell; I->isav = isav; I->t_final = t_final; I->proceeds = 1; I->n_mol_on_grain = nmol; }
    free(record); free(touts);
    if (isav <= 1) break; /* "No useful data produced!" */
    if (qual_cell == 0 || t_final >= 0.5 * t_max) break;
  }
  memcpy(abund_out, abund, (size_t)nS * sizeof(double));
  if (t_final_out) *t_final_out = t_final;
  if (quality_out) *quality_out = qual_cell;
  if (stats) *stats = acc;
  free(rates); free(rtol); free(atol); free(y); free(abund);
  return rc != 0 ? rc : niter;
}
