/* oracle_rhs.c -- RHS and Jacobian callbacks, fixed-T branch.  TEST INFRASTRUCTURE ONLY.
 * Restates src/disk.f90:4569-4659 chem_ode_f and src/disk.f90:4746-4903 chem_ode_jac
 * (evolT = .false.: ydot(NEQ) = 0, pdj(NEQ) = 0, column NEQ = 0).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* flux of reaction i, or returns 0 for "cycle" (itype not handled) */
static int flux(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
                const double *y, int i, double *out, double *ydot) {
  const int *re = net->reac + 3 * i; double rtmp;
  switch (net->itype[i]) {
    case 5: case 6: case 21: case 64:
      rtmp = rates[i] * y[re[0] - 1] * y[re[1] - 1];
      if (y[re[0] - 1] < 0.0 && y[re[1] - 1] < 0.0) rtmp = -rtmp;
      break;
    case 1: case 2: case 3: case 13: case 61: case 20: rtmp = rates[i] * y[re[0] - 1]; break;
    case 62: case 75: {
      double tmp1 = cell[ORC_P_D2H] * cell[ORC_P_SITES];
      if (net->itype[i] == 75) tmp1 = tmp1 * net->ABC[3 * i + 2];
      if (tmp1 <= 0.0) rtmp = rates[i];
      else {
        double tmp = y[re[0] - 1] / tmp1;
        if (tmp <= 1e-4) rtmp = rates[i] * tmp; else rtmp = rates[i] * (1.0 - exp(-tmp));
      }
    } break;
    case 63:
      if (!strcmp(net->reac_name1[i], "gH") && p->H2_form_use_moeq) {
        int i1 = net->counterpart[re[0] - 1];
        rtmp = rates[i] * y[i1 - 1] * y[re[0] - 1];
        if (ydot) { ydot[i1 - 1] = ydot[i1 - 1] - rtmp; ydot[re[0] - 1] = ydot[re[0] - 1] + rtmp; }
      } else rtmp = rates[i] * y[re[0] - 1] * y[re[0] - 1];
      if (y[re[0] - 1] < 0.0) rtmp = -rtmp;
      break;
    case 0: rtmp = rates[i] * y[re[0] - 1]; break;
    default: return 0;
  }
  *out = rtmp; return 1;
}

/* Developer experiment (ORC_F_ORDER=scatter in the environment): the sums of chem_ode_f in the ORDER the GPU engine's LDS scatter adds them
 * -- 64 reactions per pass; within a pass target slot by target slot (reactants first, then products), the reactions of the pass in
 * order within a slot -- instead of the reference's reaction-by-reaction order.  Same terms, other rounding: tools/dev/oracle_f_order.py uses
 * it to ask whether that order is what makes hot cells stall on the GPU. */
static int f_order_scatter(void) {
  static int v = -1;
  if (v < 0) { const char *e = getenv("ORC_F_ORDER"); v = (e && strcmp(e, "scatter") == 0) ? 1 : 0; }
  return v;
}

void orc_ode_f(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
               const double *y, double *ydot) {
  for (int i = 0; i < net->NEQ; i++) ydot[i] = 0.0;
  if (f_order_scatter()) {
    for (int p0 = 0; p0 < net->nR; p0 += 64) {
      double fl[64]; int ok[64];
      const int m = net->nR - p0 < 64 ? net->nR - p0 : 64;
      for (int l = 0; l < m; l++) ok[l] = flux(net, p, cell, rates, y, p0 + l, &fl[l], NULL);
      for (int s = 0; s < 7; s++)
        for (int l = 0; l < m; l++) {
          const int i = p0 + l;
          if (!ok[l]) continue;
          if (s < net->n_reac[i]) ydot[net->reac[3 * i + s] - 1] -= fl[l];
          else if (s - net->n_reac[i] < net->n_prod[i]) ydot[net->prod[4 * i + (s - net->n_reac[i])] - 1] += fl[l];
        }
    }
    ydot[net->nS] = 0.0;
    return;
  }
  for (int i = 0; i < net->nR; i++) {
    double rtmp;
    if (!flux(net, p, cell, rates, y, i, &rtmp, ydot)) continue;
    for (int j = 0; j < net->n_reac[i]; j++) ydot[net->reac[3 * i + j] - 1] -= rtmp;
    for (int j = 0; j < net->n_prod[i]; j++) ydot[net->prod[4 * i + j] - 1] += rtmp;
  }
  ydot[net->nS] = 0.0;
}

/* d(flux_i)/d(y_j); returns 0 for "cycle".  Mirrors the select case of chem_ode_jac. */
static int dflux(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
                 const double *y, int i, int j, double *out, double *pdj) {
  const int *re = net->reac + 3 * i; double rtmp;
  switch (net->itype[i]) {
    case 5: case 6: case 21: case 64:
      if (j == re[0]) rtmp = (re[0] != re[1]) ? rates[i] * y[re[1] - 1] : 2.0 * rates[i] * y[re[1] - 1];
      else if (j == re[1]) rtmp = (re[0] != re[1]) ? rates[i] * y[re[0] - 1] : 2.0 * rates[i] * y[re[0] - 1];
      else rtmp = 0.0;
      if (y[re[0] - 1] < 0.0 && y[re[1] - 1] < 0.0) rtmp = -rtmp;
      break;
    case 1: case 2: case 3: case 13: case 61: case 20: case 0:
      rtmp = (j != re[0]) ? 0.0 : rates[i]; break;
    case 62: case 75:
      if (j != re[0]) rtmp = 0.0;
      else {
        double tmp2 = cell[ORC_P_D2H] * cell[ORC_P_SITES];
        if (net->itype[i] == 75) tmp2 = tmp2 * net->ABC[3 * i + 2];
        if (tmp2 <= 0.0) rtmp = 0.0;
        else {
          double tmp1 = 1.0 / tmp2, tmp = y[re[0] - 1] * tmp1;
          if (tmp <= 1e-4) rtmp = rates[i] * tmp1; else rtmp = rates[i] * tmp1 * exp(-tmp);
        }
      }
      break;
    case 63:
      if (!strcmp(net->reac_name1[i], "gH") && p->H2_form_use_moeq) {
        int i1 = net->counterpart[re[0] - 1];
        if (j == re[0]) rtmp = rates[i] * y[i1 - 1];
        else if (j == i1) rtmp = rates[i] * y[re[0] - 1];
        else rtmp = 0.0;
        if (j == re[0] || j == i1) { pdj[i1 - 1] -= rtmp; pdj[re[0] - 1] += rtmp; }
      } else rtmp = (j == re[0]) ? 2.0 * rates[i] * y[re[0] - 1] : 0.0;
      if (y[re[0] - 1] < 0.0) rtmp = -rtmp;
      break;
    default: return 0;
  }
  (void)cell; *out = rtmp; return 1;
}

void orc_ode_jac_col(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
                     const double *y, int j, double *pdj) {
  for (int i = 0; i < net->NEQ; i++) pdj[i] = 0.0;
  for (int i = 0; i < net->nR; i++) {
    double rtmp;
    if (!dflux(net, p, cell, rates, y, i, j, &rtmp, pdj)) continue;
    if (rtmp != 0.0) {
      for (int k = 0; k < net->n_reac[i]; k++) pdj[net->reac[3 * i + k] - 1] -= rtmp;
      for (int k = 0; k < net->n_prod[i]; k++) pdj[net->prod[4 * i + k] - 1] += rtmp;
    }
  }
  pdj[net->nS] = 0.0;
}

/* Whole Jacobian on the reference CSC pattern.  Reaction-major: each reaction only visits its own
 * reactant columns, accumulating into a dense column scratch in the same reaction order as the
 * column-by-column form above, so the values are bitwise the same as NEQ calls of orc_ode_jac_col
 * (checked in tests/test_oracle.py) at 1/NEQ of the cost. */
void orc_jac_on_pattern(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
                        const double *y, const int *IA, const int *JA, double *vals) {
  const int NEQ = net->NEQ;
  double *dense = calloc((size_t)NEQ * NEQ, sizeof(double)); /* dense[col*NEQ+row] */
  for (int i = 0; i < net->nR; i++) {
    int cols[3], nc = 0;
    for (int k = 0; k < net->n_reac[i] && k < 3; k++) {
      int c = net->reac[3 * i + k], dup = 0;
      for (int q = 0; q < nc; q++) if (cols[q] == c) dup = 1;
      if (!dup && c > 0) cols[nc++] = c;
    }
    if (net->itype[i] == 63 && !strcmp(net->reac_name1[i], "gH") && p->H2_form_use_moeq) {
      int c = net->counterpart[net->reac[3 * i] - 1], dup = 0;
      for (int q = 0; q < nc; q++) if (cols[q] == c) dup = 1;
      if (!dup && c > 0) cols[nc++] = c;
    }
    for (int q = 0; q < nc; q++) {
      int j = cols[q]; double rtmp; double *pdj = dense + (size_t)(j - 1) * NEQ;
      if (!dflux(net, p, cell, rates, y, i, j, &rtmp, pdj)) continue;
      if (rtmp != 0.0) {
        for (int k = 0; k < net->n_reac[i]; k++) pdj[net->reac[3 * i + k] - 1] -= rtmp;
        for (int k = 0; k < net->n_prod[i]; k++) pdj[net->prod[4 * i + k] - 1] += rtmp;
      }
    }
  }
  for (int c = 0; c < NEQ; c++) {
    dense[(size_t)c * NEQ + NEQ - 1] = 0.0;
    for (int k = IA[c] - 1; k < IA[c + 1] - 1; k++) vals[k] = dense[(size_t)c * NEQ + JA[k] - 1];
  }
  free(dense);
}

void orc_jac_csc(const orc_network *net, const orc_params *p, const double *cell, const double *rates,
                 const double *y, double *vals) {
  orc_jac_on_pattern(net, p, cell, rates, y, net->IA, net->JA, vals);
}
