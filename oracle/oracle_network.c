/* oracle_network.c -- network / species / initial-abundance restatement.  TEST INFRASTRUCTURE ONLY.
 * Follows (reference paths): src/chemistry.f90:1427-1454 chem_read_reactions, :1364-1424
 * chem_load_reactions, :1221-1360 chem_parse_reactions, :1458-1529 getElements, :1532-1539
 * getVibFreq, :1188-1217 chem_get_dupli_reactions, :1089-1185 chem_get_idx_for_special_species,
 * :1858-1885 chem_make_sparse_structure, :1943-1973 chem_prepare_solver_storage (IA/JA),
 * :1978-2024 chem_load_initial_abundances, :205-268 chem_set_solver_flags_alt, :1894-1899 n_record.
 */
#define _GNU_SOURCE
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char *ELEM_NAMES[ORC_NELEM] = {"+-", "E", "Grain", "H", "D", "He", "C", "N", "O", "Si",
                                            "S", "Fe", "Na", "Mg", "Cl", "P", "F", "Ne", "Ar", "K"};
static const double ELEM_MASS[ORC_NELEM] = {0.0, 5.45e-4, 0.0, 1.0, 2.0, 4.0, 12.0, 14.0, 16.0, 28.0,
                                            32.0, 56.0, 23.0, 24.0, 35.5, 31.0, 19.0, 20.18, 39.95, 39.1};

/* Fortran Fw.0 input field: blanks ignored, blank field = 0, D/E exponents, bare-sign exponents. */
static double fortran_real(const char *s, int w) {
  char buf[64]; int n = 0;
  for (int i = 0; i < w && s[i]; i++) if (s[i] != ' ') buf[n++] = s[i];
  buf[n] = 0;
  if (n == 0) return 0.0;
  for (int i = 0; i < n; i++) if (buf[i] == 'D' || buf[i] == 'd') buf[i] = 'e';
  /* "1.5-10" form: a sign after a digit/point that is not preceded by e */
  for (int i = 1; i < n; i++)
    if ((buf[i] == '+' || buf[i] == '-') && buf[i - 1] != 'e' && buf[i - 1] != 'E') {
      memmove(buf + i + 1, buf + i, (size_t)(n - i + 1)); buf[i] = 'e'; n++; break;
    }
  return strtod(buf, NULL);
}
static int fortran_int(const char *s, int w) {
  char buf[32]; int n = 0;
  for (int i = 0; i < w && s[i]; i++) if (s[i] != ' ') buf[n++] = s[i];
  buf[n] = 0;
  return n ? atoi(buf) : 0;
}
static void trim_copy(char *dst, const char *src, int w) { /* trim trailing blanks of a w-wide field */
  int n = w; while (n > 0 && (src[n - 1] == ' ' || src[n - 1] == 0)) n--;
  memcpy(dst, src, (size_t)n); dst[n] = 0;
}

/* src/chemistry.f90:1458-1529 getElements, restated character by character (1-based positions) */
static void get_elements(const char *name12 /* blank padded, 12+ chars */, int *arr) {
  int belongto[40] = {0}; int used[40] = {0};
  int lenName = 12; while (lenName > 0 && name12[lenName - 1] == ' ') lenName--;
  for (int e = 0; e < ORC_NELEM; e++) arr[e] = 0;
  for (int e = 1; e <= ORC_NELEM; e++) {
    int lenEle = (int)strlen(ELEM_NAMES[e - 1]);
    for (int j = 1; j <= lenName - lenEle + 1; j++) {
      if (strncmp(name12 + j - 1, ELEM_NAMES[e - 1], (size_t)lenEle) != 0) continue;
      int replace = 1;
      for (int k = j; k <= j + lenEle - 1; k++) {
        if (used[k]) {
          if ((int)strlen(ELEM_NAMES[belongto[k] - 1]) >= lenEle) { replace = 0; break; }
          arr[belongto[k] - 1] -= 1;
        }
      }
      if (replace) {
        for (int k = j; k <= j + lenEle - 1; k++) { belongto[k] = e; used[k] = 1; }
        arr[e - 1] += 1;
      }
    }
  }
  for (int i = 2; i <= lenName; i++) {
    if (used[i]) continue;
    for (int j = 1; j <= i - 1; j++) if (used[i - j]) { belongto[i] = belongto[i - j]; break; }
    char p = name12[i - 2], c = name12[i - 1], nx = name12[i];
    if ((p > '9' || p < '0') && (c <= '9' && c >= '0')) {
      int ntmp;
      if (nx <= '9' && nx >= '0') ntmp = (c - '0') * 10 + (nx - '0');
      else ntmp = c - '0';
      if (ntmp == 0) continue;
      if (belongto[i] > 0) arr[belongto[i] - 1] += ntmp - 1;
    } else if (c == '+') arr[0] = 1;
    else if (c == '-') arr[0] = -1;
  }
}

static double vib_freq(double massnum, double Edesorb) { /* src/chemistry.f90:1532-1539 */
  const double kB = 1.3806503e-16, mp = 1.67262158e-24, Pi = 3.1415926535897932384626433, sites = 1e15;
  return sqrt(2.0 * sites * kB * Edesorb / (Pi * Pi) / (mp * massnum));
}

int orc_species_index(const orc_network *net, const char *name) {
  for (int i = 0; i < net->nS; i++) if (strcmp(net->names[i], name) == 0) return i + 1;
  return 0;
}

orc_network *orc_network_load(const char *path, char *err, int errlen) {
  FILE *f = fopen(path, "r");
  if (!f) { if (err) snprintf(err, (size_t)errlen, "cannot open %s", path); return NULL; }
  /* pass 1: rows = lines whose first character is neither '!' nor blank (:1442-1448) */
  char *line = NULL; size_t cap = 0; ssize_t len; int nR = 0;
  while ((len = getline(&line, &cap, f)) >= 0) {
    if (len == 0 || line[0] == '!' || line[0] == ' ' || line[0] == '\n' || line[0] == '\r') continue;
    nR++;
  }
  rewind(f);
  orc_network *net = calloc(1, sizeof *net);
  net->nR = nR;
  net->reac = calloc((size_t)nR * 3, sizeof(int)); net->prod = calloc((size_t)nR * 4, sizeof(int));
  net->n_reac = calloc((size_t)nR, sizeof(int)); net->n_prod = calloc((size_t)nR, sizeof(int));
  net->itype = calloc((size_t)nR, sizeof(int));
  net->ABC = calloc((size_t)nR * 3, sizeof(double)); net->Trange = calloc((size_t)nR * 2, sizeof(double));
  net->ctype = calloc((size_t)nR, sizeof *net->ctype);
  net->reac_name1 = calloc((size_t)nR, sizeof *net->reac_name1);
  char (*rn)[3][ORC_NAME_LEN + 1] = calloc((size_t)nR, sizeof *rn);
  char (*pn)[4][ORC_NAME_LEN + 1] = calloc((size_t)nR, sizeof *pn);
  int r = 0;
  while ((len = getline(&line, &cap, f)) >= 0 && r < nR) {
    if (len == 0 || line[0] == '!' || line[0] == ' ' || line[0] == '\n' || line[0] == '\r') continue;
    char row[151]; memset(row, ' ', 150); row[150] = 0;
    for (int i = 0; i < 150 && i < len && line[i] != '\n' && line[i] != '\r'; i++) row[i] = line[i];
    /* FMT '(7(A12), 3F9.0, 2F6.0, I3, X, A1, X, A2)' (:1386-1394) */
    for (int k = 0; k < 3; k++) trim_copy(rn[r][k], row + 12 * k, 12);
    for (int k = 0; k < 4; k++) trim_copy(pn[r][k], row + 36 + 12 * k, 12);
    for (int k = 0; k < 3; k++) net->ABC[3 * r + k] = fortran_real(row + 84 + 9 * k, 9);
    for (int k = 0; k < 2; k++) net->Trange[2 * r + k] = fortran_real(row + 111 + 6 * k, 6);
    net->itype[r] = fortran_int(row + 123, 3);
    net->ctype[r][0] = row[129]; net->ctype[r][1] = row[130]; net->ctype[r][2] = 0;
    strcpy(net->reac_name1[r], rn[r][0]);
    /* n_reac / n_prod (:1395-1422) */
    for (int k = 0; k < 3; k++) {
      if (rn[r][k][0]) net->n_reac[r]++;
      if (!strcmp(rn[r][k], "PHOTON") || !strcmp(rn[r][k], "CRPHOT") || !strcmp(rn[r][k], "CRP")) net->n_reac[r]--;
    }
    for (int k = 0; k < 4; k++) {
      if (pn[r][k][0]) net->n_prod[r]++;
      if (!strcmp(pn[r][k], "PHOTON")) net->n_prod[r]--;
    }
    r++;
  }
  fclose(f); free(line);
  /* species indexing: order of first appearance, reactants then products (:1230-1265) */
  int capS = 1024, nS = 0;
  char (*names)[ORC_NAME_LEN + 1] = calloc((size_t)capS, sizeof *names);
  if (nR > 0) { strcpy(names[0], rn[0][0]); nS = 1; }
  for (r = 0; r < nR; r++) {
    for (int side = 0; side < 2; side++) {
      int cnt = side ? net->n_prod[r] : net->n_reac[r];
      for (int k = 0; k < cnt; k++) {
        const char *nm = side ? pn[r][k] : rn[r][k];
        int found = 0;
        for (int j = 0; j < nS; j++) if (!strcmp(names[j], nm)) { found = j + 1; break; }
        if (!found) {
          if (nS >= capS) { if (err) snprintf(err, (size_t)errlen, "too many species"); return NULL; }
          strcpy(names[nS++], nm); found = nS;
        }
        if (side) net->prod[4 * r + k] = found; else net->reac[3 * r + k] = found;
      }
    }
  }
  net->nS = nS; net->NEQ = nS + 1; net->names = names;
  net->elements = calloc((size_t)nS * ORC_NELEM, sizeof(int));
  net->mass_num = calloc((size_t)nS, sizeof(double));
  net->vib_freq = malloc((size_t)nS * sizeof(double)); net->Edesorb = malloc((size_t)nS * sizeof(double));
  net->counterpart = malloc((size_t)nS * sizeof(int));
  for (int i = 0; i < nS; i++) {
    char padded[16]; memset(padded, ' ', 15); padded[15] = 0; memcpy(padded, names[i], strlen(names[i]));
    get_elements(padded, net->elements + (size_t)i * ORC_NELEM);
    double m = 0; for (int e = 0; e < ORC_NELEM; e++) m += (double)net->elements[i * ORC_NELEM + e] * ELEM_MASS[e];
    net->mass_num[i] = m; net->vib_freq[i] = NAN; net->Edesorb[i] = NAN; net->counterpart[i] = -1;
  }
  for (r = 0; r < nR; r++) if (net->itype[r] == 62) { /* :1321-1331 */
    int a = net->reac[3 * r], p = net->prod[4 * r];
    if (a > 0) {
      net->vib_freq[a - 1] = vib_freq(net->mass_num[a - 1], net->ABC[3 * r + 2]);
      net->Edesorb[a - 1] = net->ABC[3 * r + 2];
      if (p > 0) { net->counterpart[p - 1] = a; net->counterpart[a - 1] = p; }
    }
  }
  net->idxGrain = malloc((size_t)nS * sizeof(int));
  for (int i = 0; i < nS; i++) if (names[i][0] == 'g') net->idxGrain[net->nGrain++] = i + 1;
  /* duplicate sets (:1188-1217) */
  net->dupli_ptr = calloc((size_t)nR + 1, sizeof(int));
  int capD = 64, nD = 0; net->dupli_list = malloc((size_t)capD * sizeof(int));
  for (int i = 0; i < nR; i++) {
    for (int j = 0; j < i; j++) {
      if (net->itype[i] != net->itype[j] || strcmp(net->ctype[i], net->ctype[j])) continue;
      if (memcmp(net->reac + 3 * i, net->reac + 3 * j, 3 * sizeof(int))) continue;
      if (memcmp(net->prod + 4 * i, net->prod + 4 * j, 4 * sizeof(int))) continue;
      if (nD >= capD) { capD *= 2; net->dupli_list = realloc(net->dupli_list, (size_t)capD * sizeof(int)); }
      net->dupli_list[nD++] = j + 1;
    }
    net->dupli_ptr[i + 1] = nD;
  }
  /* special species (:1089-1185) */
  static const char *ten[10] = {"H2", "H", "E-", "C", "C+", "O", "O2", "CO", "H2O", "OH"};
  for (int k = 0; k < 10; k++) net->idx10[k] = orc_species_index(net, ten[k]);
  net->i_Grain0 = orc_species_index(net, "Grain0"); net->i_GrainM = orc_species_index(net, "Grain-");
  net->i_GrainP = orc_species_index(net, "Grain+"); net->i_gH = orc_species_index(net, "gH");
  net->i_gH2 = orc_species_index(net, "gH2"); net->i_gH2O = orc_species_index(net, "gH2O");
  /* sparsity mask -> CSC IA/JA, rows ascending (:1858-1885, :1962-1971) */
  int NEQ = net->NEQ; unsigned char *mask = calloc((size_t)NEQ * NEQ, 1); /* mask[col*NEQ+row] */
  for (r = 0; r < nR; r++)
    for (int j = 0; j < net->n_reac[r]; j++) {
      int cj = net->reac[3 * r + j]; if (cj <= 0) continue;
      for (int k = 0; k < net->n_reac[r]; k++) { int ri = net->reac[3 * r + k]; if (ri > 0) mask[(size_t)(cj - 1) * NEQ + ri - 1] = 1; }
      for (int k = 0; k < net->n_prod[r]; k++) { int ri = net->prod[4 * r + k]; if (ri > 0) mask[(size_t)(cj - 1) * NEQ + ri - 1] = 1; }
    }
  for (int i = 0; i < NEQ; i++) mask[(size_t)(NEQ - 1) * NEQ + i] = 1;
  for (int k = 0; k < 10; k++) if (net->idx10[k] > 0) mask[(size_t)(net->idx10[k] - 1) * NEQ + NEQ - 1] = 1;
  int nnz = 0; for (size_t q = 0; q < (size_t)NEQ * NEQ; q++) nnz += mask[q];
  net->NNZ = nnz; net->IA = malloc((size_t)(NEQ + 1) * sizeof(int)); net->JA = malloc((size_t)nnz * sizeof(int));
  int k = 1; net->IA[0] = 1;
  for (int c = 0; c < NEQ; c++) {
    for (int rr = 0; rr < NEQ; rr++) if (mask[(size_t)c * NEQ + rr]) net->JA[k++ - 1] = rr + 1;
    net->IA[c + 1] = k;
  }
  free(mask); free(rn); free(pn);
  return net;
}

void orc_network_free(orc_network *n) {
  if (!n) return;
  free(n->names); free(n->reac); free(n->prod); free(n->n_reac); free(n->n_prod); free(n->itype);
  free(n->ABC); free(n->Trange); free(n->ctype); free(n->reac_name1); free(n->dupli_ptr); free(n->dupli_list);
  free(n->elements); free(n->mass_num); free(n->vib_freq); free(n->Edesorb); free(n->counterpart);
  free(n->idxGrain); free(n->IA); free(n->JA); free(n);
}

/* src/chemistry.f90:1978-2024 */
int orc_load_initial_abundances(const orc_network *net, const char *path, double *y0) {
  FILE *f = fopen(path, "r"); if (!f) return -1;
  for (int i = 0; i < net->nS; i++) y0[i] = 0.0;
  char *line = NULL; size_t cap = 0; ssize_t len;
  while ((len = getline(&line, &cap, f)) >= 0) {
    char row[65]; memset(row, ' ', 64); row[64] = 0;
    for (int i = 0; i < 64 && i < len && line[i] != '\n' && line[i] != '\r'; i++) row[i] = line[i];
    char nm[ORC_NAME_LEN + 1]; trim_copy(nm, row, 12);
    for (int i = 0; i < net->nS; i++)
      if (!strcmp(nm, net->names[i])) { y0[i] = fortran_real(row + 12, 16); break; }
  }
  free(line); fclose(f);
  int iE = net->idx10[2];
  if (iE <= 0) return -2;
  double s = 0; for (int i = 0; i < net->nS; i++) s += y0[i] * (double)net->elements[i * ORC_NELEM + 0];
  y0[iE - 1] = y0[iE - 1] + s;
  if (y0[iE - 1] < 0.0) return -3; /* "Cannot neutralize the initial condition!" -> error_stop */
  double totH = 0; for (int i = 0; i < net->nS; i++) totH += (double)net->elements[i * ORC_NELEM + 3] * y0[i];
  for (int i = 0; i < net->nS; i++) y0[i] = y0[i] / totH;
  return 0;
}

void orc_params_default(orc_params *p) { /* type defaults src/chemistry.f90:107-135 + template values */
  p->RTOL = 1e-4; p->ATOL = 1e-30; p->t_max = 1e6; p->dt_first_step = 1e-8; p->ratio_tstep = 1.1;
  p->mxstep_per_interval = 6000; p->steps_reset_solver = 50; p->H2_form_use_moeq = 0;
  p->Diff2DesorRatio = 0.5; p->special_gH_E_diff = 225.0; p->use_special_gH_mobi = 0;
  p->update_gH_params_realtime = 0; p->max_runtime_allowed = 60.0;
  p->rt_cost_f = 47e-6; p->rt_cost_jac = 10.4e-3; p->rt_cost_lu = 1.0e-3;
}

int orc_n_record(const orc_params *p, double t0, double t_max) { /* :1894-1899 */
  return (int)ceil(log((t_max - t0) / p->dt_first_step * (p->ratio_tstep - 1.0) + 1.0) / log(p->ratio_tstep)) + 1;
}

/* chem_set_solver_flags_alt(j), src/chemistry.f90:205-268 */
void orc_set_tolerances(const orc_network *net, const orc_params *p, int j, double d2h, double *rtol, double *atol) {
  int nS = net->nS, NEQ = net->NEQ; double r, a, rT, aT;
  switch (j) {
    case 1: r = p->RTOL; a = p->ATOL; rT = 1e-3; aT = 1e-1; break;
    case 2: r = fmin(p->RTOL * 1e1, 1e-4); a = fmin(p->ATOL * 1e5, 1e-25); rT = 1e-2; aT = 1e-1; break;
    case 3: r = fmin(p->RTOL * 1e2, 1e-4); a = fmin(p->ATOL * 1e10, 1e-20); rT = 1e-3; aT = 1e0; break;
    case 4: r = fmin(p->RTOL * 1e2, 1e-4); a = fmin(p->ATOL * 1e10, 1e-18); rT = 1e-3; aT = 1e0; break;
    default: r = fmin(p->RTOL * pow(2.0, j), 1e-3); a = fmin(p->ATOL * pow(1e2, j), 1e-15); rT = 1e-2; aT = 1e0; break;
  }
  for (int i = 0; i < NEQ; i++) { rtol[i] = r; atol[i] = a; }
  rtol[nS] = rT; atol[nS] = aT;
  for (int k = 0; k < 10; k++) if (net->idx10[k] > 0) {
    rtol[net->idx10[k] - 1] = fmax(p->RTOL, 1e-4); atol[net->idx10[k] - 1] = fmax(p->ATOL, 1e-30);
  }
  if (net->i_Grain0 > 0) {
    int g[3] = {net->i_Grain0, net->i_GrainM, net->i_GrainP};
    for (int k = 0; k < 3; k++) if (g[k] > 0) { rtol[g[k] - 1] = 1e-4; atol[g[k] - 1] = fmax(d2h * 1e-6, 1e-30); }
  }
  for (int k = 0; k < net->nGrain; k++) {
    rtol[net->idxGrain[k] - 1] = fmax(p->RTOL, 1e-3); atol[net->idxGrain[k] - 1] = fmax(p->ATOL, d2h * 1e-8);
  }
}
