/* oracle.h -- CPU restatement of rac-2d's per-cell chemistry path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under oracle/ is part of the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so, and only as the checker / CPU baseline.
 * The product path (rac-2d_amd/csrc) shares no code with this directory.
 *
 * Parity status: PINNED.  Every stage is checked against outputs of the unmodified reference
 * run in the build container (oracle/_ref/ref_driver, built by `make ref`), committed as
 * tests/golden/ (generator: tests/golden/make_golden.py).  The reference ships no tests or
 * golden vectors of its own for this path.
 *
 * Each function cites the reference file:line it restates (paths relative to the reference root).
 */
#ifndef RAC2D_ORACLE_H
#define RAC2D_ORACLE_H
#include <stdint.h>

#define ORC_NAME_LEN 12
#define ORC_NELEM 20
#define ORC_NPAR 28 /* cell record length, same layout as include/racgpu.h */

/* cell record slots (0-based); mirrors the fields of type_cell_rz_phy_basic that the fixed-T
 * path reads (src/data_struct.f90:316-442) */
enum {
  ORC_P_TGAS = 0, ORC_P_TDUST, ORC_P_NGAS, ORC_P_GRAIN_RADIUS, ORC_P_SIGDUST, ORC_P_NDUST,
  ORC_P_D2H, ORC_P_SITES, ORC_P_ALBEDO, ORC_P_ZETA_CR, ORC_P_ZETA_X, ORC_P_NCOL_ISM,
  ORC_P_AV_ISM, ORC_P_AV_STAR, ORC_P_G0_ISM, ORC_P_G0_STAR, ORC_P_G0_H2PHD, ORC_P_G0_PHOTODES,
  ORC_P_LYA, ORC_P_FSS_ISM_H2, ORC_P_FSS_ISM_CO, ORC_P_FSS_ISM_H2O, ORC_P_FSS_ISM_OH,
  ORC_P_FSS_STAR_H2, ORC_P_FSS_STAR_CO, ORC_P_FSS_STAR_H2O, ORC_P_FSS_STAR_OH, ORC_P_TMAX
};

typedef struct {
  int nS, nR, NEQ;
  char (*names)[ORC_NAME_LEN + 1]; /* species names, index order = first appearance */
  /* reactions, 1-based species indices like the reference (0 = empty slot) */
  int *reac;  /* [nR][3] */
  int *prod;  /* [nR][4] */
  int *n_reac, *n_prod, *itype;
  double *ABC;     /* [nR][3] */
  double *Trange;  /* [nR][2] */
  char (*ctype)[3];
  char (*reac_name1)[ORC_NAME_LEN + 1]; /* first reactant name as read (for the 'H2'/'gH' tests) */
  int *dupli_ptr, *dupli_list; /* CSR of duplicate sets: lower-index twins of each reaction */
  /* species attributes */
  int *elements;   /* [nS][20] */
  double *mass_num, *vib_freq, *Edesorb;
  int *counterpart; /* 1-based, -1 if none */
  int nGrain; int *idxGrain; /* 1-based */
  int idx10[10];   /* H2 H E- C C+ O O2 CO H2O OH, 1-based, 0 if missing */
  int i_Grain0, i_GrainM, i_GrainP, i_gH, i_gH2, i_gH2O;
  /* Jacobian pattern as the reference builds it (CSC, 1-based, NEQ x NEQ incl. T row/col) */
  int NNZ; int *IA, *JA;
} orc_network;

typedef struct { /* the chemsol_params namelist scalars that the path reads */
  double RTOL, ATOL, t_max, dt_first_step, ratio_tstep;
  int mxstep_per_interval, steps_reset_solver, H2_form_use_moeq;
  double Diff2DesorRatio, special_gH_E_diff;
  int use_special_gH_mobi, update_gH_params_realtime;
  double max_runtime_allowed; /* seconds of MODELLED reference CPU time for the guards of :480-491; <= 0 = off */
  double rt_cost_f, rt_cost_jac, rt_cost_lu; /* modelled seconds per f call / full Jacobian / factorisation */
} orc_params;

/* one local iteration of calc_this_cell's loop (src/disk.f90:1651-1791) as orc_calc_cell ran it */
typedef struct {
  double t0, dt_first, t_end /* touts(n_record_real) */, t_final /* after this iteration */, n_mol_on_grain;
  int n_record, quality /* of the cell after this iteration */, nerr, isav, proceeds;
} orc_iter_info;

typedef struct { long nst, nfe, nje, nlu; int nnz, nzl, nzu; long nst_last, nfe_last, nje_last, nlu_last; } orc_stats;

#ifdef __cplusplus
extern "C" {
#endif
orc_network *orc_network_load(const char *path, char *err, int errlen);
void orc_network_free(orc_network *);
int orc_species_index(const orc_network *, const char *name); /* 1-based, 0 if absent */
int orc_load_initial_abundances(const orc_network *, const char *path, double *y0 /* [nS] */);
void orc_params_default(orc_params *);
int orc_n_record(const orc_params *, double t0, double t_max);
void orc_set_tolerances(const orc_network *, const orc_params *, int j, double d2h, double *rtol, double *atol /* [NEQ] */);

int orc_cal_rates(const orc_network *, const orc_params *, const double *cell, double *rates /* [nR] */,
                  double *R_H2_form);
void orc_ode_f(const orc_network *, const orc_params *, const double *cell, const double *rates,
               const double *y, double *ydot /* [NEQ] */);
void orc_ode_jac_col(const orc_network *, const orc_params *, const double *cell, const double *rates,
                     const double *y, int j /* 1-based */, double *pdj /* [NEQ] */);
void orc_jac_csc(const orc_network *, const orc_params *, const double *cell, const double *rates,
                 const double *y, double *vals /* [NNZ] on IA/JA */);

void orc_jac_on_pattern(const orc_network *, const orc_params *, const double *cell, const double *rates,
                        const double *y, const int *IA, const int *JA /* 1-based CSC */, double *vals);

/* chem_evol_solve restatement.  y: [NEQ] in/out (slot NEQ-1 = Tgas).  record may be NULL, else
 * [n_record][NEQ]; touts may be NULL, else [n_record].  Returns 0, or <0 on fatal (error_stop) paths. */
int orc_evol_solve(const orc_network *, const orc_params *, const double *cell, const double *rates,
                   double *rtol, double *atol, double *y, double t0, double t_max,
                   double *t_final, int *quality, int *nerr, int *n_record_real,
                   double *record, double *touts, orc_stats *stats);
/* calc_this_cell-style convenience: y0 -> Grain0 slot, T slot, tolerances(j=1), rates, solve. */
int orc_solve_cell(const orc_network *, const orc_params *, const double *cell, const double *y0,
                   double *y_out, double *t_final, int *quality, int *nerr, orc_stats *stats);
/* rectify_abundances, src/chemistry.f90:2170-2201 */
void orc_rectify_abundances(const orc_network *, double *y /* [nS] */);
/* The local-iteration loop of calc_this_cell (src/disk.f90:1651-1791) with set_initial_condition_4solver_continue
 * (src/disk.f90:2103-2146): y_init [nS] = abundances at t = 0 (Grain0 slot already set); abund_out [nS]; info [nlocal_iter].
 * Returns the number of iterations run (>= 1), or < 0 on a fatal path. */
int orc_calc_cell(const orc_network *, const orc_params *, const double *cell, const double *y_init, int nlocal_iter,
                  double *abund_out, double *t_final, int *quality, orc_iter_info *info, orc_stats *stats);
#ifdef __cplusplus
}
#endif
#endif
