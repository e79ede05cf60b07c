/* racgpu.h -- C ABI of the MI355X-native batched chemistry engine (drop-in for rac-2d's per-cell solve).
 *
 * The reference has no FFI: its seam is the argument-less Fortran call `call chem_evol_solve`
 * (reference src/disk.f90:1686) operating on module globals of `module chemistry`
 * (src/chemistry.f90:158-170).  Each entry point below replaces one piece of that seam and cites it.
 * Plain pointers and sizes only; all arrays are caller-owned, f64 / i32 / i64, cell-major with the
 * species index contiguous (the reference's per-cell abundance layout, src/data_dump.f90:88-162).
 * Species indices in this API are 1-based like the reference's (chem_species%names(1:nSpecies)).
 *
 * Return convention: 0 = ok, negative = error (text via racgpu_last_error()).  Per-cell outcomes
 * (the reference's chemsol_params%quality bitmask, NERR, n_record_real, t_final) come back in arrays.
 * Thread model: one host thread per device; a handle may be used by one thread at a time.
 * The Fortran binding is rac-2d_amd/fortran/racgpu_mod.f90 (ISO_C_BINDING); see INTEGRATION.md.
 */
#ifndef RACGPU_H
#define RACGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- the per-cell input record: the fields of type_cell_rz_phy_basic that the fixed-T path reads
 * (reference src/data_struct.f90:316-442; set by src/disk.f90:1653 `chem_params = leaves%list(id)%p%par`) */
#define RACGPU_NPAR 28
enum {
  RACGPU_P_TGAS = 0,        /* Tgas                       [K]      */
  RACGPU_P_TDUST,           /* Tdust                      [K]      */
  RACGPU_P_NGAS,            /* n_gas                      [cm^-3]  */
  RACGPU_P_GRAIN_RADIUS,    /* GrainRadius_CGS            [cm]     */
  RACGPU_P_SIGDUST,         /* sigdust_ave                [cm^2]   */
  RACGPU_P_NDUST,           /* ndust_tot                  [cm^-3]  */
  RACGPU_P_D2H,             /* ratioDust2HnucNum                   */
  RACGPU_P_SITES,           /* SitesPerGrain                       */
  RACGPU_P_ALBEDO,          /* omega_albedo                        */
  RACGPU_P_ZETA_CR,         /* zeta_cosmicray_H2          [s^-1]   */
  RACGPU_P_ZETA_X,          /* zeta_Xray_H2               [s^-1]   */
  RACGPU_P_NCOL_ISM,        /* Ncol_toISM                 [cm^-2]  */
  RACGPU_P_AV_ISM,          /* Av_toISM                            */
  RACGPU_P_AV_STAR,         /* Av_toStar                           */
  RACGPU_P_G0_ISM,          /* G0_UV_toISM                         */
  RACGPU_P_G0_STAR,         /* G0_UV_toStar                        */
  RACGPU_P_G0_H2PHD,        /* G0_UV_H2phd                         */
  RACGPU_P_G0_PHOTODES,     /* G0_UV_toStar_photoDesorb            */
  RACGPU_P_LYA,             /* phflux_Lya                          */
  RACGPU_P_FSS_ISM_H2, RACGPU_P_FSS_ISM_CO, RACGPU_P_FSS_ISM_H2O, RACGPU_P_FSS_ISM_OH,     /* f_selfshielding_toISM_*  */
  RACGPU_P_FSS_STAR_H2, RACGPU_P_FSS_STAR_CO, RACGPU_P_FSS_STAR_H2O, RACGPU_P_FSS_STAR_OH, /* f_selfshielding_toStar_* */
  RACGPU_P_TMAX             /* per-cell t_max [yr]; <= 0: use params.t_max (src/disk.f90:2078-2085 is the caller's rule) */
};

/* ---- the chemsol_params namelist scalars the path reads (reference src/chemistry.f90:107-135, 183-184) */
typedef struct racgpu_params {
  double RTOL, ATOL;            /* chemsol_params%RTOL, %ATOL                              */
  double t_max;                 /* %t_max   [yr]                                           */
  double dt_first_step;         /* %dt_first_step [yr]                                     */
  double ratio_tstep;           /* %ratio_tstep                                            */
  double max_runtime_allowed;   /* %max_runtime_allowed [s]: drives the guards of src/chemistry.f90:438,480-491 ("Premature
                                   finish", forced ISTATE=1 after a slow interval), applied to MODELLED reference CPU time
                                   = rt_cost_f*NFE + rt_cost_jac*NJE + rt_cost_lu*NLU (the reference reads the wall clock,
                                   which no batched engine can reproduce); <= 0 switches the guards off */
  double Diff2DesorRatio;       /* %Diff2DesorRatio (default 0.5)                          */
  double special_gH_E_diff;     /* %special_gH_E_diff (default 225)                        */
  int32_t mxstep_per_interval;  /* %mxstep_per_interval -> DLSODES MXSTEP (IWORK(6))       */
  int32_t steps_reset_solver;   /* %steps_reset_solver: ISTATE=1 every this many records   */
  int32_t H2_form_use_moeq;     /* must be 0 for the integrator (error otherwise); racgpu_rates honours it (src/chemistry.f90:876-881) */
  int32_t evol_dust_size;       /* must be 0 (error otherwise)                             */
  int32_t use_special_gH_mobi;  /* %use_special_gH_mobi                                    */
  int32_t tol_policy_j;         /* j of chem_set_solver_flags_alt(j) (src/chemistry.f90:205); 1 = as configured */
  int64_t max_steps_per_cell;   /* deterministic work budget (accepted steps); 0 = unlimited.  A cell that
                                   exhausts it stops like the reference's "Premature finish" */
  double rt_cost_f, rt_cost_jac, rt_cost_lu; /* modelled seconds per chem_ode_f call, per full Jacobian (NEQ calls of
                                   chem_ode_jac) and per factorisation incl. its solves; defaults 47e-6, 10.4e-3, 1.0e-3 =
                                   the reference on one core of the build host */
} racgpu_params;

/* per-cell counters returned by the solve calls (int64 x RACGPU_NSTAT per cell) */
#define RACGPU_NSTAT 20
enum { RACGPU_S_NST = 0, RACGPU_S_NFE, RACGPU_S_NJE, RACGPU_S_NLU, RACGPU_S_NERR, RACGPU_S_NREC_REAL,
       RACGPU_S_QSUM /* sum of the order used over accepted steps */, RACGPU_S_NCFAIL_ETFAIL,
       /* shader-clock cycles of the cell's wave, whole solve and per phase (f(y), Jacobian, LU, triangular solves) */
       RACGPU_S_CYC_TOTAL, RACGPU_S_CYC_RHS, RACGPU_S_CYC_JAC, RACGPU_S_CYC_LU, RACGPU_S_CYC_SOLVE,
       /* split of the LU cycles: column scatter, pivots applied through LDS, pivots of the dense trailing block */
       RACGPU_S_CYC_LU_SCATTER, RACGPU_S_CYC_LU_LDS, RACGPU_S_CYC_LU_REG,
       RACGPU_S_ISAV,  /* index (1-based) of the record handed back: the last one whose T and H2 entries are not NaN
                          (src/disk.f90:1716-1721); <= 1 means "No useful data produced": y and t_final were left alone */
       RACGPU_S_NITER, /* local iterations used (racgpu_calc_cells; 1 for a plain solve) */
       RACGPU_S_NREC,  /* n_record of this cell's (last) run, from its own t0, t_max and first step (src/chemistry.f90:1916-1938) */
       RACGPU_S_ERRCODES /* the error returns behind NERR by ISTATE code, 16 bits each from bit 0: -1 (MXSTEP), -4 (error test), -5
                            (convergence), any other; like NERR, of the last local iteration that proceeded (ode_solver_error_handling,
                            src/chemistry.f90:272-387) */ };

/* per-cell values the path writes back into the cell record (double x RACGPU_NOUT per cell) */
#define RACGPU_NOUT 6
enum { RACGPU_O_R_H2_FORM = 0,     /* chem_params%R_H2_form_rate_coeff [s^-1] (src/chemistry.f90:804,891); untouched if the network
                                      has no H2-formation reaction */
       RACGPU_O_N_MOL_ON_GRAIN,    /* chem_params%n_mol_on_grain: get_ice_coverage's side effect on the handed-back abundances
                                      (src/chemistry.f90:989-1003, src/disk.f90:1736); untouched when isav <= 1 */
       RACGPU_O_T_END,             /* touts(n_record_real): where the integration itself stopped (>= t_final when the tail of the
                                      record held NaNs) */
       RACGPU_O_TGAS,              /* c%par%Tgas = record(nSpecies+1, isav) (src/disk.f90:1732): the gas temperature of the hand-off
                                      record; the record's own Tgas when T is not evolving; untouched when isav <= 1 */
       RACGPU_O_EVOLT_END,         /* racgpu_evolT_solve_batch: 1 if T was still evolving at the end of the run, 0 if the T-freeze test
                                      (src/chemistry.f90:532-546) or en_gain_tot <= 0 had switched it off */
       RACGPU_O_TFREEZE_REC };     /* the record (1-based index into touts) after which that test froze T; 0: it never did.  The test
                                      compares the spread of T over five records with a threshold that sits inside the integrator's
                                      own noise, and what follows keeps the rate coefficients of whichever chem_cal_rates call came
                                      last (T, or chem_ode_jac's T + dT): two runs that freeze one record apart end percent apart */

/* flags of racgpu_evol_solve_batch */
#define RACGPU_F_RECTIFY 1 /* apply rectify_abundances (src/chemistry.f90:2170-2201) to y before integrating: the continue path */

/* where the caller's cell/abundance/output buffers live */
#define RACGPU_MEM_HOST 0
#define RACGPU_MEM_DEVICE 1

typedef struct racgpu_network racgpu_network; /* opaque: network + species tables, sparsity, symbolic LU, device copies */

const char *racgpu_last_error(void);
int racgpu_device_count(void);                      /* number of visible HIP devices (0 if none) */

/* chem_read_reactions + chem_load_reactions + chem_parse_reactions + chem_get_dupli_reactions +
 * chem_get_idx_for_special_species + chem_make_sparse_structure (+ ordering and symbolic LU, which the
 * reference redoes per cell inside DLSODES; src/chemistry.f90:1427,1364,1221,1188,1089,1858; src/disk.f90:1566-1581).
 * Host only: usable without a GPU.  Device tables are uploaded lazily by the first compute call. */
racgpu_network *racgpu_network_load(const char *path);
void racgpu_network_destroy(racgpu_network *);
/* nnzJ: species-block Jacobian pattern actually used (dead reactions and the T row/column dropped);
 * nzl/nzu: strict lower/upper fill of its LU.  Any pointer may be NULL. */
int racgpu_network_dims(const racgpu_network *, int32_t *nSpecies, int32_t *nReactions, int32_t *nnzJ, int32_t *nzl, int32_t *nzu);
/* LENRW = IWORK(17) of the reference's DLSODES for this network (the RWORK length it reports as needed).  Used for one
 * reference behaviour only: after an error return the reference re-enters DLSODES with ISTATE = 3, whose sparse-matrix
 * preprocessing zeroes NNZ words at the end of a temporary work area (src/opkda1.f:1487-1494); with the RWORK the reference
 * allocates (20 + 4 NNZ + 28 NEQ, src/chemistry.f90:1945) that area overlaps the tail of the saved Newton matrix, so the next
 * step starts from a partly zeroed P.  The engine reproduces it when it knows LENRW (built in for the four networks in data/;
 * the value depends on YSMP's compressed index storage and cannot be derived without it); 0 = unknown: the saved P survives. */
int racgpu_network_set_reference_lenrw(racgpu_network *, int32_t lenrw);
int racgpu_network_reference_lenrw(const racgpu_network *);
int racgpu_species_name(const racgpu_network *, int32_t i, char *buf, int32_t buflen); /* chem_species%names(i) */
int racgpu_species_index(const racgpu_network *, const char *name);                   /* 0 if absent */
/* reaction table as parsed: reac[nR*3], prod[nR*4] (1-based, 0 = empty), n_reac, n_prod, itype, n_dupli[nR]; any may be NULL */
int racgpu_reactions(const racgpu_network *, int32_t *reac, int32_t *prod, int32_t *n_reac, int32_t *n_prod, int32_t *itype, int32_t *n_dupli);
/* species attributes: mass_num, vib_freq, Edesorb [nS] (NaN where the reference leaves NaN), counterpart (-1 none), charge */
int racgpu_species_attrs(const racgpu_network *, double *mass_num, double *vib_freq, double *Edesorb, int32_t *counterpart, int32_t *charge);
/* chem_species%elements(1:20, i) (src/chemistry.f90:21-34: charge, E, Grain, H, D, He, C, N, O, Si, S, Fe, Na, Mg, Cl, P, F, Ne, Ar, K): elements[nS*20] */
int racgpu_species_elements(const racgpu_network *, int32_t *elements);
/* the reaction rows as read (chem_load_reactions, src/chemistry.f90:1364-1424): ABC[nR*3], T_range[nR*2], ctype[nR*2] and
 * reliability[nR] (characters, not terminated), names[nR*7*12] = reac_names(1:3), prod_names(1:4), blank padded; any may be NULL.
 * What the per-cell rate dump (save_chem_rates, src/disk.f90:3555-3592) and chem_analyse (:4136-4300) print next to a rate. */
int racgpu_reaction_rows(const racgpu_network *, double *ABC, double *T_range, char *ctype, char *reliability, char *names);
/* CSC pattern of the species-block Jacobian: colptr[nS+1], rowidx[nnzJ], 1-based */
int racgpu_jac_pattern(const racgpu_network *, int32_t *colptr, int32_t *rowidx);

/* elimination order of the species-block LU: perm[new] = old species index (1-based); *first_dense = first position (1-based) of
 * the trailing block that is factored as a dense matrix; p_storage[q] = the entry of racgpu_jac_pattern (1-based) held at position q
 * of the engine's storage of the Newton matrix (columns in elimination order).  Any pointer may be NULL.  The reference recomputes
 * its own ordering (YSMP ODRV) inside DLSODES. */
int racgpu_lu_ordering(const racgpu_network *, int32_t *perm, int32_t *first_dense, int32_t *p_storage);
/* chem_load_initial_abundances (src/chemistry.f90:1978-2024): y0[nS], neutralised and renormalised to sum(H)=1 */
int racgpu_load_initial_abundances(const racgpu_network *, const char *path, double *y0);
void racgpu_params_default(racgpu_params *);          /* type defaults + inp/template_configure.dat values */
int racgpu_n_record(const racgpu_params *, double t0, double t_max); /* src/chemistry.f90:1894-1899 */
/* chem_set_solver_flags_alt(j) (src/chemistry.f90:205-268): rtol, atol[nS+1] for one cell's ratioDust2HnucNum */
int racgpu_set_tolerances(const racgpu_network *, const racgpu_params *, int32_t j, double d2h, double *rtol, double *atol);
/* set_initial_condition_4solver (src/disk.f90:2055-2066): y[c,:] = y0, Grain0 slot <- ratioDust2HnucNum of the cell */
int racgpu_init_abundances(const racgpu_network *, const double *y0, const double *cells, int64_t ncell, double *y);

/* ---- device selection (one process per GPU) ---- */
int racgpu_set_device(int dev);
int racgpu_set_stream(racgpu_network *, void *hip_stream); /* NULL = the null stream */

/* ---- test hooks mirroring chem_cal_rates / chem_ode_f / chem_ode_jac (host buffers) ---- */
int racgpu_rates(racgpu_network *, const racgpu_params *, const double *cells, int64_t ncell, double *rates /* [ncell*nR], yr^-1 */);
int racgpu_rhs(racgpu_network *, const racgpu_params *, const double *cells, int64_t ncell, const double *y /* [ncell*nS] */, double *ydot /* [ncell*nS] */);
int racgpu_jac_csc(racgpu_network *, const racgpu_params *, const double *cells, int64_t ncell, const double *y, double *vals /* [ncell*nnzJ] */);
/* solves P x = b with P = I - gamma*J(y) through the engine's own sparse LU: b in, x out [ncell*nS] */
int racgpu_newton_solve(racgpu_network *, const racgpu_params *, const double *cells, int64_t ncell, const double *y, double gamma, double *bx);

/* ---- the hot path: chem_cal_rates + chem_set_solver_flags_alt + chem_evol_solve for a batch of cells
 * (src/disk.f90:1671-1686; src/chemistry.f90:391-588).  One cell per wavefront, all cells independent.
 *   cells    [ncell*RACGPU_NPAR]   in
 *   y        [ncell*nS]            in: abundances at t0 = 0; out: abundances at t_final (record(:, n_record_real))
 *   t_final  [ncell]               out  (NULL ok)
 *   quality  [ncell] int32         out  chemsol_params%quality bitmask {1,2,256,512} (NULL ok)
 *   stats    [ncell*RACGPU_NSTAT]  out  int64 (NULL ok)
 *   record   [ncell*n_record*(nS+1)] out, optional (NULL): chemsol_stor%record incl. the T slot; touts [ncell*n_record]
 *            n_record = racgpu_n_record(params, 0, params->t_max); only allowed when no cell overrides t_max upward
 *   mem      RACGPU_MEM_HOST: buffers are host memory (copied in/out); RACGPU_MEM_DEVICE: device pointers
 */
int racgpu_solve_batch(racgpu_network *, const racgpu_params *, int64_t ncell, const double *cells, double *y,
                       double *t_final, int32_t *quality, int64_t *stats, double *record, double *touts, int mem);
/* The same with everything chem_evol_solve reads from chemsol_params per cell and per local iteration:
 *   t0       [ncell]  chemsol_params%t0 (src/chemistry.f90:419); NULL = 0.  As in set_initial_condition_4solver_continue
 *                     (src/disk.f90:2128-2130) the first output step of a cell is max(params.dt_first_step, 1e-3*t0), and
 *                     n_record follows from (t_max - t0) and that step (chem_evol_solve_prepare_ongoing)
 *   tol_j    [ncell]  int32: j of chem_set_solver_flags_alt(j) per cell; NULL = params.tol_policy_j for all
 *   cell_out [ncell*RACGPU_NOUT] out (NULL ok): RACGPU_O_*
 *   flags    RACGPU_F_*
 * y/t_final out follow the caller's hand-off rule (src/disk.f90:1716-1733): the last record without NaN in T and H2, and its
 * time; stats[RACGPU_S_ISAV] says which record that was. */
int racgpu_evol_solve_batch(racgpu_network *, const racgpu_params *, int64_t ncell, const double *cells, double *y,
                            const double *t0, const int32_t *tol_j, double *t_final, int32_t *quality, int64_t *stats,
                            double *record, double *touts, double *cell_out, int flags, int mem);
/* ---- gas temperature co-evolving with the chemistry: chemsol_params%evolT (set_initial_condition_4solver, src/disk.f90:2066-2073:
 * T evolves in every cell with en_gain_tot > 0).  NEQ = nS + 1: the last unknown is Tgas, its derivative the net of the 11 heating
 * and 17 cooling terms (realtime_heating_cooling_rate, src/disk.f90:4664-4741; heating_minus_cooling, src/heating_cooling.f90:1204-1269),
 * the rate coefficients are recomputed at the iterate's T before every chem_ode_f, the T row and column of the Jacobian come from
 * finite differences (src/disk.f90:4878-4899), and once T has settled the run goes on with T fixed (src/chemistry.f90:532-546).
 * Implemented for the reference's default switches: use_analytical_CII_OI = IonCoolingWithLut = .true., dust_gas_linear_couple =
 * .false., no tandem dust-temperature iteration (a_disk%allow_gas_dust_en_exch / Tdust_iter_tandem = .false.). */
/* the fields of the cell record (type_cell_rz_phy_basic, src/data_struct.f90:316-442) that only the heating/cooling terms read */
#define RACGPU_NHC 28
enum {
  RACGPU_H_EN_GAIN_TOT = 0,  /* en_gain_tot: > 0 switches T evolution on for the cell (src/disk.f90:2071)        */
  RACGPU_H_NCOL_STAR,        /* Ncol_toStar                [cm^-2]                                                */
  RACGPU_H_PAH,              /* PAH_abundance                                                                     */
  RACGPU_H_MMW,              /* MeanMolWeight                                                                     */
  RACGPU_H_OMEGA_K,          /* omega_Kepler               [s^-1]                                                 */
  RACGPU_H_DV_TURB,          /* velo_width_turb            [cm s^-1]                                              */
  RACGPU_H_COHERENT,         /* coherent_length            [cm]                                                   */
  RACGPU_H_NEUFELD_G,        /* Neufeld_G                                                                         */
  RACGPU_H_NEUFELD_DVDZ,     /* Neufeld_dv_dz              [km s^-1 cm^-1]                                        */
  RACGPU_H_DUST_DEPL,        /* dust_depletion (only with use_mygasgraincooling = 0)                              */
  RACGPU_H_VOLUME,           /* volume                     [cm^3]                                                 */
  RACGPU_H_NDUSTCOMPO,       /* ndustcompo (1..4), then four slots each of:                                       */
  RACGPU_H_SIG_DUSTS,        /* sig_dusts(1:4)             [cm^2]                                                 */
  RACGPU_H_N_DUSTS = RACGPU_H_SIG_DUSTS + 4,  /* n_dusts(1:4)   [cm^-3]                                           */
  RACGPU_H_TDUSTS = RACGPU_H_N_DUSTS + 4,     /* Tdusts(1:4)    [K]                                               */
  RACGPU_H_EN_GAINS = RACGPU_H_TDUSTS + 4     /* en_gains(1:4)  [erg s^-1] (caps what the dust can give back)     */
};
/* heating_cooling_config (src/heating_cooling.f90:16-38) as far as these branches read it, a_disk%base_alpha (src/disk.f90:32) and
 * chemsol_params%maySwitchT (src/disk.f90:2070) */
typedef struct racgpu_hc_config {
  double heating_eff_chem, heating_eff_H2form, heating_eff_phd_H2, heating_eff_phd_H2O, heating_eff_phd_OH, cooling_gg_coeff, base_alpha;
  int32_t use_chemicalheatingcooling, use_Xray_heating, use_phdheating_H2, use_phdheating_H2OOH, use_mygasgraincooling, may_switch_T;
} racgpu_hc_config;
void racgpu_hc_config_default(racgpu_hc_config *); /* the reference's template (README.md:135-156) */
/* heating_cooling_prepare + chem_load_species_enthalpies + chem_get_reaction_heat (src/heating_cooling.f90:68-102, src/chemistry.f90:2027-2146):
 * the species-enthalpy file, the Neufeld cooling tables (data/neufeld_cooling_tables.dat: compiled into the reference) and the three ion
 * line-cooling tables N+ / Si+ / Fe+ _LUT.bin.  Host only. */
int racgpu_heating_cooling_load(racgpu_network *, const racgpu_hc_config *, const char *enthalpy_file, const char *neufeld_tables,
                                const char *nii_lut, const char *siii_lut, const char *feii_lut);
/* chem_net%nReacWithHeat, %iReacWithHeat (1-based), %heat [erg] as chem_get_reaction_heat builds them; any pointer may be NULL */
int racgpu_heat_reactions(const racgpu_network *, int32_t *n, int32_t *rxn, double *heat);
/* the 29 values of type_heating_cooling_rates_list in its order (src/data_struct.f90:489-520: the net rate, 11 heating, 17 cooling terms) */
#define RACGPU_NHCTERMS 29
/* test hooks (host buffers): chem_ode_f with T evolving at y [ncell*(nS+1)] (last entry = Tgas) -> ydot [ncell*(nS+1)] and the terms
 * [ncell*RACGPU_NHCTERMS, erg s^-1 cm^-3] behind its last entry; with tcol/trow also chem_ode_jac's finite-difference T column
 * [ncell*(nS+1)] and T row [ncell*10: at H2, H, E-, C, C+, O, O2, CO, H2O, OH] */
int racgpu_evolT_hooks(racgpu_network *, const racgpu_params *, const double *cells, const double *hc, int64_t ncell, const double *y,
                       double *ydot, double *terms, double *tcol, double *trow);
/* racgpu_evol_solve_batch with T evolving: hc [ncell*RACGPU_NHC]; the cell record's Tgas is the initial temperature, cell_out
 * [RACGPU_O_TGAS] the temperature handed back, record's last slot T(t).  One wave per cell. */
int racgpu_evolT_solve_batch(racgpu_network *, const racgpu_params *, int64_t ncell, const double *cells, const double *hc, double *y,
                             const double *t0, const int32_t *tol_j, double *t_final, int32_t *quality, int64_t *stats, double *record,
                             double *touts, double *cell_out, int flags, int mem);
/* calc_this_cell's chemistry for a batch (src/disk.f90:1651-1791): up to nlocal_iter local iterations per cell.  Iteration 1
 * integrates every cell from t = 0 with chem_set_solver_flags_alt(1).  A cell whose run ended with quality != 0 before half
 * of its t_max goes into iteration j = 2, 3, ...: abundances of the hand-off record, rectify_abundances, t0 = t_final,
 * first step max(dt0, 1e-3 t0), tolerances of policy j -- all cells of an iteration in ONE launch.  An iteration that does
 * not get past the previous t_final ("Local iteration does not proceed") or produces no record without NaN ends the
 * cell's loop as in the reference.  y in: abundances at t = 0 (racgpu_init_abundances); out: the handed-back abundances.
 * stats: work counters summed over the iterations, RACGPU_S_NITER = iterations used.  params.tol_policy_j is ignored. */
int racgpu_calc_cells(racgpu_network *, const racgpu_params *, int32_t nlocal_iter, int64_t ncell, const double *cells, double *y,
                      double *t_final, int32_t *quality, int64_t *stats, double *cell_out, int mem);
/* rectify_abundances (src/chemistry.f90:2170-2201) on host arrays: y[c, E-] += sum(charge * y[c, :]) */
int racgpu_rectify_abundances(const racgpu_network *, int64_t ncell, double *y);
/* The caller's sweep in dependency order, for grids whose cells form columns (reference: a cell is solved once the cells above
 * it are done, update_calculating_cells src/disk.f90:1937, because update_params_above_alt :1823-1883 puts integrals over their
 * end states into its record).  Column c holds cells col_cells[col_ptr[c] .. col_ptr[c+1]) from the surface downwards; every
 * column is solved top down by one team of waves (eight; RG_TEAM), columns side by side.  Before a cell is solved, the toISM self-shielding
 * slots of its record are rewritten from the column densities N = sum n_gas X dz of the cells above it: H2 by
 * get_H2_self_shielding(N_H2, dv_turb) (:1887-1897), H2O and OH by exp(-N sigma_Lya) (:1847-1859), CO by get_12CO_shielding(N_H2,
 * N_CO) on the table given to racgpu_set_co_shielding_table (without one the CO slot stays as given), all capped at 1; the toStar
 * slots stay as given unless racgpu_set_star_rays has described the rays.  cells is updated in place; everything
 * else as racgpu_evol_solve_batch with t0 = 0 and the handle's default tolerance policy.  The surface cell of a column gets the
 * slots for N = 0.  col_cells must be a permutation of 0..ncell-1 (checked for host buffers). */
/* 12CO shielding table for racgpu_column_sweep: f[ncol][nrow] > 0 over ascending log10 column densities logN_12CO[ncol], logN_H2[nrow]
 * (the layout of the reference's f_12CO(ncol, nrow), src/load_Visser_CO_selfshielding.f90; its own Visser et al. 2009 table is
 * compiled into it and not shipped here: the caller supplies one).  f == NULL clears it. */
int racgpu_set_co_shielding_table(racgpu_network *, int32_t nrow, int32_t ncol, const double *logN_H2, const double *logN_12CO, const double *f);
/* Rays to the star for racgpu_column_sweep (calc_Ncol_to_Star, src/disk.f90:2543-2555: the reference traces the ray from a cell to
 * the star through its grid; here the caller hands over the result of that tracing in the one-predecessor form a column grid gives):
 * inner[cell] = the cell the ray enters next on its way to the star (-1: none left), ds[cell] = path length of such a ray through
 * `cell` [cm].  With rays set, the sweep also rewrites the toStar self-shielding slots (:1842-1866) from N_toStar(cell) =
 * N_toStar(inner) + n_gas(inner) X(inner) ds(inner), X being the hand-off abundances of inner: a cell is solved once the cell above
 * it AND the cell on its ray are done -- the reference's column-by-column order from the inner edge (src/disk.f90:885-936) as a
 * wavefront.  inner[cell] must lie in a column of lower index.  Host arrays, copied; inner == NULL clears.  racgpu_star_ray_timeouts:
 * cells of the last sweep that gave up waiting for their neighbour (the sweep then fails). */
int racgpu_set_star_rays(racgpu_network *, int64_t ncell, const int32_t *inner, const double *ds);
int racgpu_star_ray_timeouts(const racgpu_network *);
int racgpu_column_sweep(racgpu_network *, const racgpu_params *, int64_t ncolumn, const int32_t *col_ptr, const int32_t *col_cells,
                        int64_t ncell, double *cells, double *y, const double *dz, double dv_turb, double *t_final, int32_t *quality,
                        int64_t *stats, double *cell_out, int mem);
/* Scheduling hint for the following racgpu_solve_batch calls (an extension: the reference has no counterpart; its
 * cell loop, src/disk.f90:864-1010, takes cells in grid order).  cost[ncell] (host memory) is any per-cell measure
 * of expected work, e.g. the step count RACGPU_S_NST or the cycle count RACGPU_S_CYC_TOTAL the same cell needed in
 * the previous global iteration of the disk model; waves then take cells in order of decreasing cost, so the few
 * cells that need many times the median work start first instead of last.  Results do not depend on the order.
 * The hint applies while ncell matches; cost == NULL or ncell == 0 clears it. */
int racgpu_set_cost_hints(racgpu_network *, const double *cost, int64_t ncell);
/* With cost hints in place, a cell whose expected cost exceeds frac x (sum of the costs / wave slots of the GPU) -- a cell that
 * would take that share of the pass's ideal length all by itself -- is solved by a team of eight waves (at most one team per CU),
 * started ahead of the rest.  Same arithmetic in the same order: results do not depend on it.  Default 0.5; frac <= 0: never.
 * Independently of hints, the cells still being integrated when the queue is empty and at most two waves per CU are left are handed
 * over to teams between two integrator steps (frac < 0 switches that off as well). */
int racgpu_set_team_threshold(racgpu_network *, double frac);
/* cells the last solve pass gave to teams */
int64_t racgpu_last_team_cells(const racgpu_network *);
/* cells the last solve pass handed over to teams at its end (synchronises the handle's stream) */
int64_t racgpu_last_parked_cells(racgpu_network *);
/* bytes of device workspace racgpu_solve_batch keeps per cell (grows the handle's workspace on demand) */
int64_t racgpu_workspace_bytes_per_cell(const racgpu_network *);
/* HIP-event time of the last racgpu_solve_batch kernel on its stream, milliseconds (-1 if none) */
double racgpu_last_kernel_ms(const racgpu_network *);

/* ---- one host process, N GPUs of one node (BASELINE.json north_star: "cells shard embarrassingly across the 8 GPUs of one node, with a
 * single RCCL gather over xGMI at output").  The reference has no counterpart (its sweep is a serial loop, src/disk.f90:864-938).
 * racgpu_multi_create loads the network once per device (devices == NULL: 0 .. ndev-1) and opens an RCCL communicator over them
 * (ncclCommInitAll; librccl.so is loaded on first use).  racgpu_multi_calc_cells = racgpu_calc_cells on host buffers with the cells
 * dealt over the devices -- in order of decreasing cost (cost == NULL: index order), round-robin, so that every device gets the same
 * mix -- one host thread per device, and ONE ncclAllGather of the result rows [y | t_final | quality | stats | cell_out] at the end,
 * after which every device holds every cell's result; device 0's copy is returned.  Results do not depend on ndev or on the dealing. */
typedef struct racgpu_multi racgpu_multi;
racgpu_multi *racgpu_multi_create(const char *network_path, int ndev, const int *devices);
void racgpu_multi_destroy(racgpu_multi *);
int racgpu_multi_ndev(const racgpu_multi *);
racgpu_network *racgpu_multi_network(racgpu_multi *, int i); /* device i's handle (e.g. for racgpu_network_set_reference_lenrw) */
const char *racgpu_multi_last_error(void);
/* the dealing rule by itself (host only): owner[c] = device of cell c, position[c] = its place in that device's batch */
int racgpu_multi_deal(int ndev, int64_t ncell, const double *cost, int32_t *owner, int32_t *position);
int racgpu_multi_calc_cells(racgpu_multi *, const racgpu_params *, int32_t nlocal_iter, int64_t ncell, const double *cells, double *y,
                            double *t_final, int32_t *quality, int64_t *stats, double *cell_out, const double *cost);

#ifdef __cplusplus
}
#endif
#endif
