// device_tables.hpp -- plain-old-data views of the network tables as the gfx950 kernels see them.
#pragma once
#include <cstdint>

// Table pointers are read by the kernels through a DevNet that itself lives in device memory; spelling out the
// global address space keeps every table access a global_load (a pointer fetched from memory is otherwise
// 'flat', and flat loads tie the LDS and vector-memory wait counters together).
#define RG_GLOBAL __attribute__((address_space(1)))

namespace racgpu {

// flux kinds (values of racgpu::Kind in network.hpp)
constexpr int K_NONE_ = 0, K_TWO_ = 1, K_ONE_ = 2, K_SURF_ = 3, K_SURF75_ = 4, K_SQ_ = 5;

// extents of column j in the U, L and P storage; ur = end of the U entries with rows < ns (the pivots applied through
// LDS); [d0, d1) = the column's slice of the pivot descriptor stream
struct alignas(64) LuCol { int u0, u1, lc0, lc1, p0, p1, ur, d0, d1, j, o0, o1, pad[4]; }; // j = the column this work item factors
#ifndef RG_JAC_UNROLL
#define RG_JAC_UNROLL 8
#endif
#ifndef RG_SWEEP_DEPTH
#define RG_SWEEP_DEPTH 8
#endif
#ifndef RG_LU_DEPTH
#define RG_LU_DEPTH 6
#endif
#ifndef RG_LU_OPS
#define RG_LU_OPS 0 // 0: one operation per pivot (DevNet::Udesc); 1: the pivots k < ns of a column as entry-parallel operations (DevNet::Uop):
                    // 30 % fewer instructions and a third fewer LDS atomics, 6-8 % SLOWER (profiles/r3_tuning.txt): kept as a measured experiment
#endif
#ifndef RG_LU_OPS_DEPTH
#define RG_LU_OPS_DEPTH 4 // operations whose L values and rows are in flight
#endif
// depths and group sizes below: measured on the configs[2] scan, round 2 (profiles/r2_tuning.txt)
constexpr int kJacUnroll = RG_JAC_UNROLL; // rows of the Jacobian term stream per unrolled step (the stream is padded to a multiple)
constexpr int kSweepDepth = RG_SWEEP_DEPTH; // chunks of a triangular-solve stream in flight; the schedules are padded to a multiple
#ifndef RG_TEAM
#define RG_TEAM 8 // (round 3: eight waves; four: evolT pass 22.9 s instead of 18.1 s, everything else within 1 %)
#endif
constexpr int kTeam = RG_TEAM; // waves of a team (k_solve_team: the cells that would otherwise set the length of a pass)
constexpr int kLuDepth = RG_LU_DEPTH; // L columns in flight per wave in the LDS pivot loop
constexpr int kLuOpsDepth = RG_LU_OPS_DEPTH;
constexpr int kRateTab = 640; // doubles of the LDS table of pow / exp values per distinct exponent / barrier (k_solve_T)

struct DevNet {
  int nS, nR, npad;          // npad = nS rounded up to 64
  int nnzJ, nzl, nzu;
  int ns;                    // first row/column of the dense trailing block of the LU (n - ns <= 128)
  // ---- rate coefficients (one row per reaction, original file order) ----
  const int16_t *r_itype;
  const uint16_t *r_re0, *r_re1;   // 0-based reactant species (0xFFFF none)
  const uint8_t *r_nreac;
  const uint8_t *r_fss;            // 0 none, 1 H2, 2 CO, 3 H2O, 4 OH
  const uint8_t *r_flags;          // bit0: first reactant is H2; bit1: first reactant is gH; bit2: type-21 pair has opposite charges
  const uint16_t *r_id3;           // type 21: the non-dust reactant
  const double *r_A, *r_B, *r_C, *r_T0, *r_T1;
  // itype 5/6: the distinct exponents B and barriers C of the network (rate06+grain: 170 and 278 for 4 191 reactions) and, per reaction,
  // their indices (ib | ic << 16).  With T evolving every f(y) redoes chem_cal_rates: pow(T/300, B) and exp(-C/T) are then evaluated
  // once per distinct value into an LDS table and looked up per reaction (dev_rates, tab) -- the same function on the same arguments.
  const double *r_ub, *r_uc; const uint32_t *r_ibc; int n_ub, n_uc;
  const double *s_mass, *s_vib, *s_Edes;
  const int *dupli_ptr, *dupli_list; // 0-based reaction numbers
  // ---- RHS: one packed row per reaction ----
  // w0: kind | n_reac<<8 | a<<16 | b<<32 ; w1: targets 0..3 ; w2: targets 4..6 (u16 each, 0xFFFF = none);
  // target slots 0..n_reac-1 subtract, the rest add
  const uint64_t *rhs_w0, *rhs_w1, *rhs_w2;
  // ---- Jacobian gather: one linear stream (built in engine.hip, upload) of jac_rows rows of 64 term words
  // (rxn | sa<<16 | kind<<32 | flags<<40 | sb<<48), entries sorted by decreasing term count, 64 per pass
  const uint64_t *jac_stream, *jac_slot;
  const uint32_t *jac_rowflag;
  int jac_rows;
  int jac_seg_row[kTeam + 1], jac_seg_pass[kTeam + 1]; // the stream in kTeam segments of whole passes (first row / first pass; the last = the end): k_solve_team
  // ---- sparse LU of the permuted species block ----
  const uint16_t *perm;      // perm[new] = old
  const uint16_t *Lrow, *Urow, *Prow; // storage layout: see network.hpp, struct Symbolic
  const uint8_t *Pdiag;       // [nnzJ] in storage order: 1 on the diagonal
  // ISTATE = 3 in the reference zeroes the entries of the saved P whose position in ITS storage is >= ref_nnz1 - Z,
  // Z = ref_zbase + NEQ * min(nq + 1, 7) (network.hpp, HostNetwork::ref_kref); ref_clobber = 0: unknown layout, nothing is zeroed
  const uint16_t *Pkref;      // [nnzJ] in storage order: position in the reference's storage
  int ref_nnz1, ref_zbase, ref_clobber;
  // triangular-solve schedules: one packed word per stored entry of the streamed part, row | col<<10 |
  // (next chunk continues this level)<<20; chunks of 64 entries hold one dependency level each; null: row == col
  const uint32_t *Lrc, *Urc;
  // LU pivot descriptors, per column j one slice [d0, d1): for every pivot k < ns of the column, in U storage order, one word
  // for (a piece of at most 64 rows of) L column k, [start, start+len) in the storage: bits 0-15 k; 16-25 8*len; 30 "reload w[k]"
  // (the pivot opens a new level within column j, or is the first of a fetch of 60); 32-52 2*start; 53-61 8*(len-1) (0 for
  // len 0) -- everything the pivot loop needs as byte offsets.
  const unsigned long long *Udesc;
  // RG_LU_OPS: the same pivots as entry-parallel operations.  Per column j one slice [o0, o1) of operations (LuCol::o0/o1); an operation is 64
  // words, one per lane: (position in L) << 16 | pivot column k (bit 12 of the first word: the next operation continues the level), for up to 64 (pivot, L entry) pairs of ONE dependency level of the
  // column (levels in order, pivots of a level in U storage order, an L column's entries in storage order; an L column may straddle
  // operations).  Lanes past the end of a level name the spare entries behind L (nzl + lane), whose row list entries name the spare LDS
  // doubles behind the work column: nothing is masked.
  const uint32_t *Uop;
  const LuCol *lucol;        // [nwork+2] the LU's work list (engine.hip, upload): columns with pivots, then the trailing block
  int nwork_sparse, nwork;   // work items with j < ns / in all
  // k_solve_team: the work items with j < ns again, one list per wave of a team, ordered by dependency level (engine.hip, upload)
  const LuCol *lucol_team;   // wave w's list starts at team_base[w]
  const int *team_lev_ptr;   // [kTeam][team_nlev + 1]: where each level starts in the wave's list
  int team_base[kTeam], team_nlev;
  const unsigned long long *leaf_diag, *leaf_ent; // pivot-free columns, factored elementwise beforehand
  int nleaf, nleaf_ent;
  int nchunkL, nchunkU;
  int nzl_stream, nzu_stream; // entries of the streamed parts; the trailing columns follow back to back (closed-form starts)
  // type-11 special indices (0-based, -1 none)
  int i_H, i_E, i_gH, i_gH2, i_gH2O, i_Grain0, i_GrainM, i_GrainP;
  int i_H2;                  // hand-off test (reference src/disk.f90:1716-1721)
  int charge_conserved;      // 1: RACGPU_CHARGE_BALANCE is set, every reaction conserves charge and E- is a species (dev_rhs; off by default)
  int grain_conserved;       // 1: every reaction has as many Grain0/Grain-/Grain+ among its reactants as among its products (dev_rhs)
  int moeq_r61, moeq_r62;    // H2_form_use_moeq: the adsorption reaction of H and the desorption reaction of gH (-1 none)
  int r_h2form;              // last reaction whose coefficient the reference copies into R_H2_form_rate_coeff (-1 none)
  const uint8_t *s_tolclass; // 0 generic, 1 one of the ten special species, 2 Grain0/+/-, 3 surface species (applied in that order)
  const int8_t *s_charge;    // elements(1, :) of every species (rectify_abundances, reference src/chemistry.f90:2170-2201)
};

struct DevParams {
  double RTOL, ATOL, t_max, dt_first_step, ratio_tstep, Diff2DesorRatio, special_gH_E_diff;
  int mxstep, steps_reset, use_special_gH_mobi, tol_j, h2_moeq;
  long long max_steps_per_cell;
  double max_runtime_allowed; // seconds of MODELLED reference CPU time (<= 0: guards off)
  double rt_cost_f, rt_cost_jac, rt_cost_lu; // modelled seconds per f evaluation / Jacobian / factorisation (racgpu_params)
  int n_record; // for params.t_max (record layout)
  int debug_max_calls; // developer aid (env RACGPU_DEBUG_TRACE): stop a cell after this many step calls; 0 = off
  int debug_dump_call; // developer builds (RG_DEBUG_NEWTON, env RACGPU_DEBUG_DUMP_CALL): the step call whose first corrector pass is dumped; -1 = none
  double elco[6][7];  // BDF coefficients el(i), i = 1..nq+1, per order nq = 1..5 (DCFODE, reference src/opkda1.f:146-171)
  double tesco[6][4]; // error-test constants tesco(1..3, nq)
};

struct DevWork { // per-cell workspace, all f64, cell-major
  double *rates;   // [ncell][nR]
  double *yh;      // [ncell][6][npad]
  double *P;       // [ncell][nnzJ]
  double *L;       // [ncell][nzl]
  double *U;       // [ncell][nzu]
  double *Dinv;    // [ncell][npad]
  double *rtol, *atol; // [ncell][npad]
  double *acor, *ewt; // [ncell][npad] accumulated correction and inverse error weights of the step in progress
  double *ygood;   // [ncell][npad] the last record whose T and H2 entries are not NaN (the hand-off record)
  double *Pb, *zb; // [ncell][npad] evolT: the T column of P and (species block)^-1 times it
  double *park;    // [nslots][128 + npad] parked cells (engine_integrate.hpp, struct Parked, then the iterate), or null
  int *counter;    // [0] work queue head; further words: see solve_pass
  double *trace;   // developer aid: [debug_max_calls][8] step log of cell 0, or null
  int *marker;     // developer aid: host-mapped progress word (null when off)
};

} // namespace racgpu
