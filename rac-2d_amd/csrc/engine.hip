// engine.hip -- kernels, device-table upload, workspace management and the extern "C" ABI (include/racgpu.h).
// gfx950 only.  The product path has no CPU fallback: every compute entry point fails loudly without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/racgpu.h"
#include "engine_integrate.hpp"
#include "hc_tables.hpp"
#include "network.hpp"

using namespace racgpu;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return -1; }
#define HIP_OK(x)                                                                                       \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_));     \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// kernels (one wave per workgroup, one cell per wave)
// ---------------------------------------------------------------------------------------------------------
struct LdsViews { double *y, *savf, *wx; }; // three LDS vectors per wave: 11.1 KB for 464 species, 12 waves per CU
__device__ __forceinline__ LdsViews carve(double *lds, int nlds) { return {lds, lds + nlds, lds + 2 * nlds}; }

__global__ __launch_bounds__(64) void k_rates(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, const double *cells, double *rates_out,
                                              double *cell_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  const int cell = blockIdx.x, lane = threadIdx.x;
  dev_rates(N, P, cells + (size_t)cell * RACGPU_NPAR, rates_out + (size_t)cell * N.nR, lane,
            cell_out ? cell_out + (size_t)cell * RACGPU_NOUT + RACGPU_O_R_H2_FORM : nullptr);
}

__global__ __launch_bounds__(64) void k_rhs(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, const double *cells, const double *yin, double *rates_ws, double *ydot_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, nlds = (N.nS + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  const double *cp = cells + (size_t)cell * RACGPU_NPAR;
  double *rates = rates_ws + (size_t)cell * N.nR;
  dev_rates(N, P, cp, rates, lane);
  for (int i = lane; i < N.nS; i += 64) v.y[i] = yin[(size_t)cell * N.nS + i];
  wave_sync();
  dev_rhs(N, rates, cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES], gptr(N.r_C), v.y, v.savf, lane);
  for (int i = lane; i < N.nS; i += 64) ydot_out[(size_t)cell * N.nS + i] = v.savf[i];
}

__global__ __launch_bounds__(64) void k_jac(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, const double *cells, const double *yin, double *rates_ws, double *vals_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, nlds = (N.nS + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  const double *cp = cells + (size_t)cell * RACGPU_NPAR;
  double *rates = rates_ws + (size_t)cell * N.nR;
  dev_rates(N, P, cp, rates, lane);
  for (int i = lane; i < N.nS; i += 64) v.y[i] = yin[(size_t)cell * N.nS + i];
  wave_sync();
  dev_build_P<false>(N, rates, cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES], v.y, 1.0, false, vals_out + (size_t)cell * N.nnzJ, lane);
}

__global__ __launch_bounds__(64) void k_newton(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, const double *cells, const double *yin, double gamma, double *bx,
                                               int repeat, long long *cyc_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, nlds = (N.nS + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  const double *cp = cells + (size_t)cell * RACGPU_NPAR;
  double *rates = W.rates + (size_t)cell * N.nR, *Pv = W.P + (size_t)cell * N.nnzJ, *Lv = W.L + (size_t)cell * N.nzl,
         *Uv = W.U + (size_t)cell * N.nzu, *Dinv = W.Dinv + (size_t)cell * N.npad;
  dev_rates(N, P, cp, rates, lane);
  for (int i = lane; i < N.nS; i += 64) v.y[i] = yin[(size_t)cell * N.nS + i];
  wave_sync();
  dev_build_P<true>(N, rates, cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES], v.y, -gamma, true, Pv, lane);
  // repeat > 1 (developer aid, RACGPU_DEBUG_REPEAT): the same factorisation and solve again and again, with the
  // cycle counts of the LU parts and of the solve written out per cell
  // (RACGPU_DEBUG_REPEAT_MODE, bits 24.. of repeat: 1 = only the factorisation is repeated, 2 = only the solve: per-phase traffic)
  long long cyc[4] = {0, 0, 0, 0}, c_lu = 0, c_solve = 0;
  const int mode = repeat >> 24;
  repeat &= 0xffffff;
  for (int r = 0; r < repeat; ++r) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (mode != 2 || r == 0) dev_lu(N, Pv, Lv, Uv, Dinv, v.wx, v.y, lane, cyc, v.wx + nlds);
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (mode != 1 || r == repeat - 1) {
      for (int i = lane; i < N.nS; i += 64) v.savf[i] = bx[(size_t)cell * N.nS + i];
      dev_solve(N, Lv, Uv, Dinv, v.savf, v.wx, lane);
    }
    c_lu += t1 - t0; c_solve += (long long)__builtin_readcyclecounter() - t1;
  }
  for (int i = lane; i < N.nS; i += 64) bx[(size_t)cell * N.nS + i] = v.savf[i];
  if (cyc_out && lane == 0) {
    long long *o = cyc_out + (size_t)cell * 8;
    o[0] = c_lu; o[1] = c_solve; o[2] = cyc[0]; o[3] = cyc[1]; o[4] = cyc[2]; o[5] = cyc[3];
  }
}

// Test hooks of the evolT path (host buffers): what = 0: chem_ode_f with T evolving (ydot incl. dT/dt) and the 29 heating/cooling
// values behind it; what = 1: additionally the finite-difference T row and T column of chem_ode_jac (con = 1: plain J entries).
__global__ __launch_bounds__(64) void k_evolT_hooks(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, const DevHC *hc_tab, const double *cells,
                                                    const double *hc, const double *yin /* [ncell][nS+1] */, int what, double *ydot_out /* [ncell][nS+1] */,
                                                    double *terms_out /* [ncell][HC_NTERMS] */, double *tcol_out /* [ncell][nS+1] */, double *trow_out /* [ncell][10] */) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, n = N.nS, nlds = (n + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  CellCtx c{};
  c.y = v.y; c.savf = v.savf; c.wx = v.wx; c.lane = lane; c.n = n; c.npad = N.npad; c.nteam = 1;
  c.rates = W.rates + (size_t)cell * N.nR; c.Pb = W.Pb + (size_t)cell * N.npad;
  c.cell = cells + (size_t)cell * RACGPU_NPAR; c.hcrec = hc + (size_t)cell * kNHC; c.hc = hc_tab; c.prm = &P;
  g_wc.nsite = c.cell[RACGPU_P_D2H] * c.cell[RACGPU_P_SITES];
  const double *yc = yin + (size_t)cell * (n + 1);
  for (int i = lane; i < n; i += 64) c.y[i] = yc[i];
  g_T.y = yc[n]; g_T.evolT = 1; g_T.freeze_rec = 0;
  wave_sync();
  dev_f<true>(N, c, c.savf);
  for (int i = lane; i < n; i += 64) ydot_out[(size_t)cell * (n + 1) + i] = c.savf[i];
  if (lane == 0) ydot_out[(size_t)cell * (n + 1) + n] = g_T.savf;
  // the terms once more, this time stored (same inputs: the rate vector is the one dev_f has just computed)
  dev_heating_cooling(N, *(const RG_GLOBAL DevHC *)hc_tab, c.cell, c.hcrec, c.y, g_T.y, c.rates, g_T.rh2, lane, terms_out + (size_t)cell * HC_NTERMS);
  if (what >= 1) { // (what == 2: the T row by ten full evaluations of the terms, to check the masks of the production path against)
    dev_T_border(N, c, 1.0, what == 2);
    for (int i = lane; i < n; i += 64) tcol_out[(size_t)cell * (n + 1) + i] = c.Pb[i];
    if (lane == 0) {
      tcol_out[(size_t)cell * (n + 1) + n] = g_T.Pd - 1.0;
      for (int k = 0; k < 10; ++k) trow_out[(size_t)cell * 10 + k] = g_T.Pc[k];
    }
  }
}

// The hot path.  Persistent: each wave pulls cells from a queue until it is empty; its workspace is per wave
// (slot), not per cell, so the HBM footprint is nslots * ~0.4 MB whatever the batch size.
struct SolveArgs {
  int ncell, flags;                 // flags: RACGPU_F_*
  const double *cells;              // [ncell][NPAR]
  double *yio;                      // [ncell][nS]
  const double *t0;                 // [ncell] or null (0)
  const int *tolj;                  // [ncell] or null (DevParams::tol_j)
  double *t_final; int *quality; long long *stats; double *record, *touts, *cell_out;
  const int *order;                 // queue order or null
  int slot0;                        // this launch's first workspace slot
  int park_max;                     // k_solve: cells may be parked for a team once the queue is empty and <= park_max waves are left (0: never)
  int *park_list, *park_count;      // slots holding parked cells, and how many
  // k_solve_columns: the cells column by column, top down; the shielding slots of a record are rewritten from what the cells
  // above it in its column ended with before the cell is solved (racgpu_column_sweep)
  const int *col_ptr, *col_cells;   // [ncolumn + 1], [ncell]
  int ncolumn, i_H2O, i_OH, i_CO;   // (0-based species, -1 absent)
  // 12CO shielding table of the caller (racgpu_set_co_shielding_table): axes ascending, f[ncol][nrow] with ln f stored; null: the CO
  // slot of the records stays as given
  const double *co_logNH2, *co_logNCO, *co_lnf; int co_nrow, co_ncol;
  const double *dz;                 // [ncell] path length through the cell towards the surface [cm]
  double dv_turb;                   // turbulent line width [cm/s] of the H2 self-shielding formula
  double *cells_rw;                 // = cells
  // rays to the star (racgpu_set_star_rays): the cell a cell's ray passes through next (-1: none), its path length through THAT cell
  // [cm] is ray_ds of that cell; ray_N[cell][4] = column densities of H2, H2O, OH, CO from the star to the far side of the cell,
  // ray_done[cell] = 1 once they are there (agent-scope release/acquire: the producer runs on another CU, possibly another XCD)
  const int *ray_inner; const double *ray_ds; double *ray_N; int *ray_done; int *ray_timeouts;
  // k_solve_T (gas temperature co-evolving): the cells' heating/cooling records [ncell][RACGPU_NHC] and the tables
  const double *hc; const DevHC *hc_tab;
};
constexpr int kParkWords = 128;     // doubles reserved per slot for struct Parked (in front of the parked iterate)
static_assert(sizeof(Parked) <= kParkWords * sizeof(double), "Parked outgrew its slot");

// TEAM = 1: one wave per workgroup and per cell (k_solve, the bulk of a batch).  TEAM = 4 (k_solve_team): four waves per cell for
// the few cells that would otherwise set the length of the pass on their own; wave 0 runs the integrator exactly as with TEAM =
// 1, the others wait at a barrier for the parts that are shared out (the factorisation, dev_lu's team mode) -- same arithmetic
// in the same order per column, same results to the last bit.
// what a column has above the cell being solved: column densities [cm^-2] of H2, H2O, OH (wave 0 of the team)
struct ColumnAcc { double N_H2, N_H2O, N_OH, N_CO; };
static __shared__ volatile ColumnAcc g_col;
static __shared__ volatile ColumnAcc g_ray; // the same towards the star, for the cell being solved
static __shared__ double g_rec[RACGPU_NPAR + kNHC]; // k_solve_T: the cell record and the heating/cooling record of the cell being solved

template <int TEAM, bool RESUME, bool COLUMN = false, bool ET = false>
RG_DEV void solve_body(const DevNet &N, const DevParams &P, const DevWork &W, const SolveArgs &A, double *lds) {
  static_assert(!ET || (!RESUME && !COLUMN), "evolT: one wave per cell, or a team from the start; no hand-over, no columns");
  const int lane = threadIdx.x & 63, wv = TEAM > 1 ? uniform_i((int)(threadIdx.x >> 6)) : 0;
  const int n = N.nS, nlds = (n + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  CellCtx c;
  c.nteam = TEAM;
  c.y = v.y; c.savf = v.savf; c.wx = v.wx;
  c.lane = lane; c.n = n; c.npad = N.npad;
  c.rates = nullptr; c.marker = nullptr;
  double *ygood = nullptr;
  auto bind = [&](int slot) { // the workspace slot the cell lives in (a resumed cell: the slot of the wave that parked it)
    c.acor = W.acor + (size_t)slot * N.npad; c.ewt = W.ewt + (size_t)slot * N.npad;
    c.yh = W.yh + (size_t)slot * 6 * N.npad; c.Pv = W.P + (size_t)slot * N.nnzJ;
    c.Lv = W.L + (size_t)slot * N.nzl; c.Uv = W.U + (size_t)slot * N.nzu; c.Dinv = W.Dinv + (size_t)slot * N.npad;
    c.rtol = W.rtol + (size_t)slot * N.npad; c.atol = W.atol + (size_t)slot * N.npad;
    ygood = W.ygood + (size_t)slot * N.npad;
    if constexpr (ET) { c.Pb = W.Pb + (size_t)slot * N.npad; c.zb = W.zb + (size_t)slot * N.npad; }
  };
  c.cell = nullptr; c.hcrec = nullptr; c.hc = A.hc_tab; c.Pb = nullptr; c.zb = nullptr; c.prm = &P;
  const int own_slot = A.slot0 + blockIdx.x;
  bind(own_slot);
  c.marker = own_slot == 0 ? W.marker : nullptr;
  if (TEAM > 1 && wv != 0) { // a helper wave: serve wave 0's requests until it has no more cells
    double *hw = lds + 3 * nlds + 64 + (wv - 1) * (nlds + 64); // work column + one spare double per lane
    for (;;) {
      team_barrier();
      const int cmd = g_team.cmd;
      if (cmd == T_EXIT) return;
      if (RESUME) bind(g_team.slot);
      if (cmd == T_LU) dev_lu(N, c.Pv, c.Lv, c.Uv, c.Dinv, hw, c.y, lane, nullptr, hw + nlds, wv, TEAM, &g_team.fail);
      if (cmd == T_JAC) {
        dev_build_P<true>(N, W.rates + (size_t)g_team.cell * N.nR, g_wc.nsite, c.y, g_team.con, true, c.Pv, lane, wv);
        team_barrier();
      }
    }
  }
  g_wc.inv_neq = 1.0 / (double)(n + 1);
  if (TEAM > 1 && !RESUME && lane == 0) atomicAdd(W.counter - 1, 1); // k_gate: this workgroup is resident (the team queue's counter is word [2], this is [1])
  const int nparked = RESUME ? gptr(A.park_count)[0] : 0;
  int kpos = 0, kend = 0; // COLUMN: the wave's place in its column's cell list
  for (;;) {
    int cell = 0;
    if (!COLUMN || kpos >= kend) { // the next work item: a cell, a parked cell or a column
      if (lane == 0) cell = atomicAdd(W.counter, 1);
      cell = uniform_i(cell);
    }
    dev_mark(c, 10 + cell);
    int slot = own_slot;
    Parked *pk = nullptr;
    if (COLUMN) {
      if (kpos >= kend) {
        if (cell >= A.ncolumn) break;
        kpos = A.col_ptr[cell]; kend = A.col_ptr[cell + 1]; // (never empty: racgpu_column_sweep checks)
        g_col.N_H2 = 0.0; g_col.N_H2O = 0.0; g_col.N_OH = 0.0; g_col.N_CO = 0.0;
      }
      cell = A.col_cells[kpos];
      if (lane == 0) { // (the surface cell too: column densities 0, as update_params_above_alt gives it, src/disk.f90:1840-1866)
        // update_params_above_alt's grid-free part (reference src/disk.f90:1840-1866): H2 by Draine & Bertoldi 1996 eq. 37
        // (get_H2_self_shielding, :1887-1897; the 0.035 is a single-precision literal there), H2O and OH by their Lyman-alpha cross
        // sections (src/sub_global_variables.f90:82-83), CO by get_12CO_shielding on the caller's table
        double *rec = A.cells_rw + (size_t)cell * RACGPU_NPAR;
        auto slots = [&](double N_H2, double N_H2O, double N_OH, double N_CO, int s_H2, int s_CO, int s_H2O, int s_OH) {
          const double x = N_H2 / 5e14, b5 = A.dv_turb / 1e5, tmp = sqrt(1.0 + x);
          const double den = 1.0 + x / b5;
          rec[s_H2] = fmin(1.0, 0.965 / (den * den) + (double)0.035f / tmp * exp(-8.5e-4 * tmp));
          rec[s_H2O] = fmin(1.0, exp(-(N_H2O * 1.2e-17)));
          rec[s_OH] = fmin(1.0, exp(-(N_OH * 1.8e-18)));
          if (A.co_lnf) { // get_12CO_shielding (reference src/load_Visser_CO_selfshielding.f90:271-309) on the caller's table
            const double xl = log10(fmax(N_CO, 1.0)), yl = log10(fmax(N_H2, 1.0));
            int i1 = 0, j1 = 0; // the enclosing table cell; the last one beyond the table, the first one below it
            for (int i = 0; i < A.co_nrow - 1; ++i) if (A.co_logNH2[i] < yl) i1 = i;
            for (int j = 0; j < A.co_ncol - 1; ++j) if (A.co_logNCO[j] < xl) j1 = j;
            const double x1 = A.co_logNCO[j1], x2 = A.co_logNCO[j1 + 1], y1 = A.co_logNH2[i1], y2 = A.co_logNH2[i1 + 1];
            const double z11 = A.co_lnf[(size_t)j1 * A.co_nrow + i1], z12 = A.co_lnf[(size_t)j1 * A.co_nrow + i1 + 1];
            const double z21 = A.co_lnf[(size_t)(j1 + 1) * A.co_nrow + i1], z22 = A.co_lnf[(size_t)(j1 + 1) * A.co_nrow + i1 + 1];
            const double k1 = (z12 - z11) / (y2 - y1), k2 = (z22 - z21) / (y2 - y1); // calc_four_point_linear_interpol, src/sub_trivials.f90:803-821
            const double v = ((k2 - k1) / (x2 - x1) * (xl - x1) + k1) * (yl - y1) + (z21 - z11) / (x2 - x1) * (xl - x1) + z11;
            rec[s_CO] = fmin(1.0, fmax(0.0, exp(v)));
          }
        };
        slots(g_col.N_H2, g_col.N_H2O, g_col.N_OH, g_col.N_CO, RACGPU_P_FSS_ISM_H2, RACGPU_P_FSS_ISM_CO, RACGPU_P_FSS_ISM_H2O, RACGPU_P_FSS_ISM_OH);
        if (A.ray_inner) { // the toStar slots (:1842-1866): what the cells along the ray to the star ended with; the cell waits for its neighbour
          double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
          const int in = A.ray_inner[cell];
          if (in >= 0) {
            const long long t0 = (long long)wall_clock64(), limit = 100000000LL * 1800; // (100 MHz; half an hour: a column starts one cell after its neighbour)
            bool there = true;
            while (__hip_atomic_load(&A.ray_done[in], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(127);
              if ((long long)wall_clock64() - t0 > limit) { there = false; atomicAdd(A.ray_timeouts, 1); break; }
            }
            if (there) {
              const double *rn = A.ray_N + (size_t)in * 4;
              r0 = __hip_atomic_load(rn + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r1 = __hip_atomic_load(rn + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              r2 = __hip_atomic_load(rn + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r3 = __hip_atomic_load(rn + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          g_ray.N_H2 = r0; g_ray.N_H2O = r1; g_ray.N_OH = r2; g_ray.N_CO = r3;
          slots(r0, r1, r2, r3, RACGPU_P_FSS_STAR_H2, RACGPU_P_FSS_STAR_CO, RACGPU_P_FSS_STAR_H2O, RACGPU_P_FSS_STAR_OH);
        }
      }
      ++kpos;
      __threadfence(); // the record as every lane (and the rate coefficients below) must see it
      dev_rates(N, P, A.cells + (size_t)cell * RACGPU_NPAR, W.rates + (size_t)cell * N.nR, lane,
                A.cell_out ? A.cell_out + (size_t)cell * RACGPU_NOUT + RACGPU_O_R_H2_FORM : nullptr);
      __threadfence();
    } else if (RESUME) {
      if (cell >= nparked) break;
      slot = A.park_list[cell];
      bind(slot);
      g_team.slot = slot;
      pk = reinterpret_cast<Parked *>(W.park + (size_t)slot * (N.npad + kParkWords));
      cell = pk->cell;
    } else {
      if (cell >= A.ncell) break;
      if (A.order) cell = A.order[cell]; // longest-expected-first schedule (racgpu_set_cost_hints)
      if (W.park) pk = reinterpret_cast<Parked *>(W.park + (size_t)slot * (N.npad + kParkWords));
    }
    double *ypark = W.park ? W.park + (size_t)slot * (N.npad + kParkWords) + kParkWords : nullptr;
    const double *cp = A.cells + (size_t)cell * RACGPU_NPAR;
    long long cyc0 = dev_clock();
    // The caller's per-cell time window (set_initial_condition_4solver / _continue, reference src/disk.f90:2075-2097,
    // 2128-2144): t_max of the cell, start time t0, first output step max(dt0, 1e-3 t0), and n_record recomputed from
    // them by chem_evol_solve_prepare_ongoing (src/chemistry.f90:1916-1938) for every cell and every local iteration.
    const double t_max = cp[RACGPU_P_TMAX] > 0.0 ? cp[RACGPU_P_TMAX] : P.t_max;
    const double t0 = A.t0 ? A.t0[cell] : 0.0;
    const double dt_first = fmax(P.dt_first_step, t0 * 1e-3);
    const bool runnable = t_max > t0;
    const int n_record = runnable ? (int)ceil(log((t_max - t0) / dt_first * (P.ratio_tstep - 1.0) + 1.0) / log(P.ratio_tstep)) + 1 : 1;
    c.rates = W.rates + (size_t)cell * N.nR; // filled by k_rates for the whole batch just before this launch
    if (TEAM > 1) g_team.cell = cell;
    if (!RESUME) {
      for (int k = 0; k < 8; ++k) g_wc.cyc[k] = 0;
      g_wc.Tgas = cp[RACGPU_P_TGAS]; g_wc.nsite = cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES];
      dev_mark(c, 1);
      { double rT, aT; dev_tolerances(N, P, A.tolj ? A.tolj[cell] : P.tol_j, cp[RACGPU_P_D2H], c.rtol, c.atol, rT, aT, lane); g_wc.rT = rT; g_wc.aT = aT; }
      if constexpr (ET) { // set_initial_condition_4solver (src/disk.f90:2066-2073): y(NEQ) = Tgas; T evolves when the cell gains energy
        // the two records go to LDS once per cell: f(y) reads ~100 of their fields one after the other (dev_rates' header, the 28 terms), and
        // every one was a dependent global load of 1-2 us under the kernel's own traffic
        if (lane < RACGPU_NPAR) g_rec[lane] = cp[lane];
        if (lane < kNHC) g_rec[RACGPU_NPAR + lane] = A.hc[(size_t)cell * kNHC + lane];
        wave_sync();
        c.cell = const_cast<const double *>(g_rec); c.hcrec = const_cast<const double *>(g_rec) + RACGPU_NPAR;
        const int j = A.tolj ? A.tolj[cell] : P.tol_j;
        g_T.y = cp[RACGPU_P_TGAS]; g_T.savf = 0.0; g_T.acor = 0.0; g_T.ewt = 0.0; g_T.rtol = g_wc.rT; g_T.atol = g_wc.aT;
        for (int k = 0; k < 6; ++k) g_T.yh[k] = 0.0;
        for (int k = 0; k < 10; ++k) g_T.Pc[k] = 0.0;
        g_T.Pd = 1.0; g_T.schur = 1.0; g_T.rh2 = 0.0;
        g_T.t_scale_tol = j == 1 ? 1e-6 : j == 2 ? 1e-4 : j == 3 ? 1e-3 : j == 4 ? 1e-2 : 1e-1; // chem_set_solver_flags_alt (src/chemistry.f90:220-244)
        g_T.evolT = c.hcrec[H_EN_GAIN_TOT] > 0.0 ? 1 : 0; g_T.freeze_rec = 0;
        g_T.maySwitchT = ((const RG_GLOBAL DevHC *)A.hc_tab)->cfg.may_switch_T;
      }
      dev_mark(c, 2);
      for (int i = lane; i < n; i += 64) c.y[i] = A.yio[(size_t)cell * n + i];
      wave_sync();
      if (A.flags & RACGPU_F_RECTIFY) { // rectify_abundances (src/chemistry.f90:2170-2201): E- takes up the net charge
        double q = 0.0;
        for (int i = lane; i < n; i += 64) q += c.y[i] * (double)gptr(N.s_charge)[i];
        q = wave_sum(q);
        if (lane == 0 && N.i_E >= 0) c.y[N.i_E] = c.y[N.i_E] + q;
        wave_sync();
      }
    } else { // the per-cell constants, the counters and the iterate as the parking wave left them
      const WaveConst &wc = pk->wc;
      g_wc.nsite = wc.nsite; g_wc.Tgas = wc.Tgas; g_wc.rT = wc.rT; g_wc.aT = wc.aT;
      for (int k = 0; k < 8; ++k) g_wc.cyc[k] = wc.cyc[k];
      cyc0 -= pk->elapsed;
      for (int i = lane; i < n; i += 64) c.y[i] = ypark[i];
      wave_sync();
    }
    double *rec = A.record ? A.record + (size_t)cell * P.n_record * (n + 1) : nullptr;
    double *tos = A.touts ? A.touts + (size_t)cell * P.n_record : nullptr;
    const int nrec = A.record || A.touts ? min(n_record, P.n_record) : n_record;
    ParkIO io{};
    io.rec = pk; io.resume = RESUME; io.counters = W.counter; io.ncell = A.ncell; io.nwaves = (int)gridDim.x;
    io.park_max = (TEAM == 1 && pk) ? A.park_max : 0;
    io.ypark = ypark; io.cell = cell; io.slot = slot; io.cyc0 = cyc0; io.park_list = A.park_list; io.park_count = A.park_count;
    CellResult R = dev_evol_solve<ET>(N, P, c, t0, t_max, dt_first, nrec, rec, tos, ygood, cell == 0 ? W.trace : nullptr, io);
    dev_mark(c, 4);
    wave_sync();
    // hand-off (src/disk.f90:1716-1733): record(:, isav), touts(isav); with isav <= 1 ("No useful data produced") the
    // caller's abundances and t_final stay as they were
    const bool useful = R.isav > 1; // (a parked cell -- TEAM == 1 only -- writes nothing here: isav = 0, and the queue is empty)
    double nmol = 0.0;
    if (useful) {
      for (int i = lane; i < n; i += 64) {
        const double yi = ygood[i];
        A.yio[(size_t)cell * n + i] = yi;
        if (gptr(N.s_tolclass)[i] == 3) nmol += yi; // get_ice_coverage's side effect (src/chemistry.f90:989-1003)
      }
      nmol = wave_sum(nmol) / cp[RACGPU_P_D2H];
    }
    if (lane == 0 && !R.parked) {
      if (A.t_final) A.t_final[cell] = useful ? R.t_good : t0;
      if (A.quality) A.quality[cell] = R.quality;
      if (A.cell_out) {
        double *o = A.cell_out + (size_t)cell * RACGPU_NOUT;
        if (useful) o[RACGPU_O_N_MOL_ON_GRAIN] = nmol;
        o[RACGPU_O_T_END] = R.t_final;
        if constexpr (ET) { // c%par%Tgas = record(nS+1, isav) (src/disk.f90:1732); the coefficient of the last chem_cal_rates call
          if (useful) o[RACGPU_O_TGAS] = R.T_good;
          o[RACGPU_O_EVOLT_END] = (double)R.evolT_end; o[RACGPU_O_TFREEZE_REC] = (double)R.freeze_rec;
          if (N.r_h2form >= 0) o[RACGPU_O_R_H2_FORM] = g_T.rh2;
        } else { if (useful) o[RACGPU_O_TGAS] = cp[RACGPU_P_TGAS]; o[RACGPU_O_EVOLT_END] = 0.0; o[RACGPU_O_TFREEZE_REC] = 0.0; }
      }
      if (A.stats) {
        long long *s = A.stats + (size_t)cell * RACGPU_NSTAT;
        s[RACGPU_S_NST] = R.nst; s[RACGPU_S_NFE] = R.nfe; s[RACGPU_S_NJE] = R.nje; s[RACGPU_S_NLU] = R.nlu;
        s[RACGPU_S_NERR] = R.nerr; s[RACGPU_S_NREC_REAL] = R.nrec_real; s[RACGPU_S_QSUM] = R.qsum; s[RACGPU_S_NCFAIL_ETFAIL] = R.nfail;
        s[RACGPU_S_CYC_TOTAL] = dev_clock() - cyc0; s[RACGPU_S_CYC_RHS] = g_wc.cyc[CYC_RHS]; s[RACGPU_S_CYC_JAC] = g_wc.cyc[CYC_JAC];
        s[RACGPU_S_CYC_LU] = g_wc.cyc[CYC_LU]; s[RACGPU_S_CYC_SOLVE] = g_wc.cyc[CYC_SOLVE];
        s[13] = g_wc.cyc[CYC_LU_PART]; s[14] = g_wc.cyc[CYC_LU_PART + 1]; s[15] = g_wc.cyc[CYC_LU_PART + 2]; // finish = LU - the three
        s[RACGPU_S_ISAV] = R.isav; s[RACGPU_S_NITER] = 1; s[RACGPU_S_NREC] = nrec; s[RACGPU_S_ERRCODES] = R.errc;
      }
    }
    if (COLUMN) { // what this cell adds to the columns above the next one: its hand-off abundances (as the caller now has them)
      __threadfence();
      if (lane == 0) {
        const double w = cp[RACGPU_P_NGAS] * A.dz[cell];
        const double *yc = A.yio + (size_t)cell * n;
        if (N.i_H2 >= 0) g_col.N_H2 = g_col.N_H2 + w * yc[N.i_H2];
        if (A.i_H2O >= 0) g_col.N_H2O = g_col.N_H2O + w * yc[A.i_H2O];
        if (A.i_OH >= 0) g_col.N_OH = g_col.N_OH + w * yc[A.i_OH];
        if (A.i_CO >= 0) g_col.N_CO = g_col.N_CO + w * yc[A.i_CO];
        if (A.ray_inner) { // ... and to the rays that leave the star through it
          const double ws = cp[RACGPU_P_NGAS] * A.ray_ds[cell];
          double *rn = A.ray_N + (size_t)cell * 4;
          __hip_atomic_store(rn + 0, g_ray.N_H2 + (N.i_H2 >= 0 ? ws * yc[N.i_H2] : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(rn + 1, g_ray.N_H2O + (A.i_H2O >= 0 ? ws * yc[A.i_H2O] : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(rn + 2, g_ray.N_OH + (A.i_OH >= 0 ? ws * yc[A.i_OH] : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(rn + 3, g_ray.N_CO + (A.i_CO >= 0 ? ws * yc[A.i_CO] : 0.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&A.ray_done[cell], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    dev_mark(c, 6);
  }
  dev_mark(c, 7);
  if (TEAM == 1 && lane == 0) atomicAdd(W.counter + 3, 1); // one wave fewer (the parking rule counts them)
  if (TEAM > 1) { g_team.cmd = T_EXIT; team_barrier(); }
}

__global__ __launch_bounds__(64) void k_solve(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, SolveArgs A) {
  extern __shared__ double lds[];
  solve_body<1, false>(*Np, *Pp, W, A, lds);
}

__global__ __launch_bounds__(64 * kTeam) void k_solve_team(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, SolveArgs A) {
  extern __shared__ double lds[];
  solve_body<kTeam, false>(*Np, *Pp, W, A, lds);
}
// gas temperature co-evolving with the chemistry (chemsol_params%evolT): one wave per cell
__global__ __launch_bounds__(64) void k_solve_T(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, SolveArgs A) {
  extern __shared__ double lds[];
  solve_body<1, false, false, true>(*Np, *Pp, W, A, lds);
}
// the same for the cells the cost hints single out: four waves (factorisation and Jacobian shared out; f, the 28 terms and the border stay on wave 0)
__global__ __launch_bounds__(64 * kTeam) void k_solve_team_T(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, SolveArgs A) {
  extern __shared__ double lds[];
  solve_body<kTeam, false, false, true>(*Np, *Pp, W, A, lds);
}
// columns of cells in dependency order, one team per column at a time (racgpu_column_sweep)
__global__ __launch_bounds__(64 * kTeam) void k_solve_columns(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, SolveArgs A) {
  extern __shared__ double lds[];
  solve_body<kTeam, false, true>(*Np, *Pp, W, A, lds);
}
// the cells k_solve parked, each taken up by a team in the workspace slot it was parked in
__global__ __launch_bounds__(64 * kTeam) void k_solve_team_resume(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, SolveArgs A) {
  extern __shared__ double lds[];
  solve_body<kTeam, true>(*Np, *Pp, W, A, lds);
}

// Holds the stream that launches the bulk kernel until the team kernel's workgroups have started (or a bound of ~20 ms has passed:
// the gate can delay, never hang).  The bulk kernel's persistent waves fill every wave slot of the chip and keep it until the queue is
// empty; team workgroups dispatched after them would only start once the pass is all but over.
__global__ void k_gate(const int *started, int want, long long max_cycles) {
  const long long t0 = (long long)__builtin_readcyclecounter();
  while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want &&
         (long long)__builtin_readcyclecounter() - t0 < max_cycles)
    __builtin_amdgcn_s_sleep(64);
}

// ---- the caller's retry loop over local iterations (calc_this_cell, reference src/disk.f90:1651-1791), batched -----
// gather the cells that go into local iteration j: compact copies of their records, abundances and start times
__global__ void k_gather_pass(int nsel, const int *__restrict__ sel, int nS, const double *__restrict__ cells, const double *__restrict__ y,
                              const double *__restrict__ t_final, double *__restrict__ cells_c, double *__restrict__ y_c, double *__restrict__ t0_c) {
  const int f = blockIdx.x, cell = sel[f];
  for (int i = threadIdx.x; i < RACGPU_NPAR; i += blockDim.x) cells_c[(size_t)f * RACGPU_NPAR + i] = cells[(size_t)cell * RACGPU_NPAR + i];
  for (int i = threadIdx.x; i < nS; i += blockDim.x) y_c[(size_t)f * nS + i] = y[(size_t)cell * nS + i];
  if (threadIdx.x == 0) t0_c[f] = t_final[cell];
}

// merge what local iteration j > 1 produced for the selected cells, in the order calc_this_cell applies it:
//   touts(n_record_real) <= t_final of the previous iteration -> "does not proceed": nothing is taken over;
//   quality is taken over; isav <= 1 -> "No useful data": abundances and t_final stay; else both are taken over.
// The work counters add up over the iterations; NERR, n_record_real, isav, n_record are those of the last iteration that proceeded.
__global__ void k_merge_pass(int nsel, const int *__restrict__ sel, int nS, int j, const double *__restrict__ y_c, const double *__restrict__ t_final_c,
                             const int *__restrict__ quality_c, const long long *__restrict__ stats_c, const double *__restrict__ out_c,
                             double *__restrict__ y, double *__restrict__ t_final, int *__restrict__ quality, long long *__restrict__ stats,
                             double *__restrict__ cell_out) {
  const int f = blockIdx.x, cell = sel[f];
  const long long *sc = stats_c + (size_t)f * RACGPU_NSTAT;
  long long *s = stats + (size_t)cell * RACGPU_NSTAT;
  const bool proceeds = out_c[(size_t)f * RACGPU_NOUT + RACGPU_O_T_END] > t_final[cell];
  const bool useful = sc[RACGPU_S_ISAV] > 1;
  __syncthreads(); // every thread has read t_final[cell] before thread 0 may overwrite it
  if (threadIdx.x == 0) {
    static const int kAdd[] = {RACGPU_S_NST, RACGPU_S_NFE, RACGPU_S_NJE, RACGPU_S_NLU, RACGPU_S_QSUM, RACGPU_S_NCFAIL_ETFAIL, RACGPU_S_CYC_TOTAL,
                               RACGPU_S_CYC_RHS, RACGPU_S_CYC_JAC, RACGPU_S_CYC_LU, RACGPU_S_CYC_SOLVE, 13, 14, 15};
    for (int k : kAdd) s[k] += sc[k];
    if (proceeds) { // (an iteration that does not proceed leaves NITER behind: the host loop stops the cell on that)
      s[RACGPU_S_NITER] = j;
      s[RACGPU_S_NERR] = sc[RACGPU_S_NERR]; s[RACGPU_S_NREC_REAL] = sc[RACGPU_S_NREC_REAL]; s[RACGPU_S_ISAV] = sc[RACGPU_S_ISAV];
      s[RACGPU_S_NREC] = sc[RACGPU_S_NREC]; s[RACGPU_S_ERRCODES] = sc[RACGPU_S_ERRCODES];
      if (cell_out) cell_out[(size_t)cell * RACGPU_NOUT + RACGPU_O_T_END] = out_c[(size_t)f * RACGPU_NOUT + RACGPU_O_T_END];
      quality[cell] = quality_c[f];
      if (useful) {
        t_final[cell] = t_final_c[f];
        if (cell_out) cell_out[(size_t)cell * RACGPU_NOUT + RACGPU_O_N_MOL_ON_GRAIN] = out_c[(size_t)f * RACGPU_NOUT + RACGPU_O_N_MOL_ON_GRAIN];
      }
    }
  }
  if (proceeds && useful)
    for (int i = threadIdx.x; i < nS; i += blockDim.x) y[(size_t)cell * nS + i] = y_c[(size_t)f * nS + i];
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct racgpu_network {
  HostNetwork net;
  bool uploaded = false;
  DevNet dn{};
  DevNet *dn_dev = nullptr;      // device copy of dn (kernels read the tables through a pointer, not kernargs)
  DevParams *dp_dev = nullptr;   // device copy of the last parameter set
  std::vector<void *> dev_allocs;
  hipStream_t stream = nullptr;
  // workspace
  DevWork ws{};
  std::vector<void *> ws_allocs;
  long ws_slots = 0, ws_rate_cells = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<double> cost_hints; // racgpu_set_cost_hints
  int *order_dev = nullptr; long order_cap = 0;
  std::vector<int> order_host;    // the queue order of the last hinted pass
  double team_frac = 0.5;         // racgpu_set_team_threshold
  bool park_enabled = true;       // racgpu_set_team_threshold: frac < 0 also switches the hand-over at the end of a pass off
  hipStream_t team_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  long last_team_cells = 0;       // cells the last pass solved four waves at a time
  int *parked_host = nullptr;     // pinned: cells the last pass (its last chunk) handed over to teams at its end
  double *co_logNH2 = nullptr, *co_logNCO = nullptr, *co_lnf = nullptr; int co_nrow = 0, co_ncol = 0; // racgpu_set_co_shielding_table
  std::vector<int> ray_inner; std::vector<double> ray_ds; // racgpu_set_star_rays (host copies: the sweep checks them against its columns)
  int last_ray_timeouts = 0;
  bool timed = false;
  int cu_count = 0;
  // heating/cooling (evolT): host tables, their device copy, and the arrays it points at
  HostHC hhc; HcConfig hcfg{}; DevHC *hc_dev = nullptr; std::vector<void *> hc_allocs;
  void upload_hc();

  template <typename T>
  const T *up(const std::vector<T> &h) {
    void *d = nullptr;
    HIP_OK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    dev_allocs.push_back(d);
    return (const T *)d;
  }
  void upload();
  void set_ref_layout() { // Z of an ISTATE = 3 call: LREQ - 4 NNZ0 - 28 NEQ + NNZ1 + NEQ * NCOLM with LREQ = LENRW - 20 - 9 NEQ
    const int neq = net.nS + 1;
    dn.ref_nnz1 = net.ref_nnz1; dn.ref_clobber = net.ref_lenrw > 0 ? 1 : 0;
    dn.ref_zbase = (net.ref_lenrw - 20 - 9 * neq) - 4 * net.ref_nnz0 - 28 * neq + net.ref_nnz1;
  }
  void ensure_workspace(long slots, long rate_cells);
  void free_ws() { for (void *p : ws_allocs) (void)hipFree(p); ws_allocs.clear(); ws_slots = 0; ws_rate_cells = 0; }
  ~racgpu_network() {
    free_ws();
    if (order_dev) (void)hipFree(order_dev);
    for (void *p : dev_allocs) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (team_stream) (void)hipStreamDestroy(team_stream);
    if (parked_host) (void)hipHostFree(parked_host);
    for (double *q : {co_logNH2, co_logNCO, co_lnf}) if (q) (void)hipFree(q);
    for (void *q : hc_allocs) (void)hipFree(q);
    if (hc_dev) (void)hipFree(hc_dev);
  }
};

static void require_gpu() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    throw std::runtime_error("no HIP device visible: the racgpu compute path has no CPU fallback");
}

void racgpu_network::upload() {
  if (uploaded) return;
  require_gpu();
  hipDeviceProp_t prop;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(hipGetDeviceProperties(&prop, dev));
  cu_count = prop.multiProcessorCount;
  const HostNetwork &h = net;
  const int nS = h.nS, nR = h.nR;
  dn.nS = nS; dn.nR = nR; dn.npad = (nS + 63) / 64 * 64;
  dn.nnzJ = (int)h.Jrow.size(); dn.nzl = h.sym.nzl; dn.nzu = h.sym.nzu; dn.ns = h.sym.ns;
  std::vector<int16_t> itype(nR);
  std::vector<uint16_t> re0(nR), re1(nR), id3(nR, 0);
  std::vector<uint8_t> nreac(nR), fss(nR), flags(nR, 0);
  std::vector<double> A(nR), B(nR), C(nR), T0(nR), T1(nR);
  std::vector<uint64_t> w0(nR), w1(nR), w2(nR);
  const int iH2 = h.idx10[0], igH = h.i_gH;
  for (int r = 0; r < nR; ++r) {
    const Reaction &x = h.R[r];
    itype[r] = (int16_t)x.itype;
    re0[r] = x.reac[0] > 0 ? (uint16_t)(x.reac[0] - 1) : 0xffff;
    re1[r] = x.reac[1] > 0 ? (uint16_t)(x.reac[1] - 1) : 0xffff;
    nreac[r] = (uint8_t)x.n_reac;
    fss[r] = (uint8_t)h.fss_selector(r);
    if (x.rname[0] == "H2" && iH2 > 0) flags[r] |= 1;
    if (x.rname[0] == "gH" && igH > 0) flags[r] |= 2;
    if (x.itype == 21) {
      const int g1 = h.elements[x.reac[0] - 1][2];
      id3[r] = (uint16_t)((g1 == 0 ? x.reac[0] : x.reac[1]) - 1);
      if (h.elements[x.reac[0] - 1][0] * h.elements[x.reac[1] - 1][0] == -1) flags[r] |= 4;
    }
    A[r] = x.ABC[0]; B[r] = x.ABC[1]; C[r] = x.ABC[2]; T0[r] = x.Trange[0]; T1[r] = x.Trange[1];
    const Kind k = h.kind(r);
    const uint64_t a = k == K_NONE ? 0 : (uint64_t)(x.reac[0] - 1), b = (k == K_TWO) ? (uint64_t)(x.reac[1] - 1) : a;
    w0[r] = (uint64_t)k | ((uint64_t)x.n_reac << 8) | (a << 16) | (b << 32);
    // unused slots name the spare double of the lane that handles this row (64 of them behind the third LDS vector: index
    // 2 * nlds + lane seen from ydot, the second vector): the kernel adds to all seven slots without testing
    uint16_t tg[8];
    for (int s = 0; s < 8; ++s) tg[s] = (uint16_t)(2 * ((nS + 1) & ~1) + (r % 64));
    if (k != K_NONE) {
      // slots 0..n_reac-1 are reactants (subtract), then products (add); unused slots stay 0xffff.
      for (int s = 0; s < x.n_reac; ++s) tg[s] = (uint16_t)(x.reac[s] - 1);
      for (int s = 0; s < x.n_prod; ++s) tg[x.n_reac + s] = (uint16_t)(x.prod[s] - 1);
    }
    w1[r] = (uint64_t)tg[0] | ((uint64_t)tg[1] << 16) | ((uint64_t)tg[2] << 32) | ((uint64_t)tg[3] << 48);
    w2[r] = (uint64_t)tg[4] | ((uint64_t)tg[5] << 16) | ((uint64_t)tg[6] << 32) | ((uint64_t)0xffff << 48);
    if (2 * ((nS + 1) & ~1) + 63 >= 0xffff) throw std::runtime_error("RHS rows: too many species for 16-bit target slots");
  }
  dn.r_itype = up(itype); dn.r_re0 = up(re0); dn.r_re1 = up(re1); dn.r_nreac = up(nreac); dn.r_fss = up(fss);
  dn.r_flags = up(flags); dn.r_id3 = up(id3);
  dn.r_A = up(A); dn.r_B = up(B); dn.r_C = up(C); dn.r_T0 = up(T0); dn.r_T1 = up(T1);
  {
    std::vector<double> ub, uc;
    std::vector<uint32_t> ibc(nR, 0);
    auto index_of = [](std::vector<double> &u, double v) { // (bit pattern: -0.0 and 0.0 stay apart, as pow and exp see them)
      for (size_t q = 0; q < u.size(); ++q) if (std::memcmp(&u[q], &v, sizeof v) == 0) return (uint32_t)q;
      u.push_back(v); return (uint32_t)(u.size() - 1);
    };
    for (int r = 0; r < nR; ++r) if (itype[r] == 5 || itype[r] == 6) { const uint32_t ib = index_of(ub, B[r]), ic = index_of(uc, C[r]); ibc[r] = ib | (ic << 16); }
    dn.n_ub = (int)ub.size(); dn.n_uc = (int)uc.size();
    if (dn.n_ub + dn.n_uc > kRateTab || dn.n_ub > 0xffff || dn.n_uc > 0xffff) { dn.n_ub = dn.n_uc = 0; } // too many distinct values for the LDS table: direct evaluation
    ub.resize(ub.size() + 64, 0.0); uc.resize(uc.size() + 64, 0.0);
    dn.r_ub = up(ub); dn.r_uc = up(uc); dn.r_ibc = up(ibc);
  }
  dn.s_mass = up(h.mass_num); dn.s_vib = up(h.vib_freq); dn.s_Edes = up(h.Edesorb);
  {
    std::vector<int> dl(h.dupli_list);
    for (int &v : dl) v -= 1;
    dn.dupli_ptr = up(h.dupli_ptr); dn.dupli_list = up(dl);
  }
  for (int r = nR; r < nR + 448; ++r) { // padding: the RHS runs to a multiple of 192 rows and prefetches 128 ahead (kind 0: flux = k * y[0], every slot spare)
    const uint64_t sp = (uint64_t)(2 * ((nS + 1) & ~1) + (r % 64));
    w0.push_back(0); w1.push_back(sp | (sp << 16) | (sp << 32) | (sp << 48)); w2.push_back(sp | (sp << 16) | (sp << 32) | ((uint64_t)0xffff << 48));
  }
  dn.rhs_w0 = up(w0); dn.rhs_w1 = up(w1); dn.rhs_w2 = up(w2);
  // Jacobian gather: entries sorted by decreasing term count so that the 64 lanes of a pass do similar work
  {
    const int nnz = dn.nnzJ;
    std::vector<int> order(nnz);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return h.term_ptr[a + 1] - h.term_ptr[a] > h.term_ptr[b + 1] - h.term_ptr[b];
    });
    order.resize((size_t)(nnz + 63) / 64 * 64, -1);
    std::vector<uint64_t> tw(h.terms.size());
    for (size_t t = 0; t < tw.size(); ++t) {
      const JacTerm &x = h.terms[t];
      tw[t] = (uint64_t)x.rxn | ((uint64_t)x.other << 16) | ((uint64_t)x.kind << 32) | ((uint64_t)x.flags << 40) | ((uint64_t)x.other2 << 48);
    }
    std::vector<uint8_t> isd(nnz, 0);
    for (int j = 0; j < nS; ++j)
      for (int q = h.Jcolptr[j]; q < h.Jcolptr[j + 1]; ++q) if (h.Jrow[q] == j) isd[q] = 1;
    // Term stream for the Jacobian gather: pass p handles the 64 entries order[64p .. 64p+63], one per lane; row i of
    // the pass holds term i of every lane's entry (null where an entry has fewer terms), so the kernel reads the terms
    // as one linear, coalesced, prefetchable stream.  rowflag marks the last row of every pass; slot words say where
    // each lane's sum goes: entry | position in the permuted P storage << 24 | diagonal << 48 | valid << 49.
    {
      const int npass = (int)order.size() / 64;
      std::vector<uint64_t> stream, slot((size_t)(npass + 1) * 64, 0ull);
      std::vector<uint32_t> rowflag;
      // k_solve_team shares the stream out over its waves: kTeam segments of whole passes with about the same number of rows; a
      // segment starts on a multiple of kJacUnroll rows (the pass before it is padded with null rows, its "last row" flag moves)
      long rows_total = 0;
      std::vector<int> pass_rows(npass, 1);
      for (int p = 0; p < npass; ++p) {
        for (int l = 0; l < 64; ++l) {
          const int e = order[(size_t)p * 64 + l];
          if (e >= 0) pass_rows[p] = std::max(pass_rows[p], h.term_ptr[e + 1] - h.term_ptr[e]);
        }
        rows_total += pass_rows[p];
      }
      int seg = 1; long rows_done = 0;
      dn.jac_seg_row[0] = 0; dn.jac_seg_pass[0] = 0;
      for (int p = 0; p < npass; ++p) {
        if (seg < kTeam && rows_done >= rows_total * seg / kTeam) { // pass p opens segment seg
          while (rowflag.size() % kJacUnroll) {
            rowflag.back() = 0u; rowflag.push_back(1u); stream.insert(stream.end(), 64, ~0ull);
          }
          dn.jac_seg_row[seg] = (int)rowflag.size(); dn.jac_seg_pass[seg] = p; ++seg;
        }
        rows_done += pass_rows[p];
        int niter = 1;
        for (int l = 0; l < 64; ++l) {
          const int e = order[(size_t)p * 64 + l];
          if (e < 0) continue;
          niter = std::max(niter, h.term_ptr[e + 1] - h.term_ptr[e]);
          slot[(size_t)p * 64 + l] = (uint64_t)e | ((uint64_t)h.sym.Ppos[e] << 24) | ((uint64_t)(isd[e] ? 1 : 0) << 48) | (1ull << 49);
        }
        for (int i = 0; i < niter; ++i) {
          for (int l = 0; l < 64; ++l) {
            const int e = order[(size_t)p * 64 + l];
            const bool has = e >= 0 && i < h.term_ptr[e + 1] - h.term_ptr[e];
            stream.push_back(has ? tw[h.term_ptr[e] + i] : ~0ull);
          }
          rowflag.push_back(i == niter - 1 ? 1u : 0u);
        }
      }
      while (rowflag.size() % kJacUnroll) { rowflag.push_back(0u); stream.insert(stream.end(), 64, ~0ull); }
      dn.jac_rows = (int)rowflag.size();
      for (; seg <= kTeam; ++seg) { dn.jac_seg_row[seg] = dn.jac_rows; dn.jac_seg_pass[seg] = npass; } // (fewer passes than waves: empty segments)
      stream.insert(stream.end(), (size_t)64 * (kJacUnroll + 8), ~0ull); // the kernel prefetches rows past the end
      rowflag.resize((rowflag.size() + 63) / 64 * 64 + 64, 0u);
      dn.jac_stream = up(stream); dn.jac_rowflag = up(rowflag); dn.jac_slot = up(slot);
    }
  }
  {
    const Symbolic &S = h.sym;
    std::vector<uint16_t> perm(S.perm.begin(), S.perm.end()), Lrow(S.Lrow.begin(), S.Lrow.end()), Urow(S.Urow.begin(), S.Urow.end()),
        Prow(S.Prow.begin(), S.Prow.end());
    Lrow.resize(Lrow.size() + 64, 0); Urow.resize(Urow.size() + 64, 0); // the LU prefetch reads up to 64 entries past a column
    // (RG_LU_OPS: the 64 entries behind L's row list name the spare LDS doubles behind the work column, one per lane)
    for (int q = 0; q < 64; ++q) Lrow[S.Lrow.size() + q] = (uint16_t)(((nS + 1) & ~1) + q);
    Prow.resize(Prow.size() + 64, 0);
    dn.perm = up(perm); dn.Lrow = up(Lrow); dn.Urow = up(Urow); dn.Prow = up(Prow);
    {
      std::vector<uint8_t> pd(S.Psrc.size(), 0);
      for (int j = 0; j < nS; ++j)
        for (int q = h.Jcolptr[j]; q < h.Jcolptr[j + 1]; ++q) if (h.Jrow[q] == j) pd[S.Ppos[q]] = 1;
      pd.resize(pd.size() + 64, 0); // read in whole blocks of 64
      dn.Pdiag = up(pd);
      std::vector<uint16_t> kr(S.Psrc.size() + 64, 0);
      if (h.ref_nnz1 > 65535) throw std::runtime_error("reference pattern too large for 16-bit storage positions");
      for (size_t q = 0; q < S.Psrc.size(); ++q) kr[q] = (uint16_t)h.ref_kref[S.Psrc[q]];
      dn.Pkref = up(kr);
    }
    auto pack = [](const std::vector<int> &row, const std::vector<int> &col, const std::vector<int> &lev, size_t nstream, int &nchunk) {
      // the streamed part only (the dense trailing block is solved in registers); the storage is level-aligned, so
      // every chunk of 64 entries has ONE level; bit 20 of every word of a chunk: the next chunk continues this level
      std::vector<uint32_t> rc(nstream);
      for (size_t e = 0; e < nstream; ++e) rc[e] = (uint32_t)row[e] | ((uint32_t)col[e] << 10);
      rc.resize((nstream + 63) / 64 * 64, 0u); // row == col == 0: skipped
      size_t nch = rc.size() / 64;
      auto chunk_level = [&](size_t c) { return lev[std::min(c * 64, nstream - 1)]; };
      for (size_t c = 0; c < nch; ++c) {
        for (size_t e = c * 64; e < std::min((c + 1) * 64, nstream); ++e)
          if (lev[e] != chunk_level(c)) throw std::runtime_error("solve schedule: a chunk spans two levels");
        if (c + 1 < nch && chunk_level(c + 1) == chunk_level(c))
          for (size_t e = c * 64; e < (c + 1) * 64; ++e) rc[e] |= 1u << 20;
      }
      while (nch % kSweepDepth) { rc.resize(rc.size() + 64, 0u); ++nch; } // null chunks: the sweep is unrolled by its depth
      nchunk = (int)nch;
      rc.resize(rc.size() + 16 * 64, 0u); // spare chunks: the sweep prefetches unconditionally
      return rc;
    };
    {
      // Work list of the LU: the columns j < ns that have pivots (U(:,j) not empty), ascending, then the trailing
      // columns ns..n-1.  Columns j < ns WITHOUT pivots ("leaves": nothing is ever subtracted from them) are
      // factored beforehand in one elementwise pass: D^-1_j = 1/P(j,j), L(:,j) = P(rows > j, j) * D^-1_j, their L
      // pattern being exactly P's (leaf_diag: position of P(j,j) | j<<32; leaf_ent: position in P | position in L
      // << 20 | j << 40).
      std::vector<unsigned long long> ud, leaf_diag, leaf_ent;
      std::vector<uint32_t> uop;
      if ((size_t)S.nzl + 64 > 0xffff) throw std::runtime_error("LU layout: L storage too large for the 16-bit positions of the pivot operations");
      std::vector<LuCol> lc;
      int nwork_sparse = 0;
      for (int j = 0; j < nS; ++j) {
        if (j < S.ns && S.Ucolend[j] == S.Ucolptr[j]) {
          bool pattern_ok = (S.Lcolend[j] - S.Lcolptr[j]) == 0;
          int nbelow = 0;
          for (int q = S.Pcolptr[j]; q < S.Pcolptr[j + 1]; ++q) if (S.Prow[q] > j) ++nbelow;
          pattern_ok = nbelow == S.Lcolend[j] - S.Lcolptr[j];
          bool has_diag = false, above = false;
          for (int q = S.Pcolptr[j]; q < S.Pcolptr[j + 1]; ++q) { if (S.Prow[q] == j) has_diag = true; if (S.Prow[q] < j) above = true; }
          if (pattern_ok && has_diag && !above) {
            for (int q = S.Pcolptr[j]; q < S.Pcolptr[j + 1]; ++q) {
              const int r = S.Prow[q];
              if (r == j) leaf_diag.push_back((unsigned long long)q | ((unsigned long long)j << 32));
              else {
                int lp = -1;
                for (int t = S.Lcolptr[j]; t < S.Lcolend[j]; ++t) if (S.Lrow[t] == r) lp = t;
                if (lp < 0) throw std::runtime_error("LU layout: leaf column pattern mismatch");
                leaf_ent.push_back((unsigned long long)q | ((unsigned long long)lp << 20) | ((unsigned long long)j << 40));
              }
            }
            continue;
          }
        }
        LuCol c{};
        c.u0 = S.Ucolptr[j]; c.u1 = S.Ucolend[j]; c.lc0 = S.Lcolptr[j]; c.lc1 = S.Lcolend[j]; c.p0 = S.Pcolptr[j]; c.p1 = S.Pcolptr[j + 1];
        c.ur = S.Ucolend[j]; // rows < ns only: the pivots ns..j-1 of a trailing column are applied in registers
        c.d0 = (int)ud.size(); c.j = j;
        constexpr int kChunk = 64 / kLuDepth * kLuDepth; // descriptors the kernel fetches at a time: the first of a fetch reloads w[k] too
        for (int q = S.Ucolptr[j]; q < c.ur; ++q) {
          const int k = S.Urow[q];
          const int len = S.Lcolend[k] - S.Lcolptr[k];
          for (int off = 0; off < std::max(len, 1); off += 64) { // L columns longer than 64 rows: one descriptor per 64 rows
            const unsigned long long rows = (unsigned long long)std::min(64, len - off), start = (unsigned long long)(S.Lcolptr[k] + off);
            const bool reload = (S.Ugrp[q] && off == 0) || ((int)ud.size() - c.d0) % kChunk == 0;
            if (start >= (1ull << 20)) throw std::runtime_error("LU layout: L storage too large for the pivot descriptors");
            ud.push_back((unsigned long long)k | ((rows * 8) << 16) | ((unsigned long long)(reload ? 1 : 0) << 30) |
                         ((start * 2) << 32) | (((rows > 0 ? rows - 1 : 0) * 8) << 53));
          }
        }
        c.d1 = (int)ud.size();
        // the same pivots as entry-parallel operations (device_tables.hpp, Uop; built only for RG_LU_OPS = 1)
        c.o0 = (int)(uop.size() / 64);
        if (RG_LU_OPS) {
          // (bit 12 of an operation's first word: the operation after it belongs to the same level, so its multipliers may be read ahead)
          size_t level_first_op = uop.size() / 64;
          auto flush = [&]() {
            while (uop.size() % 64) { const uint32_t lane = (uint32_t)(uop.size() % 64); uop.push_back(((uint32_t)(S.nzl + lane) << 16) | 0u); }
            const size_t end_op = uop.size() / 64;
            for (size_t o = level_first_op; o + 1 < end_op; ++o) uop[o * 64] |= 0x1000u;
            level_first_op = end_op;
          };
          for (int q = S.Ucolptr[j]; q < c.ur; ++q) {
            if (S.Ugrp[q] && q > S.Ucolptr[j]) flush(); // a new level: its multipliers are final only after the operations before it
            const int k = S.Urow[q];
            for (int t = S.Lcolptr[k]; t < S.Lcolend[k]; ++t) uop.push_back(((uint32_t)t << 16) | (uint32_t)k); // (k <= 1022: build_symbolic)
          }
          flush();
        }
        c.o1 = (int)(uop.size() / 64);
        lc.push_back(c);
        if (j < S.ns) ++nwork_sparse;
      }
      ud.resize(ud.size() + 64, 0ull);
      dn.Udesc = up(ud);
      for (int q = 0; q < 64 * (4 * RG_LU_OPS_DEPTH); ++q) uop.push_back(((uint32_t)(S.nzl + q % 64) << 16) | 0u); // spare operations: the prefetch runs ahead
      dn.Uop = up(uop);
      dn.nwork_sparse = nwork_sparse; dn.nwork = (int)lc.size();
      dn.nleaf = (int)leaf_diag.size(); dn.nleaf_ent = (int)leaf_ent.size();
      leaf_diag.resize(leaf_diag.size() + 64, 0ull); leaf_ent.resize(leaf_ent.size() + 64, 0ull);
      dn.leaf_diag = up(leaf_diag); dn.leaf_ent = up(leaf_ent);
      // k_solve_team: the columns k < ns that have pivots, level by level (a column's level = 1 + the highest level among its
      // pivots; columns of one level do not feed each other), the columns of a level dealt to the kTeam waves by weight
      // (pivots + a constant per column, heaviest first to the least loaded wave).  One list per wave, two spare items each.
      {
        std::vector<int> lev(nS, 0);
        int nlev = 0;
        for (int t = 0; t < nwork_sparse; ++t) {
          const int j = lc[t].j;
          int l = 0;
          for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) l = std::max(l, lev[S.Urow[q]]);
          lev[j] = l + 1; nlev = std::max(nlev, l + 1);
        }
        std::vector<std::vector<std::vector<int>>> deal(kTeam, std::vector<std::vector<int>>(nlev)); // [wave][level-1] -> work items
        for (int l = 1; l <= nlev; ++l) {
          std::vector<int> items;
          for (int t = 0; t < nwork_sparse; ++t) if (lev[lc[t].j] == l) items.push_back(t);
          std::stable_sort(items.begin(), items.end(), [&](int a, int b) { return lc[a].d1 - lc[a].d0 > lc[b].d1 - lc[b].d0; });
          long load[kTeam] = {0};
          for (int t : items) {
            const int w = (int)(std::min_element(load, load + kTeam) - load);
            deal[w][l - 1].push_back(t); load[w] += (lc[t].d1 - lc[t].d0) + 4;
          }
        }
        std::vector<LuCol> tl;
        std::vector<int> lp; // [kTeam][nlev + 1], positions relative to the wave's own list
        for (int w = 0; w < kTeam; ++w) {
          dn.team_base[w] = (int)tl.size();
          int pos = 0;
          for (int l = 0; l < nlev; ++l) {
            lp.push_back(pos);
            std::sort(deal[w][l].begin(), deal[w][l].end());
            for (int t : deal[w][l]) { tl.push_back(lc[t]); ++pos; }
          }
          lp.push_back(pos);
          tl.push_back(tl.empty() ? LuCol{} : tl.back()); tl.push_back(tl.back());
        }
        dn.team_nlev = nlev;
        dn.lucol_team = up(tl); dn.team_lev_ptr = up(lp);
      }
      lc.push_back(lc.empty() ? LuCol{} : lc.back()); lc.push_back(lc.back()); // the column prefetch reads two ahead
      dn.lucol = up(lc);
    }
    dn.nzl_stream = S.nzl_stream; dn.nzu_stream = S.nzu_stream;
    dn.Lrc = up(pack(S.Lrow, S.Lcol, S.Llev, (size_t)S.nzl_stream, dn.nchunkL));
    dn.Urc = up(pack(S.Urow, S.Ucol, S.Ulev, (size_t)S.nzu_stream, dn.nchunkU));
  }
  dn.i_H = h.idx10[1] - 1; dn.i_E = h.idx10[2] - 1; dn.i_gH = h.i_gH - 1; dn.i_gH2 = h.i_gH2 - 1; dn.i_gH2O = h.i_gH2O - 1;
  dn.i_Grain0 = h.i_Grain0 - 1; dn.i_GrainM = h.i_GrainM - 1; dn.i_GrainP = h.i_GrainP - 1;
  {
    std::vector<uint8_t> cls(nS, 0);
    for (int k = 0; k < 10; ++k) if (h.idx10[k] > 0) cls[h.idx10[k] - 1] = 1;
    if (h.i_Grain0 > 0) for (int g : {h.i_Grain0, h.i_GrainM, h.i_GrainP}) if (g > 0) cls[g - 1] = 2;
    for (int g : h.grain) cls[g - 1] = 3;
    dn.s_tolclass = up(cls);
    std::vector<int8_t> chg(nS);
    for (int i = 0; i < nS; ++i) chg[i] = (int8_t)h.elements[i][0];
    dn.s_charge = up(chg);
  }
  set_ref_layout();
  dn.i_H2 = h.idx10[0] - 1;
  {
    // The number of grains is conserved by every reaction of the reference's networks (charging and recombination move a grain between
    // Grain0 / Grain- / Grain+).  dev_rhs then makes d/dt of their sum exactly zero (see there); a network that breaks the rule is left alone.
    dn.grain_conserved = h.i_Grain0 > 0 ? 1 : 0;
    auto is_grain = [&](int sp) { return sp > 0 && (sp == h.i_Grain0 || sp == h.i_GrainM || sp == h.i_GrainP); };
    for (const Reaction &x : h.R) {
      int a = 0, b = 0;
      for (int k = 0; k < x.n_reac; ++k) a += is_grain(x.reac[k]);
      for (int k = 0; k < x.n_prod; ++k) b += is_grain(x.prod[k]);
      if (a != b && h.kind((int)(&x - &h.R[0])) != K_NONE) dn.grain_conserved = 0;
    }
  }
  {
    // charge conservation (dev_rhs): E- must be a species with charge -1 and every reaction chem_ode_f acts on must balance
    dn.charge_conserved = (h.idx10[2] > 0 && h.elements[h.idx10[2] - 1][0] == -1) ? 1 : 0;
    for (int r = 0; r < nR && dn.charge_conserved; ++r) {
      const Reaction &x = h.R[r];
      if (h.kind(r) == K_NONE) continue;
      int q = 0;
      for (int k = 0; k < x.n_reac; ++k) q += h.elements[x.reac[k] - 1][0];
      for (int k = 0; k < x.n_prod; ++k) q -= h.elements[x.prod[k] - 1][0];
      if (q != 0) dn.charge_conserved = 0;
    }
    if (!std::getenv("RACGPU_CHARGE_BALANCE")) dn.charge_conserved = 0; // developer switch, off by default (dev_rhs says why)
  }
  dn.r_h2form = -1; // chem_cal_rates stores the coefficient of every itype-0 and every gH-first itype-63 reaction in turn: the last one stays
  for (int r = 0; r < nR; ++r)
    if (h.R[r].itype == 0 || (h.R[r].itype == 63 && h.R[r].rname[0] == "gH")) dn.r_h2form = r;
  // H2_form_use_moeq (src/chemistry.f90:876-881) reads adsorb_coeff(counterpart of gH) and desorb_coeff(gH): the last adsorption
  // reaction of that species and the last desorption reaction of gH BEFORE the first gH-first itype-63 reaction in the file (a later one
  // would hand the coefficient of the previous chem_cal_rates call to it: such a network is refused when the switch is on, -2)
  dn.moeq_r61 = dn.moeq_r62 = -1;
  {
    int first63 = -1;
    for (int r = 0; r < nR && first63 < 0; ++r) if (h.R[r].itype == 63 && h.R[r].rname[0] == "gH") first63 = r;
    if (first63 >= 0) {
      const int gH = h.R[first63].reac[0], H = h.counterpart[gH - 1];
      for (int r = 0; r < nR; ++r) {
        if (h.R[r].itype == 61 && h.R[r].reac[0] == H) dn.moeq_r61 = r < first63 ? r : -2;
        if (h.R[r].itype == 62 && h.R[r].reac[0] == gH) dn.moeq_r62 = r < first63 ? r : -2;
      }
      if (dn.moeq_r61 < 0 || dn.moeq_r62 < 0) dn.moeq_r61 = dn.moeq_r62 = -2; // (the switch cannot be honoured on this network)
    }
  }
  HIP_OK(hipEventCreate(&ev0));
  HIP_OK(hipEventCreate(&ev1));
  HIP_OK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  HIP_OK(hipStreamCreateWithFlags(&team_stream, hipStreamNonBlocking));
  HIP_OK(hipHostMalloc((void **)&parked_host, sizeof(int)));
  *parked_host = 0;
  { void *d = nullptr; HIP_OK(hipMalloc(&d, sizeof(DevNet))); HIP_OK(hipMemcpy(d, &dn, sizeof(DevNet), hipMemcpyHostToDevice)); dev_allocs.push_back(d); dn_dev = (DevNet *)d; }
  { void *d = nullptr; HIP_OK(hipMalloc(&d, sizeof(DevParams))); dev_allocs.push_back(d); dp_dev = (DevParams *)d; }
  uploaded = true;
}

void racgpu_network::upload_hc() {
  if (!hhc.loaded) throw std::runtime_error("gas-temperature evolution needs racgpu_heating_cooling_load first");
  for (void *q : hc_allocs) (void)hipFree(q);
  hc_allocs.clear();
  auto H = std::make_unique<DevHC>();
  std::memset(H.get(), 0, sizeof(DevHC));
  H->cfg = hcfg; H->h2 = hhc.h2; H->h2o = hhc.h2o; H->co = hhc.co; H->nii = hhc.nii; H->siii = hhc.siii; H->feii = hhc.feii;
  auto put = [&](const void *src, size_t bytes) { void *d = nullptr; HIP_OK(hipMalloc(&d, std::max<size_t>(bytes, 8))); if (bytes) HIP_OK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice)); hc_allocs.push_back(d); return d; };
  const size_t nh = hhc.heat_rxn.size();
  std::vector<uint16_t> ha(nh), hb(nh);
  for (size_t i = 0; i < nh; ++i) { ha[i] = (uint16_t)(net.R[hhc.heat_rxn[i]].reac[0] - 1); hb[i] = (uint16_t)(net.R[hhc.heat_rxn[i]].reac[1] - 1); }
  H->nheat = (int)nh;
  H->heat_rxn = (const int *)put(hhc.heat_rxn.data(), nh * sizeof(int)); H->heat_val = (const double *)put(hhc.heat_val.data(), nh * sizeof(double));
  H->heat_a = (const uint16_t *)put(ha.data(), nh * 2); H->heat_b = (const uint16_t *)put(hb.data(), nh * 2);
  auto sp = [&](const char *nm) { return net.species_index(nm) - 1; };
  H->i_H2 = sp("H2"); H->i_HI = sp("H"); H->i_E = sp("E-"); H->i_CI = sp("C"); H->i_CII = sp("C+"); H->i_OI = sp("O"); H->i_O2 = sp("O2"); H->i_CO = sp("CO");
  H->i_H2O = sp("H2O"); H->i_OH = sp("OH"); H->i_Hplus = sp("H+"); H->i_Heplus = sp("He+"); H->i_gH = sp("gH"); H->i_NII = sp("N+"); H->i_SiII = sp("Si+");
  H->i_FeII = sp("Fe+");
  for (int k = 0; k < 10; ++k) { H->idx10[k] = net.idx10[k] - 1; H->kref_row[k] = net.ref_kref_Trow[k]; }
  H->kref_col0 = net.ref_kref_Tcol0;
  if (!hc_dev) HIP_OK(hipMalloc((void **)&hc_dev, sizeof(DevHC)));
  HIP_OK(hipMemcpy(hc_dev, H.get(), sizeof(DevHC), hipMemcpyHostToDevice));
}

void racgpu_network::ensure_workspace(long slots, long rate_cells) {
  if (slots <= ws_slots && rate_cells <= ws_rate_cells) return;
  slots = std::max(slots, ws_slots); rate_cells = std::max(rate_cells, ws_rate_cells);
  free_ws();
  auto alloc = [&](size_t count) { void *d = nullptr; HIP_OK(hipMalloc(&d, count * sizeof(double))); ws_allocs.push_back(d); return (double *)d; };
  ws.rates = alloc((size_t)rate_cells * dn.nR + 448); // per CELL; +448: the RHS of the last cell reads past nR
  ws.yh = alloc((size_t)slots * 6 * dn.npad);
  ws.P = alloc((size_t)slots * dn.nnzJ + 64); // spare: the LU's column prefetch reads up to 63 entries past a column
  ws.L = alloc((size_t)slots * std::max(dn.nzl, 1) + 2048); // spare: prefetches of the last slot read past nzl
  ws.U = alloc((size_t)slots * std::max(dn.nzu, 1) + 2048);
  ws.Dinv = alloc((size_t)slots * dn.npad);
  ws.rtol = alloc((size_t)slots * dn.npad);
  ws.atol = alloc((size_t)slots * dn.npad);
  ws.ygood = alloc((size_t)slots * dn.npad);
  ws.acor = alloc((size_t)slots * dn.npad);
  ws.ewt = alloc((size_t)slots * dn.npad);
  ws.Pb = alloc((size_t)slots * dn.npad);
  ws.zb = alloc((size_t)slots * dn.npad);
  ws.park = alloc((size_t)slots * (dn.npad + kParkWords));
  void *c = nullptr;
  HIP_OK(hipMalloc(&c, 64 + (size_t)slots * sizeof(int)));
  ws_allocs.push_back(c);
  ws.counter = (int *)c; // 16 words of counters (solve_pass), then the list of slots that hold parked cells
  ws_slots = slots; ws_rate_cells = rate_cells;
}

static void cfode_bdf(DevParams &P) { // BDF method coefficients, orders 1..5 (DCFODE METH=2, reference src/opkda1.f:146-171)
  double pc[8] = {0};
  pc[1] = 1.0;
  double rq1fac = 1.0;
  std::memset(P.elco, 0, sizeof P.elco);
  std::memset(P.tesco, 0, sizeof P.tesco);
  for (int nq = 1; nq <= 5; ++nq) {
    const double fnq = nq;
    pc[nq + 1] = 0.0;
    for (int i = nq + 1; i >= 2; --i) pc[i] = pc[i - 1] + fnq * pc[i];
    pc[1] = fnq * pc[1];
    for (int i = 1; i <= nq + 1; ++i) P.elco[nq][i] = pc[i] / pc[2];
    P.elco[nq][2] = 1.0;
    P.tesco[nq][1] = rq1fac;
    P.tesco[nq][2] = (nq + 1) / P.elco[nq][1];
    P.tesco[nq][3] = (nq + 2) / P.elco[nq][1];
    rq1fac = rq1fac / fnq;
  }
}

static void check_moeq(const racgpu_network *h, const DevParams &P, bool rates_only);
static DevParams to_dev(const racgpu_params *p) {
  if (p->evol_dust_size) throw std::runtime_error("evol_dust_size = .true. is not implemented");
  if (!(p->dt_first_step > 0.0) || !(p->ratio_tstep > 1.0) || !(p->t_max > 0.0)) throw std::runtime_error("need dt_first_step > 0, ratio_tstep > 1, t_max > 0");
  DevParams P{};
  P.RTOL = p->RTOL; P.ATOL = p->ATOL; P.t_max = p->t_max; P.dt_first_step = p->dt_first_step; P.ratio_tstep = p->ratio_tstep;
  P.Diff2DesorRatio = p->Diff2DesorRatio; P.special_gH_E_diff = p->special_gH_E_diff;
  P.mxstep = p->mxstep_per_interval; P.steps_reset = p->steps_reset_solver; P.use_special_gH_mobi = p->use_special_gH_mobi; P.h2_moeq = p->H2_form_use_moeq ? 1 : 0;
  P.tol_j = p->tol_policy_j > 0 ? p->tol_policy_j : 1;
  P.max_steps_per_cell = p->max_steps_per_cell;
  P.max_runtime_allowed = p->max_runtime_allowed;
  P.rt_cost_f = p->rt_cost_f; P.rt_cost_jac = p->rt_cost_jac; P.rt_cost_lu = p->rt_cost_lu;
  P.n_record = racgpu_n_record(p, 0.0, p->t_max);
  if (const char *e = std::getenv("RACGPU_DEBUG_TRACE")) P.debug_max_calls = std::atoi(e);
  P.debug_dump_call = -1;
  if (const char *e = std::getenv("RACGPU_DEBUG_DUMP_CALL")) P.debug_dump_call = std::atoi(e);
  cfode_bdf(P);
  return P;
}

template <typename F>
static int guarded(F &&f) {
  try { f(); return 0; }
  catch (const std::exception &e) { return fail(e.what()); }
}

struct DevBuf { // a device buffer that is either the caller's (MEM_DEVICE) or a staged copy (MEM_HOST)
  void *d = nullptr; bool own = false; size_t bytes = 0; void *host = nullptr;
  DevBuf(const void *src, size_t nbytes, int mem, bool copy_in) : bytes(nbytes) {
    if (!src) return;
    if (mem == RACGPU_MEM_DEVICE) { d = const_cast<void *>(src); return; }
    HIP_OK(hipMalloc(&d, std::max<size_t>(nbytes, 8)));
    own = true; host = const_cast<void *>(src);
    if (copy_in) HIP_OK(hipMemcpy(d, src, nbytes, hipMemcpyHostToDevice));
  }
  void copy_out() { if (own && host) HIP_OK(hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost)); }
  ~DevBuf() { if (own) (void)hipFree(d); }
};

extern "C" {

const char *racgpu_last_error(void) { return g_err.c_str(); }

int racgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

racgpu_network *racgpu_network_load(const char *path) {
  try {
    auto h = std::make_unique<racgpu_network>();
    parse_network(path, h->net);
    return h.release();
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

void racgpu_network_destroy(racgpu_network *h) { delete h; }

int racgpu_network_dims(const racgpu_network *h, int32_t *nS, int32_t *nR, int32_t *nnzJ, int32_t *nzl, int32_t *nzu) {
  if (!h) return fail("null network");
  if (nS) *nS = h->net.nS;
  if (nR) *nR = h->net.nR;
  if (nnzJ) *nnzJ = (int32_t)h->net.Jrow.size();
  if (nzl) *nzl = h->net.sym.nzl_entries; // entries of the factors (the storage also holds alignment padding)
  if (nzu) *nzu = h->net.sym.nzu_entries;
  return 0;
}

int racgpu_network_set_reference_lenrw(racgpu_network *h, int32_t lenrw) {
  if (!h) return fail("null network");
  if (lenrw < 0) return fail("lenrw must be >= 0");
  return guarded([&] {
    h->net.ref_lenrw = lenrw;
    h->set_ref_layout();
    if (h->uploaded) HIP_OK(hipMemcpy(h->dn_dev, &h->dn, sizeof(DevNet), hipMemcpyHostToDevice));
  });
}

int racgpu_network_reference_lenrw(const racgpu_network *h) { return h ? h->net.ref_lenrw : -1; }

int racgpu_species_name(const racgpu_network *h, int32_t i, char *buf, int32_t buflen) {
  if (!h || i < 1 || i > h->net.nS || buflen < 1) return fail("bad species index");
  std::snprintf(buf, (size_t)buflen, "%s", h->net.names[i - 1].c_str());
  return 0;
}

int racgpu_species_index(const racgpu_network *h, const char *name) { return h ? h->net.species_index(name) : 0; }

int racgpu_reactions(const racgpu_network *h, int32_t *reac, int32_t *prod, int32_t *n_reac, int32_t *n_prod, int32_t *itype, int32_t *n_dupli) {
  if (!h) return fail("null network");
  for (int r = 0; r < h->net.nR; ++r) {
    const Reaction &x = h->net.R[r];
    if (reac) for (int k = 0; k < 3; ++k) reac[3 * r + k] = x.reac[k];
    if (prod) for (int k = 0; k < 4; ++k) prod[4 * r + k] = x.prod[k];
    if (n_reac) n_reac[r] = x.n_reac;
    if (n_prod) n_prod[r] = x.n_prod;
    if (itype) itype[r] = x.itype;
    if (n_dupli) n_dupli[r] = h->net.dupli_ptr[r + 1] - h->net.dupli_ptr[r];
  }
  return 0;
}

int racgpu_species_attrs(const racgpu_network *h, double *mass, double *vib, double *Ed, int32_t *cp, int32_t *charge) {
  if (!h) return fail("null network");
  for (int i = 0; i < h->net.nS; ++i) {
    if (mass) mass[i] = h->net.mass_num[i];
    if (vib) vib[i] = h->net.vib_freq[i];
    if (Ed) Ed[i] = h->net.Edesorb[i];
    if (cp) cp[i] = h->net.counterpart[i];
    if (charge) charge[i] = h->net.elements[i][0];
  }
  return 0;
}

int racgpu_species_elements(const racgpu_network *h, int32_t *el) {
  if (!h || !el) return fail("null argument");
  for (int i = 0; i < h->net.nS; ++i) for (int e = 0; e < kNumElements; ++e) el[i * kNumElements + e] = h->net.elements[i][e];
  return 0;
}

int racgpu_reaction_rows(const racgpu_network *h, double *ABC, double *Tr, char *ctype, char *rel, char *names) {
  if (!h) return fail("null network");
  for (int r = 0; r < h->net.nR; ++r) {
    const Reaction &x = h->net.R[r];
    if (ABC) for (int k = 0; k < 3; ++k) ABC[3 * r + k] = x.ABC[k];
    if (Tr) for (int k = 0; k < 2; ++k) Tr[2 * r + k] = x.Trange[k];
    if (ctype) { ctype[2 * r] = x.ctype[0]; ctype[2 * r + 1] = x.ctype[1]; }
    if (rel) rel[r] = x.reliability;
    if (names)
      for (int k = 0; k < 7; ++k) {
        const std::string &nm = k < 3 ? x.rname[k] : x.pname[k - 3];
        char *dst = names + ((size_t)r * 7 + k) * 12;
        for (int c = 0; c < 12; ++c) dst[c] = c < (int)nm.size() ? nm[c] : ' ';
      }
  }
  return 0;
}

int racgpu_jac_pattern(const racgpu_network *h, int32_t *colptr, int32_t *rowidx) {
  if (!h) return fail("null network");
  if (colptr) for (int j = 0; j <= h->net.nS; ++j) colptr[j] = h->net.Jcolptr[j] + 1;
  if (rowidx) for (size_t q = 0; q < h->net.Jrow.size(); ++q) rowidx[q] = h->net.Jrow[q] + 1;
  return 0;
}

int racgpu_lu_ordering(const racgpu_network *h, int32_t *perm, int32_t *first_dense, int32_t *p_storage) {
  if (!h) return fail("null network");
  if (perm) for (int i = 0; i < h->net.nS; ++i) perm[i] = h->net.sym.perm[i] + 1;
  if (first_dense) *first_dense = h->net.sym.ns + 1;
  if (p_storage) for (size_t q = 0; q < h->net.sym.Psrc.size(); ++q) p_storage[q] = h->net.sym.Psrc[q] + 1;
  return 0;
}

int racgpu_load_initial_abundances(const racgpu_network *h, const char *path, double *y0) {
  if (!h) return fail("null network");
  return guarded([&] { load_initial_abundances(h->net, path, y0); });
}

void racgpu_params_default(racgpu_params *p) {
  std::memset(p, 0, sizeof *p);
  p->RTOL = 1e-4; p->ATOL = 1e-30; p->t_max = 1e6; p->dt_first_step = 1e-8; p->ratio_tstep = 1.1;
  p->max_runtime_allowed = 60.0; p->Diff2DesorRatio = 0.5; p->special_gH_E_diff = 225.0;
  p->mxstep_per_interval = 6000; p->steps_reset_solver = 50; p->tol_policy_j = 1; p->max_steps_per_cell = 0;
  p->rt_cost_f = 47e-6; p->rt_cost_jac = 10.4e-3; p->rt_cost_lu = 1.0e-3;
}

int racgpu_n_record(const racgpu_params *p, double t0, double t_max) {
  return (int)std::ceil(std::log((t_max - t0) / p->dt_first_step * (p->ratio_tstep - 1.0) + 1.0) / std::log(p->ratio_tstep)) + 1;
}

int racgpu_set_tolerances(const racgpu_network *h, const racgpu_params *p, int32_t j, double d2h, double *rtol, double *atol) {
  if (!h) return fail("null network");
  const HostNetwork &n = h->net;
  double r, a, rT, aT;
  switch (j) {
    case 1: r = p->RTOL; a = p->ATOL; rT = 1e-3; aT = 1e-1; break;
    case 2: r = std::fmin(p->RTOL * 1e1, 1e-4); a = std::fmin(p->ATOL * 1e5, 1e-25); rT = 1e-2; aT = 1e-1; break;
    case 3: r = std::fmin(p->RTOL * 1e2, 1e-4); a = std::fmin(p->ATOL * 1e10, 1e-20); rT = 1e-3; aT = 1e0; break;
    case 4: r = std::fmin(p->RTOL * 1e2, 1e-4); a = std::fmin(p->ATOL * 1e10, 1e-18); rT = 1e-3; aT = 1e0; break;
    default: { // x**j with integer j: repeated squaring, as flang's runtime evaluates it
      auto powi = [](double a, int b) { double r = 1.0; for (;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; } return r; };
      r = std::fmin(p->RTOL * powi(2.0, j), 1e-3); a = std::fmin(p->ATOL * powi(1e2, j), 1e-15); rT = 1e-2; aT = 1e0;
    }
  }
  for (int i = 0; i < n.nS; ++i) { rtol[i] = r; atol[i] = a; }
  rtol[n.nS] = rT; atol[n.nS] = aT;
  for (int k = 0; k < 10; ++k) if (n.idx10[k] > 0) { rtol[n.idx10[k] - 1] = std::fmax(p->RTOL, 1e-4); atol[n.idx10[k] - 1] = std::fmax(p->ATOL, 1e-30); }
  if (n.i_Grain0 > 0)
    for (int g : {n.i_Grain0, n.i_GrainM, n.i_GrainP}) if (g > 0) { rtol[g - 1] = 1e-4; atol[g - 1] = std::fmax(d2h * 1e-6, 1e-30); }
  for (int g : n.grain) { rtol[g - 1] = std::fmax(p->RTOL, 1e-3); atol[g - 1] = std::fmax(p->ATOL, d2h * 1e-8); }
  return 0;
}

int racgpu_init_abundances(const racgpu_network *h, const double *y0, const double *cells, int64_t ncell, double *y) {
  if (!h) return fail("null network");
  const int nS = h->net.nS, ig = h->net.i_Grain0;
  for (int64_t c = 0; c < ncell; ++c) {
    std::memcpy(y + c * nS, y0, (size_t)nS * sizeof(double));
    if (ig > 0) y[c * nS + ig - 1] = cells[c * RACGPU_NPAR + RACGPU_P_D2H];
  }
  return 0;
}

int racgpu_set_device(int dev) { return guarded([&] { HIP_OK(hipSetDevice(dev)); }); }

int racgpu_set_stream(racgpu_network *h, void *s) {
  if (!h) return fail("null network");
  h->stream = (hipStream_t)s;
  return 0;
}

static size_t lds_bytes_team(const DevNet &dn) { // + a work column and the spare doubles for each of the other waves
  const size_t nlds = (dn.nS + 1) & ~1;
  return (3 * nlds + 64 + (kTeam - 1) * (nlds + 64)) * sizeof(double);
}
static size_t lds_bytes(const DevNet &dn) { return (size_t)3 * ((dn.nS + 1) & ~1) * sizeof(double) + 64 * sizeof(double); } // + one spare double per lane behind the last vector

int racgpu_rates(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, double *rates) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    check_moeq(h, P, true); HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dr(rates, (size_t)ncell * h->dn.nR * 8, RACGPU_MEM_HOST, false);
    hipLaunchKernelGGL(k_rates, dim3((unsigned)ncell), dim3(64), 0, h->stream, h->dn_dev, h->dp_dev, (const double *)dc.d, (double *)dr.d, (double *)nullptr);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dr.copy_out();
  });
}

int racgpu_rhs(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, const double *y, double *ydot) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    check_moeq(h, P, false); HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dy(y, ncell * nS * 8, RACGPU_MEM_HOST, true),
        dd(ydot, ncell * nS * 8, RACGPU_MEM_HOST, false);
    h->ensure_workspace((long)ncell, (long)ncell);
    hipLaunchKernelGGL(k_rhs, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, (const double *)dc.d, (const double *)dy.d,
                       h->ws.rates, (double *)dd.d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dd.copy_out();
  });
}

int racgpu_jac_csc(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, const double *y, double *vals) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    check_moeq(h, P, false); HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dy(y, ncell * nS * 8, RACGPU_MEM_HOST, true),
        dv(vals, (size_t)ncell * h->dn.nnzJ * 8, RACGPU_MEM_HOST, false);
    h->ensure_workspace((long)ncell, (long)ncell);
    hipLaunchKernelGGL(k_jac, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, (const double *)dc.d, (const double *)dy.d,
                       h->ws.rates, (double *)dv.d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dv.copy_out();
  });
}

int racgpu_newton_solve(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, const double *y, double gamma, double *bx) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    check_moeq(h, P, false); HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dy(y, ncell * nS * 8, RACGPU_MEM_HOST, true),
        db(bx, ncell * nS * 8, RACGPU_MEM_HOST, true);
    h->ensure_workspace((long)ncell, (long)ncell);
    const char *rep_env = getenv("RACGPU_DEBUG_REPEAT"); // developer aid: time the factorisation + solve in isolation
    const char *mode_env = getenv("RACGPU_DEBUG_REPEAT_MODE");
    const int repeat = (rep_env ? std::min(std::max(1, atoi(rep_env)), 0xffffff) : 1) | ((mode_env ? atoi(mode_env) & 3 : 0) << 24);
    long long *cyc_dev = nullptr;
    if (rep_env) { HIP_OK(hipMalloc((void **)&cyc_dev, (size_t)ncell * 8 * sizeof(long long))); HIP_OK(hipMemset(cyc_dev, 0, (size_t)ncell * 8 * sizeof(long long))); }
    hipLaunchKernelGGL(k_newton, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, h->ws, (const double *)dc.d,
                       (const double *)dy.d, gamma, (double *)db.d, repeat, cyc_dev);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    if (cyc_dev) {
      std::vector<long long> cy((size_t)ncell * 8);
      HIP_OK(hipMemcpy(cy.data(), cyc_dev, cy.size() * sizeof(long long), hipMemcpyDeviceToHost));
      (void)hipFree(cyc_dev);
      double a[6] = {0, 0, 0, 0, 0, 0};
      for (int64_t c = 0; c < ncell; ++c) for (int k = 0; k < 6; ++k) a[k] += (double)cy[(size_t)c * 8 + k];
      const double den = (double)ncell * (repeat & 0xffffff);
      fprintf(stderr, "[racgpu debug] %lld cells x %d: cycles per LU %.0f (scatter %.0f, lds pivots %.0f, register pivots %.0f, finish %.0f), per solve %.0f\n",
              (long long)ncell, repeat & 0xffffff, a[0] / den, a[2] / den, a[3] / den, a[4] / den, a[5] / den, a[1] / den);
    }
    db.copy_out();
  });
}

int racgpu_set_team_threshold(racgpu_network *h, double frac) {
  if (!h) return fail("null network");
  h->team_frac = frac;
  h->park_enabled = frac >= 0.0;
  return 0;
}

int64_t racgpu_last_team_cells(const racgpu_network *h) { return h ? h->last_team_cells : -1; }

int64_t racgpu_last_parked_cells(racgpu_network *h) {
  if (!h || !h->parked_host) return -1;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
  return *h->parked_host;
}

int racgpu_set_cost_hints(racgpu_network *h, const double *cost, int64_t ncell) {
  if (!h) return fail("null network");
  if (!cost || ncell <= 0) { h->cost_hints.clear(); return 0; }
  return guarded([&] { h->cost_hints.assign(cost, cost + ncell); });
}

int64_t racgpu_workspace_bytes_per_cell(const racgpu_network *h) {
  if (!h) return -1;
  const HostNetwork &n = h->net;
  const int64_t npad = (n.nS + 63) / 64 * 64;
  return 8 * ((int64_t)n.nR + 6 * npad + (int64_t)n.Jrow.size() + n.sym.nzl + n.sym.nzu + 3 * npad);
}

double racgpu_last_kernel_ms(const racgpu_network *h) {
  if (!h || !h->timed) return -1.0;
  float ms = -1.f;
  if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0;
  return ms;
}

// One chem_evol_solve pass over ncell cells whose buffers are all in device memory (any of the outputs may be null).
// Chunks the batch so that the per-cell rate vectors stay below 8 GB; one k_rates + one k_solve launch per chunk.
struct PassBufs {
  const double *cells; double *y; const double *t0; const int *tolj; double *t_final; int *quality; long long *stats;
  double *record, *touts, *cell_out;
  const double *hc = nullptr; // [ncell][RACGPU_NHC]: gas temperature co-evolving (k_solve_T) when given
};
static void solve_pass(racgpu_network *h, const DevParams &P, long ncell, const PassBufs &B, int flags, bool use_hints, bool first_timed) {
  const size_t nS = h->dn.nS;
  const size_t lds = lds_bytes(h->dn);
  // waves per CU: 3 per SIMD by registers (<= 168 VGPRs, tests/test_build_resources.py), and what the LDS holds (512 B static)
  long cap = 12;
  if (const char *e = std::getenv("RACGPU_WAVES_PER_CU")) cap = std::max(1, std::atoi(e)); // developer aid
  const long per_cu = std::max<long>(1, std::min<long>(cap, (long)(160 * 1024 / (lds + 512))));
  const long slots = std::min<long>(ncell, per_cu * h->cu_count);
  const long chunk_cells = std::max<long>(slots, std::min<long>(ncell, (long)(8e9 / (8.0 * h->dn.nR)))); // <= 8 GB of rates
  // optional longest-expected-first order, per chunk (indices relative to the chunk)
  const bool hinted = use_hints && (int64_t)h->cost_hints.size() == ncell;
  const long team_cap = hinted && h->team_frac > 0.0 ? std::min<long>(h->cu_count, ncell) : 0; // at most one team per CU
  h->ensure_workspace(slots + team_cap, chunk_cells);
  h->last_team_cells = 0;
  HIP_OK(hipMemsetAsync(h->parked_host, 0, sizeof(int), h->stream)); // (pinned host memory, in stream order)
  if (hinted) {
    if (h->order_cap < ncell) {
      HIP_OK(hipStreamSynchronize(h->stream));
      if (h->order_dev) (void)hipFree(h->order_dev);
      HIP_OK(hipMalloc((void **)&h->order_dev, (size_t)ncell * sizeof(int)));
      h->order_cap = ncell;
    }
    std::vector<int> &order = h->order_host;
    order.resize((size_t)ncell);
    for (long c0 = 0; c0 < ncell; c0 += chunk_cells) {
      const long nc = std::min<long>(chunk_cells, ncell - c0);
      for (long i = 0; i < nc; ++i) order[c0 + i] = (int)i;
      const double *cost = h->cost_hints.data() + c0;
      std::stable_sort(order.begin() + c0, order.begin() + c0 + nc, [&](int a, int b) { return cost[a] > cost[b]; });
    }
    HIP_OK(hipMemcpyAsync(h->order_dev, order.data(), (size_t)ncell * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // 'order' lives on this stack frame
  }
  for (long c0 = 0; c0 < ncell; c0 += chunk_cells) {
    const long nc = std::min<long>(chunk_cells, ncell - c0);
    const double *cells_c = B.cells + (size_t)c0 * RACGPU_NPAR;
    // pass 1: rate coefficients of every cell of the chunk (one wave per cell)
    hipLaunchKernelGGL(k_rates, dim3((unsigned)nc), dim3(64), 0, h->stream, h->dn_dev, h->dp_dev, cells_c, h->ws.rates,
                       B.cell_out ? B.cell_out + (size_t)c0 * RACGPU_NOUT : nullptr);
    HIP_OK(hipGetLastError());
    // pass 2: the persistent integrator
    // [0] k_solve's queue, [1] team workgroups started, [2] k_solve_team's queue, [3] k_solve waves that have left, [4] parked cells,
    // [5] k_solve_team_resume's queue
    HIP_OK(hipMemsetAsync(h->ws.counter, 0, 8 * sizeof(int), h->stream));
    if (c0 == 0 && first_timed) HIP_OK(hipEventRecord(h->ev0, h->stream));
    SolveArgs A{};
    A.ncell = (int)nc; A.flags = flags; A.cells = cells_c; A.yio = B.y + (size_t)c0 * nS;
    A.t0 = B.t0 ? B.t0 + c0 : nullptr; A.tolj = B.tolj ? B.tolj + c0 : nullptr;
    A.t_final = B.t_final ? B.t_final + c0 : nullptr; A.quality = B.quality ? B.quality + c0 : nullptr;
    A.stats = B.stats ? B.stats + (size_t)c0 * RACGPU_NSTAT : nullptr;
    A.record = B.record ? B.record + (size_t)c0 * P.n_record * (nS + 1) : nullptr;
    A.touts = B.touts ? B.touts + (size_t)c0 * P.n_record : nullptr;
    A.cell_out = B.cell_out ? B.cell_out + (size_t)c0 * RACGPU_NOUT : nullptr;
    A.order = hinted ? h->order_dev + c0 : nullptr;
    A.slot0 = 0;
    // Cells still being integrated once the queue is empty and at most two waves per CU are left are parked between two integrator
    // steps and taken up by teams (k_solve_team_resume): the end of a pass is a handful of cells on an otherwise idle chip.
    int park_per_cu = 2; // (1 / 2 / 3 per CU: 7742 / 7661 / 7659 ms on the configs[2] scan, 11.93 / 11.86 / 11.87 s on configs[1] without hints)
    if (const char *e = std::getenv("RACGPU_PARK_PER_CU")) park_per_cu = std::max(1, std::atoi(e)); // developer aid
    const int park_max = h->park_enabled ? h->cu_count * park_per_cu : 0;
    A.park_max = park_max; A.park_list = h->ws.counter + 16; A.park_count = h->ws.counter + 4;
    if (B.hc) { // gas temperature co-evolving: one wave per cell, four for the cells the cost hints single out; no hand-over
      A.hc = B.hc + (size_t)c0 * kNHC; A.hc_tab = h->hc_dev; A.park_max = 0;
      long nteamc = 0;
      if (team_cap > 0) {
        const double *cost = h->cost_hints.data() + c0;
        double total = 0.0;
        for (long i = 0; i < nc; ++i) total += cost[i];
        const double thr = h->team_frac * total / (double)slots;
        while (nteamc < std::min<long>(team_cap, nc) && cost[h->order_host[c0 + nteamc]] > thr) ++nteamc;
      }
      if (nteamc > 0) {
        SolveArgs T = A;
        T.ncell = (int)nteamc; T.slot0 = (int)slots;
        DevWork Wt = h->ws;
        Wt.counter = h->ws.counter + 2;
        HIP_OK(hipEventRecord(h->ev_fork, h->stream));
        HIP_OK(hipStreamWaitEvent(h->team_stream, h->ev_fork, 0));
        hipLaunchKernelGGL(k_solve_team_T, dim3((unsigned)nteamc), dim3(64 * kTeam), lds_bytes_team(h->dn), h->team_stream, h->dn_dev, h->dp_dev, Wt, T);
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventRecord(h->ev_join, h->team_stream));
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, h->stream, (const int *)(h->ws.counter + 1), (int)nteamc, 48000000LL);
        A.ncell = (int)(nc - nteamc); A.order += nteamc;
        h->last_team_cells += nteamc;
      }
      if (A.ncell > 0) {
        const long grid = std::max<long>(1, std::min<long>(slots, A.ncell));
        hipLaunchKernelGGL(k_solve_T, dim3((unsigned)grid), dim3(64), lds, h->stream, h->dn_dev, h->dp_dev, h->ws, A);
        HIP_OK(hipGetLastError());
      }
      if (nteamc > 0) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
      continue;
    }
    // Cells that would take more than team_frac of the pass's ideal length on their own (sum of costs / wave slots) go to
    // k_solve_team, four waves each, on a second stream and ahead of the bulk kernel; they are the head of the sorted order.
    long nteamc = 0;
    if (team_cap > 0) {
      const double *cost = h->cost_hints.data() + c0;
      double total = 0.0;
      for (long i = 0; i < nc; ++i) total += cost[i];
      const double thr = h->team_frac * total / (double)slots;
      while (nteamc < std::min<long>(team_cap, nc) && cost[h->order_host[c0 + nteamc]] > thr) ++nteamc;
    }
    if (nteamc > 0) {
      SolveArgs T = A;
      T.ncell = (int)nteamc; T.slot0 = (int)slots;
      DevWork Wt = h->ws;
      Wt.counter = h->ws.counter + 2;
      HIP_OK(hipEventRecord(h->ev_fork, h->stream));
      HIP_OK(hipStreamWaitEvent(h->team_stream, h->ev_fork, 0));
      hipLaunchKernelGGL(k_solve_team, dim3((unsigned)nteamc), dim3(64 * kTeam), lds_bytes_team(h->dn), h->team_stream, h->dn_dev, h->dp_dev, Wt, T);
      HIP_OK(hipGetLastError());
      HIP_OK(hipEventRecord(h->ev_join, h->team_stream));
      hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, h->stream, (const int *)(h->ws.counter + 1), (int)nteamc, 48000000LL);
      A.ncell = (int)(nc - nteamc); A.order += nteamc;
      h->last_team_cells += nteamc;
    }
    if (A.ncell > 0) {
      const long grid = std::max<long>(1, std::min<long>(slots - kTeam * nteamc, A.ncell));
      hipLaunchKernelGGL(k_solve, dim3((unsigned)grid), dim3(64), lds, h->stream, h->dn_dev, h->dp_dev, h->ws, A);
      HIP_OK(hipGetLastError());
    }
    if (nteamc > 0) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    if (park_max > 0 && A.ncell > 0 && !std::getenv("RACGPU_NO_RESUME")) { // (developer aid: leave the parked cells where they are)
      SolveArgs Rz = A;
      DevWork Wr = h->ws;
      Wr.counter = h->ws.counter + 5;
      hipLaunchKernelGGL(k_solve_team_resume, dim3((unsigned)park_max), dim3(64 * kTeam), lds_bytes_team(h->dn), h->stream, h->dn_dev, h->dp_dev, Wr, Rz);
      HIP_OK(hipGetLastError());
      HIP_OK(hipMemcpyAsync(h->parked_host, h->ws.counter + 4, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
  }
}

static void ensure_hc(racgpu_network *h) { if (!h->hc_dev) h->upload_hc(); }

// H2_form_use_moeq = .true.: the RATE COEFFICIENT branch (src/chemistry.f90:876-881) is built and pinned; chem_ode_f / chem_ode_jac then treat
// gH + gH as H + gH with extra terms on H and gH outside the declared sparsity structure (src/disk.f90:4625-4630, 4828-4840): not built, so
// every entry point that integrates or evaluates f / J refuses the switch instead of running the rate-equation form with the other coefficient
static void check_moeq(const racgpu_network *h, const DevParams &P, bool rates_only) {
  if (P.h2_moeq && !rates_only) throw std::runtime_error("H2_form_use_moeq = .true. is not implemented beyond the rate coefficients (see DESIGN.md, out of scope rows)");
  if (P.h2_moeq && h->dn.moeq_r61 == -2)
    throw std::runtime_error("H2_form_use_moeq: the network's adsorption reaction of H and desorption reaction of gH must precede its gH + gH reaction");
}

static void push_params(racgpu_network *h, const DevParams &P) {
  check_moeq(h, P, false); HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream)); // P lives on the caller's stack frame
}

static int evol_solve_batch_impl(racgpu_network *h, const racgpu_params *p, int64_t ncell, const double *cells, const double *hc, double *y, const double *t0,
                                 const int32_t *tol_j, double *t_final, int32_t *quality, int64_t *stats, double *record, double *touts,
                                 double *cell_out, int flags, int mem);

int racgpu_evol_solve_batch(racgpu_network *h, const racgpu_params *p, int64_t ncell, const double *cells, double *y, const double *t0,
                            const int32_t *tol_j, double *t_final, int32_t *quality, int64_t *stats, double *record, double *touts,
                            double *cell_out, int flags, int mem) {
  return evol_solve_batch_impl(h, p, ncell, cells, nullptr, y, t0, tol_j, t_final, quality, stats, record, touts, cell_out, flags, mem);
}

int racgpu_evolT_solve_batch(racgpu_network *h, const racgpu_params *p, int64_t ncell, const double *cells, const double *hc, double *y, const double *t0,
                             const int32_t *tol_j, double *t_final, int32_t *quality, int64_t *stats, double *record, double *touts,
                             double *cell_out, int flags, int mem) {
  if (!hc) return fail("hc must not be null (racgpu_evol_solve_batch is the fixed-T entry point)");
  return evol_solve_batch_impl(h, p, ncell, cells, hc, y, t0, tol_j, t_final, quality, stats, record, touts, cell_out, flags, mem);
}

static int evol_solve_batch_impl(racgpu_network *h, const racgpu_params *p, int64_t ncell, const double *cells, const double *hc, double *y, const double *t0,
                                 const int32_t *tol_j, double *t_final, int32_t *quality, int64_t *stats, double *record, double *touts,
                                 double *cell_out, int flags, int mem) {
  if (!h) return fail("null network");
  if (ncell <= 0) return 0;
  if (ncell > 0x7fffffffLL) return fail("ncell exceeds 2^31-1");
  if (!cells || !y) return fail("cells and y must not be null");
  return guarded([&] {
    h->upload();
    if (hc) ensure_hc(h);
    DevParams P = to_dev(p);
    push_params(h, P);
    const size_t nS = h->dn.nS;
    std::vector<double> trace_host;
    std::unique_ptr<DevBuf> dtrace;
    h->ws.trace = nullptr;
    if (P.debug_max_calls > 0) {
      trace_host.assign((size_t)P.debug_max_calls * 16 + 64 + 8 * (size_t)h->dn.npad + (size_t)h->dn.nnzJ, 0.0); // (second half and the tail: RG_DEBUG_NEWTON builds)
      dtrace = std::make_unique<DevBuf>(trace_host.data(), trace_host.size() * 8, RACGPU_MEM_HOST, true);
      h->ws.trace = (double *)dtrace->d;
    }
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, mem, true), dy(y, ncell * nS * 8, mem, true), dt0(t0, ncell * 8, mem, true),
        dj(tol_j, ncell * 4, mem, true), dt(t_final, ncell * 8, mem, false), dq(quality, ncell * 4, mem, false),
        ds(stats, ncell * RACGPU_NSTAT * 8, mem, false), drec(record, (size_t)ncell * P.n_record * (nS + 1) * 8, mem, false),
        dto(touts, (size_t)ncell * P.n_record * 8, mem, false), dout(cell_out, (size_t)ncell * RACGPU_NOUT * 8, mem, true),
        dhc(hc, (size_t)ncell * kNHC * 8, mem, true);
    int *marker_host = nullptr;
    h->ws.marker = nullptr;
    const char *dbgwait = std::getenv("RACGPU_DEBUG_WAIT");
    if (dbgwait) {
      HIP_OK(hipHostMalloc((void **)&marker_host, 64, hipHostMallocMapped));
      *marker_host = 0;
      HIP_OK(hipHostGetDevicePointer((void **)&h->ws.marker, marker_host, 0));
    }
    PassBufs B{(const double *)dc.d, (double *)dy.d, (const double *)dt0.d, (const int *)dj.d, (double *)dt.d, (int *)dq.d, (long long *)ds.d,
               (double *)drec.d, (double *)dto.d, (double *)dout.d, (const double *)dhc.d};
    solve_pass(h, P, (long)ncell, B, flags, true, true);
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    if (dbgwait) { // developer aid: watch the progress word; give up (and leave the process) instead of hanging
      const double limit = std::atof(dbgwait);
      double waited = 0.0; int last = -1;
      while (hipStreamQuery(h->stream) == hipErrorNotReady) {
        struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr); waited += 0.1;
        const int m = *(volatile int *)marker_host;
        if (m != last) { std::fprintf(stderr, "[racgpu marker] t=%.1fs marker=%d\n", waited, m); last = m; }
        if (waited > limit) { std::fprintf(stderr, "[racgpu marker] still running after %.1fs at marker=%d: giving up\n", waited, m); std::fflush(stderr); _exit(3); }
      }
      std::fprintf(stderr, "[racgpu marker] finished after %.1fs, last marker=%d\n", waited, *(volatile int *)marker_host);
    }
    if (mem == RACGPU_MEM_HOST) {
      HIP_OK(hipStreamSynchronize(h->stream));
      dy.copy_out(); dt.copy_out(); dq.copy_out(); ds.copy_out(); drec.copy_out(); dto.copy_out(); dout.copy_out();
    }
    if (P.debug_max_calls > 0) {
      HIP_OK(hipStreamSynchronize(h->stream));
      dtrace->copy_out();
      for (int i = 0; i < P.debug_max_calls; ++i) {
        const double *tr = &trace_host[(size_t)i * 8];
        std::fprintf(stderr, "[racgpu trace] call %3d tn=%.6e h=%.6e hu=%.6e nq=%g kflag=%g nst=%g nfe=%g nje/nlu=%g\n", i, tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], tr[6], tr[7]);
#ifdef RG_DEBUG_NEWTON
        if (i == 0) if (const char *fn = std::getenv("RACGPU_DEBUG_DUMP_FILE")) { // the dumped corrector pass: header[64], 8 vectors of npad, P[nnzJ]
          if (FILE *f = std::fopen(fn, "wb")) {
            const size_t off = (size_t)P.debug_max_calls * 16, cnt = 64 + 8 * (size_t)h->dn.npad + (size_t)h->dn.nnzJ;
            std::fwrite(&trace_host[off], sizeof(double), cnt, f); std::fclose(f);
          }
        }
        const double *tx = &trace_host[(size_t)P.debug_max_calls * 8 + (size_t)i * 8];
        std::fprintf(stderr, "[racgpu newton] call %3d m=%g del=%.3e %.3e %.3e worst(idx+1e-3*w)=%.4f %.4f %.4f w0=%.3e\n", i, tx[0], tx[1], tx[2], tx[3], tx[4], tx[5], tx[6], tx[7]);
#endif
      }
    }
  });
}

void racgpu_hc_config_default(racgpu_hc_config *c) { // the reference's template (README.md:135-156)
  std::memset(c, 0, sizeof *c);
  c->heating_eff_chem = 0.3; c->heating_eff_H2form = 0.5; c->heating_eff_phd_H2 = 1.0; c->heating_eff_phd_H2O = 0.5; c->heating_eff_phd_OH = 0.5;
  c->cooling_gg_coeff = 1.0; c->base_alpha = 0.01;
  c->use_chemicalheatingcooling = 1; c->use_Xray_heating = 1; c->use_phdheating_H2 = 1; c->use_phdheating_H2OOH = 1; c->use_mygasgraincooling = 1;
  c->may_switch_T = 1;
}

int racgpu_heating_cooling_load(racgpu_network *h, const racgpu_hc_config *cfg, const char *enthalpy_file, const char *neufeld_tables,
                                const char *nii_lut, const char *siii_lut, const char *feii_lut) {
  if (!h || !cfg || !enthalpy_file || !neufeld_tables || !nii_lut || !siii_lut || !feii_lut) return fail("null argument");
  static_assert(sizeof(racgpu_hc_config) == sizeof(HcConfig), "racgpu_hc_config and HcConfig must have the same layout");
  return guarded([&] {
    h->hhc.loaded = false;
    load_species_enthalpies(h->net, enthalpy_file, h->hhc);
    load_neufeld_tables(neufeld_tables, h->hhc);
    load_ion_lut(nii_lut, h->hhc.nii); load_ion_lut(siii_lut, h->hhc.siii); load_ion_lut(feii_lut, h->hhc.feii);
    std::memcpy(&h->hcfg, cfg, sizeof(HcConfig));
    h->hhc.loaded = true;
    if (h->hc_dev) { (void)hipFree(h->hc_dev); h->hc_dev = nullptr; } // re-uploaded by the next compute call
  });
}

int racgpu_heat_reactions(const racgpu_network *h, int32_t *n, int32_t *rxn, double *heat) {
  if (!h) return fail("null network");
  if (!h->hhc.loaded) return fail("racgpu_heating_cooling_load has not been called");
  if (n) *n = (int32_t)h->hhc.heat_rxn.size();
  for (size_t i = 0; i < h->hhc.heat_rxn.size(); ++i) { if (rxn) rxn[i] = h->hhc.heat_rxn[i] + 1; if (heat) heat[i] = h->hhc.heat_val[i]; }
  return 0;
}


int racgpu_evolT_hooks(racgpu_network *h, const racgpu_params *p, const double *cells, const double *hc, int64_t ncell, const double *y,
                       double *ydot, double *terms, double *tcol, double *trow) {
  if (!h) return fail("null network");
  if (!cells || !hc || !y || !ydot || !terms) return fail("cells, hc, y, ydot and terms must not be null");
  if ((tcol == nullptr) != (trow == nullptr)) return fail("tcol and trow go together");
  return guarded([&] {
    h->upload();
    ensure_hc(h);
    DevParams P = to_dev(p);
    push_params(h, P);
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dh(hc, (size_t)ncell * kNHC * 8, RACGPU_MEM_HOST, true),
        dy(y, ncell * (nS + 1) * 8, RACGPU_MEM_HOST, true), dd(ydot, ncell * (nS + 1) * 8, RACGPU_MEM_HOST, false),
        dt(terms, (size_t)ncell * HC_NTERMS * 8, RACGPU_MEM_HOST, false), dtc(tcol, ncell * (nS + 1) * 8, RACGPU_MEM_HOST, false),
        dtr(trow, (size_t)ncell * 10 * 8, RACGPU_MEM_HOST, false);
    h->ensure_workspace((long)ncell, (long)ncell);
    hipLaunchKernelGGL(k_evolT_hooks, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, h->ws, (const DevHC *)h->hc_dev,
                       (const double *)dc.d, (const double *)dh.d, (const double *)dy.d, tcol ? (getenv("RACGPU_DEBUG_FULL_TROW") ? 2 : 1) : 0, (double *)dd.d, (double *)dt.d, (double *)dtc.d, (double *)dtr.d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dd.copy_out(); dt.copy_out(); dtc.copy_out(); dtr.copy_out();
  });
}

int racgpu_set_co_shielding_table(racgpu_network *h, int32_t nrow, int32_t ncol, const double *logN_H2, const double *logN_12CO, const double *f) {
  if (!h) return fail("null network");
  return guarded([&] {
    require_gpu();
    for (double **q : {&h->co_logNH2, &h->co_logNCO, &h->co_lnf}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    h->co_nrow = h->co_ncol = 0;
    if (!f || nrow < 2 || ncol < 2) return; // cleared
    if (!logN_H2 || !logN_12CO) throw std::runtime_error("CO shielding table: axes missing");
    for (int i = 1; i < nrow; ++i) if (!(logN_H2[i] > logN_H2[i - 1])) throw std::runtime_error("CO shielding table: logN_H2 must ascend");
    for (int j = 1; j < ncol; ++j) if (!(logN_12CO[j] > logN_12CO[j - 1])) throw std::runtime_error("CO shielding table: logN_12CO must ascend");
    std::vector<double> lnf((size_t)nrow * ncol);
    for (size_t q = 0; q < lnf.size(); ++q) { if (!(f[q] > 0.0)) throw std::runtime_error("CO shielding table: f must be positive"); lnf[q] = std::log(f[q]); }
    auto put = [&](double **d, const double *src, size_t n) { HIP_OK(hipMalloc((void **)d, n * 8)); HIP_OK(hipMemcpy(*d, src, n * 8, hipMemcpyHostToDevice)); };
    put(&h->co_logNH2, logN_H2, nrow); put(&h->co_logNCO, logN_12CO, ncol); put(&h->co_lnf, lnf.data(), lnf.size());
    h->co_nrow = nrow; h->co_ncol = ncol;
  });
}

int racgpu_set_star_rays(racgpu_network *h, int64_t ncell, const int32_t *inner, const double *ds) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->ray_inner.clear(); h->ray_ds.clear();
    if (!inner || ncell <= 0) return; // cleared
    if (!ds) throw std::runtime_error("star rays: ds missing");
    for (int64_t i = 0; i < ncell; ++i) {
      if (inner[i] < -1 || inner[i] >= ncell || inner[i] == i) throw std::runtime_error("star rays: inner must name another cell or be -1");
      if (!(ds[i] >= 0.0)) throw std::runtime_error("star rays: ds must be >= 0");
    }
    h->ray_inner.assign(inner, inner + ncell); h->ray_ds.assign(ds, ds + ncell);
  });
}

int racgpu_star_ray_timeouts(const racgpu_network *h) { return h ? h->last_ray_timeouts : -1; }

int racgpu_column_sweep(racgpu_network *h, const racgpu_params *p, int64_t ncolumn, const int32_t *col_ptr, const int32_t *col_cells,
                        int64_t ncell, double *cells, double *y, const double *dz, double dv_turb, double *t_final, int32_t *quality,
                        int64_t *stats, double *cell_out, int mem) {
  if (!h) return fail("null network");
  if (ncolumn <= 0 || ncell <= 0) return 0;
  if (ncell > 0x7fffffffLL) return fail("ncell exceeds 2^31-1");
  if (!cells || !y || !col_ptr || !col_cells || !dz) return fail("cells, y, col_ptr, col_cells and dz must not be null");
  if (!(dv_turb > 0.0)) return fail("dv_turb must be positive");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    push_params(h, P);
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, mem, true), dy(y, ncell * nS * 8, mem, true), dd(dz, ncell * 8, mem, true),
        dp(col_ptr, (ncolumn + 1) * 4, mem, true), dl(col_cells, ncell * 4, mem, true), dt(t_final, ncell * 8, mem, false),
        dq(quality, ncell * 4, mem, false), ds(stats, ncell * RACGPU_NSTAT * 8, mem, false),
        dout(cell_out, (size_t)ncell * RACGPU_NOUT * 8, mem, true);
    if (mem == RACGPU_MEM_HOST) { // (device buffers are the caller's responsibility)
      if (col_ptr[0] != 0 || col_ptr[ncolumn] != ncell) throw std::runtime_error("column sweep: col_ptr must run from 0 to ncell");
      for (int64_t c = 0; c < ncolumn; ++c) if (col_ptr[c + 1] <= col_ptr[c]) throw std::runtime_error("column sweep: empty column");
      std::vector<char> seen((size_t)ncell, 0); // two columns holding the same cell would integrate it at once and corrupt each other's output
      for (int64_t i = 0; i < ncell; ++i) {
        if (col_cells[i] < 0 || col_cells[i] >= ncell) throw std::runtime_error("column sweep: cell index out of range");
        if (seen[col_cells[i]]) throw std::runtime_error("column sweep: col_cells must be a permutation of the cells (duplicate index)");
        seen[col_cells[i]] = 1;
      }
    }
    // rays to the star: a cell waits (on the device) for the cell its ray passes through next, so that cell must belong to a column
    // that is taken from the queue EARLIER (columns are started in index order and never wait for later ones: no deadlock)
    const bool rays = !h->ray_inner.empty();
    std::unique_ptr<DevBuf> dri, drs;
    struct DevMem { void *d = nullptr; ~DevMem() { if (d) (void)hipFree(d); } } raymem;
    double *rayN = nullptr; int *rayflags = nullptr;
    if (rays) {
      if ((int64_t)h->ray_inner.size() != ncell) throw std::runtime_error("column sweep: racgpu_set_star_rays was given another number of cells");
      if (mem != RACGPU_MEM_HOST) throw std::runtime_error("column sweep: star rays need host buffers (the column order is checked on the host)");
      std::vector<int> colof((size_t)ncell, 0);
      for (int64_t c = 0; c < ncolumn; ++c) for (int q = col_ptr[c]; q < col_ptr[c + 1]; ++q) colof[col_cells[q]] = (int)c;
      for (int64_t i = 0; i < ncell; ++i)
        if (h->ray_inner[i] >= 0 && colof[h->ray_inner[i]] >= colof[i]) throw std::runtime_error("column sweep: a star ray must lead into a column of lower index");
      dri = std::make_unique<DevBuf>(h->ray_inner.data(), (size_t)ncell * 4, RACGPU_MEM_HOST, true);
      drs = std::make_unique<DevBuf>(h->ray_ds.data(), (size_t)ncell * 8, RACGPU_MEM_HOST, true);
      HIP_OK(hipMalloc(&raymem.d, (size_t)ncell * 4 * 8 + (size_t)(ncell + 1) * 4));
      rayN = (double *)raymem.d;
      rayflags = (int *)(rayN + (size_t)ncell * 4);
      HIP_OK(hipMemsetAsync(rayN, 0, (size_t)ncell * 4 * 8 + (size_t)(ncell + 1) * 4, h->stream));
    }
    h->ws.trace = nullptr; h->ws.marker = nullptr;
    const long grid = std::min<long>(ncolumn, 2L * h->cu_count); // two teams per CU by registers
    h->ensure_workspace(grid, (long)ncell);
    HIP_OK(hipMemsetAsync(h->ws.counter, 0, 8 * sizeof(int), h->stream));
    HIP_OK(hipMemsetAsync(h->parked_host, 0, sizeof(int), h->stream));
    h->last_team_cells = ncell;
    HIP_OK(hipEventRecord(h->ev0, h->stream));
    SolveArgs A{};
    A.ncell = (int)ncell; A.flags = 0; A.cells = (const double *)dc.d; A.cells_rw = (double *)dc.d; A.yio = (double *)dy.d;
    A.t_final = (double *)dt.d; A.quality = (int *)dq.d; A.stats = (long long *)ds.d; A.cell_out = (double *)dout.d;
    A.col_ptr = (const int *)dp.d; A.col_cells = (const int *)dl.d; A.ncolumn = (int)ncolumn; A.dz = (const double *)dd.d; A.dv_turb = dv_turb;
    auto find = [&](const char *nm) { for (int i = 0; i < h->net.nS; ++i) if (h->net.names[i] == nm) return i; return -1; };
    A.i_H2O = find("H2O"); A.i_OH = find("OH"); A.i_CO = find("CO");
    A.co_logNH2 = h->co_logNH2; A.co_logNCO = h->co_logNCO; A.co_lnf = h->co_lnf; A.co_nrow = h->co_nrow; A.co_ncol = h->co_ncol;
    if (rays) { A.ray_inner = (const int *)dri->d; A.ray_ds = (const double *)drs->d; A.ray_N = rayN; A.ray_done = rayflags; A.ray_timeouts = rayflags + ncell; }
    DevWork Wc = h->ws;
    Wc.counter = h->ws.counter + 6;
    hipLaunchKernelGGL(k_solve_columns, dim3((unsigned)grid), dim3(64 * kTeam), lds_bytes_team(h->dn), h->stream, h->dn_dev, h->dp_dev, Wc, A);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    if (mem == RACGPU_MEM_HOST) {
      HIP_OK(hipStreamSynchronize(h->stream));
      dc.copy_out(); dy.copy_out(); dt.copy_out(); dq.copy_out(); ds.copy_out(); dout.copy_out();
      h->last_ray_timeouts = 0;
      if (rays) {
        HIP_OK(hipMemcpy(&h->last_ray_timeouts, rayflags + ncell, sizeof(int), hipMemcpyDeviceToHost));
        if (h->last_ray_timeouts) throw std::runtime_error("column sweep: " + std::to_string(h->last_ray_timeouts) + " cell(s) gave up waiting for the cell on their ray to the star");
      }
    }
  });
}

int racgpu_solve_batch(racgpu_network *h, const racgpu_params *p, int64_t ncell, const double *cells, double *y, double *t_final,
                       int32_t *quality, int64_t *stats, double *record, double *touts, int mem) {
  return racgpu_evol_solve_batch(h, p, ncell, cells, y, nullptr, nullptr, t_final, quality, stats, record, touts, nullptr, 0, mem);
}

int racgpu_calc_cells(racgpu_network *h, const racgpu_params *p, int32_t nlocal_iter, int64_t ncell, const double *cells, double *y,
                      double *t_final, int32_t *quality, int64_t *stats, double *cell_out, int mem) {
  if (!h) return fail("null network");
  if (ncell <= 0) return 0;
  if (ncell > 0x7fffffffLL) return fail("ncell exceeds 2^31-1");
  if (!cells || !y) return fail("cells and y must not be null");
  if (nlocal_iter < 1) return fail("nlocal_iter must be >= 1");
  return guarded([&] {
    h->upload();
    racgpu_params pj = *p;
    pj.tol_policy_j = 1;
    DevParams P = to_dev(&pj);
    push_params(h, P);
    h->ws.trace = nullptr; h->ws.marker = nullptr;
    const size_t nS = h->dn.nS;
    const long n = (long)ncell;
    // the loop needs t_final, quality, stats and cell_out whether or not the caller wants them
    auto dalloc = [&](size_t bytes) { void *d = nullptr; HIP_OK(hipMalloc(&d, std::max<size_t>(bytes, 8))); return d; };
    struct Scoped { std::vector<void *> v; ~Scoped() { for (void *q : v) (void)hipFree(q); } } own;
    auto scratch = [&](size_t bytes) { void *d = dalloc(bytes); own.v.push_back(d); return d; };
    DevBuf dc(cells, (size_t)n * RACGPU_NPAR * 8, mem, true), dy(y, n * nS * 8, mem, true), dt(t_final, n * 8, mem, false),
        dq(quality, n * 4, mem, false), ds(stats, n * RACGPU_NSTAT * 8, mem, false), dout(cell_out, (size_t)n * RACGPU_NOUT * 8, mem, true);
    double *t_d = dt.d ? (double *)dt.d : (double *)scratch(n * 8);
    int *q_d = dq.d ? (int *)dq.d : (int *)scratch(n * 4);
    long long *s_d = ds.d ? (long long *)ds.d : (long long *)scratch(n * RACGPU_NSTAT * 8);
    double *o_d = dout.d ? (double *)dout.d : (double *)scratch((size_t)n * RACGPU_NOUT * 8);
    PassBufs B{(const double *)dc.d, (double *)dy.d, nullptr, nullptr, t_d, q_d, s_d, nullptr, nullptr, o_d};
    solve_pass(h, P, n, B, 0, true, true);
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    if (nlocal_iter > 1) {
      std::vector<double> tf((size_t)n), cl((size_t)n * RACGPU_NPAR);
      std::vector<int> ql((size_t)n);
      std::vector<long long> st((size_t)n * RACGPU_NSTAT);
      std::vector<char> done((size_t)n, 0);
      HIP_OK(hipMemcpyAsync(cl.data(), dc.d, cl.size() * 8, hipMemcpyDeviceToHost, h->stream));
      for (int j = 2; j <= nlocal_iter; ++j) {
        HIP_OK(hipMemcpyAsync(tf.data(), t_d, tf.size() * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipMemcpyAsync(ql.data(), q_d, ql.size() * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipMemcpyAsync(st.data(), s_d, st.size() * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_OK(hipStreamSynchronize(h->stream));
        // who goes on (src/disk.f90:1706-1714, 1724-1728, 1784-1788): the previous iteration got past the one before it,
        // produced a usable record, and ended flagged before half of the cell's t_max
        std::vector<int> sel;
        for (long c = 0; c < n; ++c) {
          if (done[c]) continue;
          const long long *s = &st[(size_t)c * RACGPU_NSTAT];
          const double tmax_this = cl[(size_t)c * RACGPU_NPAR + RACGPU_P_TMAX] > 0.0 ? cl[(size_t)c * RACGPU_NPAR + RACGPU_P_TMAX] : p->t_max;
          const bool stopped = (int)s[RACGPU_S_NITER] < j - 1 /* "does not proceed" left NITER behind */ || s[RACGPU_S_ISAV] <= 1 ||
                               ql[c] == 0 || tf[c] >= 0.5 * tmax_this;
          if (stopped) { done[c] = 1; continue; }
          sel.push_back((int)c);
        }
        if (sel.empty()) break;
        const long m = (long)sel.size();
        int *sel_d = (int *)scratch(m * 4);
        double *cells_c = (double *)scratch((size_t)m * RACGPU_NPAR * 8), *y_c = (double *)scratch(m * nS * 8), *t0_c = (double *)scratch(m * 8),
               *tf_c = (double *)scratch(m * 8), *out_c = (double *)scratch((size_t)m * RACGPU_NOUT * 8);
        int *q_c = (int *)scratch(m * 4);
        long long *st_c = (long long *)scratch((size_t)m * RACGPU_NSTAT * 8);
        HIP_OK(hipMemcpyAsync(sel_d, sel.data(), m * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_gather_pass, dim3((unsigned)m), dim3(64), 0, h->stream, (int)m, sel_d, (int)nS, (const double *)dc.d, (const double *)dy.d,
                           (const double *)t_d, cells_c, y_c, t0_c);
        HIP_OK(hipGetLastError());
        racgpu_params pk = *p;
        pk.tol_policy_j = j;
        DevParams Pk = to_dev(&pk);
        push_params(h, Pk); // also keeps 'sel' alive until its copy has completed
        PassBufs Bc{cells_c, y_c, t0_c, nullptr, tf_c, q_c, st_c, nullptr, nullptr, out_c};
        solve_pass(h, Pk, m, Bc, RACGPU_F_RECTIFY, false, false);
        hipLaunchKernelGGL(k_merge_pass, dim3((unsigned)m), dim3(64), 0, h->stream, (int)m, sel_d, (int)nS, j, (const double *)y_c, (const double *)tf_c,
                           (const int *)q_c, (const long long *)st_c, (const double *)out_c, (double *)dy.d, t_d, q_d, s_d, o_d);
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventRecord(h->ev1, h->stream));
      }
    }
    HIP_OK(hipStreamSynchronize(h->stream));
    if (mem == RACGPU_MEM_HOST) { dy.copy_out(); dt.copy_out(); dq.copy_out(); ds.copy_out(); dout.copy_out(); }
  });
}

int racgpu_rectify_abundances(const racgpu_network *h, int64_t ncell, double *y) {
  if (!h) return fail("null network");
  const HostNetwork &n = h->net;
  const int iE = n.idx10[2];
  if (iE <= 0) return fail("network has no E-");
  for (int64_t c = 0; c < ncell; ++c) {
    double *yc = y + c * n.nS, q = 0.0;
    for (int i = 0; i < n.nS; ++i) q += yc[i] * (double)n.elements[i][0];
    yc[iE - 1] = yc[iE - 1] + q;
  }
  return 0;
}

} // extern "C"
