// engine_integrate.hpp -- per-wave variable-order BDF integration of one cell (device code).
//
// What is mirrored, and where it lives in the reference:
//   output-time loop, restart/error policy, quality flags   chem_evol_solve            src/chemistry.f90:391-588
//   hand-off record (last record without NaN)               calc_this_cell             src/disk.f90:1716-1733
//   tolerance loosening after ISTATE -4/-5                  ode_solver_error_handling  src/chemistry.f90:297-377
//   driver blocks A-H for ITASK=4 (TCRIT=HMAX=t_max)        DLSODES                    src/opkdmain.f:3069-3588
//   one step in Nordsieck form, order/step selection        DSTODE                     src/opkda1.f:746-1124
//   P = I - h*el0*J: evaluate or rescale, then factor       DPRJS                      src/opkda1.f:1735-1838
//   interpolation at tout                                   DINTDY (K = 0)             src/opkda1.f:236-263
// The T slot (NEQ = nS+1) is inert at fixed temperature: its ydot, Jacobian row and column are zero, so it is
// carried as two scalars (value and weight) that only enter the "too much accuracy" norm and the divisor of
// every RMS norm, exactly as in the reference.
// Gas-temperature co-evolution (chemsol_params%evolT; template parameter ET, kernel k_solve_T): the T entry of every
// NEQ-vector is a scalar in LDS (struct TSlot) that takes part in every vector operation; the rate coefficients are
// recomputed at the iterate's T before every f(y) (chem_ode_f, reference src/disk.f90:4577-4580), dT/dt comes from
// dev_heating_cooling (realtime_heating_cooling_rate, :4664-4741), the T row and T column of the Jacobian from finite
// differences (chem_ode_jac, :4878-4899) and the Newton matrix is solved as a bordered system around the species block's
// LU (the reference orders T into its minimum-degree ordering; without pivoting the two agree to rounding).  With ET = false
// nothing of this is compiled: k_solve is the fixed-T kernel of rounds 1 and 2 to the last instruction.
#pragma once
#include "engine_device.hpp"
#include "engine_hc.hpp"

namespace racgpu {

#ifndef RG_VEC_TRIP
#define RG_VEC_TRIP 8 // blocks of 64 per trip of the integrator's vector loops (vec_trips)
#endif

struct CellCtx {
  double *y, *savf, *wx;                                       // LDS, nS doubles each: iterate, f(y), linear-solver work vector
  double *acor, *ewt;                                          // HBM, npad each: accumulated correction, inverse error weights (elementwise use only)
  double *yh, *Pv, *Lv, *Uv, *Dinv, *rates, *rtol, *atol;      // this cell's HBM slices
  int lane, n, npad;
  int nteam;      // waves working on this cell (1, or 4 in k_solve_team: wave 0 holds this context, the others serve it)
  int *marker;    // developer aid: host-visible progress word, or null
  // ET only: the cell's record and heating/cooling record, the heating/cooling tables, and two more HBM vectors of the slot:
  // the T column of P (rows = species) and (species block of P)^-1 times it
  const double *cell, *hcrec; const DevHC *hc; double *Pb, *zb;
  const DevParams *prm;
};

// Per-cell constants and the per-phase cycle counters live in LDS and are read where they are used: as kernel-long
// register values they (with hoisted constants) took ~100 VGPRs away from the factorisation.  Declared volatile and
// accessed by name, so every access is one ds_read/ds_write on the LDS address space (a pointer or reference to a
// __shared__ object is generic, and volatile accesses through it become flat_load/flat_store).
struct WaveConst {
  double nsite;   // ratioDust2HnucNum * SitesPerGrain
  double Tgas, rT, aT;
  double inv_neq; // 1 / (nS + 1)
  long long cyc[8]; // shader-clock cycles per phase (s_memtime): f(y), Jacobian, LU, triangular solves; then the LU split:
                    // column scatter, LDS pivots, register (dense) pivots, column finish
};
enum { CYC_RHS = 0, CYC_JAC, CYC_LU, CYC_SOLVE, CYC_LU_PART };
static __shared__ volatile WaveConst g_wc;
// ET: the T entry of the NEQ-vectors (iterate, f, accumulated correction, inverse weight, tolerances, Nordsieck columns), the
// border of the Newton matrix (T row at the ten special species, T-T entry, Schur complement), R_H2_form_rate_coeff of the last
// chem_cal_rates call, and the switches chemsol_params%evolT / %maySwitchT / %t_scale_tol
struct TSlot {
  double y, savf, acor, ewt, rtol, atol, yh[6];
  double Pc[10], Pd, schur, rh2, t_scale_tol;
  int evolT, maySwitchT, freeze_rec;
};
static __shared__ volatile TSlot g_T;
// wave 0's requests to the other waves of its team (k_solve_team): written before a barrier, read after it
enum { T_EXIT = 0, T_LU = 1, T_JAC = 2 };
struct TeamCtl { int cmd, fail, cell, slot; double con; }; // cell: wave 0's current cell (its rate vector); slot: its workspace slot; con: -h*el0 of the Jacobian request
static __shared__ volatile TeamCtl g_team;
RG_DEV void cyc_add(int k, long long d) { g_wc.cyc[k] = g_wc.cyc[k] + d; }

RG_DEV long long dev_clock() { return (long long)__builtin_readcyclecounter(); }

RG_DEV void dev_mark(const CellCtx &c, int id) {
  if (c.marker && c.lane == 0) __hip_atomic_store(c.marker, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr double kUround = 2.220446049250313e-16; // DUMACH()
constexpr double kCcmax = 0.3, kCcmxj = 0.2, kPsmall = 1000.0 * kUround, kRbig = 0.01 / kPsmall;
constexpr int kMaxord = 5, kMaxcor = 3, kMsbp = 20, kMxncf = 10, kMsbj = 50;

#ifdef RG_DEBUG_NEWTON
struct NewtonDbg { double del[3], big[3]; int idx[3], m; };
static __shared__ volatile NewtonDbg g_dbg;
#endif
struct Lsodes { // the scalars ODEPACK keeps in COMMON /DLS001/ and /DLSS01/ plus the driver's SAVEd locals
  double conit, crate, hold, rmax, el0, h, hmxi, hu, rc, tn;
  double con0, conmin, tcrit, h0;
  int ialth, ipup, lmax, nslp, icf, ierpj, jcur, jstart, kflag, l, nq, nst, nfe, nje, nqu;
  int iplost, nslj, nlu, init, nslast, imxer, mxstep;
  long long qsum; int nfail;
  int ncalls; double *trace; int trace_cap; // developer aid
};

template <bool ET = false, typename V>
RG_DEV double dev_vnorm(const CellCtx &c, V v, double vT = 0.0) { // DVNORM over NEQ = nS+1 entries (fixed T: the T entry is zero)
  double s = 0.0;
  const rsrc_t bE = mkbuf(c.ewt);
  vec_trips<RG_VEC_TRIP, double>(c.n, c.npad, [&](int i0) { return bload_f64(bE, c.lane * 8, i0 * 8); },
                       [&](int i0, double e) { const int i = i0 + c.lane; if (i < c.n) { const double q = v(i) * e; s += q * q; } });
  if constexpr (ET) { const double q = vT * g_T.ewt; return sqrt((wave_sum(s) + q * q) * g_wc.inv_neq); }
  return sqrt(wave_sum(s) * g_wc.inv_neq);
}

RG_DEV void dev_set_order(const DevParams &P, Lsodes &s) { // DSTODE label 150
  s.rc = s.rc * P.elco[s.nq][1] / s.el0;
  s.el0 = P.elco[s.nq][1];
  s.conit = 0.5 / (s.nq + 2);
}

// The wave's HBM vectors (Nordsieck columns, tolerances, hand-off record) are addressed through buffer resources:
// base in SGPRs, lane part lane*8 in ONE VGPR, everything else (column, block of 64) in the scalar/immediate offset.
// With plain pointers hipcc keeps one 64-bit per-lane address per (column, block) alive across the whole kernel.
// A column holds npad = 64*ceil(n/64) doubles, so whole blocks may be read and written without a bound test where
// only HBM is involved; entries >= n of a column are never used.
RG_DEV int col_off(const CellCtx &c, int j) { return j * c.npad * 8; } // byte offset of Nordsieck column j (0-based)

// Block loops over these vectors run B blocks of 64 per trip with all loads of a trip issued before anything is used: a loop
// whose trip count the compiler does not know is not unrolled, and every block would otherwise wait out its own round trip to
// HBM/L2.  ld(i0) fetches what block i0 needs (blocks past the end re-read the last one), use(i0, v) is called for i0 < n.
template <int B, typename T, typename LD, typename USE>
RG_DEV void vec_trips(int n, int npad, LD ld, USE use) {
  for (int c0 = 0; c0 < n; c0 += 64 * B) {
    T v[B];
#pragma unroll
    for (int u = 0; u < B; ++u) v[u] = ld(min(c0 + 64 * u, npad - 64));
#pragma unroll
    for (int u = 0; u < B; ++u) if (c0 + 64 * u < n) use(c0 + 64 * u, v[u]);
  }
}
struct D2 { double a, b; };
struct D3 { double a, b, c; };
struct DCols { double v[kMaxord + 2]; }; // the Nordsieck columns of one block (+ one more vector)

template <bool ET = false>
RG_DEV void dev_rescale(const CellCtx &c, Lsodes &s, double rh, bool apply_hmin) { // DSTODE labels 170/175
  if (apply_hmin) rh = fmax(rh, 0.0); // RH = MAX(RH, HMIN/ABS(H)) with HMIN = 0
  rh = fmin(rh, s.rmax);
  rh = rh / fmax(1.0, fabs(s.h) * s.hmxi * rh);
  const rsrc_t bY = mkbuf(c.yh);
  const int l8 = c.lane * 8;
  double r = 1.0;
  for (int j = 2; j <= s.l; ++j) {
    r = r * rh;
    const int co = col_off(c, j - 1);
    vec_trips<RG_VEC_TRIP, double>(c.n, c.npad, [&](int i0) { return bload_f64(bY, l8, co + i0 * 8); }, [&](int i0, double v) { bstore_f64(bY, l8, co + i0 * 8, v * r); });
    if constexpr (ET) g_T.yh[j - 1] = g_T.yh[j - 1] * r;
  }
  s.h = s.h * rh; s.rc = s.rc * rh; s.ialth = s.l;
}

// YH <- YH * Pascal (forward) or its inverse; DSTODE :865-874, :956-962
template <bool ET = false>
RG_DEV void dev_pascal(const CellCtx &c, const Lsodes &s, bool forward, bool to_y = false) { // to_y: the new first column goes to y (LDS) as well
  const int nq = s.nq;
  const rsrc_t bY = mkbuf(c.yh);
  const int l8 = c.lane * 8;
  vec_trips<2, DCols>(c.n, c.npad,
    [&](int i0) {
      DCols d;
#pragma unroll
      for (int j = 0; j <= kMaxord; ++j) d.v[j] = (j <= nq) ? bload_f64(bY, l8, col_off(c, j) + i0 * 8) : 0.0;
      return d;
    },
    [&](int i0, DCols d) {
#pragma unroll
      for (int jb = 1; jb <= kMaxord; ++jb) {
        if (jb > nq) break;
#pragma unroll
        for (int j = 0; j < kMaxord; ++j)
          if (j >= nq - jb && j < nq) d.v[j] = forward ? d.v[j] + d.v[j + 1] : d.v[j] - d.v[j + 1];
      }
#pragma unroll
      for (int j = 0; j < kMaxord; ++j)
        if (j < nq) bstore_f64(bY, l8, col_off(c, j) + i0 * 8, d.v[j]);
      if (to_y && i0 + c.lane < c.n) c.y[i0 + c.lane] = d.v[0];
    });
  if constexpr (ET) { // the same triangle on the T entries
    double t[kMaxord + 1];
    for (int j = 0; j <= kMaxord; ++j) t[j] = g_T.yh[j];
    for (int jb = 1; jb <= nq; ++jb)
      for (int j = nq - jb; j < nq; ++j) t[j] = forward ? t[j] + t[j + 1] : t[j] - t[j + 1];
    for (int j = 0; j < nq; ++j) g_T.yh[j] = t[j];
    if (to_y) g_T.y = t[0];
  }
}

static __shared__ double g_rate_tab[kRateTab]; // dev_rates' table of pow / exp values (k_solve_T and its test hook only)

// f(y) into savf.  ET and the cell's T still evolving: chem_ode_f's three steps (reference src/disk.f90:4569-4659) -- rate
// coefficients at the iterate's T, the species part, dT/dt.  The rate vector and R_H2_form_rate_coeff stay as this call leaves them.
template <bool ET>
RG_DEV void dev_f(const DevNet &N, const CellCtx &c, double *ydot, double *ydotT = nullptr, const double *Tat = nullptr) {
  if constexpr (ET) {
    if (g_T.evolT) {
      double T = Tat ? *Tat : g_T.y;
      dev_rates(N, *c.prm, c.cell, c.rates, c.lane, (double *)&g_T.rh2, &T, g_rate_tab);
      dev_rhs(N, c.rates, g_wc.nsite, gptr(N.r_C), c.y, ydot, c.lane);
      const double td = dev_heating_cooling(N, *(const RG_GLOBAL DevHC *)c.hc, c.cell, c.hcrec, c.y, T, c.rates, g_T.rh2, c.lane);
      if (ydotT) *ydotT = td; else g_T.savf = td;
      return;
    }
    if (ydotT) *ydotT = 0.0; else g_T.savf = 0.0;
  }
  dev_rhs(N, c.rates, g_wc.nsite, gptr(N.r_C), c.y, ydot, c.lane);
}

// chem_ode_jac's finite differences (reference src/disk.f90:4878-4899), taken after the species columns as DPRJS calls them (j = 1..NEQ),
// scaled into the border of P = I + con J.  T row: d(dT/dt)/dy_j for the ten special species with y_j >= 0, step 1e-2 y_j + 1e-6 D2H; the
// base value is the dT/dt of the f(y) that preceded this call (same y, same rates: g_T.savf).  T column: (f(y, T + dT) - f(y, T)) / dT with
// dT = 1e-2 T + 1, f(y, T) being savf; the rate vector is left at T + dT, as in the reference.  wx (LDS) is used as scratch.
// full_rows (test hook): every row entry by a full evaluation of the 28 terms; otherwise only the blocks that read the moved abundance are
// evaluated again (kHcRowMask), the others keep the value of the unperturbed state, which g_hc_terms holds from the f(y) before this call.
RG_DEV void dev_T_border(const DevNet &N, const CellCtx &c, double con, bool full_rows = false) {
  const RG_GLOBAL DevHC &H = *(const RG_GLOBAL DevHC *)c.hc;
  const double Tc = g_T.y, r1 = g_T.savf, d2h = c.cell[6];
  wave_sync();
  if (c.lane < HC_NTERMS) g_hc_base[c.lane] = g_hc_terms[c.lane];
  wave_sync();
  for (int k = 0; k < 10; ++k) {
    const int j = H.idx10[k];
    double pc = 0.0;
    if (j >= 0) {
      const double yj = c.y[j];
      if (yj >= 0.0) {
        const double dy = yj * 1e-2 + d2h * 1e-6;
        wave_sync();
        if (c.lane == 0) c.y[j] = yj + dy;
        wave_sync();
        unsigned mask = kHcRowMask[k];
        if (k == 1 && H.i_gH < 0) mask |= 1u << HCB_H2FORM;
        if (full_rows) mask = ~0u;
        if (c.lane < HC_NTERMS) g_hc_terms[c.lane] = g_hc_base[c.lane];
        wave_sync();
        const double r2 = dev_heating_cooling(N, H, c.cell, c.hcrec, c.y, Tc, c.rates, g_T.rh2, c.lane, nullptr, mask);
        wave_sync();
        if (c.lane == 0) c.y[j] = yj;
        wave_sync();
        pc = (r2 - r1) / dy;
      }
    }
    g_T.Pc[k] = pc * con;
  }
  const double dT = Tc * 1e-2 + 1.0, T2 = Tc + dT;
  double td2;
  dev_f<true>(N, c, c.wx, &td2, &T2);
  const rsrc_t bB = mkbuf(c.Pb);
  for (int i0 = 0; i0 < c.n; i0 += 64) { const int i = i0 + c.lane; if (i < c.n) bstore_f64(bB, c.lane * 8, i0 * 8, (c.wx[i] - c.savf[i]) / dT * con); }
  g_T.Pd = (td2 - r1) / dT * con + 1.0;
  wave_sync();
}

// DPRJS for MITER = 1.  y (LDS) holds the predicted values.
template <bool ET = false>
RG_DEV void dev_prjs(const DevNet &N, const CellCtx &c, Lsodes &s) {
  const double hl0 = s.h * s.el0, con = -hl0;
  bool jok = true;
  if (s.nst == 0 || s.nst >= s.nslj + kMsbj) jok = false;
  if (s.icf == 1 && fabs(s.rc - 1.0) < kCcmxj) jok = false;
  if (s.icf == 2) jok = false;
  if (jok) {
    s.jcur = 0;
    const double rcon = con / s.con0, rcont = fabs(con) / s.conmin;
    if (rcont > kRbig && s.iplost == 1) jok = false;
    else {
      bool lost = false;
      const rsrc_t bP = mkbuf(c.Pv), bD = mkbuf(N.Pdiag);
      const int l8 = c.lane * 8;
      constexpr int B = 8; // blocks per trip, every load of a trip before any use: a trip costs one memory round trip, not B
      for (int c0 = 0; c0 < N.nnzJ; c0 += 64 * B) { // (the wave's slice of P ends at nnzJ: the next slot's slice follows directly)
        double pv[B]; uint8_t dv[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const int e0 = min(c0 + 64 * u, (N.nnzJ - 1) / 64 * 64); // (blocks past the end re-read the last one; P and Pdiag are padded by 64)
          pv[u] = bload_f64(bP, l8, e0 * 8); dv[u] = bload_u8(bD, c.lane, e0);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const int e0 = c0 + 64 * u;
          double pij = pv[u];
          const bool in = e0 + c.lane < N.nnzJ;
          const bool dg = in && dv[u] != 0; // P is stored in permuted-column order
          if (dg) { pij = pij - 1.0; if (fabs(pij) < kPsmall) lost = true; }
          pij = pij * rcon;
          if (dg) pij = pij + 1.0;
          if (in) bstore_f64(bP, l8, e0 * 8, pij);
        }
      }
      if constexpr (ET) { // the border of P: T column (rows = species), T row, T-T entry
        if (g_T.evolT) {
          const rsrc_t bB = mkbuf(c.Pb);
          vec_trips<RG_VEC_TRIP, double>(c.n, c.npad, [&](int i0) { return bload_f64(bB, l8, i0 * 8); }, [&](int i0, double v) { bstore_f64(bB, l8, i0 * 8, v * rcon); });
          for (int k = 0; k < 10; ++k) g_T.Pc[k] = g_T.Pc[k] * rcon;
          double pd = g_T.Pd - 1.0;
          if (fabs(pd) < kPsmall) lost = true;
          g_T.Pd = pd * rcon + 1.0;
        }
      }
      if (wave_any(lost)) { s.iplost = 1; s.conmin = fmin(fabs(s.con0), s.conmin); }
    }
  }
  if (!jok) {
    s.jcur = 1; s.nje++; s.nslj = s.nst; s.iplost = 0; s.conmin = fabs(con);
    dev_mark(c, 3000);
    {
      const long long t0 = dev_clock();
      if (c.nteam > 1) { // every wave of the team builds the entries of its segment of the term stream
        g_team.con = con; g_team.cmd = T_JAC; team_barrier();
        dev_build_P<true>(N, c.rates, g_wc.nsite, c.y, con, true, c.Pv, c.lane, 0);
        team_barrier();
      } else {
        dev_build_P<true>(N, c.rates, g_wc.nsite, c.y, con, true, c.Pv, c.lane);
      }
      if constexpr (ET) { if (g_T.evolT) dev_T_border(N, c, con); }
      cyc_add(CYC_JAC, dev_clock() - t0);
    }
    dev_mark(c, 3001);
  }
  s.nlu++; s.con0 = con; s.ierpj = 0;
  wave_sync();
  {
    const long long t0 = dev_clock();
    long long part[4] = {0, 0, 0, 0};
    if (c.nteam > 1) { g_team.fail = 0; g_team.cmd = T_LU; team_barrier(); } // the helpers enter dev_lu with their own work columns
#ifndef RG_LU_TICKS
#define RG_LU_TICKS 1
#endif
    if (!dev_lu(N, c.Pv, c.Lv, c.Uv, c.Dinv, c.wx, c.y, c.lane, RG_LU_TICKS ? part : nullptr, c.wx + ((c.n + 1) & ~1), 0, c.nteam, (volatile int *)&g_team.fail)) s.ierpj = 1;
    cyc_add(CYC_LU, dev_clock() - t0);
    for (int k = 0; k < 4; ++k) cyc_add(CYC_LU_PART + k, part[k]);
  }
  if constexpr (ET) {
    // the bordered system [A b; c^T d]: z = A^-1 b once per factorisation, Schur complement d - c.z.  f(y) is parked in the
    // accumulated-correction vector (dead until the corrector zeroes it) while its LDS vector serves the solve.
    if (g_T.evolT && s.ierpj == 0) {
      const rsrc_t bA = mkbuf(c.acor), bB = mkbuf(c.Pb), bZ = mkbuf(c.zb);
      const int l8 = c.lane * 8;
      wave_sync();
      for (int i0 = 0; i0 < c.n; i0 += 64) { const int i = i0 + c.lane; if (i < c.n) { bstore_f64(bA, l8, i0 * 8, c.savf[i]); c.savf[i] = bload_f64(bB, l8, i0 * 8); } }
      wave_sync();
      dev_solve(N, c.Lv, c.Uv, c.Dinv, c.savf, c.wx, c.lane);
      const RG_GLOBAL DevHC &H = *(const RG_GLOBAL DevHC *)c.hc;
      double dot = 0.0;
      for (int k = 0; k < 10; ++k) { const int j = H.idx10[k]; if (j >= 0) dot = dot + g_T.Pc[k] * c.savf[j]; }
      const double sc = g_T.Pd - dot;
      g_T.schur = sc;
      if (sc == 0.0) s.ierpj = 1;
      wave_sync();
      for (int i0 = 0; i0 < c.n; i0 += 64) { const int i = i0 + c.lane; if (i < c.n) { bstore_f64(bZ, l8, i0 * 8, c.savf[i]); c.savf[i] = bload_f64(bA, l8, i0 * 8); } }
      wave_sync();
    }
  }
  s.ierpj = uniform_i(wave_any(s.ierpj != 0) ? 1 : 0);
}

// x <- P^-1 x for the full NEQ system: the species block's factors, then the border (ET: the T entry of x is g_T.y)
template <bool ET>
RG_DEV void dev_solve_neq(const DevNet &N, const CellCtx &c) {
  dev_solve(N, c.Lv, c.Uv, c.Dinv, c.y, c.wx, c.lane);
  if constexpr (ET) {
    if (g_T.evolT) {
      const RG_GLOBAL DevHC &H = *(const RG_GLOBAL DevHC *)c.hc;
      double dot = 0.0;
      for (int k = 0; k < 10; ++k) { const int j = H.idx10[k]; if (j >= 0) dot = dot + g_T.Pc[k] * c.y[j]; }
      const double xT = (g_T.y - dot) / g_T.schur;
      const rsrc_t bZ = mkbuf(c.zb);
      wave_sync();
      for (int i0 = 0; i0 < c.n; i0 += 64) { const int i = i0 + c.lane; if (i < c.n) c.y[i] = c.y[i] - bload_f64(bZ, c.lane * 8, i0 * 8) * xT; }
      g_T.y = xT;
      wave_sync();
    }
  }
}

// One step.  Returns kflag (0, -1, -2).  Structure follows the restatement validated on the CPU side; every
// expression keeps the reference's operand order.
template <bool ET = false>
RG_DEV int dev_stode(const DevNet &N, const DevParams &P, const CellCtx &c, Lsodes &s) {
  const int n = c.n, lane = c.lane, l8 = c.lane * 8;
  const rsrc_t bY = mkbuf(c.yh), bA = mkbuf(c.acor), bE = mkbuf(c.ewt);
  const double told = s.tn;
  double delp = 0.0, del = 0.0, dsm = 0.0, rh = 0.0;
  int ncf = 0, m = 0, iredo = 0;
  s.kflag = 0; s.ierpj = 0; s.jcur = 0; s.icf = 0;

  if (s.jstart == 0) {
    s.lmax = kMaxord + 1; s.nq = 1; s.l = 2; s.ialth = 2; s.rmax = 10000.0; s.rc = 0.0;
    s.el0 = 1.0; s.crate = 0.7; s.hold = s.h; s.nslp = 0; s.ipup = 1;
    dev_set_order(P, s);
  } else if (s.jstart < 0) {
    if (s.jstart == -1) { s.ipup = 1; s.lmax = kMaxord + 1; if (s.ialth == 1) s.ialth = 2; }
    if (s.h != s.hold) { rh = s.h / s.hold; s.h = s.hold; iredo = 3; dev_rescale<ET>(c, s, rh, false); }
  }

  for (int guard = 0; guard < 64; ++guard) { // label 200; at most 10 + 10 + a few retries are possible
    if (fabs(s.rc - 1.0) > kCcmax) s.ipup = 1;
    if (s.nst >= s.nslp + kMsbp) s.ipup = 1;
    s.tn = s.tn + s.h;
    dev_mark(c, 2000 + guard);
    dev_pascal<ET>(c, s, true, true); // (the prediction is the corrector's first iterate: one pass over the array instead of two)
    dev_mark(c, 2100 + guard);

    bool converged = false;
    for (int pass = 0; pass < 4; ++pass) { // label 220: re-entered after a P refresh (at most twice: rescaled P, then fresh J)
      m = 0;
      if (pass > 0) { vec_trips<RG_VEC_TRIP, double>(n, c.npad, [&](int i0) { return bload_f64(bY, l8, i0 * 8); }, [&](int i0, double v) { if (i0 + lane < n) c.y[i0 + lane] = v; }); if constexpr (ET) g_T.y = g_T.yh[0]; }
      else wave_sync();
#ifdef RG_DEBUG_NEWTON
      const bool dumping = s.trace && s.ncalls == P.debug_dump_call && pass == 0 && guard == 0;
      double *dump = s.trace ? s.trace + (size_t)s.trace_cap * 16 : nullptr;
      auto dump_vec = [&](int slot, auto get) { if (dumping) for (int i = lane; i < n; i += 64) dump[64 + (size_t)slot * c.npad + i] = get(i); };
      dump_vec(0, [&](int i) { return c.y[i]; }); // the predicted y
#endif
      { const long long t0 = dev_clock(); dev_f<ET>(N, c, c.savf); cyc_add(CYC_RHS, dev_clock() - t0); } s.nfe++;
#ifdef RG_DEBUG_NEWTON
      dump_vec(1, [&](int i) { return c.savf[i]; }); // f(y_pred)
#endif
      dev_mark(c, 2200 + pass);
      if (s.ipup > 0) {
        dev_prjs<ET>(N, c, s);
        dev_mark(c, 2300 + pass);
        s.ipup = 0; s.rc = 1.0; s.nslp = s.nst; s.crate = 0.7;
        if (s.ierpj != 0) break;
      }
      for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bA, l8, i0 * 8, 0.0);
      if constexpr (ET) g_T.acor = 0.0;
      bool fail410 = false;
      for (;;) {
        vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, col_off(c, 1) + i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                         [&](int i0, D2 v) { const int i = i0 + lane; if (i < n) c.y[i] = s.h * c.savf[i] - (v.a + v.b); });
        if constexpr (ET) g_T.y = s.h * g_T.savf - (g_T.yh[1] + g_T.acor);
#ifdef RG_DEBUG_NEWTON
        if (dumping && m == 0) {
          wave_sync();
          dump_vec(2, [&](int i) { return c.y[i]; }); // the residual handed to the solve
          dump_vec(4, [&](int i) { return bload_f64(bY, (i & 63) * 8, col_off(c, 1) + (i & ~63) * 8); }); // yh(:, 2)
          dump_vec(5, [&](int i) { return bload_f64(bE, (i & 63) * 8, (i & ~63) * 8); });                 // inverse weights
          dump_vec(6, [&](int i) { return bload_f64(bY, (i & 63) * 8, (i & ~63) * 8); });                 // yh(:, 1)
          for (int e = lane; e < N.nnzJ; e += 64) dump[64 + 8 * (size_t)c.npad + e] = c.Pv[e];        // P as factored (permuted-column storage)
          if (lane == 0) { dump[0] = s.h; dump[1] = s.el0; dump[2] = s.nq; dump[3] = s.tn; dump[4] = s.con0; dump[5] = s.jcur; dump[6] = s.rc; dump[7] = s.nst; dump[8] = s.ncalls; dump[9] = 1.0; }
        }
#endif
        dev_mark(c, 2400 + m);
        { const long long t0 = dev_clock(); dev_solve_neq<ET>(N, c); cyc_add(CYC_SOLVE, dev_clock() - t0); }
        dev_mark(c, 2500 + m);
        del = dev_vnorm<ET>(c, [&](int i) { return c.y[i]; }, ET ? (double)g_T.y : 0.0);
#ifdef RG_DEBUG_NEWTON
        if (dumping && m == 0) dump_vec(3, [&](int i) { return c.y[i]; }); // the solve's answer
        if (s.trace_cap > 0) { // developer build: the component with the largest weighted Newton correction of this iteration
          double big = -1.0; int idx = 0;
          for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; const double sz = i < n ? fabs(c.y[i] * bload_f64(bE, l8, i0 * 8)) : -1.0; if (sz > big) { big = sz; idx = i; } }
          for (int mm = 32; mm >= 1; mm >>= 1) { const double ob = __shfl_xor(big, mm, 64); const int oi = __shfl_xor(idx, mm, 64); if (ob > big || (ob == big && oi < idx)) { big = ob; idx = oi; } }
          g_dbg.del[m < 3 ? m : 2] = del; g_dbg.idx[m < 3 ? m : 2] = uniform_i(idx); g_dbg.big[m < 3 ? m : 2] = uniform_d(big); g_dbg.m = m;
        }
#endif
        const double el1 = P.elco[s.nq][1];
        vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                         [&](int i0, D2 v) { const int i = i0 + lane; if (i < n) { const double a = v.b + c.y[i]; bstore_f64(bA, l8, i0 * 8, a); c.y[i] = v.a + el1 * a; } });
        if constexpr (ET) { const double a = g_T.acor + g_T.y; g_T.acor = a; g_T.y = g_T.yh[0] + el1 * a; }
        if (m != 0) s.crate = fmax(0.2 * s.crate, del / delp);
        const double dcon = del * fmin(1.0, 1.5 * s.crate) / (P.tesco[s.nq][2] * s.conit);
        if (dcon <= 1.0) { converged = true; break; }
        m++;
        if (m == kMaxcor) { fail410 = true; break; }
        if (m >= 2 && del > 2.0 * delp) { fail410 = true; break; }
        delp = del;
        { const long long t0 = dev_clock(); dev_f<ET>(N, c, c.savf); cyc_add(CYC_RHS, dev_clock() - t0); } s.nfe++;
      }
      if (converged) break;
      if (fail410 && s.jcur != 1) { s.icf = 1; s.ipup = 1; continue; }
      break;
    }

    if (!converged) { // label 430
      s.icf = 2; ncf++; s.rmax = 2.0; s.tn = told; s.nfail++;
      dev_pascal<ET>(c, s, false);
      if (fabs(s.h) <= 0.0 || ncf == kMxncf) { s.kflag = -2; break; }
      rh = 0.25; s.ipup = 1; iredo = 1;
      dev_rescale<ET>(c, s, rh, true);
      continue;
    }

    // label 450: local error test
    s.jcur = 0;
    if (m == 0) dsm = del / P.tesco[s.nq][2];
    else {
      double q = 0.0;
      vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bA, l8, i0 * 8), bload_f64(bE, l8, i0 * 8)}; },
                       [&](int i0, D2 v) { const double w = v.a * v.b; if (i0 + lane < n) q += w * w; });
      double qT = 0.0;
      if constexpr (ET) { const double w = g_T.acor * g_T.ewt; qT = w * w; }
      dsm = sqrt((ET ? wave_sum(q) + qT : wave_sum(q)) * g_wc.inv_neq) / P.tesco[s.nq][2];
    }

    bool consider = false;
    double rhup = 0.0;
    if (dsm > 1.0) { // label 500
      s.kflag = s.kflag - 1; s.tn = told; s.nfail++;
      dev_pascal<ET>(c, s, false);
      s.rmax = 2.0;
      if (fabs(s.h) <= 0.0) { s.kflag = -1; break; }
      if (s.kflag <= -3) { // label 640
        if (s.kflag == -10) { s.kflag = -1; break; }
        rh = 0.1;
        s.h = s.h * rh;
        vec_trips<RG_VEC_TRIP, double>(n, c.npad, [&](int i0) { return bload_f64(bY, l8, i0 * 8); }, [&](int i0, double v) { if (i0 + lane < n) c.y[i0 + lane] = v; });
        if constexpr (ET) g_T.y = g_T.yh[0];
        dev_f<ET>(N, c, c.savf); s.nfe++;
        for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; if (i < n) bstore_f64(bY, l8, col_off(c, 1) + i0 * 8, s.h * c.savf[i]); }
        if constexpr (ET) g_T.yh[1] = s.h * g_T.savf;
        s.ipup = 1; s.ialth = 5;
        if (s.nq != 1) { s.nq = 1; s.l = 2; dev_set_order(P, s); }
        continue;
      }
      iredo = 2; rhup = 0.0; consider = true;
    } else {
      s.kflag = 0; iredo = 0; s.nst++; s.hu = s.h; s.nqu = s.nq; s.qsum += s.nq;
      {
        const int l = s.l;
        double el[kMaxord + 1];
#pragma unroll
        for (int j = 0; j <= kMaxord; ++j) el[j] = (j < l) ? P.elco[s.nq][j + 1] : 0.0;
        vec_trips<2, DCols>(n, c.npad,
          [&](int i0) { // acor is read once per block
            DCols d;
#pragma unroll
            for (int j = 0; j <= kMaxord; ++j) d.v[j] = (j < l) ? bload_f64(bY, l8, col_off(c, j) + i0 * 8) : 0.0;
            d.v[kMaxord + 1] = bload_f64(bA, l8, i0 * 8);
            return d;
          },
          [&](int i0, DCols d) {
#pragma unroll
            for (int j = 0; j <= kMaxord; ++j)
              if (j < l) bstore_f64(bY, l8, col_off(c, j) + i0 * 8, d.v[j] + el[j] * d.v[kMaxord + 1]);
          });
        if constexpr (ET) for (int j = 0; j < l; ++j) g_T.yh[j] = g_T.yh[j] + P.elco[s.nq][j + 1] * g_T.acor;
      }
      s.ialth--;
      if (s.ialth == 0) { // label 520
        rhup = 0.0;
        if (s.l != s.lmax) {
          vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, col_off(c, s.lmax - 1) + i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                           [&](int i0, D2 v) { const int i = i0 + lane; if (i < n) c.savf[i] = v.b - v.a; });
          if constexpr (ET) g_T.savf = g_T.acor - g_T.yh[s.lmax - 1];
          const double dup = dev_vnorm<ET>(c, [&](int i) { return c.savf[i]; }, ET ? (double)g_T.savf : 0.0) / P.tesco[s.nq][3];
          const double exup = 1.0 / (s.l + 1);
          rhup = 1.0 / (1.4 * pow(dup, exup) + 0.0000014);
        }
        consider = true;
      } else {
        if (s.ialth <= 1 && s.l != s.lmax) {
          for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bY, l8, col_off(c, s.lmax - 1) + i0 * 8, bload_f64(bA, l8, i0 * 8));
          if constexpr (ET) g_T.yh[s.lmax - 1] = g_T.acor;
        }
        goto done700;
      }
    }

    if (consider) { // labels 540-630
      const double exsm = 1.0 / s.l;
      const double rhsm = 1.0 / (1.2 * pow(dsm, exsm) + 0.0000012);
      double rhdn = 0.0;
      if (s.nq != 1) {
        double q = 0.0;
        vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, col_off(c, s.l - 1) + i0 * 8), bload_f64(bE, l8, i0 * 8)}; },
                         [&](int i0, D2 v) { const double w = v.a * v.b; if (i0 + lane < n) q += w * w; });
        double qT = 0.0;
        if constexpr (ET) { const double w = g_T.yh[s.l - 1] * g_T.ewt; qT = w * w; }
        const double ddn = sqrt((ET ? wave_sum(q) + qT : wave_sum(q)) * g_wc.inv_neq) / P.tesco[s.nq][1];
        const double exdn = 1.0 / s.nq;
        rhdn = 1.0 / (1.3 * pow(ddn, exdn) + 0.0000013);
      }
      int newq, sel;
      if (rhsm >= rhup) sel = (rhsm < rhdn) ? 1 : 0;
      else sel = (rhup > rhdn) ? 2 : 1;
      if (sel == 2) { // label 590
        newq = s.l; rh = rhup;
        if (rh < 1.1) { s.ialth = 3; goto done700; }
        const double r = P.elco[s.nq][s.l] / s.l;
        for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bY, l8, col_off(c, newq) + i0 * 8, bload_f64(bA, l8, i0 * 8) * r);
        if constexpr (ET) g_T.yh[newq] = g_T.acor * r;
      } else {
        if (sel == 0) { newq = s.nq; rh = rhsm; }
        else { newq = s.nq - 1; rh = rhdn; if (s.kflag < 0 && rh > 1.0) rh = 1.0; }
        if (s.kflag == 0 && rh < 1.1) { s.ialth = 3; goto done700; }
        if (s.kflag <= -2) rh = fmin(rh, 0.2);
      }
      if (newq != s.nq) { s.nq = newq; s.l = s.nq + 1; dev_set_order(P, s); }
      dev_rescale<ET>(c, s, rh, true);
      if (iredo == 0) { s.rmax = 10.0; goto done700; }
      continue;
    }
  }
  if (s.kflag == 0) s.kflag = -2; // guard exhausted: treat as repeated convergence failure (cannot happen)
  s.hold = s.h; s.jstart = 1;
  return s.kflag;

done700: {
    const double r = 1.0 / P.tesco[s.nqu][2];
    vec_trips<RG_VEC_TRIP, double>(n, c.npad, [&](int i0) { return bload_f64(bA, l8, i0 * 8); }, [&](int i0, double v) { bstore_f64(bA, l8, i0 * 8, v * r); });
    if constexpr (ET) g_T.acor = g_T.acor * r;
  }
  s.hold = s.h; s.jstart = 1;
  return s.kflag;
}

template <bool ET = false>
RG_DEV void dev_intdy0(const CellCtx &c, const Lsodes &s, double t) { // y <- interpolant at t
  const double sf = (t - s.tn) / s.h;
  const rsrc_t bY = mkbuf(c.yh);
  const int l8 = c.lane * 8;
  const int nq = s.nq;
  vec_trips<2, DCols>(c.n, c.npad,
    [&](int i0) {
      DCols d;
#pragma unroll
      for (int j = 0; j <= kMaxord; ++j) d.v[j] = (j <= nq) ? bload_f64(bY, l8, col_off(c, j) + i0 * 8) : 0.0;
      return d;
    },
    [&](int i0, DCols v) {
      double d = 0.0; // Horner from column nq down; the columns above nq were loaded as zeros
#pragma unroll
      for (int j = kMaxord; j >= 0; --j) d = (j <= nq) ? ((j == nq) ? v.v[j] : v.v[j] + sf * d) : d;
      if (i0 + c.lane < c.n) c.y[i0 + c.lane] = d;
    });
  if constexpr (ET) {
    double d = g_T.yh[nq];
    for (int j = nq - 1; j >= 0; --j) d = g_T.yh[j] + sf * d;
    g_T.y = d;
  }
}

template <bool ET = false>
RG_DEV bool dev_ewset(const CellCtx &c) { // DEWSET + inversion; false if some weight is <= 0
  // (folding the driver's TOLSF sum, which reads the same two vectors next, into this pass costs two registers kernel-wide and
  // with them the third wave per SIMD: tried, not kept)
  bool bad = false;
  const rsrc_t bY = mkbuf(c.yh), bR = mkbuf(c.rtol), bA = mkbuf(c.atol), bE = mkbuf(c.ewt);
  const int l8 = c.lane * 8;
  vec_trips<RG_VEC_TRIP, D3>(c.n, c.npad, [&](int i0) { return D3{bload_f64(bR, l8, i0 * 8), bload_f64(bY, l8, i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                   [&](int i0, D3 v) {
                     const double e = v.a * fabs(v.b) + v.c;
                     if (i0 + c.lane < c.n && e <= 0.0) bad = true;
                     bstore_f64(bE, l8, i0 * 8, 1.0 / e);
                   });
  if constexpr (ET) {
    const double eT = g_T.rtol * fabs(g_T.yh[0]) + g_T.atol;
    if (eT <= 0.0) bad = true;
    g_T.ewt = 1.0 / eT;
  } else {
    const double Tg = g_wc.Tgas, eT = g_wc.rT * fabs(Tg) + g_wc.aT;
    if (eT <= 0.0) bad = true;
  }
  return !wave_any(bad);
}

template <bool ET = false>
RG_DEV void dev_finish(const CellCtx &c, const Lsodes &s, double &t) { // label 580 / 400
  const rsrc_t bY = mkbuf(c.yh);
  vec_trips<RG_VEC_TRIP, double>(c.n, c.npad, [&](int i0) { return bload_f64(bY, c.lane * 8, i0 * 8); }, [&](int i0, double v) { if (i0 + c.lane < c.n) c.y[i0 + c.lane] = v; });
  if constexpr (ET) g_T.y = g_T.yh[0];
  t = s.tn;
}

// One DLSODES call, ITASK = 4.  On entry y (LDS) is the user's Y; on exit it is Y at t.
// The output loop's own scalars live in LDS and are read and written through a volatile view: between two DLSODES calls
// they are touched a few times, while inside the call every register is wanted by the factorisation.
struct EvolState {
  double t, t_step, tout, t_good, rt_total, rt_last;
  long long nst_acc, nfe_acc, nje_acc, nlu_acc;
  int istate, nerr, nerr_c, qual, nrr, isav;
  long long errc; // error returns by ISTATE code, 16 bits each: -1, -4, -5, any other
};

// A cell between two integrator steps, set aside by the wave that was solving it (k_solve once the queue is empty and few waves
// are left) and taken up again by a team (k_solve_team_resume).  Everything else the integration needs -- Nordsieck array, P, L, U,
// weights, tolerances, the hand-off record -- stays where it is, in the workspace slot of the wave that parked it.
struct Parked {
  int cell, i;           // the cell; the last output interval completed (the DLSODES call for interval i + 1 is under way)
  int nst0, nfe0, nje0, nlu0; // the counters as that call found them
  long long elapsed;     // shader-clock cycles spent on the cell so far
  Lsodes s; EvolState e; WaveConst wc;
};
struct ParkIO {
  Parked *rec;           // this slot's record (null: the cell can neither be parked nor resumed)
  bool resume;           // take the cell up from *rec instead of starting it
  const int *counters;   // [0] cells handed out, [3] waves that have left the kernel
  int ncell, nwaves, park_max; // park_max > 0: park once every cell has been handed out and at most park_max waves are left
  double *ypark;         // where the iterate of a parked cell goes (behind *rec)
  int cell, slot; long long cyc0; // the cell, the workspace slot it lives in, the clock when it was started
  int *park_list, *park_count;    // the list of slots holding parked cells
};

constexpr int kIstateParked = 99, kIstateResume = 98; // dev_lsodes_call: left between two steps / re-entered there

RG_DEV bool dev_should_park(const ParkIO &io, int lane) {
  int go = 0;
  if (lane == 0) {
    const int handed = __hip_atomic_load(io.counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int left = __hip_atomic_load(io.counters + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    go = handed >= io.ncell && io.nwaves - left <= io.park_max;
  }
  return uniform_i(go) != 0;
}

template <bool ET = false>
RG_DEV void dev_lsodes_call(const DevNet &N, const DevParams &P, const CellCtx &c, Lsodes &s, double &t, double tout, int &istate, const ParkIO &io) {
  const double u = kUround;
  const int n = c.n, lane = c.lane, l8 = c.lane * 8;
  const rsrc_t bY = mkbuf(c.yh), bE = mkbuf(c.ewt);
  const bool reentry = istate == kIstateResume; // a parked cell: straight back into the step loop
  if (reentry) istate = 2;
  if (!reentry) {
  if (istate != 1 && s.init == 0) { istate = -3; return; }
  if (istate == 1) { s.init = 0; if (tout == t) return; }
  if (istate == 3) {
    // DIPREP/DPREP rerun on ISTATE=3 with an unchanged layout: the words DPREP zeroes (reference src/opkda1.f:1487-1494)
    // lie at the far end of a temporary work area, not at the saved P, so DPRJS may rescale what survives of it (below);
    // the "parameters changed" flag is raised.
    s.jstart = -1;
    // What the reference's rerun does do to the saved P: with the RWORK length the reference allocates, the zeroed words
    // overlap P's tail, NCOLM = min(nq + 1, MAXORD + 2) columns of YH deciding by how much (device_tables.hpp, Pkref).
    if (N.ref_clobber) {
      const int z = N.ref_zbase + (n + 1) * min(s.nq + 1, kMaxord + 2), thresh = N.ref_nnz1 - z;
      if (z > 0) {
        const rsrc_t bP = mkbuf(c.Pv), bK = mkbuf(N.Pkref);
        for (int e0 = 0; e0 < N.nnzJ; e0 += 64)
          if (e0 + lane < N.nnzJ && (int)bload_u16(bK, lane * 2, e0 * 2) >= thresh) bstore_f64(bP, l8, e0 * 8, 0.0);
        if constexpr (ET) { // the T row and the T column of the saved P lie in the same storage (the T column is its very end)
          if (g_T.evolT) {
            const RG_GLOBAL DevHC &H = *(const RG_GLOBAL DevHC *)c.hc;
            for (int k = 0; k < 10; ++k) if (H.kref_row[k] >= thresh) g_T.Pc[k] = 0.0;
            const rsrc_t bB = mkbuf(c.Pb);
            for (int i0 = 0; i0 < n; i0 += 64) if (i0 + lane < n && H.kref_col0 + i0 + lane >= thresh) bstore_f64(bB, l8, i0 * 8, 0.0);
            if (H.kref_col0 + n >= thresh) g_T.Pd = 0.0;
          }
        }
      }
    }
  }
  if (istate == 1) { // Block C
    s.h0 = 0.0;
    s.tn = t; s.nst = 0; s.h = 1.0;
    for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; if (i < n) bstore_f64(bY, l8, i0 * 8, c.y[i]); }
    if constexpr (ET) g_T.yh[0] = g_T.y;
    dev_f<ET>(N, c, c.savf); s.nfe = 1;
    for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; if (i < n) bstore_f64(bY, l8, col_off(c, 1) + i0 * 8, c.savf[i]); }
    if constexpr (ET) g_T.yh[1] = g_T.savf;
    if (!dev_ewset<ET>(c)) { istate = -3; return; }
    { const rsrc_t bP = mkbuf(c.Pv); for (int e0 = 0; e0 < N.nnzJ; e0 += 64) if (e0 + lane < N.nnzJ) bstore_f64(bP, l8, e0 * 8, 0.0); }
    if ((s.tcrit - tout) * (tout - t) < 0.0) { istate = -3; return; }
    s.jstart = 0; s.nslj = 0; s.nje = 0; s.nlu = 0; s.nslast = 0; s.hu = 0.0; s.nqu = 0;
    {
      const double tdist = fabs(tout - t), w0 = fmax(fabs(t), fabs(tout));
      if (tdist < 2.0 * u * w0) { istate = -3; return; }
      double tol = 0.0;
      const rsrc_t bR = mkbuf(c.rtol), bA = mkbuf(c.atol);
      for (int i0 = 0; i0 < n; i0 += 64) { const double r = bload_f64(bR, l8, i0 * 8); if (i0 + lane < n) tol = fmax(tol, r); }
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) tol = fmax(tol, __shfl_xor(tol, mm, 64));
      tol = uniform_d(fmax(tol, ET ? (double)g_T.rtol : (double)g_wc.rT));
      if (tol <= 0.0) {
        double tl = 0.0;
        for (int i0 = 0; i0 < n; i0 += 64) {
          const double a = bload_f64(bA, l8, i0 * 8);
          const int i = i0 + lane;
          if (i < n) { const double ay = fabs(c.y[i]); if (ay != 0.0) tl = fmax(tl, a / ay); }
        }
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) tl = fmax(tl, __shfl_xor(tl, mm, 64));
        tol = uniform_d(tl);
        if constexpr (ET) { const double Tg = g_T.y; if (Tg != 0.0) tol = fmax(tol, g_T.atol / fabs(Tg)); }
        else { const double Tg = g_wc.Tgas; if (Tg != 0.0) tol = fmax(tol, g_wc.aT / fabs(Tg)); }
      }
      tol = fmax(tol, 100.0 * u); tol = fmin(tol, 0.001);
      double sum = dev_vnorm<ET>(c, [&](int i) { return c.savf[i]; }, ET ? (double)g_T.savf : 0.0);
      sum = 1.0 / (tol * w0 * w0) + tol * sum * sum;
      s.h0 = 1.0 / sqrt(sum);
      s.h0 = fmin(s.h0, tdist);
      s.h0 = copysign(s.h0, tout - t);
    }
    const double rh = fabs(s.h0) * s.hmxi;
    if (rh > 1.0) s.h0 = s.h0 / rh;
    s.h = s.h0;
    for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bY, l8, col_off(c, 1) + i0 * 8, s.h0 * bload_f64(bY, l8, col_off(c, 1) + i0 * 8));
    if constexpr (ET) g_T.yh[1] = s.h0 * g_T.yh[1];
  } else { // Block D
    s.nslast = s.nst;
    if ((s.tn - s.tcrit) * s.h > 0.0) { istate = -3; return; }
    if ((s.tcrit - tout) * s.h < 0.0) { istate = -3; return; }
    if ((s.tn - tout) * s.h >= 0.0) { dev_intdy0<ET>(c, s, tout); t = tout; istate = 2; return; }
    const double hmx = fabs(s.tn) + fabs(s.h);
    if (fabs(s.tn - s.tcrit) <= 100.0 * u * hmx) { dev_finish<ET>(c, s, t); t = s.tcrit; istate = 2; return; }
    const double tnext = s.tn + s.h * (1.0 + 4.0 * u);
    if ((tnext - s.tcrit) * s.h > 0.0) {
      s.h = (s.tcrit - s.tn) * (1.0 - 4.0 * u);
      if (istate == 2) s.jstart = -2;
    }
  }
  }
  bool first = !reentry && (istate == 1);
  for (;;) { // Block E
    if (!first) {
      if (s.nst - s.nslast >= s.mxstep) { istate = -1; dev_finish<ET>(c, s, t); return; }
      if (!dev_ewset<ET>(c)) { istate = -6; dev_finish<ET>(c, s, t); return; }
    }
    first = false;
    {
      double q = 0.0;
      vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, i0 * 8), bload_f64(bE, l8, i0 * 8)}; },
                       [&](int i0, D2 w) { const double v = w.a * w.b; if (i0 + lane < n) q += v * v; });
      double vT;
      if constexpr (ET) vT = g_T.yh[0] * g_T.ewt;
      else { const double Tg = g_wc.Tgas; vT = Tg / (g_wc.rT * fabs(Tg) + g_wc.aT); }
      const double tolsf = u * sqrt((wave_sum(q) + vT * vT) * g_wc.inv_neq);
      if (tolsf > 1.0) {
        if (s.nst == 0) { istate = -3; return; }
        istate = -2; dev_finish<ET>(c, s, t); return;
      }
    }
    dev_mark(c, 100000 + s.nst);
    const int kflag = dev_stode<ET>(N, P, c, s);
    dev_mark(c, 200000 + s.nst);
    if (s.trace_cap > 0) {
      if (s.trace && s.ncalls < s.trace_cap && lane == 0) {
        double *tr = s.trace + (size_t)s.ncalls * 8;
        tr[0] = s.tn; tr[1] = s.h; tr[2] = s.hu; tr[3] = s.nq; tr[4] = kflag; tr[5] = s.nst; tr[6] = s.nfe; tr[7] = s.nje * 10000.0 + s.nlu;
#ifdef RG_DEBUG_NEWTON
        double *tx = s.trace + (size_t)s.trace_cap * 8 + (size_t)s.ncalls * 8; // second block: the last corrector pass of the call
        tx[0] = g_dbg.m; for (int k = 0; k < 3; ++k) { tx[1 + k] = g_dbg.del[k]; tx[4 + k] = g_dbg.idx[k] + 1e-3 * fmin(g_dbg.big[k], 900.0); }
        tx[7] = g_dbg.big[0];
#endif
      }
      if (++s.ncalls >= s.trace_cap) { istate = -3; dev_finish<ET>(c, s, t); return; }
    }
    if (kflag != 0) {
      istate = (kflag == -1) ? -4 : -5;
      // IMXER: first index of the largest |acor*ewt| (label 560)
      double big = -1.0; int idx = 0x7fffffff;
      {
        const rsrc_t bA = mkbuf(c.acor);
        for (int i0 = 0; i0 < n; i0 += 64) {
          const double sz = fabs(bload_f64(bA, l8, i0 * 8) * bload_f64(bE, l8, i0 * 8));
          const int i = i0 + lane;
          if (i < n && sz > big) { big = sz; idx = i; }
        }
      }
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) {
        const double ob = __shfl_xor(big, mm, 64); const int oi = __shfl_xor(idx, mm, 64);
        if (ob > big || (ob == big && oi < idx)) { big = ob; idx = oi; }
      }
      if constexpr (ET) { // the T entry is the last of the NEQ: it is the first index of the maximum only if strictly larger
        const double szT = fabs(g_T.acor * g_T.ewt);
        if (szT > big) { big = szT; idx = n; }
      }
      s.imxer = uniform_i(big > 0.0 ? idx : 0);
      dev_finish<ET>(c, s, t); return;
    }
    s.init = 1;
    if ((s.tn - tout) * s.h >= 0.0) { dev_intdy0<ET>(c, s, tout); t = tout; istate = 2; return; }
    const double hmx = fabs(s.tn) + fabs(s.h);
    if (fabs(s.tn - s.tcrit) <= 100.0 * u * hmx) { dev_finish<ET>(c, s, t); t = s.tcrit; istate = 2; return; }
    const double tnext = s.tn + s.h * (1.0 + 4.0 * u);
    if ((tnext - s.tcrit) * s.h > 0.0) { s.h = (s.tcrit - s.tn) * (1.0 - 4.0 * u); s.jstart = -2; }
    if (io.park_max > 0 && dev_should_park(io, lane)) { istate = kIstateParked; return; } // between two steps: hand the cell over to a team?
  }
}

// chem_evol_solve for one cell.  y (LDS) in: abundances at t0; out: abundances at the end of the run.
// ygood (HBM): the hand-off record, i.e. record(:, isav) of the caller's loop in calc_this_cell (reference
// src/disk.f90:1716-1733): the last record whose T and H2 entries are not NaN.
struct CellResult { double t_final, t_good; int quality, nerr, nrec_real, isav; long long nst, nfe, nje, nlu, qsum, errc; int nfail; bool parked;
                    double T_good; int evolT_end, freeze_rec; }; // ET: T of the hand-off record; whether T was still evolving at the end
// ET: the last records' T and times for the T-freeze test of chem_evol_solve (reference src/chemistry.f90:532-546), T of the hand-off record
struct THist { double T[8], t[8], T_good; };
static __shared__ volatile THist g_Th;


template <bool ET = false>
RG_DEV CellResult dev_evol_solve(const DevNet &N, const DevParams &P, const CellCtx &c, double t0, double t_max, double dt_first,
                                 int n_record, double *__restrict__ record, double *__restrict__ touts, double *__restrict__ ygood,
                                 double *trace, const ParkIO &io) {
  __shared__ Lsodes s_lds;
  __shared__ volatile EvolState e; // accessed by name: a reference would be a generic pointer (flat_load/flat_store)
  Lsodes &s = s_lds;
  const int lane = c.lane, n = c.n, neq = c.n + 1;
  if (!io.resume) {
    s = Lsodes{};
    s.trace = trace; s.trace_cap = P.debug_max_calls;
    s.tcrit = t_max; s.hmxi = (t_max > 0.0) ? 1.0 / t_max : 0.0; s.mxstep = P.mxstep > 0 ? P.mxstep : 500;
    e.istate = 1; e.nerr = 0; e.nerr_c = 0; e.qual = 0; e.nrr = 1; e.isav = 1; e.errc = 0;
    e.t = t0; e.t_step = dt_first; e.tout = t0 + dt_first; e.t_good = t0;
    e.nst_acc = 0; e.nfe_acc = 0; e.nje_acc = 0; e.nlu_acc = 0;
    // Deterministic stand-in for the reference's CPU-time guards (src/chemistry.f90:438, 480-491): the time the
    // reference would have spent is MODELLED from the call counters with per-call costs (racgpu_params; defaults =
    // the reference's measured costs on one core, SURVEY.md section 6: f 47 us, full Jacobian 10.4 ms, LU+solves
    // ~1.0 ms per factorisation).
    e.rt_total = 0.0; e.rt_last = 1e300;
    if (touts) { if (lane == 0) touts[0] = t0; }
    if (record) { for (int i = lane; i < n; i += 64) record[i] = c.y[i]; if (lane == 0) record[n] = ET ? (double)g_T.y : (double)g_wc.Tgas; }
    if constexpr (ET) { g_Th.T[1] = g_T.y; g_Th.t[1] = t0; g_Th.T_good = g_T.y; }
  } else {
    s = io.rec->s;
    s.trace = nullptr;
    const EvolState &pe = io.rec->e;
    e.t = pe.t; e.t_step = pe.t_step; e.tout = pe.tout; e.t_good = pe.t_good; e.rt_total = pe.rt_total; e.rt_last = pe.rt_last;
    e.nst_acc = pe.nst_acc; e.nfe_acc = pe.nfe_acc; e.nje_acc = pe.nje_acc; e.nlu_acc = pe.nlu_acc;
    e.istate = pe.istate; e.nerr = pe.nerr; e.nerr_c = pe.nerr_c; e.qual = pe.qual; e.nrr = pe.nrr; e.isav = pe.isav; e.errc = pe.errc;
  }
  bool parked = false;
  int park0[4] = {0, 0, 0, 0};
  for (int i = io.resume ? io.rec->i + 1 : 2; i <= n_record; ++i) {
    double tout = e.tout;
    if (tout >= s.tcrit) tout = s.tcrit;
    int istate = e.istate;
    const bool restart = (istate == 1);
    int nst0 = restart ? 0 : s.nst, nfe0 = restart ? 0 : s.nfe, nje0 = restart ? 0 : s.nje, nlu0 = restart ? 0 : s.nlu;
    if (io.resume && i == io.rec->i + 1) { // the call the cell was parked in
      nst0 = io.rec->nst0; nfe0 = io.rec->nfe0; nje0 = io.rec->nje0; nlu0 = io.rec->nlu0;
      istate = kIstateResume;
    }
    double t = e.t;
    dev_lsodes_call<ET>(N, P, c, s, t, tout, istate, io);
    if (istate == kIstateParked) { park0[0] = nst0; park0[1] = nfe0; park0[2] = nje0; park0[3] = nlu0; parked = true; break; }
    e.t = t;
    e.nst_acc = e.nst_acc + (s.nst - nst0); e.nfe_acc = e.nfe_acc + (s.nfe - nfe0); e.nje_acc = e.nje_acc + (s.nje - nje0); e.nlu_acc = e.nlu_acc + (s.nlu - nlu0);
    const double rt_this = P.rt_cost_f * (double)(s.nfe - nfe0) + P.rt_cost_jac * (double)(s.nje - nje0) + P.rt_cost_lu * (double)(s.nlu - nlu0);
    e.rt_total = e.rt_total + rt_this;
    wave_sync();
    if (touts) { if (lane == 0) touts[i - 1] = t; }
    if (record) { double *rec = record + (size_t)(i - 1) * neq; for (int k = lane; k < n; k += 64) rec[k] = c.y[k]; if (lane == 0) rec[n] = ET ? (double)g_T.y : (double)g_wc.Tgas; }
    e.nrr = i;
    if constexpr (ET) { g_Th.T[i & 7] = g_T.y; g_Th.t[i & 7] = t; }
    {
      // the record calc_this_cell would hand back if the run ended here (src/disk.f90:1716-1721)
      const double yh2 = N.i_H2 >= 0 ? c.y[N.i_H2] : 0.0;
      const double Tnow = ET ? (double)g_T.y : (double)g_wc.Tgas;
      if (!(isnan(yh2) || isnan(Tnow))) {
        if constexpr (ET) g_Th.T_good = Tnow;
        e.isav = i; e.t_good = t;
        const rsrc_t bG = mkbuf(ygood);
        for (int i0 = 0; i0 < n; i0 += 64) { const int k = i0 + lane; if (k < n) bstore_f64(bG, lane * 8, i0 * 8, c.y[k]); }
      }
    }
    if (P.max_steps_per_cell > 0 && e.nst_acc >= P.max_steps_per_cell) break; // deterministic "Premature finish"
    const double rt_max = P.max_runtime_allowed;
    if (rt_max > 0.0) { // src/chemistry.f90:482-491 on modelled time
      if (rt_this > fmax(10.0 * e.rt_last, 0.5 * rt_max) || e.rt_total > rt_max) break;
      if (rt_this > 5.0 / (double)n_record * rt_max) istate = 1;
      e.rt_last = rt_this;
    }
    if (t >= s.tcrit) break;
    if (istate < 0) {
      e.nerr = e.nerr + 1; e.nerr_c = e.nerr_c + 1;
      e.errc = e.errc + (1ll << (istate == -1 ? 0 : istate == -4 ? 16 : istate == -5 ? 32 : 48));
      if (istate == -4 || istate == -5) { // loosen the offending component's tolerances
        const int idx = s.imxer;
        if (ET && idx >= n) { g_T.rtol = fmin(g_T.rtol * 10.0, 1e-2); g_T.atol = fmin(g_T.atol * 100.0, 1.0); } // the T slot has caps of its own (src/chemistry.f90:334-337)
        else if (lane == 0) { c.rtol[idx] = fmin(c.rtol[idx] * 10.0, 1e-3); c.atol[idx] = fmin(c.atol[idx] * 100.0, 1e-20); }
        wave_sync();
      }
      if (istate == -3) { e.qual = e.qual + 256; break; }
      if (e.nerr_c < 3) istate = 3; else { istate = 1; e.nerr_c = 0; }
    }
    {
      bool bad = ET ? !(g_T.y > 0.0) : !(g_wc.Tgas > 0.0);
      if (N.i_gH2 >= 0 && fabs(c.y[N.i_gH2]) > 1.0) bad = true;
      if (N.i_gH2O >= 0 && fabs(c.y[N.i_gH2O]) > 1.0) bad = true;
      if (N.i_gH >= 0 && fabs(c.y[N.i_gH]) > 1.0) bad = true;
      if (N.i_H >= 0 && fabs(c.y[N.i_H]) > 2.0) bad = true;
      if (N.i_E >= 0 && fabs(c.y[N.i_E]) > 1.0) bad = true;
      if (wave_any(bad)) { e.qual = e.qual + 512; break; }
    }
    if constexpr (ET) {
      // the T-freeze test (src/chemistry.f90:532-546): once T has moved by less than t_scale_tol (T1 + T2) dt / t_max over the last
      // five records, the solver restarts with T held fixed -- and the rate coefficients as the last chem_cal_rates call left them
      if (g_T.maySwitchT && g_T.evolT && i > 10 && t > 1e-2 * s.tcrit) {
        double T1 = g_Th.T[i & 7], T2 = T1;
        for (int k = 1; k < 5; ++k) { const double v = g_Th.T[(i - k) & 7]; T1 = fmax(T1, v); T2 = fmin(T2, v); }
        const double dtt = g_Th.t[i & 7] - g_Th.t[(i - 5) & 7];
        if ((T1 - T2) < g_T.t_scale_tol * (T1 + T2) * dtt / s.tcrit) { istate = 1; g_T.evolT = 0; g_T.freeze_rec = i; }
      }
    }
    if (P.steps_reset > 0 && i % P.steps_reset == 0) istate = 1;
    e.istate = istate;
    const double t_step = e.t_step * P.ratio_tstep;
    e.t_step = t_step;
    e.tout = t + t_step;
  }
  if (parked) { // the iterate, the integrator's scalars, the per-cell constants and counters; the slot goes on the list
    for (int k = lane; k < n; k += 64) io.ypark[k] = c.y[k];
    if (lane == 0) {
      Parked *r = io.rec;
      r->cell = io.cell; r->i = e.nrr; r->elapsed = dev_clock() - io.cyc0; r->s = s;
      r->nst0 = park0[0]; r->nfe0 = park0[1]; r->nje0 = park0[2]; r->nlu0 = park0[3];
      EvolState &pe = r->e;
      pe.t = e.t; pe.t_step = e.t_step; pe.tout = e.tout; pe.t_good = e.t_good; pe.rt_total = e.rt_total; pe.rt_last = e.rt_last;
      pe.nst_acc = e.nst_acc; pe.nfe_acc = e.nfe_acc; pe.nje_acc = e.nje_acc; pe.nlu_acc = e.nlu_acc;
      pe.istate = e.istate; pe.nerr = e.nerr; pe.nerr_c = e.nerr_c; pe.qual = e.qual; pe.nrr = e.nrr; pe.isav = e.isav; pe.errc = e.errc;
      WaveConst &wc = r->wc;
      wc.nsite = g_wc.nsite; wc.Tgas = g_wc.Tgas; wc.rT = g_wc.rT; wc.aT = g_wc.aT; wc.inv_neq = g_wc.inv_neq;
      for (int k = 0; k < 8; ++k) wc.cyc[k] = g_wc.cyc[k];
      io.park_list[atomicAdd(io.park_count, 1)] = io.slot;
    }
    CellResult R{};
    R.parked = true;
    return R;
  }
  const int nrr = e.nrr;
  const double t = e.t;
  dev_mark(c, 300 + nrr);
  if (touts) { if (lane == 0) for (int i = nrr + 1; i <= n_record; ++i) touts[i - 1] = t; }
  dev_mark(c, 400);
  if (record) {
    for (int i = nrr + 1; i <= n_record; ++i) {
      double *rec = record + (size_t)(i - 1) * neq;
      for (int k = lane; k < n; k += 64) rec[k] = c.y[k];
      if (lane == 0) rec[n] = ET ? (double)g_T.y : (double)g_wc.Tgas;
    }
  }
  dev_mark(c, 401);
  int qual = e.qual;
  if (e.nerr > (int)(0.1f * (float)n_record)) qual += 1;
  if (t <= 0.5 * s.tcrit) qual += 2;
  CellResult R{};
  R.t_final = t; R.t_good = e.t_good; R.isav = e.isav; R.quality = qual; R.nerr = e.nerr; R.nrec_real = nrr;
  R.nst = e.nst_acc; R.nfe = e.nfe_acc; R.nje = e.nje_acc; R.nlu = e.nlu_acc; R.qsum = s.qsum; R.nfail = s.nfail; R.errc = e.errc;
  if constexpr (ET) { R.T_good = g_Th.T_good; R.evolT_end = g_T.evolT; R.freeze_rec = g_T.freeze_rec; }
  return R;
}

} // namespace racgpu
