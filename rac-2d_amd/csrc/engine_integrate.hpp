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
#pragma once
#include "engine_device.hpp"

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
// wave 0's requests to the other waves of its team (k_solve_team): written before a barrier, read after it
enum { T_EXIT = 0, T_LU = 1, T_JAC = 2 };
struct TeamCtl { int cmd, fail, cell, slot; double con; }; // cell: wave 0's current cell (its rate vector); slot: its workspace slot; con: -h*el0 of the Jacobian request
static __shared__ volatile TeamCtl g_team;
RG_DEV void cyc_add(int k, long long d) { g_wc.cyc[k] = g_wc.cyc[k] + d; }

RG_DEV long long dev_clock() { return (long long)__builtin_readcyclecounter(); }

RG_DEV void dev_mark(const CellCtx &c, int id) {
  if (c.marker && c.lane == 0) __hip_atomic_store(c.marker, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Lsodes { // the scalars ODEPACK keeps in COMMON /DLS001/ and /DLSS01/ plus the driver's SAVEd locals
  double conit, crate, hold, rmax, el0, h, hmxi, hu, rc, tn;
  double con0, conmin, tcrit, h0;
  int ialth, ipup, lmax, nslp, icf, ierpj, jcur, jstart, kflag, l, nq, nst, nfe, nje, nqu;
  int iplost, nslj, nlu, init, nslast, imxer, mxstep;
  long long qsum; int nfail;
  int ncalls; double *trace; int trace_cap; // developer aid
};

constexpr double kUround = 2.220446049250313e-16; // DUMACH()
constexpr double kCcmax = 0.3, kCcmxj = 0.2, kPsmall = 1000.0 * kUround, kRbig = 0.01 / kPsmall;
constexpr int kMaxord = 5, kMaxcor = 3, kMsbp = 20, kMxncf = 10, kMsbj = 50;

template <typename V>
RG_DEV double dev_vnorm(const CellCtx &c, V v) { // DVNORM over NEQ = nS+1 entries, the T entry being zero
  double s = 0.0;
  const rsrc_t bE = mkbuf(c.ewt);
  vec_trips<RG_VEC_TRIP, double>(c.n, c.npad, [&](int i0) { return bload_f64(bE, c.lane * 8, i0 * 8); },
                       [&](int i0, double e) { const int i = i0 + c.lane; if (i < c.n) { const double q = v(i) * e; s += q * q; } });
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
  }
  s.h = s.h * rh; s.rc = s.rc * rh; s.ialth = s.l;
}

// YH <- YH * Pascal (forward) or its inverse; DSTODE :865-874, :956-962
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
}

// DPRJS for MITER = 1.  y (LDS) holds the predicted values.
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
  s.ierpj = uniform_i(wave_any(s.ierpj != 0) ? 1 : 0);
}

// One step.  Returns kflag (0, -1, -2).  Structure follows the restatement validated on the CPU side; every
// expression keeps the reference's operand order.
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
    if (s.h != s.hold) { rh = s.h / s.hold; s.h = s.hold; iredo = 3; dev_rescale(c, s, rh, false); }
  }

  for (int guard = 0; guard < 64; ++guard) { // label 200; at most 10 + 10 + a few retries are possible
    if (fabs(s.rc - 1.0) > kCcmax) s.ipup = 1;
    if (s.nst >= s.nslp + kMsbp) s.ipup = 1;
    s.tn = s.tn + s.h;
    dev_mark(c, 2000 + guard);
    dev_pascal(c, s, true, true); // (the prediction is the corrector's first iterate: one pass over the array instead of two)
    dev_mark(c, 2100 + guard);

    bool converged = false;
    for (int pass = 0; pass < 4; ++pass) { // label 220: re-entered after a P refresh (at most twice: rescaled P, then fresh J)
      m = 0;
      if (pass > 0) vec_trips<RG_VEC_TRIP, double>(n, c.npad, [&](int i0) { return bload_f64(bY, l8, i0 * 8); }, [&](int i0, double v) { if (i0 + lane < n) c.y[i0 + lane] = v; });
      else wave_sync();
      { const long long t0 = dev_clock(); dev_rhs(N, c.rates, g_wc.nsite, gptr(N.r_C), c.y, c.savf, lane); cyc_add(CYC_RHS, dev_clock() - t0); } s.nfe++;
      dev_mark(c, 2200 + pass);
      if (s.ipup > 0) {
        dev_prjs(N, c, s);
        dev_mark(c, 2300 + pass);
        s.ipup = 0; s.rc = 1.0; s.nslp = s.nst; s.crate = 0.7;
        if (s.ierpj != 0) break;
      }
      for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bA, l8, i0 * 8, 0.0);
      bool fail410 = false;
      for (;;) {
        vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, col_off(c, 1) + i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                         [&](int i0, D2 v) { const int i = i0 + lane; if (i < n) c.y[i] = s.h * c.savf[i] - (v.a + v.b); });
        dev_mark(c, 2400 + m);
        { const long long t0 = dev_clock(); dev_solve(N, c.Lv, c.Uv, c.Dinv, c.y, c.wx, lane); cyc_add(CYC_SOLVE, dev_clock() - t0); }
        dev_mark(c, 2500 + m);
        del = dev_vnorm(c, [&](int i) { return c.y[i]; });
        const double el1 = P.elco[s.nq][1];
        vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                         [&](int i0, D2 v) { const int i = i0 + lane; if (i < n) { const double a = v.b + c.y[i]; bstore_f64(bA, l8, i0 * 8, a); c.y[i] = v.a + el1 * a; } });
        if (m != 0) s.crate = fmax(0.2 * s.crate, del / delp);
        const double dcon = del * fmin(1.0, 1.5 * s.crate) / (P.tesco[s.nq][2] * s.conit);
        if (dcon <= 1.0) { converged = true; break; }
        m++;
        if (m == kMaxcor) { fail410 = true; break; }
        if (m >= 2 && del > 2.0 * delp) { fail410 = true; break; }
        delp = del;
        { const long long t0 = dev_clock(); dev_rhs(N, c.rates, g_wc.nsite, gptr(N.r_C), c.y, c.savf, lane); cyc_add(CYC_RHS, dev_clock() - t0); } s.nfe++;
      }
      if (converged) break;
      if (fail410 && s.jcur != 1) { s.icf = 1; s.ipup = 1; continue; }
      break;
    }

    if (!converged) { // label 430
      s.icf = 2; ncf++; s.rmax = 2.0; s.tn = told; s.nfail++;
      dev_pascal(c, s, false);
      if (fabs(s.h) <= 0.0 || ncf == kMxncf) { s.kflag = -2; break; }
      rh = 0.25; s.ipup = 1; iredo = 1;
      dev_rescale(c, s, rh, true);
      continue;
    }

    // label 450: local error test
    s.jcur = 0;
    if (m == 0) dsm = del / P.tesco[s.nq][2];
    else {
      double q = 0.0;
      vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bA, l8, i0 * 8), bload_f64(bE, l8, i0 * 8)}; },
                       [&](int i0, D2 v) { const double w = v.a * v.b; if (i0 + lane < n) q += w * w; });
      dsm = sqrt(wave_sum(q) * g_wc.inv_neq) / P.tesco[s.nq][2];
    }

    bool consider = false;
    double rhup = 0.0;
    if (dsm > 1.0) { // label 500
      s.kflag = s.kflag - 1; s.tn = told; s.nfail++;
      dev_pascal(c, s, false);
      s.rmax = 2.0;
      if (fabs(s.h) <= 0.0) { s.kflag = -1; break; }
      if (s.kflag <= -3) { // label 640
        if (s.kflag == -10) { s.kflag = -1; break; }
        rh = 0.1;
        s.h = s.h * rh;
        vec_trips<RG_VEC_TRIP, double>(n, c.npad, [&](int i0) { return bload_f64(bY, l8, i0 * 8); }, [&](int i0, double v) { if (i0 + lane < n) c.y[i0 + lane] = v; });
        dev_rhs(N, c.rates, g_wc.nsite, gptr(N.r_C), c.y, c.savf, lane); s.nfe++;
        for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; if (i < n) bstore_f64(bY, l8, col_off(c, 1) + i0 * 8, s.h * c.savf[i]); }
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
      }
      s.ialth--;
      if (s.ialth == 0) { // label 520
        rhup = 0.0;
        if (s.l != s.lmax) {
          vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, col_off(c, s.lmax - 1) + i0 * 8), bload_f64(bA, l8, i0 * 8)}; },
                           [&](int i0, D2 v) { const int i = i0 + lane; if (i < n) c.savf[i] = v.b - v.a; });
          const double dup = dev_vnorm(c, [&](int i) { return c.savf[i]; }) / P.tesco[s.nq][3];
          const double exup = 1.0 / (s.l + 1);
          rhup = 1.0 / (1.4 * pow(dup, exup) + 0.0000014);
        }
        consider = true;
      } else {
        if (s.ialth <= 1 && s.l != s.lmax) {
          for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bY, l8, col_off(c, s.lmax - 1) + i0 * 8, bload_f64(bA, l8, i0 * 8));
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
        const double ddn = sqrt(wave_sum(q) * g_wc.inv_neq) / P.tesco[s.nq][1];
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
      } else {
        if (sel == 0) { newq = s.nq; rh = rhsm; }
        else { newq = s.nq - 1; rh = rhdn; if (s.kflag < 0 && rh > 1.0) rh = 1.0; }
        if (s.kflag == 0 && rh < 1.1) { s.ialth = 3; goto done700; }
        if (s.kflag <= -2) rh = fmin(rh, 0.2);
      }
      if (newq != s.nq) { s.nq = newq; s.l = s.nq + 1; dev_set_order(P, s); }
      dev_rescale(c, s, rh, true);
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
  }
  s.hold = s.h; s.jstart = 1;
  return s.kflag;
}

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
}

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
  const double Tg = g_wc.Tgas, eT = g_wc.rT * fabs(Tg) + g_wc.aT;
  if (eT <= 0.0) bad = true;
  return !wave_any(bad);
}

RG_DEV void dev_finish(const CellCtx &c, const Lsodes &s, double &t) { // label 580 / 400
  const rsrc_t bY = mkbuf(c.yh);
  vec_trips<RG_VEC_TRIP, double>(c.n, c.npad, [&](int i0) { return bload_f64(bY, c.lane * 8, i0 * 8); }, [&](int i0, double v) { if (i0 + c.lane < c.n) c.y[i0 + c.lane] = v; });
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
      }
    }
  }
  if (istate == 1) { // Block C
    s.h0 = 0.0;
    s.tn = t; s.nst = 0; s.h = 1.0;
    for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; if (i < n) bstore_f64(bY, l8, i0 * 8, c.y[i]); }
    dev_rhs(N, c.rates, g_wc.nsite, gptr(N.r_C), c.y, c.savf, lane); s.nfe = 1;
    for (int i0 = 0; i0 < n; i0 += 64) { const int i = i0 + lane; if (i < n) bstore_f64(bY, l8, col_off(c, 1) + i0 * 8, c.savf[i]); }
    if (!dev_ewset(c)) { istate = -3; return; }
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
      tol = uniform_d(fmax(tol, g_wc.rT));
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
        { const double Tg = g_wc.Tgas; if (Tg != 0.0) tol = fmax(tol, g_wc.aT / fabs(Tg)); }
      }
      tol = fmax(tol, 100.0 * u); tol = fmin(tol, 0.001);
      double sum = dev_vnorm(c, [&](int i) { return c.savf[i]; });
      sum = 1.0 / (tol * w0 * w0) + tol * sum * sum;
      s.h0 = 1.0 / sqrt(sum);
      s.h0 = fmin(s.h0, tdist);
      s.h0 = copysign(s.h0, tout - t);
    }
    const double rh = fabs(s.h0) * s.hmxi;
    if (rh > 1.0) s.h0 = s.h0 / rh;
    s.h = s.h0;
    for (int i0 = 0; i0 < n; i0 += 64) bstore_f64(bY, l8, col_off(c, 1) + i0 * 8, s.h0 * bload_f64(bY, l8, col_off(c, 1) + i0 * 8));
  } else { // Block D
    s.nslast = s.nst;
    if ((s.tn - s.tcrit) * s.h > 0.0) { istate = -3; return; }
    if ((s.tcrit - tout) * s.h < 0.0) { istate = -3; return; }
    if ((s.tn - tout) * s.h >= 0.0) { dev_intdy0(c, s, tout); t = tout; istate = 2; return; }
    const double hmx = fabs(s.tn) + fabs(s.h);
    if (fabs(s.tn - s.tcrit) <= 100.0 * u * hmx) { dev_finish(c, s, t); t = s.tcrit; istate = 2; return; }
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
      if (s.nst - s.nslast >= s.mxstep) { istate = -1; dev_finish(c, s, t); return; }
      if (!dev_ewset(c)) { istate = -6; dev_finish(c, s, t); return; }
    }
    first = false;
    {
      double q = 0.0;
      vec_trips<RG_VEC_TRIP, D2>(n, c.npad, [&](int i0) { return D2{bload_f64(bY, l8, i0 * 8), bload_f64(bE, l8, i0 * 8)}; },
                       [&](int i0, D2 w) { const double v = w.a * w.b; if (i0 + lane < n) q += v * v; });
      const double Tg = g_wc.Tgas, vT = Tg / (g_wc.rT * fabs(Tg) + g_wc.aT);
      const double tolsf = u * sqrt((wave_sum(q) + vT * vT) * g_wc.inv_neq);
      if (tolsf > 1.0) {
        if (s.nst == 0) { istate = -3; return; }
        istate = -2; dev_finish(c, s, t); return;
      }
    }
    dev_mark(c, 100000 + s.nst);
    const int kflag = dev_stode(N, P, c, s);
    dev_mark(c, 200000 + s.nst);
    if (s.trace_cap > 0) {
      if (s.trace && s.ncalls < s.trace_cap && lane == 0) {
        double *tr = s.trace + (size_t)s.ncalls * 8;
        tr[0] = s.tn; tr[1] = s.h; tr[2] = s.hu; tr[3] = s.nq; tr[4] = kflag; tr[5] = s.nst; tr[6] = s.nfe; tr[7] = s.nje * 10000.0 + s.nlu;
      }
      if (++s.ncalls >= s.trace_cap) { istate = -3; dev_finish(c, s, t); return; }
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
      s.imxer = uniform_i(big > 0.0 ? idx : 0);
      dev_finish(c, s, t); return;
    }
    s.init = 1;
    if ((s.tn - tout) * s.h >= 0.0) { dev_intdy0(c, s, tout); t = tout; istate = 2; return; }
    const double hmx = fabs(s.tn) + fabs(s.h);
    if (fabs(s.tn - s.tcrit) <= 100.0 * u * hmx) { dev_finish(c, s, t); t = s.tcrit; istate = 2; return; }
    const double tnext = s.tn + s.h * (1.0 + 4.0 * u);
    if ((tnext - s.tcrit) * s.h > 0.0) { s.h = (s.tcrit - s.tn) * (1.0 - 4.0 * u); s.jstart = -2; }
    if (io.park_max > 0 && dev_should_park(io, lane)) { istate = kIstateParked; return; } // between two steps: hand the cell over to a team?
  }
}

// chem_evol_solve for one cell.  y (LDS) in: abundances at t0; out: abundances at the end of the run.
// ygood (HBM): the hand-off record, i.e. record(:, isav) of the caller's loop in calc_this_cell (reference
// src/disk.f90:1716-1733): the last record whose T and H2 entries are not NaN.
struct CellResult { double t_final, t_good; int quality, nerr, nrec_real, isav; long long nst, nfe, nje, nlu, qsum, errc; int nfail; bool parked; };


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
    if (record) { for (int i = lane; i < n; i += 64) record[i] = c.y[i]; if (lane == 0) record[n] = g_wc.Tgas; }
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
    dev_lsodes_call(N, P, c, s, t, tout, istate, io);
    if (istate == kIstateParked) { park0[0] = nst0; park0[1] = nfe0; park0[2] = nje0; park0[3] = nlu0; parked = true; break; }
    e.t = t;
    e.nst_acc = e.nst_acc + (s.nst - nst0); e.nfe_acc = e.nfe_acc + (s.nfe - nfe0); e.nje_acc = e.nje_acc + (s.nje - nje0); e.nlu_acc = e.nlu_acc + (s.nlu - nlu0);
    const double rt_this = P.rt_cost_f * (double)(s.nfe - nfe0) + P.rt_cost_jac * (double)(s.nje - nje0) + P.rt_cost_lu * (double)(s.nlu - nlu0);
    e.rt_total = e.rt_total + rt_this;
    wave_sync();
    if (touts) { if (lane == 0) touts[i - 1] = t; }
    if (record) { double *rec = record + (size_t)(i - 1) * neq; for (int k = lane; k < n; k += 64) rec[k] = c.y[k]; if (lane == 0) rec[n] = g_wc.Tgas; }
    e.nrr = i;
    {
      // the record calc_this_cell would hand back if the run ended here (src/disk.f90:1716-1721)
      const double yh2 = N.i_H2 >= 0 ? c.y[N.i_H2] : 0.0;
      if (!(isnan(yh2) || isnan(g_wc.Tgas))) {
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
        if (lane == 0) { c.rtol[idx] = fmin(c.rtol[idx] * 10.0, 1e-3); c.atol[idx] = fmin(c.atol[idx] * 100.0, 1e-20); }
        wave_sync();
      }
      if (istate == -3) { e.qual = e.qual + 256; break; }
      if (e.nerr_c < 3) istate = 3; else { istate = 1; e.nerr_c = 0; }
    }
    {
      bool bad = !(g_wc.Tgas > 0.0);
      if (N.i_gH2 >= 0 && fabs(c.y[N.i_gH2]) > 1.0) bad = true;
      if (N.i_gH2O >= 0 && fabs(c.y[N.i_gH2O]) > 1.0) bad = true;
      if (N.i_gH >= 0 && fabs(c.y[N.i_gH]) > 1.0) bad = true;
      if (N.i_H >= 0 && fabs(c.y[N.i_H]) > 2.0) bad = true;
      if (N.i_E >= 0 && fabs(c.y[N.i_E]) > 1.0) bad = true;
      if (wave_any(bad)) { e.qual = e.qual + 512; break; }
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
      if (lane == 0) rec[n] = g_wc.Tgas;
    }
  }
  dev_mark(c, 401);
  int qual = e.qual;
  if (e.nerr > (int)(0.1f * (float)n_record)) qual += 1;
  if (t <= 0.5 * s.tcrit) qual += 2;
  CellResult R{};
  R.t_final = t; R.t_good = e.t_good; R.isav = e.isav; R.quality = qual; R.nerr = e.nerr; R.nrec_real = nrr;
  R.nst = e.nst_acc; R.nfe = e.nfe_acc; R.nje = e.nje_acc; R.nlu = e.nlu_acc; R.qsum = s.qsum; R.nfail = s.nfail; R.errc = e.errc;
  return R;
}

} // namespace racgpu
