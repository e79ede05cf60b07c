// engine_device.hpp -- gfx950 device code of the batched stiff chemistry integrator.
//
// Execution model: ONE CELL PER WAVEFRONT (64 lanes), workgroup = one wave, no barriers anywhere.  A wave
// owns 5 LDS vectors of nS doubles (iterate y, f(y), accumulated correction, inverse error weights, linear-
// solver work vector); everything bigger (rate vector, Nordsieck array, P, L, U) is the wave's private,
// contiguous slice of HBM, read and written with unit-stride 512-byte wave accesses.  Tables that are the
// same for every cell (reaction rows, Jacobian gather terms, LU pattern) are shared and stay in L2.
//
// Numerics follow the reference's ODEPACK path (DLSODES MF=21, ITOL=4, ITASK=4); citations per function.
#pragma once
#include <hip/hip_runtime.h>

#include "device_tables.hpp"

namespace racgpu {

#define RG_DEV __device__ __forceinline__

template <typename T>
RG_DEV const RG_GLOBAL T *gptr(const T *p) { return (const RG_GLOBAL T *)p; } // table pointer -> global address space

RG_DEV void wave_sync() {
  // one wave per workgroup: ordering between lanes needs only that the compiler keeps program order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Buffer addressing for the hot loops: the base of an array lives in a 4-SGPR resource, the wave-uniform part of
// the index in a scalar offset and the lane part in ONE VGPR shared by every array of the same element size.
// Compared with global_load on per-lane 64-bit pointers this removes the per-lane base pointers (two VGPRs per
// array, hoisted out of every loop and live through the whole kernel) and the 64-bit VALU address arithmetic.
// Offsets are bytes, unsigned, < 2 GiB from the base.
using rsrc_t = __amdgpu_buffer_rsrc_t;
RG_DEV rsrc_t mkbuf(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000); }
RG_DEV rsrc_t mkbuf_n(const void *p, int nbytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, nbytes, 0x00020000); } // loads past nbytes return 0
RG_DEV double bload_f64(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
RG_DEV unsigned long long bload_u64(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
RG_DEV uint32_t bload_u32(rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
RG_DEV uint16_t bload_u16(rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0); }
RG_DEV uint8_t bload_u8(rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0); }
RG_DEV void bstore_f64(rsrc_t r, int voff, int soff, double v) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
}
// The same for a wave's big private streams (rate vector, P, L, U), with their own cache-policy switch.  RG_STREAM_AUX: 0 = default
// policy, 2 = nt.  Measured on the configs[2] workload (round 2): nt on all of them is 22 % SLOWER (10.98 s against 9.00 s per
// pass): the left-looking LU reads back columns it has just written, and the solves follow a factorisation closely enough to
// find part of L and U still in L2.
#ifndef RG_STREAM_AUX
#define RG_STREAM_AUX 0
#endif
RG_DEV double sload_f64(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, RG_STREAM_AUX)); }
RG_DEV void sstore_f64(rsrc_t r, int voff, int soff, double v) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, RG_STREAM_AUX);
}
#ifndef RG_SOLVE_AUX
#define RG_SOLVE_AUX 2 // cache policy of the factor streams in the triangular solves (read once per solve)
#endif
RG_DEV double tload_f64(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, RG_SOLVE_AUX)); }
#ifndef RG_RATE_AUX
#define RG_RATE_AUX RG_STREAM_AUX // the rate vector as dev_rhs streams it
#endif
#ifndef RG_USTORE_AUX
#define RG_USTORE_AUX RG_STREAM_AUX // U as the factorisation writes it (only the solves read it back)
#endif
#ifndef RG_PLOAD_AUX
#define RG_PLOAD_AUX RG_STREAM_AUX // P as the factorisation reads it (once)
#endif
RG_DEV double kload_f64(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, RG_RATE_AUX)); }
RG_DEV double pload_f64(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, RG_PLOAD_AUX)); }
RG_DEV void ustore_f64(rsrc_t r, int voff, int soff, double v) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, RG_USTORE_AUX);
}
RG_DEV double uniform_d(double v) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
RG_DEV int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
RG_DEV double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return uniform_d(v);
}
RG_DEV bool wave_any(bool p) { return __ballot(p) != 0ULL; }

// ---------------------------------------------------------------------------------------------------------
// rate coefficients: chem_cal_rates, reference src/chemistry.f90:591-966 (helpers :1007-1086, :1542-1590)
// ---------------------------------------------------------------------------------------------------------
namespace cst {
constexpr double Pi = 3.1415926535897932384626433, kB_SI = 1.3806503e-23, kB = 1.3806503e-16,
                 eCharge_SI = 1.602176487e-19, Coulomb_SI = 8.9875517873681764e9, mP = 1.67262158e-24,
                 hbar = 1.054571628e-27, SecPerYear = 3600.0 * 24.0 * 365.0, HabingFlux = 6e7, UVext2Av = 2.6,
                 CR0 = 1.36e-17, CRattenN = 5.75e25, CosmicDesorpPre = 3.16e-19, CosmicDesorpT = 70.0;
}

RG_DEV double dev_sticking(double mass, double T) { // getStickingCoeff
  const double T0 = mass * (0.5 * (52.0 + 25.0)), r = T / T0;
  const double tmp = (1.0 + r) * (1.0 + r) * sqrt(1.0 + r);
  return 1.0 * (1.0 + 2.5 * r) / tmp;
}
RG_DEV double dev_mobility(const DevParams &P, double vib, double mass, double Ed, double Td) { // getMobility
  double m = vib * exp(fmax(-Ed * P.Diff2DesorRatio / Td,
                            -2.0 * 1e-8 / cst::hbar * sqrt(2.0 * mass * (cst::mP * cst::kB * P.Diff2DesorRatio) * Ed)));
  if (fabs(mass - 1.0) <= 1e-4 && P.use_special_gH_mobi) {
    const double E = P.special_gH_E_diff;
    m = vib * exp(fmax(-E / Td, -2.0 * 1e-8 / cst::hbar * sqrt(2.0 * mass * (cst::mP * cst::kB * E))));
  }
  return isnan(m) ? 0.0 : m;
}
RG_DEV double dev_branching(int itype, double A, double B, double C, double T0, double Td) { // getBranchingRatio
  if (itype < 63) return 1.0;
  double b = (C != 0.0) ? A * exp(fmax(-C / Td, -2.0 * B * 1e-8 / cst::hbar * sqrt(2.0 * T0 * cst::mP * cst::kB * C))) : A;
  return isnan(b) ? 0.0 : b;
}

// rh2 (may be null): where the cell's R_H2_form_rate_coeff goes -- the coefficient, still per second, of the last H2-formation
// reaction (itype 0, or 63 with gH first), which the reference stores as a side effect (:804, :891)
// Tover (may be null): the gas temperature to use instead of the record's (evolT: chem_ode_f sets chem_params%Tgas = y(NEQ) before
// it calls chem_cal_rates, reference src/disk.f90:4577-4580)
RG_DEV void dev_rates(const DevNet &N, const DevParams &P, const double *__restrict__ cell, double *__restrict__ rates, int lane,
                      double *rh2 = nullptr, const double *Tover = nullptr, double *tab = nullptr) {
  // tab (LDS, kRateTab doubles; null: every reaction evaluates its own pow and exp): see DevNet::r_ub
  const double Tgas = Tover ? *Tover : cell[0], Tdust = cell[1], n_gas = cell[2], D2H = cell[6], sites = cell[7];
  const double T300 = Tgas / 300.0;
  const double Tred = cst::kB_SI * Tgas / (cst::eCharge_SI * cst::eCharge_SI * cst::Coulomb_SI / (cell[3] * 1e-2));
  double JNegaPosi = 0.0, JChargeNeut = 0.0;
  if (Tred > 0.0) {
    JNegaPosi = (1.0 + 1.0 / Tred) * (1.0 + sqrt(2.0 / (2.0 + Tred)));
    JChargeNeut = 1.0 + sqrt(cst::Pi / 2.0 / Tred);
  }
  const double sig = cell[4];
  const double cr = cell[9] / cst::CR0 * exp(-cell[11] / cst::CRattenN), xr = cell[10] / cst::CR0;
  const bool tabbed = tab != nullptr && N.n_ub > 0;
  if (tabbed) {
    for (int q = lane; q < N.n_ub; q += 64) tab[q] = pow(T300, gptr(N.r_ub)[q]);
    for (int q = lane; q < N.n_uc; q += 64) tab[N.n_ub + q] = exp(-gptr(N.r_uc)[q] / Tgas);
    wave_sync();
  }
  // a reaction's row is fetched while the row before it is being evaluated: the loop is a chain of table loads otherwise (75 trips of ~2 us)
  struct Row { int it, fsel, a, b; double A, B, C, T0, T1; };
  auto load_row = [&](int r_) {
    const int r = min(r_, N.nR - 1);
    Row q;
    q.it = gptr(N.r_itype)[r]; q.fsel = gptr(N.r_fss)[r]; q.a = gptr(N.r_re0)[r]; q.b = gptr(N.r_re1)[r];
    q.A = gptr(N.r_A)[r]; q.B = gptr(N.r_B)[r]; q.C = gptr(N.r_C)[r]; q.T0 = gptr(N.r_T0)[r]; q.T1 = gptr(N.r_T1)[r];
    return q;
  };
  Row nxt = load_row(lane);
  for (int r = lane; r < N.nR; r += 64) {
    const Row row = nxt;
    nxt = load_row(r + 64);
    const int it = row.it;
    const double A = row.A, B = row.B, C = row.C, T0 = row.T0, T1 = row.T1;
    const int fsel = row.fsel;
    const double fI = fsel ? cell[19 + fsel - 1] : 1.0, fS = fsel ? cell[23 + fsel - 1] : 1.0;
    const int a = row.a, b = row.b;
    double k = 0.0;
    switch (it) {
      case 5: case 6: {
        // one pow and one exp for every branch of the reference's itype 5/6 selection (:681-717): which
        // temperature enters is decided first (clamped into [Tmin,Tmax] only for negative barriers, itype 5)
        double Te = Tgas;
        if (it == 5 && C < 0.0) { if (T0 > Tgas) Te = T0; else if (T1 < Tgas) Te = T1; }
        const bool zero = (it == 5) ? (Tgas <= 0.0) : (T0 > Tgas || T1 < Tgas);
        const double base = (Te == Tgas) ? T300 : Te / 300.0;
        if (tabbed && Te == Tgas) { const uint32_t ibc = gptr(N.r_ibc)[r]; k = zero ? 0.0 : A * tab[ibc & 0xffffu] * tab[N.n_ub + (ibc >> 16)]; }
        else k = zero ? 0.0 : A * pow(base, B) * exp(-C / Te);
      } break;
      case 1: k = A * (cr + xr); break;
      case 2: case 20: k = A * (C / (1.0 - cell[8]) * cr + xr); break;
      case 3:
        if (!(gptr(N.r_flags)[r] & 1)) k = A * (cell[14] * exp(-C * cell[12]) * fI + cell[15] * exp(-C * cell[13]) * fS);
        else k = A * (cell[14] * exp(-C * cell[12]) * fI + cell[16] * fS);
        break;
      case 21:
        if (Tgas <= 0.0) k = 0.0;
        else {
          const double m = gptr(N.s_mass)[gptr(N.r_id3)[r]] * cst::mP;
          k = sqrt(8.0 * cst::kB / cst::Pi * Tgas / m) * sig * ((gptr(N.r_flags)[r] & 4) ? JNegaPosi : JChargeNeut);
          if (sig <= 1e-30) k = 0.0;
        }
        break;
      case 13: k = cell[18] * A * fS; break;
      case 0:
        if (Tgas <= 0.0) k = 0.0;
        else {
          k = 0.5 * dev_sticking(gptr(N.s_mass)[a], Tgas) * sig * sqrt(8.0 / cst::Pi * cst::kB * Tgas / cst::mP) * D2H;
          if (sig <= 1e-30) k = 0.0;
        }
        break;
      case 61:
        if (Tgas <= 0.0) k = 0.0;
        else {
          const double m = gptr(N.s_mass)[a] * cst::mP;
          k = dev_sticking(gptr(N.s_mass)[a], Tgas) * A * sig * cell[5] * sqrt(8.0 / cst::Pi * cst::kB * Tgas / m);
          if (sig <= 1e-30) k = 0.0;
        }
        break;
      case 62:
        k = gptr(N.s_vib)[a] * (exp(-C / Tdust) + cst::CosmicDesorpPre * cr * exp(-C / cst::CosmicDesorpT));
        if (sig <= 1e-30) k = 0.0;
        k = k * (sites * D2H);
        break;
      case 63: {
        const double tmp = dev_mobility(P, gptr(N.s_vib)[a], gptr(N.s_mass)[a], gptr(N.s_Edes)[a], Tdust) / sites;
        k = tmp / D2H * dev_branching(it, A, B, C, T0, Tdust);
        if ((gptr(N.r_flags)[r] & 2) && P.h2_moeq && N.moeq_r61 >= 0 && N.moeq_r62 >= 0) {
          // chemsol_params%H2_form_use_moeq (src/chemistry.f90:876-881): gH + gH by the rate equation's steady state,
          // k_mig / (k_mig + desorb_coeff(gH)) * adsorb_coeff(H) / D2H, the two coefficients as the adsorption (:806-826) and desorption
          // (:827-846, before the surface-layer factor) cases of this same call leave them
          const int r61 = N.moeq_r61, r62 = N.moeq_r62, h = gptr(N.r_re0)[r61];
          double ads = 0.0;
          if (Tgas > 0.0) {
            const double m = gptr(N.s_mass)[h] * cst::mP;
            ads = dev_sticking(gptr(N.s_mass)[h], Tgas) * gptr(N.r_A)[r61] * sig * cell[5] * sqrt(8.0 / cst::Pi * cst::kB * Tgas / m);
            if (sig <= 1e-30) ads = 0.0;
          }
          const double C62 = gptr(N.r_C)[r62];
          double des = gptr(N.s_vib)[a] * (exp(-C62 / Tdust) + cst::CosmicDesorpPre * cr * exp(-C62 / cst::CosmicDesorpT));
          if (sig <= 1e-30) des = 0.0;
          k = tmp / (tmp + des) * ads / D2H;
        }
        if ((gptr(N.r_flags)[r] & 2) && sig <= 1e-30) k = 0.0;
      } break;
      case 64:
        k = (dev_mobility(P, gptr(N.s_vib)[a], gptr(N.s_mass)[a], gptr(N.s_Edes)[a], Tdust) +
             dev_mobility(P, gptr(N.s_vib)[b], gptr(N.s_mass)[b], gptr(N.s_Edes)[b], Tdust)) / (sites * D2H) * dev_branching(it, A, B, C, T0, Tdust);
        if (sig <= 1e-30) k = 0.0;
        break;
      case 75:
        k = (cell[17] * cst::HabingFlux + cell[14] * cst::HabingFlux * exp(-cst::UVext2Av * cell[12])) * sig * D2H * (A + B * Tdust);
        if (sig <= 1e-30) k = 0.0;
        break;
      default: k = 0.0;
    }
    if (rh2 && r == N.r_h2form) *rh2 = k;
    k = k * cst::SecPerYear;
    if (gptr(N.r_nreac)[r] == 2 && it < 60) k = k * n_gas;
    rates[r] = k;
  }
  wave_sync();
  // duplicate pruning (:948-964).  Which rate gets zeroed depends only on Tgas and the temperature ranges,
  // never on the rate values, so all reactions decide at once; the only writes are zeros.
  for (int r = lane; r < N.nR; r += 64) {
    const int q0 = gptr(N.dupli_ptr)[r], q1 = gptr(N.dupli_ptr)[r + 1];
    for (int q = q0; q < q1; ++q) {
      const int kk = gptr(N.dupli_list)[q];
      const double v0 = fabs(gptr(N.r_T0)[kk] - Tgas), v1 = fabs(gptr(N.r_T1)[kk] - Tgas), v2 = fabs(gptr(N.r_T0)[r] - Tgas), v3 = fabs(gptr(N.r_T1)[r] - Tgas);
      int im = 0; double vm = v0;
      if (v1 < vm) { vm = v1; im = 1; }
      if (v2 < vm) { vm = v2; im = 2; }
      if (v3 < vm) { vm = v3; im = 3; }
      if (im <= 1) { rates[r] = 0.0; break; }
      rates[kk] = 0.0;
    }
  }
  wave_sync();
}

RG_DEV double dev_powi(double a, int b) { // x**j with integer j as flang evaluates it (repeated squaring, compiler-rt __powidf2)
  double r = 1.0;
  for (;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
  return r;
}

// chem_set_solver_flags_alt(j), reference src/chemistry.f90:205-268 (species part; T slot returned separately)
RG_DEV void dev_tolerances(const DevNet &N, const DevParams &P, int j, double d2h, double *__restrict__ rtol, double *__restrict__ atol,
                           double &rT, double &aT, int lane) {
  double r, a;
  switch (j) {
    case 1: r = P.RTOL; a = P.ATOL; rT = 1e-3; aT = 1e-1; break;
    case 2: r = fmin(P.RTOL * 1e1, 1e-4); a = fmin(P.ATOL * 1e5, 1e-25); rT = 1e-2; aT = 1e-1; break;
    case 3: r = fmin(P.RTOL * 1e2, 1e-4); a = fmin(P.ATOL * 1e10, 1e-20); rT = 1e-3; aT = 1e0; break;
    case 4: r = fmin(P.RTOL * 1e2, 1e-4); a = fmin(P.ATOL * 1e10, 1e-18); rT = 1e-3; aT = 1e0; break;
    default: r = fmin(P.RTOL * dev_powi(2.0, j), 1e-3); a = fmin(P.ATOL * dev_powi(1e2, j), 1e-15); rT = 1e-2; aT = 1e0;
  }
  for (int i = lane; i < N.nS; i += 64) {
    const int c = gptr(N.s_tolclass)[i];
    double ri = r, ai = a;
    if (c == 1) { ri = fmax(P.RTOL, 1e-4); ai = fmax(P.ATOL, 1e-30); }
    else if (c == 2) { ri = 1e-4; ai = fmax(d2h * 1e-6, 1e-30); }
    else if (c == 3) { ri = fmax(P.RTOL, 1e-3); ai = fmax(P.ATOL, d2h * 1e-8); }
    rtol[i] = ri; atol[i] = ai;
  }
}

// ---------------------------------------------------------------------------------------------------------
// f(y): chem_ode_f, reference src/disk.f90:4569-4659 (fixed-T branch).  Reaction-major, ydot scattered with
// LDS f64 atomics (one wave owns the vector, so the result is deterministic).  ydot must be the SECOND of the wave's three LDS
// vectors: the rows' unused target slots point at the spare doubles behind the third.
// ---------------------------------------------------------------------------------------------------------
RG_DEV void dev_rhs(const DevNet &N, const double *__restrict__ rates, double nsite, const RG_GLOBAL double *__restrict__ r_C,
                    const double *y, double *ydot, int lane) {
  for (int i = lane; i < N.nS; i += 64) ydot[i] = 0.0;
  wave_sync();
  // 64 reactions per step, one per lane; the packed rows (w0: kind | n_reac<<8 | a<<16 | b<<32; w1, w2: the up to
  // seven species the flux is subtracted from / added to) and the rate vector are fetched two steps ahead (all
  // padded by 192 entries).  a and b are always valid species indices, so both abundances are read up front and
  // the three common flux forms are selected without branching; only the surface-layer forms (62, 75) branch.
  const rsrc_t bW0 = mkbuf(N.rhs_w0), bW1 = mkbuf(N.rhs_w1), bW2 = mkbuf(N.rhs_w2), bK = mkbuf(rates);
  const int l8 = lane * 8, spare0 = 2 * ((N.nS + 1) & ~1);
#ifndef RG_RHS_DEPTH
#define RG_RHS_DEPTH 3
#endif
  constexpr int D = RG_RHS_DEPTH;
  uint64_t w0[D], w1[D], w2[D];
  double kk[D];
#pragma unroll
  for (int s = 0; s < D - 1; ++s) { w0[s] = bload_u64(bW0, l8, s * 512); w1[s] = bload_u64(bW1, l8, s * 512); w2[s] = bload_u64(bW2, l8, s * 512); kk[s] = kload_f64(bK, l8, s * 512); }
  for (int r0 = 0; r0 < N.nR; r0 += 64 * D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int rb = r0 + 64 * s; // rows rb >= nR are padding (kind 0, no targets)
      {
        const int nx = (rb + 64 * (D - 1)) * 8, sl = (s + D - 1) % D;
        w0[sl] = bload_u64(bW0, l8, nx); w1[sl] = bload_u64(bW1, l8, nx); w2[sl] = bload_u64(bW2, l8, nx); kk[sl] = kload_f64(bK, l8, nx);
      }
      const uint64_t c0 = w0[s], c1 = w1[s], c2 = w2[s];
      const double k = kk[s];
      const int kind = (int)(c0 & 0xff), nre = (int)((c0 >> 8) & 0xff), a = (int)((c0 >> 16) & 0xffff), b = (int)((c0 >> 32) & 0xffff);
      const double ya = y[a], yb = y[b];
      const double kya = k * ya;
      double f2 = kya * yb; if (ya < 0.0 && yb < 0.0) f2 = -f2; // two-body (both-negative sign rule)
      double fq = kya * ya; if (ya < 0.0) fq = -fq;             // A + A
      double f = (kind == K_TWO_) ? f2 : ((kind == K_SQ_) ? fq : kya);
      if (kind == K_SURF_ || kind == K_SURF75_) { // surface layer forms (62, 75)
        double t1 = nsite; if (kind == K_SURF75_) t1 = t1 * r_C[rb + lane];
        if (t1 <= 0.0) f = k;
        else { const double t = ya / t1; f = (t <= 1e-4) ? k * t : k * (1.0 - exp(-t)); }
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int t = (int)(((q < 4 ? (c1 >> (16 * q)) : (c2 >> (16 * (q - 4))))) & 0xffff);
        // (an unused slot names the lane's spare double behind the third LDS vector: adding there untested is correct but the
        // extra LDS traffic costs 1.4 % of the pass; testing for it is cheaper than testing for 0xffff was)
        if (t < spare0) atomicAdd(&ydot[t], q < nre ? -f : f);
      }
    }
  }
  wave_sync();
  // Grain number.  Every reaction that touches Grain0 / Grain- / Grain+ moves ONE grain from one of them to another, so d/dt of their
  // sum is zero term by term; in floating point the three sums round separately and leave a residue of ~1e-16 of the gross charging
  // flux.  Grains are few (X ~ 1e-12) and turn over ~1e9 times a year, and where ions recombine on them X(H+) goes as 1 / X(grains):
  // the Newton matrix carries that residue into H+ multiplied by X(H+) / X(grains) ~ 1e5.  Measured on grid cell 39 (800 K, n = 8e9) at
  // RTOL 1e-8: the residue of this scatter (reactant slots before product slots within 64 reactions) was 1.3e-18, sixteen times the
  // reference's (which adds reaction by reaction, so that its three partial sums mirror each other), and put 16 tolerance units of
  // noise into every Newton correction of H+ -- 32 times the reference's Jacobian evaluations, end states of hot cells 6e-5 off in
  // H+.  Taking f(Grain0) from the other two makes the balance exact (the reference's is merely small); each of the three keeps an
  // error of its own rounding size, which the Newton matrix does not amplify.
  if (N.grain_conserved) {
    if (lane == 0) ydot[N.i_Grain0] = -((N.i_GrainM >= 0 ? ydot[N.i_GrainM] : 0.0) + (N.i_GrainP >= 0 ? ydot[N.i_GrainP] : 0.0));
    wave_sync();
  }
  // Charge (OFF by default: racgpu developer switch RACGPU_CHARGE_BALANCE=1).  Every reaction conserves it, so in exact arithmetic
  // sum_i q_i f_i = 0 and the electron's rate could be taken from the balance of all the others, as the grain balance above.  Measured
  // (profiles/r3_tuning.txt, item 5): it lets the hot configs[1] cells that used to stall finish like the reference's -- and makes other
  // cells stall instead: 10 flagged cells in 3 x 10^4 with it, 16 without, different ones.  Not a cure, so not the default.
  if (N.charge_conserved) {
    double q = 0.0;
    for (int i = lane; i < N.nS; i += 64) if (i != N.i_E) q += (double)gptr(N.s_charge)[i] * ydot[i];
    q = wave_sum(q);
    if (lane == 0) ydot[N.i_E] = q;
    wave_sync();
  }
}
// d(flux of one reaction)/d(y of the column species), k = the reaction's rate coefficient (chem_ode_jac, reference
// src/disk.f90:4746-4900: same forms and sign rules as the RHS)
RG_DEV double dev_dflux(uint64_t term, double k, const RG_GLOBAL double *__restrict__ r_C, double nsite, const double *y) {
  const int rxn = (int)(term & 0xffff), sa = (int)((term >> 16) & 0xffff), kind = (int)((term >> 32) & 0xff);
  const int flags = (int)((term >> 40) & 0xff), sb = (int)((term >> 48) & 0xffff);
  // sa and sb are valid species indices for every kind (network.cpp), so both abundances are read up front and the
  // common forms are selected without branching; only the surface-layer forms (62, 75) branch
  const double ya = y[sa], yb = y[sb];
  double v2 = (flags & 2) ? 2.0 * k * yb : ((flags & 4) ? k * yb : k * ya); // two-body: d/dy of k*ya*yb
  if (ya < 0.0 && yb < 0.0) v2 = -v2;
  double vq = 2.0 * k * ya; // A + A
  if (ya < 0.0) vq = -vq;
  double v = (kind == K_TWO_) ? v2 : ((kind == K_SQ_) ? vq : k);
  if (kind == K_SURF_ || kind == K_SURF75_) {
    double t2 = nsite; if (kind == K_SURF75_) t2 = t2 * r_C[rxn];
    if (t2 <= 0.0) v = 0.0;
    else { const double t1 = 1.0 / t2, t = ya * t1; v = (t <= 1e-4) ? k * t1 : k * t1 * exp(-t); }
  }
  return (flags & 1) ? -v : v;
}

// P = I - gamma*J (or J itself) on the pattern, entry by entry: every lane owns one entry per pass and adds up its
// terms in the reference's accumulation order (reaction order).  The terms come as one linear stream of rows of 64
// words (device_tables.hpp, jac_stream): term words are fetched 7 rows ahead, the rate coefficient each term names
// is gathered 3 rows ahead, so the only waits left in the loop are the LDS reads of y.
template <bool PERMUTED>
RG_DEV void dev_build_P(const DevNet &N, const double *__restrict__ rates, double nsite, const double *y, double con,
                        bool add_identity, double *__restrict__ Pv, int lane, const int seg = -1) {
  // PERMUTED: write each entry at its place in the permuted-column storage the LU reads with unit stride
  // seg >= 0: only the passes of that segment of the stream (one wave of a team; the entries are independent of each other)
  const int row0 = seg < 0 ? 0 : N.jac_seg_row[seg], row1 = seg < 0 ? N.jac_rows : N.jac_seg_row[seg + 1], pass0 = seg < 0 ? 0 : N.jac_seg_pass[seg];
  wave_sync();
  constexpr int U = kJacUnroll, DT = U - 1, DR = 3;
  const rsrc_t bT = mkbuf(N.jac_stream), bF = mkbuf(N.jac_rowflag), bS = mkbuf(N.jac_slot), bP = mkbuf(Pv);
  const rsrc_t bK = mkbuf_n(rates, N.nR * 8); // sized: the rate index of a null term (0xffff) is out of range and reads as 0
  const RG_GLOBAL double *r_C = gptr(N.r_C);
  const int l8 = lane * 8, l4 = lane * 4;
  uint64_t tw[U];
  double rk[U];
#pragma unroll
  for (int s = 0; s < DT; ++s) tw[s] = bload_u64(bT, l8, (row0 + s) * 512);
#pragma unroll
  for (int s = 0; s < DR; ++s) rk[s] = sload_f64(bK, (int)(tw[s] & 0xffff) * 8, 0);
  uint64_t slot = bload_u64(bS, l8, pass0 * 512), slot_nx = bload_u64(bS, l8, (pass0 + 1) * 512);
  int pass = pass0;
  double sum = 0.0;
  uint32_t fl = bload_u32(bF, l4, (row0 & ~63) * 4); // (a segment starts on a multiple of U rows, not of 64)
  for (int r0 = row0; r0 < row1; r0 += U) {
    if ((r0 & 63) == 0) fl = bload_u32(bF, l4, r0 * 4); // "last row of its pass" flags of the next 64 rows, one per lane
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int r = r0 + s;
      tw[(s + DT) % U] = bload_u64(bT, l8, (r + DT) * 512);
      rk[(s + DR) % U] = sload_f64(bK, (int)(tw[(s + DR) % U] & 0xffff) * 8, 0);
      const uint64_t term = tw[s];
      if (term != ~0ull) sum += dev_dflux(term, rk[s], r_C, nsite, y);
      if (__builtin_amdgcn_readlane((int)fl, r & 63)) { // the pass is complete: every lane stores its entry
        if ((slot >> 49) & 1ull) {
          double p = sum * con;
          if (add_identity && ((slot >> 48) & 1ull)) p = p + 1.0;
          const int dest = PERMUTED ? (int)((slot >> 24) & 0xffffff) : (int)(slot & 0xffffff);
          sstore_f64(bP, dest * 8, 0, p);
        }
        sum = 0.0; ++pass;
        slot = slot_nx; slot_nx = bload_u64(bS, l8, (pass + 1) * 512);
      }
    }
  }
  wave_sync();
}

RG_DEV void lds_order() {
  // Hot-loop variant of lds_sync: only pins the instruction order.  The LDS executes one wave's operations in
  // issue order, and every access it separates goes through a run-time index into the same array, so the
  // compiler has to keep them in program order anyway; no counter wait is inserted, which lets the next pivot's
  // LDS read issue right behind this pivot's write instead of waiting for the write to retire.
  __builtin_amdgcn_wave_barrier();
}

RG_DEV void lds_sync() {
  // LDS traffic of one wave is processed in issue order, so cross-lane hand-offs through LDS only need the
  // compiler to keep program order; no counter wait, so global prefetches stay in flight across it.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// Left-looking column LDU of the permuted P: P' = L * D * U, L unit lower, U unit upper, D^-1 stored.
// w is the wave's LDS work column.  Returns false on an exactly zero pivot (DPRJS IERPJ = 1).
// Everything a column needs first (its extents, its P entries, its pivot descriptors, the row lists of its U and
// L parts) is fetched while the previous column is being worked on.
#ifndef RG_BRANCHFREE
#define RG_BRANCHFREE 1
#endif
#ifndef RG_CLAMP_LOADS
#define RG_CLAMP_LOADS 1
#endif
//
// Team mode (nteam > 1 waves of one workgroup on one cell, k_solve_team): the columns k < ns go dependency level by level, the
// columns of a level dealt to the waves; the trailing columns then go round by round, nteam groups of G columns per round, one
// group per wave.  What a
// group needs from the columns k < ns and from the groups of earlier rounds (its LDS pivots and the dense pivots below the
// round's first column) is independent of the round's other groups and runs in parallel, each wave on a work column of its own;
// the rest of the round (the dense pivots that belong to the round's earlier groups, the group among itself, the stores) is
// taken in turns between barriers.  Every column sees its pivots in the same order as with one wave: the factors are the same
// to the last bit.  Every wave of the team executes the same number of barriers: 2 + levels + (nteam + 1) * rounds.
RG_DEV void team_barrier() { __syncthreads(); }

RG_DEV bool dev_lu(const DevNet &N, const double *__restrict__ Pv, double *__restrict__ Lv, double *__restrict__ Uv,
                   double *__restrict__ Dinv, double *w, double *dl, int lane, long long *cyc = nullptr, double *dmy = nullptr,
                   const int wv = 0, const int nteam = 1, volatile int *tfail = nullptr) {
  // dmy: 64 spare LDS doubles, one per lane (RG_BRANCHFREE)
  // w: LDS work column (one per wave);
  // dl: LDS copy of D^-1 (the U columns are scaled by it, and a gather from LDS beats one from HBM); shared by the team
  // wv, nteam, tfail: this wave's place in the team, the team's size, the team's "zero pivot" flag (LDS, cleared by the caller)
  bool ok = true;
  long long c_scatter = 0, c_rect = 0, c_dense = 0, c_fin = 0, tq = 0;
#define RG_TICK(acc) if (cyc) { const long long now_ = (long long)__builtin_readcyclecounter(); acc += now_ - tq; tq = now_; }
  const int n = N.nS, ns = N.ns;
  const rsrc_t bLrow = mkbuf(N.Lrow), bUrow = mkbuf(N.Urow), bProw = mkbuf(N.Prow), bUdesc = mkbuf(N.Udesc), bP = mkbuf(Pv), bL = mkbuf(Lv),
               bU = mkbuf(Uv);
  const int l2 = lane * 2, l8 = lane * 8; // lane part of every byte offset (u16 and 8-byte arrays)
#ifdef RG_EXP_LSHARED // timing experiment (k_newton only, results are garbage): every wave reads the L pieces of its LDS pivots from cell 0's slice
  const rsrc_t bLexp = mkbuf(Lv - (size_t)blockIdx.x * N.nzl);
#define RG_EXP_BL bLexp
#else
#define RG_EXP_BL bL
#endif
  const RG_GLOBAL int *cols = gptr(reinterpret_cast<const int *>(N.lucol)); // 16 ints per column (team mode: first the wave's own list)
  auto load_col = [&](int j) { const RG_GLOBAL int *c = cols + 16 * j; LuCol r; r.u0 = c[0]; r.u1 = c[1]; r.lc0 = c[2]; r.lc1 = c[3]; r.p0 = c[4]; r.p1 = c[5]; r.ur = c[6]; r.d0 = c[7]; r.d1 = c[8]; r.j = c[9]; r.o0 = c[10]; r.o1 = c[11]; return r; };
  for (int i = lane; i < n; i += 64) w[i] = 0.0; // the work column is kept all-zero between columns
  lds_sync();
  // the trailing columns are stored back to back in index order, so their starts have closed forms (scalar ALU)
  const int nzls = N.nzl_stream, nzus = N.nzu_stream, nt = n - ns;
  const int rowA = ns + lane, rowB = ns + 64 + lane, rA8 = rowA * 8, rB8 = rowB * 8;

  // one-column-ahead prefetch: extents (scalar), first 64 pivot descriptors, first 64 P entries, first 64 rows of
  // the U and L parts.  All tables and value slices are padded by 64 entries, so these loads are unconditional.
  LuCol nxc = load_col(0), nx2 = load_col(1), cur = nxc; // extents are fetched two work items ahead (scalar loads; the list is padded by two)
  unsigned long long nx_dq; double nx_pv; uint16_t nx_pr, nx_fu, nx_fl, cu_fu = 0, cu_fl = 0;
  auto prefetch_col = [&]() {
#if RG_CLAMP_LOADS
    const int op8 = min(l8, max(nxc.p1 - nxc.p0 - 1, 0) * 8); // (same clamp for the column's P entries)
    nx_dq = bload_u64(bUdesc, l8, nxc.d0 * 8); nx_pv = pload_f64(bP, op8, nxc.p0 * 8); nx_pr = bload_u16(bProw, op8 >> 2, nxc.p0 * 2);
#else
    nx_dq = bload_u64(bUdesc, l8, nxc.d0 * 8); nx_pv = pload_f64(bP, l8, nxc.p0 * 8); nx_pr = bload_u16(bProw, l2, nxc.p0 * 2);
#endif
    nx_fu = bload_u16(bUrow, l2, nxc.u0 * 2); nx_fl = bload_u16(bLrow, l2, nxc.lc0 * 2);
  };
  prefetch_col();
  auto prime = [&](int widx) { nxc = load_col(widx); nx2 = load_col(widx + 1); prefetch_col(); }; // (team mode: a wave's groups are not consecutive work items)
  if (cyc) tq = (long long)__builtin_readcyclecounter();

  // ---- column j, part 1: scatter P(:,j) into the work column and apply the pivots k < ns through LDS -------------
  // Pivot k of column j does w[i] -= L(i,k) * w[k] over the rows i of L(:,k).  The pivots come as packed
  // descriptors (device_tables.hpp, Udesc), 64 at a time, one per lane, sorted so that pivots which
  // do not feed each other (same level within the column, see build_symbolic) are adjacent: when a level opens,
  // ONE LDS read fetches w[k] for every lane's pivot, and the pivots of that level take their multiplier from the
  // owning lane (v_readlane) - no LDS round trip per pivot.  The updates are LDS atomics (ds_add_f64, no return),
  // so the wave never waits for them either; the LDS executes one wave's operations in issue order, which is what
  // makes the next level's read see them.  L columns are loaded kLuDepth-1 pivots ahead into register sets
  // with static indices (the loop is unrolled by the depth).
#if RG_LU_OPS
  // Entry-parallel form (device_tables.hpp, Uop): an operation is up to 64 (pivot k, entry of L(:,k)) pairs of ONE level of the column, one per
  // lane: the lane fetches its L value and row through the position its table word names, its multiplier w[k] from LDS, and adds -l * w[k] to
  // w[row] (LDS atomic, no return).  The LDS executes a wave's operations in issue order, so the reads of a level see every update of the levels
  // before it; lanes of one operation that hit the same row are added one after the other by the LDS (same order every time).  6 159 pivots of
  // on average 30 entries become 4 093 operations of on average 45: about 17 instructions per operation against 24 per pivot.  Table words
  // are two turns of the register sets ahead, L values and rows one turn.
  const rsrc_t bUop = mkbuf(N.Uop);
  const int l4 = lane * 4;
  auto rect_phase = [&](int widx, double *wv) { // widx: position in the work list; the column is cur.j afterwards
    cur = nxc; cu_fu = nx_fu; cu_fl = nx_fl;
    const double pv = nx_pv; const int pr = nx_pr;
    nxc = nx2; nx2 = load_col(widx + 2);
    prefetch_col();
    if (lane < cur.p1 - cur.p0) wv[pr] = pv;
    for (int q = cur.p0 + 64 + lane; q < cur.p1; q += 64) wv[bload_u16(bProw, q * 2, 0)] = pload_f64(bP, q * 8, 0); // rare: > 64 entries
    lds_sync();
    RG_TICK(c_scatter)
    constexpr int D = kLuOpsDepth;
    const int nop = cur.o1 - cur.o0;
    if (nop > 0) {
      uint32_t ta[D], tb[D]; // table words: ta of the operations whose values are in flight, tb of the turn after
      double l[D]; uint16_t i[D];
      const int o256 = cur.o0 * 256;
#pragma unroll
      for (int s = 0; s < D; ++s) ta[s] = bload_u32(bUop, l4, o256 + s * 256);
#pragma unroll
      for (int s = 0; s < D; ++s) tb[s] = bload_u32(bUop, l4, o256 + (D + s) * 256);
#pragma unroll
      for (int s = 0; s < D; ++s) { // (position << 16 | k: >> 13 and >> 15 leave the byte offsets of the f64 value and of the u16 row)
        l[s] = sload_f64(bL, (int)(ta[s] >> 13), 0); i[s] = bload_u16(bLrow, (int)(ta[s] >> 15), 0);
        __builtin_amdgcn_sched_barrier(0); // keep the issue order: data returns in order
      }
      double tvn = 0.0; bool have = false; // the multipliers of the next operation, read ahead when it continues the level
      for (int t = 0; t < nop; t += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
          if (t + s < nop) {
            double tv;
            if (have) tv = tvn;
            else { lds_order(); tv = wv[ta[s] & 0x3ffu]; }
            have = (__builtin_amdgcn_readfirstlane((int)ta[s]) & 0x1000) != 0; // lane 0's word, bit 12: the next operation belongs to the same level
            if (have) tvn = wv[ta[(s + 1) % D] & 0x3ffu]; // (before this operation's updates are queued)
            atomicAdd(&wv[i[s]], -(l[s] * tv));
          }
          ta[s] = tb[s];
          l[s] = sload_f64(bL, (int)(ta[s] >> 13), 0); i[s] = bload_u16(bLrow, (int)(ta[s] >> 15), 0);
          tb[s] = bload_u32(bUop, l4, o256 + (t + 2 * D + s) * 256);
        }
      }
    }
    RG_TICK(c_rect)
  };
#else
  auto rect_phase = [&](int widx, double *wv) { // widx: position in the work list; the column is cur.j afterwards
    cur = nxc; cu_fu = nx_fu; cu_fl = nx_fl;
    const unsigned long long dq0 = nx_dq;
    const double pv = nx_pv; const int pr = nx_pr;
    nxc = nx2; nx2 = load_col(widx + 2);
    prefetch_col();
    if (lane < cur.p1 - cur.p0) wv[pr] = pv;
    for (int q = cur.p0 + 64 + lane; q < cur.p1; q += 64) wv[bload_u16(bProw, q * 2, 0)] = pload_f64(bP, q * 8, 0); // rare: > 64 entries
    lds_sync();
    RG_TICK(c_scatter)
    constexpr int D = kLuDepth;
    constexpr int CH = 64 / D * D; // descriptors per fetch: one per lane, a multiple of D
    for (int base = cur.d0; base < cur.d1; base += CH) {
      const int nk = min(CH, cur.d1 - base), nmain = nk / D * D; // whole turns of the register sets, then up to D - 1 single pivots
      const unsigned long long dq = (base == cur.d0) ? dq0 : bload_u64(bUdesc, l8, base * 8);
      const int dlo = (int)(dq & 0xffffffffull), dhi = (int)(dq >> 32);
      const int myk = dlo & 0xffff;
      double tvv = 0.0;
      int ds[D]; // low word of the descriptor in each register set (wave-uniform)
      uint16_t i[D];
      double l[D];
      // lanes past the end of the L column re-read its LAST entry instead of whatever follows it: a 512-byte wave load of a column
      // of 8 entries then touches one cache line instead of eight (the pivot loop alone fetched 2.6x its useful bytes before).
      // The lane index of the prefetch may run past the chunk (v_readlane takes it modulo 64): whatever descriptor sits there
      // names a valid piece of L, and what is loaded for it is never applied.
#define RG_LU_ISSUE(S, tt)                                                                                        \
  {                                                                                                               \
    const int hi_ = __builtin_amdgcn_readlane(dhi, (tt));                                                          \
    ds[S] = __builtin_amdgcn_readlane(dlo, (tt));                                                                  \
    const int a2_ = hi_ & 0x1fffff, o8_ = min(l8, (int)((unsigned)hi_ >> 21));                                     \
    i[S] = bload_u16(bLrow, o8_ >> 2, a2_); l[S] = sload_f64(RG_EXP_BL, o8_, a2_ << 2);                            \
  }
#pragma unroll
      for (int s = 0; s < D - 1; ++s) { RG_LU_ISSUE(s, s) __builtin_amdgcn_sched_barrier(0); } // keep the issue order: data returns in order
#define RG_LU_PIVOT(s, t_)                                                                                        \
  {                                                                                                               \
    RG_LU_ISSUE(((s) + D - 1) % D, (t_) + (s) + D - 1)                                                             \
    const int tt = (t_) + (s);                                                                                     \
    if (ds[s] & (1 << 30)) { /* w[k] of this level's pivots is final now */                                        \
      lds_order();                                                                                                 \
      tvv = wv[myk];                                                                                               \
      asm volatile("" : "+v"(tvv)); /* take the LDS wait here, so that pivots which do not open a level never wait on LDS */ \
    }                                                                                                              \
    union { double d; int w[2]; } src, tv;                                                                         \
    src.d = tvv;                                                                                                   \
    tv.w[0] = __builtin_amdgcn_readlane(src.w[0], tt);                                                             \
    tv.w[1] = __builtin_amdgcn_readlane(src.w[1], tt);                                                             \
    /* no exec-masked branch: lanes past the end of the L column add (a finite product of neighbouring entries) to a slot of their own */ \
    atomicAdd(l8 < ((ds[s] >> 16) & 0x3ff) ? &wv[i[s]] : &dmy[lane], -(l[s] * tv.d));                              \
  }
      for (int t = 0; t < nmain; t += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) RG_LU_PIVOT(s, t)
      }
#pragma unroll
      for (int s = 0; s < D - 1; ++s) // the rest of the slice (padding it to the depth instead made every tenth pivot a null one)
        if (s < nk - nmain) RG_LU_PIVOT(s, nmain)
#undef RG_LU_PIVOT
#undef RG_LU_ISSUE
    }
    RG_TICK(c_rect)
  };
#endif

  // U part of the current column with rows < uend_rect: final after the LDS pivots; scaled and stored
  auto store_u = [&](double *wv, int uend) {
    int q = cur.u0 + lane;
    if (q < uend) { const int k = cu_fu; ustore_f64(bU, l8, cur.u0 * 8, wv[k] * dl[k]); wv[k] = 0.0; }
    for (q += 64; q < uend; q += 64) { const int k = bload_u16(bUrow, q * 2, 0); ustore_f64(bU, q * 8, 0, wv[k] * dl[k]); wv[k] = 0.0; }
  };

  // ---- column j, last part: pivot, scaled U and L columns to HBM, work column back to zero ------------------------
  auto finish = [&](int j, double *wv) {
    // all LDS reads of the column's first 64 U and L entries go out together (one wait), then zeroing and stores
    const double d = wv[j];
    const bool hasU = cur.u0 + lane < cur.u1, hasL = cur.lc0 + lane < cur.lc1;
    const int ku = hasU ? (int)cu_fu : j, il = hasL ? (int)cu_fl : j;
    const double uv = wv[ku], dk = dl[ku], lv = wv[il];
    lds_order();
    if (hasU) { ustore_f64(bU, l8, cur.u0 * 8, uv * dk); wv[ku] = 0.0; }
    for (int q = cur.u0 + 64 + lane; q < cur.u1; q += 64) { const int k = bload_u16(bUrow, q * 2, 0); ustore_f64(bU, q * 8, 0, wv[k] * dl[k]); wv[k] = 0.0; }
    if (d == 0.0) ok = false;
    const double dinv = 1.0 / d;
    if (lane == 0) { Dinv[j] = dinv; dl[j] = dinv; wv[j] = 0.0; }
    if (hasL) { sstore_f64(bL, l8, cur.lc0 * 8, lv * dinv); wv[il] = 0.0; }
    for (int q = cur.lc0 + 64 + lane; q < cur.lc1; q += 64) { const int i = bload_u16(bLrow, q * 2, 0); sstore_f64(bL, q * 8, 0, wv[i] * dinv); wv[i] = 0.0; }
    wave_sync(); // L, U, Dinv of this column are read back from HBM by later columns
    RG_TICK(c_fin)
  };

  auto rlane = [&](double v, int src) -> double { // v of lane src
    union { double d; int i[2]; } s, t;
    s.d = v;
    t.i[0] = __builtin_amdgcn_readlane(s.i[0], src);
    t.i[1] = __builtin_amdgcn_readlane(s.i[1], src);
    return t.d;
  };
  auto bcast = [&](double a, double b, int kk) -> double { // value held by the lane that owns tail row ns + kk
    union { double d; int i[2]; } s, t;
    s.d = (kk < 64) ? a : b;
    const int src = (kk < 64) ? kk : kk - 64;
    t.i[0] = __builtin_amdgcn_readlane(s.i[0], src);
    t.i[1] = __builtin_amdgcn_readlane(s.i[1], src);
    return t.d;
  };

  // ---- pivot-free columns, all at once: D^-1 = 1/P(j,j), L(:,j) = P(rows > j, j) * D^-1 (a team shares the blocks out) ----
  {
    const rsrc_t bLd = mkbuf(N.leaf_diag), bLe = mkbuf(N.leaf_ent);
    for (int q0 = 64 * wv; q0 < N.nleaf; q0 += 64 * nteam) {
      const unsigned long long e = bload_u64(bLd, l8, q0 * 8);
      if (q0 + lane < N.nleaf) {
        const int jj = (int)(e >> 32);
        const double d = pload_f64(bP, (int)(e & 0xfffff) * 8, 0);
        if (d == 0.0) ok = false;
        const double dinv = 1.0 / d;
        Dinv[jj] = dinv; dl[jj] = dinv;
      }
    }
    lds_sync();
    if (nteam > 1) { if (!ok) *tfail = 1; team_barrier(); } // every wave's D^-1 of these columns is in the shared LDS copy
    for (int q0 = 64 * wv; q0 < N.nleaf_ent; q0 += 64 * nteam) {
      const unsigned long long e = bload_u64(bLe, l8, q0 * 8);
      if (q0 + lane < N.nleaf_ent)
        sstore_f64(bL, (int)((e >> 20) & 0xfffff) * 8, 0, pload_f64(bP, (int)(e & 0xfffff) * 8, 0) * dl[(int)(e >> 40)]);
    }
    wave_sync();
    RG_TICK(c_fin)
  }

  if (nteam == 1) {
    for (int c = 0; c < N.nwork_sparse; ++c) { rect_phase(c, w); RG_TICK(c_dense) finish(cur.j, w); }
  } else {
    // the columns k < ns that have pivots, dependency level by level: the columns of a level do not feed each other and are dealt
    // to the waves (device_tables.hpp, lucol_team); a barrier per level
    team_barrier(); // the pivot-free columns are complete
    const RG_GLOBAL int *lp = gptr(N.team_lev_ptr) + wv * (N.team_nlev + 1);
    cols = gptr(reinterpret_cast<const int *>(N.lucol_team)) + 16 * N.team_base[wv];
    prime(0);
    for (int l = 0; l < N.team_nlev; ++l) {
      for (int c = lp[l]; c < lp[l + 1]; ++c) { rect_phase(c, w); RG_TICK(c_dense) finish(cur.j, w); }
      if (!ok) *tfail = 1; // (a network without trailing columns never reaches the rounds below, where the flag is otherwise raised)
      team_barrier(); // this level's columns and their D^-1 are in place for the whole team
    }
    cols = gptr(reinterpret_cast<const int *>(N.lucol));
  }

  // ---- dense trailing block, G columns at a time (G = 12: every L column of the block is read n/12 times, not n times) ----
  // After a column's LDS pivots (k < ns) its entries in rows < ns are final and go straight to U; its tail rows
  // (ns+lane, ns+64+lane) move to registers and stay there.  Pivot k >= ns reads its multiplier from the lane that
  // owns row k (v_readlane) and its L column (rows k+1..n-1, contiguous) with one load that serves all G columns.
  // Nothing of this phase touches LDS except the D^-1 copy; results are stored from registers.
#ifndef RG_DENSE_G
#define RG_DENSE_G 12
#endif
  constexpr int G = RG_DENSE_G;
  const int ngroups = (nt + G - 1) / G;
#ifndef RG_DENSE_AHEAD
#define RG_DENSE_AHEAD 2 // L columns of the block per register set (two sets: one being applied, one in flight)
#endif
  static_assert(RG_DENSE_G % (2 * RG_DENSE_AHEAD) == 0 && 64 % (2 * RG_DENSE_AHEAD) == 0, "a group's first column and row 64 of the block must start a pair of register sets");
  // Lane l holds rows ns + l ("A": wA) and ns + 64 + l ("B": wB) of every column of the group.  A pivot among the first 64 rows of
  // the block (FIRST) takes its multipliers from wA, touches the A rows below it (select on the L values) and all B rows; a later
  // one takes them from wB and touches B rows only: nothing is loaded or computed for A.  A range of pivots is cut at row 64 of
  // the block (a multiple of the register sets' stride), so each loop is of one kind.  Lanes whose row lies outside the matrix
  // carry garbage that is never stored or broadcast.
#define RG_DENSE_LOAD(LA, LB, kb_, kend_, FIRST)                                                                 \
  _Pragma("unroll") for (int u = 0; u < RG_DENSE_AHEAD; ++u) {                                                   \
    const int k = min((kb_) + u, max((kend_) - 1, ns)), kk = k - ns;                                              \
    const int cb = ((nzls + kk * (nt - 1) - kk * (kk - 1) / 2) - k - 1) * 8; /* byte offset of L(0, k): L(row, k) sits row*8 further */ \
    const double vb = sload_f64(bL, rB8, cb); /* unconditional: rows outside the column read neighbouring entries */ \
    if (FIRST) { const double va = sload_f64(bL, rA8, cb); LA[u] = (lane > kk) ? va : 0.0; LB[u] = vb; }          \
    else LB[u] = (lane > kk - 64) ? vb : 0.0;                                                                     \
  }
#define RG_DENSE_APPLY(LA, LB, kb_, kend_, FIRST)                                                                \
  _Pragma("unroll") for (int u = 0; u < RG_DENSE_AHEAD; ++u) {                                                   \
    const int k = (kb_) + u, kk = k - ns;                                                                         \
    if (k < (kend_)) {                                                                                            \
      _Pragma("unroll") for (int c = 0; c < G; ++c) {                                                            \
        if (FIRST) { const double t = rlane(wA[c], kk); wA[c] -= LA[u] * t; wB[c] -= LB[u] * t; }                 \
        else wB[c] -= LB[u] * rlane(wB[c], kk - 64);                                                              \
      }                                                                                                           \
    }                                                                                                             \
  }
#define RG_DENSE_RANGE1(kbeg_, kend_, FIRST)                                                                     \
  if ((kend_) > (kbeg_)) {                                                                                        \
    double la0[RG_DENSE_AHEAD], lb0[RG_DENSE_AHEAD], la1[RG_DENSE_AHEAD], lb1[RG_DENSE_AHEAD];                    \
    RG_DENSE_LOAD(la0, lb0, (kbeg_), (kend_), FIRST)                                                              \
    for (int kb = (kbeg_); kb < (kend_); kb += 2 * RG_DENSE_AHEAD) {                                              \
      RG_DENSE_LOAD(la1, lb1, kb + RG_DENSE_AHEAD, (kend_), FIRST)                                                \
      RG_DENSE_APPLY(la0, lb0, kb, (kend_), FIRST)                                                                \
      RG_DENSE_LOAD(la0, lb0, kb + 2 * RG_DENSE_AHEAD, (kend_), FIRST)                                            \
      RG_DENSE_APPLY(la1, lb1, kb + RG_DENSE_AHEAD, (kend_), FIRST)                                               \
    }                                                                                                             \
  }
  // pivots kbeg <= k < kend of the block on the group's columns (kbeg - ns is a multiple of G, hence of 2 * RG_DENSE_AHEAD)
#define RG_DENSE_RANGE(kbeg_, kend_)                                                                             \
  {                                                                                                               \
    const int kcut_ = ns + 64;                                                                                    \
    RG_DENSE_RANGE1((kbeg_), min((kend_), kcut_), true)                                                           \
    RG_DENSE_RANGE1(max((kbeg_), kcut_), (kend_), false)                                                          \
  }
  for (int g0 = 0; g0 < ngroups; g0 += nteam) { // one round: groups g0 .. g0 + nteam - 1, this wave's is g0 + wv
    const int j = ns + G * (g0 + wv), jround = ns + G * g0; // first column of the group / of the round
    const bool active = g0 + wv < ngroups;
    const int ng = active ? min(G, n - j) : 0;
    double wA[G], wB[G];
    if (nteam > 1) { if (active) prime(N.nwork_sparse + (j - ns)); }
#pragma unroll
    for (int c = 0; c < G; ++c) {
      wA[c] = 0.0; wB[c] = 0.0;
      if (c < ng) {
        const int jc = j + c;
        rect_phase(N.nwork_sparse + (jc - ns), w);
        store_u(w, cur.ur);
        if (rowA < n) { wA[c] = w[rowA]; w[rowA] = 0.0; }
        if (rowB < n) { wB[c] = w[rowB]; w[rowB] = 0.0; }
        lds_sync();
        RG_TICK(c_fin)
      }
    }
    if (nteam == 1) {
      RG_DENSE_RANGE(ns, j)
    } else {
      if (active) { RG_DENSE_RANGE(ns, jround) } // the columns of earlier rounds: complete since the last barrier
    }
    RG_TICK(c_dense)
    for (int turn = 0; turn < nteam; ++turn) {
      if (nteam > 1) team_barrier(); // the groups before this one in the round are complete
      if (turn != wv || !active) continue;
      if (nteam > 1) { RG_DENSE_RANGE(jround, j) }
      RG_TICK(c_dense)
      // the G columns of the group among themselves, and the stores (from registers)
#pragma unroll
      for (int c = 0; c < G; ++c) {
        if (c < ng) {
          const int jc = j + c, kk = jc - ns;
          const double d = bcast(wA[c], wB[c], kk);
          if (d == 0.0) ok = false;
          const double dinv = 1.0 / d;
          if (lane == 0) { Dinv[jc] = dinv; dl[jc] = dinv; }
          const double lA = (rowA > jc && rowA < n) ? wA[c] * dinv : 0.0, lB = (rowB > jc && rowB < n) ? wB[c] * dinv : 0.0;
#pragma unroll
          for (int c2 = c + 1; c2 < G; ++c2) {
            if (c2 < ng) {
              const double t = bcast(wA[c2], wB[c2], kk);
              wA[c2] -= lA * t; wB[c2] -= lB * t;
            }
          }
          lds_sync(); // dl[jc] is read below by the lanes of later columns
          const int ub = ((nzus + kk * (kk - 1) / 2) - ns) * 8;                    // byte offset of U(0, jc): rows ns <= row < jc are stored
          const int lb = ((nzls + kk * (nt - 1) - kk * (kk - 1) / 2) - jc - 1) * 8; // byte offset of L(0, jc): rows > jc are stored
          if (rowA < jc) ustore_f64(bU, rA8, ub, wA[c] * dl[rowA]);
          else if (rowA > jc && rowA < n) sstore_f64(bL, rA8, lb, lA);
          if (rowB < jc) ustore_f64(bU, rB8, ub, wB[c] * dl[rowB]);
          else if (rowB > jc && rowB < n) sstore_f64(bL, rB8, lb, lB);
        }
      }
      wave_sync(); // L, U, Dinv of these columns are read back from HBM by later columns
      RG_TICK(c_fin)
    }
    if (nteam > 1) { // the round is complete (the next one reads its columns; after the last round wave 0 goes on alone)
      if (!ok) *tfail = 1;
      team_barrier();
    }
  }
#undef RG_DENSE_RANGE
#undef RG_DENSE_RANGE1
#undef RG_DENSE_LOAD
#undef RG_DENSE_APPLY
  if (nteam > 1) ok = *tfail == 0;
  if (cyc) { cyc[0] += c_scatter; cyc[1] += c_rect; cyc[2] += c_dense; cyc[3] += c_fin; }
#undef RG_TICK
  return ok;
}

// One triangular sweep over the streamed part: entries come in dependency-level order, 64 per chunk, one level per
// chunk (the storage is level-aligned); every entry does x[row] -= v * x[col].  Within a level the columns are
// independent; entries of different columns may hit the same row, hence the LDS atomic (no return: the wave never
// waits for it).  Schedule words and values are kSweepDepth-1 chunks ahead in registers; the x[col] of the next
// chunk are read from LDS before this chunk's updates are issued whenever that chunk continues the level, so a
// level costs one LDS round trip, not one per chunk.
RG_DEV void dev_tri_sweep(const uint32_t *__restrict__ rc, const double *__restrict__ val, int nchunk, double *w, int lane) {
  if (nchunk <= 0) return;
  constexpr int D = kSweepDepth;
  uint32_t r[D];
  double v[D];
  const rsrc_t brc = mkbuf(rc), bval = mkbuf(val);
  const int l4 = lane * 4, l8 = lane * 8;
#pragma unroll
  for (int s = 0; s < D - 1; ++s) { r[s] = bload_u32(brc, l4, s * 256); v[s] = tload_f64(bval, l8, s * 512); }
  double x = 0.0;
  bool have = false; // x already holds this chunk's x[col] (read while the previous chunk of the same level was applied)
  for (int c = 0; c < nchunk; c += D) { // nchunk is a multiple of D (null chunks at the end)
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int sl = (s + D - 1) % D; // the set applied one sub-step ago is free again
      r[sl] = bload_u32(brc, l4, (c + s + D - 1) * 256); v[sl] = tload_f64(bval, l8, (c + s + D - 1) * 512);
      const uint32_t wd = r[s], wn = r[(s + 1) % D];
      const int row = (int)(wd & 1023u), col = (int)((wd >> 10) & 1023u);
      const bool cont = (__builtin_amdgcn_readfirstlane((int)wd) >> 20) & 1;
      if (!have) { lds_order(); x = w[col]; }
      double xn = 0.0;
      if (cont) xn = w[(wn >> 10) & 1023u];
      if (row != col) atomicAdd(&w[row], -(v[s] * x));
      have = cont; x = xn;
    }
  }
}

// x <- P^-1 x with the factors above; x (species order) and w are LDS vectors (DSOLSS / CDRV path 4).
// The streamed parts of L and U go through dev_tri_sweep; the dense trailing block (rows/columns >= ns) is solved
// in registers, two rows per lane: column k takes its x_k from the owning lane (v_readlane) and its L (or U) column
// with one contiguous load per half, RG_DS_DEPTH columns in flight.  Loads are unconditional (row index clamped into
// the column), the unused lanes are switched off by a select, so the loops are free of branches.
#ifndef RG_DS_DEPTH
#define RG_DS_DEPTH 8
#endif
RG_DEV void dev_solve(const DevNet &N, const double *__restrict__ Lv, const double *__restrict__ Uv, const double *__restrict__ Dinv,
                      double *x, double *w, int lane) {
  const int n = N.nS, ns = N.ns, nt = n - ns;
  lds_sync();
  for (int i = lane; i < n; i += 64) w[i] = x[gptr(N.perm)[i]];
  lds_sync();
  dev_tri_sweep(N.Lrc, Lv, N.nchunkL, w, lane); // columns k < ns
  lds_sync();
  const int rowA = ns + lane, rowB = ns + 64 + lane, rA8 = rowA * 8, rB8 = rowB * 8;
  const bool hasA = rowA < n, hasB = rowB < n;
  const rsrc_t bL = mkbuf(Lv), bU = mkbuf(Uv);
  double xA = hasA ? w[rowA] : 0.0, xB = hasB ? w[rowB] : 0.0;
  // The trailing columns are stored back to back in index order: the start of column k+1 follows from that of column k by one
  // scalar add.  Lane l holds rows ns + l ("A") and ns + 64 + l ("B").  A column of L among the first 64 of the block touches
  // the A rows below its diagonal (select) and every B row (no select); a later one touches B rows only: nothing is loaded or
  // computed for A.  U mirrors that.  Lanes whose row lies outside the matrix carry garbage that is never read (the broadcasts
  // take rows < n only).  The loops run to a multiple of the depth: columns past the end are switched off by the same selects
  // (the clamped multiplier is finite, and the values loaded for a column past the end meet only switched-off or unused lanes);
  // their loads stay inside the slot's storage or the spare behind it.
  const int nzls = N.nzl_stream, nzus = N.nzu_stream;
  constexpr int D = RG_DS_DEPTH;
  static_assert(64 % D == 0, "the first 64 columns of the block are taken in whole groups of D");
  auto rl = [&](double v, int src) -> double { // v of lane src
    union { double d; int i[2]; } s_, t_;
    s_.d = v;
    t_.i[0] = __builtin_amdgcn_readlane(s_.i[0], src);
    t_.i[1] = __builtin_amdgcn_readlane(s_.i[1], src);
    return t_.d;
  };
  if (nt > 1) {
    // forward: columns k = ns .. n-2 of L, rows k+1 .. n-1; byte offset of L(0, k): ((nzls + kk (nt-1) - kk (kk-1)/2) - k - 1) * 8
    double la[D], lb[D];
    int cb = (nzls - ns - 1) * 8, str = (nt - 2) * 8; // of the next column to load, and the step to the one after it
#define RG_DS_LOAD(S, WITH_A)                                                                                      \
  {                                                                                                               \
    if (WITH_A) la[S] = tload_f64(bL, rA8, cb);                                                                    \
    lb[S] = tload_f64(bL, rB8, cb);                                                                                \
    cb += str; str -= 8;                                                                                           \
  }
#pragma unroll
    for (int s = 0; s < D; ++s) RG_DS_LOAD(s, true)
    const int kend = nt - 1, kmid = min(64, kend);
    int k0 = 0;
    for (; k0 < kmid; k0 += D) { // columns among the first 64: A below the diagonal, all of B
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int kk = k0 + s;
        const double t = rl(xA, min(kk, nt - 1));
        xA -= ((lane > kk) ? la[s] : 0.0) * t;
        xB -= lb[s] * t;
        RG_DS_LOAD(s, true)
      }
    }
    for (; k0 < kend; k0 += D) { // later columns: B below the diagonal, nothing of A
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int kk = k0 + s;
        const double t = rl(xB, min(kk, nt - 1) - 64);
        xB -= ((lane > kk - 64) ? lb[s] : 0.0) * t;
        RG_DS_LOAD(s, false)
      }
    }
#undef RG_DS_LOAD
  }
  // D^-1: four blocks' loads go out before any is used (they were one memory round trip each)
  {
    const rsrc_t bD = mkbuf(Dinv);
    const int l8d = lane * 8;
    const double dA = bload_f64(bD, min(rowA, N.npad - 1) * 8, 0), dB = bload_f64(bD, min(rowB, N.npad - 1) * 8, 0);
    for (int c0 = 0; c0 < ns; c0 += 256) {
      double dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) dv[u] = bload_f64(bD, l8d, min(c0 + 64 * u, N.npad - 64) * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = c0 + 64 * u + lane; if (i < ns) w[i] = w[i] * dv[u]; }
    }
    if (hasA) xA = xA * dA;
    if (hasB) xB = xB * dB;
  }
  if (nt > 1) {
    // backward: columns k = n-1 .. ns+1 of U, rows ns .. k-1; byte offset of U(0, k): ((nzus + kk (kk-1)/2) - ns) * 8.
    // Columns kk >= 64: all of A, B above the diagonal; in whole groups of D while they last, the rest one by one.
    double ua[D], ub[D];
    int kk = nt - 1;
    int cb = ((nzus + kk * (kk - 1) / 2) - ns) * 8, str = -(kk - 1) * 8; // of the next column to load, and the step down from it
#define RG_DS_LOAD(S, WITH_B)                                                                                      \
  {                                                                                                               \
    ua[S] = tload_f64(bU, rA8, cb);                                                                                \
    if (WITH_B) ub[S] = tload_f64(bU, rB8, cb);                                                                    \
    cb += str; str += 8;                                                                                           \
  }
    if (kk - (D - 1) >= 64) {
#pragma unroll
      for (int s = 0; s < D; ++s) RG_DS_LOAD(s, true)
      for (; kk - (D - 1) >= 64; kk -= D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const int k1 = kk - s;
          const double t = rl(xB, k1 - 64);
          xA -= ua[s] * t;
          xB -= ((lane < k1 - 64) ? ub[s] : 0.0) * t;
          RG_DS_LOAD(s, true) // (runs up to D columns past the last group: loaded, not applied)
        }
      }
      cb = ((nzus + kk * (kk - 1) / 2) - ns) * 8; str = -(kk - 1) * 8;
    }
    for (; kk >= 64; --kk) { // at most D - 1 columns
      const double t = rl(xB, kk - 64);
      const double va = tload_f64(bU, rA8, cb), vb = tload_f64(bU, rB8, cb);
      cb += str; str += 8;
      xA -= va * t;
      xB -= ((lane < kk - 64) ? vb : 0.0) * t;
    }
    // columns kk < 64: A above the diagonal, nothing of B; kk <= 0 (the loop runs to a multiple of D): no rows, every lane off
    if (kk > 0) {
#pragma unroll
      for (int s = 0; s < D; ++s) RG_DS_LOAD(s, false)
      for (; kk > 0; kk -= D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
          const int k1 = kk - s;
          const double t = rl(xA, max(k1, 0));
          xA -= ((lane < k1) ? ua[s] : 0.0) * t;
          RG_DS_LOAD(s, false)
        }
      }
    }
#undef RG_DS_LOAD
  }
  if (hasA) w[rowA] = xA;
  if (hasB) w[rowB] = xB;
  lds_sync();
  dev_tri_sweep(N.Urc, Uv, N.nchunkU, w, lane); // rows < ns of every column
  for (int i = lane; i < n; i += 64) x[gptr(N.perm)[i]] = w[i];
  lds_sync();
}

} // namespace racgpu
