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

RG_DEV void wave_sync() {
  // one wave per workgroup: ordering between lanes needs only that the compiler keeps program order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

RG_DEV void dev_rates(const DevNet &N, const DevParams &P, const double *__restrict__ cell, double *__restrict__ rates, int lane) {
  const double Tgas = cell[0], Tdust = cell[1], n_gas = cell[2], D2H = cell[6], sites = cell[7];
  const double T300 = Tgas / 300.0;
  const double Tred = cst::kB_SI * Tgas / (cst::eCharge_SI * cst::eCharge_SI * cst::Coulomb_SI / (cell[3] * 1e-2));
  double JNegaPosi = 0.0, JChargeNeut = 0.0;
  if (Tred > 0.0) {
    JNegaPosi = (1.0 + 1.0 / Tred) * (1.0 + sqrt(2.0 / (2.0 + Tred)));
    JChargeNeut = 1.0 + sqrt(cst::Pi / 2.0 / Tred);
  }
  const double sig = cell[4];
  const double cr = cell[9] / cst::CR0 * exp(-cell[11] / cst::CRattenN), xr = cell[10] / cst::CR0;
  for (int r = lane; r < N.nR; r += 64) {
    const int it = N.r_itype[r];
    const double A = N.r_A[r], B = N.r_B[r], C = N.r_C[r], T0 = N.r_T0[r], T1 = N.r_T1[r];
    const int fsel = N.r_fss[r];
    const double fI = fsel ? cell[19 + fsel - 1] : 1.0, fS = fsel ? cell[23 + fsel - 1] : 1.0;
    const int a = N.r_re0[r], b = N.r_re1[r];
    double k = 0.0;
    switch (it) {
      case 5:
        if (Tgas <= 0.0) k = 0.0;
        else if (C < 0.0) {
          if (T0 > Tgas) k = A * pow(T0 / 300.0, B) * exp(-C / T0);
          else if (T1 < Tgas) k = A * pow(T1 / 300.0, B) * exp(-C / T1);
          else k = A * pow(T300, B) * exp(-C / Tgas);
        } else k = A * pow(T300, B) * exp(-C / Tgas);
        break;
      case 6:
        k = (T0 > Tgas || T1 < Tgas) ? 0.0 : A * pow(T300, B) * exp(-C / Tgas);
        break;
      case 1: k = A * (cr + xr); break;
      case 2: case 20: k = A * (C / (1.0 - cell[8]) * cr + xr); break;
      case 3:
        if (!(N.r_flags[r] & 1)) k = A * (cell[14] * exp(-C * cell[12]) * fI + cell[15] * exp(-C * cell[13]) * fS);
        else k = A * (cell[14] * exp(-C * cell[12]) * fI + cell[16] * fS);
        break;
      case 21:
        if (Tgas <= 0.0) k = 0.0;
        else {
          const double m = N.s_mass[N.r_id3[r]] * cst::mP;
          k = sqrt(8.0 * cst::kB / cst::Pi * Tgas / m) * sig * ((N.r_flags[r] & 4) ? JNegaPosi : JChargeNeut);
          if (sig <= 1e-30) k = 0.0;
        }
        break;
      case 13: k = cell[18] * A * fS; break;
      case 0:
        if (Tgas <= 0.0) k = 0.0;
        else {
          k = 0.5 * dev_sticking(N.s_mass[a], Tgas) * sig * sqrt(8.0 / cst::Pi * cst::kB * Tgas / cst::mP) * D2H;
          if (sig <= 1e-30) k = 0.0;
        }
        break;
      case 61:
        if (Tgas <= 0.0) k = 0.0;
        else {
          const double m = N.s_mass[a] * cst::mP;
          k = dev_sticking(N.s_mass[a], Tgas) * A * sig * cell[5] * sqrt(8.0 / cst::Pi * cst::kB * Tgas / m);
          if (sig <= 1e-30) k = 0.0;
        }
        break;
      case 62:
        k = N.s_vib[a] * (exp(-C / Tdust) + cst::CosmicDesorpPre * cr * exp(-C / cst::CosmicDesorpT));
        if (sig <= 1e-30) k = 0.0;
        k = k * (sites * D2H);
        break;
      case 63: {
        const double tmp = dev_mobility(P, N.s_vib[a], N.s_mass[a], N.s_Edes[a], Tdust) / sites;
        k = tmp / D2H * dev_branching(it, A, B, C, T0, Tdust);
        if ((N.r_flags[r] & 2) && sig <= 1e-30) k = 0.0;
      } break;
      case 64:
        k = (dev_mobility(P, N.s_vib[a], N.s_mass[a], N.s_Edes[a], Tdust) +
             dev_mobility(P, N.s_vib[b], N.s_mass[b], N.s_Edes[b], Tdust)) / (sites * D2H) * dev_branching(it, A, B, C, T0, Tdust);
        if (sig <= 1e-30) k = 0.0;
        break;
      case 75:
        k = (cell[17] * cst::HabingFlux + cell[14] * cst::HabingFlux * exp(-cst::UVext2Av * cell[12])) * sig * D2H * (A + B * Tdust);
        if (sig <= 1e-30) k = 0.0;
        break;
      default: k = 0.0;
    }
    k = k * cst::SecPerYear;
    if (N.r_nreac[r] == 2 && it < 60) k = k * n_gas;
    rates[r] = k;
  }
  wave_sync();
  // duplicate pruning (:948-964).  Which rate gets zeroed depends only on Tgas and the temperature ranges,
  // never on the rate values, so all reactions decide at once; the only writes are zeros.
  for (int r = lane; r < N.nR; r += 64) {
    const int q0 = N.dupli_ptr[r], q1 = N.dupli_ptr[r + 1];
    for (int q = q0; q < q1; ++q) {
      const int kk = N.dupli_list[q];
      const double v0 = fabs(N.r_T0[kk] - Tgas), v1 = fabs(N.r_T1[kk] - Tgas), v2 = fabs(N.r_T0[r] - Tgas), v3 = fabs(N.r_T1[r] - Tgas);
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

// chem_set_solver_flags_alt(j), reference src/chemistry.f90:205-268 (species part; T slot returned separately)
RG_DEV void dev_tolerances(const DevNet &N, const DevParams &P, double d2h, double *__restrict__ rtol, double *__restrict__ atol,
                           double &rT, double &aT, int lane) {
  double r, a;
  switch (P.tol_j) {
    case 1: r = P.RTOL; a = P.ATOL; rT = 1e-3; aT = 1e-1; break;
    case 2: r = fmin(P.RTOL * 1e1, 1e-4); a = fmin(P.ATOL * 1e5, 1e-25); rT = 1e-2; aT = 1e-1; break;
    case 3: r = fmin(P.RTOL * 1e2, 1e-4); a = fmin(P.ATOL * 1e10, 1e-20); rT = 1e-3; aT = 1e0; break;
    case 4: r = fmin(P.RTOL * 1e2, 1e-4); a = fmin(P.ATOL * 1e10, 1e-18); rT = 1e-3; aT = 1e0; break;
    default: r = fmin(P.RTOL * pow(2.0, (double)P.tol_j), 1e-3); a = fmin(P.ATOL * pow(1e2, (double)P.tol_j), 1e-15); rT = 1e-2; aT = 1e0;
  }
  for (int i = lane; i < N.nS; i += 64) {
    const int c = N.s_tolclass[i];
    double ri = r, ai = a;
    if (c == 1) { ri = fmax(P.RTOL, 1e-4); ai = fmax(P.ATOL, 1e-30); }
    else if (c == 2) { ri = 1e-4; ai = fmax(d2h * 1e-6, 1e-30); }
    else if (c == 3) { ri = fmax(P.RTOL, 1e-3); ai = fmax(P.ATOL, d2h * 1e-8); }
    rtol[i] = ri; atol[i] = ai;
  }
}

// ---------------------------------------------------------------------------------------------------------
// f(y): chem_ode_f, reference src/disk.f90:4569-4659 (fixed-T branch).  Reaction-major, ydot scattered with
// LDS f64 atomics (one wave owns the vector, so the result is deterministic).
// ---------------------------------------------------------------------------------------------------------
RG_DEV void dev_rhs(const DevNet &N, const double *__restrict__ rates, double nsite, const double *__restrict__ r_C,
                    const double *y, double *ydot, int lane) {
  for (int i = lane; i < N.nS; i += 64) ydot[i] = 0.0;
  wave_sync();
  for (int r = lane; r < N.nR; r += 64) {
    const uint64_t w0 = N.rhs_w0[r];
    const int kind = (int)(w0 & 0xff);
    if (kind == K_NONE_) continue;
    const int nre = (int)((w0 >> 8) & 0xff), a = (int)((w0 >> 16) & 0xffff), b = (int)((w0 >> 32) & 0xffff);
    const double k = rates[r], ya = y[a];
    double f;
    if (kind == K_TWO_) {
      const double yb = y[b];
      f = k * ya * yb;
      if (ya < 0.0 && yb < 0.0) f = -f;
    } else if (kind == K_ONE_) f = k * ya;
    else if (kind == K_SQ_) { f = k * ya * ya; if (ya < 0.0) f = -f; }
    else { // surface layer forms (62, 75)
      double t1 = nsite; if (kind == K_SURF75_) t1 = t1 * r_C[r];
      if (t1 <= 0.0) f = k;
      else { const double t = ya / t1; f = (t <= 1e-4) ? k * t : k * (1.0 - exp(-t)); }
    }
    const uint64_t w1 = N.rhs_w1[r], w2 = N.rhs_w2[r];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const int t = (int)(((s < 4 ? (w1 >> (16 * s)) : (w2 >> (16 * (s - 4))))) & 0xffff);
      if (t != 0xffff) atomicAdd(&ydot[t], s < nre ? -f : f);
    }
  }
  wave_sync();
}

// d(flux)/d(y_col) for one Jacobian term; chem_ode_jac, reference src/disk.f90:4764-4866
RG_DEV double dev_dflux(uint64_t term, const double *__restrict__ rates, const double *__restrict__ r_C, double nsite, const double *y) {
  const int rxn = (int)(term & 0xffff), sa = (int)((term >> 16) & 0xffff), kind = (int)((term >> 32) & 0xff);
  const int flags = (int)((term >> 40) & 0xff), sb = (int)((term >> 48) & 0xffff);
  const double k = rates[rxn];
  double v;
  if (kind == K_TWO_) {
    const double ya = y[sa], yb = y[sb];
    if (flags & 2) v = 2.0 * k * yb;
    else v = (flags & 4) ? k * yb : k * ya;
    if (ya < 0.0 && yb < 0.0) v = -v;
  } else if (kind == K_ONE_) v = k;
  else if (kind == K_SQ_) { const double ya = y[sa]; v = 2.0 * k * ya; if (ya < 0.0) v = -v; }
  else {
    double t2 = nsite; if (kind == K_SURF75_) t2 = t2 * r_C[rxn];
    if (t2 <= 0.0) v = 0.0;
    else { const double t1 = 1.0 / t2, t = y[sa] * t1; v = (t <= 1e-4) ? k * t1 : k * t1 * exp(-t); }
  }
  return (flags & 1) ? -v : v;
}

// J(y) gathered entry by entry in the reference's accumulation order, then P = I + con*J
// (DPRJS label 100-130, reference src/opkda1.f:1754-1767).  con = 1 and add_identity = false gives plain J.
RG_DEV void dev_build_P(const DevNet &N, const double *__restrict__ rates, double nsite, const double *y, double con,
                        bool add_identity, double *__restrict__ Pv, int lane) {
  wave_sync();
  for (int s = lane; s < N.jac_slots; s += 64) {
    const int e = N.jac_order[s];
    if (e < 0) continue;
    double sum = 0.0;
    const int t0 = N.term_ptr[e], t1 = N.term_ptr[e + 1];
    for (int t = t0; t < t1; ++t) sum += dev_dflux(N.terms[t], rates, N.r_C, nsite, y);
    double p = sum * con;
    if (add_identity && N.jac_isdiag[e]) p = p + 1.0;
    Pv[e] = p;
  }
  wave_sync();
}

// Left-looking column LDU of the permuted P: P' = L * D * U, L unit lower, U unit upper, D^-1 stored.
// w is the wave's LDS work column.  Returns false on an exactly zero pivot (DPRJS IERPJ = 1).
RG_DEV bool dev_lu(const DevNet &N, const double *__restrict__ Pv, double *__restrict__ Lv, double *__restrict__ Uv,
                   double *__restrict__ Dinv, double *w, int lane) {
  bool ok = true;
  const int n = N.nS;
  for (int j = 0; j < n; ++j) {
    const int u0 = N.Ucolptr[j], u1 = N.Ucolptr[j + 1], l0 = N.Lcolptr[j], l1 = N.Lcolptr[j + 1];
    for (int q = u0 + lane; q < u1; q += 64) w[N.Urow[q]] = 0.0;
    for (int q = l0 + lane; q < l1; q += 64) w[N.Lrow[q]] = 0.0;
    if (lane == 0) w[j] = 0.0;
    wave_sync();
    for (int q = N.Pcolptr[j] + lane; q < N.Pcolptr[j + 1]; q += 64) w[N.Prow[q]] = Pv[N.Psrc[q]];
    wave_sync();
    for (int qk = u0; qk < u1; ++qk) {
      const int k = N.Urow[qk];
      const double t = w[k]; // = d_k * u_kj, final
      const int c0 = N.Lcolptr[k], c1 = N.Lcolptr[k + 1];
      for (int q = c0 + lane; q < c1; q += 64) { const int i = N.Lrow[q]; w[i] -= Lv[q] * t; }
      wave_sync();
    }
    const double d = w[j];
    if (d == 0.0) ok = false;
    const double dinv = 1.0 / d;
    if (lane == 0) Dinv[j] = dinv;
    for (int q = u0 + lane; q < u1; q += 64) { const int k = N.Urow[q]; Uv[q] = w[k] * Dinv[k]; }
    for (int q = l0 + lane; q < l1; q += 64) Lv[q] = w[N.Lrow[q]] * dinv;
    wave_sync();
  }
  return ok;
}

// x <- P^-1 x with the factors above; x (species order) and w are LDS vectors (DSOLSS / CDRV path 4)
RG_DEV void dev_solve(const DevNet &N, const double *__restrict__ Lv, const double *__restrict__ Uv, const double *__restrict__ Dinv,
                      double *x, double *w, int lane) {
  const int n = N.nS;
  wave_sync();
  for (int i = lane; i < n; i += 64) w[i] = x[N.perm[i]];
  wave_sync();
  for (int k = 0; k < n; ++k) {
    const int c0 = N.Lcolptr[k], c1 = N.Lcolptr[k + 1];
    if (c0 == c1) continue;
    const double xk = w[k];
    for (int q = c0 + lane; q < c1; q += 64) { const int i = N.Lrow[q]; w[i] -= Lv[q] * xk; }
    wave_sync();
  }
  for (int i = lane; i < n; i += 64) w[i] = w[i] * Dinv[i];
  wave_sync();
  for (int k = n - 1; k >= 0; --k) {
    const int c0 = N.Ucolptr[k], c1 = N.Ucolptr[k + 1];
    if (c0 == c1) continue;
    const double xk = w[k];
    for (int q = c0 + lane; q < c1; q += 64) { const int i = N.Urow[q]; w[i] -= Uv[q] * xk; }
    wave_sync();
  }
  for (int i = lane; i < n; i += 64) x[N.perm[i]] = w[i];
  wave_sync();
}

} // namespace racgpu
