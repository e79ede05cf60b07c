// engine_hc.hpp -- the gas heating and cooling terms on the device: dT/dt of a cell whose gas temperature co-evolves with its
// chemistry (chemsol_params%evolT).
//
// Reference: realtime_heating_cooling_rate (src/disk.f90:4664-4741) -> heating_minus_cooling (src/heating_cooling.f90:1204-1269)
// and the 28 term functions it sums (src/heating_cooling.f90:190-1201), with the reference's default switches
// (use_analytical_CII_OI = IonCoolingWithLut = .true., dust_gas_linear_couple = .false., no tandem dust-temperature iteration);
// get_alpha_viscosity_alt / get_ion_charge_y / get_alpha_viscosity (src/disk.f90:3415-3475); get_H2_form_rate (:4302-4315);
// table look-ups of load_Neufeld_cooling_{H2,H2O,CO}.f90 and spline2d_interpol with itype = 0 (src/interpolation.f90:25-58).
//
// One wave evaluates one cell: the scalar terms are computed by every lane alike (a few hundred operations), the two sums over
// the network (reaction heats, ion charge) are shared out over the lanes.  Each term keeps the reference's operand order; single
// precision literals of the reference (e.g. the exponent -0.58 of the [OI] 6300 critical density) are kept as such.
#pragma once
#include "engine_device.hpp"
#include "hc_tables.hpp"

namespace racgpu {

namespace hcc {
constexpr double very_small = 1e-100, frac_dust_lose_en = 0.8, PAH0 = 1.6e-7, eV2erg = 1.60217657e-12, hPlanck_SI = 6.62606896e-34,
                 TwoPi = 6.283185307179586476925, LyA_H2O = 1.2e-17, LyA_OH = 1.8e-18, beta_ion_neutral = 2e-9;
}

// 28 terms in the order of type_heating_cooling_rates_list (src/data_struct.f90:489-520), the net rate first
enum { HC_NET = 0, HC_H_PE, HC_H_H2FORM, HC_H_CR, HC_H_VIBH2, HC_H_CI, HC_H_PHD_H2, HC_H_PHD_H2O, HC_H_PHD_OH, HC_H_XRAY, HC_H_VISC, HC_H_CHEM,
       HC_C_PE, HC_C_VIBH2, HC_C_GG, HC_C_OI, HC_C_CII, HC_C_H2O_ROT, HC_C_H2O_VIB, HC_C_CO_ROT, HC_C_CO_VIB, HC_C_H2_ROT, HC_C_LYA, HC_C_FB,
       HC_C_FF, HC_C_NII, HC_C_SIII, HC_C_FEII, HC_C_OH_ROT, HC_NTERMS };

// the search all the Neufeld look-ups share: 1-based idx with y(idx-1) .. y(idx) the interval used (extrapolating at both ends)
RG_DEV int hc_locate(const RG_GLOBAL double *y, int n, double x) {
  if (y[0] >= x) return 2;
  if (y[n - 1] <= x) return n;
  int idx = 2;
  for (; idx <= n; ++idx) if (y[idx - 1] > x) break;
  return idx > n ? n : idx;
}
RG_DEV double hc_four_point(double x, double y, double x1, double x2, double y1, double y2, double z11, double z12, double z21, double z22) {
  const double k1 = (z12 - z11) / (y2 - y1), k2 = (z22 - z21) / (y2 - y1);
  return ((k2 - k1) / (x2 - x1) * (x - x1) + k1) * (y - y1) + (z21 - z11) / (x2 - x1) * (x - x1) + z11;
}
// z(T, log10N) on a table tab(nT, nN) (column-major) with axes Ta, Na: interpolated in log T and log10 N
RG_DEV double hc_table2(const RG_GLOBAL double *Ta, int nT, const RG_GLOBAL double *Na, int nN, const RG_GLOBAL double *tab, double T, double log10N) {
  const int i = hc_locate(Ta, nT, T), j = hc_locate(Na, nN, log10N);
  return hc_four_point(log(T), log10N, log(Ta[i - 2]), log(Ta[i - 1]), Na[j - 2], Na[j - 1], tab[(i - 2) + nT * (j - 2)], tab[(i - 2) + nT * (j - 1)],
                       tab[(i - 1) + nT * (j - 2)], tab[(i - 1) + nT * (j - 1)]);
}
RG_DEV double hc_spline1_lin(const RG_GLOBAL double *xi, const RG_GLOBAL double *yi, int stride, int n, double x) {
  // spline1d_interpol with zero second derivatives and extrapolate = .false. (src/interpolation.f90:141-205); yi(j) at yi[j * stride]
  if (isnan(x)) return x;
  if (xi[0] > x) return yi[0];
  if (xi[n - 1] < x) return yi[(n - 1) * stride];
  int j = 0;
  for (; j < n - 1; ++j) if (xi[j] <= x && xi[j + 1] >= x) break;
  const double dx = xi[j + 1] - xi[j];
  const double A = (xi[j + 1] - x) / dx, B = 1.0 - A;
  const double C = (A * A * A - A) * (dx * dx) / 6.0, D = (B * B * B - B) * (dx * dx) / 6.0;
  return A * yi[j * stride] + B * yi[(j + 1) * stride] + C * 0.0 + D * 0.0;
}
RG_DEV double hc_lut(const RG_GLOBAL IonLut &L, double x, double y) { // spline2d_interpol(x, y, spl, extrapolate=.false.)
  // every x node's value at y first (the reference interpolates all nx of them; only the bracketing ones enter the result)
  const int nx = L.nx, ny = L.ny;
  if (isnan(x)) return x;
  int j0, j1;
  if (L.x[0] > x) j0 = j1 = 0;
  else if (L.x[nx - 1] < x) j0 = j1 = nx - 1;
  else { int j = 0; for (; j < nx - 1; ++j) if (L.x[j] <= x && L.x[j + 1] >= x) break; j0 = j; j1 = j + 1; }
  const double v0 = hc_spline1_lin(L.y, L.v + j0, nx, ny, y);
  if (j0 == j1) return v0;
  const double v1 = hc_spline1_lin(L.y, L.v + j1, nx, ny, y);
  const double dx = L.x[j1] - L.x[j0];
  const double A = (L.x[j1] - x) / dx, B = 1.0 - A;
  const double C = (A * A * A - A) * (dx * dx) / 6.0, D = (B * B * B - B) * (dx * dx) / 6.0;
  return A * v0 + B * v1 + C * 0.0 + D * 0.0;
}
static __shared__ volatile double g_hc_terms[HC_NTERMS];
static __shared__ double g_hc_base[HC_NTERMS]; // dev_T_border: the terms of the unperturbed state
// the blocks of dev_heating_cooling in the order they are evaluated (bits of its mask)
enum { HCB_PE = 0, HCB_H2FORM, HCB_CR, HCB_VIBH2, HCB_CI, HCB_PHD_H2, HCB_PHD_H2O, HCB_PHD_OH, HCB_XRAY, HCB_VISC, HCB_CHEM, HCB_C_PE, HCB_C_VIBH2, HCB_C_GG,
       HCB_C_OI, HCB_C_CII, HCB_C_H2O, HCB_C_CO, HCB_C_H2, HCB_C_LYA, HCB_C_FB, HCB_C_FF, HCB_C_IONS, HCB_C_OH };
// Which blocks read the abundance of each of chem_ode_jac's ten T-row species (H2, H, E-, C, C+, O, O2, CO, H2O, OH, reference
// src/disk.f90:4878-4899): the chemical heating reads every abundance; the viscous term reads the charge of the positive ions (C+ here);
// H2 formation reads X(gH), or X(H) in a network without gH.  tests/test_gpu_evolT.py compares the row these masks give with the row of
// ten full evaluations, bit for bit.
#define HCM(b) (1u << (b))
constexpr unsigned kHcRowMask[10] = {
    HCM(HCB_CHEM) | HCM(HCB_VIBH2) | HCM(HCB_PHD_H2) | HCM(HCB_XRAY) | HCM(HCB_C_VIBH2) | HCM(HCB_C_GG) | HCM(HCB_C_CII) | HCM(HCB_C_H2O) | HCM(HCB_C_CO) | HCM(HCB_C_H2) | HCM(HCB_C_OH), // H2
    HCM(HCB_CHEM) | HCM(HCB_XRAY) | HCM(HCB_C_GG) | HCM(HCB_C_OI) | HCM(HCB_C_LYA),                                                                                              // H (+ H2 formation without gH)
    HCM(HCB_CHEM) | HCM(HCB_PE) | HCM(HCB_XRAY) | HCM(HCB_C_PE) | HCM(HCB_C_OI) | HCM(HCB_C_LYA) | HCM(HCB_C_FB) | HCM(HCB_C_FF) | HCM(HCB_C_IONS),                              // E-
    HCM(HCB_CHEM) | HCM(HCB_CI),                                                                                                                                                 // C
    HCM(HCB_CHEM) | HCM(HCB_C_CII) | HCM(HCB_VISC),                                                                                                                              // C+
    HCM(HCB_CHEM) | HCM(HCB_C_OI),                                                                                                                                               // O
    HCM(HCB_CHEM),                                                                                                                                                               // O2
    HCM(HCB_CHEM) | HCM(HCB_C_CO),                                                                                                                                               // CO
    HCM(HCB_CHEM) | HCM(HCB_PHD_H2O) | HCM(HCB_C_H2O),                                                                                                                           // H2O
    HCM(HCB_CHEM) | HCM(HCB_PHD_OH) | HCM(HCB_C_OH)};                                                                                                                            // OH
#undef HCM
#ifndef RG_HC_FENCES
#define RG_HC_FENCES 1
#endif
#if RG_HC_FENCES
#define HC_BLOCK() asm volatile("" ::: "memory")
#else
#define HC_BLOCK() do {} while (0)
#endif
RG_DEV double hc_tau2beta(double tau) { // tau2beta (src/sub_trivials.f90:1064-1085), factor 3
  if (tau <= 1e-4) return 1.0;
  const double tmp = 3.0 * tau;
  return tmp <= 40.0 ? (1.0 - exp(-tmp)) / tmp : 1.0 / tmp;
}

// dT/dt [K yr^-1] of one cell at (y, T).  cell: the RACGPU_NPAR record, hr: the RACGPU_NHC record, rates: the cell's rate
// coefficients AT T (chem_cal_rates has just run for it), rh2: R_H2_form_rate_coeff of that call.  terms (nullable): where
// lane 0 stores the 29 values of HC_* [erg s^-1 cm^-3].
#ifndef RG_HC_NOINLINE
#define RG_HC_NOINLINE 1 // a real call: inlined at its twelve call sites the terms' registers add to the integrator's (256 VGPRs + AGPR spills, one wave per SIMD)
#endif
#if RG_HC_NOINLINE
__device__ __attribute__((noinline))
#else
RG_DEV
#endif
double dev_heating_cooling(const DevNet &N, const RG_GLOBAL DevHC &H, const double *__restrict__ cell, const double *__restrict__ hr,
                                  const double *y, double T, const double *__restrict__ rates, double rh2, int lane, double *terms = nullptr, unsigned mask = ~0u) {
  // mask: bit b set = block b (HCB_*) is evaluated; a block that is not keeps the value g_hc_terms holds (dev_T_border: the terms of the
  // unperturbed state), and the sum is taken over all of them in the reference's order either way
  using namespace hcc;
  const RG_GLOBAL HcConfig &cfg = H.cfg;
  auto ab = [&](int i) { return i >= 0 ? y[i] : 0.0; };
  // (abundances and record fields are read where they are used: see HC_BLOCK)
#define X_H2 ab(H.i_H2)
#define X_HI ab(H.i_HI)
#define X_CI ab(H.i_CI)
#define X_CII ab(H.i_CII)
#define X_OI ab(H.i_OI)
#define X_NII ab(H.i_NII)
#define X_FeII ab(H.i_FeII)
#define X_SiII ab(H.i_SiII)
#define X_CO ab(H.i_CO)
#define X_H2O ab(H.i_H2O)
#define X_OH ab(H.i_OH)
#define X_E ab(H.i_E)
#define X_Hplus ab(H.i_Hplus)
#define X_Heplus ab(H.i_Heplus)
#define X_gH ab(H.i_gH)
#define n_gas cell[2]
#define Tdust cell[1]
  // get_H2_form_rate (H2_form_use_moeq = .false.)
  const double R_H2_form = H.i_gH >= 0 ? rh2 * X_gH * X_gH * n_gas : rh2 * X_HI * n_gas;
  // get_alpha_viscosity_alt: ion charge = sum of charge * y over the positively charged species with y >= 1e-30
  double q = 0.0;
  for (int i = lane; i < N.nS; i += 64) { const int ch = gptr(N.s_charge)[i]; const double yi = y[i]; if (yi >= 1e-30 && ch > 0) q += (double)ch * yi; }
  const double ion_charge = wave_sum(q);
#define omega_K hr[H_OMEGA_K]
  const double ambipolar_f = n_gas * ion_charge * beta_ion_neutral / omega_K;
  double alpha_visc = 0.0;
  if (ambipolar_f > 1e-20) {
    const double tmp = log(ambipolar_f), t1 = exp(-2.4 * tmp), t2 = exp(-0.3 * tmp);
    alpha_visc = 0.5 / sqrt(2500.0 * t1 + (8.0 * t2 + 1.0) * (8.0 * t2 + 1.0));
  }
  alpha_visc = cfg.base_alpha * alpha_visc;

#define G0_ISM cell[14]
#define G0_star cell[15]
#define Av_ISM cell[12]
#define Av_star cell[13]
#define Ncol_ISM cell[11]
#define Ncol_star hr[H_NCOL_STAR]
  const double chi_all = G0_ISM * exp(-cst::UVext2Av * Av_ISM) + G0_star * exp(-cst::UVext2Av * Av_star);
  const double chi_H2 = G0_ISM * exp(-cst::UVext2Av * Av_ISM) * cell[19] + cell[16] * cell[23];
#define PAH hr[H_PAH]
#define coh hr[H_COHERENT]
#define dv_turb hr[H_DV_TURB]
  // the terms go to LDS one by one (every lane writes the same value), and every block below starts from memory again (HC_BLOCK): kept
  // in registers with their inputs they make this scalar function the register-hungriest of the kernel (248 VGPRs)
  volatile double *r = g_hc_terms;
  // ---- heating -------------------------------------------------------------------------------------------------------------
  HC_BLOCK();
  if (mask & (1u << HCB_PE)) { // photoelectric, small grains (Bakes & Tielens 1994 as the reference codes it)
    double v = 0.0;
    if (!(X_E <= 0.0 || T <= 0.0)) {
      const double n_e = X_E * n_gas, tmp = chi_all * sqrt(T) / (n_e + very_small);
      const double t1 = (tmp <= 0.0 || isnan(tmp)) ? 0.0 : exp(0.73 * log(tmp));
      const double t2 = exp(0.70 * log(1e-4 * T));
      v = 1e-24 * chi_all * n_gas * PAH / PAH0 * (4.87e-2 / (1.0 + 4e-3 * t1) + 3.65e-2 * t2 / (1.0 + 2e-4 * tmp));
    }
    r[HC_H_PE] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_H2FORM)) r[HC_H_H2FORM] = 2.4e-12 * R_H2_form * cfg.heating_eff_H2form;
  HC_BLOCK();
  if (mask & (1u << HCB_CR)) r[HC_H_CR] = 1.5e-11 * cell[9] * n_gas * exp(-Ncol_ISM / cst::CRattenN);
  HC_BLOCK();
  if (mask & (1u << HCB_VIBH2)) {
    double v = 0.0;
    if (T > 0.0) { const double g10 = 5.4e-13 * sqrt(T); v = (n_gas * X_H2) * chi_H2 * 9.4e-22 / (1.0 + (1.9e-6 + chi_H2 * 4.7e-10) / (n_gas * g10)); }
    r[HC_H_VIBH2] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_CI)) r[HC_H_CI] = 2.2e-22 * X_CI * n_gas * chi_all;
  HC_BLOCK();
  if (mask & (1u << HCB_PHD_H2)) r[HC_H_PHD_H2] = cfg.use_phdheating_H2 ? 4e-14 * (n_gas * X_H2) * 3.4e-10 * chi_H2 * cfg.heating_eff_phd_H2 : 0.0;
  HC_BLOCK();
  if (mask & (1u << HCB_PHD_H2O)) r[HC_H_PHD_H2O] = cfg.use_phdheating_H2OOH ? (8.07e-12 * cfg.heating_eff_phd_H2O) * (n_gas * X_H2O) * LyA_H2O * (cell[18] * cell[25]) : 0.0;
  HC_BLOCK();
  if (mask & (1u << HCB_PHD_OH)) r[HC_H_PHD_OH] = cfg.use_phdheating_H2OOH ? (9.19e-12 * cfg.heating_eff_phd_OH) * (n_gas * X_OH) * LyA_OH * (cell[18] * cell[26]) : 0.0;
  HC_BLOCK();
  if (mask & (1u << HCB_XRAY)) { // X-ray heating per ion pair (Glassgold et al. 2012)
    double v = 0.0;
    if (cfg.use_Xray_heating) {
      double gam1 = 0.0, gam2 = 0.0;
      if (T > 0.0) { gam1 = 1e-12 * sqrt(T) * exp(-1000.0 / T); gam2 = 1.4e-12 * sqrt(T) * exp(-18100.0 / (T + 1200.0)); }
      const double tmp1 = X_H2 / (X_H2 + X_HI);
      double t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, t8 = 0;
      if (X_E > 0.0) {
        t2 = 7.95 * pow(X_E, 0.678); t3 = 2.17 * pow(X_E, 0.366); t4 = 22.0 * pow(X_E, 0.574); t5 = 23500.0 * pow(X_E, 0.955);
        t6 = 10700.0 * pow(X_E, 0.907); t7 = 7.09 * pow(X_E, 0.779); t8 = 6.88 * pow(X_E, 0.802);
      }
      const double eta_H_e = 1.0 - (1.0 - 0.117) / (1.0 + t2), eta_H2_e = 1.0 - (1.0 - 0.055) / (1.0 + t3);
      const double Q_el_rot = 37.0 * (X_HI * eta_H_e + X_H2 * eta_H2_e) / (X_HI + X_H2);
      const double Q_diss = 2.14 * tmp1 / (1.0 + t4);
      const double eps1 = 7.81 * (1.0 + t5), eps2 = 109.0 * (1.0 + t6);
      const double Q_dirvib = 19.0 * tmp1 * (1.0 / eps1 + 2.0 / eps2);
      const double epsB = 117.0 * (1.0 + t7), epsC = 132.0 * (1.0 + t8);
      const double Q_BCvib = 147.0 * tmp1 * (1.0 / epsB + 1.0 / epsC);
      double Q_vib = 0.0;
      if (gam1 + gam2 > 0.0) { const double ncrit = 2e-7 / (gam1 * X_HI + gam2 * X_H2); Q_vib = n_gas / (n_gas + ncrit) * (Q_dirvib + Q_BCvib); }
      v = cell[10] * n_gas * eV2erg * (Q_el_rot + Q_diss + Q_vib);
    }
    r[HC_H_XRAY] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_VISC)) {
    double v = 0.0;
    if (T > 0.0) {
      const double mmw = hr[H_MMW], rho = n_gas * cst::mP * mmw, c2 = cst::kB * T / (cst::mP * mmw), fcut = fmax(1.0 - T / 2e4, 0.0);
      v = 2.25 * alpha_visc * rho * c2 * omega_K * fcut;
    }
    r[HC_H_VISC] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_CHEM)) { // chemical heating: sum over the reactions with a heat, k * y_a * y_b * heat
    double v = 0.0;
    if (cfg.use_chemicalheatingcooling && T > 0.0) {
      double s = 0.0;
      for (int i = lane; i < H.nheat; i += 64) {
        const int i0 = ((const RG_GLOBAL int *)H.heat_rxn)[i];
        s += rates[i0] * y[((const RG_GLOBAL uint16_t *)H.heat_a)[i]] * y[((const RG_GLOBAL uint16_t *)H.heat_b)[i]] * ((const RG_GLOBAL double *)H.heat_val)[i];
      }
      v = wave_sum(s) * n_gas / cst::SecPerYear * cfg.heating_eff_chem;
    }
    r[HC_H_CHEM] = v;
  }
  // ---- cooling -------------------------------------------------------------------------------------------------------------
  HC_BLOCK();
  if (mask & (1u << HCB_C_PE)) {
    double v = 0.0;
    if (!(X_E <= 0.0 || T <= 0.0 || PAH <= 0.0)) {
      const double n_e = X_E * n_gas, tmp = chi_all * sqrt(T) / (n_e + very_small);
      if (tmp > 0.0) {
        const double t0 = log(T), t1 = exp(0.944 * t0), t2 = 0.735 * exp(-0.068 * t0), t3 = exp(t2 * log(tmp));
        v = PAH / PAH0 * 3.49e-30 * t1 * t3 * n_e * n_gas;
      }
    }
    r[HC_C_PE] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_VIBH2)) {
    double v = 0.0;
    if (T > 0.0) {
      const double g10 = 5.4e-13 * sqrt(T), A10 = 8.6e-7, D1 = 2.6e-11;
      v = 8.26e-13 * g10 * exp(-5988.0 / T) * (n_gas * n_gas * X_H2) * (A10 + chi_H2 * D1) / (g10 * n_gas + A10 + chi_H2 * D1);
    }
    r[HC_C_VIBH2] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_GG)) { // gas-grain collisions
    double v = 0.0;
    if (T > 0.0) {
      if (!cfg.use_mygasgraincooling) {
        v = 4.76e-33 * (1.0 - 0.8 * exp(-75.0 / T)) * n_gas * n_gas * sqrt(T) * (T - Tdust) * hr[H_DUST_DEPL] * (0.05e-4 / cell[3]);
      } else {
        const double cs_H = sqrt((8.0 / cst::Pi * cst::kB / cst::mP) * T), cs_H2 = cs_H / sqrt(2.0);
        const double tmp = 2.0 * cst::kB * cfg.cooling_gg_coeff * n_gas * (cs_H * (X_HI + X_Hplus) + cs_H2 * X_H2);
        const int nd = (int)hr[H_NDUSTCOMPO];
        for (int i = 0; i < 4; ++i) {
          if (i < nd) {
            const double coeff = tmp * hr[H_SIG_DUSTS + i] * hr[H_N_DUSTS + i];
            const double ex = fmax(coeff * (T - hr[H_TDUSTS + i]), -frac_dust_lose_en * hr[H_EN_GAINS + i] / hr[H_VOLUME]);
            v = v + ex;
          }
        }
      }
    }
    r[HC_C_GG] = v;
  }
#define Ncool fmin(fmin(Ncol_ISM, Ncol_star), n_gas * coh)
  HC_BLOCK();
  if (mask & (1u << HCB_C_OI)) { // [OI] 63, 146 um and 6300 A, analytic
    double v = 0.0;
    if (T > 0.0) {
      const double Z = X_OI / 3.2e-4;
      const double beta_63 = hc_tau2beta(Ncool * Z / 4.9e20), beta_146 = hc_tau2beta(Ncool * Z / 3.7e20);
      const double t1 = log(T), t2 = exp(0.45 * t1), t3 = exp(0.66 * t1);
      const double tmp1 = n_gas + beta_63 * 1.66e-5 / (1.35e-11 * t2), tmp2 = n_gas + beta_146 * 8.46e-5 / (4.37e-12 * t3);
      const double tmp3 = exp(98.0 / T), tmp4 = exp(228.0 / T);
      const double tmp5 = n_gas * n_gas + tmp3 * tmp1 * (3.0 * n_gas + tmp4 * 5.0 * tmp2);
      const double c63 = 3.15e-14 * 8.46e-5 * beta_63 * Z * 3.2e-4 * n_gas * tmp3 * 3.0 * n_gas * tmp1 / tmp5;
      const double c146 = 1.35e-14 * 1.66e-5 * beta_146 * Z * 3.2e-4 * n_gas * n_gas * n_gas / tmp5;
      const double n_cr_E = 1.3e6 * pow(T / 1e4, (double)(-0.58f)), n_cr_HI = 6.6e9; // (the exponent is a single precision literal in the reference)
      const double c6300 = hPlanck_SI * 4.7e14 * (6.5e-3 + 2.1e-3) * X_OI * (X_E / n_cr_E + X_HI / n_cr_HI) * (n_gas * n_gas);
      v = c63 + c146 + c6300;
    }
    r[HC_C_OI] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_CII)) {
    double v = 0.0;
    if (T > 0.0) {
      const double Z = X_CII / 1.4e-4, beta = hc_tau2beta(Ncool * Z / 6.5e20);
      v = 4.04e-24 * n_gas * Z * beta / (1.0 + 0.5 * exp(92.0 / T) * (1.0 + 2600.0 * beta / n_gas));
    }
    r[HC_C_CII] = v;
  }
#define n_H2 (n_gas * X_H2)
  const double ln10 = log(10.0);
  HC_BLOCK();
  if (mask & (1u << HCB_C_H2O)) { // Neufeld H2O, rotational and vibrational
    double vr = 0.0, vv = 0.0;
    if (!(X_H2O <= 0.0 || X_H2 <= 0.0 || T <= 0.0)) {
      const RG_GLOBAL NeufeldH2O &W = H.h2o;
      const double n_M = n_gas * X_H2O;
      const double log10N = log10(fmin(hr[H_NEUFELD_G] * n_M / (hr[H_NEUFELD_DVDZ] + very_small), n_M * Ncol_ISM / n_gas / (9.0 * dv_turb * 1e-5)));
      double L0, LLTE, n12, alpha;
      const double ortho = 0.75, para = 0.25;
      if (T >= W.T_high[0]) {
        const int i = hc_locate(W.T_high, 6, T);
        const double k = (W.L0_high[i - 1] - W.L0_high[i - 2]) / (log(W.T_high[i - 1]) - log(W.T_high[i - 2]));
        L0 = k * (log(T) - log(W.T_high[i - 2])) + W.L0_high[i - 2];
      } else {
        const int i = hc_locate(W.T_low_o, 6, T);
        const double k1 = (W.L0_low_o[i - 1] - W.L0_low_o[i - 2]) / (W.T_low_o[i - 1] - W.T_low_o[i - 2]);
        const double k2 = (W.L0_low_p[i - 1] - W.L0_low_p[i - 2]) / (W.T_low_p[i - 1] - W.T_low_p[i - 2]);
        L0 = ortho * ((T - W.T_low_o[i - 2]) * k1 + W.L0_low_o[i - 2]) + para * ((T - W.T_low_p[i - 2]) * k2 + W.L0_low_p[i - 2]);
      }
      L0 = exp(-L0 * ln10) + very_small;
      if (T >= 100.0) {
        LLTE = hc_table2(W.T_high, 6, W.N_high, 10, W.LLTE_high, T, log10N);
        n12 = hc_table2(W.T_high, 6, W.N_high, 10, W.n12_high, T, log10N);
        alpha = hc_table2(W.T_high, 6, W.N_high, 10, W.a_high, T, log10N);
      } else { // (the para tables are looked up with the ortho indices, as in the reference)
        const int i = hc_locate(W.T_low_o, 6, T), j = hc_locate(W.N_low_o, 10, log10N);
        auto both = [&](const RG_GLOBAL double *to, const RG_GLOBAL double *tp) {
          const double a = hc_four_point(log(T), log10N, log(W.T_low_o[i - 2]), log(W.T_low_o[i - 1]), W.N_low_o[j - 2], W.N_low_o[j - 1],
                                         to[(i - 2) + 6 * (j - 2)], to[(i - 2) + 6 * (j - 1)], to[(i - 1) + 6 * (j - 2)], to[(i - 1) + 6 * (j - 1)]);
          const double b = hc_four_point(log(T), log10N, log(W.T_low_p[i - 2]), log(W.T_low_p[i - 1]), W.N_low_p[j - 2], W.N_low_p[j - 1],
                                         tp[(i - 2) + 6 * (j - 2)], tp[(i - 2) + 6 * (j - 1)], tp[(i - 1) + 6 * (j - 2)], tp[(i - 1) + 6 * (j - 1)]);
          return ortho * a + para * b;
        };
        LLTE = both(W.LLTE_low_o, W.LLTE_low_p); n12 = both(W.n12_low_o, W.n12_low_p); alpha = both(W.a_low_o, W.a_low_p);
      }
      LLTE = exp(-LLTE * ln10) + very_small;
      n12 = exp(-n12 * ln10) + very_small;
      const double t1 = exp(alpha * log(n_H2 / n12));
      vr = n_H2 * n_M / (1.0 / L0 + n_H2 / LLTE + 1.0 / L0 * t1 * (1.0 - n12 * L0 / LLTE));
      const double tv = exp(-log(T) / 3.0);
      const double L0v = 1.03e-26 * T * exp(-47.5 * tv - 2325.0 / T) + very_small;
      const double LLv = exp(-hc_table2(W.T_high_vib, 6, W.N_high_vib, 8, W.LLTE_vib, T, log10N) * ln10 - 2325.0 / T) + very_small;
      vv = n_H2 * n_M / (1.0 / L0v + n_H2 / LLv);
    }
    r[HC_C_H2O_ROT] = vr; r[HC_C_H2O_VIB] = vv;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_CO)) { // Neufeld CO
    double vr = 0.0, vv = 0.0;
    if (!(X_CO <= 0.0 || X_H2 <= 0.0 || T <= 0.0)) {
      const RG_GLOBAL NeufeldCO &W = H.co;
      const double n_M = n_gas * X_CO;
      const double log10N = log10(fmin(hr[H_NEUFELD_G] * n_M / (hr[H_NEUFELD_DVDZ] + very_small), n_M * Ncol_ISM / n_gas / (9.0 * dv_turb * 1e-5)));
      double L0, LLTE, n12, alpha;
      if (T >= W.T_high[0]) {
        const int i = hc_locate(W.T_high, 6, T);
        const double k = (W.L0_high[i - 1] - W.L0_high[i - 2]) / (log(W.T_high[i - 1]) - log(W.T_high[i - 2]));
        L0 = k * (log(T) - log(W.T_high[i - 2])) + W.L0_high[i - 2];
      } else {
        const int i = hc_locate(W.T_low, 6, T);
        const double k = (W.L0_low[i - 1] - W.L0_low[i - 2]) / (W.T_low[i - 1] - W.T_low[i - 2]);
        L0 = k * (T - W.T_low[i - 2]) + W.L0_low[i - 2];
      }
      L0 = exp(-L0 * ln10) + very_small;
      if (T >= 100.0) {
        LLTE = hc_table2(W.T_high, 6, W.N_high, 10, W.LLTE_high, T, log10N);
        n12 = hc_table2(W.T_high, 6, W.N_high, 10, W.n12_high, T, log10N);
        alpha = hc_table2(W.T_high, 6, W.N_high, 10, W.a_high, T, log10N);
      } else {
        LLTE = hc_table2(W.T_low, 6, W.N_low, 10, W.LLTE_low, T, log10N);
        n12 = hc_table2(W.T_low, 6, W.N_low, 10, W.n12_low, T, log10N);
        alpha = hc_table2(W.T_low, 6, W.N_low, 10, W.a_low, T, log10N);
      }
      LLTE = exp(-LLTE * ln10) + very_small;
      n12 = exp(-n12 * ln10) + very_small;
      vr = n_H2 * n_M / (1.0 / L0 + n_H2 / LLTE + 1.0 / L0 * pow(n_H2 / n12, alpha) * (1.0 - n12 * L0 / LLTE));
      const double tv = exp(-log(T) / 3.0);
      const double L0v = 1.83e-26 * T * exp(-68.0 * tv - 3080.0 / T) + very_small;
      const double LLv = exp(-hc_table2(W.T_high_vib, 6, W.N_high_vib, 8, W.LLTE_vib, T, log10N) * ln10 - 3080.0 / T) + very_small;
      vv = n_H2 * n_M / (1.0 / L0v + n_H2 / LLv);
    }
    r[HC_C_CO_ROT] = vr; r[HC_C_CO_VIB] = vv;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_H2)) { // Neufeld H2 rotational
    double v = 0.0;
    if (!(T <= 0.0 || X_H2 <= 0.0)) {
      const RG_GLOBAL NeufeldH2 &W = H.h2;
      const double lT = log10(T);
      const int i = hc_locate(W.logT, 22, lT);
      auto lin = [&](const RG_GLOBAL double *t) { return (t[i - 1] - t[i - 2]) / (W.logT[i - 1] - W.logT[i - 2]) * (lT - W.logT[i - 2]) + t[i - 2]; };
      double L0 = exp(-lin(W.L0) * ln10 - 509.0 / T), LLTE = exp(-lin(W.LLTE) * ln10 - 509.0 / T);
      const double n12 = exp(lin(W.n12) * ln10);
      double alpha = lin(W.alpha); if (alpha < 0.0) alpha = 0.0;
      L0 = L0 + very_small; LLTE = LLTE + very_small;
      if (alpha > 0.0) { const double t1 = exp(alpha * log(n_H2 / n12)); v = n_H2 * n_H2 / (1.0 / L0 + n_H2 / LLTE + 1.0 / L0 * t1 * (1.0 - n12 * L0 / LLTE)); }
      else v = n_H2 * n_H2 / (1.0 / L0 + n_H2 / LLTE);
    }
    r[HC_C_H2_ROT] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_LYA)) r[HC_C_LYA] = T > 0.0 ? 7.3e-19 * (n_gas * n_gas) * X_HI * X_E * exp(-118400.0 / T) : 0.0;
  HC_BLOCK();
  if (mask & (1u << HCB_C_FB)) {
    double v = 0.0;
    if (T > 0.0) {
      const double n_p = n_gas * X_Hplus, n_E = n_gas * X_E, t1 = log(T / 1e4 / 1.0), t2 = exp(t1 * (-0.7131 - 0.0115 * t1));
      v = n_E * n_p * 4.13e-13 * 1.0 * t2 * (0.787 - 0.0230 * t1) * cst::kB * T;
    }
    r[HC_C_FB] = v;
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_FF)) r[HC_C_FF] = T > 0.0 ? 1.4e-27 * 1.0 * sqrt(T) * 1.3 * (n_gas * X_E) * (n_gas * (X_Hplus + X_Heplus)) : 0.0;
  HC_BLOCK();
  if (mask & (1u << HCB_C_IONS)) {
    auto ion = [&](double X, const RG_GLOBAL IonLut &L) {
      if (X <= 1e-15 || X_E <= 0.0 || n_gas <= 0.0 || T <= 0.0) return 0.0;
      return X * n_gas * exp(ln10 * hc_lut(L, log10(X_E * n_gas), log10(T)));
    };
    r[HC_C_NII] = ion(X_NII, H.nii); r[HC_C_SIII] = ion(X_SiII, H.siii); r[HC_C_FEII] = ion(X_FeII, H.feii);
  }
  HC_BLOCK();
  if (mask & (1u << HCB_C_OH)) { // OH rotational (Hollenbach & McKee 1989; Gorti & Hollenbach 2004)
    double v = 0.0;
    if (!(X_OH <= 0.0 || X_H2 < 0.0 || X_H2 >= 1.0 || T <= 0.0)) {
      const double A0 = 7.6e-4, E0 = 5.4, sig = 8e-16, eta = 10.0;
      const double Nn = X_OH * n_gas * coh, N_tau = 1.18e7 * dv_turb * 1e-5 * (E0 * E0 * E0) / A0;
      const double tau = 4.0 * Nn / N_tau / (eta * T / E0);
      const double te = tau / exp(1.0);
      const double ctau = tau * sqrt(TwoPi * log(2.13 + te * te));
      const double v_T = sqrt((8.0 / cst::Pi * cst::kB / cst::mP) * T);
      const double tmp = 4.0 * (T / E0) * A0 / (n_gas * (1.0 - X_H2) * sig * v_T);
      const double ym = log(1.0 + ctau / (1.0 + 10.0 * tmp));
      const double tmp1 = (2.0 + ym + 0.6 * (ym * ym)) / (1.0 + ctau + tmp + 1.5 * sqrt(tmp));
      const double L = 2.0 * cst::kB * (T * T) * A0 / E0 * tmp1;
      v = L * n_gas * X_OH;
    }
    r[HC_C_OH_ROT] = v;
  }
  HC_BLOCK();
  // heating_minus_cooling, in the reference's order of summation
  double net = r[HC_H_PE] + r[HC_H_H2FORM] + r[HC_H_CR] + r[HC_H_VIBH2] + r[HC_H_CI] + r[HC_H_PHD_H2] + r[HC_H_PHD_H2O] + r[HC_H_PHD_OH] + r[HC_H_XRAY] +
               r[HC_H_VISC] + r[HC_H_CHEM];
  net = net - r[HC_C_PE] - r[HC_C_VIBH2] - r[HC_C_GG] - r[HC_C_OI] - r[HC_C_CII] - r[HC_C_H2O_ROT] - r[HC_C_H2O_VIB] - r[HC_C_CO_ROT] - r[HC_C_CO_VIB] -
        r[HC_C_H2_ROT] - r[HC_C_LYA] - r[HC_C_FB] - r[HC_C_FF] - r[HC_C_NII] - r[HC_C_SIII] - r[HC_C_FEII] - r[HC_C_OH_ROT];
  HC_BLOCK();
  r[HC_NET] = net;
  if (terms && lane == 0) for (int k = 0; k < HC_NTERMS; ++k) terms[k] = r[k];
  HC_BLOCK();
  return uniform_d(net * cst::SecPerYear / (n_gas * cst::kB));
#undef X_H2
#undef X_HI
#undef X_CI
#undef X_CII
#undef X_OI
#undef X_NII
#undef X_FeII
#undef X_SiII
#undef X_CO
#undef X_H2O
#undef X_OH
#undef X_E
#undef X_Hplus
#undef X_Heplus
#undef X_gH
#undef n_gas
#undef Tdust
#undef G0_ISM
#undef G0_star
#undef Av_ISM
#undef Av_star
#undef Ncol_ISM
#undef Ncol_star
#undef PAH
#undef coh
#undef dv_turb
#undef omega_K
#undef Ncool
#undef n_H2
}

} // namespace racgpu
