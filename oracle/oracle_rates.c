/* oracle_rates.c -- rate coefficients.  TEST INFRASTRUCTURE ONLY.
 * Restates src/chemistry.f90:591-966 chem_cal_rates with helpers :1007-1063 (self-shielding
 * selectors), :1068-1086 getStickingCoeff, :1542-1568 getMobility, :1571-1590 getBranchingRatio;
 * constants from src/sub_global_variables.f90 and src/chemistry.f90:179-181.
 * Scope: evol_dust_size = .false. (sig_dust = sigdust_ave), the fixed-T call made once per solve.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

#define PI 3.1415926535897932384626433
static const double kB_SI = 1.3806503e-23, kB_CGS = 1.3806503e-16, eCharge_SI = 1.602176487e-19,
                    Coulomb_SI = 8.9875517873681764e9, mP_CGS = 1.67262158e-24,
                    hbar_CGS = 1.054571628e-27, SecPerYear = 3600.0 * 24.0 * 365.0,
                    Habing_flux = 6e7, UVext2Av = 2.6, CR0 = 1.36e-17, CR_attenuate_N = 5.75e25,
                    CosmicDesorpPreFactor = 3.16e-19, CosmicDesorpGrainT = 70.0;

static double sticking(const orc_network *net, int iSpe, double T) { /* :1068-1086 */
  const double beta = 2.5, S0 = 1.0, T0_H = 0.5 * (52.0 + 25.0);
  double T0 = net->mass_num[iSpe - 1] * T0_H, r = T / T0;
  double tmp = (1.0 + r) * (1.0 + r) * sqrt(1.0 + r);
  return S0 * (1.0 + beta * r) / tmp;
}

static double mobility(const orc_params *p, double vibfreq, double massnum, double Edesorb, double Tdust) { /* :1542-1568 */
  const double width = 1e-8;
  double m = vibfreq * exp(fmax(-Edesorb * p->Diff2DesorRatio / Tdust,
                                -2.0 * width / hbar_CGS * sqrt(2.0 * massnum * (mP_CGS * kB_CGS * p->Diff2DesorRatio) * Edesorb)));
  if (fabs(massnum - 1.0) <= 1e-4 && p->use_special_gH_mobi && !p->update_gH_params_realtime) {
    double E = p->special_gH_E_diff;
    m = vibfreq * exp(fmax(-E / Tdust, -2.0 * width / hbar_CGS * sqrt(2.0 * massnum * (mP_CGS * kB_CGS * E))));
  }
  if (isnan(m)) m = 0.0;
  return m;
}

static double branching(const orc_network *net, int r, double Tdust) { /* :1571-1590 */
  if (net->itype[r] < 63) return 1.0;
  const double *ABC = net->ABC + 3 * r; double b;
  if (ABC[2] != 0.0)
    b = ABC[0] * exp(fmax(-ABC[2] / Tdust,
                          -2.0 * ABC[1] * 1e-8 / hbar_CGS * sqrt(2.0 * net->Trange[2 * r] * mP_CGS * kB_CGS * ABC[2])));
  else b = ABC[0];
  if (isnan(b)) b = 0.0;
  return b;
}

static double fss(const orc_network *net, int r, const double *cell, int toStar) { /* :1007-1063 */
  if (strcmp(net->ctype[r], "PH") && strcmp(net->ctype[r], "LA")) return 1.0;
  int a = net->reac[3 * r]; if (a <= 0) return 1.0;
  const char *nm = net->names[a - 1]; int base = toStar ? ORC_P_FSS_STAR_H2 : ORC_P_FSS_ISM_H2;
  if (!strcmp(nm, "H2")) return cell[base];
  if (!strcmp(nm, "CO")) return cell[base + 1];
  if (!strcmp(nm, "H2O")) return cell[base + 2];
  if (!strcmp(nm, "OH")) return cell[base + 3];
  return 1.0;
}

int orc_cal_rates(const orc_network *net, const orc_params *p, const double *cell, double *rates, double *R_H2_form) {
  const int nR = net->nR, nS = net->nS;
  const double Tgas = cell[ORC_P_TGAS], Tdust = cell[ORC_P_TDUST], n_gas = cell[ORC_P_NGAS];
  const double D2H = cell[ORC_P_D2H], sites = cell[ORC_P_SITES];
  double T300 = Tgas / 300.0;
  double Tred = kB_SI * Tgas / (eCharge_SI * eCharge_SI * Coulomb_SI / (cell[ORC_P_GRAIN_RADIUS] * 1e-2));
  double JNegaPosi, JChargeNeut;
  if (Tred > 0.0) {
    JNegaPosi = (1.0 + 1.0 / Tred) * (1.0 + sqrt(2.0 / (2.0 + Tred)));
    JChargeNeut = 1.0 + sqrt(PI / 2.0 / Tred);
  } else { JNegaPosi = 0.0; JChargeNeut = 0.0; }
  double sig_dust = cell[ORC_P_SIGDUST];
  double cr_rela = cell[ORC_P_ZETA_CR] / CR0 * exp(-cell[ORC_P_NCOL_ISM] / CR_attenuate_N);
  double xr_rela = cell[ORC_P_ZETA_X] / CR0;
  const double f_H2_cov_modi = 1.0;
  /* adsorb/desorb coefficients per species; the reference keeps them in module state initialised
   * to NaN (:1284-1285); here they are per call (differs only if a 63/moeq row precedes its 61/62). */
  double ads[2048], des[2048];
  if (nS > 2048) return -10;
  for (int i = 0; i < nS; i++) { ads[i] = NAN; des[i] = NAN; }
  if (R_H2_form) *R_H2_form = 0.0;

  for (int i = 0; i < nR; i++) {
    const double *ABC = net->ABC + 3 * i, *Tr = net->Trange + 2 * i;
    double k = 0.0;
    switch (net->itype[i]) {
      case 5:
        if (Tgas <= 0.0) k = 0.0;
        else if (ABC[2] < 0.0) {
          if (Tr[0] > Tgas) k = ABC[0] * pow(Tr[0] / 300.0, ABC[1]) * exp(-ABC[2] / Tr[0]);
          else if (Tr[1] < Tgas) k = ABC[0] * pow(Tr[1] / 300.0, ABC[1]) * exp(-ABC[2] / Tr[1]);
          else k = ABC[0] * pow(T300, ABC[1]) * exp(-ABC[2] / Tgas);
        } else k = ABC[0] * pow(T300, ABC[1]) * exp(-ABC[2] / Tgas);
        break;
      case 6:
        if (Tr[0] > Tgas) k = 0.0; else if (Tr[1] < Tgas) k = 0.0;
        else k = ABC[0] * pow(T300, ABC[1]) * exp(-ABC[2] / Tgas);
        break;
      case 1: k = ABC[0] * (cr_rela + xr_rela); break;
      case 2: case 20: k = ABC[0] * (ABC[2] / (1.0 - cell[ORC_P_ALBEDO]) * cr_rela + xr_rela); break;
      case 3:
        if (strcmp(net->reac_name1[i], "H2"))
          k = ABC[0] * (cell[ORC_P_G0_ISM] * exp(-ABC[2] * cell[ORC_P_AV_ISM]) * fss(net, i, cell, 0) +
                        cell[ORC_P_G0_STAR] * exp(-ABC[2] * cell[ORC_P_AV_STAR]) * fss(net, i, cell, 1));
        else
          k = ABC[0] * (cell[ORC_P_G0_ISM] * exp(-ABC[2] * cell[ORC_P_AV_ISM]) * fss(net, i, cell, 0) +
                        cell[ORC_P_G0_H2PHD] * fss(net, i, cell, 1));
        break;
      case 21:
        if (Tgas <= 0.0) k = 0.0;
        else {
          int id1 = net->reac[3 * i], id2 = net->reac[3 * i + 1], id3;
          if (id1 <= 0 || id2 <= 0) return -21;
          if (net->elements[(id1 - 1) * ORC_NELEM + 2] == 0) id3 = id1;
          else if (net->elements[(id2 - 1) * ORC_NELEM + 2] == 0) id3 = id2;
          else return -21; /* 'Species name problem with type 21.' -> error_stop */
          int c3 = net->elements[(id1 - 1) * ORC_NELEM] * net->elements[(id2 - 1) * ORC_NELEM];
          double m = net->mass_num[id3 - 1] * mP_CGS;
          if (c3 == -1) k = sqrt(8.0 * kB_CGS / PI * Tgas / m) * sig_dust * JNegaPosi;
          else if (c3 == 0) k = sqrt(8.0 * kB_CGS / PI * Tgas / m) * sig_dust * JChargeNeut;
          else return -22; /* 'Charge problem with type 21.' */
          if (sig_dust <= 1e-30) k = 0.0;
        }
        break;
      case 13: k = cell[ORC_P_LYA] * ABC[0] * fss(net, i, cell, 1); break;
      case 0:
        if (Tgas <= 0.0) k = 0.0;
        else {
          double s = sticking(net, net->reac[3 * i], Tgas);
          double tmp = sqrt(8.0 / PI * kB_CGS * Tgas / mP_CGS);
          k = 0.5 * s * sig_dust * tmp * D2H;
          if (sig_dust <= 1e-30) k = 0.0;
        }
        if (R_H2_form) *R_H2_form = k;
        break;
      case 61:
        if (Tgas <= 0.0) k = 0.0;
        else {
          int a = net->reac[3 * i];
          double s = sticking(net, a, Tgas), m = net->mass_num[a - 1] * mP_CGS;
          k = s * ABC[0] * sig_dust * cell[ORC_P_NDUST] * sqrt(8.0 / PI * kB_CGS * Tgas / m);
          if (sig_dust <= 1e-30) k = 0.0;
        }
        ads[net->reac[3 * i] - 1] = k;
        break;
      case 62: {
        double Eeff = ABC[2] * f_H2_cov_modi; int a = net->reac[3 * i];
        k = net->vib_freq[a - 1] * (exp(-Eeff / Tdust) + CosmicDesorpPreFactor * cr_rela * exp(-Eeff / CosmicDesorpGrainT));
        if (sig_dust <= 1e-30) k = 0.0;
        des[a - 1] = k;
        k = k * (sites * D2H);
      } break;
      case 63: {
        int i1 = net->reac[3 * i];
        double tmp = mobility(p, net->vib_freq[i1 - 1], net->mass_num[i1 - 1], net->Edesorb[i1 - 1] * f_H2_cov_modi, Tdust) / sites;
        double br = branching(net, i, Tdust);
        if (!strcmp(net->reac_name1[i], "gH")) {
          if (p->H2_form_use_moeq) {
            int ig = net->counterpart[i1 - 1];
            k = tmp / (tmp + des[i1 - 1]) * (ig > 0 ? ads[ig - 1] : NAN) / D2H;
          } else k = tmp / D2H * br;
          if (sig_dust <= 1e-30) k = 0.0;
          if (R_H2_form) *R_H2_form = k;
        } else k = tmp / D2H * br;
      } break;
      case 64: {
        int i1 = net->reac[3 * i], i2 = net->reac[3 * i + 1];
        double br = branching(net, i, Tdust);
        k = (mobility(p, net->vib_freq[i1 - 1], net->mass_num[i1 - 1], net->Edesorb[i1 - 1] * f_H2_cov_modi, Tdust) +
             mobility(p, net->vib_freq[i2 - 1], net->mass_num[i2 - 1], net->Edesorb[i2 - 1] * f_H2_cov_modi, Tdust)) /
            (sites * D2H) * br;
        if (sig_dust <= 1e-30) k = 0.0;
      } break;
      case 75: {
        double yield = ABC[0] + ABC[1] * Tdust;
        k = (cell[ORC_P_G0_PHOTODES] * Habing_flux + cell[ORC_P_G0_ISM] * Habing_flux * exp(-UVext2Av * cell[ORC_P_AV_ISM])) *
            sig_dust * D2H * yield;
        if (sig_dust <= 1e-30) k = 0.0;
      } break;
      default: k = 0.0;
    }
    k = k * SecPerYear;                                        /* :937 */
    if (net->n_reac[i] == 2 && net->itype[i] < 60) k = k * n_gas; /* :940-942 */
    rates[i] = k;
    /* duplicate pruning by nearest temperature range (:948-964); minloc = first minimum */
    for (int q = net->dupli_ptr[i]; q < net->dupli_ptr[i + 1]; q++) {
      int kk = net->dupli_list[q] - 1;
      double v[4] = {fabs(net->Trange[2 * kk] - Tgas), fabs(net->Trange[2 * kk + 1] - Tgas),
                     fabs(Tr[0] - Tgas), fabs(Tr[1] - Tgas)};
      int im = 0; for (int t = 1; t < 4; t++) if (v[t] < v[im]) im = t;
      if (im <= 1) { rates[i] = 0.0; break; }
      rates[kk] = 0.0;
    }
  }
  return 0;
}
