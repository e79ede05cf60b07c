// hc_tables.hpp -- plain-old-data tables of the heating/cooling terms (gas-temperature co-evolution, evolT) as the kernels see
// them, and the host loaders that fill them from the data files (data/README.md).
//
// Reference: module heating_cooling (src/heating_cooling.f90) with its table modules load_Neufeld_cooling_{H2,H2O,CO}
// (array constants of the Neufeld & Kaufman 1993 / Neufeld et al. 1995 cooling functions), the ion line-cooling look-up
// tables it reads through create_spline2d_from_file (src/binary_array_io.f90:19-90; linear: itype = 0) and the reaction heats
// chem_load_species_enthalpies / chem_get_reaction_heat derive from the species-enthalpy file (src/chemistry.f90:2027-2146).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace racgpu {

struct HostNetwork;

// heating_cooling_config (src/heating_cooling.f90:16-38) as far as the default branches read it, plus a_disk%base_alpha
// (src/disk.f90:32) and chemsol_params%maySwitchT (src/disk.f90:2070).  Layout = racgpu_hc_config of include/racgpu.h.
struct HcConfig {
  double heating_eff_chem, heating_eff_H2form, heating_eff_phd_H2, heating_eff_phd_H2O, heating_eff_phd_OH, cooling_gg_coeff, base_alpha;
  int32_t use_chemicalheatingcooling, use_Xray_heating, use_phdheating_H2, use_phdheating_H2OOH, use_mygasgraincooling, may_switch_T;
};

// load_Neufeld_cooling_H2: one axis (log10 T), 22 nodes
struct NeufeldH2 { double logT[22], L0[22], LLTE[22], n12[22], alpha[22]; };
// load_Neufeld_cooling_H2O: T axes of 6 nodes, log10 N~ axes of 10 (8 for the vibrational table); 2-D tables column-major (T fastest)
struct NeufeldH2O {
  double T_high[6], T_low_o[6], T_low_p[6], T_high_vib[6], N_high[10], N_high_vib[8], N_low_o[10], N_low_p[10];
  double L0_high[6], L0_low_o[6], L0_low_p[6];
  double LLTE_high[60], LLTE_vib[48], LLTE_low_o[60], LLTE_low_p[60], n12_high[60], n12_low_o[60], n12_low_p[60], a_high[60], a_low_o[60], a_low_p[60];
};
struct NeufeldCO {
  double T_high[6], T_high_vib[6], T_low[6], N_high[10], N_high_vib[8], N_low[10];
  double L0_high[6], L0_low[6];
  double LLTE_high[60], LLTE_vib[48], LLTE_low[60], n12_high[60], n12_low[60], a_high[60], a_low[60];
};
constexpr int kLutMax = 64;
struct IonLut { int nx, ny; double x[kLutMax], y[kLutMax], v[kLutMax * kLutMax]; }; // v(ix, iy) at ix + nx * iy

constexpr int kNHC = 28; // per-cell heating/cooling record (include/racgpu.h, RACGPU_H_*)
enum { H_EN_GAIN_TOT = 0, H_NCOL_STAR, H_PAH, H_MMW, H_OMEGA_K, H_DV_TURB, H_COHERENT, H_NEUFELD_G, H_NEUFELD_DVDZ, H_DUST_DEPL,
       H_VOLUME, H_NDUSTCOMPO, H_SIG_DUSTS, H_N_DUSTS = H_SIG_DUSTS + 4, H_TDUSTS = H_N_DUSTS + 4, H_EN_GAINS = H_TDUSTS + 4 };

struct DevHC {
  HcConfig cfg;
  NeufeldH2 h2; NeufeldH2O h2o; NeufeldCO co;
  IonLut nii, siii, feii;
  // reactions that release or take up heat (chem_net%iReacWithHeat, %heat [erg]); a, b = their two reactants (0-based)
  int nheat;
  const int *heat_rxn; const double *heat_val; const uint16_t *heat_a, *heat_b;
  // species the terms read (0-based, -1 absent): chem_idx_some_spe (src/chemistry.f90:1089-1187)
  int i_H2, i_HI, i_E, i_CI, i_CII, i_OI, i_O2, i_CO, i_H2O, i_OH, i_Hplus, i_Heplus, i_gH, i_NII, i_SiII, i_FeII;
  int idx10[10]; // chem_idx_some_spe%idx: the species whose columns carry a dT/dt entry (finite differences, src/disk.f90:4878-4891)
  // where the T row / T column of the reference's P sit in ITS storage (DevNet::Pkref for the species block): entry of the T row in
  // the column of idx10[k]; first entry of the T column (its nS + 1 entries follow in row order)
  int kref_row[10], kref_col0;
};

// host side: throws std::runtime_error
struct HostHC {
  std::vector<double> enthalpy; std::vector<char> has_enthalpy; // per species [erg], chem_load_species_enthalpies
  std::vector<int> heat_rxn; std::vector<double> heat_val;       // chem_get_reaction_heat
  NeufeldH2 h2; NeufeldH2O h2o; NeufeldCO co; IonLut nii, siii, feii;
  bool loaded = false;
};
void load_species_enthalpies(const HostNetwork &net, const std::string &path, HostHC &hc);
void load_neufeld_tables(const std::string &path, HostHC &hc);
void load_ion_lut(const std::string &path, IonLut &lut);

} // namespace racgpu
