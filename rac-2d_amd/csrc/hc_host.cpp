// hc_host.cpp -- host loaders of the heating/cooling inputs (see hc_tables.hpp).  Product code, no dependency on oracle/.
#include "hc_tables.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "network.hpp"

namespace racgpu {

static std::string rtrim(const std::string &s) {
  size_t e = s.size();
  while (e > 0 && (s[e - 1] == ' ' || s[e - 1] == '\r' || s[e - 1] == '\n' || s[e - 1] == '\t')) --e;
  return s.substr(0, e);
}

// chem_load_species_enthalpies (reference src/chemistry.f90:2027-2079): rows `name(A12) value(F9.0)` taken from the first 32
// characters of a line; lines starting with '!' or a blank are skipped; kJ/mol -> erg through 1e3 / R * k_B; a species named
// twice keeps the later row.  Then chem_get_reaction_heat (:2083-2146): heat of a reaction = sum H(reactants) - sum H(products)
// for every itype-5 reaction that is not radiative (ctype RA / RR) and whose species all have an enthalpy; |heat| <= 1e-50 dropped.
void load_species_enthalpies(const HostNetwork &net, const std::string &path, HostHC &hc) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open species-enthalpy file " + path);
  const double nan = std::nan("");
  hc.enthalpy.assign(net.nS, nan);
  hc.has_enthalpy.assign(net.nS, 0);
  std::string line;
  while (std::getline(f, line)) {
    std::string s = line.substr(0, 32);
    s.resize(32, ' ');
    if (s[0] == '!' || s[0] == ' ') continue;
    const std::string name = rtrim(s.substr(0, 12));
    const double v = fortran_real_field(s.c_str() + 12, 9);
    for (int j = 0; j < net.nS; ++j)
      if (net.names[j] == name) {
        hc.enthalpy[j] = v * 1e3 / 8.314472 * 1.3806503e-16;
        hc.has_enthalpy[j] = 1;
        break;
      }
  }
  hc.heat_rxn.clear(); hc.heat_val.clear();
  for (int r = 0; r < net.nR; ++r) {
    const Reaction &x = net.R[r];
    if (x.itype != 5) continue;
    if ((x.ctype[0] == 'R' && x.ctype[1] == 'A') || (x.ctype[0] == 'R' && x.ctype[1] == 'R')) continue;
    bool has = true;
    double h = 0.0;
    for (int j = 0; j < x.n_reac && has; ++j) {
      if (!hc.has_enthalpy[x.reac[j] - 1]) has = false; else h = h + hc.enthalpy[x.reac[j] - 1];
    }
    for (int j = 0; j < x.n_prod && has; ++j) {
      if (!hc.has_enthalpy[x.prod[j] - 1]) has = false; else h = h - hc.enthalpy[x.prod[j] - 1];
    }
    if (has && std::fabs(h) > 1e-50) {
      if (x.reac[0] <= 0 || x.reac[1] <= 0) throw std::runtime_error("reaction with heat has fewer than two reactants (heating_chemical reads both)");
      hc.heat_rxn.push_back(r); hc.heat_val.push_back(h);
    }
  }
}

// data/neufeld_cooling_tables.dat: sections "# name n1 [n2]" followed by the values in column-major order (tools/extract_reference_tables.py)
void load_neufeld_tables(const std::string &path, HostHC &hc) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open Neufeld cooling tables " + path);
  std::map<std::string, std::vector<double>> t;
  std::string line, cur;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '!') continue;
    if (line[0] == '#') { std::istringstream is(line.substr(1)); is >> cur; t[cur].clear(); continue; }
    if (cur.empty()) continue;
    t[cur].push_back(std::strtod(line.c_str(), nullptr));
  }
  auto take = [&](const char *name, double *dst, size_t n) {
    auto it = t.find(name);
    if (it == t.end() || it->second.size() != n) throw std::runtime_error(std::string("Neufeld tables: missing or mis-sized array ") + name);
    std::memcpy(dst, it->second.data(), n * sizeof(double));
  };
  take("H2.log10_T_s", hc.h2.logT, 22); take("H2.log10_L0", hc.h2.L0, 22); take("H2.log10_L_LTE", hc.h2.LLTE, 22);
  take("H2.log10_n_12", hc.h2.n12, 22); take("H2.alpha_s", hc.h2.alpha, 22);
  NeufeldH2O &w = hc.h2o;
  take("H2O.T_high", w.T_high, 6); take("H2O.T_low_ortho", w.T_low_o, 6); take("H2O.T_low_para", w.T_low_p, 6); take("H2O.T_high_vib", w.T_high_vib, 6);
  take("H2O.log10N_high", w.N_high, 10); take("H2O.log10N_high_vib", w.N_high_vib, 8); take("H2O.log10N_low_ortho", w.N_low_o, 10);
  take("H2O.log10N_low_para", w.N_low_p, 10);
  take("H2O.log10_L0_high", w.L0_high, 6); take("H2O.log10_L0_low_ortho", w.L0_low_o, 6); take("H2O.log10_L0_low_para", w.L0_low_p, 6);
  take("H2O.log10_L_LTE_high", w.LLTE_high, 60); take("H2O.log10_X_L_LTE_high_vib", w.LLTE_vib, 48);
  take("H2O.log10_L_LTE_low_ortho", w.LLTE_low_o, 60); take("H2O.log10_L_LTE_low_para", w.LLTE_low_p, 60);
  take("H2O.log10_n_12_high", w.n12_high, 60); take("H2O.log10_n_12_low_ortho", w.n12_low_o, 60); take("H2O.log10_n_12_low_para", w.n12_low_p, 60);
  take("H2O.alpha_high", w.a_high, 60); take("H2O.alpha_low_ortho", w.a_low_o, 60); take("H2O.alpha_low_para", w.a_low_p, 60);
  NeufeldCO &c = hc.co;
  take("CO.T_high", c.T_high, 6); take("CO.T_high_vib", c.T_high_vib, 6); take("CO.T_low", c.T_low, 6);
  take("CO.log10N_high", c.N_high, 10); take("CO.log10N_high_vib", c.N_high_vib, 8); take("CO.log10N_low", c.N_low, 10);
  take("CO.log10_L0_high", c.L0_high, 6); take("CO.log10_L0_low", c.L0_low, 6);
  take("CO.log10_L_LTE_high", c.LLTE_high, 60); take("CO.log10_X_L_LTE_high_vib", c.LLTE_vib, 48); take("CO.log10_L_LTE_low", c.LLTE_low, 60);
  take("CO.log10_n_12_high", c.n12_high, 60); take("CO.log10_n_12_low", c.n12_low, 60); take("CO.alpha_high", c.a_high, 60); take("CO.alpha_low", c.a_low, 60);
}

// read_binary_array (reference src/binary_array_io.f90:19-61): a stream of f64: ndim, dims(1..ndim), x(nx), y(ny), val(nx, ny)
void load_ion_lut(const std::string &path, IonLut &lut) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open ion-cooling table " + path);
  std::vector<double> a;
  double v;
  while (f.read(reinterpret_cast<char *>(&v), sizeof v)) a.push_back(v);
  if (a.size() < 3 || (int)a[0] != 2) throw std::runtime_error("ion-cooling table " + path + ": not a 2-D table");
  const int nx = (int)a[1], ny = (int)a[2];
  if (nx < 2 || ny < 2 || nx > kLutMax || ny > kLutMax || a.size() != (size_t)(3 + nx + ny + nx * ny))
    throw std::runtime_error("ion-cooling table " + path + ": unexpected size");
  lut.nx = nx; lut.ny = ny;
  std::memset(lut.x, 0, sizeof lut.x); std::memset(lut.y, 0, sizeof lut.y); std::memset(lut.v, 0, sizeof lut.v);
  std::memcpy(lut.x, a.data() + 3, nx * sizeof(double));
  std::memcpy(lut.y, a.data() + 3 + nx, ny * sizeof(double));
  std::memcpy(lut.v, a.data() + 3 + nx + ny, (size_t)nx * ny * sizeof(double));
}

} // namespace racgpu
