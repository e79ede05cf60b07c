// network.cpp -- host front end: reference text formats -> tables the device kernels run on.
// Reference behaviour mirrored (paths relative to the reference root):
//   chem_read_reactions   src/chemistry.f90:1427-1454   rows = lines not starting with '!' or blank
//   chem_load_reactions   src/chemistry.f90:1364-1424   '(7(A12), 3F9.0, 2F6.0, I3, X, A1, X, A2)', n_reac/n_prod rules
//   chem_parse_reactions  src/chemistry.f90:1221-1360   species index = order of first appearance; elements; 62-rows
//   getElements           src/chemistry.f90:1458-1529
//   getVibFreq            src/chemistry.f90:1532-1539
//   chem_get_dupli_reactions            src/chemistry.f90:1188-1217
//   chem_get_idx_for_special_species    src/chemistry.f90:1089-1185
//   chem_load_initial_abundances        src/chemistry.f90:1978-2024
// The sparsity pattern here is NOT chem_make_sparse_structure's (src/chemistry.f90:1858-1885): that one also
// marks reactions chem_ode_jac never touches (itype 53, 67, ...) and a full T row/column whose values are
// identically zero at fixed T.  Explicit zeros change no result of an LU without pivoting, so they are dropped.
#include "network.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <unordered_map>

namespace racgpu {

static const char *kElemNames[kNumElements] = {"+-", "E", "Grain", "H", "D", "He", "C", "N", "O", "Si",
                                               "S", "Fe", "Na", "Mg", "Cl", "P", "F", "Ne", "Ar", "K"};
static const double kElemMass[kNumElements] = {0.0, 5.45e-4, 0.0, 1.0, 2.0, 4.0, 12.0, 14.0, 16.0, 28.0,
                                               32.0, 56.0, 23.0, 24.0, 35.5, 31.0, 19.0, 20.18, 39.95, 39.1};

double fortran_real_field(const char *s, int w) {
  // list of what a Fortran Fw.0 edit descriptor accepts that strtod does not: embedded blanks,
  // D exponents, exponents written as a bare sign; an all-blank field reads as zero.
  std::string t;
  for (int i = 0; i < w && s[i]; ++i)
    if (s[i] != ' ') t.push_back((s[i] == 'D' || s[i] == 'd') ? 'e' : s[i]);
  if (t.empty()) return 0.0;
  for (size_t i = 1; i < t.size(); ++i)
    if ((t[i] == '+' || t[i] == '-') && t[i - 1] != 'e' && t[i - 1] != 'E') { t.insert(i, "e"); break; }
  return std::strtod(t.c_str(), nullptr);
}

static std::string field(const std::string &row, int pos, int w) { // trailing blanks trimmed
  std::string f = row.substr(pos, w);
  size_t e = f.find_last_not_of(' ');
  return e == std::string::npos ? std::string() : f.substr(0, e + 1);
}

static void element_counts(const std::string &name, std::array<int, kNumElements> &cnt) {
  // Longest-match tokenisation with the reference's exact replacement and digit rules.
  cnt.fill(0);
  const int len = (int)name.size();
  std::vector<int> owner(len + 3, 0); // 1-based element id owning each character, 0 = free
  auto ch = [&](int pos1) -> char { return pos1 >= 1 && pos1 <= len ? name[pos1 - 1] : ' '; };
  for (int e = 1; e <= kNumElements; ++e) {
    const int el = (int)std::strlen(kElemNames[e - 1]);
    for (int j = 1; j + el - 1 <= len; ++j) {
      if (name.compare(j - 1, el, kElemNames[e - 1]) != 0) continue;
      bool take = true;
      for (int k = j; k < j + el; ++k) {
        if (!owner[k]) continue;
        if ((int)std::strlen(kElemNames[owner[k] - 1]) >= el) { take = false; break; }
        cnt[owner[k] - 1] -= 1;
      }
      if (take) { for (int k = j; k < j + el; ++k) owner[k] = e; cnt[e - 1] += 1; }
    }
  }
  std::vector<int> belongs(owner);
  for (int i = 2; i <= len; ++i) {
    if (owner[i]) continue;
    for (int j = 1; j <= i - 1; ++j) if (owner[i - j]) { belongs[i] = owner[i - j]; break; }
    const char p = ch(i - 1), c = ch(i), nx = ch(i + 1);
    const bool pdig = p >= '0' && p <= '9', cdig = c >= '0' && c <= '9', ndig = nx >= '0' && nx <= '9';
    if (!pdig && cdig) {
      const int mult = ndig ? (c - '0') * 10 + (nx - '0') : (c - '0');
      if (mult == 0) continue;
      if (belongs[i] > 0) cnt[belongs[i] - 1] += mult - 1;
    } else if (c == '+') cnt[0] = 1;
    else if (c == '-') cnt[0] = -1;
  }
}

int HostNetwork::species_index(const std::string &name) const {
  for (int i = 0; i < nS; ++i) if (names[i] == name) return i + 1;
  return 0;
}

Kind HostNetwork::kind(int r) const {
  switch (R[r].itype) {
    case 5: case 6: case 21: case 64: return K_TWO;
    case 1: case 2: case 3: case 13: case 61: case 20: case 0: return K_ONE;
    case 62: return K_SURF;
    case 75: return K_SURF75;
    case 63: return K_SQ;
    default: return K_NONE;
  }
}

int HostNetwork::fss_selector(int r) const { // f_selfshielding_toISM/toStar, src/chemistry.f90:1007-1063
  const Reaction &x = R[r];
  if (std::strncmp(x.ctype, "PH", 2) != 0 && std::strncmp(x.ctype, "LA", 2) != 0) return 0;
  if (x.reac[0] <= 0) return 0;
  const std::string &nm = names[x.reac[0] - 1];
  if (nm == "H2") return 1;
  if (nm == "CO") return 2;
  if (nm == "H2O") return 3;
  if (nm == "OH") return 4;
  return 0;
}

void parse_network(const std::string &path, HostNetwork &net) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open network file " + path);
  std::string line;
  net.R.clear();
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '!' || line[0] == ' ') continue;
    std::string row = line.substr(0, 150);
    row.resize(150, ' ');
    Reaction x;
    for (int k = 0; k < 3; ++k) x.rname[k] = field(row, 12 * k, 12);
    for (int k = 0; k < 4; ++k) x.pname[k] = field(row, 36 + 12 * k, 12);
    for (int k = 0; k < 3; ++k) x.ABC[k] = fortran_real_field(row.c_str() + 84 + 9 * k, 9);
    for (int k = 0; k < 2; ++k) x.Trange[k] = fortran_real_field(row.c_str() + 111 + 6 * k, 6);
    {
      std::string it = field(row, 123, 3);
      it.erase(std::remove(it.begin(), it.end(), ' '), it.end());
      x.itype = it.empty() ? 0 : std::atoi(it.c_str());
    }
    x.ctype[0] = row[129]; x.ctype[1] = row[130]; x.ctype[2] = 0;
    x.reliability = row[127];
    for (int k = 0; k < 3; ++k) {
      if (!x.rname[k].empty()) x.n_reac++;
      if (x.rname[k] == "PHOTON" || x.rname[k] == "CRPHOT" || x.rname[k] == "CRP") x.n_reac--;
    }
    for (int k = 0; k < 4; ++k) {
      if (!x.pname[k].empty()) x.n_prod++;
      if (x.pname[k] == "PHOTON") x.n_prod--;
    }
    net.R.push_back(x);
  }
  net.nR = (int)net.R.size();
  if (net.nR == 0) throw std::runtime_error("no reactions in " + path);
  if (net.nR > 65535) throw std::runtime_error("more than 65535 reactions: 16-bit reaction indices exhausted");

  // species: order of first appearance, reactant slots 1..n_reac then product slots 1..n_prod
  std::unordered_map<std::string, int> seen;
  net.names.clear();
  auto intern = [&](const std::string &nm) {
    auto it = seen.find(nm);
    if (it != seen.end()) return it->second;
    net.names.push_back(nm);
    seen.emplace(nm, (int)net.names.size());
    return (int)net.names.size();
  };
  intern(net.R[0].rname[0]);
  for (Reaction &x : net.R) {
    for (int k = 0; k < x.n_reac && k < 3; ++k) x.reac[k] = intern(x.rname[k]);
    for (int k = 0; k < x.n_prod && k < 4; ++k) x.prod[k] = intern(x.pname[k]);
  }
  net.nS = (int)net.names.size();
  if (net.nS > 65534) throw std::runtime_error("more than 65534 species: 16-bit species indices exhausted");

  const double nan = std::numeric_limits<double>::quiet_NaN();
  net.elements.resize(net.nS);
  net.mass_num.assign(net.nS, 0.0);
  net.vib_freq.assign(net.nS, nan);
  net.Edesorb.assign(net.nS, nan);
  net.counterpart.assign(net.nS, -1);
  net.grain.clear();
  for (int i = 0; i < net.nS; ++i) {
    element_counts(net.names[i], net.elements[i]);
    double m = 0.0;
    for (int e = 0; e < kNumElements; ++e) m += (double)net.elements[i][e] * kElemMass[e];
    net.mass_num[i] = m;
    if (!net.names[i].empty() && net.names[i][0] == 'g') net.grain.push_back(i + 1);
  }
  const double kB = 1.3806503e-16, mp = 1.67262158e-24, Pi = 3.1415926535897932384626433, sites = 1e15;
  for (const Reaction &x : net.R)
    if (x.itype == 62 && x.reac[0] > 0) {
      const int a = x.reac[0];
      net.vib_freq[a - 1] = std::sqrt(2.0 * sites * kB * x.ABC[2] / (Pi * Pi) / (mp * net.mass_num[a - 1]));
      net.Edesorb[a - 1] = x.ABC[2];
      if (x.prod[0] > 0) { net.counterpart[x.prod[0] - 1] = a; net.counterpart[a - 1] = x.prod[0]; }
    }

  // duplicate sets: all lower-index reactions with the same ctype, itype, reactants and products
  net.dupli_ptr.assign(net.nR + 1, 0);
  net.dupli_list.clear();
  {
    // bucket by (itype, reac, prod) so the search is not quadratic
    std::unordered_map<std::string, std::vector<int>> bucket;
    for (int i = 0; i < net.nR; ++i) {
      const Reaction &x = net.R[i];
      char key[128];
      std::snprintf(key, sizeof key, "%d|%c%c|%d,%d,%d|%d,%d,%d,%d", x.itype, x.ctype[0], x.ctype[1], x.reac[0], x.reac[1],
                    x.reac[2], x.prod[0], x.prod[1], x.prod[2], x.prod[3]);
      std::vector<int> &b = bucket[key];
      for (int j : b) net.dupli_list.push_back(j + 1);
      net.dupli_ptr[i + 1] = (int)net.dupli_list.size();
      b.push_back(i);
    }
  }

  static const char *ten[10] = {"H2", "H", "E-", "C", "C+", "O", "O2", "CO", "H2O", "OH"};
  for (int k = 0; k < 10; ++k) net.idx10[k] = net.species_index(ten[k]);
  net.i_Grain0 = net.species_index("Grain0");
  net.i_GrainM = net.species_index("Grain-");
  net.i_GrainP = net.species_index("Grain+");
  net.i_gH = net.species_index("gH");
  net.i_gH2 = net.species_index("gH2");
  net.i_gH2O = net.species_index("gH2O");

  // sanity the device kernels rely on: every acted-on reaction has the reactants its flux formula reads
  for (int r = 0; r < net.nR; ++r) {
    const Reaction &x = net.R[r];
    const Kind k = net.kind(r);
    if (k == K_NONE) continue;
    if (x.reac[0] <= 0 || (k == K_TWO && x.reac[1] <= 0))
      throw std::runtime_error("reaction " + std::to_string(r + 1) + " (itype " + std::to_string(x.itype) +
                               ") lacks a reactant its rate law needs");
    if (x.itype == 21) {
      const int g1 = net.elements[x.reac[0] - 1][2], g2 = net.elements[x.reac[1] - 1][2];
      if (g1 != 0 && g2 != 0) throw std::runtime_error("Species name problem with type 21."); // reference: error_stop
      const int c = net.elements[x.reac[0] - 1][0] * net.elements[x.reac[1] - 1][0];
      if (c != -1 && c != 0) throw std::runtime_error("Charge problem with type 21.");
    }
  }
  build_jacobian_tables(net);
  build_symbolic(net);
  build_reference_layout(net);
}

void load_initial_abundances(const HostNetwork &net, const std::string &path, double *y0) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open initial abundance file " + path);
  std::fill(y0, y0 + net.nS, 0.0);
  std::string line;
  while (std::getline(in, line)) {
    std::string row = line.substr(0, 64);
    row.resize(64, ' ');
    const std::string nm = field(row, 0, 12);
    const int i = nm.empty() ? 0 : net.species_index(nm);
    if (i > 0) y0[i - 1] = fortran_real_field(row.c_str() + 12, 16);
  }
  const int iE = net.idx10[2];
  if (iE <= 0) throw std::runtime_error("network has no E-");
  double q = 0.0;
  for (int i = 0; i < net.nS; ++i) q += y0[i] * (double)net.elements[i][0];
  y0[iE - 1] = y0[iE - 1] + q;
  if (y0[iE - 1] < 0.0) throw std::runtime_error("Cannot neutralize the initial condition!");
  double totH = 0.0;
  for (int i = 0; i < net.nS; ++i) totH += (double)net.elements[i][3] * y0[i];
  for (int i = 0; i < net.nS; ++i) y0[i] = y0[i] / totH;
}

// Jacobian entry = (row species, column species).  For every reaction chem_ode_jac acts on and every distinct
// reactant column j, the derivative lands on each stoichiometric slot (reactants: subtract, products: add), in
// that order -- exactly the order in which the reference accumulates pdj (src/disk.f90:4868-4875).
void build_jacobian_tables(HostNetwork &net) {
  const int n = net.nS;
  struct Raw { int col, row; JacTerm t; };
  std::vector<Raw> raw;
  raw.reserve((size_t)net.nR * 8);
  for (int r = 0; r < net.nR; ++r) {
    const Reaction &x = net.R[r];
    const Kind k = net.kind(r);
    if (k == K_NONE) continue;
    int cols[2], nc = 0;
    cols[nc++] = x.reac[0];
    if (k == K_TWO && x.reac[1] != x.reac[0]) cols[nc++] = x.reac[1];
    for (int q = 0; q < nc; ++q) {
      JacTerm t{};
      t.rxn = (uint16_t)r; t.kind = (uint8_t)k;
      t.other = (uint16_t)(x.reac[0] - 1);
      t.other2 = (uint16_t)((k == K_TWO ? x.reac[1] : x.reac[0]) - 1);
      t.flags = 0;
      if (k == K_TWO && x.reac[0] == x.reac[1]) t.flags |= 2;
      if (q == 0) t.flags |= 4; // column is reactant 1
      for (int s = 0; s < x.n_reac; ++s) { JacTerm u = t; u.flags |= 1; raw.push_back({cols[q] - 1, x.reac[s] - 1, u}); }
      for (int s = 0; s < x.n_prod; ++s) raw.push_back({cols[q] - 1, x.prod[s] - 1, t});
    }
  }
  // pattern: union of touched (row, col) plus the diagonal (P = I - gamma*J needs it)
  std::vector<std::vector<int>> colrows(n);
  for (const Raw &e : raw) colrows[e.col].push_back(e.row);
  for (int j = 0; j < n; ++j) colrows[j].push_back(j);
  net.Jcolptr.assign(n + 1, 0);
  net.Jrow.clear();
  for (int j = 0; j < n; ++j) {
    auto &v = colrows[j];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (int i : v) net.Jrow.push_back(i);
    net.Jcolptr[j + 1] = (int)net.Jrow.size();
  }
  const int nnz = (int)net.Jrow.size();
  auto entry_of = [&](int row, int col) {
    const int *b = net.Jrow.data() + net.Jcolptr[col], *e = net.Jrow.data() + net.Jcolptr[col + 1];
    return (int)(std::lower_bound(b, e, row) - net.Jrow.data());
  };
  std::vector<int> count(nnz + 1, 0);
  std::vector<int> ent(raw.size());
  for (size_t q = 0; q < raw.size(); ++q) { ent[q] = entry_of(raw[q].row, raw[q].col); count[ent[q] + 1]++; }
  for (int e = 0; e < nnz; ++e) count[e + 1] += count[e];
  net.term_ptr = count;
  net.terms.resize(raw.size());
  std::vector<int> fill(count.begin(), count.end() - 1);
  for (size_t q = 0; q < raw.size(); ++q) net.terms[fill[ent[q]]++] = raw[q].t; // stable: keeps reaction order
}

// Minimum-degree ordering on the structure of J + J^T and symbolic LU of the permuted pattern (no pivoting).
// Plays the role of YSMP's ODRV + CDRV(path 5), which the reference reruns inside DLSODES for every cell and
// every solver reset (src/opkda1.f:1470-1504); here it runs once per network.
void build_symbolic(HostNetwork &net) {
  const int n = net.nS;
  Symbolic &S = net.sym;
  S.n = n;
  const int W = (n + 63) / 64;
  using u64 = unsigned long long;
  std::vector<u64> adj((size_t)n * W, 0);
  auto bit = [&](std::vector<u64> &m, int r, int c) -> bool { return (m[(size_t)r * W + (c >> 6)] >> (c & 63)) & 1ULL; };
  auto set = [&](std::vector<u64> &m, int r, int c) { m[(size_t)r * W + (c >> 6)] |= 1ULL << (c & 63); };
  auto clr = [&](std::vector<u64> &m, int r, int c) { m[(size_t)r * W + (c >> 6)] &= ~(1ULL << (c & 63)); };
  for (int j = 0; j < n; ++j)
    for (int q = net.Jcolptr[j]; q < net.Jcolptr[j + 1]; ++q) {
      const int i = net.Jrow[q];
      if (i != j) { set(adj, i, j); set(adj, j, i); }
    }
  S.perm.assign(n, 0); S.iperm.assign(n, 0);
  std::vector<char> alive(n, 1);
  for (int step = 0; step < n; ++step) {
    int best = -1, bestdeg = n + 1;
    for (int v = 0; v < n; ++v) {
      if (!alive[v]) continue;
      int d = 0;
      for (int w = 0; w < W; ++w) d += __builtin_popcountll(adj[(size_t)v * W + w]);
      if (d < bestdeg) { bestdeg = d; best = v; }
    }
    S.perm[step] = best; S.iperm[best] = step; alive[best] = 0;
    for (int u = 0; u < n; ++u) {
      if (!bit(adj, best, u)) continue;
      for (int w = 0; w < W; ++w) adj[(size_t)u * W + w] |= adj[(size_t)best * W + w];
      clr(adj, u, best); clr(adj, u, u);
    }
    for (int w = 0; w < W; ++w) adj[(size_t)best * W + w] = 0;
  }
  // symbolic elimination on row bitsets of the permuted matrix
  std::vector<u64> rows((size_t)n * W, 0);
  for (int j = 0; j < n; ++j)
    for (int q = net.Jcolptr[j]; q < net.Jcolptr[j + 1]; ++q) set(rows, S.iperm[net.Jrow[q]], S.iperm[j]);
  for (int k = 0; k < n; ++k)
    for (int i = k + 1; i < n; ++i) {
      if (!bit(rows, i, k)) continue;
      for (int w = (k + 1) >> 6; w < W; ++w) {
        u64 m = rows[(size_t)k * W + w];
        if (w == ((k + 1) >> 6)) m &= ~0ULL << ((k + 1) & 63);
        rows[(size_t)i * W + w] |= m;
      }
    }
  // Trailing dense block: the widest m <= 128 whose m x m corner of the filled pattern is >= 90 % dense; the few
  // missing positions are added as explicit zeros (they stay exactly zero numerically).
  {
    int best = n;
    for (int m = std::min(n, 128); m >= 8; --m) {
      const int s0 = n - m;
      long cnt = 0;
      for (int i = s0; i < n; ++i) for (int j = s0; j < n; ++j) cnt += bit(rows, i, j) ? 1 : 0;
      if ((double)cnt >= 0.90 * (double)m * (double)m) { best = s0; break; }
    }
    S.ns = best;
    for (int i = S.ns; i < n; ++i) for (int j = S.ns; j < n; ++j) set(rows, i, j);
  }
  std::vector<std::vector<int>> Lc(n), Uc(n);
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < j; ++i) if (bit(rows, i, j)) Uc[j].push_back(i);
    for (int i = j + 1; i < n; ++i) if (bit(rows, i, j)) Lc[j].push_back(i);
  }
  // Within column j of U the pivots k < ns are applied through the LDS work column.  Pivot k needs the final value
  // of row k, i.e. every earlier pivot k' of the same column with L(k,k') != 0 must have been applied: that gives
  // each pivot a level within its column.  Rows k < ns are stored sorted by (level, k), so the pivots of one level
  // are adjacent and independent of each other (the kernel fetches all their multipliers with one LDS read);
  // Ugrp marks the first pivot of every level.  Rows >= ns (dense tail) follow, ascending.
  std::vector<std::vector<int>> Uflag(n);
  for (int j = 0; j < n; ++j) {
    std::vector<int> &r = Uc[j];
    size_t m = 0;
    while (m < r.size() && r[m] < S.ns) ++m;
    std::vector<int> plev(m, 0);
    for (size_t a = 0; a < m; ++a)
      for (size_t b = 0; b < a; ++b) if (bit(rows, r[a], r[b])) plev[a] = std::max(plev[a], plev[b] + 1);
    std::vector<int> idx(m);
    for (size_t a = 0; a < m; ++a) idx[a] = (int)a;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return plev[a] < plev[b]; });
    std::vector<int> sorted(r);
    Uflag[j].assign(r.size(), 0);
    for (size_t a = 0; a < m; ++a) {
      sorted[a] = r[idx[a]];
      Uflag[j][a] = (a == 0 || plev[idx[a]] != plev[idx[a - 1]]) ? 1 : 0;
    }
    r = sorted;
  }
  // Storage.  The triangular solves stream the "sparse" part of L and U in dependency-level order (all columns
  // whose x_k is final at the same depth are adjacent, so a solve is one linear sweep over the value array) and do
  // the dense trailing block in registers:
  //   L stream: columns k < ns (their rows may lie anywhere), by forward level; then, outside the stream, the
  //             columns k >= ns (rows k+1..n-1, contiguous) in index order.
  //   U stream: for every column its rows < ns; the columns k >= ns come first, at level 0, because the backward
  //             solve finishes x_k, k >= ns, in registers before the stream starts; then the columns k < ns by
  //             backward level.  Outside the stream, for k >= ns, the rows ns..k-1 (contiguous) at Udptr[k].
  std::vector<int> llev(n, 0), ulev(n, 0);
  for (int k = 0; k < n; ++k) for (int i : Lc[k]) llev[i] = std::max(llev[i], llev[k] + 1);
  for (int k = n - 1; k >= 0; --k) {
    if (k >= S.ns) ulev[k] = 0;
    for (int i : Uc[k]) if (i < S.ns) ulev[i] = std::max(ulev[i], ulev[k] + 1);
  }
  auto level_order = [&](const std::vector<int> &lev, int kend, bool descending) {
    std::vector<int> order(kend);
    for (int k = 0; k < kend; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (lev[a] != lev[b]) return lev[a] < lev[b];
      return descending ? a > b : a < b;
    });
    return order;
  };
  S.Lcolptr.assign(n, 0); S.Lcolend.assign(n, 0); S.Lrow.clear(); S.Lcol.clear(); S.Llev.clear(); S.nlevL = 0;
  // a chunk of 64 stream entries never spans two levels: when the level changes the stream is padded to a multiple of
  // 64 with null entries (row == col == 0; their value slots are never written nor used)
  auto pad_level = [](std::vector<int> &row, std::vector<int> &col, std::vector<int> &lev, int newlev) {
    if (row.empty() || lev.back() == newlev) return;
    const int l = lev.back();
    while (row.size() % 64) { row.push_back(0); col.push_back(0); lev.push_back(l); }
  };
  for (int k : level_order(llev, S.ns, false)) {
    if (!Lc[k].empty()) pad_level(S.Lrow, S.Lcol, S.Llev, llev[k]);
    S.Lcolptr[k] = (int)S.Lrow.size();
    for (int i : Lc[k]) { S.Lrow.push_back(i); S.Lcol.push_back(k); S.Llev.push_back(llev[k]); }
    S.Lcolend[k] = (int)S.Lrow.size();
    if (!Lc[k].empty()) S.nlevL = std::max(S.nlevL, llev[k] + 1);
  }
  S.nzl_stream = (int)S.Lrow.size();
  for (int k = S.ns; k < n; ++k) {
    S.Lcolptr[k] = (int)S.Lrow.size();
    for (int i : Lc[k]) { S.Lrow.push_back(i); S.Lcol.push_back(k); S.Llev.push_back(0); }
    S.Lcolend[k] = (int)S.Lrow.size();
  }
  S.Ucolptr.assign(n, 0); S.Ucolend.assign(n, 0); S.Udptr.assign(n, 0); S.Urow.clear(); S.Ucol.clear(); S.Ulev.clear(); S.nlevU = 0;
  for (int k : level_order(ulev, n, true)) {
    { bool any = false; for (int i : Uc[k]) if (i < S.ns) any = true; if (any) pad_level(S.Urow, S.Ucol, S.Ulev, ulev[k]); }
    S.Ucolptr[k] = (int)S.Urow.size();
    for (int i : Uc[k]) if (i < S.ns) { S.Urow.push_back(i); S.Ucol.push_back(k); S.Ulev.push_back(ulev[k]); }
    S.Ucolend[k] = (int)S.Urow.size();
    if (S.Ucolend[k] > S.Ucolptr[k]) S.nlevU = std::max(S.nlevU, ulev[k] + 1);
  }
  S.nzu_stream = (int)S.Urow.size();
  for (int k = S.ns; k < n; ++k) {
    S.Udptr[k] = (int)S.Urow.size();
    for (int i = S.ns; i < k; ++i) { S.Urow.push_back(i); S.Ucol.push_back(k); S.Ulev.push_back(0); }
  }
  S.nzl = (int)S.Lrow.size(); S.nzu = (int)S.Urow.size();
  S.nzl_entries = 0; S.nzu_entries = 0; // without the alignment padding
  for (int k = 0; k < n; ++k) { S.nzl_entries += (int)Lc[k].size(); S.nzu_entries += (int)Uc[k].size(); }
  S.Ugrp.assign(S.nzu, 0);
  for (int j = 0; j < n; ++j) for (int a = 0; a < S.Ucolend[j] - S.Ucolptr[j]; ++a) S.Ugrp[S.Ucolptr[j] + a] = Uflag[j][a];
  if (n > 1023 || S.nlevL > 4094 || S.nlevU > 4094)
    throw std::runtime_error("network too large for the packed 10/10/12-bit solve schedule (n <= 1023 species)");
  // permuted columns of P
  S.Pcolptr.assign(n + 1, 0);
  std::vector<std::vector<std::pair<int, int>>> pc(n);
  for (int j = 0; j < n; ++j)
    for (int q = net.Jcolptr[j]; q < net.Jcolptr[j + 1]; ++q) pc[S.iperm[j]].push_back({S.iperm[net.Jrow[q]], q});
  S.Psrc.clear(); S.Prow.clear();
  for (int j = 0; j < n; ++j) {
    std::sort(pc[j].begin(), pc[j].end());
    for (auto &e : pc[j]) { S.Prow.push_back(e.first); S.Psrc.push_back(e.second); }
    S.Pcolptr[j + 1] = (int)S.Psrc.size();
  }
  S.Ppos.assign(S.Psrc.size(), 0);
  for (size_t q = 0; q < S.Psrc.size(); ++q) S.Ppos[S.Psrc[q]] = (int)q;
}

// The reference's pattern and the storage order of its P (see HostNetwork::ref_kref).
void build_reference_layout(HostNetwork &net) {
  const int nS = net.nS, neq = nS + 1;
  std::vector<std::vector<int>> col(neq); // rows per column
  for (const Reaction &x : net.R)
    for (int j = 0; j < x.n_reac && j < 3; ++j) {
      const int cj = x.reac[j] - 1;
      for (int k = 0; k < x.n_reac && k < 3; ++k) col[cj].push_back(x.reac[k] - 1);
      for (int k = 0; k < x.n_prod && k < 4; ++k) col[cj].push_back(x.prod[k] - 1);
    }
  for (int k = 0; k < 10; ++k) if (net.idx10[k] > 0) col[net.idx10[k] - 1].push_back(nS); // the T row at the special species
  col[nS].clear();
  for (int i = 0; i < neq; ++i) col[nS].push_back(i);                                        // the full T column
  net.ref_nnz0 = 0;
  std::vector<int> start(neq + 1, 0);
  for (int j = 0; j < neq; ++j) {
    auto &v = col[j];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    net.ref_nnz0 += (int)v.size();
    if (!std::binary_search(v.begin(), v.end(), j)) v.push_back(j); // DPREP appends a missing diagonal entry
    start[j + 1] = start[j] + (int)v.size();
  }
  net.ref_nnz1 = start[neq];
  net.ref_kref.assign(net.Jrow.size(), 0);
  for (int j = 0; j < nS; ++j)
    for (int q = net.Jcolptr[j]; q < net.Jcolptr[j + 1]; ++q) {
      const auto it = std::find(col[j].begin(), col[j].end(), net.Jrow[q]);
      if (it == col[j].end()) throw std::runtime_error("pattern entry outside the reference's pattern");
      net.ref_kref[q] = start[j] + (int)(it - col[j].begin());
    }
  // the T row (at the ten special species) and the T column in that storage (evolT: engine_integrate.hpp, ISTATE = 3)
  for (int k = 0; k < 10; ++k) {
    net.ref_kref_Trow[k] = -1;
    if (net.idx10[k] > 0) {
      const int j = net.idx10[k] - 1;
      const auto it = std::find(col[j].begin(), col[j].end(), nS);
      if (it != col[j].end()) net.ref_kref_Trow[k] = start[j] + (int)(it - col[j].begin());
    }
  }
  net.ref_kref_Tcol0 = start[nS];
  // IWORK(17) (LENRW) as the reference's DLSODES reports it for the networks shipped in data/ (measured with
  // oracle/_ref/ref_driver, section '# workspace'); it depends on YSMP's compressed index storage and cannot be derived
  // without it.  Other networks: racgpu_network_set_reference_lenrw, or 0 = the saved P is taken to survive ISTATE = 3.
  // Keyed on the pattern's own sizes as well (entries before / after DPREP's diagonal additions): a user network with the same
  // species and reaction counts but another sparsity pattern must not inherit another network's LENRW.
  static const struct { int nS, nR, nnz0, nnz1, lenrw; } known[] = {{464, 4767, 13348, 13351, 59430}, {467, 4801, 13469, 13472, 60324},
                                                                    {484, 5830, 14994, 14994, 65578}, {524, 6425, 18205, 18205, 82134}};
  net.ref_lenrw = 0;
  for (const auto &k : known)
    if (k.nS == nS && k.nR == net.nR && k.nnz0 == net.ref_nnz0 && k.nnz1 == net.ref_nnz1) net.ref_lenrw = k.lenrw;
}

} // namespace racgpu
