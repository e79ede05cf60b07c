// network.hpp -- host-side model of a reaction network and everything cell-independent derived from it.
// Product code (no dependency on oracle/).  Reference behaviour cited per function in network.cpp.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace racgpu {

constexpr int kNumElements = 20;

// flux "kinds" the device kernels dispatch on (reference select case in src/disk.f90:4583-4643)
enum Kind : uint8_t {
  K_NONE = 0,   // itype not handled by chem_ode_f ("cycle")
  K_TWO = 1,    // 5, 6, 21, 64 : k*y_a*y_b, negated when both are negative
  K_ONE = 2,    // 1, 2, 3, 13, 61, 20, 0 : k*y_a
  K_SURF = 3,   // 62 : k*(1-exp(-y_a/N)), N = D2H*sites           (linearised for y_a/N <= 1e-4)
  K_SURF75 = 4, // 75 : same with N = D2H*sites*ABC(3)
  K_SQ = 5      // 63 : k*y_a^2, negated when y_a is negative
};

struct Reaction {
  std::string rname[3], pname[4];
  double ABC[3] = {0, 0, 0}, Trange[2] = {0, 0};
  int itype = 0;
  char ctype[3] = {' ', ' ', 0};
  char reliability = ' ';
  int n_reac = 0, n_prod = 0;
  int reac[3] = {0, 0, 0}, prod[4] = {0, 0, 0, 0}; // 1-based species indices
};

// One contribution to a Jacobian entry: d(flux of reaction r)/d(y_col) landing on one stoichiometric slot.
struct JacTerm {
  uint16_t rxn;    // reaction index (0-based)
  uint16_t other;  // species whose abundance multiplies k (K_TWO), or the reactant itself (K_SQ/K_SURF*)
  uint8_t kind;    // Kind
  uint8_t flags;   // bit0: subtract (reactant slot), bit1: both reactants identical (factor 2), bit2: 'other' is reactant 1 of the pair
  uint16_t other2; // K_TWO: the column species (needed for the both-negative sign rule)
};

struct Symbolic { // LU structure of P = I - gamma*J on the species block, cell independent
  int n = 0;
  std::vector<int> perm, iperm;          // perm[new] = old, iperm[old] = new (0-based)
  // L (strict lower, unit diagonal) and U (strict upper, unit diagonal after D scaling) are stored by columns of
  // the permuted matrix; the layout (a level-ordered "stream" part plus the dense trailing block) is described in
  // build_symbolic.  Column k of L occupies [Lcolptr[k], Lcolend[k]); the rows < ns of column k of U occupy
  // [Ucolptr[k], Ucolend[k]) and, for k >= ns, its rows ns..k-1 start at Udptr[k].  Per entry: row, column, level.
  std::vector<int> Lcolptr, Lcolend, Lrow, Lcol, Llev;
  std::vector<int> Ucolptr, Ucolend, Udptr, Urow, Ucol, Ulev;
  std::vector<int> Ugrp; // per U entry: 1 if this pivot (row < ns) opens a new level within its column, see build_symbolic
  int nzl_stream = 0, nzu_stream = 0; // entries of the streamed parts (they come first in the storage)
  int nlevL = 0, nlevU = 0;
  // Columns/rows >= ns form a trailing block that is (made) fully dense in L and U (at most 128 wide): the LU keeps
  // that part of its work column in registers, two rows per lane.
  int ns = 0;
  // P is STORED in permuted-column order (column j' = perm^-1 of species column, rows ascending): Pcolptr/Prow
  // describe that storage; Psrc[q] = CSC entry held at position q, Ppos = its inverse
  std::vector<int> Pcolptr, Psrc, Prow, Ppos;
  int nzl = 0, nzu = 0;                 // storage sizes of L and U (with the level-alignment padding of the streamed parts)
  int nzl_entries = 0, nzu_entries = 0; // entries of the factors proper
};

struct HostNetwork {
  int nS = 0, nR = 0;
  std::vector<std::string> names;
  std::vector<Reaction> R;
  std::vector<std::array<int, kNumElements>> elements;
  std::vector<double> mass_num, vib_freq, Edesorb;
  std::vector<int> counterpart; // 1-based, -1 none
  std::vector<int> grain;       // 1-based indices of species whose name starts with 'g'
  int idx10[10] = {0};          // H2 H E- C C+ O O2 CO H2O OH
  int i_Grain0 = 0, i_GrainM = 0, i_GrainP = 0, i_gH = 0, i_gH2 = 0, i_gH2O = 0;
  std::vector<int> dupli_ptr, dupli_list; // lower-index twins (1-based reaction numbers)
  // species-block Jacobian pattern (CSC, 0-based) restricted to reactions chem_ode_jac acts on
  std::vector<int> Jcolptr, Jrow;
  std::vector<int> term_ptr;       // per Jacobian entry
  std::vector<JacTerm> terms;      // in reference accumulation order (reaction order, reactant slots then product slots)
  Symbolic sym;
  // Where every entry of OUR pattern sits in the reference's own storage of P (DLSODES' WM segment): position in the
  // column-major IAN/JAN arrays DPREP builds from chem_make_sparse_structure's mask (reference src/chemistry.f90:1858-1885,
  // 1962-1972; src/opkda1.f:1386-1410: a missing diagonal entry is appended to its column).  Needed for one thing only:
  // on an ISTATE = 3 call DPREP zeroes NNZ words at the far end of a temporary work area (src/opkda1.f:1487-1494) which,
  // with the RWORK length the reference allocates (20 + 4 NNZ + 28 NEQ), overlaps the tail of the saved P.
  int ref_nnz0 = 0, ref_nnz1 = 0;   // entries of the reference's mask / of IAN-JAN with the diagonal added
  std::vector<int> ref_kref;        // per entry of Jrow (CSC order): 0-based position in that storage
  int ref_kref_Trow[10] = {0}, ref_kref_Tcol0 = 0; // the same for the T row's entries (columns of idx10) and the first entry of the T column
  int ref_lenrw = 0;                // IWORK(17) of the reference's DLSODES for this network (0 = unknown: P is taken to survive)

  int species_index(const std::string &name) const; // 1-based, 0 if absent
  Kind kind(int r) const;
  int fss_selector(int r) const; // 0 none, 1 H2, 2 CO, 3 H2O, 4 OH (PH/LA reactions only)
};

// throws std::runtime_error
void parse_network(const std::string &path, HostNetwork &net);
void load_initial_abundances(const HostNetwork &net, const std::string &path, double *y0);
void build_jacobian_tables(HostNetwork &net);
void build_symbolic(HostNetwork &net);
void build_reference_layout(HostNetwork &net);
double fortran_real_field(const char *s, int w);

} // namespace racgpu
