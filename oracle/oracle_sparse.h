/* oracle_sparse.h -- internal to oracle/ (TEST INFRASTRUCTURE ONLY). */
#ifndef RAC2D_ORACLE_SPARSE_H
#define RAC2D_ORACLE_SPARSE_H
typedef struct {
  int n, nnz;
  int *IAN, *JAN;     /* CSC pattern with diagonals added (1-based), values array is on this */
  int *perm, *iperm;  /* perm[new] = old, iperm[old] = new (0-based) */
  int *Lp, *Lj, *Up, *Uj, nzl, nzu; /* strict lower / strict upper rows of the permuted LU */
  int *Arp, *Ak, *Ac; /* rows of the permuted A: value index, permuted column */
} orc_symbolic;
orc_symbolic *orc_symbolic_build(int n, const int *IA, const int *JA);
void orc_symbolic_free(orc_symbolic *);
int orc_numeric_lu(const orc_symbolic *, const double *A, double *L, double *U, double *Dinv, double *w);
void orc_lu_solve(const orc_symbolic *, const double *L, const double *U, const double *Dinv, double *x, double *z);
#endif
