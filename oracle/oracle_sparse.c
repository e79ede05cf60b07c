/* oracle_sparse.c -- sparse LU without pivoting on a static pattern.  TEST INFRASTRUCTURE ONLY.
 *
 * Stands in for the Yale Sparse Matrix Package calls made by DLSODES (src/opkda1.f:1995-3800:
 * ODRV minimum-degree ordering, CDRV path 5 symbolic LU, path 2 numeric LU, path 4 solve; driven
 * from DPREP src/opkda1.f:1470-1504, DPRJS :1826-1838, DSOLSS :1921-1923).  Like YSMP it orders
 * symmetrically by minimum degree on the structure of M + M^T, factors without pivoting and
 * solves with the factors; it is NOT a transcription of YSMP: the elimination order may differ in
 * tie-breaks and YSMP factors as L*D*U where this factors as L*U, so results agree to rounding,
 * not bitwise.  (Fill measured against the reference's own NZL/NZU in tests/test_oracle.py.)
 */
#include "oracle_sparse.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned long long u64;
#define BIT(row, j) (((row)[(j) >> 6] >> ((j) & 63)) & 1ULL)
#define SET(row, j) ((row)[(j) >> 6] |= 1ULL << ((j) & 63))
#define CLR(row, j) ((row)[(j) >> 6] &= ~(1ULL << ((j) & 63)))

orc_symbolic *orc_symbolic_build(int n, const int *IA, const int *JA) {
  orc_symbolic *S = calloc(1, sizeof *S);
  S->n = n;
  int W = (n + 63) / 64;
  /* augmented pattern: add missing diagonals at the end of each column (DPREP :1373-1393) */
  S->IAN = malloc((size_t)(n + 1) * sizeof(int));
  S->JAN = malloc((size_t)(IA[n] - 1 + n) * sizeof(int));
  int knew = 0; S->IAN[0] = 1;
  for (int j = 0; j < n; j++) {
    int found = 0;
    for (int k = IA[j] - 1; k < IA[j + 1] - 1; k++) { if (JA[k] == j + 1) found = 1; S->JAN[knew++] = JA[k]; }
    if (!found) S->JAN[knew++] = j + 1;
    S->IAN[j + 1] = knew + 1;
  }
  S->nnz = knew;
  /* minimum degree on M + M^T (ODRV flag 1), exact external degree, lowest index wins ties */
  u64 *adj = calloc((size_t)n * W, sizeof(u64));
  for (int j = 0; j < n; j++)
    for (int k = S->IAN[j] - 1; k < S->IAN[j + 1] - 1; k++) {
      int i = S->JAN[k] - 1; if (i == j) continue;
      SET(adj + (size_t)i * W, j); SET(adj + (size_t)j * W, i);
    }
  S->perm = malloc((size_t)n * sizeof(int)); S->iperm = malloc((size_t)n * sizeof(int));
  char *alive = malloc((size_t)n); memset(alive, 1, (size_t)n);
  for (int step = 0; step < n; step++) {
    int best = -1, bestdeg = n + 1;
    for (int v = 0; v < n; v++) if (alive[v]) {
      int d = 0; for (int w = 0; w < W; w++) d += __builtin_popcountll(adj[(size_t)v * W + w]);
      if (d < bestdeg) { bestdeg = d; best = v; }
    }
    S->perm[step] = best; S->iperm[best] = step; alive[best] = 0;
    u64 *av = adj + (size_t)best * W;
    for (int u = 0; u < n; u++) if (BIT(av, u)) {
      u64 *au = adj + (size_t)u * W;
      for (int w = 0; w < W; w++) au[w] |= av[w];
      CLR(au, best); CLR(au, u);
    }
    memset(av, 0, (size_t)W * sizeof(u64));
  }
  free(alive);
  /* symbolic LU of the permuted matrix, rows as bitsets */
  u64 *R = adj; memset(R, 0, (size_t)n * W * sizeof(u64));
  for (int j = 0; j < n; j++)
    for (int k = S->IAN[j] - 1; k < S->IAN[j + 1] - 1; k++)
      SET(R + (size_t)S->iperm[S->JAN[k] - 1] * W, S->iperm[j]);
  for (int k = 0; k < n; k++) {
    const u64 *rk = R + (size_t)k * W;
    for (int i = k + 1; i < n; i++) {
      u64 *ri = R + (size_t)i * W;
      if (!BIT(ri, k)) continue;
      for (int w = (k + 1) >> 6; w < W; w++) {
        u64 m = rk[w];
        if (w == ((k + 1) >> 6)) m &= ~0ULL << ((k + 1) & 63);
        ri[w] |= m;
      }
    }
  }
  S->Lp = calloc((size_t)n + 1, sizeof(int)); S->Up = calloc((size_t)n + 1, sizeof(int));
  for (int i = 0; i < n; i++) {
    const u64 *ri = R + (size_t)i * W; int l = 0, u = 0;
    for (int j = 0; j < n; j++) if (BIT(ri, j)) { if (j < i) l++; else if (j > i) u++; }
    S->Lp[i + 1] = S->Lp[i] + l; S->Up[i + 1] = S->Up[i] + u;
  }
  S->nzl = S->Lp[n]; S->nzu = S->Up[n];
  S->Lj = malloc((size_t)(S->nzl + 1) * sizeof(int)); S->Uj = malloc((size_t)(S->nzu + 1) * sizeof(int));
  for (int i = 0; i < n; i++) {
    const u64 *ri = R + (size_t)i * W; int l = S->Lp[i], u = S->Up[i];
    for (int j = 0; j < n; j++) if (BIT(ri, j)) { if (j < i) S->Lj[l++] = j; else if (j > i) S->Uj[u++] = j; }
  }
  free(adj);
  /* row-wise view of A in permuted space */
  S->Arp = calloc((size_t)n + 1, sizeof(int)); S->Ak = malloc((size_t)S->nnz * sizeof(int)); S->Ac = malloc((size_t)S->nnz * sizeof(int));
  for (int j = 0; j < n; j++) for (int k = S->IAN[j] - 1; k < S->IAN[j + 1] - 1; k++) S->Arp[S->iperm[S->JAN[k] - 1] + 1]++;
  for (int i = 0; i < n; i++) S->Arp[i + 1] += S->Arp[i];
  int *fill = malloc((size_t)n * sizeof(int)); memcpy(fill, S->Arp, (size_t)n * sizeof(int));
  for (int j = 0; j < n; j++) for (int k = S->IAN[j] - 1; k < S->IAN[j + 1] - 1; k++) {
    int i = S->iperm[S->JAN[k] - 1]; S->Ak[fill[i]] = k; S->Ac[fill[i]] = S->iperm[j]; fill[i]++;
  }
  free(fill);
  return S;
}

void orc_symbolic_free(orc_symbolic *S) {
  if (!S) return;
  free(S->IAN); free(S->JAN); free(S->perm); free(S->iperm); free(S->Lp); free(S->Lj); free(S->Up); free(S->Uj);
  free(S->Arp); free(S->Ak); free(S->Ac); free(S);
}

/* numeric factorisation, row by row (up-looking); returns 0 or 1+row on an exactly zero pivot */
int orc_numeric_lu(const orc_symbolic *S, const double *A, double *L, double *U, double *Dinv, double *w) {
  const int n = S->n;
  for (int i = 0; i < n; i++) {
    for (int q = S->Lp[i]; q < S->Lp[i + 1]; q++) w[S->Lj[q]] = 0.0;
    for (int q = S->Up[i]; q < S->Up[i + 1]; q++) w[S->Uj[q]] = 0.0;
    w[i] = 0.0;
    for (int q = S->Arp[i]; q < S->Arp[i + 1]; q++) w[S->Ac[q]] = A[S->Ak[q]];
    for (int q = S->Lp[i]; q < S->Lp[i + 1]; q++) {
      int k = S->Lj[q]; double lik = w[k] * Dinv[k]; L[q] = lik;
      for (int r = S->Up[k]; r < S->Up[k + 1]; r++) w[S->Uj[r]] -= lik * U[r];
    }
    if (w[i] == 0.0) return i + 1;
    Dinv[i] = 1.0 / w[i];
    for (int q = S->Up[i]; q < S->Up[i + 1]; q++) U[q] = w[S->Uj[q]];
  }
  return 0;
}

void orc_lu_solve(const orc_symbolic *S, const double *L, const double *U, const double *Dinv, double *x, double *z) {
  const int n = S->n;
  for (int i = 0; i < n; i++) {
    double s = x[S->perm[i]];
    for (int q = S->Lp[i]; q < S->Lp[i + 1]; q++) s -= L[q] * z[S->Lj[q]];
    z[i] = s;
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = z[i];
    for (int q = S->Up[i]; q < S->Up[i + 1]; q++) s -= U[q] * z[S->Uj[q]];
    z[i] = s * Dinv[i];
  }
  for (int i = 0; i < n; i++) x[S->perm[i]] = z[i];
}
