// developer aid: statistics of the symbolic LU layout (build: see comment at the end)
#include "../rac-2d_amd/csrc/network.hpp"
#include <cstdio>
#include <algorithm>
#include <map>
using namespace racgpu;
int main(int argc, char **argv) {
  HostNetwork net;
  parse_network(argv[1], net);
  const Symbolic &S = net.sym;
  const int n = net.nS, ns = S.ns;
  printf("nS %d nR %d nnzJ %zu ns %d nt %d nzl %d nzu %d entries L %d U %d nlevL %d nlevU %d\n", n, net.nR, net.Jrow.size(), ns, n - ns, S.nzl, S.nzu,
         S.nzl_entries, S.nzu_entries, S.nlevL, S.nlevU);
  // pivot steps: for each column j, its U rows k < ns
  long piv_sparse = 0, piv_trail = 0, madd_sparse = 0, madd_trail = 0, madd_trail_tailrows = 0, madd_sparse_tailrows = 0;
  int cols_with_piv = 0;
  std::map<int,int> lenhist;
  for (int j = 0; j < n; ++j) {
    int np = S.Ucolend[j] - S.Ucolptr[j];
    if (j < ns && np > 0) cols_with_piv++;
    for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) {
      int k = S.Urow[q];
      int len = S.Lcolend[k] - S.Lcolptr[k];
      int tail = 0;
      for (int t = S.Lcolptr[k]; t < S.Lcolend[k]; ++t) if (S.Lrow[t] >= ns) tail++;
      if (j < ns) { piv_sparse++; madd_sparse += len; madd_sparse_tailrows += tail; } else { piv_trail++; madd_trail += len; madd_trail_tailrows += tail; }
      lenhist[(len + 7) / 8 * 8]++;
    }
  }
  printf("pivot steps: sparse cols %ld (cols with pivots %d of %d), trailing cols %ld; madds sparse %ld (tail rows %ld) trailing %ld (tail rows %ld)\n", piv_sparse, cols_with_piv, ns, piv_trail, madd_sparse, madd_sparse_tailrows,
         madd_trail, madd_trail_tailrows);
  printf("L col len hist (<=len: count of pivot steps):"); for (auto &e : lenhist) printf(" %d:%d", e.first, e.second); printf("\n");
  // L11 (rows<ns) vs L21 (rows>=ns) entries for k<ns
  long l11 = 0, l21 = 0, u11 = 0, u12 = 0;
  for (int k = 0; k < ns; ++k) for (int t = S.Lcolptr[k]; t < S.Lcolend[k]; ++t) (S.Lrow[t] >= ns ? l21 : l11)++;
  for (int j = 0; j < n; ++j) for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) (j >= ns ? u12 : u11)++;
  printf("L11 %ld L21 %ld U11 %ld U12 %ld dense %d\n", l11, l21, u11, u12, (n - ns) * (n - ns));
  // number of distinct pivots k used by trailing columns; per k: number of trailing columns using it
  std::vector<int> use(ns, 0);
  for (int j = ns; j < n; ++j) for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) use[S.Urow[q]]++;
  int nuse = 0; long full = 0; for (int k = 0; k < ns; ++k) if (use[k]) { nuse++; if (use[k] >= (n - ns) * 9 / 10) full++; }
  printf("pivots k<ns used by trailing columns: %d, of which used by >=90%% of them: %ld\n", nuse, full);
  // per trailing column: number of pivots and number of levels
  int minp = 1 << 30, maxp = 0; long totlev = 0;
  for (int j = ns; j < n; ++j) { int np = S.Ucolend[j] - S.Ucolptr[j]; minp = std::min(minp, np); maxp = std::max(maxp, np); int lv = 0; for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) lv += S.Ugrp[q]; totlev += lv; }
  printf("trailing cols: pivots per col min %d max %d, mean levels per col %.1f\n", minp, maxp, (double)totlev / (n - ns));
  long totlev_s = 0; for (int j = 0; j < ns; ++j) for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) totlev_s += S.Ugrp[q];
  printf("sparse cols: total levels %ld\n", totlev_s);
  // rows of L21 nnz per tail row
  std::vector<int> rown(n, 0); for (int k = 0; k < ns; ++k) for (int t = S.Lcolptr[k]; t < S.Lcolend[k]; ++t) rown[S.Lrow[t]]++;
  printf("L21 nnz per tail row:"); for (int i = ns; i < n; ++i) printf(" %d", rown[i]); printf("\n");
  {
    std::vector<int> l11len(ns, 0), l21len(ns, 0);
    for (int k = 0; k < ns; ++k) for (int t = S.Lcolptr[k]; t < S.Lcolend[k]; ++t) (S.Lrow[t] >= ns ? l21len[k] : l11len[k])++;
    long pairs11 = 0, pairs = 0, pairs21 = 0;
    for (int j = ns; j < n; ++j) for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) { pairs++; if (l11len[S.Urow[q]]) pairs11++; if (l21len[S.Urow[q]]) pairs21++; }
    printf("trailing (j,k) pairs %ld, with nonempty L11 column %ld, with nonempty L21 column %ld\n", pairs, pairs11, pairs21);
    for (int G : {2, 3, 4, 8, 16}) {
      long tot = 0, totmadd = 0; int ngroups = 0;
      for (int j0 = ns; j0 < n; j0 += G) {
        std::vector<char> in(ns, 0);
        for (int j = j0; j < std::min(n, j0 + G); ++j) for (int q = S.Ucolptr[j]; q < S.Ucolend[j]; ++q) in[S.Urow[q]] = 1;
        for (int k = 0; k < ns; ++k) if (in[k]) { tot++; totmadd += l21len[k]; }
        ngroups++;
      }
      printf("G=%d: groups %d, sum of union sizes %ld (x G = %ld slots vs %ld real pairs), L21 entries loaded %ld\n", G, ngroups, tot, tot * G, pairs, totmadd);
    }
  }
  return 0;
}
// hipcc -O2 -std=c++17 tools/dev/dev_lustruct.cpp rac-2d_amd/csrc/network.o -o build/dev_lustruct
