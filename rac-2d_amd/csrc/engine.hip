// engine.hip -- kernels, device-table upload, workspace management and the extern "C" ABI (include/racgpu.h).
// gfx950 only.  The product path has no CPU fallback: every compute entry point fails loudly without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/racgpu.h"
#include "engine_integrate.hpp"
#include "network.hpp"

using namespace racgpu;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return -1; }
#define HIP_OK(x)                                                                                       \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_));     \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// kernels (one wave per workgroup, one cell per wave)
// ---------------------------------------------------------------------------------------------------------
struct LdsViews { double *y, *savf, *acor, *ewt, *wx; };
__device__ __forceinline__ LdsViews carve(double *lds, int nlds) { return {lds, lds + nlds, lds + 2 * nlds, lds + 3 * nlds, lds + 4 * nlds}; }

__global__ __launch_bounds__(64) void k_rates(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, const double *cells, double *rates_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  const int cell = blockIdx.x, lane = threadIdx.x;
  dev_rates(N, P, cells + (size_t)cell * RACGPU_NPAR, rates_out + (size_t)cell * N.nR, lane);
}

__global__ __launch_bounds__(64) void k_rhs(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, const double *cells, const double *yin, double *rates_ws, double *ydot_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, nlds = (N.nS + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  const double *cp = cells + (size_t)cell * RACGPU_NPAR;
  double *rates = rates_ws + (size_t)cell * N.nR;
  dev_rates(N, P, cp, rates, lane);
  for (int i = lane; i < N.nS; i += 64) v.y[i] = yin[(size_t)cell * N.nS + i];
  wave_sync();
  dev_rhs(N, rates, cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES], gptr(N.r_C), v.y, v.savf, lane);
  for (int i = lane; i < N.nS; i += 64) ydot_out[(size_t)cell * N.nS + i] = v.savf[i];
}

__global__ __launch_bounds__(64) void k_jac(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, const double *cells, const double *yin, double *rates_ws, double *vals_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, nlds = (N.nS + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  const double *cp = cells + (size_t)cell * RACGPU_NPAR;
  double *rates = rates_ws + (size_t)cell * N.nR;
  dev_rates(N, P, cp, rates, lane);
  for (int i = lane; i < N.nS; i += 64) v.y[i] = yin[(size_t)cell * N.nS + i];
  wave_sync();
  dev_build_P<false>(N, rates, cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES], v.y, 1.0, false, vals_out + (size_t)cell * N.nnzJ, lane);
}

__global__ __launch_bounds__(64) void k_newton(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, const double *cells, const double *yin, double gamma, double *bx,
                                               int repeat, long long *cyc_out) {
  const DevNet &N = *Np; const DevParams &P = *Pp;
  extern __shared__ double lds[];
  const int cell = blockIdx.x, lane = threadIdx.x, nlds = (N.nS + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  const double *cp = cells + (size_t)cell * RACGPU_NPAR;
  double *rates = W.rates + (size_t)cell * N.nR, *Pv = W.P + (size_t)cell * N.nnzJ, *Lv = W.L + (size_t)cell * N.nzl,
         *Uv = W.U + (size_t)cell * N.nzu, *Dinv = W.Dinv + (size_t)cell * N.npad;
  dev_rates(N, P, cp, rates, lane);
  for (int i = lane; i < N.nS; i += 64) v.y[i] = yin[(size_t)cell * N.nS + i];
  wave_sync();
  dev_build_P<true>(N, rates, cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES], v.y, -gamma, true, Pv, lane);
  // repeat > 1 (developer aid, RACGPU_DEBUG_REPEAT): the same factorisation and solve again and again, with the
  // cycle counts of the LU parts and of the solve written out per cell
  long long cyc[4] = {0, 0, 0, 0}, c_lu = 0, c_solve = 0;
  for (int r = 0; r < repeat; ++r) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    dev_lu(N, Pv, Lv, Uv, Dinv, v.wx, v.y, lane, cyc);
    const long long t1 = (long long)__builtin_readcyclecounter();
    for (int i = lane; i < N.nS; i += 64) v.savf[i] = bx[(size_t)cell * N.nS + i];
    dev_solve(N, Lv, Uv, Dinv, v.savf, v.wx, lane);
    c_lu += t1 - t0; c_solve += (long long)__builtin_readcyclecounter() - t1;
  }
  for (int i = lane; i < N.nS; i += 64) bx[(size_t)cell * N.nS + i] = v.savf[i];
  if (cyc_out && lane == 0) {
    long long *o = cyc_out + (size_t)cell * 8;
    o[0] = c_lu; o[1] = c_solve; o[2] = cyc[0]; o[3] = cyc[1]; o[4] = cyc[2]; o[5] = cyc[3];
  }
}

// The hot path.  Persistent: each wave pulls cells from a queue until it is empty; its workspace is per wave
// (slot), not per cell, so the HBM footprint is nslots * ~0.4 MB whatever the batch size.
__global__ __launch_bounds__(64) void k_solve(const DevNet *__restrict__ Np, const DevParams *__restrict__ Pp, DevWork W, int ncell, const double *__restrict__ cells,
                                              double *__restrict__ yio, double *__restrict__ t_final, int *__restrict__ quality,
                                              long long *__restrict__ stats, double *__restrict__ record, double *__restrict__ touts,
                                              const int *__restrict__ order) {
  extern __shared__ double lds[];
  const DevNet &N = *Np; const DevParams &P = *Pp;
  const int lane = threadIdx.x, slot = blockIdx.x, n = N.nS, nlds = (n + 1) & ~1;
  LdsViews v = carve(lds, nlds);
  CellCtx c;
  c.y = v.y; c.savf = v.savf; c.acor = v.acor; c.ewt = v.ewt; c.wx = v.wx;
  c.rates = nullptr; c.yh = W.yh + (size_t)slot * 6 * N.npad; c.Pv = W.P + (size_t)slot * N.nnzJ;
  c.Lv = W.L + (size_t)slot * N.nzl; c.Uv = W.U + (size_t)slot * N.nzu; c.Dinv = W.Dinv + (size_t)slot * N.npad;
  c.rtol = W.rtol + (size_t)slot * N.npad; c.atol = W.atol + (size_t)slot * N.npad;
  c.lane = lane; c.n = n; c.npad = N.npad; c.inv_neq = 1.0 / (double)(n + 1);
  c.marker = slot == 0 ? W.marker : nullptr;
  for (;;) {
    int cell = 0;
    if (lane == 0) cell = atomicAdd(W.counter, 1);
    cell = uniform_i(cell);
    dev_mark(c, 10 + cell);
    if (cell >= ncell) break;
    if (order) cell = order[cell]; // longest-expected-first schedule (racgpu_set_cost_hints)
    const double *cp = cells + (size_t)cell * RACGPU_NPAR;
    const long long cyc0 = dev_clock();
    c.cyc_rhs = c.cyc_jac = c.cyc_lu = c.cyc_solve = 0;
    c.cyc_lu_part[0] = c.cyc_lu_part[1] = c.cyc_lu_part[2] = 0; c.cyc_lu_part[3] = 0;
    c.Tgas = cp[RACGPU_P_TGAS]; c.nsite = cp[RACGPU_P_D2H] * cp[RACGPU_P_SITES];
    const double t_max = cp[RACGPU_P_TMAX] > 0.0 ? cp[RACGPU_P_TMAX] : P.t_max;
    const int n_record = (int)ceil(log((t_max - 0.0) / P.dt_first_step * (P.ratio_tstep - 1.0) + 1.0) / log(P.ratio_tstep)) + 1;
    dev_mark(c, 1);
    dev_tolerances(N, P, cp[RACGPU_P_D2H], c.rtol, c.atol, c.rT, c.aT, lane);
    dev_mark(c, 2);
    c.rates = W.rates + (size_t)cell * N.nR; // filled by k_rates for the whole batch just before this launch
    dev_mark(c, 3);
    for (int i = lane; i < n; i += 64) c.y[i] = yio[(size_t)cell * n + i];
    wave_sync();
    double *rec = record ? record + (size_t)cell * P.n_record * (n + 1) : nullptr;
    double *tos = touts ? touts + (size_t)cell * P.n_record : nullptr;
    const int nrec = record || touts ? min(n_record, P.n_record) : n_record;
    CellResult R = dev_evol_solve(N, P, c, t_max, nrec, rec, tos, cell == 0 ? W.trace : nullptr);
    dev_mark(c, 4);
    wave_sync();
    for (int i = lane; i < n; i += 64) yio[(size_t)cell * n + i] = c.y[i];
    if (lane == 0) {
      if (t_final) t_final[cell] = R.t_final;
      if (quality) quality[cell] = R.quality;
      if (stats) {
        long long *s = stats + (size_t)cell * RACGPU_NSTAT;
        s[RACGPU_S_NST] = R.nst; s[RACGPU_S_NFE] = R.nfe; s[RACGPU_S_NJE] = R.nje; s[RACGPU_S_NLU] = R.nlu;
        s[RACGPU_S_NERR] = R.nerr; s[RACGPU_S_NREC_REAL] = R.nrec_real; s[RACGPU_S_QSUM] = R.qsum; s[RACGPU_S_NCFAIL_ETFAIL] = R.nfail;
        s[RACGPU_S_CYC_TOTAL] = dev_clock() - cyc0; s[RACGPU_S_CYC_RHS] = c.cyc_rhs; s[RACGPU_S_CYC_JAC] = c.cyc_jac;
        s[RACGPU_S_CYC_LU] = c.cyc_lu; s[RACGPU_S_CYC_SOLVE] = c.cyc_solve;
        s[13] = c.cyc_lu_part[0]; s[14] = c.cyc_lu_part[1]; s[15] = c.cyc_lu_part[2]; // finish = LU - the three
      }
    }
    dev_mark(c, 6);
  }
  dev_mark(c, 7);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct racgpu_network {
  HostNetwork net;
  bool uploaded = false;
  DevNet dn{};
  DevNet *dn_dev = nullptr;      // device copy of dn (kernels read the tables through a pointer, not kernargs)
  DevParams *dp_dev = nullptr;   // device copy of the last parameter set
  std::vector<void *> dev_allocs;
  hipStream_t stream = nullptr;
  // workspace
  DevWork ws{};
  std::vector<void *> ws_allocs;
  long ws_slots = 0, ws_rate_cells = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<double> cost_hints; // racgpu_set_cost_hints
  int *order_dev = nullptr; long order_cap = 0;
  bool timed = false;
  int cu_count = 0;

  template <typename T>
  const T *up(const std::vector<T> &h) {
    void *d = nullptr;
    HIP_OK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    dev_allocs.push_back(d);
    return (const T *)d;
  }
  void upload();
  void ensure_workspace(long slots, long rate_cells);
  void free_ws() { for (void *p : ws_allocs) (void)hipFree(p); ws_allocs.clear(); ws_slots = 0; ws_rate_cells = 0; }
  ~racgpu_network() {
    free_ws();
    if (order_dev) (void)hipFree(order_dev);
    for (void *p : dev_allocs) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

static void require_gpu() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    throw std::runtime_error("no HIP device visible: the racgpu compute path has no CPU fallback");
}

void racgpu_network::upload() {
  if (uploaded) return;
  require_gpu();
  hipDeviceProp_t prop;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(hipGetDeviceProperties(&prop, dev));
  cu_count = prop.multiProcessorCount;
  const HostNetwork &h = net;
  const int nS = h.nS, nR = h.nR;
  dn.nS = nS; dn.nR = nR; dn.npad = (nS + 63) / 64 * 64;
  dn.nnzJ = (int)h.Jrow.size(); dn.nzl = h.sym.nzl; dn.nzu = h.sym.nzu; dn.ns = h.sym.ns;
  std::vector<int16_t> itype(nR);
  std::vector<uint16_t> re0(nR), re1(nR), id3(nR, 0);
  std::vector<uint8_t> nreac(nR), fss(nR), flags(nR, 0);
  std::vector<double> A(nR), B(nR), C(nR), T0(nR), T1(nR);
  std::vector<uint64_t> w0(nR), w1(nR), w2(nR);
  const int iH2 = h.idx10[0], igH = h.i_gH;
  for (int r = 0; r < nR; ++r) {
    const Reaction &x = h.R[r];
    itype[r] = (int16_t)x.itype;
    re0[r] = x.reac[0] > 0 ? (uint16_t)(x.reac[0] - 1) : 0xffff;
    re1[r] = x.reac[1] > 0 ? (uint16_t)(x.reac[1] - 1) : 0xffff;
    nreac[r] = (uint8_t)x.n_reac;
    fss[r] = (uint8_t)h.fss_selector(r);
    if (x.rname[0] == "H2" && iH2 > 0) flags[r] |= 1;
    if (x.rname[0] == "gH" && igH > 0) flags[r] |= 2;
    if (x.itype == 21) {
      const int g1 = h.elements[x.reac[0] - 1][2];
      id3[r] = (uint16_t)((g1 == 0 ? x.reac[0] : x.reac[1]) - 1);
      if (h.elements[x.reac[0] - 1][0] * h.elements[x.reac[1] - 1][0] == -1) flags[r] |= 4;
    }
    A[r] = x.ABC[0]; B[r] = x.ABC[1]; C[r] = x.ABC[2]; T0[r] = x.Trange[0]; T1[r] = x.Trange[1];
    const Kind k = h.kind(r);
    const uint64_t a = k == K_NONE ? 0 : (uint64_t)(x.reac[0] - 1), b = (k == K_TWO) ? (uint64_t)(x.reac[1] - 1) : a;
    w0[r] = (uint64_t)k | ((uint64_t)x.n_reac << 8) | (a << 16) | (b << 32);
    uint16_t tg[8];
    for (int s = 0; s < 8; ++s) tg[s] = 0xffff;
    if (k != K_NONE) {
      // slots 0..n_reac-1 are reactants (subtract), then products (add); unused slots stay 0xffff.
      for (int s = 0; s < x.n_reac; ++s) tg[s] = (uint16_t)(x.reac[s] - 1);
      for (int s = 0; s < x.n_prod; ++s) tg[x.n_reac + s] = (uint16_t)(x.prod[s] - 1);
    }
    w1[r] = (uint64_t)tg[0] | ((uint64_t)tg[1] << 16) | ((uint64_t)tg[2] << 32) | ((uint64_t)tg[3] << 48);
    w2[r] = (uint64_t)tg[4] | ((uint64_t)tg[5] << 16) | ((uint64_t)tg[6] << 32) | ((uint64_t)0xffff << 48);
  }
  dn.r_itype = up(itype); dn.r_re0 = up(re0); dn.r_re1 = up(re1); dn.r_nreac = up(nreac); dn.r_fss = up(fss);
  dn.r_flags = up(flags); dn.r_id3 = up(id3);
  dn.r_A = up(A); dn.r_B = up(B); dn.r_C = up(C); dn.r_T0 = up(T0); dn.r_T1 = up(T1);
  dn.s_mass = up(h.mass_num); dn.s_vib = up(h.vib_freq); dn.s_Edes = up(h.Edesorb);
  {
    std::vector<int> dl(h.dupli_list);
    for (int &v : dl) v -= 1;
    dn.dupli_ptr = up(h.dupli_ptr); dn.dupli_list = up(dl);
  }
  w0.resize(w0.size() + 448, 0); w1.resize(w1.size() + 448, ~0ull); w2.resize(w2.size() + 448, ~0ull); // padding: the RHS runs to a multiple of 192 rows and prefetches 128 ahead (kind 0, no targets)
  dn.rhs_w0 = up(w0); dn.rhs_w1 = up(w1); dn.rhs_w2 = up(w2);
  // Jacobian gather: entries sorted by decreasing term count so that the 64 lanes of a pass do similar work
  {
    const int nnz = dn.nnzJ;
    std::vector<int> order(nnz);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return h.term_ptr[a + 1] - h.term_ptr[a] > h.term_ptr[b + 1] - h.term_ptr[b];
    });
    order.resize((size_t)(nnz + 63) / 64 * 64, -1);
    dn.jac_slots = (int)order.size();
    dn.jac_order = up(order);
    dn.term_ptr = up(h.term_ptr);
    std::vector<uint64_t> tw(h.terms.size());
    for (size_t t = 0; t < tw.size(); ++t) {
      const JacTerm &x = h.terms[t];
      tw[t] = (uint64_t)x.rxn | ((uint64_t)x.other << 16) | ((uint64_t)x.kind << 32) | ((uint64_t)x.flags << 40) | ((uint64_t)x.other2 << 48);
    }
    dn.terms = up(tw);
    std::vector<uint8_t> isd(nnz, 0);
    for (int j = 0; j < nS; ++j)
      for (int q = h.Jcolptr[j]; q < h.Jcolptr[j + 1]; ++q) if (h.Jrow[q] == j) isd[q] = 1;
    dn.jac_isdiag = up(isd);
    // Term stream for the Jacobian gather: pass p handles the 64 entries order[64p .. 64p+63], one per lane; row i of
    // the pass holds term i of every lane's entry (null where an entry has fewer terms), so the kernel reads the terms
    // as one linear, coalesced, prefetchable stream.  rowflag marks the last row of every pass; slot words say where
    // each lane's sum goes: entry | position in the permuted P storage << 24 | diagonal << 48 | valid << 49.
    {
      const int npass = (int)order.size() / 64;
      std::vector<uint64_t> stream, slot((size_t)(npass + 1) * 64, 0ull);
      std::vector<uint32_t> rowflag;
      for (int p = 0; p < npass; ++p) {
        int niter = 1;
        for (int l = 0; l < 64; ++l) {
          const int e = order[(size_t)p * 64 + l];
          if (e < 0) continue;
          niter = std::max(niter, h.term_ptr[e + 1] - h.term_ptr[e]);
          slot[(size_t)p * 64 + l] = (uint64_t)e | ((uint64_t)h.sym.Ppos[e] << 24) | ((uint64_t)(isd[e] ? 1 : 0) << 48) | (1ull << 49);
        }
        for (int i = 0; i < niter; ++i) {
          for (int l = 0; l < 64; ++l) {
            const int e = order[(size_t)p * 64 + l];
            const bool has = e >= 0 && i < h.term_ptr[e + 1] - h.term_ptr[e];
            stream.push_back(has ? tw[h.term_ptr[e] + i] : ~0ull);
          }
          rowflag.push_back(i == niter - 1 ? 1u : 0u);
        }
      }
      while (rowflag.size() % kJacUnroll) { rowflag.push_back(0u); stream.insert(stream.end(), 64, ~0ull); }
      dn.jac_rows = (int)rowflag.size();
      stream.insert(stream.end(), (size_t)64 * (kJacUnroll + 8), ~0ull); // the kernel prefetches rows past the end
      rowflag.resize((rowflag.size() + 63) / 64 * 64 + 64, 0u);
      dn.jac_stream = up(stream); dn.jac_rowflag = up(rowflag); dn.jac_slot = up(slot);
    }
  }
  {
    const Symbolic &S = h.sym;
    std::vector<uint16_t> perm(S.perm.begin(), S.perm.end()), Lrow(S.Lrow.begin(), S.Lrow.end()), Urow(S.Urow.begin(), S.Urow.end()),
        Prow(S.Prow.begin(), S.Prow.end());
    Lrow.resize(Lrow.size() + 64, 0); Urow.resize(Urow.size() + 64, 0); // the LU prefetch reads up to 64 entries past a column
    Prow.resize(Prow.size() + 64, 0);
    dn.perm = up(perm); dn.Lrow = up(Lrow); dn.Urow = up(Urow); dn.Prow = up(Prow);
    dn.Lcolptr = up(S.Lcolptr); dn.Lcolend = up(S.Lcolend); dn.Ucolptr = up(S.Ucolptr); dn.Ucolend = up(S.Ucolend); dn.Udptr = up(S.Udptr);
    dn.Pcolptr = up(S.Pcolptr); dn.Ppos = up(S.Ppos);
    {
      std::vector<uint8_t> pd(S.Psrc.size(), 0);
      for (int j = 0; j < nS; ++j)
        for (int q = h.Jcolptr[j]; q < h.Jcolptr[j + 1]; ++q) if (h.Jrow[q] == j) pd[S.Ppos[q]] = 1;
      dn.Pdiag = up(pd);
    }
    auto pack = [](const std::vector<int> &row, const std::vector<int> &col, const std::vector<int> &lev, size_t nstream, int &nchunk) {
      // the streamed part only (the dense trailing block is solved in registers); the storage is level-aligned, so
      // every chunk of 64 entries has ONE level; bit 20 of every word of a chunk: the next chunk continues this level
      std::vector<uint32_t> rc(nstream);
      for (size_t e = 0; e < nstream; ++e) rc[e] = (uint32_t)row[e] | ((uint32_t)col[e] << 10);
      rc.resize((nstream + 63) / 64 * 64, 0u); // row == col == 0: skipped
      size_t nch = rc.size() / 64;
      auto chunk_level = [&](size_t c) { return lev[std::min(c * 64, nstream - 1)]; };
      for (size_t c = 0; c < nch; ++c) {
        for (size_t e = c * 64; e < std::min((c + 1) * 64, nstream); ++e)
          if (lev[e] != chunk_level(c)) throw std::runtime_error("solve schedule: a chunk spans two levels");
        if (c + 1 < nch && chunk_level(c + 1) == chunk_level(c))
          for (size_t e = c * 64; e < (c + 1) * 64; ++e) rc[e] |= 1u << 20;
      }
      while (nch % kSweepDepth) { rc.resize(rc.size() + 64, 0u); ++nch; } // null chunks: the sweep is unrolled by its depth
      nchunk = (int)nch;
      rc.resize(rc.size() + 16 * 64, 0u); // spare chunks: the sweep prefetches unconditionally
      return rc;
    };
    {
      // Work list of the LU: the columns j < ns that have pivots (U(:,j) not empty), ascending, then the trailing
      // columns ns..n-1.  Columns j < ns WITHOUT pivots ("leaves": nothing is ever subtracted from them) are
      // factored beforehand in one elementwise pass: D^-1_j = 1/P(j,j), L(:,j) = P(rows > j, j) * D^-1_j, their L
      // pattern being exactly P's (leaf_diag: position of P(j,j) | j<<32; leaf_ent: position in P | position in L
      // << 20 | j << 40).
      std::vector<unsigned long long> ud, leaf_diag, leaf_ent;
      std::vector<LuCol> lc;
      int nwork_sparse = 0;
      for (int j = 0; j < nS; ++j) {
        if (j < S.ns && S.Ucolend[j] == S.Ucolptr[j]) {
          bool pattern_ok = (S.Lcolend[j] - S.Lcolptr[j]) == 0;
          int nbelow = 0;
          for (int q = S.Pcolptr[j]; q < S.Pcolptr[j + 1]; ++q) if (S.Prow[q] > j) ++nbelow;
          pattern_ok = nbelow == S.Lcolend[j] - S.Lcolptr[j];
          bool has_diag = false, above = false;
          for (int q = S.Pcolptr[j]; q < S.Pcolptr[j + 1]; ++q) { if (S.Prow[q] == j) has_diag = true; if (S.Prow[q] < j) above = true; }
          if (pattern_ok && has_diag && !above) {
            for (int q = S.Pcolptr[j]; q < S.Pcolptr[j + 1]; ++q) {
              const int r = S.Prow[q];
              if (r == j) leaf_diag.push_back((unsigned long long)q | ((unsigned long long)j << 32));
              else {
                int lp = -1;
                for (int t = S.Lcolptr[j]; t < S.Lcolend[j]; ++t) if (S.Lrow[t] == r) lp = t;
                if (lp < 0) throw std::runtime_error("LU layout: leaf column pattern mismatch");
                leaf_ent.push_back((unsigned long long)q | ((unsigned long long)lp << 20) | ((unsigned long long)j << 40));
              }
            }
            continue;
          }
        }
        LuCol c{};
        c.u0 = S.Ucolptr[j]; c.u1 = S.Ucolend[j]; c.lc0 = S.Lcolptr[j]; c.lc1 = S.Lcolend[j]; c.p0 = S.Pcolptr[j]; c.p1 = S.Pcolptr[j + 1];
        c.ur = S.Ucolend[j]; // rows < ns only: the pivots ns..j-1 of a trailing column are applied in registers
        c.d0 = (int)ud.size(); c.j = j;
        for (int q = S.Ucolptr[j]; q < c.ur; ++q) {
          const int k = S.Urow[q];
          const int len = S.Lcolend[k] - S.Lcolptr[k];
          for (int off = 0; off < std::max(len, 1); off += 64) // L columns longer than 64 rows: one descriptor per 64 rows
            ud.push_back((unsigned long long)k | ((unsigned long long)std::min(64, len - off) << 16) |
                         ((unsigned long long)((S.Ugrp[q] && off == 0) ? 1 : 0) << 30) | ((unsigned long long)(S.Lcolptr[k] + off) << 32));
        }
        while (ud.size() % kLuDepth) ud.push_back(0ull);
        c.d1 = (int)ud.size();
        lc.push_back(c);
        if (j < S.ns) ++nwork_sparse;
      }
      ud.resize(ud.size() + 64, 0ull);
      dn.Udesc = up(ud);
      dn.nwork_sparse = nwork_sparse; dn.nwork = (int)lc.size();
      dn.nleaf = (int)leaf_diag.size(); dn.nleaf_ent = (int)leaf_ent.size();
      leaf_diag.resize(leaf_diag.size() + 64, 0ull); leaf_ent.resize(leaf_ent.size() + 64, 0ull);
      dn.leaf_diag = up(leaf_diag); dn.leaf_ent = up(leaf_ent);
      lc.push_back(lc.empty() ? LuCol{} : lc.back()); lc.push_back(lc.back()); // the column prefetch reads two ahead
      dn.lucol = up(lc);
    }
    dn.nzl_stream = S.nzl_stream; dn.nzu_stream = S.nzu_stream;
    dn.Lrc = up(pack(S.Lrow, S.Lcol, S.Llev, (size_t)S.nzl_stream, dn.nchunkL));
    dn.Urc = up(pack(S.Urow, S.Ucol, S.Ulev, (size_t)S.nzu_stream, dn.nchunkU));
  }
  dn.i_H = h.idx10[1] - 1; dn.i_E = h.idx10[2] - 1; dn.i_gH = h.i_gH - 1; dn.i_gH2 = h.i_gH2 - 1; dn.i_gH2O = h.i_gH2O - 1;
  dn.i_Grain0 = h.i_Grain0 - 1; dn.i_GrainM = h.i_GrainM - 1; dn.i_GrainP = h.i_GrainP - 1;
  {
    std::vector<uint8_t> cls(nS, 0);
    for (int k = 0; k < 10; ++k) if (h.idx10[k] > 0) cls[h.idx10[k] - 1] = 1;
    if (h.i_Grain0 > 0) for (int g : {h.i_Grain0, h.i_GrainM, h.i_GrainP}) if (g > 0) cls[g - 1] = 2;
    for (int g : h.grain) cls[g - 1] = 3;
    dn.s_tolclass = up(cls);
  }
  HIP_OK(hipEventCreate(&ev0));
  HIP_OK(hipEventCreate(&ev1));
  { void *d = nullptr; HIP_OK(hipMalloc(&d, sizeof(DevNet))); HIP_OK(hipMemcpy(d, &dn, sizeof(DevNet), hipMemcpyHostToDevice)); dev_allocs.push_back(d); dn_dev = (DevNet *)d; }
  { void *d = nullptr; HIP_OK(hipMalloc(&d, sizeof(DevParams))); dev_allocs.push_back(d); dp_dev = (DevParams *)d; }
  uploaded = true;
}

void racgpu_network::ensure_workspace(long slots, long rate_cells) {
  if (slots <= ws_slots && rate_cells <= ws_rate_cells) return;
  slots = std::max(slots, ws_slots); rate_cells = std::max(rate_cells, ws_rate_cells);
  free_ws();
  auto alloc = [&](size_t count) { void *d = nullptr; HIP_OK(hipMalloc(&d, count * sizeof(double))); ws_allocs.push_back(d); return (double *)d; };
  ws.rates = alloc((size_t)rate_cells * dn.nR + 448); // per CELL; +448: the RHS of the last cell reads past nR
  ws.yh = alloc((size_t)slots * 6 * dn.npad);
  ws.P = alloc((size_t)slots * dn.nnzJ + 64); // spare: the LU's column prefetch reads up to 63 entries past a column
  ws.L = alloc((size_t)slots * std::max(dn.nzl, 1) + 2048); // spare: prefetches of the last slot read past nzl
  ws.U = alloc((size_t)slots * std::max(dn.nzu, 1) + 2048);
  ws.Dinv = alloc((size_t)slots * dn.npad);
  ws.rtol = alloc((size_t)slots * dn.npad);
  ws.atol = alloc((size_t)slots * dn.npad);
  void *c = nullptr;
  HIP_OK(hipMalloc(&c, 64));
  ws_allocs.push_back(c);
  ws.counter = (int *)c;
  ws_slots = slots; ws_rate_cells = rate_cells;
}

static void cfode_bdf(DevParams &P) { // BDF method coefficients, orders 1..5 (DCFODE METH=2, reference src/opkda1.f:146-171)
  double pc[8] = {0};
  pc[1] = 1.0;
  double rq1fac = 1.0;
  std::memset(P.elco, 0, sizeof P.elco);
  std::memset(P.tesco, 0, sizeof P.tesco);
  for (int nq = 1; nq <= 5; ++nq) {
    const double fnq = nq;
    pc[nq + 1] = 0.0;
    for (int i = nq + 1; i >= 2; --i) pc[i] = pc[i - 1] + fnq * pc[i];
    pc[1] = fnq * pc[1];
    for (int i = 1; i <= nq + 1; ++i) P.elco[nq][i] = pc[i] / pc[2];
    P.elco[nq][2] = 1.0;
    P.tesco[nq][1] = rq1fac;
    P.tesco[nq][2] = (nq + 1) / P.elco[nq][1];
    P.tesco[nq][3] = (nq + 2) / P.elco[nq][1];
    rq1fac = rq1fac / fnq;
  }
}

static DevParams to_dev(const racgpu_params *p) {
  if (p->H2_form_use_moeq) throw std::runtime_error("H2_form_use_moeq = .true. is not implemented (see DESIGN.md, out of scope rows)");
  if (p->evol_dust_size) throw std::runtime_error("evol_dust_size = .true. is not implemented");
  if (!(p->dt_first_step > 0.0) || !(p->ratio_tstep > 1.0) || !(p->t_max > 0.0)) throw std::runtime_error("need dt_first_step > 0, ratio_tstep > 1, t_max > 0");
  DevParams P{};
  P.RTOL = p->RTOL; P.ATOL = p->ATOL; P.t_max = p->t_max; P.dt_first_step = p->dt_first_step; P.ratio_tstep = p->ratio_tstep;
  P.Diff2DesorRatio = p->Diff2DesorRatio; P.special_gH_E_diff = p->special_gH_E_diff;
  P.mxstep = p->mxstep_per_interval; P.steps_reset = p->steps_reset_solver; P.use_special_gH_mobi = p->use_special_gH_mobi;
  P.tol_j = p->tol_policy_j > 0 ? p->tol_policy_j : 1;
  P.max_steps_per_cell = p->max_steps_per_cell;
  P.max_runtime_allowed = p->max_runtime_allowed;
  P.n_record = racgpu_n_record(p, 0.0, p->t_max);
  if (const char *e = std::getenv("RACGPU_DEBUG_TRACE")) P.debug_max_calls = std::atoi(e);
  cfode_bdf(P);
  return P;
}

template <typename F>
static int guarded(F &&f) {
  try { f(); return 0; }
  catch (const std::exception &e) { return fail(e.what()); }
}

struct DevBuf { // a device buffer that is either the caller's (MEM_DEVICE) or a staged copy (MEM_HOST)
  void *d = nullptr; bool own = false; size_t bytes = 0; void *host = nullptr;
  DevBuf(const void *src, size_t nbytes, int mem, bool copy_in) : bytes(nbytes) {
    if (!src) return;
    if (mem == RACGPU_MEM_DEVICE) { d = const_cast<void *>(src); return; }
    HIP_OK(hipMalloc(&d, std::max<size_t>(nbytes, 8)));
    own = true; host = const_cast<void *>(src);
    if (copy_in) HIP_OK(hipMemcpy(d, src, nbytes, hipMemcpyHostToDevice));
  }
  void copy_out() { if (own && host) HIP_OK(hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost)); }
  ~DevBuf() { if (own) (void)hipFree(d); }
};

extern "C" {

const char *racgpu_last_error(void) { return g_err.c_str(); }

int racgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

racgpu_network *racgpu_network_load(const char *path) {
  try {
    auto h = std::make_unique<racgpu_network>();
    parse_network(path, h->net);
    return h.release();
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

void racgpu_network_destroy(racgpu_network *h) { delete h; }

int racgpu_network_dims(const racgpu_network *h, int32_t *nS, int32_t *nR, int32_t *nnzJ, int32_t *nzl, int32_t *nzu) {
  if (!h) return fail("null network");
  if (nS) *nS = h->net.nS;
  if (nR) *nR = h->net.nR;
  if (nnzJ) *nnzJ = (int32_t)h->net.Jrow.size();
  if (nzl) *nzl = h->net.sym.nzl_entries; // entries of the factors (the storage also holds alignment padding)
  if (nzu) *nzu = h->net.sym.nzu_entries;
  return 0;
}

int racgpu_species_name(const racgpu_network *h, int32_t i, char *buf, int32_t buflen) {
  if (!h || i < 1 || i > h->net.nS || buflen < 1) return fail("bad species index");
  std::snprintf(buf, (size_t)buflen, "%s", h->net.names[i - 1].c_str());
  return 0;
}

int racgpu_species_index(const racgpu_network *h, const char *name) { return h ? h->net.species_index(name) : 0; }

int racgpu_reactions(const racgpu_network *h, int32_t *reac, int32_t *prod, int32_t *n_reac, int32_t *n_prod, int32_t *itype, int32_t *n_dupli) {
  if (!h) return fail("null network");
  for (int r = 0; r < h->net.nR; ++r) {
    const Reaction &x = h->net.R[r];
    if (reac) for (int k = 0; k < 3; ++k) reac[3 * r + k] = x.reac[k];
    if (prod) for (int k = 0; k < 4; ++k) prod[4 * r + k] = x.prod[k];
    if (n_reac) n_reac[r] = x.n_reac;
    if (n_prod) n_prod[r] = x.n_prod;
    if (itype) itype[r] = x.itype;
    if (n_dupli) n_dupli[r] = h->net.dupli_ptr[r + 1] - h->net.dupli_ptr[r];
  }
  return 0;
}

int racgpu_species_attrs(const racgpu_network *h, double *mass, double *vib, double *Ed, int32_t *cp, int32_t *charge) {
  if (!h) return fail("null network");
  for (int i = 0; i < h->net.nS; ++i) {
    if (mass) mass[i] = h->net.mass_num[i];
    if (vib) vib[i] = h->net.vib_freq[i];
    if (Ed) Ed[i] = h->net.Edesorb[i];
    if (cp) cp[i] = h->net.counterpart[i];
    if (charge) charge[i] = h->net.elements[i][0];
  }
  return 0;
}

int racgpu_jac_pattern(const racgpu_network *h, int32_t *colptr, int32_t *rowidx) {
  if (!h) return fail("null network");
  if (colptr) for (int j = 0; j <= h->net.nS; ++j) colptr[j] = h->net.Jcolptr[j] + 1;
  if (rowidx) for (size_t q = 0; q < h->net.Jrow.size(); ++q) rowidx[q] = h->net.Jrow[q] + 1;
  return 0;
}

int racgpu_load_initial_abundances(const racgpu_network *h, const char *path, double *y0) {
  if (!h) return fail("null network");
  return guarded([&] { load_initial_abundances(h->net, path, y0); });
}

void racgpu_params_default(racgpu_params *p) {
  std::memset(p, 0, sizeof *p);
  p->RTOL = 1e-4; p->ATOL = 1e-30; p->t_max = 1e6; p->dt_first_step = 1e-8; p->ratio_tstep = 1.1;
  p->max_runtime_allowed = 60.0; p->Diff2DesorRatio = 0.5; p->special_gH_E_diff = 225.0;
  p->mxstep_per_interval = 6000; p->steps_reset_solver = 50; p->tol_policy_j = 1; p->max_steps_per_cell = 0;
}

int racgpu_n_record(const racgpu_params *p, double t0, double t_max) {
  return (int)std::ceil(std::log((t_max - t0) / p->dt_first_step * (p->ratio_tstep - 1.0) + 1.0) / std::log(p->ratio_tstep)) + 1;
}

int racgpu_set_tolerances(const racgpu_network *h, const racgpu_params *p, int32_t j, double d2h, double *rtol, double *atol) {
  if (!h) return fail("null network");
  const HostNetwork &n = h->net;
  double r, a, rT, aT;
  switch (j) {
    case 1: r = p->RTOL; a = p->ATOL; rT = 1e-3; aT = 1e-1; break;
    case 2: r = std::fmin(p->RTOL * 1e1, 1e-4); a = std::fmin(p->ATOL * 1e5, 1e-25); rT = 1e-2; aT = 1e-1; break;
    case 3: r = std::fmin(p->RTOL * 1e2, 1e-4); a = std::fmin(p->ATOL * 1e10, 1e-20); rT = 1e-3; aT = 1e0; break;
    case 4: r = std::fmin(p->RTOL * 1e2, 1e-4); a = std::fmin(p->ATOL * 1e10, 1e-18); rT = 1e-3; aT = 1e0; break;
    default: r = std::fmin(p->RTOL * std::pow(2.0, j), 1e-3); a = std::fmin(p->ATOL * std::pow(1e2, j), 1e-15); rT = 1e-2; aT = 1e0;
  }
  for (int i = 0; i < n.nS; ++i) { rtol[i] = r; atol[i] = a; }
  rtol[n.nS] = rT; atol[n.nS] = aT;
  for (int k = 0; k < 10; ++k) if (n.idx10[k] > 0) { rtol[n.idx10[k] - 1] = std::fmax(p->RTOL, 1e-4); atol[n.idx10[k] - 1] = std::fmax(p->ATOL, 1e-30); }
  if (n.i_Grain0 > 0)
    for (int g : {n.i_Grain0, n.i_GrainM, n.i_GrainP}) if (g > 0) { rtol[g - 1] = 1e-4; atol[g - 1] = std::fmax(d2h * 1e-6, 1e-30); }
  for (int g : n.grain) { rtol[g - 1] = std::fmax(p->RTOL, 1e-3); atol[g - 1] = std::fmax(p->ATOL, d2h * 1e-8); }
  return 0;
}

int racgpu_init_abundances(const racgpu_network *h, const double *y0, const double *cells, int64_t ncell, double *y) {
  if (!h) return fail("null network");
  const int nS = h->net.nS, ig = h->net.i_Grain0;
  for (int64_t c = 0; c < ncell; ++c) {
    std::memcpy(y + c * nS, y0, (size_t)nS * sizeof(double));
    if (ig > 0) y[c * nS + ig - 1] = cells[c * RACGPU_NPAR + RACGPU_P_D2H];
  }
  return 0;
}

int racgpu_set_device(int dev) { return guarded([&] { HIP_OK(hipSetDevice(dev)); }); }

int racgpu_set_stream(racgpu_network *h, void *s) {
  if (!h) return fail("null network");
  h->stream = (hipStream_t)s;
  return 0;
}

static size_t lds_bytes(const DevNet &dn) { return (size_t)5 * ((dn.nS + 1) & ~1) * sizeof(double); }

int racgpu_rates(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, double *rates) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dr(rates, (size_t)ncell * h->dn.nR * 8, RACGPU_MEM_HOST, false);
    hipLaunchKernelGGL(k_rates, dim3((unsigned)ncell), dim3(64), 0, h->stream, h->dn_dev, h->dp_dev, (const double *)dc.d, (double *)dr.d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dr.copy_out();
  });
}

int racgpu_rhs(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, const double *y, double *ydot) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dy(y, ncell * nS * 8, RACGPU_MEM_HOST, true),
        dd(ydot, ncell * nS * 8, RACGPU_MEM_HOST, false);
    h->ensure_workspace((long)ncell, (long)ncell);
    hipLaunchKernelGGL(k_rhs, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, (const double *)dc.d, (const double *)dy.d,
                       h->ws.rates, (double *)dd.d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dd.copy_out();
  });
}

int racgpu_jac_csc(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, const double *y, double *vals) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dy(y, ncell * nS * 8, RACGPU_MEM_HOST, true),
        dv(vals, (size_t)ncell * h->dn.nnzJ * 8, RACGPU_MEM_HOST, false);
    h->ensure_workspace((long)ncell, (long)ncell);
    hipLaunchKernelGGL(k_jac, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, (const double *)dc.d, (const double *)dy.d,
                       h->ws.rates, (double *)dv.d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    dv.copy_out();
  });
}

int racgpu_newton_solve(racgpu_network *h, const racgpu_params *p, const double *cells, int64_t ncell, const double *y, double gamma, double *bx) {
  if (!h) return fail("null network");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, RACGPU_MEM_HOST, true), dy(y, ncell * nS * 8, RACGPU_MEM_HOST, true),
        db(bx, ncell * nS * 8, RACGPU_MEM_HOST, true);
    h->ensure_workspace((long)ncell, (long)ncell);
    const char *rep_env = getenv("RACGPU_DEBUG_REPEAT"); // developer aid: time the factorisation + solve in isolation
    const int repeat = rep_env ? std::max(1, atoi(rep_env)) : 1;
    long long *cyc_dev = nullptr;
    if (rep_env) { HIP_OK(hipMalloc((void **)&cyc_dev, (size_t)ncell * 8 * sizeof(long long))); HIP_OK(hipMemset(cyc_dev, 0, (size_t)ncell * 8 * sizeof(long long))); }
    hipLaunchKernelGGL(k_newton, dim3((unsigned)ncell), dim3(64), lds_bytes(h->dn), h->stream, h->dn_dev, h->dp_dev, h->ws, (const double *)dc.d,
                       (const double *)dy.d, gamma, (double *)db.d, repeat, cyc_dev);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(h->stream));
    if (cyc_dev) {
      std::vector<long long> cy((size_t)ncell * 8);
      HIP_OK(hipMemcpy(cy.data(), cyc_dev, cy.size() * sizeof(long long), hipMemcpyDeviceToHost));
      (void)hipFree(cyc_dev);
      double a[6] = {0, 0, 0, 0, 0, 0};
      for (int64_t c = 0; c < ncell; ++c) for (int k = 0; k < 6; ++k) a[k] += (double)cy[(size_t)c * 8 + k];
      const double den = (double)ncell * repeat;
      fprintf(stderr, "[racgpu debug] %lld cells x %d: cycles per LU %.0f (scatter %.0f, lds pivots %.0f, register pivots %.0f, finish %.0f), per solve %.0f\n",
              (long long)ncell, repeat, a[0] / den, a[2] / den, a[3] / den, a[4] / den, a[5] / den, a[1] / den);
    }
    db.copy_out();
  });
}

int racgpu_set_cost_hints(racgpu_network *h, const double *cost, int64_t ncell) {
  if (!h) return fail("null network");
  if (!cost || ncell <= 0) { h->cost_hints.clear(); return 0; }
  return guarded([&] { h->cost_hints.assign(cost, cost + ncell); });
}

int64_t racgpu_workspace_bytes_per_cell(const racgpu_network *h) {
  if (!h) return -1;
  const HostNetwork &n = h->net;
  const int64_t npad = (n.nS + 63) / 64 * 64;
  return 8 * ((int64_t)n.nR + 6 * npad + (int64_t)n.Jrow.size() + n.sym.nzl + n.sym.nzu + 3 * npad);
}

double racgpu_last_kernel_ms(const racgpu_network *h) {
  if (!h || !h->timed) return -1.0;
  float ms = -1.f;
  if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0;
  return ms;
}

int racgpu_solve_batch(racgpu_network *h, const racgpu_params *p, int64_t ncell, const double *cells, double *y, double *t_final,
                       int32_t *quality, int64_t *stats, double *record, double *touts, int mem) {
  if (!h) return fail("null network");
  if (ncell <= 0) return 0;
  if (ncell > 0x7fffffffLL) return fail("ncell exceeds 2^31-1");
  return guarded([&] {
    h->upload();
    DevParams P = to_dev(p);
    HIP_OK(hipMemcpyAsync(h->dp_dev, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream)); // P lives on this stack frame
    const size_t nS = h->dn.nS;
    const size_t lds = lds_bytes(h->dn);
    const long per_cu = std::max<long>(1, std::min<long>(8, (long)(160 * 1024 / lds)));
    const long slots = std::min<long>((long)ncell, per_cu * h->cu_count);
    const long chunk_cells = std::max<long>(slots, std::min<long>((long)ncell, (long)(8e9 / (8.0 * h->dn.nR)))); // <= 8 GB of rates
    h->ensure_workspace(slots, chunk_cells);
    std::vector<double> trace_host;
    DevBuf dtrace(nullptr, 0, RACGPU_MEM_HOST, false);
    h->ws.trace = nullptr;
    if (P.debug_max_calls > 0) {
      trace_host.assign((size_t)P.debug_max_calls * 8, 0.0);
      new (&dtrace) DevBuf(trace_host.data(), trace_host.size() * 8, RACGPU_MEM_HOST, true);
      h->ws.trace = (double *)dtrace.d;
    }
    DevBuf dc(cells, (size_t)ncell * RACGPU_NPAR * 8, mem, true), dy(y, ncell * nS * 8, mem, true), dt(t_final, ncell * 8, mem, false),
        dq(quality, ncell * 4, mem, false), ds(stats, ncell * RACGPU_NSTAT * 8, mem, false),
        drec(record, (size_t)ncell * P.n_record * (nS + 1) * 8, mem, false), dto(touts, (size_t)ncell * P.n_record * 8, mem, false);
    int *marker_host = nullptr;
    h->ws.marker = nullptr;
    const char *dbgwait = std::getenv("RACGPU_DEBUG_WAIT");
    if (dbgwait) {
      HIP_OK(hipHostMalloc((void **)&marker_host, 64, hipHostMallocMapped));
      *marker_host = 0;
      HIP_OK(hipHostGetDevicePointer((void **)&h->ws.marker, marker_host, 0));
    }
    // optional longest-expected-first order, per chunk (indices relative to the chunk)
    const bool hinted = (int64_t)h->cost_hints.size() == ncell;
    if (hinted) {
      if (h->order_cap < (long)ncell) {
        HIP_OK(hipStreamSynchronize(h->stream));
        if (h->order_dev) (void)hipFree(h->order_dev);
        HIP_OK(hipMalloc((void **)&h->order_dev, (size_t)ncell * sizeof(int)));
        h->order_cap = (long)ncell;
      }
      std::vector<int> order((size_t)ncell);
      for (long c0 = 0; c0 < (long)ncell; c0 += chunk_cells) {
        const long nc = std::min<long>(chunk_cells, (long)ncell - c0);
        for (long i = 0; i < nc; ++i) order[c0 + i] = (int)i;
        const double *cost = h->cost_hints.data() + c0;
        std::stable_sort(order.begin() + c0, order.begin() + c0 + nc, [&](int a, int b) { return cost[a] > cost[b]; });
      }
      HIP_OK(hipMemcpy(h->order_dev, order.data(), (size_t)ncell * sizeof(int), hipMemcpyHostToDevice));
    }
    for (long c0 = 0; c0 < (long)ncell; c0 += chunk_cells) {
      const long nc = std::min<long>(chunk_cells, (long)ncell - c0);
      const double *cells_c = (const double *)dc.d + (size_t)c0 * RACGPU_NPAR;
      // pass 1: rate coefficients of every cell of the chunk (one wave per cell)
      hipLaunchKernelGGL(k_rates, dim3((unsigned)nc), dim3(64), 0, h->stream, h->dn_dev, h->dp_dev, cells_c, h->ws.rates);
      HIP_OK(hipGetLastError());
      // pass 2: the persistent integrator
      HIP_OK(hipMemsetAsync(h->ws.counter, 0, sizeof(int), h->stream));
      if (c0 == 0) HIP_OK(hipEventRecord(h->ev0, h->stream));
      hipLaunchKernelGGL(k_solve, dim3((unsigned)std::min<long>(slots, nc)), dim3(64), lds, h->stream, h->dn_dev, h->dp_dev, h->ws, (int)nc, cells_c,
                         (double *)dy.d + (size_t)c0 * nS, dt.d ? (double *)dt.d + c0 : nullptr, dq.d ? (int *)dq.d + c0 : nullptr,
                         ds.d ? (long long *)ds.d + (size_t)c0 * RACGPU_NSTAT : nullptr,
                         drec.d ? (double *)drec.d + (size_t)c0 * P.n_record * (nS + 1) : nullptr,
                         dto.d ? (double *)dto.d + (size_t)c0 * P.n_record : nullptr, hinted ? h->order_dev + c0 : nullptr);
      HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    if (dbgwait) { // developer aid: watch the progress word; give up (and leave the process) instead of hanging
      const double limit = std::atof(dbgwait);
      double waited = 0.0; int last = -1;
      while (hipStreamQuery(h->stream) == hipErrorNotReady) {
        struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr); waited += 0.1;
        const int m = *(volatile int *)marker_host;
        if (m != last) { std::fprintf(stderr, "[racgpu marker] t=%.1fs marker=%d\n", waited, m); last = m; }
        if (waited > limit) { std::fprintf(stderr, "[racgpu marker] still running after %.1fs at marker=%d: giving up\n", waited, m); std::fflush(stderr); _exit(3); }
      }
      std::fprintf(stderr, "[racgpu marker] finished after %.1fs, last marker=%d\n", waited, *(volatile int *)marker_host);
    }
    if (mem == RACGPU_MEM_HOST) {
      HIP_OK(hipStreamSynchronize(h->stream));
      dy.copy_out(); dt.copy_out(); dq.copy_out(); ds.copy_out(); drec.copy_out(); dto.copy_out();
    }
    if (P.debug_max_calls > 0) {
      HIP_OK(hipStreamSynchronize(h->stream));
      dtrace.copy_out();
      for (int i = 0; i < P.debug_max_calls; ++i) {
        const double *tr = &trace_host[(size_t)i * 8];
        std::fprintf(stderr, "[racgpu trace] call %3d tn=%.6e h=%.6e hu=%.6e nq=%g kflag=%g nst=%g nfe=%g nje/nlu=%g\n", i, tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], tr[6], tr[7]);
      }
    }
  });
}

} // extern "C"
