// multi.hip -- one host process, N GPUs of one node: the cells of a batch dealt over the devices, every device integrating its share
// with the single-device engine, ONE RCCL all-gather of the results over xGMI (BASELINE.json north_star: "cells shard embarrassingly
// across the 8 GPUs of one node, with a single RCCL gather over xGMI at output").  Built on the public C ABI (include/racgpu.h) and the
// HIP runtime only; RCCL is loaded on first use (dlopen), so the single-device path neither links nor loads it.
//
// The reference has no counterpart: its cell sweep is a serial loop (src/disk.f90:864-938).  What is mirrored is the caller's per-cell
// contract of racgpu_calc_cells (calc_this_cell's local-iteration loop, src/disk.f90:1651-1791).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/racgpu.h"

namespace {

// the few RCCL entry points used, bound at run time (rccl/rccl.h: ncclResult_t = int, ncclSuccess = 0, ncclFloat64 = 8)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  void load() {
    if (lib) return;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) throw std::runtime_error(std::string("cannot load librccl.so: ") + dlerror());
    auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) throw std::runtime_error(std::string("librccl.so lacks ") + n); return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllGather = (decltype(AllGather))sym("ncclAllGather"); GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
  }
};
constexpr int kNcclFloat64 = 8;

#define HIP_OK(x)                                                                                       \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_));     \
  } while (0)

thread_local std::string g_merr;

// one flat row per cell: [y(nS) | t_final | quality | stats(NSTAT) | cell_out(NOUT)] as f64, so that ONE collective moves everything
__global__ void k_pack_rows(int n, int nS, int ncol, const double *__restrict__ y, const double *__restrict__ tf, const int *__restrict__ q,
                            const long long *__restrict__ st, const double *__restrict__ co, double *__restrict__ rows) {
  const int c = blockIdx.x;
  if (c >= n) return;
  double *r = rows + (size_t)c * ncol;
  for (int i = threadIdx.x; i < nS; i += blockDim.x) r[i] = y[(size_t)c * nS + i];
  if (threadIdx.x == 0) {
    r[nS] = tf[c]; r[nS + 1] = (double)q[c];
    for (int k = 0; k < RACGPU_NSTAT; ++k) r[nS + 2 + k] = (double)st[(size_t)c * RACGPU_NSTAT + k];
    for (int k = 0; k < RACGPU_NOUT; ++k) r[nS + 2 + RACGPU_NSTAT + k] = co[(size_t)c * RACGPU_NOUT + k];
  }
}

} // namespace

struct racgpu_multi {
  int ndev = 0;
  std::vector<int> devices;
  std::vector<racgpu_network *> nets;
  std::vector<hipStream_t> streams;
  std::vector<void *> comms;
  Rccl rccl;
  int nS = 0;
  ~racgpu_multi() {
    for (size_t i = 0; i < comms.size(); ++i) if (comms[i]) rccl.CommDestroy(comms[i]);
    for (size_t i = 0; i < nets.size(); ++i) { (void)hipSetDevice(devices[i]); if (streams[i]) (void)hipStreamDestroy(streams[i]); racgpu_network_destroy(nets[i]); }
  }
};

// The dealing rule, also used by the tests: cells in order of decreasing cost (stable; cost == NULL: index order), dealt round-robin,
// so every device gets the same mix.  share[d] = the cells of device d in the order it solves them.
extern "C" int racgpu_multi_deal(int ndev, int64_t ncell, const double *cost, int32_t *owner, int32_t *position) {
  if (ndev < 1 || ncell < 0 || !owner || !position) { g_merr = "racgpu_multi_deal: bad argument"; return -1; }
  std::vector<int64_t> order((size_t)ncell);
  std::iota(order.begin(), order.end(), 0);
  if (cost) std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return cost[a] > cost[b]; });
  for (int64_t k = 0; k < ncell; ++k) { owner[order[k]] = (int32_t)(k % ndev); position[order[k]] = (int32_t)(k / ndev); }
  return 0;
}

extern "C" {

const char *racgpu_multi_last_error(void) { return g_merr.c_str(); }

racgpu_multi *racgpu_multi_create(const char *network_path, int ndev, const int *devices) {
  try {
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) throw std::runtime_error("no HIP device visible: the racgpu compute path has no CPU fallback");
    if (ndev < 1 || ndev > have) throw std::runtime_error("racgpu_multi_create: ndev must be between 1 and the number of visible devices");
    auto m = std::make_unique<racgpu_multi>();
    m->ndev = ndev;
    for (int i = 0; i < ndev; ++i) m->devices.push_back(devices ? devices[i] : i);
    m->nets.assign(ndev, nullptr); m->streams.assign(ndev, nullptr); m->comms.assign(ndev, nullptr);
    for (int i = 0; i < ndev; ++i) {
      HIP_OK(hipSetDevice(m->devices[i]));
      m->nets[i] = racgpu_network_load(network_path);
      if (!m->nets[i]) throw std::runtime_error(racgpu_last_error());
      HIP_OK(hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking));
      if (racgpu_set_stream(m->nets[i], m->streams[i]) != 0) throw std::runtime_error(racgpu_last_error());
    }
    int32_t nS = 0;
    racgpu_network_dims(m->nets[0], &nS, nullptr, nullptr, nullptr, nullptr);
    m->nS = nS;
    m->rccl.load();
    const int rc = m->rccl.CommInitAll(m->comms.data(), ndev, m->devices.data());
    if (rc != 0) throw std::runtime_error(std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(rc));
    return m.release();
  } catch (const std::exception &e) { g_merr = e.what(); return nullptr; }
}

void racgpu_multi_destroy(racgpu_multi *m) { delete m; }
int racgpu_multi_ndev(const racgpu_multi *m) { return m ? m->ndev : 0; }
racgpu_network *racgpu_multi_network(racgpu_multi *m, int i) { return (m && i >= 0 && i < m->ndev) ? m->nets[i] : nullptr; }

int racgpu_multi_calc_cells(racgpu_multi *m, const racgpu_params *p, int32_t nlocal_iter, int64_t ncell, const double *cells, double *y,
                            double *t_final, int32_t *quality, int64_t *stats, double *cell_out, const double *cost) {
  if (!m || !p || !cells || !y) { g_merr = "racgpu_multi_calc_cells: null argument"; return -1; }
  if (ncell <= 0) return 0;
  try {
    const int ndev = m->ndev, nS = m->nS;
    const int ncol = nS + 2 + RACGPU_NSTAT + RACGPU_NOUT;
    std::vector<int32_t> owner((size_t)ncell), pos((size_t)ncell);
    racgpu_multi_deal(ndev, ncell, cost, owner.data(), pos.data());
    const int64_t nmax = (ncell + ndev - 1) / ndev; // rows per device in the gathered block (the last ones of some devices stay empty)
    struct Dev { std::vector<int64_t> mine; double *cells = nullptr, *y = nullptr, *tf = nullptr, *co = nullptr, *rows = nullptr, *all = nullptr; int *q = nullptr; long long *st = nullptr; };
    std::vector<Dev> D(ndev);
    for (int d = 0; d < ndev; ++d) D[d].mine.assign((size_t)nmax, -1);
    for (int64_t c = 0; c < ncell; ++c) D[owner[c]].mine[pos[c]] = c;
    std::vector<std::string> errs(ndev);
    auto work = [&](int d) { // one host thread per device (racgpu.h: a handle is used by one thread at a time)
      try {
        Dev &v = D[d];
        HIP_OK(hipSetDevice(m->devices[d]));
        const int64_t n = std::count_if(v.mine.begin(), v.mine.end(), [](int64_t c) { return c >= 0; });
        auto dmalloc = [&](size_t bytes) { void *q = nullptr; HIP_OK(hipMalloc(&q, std::max<size_t>(bytes, 8))); return q; };
        v.cells = (double *)dmalloc((size_t)nmax * RACGPU_NPAR * 8); v.y = (double *)dmalloc((size_t)nmax * nS * 8); v.tf = (double *)dmalloc(nmax * 8);
        v.q = (int *)dmalloc(nmax * 4); v.st = (long long *)dmalloc((size_t)nmax * RACGPU_NSTAT * 8); v.co = (double *)dmalloc((size_t)nmax * RACGPU_NOUT * 8);
        v.rows = (double *)dmalloc((size_t)nmax * ncol * 8); v.all = (double *)dmalloc((size_t)ndev * nmax * ncol * 8);
        std::vector<double> hc((size_t)n * RACGPU_NPAR), hy((size_t)n * nS);
        for (int64_t k = 0; k < n; ++k) {
          std::memcpy(&hc[(size_t)k * RACGPU_NPAR], cells + (size_t)v.mine[k] * RACGPU_NPAR, RACGPU_NPAR * 8);
          std::memcpy(&hy[(size_t)k * nS], y + (size_t)v.mine[k] * nS, (size_t)nS * 8);
        }
        HIP_OK(hipMemcpyAsync(v.cells, hc.data(), hc.size() * 8, hipMemcpyHostToDevice, m->streams[d]));
        HIP_OK(hipMemcpyAsync(v.y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice, m->streams[d]));
        HIP_OK(hipMemsetAsync(v.rows, 0, (size_t)nmax * ncol * 8, m->streams[d]));
        HIP_OK(hipMemsetAsync(v.co, 0, (size_t)nmax * RACGPU_NOUT * 8, m->streams[d]));
        HIP_OK(hipStreamSynchronize(m->streams[d])); // the staging vectors go out of scope below
        if (n > 0) {
          if (racgpu_calc_cells(m->nets[d], p, nlocal_iter, n, v.cells, v.y, v.tf, v.q, (int64_t *)v.st, v.co, RACGPU_MEM_DEVICE) != 0)
            throw std::runtime_error(racgpu_last_error());
          hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)n), dim3(64), 0, m->streams[d], (int)n, nS, ncol, (const double *)v.y, (const double *)v.tf,
                             (const int *)v.q, (const long long *)v.st, (const double *)v.co, v.rows);
          HIP_OK(hipGetLastError());
        }
        HIP_OK(hipStreamSynchronize(m->streams[d]));
      } catch (const std::exception &e) { errs[d] = e.what(); }
    };
    std::vector<std::thread> th;
    for (int d = 0; d < ndev; ++d) th.emplace_back(work, d);
    for (auto &t : th) t.join();
    for (int d = 0; d < ndev; ++d) if (!errs[d].empty()) throw std::runtime_error("device " + std::to_string(m->devices[d]) + ": " + errs[d]);
    // the path's single exchange: one all-gather of the result rows (RCCL over xGMI), all devices in one group call
    int rc = m->rccl.GroupStart();
    for (int d = 0; d < ndev && rc == 0; ++d) rc = m->rccl.AllGather(D[d].rows, D[d].all, (size_t)nmax * ncol, kNcclFloat64, m->comms[d], m->streams[d]);
    const int rc2 = m->rccl.GroupEnd();
    if (rc != 0 || rc2 != 0) throw std::runtime_error(std::string("ncclAllGather: ") + m->rccl.GetErrorString(rc != 0 ? rc : rc2));
    for (int d = 0; d < ndev; ++d) { HIP_OK(hipSetDevice(m->devices[d])); HIP_OK(hipStreamSynchronize(m->streams[d])); }
    // device 0's copy of the gathered block -> the caller's arrays
    HIP_OK(hipSetDevice(m->devices[0]));
    std::vector<double> all((size_t)ndev * nmax * ncol);
    HIP_OK(hipMemcpy(all.data(), D[0].all, all.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < ncell; ++c) {
      const double *r = &all[((size_t)owner[c] * nmax + pos[c]) * ncol];
      std::memcpy(y + (size_t)c * nS, r, (size_t)nS * 8);
      if (t_final) t_final[c] = r[nS];
      if (quality) quality[c] = (int32_t)r[nS + 1];
      if (stats) for (int k = 0; k < RACGPU_NSTAT; ++k) stats[(size_t)c * RACGPU_NSTAT + k] = (int64_t)r[nS + 2 + k];
      if (cell_out) for (int k = 0; k < RACGPU_NOUT; ++k) cell_out[(size_t)c * RACGPU_NOUT + k] = r[nS + 2 + RACGPU_NSTAT + k];
    }
    for (int d = 0; d < ndev; ++d) {
      (void)hipSetDevice(m->devices[d]);
      for (void *q : {(void *)D[d].cells, (void *)D[d].y, (void *)D[d].tf, (void *)D[d].q, (void *)D[d].st, (void *)D[d].co, (void *)D[d].rows, (void *)D[d].all}) (void)hipFree(q);
    }
    return 0;
  } catch (const std::exception &e) { g_merr = e.what(); return -1; }
}

} // extern "C"
