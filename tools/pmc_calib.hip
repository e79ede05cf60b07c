// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access
// pattern k_solve uses: 64-lane waves, 8 bytes per lane, raw buffer loads/stores (buffer_load_dwordx2 ... offen), every wave
// streaming its OWN contiguous slice.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
//   k_calib_read   reads  NBYTES once (checksum keeps the loads alive), writes 8 B per wave
//   k_calib_write  writes NBYTES once, reads nothing
//   k_calib_reread reads a 0.37 MB slice per wave REPS times (the re-read pattern of the solver's LU/solve streams)
// Build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t mkbuf(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ double bload(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
__device__ __forceinline__ void bstore(rsrc_t r, int voff, int soff, double v) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
}
template <int AUX>
__device__ __forceinline__ double bload_aux(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX)); }
// the same sweeps with the `nt` cache policy (aux = 2), which the triangular solves use on the factor streams
__global__ __launch_bounds__(64) void k_calib_read_nt(const double *src, long nper, int reps, double *out) {
  const rsrc_t b = mkbuf(src + (size_t)blockIdx.x * nper);
  double s = 0.0;
  for (int r = 0; r < reps; ++r)
    for (long i0 = 0; i0 < nper; i0 += 64 * 4) {
      const double a0 = bload_aux<2>(b, threadIdx.x * 8, (int)(i0 * 8)), a1 = bload_aux<2>(b, threadIdx.x * 8, (int)(i0 * 8 + 512)),
                   a2 = bload_aux<2>(b, threadIdx.x * 8, (int)(i0 * 8 + 1024)), a3 = bload_aux<2>(b, threadIdx.x * 8, (int)(i0 * 8 + 1536));
      s += (a0 + a1) + (a2 + a3);
    }
  if (s == 12345.678) out[blockIdx.x] = s;
}
// one wave per workgroup, slice of nper doubles per wave, `reps` sweeps over it
__global__ __launch_bounds__(64) void k_calib_read(const double *src, long nper, int reps, double *out) {
  const rsrc_t b = mkbuf(src + (size_t)blockIdx.x * nper);
  double s = 0.0;
  for (int r = 0; r < reps; ++r)
    for (long i0 = 0; i0 < nper; i0 += 64 * 4) {
      const double a0 = bload(b, threadIdx.x * 8, (int)(i0 * 8)), a1 = bload(b, threadIdx.x * 8, (int)(i0 * 8 + 512)),
                   a2 = bload(b, threadIdx.x * 8, (int)(i0 * 8 + 1024)), a3 = bload(b, threadIdx.x * 8, (int)(i0 * 8 + 1536));
      s += (a0 + a1) + (a2 + a3);
    }
  if (s == 12345.678) out[blockIdx.x] = s; // never true for the fill pattern: keeps the loads, writes nothing
}
__global__ __launch_bounds__(64) void k_calib_write(double *dst, long nper) {
  const rsrc_t b = mkbuf(dst + (size_t)blockIdx.x * nper);
  for (long i0 = 0; i0 < nper; i0 += 64) bstore(b, threadIdx.x * 8, (int)(i0 * 8), (double)i0);
}
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int waves = 3072;                    // as many slices as the solver keeps in flight
  const long nper_stream = 131072;           // 1 MiB per wave -> 3 GiB streamed once (well beyond the 256 MiB Infinity Cache)
  const long nper_slice = 48640;             // 0.37 MiB per wave, as a solver slot; re-read 8 times
  double *buf = nullptr, *out = nullptr;
  OK(hipMalloc(&buf, (size_t)waves * nper_stream * 8));
  OK(hipMalloc(&out, waves * 8));
  OK(hipMemset(buf, 0, (size_t)waves * nper_stream * 8));
  OK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_calib_write, dim3(waves), dim3(64), 0, 0, buf, nper_stream);
  OK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_calib_read, dim3(waves), dim3(64), 0, 0, buf, nper_stream, 1, out);
  OK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_calib_read, dim3(waves), dim3(64), 0, 0, buf, nper_slice, 8, out);
  OK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_calib_read_nt, dim3(waves), dim3(64), 0, 0, buf, nper_stream, 1, out);
  OK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_calib_read_nt, dim3(waves), dim3(64), 0, 0, buf, nper_slice, 8, out);
  OK(hipDeviceSynchronize());
  printf("{\"k_calib_write_bytes\": %ld, \"k_calib_read_stream_bytes\": %ld, \"k_calib_read_reread_bytes\": %ld, \"reread_footprint_bytes\": %ld, "
         "\"k_calib_read_nt_stream_bytes\": %ld, \"k_calib_read_nt_reread_bytes\": %ld}\n",
         waves * nper_stream * 8, waves * nper_stream * 8, waves * nper_slice * 8 * 8, waves * nper_slice * 8, waves * nper_stream * 8, waves * nper_slice * 8 * 8);
  return 0;
}
