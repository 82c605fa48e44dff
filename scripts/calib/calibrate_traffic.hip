// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950: every kernel moves a KNOWN
// number of bytes through a KNOWN access shape, so that the factor between the counter and the bytes can be read
// off per access width (MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads: x2).
// Built into ryujin_amd/lib/libryujin_calib.so (scripts/calibrate_traffic.py); not part of the product library.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace
{
  template <typename T>
  __global__ void __launch_bounds__(256) k_calib_stream_read(const T *__restrict__ in, double *__restrict__ out, size_t n)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) {
      const T v = in[q];
      acc += reinterpret_cast<const double *>(&v)[0];
    }
    if (acc == 12345.678)
      out[0] = acc; /* never true: keeps the loads alive, writes nothing */
  }

  __global__ void __launch_bounds__(256) k_calib_stream_write(double2 *__restrict__ out, size_t n)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride)
      out[q] = double2{1., 2.};
  }

  /* gather of RECORD-byte records (as the per-node state / Riemann-record gathers of the sweeps): lane l of a wave
   * reads record idx[q]; idx is either the identity shifted by a stencil offset (the locality-preserving numbering
   * of the structured meshes: consecutive lanes, consecutive records) or a random permutation */
  template <int RECORD>
  __global__ void __launch_bounds__(256) k_calib_gather(const uint32_t *__restrict__ idx, const double *__restrict__ rec,
                                                        double *__restrict__ out, size_t n)
  {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n)
      return;
    const double2 *b = reinterpret_cast<const double2 *>(rec + (size_t)idx[q] * (RECORD / 8));
    double acc = 0.;
#pragma unroll
    for (int g = 0; g < RECORD / 16; ++g) {
      const double2 t = b[g];
      acc += t.x + t.y;
    }
    if (acc == 12345.678)
      out[0] = acc;
  }
} // namespace

extern "C" {
/* which: 0 stream read 16 B/lane, 1 stream read 8 B/lane, 2 stream read 4 B/lane, 3 stream write 16 B/lane,
 *        4 gather 32-B records, 5 gather 64-B records (idx: n entries). Returns hipError_t. */
int ryujin_calib_run(int which, const void *in, const uint32_t *idx, void *out, size_t n)
{
  const int grid = 256 * 16;
  switch (which) {
  case 0: hipLaunchKernelGGL(k_calib_stream_read<double2>, dim3(grid), dim3(256), 0, nullptr, (const double2 *)in, (double *)out, n); break;
  case 1: hipLaunchKernelGGL(k_calib_stream_read<double>, dim3(grid), dim3(256), 0, nullptr, (const double *)in, (double *)out, n); break;
  case 2: hipLaunchKernelGGL(k_calib_stream_read<float>, dim3(grid), dim3(256), 0, nullptr, (const float *)in, (double *)out, n); break;
  case 3: hipLaunchKernelGGL(k_calib_stream_write, dim3(grid), dim3(256), 0, nullptr, (double2 *)out, n); break;
  case 4: hipLaunchKernelGGL(k_calib_gather<32>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, idx, (const double *)in, (double *)out, n); break;
  case 5: hipLaunchKernelGGL(k_calib_gather<64>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, idx, (const double *)in, (double *)out, n); break;
  default: return -1;
  }
  return (int)hipDeviceSynchronize();
}
void *ryujin_calib_alloc(size_t bytes)
{
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess)
    return nullptr;
  (void)hipMemset(p, 0, bytes);
  (void)hipDeviceSynchronize();
  return p;
}
int ryujin_calib_upload(void *dst, const void *src, size_t bytes) { return (int)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); }
void ryujin_calib_free(void *p) { (void)hipFree(p); }
}
