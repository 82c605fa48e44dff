// What HBM delivers for the read/write mix and stream count of the limiter sweeps (scripts/hbm_mix.sh).
// One thread per 16-byte element, one wave per 64 consecutive elements, as the sweeps; per element the kernel reads
// NR streams and writes NW streams (separate arrays, 1 GiB each: far beyond the 256 MiB Infinity Cache), optionally
// with the non-temporal hint. Reports bytes / time for every (NR, NW).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_mix(const v2d *const *__restrict__ in, v2d *const *__restrict__ out, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n)
    return;
  v2d acc = {0., 0.};
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const v2d v = NT ? __builtin_nontemporal_load(in[r] + i) : in[r][i];
    acc += v;
  }
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    v2d t = acc;
    t.x += (double)w;
    if (NT)
      __builtin_nontemporal_store(t, out[w] + i);
    else
      out[w][i] = t;
  }
}

template <int NR, int NW, bool NT>
int run(const v2d *const *d_in, v2d *const *d_out, size_t n)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const dim3 grid((unsigned)((n + 255) / 256));
  for (int k = 0; k < 2; ++k)
    hipLaunchKernelGGL((k_mix<NR, NW, NT>), grid, dim3(256), 0, 0, d_in, d_out, n);
  CHECK(hipEventRecord(a, 0));
  const int reps = 10;
  for (int k = 0; k < reps; ++k)
    hipLaunchKernelGGL((k_mix<NR, NW, NT>), grid, dim3(256), 0, 0, d_in, d_out, n);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)(NR + NW) * 16. * (double)n * reps;
  printf("| %d | %d | %.0f %% | %s | %.3f | %.2f |\n", NR, NW, 100. * NW / (NR + NW), NT ? "nt" : "plain", ms / reps,
         bytes / (ms * 1e-3) / 1e12);
  return 0;
}

int main()
{
  const size_t n = (size_t)1 << 26; /* 1 GiB per stream */
  constexpr int kMax = 8;
  std::vector<v2d *> in(kMax), out(kMax);
  for (int k = 0; k < kMax; ++k) {
    CHECK(hipMalloc(&in[k], n * sizeof(v2d)));
    CHECK(hipMalloc(&out[k], n * sizeof(v2d)));
    CHECK(hipMemset(in[k], 0, n * sizeof(v2d)));
    CHECK(hipMemset(out[k], 0, n * sizeof(v2d)));
  }
  v2d **d_in, **d_out;
  CHECK(hipMalloc(&d_in, kMax * sizeof(v2d *)));
  CHECK(hipMalloc(&d_out, kMax * sizeof(v2d *)));
  CHECK(hipMemcpy(d_in, in.data(), kMax * sizeof(v2d *), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_out, out.data(), kMax * sizeof(v2d *), hipMemcpyHostToDevice));
  printf("| read streams | write streams | written share | hint | ms per launch | TB/s |\n|---|---|---|---|---|---|\n");
  run<1, 0, false>(d_in, d_out, n);
  run<4, 0, false>(d_in, d_out, n);
  run<8, 0, false>(d_in, d_out, n);
  run<8, 0, true>(d_in, d_out, n);
  run<0, 1, false>(d_in, d_out, n);
  run<0, 4, false>(d_in, d_out, n);
  run<0, 4, true>(d_in, d_out, n);
  run<1, 1, false>(d_in, d_out, n);
  run<1, 1, true>(d_in, d_out, n);
  run<4, 1, false>(d_in, d_out, n);
  run<4, 1, true>(d_in, d_out, n);
  run<4, 2, false>(d_in, d_out, n);
  run<4, 2, true>(d_in, d_out, n);
  run<3, 2, true>(d_in, d_out, n);
  run<4, 4, false>(d_in, d_out, n);
  run<4, 4, true>(d_in, d_out, n);
  run<6, 2, true>(d_in, d_out, n);
  run<8, 2, true>(d_in, d_out, n);
  return 0;
}
