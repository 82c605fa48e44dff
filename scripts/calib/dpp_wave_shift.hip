#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out_shl, int *out_shr, int *ref_dn, int *ref_up)
{
  const int lane = threadIdx.x;
  const int x = 1000 + lane;
  out_shl[lane] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);
  out_shr[lane] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);
  ref_dn[lane] = __shfl_down(x, 1, 64);
  ref_up[lane] = __shfl_up(x, 1, 64);
}
int main()
{
  int *d; hipMalloc(&d, 4 * 64 * sizeof(int));
  k<<<1, 64>>>(d, d + 64, d + 128, d + 192);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 31, 32, 47, 48, 62, 63})
    printf("lane %2d: wave_shl %5d wave_shr %5d shfl_down %5d shfl_up %5d\n", l, h[l], h[64 + l], h[128 + l], h[192 + l]);
  int bad = 0;
  for (int l = 0; l < 63; ++l) bad += h[l] != 1000 + l + 1;
  for (int l = 1; l < 64; ++l) bad += h[64 + l] != 1000 + l - 1;
  printf("shl==take lane+1, shr==take lane-1: %s (lane63 shl=%d, lane0 shr=%d)\n", bad ? "NO" : "yes", h[63], h[64]);
  return 0;
}
