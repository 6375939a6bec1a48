#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-7, lane + 100, 0x111, 0xF, 0xF, false);
  out[64 + lane] = __builtin_amdgcn_update_dpp(-7, lane + 100, 0x101, 0xF, 0xF, false);
  int v = lane + 100;
  if (lane % 3 == 0) v = 999;     // divergent assignment then converge
  out[128 + lane] = __builtin_amdgcn_update_dpp(-7, v, 0x111, 0xF, 0xF, false);
}
int main() {
  int* d; (void)hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); int h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 3; ++r) { for (int i = 0; i < 64; ++i) printf("%d ", h[r * 64 + i]); printf("\n"); }
  return 0;
}
