// micro-benchmark: global (memory-side) atomic throughput on gfx950 by type and address pattern
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(void* buf, int iters, unsigned nmask) {
  float* f = (float*)buf; double* d = (double*)buf; unsigned* u = (unsigned*)buf;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const unsigned ar = (s >> 8) & nmask;                 // random element
    const unsigned ac = (tid + it * 9973u * 64u) & nmask;  // coalesced: consecutive lanes -> consecutive elements
    if (MODE == 0) atomicAdd(&f[ac], 1.0f);
    if (MODE == 1) atomicAdd(&f[ar], 1.0f);
    if (MODE == 2) atomicAdd(&d[ac], 1.0);
    if (MODE == 3) atomicAdd(&d[ar], 1.0);
    if (MODE == 4) atomicAdd(&u[ac], 1u);
    if (MODE == 5) atomicAdd(&u[ar], 1u);
    if (MODE == 6) f[ac] += 1.0f;                          // plain RMW baseline
    if (MODE == 7) f[ar] += 1.0f;
    if (MODE == 8) __hip_atomic_fetch_add(&f[ar], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 9) { const unsigned a6 = (ar / 6) * 6; for (int c = 0; c < 6; ++c) atomicAdd(&f[(a6 + c) & nmask], 1.0f); }  // 6 consecutive floats per lane
  }
}
int main() {
  void* buf; (void)hipMalloc(&buf, 64 << 20); (void)hipMemset(buf, 0, 64 << 20);
  const int iters = 64, blocks = 256 * 16;
  const char* names[] = {"f32 coalesced", "f32 random", "f64 coalesced", "f64 random", "u32 coalesced", "u32 random", "plain rmw coalesced", "plain rmw random", "f32 random wg-scope", "f32 6 consecutive random"};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (unsigned nm : {(1u << 20) - 1, (1u << 23) - 1}) {     // 4 MB / 32 MB of floats
    printf("region: %u elements\n", nm + 1);
#define RUN(M) { k<M><<<blocks, 256>>>(buf, 4, nm); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); k<M><<<blocks, 256>>>(buf, iters, nm); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); double n = (double)blocks * 256 * iters * (M == 9 ? 6 : 1); \
    printf("  %-26s %8.3f ms  %.1f G atomics/s\n", names[M], ms, n / ms * 1e-6); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
  }
  return 0;
}
