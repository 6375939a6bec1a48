#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void cas_add(float* p, float x) {
  int* ip = (int*)p; int old = *ip;
  while (true) { const int assumed = old; old = atomicCAS(ip, assumed, __float_as_int(__int_as_float(assumed) + x)); if (old == assumed) break; }
}
template <int MODE>
__global__ void __launch_bounds__(256, 3) k(float* out, int iters) {
  __shared__ double tabd[2048];
  float* tab = (float*)tabd; unsigned* tabu = (unsigned*)tabd; unsigned long long* tabl = (unsigned long long*)tabd;
  for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    const int a = (threadIdx.x + it * 64) & 2047;
    const int a4 = (it * 64 + (lane >> 2)) & 2047;
    if (MODE == 0) atomicAdd(&tabu[a], 3u);
    if (MODE == 1) atomicAdd(&tabu[a4], 3u);
    if (MODE == 2) atomicAdd(&tabl[a], 3ull);
    if (MODE == 3) atomicAdd(&tabd[a], 1.0);
    if (MODE == 4) cas_add(&tab[a], 1.0f);
    if (MODE == 5) cas_add(&tab[a4], 1.0f);
    if (MODE == 6) atomicAdd(&tab[a], 1.0f);
    if (MODE == 7) atomicMax(&tabu[a], (unsigned)it);
    if (MODE == 8) atomicAdd(&tabd[a4], 1.0);
    if (MODE == 9) { if (lane < 16) atomicAdd(&tab[a], 1.0f); }     // 16 active lanes
    if (MODE == 10) { if (lane < 4) atomicAdd(&tab[a], 1.0f); }     // 4 active lanes
    if (MODE == 11) { if ((lane & 3) == 0) atomicAdd(&tab[a], 1.0f); }     // 16 active lanes, spread
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[5];
}
int main() {
  float* out; (void)hipMalloc(&out, 1 << 20);
  const int iters = 2000, blocks = 256 * 3 * 4;
  const char* names[] = {"u32 add distinct", "u32 add 4/addr", "u64 add distinct", "f64 add distinct", "cas-loop f32 distinct", "cas-loop f32 4/addr", "f32 add distinct", "u32 max distinct", "f64 add 4/addr", "f32 add 16 lanes", "f32 add 4 lanes", "f32 add 16 lanes spread"};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define RUN(M) { k<M><<<blocks, 256>>>(out, 10); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0); k<M><<<blocks, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); \
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); double wi = (double)blocks * 4 * iters; \
    printf("%-26s %8.3f ms  %.1f clk per wave-instr per CU\n", names[M], ms, ms * 1e-3 * 2.4e9 * 256 / wi); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  return 0;
}
