// micro-benchmark: cost of LDS float atomics / CAS on gfx950 under different address patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256, 3) k(float* out, int iters, const int* idx) {
  __shared__ float tab[4096];
  __shared__ int keys[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) { tab[i] = 0.f; keys[i] = -1; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int a;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) a = (threadIdx.x + it * 64) & 4095;                 // distinct, conflict-free
    if (MODE == 1) a = idx[(it * 256 + threadIdx.x) & 65535] & 4095;   // random
    if (MODE == 2) a = (it * 7 + (threadIdx.x >> 6)) & 4095;           // same address in the wave
    if (MODE == 3) a = (it * 64 + (lane >> 2) * 1) & 4095;             // 4 lanes per address
    if (MODE == 4) a = (it * 64 + (lane >> 4) * 1) & 4095;             // 16 lanes per address
    if (MODE == 5) a = (idx[(it * 256 + threadIdx.x) & 65535] & 4095);  // random, stride 6 (like val[slot][6])
    if (MODE == 5) { a = (a % 680) * 6; }
    if (MODE == 6) {                                                    // CAS find on random key
      const int key = idx[(it * 256 + threadIdx.x) & 65535] & 1023;
      unsigned h = ((unsigned)key * 2654435761u) & 2047;
      for (int p = 0; p < 32; ++p) { const int kk = atomicCAS(&keys[h], -1, key); if (kk == -1 || kk == key) break; h = (h + 1) & 2047; }
      acc += h;
      continue;
    }
    if (MODE == 7) { acc += atomicAdd(&tab[(threadIdx.x + it * 64) & 4095], 1.0f); continue; }   // returning, distinct
    if (MODE == 8) { tab[(threadIdx.x + it * 64) & 4095] += 1.0f; continue; }                    // plain RMW (racy) baseline
    atomicAdd(&tab[a], 1.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[5] + acc;
}
int main() {
  float* out; int* idx; hipMalloc(&out, 1 << 20); hipMalloc(&idx, 65536 * 4);
  std::vector<int> h(65536); unsigned s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
  hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 2000, blocks = 256 * 3 * 4;
  const char* names[] = {"distinct", "random", "same-addr x64", "4 lanes/addr", "16 lanes/addr", "random stride6", "CAS find random key", "returning distinct", "plain rmw"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(M) { k<M><<<blocks, 256>>>(out, 10, idx); hipDeviceSynchronize(); hipEventRecord(e0); k<M><<<blocks, 256>>>(out, iters, idx); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); double wi = (double)blocks * 4 * iters; /* wave instrs */ \
    printf("%-22s %8.3f ms  %.1f clk per wave-instr per CU (2.4 GHz, 256 CUs)\n", names[M], ms, ms * 1e-3 * 2.4e9 * 256 / wi); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
  return 0;
}
