// micro-benchmark: VALU issue rate of one gfx950 SIMD — how many cycles does a wave64 v_fma_f32 / v_pk_fma_f32 / v_add_u32 occupy the
// SIMD for, at 1, 2, 4, 8 waves per SIMD, with independent (8 accumulators) and fully dependent (1 accumulator) instruction streams.
// Settles whether "SQ_INSTS_VALU x 4 cycles"  or the guide's 2 cycles per wave64 instruction prices the rasterisers.
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0000001f, c = 1e-9f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
  unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // 64 independent-ish fma: 8 accumulators round robin (distance 8 between dependent instructions)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (MODE == 1) {     // 64 dependent fma on one accumulator
      REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        : "+v"(a0) : "v"(b), "v"(c));)
    } else if (MODE == 2) {     // 64 packed fma (2 floats per lane each), 4 accumulators
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
    } else if (MODE == 3) {     // 64 integer adds, 8 accumulators
      REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u1));)
    } else if (MODE == 4) {     // the rasteriser's diet: compare + cndmask + med3 + mul/fma mix, 8 independent streams
      REP8(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_mul_f32 %1, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_sub_f32 %5, %5, %9\n v_max_f32 %6, %6, %9\n v_and_b32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                        (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc, int cus) {
  const int iters = 4096;                      // 64 instructions per iteration -> 262 144 instructions per wave
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps : {1, 2, 4, 8}) {               // waves per SIMD = workgroups (of 4 waves) per CU
    const int blocks = cus * wps;
    k<MODE><<<blocks, 256>>>(out, cyc, 16); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, cyc, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; unsigned long long mx = 0; for (auto v : h) { sum += v; mx = v > mx ? v : mx; }
    const double ninst = 64.0 * iters;
    // per-SIMD rate from the wall clock: every SIMD of every CU hosts `wps` waves (4 waves of a workgroup -> 4 SIMDs)
    const double inst_per_simd_per_s = ninst * wps / (ms * 1e-3);
    printf("  %-34s waves/SIMD %d  %8.3f ms  counter ticks / instr / wave: avg %.3f max %.3f   wave-instr / s / SIMD %.3f G  (= %.2f cycles per instr at 2.4 GHz)\n",
           name, wps, ms, sum / h.size() / ninst, mx / ninst, inst_per_simd_per_s * 1e-9, 2.4e9 / inst_per_simd_per_s);
  }
}

int main() {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("%s: %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4); (void)hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8);
  run<0>("v_fma_f32, 8 accumulators", out, cyc, cus);
  run<1>("v_fma_f32, dependent chain", out, cyc, cus);
  run<2>("v_pk_fma_f32, 4 accumulators", out, cyc, cus);
  run<3>("v_add_u32, 8 accumulators", out, cyc, cus);
  run<4>("raster mix (med3/cmp/cndmask/fma)", out, cyc, cus);
  return 0;
}
