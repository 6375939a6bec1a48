// Sustained MFMA rate on random operands (the chip clocks to its power budget: the datasheet peak is not what a long kernel sees).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/micro/mfma_rate.hip -o tools/dev/micro/mfma_rate.bin && tools/dev/micro/mfma_rate.bin
// Per wave: 4 independent 32x32 accumulators, register operands only (no LDS, no memory in the loop); 1 / 2 waves per SIMD; ~1 ms per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the same MFMA stream with its operands read from LDS every step (4 x ds_read_b128 per 16 float32 MFMAs, 8 per 12 bf16 MFMAs: the
// convolution's inner loop without staging): what the LDS -> MFMA hand-off alone costs
template <int MODE, int DBUF>
__global__ __launch_bounds__(256, 2) void kl(const float* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < 4032; i += 256) lds[i] = ((const float4*)in)[i & 4095];
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const float4* base = lds + lane;
  constexpr int NF = MODE == 0 ? 4 : 8;
  float4 f[2][NF];
  auto rd = [&](float4* d, int it) {
#pragma unroll
    for (int i = 0; i < NF; ++i) d[i] = base[((it * NF + i) * 64) & 2047];
  };
  if (DBUF) rd(f[0], 0);
#pragma unroll 2
  for (int it = 0; it < iters; ++it) {
    float4* c = f[DBUF ? (it & 1) : 0];
    if (DBUF) rd(f[(it + 1) & 1], it + 1); else rd(c, it);
    if (MODE == 0) {
      const float a[2][4] = {{c[0].x, c[0].y, c[0].z, c[0].w}, {c[1].x, c[1].y, c[1].z, c[1].w}}, b[2][4] = {{c[2].x, c[2].y, c[2].z, c[2].w}, {c[3].x, c[3].y, c[3].z, c[3].w}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][e], b[0][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][e], b[1][e], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][e], b[0][e], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][e], b[1][e], acc[3], 0, 0, 0);
      }
    } else {
      bf16x8 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(bf16x8, c[i]);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[(2 * e) & 7], v[4 + ((2 * e) & 3)], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[(2 * e) & 7], v[5 + ((2 * e) & 2)], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[(2 * e + 1) & 7], v[4 + ((2 * e) & 3)], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[(2 * e + 1) & 7], v[5 + ((2 * e) & 2)], acc[3], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int t = blockIdx.x * 256 + threadIdx.x;
  float4 a0 = ((const float4*)in)[(t * 4 + 0) & 4095], a1 = ((const float4*)in)[(t * 4 + 1) & 4095];
  float4 b0 = ((const float4*)in)[(t * 4 + 2) & 4095], b1 = ((const float4*)in)[(t * 4 + 3) & 4095];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      const float a[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}}, b[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][e], b[0][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][e], b[1][e], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][e], b[0][e], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][e], b[1][e], acc[3], 0, 0, 0);
      }
    } else {
      const bf16x8 A0 = __builtin_bit_cast(bf16x8, a0), A1 = __builtin_bit_cast(bf16x8, a1), B0 = __builtin_bit_cast(bf16x8, b0), B1 = __builtin_bit_cast(bf16x8, b1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[3], 0, 0, 0);
      }
    }
    a0.x += 1e-7f;       // (keeps the loop from being folded; negligible issue cost)
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

int main() {
  std::vector<float> h(16384);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;          // as bf16 pairs these bits are random bf16 values too
  float *in, *out;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 256 * 8 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = 256 * wps, iters = mode == 0 ? 20000 : 40000;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 16 * (mode == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
        if (rep == 2)
          printf("%s  %d wave(s)/SIMD  %8.3f ms  %8.1f TFLOP/s  (%.1f %% of the %s peak)\n", mode == 0 ? "v_mfma_f32_32x32x2_f32  " : "v_mfma_f32_32x32x16_bf16", wps, ms,
                 flop / ms / 1e9, 100 * flop / ms / 1e9 / (mode == 0 ? 157.3 : 2500.0), mode == 0 ? "157.3 TFLOP/s" : "2.5 PFLOP/s");
      }
    }
  // operands through LDS: 512 workgroups (2 per CU), 64 512 B of dynamic LDS each like the convolution
  for (int mode = 0; mode < 2; ++mode)
    for (int dbuf = 0; dbuf < 2; ++dbuf) {
      const int blocks = 512, iters = mode == 0 ? 10000 : 20000, lds = 64512;
      auto kern = mode == 0 ? (dbuf ? kl<0, 1> : kl<0, 0>) : (dbuf ? kl<1, 1> : kl<1, 0>);
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double per = mode == 0 ? 16 * 2.0 * 32 * 32 * 2 : 12 * 2.0 * 32 * 32 * 16;
      const double flop = (double)blocks * 4 * iters * per;
      printf("%s operands via LDS (%d ds_read_b128 per %d MFMAs, %s)  %8.3f ms  %8.1f TFLOP/s  (%.1f %% of peak)\n", mode == 0 ? "f32 " : "bf16", mode == 0 ? 4 : 8,
             mode == 0 ? 16 : 12, dbuf ? "reads one step ahead" : "reads in front of their MFMAs", ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / (mode == 0 ? 157.3 : 2500.0));
    }
  return 0;
}
