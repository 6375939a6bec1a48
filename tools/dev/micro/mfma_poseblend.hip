// Micro-benchmark for the MFMA decision: the only dense contraction on C5's path, the SMPL-X-arm pose blend
//   out(b, m) = sum_k pm(b, k) * P(k, m),  B = 32 frames, K = 486 pose features, M = 3078 = 1026 vertices x 3
// (hand_models_harp/body_models.py:2335 -> smplx.lbs `pose_feature @ posedirs`), as
//   (a) the VALU form the product kernel uses (thread = vertex, 4 frames per workgroup, one pass over k)   [csrc/lbs_tree.hip]
//   (b) v_mfma_f32_16x16x4_f32 tiles: workgroup = 16 frames x 16 columns, 4 waves split K, LDS reduction
// build: hipcc --offload-arch=gfx950 -O3 tools/dev/micro/mfma_poseblend.hip -o tools/dev/micro/mfma_poseblend.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

constexpr int B = 32, K = 486, NV = 1026, M = NV * 3, FPB = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) valu_kernel(const float* __restrict__ pm, const float* __restrict__ P, float* __restrict__ out) {
  __shared__ float s_pm[FPB * K];
  const int v = blockIdx.x * 256 + threadIdx.x, b0 = blockIdx.y * FPB;
  for (int i = threadIdx.x; i < FPB * K; i += 256) s_pm[i] = pm[(size_t)b0 * K + i];
  __syncthreads();
  if (v >= NV) return;
  float acc[FPB][3] = {};
  for (int k = 0; k < K; ++k) {
    const float p0 = P[(size_t)k * M + 3 * v], p1 = P[(size_t)k * M + 3 * v + 1], p2 = P[(size_t)k * M + 3 * v + 2];
#pragma unroll
    for (int f = 0; f < FPB; ++f) { const float c = s_pm[f * K + k]; acc[f][0] += p0 * c; acc[f][1] += p1 * c; acc[f][2] += p2 * c; }
  }
  for (int f = 0; f < FPB; ++f)
    for (int c = 0; c < 3; ++c) out[(size_t)(b0 + f) * M + 3 * v + c] = acc[f][c];
}

// A[l&15][k=l>>4], B[k=l>>4][l&15]; D: col = lane & 15, row = (lane >> 4) * 4 + reg
__global__ void __launch_bounds__(256) mfma_kernel(const float* __restrict__ pm, const float* __restrict__ P, float* __restrict__ out) {
  __shared__ f32x4 s_acc[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int col = n0 + (lane & 15);
  const bool col_ok = col < M;
  const int steps = (K + 3) / 4, per = (steps + 3) / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* arow = pm + (size_t)(b0 + (lane & 15)) * K;
  for (int s = w * per; s < min(steps, (w + 1) * per); ++s) {
    const int k = 4 * s + (lane >> 4);
    const float a = k < K ? arow[k] : 0.f;
    const float b = (k < K && col_ok) ? P[(size_t)k * M + col] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  s_acc[w][lane] = acc;
  __syncthreads();
  if (w == 0 && col_ok) {
    const f32x4 t = s_acc[0][lane] + s_acc[1][lane] + s_acc[2][lane] + s_acc[3][lane];
    for (int r = 0; r < 4; ++r) out[(size_t)(b0 + (lane >> 4) * 4 + r) * M + col] = t[r];
  }
}

int main() {
  std::vector<float> hpm(B * K), hP((size_t)K * M);
  srand(1);
  for (auto& x : hpm) x = rand() / (float)RAND_MAX - 0.5f;
  for (auto& x : hP) x = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
  float *pm, *P, *o1, *o2;
  hipMalloc(&pm, hpm.size() * 4); hipMalloc(&P, hP.size() * 4); hipMalloc(&o1, (size_t)B * M * 4); hipMalloc(&o2, (size_t)B * M * 4);
  hipMemcpy(pm, hpm.data(), hpm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(P, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 200 * 1e3f;
  };
  const float t_valu = time([&] { hipLaunchKernelGGL(valu_kernel, dim3((NV + 255) / 256, B / FPB), dim3(256), 0, 0, pm, P, o1); });
  const float t_mfma = time([&] { hipLaunchKernelGGL(mfma_kernel, dim3((M + 15) / 16, B / 16), dim3(256), 0, 0, pm, P, o2); });
  std::vector<float> h1((size_t)B * M), h2((size_t)B * M);
  hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mr = 0;
  for (size_t i = 0; i < h1.size(); ++i) { md = fmax(md, fabs((double)h1[i] - h2[i])); mr = fmax(mr, fabs((double)h1[i])); }
  printf("pose blend (%d,%d)x(%d,%d): VALU %.2f us  MFMA(16x16x4 f32, split-K 4) %.2f us  back-to-back launches incl. ~launch gap; max|diff| %.3e of max|out| %.3e; %.1f MFLOP\n",
         B, K, K, M, t_valu, t_mfma, md, mr, 2.0 * B * K * M * 1e-6);
  return 0;
}
