// Second half of the texel gradient of the shader backward (harp_shade_args.trec, include/harp_hip.h): the backward of
// TexturesUV.sample_textures / F.grid_sample(bilinear, align_corners=True) (renderer/renderer_helper.py:472-523,
// renderer/pbr_materials.py:82-124) as "emit, then reduce by owner".
//
// Why: scattered from the pixel side, the 4 corners x 6 channels of every shaded pixel cost the shader backward a 6.3-KB LDS table per
// wave, its flush and 11.8 M memory-side float atomics per launch (C3) — while the 1.39 M shaded pixels of the 32 frames of a step land
// on only ~44 k distinct texels (all frames show the same chart of the same atlas).  The pixel pass now appends one 36-byte record per
// pixel to the list of the 32x32-texel UV tile its footprint starts in (coalesced 4-byte planes); here a workgroup owns one chunk of
// one tile's list, accumulates the footprints in a (33 x 33 texel) x 6 channel 64-bit fixed-point table in LDS (ds_add_u64, one
// scale per map and chunk: order-independent sums) and adds the table to the maps with row-contiguous memory atomics:
// ~0.3 M atomics per tile chunk set instead of 11.8 M, and the kernel feeds nothing but the optimiser — it runs beside the mesh / hand
// backward tail, off the step's critical path.
#include <stdlib.h>
#include "harp_common.h"
#include "harp_hip.h"

namespace {

constexpr int kBin = 32;                       // texels per tile side
constexpr int kAcc = kBin + 1;                 // + the row / column the last footprints reach into
constexpr int kChunk = 2048;                   // records per workgroup pass
constexpr int kThreads = 512;
constexpr int kMaxBins = 1024;                 // (1024 x 1024 texels)

// power-of-two scale s with |x| * s < 2^40 for every |x| <= m; inv = 1 / s exactly
__device__ __forceinline__ void fixed_scale40(float m, double& s, double& inv) {
  int e = ((__float_as_int(m) >> 23) & 0xff) - 126;          // m < 2^e
  e = min(max(e, -100), 100);
  s = __longlong_as_double((long long)(1023 + 40 - e) << 52);
  inv = __longlong_as_double((long long)(1023 - 40 + e) << 52);
}
// rint(x) for |x| < 2^51 as a 64-bit integer: the integer sits in the low mantissa bits of x + 1.5 * 2^52 (gfx950 has no f64 -> i64 convert)
__device__ __forceinline__ long long to_fixed(double x) {
  return __double_as_longlong(x + 6755399441055744.0) - 0x4338000000000000ll;
}

__global__ void __launch_bounds__(kThreads) texel_reduce_kernel(const float* __restrict__ rec, int32_t* __restrict__ cnt, int cap, int nbins, int nbx,
                                                                int Wt, int Ht, double* __restrict__ g_tex, double* __restrict__ g_nmap) {
  __shared__ long long acc[2][kAcc * kAcc * 3];   // [map][(ly * 33 + lx) * 3 + channel], 64-bit fixed point (see the scales below)
  __shared__ float wmax[2][kThreads / 64];
  __shared__ int pre[kMaxBins];                // inclusive prefix of the tiles' chunk counts
  __shared__ int cl[kMaxBins];                 // the tiles' record counts 
  __shared__ int wsum[kThreads / 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  // ---- chunk counts of all tiles -> inclusive prefix (every workgroup forms it: nbins loads)
  int carry = 0;
  for (int b0 = 0; b0 < nbins; b0 += kThreads) {
    const int b = b0 + t;
    const int c = b < nbins ? min(cnt[16 * b], cap) : 0;
    int v = (c + kChunk - 1) / kChunk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int off = carry;
    for (int i = 0; i < w; ++i) off += wsum[i];
    if (b < nbins) { pre[b] = v + off; cl[b] = c; }
    int tot = 0;
    for (int i = 0; i < kThreads / 64; ++i) tot += wsum[i];
    carry += tot;
    __syncthreads();
  }
  const int total = carry;
  for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
    // tile of this chunk: first b with pre[b] > chunk
    int lo = 0, hi = nbins - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pre[mid] > chunk) hi = mid; else lo = mid + 1; }
    const int bin = lo;
    const int first = (chunk - (bin ? pre[bin - 1] : 0)) * kChunk;
    const int n = min(cl[bin] - first, kChunk);
    for (int i = t; i < 2 * kAcc * kAcc * 3; i += kThreads) (&acc[0][0])[i] = 0;
    const float* r0 = rec + (size_t)bin * 9 * (size_t)cap + (size_t)first;
    const size_t capz = (size_t)cap;
    // thread t owns records 4 t .. 4 t + 3 of the chunk: one 16-byte load per plane, and — what matters — the lanes of a wave instruction
    // are then 4 records apart: x-neighbouring pixels of a frame that share a texel (half of 64 consecutive records are such duplicates,
    // up to 9 lanes on one texel) meet in ONE thread, one after the other, instead of in one LDS atomic instruction
    float rv[kChunk / kThreads][9];
    static_assert(kChunk / kThreads == 4, "one float4 per plane and thread");
    {
      const int i = 4 * t;
      if (i < n) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float4 q = *(const float4*)(r0 + k * capz + i);   // (slots past n: stale, never used)
          rv[0][k] = q.x; rv[1][k] = q.y; rv[2][k] = q.z; rv[3][k] = q.w;
        }
      }
    }
    // 64-bit FIXED-POINT accumulators (ds_add_u64: 6.9 clk per wave instruction on gfx950 against 8.7 for ds_add_f64, which also pays ~11 clk
    // for every further lane on the same address — and neighbouring pixels of a frame DO share texels): one power-of-two scale per map and
    // chunk from the chunk's largest gradient component, |x| * s < 2^40; a texel takes at most one corner of each of the <= 2048 records
    // (weight <= 1): |sum| < 2^51.  Resolution 2^-40 of the chunk's largest component: every contribution down to 2^-16 of it keeps its whole
    // float mantissa, and the sum is independent of the order of the records (a 32-bit table at 2^-20 was measured first: its rounding of the
    // small contributions depends on what else is in the chunk, which Adam's sign-like first steps turned into visible differences between a
    // 2-rank and a 1-rank fit of the same frames).
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int j = 0; j < kChunk / kThreads; ++j) {
      if (4 * t + j < n) {
        m0 = fmaxf(m0, fmaxf(fabsf(rv[j][3]), fmaxf(fabsf(rv[j][4]), fabsf(rv[j][5]))));
        m1 = fmaxf(m1, fmaxf(fabsf(rv[j][6]), fmaxf(fabsf(rv[j][7]), fabsf(rv[j][8]))));
      }
    }
    m0 = wave_max_u(m0); m1 = wave_max_u(m1);
    if (lane == 0) { wmax[0][w] = m0; wmax[1][w] = m1; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kThreads / 64; ++i) { m0 = fmaxf(m0, wmax[0][i]); m1 = fmaxf(m1, wmax[1][i]); }
    double s0, i0, s1, i1;
    fixed_scale40(m0, s0, i0);
    fixed_scale40(m1, s1, i1);
#pragma unroll
    for (int j = 0; j < kChunk / kThreads; ++j) {
      const int i = 4 * t + j;
      if (i < n) {
        const int key = __float_as_int(rv[j][0]);
        const int x0 = key & 0xffff, y0 = (int)((unsigned)key >> 16);
        const float wx = rv[j][1], wy = rv[j][2];
        const float ax = 1.f - wx, ay = 1.f - wy;
        const float cw[4] = {ax * ay, wx * ay, ax * wy, wx * wy};
        const int lx = x0 & (kBin - 1), ly = y0 & (kBin - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
          if (cx < Wt && cy < Ht && cw[k] != 0.f) {
            const int o = ((ly + (k >> 1)) * kAcc + lx + (k & 1)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (g_tex) atomicAdd((unsigned long long*)&acc[0][o + c], (unsigned long long)to_fixed((double)(rv[j][3 + c] * cw[k]) * s0));
              if (g_nmap) atomicAdd((unsigned long long*)&acc[1][o + c], (unsigned long long)to_fixed((double)(rv[j][6 + c] * cw[k]) * s1));
            }
          }
        }
      }
    }
    __syncthreads();
    // flush, lanes = (texel, channel) along a tile row: 99 consecutive floats per row
    const int bx = bin % nbx, by = bin / nbx;
    for (int i = t; i < kAcc * kAcc * 3; i += kThreads) {
      const int ly = i / (kAcc * 3), q = i - ly * (kAcc * 3);
      const int gx3 = bx * kBin * 3 + q, gy = by * kBin + ly;
      if (gy < Ht && gx3 < Wt * 3) {
        const size_t o = (size_t)gy * Wt * 3 + gx3;
        const long long a0 = acc[0][i], a1 = acc[1][i];
        if (g_tex && a0 != 0) atomicAdd(g_tex + o, (double)a0 * i0);          // (exact product: |a0| < 2^51, i0 a power of two)
        if (g_nmap && a1 != 0) atomicAdd(g_nmap + o, (double)a1 * i1);
      }
    }
    __syncthreads();
  }
}

// acc (n_texels, 3) doubles -> g (+=, float), acc cleared; with nmap_raw the normal map's chain rule rides along: acc is the gradient of the
// NORMALISED map F.normalize(nmap_raw, dim=-1) (utils/visualize.py:99) and g that of the raw one (harp_normalize3_bwd's arithmetic)
__global__ void __launch_bounds__(256) texel_finish_kernel(double* __restrict__ acc_tex, float* __restrict__ g_tex, double* __restrict__ acc_nmap,
                                                           float* __restrict__ g_nmap, const float* __restrict__ nmap_raw, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (acc_tex) {
    const double a = acc_tex[3 * (size_t)i], b = acc_tex[3 * (size_t)i + 1], c = acc_tex[3 * (size_t)i + 2];
    if (a != 0.0 || b != 0.0 || c != 0.0) {
      g_tex[3 * (size_t)i] += (float)a; g_tex[3 * (size_t)i + 1] += (float)b; g_tex[3 * (size_t)i + 2] += (float)c;
      acc_tex[3 * (size_t)i] = 0.0; acc_tex[3 * (size_t)i + 1] = 0.0; acc_tex[3 * (size_t)i + 2] = 0.0;
    }
  }
  if (acc_nmap) {
    const double a = acc_nmap[3 * (size_t)i], b = acc_nmap[3 * (size_t)i + 1], c = acc_nmap[3 * (size_t)i + 2];
    if (a != 0.0 || b != 0.0 || c != 0.0) {
      acc_nmap[3 * (size_t)i] = 0.0; acc_nmap[3 * (size_t)i + 1] = 0.0; acc_nmap[3 * (size_t)i + 2] = 0.0;
      if (nmap_raw) {
        normalize3_bwd_vals(nmap_raw, (float)a, (float)b, (float)c, g_nmap, (size_t)i);
      } else {
        g_nmap[3 * (size_t)i] += (float)a; g_nmap[3 * (size_t)i + 1] += (float)b; g_nmap[3 * (size_t)i + 2] += (float)c;
      }
    }
  }
}

__global__ void texel_counters_clear_kernel(int32_t* __restrict__ cnt, int nbins) {
  for (int b = threadIdx.x; b < nbins; b += blockDim.x) cnt[16 * b] = 0;
}

}  // namespace

extern "C" {

int harp_texel_bins(int Ht, int Wt) { return (Ht <= 0 || Wt <= 0) ? 0 : ((Ht + kBin - 1) / kBin) * ((Wt + kBin - 1) / kBin); }

int harp_texel_reduce(const float* trec, int32_t* trec_cnt, int trec_cap, int Ht, int Wt, double* acc_tex, double* acc_nmap, int expected_records,
                      hipStream_t stream) {
  const int nbins = harp_texel_bins(Ht, Wt);
  if (!trec || !trec_cnt || trec_cap <= 0 || (trec_cap & 3) || ((size_t)trec & 15) || nbins <= 0 || nbins > kMaxBins || Wt > 65535 || Ht > 65535) return HARP_ERR_ARG;
  // Launch shape.  The kernel runs beside the mesh / hand backward tail, whose workgroups (76 KB of LDS each) must find room on every CU:
  //   small job (expected_records <= 2 M, or unknown: the C3 step's 1.4 M records are ~680 chunks): 256 workgroups, ONE per CU (61 KB + 21.5 KB of
  //     dynamic LDS it does not use: a second one does not fit, a chain workgroup does) — chain_wide_bwd_b 21 -> 14 us, step -3.5 us
  //   large job (C5: 4.9 M records, ~2 600 chunks): 512 workgroups, two per CU (12 KB pad: a third does not fit)
  const bool small_job = expected_records <= 2000000;
  const unsigned grid = small_job ? 256u : 512u;
  const size_t pad = small_job ? 21504 : 12288;
  hipLaunchKernelGGL(texel_reduce_kernel, dim3(grid), dim3(kThreads), pad, stream, trec, trec_cnt, trec_cap, nbins, (Wt + kBin - 1) / kBin, Wt, Ht, acc_tex, acc_nmap);
  // (a "last workgroup clears" ticket was measured first: the 512 same-address returning atomics put ~15 us under every workgroup's record loads)
  hipLaunchKernelGGL(texel_counters_clear_kernel, dim3(1), dim3(256), 0, stream, trec_cnt, nbins);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_texel_finish(double* acc_tex, float* g_tex, double* acc_nmap, float* g_nmap, const float* nmap_raw, int n_texels, hipStream_t stream) {
  if (n_texels <= 0 || (acc_tex && !g_tex) || (acc_nmap && !g_nmap) || (!acc_tex && !acc_nmap)) return HARP_ERR_ARG;
  hipLaunchKernelGGL(texel_finish_kernel, dim3((n_texels + 255) / 256), dim3(256), 0, stream, acc_tex, g_tex, acc_nmap, g_nmap, nmap_raw, n_texels);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
