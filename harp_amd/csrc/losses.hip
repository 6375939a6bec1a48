#include <stdlib.h>
// Loss terms of the HARP fitting loop (optimize_sequence.py:517-553) with their gradients, texture-map helpers and
// the fused Adam update, for gfx950.  Each kernel evaluates a term AND, when a device weight array is given, its
// gradient (weight = d total / d term, i.e. the loss weight of optimize_sequence.py:411-422 in the fused engine or
// the autograd upstream scalar in the op-level API), so forward and backward share one pass over the data.
#include "harp_common.h"
#include "harp_hip.h"

namespace {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

// L1Loss(mean) between pred*mask and target*mask, C channels, mask broadcast over channels (optimize_sequence.py:519,543).
// frame b of pred pairs with frame fid[b] of target/mask (fid NULL => identity).  n_per_frame = S*S*C.
__global__ void __launch_bounds__(256) image_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       const float* __restrict__ mask, const int32_t* __restrict__ fid,
                                                       int n_per_frame, int C, float inv_count, const float* __restrict__ w,
                                                       float* __restrict__ loss, float* __restrict__ g_pred) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int tb = fid ? fid[b] : b;
  const float* p = pred + (size_t)b * n_per_frame;
  const float* t = target + (size_t)tb * n_per_frame;
  const float* m = mask ? mask + (size_t)tb * (n_per_frame / C) : nullptr;
  const float wk = w ? w[0] * inv_count : 0.f;
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_per_frame; i += gridDim.x * 256) {
    const float mm = m ? m[i / C] : 1.f;
    const float d = p[i] * mm - t[i] * mm;
    acc += fabsf(d);
    if (g_pred) g_pred[(size_t)b * n_per_frame + i] = wk * sgn(d) * mm;
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss, s * inv_count);
}

// kps_loss (loss/kps_loss.py:4-17): mean_{b,j} (|| (gt-gt0) - 1000 (pred-pred0) ||/100)^2 over the first 21 joints.
__device__ __forceinline__ void kps_body(const float* __restrict__ gt, const int32_t* __restrict__ fid, const float* __restrict__ pred,
                                         int b, int j, int B, int NJp, const float* __restrict__ w, float* __restrict__ loss,
                                         float* __restrict__ g_pred) {
  const float* g = gt + (size_t)(fid ? fid[b] : b) * 63;
  const float* p = pred + (size_t)b * NJp * 3;
  float d[3] = {0.f, 0.f, 0.f}, l = 0.f;
  if (j < 21) {
    for (int c = 0; c < 3; ++c) {
      d[c] = (g[3 * j + c] - g[c]) - (p[3 * j + c] - p[c]) * 1000.0f;
      l += d[c] * d[c];
    }
    l = sqrtf(l) / 100.0f;
    l = l * l;
  }
  const float inv = 1.0f / (float)(B * 21);
  const float s = wave_sum(l);
  if (j == 0) atomicAdd(loss, s * inv);
  if (w && g_pred) {
    const float k = w[0] * inv * -2000.0f / 10000.0f;
    float c3[3];
    for (int c = 0; c < 3; ++c) {
      c3[c] = (j < 21) ? k * d[c] : 0.f;
      const float tot = wave_sum(c3[c]);
      if (j > 0 && j < 21) g_pred[((size_t)b * NJp + j) * 3 + c] += c3[c];
      if (j == 0) g_pred[((size_t)b * NJp) * 3 + c] += c3[c] - tot;
    }
  }
}
__global__ void __launch_bounds__(64) kps_kernel(const float* __restrict__ gt, const int32_t* __restrict__ fid,
                                                 const float* __restrict__ pred, int B, int NJp, const float* __restrict__ w,
                                                 float* __restrict__ loss, float* __restrict__ g_pred) {
  kps_body(gt, fid, pred, blockIdx.x, threadIdx.x, B, NJp, w, loss, g_pred);
}
// the key-point term riding in the mesh regularisers' launch (grid slice z == 3: one wave per frame) instead of a launch of its own
struct KpsArgs { const float* gt; const int32_t* fid; const float* pred; int NJp; const float* w; float* loss; float* g_pred; };

// laplacian (uniform, Appendix A.11) | normal consistency (A.12) | ARAP (loss/arap.py:45-57); blockIdx.y = frame.
// GATHER formulation, one lane per (frame, vertex), no atomics on the gradient: every lane re-derives the terms its vertex
// takes part in (its own Laplacian row + its neighbours' rows, its incident edges, the face pairs it belongs to) from the
// static CSR tables, and owns g_verts[b, u, :].  w[0..2] = weights of (laplacian, normal, arap); loss[0..2] accumulate.
// STAGE: the frame's vertices are first copied to LDS (V*12 B, dynamic) and every neighbour / pair gather reads from there: a lane
// walks ~100 dependent gathers and with only ~1.5 waves per SIMD in flight the L2 round trips were the whole cost (53 -> 15 us).
// BS: workgroup size.  A workgroup stages the whole frame (V * 12 B) for BS vertices' worth of gathers: 256 threads = 13 workgroups per
// (frame, term) for the hand mesh, each copying 37 KB; 1024 threads = 4.
template <int BS>
__device__ __forceinline__ float block_sum_bs(float v, float* red16) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < BS / 64; ++i) r += red16[i];
  __syncthreads();
  return r;
}
template <bool STAGE, int BS = 256>
__global__ void __launch_bounds__(BS) mesh_reg_kernel(const float* __restrict__ verts, const float* __restrict__ ref_verts,
                                                       const int32_t* __restrict__ nbr_off, const int32_t* __restrict__ nbr_idx,
                                                       const int32_t* __restrict__ pairs, const int32_t* __restrict__ vp_off,
                                                       const int32_t* __restrict__ vp_idx, int B, int V, int P, int E,
                                                       const float* __restrict__ w, float* __restrict__ loss,
                                                       float* __restrict__ g_verts, const KpsArgs K) {
  __shared__ float red[16];
  extern __shared__ float s_verts[];
  if (blockIdx.z == 3) {                      // (only launched when K.gt is set)
    if (blockIdx.x == 0 && threadIdx.x < 64) kps_body(K.gt, K.fid, K.pred, blockIdx.y, threadIdx.x, B, K.NJp, K.w, K.loss, K.g_pred);
    return;
  }
  const int b = blockIdx.y, u = blockIdx.x * BS + threadIdx.x;
  const float* vb = verts + (size_t)b * V * 3;
  if (STAGE) {
    for (int i = threadIdx.x; i < V * 3; i += BS) s_verts[i] = vb[i];
    __syncthreads();
    vb = s_verts;
  }
  const bool grad = (w != nullptr) && (g_verts != nullptr);
  // blockIdx.z picks the term (0 Laplacian, 1 ARAP, 2 normal consistency): three times the waves in flight and a third of the
  // dependent gather chain per lane; the three partial gradients meet in g_verts through (coalesced) float atomics.
  const int term = blockIdx.z;
  const float w_lap = grad ? w[0] : 0.f, w_nc = grad ? w[1] : 0.f, w_ar = grad ? w[2] : 0.f;
  float l_lap = 0.f, l_nc = 0.f, l_ar = 0.f;
  float g[3] = {0.f, 0.f, 0.f};
  if (u < V) {
    const float pu[3] = {vb[3 * u], vb[3 * u + 1], vb[3 * u + 2]};
    const float sc_lap = 1.0f / ((float)V * (float)B), sc_ar = 1.0f / ((float)E * (float)B), sc_nc = 1.0f / ((float)P * (float)B);
    const int s = nbr_off[u], e = nbr_off[u + 1];
    // --- own Laplacian row
    if (term == 0) {
      float a[3] = {0.f, 0.f, 0.f};
      for (int k = s; k < e; ++k) for (int c = 0; c < 3; ++c) a[c] += vb[3 * nbr_idx[k] + c];
      const float invd = 1.0f / (float)(e - s);
      float lv[3], n2 = 0.f;
      for (int c = 0; c < 3; ++c) { lv[c] = a[c] * invd - pu[c]; n2 += lv[c] * lv[c]; }
      const float n = sqrtf(n2);
      l_lap = n * sc_lap;
      if (n > 0.f) for (int c = 0; c < 3; ++c) g[c] -= w_lap * sc_lap / n * lv[c];
    }
    if (term < 2) for (int k = s; k < e; ++k) {
      const int nb = nbr_idx[k];
      const float pn[3] = {vb[3 * nb], vb[3 * nb + 1], vb[3 * nb + 2]};
      // --- neighbour's Laplacian row contains u with weight 1/deg(nb)
      if (grad && term == 0) {
        const int s2 = nbr_off[nb], e2 = nbr_off[nb + 1];
        float a[3] = {0.f, 0.f, 0.f};
        for (int q = s2; q < e2; ++q) for (int c = 0; c < 3; ++c) a[c] += vb[3 * nbr_idx[q] + c];
        const float invd = 1.0f / (float)(e2 - s2);
        float lv[3], n2 = 0.f;
        for (int c = 0; c < 3; ++c) { lv[c] = a[c] * invd - pn[c]; n2 += lv[c] * lv[c]; }
        const float n = sqrtf(n2);
        if (n > 0.f) for (int c = 0; c < 3; ++c) g[c] += w_lap * sc_lap / n * lv[c] * invd;
      }
      // --- ARAP on edge (u, nb); every edge is visited from both ends -> half the loss each
      if (ref_verts && term == 1) {
        float d[3], l2 = 0.f, r2 = 0.f;
        for (int c = 0; c < 3; ++c) {
          d[c] = pu[c] - pn[c];
          l2 += d[c] * d[c];
          const float r = ref_verts[3 * u + c] - ref_verts[3 * nb + c];
          r2 += r * r;
        }
        const float l = sqrtf(l2);
        const float diff = l * 1000.0f - sqrtf(r2) * 1000.0f;
        l_ar += 0.5f * diff * diff * sc_ar;
        if (l > 0.f) for (int c = 0; c < 3; ++c) g[c] += w_ar * sc_ar * 2.0f * diff * 1000.0f / l * d[c];
      }
    }
    // --- normal consistency: the face pairs u belongs to (role 0 = v0, 1 = v1, 2 = a, 3 = b)
    if (term == 2) for (int k = vp_off[u]; k < vp_off[u + 1]; ++k) {
      const int pr = vp_idx[k] >> 2, role = vp_idx[k] & 3;
      const int i0 = pairs[4 * pr], i1 = pairs[4 * pr + 1], ia = pairs[4 * pr + 2], ib = pairs[4 * pr + 3];
      float ev[3], da[3], db[3];
      for (int c = 0; c < 3; ++c) { ev[c] = vb[3 * i1 + c] - vb[3 * i0 + c]; da[c] = vb[3 * ia + c] - vb[3 * i0 + c]; db[c] = vb[3 * ib + c] - vb[3 * i0 + c]; }
      const float n0[3] = {ev[1] * da[2] - ev[2] * da[1], ev[2] * da[0] - ev[0] * da[2], ev[0] * da[1] - ev[1] * da[0]};
      const float n1[3] = {-(ev[1] * db[2] - ev[2] * db[1]), -(ev[2] * db[0] - ev[0] * db[2]), -(ev[0] * db[1] - ev[1] * db[0])};
      const float l0 = sqrtf(n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]), l1 = sqrtf(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
      const float dp = n0[0] * n1[0] + n0[1] * n1[1] + n0[2] * n1[2];
      const float den = fmaxf(l0 * l1, 1e-8f);
      const float cs = dp / den;
      if (role == 0) l_nc += (1.0f - cs) * sc_nc;
      if (grad) {
        const float kk = -w_nc * sc_nc;                   // d/d cos
        // torch 1.11 cosine_similarity: w12 / sqrt(clamp_min(w1*w2, eps^2)); the clamp is ACTIVE for mm-sized triangles
        // (|n0||n1| ~ 1e-11 < 1e-8), where cos = w12 / eps and only the numerator carries gradient.
        const bool clamped = !(l0 * l1 > 1e-8f);
        float g0[3], g1[3];
        for (int c = 0; c < 3; ++c) {
          g0[c] = kk * (n1[c] / den - (clamped ? 0.f : cs * n0[c] / (l0 * l0)));
          g1[c] = kk * (n0[c] / den - (clamped ? 0.f : cs * n1[c] / (l1 * l1)));
        }
        // n0 = e x da : g_e = da x g0, g_da = g0 x e ;  n1 = -(e x db): g_e += -(db x g1), g_db = -(g1 x e)
        const float ge[3] = {da[1] * g0[2] - da[2] * g0[1] - (db[1] * g1[2] - db[2] * g1[1]),
                             da[2] * g0[0] - da[0] * g0[2] - (db[2] * g1[0] - db[0] * g1[2]),
                             da[0] * g0[1] - da[1] * g0[0] - (db[0] * g1[1] - db[1] * g1[0])};
        const float gda[3] = {g0[1] * ev[2] - g0[2] * ev[1], g0[2] * ev[0] - g0[0] * ev[2], g0[0] * ev[1] - g0[1] * ev[0]};
        const float gdb[3] = {-(g1[1] * ev[2] - g1[2] * ev[1]), -(g1[2] * ev[0] - g1[0] * ev[2]), -(g1[0] * ev[1] - g1[1] * ev[0])};
        for (int c = 0; c < 3; ++c)
          g[c] += (role == 1) ? ge[c] : (role == 2) ? gda[c] : (role == 3) ? gdb[c] : (-ge[c] - gda[c] - gdb[c]);
      }
    }
    if (grad) {
      float* o = g_verts + ((size_t)b * V + u) * 3;
      atomicAdd(o, g[0]); atomicAdd(o + 1, g[1]); atomicAdd(o + 2, g[2]);
    }
  }
  const float s0 = block_sum_bs<BS>(l_lap, red), s1 = block_sum_bs<BS>(l_nc, red), s2 = block_sum_bs<BS>(l_ar, red);
  if (threadIdx.x == 0) {
    if (s0 != 0.f) atomicAdd(loss, s0);
    if (s1 != 0.f) atomicAdd(loss + 1, s1);
    if (s2 != 0.f) atomicAdd(loss + 2, s2);
  }
}

// sum(d^2) (optimize_sequence.py:533)
__device__ __forceinline__ void sumsq_body(int bid, int nb, const float* __restrict__ d, int n, const float* __restrict__ w,
                                           float* __restrict__ loss, float* __restrict__ g, float* red) {
  float acc = 0.f;
  for (int i = bid * 256 + threadIdx.x; i < n; i += nb * 256) {
    acc += d[i] * d[i];
    if (w && g) atomicAdd(&g[i], 2.0f * w[0] * d[i]);
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss, s);
}
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ d, int n, const float* __restrict__ w,
                                                    float* __restrict__ loss, float* __restrict__ g) {
  __shared__ float red[4];
  sumsq_body(blockIdx.x, gridDim.x, d, n, w, loss, g, red);
}

// torch.nn.MSELoss()(x, y) and its gradient w.r.t. x (metro_modifications/hand_utils.py:71-87): the per-iteration objective of the
// MANO-to-METRO fit.  g is overwritten (not accumulated): it is the only term of that loop.
__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float* __restrict__ loss,
                                                  float* __restrict__ g) {
  __shared__ float red[4];
  const float inv_n = 1.0f / (float)n;
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float d = x[i] - y[i];
    acc += d * d;
    if (g) g[i] = 2.0f * d * inv_n;
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss, s * inv_n);
}

// albedo_reg / smooth_texture_reg (loss/texture_reg.py:5-30, 48-66): mean_xy( ||t[x,y]-t[x+dx,y+dy]||_1 / 3 * mask )
__device__ __forceinline__ void tex_smooth_body(int bid, int nb, const float* __restrict__ t, const int32_t* __restrict__ dist,
                                                const float* __restrict__ mask, int H, int W, const float* __restrict__ w,
                                                float* __restrict__ loss, float* __restrict__ g, float* red) {
  float acc = 0.f;
  for (int i = bid * 256 + threadIdx.x; i < H * W; i += nb * 256) {
    const int x = i / W, y = i % W;                    // reference names: x = row, y = column
    const int tx = min(max(x + dist[2 * i], 0), H - 1), ty = min(max(y + dist[2 * i + 1], 0), W - 1);
    const int j = tx * W + ty;
    const float m = mask ? mask[i] : 1.f;
    const float k = (w ? w[0] : 0.f) * m / (3.0f * (float)(H * W));
    float a1 = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float d = t[3 * i + c] - t[3 * j + c];
      a1 += fabsf(d);
      // (texels outside the UV mask carry k == 0: no gradient traffic for them)
      if (w && g && d != 0.f && k != 0.f) { atomicAdd(g + 3 * i + c, k * sgn(d)); atomicAdd(g + 3 * j + c, -k * sgn(d)); }
    }
    acc += a1 / 3.0f * m;
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss, s / (float)(H * W));
}
__global__ void __launch_bounds__(256) tex_smooth_kernel(const float* __restrict__ t, const int32_t* __restrict__ dist,
                                                         const float* __restrict__ mask, int H, int W, const float* __restrict__ w,
                                                         float* __restrict__ loss, float* __restrict__ g) {
  __shared__ float red[4];
  tex_smooth_body(blockIdx.x, gridDim.x, t, dist, mask, H, W, w, loss, g, red);
}

// The same term by TILE OWNERS (the form the fused launch below runs): a workgroup owns a 32x32-texel tile and forms the complete
// gradient of its texels — the own part k sgn(t[p] - t[j(p)]) and what every source s with j(s) = p hands over, -k_s sgn(t[s] - t[p]) —
// in registers, then adds it to g with ONE row-contiguous memory atomic per texel and channel, and only where the UV mask is set.  The
// stand-alone form above scatters two memory atomics per texel and channel to random addresses (3 M per step at 21 G/s: the kernel at the
// head of the step's second stream was bound by exactly that).  How a target finds its sources: the offsets are int(N(0, 1 | 2)), so all
// but ~1e-5 of the sources lie within 8 texels of their target; the workgroup stages texels and mask of its tile + an 8-texel halo in LDS,
// walks the staged sources, and every source whose (clamped) target lies in the tile appends itself to the target's list in LDS (4 slots,
// ds_add_rtn_u32; mean occupancy 1).  The rest keep today's path: a source whose target is further than the halo reaches (decided by the
// source's own tile, which alone knows it) or whose target's list is full adds its three values to the target with memory atomics.
constexpr int kST = 32, kSH = 8, kSR = kST + 2 * kSH, kSL = 4;
constexpr int kSE = (kSR * kSR + 255) / 256;       // staged texels per thread (9)
struct SmoothTileSmem {
  float t[kSR * kSR * 3];
  float m[kSR * kSR];
  int cnt[kST * kST];
  int tj[kST * kST];                               // (clamped) target of the tile's own texels, row << 16 | column
  unsigned short list[kST * kST][kSL];
  float red[4];
};
__device__ __forceinline__ void tex_smooth_tile_body(int tile, const float* __restrict__ t, const int32_t* __restrict__ dist, const float* __restrict__ mask,
                                                     int H, int W, const float* __restrict__ w, float* __restrict__ loss, float* __restrict__ g,
                                                     SmoothTileSmem& S) {
  const int ntc = (W + kST - 1) / kST;
  const int tr = tile / ntc, tc = tile - tr * ntc;
  const int r0 = tr * kST - kSH, c0 = tc * kST - kSH;          // origin of the staged region (may lie outside the image)
  // every thread owns kSE fixed texels of the staged region; its loads are issued together (three round trips per workgroup in all:
  // mask | texels + offsets | — rarely — a target outside the region)
  int gi[kSE];
  float mv[kSE];
  // ---- mask of tile + halo; a region without a single masked texel has no loss and no gradient (3/5 of the tiles of the hand atlas)
  float mmax = 0.f;
#pragma unroll
  for (int e = 0; e < kSE; ++e) {
    const int i = threadIdx.x + 256 * e;
    const int rr = r0 + i / kSR, cc = c0 + i % kSR;
    const bool in = i < kSR * kSR && rr >= 0 && rr < H && cc >= 0 && cc < W;
    gi[e] = in ? rr * W + cc : -1;
    mv[e] = in ? (mask ? mask[gi[e]] : 1.f) : 0.f;
  }
#pragma unroll
  for (int e = 0; e < kSE; ++e) {
    const int i = threadIdx.x + 256 * e;
    if (i < kSR * kSR) S.m[i] = mv[e];
    mmax = fmaxf(mmax, fabsf(mv[e]));
  }
  for (int i = threadIdx.x; i < kST * kST; i += 256) S.cnt[i] = 0;
  if (__syncthreads_or(mmax != 0.f) == 0) return;
  float tx[kSE][3];
  int dr[kSE], dc[kSE];
#pragma unroll
  for (int e = 0; e < kSE; ++e) {
    const bool in = gi[e] >= 0;
    const float* q = t + (size_t)(in ? gi[e] : 0) * 3;
    tx[e][0] = in ? q[0] : 0.f; tx[e][1] = in ? q[1] : 0.f; tx[e][2] = in ? q[2] : 0.f;
    const bool on = in && mv[e] != 0.f;                  // (a source outside the mask: no loss, nothing handed over — its offsets are not needed)
    const int2 d2 = on ? *reinterpret_cast<const int2*>(dist + 2 * (size_t)gi[e]) : make_int2(0, 0);
    dr[e] = d2.x; dc[e] = d2.y;
  }
#pragma unroll
  for (int e = 0; e < kSE; ++e) {
    const int i = threadIdx.x + 256 * e;
    if (i < kSR * kSR) { S.t[3 * i] = tx[e][0]; S.t[3 * i + 1] = tx[e][1]; S.t[3 * i + 2] = tx[e][2]; }
  }
  __syncthreads();
  const float kk = (w ? w[0] : 0.f) / (3.0f * (float)(H * W));
  const bool grad = w != nullptr && g != nullptr;
  // texel (rr, cc) of the image: from the staged region if it lies there, else from memory
  auto fetch = [&](int rr, int cc, float* o) {
    const int lr = rr - r0, lc = cc - c0;
    if (lr >= 0 && lr < kSR && lc >= 0 && lc < kSR) { const float* q = &S.t[3 * (lr * kSR + lc)]; o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; }
    else { const float* q = t + (size_t)(rr * W + cc) * 3; o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; }
  };
  // ---- sources: every staged texel inside the mask
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < kSE; ++e) {
    const float ms = mv[e];
    if (gi[e] < 0 || ms == 0.f) continue;
    const int i = threadIdx.x + 256 * e;
    const int lr = i / kSR, lc = i % kSR, rr = r0 + lr, cc = c0 + lc;
    const bool own = lr >= kSH && lr < kSH + kST && lc >= kSH && lc < kSH + kST;
    const int jr = min(max(rr + dr[e], 0), H - 1), jc = min(max(cc + dc[e], 0), W - 1);
    const int ttr = jr / kST, ttc = jc / kST;          // the target's tile
    const bool mine = ttr == tr && ttc == tc;
    float tj[3];
    if (own || !mine) fetch(jr, jc, tj);               // (a halo source that targets this tile is read by the target, below)
    if (own) {
      acc += (fabsf(tx[e][0] - tj[0]) + fabsf(tx[e][1] - tj[1]) + fabsf(tx[e][2] - tj[2])) / 3.0f * ms;
      S.tj[(lr - kSH) * kST + lc - kSH] = (jr << 16) | jc;
    }
    if (!grad) continue;
    // does the target's owner see this source?  (its staged region: the target tile + halo)
    const bool reach = rr >= ttr * kST - kSH && rr < ttr * kST + kST + kSH && cc >= ttc * kST - kSH && cc < ttc * kST + kST + kSH;
    bool direct = own && !reach;                       // an outlier draw: only the source's own tile knows about it
    if (mine) {
      const int tl = (jr - tr * kST) * kST + (jc - tc * kST);
      const int slot = atomicAdd(&S.cnt[tl], 1);
      if (slot < kSL) S.list[tl][slot] = (unsigned short)i;
      else { direct = true; fetch(jr, jc, tj); }       // the target's list is full
    }
    if (direct) {
      const float k = kk * ms;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = tx[e][c] - tj[c];
        if (d != 0.f) atomicAdd(g + 3 * (size_t)(jr * W + jc) + c, -k * sgn(d));
      }
    }
  }
  __syncthreads();
  // ---- targets: the tile's texels
  if (grad) {
#pragma unroll
    for (int e = 0; e < kST * kST / 256; ++e) {
      const int q = threadIdx.x + 256 * e;
      const int pr = q / kST, pc = q % kST, rr = tr * kST + pr, cc = tc * kST + pc;
      if (rr >= H || cc >= W) continue;
      const int i = (pr + kSH) * kSR + pc + kSH;
      const float tp[3] = {S.t[3 * i], S.t[3 * i + 1], S.t[3 * i + 2]};
      float tot[3] = {0.f, 0.f, 0.f};
      const float kp = kk * S.m[i];
      if (kp != 0.f) {
        const int pj = S.tj[q];
        float tj[3];
        fetch(pj >> 16, pj & 0xffff, tj);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = tp[c] - tj[c]; if (d != 0.f) tot[c] += kp * sgn(d); }
      }
      const int n = min(S.cnt[q], kSL);
      for (int u = 0; u < n; ++u) {
        const int si = S.list[q][u];
        const float ks = kk * S.m[si];
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = S.t[3 * si + c] - tp[c]; if (d != 0.f) tot[c] -= ks * sgn(d); }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) if (tot[c] != 0.f) atomicAdd(g + 3 * (size_t)(rr * W + cc) + c, tot[c]);
    }
  }
  const float s = block_sum_256(acc, S.red);
  if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss, s / (float)(H * W));
}

// close_to_z_reg (loss/texture_reg.py:40-45) as the reference computes it: L2 norm over the WIDTH axis of
// (nm - (0,0,1)) for every (row, channel), /3, mean over rows x channels. One block per row.
// ATOMIC: the gradient is added with atomics (the fused launch below runs this term next to the smoothness term, which scatters into
// the same gradient image)
template <bool ATOMIC>
__device__ __forceinline__ void close_to_z_body(int h, const float* __restrict__ nm, int H, int W, const float* __restrict__ w, float scale,
                                                float* __restrict__ loss, float* __restrict__ g, float* red, float* s_norm) {
  float a[3] = {0.f, 0.f, 0.f};
  for (int x = threadIdx.x; x < W; x += 256)
    for (int c = 0; c < 3; ++c) { const float d = nm[((size_t)h * W + x) * 3 + c] - (c == 2 ? 1.f : 0.f); a[c] += d * d; }
  for (int c = 0; c < 3; ++c) {
    const float s = block_sum_256(a[c], red);
    if (threadIdx.x == 0) s_norm[c] = sqrtf(s);
  }
  __syncthreads();
  const float inv = 1.0f / (3.0f * (float)(H * 3));
  if (threadIdx.x == 0) atomicAdd(loss, scale * (s_norm[0] + s_norm[1] + s_norm[2]) * inv);
  if (w && g) {
    const float k = w[0] * scale * inv;
    for (int x = threadIdx.x; x < W; x += 256)
      for (int c = 0; c < 3; ++c)
        if (s_norm[c] > 0.f) {
          const float v = k * (nm[((size_t)h * W + x) * 3 + c] - (c == 2 ? 1.f : 0.f)) / s_norm[c];
          if (ATOMIC) atomicAdd(g + ((size_t)h * W + x) * 3 + c, v);
          else g[((size_t)h * W + x) * 3 + c] += v;
        }
  }
}
__global__ void __launch_bounds__(256) close_to_z_kernel(const float* __restrict__ nm, int H, int W, const float* __restrict__ w,
                                                         float scale, float* __restrict__ loss, float* __restrict__ g) {
  __shared__ float red[4];
  __shared__ float s_norm[3];
  close_to_z_body<false>(blockIdx.x, nm, H, W, w, scale, loss, g, red, s_norm);
}

// The parameter-only regularisers of an appearance + geometry step in ONE launch (they were four: albedo smoothness, close-to-z, normal-map
// smoothness, displacement sum of squares — 15 + 16 + 18 + 5 us of mostly launch latency on the second stream of a step): the grid is cut
// into four block ranges, one per term, each running the body of its stand-alone kernel.
struct TexTerms {
  const float *tex, *nmap, *mask, *disp;
  const int32_t *dist_a, *dist_n;
  int H, W, n_disp;
  float z_scale;
  const float *w_a, *w_n, *w_d;        // device weights (albedo | close-to-z and normal smoothness share one | displacement)
  float *l_a, *l_n, *l_d;
  float *g_tex, *g_nmap, *g_disp;
  int nb_smooth, nb_disp;
  int* bump;                           // optional: the draw counter the offsets were drawn for (advanced here: the offsets are consumed)
};
__global__ void __launch_bounds__(256) texture_terms_kernel(const TexTerms A) {
  __shared__ SmoothTileSmem tile;
  __shared__ float s_norm[3];
  float* red = tile.red;
  int bid = blockIdx.x;
  if (A.bump && bid == 0 && threadIdx.x == 0) A.bump[0] += 1;      // (harp_step_prologue, an earlier launch, drew with the old value)
  if (bid < A.nb_smooth) { tex_smooth_tile_body(bid, A.tex, A.dist_a, A.mask, A.H, A.W, A.w_a, A.l_a, A.g_tex, tile); return; }
  bid -= A.nb_smooth;
  if (bid < A.nb_smooth) { tex_smooth_tile_body(bid, A.nmap, A.dist_n, A.mask, A.H, A.W, A.w_n, A.l_n, A.g_nmap, tile); return; }
  bid -= A.nb_smooth;
  if (bid < A.H) { close_to_z_body<true>(bid, A.nmap, A.H, A.W, A.w_n, A.z_scale, A.l_n, A.g_nmap, red, s_norm); return; }
  bid -= A.H;
  if (A.disp) sumsq_body(bid, A.nb_disp, A.disp, A.n_disp, A.w_d, A.l_d, A.g_disp, red);
}

// F.normalize(normal_map, dim=-1) per texel (utils/visualize.py:99), eps 1e-12
__global__ void normalize3_fwd_kernel(const float* __restrict__ x, int n, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  normalize3_texel(x[3 * i], x[3 * i + 1], x[3 * i + 2], y[3 * i], y[3 * i + 1], y[3 * i + 2]);
}
__global__ void normalize3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, int n, float* __restrict__ gx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) normalize3_bwd_texel(x, gy, gx, (size_t)i);
}

// torch.optim.Adam (betas, eps, no weight decay, no amsgrad) on a flat segment; bias corrections computed on the host.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float step_size, float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);        // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

// The random neighbour offsets of albedo_reg / smooth_texture_reg (loss/texture_reg.py:15, 51): int(N(0, std)) per texel and axis,
// truncated toward zero like torch's .to(torch.int).  Counter-based (hash of seed, draw counter, element index) + Box-Muller, so
// every rank draws the same offsets and a captured graph produces fresh ones on each replay (the counter lives in device memory).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void draw_pair(uint32_t seed, uint32_t c, int i, float std, int32_t* __restrict__ out) {
  const uint32_t h1 = mix32(seed ^ mix32(c * 0x9E3779B9U + 0x85EBCA6BU) ^ mix32((uint32_t)i * 2u + 1u));
  const uint32_t h2 = mix32(h1 ^ 0xC2B2AE35U ^ mix32((uint32_t)i * 2u + 2u + c));
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  const float z0 = r * cosf(6.28318530718f * u2), z1 = r * sinf(6.28318530718f * u2);
  out[2 * i] = (int32_t)(z0 * std);
  out[2 * i + 1] = (int32_t)(z1 * std);
}
__global__ void draw_offsets_kernel(uint32_t seed, int* __restrict__ counter, int n, float std, int32_t* __restrict__ out, float std2,
                                    int32_t* __restrict__ out2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = (uint32_t)counter[0];
  if (i < n) {
    draw_pair(seed, c, i, std, out);
    if (out2) draw_pair(seed ^ 0x5bd1e995U, c + 0x632BE5ABU, i, std2, out2);
  }
}
__global__ void bump_counter_kernel(int* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1; }

// device-resident hyper-parameters so that a captured hipGraph can be replayed while step / lr change
__global__ void adam_tick_kernel(harp_adam_hyper* hs, int count) {
  if (blockIdx.x == 0 && (int)threadIdx.x < count) {
    harp_adam_hyper* h = hs + threadIdx.x;
    h->step += 1;
    const double bc1 = 1.0 - pow((double)h->beta1, (double)h->step), bc2 = 1.0 - pow((double)h->beta2, (double)h->step);
    h->step_size = (float)((double)h->lr / bc1);
    h->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
}
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                size_t n, const harp_adam_hyper* __restrict__ h) {
  const float step_size = h->step_size, beta1 = h->beta1, beta2 = h->beta2, eps = h->eps, isb = h->inv_sqrt_bc2, gs = h->grad_scale;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gs;
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) * isb + eps));
  }
}

// The head of a step's second stream in ONE launch (it was a fill, adam_tick, draw_offsets and bump_counter: four launches of ~5 us each in
// front of the texture terms): clear the gradient slab, advance the optimisers' hyper-parameter structs, draw the texture offsets for the
// current value of the draw counter (which is advanced elsewhere: harp_step_frame's epilogue).
__global__ void __launch_bounds__(256) step_prologue_kernel(float* __restrict__ zero, size_t n_zero, harp_adam_hyper* hs, int n_hyper,
                                                            uint32_t seed, const int* __restrict__ counter, int n_draw, float std,
                                                            int32_t* __restrict__ out, float std2, int32_t* __restrict__ out2) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  if (zero) {
    const size_t head = min(n_zero, (size_t)((16 - ((uintptr_t)zero & 15)) & 15) / 4);       // floats up to the first 16-B boundary
    float4* z4 = reinterpret_cast<float4*>(zero + head);
    const size_t n4 = (n_zero - head) / 4;
    for (size_t i = tid; i < n4; i += nth) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < head) zero[tid] = 0.f;
    const size_t tail = head + 4 * n4;
    if (tid < n_zero - tail) zero[tail + tid] = 0.f;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_hyper) {
    harp_adam_hyper* h = hs + threadIdx.x;
    h->step += 1;
    const double bc1 = 1.0 - pow((double)h->beta1, (double)h->step), bc2 = 1.0 - pow((double)h->beta2, (double)h->step);
    h->step_size = (float)((double)h->lr / bc1);
    h->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
  if (out) {
    const uint32_t c = (uint32_t)counter[0];
    for (size_t i = tid; i < (size_t)n_draw; i += nth) {
      draw_pair(seed, c, (int)i, std, out);
      if (out2) draw_pair(seed ^ 0x5bd1e995U, c + 0x632BE5ABU, (int)i, std2, out2);
    }
  }
}

// two parameter groups (their own hyper-parameter structs, adjacent in memory) in ONE launch: elements [0, n0) of the grid map to
// [o0, o0 + n0), the rest to [o1, o1 + n1) of the same four arenas
__global__ void adam_dev2_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 size_t o0, size_t n0, size_t o1, size_t n1, const harp_adam_hyper* __restrict__ hs) {
  // four elements per lane (16-B loads / stores of all four arenas) when both spans are whole quads — the engine's arena segments are
  // 64-float aligned —, same per-element arithmetic
  if (((o0 | n0 | o1 | n1) & 3) == 0) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 4 * q < n0 + n1; q += (size_t)gridDim.x * blockDim.x) {
      const size_t k = 4 * q;
      const bool first = k < n0;
      const harp_adam_hyper* h = hs + (first ? 0 : 1);
      const size_t i = first ? o0 + k : o1 + (k - n0);
      const float beta1 = h->beta1, beta2 = h->beta2, gs = h->grad_scale, ss = h->step_size, isb = h->inv_sqrt_bc2, eps = h->eps;
      const float4 G = *reinterpret_cast<const float4*>(g + i);
      float4 M = *reinterpret_cast<float4*>(m + i), Vv = *reinterpret_cast<float4*>(v + i), P = *reinterpret_cast<float4*>(p + i);
      const float gq[4] = {G.x * gs, G.y * gs, G.z * gs, G.w * gs};
      float mq[4] = {M.x, M.y, M.z, M.w}, vq[4] = {Vv.x, Vv.y, Vv.z, Vv.w}, pq[4] = {P.x, P.y, P.z, P.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        mq[c] = mq[c] + (gq[c] - mq[c]) * (1.0f - beta1);
        vq[c] = vq[c] * beta2 + (1.0f - beta2) * gq[c] * gq[c];
        pq[c] -= ss * (mq[c] / (sqrtf(vq[c]) * isb + eps));
      }
      *reinterpret_cast<float4*>(m + i) = make_float4(mq[0], mq[1], mq[2], mq[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vq[0], vq[1], vq[2], vq[3]);
      *reinterpret_cast<float4*>(p + i) = make_float4(pq[0], pq[1], pq[2], pq[3]);
    }
    return;
  }
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n0 + n1; k += (size_t)gridDim.x * blockDim.x) {
    const bool first = k < n0;
    const harp_adam_hyper* h = hs + (first ? 0 : 1);
    const size_t i = first ? o0 + k : o1 + (k - n0);
    const float beta1 = h->beta1, beta2 = h->beta2;
    const float gi = g[i] * h->grad_scale;
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= h->step_size * (mi / (sqrtf(vi) * h->inv_sqrt_bc2 + h->eps));
  }
}

}  // namespace

extern "C" {

int harp_draw_texture_offsets(unsigned seed, int* counter_dev, int H, int W, float std, int32_t* dist, float std2, int32_t* dist2,
                              hipStream_t stream) {
  if (!counter_dev || !dist) return HARP_ERR_ARG;
  hipLaunchKernelGGL(draw_offsets_kernel, dim3((H * W + 255) / 256), dim3(256), 0, stream, seed, counter_dev, H * W, std, dist, std2, dist2);
  hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(64), 0, stream, counter_dev);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

static size_t side_lds() { const char* e = getenv("HARP_SIDE_LDS"); return e ? (size_t)atoi(e) : 512; }

int harp_step_prologue(float* zero, size_t n_zero, harp_adam_hyper* hyper, int n_hyper, unsigned seed, const int* draw_counter, int H, int W,
                       float std, int32_t* dist, float std2, int32_t* dist2, hipStream_t stream) {
  if ((n_zero && !zero) || (n_hyper && !hyper) || n_hyper < 0 || n_hyper > 64 || (dist && (!draw_counter || H <= 0 || W <= 0)) || (dist2 && !dist))
    return HARP_ERR_ARG;
  const size_t work = max(n_zero / 4, dist ? (size_t)H * W : (size_t)0);
  const int blocks = (int)max((size_t)1, min((size_t)2048, (work + 255) / 256));
  // (512 B of dynamic LDS it does not use: a workgroup with LDS cannot land on a CU whose LDS hand_front has claimed, hand_front.hip)
  hipLaunchKernelGGL(step_prologue_kernel, dim3(blocks), dim3(256), side_lds(), stream, n_zero ? zero : nullptr, n_zero, hyper, n_hyper, seed,
                     draw_counter, dist ? H * W : 0, std, dist, std2, dist2);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_adam_tick(harp_adam_hyper* h, int count, hipStream_t stream) {
  if (!h || count < 1 || count > 64) return HARP_ERR_ARG;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, h, count);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_adam_apply(float* p, const float* g, float* m, float* v, size_t n, const harp_adam_hyper* h, hipStream_t stream) {
  if (!p || !g || !m || !v || !h) return HARP_ERR_ARG;
  const int blocks = (int)min((size_t)2048, (n + 255) / 256);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, h);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_adam_apply2(float* p, const float* g, float* m, float* v, size_t o0, size_t n0, size_t o1, size_t n1, const harp_adam_hyper* h2,
                     hipStream_t stream) {
  if (!p || !g || !m || !v || !h2 || n0 + n1 == 0) return HARP_ERR_ARG;
  const int blocks = (int)min((size_t)2048, ((n0 + n1) / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(adam_dev2_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, o0, n0, o1, n1, h2);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_image_l1(const float* pred, const float* target, const float* mask, const int32_t* fid, int B, int n_per_frame, int C,
                  const float* w, float* loss, float* g_pred, hipStream_t stream) {
  if (!pred || !target || !loss || B <= 0) return HARP_ERR_ARG;
  const float inv = 1.0f / ((float)B * (float)n_per_frame);
  const int gx = min((n_per_frame + 255) / 256, 64);
  hipLaunchKernelGGL(image_l1_kernel, dim3(gx, B), dim3(256), 0, stream, pred, target, mask, fid, n_per_frame, C, inv, w, loss, g_pred);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_kps_loss(const float* gt, const int32_t* fid, const float* pred, int B, int n_joints_pred, const float* w, float* loss,
                  float* g_pred, hipStream_t stream) {
  if (!gt || !pred || !loss || n_joints_pred < 21) return HARP_ERR_ARG;
  hipLaunchKernelGGL(kps_kernel, dim3(B), dim3(64), 0, stream, gt, fid, pred, B, n_joints_pred, w, loss, g_pred);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

static int mesh_terms_launch(const float* verts, const float* ref_verts, const int32_t* nbr_off, const int32_t* nbr_idx,
                             const int32_t* nc_pairs, const int32_t* vp_off, const int32_t* vp_idx, int B, int V, int P, int E,
                             const float* w, float* loss, float* g_verts, const KpsArgs& K, hipStream_t stream) {
  if (!verts || !nbr_off || !nbr_idx || !nc_pairs || !vp_off || !vp_idx || !loss) return HARP_ERR_ARG;
  const int nz = K.gt ? 4 : 3;
  const size_t lds = (size_t)V * 3 * sizeof(float);
  // 512 threads per workgroup: a workgroup stages the whole frame (V * 12 B) whatever its size — 7 stagings per (frame, term) instead of the
  // 13 of 256-thread workgroups (40 -> 30 us, and 9 us off the step: the kernel runs next to the camera-view set-up); 1024 threads: 41 us
  if (lds <= 60 * 1024) {
    // HARP_MESHREG_LDS=<bytes>: ask for more dynamic LDS than the V * 12 B used, i.e. fewer resident workgroups per CU.  80000 (two per CU
    // instead of four) is worth 16 us at 1024^2 on the arm (C5 1.736 -> 1.720: the kernel runs next to the camera view's set-up, which is
    // four times longer there) and costs 4 us at 512^2 (0.670 vs 0.666 in bench.py, three pairs) — off by default.
    size_t ask = lds;
    if (const char* e = getenv("HARP_MESHREG_LDS")) ask = max(lds, (size_t)atoi(e));
    if (ask > 64 * 1024 && hipFuncSetAttribute((const void*)mesh_reg_kernel<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ask) != hipSuccess)
      return HARP_ERR_ARG;
    hipLaunchKernelGGL((mesh_reg_kernel<true, 512>), dim3((V + 511) / 512, B, nz), dim3(512), ask, stream, verts, ref_verts, nbr_off, nbr_idx, nc_pairs,
                       vp_off, vp_idx, B, V, P, E, w, loss, g_verts, K);
  } else
    hipLaunchKernelGGL(mesh_reg_kernel<false>, dim3((V + 255) / 256, B, nz), dim3(256), 0, stream, verts, ref_verts, nbr_off, nbr_idx, nc_pairs,
                       vp_off, vp_idx, B, V, P, E, w, loss, g_verts, K);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_mesh_regularizers(const float* verts, const float* ref_verts, const int32_t* nbr_off, const int32_t* nbr_idx,
                           const int32_t* nc_pairs, const int32_t* vp_off, const int32_t* vp_idx, int B, int V, int P, int E,
                           const float* w, float* loss, float* g_verts, hipStream_t stream) {
  return mesh_terms_launch(verts, ref_verts, nbr_off, nbr_idx, nc_pairs, vp_off, vp_idx, B, V, P, E, w, loss, g_verts, KpsArgs{}, stream);
}

int harp_mesh_kps_terms(const float* verts, const float* ref_verts, const int32_t* nbr_off, const int32_t* nbr_idx,
                        const int32_t* nc_pairs, const int32_t* vp_off, const int32_t* vp_idx, int B, int V, int P, int E,
                        const float* w, float* loss, float* g_verts, const float* kps_gt, const int32_t* fid, const float* kps_pred,
                        int n_joints_pred, const float* w_kps, float* loss_kps, float* g_kps_pred, hipStream_t stream) {
  if (!kps_gt || !kps_pred || !loss_kps || n_joints_pred < 21) return HARP_ERR_ARG;
  const KpsArgs K{kps_gt, fid, kps_pred, n_joints_pred, w_kps, loss_kps, g_kps_pred};
  return mesh_terms_launch(verts, ref_verts, nbr_off, nbr_idx, nc_pairs, vp_off, vp_idx, B, V, P, E, w, loss, g_verts, K, stream);
}

int harp_texture_terms(const float* tex, const float* nmap, const float* mask, const int32_t* dist_albedo, const int32_t* dist_normal,
                       int H, int W, float z_scale, const float* w_albedo, float* loss_albedo, float* g_tex, const float* w_normal,
                       float* loss_normal, float* g_nmap, const float* disp, int n_disp, const float* w_disp, float* loss_disp,
                       float* g_disp, int* draw_counter_bump, hipStream_t stream) {
  if (!tex || !nmap || !dist_albedo || !dist_normal || !loss_albedo || !loss_normal || H <= 0 || W <= 0 || (disp && (!loss_disp || n_disp <= 0)))
    return HARP_ERR_ARG;
  TexTerms A;
  A.tex = tex; A.nmap = nmap; A.mask = mask; A.disp = disp; A.dist_a = dist_albedo; A.dist_n = dist_normal;
  A.H = H; A.W = W; A.n_disp = n_disp; A.z_scale = z_scale;
  A.w_a = w_albedo; A.w_n = w_normal; A.w_d = w_disp; A.l_a = loss_albedo; A.l_n = loss_normal; A.l_d = loss_disp;
  A.g_tex = g_tex; A.g_nmap = g_nmap; A.g_disp = g_disp; A.bump = draw_counter_bump;
  A.nb_smooth = ((H + kST - 1) / kST) * ((W + kST - 1) / kST);          // one workgroup per 32x32-texel tile and map
  A.nb_disp = disp ? min((n_disp + 255) / 256, 64) : 0;
  // ONE workgroup per CU (100 KB of dynamic LDS the kernel does not use; HARP_TEXTERMS_LDS=<bytes> overrides, 0 = none): the kernel is bound
  // by its ~5 M scattered memory-side atomics, which four waves per CU keep as busy as thirty-two do (34 -> 37 us) — but with every CU full
  // of its waves the frames' latency chain that runs next to it (hand_front) took 75 us instead of 57: step -13 us, same-box A/B x4.
  static size_t fence = 0;
  const size_t pad = harp_lds_fence((const void*)texture_terms_kernel, "HARP_TEXTERMS_LDS", 100 * 1024, 0, &fence);
  hipLaunchKernelGGL(texture_terms_kernel, dim3(2 * A.nb_smooth + H + A.nb_disp), dim3(256), pad, stream, A);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_sum_squares(const float* x, int n, const float* w, float* loss, float* g, hipStream_t stream) {
  if (!x || !loss) return HARP_ERR_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(min((n + 255) / 256, 64)), dim3(256), 0, stream, x, n, w, loss, g);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_mse(const float* x, const float* y, int n, float* loss, float* g_x, hipStream_t stream) {
  if (!x || !y || !loss || n <= 0) return HARP_ERR_ARG;
  hipLaunchKernelGGL(mse_kernel, dim3(min((n + 255) / 256, 1024)), dim3(256), 0, stream, x, y, n, loss, g_x);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_texture_smooth_reg(const float* tex, const int32_t* dist, const float* mask, int H, int W, const float* w, float* loss,
                            float* g_tex, hipStream_t stream) {
  if (!tex || !dist || !loss) return HARP_ERR_ARG;
  hipLaunchKernelGGL(tex_smooth_kernel, dim3(min((H * W + 255) / 256, 512)), dim3(256), 0, stream, tex, dist, mask, H, W, w, loss, g_tex);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_close_to_z_reg(const float* nm, int H, int W, float scale, const float* w, float* loss, float* g_nm, hipStream_t stream) {
  if (!nm || !loss) return HARP_ERR_ARG;
  hipLaunchKernelGGL(close_to_z_kernel, dim3(H), dim3(256), 0, stream, nm, H, W, w, scale, loss, g_nm);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_normalize3_fwd(const float* x, int n, float* y, hipStream_t stream) {
  if (!x || !y) return HARP_ERR_ARG;
  hipLaunchKernelGGL(normalize3_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n, y);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_normalize3_bwd(const float* x, const float* gy, int n, float* gx, hipStream_t stream) {
  if (!x || !gy || !gx) return HARP_ERR_ARG;
  hipLaunchKernelGGL(normalize3_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, gy, n, gx);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                   float grad_scale, hipStream_t stream) {
  if (!p || !g || !m || !v || step < 1) return HARP_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int blocks = (int)min((size_t)2048, (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
