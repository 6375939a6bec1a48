// Tile rasteriser for gfx950: replaces PyTorch3D's rasterize_meshes (coarse+fine) and, fused into the same
// per-pixel face walk, SoftSilhouetteShader/sigmoid_alpha_blend.
//   reference call sites: renderer/renderer_helper.py:52-58 (K=50 soft silhouette), :76-79 / :444-447 (K=1),
//   :344, :353 (light view, camera view).  Semantics: SURVEY.md Appendix A.2/A.3.
//
// Design (not PyTorch3D's): nothing of shape (B,S,S,K) is ever materialised.
//   1. face_setup   one thread per (frame, face): gather 3 NDC vertices, write a 64-B FaceRec with the
//                   blur-dilated bbox; culled faces get an empty box.
//   2. bin_faces    one workgroup per (frame, 64x64 super-tile): scan the frame's bboxes (coalesced float4),
//                   wave-ballot compaction -> ascending face-id list in HBM/L2 (deterministic, no atomics).
//   3. raster       one workgroup (4 waves) per 16x16 tile, one pixel per lane (wave = 16x4 strip).
//                   Stages the tile's faces into LDS (SoA float4, broadcast reads), each wave ballots the
//                   staged faces against its own strip and walks only the hits.  Per pixel, in registers:
//                   nearest-z face (K=1 semantics, ties -> lower face id like PyTorch3D) and the running
//                   silhouette product prod_f (1 - sigmoid(-d_f/sigma)).
//   4. sil_bwd      same walk, rim pixels only: dL/dalpha -> dL/d(ndc xy) of the face vertices (atomics).
#include "harp_common.h"
#include "harp_hip.h"

#ifndef RASTER_EXP
#define RASTER_EXP 0
#endif
namespace {

constexpr int kStage = 256;   // faces staged in LDS per round (17 KB)

struct Tri {
  float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

__device__ __forceinline__ Tri tri_from(const float4 a, const float4 b, const float4 c) {
  Tri t;
  t.x0 = a.x; t.y0 = a.y; t.z0 = a.z; t.x1 = a.w;
  t.y1 = b.x; t.z1 = b.y; t.x2 = b.z; t.y2 = b.w;
  t.z2 = c.x;
  return t;
}

// BarycentricCoordsForward + BarycentricPerspectiveCorrectionForward; returns "inside" (all bary > 0).
__device__ __forceinline__ bool tri_bary(const Tri& t, float px, float py, float& b0, float& b1, float& b2) {
  const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  const float w0 = edge_fn(px, py, t.x1, t.y1, t.x2, t.y2) / area;
  const float w1 = edge_fn(px, py, t.x2, t.y2, t.x0, t.y0) / area;
  const float w2 = edge_fn(px, py, t.x0, t.y0, t.x1, t.y1) / area;
  const float t0 = w0 * t.z1 * t.z2, t1 = t.z0 * w1 * t.z2, t2 = t.z0 * t.z1 * w2;
  const float den = fmaxf(t0 + t1 + t2, kEps);
  b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
  return b0 > 0.f && b1 > 0.f && b2 > 0.f;
}

// PointLineDistanceForward: squared distance to segment (a,b); also returns the clamped parameter t.
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by, float& tt) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= kEps) { tt = 1.f; return (px - bx) * (px - bx) + (py - by) * (py - by); }
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = fminf(fmaxf(t, 0.f), 1.f);
  tt = t;
  const float qx = ax + t * bax, qy = ay + t * bay;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

__global__ void __launch_bounds__(256) face_setup_kernel(const float* __restrict__ ndc, const int32_t* __restrict__ faces,
                                                         int V, int F, float r, FaceRec* __restrict__ recs,
                                                         float4* __restrict__ bbs) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (f >= F) return;
  const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
  const float* vb = ndc + (size_t)b * V * 3;
  Tri t;
  t.x0 = vb[3 * i0]; t.y0 = vb[3 * i0 + 1]; t.z0 = vb[3 * i0 + 2];
  t.x1 = vb[3 * i1]; t.y1 = vb[3 * i1 + 1]; t.z1 = vb[3 * i1 + 2];
  t.x2 = vb[3 * i2]; t.y2 = vb[3 * i2 + 1]; t.z2 = vb[3 * i2 + 2];
  const float area = edge_fn(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2);
  const float zmax = fmaxf(t.z0, fmaxf(t.z1, t.z2)), zmin = fminf(t.z0, fminf(t.z1, t.z2));
  // skipped for every pixel: behind camera, |area| <= eps, any vertex with z < eps (z_invalid), non-finite
  const bool cull = (zmax < 0.f) || (area <= kEps && area >= -kEps) || (zmin < kEps) || !(area == area);
  FaceRec rec;
  rec.a = make_float4(t.x0, t.y0, t.z0, t.x1);
  rec.b = make_float4(t.y1, t.z1, t.x2, t.y2);
  rec.c = make_float4(t.z2, area, 0.f, 0.f);
  if (cull) {
    rec.bb = make_float4(3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f);
  } else {
    rec.bb = make_float4(fminf(t.x0, fminf(t.x1, t.x2)) - r, fmaxf(t.x0, fmaxf(t.x1, t.x2)) + r,
                         fminf(t.y0, fminf(t.y1, t.y2)) - r, fmaxf(t.y0, fmaxf(t.y1, t.y2)) + r);
  }
  recs[(size_t)b * F + f] = rec;
  bbs[(size_t)b * F + f] = rec.bb;      // contiguous copy: the binning pass streams it with fully coalesced 16-B loads
}

// Ascending-order compaction of `pred` across a 256-thread block. Returns this thread's slot (or -1) and
// advances *running (uniform). lds_cnt: 4 ints.
__device__ __forceinline__ int block_compact(bool pred, int running, int* lds_cnt, int& total) {
  const unsigned long long m = __ballot(pred);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds_cnt[w] = __popcll(m);
  __syncthreads();
  int base = running;
  for (int i = 0; i < w; ++i) base += lds_cnt[i];
  total = lds_cnt[0] + lds_cnt[1] + lds_cnt[2] + lds_cnt[3];
  const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  return pred ? pos : -1;
}

// One WAVE per (frame, 64x64 super-tile): streams the frame's bboxes 64 at a time, ballot + popcount compaction,
// no LDS, no barriers; the list comes out in ascending face order (== PyTorch3D's tie-break order).
__global__ void __launch_bounds__(256) bin_faces_kernel(const float4* __restrict__ bbs, int F, int S, int nsx,
                                                        int32_t* __restrict__ bins, int32_t* __restrict__ bin_count) {
  const int lane = threadIdx.x & 63;
  const int nst = nsx * nsx;
  const int st = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (st >= nst) return;
  const int sx = st % nsx, sy = st / nsx;
  const int x_lo = sx * kSuper, x_hi = min(x_lo + kSuper, S) - 1;
  const int y_lo = sy * kSuper, y_hi = min(y_lo + kSuper, S) - 1;
  // NDC decreases with pixel index
  const float nx_hi = pix_to_ndc(x_lo, S), nx_lo = pix_to_ndc(x_hi, S);
  const float ny_hi = pix_to_ndc(y_lo, S), ny_lo = pix_to_ndc(y_hi, S);
  const float4* bb = bbs + (size_t)b * F;
  int32_t* out = bins + ((size_t)b * nst + st) * F;
  int running = 0;
  for (int base = 0; base < F; base += 64) {
    const int f = base + lane;
    bool hit = false;
    if (f < F) {
      const float4 q = bb[f];
      hit = !(nx_lo > q.y || nx_hi < q.x || ny_lo > q.w || ny_hi < q.z);
    }
    const unsigned long long m = __ballot(hit);
    if (hit) out[running + __popcll(m & ((1ull << lane) - 1ull))] = f;
    running += __popcll(m);
  }
  if (lane == 0) bin_count[b * nst + st] = running;
}

// Exact evaluation of one (pixel, face) pair on the soft-silhouette rim (PointTriangleDistanceForward + sigmoid):
// MODE 1 accumulates log2(1-p) in Q20 fixed point (or sets the saturation flag), MODE 2 the gradient of the signed distance.
template <int MODE>
__device__ __forceinline__ void raster_exact(const Tri& t, float px, float py, int pl, int slot, float blur, float sigma, int* s_logp,
                                             unsigned char* s_deep, const float* s_coef, float (*s_g)[6]) {
  const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  const float sg = (area > 0.f) ? 1.f : ((area < 0.f) ? -1.f : 0.f);
  const bool inside = (edge_fn(px, py, t.x1, t.y1, t.x2, t.y2) * sg > 0.f) && (edge_fn(px, py, t.x2, t.y2, t.x0, t.y0) * sg > 0.f) &&
                      (edge_fn(px, py, t.x0, t.y0, t.x1, t.y1) * sg > 0.f);
  float ta, tb, tc;
  const float d01 = seg_dist2(px, py, t.x0, t.y0, t.x1, t.y1, ta);
  const float d02 = seg_dist2(px, py, t.x0, t.y0, t.x2, t.y2, tb);
  const float d12 = seg_dist2(px, py, t.x1, t.y1, t.x2, t.y2, tc);
  const float dist = fminf(d01, fminf(d02, d12));
  if (!(inside || dist < blur)) return;
  const float sd = inside ? -dist : dist;
  const float p = 1.0f / (1.0f + expf(sd / sigma));   // sigmoid(-sd/sigma)
  if (MODE == 1) {
    const float q = 1.0f - p;
    if (q == 0.f) s_deep[pl] = 1;
    else atomicAdd(&s_logp[pl], (int)rintf(log2f(q) * 1048576.0f));
  } else {
    // d alpha / d sd = -P p / sigma; s_coef = ga * (-P / sigma)
    const float g_sd = s_coef[pl] * p;
    const float gd = inside ? -g_sd : g_sd;     // d/d(dist^2)
    // PointLineDistanceBackward on the argmin edge (t treated as constant)
    int ia, ib; float ax, ay, bx, by, tt;
    if (d01 <= d02 && d01 <= d12) { ia = 0; ib = 1; ax = t.x0; ay = t.y0; bx = t.x1; by = t.y1; tt = ta; }
    else if (d02 <= d12)          { ia = 0; ib = 2; ax = t.x0; ay = t.y0; bx = t.x2; by = t.y2; tt = tb; }
    else                          { ia = 1; ib = 2; ax = t.x1; ay = t.y1; bx = t.x2; by = t.y2; tt = tc; }
    const float qx = ax + tt * (bx - ax), qy = ay + tt * (by - ay);
    const float cx = gd * 2.f * (qx - px), cy = gd * 2.f * (qy - py);
    atomicAdd(&s_g[slot][2 * ia], (1.f - tt) * cx);
    atomicAdd(&s_g[slot][2 * ia + 1], (1.f - tt) * cy);
    atomicAdd(&s_g[slot][2 * ib], tt * cx);
    atomicAdd(&s_g[slot][2 * ib + 1], tt * cy);
  }
}

constexpr int kQCap = 4096;          // exact-path candidate queue (entries) per round
constexpr int kPix = kSuper * kSuper;  // 4096 pixels of the LDS-resident screen tile
constexpr int kBigArea = 128;          // faces whose clipped bbox exceeds this many pixels are rasterised by a whole wave
constexpr int kBigCap = 64;            // per round; further large faces fall back to their owner lane

__device__ __forceinline__ Tri tri_of(const FaceRec& r) {
  Tri t;
  t.x0 = r.a.x; t.y0 = r.a.y; t.z0 = r.a.z; t.x1 = r.a.w;
  t.y1 = r.b.x; t.z1 = r.b.y; t.x2 = r.b.z; t.y2 = r.b.w;
  t.z2 = r.c.x;
  return t;
}

// per-face constants of the division-free classification
struct FaceCtx {
  Tri t;
  float4 bb;
  float sg, rcp_a, K12, K20, K01, B12, B20, B01;
  int id, slot;
};

__device__ __forceinline__ FaceCtx face_ctx(const FaceRec& r, int id, int slot, float blur, float sigma) {
  FaceCtx c;
  c.t = tri_of(r);
  c.bb = r.bb;
  c.id = id; c.slot = slot;
  const Tri& t = c.t;
  const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  c.sg = (area > 0.f) ? 1.f : ((area < 0.f) ? -1.f : 0.f);
  c.rcp_a = __builtin_amdgcn_rcpf(area);
  // squared edge lengths: e0 <-> (v1,v2), e1 <-> (v2,v0), e2 <-> (v0,v1)
  const float l12 = (t.x2 - t.x1) * (t.x2 - t.x1) + (t.y2 - t.y1) * (t.y2 - t.y1);
  const float l20 = (t.x0 - t.x2) * (t.x0 - t.x2) + (t.y0 - t.y2) * (t.y0 - t.y2);
  const float l01 = (t.x1 - t.x0) * (t.x1 - t.x0) + (t.y1 - t.y0) * (t.y1 - t.y0);
  c.K12 = 18.0f * sigma * l12; c.K20 = 18.0f * sigma * l20; c.K01 = 18.0f * sigma * l01;
  const float Bf = blur * 1.00001f;
  c.B12 = Bf * l12; c.B20 = Bf * l20; c.B01 = Bf * l01;
  return c;
}

// One (pixel, face) visit of the cheap pass. px,py = pixel centre (NDC), pl = pixel index inside the LDS tile.
template <int MODE>
__device__ __forceinline__ void visit_pixel(const FaceCtx& c, float px, float py, int pl, float blur, float sigma,
                                            unsigned long long* s_zb, int* s_logp, unsigned char* s_deep, const float* s_coef,
                                            float (*s_g)[6], uint32_t* s_q, int* s_qn) {
  const Tri& t = c.t;
  // inside  <=>  all perspective-corrected barycentrics > 0  <=>  e_i * sign(area') > 0 for all i   (z > 0 by culling)
  const float e0 = edge_fn(px, py, t.x1, t.y1, t.x2, t.y2);
  const float e1 = edge_fn(px, py, t.x2, t.y2, t.x0, t.y0);
  const float e2 = edge_fn(px, py, t.x0, t.y0, t.x1, t.y1);
  const float e0s = e0 * c.sg, e1s = e1 * c.sg, e2s = e2 * c.sg;
  const bool inside = (e0s > 0.f) && (e1s > 0.f) && (e2s > 0.f);
  if (MODE != 2 && inside) {
    // BarycentricCoordsForward + PerspectiveCorrection with reciprocals (<= 2 ulp from the IEEE-division form)
    const float t0 = (e0 * c.rcp_a) * t.z1 * t.z2, t1 = t.z0 * (e1 * c.rcp_a) * t.z2, t2 = t.z0 * t.z1 * (e2 * c.rcp_a);
    const float rd = __builtin_amdgcn_rcpf(fmaxf(t0 + t1 + t2, kEps));
    const float pz = (t0 * rd) * t.z0 + (t1 * rd) * t.z1 + (t2 * rd) * t.z2;
#if RASTER_EXP == 1
    if (pz >= 0.f && pz < 0.0001f) atomicMin(&s_zb[pl], ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)c.id);
#else
    if (pz >= 0.f) atomicMin(&s_zb[pl], ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)c.id);
#endif
  }
  if (MODE >= 1) {
    bool soft = (MODE == 1) ? (s_deep[pl] == 0) : (s_coef[pl] != 0.f);
    if (soft) {
      if (inside) {
        // deeper than sqrt(18 sigma) inside every edge LINE (<= segment distance): sigmoid saturates to exactly 1 in fp32
        // (exp(-18) < 2^-24), the factor (1-p) is exactly 0 -> alpha = 1, no gradient.
        if (e0 * e0 > c.K12 && e1 * e1 > c.K20 && e2 * e2 > c.K01) {
          if (MODE == 1) s_deep[pl] = 1;
          soft = false;
        }
      } else {
        // outside: the triangle lies beyond the line of any violated edge, so dist >= that line distance >= blur
        if ((e0s < 0.f && e0 * e0 >= c.B12) || (e1s < 0.f && e1 * e1 >= c.B20) || (e2s < 0.f && e2 * e2 >= c.B01)) soft = false;
      }
      if (soft) {
        const int pos = atomicAdd(s_qn, 1);
        const uint32_t ent = ((uint32_t)pl << 20) | (uint32_t)((MODE == 1) ? c.id : c.slot);
        if (pos < kQCap) s_q[pos] = ent;
        else raster_exact<MODE>(t, px, py, pl, c.slot, blur, sigma, s_logp, s_deep, s_coef, s_g);   // queue full: in place
      }
    }
  }
}

// MODE 0: depth only (light view).  MODE 1: nearest face + soft-silhouette alpha (camera view).
// MODE 2: silhouette backward: dL/dalpha -> dL/d(ndc xy) of the face vertices.
//
// One workgroup owns one 64x64 SCREEN TILE of one frame and keeps it in LDS for the whole kernel:
//   zb[4096]   u64  (z bits << 32 | face id): K=1 z-buffer resolved with ds_min_u64 — ties go to the lower face id, exactly
//                   PyTorch3D's visiting order, and the result does not depend on scheduling;
//   logp[4096] i32  sum of round(2^20 log2(1-p_f)): the silhouette product in a fixed-point log domain — integer adds
//                   commute, so alpha is bit-reproducible run to run although faces are processed concurrently;
//   deep[4096] u8   "some face saturates this pixel" (1-p == 0 exactly in fp32)  =>  alpha = 1 exactly.
// Each round takes 256 faces of the tile's list:
//   1a  FACE-PARALLEL: one lane = one small face, walking the pixels of its own clipped bbox (small triangles — 8x8-px boxes
//       here — keep ~40 % of the lanes useful instead of ~10 % for a pixel-parallel walk); faces with a large bbox are deferred;
//   1b  the deferred large faces are rasterised by the whole workgroup, one pixel per lane (no lane waits for a sliver);
//   2   the (pixel, face) pairs on the 0.3-px rim that need segment distances + exp were appended to an LDS queue by 1a/1b and
//       are evaluated here fully lane-packed.
// 1a/1b only do the division-free classification (edge-function signs, exact-saturation and far-outside bounds).
template <int MODE>
__global__ void __launch_bounds__(256) raster_kernel(const FaceRec* __restrict__ recs, const int32_t* __restrict__ bins,
                                                     const int32_t* __restrict__ bin_count, int F, int S, int nsx, float blur,
                                                     float sigma, int32_t* __restrict__ face_id, float* __restrict__ zbuf,
                                                     float* __restrict__ alpha, const float* __restrict__ g_alpha,
                                                     const int32_t* __restrict__ faces, int V, float* __restrict__ g_ndc) {
  __shared__ unsigned long long s_zb[MODE == 2 ? 1 : kPix];
  __shared__ int s_logp[MODE == 1 ? kPix : 1];
  __shared__ unsigned char s_deep[MODE == 1 ? kPix : 1];
  __shared__ float s_coef[MODE == 2 ? kPix : 1];        // MODE 2: ga * (-P / sigma) per pixel (0 = pixel needs nothing)
  __shared__ float s_g[MODE == 2 ? 256 : 1][6];         // MODE 2: per-face gradient accumulators of the current round
  __shared__ uint32_t s_q[MODE == 0 ? 1 : kQCap];
  __shared__ int s_qn, s_nbig;
  __shared__ FaceRec s_bigrec[kBigCap];                  // records of this round's deferred large faces
  __shared__ int s_bigid[kBigCap], s_bigslot[kBigCap];
  __shared__ float s_px[kSuper], s_py[kSuper];          // exact pixel-centre NDC of the tile's columns / rows (one division each)

  const int st = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int nst = nsx * nsx;
  const int n = bin_count[b * nst + st];
  const int sx0 = (st % nsx) * kSuper, sy0 = (st / nsx) * kSuper;
  const int sx1 = min(sx0 + kSuper, S) - 1, sy1 = min(sy0 + kSuper, S) - 1;
  const int32_t* list = bins + ((size_t)b * nst + st) * F;
  const FaceRec* rb = recs + (size_t)b * F;
  const float fS = (float)S;
  if (MODE != 2 && n == 0) {
    // empty tile: nothing to resolve
    for (int k = tid; k < kPix; k += 256) {
      const int xi = sx0 + (k & 63), yi = sy0 + (k >> 6);
      if (xi >= S || yi >= S) continue;
      const size_t o = ((size_t)b * S + yi) * S + xi;
      face_id[o] = -1;
      if (zbuf) zbuf[o] = -1.0f;
      if (MODE == 1) alpha[o] = 0.0f;
    }
    return;
  }

  // ---- init the LDS tile
  bool any_need = false;
  for (int k = tid; k < kPix; k += 256) {
    if (MODE != 2) s_zb[k] = ~0ull;
    if (MODE == 1) { s_logp[k] = 0; s_deep[k] = 0; }
    if (MODE == 2) {
      const int xi = sx0 + (k & 63), yi = sy0 + (k >> 6);
      float c = 0.f;
      if (xi < S && yi < S) {
        const size_t o = ((size_t)b * S + yi) * S + xi;
        const float P = 1.0f - alpha[o], ga = g_alpha[o];
        if (P != 0.f && ga != 0.f) c = ga * (-P / sigma);
      }
      s_coef[k] = c;
      any_need |= (c != 0.f);
    }
  }
  if (tid == 0) { s_qn = 0; s_nbig = 0; }
  if (tid < kSuper) { s_px[tid] = pix_to_ndc(sx0 + tid, S); s_py[tid] = pix_to_ndc(sy0 + tid, S); }
  if (MODE == 2) {
    if (__syncthreads_or(any_need ? 1 : 0) == 0) return;
  } else {
    __syncthreads();
  }

  for (int base = 0; base < n; base += 256) {
    // ================= phase 1a: one lane = one (small) face
    const int e = base + tid;
    if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) s_g[tid][c] = 0.f;
    }
    if (e < n) {
      const int id = list[e];
      const FaceRec r = rb[id];
      // pixel range of the dilated bbox (conservative by one pixel; the exact NDC test follows per pixel)
      const int x_lo = max(sx0, (int)floorf((fS * (1.0f - r.bb.y) - 1.0f) * 0.5f));
      const int x_hi = min(sx1, (int)ceilf((fS * (1.0f - r.bb.x) - 1.0f) * 0.5f));
      const int y_lo = max(sy0, (int)floorf((fS * (1.0f - r.bb.w) - 1.0f) * 0.5f));
      const int y_hi = min(sy1, (int)ceilf((fS * (1.0f - r.bb.z) - 1.0f) * 0.5f));
      const int area_px = max(0, x_hi - x_lo + 1) * max(0, y_hi - y_lo + 1);
      bool deferred = false;
      if (area_px > kBigArea) {
        const int pos = atomicAdd(&s_nbig, 1);
        if (pos < kBigCap) { s_bigrec[pos] = r; s_bigid[pos] = id; s_bigslot[pos] = tid; deferred = true; }
      }
      if (!deferred && area_px > 0) {
        const FaceCtx c = face_ctx(r, id, tid, blur, sigma);
        // flattened walk over the clipped bbox: the wave iterates max(area) times, not max(width) * max(height)
        int xi = x_lo, yi = y_lo;
        for (int k = 0; k < area_px; ++k) {
          const int lx = xi - sx0, ly = yi - sy0;
          const float px = s_px[lx], py = s_py[ly];
#if RASTER_EXP == 2
          if (px > 5.f)
#else
          if (!(px > r.bb.y || px < r.bb.x || py > r.bb.w || py < r.bb.z))
#endif
            visit_pixel<MODE>(c, px, py, (ly << 6) | lx, blur, sigma, s_zb, s_logp, s_deep, s_coef, s_g, s_q, &s_qn);
          if (++xi > x_hi) { xi = x_lo; ++yi; }
        }
      }
    }
    __syncthreads();
    // ================= phase 1b: large faces, one WAVE per face, lanes stride over the pixels of its clipped bbox
    const int nbig = min(s_nbig, kBigCap);
    for (int i = (tid >> 6); i < nbig; i += 4) {
      const FaceRec r = s_bigrec[i];
      const FaceCtx c = face_ctx(r, s_bigid[i], s_bigslot[i], blur, sigma);
      const int x_lo = max(sx0, (int)floorf((fS * (1.0f - r.bb.y) - 1.0f) * 0.5f));
      const int x_hi = min(sx1, (int)ceilf((fS * (1.0f - r.bb.x) - 1.0f) * 0.5f));
      const int y_lo = max(sy0, (int)floorf((fS * (1.0f - r.bb.w) - 1.0f) * 0.5f));
      const int y_hi = min(sy1, (int)ceilf((fS * (1.0f - r.bb.z) - 1.0f) * 0.5f));
      const int wpx = x_hi - x_lo + 1, area_px = wpx * (y_hi - y_lo + 1);
      const float inv_w = 1.0f / (float)wpx;
      for (int k = (tid & 63); k < area_px; k += 64) {
        const int row = (int)(((float)k + 0.5f) * inv_w), col = k - row * wpx;      // exact for k < 4096
        const int lx = x_lo + col - sx0, ly = y_lo + row - sy0;
        const float px = s_px[lx], py = s_py[ly];
        if (px > r.bb.y || px < r.bb.x || py > r.bb.w || py < r.bb.z) continue;
        visit_pixel<MODE>(c, px, py, (ly << 6) | lx, blur, sigma, s_zb, s_logp, s_deep, s_coef, s_g, s_q, &s_qn);
      }
    }
    __syncthreads();
    if (tid == 0) s_nbig = 0;
    if (MODE >= 1) {
      // ================= phase 2: drain the queue, one lane = one (pixel, face) candidate
      const int qn = min(s_qn, kQCap);
      for (int i = tid; i < qn; i += 256) {
        const uint32_t ent = s_q[i];
        const int pl = (int)(ent >> 20), key = (int)(ent & 0xFFFFFu);
        const int id = (MODE == 1) ? key : list[base + key];
        const Tri t = tri_of(rb[id]);
        const float px = s_px[pl & 63], py = s_py[pl >> 6];
        raster_exact<MODE>(t, px, py, pl, key, blur, sigma, s_logp, s_deep, s_coef, s_g);
      }
      __syncthreads();
      if (tid == 0) s_qn = 0;
      if (MODE == 2 && e < n) {
        // flush this round's per-face accumulators: <= 6 global atomics per (face, tile)
        const int id = list[e];
        float* gb = g_ndc + (size_t)b * V * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float gx = s_g[tid][2 * k], gy = s_g[tid][2 * k + 1];
          if (gx != 0.f || gy != 0.f) {
            const int v = faces[3 * id + k];
            atomicAdd(gb + 3 * v, gx);
            atomicAdd(gb + 3 * v + 1, gy);
          }
        }
      }
    }
    __syncthreads();
  }

  if (MODE != 2) {
    for (int k = tid; k < kPix; k += 256) {
      const int xi = sx0 + (k & 63), yi = sy0 + (k >> 6);
      if (xi >= S || yi >= S) continue;
      const size_t o = ((size_t)b * S + yi) * S + xi;
      const unsigned long long zk = s_zb[k];
      const bool hit = zk != ~0ull;
      face_id[o] = hit ? (int)(zk & 0xFFFFFFFFu) : -1;
      if (zbuf) zbuf[o] = hit ? __uint_as_float((unsigned)(zk >> 32)) : -1.0f;
      if (MODE == 1) alpha[o] = s_deep[k] ? 1.0f : (1.0f - exp2f((float)s_logp[k] * (1.0f / 1048576.0f)));
    }
  }
}

}  // namespace

extern "C" {

size_t harp_rasterize_ws_bytes(int B, int F, int S) {
  const int nsx = (S + kSuper - 1) / kSuper;
  size_t recs = (size_t)B * F * sizeof(FaceRec);
  size_t bbs = (size_t)B * F * sizeof(float4);
  size_t bins = (size_t)B * nsx * nsx * F * sizeof(int32_t);
  size_t cnt = (((size_t)B * nsx * nsx * sizeof(int32_t)) + 255) / 256 * 256;
  return recs + bbs + bins + cnt;
}

static void ws_split(void* ws, int B, int F, int S, FaceRec** recs, int32_t** bins, int32_t** cnt, float4** bbs = nullptr) {
  const int nsx = (S + kSuper - 1) / kSuper;
  char* p = (char*)ws;
  *recs = (FaceRec*)p;
  p += (size_t)B * F * sizeof(FaceRec);
  if (bbs) *bbs = (float4*)p;
  p += (size_t)B * F * sizeof(float4);
  *bins = (int32_t*)p;
  p += (size_t)B * nsx * nsx * F * sizeof(int32_t);
  *cnt = (int32_t*)p;
}

// Forward rasterisation of B frames sharing one face table.
//   ndc (B,V,3) f32 [x_ndc, y_ndc, z_view]; faces (F,3) i32.
//   soft != 0: also accumulate the soft-silhouette alpha (blur_radius, sigma as in renderer_helper.py:44-58).
//   Outputs (B,S,S): face_id i32 (frame-local, -1 empty), zbuf f32 or NULL (-1 empty), alpha f32 (soft only).
//   ws: harp_rasterize_ws_bytes() bytes, 256-B aligned; must stay untouched until the matching backward ran.
int harp_rasterize_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                       float sigma, void* ws, int32_t* face_id, float* zbuf, float* alpha, hipStream_t stream) {
  if (!ndc || !faces || !ws || !face_id || B <= 0 || F <= 0 || S <= 0 || (soft && !alpha)) return HARP_ERR_ARG;
  FaceRec* recs; int32_t *bins, *cnt; float4* bbs;
  ws_split(ws, B, F, S, &recs, &bins, &cnt, &bbs);
  const int nsx = (S + kSuper - 1) / kSuper;
  const float r = soft ? sqrtf(blur_radius) : 0.f;
  hipLaunchKernelGGL(face_setup_kernel, dim3((F + 255) / 256, B), dim3(256), 0, stream, ndc, faces, V, F, r, recs, bbs);
  hipLaunchKernelGGL(bin_faces_kernel, dim3((nsx * nsx + 3) / 4, B), dim3(256), 0, stream, bbs, F, S, nsx, bins, cnt);
  const dim3 grid(nsx * nsx, B);
  if (soft)
    hipLaunchKernelGGL(raster_kernel<1>, grid, dim3(256), 0, stream, recs, bins, cnt, F, S, nsx, blur_radius, sigma,
                       face_id, zbuf, alpha, nullptr, nullptr, 0, nullptr);
  else
    hipLaunchKernelGGL(raster_kernel<0>, grid, dim3(256), 0, stream, recs, bins, cnt, F, S, nsx, 0.f, 1.f, face_id, zbuf,
                       nullptr, nullptr, nullptr, 0, nullptr);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// Soft-silhouette backward: g_alpha (B,S,S) -> accumulates (atomicAdd) into g_ndc (B,V,3) (x,y components).
// ws must be the workspace of the matching harp_rasterize_fwd(soft=1); alpha its output.
int harp_silhouette_bwd(const int32_t* faces, int B, int V, int F, int S, float blur_radius, float sigma, const void* ws,
                        const float* alpha, const float* g_alpha, float* g_ndc, hipStream_t stream) {
  if (!faces || !ws || !alpha || !g_alpha || !g_ndc) return HARP_ERR_ARG;
  FaceRec* recs; int32_t *bins, *cnt; float4* bbs;
  ws_split((void*)ws, B, F, S, &recs, &bins, &cnt, &bbs);
  const int nsx = (S + kSuper - 1) / kSuper;
  const dim3 grid(nsx * nsx, B);
  hipLaunchKernelGGL(raster_kernel<2>, grid, dim3(256), 0, stream, recs, bins, cnt, F, S, nsx, blur_radius, sigma, nullptr,
                     nullptr, (float*)alpha, g_alpha, faces, V, g_ndc);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
