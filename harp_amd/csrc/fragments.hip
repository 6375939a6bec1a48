// Fragment-level rasterisation for gfx950: the PyTorch3D op pair `_C.rasterize_meshes` / `_C.rasterize_meshes_backward` itself
// (SURVEY.md §8(b)), K >= 1 fragments per pixel with face id, depth, barycentrics and signed squared edge distance — for callers
// that keep PyTorch3D-style shader classes: the K=10 normal-visualisation renderer of renderer/renderer_helper.py:82-101, the
// silhouette renderer's K=50 fragments (:44-58), or any custom shader.  NOT on the fitting loop's path (that never materialises
// fragments: csrc/raster.hip, csrc/shade*.hip); written for exactness and simplicity, one thread per pixel.
// Semantics (restated in oracle/p3d_like.py:rasterize_meshes): perspective-correct barycentrics, clipped to
// [0,1] and renormalised when blur_radius > 0; a face is a candidate if the pixel centre is inside its blur-dilated bbox and either
// inside the face or closer than sqrt(blur_radius) to an edge; pz = sum bary_i z_i >= 0; the K nearest by pz are kept in ascending
// order, ties keep the lower face index; empty slots hold -1 in every output.
#include "shade_common.h"

int harp_detail_raster_setup(const float* ndc, const int32_t* faces, int B, int V, int F, int S, float r, void* ws, hipStream_t stream);

namespace {

constexpr int kMaxK = 64;

__device__ __forceinline__ float seg_d2(float px, float py, float ax, float ay, float bx, float by, float& tt) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= kEps) { tt = 1.f; return (px - bx) * (px - bx) + (py - by) * (py - by); }
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = fminf(fmaxf(t, 0.f), 1.f);
  tt = t;
  const float qx = ax + t * bax, qy = ay + t * bay;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

struct Pair { float c0, c1, c2, pz, dist; bool inside; Bary br; float s; int amin; float tmin; };

// everything rasterize_meshes computes for one (pixel, face) pair; exact IEEE divisions (this is the reference-shaped op)
__device__ __forceinline__ Pair eval_pair(const Tri& t, float px, float py, bool clip) {
  Pair p;
  Bary& r = p.br;
  r.area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  r.w0 = edge_fn(px, py, t.x1, t.y1, t.x2, t.y2) / r.area;
  r.w1 = edge_fn(px, py, t.x2, t.y2, t.x0, t.y0) / r.area;
  r.w2 = edge_fn(px, py, t.x0, t.y0, t.x1, t.y1) / r.area;
  const float t0 = r.w0 * t.z1 * t.z2, t1 = t.z0 * r.w1 * t.z2, t2 = t.z0 * t.z1 * r.w2;
  const float sum = t0 + t1 + t2;
  r.den_clamped = !(sum > kEps);
  r.den = fmaxf(sum, kEps);
  r.b0 = t0 / r.den; r.b1 = t1 / r.den; r.b2 = t2 / r.den;
  p.inside = r.b0 > 0.f && r.b1 > 0.f && r.b2 > 0.f;
  p.s = 1.f;
  if (clip) {
    const float c0 = fminf(fmaxf(r.b0, 0.f), 1.f), c1 = fminf(fmaxf(r.b1, 0.f), 1.f), c2 = fminf(fmaxf(r.b2, 0.f), 1.f);
    p.s = fmaxf(c0 + c1 + c2, 1e-5f);
    p.c0 = c0 / p.s; p.c1 = c1 / p.s; p.c2 = c2 / p.s;
  } else {
    p.c0 = r.b0; p.c1 = r.b1; p.c2 = r.b2;
  }
  p.pz = p.c0 * t.z0 + p.c1 * t.z1 + p.c2 * t.z2;
  float ta, tb, tc;
  const float d01 = seg_d2(px, py, t.x0, t.y0, t.x1, t.y1, ta);
  const float d02 = seg_d2(px, py, t.x0, t.y0, t.x2, t.y2, tb);
  const float d12 = seg_d2(px, py, t.x1, t.y1, t.x2, t.y2, tc);
  if (d01 <= d02 && d01 <= d12) { p.amin = 0; p.tmin = ta; p.dist = d01; }
  else if (d02 <= d12) { p.amin = 1; p.tmin = tb; p.dist = d02; }
  else { p.amin = 2; p.tmin = tc; p.dist = d12; }
  return p;
}

__global__ void __launch_bounds__(256) fragments_fwd_kernel(const FaceRec* __restrict__ recs, const float4* __restrict__ bbs,
                                                            const int32_t* __restrict__ bins, const int32_t* __restrict__ bin_count,
                                                            int B, int F, int S, int nsx, float blur, int K,
                                                            int32_t* __restrict__ pix_to_face, float* __restrict__ zbuf,
                                                            float* __restrict__ bary, float* __restrict__ dists) {
  const int xi = blockIdx.x * 16 + (threadIdx.x & 15), yi = blockIdx.y * 16 + (threadIdx.x >> 4), b = blockIdx.z;
  if (xi >= S || yi >= S) return;
  const float px = pix_to_ndc(xi, S), py = pix_to_ndc(yi, S);
  const int st = (yi / kSuper) * nsx + (xi / kSuper);
  const int nst = nsx * nsx;
  const int n = bin_count[b * nst + st];
  const int32_t* list = bins + ((size_t)b * nst + st) * F;
  const FaceRec* rb = recs + (size_t)b * F;
  const float4* bbb = bbs + (size_t)b * F;
  const bool clip = blur > 0.f;
  float kz[kMaxK];
  int kf[kMaxK];
  int cnt = 0;
  for (int e = 0; e < n; ++e) {
    const int f = list[e];
    const float4 q = bbb[f];
    if (px > q.y || px < q.x || py > q.w || py < q.z) continue;
    const Tri t = load_tri(rb + f);
    const Pair p = eval_pair(t, px, py, clip);
    if (p.pz < 0.f) continue;
    if (!p.inside && p.dist >= blur) continue;
    // sorted insert; faces arrive in ascending index, so a strict comparison keeps the lower index first among equal depths
    if (cnt == K && !(p.pz < kz[K - 1])) continue;
    int pos = cnt < K ? cnt : K - 1;
    while (pos > 0 && p.pz < kz[pos - 1]) { kz[pos] = kz[pos - 1]; kf[pos] = kf[pos - 1]; --pos; }
    kz[pos] = p.pz; kf[pos] = f;
    if (cnt < K) ++cnt;
  }
  const size_t o = (((size_t)b * S + yi) * S + xi) * K;
  for (int k = 0; k < K; ++k) {
    if (k < cnt) {
      const Pair p = eval_pair(load_tri(rb + kf[k]), px, py, clip);
      pix_to_face[o + k] = kf[k];
      zbuf[o + k] = p.pz;
      bary[(o + k) * 3] = p.c0; bary[(o + k) * 3 + 1] = p.c1; bary[(o + k) * 3 + 2] = p.c2;
      dists[o + k] = p.inside ? -p.dist : p.dist;
    } else {
      pix_to_face[o + k] = -1;
      zbuf[o + k] = -1.f;
      bary[(o + k) * 3] = -1.f; bary[(o + k) * 3 + 1] = -1.f; bary[(o + k) * 3 + 2] = -1.f;
      dists[o + k] = -1.f;
    }
  }
}

// one thread per (pixel, k): recompute the pair, chain grad_zbuf / grad_bary / grad_dists back to the face's NDC vertices
__global__ void __launch_bounds__(256) fragments_bwd_kernel(const float* __restrict__ ndc, const int32_t* __restrict__ faces,
                                                            const int32_t* __restrict__ pix_to_face, const float* __restrict__ g_zbuf,
                                                            const float* __restrict__ g_bary, const float* __restrict__ g_dists,
                                                            int B, int V, int S, float blur, int K, float* __restrict__ g_ndc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)B * S * S * K;
  if (i >= total) return;
  const int f = pix_to_face[i];
  if (f < 0) return;
  const size_t pixel = i / K;
  const int xi = (int)(pixel % S), yi = (int)((pixel / S) % S), b = (int)(pixel / ((size_t)S * S));
  const float px = pix_to_ndc(xi, S), py = pix_to_ndc(yi, S);
  const int vi[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
  const float* vb = ndc + (size_t)b * V * 3;
  Tri t;
  t.x0 = vb[3 * vi[0]]; t.y0 = vb[3 * vi[0] + 1]; t.z0 = vb[3 * vi[0] + 2];
  t.x1 = vb[3 * vi[1]]; t.y1 = vb[3 * vi[1] + 1]; t.z1 = vb[3 * vi[1] + 2];
  t.x2 = vb[3 * vi[2]]; t.y2 = vb[3 * vi[2] + 1]; t.z2 = vb[3 * vi[2] + 2];
  const bool clip = blur > 0.f;
  const Pair p = eval_pair(t, px, py, clip);
  const float gz = g_zbuf ? g_zbuf[i] : 0.f;
  float gc0 = (g_bary ? g_bary[i * 3] : 0.f) + gz * t.z0, gc1 = (g_bary ? g_bary[i * 3 + 1] : 0.f) + gz * t.z1,
        gc2 = (g_bary ? g_bary[i * 3 + 2] : 0.f) + gz * t.z2;
  float out[9] = {0.f, 0.f, gz * p.c0, 0.f, 0.f, gz * p.c1, 0.f, 0.f, gz * p.c2};     // x0 y0 z0 x1 y1 z1 x2 y2 z2
  float gb0 = gc0, gb1 = gc1, gb2 = gc2;
  if (clip) {
    // c_i = clamp(b_i, 0, 1) / max(sum clamp, 1e-5)
    const float cl0 = p.c0 * p.s, cl1 = p.c1 * p.s, cl2 = p.c2 * p.s;
    const bool s_free = (cl0 + cl1 + cl2) > 1e-5f;
    const float dotc = s_free ? (gc0 * cl0 + gc1 * cl1 + gc2 * cl2) / (p.s * p.s) : 0.f;
    const float g0 = gc0 / p.s - dotc, g1 = gc1 / p.s - dotc, g2 = gc2 / p.s - dotc;
    gb0 = (p.br.b0 > 0.f && p.br.b0 < 1.f) ? g0 : 0.f;
    gb1 = (p.br.b1 > 0.f && p.br.b1 < 1.f) ? g1 : 0.f;
    gb2 = (p.br.b2 > 0.f && p.br.b2 < 1.f) ? g2 : 0.f;
  }
  bary_bwd(t, px, py, p.br, gb0, gb1, gb2, out);
  const float gsd = g_dists ? g_dists[i] : 0.f;
  if (gsd != 0.f) {
    const float gd = p.inside ? -gsd : gsd;
    // closest point q = a + t (b - a) on the nearest edge; d = |p - q|^2; dd/dt = 0 at an interior optimum, t constant when clamped
    const int ia = p.amin == 2 ? 1 : 0, ib = p.amin == 0 ? 1 : 2;
    const float ax = ia == 0 ? t.x0 : t.x1, ay = ia == 0 ? t.y0 : t.y1;
    const float bx = ib == 1 ? t.x1 : t.x2, by = ib == 1 ? t.y1 : t.y2;
    const float qx = ax + p.tmin * (bx - ax), qy = ay + p.tmin * (by - ay);
    const float cx = gd * 2.f * (qx - px), cy = gd * 2.f * (qy - py);
    out[3 * ia] += (1.f - p.tmin) * cx; out[3 * ia + 1] += (1.f - p.tmin) * cy;
    out[3 * ib] += p.tmin * cx; out[3 * ib + 1] += p.tmin * cy;
  }
  float* gb = g_ndc + (size_t)b * V * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (out[3 * k + c] != 0.f) atomicAdd(gb + 3 * vi[k] + c, out[3 * k + c]);
}

}  // namespace

extern "C" {

int harp_rasterize_fragments_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, float blur_radius, int K, void* ws,
                                 int32_t* pix_to_face, float* zbuf, float* bary, float* dists, hipStream_t stream) {
  if (!ndc || !faces || !ws || !pix_to_face || !zbuf || !bary || !dists || B <= 0 || F <= 0 || S <= 0 || K < 1 || K > kMaxK || blur_radius < 0.f)
    return HARP_ERR_ARG;
  const int rc = harp_detail_raster_setup(ndc, faces, B, V, F, S, sqrtf(blur_radius), ws, stream);
  if (rc != HARP_OK) return rc;
  const RasterWs W = raster_ws_split(ws, B, F, S);
  hipLaunchKernelGGL(fragments_fwd_kernel, dim3((S + 15) / 16, (S + 15) / 16, B), dim3(256), 0, stream, W.recs, W.bbs, W.bins, W.cnt, B, F, S, W.nsx,
                     blur_radius, K, pix_to_face, zbuf, bary, dists);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_rasterize_fragments_bwd(const float* ndc, const int32_t* faces, const int32_t* pix_to_face, const float* g_zbuf,
                                 const float* g_bary, const float* g_dists, int B, int V, int F, int S, float blur_radius, int K,
                                 float* g_ndc, hipStream_t stream) {
  (void)F;
  if (!ndc || !faces || !pix_to_face || !g_ndc || B <= 0 || S <= 0 || K < 1 || K > kMaxK) return HARP_ERR_ARG;
  const size_t total = (size_t)B * S * S * K;
  hipLaunchKernelGGL(fragments_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ndc, faces, pix_to_face, g_zbuf, g_bary,
                     g_dists, B, V, S, blur_radius, K, g_ndc);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
