// Device helpers shared by the shading kernels (shade.hip: forward pass, general backward pass, light-depth backward;
// shade_bwd.hip: the wave-autonomous backward pass of the fitting loop).  Per-pixel arithmetic of SoftPhongShaderShadow /
// SoftPhongShaderPBR (renderer/renderer_helper.py:106-190, 472-523, 565-592; renderer/pbr_materials.py:58-124) and its derivatives.
#pragma once
#include "harp_common.h"
#include "harp_hip.h"

namespace {

// Reciprocal-based division / square root (v_rcp_f32, v_sqrt_f32: 1 ulp each; the IEEE-exact forms cost ~10 VALU instructions apiece
// and made up a quarter of the forward shader's instruction count).  Image tolerance is 1e-4 absolute, gradients 1e-3 relative.
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// element `i` of a wave-uniform array, addressed as scalar base + 32-bit BYTE offset per lane (global_load ... v_off, s[base:base+1]):
// with the byte offset formed in 32 bits the compiler need not widen it, which it must for `base[i]` (i * sizeof(T) may exceed 32 bits
// as far as it knows) — one VGPR and no 64-bit arithmetic per address instead of a register pair and a v_lshl_add_u64
template <typename T>
__device__ __forceinline__ const T* at32(const T* base, unsigned i) {
  return (const T*)((const char*)base + i * (unsigned)sizeof(T));
}

template <typename T>
__device__ __forceinline__ T* at32m(T* base, unsigned i) {
  return (T*)((char*)base + i * (unsigned)sizeof(T));
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Tri { float x0, y0, z0, x1, y1, z1, x2, y2, z2; };

__device__ __forceinline__ Tri load_tri(const FaceRec* r) {
  const float4 a = r->a, b = r->b;
  Tri t;
  t.x0 = a.x; t.y0 = a.y; t.z0 = a.z; t.x1 = a.w; t.y1 = b.x; t.z1 = b.y; t.x2 = b.z; t.y2 = b.w; t.z2 = r->c.x;
  return t;
}

struct Bary { float b0, b1, b2, w0, w1, w2, area, den; bool den_clamped; };

__device__ __forceinline__ Bary bary_fwd(const Tri& t, float px, float py) {
  Bary r;
  r.area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  const float ra = rcp(r.area);
  r.w0 = edge_fn(px, py, t.x1, t.y1, t.x2, t.y2) * ra;
  r.w1 = edge_fn(px, py, t.x2, t.y2, t.x0, t.y0) * ra;
  r.w2 = edge_fn(px, py, t.x0, t.y0, t.x1, t.y1) * ra;
  const float t0 = r.w0 * t.z1 * t.z2, t1 = t.z0 * r.w1 * t.z2, t2 = t.z0 * t.z1 * r.w2;
  const float s = t0 + t1 + t2;
  r.den_clamped = !(s > kEps);
  r.den = fmaxf(s, kEps);
  const float rd = rcp(r.den);
  r.b0 = t0 * rd; r.b1 = t1 * rd; r.b2 = t2 * rd;
  return r;
}

// d(edge(p,a,b)) for a, b
__device__ __forceinline__ void edge_bwd(float g, float px, float py, float ax, float ay, float bx, float by, float& gax,
                                         float& gay, float& gbx, float& gby) {
  gax += g * (py - by); gay += g * (bx - px); gbx += g * -(py - ay); gby += g * (px - ax);
}

// BarycentricPerspectiveCorrectionBackward + BarycentricCoordsBackward: g_b -> g on the 9 face-vertex NDC comps.
// out[9] = g x0 y0 z0 x1 y1 z1 x2 y2 z2 (accumulated into).
__device__ __forceinline__ void bary_bwd(const Tri& t, float px, float py, const Bary& r, float gb0, float gb1, float gb2,
                                         float* out) {
  const float t0 = r.w0 * t.z1 * t.z2, t1 = t.z0 * r.w1 * t.z2, t2 = t.z0 * t.z1 * r.w2;
  const float rd = rcp(r.den);
  float gt0 = gb0 * rd, gt1 = gb1 * rd, gt2 = gb2 * rd;
  if (!r.den_clamped) {
    const float c = (gb0 * t0 + gb1 * t1 + gb2 * t2) * (rd * rd);
    gt0 -= c; gt1 -= c; gt2 -= c;
  }
  const float gw0 = gt0 * t.z1 * t.z2, gw1 = gt1 * t.z0 * t.z2, gw2 = gt2 * t.z0 * t.z1;
  out[2] += gt1 * r.w1 * t.z2 + gt2 * r.w2 * t.z1;                // z0
  out[5] += gt0 * r.w0 * t.z2 + gt2 * r.w2 * t.z0;                // z1
  out[8] += gt0 * r.w0 * t.z1 + gt1 * r.w1 * t.z0;                // z2
  const float ra = rcp(r.area);
  const float ge0 = gw0 * ra, ge1 = gw1 * ra, ge2 = gw2 * ra;
  const float garea = -(gw0 * r.w0 + gw1 * r.w1 + gw2 * r.w2) * ra;
  // e0 = edge(p, v1, v2); e1 = edge(p, v2, v0); e2 = edge(p, v0, v1); area = edge(v2, v0, v1)
  edge_bwd(ge0, px, py, t.x1, t.y1, t.x2, t.y2, out[3], out[4], out[6], out[7]);
  edge_bwd(ge1, px, py, t.x2, t.y2, t.x0, t.y0, out[6], out[7], out[0], out[1]);
  edge_bwd(ge2, px, py, t.x0, t.y0, t.x1, t.y1, out[0], out[1], out[3], out[4]);
  edge_bwd(garea, t.x2, t.y2, t.x0, t.y0, t.x1, t.y1, out[0], out[1], out[3], out[4]);
  out[6] += garea * (t.y1 - t.y0);
  out[7] += garea * -(t.x1 - t.x0);
}

// bilinear, align_corners=True, border padding, v flipped
struct Bil { int x0, y0; float wx, wy; float gxm, gym; };
__device__ __forceinline__ Bil bil_setup(float u, float v, int W, int H) {
  Bil s;
  float x = u * (float)(W - 1), y = (1.0f - v) * (float)(H - 1);   // grid = (2u-1, 1-2v); ix = (g+1)/2*(W-1)
  s.gxm = (x > 0.f && x < (float)(W - 1)) ? 1.f : 0.f;
  s.gym = (y > 0.f && y < (float)(H - 1)) ? 1.f : 0.f;
  x = fminf(fmaxf(x, 0.f), (float)(W - 1));
  y = fminf(fmaxf(y, 0.f), (float)(H - 1));
  const float fx = floorf(x), fy = floorf(y);
  s.x0 = (int)fx; s.y0 = (int)fy; s.wx = x - fx; s.wy = y - fy;
  return s;
}
__device__ __forceinline__ V3 texel(const float* m, int x, int y, int W, int H) {
  if (x >= W || y >= H) return mk(0.f, 0.f, 0.f);
  return ld(m + ((size_t)y * W + x) * 3);
}
__device__ __forceinline__ V3 bil_sample(const float* m, const Bil& s, int W, int H, V3* ddx, V3* ddy) {
  const V3 t00 = texel(m, s.x0, s.y0, W, H), t10 = texel(m, s.x0 + 1, s.y0, W, H);
  const V3 t01 = texel(m, s.x0, s.y0 + 1, W, H), t11 = texel(m, s.x0 + 1, s.y0 + 1, W, H);
  const float ax = 1.f - s.wx, ay = 1.f - s.wy;
  if (ddx) {
    *ddx = ((t10 - t00) * ay + (t11 - t01) * s.wy) * s.gxm;
    *ddy = ((t01 - t00) * ax + (t11 - t10) * s.wx) * s.gym;
  }
  return t00 * (ax * ay) + t10 * (s.wx * ay) + t01 * (ax * s.wy) + t11 * (s.wx * s.wy);
}
// albedo + (normalised) normal map from the interleaved texel array of harp_pack_texels: texel i = [r g b nx | ny nz 0 0]; the 2x2
// footprint is 4 x 32 B (2 cache lines) instead of 8 x 12 B in two arrays (4 lines)
__device__ __forceinline__ void bil_sample2(const float4* tn, const Bil& s, int W, int H, V3& alb, V3& nm, V3* adx, V3* ady, V3* mdx, V3* mdy) {
  const int x1 = min(s.x0 + 1, W - 1), y1 = min(s.y0 + 1, H - 1);
  const float k10 = (s.x0 + 1 < W) ? 1.f : 0.f, k01 = (s.y0 + 1 < H) ? 1.f : 0.f;
  // unsigned 32-bit element offsets from the (wave-uniform) base: the loads take the scalar-base + 32-bit-offset form instead of a
  // 64-bit address per lane and corner
  const unsigned o00 = 2u * (unsigned)(s.y0 * W + s.x0), o10 = 2u * (unsigned)(s.y0 * W + x1);
  const unsigned o01 = 2u * (unsigned)(y1 * W + s.x0), o11 = 2u * (unsigned)(y1 * W + x1);
  const float4 a00 = *at32(tn, o00), b00 = *at32(tn, o00 + 1u), a10 = *at32(tn, o10), b10 = *at32(tn, o10 + 1u);
  const float4 a01 = *at32(tn, o01), b01 = *at32(tn, o01 + 1u), a11 = *at32(tn, o11), b11 = *at32(tn, o11 + 1u);
  const V3 t00 = mk(a00.x, a00.y, a00.z), t10 = mk(a10.x, a10.y, a10.z) * k10, t01 = mk(a01.x, a01.y, a01.z) * k01,
           t11 = mk(a11.x, a11.y, a11.z) * (k10 * k01);
  const V3 m00 = mk(a00.w, b00.x, b00.y), m10 = mk(a10.w, b10.x, b10.y) * k10, m01 = mk(a01.w, b01.x, b01.y) * k01,
           m11 = mk(a11.w, b11.x, b11.y) * (k10 * k01);
  const float ax = 1.f - s.wx, ay = 1.f - s.wy;
  if (adx) {
    *adx = ((t10 - t00) * ay + (t11 - t01) * s.wy) * s.gxm;
    *mdx = ((m10 - m00) * ay + (m11 - m01) * s.wy) * s.gxm;
  }
  *ady = ((t01 - t00) * ax + (t11 - t10) * s.wx) * s.gym;
  *mdy = ((m01 - m00) * ax + (m11 - m10) * s.wx) * s.gym;
  alb = t00 * (ax * ay) + t10 * (s.wx * ay) + t01 * (ax * s.wy) + t11 * (s.wx * s.wy);
  nm = m00 * (ax * ay) + m10 * (s.wx * ay) + m01 * (ax * s.wy) + m11 * (s.wx * s.wy);
}
__device__ __forceinline__ void bil_scatter(float* g, const Bil& s, int W, int H, V3 v) {
  const float ax = 1.f - s.wx, ay = 1.f - s.wy;
  const float w[4] = {ax * ay, s.wx * ay, ax * s.wy, s.wx * s.wy};
  const int xs[4] = {s.x0, s.x0 + 1, s.x0, s.x0 + 1}, ys[4] = {s.y0, s.y0, s.y0 + 1, s.y0 + 1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (xs[k] < W && ys[k] < H && w[k] != 0.f) {
      float* p = g + ((size_t)ys[k] * W + xs[k]) * 3;
      atomicAdd(p, v.x * w[k]); atomicAdd(p + 1, v.y * w[k]); atomicAdd(p + 2, v.z * w[k]);
    }
  }
}

// the same into a double map (the accumulators of harp_texel_reduce)
__device__ __forceinline__ void bil_scatter_d(double* g, const Bil& s, int W, int H, V3 v) {
  const float ax = 1.f - s.wx, ay = 1.f - s.wy;
  const float w[4] = {ax * ay, s.wx * ay, ax * s.wy, s.wx * s.wy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = s.x0 + (k & 1), y = s.y0 + (k >> 1);
    if (x < W && y < H && w[k] != 0.f) {
      double* p = g + ((size_t)y * W + x) * 3;
      atomicAdd(p, (double)(v.x * w[k])); atomicAdd(p + 1, (double)(v.y * w[k])); atomicAdd(p + 2, (double)(v.z * w[k]));
    }
  }
}

// shadow test sigmoid: fast exp + reciprocal (rel. error < 1e-6 where it is not saturated; image tolerance 1e-4)
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Everything the forward computes for one covered pixel (recomputed by the backward).
struct Frag {
  Tri t; Bary br; int i0, i1, i2, u0, u1, u2;
  V3 p, n, texel, m, nprime, nhat, nn, lhat, ldir; float lnp, lnh, llen;
  float u, v, cosr, vis, zq; int ix, iy; V3 q; Bil bs;
  V3 tu, tv;   // tangent frame
  float s, a;
};

}  // namespace
