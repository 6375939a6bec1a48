// Per-frame fused mesh chain (see chain.hip for the design note): the device code, shared by the stand-alone chain kernels
// (chain.hip) and the fused per-frame front kernel (hand_front.hip).
#pragma once
#include "harp_common.h"
#include "harp_hip.h"

namespace cb {

constexpr int kChainThreads = 1024;
constexpr int kMaxPerThread = 4;          // vertices per thread held in registers between stages => V <= 4096
#ifndef HARP_CHAIN_PARTS
#define HARP_CHAIN_PARTS 4
#endif
constexpr int kChainParts = HARP_CHAIN_PARTS;            // workgroups per frame of the wide forms (chain_wide.hip, hand_front.hip)

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ void st(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 normz(V3 a, float eps, float& len) { len = sqrtf(dot(a, a)); return a * (1.0f / fmaxf(len, eps)); }
__device__ __forceinline__ V3 normz_bwd(V3 n, float len, float eps, V3 g) {
  return (len > eps) ? (g - n * dot(n, g)) * (1.0f / len) : g * (1.0f / eps);
}

// N sums over the workgroup with two barriers in total (instead of 2N): red = 16*N floats of LDS, results in out[0..N) (LDS).
template <int N>
__device__ __forceinline__ void block_sum_n(const float* v, float* red, float* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#ifdef CHAIN_SLOW_SUM
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) red[w * N + k] = s;
  }
#else
  // DPP reductions (~8 cycles a step; the shuffle form goes through ds_bpermute, ~100), the N chains interleaved, lane 63 stores the sums
  float t[N];
#pragma unroll
  for (int k = 0; k < N; ++k) t[k] = v[k];
  wave_sum_u_n<N>(t);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < N; ++k) red[w * N + k] = t[k];
  }
#endif
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kChainThreads / 64; ++i) s += red[i * N + threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

// process_info_for_shadow (renderer_helper.py:454-468) + PyTorch3D look_at_rotation, shared by the fused chain
// below and glue.hip:light_setup_kernel.  Includes look_at_rotation's REPLACEMENT branch: when every component of x = normalize(up x z)
// is within 5e-3 of zero (torch.isclose(x, 0, atol=5e-3): the light sits within ~5e-8 rad of the vertical through the centroid, so
// |up x z| < 1e-5 and the eps-clamped normalisation leaves a tiny vector), x is replaced by normalize(y x z), with y = normalize(z x x)
// still formed from the tiny x — the same sequence of operations as the reference, forward and backward.
struct LightCam { V3 c, d, pos, x, y, z, x0; float dl, s, lx, ly, lz, lx2; bool repl; };
__device__ __forceinline__ LightCam light_cam(V3 c, V3 L) {
  LightCam k;
  k.c = c; k.d = L - c;
  k.dl = sqrtf(dot(k.d, k.d));
  k.s = 1.5f / k.dl;
  k.pos = c + k.d * k.s;
  const V3 up = mk(0.f, 1.f, 0.f);
  k.z = normz(c - k.pos, 1e-5f, k.lz);
  k.x0 = normz(cross(up, k.z), 1e-5f, k.lx);
  k.y = normz(cross(k.z, k.x0), 1e-5f, k.ly);
  k.repl = fabsf(k.x0.x) <= 5e-3f && fabsf(k.x0.y) <= 5e-3f && fabsf(k.x0.z) <= 5e-3f;
  k.lx2 = 0.f;
  k.x = k.repl ? normz(cross(k.y, k.z), 1e-5f, k.lx2) : k.x0;
  return k;
}
// rotation (row-major, columns x y z) and translation T = -R^T pos of the light camera
__device__ __forceinline__ void light_cam_RT(const LightCam& k, float* R, float* T) {
  R[0] = k.x.x; R[1] = k.y.x; R[2] = k.z.x; R[3] = k.x.y; R[4] = k.y.y; R[5] = k.z.y; R[6] = k.x.z; R[7] = k.y.z; R[8] = k.z.z;
  T[0] = -dot(k.x, k.pos); T[1] = -dot(k.y, k.pos); T[2] = -dot(k.z, k.pos);
}
// dL/d(light_R) (9, row-major), dL/d(light_T) (3) -> dL/d(light position) `gd`, dL/d(centroid) `gc`
__device__ __forceinline__ void light_cam_bwd(const LightCam& k, const float* gR, const float* gT, V3& gd, V3& gc) {
  const V3 up = mk(0.f, 1.f, 0.f);
  V3 gx = mk(gR[0], gR[3], gR[6]) - k.pos * gT[0];
  V3 gy = mk(gR[1], gR[4], gR[7]) - k.pos * gT[1];
  V3 gz = mk(gR[2], gR[5], gR[8]) - k.pos * gT[2];
  V3 gpos = (k.x * gT[0] + k.y * gT[1] + k.z * gT[2]) * -1.0f;
  V3 gx0 = gx;
  if (k.repl) {                                   // x = normalize(y x z): g_y += z x g, g_z += g x y; the first x only feeds y
    const V3 gq = normz_bwd(k.x, k.lx2, 1e-5f, gx);
    gy = gy + cross(k.z, gq);
    gz = gz + cross(gq, k.y);
    gx0 = mk(0.f, 0.f, 0.f);
  }
  const V3 gyr = normz_bwd(k.y, k.ly, 1e-5f, gy);  // y = normalize(z x x0)
  gz = gz + cross(k.x0, gyr);
  gx0 = gx0 + cross(gyr, k.z);
  const V3 gxr = normz_bwd(k.x0, k.lx, 1e-5f, gx0);   // x0 = normalize(up x z)
  gz = gz + cross(gxr, up);
  const V3 gzr = normz_bwd(k.z, k.lz, 1e-5f, gz);  // z = normalize(c - pos)
  gc = gzr;
  gpos = gpos - gzr;
  gc = gc + gpos;
  gd = gpos * k.s - k.d * (1.5f * dot(k.d, gpos) / (k.dl * k.dl * k.dl));      // pos = c + d * 1.5 / |d|
  gc = gc - gd;
}

// area-weighted vertex normal of vertex i from positions `p` (LDS or global): N, n = N / max(|N|, 1e-6), inv = 1/|N| or 0 if clamped
// The incident-face table is read from L2 (it does not fit next to the positions in LDS): a lane walks ~6 entries and, one load per
// loop trip, each costs a full round trip.  The first kPre entries are therefore fetched up front (independent loads, one latency).
constexpr int kPre = 8;
__device__ __forceinline__ V3 vertex_normal(const float* p, const int4* __restrict__ vf_tri, const int32_t* __restrict__ vf_off, int i,
                                            float& inv_out) {
  V3 N = mk(0.f, 0.f, 0.f);
  const int ks = vf_off[i], n = vf_off[i + 1] - ks;
  int4 t[kPre];
#pragma unroll
  for (int q = 0; q < kPre; ++q) t[q] = vf_tri[ks + min(q, n - 1)];      // (i0, i1, i2, corner); n >= 1 for every mesh vertex
#pragma unroll
  for (int q = 0; q < kPre; ++q) {
    if (q < n) {
      const V3 p0 = ld(p + 3 * t[q].x), p1 = ld(p + 3 * t[q].y), p2 = ld(p + 3 * t[q].z);
      N = N + cross(p2 - p1, p0 - p1);
    }
  }
  for (int k = ks + kPre; k < ks + n; ++k) {
    const int4 u = vf_tri[k];
    const V3 p0 = ld(p + 3 * u.x), p1 = ld(p + 3 * u.y), p2 = ld(p + 3 * u.z);
    N = N + cross(p2 - p1, p0 - p1);
  }
  const float len = sqrtf(dot(N, N));
  const float inv = 1.0f / fmaxf(len, 1e-6f);
  inv_out = (len > 1e-6f) ? inv : 0.f;
  return N * inv;
}

// The same in two halves for kernels that can ask for the (static) incident-face table BEFORE the positions are in LDS (chain_wide.hip):
// TriPre = the vertex's CSR range + its first kPre entries.
struct TriPre { int ks, n; int4 t[kPre]; };
__device__ __forceinline__ TriPre tri_fetch(const int4* __restrict__ vf_tri, const int32_t* __restrict__ vf_off, int i) {
  TriPre r;
  r.ks = vf_off[i]; r.n = vf_off[i + 1] - r.ks;
#pragma unroll
  for (int q = 0; q < kPre; ++q) r.t[q] = vf_tri[r.ks + min(q, r.n - 1)];
  return r;
}
__device__ __forceinline__ V3 vertex_normal_pre(const float* p, const int4* __restrict__ vf_tri, const TriPre& r, float& inv_out) {
  V3 N = mk(0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < kPre; ++q) {
    if (q < r.n) {
      const V3 p0 = ld(p + 3 * r.t[q].x), p1 = ld(p + 3 * r.t[q].y), p2 = ld(p + 3 * r.t[q].z);
      N = N + cross(p2 - p1, p0 - p1);
    }
  }
  for (int k = r.ks + kPre; k < r.ks + r.n; ++k) {
    const int4 u = vf_tri[k];
    const V3 p0 = ld(p + 3 * u.x), p1 = ld(p + 3 * u.y), p2 = ld(p + 3 * u.z);
    N = N + cross(p2 - p1, p0 - p1);
  }
  const float len = sqrtf(dot(N, N));
  const float inv = 1.0f / fmaxf(len, 1e-6f);
  inv_out = (len > 1e-6f) ? inv : 0.f;
  return N * inv;
}

// n floats global -> LDS by the whole workgroup, kU loads per thread in flight (a plain copy loop waits for every load before its LDS store)
template <int kU, int T>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int n, float scale) {
  for (int k0 = threadIdx.x; k0 < n; k0 += kU * T) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) v[u] = src[min(k0 + u * T, n - 1)];
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (k0 + u * T < n) dst[k0 + u * T] = v[u] * scale;
  }
}

__device__ __forceinline__ V3 project(V3 p, const float* r, const float* T, float focal, float pp, float half) {
  const float X = p.x * r[0] + p.y * r[3] + p.z * r[6] + T[0];
  const float Y = p.x * r[1] + p.y * r[4] + p.z * r[7] + T[1];
  const float Z = p.x * r[2] + p.y * r[5] + p.z * r[8] + T[2];
  const float xs = focal * X / Z + pp, ys = focal * Y / Z + pp;
  return mk((xs - 2.0f * pp + half) / half, (ys - 2.0f * pp + half) / half, Z);
}

// Forward chain of frame b by one 1024-thread workgroup.  s_p: V*3 floats of LDS (first the subdivided mesh, then the displaced one).
// verts_ready: s_p[0, V0*3) already holds the hand-layer vertices in metres and joints_m is written (fused front kernel);
// cR / cT / lpos: this frame's camera rotation (9), translation (3) and light position (3) — global or LDS.
__device__ __forceinline__ void mesh_chain_fwd_body(const harp_mesh_chain& A, float* s_p, int b, bool verts_ready, const float* cR,
                                                    const float* cT, const float* lpos) {
  __shared__ float s_red3[16 * 3], s_tot3[3];
  __shared__ float s_cam[12];                  // light R (9) + T (3)
  const int tid = threadIdx.x;
  const int V0 = A.V0, V = A.V0 + A.E0;
  const float half = 0.5f * (float)A.S;
  // ---- metres + SubdivideMeshes (edge midpoints appended after the originals)
  if (!verts_ready) {
    const float* src = A.verts_mm + (size_t)b * V0 * 3;
    for (int i = tid; i < V0 * 3; i += kChainThreads) s_p[i] = src[i] * 1e-3f;
    if (tid < A.NJ * 3) A.joints_m[(size_t)b * A.NJ * 3 + tid] = A.joints_mm[(size_t)b * A.NJ * 3 + tid] * 1e-3f;
  }
  __syncthreads();
  for (int e = tid; e < A.E0; e += kChainThreads) {
    const int a = A.edges0[2 * e], c = A.edges0[2 * e + 1];
    st(s_p + 3 * (V0 + e), (ld(s_p + 3 * a) + ld(s_p + 3 * c)) * 0.5f);
  }
  __syncthreads();
  float* vs = A.vs + (size_t)b * V * 3;
  for (int i = tid; i < V * 3; i += kChainThreads) vs[i] = s_p[i];
  // ---- normals of the subdivided mesh + displacement along them (kept in registers until every thread has read its neighbours)
  V3 vd[kMaxPerThread];
#pragma unroll
  for (int j = 0; j < kMaxPerThread; ++j) {
    const int i = tid + j * kChainThreads;
    if (i < V) {
      float inv;
      const V3 n = vertex_normal(s_p, (const int4*)A.vf_tri, A.vf_off, i, inv);
      const size_t o = (size_t)b * V + i;
      st(A.n1 + o * 3, n);
      A.il1[o] = inv;
      vd[j] = ld(s_p + 3 * i) + n * A.disp[i];
      st(A.vd + o * 3, vd[j]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kMaxPerThread; ++j) {
    const int i = tid + j * kChainThreads;
    if (i < V) st(s_p + 3 * i, vd[j]);
  }
  __syncthreads();
  // ---- normals of the displaced mesh, camera-view projection, centroid
  V3 csum = mk(0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < kMaxPerThread; ++j) {
    const int i = tid + j * kChainThreads;
    if (i < V) {
      float inv;
      const V3 n = vertex_normal(s_p, (const int4*)A.vf_tri, A.vf_off, i, inv);
      const size_t o = (size_t)b * V + i;
      st(A.n2 + o * 3, n);
      A.il2[o] = inv;
      st(A.ndc_c + o * 3, project(vd[j], cR, cT, A.focal, half, half));
      csum = csum + vd[j];
    }
  }
  if (!A.shadow) return;
  const float cs3[3] = {csum.x, csum.y, csum.z};
  block_sum_n<3>(cs3, s_red3, s_tot3);
  if (tid == 0) {
    const V3 c = mk(s_tot3[0] / V, s_tot3[1] / V, s_tot3[2] / V);
    const LightCam k = light_cam(c, ld(lpos));
    st(A.centroid + 3 * b, c);
    float R[9], T[3];
    light_cam_RT(k, R, T);
    for (int q = 0; q < 9; ++q) { s_cam[q] = R[q]; A.light_R[9 * b + q] = R[q]; }
    for (int q = 0; q < 3; ++q) { s_cam[9 + q] = T[q]; A.light_T[3 * b + q] = T[q]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kMaxPerThread; ++j) {
    const int i = tid + j * kChainThreads;
    if (i < V) st(A.ndc_l + ((size_t)b * V + i) * 3, project(vd[j], s_cam, s_cam + 9, A.focal, half, half));
  }
}

// project backward of one vertex: g_ndc -> g_v contribution; accumulates the 12 camera sums (R 9, T 3) into gr
__device__ __forceinline__ V3 project_bwd(V3 p, V3 g, const float* r, const float* T, float focal, float half, float* gr) {
  const float X = p.x * r[0] + p.y * r[3] + p.z * r[6] + T[0];
  const float Y = p.x * r[1] + p.y * r[4] + p.z * r[7] + T[1];
  const float Z = p.x * r[2] + p.y * r[5] + p.z * r[8] + T[2];
  const float k = focal / (Z * half);
  const float gX = g.x * k, gY = g.y * k;
  const float gZ = g.z - (gX * X + gY * Y) / Z;
  gr[0] += p.x * gX; gr[1] += p.x * gY; gr[2] += p.x * gZ;
  gr[3] += p.y * gX; gr[4] += p.y * gY; gr[5] += p.y * gZ;
  gr[6] += p.z * gX; gr[7] += p.z * gY; gr[8] += p.z * gZ;
  gr[9] += gX; gr[10] += gY; gr[11] += gZ;
  return mk(r[0] * gX + r[1] * gY + r[2] * gZ, r[3] * gX + r[4] * gY + r[5] * gZ, r[6] * gX + r[7] * gY + r[8] * gZ);
}

// vertex-normal backward, stage 2 for vertex i: sum over incident (face, corner) of d(face normal)/d(corner)^T (gN[i0]+gN[i1]+gN[i2])
__device__ __forceinline__ V3 nb_term(const float* p, const float* gN, int4 t) {
  const V3 p0 = ld(p + 3 * t.x), p1 = ld(p + 3 * t.y), p2 = ld(p + 3 * t.z);
  const V3 g = ld(gN + 3 * t.x) + ld(gN + 3 * t.y) + ld(gN + 3 * t.z);
  const V3 Av = p2 - p1, Bv = p0 - p1;
  const V3 gA = cross(Bv, g), gB = cross(g, Av);
  return (t.w == 0) ? gB : (t.w == 2) ? gA : (gA + gB) * -1.0f;
}
__device__ __forceinline__ V3 normals_bwd_gather(const float* p, const float* gN, const int4* __restrict__ vf_tri,
                                                 const int32_t* __restrict__ vf_off, int i) {
  V3 acc = mk(0.f, 0.f, 0.f);
  const int ks = vf_off[i], n = vf_off[i + 1] - ks;
  int4 t[kPre];
#pragma unroll
  for (int q = 0; q < kPre; ++q) t[q] = vf_tri[ks + min(q, n - 1)];
#pragma unroll
  for (int q = 0; q < kPre; ++q)
    if (q < n) acc = acc + nb_term(p, gN, t[q]);
  for (int k = ks + kPre; k < ks + n; ++k) acc = acc + nb_term(p, gN, vf_tri[k]);
  return acc;
}
__device__ __forceinline__ V3 normals_bwd_gather_pre(const float* p, const float* gN, const int4* __restrict__ vf_tri, const TriPre& r) {
  V3 acc = mk(0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < kPre; ++q)
    if (q < r.n) acc = acc + nb_term(p, gN, r.t[q]);
  for (int k = r.ks + kPre; k < r.ks + r.n; ++k) acc = acc + nb_term(p, gN, vf_tri[k]);
  return acc;
}
__device__ __forceinline__ V3 normal_len_bwd(V3 n, float il, V3 g) { return (il == 0.f) ? g * 1e6f : (g - n * dot(n, g)) * il; }

// SubdivideMeshes backward + millimetres for base vertex i: 1e-3 (g[i] + 0.5 sum_children g[child]); g: the frame's (V,3) rows (global or LDS)
__device__ __forceinline__ V3 subdivide_bwd_vertex(const float* g, const int32_t* __restrict__ sub_off, const int32_t* __restrict__ sub_idx, int i) {
  V3 ch = mk(0.f, 0.f, 0.f);
  for (int k = sub_off[i]; k < sub_off[i + 1]; ++k) ch = ch + ld(g + 3 * sub_idx[k]);
  return (ld(g + 3 * i) + ch * 0.5f) * 1e-3f;
}

// Backward chain of frame b by one 1024-thread workgroup.  s_mem: 9*V floats of LDS [positions V*3 | gN V*3 | g V*3].
__device__ __forceinline__ void mesh_chain_bwd_body(const harp_mesh_chain& A, float* s_mem, int b) {
  __shared__ float s_red12[16 * 12], s_tot[12];
  __shared__ float s_gc[3];
  const int tid = threadIdx.x;
  const int V0 = A.V0, V = A.V0 + A.E0;
  if (A.light_only) {
    // appearance-only stage: the light position is the only optimised parameter this chain feeds — projection backward of the light view
    // (its 12 camera sums; the vertex gradients it would also produce are not wanted), light camera backward
    if (!A.shadow) return;
    const float half = 0.5f * (float)A.S;
    const size_t fo = (size_t)b * V * 3;
    float gr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gr[k] = 0.f;
    const float* lR = A.light_R + 9 * b;
    const float* lT = A.light_T + 3 * b;
    for (int i = tid; i < V; i += kChainThreads) project_bwd(ld(A.vd + fo + 3 * i), ld(A.g_ndc_l + fo + 3 * i), lR, lT, A.focal, half, gr);
    block_sum_n<12>(gr, s_red12, s_tot);
    if (tid == 0) {
      float gR[9], gT[3];
      for (int k = 0; k < 9; ++k) gR[k] = A.g_light_R[9 * b + k] + s_tot[k];
      for (int k = 0; k < 3; ++k) gT[k] = A.g_light_T[3 * b + k] + s_tot[9 + k];
      const LightCam k = light_cam(ld(A.centroid + 3 * b), ld(A.light_pos + 3 * b));
      V3 gd, gc;
      light_cam_bwd(k, gR, gT, gd, gc);
      float* gl = A.g_light_pos + 3 * b;
      gl[0] += gd.x; gl[1] += gd.y; gl[2] += gd.z;
    }
    return;
  }
  float* s_p = s_mem;
  float* s_gN = s_mem + (size_t)V * 3;
  float* s_g = s_mem + (size_t)V * 6;
  const float half = 0.5f * (float)A.S;
  const size_t fo = (size_t)b * V * 3;
  // ---- load the displaced positions and the gradient gathered so far (mesh regularisers, shader)
  for (int i = tid; i < V * 3; i += kChainThreads) { s_p[i] = A.vd[fo + i]; s_g[i] = A.g_vd[fo + i]; }
  if (tid < A.NJ * 3) A.g_joints_mm[(size_t)b * A.NJ * 3 + tid] = A.g_joints_m[(size_t)b * A.NJ * 3 + tid] * 1e-3f;
  __syncthreads();
  // ---- light view: projection backward, then the light camera backward and the centroid
  if (A.shadow) {
    float gr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gr[k] = 0.f;
    const float* lR = A.light_R + 9 * b;
    const float* lT = A.light_T + 3 * b;
    for (int i = tid; i < V; i += kChainThreads) {
      const V3 gv = project_bwd(ld(s_p + 3 * i), ld(A.g_ndc_l + fo + 3 * i), lR, lT, A.focal, half, gr);
      st(s_g + 3 * i, ld(s_g + 3 * i) + gv);
    }
    block_sum_n<12>(gr, s_red12, s_tot);
    if (tid == 0) {
      const float* gs = s_tot;
      // totals of dL/d(light_R), dL/d(light_T): the shader's share is already in g_light_R / g_light_T
      float gR[9], gT[3];
      for (int k = 0; k < 9; ++k) gR[k] = A.g_light_R[9 * b + k] + gs[k];
      for (int k = 0; k < 3; ++k) gT[k] = A.g_light_T[3 * b + k] + gs[9 + k];
      const LightCam k = light_cam(ld(A.centroid + 3 * b), ld(A.light_pos + 3 * b));
      V3 gd, gc;
      light_cam_bwd(k, gR, gT, gd, gc);
      float* gl = A.g_light_pos + 3 * b;
      gl[0] += gd.x; gl[1] += gd.y; gl[2] += gd.z;
      const float inv = 1.0f / (float)V;
      s_gc[0] = gc.x * inv; s_gc[1] = gc.y * inv; s_gc[2] = gc.z * inv;
    }
    __syncthreads();
    for (int i = tid; i < V * 3; i += kChainThreads) s_g[i] += s_gc[i % 3];
    __syncthreads();
  }
  // ---- camera view: projection backward (only the translation of the camera is a parameter)
  {
    float gr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gr[k] = 0.f;
    const float* cR = A.cam_R + 9 * b;
    const float* cT = A.cam_T + 3 * b;
    for (int i = tid; i < V; i += kChainThreads) {
      const V3 gv = project_bwd(ld(s_p + 3 * i), ld(A.g_ndc_c + fo + 3 * i), cR, cT, A.focal, half, gr);
      st(s_g + 3 * i, ld(s_g + 3 * i) + gv);
    }
    block_sum_n<3>(gr + 9, s_red12, s_tot);
    if (tid < 3 && s_tot[tid] != 0.f) A.g_cam_T[3 * b + tid] += s_tot[tid];
  }
  // ---- normals of the displaced mesh (the shader's g_n2)
  if (A.has_normal_grad) {
    for (int i = tid; i < V; i += kChainThreads)
      st(s_gN + 3 * i, normal_len_bwd(ld(A.n2 + fo + 3 * i), A.il2[(size_t)b * V + i], ld(A.g_n2 + fo + 3 * i)));
    __syncthreads();
    V3 add[kMaxPerThread];
#pragma unroll
    for (int j = 0; j < kMaxPerThread; ++j) {
      const int i = tid + j * kChainThreads;
      if (i < V) add[j] = normals_bwd_gather(s_p, s_gN, (const int4*)A.vf_tri, A.vf_off, i);
    }
#pragma unroll
    for (int j = 0; j < kMaxPerThread; ++j) {
      const int i = tid + j * kChainThreads;
      if (i < V) st(s_g + 3 * i, ld(s_g + 3 * i) + add[j]);      // own element only: no hazard with the gathers (they read gN, p)
    }
  }
  __syncthreads();
  // ---- displacement vd = vs + n1 * d: g_n1 = g_vd * d, g_d += g_vd . n1 (summed over frames by atomics); then the first normals
  for (int i = tid; i < V; i += kChainThreads) {
    const V3 g = ld(s_g + 3 * i), n = ld(A.n1 + fo + 3 * i);
    const float d = A.disp[i];
    atomicAdd(A.g_disp + i, dot(g, n));
    st(s_gN + 3 * i, normal_len_bwd(n, A.il1[(size_t)b * V + i], g * d));
  }
  for (int i = tid; i < V * 3; i += kChainThreads) s_p[i] = A.vs[fo + i];       // positions of the subdivided mesh
  __syncthreads();
  {
    V3 add[kMaxPerThread];
#pragma unroll
    for (int j = 0; j < kMaxPerThread; ++j) {
      const int i = tid + j * kChainThreads;
      if (i < V) add[j] = normals_bwd_gather(s_p, s_gN, (const int4*)A.vf_tri, A.vf_off, i);
    }
#pragma unroll
    for (int j = 0; j < kMaxPerThread; ++j) {
      const int i = tid + j * kChainThreads;
      if (i < V) st(s_g + 3 * i, ld(s_g + 3 * i) + add[j]);
    }
  }
  __syncthreads();
  // ---- SubdivideMeshes backward + millimetres: g_v0[i] = 1e-3 (g[i] + 0.5 sum_children g[child])
  float* gv0 = A.g_v0 + (size_t)b * V0 * 3;
  for (int i = tid; i < V0; i += kChainThreads) st(gv0 + 3 * i, subdivide_bwd_vertex(s_g, A.sub_off, A.sub_idx, i));
}

}  // namespace cb
