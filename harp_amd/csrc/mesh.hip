// Mesh preparation kernels (gfx950): what utils/visualize.py:prepare_mesh does with PyTorch3D objects every
// iteration (Meshes(...), SubdivideMeshes, verts_normals_padded, displacement; utils/visualize.py:45-64) and the
// camera transform of MeshRasterizer.transform — as gather-style, atomics-free kernels
// over static CSR adjacency tables (built once on the host, harp_amd/topology.py).
//
// All kernels: one thread per (frame, vertex), blockIdx.y = frame; (B,V,3) float32 row-major.
#include "harp_common.h"
#include "harp_hip.h"

namespace {

__device__ __forceinline__ float3 ld3(const float* p) { return make_float3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, float3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
  return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// vs = [scale*v0 ; midpoints]  (SubdivideMeshes: verts[:, edges].mean(2) appended; visualize.py:45-52)
__global__ void subdivide_fwd_kernel(const float* __restrict__ v0, const int32_t* __restrict__ edges0, int V0, int E0,
                                     float scale, float* __restrict__ vs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, V = V0 + E0;
  if (i >= V) return;
  const float* src = v0 + (size_t)b * V0 * 3;
  float3 r;
  if (i < V0) {
    r = ld3(src + 3 * i) * scale;
  } else {
    const int a = edges0[2 * (i - V0)], c = edges0[2 * (i - V0) + 1];
    const float3 pa = ld3(src + 3 * a) * scale, pc = ld3(src + 3 * c) * scale;
    r = (pa + pc) * 0.5f;       // mean over 2 (torch: sum then /2)
  }
  st3(vs + ((size_t)b * V + i) * 3, r);
}

// g_v0[i] = scale * (g_vs[i] + 0.5 * sum_{children} g_vs[child])
__global__ void subdivide_bwd_kernel(const float* __restrict__ g_vs, const int32_t* __restrict__ sub_off,
                                     const int32_t* __restrict__ sub_idx, int V0, int V, float scale,
                                     float* __restrict__ g_v0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= V0) return;
  const float* g = g_vs + (size_t)b * V * 3;
  float3 acc = ld3(g + 3 * i);
  float3 ch = make_float3(0.f, 0.f, 0.f);
  for (int k = sub_off[i]; k < sub_off[i + 1]; ++k) ch = ch + ld3(g + 3 * sub_idx[k]);
  st3(g_v0 + ((size_t)b * V0 + i) * 3, (acc + ch * 0.5f) * scale);
}

// Area-weighted vertex normals (Meshes.verts_normals_packed, Appendix A.7): N = sum over incident faces of
// (v2-v1)x(v0-v1); n = N / max(|N|, 1e-6).  Optionally fused displacement: vd = v + n * disp[v]
// (visualize.py:58-64).
__global__ void normals_fwd_kernel(const float* __restrict__ v, const int32_t* __restrict__ faces,
                                   const int32_t* __restrict__ vf_off, const int32_t* __restrict__ vf_idx, int V,
                                   float* __restrict__ n_out, float* __restrict__ inv_len_out,
                                   const float* __restrict__ disp, float* __restrict__ vd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= V) return;
  const float* vb = v + (size_t)b * V * 3;
  float3 N = make_float3(0.f, 0.f, 0.f);
  for (int k = vf_off[i]; k < vf_off[i + 1]; ++k) {
    const int f = vf_idx[k] / 3;
    const float3 p0 = ld3(vb + 3 * faces[3 * f]), p1 = ld3(vb + 3 * faces[3 * f + 1]), p2 = ld3(vb + 3 * faces[3 * f + 2]);
    N = N + cross3(p2 - p1, p0 - p1);
  }
  const float len = sqrtf(dot3(N, N));
  const float inv = 1.0f / fmaxf(len, 1e-6f);
  const float3 n = N * inv;
  const size_t o = ((size_t)b * V + i) * 3;
  st3(n_out + o, n);
  if (inv_len_out) inv_len_out[(size_t)b * V + i] = (len > 1e-6f) ? inv : 0.f;   // 0 => clamped, no grad through |N|
  if (vd) st3(vd + o, ld3(vb + 3 * i) + n * disp[i]);
}

// Stage 1 of normals backward: g_N = (g_n - n (n.g_n)) / |N|   (per vertex)
__global__ void normals_bwd_gN_kernel(const float* __restrict__ n, const float* __restrict__ inv_len,
                                      const float* __restrict__ g_n, int total, float* __restrict__ g_N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float3 nn = ld3(n + 3 * (size_t)i), g = ld3(g_n + 3 * (size_t)i);
  const float il = inv_len[i];
  float3 r;
  if (il == 0.f) {
    r = g * 1e6f;              // clamp_min branch: n = N / 1e-6
  } else {
    r = (g - nn * dot3(nn, g)) * il;
  }
  st3(g_N + 3 * (size_t)i, r);
}

// Stage 2: g_v[i] += sum over incident (face, corner) of d(face normal)/d(corner vertex)^T (g_N[i0]+g_N[i1]+g_N[i2])
__global__ void normals_bwd_gv_kernel(const float* __restrict__ v, const int32_t* __restrict__ faces,
                                      const int32_t* __restrict__ vf_off, const int32_t* __restrict__ vf_idx, int V,
                                      const float* __restrict__ g_N, float* __restrict__ g_v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= V) return;
  const float* vb = v + (size_t)b * V * 3;
  const float* gb = g_N + (size_t)b * V * 3;
  float3 acc = make_float3(0.f, 0.f, 0.f);
  for (int k = vf_off[i]; k < vf_off[i + 1]; ++k) {
    const int fc = vf_idx[k], f = fc / 3, c = fc - 3 * f;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float3 p0 = ld3(vb + 3 * i0), p1 = ld3(vb + 3 * i1), p2 = ld3(vb + 3 * i2);
    const float3 g = ld3(gb + 3 * i0) + ld3(gb + 3 * i1) + ld3(gb + 3 * i2);
    // fn = A x Bv, A = p2-p1, Bv = p0-p1:  g_A = Bv x g,  g_Bv = g x A
    const float3 A = p2 - p1, Bv = p0 - p1;
    const float3 gA = cross3(Bv, g), gB = cross3(g, A);
    if (c == 0) acc = acc + gB;
    else if (c == 2) acc = acc + gA;
    else acc = acc - gA - gB;
  }
  float* o = g_v + ((size_t)b * V + i) * 3;
  o[0] += acc.x; o[1] += acc.y; o[2] += acc.z;
}

// vd = v + n*d backward: g_v += g_vd (done by caller sharing the buffer), g_n = g_vd * d, g_d[i] = sum_b g_vd . n
__global__ void displace_bwd_kernel(const float* __restrict__ g_vd, const float* __restrict__ n, const float* __restrict__ disp,
                                    int B, int V, float* __restrict__ g_n, float* __restrict__ g_disp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const float d = disp[i];
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const size_t o = ((size_t)b * V + i) * 3;
    const float3 g = ld3(g_vd + o);
    acc += dot3(g, ld3(n + o));
    st3(g_n + o, g * d);
  }
  atomicAdd(&g_disp[i], acc);      // several micro-batches may run this concurrently
}

// MeshRasterizer.transform for PerspectiveCameras(in_ndc=False): view = v R + T (row vectors);
// x_ndc = (f X/Z + px - 2 px + S/2)/(S/2), y alike, z = view z.
__global__ void project_fwd_kernel(const float* __restrict__ v, const float* __restrict__ R, const float* __restrict__ T,
                                   int V, float focal, float ppx, float ppy, float half, float* __restrict__ ndc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= V) return;
  const float* r = R + 9 * b;
  const float3 p = ld3(v + ((size_t)b * V + i) * 3);
  const float X = p.x * r[0] + p.y * r[3] + p.z * r[6] + T[3 * b];
  const float Y = p.x * r[1] + p.y * r[4] + p.z * r[7] + T[3 * b + 1];
  const float Z = p.x * r[2] + p.y * r[5] + p.z * r[8] + T[3 * b + 2];
  const float xs = focal * X / Z + ppx, ys = focal * Y / Z + ppy;
  st3(ndc + ((size_t)b * V + i) * 3, make_float3((xs - 2.0f * ppx + half) / half, (ys - 2.0f * ppy + half) / half, Z));
}

// g_ndc -> g_v (+=), g_R (B,9) (+=), g_T (B,3) (+=)
__global__ void __launch_bounds__(256) project_bwd_kernel(const float* __restrict__ v, const float* __restrict__ R,
                                                          const float* __restrict__ T, const float* __restrict__ g_ndc, int V,
                                                          float focal, float half, float* __restrict__ g_v,
                                                          float* __restrict__ g_R, float* __restrict__ g_T) {
  __shared__ float red[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  const float* r = R + 9 * b;
  float gr[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) gr[k] = 0.f;
  if (i < V) {
    const size_t o = ((size_t)b * V + i) * 3;
    const float3 p = ld3(v + o), g = ld3(g_ndc + o);
    const float X = p.x * r[0] + p.y * r[3] + p.z * r[6] + T[3 * b];
    const float Y = p.x * r[1] + p.y * r[4] + p.z * r[7] + T[3 * b + 1];
    const float Z = p.x * r[2] + p.y * r[5] + p.z * r[8] + T[3 * b + 2];
    const float k = focal / (Z * half);
    const float gX = g.x * k, gY = g.y * k;
    const float gZ = g.z - (gX * X + gY * Y) / Z;
    float* go = g_v + o;
    go[0] += r[0] * gX + r[1] * gY + r[2] * gZ;
    go[1] += r[3] * gX + r[4] * gY + r[5] * gZ;
    go[2] += r[6] * gX + r[7] * gY + r[8] * gZ;
    gr[0] = p.x * gX; gr[1] = p.x * gY; gr[2] = p.x * gZ;
    gr[3] = p.y * gX; gr[4] = p.y * gY; gr[5] = p.y * gZ;
    gr[6] = p.z * gX; gr[7] = p.z * gY; gr[8] = p.z * gZ;
    gr[9] = gX; gr[10] = gY; gr[11] = gZ;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float s = block_sum_256(gr[k], red);
    if (threadIdx.x == 0 && s != 0.f) {
      if (k < 9) { if (g_R) atomicAdd(g_R + 9 * b + k, s); }
      else if (g_T) atomicAdd(g_T + 3 * b + (k - 9), s);
    }
  }
}

// per-frame centroid: hand_verts.mean(1) (optimize_sequence.py:476)
__global__ void __launch_bounds__(256) centroid_kernel(const float* __restrict__ v, int V, float* __restrict__ c) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float3 acc = make_float3(0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < V; i += 256) acc = acc + ld3(v + ((size_t)b * V + i) * 3);
  const float sx = block_sum_256(acc.x, red), sy = block_sum_256(acc.y, red), sz = block_sum_256(acc.z, red);
  if (threadIdx.x == 0) { c[3 * b] = sx / V; c[3 * b + 1] = sy / V; c[3 * b + 2] = sz / V; }
}

}  // namespace

extern "C" {

int harp_subdivide_fwd(const float* v0, const int32_t* edges0, int B, int V0, int E0, float scale, float* vs,
                       hipStream_t stream) {
  if (!v0 || !edges0 || !vs) return HARP_ERR_ARG;
  hipLaunchKernelGGL(subdivide_fwd_kernel, dim3((V0 + E0 + 255) / 256, B), dim3(256), 0, stream, v0, edges0, V0, E0, scale, vs);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_subdivide_bwd(const float* g_vs, const int32_t* sub_off, const int32_t* sub_idx, int B, int V0, int V, float scale,
                       float* g_v0, hipStream_t stream) {
  if (!g_vs || !sub_off || !sub_idx || !g_v0) return HARP_ERR_ARG;
  hipLaunchKernelGGL(subdivide_bwd_kernel, dim3((V0 + 255) / 256, B), dim3(256), 0, stream, g_vs, sub_off, sub_idx, V0, V, scale, g_v0);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// n (B,V,3) unit normals, inv_len (B,V) saved for backward; if disp != NULL also vd = v + n*disp (disp (V,))
int harp_vertex_normals_fwd(const float* v, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_idx, int B, int V,
                            float* n, float* inv_len, const float* disp, float* vd, hipStream_t stream) {
  if (!v || !faces || !vf_off || !vf_idx || !n || (disp && !vd)) return HARP_ERR_ARG;
  hipLaunchKernelGGL(normals_fwd_kernel, dim3((V + 255) / 256, B), dim3(256), 0, stream, v, faces, vf_off, vf_idx, V, n, inv_len, disp, vd);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// g_v (B,V,3) += d n / d v ^T g_n ; tmp: (B,V,3) scratch
int harp_vertex_normals_bwd(const float* v, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_idx, int B, int V,
                            const float* n, const float* inv_len, const float* g_n, float* tmp, float* g_v, hipStream_t stream) {
  if (!v || !faces || !n || !inv_len || !g_n || !tmp || !g_v) return HARP_ERR_ARG;
  hipLaunchKernelGGL(normals_bwd_gN_kernel, dim3((B * V + 255) / 256), dim3(256), 0, stream, n, inv_len, g_n, B * V, tmp);
  hipLaunchKernelGGL(normals_bwd_gv_kernel, dim3((V + 255) / 256, B), dim3(256), 0, stream, v, faces, vf_off, vf_idx, V, tmp, g_v);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_displace_bwd(const float* g_vd, const float* n, const float* disp, int B, int V, float* g_n, float* g_disp,
                      hipStream_t stream) {
  if (!g_vd || !n || !disp || !g_n || !g_disp) return HARP_ERR_ARG;
  hipLaunchKernelGGL(displace_bwd_kernel, dim3((V + 255) / 256), dim3(256), 0, stream, g_vd, n, disp, B, V, g_n, g_disp);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_project_fwd(const float* v, const float* R, const float* T, int B, int V, float focal, float ppx, float ppy, int S,
                     float* ndc, hipStream_t stream) {
  if (!v || !R || !T || !ndc) return HARP_ERR_ARG;
  hipLaunchKernelGGL(project_fwd_kernel, dim3((V + 255) / 256, B), dim3(256), 0, stream, v, R, T, V, focal, ppx, ppy, 0.5f * S, ndc);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// accumulating: g_v (B,V,3), g_R (B,9) or NULL, g_T (B,3) or NULL
int harp_project_bwd(const float* v, const float* R, const float* T, const float* g_ndc, int B, int V, float focal, int S,
                     float* g_v, float* g_R, float* g_T, hipStream_t stream) {
  if (!v || !R || !T || !g_ndc || !g_v) return HARP_ERR_ARG;
  hipLaunchKernelGGL(project_bwd_kernel, dim3((V + 255) / 256, B), dim3(256), 0, stream, v, R, T, g_ndc, V, focal, 0.5f * S, g_v, g_R, g_T);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_centroid(const float* v, int B, int V, float* c, hipStream_t stream) {
  if (!v || !c) return HARP_ERR_ARG;
  hipLaunchKernelGGL(centroid_kernel, dim3(B), dim3(256), 0, stream, v, V, c);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
