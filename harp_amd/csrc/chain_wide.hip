// The per-frame mesh chain on MORE THAN ONE CU per frame (gfx950).  chain.hip runs a frame's chain as one 1024-thread workgroup: its
// stages are separated by workgroup barriers, but a 3093-vertex (hand) / 4083-vertex (arm) mesh is FOUR serial passes of that workgroup per
// stage, and a step's 32 frames use 32 of the chip's 256 CUs — the head and the tail of a step were 31 % of it on 12 % of the chip
// (profiles/r05_e_timeline_one_step.txt).  Here a frame is kParts workgroups, each owning a contiguous quarter of the VERTICES (one per
// thread: a stage is one pass); every workgroup stages the whole frame's positions in its LDS (37 - 49 KB, one coalesced read), so
// neighbour gathers stay LDS reads, and the stages that need another part's results are kernel boundaries (1.5 - 2 us inside a replayed
// hipGraph) instead of grid barriers:
//   forward  A: metres + SubdivideMeshes (whole mesh, in LDS) | normals + displacement of the own vertices | partial centroid sums
//            B: displaced mesh -> LDS | centroid, light camera (every workgroup, same arithmetic) | normals, both projections (own vertices)
//   backward A: both projections backward (own vertices) -> g, partial camera sums | length backward of the displaced mesh's normals -> gN
//            B: displaced mesh + gN -> LDS | light camera backward (every workgroup), centroid share | normal gather | displacement
//            C: subdivided mesh + gN -> LDS | normal gather -> g
//            D: SubdivideMeshes backward + millimetres -> g_v0     (the fused tails of hand_back.hip / arm_front.hip do this themselves)
// Same arithmetic per vertex as chain_body.h (the centroid / the camera sums are sums of kParts partial sums instead of one block sum).
//   reference: utils/visualize.py:45-64 (prepare_mesh), renderer_helper.py:454-468 (process_info_for_shadow), MeshRasterizer.transform
//   for both views (renderer_helper.py:344, 353), hand_verts.mean(1) (optimize_sequence.py:476).
#include "chain_body.h"

namespace {

using namespace cb;

constexpr int kParts = kChainParts;

__device__ __forceinline__ int part_size(int V) { return (V + kParts - 1) / kParts; }

__global__ void __launch_bounds__(kChainThreads) chain_wide_a_kernel(const harp_mesh_chain A, float* __restrict__ cpart, int clear_grads) {
  extern __shared__ float s_p[];               // V*3 positions of the subdivided mesh
  __shared__ float s_red3[16 * 3], s_tot3[3];
  const int b = blockIdx.x / kParts, part = blockIdx.x % kParts, tid = threadIdx.x;
  const int V0 = A.V0, V = A.V0 + A.E0, per = part_size(V);
  const int i = part * per + tid;
  const bool own = tid < per && i < V;
  // the static tables this thread will need (its edges, its vertex's incident faces) are requested first: they arrive while the positions are staged
  const int iv = own ? i : 0;
  const TriPre tp = tri_fetch((const int4*)A.vf_tri, A.vf_off, iv);
  constexpr int kEdgeIter = 3;                 // edges per thread held ahead (E0 <= 3 * 1024: 2315 hand, 3057 arm; the rest in the loop below)
  int ea[kEdgeIter], ec[kEdgeIter];
#pragma unroll
  for (int u = 0; u < kEdgeIter; ++u) {
    const int e = min(tid + u * kChainThreads, max(A.E0 - 1, 0));
    ea[u] = A.E0 > 0 ? A.edges0[2 * e] : 0; ec[u] = A.E0 > 0 ? A.edges0[2 * e + 1] : 0;
  }
  const float dsp = A.disp[iv];
  stage_rows<3, kChainThreads>(s_p, A.verts_mm + (size_t)b * V0 * 3, V0 * 3, 1e-3f);
  if (part == 0 && tid < A.NJ * 3) A.joints_m[(size_t)b * A.NJ * 3 + tid] = A.joints_mm[(size_t)b * A.NJ * 3 + tid] * 1e-3f;
  if (clear_grads) {                           // the two gradient segments the key-point / mesh terms accumulate into (they start after B)
    float* gv = const_cast<float*>(A.g_vd) + (size_t)b * V * 3;
    const int lo = part * per * 3, hi = min((part + 1) * per, V) * 3;
    for (int k = lo + tid; k < hi; k += kChainThreads) gv[k] = 0.f;
    if (part == 0 && tid < A.NJ * 3) const_cast<float*>(A.g_joints_m)[(size_t)b * A.NJ * 3 + tid] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kEdgeIter; ++u) {
    const int e = tid + u * kChainThreads;
    if (e < A.E0) st(s_p + 3 * (V0 + e), (ld(s_p + 3 * ea[u]) + ld(s_p + 3 * ec[u])) * 0.5f);
  }
  for (int e = tid + kEdgeIter * kChainThreads; e < A.E0; e += kChainThreads) {
    const int a = A.edges0[2 * e], c = A.edges0[2 * e + 1];
    st(s_p + 3 * (V0 + e), (ld(s_p + 3 * a) + ld(s_p + 3 * c)) * 0.5f);
  }
  __syncthreads();
  V3 csum = mk(0.f, 0.f, 0.f);
  if (own) {
    const size_t o = (size_t)b * V + i;
    const V3 p = ld(s_p + 3 * i);
    st(A.vs + o * 3, p);
    float inv;
    const V3 n = vertex_normal_pre(s_p, (const int4*)A.vf_tri, tp, inv);
    st(A.n1 + o * 3, n);
    A.il1[o] = inv;
    const V3 vd = p + n * dsp;
    st(A.vd + o * 3, vd);
    csum = vd;
  }
  if (!A.shadow) return;
  const float cs3[3] = {csum.x, csum.y, csum.z};
  block_sum_n<3>(cs3, s_red3, s_tot3);
  if (tid < 3) cpart[((size_t)b * kParts + part) * 3 + tid] = s_tot3[tid];
}

__global__ void __launch_bounds__(kChainThreads) chain_wide_b_kernel(const harp_mesh_chain A, const float* __restrict__ cpart) {
  extern __shared__ float s_p[];               // V*3 positions of the displaced mesh
  __shared__ float s_cam[12];                  // light R (9) + T (3)
  const int b = blockIdx.x / kParts, part = blockIdx.x % kParts, tid = threadIdx.x;
  const int V = A.V0 + A.E0, per = part_size(V);
  const int i = part * per + tid;
  const bool own = tid < per && i < V;
  const float half = 0.5f * (float)A.S;
  const TriPre tp = tri_fetch((const int4*)A.vf_tri, A.vf_off, own ? i : 0);
  stage_rows<6, kChainThreads>(s_p, A.vd + (size_t)b * V * 3, V * 3, 1.0f);
  if (A.shadow && tid == 0) {
    float cs[3] = {0.f, 0.f, 0.f};
    for (int q = 0; q < kParts; ++q)
      for (int c = 0; c < 3; ++c) cs[c] += cpart[((size_t)b * kParts + q) * 3 + c];
    const V3 c = mk(cs[0] / V, cs[1] / V, cs[2] / V);
    const LightCam k = light_cam(c, ld(A.light_pos + 3 * b));
    float R[9], T[3];
    light_cam_RT(k, R, T);
    for (int q = 0; q < 9; ++q) s_cam[q] = R[q];
    for (int q = 0; q < 3; ++q) s_cam[9 + q] = T[q];
    if (part == 0) {
      st(A.centroid + 3 * b, c);
      for (int q = 0; q < 9; ++q) A.light_R[9 * b + q] = R[q];
      for (int q = 0; q < 3; ++q) A.light_T[3 * b + q] = T[q];
    }
  }
  __syncthreads();
  if (own) {
    const size_t o = (size_t)b * V + i;
    float inv;
    const V3 n = vertex_normal_pre(s_p, (const int4*)A.vf_tri, tp, inv);
    st(A.n2 + o * 3, n);
    A.il2[o] = inv;
    const V3 vd = ld(s_p + 3 * i);
    st(A.ndc_c + o * 3, project(vd, A.cam_R + 9 * b, A.cam_T + 3 * b, A.focal, half, half));
    if (A.shadow) st(A.ndc_l + o * 3, project(vd, s_cam, s_cam + 9, A.focal, half, half));
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------------------
// scratch (harp_mesh_chain_wide_ws_floats): centroid parts B*kParts*3 | camera-sum parts B*kParts*16 | G, gNa, gNb: B*V*3 each
struct WideWs { float *cpart, *psum, *G, *gNa, *gNb; };
__host__ __device__ inline WideWs wide_ws(float* ws, int B, int V) {
  WideWs w; float* p = ws;
  w.cpart = p; p += (size_t)B * kParts * 3; w.psum = p; p += (size_t)B * kParts * 16;
  w.G = p; p += (size_t)B * V * 3; w.gNa = p; p += (size_t)B * V * 3; w.gNb = p;
  return w;
}

__global__ void __launch_bounds__(kChainThreads) chain_wide_bwd_a_kernel(const harp_mesh_chain A, const WideWs W) {
  __shared__ float s_red[16 * 15], s_tot[15];
  const int b = blockIdx.x / kParts, part = blockIdx.x % kParts, tid = threadIdx.x;
  const int V = A.V0 + A.E0, per = part_size(V);
  const int i = part * per + tid;
  const bool own = tid < per && i < V;
  const float half = 0.5f * (float)A.S;
  const size_t fo = (size_t)b * V * 3;
  if (part == 0 && tid < A.NJ * 3) A.g_joints_mm[(size_t)b * A.NJ * 3 + tid] = A.g_joints_m[(size_t)b * A.NJ * 3 + tid] * 1e-3f;
  float gr[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) gr[k] = 0.f;
  if (own) {
    const V3 p = ld(A.vd + fo + 3 * i);
    V3 g = ld(A.g_vd + fo + 3 * i);
    if (A.shadow) g = g + project_bwd(p, ld(A.g_ndc_l + fo + 3 * i), A.light_R + 9 * b, A.light_T + 3 * b, A.focal, half, gr);
    {
      float gc[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) gc[k] = 0.f;
      g = g + project_bwd(p, ld(A.g_ndc_c + fo + 3 * i), A.cam_R + 9 * b, A.cam_T + 3 * b, A.focal, half, gc);
      gr[12] = gc[9]; gr[13] = gc[10]; gr[14] = gc[11];
    }
    st(W.G + fo + 3 * i, g);
    if (A.has_normal_grad) st(W.gNa + fo + 3 * i, normal_len_bwd(ld(A.n2 + fo + 3 * i), A.il2[(size_t)b * V + i], ld(A.g_n2 + fo + 3 * i)));
  }
  block_sum_n<15>(gr, s_red, s_tot);
  if (tid < 15) W.psum[((size_t)b * kParts + part) * 16 + tid] = s_tot[tid];
}

// normals of the displaced mesh backward + displacement: G (own) <- G + centroid share + gather(vd, gNa); gNb (own) = length backward of n1
__global__ void __launch_bounds__(kChainThreads) chain_wide_bwd_b_kernel(const harp_mesh_chain A, const WideWs W) {
  extern __shared__ float s_mem[];             // [positions V*3 | gN V*3]
  __shared__ float s_gc[3];
  const int b = blockIdx.x / kParts, part = blockIdx.x % kParts, tid = threadIdx.x;
  const int V = A.V0 + A.E0, per = part_size(V);
  const int i = part * per + tid;
  const bool own = tid < per && i < V;
  const size_t fo = (size_t)b * V * 3;
  float* s_p = s_mem;
  float* s_gN = s_mem + (size_t)V * 3;
  const TriPre tp = tri_fetch((const int4*)A.vf_tri, A.vf_off, (own && A.has_normal_grad) ? i : 0);
  // (this vertex's own rows too: in flight while the frame is staged)
  const int iv = own ? i : 0;
  const V3 g_in = ld(W.G + fo + 3 * iv), n1v = ld(A.n1 + fo + 3 * iv);
  const float il1v = A.il1[(size_t)b * V + iv], dv = A.disp[iv];
  if (A.has_normal_grad) {
    stage_rows<5, kChainThreads>(s_p, A.vd + fo, V * 3, 1.0f);
    stage_rows<5, kChainThreads>(s_gN, W.gNa + fo, V * 3, 1.0f);
  }
  if (tid == 0) {
    float tot[15];
    for (int k = 0; k < 15; ++k) {
      float a = 0.f;
      for (int q = 0; q < kParts; ++q) a += W.psum[((size_t)b * kParts + q) * 16 + k];
      tot[k] = a;
    }
    s_gc[0] = s_gc[1] = s_gc[2] = 0.f;
    if (A.shadow) {
      // totals of dL/d(light_R), dL/d(light_T): the shader's share is already in g_light_R / g_light_T
      float gR[9], gT[3];
      for (int k = 0; k < 9; ++k) gR[k] = A.g_light_R[9 * b + k] + tot[k];
      for (int k = 0; k < 3; ++k) gT[k] = A.g_light_T[3 * b + k] + tot[9 + k];
      const LightCam k = light_cam(ld(A.centroid + 3 * b), ld(A.light_pos + 3 * b));
      V3 gd, gc;
      light_cam_bwd(k, gR, gT, gd, gc);
      const float inv = 1.0f / (float)V;
      s_gc[0] = gc.x * inv; s_gc[1] = gc.y * inv; s_gc[2] = gc.z * inv;
      if (part == 0) {
        float* gl = A.g_light_pos + 3 * b;
        gl[0] += gd.x; gl[1] += gd.y; gl[2] += gd.z;
      }
    }
    if (part == 0)
      for (int c = 0; c < 3; ++c)
        if (tot[12 + c] != 0.f) A.g_cam_T[3 * b + c] += tot[12 + c];
  }
  __syncthreads();
  if (own) {
    V3 g = g_in + mk(s_gc[0], s_gc[1], s_gc[2]);
    if (A.has_normal_grad) g = g + normals_bwd_gather_pre(s_p, s_gN, (const int4*)A.vf_tri, tp);
    // displacement vd = vs + n1 * d: g_n1 = g_vd * d, g_d += g_vd . n1 (summed over frames by atomics)
    const V3 n = n1v;
    const float d = dv;
    atomicAdd(A.g_disp + i, dot(g, n));
    st(W.gNb + fo + 3 * i, normal_len_bwd(n, il1v, g * d));
    st(W.G + fo + 3 * i, g);
  }
}

// normals of the subdivided mesh backward: G (own) <- G + gather(vs, gNb)
__global__ void __launch_bounds__(kChainThreads) chain_wide_bwd_c_kernel(const harp_mesh_chain A, const WideWs W) {
  extern __shared__ float s_mem[];
  const int b = blockIdx.x / kParts, part = blockIdx.x % kParts, tid = threadIdx.x;
  const int V = A.V0 + A.E0, per = part_size(V);
  const int i = part * per + tid;
  const size_t fo = (size_t)b * V * 3;
  float* s_p = s_mem;
  float* s_gN = s_mem + (size_t)V * 3;
  const bool own = tid < per && i < V;
  const TriPre tp = tri_fetch((const int4*)A.vf_tri, A.vf_off, own ? i : 0);
  V3 g = mk(0.f, 0.f, 0.f);
  if (own) g = ld(W.G + fo + 3 * i);
  stage_rows<5, kChainThreads>(s_p, A.vs + fo, V * 3, 1.0f);
  stage_rows<5, kChainThreads>(s_gN, W.gNb + fo, V * 3, 1.0f);
  __syncthreads();
  if (own) st(W.G + fo + 3 * i, g + normals_bwd_gather_pre(s_p, s_gN, (const int4*)A.vf_tri, tp));
}

// SubdivideMeshes backward + millimetres: g_v0[i] = 1e-3 (g[i] + 0.5 sum_children g[child])
__global__ void __launch_bounds__(256) chain_wide_bwd_d_kernel(const harp_mesh_chain A, const WideWs W) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.V0) return;
  const float* G = W.G + (size_t)b * (A.V0 + A.E0) * 3;
  const V3 v = subdivide_bwd_vertex(G, A.sub_off, A.sub_idx, i);
  st(A.g_v0 + ((size_t)b * A.V0 + i) * 3, v);
}

}  // namespace

int harp_detail_chain_wide_tail(const harp_mesh_chain& a, int clear_grads, float* part_ws, hipStream_t stream) {
  const size_t need = (size_t)(a.V0 + a.E0) * 3 * sizeof(float);
  // HARP_WIDE_LDS=<bytes>: LDS fence of the two forward kernels (0 = what they need; see harp_lds_fence / hand_front.hip)
  static size_t fa = 0, fb = 0;
  const size_t la = harp_lds_fence((const void*)chain_wide_a_kernel, "HARP_WIDE_LDS", 0, need, &fa);
  const size_t lb = harp_lds_fence((const void*)chain_wide_b_kernel, "HARP_WIDE_LDS", 0, need, &fb);
  hipLaunchKernelGGL(chain_wide_a_kernel, dim3(a.B * kParts), dim3(kChainThreads), la, stream, a, part_ws, clear_grads);
  hipLaunchKernelGGL(chain_wide_b_kernel, dim3(a.B * kParts), dim3(kChainThreads), lb, stream, a, part_ws);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// backward A - C: leaves dL/d(subdivided vertices) in the scratch's G rows (returned); the caller finishes with SubdivideMeshes backward
int harp_detail_chain_wide_bwd(const harp_mesh_chain& a, float* part_ws, const float** G_out, hipStream_t stream) {
  const int V = a.V0 + a.E0;
  const WideWs W = wide_ws(part_ws, a.B, V);
  const size_t lds2 = (size_t)V * 6 * sizeof(float);
  // dynamic LDS above 64 KB has to be requested (2 buffers of V*12 B: 74 KB hand, 98 KB arm); per-device attribute, set on every call
  if (hipFuncSetAttribute((const void*)chain_wide_bwd_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
      hipFuncSetAttribute((const void*)chain_wide_bwd_c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
    return HARP_ERR_ARG;
  hipLaunchKernelGGL(chain_wide_bwd_a_kernel, dim3(a.B * kParts), dim3(kChainThreads), 0, stream, a, W);
  hipLaunchKernelGGL(chain_wide_bwd_b_kernel, dim3(a.B * kParts), dim3(kChainThreads), lds2, stream, a, W);
  hipLaunchKernelGGL(chain_wide_bwd_c_kernel, dim3(a.B * kParts), dim3(kChainThreads), lds2, stream, a, W);
  HARP_CHECK_LAUNCH();
  if (G_out) *G_out = W.G;
  return HARP_OK;
}

extern "C" {

size_t harp_mesh_chain_wide_ws_floats(int B, int V) { return (size_t)B * kParts * (3 + 16) + (size_t)3 * B * V * 3; }

int harp_mesh_chain_fwd_wide(const harp_mesh_chain* a, int clear_grads, float* part_ws, hipStream_t stream) {
  if (!a || !a->edges0 || !a->vf_off || !a->vf_tri || !a->disp || a->B <= 0 || a->V0 <= 0 || a->E0 < 0 ||
      (a->V0 + a->E0 + kParts - 1) / kParts > kChainThreads || (a->V0 + a->E0) * 12 > 64 * 1024 || a->NJ * 3 > kChainThreads || !part_ws)
    return HARP_ERR_ARG;
  if (!a->verts_mm || !a->joints_mm || !a->cam_R || !a->cam_T || !a->joints_m || !a->vs || !a->n1 || !a->il1 || !a->vd || !a->n2 || !a->il2 ||
      !a->ndc_c || (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->ndc_l)) ||
      (clear_grads && (!a->g_vd || !a->g_joints_m)))
    return HARP_ERR_ARG;
  return harp_detail_chain_wide_tail(*a, clear_grads, part_ws, stream);
}

// the wide form of harp_mesh_chain_bwd (not for light_only: that variant is one small pass, harp_mesh_chain_bwd keeps it)
int harp_mesh_chain_bwd_wide(const harp_mesh_chain* a, float* part_ws, hipStream_t stream) {
  if (!a || !a->vf_off || !a->vf_tri || !a->disp || a->B <= 0 || a->V0 <= 0 || a->E0 < 0 || a->light_only ||
      (a->V0 + a->E0 + kParts - 1) / kParts > kChainThreads || (a->V0 + a->E0) * 24 > 160 * 1024 - 256 || a->NJ * 3 > kChainThreads || !part_ws)
    return HARP_ERR_ARG;
  if (!a->sub_off || !a->sub_idx || !a->vd || !a->vs || !a->n1 || !a->il1 || !a->cam_R || !a->cam_T || !a->g_vd || !a->g_ndc_c ||
      !a->g_joints_m || !a->g_joints_mm || !a->g_v0 || !a->g_cam_T || !a->g_disp || (a->has_normal_grad && (!a->n2 || !a->il2 || !a->g_n2)) ||
      (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->g_ndc_l || !a->g_light_R || !a->g_light_T ||
                     !a->g_light_pos)))
    return HARP_ERR_ARG;
  const int rc = harp_detail_chain_wide_bwd(*a, part_ws, nullptr, stream);
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(chain_wide_bwd_d_kernel, dim3((a->V0 + 255) / 256, a->B), dim3(256), 0, stream, *a, wide_ws(part_ws, a->B, a->V0 + a->E0));
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
