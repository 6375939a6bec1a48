// General kinematic-tree linear-blend skinning for gfx950 (forward + hand-written backward), used for the SMPL-X right-arm
// layer: replaces SMPLXARM.forward (hand_models_harp/body_models.py:2163-2390) = smplx.lbs over the full 10 475-vertex body
// followed by a slice to the 1026 arm vertices.  Here the model is sliced ONCE on the host (vertex arrays restricted to the arm,
// joint regressor folded into J_template / J_dirs), so the kernels only ever touch the ~10 % of the body the reference keeps.
//
// Same structure as csrc/lbs.hip (MANO) with run-time sizes: Rodrigues in the smplx matrix form
// (I + sin K + (1-cos) K^2, angle = |r + 1e-8|), arbitrary parents[] (parents[j] < j), an input->joint map (only the root, the
// right wrist and the 15 right-hand joints are driven), recentring on one chain joint, and output joints that are either chain
// joints or mesh vertices (finger tips).
#include "lbs_tree_body.h"

namespace {

using namespace lt;

// one wave per frame.  in_pose (B, n_in, 3); writes pose_map (B,NP), A (B,NJ,12), G (B,NJ,12), Jrest (B,NJ,3), Rloc (B,NJ,9)  (lbs_tree_body.h)
__global__ void __launch_bounds__(64) tree_joints_kernel(const harp_tree_model M, const float* __restrict__ in_pose,
                                                         const float* __restrict__ betas, const TreeWs W) {
  __shared__ JointsLds S;
  const int b = blockIdx.x;
  joints_body<64>(M, in_pose + (size_t)b * M.n_pose_in * 3, betas + b * M.NB, b, W, S);
}

// ---- dense contractions on the matrix cores ------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact f32, an fmaf chain): D(16x16) += A(16x4) B(4x16) with lane l holding
// A[l & 15][l >> 4], B[l >> 4][l & 15] and D[(l >> 4) * 4 + r][l & 15], r = 0..3.  These are the contractions BASELINE.json's
// north_star reserves MFMA for; on the SMPL-X arm (C5) their VALU forms were 17 % of the step (profiles/r02_c5_kernel_stats.txt:
// tree_skin 111 + 119 us, tree_gA 65 us, tree_gpm 63 us of 2.73 ms; tools/dev/micro/mfma_poseblend.hip: 39 -> 10 us for the blend).
constexpr int kBatch = 8;      // MFMA steps whose operand loads are in flight together

// v_posed(b, m) = v_template[m] + sum_k pose_map(b,k) posedirs_T[k][m] + sum_k betas(b,k) shapedirs_T[k][m]      (smplx.lbs, Appendix A.13)
// workgroup = 16 frames x 16 columns of NV*3, its 4 waves split the K = NP + NB reduction, LDS sum, one store.
__global__ void __launch_bounds__(256) tree_blend_mfma_kernel(const harp_tree_model M, const float* __restrict__ pose_map,
                                                              const float* __restrict__ betas, int B, float* __restrict__ vp) {
  __shared__ f32x4 s_acc[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NV3 = M.NV * 3, NP = (M.NJ - 1) * 9, NB = M.NB, K = NP + NB;
  const int col = blockIdx.x * 16 + (lane & 15), b0 = blockIdx.y * 16;
  const bool col_ok = col < NV3;
  const int row = b0 + (lane & 15);
  const bool row_ok = row < B;
  const int steps = (K + 3) / 4, per = (steps + 3) / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // kBatch steps' operands are requested together (clamped addresses, zeroed by select): the loop used to be load -> s_waitcnt vmcnt(0) ->
  // MFMA, one memory round trip per step — 31 in a row per wave, 41 us for a contraction that takes 10 (the blend-shape rows do not stay in L2
  // from one step to the next).  Padding steps multiply zeros: the accumulation order of the real ones is unchanged.
  const int colc = min(col, NV3 - 1), rowc = min(row, B - 1);
  const int s_lo = w * per, s_hi = min(steps, (w + 1) * per);
  for (int st0 = s_lo; st0 < s_hi; st0 += kBatch) {
    float a[kBatch], bv[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int k = 4 * (st0 + u) + (lane >> 4), kc = min(k, K - 1);
      const float* pa = (kc < NP) ? pose_map + (size_t)rowc * NP + kc : betas + rowc * NB + (kc - NP);
      const float* pb = (kc < NP) ? M.posedirs_T + (size_t)kc * NV3 + colc : M.shapedirs_T + (size_t)(kc - NP) * NV3 + colc;
      a[u] = *pa; bv[u] = *pb;
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const bool ok = (st0 + u < s_hi) && (4 * (st0 + u) + (lane >> 4) < K);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((ok && row_ok) ? a[u] : 0.f, (ok && col_ok) ? bv[u] : 0.f, acc, 0, 0, 0);
    }
  }
  s_acc[w][lane] = acc;
  __syncthreads();
  if (w == 0 && col_ok) {
    const f32x4 t = s_acc[0][lane] + s_acc[1][lane] + s_acc[2][lane] + s_acc[3][lane];
    const float base = M.v_template[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bb = b0 + (lane >> 4) * 4 + r;
      if (bb < B) vp[(size_t)bb * NV3 + col] = base + t[r];
    }
  }
}

// skinning: workgroup = 64 vertices x 4 frames; the weight rows of the 64 vertices (coalesced load, odd stride => conflict-free LDS
// rows) and the 4 frames' joint transforms are staged in LDS.  Forward: verts = ((T [vp;1]) - centre + transl) * 1000.
// Backward: g_vp = T^T g (g = g_verts * 1000); the outer products g (x) [vp;1] are contracted with the weights by tree_gA_mfma_kernel.
constexpr int kSkinV = 64, kSkinF = 4;
template <bool BWD>
__global__ void __launch_bounds__(256) tree_skin_kernel(const harp_tree_model M, const float* __restrict__ transl, const float* __restrict__ vp,
                                                        const float* __restrict__ A, const float* __restrict__ G, int B,
                                                        float* __restrict__ verts, const float* __restrict__ g_verts, float* __restrict__ g_vp) {
  extern __shared__ float smem[];
  const int NJ = M.NJ, NV = M.NV;
  float* s_W = smem;                        // [kSkinV][NJ]
  float* s_A = s_W + kSkinV * NJ;           // [kSkinF][NJ*12]
  float* s_c = s_A + kSkinF * NJ * 12;      // [kSkinF][3]  centre joint position
  const int v0 = blockIdx.x * kSkinV, b0 = blockIdx.y * kSkinF;
  const int nv = min(kSkinV, NV - v0), nb = min(kSkinF, B - b0);
  for (int i = threadIdx.x; i < nv * NJ; i += 256) s_W[i] = M.weights[(size_t)v0 * NJ + i];
  for (int i = threadIdx.x; i < nb * NJ * 12; i += 256) s_A[i] = A[(size_t)b0 * NJ * 12 + i];
  for (int i = threadIdx.x; i < nb * 3; i += 256)
    s_c[i] = (M.center_joint >= 0) ? G[((size_t)(b0 + i / 3) * NJ + M.center_joint) * 12 + (i % 3) * 4 + 3] : 0.f;
  __syncthreads();
  const int vl = threadIdx.x & (kSkinV - 1), f = threadIdx.x >> 6;
  if (vl >= nv || f >= nb) return;
  const int v = v0 + vl, b = b0 + f;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = 0.f;
  for (int j = 0; j < NJ; ++j) {
    const float wt = s_W[vl * NJ + j];
    if (wt != 0.f) {
#pragma unroll
      for (int k = 0; k < 12; ++k) T[k] += wt * s_A[(f * NJ + j) * 12 + k];
    }
  }
  if (!BWD) {
    const float* p = vp + ((size_t)b * NV + v) * 3;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    for (int r = 0; r < 3; ++r) {
      const float o = T[r * 4] * p0 + T[r * 4 + 1] * p1 + T[r * 4 + 2] * p2 + T[r * 4 + 3];
      verts[((size_t)b * NV + v) * 3 + r] = (o - s_c[f * 3 + r] + transl[3 * b + r]) * 1000.0f;
    }
  } else {
    float g[3];
    for (int r = 0; r < 3; ++r) g[r] = g_verts[((size_t)b * NV + v) * 3 + r] * 1000.0f;
    for (int c = 0; c < 3; ++c) g_vp[((size_t)b * NV + v) * 3 + c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
  }
}

// joints (B,n_out,3) mm: chain joints (G_j.t - centre + transl) * 1000, vertex joints = verts[vid] (already final)
__global__ void tree_joints_out_kernel(const harp_tree_model M, const float* __restrict__ G, const float* __restrict__ verts,
                                       const float* __restrict__ transl, int B, float* __restrict__ joints) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, no = M.n_joints_out;
  if (i >= B * no * 3) return;
  const int b = i / (no * 3), k = (i / 3) % no, c = i % 3, src = M.joint_src[k];
  if (src >= 0) {
    const float ctr = (M.center_joint >= 0) ? G[((size_t)b * M.NJ + M.center_joint) * 12 + c * 4 + 3] : 0.f;
    joints[i] = (G[((size_t)b * M.NJ + src) * 12 + c * 4 + 3] - ctr + transl[3 * b + c]) * 1000.0f;
  } else {
    joints[i] = verts[((size_t)b * M.NV + (-src - 1)) * 3 + c];
  }
}

// g_joints -> g_Gt (B,NJ,3) chain-joint translation gradients [metres] (zero-initialised by the caller) and tip part into g_verts
__global__ void tree_joints_bwd_kernel(const harp_tree_model M, const float* __restrict__ g_joints, int B, float* __restrict__ g_Gt,
                                       float* __restrict__ g_verts, float* __restrict__ z_betas) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, no = M.n_joints_out;
  if (i < B * M.NB) z_betas[i] = 0.f;                    // (the shape gradient is accumulated with atomics by a later launch)
  if (i >= B * no * 3) return;
  const int b = i / (no * 3), k = (i / 3) % no, c = i % 3, src = M.joint_src[k];
  if (src >= 0) atomicAdd(&g_Gt[((size_t)b * M.NJ + src) * 3 + c], g_joints[i] * 1000.0f);
  else atomicAdd(&g_verts[((size_t)b * M.NV + (-src - 1)) * 3 + c], g_joints[i]);
}

// per frame: s = 1000 * sum_v g_verts + sum_chain g_Gt ; g_transl = s ; g_Gt[centre] -= s
__global__ void __launch_bounds__(256) tree_center_bwd_kernel(const harp_tree_model M, const float* __restrict__ g_verts,
                                                              float* __restrict__ g_Gt, float* __restrict__ g_transl) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float a[3] = {0.f, 0.f, 0.f};
  for (int v = threadIdx.x; v < M.NV; v += 256)
    for (int c = 0; c < 3; ++c) a[c] += g_verts[((size_t)b * M.NV + v) * 3 + c] * 1000.0f;
  for (int j = threadIdx.x; j < M.NJ; j += 256)
    for (int c = 0; c < 3; ++c) a[c] += g_Gt[((size_t)b * M.NJ + j) * 3 + c];
  for (int c = 0; c < 3; ++c) {
    const float s = block_sum_256(a[c], red);
    if (threadIdx.x == 0) {
      g_transl[3 * b + c] = s;
      if (M.center_joint >= 0) g_Gt[((size_t)b * M.NJ + M.center_joint) * 3 + c] -= s;
    }
  }
}

// g_A(b, j, c) = sum_v W[v][j] * g(b,v,c/4) * [vp(b,v); 1](c%4)    (g = g_verts * 1000): per frame a (NJ x NV)(NV x 12) product.
// workgroup = (16 joints, one frame); 4 waves split the NV reduction; the outer products are formed on the fly (no (B,NV,12) buffer).
__global__ void __launch_bounds__(256) tree_gA_mfma_kernel(const harp_tree_model M, const float* __restrict__ g_verts, const float* __restrict__ vp,
                                                           float* __restrict__ g_A) {
  __shared__ f32x4 s_acc[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NJ = M.NJ, NV = M.NV;
  const int j0 = blockIdx.x * 16, b = blockIdx.y;
  const int ja = j0 + (lane & 15);             // A operand: row = joint
  const int cb = lane & 15;                    // B operand: column = c (12 used)
  const int steps = (NV + 3) / 4, per = (steps + 3) / 4;
  const float* gb = g_verts + (size_t)b * NV * 3;
  const float* pb = vp + (size_t)b * NV * 3;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int jac = min(ja, NJ - 1), cbc = min(cb, 11);
  const int s_lo = w * per, s_hi = min(steps, (w + 1) * per);
  for (int st0 = s_lo; st0 < s_hi; st0 += kBatch) {          // operands of kBatch steps in flight together (see tree_blend_mfma_kernel)
    float a[kBatch], g[kBatch], q[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int vc = min(4 * (st0 + u) + (lane >> 4), NV - 1);
      a[u] = M.weights[(size_t)vc * NJ + jac];
      g[u] = gb[3 * vc + (cbc >> 2)];
      q[u] = pb[3 * vc + min(cbc & 3, 2)];
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const bool ok = (st0 + u < s_hi) && (4 * (st0 + u) + (lane >> 4) < NV);
      const float bv = g[u] * 1000.0f * (((cb & 3) < 3) ? q[u] : 1.0f);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((ok && ja < NJ) ? a[u] : 0.f, (ok && cb < 12) ? bv : 0.f, acc, 0, 0, 0);
    }
  }
  s_acc[w][lane] = acc;
  __syncthreads();
  if (w == 0 && cb < 12) {
    const f32x4 t = s_acc[0][lane] + s_acc[1][lane] + s_acc[2][lane] + s_acc[3][lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + (lane >> 4) * 4 + r;
      if (j < NJ) g_A[((size_t)b * NJ + j) * 12 + cb] = t[r];
    }
  }
}

// g_pose_map(b, k) = sum_i g_vp(b,i) posedirs[i][k];  g_beta(b, k) += sum_i g_vp(b,i) shapedirs_T[k][i]    (i over NV*3)
// workgroup = (16 frames, 16 output columns, one of kSplit slices of the NV*3 reduction); 4 waves split the slice; LDS sum; the slices
// meet in memory with float atomics (g_pm / g_beta pre-zeroed).
constexpr int kSplitP = 8;
__global__ void __launch_bounds__(256) tree_gpm_mfma_kernel(const harp_tree_model M, const float* __restrict__ g_vp, int B, float* __restrict__ g_pm,
                                                            float* __restrict__ g_beta_b) {
  __shared__ f32x4 s_acc[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NV3 = M.NV * 3, NP = (M.NJ - 1) * 9, NB = M.NB, K = NP + NB;
  const int col = blockIdx.x * 16 + (lane & 15), b0 = blockIdx.y * 16;
  const int row = b0 + (lane & 15);
  const int steps = (NV3 + 3) / 4, per_wg = (steps + kSplitP - 1) / kSplitP, per = (per_wg + 3) / 4;
  const int s_lo = blockIdx.z * per_wg + w * per, s_hi = min(min(steps, (int)(blockIdx.z + 1) * per_wg), s_lo + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int rowc = min(row, B - 1), colc = min(col, K - 1);
  for (int st0 = s_lo; st0 < s_hi; st0 += kBatch) {          // operands of kBatch steps in flight together (see tree_blend_mfma_kernel)
    float a[kBatch], bv[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int ic = min(4 * (st0 + u) + (lane >> 4), NV3 - 1);
      const float* pb = (colc < NP) ? M.posedirs + (size_t)ic * NP + colc : M.shapedirs_T + (size_t)(colc - NP) * NV3 + ic;
      a[u] = g_vp[(size_t)rowc * NV3 + ic]; bv[u] = *pb;
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const bool ok = (st0 + u < s_hi) && (4 * (st0 + u) + (lane >> 4) < NV3);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((ok && row < B) ? a[u] : 0.f, (ok && col < K) ? bv[u] : 0.f, acc, 0, 0, 0);
    }
  }
  s_acc[w][lane] = acc;
  __syncthreads();
  if (w == 0 && col < K) {
    const f32x4 t = s_acc[0][lane] + s_acc[1][lane] + s_acc[2][lane] + s_acc[3][lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bb = b0 + (lane >> 4) * 4 + r;
      if (bb < B && t[r] != 0.f) {
        if (col < NP) atomicAdd(&g_pm[(size_t)bb * NP + col], t[r]);
        else atomicAdd(&g_beta_b[bb * NB + (col - NP)], t[r]);
      }
    }
  }
}

// one wave per frame: chain + Rodrigues backward (lbs_tree_body.h)
__global__ void __launch_bounds__(64) tree_chain_bwd_kernel(const harp_tree_model M, const float* __restrict__ in_pose, const TreeWs W,
                                                            float* __restrict__ g_in_pose, float* __restrict__ g_beta_b) {
  __shared__ ChainBwdLds S;
  const int b = blockIdx.x;
  chain_bwd_body<false>(M, in_pose + (size_t)b * M.n_pose_in * 3, b, W, g_in_pose + (size_t)b * M.n_pose_in * 3, g_beta_b + b * M.NB, S, nullptr, 0);
}


size_t skin_smem(const harp_tree_model* m) { return sizeof(float) * ((size_t)kSkinV * m->NJ + (size_t)kSkinF * (m->NJ * 12 + 3)); }

}  // namespace

// launchers shared with the fused arm front / back (arm_front.hip): the contractions that stay spread over the chip
int harp_detail_tree_blend(const harp_tree_model& m, float* ws, const float* betas, int B, hipStream_t stream) {
  const TreeWs w = tree_ws(&m, ws, B);
  hipLaunchKernelGGL(tree_blend_mfma_kernel, dim3((m.NV * 3 + 15) / 16, (B + 15) / 16), dim3(256), 0, stream, m, w.pm, betas, B, w.vp);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_detail_tree_gA_gpm(const harp_tree_model& m, float* ws, const float* g_verts, float* g_betas, int B, hipStream_t stream) {
  const TreeWs w = tree_ws(&m, ws, B);
  hipLaunchKernelGGL(tree_gA_mfma_kernel, dim3((m.NJ + 15) / 16, B), dim3(256), 0, stream, m, g_verts, w.vp, w.g_A);
  const int ncol = (m.NJ - 1) * 9 + m.NB;
  hipLaunchKernelGGL(tree_gpm_mfma_kernel, dim3((ncol + 15) / 16, (B + 15) / 16, kSplitP), dim3(256), 0, stream, m, w.g_vp, B, w.g_pm, g_betas);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

extern "C" {

size_t harp_lbs_tree_ws_floats(const harp_tree_model* m, int B) {
  const size_t NJ = m->NJ, NV = m->NV, NP = (NJ - 1) * 9;
  return (size_t)B * (NP + NJ * 12 * 2 + NJ * 3 + NJ * 9 + NV * 3 + NV * 3 + NJ * 12 + NP + NJ * 3);
}

int harp_lbs_tree_fwd(const harp_tree_model* m, const float* in_pose, const float* betas, const float* transl, int B, float* ws,
                      float* verts, float* joints, hipStream_t stream) {
  if (!m || !in_pose || !betas || !transl || !ws || !verts || !joints || B <= 0 || m->NJ > MAXJ || m->NB > MAXB) return HARP_ERR_ARG;
  const TreeWs w = tree_ws(m, ws, B);
  hipLaunchKernelGGL(tree_joints_kernel, dim3(B), dim3(64), 0, stream, *m, in_pose, betas, w);
  hipLaunchKernelGGL(tree_blend_mfma_kernel, dim3((m->NV * 3 + 15) / 16, (B + 15) / 16), dim3(256), 0, stream, *m, w.pm, betas, B, w.vp);
  hipLaunchKernelGGL(tree_skin_kernel<false>, dim3((m->NV + kSkinV - 1) / kSkinV, (B + kSkinF - 1) / kSkinF), dim3(256), skin_smem(m), stream, *m,
                     transl, w.vp, w.A, w.G, B, verts, nullptr, nullptr);
  hipLaunchKernelGGL(tree_joints_out_kernel, dim3((B * m->n_joints_out * 3 + 255) / 256), dim3(256), 0, stream, *m, w.G, verts, transl, B,
                     joints);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// g_verts (B,NV,3) is MODIFIED (vertex-joint gradients are folded in). Outputs: g_in_pose (B,n_pose_in,3), g_betas (B,NB), g_transl (B,3).
int harp_lbs_tree_bwd(const harp_tree_model* m, const float* in_pose, const float* betas, const float* transl, int B, float* ws,
                      float* g_verts, const float* g_joints, float* g_in_pose, float* g_betas, float* g_transl, hipStream_t stream) {
  if (!m || !in_pose || !betas || !ws || !g_verts || !g_joints || !g_in_pose || !g_betas || !g_transl) return HARP_ERR_ARG;
  const TreeWs w = tree_ws(m, ws, B);
  // (g_pm / g_Gt were cleared by harp_lbs_tree_fwd on this workspace, and again by the last backward call that consumed them)
  const int nth = max(B * m->n_joints_out * 3, B * m->NB);
  hipLaunchKernelGGL(tree_joints_bwd_kernel, dim3((nth + 255) / 256), dim3(256), 0, stream, *m, g_joints, B, w.g_Gt, g_verts, g_betas);
  hipLaunchKernelGGL(tree_center_bwd_kernel, dim3(B), dim3(256), 0, stream, *m, g_verts, w.g_Gt, g_transl);
  hipLaunchKernelGGL(tree_skin_kernel<true>, dim3((m->NV + kSkinV - 1) / kSkinV, (B + kSkinF - 1) / kSkinF), dim3(256), skin_smem(m), stream, *m,
                     transl, w.vp, w.A, w.G, B, nullptr, g_verts, w.g_vp);
  hipLaunchKernelGGL(tree_gA_mfma_kernel, dim3((m->NJ + 15) / 16, B), dim3(256), 0, stream, *m, g_verts, w.vp, w.g_A);
  const int ncol = (m->NJ - 1) * 9 + m->NB;
  hipLaunchKernelGGL(tree_gpm_mfma_kernel, dim3((ncol + 15) / 16, (B + 15) / 16, kSplitP), dim3(256), 0, stream, *m, w.g_vp, B, w.g_pm, g_betas);
  hipLaunchKernelGGL(tree_chain_bwd_kernel, dim3(B), dim3(64), 0, stream, *m, in_pose, w, g_in_pose, g_betas);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
