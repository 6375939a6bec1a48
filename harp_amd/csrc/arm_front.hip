// Fused per-frame FRONT and BACK of a fitting step for gfx950, SMPL-X arm path (use_arm): what hand_front.hip / hand_back.hip are for
// the MANO path.  A captured arm step was 32 kernel nodes (profiles/r04_j_c5_timeline_one_step.txt): schedule_next, frame_setup_fwd,
// tree_joints, tree_blend_mfma, tree_skin, tree_joints_out, mesh_chain_fwd in front of the rasterisers (130 us), and mesh_chain_bwd,
// tree_joints_bwd, tree_center_bwd, tree_skin<true>, tree_gA_mfma, tree_gpm_mfma, tree_chain_bwd, frame_setup_bwd behind the shader
// (176 us) — every small launch paying its cold dependent loads on 32 of 256 CUs.  Here:
//   front : arm_front_kernel   one 256-thread workgroup per frame: schedule row -> frame, parameter-row gathers, Rodrigues + joint
//                              regression + kinematic chain (lbs_tree_body.h)
//           tree_blend_mfma    UNCHANGED (lbs_tree.hip): the blend-shape contraction is shared by all frames on the matrix cores
//           arm_mid_kernel     one 1024-thread workgroup per frame: skinning, output joints, then the whole mesh chain on the frame's
//                              vertices while they are in LDS (subdivision, normals, displacement, normals, both projections, light camera)
//   back  : arm_back_kernel    one workgroup per frame: mesh-chain backward, then on the frame's g_v0 the joint split, the recentring /
//                              translation sums, the per-vertex skinning backward, the trans / cam / light / ambient scatter into the
//                              parameter tables' gradient rows and the step epilogue
//           tree_gA_mfma, tree_gpm_mfma   UNCHANGED: the two reductions over the vertices stay on the matrix cores
//           arm_chain_bwd_kernel          kinematic chain + Rodrigues backward, adding rot / wrist_pose / pose / shape to their rows
// 32 -> 23 kernel nodes per step.  Same arithmetic as the stand-alone kernels (lbs_tree.hip, chain.hip, glue.hip), which stay as the
// C-ABI building blocks and serve the API path.
//   reference: hand_models_harp/body_models.py:2163-2390 (SMPLXARM.forward), utils/visualize.py:16-88 (prepare_mesh),
//   renderer_helper.py:454-468, the row gathers params[...][fid] (utils/visualize.py:26-27, 37-40) and their autograd.
#include "chain_body.h"
#include "lbs_tree_body.h"

int harp_detail_tree_blend(const harp_tree_model& m, float* ws, const float* betas, int B, hipStream_t stream);
int harp_detail_tree_gA_gpm(const harp_tree_model& m, float* ws, const float* g_verts, float* g_betas, int B, hipStream_t stream);
int harp_detail_chain_wide_tail(const harp_mesh_chain& a, int clear_grads, float* part_ws, hipStream_t stream);
int harp_detail_chain_wide_bwd(const harp_mesh_chain& a, float* part_ws, const float** G_out, hipStream_t stream);

namespace {

using namespace lt;
using cb::kChainThreads;

constexpr int kFrontThreads = 256;

__global__ void __launch_bounds__(kFrontThreads) arm_front_kernel(const harp_arm_front H) {
  __shared__ JointsLds S;
  __shared__ float s_pose[64], s_beta[MAXB];
  const harp_mesh_chain& A = H.chain;
  const harp_tree_model& M = H.tree;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x, tid = threadIdx.x, B = A.B;
  // ---- optional step prologue (harp_step_frame): this workgroup's frame from the device schedule
  int f;
  if (H.step.schedule) {
    const int row = (int)((unsigned)H.step.sched_row[0] % (unsigned)H.step.n_rows);     // bumped by arm_back_kernel, a later launch
    f = H.step.schedule[(size_t)row * B + b];
    if (tid == 0) {
      const_cast<int32_t*>(H.fid)[b] = f;
      if (H.step.tfid_out) H.step.tfid_out[b] = H.step.tschedule ? H.step.tschedule[(size_t)row * B + b] : f - H.step.target_offset;
    }
  } else {
    f = H.fid[b];
  }
  // ---- frame set-up (glue.hip: frame_setup_fwd_kernel, arm rows [rot(3), wrist_pose(3), pose(45)], betas padded with zeros)
  const int ps = M.n_pose_in * 3, nbo = M.NB;
  if (tid < ps) {
    const float p = (tid < 3) ? T.rot[f * 3 + tid] : (tid < 6) ? T.wrist_pose[f * 3 + tid - 3] : T.pose[f * 45 + tid - 6];
    s_pose[tid] = p; H.pose_in[b * ps + tid] = p;
  } else if (tid >= 64 && tid < 64 + nbo) {
    const int k = tid - 64;
    const float v = (k < 10) ? T.shape[k] : 0.f;
    s_beta[k] = v; H.betas[b * nbo + k] = v;
  } else if (tid >= 128 && tid < 131) {
    const int k = tid - 128;
    H.trans_b[b * 3 + k] = T.trans[f * 3 + k];
    const int lf = T.share_light ? 0 : f;
    H.light_pos[b * 3 + k] = T.light_positions[lf * 3 + k];
  } else if (tid == 192) {
    const float c0 = T.cam[f * 3], c1 = T.cam[f * 3 + 1], c2 = T.cam[f * 3 + 2];
    const float ct[3] = {-c1, -c2, 2.0f * A.focal / ((float)A.S * c0 + 1e-9f)};
    const float R[9] = {-1.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 1.f};
    for (int k = 0; k < 9; ++k) H.cam_R[b * 9 + k] = R[k];
    for (int k = 0; k < 3; ++k) H.cam_T[b * 3 + k] = ct[k];
  } else if (tid == 193 && b == 0) {
    if (H.self_shadow) {
      const float amb = 1.0f / (1.0f + expf(-T.amb_ratio[0]));            // nn.Sigmoid()(params['amb_ratio'])
      for (int c = 0; c < 3; ++c) { H.colors[c] = amb; H.colors[3 + c] = 1.0f - amb; H.colors[6 + c] = 0.f; }
    } else {
      for (int c = 0; c < 3; ++c) { H.colors[c] = 0.5f; H.colors[3 + c] = 0.4f; H.colors[6 + c] = 0.1f; }   // renderer_helper.py:70-73
    }
  }
  __syncthreads();
  joints_body<kFrontThreads>(M, s_pose, s_beta, b, tree_ws(&M, H.lbs_ws, B), S);
}

// T = sum_j W[v][j] A[j] from the transposed weights (NJ, NV): lane = vertex, so every row j is one coalesced read; kSkinBatch rows in
// flight (the 55 rows of the arm in two round trips).  Ascending j with the zero weights skipped: the summation order of tree_skin_kernel.
constexpr int kSkinBatch = 28;
__device__ __forceinline__ void skin_transform(const float* __restrict__ wT, const float* s_A, int NJ, int NV, int v, float Tm[12]) {
#pragma unroll
  for (int k = 0; k < 12; ++k) Tm[k] = 0.f;
  for (int j0 = 0; j0 < NJ; j0 += kSkinBatch) {
    float wj[kSkinBatch];
#pragma unroll
    for (int u = 0; u < kSkinBatch; ++u) wj[u] = wT[(size_t)min(j0 + u, NJ - 1) * NV + v];
#pragma unroll
    for (int u = 0; u < kSkinBatch; ++u)
      if (j0 + u < NJ && wj[u] != 0.f) {
#pragma unroll
        for (int k = 0; k < 12; ++k) Tm[k] += wj[u] * s_A[(j0 + u) * 12 + k];
      }
  }
}

__global__ void __launch_bounds__(kChainThreads) arm_mid_kernel(const harp_arm_front H) {
  extern __shared__ float s_dyn[];             // V*3 positions
  __shared__ float s_A[MAXJ * 12], s_ctr[3], s_tr[3], s_cam[12], s_lpos[3];
  const harp_mesh_chain& A = H.chain;
  const harp_tree_model& M = H.tree;
  const int b = blockIdx.x, tid = threadIdx.x, B = A.B;
  const int V = A.V0 + A.E0, NJ = M.NJ, NV = M.NV;
  const TreeWs W = tree_ws(&M, H.lbs_ws, B);
  float* s_p = s_dyn;
  if (H.step.clear_mesh_grads) {               // the two gradient segments the key-point / mesh terms accumulate into (they start after this kernel)
    float* gv = const_cast<float*>(A.g_vd) + (size_t)b * V * 3;
    for (int i = tid; i < V * 3; i += kChainThreads) gv[i] = 0.f;
    if (tid < A.NJ * 3) const_cast<float*>(A.g_joints_m)[(size_t)b * A.NJ * 3 + tid] = 0.f;
  }
  for (int i = tid; i < NJ * 12; i += kChainThreads) s_A[i] = W.A[(size_t)b * NJ * 12 + i];
  if (tid >= 960 && tid < 963) {
    const int c = tid - 960;
    s_ctr[c] = (M.center_joint >= 0) ? W.G[((size_t)b * NJ + M.center_joint) * 12 + c * 4 + 3] : 0.f;
    s_tr[c] = H.trans_b[b * 3 + c];
    s_lpos[c] = H.light_pos[b * 3 + c];
  } else if (tid >= 896 && tid < 908) {
    const int k = tid - 896;
    s_cam[k] = (k < 9) ? H.cam_R[b * 9 + k] : H.cam_T[b * 3 + k - 9];
  }
  __syncthreads();
  // ---- skinning (lbs_tree.hip: tree_skin_kernel<false>): verts = ((T [vp;1]) - centre + transl) * 1000
  for (int v = tid; v < NV; v += kChainThreads) {
    const float* p = W.vp + ((size_t)b * NV + v) * 3;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    float Tm[12];
    skin_transform(H.weights_T, s_A, NJ, NV, v, Tm);
    float* vo = (float*)A.verts_mm + ((size_t)b * NV + v) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float o = Tm[r * 4] * p0 + Tm[r * 4 + 1] * p1 + Tm[r * 4 + 2] * p2 + Tm[r * 4 + 3];
      const float mm = (o - s_ctr[r] + s_tr[r]) * 1000.0f;
      vo[r] = mm;
      s_p[3 * v + r] = mm * 1e-3f;
    }
  }
  __syncthreads();
  // ---- output joints (tree_joints_out_kernel), millimetres and metres (visualize.py:46)
  const int no = M.n_joints_out;
  if (tid < no * 3) {
    const int k = tid / 3, c = tid % 3, src = M.joint_src[k];
    const float jm = (src >= 0) ? (W.G[((size_t)b * NJ + src) * 12 + c * 4 + 3] - s_ctr[c] + s_tr[c]) * 1000.0f
                                : A.verts_mm[((size_t)b * NV + (-src - 1)) * 3 + c];      // (written above by this workgroup, behind the barrier)
    ((float*)A.joints_mm)[(size_t)b * no * 3 + tid] = jm;
    A.joints_m[(size_t)b * no * 3 + tid] = jm * 1e-3f;
  }
  // ---- mesh chain (chain_body.h); its first statement after the (skipped) load is a barrier
  cb::mesh_chain_fwd_body(A, s_p, b, true, s_cam, s_cam + 9, s_lpos);
}

__global__ void __launch_bounds__(kChainThreads) arm_back_kernel(const harp_arm_front H, const float* __restrict__ g_colors,
                                                                 float* __restrict__ g_betas) {
  extern __shared__ float s_mem[];             // the chain backward's three V*3 buffers
  __shared__ float s_A[MAXJ * 12], s_gGt[MAXJ * 3], s_red[16 * 3], s_tot[3];
  const harp_mesh_chain& A = H.chain;
  const harp_tree_model& M = H.tree;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x, tid = threadIdx.x, B = A.B;
  const int NJ = M.NJ, NV = M.NV, NB = M.NB;
  const TreeWs W = tree_ws(&M, H.lbs_ws, B);
  const bool lean = A.light_only != 0;         // appearance-only stage: light-view part, light / ambient scatter and the step epilogue only

  cb::mesh_chain_bwd_body(A, s_mem, b);        // ... -> g_v0 (this frame's NV x 3), g_cam_T, g_light_pos, g_disp
  __threadfence_block();
  __syncthreads();
  if (!lean) {
    for (int i = tid; i < NJ * 12; i += kChainThreads) s_A[i] = W.A[(size_t)b * NJ * 12 + i];
    if (tid < NJ * 3) s_gGt[tid] = 0.f;
    if (tid >= 256 && tid < 256 + NB) g_betas[b * NB + tid - 256] = 0.f;      // (the shape gradient is accumulated with atomics by the next launch)
    __syncthreads();
    // ---- joint gradients (tree_joints_bwd_kernel): chain joints -> g_Gt [metres], vertex joints (finger tips) -> their vertices
    const int no = M.n_joints_out;
    float* gv0 = A.g_v0 + (size_t)b * NV * 3;
    if (tid < no * 3) {
      const int k = tid / 3, c = tid % 3, src = M.joint_src[k];
      const float gj = A.g_joints_m[(size_t)b * no * 3 + tid] * 1e-3f;      // (= g_joints_mm)
      if (src >= 0) atomicAdd(&s_gGt[src * 3 + c], gj * 1000.0f);
      else atomicAdd(&gv0[(-src - 1) * 3 + c], gj);
    }
    __threadfence_block();
    __syncthreads();
    // ---- recentring / translation sums (tree_center_bwd_kernel) and the skinning backward (tree_skin_kernel<true>): g_vp = T^T g
    float a3[3] = {0.f, 0.f, 0.f};
    for (int v = tid; v < NV; v += kChainThreads) {
      float g[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) { g[r] = gv0[3 * v + r] * 1000.0f; a3[r] += g[r]; }
      float Tm[12];
      skin_transform(H.weights_T, s_A, NJ, NV, v, Tm);
      float* gvp = W.g_vp + ((size_t)b * NV + v) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) gvp[c] = Tm[c] * g[0] + Tm[4 + c] * g[1] + Tm[8 + c] * g[2];
    }
    if (tid < NJ)
      for (int c = 0; c < 3; ++c) a3[c] += s_gGt[tid * 3 + c];
    cb::block_sum_n<3>(a3, s_red, s_tot);
    if (tid < 3 && M.center_joint >= 0) s_gGt[M.center_joint * 3 + tid] -= s_tot[tid];
    __syncthreads();
    if (tid < NJ * 3) W.g_Gt[(size_t)b * NJ * 3 + tid] = s_gGt[tid];
  }
  // ---- scatter what is final by now into the gradient rows of the parameter tables (frame_setup_bwd_kernel's trans / cam / light part);
  //      duplicates of a frame in one batch are legal and the shared light is summed over the frames -> atomics
  const int f = H.fid[b];
  if (tid < 3) {
    const int k = tid;
    if (T.g_trans && !lean) atomicAdd(T.g_trans + f * 3 + k, s_tot[k]);
    if (T.g_cam && !lean) {
      if (k == 0) {
        const float c0 = T.cam[f * 3];
        const float den = (float)A.S * c0 + 1e-9f;
        atomicAdd(T.g_cam + f * 3, A.g_cam_T[b * 3 + 2] * (-2.0f * A.focal * (float)A.S / (den * den)));
      } else {
        atomicAdd(T.g_cam + f * 3 + k, -A.g_cam_T[b * 3 + (k - 1)]);
      }
    }
    if (g_colors && A.g_light_pos && T.g_light_positions) {
      const int lf = T.share_light ? 0 : f;
      atomicAdd(T.g_light_positions + lf * 3 + k, A.g_light_pos[b * 3 + k]);
    }
  } else if (tid == 64 && b == 0 && H.self_shadow && g_colors && T.g_amb_ratio) {
    const float amb = 1.0f / (1.0f + expf(-T.amb_ratio[0]));
    const float g_amb = (g_colors[0] + g_colors[1] + g_colors[2]) - (g_colors[3] + g_colors[4] + g_colors[5]);
    atomicAdd(T.g_amb_ratio, g_amb * amb * (1.0f - amb));
  }
  // ---- optional step epilogue (harp_step_frame): every kernel that reads the schedule row, adds to the loss vector or reads the draw
  //      counter is an EARLIER launch of the step (stream order / joins), so one workgroup can turn the three over for the next step
  if (b == 0) {
    const harp_step_frame& E = H.step;
    if ((tid >> 6) == 2) {                          // wave 2 (lanes 128 .. 191), whole: the wave sum below needs every lane
      const int k = tid - 128;
      const bool on = E.loss && k < E.n_loss;       // n_loss <= 64 (checked by the launcher)
      const float v = on ? E.loss[k] : 0.f;
      if (on) {
        if (E.loss_out) E.loss_out[k] = v;
        E.loss[k] = 0.f;
      }
      if (E.loss_w && E.loss_total) {
        const float tot = wave_sum_u(on ? E.loss_w[k] * v : 0.f);
        if (k == 0) E.loss_total[0] += tot;
      }
    } else if (tid == 192 && E.schedule) {
      E.sched_row[0] = (int)((unsigned)E.sched_row[0] % (unsigned)E.n_rows) + 1;
    } else if (tid == 193 && E.draw_counter) {
      E.draw_counter[0] += 1;
    }
  }
}

// ---- wide forms (csrc/chain_wide.hip): skinning / skinning backward on kChainParts workgroups per frame (a contiguous quarter of the NV
//      vertices each), around the wide mesh chain.  kArmWide threads >= ceil(NV / kChainParts) (checked by the launchers).
constexpr int kArmWide = 320;

__global__ void __launch_bounds__(kArmWide) arm_skin_wide_kernel(const harp_arm_front H) {
  __shared__ float s_A[MAXJ * 12], s_ctr[3], s_tr[3];
  __shared__ int s_jsrc[64];
  const harp_mesh_chain& A = H.chain;
  const harp_tree_model& M = H.tree;
  const int b = blockIdx.x / cb::kChainParts, part = blockIdx.x % cb::kChainParts, tid = threadIdx.x, B = A.B;
  const int NJ = M.NJ, NV = M.NV, no = M.n_joints_out, per = (NV + cb::kChainParts - 1) / cb::kChainParts;
  const TreeWs W = tree_ws(&M, H.lbs_ws, B);
  for (int i = tid; i < NJ * 12; i += kArmWide) s_A[i] = W.A[(size_t)b * NJ * 12 + i];
  if (tid < 3) {
    s_ctr[tid] = (M.center_joint >= 0) ? W.G[((size_t)b * NJ + M.center_joint) * 12 + tid * 4 + 3] : 0.f;
    s_tr[tid] = H.trans_b[b * 3 + tid];
  } else if (tid >= 64 && tid < 64 + no) {
    s_jsrc[tid - 64] = M.joint_src[tid - 64];
  }
  __syncthreads();
  const int v = part * per + tid;
  if (tid < per && v < NV) {
    const float* p = W.vp + ((size_t)b * NV + v) * 3;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    float Tm[12];
    skin_transform(H.weights_T, s_A, NJ, NV, v, Tm);
    float* vo = (float*)A.verts_mm + ((size_t)b * NV + v) * 3;
    float mm[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float o = Tm[r * 4] * p0 + Tm[r * 4 + 1] * p1 + Tm[r * 4 + 2] * p2 + Tm[r * 4 + 3];
      mm[r] = (o - s_ctr[r] + s_tr[r]) * 1000.0f;
      vo[r] = mm[r];
    }
    for (int k = 0; k < no; ++k)              // vertex joints (finger tips): written by the vertex's owner
      if (s_jsrc[k] == -v - 1)
        for (int r = 0; r < 3; ++r) ((float*)A.joints_mm)[((size_t)b * no + k) * 3 + r] = mm[r];
  }
  if (part == 0 && tid >= 256 && tid < 256 + no) {           // chain joints (tree_joints_out_kernel)
    const int k = tid - 256, src = s_jsrc[k];
    if (src >= 0)
      for (int c = 0; c < 3; ++c)
        ((float*)A.joints_mm)[((size_t)b * no + k) * 3 + c] = (W.G[((size_t)b * NJ + src) * 12 + c * 4 + 3] - s_ctr[c] + s_tr[c]) * 1000.0f;
  }
}

// G: dL/d(subdivided vertices) left by the wide mesh-chain backward.  SubdivideMeshes backward, joint split, recentring / translation sums
// (partial, joined with atomics: g_Gt of the frame is all-zero on entry), skinning backward; part 0: table scatter + step epilogue.
__global__ void __launch_bounds__(kArmWide) arm_back_wide_kernel(const harp_arm_front H, const float* __restrict__ G,
                                                                 const float* __restrict__ g_colors, float* __restrict__ g_betas) {
  __shared__ float s_A[MAXJ * 12], s_gGt[MAXJ * 3], s_red[kArmWide / 64][3], s_tot[3];
  __shared__ int s_jsrc[64];
  const harp_mesh_chain& A = H.chain;
  const harp_tree_model& M = H.tree;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x / cb::kChainParts, part = blockIdx.x % cb::kChainParts, tid = threadIdx.x, B = A.B;
  const bool lead = part == 0;
  const int NJ = M.NJ, NV = M.NV, NB = M.NB, no = M.n_joints_out, per = (NV + cb::kChainParts - 1) / cb::kChainParts;
  const int V = A.V0 + A.E0;
  const TreeWs W = tree_ws(&M, H.lbs_ws, B);
  for (int i = tid; i < NJ * 12; i += kArmWide) s_A[i] = W.A[(size_t)b * NJ * 12 + i];
  if (tid < NJ * 3) s_gGt[tid] = 0.f;
  if (lead && tid >= 256 && tid < 256 + NB) g_betas[b * NB + tid - 256] = 0.f;      // (the shape gradient is accumulated with atomics by the next launch)
  if (tid >= 64 && tid < 64 + no) s_jsrc[tid - 64] = M.joint_src[tid - 64];
  __syncthreads();
  // ---- joint gradients (tree_joints_bwd_kernel): chain joints -> g_Gt [metres] (part 0), vertex joints -> their vertices (by the owner, below)
  if (lead && tid < no * 3) {
    const int k = tid / 3, c = tid % 3, src = s_jsrc[k];
    if (src >= 0) atomicAdd(&s_gGt[src * 3 + c], A.g_joints_m[(size_t)b * no * 3 + tid] * 1e-3f * 1000.0f);
  }
  __syncthreads();
  float a3[3] = {0.f, 0.f, 0.f};
  const int v = part * per + tid;
  if (tid < per && v < NV) {
    const cb::V3 g0 = cb::subdivide_bwd_vertex(G + (size_t)b * V * 3, A.sub_off, A.sub_idx, v);
    float g[3] = {g0.x, g0.y, g0.z};
    for (int k = 0; k < no; ++k)
      if (s_jsrc[k] == -v - 1)
        for (int r = 0; r < 3; ++r) g[r] += A.g_joints_m[((size_t)b * no + k) * 3 + r] * 1e-3f;      // (= g_joints_mm)
    float* gv0 = A.g_v0 + ((size_t)b * NV + v) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) { gv0[r] = g[r]; g[r] *= 1000.0f; a3[r] = g[r]; }
    float Tm[12];
    skin_transform(H.weights_T, s_A, NJ, NV, v, Tm);
    float* gvp = W.g_vp + ((size_t)b * NV + v) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) gvp[c] = Tm[c] * g[0] + Tm[4 + c] * g[1] + Tm[8 + c] * g[2];
  }
  if (lead && tid >= 256 && tid < 256 + NJ)
    for (int c = 0; c < 3; ++c) a3[c] += s_gGt[(tid - 256) * 3 + c];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sw = wave_sum_u(a3[c]);
    if ((tid & 63) == 0) s_red[tid >> 6][c] = sw;
  }
  __syncthreads();
  if (tid < 3) {
    float sum = 0.f;
    for (int w = 0; w < kArmWide / 64; ++w) sum += s_red[w][tid];
    s_tot[tid] = sum;
  }
  __syncthreads();
  // g_Gt rows of the frame: chain-joint gradients (part 0) and -(this part's share of the translation sum) on the centre joint
  float* gGt = W.g_Gt + (size_t)b * NJ * 3;
  if (lead && tid < NJ * 3 && s_gGt[tid] != 0.f) atomicAdd(gGt + tid, s_gGt[tid]);
  if (tid >= 64 && tid < 67 && M.center_joint >= 0) atomicAdd(gGt + M.center_joint * 3 + tid - 64, -s_tot[tid - 64]);
  // ---- scatter into the gradient rows of the parameter tables (frame_setup_bwd_kernel's trans / cam / light part)
  const int f = H.fid[b];
  if (tid < 3) {
    const int k = tid;
    if (T.g_trans) atomicAdd(T.g_trans + f * 3 + k, s_tot[k]);
    if (T.g_cam && lead) {
      if (k == 0) {
        const float c0 = T.cam[f * 3];
        const float den = (float)A.S * c0 + 1e-9f;
        atomicAdd(T.g_cam + f * 3, A.g_cam_T[b * 3 + 2] * (-2.0f * A.focal * (float)A.S / (den * den)));
      } else {
        atomicAdd(T.g_cam + f * 3 + k, -A.g_cam_T[b * 3 + (k - 1)]);
      }
    }
    if (lead && g_colors && A.g_light_pos && T.g_light_positions) {
      const int lf = T.share_light ? 0 : f;
      atomicAdd(T.g_light_positions + lf * 3 + k, A.g_light_pos[b * 3 + k]);
    }
  } else if (tid == 128 && lead && b == 0 && H.self_shadow && g_colors && T.g_amb_ratio) {
    const float amb = 1.0f / (1.0f + expf(-T.amb_ratio[0]));
    const float g_amb = (g_colors[0] + g_colors[1] + g_colors[2]) - (g_colors[3] + g_colors[4] + g_colors[5]);
    atomicAdd(T.g_amb_ratio, g_amb * amb * (1.0f - amb));
  }
  if (b == 0 && lead) {                             // step epilogue (harp_step_frame), as in arm_back_kernel
    const harp_step_frame& E = H.step;
    if ((tid >> 6) == 3) {                          // wave 3 (lanes 192 .. 255), whole: the wave sum below needs every lane
      const int k = tid - 192;
      const bool on = E.loss && k < E.n_loss;
      const float vv = on ? E.loss[k] : 0.f;
      if (on) {
        if (E.loss_out) E.loss_out[k] = vv;
        E.loss[k] = 0.f;
      }
      if (E.loss_w && E.loss_total) {
        const float tot = wave_sum_u(on ? E.loss_w[k] * vv : 0.f);
        if (k == 0) E.loss_total[0] += tot;
      }
    } else if (tid == 256 && E.schedule) {
      E.sched_row[0] = (int)((unsigned)E.sched_row[0] % (unsigned)E.n_rows) + 1;
    } else if (tid == 257 && E.draw_counter) {
      E.draw_counter[0] += 1;
    }
  }
}

// one wave per frame: chain + Rodrigues backward, adding rot / wrist_pose / pose / shape straight to their rows (lbs_tree_body.h)
__global__ void __launch_bounds__(64) arm_chain_bwd_kernel(const harp_arm_front H, float* __restrict__ g_pose_in, float* __restrict__ g_betas) {
  __shared__ ChainBwdLds S;
  const harp_tree_model& M = H.tree;
  const int b = blockIdx.x, ps = M.n_pose_in * 3;
  chain_bwd_body<true>(M, H.pose_in + (size_t)b * ps, b, tree_ws(&M, H.lbs_ws, H.chain.B), g_pose_in + (size_t)b * ps, g_betas + b * M.NB, S,
                       &H.tables, H.fid[b]);
}

bool arm_ok(const harp_arm_front* h) {
  if (!h) return false;
  const harp_mesh_chain& a = h->chain;
  const harp_tree_model& m = h->tree;
  return a.edges0 && a.vf_off && a.vf_tri && a.disp && a.B > 0 && a.E0 >= 0 && a.V0 == m.NV && a.NJ == m.n_joints_out &&
         a.V0 + a.E0 <= harp_mesh_chain_max_vertices() && a.NJ * 3 <= kFrontThreads && m.NJ > 0 && m.NJ <= MAXJ && m.NB >= 10 && m.NB <= MAXB &&
         m.n_pose_in == 17 && m.v_template && m.shapedirs_T && m.posedirs_T && m.posedirs && m.J_template && m.J_dirs && m.weights &&
         m.pose_mean && m.parents && m.pose_src && m.joint_src && h->weights_T && h->fid && h->pose_in && h->betas && h->trans_b && h->cam_R &&
         h->cam_T && h->light_pos && h->colors && h->lbs_ws && h->tables.wrist_pose && h->tables.n_betas_out == m.NB &&
         !(h->step.schedule && (!h->step.sched_row || h->step.n_rows <= 0));
}

}  // namespace

extern "C" {

int harp_arm_front_fwd(const harp_arm_front* h, hipStream_t stream) {
  if (!arm_ok(h)) return HARP_ERR_ARG;
  const harp_mesh_chain& a = h->chain;
  if (!a.verts_mm || !a.joints_mm || !a.joints_m || !a.vs || !a.n1 || !a.il1 || !a.vd || !a.n2 || !a.il2 || !a.ndc_c ||
      (a.shadow && (!a.centroid || !a.light_R || !a.light_T || !a.ndc_l)) || (h->step.clear_mesh_grads && (!a.g_vd || !a.g_joints_m)))
    return HARP_ERR_ARG;
  hipLaunchKernelGGL(arm_front_kernel, dim3(a.B), dim3(kFrontThreads), 0, stream, *h);
  const int rc = harp_detail_tree_blend(h->tree, h->lbs_ws, h->betas, a.B, stream);
  if (rc != HARP_OK) return rc;
  const size_t lds = (size_t)(a.V0 + a.E0) * 3 * sizeof(float);
  hipLaunchKernelGGL(arm_mid_kernel, dim3(a.B), dim3(kChainThreads), lds, stream, *h);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_arm_back_bwd(const harp_arm_front* h, const float* g_colors, float* g_pose_scratch, float* g_betas_scratch, hipStream_t stream) {
  if (!arm_ok(h) || !g_pose_scratch || !g_betas_scratch) return HARP_ERR_ARG;
  const harp_mesh_chain* a = &h->chain;
  if (!a->sub_off || !a->sub_idx || !a->vd || !a->vs || !a->n1 || !a->il1 || !a->cam_R || !a->cam_T || !a->g_vd || !a->g_ndc_c ||
      !a->g_joints_m || !a->g_joints_mm || !a->g_v0 || !a->g_cam_T || !a->g_disp || (a->has_normal_grad && (!a->n2 || !a->il2 || !a->g_n2)) ||
      (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->g_ndc_l || !a->g_light_R || !a->g_light_T ||
                     !a->g_light_pos)))
    return HARP_ERR_ARG;
  if (h->step.loss && (h->step.n_loss < 0 || h->step.n_loss > 64)) return HARP_ERR_ARG;
  const size_t lds = (size_t)(a->V0 + a->E0) * 9 * sizeof(float);
  // dynamic LDS above 64 KB has to be requested (147 KB on the arm mesh); per-device attribute, set on every call
  if (hipFuncSetAttribute((const void*)arm_back_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return HARP_ERR_ARG;
  hipLaunchKernelGGL(arm_back_kernel, dim3(a->B), dim3(kChainThreads), lds, stream, *h, g_colors, g_betas_scratch);
  HARP_CHECK_LAUNCH();
  if (a->light_only) return HARP_OK;           // no arm-layer backward: nothing of it reaches the appearance optimiser's parameters
  const int rc = harp_detail_tree_gA_gpm(h->tree, h->lbs_ws, a->g_v0, g_betas_scratch, a->B, stream);
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(arm_chain_bwd_kernel, dim3(a->B), dim3(64), 0, stream, *h, g_pose_scratch, g_betas_scratch);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// Wide forms: the per-frame kernels on four workgroups per frame around the wide mesh chain (csrc/chain_wide.hip).  Front: joint chain |
// blend (MFMA) | skinning + output joints | chain A | chain B.  Back: chain backward A, B, C | SubdivideMeshes backward + joint split + skinning
// backward + scatter | gA, gpm (MFMA) | kinematic chain backward.  part_ws: harp_mesh_chain_wide_ws_floats(B, V0 + E0).
static bool arm_wide_ok(const harp_arm_front* h) {
  const int V = h->chain.V0 + h->chain.E0;
  return (h->tree.NV + cb::kChainParts - 1) / cb::kChainParts <= kArmWide && (V + cb::kChainParts - 1) / cb::kChainParts <= kChainThreads &&
         h->tree.NJ <= 64 && h->tree.n_joints_out <= 64 && h->tree.NB <= 64;
}

int harp_arm_front_wide_fwd(const harp_arm_front* h, float* part_ws, hipStream_t stream) {
  if (!arm_ok(h) || !part_ws || !arm_wide_ok(h) || (h->chain.V0 + h->chain.E0) * 12 > 64 * 1024) return HARP_ERR_ARG;
  const harp_mesh_chain& a = h->chain;
  if (!a.verts_mm || !a.joints_mm || !a.joints_m || !a.vs || !a.n1 || !a.il1 || !a.vd || !a.n2 || !a.il2 || !a.ndc_c || !a.cam_R || !a.cam_T ||
      (a.shadow && (!a.centroid || !a.light_R || !a.light_T || !a.ndc_l || !a.light_pos)) || (h->step.clear_mesh_grads && (!a.g_vd || !a.g_joints_m)))
    return HARP_ERR_ARG;
  hipLaunchKernelGGL(arm_front_kernel, dim3(a.B), dim3(kFrontThreads), 0, stream, *h);
  const int rc = harp_detail_tree_blend(h->tree, h->lbs_ws, h->betas, a.B, stream);
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(arm_skin_wide_kernel, dim3(a.B * cb::kChainParts), dim3(kArmWide), 0, stream, *h);
  HARP_CHECK_LAUNCH();
  return harp_detail_chain_wide_tail(a, h->step.clear_mesh_grads, part_ws, stream);
}

int harp_arm_back_wide_bwd(const harp_arm_front* h, const float* g_colors, float* g_pose_scratch, float* g_betas_scratch, float* part_ws,
                           hipStream_t stream) {
  if (!h || !part_ws) return HARP_ERR_ARG;
  if (h->chain.light_only) return harp_arm_back_bwd(h, g_colors, g_pose_scratch, g_betas_scratch, stream);
  if (!arm_ok(h) || !g_pose_scratch || !g_betas_scratch || !arm_wide_ok(h) || (h->chain.V0 + h->chain.E0) * 24 > 160 * 1024 - 256) return HARP_ERR_ARG;
  const harp_mesh_chain* a = &h->chain;
  if (!a->sub_off || !a->sub_idx || !a->vd || !a->vs || !a->n1 || !a->il1 || !a->cam_R || !a->cam_T || !a->g_vd || !a->g_ndc_c ||
      !a->g_joints_m || !a->g_joints_mm || !a->g_v0 || !a->g_cam_T || !a->g_disp || (a->has_normal_grad && (!a->n2 || !a->il2 || !a->g_n2)) ||
      (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->g_ndc_l || !a->g_light_R || !a->g_light_T ||
                     !a->g_light_pos)))
    return HARP_ERR_ARG;
  if (h->step.loss && (h->step.n_loss < 0 || h->step.n_loss > 64)) return HARP_ERR_ARG;
  const float* G = nullptr;
  int rc = harp_detail_chain_wide_bwd(*a, part_ws, &G, stream);
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(arm_back_wide_kernel, dim3(a->B * cb::kChainParts), dim3(kArmWide), 0, stream, *h, G, g_colors, g_betas_scratch);
  HARP_CHECK_LAUNCH();
  rc = harp_detail_tree_gA_gpm(h->tree, h->lbs_ws, a->g_v0, g_betas_scratch, a->B, stream);
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(arm_chain_bwd_kernel, dim3(a->B), dim3(64), 0, stream, *h, g_pose_scratch, g_betas_scratch);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
