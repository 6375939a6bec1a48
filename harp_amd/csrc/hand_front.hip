// Fused per-frame FRONT of a fitting step for gfx950 (MANO path): everything between the frame schedule and the rasteriser
// set-up in ONE launch, one 1024-thread workgroup per frame:
//     frame set-up (gather this frame's rows of the parameter tables: harp_frame_setup_fwd)
//  -> MANO hand layer (Rodrigues, joint chain, blend shapes, skinning, 21 joints: harp_lbs_mano_fwd)
//  -> mesh chain (subdivision, normals, displacement, normals, both projections, light camera: harp_mesh_chain_fwd)
// Why: these were 6 dependent kernel nodes of a captured step (profiles/r02_e_timeline_one_step.txt: 105 us from frame_setup to the
// end of mesh_chain_fwd, 67 us of kernel time + 38 us of node-to-node gaps), every one of them one workgroup per frame or less.  A
// frame's data fits one CU, so the stages are separated by workgroup barriers instead of graph edges: 27 us (tools/dev/
// gpu_front_stages.py).  Same arithmetic as the stand-alone kernels, which stay as the C-ABI building blocks (and serve the SMPL-X arm
// path, whose skinning is shared across frames on MFMA: lbs_tree.hip).  The rasteriser set-up was tried in here as well and taken
// out again: face records + binning of 6152 faces x 2 views are throughput work, and on ONE CU per frame they cost 130 us.
//   reference: utils/visualize.py:16-88 (prepare_mesh), manopth/manolayer.py:108-296, renderer_helper.py:454-468,
//   MeshRasterizer.transform for both views (renderer_helper.py:344, 353).
#include "chain_body.h"
#include "lbs_body.h"

namespace {

using namespace lb;
using cb::kChainThreads;

__global__ void __launch_bounds__(kChainThreads) hand_front_kernel(const harp_hand_front H) {
  extern __shared__ float s_dyn[];             // V*3 positions
  __shared__ float s_pose[48], s_beta[NB], s_tr[3], s_pm[NP], s_A[NJ * 12], s_cam[12], s_lpos[3];
  __shared__ float sR[NJ][9], sJ[NJ][3], sG[NJ][12], s_j16[NJ][3], s_tip[5][3];
  const harp_mesh_chain& A = H.chain;
  const harp_mano_model& M = H.mano;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x, tid = threadIdx.x, B = A.B;
  const int V = A.V0 + A.E0;
  float* s_p = s_dyn;
  // ---- L2 warm-up: the blend-shape rows (1.35 MB) were evicted from this XCD's L2 by the rest of the previous step, and one CU
  //      pulling them from HBM / MALL with ~45 loads per wave in flight is latency-bound (stage stamps in a replayed step: 23 us for
  //      the skinning stage, 17 us with the warm-up; 4 us when everything is L2-resident).  One load per 128-B line, all independent,
  //      issued before anything else and consumed after the joint chain: the lines arrive while the set-up and the joints run.
  float warm = 0.f;
  {
    constexpr int kLines = (NP * NV * 3 + 31) / 32, kLinesS = (NB * NV * 3 + 31) / 32;
    constexpr int kPer = (kLines + kChainThreads - 1) / kChainThreads;
    float t[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) t[i] = M.posedirs_T[(size_t)min(i * kChainThreads + tid, kLines - 1) * 32];
    const float u = M.shapedirs_T[(size_t)min(tid, kLinesS - 1) * 32];
#pragma unroll
    for (int i = 0; i < kPer; ++i) warm += t[i];
    warm += u;
  }
  // ---- optional step prologue (harp_step_frame): this workgroup's frame from the device schedule — what schedule_next_kernel did as a
  //      launch of its own in front of this one (6 us + a node gap on the critical path) —, and the clear of the frame's slice of the two
  //      gradient segments the key-point / mesh terms accumulate into (they start after this kernel)
  int f;
  if (H.step.schedule) {
    const int row = (int)((unsigned)H.step.sched_row[0] % (unsigned)H.step.n_rows);     // bumped by hand_back_kernel, a later launch
    f = H.step.schedule[(size_t)row * B + b];
    if (tid == 0) {
      const_cast<int32_t*>(H.fid)[b] = f;
      if (H.step.tfid_out) H.step.tfid_out[b] = H.step.tschedule ? H.step.tschedule[(size_t)row * B + b] : f - H.step.target_offset;
    }
  } else {
    f = H.fid[b];
  }
  if (H.step.clear_mesh_grads) {
    float* gv = const_cast<float*>(A.g_vd) + (size_t)b * V * 3;          // (inputs of the backward launch of the same struct)
    for (int i = tid; i < V * 3; i += kChainThreads) gv[i] = 0.f;
    if (tid < A.NJ * 3) const_cast<float*>(A.g_joints_m)[(size_t)b * A.NJ * 3 + tid] = 0.f;
  }
  // ---- frame set-up (glue.hip: frame_setup_fwd_kernel)
  if (tid < 48) {
    const float p = (tid < 3) ? T.rot[f * 3 + tid] : T.pose[f * 45 + tid - 3];
    s_pose[tid] = p; H.pose48[b * 48 + tid] = p;
  } else if (tid >= 64 && tid < 64 + NB) {
    const int k = tid - 64;
    const float v = T.shape[k];
    s_beta[k] = v; H.betas[b * NB + k] = v;
  } else if (tid >= 128 && tid < 131) {
    const int k = tid - 128;
    const float v = T.trans[f * 3 + k];
    s_tr[k] = v; H.trans_b[b * 3 + k] = v;
    const int lf = T.share_light ? 0 : f;
    const float lp = T.light_positions[lf * 3 + k];
    s_lpos[k] = lp; H.light_pos[b * 3 + k] = lp;
  } else if (tid == 192) {
    const float c0 = T.cam[f * 3], c1 = T.cam[f * 3 + 1], c2 = T.cam[f * 3 + 2];
    const float ct[3] = {-c1, -c2, 2.0f * A.focal / ((float)A.S * c0 + 1e-9f)};
    const float R[9] = {-1.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 1.f};
    for (int k = 0; k < 9; ++k) { s_cam[k] = R[k]; H.cam_R[b * 9 + k] = R[k]; }
    for (int k = 0; k < 3; ++k) { s_cam[9 + k] = ct[k]; H.cam_T[b * 3 + k] = ct[k]; }
  } else if (tid == 256 && b == 0) {
    if (H.self_shadow) {
      const float amb = 1.0f / (1.0f + expf(-T.amb_ratio[0]));            // nn.Sigmoid()(params['amb_ratio'])
      for (int c = 0; c < 3; ++c) { H.colors[c] = amb; H.colors[3 + c] = 1.0f - amb; H.colors[6 + c] = 0.f; }
    } else {
      for (int c = 0; c < 3; ++c) { H.colors[c] = 0.5f; H.colors[3 + c] = 0.4f; H.colors[6 + c] = 0.1f; }   // renderer_helper.py:70-73
    }
  }
  __syncthreads();
  // ---- hand layer, joints (lbs.hip: lbs_joints_kernel); workspace rows are kept for the backward pass
  const LbsWs Wl = lbs_ws(H.lbs_ws, B);
  float *w_pm = Wl.pm, *w_A = Wl.A, *w_j16 = Wl.j16, *w_Jrest = Wl.Jrest, *w_Rloc = Wl.Rloc, *w_G = Wl.G, *w_vposed = Wl.vposed;
  if (tid < NJ) {
    float aa[3];
    for (int c = 0; c < 3; ++c) {
      const float p = s_pose[3 * tid + c];
      aa[c] = (tid == 0) ? p : (M.hands_mean[3 * (tid - 1) + c] + p);       // manolayer.py:139-143
    }
    float R[9];
    rodrigues_fwd(aa, R);
    for (int k = 0; k < 9; ++k) { sR[tid][k] = R[k]; w_Rloc[(b * NJ + tid) * 9 + k] = R[k]; }
    if (tid > 0)
      for (int k = 0; k < 9; ++k) {
        const float v = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        s_pm[(tid - 1) * 9 + k] = v; w_pm[b * NP + (tid - 1) * 9 + k] = v;
      }
  } else if (tid >= 64 && tid < 64 + NJ * 3) {
    const int l = tid - 64;
    float acc = M.J_template[l];
    for (int k = 0; k < NB; ++k) acc += M.J_dirs[l * NB + k] * s_beta[k];
    sJ[l / 3][l % 3] = acc;
    w_Jrest[b * NJ * 3 + l] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 9; ++k) sG[0][(k / 3) * 4 + (k % 3)] = sR[0][k];
    for (int r = 0; r < 3; ++r) sG[0][r * 4 + 3] = sJ[0][r];
  }
  __syncthreads();
  if (tid < 5) {                          // kinematic chain: one finger per lane
    for (int lev = 0; lev < 3; ++lev) {
      const int j = 3 * tid + 1 + lev, p = parent_of(j);
      const float rel[3] = {sJ[j][0] - sJ[p][0], sJ[j][1] - sJ[p][1], sJ[j][2] - sJ[p][2]};
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          sG[j][r * 4 + c] = sG[p][r * 4] * sR[j][c] + sG[p][r * 4 + 1] * sR[j][3 + c] + sG[p][r * 4 + 2] * sR[j][6 + c];
        sG[j][r * 4 + 3] = sG[p][r * 4] * rel[0] + sG[p][r * 4 + 1] * rel[1] + sG[p][r * 4 + 2] * rel[2] + sG[p][r * 4 + 3];
      }
    }
  }
  __syncthreads();
  if (tid < NJ) {
    float* Ao = w_A + (b * NJ + tid) * 12;
    float* Go = w_G + (b * NJ + tid) * 12;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { const float g = sG[tid][r * 4 + c]; Ao[r * 4 + c] = g; Go[r * 4 + c] = g; s_A[tid * 12 + r * 4 + c] = g; }
      Go[r * 4 + 3] = sG[tid][r * 4 + 3];
      const float t3 = sG[tid][r * 4 + 3] - (sG[tid][r * 4] * sJ[tid][0] + sG[tid][r * 4 + 1] * sJ[tid][1] + sG[tid][r * 4 + 2] * sJ[tid][2]);  // :241-247
      Ao[r * 4 + 3] = t3; s_A[tid * 12 + r * 4 + 3] = t3;
      s_j16[tid][r] = sG[tid][r * 4 + 3];
      w_j16[(b * NJ + tid) * 3 + r] = sG[tid][r * 4 + 3];
    }
  }
  __syncthreads();
  if (warm == 1.2345e-30f) H.colors[9] = warm;      // keeps the warm-up loads alive (never true)
  // ---- blend shapes + skinning (lbs.hip: lbs_skin_kernel), one lane per vertex.  The 145 blend-shape rows of a vertex are 145
  //      independent 12-B loads (1.35 MB per frame through this CU's L1: ~9 us, the floor of this stage)
  if (tid < NV) {
    const int v = tid;
    const float4* wr = (const float4*)(M.weights + (size_t)v * NJ);
    const float4 w4s[4] = {wr[0], wr[1], wr[2], wr[3]};        // issued ahead of the blend-shape rows
    float q0 = M.v_template[3 * v], q1 = M.v_template[3 * v + 1], q2 = M.v_template[3 * v + 2];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const float* r = M.shapedirs_T + (size_t)k * NV * 3 + 3 * v;
      const float c = s_beta[k];
      q0 += r[0] * c; q1 += r[1] * c; q2 += r[2] * c;
    }
#pragma unroll 15
    for (int k = 0; k < NP; ++k) {
      const float* r = M.posedirs_T + (size_t)k * NV * 3 + 3 * v;
      const float c = s_pm[k];
      q0 += r[0] * c; q1 += r[1] * c; q2 += r[2] * c;
    }
    float* vpo = w_vposed + ((size_t)b * NV + v) * 3;
    vpo[0] = q0; vpo[1] = q1; vpo[2] = q2;
    float Tm[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Tm[k] = 0.f;
#pragma unroll
    for (int j4 = 0; j4 < NJ / 4; ++j4) {
      const float4 w4 = w4s[j4];
      const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 12; ++k) Tm[k] += wj[u] * s_A[(j4 * 4 + u) * 12 + k];
    }
    float o[3];
    for (int r = 0; r < 3; ++r) o[r] = (Tm[r * 4] * q0 + Tm[r * 4 + 1] * q1 + Tm[r * 4 + 2] * q2 + Tm[r * 4 + 3] + s_tr[r]) * 1000.0f;
    float* vo = (float*)A.verts_mm + ((size_t)b * NV + v) * 3;
    for (int r = 0; r < 3; ++r) { vo[r] = o[r]; s_p[3 * v + r] = o[r] * 1e-3f; }
    for (int k = 0; k < 5; ++k)
      if (v == c_tips[k]) { s_tip[k][0] = o[0]; s_tip[k][1] = o[1]; s_tip[k][2] = o[2]; }
  }
  __syncthreads();
  if (tid < 63) {                          // 21 joints, millimetres (lbs_joints_out_kernel) and metres (visualize.py:46)
    const int k = tid / 3, c = tid % 3, src = c_reorder[k];
    const float jm = (src < NJ) ? (s_j16[src][c] + s_tr[c]) * 1000.0f : s_tip[src - NJ][c];
    ((float*)A.joints_mm)[(size_t)b * 63 + tid] = jm;
    A.joints_m[(size_t)b * 63 + tid] = jm * 1e-3f;
  }
  // ---- mesh chain (chain_body.h); its first statement after the (skipped) load is a barrier
  cb::mesh_chain_fwd_body(A, s_p, b, true, s_cam, s_cam + 9, s_lpos);
}

// ---- wide form: the hand layer of a frame on kChainParts workgroups (a contiguous quarter of the 778 vertices each: the 1.35 MB of
//      blend-shape rows of a frame go through FOUR CUs' L1 instead of one), followed by the wide mesh chain (chain_wide.hip).  Every
//      workgroup gathers the frame's rows and runs the 16-joint chain (same arithmetic, a few hundred flops); part 0 writes the shared rows.
constexpr int kWideThreads = 256;
__global__ void __launch_bounds__(kWideThreads) hand_front_wide_kernel(const harp_hand_front H, const int clear_here) {
  __shared__ float s_pose[48], s_beta[NB], s_tr[3], s_pm[NP], s_A[NJ * 12];
  __shared__ float sR[NJ][9], sJ[NJ][3], sG[NJ][12], s_j16[NJ][3];
  const harp_mesh_chain& A = H.chain;
  const harp_mano_model& M = H.mano;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x / cb::kChainParts, part = blockIdx.x % cb::kChainParts, tid = threadIdx.x, B = A.B;
  const bool lead = part == 0;
  constexpr int kPer = (NV + cb::kChainParts - 1) / cb::kChainParts;
  static_assert(kPer <= kWideThreads, "one vertex per thread");
  // ---- L2 / L1 warm-up of this part's columns of the 145 blend-shape rows (one load per 128-B line, all independent, consumed after the
  //      joint chain): the rows were evicted by the rest of the previous step, see hand_front_kernel
  float warm = 0.f;
  {
    constexpr int kLinesRow = (kPer * 3 + 31) / 32 + 1, kLines = (NP + NB) * kLinesRow, kEach = (kLines + kWideThreads - 1) / kWideThreads;
    const int col0 = part * kPer * 3;
    float t[kEach];
#pragma unroll
    for (int q = 0; q < kEach; ++q) {
      const int idx = min(q * kWideThreads + tid, kLines - 1), row = idx / kLinesRow, col = min(col0 + (idx % kLinesRow) * 32, NV * 3 - 1);
      t[q] = (row < NP) ? M.posedirs_T[(size_t)row * NV * 3 + col] : M.shapedirs_T[(size_t)(row - NP) * NV * 3 + col];
    }
#pragma unroll
    for (int q = 0; q < kEach; ++q) warm += t[q];
  }
  int f;
  if (H.step.schedule) {
    const int row = (int)((unsigned)H.step.sched_row[0] % (unsigned)H.step.n_rows);     // bumped by hand_back_kernel, a later launch
    f = H.step.schedule[(size_t)row * B + b];
    if (tid == 0 && lead) {
      const_cast<int32_t*>(H.fid)[b] = f;
      if (H.step.tfid_out) H.step.tfid_out[b] = H.step.tschedule ? H.step.tschedule[(size_t)row * B + b] : f - H.step.target_offset;
    }
  } else {
    f = H.fid[b];
  }
  // ---- frame set-up (glue.hip: frame_setup_fwd_kernel)
  if (tid < 48) {
    const float p = (tid < 3) ? T.rot[f * 3 + tid] : T.pose[f * 45 + tid - 3];
    s_pose[tid] = p;
    if (lead) H.pose48[b * 48 + tid] = p;
  } else if (tid >= 64 && tid < 64 + NB) {
    const int k = tid - 64;
    const float v = T.shape[k];
    s_beta[k] = v;
    if (lead) H.betas[b * NB + k] = v;
  } else if (tid >= 128 && tid < 131) {
    const int k = tid - 128;
    const float v = T.trans[f * 3 + k];
    s_tr[k] = v;
    if (lead) {
      H.trans_b[b * 3 + k] = v;
      const int lf = T.share_light ? 0 : f;
      H.light_pos[b * 3 + k] = T.light_positions[lf * 3 + k];
    }
  } else if (tid == 192 && lead) {
    const float c0 = T.cam[f * 3], c1 = T.cam[f * 3 + 1], c2 = T.cam[f * 3 + 2];
    const float ct[3] = {-c1, -c2, 2.0f * A.focal / ((float)A.S * c0 + 1e-9f)};
    const float R[9] = {-1.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 1.f};
    for (int k = 0; k < 9; ++k) H.cam_R[b * 9 + k] = R[k];
    for (int k = 0; k < 3; ++k) H.cam_T[b * 3 + k] = ct[k];
  } else if (tid == 193 && lead && b == 0) {
    if (H.self_shadow) {
      const float amb = 1.0f / (1.0f + expf(-T.amb_ratio[0]));            // nn.Sigmoid()(params['amb_ratio'])
      for (int c = 0; c < 3; ++c) { H.colors[c] = amb; H.colors[3 + c] = 1.0f - amb; H.colors[6 + c] = 0.f; }
    } else {
      for (int c = 0; c < 3; ++c) { H.colors[c] = 0.5f; H.colors[3 + c] = 0.4f; H.colors[6 + c] = 0.1f; }   // renderer_helper.py:70-73
    }
  }
  __syncthreads();
  // ---- hand layer, joints (lbs.hip: lbs_joints_kernel); workspace rows are kept for the backward pass (written by part 0)
  const LbsWs Wl = lbs_ws(H.lbs_ws, B);
  if (tid < NJ) {
    float aa[3];
    for (int c = 0; c < 3; ++c) {
      const float p = s_pose[3 * tid + c];
      aa[c] = (tid == 0) ? p : (M.hands_mean[3 * (tid - 1) + c] + p);       // manolayer.py:139-143
    }
    float R[9];
    rodrigues_fwd(aa, R);
    for (int k = 0; k < 9; ++k) { sR[tid][k] = R[k]; if (lead) Wl.Rloc[(b * NJ + tid) * 9 + k] = R[k]; }
    if (tid > 0)
      for (int k = 0; k < 9; ++k) {
        const float v = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        s_pm[(tid - 1) * 9 + k] = v;
        if (lead) Wl.pm[b * NP + (tid - 1) * 9 + k] = v;
      }
  } else if (tid >= 64 && tid < 64 + NJ * 3) {
    const int l = tid - 64;
    float acc = M.J_template[l];
    for (int k = 0; k < NB; ++k) acc += M.J_dirs[l * NB + k] * s_beta[k];
    sJ[l / 3][l % 3] = acc;
    if (lead) Wl.Jrest[b * NJ * 3 + l] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 9; ++k) sG[0][(k / 3) * 4 + (k % 3)] = sR[0][k];
    for (int r = 0; r < 3; ++r) sG[0][r * 4 + 3] = sJ[0][r];
  }
  __syncthreads();
  if (tid < 5) {                          // kinematic chain: one finger per lane
    for (int lev = 0; lev < 3; ++lev) {
      const int j = 3 * tid + 1 + lev, p = parent_of(j);
      const float rel[3] = {sJ[j][0] - sJ[p][0], sJ[j][1] - sJ[p][1], sJ[j][2] - sJ[p][2]};
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          sG[j][r * 4 + c] = sG[p][r * 4] * sR[j][c] + sG[p][r * 4 + 1] * sR[j][3 + c] + sG[p][r * 4 + 2] * sR[j][6 + c];
        sG[j][r * 4 + 3] = sG[p][r * 4] * rel[0] + sG[p][r * 4 + 1] * rel[1] + sG[p][r * 4 + 2] * rel[2] + sG[p][r * 4 + 3];
      }
    }
  }
  __syncthreads();
  if (tid < NJ) {
    float* Ao = Wl.A + (b * NJ + tid) * 12;
    float* Go = Wl.G + (b * NJ + tid) * 12;
    for (int r = 0; r < 3; ++r) {
      const float t3 = sG[tid][r * 4 + 3] - (sG[tid][r * 4] * sJ[tid][0] + sG[tid][r * 4 + 1] * sJ[tid][1] + sG[tid][r * 4 + 2] * sJ[tid][2]);  // :241-247
      for (int c = 0; c < 3; ++c) { const float g = sG[tid][r * 4 + c]; s_A[tid * 12 + r * 4 + c] = g; if (lead) { Ao[r * 4 + c] = g; Go[r * 4 + c] = g; } }
      s_A[tid * 12 + r * 4 + 3] = t3;
      s_j16[tid][r] = sG[tid][r * 4 + 3];
      if (lead) { Go[r * 4 + 3] = sG[tid][r * 4 + 3]; Ao[r * 4 + 3] = t3; Wl.j16[(b * NJ + tid) * 3 + r] = sG[tid][r * 4 + 3]; }
    }
  }
  __syncthreads();
  if (warm == 1.2345e-30f) H.colors[9] = warm;      // keeps the warm-up loads alive (never true)
  if (clear_here && H.step.clear_mesh_grads) {       // (hybrid front: the one-workgroup chain that follows does not clear)
    const int V = A.V0 + A.E0, per = (V * 3 + cb::kChainParts - 1) / cb::kChainParts;
    float* gv = const_cast<float*>(A.g_vd) + (size_t)b * V * 3;
    for (int k = part * per + tid; k < min((part + 1) * per, V * 3); k += kWideThreads) gv[k] = 0.f;
    if (lead && tid < A.NJ * 3) const_cast<float*>(A.g_joints_m)[(size_t)b * A.NJ * 3 + tid] = 0.f;
  }
  // ---- blend shapes + skinning (lbs.hip: lbs_skin_kernel), one lane per vertex of this part's quarter
  const int v = part * kPer + tid;
  if (tid < kPer && v < NV) {
    const float4* wr = (const float4*)(M.weights + (size_t)v * NJ);
    const float4 w4s[4] = {wr[0], wr[1], wr[2], wr[3]};        // issued ahead of the blend-shape rows
    float q0 = M.v_template[3 * v], q1 = M.v_template[3 * v + 1], q2 = M.v_template[3 * v + 2];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const float* r = M.shapedirs_T + (size_t)k * NV * 3 + 3 * v;
      const float c = s_beta[k];
      q0 += r[0] * c; q1 += r[1] * c; q2 += r[2] * c;
    }
#pragma unroll 15
    for (int k = 0; k < NP; ++k) {
      const float* r = M.posedirs_T + (size_t)k * NV * 3 + 3 * v;
      const float c = s_pm[k];
      q0 += r[0] * c; q1 += r[1] * c; q2 += r[2] * c;
    }
    float* vpo = Wl.vposed + ((size_t)b * NV + v) * 3;
    vpo[0] = q0; vpo[1] = q1; vpo[2] = q2;
    float Tm[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Tm[k] = 0.f;
#pragma unroll
    for (int j4 = 0; j4 < NJ / 4; ++j4) {
      const float4 w4 = w4s[j4];
      const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 12; ++k) Tm[k] += wj[u] * s_A[(j4 * 4 + u) * 12 + k];
    }
    float o[3];
    for (int r = 0; r < 3; ++r) o[r] = (Tm[r * 4] * q0 + Tm[r * 4 + 1] * q1 + Tm[r * 4 + 2] * q2 + Tm[r * 4 + 3] + s_tr[r]) * 1000.0f;
    float* vo = (float*)A.verts_mm + ((size_t)b * NV + v) * 3;
    for (int r = 0; r < 3; ++r) vo[r] = o[r];
    for (int k = 0; k < 5; ++k)
      if (v == c_tips[k])                  // finger-tip joints (c_reorder: tip k is output joint 4 (k + 1)), millimetres; metres by the chain
        for (int r = 0; r < 3; ++r) ((float*)A.joints_mm)[(size_t)b * 63 + 12 * (k + 1) + r] = o[r];
  }
  if (lead && tid >= 192 && tid < 192 + 63) {                  // chain joints (lbs_joints_out_kernel)
    const int t = tid - 192, k = t / 3, c = t % 3, src = c_reorder[k];
    if (src < NJ) ((float*)A.joints_mm)[(size_t)b * 63 + t] = (s_j16[src][c] + s_tr[c]) * 1000.0f;
  }
}

}  // namespace

int harp_detail_chain_wide_tail(const harp_mesh_chain& a, int clear_grads, float* part_ws, hipStream_t stream);

extern "C" {

int harp_hand_front_fwd(const harp_hand_front* h, hipStream_t stream) {
  if (!h) return HARP_ERR_ARG;
  const harp_mesh_chain& a = h->chain;
  if (!a.edges0 || !a.vf_off || !a.vf_tri || !a.disp || a.B <= 0 || a.V0 != NV || a.E0 < 0 || a.NJ != 21 ||
      a.V0 + a.E0 > harp_mesh_chain_max_vertices() || !a.verts_mm || !a.joints_mm || !a.joints_m || !a.vs || !a.n1 || !a.il1 || !a.vd ||
      !a.n2 || !a.il2 || !a.ndc_c || (a.shadow && (!a.centroid || !a.light_R || !a.light_T || !a.ndc_l)))
    return HARP_ERR_ARG;
  if (!h->fid || !h->pose48 || !h->betas || !h->trans_b || !h->cam_R || !h->cam_T || !h->light_pos || !h->colors || !h->lbs_ws ||
      h->tables.wrist_pose)
    return HARP_ERR_ARG;
  if ((h->step.schedule && (!h->step.sched_row || h->step.n_rows <= 0)) || (h->step.clear_mesh_grads && (!a.g_vd || !a.g_joints_m)))
    return HARP_ERR_ARG;
  // The kernel needs V*12 B (<= 48 KB) of dynamic LDS and asks for (almost) the whole CU's 160 KB: a workgroup of it is a frame's latency
  // chain, and while it runs the parameter-only regularisers start on the second stream — a kernel of 1 500 small workgroups with a few
  // bytes of LDS each, which can then not land on this CU and take issue slots / L1 from the chain (-3 us / step, same-box A/B x3;
  // forcing 128 VGPRs — no co-resident wave at all — was bimodal: DESIGN.md 6.4).  HARP_FRONT_LDS=<bytes> overrides (0: only what is needed).
  const size_t need = (size_t)(a.V0 + a.E0) * 3 * sizeof(float);
  static size_t fence = 0;                                              // 160 KB - the kernel's static LDS: nothing is left on the CU
  const size_t lds = harp_lds_fence((const void*)hand_front_kernel, "HARP_FRONT_LDS", 159744, need, &fence);
  hipLaunchKernelGGL(hand_front_kernel, dim3(a.B), dim3(kChainThreads), lds, stream, *h);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// The same front on kChainParts workgroups per frame: hand layer (hand_front_wide_kernel) + the wide mesh chain (chain_wide.hip), three
// launches.  part_ws: harp_mesh_chain_wide_ws_floats(B) floats.
int harp_hand_front_wide_fwd(const harp_hand_front* h, float* part_ws, hipStream_t stream) {
  if (!h || !part_ws) return HARP_ERR_ARG;
  const harp_mesh_chain& a = h->chain;
  if (!a.edges0 || !a.vf_off || !a.vf_tri || !a.disp || a.B <= 0 || a.V0 != NV || a.E0 < 0 || a.NJ != 21 ||
      (a.V0 + a.E0 + cb::kChainParts - 1) / cb::kChainParts > kChainThreads || (a.V0 + a.E0) * 12 > 64 * 1024 || !a.verts_mm || !a.joints_mm ||
      !a.joints_m || !a.vs || !a.n1 || !a.il1 || !a.vd || !a.n2 || !a.il2 || !a.ndc_c ||
      (a.shadow && (!a.centroid || !a.light_R || !a.light_T || !a.ndc_l)))
    return HARP_ERR_ARG;
  if (!h->fid || !h->pose48 || !h->betas || !h->trans_b || !h->cam_R || !h->cam_T || !h->light_pos || !h->colors || !h->lbs_ws ||
      h->tables.wrist_pose)
    return HARP_ERR_ARG;
  if ((h->step.schedule && (!h->step.sched_row || h->step.n_rows <= 0)) || (h->step.clear_mesh_grads && (!a.g_vd || !a.g_joints_m)))
    return HARP_ERR_ARG;
  hipLaunchKernelGGL(hand_front_wide_kernel, dim3(a.B * cb::kChainParts), dim3(kWideThreads), 0, stream, *h, 0);
  HARP_CHECK_LAUNCH();
  return harp_detail_chain_wide_tail(a, h->step.clear_mesh_grads, part_ws, stream);
}

// Hybrid: the hand layer on four workgroups per frame (above), then the mesh chain as ONE workgroup per frame (harp_mesh_chain_fwd): two
// launches.  Same outputs.
int harp_hand_front_hybrid_fwd(const harp_hand_front* h, hipStream_t stream) {
  if (!h) return HARP_ERR_ARG;
  const harp_mesh_chain& a = h->chain;
  if (a.B <= 0 || a.V0 != NV || a.E0 < 0 || a.NJ != 21 || !a.verts_mm || !a.joints_mm || !h->fid || !h->pose48 || !h->betas || !h->trans_b ||
      !h->cam_R || !h->cam_T || !h->light_pos || !h->colors || !h->lbs_ws || h->tables.wrist_pose ||
      (h->step.schedule && (!h->step.sched_row || h->step.n_rows <= 0)) || (h->step.clear_mesh_grads && (!a.g_vd || !a.g_joints_m)))
    return HARP_ERR_ARG;
  hipLaunchKernelGGL(hand_front_wide_kernel, dim3(a.B * cb::kChainParts), dim3(kWideThreads), 0, stream, *h, 1);
  HARP_CHECK_LAUNCH();
  return harp_mesh_chain_fwd(&a, stream);
}

}  // extern "C"
