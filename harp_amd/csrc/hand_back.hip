// Fused per-frame BACK of a fitting step for gfx950 (MANO path), the counterpart of hand_front.hip.  The backward tail of a step was six
// dependent kernel nodes — mesh-chain backward, lbs_joints_bwd, lbs_skin<true>, lbs_gA_gpm, lbs_chain_bwd, frame_setup_bwd: 52 + 6 + 6
// + 16 + 13 + 6 us in profiles/r03_c_timeline_one_step.txt, every launch but the first paying its cold dependent loads — and is three:
//   1. hand_back_kernel (here), one 1024-thread workgroup per frame: the mesh-chain backward, then — on the frame's g_v0 while it is still
//      in this CU's cache — the joint split and the per-vertex skinning backward of the hand layer (g_vp, the outer products M), and the
//      scatter of everything that is final by then (trans, cam, light position, ambient ratio) into the parameter tables' gradient rows;
//   2. lbs_gA_gpm (lbs.hip), unchanged: the two reductions over the vertices stay spread over 72 workgroups per frame.  (Round 3 built
//      them into this kernel too: 1.26 MB of blend-shape rows per frame through ONE CU took longer than the launch they saved —
//      0.790 vs 0.782 ms / step — the same finding as for the rasteriser set-up inside hand_front.)
//   3. lbs_chain_bwd<SCATTER> (lbs.hip): kinematic chain + Rodrigues backward, adding pose / rot / shape straight to their rows.
// Same arithmetic as the stand-alone kernels (lbs.hip, glue.hip), which stay as the C-ABI building blocks and serve the API path.
//   reference: autograd of manopth/manolayer.py:108-296 and utils/visualize.py:45-64; the row gathers params[...][fid] (:26-27).
#include "chain_body.h"
#include "lbs_body.h"

int harp_detail_lbs_back_tail(const harp_mano_model& m, const float* pose, int B, float* ws, float* g_betas, const harp_frame_tables& tables,
                              const int32_t* fid, hipStream_t stream);

namespace {

using namespace lb;
using cb::kChainThreads;

__global__ void __launch_bounds__(kChainThreads) hand_back_kernel(const harp_hand_front H, const float* __restrict__ g_colors,
                                                                  float* __restrict__ g_betas) {
  extern __shared__ float s_mem[];             // the chain backward's three V*3 buffers
  __shared__ float s_A[NJ * 12], s_gj16[NJ][3], s_gtip[5][3], s_red[16 * 3], s_tot[3];
  const harp_mesh_chain& A = H.chain;
  const harp_mano_model& M = H.mano;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x, tid = threadIdx.x, B = A.B;
  const LbsWs Wl = lbs_ws(H.lbs_ws, B);
  const bool lean = A.light_only != 0;         // appearance-only stage: light-view part, light / ambient scatter and the step epilogue only
  if (!lean) {
    // the reduction buffers of the next launch (g_A | g_pm adjacent in the workspace, g_betas) are cleared by all workgroups together
    for (int k = b * kChainThreads + tid; k < B * (192 + 135); k += gridDim.x * kChainThreads) Wl.g_A[k] = 0.f;
    for (int k = b * kChainThreads + tid; k < B * NB; k += gridDim.x * kChainThreads) g_betas[k] = 0.f;
  }

  cb::mesh_chain_bwd_body(A, s_mem, b);        // ... -> g_v0 (this frame's 778 x 3), g_cam_T, g_light_pos, g_disp
  __threadfence_block();
  __syncthreads();
  if (!lean) {

  // ---- joint gradients (lbs_joints_bwd): chain joints -> g_j16 [metres], finger tips -> their vertices; stage this frame's A
  if (tid < 63) {
    const int k = tid / 3, c = tid % 3, src = c_reorder[k];
    const float gj = A.g_joints_m[(size_t)b * 63 + tid] * 1e-3f;            // (= g_joints_mm)
    if (src < NJ) { s_gj16[src][c] = gj * 1000.0f; Wl.g_j16[((size_t)b * NJ + src) * 3 + c] = gj * 1000.0f; }
    else s_gtip[src - NJ][c] = gj;
  } else if (tid >= 64 && tid < 64 + NJ * 12) {
    s_A[tid - 64] = Wl.A[(size_t)b * NJ * 12 + tid - 64];
  }
  __syncthreads();
  // ---- skinning backward (lbs_skin<true>), one lane per vertex: g_vp = T^T g, M = [g (x) v_posed | g]; g_trans partial sums
  float gt3[3] = {0.f, 0.f, 0.f};
  if (tid < NV) {
    const int v = tid;
    const float4* wr = (const float4*)(M.weights + (size_t)v * NJ);
    const float4 w4s[4] = {wr[0], wr[1], wr[2], wr[3]};
    const float* gv = A.g_v0 + ((size_t)b * NV + v) * 3;
    float g[3] = {gv[0], gv[1], gv[2]};
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (v == c_tips[k]) { g[0] += s_gtip[k][0]; g[1] += s_gtip[k][1]; g[2] += s_gtip[k][2]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { g[c] *= 1000.0f; gt3[c] = g[c]; }
    const float* qp = Wl.vposed + ((size_t)b * NV + v) * 3;
    const float q[3] = {qp[0], qp[1], qp[2]};
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
    float* gvp = Wl.g_vp + ((size_t)b * NV + v) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) gvp[c] = Tm[c] * g[0] + Tm[4 + c] * g[1] + Tm[8 + c] * g[2];
    float* mo = Wl.Mo + ((size_t)b * NV + v) * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      mo[r * 4] = g[r] * q[0]; mo[r * 4 + 1] = g[r] * q[1]; mo[r * 4 + 2] = g[r] * q[2]; mo[r * 4 + 3] = g[r];
    }
  } else if (tid >= 800 && tid < 800 + NJ) {
    for (int c = 0; c < 3; ++c) gt3[c] = s_gj16[tid - 800][c];               // g_trans also collects the chain joints
  }
  cb::block_sum_n<3>(gt3, s_red, s_tot);
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

// ---- wide form (csrc/chain_wide.hip): the mesh-chain backward has run on four workgroups per frame and left dL/d(subdivided vertices) in
//      its scratch (G); here, also on four workgroups per frame with a quarter of the 778 base vertices each: SubdivideMeshes backward +
//      millimetres, the joint split, the per-vertex skinning backward, the partial translation sums; part 0 scatters cam / light / ambient
//      and turns the step's book-keeping over.
constexpr int kWideThreads = 256;
__global__ void __launch_bounds__(kWideThreads) hand_back_wide_kernel(const harp_hand_front H, const float* __restrict__ G,
                                                                      const float* __restrict__ g_colors, float* __restrict__ g_betas) {
  __shared__ float s_A[NJ * 12], s_gj16[NJ][3], s_gtip[5][3], s_red[4][3];
  const harp_mesh_chain& A = H.chain;
  const harp_mano_model& M = H.mano;
  const harp_frame_tables& T = H.tables;
  const int b = blockIdx.x / cb::kChainParts, part = blockIdx.x % cb::kChainParts, tid = threadIdx.x, B = A.B;
  const bool lead = part == 0;
  constexpr int kPer = (NV + cb::kChainParts - 1) / cb::kChainParts;
  const LbsWs Wl = lbs_ws(H.lbs_ws, B);
  const int V = A.V0 + A.E0;
  // the reduction buffers of the next launch (g_A | g_pm adjacent in the workspace, g_betas) are cleared by all workgroups together
  for (int k = blockIdx.x * kWideThreads + tid; k < B * (192 + 135); k += gridDim.x * kWideThreads) Wl.g_A[k] = 0.f;
  for (int k = blockIdx.x * kWideThreads + tid; k < B * NB; k += gridDim.x * kWideThreads) g_betas[k] = 0.f;
  // ---- joint gradients (lbs_joints_bwd): chain joints -> g_j16 [metres], finger tips -> their vertices; stage this frame's A
  if (tid < 63) {
    const int k = tid / 3, c = tid % 3, src = c_reorder[k];
    const float gj = A.g_joints_m[(size_t)b * 63 + tid] * 1e-3f;            // (= g_joints_mm)
    if (src < NJ) { s_gj16[src][c] = gj * 1000.0f; if (lead) Wl.g_j16[((size_t)b * NJ + src) * 3 + c] = gj * 1000.0f; }
    else s_gtip[src - NJ][c] = gj;
  } else if (tid >= 64 && tid < 64 + NJ * 12) {
    s_A[tid - 64] = Wl.A[(size_t)b * NJ * 12 + tid - 64];
  }
  __syncthreads();
  // ---- SubdivideMeshes backward, then the skinning backward (lbs_skin<true>), one lane per vertex of this part's quarter
  float gt3[3] = {0.f, 0.f, 0.f};
  const int v = part * kPer + tid;
  if (tid < kPer && v < NV) {
    const float4* wr = (const float4*)(M.weights + (size_t)v * NJ);
    const float4 w4s[4] = {wr[0], wr[1], wr[2], wr[3]};
    const float* qp = Wl.vposed + ((size_t)b * NV + v) * 3;
    const float q[3] = {qp[0], qp[1], qp[2]};
    const cb::V3 g0 = cb::subdivide_bwd_vertex(G + (size_t)b * V * 3, A.sub_off, A.sub_idx, v);
    cb::st(A.g_v0 + ((size_t)b * NV + v) * 3, g0);
    float g[3] = {g0.x, g0.y, g0.z};
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (v == c_tips[k]) { g[0] += s_gtip[k][0]; g[1] += s_gtip[k][1]; g[2] += s_gtip[k][2]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { g[c] *= 1000.0f; gt3[c] = g[c]; }
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
    float* gvp = Wl.g_vp + ((size_t)b * NV + v) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) gvp[c] = Tm[c] * g[0] + Tm[4 + c] * g[1] + Tm[8 + c] * g[2];
    float* mo = Wl.Mo + ((size_t)b * NV + v) * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      mo[r * 4] = g[r] * q[0]; mo[r * 4 + 1] = g[r] * q[1]; mo[r * 4 + 2] = g[r] * q[2]; mo[r * 4 + 3] = g[r];
    }
  } else if (lead && tid >= 224 && tid < 224 + NJ) {
    for (int c = 0; c < 3; ++c) gt3[c] = s_gj16[tid - 224][c];               // g_trans also collects the chain joints
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sw = wave_sum_u(gt3[c]);
    if ((tid & 63) == 0) s_red[tid >> 6][c] = sw;
  }
  __syncthreads();
  // ---- scatter into the gradient rows of the parameter tables (frame_setup_bwd_kernel's trans / cam / light part): atomics
  const int f = H.fid[b];
  if (tid < 3) {
    const int k = tid;
    if (T.g_trans) atomicAdd(T.g_trans + f * 3 + k, s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k]);
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
  } else if (tid == 64 && lead && b == 0 && H.self_shadow && g_colors && T.g_amb_ratio) {
    const float amb = 1.0f / (1.0f + expf(-T.amb_ratio[0]));
    const float g_amb = (g_colors[0] + g_colors[1] + g_colors[2]) - (g_colors[3] + g_colors[4] + g_colors[5]);
    atomicAdd(T.g_amb_ratio, g_amb * amb * (1.0f - amb));
  }
  if (b == 0 && lead) {                             // step epilogue (harp_step_frame), as in hand_back_kernel
    const harp_step_frame& E = H.step;
    if ((tid >> 6) == 2) {
      const int k = tid - 128;
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
    } else if (tid == 192 && E.schedule) {
      E.sched_row[0] = (int)((unsigned)E.sched_row[0] % (unsigned)E.n_rows) + 1;
    } else if (tid == 193 && E.draw_counter) {
      E.draw_counter[0] += 1;
    }
  }
}

}  // namespace

int harp_detail_chain_wide_bwd(const harp_mesh_chain& a, float* part_ws, const float** G_out, hipStream_t stream);

extern "C" {

int harp_hand_back_bwd(const harp_hand_front* h, const float* g_colors, float* g_betas_scratch, hipStream_t stream) {
  if (!h || !g_betas_scratch) return HARP_ERR_ARG;
  const harp_mesh_chain* a = &h->chain;
  if (!a->edges0 || !a->vf_off || !a->vf_tri || !a->disp || a->B <= 0 || a->V0 != NV || a->E0 < 0 || a->NJ != 21 ||
      a->V0 + a->E0 > harp_mesh_chain_max_vertices() || !a->sub_off || !a->sub_idx || !a->vd || !a->vs || !a->n1 || !a->il1 || !a->cam_R ||
      !a->cam_T || !a->g_vd || !a->g_ndc_c || !a->g_joints_m || !a->g_joints_mm || !a->g_v0 || !a->g_cam_T || !a->g_disp ||
      (a->has_normal_grad && (!a->n2 || !a->il2 || !a->g_n2)) ||
      (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->g_ndc_l || !a->g_light_R || !a->g_light_T ||
                     !a->g_light_pos)))
    return HARP_ERR_ARG;
  if (!h->fid || !h->pose48 || !h->lbs_ws || h->tables.wrist_pose) return HARP_ERR_ARG;
  if ((h->step.schedule && (!h->step.sched_row || h->step.n_rows <= 0)) || (h->step.loss && (h->step.n_loss < 0 || h->step.n_loss > 64)))
    return HARP_ERR_ARG;
  const size_t lds = (size_t)(a->V0 + a->E0) * 9 * sizeof(float);
  // dynamic LDS above 64 KB has to be requested; the attribute is per DEVICE, so it is set on every call (cheap) rather than cached in a
  // process-wide static that a second device or a concurrent first call would defeat
  if (hipFuncSetAttribute((const void*)hand_back_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return HARP_ERR_ARG;
  hipLaunchKernelGGL(hand_back_kernel, dim3(a->B), dim3(kChainThreads), lds, stream, *h, g_colors, g_betas_scratch);
  HARP_CHECK_LAUNCH();
  if (a->light_only) return HARP_OK;           // no hand-layer backward: nothing of it reaches the appearance optimiser's parameters
  return harp_detail_lbs_back_tail(h->mano, h->pose48, a->B, h->lbs_ws, g_betas_scratch, h->tables, h->fid, stream);
}

// The same tail with the mesh-chain backward on four workgroups per frame (csrc/chain_wide.hip: three launches) and the hand-layer part on
// four as well (hand_back_wide_kernel), then the two launches of harp_detail_lbs_back_tail.  part_ws: harp_mesh_chain_wide_ws_floats(B, V).
// chain.light_only (appearance-only stage: one small pass) goes through harp_hand_back_bwd.
int harp_hand_back_wide_bwd(const harp_hand_front* h, const float* g_colors, float* g_betas_scratch, float* part_ws, hipStream_t stream) {
  if (!h || !part_ws) return HARP_ERR_ARG;
  if (h->chain.light_only) return harp_hand_back_bwd(h, g_colors, g_betas_scratch, stream);
  if (!g_betas_scratch) return HARP_ERR_ARG;
  const harp_mesh_chain* a = &h->chain;
  if (!a->vf_off || !a->vf_tri || !a->disp || a->B <= 0 || a->V0 != NV || a->E0 < 0 || a->NJ != 21 ||
      (a->V0 + a->E0 + cb::kChainParts - 1) / cb::kChainParts > kChainThreads || (a->V0 + a->E0) * 24 > 160 * 1024 - 256 || !a->sub_off ||
      !a->sub_idx || !a->vd || !a->vs || !a->n1 || !a->il1 || !a->cam_R || !a->cam_T || !a->g_vd || !a->g_ndc_c || !a->g_joints_m ||
      !a->g_joints_mm || !a->g_v0 || !a->g_cam_T || !a->g_disp || (a->has_normal_grad && (!a->n2 || !a->il2 || !a->g_n2)) ||
      (a->shadow && (!a->light_pos || !a->centroid || !a->light_R || !a->light_T || !a->g_ndc_l || !a->g_light_R || !a->g_light_T ||
                     !a->g_light_pos)))
    return HARP_ERR_ARG;
  if (!h->fid || !h->pose48 || !h->lbs_ws || h->tables.wrist_pose) return HARP_ERR_ARG;
  if ((h->step.schedule && (!h->step.sched_row || h->step.n_rows <= 0)) || (h->step.loss && (h->step.n_loss < 0 || h->step.n_loss > 64)))
    return HARP_ERR_ARG;
  const float* G = nullptr;
  const int rc = harp_detail_chain_wide_bwd(*a, part_ws, &G, stream);
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(hand_back_wide_kernel, dim3(a->B * cb::kChainParts), dim3(kWideThreads), 0, stream, *h, G, g_colors, g_betas_scratch);
  HARP_CHECK_LAUNCH();
  return harp_detail_lbs_back_tail(h->mano, h->pose48, a->B, h->lbs_ws, g_betas_scratch, h->tables, h->fid, stream);
}

}  // extern "C"
