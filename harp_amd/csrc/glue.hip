// Per-frame "glue" of the fitting loop for gfx950: everything the reference does with dozens of tiny torch ops between
// the parameter dict and the renderers — row gathers params[k][fid] (utils/visualize.py:26-27,38-39), the camera
// convention (utils/visualize.py:268-271), shared light + ambient ratio (optimize_sequence.py:453-456, 478-480;
// renderer_helper.py:435-441) and the light camera of process_info_for_shadow (renderer_helper.py:454-468, with
// PyTorch3D look_at_rotation) — as one forward and one backward kernel each, one lane per frame.  The backward scatters
// straight into the flat gradient arena of the parameter tables (dense-Adam semantics: untouched rows keep zero grad).
#include "chain_body.h"      // V3 helpers + the light camera (process_info_for_shadow / look_at_rotation), shared with the fused chain

namespace {

using namespace cb;

__global__ void frame_setup_fwd_kernel(const harp_frame_tables t, const int32_t* __restrict__ fid, int B, int S, float focal,
                                       int self_shadow, float* __restrict__ pose48, float* __restrict__ betas,
                                       float* __restrict__ trans_b, float* __restrict__ cam_R, float* __restrict__ cam_T,
                                       float* __restrict__ light_pos, float* __restrict__ colors) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) {
    if (self_shadow) {
      const float amb = 1.0f / (1.0f + expf(-t.amb_ratio[0]));            // nn.Sigmoid()(params['amb_ratio'])
      for (int c = 0; c < 3; ++c) { colors[c] = amb; colors[3 + c] = 1.0f - amb; colors[6 + c] = 0.f; }
    } else {
      for (int c = 0; c < 3; ++c) { colors[c] = 0.5f; colors[3 + c] = 0.4f; colors[6 + c] = 0.1f; }   // renderer_helper.py:70-73
    }
  }
  if (b >= B) return;
  const int f = fid[b];
  const int ps = t.wrist_pose ? 51 : 48, ho = t.wrist_pose ? 6 : 3, nbo = t.n_betas_out > 0 ? t.n_betas_out : 10;
  for (int k = 0; k < 3; ++k) pose48[b * ps + k] = t.rot[f * 3 + k];
  if (t.wrist_pose) for (int k = 0; k < 3; ++k) pose48[b * ps + 3 + k] = t.wrist_pose[f * 3 + k];
  for (int k = 0; k < 45; ++k) pose48[b * ps + ho + k] = t.pose[f * 45 + k];
  for (int k = 0; k < nbo; ++k) betas[b * nbo + k] = (k < 10) ? t.shape[k] : 0.f;
  for (int k = 0; k < 3; ++k) trans_b[b * 3 + k] = t.trans[f * 3 + k];
  const float c0 = t.cam[f * 3], c1 = t.cam[f * 3 + 1], c2 = t.cam[f * 3 + 2];
  cam_T[b * 3] = -c1; cam_T[b * 3 + 1] = -c2; cam_T[b * 3 + 2] = 2.0f * focal / ((float)S * c0 + 1e-9f);
  const float R[9] = {-1.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, 1.f};
  for (int k = 0; k < 9; ++k) cam_R[b * 9 + k] = R[k];
  const int lf = t.share_light ? 0 : f;
  for (int k = 0; k < 3; ++k) light_pos[b * 3 + k] = t.light_positions[lf * 3 + k];
}

__global__ void frame_setup_bwd_kernel(const harp_frame_tables t, const int32_t* __restrict__ fid, int B, int S, float focal,
                                       int self_shadow, const float* __restrict__ g_pose48, const float* __restrict__ g_betas,
                                       const float* __restrict__ g_trans_b, const float* __restrict__ g_cam_T,
                                       const float* __restrict__ g_light_pos, const float* __restrict__ g_colors) {
  // one wave per frame, one lane per element: the ~70 scatter atomics of a frame are issued at once (one lane per frame walked
  // them one after the other: 22 us of pure latency)
  const int b = blockIdx.x, k = threadIdx.x;
  if (b == 0 && k == 0 && self_shadow && g_colors && t.g_amb_ratio) {
    const float amb = 1.0f / (1.0f + expf(-t.amb_ratio[0]));
    const float g_amb = (g_colors[0] + g_colors[1] + g_colors[2]) - (g_colors[3] + g_colors[4] + g_colors[5]);
    atomicAdd(t.g_amb_ratio, g_amb * amb * (1.0f - amb));
  }
  if (b >= B) return;
  const int f = fid[b];
  // duplicates of a frame inside one batch are legal -> atomics
  const int ps = t.wrist_pose ? 51 : 48, ho = t.wrist_pose ? 6 : 3, nbo = t.n_betas_out > 0 ? t.n_betas_out : 10;
  if (g_pose48 && k < ps) {
    const float g = g_pose48[b * ps + k];
    if (k < 3) { if (t.g_rot) atomicAdd(t.g_rot + f * 3 + k, g); }
    else if (k < ho) { if (t.g_wrist_pose) atomicAdd(t.g_wrist_pose + f * 3 + (k - 3), g); }
    else if (t.g_pose) atomicAdd(t.g_pose + f * 45 + (k - ho), g);
  }
  if (g_betas && t.g_shape && k < 10) atomicAdd(t.g_shape + k, g_betas[b * nbo + k]);
  if (g_trans_b && t.g_trans && k < 3) atomicAdd(t.g_trans + f * 3 + k, g_trans_b[b * 3 + k]);
  if (g_cam_T && t.g_cam && k < 3) {
    if (k == 0) {
      const float c0 = t.cam[f * 3];
      const float den = (float)S * c0 + 1e-9f;
      atomicAdd(t.g_cam + f * 3, g_cam_T[b * 3 + 2] * (-2.0f * focal * (float)S / (den * den)));
    } else {
      atomicAdd(t.g_cam + f * 3 + k, -g_cam_T[b * 3 + (k - 1)]);
    }
  }
  if (g_light_pos && t.g_light_positions && k < 3) {
    const int lf = t.share_light ? 0 : f;
    atomicAdd(t.g_light_positions + lf * 3 + k, g_light_pos[b * 3 + k]);
  }
}

// process_info_for_shadow (renderer_helper.py:461-467) + look_at_rotation
template <bool BWD>
__global__ void light_setup_kernel(const float* __restrict__ centroid, const float* __restrict__ light_pos, int B,
                                   float* __restrict__ light_R, float* __restrict__ light_T, const float* __restrict__ g_R,
                                   const float* __restrict__ g_T, float* __restrict__ g_light_pos, float* __restrict__ g_centroid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const LightCam k = light_cam(ld(centroid + 3 * b), ld(light_pos + 3 * b));
  if (!BWD) {
    light_cam_RT(k, light_R + 9 * b, light_T + 3 * b);
  } else {
    V3 gd, gc;
    light_cam_bwd(k, g_R + 9 * b, g_T + 3 * b, gd, gc);
    float* gl = g_light_pos + 3 * b;
    gl[0] += gd.x; gl[1] += gd.y; gl[2] += gd.z;
    float* gco = g_centroid + 3 * b;
    gco[0] = gc.x; gco[1] = gc.y; gco[2] = gc.z;
  }
}

// centroid = mean_v verts  ->  g_verts[b, v, :] += g_centroid[b] / V
__global__ void centroid_bwd_kernel(const float* __restrict__ g_c, int V, float* __restrict__ g_v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= V) return;
  const float inv = 1.0f / (float)V;
  float* o = g_v + ((size_t)b * V + i) * 3;
  o[0] += g_c[3 * b] * inv; o[1] += g_c[3 * b + 1] * inv; o[2] += g_c[3 * b + 2] * inv;
}

__global__ void scale_kernel(const float* __restrict__ x, float s, int n, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * s;
}

// One row of a device-resident batch schedule -> the step's frame ids (and target-local ids); bumps the row counter.  Lives inside
// the step's hipGraph so that a replay needs no host-side copy at all.
__global__ void schedule_next_kernel(const int32_t* __restrict__ sched, const int32_t* __restrict__ tsched, int n_rows, int B, int target_offset,
                                     int32_t* __restrict__ counter, int32_t* __restrict__ fid, int32_t* __restrict__ tfid,
                                     float* __restrict__ zero, int n_zero) {
  for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero[i] = 0.f;      // (the step's loss vector: no launch of its own)
  const int row = (int)((unsigned)counter[0] % (unsigned)n_rows);
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int f = sched[(size_t)row * B + i];
    fid[i] = f;
    tfid[i] = tsched ? tsched[(size_t)row * B + i] : f - target_offset;
  }
  __syncthreads();
  if (threadIdx.x == 0) counter[0] = row + 1;
}

}  // namespace

extern "C" {

int harp_frame_setup_fwd(const harp_frame_tables* t, const int32_t* fid, int B, int S, float focal, int self_shadow, float* pose48,
                         float* betas, float* trans_b, float* cam_R, float* cam_T, float* light_pos, float* colors,
                         hipStream_t stream) {
  if (!t || !fid || !pose48 || !betas || !trans_b || !cam_R || !cam_T || !light_pos || !colors) return HARP_ERR_ARG;
  hipLaunchKernelGGL(frame_setup_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, *t, fid, B, S, focal, self_shadow, pose48, betas,
                     trans_b, cam_R, cam_T, light_pos, colors);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_frame_setup_bwd(const harp_frame_tables* t, const int32_t* fid, int B, int S, float focal, int self_shadow,
                         const float* g_pose48, const float* g_betas, const float* g_trans_b, const float* g_cam_T,
                         const float* g_light_pos, const float* g_colors, hipStream_t stream) {
  if (!t || !fid) return HARP_ERR_ARG;
  hipLaunchKernelGGL(frame_setup_bwd_kernel, dim3(B), dim3(64), 0, stream, *t, fid, B, S, focal, self_shadow, g_pose48,
                     g_betas, g_trans_b, g_cam_T, g_light_pos, g_colors);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_light_setup_fwd(const float* centroid, const float* light_pos, int B, float* light_R, float* light_T, hipStream_t stream) {
  if (!centroid || !light_pos || !light_R || !light_T) return HARP_ERR_ARG;
  hipLaunchKernelGGL(light_setup_kernel<false>, dim3((B + 63) / 64), dim3(64), 0, stream, centroid, light_pos, B, light_R, light_T,
                     nullptr, nullptr, nullptr, nullptr);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// g_light_pos (B,3) (+=); g_centroid (B,3) overwritten; then g_verts (B,V,3) (+=) g_centroid / V if g_verts != NULL
int harp_light_setup_bwd(const float* centroid, const float* light_pos, const float* g_light_R, const float* g_light_T, int B, int V,
                         float* g_light_pos, float* g_centroid, float* g_verts, hipStream_t stream) {
  if (!centroid || !light_pos || !g_light_R || !g_light_T || !g_light_pos || !g_centroid) return HARP_ERR_ARG;
  hipLaunchKernelGGL(light_setup_kernel<true>, dim3((B + 63) / 64), dim3(64), 0, stream, centroid, light_pos, B, nullptr, nullptr,
                     g_light_R, g_light_T, g_light_pos, g_centroid);
  if (g_verts) hipLaunchKernelGGL(centroid_bwd_kernel, dim3((V + 255) / 256, B), dim3(256), 0, stream, g_centroid, V, g_verts);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_scale(const float* x, float s, int n, float* y, hipStream_t stream) {
  if (!x || !y) return HARP_ERR_ARG;
  hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, s, n, y);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_schedule_next_rows(const int32_t* schedule, const int32_t* tschedule, int n_rows, int B, int target_offset, int32_t* counter,
                            int32_t* fid, int32_t* tfid, float* zero, int n_zero, hipStream_t stream) {
  if (!schedule || !counter || !fid || !tfid || n_rows <= 0 || B <= 0 || (n_zero > 0 && !zero)) return HARP_ERR_ARG;
  hipLaunchKernelGGL(schedule_next_kernel, dim3(1), dim3(256), 0, stream, schedule, tschedule, n_rows, B, target_offset, counter, fid, tfid, zero,
                     zero ? n_zero : 0);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_schedule_next(const int32_t* schedule, int n_rows, int B, int target_offset, int32_t* counter, int32_t* fid, int32_t* tfid,
                       float* zero, int n_zero, hipStream_t stream) {
  return harp_schedule_next_rows(schedule, nullptr, n_rows, B, target_offset, counter, fid, tfid, zero, n_zero, stream);
}

}  // extern "C"
