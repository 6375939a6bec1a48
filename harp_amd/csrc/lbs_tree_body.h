// Kinematic-tree LBS arithmetic shared by the stand-alone kernels (lbs_tree.hip) and the fused per-frame front / back kernels of the
// SMPL-X arm path (arm_front.hip): Rodrigues in the smplx matrix form, the workspace layout, the joint chain forward (Rodrigues, joint
// regression, level-parallel batch_rigid_transform) and its backward, as workgroup-size-generic device functions.
//   reference: hand_models_harp/body_models.py:2163-2390 (SMPLXARM.forward -> smplx.lbs :2335), SURVEY.md Appendix A.13.
#pragma once
#include "harp_common.h"
#include "harp_hip.h"

namespace lt {

constexpr int MAXJ = 64;          // joints
constexpr int MAXB = 32;          // shape coefficients

// smplx.lbs.batch_rodrigues
__device__ __forceinline__ void rod_fwd(const float r[3], float R[9]) {
  const float e0 = r[0] + 1e-8f, e1 = r[1] + 1e-8f, e2 = r[2] + 1e-8f;
  const float th = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float dx = r[0] / th, dy = r[1] / th, dz = r[2] / th;
  const float s = sinf(th), c = 1.0f - cosf(th);
  // K = [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]];  K^2 = d d^T - |d|^2 I
  const float n2 = dx * dx + dy * dy + dz * dz;
  R[0] = 1.f + c * (dx * dx - n2); R[1] = -s * dz + c * dx * dy;    R[2] = s * dy + c * dx * dz;
  R[3] = s * dz + c * dx * dy;     R[4] = 1.f + c * (dy * dy - n2); R[5] = -s * dx + c * dy * dz;
  R[6] = -s * dy + c * dx * dz;    R[7] = s * dx + c * dy * dz;     R[8] = 1.f + c * (dz * dz - n2);
}

__device__ __forceinline__ void rod_bwd(const float r[3], const float g[9], float gr[3]) {
  const float e[3] = {r[0] + 1e-8f, r[1] + 1e-8f, r[2] + 1e-8f};
  const float th = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const float d[3] = {r[0] / th, r[1] / th, r[2] / th};
  const float s = sinf(th), cs = cosf(th), c = 1.0f - cs;
  const float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const float K[9] = {0.f, -d[2], d[1], d[2], 0.f, -d[0], -d[1], d[0], 0.f};
  float K2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i * 3 + j] = d[i] * d[j] - (i == j ? n2 : 0.f);
  float g_th = 0.f;
  for (int k = 0; k < 9; ++k) g_th += g[k] * (cs * K[k] + s * K2[k]);
  // R = I + s K + c K K:  g_K = s G + c (G K^T + K^T G)
  float gK[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
      for (int m = 0; m < 3; ++m) a += g[i * 3 + m] * K[j * 3 + m] + K[m * 3 + i] * g[m * 3 + j];
      gK[i * 3 + j] = s * g[i * 3 + j] + c * a;
    }
  const float gd[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
  g_th -= (d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2]) / th;
  for (int k = 0; k < 3; ++k) gr[k] = gd[k] / th + g_th * e[k] / th;
}


// workspace of harp_lbs_tree_fwd / bwd (harp_lbs_tree_ws_floats)
struct TreeWs { float *pm, *A, *G, *Jrest, *Rloc, *vp, *g_vp, *g_A, *g_pm, *g_Gt; };
__host__ __device__ inline TreeWs tree_ws(const harp_tree_model* m, float* ws, int B) {
  const size_t NJ = m->NJ, NV = m->NV, NP = (NJ - 1) * 9;
  TreeWs w; float* p = ws;
  w.pm = p; p += B * NP; w.A = p; p += B * NJ * 12; w.G = p; p += B * NJ * 12; w.Jrest = p; p += B * NJ * 3; w.Rloc = p; p += B * NJ * 9;
  w.vp = p; p += B * NV * 3; w.g_vp = p; p += B * NV * 3;
  w.g_A = p; p += B * NJ * 12; w.g_pm = p; p += B * NP; w.g_Gt = p;     // g_pm | g_Gt adjacent (zeroed together)
  return w;
}

// LDS of the joint chain (forward): one instance per workgroup
struct JointsLds {
  float sR[MAXJ][9], sJ[MAXJ][3], sG[MAXJ][12];
  int sPar[MAXJ];     // parents[] in LDS: a lane's depth is a walk of up to 11 DEPENDENT reads (11 memory round trips from global memory)
  int maxd;
};

// Joint chain of frame b by a workgroup of T threads (T >= 64, a multiple of 64; every thread of the workgroup must call).
// pose_row: this frame's (n_pose_in, 3) axis-angle rows, beta_row: its NB shape coefficients (global or LDS).
// Writes pose_map (NP), A, G (NJ,12), Jrest (NJ,3), Rloc (NJ,9) rows of frame b and clears the frame's g_pm / g_Gt rows (the backward pass
// accumulates into them with atomics; the chain backward clears what it consumes, so a second backward call starts from zero as well).
template <int T>
__device__ __forceinline__ void joints_body(const harp_tree_model& M, const float* pose_row, const float* beta_row, int b, const TreeWs& W,
                                            JointsLds& S) {
  const int l = threadIdx.x, NJ = M.NJ, NB = M.NB, NP = (NJ - 1) * 9;
  if (l < NJ) S.sPar[l] = M.parents[l];
  if (l == 0) S.maxd = 0;
  for (int k = l; k < NP; k += T) W.g_pm[(size_t)b * NP + k] = 0.f;
  for (int k = l; k < NJ * 3; k += T) W.g_Gt[(size_t)b * NJ * 3 + k] = 0.f;
  // Rodrigues: the LAST NJ-rounded-up-to-64 threads of the workgroup when it has more than one wave (the joint regression below then runs
  // on the other waves at the same time), the only wave otherwise
  {
    const int j = (T > 64) ? l - (T - 64) : l;
    if (j >= 0 && j < NJ) {
      float aa[3];
      const int src = M.pose_src[j];
      for (int c = 0; c < 3; ++c) aa[c] = M.pose_mean[3 * j + c] + (src >= 0 ? pose_row[src * 3 + c] : 0.f);
      float R[9];
      rod_fwd(aa, R);
      for (int k = 0; k < 9; ++k) { S.sR[j][k] = R[k]; W.Rloc[((size_t)b * NJ + j) * 9 + k] = R[k]; }
      if (j > 0)
        for (int k = 0; k < 9; ++k) W.pm[(size_t)b * NP + (j - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
    }
  }
  // joint regression J = J_template + J_dirs beta: the NB rows of a joint coordinate are requested 8 at a time (the loop with one load per
  // trip was 3 x NB dependent round trips per lane: most of this kernel's 21 us); same summation order
  for (int i = l; i < NJ * 3; i += T) {
    float acc = M.J_template[i];
    for (int k0 = 0; k0 < NB; k0 += 8) {
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = M.J_dirs[i * NB + min(k0 + u, NB - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < NB) acc += d[u] * beta_row[k0 + u];
    }
    S.sJ[i / 3][i % 3] = acc;
    W.Jrest[(size_t)b * NJ * 3 + i] = acc;
  }
  __syncthreads();
  // batch_rigid_transform: chain[i] = chain[parent] @ [R_i | J_i - J_parent].  Level-parallel: thread j waits until its parent's depth has
  // been processed (the SMPL-X right-arm tree is 11 levels deep, 55 joints: 11 steps instead of 55 serial products on one lane)
  const int j = l;
  int depth = 0;
  if (j < NJ) {
    for (int q = S.sPar[j]; q >= 0; q = S.sPar[q]) ++depth;
    atomicMax(&S.maxd, depth);
  }
  if (j == 0) {
    for (int k = 0; k < 9; ++k) S.sG[0][(k / 3) * 4 + (k % 3)] = S.sR[0][k];
    for (int r = 0; r < 3; ++r) S.sG[0][r * 4 + 3] = S.sJ[0][r];
  }
  __syncthreads();
  const int maxd = S.maxd;
  for (int d = 1; d <= maxd; ++d) {
    if (j < NJ && depth == d) {
      const int p = S.sPar[j];
      const float rel[3] = {S.sJ[j][0] - S.sJ[p][0], S.sJ[j][1] - S.sJ[p][1], S.sJ[j][2] - S.sJ[p][2]};
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          S.sG[j][r * 4 + c] = S.sG[p][r * 4] * S.sR[j][c] + S.sG[p][r * 4 + 1] * S.sR[j][3 + c] + S.sG[p][r * 4 + 2] * S.sR[j][6 + c];
        S.sG[j][r * 4 + 3] = S.sG[p][r * 4] * rel[0] + S.sG[p][r * 4 + 1] * rel[1] + S.sG[p][r * 4 + 2] * rel[2] + S.sG[p][r * 4 + 3];
      }
    }
    __syncthreads();
  }
  for (int jj = l; jj < NJ; jj += T) {
    float* Ao = W.A + ((size_t)b * NJ + jj) * 12;
    float* Go = W.G + ((size_t)b * NJ + jj) * 12;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { Ao[r * 4 + c] = S.sG[jj][r * 4 + c]; Go[r * 4 + c] = S.sG[jj][r * 4 + c]; }
      Go[r * 4 + 3] = S.sG[jj][r * 4 + 3];
      Ao[r * 4 + 3] = S.sG[jj][r * 4 + 3] - (S.sG[jj][r * 4] * S.sJ[jj][0] + S.sG[jj][r * 4 + 1] * S.sJ[jj][1] + S.sG[jj][r * 4 + 2] * S.sJ[jj][2]);
    }
  }
}

// LDS of the chain backward
// (gRG / gtG / gJ collect the children's contributions with LDS atomics: double, because ds_add_f32 costs 193 clk per wave instruction on
//  gfx950 against 8.7 for ds_add_f64 — 15 of them per level x 11 levels were 13 of this kernel's 31 us)
struct ChainBwdLds {
  double gRG[MAXJ][9], gtG[MAXJ][3], gJ[MAXJ][3];
  float gRl[MAXJ][9];
  int sPar[MAXJ];
  int maxd;
};

// Chain + Rodrigues backward of frame b by ONE wave (64 threads).  pose_row / g_pose_row: this frame's (n_pose_in, 3) rows; g_beta_row: its
// NB shape gradients (+=, holds the blend-shape share on entry).  Consumes (and clears) the frame's g_pm / g_Gt rows.
// SCATTER: the pose / shape gradients are ALSO added to the rows tables.g_rot / g_wrist_pose / g_pose / g_shape of frame f — what
// harp_frame_setup_bwd does with the (B, n_pose_in*3) / (B, NB) buffers as a launch of its own (rows [rot, wrist_pose, pose(15)]).
template <bool SCATTER>
__device__ __forceinline__ void chain_bwd_body(const harp_tree_model& M, const float* pose_row, int b, const TreeWs& W, float* g_pose_row,
                                               float* g_beta_row, ChainBwdLds& S, const harp_frame_tables* tables, int f) {
  auto& gRG = S.gRG; auto& gtG = S.gtG; auto& gRl = S.gRl; auto& gJ = S.gJ;
  const int l = threadIdx.x, NJ = M.NJ, NB = M.NB, NP = (NJ - 1) * 9;
  const float* Gb = W.G + (size_t)b * NJ * 12;
  const float* Rb = W.Rloc + (size_t)b * NJ * 9;
  const float* Jb = W.Jrest + (size_t)b * NJ * 3;
  if (l < NJ) S.sPar[l] = M.parents[l];
  for (int j = l; j < NJ; j += 64) {
    const float* ga = W.g_A + ((size_t)b * NJ + j) * 12;
    for (int r = 0; r < 3; ++r) {
      const float gt = ga[r * 4 + 3];
      for (int c = 0; c < 3; ++c) gRG[j][r * 3 + c] = ga[r * 4 + c] - gt * Jb[j * 3 + c];
      gtG[j][r] = gt + W.g_Gt[((size_t)b * NJ + j) * 3 + r];
      W.g_Gt[((size_t)b * NJ + j) * 3 + r] = 0.f;          // consumed: the two accumulators are all-zero again for the next backward call
    }
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      for (int r = 0; r < 3; ++r) acc -= Gb[j * 12 + r * 4 + c] * ga[r * 4 + 3];
      gJ[j][c] = acc;
    }
    for (int k = 0; k < 9; ++k) {
      gRl[j][k] = (j > 0) ? W.g_pm[(size_t)b * NP + (j - 1) * 9 + k] : 0.f;
      if (j > 0) W.g_pm[(size_t)b * NP + (j - 1) * 9 + k] = 0.f;
    }
  }
  __syncthreads();
  // chain backward, level-parallel from the leaves up: lane j (depth d) is final once every deeper level has been folded into it;
  // siblings add into their common parent with LDS atomics (a few dozen per frame)
  {
    const int j = l;
    int depth = 0;
    if (j < NJ) for (int q = S.sPar[j]; q >= 0; q = S.sPar[q]) ++depth;
    int maxd = depth;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) maxd = max(maxd, __shfl_xor(maxd, o2, 64));
    // this lane's share of the forward state (its parent's rotation, its own local rotation, its offset from the parent) is fetched ONCE,
    // in front of the level loop: fetched inside it, every one of the 11 levels paid a memory round trip behind its barrier (~2 us each)
    float Gp[9], Rj[9], rel[3] = {0.f, 0.f, 0.f};
    const int p = (j < NJ && j > 0) ? S.sPar[j] : 0;
    {
      const int jc = min(j, NJ - 1);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { Gp[r * 3 + c] = Gb[p * 12 + r * 4 + c]; Rj[r * 3 + c] = Rb[jc * 9 + r * 3 + c]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) rel[c] = Jb[jc * 3 + c] - Jb[p * 3 + c];
    }
    for (int d = maxd; d >= 1; --d) {
      __syncthreads();
      if (j < NJ && depth == d) {
        float rg[9], tg[3], gl[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) rg[k] = (float)gRG[j][k];
#pragma unroll
        for (int c = 0; c < 3; ++c) tg[c] = (float)gtG[j][c];
        for (int c = 0; c < 3; ++c) gl[c] = Gp[c] * tg[0] + Gp[3 + c] * tg[1] + Gp[6 + c] * tg[2];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            gRl[j][r * 3 + c] += Gp[r] * rg[c] + Gp[3 + r] * rg[3 + c] + Gp[6 + r] * rg[6 + c];
            atomicAdd(&gRG[p][r * 3 + c], (double)(rg[r * 3] * Rj[c * 3] + rg[r * 3 + 1] * Rj[c * 3 + 1] + rg[r * 3 + 2] * Rj[c * 3 + 2] + tg[r] * rel[c]));
          }
        for (int c = 0; c < 3; ++c) { atomicAdd(&gtG[p][c], (double)tg[c]); atomicAdd(&gJ[p][c], (double)(-gl[c])); gJ[j][c] += (double)gl[c]; }
      }
    }
    __syncthreads();
    if (j == 0) {
      for (int k = 0; k < 9; ++k) gRl[0][k] += (float)gRG[0][k];
      for (int c = 0; c < 3; ++c) gJ[0][c] += gtG[0][c];
    }
  }
  __syncthreads();
  for (int j = l; j < NJ; j += 64) {
    const int src = M.pose_src[j];
    if (src < 0) continue;
    float aa[3], gaa[3];
    for (int c = 0; c < 3; ++c) aa[c] = M.pose_mean[3 * j + c] + pose_row[src * 3 + c];
    rod_bwd(aa, gRl[j], gaa);
    for (int c = 0; c < 3; ++c) {
      g_pose_row[src * 3 + c] = gaa[c];
      if (SCATTER) {
        // rows of the arm's pose input: 0 = rot, 1 = wrist_pose, 2.. = pose (15 x 3); duplicates of a frame in one batch are legal -> atomics
        float* dst = (src == 0) ? tables->g_rot : (src == 1) ? tables->g_wrist_pose : tables->g_pose;
        if (dst) atomicAdd(dst + (src <= 1 ? f * 3 + c : f * 45 + (src - 2) * 3 + c), gaa[c]);
      }
    }
  }
  // shape gradient through the joint regression: g_beta[k] += sum_i J_dirs[i][k] gJ[i] over the NJ*3 = 165 rows.  Lanes = (coefficient k,
  // one of 64 / NB chunks of the rows), 28 rows in flight: two or three round trips instead of 165 (one load per trip) or 21 (8 per trip)
  {
    const int nch = max(1, 64 / NB), rows = NJ * 3, rper = (rows + nch - 1) / nch;
    const int k = l % NB, ch = l / NB;
    float part = 0.f;
    if (NB <= 64 && ch < nch) {
      const double* gJf = &gJ[0][0];
      const int r0 = ch * rper, r1 = min(rows, r0 + rper);
      for (int i0 = r0; i0 < r1; i0 += 28) {
        float d[28];
#pragma unroll
        for (int u = 0; u < 28; ++u) d[u] = M.J_dirs[min(i0 + u, rows - 1) * NB + k];
#pragma unroll
        for (int u = 0; u < 28; ++u)
          if (i0 + u < r1) part += d[u] * (float)gJf[i0 + u];
      }
    }
    __syncthreads();                                   // gRl is dead: its first 64 floats carry the partial sums
    float* s_part = &gRl[0][0];
    s_part[l] = part;
    __syncthreads();
    if (l < NB) {
      float acc = g_beta_row[l];
      for (int c = 0; c < nch; ++c) acc += s_part[c * NB + l];
      g_beta_row[l] = acc;
      if (SCATTER && l < 10 && tables->g_shape) atomicAdd(tables->g_shape + l, acc);
    }
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace lt
