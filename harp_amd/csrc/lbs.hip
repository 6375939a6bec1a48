// MANO linear-blend skinning for gfx950, forward and hand-written backward.
// Replaces ManoLayer.forward (manopth/manolayer.py:108-296; batch_rodrigues rodrigues_layer.py:43-54, quat2mat :15-40,
// th_posemap_axisang tensutils.py:6-12) for the configuration HARP builds (use_pca=False, flat_hand_mean=False,
// axis-angle root, right hand: utils/hand_model_utils.py:74) — ~35 tiny torch kernels per call in the reference.
//
//   joints kernel   one wave per frame: Rodrigues (via quaternion, with the reference's norm(aa + 1e-8)) for the 16
//                   joints in 16 lanes, J = J_template + J_dirs . beta (J_regressor folded into the shape basis on the
//                   host), the 3-level kinematic chain, A_j = [R_j | t_j - R_j J_j]; writes pose_map (B,135), A (B,16,12)
//                   and the 16 joint positions.
//   skin kernel     one lane per vertex, 8 frames per workgroup sharing every blend-shape read:
//                   v_posed = v_t + S beta + P pose_map (the (B,135)x(135,2334) contraction, transposed bases so that
//                   lanes read consecutive floats), T = sum_j w_j A_j, out = (T [v_posed;1] + trans) * 1000.
//   backward        skin_bwd (per vertex: g_v_posed, M = g_out (x) [v_posed;1]) -> two small reductions over vertices
//                   (g_A = W^T M, g_pose_map = P^T g_vp, g_beta) -> chain_bwd (one wave per frame: chain, Rodrigues).
#include "lbs_body.h"
#include "harp_hip.h"

namespace {

using namespace lb;
constexpr int FRAMES_PER_BLOCK = 8;

// one wave per frame
__global__ void __launch_bounds__(64) lbs_joints_kernel(const harp_mano_model M, const float* __restrict__ pose,
                                                        const float* __restrict__ betas, float* __restrict__ pose_map,
                                                        float* __restrict__ A, float* __restrict__ j16,
                                                        float* __restrict__ Jrest, float* __restrict__ Rloc,
                                                        float* __restrict__ G) {
  __shared__ float sR[NJ][9], sJ[NJ][3], sG[NJ][12];
  const int b = blockIdx.x, l = threadIdx.x;
  if (l < NJ) {
    float aa[3];
    for (int c = 0; c < 3; ++c) {
      const float p = pose[b * 48 + 3 * l + c];
      aa[c] = (l == 0) ? p : (M.hands_mean[3 * (l - 1) + c] + p);       // manolayer.py:139-143
    }
    float R[9];
    rodrigues_fwd(aa, R);
    for (int k = 0; k < 9; ++k) { sR[l][k] = R[k]; Rloc[(b * NJ + l) * 9 + k] = R[k]; }
    if (l > 0)
      for (int k = 0; k < 9; ++k) pose_map[b * NP + (l - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
  }
  if (l < NJ * 3) {
    float acc = M.J_template[l];
    for (int k = 0; k < NB; ++k) acc += M.J_dirs[l * NB + k] * betas[b * NB + k];
    sJ[l / 3][l % 3] = acc;
    Jrest[b * NJ * 3 + l] = acc;
  }
  __syncthreads();
  // kinematic chain: lanes 0..4 own one finger each (after the root, lane 0)
  if (l == 0) {
    for (int k = 0; k < 9; ++k) sG[0][(k / 3) * 4 + (k % 3)] = sR[0][k];
    for (int r = 0; r < 3; ++r) sG[0][r * 4 + 3] = sJ[0][r];
  }
  __syncthreads();
  if (l < 5) {
    for (int lev = 0; lev < 3; ++lev) {
      const int j = 3 * l + 1 + lev, p = parent_of(j);
      float rel[3] = {sJ[j][0] - sJ[p][0], sJ[j][1] - sJ[p][1], sJ[j][2] - sJ[p][2]};
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          sG[j][r * 4 + c] = sG[p][r * 4] * sR[j][c] + sG[p][r * 4 + 1] * sR[j][3 + c] + sG[p][r * 4 + 2] * sR[j][6 + c];
        sG[j][r * 4 + 3] = sG[p][r * 4] * rel[0] + sG[p][r * 4 + 1] * rel[1] + sG[p][r * 4 + 2] * rel[2] + sG[p][r * 4 + 3];
      }
    }
  }
  __syncthreads();
  if (l < NJ) {
    float* Ao = A + (b * NJ + l) * 12;
    float* Go = G + (b * NJ + l) * 12;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { Ao[r * 4 + c] = sG[l][r * 4 + c]; Go[r * 4 + c] = sG[l][r * 4 + c]; }
      Go[r * 4 + 3] = sG[l][r * 4 + 3];
      Ao[r * 4 + 3] = sG[l][r * 4 + 3] - (sG[l][r * 4] * sJ[l][0] + sG[l][r * 4 + 1] * sJ[l][1] + sG[l][r * 4 + 2] * sJ[l][2]);  // :241-247
      j16[(b * NJ + l) * 3 + r] = sG[l][r * 4 + 3];
    }
  }
}

// blockIdx.x: vertex chunk (256), blockIdx.y: frame chunk (8)
constexpr int kSkinVerts = 64, kSkinSlices = 4;     // 256 threads = 64 vertices x 4 blend-shape slices
template <bool BWD>
__global__ void __launch_bounds__(256) lbs_skin_kernel(const harp_mano_model M, const float* __restrict__ betas,
                                                       const float* __restrict__ trans, const float* __restrict__ pose_map,
                                                       const float* __restrict__ A, int B, float* __restrict__ verts,
                                                       const float* __restrict__ g_verts, float* __restrict__ g_vp,
                                                       float* __restrict__ Mo, float* __restrict__ vposed) {
  // 64 vertices x 4 slices of the blend-shape index per workgroup: a lane walks 135/4 + 10/4 dependent load trips instead of 145
  // (the kernel is a chain of L2 round trips, not arithmetic: 16 workgroups of 256 vertices took 31 us), partial sums meet in LDS,
  // then every lane skins its vertex for the frames f = slice, slice + 4, ...
  __shared__ float s_pm[FRAMES_PER_BLOCK][NP], s_beta[FRAMES_PER_BLOCK][NB], s_A[FRAMES_PER_BLOCK][NJ * 12];
  __shared__ float s_part[kSkinSlices][FRAMES_PER_BLOCK * 3][kSkinVerts];
  const int vl = threadIdx.x & (kSkinVerts - 1), ks = threadIdx.x / kSkinVerts;
  const int v = blockIdx.x * kSkinVerts + vl;
  const int b0 = blockIdx.y * FRAMES_PER_BLOCK;
  const int nb = min(FRAMES_PER_BLOCK, B - b0);
  // Every load of the (static) model tables is issued BEFORE the staging barrier: the rows come from HBM / MALL (1.26 MB of pose
  // blend shapes, evicted from L2 by the rest of the step), so the kernel is a chain of DRAM round trips, not arithmetic — with all of
  // them in flight at once (this slice's 34 pose rows, 3 shape rows, template, 16 skinning weights) it pays one trip instead of a dozen.
  const bool ok = v < NV;
  const int vc = ok ? v : NV - 1;
  constexpr int kRowsP = (NP + kSkinSlices - 1) / kSkinSlices, kRowsB = (NB + kSkinSlices - 1) / kSkinSlices;
  float pr[BWD ? 1 : kRowsP][3], sr[BWD ? 1 : kRowsB][3], w[NJ], tp[3];
  if (!BWD) {
#pragma unroll
    for (int u = 0; u < kRowsP; ++u) {
      const int k = min(ks + u * kSkinSlices, NP - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) pr[BWD ? 0 : u][c] = M.posedirs_T[k * NV * 3 + 3 * vc + c];
    }
#pragma unroll
    for (int u = 0; u < kRowsB; ++u) {
      const int k = min(ks + u * kSkinSlices, NB - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) sr[BWD ? 0 : u][c] = M.shapedirs_T[k * NV * 3 + 3 * vc + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) tp[c] = (ks == 0) ? M.v_template[3 * vc + c] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) w[j] = M.weights[vc * NJ + j];
  if (!BWD) {
    for (int i = threadIdx.x; i < nb * NP; i += 256) s_pm[i / NP][i % NP] = pose_map[b0 * NP + i];
    for (int i = threadIdx.x; i < nb * NB; i += 256) s_beta[i / NB][i % NB] = betas[b0 * NB + i];
  }
  for (int i = threadIdx.x; i < nb * NJ * 12; i += 256) s_A[i / (NJ * 12)][i % (NJ * 12)] = A[b0 * NJ * 12 + i];
  __syncthreads();
  float vp[BWD ? 1 : FRAMES_PER_BLOCK][3];
  if (!BWD) {
#pragma unroll
  for (int f = 0; f < FRAMES_PER_BLOCK; ++f) { vp[f][0] = tp[0]; vp[f][1] = tp[1]; vp[f][2] = tp[2]; }
#pragma unroll
  for (int u = 0; u < kRowsB; ++u) {
    const int k = ks + u * kSkinSlices;
    if (k < NB) {
#pragma unroll
      for (int f = 0; f < FRAMES_PER_BLOCK; ++f) { const float c = s_beta[f][k]; vp[f][0] += sr[u][0] * c; vp[f][1] += sr[u][1] * c; vp[f][2] += sr[u][2] * c; }
    }
  }
#pragma unroll
  for (int u = 0; u < kRowsP; ++u) {
    const int k = ks + u * kSkinSlices;
    if (k < NP) {
#pragma unroll
      for (int f = 0; f < FRAMES_PER_BLOCK; ++f) { const float c = s_pm[f][k]; vp[f][0] += pr[u][0] * c; vp[f][1] += pr[u][1] * c; vp[f][2] += pr[u][2] * c; }
    }
  }
#pragma unroll
  for (int f = 0; f < FRAMES_PER_BLOCK; ++f)
#pragma unroll
    for (int c = 0; c < 3; ++c) s_part[ks][f * 3 + c][vl] = vp[BWD ? 0 : f][c];
  __syncthreads();
  }   // !BWD
  if (!ok) return;
  for (int f = ks; f < nb; f += kSkinSlices) {
    float q[3];
    if (!BWD) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float a = 0.f;
#pragma unroll
        for (int sl = 0; sl < kSkinSlices; ++sl) a += s_part[sl][f * 3 + c][vl];
        q[c] = a;
      }
      // the posed vertex is kept in the workspace for the backward pass: that kernel then needs neither the blend-shape rows (a DRAM
      // round trip over 1.26 MB) nor the partial-sum exchange
      for (int c = 0; c < 3; ++c) vposed[((size_t)(b0 + f) * NV + v) * 3 + c] = q[c];
    } else {
      for (int c = 0; c < 3; ++c) q[c] = vposed[((size_t)(b0 + f) * NV + v) * 3 + c];
    }
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int k = 0; k < 12; ++k) T[k] += w[j] * s_A[f][j * 12 + k];
    const int b = b0 + f;
    if (!BWD) {
      for (int r = 0; r < 3; ++r) {
        const float o = T[r * 4] * q[0] + T[r * 4 + 1] * q[1] + T[r * 4 + 2] * q[2] + T[r * 4 + 3];
        verts[((size_t)b * NV + v) * 3 + r] = (o + trans[3 * b + r]) * 1000.0f;
      }
    } else {
      float g[3];
      for (int r = 0; r < 3; ++r) g[r] = g_verts[((size_t)b * NV + v) * 3 + r] * 1000.0f;
      for (int c = 0; c < 3; ++c) g_vp[((size_t)b * NV + v) * 3 + c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
      float* mo = Mo + ((size_t)b * NV + v) * 12;
      for (int r = 0; r < 3; ++r) {
        mo[r * 4] = g[r] * q[0]; mo[r * 4 + 1] = g[r] * q[1]; mo[r * 4 + 2] = g[r] * q[2]; mo[r * 4 + 3] = g[r];
      }
    }
  }
}

// g_A[b][j][k] += sum_{v in chunk} w[v][j] M[b][v][k]   (192 lanes per frame x kChunksA vertex chunks; g_A pre-zeroed)
constexpr int kChunksA = 16;     // (round 4: 8 -> 16 chunks, a lane walks 49 vertices instead of 98: the launch 15.5 -> 13.7 us; 32: 14.2)
// g_pose_map[b][k] += sum_{vc in chunk} posedirs[vc][k] g_vp[b][vc];  g_beta_part[b][k] likewise with shapedirs (k<10)
constexpr int kChunksP = 64;      // short dependent load chains: 37 trips per lane instead of 146 (23 -> ~8 us)

// one wave per frame: chain + Rodrigues backward. g_j16 (B,16,3): gradient on the 16 chain joint positions (metres).
// SCATTER (harp_hand_back_bwd): instead of writing g_pose (B,48) / g_beta_b (B,10) for a later harp_frame_setup_bwd launch, add them
// straight to the rows of the parameter tables' gradient arena (rot, pose rows of frame fid[b]; shape summed over the frames).
template <bool SCATTER>
__global__ void __launch_bounds__(64) lbs_chain_bwd_kernel(const harp_mano_model M, const float* __restrict__ pose,
                                                           const float* __restrict__ Rloc, const float* __restrict__ G,
                                                           const float* __restrict__ Jrest, const float* __restrict__ g_A,
                                                           const float* __restrict__ g_pm, const float* __restrict__ g_j16,
                                                           float* __restrict__ g_pose, float* __restrict__ g_beta_b,
                                                           const harp_frame_tables T, const int32_t* __restrict__ fid) {
  __shared__ float gRG[NJ][9], gtG[NJ][3], gRl[NJ][9], gJ[NJ][3];
  const int b = blockIdx.x, l = threadIdx.x;
  const float* Gb = G + b * NJ * 12;
  const float* Rb = Rloc + b * NJ * 9;
  const float* Jb = Jrest + b * NJ * 3;
  if (l < NJ) {
    const float* ga = g_A + (b * NJ + l) * 12;
    for (int r = 0; r < 3; ++r) {
      const float gt = ga[r * 4 + 3];
      for (int c = 0; c < 3; ++c) gRG[l][r * 3 + c] = ga[r * 4 + c] - gt * Jb[l * 3 + c];
      gtG[l][r] = gt + g_j16[(b * NJ + l) * 3 + r];
    }
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      for (int r = 0; r < 3; ++r) acc -= Gb[l * 12 + r * 4 + c] * ga[r * 4 + 3];
      gJ[l][c] = acc;
    }
    for (int k = 0; k < 9; ++k) gRl[l][k] = (l > 0) ? g_pm[b * NP + (l - 1) * 9 + k] : 0.f;
  }
  // leaves -> root, one finger per lane; the root accumulators are touched by 5 lanes -> LDS float atomics
  // a finger's share of the forward state (parent rotations, local rotations, offsets of its three joints) is fetched ahead of the barrier
  // by the lane that walks the finger — fetched inside the walk, every level paid a memory round trip
  float Gp3[3][9], Rj3[3][9], rel3[3][3];
  if (l < 5) {
#pragma unroll
    for (int lev = 0; lev < 3; ++lev) {
      const int j = 3 * l + 1 + lev, p = parent_of(j);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { Gp3[lev][r * 3 + c] = Gb[p * 12 + r * 4 + c]; Rj3[lev][r * 3 + c] = Rb[j * 9 + r * 3 + c]; }
        rel3[lev][r] = Jb[j * 3 + r] - Jb[p * 3 + r];
      }
    }
  }
  __syncthreads();
  if (l < 5) {
#pragma unroll
    for (int lev = 2; lev >= 0; --lev) {
      const int j = 3 * l + 1 + lev, p = parent_of(j);
      const float* Gp = Gp3[lev];
      const float* Rj = Rj3[lev];
      const float* rel = rel3[lev];
      float gl[3];
      for (int c = 0; c < 3; ++c) gl[c] = Gp[c] * gtG[j][0] + Gp[3 + c] * gtG[j][1] + Gp[6 + c] * gtG[j][2];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          // g_Rloc_j = RG_p^T g_RG_j
          gRl[j][r * 3 + c] += Gp[r] * gRG[j][c] + Gp[3 + r] * gRG[j][3 + c] + Gp[6 + r] * gRG[j][6 + c];
          // g_RG_p += g_RG_j R_j^T + g_tG_j (x) rel
          const float add = gRG[j][r * 3] * Rj[c * 3] + gRG[j][r * 3 + 1] * Rj[c * 3 + 1] + gRG[j][r * 3 + 2] * Rj[c * 3 + 2] + gtG[j][r] * rel[c];
          if (p == 0) atomicAdd(&gRG[0][r * 3 + c], add); else gRG[p][r * 3 + c] += add;
        }
      for (int c = 0; c < 3; ++c) {
        if (p == 0) { atomicAdd(&gtG[0][c], gtG[j][c]); atomicAdd(&gJ[0][c], -gl[c]); }
        else { gtG[p][c] += gtG[j][c]; gJ[p][c] -= gl[c]; }
        gJ[j][c] += gl[c];
      }
    }
  }
  __syncthreads();
  if (l == 0) {
    for (int k = 0; k < 9; ++k) gRl[0][k] += gRG[0][k];
    for (int c = 0; c < 3; ++c) gJ[0][c] += gtG[0][c];
  }
  __syncthreads();
  if (l < NJ) {
    float aa[3];
    for (int c = 0; c < 3; ++c) {
      const float p = pose[b * 48 + 3 * l + c];
      aa[c] = (l == 0) ? p : (M.hands_mean[3 * (l - 1) + c] + p);
    }
    float gaa[3];
    rodrigues_bwd(aa, gRl[l], gaa);
    if (SCATTER) {
      const int f = fid[b];
      for (int c = 0; c < 3; ++c) {
        if (l == 0) { if (T.g_rot) atomicAdd(T.g_rot + f * 3 + c, gaa[c]); }
        else if (T.g_pose) atomicAdd(T.g_pose + f * 45 + 3 * (l - 1) + c, gaa[c]);
      }
    } else {
      for (int c = 0; c < 3; ++c) g_pose[b * 48 + 3 * l + c] = gaa[c];
    }
  }
  if (l < NB) {
    float acc = g_beta_b[b * NB + l];
    for (int i = 0; i < NJ * 3; ++i) acc += M.J_dirs[i * NB + l] * gJ[i / 3][i % 3];
    if (SCATTER) { if (T.g_shape) atomicAdd(T.g_shape + l, acc); }
    else g_beta_b[b * NB + l] = acc;
  }
}

// the two independent reductions of the skinning backward in ONE launch: blockIdx.y < kChunksA -> g_A chunk, else g_pm / g_beta chunk
__global__ void __launch_bounds__(192) lbs_gA_gpm_kernel(const harp_mano_model M, const float* __restrict__ weights, const float* __restrict__ Mo,
                                                         float* __restrict__ g_A, const float* __restrict__ g_vp, float* __restrict__ g_pm,
                                                         float* __restrict__ g_beta_b) {
  const int b = blockIdx.x;
  if ((int)blockIdx.y < kChunksA) {
    const int j = threadIdx.x / 12, k = threadIdx.x % 12;
    const int per = (NV + kChunksA - 1) / kChunksA, v0 = blockIdx.y * per, v1 = min(NV, v0 + per);
    float acc = 0.f;
    // (explicit batches: `#pragma unroll 4` left 12 dependent round trips per lane for the 49 vertices of a chunk, and its remainder loop
    //  one load per trip; 17 operand pairs in flight = 3 trips, same summation order)
    constexpr int kU = 17;
    for (int va = v0; va < v1; va += kU) {
      float w[kU], m[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) { const int vc = min(va + u, v1 - 1); w[u] = weights[vc * NJ + j]; m[u] = Mo[((size_t)b * NV + vc) * 12 + k]; }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (va + u < v1) acc += w[u] * m[u];
    }
    atomicAdd(&g_A[(b * NJ + j) * 12 + k], acc);
    return;
  }
  const int k = threadIdx.x, cy = blockIdx.y - kChunksA;
  const int per = (NV * 3 + kChunksP - 1) / kChunksP, i0 = cy * per, i1 = min(NV * 3, i0 + per);
  const float* g = g_vp + (size_t)b * NV * 3;
  constexpr int kUp = 19;                  // 37 rows per chunk: two batches
  if (k < NP) {
    float acc = 0.f;
    for (int ia = i0; ia < i1; ia += kUp) {
      float d[kUp], gg[kUp];
#pragma unroll
      for (int u = 0; u < kUp; ++u) { const int ic = min(ia + u, i1 - 1); d[u] = M.posedirs[ic * NP + k]; gg[u] = g[ic]; }
#pragma unroll
      for (int u = 0; u < kUp; ++u)
        if (ia + u < i1) acc += d[u] * gg[u];
    }
    atomicAdd(&g_pm[b * NP + k], acc);
  } else if (k < NP + NB) {
    const int kk = k - NP;
    float acc = 0.f;
    for (int ia = i0; ia < i1; ia += kUp) {
      float d[kUp], gg[kUp];
#pragma unroll
      for (int u = 0; u < kUp; ++u) { const int ic = min(ia + u, i1 - 1); d[u] = M.shapedirs_T[kk * NV * 3 + ic]; gg[u] = g[ic]; }
#pragma unroll
      for (int u = 0; u < kUp; ++u)
        if (ia + u < i1) acc += d[u] * gg[u];
    }
    atomicAdd(&g_beta_b[b * NB + kk], acc);
  }
}

// joints (B,21,3) mm from chain joints (m, no trans) and tip vertices (already mm, trans included)
__global__ void lbs_joints_out_kernel(const float* __restrict__ j16, const float* __restrict__ verts, const float* __restrict__ trans,
                                      int B, float* __restrict__ joints) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 21 * 3) return;
  const int b = i / 63, k = (i % 63) / 3, c = i % 3, src = c_reorder[k];
  joints[i] = (src < NJ) ? (j16[(b * NJ + src) * 3 + c] + trans[3 * b + c]) * 1000.0f
                         : verts[((size_t)b * NV + c_tips[src - NJ]) * 3 + c];
}

// One workgroup per frame: splits g_joints (21,3) into g_j16 (16,3) [metres] and adds the finger-tip part into g_verts (778,3);
// then g_trans[b] = 1000 * (sum_v g_verts + sum_{chain joints} g_joints).  All workgroups together also clear the two reduction
// buffers of the later stages (g_A | g_pm adjacent in the workspace, g_betas).
__global__ void __launch_bounds__(256) lbs_joints_bwd_kernel(const float* __restrict__ g_joints, int B, float* __restrict__ g_j16,
                                                             float* __restrict__ g_verts, float* __restrict__ g_trans,
                                                             float* __restrict__ za, int na, float* __restrict__ zb, int nb) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x, i = b * 256 + t;
  for (int k = i; k < na; k += gridDim.x * 256) za[k] = 0.f;
  for (int k = i; k < nb; k += gridDim.x * 256) zb[k] = 0.f;
  if (t < 63) {
    const int k = t / 3, c = t % 3, src = c_reorder[k];
    const float gj = g_joints[b * 63 + t];
    if (src < NJ) g_j16[(b * NJ + src) * 3 + c] = gj * 1000.0f;
    else g_verts[((size_t)b * NV + c_tips[src - NJ]) * 3 + c] += gj;
  }
  __syncthreads();                                     // the tip gradients and g_j16 of this frame are read back below
  float a[3] = {0.f, 0.f, 0.f};
  for (int v = t; v < NV; v += 256)
    for (int c = 0; c < 3; ++c) a[c] += g_verts[((size_t)b * NV + v) * 3 + c] * 1000.0f;
  for (int j = t; j < NJ; j += 256)
    for (int c = 0; c < 3; ++c) a[c] += g_j16[(b * NJ + j) * 3 + c];
  for (int c = 0; c < 3; ++c) {
    const float sum = block_sum_256(a[c], red);
    if (t == 0) g_trans[3 * b + c] = sum;
  }
}

}  // namespace

extern "C" {

size_t harp_lbs_mano_ws_floats(int B) {
  // pose_map 135 | A 192 | j16 48 | Jrest 48 | Rloc 144 | G 192 | g_vp 2334 | M 9336 | g_A 192 | g_pm 135 | g_j16 48 | v_posed 2334
  return (size_t)B * (135 + 192 + 48 + 48 + 144 + 192 + 2334 + 9336 + 192 + 135 + 48 + 2334);
}

int harp_lbs_mano_fwd(const harp_mano_model* m, const float* pose, const float* betas, const float* trans, int B, float* ws,
                      float* verts, float* joints, hipStream_t stream) {
  if (!m || !pose || !betas || !trans || !ws || !verts || !joints || B <= 0) return HARP_ERR_ARG;
  const LbsWs w = lbs_ws(ws, B);
  hipLaunchKernelGGL(lbs_joints_kernel, dim3(B), dim3(64), 0, stream, *m, pose, betas, w.pm, w.A, w.j16, w.Jrest, w.Rloc, w.G);
  hipLaunchKernelGGL(lbs_skin_kernel<false>, dim3((NV + kSkinVerts - 1) / kSkinVerts, (B + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK), dim3(256), 0, stream,
                     *m, betas, trans, w.pm, w.A, B, verts, nullptr, nullptr, nullptr, w.vposed);
  hipLaunchKernelGGL(lbs_joints_out_kernel, dim3((B * 63 + 255) / 256), dim3(256), 0, stream, w.j16, verts, trans, B, joints);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// g_verts (B,778,3) is MODIFIED (tip-joint gradients are folded in). Outputs: g_pose (B,48), g_betas (B,10), g_trans (B,3).
int harp_lbs_mano_bwd(const harp_mano_model* m, const float* pose, const float* betas, const float* trans, int B, float* ws,
                      float* g_verts, const float* g_joints, float* g_pose, float* g_betas, float* g_trans, hipStream_t stream) {
  if (!m || !pose || !betas || !ws || !g_verts || !g_joints || !g_pose || !g_betas || !g_trans) return HARP_ERR_ARG;
  const LbsWs w = lbs_ws(ws, B);
  hipLaunchKernelGGL(lbs_joints_bwd_kernel, dim3(B), dim3(256), 0, stream, g_joints, B, w.g_j16, g_verts, g_trans, w.g_A, B * (192 + 135),
                     g_betas, B * NB);
  hipLaunchKernelGGL(lbs_skin_kernel<true>, dim3((NV + kSkinVerts - 1) / kSkinVerts, (B + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK), dim3(256), 0, stream,
                     *m, betas, trans, w.pm, w.A, B, nullptr, g_verts, w.g_vp, w.Mo, w.vposed);
  hipLaunchKernelGGL(lbs_gA_gpm_kernel, dim3(B, kChunksA + kChunksP), dim3(192), 0, stream, *m, m->weights, w.Mo, w.g_A, w.g_vp, w.g_pm, g_betas);
  hipLaunchKernelGGL(lbs_chain_bwd_kernel<false>, dim3(B), dim3(64), 0, stream, *m, pose, w.Rloc, w.G, w.Jrest, w.g_A, w.g_pm, w.g_j16,
                     g_pose, g_betas, harp_frame_tables{}, nullptr);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"

// second and third launch of harp_hand_back_bwd (csrc/hand_back.hip): the two reductions over the vertices (spread over 72 workgroups per
// frame: throughput work), then the kinematic chain + Rodrigues backward with the scatter into the parameter tables' gradient rows
int harp_detail_lbs_back_tail(const harp_mano_model& m, const float* pose, int B, float* ws, float* g_betas, const harp_frame_tables& tables,
                              const int32_t* fid, hipStream_t stream) {
  const LbsWs w = lbs_ws(ws, B);
  hipLaunchKernelGGL(lbs_gA_gpm_kernel, dim3(B, kChunksA + kChunksP), dim3(192), 0, stream, m, m.weights, w.Mo, w.g_A, w.g_vp, w.g_pm, g_betas);
  hipLaunchKernelGGL(lbs_chain_bwd_kernel<true>, dim3(B), dim3(64), 0, stream, m, pose, w.Rloc, w.G, w.Jrest, w.g_A, w.g_pm, w.g_j16,
                     nullptr, g_betas, tables, fid);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}
