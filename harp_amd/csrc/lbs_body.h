// MANO hand-layer arithmetic shared by the stand-alone LBS kernels (lbs.hip) and the fused per-frame front / back kernels
// (hand_front.hip): Rodrigues via quaternion forward / backward (rodrigues_layer.py:43-54, :15-40), the kinematic parents
// (manolayer.py:209-239), the finger-tip vertices and the joint re-ordering (:270, :279).
#pragma once
#include "harp_common.h"

namespace lb {

constexpr int NJ = 16;
constexpr int NV = 778;
constexpr int NB = 10;
constexpr int NP = 135;

__device__ __forceinline__ int parent_of(int j) { return (j % 3 == 1) ? 0 : j - 1; }   // manolayer.py:209-239

// full_pose joint j axis-angle -> R (row-major 9)   [rodrigues_layer.py:43-54 + quat2mat :15-40]
__device__ __forceinline__ void rodrigues_fwd(const float aa[3], float R[9]) {
  const float e0 = aa[0] + 1e-8f, e1 = aa[1] + 1e-8f, e2 = aa[2] + 1e-8f;
  const float n = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float ax = aa[0] / n, ay = aa[1] / n, az = aa[2] / n;
  const float h = n * 0.5f, c = cosf(h), s = sinf(h);
  float q0 = c, q1 = s * ax, q2 = s * ay, q3 = s * az;
  const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const float w = q0 / qn, x = q1 / qn, y = q2 / qn, z = q3 / qn;
  const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
  const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy;    R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy;    R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
}

__device__ __forceinline__ void rodrigues_bwd(const float aa[3], const float g[9], float gaa[3]) {
  const float e0 = aa[0] + 1e-8f, e1 = aa[1] + 1e-8f, e2 = aa[2] + 1e-8f;
  const float n = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float a[3] = {aa[0] / n, aa[1] / n, aa[2] / n};
  const float h = n * 0.5f, c = cosf(h), s = sinf(h);
  const float q[4] = {c, s * a[0], s * a[1], s * a[2]};
  const float qn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] / qn, x = q[1] / qn, y = q[2] / qn, z = q[3] / qn;
  float gn4[4];
  gn4[0] = 2 * w * (g[0] + g[4] + g[8]) + 2 * (-z * g[1] + y * g[2] + z * g[3] - x * g[5] - y * g[6] + x * g[7]);
  gn4[1] = 2 * x * (g[0] - g[4] - g[8]) + 2 * (y * g[1] + z * g[2] + y * g[3] - w * g[5] + z * g[6] + w * g[7]);
  gn4[2] = 2 * y * (-g[0] + g[4] - g[8]) + 2 * (x * g[1] + w * g[2] + x * g[3] + z * g[5] - w * g[6] + z * g[7]);
  gn4[3] = 2 * z * (-g[0] - g[4] + g[8]) + 2 * (-w * g[1] + x * g[2] + w * g[3] + y * g[5] + x * g[6] + y * g[7]);
  const float nq[4] = {w, x, y, z};
  const float d = nq[0] * gn4[0] + nq[1] * gn4[1] + nq[2] * gn4[2] + nq[3] * gn4[3];
  float gq[4];
  for (int k = 0; k < 4; ++k) gq[k] = (gn4[k] - nq[k] * d) / qn;
  const float g_h = -s * gq[0] + c * (a[0] * gq[1] + a[1] * gq[2] + a[2] * gq[3]);
  const float ga[3] = {s * gq[1], s * gq[2], s * gq[3]};
  const float g_n = -(a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2]) / n + 0.5f * g_h;
  gaa[0] = ga[0] / n + g_n * e0 / n;
  gaa[1] = ga[1] / n + g_n * e1 / n;
  gaa[2] = ga[2] / n + g_n * e2 / n;
}


static __constant__ int c_tips[5] = {745, 317, 444, 556, 673};                                                  // manolayer.py:270
static __constant__ int c_reorder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};  // :279

// workspace of harp_lbs_mano_fwd/bwd (harp_lbs_mano_ws_floats):
// pose_map 135 | A 192 | j16 48 | Jrest 48 | Rloc 144 | G 192 | g_vp 2334 | M 9336 | g_A 192 | g_pm 135 | g_j16 48 | v_posed 2334  (x B)
struct LbsWs { float *pm, *A, *j16, *Jrest, *Rloc, *G, *g_vp, *Mo, *g_A, *g_pm, *g_j16, *vposed; };
__host__ __device__ inline LbsWs lbs_ws(float* ws, int B) {
  LbsWs w; float* p = ws;
  w.pm = p; p += (size_t)B * 135; w.A = p; p += (size_t)B * 192; w.j16 = p; p += (size_t)B * 48; w.Jrest = p; p += (size_t)B * 48;
  w.Rloc = p; p += (size_t)B * 144; w.G = p; p += (size_t)B * 192; w.g_vp = p; p += (size_t)B * 2334; w.Mo = p; p += (size_t)B * 9336;
  w.g_A = p; p += (size_t)B * 192; w.g_pm = p; p += (size_t)B * 135; w.g_j16 = p; p += (size_t)B * 48;
  w.vposed = p;      // posed vertices of the forward pass, read by the backward skinning kernel
  return w;
}


}  // namespace lb
