// Fused per-pixel shader + its backward for gfx950.  Replaces, for the K=1 RGB passes of HARP, the ~40 torch /
// PyTorch3D ops of SoftPhongShaderShadow / SoftPhongShaderPBR:
//   interpolate_face_attributes x3 (positions, normals, uvs)       renderer_helper.py:173-178, 364, 498-503
//   TexturesUV.sample_textures x2 (albedo, normal map; grid_sample) renderer_helper.py:127, 572; pbr_materials.py:110
//   PBRMaterials.apply_normal_map / compute_tangent                 pbr_materials.py:58-124
//   _apply_lighting (PointLights diffuse, shininess-0 specular)     renderer_helper.py:186, 513-515
//   shadow-map test (3x3 taps of sigmoid(1000 (z_light - z_hit + 0.008)))  renderer_helper.py:360-408
//   colour composition + softmax_rgb_blend (K=1)                    renderer_helper.py:188, 517-518, 589-591
// One lane = one pixel; fragments (bary, position, normal, uv, texels) live only in registers: the barycentrics are
// recomputed from the 64-B face record of the hit face instead of being stored as (B,S,S,K,3) tensors.
// Backward recomputes the forward per pixel and scatters with float atomics (HW global_atomic_add_f32).
#include "shade_common.h"

namespace {

#ifndef SHADE_TEX_SLOTS
#define SHADE_TEX_SLOTS 512
#endif
#ifndef SHADE_VTX_SLOTS
#define SHADE_VTX_SLOTS 256
#endif
#ifndef SHADE_ACC_T
#define SHADE_ACC_T double
#endif
#ifndef SHADE_OCC
#define SHADE_OCC 3
#endif
constexpr int kTexSlots = SHADE_TEX_SLOTS, kVtxSlots = SHADE_VTX_SLOTS;
constexpr int kZlW = 32, kZlH = 24;   // LDS window of the shadow-map gradient (light pixels)
constexpr int kTexRows = 512;   // row buckets of the sorted texel flush (a tile touching a taller span of texel rows flushes unsorted)
#ifndef SHADE_PRERED
#define SHADE_PRERED 19      // xor distances of the lane-merge butterfly before the vertex LDS atomics (1 | 2 | 16)
#endif
// Counting sort of the occupied slots of an LDS hash table by key row (key / width): returns the number of occupied slots and
// fills s_order[0..n) with slot indices in row order, or -1 when the rows span more than kTexRows (caller flushes unsorted).
// Called by all 256 threads of the workgroup; ylo/yhi = the row range this lane touched (0x7fffffff / -1 if none).
template <int SLOTS>
__device__ __forceinline__ int sort_slots_by_row(const int* __restrict__ keys, int width, int ylo, int yhi, int* s_box, int* s_hist,
                                                 int* s_order, int* s_wsum) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                        // previous users of the scratch arrays are done
  if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = -1; }
  for (int i = threadIdx.x; i < kTexRows; i += 256) s_hist[i] = 0;
  __syncthreads();
#pragma unroll
  for (int o2 = 32; o2 >= 1; o2 >>= 1) { ylo = min(ylo, __shfl_xor(ylo, o2)); yhi = max(yhi, __shfl_xor(yhi, o2)); }
  if (lane == 0 && yhi >= 0) { atomicMin(&s_box[0], ylo); atomicMax(&s_box[1], yhi); }
  __syncthreads();
  const int ymin = s_box[0], bh = s_box[1] - s_box[0] + 1;
  if (s_box[1] < 0 || bh > kTexRows) return -1;
  constexpr int SPT = (SLOTS + 255) / 256;
  int rank[SPT], row[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int key = i < SLOTS ? keys[i] : -1;
    row[j] = key >= 0 ? key / width - ymin : -1;
    if (row[j] >= kTexRows) row[j] = -1;    // cannot happen (keys lie inside the lanes' row range); keeps the scatter in bounds
    rank[j] = row[j] >= 0 ? atomicAdd(&s_hist[row[j]], 1) : 0;
  }
  __syncthreads();
  constexpr int RPT = kTexRows / 256;
  int cnt[RPT], sum = 0;
#pragma unroll
  for (int r = 0; r < RPT; ++r) { cnt[r] = s_hist[threadIdx.x * RPT + r]; sum += cnt[r]; }
  int incl = sum;
#pragma unroll
  for (int o2 = 1; o2 < 64; o2 <<= 1) { const int up = __shfl_up(incl, o2); if (lane >= o2) incl += up; }
  if (lane == 63) s_wsum[w] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int i = 0; i < w; ++i) base += s_wsum[i];
  const int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
#pragma unroll
  for (int r = 0; r < RPT; ++r) { s_hist[threadIdx.x * RPT + r] = base; base += cnt[r]; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if (row[j] >= 0) s_order[s_hist[row[j]] + rank[j]] = threadIdx.x + 256 * j;
  __syncthreads();
  return total;
}

template <bool BWD>
__global__ void __launch_bounds__(256, SHADE_OCC) shade_kernel(const harp_shade_args A, const int32_t* __restrict__ order,
                                                               const int32_t* __restrict__ nact, int nsx) {
  __shared__ float s_red[32];
  // per-vertex accumulators: 0-2 g_verts, 3-5 g_vnormals, 6-8 g_ndc
  __shared__ VertexAccum<BWD ? kVtxSlots : 1, 9, SHADE_ACC_T> s_acc;
  // per-texel accumulators (key = texel index): 0-2 albedo gradient, 3-5 normal-map gradient. Neighbouring pixels share
  // bilinear corners (~1.4 px per texel), and same-line float atomics serialise in L2: pre-summing in LDS cuts the global
  // atomics ~4x and removes the contention (ablation: the two texture scatters were 1.75 of 2.6 ms).
  __shared__ VertexAccum<BWD ? kTexSlots : 1, 6, SHADE_ACC_T> s_tex;
  __shared__ double s_zwin[BWD ? kZlW * kZlH : 1];   // light-view depth-map gradient window (see below)
  __shared__ int s_zbox[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int S = A.S, V = A.V;
  // one 16x16 tile per workgroup, dispatched in the rasteriser's heaviest-first super-tile order (harp_common.h: tile_decode): the
  // tiles with covered pixels run first and densely instead of interleaved with the ~80 % background tiles.
  int b, st_unused, tx0, ty0, tsub;
  // backward pass with l1_target set and g_rgb == NULL = FUSED-LOSS mode: no forward launch at all — the backward pass recomputes the
  // colour anyway, so it also forms the photometric L1 (value and gradient) itself; the forward kernel's only remaining products in a
  // fitting step were that loss and a gradient image the backward pass read back.
  const bool fused_loss = BWD && A.l1_target != nullptr && A.g_rgb == nullptr;
  const int kind = tile_decode(order, nact, A.B, nsx, S, b, st_unused, tx0, ty0, tsub, !BWD || fused_loss);
  if (kind == 0) return;
  if (kind == 2) {
    // super-tile without a single face (3/4 of the launch): no gradient in the backward pass; in the forward pass its first workgroup
    // writes the background colour for all 64x64 pixels and the other 15 leave at once
    if ((BWD && !fused_loss) || tsub != 0 || (!A.rgb && !A.l1_target)) return;
    if ((!A.rgb || BWD) && A.l1_bg_sums) {
      // loss-only mode against static targets: the background term of this super-tile is a constant, looked up
      if (threadIdx.x == 0) {
        const float sum = A.l1_bg_sums[(size_t)A.l1_fid[b] * nsx * nsx + st_unused];
        if (sum != 0.f) atomicAdd(A.l1_loss, sum * A.l1_inv);
      }
      return;
    }
    // background colour everywhere; the fused photometric term still counts these pixels where the mask is set (|bg - y| m), they
    // only have no gradient
    float acc = 0.f;
    float mk_[16];
    const size_t tbase = A.l1_target ? (size_t)A.l1_fid[b] * S * S : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int xi, yi;
      supertile_pixel(k, tx0, ty0, xi, yi);
      mk_[k] = (A.l1_target && xi < S && yi < S) ? (A.l1_mask ? A.l1_mask[tbase + (size_t)yi * S + xi] : 1.f) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int xi, yi;
      supertile_pixel(k, tx0, ty0, xi, yi);
      if (xi < S && yi < S) {
        if (!BWD && A.rgb) { float* r = A.rgb + (((size_t)b * S + yi) * S + xi) * 3; r[0] = A.bg[0]; r[1] = A.bg[1]; r[2] = A.bg[2]; }
        const float m = mk_[k];
        if (m != 0.f) {
          const float* t = A.l1_target + (tbase + (size_t)yi * S + xi) * 3;
          acc += fabsf(A.bg[0] * m - t[0] * m) + fabsf(A.bg[1] * m - t[1] * m) + fabsf(A.bg[2] * m - t[2] * m);
        }
      }
    }
    if (A.l1_target) {
      const float sum = block_sum_256(acc, s_red);
      if (threadIdx.x == 0 && sum != 0.f) atomicAdd(A.l1_loss, sum * A.l1_inv);
    }
    return;
  }
  constexpr int R = 1;
  if (BWD) {
    bool any_act = false;
    float bg_loss = 0.f;
#pragma unroll
    for (int sub = 0; sub < R * R; ++sub) {
      const int xi = tx0 + (lane & 15), yi = ty0 + w * 4 + (lane >> 4);
      if (xi < S && yi < S) {
        const size_t o = ((size_t)b * S + yi) * S + xi;
        const bool hit = A.face_id[o] >= 0;
        if (fused_loss) {
          const size_t to = ((size_t)A.l1_fid[b] * S + yi) * S + xi;
          const float m = A.l1_mask ? A.l1_mask[to] : 1.f;
          any_act |= hit && (m != 0.f);
          if (!hit && m != 0.f) {        // uncovered pixel inside the mask: background colour against the target, no gradient
            const float* t = A.l1_target + to * 3;
            bg_loss += fabsf(A.bg[0] * m - t[0] * m) + fabsf(A.bg[1] * m - t[1] * m) + fabsf(A.bg[2] * m - t[2] * m);
          }
        } else if (hit) {
          const V3 gq = ld(A.g_rgb + o * 3);
          any_act |= (gq.x != 0.f || gq.y != 0.f || gq.z != 0.f);
        }
      }
    }
    if (fused_loss) {
      const float sum = block_sum_256(bg_loss, s_red);
      if (threadIdx.x == 0 && sum != 0.f) atomicAdd(A.l1_loss, sum * A.l1_inv);
    }
    __syncthreads();
    if (threadIdx.x < 32) s_red[threadIdx.x] = 0.f;
    if (__syncthreads_or(any_act ? 1 : 0) == 0) return;
    s_acc.clear();
    s_tex.clear();
    __syncthreads();
  }
  float racc[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) racc[k] = 0.f;
  int tb_y0 = 0x7fffffff, tb_y1 = -1;   // backward: range of texel rows this lane touched

  for (int sub = 0; sub < R * R; ++sub) {
  const int xi = tx0 + (lane & 15), yi = ty0 + w * 4 + (lane >> 4);
  const bool in_img = xi < S && yi < S;
  const size_t o = ((size_t)b * S + (in_img ? yi : 0)) * S + (in_img ? xi : 0);
  const int f = in_img ? A.face_id[o] : -1;
  V3 gc = mk(0.f, 0.f, 0.f);
  bool act = f >= 0;
  float l1_m = 0.f;
  size_t l1_to = 0;
  if (BWD && fused_loss) {
    if (act) {
      l1_to = ((size_t)A.l1_fid[b] * S + yi) * S + xi;
      l1_m = A.l1_mask ? A.l1_mask[l1_to] : 1.f;
    }
    act = act && (l1_m != 0.f);
  } else if (BWD) {
    if (act) gc = ld(A.g_rgb + o * 3);
    act = act && (gc.x != 0.f || gc.y != 0.f || gc.z != 0.f);
  } else if (!act) {
    if (in_img && A.rgb) { float* r = A.rgb + o * 3; r[0] = A.bg[0]; r[1] = A.bg[1]; r[2] = A.bg[2]; }
  }
  float out_rgb[3] = {A.bg[0], A.bg[1], A.bg[2]};
  float vsc[BWD ? 27 : 1];   // backward: gradient of the face's 3 vertices x (position, normal, ndc), scattered after the branch
  int vidx[3] = {0, 0, 0};
  float zd[BWD ? 9 : 1];     // backward: gradient of the 3x3 shadow-map taps (row-major), scattered after the branch
  int zix = -0x40000000, ziy = -0x40000000;   // light-view pixel of this lane's hit point (unclamped); sentinel = no taps
  if (BWD) {
#pragma unroll
    for (int c = 0; c < 27; ++c) vsc[BWD ? c : 0] = 0.f;
#pragma unroll
    for (int c = 0; c < 9; ++c) zd[BWD ? c : 0] = 0.f;
  }
  if (act) {
    const float px = pix_to_ndc(xi, S), py = pix_to_ndc(yi, S);
    const float* col = A.colors;               // amb(3) diff(3) spec(3)
    Frag g;
    g.t = load_tri(((const FaceRec*)A.recs) + (size_t)b * A.F + f);
    g.br = bary_fwd(g.t, px, py);
    const float b0 = g.br.b0, b1 = g.br.b1, b2 = g.br.b2;
    g.i0 = A.faces[3 * f]; g.i1 = A.faces[3 * f + 1]; g.i2 = A.faces[3 * f + 2];
    g.u0 = A.faces_uvs[3 * f]; g.u1 = A.faces_uvs[3 * f + 1]; g.u2 = A.faces_uvs[3 * f + 2];
    const float* vb = A.verts + (size_t)b * V * 3;
    const float* nb = A.vnormals + (size_t)b * V * 3;
    const V3 v0 = ld(vb + 3 * g.i0), v1 = ld(vb + 3 * g.i1), v2 = ld(vb + 3 * g.i2);
    const V3 n0 = ld(nb + 3 * g.i0), n1 = ld(nb + 3 * g.i1), n2 = ld(nb + 3 * g.i2);
    const float uv0x = A.verts_uvs[2 * g.u0], uv0y = A.verts_uvs[2 * g.u0 + 1];
    const float uv1x = A.verts_uvs[2 * g.u1], uv1y = A.verts_uvs[2 * g.u1 + 1];
    const float uv2x = A.verts_uvs[2 * g.u2], uv2y = A.verts_uvs[2 * g.u2 + 1];
    g.p = v0 * b0 + v1 * b1 + v2 * b2;
    g.n = n0 * b0 + n1 * b1 + n2 * b2;
    g.u = uv0x * b0 + uv1x * b1 + uv2x * b2;
    g.v = uv0y * b0 + uv1y * b1 + uv2y * b2;
    g.bs = bil_setup(g.u, g.v, A.Wt, A.Ht);
    V3 tdx, tdy, mdx, mdy;
    const bool packed = A.texnm != nullptr && A.nmap != nullptr;
    if (packed) bil_sample2((const float4*)A.texnm, g.bs, A.Wt, A.Ht, g.texel, g.m, BWD ? &tdx : nullptr, &tdy, BWD ? &mdx : nullptr, &mdy);
    else g.texel = bil_sample(A.tex, g.bs, A.Wt, A.Ht, BWD ? &tdx : nullptr, &tdy);
    // normal map (pbr_materials.py:58-124): n' = normalize(-u m.x - v m.y + n m.z)
    V3 nfin = g.n;
    if (A.nmap) {
      if (!packed) g.m = bil_sample(A.nmap, g.bs, A.Wt, A.Ht, BWD ? &mdx : nullptr, &mdy);
      g.s = (g.n.z >= 0.f) ? 1.f : -1.f;
      g.a = -rcp(g.s + g.n.z);
      const float bb = g.n.x * g.n.y * g.a;
      g.tu = mk(1.f + g.s * g.n.x * g.n.x * g.a, g.s * bb, -g.s * g.n.x);
      g.tv = mk(bb, g.s + g.n.y * g.n.y * g.a, -g.n.y);
      g.nprime = g.tu * (-g.m.x) + g.tv * (-g.m.y) + g.n * g.m.z;
      g.lnp = fsqrt(dot(g.nprime, g.nprime));
      g.nhat = g.nprime * rcp(fmaxf(g.lnp, 1e-12f));
      nfin = g.nhat;
    }
    // PointLights.diffuse: normalize(n, eps 1e-6) . normalize(L - p, eps 1e-6)
    g.lnh = fsqrt(dot(nfin, nfin));
    g.nn = nfin * rcp(fmaxf(g.lnh, 1e-6f));
    g.ldir = ld(A.light_pos + 3 * b) - g.p;
    g.llen = fsqrt(dot(g.ldir, g.ldir));
    g.lhat = g.ldir * rcp(fmaxf(g.llen, 1e-6f));
    g.cosr = dot(g.nn, g.lhat);
    const float cosang = fmaxf(g.cosr, 0.f);
    // shadow (renderer_helper.py:379-408)
    g.vis = 1.f;
    float sg[9];
    int tapo[9];
    const float half = 0.5f * (float)S;
    if (A.zl) {
      const float* R = A.light_R + 9 * b;
      const float* T = A.light_T + 3 * b;
      g.q = mk(g.p.x * R[0] + g.p.y * R[3] + g.p.z * R[6] + T[0], g.p.x * R[1] + g.p.y * R[4] + g.p.z * R[7] + T[1],
               g.p.x * R[2] + g.p.y * R[5] + g.p.z * R[8] + T[2]);
      const float rqz = rcp(g.q.z), rhalf = rcp(half);
      const float xn = (A.focal * g.q.x * rqz - A.ppx + half) * rhalf, yn = (A.focal * g.q.y * rqz - A.ppy + half) * rhalf;
      const float xs = half - half * xn, ys = half - half * yn;
      // torch .round().long(): half-to-even; non-finite -> clamp below makes it harmless
      g.ix = (int)rintf(fminf(fmaxf(xs, -1.0e6f), 1.0e6f));
      g.iy = (int)rintf(fminf(fmaxf(ys, -1.0e6f), 1.0e6f));
      const float aa = g.q.z - 0.008f;
      float acc = 0.f;
      int k = 0;
      for (int ii = -1; ii <= 1; ++ii)
        for (int jj = -1; jj <= 1; ++jj, ++k) {
          const int yy = min(max(g.iy + ii, 0), S - 1), xx = min(max(g.ix + jj, 0), S - 1);
          tapo[k] = yy * S + xx;
          sg[k] = sigmoidf((A.zl[(size_t)b * S * S + tapo[k]] - aa) * 1000.0f);
          acc += sg[k];
        }
      g.vis = acc * (1.0f / 9.0f);
    }
    const V3 amb = ld(col), dfc = ld(col + 3), spc = ld(col + 6);
    const V3 lightc = mk(amb.x + dfc.x * cosang * g.vis, amb.y + dfc.y * cosang * g.vis, amb.z + dfc.z * cosang * g.vis);
    const V3 c = mk(lightc.x * g.texel.x + spc.x, lightc.y * g.texel.y + spc.y, lightc.z * g.texel.z + spc.z);
    // softmax_rgb_blend, K=1, blur=0 (Appendix A.4); prob in (0.5,1] is taken as 1 (error <= 2e-10)
    const float zpix = b0 * g.t.z0 + b1 * g.t.z1 + b2 * g.t.z2;
    const float zinv = (100.0f - zpix) * (1.0f / 99.0f);
    const float zmax = fmaxf(zinv, 1e-10f);
    const float wnum = __expf((zinv - zmax) * 1e4f);
    const float delta = fmaxf(__expf((1e-10f - zmax) * 1e4f), 1e-10f);
    const float denom = wnum + delta;
    if (!BWD) {
      const float rden = rcp(denom);
      out_rgb[0] = (wnum * c.x + delta * A.bg[0]) * rden;
      out_rgb[1] = (wnum * c.y + delta * A.bg[1]) * rden;
      out_rgb[2] = (wnum * c.z + delta * A.bg[2]) * rden;
      if (A.rgb) { float* r = A.rgb + o * 3; r[0] = out_rgb[0]; r[1] = out_rgb[1]; r[2] = out_rgb[2]; }
    } else {
      const float wk = wnum * rcp(denom);
      if (fused_loss) {
        // the forward colour of this pixel, the L1 against the target and its gradient (same expressions as the forward kernel)
        const float rden = rcp(denom);
        const float o3[3] = {(wnum * c.x + delta * A.bg[0]) * rden, (wnum * c.y + delta * A.bg[1]) * rden, (wnum * c.z + delta * A.bg[2]) * rden};
        const float wl = A.l1_w[0] * A.l1_inv * l1_m;
        float gq[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float d = o3[ch] * l1_m - A.l1_target[l1_to * 3 + ch] * l1_m;
          racc[16] += fabsf(d);
          gq[ch] = wl * (float)((d > 0.f) - (d < 0.f));
        }
        gc = mk(gq[0], gq[1], gq[2]);
      }
      const V3 g_c = gc * wk;
      // c = lightc * texel + spec
      const V3 g_tex = mk(g_c.x * lightc.x, g_c.y * lightc.y, g_c.z * lightc.z);
      const V3 g_lc = mk(g_c.x * g.texel.x, g_c.y * g.texel.y, g_c.z * g.texel.z);
      racc[0] += g_lc.x; racc[1] += g_lc.y; racc[2] += g_lc.z;                               // amb
      racc[3] += g_lc.x * cosang * g.vis; racc[4] += g_lc.y * cosang * g.vis; racc[5] += g_lc.z * cosang * g.vis;  // diff
      racc[6] += g_c.x; racc[7] += g_c.y; racc[8] += g_c.z;                                  // spec
      const float g_dv = g_lc.x * dfc.x + g_lc.y * dfc.y + g_lc.z * dfc.z;               // d/d(cosang*vis)
      const float g_vis = g_dv * cosang;
      const float g_cos = (g.cosr > 0.f) ? g_dv * g.vis : 0.f;
      float gu = 0.f, gv = 0.f;                     // d/d(u,v)
      V3 g_m_keep = mk(0.f, 0.f, 0.f);
      gu += dot(g_tex, tdx) * (float)(A.Wt - 1);
      gv += dot(g_tex, tdy) * -(float)(A.Ht - 1);
      V3 g_p = mk(0.f, 0.f, 0.f);
      // cos = nn . lhat
      const V3 g_nn = g.lhat * g_cos, g_lhat = g.nn * g_cos;
      V3 g_ldir = (g.llen > 1e-6f) ? (g_lhat - g.lhat * dot(g.lhat, g_lhat)) * rcp(g.llen) : g_lhat * 1e6f;
      racc[9] += g_ldir.x; racc[10] += g_ldir.y; racc[11] += g_ldir.z;                       // light_pos
      g_p = g_p - g_ldir;
      V3 g_nfin = (g.lnh > 1e-6f) ? (g_nn - g.nn * dot(g.nn, g_nn)) * rcp(g.lnh) : g_nn * 1e6f;
      V3 g_n = g_nfin;
      if (A.nmap) {
        const V3 g_np = (g.lnp > 1e-12f) ? (g_nfin - g.nhat * dot(g.nhat, g_nfin)) * rcp(g.lnp) : g_nfin * 1e12f;
        const V3 g_m = mk(-dot(g.tu, g_np), -dot(g.tv, g_np), dot(g.n, g_np));
        g_m_keep = g_m;
        gu += dot(g_m, mdx) * (float)(A.Wt - 1);
        gv += dot(g_m, mdy) * -(float)(A.Ht - 1);
        const V3 g_tu = g_np * (-g.m.x), g_tv = g_np * (-g.m.y);
        g_n = g_np * g.m.z;
        const float x = g.n.x, y = g.n.y, s = g.s, a = g.a;
        // tu = (1 + s x^2 a, s b, -s x), tv = (b, s + y^2 a, -y), b = x y a, a = -1/(s+z)
        const float g_b = s * g_tu.y + g_tv.x;
        const float g_a = s * x * x * g_tu.x + y * y * g_tv.y + g_b * x * y;
        g_n.x += 2.f * s * x * a * g_tu.x - s * g_tu.z + g_b * y * a;
        g_n.y += 2.f * y * a * g_tv.y - g_tv.z + g_b * x * a;
        g_n.z += g_a * a * a;
      }
      // texture + normal-map gradient: 4 bilinear corners x 6 channels into the LDS texel table
      if (!(A.debug_skip & 1)) {
        const float ax = 1.f - g.bs.wx, ay = 1.f - g.bs.wy;
        const float cw[4] = {ax * ay, g.bs.wx * ay, ax * g.bs.wy, g.bs.wx * g.bs.wy};
        const int cx[4] = {g.bs.x0, g.bs.x0 + 1, g.bs.x0, g.bs.x0 + 1}, cy[4] = {g.bs.y0, g.bs.y0, g.bs.y0 + 1, g.bs.y0 + 1};
        const bool do_t = A.g_tex != nullptr, do_n = (A.nmap != nullptr) && (A.g_nmap != nullptr);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (cx[k] >= A.Wt || cy[k] >= A.Ht || cw[k] == 0.f) continue;
          const int key = cy[k] * A.Wt + cx[k];
          tb_y0 = min(tb_y0, cy[k]); tb_y1 = max(tb_y1, cy[k]);
          const int slot = s_tex.find(key);
          const float vals[6] = {g_tex.x * cw[k], g_tex.y * cw[k], g_tex.z * cw[k], g_m_keep.x * cw[k], g_m_keep.y * cw[k], g_m_keep.z * cw[k]};
          if (slot >= 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { if (do_t) s_tex.add(slot, c, vals[c]); if (do_n) s_tex.add(slot, 3 + c, vals[3 + c]); }
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (do_t) atomicAdd(A.g_tex + (size_t)key * 3 + c, vals[c]);
              if (do_n) atomicAdd(A.g_nmap + (size_t)key * 3 + c, vals[3 + c]);
            }
          }
        }
      }
      // shadow
      if (A.zl) {
        float g_zq = 0.f;
        for (int k = 0; k < 9; ++k) {
          const float d = g_vis * (1.0f / 9.0f) * sg[k] * (1.0f - sg[k]) * 1000.0f;
          if (d != 0.f) {
            zd[BWD ? k : 0] = d;
            g_zq -= d;
          }
        }
        zix = g.ix; ziy = g.iy;
        const float* R = A.light_R + 9 * b;
        g_p = g_p + mk(R[2], R[5], R[8]) * g_zq;
        racc[12] += g.p.x * g_zq; racc[13] += g.p.y * g_zq; racc[14] += g.p.z * g_zq;        // light_R[:,2]
        racc[15] += g_zq;                                                                   // light_T.z
      }
      // interpolation backward
      float gb0 = dot(v0, g_p) + dot(n0, g_n) + uv0x * gu + uv0y * gv;
      float gb1 = dot(v1, g_p) + dot(n1, g_n) + uv1x * gu + uv1y * gv;
      float gb2 = dot(v2, g_p) + dot(n2, g_n) + uv2x * gu + uv2y * gv;
      float gnd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      bary_bwd(g.t, px, py, g.br, gb0, gb1, gb2, gnd);
      const float bw[3] = {b0, b1, b2};
      vidx[0] = g.i0; vidx[1] = g.i1; vidx[2] = g.i2;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vsc[9 * k + 0] = g_p.x * bw[k]; vsc[9 * k + 1] = g_p.y * bw[k]; vsc[9 * k + 2] = g_p.z * bw[k];
        vsc[9 * k + 3] = g_n.x * bw[k]; vsc[9 * k + 4] = g_n.y * bw[k]; vsc[9 * k + 5] = g_n.z * bw[k];
        vsc[9 * k + 6] = gnd[3 * k]; vsc[9 * k + 7] = gnd[3 * k + 1]; vsc[9 * k + 8] = gnd[3 * k + 2];
      }
    }
  }
  if (BWD && A.zl && A.g_zl && !(A.debug_skip & 4)) {
    // light-view depth-map gradient: 9 taps per pixel, and the 3x3 windows of neighbouring pixels overlap almost completely.  The
    // taps of the tile are summed in a direct-mapped LDS window (kZlW x kZlH light pixels anchored at the tile's smallest tap
    // column / row: no hashing, ds_add_f64) and flushed row-major with one coalesced atomic per touched light pixel; taps outside
    // the window (strong magnification) go straight to memory.  (Tried before: an LDS hash table - 9 probes per pixel, slower than
    // the direct atomics; DPP hand-over between x-neighbours - exact but only -20 %, it needs equal light rows.)
    if (threadIdx.x == 0) { s_zbox[0] = 0x7fffffff; s_zbox[1] = 0x7fffffff; }
    for (int i = threadIdx.x; i < kZlW * kZlH; i += 256) s_zwin[i] = 0.0;
    __syncthreads();
    const bool has = zix > -0x40000000;
    int mx = has ? min(max(zix - 1, 0), S - 1) : 0x7fffffff, my = has ? min(max(ziy - 1, 0), S - 1) : 0x7fffffff;
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) { mx = min(mx, __shfl_xor(mx, o2)); my = min(my, __shfl_xor(my, o2)); }
    if (lane == 0 && mx != 0x7fffffff) { atomicMin(&s_zbox[0], mx); atomicMin(&s_zbox[1], my); }
    __syncthreads();
    const int x0 = s_zbox[0], y0 = s_zbox[1];
    float* gz = A.g_zl + (size_t)b * S * S;
    const int znt = (S + 15) >> 4;
    if (has) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yy = min(max(ziy + r - 1, 0), S - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = zd[BWD ? 3 * r + c : 0];
          if (d == 0.f) continue;
          const int xx = min(max(zix + c - 1, 0), S - 1);
          const int wx = xx - x0, wy = yy - y0;
          if (wx < kZlW && wy < kZlH) atomicAdd(&s_zwin[wy * kZlW + wx], (double)d);
          else {
            atomicAdd(gz + yy * S + xx, d);
            if (A.g_zl_tiles) A.g_zl_tiles[((size_t)b * znt + (yy >> 4)) * znt + (xx >> 4)] = 1;
          }
        }
      }
    }
    __syncthreads();
    if (x0 != 0x7fffffff) {
      for (int i = threadIdx.x; i < kZlW * kZlH; i += 256) {
        const double v = s_zwin[i];
        if (v != 0.0) {
          atomicAdd(gz + (size_t)(y0 + i / kZlW) * S + x0 + (i % kZlW), (float)v);
          if (A.g_zl_tiles) A.g_zl_tiles[((size_t)b * znt + ((y0 + i / kZlW) >> 4)) * znt + ((x0 + i % kZlW) >> 4)] = 1;
        }
      }
    }
  }
  if (BWD && !(A.debug_skip & 8)) {
    // vertex gradients: lanes are 16x4 pixels and a face covers ~16 of them, so neighbouring lanes mostly add to the SAME three
    // vertices and same-address LDS atomics serialise.  Merge lanes that hit the same face first (butterfly over SHADE_PRERED's
    // xor distances: 1, 2 = x neighbours through DPP, 16 = the row below through ds_bpermute), then only the surviving lanes add.
    bool alive = act;
    merge_same_face<27, SHADE_PRERED>(vsc, act ? f : -1, alive, lane);
    if (alive) {
      float* gvb = A.g_verts + (size_t)b * V * 3;
      float* gnb = A.g_vnormals + (size_t)b * V * 3;
      float* gdb = A.g_ndc + (size_t)b * V * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int slot = s_acc.find(vidx[k]);
        if (slot >= 0) {
#pragma unroll
          for (int c = 0; c < 9; ++c) s_acc.add(slot, c, vsc[9 * k + c]);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            atomicAdd(gvb + 3 * vidx[k] + c, vsc[9 * k + c]); atomicAdd(gnb + 3 * vidx[k] + c, vsc[9 * k + 3 + c]);
            atomicAdd(gdb + 3 * vidx[k] + c, vsc[9 * k + 6 + c]);
          }
        }
      }
    }
  }
  if (!BWD && A.l1_target) {
    // fused photometric term: torch.nn.L1Loss(y_true * m, y_pred * m) (optimize_sequence.py:543) and its gradient w.r.t. y_pred
    float acc = 0.f;
    if (in_img) {
      const size_t to = ((size_t)A.l1_fid[b] * S + yi) * S + xi;
      const float m = A.l1_mask ? A.l1_mask[to] : 1.f;
      const float wk = A.l1_w[0] * A.l1_inv * m;
      // a masked-out pixel (the eroded silhouette is 0 on ~80 % of the image) contributes exactly 0: its target is not even read;
      // the gradient image is only consumed where a face was hit (harp_shade_bwd), so it is only written there
      if (m != 0.f) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float d = out_rgb[ch] * m - A.l1_target[to * 3 + ch] * m;
          acc += fabsf(d);
          if (f >= 0) A.l1_grad[o * 3 + ch] = wk * ((d > 0.f) - (d < 0.f));
        }
      } else if (f >= 0) {
        A.l1_grad[o * 3] = 0.f; A.l1_grad[o * 3 + 1] = 0.f; A.l1_grad[o * 3 + 2] = 0.f;
      }
    }
    const float sum = block_sum_256(acc, s_red);
    if (threadIdx.x == 0 && sum != 0.f) atomicAdd(A.l1_loss, sum * A.l1_inv);
  }
  }   // sub-tile loop
  if (BWD && fused_loss) {
    const float sum = block_sum_256(racc[16], s_red + 16);      // (s_red[0..15] are the scalar-gradient accumulators below)
    if (threadIdx.x == 0 && sum != 0.f) atomicAdd(A.l1_loss, sum * A.l1_inv);
  }
  if (BWD) {
    // block-level reduction of the 16 per-frame / global scalars, then one atomic each
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float s = wave_sum(racc[k]);
      if (lane == 0 && s != 0.f) atomicAdd(&s_red[k], s);
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      const float s = s_red[threadIdx.x];
      const int k = threadIdx.x;
      if (s != 0.f) {
        if (k < 9) { if (A.g_colors) atomicAdd(A.g_colors + k, s); }
        else if (k < 12) { if (A.g_light_pos) atomicAdd(A.g_light_pos + 3 * b + (k - 9), s); }
        else if (k < 15) { if (A.g_light_R) atomicAdd(A.g_light_R + 9 * b + 3 * (k - 12) + 2, s); }
        else if (A.g_light_T) atomicAdd(A.g_light_T + 3 * b + 2, s);
      }
    }
    // flush the LDS accumulators: one global float atomic per (key, component) per workgroup.  (A variant that appended the
    // entries to per-bucket lists and summed them in a second kernel with plain stores was measured SLOWER: the memory-side
    // atomics are fire-and-forget, the cost of this kernel was the LDS side — see DESIGN.md.)
    float* gvb = A.g_verts + (size_t)b * V * 3;
    float* gnb = A.g_vnormals + (size_t)b * V * 3;
    float* gdb = A.g_ndc + (size_t)b * V * 3;
    for (int i = threadIdx.x; i < kVtxSlots; i += 256) {
      const int v = s_acc.key[i];
      if (v < 0 || (A.debug_skip & 32)) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float a0 = (float)s_acc.val[i][c], a1 = (float)s_acc.val[i][3 + c], a2 = (float)s_acc.val[i][6 + c];
        if (a0 != 0.f) atomicAdd(gvb + 3 * v + c, a0);
        if (a1 != 0.f) atomicAdd(gnb + 3 * v + c, a1);
        if (a2 != 0.f) atomicAdd(gdb + 3 * v + c, a2);
      }
    }
    // texels: random-address float atomics cost one memory request per lane (21 G/s measured) while lanes on
    // consecutive addresses share requests (330 G/s).  The occupied slots are therefore counting-sorted by row (LDS histogram +
    // scan; the rows of one screen tile hold <= ~25 neighbouring texels, also across a chart seam) and flushed in that order.
    __shared__ int s_box[2];
    __shared__ int s_wsum[4];
    // the sort scratch reuses the (already flushed) shadow-gradient window: keeps the kernel at 3 workgroups per CU
    static_assert(sizeof(double) * kZlW * kZlH >= sizeof(int) * (kTexRows + kTexSlots), "window too small for the sort scratch");
    int* s_hist = (int*)s_zwin;
    int* s_order = s_hist + kTexRows;
    if (!(A.debug_skip & 16)) {
      const int total = sort_slots_by_row<kTexSlots>(s_tex.key, A.Wt, tb_y0, tb_y1, s_box, s_hist, s_order, s_wsum);
      if (total >= 0) {
        for (int i = threadIdx.x; i < total * 3; i += 256) {      // lanes = (texel, channel): consecutive addresses within a row
          const int e = i / 3, c = i - 3 * e;
          const int slot = s_order[e];
          const int key = s_tex.key[slot];
          const float a0 = (float)s_tex.val[slot][c], a1 = (float)s_tex.val[slot][3 + c];
          if (a0 != 0.f) atomicAdd(A.g_tex + (size_t)key * 3 + c, a0);
          if (a1 != 0.f) atomicAdd(A.g_nmap + (size_t)key * 3 + c, a1);
        }
      } else {
        for (int i = threadIdx.x; i < kTexSlots; i += 256) {
          const int key = s_tex.key[i];
          if (key < 0) continue;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float a0 = (float)s_tex.val[i][c], a1 = (float)s_tex.val[i][3 + c];
            if (a0 != 0.f) atomicAdd(A.g_tex + (size_t)key * 3 + c, a0);
            if (a1 != 0.f) atomicAdd(A.g_nmap + (size_t)key * 3 + c, a1);
          }
        }
      }
    }
  }
}

__global__ void pack_texels_kernel(const float* __restrict__ tex, const float* __restrict__ nmap, int n, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[2 * i] = make_float4(tex[3 * i], tex[3 * i + 1], tex[3 * i + 2], nmap[3 * i]);
  out[2 * i + 1] = make_float4(nmap[3 * i + 1], nmap[3 * i + 2], 0.f, 0.f);
}
// F.normalize of the raw normal map (losses.hip: normalize3_fwd_kernel, same expression) and the packing above in one launch
__global__ void normalize_pack_kernel(const float* __restrict__ tex, const float* __restrict__ nmap_raw, int n, float* __restrict__ nmap_n,
                                      float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float n0, n1, n2;
  normalize3_texel(nmap_raw[3 * i], nmap_raw[3 * i + 1], nmap_raw[3 * i + 2], n0, n1, n2);
  nmap_n[3 * i] = n0; nmap_n[3 * i + 1] = n1; nmap_n[3 * i + 2] = n2;
  if (out) {
    out[2 * i] = make_float4(tex[3 * i], tex[3 * i + 1], tex[3 * i + 2], n0);
    out[2 * i + 1] = make_float4(n1, n2, 0.f, 0.f);
  }
}

// zbuf backward of a K=1 pass (used for the light-view depth map the shadow test gathers from):
// zbuf = sum_i bary_i z_i  ->  g on the face's NDC vertices (rasterize_meshes_backward, grad_zbuf path).
// CONSUME: the pass also CLEARS every non-zero entry of g_z it reads.  In a fitting step g_z is the shadow-map gradient image the shader
// backward scatters into: non-zero only at light-view pixels that hold a face (an empty texel reads depth -1, whose shadow-test sigmoid
// and gradient are exactly 0), i.e. only inside the tiles this kernel visits — so the image is all-zero again when the kernel is done and
// the 33.5-MB clear of every step (B x 512 x 512 floats) goes away.
// NMAP: the launch carries `M.blocks` more workgroups behind the tile workgroups, which apply the chain rule of the normal-map
// normalisation (losses.hip: normalize3_bwd_kernel) — the other small kernel between the shader backward and the per-frame backward
// tail; both only need the shader backward's output and neither reads what the other writes.
struct NmapBwd {
  const float* x; const float* gy; float* gx; int n; int blocks;
  // ... and `v9_blocks` more that add the shader backward's interleaved vertex gradients (harp_shade_args.g_vert9: 9 floats per vertex) into
  // the three per-vertex arrays the mesh-chain backward reads, and clear them
  float* v9; float* gv; float* gn; float* gd; int nv; int v9_blocks;
};
// g9 (nv, 9) = [g_verts(3) | g_vnormals(3) | g_ndc(3)] per vertex -> += into the three (nv,3) arrays, g9 cleared.  Atomic adds: the silhouette
// backward may still be adding to g_ndc on another stream (consecutive addresses: the coalesced rate)
__device__ __forceinline__ void vert9_unpack_body(float* __restrict__ g9, float* __restrict__ gv, float* __restrict__ gn, float* __restrict__ gd,
                                                  int nv, size_t first, size_t stride) {
  for (size_t i = first; i < (size_t)nv * 9; i += stride) {
    const float x = g9[i];
    if (x != 0.f) {
      const size_t v = i / 9;
      const int c = (int)(i - 9 * v);
      float* dst = c < 3 ? gv : (c < 6 ? gn : gd);
      atomicAdd(dst + 3 * v + (c % 3), x);
      g9[i] = 0.f;
    }
  }
}
__global__ void __launch_bounds__(256) vert9_unpack_kernel(float* __restrict__ g9, float* __restrict__ gv, float* __restrict__ gn, float* __restrict__ gd, int nv) {
  vert9_unpack_body(g9, gv, gn, gd, nv, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}
template <bool CONSUME, bool NMAP = false>
__global__ void __launch_bounds__(256) depth_bwd_kernel(const int32_t* __restrict__ face_id, const FaceRec* __restrict__ recs,
                                                        const int32_t* __restrict__ faces, float* __restrict__ g_z,
                                                        int V, int F, int S, float* __restrict__ g_ndc,
                                                        const int32_t* __restrict__ order, const int32_t* __restrict__ nact,
                                                        int B, int nsx, const NmapBwd M, unsigned char* __restrict__ tiles) {
  __shared__ VertexAccum<256, 3> s_acc;
  const unsigned tile_blocks = NMAP ? gridDim.x - (unsigned)M.blocks - (unsigned)M.v9_blocks : gridDim.x;
  if (NMAP && blockIdx.x >= tile_blocks) {
    const unsigned e = blockIdx.x - tile_blocks;
    if (e < (unsigned)M.blocks) {
      for (size_t i = (size_t)e * 256 + threadIdx.x; i < (size_t)M.n; i += (size_t)M.blocks * 256) normalize3_bwd_texel(M.x, M.gy, M.gx, i);
    } else {
      vert9_unpack_body(M.v9, M.gv, M.gn, M.gd, M.nv, (size_t)(e - M.blocks) * 256 + threadIdx.x, (size_t)M.v9_blocks * 256);
    }
    return;
  }
  // capped grid: a workgroup strides over the tiles of the super-tiles that hold faces (launch order).  The full grid is 16 workgroups
  // per (frame, super-tile) — 32 768 at 512^2, 131 072 at 1024^2 — of which a sixth have work here, and this small kernel was as long
  // as their dispatch (0.7 ns each: 24 us at 512^2, 107 us at 1024^2).  (The rasterisers keep the full grid: their tiles differ too much
  // in cost for a static assignment, measured.)
  const unsigned limit = (unsigned)(((nact[0] + 7) / 8) * 8 * (kSuper / kTile) * (kSuper / kTile));
  for (unsigned vb = blockIdx.x; vb < limit; vb += tile_blocks) {
  if (vb != blockIdx.x) __syncthreads();                 // s_acc of the previous tile has been flushed
  int b, st, tx0, ty0, tsub;
  if (tile_decode_v(vb, order, nact, B, nsx, S, b, st, tx0, ty0, tsub, false) != 1) continue;   // no tile / super-tile without a single face
  if (tiles) {                                           // the shader backward flags the tiles it adds to (about half of the ones visited here)
    const int znt = (S + 15) >> 4;
    unsigned char* fl = tiles + ((size_t)b * znt + (ty0 >> 4)) * znt + (tx0 >> 4);
    if (*fl == 0) continue;                              // (uniform: no barrier is skipped by part of the workgroup)
    __syncthreads();                                     // every wave has read the flag
    if (threadIdx.x == 0) *fl = 0;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xi = tx0 + (lane & 15), yi = ty0 + w * 4 + (lane >> 4);
  const bool in_img = xi < S && yi < S;
  const size_t o = ((size_t)b * S + (in_img ? yi : 0)) * S + (in_img ? xi : 0);
  // face id and gradient are requested together: one dependent round trip and one barrier per tile instead of two (the tile loop is a
  // chain of such trips: 81 -> 70 us at 1024^2); the 4 bytes per pixel read for nothing in tiles without faces do not show
  const int f = in_img ? face_id[o] : -1;
  const float g = in_img ? g_z[o] : 0.f;
  if (CONSUME && g != 0.f) g_z[o] = 0.f;
  const bool act = f >= 0 && g != 0.f;
  if (__syncthreads_or(act ? 1 : 0) == 0) continue;
  s_acc.clear();
  __syncthreads();
  float* gdb = g_ndc + (size_t)b * V * 3;
  float gnd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (act) {
    const Tri t = load_tri(recs + (size_t)b * F + f);
    const float px = pix_to_ndc(xi, S), py = pix_to_ndc(yi, S);
    const Bary br = bary_fwd(t, px, py);
    gnd[2] = g * br.b0; gnd[5] = g * br.b1; gnd[8] = g * br.b2;
    bary_bwd(t, px, py, br, g * t.z0, g * t.z1, g * t.z2, gnd);
  }
  bool alive = act;
  merge_same_face<9, SHADE_PRERED>(gnd, act ? f : -1, alive, lane);   // lanes on the same face add to the same three vertices
  if (alive) {
    const int vi[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int slot = s_acc.find(vi[k]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (slot >= 0) s_acc.add(slot, c, gnd[3 * k + c]); else atomicAdd(gdb + 3 * vi[k] + c, gnd[3 * k + c]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 256) {
    const int v = s_acc.key[i];
    if (v < 0) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (s_acc.val[i][c] != 0) atomicAdd(gdb + 3 * v + c, (float)s_acc.val[i][c]);
  }
  }
}

}  // namespace

int harp_detail_shade_bwd_wave(const harp_shade_args& a, const int32_t* order, const int32_t* nact, int nsx, unsigned grid, hipStream_t stream);
int harp_detail_fused_bwd(const harp_shade_args& a, const void* ws, const int32_t* faces, float blur, float sigma, const float* alpha,
                          const float* g_alpha, hipStream_t stream);

extern "C" {

int harp_pack_texels(const float* tex, const float* nmap, int n_texels, float* out, hipStream_t stream) {
  if (!tex || !nmap || !out || n_texels <= 0) return HARP_ERR_ARG;
  hipLaunchKernelGGL(pack_texels_kernel, dim3((n_texels + 255) / 256), dim3(256), 0, stream, tex, nmap, n_texels, (float4*)out);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_shade_fwd(const harp_shade_args* a, hipStream_t stream) {
  if (!a || !a->face_id || !a->recs || !a->faces || !a->faces_uvs || !a->verts_uvs || !a->verts || !a->vnormals || !a->tex ||
      !a->light_pos || !a->colors || (!a->rgb && !a->l1_target) || (a->zl && (!a->light_R || !a->light_T)))
    return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)a->recs, a->B, a->F, a->S);
  const dim3 grid(tile_grid(a->B, W.nsx));
  harp_shade_args b = *a;
  if (b.l1_target) {
    if (!b.l1_fid || !b.l1_w || !b.l1_loss || !b.l1_grad) return HARP_ERR_ARG;
    b.l1_inv = 1.0f / ((float)b.B * (float)b.S * (float)b.S * 3.0f);
  }
  hipLaunchKernelGGL(shade_kernel<false>, grid, dim3(256), 0, stream, b, (const int32_t*)W.order, (const int32_t*)W.nact, W.nsx);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_shade_bwd(const harp_shade_args* a, hipStream_t stream) {
  if (!a || !a->face_id || !a->recs) return HARP_ERR_ARG;
  // all three geometry gradients, or none of them (appearance-only stage: production kernel only)
  const bool geom = a->g_verts && a->g_vnormals && a->g_ndc;
  if (!geom && (a->g_verts || a->g_vnormals || a->g_ndc || (a->debug_skip & 0xff))) return HARP_ERR_ARG;
  harp_shade_args b = *a;
  if (!b.g_rgb) {
    // fused-loss mode (no harp_shade_fwd call at all): the photometric L1 and its gradient are formed here; the background part of
    // the term in super-tiles without a face comes from the static-target table, which is mandatory in this mode
    if (!b.l1_target || !b.l1_fid || !b.l1_w || !b.l1_loss || !b.l1_bg_sums) return HARP_ERR_ARG;
    b.l1_inv = 1.0f / ((float)b.B * (float)b.S * (float)b.S * 3.0f);
  } else {
    b.l1_target = nullptr;          // a caller that hands over the gradient image gets the plain backward pass
  }
  if (b.g_vert9 && (!geom || (b.debug_skip & 0xff))) return HARP_ERR_ARG;      // interleaved vertex gradients: production kernel, geometry gradients wanted
  // texel-gradient records (production kernel only): the list counters and a capacity come with the record buffer
  if (b.trec && (!b.trec_cnt || b.trec_cap <= 0 || (b.trec_cap & 3) || (b.g_tex && !b.trec_acc_tex) || (b.nmap && b.g_nmap && !b.trec_acc_nmap) || (b.debug_skip & 0xff) || b.Wt > 65535 || b.Ht > 65535)) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)a->recs, a->B, a->F, a->S);
  // production: the wave-autonomous kernel of shade_bwd.hip; debug_skip != 0 selects the first, barrier-synchronised version below
  // (bit 6 alone = that kernel unmodified, for A/B timing; bits 0-5 = its ablation switches)
  if ((b.debug_skip & 0xff) == 0) return harp_detail_shade_bwd_wave(b, (const int32_t*)W.order, (const int32_t*)W.nact, W.nsx, tile_grid(a->B, W.nsx), stream);
  hipLaunchKernelGGL(shade_kernel<true>, dim3(tile_grid(a->B, W.nsx)), dim3(256), 0, stream, b, (const int32_t*)W.order, (const int32_t*)W.nact, W.nsx);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// harp_shade_bwd + harp_silhouette_bwd of the SAME camera-view rasterisation (a->recs) as one launch
int harp_shade_sil_bwd(const harp_shade_args* a, float blur_radius, float sigma, const float* alpha, const float* g_alpha, hipStream_t stream) {
  if (!a || !a->face_id || !a->recs || !a->faces || !a->g_verts || !a->g_vnormals || !a->g_ndc || !alpha || !g_alpha) return HARP_ERR_ARG;
  harp_shade_args b = *a;
  if (!b.g_rgb) {
    if (!b.l1_target || !b.l1_fid || !b.l1_w || !b.l1_loss || !b.l1_bg_sums) return HARP_ERR_ARG;
    b.l1_inv = 1.0f / ((float)b.B * (float)b.S * (float)b.S * 3.0f);
  } else {
    b.l1_target = nullptr;
  }
  b.debug_skip = 0;
  b.trec = nullptr;               // (the one-launch pair hosts the table form of the shader tile: texel gradients go straight into g_tex / g_nmap)
  return harp_detail_fused_bwd(b, a->recs, a->faces, blur_radius, sigma, alpha, g_alpha, stream);
}

// g_z (B,S,S) -> g_ndc (B,V,3) += ; ws = workspace of the harp_rasterize_fwd call that produced face_id
int harp_depth_bwd(const int32_t* face_id, const void* ws, const int32_t* faces, const float* g_z, int B, int V, int F, int S,
                   float* g_ndc, hipStream_t stream) {
  if (!face_id || !ws || !faces || !g_z || !g_ndc) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)ws, B, F, S);
  hipLaunchKernelGGL(depth_bwd_kernel<false>, dim3(min(tile_grid(B, W.nsx), 4096u)), dim3(256), 0, stream, face_id, (const FaceRec*)ws, faces, (float*)g_z, V, F, S,
                     g_ndc, (const int32_t*)W.order, (const int32_t*)W.nact, B, W.nsx, NmapBwd{}, nullptr);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// the same, and g_z is all-zero again afterwards (see depth_bwd_kernel<CONSUME>): for callers that keep ONE gradient image across steps
int harp_depth_bwd_consume(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S,
                           float* g_ndc, hipStream_t stream) {
  if (!face_id || !ws || !faces || !g_z || !g_ndc) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)ws, B, F, S);
  hipLaunchKernelGGL(depth_bwd_kernel<true>, dim3(min(tile_grid(B, W.nsx), 4096u)), dim3(256), 0, stream, face_id, (const FaceRec*)ws, faces, g_z, V, F, S, g_ndc,
                     (const int32_t*)W.order, (const int32_t*)W.nact, B, W.nsx, NmapBwd{}, nullptr);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_depth_bwd_tiles(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S,
                         float* g_ndc, unsigned char* g_z_tiles, hipStream_t stream) {
  if (!face_id || !ws || !faces || !g_z || !g_ndc) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)ws, B, F, S);
  hipLaunchKernelGGL(depth_bwd_kernel<true>, dim3(min(tile_grid(B, W.nsx), 4096u)), dim3(256), 0, stream, face_id, (const FaceRec*)ws, faces, g_z, V, F, S, g_ndc,
                     (const int32_t*)W.order, (const int32_t*)W.nact, B, W.nsx, NmapBwd{}, g_z_tiles);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// harp_depth_bwd_consume + harp_normalize3_bwd(nmap, g_nmap_n, n_texels, g_nmap) as ONE launch
int harp_depth_nmap_bwd(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S, float* g_ndc,
                        const float* nmap, const float* g_nmap_n, int n_texels, float* g_nmap, unsigned char* g_z_tiles, hipStream_t stream) {
  if (!face_id || !ws || !faces || !g_z || !g_ndc || !nmap || !g_nmap_n || !g_nmap || n_texels <= 0) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)ws, B, F, S);
  const NmapBwd M{nmap, g_nmap_n, g_nmap, n_texels, min((n_texels + 255) / 256, 1024), nullptr, nullptr, nullptr, nullptr, 0, 0};
  hipLaunchKernelGGL((depth_bwd_kernel<true, true>), dim3(min(tile_grid(B, W.nsx), 4096u) + (unsigned)M.blocks), dim3(256), 0, stream, face_id,
                     (const FaceRec*)ws, faces, g_z, V, F, S, g_ndc, (const int32_t*)W.order, (const int32_t*)W.nact, B, W.nsx, M, g_z_tiles);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// harp_depth_bwd_tiles (consume; g_z_tiles optional) with up to two riders in the same launch: the normal map's chain rule (nmap != NULL:
// harp_depth_nmap_bwd's) and the unpacking of the shader backward's interleaved vertex gradients (g_vert9 != NULL)
int harp_depth_bwd_riders(const int32_t* face_id, const void* ws, const int32_t* faces, float* g_z, int B, int V, int F, int S, float* g_ndc,
                          unsigned char* g_z_tiles, const float* nmap, const float* g_nmap_n, int n_texels, float* g_nmap, float* g_vert9,
                          float* g_verts, float* g_vnormals, float* g_ndc_cam, hipStream_t stream) {
  if (!face_id || !ws || !faces || !g_z || !g_ndc) return HARP_ERR_ARG;
  if (nmap && (!g_nmap_n || !g_nmap || n_texels <= 0)) return HARP_ERR_ARG;
  if (g_vert9 && (!g_verts || !g_vnormals || !g_ndc_cam)) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)ws, B, F, S);
  NmapBwd M{nmap, g_nmap_n, g_nmap, nmap ? n_texels : 0, nmap ? min((n_texels + 255) / 256, 1024) : 0,
            g_vert9, g_verts, g_vnormals, g_ndc_cam, g_vert9 ? B * V : 0, g_vert9 ? min((B * V * 9 + 255) / 256, 1024) : 0};
  hipLaunchKernelGGL((depth_bwd_kernel<true, true>), dim3(min(tile_grid(B, W.nsx), 4096u) + (unsigned)M.blocks + (unsigned)M.v9_blocks), dim3(256), 0, stream,
                     face_id, (const FaceRec*)ws, faces, g_z, V, F, S, g_ndc, (const int32_t*)W.order, (const int32_t*)W.nact, B, W.nsx, M, g_z_tiles);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_vert9_unpack(float* g_vert9, int n_verts, float* g_verts, float* g_vnormals, float* g_ndc, hipStream_t stream) {
  if (!g_vert9 || !g_verts || !g_vnormals || !g_ndc || n_verts <= 0) return HARP_ERR_ARG;
  hipLaunchKernelGGL(vert9_unpack_kernel, dim3(min((n_verts * 9 + 255) / 256, 1024)), dim3(256), 0, stream, g_vert9, g_verts, g_vnormals, g_ndc, n_verts);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_normalize3_pack(const float* tex, const float* nmap_raw, int n_texels, float* nmap_n, float* packed, hipStream_t stream) {
  if (!nmap_raw || !nmap_n || n_texels <= 0 || (packed && !tex)) return HARP_ERR_ARG;
  const char* e = getenv("HARP_SIDE_LDS");           // (see harp_step_prologue: keeps this kernel off the CUs that run hand_front)
  hipLaunchKernelGGL(normalize_pack_kernel, dim3((n_texels + 255) / 256), dim3(256), e ? (size_t)atoi(e) : 512, stream, tex, nmap_raw, n_texels, nmap_n, (float4*)packed);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
