// Tile rasteriser for gfx950: replaces PyTorch3D's rasterize_meshes (coarse+fine) and, fused into the same
// per-pixel face walk, SoftSilhouetteShader/sigmoid_alpha_blend.
//   reference call sites: renderer/renderer_helper.py:52-58 (K=50 soft silhouette), :76-79 / :444-447 (K=1),
//   :344, :353 (light view, camera view).  Semantics: SURVEY.md Appendix A.2/A.3.
//
// Design (not PyTorch3D's): nothing of shape (B,S,S,K) is ever materialised.
//   1. face_setup   one thread per (frame, face): gather 3 NDC vertices, write a 64-B FaceRec with the
//                   blur-dilated bbox; culled faces get an empty box.
//   2. bin_faces    one workgroup per (frame, 64x64 super-tile): scan the frame's bboxes (coalesced float4),
//                   wave-ballot compaction -> ascending face-id list in HBM/L2 (deterministic, no atomics).
//   3. raster       one workgroup (4 waves) per 16x16 tile.  Stages the tile's faces into LDS (SoA float4); the hard pass walks the (staged face,
//                   bbox pixel) pairs densely packed over the lanes (prefix sum over the clipped bbox sizes) and keeps the nearest face per
//                   pixel as a 64-bit LDS min over (depth bits, face id) (K=1 semantics, ties -> lower face id like PyTorch3D), flagging the
//                   pixels a face saturates; the soft pass pairs the remaining pixels with the faces near them and multiplies the silhouette
//                   product prod_f (1 - sigmoid(-d_f/sigma)) in ascending face order (raster_body.h).
//   4. sil_bwd      same walk, rim pixels only: dL/dalpha -> dL/d(ndc xy) of the face vertices (atomics).
#include <stdlib.h>
#include "raster_body.h"

namespace {

using rb::Tri;

// BITS: the pass also writes the super-tile hit bitmaps (round 4).  A wave owns 64 consecutive faces = exactly one 64-bit word of every
// (frame, super-tile) bitmap, so the words are plain stores — no atomics, nothing to clear, every word is rewritten by every call.  The hit
// test of the binning pass is separable in x and y: nsx + nsx ballots per wave give the column masks and the row masks, lane l then forms
// the word of super-tile l as (column mask) & (row mask).  Same comparison, same floats as rb::bin_super_tile -> identical lists.
template <bool BITS>
__device__ __forceinline__ void face_setup_body(const float* __restrict__ ndc, const int32_t* __restrict__ faces,
                                                int V, int F, float r, FaceRec* __restrict__ recs,
                                                float4* __restrict__ bbs, int S, int nsx, unsigned long long* __restrict__ bits, int W64) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  float4 bb = make_float4(3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f);
  if (f < F) {
    const FaceRec rec = rb::face_rec(ndc + (size_t)b * V * 3, faces, f, r);
    recs[(size_t)b * F + f] = rec;
    bbs[(size_t)b * F + f] = rec.bb;      // contiguous copy: the tile kernels' staging streams it with fully coalesced 16-B loads
    bb = rec.bb;
  }
  if constexpr (BITS) {
    const int lane = threadIdx.x & 63;
    const int chunk = f >> 6;             // wave-uniform: blockDim.x is a multiple of 64
    if (chunk >= W64) return;
    unsigned long long kx = 0ull, ky = 0ull;
    for (int s = 0; s < nsx; ++s) {
      const int lo = s * kSuper, hi = min(lo + kSuper, S) - 1;
      const float n_hi = pix_to_ndc(lo, S), n_lo = pix_to_ndc(hi, S);       // NDC decreases with the pixel index
      const unsigned long long mx = __ballot(!(n_lo > bb.y || n_hi < bb.x));
      const unsigned long long my = __ballot(!(n_lo > bb.w || n_hi < bb.z));
      if (lane == s) { kx = mx; ky = my; }
    }
    const int nst = nsx * nsx;
    for (int g = 0; g < nst; g += 64) {
      const int st = g + lane;
      const int sx = st % nsx, sy = min(st / nsx, nsx - 1);
      const unsigned xl = __shfl((unsigned)kx, sx, 64), xh = __shfl((unsigned)(kx >> 32), sx, 64);
      const unsigned yl = __shfl((unsigned)ky, sy, 64), yh = __shfl((unsigned)(ky >> 32), sy, 64);
      if (st < nst) bits[((size_t)b * nst + st) * W64 + chunk] = ((unsigned long long)(xh & yh) << 32) | (unsigned long long)(xl & yl);
    }
  }
}

template <bool BITS>
__global__ void __launch_bounds__(256) face_setup_kernel(const float* __restrict__ ndc, const int32_t* __restrict__ faces,
                                                         int V, int F, float r, FaceRec* __restrict__ recs,
                                                         float4* __restrict__ bbs, int S, int nsx, unsigned long long* __restrict__ bits, int W64) {
  face_setup_body<BITS>(ndc, faces, V, F, r, recs, bbs, S, nsx, bits, W64);
}

// Both views of a fitting step in ONE launch each (harp_raster_setup_pair): blockIdx.z selects the view.  The light view's three set-up
// launches were 61 us of the second stream's chain in front of the light raster (profiles/r05_f_timeline_one_step.txt); as the second half
// of the camera view's grids they cost the main stream a few microseconds, and the light raster can start when the camera raster does.
struct SetupView { const float* ndc; float r; RasterWs W; };

__global__ void __launch_bounds__(256) face_setup_pair_kernel(const SetupView a, const SetupView b, const int32_t* __restrict__ faces, int V, int F, int S) {
  const SetupView& v = blockIdx.z ? b : a;
  face_setup_body<true>(v.ndc, faces, V, F, v.r, v.W.recs, v.W.bbs, S, v.W.nsx, v.W.bits, v.W.W64);
}

// One WAVE per (frame, 64x64 super-tile): streams the frame's bboxes 64 at a time, ballot + popcount compaction,
// no LDS, no barriers; the list comes out in ascending face order (== PyTorch3D's tie-break order).
// (rounds 1-3; since round 4 only for images above 4096 pixels a side, whose super-tile columns no longer fit the lanes of a wave)
__global__ void __launch_bounds__(256) bin_faces_kernel(const float4* __restrict__ bbs, int F, int S, int nsx,
                                                        int32_t* __restrict__ bins, int32_t* __restrict__ bin_count) {
  const int nst = nsx * nsx;
  const int st = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (st >= nst) return;
  rb::bin_super_tile(bbs + (size_t)b * F, F, S, nsx, st, bins + ((size_t)b * nst + st) * F, bin_count + b * nst + st);
}

// Round 4: the lists from the hit bitmaps.  One wave per (frame, super-tile) reads its W64 words (coalesced, ONE round trip for up to
// 4096 faces instead of the 13 dependent trips of the bbox scan), prefix-sums the popcounts and writes the set bits in ascending order.
// The scan took 32 us of the 51 us between the end of the hand layer and the first raster workgroup (profiles/r04_a_timeline_one_step.txt).
__device__ __forceinline__ void expand_bits_body(const unsigned long long* __restrict__ bits, int F, int W64, int nst,
                                                 int32_t* __restrict__ bins, int32_t* __restrict__ bin_count) {
  const int st = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, lane = threadIdx.x & 63;
  if (st >= nst) return;
  const unsigned long long* src = bits + ((size_t)b * nst + st) * W64;
  int32_t* out = bins + ((size_t)b * nst + st) * F;
  int total = 0;
  for (int base = 0; base < W64; base += 128) {            // two words per lane in flight (6152 / 8128 faces: ONE round trip per wave instead of two)
    unsigned long long w0 = (base + lane < W64) ? src[base + lane] : 0ull;
    unsigned long long w1 = (base + 64 + lane < W64) ? src[base + 64 + lane] : 0ull;
    const int c0 = __popcll(w0), c1 = __popcll(w1);
    int incl = c0 | (c1 << 16);                            // both prefix sums in one scan (<= 4096 set bits per 64 words)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    const int tot = __shfl(incl, 63, 64);
    int pos = total + (incl & 0xffff) - c0;
    const int f0 = (base + lane) * 64;
    while (w0) {
      out[pos++] = f0 + (int)__builtin_ctzll(w0);
      w0 &= w0 - 1ull;
    }
    pos = total + (tot & 0xffff) + (incl >> 16) - c1;
    while (w1) {
      out[pos++] = f0 + 4096 + (int)__builtin_ctzll(w1);
      w1 &= w1 - 1ull;
    }
    total += (tot & 0xffff) + (tot >> 16);
  }
  if (lane == 0) bin_count[b * nst + st] = total;
}
__global__ void __launch_bounds__(256) expand_bits_kernel(const unsigned long long* __restrict__ bits, int F, int W64, int nst,
                                                          int32_t* __restrict__ bins, int32_t* __restrict__ bin_count) {
  expand_bits_body(bits, F, W64, nst, bins, bin_count);
}
__global__ void __launch_bounds__(256) expand_bits_pair_kernel(const RasterWs a, const RasterWs b, int F) {
  const RasterWs& W = blockIdx.z ? b : a;
  expand_bits_body(W.bits, F, W.W64, W.nsx * W.nsx, W.bins, W.cnt);
}

// (Folding this into the binning pass with a "last workgroup done" ticket was measured: 512 same-address ticket atomics cost ~35 us,
//  8x the launch they save.)
__global__ void __launch_bounds__(1024) order_tiles_kernel(const int32_t* __restrict__ bin_count, int total, int32_t* __restrict__ order,
                                                           int32_t* __restrict__ nact) {
  __shared__ int s_hist[33], s_base[33];
  rb::order_tiles(bin_count, total, order, nact, s_hist, s_base);
}

__global__ void __launch_bounds__(1024) order_tiles_pair_kernel(const RasterWs a, const RasterWs b, int total) {
  __shared__ int s_hist[33], s_base[33];
  const RasterWs& W = blockIdx.x ? b : a;
  rb::order_tiles(W.cnt, total, W.order, W.nact, s_hist, s_base);
}

#ifndef RASTER_OCC1
#define RASTER_OCC1 7
#endif
#ifndef RASTER_OCC0
#define RASTER_OCC0 7
#endif
#ifndef RASTER_OCC2
#define RASTER_OCC2 8   // silhouette backward: 64 VGPRs, 8 waves per SIMD (18 KB of LDS per workgroup): 53 -> 51 us, step -3 us (5: the round-3 setting for its 32-KB form)
#endif
constexpr unsigned kRasterGrid = 16384;

template <int MODE, bool LOOP, bool BWD = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? RASTER_OCC2 : (MODE == 0 ? RASTER_OCC0 : (BWD ? 5 : RASTER_OCC1)), 8))) raster_kernel(const FaceRec* __restrict__ recs, const float4* __restrict__ bbs,
                                                     const int32_t* __restrict__ bins,
                                                     const int32_t* __restrict__ bin_count, const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ nact, int B, int F, int S, int nsx,
                                                     float blur, float sigma, int32_t* __restrict__ face_id,
                                                     float* __restrict__ zbuf, float* __restrict__ alpha,
                                                     const float* __restrict__ g_alpha, const int32_t* __restrict__ faces,
                                                     int V, float* __restrict__ g_ndc, const float* __restrict__ l1_target,
                                                     const int32_t* __restrict__ l1_fid, const float* __restrict__ l1_w,
                                                     float* __restrict__ l1_loss, float* __restrict__ l1_grad, float l1_inv, int sparse,
                                                     const float* __restrict__ l1_bg_sums, int32_t* __restrict__ st_state) {
  __shared__ rb::RasterSmem<MODE, BWD> sm;
  if constexpr (!LOOP) {
    rb::raster_tile<MODE, BWD>(sm, blockIdx.x, recs, bbs, bins, bin_count, order, nact, B, F, S, nsx, blur, sigma, face_id, zbuf, alpha, g_alpha, faces, V, g_ndc,
                          l1_target, l1_fid, l1_w, l1_loss, l1_grad, l1_inv, sparse, l1_bg_sums, st_state);
    return;
  }
  // LOOP (grids above 64 k workgroups, i.e. 1024^2 and up): the grid is capped at kRasterGrid workgroups (a multiple of 8, so a workgroup's tiles stay on its XCD) and a workgroup strides over
  // the heaviest-first tile order: 131 072 workgroups of which a quarter have work cost more to dispatch than a few tiles per workgroup
  // cost in balance (C5 step 2.30 -> 2.23 ms; at 512^2 the plain grid is faster: camera view 0.191 vs 0.208 ms).  Backward, and forward with sparse outputs, stop at the last slot that holds faces; the fused silhouette L1 of the
  // un-rendered super-tiles (a table look-up per super-tile) is added by a grid-stride loop at the end.
  const unsigned total = tile_grid(B, nsx);
  const int nst = nsx * nsx;
  const bool bg_table = (MODE == 1) && sparse && l1_target && l1_bg_sums;
  const bool skip_empty = (MODE == 2) || bg_table || (sparse && !zbuf && !(MODE == 1 && l1_target));      // (depth maps are always complete)
  const unsigned limit = skip_empty ? min(total, (unsigned)(((nact[0] + 7) / 8) * 8 * (kSuper / kTile) * (kSuper / kTile))) : total;
  for (unsigned v = blockIdx.x; v < limit; v += gridDim.x) {
    if (v != blockIdx.x) __syncthreads();                 // LDS of the previous tile
    rb::raster_tile<MODE, BWD>(sm, v, recs, bbs, bins, bin_count, order, nact, B, F, S, nsx, blur, sigma, face_id, zbuf, alpha, g_alpha, faces, V, g_ndc,
                          l1_target, l1_fid, l1_w, l1_loss, l1_grad, l1_inv, sparse, l1_bg_sums, st_state);
  }
  if (bg_table) {
    // slots below `limit` (nact rounded up to 8) already went through raster_tile, whose sub == 0 workgroup adds the table value of an
    // empty super-tile: start behind them, or up to 7 empty super-tiles are counted twice whenever nact % 8 != 0
    float acc = 0.f;
    const int first = min(B * nst, (nact[0] + 7) / 8 * 8);
    for (int slot = first + (int)(blockIdx.x * blockDim.x + threadIdx.x); slot < B * nst; slot += (int)(gridDim.x * blockDim.x)) {
      const int entry = order[slot], b = entry / nst, st = entry - b * nst;
      acc += l1_bg_sums[(size_t)l1_fid[b] * nst + st];
    }
    if (acc != 0.f) atomicAdd(l1_loss, acc * l1_inv);
  }
}

}  // namespace

static void raster_setup_any(const float* ndc, const int32_t* faces, int B, int V, int F, int S, float r, void* ws, hipStream_t stream) {
  const RasterWs W = raster_ws_split(ws, B, F, S);
  static const bool old_bins = [] { const char* e = getenv("HARP_BIN_SCAN"); return e && atoi(e) != 0; }();     // A/B: the bbox scan of rounds 1-3
  if (W.nsx <= 64 && !old_bins) {
    hipLaunchKernelGGL(face_setup_kernel<true>, dim3((F + 255) / 256, B), dim3(256), 0, stream, ndc, faces, V, F, r, W.recs, W.bbs, S, W.nsx, W.bits, W.W64);
    hipLaunchKernelGGL(expand_bits_kernel, dim3((W.nsx * W.nsx + 3) / 4, B), dim3(256), 0, stream, W.bits, F, W.W64, W.nsx * W.nsx, W.bins, W.cnt);
  } else {
    hipLaunchKernelGGL(face_setup_kernel<false>, dim3((F + 255) / 256, B), dim3(256), 0, stream, ndc, faces, V, F, r, W.recs, W.bbs, S, W.nsx, nullptr, 0);
    hipLaunchKernelGGL(bin_faces_kernel, dim3((W.nsx * W.nsx + 3) / 4, B), dim3(256), 0, stream, W.bbs, F, S, W.nsx, W.bins, W.cnt);
  }
  hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, stream, W.cnt, B * W.nsx * W.nsx, W.order, W.nact);
}

// capped, striding grid for launches above 64 k workgroups.  HARP_RASTER_LOOP=0 forces the plain grid, HARP_RASTER_LOOP=<n >= 8>
// forces the striding kernels with n workgroups (tests run them on small images).  Returns the grid size, 0 = plain grid.
static unsigned raster_loop_grid(unsigned full_grid) {
  const char* e = getenv("HARP_RASTER_LOOP");
  if (e) {
    const int n = atoi(e);
    if (n <= 0) return 0u;
    return min(full_grid, (unsigned)((max(n, 8) + 7) / 8 * 8));
  }
  return full_grid > 65536u ? kRasterGrid : 0u;
}

// per-face records with a bbox dilated by r, per-super-tile face lists (ascending), launch order: shared with csrc/fragments.hip
int harp_detail_raster_setup(const float* ndc, const int32_t* faces, int B, int V, int F, int S, float r, void* ws, hipStream_t stream) {
  raster_setup_any(ndc, faces, B, V, F, S, r, ws, stream);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

extern "C" {

size_t harp_rasterize_ws_bytes(int B, int F, int S) {
  const int nsx = (S + kSuper - 1) / kSuper;
  size_t recs = (size_t)B * F * sizeof(FaceRec);
  size_t bbs = (size_t)B * F * sizeof(float4);
  size_t bins = (size_t)B * nsx * nsx * F * sizeof(int32_t);
  size_t cnt = (((size_t)B * nsx * nsx * sizeof(int32_t)) + 255) / 256 * 256;
  size_t bits = (size_t)B * nsx * nsx * ((F + 63) / 64) * sizeof(unsigned long long);
  return recs + bbs + bins + 2 * cnt + 256 + bits;      // counts + launch order + number of non-empty super-tiles + hit bitmaps
}

// Forward rasterisation of B frames sharing one face table.
//   ndc (B,V,3) f32 [x_ndc, y_ndc, z_view]; faces (F,3) i32.
//   soft != 0: also accumulate the soft-silhouette alpha (blur_radius, sigma as in renderer_helper.py:44-58).
//   Outputs (B,S,S): face_id i32 (frame-local, -1 empty), zbuf f32 or NULL (-1 empty), alpha f32 (soft only).
//   ws: harp_rasterize_ws_bytes() bytes, 256-B aligned; must stay untouched until the matching backward ran.
// Same, with the silhouette L1 loss fused into the camera-view raster epilogue (soft != 0): loss (+=) mean |alpha - y_sil[fid]|,
// g_alpha = w * d loss / d alpha.  y_sil == NULL: plain rasterisation.
static int rasterize_impl(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                          float sigma, void* ws, int32_t* face_id, float* zbuf, float* alpha, const float* l1_target,
                          const int32_t* l1_fid, const float* l1_w, float* l1_loss, float* l1_grad, const float* l1_bg_sums,
                          int32_t* st_state, hipStream_t stream, float* g_ndc = nullptr) {
  // face_id == NULL: silhouette only (camera view of a geometry-only step: nothing reads the nearest face) — soft pass, no depth map
  if (!ndc || !faces || !ws || (!face_id && (!(soft & 1) || zbuf)) || B <= 0 || F <= 0 || S <= 0 || ((soft & 1) && !alpha)) return HARP_ERR_ARG;
  if (l1_target && (!(soft & 1) || !l1_fid || !l1_w || !l1_loss || !l1_grad)) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split(ws, B, F, S);
  FaceRec* recs = W.recs; int32_t *bins = W.bins, *cnt = W.cnt, *order = W.order; float4* bbs = W.bbs;
  const int nsx = W.nsx;
  const float r = (soft & 1) ? sqrtf(blur_radius) : 0.f;
  if (!(soft & 4)) raster_setup_any(ndc, faces, B, V, F, S, r, ws, stream);      // (bit 2: the workspace was set up by harp_raster_setup_pair)
  const unsigned lgrid = raster_loop_grid(tile_grid(B, nsx));
  const bool loop = lgrid != 0u;
  const dim3 grid(loop ? lgrid : tile_grid(B, nsx));
  const float l1_inv = 1.0f / ((float)B * (float)S * (float)S);
  const int sp = (soft & 2) ? 1 : 0;
#define HARP_RASTER_LAUNCH(MODE, LOOP, ...) hipLaunchKernelGGL((raster_kernel<MODE, LOOP>), grid, dim3(256), 0, stream, __VA_ARGS__)
  if ((soft & 1) && g_ndc) {
    // camera view with the silhouette backward of every tile fused in (harp_rasterize_l1_fwd_bwd)
    if (loop) hipLaunchKernelGGL((raster_kernel<1, true, true>), grid, dim3(256), 0, stream, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, blur_radius, sigma, face_id, zbuf,
                                 alpha, nullptr, faces, V, g_ndc, l1_target, l1_fid, l1_w, l1_loss, l1_grad, l1_inv, sp, l1_bg_sums, nullptr);
    else hipLaunchKernelGGL((raster_kernel<1, false, true>), grid, dim3(256), 0, stream, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, blur_radius, sigma, face_id, zbuf,
                            alpha, nullptr, faces, V, g_ndc, l1_target, l1_fid, l1_w, l1_loss, l1_grad, l1_inv, sp, l1_bg_sums, nullptr);
  } else if (soft & 1) {
    if (loop) HARP_RASTER_LAUNCH(1, true, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, blur_radius, sigma, face_id, zbuf, alpha, nullptr, nullptr, 0, nullptr,
                                 l1_target, l1_fid, l1_w, l1_loss, l1_grad, l1_inv, sp, l1_bg_sums, nullptr);
    else HARP_RASTER_LAUNCH(1, false, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, blur_radius, sigma, face_id, zbuf, alpha, nullptr, nullptr, 0, nullptr,
                            l1_target, l1_fid, l1_w, l1_loss, l1_grad, l1_inv, sp, l1_bg_sums, nullptr);
  } else {
    if (loop) HARP_RASTER_LAUNCH(0, true, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, 0.f, 1.f, face_id, zbuf, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, 0.f, sp, nullptr, st_state);
    else HARP_RASTER_LAUNCH(0, false, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, 0.f, 1.f, face_id, zbuf, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, 0.f, sp, nullptr, st_state);
  }
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// The rasteriser set-up (face records with the blur-dilated bbox, super-tile lists, launch order) of TWO views of the same meshes — the
// camera view and the light view of a fitting step — as three launches instead of six.  The rasteriser calls that follow pass bit 2 (value 4)
// in `soft` / `sparse`: "the workspace is set up".  blur_radius_x: the soft-silhouette radius of that view's raster pass (0: hard K = 1 pass).
int harp_raster_setup_pair(const float* ndc_a, float blur_radius_a, void* ws_a, const float* ndc_b, float blur_radius_b, void* ws_b,
                           const int32_t* faces, int B, int V, int F, int S, hipStream_t stream) {
  if (!ndc_a || !ndc_b || !ws_a || !ws_b || !faces || B <= 0 || V <= 0 || F <= 0 || S <= 0 || blur_radius_a < 0.f || blur_radius_b < 0.f) return HARP_ERR_ARG;
  SetupView a, b;
  a.ndc = ndc_a; a.r = sqrtf(blur_radius_a); a.W = raster_ws_split(ws_a, B, F, S);
  b.ndc = ndc_b; b.r = sqrtf(blur_radius_b); b.W = raster_ws_split(ws_b, B, F, S);
  if (a.W.nsx > 64) {                          // (images above 4096 px a side: the bbox-scan binning, one view after the other)
    raster_setup_any(ndc_a, faces, B, V, F, S, a.r, ws_a, stream);
    raster_setup_any(ndc_b, faces, B, V, F, S, b.r, ws_b, stream);
    HARP_CHECK_LAUNCH();
    return HARP_OK;
  }
  const int nst = a.W.nsx * a.W.nsx;
  hipLaunchKernelGGL(face_setup_pair_kernel, dim3((F + 255) / 256, B, 2), dim3(256), 0, stream, a, b, faces, V, F, S);
  hipLaunchKernelGGL(expand_bits_pair_kernel, dim3((nst + 3) / 4, B, 2), dim3(256), 0, stream, a.W, b.W, F);
  hipLaunchKernelGGL(order_tiles_pair_kernel, dim3(2), dim3(1024), 0, stream, a.W, b.W, B * nst);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_rasterize_l1_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                          float sigma, void* ws, int32_t* face_id, float* zbuf, float* alpha, const float* l1_target,
                          const int32_t* l1_fid, const float* l1_w, float* l1_loss, float* l1_grad, const float* l1_bg_sums,
                          hipStream_t stream) {
  return rasterize_impl(ndc, faces, B, V, F, S, soft, blur_radius, sigma, ws, face_id, zbuf, alpha, l1_target, l1_fid, l1_w, l1_loss, l1_grad,
                        l1_bg_sums, nullptr, stream);
}

// harp_rasterize_l1_fwd with the silhouette backward fused into the same launch: g_ndc (B,V,3) += d (w * L1) / d ndc (x, y components),
// what harp_silhouette_bwd(alpha, l1_grad) would add — formed tile by tile while the tile's faces are still staged in LDS.
int harp_rasterize_l1_fwd_bwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                              float sigma, void* ws, int32_t* face_id, float* alpha, const float* l1_target, const int32_t* l1_fid,
                              const float* l1_w, float* l1_loss, float* l1_grad, const float* l1_bg_sums, float* g_ndc, hipStream_t stream) {
  if (!g_ndc || !l1_target || !(soft & 1)) return HARP_ERR_ARG;
  return rasterize_impl(ndc, faces, B, V, F, S, soft, blur_radius, sigma, ws, face_id, nullptr, alpha, l1_target, l1_fid, l1_w, l1_loss, l1_grad,
                        l1_bg_sums, nullptr, stream, g_ndc);
}

int harp_rasterize_fwd(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int soft, float blur_radius,
                       float sigma, void* ws, int32_t* face_id, float* zbuf, float* alpha, hipStream_t stream) {
  return rasterize_impl(ndc, faces, B, V, F, S, soft, blur_radius, sigma, ws, face_id, zbuf, alpha, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, stream);
}

// K = 1 depth pass into a depth map the caller KEEPS between calls: st_state (B * nsx * nsx ints, zero before the first call, owned by
// the library afterwards) records which super-tiles hold -1 everywhere; with sparse outputs (sparse != 0: face ids of empty super-tiles
// are not written) a super-tile that is empty AGAIN is then not touched at all.  zbuf must not be written by anyone else between calls.
int harp_rasterize_fwd_keep(const float* ndc, const int32_t* faces, int B, int V, int F, int S, int sparse, void* ws, int32_t* face_id,
                            float* zbuf, int32_t* st_state, hipStream_t stream) {
  if (!zbuf || !st_state) return HARP_ERR_ARG;
  return rasterize_impl(ndc, faces, B, V, F, S, ((sparse & 3) ? 2 : 0) | (sparse & 4), 0.f, 1.f, ws, face_id, zbuf, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, st_state, stream);
}

// Soft-silhouette backward: g_alpha (B,S,S) -> accumulates (atomicAdd) into g_ndc (B,V,3) (x,y components).
// ws must be the workspace of the matching harp_rasterize_fwd(soft=1); alpha its output.
int harp_silhouette_bwd(const int32_t* faces, int B, int V, int F, int S, float blur_radius, float sigma, const void* ws,
                        const float* alpha, const float* g_alpha, float* g_ndc, hipStream_t stream) {
  if (!faces || !ws || !alpha || !g_alpha || !g_ndc) return HARP_ERR_ARG;
  const RasterWs W = raster_ws_split((void*)ws, B, F, S);
  FaceRec* recs = W.recs; int32_t *bins = W.bins, *cnt = W.cnt, *order = W.order; float4* bbs = W.bbs;
  const int nsx = W.nsx;
  const unsigned lgrid = raster_loop_grid(tile_grid(B, nsx));
  const bool loop = lgrid != 0u;
  const dim3 grid(loop ? lgrid : tile_grid(B, nsx));
  if (loop) HARP_RASTER_LAUNCH(2, true, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, blur_radius, sigma, nullptr, nullptr, (float*)alpha, g_alpha, faces, V, g_ndc,
                               nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, nullptr);
  else HARP_RASTER_LAUNCH(2, false, recs, bbs, bins, cnt, order, W.nact, B, F, S, nsx, blur_radius, sigma, nullptr, nullptr, (float*)alpha, g_alpha, faces, V, g_ndc,
                          nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, nullptr);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
