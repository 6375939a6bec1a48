// Tile rasteriser body shared by csrc/raster.hip (its three launches) and csrc/shade_bwd.hip (the fused backward launch runs the
// silhouette-backward tiles, MODE 2, interleaved with the shading tiles).  See raster.hip for the design notes.
#pragma once
#include "harp_common.h"
#include "harp_hip.h"
#include <type_traits>

namespace rb {

constexpr int kStage = 256;   // faces staged in LDS per round (17 KB)

struct Tri {
  float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

__device__ __forceinline__ Tri tri_from(const float4 a, const float4 b, const float4 c) {
  Tri t;
  t.x0 = a.x; t.y0 = a.y; t.z0 = a.z; t.x1 = a.w;
  t.y1 = b.x; t.z1 = b.y; t.x2 = b.z; t.y2 = b.w;
  t.z2 = c.x;
  return t;
}

// BarycentricCoordsForward + BarycentricPerspectiveCorrectionForward; returns "inside" (all bary > 0).
__device__ __forceinline__ bool tri_bary(const Tri& t, float px, float py, float& b0, float& b1, float& b2) {
  const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  const float w0 = edge_fn(px, py, t.x1, t.y1, t.x2, t.y2) / area;
  const float w1 = edge_fn(px, py, t.x2, t.y2, t.x0, t.y0) / area;
  const float w2 = edge_fn(px, py, t.x0, t.y0, t.x1, t.y1) / area;
  const float t0 = w0 * t.z1 * t.z2, t1 = t.z0 * w1 * t.z2, t2 = t.z0 * t.z1 * w2;
  const float den = fmaxf(t0 + t1 + t2, kEps);
  b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
  return b0 > 0.f && b1 > 0.f && b2 > 0.f;
}

// PointLineDistanceForward: squared distance to segment (a,b); also returns the clamped parameter t.
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by, float& tt) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= kEps) { tt = 1.f; return (px - bx) * (px - bx) + (py - by) * (py - by); }
  float t = (bax * (px - ax) + bay * (py - ay)) * __builtin_amdgcn_rcpf(l2);      // <= 1.5 ulp from the IEEE quotient
  t = fminf(fmaxf(t, 0.f), 1.f);
  tt = t;
  const float qx = ax + t * bax, qy = ay + t * bay;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

// Ascending-order compaction of `pred` across a 256-thread block. Returns this thread's slot (or -1) and
// advances *running (uniform). lds_cnt: 4 ints.
__device__ __forceinline__ int block_compact(bool pred, int running, int* lds_cnt, int& total) {
  const unsigned long long m = __ballot(pred);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds_cnt[w] = __popcll(m);
  __syncthreads();
  int base = running;
  for (int i = 0; i < w; ++i) base += lds_cnt[i];
  total = lds_cnt[0] + lds_cnt[1] + lds_cnt[2] + lds_cnt[3];
  const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  return pred ? pos : -1;
}

// MODE 0: depth only (light view).  MODE 1: nearest face + silhouette product (camera view).
// MODE 2: silhouette backward (rim pixels): scatter dL/d(ndc xy).
__device__ __forceinline__ int nst_of(int nsx) { return nsx * nsx; }

// staging capacity per round.  The silhouette backward (MODE 2) stages 128 faces at a time (a tile sees 30 - 100 of its super-tile's faces,
// so one round almost always): with 256 slots its accumulators made the workgroup 32 KB of LDS = 5 workgroups per CU, and the kernel is a
// chain of dependent round trips per tile (alpha -> list -> bbox -> record, ~15 us) whose only cover is other workgroups.
#ifndef RASTER_CAP2
#define RASTER_CAP2 128
#endif
template <int MODE> constexpr int stage_cap() { return MODE == 2 ? RASTER_CAP2 : kStage; }
// BWD (MODE 1 only): the camera-view pass also runs the silhouette backward of its own tile (see the end of raster_tile)
template <int MODE, bool BWD = false>
struct RasterSmem {
  static constexpr int kCap = stage_cap<MODE>();
  float4 s_a[kCap], s_b[kCap], s_bb[kCap];
  float s_z2[MODE == 2 ? 1 : kCap];
  int32_t s_id[kCap];
  int lds_cnt[4];
  unsigned long long zkey[MODE <= 1 ? 256 : 1];   // face-scan walk (MODE 0 / 1): per pixel min over (depth bits << 32 | face id)
  // dense scan: the staged faces with pixels in this tile, compacted: start slot of each in the tile's (face, pixel) slot sequence (+ the
  // total behind the last), and staged index | tile-local bbox origin | bbox width - 1
  int s_pre[MODE <= 1 ? kStage + 1 : 1];
  unsigned s_fb[MODE == 0 ? kStage : 1];           // (MODE 1: the pair rings' 1 KB, which only the soft pass behind the scan uses)
  // MODE 1, soft silhouette: sat = some face covers the pixel deeper than the sigmoid's float32 range (alpha = 1 exactly);
  // prodl = running product of (1 - p) over the other faces within the blur radius, in ascending face order; cand = pixels not (yet) saturated
  // (bytes; the fused-backward form re-uses the array as 256 floats)
  typename std::conditional<BWD, int, unsigned char>::type sat[MODE == 1 ? 256 : 1];
  float prodl[MODE == 1 ? 256 : 1];
  unsigned char cand[MODE == 1 ? 256 : 1];
  float ndc_x[kTile], ndc_y[kTile];               // pixel-centre NDC of the tile's columns / rows (pix_to_ndc holds an IEEE division)
  // pixel-centric pair walk (MODE 2): the tile's rim pixels (coordinates, P = 1 - alpha, upstream gradient), compacted, and the
  // (rim pixel, staged face) pairs whose pixel lies in the face's bbox
  float rp_x[MODE == 2 ? 256 : 1], rp_y[MODE == 2 ? 256 : 1], rp_P[MODE == 2 ? 256 : 1], rp_g[MODE == 2 ? 256 : 1];
  alignas(4) unsigned short pairs[MODE >= 1 ? 512 : 1];      // 4 waves x 128-entry ring of (pixel, staged face) pairs
  // MODE 2: per-staged-face gradient accumulators (x,y of 3 verts).  double: ds_add_f64 is ~20x faster than ds_add_f32 on gfx950
  double s_g[MODE == 2 ? kCap : 1][6];
  // MODE 1 + BWD: the same accumulators as float (6 KB instead of 12: the forward pass lives on its occupancy) — a few dozen pairs per tile add
  // into them through an fp32 compare-and-swap loop (11 clk per wave instruction; ds_add_f32 takes 193 on gfx950)
  float s_gf[(MODE == 1 && BWD) ? kCap : 1][6];
  float red[4];
};

__device__ __forceinline__ void lds_add_f32(float* p, float v) {
  unsigned* u = reinterpret_cast<unsigned*>(p);
  unsigned old = *u, assumed;
  do {
    assumed = old;
    old = atomicCAS(u, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}

// One 16x16 tile; `vblock` = index in the 1-D heaviest-first tile grid (harp_common.h: tile_decode_v).  A __device__ function so that
// the silhouette backward (MODE 2) can also run as part of the fused backward launch (shade_bwd.hip), interleaved with the shading tiles.
template <int MODE, bool BWD = false>
__device__ __forceinline__ void raster_tile(RasterSmem<MODE, BWD>& sm, unsigned vblock, const FaceRec* __restrict__ recs, const float4* __restrict__ bbs,
                                                     const int32_t* __restrict__ bins,
                                                     const int32_t* __restrict__ bin_count, const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ nact, int B, int F, int S, int nsx,
                                                     float blur, float sigma, int32_t* __restrict__ face_id,
                                                     float* __restrict__ zbuf, float* __restrict__ alpha,
                                                     const float* __restrict__ g_alpha, const int32_t* __restrict__ faces,
                                                     int V, float* __restrict__ g_ndc, const float* __restrict__ l1_target,
                                                     const int32_t* __restrict__ l1_fid, const float* __restrict__ l1_w,
                                                     float* __restrict__ l1_loss, float* __restrict__ l1_grad, float l1_inv, int sparse,
                                                     const float* __restrict__ l1_bg_sums, int32_t* __restrict__ st_state = nullptr) {
  auto& s_a = sm.s_a; auto& s_b = sm.s_b; auto& s_bb = sm.s_bb; auto& s_z2 = sm.s_z2; auto& s_id = sm.s_id;
  auto& s_g = sm.s_g;
  int* lds_cnt = sm.lds_cnt;

  int b, st, tx0, ty0, sub;
  const int kind = tile_decode_v(vblock, order, nact, B, nsx, S, b, st, tx0, ty0, sub, MODE != 2);   // 1-D grid in heaviest-first order (harp_common.h)
  if (kind == 0) return;
  if (kind == 2) {
    // super-tile without a single face.  Backward: nothing to do.  Forward: its first workgroup writes the empty-pixel outputs (and
    // the fused silhouette L1 against alpha = 0) for all 64x64 pixels, the other 15 leave at once.
    if (MODE == 2 || sub != 0) return;
    // depth pass of a caller that keeps ONE depth map across calls (st_state, harp_rasterize_fwd_keep): a super-tile that was empty — and
    // therefore filled with -1 — the last time as well is left alone; 3/4 of the map are such super-tiles (25 MB of writes per step)
    // (decided once per workgroup, behind a barrier: thread 0 stores the state below, and a wave that read it after that store would leave
    // without clearing its quarter of the super-tile)
    if (MODE == 0 && sparse && st_state && __syncthreads_or(st_state[b * nst_of(nsx) + st] == 1)) return;
    if (MODE == 1 && sparse && l1_target && l1_bg_sums) {
      // nothing to write, and the loss of an un-rendered super-tile against a static target is a constant: one table look-up
      if (threadIdx.x == 0) {
        const float sum = l1_bg_sums[(size_t)l1_fid[b] * nst_of(nsx) + st];
        if (sum != 0.f) atomicAdd(l1_loss, sum * l1_inv);
      }
      return;
    }
    float acc = 0.f;
    float tg[16];
    const bool l1 = (MODE == 1) && l1_target != nullptr;
    const float* trow = l1 ? l1_target + (size_t)l1_fid[b] * S * S : nullptr;
#pragma unroll
    for (int k = 0; k < 16; ++k) {            // all 16 target loads in flight before the first store
      int xi, yi;
      supertile_pixel(k, tx0, ty0, xi, yi);
      tg[k] = (l1 && xi < S && yi < S) ? trow[(size_t)yi * S + xi] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int xi, yi;
      supertile_pixel(k, tx0, ty0, xi, yi);
      if (xi < S && yi < S) {
        const size_t o = ((size_t)b * S + yi) * S + xi;
        if (l1) acc += fabsf(tg[k]);                       // |alpha - y| with alpha = 0
        if (zbuf) zbuf[o] = -1.0f;                          // depth maps stay complete: shadow taps may land one pixel outside a face's super-tile
        if (!sparse) {
          if (face_id) face_id[o] = -1;
          if (MODE == 1) {
            alpha[o] = 0.f;
            if (l1) l1_grad[o] = l1_w[0] * l1_inv * (float)((0.f > tg[k]) - (0.f < tg[k]));
          }
        }
      }
    }
    if (MODE == 1 && l1_target) {
      const float sum = block_sum_256(acc, sm.red);
      if (threadIdx.x == 0 && sum != 0.f) atomicAdd(l1_loss, sum * l1_inv);
    }
    if (MODE == 0 && st_state && threadIdx.x == 0) st_state[b * nst_of(nsx) + st] = 1;               // all -1 from now on
    return;
  }
  const int nst = nsx * nsx;
  if (MODE == 0 && st_state && sub == 0 && threadIdx.x == 0) st_state[b * nst + st] = 0;               // holds depths: to be cleared when it empties
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xi = tx0 + (lane & 15), yi = ty0 + w * 4 + (lane >> 4);
  const bool in_img = (xi < S) && (yi < S);
  const float px = pix_to_ndc(xi, S), py = pix_to_ndc(yi, S);
  const int n = bin_count[b * nst + st];
  const int32_t* list = bins + ((size_t)b * nst + st) * F;
  const FaceRec* rb = recs + (size_t)b * F;
  const float4* bbb = bbs + (size_t)b * F;

  // tile and wave-strip bounds in NDC (conservative supersets of the per-pixel bbox test)
  const float t_xhi = pix_to_ndc(tx0, S), t_xlo = pix_to_ndc(min(tx0 + kTile, S) - 1, S);
  const float t_yhi = pix_to_ndc(ty0, S), t_ylo = pix_to_ndc(min(ty0 + kTile, S) - 1, S);

  float best_z = 3.0e38f;
  int best_f = -1;
  float prod = 1.0f;
  float P = 0.f, ga = 0.f;
  bool need = in_img;
  if (MODE == 2) {
    if (in_img) {
      const size_t o = ((size_t)b * S + yi) * S + xi;
      P = 1.0f - alpha[o];
      ga = g_alpha[o];
    }
    // P == 1 exactly: alpha == 0, i.e. NO face lies within the blur radius of this pixel (a face that does has p >= sigmoid(-9.21) = 1e-4,
    // which leaves prod <= 0.9999 in float32): every term of its gradient is exactly zero, as it is for P == 0 (saturated).  Only the true
    // rim takes part in the pair walk, not the background pixels of every tile near the mesh (exact; measured: no time, the kernel's cost is
    // the ~4 000 tiles that do hold rim pixels and stage their faces again)
    need = in_img && (P != 0.f) && (P != 1.0f) && (ga != 0.f);
    // whole tile saturated / no upstream gradient -> nothing to do
    if (__syncthreads_or(need ? 1 : 0) == 0) return;
  }
  constexpr bool kPix = (MODE == 2);
  int npx = 0;
  if constexpr (kPix) {
    const int slot = block_compact(need, 0, lds_cnt, npx);
    if (slot >= 0) { sm.rp_x[slot] = px; sm.rp_y[slot] = py; sm.rp_P[slot] = P; sm.rp_g[slot] = ga; }
  }

  const float inv_sigma = 1.0f / sigma;
  constexpr bool kScan = (MODE <= 1);
  if (kScan) {                                  // (the first barrier of the staging below orders these before their first use)
    sm.zkey[threadIdx.x] = ~0ull;
    if (MODE == 1) { sm.sat[threadIdx.x] = 0; sm.prodl[threadIdx.x] = 1.0f; }
    if (threadIdx.x < kTile) sm.ndc_x[threadIdx.x] = pix_to_ndc(tx0 + threadIdx.x, S);
    else if (threadIdx.x < 2 * kTile) sm.ndc_y[threadIdx.x - kTile] = pix_to_ndc(ty0 + threadIdx.x - kTile, S);
  }
  auto stage_write = [&](int pos, int id, const float4& bb) {
    if (pos >= 0) {
      const FaceRec r = rb[id];
      s_a[pos] = r.a; s_b[pos] = r.b; s_bb[pos] = bb; s_id[pos] = id;
      if (MODE != 2) s_z2[pos] = r.c.x;
    }
  };
  // The super-tile's list is filtered against this 16x16 tile 256 entries at a time and the hits ACCUMULATE in the staging arrays:
  // a walk runs when the next chunk would not fit, or at the end of the list — one walk per tile almost always (a tile sees 30 - 100
  // of its super-tile's ~350 faces), instead of one per 256 list entries.
  int staged = 0, skip = 0;                    // skip: hits of the current chunk staged in an earlier round (a chunk with more hits than a round holds)
  int rounds = 0, last_nl = 0;                 // (BWD: a tile that took ONE round still holds all its faces in LDS when the forward pass is done)
  for (int base = 0; base < n || staged > 0;) {
    bool flush = true;
    if (base < n) {
      // ---- stage: filter this chunk's list entries against the tile, compact into LDS behind what is already staged
      const int e = base + threadIdx.x;
      bool hit = false;
      int id = 0;
      float4 bb;
      if (e < n) {
        id = list[e];
        bb = bbb[id];          // contiguous 16-B bbox array (4 per 64-B line) instead of the 64-B-strided records
        hit = !(t_xlo > bb.y || t_xhi < bb.x || t_ylo > bb.w || t_yhi < bb.z);
      }
      int cnt;
      const int posc = block_compact(hit, 0, lds_cnt, cnt);       // rank among this chunk's hits
      const int rem = cnt - skip, room = stage_cap<MODE>() - staged;
      if (rem <= room) {
        stage_write((posc >= skip) ? staged + posc - skip : -1, id, bb);
        staged += rem;
        skip = 0;
        base += kStage;
        flush = base >= n;
      } else if (staged == 0) {
        // more hits in one chunk than a round holds (only with the 128-slot rounds of MODE 2, on tiles with > 128 faces): take what
        // fits, walk, come back to the same chunk for the rest
        stage_write((posc >= skip && posc < skip + room) ? posc - skip : -1, id, bb);
        staged = room;
        skip += room;
      }
    }
    if (!flush) continue;
    const int nl = staged;
    staged = 0;
    if (nl == 0) continue;
    ++rounds; last_nl = nl;
    if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 6; ++c)
        if ((int)threadIdx.x < stage_cap<MODE>()) s_g[threadIdx.x][c] = 0.0;
    }
    __syncthreads();
    if constexpr (kScan) {
      // ---- face scan (hard K = 1 pass): the nearest face of a pixel is a 64-bit LDS min over (depth bits, face id) — the depth expression and
      //      the tie-break (lower id) of PyTorch3D's K = 1 rasterisation.
      // DENSE PACKING: the walk visits (staged face, pixel of its bbox clipped to the tile) pairs, one pair per lane, every lane busy: the
      // pairs are numbered face after face (a prefix sum over the faces' clipped bbox sizes), a wave owns a contiguous quarter of the
      // numbers and tracks the face its first lane is in; the starts of the next 64 faces sit in one register across the wave, and a lane's
      // face is the number of those starts at or below its slot.  (The blocked walk spent 3.7 blocks x 16 lanes per (face, tile) for a
      // bbox of ~25 pixels.)
      const float hs = 0.5f * (float)S;
      unsigned* s_fb = (MODE == 1) ? reinterpret_cast<unsigned*>(sm.pairs) : sm.s_fb;
      int nf, T;
      {
        int cnt = 0;
        unsigned box = 0;
        if ((int)threadIdx.x < nl) {
          const float4 q = s_bb[threadIdx.x];
          // pixel-centre columns / rows that can lie in the bbox (pixel coordinate of NDC n: (1 - n) S / 2 - 1/2; 1e-3 px of slack,
          // the exact comparison below decides), clipped to the tile
          const int x0 = max((int)ceilf((1.0f - q.y) * hs - 0.501f), tx0), x1 = min(min((int)floorf((1.0f - q.x) * hs - 0.499f), tx0 + kTile - 1), S - 1);
          const int y0 = max((int)ceilf((1.0f - q.w) * hs - 0.501f), ty0), y1 = min(min((int)floorf((1.0f - q.z) * hs - 0.499f), ty0 + kTile - 1), S - 1);
          if (x0 <= x1 && y0 <= y1) {
            cnt = (x1 - x0 + 1) * (y1 - y0 + 1);                                                 // 1 .. 256
            box = (unsigned)(x0 - tx0) | ((unsigned)(y0 - ty0) << 4) | ((unsigned)(x1 - x0) << 8);
          }
        }
        // block-wide exclusive scan of (pixels | faces-with-pixels << 20)
        const int own = cnt + ((cnt > 0) << 20);
        int v = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int u = __shfl_up(v, d);
          if (lane >= d) v += u;
        }
        if (lane == 63) lds_cnt[w] = v;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int c = lds_cnt[i]; tot += c; if (i < w) off += c; }
        const int excl = v + off - own;
        nf = tot >> 20; T = tot & 0xfffff;
        if (cnt > 0) {
          sm.s_pre[excl >> 20] = excl & 0xfffff;
          s_fb[excl >> 20] = threadIdx.x | (box << 8);
        }
        if (threadIdx.x == 0) sm.s_pre[nf] = T;
        __syncthreads();
      }
      if (T > 0) {
        const int per = (((T + 3) >> 2) + 63) & ~63;
        const int s_lo = w * per, s_end = min(T, s_lo + per);
        if (s_lo < s_end) {
          // the face that holds this wave's first slot: the number of starts <= s_lo, minus one
          int kb = -1;
#pragma unroll
          for (int j = 0; j < kStage / 64; ++j) {
            const int idx = j * 64 + lane;
            kb += __popcll(__ballot(idx < nf && sm.s_pre[idx] <= s_lo));
          }
          int pkb = sm.s_pre[kb];
          const float K18 = 18.0f * sigma;
          for (int sb = s_lo; sb < s_end; sb += 64) {
            const int sl = sb + lane;
            const int pw = sm.s_pre[min(kb + 1 + lane, nf)];             // lane j: the start of face kb + 1 + j (the total behind the last)
            int k = kb, pk = pkb;
            const int lim = min(sb + 64, s_end);
            for (int j = 0; j < 64; ++j) {
              const int p = __builtin_amdgcn_readlane(pw, j);
              if (p >= lim) break;
              if (sl >= p) { ++k; pk = p; }
              ++kb; pkb = p;
            }
            if (sl < s_end) {
              const unsigned fb = s_fb[k];
              const int kk = fb & 255, wd = (int)((fb >> 16) & 15) + 1;
              const int r = sl - pk;
              const int ry = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)wd));     // r / wd ((r + 1/2) / wd is >= 1/32 from an integer)
              const int xl = (int)((fb >> 8) & 15) + r - ry * wd, yl = (int)((fb >> 12) & 15) + ry;
              const float qx = sm.ndc_x[xl], qy = sm.ndc_y[yl];
              const float4 q = s_bb[kk];
              const Tri t = tri_from(s_a[kk], s_b[kk], make_float4(s_z2[kk], 0.f, 0.f, 0.f));
              const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
              const float sg = (area > 0.f) ? 1.f : ((area < 0.f) ? -1.f : 0.f);
              // (same predicate as !(qx > q.y || qx < q.x || qy > q.w || qy < q.z) for finite operands, as two median-of-three
              //  instructions + two compares instead of four compares and the mask arithmetic that joins them)
              const bool inbox = __builtin_amdgcn_fmed3f(qx, q.x, q.y) == qx && __builtin_amdgcn_fmed3f(qy, q.z, q.w) == qy;
              const float e0 = edge_fn(qx, qy, t.x1, t.y1, t.x2, t.y2);
              const float e1 = edge_fn(qx, qy, t.x2, t.y2, t.x0, t.y0);
              const float e2 = edge_fn(qx, qy, t.x0, t.y0, t.x1, t.y1);
              if (inbox && (e0 * sg > 0.f) && (e1 * sg > 0.f) && (e2 * sg > 0.f)) {
                // BarycentricCoordsForward + PerspectiveCorrection with reciprocals (<= 2 ulp from the IEEE-division form)
                const int pix = yl * kTile + xl;
                if (MODE != 1 || face_id) {        // (camera view without face ids — the geometry-only stage: the silhouette needs no nearest face)
                  const float ra = __builtin_amdgcn_rcpf(area);
                  const float t0 = (e0 * ra) * t.z1 * t.z2, t1 = t.z0 * (e1 * ra) * t.z2, t2 = t.z0 * t.z1 * (e2 * ra);
                  const float rd = __builtin_amdgcn_rcpf(fmaxf(t0 + t1 + t2, kEps));
                  const float pz = (t0 * rd) * t.z0 + (t1 * rd) * t.z1 + (t2 * rd) * t.z2;
                  if (pz >= 0.f && pz < 3.0e38f) atomicMin(&sm.zkey[pix], ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)s_id[kk]);
                }
                if (MODE == 1) {
                  // deeper than sqrt(18 sigma) inside every edge LINE (<= segment distance): sigmoid saturates to exactly 1 in
                  // fp32 (exp(-18) < 2^-24), so the factor (1-p) is exactly 0 — the pixel's alpha is 1 whatever the other faces do
                  const float l12 = (t.x2 - t.x1) * (t.x2 - t.x1) + (t.y2 - t.y1) * (t.y2 - t.y1);
                  const float l20 = (t.x0 - t.x2) * (t.x0 - t.x2) + (t.y0 - t.y2) * (t.y0 - t.y2);
                  const float l01 = (t.x1 - t.x0) * (t.x1 - t.x0) + (t.y1 - t.y0) * (t.y1 - t.y0);
                  if (e0 * e0 > K18 * l12 && e1 * e1 > K18 * l20 && e2 * e2 > K18 * l01) sm.sat[pix] = 1;
                }
              }
            }
          }
        }
      }
      if constexpr (MODE == 1) {
        // ---- soft silhouette of the pixels no face saturates: pair walk over (candidate pixel, staged face), as in the backward pass.
        //      A wave owns a contiguous quarter of the candidates, so a pixel's pairs stay in one wave, in ascending face order:
        //      the factors of a 64-pair batch are multiplied into the pixel's running product by the first lane of the pixel's run,
        //      one after the other — the same sequence of float multiplications as a sequential walk over the faces.
        __syncthreads();                                   // saturation flags of this round are complete
        int ncand;
        const int slot = block_compact(in_img && sm.sat[threadIdx.x] == 0, 0, lds_cnt, ncand);
        if (slot >= 0) sm.cand[slot] = (unsigned char)threadIdx.x;
        __syncthreads();
        if (ncand > 0) {
          unsigned short* wl = sm.pairs + w * 128;
          const int cper = (ncand + 3) >> 2, c_lo = min(ncand, w * cper), c_hi = min(ncand, (w + 1) * cper);
          // the wave's candidates are a band of the tile's rows (the list is in pixel order): only the staged faces whose bbox reaches the band
          // are paired with them — a wave-local sub-list, in the staged (= ascending face) order, in the scan's dead prefix array
          unsigned char* sub = reinterpret_cast<unsigned char*>(sm.s_pre) + w * 256;
          int nlw = 0;
          if (c_lo < c_hi) {
            const float ya = sm.ndc_y[sm.cand[c_lo] >> 4], yb = sm.ndc_y[sm.cand[c_hi - 1] >> 4];
            const float by_lo = fminf(ya, yb), by_hi = fmaxf(ya, yb);
            for (int k0 = 0; k0 < nl; k0 += 64) {
              const int k = k0 + lane;
              bool hit = false;
              if (k < nl) { const float4 q = s_bb[k]; hit = !(by_lo > q.w || by_hi < q.z); }
              const unsigned long long m = __ballot(hit);
              if (hit) sub[nlw + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)k;
              nlw += __popcll(m);
            }
          }
          const int i_end = (c_hi - c_lo) * nlw;
          const float inv_nl = 1.0f / (float)max(nlw, 1);
          int head = 0, tail = 0;
          auto process = [&](int nvalid) {
            float f = 1.0f;
            int c = -1 - lane;                             // (distinct keys for idle lanes)
            if (lane < nvalid) {
              const int pr = wl[(head + lane) & 127], j = pr & 255;
              c = pr >> 8;
              const float qx = sm.ndc_x[c & (kTile - 1)], qy = sm.ndc_y[c >> 4];
              const Tri t = tri_from(s_a[j], s_b[j], make_float4(0.f, 0.f, 0.f, 0.f));
              const float e0 = edge_fn(qx, qy, t.x1, t.y1, t.x2, t.y2);
              const float e1 = edge_fn(qx, qy, t.x2, t.y2, t.x0, t.y0);
              const float e2 = edge_fn(qx, qy, t.x0, t.y0, t.x1, t.y1);
              const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
              const float sg = (area > 0.f) ? 1.f : ((area < 0.f) ? -1.f : 0.f);
              const float e0s = e0 * sg, e1s = e1 * sg, e2s = e2 * sg;
              const bool inside = (e0s > 0.f) && (e1s > 0.f) && (e2s > 0.f);
              const float l12 = (t.x2 - t.x1) * (t.x2 - t.x1) + (t.y2 - t.y1) * (t.y2 - t.y1);
              const float l20 = (t.x0 - t.x2) * (t.x0 - t.x2) + (t.y0 - t.y2) * (t.y0 - t.y2);
              const float l01 = (t.x1 - t.x0) * (t.x1 - t.x0) + (t.y1 - t.y0) * (t.y1 - t.y0);
              bool soft = true;
              if (!inside) {                               // outside: the triangle lies beyond the line of any violated edge, so dist >= that line distance
                const float Bf = blur * 1.00001f;
                if ((e0s < 0.f && e0 * e0 >= Bf * l12) || (e1s < 0.f && e1 * e1 >= Bf * l20) || (e2s < 0.f && e2 * e2 >= Bf * l01)) soft = false;
              }
              if (soft) {
                float ta, tb, tc;
                const float d01 = seg_dist2(qx, qy, t.x0, t.y0, t.x1, t.y1, ta);
                const float d02 = seg_dist2(qx, qy, t.x0, t.y0, t.x2, t.y2, tb);
                const float d12 = seg_dist2(qx, qy, t.x1, t.y1, t.x2, t.y2, tc);
                const float dist = fminf(d01, fminf(d02, d12));
                if (inside || dist < blur) {
                  const float sd = inside ? -dist : dist;
                  const float p = __builtin_amdgcn_rcpf(1.0f + __expf(sd * inv_sigma));   // sigmoid(-sd/sigma): fast exp + reciprocal (rel. error ~1e-6 at |x| ~ 18; image tolerance 1e-4)
                  f = 1.0f - p;
                }
              }
            }
            const int cprev = __shfl_up(c, 1);
            const bool first = (lane < nvalid) && (lane == 0 || cprev != c);
            const unsigned long long fm = __ballot((lane >= nvalid) || lane == 0 || cprev != c);      // run starts (idle lanes are runs of their own)
            int len = 0;
            float pacc = 1.0f;
            if (first) {
              const unsigned long long above = (lane == 63) ? 0ull : (fm >> (lane + 1));
              len = above ? __ffsll((unsigned long long)above) : 64 - lane;
              pacc = sm.prodl[c];
            }
            for (int r = 0; __any(r < len); ++r) {
              const float fr = __shfl(f, (lane + r) & 63);
              if (r < len) pacc *= fr;
            }
            if (first) sm.prodl[c] = pacc;
          };
          for (int i0 = 0; i0 < i_end; i0 += 64) {
            const int i = i0 + lane;
            bool pass = false;
            int c = 0, k = 0;
            if (i < i_end) {
              int ci = (int)(((float)i + 0.5f) * inv_nl);
              k = i - ci * nlw;
              if (k < 0) { --ci; k += nlw; } else if (k >= nlw) { ++ci; k -= nlw; }
              c = sm.cand[c_lo + ci];
              k = sub[k];
              const float4 q = s_bb[k];
              const float qx = sm.ndc_x[c & (kTile - 1)], qy = sm.ndc_y[c >> 4];
              pass = !(qx > q.y || qx < q.x || qy > q.w || qy < q.z);
            }
            const unsigned long long m = __ballot(pass);
            if (pass) wl[(tail + __popcll(m & ((1ull << lane) - 1ull))) & 127] = (unsigned short)((c << 8) | k);
            tail += __popcll(m);
            if (tail - head >= 64) { process(64); head += 64; }
          }
          if (tail > head) process(tail - head);
        }
      }
    } else if constexpr (kPix) {
      // ---- pair walk (silhouette backward): only the tile's rim pixels (P != 0 and an upstream gradient: ~20 of 256) take part.
      //      Phase 1 tests every (rim pixel, staged face) pair against the face's bbox, one pair per lane, and compacts the ~4 hits
      //      per pixel into an LDS list; phase 2 does the distance / sigmoid / gradient arithmetic of the strip walk below on that
      //      list, one pair per lane, every lane busy.  (The strip walk classified each face of a strip's hit list on all 64 lanes.)
      const int npairs = npx * nl;
      const float inv_nl = 1.0f / (float)nl;
      // every wave walks its own quarter of the pairs and keeps its own hit list (a 128-entry ring in LDS, head / tail in SGPRs):
      // no barrier and no atomic between the two phases — a wave only reads back what it wrote, and LDS operations of a wave are in order
      unsigned short* wl = sm.pairs + w * 128;
      const int per = (npairs + 3) >> 2, i_end = min(npairs, (w + 1) * per);
      int head = 0, tail = 0;
      auto process = [&](int nvalid) {
        if (lane < nvalid) {
          const int pr = wl[(head + lane) & 127], c = pr >> 8, j = pr & 255;
          const float qx = sm.rp_x[c], qy = sm.rp_y[c], Pq = sm.rp_P[c], gq = sm.rp_g[c];
          const Tri t = tri_from(s_a[j], s_b[j], make_float4(0.f, 0.f, 0.f, 0.f));
          const float e0 = edge_fn(qx, qy, t.x1, t.y1, t.x2, t.y2);
          const float e1 = edge_fn(qx, qy, t.x2, t.y2, t.x0, t.y0);
          const float e2 = edge_fn(qx, qy, t.x0, t.y0, t.x1, t.y1);
          const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
          const float sg = (area > 0.f) ? 1.f : ((area < 0.f) ? -1.f : 0.f);
          const float e0s = e0 * sg, e1s = e1 * sg, e2s = e2 * sg;
          const bool inside = (e0s > 0.f) && (e1s > 0.f) && (e2s > 0.f);
          const float l12 = (t.x2 - t.x1) * (t.x2 - t.x1) + (t.y2 - t.y1) * (t.y2 - t.y1);
          const float l20 = (t.x0 - t.x2) * (t.x0 - t.x2) + (t.y0 - t.y2) * (t.y0 - t.y2);
          const float l01 = (t.x1 - t.x0) * (t.x1 - t.x0) + (t.y1 - t.y0) * (t.y1 - t.y0);
          bool soft = true;
          if (inside) {
            const float K = 18.0f * sigma;               // saturated: the factor (1 - p) is exactly 0 and so is its gradient
            if (e0 * e0 > K * l12 && e1 * e1 > K * l20 && e2 * e2 > K * l01) soft = false;
          } else {
            const float Bf = blur * 1.00001f;            // beyond the blur radius of a violated edge line
            if ((e0s < 0.f && e0 * e0 >= Bf * l12) || (e1s < 0.f && e1 * e1 >= Bf * l20) || (e2s < 0.f && e2 * e2 >= Bf * l01)) soft = false;
          }
          if (soft) {
            float ta, tb, tc;
            const float d01 = seg_dist2(qx, qy, t.x0, t.y0, t.x1, t.y1, ta);
            const float d02 = seg_dist2(qx, qy, t.x0, t.y0, t.x2, t.y2, tb);
            const float d12 = seg_dist2(qx, qy, t.x1, t.y1, t.x2, t.y2, tc);
            const float dist = fminf(d01, fminf(d02, d12));
            if (inside || dist < blur) {
              const float sd = inside ? -dist : dist;
              const float p = __builtin_amdgcn_rcpf(1.0f + __expf(sd * inv_sigma));
              // d alpha / d sd = -P * p / sigma  (P = prod over all faces)
              const float g_sd = gq * (-Pq * p * inv_sigma);
              const float gd = inside ? -g_sd : g_sd;     // d/d(dist^2)
              // PointLineDistanceBackward on the argmin edge (t treated as constant)
              int ia, ib; float ax, ay, bx, by, tt;
              if (d01 <= d02 && d01 <= d12) { ia = 0; ib = 1; ax = t.x0; ay = t.y0; bx = t.x1; by = t.y1; tt = ta; }
              else if (d02 <= d12)          { ia = 0; ib = 2; ax = t.x0; ay = t.y0; bx = t.x2; by = t.y2; tt = tb; }
              else                          { ia = 1; ib = 2; ax = t.x1; ay = t.y1; bx = t.x2; by = t.y2; tt = tc; }
              const float hx = ax + tt * (bx - ax), hy = ay + tt * (by - ay);
              const float cx = gd * 2.f * (hx - qx), cy = gd * 2.f * (hy - qy);
              atomicAdd(&s_g[j][2 * ia], (double)((1.f - tt) * cx));
              atomicAdd(&s_g[j][2 * ia + 1], (double)((1.f - tt) * cy));
              atomicAdd(&s_g[j][2 * ib], (double)(tt * cx));
              atomicAdd(&s_g[j][2 * ib + 1], (double)(tt * cy));
            }
          }
        }
      };
      for (int i0 = w * per; i0 < i_end; i0 += 64) {
        const int i = i0 + lane;
        bool pass = false;
        int c = 0, k = 0;
        if (i < i_end) {
          c = (int)(((float)i + 0.5f) * inv_nl);
          k = i - c * nl;
          if (k < 0) { --c; k += nl; } else if (k >= nl) { ++c; k -= nl; }
          const float4 q = s_bb[k];
          const float qx = sm.rp_x[c], qy = sm.rp_y[c];
          pass = !(qx > q.y || qx < q.x || qy > q.w || qy < q.z);
        }
        const unsigned long long m = __ballot(pass);
        if (pass) wl[(tail + __popcll(m & ((1ull << lane) - 1ull))) & 127] = (unsigned short)((c << 8) | k);
        tail += __popcll(m);
        if (tail - head >= 64) { process(64); head += 64; }
      }
      if (tail > head) process(tail - head);
    }
    __syncthreads();
    if (MODE == 2 && (int)threadIdx.x < nl) {
      // flush: one global atomic per (staged face, vertex, component) per workgroup
      const int fid = s_id[threadIdx.x];
      float* gb = g_ndc + (size_t)b * V * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float gx = (float)s_g[threadIdx.x][2 * k], gy = (float)s_g[threadIdx.x][2 * k + 1];
        if (gx != 0.f || gy != 0.f) {
          const int v = faces[3 * fid + k];
          atomicAdd(gb + 3 * v, gx);
          atomicAdd(gb + 3 * v + 1, gy);
        }
      }
    }
    __syncthreads();
  }

  if constexpr (kScan) {
    // (the last round ended with a barrier) pixel of this lane = tile-local index threadIdx.x
    const unsigned long long key = sm.zkey[threadIdx.x];
    if (key != ~0ull) { best_f = (int)(unsigned)(key & 0xffffffffull); best_z = __uint_as_float((unsigned)(key >> 32)); }
    if (MODE == 1) prod = sm.sat[threadIdx.x] ? 0.f : sm.prodl[threadIdx.x];
  }
  float l1_acc = 0.f;
  if (MODE != 2 && in_img) {
    const size_t o = ((size_t)b * S + yi) * S + xi;
    if (face_id) face_id[o] = best_f;
    if (zbuf) zbuf[o] = (best_f >= 0) ? best_z : -1.0f;
    if (MODE == 1) {
      const float a = 1.0f - prod;
      alpha[o] = a;
      if (l1_target) {
        // fused torch.nn.L1Loss(y_sil_true, y_sil_pred) (optimize_sequence.py:519) and its gradient w.r.t. alpha
        const float d = a - l1_target[((size_t)l1_fid[b] * S + yi) * S + xi];
        l1_acc = fabsf(d);
        l1_grad[o] = l1_w[0] * l1_inv * ((d > 0.f) - (d < 0.f));
      }
    }
  }
  if (MODE == 1 && l1_target) {
    const float sum = block_sum_256(l1_acc, sm.red);
    if (threadIdx.x == 0 && sum != 0.f) atomicAdd(l1_loss, sum * l1_inv);
  }
  if constexpr (MODE == 1 && BWD) {
    // ---- FUSED SILHOUETTE BACKWARD (rasterize_meshes_backward, dists path + sigmoid_alpha_blend's backward; the stand-alone form is MODE 2).
    //      The pixel's alpha is final, the fused L1 has just formed d loss / d alpha, and the tile's faces are still staged in LDS: the rim
    //      pixels walk them again right here instead of in a second launch that re-reads alpha and the gradient image and stages the same
    //      faces through the same list -> bbox -> record chain (~4 000 rim tiles per camera view: 50 us alone, 117 us beside the shader
    //      backward, which it slowed by 30).  Same pairs, same arithmetic as MODE 2; the per-face sums are float instead of double.
    //      A tile that needed more than one staging round (> 256 faces: dense, minified meshes) stages its rounds once more.
    if (g_ndc != nullptr && l1_target != nullptr) {
      float Pq = 0.f, gq = 0.f;
      if (in_img) {
        const float a = 1.0f - prod;
        const float d = a - l1_target[((size_t)l1_fid[b] * S + yi) * S + xi];
        Pq = 1.0f - a;                                   // (as the stand-alone pass forms it from the stored alpha)
        gq = l1_w[0] * l1_inv * ((d > 0.f) - (d < 0.f));
      }
      const bool rim = in_img && (Pq != 0.f) && (Pq != 1.0f) && (gq != 0.f);
      if (__syncthreads_or(rim ? 1 : 0) != 0) {
        // the pixel's forward state is in registers: its LDS (min keys, saturation flags, running products) becomes the rim-pixel list
        float* rp_x = reinterpret_cast<float*>(sm.zkey);
        float* rp_y = rp_x + 256;
        float* rp_P = reinterpret_cast<float*>(sm.sat);
        float* rp_g = sm.prodl;
        int npx;
        const int slot = block_compact(rim, 0, lds_cnt, npx);
        if (slot >= 0) { rp_x[slot] = px; rp_y[slot] = py; rp_P[slot] = Pq; rp_g[slot] = gq; }
        auto bwd_round = [&](int nl) {
#pragma unroll
          for (int c = 0; c < 6; ++c) sm.s_gf[threadIdx.x][c] = 0.f;
          __syncthreads();
          const int npairs = npx * nl;
          const float inv_nl = 1.0f / (float)nl;
          unsigned short* wl = sm.pairs + w * 128;
          const int per = (npairs + 3) >> 2, i_end = min(npairs, (w + 1) * per);
          int head = 0, tail = 0;
          auto process = [&](int nvalid) {
            if (lane < nvalid) {
              const int pr = wl[(head + lane) & 127], c = pr >> 8, j = pr & 255;
              const float qx = rp_x[c], qy = rp_y[c], Pc = rp_P[c], gc = rp_g[c];
              const Tri t = tri_from(s_a[j], s_b[j], make_float4(0.f, 0.f, 0.f, 0.f));
              const float e0 = edge_fn(qx, qy, t.x1, t.y1, t.x2, t.y2);
              const float e1 = edge_fn(qx, qy, t.x2, t.y2, t.x0, t.y0);
              const float e2 = edge_fn(qx, qy, t.x0, t.y0, t.x1, t.y1);
              const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
              const float sg = (area > 0.f) ? 1.f : ((area < 0.f) ? -1.f : 0.f);
              const float e0s = e0 * sg, e1s = e1 * sg, e2s = e2 * sg;
              const bool inside = (e0s > 0.f) && (e1s > 0.f) && (e2s > 0.f);
              const float l12 = (t.x2 - t.x1) * (t.x2 - t.x1) + (t.y2 - t.y1) * (t.y2 - t.y1);
              const float l20 = (t.x0 - t.x2) * (t.x0 - t.x2) + (t.y0 - t.y2) * (t.y0 - t.y2);
              const float l01 = (t.x1 - t.x0) * (t.x1 - t.x0) + (t.y1 - t.y0) * (t.y1 - t.y0);
              bool soft = true;
              if (inside) {
                const float K = 18.0f * sigma;               // saturated: the factor (1 - p) is exactly 0 and so is its gradient
                if (e0 * e0 > K * l12 && e1 * e1 > K * l20 && e2 * e2 > K * l01) soft = false;
              } else {
                const float Bf = blur * 1.00001f;            // beyond the blur radius of a violated edge line
                if ((e0s < 0.f && e0 * e0 >= Bf * l12) || (e1s < 0.f && e1 * e1 >= Bf * l20) || (e2s < 0.f && e2 * e2 >= Bf * l01)) soft = false;
              }
              if (soft) {
                float ta, tb, tc;
                const float d01 = seg_dist2(qx, qy, t.x0, t.y0, t.x1, t.y1, ta);
                const float d02 = seg_dist2(qx, qy, t.x0, t.y0, t.x2, t.y2, tb);
                const float d12 = seg_dist2(qx, qy, t.x1, t.y1, t.x2, t.y2, tc);
                const float dist = fminf(d01, fminf(d02, d12));
                if (inside || dist < blur) {
                  const float sd = inside ? -dist : dist;
                  const float p = __builtin_amdgcn_rcpf(1.0f + __expf(sd * inv_sigma));
                  const float g_sd = gc * (-Pc * p * inv_sigma);          // d alpha / d sd = -P * p / sigma
                  const float gd = inside ? -g_sd : g_sd;                 // d / d(dist^2)
                  int ia, ib; float ax, ay, bx, by, tt;                   // PointLineDistanceBackward on the argmin edge (t treated as constant)
                  if (d01 <= d02 && d01 <= d12) { ia = 0; ib = 1; ax = t.x0; ay = t.y0; bx = t.x1; by = t.y1; tt = ta; }
                  else if (d02 <= d12)          { ia = 0; ib = 2; ax = t.x0; ay = t.y0; bx = t.x2; by = t.y2; tt = tb; }
                  else                          { ia = 1; ib = 2; ax = t.x1; ay = t.y1; bx = t.x2; by = t.y2; tt = tc; }
                  const float hx = ax + tt * (bx - ax), hy = ay + tt * (by - ay);
                  const float cx = gd * 2.f * (hx - qx), cy = gd * 2.f * (hy - qy);
                  lds_add_f32(&sm.s_gf[j][2 * ia], (1.f - tt) * cx);
                  lds_add_f32(&sm.s_gf[j][2 * ia + 1], (1.f - tt) * cy);
                  lds_add_f32(&sm.s_gf[j][2 * ib], tt * cx);
                  lds_add_f32(&sm.s_gf[j][2 * ib + 1], tt * cy);
                }
              }
            }
          };
          for (int i0 = w * per; i0 < i_end; i0 += 64) {
            const int i = i0 + lane;
            bool pass = false;
            int c = 0, k = 0;
            if (i < i_end) {
              c = (int)(((float)i + 0.5f) * inv_nl);
              k = i - c * nl;
              if (k < 0) { --c; k += nl; } else if (k >= nl) { ++c; k -= nl; }
              const float4 q = s_bb[k];
              const float qx = rp_x[c], qy = rp_y[c];
              pass = !(qx > q.y || qx < q.x || qy > q.w || qy < q.z);
            }
            const unsigned long long m = __ballot(pass);
            if (pass) wl[(tail + __popcll(m & ((1ull << lane) - 1ull))) & 127] = (unsigned short)((c << 8) | k);
            tail += __popcll(m);
            if (tail - head >= 64) { process(64); head += 64; }
          }
          if (tail > head) process(tail - head);
          __syncthreads();
          if ((int)threadIdx.x < nl) {          // flush: one memory atomic per (staged face, vertex, component) with a gradient
            const int fid = s_id[threadIdx.x];
            float* gb = g_ndc + (size_t)b * V * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const float gx = sm.s_gf[threadIdx.x][2 * k], gy = sm.s_gf[threadIdx.x][2 * k + 1];
              if (gx != 0.f || gy != 0.f) {
                const int v = faces[3 * fid + k];
                atomicAdd(gb + 3 * v, gx);
                atomicAdd(gb + 3 * v + 1, gy);
              }
            }
          }
        };
        if (rounds == 1) {
          bwd_round(last_nl);
        } else {
          // more than one round: stage the list again, round by round (same filter, same order as the forward loop above)
          int st2 = 0, sk2 = 0;
          for (int base = 0; base < n || st2 > 0;) {
            bool flush = true;
            __syncthreads();                    // the previous round's flush is done with the staged ids
            if (base < n) {
              const int e = base + threadIdx.x;
              bool hit = false;
              int id = 0;
              float4 bb;
              if (e < n) {
                id = list[e];
                bb = bbb[id];
                hit = !(t_xlo > bb.y || t_xhi < bb.x || t_ylo > bb.w || t_yhi < bb.z);
              }
              int cnt;
              const int posc = block_compact(hit, 0, lds_cnt, cnt);
              const int rem = cnt - sk2, room = stage_cap<MODE>() - st2;
              if (rem <= room) {
                stage_write((posc >= sk2) ? st2 + posc - sk2 : -1, id, bb);
                st2 += rem; sk2 = 0; base += kStage;
                flush = base >= n;
              } else if (st2 == 0) {
                stage_write((posc >= sk2 && posc < sk2 + room) ? posc - sk2 : -1, id, bb);
                st2 = room; sk2 += room;
              }
            }
            if (!flush) continue;
            const int nl2 = st2;
            st2 = 0;
            if (nl2 == 0) continue;
            __syncthreads();
            bwd_round(nl2);
          }
        }
      }
    }
  }
}



__device__ __forceinline__ FaceRec face_rec(const float* vb, int i0, int i1, int i2, float r);
__device__ __forceinline__ FaceRec face_rec(const float* vb, const int32_t* __restrict__ faces, int f, float r) {
  return face_rec(vb, faces[3 * f], faces[3 * f + 1], faces[3 * f + 2], r);
}

// ---- set-up pieces (face records, super-tile binning, launch order): used by raster.hip's three set-up launches and, fused per
//      frame, by hand_front.hip

// per-face record with a bbox dilated by r; culled faces get an empty box.  vb: this frame's (V,3) NDC vertices (global or LDS)
__device__ __forceinline__ FaceRec face_rec(const float* vb, int i0, int i1, int i2, float r) {
  Tri t;
  t.x0 = vb[3 * i0]; t.y0 = vb[3 * i0 + 1]; t.z0 = vb[3 * i0 + 2];
  t.x1 = vb[3 * i1]; t.y1 = vb[3 * i1 + 1]; t.z1 = vb[3 * i1 + 2];
  t.x2 = vb[3 * i2]; t.y2 = vb[3 * i2 + 1]; t.z2 = vb[3 * i2 + 2];
  const float area = edge_fn(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2);
  const float zmax = fmaxf(t.z0, fmaxf(t.z1, t.z2)), zmin = fminf(t.z0, fminf(t.z1, t.z2));
  // skipped for every pixel: behind camera, |area| <= eps, any vertex with z < eps (z_invalid), non-finite
  const bool cull = (zmax < 0.f) || (area <= kEps && area >= -kEps) || (zmin < kEps) || !(area == area);
  FaceRec rec;
  rec.a = make_float4(t.x0, t.y0, t.z0, t.x1);
  rec.b = make_float4(t.y1, t.z1, t.x2, t.y2);
  rec.c = make_float4(t.z2, area, 0.f, 0.f);
  if (cull) {
    rec.bb = make_float4(3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f);
  } else {
    rec.bb = make_float4(fminf(t.x0, fminf(t.x1, t.x2)) - r, fmaxf(t.x0, fmaxf(t.x1, t.x2)) + r,
                         fminf(t.y0, fminf(t.y1, t.y2)) - r, fmaxf(t.y0, fmaxf(t.y1, t.y2)) + r);
  }
  return rec;
}

// One WAVE bins one (frame, 64x64 super-tile): streams the frame's bboxes `bb` (global or LDS) 64 at a time, ballot + popcount
// compaction, no barriers; the list `out` comes out in ascending face order (== PyTorch3D's tie-break order).
__device__ __forceinline__ void bin_super_tile(const float4* bb, int F, int S, int nsx, int st, int32_t* __restrict__ out,
                                               int32_t* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int sx = st % nsx, sy = st / nsx;
  const int x_lo = sx * kSuper, x_hi = min(x_lo + kSuper, S) - 1;
  const int y_lo = sy * kSuper, y_hi = min(y_lo + kSuper, S) - 1;
  // NDC decreases with pixel index
  const float nx_hi = pix_to_ndc(x_lo, S), nx_lo = pix_to_ndc(x_hi, S);
  const float ny_hi = pix_to_ndc(y_lo, S), ny_lo = pix_to_ndc(y_hi, S);
  int running = 0;
  // the loop is a chain of dependent ballots but the loads are independent: issue kUnroll of them before the first use, otherwise
  // every iteration pays a full L2 round trip (measured 50 us for 97 iterations at F = 6152; 2 waves per SIMD cannot hide it)
  constexpr int kUnroll = 8;
  for (int base = 0; base < F; base += 64 * kUnroll) {
    float4 q[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int f = base + u * 64 + lane;
      q[u] = (f < F) ? bb[f] : make_float4(3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int f = base + u * 64 + lane;
      const bool hit = !(nx_lo > q[u].y || nx_hi < q[u].x || ny_lo > q[u].w || ny_hi < q[u].z);
      const unsigned long long m = __ballot(hit);
      if (hit) out[running + __popcll(m & ((1ull << lane) - 1ull))] = f;
      running += __popcll(m);
    }
  }
  if (lane == 0) count[0] = running;
}

// Heaviest-first launch order of the (frame, super-tile) pairs: a counting sort of the bin counts by magnitude (33 buckets of
// count leading zeros).  The raster grid is 1-D over this order, so the workgroups with real work are dispatched first and
// densely (the natural (x, y, frame) order interleaves them with ~75 % empty tiles: measured average occupancy was < 3 of 8
// waves per SIMD) and the longest ones do not end up in the tail.   One workgroup; hist / base: 33 ints of LDS each.
__device__ __forceinline__ void order_tiles(const int32_t* __restrict__ bin_count, int total, int32_t* __restrict__ order, int32_t* __restrict__ nact,
                                            int* hist, int* base) {
  // (wave-aggregated bucket counts — one LDS atomic per distinct bucket of a wave — were built and measured SLOWER: 7.8 vs 5.4 us at 2048
  //  entries, 26.7 vs 19.6 at 8192: the ballot / permute trips cost more than the same-address LDS atomics they replace.)  The counts of a
  //  thread are requested 8 at a time instead of one per loop trip behind the previous trip's atomic.
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < total; i0 += (int)blockDim.x * 8) {
    int n[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x; n[u] = (i < total) ? bin_count[i] : -1; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (n[u] >= 0) atomicAdd(&hist[n[u] > 0 ? __clz(n[u]) : 32], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < 33; ++k) { base[k] = run; run += hist[k]; }
    nact[0] = base[32];                      // bucket 32 = empty lists: everything before it has work
  }
  __syncthreads();
  for (int i0 = 0; i0 < total; i0 += (int)blockDim.x * 8) {
    int n[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * (int)blockDim.x + (int)threadIdx.x; n[u] = (i < total) ? bin_count[i] : -1; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (n[u] >= 0) order[atomicAdd(&base[n[u] > 0 ? __clz(n[u]) : 32], 1)] = i0 + u * (int)blockDim.x + (int)threadIdx.x;
  }
}

}  // namespace rb
