// Backward pass of the fused shader for gfx950, wave-autonomous form (the one the fitting loop runs: harp_shade_bwd).
// Same per-pixel arithmetic as shade_kernel<true> in shade.hip (the autograd of SoftPhongShaderShadow / SoftPhongShaderPBR,
// renderer/renderer_helper.py:106-190, 472-523, 565-592; renderer/pbr_materials.py:58-124; fused torch.nn.L1Loss of
// optimize_sequence.py:543) — what changed is how the gradients leave the lanes.
//
// Why: the first version shared three LDS accumulators (vertex hash, texel hash, shadow-tap window) among the four waves of a
// 16x16-pixel tile.  That costs 17 workgroup barriers per tile (clears, window anchoring, block sums, the counting sort of the
// texel flush) around ~2 000 instructions per wave; its SQ counters (profiles/r02_a_pmc_sq_*) show where the time went: waves
// parked 56 % of their life (SQ_WAIT_ANY / SQ_WAVE_CYCLES), VALU busy 31 %, LDS busy 38 %, 2.6 waves per SIMD, ~39 k cycles of
// lifetime per active wave for ~2 k instructions.  Nothing was saturated; the kernel waited on itself.
//
// Removing the barriers alone (one private LDS slice per wave) changed almost nothing (0.355 -> 0.32 ms): the kernel is bound by
// LATENCY x WAVE COUNT, not by any unit.  Every wave walks a chain of 7 dependent memory round trips (launch-order slot -> frame /
// tile -> target row -> face id + mask -> face record + vertex ids -> vertex attributes -> texels -> shadow taps); a launch with
// the shading switched off still took 0.09 of the 0.32 ms; and fewer than half of the lanes of a 16x4-pixel strip are active (the
// photometric mask is the ERODED silhouette), so 58 k strips were shaded where 26 k waves' worth of pixels exist.  (A persistent
// form — fixed grid, each wave looping over tiles — was built too: the loop keeps the ~40 pointers of harp_shade_args and the
// loop state live, 120 VGPRs spill, and the shading part alone went from 0.19 to 0.30 ms.)
//
// Now, per 16x16 tile (one workgroup, dispatched heaviest-first):
//   * every wave reads face id + mask of its 16x4 strip (coalesced rows); the ACTIVE pixels of the whole tile are compacted with
//     ballots into one list in LDS (two barriers at the very top, when all four waves are still in step), and wave w shades
//     entries [64 w, 64 w + 64) of it — typically one or two full waves per tile instead of four half-empty ones; the others leave.
//   * after that the waves never meet again: gradients leave through WAVE-PRIVATE LDS tables:
//       - texels (4 bilinear corners x 6 channels per pixel): a DIRECT-MAPPED table indexed by (x mod 32, y mod 8) with a tag per
//         slot — neighbouring texels never collide, a second-chance slot half a table away catches chart seams, a real conflict
//         falls through to memory-side atomics.  Accumulators are 32-bit FIXED POINT (ds_add_u32 is the fastest LDS atomic on
//         gfx950: 4.8 clk per wave instruction against 8.7 - 44 for ds_add_f64 and 193 for ds_add_f32) with a
//         per-wave power-of-two scale taken from the wave's largest contribution: |sum| <= 64 lanes x max < 2^30, resolution 2^-23
//         of that maximum, i.e. float32-grade.  Flushed row-major, lanes = (texel, channel): consecutive addresses share memory
//         requests (330 G atomics/s against 21 G/s for random addresses), without the counting sort the hash table needed.
//       - vertices (position, normal, NDC: 27 values per pixel): lanes on the same face are merged first (DPP butterfly over the
//         x neighbours of the compacted order), the survivors add into a 32-slot fixed-point table (round 6; it was double: -3 us).
//       - shadow-map taps: 16x16 fixed-point window anchored at the wave's smallest tap.
//       - the 17 per-frame scalars (light colours, light position, light camera, loss): wave reductions -> LDS partials; the LAST
//         wave of the tile to finish (LDS ticket, no barrier) issues one memory atomic per scalar.
#include <stdlib.h>
#include "raster_body.h"
#include "shade_common.h"

#ifdef SHADE_STAMPS
// debug build only (tools/dev/gpu_shade_stamps.py): shader-clock stamps at the phase boundaries of a working wave, summed per phase
__device__ unsigned long long g_shade_stamps[4096][16];
#define STAMP(k) ts[k] = __builtin_readcyclecounter()
#else
#define STAMP(k)
#endif

namespace {

#ifndef SHADE_BWD_OCC
#define SHADE_BWD_OCC 4
#endif
// xor distances of the same-face lane merge in front of the vertex table: 1, 2, 4 (x neighbours of the compacted order) and 16 (the row
// below): measured 0.2453 (7) / 0.2502 (3) / 0.2526 (1) / 0.2478 (15) / 0.2414 (23) / 0.258 (31) ms — what the merge saves is same-address
// ds_add_f64 conflicts, which cost more than the DPP / permute steps
#ifndef SHADE_MERGE_MASK
#define SHADE_MERGE_MASK 23
#endif
#ifndef SHADE_BWD_TH
#define SHADE_BWD_TH 7
#endif
constexpr int kTW = 32, kTH = SHADE_BWD_TH, kTSlots = kTW * kTH;     // texel table: (x & 31, y mod kTH)
#ifndef SHADE_VSLOTS_LOG2
#define SHADE_VSLOTS_LOG2 5
#endif
#ifndef SHADE_ZW
#define SHADE_ZW 16
#endif
#ifndef SHADE_ZH
#define SHADE_ZH 16
#endif
constexpr int kVSlots = 1 << SHADE_VSLOTS_LOG2;           // vertex table
constexpr int kZW = SHADE_ZW, kZH = SHADE_ZH;             // shadow-tap window (light-view pixels)
constexpr int kScalars = 17;                              // 0-8 colours, 9-11 light_pos, 12-14 light_R[:,2], 15 light_T.z, 16 loss

// REC = the texel gradients leave as RECORDS (harp_shade_args.trec): no texel table — 3.4 KB of LDS per wave instead of 9.6
template <bool REC>
struct alignas(16) WaveLds {
  int tkey[REC ? 4 : kTSlots];
  int tval[6][REC ? 4 : kTSlots];        // fixed point; 0-2 albedo, 3-5 normal map
  int vkey[kVSlots];
#ifndef SHADE_VFIXED
#define SHADE_VFIXED 1
#endif
#if SHADE_VFIXED
  int vval[9][kVSlots];        // 0-2 g_verts, 3-5 g_vnormals, 6-8 g_ndc; 32-bit fixed point, one power-of-two scale per group and wave
#else
  double vval[9][kVSlots];     // 0-2 g_verts, 3-5 g_vnormals, 6-8 g_ndc
#endif
  int zwin[kZW * kZH];         // fixed point
};

// power-of-two scale s with |x| * s < 2^24 for every |x| <= m (so 64 such terms stay below 2^30); inv = 1 / s exactly
__device__ __forceinline__ void fixed_scale(float m, float& s, float& inv) {
  int e = ((__float_as_int(m) >> 23) & 0xff) - 126;          // m < 2^e
  e = min(max(e, -100), 100);
  s = __int_as_float((127 + 24 - e) << 23);
  inv = __int_as_float((127 - 24 + e) << 23);
}

template <bool REC>
struct ShadeSmem {
  WaveLds<REC> w[4];
  float part[4][20];
  int ticket;
  int cnt[4];
  int list[256];          // compacted active pixels of the tile: face id | (pixel in tile) << 24
};
#ifndef SHADE_SKIP_LDS_ASSERT
static_assert(sizeof(ShadeSmem<false>) <= (160 / SHADE_BWD_OCC) * 1024, "LDS budget per workgroup");
#endif

// one 16x16 tile of the shading backward; `vblock` = index in the 1-D heaviest-first tile grid
// IMG: the fused-loss pass also writes the colour it recomputes (a template flag: the loss-only kernel must not pay for the branch — 9 % measured)
template <bool IMG, bool REC>
__device__ __forceinline__ void shade_bwd_tile(ShadeSmem<REC>& sm, unsigned vblock, const harp_shade_args& A, const int32_t* __restrict__ order,
                                               const int32_t* __restrict__ nact, int nsx) {
  auto& s_w = sm.w; auto& s_part = sm.part; int& s_ticket = sm.ticket; auto& s_cnt = sm.cnt; auto& s_list = sm.list;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int S = A.S, V = A.V;
  // g_rgb == NULL = FUSED-LOSS mode: the pass forms torch.nn.L1Loss(y_true * m, y_pred * m) and its gradient from the colour it
  // recomputes anyway (no forward launch at all in a fitting step)
  const bool fused = A.g_rgb == nullptr;
  const int dbg = A.debug_skip >> 8;          // ablation switches (timing only, results WRONG): see harp_hip.h
  // g_verts == NULL (with g_vnormals, g_ndc): the caller wants no geometry gradients — the appearance-only stage of a fit, whose optimiser
  // holds texture, normal map, light and ambient ratio only: no second set of face loads, no barycentric backward, no vertex table
  const bool geom = A.g_verts != nullptr;
  // target rows of all frames of the batch, requested before anything else: the row of this tile's frame is then a lane read instead
  // of one more dependent trip behind the launch-order entry
  const int tf_all = (fused && lane < A.B) ? A.l1_fid[lane] : 0;
  int b, st, tx0, ty0, tsub;
#ifdef SHADE_STAMPS
  unsigned long long ts[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) ts[k] = 0;
#endif
  STAMP(0);
  const int kind = tile_decode_v(vblock, order, nact, A.B, nsx, S, b, st, tx0, ty0, tsub, fused);
  if (kind == 0 || (dbg & 64)) return;
  // fused-loss mode with an image pointer: the pass also WRITES the colour it recomputes (y_pred, every pixel of the frame) — the
  // "keep the rendered image" step without a forward shading launch
  const bool img = IMG && fused;
  if (kind == 2) {
    // super-tile without a face: no gradient; its part of the loss against the static targets is a table look-up
    if (img) {
      const int xi = tx0 + (lane & 15), yi = ty0 + w * 4 + (lane >> 4);
      if (xi < S && yi < S) {
        float* o = at32m(A.rgb + (size_t)b * S * S * 3, 3u * (unsigned)(yi * S + xi));
        o[0] = A.bg[0]; o[1] = A.bg[1]; o[2] = A.bg[2];
      }
    }
    if (fused && tsub == 0 && threadIdx.x == 0) {
      const float sum = A.l1_bg_sums[(size_t)A.l1_fid[b] * nsx * nsx + st];
      if (sum != 0.f) atomicAdd(A.l1_loss, sum * A.l1_inv);
    }
    return;
  }
  const int tfid = !fused ? 0 : (A.B <= 64 ? __builtin_amdgcn_readlane(tf_all, b) : A.l1_fid[b]);
  WaveLds<REC>& L = s_w[w];
  if (threadIdx.x == 0) s_ticket = 0;

  // ---- own 16x4 strip: face id, mask -> active flag (coalesced rows)
  float loss_acc = 0.f;
  size_t tbase = 0;
  {
    const int xi = tx0 + (lane & 15), yi = ty0 + w * 4 + (lane >> 4);
    const bool in_img = xi < S && yi < S;
    const unsigned pix_o = (unsigned)(yi * S + xi);                  // (32-bit offsets from wave-uniform bases: scalar-base addressing)
    const int f0 = in_img ? *at32(A.face_id + (size_t)b * S * S, pix_o) : -1;
    bool act0 = f0 >= 0;
    float m0 = 0.f;
    if (fused) {
      tbase = (size_t)tfid * S * S;
      if (in_img) {
        m0 = A.l1_mask ? *at32(A.l1_mask + tbase, pix_o) : 1.f;
        if (!act0 && m0 != 0.f) {         // uncovered pixel inside the mask: background colour against the target, no gradient
          const float* t = at32(A.l1_target + tbase * 3, 3u * pix_o);
          loss_acc = fabsf(A.bg[0] * m0 - t[0] * m0) + fabsf(A.bg[1] * m0 - t[1] * m0) + fabsf(A.bg[2] * m0 - t[2] * m0);
        }
      }
      if (img && in_img && f0 < 0) {
        float* o = at32m(A.rgb + (size_t)b * S * S * 3, 3u * pix_o);
        o[0] = A.bg[0]; o[1] = A.bg[1]; o[2] = A.bg[2];
      }
      act0 = act0 && (m0 != 0.f || img);       // (with an image, covered pixels outside the mask are shaded too: weight 0, colour kept)
    }
    if (dbg & 32) act0 = false;
    // ---- compaction of the tile's active pixels (row-major): wave w then shades entries [64 w, 64 w + 64)
    const unsigned long long bal = __ballot(act0 ? 1 : 0);
    if (lane == 0) s_cnt[w] = __popcll(bal);
    __syncthreads();                     // every wave is still at the top of the kernel: cheap
    int base = 0;
    for (int i = 0; i < w; ++i) base += s_cnt[i];
    if (act0) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      s_list[pos] = f0 | (((w * 4 + (lane >> 4)) * 16 + (lane & 15)) << 24);
    }
    __syncthreads();
  }
  const int n_tile = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  if (64 * w < n_tile) {                     // only waves that got a share of the tile's active pixels clear their tables
    // (16-B stores over the whole 9.5-KB block, then the two key arrays: 14 LDS instructions instead of 43)
    int4* L4 = reinterpret_cast<int4*>(&L);
    for (int i = lane; i < (int)(sizeof(WaveLds<REC>) / 16); i += 64) L4[i] = make_int4(0, 0, 0, 0);
    if constexpr (!REC) for (int i = lane; i < kTSlots; i += 64) L.tkey[i] = -1;
    for (int i = lane; i < kVSlots; i += 64) L.vkey[i] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  bool act = 64 * w + lane < n_tile;
  const int ent = act ? s_list[64 * w + lane] : 0;
  const int f = ent & 0x00ffffff;
  const int pix = (int)((unsigned)ent >> 24);
  const int xi = tx0 + (pix & 15), yi = ty0 + (pix >> 4);
  const unsigned po = (unsigned)(yi * S + xi);
  const float* l1_trow = A.l1_target + tbase * 3;
  const float l1_m = (act && fused) ? (A.l1_mask ? *at32(A.l1_mask + tbase, po) : 1.f) : 0.f;      // (re-read: L1 / L2 hit, one LDS kilobyte less)
  V3 gc = mk(0.f, 0.f, 0.f);
  if (!fused) {
    if (act) gc = ld(at32(A.g_rgb + (size_t)b * S * S * 3, 3u * po));
    act = act && (gc.x != 0.f || gc.y != 0.f || gc.z != 0.f);
  }

  STAMP(1);
  float racc[kScalars];
#pragma unroll
  for (int k = 0; k < kScalars; ++k) racc[k] = 0.f;
  racc[16] = loss_acc;
  float vsc[27];          // gradient of the face's 3 vertices x (position, normal, ndc)
  int vidx[3] = {0, 0, 0};
  float zd[9];            // gradient of the 3x3 shadow-map taps (row-major)
  int zix = -0x40000000, ziy = -0x40000000;
  V3 g_tex = mk(0.f, 0.f, 0.f), g_m_keep = mk(0.f, 0.f, 0.f);
  Bil bs;
  bs.x0 = 0; bs.y0 = 0; bs.wx = 0.f; bs.wy = 0.f; bs.gxm = 0.f; bs.gym = 0.f;
#pragma unroll
  for (int c = 0; c < 27; ++c) vsc[c] = 0.f;
#pragma unroll
  for (int c = 0; c < 9; ++c) zd[c] = 0.f;

  bool dead = false;      // IMG mode: covered pixel outside the L1 mask — its colour is written, it stays out of the table phases
  // REC: the lane's slot in its UV tile's record list = (count returned to the group's leader lane) + rank in the group
  int rbase = 0, rwho = 0;
  const bool rec_on = REC && (A.g_tex != nullptr || (A.nmap != nullptr && A.g_nmap != nullptr));
  if (act) {
    const float px = pix_to_ndc(xi, S), py = pix_to_ndc(yi, S);
    const float* col = A.colors;               // amb(3) diff(3) spec(3)
    Frag g;
    const unsigned uf = (unsigned)f;
    g.t = load_tri(at32(((const FaceRec*)A.recs) + (size_t)b * A.F, uf));
    g.br = bary_fwd(g.t, px, py);
    const float b0 = g.br.b0, b1 = g.br.b1, b2 = g.br.b2;
    { const int32_t* fi = at32(A.faces, 3u * uf); g.i0 = fi[0]; g.i1 = fi[1]; g.i2 = fi[2]; }
    { const int32_t* fu = at32(A.faces_uvs, 3u * uf); g.u0 = fu[0]; g.u1 = fu[1]; g.u2 = fu[2]; }
    const float* vb = A.verts + (size_t)b * V * 3;
    const float* nb = A.vnormals + (size_t)b * V * 3;
    const unsigned j0 = 3u * (unsigned)g.i0, j1 = 3u * (unsigned)g.i1, j2 = 3u * (unsigned)g.i2;
    const unsigned k0 = 2u * (unsigned)g.u0, k1 = 2u * (unsigned)g.u1, k2 = 2u * (unsigned)g.u2;
    const V3 v0 = ld(at32(vb, j0)), v1 = ld(at32(vb, j1)), v2 = ld(at32(vb, j2));
    const V3 n0 = ld(at32(nb, j0)), n1 = ld(at32(nb, j1)), n2 = ld(at32(nb, j2));
    const float uv0x = at32(A.verts_uvs, k0)[0], uv0y = at32(A.verts_uvs, k0)[1];
    const float uv1x = at32(A.verts_uvs, k1)[0], uv1y = at32(A.verts_uvs, k1)[1];
    const float uv2x = at32(A.verts_uvs, k2)[0], uv2y = at32(A.verts_uvs, k2)[1];
    g.p = v0 * b0 + v1 * b1 + v2 * b2;
    g.n = n0 * b0 + n1 * b1 + n2 * b2;
    g.u = uv0x * b0 + uv1x * b1 + uv2x * b2;
    g.v = uv0y * b0 + uv1y * b1 + uv2y * b2;
    g.bs = bil_setup(g.u, g.v, A.Wt, A.Ht);
    bs = g.bs;
    // the 9 shadow-map taps only depend on the surface point: requested here, next to the texel fetch, instead of behind it (one
    // dependent round trip less in the wave's chain: step -2 us, same-box A/B x3)
    float zt[9];
    const float half = 0.5f * (float)S;
    if (A.zl) {
      const float* R = A.light_R + 9 * b;
      const float* T = A.light_T + 3 * b;
      g.q = mk(g.p.x * R[0] + g.p.y * R[3] + g.p.z * R[6] + T[0], g.p.x * R[1] + g.p.y * R[4] + g.p.z * R[7] + T[1],
               g.p.x * R[2] + g.p.y * R[5] + g.p.z * R[8] + T[2]);
      const float rqz = rcp(g.q.z), rhalf = rcp(half);
      const float xn = (A.focal * g.q.x * rqz - A.ppx + half) * rhalf, yn = (A.focal * g.q.y * rqz - A.ppy + half) * rhalf;
      const float xs = half - half * xn, ys = half - half * yn;
      g.ix = (int)rintf(fminf(fmaxf(xs, -1.0e6f), 1.0e6f));     // torch .round().long(): half-to-even
      g.iy = (int)rintf(fminf(fmaxf(ys, -1.0e6f), 1.0e6f));
      const float* zlb = A.zl + (size_t)b * S * S;
      int k = 0;
#pragma unroll
      for (int ii = -1; ii <= 1; ++ii)
#pragma unroll
        for (int jj = -1; jj <= 1; ++jj, ++k) {
          const int yy = min(max(g.iy + ii, 0), S - 1), xx = min(max(g.ix + jj, 0), S - 1);
          zt[k] = *at32(zlb, (unsigned)(yy * S + xx));
        }
    }
#ifdef SHADE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)");
#endif
    STAMP(2);
    V3 tdx, tdy, mdx = mk(0.f, 0.f, 0.f), mdy = mk(0.f, 0.f, 0.f);
    const bool packed = A.texnm != nullptr && A.nmap != nullptr;
    if (packed) bil_sample2((const float4*)A.texnm, g.bs, A.Wt, A.Ht, g.texel, g.m, &tdx, &tdy, &mdx, &mdy);
    else g.texel = bil_sample(A.tex, g.bs, A.Wt, A.Ht, &tdx, &tdy);
    if constexpr (REC) {
      // a slot in the record list of the 32x32-texel UV tile the footprint starts in: ONE returning atomic per distinct tile of the wave
      // (1.8 on average: screen neighbours are UV neighbours), issued behind the texel fetch and not waited for until the record is written
      if (rec_on) {
        const bool emit = !(IMG && fused && l1_m == 0.f);
        const int rbin = (g.bs.y0 >> 5) * ((A.Wt + 31) >> 5) + (g.bs.x0 >> 5);
        unsigned long long todo = __ballot(emit ? 1 : 0);
        while (todo) {
          const int l = __builtin_ctzll(todo);
          const int bb = __builtin_amdgcn_readlane(rbin, l);
          const bool mine = emit && rbin == bb;
          const unsigned long long m = __ballot(mine ? 1 : 0);
          if (mine) rwho = l | ((int)__popcll(m & ((1ull << lane) - 1ull)) << 8);
          if (lane == l) rbase = atomicAdd(A.trec_cnt + 16 * bb, (int)__popcll(m));
          todo &= ~m;
        }
      }
    }
    // normal map (pbr_materials.py:58-124): n' = normalize(-u m.x - v m.y + n m.z)
    V3 nfin = g.n;
    if (A.nmap) {
      if (!packed) g.m = bil_sample(A.nmap, g.bs, A.Wt, A.Ht, &mdx, &mdy);
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
#ifdef SHADE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)");
#endif
    STAMP(3);
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
    if (A.zl) {
      const float aa = g.q.z - 0.008f;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) { sg[k] = sigmoidf((zt[k] - aa) * 1000.0f); acc += sg[k]; }
      g.vis = acc * (1.0f / 9.0f);
    }
#ifdef SHADE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)");
#endif
    STAMP(4);
    const V3 amb = ld(col), dfc = ld(col + 3), spc = ld(col + 6);
    const V3 lightc = mk(amb.x + dfc.x * cosang * g.vis, amb.y + dfc.y * cosang * g.vis, amb.z + dfc.z * cosang * g.vis);
    const V3 c = mk(lightc.x * g.texel.x + spc.x, lightc.y * g.texel.y + spc.y, lightc.z * g.texel.z + spc.z);
    // softmax_rgb_blend, K=1, blur=0 (Appendix A.4); prob in (0.5,1] is taken as 1 (error <= 2e-10)
    const float zpix = b0 * g.t.z0 + b1 * g.t.z1 + b2 * g.t.z2;
    const float zinv = (100.0f - zpix) * (1.0f / 99.0f);
    const float zmax = fmaxf(zinv, 1e-10f);
    const float wnum = __expf((zinv - zmax) * 1e4f);
    const float delta = fmaxf(__expf((1e-10f - zmax) * 1e4f), 1e-10f);
    const float rden = rcp(wnum + delta);
    const float wk = wnum * rden;
    if (fused) {
      // the forward colour of this pixel, the L1 against the target and its gradient (same expressions as the forward kernel)
      const float o3[3] = {(wnum * c.x + delta * A.bg[0]) * rden, (wnum * c.y + delta * A.bg[1]) * rden, (wnum * c.z + delta * A.bg[2]) * rden};
      if (img) {
        float* o = at32m(A.rgb + (size_t)b * S * S * 3, 3u * po);
        o[0] = o3[0]; o[1] = o3[1]; o[2] = o3[2];
      }
      dead = IMG && l1_m == 0.f;       // (no table traffic, no zero-valued atomics, no 0 * inf in the sums for such a pixel)
      const float wl = A.l1_w[0] * A.l1_inv * l1_m;
      float gq[3] = {0.f, 0.f, 0.f};
      if (!dead) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float d = o3[ch] * l1_m - at32(l1_trow, 3u * po)[ch] * l1_m;
          racc[16] += fabsf(d);
          gq[ch] = wl * (float)((d > 0.f) - (d < 0.f));
        }
      }
      gc = mk(gq[0], gq[1], gq[2]);
    }
    const V3 g_c = gc * wk;
    // c = lightc * texel + spec
    g_tex = mk(g_c.x * lightc.x, g_c.y * lightc.y, g_c.z * lightc.z);
    const V3 g_lc = mk(g_c.x * g.texel.x, g_c.y * g.texel.y, g_c.z * g.texel.z);
    racc[0] = g_lc.x; racc[1] = g_lc.y; racc[2] = g_lc.z;                                     // amb
    racc[3] = g_lc.x * cosang * g.vis; racc[4] = g_lc.y * cosang * g.vis; racc[5] = g_lc.z * cosang * g.vis;  // diff
    racc[6] = g_c.x; racc[7] = g_c.y; racc[8] = g_c.z;                                        // spec
    const float g_dv = g_lc.x * dfc.x + g_lc.y * dfc.y + g_lc.z * dfc.z;                      // d/d(cosang*vis)
    const float g_vis = g_dv * cosang;
    const float g_cos = (g.cosr > 0.f) ? g_dv * g.vis : 0.f;
    float gu = dot(g_tex, tdx) * (float)(A.Wt - 1), gv = dot(g_tex, tdy) * -(float)(A.Ht - 1);   // d/d(u,v)
    // cos = nn . lhat
    const V3 g_nn = g.lhat * g_cos, g_lhat = g.nn * g_cos;
    const V3 g_ldir = (g.llen > 1e-6f) ? (g_lhat - g.lhat * dot(g.lhat, g_lhat)) * rcp(g.llen) : g_lhat * 1e6f;
    racc[9] = g_ldir.x; racc[10] = g_ldir.y; racc[11] = g_ldir.z;                             // light_pos
    V3 g_p = mk(-g_ldir.x, -g_ldir.y, -g_ldir.z);
    const V3 g_nfin = (g.lnh > 1e-6f) ? (g_nn - g.nn * dot(g.nn, g_nn)) * rcp(g.lnh) : g_nn * 1e6f;
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
    // shadow
    if (A.zl) {
      float g_zq = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float d = g_vis * (1.0f / 9.0f) * sg[k] * (1.0f - sg[k]) * 1000.0f;
        zd[k] = d;
        g_zq -= d;
      }
      zix = g.ix; ziy = g.iy;
      const float* R = A.light_R + 9 * b;
      g_p = g_p + mk(R[2], R[5], R[8]) * g_zq;
      racc[12] = g.p.x * g_zq; racc[13] = g.p.y * g_zq; racc[14] = g.p.z * g_zq;             // light_R[:,2]
      racc[15] = g_zq;                                                                        // light_T.z
    }
    // interpolation backward.  The 33 per-face values (vertex positions, normals, uvs, NDC record) are LOADED AGAIN here (L1 / L2
    // hits) instead of being held in registers across the texel fetch, the lighting and the shadow test: what keeps this kernel at
    // 3 waves per SIMD is its register peak in that middle section, and its time goes with 1 / waves.  The base
    // pointers pass through an empty asm so that the compiler cannot merge the second set of loads with the first.
    float gnd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (geom) {
      const float* vb2 = vb; const float* nb2 = nb; const float* uvb2 = A.verts_uvs; const FaceRec* rec2 = ((const FaceRec*)A.recs) + (size_t)b * A.F;
      asm volatile("" : "+s"(vb2), "+s"(nb2), "+s"(uvb2), "+s"(rec2));
      const V3 w0 = ld(at32(vb2, j0)), w1 = ld(at32(vb2, j1)), w2 = ld(at32(vb2, j2));
      const V3 m0 = ld(at32(nb2, j0)), m1 = ld(at32(nb2, j1)), m2 = ld(at32(nb2, j2));
      const float q0x = at32(uvb2, k0)[0], q0y = at32(uvb2, k0)[1], q1x = at32(uvb2, k1)[0], q1y = at32(uvb2, k1)[1];
      const float q2x = at32(uvb2, k2)[0], q2y = at32(uvb2, k2)[1];
      const Tri t2 = load_tri(at32(rec2, uf));
      const float gb0 = dot(w0, g_p) + dot(m0, g_n) + q0x * gu + q0y * gv;
      const float gb1 = dot(w1, g_p) + dot(m1, g_n) + q1x * gu + q1y * gv;
      const float gb2 = dot(w2, g_p) + dot(m2, g_n) + q2x * gu + q2y * gv;
      const Bary br2 = bary_fwd(t2, px, py);            // (same values as in the forward half)
      bary_bwd(t2, px, py, br2, gb0, gb1, gb2, gnd);
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
  if (IMG && dead) {
    // its gradient section ran with weight 0 (every product is an exact zero unless a colour overflowed: clear the sums anyway); the
    // lane takes no part in the table phases below
    act = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) racc[k] = 0.f;
  }

  STAMP(5);
  // ---- per-frame scalars: wave sums -> this wave's LDS partials (the last wave of the tile sends them on, below)
  const bool working = 64 * w < n_tile;          // (wave-uniform) this wave has a share of the tile's active pixels
  if (working) {
    wave_sum_u_n<kScalars>(racc);            // (all 17 chains interleaved; lane 63 holds the sums)
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < kScalars; ++k) s_part[w][k] = racc[k];
    }
  } else {
    const float s = wave_sum_u(loss_acc);
    if (lane < kScalars) s_part[w][lane] = (lane == 16) ? s : 0.f;
  }
  const bool any_act = working && __any(act ? 1 : 0) != 0;
  STAMP(6);

#if SHADE_VFIXED
  float vinv[3] = {0.f, 0.f, 0.f};        // 1 / scale of the three groups of the vertex table
#define VVAL_F(x, c) ((float)(x) * vinv[(c) / 3])
#else
#define VVAL_F(x, c) ((float)(x))
#endif
  float* gvb = A.g_verts + (size_t)b * V * 3;
  float* gnb = A.g_vnormals + (size_t)b * V * 3;
  float* gdb = A.g_ndc + (size_t)b * V * 3;
  float* g9b = A.g_vert9 ? A.g_vert9 + (size_t)b * V * 9 : nullptr;
  if (any_act) {
    if (!(dbg & 4) && geom) {
    // ---- vertex gradients: lanes on the same face add to the same three vertices: merge them first (xor distances 1, 2, 4 = x
    //      neighbours in the compacted order), the survivors add into the wave's double table
    bool alive = act;
    merge_same_face<27, SHADE_MERGE_MASK>(vsc, act ? f : -1, alive, lane);
#if SHADE_VFIXED
    // fixed-point adds (ds_add_u32: 4.8 clk per wave instruction and 3.6 per further lane on the same address, against 8.7 and 11 for
    // ds_add_f64 — and neighbouring faces DO share vertices): one scale per gradient group from the wave's largest (merged) contribution, as
    // for the texel table: |sum| <= 64 lanes x max < 2^30, resolution 2^-23 of that maximum
    float vm[3] = {0.f, 0.f, 0.f};
    if (alive) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 9; ++c) vm[c / 3] = fmaxf(vm[c / 3], fabsf(vsc[9 * k + c]));
    }
    float vs[3], vi[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { vm[q] = wave_max_u(vm[q]); fixed_scale(vm[q], vs[q], vi[q]); }
    vinv[0] = vi[0]; vinv[1] = vi[1]; vinv[2] = vi[2];
#endif
    if (alive) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int v = vidx[k];
        unsigned h = ((unsigned)v * 2654435761u) >> (32 - SHADE_VSLOTS_LOG2);
        int slot = -1;
        for (int probe = 0; probe < 4; ++probe) {
          const int old = atomicCAS(&L.vkey[h], -1, v);
          if (old == -1 || old == v) { slot = (int)h; break; }
          h = (h + 1) & (kVSlots - 1);
        }
        if (slot >= 0) {
#pragma unroll
#if SHADE_VFIXED
          for (int c = 0; c < 9; ++c) atomicAdd(&L.vval[c][slot], __float2int_rn(vsc[9 * k + c] * vs[c / 3]));
#else
          for (int c = 0; c < 9; ++c) atomicAdd(&L.vval[c][slot], (double)vsc[9 * k + c]);
#endif
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            if (g9b) {
              atomicAdd(at32m(g9b, 9u * (unsigned)v + c), vsc[9 * k + c]); atomicAdd(at32m(g9b, 9u * (unsigned)v + 3 + c), vsc[9 * k + 3 + c]);
              atomicAdd(at32m(g9b, 9u * (unsigned)v + 6 + c), vsc[9 * k + 6 + c]);
            } else {
              atomicAdd(at32m(gvb, 3u * (unsigned)v + c), vsc[9 * k + c]); atomicAdd(at32m(gnb, 3u * (unsigned)v + c), vsc[9 * k + 3 + c]);
              atomicAdd(at32m(gdb, 3u * (unsigned)v + c), vsc[9 * k + 6 + c]);
            }
          }
        }
      }
    }

    }
    STAMP(7);
    // ---- shadow-map tap gradients: fixed-point window anchored at the strip's smallest (clamped) tap column / row
    if (A.zl && A.g_zl && !(dbg & 2)) {
      const bool has = zix > -0x40000000;
      const int x0 = wave_min_u(has ? min(max(zix - 1, 0), S - 1) : 0x7fffffff);
      const int y0 = wave_min_u(has ? min(max(ziy - 1, 0), S - 1) : 0x7fffffff);
      float zm = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) zm = fmaxf(zm, fabsf(zd[k]));
      zm = wave_max_u(zm);
      if (zm > 0.f) {
        float zs, zinv_s;
        fixed_scale(zm * 16.0f, zs, zinv_s);           // up to 9 taps of 64 lanes on one light pixel: 4 more bits of head room
        float* gz = A.g_zl + (size_t)b * S * S;
        const int znt = (S + 15) >> 4;
        if (has) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int yy = min(max(ziy + r - 1, 0), S - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float d = zd[3 * r + c];
              if (d == 0.f) continue;
              const int xx = min(max(zix + c - 1, 0), S - 1);
              const int wx = xx - x0, wy = yy - y0;
              if (wx < kZW && wy < kZH) atomicAdd(&L.zwin[wy * kZW + wx], __float2int_rn(d * zs));
              else {
                atomicAdd(at32m(gz, (unsigned)(yy * S + xx)), d);
                if (A.g_zl_tiles) A.g_zl_tiles[((size_t)b * znt + (yy >> 4)) * znt + (xx >> 4)] = 1;
              }
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int zq[kZW * kZH / 64];
#pragma unroll
        for (int j = 0; j < kZW * kZH / 64; ++j) zq[j] = L.zwin[lane + 64 * j];
#pragma unroll
        for (int j = 0; j < kZW * kZH / 64; ++j) {
          const int i = lane + 64 * j;
          if (zq[j] != 0) {
            atomicAdd(at32m(gz, (unsigned)((y0 + i / kZW) * S + x0 + (i % kZW))), (float)zq[j] * zinv_s);
            // the light-view tile now holds a gradient (plain byte stores of the same value: no ordering needed)
            if (A.g_zl_tiles) A.g_zl_tiles[((size_t)b * znt + ((y0 + i / kZW) >> 4)) * znt + ((x0 + i % kZW) >> 4)] = 1;
          }
        }
      }
    }

    STAMP(8);
    if constexpr (REC) {
      // ---- texture + normal-map gradients leave as one 36-byte record per pixel (planes of 4 bytes: consecutive slots of a tile's list
      //      are consecutive addresses); harp_texel_reduce adds the footprints up, tile by tile
      if (rec_on && !(dbg & 1)) {
        const int pos = __shfl(rbase, rwho & 63, 64) + (rwho >> 8);
        if (act) {
          if (pos < A.trec_cap) {
            const int rbin = (bs.y0 >> 5) * ((A.Wt + 31) >> 5) + (bs.x0 >> 5);
            const size_t cap = (size_t)A.trec_cap;
            float* r = A.trec + (size_t)rbin * 9 * cap + (size_t)pos;
            r[0] = __int_as_float(bs.x0 | (bs.y0 << 16));
            r[cap] = bs.wx; r[2 * cap] = bs.wy;
            r[3 * cap] = g_tex.x; r[4 * cap] = g_tex.y; r[5 * cap] = g_tex.z;
            r[6 * cap] = g_m_keep.x; r[7 * cap] = g_m_keep.y; r[8 * cap] = g_m_keep.z;
          } else {                                   // the tile's list is full: straight into the reduce's double maps
            if (A.g_tex) bil_scatter_d(A.trec_acc_tex, bs, A.Wt, A.Ht, g_tex);
            if (A.nmap && A.g_nmap) bil_scatter_d(A.trec_acc_nmap, bs, A.Wt, A.Ht, g_m_keep);
          }
        }
      }
    } else
    // ---- texture + normal-map gradients: 4 bilinear corners x 6 channels into the direct-mapped fixed-point table
    if (!(dbg & 1) && (A.g_tex != nullptr || (A.nmap != nullptr && A.g_nmap != nullptr))) {      // (both maps frozen — known_appearance fits: no texel phase)
      const bool do_t = A.g_tex != nullptr, do_n = (A.nmap != nullptr) && (A.g_nmap != nullptr);
      const float ma = wave_max_u(fmaxf(fabsf(g_tex.x), fmaxf(fabsf(g_tex.y), fabsf(g_tex.z))));
      const float mm = wave_max_u(fmaxf(fabsf(g_m_keep.x), fmaxf(fabsf(g_m_keep.y), fabsf(g_m_keep.z))));
      float sa, ia, sm, im;
      fixed_scale(ma, sa, ia);
      fixed_scale(mm, sm, im);
      if (act) {
        const float ax = 1.f - bs.wx, ay = 1.f - bs.wy;
        const float cw[4] = {ax * ay, bs.wx * ay, ax * bs.wy, bs.wx * bs.wy};
        int key[4], slot[4], old[4];
        bool valid[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int cx = bs.x0 + (k & 1), cy = bs.y0 + (k >> 1);
          valid[k] = cx < A.Wt && cy < A.Ht && cw[k] != 0.f;
          key[k] = cy * A.Wt + cx;
          slot[k] = (cx & (kTW - 1)) + kTW * (int)((unsigned)cy % (unsigned)kTH);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) old[k] = valid[k] ? atomicCAS(&L.tkey[slot[k]], -1, key[k]) : key[k];     // four independent LDS round trips
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!valid[k]) continue;
          bool ok = old[k] == -1 || old[k] == key[k];
          if (!ok) {                      // taken by a texel >= 32 columns / 7 rows away (chart seam, strong minification): second chance
            slot[k] = slot[k] >= kTSlots / 2 ? slot[k] - kTSlots / 2 : slot[k] + kTSlots / 2;
            const int o2 = atomicCAS(&L.tkey[slot[k]], -1, key[k]);
            ok = o2 == -1 || o2 == key[k];
          }
          const float va[3] = {g_tex.x * cw[k], g_tex.y * cw[k], g_tex.z * cw[k]};
          const float vm[3] = {g_m_keep.x * cw[k], g_m_keep.y * cw[k], g_m_keep.z * cw[k]};
          if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (do_t) atomicAdd(&L.tval[c][slot[k]], __float2int_rn(va[c] * sa));
              if (do_n) atomicAdd(&L.tval[3 + c][slot[k]], __float2int_rn(vm[c] * sm));
            }
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (do_t) atomicAdd(at32m(A.g_tex, 3u * (unsigned)key[k] + c), va[c]);
              if (do_n) atomicAdd(at32m(A.g_nmap, 3u * (unsigned)key[k] + c), vm[c]);
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      // flush, lanes = (slot, channel): a table row holds consecutive texels of one texture row -> consecutive addresses
#ifndef SHADE_FLUSH_SERIAL
      if (!(dbg & 8)) {
        // all LDS reads of the flush in flight at once (3 x 11 per lane), then the memory atomics: the loop form paid two dependent
        // LDS round trips per pass, 21 in a row
        constexpr int kP = (kTSlots * 3 + 63) / 64;
        int key[kP], a0[kP], a1[kP];
#pragma unroll
        for (int j = 0; j < kP; ++j) {
          const int i = min(lane + 64 * j, kTSlots * 3 - 1), sl = i / 3, c = i - 3 * sl;
          key[j] = L.tkey[sl]; a0[j] = L.tval[c][sl]; a1[j] = L.tval[3 + c][sl];
        }
#pragma unroll
        for (int j = 0; j < kP; ++j) {
          const int i = lane + 64 * j, c = i % 3;
          if (i < kTSlots * 3 && key[j] >= 0) {
            if (a0[j] != 0) atomicAdd(at32m(A.g_tex, 3u * (unsigned)key[j] + c), (float)a0[j] * ia);
            if (a1[j] != 0) atomicAdd(at32m(A.g_nmap, 3u * (unsigned)key[j] + c), (float)a1[j] * im);
          }
        }
      }
#else
      if (!(dbg & 8)) for (int i = lane; i < kTSlots * 3; i += 64) {
        const int sl = i / 3, c = i - 3 * sl;
        const int key = L.tkey[sl];
        if (key < 0) continue;
        const int a0 = L.tval[c][sl], a1 = L.tval[3 + c][sl];
        if (a0 != 0) atomicAdd(at32m(A.g_tex, 3u * (unsigned)key + c), (float)a0 * ia);
        if (a1 != 0) atomicAdd(at32m(A.g_nmap, 3u * (unsigned)key + c), (float)a1 * im);
      }
#endif
    }

    STAMP(9);
    // ---- flush the vertex table, lanes = (slot, component)
#ifndef SHADE_FLUSH_SERIAL
    if (!(dbg & 12) && geom) {
      constexpr int kQ = (kVSlots * 9 + 63) / 64;
      int vk[kQ]; float vv[kQ];
#pragma unroll
      for (int j = 0; j < kQ; ++j) {
        const int i = min(lane + 64 * j, kVSlots * 9 - 1), sl = i / 9, c = i - 9 * sl;
        vk[j] = L.vkey[sl]; vv[j] = VVAL_F(L.vval[c][sl], c);
      }
#pragma unroll
      for (int j = 0; j < kQ; ++j) {
        const int i = lane + 64 * j, c = i % 9;
        if (i < kVSlots * 9 && vk[j] >= 0 && vv[j] != 0.f) {
          if (g9b) {
            atomicAdd(at32m(g9b, 9u * (unsigned)vk[j] + c), vv[j]);          // a slot's 9 values are ONE 36-byte run (harp_shade_args.g_vert9)
          } else {
            float* dst = c < 3 ? gvb : (c < 6 ? gnb : gdb);
            atomicAdd(at32m(dst, 3u * (unsigned)vk[j] + (c % 3)), vv[j]);
          }
        }
      }
    }
#else
    if (!(dbg & 12) && geom) for (int i = lane; i < kVSlots * 9; i += 64) {
      const int sl = i / 9, c = i - 9 * sl;
      const int v = L.vkey[sl];
      if (v < 0) continue;
      const float val = VVAL_F(L.vval[c][sl], c);
      if (val == 0.f) continue;
      if (g9b) { atomicAdd(at32m(g9b, 9u * (unsigned)v + c), val); continue; }
      float* dst = c < 3 ? gvb : (c < 6 ? gnb : gdb);
      atomicAdd(at32m(dst, 3u * (unsigned)v + (c % 3)), val);
    }
#endif
  }

#ifdef SHADE_STAMPS
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  STAMP(10);
  if (any_act && lane == 0) {
    unsigned long long* dst = g_shade_stamps[(blockIdx.x * 4 + w) & 4095];
#pragma unroll
    for (int k = 1; k <= 10; ++k) atomicAdd(&dst[k], ts[k] - ts[k - 1]);
    atomicAdd(&dst[0], 1ull);
  }
#endif
  // ---- the last wave of the tile to get here sends the tile's scalar sums on (one memory atomic per scalar per tile)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  int ticket = 0;
  if (lane == 0) ticket = atomicAdd(&s_ticket, 1);
  ticket = __shfl(ticket, 0, 64);
  if (ticket == 3) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < kScalars) {
      const int k = lane;
      const float s = s_part[0][k] + s_part[1][k] + s_part[2][k] + s_part[3][k];
      if (s != 0.f) {
        if (k < 9) { if (A.g_colors) atomicAdd(A.g_colors + k, s); }
        else if (k < 12) { if (A.g_light_pos) atomicAdd(A.g_light_pos + 3 * b + (k - 9), s); }
        else if (k < 15) { if (A.g_light_R) atomicAdd(A.g_light_R + 9 * b + 3 * (k - 12) + 2, s); }
        else if (k == 15) { if (A.g_light_T) atomicAdd(A.g_light_T + 3 * b + 2, s); }
        else if (fused) atomicAdd(A.l1_loss, s * A.l1_inv);
      }
    }
  }
}

// (launch-bounds hint of the record form: 120 - 122 VGPRs = 4 waves per SIMD whatever the hint says; with a lower hint hipcc schedules the same
//  registers a little better — 3 instead of 4: step -2.5 us in 4 of 4 fresh-process pairs under the default scheduler; 2 instead of 3 under the
//  max-memory-clause scheduler (harp_amd/build.py): -3 us, better in 6 of 8 pairs, 2 ties; 5 = 96 VGPRs + 51 spills: +40 us.
//  profiles/r06_ab_record.txt items 10, 28)
#ifndef SHADE_REC_OCC
#define SHADE_REC_OCC 2
#endif
template <bool IMG, bool REC>
__global__ void __launch_bounds__(256, REC ? SHADE_REC_OCC : SHADE_BWD_OCC) shade_bwd_wave_kernel(const harp_shade_args A, const int32_t* __restrict__ order,
                                                                                                  const int32_t* __restrict__ nact, int nsx) {
  __shared__ ShadeSmem<REC> sm;
  shade_bwd_tile<IMG, REC>(sm, blockIdx.x, A, order, nact, nsx);
}

// FUSED BACKWARD LAUNCH: the shading backward and the silhouette backward (csrc/raster_body.h, MODE 2) as ONE grid.  As two kernels on
// two streams they could not share a CU (3 x 168 VGPRs and 3 x 40 KB of LDS of the shader leave no room for a rasteriser workgroup), so
// "concurrent" meant taking turns: 0.30 + 0.12 ms alone, 0.39 ms together — and the second stream cost a fork and a join (~10 us each).
// In one grid the two kinds of workgroup co-reside, the rasteriser's waves issue while the shader's wait.  Workgroups alternate in
// groups of 8 (one per XCD): ids 16 g + 0..7 = shading tiles 8 g + 0..7, ids 16 g + 8..15 = silhouette tiles 8 g + 0..7, so a tile's
// XCD (real id & 7) is the same for both kinds and for the stand-alone launches.
struct SilBwdArgs {
  const FaceRec* recs; const float4* bbs; const int32_t* bins; const int32_t* cnt; const int32_t* faces;
  const float* alpha; const float* g_alpha; float* g_ndc; int F; float blur, sigma;
};
union FusedSmem {
  ShadeSmem<false> sh;
  rb::RasterSmem<2, false> rs;
  __device__ FusedSmem() {}
};
// (its own occupancy target: the rasteriser tiles it hosts want 168 VGPRs; under the shader's 4-waves-per-SIMD cap of 128 they would spill)
#ifndef FUSED_BWD_OCC
#define FUSED_BWD_OCC 3
#endif
__global__ void __launch_bounds__(256, FUSED_BWD_OCC) fused_bwd_kernel(const harp_shade_args A, const SilBwdArgs R, const int32_t* __restrict__ order,
                                                                       const int32_t* __restrict__ nact, int nsx) {
  __shared__ FusedSmem sm;
  const unsigned g = blockIdx.x >> 4, r = blockIdx.x & 15;
  const unsigned v = g * 8 + (r & 7);
  if (r < 8) {
    shade_bwd_tile<false, false>(sm.sh, v, A, order, nact, nsx);
  } else {
    rb::raster_tile<2>(sm.rs, v, R.recs, R.bbs, R.bins, R.cnt, order, nact, A.B, R.F, A.S, nsx, R.blur, R.sigma, nullptr, nullptr, (float*)R.alpha,
                       R.g_alpha, R.faces, A.V, R.g_ndc, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr);
  }
}

}  // namespace

// harp_shade_sil_bwd (C ABI below): both backward passes of the camera view in one launch
int harp_detail_fused_bwd(const harp_shade_args& a, const void* ws, const int32_t* faces, float blur, float sigma, const float* alpha,
                          const float* g_alpha, hipStream_t stream) {
  const RasterWs W = raster_ws_split((void*)ws, a.B, a.F, a.S);
  SilBwdArgs R;
  R.recs = W.recs; R.bbs = W.bbs; R.bins = W.bins; R.cnt = W.cnt; R.faces = faces; R.alpha = alpha; R.g_alpha = g_alpha; R.g_ndc = a.g_ndc;
  R.F = a.F; R.blur = blur; R.sigma = sigma;
  hipLaunchKernelGGL(fused_bwd_kernel, dim3(2 * tile_grid(a.B, W.nsx)), dim3(256), 0, stream, a, R, (const int32_t*)W.order, (const int32_t*)W.nact, W.nsx);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

// launched by harp_shade_bwd (shade.hip)
int harp_detail_shade_bwd_wave(const harp_shade_args& a, const int32_t* order, const int32_t* nact, int nsx, unsigned grid, hipStream_t stream) {
  // HARP_SHADE_LDS_PAD=<bytes> (timing experiments only) adds dynamic LDS to the launch to LOWER the number of resident workgroups:
  // how the kernel's time scales with occupancy (round 3: t = 0.134 + 0.454 / n ms for n workgroups per CU)
  static const unsigned pad = [] { const char* e = getenv("HARP_SHADE_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
  const bool img = a.g_rgb == nullptr && a.rgb != nullptr;
  if (a.trec != nullptr) {
    if (img) hipLaunchKernelGGL((shade_bwd_wave_kernel<true, true>), dim3(grid), dim3(256), pad, stream, a, order, nact, nsx);
    else hipLaunchKernelGGL((shade_bwd_wave_kernel<false, true>), dim3(grid), dim3(256), pad, stream, a, order, nact, nsx);
  } else if (img) hipLaunchKernelGGL((shade_bwd_wave_kernel<true, false>), dim3(grid), dim3(256), pad, stream, a, order, nact, nsx);
  else hipLaunchKernelGGL((shade_bwd_wave_kernel<false, false>), dim3(grid), dim3(256), pad, stream, a, order, nact, nsx);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

#ifdef SHADE_STAMPS
extern "C" int harp_debug_shade_stamps(unsigned long long* host16) {
  static unsigned long long all[4096][16];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_shade_stamps), sizeof(all)) != hipSuccess) return -1;
  for (int k = 0; k < 16; ++k) { host16[k] = 0; for (int i = 0; i < 4096; ++i) host16[k] += all[i][k]; }
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_shade_stamps)) != hipSuccess) return -1;
  return hipMemset(dev, 0, sizeof(all)) == hipSuccess ? 0 : -1;
}
#endif
