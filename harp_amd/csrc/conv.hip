// The perceptual term of the appearance stage on the matrix cores: the ten 3x3 / pad 1 / stride 1 convolutions of
// VGG16 features[0:23] (reference: model/vgg.py:10-56, built at optimize_sequence.py:405, used at :546-547 with weight 1.0, :419), their
// data gradients (the filters are frozen: requires_grad=False, model/vgg.py:34-36 — no weight gradient exists), and everything between
// them (bias, ReLU, 2x2 max pool, the L1 against the target frames' features, ReLU / pool backward) fused into the epilogues.
//
// Activations are NHWC float32 in HBM.  One workgroup (4 waves) owns a 16x16-pixel x 64-channel output tile and walks the input channels
// in chunks of 16: the (16+2)^2 x 16 input patch and the 9 x 16 x 64 filter slab of the chunk are staged in LDS once and every one of the
// 9 taps reads its shifted window out of the same patch (implicit GEMM: M = pixels, N = output channels, K = 9 x Cin).  Each wave holds a
// 16x4-pixel x 64-channel block of the tile as 2 x 2 accumulators of a 32x32 MFMA.  The 32 pixels of an MFMA row block are eight 2x2
// squares (4 across, 2 down) numbered so that the four accumulator registers r = 4q..4q+3 of a lane are one 2x2 square: ReLU + max pool
// (and its backward, the arg-max routing) need no cross-lane traffic.
//
// Two arithmetic modes (harp_conv3x3_args.precision):
//   0  v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate — bitwise a float32 fma chain (MI355X_MICROARCH.md), 157 TFLOP/s peak.
//   1  three-term bf16 split on v_mfma_f32_32x32x16_bf16: x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a.b ~ hi.hi + hi.lo + lo.hi
//      (the dropped lo.lo is 2^-18 relative), float32 accumulate: ~16 mantissa bits per product (the reference's stack — torch 1.11 + cuDNN,
//      requirements.txt:81 — runs these convolutions with TF32 allowed, 10 bits), 3/16 of the float32 MFMA time.  The split of the
//      activations happens while the patch is staged; the filters are split once (harp_conv3x3_pack_filters).
#include "harp_common.h"
#include "harp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kCT = 16;                   // output tile: 16 x 16 pixels
constexpr int kCK = 16;                   // input channels per staged chunk
constexpr int kCN = 64;                   // output channels per workgroup
constexpr int kPatch = kCT + 2;           // 18: tile + 1-pixel halo
// LDS patch: 4 planes of [18 rows][24 pixels][16 B].  float32 mode: plane q = channels 4q..4q+3 of the chunk; bf16 mode: planes 0/1 = hi of
// channels 0-7 / 8-15, planes 2/3 = lo.  A lane's A fragment is ONE 16-byte read at (its pixel + tap offset) of the plane of its k half
// (lanes 0-31 / 32-63).  Row pitch 24 pixels = 96 banks = 32 mod 64: the 16 lanes the LDS serves per cycle (ds_read_b128 lane groups)
// touch two rows x four 2-pixel squares whose 16-B slots fall on distinct banks for every tap (checked in tools/dev/conv_banks.py).
constexpr int kRow = 24;
constexpr int kPlane = kPatch * kRow + 2; // +2 float4: consecutive planes 8 banks apart, so the staging stores of one pixel's 4 quads do not collide
constexpr int kInF4 = 4 * kPlane;
constexpr int kWF4 = 9 * 4 * kCN;         // filter slab of one (64 output channels, 16 input channels) pair: [tap][plane][co][16 B] = 36 864 B
constexpr int kLdsBytes = (kInF4 + kWF4) * 16;
constexpr int kUnits = (kPatch * kPatch * 4 + 255) / 256;   // 16-byte staging units per thread (6; the last one is mostly idle)
// SUB8 form (bounded mode at the coarse levels, harp_conv3x3_args.tile_side == 8): the workgroup's four waves own four INDEPENDENT 8x8-pixel
// tiles of the frame's list (x 64 channels, the same filter slab); a wave stages its own 10x10 patch and its two MFMA row blocks are the
// upper and the lower 8x4 half of its tile.  Two waves share a set of planes [10 rows][24 pixels], 12 pixels apart: the same row pitch, hence
// the same conflict-free fragment reads.
constexpr int kPatch8 = 10;
constexpr int kPlane8 = kPatch8 * kRow + 2;
constexpr int kInF4_8 = 2 * 4 * kPlane8;
constexpr int kUnits8 = (kPatch8 * kPatch8 * 4 + 63) / 64;   // 16-byte staging units per LANE (7)
constexpr int kLdsBytes8 = (kInF4_8 + kWF4) * 16;

// experiment switches (tools/dev/build_variant.sh): CONV_PIPE 0 = the compiler's own placement of the fragment reads, 1 = reads of step s + 1
// issued before the MFMAs of step s, pinned by sched_barrier, 2 = the same without the pins.  Measured on one box, 256 -> 256 channels at
// 128^2 x 8 images (profiles/r05_conv_variants.txt): float32 94.5 / 93.9 / 93.3 TFLOP/s, bf16 split 203 / 196 / 202 — the explicit pipeline
// buys nothing; neither does removing the staging or the global fetch altogether (94.6 / 207): the loop is bound by MFMA issue.
#ifndef CONV_PIPE
#define CONV_PIPE 0
#endif
#ifndef CONV_ORDER
#define CONV_ORDER 0                      // order of a step's MFMAs over the four accumulators: 0 round-robin, 1 two at a time, 2 one at a time
#endif
#ifndef CONV_PRIO
#define CONV_PRIO 0
#endif
#ifndef CONV_WAVES
#define CONV_WAVES 2                      // waves per SIMD the register budget is sized for (2 workgroups per CU)
#endif
enum { EPI_RELU = 0, EPI_RELU_TAP = 1, EPI_GATE = 2, EPI_UNPOOL = 3 };

__device__ __forceinline__ float sgn(float d) { return (float)((d > 0.f) - (d < 0.f)); }

// ---- filter packing ---------------------------------------------------------------------------------------------------------------
// w (Cout_src, Cin_src, 3, 3) in torch's layout -> slabs [Cout/64][Cin/16] of kWF4 float4 each, zero-padded to the multiples.
// transpose != 0: the filters of the DATA GRADIENT, a convolution of the output gradient with w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx].
__global__ void pack_filters_kernel(const float* __restrict__ w, int Cout_src, int Cin_src, int Cout, int Cin, int transpose, int precision,
                                    float* __restrict__ packed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)Cout * Cin * 9;
  if (i >= total) return;
  // one thread per (slab, tap, co in block, channel in chunk)
  const int c = i % kCK, co_l = (i / kCK) % kCN, tap = (i / (kCK * kCN)) % 9;
  const size_t slab = i / (kCK * kCN * 9);
  const int nchunk = Cin / kCK;
  const int cb = slab / nchunk, cc = slab % nchunk;
  const int co = cb * kCN + co_l, ci = cc * kCK + c;
  float v = 0.f;
  if (!transpose) {
    if (co < Cout_src && ci < Cin_src) v = w[((size_t)co * Cin_src + ci) * 9 + tap];
  } else {
    if (ci < Cout_src && co < Cin_src) v = w[((size_t)ci * Cin_src + co) * 9 + (8 - tap)];
  }
  if (precision == 0) {
    packed[slab * (kWF4 * 4) + ((size_t)(tap * 4 + (c >> 2)) * kCN + co_l) * 4 + (c & 3)] = v;
  } else {
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    __bf16* p = (__bf16*)packed + slab * (kWF4 * 8);
    p[((size_t)(tap * 4 + (c >> 3)) * kCN + co_l) * 8 + (c & 7)] = hi;
    p[((size_t)(tap * 4 + 2 + (c >> 3)) * kCN + co_l) * 8 + (c & 7)] = lo;
  }
}

// ---- the convolution ----------------------------------------------------------------------------------------------------------------
template <int PREC, int EPI, bool SUB8 = false>
__global__ __launch_bounds__(256, CONV_WAVES) void conv3x3_kernel(const harp_conv3x3_args a, const int tiles_x, const int tiles_y) {
  extern __shared__ float4 smem[];          // ONE LDS object (a second one makes hipcc drain vmcnt before every ds_read, cdna_hip_programming.md §5)
  float4* s_in = smem;
  float4* s_w = smem + (SUB8 ? kInF4_8 : kInF4);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, half = lane >> 5, m = lane & 31;
  constexpr int kT = SUB8 ? 8 : kCT, kP = SUB8 ? kPatch8 : kPatch, kPl = SUB8 ? kPlane8 : kPlane, kU = SUB8 ? kUnits8 : kUnits;

  const int ncb = a.Cout / kCN, nchunk = a.Cin / kCK;
  int id = blockIdx.x;
  const int cb = id % ncb; id /= ncb;
  int tx, ty, n;
  int x0, y0;
  bool active = true;                       // SUB8: this wave holds a tile of the list (the last group of a frame may not be full)
  if (SUB8) {
    const int groups = (a.max_tiles + 3) >> 2;
    const int i = 4 * (id % groups) + wv;
    n = id / groups;
    const int r = a.target_row ? a.target_row[n] : n;
    const int cnt = a.tile_count[r];
    if (4 * (id % groups) >= cnt) return;                          // (the whole workgroup)
    active = i < cnt;
    const int tile = a.tile_list[(size_t)r * a.max_tiles + (active ? i : 0)];
    ty = tile / a.tile_pitch; tx = tile - ty * a.tile_pitch;
    y0 = ty * kT - (a.tile_origin ? a.tile_origin[2 * r] : 0);
    x0 = tx * kT - (a.tile_origin ? a.tile_origin[2 * r + 1] : 0);
  } else if (a.tile_list) {
    // bounded mode: slot i of image n's frame; frames hold different numbers of tiles, the grid is sized for the largest.  The frame's tile
    // grid may be shifted by an (even) origin so that its tiles hug the frame's active region: tile (ty, tx) covers pixels [16 ty - oy, +16)
    const int i = id % a.max_tiles;
    n = id / a.max_tiles;
    const int r = a.target_row ? a.target_row[n] : n;
    if (i >= a.tile_count[r]) return;
    const int tile = a.tile_list[(size_t)r * a.max_tiles + i];
    ty = tile / a.tile_pitch; tx = tile - ty * a.tile_pitch;
    y0 = ty * kCT - (a.tile_origin ? a.tile_origin[2 * r] : 0);
    x0 = tx * kCT - (a.tile_origin ? a.tile_origin[2 * r + 1] : 0);
  } else {
    tx = id % tiles_x; id /= tiles_x;
    ty = id % tiles_y;
    n = id / tiles_y;
    x0 = tx * kCT; y0 = ty * kCT;
  }
  const size_t row = a.target_row ? (size_t)a.target_row[n] : (size_t)n;
  const int H = a.H, W = a.W;
  const int Cin = a.in_channels > 0 ? a.in_channels : a.Cin;     // channels per pixel in memory (the rest of a.Cin reads as zero)
  const float* __restrict__ in_n = a.in + (size_t)n * H * W * Cin;
  const float* __restrict__ alt_n = a.in_alt ? a.in_alt + row * H * W * Cin : nullptr;
  const float4* __restrict__ wslab = (const float4*)a.filters + (size_t)cb * nchunk * kWF4;

  // bounded mode: which cells (8 << shift input pixels square) under the patch hold values of THIS pass; the patch spans at most 4 x 4 cells.
  // Wave-uniform: 16 scalar loads, one bit each.
  unsigned cellmask = 0xffffu;
  // cells of in_valid: the producer's tiles in INPUT pixels — (8 << in_valid_shift) for a producer on 16-pixel tiles, in_valid_cell
  // (4, 8 or 16) when given: a producer on 8-pixel tiles, behind a pool (shift 0) half of that
  const int csh = a.in_valid_cell ? 31 - __builtin_clz(a.in_valid_cell) : 3 + a.in_valid_shift;
  // origin of the producer's tile grid in INPUT pixels (a producer behind a pool runs at twice the resolution: its even origin halves)
  const int poy = a.in_valid_origin ? a.in_valid_origin[2 * row] >> (1 - a.in_valid_shift) : 0;
  const int pox = a.in_valid_origin ? a.in_valid_origin[2 * row + 1] >> (1 - a.in_valid_shift) : 0;
  const int cy0 = (max(y0 - 1, 0) + poy) >> csh, cx0 = (max(x0 - 1, 0) + pox) >> csh;
  if (a.in_valid) {
    const int pitch = a.in_valid_pitch;
    const int32_t* __restrict__ v = a.in_valid + row * pitch * pitch;
    cellmask = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int cy = cy0 + (c >> 2), cx = cx0 + (c & 3);
      if (cy < pitch && cx < pitch && v[cy * pitch + cx]) cellmask |= 1u << c;
    }
  }

  // staging units of this thread: unit u = (patch pixel u >> 2, channel quad u & 3): four consecutive lanes fetch one pixel's 64 bytes
  // (SUB8: the units of this WAVE's own patch, lane by lane; its planes start at sbase, its columns at 12 * (wv & 1))
  const int sbase = SUB8 ? (wv >> 1) * 4 * kPlane8 + 12 * (wv & 1) : 0;
  int goff[kU], lidx[kU];
  unsigned use_alt = 0;                    // bit j: unit j lies in a cell this pass did not write -> read in_alt (or zero)
#pragma unroll
  for (int j = 0; j < kU; ++j) {
    const int u = SUB8 ? j * 64 + lane : j * 256 + t, pix = u >> 2, q = u & 3;
    goff[j] = -1; lidx[j] = -1;
    if (pix < kP * kP && active) {
      const int py = pix / kP, px = pix - py * kP;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (PREC == 0) lidx[j] = sbase + q * kPl + py * kRow + px;                       // float4 index
      else lidx[j] = ((sbase + (q >> 1) * kPl + py * kRow + px) << 1) | (q & 1);      // 8-byte index of the hi half; lo is 2 planes further
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        goff[j] = (gy * W + gx) * Cin + 4 * q;
        const int c = ((((gy + poy) >> csh) - cy0) << 2) | (((gx + pox) >> csh) - cx0);
        if (!((cellmask >> c) & 1u)) {
          if (alt_n) use_alt |= 1u << j;
          else goff[j] = -1;
        }
      }
    }
  }
  // (named registers, not an array: hipcc left a 9 x float4 array in SCRATCH — 160 B per lane, every chunk's filter fetch waited for right
  //  behind its loads to be stored there — although every index is a compile-time constant after unrolling)
  float4 rin[kU], rw0, rw1, rw2, rw3, rw4, rw5, rw6, rw7, rw8;
#define HARP_RW_EACH(X) X(0, rw0) X(1, rw1) X(2, rw2) X(3, rw3) X(4, rw4) X(5, rw5) X(6, rw6) X(7, rw7) X(8, rw8)
  // (out-of-image units read the image's first bytes and are zeroed when they are STAGED: a select instead of a branch around every load, and
  //  placed behind the chunk's MFMAs — a select right behind the load made the wave wait for the fetch before it started the chunk's MFMAs:
  //  120 instead of 98 TFLOP/s float32, 365 instead of 208 bf16 split without it, profiles/r05_conv_variants.txt)
  auto unit_ok = [&](int j, int cc) { return goff[j] >= 0 && cc * kCK + 4 * (t & 3) < Cin; };      // (unit j's channel quad = (j * 256 + t) & 3 = (j * 64 + lane) & 3 = t & 3)
  auto fetch = [&](int cc) {
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const float* __restrict__ src = ((use_alt >> j) & 1u) ? alt_n : in_n;
      rin[j] = *(const float4*)(src + (unit_ok(j, cc) ? goff[j] + cc * kCK : 0));
    }
#define HARP_RW_LOAD(j, r) r = wslab[(size_t)cc * kWF4 + j * 256 + t];
    HARP_RW_EACH(HARP_RW_LOAD)
#undef HARP_RW_LOAD
  };
  auto stage = [&](int cc) {
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      if (lidx[j] < 0) continue;
      const float4 v = unit_ok(j, cc) ? rin[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (PREC == 0) {
        s_in[lidx[j]] = v;
      } else {
        bf16x4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        bf16x4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]), (__bf16)(v.z - (float)hi[2]), (__bf16)(v.w - (float)hi[3])};
        uint2* s8 = (uint2*)s_in;
        s8[lidx[j]] = __builtin_bit_cast(uint2, hi);
        s8[lidx[j] + 4 * kPl] = __builtin_bit_cast(uint2, lo);
      }
    }
#define HARP_RW_STORE(j, r) s_w[j * 256 + t] = r;
    HARP_RW_EACH(HARP_RW_STORE)
#undef HARP_RW_STORE
  };

  // this lane's pixel in each of the wave's two 8x4 row blocks: m = 4 * square + (dy, dx); squares 4 across, 2 down
  // (SUB8: the two row blocks are rows 0-3 and 4-7 of the wave's own 8x8 tile)
  const int ly = (SUB8 ? 0 : 4 * wv) + 2 * (m >> 4) + ((m >> 1) & 1);
  const int lx = 2 * ((m >> 2) & 3) + (m & 1);
  const int pixA = sbase + ly * kRow + lx;
  constexpr int kBlk1 = SUB8 ? 4 * kRow : 8;          // the second row block: 4 rows down (SUB8) or 8 pixels to the right

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#if CONV_PRIO == 1
  // the two waves that share a SIMD (one of each resident workgroup) get different static priorities, by the parity of their hardware wave
  // slot: with equal priorities the matrix pipe serves them alternately, so both run out of fragments and wait for the LDS at the same time
  if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(1);
#endif
  fetch(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();                      // every wave is done with the previous chunk's patch and slab
    stage(cc);
    __syncthreads();
    if (cc + 1 < nchunk) fetch(cc + 1);   // in flight under the chunk's MFMAs
    if (SUB8 && !active) continue;        // (a wave without a tile only helps staging the slab)
    // Software pipeline over the chunk's steps (float32: 18 = 9 taps x 2 k groups of 8 channels; bf16: 9 taps of 16 channels): the
    // fragments of step s + 1 are read from LDS before the MFMAs of step s issue, so one wave alone covers its LDS latency (the compiler's
    // own schedule read each step's fragments right in front of its MFMAs: MFMA pipe 78 % / 37 % busy, profiles/r05_a_pmc_sq_conv_*).
    constexpr int kSteps = PREC == 0 ? 18 : 9;
    constexpr int kFrag = PREC == 0 ? 4 : 8;
    float4 fr[2][kFrag];
    auto read_frags = [&](float4* f, int step) {
      if (PREC == 0) {
        const int tap = step >> 1, g = step & 1;
        const int plane = 2 * g + half, toff = (tap / 3) * kRow + (tap % 3);
        f[0] = s_in[plane * kPl + pixA + toff]; f[1] = s_in[plane * kPl + pixA + toff + kBlk1];
        f[2] = s_w[(tap * 4 + plane) * kCN + m]; f[3] = s_w[(tap * 4 + plane) * kCN + 32 + m];
      } else {
        const int tap = step, toff = (tap / 3) * kRow + (tap % 3);
        f[0] = s_in[half * kPl + pixA + toff]; f[1] = s_in[half * kPl + pixA + toff + kBlk1];                       // A hi
        f[2] = s_in[(2 + half) * kPl + pixA + toff]; f[3] = s_in[(2 + half) * kPl + pixA + toff + kBlk1];           // A lo
        f[4] = s_w[(tap * 4 + half) * kCN + m]; f[5] = s_w[(tap * 4 + half) * kCN + 32 + m];                          // B hi
        f[6] = s_w[(tap * 4 + 2 + half) * kCN + m]; f[7] = s_w[(tap * 4 + 2 + half) * kCN + 32 + m];                  // B lo
      }
    };
#if CONV_PIPE
    read_frags(fr[0], 0);
#endif
#ifdef CONV_ROLL
#pragma unroll CONV_ROLL
#else
#pragma unroll
#endif
    for (int step = 0; step < kSteps; ++step) {
#if CONV_PIPE
      float4* f = fr[step & 1];
      if (step + 1 < kSteps) read_frags(fr[(step + 1) & 1], step + 1);
#else
      float4* f = fr[0];
      read_frags(f, step);
#endif
#if CONV_PIPE == 1
      __builtin_amdgcn_sched_barrier(0);
#endif
#if CONV_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      if (PREC == 0) {
        const float a0[4] = {f[0].x, f[0].y, f[0].z, f[0].w}, a1[4] = {f[1].x, f[1].y, f[1].z, f[1].w};
        const float b0[4] = {f[2].x, f[2].y, f[2].z, f[2].w}, b1[4] = {f[3].x, f[3].y, f[3].z, f[3].w};
#if CONV_ORDER == 0
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
        }
#elif CONV_ORDER == 1      // two accumulators at a time
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
        }
#else                      // one accumulator at a time (dependent chains of 4)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
#endif
      } else {
        const bf16x8 Ah0 = __builtin_bit_cast(bf16x8, f[0]), Ah1 = __builtin_bit_cast(bf16x8, f[1]);
        const bf16x8 Al0 = __builtin_bit_cast(bf16x8, f[2]), Al1 = __builtin_bit_cast(bf16x8, f[3]);
        const bf16x8 Bh0 = __builtin_bit_cast(bf16x8, f[4]), Bh1 = __builtin_bit_cast(bf16x8, f[5]);
        const bf16x8 Bl0 = __builtin_bit_cast(bf16x8, f[6]), Bl1 = __builtin_bit_cast(bf16x8, f[7]);
        // the two small terms first, the leading one last
#if CONV_ORDER == 0
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl1, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh1, acc[1][1], 0, 0, 0);
#elif CONV_ORDER == 1
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh1, acc[0][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl1, acc[0][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh1, acc[1][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl1, acc[1][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh1, acc[1][1], 0, 0, 0);
#else
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh0, acc[0][0], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl0, acc[0][0], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh1, acc[0][1], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl1, acc[0][1], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh0, acc[1][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl0, acc[1][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh1, acc[1][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl1, acc[1][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh1, acc[1][1], 0, 0, 0);
#endif
      }
#if CONV_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
#if CONV_PIPE == 1
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }

  // ---- epilogue.  Accumulator register r of lane (half, m) in row block i / column block j: output channel co = 64 cb + 32 j + m;
  // pixel = square 2 (r >> 2) + half of the row block (4 across, 2 down), corner (dy, dx) = ((r >> 1) & 1, r & 1).
  const int Cout = a.Cout;
  float lsum = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = cb * kCN + 32 * j + m;
    const float bias = (EPI <= EPI_RELU_TAP && a.bias) ? a.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sq = 2 * q + half;
        // top-left pixel of the 2x2 square (even, even)
        const int gy = y0 + (SUB8 ? 4 * i : 4 * wv) + 2 * (sq >> 2), gx = x0 + (SUB8 ? 0 : 8 * i) + 2 * (sq & 3);
        if (gy >= H || gx >= W || gy < 0 || gx < 0 || !active) continue;               // (a shifted tile grid reaches beyond the image on all sides)
        if (EPI == EPI_RELU || EPI == EPI_RELU_TAP) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[i][j][4 * q + c] + bias, 0.f);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yy = gy + (c >> 1), xx = gx + (c & 1);
            if (yy >= H || xx >= W) continue;
            const size_t o = ((size_t)yy * W + xx) * Cout + co;
            if (a.out) a.out[(size_t)n * H * W * Cout + o] = v[c];
            if (EPI == EPI_RELU_TAP) {
              const float d = v[c] - a.target[row * H * W * Cout + o];
              lsum += fabsf(d);
              a.g_tap[(size_t)n * H * W * Cout + o] = v[c] > 0.f ? a.tap_scale * sgn(d) : 0.f;
            }
          }
          if (a.pooled)   // torch.nn.MaxPool2d(2, 2) (model/vgg.py slices 2-4 open with it): H, W even here (checked on the host)
            a.pooled[(((size_t)n * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * Cout + co] = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        } else if (EPI == EPI_GATE) {
          // ReLU backward: the gradient passes where the forward activation was positive (threshold_backward)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yy = gy + (c >> 1), xx = gx + (c & 1);
            if (yy >= H || xx >= W) continue;
            const size_t o = (((size_t)n * H + yy) * W + xx) * Cout + co;
            a.out[o] = a.gate[o] > 0.f ? acc[i][j][4 * q + c] : 0.f;
          }
        } else {
          // max-pool backward + ReLU backward + the tap's own L1 gradient: this convolution's pixels are the POOLED pixels; each routes its
          // gradient to the first maximum of its 2x2 window in row-major order (max_pool2d's arg-max), if that activation was positive.
          // a.out (N, 2H, 2W, Cout) holds the tap gradient on entry; every element belongs to exactly one window: no atomics.
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yy = gy + (c >> 1), xx = gx + (c & 1);
            if (yy >= H || xx >= W) continue;
            const size_t o00 = (((size_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * Cout + co;
            const size_t dx = Cout, dy = (size_t)2 * W * Cout;
            const float g00 = a.gate[o00], g01 = a.gate[o00 + dx], g10 = a.gate[o00 + dy], g11 = a.gate[o00 + dy + dx];
            float best = g00; size_t ob = o00;
            if (g01 > best) { best = g01; ob = o00 + dx; }
            if (g10 > best) { best = g10; ob = o00 + dy; }
            if (g11 > best) { best = g11; ob = o00 + dy + dx; }
            // (bounded mode: windows in output tiles this pass does not own are left alone — nothing downstream reads them)
            const int osh = a.out_valid_cell == 8 ? 3 : 4;          // `out`'s tiles: 16 pixels, or 8
            if (a.out_valid && !a.out_valid[(row * a.out_valid_pitch + ((2 * yy + (a.out_valid_origin ? a.out_valid_origin[2 * row] : 0)) >> osh)) * a.out_valid_pitch +
                                            ((2 * xx + (a.out_valid_origin ? a.out_valid_origin[2 * row + 1] : 0)) >> osh)]) continue;
            if (best > 0.f) a.out[ob] += acc[i][j][4 * q + c];
          }
        }
      }
    }
  }
  if (EPI == EPI_RELU_TAP) {
    __syncthreads();                       // the patch is dead: its first floats carry the block sum (no second LDS object, see above)
    const float s = block_sum_256(lsum, (float*)smem);
    if (t == 0 && s != 0.f) atomicAdd(a.loss, (double)s * (double)a.tap_scale);
  }
}

template <int PREC, int EPI, bool SUB8>
int launch_conv(const harp_conv3x3_args& a, hipStream_t stream) {
  static bool ready = false;               // per instantiation: raise the dynamic-LDS limit once, not per launch (and never inside a capture)
  auto kern = conv3x3_kernel<PREC, EPI, SUB8>;
  constexpr int lds = SUB8 ? kLdsBytes8 : kLdsBytes;
  if (!ready) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HARP_ERR_LAUNCH;
    ready = true;
  }
  const int tiles_x = (a.W + kCT - 1) / kCT, tiles_y = (a.H + kCT - 1) / kCT;
  // (SUB8: four tiles of the list per workgroup)
  const size_t blocks = (SUB8 ? (size_t)((a.max_tiles + 3) >> 2) : a.tile_list ? (size_t)a.max_tiles : (size_t)tiles_x * tiles_y) * a.N * (a.Cout / kCN);
  if (blocks == 0 || blocks > 0x7fffffffu) return HARP_ERR_ARG;
#ifdef CONV_LDS_PAD
  static bool padded = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds + CONV_LDS_PAD), true);
  (void)padded;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds + CONV_LDS_PAD, stream, a, tiles_x, tiles_y);
#else
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a, tiles_x, tiles_y);
#endif
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

template <int PREC>
int launch_conv_prec(const harp_conv3x3_args& a, hipStream_t stream) {
  const bool sub8 = a.tile_side == 8;
  switch (a.epilogue) {
    case EPI_RELU: return sub8 ? launch_conv<PREC, EPI_RELU, true>(a, stream) : launch_conv<PREC, EPI_RELU, false>(a, stream);
    case EPI_RELU_TAP: return sub8 ? launch_conv<PREC, EPI_RELU_TAP, true>(a, stream) : launch_conv<PREC, EPI_RELU_TAP, false>(a, stream);
    case EPI_GATE: return sub8 ? launch_conv<PREC, EPI_GATE, true>(a, stream) : launch_conv<PREC, EPI_GATE, false>(a, stream);
    case EPI_UNPOOL: return sub8 ? launch_conv<PREC, EPI_UNPOOL, true>(a, stream) : launch_conv<PREC, EPI_UNPOOL, false>(a, stream);
  }
  return HARP_ERR_ARG;
}

// ---- the two ends of the stack: 3 image channels, vector ALU ------------------------------------------------------------------------
// x0 = image * mask (optimize_sequence.py:546-547: vgg(y_pred * mask) / vgg(y_true * mask)), written as 4 channels per pixel (r, g, b, 0);
// with y_true given, also the term's first row (model/vgg.py:41: the flattened input itself): *loss += scale0 * sum |x0 - y_true * mask|.
__global__ __launch_bounds__(256) void vgg_prep_kernel(const float* __restrict__ image, const int32_t* __restrict__ image_rows,
                                                       const float* __restrict__ mask, const int32_t* __restrict__ mask_rows,
                                                       const float* __restrict__ y_true, int S, float scale0, float4* __restrict__ x0,
                                                       double* __restrict__ loss) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  const size_t ir = image_rows ? (size_t)image_rows[n] : (size_t)n, mr = mask_rows ? (size_t)mask_rows[n] : (size_t)n;
  const int p = blockIdx.x * 256 + threadIdx.x;
  float l = 0.f;
  if (p < S * S) {
    const float m = mask[mr * S * S + p];
    const float* px = image + (ir * S * S + p) * 3;
    const float r = px[0] * m, g = px[1] * m, b = px[2] * m;
    x0[(size_t)n * S * S + p] = make_float4(r, g, b, 0.f);
    if (y_true) {
      const float* t = y_true + (mr * S * S + p) * 3;
      l = fabsf(r - t[0] * m) + fabsf(g - t[1] * m) + fabsf(b - t[2] * m);
    }
  }
  if (y_true) {
    const float s = block_sum_256(l, red);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss, (double)s * (double)scale0);
  }
}

// The data gradient of the first convolution (64 -> 3 channels) and the end of the chain:
//   g_x0 = conv(G, w0^T mirrored) + scale0 * sign(x0 - y_true * mask);  d term / d rgb = mask * g_x0;
//   g_rgb = covered ? g_rgb + weight * (d term / d rgb) : 0     (the photometric gradient buffer is only defined at covered pixels)
// One pixel per thread, 16x16-pixel tiles; the 64 channels of G pass through LDS in four 18x18x16 patches; the filter taps are
// wave-uniform (scalar loads from w0t (9,3,64), w0t[t][c][co] = w0[co][c][8 - t]).  *loss_out = the term's value (the double accumulator is complete when this kernel starts).
constexpr int kGRow = 20;                                   // LDS row pitch (float4) of the gradient patch
__global__ __launch_bounds__(256) void vgg_grad_image_kernel(const float* __restrict__ G, const float* __restrict__ w0t, const float* __restrict__ rgb,
                                                             const float* __restrict__ y_true, const float* __restrict__ mask,
                                                             const int32_t* __restrict__ rows, const int32_t* __restrict__ covered, int S, float scale0,
                                                             float weight, float* __restrict__ g_rgb, double* __restrict__ loss_acc,
                                                             float* __restrict__ loss_out, const int32_t* __restrict__ tiles,
                                                             const int32_t* __restrict__ origin, int pitch) {
  __shared__ float4 patch[4 * kPatch * kGRow];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int n = blockIdx.z, x0 = blockIdx.x * kCT, y0 = blockIdx.y * kCT;
  // (the accumulator is complete: every forward launch is behind this one on the stream.  It is handed back ZERO for the next call — the
  //  workspace starts zero-filled —, so the term needs no clear of its own: a captured hipMemsetAsync did not re-execute on replay, App. A)
  // (loss_acc == NULL: the term runs as two half batches on two streams and a launch behind their join does this, vgg_loss_finish_kernel)
  if (loss_acc && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && n == 0) {
    if (loss_out) loss_out[0] = (float)loss_acc[0];
    loss_acc[0] = 0.0;
  }
  const float* __restrict__ Gn = G + (size_t)n * S * S * 64;
  // bounded mode: G exists in the frame's active 16x16 tiles only (zero elsewhere); a tile that is not active has no gradient from the stack
  const size_t fr = rows ? (size_t)rows[n] : (size_t)n;
  const int32_t* __restrict__ act = tiles ? tiles + fr * pitch * pitch : nullptr;
  const int oy = (tiles && origin) ? origin[2 * fr] : 0, ox = (tiles && origin) ? origin[2 * fr + 1] : 0;
  auto in_active = [&](int gy, int gx) { return !act || act[((gy + oy) >> 4) * pitch + ((gx + ox) >> 4)] != 0; };
  // (this kernel's blocks sit on the fixed 16x16 grid; the frame's active tiles may be shifted: a block overlaps at most four of them)
  const int yb = min(y0 + kCT - 1, S - 1), xb = min(x0 + kCT - 1, S - 1);
  const bool active = in_active(y0, x0) || in_active(y0, xb) || in_active(yb, x0) || in_active(yb, xb);
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  for (int cc = 0; cc < (active ? 4 : 0); ++cc) {
    __syncthreads();
    for (int u = t; u < kPatch * kPatch * 4; u += 256) {
      const int pix = u >> 2, q = u & 3, py = pix / kPatch, px = pix - py * kPatch;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < S && gx >= 0 && gx < S && in_active(gy, gx))
        v = *(const float4*)(Gn + ((size_t)gy * S + gx) * 64 + cc * 16 + 4 * q);
      patch[(q * kPatch + py) * kGRow + px] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      // output channel c, input pixel (y + ky - 1, x + kx - 1), channel co of G: weight w0[co][c][2 - ky][2 - kx] = w0t[tap][c][co]
      // (wave-uniform: 48 scalar loads per tap and chunk; the tap loop stays rolled so that they are not all hoisted and spilled)
      const int ky = tap / 3, kx = tap - 3 * ky;
      const float* __restrict__ w = w0t + tap * 192 + cc * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g = patch[(q * kPatch + ty + ky) * kGRow + tx + kx];
        const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0 = fmaf(gv[e], w[4 * q + e], o0);
          o1 = fmaf(gv[e], w[64 + 4 * q + e], o1);
          o2 = fmaf(gv[e], w[128 + 4 * q + e], o2);
        }
      }
    }
  }
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx >= S || gy >= S) return;
  const size_t p = (size_t)gy * S + gx, o = ((size_t)n * S * S + p) * 3;
  if (covered && covered[(size_t)n * S * S + p] < 0) {
    g_rgb[o] = 0.f; g_rgb[o + 1] = 0.f; g_rgb[o + 2] = 0.f;
    return;
  }
  const size_t r = rows ? (size_t)rows[n] : (size_t)n;
  const float m = mask[r * S * S + p];
  const float* yt = y_true + (r * S * S + p) * 3;
  const float d0 = (rgb[o] - yt[0]) * m, d1 = (rgb[o + 1] - yt[1]) * m, d2 = (rgb[o + 2] - yt[2]) * m;
  g_rgb[o] += weight * m * (o0 + scale0 * sgn(d0));
  g_rgb[o + 1] += weight * m * (o1 + scale0 * sgn(d1));
  g_rgb[o + 2] += weight * m * (o2 + scale0 * sgn(d2));
}

__global__ void vgg_loss_finish_kernel(double* __restrict__ loss_acc, float* __restrict__ loss_out) {
  if (loss_out) loss_out[0] = (float)loss_acc[0];
  loss_acc[0] = 0.0;
}

// ---- workspace of the whole term ------------------------------------------------------------------------------------------------------
// VGG16 features[0:23]: convolution k -> (Cin, Cout, image side divisor)
constexpr int kVggCin[10] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512};
constexpr int kVggCout[10] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512};
constexpr int kVggDiv[10] = {1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
constexpr int kVggTap[4] = {1, 3, 6, 9};                     // relu1_2, relu2_2, relu3_3, relu4_3 (model/vgg.py:42-49)

struct VggWs {
  float4* x0;
  float* act[10];
  float* pool[3];
  float* g_tap[4];
  float* gbuf[2];
  double* loss;
  size_t bytes;
};
inline VggWs vgg_ws_split(void* ws, int N, int S, int with_gradient) {
  VggWs w;
  char* p = (char*)ws;
  auto take = [&](size_t floats) { char* r = p; p += (floats * 4 + 255) / 256 * 256; return (float*)r; };
  const size_t NS2 = (size_t)N * S * S;
  w.loss = (double*)take(64);
  w.x0 = (float4*)take(NS2 * 4);
  for (int k = 0; k < 10; ++k) w.act[k] = take(NS2 / (kVggDiv[k] * kVggDiv[k]) * kVggCout[k]);
  for (int k = 0; k < 3; ++k) w.pool[k] = take(NS2 / (4 << (2 * k)) * kVggCout[kVggTap[k]]);
  for (int k = 0; k < 4; ++k) w.g_tap[k] = with_gradient ? take(NS2 / (kVggDiv[kVggTap[k]] * kVggDiv[kVggTap[k]]) * kVggCout[kVggTap[k]]) : nullptr;
  for (int k = 0; k < 2; ++k) w.gbuf[k] = with_gradient ? take(NS2 * 64) : nullptr;
  w.bytes = (size_t)(p - (char*)ws);
  return w;
}

// bounded mode of the term (harp_vgg16_term_args.tiles[0] != NULL): per resolution level L = 0..3 (image side S >> L) the 16x16 tiles of
// every target frame in which the rendered image's activations can differ from the target frame's (the mask's support grown by the
// receptive field).  Outside them pred == target exactly, the L1 and its gradient vanish, and a convolution that needs an input pixel
// from there reads the TARGET frame's cached activation (forward) or zero (backward).
struct VggBound {
  const int32_t* tiles[4]; const int32_t* list[4]; const int32_t* count[4]; const int32_t* origin[4]; int max_tiles[4]; int pitch[4]; int side[4];
  const float* target_in[10];
  const int32_t* rows;
};
inline int vgg_level(int k) { return kVggDiv[k] == 1 ? 0 : kVggDiv[k] == 2 ? 1 : kVggDiv[k] == 4 ? 2 : 3; }

// forward pass of the stack over x0; out[k] (k < 10: convolution k's activation, 10..12: the three pooled maps) or the workspace slot;
// with target != NULL the tap layers take the L1 epilogue
int vgg_forward(const harp_vgg16* net, const VggWs& w, int N, int S, float* const* out, const float* const target[4],
                const int32_t* target_row, const float scale[5], const VggBound* bd, hipStream_t stream) {
  const float* in = (const float*)w.x0;
  int tap = 0;
  for (int k = 0; k < 10; ++k) {
    harp_conv3x3_args a = {};
    const bool is_tap = (tap < 4 && k == kVggTap[tap]);
    const int s = S / kVggDiv[k], lv = vgg_level(k);
    a.in = in; a.filters = net->filters[k]; a.bias = net->bias[k];
    a.N = N; a.H = s; a.W = s; a.Cin = (kVggCin[k] + kCK - 1) / kCK * kCK; a.Cout = kVggCout[k];
    a.in_channels = k == 0 ? 4 : 0;
    a.precision = net->precision;
    a.out = (out && out[k]) ? out[k] : w.act[k];
    a.epilogue = HARP_CONV_RELU;
    float* pooled = nullptr;
    if (is_tap) {
      if (tap < 3) pooled = a.pooled = (out && out[10 + tap]) ? out[10 + tap] : w.pool[tap];
      if (target) {
        a.epilogue = HARP_CONV_RELU_TAP;
        a.target = target[tap]; a.tap_scale = scale[tap + 1]; a.g_tap = w.g_tap[tap]; a.loss = w.loss;
      }
    }
    a.target_row = target_row;
    if (bd) {
      a.target_row = bd->rows;
      a.tile_list = bd->list[lv]; a.tile_count = bd->count[lv]; a.max_tiles = bd->max_tiles[lv];
      a.tile_origin = bd->origin[lv]; a.tile_pitch = bd->pitch[lv]; a.tile_side = bd->side[lv];
      if (k > 0) {                                  // (x0 is written everywhere)
        const bool behind_pool = (k == 2 || k == 4 || k == 7);
        const int pl = behind_pool ? lv - 1 : lv;
        a.in_valid = bd->tiles[pl]; a.in_valid_origin = bd->origin[pl]; a.in_valid_pitch = bd->pitch[pl]; a.in_valid_shift = behind_pool ? 0 : 1;
        a.in_valid_cell = behind_pool ? bd->side[pl] / 2 : bd->side[pl];          // the producer's tiles in this convolution's input pixels
        a.in_alt = bd->target_in[k];
      }
    }
    const int rc = harp_conv3x3(&a, stream);
    if (rc != HARP_OK) return rc;
    in = pooled ? pooled : a.out;
    if (is_tap) ++tap;
  }
  return HARP_OK;
}

void vgg_scales(const harp_vgg16* net, int N, int S, float scale[5]) {
  // L1Loss (mean) over the concatenated rows (model/vgg.py:51-55): n = N * (3 S^2 + 64 S^2 + 128 (S/2)^2 + 256 (S/4)^2 + 512 (S/8)^2)
  const double S2 = (double)S * S;
  const double n = (double)N * (3 * S2 + 64 * S2 + 128 * S2 / 4 + 256 * S2 / 16 + 512 * S2 / 64);
  for (int k = 0; k < 5; ++k) scale[k] = (float)(fabs((double)net->layer_w[k]) / n);
}

bool vgg_net_ok(const harp_vgg16* net, bool gradient) {
  if (!net || (net->precision != 0 && net->precision != 1)) return false;
  for (int k = 0; k < 10; ++k) {
    if (!net->filters[k] || !net->bias[k]) return false;
    if (gradient && k > 0 && !net->filters_t[k]) return false;
  }
  return !gradient || net->w0t;
}

}  // namespace

extern "C" {

size_t harp_conv3x3_filter_bytes(int Cout, int Cin) {
  const size_t co = (size_t)(Cout + kCN - 1) / kCN, ci = (size_t)(Cin + kCK - 1) / kCK;
  return co * ci * kWF4 * 16;
}

int harp_conv3x3_pack_filters(const float* w, int Cout, int Cin, int transpose, int precision, void* packed, hipStream_t stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || (precision != 0 && precision != 1)) return HARP_ERR_ARG;
  const int co_src = Cout, ci_src = Cin;
  const int out_c = transpose ? Cin : Cout, in_c = transpose ? Cout : Cin;          // channels of the packed convolution
  const int Cout_p = (out_c + kCN - 1) / kCN * kCN, Cin_p = (in_c + kCK - 1) / kCK * kCK;
  const size_t total = (size_t)Cout_p * Cin_p * 9;
  hipLaunchKernelGGL(pack_filters_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, co_src, ci_src, Cout_p, Cin_p, transpose,
                     precision, (float*)packed);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

int harp_conv3x3(const harp_conv3x3_args* a, hipStream_t stream) {
  if (!a || !a->in || !a->filters || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return HARP_ERR_ARG;
  if (a->Cin % kCK || a->Cout % kCN) return HARP_ERR_ARG;
  if ((size_t)a->H * a->W * a->Cin > 0x7fffffffu) return HARP_ERR_ARG;               // 32-bit offsets inside one image
  if (a->in_channels < 0 || a->in_channels > a->Cin || (a->in_channels & 3)) return HARP_ERR_ARG;
  switch (a->epilogue) {
    case EPI_RELU: if (!a->out && !a->pooled) return HARP_ERR_ARG; break;
    case EPI_RELU_TAP: if (!a->target || !a->g_tap || !a->loss) return HARP_ERR_ARG; break;
    case EPI_GATE: case EPI_UNPOOL: if (!a->out || !a->gate) return HARP_ERR_ARG; break;
    default: return HARP_ERR_ARG;
  }
  if (a->pooled && ((a->H | a->W) & 1)) return HARP_ERR_ARG;
  if (a->tile_list && (!a->tile_count || a->max_tiles <= 0 || a->tile_pitch <= 0)) return HARP_ERR_ARG;
  if ((a->in_valid && a->in_valid_pitch <= 0) || (a->out_valid && a->out_valid_pitch <= 0)) return HARP_ERR_ARG;
  if (a->in_valid_shift < 0 || a->in_valid_shift > 1) return HARP_ERR_ARG;
  if ((a->tile_side != 0 && a->tile_side != 8 && a->tile_side != 16) || (a->tile_side == 8 && !a->tile_list)) return HARP_ERR_ARG;
  if ((a->in_valid_cell != 0 && a->in_valid_cell != 4 && a->in_valid_cell != 8 && a->in_valid_cell != 16) || (a->in_valid_cell == 4 && a->tile_side != 8) ||
      (a->out_valid_cell != 0 && a->out_valid_cell != 8 && a->out_valid_cell != 16)) return HARP_ERR_ARG;
  if (a->precision == 0) return launch_conv_prec<0>(*a, stream);
  if (a->precision == 1) return launch_conv_prec<1>(*a, stream);
  return HARP_ERR_ARG;
}

size_t harp_vgg16_ws_bytes(int N, int S, int with_gradient) {
  if (N <= 0 || S <= 0 || (S & 7)) return 0;
  // (harp_vgg16_term with side streams lays the workspace out as 2 - 4 parts of the batch, each with its own 256-byte paddings)
  size_t most = vgg_ws_split(nullptr, N, S, with_gradient).bytes;
  for (int parts = 2; parts <= 4 && parts <= N; ++parts) {
    size_t sum = 0;
    for (int i = 0; i < parts; ++i) sum += vgg_ws_split(nullptr, N / parts + (i < N % parts ? 1 : 0), S, with_gradient).bytes;
    if (sum > most) most = sum;
  }
  return most;
}

int harp_vgg16_features(const harp_vgg16* net, const float* image, const float* mask, const int32_t* rows, int N, int S, void* ws,
                        float* const* out, hipStream_t stream) {
  if (!vgg_net_ok(net, false) || !image || !mask || !ws || N <= 0 || S <= 0 || (S & 7) || !out) return HARP_ERR_ARG;
  for (int k = 0; k < 4; ++k)
    if (!out[kVggTap[k]]) return HARP_ERR_ARG;
  const VggWs w = vgg_ws_split(ws, N, S, 0);
  hipLaunchKernelGGL(vgg_prep_kernel, dim3((S * S + 255) / 256, N), dim3(256), 0, stream, image, rows, mask, rows, (const float*)nullptr, S, 0.f, w.x0,
                     w.loss);
  HARP_CHECK_LAUNCH();
  return vgg_forward(net, w, N, S, out, nullptr, nullptr, nullptr, nullptr, stream);
}

int harp_vgg16_term(const harp_vgg16* net, const harp_vgg16_term_args* t, hipStream_t stream) {
  if (!vgg_net_ok(net, true) || !t || !t->rgb || !t->y_true || !t->mask || !t->g_rgb || !t->ws || t->N <= 0 || t->S <= 0 || (t->S & 7))
    return HARP_ERR_ARG;
  for (int k = 0; k < 4; ++k)
    if (!t->target[k]) return HARP_ERR_ARG;
  const int N = t->N, S = t->S;
  float scale[5];
  vgg_scales(net, N, S, scale);
  // every argument is checked before the first launch: vgg_prep_kernel adds into the workspace's loss accumulator, which only the LAST
  // launch of the term hands back zeroed — an error return between the two would leave it dirty for the next call
  VggBound bd = {};
  const bool bounded = t->tiles[0] != nullptr;
  if (bounded) {
    if (!t->target_by_row || !t->rows) return HARP_ERR_ARG;
    for (int l = 0; l < 4; ++l) {
      if (!t->tiles[l] || !t->tile_list[l] || !t->tile_count[l] || !t->tile_origin[l] || t->max_tiles[l] <= 0 || t->tile_pitch[l] <= 0) return HARP_ERR_ARG;
      bd.tiles[l] = t->tiles[l]; bd.list[l] = t->tile_list[l]; bd.count[l] = t->tile_count[l]; bd.max_tiles[l] = t->max_tiles[l];
      bd.origin[l] = t->tile_origin[l]; bd.pitch[l] = t->tile_pitch[l];
      bd.side[l] = t->tile_side[l] ? t->tile_side[l] : 16;
      // (the image-side kernels sit on the 16-pixel grid; a 16-pixel level behind an 8-pixel one would see validity cells of 4 pixels, of which
      //  its 18-pixel patch spans more than the four a workgroup looks up)
      if ((bd.side[l] != 16 && bd.side[l] != 8) || (l == 0 && bd.side[l] != 16) || (l > 0 && bd.side[l] > bd.side[l - 1])) return HARP_ERR_ARG;
    }
    for (int k = 1; k < 10; ++k) {
      if (!t->target_in[k]) return HARP_ERR_ARG;
      bd.target_in[k] = t->target_in[k];
    }
  }
  // images [n0, n0 + Nh) of the batch on stream st with workspace w; `finish`: the last launch hands the loss out and its accumulator back zeroed
  auto run = [&](int n0, int Nh, const VggWs& w, hipStream_t st, bool finish) -> int {
    const size_t px = (size_t)n0 * S * S;
    const int32_t* rows = t->rows ? t->rows + n0 : nullptr;
    VggBound b = bd;
    b.rows = rows;
    hipLaunchKernelGGL(vgg_prep_kernel, dim3((S * S + 255) / 256, Nh), dim3(256), 0, st, t->rgb + px * 3, (const int32_t*)nullptr, t->mask, rows, t->y_true, S,
                       scale[0], w.x0, w.loss);
    HARP_CHECK_LAUNCH();
    int rc = vgg_forward(net, w, Nh, S, nullptr, t->target, t->target_by_row ? rows : nullptr, scale, bounded ? &b : nullptr, st);
    if (rc != HARP_OK) return rc;
    // backward: data gradients only.  G(relu4_3) = its tap gradient; then convolution by convolution towards the image
    const float* g = w.g_tap[3];
    int flip = 0;
    for (int k = 9; k >= 1; --k) {
      harp_conv3x3_args a = {};
      const int s = S / kVggDiv[k], lv = vgg_level(k);
      a.in = g; a.filters = net->filters_t[k];
      a.N = Nh; a.H = s; a.W = s; a.Cin = kVggCout[k]; a.Cout = kVggCin[k];
      a.precision = net->precision;
      int tap = -1;
      for (int j = 0; j < 3; ++j)
        if (kVggTap[j] == k - 1) tap = j;                 // the layer below is a tap layer followed by the pool: route through it
      if (tap >= 0) {
        a.epilogue = HARP_CONV_UNPOOL; a.out = w.g_tap[tap]; a.gate = w.act[k - 1];
      } else {
        a.epilogue = HARP_CONV_GATE; a.out = w.gbuf[flip]; a.gate = w.act[k - 1];
        flip ^= 1;
      }
      if (bounded) {       // the gradient lives in the level's active tiles and is zero elsewhere
        a.target_row = b.rows;
        a.tile_list = b.list[lv]; a.tile_count = b.count[lv]; a.max_tiles = b.max_tiles[lv];
        a.tile_origin = b.origin[lv]; a.tile_pitch = b.pitch[lv]; a.tile_side = b.side[lv];
        a.in_valid = b.tiles[lv]; a.in_valid_origin = b.origin[lv]; a.in_valid_pitch = b.pitch[lv]; a.in_valid_shift = 1; a.in_valid_cell = b.side[lv];
        if (tap >= 0) { a.out_valid = b.tiles[lv - 1]; a.out_valid_origin = b.origin[lv - 1]; a.out_valid_pitch = b.pitch[lv - 1]; a.out_valid_cell = b.side[lv - 1]; }
      }
      rc = harp_conv3x3(&a, st);
      if (rc != HARP_OK) return rc;
      g = a.out;
    }
    hipLaunchKernelGGL(vgg_grad_image_kernel, dim3((S + kCT - 1) / kCT, (S + kCT - 1) / kCT, Nh), dim3(256), 0, st, g, net->w0t, t->rgb + px * 3, t->y_true,
                       t->mask, rows, t->covered ? t->covered + px : (const int32_t*)nullptr, S, scale[0], t->weight, t->g_rgb + px * 3,
                       finish ? w.loss : (double*)nullptr, t->loss,
                       bounded ? b.tiles[0] : (const int32_t*)nullptr, bounded ? b.origin[0] : (const int32_t*)nullptr, bounded ? b.pitch[0] : 0);
    HARP_CHECK_LAUNCH();
    return HARP_OK;
  };
  // The batch in 2 - 4 parts on as many streams (t->side_streams): the term is a chain of 21 dependent launches of a few rounds of workgroups
  // each, and every launch ends with a partly filled round (measured: 1.17 ms per frame at 32 frames, 1.26 at 16, 1.47 at 8); the chains fill
  // each other's tails.  Needs per-frame target rows (the engine's form); the parts share the loss accumulator.
  int nside = 0;
  while (nside < 3 && t->side_streams[nside]) ++nside;
  if (nside > N - 1) nside = N - 1;
  const bool split = nside > 0 && t->rows != nullptr && t->target_by_row;
  if (!split) return run(0, N, vgg_ws_split(t->ws, N, S, 1), stream, true);
  static thread_local hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};      // (per host thread: one term is enqueued at a time per thread)
  if (!ev_fork) {
    bool ok = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 3; ++i) ok = ok && hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { ev_fork = nullptr; return HARP_ERR_LAUNCH; }
  }
  const int parts = nside + 1;
  if (hipEventRecord(ev_fork, stream) != hipSuccess) return HARP_ERR_LAUNCH;
  for (int i = 0; i < nside; ++i)
    if (hipStreamWaitEvent((hipStream_t)t->side_streams[i], ev_fork, 0) != hipSuccess) return HARP_ERR_LAUNCH;
  int rc = HARP_OK, n0 = 0;
  char* wp = (char*)t->ws;
  double* acc = nullptr;
  for (int i = 0; i < parts; ++i) {
    const int Nh = N / parts + (i < N % parts ? 1 : 0);
    VggWs w = vgg_ws_split(wp, Nh, S, 1);
    wp += w.bytes;
    if (i == 0) acc = w.loss;
    w.loss = acc;
    if (rc == HARP_OK) rc = run(n0, Nh, w, i == 0 ? stream : (hipStream_t)t->side_streams[i - 1], false);
    n0 += Nh;
  }
  // (joined even after an error: a capture must not end with a stream forked off)
  for (int i = 0; i < nside; ++i)
    if (hipEventRecord(ev_join[i], (hipStream_t)t->side_streams[i]) != hipSuccess || hipStreamWaitEvent(stream, ev_join[i], 0) != hipSuccess) return HARP_ERR_LAUNCH;
  if (rc != HARP_OK) return rc;
  hipLaunchKernelGGL(vgg_loss_finish_kernel, dim3(1), dim3(1), 0, stream, acc, t->loss);
  HARP_CHECK_LAUNCH();
  return HARP_OK;
}

}  // extern "C"
