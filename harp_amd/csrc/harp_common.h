// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the HARP render-and-compare path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HARP_OK 0
#define HARP_ERR_ARG 1
#define HARP_ERR_LAUNCH 2

#define HARP_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return HARP_ERR_LAUNCH + (int)e__; \
  } while (0)

// Dynamic-LDS "fence" size of a kernel that asks for more LDS than it uses in order to keep other workgroups off its CU (hand_front,
// texture_terms): decided ONCE per process and kernel — `want` (or the environment override `env`, bytes; 0 = none) clamped to what the device
// and the kernel's static LDS leave, never below `need`; if the attribute cannot be raised the launch just uses `need` (the fence is an
// optimisation, not an argument error).  Call with the first launch's `need`; later launches with a larger `need` get max(need, cached).
#include <stdlib.h>
inline size_t harp_lds_fence(const void* kernel, const char* env, size_t want, size_t need, size_t* cache) {
  if (*cache == 0) {
    size_t ask = want;
    if (const char* e = getenv(env)) ask = (size_t)atoi(e);
    int dev = 0, max_lds = 64 * 1024;
    hipFuncAttributes fa;
    size_t stat = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) stat = fa.sharedSizeBytes;
    const size_t room = (size_t)max_lds > stat ? (size_t)max_lds - stat : 0;
    if (ask > room) ask = room;
    if (ask > 48 * 1024 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ask) != hipSuccess) ask = 0;
    (void)hipGetLastError();
    *cache = ask + 1;                 // (+1: 0 means "not decided yet")
  }
  const size_t fence = *cache - 1;
  return fence > need ? fence : need;
}

constexpr float kEps = 1e-8f;      // PyTorch3D kEpsilon
constexpr int kWave = 64;          // gfx950 wavefront

// Screen tiling shared by the binning and raster kernels: 64x64-px super-tiles hold the coarse face
// lists; a workgroup of 4 waves owns a 16x16 tile, each wave a 16x4 strip (64-B output rows).
constexpr int kSuper = 64;
constexpr int kTile = 16;

// Per-face setup record (64 B, one float4-aligned struct per (frame, face)).
struct FaceRec {
  float4 a;   // x0 y0 z0 x1
  float4 b;   // y1 z1 x2 y2
  float4 c;   // z2 face_area - -
  float4 bb;  // xmin-r xmax+r ymin-r ymax+r   (empty box => culled face)
};

// Layout of a rasteriser workspace (harp_rasterize_ws_bytes): face records | contiguous bboxes | per-super-tile face lists |
// list lengths | heaviest-first launch order of the (frame, super-tile) pairs.
// ... | hit bitmaps: one bit per (frame, super-tile, face), 64 faces per word (written whole by the face set-up pass, expanded into the lists)
struct RasterWs { FaceRec* recs; float4* bbs; int32_t* bins; int32_t* cnt; int32_t* order; int32_t* nact; int nsx; unsigned long long* bits; int W64; };
__host__ __device__ inline RasterWs raster_ws_split(void* ws, int B, int F, int S) {
  RasterWs r;
  r.nsx = (S + kSuper - 1) / kSuper;
  char* p = (char*)ws;
  r.recs = (FaceRec*)p;   p += (size_t)B * F * sizeof(FaceRec);
  r.bbs = (float4*)p;     p += (size_t)B * F * sizeof(float4);
  r.bins = (int32_t*)p;   p += (size_t)B * r.nsx * r.nsx * F * sizeof(int32_t);
  r.cnt = (int32_t*)p;    p += (((size_t)B * r.nsx * r.nsx * sizeof(int32_t)) + 255) / 256 * 256;
  r.order = (int32_t*)p;  p += (((size_t)B * r.nsx * r.nsx * sizeof(int32_t)) + 255) / 256 * 256;
  r.nact = (int32_t*)p;   p += 256;   // one int: number of super-tiles that hold faces = number of leading launch-order slots with work
  r.bits = (unsigned long long*)p;
  r.W64 = (F + 63) / 64;
  return r;
}
// workgroups of the 1-D tile grid: launch-order slots rounded up to a multiple of 8 (one per XCD) x 16 tiles per super-tile
__host__ __device__ inline unsigned tile_grid(int B, int nsx) { return (unsigned)(((B * nsx * nsx + 7) / 8) * 8 * (kSuper / kTile) * (kSuper / kTile)); }
// Decode a workgroup id of that grid: consecutive ids go round-robin over the 8 XCDs, so the 16 tiles of one super-tile (same bin
// list, same face records, neighbouring pixels) stay on one XCD / one L2; slots follow the heaviest-first order.
// Returns 0: no tile; 1: tile of a super-tile that holds faces; 2: tile of an EMPTY super-tile.  The slots are ordered heaviest-first,
// so "empty" is slot >= *nact: one scalar load, no dependent trip through the bin counts.  3/4 of the workgroups of a launch are such
// tiles and every one of them used to walk order -> bin count / face ids -> exit: with 5 workgroups per CU in flight that chain, not
// the shading, was a third of the forward shader's time.  coords_if_empty = false: return 2 without touching `order`.
// (tile_decode_v: the same for an explicit index `bid` of that grid — a launch that interleaves two kinds of tiles passes a virtual index
//  whose low 3 bits still equal the real workgroup id's, i.e. the XCD)
__device__ __forceinline__ int tile_decode_v(unsigned bid, const int32_t* __restrict__ order, const int32_t* __restrict__ nact, int B, int nsx, int S,
                                             int& b, int& st, int& tx0, int& ty0, int& sub, bool coords_if_empty = true) {
  constexpr int kTps = (kSuper / kTile) * (kSuper / kTile);
  const int nst = nsx * nsx;
  const int xcd = bid & 7, rr = bid >> 3;
  const int slot = (rr / kTps) * 8 + xcd;
  sub = rr % kTps;
  if (slot >= B * nst) return 0;
  const bool empty = slot >= nact[0];
  if (empty && !coords_if_empty) return 2;
  const int entry = order[slot];
  // (integer division runs on the vector ALU: its wave-uniform results are moved back to scalar registers explicitly, so that every
  //  address formed from the frame / tile stays a scalar base + per-lane 32-bit offset instead of a 64-bit address per lane)
  b = __builtin_amdgcn_readfirstlane(entry / nst); st = entry - b * nst;
  tx0 = __builtin_amdgcn_readfirstlane(((st % nsx) * (kSuper / kTile) + (sub & 3)) * kTile);
  ty0 = __builtin_amdgcn_readfirstlane(((st / nsx) * (kSuper / kTile) + (sub >> 2)) * kTile);
  if (!(tx0 < S && ty0 < S)) return 0;
  return empty ? 2 : 1;
}
__device__ __forceinline__ int tile_decode(const int32_t* __restrict__ order, const int32_t* __restrict__ nact, int B, int nsx, int S,
                                           int& b, int& st, int& tx0, int& ty0, int& sub, bool coords_if_empty = true) {
  return tile_decode_v(blockIdx.x, order, nact, B, nsx, S, b, st, tx0, ty0, sub, coords_if_empty);
}
// Pixels of a whole 64x64 super-tile spread over a 256-thread workgroup (16 each, 64-wide coalesced rows): k = 0..15
__device__ __forceinline__ void supertile_pixel(int k, int sx0, int sy0, int& xi, int& yi) {
  const int idx = k * 256 + (int)threadIdx.x;
  xi = sx0 + (idx & (kSuper - 1));
  yi = sy0 + (idx >> 6);
}

__device__ __forceinline__ float pix_to_ndc(int i, int S) {
  // pixel index -> NDC of its centre; PyTorch3D flips both axes: pixel 0 is at +1-1/S.
  return -1.0f + (2.0f * (float)(S - 1 - i) + 1.0f) / (float)S;
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0.
__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Workgroup-local gradient accumulator keyed by vertex id: an open-addressing hash table in LDS (NV floats per key).
// Pixels of one screen tile touch only a few dozen distinct vertices, so gradients are first summed with LDS float
// atomics (ds_add_f32) and each (vertex, component) is then flushed with ONE global atomic per workgroup instead of one
// per pixel.  A full table (probe limit hit) falls back to direct global atomics, so it is always correct.
// T = double by default: on gfx950 ds_add_f32 retires ~1 lane per 3 clocks (193 clk per wave64 instruction, measured with
// tools/dev/micro/lds_atomics*.hip, independent of the address pattern) while ds_add_f64 takes 8.7 clk (44 clk with 4 lanes per
// address) and ds_add_u32 4.8 clk — fp32 LDS atomics were the dominant cost of every backward kernel that used them.
template <int SLOTS, int NV, typename T = double>
struct VertexAccum {
  int key[SLOTS];
  T val[SLOTS][NV];

  __device__ __forceinline__ void clear() {
    for (int i = threadIdx.x; i < SLOTS; i += blockDim.x) {
      key[i] = -1;
#pragma unroll
      for (int c = 0; c < NV; ++c) val[i][c] = (T)0;
    }
  }
  // returns slot or -1 (table full)
  __device__ __forceinline__ int find(int v) {
    unsigned h = ((unsigned)v * 2654435761u) & (SLOTS - 1);
    for (int probe = 0; probe < 32; ++probe) {
      const int k = atomicCAS(&key[h], -1, v);
      if (k == -1 || k == v) return (int)h;
      h = (h + 1) & (SLOTS - 1);
    }
    return -1;
  }
  // slot of an existing key or -1 (never inserts)
  __device__ __forceinline__ int lookup(int v) const {
    unsigned h = ((unsigned)v * 2654435761u) & (SLOTS - 1);
    for (int probe = 0; probe < 32; ++probe) {
      const int k = key[h];
      if (k == v) return (int)h;
      if (k == -1) return -1;
      h = (h + 1) & (SLOTS - 1);
    }
    return -1;
  }
  __device__ __forceinline__ void add(int slot, int c, float x) { atomicAdd(&val[slot][c], (T)x); }
};

// value of lane (lane ^ bit): DPP only for 1, 2 (quad permutes), 4 (half-row mirror, then quad reversal: i -> 7-i -> (7-i)^3 = i^4)
// and 8 (row rotate by 8); ds_bpermute (an LDS round trip) otherwise
__device__ __forceinline__ int lane_xor(int x, int bit) {
  if (bit == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
  if (bit == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
  if (bit == 4) return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  if (bit == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);
  return __shfl_xor(x, bit);
}

// Wave-wide reductions on the VALU only (no LDS round trips): DPP inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8),
// then row_bcast15 / row_bcast31 carry the row totals forward; lane 63 holds the result, returned as a wave-uniform value.
// (wave_sum above goes through ds_bpermute for the distances DPP cannot express: ~100 cycles per step instead of ~8.)
#define HARP_DPP_STEP(OP, T_AS_INT, T_FROM_INT, ctrl, rmask, ident_is_self)                                              \
  {                                                                                                                     \
    const int self_ = T_AS_INT(v);                                                                                      \
    const int oth_ = __builtin_amdgcn_update_dpp((ident_is_self) ? self_ : 0, self_, ctrl, rmask, 0xF, false);         \
    v = OP(v, T_FROM_INT(oth_));                                                                                        \
  }
__device__ __forceinline__ float dpp_fadd_(float a, float b) { return a + b; }
__device__ __forceinline__ int dpp_imin_(int a, int b) { return min(a, b); }
__device__ __forceinline__ int dpp_ident_(int a) { return a; }
__device__ __forceinline__ float wave_sum_u(float v) {
  HARP_DPP_STEP(dpp_fadd_, __float_as_int, __int_as_float, 0x111, 0xF, false)
  HARP_DPP_STEP(dpp_fadd_, __float_as_int, __int_as_float, 0x112, 0xF, false)
  HARP_DPP_STEP(dpp_fadd_, __float_as_int, __int_as_float, 0x114, 0xF, false)
  HARP_DPP_STEP(dpp_fadd_, __float_as_int, __int_as_float, 0x118, 0xF, false)
  HARP_DPP_STEP(dpp_fadd_, __float_as_int, __int_as_float, 0x142, 0xA, false)
  HARP_DPP_STEP(dpp_fadd_, __float_as_int, __int_as_float, 0x143, 0xC, false)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// N sums at once, step-major: the N dependent chains of six DPP steps interleave (one after the other, every step waits for the previous
// one's result: ~150 cycles per sum instead of ~35).  The sums end up in LANE 63 of v[0..N) (no readlane).
template <int N>
__device__ __forceinline__ void wave_sum_u_n(float (&v)[N]) {
#define HARP_DPP_STEP_N(ctrl, rmask)                                                                                         \
  _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_) {                                                                         \
    const int self_ = __float_as_int(v[k_]);                                                                                 \
    v[k_] += __int_as_float(__builtin_amdgcn_update_dpp(0, self_, ctrl, rmask, 0xF, false));                                 \
  }
  HARP_DPP_STEP_N(0x111, 0xF)
  HARP_DPP_STEP_N(0x112, 0xF)
  HARP_DPP_STEP_N(0x114, 0xF)
  HARP_DPP_STEP_N(0x118, 0xF)
  HARP_DPP_STEP_N(0x142, 0xA)
  HARP_DPP_STEP_N(0x143, 0xC)
#undef HARP_DPP_STEP_N
}
__device__ __forceinline__ float wave_max_u(float v) {
  HARP_DPP_STEP(fmaxf, __float_as_int, __int_as_float, 0x111, 0xF, true)
  HARP_DPP_STEP(fmaxf, __float_as_int, __int_as_float, 0x112, 0xF, true)
  HARP_DPP_STEP(fmaxf, __float_as_int, __int_as_float, 0x114, 0xF, true)
  HARP_DPP_STEP(fmaxf, __float_as_int, __int_as_float, 0x118, 0xF, true)
  HARP_DPP_STEP(fmaxf, __float_as_int, __int_as_float, 0x142, 0xA, true)
  HARP_DPP_STEP(fmaxf, __float_as_int, __int_as_float, 0x143, 0xC, true)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_u(int v) {
  HARP_DPP_STEP(dpp_imin_, dpp_ident_, dpp_ident_, 0x111, 0xF, true)
  HARP_DPP_STEP(dpp_imin_, dpp_ident_, dpp_ident_, 0x112, 0xF, true)
  HARP_DPP_STEP(dpp_imin_, dpp_ident_, dpp_ident_, 0x114, 0xF, true)
  HARP_DPP_STEP(dpp_imin_, dpp_ident_, dpp_ident_, 0x118, 0xF, true)
  HARP_DPP_STEP(dpp_imin_, dpp_ident_, dpp_ident_, 0x142, 0xA, true)
  HARP_DPP_STEP(dpp_imin_, dpp_ident_, dpp_ident_, 0x143, 0xC, true)
  return __builtin_amdgcn_readlane(v, 63);
}

// Lanes of a 16x4-pixel wave that hit the same face add to the same vertices, and same-address LDS atomics serialise: merge such
// lanes first with a butterfly over the xor distances in MASK (1, 2 = x neighbours, 16 = the row below).  After the call only lanes
// with alive == true hold (summed) values; `fk` = face id (or a negative value for inactive lanes).
template <int N, int MASK>
__device__ __forceinline__ void merge_same_face(float* v, int fk, bool& alive, int lane) {
#pragma unroll
  for (int bit = 1; bit < 64; bit <<= 1) {
    if (!(MASK & bit)) continue;
    const int fo = lane_xor(fk, bit);
    const int ao = lane_xor(alive ? 1 : 0, bit);
    const bool same = alive && ao && fo == fk;
    const bool lower = !(lane & bit);
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const float o2 = __int_as_float(lane_xor(__float_as_int(v[c]), bit));
      if (same && lower) v[c] += o2;
    }
    if (same && !lower) alive = false;
  }
}

// chain rule of F.normalize(x, dim=-1) (eps 1e-12) for texel i: gx[i] += d normalize / dx ^T gy[i]   (utils/visualize.py:99)
__device__ __forceinline__ void normalize3_bwd_vals(const float* __restrict__ x, float g0, float g1, float g2, float* __restrict__ gx, size_t i) {
  const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
  const float l = sqrtf(a * a + b * b + c * c);
  if (l > 1e-12f) {
    const float inv = 1.0f / l, na = a * inv, nb = b * inv, nc = c * inv, d = na * g0 + nb * g1 + nc * g2;
    gx[3 * i] += (g0 - na * d) * inv; gx[3 * i + 1] += (g1 - nb * d) * inv; gx[3 * i + 2] += (g2 - nc * d) * inv;
  } else {
    gx[3 * i] += g0 * 1e12f; gx[3 * i + 1] += g1 * 1e12f; gx[3 * i + 2] += g2 * 1e12f;
  }
}

__device__ __forceinline__ void normalize3_bwd_texel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, size_t i) {
  normalize3_bwd_vals(x, gy[3 * i], gy[3 * i + 1], gy[3 * i + 2], gx, i);
}

// F.normalize(x, dim=-1) of one texel (eps 1e-12; utils/visualize.py:99) with the sum of squares spelled as fused multiply-adds, so that
// every translation unit that normalises the normal map (losses.hip, shade.hip) produces the same bits
__device__ __forceinline__ void normalize3_texel(float a, float b, float c, float& x, float& y, float& z) {
  const float inv = 1.0f / fmaxf(sqrtf(__fmaf_rn(a, a, __fmaf_rn(b, b, c * c))), 1e-12f);
  x = a * inv; y = b * inv; z = c * inv;
}
