// Shared helpers for the fplx HIP kernels (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/fplx.h"

typedef __bf16 bf16_t;

#define FPLX_WAVE 64

// ---- error reporting (thread-local last message, no global mutable state shared across threads)
inline char* fplx_err_buf() {
  static thread_local char buf[512];
  return buf;
}
inline int fplx_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(fplx_err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
inline int fplx_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fplx_fail(FPLX_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return FPLX_OK;
}
#define FPLX_REQUIRE(cond, code, ...) \
  do {                                 \
    if (!(cond)) return fplx_fail(code, __VA_ARGS__); \
  } while (0)

// ---- dtype traits
template <typename T> struct Act;
template <> struct Act<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Act<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// load/store of a generic element by dtype enum (slow paths only)
__device__ __forceinline__ float ld_dt(const void* p, int64_t i, int dt) {
  return dt == FPLX_F32 ? ((const float*)p)[i] : (float)((const bf16_t*)p)[i];
}
__device__ __forceinline__ void st_dt(void* p, int64_t i, int dt, float v) {
  if (dt == FPLX_F32) ((float*)p)[i] = v;
  else ((bf16_t*)p)[i] = (bf16_t)v;
}

// ---- wave / block reductions (fixed order => deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#ifdef __HIPCC__
// one element of torch.optim.Adam (get_optimizer.py:17) - shared by adam_k (elementwise.hip) and the fused Adam + weight-pack
// kernel (conv_generic.hip) so that both form the same expression tree (same contractions): bit-identical parameters
struct FplxAdamConst { float step_size, b1, b2, eps, wd, inv_sqrt_bc2, gscale, omb1, omb2; };      // omb: 1 - beta
__device__ __forceinline__ void fplx_adam_elem(float& pi, float g, float& mi, float& vi, const FplxAdamConst& c) {
  // every addition is written as an explicit fma and every product that feeds one stands alone: nothing is left for
  // -ffp-contract to decide differently in a scalar and in a float4 context (it did: the two kernels differed in the last bit)
  const float gi = fmaf(c.wd, pi, g * c.gscale);
  mi = fmaf(c.b1, mi, c.omb1 * gi);
  vi = fmaf(c.b2, vi, (c.omb2 * gi) * gi);
  const float denom = fmaf(sqrtf(vi), c.inv_sqrt_bc2, c.eps);
  pi = fmaf(-c.step_size, mi / denom, pi);
}
// XCD-aware block order.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own
// 4-MB L2, so blocks that share data (the cout blocks / tap splits of one voxel footprint, neighbouring footprints and
// their halos) land on eight different L2s and the shared bytes are fetched eight times.  fplx_xcd_block re-labels the
// hardware id L = (z * gy + y) * gx + x: XCD L % 8 takes the L / 8-th block of ITS contiguous eighth of the logical
// order, and the logical order runs y fastest, then z, then x - all blocks of one footprint are consecutive, resident
// at the same time and on the same L2.  on = 0 returns the hardware ids (A/B knob FPLX_XCD=0).
struct FplxBlock { int x, y, z; };
__device__ __forceinline__ FplxBlock fplx_xcd_block(int on) {
  FplxBlock b = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (!on) return b;
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, t = gx * gy * gz;
  const unsigned l = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
  const unsigned xcd = l & 7u, q = t >> 3, r = t & 7u;       // XCD j owns q + (j < r) blocks
  const unsigned lp = xcd * q + (xcd < r ? xcd : r) + (l >> 3);
  const unsigned per = gy * gz, xq = lp / per, rem = lp - xq * per;
  b.x = __builtin_amdgcn_readfirstlane((int)xq);
  b.z = __builtin_amdgcn_readfirstlane((int)(rem / gy));
  b.y = __builtin_amdgcn_readfirstlane((int)(rem % gy));
  return b;
}
// The same for persistent kernels that stride a block over a list of tiles (grid.x blocks, tile += grid.x): XCD j sweeps
// its contiguous eighth of the tile list with its own blocks, so neighbouring tiles (shared halos) run on one L2.
struct FplxTileRange { int64_t first, end, step; };
__device__ __forceinline__ FplxTileRange fplx_xcd_tiles(int64_t ntiles, int on) {
  FplxTileRange t = {(int64_t)blockIdx.x, ntiles, (int64_t)gridDim.x};
  if (!on || gridDim.x < 16) return t;
  const unsigned c = blockIdx.x & 7u, xcd = (blockIdx.y * gridDim.x + blockIdx.x) & 7u;
  const int64_t q = ntiles >> 3, r = ntiles & 7;
  const int64_t start = xcd * q + ((int64_t)xcd < r ? xcd : r);
  t.first = start + (blockIdx.x >> 3);
  t.end = start + q + ((int64_t)xcd < r ? 1 : 0);
  t.step = (gridDim.x - c + 7u) >> 3;
  return t;
}
#endif
// ---- tuning table (fplx_set_tuning / fplx_get_tuning, include/fplx.h): ONE process-wide table of named integer knobs
// for A/B measurements - every default is the shipped configuration and no knob changes a result, only which kernel /
// geometry computes it.  It replaces the getenv() statics the dispatchers used to cache; it is the library's only
// mutable state (relaxed atomics: a knob may be flipped between launches from any thread).
//   X(id, key, default)
#define FPLX_KNOB_LIST(X)                                                                                              \
  X(XCD, "xcd", 1)                         /* 0: hardware block order (fplx_xcd_block off) */                           \
  X(BRICK, "brick", 1)                     /* 0: no brick kernel; 3: only layers no march kernel takes */               \
  X(BRICK_FILL, "brick_fill", 192)         /* the Cin split of the brick kernel grows until the launch has this many blocks */  \
  X(BRICK_GEO, "brick_geo", -1)            /* >= 0: force this brick geometry on every eligible layer (tests) */        \
  X(BRICK_KSPLIT, "brick_ksplit", 0)       /* > 0: force this Cin split (tests) */                                      \
  X(EDGE_BLOCKS, "edge_blocks", 1024)      /* persistent blocks of the stem / out_conv kernels */                       \
  X(STEM_ROWS, "stem_rows", 1)             /* 0: the tile kernel for in_chns = 1 too */                                 \
  X(OUTCONV_T, "outconv_t", 1)             /* 0: the 32 x 32 out_conv forward (classes as columns) */                    \
  X(OUTCONV_DGRAD_MFMA, "outconv_dgrad_mfma", 1)                                                                       \
  X(OUTCONV_FWD_ROWS, "outconv_fwd_rows", 1) /* fused out_conv forward as a march of row segments (outconv_fwd_rows); 0: the tile kernel outconv_fwd_t<1, true> (A/B, tested both ways) */ \
  X(OUTCONV_DGRAD_ROWS, "outconv_dgrad_rows", 1) /* fused out_conv backward as a stream of row segments (outconv_dgrad_rows); 0: the tile kernel (A/B, tested both ways) */ \
  X(PACK_TILED, "pack_tiled", 1)                                                                                       \
  X(PACK_MULTI, "pack_multi", 1)                                                                                       \
  X(MARCH, "march", 1)                     /* 0: previous-generation stream kernels; 2: Cin = 32 march only */          \
  X(MARCH64_FW, "march64_fw", 0)           /* 16 / 32: force the footprint of the Cin = 64 march */                     \
  X(MARCH_DS, "march_ds", 0)               /* > 0: force the depth split of the march kernels */                        \
  X(MARCH128, "march128", 1)                                                                                           \
  X(MARCH64_LW, "march64_lw", 1)           /* loader-wave form of conv_fwd_march64 for the 2.5D (Conv2d) forms; 0: conv_fwd_march64 (A/B, tested both ways) */ \
  X(MARCH32_V2, "march32_v2", 4)           /* 0: 8-wave kernel; 1: v2 everywhere; 4: v2 forward + v3 data gradient */   \
  X(WG_COT_MINVOX, "wg_cot_minvox", 0)                                                                                 \
  X(WG_TW, "wg_tw", 0)                                                                                                 \
  X(WG_CIT, "wg_cit", 2)                                                                                               \
  X(WG_COT, "wg_cot", 2)                                                                                               \
  X(WG_DS, "wg_ds", 0)                                                                                                 \
  X(WG_VOX, "wg_vox", 1)                   /* 0: footprint march everywhere; 2: voxel GEMM wherever it can run */           \
  X(WG_ROLL, "wg_roll", 1)                 /* 0: conv_wgrad_stream everywhere (the rolling-window weight gradient off) */    \
  X(WG_ROLL_GEO, "wg_roll_geo", 0)         /* footprint of conv_wgrad_roll: 0 least padding, 1: 8 x 32, 2: 16 x 16, 3: 8 x 16 */ \
  X(WG_ROLL_M16, "wg_roll_m16", 1)         /* 0: the 8 x 32 footprint on v_mfma_f32_32x32x16_bf16 instead of 16x16x32 */        \
  X(WG_ROLL_MB, "wg_roll_mb", 1)           /* 0: barrier at the end of a depth step (4 + 2 ring slots); 2: mid-step form for 8 x 16 too */ \
  X(WG_ROLL_CUS, "wg_roll_cus", 256)       /* CUs the rolling-window kernel's depth split plans for */                         \
  X(WG_ROLL_OVH, "wg_roll_ovh", 9)         /* per-block overhead of its depth-split cost model, in depth steps */            \
  X(WG_ROLL_MINVOX, "wg_roll_minvox", 0)   /* smallest d x h x w it takes */                                                \
  X(WG_ROLL2D, "wg_roll2d", 1)             /* 0: the 2.5D levels' weight gradients on conv_wgrad_stream<.., TWOD> instead of conv_wgrad_roll2d */ \
  X(WG_ROLL2D_OVH, "wg_roll2d_ovh", 3)     /* per-block overhead of conv_wgrad_roll2d in depth steps (depth-split cost model) */ \
  X(WG_REDUCE_ROWS, "wg_reduce_rows", 8)   /* largest number of partial blocks the row-wise finish of the weight gradients takes (0: the lane kernel everywhere) */ \
  X(WG_VOX_LW, "wg_vox_lw", 1)             /* loader-wave voxel-GEMM weight gradient: 1 volumes <= 2048 voxels, 2 everywhere, 0 never */ \
  X(WG_VOX_CUS, "wg_vox_cus", 128)         /* blocks the voxel-GEMM weight gradient aims at: HALF the CUs - the main stream's deep-level kernels are small latency-bound grids that need free CUs at once (step -0.9 % against 256) */ \
  X(WG_VOX_MAXV, "wg_vox_maxv", 10000)     /* largest voxel count (whole batch) the voxel-GEMM weight gradient takes */    \
  X(STREAM_MIN_W, "stream_min_w", 64)                                                                                  \
  X(TILE_MT, "tile_mt", 0)                                                                                             \
  X(TILE_NT, "tile_nt", 0)                                                                                             \
  X(TILE_KS, "tile_ks", 0)                                                                                             \
  X(MID_TILE, "mid_tile", 1)                                                                                           \
  X(DECONV_ROWS, "deconv_rows", 1)                                                                                     \
  X(DECONV_DGRAD_ROWS, "deconv_dgrad_rows", 1)  /* 0: conv_fwd_direct for the shallow transposed-convolution data gradients */ \
  X(ROWS_SMALL_DIV, "rows_small_div", 16)  /* voxels per partial row of the streaming reductions for volumes <= 32768 voxels (64: as for the large ones) */ \
  X(EW_GROUP, "ew_group", 1)                                                                                           \
  X(POOL_COL, "pool_col", 1)               /* 0: the fused DownBlock-tail passes with a thread per pooled voxel */          \
  X(EW_INFLIGHT, "ew_inflight", 2)         /* voxels (pairs of 16-byte loads) in flight per lane of bn_act_bwd_apply: 4, 2, 1 (131 / 99 / 72 registers) */ \
  X(EW_INFLIGHT_REDUCE, "ew_inflight_reduce", 2) /* the same for bn_act_bwd_reduce: 4, 2, 1 */
enum FplxKnobId {
#define FPLX_KNOB_ENUM(id, key, def) FPLX_K_##id,
  FPLX_KNOB_LIST(FPLX_KNOB_ENUM)
#undef FPLX_KNOB_ENUM
  FPLX_K_COUNT
};
extern "C" __attribute__((visibility("hidden"))) int64_t fplx_knob_values[FPLX_K_COUNT];     // conv_generic.hip
static inline int64_t fplx_knob(int id) { return __atomic_load_n(&fplx_knob_values[id], __ATOMIC_RELAXED); }
static inline int fplx_xcd_on() { return (int)fplx_knob(FPLX_K_XCD); }

// number of partial rows (= blocks) used by the streaming reductions: a fixed function of the voxel count
// so that producer and consumer agree without extra plumbing.  One row per 64 voxels, at most 512: the deep levels
// (8000 / 1000 voxels x 256 / 512 channels) still spread over the chip - with one row per 2048 voxels they ran on 1-32
// CUs and cost as much as level 0 - and the finalize kernels read at most 512 rows (measured: +4% on the train step).
// Round 5: one row per 16 voxels for volumes of at most 32768 voxels - with 64 the reduction of a 1000-voxel site ran on 16
// blocks, 16 voxels one after the other per lane (14 us; 63 blocks: see profiles/r05_kernel_ab.txt section 14).
// (Part of the ABI's contract - callers size their buffers from fplx_num_partials / fplx_*_stats_rows.  The knob "rows_small_div"
// exists for A/B only: it must not change between a producer and its consumer, i.e. inside a step.)
static inline int fplx_rows_for(int64_t voxels) {
  constexpr int cap = 512;
  const int div = voxels <= 32768 ? (int)fplx_knob(FPLX_K_ROWS_SMALL_DIV) : 64;
  int64_t r = (voxels + div - 1) / div;
  if (r > cap) r = cap;
  if (r < 1) r = 1;
  return (int)r;
}

