// Depth-marching 3x3x3 convolutions for gfx950 (bf16 NDHWC, fp32 accumulate, v_mfma_f32_32x32x16_bf16).
//
// conv_fwd_march32 (Cin = 32, Cout % 32 == 0, H >= 16, W >= 64: level 0 of the 32-base network)
// One block (8 waves, two per SIMD) owns a 16 x 32 (h, w) output footprint x 32 output channels and marches along
// d.  It is INPUT-stationary in depth: each input slab (18 x 34 voxels x 32 channels, 1-voxel halo, zero fill =
// padding) is staged ONCE - by LDS-DMA - into a two-slot LDS ring and scattered into the accumulators of the three
// output depths it touches (d+1 through kd = 0, d through kd = 1, d-1 through kd = 2), so
//   * every voxel fragment read from LDS feeds three MFMAs (three kd taps), every weight fragment two (the wave's two
//     M-tiles): 0.83 ds_read_b128 per MFMA instead of 1.5 in the slab-ring kernel it replaces;
//   * only the current slab and the one in flight live in LDS (2 x 39 KB) next to the block's 27 x 32 x 32 weights
//     (54 KB, resident), which leaves room for the 16-row footprint (halo overhead 1.20 instead of 1.33);
//   * four accumulator sets per wave: the depths d+1, d, d-1 of the current slab and the depth that completed in the
//     previous step, which is written out WHILE the next slab is multiplied: bias, bf16, a transpose through a 2-KiB
//     per-wave LDS tile, 16-byte global stores of whole 128-byte lines (2-byte stores made the CU's store path, not
//     the matrix core, set the pace).  The roles shift by register moves at the end of a step.
// DSBN statistics: per-lane sums (lane = output channel), reduced in a fixed order, one partial row per block.
// conv_fwd_march64 (Cin = 64): see its own header further down.
//
// Replaces (for these shapes) nn.Conv3d forward and, with the mirrored pack, its data gradient -
// reference PyMIC/pymic/net/net3d/unet2d5_dsbn.py:54-55,75,79 (ConvolutionLayer / ConvBlockND), and the
// torch.cat of unet2d5_dsbn.py:182 when the input / output is given as two tensors (x1 / y1).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int fplx_zero16[4] = {0u, 0u, 0u, 0u};
#ifdef FPLX_STAMP
__device__ long long* fplx_stamp_buf;      // set by the micro-benchmark
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Barrier for LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for the acknowledgement of every global store
// the wave has in flight - at the end of a block that is the whole write-out of its last depth (2-3 us per block, 12-24 us
// per launch measured on the statistics forms); the statistics tail only exchanges LDS data.  Callers guarantee that no
// LDS-DMA is in flight (the march loops end with vmcnt(0) + barrier).
#define FPLX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct MG {
  static constexpr int CIN = 32, ROWB = 64, CH = 4;
  static constexpr int FH = 16, FW = 32, SH = FH + 2, SW = FW + 2, SLAB = SH * SW;
  static constexpr int THREADS = 512;
  static constexpr int SLAB_CHUNKS = SLAB * CH, SLAB_DMA = (SLAB_CHUNKS + 63) / 64;   // 1-KiB LDS-DMA pieces
  static constexpr int SLAB_BYTES = SLAB_DMA * 1024, W_BYTES = 27 * 32 * ROWB;
  static constexpr int NPIECE = (SLAB_DMA + 7) / 8;        // DMA wave-instructions per wave and slab
  static constexpr int STAGE_BYTES = 32 * 32 * 2;          // one M-tile of bf16 outputs per wave (store transpose)
  static constexpr int LDS = 2 * SLAB_BYTES + W_BYTES + 32 * 4 + NPIECE * THREADS * 4 + (THREADS / 64) * STAGE_BYTES;
  static __device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
};

// one input slab -> the three output depths it touches.  A0/A1/A2 = accumulators of depth s+1 / s / s-1.
// side(q) runs behind the MFMAs of stage q: the slab DMA for the next step and the write-out of the depth that
// completed in the previous step are spread over the stages, so they overlap with matrix work of the SAME wave
// (the block's waves move in lock-step from barrier to barrier: separate phases would not overlap at all).
template <int MASK, class Side>
__device__ __forceinline__ void march_step(const char* __restrict__ sl, const char* __restrict__ wbuf, int wave, int r,
                                           int khalf, f32x16& A00, f32x16& A01, f32x16& A10, f32x16& A11, f32x16& A20,
                                           f32x16& A21, Side&& side) {
  bf16x8 fa[2][2], fb[2][3];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // lane constants are re-derived from opaque copies every slab, so that the 36 + 54 fragment addresses are
  // computed next to their use (a handful of VALU ops under the MFMAs) instead of being hoisted out of the depth
  // loop and spilled
  int vb = wave * 2 * MG::SW + r, rb = r;
  asm volatile("" : "+v"(vb), "+v"(rb));
  // weight rows: swz(tap * 32 + r) == swz(r), so a tap is an immediate offset from two lane bases
  const char* wl0 = wbuf + rb * MG::ROWB + ((khalf ^ MG::swz(rb)) << 4);
  const char* wl1 = wbuf + rb * MG::ROWB + (((2 + khalf) ^ MG::swz(rb)) << 4);
  auto load_stage = [&](int q, int buf) {
    const int p = q >> 1, ks = q & 1;
    const int kh = p / 3, kw = p % 3;
    const int c = 2 * ks + khalf;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int vox = vb + (m + kh) * MG::SW + kw;
      fa[buf][m] = *reinterpret_cast<const bf16x8*>(sl + vox * MG::ROWB + ((c ^ MG::swz(vox)) << 4));
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
      if ((MASK >> kd) & 1) fb[buf][kd] = *reinterpret_cast<const bf16x8*>((ks ? wl1 : wl0) + (kd * 9 + p) * 32 * MG::ROWB);
  };
  load_stage(0, 0);
#pragma unroll
  for (int q = 0; q < 18; ++q) {
    if (q + 1 < 18) load_stage(q + 1, (q + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const int b = q & 1;
    {
      if constexpr ((MASK & 1) != 0) A00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][0], q == 0 ? zero : A00, 0, 0, 0);
      if constexpr ((MASK & 2) != 0) A10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][1], A10, 0, 0, 0);
      if constexpr ((MASK & 4) != 0) A20 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][2], A20, 0, 0, 0);
      if constexpr ((MASK & 1) != 0) A01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][0], q == 0 ? zero : A01, 0, 0, 0);
      if constexpr ((MASK & 2) != 0) A11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][1], A11, 0, 0, 0);
      if constexpr ((MASK & 4) != 0) A21 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][2], A21, 0, 0, 0);
    }
    side(q);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// TWOD: the pack is a Conv2d in the middle depth plane (2.5D levels, fplx_pack_conv2d_weight): a slab feeds only its own
// output depth, through the kd = 1 taps - step mask 2, a third of the MFMAs, no depth halo; the accumulator roles and
// the write-out pipeline are the same (K0 stays zero and keeps re-initialising K1 through the role shift).
// ACT (inference, eval-mode BatchNorm folded into the pack): the write-out applies PReLU(slope) and keeps no statistics
template <bool TWOD, bool ACT = false>
__global__ void __launch_bounds__(MG::THREADS)
conv_fwd_march32(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                 const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                 int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen,
                 bf16_t* __restrict__ y1, int ysplit, int xcd, const float* __restrict__ slope_p = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FPLX_STAMP
  const long long st_launch = __builtin_amdgcn_s_memtime();
#endif
  char* slabs = smem;
  char* wbuf = smem + 2 * MG::SLAB_BYTES;
  float* bias_s = reinterpret_cast<float*>(wbuf + MG::W_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // block-uniform values live in SGPRs
  const int r = lane & 31, khalf = lane >> 5;
  const FplxBlock bid = fplx_xcd_block(xcd);
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int h0 = __builtin_amdgcn_readfirstlane(th * MG::FH), w0 = __builtin_amdgcn_readfirstlane(tw * MG::FW);
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = bid.y * 32;

  // global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  A wave-instruction
  // fills 1 KiB = 16 voxel rows linearly, so the XOR swizzle is applied to each lane's SOURCE chunk; halo voxels
  // outside the volume read a 16-byte zero constant.
  // Issued as inline asm: with the builtin in the loop hipcc stops counting lgkmcnt and drains every ds_read with
  // lgkmcnt(0), which defeats the fragment prefetch.  The DMA is retired by the explicit vmcnt(0) in front of the
  // barrier that publishes the slab.
  auto lds_dma = [&](const void* g, char* l) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)l));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  // vmcnt(0) retires this wave's DMA pieces (and its output stores: loads and stores share the counter and may
  // complete out of order, so only 0 is a safe count); the raw barrier then publishes the slab to the block
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int NPIECE = MG::NPIECE;
  // per lane and piece: byte offset of the source chunk inside one depth slice of x, or -1 for the zero constant
  // (halo outside the volume, and the tail of the last 1-KiB piece).  Parked in LDS: as registers they would be
  // spilled, and a scratch reload in the loop makes hipcc wait for vmcnt(0), i.e. for the DMA just issued.
  int* soff_s = reinterpret_cast<int*>(bias_s + 32);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int i = (wave + 8 * k) * 64 + lane;
    const int vox = i >> 2, c = (i & 3) ^ MG::swz(vox);
    const int hh = vox / MG::SW + h0 - 1, ww = vox % MG::SW + w0 - 1;
    const bool in = i < MG::SLAB_CHUNKS && hh >= 0 && hh < H && ww >= 0 && ww < W;
    soff_s[k * MG::THREADS + tid] = in ? (int)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : -1;
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2;        // bytes per depth slice
  const char* xn = reinterpret_cast<const char*>(x) + (int64_t)n * D * xslice;
  auto slab_piece = [&](int s, int k, int so) {           // piece k of slab s -> slot (s + 1) & 1
    if (wave + 8 * k < MG::SLAB_DMA) {
      const char* xs = xn + s * xslice;                   // uniform
      const void* src = so >= 0 ? (const void*)(xs + (unsigned)so) : (const void*)fplx_zero16;
      lds_dma(src, slabs + ((s + 1) & 1) * MG::SLAB_BYTES + (wave + 8 * k) * 1024);
    }
  };

  // accumulators by role (2 M-tiles each): K0 / K1 / K2 = output depth s+1 / s / s-1 of the current slab s,
  // R = the depth that completed in the previous step and is being written out.  K0 starts every step from zero
  // (first MFMA with C = 0); at the end of a step the roles shift by plain register moves (R <- K2 <- K1 <- K0),
  // which keeps ONE straight-line instantiation per step mask.
  f32x16 K0a, K0b, K1a, K1b, K2a, K2b, Ra, Rb;
#pragma unroll
  for (int i = 0; i < 16; ++i) K0a[i] = K0b[i] = K1a[i] = K1b[i] = K2a[i] = K2b[i] = Ra[i] = Rb[i] = 0.f;

  // prologue: first slab, resident weights (same LDS-DMA, source-side swizzle), bias
  const int sbase = TWOD ? d0 : d0 - 1;           // first slab of the march
  if (sbase >= 0) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) slab_piece(sbase, k, soff_s[k * MG::THREADS + tid]);
  }
  for (int j = wave; j < 27 * 32 * MG::CH / 64; j += 8) {
    const int i = j * 64 + lane;
    const int row = i >> 2, c = (i & 3) ^ MG::swz(row);
    lds_dma(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * MG::CIN + c * 8, wbuf + j * 1024);
  }
  if (tid < 32) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
  dma_wait();
  block_sync();

  // write-out of a finished depth o: lane = output channel r, registers = 16 voxels of the M-tile's w-row.
  // retire_pair handles voxels i0, i0 + 1 of both M-tiles (called from 8 stages), retire_zero clears the set.
  // write-out of a finished depth: the accumulator layout is lane = channel, registers = 16 voxels, i.e. 2-byte
  // pieces per lane - as global stores that is 32 instructions per wave and depth and the memory pipeline, not the
  // matrix core, sets the pace.  Instead the bf16 values are transposed through a 2-KiB per-wave LDS tile
  // ([voxel][channel]) and leave as 16-byte stores: 64 lanes x 16 B = 16 voxels x 32 channels = whole 128-B lines.
  const float bv = bias_s[r];
  float ssum = 0.f, qsum = 0.f;
  char* stg = reinterpret_cast<char*>(soff_s + NPIECE * MG::THREADS) + wave * MG::STAGE_BYTES;
  char* stg_w = stg + (4 * khalf) * 64 + r * 2;           // + wu(i) * 64: an immediate
  const char* stg_r = stg + lane * 16;                    // voxel lane >> 2 (+16), channels 8 * (lane & 3) ..
  unsigned wmask = 0;                                     // bit i: voxel i of the w-row lies inside the volume
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (w0 + (i & 3) + 8 * (i >> 2) + 4 * khalf < W) wmask |= 1u << i;
  const bool hok0 = h0 + wave * 2 < H, hok1 = h0 + wave * 2 + 1 < H;
  const unsigned ldy2 = (unsigned)ldy * 2u;
  // split output (the data gradient of a conv on concatenated inputs): channel blocks >= ysplit go to y1
  bf16_t* ysel = bid.y >= ysplit ? y1 + (bid.y - ysplit) * 32 : y + n0;
  char* yn = reinterpret_cast<char*>(ysel) + (((int64_t)n * D * H + (h0 + wave * 2)) * W + w0) * ldy * 2;
  const int64_t yslice = (int64_t)H * W * ldy * 2;        // bytes per output depth
  const unsigned soffb = (unsigned)(lane >> 2) * ldy2 + (unsigned)(lane & 3) * 16u;
  const bool sok0 = w0 + (lane >> 2) < W, sok1 = w0 + (lane >> 2) + 16 < W;
  const float slope_v = ACT ? *slope_p : 0.f;
  auto retire_elem = [&](f32x16& A, int m, int i) {       // voxel i of M-tile m: statistics + bf16 into the LDS tile
    const int wu = (i & 3) + 8 * (i >> 2);
    float ov = A[i] + bv;
    if (ACT) ov = ov > 0.f ? ov : ov * slope_v;
    *reinterpret_cast<bf16_t*>(stg_w + wu * 64) = (bf16_t)ov;
#ifndef FPLX_ABL_NOSTATS
    if (!ACT && (m ? hok1 : hok0) && ((wmask >> i) & 1u)) {
      ssum += ov;
      qsum = fmaf(ov, ov, qsum);
    }
#endif
  };
  auto retire_flush = [&](int m, int o) {                 // LDS tile -> y, two 16-byte stores per lane
    if (m ? hok1 : hok0) {
      unsigned l2 = ldy2;
      asm volatile("" : "+s"(l2));
      char* rowp = yn + o * yslice + (unsigned)(m * W) * l2;            // uniform
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(stg_r);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(stg_r + 1024);
      // inline asm: a store hipcc knows about makes it guard later register reuse with vmcnt(N) waits, and since it
      // does not know about the DMA pieces in flight, those waits end up waiting for the DMA.
      // s_nop 1: a VMEM store of more than 8 bytes reads its data VGPRs for two more cycles ("12-dword store" hazard:
      // a VALU write of those registers needs 2 wait states on gfx940+); hipcc's hazard recognizer does not look inside
      // inline asm, and the register allocator reuses v0 / v1 at once.  Without the nop the first dword of the store's
      // last lanes picked up the next instruction's result whenever another kernel's waves shared the SIMD
      // (profiles/r02_race25_hazard_location.txt: the 2.5D stream-order hazard of round 1)
      if (sok0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(v0) : "memory");
      if (sok1) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + 16u * l2 + soffb), "v"(v1) : "memory");
    }
  };

  // slab s = d0 - 1 + t touches output depths s+1 (kd = 0), s (kd = 1), s-1 (kd = 2); MASK = which of them lie in
  // [d0, d1): 1, 3, 7 ... 7, 6, 4 over the block's nd + 2 slabs.  The depth that completed in step t - 1 is written
  // out during step t (stages 0..8: M-tile 0 -> LDS tile, flush, M-tile 1, flush); the DMA of slab s + 1 goes out in stages
  // 0..NPIECE-1; both are long finished when the step's closing vmcnt(0) + barrier is reached.
  // TWOD: slabs d0 .. d1-1 (step mask 2) and one closing step; a result is written out two steps after its slab, like
  // the kd = 1 plane of the 3D march (K1 -> K2 -> R).
  const int nd = d1 - d0;                         // >= 2 (march_cfg)
  const int nsteps = TWOD ? nd + 1 : nd + 2;
#ifdef FPLX_STAMP
  // diagnostic build only (tools/micro/march_bench.hip): shader-clock stamps around the three phases of a step
  long long st_step = 0, st_wait = 0, st_bar = 0;
  const long long st_begin = __builtin_amdgcn_s_memtime();
  const long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = 0; t < nsteps; ++t) {
#ifdef FPLX_STAMP
    const long long st0 = __builtin_amdgcn_s_memtime();
#endif
    const int s = sbase + t;
    const bool fetch = TWOD ? s + 1 < d1 : (s + 1 <= d1 && s + 1 < D);
    const bool wout = t >= (TWOD ? 2 : 3);
    const int o = s - 2;                           // depth written out during this step
    const char* sl = slabs + ((s + 1) & 1) * MG::SLAB_BYTES;
    int so[NPIECE];                                 // one batch of LDS reads, consumed over the first stages
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) so[k] = soff_s[k * MG::THREADS + tid];
    auto side = [&](int q) {
#ifndef FPLX_ABL_NODMA
      if (q < NPIECE && fetch) slab_piece(s + 1, q, so[q < NPIECE ? q : 0]);
#endif
#ifdef FPLX_ABL_NORETIRE
      return;
#endif
      // write-out of the depth that completed in the previous step: stages 0-3 M-tile 0 -> LDS tile, 4 flush,
      // 4-7 M-tile 1, 8 flush.  (Staggering the two waves of a SIMD - waves 4-7 in stages 9..17 - was tried: it
      // keeps the set alive for the whole step and the spills cost more than the overlap gains.)
      if (wout && q < 9) {
        if (q < 4) { retire_elem(Ra, 0, 4 * q); retire_elem(Ra, 0, 4 * q + 1); retire_elem(Ra, 0, 4 * q + 2); retire_elem(Ra, 0, 4 * q + 3); }
        if (q == 4) retire_flush(0, o);
        if (q >= 4 && q < 8) { retire_elem(Rb, 1, 4 * q - 16); retire_elem(Rb, 1, 4 * q - 15); retire_elem(Rb, 1, 4 * q - 14); retire_elem(Rb, 1, 4 * q - 13); }
        if (q == 8) retire_flush(1, o);
      }
    };
#define MARCH_STEP(MASK) march_step<MASK>(sl, wbuf, wave, r, khalf, K0a, K0b, K1a, K1b, K2a, K2b, side)
    if (TWOD ? t < nd : (s >= 0 && s < D)) {   // a padding slab contributes nothing to the accumulators
      if constexpr (TWOD) {
        MARCH_STEP(2);
      } else {
        if (t == 0) MARCH_STEP(1);
        else if (t == 1) MARCH_STEP(3);
        else if (t < nd) MARCH_STEP(7);
        else if (t == nd) MARCH_STEP(6);
        else MARCH_STEP(4);
      }
    } else {                                           // ... but the DMA and the write-out still have to happen
#pragma unroll
      for (int q = 0; q < 9; ++q) side(q);
#pragma unroll
      for (int i = 0; i < 16; ++i) K0a[i] = K0b[i] = 0.f;
    }
#undef MARCH_STEP
    Ra = K2a; Rb = K2b; K2a = K1a; K2b = K1b; K1a = K0a; K1b = K0b;
#ifdef FPLX_STAMP
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif
    dma_wait();                                        // slab s + 1 landed and the stores left stages ago
#ifdef FPLX_STAMP
    const long long st2 = __builtin_amdgcn_s_memtime();
#endif
    block_sync();
#ifdef FPLX_STAMP
    const long long st3 = __builtin_amdgcn_s_memtime();
    st_step += st1 - st0; st_wait += st2 - st1; st_bar += st3 - st2;
#endif
  }
#ifdef FPLX_STAMP
  if (lane == 0) {
    long long* o = fplx_stamp_buf + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 6;
    o[0] = st_step; o[1] = st_wait; o[2] = st_bar; o[3] = __builtin_amdgcn_s_memtime() - st_begin; o[4] = nsteps;
    o[5] = __builtin_amdgcn_s_memrealtime() - st_rbegin;        // 100 MHz ticks over the loop
  }
#endif
  // drain: the last depth completed in the final step
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) retire_elem(Ra, 0, i);
    retire_flush(0, d1 - 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) retire_elem(Rb, 1, i);
    retire_flush(1, d1 - 1);
  }

  if (stats) {
    FPLX_LDS_BARRIER();
    float* red = reinterpret_cast<float*>(smem);            // [8 waves][2][32]; the slabs are dead
    const float a = ssum + __shfl_xor(ssum, 32, 64), q2 = qsum + __shfl_xor(qsum, 32, 64);
    if (lane < 32) { red[(wave * 2 + 0) * 32 + r] = a; red[(wave * 2 + 1) * 32 + r] = q2; }
    FPLX_LDS_BARRIER();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) t += red[(wv * 2 + which) * 32 + c];
      stats[((int64_t)bid.x * 2 + which) * Cout + n0 + c] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv_fwd_march32v2: the Cin = 32 depth march re-cut for ONE wave per SIMD (4 waves, up to 512 registers each), for
// footprints that lie entirely inside the volume (H % 16 == 0, W % 32 == 0) - the benchmark's level 0.
// What the stamps of round 2 showed about the 8-wave kernel above (tools/micro/march_bench.hip): the matrix pipe is busy
// 52 % of a block's cycles; a depth step costs 9650 cycles against 6144 of MFMA issue, and the difference is vector work
// of the wave pair sharing a SIMD - about 7 VALU + 5 SALU instructions per MFMA: write-out with masked statistics (1600
// cycles), 96 register moves for the accumulator role shift, per-stage fragment addresses, the DMA's address selects.
// Here a wave owns FOUR M-tiles (rows 4 wave .. 4 wave + 3 of the 16 x 32 footprint):
//   * a group = one (kw, k-half): the 6 slab rows the four M-tiles touch are read ONCE (6 A fragments) and serve the 12
//     (M-tile, kh) pairs x 3 kd = 36 MFMAs with 9 weight fragments: 0.42 ds_read_b128 per MFMA (0.83 above);
//   * the accumulator roles rotate by NAME (four unrolled copies of the step), no register moves;
//   * the slab DMA goes through a buffer descriptor: halo and padding come out of the hardware's range check as zeros,
//     a piece costs a handful of scalar instructions and no vector one;
//   * statistics are unmasked (interior footprints only); segment borders are handled by data, not by flags (below).
// LDS: two 39-KB slab slots, the 27 x 32 x 32 weights (54 KB), two 2-KiB store-transpose tiles per wave, the DMA offsets.
struct MG2 {
  static constexpr int CIN = 32, ROWB = 64, CH = 4;
  static constexpr int FH = 16, FW = 32, SH = FH + 2, SW = FW + 2, SLAB = SH * SW;
  static constexpr int THREADS = 256, WAVES = 4;
  static constexpr int SLAB_CHUNKS = SLAB * CH, SLAB_DMA = (SLAB_CHUNKS + 63) / 64;
  static constexpr int SLAB_BYTES = SLAB_DMA * 1024, W_BYTES = 27 * 32 * ROWB;
  static constexpr int NPIECE = (SLAB_DMA + WAVES - 1) / WAVES;       // DMA wave-instructions per wave and slab (10)
  static constexpr int STAGE_BYTES = 32 * 32 * 2;
  static constexpr int LDS = 2 * SLAB_BYTES + W_BYTES + 32 * 4 + WAVES * 2 * STAGE_BYTES + NPIECE * THREADS * 4;
  static __device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
};

template <bool STATS>
__global__ void __launch_bounds__(MG2::THREADS)
conv_fwd_march32v2(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                   const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                   int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen,
                   bf16_t* __restrict__ y1, int ysplit, int xcd) {
  using G = MG2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabs = smem;
  char* wbuf = smem + 2 * G::SLAB_BYTES;
  float* bias_s = reinterpret_cast<float*>(wbuf + G::W_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, khalf = lane >> 5;
  const FplxBlock bid = fplx_xcd_block(xcd);
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int h0 = __builtin_amdgcn_readfirstlane(th * G::FH), w0 = __builtin_amdgcn_readfirstlane(tw * G::FW);
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = bid.y * 32;

  auto lds_dma = [&](const void* g, char* l) {          // see conv_fwd_march32 (the resident weights come this way)
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)l));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int NPIECE = G::NPIECE;
  // Slab DMA through a buffer descriptor over this sample's volume (base x[n], num_records = D slices): `buffer_load_dwordx4
  // ... offen lds` with a per-lane byte offset inside the depth slice (a constant of the whole march), the depth as the
  // scalar offset, and 0x40000000 for halo voxels outside the volume and the tail of the last 1-KiB piece - out of range
  // (the launcher keeps a sample below 1 GiB), and the hardware writes ZEROS into LDS for such lanes
  // (tools/micro/bufload_lds_test.hip; the scalar offset takes part in the range check on gfx950).  A depth slice outside
  // the volume gets the scalar offset 0x40000000: every lane out of range, no 32-bit wrap, a slab of zeros = the padding.
  // The ten offsets of a lane are parked in LDS and read back a group ahead of their use: as registers they push the step
  // into spills, and a scratch reload waits on vmcnt, i.e. on the DMA in flight.
  unsigned* voff_s = reinterpret_cast<unsigned*>(smem + 2 * G::SLAB_BYTES + G::W_BYTES + 32 * 4 + G::WAVES * 2 * G::STAGE_BYTES);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    int piece = wave + G::WAVES * k;
    if (piece > G::SLAB_DMA - 1) piece = G::SLAB_DMA - 1;      // wave 3's tenth piece repeats wave 2's (same bytes, same place)
    const int i = piece * 64 + lane;
    const int vox = i >> 2, c = (i & 3) ^ G::swz(vox);
    const int hh = vox / G::SW + h0 - 1, ww = vox % G::SW + w0 - 1;
    const bool in = i < G::SLAB_CHUNKS && hh >= 0 && hh < H && ww >= 0 && ww < W;
    voff_s[k * G::THREADS + tid] = in ? (unsigned)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : 0x40000000u;
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2;
  const char* xn = reinterpret_cast<const char*>(x) + (int64_t)n * D * xslice;
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xn);
  rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xn >> 32) & 0xFFFFu);
  rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * xslice));
  rsrc[3] = 0x00020000u;
  const unsigned xslice32 = __builtin_amdgcn_readfirstlane((unsigned)xslice);
  auto slab_piece = [&](int s, int k, unsigned vo) {    // piece k of slab s -> slot (s + 1) & 1; vo = voff_s[k][tid]
    int piece = wave + G::WAVES * k;
    if (piece > G::SLAB_DMA - 1) piece = G::SLAB_DMA - 1;
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)((__attribute__((address_space(3))) char*)(slabs + ((s + 1) & 1) * G::SLAB_BYTES + piece * 1024)));
    const unsigned so = __builtin_amdgcn_readfirstlane((s >= 0 && s < D) ? (unsigned)s * xslice32 : 0x40000000u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so), "s"(dst) : "memory");
  };

  // accumulator sets: at step t role j (K0 = depth s + 1, K1 = s, K2 = s - 1, R = being written out) is set (j - t) & 3
  f32x16 S[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) S[a][m][i] = 0.f;

  __syncthreads();                                       // voff_s is complete
  const int sbase = d0 - 1;
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) slab_piece(sbase, k, voff_s[k * G::THREADS + tid]);
  for (int j = wave; j < 27 * 32 * G::CH / 64; j += G::WAVES) {
    const int i = j * 64 + lane;
    const int row = i >> 2, c = (i & 3) ^ G::swz(row);
    lds_dma(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * G::CIN + c * 8, wbuf + j * 1024);
  }
  if (tid < 32) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
  dma_wait();
  block_sync();

  const float bv = bias_s[r];
  float ssum = 0.f, qsum = 0.f;
  char* stg = reinterpret_cast<char*>(bias_s + 32) + wave * 2 * G::STAGE_BYTES;       // two tiles per wave, used alternately
  const unsigned ldy2 = (unsigned)ldy * 2u;
  bf16_t* ysel = bid.y >= ysplit ? y1 + (bid.y - ysplit) * 32 : y + n0;
  char* yn = reinterpret_cast<char*>(ysel) + (((int64_t)n * D * H + (h0 + wave * 4)) * W + w0) * ldy * 2;
  const int64_t yslice = (int64_t)H * W * ldy * 2;
  const unsigned soffb = (unsigned)(lane >> 2) * ldy2 + (unsigned)(lane & 3) * 16u;
  // M-tile m of set A: bias, statistics, bf16 -> LDS tile (m & 1) -> two 16-byte stores per lane
  auto retire_elems = [&](const f32x16& A, int m, int i0, int cnt) {
    char* w_ = stg + (m & 1) * G::STAGE_BYTES + (4 * khalf) * 64 + r * 2;
#pragma unroll
    for (int i = i0; i < i0 + cnt; ++i) {
      const int wu = (i & 3) + 8 * (i >> 2);
      const float ov = A[i] + bv;
      *reinterpret_cast<bf16_t*>(w_ + wu * 64) = (bf16_t)ov;
      if (STATS) { ssum += ov; qsum = fmaf(ov, ov, qsum); }
    }
  };
  auto retire_flush = [&](int m, int o) {
    const char* r_ = stg + (m & 1) * G::STAGE_BYTES + lane * 16;
    unsigned l2 = ldy2;
    asm volatile("" : "+s"(l2));
    char* rowp = yn + o * yslice + (unsigned)(m * W) * l2;
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(r_);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(r_ + 1024);
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(v0) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + 16u * l2 + soffb), "v"(v1) : "memory");
  };

  const int nd = d1 - d0;                         // >= 2 (march_cfg)
  const int nsteps = nd + 2;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // One depth step, the same code for EVERY step of the segment; ROT = t & 3 names the accumulator sets.
  // Segment borders are handled by data, not by special steps:
  //   * a slab outside the volume (s = -1, s = D) is fetched like any other - its scalar offset is out of the descriptor's
  //     range, so the DMA fills the slot with zeros (= the padding) and the MFMAs add nothing;
  //   * depth taps whose output depth lies outside [d0, d1) are skipped by block-uniform branches around 12 MFMAs each;
  //   * every step writes out its R set: before step 3 that set holds no finished depth - the stores go to depth d0, which
  //     step 3 overwrites (the block's vmcnt(0) in between retires them first), and the statistics restart at step 3.
  // A step = 6 groups (kw, k-half) x 3 depth taps x 12 MFMAs.  Per group: the tile written in the previous group is read
  // back, M-tile g of R goes to the other tile (8 elements beside the kd = 0 MFMAs, 8 beside the kd = 2 ones), the previous
  // tile's two 16-byte stores, the slab DMA and next group's A fragments go beside the kd = 1 MFMAs; the three weight
  // fragments of a depth tap are reloaded for the next group right behind their 12 MFMAs.
  // The fragments of a step's FIRST group are loaded by the step before it: the block's vmcnt(0) + barrier sits behind
  // group 4 (the last group that reads this slab's A fragments; by then the next slab's ten DMA pieces, issued in groups
  // 0-3, have had time to land), so group 5 already reads the next slab and no step begins with an empty pipe.
  bf16x8 fa[2][6], fb[9];
  auto load_a = [&](const char* sl, int g, int buf) {    // g = kw * 2 + ks
    const int kw = g >> 1, ks = g & 1, c = 2 * ks + khalf;
    int vb = wave * 4 * G::SW + r;
    asm volatile("" : "+v"(vb));                         // lane bases re-derived at the point of use (no hoisted address zoo)
#pragma unroll
    for (int rho = 0; rho < 6; ++rho) {
      const int vox = vb + rho * G::SW + kw;
      fa[buf][rho] = *reinterpret_cast<const bf16x8*>(sl + vox * G::ROWB + ((c ^ G::swz(vox)) << 4));
    }
  };
  auto load_b = [&](int g, int kd) {
    const int kw = g >> 1, ks = g & 1;
    int rb = r;
    asm volatile("" : "+v"(rb));
    const char* wl = wbuf + rb * G::ROWB + (((2 * ks + khalf) ^ G::swz(rb)) << 4);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
      fb[kd * 3 + kh] = *reinterpret_cast<const bf16x8*>(wl + ((kd * 9 + kh * 3 + kw) * 32) * G::ROWB);
  };
  load_a(slabs + ((sbase + 1) & 1) * G::SLAB_BYTES, 0, 0);
  load_b(0, 0); load_b(0, 1); load_b(0, 2);
  auto step = [&](auto rot_c, int t) {
    constexpr int ROT = decltype(rot_c)::value;
    const int s = sbase + t;
    const int o = (s - 2 > d0) ? s - 2 : d0;
    const char* sl = slabs + ((s + 1) & 1) * G::SLAB_BYTES;
    const char* sl_next = slabs + (s & 1) * G::SLAB_BYTES;
    const bool on0 = t < nd, on1 = t >= 1 && t <= nd, on2 = t >= 2;
    f32x16 (&K0)[4] = S[(0 - ROT) & 3];
    f32x16 (&K1)[4] = S[(1 - ROT) & 3];
    f32x16 (&K2)[4] = S[(2 - ROT) & 3];
    f32x16 (&R)[4] = S[(3 - ROT) & 3];
    u32x4 fl0, fl1;
    unsigned vo0 = 0, vo1 = 0, vo2 = 0;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      const int bf = g & 1;
      const int p0 = g < 2 ? 3 * g : 2 * g + 2, np = g < 2 ? 3 : (g < 4 ? 2 : 0);     // DMA pieces of this group: 3, 3, 2, 2
      // ---- kd = 0: read back the previous tile (and this group's DMA offsets), first half of M-tile g of R
      if (np > 0) {
        vo0 = voff_s[p0 * G::THREADS + tid];
        vo1 = voff_s[(p0 + 1) * G::THREADS + tid];
        if (np > 2) vo2 = voff_s[(p0 + 2) * G::THREADS + tid];
      }
      if (g >= 1 && g <= 4) {
        const char* r_ = stg + ((g - 1) & 1) * G::STAGE_BYTES + lane * 16;
        fl0 = *reinterpret_cast<const u32x4*>(r_);
        fl1 = *reinterpret_cast<const u32x4*>(r_ + 1024);
      }
      if (g < 4) retire_elems(R[g], g, 0, 8);
      if (on0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            K0[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bf][m + kh], fb[0 + kh], (g == 0 && kh == 0) ? zero : K0[m], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- kd = 1: next group's A fragments and kd = 0 weights, the previous tile's stores, the slab DMA
      if (g < 5) load_a(sl, g + 1, bf ^ 1); else load_a(sl_next, 0, 0);
      load_b((g + 1) % 6, 0);
      if (g >= 1 && g <= 4) {
        unsigned l2 = ldy2;
        asm volatile("" : "+s"(l2));
        char* rowp = yn + o * yslice + (unsigned)((g - 1) * W) * l2;
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(fl0) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + 16u * l2 + soffb), "v"(fl1) : "memory");
      }
      if (np > 0) {
        slab_piece(s + 1, p0, vo0);
        slab_piece(s + 1, p0 + 1, vo1);
        if (np > 2) slab_piece(s + 1, p0 + 2, vo2);
      }
      if (on1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            K1[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bf][m + kh], fb[3 + kh], K1[m], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- kd = 2: second half of M-tile g, next group's kd = 1 weights; its kd = 2 weights behind the MFMAs
      load_b((g + 1) % 6, 1);
      if (g < 4) retire_elems(R[g], g, 8, 8);
      if (on2) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            K2[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bf][m + kh], fb[6 + kh], K2[m], 0, 0, 0);
      }
      load_b((g + 1) % 6, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (g == 4) {                                      // the next slab has landed; nobody reads this one any more
        dma_wait();
        block_sync();
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
#ifdef FPLX_STAMP
  const long long st_begin = __builtin_amdgcn_s_memtime();
  const long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  int t = 0;
  for (;;) {
    step(C0{}, t); ++t;
    if (t >= nsteps) break;
    step(C1{}, t); ++t;
    if (t >= nsteps) break;
    step(C2{}, t); ++t;
    if (t >= nsteps) break;
    if (STATS && t == 3) { ssum = 0.f; qsum = 0.f; }      // what steps 0-2 wrote out (and counted) was no finished depth
    step(C3{}, t); ++t;
    if (t >= nsteps) break;
  }
#ifdef FPLX_STAMP
  if (lane == 0) {
    long long* o_ = fplx_stamp_buf + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 6;
    o_[0] = 0; o_[1] = 0; o_[2] = 0; o_[3] = __builtin_amdgcn_s_memtime() - st_begin; o_[4] = nsteps;
    o_[5] = __builtin_amdgcn_s_memrealtime() - st_rbegin;
  }
#endif
  // ---- drain: the depth completed in the final step sits in the set that would be R of step nsteps
  auto drain = [&](auto rot_c) {
    constexpr int ROT = decltype(rot_c)::value;
    f32x16 (&R)[4] = S[(3 - ROT) & 3];
#pragma unroll
    for (int m = 0; m < 4; ++m) { retire_elems(R[m], m, 0, 16); retire_flush(m, d1 - 1); }
  };
  switch (nsteps & 3) {
    case 0: drain(C0{}); break;
    case 1: drain(C1{}); break;
    case 2: drain(C2{}); break;
    default: drain(C3{}); break;
  }

  if (STATS && stats) {
    FPLX_LDS_BARRIER();
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][32]; the slabs are dead
    const float a = ssum + __shfl_xor(ssum, 32, 64), q2 = qsum + __shfl_xor(qsum, 32, 64);
    if (lane < 32) { red[(wave * 2 + 0) * 32 + r] = a; red[(wave * 2 + 1) * 32 + r] = q2; }
    FPLX_LDS_BARRIER();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float tt = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) tt += red[(wv * 2 + which) * 32 + c];
      stats[((int64_t)bid.x * 2 + which) * Cout + n0 + c] = tt;
    }
  }
}

// conv_fwd_march32v3: conv_fwd_march32v2 on v_mfma_f32_16x16x32_bf16.  The kernels of this file run against the chip's power
// limit, not its issue rate (in-kernel clock 1.4-1.6 GHz on random data, 1.8-1.9 GHz on zeros, profiles/r02_march_bench_*):
// cycles saved come back as a lower clock, energy saved does not, and the 16 x 16 x 32 shape moves the same FLOPs for less
// (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15 x the FLOP/s of the 32 x 32 x 16 loop at equal cycles per FLOP).
// Same tiling, LDS image, DMA, rotation and write-out as v2; what changes is the fragment geometry - a lane is (row or
// column r16 = lane & 15, k-group kg = lane >> 4), one MFMA spans all 32 input channels, an accumulator tile is four
// 16 x 16 blocks (voxel half x cout half) - and the weight image's swizzle (wswz).
template <bool STATS, int ASWZ, bool ACT = false>
__global__ void __launch_bounds__(MG2::THREADS)
conv_fwd_march32v3(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                   const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                   int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen,
                   bf16_t* __restrict__ y1, int ysplit, int xcd, const float* __restrict__ slope_p = nullptr) {
  using G = MG2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabs = smem;
  char* wbuf = smem + 2 * G::SLAB_BYTES;
  float* bias_s = reinterpret_cast<float*>(wbuf + G::W_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;          // v_mfma_f32_16x16x32_bf16: lane = (row | column, k-group of 8)
  const FplxBlock bid = fplx_xcd_block(xcd);
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int h0 = __builtin_amdgcn_readfirstlane(th * G::FH), w0 = __builtin_amdgcn_readfirstlane(tw * G::FW);
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = bid.y * 32;

  // weight rows are read 16 at a time from a 16-aligned row (one cout half): a ds_read_b128 lane group then holds rows
  // {0-3, 12-15} of one k-group and rows {4-11} of the next, and the chunk swizzle [0, 3, 2, 1] by row quad keeps its 16
  // lanes on 16 different bank quads (the slab keeps the (row >> 2) & 3 swizzle: its fragments start at any voxel, a
  // fixed function cannot serve every alignment, and a 2-way conflict on a third of the reads is far from the LDS limit)
  auto wswz = [](int row) { const int q = (row >> 2) & 3; return (4 - q) & 3; };
  // slab image: the 16 x 16 x 32 A fragments (lane = voxel r16, 16-byte chunk kg = lane >> 4) are conflict-free with the chunk
  // XORed by 2 * ((vox >> 2) & 1); the (vox >> 2) & 3 of the 32 x 32 x 16 kernels costs this read pattern 2-way conflicts in
  // most lane groups (SQ_LDS_BANK_CONFLICT 0.31 of the LDS cycles, profiles/r02_pmc_sq_counters.txt)
  auto aswz = [](int vox) { return ASWZ ? ((vox >> 2) & 1) << 1 : (vox >> 2) & 3; };
  auto lds_dma = [&](const void* g, char* l) {          // see conv_fwd_march32 (the resident weights come this way)
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)l));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int NPIECE = G::NPIECE;
  // Slab DMA through a buffer descriptor over this sample's volume (base x[n], num_records = D slices): `buffer_load_dwordx4
  // ... offen lds` with a per-lane byte offset inside the depth slice (a constant of the whole march), the depth as the
  // scalar offset, and 0x40000000 for halo voxels outside the volume and the tail of the last 1-KiB piece - out of range
  // (the launcher keeps a sample below 1 GiB), and the hardware writes ZEROS into LDS for such lanes
  // (tools/micro/bufload_lds_test.hip; the scalar offset takes part in the range check on gfx950).  A depth slice outside
  // the volume gets the scalar offset 0x40000000: every lane out of range, no 32-bit wrap, a slab of zeros = the padding.
  // The ten offsets of a lane are parked in LDS and read back a group ahead of their use: as registers they push the step
  // into spills, and a scratch reload waits on vmcnt, i.e. on the DMA in flight.
  unsigned* voff_s = reinterpret_cast<unsigned*>(smem + 2 * G::SLAB_BYTES + G::W_BYTES + 32 * 4 + G::WAVES * 2 * G::STAGE_BYTES);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    int piece = wave + G::WAVES * k;
    if (piece > G::SLAB_DMA - 1) piece = G::SLAB_DMA - 1;      // wave 3's tenth piece repeats wave 2's (same bytes, same place)
    const int i = piece * 64 + lane;
    const int vox = i >> 2, c = (i & 3) ^ aswz(vox);
    const int hh = vox / G::SW + h0 - 1, ww = vox % G::SW + w0 - 1;
    const bool in = i < G::SLAB_CHUNKS && hh >= 0 && hh < H && ww >= 0 && ww < W;
    voff_s[k * G::THREADS + tid] = in ? (unsigned)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : 0x40000000u;
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2;
  const char* xn = reinterpret_cast<const char*>(x) + (int64_t)n * D * xslice;
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xn);
  rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xn >> 32) & 0xFFFFu);
  rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * xslice));
  rsrc[3] = 0x00020000u;
  const unsigned xslice32 = __builtin_amdgcn_readfirstlane((unsigned)xslice);
  auto slab_piece = [&](int s, int k, unsigned vo) {    // piece k of slab s -> slot (s + 1) & 1; vo = voff_s[k][tid]
    int piece = wave + G::WAVES * k;
    if (piece > G::SLAB_DMA - 1) piece = G::SLAB_DMA - 1;
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)((__attribute__((address_space(3))) char*)(slabs + ((s + 1) & 1) * G::SLAB_BYTES + piece * 1024)));
    const unsigned so = __builtin_amdgcn_readfirstlane((s >= 0 && s < D) ? (unsigned)s * xslice32 : 0x40000000u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so), "s"(dst) : "memory");
  };

  // accumulator sets: at step t role j (K0 = depth s + 1, K1 = s, K2 = s - 1, R = being written out) is set (j - t) & 3
  f32x4 S[4][4][4];                                      // [set][M-tile][voxel half * 2 + cout half]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) S[a][m][q][i] = 0.f;

  __syncthreads();                                       // voff_s is complete
  const int sbase = d0 - 1;
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) slab_piece(sbase, k, voff_s[k * G::THREADS + tid]);
  for (int j = wave; j < 27 * 32 * G::CH / 64; j += G::WAVES) {
    const int i = j * 64 + lane;
    const int row = i >> 2, c = (i & 3) ^ wswz(row);
    lds_dma(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * G::CIN + c * 8, wbuf + j * 1024);
  }
  if (tid < 32) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
  dma_wait();
  block_sync();

  const float bv0 = bias_s[r16], bv1 = bias_s[16 + r16];
  float ssum0 = 0.f, qsum0 = 0.f, ssum1 = 0.f, qsum1 = 0.f;
  char* stg = reinterpret_cast<char*>(bias_s + 32) + wave * 2 * G::STAGE_BYTES;       // two tiles per wave, used alternately
  const unsigned ldy2 = (unsigned)ldy * 2u;
  bf16_t* ysel = bid.y >= ysplit ? y1 + (bid.y - ysplit) * 32 : y + n0;
  char* yn = reinterpret_cast<char*>(ysel) + (((int64_t)n * D * H + (h0 + wave * 4)) * W + w0) * ldy * 2;
  const int64_t yslice = (int64_t)H * W * ldy * 2;
  const unsigned soffb = (unsigned)(lane >> 2) * ldy2 + (unsigned)(lane & 3) * 16u;
  // M-tile m of set A: bias, statistics, bf16 -> LDS tile (m & 1) -> two 16-byte stores per lane.  A lane holds, for its
  // two channels r16 and 16 + r16, the voxels 4 kg + i of either 16-voxel half: half mh is written by retire_half(.., mh)
  const float slope_v = ACT ? *slope_p : 0.f;
  auto retire_half = [&](const f32x4 (&A)[4], int m, int mh) {
    char* w_ = stg + (m & 1) * G::STAGE_BYTES + (mh * 16 + 4 * kg) * 64 + r16 * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o0 = A[mh * 2 + 0][i] + bv0, o1 = A[mh * 2 + 1][i] + bv1;
      if (ACT) { o0 = o0 > 0.f ? o0 : o0 * slope_v; o1 = o1 > 0.f ? o1 : o1 * slope_v; }
      *reinterpret_cast<bf16_t*>(w_ + i * 64) = (bf16_t)o0;
      *reinterpret_cast<bf16_t*>(w_ + i * 64 + 32) = (bf16_t)o1;
      if (STATS) { ssum0 += o0; qsum0 = fmaf(o0, o0, qsum0); ssum1 += o1; qsum1 = fmaf(o1, o1, qsum1); }
    }
  };
  auto retire_flush = [&](int m, int o) {
    const char* r_ = stg + (m & 1) * G::STAGE_BYTES + lane * 16;
    unsigned l2 = ldy2;
    asm volatile("" : "+s"(l2));
    char* rowp = yn + o * yslice + (unsigned)(m * W) * l2;
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(r_);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(r_ + 1024);
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(v0) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + 16u * l2 + soffb), "v"(v1) : "memory");
  };

  const int nd = d1 - d0;                         // >= 2 (march_cfg)
  const int nsteps = nd + 2;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // One depth step (see conv_fwd_march32v2, whose schedule this kernel keeps), the same code for EVERY step of the segment.
  // Segment borders are handled by data, not by special steps:
  //   * a slab outside the volume (s = -1, s = D) is fetched like any other - its scalar offset is out of the descriptor's
  //     range, so the DMA fills the slot with zeros (= the padding) and the MFMAs add nothing;
  //   * depth taps whose output depth lies outside [d0, d1) are skipped by block-uniform branches around 12 MFMAs each;
  //   * every step writes out its R set: before step 3 that set holds no finished depth - the stores go to depth d0, which
  //     step 3 overwrites (the block's vmcnt(0) in between retires them first), and the statistics restart at step 3.
  // A step = 6 groups (kw, k-half) x 3 depth taps x 12 MFMAs.  Per group: the tile written in the previous group is read
  // back, M-tile g of R goes to the other tile (8 elements beside the kd = 0 MFMAs, 8 beside the kd = 2 ones), the previous
  // tile's two 16-byte stores, the slab DMA and next group's A fragments go beside the kd = 1 MFMAs; the three weight
  // fragments of a depth tap are reloaded for the next group right behind their 12 MFMAs.
  // The fragments of a step's FIRST group are loaded by the step before it: the block's vmcnt(0) + barrier sits behind
  // group 4 (the last group that reads this slab's A fragments; by then the next slab's ten DMA pieces, issued in groups
  // 0-3, have had time to land), so group 5 already reads the next slab and no step begins with an empty pipe.
  // v_mfma_f32_16x16x32_bf16 covers all 32 input channels at once: a group is (kw, 16-voxel half mh) - 6 A fragments (the
  // half's six slab rows) x the 18 weight fragments (kd, kh, cout half) of the kw, which stay in registers for both halves
  // and are reloaded for the next kw behind their MFMAs of the second half: the same 0.42 reads per 32 cycles of MFMA.
  bf16x8 fa[2][6], fb[18];
  auto load_a = [&](const char* sl, int g, int buf) {    // g = kw * 2 + mh
    const int kw = g >> 1, mh = g & 1;
    int vb = wave * 4 * G::SW + r16;
    asm volatile("" : "+v"(vb));
#pragma unroll
    for (int rho = 0; rho < 6; ++rho) {
      const int vox = vb + rho * G::SW + kw + mh * 16;
      fa[buf][rho] = *reinterpret_cast<const bf16x8*>(sl + vox * G::ROWB + ((kg ^ aswz(vox)) << 4));
    }
  };
  auto load_b = [&](int kw, int kd) {                    // the six fragments (kh, cout half) of depth tap kd
    int rb = r16;
    asm volatile("" : "+v"(rb));
    const char* wl = wbuf + rb * G::ROWB + ((kg ^ wswz(rb)) << 4);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
        fb[kd * 6 + kh * 2 + nh] = *reinterpret_cast<const bf16x8*>(wl + (((kd * 9 + kh * 3 + kw) * 32) + nh * 16) * G::ROWB);
  };
  load_a(slabs + ((sbase + 1) & 1) * G::SLAB_BYTES, 0, 0);
  load_b(0, 0); load_b(0, 1); load_b(0, 2);
  auto step = [&](auto rot_c, int t) {
    constexpr int ROT = decltype(rot_c)::value;
    const int s = sbase + t;
    const int o = (s - 2 > d0) ? s - 2 : d0;
    const char* sl = slabs + ((s + 1) & 1) * G::SLAB_BYTES;
    const char* sl_next = slabs + (s & 1) * G::SLAB_BYTES;
    const bool on0 = t < nd, on1 = t >= 1 && t <= nd, on2 = t >= 2;
    f32x4 (&K0)[4][4] = S[(0 - ROT) & 3];
    f32x4 (&K1)[4][4] = S[(1 - ROT) & 3];
    f32x4 (&K2)[4][4] = S[(2 - ROT) & 3];
    f32x4 (&R)[4][4] = S[(3 - ROT) & 3];
    u32x4 fl0, fl1;
    unsigned vo0 = 0, vo1 = 0, vo2 = 0;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      const int bf = g & 1, kw = g >> 1, mh = g & 1;
      const int p0 = g < 2 ? 3 * g : 2 * g + 2, np = g < 2 ? 3 : (g < 4 ? 2 : 0);     // DMA pieces of this group: 3, 3, 2, 2
      // ---- kd = 0: read back the previous tile (and this group's DMA offsets), first half of M-tile g of R
      if (np > 0) {
        vo0 = voff_s[p0 * G::THREADS + tid];
        vo1 = voff_s[(p0 + 1) * G::THREADS + tid];
        if (np > 2) vo2 = voff_s[(p0 + 2) * G::THREADS + tid];
      }
      if (g >= 1 && g <= 4) {
        const char* r_ = stg + ((g - 1) & 1) * G::STAGE_BYTES + lane * 16;
        fl0 = *reinterpret_cast<const u32x4*>(r_);
        fl1 = *reinterpret_cast<const u32x4*>(r_ + 1024);
      }
      if (g < 4) retire_half(R[g], g, 0);
      if (on0) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
              K0[m][mh * 2 + nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[bf][m + kh], fb[0 + kh * 2 + nh],
                                                                          (kw == 0 && kh == 0) ? zero : K0[m][mh * 2 + nh], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- kd = 1: next group's A fragments and kd = 0 weights, the previous tile's stores, the slab DMA
      if (g < 5) load_a(sl, g + 1, bf ^ 1); else load_a(sl_next, 0, 0);
      if (mh == 1) load_b((kw + 1) % 3, 0);
      if (g >= 1 && g <= 4) {
        unsigned l2 = ldy2;
        asm volatile("" : "+s"(l2));
        char* rowp = yn + o * yslice + (unsigned)((g - 1) * W) * l2;
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(fl0) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + 16u * l2 + soffb), "v"(fl1) : "memory");
      }
      if (np > 0) {
        slab_piece(s + 1, p0, vo0);
        slab_piece(s + 1, p0 + 1, vo1);
        if (np > 2) slab_piece(s + 1, p0 + 2, vo2);
      }
      if (on1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
              K1[m][mh * 2 + nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[bf][m + kh], fb[6 + kh * 2 + nh], K1[m][mh * 2 + nh], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- kd = 2: second half of M-tile g, next group's kd = 1 weights; its kd = 2 weights behind the MFMAs
      if (mh == 1) load_b((kw + 1) % 3, 1);
      if (g < 4) retire_half(R[g], g, 1);
      if (on2) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
              K2[m][mh * 2 + nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[bf][m + kh], fb[12 + kh * 2 + nh], K2[m][mh * 2 + nh], 0, 0, 0);
      }
      if (mh == 1) load_b((kw + 1) % 3, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (g == 4) {                                      // the next slab has landed; nobody reads this one any more
        dma_wait();
        block_sync();
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
#ifdef FPLX_STAMP
  const long long st_begin = __builtin_amdgcn_s_memtime();
  const long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  int t = 0;
  for (;;) {
    step(C0{}, t); ++t;
    if (t >= nsteps) break;
    step(C1{}, t); ++t;
    if (t >= nsteps) break;
    step(C2{}, t); ++t;
    if (t >= nsteps) break;
    if (STATS && t == 3) { ssum0 = qsum0 = ssum1 = qsum1 = 0.f; }      // what steps 0-2 wrote out (and counted) was no finished depth
    step(C3{}, t); ++t;
    if (t >= nsteps) break;
  }
#ifdef FPLX_STAMP
  if (lane == 0) {
    long long* o_ = fplx_stamp_buf + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 6;
    o_[0] = 0; o_[1] = 0; o_[2] = 0; o_[3] = __builtin_amdgcn_s_memtime() - st_begin; o_[4] = nsteps;
    o_[5] = __builtin_amdgcn_s_memrealtime() - st_rbegin;
  }
#endif
  // ---- drain: the depth completed in the final step sits in the set that would be R of step nsteps
  auto drain = [&](auto rot_c) {
    constexpr int ROT = decltype(rot_c)::value;
    f32x4 (&R)[4][4] = S[(3 - ROT) & 3];
#pragma unroll
    for (int m = 0; m < 4; ++m) { retire_half(R[m], m, 0); retire_half(R[m], m, 1); retire_flush(m, d1 - 1); }
  };
  switch (nsteps & 3) {
    case 0: drain(C0{}); break;
    case 1: drain(C1{}); break;
    case 2: drain(C2{}); break;
    default: drain(C3{}); break;
  }

  if (STATS && stats) {
    FPLX_LDS_BARRIER();
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][32]; the slabs are dead
    float a0 = ssum0, a1 = ssum1, q0 = qsum0, q1 = qsum1;   // lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same two channels
    a0 += __shfl_xor(a0, 16, 64); a0 += __shfl_xor(a0, 32, 64);
    a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
    q0 += __shfl_xor(q0, 16, 64); q0 += __shfl_xor(q0, 32, 64);
    q1 += __shfl_xor(q1, 16, 64); q1 += __shfl_xor(q1, 32, 64);
    if (lane < 16) {
      red[(wave * 2 + 0) * 32 + r16] = a0; red[(wave * 2 + 0) * 32 + 16 + r16] = a1;
      red[(wave * 2 + 1) * 32 + r16] = q0; red[(wave * 2 + 1) * 32 + 16 + r16] = q1;
    }
    FPLX_LDS_BARRIER();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float tt = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) tt += red[(wv * 2 + which) * 32 + c];
      stats[((int64_t)bid.x * 2 + which) * Cout + n0 + c] = tt;
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv_fwd_march64: the same input-stationary depth march for Cin = 64 (level-0 decoder conv1 on the skip | up
// concatenation, level-1 convs).  All 27 x 32 x 64 weights of the block's 32 output channels stay resident in LDS
// (108 KB) as two 32-channel halves; an input slab is streamed as two 32-channel half-slabs (8 x 32 footprint:
// 10 x 34 voxels x 64 B = 21 KB each) through a two-slot ring, so a depth step is two half-steps of 108 MFMAs per
// wave into the same accumulators.  LDS holds 162 KB, one block of 4 waves per CU, ONE wave per SIMD: nothing hides
// a stall, so every fragment read, DMA piece and output element is issued inside an MFMA gap (fences pin the order).
// Two footprints: 8 x 32 (an MFMA M-tile = 32 voxels of one row) and 16 x 16 (an M-tile = 16 voxels of two consecutive
// rows) for widths such as 80 that 32-wide tiles cover with 17 % waste.  FW16 changes only the lane -> voxel map of the
// A fragments and of the write-out; slab, weights, ring and MFMA schedule are the same.
template <int FWV>
struct MG64T {
  static constexpr int ROWB = 64, CH = 4;                  // a half-slab / half-weight row: 32 channels
  static constexpr int FW = FWV, FH = 256 / FWV, SH = FH + 2, SW = FW + 2, SLAB = SH * SW;
  static constexpr int HPM = 32 / FW;                      // rows of the footprint per M-tile (1 or 2)
  static constexpr int THREADS = 256;
  static constexpr int SLAB_CHUNKS = SLAB * CH, SLAB_DMA = (SLAB_CHUNKS + 63) / 64;   // 1-KiB pieces (last: 16 lanes)
  static constexpr int SLAB_BYTES = SLAB * ROWB, WH_BYTES = 27 * 32 * ROWB;           // per channel half
  static constexpr int NPIECE = (SLAB_DMA + 3) / 4;        // DMA wave-instructions per wave and half-slab
  static constexpr int STAGE_BYTES = 32 * 32 * 2;
  static constexpr int LDS = 2 * SLAB_BYTES + 2 * WH_BYTES + 32 * 4 + (THREADS / 64) * STAGE_BYTES;
  static __device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
};
using MG64 = MG64T<32>;

// one half-slab -> the three output depths it touches.  FIRST: the kd = 0 accumulators start from zero.
// side(q, g) runs in gap g (0..5) of stage q (0..17), i.e. right before MFMA g of that stage.
template <int MASK, bool FIRST, class G, class Side>
__device__ __forceinline__ void march64_half(const char* __restrict__ sl, const char* __restrict__ wh, int wave, int r,
                                             int khalf, f32x16& A00, f32x16& A01, f32x16& A10, f32x16& A11,
                                             f32x16& A20, f32x16& A21, Side&& side) {
  bf16x8 fa[2][2], fb[2][3];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // M-tile m of this wave starts at footprint row (2 * wave + m) * HPM; lane row r is voxel (r / FW, r % FW) of it
  int vb = (wave * 2 * G::HPM + r / G::FW) * G::SW + r % G::FW, rb = r;
  asm volatile("" : "+v"(vb), "+v"(rb));         // lane bases re-derived per half-step (no hoisted address zoo)
  const char* wl0 = wh + rb * G::ROWB + ((khalf ^ G::swz(rb)) << 4);
  const char* wl1 = wh + rb * G::ROWB + (((2 + khalf) ^ G::swz(rb)) << 4);
  auto load_a = [&](int q, int m) {
    const int p = q >> 1, ks = q & 1, kh = p / 3, kw = p % 3, c = 2 * ks + khalf;
    const int vox = vb + (m * G::HPM + kh) * G::SW + kw;
    return *reinterpret_cast<const bf16x8*>(sl + vox * G::ROWB + ((c ^ G::swz(vox)) << 4));
  };
  auto load_b = [&](int q, int kd) {
    const int p = q >> 1, ks = q & 1;
    return *reinterpret_cast<const bf16x8*>((ks ? wl1 : wl0) + (kd * 9 + p) * 32 * G::ROWB);
  };
  fa[0][0] = load_a(0, 0); fa[0][1] = load_a(0, 1);
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
    if ((MASK >> kd) & 1) fb[0][kd] = load_b(0, kd);
#pragma unroll
  for (int q = 0; q < 18; ++q) {
    const int b = q & 1, nb = b ^ 1;
    // gap g: one fragment of stage q + 1 (5 per stage), the side work, then MFMA g of stage q
#define M64_GAP(G, LOADSTMT, MFMASTMT)                                                                              \
    if (q + 1 < 18) { LOADSTMT; }                                                                                   \
    side(q, G);                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    MFMASTMT;                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);
    M64_GAP(0, fa[nb][0] = load_a(q + 1, 0),
            if constexpr ((MASK & 1) != 0) A00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][0], (FIRST && q == 0) ? zero : A00, 0, 0, 0))
    M64_GAP(1, fa[nb][1] = load_a(q + 1, 1),
            if constexpr ((MASK & 2) != 0) A10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][1], A10, 0, 0, 0))
    M64_GAP(2, if constexpr ((MASK & 1) != 0) fb[nb][0] = load_b(q + 1, 0),
            if constexpr ((MASK & 4) != 0) A20 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][2], A20, 0, 0, 0))
    M64_GAP(3, if constexpr ((MASK & 2) != 0) fb[nb][1] = load_b(q + 1, 1),
            if constexpr ((MASK & 1) != 0) A01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][0], (FIRST && q == 0) ? zero : A01, 0, 0, 0))
    M64_GAP(4, if constexpr ((MASK & 4) != 0) fb[nb][2] = load_b(q + 1, 2),
            if constexpr ((MASK & 2) != 0) A11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][1], A11, 0, 0, 0))
    M64_GAP(5, (void)0,
            if constexpr ((MASK & 4) != 0) A21 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][2], A21, 0, 0, 0))
#undef M64_GAP
  }
}

// NQ = Cin / 32 channel quarters per slab.  NQ = 2: the resident form above.  NQ = 4 (Cin = 128, the level-1 decoder
// conv1 on its 64 | 64 concatenation): 27 x 32 x 128 weights do not fit, so the two weight slots become a ring like
// the slab slots - quarter-step hf computes from slab slot hf & 1 and weight slot hf & 1 while the DMA fills the other
// two with quarter hf + 1 (55 KB of weights + 21 KB of slab per 108 MFMAs per wave, all L2 hits: the pack is shared by
// every block).  The tile kernel this replaces re-gathers its A tile from L2 for each of the 27 taps.
template <class G, bool TWOD, int NQ, bool ACT = false>          // TWOD, ACT: see conv_fwd_march32
__global__ void __launch_bounds__(256)
conv_fwd_march64(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                 const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                 int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen,
                 const bf16_t* __restrict__ x1, int xcd, const float* __restrict__ slope_p = nullptr, int nmod0 = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabs = smem;                                        // [2 channel halves][SLAB][32]
  char* wbuf = smem + 2 * G::SLAB_BYTES;                  // [2 channel halves][27][32 co][32 ci]
  float* bias_s = reinterpret_cast<float*>(wbuf + 2 * G::WH_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, khalf = lane >> 5;
  const FplxBlock bid = fplx_xcd_block(xcd);
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int h0 = __builtin_amdgcn_readfirstlane(th * G::FH), w0 = __builtin_amdgcn_readfirstlane(tw * G::FW);
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = bid.y * 32;

  auto lds_dma = [&](const void* g, const char* l) {         // see conv_fwd_march32
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((const __attribute__((address_space(3))) char*)l));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int NPIECE = G::NPIECE;
  // per lane and piece: byte offset of the source chunk (channel half 0) inside one depth slice of x, -1 = zero,
  // -2 = lane past the end of the slab (the last piece is 16 lanes wide)
  int soff[NPIECE];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int i = (wave + 4 * k) * 64 + lane;
    const int vox = i >> 2, c = (i & 3) ^ G::swz(vox);
    const int hh = vox / G::SW + h0 - 1, ww = vox % G::SW + w0 - 1;
    const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
    soff[k] = i >= G::SLAB_CHUNKS ? -2 : (in ? (int)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : -1);
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2;
  // nmod0 > 0 (inference, Monte-Carlo passes over a shared encoder): x holds nmod0 samples, sample n reads x[n % nmod0] - the
  // skip tensor of the shared levels is not replicated per pass
  const char* xn = reinterpret_cast<const char*>(x) + (int64_t)(nmod0 > 0 ? n % nmod0 : n) * D * xslice;
  // channel half 1: the next 32 channels of x, or a second tensor (torch.cat([x, x1], 1) never materialised)
  const char* xn1 = reinterpret_cast<const char*>(x1 ? x1 : x + 32) + (int64_t)n * D * xslice;
  auto slab_piece = [&](int s, int hf, int k) {             // piece k of channel part hf of slab s -> slot hf & 1
    if (wave + 4 * k < G::SLAB_DMA && soff[k] != -2) {
      const char* xs = (NQ == 2 ? (hf ? xn1 : xn) : xn + hf * 64) + s * xslice;        // uniform
      const void* src = soff[k] >= 0 ? (const void*)(xs + (unsigned)soff[k]) : (const void*)fplx_zero16;
      lds_dma(src, slabs + (hf & 1) * G::SLAB_BYTES + (wave + 4 * k) * 1024);
    }
  };
  // NQ = 4: 1-KiB piece j (0..53: 16 rows [tap j / 2][co (j & 1) * 16 ..] x 4 chunks) of weight quarter qt -> slot qt & 1
  constexpr int CIN = NQ * 32;
  const char* wlane = reinterpret_cast<const char*>(wp) + ((int64_t)(n0 + (lane >> 2)) * CIN) * 2 +
                      (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  auto w_piece = [&](int qt, int j) {
    const int64_t row0 = (int64_t)(j >> 1) * Cout + (j & 1) * 16;      // uniform
    lds_dma(wlane + row0 * (CIN * 2) + qt * 64, wbuf + (qt & 1) * G::WH_BYTES + j * 1024);
  };

  f32x16 K0a, K0b, K1a, K1b, K2a, K2b, Ra, Rb;
#pragma unroll
  for (int i = 0; i < 16; ++i) K0a[i] = K0b[i] = K1a[i] = K1b[i] = K2a[i] = K2b[i] = Ra[i] = Rb[i] = 0.f;

  // prologue: both halves of the first slab, the resident weights (source-side swizzle), bias
  const int sbase = TWOD ? d0 : d0 - 1;
  if (sbase >= 0) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) { slab_piece(sbase, 0, k); }
  }
  if constexpr (NQ == 2) {
    for (int j = wave; j < 2 * 27 * 32 * G::CH / 64; j += 4) {
      const int i = j * 64 + lane;                           // chunk index over [half][tap][co][4 chunks]
      const int hf = i / (27 * 32 * G::CH), ii = i % (27 * 32 * G::CH);
      const int row = ii >> 2, c = (ii & 3) ^ G::swz(row);
      lds_dma(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * 64 + hf * 32 + c * 8, wbuf + j * 1024);
    }
  } else {
    for (int j = wave; j < 54; j += 4) w_piece(0, j);
  }
  if (tid < 32) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
  dma_wait();
  block_sync();

  const float bv = bias_s[r];
  float ssum = 0.f, qsum = 0.f;
  char* stg = reinterpret_cast<char*>(bias_s + 32) + wave * G::STAGE_BYTES;
  char* stg_w = stg + (4 * khalf) * 64 + r * 2;
  const char* stg_r = stg + lane * 16;
  // accumulator element i of a lane is M-tile row mr = (i & 3) + 8 (i >> 2) + 4 khalf = voxel (mr / FW, mr % FW);
  // wmask[m] bit i: that voxel of M-tile m lies inside the volume (statistics only count real voxels)
  const int hb = h0 + wave * 2 * G::HPM;                     // first row of this wave's M-tile 0
  unsigned wmask0 = 0, wmask1 = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int mr = (i & 3) + 8 * (i >> 2) + 4 * khalf;
    if (w0 + mr % G::FW < W) {
      if (hb + mr / G::FW < H) wmask0 |= 1u << i;
      if (hb + G::HPM + mr / G::FW < H) wmask1 |= 1u << i;
    }
  }
  const unsigned ldy2 = (unsigned)ldy * 2u;
  char* yn = reinterpret_cast<char*>(y) + ((((int64_t)n * D * H + hb) * W + w0) * ldy + n0) * 2;
  const int64_t yslice = (int64_t)H * W * ldy * 2;
  // write-out: lane -> (staged row lane >> 2 and + 16, 16-byte chunk lane & 3).  FW32: both rows are voxels of one
  // volume row (w and w + 16); FW16: the second one is the same w of the NEXT volume row.
  const unsigned soffb = (unsigned)(lane >> 2) * ldy2 + (unsigned)(lane & 3) * 16u;
  const unsigned step1 = G::FW == 32 ? 16u : (unsigned)W;    // voxels between the two staged halves
  const bool wok = w0 + (lane >> 2) < W;
  bool sok[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    sok[m][0] = wok && hb + m * G::HPM < H;
    sok[m][1] = G::FW == 32 ? (sok[m][0] && w0 + (lane >> 2) + 16 < W) : (wok && hb + m * G::HPM + 1 < H);
  }
  const float slope_v = ACT ? *slope_p : 0.f;
  auto retire_elem = [&](f32x16& A, int m, int i) {
    const int wu = (i & 3) + 8 * (i >> 2);
    float ov = A[i] + bv;
    if (ACT) ov = ov > 0.f ? ov : ov * slope_v;
    *reinterpret_cast<bf16_t*>(stg_w + wu * 64) = (bf16_t)ov;
    if (!ACT && (((m ? wmask1 : wmask0) >> i) & 1u)) {
      ssum += ov;
      qsum = fmaf(ov, ov, qsum);
    }
  };
  auto retire_flush = [&](int m, int o) {
    if (sok[m][0] || sok[m][1]) {
      unsigned l2 = ldy2;
      asm volatile("" : "+s"(l2));
      char* rowp = yn + o * yslice + (unsigned)(m * G::HPM * W) * l2;
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(stg_r);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(stg_r + 1024);
      if (sok[m][0]) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(v0) : "memory");
      if (sok[m][1]) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + step1 * l2 + soffb), "v"(v1) : "memory");
    }
  };

  // half-step (t, hf): slab s = d0 - 1 + t, channel half hf, out of slot hf.  In its gaps: the DMA of the NEXT
  // half-slab (the other half of s, or half 0 of s + 1) into the other slot; during half 0 also the write-out of the
  // depth that completed in step t - 1 (R): stages 0-3 M-tile 0 -> LDS tile, flush, stages 4-7 M-tile 1, flush.
  const int nd = d1 - d0;                         // >= 2 (march_cfg)
  // (half 1 of the first slab is fetched by half-step (0, 0) like every other half-slab)
  const int nsteps = TWOD ? nd + 1 : nd + 2;
  for (int t = 0; t < nsteps; ++t) {
    const int s = sbase + t;
    const bool live = TWOD ? t < nd : (s >= 0 && s < D);   // a padding slab contributes nothing
    const bool wout = t >= (TWOD ? 2 : 3);
    const int o = s - 2;
#pragma unroll
    for (int hf = 0; hf < NQ; ++hf) {
      // next part-slab: (s, hf + 1) after (s, hf); (s + 1, 0) after the last part of s
      const int ns = hf < NQ - 1 ? s : s + 1, nh = (hf + 1) % NQ;
      const bool fetch = hf < NQ - 1 ? live : (TWOD ? ns < d1 : (ns >= 0 && ns < D && ns <= d1));
      const bool wfetch = NQ > 2 && !(t == nsteps - 1 && hf == NQ - 1);     // weight quarter nh for the next part-step
      auto side = [&](int q, int g) {
        if (g == 5 && q < NPIECE && fetch) slab_piece(ns, nh, q < NPIECE ? q : 0);
        if constexpr (NQ > 2) {                              // 14 (TWOD: 5) pieces per wave in the flush-free g = 4 gaps
          if (g == 4 && q != 3 && q != 7 && wfetch) {
            const int j = (TWOD ? 18 : 0) + wave + 4 * (q - (q > 3) - (q > 7));
            if (j < (TWOD ? 36 : 54)) w_piece(nh, j);
          }
        }
        if (hf == 0 && wout && q < 8) {
          if (g < 4) {
            if (q < 4) retire_elem(Ra, 0, 4 * q + g);
            else retire_elem(Rb, 1, 4 * (q - 4) + g);
          }
          if (g == 4 && q == 3) retire_flush(0, o);
          if (g == 4 && q == 7) retire_flush(1, o);
        }
      };
      const char* sl = slabs + (hf & 1) * G::SLAB_BYTES;
      const char* wh = wbuf + (hf & 1) * G::WH_BYTES;
#define M64_STEP(MASK)                                                                                              \
  do {                                                                                                              \
    if (hf == 0) march64_half<MASK, true, G>(sl, wh, wave, r, khalf, K0a, K0b, K1a, K1b, K2a, K2b, side);           \
    else march64_half<MASK, false, G>(sl, wh, wave, r, khalf, K0a, K0b, K1a, K1b, K2a, K2b, side);                  \
  } while (0)
      if (live) {
        if constexpr (TWOD) {
          M64_STEP(2);
        } else {
          if (t == 0) M64_STEP(1);
          else if (t == 1) M64_STEP(3);
          else if (t < nd) M64_STEP(7);
          else if (t == nd) M64_STEP(6);
          else M64_STEP(4);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int g = 0; g < 6; ++g) side(q, g);
        if (hf == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) K0a[i] = K0b[i] = 0.f;
        }
      }
#undef M64_STEP
      dma_wait();
      block_sync();
    }
    Ra = K2a; Rb = K2b; K2a = K1a; K2b = K1b; K1a = K0a; K1b = K0b;
  }
  // drain: the last depth completed in the final step
#pragma unroll
  for (int i = 0; i < 16; ++i) retire_elem(Ra, 0, i);
  retire_flush(0, d1 - 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) retire_elem(Rb, 1, i);
  retire_flush(1, d1 - 1);

  if (stats) {
    FPLX_LDS_BARRIER();
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][32]; the slabs are dead
    const float a = ssum + __shfl_xor(ssum, 32, 64), q2 = qsum + __shfl_xor(qsum, 32, 64);
    if (lane < 32) { red[(wave * 2 + 0) * 32 + r] = a; red[(wave * 2 + 1) * 32 + r] = q2; }
    FPLX_LDS_BARRIER();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float tt = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) tt += red[(wv * 2 + which) * 32 + c];
      stats[((int64_t)bid.x * 2 + which) * Cout + n0 + c] = tt;
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv_fwd_march64_lw (round 5): conv_fwd_march64 with DEDICATED LOADER WAVES, the form that took 8-20 % off the brick kernels
// (conv_brick.hip: conv_fwd_brick_lw).  conv_fwd_march64 issues every slab / weight piece from its four computing waves, in
// the MFMA gaps: a 64-bit source address and a zero-page select per piece (about 2300 vector instructions and 200 branches per
// depth step, DESIGN 7 (b) of round 4) and an issue stall of about 120 cycles per piece with the matrix pipe of that SIMD idle.
// Here a block is 8 waves, two per SIMD:
//   waves 0-3 (compute): fragment reads, MFMAs and the write-out of conv_fwd_march64, unchanged - no DMA, no source addresses;
//   waves 4-7 (loaders): the next part-slab (and weight quarter, NQ = 4) through a BUFFER DESCRIPTOR on the depth slice (a
//   lane's offset inside a slice is constant over the march; halo lanes carry an out-of-range offset and the hardware writes
//   zeros: no zero page, no select), s_waitcnt vmcnt(0), the part-step's barrier.
// The barrier count and order are conv_fwd_march64's: one per part-step; what it publishes (the part-slab fetched during the
// part-step) and what it frees (the slot just read) are unchanged.  Same MFMAs in the same order: bit-identical output.
template <class G, bool TWOD, int NQ, bool ACT = false>
__global__ void __launch_bounds__(512)
conv_fwd_march64_lw(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                    const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                    int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen,
                    const bf16_t* __restrict__ x1, int xcd, const float* __restrict__ slope_p = nullptr, int nmod0 = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabs = smem;                                        // [2 channel halves][SLAB][32]
  char* wbuf = smem + 2 * G::SLAB_BYTES;                     // [2 channel halves][27][32 co][32 ci]
  float* bias_s = reinterpret_cast<float*>(wbuf + 2 * G::WH_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;                            // uniform per wave
  const int wave = wave8 & 3;
  const int r = lane & 31, khalf = lane >> 5;
  const FplxBlock bid = fplx_xcd_block(xcd);
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int h0 = __builtin_amdgcn_readfirstlane(th * G::FH), w0 = __builtin_amdgcn_readfirstlane(tw * G::FW);
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = bid.y * 32;
  auto lds_dma = [&](const void* g, const char* l) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((const __attribute__((address_space(3))) char*)l));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int CIN = NQ * 32;
  const int sbase = TWOD ? d0 : d0 - 1;
  const int nd = d1 - d0;                         // >= 2 (march_cfg)
  const int nsteps = TWOD ? nd + 1 : nd + 2;
  // ---- prologue, all 8 waves: the resident weights (NQ = 2: both halves; NQ = 4: quarter 0), bias
  if constexpr (NQ == 2) {
    for (int j = wave8; j < 2 * 27 * 32 * G::CH / 64; j += 8) {
      const int i = j * 64 + lane;                           // chunk index over [half][tap][co][4 chunks]
      const int hf = i / (27 * 32 * G::CH), ii = i % (27 * 32 * G::CH);
      const int row = ii >> 2, c = (ii & 3) ^ G::swz(row);
      lds_dma(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * 64 + hf * 32 + c * 8, wbuf + j * 1024);
    }
  }
  const char* wlane = reinterpret_cast<const char*>(wp) + ((int64_t)(n0 + (lane >> 2)) * CIN) * 2 +
                      (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
  auto w_piece = [&](int qt, int j) {
    const int64_t row0 = (int64_t)(j >> 1) * Cout + (j & 1) * 16;      // uniform
    lds_dma(wlane + row0 * (CIN * 2) + qt * 64, wbuf + (qt & 1) * G::WH_BYTES + j * 1024);
  };
  if constexpr (NQ != 2) {
    for (int j = wave8; j < 54; j += 8) w_piece(0, j);
  }
  if (tid < 32) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;

  if (loader) {
    // ================================================================ loader waves
    constexpr int NPIECE = G::NPIECE;
    // per lane and piece: byte offset of the source chunk (channel part 0) inside ONE depth slice, 0x40000000 = outside the
    // volume (the descriptor's range check writes zeros), 0xFFFFFFFF = lane past the end of the slab (the last piece is 16 lanes)
    unsigned soff[NPIECE];
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int i = (wave + 4 * k) * 64 + lane;
      const int vox = i >> 2, c = (i & 3) ^ G::swz(vox);
      const int hh = vox / G::SW + h0 - 1, ww = vox % G::SW + w0 - 1;
      const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
      soff[k] = i >= G::SLAB_CHUNKS ? 0xFFFFFFFFu : (in ? (unsigned)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : 0x40000000u);
    }
    const int64_t xslice = (int64_t)H * W * ldx * 2;
    const char* xn = reinterpret_cast<const char*>(x) + (int64_t)(nmod0 > 0 ? n % nmod0 : n) * D * xslice;
    const char* xn1 = reinterpret_cast<const char*>(x1 ? x1 : x + 32) + (int64_t)n * D * xslice;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
    auto slab_part = [&](int s_, int hf) {                   // all of this wave's pieces of channel part hf of slab s_ -> slot hf & 1
      const char* xs = (NQ == 2 ? (hf ? xn1 : xn) : xn + hf * 64) + s_ * xslice;        // uniform
      u32x4 rx;
      rx[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xs);
      rx[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xs >> 32) & 0xFFFFu);
      rx[2] = __builtin_amdgcn_readfirstlane((unsigned)xslice);
      rx[3] = 0x00020000u;
#pragma unroll
      for (int k = 0; k < NPIECE; ++k)
        if (wave + 4 * k < G::SLAB_DMA && soff[k] != 0xFFFFFFFFu) {
          const unsigned dst = lds0 + (unsigned)((hf & 1) * G::SLAB_BYTES + (wave + 4 * k) * 1024);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(soff[k]), "s"(rx), "s"(0u), "s"(dst) : "memory");
        }
    };
    if (sbase >= 0) slab_part(sbase, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    block_sync();                                            // P0
    for (int t = 0; t < nsteps; ++t) {
      const int s = sbase + t;
      const bool live = TWOD ? t < nd : (s >= 0 && s < D);
#pragma unroll
      for (int hf = 0; hf < NQ; ++hf) {
        const int ns = hf < NQ - 1 ? s : s + 1, nh = (hf + 1) % NQ;
        const bool fetch = hf < NQ - 1 ? live : (TWOD ? ns < d1 : (ns >= 0 && ns < D && ns <= d1));
        const bool wfetch = NQ > 2 && !(t == nsteps - 1 && hf == NQ - 1);
        if (fetch) slab_part(ns, nh);
        if constexpr (NQ > 2) {
          if (wfetch)
            for (int j = (TWOD ? 18 : 0) + wave; j < (TWOD ? 36 : 54); j += 4) w_piece(nh, j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_sync();
      }
    }
    if (stats) { block_sync(); block_sync(); }               // the compute waves' statistics tail
    return;
  }

  // ================================================================ compute waves (conv_fwd_march64 without its DMA)
  f32x16 K0a, K0b, K1a, K1b, K2a, K2b, Ra, Rb;
#pragma unroll
  for (int i = 0; i < 16; ++i) K0a[i] = K0b[i] = K1a[i] = K1b[i] = K2a[i] = K2b[i] = Ra[i] = Rb[i] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of the weights
  block_sync();                                              // P0

  const float bv = bias_s[r];
  float ssum = 0.f, qsum = 0.f;
  char* stg = reinterpret_cast<char*>(bias_s + 32) + wave * G::STAGE_BYTES;
  char* stg_w = stg + (4 * khalf) * 64 + r * 2;
  const char* stg_r = stg + lane * 16;
  const int hb = h0 + wave * 2 * G::HPM;                     // first row of this wave's M-tile 0
  unsigned wmask0 = 0, wmask1 = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int mr = (i & 3) + 8 * (i >> 2) + 4 * khalf;
    if (w0 + mr % G::FW < W) {
      if (hb + mr / G::FW < H) wmask0 |= 1u << i;
      if (hb + G::HPM + mr / G::FW < H) wmask1 |= 1u << i;
    }
  }
  const unsigned ldy2 = (unsigned)ldy * 2u;
  char* yn = reinterpret_cast<char*>(y) + ((((int64_t)n * D * H + hb) * W + w0) * ldy + n0) * 2;
  const int64_t yslice = (int64_t)H * W * ldy * 2;
  const unsigned soffb = (unsigned)(lane >> 2) * ldy2 + (unsigned)(lane & 3) * 16u;
  const unsigned step1 = G::FW == 32 ? 16u : (unsigned)W;    // voxels between the two staged halves
  const bool wok = w0 + (lane >> 2) < W;
  bool sok[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    sok[m][0] = wok && hb + m * G::HPM < H;
    sok[m][1] = G::FW == 32 ? (sok[m][0] && w0 + (lane >> 2) + 16 < W) : (wok && hb + m * G::HPM + 1 < H);
  }
  const float slope_v = ACT ? *slope_p : 0.f;
  auto retire_elem = [&](f32x16& A, int m, int i) {
    const int wu = (i & 3) + 8 * (i >> 2);
    float ov = A[i] + bv;
    if (ACT) ov = ov > 0.f ? ov : ov * slope_v;
    *reinterpret_cast<bf16_t*>(stg_w + wu * 64) = (bf16_t)ov;
    if (!ACT && (((m ? wmask1 : wmask0) >> i) & 1u)) {
      ssum += ov;
      qsum = fmaf(ov, ov, qsum);
    }
  };
  auto retire_flush = [&](int m, int o) {
    if (sok[m][0] || sok[m][1]) {
      unsigned l2 = ldy2;
      asm volatile("" : "+s"(l2));
      char* rowp = yn + o * yslice + (unsigned)(m * G::HPM * W) * l2;
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(stg_r);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(stg_r + 1024);
      if (sok[m][0]) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + soffb), "v"(v0) : "memory");
      if (sok[m][1]) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(rowp + step1 * l2 + soffb), "v"(v1) : "memory");
    }
  };
  for (int t = 0; t < nsteps; ++t) {
    const int s = sbase + t;
    const bool live = TWOD ? t < nd : (s >= 0 && s < D);   // a padding slab contributes nothing
    const bool wout = t >= (TWOD ? 2 : 3);
    const int o = s - 2;
#pragma unroll
    for (int hf = 0; hf < NQ; ++hf) {
      auto side = [&](int q, int g) {
        if (hf == 0 && wout && q < 8) {
          if (g < 4) {
            if (q < 4) retire_elem(Ra, 0, 4 * q + g);
            else retire_elem(Rb, 1, 4 * (q - 4) + g);
          }
          if (g == 4 && q == 3) retire_flush(0, o);
          if (g == 4 && q == 7) retire_flush(1, o);
        }
      };
      const char* sl = slabs + (hf & 1) * G::SLAB_BYTES;
      const char* wh = wbuf + (hf & 1) * G::WH_BYTES;
#define M64_STEP(MASK)                                                                                              \
  do {                                                                                                              \
    if (hf == 0) march64_half<MASK, true, G>(sl, wh, wave, r, khalf, K0a, K0b, K1a, K1b, K2a, K2b, side);           \
    else march64_half<MASK, false, G>(sl, wh, wave, r, khalf, K0a, K0b, K1a, K1b, K2a, K2b, side);                  \
  } while (0)
      if (live) {
        if constexpr (TWOD) {
          M64_STEP(2);
        } else {
          if (t == 0) M64_STEP(1);
          else if (t == 1) M64_STEP(3);
          else if (t < nd) M64_STEP(7);
          else if (t == nd) M64_STEP(6);
          else M64_STEP(4);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int g = 0; g < 6; ++g) side(q, g);
        if (hf == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) K0a[i] = K0b[i] = 0.f;
        }
      }
#undef M64_STEP
      block_sync();
    }
    Ra = K2a; Rb = K2b; K2a = K1a; K2b = K1b; K1a = K0a; K1b = K0b;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) retire_elem(Ra, 0, i);
  retire_flush(0, d1 - 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) retire_elem(Rb, 1, i);
  retire_flush(1, d1 - 1);

  if (stats) {
    FPLX_LDS_BARRIER();
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][32]; the slabs are dead
    const float a = ssum + __shfl_xor(ssum, 32, 64), q2 = qsum + __shfl_xor(qsum, 32, 64);
    if (lane < 32) { red[(wave * 2 + 0) * 32 + r] = a; red[(wave * 2 + 1) * 32 + r] = q2; }
    FPLX_LDS_BARRIER();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float tt = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) tt += red[(wv * 2 + which) * 32 + c];
      stats[((int64_t)bid.x * 2 + which) * Cout + n0 + c] = tt;
    }
  }
}

struct MarchCfg { int tilesH, tilesW, dsegs, dlen, nblk, fw; };

inline int march_enabled() {
  return (int)fplx_knob(FPLX_K_MARCH);                    // tuning knob (benchmarks only): 0 = previous kernel
}

inline MarchCfg march_cfg(int n, int d, int h, int w, int cin, int cout) {
  MarchCfg c;
  int fh = cin >= 64 ? MG64::FH : MG::FH;
  c.fw = 32;
  if (cin >= 64) {                                           // 16 x 16 footprint when it wastes less area than 8 x 32
    const int kfw = (int)fplx_knob(FPLX_K_MARCH64_FW);
    const int64_t a32 = (int64_t)((h + 7) / 8) * 8 * ((w + 31) / 32) * 32;
    const int64_t a16 = (int64_t)((h + 15) / 16) * 16 * ((w + 15) / 16) * 16;
    if ((kfw == 0 && a16 < a32) || kfw == 16) { c.fw = 16; fh = 16; }
  }
  c.tilesH = (h + fh - 1) / fh;
  c.tilesW = (w + c.fw - 1) / c.fw;
  const int64_t tiles = (int64_t)n * c.tilesH * c.tilesW * (cout / 32);
  // one block per CU at a time: choose the depth split that minimises rounds x (slabs per block + prologue)
  double best = 1e30;
  int best_ds = 1;
  for (int ds = 1; ds <= d; ++ds) {
    const int dl = (d + ds - 1) / ds;
    if (dl < 4 && ds > 1) break;
    const int segs = (d + dl - 1) / dl;
    if (d - (segs - 1) * dl < 2 && d >= 2) continue;       // every segment needs two depths (step masks 1, 3 ... 6, 4)
    const int64_t rounds = (tiles * segs + 255) / 256;
    const double cost = (double)rounds * (dl + 2 + 1.5);
    if (cost < best - 1e-9) { best = cost; best_ds = segs; }
  }
  {
    const int e = (int)fplx_knob(FPLX_K_MARCH_DS);           // tuning knob (benchmarks only)
    if (e > 0) best_ds = e;
  }
  c.dlen = (d + best_ds - 1) / best_ds;
  c.dsegs = (d + c.dlen - 1) / c.dlen;
  c.nblk = n * c.tilesH * c.tilesW * c.dsegs;
  return c;
}

}  // namespace

extern "C" int fplx_march_ok(int n, int d, int h, int w, int cin, int cout) {
  const int en = march_enabled();                            // 1: both kernels, 2: Cin = 32 only
  if (!en || cout % 32 != 0 || d < 4 || w < 64) return 0;
  if (cin == 32) return h >= 16;
  if (cin == 64) return en == 1 && h >= 8 && (int64_t)h * w * 64 * 2 < (int64_t)1 << 31;
  if (cin == 128) {                                          // streamed-weight form of the Cin = 64 kernel
    const int k128 = (int)fplx_knob(FPLX_K_MARCH128);        // A/B knob
    return k128 && en == 1 && h >= 8 && (int64_t)h * w * 128 * 2 < (int64_t)1 << 31;
  }
  return 0;
}

// which depth march a layer takes (fplx_conv3d_plan_query's geometry for FPLX_KERNEL_MARCH): Cin = 32: 2 = the one-wave-per-SIMD
// kernels for footprints inside the volume (conv_fwd_march32v2 with statistics, conv_fwd_march32v3 without, for operands
// with ld = channels), 0 = the 8-wave conv_fwd_march32; Cin >= 64: the footprint width of conv_fwd_march64 (16 | 32)
extern "C" int fplx_march_variant(int n, int d, int h, int w, int cin, int cout) {
  if (cin == 32)
    return (fplx_knob(FPLX_K_MARCH32_V2) && h % MG2::FH == 0 && w % MG2::FW == 0 && (int64_t)d * h * w * cin * 2 <= ((int64_t)1 << 30)) ? 2 : 0;
  return march_cfg(n, d, h, w, cin, cout).fw;
}

extern "C" int fplx_march_rows(int n, int d, int h, int w, int cin, int cout) {
  return march_cfg(n, d, h, w, cin, cout).nblk;
}

// returns 1 if launched, 0 if the pointers do not allow the vector stores, <0 on error
// twod: the pack is a Conv2d in the middle depth plane (fplx_pack_conv2d_weight) - the Cin = 32 march then runs its
// kd = 1 taps only (same result, a third of the MFMAs)
// slope != NULL (inference): PReLU in the write-out, no statistics (stats must be NULL)
extern "C" int fplx_march_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                         int n, int d, int h, int w, int cin, int cout, float* stats, hipStream_t st,
                                         const void* x1, void* y1, int twod, const float* slope, int nmod0) {
  if (ldy % 8 != 0 || ((uintptr_t)y % 16) != 0 || ldx % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)wp % 16) != 0 ||
      ((uintptr_t)x1 % 16) != 0 || ((uintptr_t)y1 % 16) != 0)
    return 0;
  const MarchCfg c = march_cfg(n, d, h, w, cin, cout);
  dim3 grid(c.nblk, cout / 32);
  if (cin >= 64) {
    if (y1 || (cin == 128 && x1)) return 0;
#define LAUNCH_M64Q(G_, TWOD_, NQ_)                                                                                 \
  do {                                                                                                              \
    if (slope) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)conv_fwd_march64<G_, TWOD_, NQ_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
      conv_fwd_march64<G_, TWOD_, NQ_, true><<<grid, G_::THREADS, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, \
                                                                         (bf16_t*)y, ldy, n, d, h, w, cout, nullptr, c.tilesH, \
                                                                         c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on(), slope, nmod0); \
    } else {                                                                                                        \
    (void)hipFuncSetAttribute((const void*)conv_fwd_march64<G_, TWOD_, NQ_>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
    conv_fwd_march64<G_, TWOD_, NQ_><<<grid, G_::THREADS, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, \
                                                                         (bf16_t*)y, ldy, n, d, h, w, cout, stats, c.tilesH, \
                                                                         c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on()); \
    }                                                                                                               \
  } while (0)
    // loader-wave form (knob march64_lw, default on): the Conv2d-per-slice forms only (TWOD: one live accumulator role, 135-200
    // registers: -21..-28 % per launch, profiles/r05_kernel_ab.txt).  Needs a depth slice below 1 GiB (32-bit buffer offsets, the
    // out-of-range marker).  The 3D forms stay on conv_fwd_march64: their eight accumulator tiles + the rotation's copies do not
    // fit 256 registers (57-78 spilled registers inside the MFMA loop: +23 %, profiles/r05_kernel_ab.txt section 4) - that
    // instantiation was removed in round 6
    const bool lw64 = twod && fplx_knob(FPLX_K_MARCH64_LW) != 0 && (int64_t)h * w * ldx * 2 < ((int64_t)1 << 30);
#define LAUNCH_M64LW(G_, NQ_)                                                                                       \
  do {                                                                                                              \
    if (slope) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)conv_fwd_march64_lw<G_, true, NQ_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
      conv_fwd_march64_lw<G_, true, NQ_, true><<<grid, 512, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, \
                                                                         (bf16_t*)y, ldy, n, d, h, w, cout, nullptr, c.tilesH, \
                                                                         c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on(), slope, nmod0); \
    } else {                                                                                                        \
    (void)hipFuncSetAttribute((const void*)conv_fwd_march64_lw<G_, true, NQ_>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
    conv_fwd_march64_lw<G_, true, NQ_><<<grid, 512, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, \
                                                                         (bf16_t*)y, ldy, n, d, h, w, cout, stats, c.tilesH, \
                                                                         c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on()); \
    }                                                                                                               \
  } while (0)
#define LAUNCH_M64(G_, TWOD_)                                                                                       \
  do { if (cin == 128) LAUNCH_M64Q(G_, TWOD_, 4); else LAUNCH_M64Q(G_, TWOD_, 2); } while (0)
#define LAUNCH_M64_2D(G_)                                                                                           \
  do {                                                                                                              \
    if (lw64) { if (cin == 128) LAUNCH_M64LW(G_, 4); else LAUNCH_M64LW(G_, 2); }                                    \
    else LAUNCH_M64(G_, true);                                                                                      \
  } while (0)
    using G16 = MG64T<16>;
    if (c.fw == 16) { if (twod) LAUNCH_M64_2D(G16); else LAUNCH_M64(G16, false); }
    else { if (twod) LAUNCH_M64_2D(MG64); else LAUNCH_M64(MG64, false); }
#undef LAUNCH_M64_2D
#undef LAUNCH_M64
#undef LAUNCH_M64LW
#undef LAUNCH_M64Q
    const int rc64 = fplx_check_launch("march64_conv3d_fwd");
    return rc64 < 0 ? rc64 : 1;
  }
  if (x1) return 0;
  {
    const int kv2 = (int)fplx_knob(FPLX_K_MARCH32_V2);       // A/B knob
    if (kv2 && !twod && h % MG2::FH == 0 && w % MG2::FW == 0 && (int64_t)d * h * w * ldx * 2 <= ((int64_t)1 << 30)) {
#define LAUNCH_M32V2(STATS_)                                                                                        \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)conv_fwd_march32v2<STATS_>, hipFuncAttributeMaxDynamicSharedMemorySize, MG2::LDS); \
    conv_fwd_march32v2<STATS_><<<grid, MG2::THREADS, MG2::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, \
                                                                ldy, n, d, h, w, cout, stats, c.tilesH, c.tilesW, c.dsegs, \
                                                                c.dlen, (bf16_t*)y1, y1 ? cout / 64 : cout / 32, fplx_xcd_on()); \
  } while (0)
#define LAUNCH_M32V3X(STATS_, ASWZ_)                                                                                      \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)conv_fwd_march32v3<STATS_, ASWZ_>, hipFuncAttributeMaxDynamicSharedMemorySize, MG2::LDS); \
    conv_fwd_march32v3<STATS_, ASWZ_><<<grid, MG2::THREADS, MG2::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, \
                                                                ldy, n, d, h, w, cout, stats, c.tilesH, c.tilesW, c.dsegs, \
                                                                c.dlen, (bf16_t*)y1, y1 ? cout / 64 : cout / 32, fplx_xcd_on()); \
  } while (0)
#define LAUNCH_M32V3(STATS_) LAUNCH_M32V3X(STATS_, 1)      /* ASWZ = 0: the 32 x 32 x 16 kernels' swizzle (A/B builds) */
      // 1: v2 everywhere; 3: v3 everywhere; 4: v3 where no statistics are wanted (its STATS form spills), v2 otherwise
      if (slope) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_march32v3<false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MG2::LDS);
        conv_fwd_march32v3<false, 1, true><<<grid, MG2::THREADS, MG2::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y,
                                                                ldy, n, d, h, w, cout, nullptr, c.tilesH, c.tilesW, c.dsegs,
                                                                c.dlen, (bf16_t*)y1, y1 ? cout / 64 : cout / 32, fplx_xcd_on(), slope);
      }
      else if (stats) { if (kv2 == 3) LAUNCH_M32V3(true); else LAUNCH_M32V2(true); }
      else { if (kv2 >= 3) LAUNCH_M32V3(false); else LAUNCH_M32V2(false); }
#undef LAUNCH_M32V3
#undef LAUNCH_M32V3X
#undef LAUNCH_M32V2
      const int rc2 = fplx_check_launch("march32v2_conv3d_fwd");
      return rc2 < 0 ? rc2 : 1;
    }
  }
#define LAUNCH_M32(TWOD_)                                                                                           \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)conv_fwd_march32<TWOD_>, hipFuncAttributeMaxDynamicSharedMemorySize, MG::LDS); \
    conv_fwd_march32<TWOD_><<<grid, MG::THREADS, MG::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, \
                                                                ldy, n, d, h, w, cout, stats, c.tilesH, c.tilesW, c.dsegs, \
                                                                c.dlen, (bf16_t*)y1, y1 ? cout / 64 : cout / 32, fplx_xcd_on()); \
  } while (0)
#define LAUNCH_M32A(TWOD_)                                                                                          \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)conv_fwd_march32<TWOD_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MG::LDS); \
    conv_fwd_march32<TWOD_, true><<<grid, MG::THREADS, MG::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, \
                                                                ldy, n, d, h, w, cout, nullptr, c.tilesH, c.tilesW, c.dsegs, \
                                                                c.dlen, (bf16_t*)y1, y1 ? cout / 64 : cout / 32, fplx_xcd_on(), slope); \
  } while (0)
  if (slope) { if (twod) LAUNCH_M32A(true); else LAUNCH_M32A(false); }
  else if (twod) LAUNCH_M32(true); else LAUNCH_M32(false);
#undef LAUNCH_M32A
#undef LAUNCH_M32
  const int rc = fplx_check_launch("march_conv3d_fwd");
  return rc < 0 ? rc : 1;
}

extern "C" int fplx_march_conv3d_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                     int n, int d, int h, int w, int cin, int cout, float* stats, hipStream_t st,
                                     const void* x1, void* y1, int twod) {
  return fplx_march_conv3d_fwd_act(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, st, x1, y1, twod, nullptr, 0);
}
