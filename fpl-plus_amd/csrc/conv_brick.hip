// conv_fwd_brick_lw: 3x3x3 convolution (forward and, with the mirrored pack, data gradient) for the levels whose planes are too
// small for the depth march (W < 64) and whose channel counts are too large for resident weights: Cin % 32 == 0,
// Cout % 128 == 0 - levels 2.. of the 32-base network (128 / 256 channels on 20 x 40 x 40 at the benchmark shape).
// (reference op: PyMIC/pymic/net/net3d/unet2d5_dsbn.py:66-81 ConvBlockND's nn.Conv3d(k=3, padding=1))
//
// What it replaces there is conv_fwd_tile (conv_mfma.hip), an LDS-tiled implicit GEMM that re-stages the shifted voxel rows
// for every one of the 27 taps: 32 KB of L2 -> registers -> LDS traffic per 64 MFMAs, 1.5 KB of LDS traffic per MFMA - it
// runs at 0.27-0.31 of the MFMA peak, bound by the fill path.  Here the INPUT is stationary:
//   * a block owns a brick of 4 x 8 x 8 output voxels x 128 output channels; the brick's halo (6 x 10 x 10 voxels) of one
//     32-channel chunk is staged ONCE by LDS-DMA through a buffer descriptor (padding = the hardware's out-of-range zeros,
//     see conv_fwd_march32v2) and serves all 27 taps: 46 KB per 1728 MFMAs instead of 16 KB per 64;
//   * the weights stream through a two-slot ring, one stage = the three depth taps of one (kh, kw) for the chunk (24 KB,
//     L2 hits: every block reads the same pack), DMA'd a stage ahead;
//   * ONE wave per SIMD, a wave = 4 depth planes x (4 x 8 voxels) x 64 channels: an A fragment (input plane q) feeds the
//     up to three output planes q - kd, so a half-stage (16 input channels) reads 6 A + 6 B fragments for 24 MFMAs
//     (0.5 ds_read_b128 per MFMA); fragments are prefetched a half-stage ahead, one barrier per stage (between its halves);
//   * LDS image: voxel rows of 64 B, voxel index L = plane * 128 + row * 12 + col, 16-byte chunks XOR-swizzled with
//     (L >> 2) & 3; the lane -> voxel map of an M-tile follows ds_read_b128's lane groups so that each group touches
//     16 distinct 16-byte slots for every tap shift (rows 0, 2 in one group, rows 1, 3 in the other).
#include "common.h"
#include <type_traits>

#ifdef FPLX_STAMP
__device__ long long* fplx_brick_stamp_buf;      // set by tools/micro/brick_bench.hip: 8 counters per wave
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct BK {
  static constexpr int TW = 8, KC = 32, ROWB = 64, SWP = 12, THREADS = 256;
};
// A block = TD depth planes x (4 WH) rows x 8 columns of output voxels x NT output channels.  Its four waves are WH row
// groups (a wave's patch is 4 x 8 voxels per plane) x WN = 4 / WH channel groups of NTW N-tiles (32 channels) each:
//   <4, 2, 2>: 4 x 8 x 8 x 128   <4, 2, 1>: 4 x 8 x 8 x 64   <5, 1, 1>: 5 x 4 x 8 x 128 (level 3: 10 x 20 x 20 without
//   padding in d and h)
template <int TD_, int WH_, int NTW_>
struct BKG : BK {
  static constexpr int TD = TD_, WH = WH_, NTW = NTW_, WN = 4 / WH_, TH = 4 * WH_;
  static constexpr int NT = WN * 32 * NTW;
  static constexpr int SD = TD + 2, SH = TH + 2;
  static constexpr int PL = (SH * SWP + 15) / 16 * 16;        // voxel slots per plane of the LDS image: 128 / 80
  static constexpr int BRICK_BYTES = SD * PL * ROWB;          // 49152 / 35840
  static constexpr int NP_TOT = BRICK_BYTES / 1024;           // DMA wave-instructions per brick chunk
  static constexpr int NPB = (NP_TOT + 3) / 4;                // ... per wave (12 / 9)
  static constexpr int WST_BYTES = 3 * NT * ROWB;             // one weight stage: 24576 / 12288
  static constexpr int NPW = WST_BYTES / 1024 / 4;            // 6 / 3
  // weight ring: THREE slots where they fit the 160 KB beside the two brick slots (<4,2,1>: 132 KB, <5,1,1>: 144 KB; <4,2,2>
  // would need 168 KB and keeps two).  With three, the slot of stage s + 2 is free during ALL of stage s, so its DMA pieces are
  // spread over both halves of the stage, one every few MFMAs; with two, only behind the stage's barrier (see the stage loop)
  static constexpr int NWS = (2 * BRICK_BYTES + 3 * WST_BYTES + NT * 4 <= 160 * 1024) ? 3 : 2;
  static constexpr int LDS = 2 * BRICK_BYTES + NWS * WST_BYTES + NT * 4;
  static constexpr int M1 = 3 * TD * NTW;                     // MFMAs per half-stage and wave
};

// M-tile row m (0..31, = lane & 31 of an A fragment) -> (row 0..3, col 0..7) of the wave's 4 x 8 patch
__device__ __forceinline__ int bk_row(int m) { return 2 * (m >> 4) + (((m >> 4) ^ (m >> 3) ^ (m >> 2)) & 1); }
__device__ __forceinline__ int bk_col(int m) { return ((m >> 3) & 1) * 4 + (m & 3); }

// ------------------------------------------------------------------------------------------
// conv_fwd_brick_lw (round 5): the same bricks, rings and LDS images with DEDICATED LOADER WAVES.  Cycle stamps of round 4
// (profiles/r04_brick_stamps_after.txt): a stage's MFMAs are 768-960 cycles of issue, the stage takes 1135-1798 - every LDS-DMA
// piece holds the wave that issues it for about 120 cycles wherever it is placed, and an in-order wave that waits for its
// memory path issues no MFMAs.  The stall belongs to the WAVE, not to the SIMD: here a block is 8 waves, two per SIMD -
//   waves 0-3 (compute): fragment reads + MFMAs + the write-out, no vector-memory instruction in the stage loop at all;
//   waves 4-7 (loaders): nothing but the stage's DMA pieces (weights of stage s + 2 - s + 1 with two slots - and two pieces of
//   the next chunk of the brick), a counted s_waitcnt and the stage's barrier: their issue stalls overlap the partner wave's
//   MFMAs, the matrix pipe of every SIMD sees one uninterrupted MFMA stream.
// One s_barrier per stage for all 8 waves, at the compute waves' mid-stage point.  What the barrier of stage s orders:
//   * the loaders' counted wait in front of it has retired every piece issued before this iteration (three weight slots) /
//     every piece (two): stage s + 1's weights and, at the chunk's last stage, the next chunk of the brick have landed -
//     the compute waves read them only in the second half of stage s, behind the barrier;
//   * every fragment read of weight slot s happens before it (first-half fragments in stage s - 1's second half, second-half
//     fragments in stage s's first half): behind it the loaders may overwrite that slot (stage s + 3 / s + 2);
//   * the write-out's staging tiles live in the dead brick slot, which is also where the NEXT brick's second chunk goes: the
//     loaders pass one more barrier (behind the compute waves' write-out) before they issue brick pieces in a brick's first stage.
// Registers: 2 waves per SIMD = 256 per lane; the accumulators (64-128) and the two fragment sets fit.
// ACT (inference, eval-mode BatchNorm folded into the pack): PReLU(slope) in the write-out (STATS must be false)
// CAT2 (inference forms only): the input is the channel concatenation of TWO tensors of Cin / 2 channels and one leading
// dimension - chunks below Cin / 2 come from x, sample n % nmod0 (nmod0 > 0: the skip tensor that the Monte-Carlo passes
// share, one copy for all of them), the rest from x1
// (The round-2 kernel without loader waves, conv_fwd_brick, computed the same bits; it was removed in round 6 - git history.)
template <bool STATS, int NTW, int TD, int WH, bool ACT = false, bool CAT2 = false>
__global__ void __launch_bounds__(512)
conv_fwd_brick_lw(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
                  bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int Cin, int Cout,
                  float* __restrict__ stats, float* __restrict__ partial, int bD, int bH, int bW, int xcd, const float* __restrict__ slope_p = nullptr,
                  const bf16_t* __restrict__ x1 = nullptr, int nmod0 = 0) {
  using G = BKG<TD, WH, NTW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bricks = smem;
  char* wring = smem + 2 * G::BRICK_BYTES;
  float* bias_s = reinterpret_cast<float*>(wring + G::NWS * G::WST_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;                              // uniform per wave
  const int wave = wave8 & 3;                                  // index inside the role
  const int r = lane & 31, khalf = lane >> 5;
  const int hhalf = wave / G::WN, wn = wave % G::WN;
  // Which (output-channel tile, Cin split) = weight slice and which part of the brick list this block takes.  The dispatcher
  // deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own 4-MB L2 (common.h):
  //   xcd == 1: XCD j sweeps its contiguous eighth of the brick list for every slice (fplx_xcd_tiles) - neighbouring bricks,
  //             shared halos, one L2: the levels whose weights fit an L2 beside the activations (levels 1-2);
  //   xcd == 2: (the launcher made grid.x * U a multiple of 8, U = grid.y * grid.z slices) every XCD works on ONE slice (U
  //             divides 8) or on U / 8 of them: the slice - 27 x NT x Cin / grid.z weights, streamed once per brick - stays
  //             in that XCD's L2.  The deep levels' packs are 3.5-14 MB: dealt the other way every L2 streams all of it for
  //             every brick (level 3, 256 -> 512: 122 -> 75 us).
  int by = blockIdx.y, bz = blockIdx.z;
  FplxTileRange tr = fplx_xcd_tiles((int64_t)N * bD * bH * bW, xcd == 1);
  if (xcd == 2) {
    const unsigned gx = gridDim.x, U = gridDim.y * gridDim.z;
    const unsigned L = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x, xc = L & 7u, idx = L >> 3;
    unsigned u, stripe;
    if (8 % U == 0) { u = xc % U; stripe = idx * (8 / U) + xc / U; }
    else { u = xc + 8 * (idx % (U / 8)); stripe = idx / (U / 8); }
    by = __builtin_amdgcn_readfirstlane((int)(u % gridDim.y));
    bz = __builtin_amdgcn_readfirstlane((int)(u / gridDim.y));
    tr.first = __builtin_amdgcn_readfirstlane((int)stripe);
    tr.step = gx;
  }
  const int n0 = by * G::NT;
  if (tr.first >= tr.end) return;
  const int c_lo = (int)((int64_t)(Cin / G::KC) * bz / gridDim.z);
  const int nch = (int)((int64_t)(Cin / G::KC) * (bz + 1) / gridDim.z) - c_lo;
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  struct BrickPos { int n, d0, h0, w0; };
  auto position = [&](int64_t tile, BrickPos& B) {
    int b = (int)tile;
    const int bw = b % bW; b /= bW;
    const int bh = b % bH; b /= bH;
    const int bd = b % bD; b /= bD;
    B.n = __builtin_amdgcn_readfirstlane(b);
    B.d0 = __builtin_amdgcn_readfirstlane(bd * G::TD);
    B.h0 = __builtin_amdgcn_readfirstlane(bh * G::TH);
    B.w0 = __builtin_amdgcn_readfirstlane(bw * G::TW);
  };

  if (loader) {
    // ================================================================ loader waves
    const int64_t xsample = (int64_t)D * H * W * ldx * 2;
    u32x4 rw;
    rw[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)wp);
    rw[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)wp >> 32) & 0xFFFFu);
    rw[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)27 * Cout * Cin * 2));
    rw[3] = 0x00020000u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
    auto buf_dma = [&](const u32x4& rsrc, unsigned vo, unsigned so, unsigned dst_off) {
      const unsigned dst = lds0 + dst_off;
      const unsigned so_ = __builtin_amdgcn_readfirstlane(so);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so_), "s"(dst) : "memory");
    };
    // A brick's DMA lane offsets are NOT tabulated per brick (the round-2 kernel's setup(): 12 pieces x 64-bit address arithmetic,
    // about 5 K cycles per brick in front of its first stage - 18 % of a level-1 launch by the stamps): a lane's offset is
    // brick base + a brick-independent delta, only its validity depends on the brick - the loaders have idle vector slots and
    // form it when they issue the piece.  dlt[k]: byte offset of the lane's 16 bytes of piece k relative to the voxel
    // (d0 - 1, h0 - 1, w0 - 1); pos[k]: (q, hh, ww) of that voxel inside the halo, packed, or bit 31 for lanes past the image
    struct Brick { int n, n0, d0, h0, w0; unsigned base; };
    unsigned dlt[G::NPB], pos[G::NPB];
#pragma unroll
    for (int k = 0; k < G::NPB; ++k) {
      const int pidx = wave + 4 * k < G::NP_TOT ? wave + 4 * k : G::NP_TOT - 1;
      const int ci = pidx * 64 + lane;
      const int L = ci >> 2, cc = (ci & 3) ^ ((L >> 2) & 3);
      const int q = L / G::PL, rem = L % G::PL, hh = rem / G::SWP, ww = rem % G::SWP;
      const bool img = q < G::SD && rem < G::SH * G::SWP && ww < 10;
      dlt[k] = (unsigned)(((((int64_t)q * H + hh) * W + ww) * ldx + cc * 8) * 2);
      pos[k] = img ? (unsigned)(q | (hh << 8) | (ww << 16)) : 0x80000000u;
    }
    auto setup = [&](int64_t tile, Brick& B) {
      BrickPos P;
      position(tile, P);
      B.n = P.n;
      B.n0 = (CAT2 && nmod0 > 0) ? __builtin_amdgcn_readfirstlane(P.n % nmod0) : P.n;
      B.d0 = P.d0; B.h0 = P.h0; B.w0 = P.w0;
      // (the base may be "negative": it only ever meets a delta that brings a valid lane's sum into the sample)
      B.base = __builtin_amdgcn_readfirstlane((unsigned)(((((int64_t)(P.d0 - 1) * H + (P.h0 - 1)) * W + (P.w0 - 1)) * ldx) * 2));
    };
    auto lane_off = [&](const Brick& B, int k) -> unsigned {
      const unsigned pk = pos[k];
      const int gd = B.d0 - 1 + (int)(pk & 0xFFu), gh = B.h0 - 1 + (int)((pk >> 8) & 0xFFu), gw = B.w0 - 1 + (int)((pk >> 16) & 0xFFu);
      const bool in = (int)pk >= 0 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      return in ? B.base + dlt[k] : 0x40000000u;
    };
    const unsigned wvo = (unsigned)((((int64_t)(n0 + (lane >> 2))) * Cin + ((lane & 3) ^ ((lane >> 4) & 3)) * 8) * 2);
    const unsigned tapstride = (unsigned)((int64_t)Cout * Cin * 2);
    struct BrickSrc { u32x4 rx; unsigned so; };
    auto brick_rsrc = [&](const Brick& B, int ch, bool on) {
      int c = c_lo + ch;
      const char* xn = reinterpret_cast<const char*>(x) + (int64_t)B.n * xsample;
      if (CAT2) {
        const int half = Cin / (2 * G::KC);
        if (c >= half) { xn = reinterpret_cast<const char*>(x1) + (int64_t)B.n * xsample; c -= half; }
        else xn = reinterpret_cast<const char*>(x) + (int64_t)B.n0 * xsample;
      }
      BrickSrc rs;
      rs.rx[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xn);
      rs.rx[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xn >> 32) & 0xFFFFu);
      rs.rx[2] = __builtin_amdgcn_readfirstlane((unsigned)xsample);
      rs.rx[3] = 0x00020000u;
      rs.so = __builtin_amdgcn_readfirstlane(on ? (unsigned)(c * G::KC * 2) : 0x40000000u);
      return rs;
    };
    auto brick_piece = [&](const BrickSrc& src, const Brick& B, int slot, int k) {
      const int pidx = wave + 4 * k < G::NP_TOT ? wave + 4 * k : G::NP_TOT - 1;
      buf_dma(src.rx, lane_off(B, k), src.so, (unsigned)(slot * G::BRICK_BYTES + pidx * 1024));
    };
    auto weight_piece = [&](int ch, int t9, int slot, int k, bool on) {
      const int j = wave + 4 * k, kd = j / (G::NT / 16);
      const unsigned so = (unsigned)(kd * 9 + t9) * tapstride + (unsigned)(((j % (G::NT / 16)) * 16 * Cin + (c_lo + ch) * G::KC) * 2);
      buf_dma(rw, wvo, on ? so : 0x40000000u, (unsigned)(2 * G::BRICK_BYTES + slot * G::WST_BYTES + j * 1024));
    };
    Brick cur, nxt;
    int64_t tile = tr.first;
    setup(tile, cur);
    nxt = cur;
    // prologue: brick chunk 0 -> brick slot 0, weight stage 0 (and 1 with three slots) -> slots 0 (, 1)
    {
      const BrickSrc s0 = brick_rsrc(cur, 0, true);
#pragma unroll
      for (int k = 0; k < G::NPB; ++k) brick_piece(s0, cur, 0, k);
#pragma unroll
      for (int k = 0; k < G::NPW; ++k) weight_piece(0, 0, 0, k, true);
      if (G::NWS == 3) {
#pragma unroll
        for (int k = 0; k < G::NPW; ++k) weight_piece(0, 1, 1, k, true);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    block_sync();                                              // P0
    int cc = 0, ws = 0;
    bool first_brick = true;
#ifdef FPLX_STAMP
    long long st_h1 = 0, st_wait = 0, st_bar = 0, st_n = 0;
    const long long st_begin = __builtin_amdgcn_s_memtime();
    const long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#define STAMP(var_) do { const long long t__ = __builtin_amdgcn_s_memtime(); var_ += t__ - st_t; st_t = t__; } while (0)
#else
#define STAMP(var_) do { } while (0)
#endif
    for (;;) {
      const int64_t tile_nx = tile + tr.step;
      const bool has_next = tile_nx < tr.end;
      if (has_next) setup(tile_nx, nxt);
      for (int ch = 0; ch < nch; ++ch, ++cc) {
        const bool last_ch = ch + 1 == nch;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
#ifdef FPLX_STAMP
          long long st_t = __builtin_amdgcn_s_memtime();
          ++st_n;
#endif
          const int s1 = ws + 1 == G::NWS ? 0 : ws + 1, s2 = s1 + 1 == G::NWS ? 0 : s1 + 1;
          // weights of stage gs + 2 (three slots: its slot was last read in front of the previous stage's barrier) or gs + 1 (two)
          constexpr int AH = G::NWS == 3 ? 2 : 1;
          int cha = ch, t9a = t9 + AH;
          if (t9a >= 9) { t9a -= 9; ++cha; }
          const bool w_on = cha < nch || has_next;             // uniform
          if (cha >= nch) cha = 0;
          const int wslot = G::NWS == 3 ? s2 : s1;
#pragma unroll
          for (int k = 0; k < G::NPW; ++k) weight_piece(cha, t9a, wslot, k, w_on);
          // the brick slot of the next chunk doubles as the previous brick's write-out staging: wait for the compute waves
          // (their statistics tail has a barrier of its own in front of that one; this brick's first weights are on their way)
          if (t9 == 0 && ch == 0 && !first_brick) {
            if (STATS && stats) block_sync();
            block_sync();
          }
          const int nbp = 2 * t9 + 1 < G::NPB ? 2 : (2 * t9 < G::NPB ? 1 : 0);
          if (nbp > 0) {
            const bool b_on = !last_ch || has_next;
            const BrickSrc bsrc = last_ch ? brick_rsrc(nxt, 0, b_on) : brick_rsrc(cur, ch + 1, true);
#pragma unroll
            for (int i = 0; i < nbp; ++i) brick_piece(bsrc, last_ch ? nxt : cur, (cc + 1) & 1, 2 * t9 + i);
          }
          STAMP(st_h1);
          // three slots: what was issued before this iteration has landed (this iteration's NPW + nbp pieces may fly on);
          // two slots: everything has
          if (G::NWS == 3) {
            const int nps = G::NPW + nbp;
            if (nps == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (nps == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (nps == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (nps == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (nps == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (nps == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (nps == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (nps == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          STAMP(st_wait);
          block_sync();                                        // the stage's barrier
          STAMP(st_bar);
          ws = s1;
        }
      }
      if (!has_next) {
        if (STATS && stats) block_sync();                      // the compute waves' statistics tail has one
        break;
      }
      first_brick = false;
      tile = tile_nx;
      cur = nxt;
    }
#ifdef FPLX_STAMP
    if (lane == 0 && fplx_brick_stamp_buf) {
      long long* o_ = fplx_brick_stamp_buf + ((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave8) * 10;
      o_[0] = st_h1; o_[1] = st_wait; o_[2] = st_bar; o_[3] = 0; o_[4] = 0; o_[5] = 0; o_[6] = st_n;
      o_[7] = __builtin_amdgcn_s_memtime() - st_begin; o_[8] = __builtin_amdgcn_s_memrealtime() - st_rbegin; o_[9] = 2;
    }
#endif
#undef STAMP
    return;
  }

  // ================================================================ compute waves
  f32x16 acc[TD][NTW];
#pragma unroll
  for (int p = 0; p < TD; ++p)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[p][j][i] = 0.f;
  BrickPos cur, nxt;
  int64_t tile = tr.first;
  position(tile, cur);
  nxt = cur;
  if (tid < G::NT) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
  block_sync();                                                // P0

  const int L0 = (hhalf * 4 + bk_row(r)) * G::SWP + bk_col(r);
  const int bb = (wn * (32 * NTW) + r) * G::ROWB + ((khalf ^ ((r >> 2) & 3)) << 4);
  bf16x8 fa[2][G::SD], fb[2][3 * NTW];
  auto load_a = [&](const char* brick, int kh, int kw, int ks, int buf) {
    int a0 = L0 + kh * G::SWP + kw;
    asm volatile("" : "+v"(a0));
    const char* p = brick + a0 * G::ROWB + (((2 * ks + khalf) ^ ((a0 >> 2) & 3)) << 4);
#pragma unroll
    for (int q = 0; q < G::SD; ++q) fa[buf][q] = *reinterpret_cast<const bf16x8*>(p + q * G::PL * G::ROWB);
  };
  auto load_b = [&](const char* wslot, int ks, int buf) {
    int b0 = bb;
    asm volatile("" : "+v"(b0));
    const char* p = wslot + (b0 ^ (ks << 5));
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        fb[buf][kd * NTW + j] = *reinterpret_cast<const bf16x8*>(p + (kd * G::NT + j * 32) * G::ROWB);
  };
  const int rh = khalf * 4;
  load_a(bricks, 0, 0, 0, 0);
  load_b(wring, 0, 0);
  int cc = 0, ws = 0;
#ifdef FPLX_STAMP
  long long st_h1 = 0, st_bar = 0, st_h2 = 0, st_wo = 0, st_n = 0;
  const long long st_begin = __builtin_amdgcn_s_memtime();
  const long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#define STAMP(var_) do { const long long t__ = __builtin_amdgcn_s_memtime(); var_ += t__ - st_t; st_t = t__; } while (0)
#else
#define STAMP(var_) do { } while (0)
#endif
  for (;;) {
    const int64_t tile_nx = tile + tr.step;
    const bool has_next = tile_nx < tr.end;
    if (has_next) position(tile_nx, nxt);
    for (int ch = 0; ch < nch; ++ch, ++cc) {
      const char* brick = bricks + (cc & 1) * G::BRICK_BYTES;
      const char* brick_nx = bricks + ((cc + 1) & 1) * G::BRICK_BYTES;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int kh = t9 / 3, kw = t9 % 3;
        const int s1 = ws + 1 == G::NWS ? 0 : ws + 1;
        const char* wslot = wring + ws * G::WST_BYTES;
        const char* wslot_nx = wring + s1 * G::WST_BYTES;
        // a half-stage: M1 MFMAs on fragment set hf, and the OTHER set's SD + 3 NTW fragment reads spread between them, one
        // read in front of an MFMA (knob-free A/B: -DFPLX_BRICK_LW_BURST restores the bursts of the round-2 kernel - all A reads
        // in front of the half, the B reads in its middle; the stamps showed the half behind the barrier, where four waves
        // burst 6-7 reads each while the loaders' DMA writes land, 90 cycles longer than the other one)
        auto half = [&](int hf, const char* abrick, int akh, int akw, int aks, const char* bslot, int bks) {
          constexpr int NL = G::SD + 3 * NTW, M1 = G::M1;
          int a0 = L0 + akh * G::SWP + akw;
          asm volatile("" : "+v"(a0));
          const char* pa = abrick + a0 * G::ROWB + (((2 * aks + khalf) ^ ((a0 >> 2) & 3)) << 4);
          int b0 = bb;
          asm volatile("" : "+v"(b0));
          const char* pb = bslot + (b0 ^ (bks << 5));
          auto load = [&](int li) {
            if (li < G::SD) fa[hf ^ 1][li] = *reinterpret_cast<const bf16x8*>(pa + li * G::PL * G::ROWB);
            else {
              const int t = li - G::SD, kd = t / NTW, j = t % NTW;
              fb[hf ^ 1][t] = *reinterpret_cast<const bf16x8*>(pb + (kd * G::NT + j * 32) * G::ROWB);
            }
          };
#ifdef FPLX_BRICK_LW_BURST
#pragma unroll
          for (int li = 0; li < NL; ++li) load(li);
#endif
          int mi = 0, li = 0;
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
            for (int p = 0; p < TD; ++p)
#pragma unroll
              for (int j = 0; j < NTW; ++j) {
#ifndef FPLX_BRICK_LW_BURST
                // reads li with li * M1 / NL <= mi go in front of MFMA mi (all of them by the last quarter of the half)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                  if (li < NL && (li * (M1 - M1 / 4)) / NL <= mi) { load(li); ++li; }
                __builtin_amdgcn_sched_barrier(0);
#endif
                // statistics-free forms (data gradients, split-K partials, inference): the product TRANSPOSED, D^T[cout][voxel]
                // (operands swapped: same fragments, same products in the same k order) - a lane then holds 16 output
                // channels of ONE voxel and the write-out needs no LDS transpose
                if constexpr (STATS) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[hf][p + kd], fb[hf][kd * NTW + j], acc[p][j], 0, 0, 0);
                else acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[hf][kd * NTW + j], fa[hf][p + kd], acc[p][j], 0, 0, 0);
                ++mi;
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
#ifdef FPLX_STAMP
        long long st_t = __builtin_amdgcn_s_memtime();
        ++st_n;
#endif
        half(0, brick, kh, kw, 1, wslot, 1);
        STAMP(st_h1);
        block_sync();                                          // the stage's barrier (see the header)
        STAMP(st_bar);
        if (t9 < 8) half(1, brick, (t9 + 1) / 3, (t9 + 1) % 3, 0, wslot_nx, 0);
        else half(1, brick_nx, 0, 0, 0, wslot_nx, 0);
        STAMP(st_h2);
        ws = s1;
      }
    }
#ifdef FPLX_STAMP
    const long long st_wo0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- write-out: per-wave LDS tiles in the dead brick slot, 16-byte stores
    char* dead = bricks + ((cc - 1) & 1) * G::BRICK_BYTES;
    char* stg = dead + wave * 4096;
    const int d0 = cur.d0, h0 = cur.h0, w0 = cur.w0, n = cur.n;
    const bool full = d0 + G::TD <= D && h0 + G::TH <= H && w0 + G::TW <= W;          // uniform
    unsigned vmask = 0xFFFFu;
    if (!full) {
      vmask = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + rh;
        if (h0 + hhalf * 4 + bk_row(m) < H && w0 + bk_col(m) < W) vmask |= 1u << i;
      }
    }
    f32x2 s1v[NTW], s2v[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) { s1v[j] = f32x2{0.f, 0.f}; s2v[j] = f32x2{0.f, 0.f}; }
    const int mrow = lane >> 2;
    const int64_t yrow0 = ((int64_t)(h0 + hhalf * 4 + bk_row(mrow)) * W + w0 + bk_col(mrow)) * ldy;
    const int64_t yrow1 = ((int64_t)(h0 + hhalf * 4 + bk_row(mrow + 16)) * W + w0 + bk_col(mrow + 16)) * ldy;
    const bool ok0 = h0 + hhalf * 4 + bk_row(mrow) < H && w0 + bk_col(mrow) < W;
    const bool ok1 = h0 + hhalf * 4 + bk_row(mrow + 16) < H && w0 + bk_col(mrow + 16) < W;
    bf16_t* ycol = y + n0 + wn * (32 * NTW) + (lane & 3) * 8;
    const float slope_v = ACT ? *slope_p : 0.f;
    auto write_out = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int p = 0; p < TD; ++p) {
        const int dd = d0 + p;
        if (FULL || dd < D) {
          bf16_t* yp = ycol + ((int64_t)n * D + dd) * H * W * ldy;
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            const float bv = bias_s[wn * (32 * NTW) + j * 32 + r];
            char* tile_ = stg + ((p * NTW + j) & 1) * 2048;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const int m = (i & 3) + 8 * (i >> 2) + rh;
              f32x2 o = f32x2{acc[p][j][i], acc[p][j][i + 1]} + f32x2{bv, bv};
              if (ACT) { o[0] = o[0] > 0.f ? o[0] : o[0] * slope_v; o[1] = o[1] > 0.f ? o[1] : o[1] * slope_v; }
              *reinterpret_cast<bf16_t*>(tile_ + m * 64 + r * 2) = (bf16_t)o[0];
              *reinterpret_cast<bf16_t*>(tile_ + (m + 1) * 64 + r * 2) = (bf16_t)o[1];
              if (STATS) {
                if (!FULL) {
                  if (!((vmask >> i) & 1u)) o[0] = 0.f;
                  if (!((vmask >> (i + 1)) & 1u)) o[1] = 0.f;
                }
                s1v[j] += o;
                s2v[j] = __builtin_elementwise_fma(o, o, s2v[j]);
              }
            }
            const uint4 pk0 = *reinterpret_cast<const uint4*>(tile_ + mrow * 64 + (lane & 3) * 16);
            const uint4 pk1 = *reinterpret_cast<const uint4*>(tile_ + (mrow + 16) * 64 + (lane & 3) * 16);
            if (FULL || ok0) *reinterpret_cast<uint4*>(yp + yrow0 + j * 32) = pk0;
            if (FULL || ok1) *reinterpret_cast<uint4*>(yp + yrow1 + j * 32) = pk1;
          }
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[p][j][i] = 0.f;
      }
    };
    auto write_partial = [&]() {
      float* pz = partial + (int64_t)bz * ((int64_t)N * D * H * W) * Cout + n0 + wn * (32 * NTW) + (lane & 7) * 4;
      float* stf = reinterpret_cast<float*>(stg);
#pragma unroll
      for (int p = 0; p < TD; ++p) {
        const int dd = d0 + p;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            stf[((i & 3) + 8 * (i >> 2) + rh) * 32 + r] = acc[p][j][i];
            acc[p][j][i] = 0.f;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = (lane >> 3) + 8 * q;
            const int hh = h0 + hhalf * 4 + bk_row(m), ww = w0 + bk_col(m);
            const float4 pk = *reinterpret_cast<const float4*>(stf + m * 32 + (lane & 7) * 4);
            if (dd < D && hh < H && ww < W)
              *reinterpret_cast<float4*>(pz + ((((int64_t)n * D + dd) * H + hh) * W + ww) * Cout + j * 32) = pk;
          }
        }
      }
    };
    // transposed accumulators (STATS = false): lane (r, khalf) holds, for the voxel of patch row r, the channels
    // c(i) = 8 (i >> 2) + (i & 3) + 4 khalf of N-tile j in register i
    auto write_transposed = [&]() {
      const int vh = h0 + hhalf * 4 + bk_row(r), vw = w0 + bk_col(r);
      const bool vok = vh < H && vw < W;
      if (partial) {
        // split-K: four 16-byte fp32 stores per tile straight from the registers into partial[z][voxel][Cout]
        float* pz = partial + (int64_t)bz * ((int64_t)N * D * H * W) * Cout + n0 + wn * (32 * NTW) + 4 * khalf;
#pragma unroll
        for (int p = 0; p < TD; ++p) {
          const int dd = d0 + p;
          float* dst = pz + ((((int64_t)n * D + dd) * H + vh) * W + vw) * Cout;
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            if (dd < D && vok) {
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(dst + j * 32 + 8 * g4) =
                    make_float4(acc[p][j][4 * g4], acc[p][j][4 * g4 + 1], acc[p][j][4 * g4 + 2], acc[p][j][4 * g4 + 3]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[p][j][i] = 0.f;
          }
        }
        return;
      }
      const bool has_bias = bias != nullptr;                   // uniform
#pragma unroll
      for (int p = 0; p < TD; ++p) {
        const int dd = d0 + p;
        bf16_t* dst = y + ((((int64_t)n * D + dd) * H + vh) * W + vw) * ldy + n0 + wn * (32 * NTW) + 8 * khalf;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          unsigned pk[8];
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            float o0 = acc[p][j][i], o1 = acc[p][j][i + 1];
            if (has_bias) {
              const int c = wn * (32 * NTW) + j * 32 + 8 * (i >> 2) + (i & 3) + 4 * khalf;
              o0 += bias_s[c]; o1 += bias_s[c + 1];
            }
            if (ACT) { o0 = o0 > 0.f ? o0 : o0 * slope_v; o1 = o1 > 0.f ? o1 : o1 * slope_v; }
            const bf16_t b0 = (bf16_t)o0, b1 = (bf16_t)o1;
            pk[i >> 1] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
            acc[p][j][i] = 0.f; acc[p][j][i + 1] = 0.f;
          }
          // channels 8 g + 4 khalf + (0..3) sit in pk[2 g], pk[2 g + 1]: two v_permlane32_swap per pair of groups give the low
          // lane of a voxel channels 0-7 and 16-23, the high lane 8-15 and 24-31 - two 16-byte stores each
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const u32x2 sw = __builtin_amdgcn_permlane32_swap(pk[4 * g2 + e], pk[4 * g2 + 2 + e], false, false);
              pk[4 * g2 + e] = sw[0];
              pk[4 * g2 + 2 + e] = sw[1];
            }
          if (dd < D && vok) {
            *reinterpret_cast<u32x4*>(dst + j * 32) = u32x4{pk[0], pk[1], pk[2], pk[3]};
            *reinterpret_cast<u32x4*>(dst + j * 32 + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
          }
        }
      }
    };
    if constexpr (!STATS) write_transposed();
    else if (partial) write_partial();
    else if (full) write_out(std::true_type{});
    else write_out(std::false_type{});
    if (STATS && stats) {
      float* red = reinterpret_cast<float*>(dead + 16384);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const float a_ = s1v[j][0] + s1v[j][1], q_ = s2v[j][0] + s2v[j][1];
        const float a = a_ + __shfl_xor(a_, 32, 64), q2 = q_ + __shfl_xor(q_, 32, 64);
        if (lane < 32) {
          red[(hhalf * 2 + 0) * G::NT + wn * (32 * NTW) + j * 32 + r] = a;
          red[(hhalf * 2 + 1) * G::NT + wn * (32 * NTW) + j * 32 + r] = q2;
        }
      }
      block_sync();
      if (tid < 2 * G::NT) {
        const int which = tid / G::NT, c = tid % G::NT;
        float t_ = red[(0 * 2 + which) * G::NT + c];
        if (WH == 2) t_ += red[(1 * 2 + which) * G::NT + c];
        stats[((int64_t)tile * 2 + which) * Cout + n0 + c] = t_;
      }
    }
#ifdef FPLX_STAMP
    st_wo += __builtin_amdgcn_s_memtime() - st_wo0;
#endif
    if (!has_next) break;
    block_sync();                       // the loaders may now fetch into the staging slot (their barrier in a brick's first stage)
    tile = tile_nx;
    cur = nxt;
  }
#ifdef FPLX_STAMP
  if (lane == 0 && fplx_brick_stamp_buf) {
    long long* o_ = fplx_brick_stamp_buf + ((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave8) * 10;
    o_[0] = st_h1; o_[1] = 0; o_[2] = st_bar; o_[3] = 0; o_[4] = st_h2; o_[5] = st_wo; o_[6] = st_n;
    o_[7] = __builtin_amdgcn_s_memtime() - st_begin; o_[8] = __builtin_amdgcn_s_memrealtime() - st_rbegin; o_[9] = 1;
  }
#endif
#undef STAMP
}

inline int brick_enabled() {
  return (int)fplx_knob(FPLX_K_BRICK);              // A/B knob (benchmarks only)
}

struct BrickCfg { int ok, geo, ntw, nt, ksplit, bD, bH, bW; int64_t bricks; };

// geo 0: 4 x 8 x 8 bricks (128 or 64 output channels per block), geo 1: 5 x 4 x 8 (128): whichever pads the volume less;
// ksplit > 1 deals the 32-channel chunks of Cin to blockIdx.z when the bricks alone leave most of the 256 CUs idle (the
// deep levels) - the smallest split that fills them, because each split adds an fp32 partial tensor to write and re-read
inline BrickCfg brick_cfg(int n, int d, int h, int w, int cin, int cout) {
  BrickCfg c = {};
  if (!brick_enabled() || cin % BK::KC != 0 || cin < 64 || cout % 64 != 0) return c;
  if ((int64_t)d * h * w * cin * 2 >= ((int64_t)1 << 30)) return c;
  const int64_t V = (int64_t)n * d * h * w;
  const int bw = (w + 7) / 8;
  const int64_t b0 = (int64_t)n * ((d + 3) / 4) * ((h + 7) / 8) * bw, p0 = b0 * 256;
  const int64_t b1 = (int64_t)n * ((d + 4) / 5) * ((h + 3) / 4) * bw, p1 = b1 * 160;
  c.geo = (cout % 128 == 0 && p1 < p0) ? 1 : 0;
  // tests (fplx_set_tuning "brick_geo" / "brick_ksplit"): this geometry / Cin split on every layer the kernel can compute at
  // all, also where the plan would leave it to another kernel (few or ragged bricks)
  const int force_geo = (int)fplx_knob(FPLX_K_BRICK_GEO), force_ks = (int)fplx_knob(FPLX_K_BRICK_KSPLIT);
  if (force_geo >= 0) {
    if (force_geo > 1 || (force_geo == 1 && cout % 128 != 0)) return c;
    c.geo = force_geo;
  }
  c.nt = (c.geo == 1 || cout % 128 == 0) ? 128 : 64;
  c.ntw = c.geo == 1 ? 1 : c.nt / 64;
  c.bricks = c.geo ? b1 : b0;
  c.bD = c.geo ? (d + 4) / 5 : (d + 3) / 4;
  c.bH = c.geo ? (h + 3) / 4 : (h + 7) / 8;
  c.bW = bw;
  // more than 25 % padding: the tile kernel's, except for tiny volumes (below)
  if (c.bricks >= ((int64_t)1 << 24) || cout % 8 != 0 || cout > 2048) return c;
  const int64_t blocks = c.bricks * (cout / c.nt);
  const int nch = cin / BK::KC;
  c.ksplit = 1;
  if (force_geo >= 0 || force_ks > 0) {
    if (force_ks > nch) return c;
    if (force_ks > 0) c.ksplit = force_ks;
    c.ok = 1;
    return c;
  }
  // (tiny volumes - level 4 of the benchmark, 5 x 10 x 10, pads 1.9x as 5 x 4 x 8 bricks - still beat the tile kernel since the
  // brick kernel spreads its DMA pieces: 512 -> 512 53 -> 45 us, 256 -> 512 39 -> 34, 512 -> 256 37 -> 33 with the split below)
  const bool tiny = c.geo == 1 && V <= 2048 && p1 <= 2 * V;
  if ((c.geo ? p1 : p0) * 4 > V * 5 && !tiny) return c;
  const int64_t fill = fplx_knob(FPLX_K_BRICK_FILL);          // blocks the launch should have before the Cin split stops growing
  while (blocks * c.ksplit < fill && nch / (c.ksplit + 1) >= 2) ++c.ksplit;
  if (blocks * c.ksplit < (fill < 192 ? fill : 192)) return c;
  c.ok = 1;
  return c;
}

}  // namespace

// 1 if the brick kernel takes this 3x3x3 layer
extern "C" int fplx_brick_ok(int n, int d, int h, int w, int cin, int cout) { return brick_cfg(n, d, h, w, cin, cout).ok; }
// the layers a march kernel could take as well: the brick kernel is the faster one on all of them (level 1 of the benchmark,
// 2 x 40 x 80 x 80, with statistics: 128 -> 64 305 -> 219 us against the streamed-weight march, 64 -> 128 255 -> 213 us,
// 64 -> 64 157 -> 142 us).  A/B knob FPLX_BRICK=3: only the layers no march kernel takes (benchmarks only)
extern "C" int fplx_brick_first(int n, int d, int h, int w, int cin, int cout) {
  const int en = brick_enabled();
  return en && en != 3 && brick_cfg(n, d, h, w, cin, cout).ok;
}
// the plan for a layer fplx_brick_ok accepts: brick geometry, Cin split (1 = none) and the number of bricks = the statistics rows
// the kernel itself writes (with ksplit > 1 the rows are splitk_finish_k's)
extern "C" int fplx_brick_plan(int n, int d, int h, int w, int cin, int cout, int* geo, int* ksplit, int* bricks) {
  const BrickCfg c = brick_cfg(n, d, h, w, cin, cout);
  if (geo) *geo = c.geo;
  if (ksplit) *ksplit = c.ksplit;
  if (bricks) *bricks = (int)c.bricks;
  return c.ok;
}
// bricks of a given geometry (tests that call fplx_brick_conv3d_fwd_ex directly)
extern "C" int fplx_brick_rows(int n, int d, int h, int w, int geo) {
  return n * (geo ? (d + 4) / 5 : (d + 3) / 4) * (geo ? (h + 3) / 4 : (h + 7) / 8) * ((w + 7) / 8);
}

// returns 1 if launched, 0 if the operands do not allow it (alignment), <0 on error.  geo / ksplit: fplx_brick_plan's, or a
// test's choice on shapes the plan leaves to other kernels.  ksplit > 1: the kernel writes partial[ksplit][V][cout]
// fp32 and the caller finishes (splitk_finish_k)
extern "C" int fplx_brick_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                         int n, int d, int h, int w, int cin, int cout, float* stats, float* partial, int geo,
                                         int ksplit, hipStream_t st, const float* slope, const void* x1, int nmod0) {
  if (ldy % 8 != 0 || ((uintptr_t)y % 16) != 0 || ldx % 8 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)wp % 16) != 0)
    return 0;
  // two-tensor input: the activation forms without a Cin split only
  if (x1 && (((uintptr_t)x1 % 16) != 0 || !slope || ksplit != 1 || cin % (2 * BK::KC) != 0 || nmod0 < 0 || (nmod0 && n % nmod0)))
    return 0;
  if ((int64_t)d * h * w * ldx * 2 >= ((int64_t)1 << 30) || cin % BK::KC != 0 || cin < 64 || cout % 64 != 0) return 0;
  if (geo == 1 && cout % 128 != 0) return 0;
  if (ksplit < 1 || ksplit > cin / BK::KC || (ksplit > 1 && !partial)) return 0;
  const int bD = geo ? (d + 4) / 5 : (d + 3) / 4, bH = geo ? (h + 3) / 4 : (h + 7) / 8, bW = (w + 7) / 8;
  const int nt = (geo == 1 || cout % 128 == 0) ? 128 : 64;
  // persistent blocks: one per CU (LDS), each walking its share of the bricks
  const int gy = cout / nt;
  int64_t gx = (int64_t)n * bD * bH * bW;
  const int64_t per = (256 + gy * ksplit - 1) / (gy * ksplit);
  if (gx > per) gx = per;
  // XCD order (see the kernel): by weight slice when the pack does not fit an L2 beside the activations - that needs
  // grid.x * slices to be a multiple of 8 (and 8 | slices or slices | 8) - else by brick neighbourhood
  const int U = gy * ksplit;
  int xcd_on = fplx_xcd_on() ? 1 : 0;
  if (xcd_on && (int64_t)27 * cout * cin * 2 > ((int64_t)3 << 20) && (8 % U == 0 || U % 8 == 0)) {
    const int q = U >= 8 ? 1 : 8 / U;
    gx = (gx + q - 1) / q * q;                                 // blocks past the brick list leave at once
    xcd_on = 2;
  }
  dim3 grid((unsigned)gx, gy, ksplit);
  float* part = ksplit > 1 ? partial : nullptr;
#define LAUNCH_BRICK(STATS_, NTW_, TD_, WH_)                                                                         \
  do {                                                                                                               \
    using G_ = BKG<TD_, WH_, NTW_>;                                                                                  \
    (void)hipFuncSetAttribute((const void*)conv_fwd_brick_lw<STATS_, NTW_, TD_, WH_>,                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS);                                  \
    conv_fwd_brick_lw<STATS_, NTW_, TD_, WH_><<<grid, 512, G_::LDS, st>>>(                                            \
        (const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, ldy, n, d, h, w, cin, cout, stats, part, bD, bH,  \
        bW, xcd_on);                                                                                                 \
  } while (0)
  const bool st_ = stats && !part;
#define LAUNCH_BRICK_ACT(NTW_, TD_, WH_)                                                                              \
  do {                                                                                                               \
    using G_ = BKG<TD_, WH_, NTW_>;                                                                                  \
    (void)hipFuncSetAttribute((const void*)conv_fwd_brick_lw<false, NTW_, TD_, WH_, true>,                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS);                                  \
    conv_fwd_brick_lw<false, NTW_, TD_, WH_, true><<<grid, 512, G_::LDS, st>>>(                                       \
        (const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, ldy, n, d, h, w, cin, cout, nullptr, nullptr, bD,  \
        bH, bW, xcd_on, slope);                                                                                      \
  } while (0)
#define LAUNCH_BRICK_ACT2(NTW_, TD_, WH_)                                                                             \
  do {                                                                                                               \
    using G_ = BKG<TD_, WH_, NTW_>;                                                                                  \
    (void)hipFuncSetAttribute((const void*)conv_fwd_brick_lw<false, NTW_, TD_, WH_, true, true>,                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS);                                  \
    conv_fwd_brick_lw<false, NTW_, TD_, WH_, true, true><<<grid, 512, G_::LDS, st>>>(                                 \
        (const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, ldy, n, d, h, w, cin, cout, nullptr, nullptr, bD,  \
        bH, bW, xcd_on, slope, (const bf16_t*)x1, nmod0);                                                            \
  } while (0)
  if (x1) {
    if (geo == 1) LAUNCH_BRICK_ACT2(1, 5, 1);
    else if (nt == 128) LAUNCH_BRICK_ACT2(2, 4, 2);
    else LAUNCH_BRICK_ACT2(1, 4, 2);
  }
  else if (slope && !part) {          // (with a Cin split the activation is the finish kernel's)
    if (geo == 1) LAUNCH_BRICK_ACT(1, 5, 1);
    else if (nt == 128) LAUNCH_BRICK_ACT(2, 4, 2);
    else LAUNCH_BRICK_ACT(1, 4, 2);
  }
  else if (geo == 1) { if (st_) LAUNCH_BRICK(true, 1, 5, 1); else LAUNCH_BRICK(false, 1, 5, 1); }
  else if (nt == 128) { if (st_) LAUNCH_BRICK(true, 2, 4, 2); else LAUNCH_BRICK(false, 2, 4, 2); }
  else { if (st_) LAUNCH_BRICK(true, 1, 4, 2); else LAUNCH_BRICK(false, 1, 4, 2); }
#undef LAUNCH_BRICK_ACT2
#undef LAUNCH_BRICK_ACT
#undef LAUNCH_BRICK
  const int rc = fplx_check_launch("brick_conv3d_fwd");
  return rc < 0 ? rc : 1;
}

extern "C" int fplx_brick_conv3d_fwd_ex(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                        int n, int d, int h, int w, int cin, int cout, float* stats, float* partial, int geo,
                                        int ksplit, hipStream_t st) {
  return fplx_brick_conv3d_fwd_act(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, partial, geo, ksplit, st, nullptr, nullptr,
                                   0);
}
