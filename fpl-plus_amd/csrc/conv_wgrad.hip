// Rolling-window weight gradient of the 3x3x3 convolution for gfx950 (bf16 NDHWC operands, fp32 accumulate).
//
//   dW[tap][ci][co] = sum over voxels v of x[v + tap][ci] * dy[v][co]          (autograd of nn.Conv3d,
//   reference PyMIC/pymic/net/net3d/unet2d5_dsbn.py:54-55,75,79 - ConvolutionLayer inside ConvBlockND)
//
// conv_wgrad_roll replaces conv_wgrad_stream (conv_mfma.hip) on the large levels.  Same decomposition - a block owns a
// TH x TW footprint in (h, w) of one 32 x 32 (ci, co) tile pair and marches along d; voxels are the K dimension; both MFMA
// operands are transposes of the NDHWC image and come out of LDS through ds_read_b64_tr_b16; per-block partial tiles in the
// same [tap][co][ci] format, summed in a fixed order by wgrad_stream_reduce - but:
//   * ROLLING WINDOW over the kernel row.  The x fragment (slab kd, slab row rx, column offset kw) is the A operand of the
//     three taps (kd, kh, kw), kh = 0..2, paired with the dy rows rx - kh.  A wave owns whole kh-triples, walks rx over the
//     TH + 2 slab rows, loads that fragment ONCE and keeps the last three dy fragments in registers: 4 fragments per 7 MFMAs
//     (1.3 transposed reads per MFMA with the edge rows) where the tap-per-fragment form reads 8 (2.3).  The 9 triples
//     (kd, kw) are dealt 2 + 2 + 2 + 2 to the four waves and the ninth is split into its three taps: 7 / 7 / 7 / 6 tiles.
//   * <= 256 REGISTERS per lane (7 accumulator tiles + 10 fragments + 10 DMA offsets), one wave per SIMD: half of every
//     SIMD's register file stays free, so the HBM-bound BatchNorm passes of the main stream co-reside on every CU while the
//     weight gradients run on the second stream (profiles/r03_coresidency_probe.txt: beside the 504-register form the
//     level-0 apply pass took 361-522 us instead of 128).
//   * LDS-DMA staging through buffer descriptors (buffer_load_dwordx4 ... offen lds): halo, padding and ragged footprints are
//     the hardware's out-of-range zeros, the per-lane offsets are constants of the march, the depth is the scalar offset;
//     no staging registers, no commit phase, a 4-slot x ring + 2-slot dy ring.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"      // LDS pointers are 32 bits wide: they are formed from 32-bit addresses

extern "C" int fplx_wgrad_reduce_launch(const float* part, int nblk, int npairs, int cin, int cout, float* dw, int mid,
                                        hipStream_t st);      // conv_mfma.hip

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// MB: the step's one barrier sits in the MIDDLE of the step and the rings get one x slot and one dy slot more (5 + 3): the DMA
// of a slab is issued behind the barrier of the step BEFORE the one whose barrier publishes it, into a slot nobody has read
// since the step before that - so the first fragments of step t + 1 are requested during the last MFMAs of step t and no
// step begins with an empty pipe (with the barrier at the end of the step they can only be requested behind it).
template <int TH_, int TW_, bool MB_>
struct WR {
  static constexpr bool MB = MB_;
  static constexpr int TH = TH_, TW = TW_, SH = TH + 2, SW = TW + 2, SLAB = SH * SW;
  static constexpr int XP = (SLAB * 4 + 63) / 64;          // 1-KiB DMA pieces per x slab (64-byte voxel rows)
  static constexpr int XSLOT = XP * 1024;
  static constexpr int YP = TH * TW * 4 / 64;              // ... per dy slab
  static constexpr int YSLOT = YP * 1024;
  static constexpr int NXS = MB ? 5 : 4, NYS = MB ? 3 : 2;
  static constexpr int LDS = NXS * XSLOT + NYS * YSLOT;
  static constexpr int NCH = TW / 16;                      // 16-voxel K chunks per row
  static constexpr int NCELL = NCH * SH;                   // (chunk, slab row) cells per depth step
  static constexpr int XPW = (XP + 3) / 4, YPW = (YP + 3) / 4;   // pieces per wave
};

// fragment reads take 32-bit LDS byte addresses: an opaque per-step base (one VGPR per slab, see lds_base) + a compile-time
// offset that lands in the instruction's 16-bit offset field
__device__ __forceinline__ bf16x8 tr_frag(unsigned base_lo) {
  // two transposed 4 x 16 block reads: voxels +0..3 and +4..7 of a lane group's 8 (64 bytes per voxel row)
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base_lo));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base_lo + 4 * 64));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// hipcc must not take a base apart and hoist base + constant sums out of the depth loop (it did: 30 address registers)
__device__ __forceinline__ unsigned lds_base(unsigned a) {
  asm volatile("" : "+v"(a));
  return a;
}

// Tap ownership of wave WV: triples (kd, kw) = 2 WV, 2 WV + 1 of the row-major (kd, kw) enumeration, each with its three kh;
// waves 0-2 also own tap (kd 2, kh WV, kw 2) of the ninth triple.  Local tile j * 3 + kh (triple j), 6 = the single.
template <int WV> struct RollTaps {
  static constexpr int NT = WV < 3 ? 7 : 6;
  static constexpr int kd(int j) { return j < 2 ? (2 * WV + j) / 3 : 2; }
  static constexpr int kw(int j) { return j < 2 ? (2 * WV + j) % 3 : 2; }
  static constexpr int tap(int i) { return i < 6 ? kd(i / 3) * 9 + (i % 3) * 3 + kw(i / 3) : 18 + WV * 3 + 2; }
};

template <class G, int WV>
__device__ __forceinline__ void roll_march(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                           int64_t ldy, float* __restrict__ part, int D, int H, int W, int Cin, int Cout,
                                           int tilesH, int tilesW, int dsegs, int dlen, const bf16_t* __restrict__ x1,
                                           const FplxBlock bid) {
  using T = RollTaps<WV>;
  constexpr int TH = G::TH, TW = G::TW, SH = G::SH, SW = G::SW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ys = smem + G::NXS * G::XSLOT;
  const int lane = threadIdx.x & 63;
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int ncg = Cin / 32;
  const int cot = bid.y / ncg, cg = bid.y % ncg;
  const int h0 = th * TH, w0 = tw * TW;
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;

  // ---- DMA geometry: per-lane byte offsets inside a depth slice (constants of the march), 0x40000000 = out of range
  unsigned xvo[G::XPW], yvo[G::YPW];
#pragma unroll
  for (int k = 0; k < G::XPW; ++k) {
    const int i = (WV + 4 * k) * 64 + lane;
    const int vox = i >> 2, c = i & 3;
    const int hh = vox / SW + h0 - 1, ww = vox % SW + w0 - 1;
    const bool in = vox < G::SLAB && hh >= 0 && hh < H && ww >= 0 && ww < W;
    xvo[k] = in ? (unsigned)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : 0x40000000u;
  }
#pragma unroll
  for (int k = 0; k < G::YPW; ++k) {
    const int i = (WV + 4 * k) * 64 + lane;
    const int vox = i >> 2, c = i & 3;
    const int hh = vox / TW + h0, ww = vox % TW + w0;
    yvo[k] = (hh < H && ww < W) ? (unsigned)((((int64_t)hh * W + ww) * ldy + c * 8) * 2) : 0x40000000u;
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2, yslice = (int64_t)H * W * ldy * 2;
  // the two ci tiles of a split concatenation (x1 != NULL, Cin = 64) come from two tensors
  const char* xn = reinterpret_cast<const char*>((x1 && cg == 1) ? x1 : x + cg * 32) + (int64_t)n * D * xslice;
  const char* yn = reinterpret_cast<const char*>(dy + cot * 32) + (int64_t)n * D * yslice;
  u32x4 xr, yr;
  xr[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xn);
  xr[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xn >> 32) & 0xFFFFu);
  xr[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * xslice - (ldx - 32) * 2));
  xr[3] = 0x00020000u;
  yr[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)yn);
  yr[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)yn >> 32) & 0xFFFFu);
  yr[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * yslice - (ldy - 32) * 2));
  yr[3] = 0x00020000u;
  const unsigned xslice32 = __builtin_amdgcn_readfirstlane((unsigned)xslice);
  const unsigned yslice32 = __builtin_amdgcn_readfirstlane((unsigned)yslice);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  auto dma = [&](const u32x4& rsrc, unsigned vo, unsigned so, unsigned dst) {    // dst: LDS byte address of the 1-KiB piece
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so), "s"(dst) : "memory");
  };
  // x slab of depth d0 - 1 + m lives in slot m & 3, the dy slab of depth d0 + t in slot t & 1
  // on = false (nothing follows this block's last depth): the scalar offset is out of range, the slot - a free one - gets zeros
  auto dma_x = [&](int m, int k, bool on) {            // piece WV + 4 k of slab m
    if (WV + 4 * k < G::XP) {
      const int s = d0 - 1 + m;
      const unsigned so = __builtin_amdgcn_readfirstlane((on && s >= 0 && s < D) ? (unsigned)s * xslice32 : 0x40000000u);
      dma(xr, xvo[k], so, lds0 + (unsigned)(((unsigned)m % G::NXS) * G::XSLOT + (WV + 4 * k) * 1024));
    }
  };
  auto dma_y = [&](int t, int k, bool on) {
    if (WV + 4 * k < G::YP) {
      const int s = d0 + t;
      const unsigned so = __builtin_amdgcn_readfirstlane((on && s < D) ? (unsigned)s * yslice32 : 0x40000000u);
      dma(yr, yvo[k], so, lds0 + (unsigned)(G::NXS * G::XSLOT + ((unsigned)t % G::NYS) * G::YSLOT + (WV + 4 * k) * 1024));
    }
  };

  // transposed-read lane geometry: group g = lane / 16 reads voxel rows 8 (g >> 1) + q, channels 16 (g & 1) + 4 p ..
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;

  f32x16 acc[T::NT];
#pragma unroll
  for (int i = 0; i < T::NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // prologue: slabs d0 - 1, d0, d0 + 1 and dy(d0) (MB: also the slab and dy of the second step)
  const int nd = d1 - d0;
#pragma unroll
  for (int m = 0; m < (G::MB ? 4 : 3); ++m)
#pragma unroll
    for (int k = 0; k < G::XPW; ++k) dma_x(m, k, m < 3 || nd > 1);
#pragma unroll
  for (int k = 0; k < G::YPW; ++k) dma_y(0, k, true);
  if (G::MB) {
#pragma unroll
    for (int k = 0; k < G::YPW; ++k) dma_y(1, k, nd > 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 fx[2][3];       // [cell parity][triple 0, triple 1, single]
  bf16x8 fy[4];          // dy rows, ring by row & 3
  constexpr int PB = G::NCELL / 2;                        // MB: the barrier sits behind cell PB - 1
  static_assert(!G::MB || 2 * (G::NCELL - PB) >= G::XPW + G::YPW, "DMA pieces must fit behind the barrier");
  // load item i of cell k: 0 / 1 = the triples' x fragments, 2 = the single's, 3 = the dy row entering the window
  auto load_item = [&](const unsigned (&sb)[3], unsigned yb, int k, int i) {
    const int c = k / SH, rx = k % SH;
    if (i < 2) fx[k & 1][i] = tr_frag(sb[T::kd(i)] + ((rx * SW + c * 16 + T::kw(i)) * 64));
    else if (i == 2) { if (WV < 3 && rx - WV >= 0 && rx - WV < TH) fx[k & 1][2] = tr_frag(sb[2] + ((rx * SW + c * 16 + 2) * 64)); }
    else if (rx < TH) fy[rx & 3] = tr_frag(yb + ((rx * TW + c * 16) * 64));
  };
  auto mfma_item = [&](int k, int i) {
    const int rx = k % SH;
    const int r = i < 6 ? rx - i % 3 : rx - WV;        // dy row of this tap
    if (i < T::NT && r >= 0 && r < TH)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[k & 1][i < 6 ? i / 3 : 2], fy[r & 3], acc[i], 0, 0, 0);
  };
  auto bases = [&](int t, unsigned (&sb)[3], unsigned& yb) {
#pragma unroll
    for (int k = 0; k < 3; ++k) sb[k] = lds_base(lds0 + ((unsigned)(t + k) % G::NXS) * G::XSLOT + lane_off);
    yb = lds_base(lds0 + G::NXS * G::XSLOT + ((unsigned)t % G::NYS) * G::YSLOT + lane_off);
  };
  if (G::MB) {                                            // the first cell of the first step
    unsigned sb0[3], yb0;
    bases(0, sb0, yb0);
    load_item(sb0, yb0, 0, 3); load_item(sb0, yb0, 0, 0); load_item(sb0, yb0, 0, 1); load_item(sb0, yb0, 0, 2);
  }
#pragma unroll 1
  for (int t = 0; t < nd; ++t) {
    unsigned sb[3], yb, sn[3], yn;
    bases(t, sb, yb);
    bases(t + 1, sn, yn);
    const bool more = t + 1 < nd, more2 = t + 2 < nd;
    if (!G::MB) { load_item(sb, yb, 0, 3); load_item(sb, yb, 0, 0); load_item(sb, yb, 0, 1); load_item(sb, yb, 0, 2); }
#pragma unroll
    for (int k = 0; k < G::NCELL; ++k) {
      // the fragments of cell k + 1 (MB: of the next step's first cell behind the last one) are requested one at a time in
      // the gaps between the MFMAs of cell k; the DMA pieces of the coming slabs ride in the gaps as well
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int li = i == 0 ? 3 : (i == 2 ? 0 : (i == 4 ? 1 : (i == 5 ? 2 : -1)));
        if (li >= 0) {
          if (k + 1 < G::NCELL) load_item(sb, yb, k + 1, li);
          else if (G::MB) load_item(sn, yn, 0, li);
        }
        if (!G::MB) {
          if (i == 3) {
            if (k < G::XPW) dma_x(t + 3, k, more);
            else if (k - G::XPW < G::YPW) dma_y(t + 1, k - G::XPW, more);
          }
        } else if ((i == 1 || i == 3) && k >= PB) {       // behind this step's barrier: slab t + 4 and dy(t + 2), two pieces per cell
          const int q = 2 * (k - PB) + (i == 3);
          if (q < G::XPW) dma_x(t + 4, q, more2);
          else if (q - G::XPW < G::YPW) dma_y(t + 2, q - G::XPW, more2);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_item(k, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (G::MB && k == PB - 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!G::MB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // partial tiles: part[block][pair][tap][co][ci] - a lane owns 4 consecutive ci per register quad: 16-byte stores
  const int co = lane & 31, rbase = (lane >> 5) * 4;
  const int pair = cot * ncg + cg;
  float* out = part + ((int64_t)bid.x * (ncg * (Cout / 32)) + pair) * (27 * 1024);
#pragma unroll
  for (int i = 0; i < T::NT; ++i) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<float4*>(out + (T::tap(i) * 32 + co) * 32 + 8 * g4 + rbase) =
          make_float4(acc[i][4 * g4 + 0], acc[i][4 * g4 + 1], acc[i][4 * g4 + 2], acc[i][4 * g4 + 3]);
  }
}

// ------------------------------------------------------------------------------------------
// roll16_march: the same march on v_mfma_f32_16x16x32_bf16 (TW = 32: one K = 32-voxel chunk per footprint row).  These kernels
// run against the chip's power limit, and the 16 x 16 x 32 shape moves the same FLOPs for less (MI355X_MICROARCH.md, DVFS
// give-back item 7).  A 32 x 32 tile is four 16 x 16 blocks (ci half a, co half b); per tap and row 2 + 2 fragments feed 4 MFMAs
// - the LDS traffic per FLOP of the 32 x 32 x 16 form.  The LDS image is two PLANES per slab, one per 16-channel half, with
// 32-byte voxel rows: a fragment's 8 voxels x 16 channels per 32 lanes are then 256 contiguous bytes - conflict-free for every
// tap shift without a swizzle - and the lane offset of a transposed read is simply lane * 8.  MFMA k-slot 8 g + j (g = lane / 16)
// holds voxel 4 g + (j & 3) + 16 (j >> 2) of the chunk, for x and dy alike (any bijection serves, both operands use this one).
// A DMA piece is 32 voxels of one plane: lane i fetches 16-byte chunk 2 cb + (i & 1) of voxel i / 2.
template <int TH_, bool MB_>
struct WR16 {
  static constexpr bool MB = MB_;
  static constexpr int TH = TH_, TW = 32, SH = TH + 2, SW = TW + 2, SLAB = SH * SW;
  static constexpr int XPP = (SLAB * 2 + 63) / 64;         // 1-KiB pieces per x plane (32-byte rows)
  static constexpr int XPLANE = XPP * 1024, XP = 2 * XPP, XSLOT = 2 * XPLANE;
  static constexpr int YPP = TH * TW * 2 / 64;
  static constexpr int YPLANE = YPP * 1024, YP = 2 * YPP, YSLOT = 2 * YPLANE;
  static constexpr int NXS = MB ? 5 : 4, NYS = MB ? 3 : 2;
  static constexpr int LDS = NXS * XSLOT + NYS * YSLOT;
  static constexpr int NCELL = SH;
  static constexpr int XPW = (XP + 3) / 4, YPW = (YP + 3) / 4;
};

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <class G, int WV>
__device__ __forceinline__ void roll16_march(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                             int64_t ldy, float* __restrict__ part, int D, int H, int W, int Cin, int Cout,
                                             int tilesH, int tilesW, int dsegs, int dlen, const bf16_t* __restrict__ x1,
                                             const FplxBlock bid) {
  using T = RollTaps<WV>;
  constexpr int TH = G::TH, TW = G::TW, SH = G::SH, SW = G::SW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int ncg = Cin / 32;
  const int cot = bid.y / ncg, cg = bid.y % ncg;
  const int h0 = th * TH, w0 = tw * TW;
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;

  // ---- DMA geometry.  Piece P = 2 p + cb (p-th KiB of plane cb): the two halves of a voxel row are fetched by neighbouring
  // waves in the same cell, i.e. close in time (the second request for the line hits in L2)
  unsigned xvo[G::XPW], yvo[G::YPW];
#pragma unroll
  for (int k = 0; k < G::XPW; ++k) {
    const int P = WV + 4 * k, pp = P >> 1, cb = P & 1;
    const int vox = pp * 32 + (lane >> 1), c = 2 * cb + (lane & 1);
    const int hh = vox / SW + h0 - 1, ww = vox % SW + w0 - 1;
    const bool in = vox < G::SLAB && hh >= 0 && hh < H && ww >= 0 && ww < W;
    xvo[k] = in ? (unsigned)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : 0x40000000u;
  }
#pragma unroll
  for (int k = 0; k < G::YPW; ++k) {
    const int P = WV + 4 * k, pp = P >> 1, cb = P & 1;
    const int vox = pp * 32 + (lane >> 1), c = 2 * cb + (lane & 1);
    const int hh = vox / TW + h0, ww = vox % TW + w0;
    yvo[k] = (hh < H && ww < W) ? (unsigned)((((int64_t)hh * W + ww) * ldy + c * 8) * 2) : 0x40000000u;
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2, yslice = (int64_t)H * W * ldy * 2;
  const char* xn = reinterpret_cast<const char*>((x1 && cg == 1) ? x1 : x + cg * 32) + (int64_t)n * D * xslice;
  const char* yn = reinterpret_cast<const char*>(dy + cot * 32) + (int64_t)n * D * yslice;
  u32x4 xr, yr;
  xr[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xn);
  xr[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xn >> 32) & 0xFFFFu);
  xr[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * xslice - (ldx - 32) * 2));
  xr[3] = 0x00020000u;
  yr[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)yn);
  yr[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)yn >> 32) & 0xFFFFu);
  yr[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * yslice - (ldy - 32) * 2));
  yr[3] = 0x00020000u;
  const unsigned xslice32 = __builtin_amdgcn_readfirstlane((unsigned)xslice);
  const unsigned yslice32 = __builtin_amdgcn_readfirstlane((unsigned)yslice);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  auto dma = [&](const u32x4& rsrc, unsigned vo, unsigned so, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so), "s"(dst) : "memory");
  };
  auto dma_x = [&](int m, int k, bool on) {
    if (WV + 4 * k < G::XP) {
      constexpr int dummy = 0; (void)dummy;
      const int P = WV + 4 * k;
      const int s = d0 - 1 + m;
      const unsigned so = __builtin_amdgcn_readfirstlane((on && s >= 0 && s < D) ? (unsigned)s * xslice32 : 0x40000000u);
      dma(xr, xvo[k], so, lds0 + (unsigned)(((unsigned)m % G::NXS) * G::XSLOT + (P & 1) * G::XPLANE + (P >> 1) * 1024));
    }
  };
  auto dma_y = [&](int t, int k, bool on) {
    if (WV + 4 * k < G::YP) {
      const int P = WV + 4 * k;
      const int s = d0 + t;
      const unsigned so = __builtin_amdgcn_readfirstlane((on && s < D) ? (unsigned)s * yslice32 : 0x40000000u);
      dma(yr, yvo[k], so, lds0 + (unsigned)(G::NXS * G::XSLOT + ((unsigned)t % G::NYS) * G::YSLOT + (P & 1) * G::YPLANE + (P >> 1) * 1024));
    }
  };

  f32x4 acc[T::NT][4];                                   // [tap][2 a + b]: ci half a x co half b
#pragma unroll
  for (int i = 0; i < T::NT; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][q][r] = 0.f;

  const int nd = d1 - d0;
#pragma unroll
  for (int m = 0; m < (G::MB ? 4 : 3); ++m)
#pragma unroll
    for (int k = 0; k < G::XPW; ++k) dma_x(m, k, m < 3 || nd > 1);
#pragma unroll
  for (int k = 0; k < G::YPW; ++k) dma_y(0, k, true);
  if (G::MB) {
#pragma unroll
    for (int k = 0; k < G::YPW; ++k) dma_y(1, k, nd > 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // a fragment = 16 channels x 32 voxels: two transposed reads of 512 contiguous bytes (voxels +0..15 and +16..31 of the plane)
  auto frag = [&](unsigned pl) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pl));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pl + 16 * 32));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  };
  bf16x8 fx[2][3][2];    // [cell parity][triple 0, triple 1, single][ci half]
  bf16x8 fy[4][2];       // dy rows (ring by row & 3) x co half
  const int lane8 = lane * 8;
  constexpr int PB = SH / 2;                              // MB: the barrier sits behind cell PB - 1
  static_assert(!G::MB || 2 * (SH - PB) >= (G::XPW > G::YPW ? G::XPW : G::YPW), "DMA pieces must fit behind the barrier");
  // load item i of cell rx: 2 j + a = x fragment of triple j (2: the single), ci half a; 6 + b = the entering dy row, co half b
  auto load_item = [&](const unsigned (&sb)[3], unsigned yb, int rx, int i) {
    if (i < 6) {
      const int j = i >> 1, a = i & 1;
      if (j < 2 || (WV < 3 && rx - WV >= 0 && rx - WV < TH))
        fx[rx & 1][j][a] = frag(sb[T::kd(j)] + a * G::XPLANE + (rx * SW + T::kw(j)) * 32);
    } else if (rx < TH) fy[rx & 3][i - 6] = frag(yb + (i - 6) * G::YPLANE + (rx * TW) * 32);
  };
  // MFMA item i of cell rx: tap i / 4 (local index), block i % 4 = 2 a + b
  auto mfma_item = [&](int rx, int i) {
    const int tp = i >> 2, a = (i >> 1) & 1, bb = i & 1;
    const int r = tp < 6 ? rx - tp % 3 : rx - WV;
    if (tp < T::NT && r >= 0 && r < TH)
      acc[tp][2 * a + bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[rx & 1][tp < 6 ? tp / 3 : 2][a], fy[r & 3][bb],
                                                                   acc[tp][2 * a + bb], 0, 0, 0);
  };
  auto bases = [&](int t, unsigned (&sb)[3], unsigned& yb) {
#pragma unroll
    for (int k = 0; k < 3; ++k) sb[k] = lds_base(lds0 + ((unsigned)(t + k) % G::NXS) * G::XSLOT + lane8);
    yb = lds_base(lds0 + G::NXS * G::XSLOT + ((unsigned)t % G::NYS) * G::YSLOT + lane8);
  };
  auto first_cell = [&](const unsigned (&sb)[3], unsigned yb) {
    load_item(sb, yb, 0, 6); load_item(sb, yb, 0, 7);
#pragma unroll
    for (int i = 0; i < 6; ++i) load_item(sb, yb, 0, i);
  };
  if (G::MB) {
    unsigned sb0[3], yb0;
    bases(0, sb0, yb0);
    first_cell(sb0, yb0);
  }
#pragma unroll 1
  for (int t = 0; t < nd; ++t) {
    unsigned sb[3], yb, sn[3], yn;
    bases(t, sb, yb);
    bases(t + 1, sn, yn);
    const bool more = t + 1 < nd, more2 = t + 2 < nd;
    if (!G::MB) first_cell(sb, yb);
#pragma unroll
    for (int rx = 0; rx < SH; ++rx) {
#pragma unroll
      for (int i = 0; i < 28; ++i) {
        // 8 fragments of the next cell (MB: of the next step's first cell behind the last one) spread over the 28 gaps
        const int li = i == 0 ? 6 : (i == 3 ? 7 : ((i >= 6 && i <= 21 && i % 3 == 0) ? (i - 6) / 3 : -1));
        if (li >= 0) {
          if (rx + 1 < SH) load_item(sb, yb, rx + 1, li);
          else if (G::MB) load_item(sn, yn, 0, li);
        }
        if (!G::MB) {
          if (i == 13 && rx < G::XPW) dma_x(t + 3, rx, more);
          if (i == 25 && rx < G::YPW) dma_y(t + 1, rx, more);
        } else if (rx >= PB) {                            // behind this step's barrier: slab t + 4 and dy(t + 2)
          if ((i == 7 || i == 19) && 2 * (rx - PB) + (i == 19) < G::XPW) dma_x(t + 4, 2 * (rx - PB) + (i == 19), more2);
          if ((i == 13 || i == 25) && 2 * (rx - PB) + (i == 25) < G::YPW) dma_y(t + 2, 2 * (rx - PB) + (i == 25), more2);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_item(rx, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (G::MB && rx == PB - 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!G::MB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // partial tiles [tap][co][ci]: block (a, b) of a lane = ci 16 a + 4 (lane / 16) .. + 3 at co 16 b + lane % 16: one 16-byte store
  const int pair = cot * ncg + cg;
  float* out = part + ((int64_t)bid.x * (ncg * (Cout / 32)) + pair) * (27 * 1024);
#pragma unroll
  for (int i = 0; i < T::NT; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(out + (T::tap(i) * 32 + 16 * (q & 1) + (lane & 15)) * 32 + 16 * (q >> 1) + 4 * (lane >> 4)) =
          make_float4(acc[i][q][0], acc[i][q][1], acc[i][q][2], acc[i][q][3]);
}

template <int TH, bool MB>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_roll16(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
                  float* __restrict__ part, int D, int H, int W, int Cin, int Cout, int tilesH, int tilesW, int dsegs, int dlen,
                  const bf16_t* __restrict__ x1, int xcd) {
  using G = WR16<TH, MB>;
  const FplxBlock bid = fplx_xcd_block(xcd);
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: roll16_march<G, 0>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 1: roll16_march<G, 1>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 2: roll16_march<G, 2>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    default: roll16_march<G, 3>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
  }
}

template <int TH, int TW, bool MB>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_roll(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
                float* __restrict__ part, int D, int H, int W, int Cin, int Cout, int tilesH, int tilesW, int dsegs, int dlen,
                const bf16_t* __restrict__ x1, int xcd) {
  using G = WR<TH, TW, MB>;
  const FplxBlock bid = fplx_xcd_block(xcd);
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {            // wave-uniform: four copies of the march
    case 0: roll_march<G, 0>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 1: roll_march<G, 1>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 2: roll_march<G, 2>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    default: roll_march<G, 3>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
  }
}

// ------------------------------------------------------------------------------------------
// roll2d_march: the weight gradient of a Conv2d per depth slice (the 2.5D levels: dW[kh][kw] sums over ALL depths, a tap
// never crosses a slice).  A depth step needs one x slab and one dy slab and runs 9 taps on them - a third of the 3D form's
// MFMAs on the same bytes, so the kernel is bound by the slab stream, not the matrix pipe: waves 0-2 own the kernel column
// kw = WV (the rolling window over kh: one x fragment per cell, three MFMAs against the last three dy rows), wave 3 only
// fetches.  Three ring slots per operand: the slabs of depth t + 2 are requested during step t, and the step's closing wait
// is COUNTED - every wave issues the same number of pieces per step (a piece index past the end repeats the last piece:
// same bytes, same place), so vmcnt(NPW) leaves exactly the newest slab pair in flight and a fetch has a whole step to land.
// Partial tiles go to the taps 9..17 of the usual [block][pair][27][co][ci] layout (wgrad_stream_reduce, mid form).
template <int TH_, int TW_>
struct WR2 {
  static constexpr int TH = TH_, TW = TW_, SH = TH + 2, SW = TW + 2, SLAB = SH * SW;
  static constexpr int XP = (SLAB * 4 + 63) / 64, XSLOT = XP * 1024;
  static constexpr int YP = TH * TW * 4 / 64, YSLOT = YP * 1024;
  static constexpr int NS = 3;
  static constexpr int LDS = NS * (XSLOT + YSLOT);
  static constexpr int NCH = TW / 16, NCELL = NCH * SH;
  static constexpr int XPW = (XP + 3) / 4, YPW = (YP + 3) / 4, NPW = XPW + YPW;
  static_assert(NPW <= NCELL, "one DMA piece per cell");
};

template <class G, int WV>
__device__ __forceinline__ void roll2d_march(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                             int64_t ldy, float* __restrict__ part, int D, int H, int W, int Cin, int Cout,
                                             int tilesH, int tilesW, int dsegs, int dlen, const bf16_t* __restrict__ x1,
                                             const FplxBlock bid) {
  constexpr int TH = G::TH, TW = G::TW, SH = G::SH, SW = G::SW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int ncg = Cin / 32;
  const int cot = bid.y / ncg, cg = bid.y % ncg;
  const int h0 = th * TH, w0 = tw * TW;
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;

  unsigned xvo[G::XPW], yvo[G::YPW];
#pragma unroll
  for (int k = 0; k < G::XPW; ++k) {
    const int piece = WV + 4 * k < G::XP ? WV + 4 * k : G::XP - 1;
    const int i = piece * 64 + lane;
    const int vox = i >> 2, c = i & 3;
    const int hh = vox / SW + h0 - 1, ww = vox % SW + w0 - 1;
    const bool in = vox < G::SLAB && hh >= 0 && hh < H && ww >= 0 && ww < W;
    xvo[k] = in ? (unsigned)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : 0x40000000u;
  }
#pragma unroll
  for (int k = 0; k < G::YPW; ++k) {
    const int piece = WV + 4 * k < G::YP ? WV + 4 * k : G::YP - 1;
    const int i = piece * 64 + lane;
    const int vox = i >> 2, c = i & 3;
    const int hh = vox / TW + h0, ww = vox % TW + w0;
    yvo[k] = (hh < H && ww < W) ? (unsigned)((((int64_t)hh * W + ww) * ldy + c * 8) * 2) : 0x40000000u;
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2, yslice = (int64_t)H * W * ldy * 2;
  const char* xn = reinterpret_cast<const char*>((x1 && cg == 1) ? x1 : x + cg * 32) + (int64_t)n * D * xslice;
  const char* yn = reinterpret_cast<const char*>(dy + cot * 32) + (int64_t)n * D * yslice;
  u32x4 xr, yr;
  xr[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xn);
  xr[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xn >> 32) & 0xFFFFu);
  xr[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * xslice - (ldx - 32) * 2));
  xr[3] = 0x00020000u;
  yr[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)yn);
  yr[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)yn >> 32) & 0xFFFFu);
  yr[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)D * yslice - (ldy - 32) * 2));
  yr[3] = 0x00020000u;
  const unsigned xslice32 = __builtin_amdgcn_readfirstlane((unsigned)xslice);
  const unsigned yslice32 = __builtin_amdgcn_readfirstlane((unsigned)yslice);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  auto dma = [&](const u32x4& rsrc, unsigned vo, unsigned so, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so), "s"(dst) : "memory");
  };
  const int nd = d1 - d0;
  // piece j of this wave for the slabs of depth d0 + t (ring slot t % 3); off (t >= nd): out of range, the free slot gets zeros
  auto issue = [&](int t, int j) {
    const bool on = t < nd;
    const unsigned slot = (unsigned)t % G::NS;
    if (j < G::XPW) {
      const int piece = WV + 4 * j < G::XP ? WV + 4 * j : G::XP - 1;
      const unsigned so = __builtin_amdgcn_readfirstlane(on ? (unsigned)(d0 + t) * xslice32 : 0x40000000u);
      dma(xr, xvo[j], so, lds0 + slot * G::XSLOT + piece * 1024);
    } else {
      const int jj = j - G::XPW;
      const int piece = WV + 4 * jj < G::YP ? WV + 4 * jj : G::YP - 1;
      const unsigned so = __builtin_amdgcn_readfirstlane(on ? (unsigned)(d0 + t) * yslice32 : 0x40000000u);
      dma(yr, yvo[jj], so, lds0 + G::NS * G::XSLOT + slot * G::YSLOT + piece * 1024);
    }
  };
  auto wait_newest = [&]() {                               // all but the NPW pieces issued last have landed
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G::NPW) : "memory");
    __builtin_amdgcn_s_barrier();
  };

  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;

  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#pragma unroll
  for (int j = 0; j < G::NPW; ++j) issue(0, j);
#pragma unroll
  for (int j = 0; j < G::NPW; ++j) issue(1, j);
  wait_newest();

  bf16x8 fx[2], fy[4];
#pragma unroll 1
  for (int t = 0; t < nd; ++t) {
    const unsigned slot = (unsigned)t % G::NS;
    const unsigned xb = lds_base(lds0 + slot * G::XSLOT + lane_off);
    const unsigned yb = lds_base(lds0 + G::NS * G::XSLOT + slot * G::YSLOT + lane_off);
    if (WV < 3) {
      fy[0] = tr_frag(yb);
      fx[0] = tr_frag(xb + WV * 64);
    }
#pragma unroll
    for (int k = 0; k < G::NCELL; ++k) {
      const int rx = k % SH;
      if (WV < 3 && k + 1 < G::NCELL) {                     // the next cell's fragments behind this cell's MFMAs
        const int c1 = (k + 1) / SH, r1 = (k + 1) % SH;
        if (r1 < TH) fy[r1 & 3] = tr_frag(yb + ((r1 * TW + c1 * 16) * 64));
        fx[(k + 1) & 1] = tr_frag(xb + ((r1 * SW + c1 * 16 + WV) * 64));
      }
      if (k < G::NPW) issue(t + 2, k);
      __builtin_amdgcn_sched_barrier(0);
      if (WV < 3) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int r = rx - kh;
          if (r >= 0 && r < TH) acc[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[k & 1], fy[r & 3], acc[kh], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wait_newest();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (WV < 3) {
    const int co = lane & 31, rbase = (lane >> 5) * 4;
    const int pair = cot * ncg + cg;
    float* out = part + ((int64_t)bid.x * (ncg * (Cout / 32)) + pair) * (27 * 1024);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(out + ((9 + kh * 3 + WV) * 32 + co) * 32 + 8 * g4 + rbase) =
            make_float4(acc[kh][4 * g4 + 0], acc[kh][4 * g4 + 1], acc[kh][4 * g4 + 2], acc[kh][4 * g4 + 3]);
    }
  }
}

template <int TH, int TW>
__global__ void __launch_bounds__(256)
conv_wgrad_roll2d(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
                  float* __restrict__ part, int D, int H, int W, int Cin, int Cout, int tilesH, int tilesW, int dsegs, int dlen,
                  const bf16_t* __restrict__ x1, int xcd) {
  using G = WR2<TH, TW>;
  const FplxBlock bid = fplx_xcd_block(xcd);
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: roll2d_march<G, 0>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 1: roll2d_march<G, 1>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 2: roll2d_march<G, 2>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    default: roll2d_march<G, 3>(x, ldx, dy, ldy, part, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
  }
}

struct RollCfg { int th, tw, tilesH, tilesW, dsegs, dlen, nblk, npairs; size_t ws; };

inline RollCfg roll_cfg(int n, int d, int h, int w, int cin, int cout, bool twod = false) {
  RollCfg c;
  // footprint: the one that pads the (h, w) plane least; 16 x 16 on a tie (fewer slab rows per dy row, smaller halo)
  const int geo = (int)fplx_knob(FPLX_K_WG_ROLL_GEO);
  auto area = [&](int th, int tw) { return (int64_t)((h + th - 1) / th) * th * ((w + tw - 1) / tw) * tw; };
  // on a tie 8 x 32 where it runs on the 16 x 16 x 32 MFMA (less energy per FLOP: in the train step, beside the main stream's
  // kernels, that form measured -0.6 % against 16 x 16 footprints and -1.2 % against 8 x 32 on 32 x 32 x 16), else 16 x 16
  // (fewer slab rows per dy row, smaller halo)
  const bool m16 = !twod && fplx_knob(FPLX_K_WG_ROLL_M16) != 0;          // (the 2D form has no 16 x 16 x 32 kernel)
  c.th = 16; c.tw = 16;
  if (area(8, 32) < area(c.th, c.tw) || (m16 && area(8, 32) == area(c.th, c.tw))) { c.th = 8; c.tw = 32; }
  if (area(8, 16) < area(c.th, c.tw)) { c.th = 8; c.tw = 16; }
  if (geo == 1) { c.th = 8; c.tw = 32; } else if (geo == 2) { c.th = 16; c.tw = 16; } else if (geo == 3) { c.th = 8; c.tw = 16; }
  c.tilesH = (h + c.th - 1) / c.th;
  c.tilesW = (w + c.tw - 1) / c.tw;
  c.npairs = (cin / 32) * (cout / 32);
  const int64_t tiles = (int64_t)n * c.tilesH * c.tilesW * c.npairs;
  // one block per CU at a time: the depth split that minimises rounds x (depths per block + per-block overhead)
  // (a block of the 2D form has no depth halo to fetch: its overhead is the prologue's fetch latency and 9 partial tiles)
  const double ovh = (double)fplx_knob(twod ? FPLX_K_WG_ROLL2D_OVH : FPLX_K_WG_ROLL_OVH);
  const int64_t cus = fplx_knob(FPLX_K_WG_ROLL_CUS);          // CUs the depth split plans for
  int ds = 1;
  double best = 1e30;
  for (int cand = 1; cand <= d; ++cand) {
    const int dl = (d + cand - 1) / cand;
    if (dl < 4 && cand > 1) break;
    const int segs = (d + dl - 1) / dl;
    const int64_t rounds = (tiles * segs + cus - 1) / cus;
    const double cost = (double)rounds * (dl + ovh);
    if (cost < best - 1e-9) { best = cost; ds = segs; }
  }
  {
    const int e = (int)fplx_knob(FPLX_K_WG_DS);
    if (e > 0) ds = e;
  }
  c.dlen = (d + ds - 1) / ds;
  c.dsegs = (d + c.dlen - 1) / c.dlen;
  c.nblk = n * c.tilesH * c.tilesW * c.dsegs;
  c.ws = (size_t)c.nblk * c.npairs * 27 * 1024 * sizeof(float);
  return c;
}

}  // namespace

// 1 if the rolling-window kernel takes the layer (3D, both channel counts multiples of 32, a sample below 1 GiB so that the
// out-of-range marker 0x40000000 cannot alias a voxel)
extern "C" int fplx_wgroll_ok(int n, int d, int h, int w, int cin, int cout, int64_t ldx, int64_t ldy) {
  if (!fplx_knob(FPLX_K_WG_ROLL)) return 0;
  if (cin % 32 != 0 || cout % 32 != 0) return 0;
  if ((int64_t)d * h * w < fplx_knob(FPLX_K_WG_ROLL_MINVOX)) return 0;
  if (h < 8 || w < 16) return 0;
  if ((int64_t)d * h * w * ldx * 2 > ((int64_t)1 << 30) || (int64_t)d * h * w * ldy * 2 > ((int64_t)1 << 30)) return 0;
  return 1;
}

extern "C" size_t fplx_wgroll_ws_bytes(int n, int d, int h, int w, int cin, int cout) {      // the larger of the 3D and the 2D form's
  const size_t a = roll_cfg(n, d, h, w, cin, cout).ws, b = roll_cfg(n, d, h, w, cin, cout, true).ws;
  return a > b ? a : b;
}

// returns 1 if launched, 0 if not applicable, < 0 on error.  dw fp32 [Cout][Cin][27]; x1: second ci tile of a split Cin = 64
// mid != 0: a Conv2d per depth slice (conv_wgrad_roll2d), dw fp32 [Cout][Cin][3][3]
extern "C" int fplx_wgroll_conv3d_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n, int d, int h,
                                        int w, int cin, int cout, void* ws, size_t ws_bytes, hipStream_t st, const void* x1,
                                        int mid) {
  if (!fplx_wgroll_ok(n, d, h, w, cin, cout, ldx, ldy)) return 0;
  if (mid && !fplx_knob(FPLX_K_WG_ROLL2D)) return 0;
  if (ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)x1 % 16)) return 0;
  if (x1 && cin != 64) return 0;
  const RollCfg c = roll_cfg(n, d, h, w, cin, cout, mid != 0);
  if (ws_bytes < c.ws) return fplx_fail(FPLX_E_WORKSPACE, "wgroll_conv3d_wgrad: workspace %zu < %zu", ws_bytes, c.ws);
  dim3 grid(c.nblk, c.npairs);
  if (mid) {
#define LAUNCH_ROLL2D(TH_, TW_)                                                                                     \
  do {                                                                                                              \
    using G_ = WR2<TH_, TW_>;                                                                                       \
    (void)hipFuncSetAttribute((const void*)conv_wgrad_roll2d<TH_, TW_>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
    conv_wgrad_roll2d<TH_, TW_><<<grid, 256, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, d, h, w, cin, \
                                                           cout, c.tilesH, c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on()); \
  } while (0)
    if (c.th == 16) LAUNCH_ROLL2D(16, 16); else if (c.tw == 32) LAUNCH_ROLL2D(8, 32); else LAUNCH_ROLL2D(8, 16);
#undef LAUNCH_ROLL2D
    int rc2 = fplx_check_launch("wgroll_conv2d_wgrad");
    if (rc2 < 0) return rc2;
    rc2 = fplx_wgrad_reduce_launch((const float*)ws, c.nblk, c.npairs, cin, cout, dw, 1, st);
    return rc2 < 0 ? rc2 : 1;
  }
#define LAUNCH_ROLL(TH_, TW_, MB_)                                                                                   \
  do {                                                                                                              \
    using G_ = WR<TH_, TW_, MB_>;                                                                                   \
    (void)hipFuncSetAttribute((const void*)conv_wgrad_roll<TH_, TW_, MB_>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
    conv_wgrad_roll<TH_, TW_, MB_><<<grid, 256, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, d, h, w, \
                                                              cin, cout, c.tilesH, c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, \
                                                              fplx_xcd_on());                                       \
  } while (0)
#define LAUNCH_ROLL16(MB_)                                                                                          \
  do {                                                                                                              \
    using G_ = WR16<8, MB_>;                                                                                        \
    (void)hipFuncSetAttribute((const void*)conv_wgrad_roll16<8, MB_>, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS); \
    conv_wgrad_roll16<8, MB_><<<grid, 256, G_::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, d, h, w, cin, \
                                                         cout, c.tilesH, c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on()); \
  } while (0)
  // mb: the mid-step barrier form (5 + 3 ring slots); the 8 x 16 footprint keeps the 4 + 2 rings (two blocks per CU) unless mb = 2
  const int mb = (int)fplx_knob(FPLX_K_WG_ROLL_MB);
  if (c.th == 8 && c.tw == 32 && fplx_knob(FPLX_K_WG_ROLL_M16)) { if (mb) LAUNCH_ROLL16(true); else LAUNCH_ROLL16(false); }
  else if (c.th == 16) { if (mb) LAUNCH_ROLL(16, 16, true); else LAUNCH_ROLL(16, 16, false); }
  else if (c.tw == 32) { if (mb) LAUNCH_ROLL(8, 32, true); else LAUNCH_ROLL(8, 32, false); }
  else { if (mb == 2) LAUNCH_ROLL(8, 16, true); else LAUNCH_ROLL(8, 16, false); }
#undef LAUNCH_ROLL16
#undef LAUNCH_ROLL
  int rc = fplx_check_launch("wgroll_conv3d_wgrad");
  if (rc < 0) return rc;
  rc = fplx_wgrad_reduce_launch((const float*)ws, c.nblk, c.npairs, cin, cout, dw, 0, st);
  return rc < 0 ? rc : 1;
}
