// The two "edge" convolutions of the U-Net, both purely HBM-bound (AI 26 and 17):
//   stem      Conv3d(in_chns -> C0, 3x3x3) reading the fp32 NCDHW network input   (unet2d5_dsbn.py:54,75)
//   out_conv  Conv3d(C0 -> class_num, 1x3x3) writing fp32 NCDHW logits            (unet2d5_dsbn.py:293-294,307)
// They still run on the matrix cores (padding K / N to the 32x32x16 tile) because the VALU
// formulation is ~10x over the HBM time; with MFMA all five kernels sit at the memory roof.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int TH = 8, TW = 32, SH = TH + 2, SW = TW + 2;

__device__ __forceinline__ bf16x8 tr_frag64(const char* base_lo) {      // [voxel][32 ch] image, 64 B rows
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base_lo));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base_lo + 4 * 64));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

struct Tile { int n, d, h0, w0; };
// 32-bit index math (2^31 tiles would be 5e11 voxels): a 64-bit division is a ~100-instruction loop
__device__ __forceinline__ Tile tile_of(int64_t t64, int D, int tilesH, int tilesW) {
  unsigned t = (unsigned)t64;
  Tile o;
  o.w0 = (int)(t % tilesW) * TW; t /= tilesW;
  o.h0 = (int)(t % tilesH) * TH; t /= tilesH;
  o.d = (int)(t % D); t /= D;
  o.n = (int)t;
  return o;
}

// stage the fp32 planar input around a tile as bf16: xs[ci][kd 3][SH][SW]
template <int CIN>
__device__ __forceinline__ void stage_x_planar(const float* __restrict__ x, bf16_t* xs, const Tile& t, int D, int H,
                                               int W) {
  for (int i = threadIdx.x; i < CIN * 3 * SH * SW; i += 256) {
    const int ww = i % SW, hh = (i / SW) % SH, kd = (i / (SW * SH)) % 3, ci = i / (3 * SH * SW);
    const int d = t.d + kd - 1, h = t.h0 + hh - 1, w = t.w0 + ww - 1;
    float v = 0.f;
    if (d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W)
      v = x[((((int64_t)t.n * CIN + ci) * D + d) * H + h) * W + w];
    xs[i] = (bf16_t)v;
  }
}

// ------------------------------------------------------------------------------------------
// register prefetch of the fp32 planar input around a tile (bf16 image xs[ci][kd 3][SH][SW] after the commit):
// unconditional loads from clamped coordinates, zeroed at the commit (see outconv_fwd_mfma)
template <int CIN>
struct PlanarFetch {
  static constexpr int TOT = CIN * 3 * SH * SW, NLD = (TOT + 255) / 256;
  float reg[NLD];
  unsigned zmask;
  __device__ __forceinline__ void fetch(const float* __restrict__ x, const Tile& t, int D, int H, int W) {
    zmask = 0;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int ww = i % SW, hh = (i / SW) % SH, kd = (i / (SW * SH)) % 3, ci = (i / (3 * SH * SW)) % CIN;
      const int d = t.d + kd - 1, h = t.h0 + hh - 1, w = t.w0 + ww - 1;
      const bool in = i < TOT && d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W;
      const int dc = d < 0 ? 0 : (d >= D ? D - 1 : d), hc = h < 0 ? 0 : (h >= H ? H - 1 : h),
                wc = w < 0 ? 0 : (w >= W ? W - 1 : w);
      reg[k] = x[((((int64_t)t.n * CIN + ci) * D + dc) * H + hc) * W + wc];
      if (in) zmask |= 1u << k;
    }
  }
  __device__ __forceinline__ void commit(bf16_t* xs) const {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < TOT) xs[i] = (bf16_t)(((zmask >> k) & 1u) ? reg[k] : 0.f);
    }
  }
};

// ------------------------------------------------------------------------------------------
// stem forward: rows = voxels, K = (ci, tap) padded to 16*KS, cols = 32 output channels.  The next tile's input is
// requested before the MFMAs of the current one; the 32 x 32 result tile of a wave goes through a 2-KB LDS transpose and
// leaves as 16-byte stores (2-byte global stores are issue-bound: they were most of this kernel's time).
template <int CIN>
__global__ void __launch_bounds__(256)
stem_fwd_mfma(const float* __restrict__ x, const bf16_t* __restrict__ wf, const float* __restrict__ bias,
              bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int co0, int Cout,
              float* __restrict__ stats, int64_t ntiles, int tilesH, int tilesW, int vec_ok, int xcd) {
  constexpr int KTOT = 27 * CIN, KS = (KTOT + 15) / 16;
  __shared__ bf16_t xs[CIN * 3 * SH * SW];
  __shared__ float red[4][2][32];
  __shared__ __attribute__((aligned(16))) char stg_all[4][32 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, khalf = lane >> 5;
  // per-lane K geometry: element offsets of this lane's 8 k-values per k-step, and the B fragments
  int koff[KS][8];
  bf16x8 bfr[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * khalf + j;
      const int ci = k / 27, tap = k % 27;
      const bool ok = k < KTOT;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      koff[s][j] = ok ? ((ci * 3 + kd) * SH + kh) * SW + kw : -1;
      bfr[s][j] = ok ? wf[((int64_t)tap * Cout + co0 + r) * CIN + ci] : (bf16_t)0.f;
    }
  const int co = co0 + r;
  const float bv = bias ? bias[co] : 0.f;
  float ssum = 0.f, qsum = 0.f;
  char* stg = stg_all[wave];
  PlanarFetch<CIN> pf;
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  int64_t tt = tr.first;
  Tile tn = tile_of(tt < tr.end ? tt : 0, D, tilesH, tilesW);
  if (tt < tr.end) pf.fetch(x, tn, D, H, W);
  for (; tt < tr.end; tt += tr.step) {
    const Tile t = tn;
    __syncthreads();
    pf.commit(xs);
    __syncthreads();
    if (tt + tr.step < tr.end) {
      tn = tile_of(tt + tr.step, D, tilesH, tilesW);
      pf.fetch(x, tn, D, H, W);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int hr = wave * 2 + m;                     // tile row handled by this wave
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int base = hr * SW + r;                    // lane's voxel (row hr, column r) in slab coordinates
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = koff[s][j] >= 0 ? xs[base + koff[s][j]] : (bf16_t)0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[s], acc, 0, 0, 0);
      }
      const int h = t.h0 + hr;
      // accumulator element i = voxel column wu of this row, channel r  ->  stg[wu][r] (64-byte rows)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int wu = (i & 3) + 8 * (i >> 2) + 4 * khalf;
        const float o = acc[i] + bv;
        *reinterpret_cast<bf16_t*>(stg + wu * 64 + r * 2) = (bf16_t)o;
        if (h < H && t.w0 + wu < W) {
          ssum += o;
          qsum = fmaf(o, o, qsum);
        }
      }
      if (h < H) {                                     // wave-uniform
        const int64_t vrow = (((int64_t)t.n * D + t.d) * H + h) * W + t.w0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int wu = (lane >> 2) + 16 * half;
          const uint4 v = *reinterpret_cast<const uint4*>(stg + wu * 64 + (lane & 3) * 16);
          if (t.w0 + wu < W) {
            bf16_t* dst = y + (vrow + wu) * ldy + co0 + (lane & 3) * 8;
            if (vec_ok) *reinterpret_cast<uint4*>(dst) = v;
            else {                                     // unaligned rows (a channel slice of a wider buffer)
              const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
              for (int j = 0; j < 8; ++j) dst[j] = e[j];
            }
          }
        }
      }
    }
  }
  if (stats) {
    const float a = ssum + __shfl_xor(ssum, 32, 64), b = qsum + __shfl_xor(qsum, 32, 64);
    if (lane < 32) { red[wave][0][r] = a; red[wave][1][r] = b; }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      stats[((int64_t)blockIdx.x * 2 + which) * Cout + co0 + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
  }
}

// ------------------------------------------------------------------------------------------
// stem_fwd_rows (in_chns = 1, round 3): the same convolution without LDS, barriers or tiles.  The kernel above spends 7 us
// per 256-voxel tile on a chain of (commit -> barrier -> 16 two-byte LDS gathers per fragment -> MFMA -> 16 two-byte LDS
// stores -> 16-byte global stores), 1.2 TB/s for an operation that only has to WRITE 131 MB.  Here a wave owns 32
// consecutive voxels of one row at a time and computes the TRANSPOSED product D[cout][voxel] = W[cout][k] X^T[k][voxel]:
//   * B operand (lane = voxel r, k-octet = lane / 32): the lane's 16 input values come straight from global memory (fp32
//     planar, 27-fold reuse served by L1 / L2) through a buffer descriptor - out-of-tensor addresses read as zeros, the
//     in-tensor wrap-arounds of the zero padding (d +- 1, h +- 1, w +- 1 outside the volume) are cleared by selects;
//   * the 27 taps are dealt to the 32 k-slots so that slot (s, j) has the SAME kw in both lane halves (p = 8 s + j: kw =
//     p % 3, (kd, kh) = p / 3 + 5 (lane / 32)): kw is an immediate of the load, the (kd, kh) row is one of 6 lane
//     offsets, validity of a row is one bit of a per-tile scalar mask;
//   * A operand = the weights in the same slot order, two fragments for the whole kernel; the bias is the MFMA's C input;
//   * a lane ends up with 16 output channels of ONE voxel; two v_permlane32_swap exchanges between the lane halves turn
//     them into two runs of 8 consecutive channels = two 16-byte stores per lane, whole 64-byte voxel rows per wave;
//   * BatchNorm statistics: 16 + 16 running sums per lane, reduced once at the end of the kernel.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2e;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4e;

// CIN = 4 (round 4, config 5): the same two MFMAs per input channel - the planar input makes a channel just another base
// offset, the tap -> k-slot map and the padding selects are shared - 64 loads in flight per lane, two blocks per CU.
template <int CIN>
__global__ void __launch_bounds__(256, CIN == 1 ? 4 : 2)
stem_fwd_rows(const float* __restrict__ x, const bf16_t* __restrict__ wf, const float* __restrict__ bias,
              bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int co0, int Cout,
              float* __restrict__ stats, int tilesW, int64_t ntiles, int xcd) {
  __shared__ float red[4][2][32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the segment walk below is scalar work
  const int r = lane & 31, khalf = lane >> 5;
  // k-slot p = 8 s + j of lane half khalf: combo = p / 3 + 5 khalf (kd = combo / 3, kh = combo % 3), kw = p % 3
  // (low half: slots 0-14 = combos 0-4, slot 15 spare; high half: slots 0-11 = combos 5-8, slots 12-15 spare)
  auto slot_tap = [&](int p, int kh_) { return p < (kh_ ? 12 : 15) ? (p / 3 + 5 * kh_) * 3 + p % 3 : -1; };
  bf16x8 afr[CIN][2];                                // A: row = output channel r, the lane's 8 k-slots of step s, per input channel
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tap = slot_tap(8 * s_ + j, khalf);
        afr[ci][s_][j] = tap >= 0 ? wf[((int64_t)tap * Cout + co0 + r) * CIN + ci] : (bf16_t)0.f;      // pack [tap][Cout][Cin]
      }
  f32x16 cinit;                                      // D[row = channel (i & 3) + 8 (i >> 2) + 4 khalf][col = voxel r]
#pragma unroll
  for (int i = 0; i < 16; ++i) cinit[i] = bias ? bias[co0 + (i & 3) + 8 * (i >> 2) + 4 * khalf] : 0.f;
  // the lane's six (kd, kh) rows as BYTE offsets of their centre tap relative to the segment's first voxel (combo c + 5 khalf;
  // spare rows repeat the centre row: their k-slots meet zero weights)
  int rowb[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int cb = c + 5 * khalf;
    rowb[c] = ((cb < 9 ? ((cb / 3 - 1) * H + (cb % 3 - 1)) * W : 0) + r) * 4;
  }
  const int64_t xbytes = (int64_t)N * CIN * D * H * W * 4;
  // (the launcher keeps the input below 2 GiB; raw buffer, no stride: out-of-range offsets read as zeros)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)xbytes, 0x00020000);
  float ssum[16], qsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { ssum[i] = 0.f; qsum[i] = 0.f; }

  // Round 6: the segment walk in SCALAR registers.  The round-3 form derived (n, d, h, w0) of every segment from a lane-indexed
  // wave id - three 32-bit divisions and every `segment exists` test as per-lane code, an exec-mask branch around each of the 16
  // loads - and formed 16 load addresses with three VALU operations each: about 330 VALU instructions per 32-voxel segment
  // against two MFMAs, which - not the 262-MB write - set this kernel's time (profiles/r06_kernel_ab.txt section 1: the
  // store-free build still took 62 of 93 us).  Now: coordinates advance by a precomputed mixed-radix step with carries
  // (no division in the loop), a segment's loads are six vector adds (row offset + scalar segment base; kw = 0 one more each,
  // kw = 2 the instruction's immediate offset), and segments whose 3 x 3 x 3 neighbourhoods lie inside the volume for every
  // lane skip the sixteen padding selects.
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);   // tiles = 4 consecutive 32-voxel segments (one per wave)
  auto sread = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  const unsigned uW = (unsigned)tilesW, uH = (unsigned)H, uD = (unsigned)D;
  unsigned sw_, sh_, sd_, sn_;                      // 4 step as (w tile, h, d, n) digits
  struct Seg { unsigned cw, ch, cd, cn; };
  Seg cur;
  {
    unsigned t = sread((unsigned)tr.step * 4u);
    sw_ = sread(t % uW); t = sread(t / uW);
    sh_ = sread(t % uH); t = sread(t / uH);
    sd_ = sread(t % uD); sn_ = sread(t / uD);
    t = sread((unsigned)tr.first * 4u + (unsigned)wave);
    cur.cw = sread(t % uW); t = sread(t / uW);
    cur.ch = sread(t % uH); t = sread(t / uH);
    cur.cd = sread(t % uD); cur.cn = sread(t / uD);  // >= N: a segment past the end of the list (the last tile may be ragged)
  }
  auto advance = [&](Seg& g) {                        // += 4 step in (w tile, h, d, n) digits, carries as scalar selects
    g.cw += sw_;
    unsigned carry = g.cw >= uW ? 1u : 0u;
    g.cw -= carry ? uW : 0u;
    g.ch += sh_ + carry;
    carry = g.ch >= uH ? 1u : 0u;
    g.ch -= carry ? uH : 0u;
    g.cd += sd_ + carry;
    carry = g.cd >= uD ? 1u : 0u;
    g.cd -= carry ? uD : 0u;
    g.cn += sn_ + carry;
  };
  // ---- the segment's 16 loads per lane and input channel.  A row that starts before the tensor has a negative = huge
  // vector offset: out of range -> 0.  (The kw = 0 tap is a vector offset of its own, not a negative immediate: the range
  // check sees vector offset + immediate, and element -1 of the tensor must not turn into a valid address.)
  auto issue = [&](const Seg& g, float (&xv)[CIN][16]) {
    if (g.cn >= (unsigned)N) return;
    const int n = (int)g.cn, d = (int)g.cd, h = (int)g.ch, w0 = (int)g.cw * 32;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const int sbase = ((((n * CIN + ci) * D + d) * H + h) * W + w0) * 4;          // scalar byte offset of the segment's first voxel
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const unsigned oc = (unsigned)(rowb[c] + sbase), ol = oc - 4u;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int p = 3 * c + kw;
          if (p < 16)
            xv[ci][p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, kw == 0 ? ol : (kw == 1 ? oc : oc + 4u), 0, 0));
        }
      }
    }
  };
  auto consume = [&](const Seg& g, const float (&xv)[CIN][16]) {
    if (g.cn >= (unsigned)N) return;
    const int n = (int)g.cn, d = (int)g.cd, h = (int)g.ch, w0 = (int)g.cw * 32;
    // ---- padding.  interior: every tap of every lane lies inside the volume (scalar test) - no selects at all
    const bool interior = d >= 1 && d + 1 < D && h >= 1 && h + 1 < H && w0 >= 1 && w0 + 32 < W;
    const int wv = w0 + r;
    const bool wc = wv < W;
    f32x16 acc = cinit;
    if (interior) {
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        bf16x8 bfr[2];
#pragma unroll
        for (int p = 0; p < 16; ++p) bfr[p >> 3][p & 7] = (bf16_t)xv[ci][p];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ci][0], bfr[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ci][1], bfr[1], acc, 0, 0, 0);
      }
    } else {
      // rows of the zero padding: bit c of `dh` = combo c lies inside the volume (scalar); the lane's view starts at 5 khalf
      unsigned dh = 0;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const int dd = d + c / 3 - 1, hh = h + c % 3 - 1;
        if (dd >= 0 && dd < D && hh >= 0 && hh < H) dh |= 1u << c;
      }
      const unsigned dhl = dh >> (5 * khalf);
      const bool wl = wv - 1 >= 0 && wv - 1 < W, wr = wv + 1 < W;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        bf16x8 bfr[2];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const int c = p / 3, kw = p % 3;
          const bool ok = ((dhl >> c) & 1u) && (kw == 0 ? wl : (kw == 1 ? wc : wr)) && p < (khalf ? 12 : 15);
          bfr[p >> 3][p & 7] = (bf16_t)(ok ? xv[ci][p] : 0.f);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ci][0], bfr[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ci][1], bfr[1], acc, 0, 0, 0);
      }
    }
    if (wc) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { ssum[i] += acc[i]; qsum[i] = fmaf(acc[i], acc[i], qsum[i]); }
    }
    // quads q = i >> 2 hold channels 8 q + 4 khalf + (0..3); after the swaps the low half holds 0-7 and 16-23, the high
    // half 8-15 and 24-31 of its voxel
    unsigned pk[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16_t e0 = (bf16_t)acc[4 * q], e1 = (bf16_t)acc[4 * q + 1], e2 = (bf16_t)acc[4 * q + 2], e3 = (bf16_t)acc[4 * q + 3];
      pk[2 * q] = (unsigned)__builtin_bit_cast(unsigned short, e0) | ((unsigned)__builtin_bit_cast(unsigned short, e1) << 16);
      pk[2 * q + 1] = (unsigned)__builtin_bit_cast(unsigned short, e2) | ((unsigned)__builtin_bit_cast(unsigned short, e3) << 16);
    }
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)                    // (q0, q1) and (q2, q3): high half of the first <-> low half of the second
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const u32x2e sw = __builtin_amdgcn_permlane32_swap(pk[4 * g2 + e], pk[4 * g2 + 2 + e], false, false);
        pk[4 * g2 + e] = sw[0];
        pk[4 * g2 + 2 + e] = sw[1];
      }
    if (wc) {
      bf16_t* dst = y + ((int64_t)(((n * D + d) * H + h)) * W + wv) * ldy + co0 + 8 * khalf;
      *reinterpret_cast<u32x4e*>(dst) = u32x4e{pk[0], pk[1], pk[2], pk[3]};
      *reinterpret_cast<u32x4e*>(dst + 16) = u32x4e{pk[4], pk[5], pk[6], pk[7]};
    }
  };
  // (The loads of segment t + 1 issued before segment t is computed - 128 registers, still four waves per SIMD - were measured
  // again in round 6: 88.5 / 91.0 us against 90.5 / 89.7 us, no change; profiles/r06_kernel_ab.txt section 4.)
  float xa[CIN][16];
  for (int64_t tt = tr.first; tt < tr.end; tt += tr.step) {
    issue(cur, xa);
    consume(cur, xa);
    advance(cur);
  }
  if (stats) {
    // a lane's sums belong to channels (i & 3) + 8 (i >> 2) + 4 khalf: fold the 32 voxel lanes of each half, then the waves
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { ssum[i] += __shfl_xor(ssum[i], o, 64); qsum[i] += __shfl_xor(qsum[i], o, 64); }
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ch = (i & 3) + 8 * (i >> 2) + 4 * khalf;
        red[wave][0][ch] = ssum[i];
        red[wave][1][ch] = qsum[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      stats[((int64_t)blockIdx.x * 2 + which) * Cout + co0 + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
  }
}

// ------------------------------------------------------------------------------------------
// stem weight gradient: rows = (ci, tap) padded to 32*RT, cols = 32 output channels, K = voxels
// BN (round 6): the stem site's dy - the gradient w.r.t. the convolution's output - has ONE consumer, this kernel (the network
// input needs no data gradient), so the apply pass of the site's BatchNorm + PReLU backward (fplx_bn_act_bwd_apply: read y, read
// d(a), write dy - 786 MB at the benchmark shape) only feeds the 262 MB this kernel reads back.  With BN the kernel takes y and
// d(a) instead and forms dy = scale (dz - k0 - x-hat k1) -> bf16 on the 16-byte pieces it stages (a thread always stages channel
// chunk threadIdx.x & 3: its eight channels' constants sit in registers): the arithmetic of bn_act_bwd_apply_g_k on the same
// values, rounded where the stored tensor was, so dw is bit for bit the two-pass result - and dy is never written or re-read.
struct StemBnBwd {
  const bf16_t* y;                                     // the site's convolution output (pre-BatchNorm), [V][ldy]
  int64_t ldy;
  const float *mean, *rstd, *scale, *shift, *slope, *coef;      // coef: fplx_bn_act_bwd_finalize's [2][Cout]
  int cout;
};
template <int CIN, bool BN = false>
__global__ void __launch_bounds__(256)
stem_wgrad_mfma(const float* __restrict__ x, const bf16_t* __restrict__ dy, int64_t ldy, float* __restrict__ part,
                int N, int D, int H, int W, int co0, int64_t ntiles, int tilesH, int tilesW, int xcd, const StemBnBwd bn = StemBnBwd{}) {
  constexpr int KTOT = 27 * CIN, RT = (KTOT + 31) / 32;
  float bsc[BN ? 8 : 1], bsh[BN ? 8 : 1], bm[BN ? 8 : 1], brs[BN ? 8 : 1], bk0[BN ? 8 : 1], bk1[BN ? 8 : 1], bslope = 0.f;
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = co0 + (threadIdx.x & 3) * 8 + j;
      bsc[j] = bn.scale[c]; bsh[j] = bn.shift[c]; bm[j] = bn.mean[c]; brs[j] = bn.rstd[c];
      bk0[j] = bn.coef[c]; bk1[j] = bn.coef[bn.cout + c];
    }
    bslope = *bn.slope;
  }
  __shared__ bf16_t xs[CIN * 3 * SH * SW + 16];
  __shared__ __attribute__((aligned(16))) char dys[TH * TW * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, khalf = lane >> 5;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;
  int roff[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int k = rt * 32 + r;
    const int ci = k / 27, tap = k % 27;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    roff[rt] = k < KTOT ? ((ci * 3 + kd) * SH + kh) * SW + kw : -1;
  }
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  for (int64_t tt = tr.first; tt < tr.end; tt += tr.step) {
    const Tile t = tile_of(tt, D, tilesH, tilesW);
    __syncthreads();
    stage_x_planar<CIN>(x, xs, t, D, H, W);
    for (int i = threadIdx.x; i < TH * TW * 4; i += 256) {
      const int vox = i >> 2, ch = i & 3;
      const int h = t.h0 + vox / TW, w = t.w0 + vox % TW;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (h < H && w < W) {
        const int64_t vx = (((int64_t)t.n * D + t.d) * H + h) * W + w;
        v = *reinterpret_cast<const uint4*>(dy + vx * ldy + co0 + ch * 8);
        if constexpr (BN) {                            // v = d(a): the apply pass of the site's BatchNorm + PReLU backward, in place
          const uint4 yv4 = *reinterpret_cast<const uint4*>(bn.y + vx * bn.ldy + co0 + ch * 8);
          bf16x8 d8 = *reinterpret_cast<const bf16x8*>(&v);
          const bf16x8 y8 = *reinterpret_cast<const bf16x8*>(&yv4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = (float)y8[j], dd = (float)d8[j];
            const float z = fmaf(a, bsc[j], bsh[j]);
            const float dz = z > 0.f ? dd : dd * bslope;
            const float xh = (a - bm[j]) * brs[j];
            d8[j] = (bf16_t)(bsc[j] * (dz - bk0[j] - xh * bk1[j]));
          }
          v = *reinterpret_cast<uint4*>(&d8);
        }
      }
      *reinterpret_cast<uint4*>(dys + vox * 64 + ch * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {                   // each wave takes 4 of the 16 k-steps (K split)
      const int ks = wave * 4 + kk;
      const int hr = ks >> 1, ws = (ks & 1) * 16;
      const bf16x8 bfrag = tr_frag64(dys + (hr * TW + ws) * 64 + lane_off);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bf16x8 a;
        const int base = hr * SW + ws + 8 * khalf + (roff[rt] >= 0 ? roff[rt] : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = roff[rt] >= 0 ? xs[base + j] : (bf16_t)0.f;
        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag, acc[rt], 0, 0, 0);
      }
    }
  }
  // combine the 4 waves' K-slices through LDS (fixed order), then part[block][rt][row 32][co 32]
  __syncthreads();
  float* red = reinterpret_cast<float*>(dys);            // 4 x 1024 floats = 16 KB
  float* out = part + (int64_t)blockIdx.x * (RT * 1024);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * khalf;
      red[wave * 1024 + row * 32 + r] = acc[rt][i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[rt * 1024 + i] = red[i] + red[1024 + i] + red[2048 + i] + red[3072 + i];
    __syncthreads();
  }
}

// sum of `nparts` partial arrays of `total` floats each: 16 outputs x 16 partial-lanes per block (the arrays are
// small and many: short dependent chains and enough blocks matter more than wide rows), 4 loads in flight per lane,
// lanes combined in a fixed order
constexpr int SP_OUT = 16;
__device__ __forceinline__ float sum_partials(const float* __restrict__ part, int nparts, int64_t total, int64_t i,
                                              float* red) {
  const int pl = threadIdx.x >> 4, o = threadIdx.x & 15;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (i < total) {
    int b = pl;
    for (; b + 48 < nparts; b += 64) {
      t0 += part[(int64_t)b * total + i];
      t1 += part[(int64_t)(b + 16) * total + i];
      t2 += part[(int64_t)(b + 32) * total + i];
      t3 += part[(int64_t)(b + 48) * total + i];
    }
    for (; b < nparts; b += 16) t0 += part[(int64_t)b * total + i];
  }
  red[pl * 16 + o] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k * 16 + o];
  return t;
}

// dw[co][ci][tap] = sum over block partials
__global__ void __launch_bounds__(256)
stem_wgrad_reduce(const float* __restrict__ part, int nparts, int rt, int cin, int co0, float* __restrict__ dw) {
  __shared__ float red[256];
  const int total = rt * 1024;
  const int i = blockIdx.x * SP_OUT + (threadIdx.x & (SP_OUT - 1));
  const float t = sum_partials(part, nparts, total, i, red);
  if (threadIdx.x >= SP_OUT || i >= total) return;
  const int co = i & 31, k = i >> 5;
  if (k >= 27 * cin) return;
  const int ci = k / 27, tap = k % 27;
  dw[((int64_t)(co0 + co) * cin + ci) * 27 + tap] = t;
}

// ------------------------------------------------------------------------------------------
// out_conv forward: one 8 x 32 in-plane tile at a time (grid-strided).  The x tile with its in-plane halo is
// staged once in LDS (64-byte-chunk rows, XOR swizzle for CIN = 32) and feeds all 9 taps; rows = voxels,
// K = 9 taps x Cin, cols = classes padded to 32 (B fragments built once from the fp32 pack [9][ncls][Cin]).
template <int KSTEPS>      // Cin / 16
__global__ void __launch_bounds__(256)
outconv_fwd_mfma(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ wf, const float* __restrict__ bias,
                 float* __restrict__ out, int N, int D, int H, int W, int ncls, int64_t ntiles, int tilesH, int tilesW, int xcd) {
  constexpr int CIN = KSTEPS * 16, ROWB = CIN * 2, CH = ROWB / 16;
  __shared__ __attribute__((aligned(16))) char xs[SH * SW * ROWB];
  __shared__ bf16x8 bsh[9 * KSTEPS][64];            // B fragments, lane-linear
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, khalf = lane >> 5, kh8 = khalf * 8;
  const int64_t Vs = (int64_t)D * H * W;
  for (int f = wave; f < 9 * KSTEPS; f += 4) {
    const int tap = f / KSTEPS, s = f % KSTEPS;
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      b[j] = r < ncls ? (bf16_t)wf[((int64_t)tap * ncls + r) * CIN + s * 16 + kh8 + j] : (bf16_t)0.f;
    bsh[f][lane] = b;
  }
  const float bv = (r < ncls && bias) ? bias[r] : 0.f;
  auto swz = [](int vox) { return (vox / (16 / CH)) % CH; };
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  // the next tile's rows travel global -> registers while this tile computes (a synchronous fill left every block idle for
  // a memory round trip per tile: 143 us at the benchmark shape, 2.1 TB/s)
  constexpr int NLD = (SH * SW * CH + 255) / 256;
  uint4 xreg[NLD];
  auto fetch = [&](const Tile& t) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int vox = i / CH, c = i % CH;
      const int h = t.h0 + vox / SW - 1, w = t.w0 + vox % SW - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < SH * SW * CH && h >= 0 && h < H && w >= 0 && w < W)
        v = *reinterpret_cast<const uint4*>(x + ((((int64_t)t.n * D + t.d) * H + h) * W + w) * ldx + c * 8);
      xreg[k] = v;
    }
  };
  int64_t tt = tr.first;
  Tile tn = tile_of(tt < tr.end ? tt : 0, D, tilesH, tilesW);
  if (tt < tr.end) fetch(tn);
  for (; tt < tr.end; tt += tr.step) {
    const Tile t = tn;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int vox = i / CH, c = i % CH;
      if (i < SH * SW * CH) *reinterpret_cast<uint4*>(xs + vox * ROWB + ((c ^ swz(vox)) * 16)) = xreg[k];
    }
    __syncthreads();
    if (tt + tr.step < tr.end) {
      tn = tile_of(tt + tr.step, D, tilesH, tilesW);
      fetch(tn);
    }
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const bf16x8 b = bsh[tap * KSTEPS + s][lane];
        const int c = 2 * s + khalf;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int vox = (wave * 2 + m + kh) * SW + r + kw;
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(xs + vox * ROWB + ((c ^ swz(vox)) * 16));
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        }
      }
    }
    if (r < ncls) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int h = t.h0 + wave * 2 + m;
        if (h < H) {
          float* orow = out + ((int64_t)t.n * ncls + r) * Vs + ((int64_t)t.d * H + h) * W;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int w = t.w0 + (i & 3) + 8 * (i >> 2) + 4 * khalf;
            if (w < W) orow[w] = acc[m][i] + bv;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// outconv_fwd_t (round 3, Cin = 32 | 64): the same tile pipeline with the product TRANSPOSED and on the 16 x 16 x 32 shape:
// D[class][voxel] = W[class][k] X^T[k][voxel].  The kernel above pads 2 classes to the 32 columns of a 32 x 32 tile (its matrix
// pipe is busy a third of the time computing zeros), re-reads its B fragments from LDS for every tile and writes the logits
// as 4-byte scattered stores from four lanes.  Here the classes are the 16 ROWS of the tile (half the padding, one MFMA per
// tap spans all 32 channels), the weights are the A operand and live in registers for the whole kernel (9 fragments), the only
// LDS traffic of a tile is one 16-byte read per (tap, 16 voxels), and the result comes out voxel-contiguous: lanes 0-15 hold
// class 0 / 1 (registers 0 / 1) of 16 consecutive voxels = 64-byte stores into the planar logits.
typedef __attribute__((ext_vector_type(4))) float f32x4e;
// BN (round 4, Cin = 32): the input is the PRE-BatchNorm output y of the network's last 3x3x3 convolution; the BatchNorm
// apply + PReLU pass of that site (fplx_bn_act_fwd, dropout-free) happens HERE, on the staged registers on their way into LDS,
// and the activation a = PReLU(scale y + shift) - the tensor backward needs - is written out for the tile's own voxels: the
// separate pass's read of y and this kernel's read of a (262 MB at the benchmark shape) become one.  Same arithmetic on the same
// values: a and the logits are the bits of the two-kernel path.
template <int KS32, bool BN = false>      // Cin / 32
__global__ void __launch_bounds__(256, KS32 == 1 ? (BN ? 2 : 3) : 1)     // Cin = 32: three blocks per CU (the per-tile chain is latency, occupancy hides it); two with the BatchNorm constants in registers
outconv_fwd_t(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ wf, const float* __restrict__ bias,
              float* __restrict__ out, int N, int D, int H, int W, int ncls, int64_t ntiles, int tilesH, int tilesW, int xcd,
              const float* __restrict__ bn_scale = nullptr, const float* __restrict__ bn_shift = nullptr,
              const float* __restrict__ slope_p = nullptr, bf16_t* __restrict__ aout = nullptr, int64_t lda = 0) {
  static_assert(!BN || KS32 == 1, "the fused BatchNorm form exists for Cin = 32");
  constexpr int CIN = KS32 * 32, ROWB = CIN * 2, CH = ROWB / 16;
  float bsc[8], bsh[8], bslope = 0.f;                   // BN: this thread always stages chunk threadIdx.x % CH = channels 8 c ..
  if (BN) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bsc[j] = bn_scale[(threadIdx.x % CH) * 8 + j]; bsh[j] = bn_shift[(threadIdx.x % CH) * 8 + j]; }
    bslope = *slope_p;
  }
  __shared__ __attribute__((aligned(16))) char xs[SH * SW * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int64_t Vs = (int64_t)D * H * W;
  // A fragments: row = class r16 (zero rows beyond ncls), the lane's 8 channels 8 kg .. 8 kg + 7 of 32-channel step s
  bf16x8 afr[9][KS32];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int s_ = 0; s_ < KS32; ++s_)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        afr[tap][s_][j] = r16 < ncls ? (bf16_t)wf[((int64_t)tap * ncls + r16) * CIN + s_ * 32 + 8 * kg + j] : (bf16_t)0.f;
  f32x4e cinit;                                       // D rows 4 kg + i = classes
#pragma unroll
  for (int i = 0; i < 4; ++i) cinit[i] = (bias && 4 * kg + i < ncls) ? bias[4 * kg + i] : 0.f;
  // 16-byte chunk c of voxel v sits at slot c ^ swz(v) of the voxel's row.  ds_read_b128 is serviced in four groups of 16 lanes
  // that are NOT contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS): a B-fragment read
  // (lane = voxel r16, chunk kg) puts voxels 0-3, 12-15 with chunk kg and voxels 4-11 with chunk kg + 1 into one group.  The
  // round-3 swizzle (v / 4) % 4 was chosen for contiguous groups and is 2-way conflicted for these (36 % of this kernel's LDS
  // cycles, profiles/r05_pmc_sq_counters.txt); (v >> 1) & 2 - for 128-byte rows 2 ((v >> 1) & 3) - is conflict-free for
  // every tap shift (exhaustive check over all bases: profiles/r06_kernel_ab.txt section 3).  The staging stores (8 contiguous
  // lanes = 128 contiguous bytes, permuted inside 64-byte rows) are conflict-free under any such swizzle.
  auto swz = [](int vox) { return CH == 4 ? ((vox >> 1) & 2) : (((vox >> 1) & 3) * 2); };
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  constexpr int NLD = (SH * SW * CH + 255) / 256;
  uint4 xreg[NLD];
  auto fetch = [&](const Tile& t) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int vox = i / CH, c = i % CH;
      const int h = t.h0 + vox / SW - 1, w = t.w0 + vox % SW - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < SH * SW * CH && h >= 0 && h < H && w >= 0 && w < W)
        v = *reinterpret_cast<const uint4*>(x + ((((int64_t)t.n * D + t.d) * H + h) * W + w) * ldx + c * 8);
      xreg[k] = v;
    }
  };
  int64_t tt = tr.first;
  Tile tn = tile_of(tt < tr.end ? tt : 0, D, tilesH, tilesW);
  if (tt < tr.end) fetch(tn);
  for (; tt < tr.end; tt += tr.step) {
    const Tile t = tn;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int vox = i / CH, c = i % CH;
      if (BN && i < SH * SW * CH) {
        // BatchNorm apply + PReLU on the staged chunk (voxels outside the volume stay the convolution's zero padding)
        const int hh = vox / SW, ww = vox % SW, h = t.h0 + hh - 1, w = t.w0 + ww - 1;
        if (h >= 0 && h < H && w >= 0 && w < W) {
          bf16x8 v8 = *reinterpret_cast<bf16x8*>(&xreg[k]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = fmaf((float)v8[j], bsc[j], bsh[j]);
            z = z > 0.f ? z : z * bslope;
            v8[j] = (bf16_t)z;
          }
          xreg[k] = *reinterpret_cast<uint4*>(&v8);
          if (hh >= 1 && hh <= TH && ww >= 1 && ww <= TW)          // the tile's own voxel: its activation goes to memory
            *reinterpret_cast<uint4*>(aout + ((((int64_t)t.n * D + t.d) * H + h) * W + w) * lda + c * 8) = xreg[k];
        }
      }
      if (i < SH * SW * CH) *reinterpret_cast<uint4*>(xs + vox * ROWB + ((c ^ swz(vox)) * 16)) = xreg[k];
    }
    __syncthreads();
    if (tt + tr.step < tr.end) {
      tn = tile_of(tt + tr.step, D, tilesH, tilesW);
      fetch(tn);
    }
    // a wave owns rows 2 wave, 2 wave + 1 of the tile: four segments of 16 voxels
    f32x4e acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = cinit;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int s_ = 0; s_ < KS32; ++s_) {
        const int c = 4 * s_ + kg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int vox = (wave * 2 + (q >> 1) + kh) * SW + (q & 1) * 16 + r16 + kw;
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(xs + vox * ROWB + ((c ^ swz(vox)) * 16));
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[tap][s_], b, acc[q], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = t.h0 + wave * 2 + (q >> 1), w = t.w0 + (q & 1) * 16 + r16;
      if (h < H && w < W) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cls = 4 * kg + i;
          if (cls < ncls) out[((int64_t)t.n * ncls + cls) * Vs + ((int64_t)t.d * H + h) * W + w] = acc[q][i];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// outconv_fwd_rows (round 6, Cin = 32, the BatchNorm-fused form): outconv_fwd_t<1, true> as a march of row segments, one strip
// per WAVE - no block barrier, a wave-private LDS ring.  The tile kernel moves its 557 MB at 3.5 TB/s (157 us alone): a tile is
// a chain fill -> barrier -> BatchNorm + commit -> barrier -> MFMA with two blocks per CU.  Here a wave owns a strip of 32 voxels
// (w) x S rows (h) of one depth slice and marches down h:
//   * row h + 1 of the pre-BatchNorm tensor y (34 voxels with the w halo, 136 pieces of 16 bytes: lane -> pieces lane, lane + 64,
//     and for lanes 0-7 lane + 128; a lane always meets channel chunk lane & 3, its constants sit in registers) is requested TWO
//     rows ahead, gets a = PReLU(scale y + shift) -> bf16 applied ONCE, is written out as the activation for the strip's own
//     voxels and committed to slot (h + 1) mod 3 of the wave's ring (voxel rows of 64 bytes, chunk ^ ((v >> 1) & 2): conflict-free
//     for ds_read_b128's lane groups under every tap shift, see outconv_fwd_t);
//   * output row h = 18 MFMAs (16 x 16 x 32, two 16-voxel column groups x 9 taps) on the three ring slots, B fragments by
//     ds_read_b128, weights resident as the A operand; the logits leave as 64-byte stores per class and column group.
// Rows outside the volume and the w halo outside it are zeros (the convolution's padding applies to a, not to y).
// Same arithmetic per element as the tile kernel; a and the logits are its bits.
__global__ void __launch_bounds__(256, 4)
outconv_fwd_rows(const bf16_t* __restrict__ y, int64_t ldy, const float* __restrict__ wf, const float* __restrict__ bias,
                 float* __restrict__ out, int N, int D, int H, int W, int ncls, int tilesW, int hsegs, int S, int64_t ntask, int xcd,
                 const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, const float* __restrict__ slope_p,
                 bf16_t* __restrict__ aout, int64_t lda) {
  constexpr int RW = 34, SLOT = RW * 64;               // a ring slot: 34 voxel rows of 64 bytes
  __shared__ __attribute__((aligned(16))) char ring_all[4][3 * SLOT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  char* ring = ring_all[wave];
  const int64_t Vs = (int64_t)D * H * W;
  // A fragments: row = class r16 (zero rows beyond ncls), the lane's 8 channels 8 kg .. 8 kg + 7.  They live in LDS ([tap][lane],
  // one conflict-free 16-byte read per MFMA pair): as 36 registers they cost the kernel its fourth wave per SIMD, and with one
  // strip per wave the number of resident waves is what decides whether the benchmark's 4000 strips run in one round or two
  __shared__ __attribute__((aligned(16))) bf16x8 afr_l[9][64];
  if (wave == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = r16 < ncls ? (bf16_t)wf[((int64_t)tap * ncls + r16) * 32 + 8 * kg + j] : (bf16_t)0.f;
      afr_l[tap][lane] = a;
    }
  }
  __syncthreads();
  f32x4e cinit;                                       // D rows 4 kg + i = classes
#pragma unroll
  for (int i = 0; i < 4; ++i) cinit[i] = (bias && 4 * kg + i < ncls) ? bias[4 * kg + i] : 0.f;
  float bsc[8], bsh[8];                                // the lane's channel chunk is lane & 3 for all three of its pieces
#pragma unroll
  for (int j = 0; j < 8; ++j) { bsc[j] = bn_scale[(lane & 3) * 8 + j]; bsh[j] = bn_shift[(lane & 3) * 8 + j]; }
  const float bslope = *slope_p;
  auto swz = [](int v) { return (v >> 1) & 2; };
  // the lane's three pieces of a row: voxel (0..33) of piece t, LDS byte offset inside a slot
  int pvox[3], poff[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    pvox[t] = (lane >> 2) + 16 * t;
    poff[t] = pvox[t] * 64 + (((lane & 3) ^ swz(pvox[t])) * 16);
  }
  const bool p2 = lane < 8;                            // piece 2 exists for voxels 32, 33 only
  // B fragment offsets: column group cg, tap column kw -> voxel cg * 16 + r16 + kw, chunk kg
  int boff[2][3];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int v = cg * 16 + r16 + kw;
      boff[cg][kw] = v * 64 + ((kg ^ swz(v)) * 16);
    }
  const FplxTileRange tr = fplx_xcd_tiles((ntask + 3) / 4, xcd);      // a block takes 4 consecutive tasks, one per wave
  for (int64_t tt = tr.first; tt < tr.end; tt += tr.step) {
    const int64_t task = tt * 4 + wave;
    if (task >= ntask) continue;                       // wave-uniform
    unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)task);
    const int hs = (int)(t % (unsigned)hsegs); t /= (unsigned)hsegs;
    const int w0 = (int)(t % (unsigned)tilesW) * 32; t /= (unsigned)tilesW;
    const int d = (int)(t % (unsigned)D);
    const int n = (int)(t / (unsigned)D);
    const int hA = hs * S, hB = hA + S < H ? hA + S : H;          // the strip's output rows [hA, hB)
    const int64_t plane = ((int64_t)n * D + d) * H;                // voxel index of (n, d, 0, 0) / W
    // ---- loads of input row hh (hA - 1 .. hB): three 16-byte pieces per lane, zeros outside the volume
    auto issue = [&](int hh, u32x4e (&v)[3]) {
      const bool rowin = hh >= 0 && hh < H;
#pragma unroll
      for (int t_ = 0; t_ < 3; ++t_) {
        const int wv = w0 - 1 + pvox[t_];
        const bool ok = rowin && wv >= 0 && wv < W && (t_ < 2 || p2);
        v[t_] = ok ? *reinterpret_cast<const u32x4e*>(y + ((plane + hh) * W + wv) * ldy + (lane & 3) * 8) : u32x4e{0u, 0u, 0u, 0u};
      }
    };
    // ---- BatchNorm apply + PReLU on a loaded row, activation out for the strip's own voxels, commit to ring slot (hh + 1) mod 3
    auto commit = [&](int hh, const u32x4e (&v)[3]) {
      const bool rowin = hh >= 0 && hh < H, rowown = hh >= hA && hh < hB;
      char* slot = ring + ((hh + 1) % 3) * SLOT;       // hh >= -1
#pragma unroll
      for (int t_ = 0; t_ < 3; ++t_) {
        if (t_ == 2 && !p2) continue;
        const int wv = w0 - 1 + pvox[t_];
        u32x4e o = u32x4e{0u, 0u, 0u, 0u};
        if (rowin && wv >= 0 && wv < W) {
          bf16x8 v8 = __builtin_bit_cast(bf16x8, v[t_]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = fmaf((float)v8[j], bsc[j], bsh[j]);
            z = z > 0.f ? z : z * bslope;
            v8[j] = (bf16_t)z;
          }
          o = __builtin_bit_cast(u32x4e, v8);
          if (aout && rowown && pvox[t_] >= 1 && pvox[t_] <= 32)  // the strip's own voxel: its activation goes to memory (if wanted)
            *reinterpret_cast<u32x4e*>(aout + ((plane + hh) * W + wv) * lda + (lane & 3) * 8) = o;
        }
        *reinterpret_cast<u32x4e*>(slot + poff[t_]) = o;
      }
    };
    u32x4e ra[3], rb[3], rc[3];
    issue(hA - 1, ra);
    issue(hA, rb);
    commit(hA - 1, ra);
    issue(hA + 1, rc);
    commit(hA, rb);
    // rows: at step ho the ring holds rows ho - 1, ho (committed) and ho + 1 arrives from `rc`; row ho + 2 is requested
    for (int ho = hA; ho < hB; ++ho) {
#pragma unroll
      for (int t_ = 0; t_ < 3; ++t_) ra[t_] = rc[t_];
      if (ho + 1 < hB) issue(ho + 2, rc);
      commit(ho + 1, ra);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x4e acc[2] = {cinit, cinit};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const char* slot = ring + ((ho + kh) % 3) * SLOT;          // input row ho - 1 + kh sits in slot (ho + kh) mod 3
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const bf16x8 a = afr_l[kh * 3 + kw][lane];
#pragma unroll
          for (int cg = 0; cg < 2; ++cg) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(slot + boff[cg][kw]);
            acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[cg], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();                 // the slot of row ho - 1 is overwritten by the next commit
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const int wv = w0 + cg * 16 + r16;
        if (wv < W) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cls = 4 * kg + i;
            if (cls < ncls) out[((int64_t)n * ncls + cls) * Vs + ((int64_t)d * H + ho) * W + wv] = acc[cg][i];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// out_conv data gradient (class_num -> C0): thread = one voxel x all C0 channels; the fp32 planar dlogits
// are read once (9 taps x classes, coalesced along w), the mirrored pack wb[tap'][ci][co] sits in LDS as
// fp32 [tap][co][ci] and is read by broadcast; one 16-byte store per 8 channels.
template <int C0>
__global__ void __launch_bounds__(256)
outconv_dgrad_valu(const float* __restrict__ dl, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx, int64_t ldx,
                   int N, int D, int H, int W, int ncls) {
  __shared__ float wsh[9 * 4 * C0];                  // ncls <= 4 on this path
  for (int i = threadIdx.x; i < 9 * ncls * C0; i += 256) {
    const int ci = i % C0, co = (i / C0) % ncls, tap = i / (C0 * ncls);
    wsh[(tap * 4 + co) * C0 + ci] = (float)wb[((int64_t)tap * C0 + ci) * ncls + co];
  }
  __syncthreads();
  const int64_t Vs = (int64_t)D * H * W, V = (int64_t)N * Vs;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  // 32-bit index math (the launcher refuses V >= 2^31): four 64-bit divisions cost as much as half of the FMAs below
  const unsigned vu = (unsigned)v, Vsu = (unsigned)Vs;
  const int w = (int)(vu % (unsigned)W), h = (int)((vu / (unsigned)W) % (unsigned)H);
  const int64_t n = vu / Vsu, vs = vu % Vsu;
  float acc[C0];
#pragma unroll
  for (int j = 0; j < C0; ++j) acc[j] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3 - 1, kw = tap % 3 - 1;
    const int hh = h + kh, ww = w + kw;
    const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
    for (int co = 0; co < ncls; ++co) {
      const float g = ok ? dl[(n * ncls + co) * Vs + vs + kh * W + kw] : 0.f;
      const float* wr = wsh + (tap * 4 + co) * C0;
#pragma unroll
      for (int j = 0; j < C0; ++j) acc[j] = fmaf(g, wr[j], acc[j]);     // (v_pk_fma_f32 pairs were tried: slower)
    }
  }
#pragma unroll
  for (int j0 = 0; j0 < C0; j0 += 8) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j0 + j];
    *reinterpret_cast<bf16x8*>(dx + v * ldx + j0) = o;
  }
}

// ------------------------------------------------------------------------------------------
// out_conv data gradient on the matrix cores.  dx[v][ci] = sum over (tap, class) of dlogits[class][v + tap] * wb[tap][ci][class]
// is a GEMM with K = 9 * classes, far too thin to matter - but as VALU work (outconv_dgrad_valu) it is 576 FMAs per voxel
// and sets that kernel's time (125 us for 295 MB at the benchmark shape).  The fp32 dlogits must not simply be rounded to
// bf16 (the VALU kernel multiplies them in fp32): each value goes in as TWO bf16 terms, hi = bf16(g) and lo = bf16(g - hi),
// adjacent in K against the same weight, which carries 16 mantissa bits into an fp32 accumulation of <= 72 products that is
// rounded to bf16 (8 bits) at the end.  One 8 x 32 in-plane tile at a time (grid-strided): the fp32 dlogit planes with
// their in-plane halo sit in LDS (next tile prefetched into registers), a wave owns two rows of the tile, the B fragments
// (weights) stay in registers, the result leaves through the per-wave LDS transpose as 16-byte stores.
// MODE 1 / 2 (round 4, C0 = 32): the data gradient is not stored at all.  The site it feeds - the last 3x3x3 convolution's
// BatchNorm + PReLU (dropout-free) - needs it twice, in the reduction and in the apply pass of its backward, and the dlogits it
// is formed from are 1 / 8 of its size: both passes RECOMPUTE it here (rounded to bf16 where the stored tensor was) and run
// their arithmetic on the registers that would have been stored:
//   MODE 1  reads y, accumulates sum dz, sum dz x-hat, sum d min(z, 0) -> one partial row [2 C0 + 1] per block (fixed order),
//           for fplx_bn_act_bwd_finalize;
//   MODE 2  reads y and the finalize's coefficients, writes dy = scale (dz - k0 - x-hat k1).
// Against data gradient + reduction + apply (33 + 262 | 524 | 786 MB) the pair moves 295 + 557 MB.
template <int NT, int MODE = 0>                 // C0 / 32
__global__ void __launch_bounds__(256, (NT == 1 ? (MODE == 0 ? 3 : 2) : 1))      // fused forms: two blocks per CU without spills (three with: measured slower)
outconv_dgrad_mfma(const float* __restrict__ dl, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx, int64_t ldx,
                   int N, int D, int H, int W, int ncls, int64_t ntiles, int tilesH, int tilesW, int xcd,
                   const bf16_t* __restrict__ yv = nullptr, int64_t ldy = 0, const float* __restrict__ bn_mean = nullptr,
                   const float* __restrict__ bn_rstd = nullptr, const float* __restrict__ bn_scale = nullptr,
                   const float* __restrict__ bn_shift = nullptr, const float* __restrict__ slope_p = nullptr,
                   const float* __restrict__ coef = nullptr, float* __restrict__ part = nullptr) {
  static_assert(MODE == 0 || NT == 1, "the fused BatchNorm forms exist for C0 = 32");
  constexpr int C0 = NT * 32, MAXK = 5;              // k-steps of 16: 8 (tap, class) pairs x (hi, lo) each; ncls <= 4 -> <= 5
  __shared__ float gs[4 * SH * SW];                  // [class][SH][SW]
  __shared__ __attribute__((aligned(16))) char stg_all[4][32 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, khalf = lane >> 5;
  const int npairs = 9 * ncls, ksteps = (npairs + 7) / 8;
  const int64_t Vs = (int64_t)D * H * W;
  // B fragments: lane = ci, its 8 k-values of step s are the pairs 8 s + 4 khalf + (0..3), each twice (hi and lo terms)
  bf16x8 bfr[MAXK][NT];
#pragma unroll
  for (int sx = 0; sx < MAXK; ++sx)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pr = 8 * sx + 4 * khalf + e;
        bf16_t wv = (bf16_t)0.f;
        if (pr < npairs) wv = wb[((int64_t)(pr / ncls) * C0 + j * 32 + r) * ncls + pr % ncls];
        bfr[sx][j][2 * e] = wv;
        bfr[sx][j][2 * e + 1] = wv;
      }
  // this lane's (tap, class) pairs -> LDS offsets of the dlogit tile (relative to the lane's voxel), -1 = padding
  int goff[MAXK][4];
#pragma unroll
  for (int sx = 0; sx < MAXK; ++sx)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pr = 8 * sx + 4 * khalf + e;
      const int tap = pr / ncls, cls = pr % ncls;
      goff[sx][e] = pr < npairs ? (cls * SH + tap / 3) * SW + tap % 3 : -1;
    }
  constexpr int NLD = (4 * SH * SW + 255) / 256;
  float greg[NLD];
  // MODE 1 / 2: the lane's 8 channels at the store point are 8 (lane & 3) ..: their BatchNorm constants live in registers, the
  // four 16-byte pieces of y the lane will meet in a tile (rows 2 wave + m, voxels (lane >> 2) + 16 half) travel with the
  // next tile's dlogits
  // per channel: z = fma(y, bsc, bsh); MODE 1: x-hat = fma(y, bxa, bxb) (= (y - mean) rstd); MODE 2: dy = scale (dz - k0 -
  // x-hat k1) = fma(dz, bsc, fma(y, bxa, bxb)) with the finalize's coefficients folded into bxa / bxb
  float bsc[8], bsh[8], bxa[8], bxb[8], sdz[8], sdx[8], sds = 0.f, bslope = 0.f;
  uint4 ycur[4];
  if (MODE != 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (lane & 3) * 8 + j;
      const float m_ = bn_mean[c], rs_ = bn_rstd[c];
      bsc[j] = bn_scale[c]; bsh[j] = bn_shift[c];
      if (MODE == 1) { bxa[j] = rs_; bxb[j] = -m_ * rs_; }
      else { const float k0 = coef[c], k1 = coef[C0 + c]; bxa[j] = -bsc[j] * k1 * rs_; bxb[j] = -bsc[j] * (k0 - k1 * m_ * rs_); }
      sdz[j] = 0.f; sdx[j] = 0.f;
    }
    bslope = *slope_p;
  }
  auto fetch_y = [&](const Tile& t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = t.h0 + wave * 2 + (q >> 1), w = t.w0 + (lane >> 2) + 16 * (q & 1);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (h < H && w < W) v = *reinterpret_cast<const uint4*>(yv + ((((int64_t)t.n * D + t.d) * H + h) * W + w) * ldy + (lane & 3) * 8);
      ycur[q] = v;
    }
  };
  auto fetch = [&](const Tile& t) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int ww = i % SW, hh = (i / SW) % SH, cls = i / (SH * SW);
      const int h = t.h0 + hh - 1, w = t.w0 + ww - 1;
      float v = 0.f;
      if (cls < ncls && h >= 0 && h < H && w >= 0 && w < W)
        v = dl[((int64_t)t.n * ncls + cls) * Vs + ((int64_t)t.d * H + h) * W + w];
      greg[k] = v;
    }
  };
  char* stg = stg_all[wave];
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  int64_t tt = tr.first;
  Tile tn = tile_of(tt < tr.end ? tt : 0, D, tilesH, tilesW);
  if (tt < tr.end) fetch(tn);
  for (; tt < tr.end; tt += tr.step) {
    const Tile t = tn;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < 4 * SH * SW) gs[i] = greg[k];
    }
    if (MODE != 0) fetch_y(t);                         // this tile's y: in flight during the commit barrier and the MFMAs
    __syncthreads();
    if (tt + tr.step < tr.end) {
      tn = tile_of(tt + tr.step, D, tilesH, tilesW);
      fetch(tn);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int hr = wave * 2 + m;                     // tile row of this M-tile; lane r = column
      const float* g0 = gs + hr * SW + r;
      f32x16 acc[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#pragma unroll
      for (int sx = 0; sx < MAXK; ++sx) {
        if (sx < ksteps) {                             // uniform
          bf16x8 a;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = goff[sx][e] >= 0 ? g0[goff[sx][e]] : 0.f;
            const bf16_t hi = (bf16_t)g;
            a[2 * e] = hi;
            a[2 * e + 1] = (bf16_t)(g - (float)hi);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[sx][j], acc[j], 0, 0, 0);
        }
      }
      const int h = t.h0 + hr;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          *reinterpret_cast<bf16_t*>(stg + ((i & 3) + 8 * (i >> 2) + 4 * khalf) * 64 + r * 2) = (bf16_t)acc[j][i];
        if (h < H) {                                   // wave-uniform
          const int64_t vrow = (((int64_t)t.n * D + t.d) * H + h) * W + t.w0;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int wu = (lane >> 2) + 16 * half;
            uint4 v = *reinterpret_cast<const uint4*>(stg + wu * 64 + (lane & 3) * 16);
            if (MODE == 0) {
              if (t.w0 + wu < W) *reinterpret_cast<uint4*>(dx + (vrow + wu) * ldx + j * 32 + (lane & 3) * 8) = v;
            } else if (t.w0 + wu < W) {
              // the BatchNorm + PReLU backward of fplx_bn_act_bwd_reduce / _apply (elementwise.hip: dz_of, dropout-free) on
              // the eight bf16 values that would have been stored and the y values of the same voxel
              bf16x8 d8 = *reinterpret_cast<bf16x8*>(&v);
              const bf16x8 y8 = *reinterpret_cast<const bf16x8*>(&ycur[m * 2 + half]);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float a = (float)y8[e], d = (float)d8[e];
                const float z = fmaf(a, bsc[e], bsh[e]);
                const float dz = z > 0.f ? d : d * bslope;
                const float t_ = fmaf(a, bxa[e], bxb[e]);
                if (MODE == 1) {
                  sds += z > 0.f ? 0.f : d * z;
                  sdz[e] += dz;
                  sdx[e] = fmaf(dz, t_, sdx[e]);
                } else d8[e] = (bf16_t)fmaf(dz, bsc[e], t_);
              }
              if (MODE == 2) *reinterpret_cast<uint4*>(dx + (vrow + wu) * ldx + (lane & 3) * 8) = *reinterpret_cast<uint4*>(&d8);
            }
          }
        }
      }
    }
  }
  if (MODE == 1) {
    // the block's partial row: lanes that share a channel group sit 4 apart - butterfly inside the wave, then the four waves
    // through LDS, every addition in a fixed order
    for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sdz[e] += __shfl_xor(sdz[e], o, 64); sdx[e] += __shfl_xor(sdx[e], o, 64); }
      sds += __shfl_xor(sds, o, 64);
    }
    __syncthreads();
    float* red = gs;                                     // [wave][group 0..3][17]; the dlogit tile is dead
    if (lane < 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(wave * 4 + lane) * 17 + e] = sdz[e]; red[(wave * 4 + lane) * 17 + 8 + e] = sdx[e]; }
      red[(wave * 4 + lane) * 17 + 16] = sds;
    }
    __syncthreads();
    float* row = part + (int64_t)blockIdx.x * (2 * C0 + 1);
    if (threadIdx.x < 64) {                              // thread = (which sum 0 | 1, channel 0..31)
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31, g4 = c >> 3, e = c & 7;
      float t_ = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) t_ += red[(wv * 4 + g4) * 17 + which * 8 + e];
      row[which * C0 + c] = t_;
    }
    if (threadIdx.x == 64) {
      float t_ = 0.f;
      for (int wv = 0; wv < 4; ++wv)
        for (int g4 = 0; g4 < 4; ++g4) t_ += red[(wv * 4 + g4) * 17 + 16];
      row[2 * C0] = t_;
    }
  }
}

// ------------------------------------------------------------------------------------------
// outconv_dgrad_rows<MODE> (round 6): the fused out_conv backward (MODE 1 | 2 of outconv_dgrad_mfma above: out_conv's data
// gradient is recomputed and meets the BatchNorm + PReLU backward of the site in front of it in registers) as a STREAM of
// 32-voxel row segments, one per wave, in the form of stem_fwd_rows - no LDS tile, no block barrier.  The tile kernel moves its
// 295 | 557 MB at 2.2 | 3.9 TB/s (profiles/r05_kernel_trace_by_shape.csv: 134 | 143 us alone): every tile is a chain fill ->
// barrier -> commit -> barrier -> MFMA -> LDS transpose -> store with two blocks per CU, i.e. 43 KB in flight per CU against a
// memory latency of microseconds.  Here:
//   * the product is TRANSPOSED, D[ci][voxel] = Wb[ci][k] G^T[k][voxel], with the k-slots of outconv_dgrad_mfma (pair p = 8 s + 4
//     khalf + e = (tap, class), slots 2 e / 2 e + 1 = the hi / lo bf16 terms of the fp32 dlogit against the same weight): the same
//     products in the same slots, so d is the value the tile kernel forms;
//   * B operand: the lane's <= 10 dlogits come straight from the fp32 planes through a buffer descriptor (27-fold reuse in L1 / L2;
//     a row before the tensor = negative = huge offset -> 0); segments whose 3 x 3 neighbourhoods lie inside the plane for every
//     lane skip the padding selects;
//   * a lane ends up with 16 channels of ONE voxel (accumulator rows (i & 3) + 8 (i >> 2) + 4 khalf): y arrives as two fully
//     coalesced 16-byte loads per lane (whole 64-byte voxel rows per instruction) and is brought into that order by
//     v_permlane16_swap + v_permlane32_swap - the store path of stem_fwd_rows backwards - so the BatchNorm arithmetic runs on the
//     accumulator registers with the lane's 16 channels' constants resident;
//   * MODE 1: sum dz, sum dz x-hat, the slope term as 16 + 16 + 1 running sums per lane, folded once per kernel -> one partial
//     row per block; MODE 2: dy = fma(dz, scale, fma(y, A, B)) leaves through the two lane exchanges of stem_fwd_rows.
// The segment walk is scalar (mixed-radix step with carries, see stem_fwd_rows).
template <int MODE, int NCLS>
__global__ void __launch_bounds__(256, 4)
outconv_dgrad_rows(const float* __restrict__ dl, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx, int64_t ldx, int N, int D,
                   int H, int W, int tilesW, int64_t ntiles, int xcd, const bf16_t* __restrict__ yv, int64_t ldy,
                   const float* __restrict__ bn_mean, const float* __restrict__ bn_rstd, const float* __restrict__ bn_scale,
                   const float* __restrict__ bn_shift, const float* __restrict__ slope_p, const float* __restrict__ coef,
                   float* __restrict__ part) {
  static_assert(MODE == 1 || MODE == 2, "reduction or apply");
  static_assert(NCLS == 2, "k-slot tables below are written for two classes (every shipped configuration); others keep the tile kernel");
  constexpr int C0 = 32, NPAIR = 9 * NCLS, KS = (NPAIR + 7) / 8;      // 18 pairs, 3 k-steps of 16
  __shared__ float red[4][3][32];
  __shared__ __attribute__((aligned(16))) float cst[4][32];          // BatchNorm constants per channel: z scale, z shift, A, B
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, khalf = lane >> 5;
  const int64_t Vs = (int64_t)D * H * W;
  // k-slots as in outconv_dgrad_mfma: pair p = 8 s + 4 khalf + e = (tap, class) = (p / 2, p % 2), slots 2 e / 2 e + 1 of the
  // lane's 8 = its hi / lo terms.  With two classes e >> 1 picks the tap (4 s + 2 khalf + (e >> 1)) and e & 1 the class: the
  // class plane is the load's SCALAR offset (non-negative: the range check, which sees the vector offset alone, is not weakened),
  // so a lane keeps one vector offset per (s, e >> 1).
  bf16x8 afr[KS];                                      // A: row = channel r
  int toff[KS][2];                                     // byte offset of tap (s, e >> 1) relative to the segment's first voxel, + 4 r
  unsigned tapsel = 0;                                 // 4 bits per (s, e >> 1): the tap index (9 = spare)
#pragma unroll
  for (int sx = 0; sx < KS; ++sx)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pr = 8 * sx + 4 * khalf + e;
      const bool ok = pr < NPAIR;
      const int tap = ok ? pr / NCLS : 4, cls = pr % NCLS;
      const bf16_t wv = ok ? wb[((int64_t)tap * C0 + r) * NCLS + cls] : (bf16_t)0.f;
      afr[sx][2 * e] = wv;
      afr[sx][2 * e + 1] = wv;
      if ((e & 1) == 0) {
        toff[sx][e >> 1] = ((tap / 3 - 1) * W + (tap % 3 - 1) + r) * 4;
        tapsel |= (unsigned)(ok ? tap : 9) << (4 * (2 * sx + (e >> 1)));
      }
    }
  const int cls_bytes = (int)(Vs * 4);
  const int64_t gbytes = (int64_t)N * NCLS * Vs * 4;
  // (the launcher keeps the dlogits below 2 GiB; raw buffer, no stride: out-of-range offsets read as zeros)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dl, 0, (int)gbytes, 0x00020000);
  // BatchNorm constants.  z = fma(y, c0, c1); MODE 1: x-hat = fma(y, c2, c3) (= (y - mean) rstd); MODE 2: dy = scale (dz - k0 -
  // x-hat k1) = fma(dz, c0, fma(y, c2, c3)) with the finalize's coefficients folded into c2 / c3 (outconv_dgrad_mfma's constants).
  // They live in LDS: a lane's 16 channels are four runs of four (8 q + 4 khalf ..), one 16-byte broadcast read per run and array.
  if (threadIdx.x < 32) {
    const int c = threadIdx.x;
    const float m_ = bn_mean[c], rs_ = bn_rstd[c], sc_ = bn_scale[c];
    cst[0][c] = sc_;
    cst[1][c] = bn_shift[c];
    if (MODE == 1) { cst[2][c] = rs_; cst[3][c] = -m_ * rs_; }
    else { const float k0 = coef[c], k1 = coef[C0 + c]; cst[2][c] = -sc_ * k1 * rs_; cst[3][c] = -sc_ * (k0 - k1 * m_ * rs_); }
  }
  __syncthreads();
  const float bslope = *slope_p;
  float sdz[MODE == 1 ? 16 : 1], sdx[MODE == 1 ? 16 : 1], sds = 0.f;
#pragma unroll
  for (int i = 0; i < (MODE == 1 ? 16 : 1); ++i) { sdz[i] = 0.f; sdx[i] = 0.f; }
  // row-contiguous register sets of the y loads: set 0 = voxel lane & 15, set 1 = voxel 16 + (lane & 15), 16-byte run
  // (lane >> 5) + 2 ((lane >> 4) & 1) of the voxel's 64-byte row
  const int rv = lane & 15, rrun = (lane >> 5) + 2 * ((lane >> 4) & 1);

  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);   // tiles = 4 consecutive 32-voxel segments (one per wave)
  auto sread = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  const unsigned uW = (unsigned)tilesW, uH = (unsigned)H, uD = (unsigned)D;
  unsigned sw_, sh_, sd_, sn_;                         // 4 step as (w tile, h, d, n) digits
  struct Seg { unsigned cw, ch, cd, cn; };
  Seg nx;                                              // the segment whose loads are issued next
  {
    unsigned t = sread((unsigned)tr.step * 4u);
    sw_ = sread(t % uW); t = sread(t / uW);
    sh_ = sread(t % uH); t = sread(t / uH);
    sd_ = sread(t % uD); sn_ = sread(t / uD);
    t = sread((unsigned)tr.first * 4u + (unsigned)wave);
    nx.cw = sread(t % uW); t = sread(t / uW);
    nx.ch = sread(t % uH); t = sread(t / uH);
    nx.cd = sread(t % uD); nx.cn = sread(t / uD);     // >= N: a segment past the end of the list (the last tile may be ragged)
  }
  auto advance = [&](Seg& g) {                         // += 4 step, carries as scalar selects
    g.cw += sw_;
    unsigned carry = g.cw >= uW ? 1u : 0u;
    g.cw -= carry ? uW : 0u;
    g.ch += sh_ + carry;
    carry = g.ch >= uH ? 1u : 0u;
    g.ch -= carry ? uH : 0u;
    g.cd += sd_ + carry;
    carry = g.cd >= uD ? 1u : 0u;
    g.cd -= carry ? uD : 0u;
    g.cn += sn_ + carry;
  };
  // one segment's loads: the lane's dlogits (2 classes x 2 taps per k-step) and y of the segment's 32 voxels.  They are issued a
  // whole segment ahead of their use: with three waves per SIMD that keeps about 50 KB of y in flight per CU
  auto issue = [&](const Seg& g, float (&gv)[KS][4], u32x4e (&yq)[2]) {
    if (g.cn >= (unsigned)N) return;
    const int n = (int)g.cn, d = (int)g.cd, h = (int)g.ch, w0 = (int)g.cw * 32;
    const int sbase = (((n * NCLS * D + d) * H + h) * W + w0) * 4;                     // scalar byte offset in the class-0 plane
#pragma unroll
    for (int sx = 0; sx < KS; ++sx)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        gv[sx][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(toff[sx][e >> 1] + sbase),
                                                                                    (e & 1) ? cls_bytes : 0, 0));
    const int64_t vrow = (((int64_t)n * D + d) * H + h) * W + w0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bool ok = w0 + rv + 16 * q < W;
      yq[q] = ok ? *reinterpret_cast<const u32x4e*>(yv + (vrow + rv + 16 * q) * ldy + 8 * rrun) : u32x4e{0u, 0u, 0u, 0u};
    }
  };
  auto consume = [&](const Seg& g, const float (&gv)[KS][4], const u32x4e (&yq)[2]) {
    if (g.cn >= (unsigned)N) return;
    const int n = (int)g.cn, d = (int)g.cd, h = (int)g.ch, w0 = (int)g.cw * 32;
    // ---- out_conv's data gradient of the lane's voxel: 16 channels.  Padding: bit t of `tapok` = tap t lies inside the plane
    // for this lane (bit 9 = the spare slots: never); interior segments (scalar test) have every tap valid
    const bool interior = h >= 1 && h + 1 < H && w0 >= 1 && w0 + 32 < W;
    const int wv = w0 + r;
    const bool wc = wv < W;
    unsigned tapok = 0x1ffu;
    if (!interior) {
      const unsigned colm = (wv - 1 >= 0 && wv - 1 < W ? 1u : 0u) | (wc ? 2u : 0u) | (wv + 1 < W ? 4u : 0u);     // kw = 0, 1, 2
      tapok = (h >= 1 ? colm : 0u) | (colm << 3) | (h + 1 < H ? colm << 6 : 0u);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) {
      bf16x8 b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned tap = (tapsel >> (4 * (2 * sx + (e >> 1)))) & 15u;
        const float gq = ((tapok >> tap) & 1u) ? gv[sx][e] : 0.f;
        const bf16_t hi = (bf16_t)gq;
        b[2 * e] = hi;
        b[2 * e + 1] = (bf16_t)(gq - (float)hi);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[sx], b, acc, 0, 0, 0);
    }
    // ---- y of the lane's voxel in accumulator order: the two exchanges of the store path, backwards (both are involutions)
    unsigned pk[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32x2e sw = __builtin_amdgcn_permlane16_swap(yq[0][e], yq[1][e], false, false);
      pk[e] = sw[0];
      pk[4 + e] = sw[1];
    }
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const u32x2e sw = __builtin_amdgcn_permlane32_swap(pk[4 * g2 + e], pk[4 * g2 + 2 + e], false, false);
        pk[4 * g2 + e] = sw[0];
        pk[4 * g2 + 2 + e] = sw[1];
      }
    // ---- the BatchNorm + PReLU backward of fplx_bn_act_bwd_reduce / _apply (elementwise.hip: dz_of, dropout-free) on the bf16
    // values that would have been stored; quad q = channels 8 q + 4 khalf + (0..3)
    unsigned ok8[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4e c0 = *reinterpret_cast<const f32x4e*>(&cst[0][8 * q + 4 * khalf]);
      const f32x4e c1 = *reinterpret_cast<const f32x4e*>(&cst[1][8 * q + 4 * khalf]);
      const f32x4e c2 = *reinterpret_cast<const f32x4e*>(&cst[2][8 * q + 4 * khalf]);
      const f32x4e c3 = *reinterpret_cast<const f32x4e*>(&cst[3][8 * q + 4 * khalf]);
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 4 * q + j;
        const unsigned word = pk[2 * q + (j >> 1)];
        const float a = __builtin_bit_cast(float, (j & 1) ? (word & 0xffff0000u) : (word << 16));
        const float dd = (float)(bf16_t)acc[i];
        const float z = fmaf(a, c0[j], c1[j]);
        const float dz = z > 0.f ? dd : dd * bslope;
        const float t_ = fmaf(a, c2[j], c3[j]);
        if (MODE == 1) {
          if (wc) {
            sds += z > 0.f ? 0.f : dd * z;
            sdz[i] += dz;
            sdx[i] = fmaf(dz, t_, sdx[i]);
          }
        } else o[j] = fmaf(dz, c0[j], t_);
      }
      if (MODE == 2) {
        const bf16_t e0 = (bf16_t)o[0], e1 = (bf16_t)o[1], e2 = (bf16_t)o[2], e3 = (bf16_t)o[3];
        ok8[2 * q] = (unsigned)__builtin_bit_cast(unsigned short, e0) | ((unsigned)__builtin_bit_cast(unsigned short, e1) << 16);
        ok8[2 * q + 1] = (unsigned)__builtin_bit_cast(unsigned short, e2) | ((unsigned)__builtin_bit_cast(unsigned short, e3) << 16);
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const u32x2e sw = __builtin_amdgcn_permlane32_swap(ok8[4 * g2 + e], ok8[4 * g2 + 2 + e], false, false);
          ok8[4 * g2 + e] = sw[0];
          ok8[4 * g2 + 2 + e] = sw[1];
        }
      if (wc) {
        bf16_t* dst = dx + ((((int64_t)n * D + d) * H + h) * W + wv) * ldx + 8 * khalf;
        *reinterpret_cast<u32x4e*>(dst) = u32x4e{ok8[0], ok8[1], ok8[2], ok8[3]};
        *reinterpret_cast<u32x4e*>(dst + 16) = u32x4e{ok8[4], ok8[5], ok8[6], ok8[7]};
      }
    }
  };
  float gA[KS][4], gB[KS][4];
  u32x4e yA[2], yB[2];
  Seg cur = nx;
  issue(cur, gA, yA);
  for (int64_t tt = tr.first; tt < tr.end; tt += tr.step) {
    nx = cur;
    advance(nx);
    if (tt + tr.step < tr.end) issue(nx, gB, yB);
    consume(cur, gA, yA);
    cur = nx;
#pragma unroll
    for (int sx = 0; sx < KS; ++sx)
#pragma unroll
      for (int e = 0; e < 4; ++e) gA[sx][e] = gB[sx][e];
    yA[0] = yB[0];
    yA[1] = yB[1];
  }
  if (MODE == 1) {
    // a lane's sums belong to channels (i & 3) + 8 (i >> 2) + 4 khalf: fold the 32 voxel lanes of each half, then the waves -
    // every addition in a fixed order
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { sdz[i] += __shfl_xor(sdz[i], o, 64); sdx[i] += __shfl_xor(sdx[i], o, 64); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sds += __shfl_xor(sds, o, 64);
    if (lane == 0) red[wave][2][0] = sds;
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = (i & 3) + 8 * (i >> 2) + 4 * khalf;
        red[wave][0][c] = sdz[i];
        red[wave][1][c] = sdx[i];
      }
    }
    __syncthreads();
    float* row = part + (int64_t)blockIdx.x * (2 * C0 + 1);
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      row[which * C0 + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
    if (threadIdx.x == 64) row[2 * C0] = red[0][2][0] + red[1][2][0] + red[2][2][0] + red[3][2][0];
  }
}

// ------------------------------------------------------------------------------------------
// out_conv weight gradient: rows = ci (32 per block column), cols = classes (padded to 32), K = voxels.
// x tile [voxel][32 ch] with an in-plane halo in LDS (transposed reads), dlogits straight from the
// fp32 planes; the 9 taps are dealt to the 4 waves (3/2/2/2).
__global__ void __launch_bounds__(256)
outconv_wgrad_mfma(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ dl, float* __restrict__ part,
                   int N, int D, int H, int W, int ncls, int64_t ntiles, int tilesH, int tilesW, int xcd) {
  __shared__ __attribute__((aligned(16))) char xs[SH * SW * 64];
  __shared__ __attribute__((aligned(16))) bf16_t dls[32][TH * TW];   // dlogits tile per class, bf16
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, khalf = lane >> 5;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;
  const int cit = blockIdx.y;
  const int64_t Vs = (int64_t)D * H * W;
  f32x16 acc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  // the next tile's rows (x with its in-plane halo, the dlogit planes) travel global -> registers while this tile computes
  // (round 3; the synchronous fill left every block idle for a memory round trip per tile, as in the forward kernel)
  constexpr int NLX = (SH * SW * 4 + 255) / 256, NLG = 4 * TH * TW / 256;      // ncls <= 4 on the prefetched path
  uint4 xreg[NLX];
  float greg[NLG];
  const bool pre = ncls <= 4;
  auto fetch = [&](const Tile& t) {
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int vox = i >> 2, ch = i & 3;
      const int h = t.h0 + vox / SW - 1, w = t.w0 + vox % SW - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < SH * SW * 4 && h >= 0 && h < H && w >= 0 && w < W)
        v = *reinterpret_cast<const uint4*>(x + ((((int64_t)t.n * D + t.d) * H + h) * W + w) * ldx + cit * 32 + ch * 8);
      xreg[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int co = i / (TH * TW), vox = i % (TH * TW);
      const int h = t.h0 + vox / TW, w = t.w0 + vox % TW;
      float v = 0.f;
      if (co < ncls && h < H && w < W) v = dl[((int64_t)t.n * ncls + co) * Vs + ((int64_t)t.d * H + h) * W + w];
      greg[k] = v;
    }
  };
  int64_t tt = tr.first;
  Tile tn = tile_of(tt < tr.end ? tt : 0, D, tilesH, tilesW);
  if (pre && tt < tr.end) fetch(tn);
  for (; tt < tr.end; tt += tr.step) {
    const Tile t = pre ? tn : tile_of(tt, D, tilesH, tilesW);
    __syncthreads();
    if (pre) {
#pragma unroll
      for (int k = 0; k < NLX; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < SH * SW * 4) *reinterpret_cast<uint4*>(xs + (i >> 2) * 64 + (i & 3) * 16) = xreg[k];
      }
#pragma unroll
      for (int k = 0; k < NLG; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < ncls * TH * TW) dls[i / (TH * TW)][i % (TH * TW)] = (bf16_t)greg[k];
      }
    } else {
    for (int i = threadIdx.x; i < SH * SW * 4; i += 256) {
      const int vox = i >> 2, ch = i & 3;
      const int h = t.h0 + vox / SW - 1, w = t.w0 + vox % SW - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (h >= 0 && h < H && w >= 0 && w < W)
        v = *reinterpret_cast<const uint4*>(x + ((((int64_t)t.n * D + t.d) * H + h) * W + w) * ldx + cit * 32 + ch * 8);
      *reinterpret_cast<uint4*>(xs + vox * 64 + ch * 16) = v;
    }
    for (int i = threadIdx.x; i < ncls * TH * TW; i += 256) {      // coalesced along w
      const int co = i / (TH * TW), vox = i % (TH * TW);
      const int h = t.h0 + vox / TW, w = t.w0 + vox % TW;
      float v = 0.f;
      if (h < H && w < W) v = dl[((int64_t)t.n * ncls + co) * Vs + ((int64_t)t.d * H + h) * W + w];
      dls[co][vox] = (bf16_t)v;
    }
    }
    __syncthreads();
    if (pre && tt + tr.step < tr.end) {
      tn = tile_of(tt + tr.step, D, tilesH, tilesW);
      fetch(tn);
    }
    // k-steps two at a time with the fragments of the next step requested before the MFMAs of the current one
    auto load_ks = [&](int ks, bf16x8& fb, bf16x8 (&fa)[3]) {
      const int hr = ks >> 1, ws = (ks & 1) * 16;
      fb = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (r < ncls) fb = *reinterpret_cast<const bf16x8*>(&dls[r][hr * TW + ws + 8 * khalf]);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int tap = wave + 4 * a;                  // wave-uniform
        if (tap < 9) {
          const int kh = tap / 3, kw = tap % 3;
          fa[a] = tr_frag64(xs + ((hr + kh) * SW + ws + kw) * 64 + lane_off);
        }
      }
    };
    bf16x8 fb0, fb1, fa0[3], fa1[3];
    load_ks(0, fb0, fa0);
#pragma unroll 1
    for (int ks = 0; ks < 16; ks += 2) {
      load_ks(ks + 1, fb1, fa1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (wave + 4 * a < 9) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[a], fb0, acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < 16) load_ks(ks + 2, fb0, fa0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (wave + 4 * a < 9) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[a], fb1, acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // part[block][cit][tap][ci 32][co 32]
  float* out = part + ((int64_t)blockIdx.x * gridDim.y + cit) * (9 * 1024);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int tap = wave + 4 * a;
    if (tap < 9) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2) + 4 * khalf;
        out[(tap * 32 + ci) * 32 + r] = acc[a][i];
      }
    }
  }
}

__global__ void __launch_bounds__(256)
outconv_wgrad_reduce(const float* __restrict__ part, int nblk, int ncit, int C0, int ncls, float* __restrict__ dw) {
  __shared__ float red[256];
  const int total = ncit * 9 * 1024;
  const int i = blockIdx.x * SP_OUT + (threadIdx.x & (SP_OUT - 1));
  const float t = sum_partials(part, nblk, total, i, red);
  if (threadIdx.x >= SP_OUT || i >= total) return;
  const int co = i & 31;
  if (co >= ncls) return;
  const int ci_l = (i >> 5) & 31, tap = (i >> 10) % 9, cit = i / (9 * 1024);
  dw[((int64_t)co * C0 + cit * 32 + ci_l) * 9 + tap] = t;
}

// ------------------------------------------------------------------------------------------
// outconv_wgrad_rows (round 6): out_conv's weight and bias gradients from the PRE-BatchNorm tensor y of the site in front of
// it, as a stream of 32-voxel row segments (one per wave, no block barrier; the form of outconv_dgrad_rows).  With it the
// activation a = PReLU(BN(y)) of that site has no reader left in the training step - the fused forward (outconv_fwd_rows)
// and backward (outconv_dgrad_rows) form it from y on their own - so it is never stored: one 262-MB write and the kernel
// that re-read it (outconv_wgrad_mfma, 135 us) leave the step.
//   D[ci][pair] += sum over the segment's 32 voxels u of a[ci][u] g[pair][u],  pair = (tap, class),  g = dlogit[class][u - off(tap)]
//   * A (rows = channels, K = voxels): y arrives as two coalesced 16-byte pieces per lane (voxel (lane >> 2) + 16 t, channel
//     chunk lane & 3 - the lane's 8 channels' constants stay in registers), gets BatchNorm + PReLU -> bf16 exactly as the
//     stored activation did, goes to the wave's 2-KB LDS tile [voxel][32 ch] (lane-linear 16-byte writes) and comes back
//     transposed through ds_read_b64_tr_b16;
//   * B (cols = pairs, K = voxels): the lane's 8 consecutive voxels of its pair are 8 consecutive floats of a dlogit row -
//     two 16-byte buffer loads per k-step straight from the fp32 planes (27-fold reuse in L1 / L2), rounded to bf16 (the
//     tile kernel's operand); rows above / below the plane and the two columns beside it are zeros; lanes beyond the
//     last pair read out of range (zeros);
//   * the bias gradient is the plain fp32 sum of the centre tap's lanes.
// One partial row of 9 ncls 32 + ncls floats per block, summed in a fixed order by outconv_wgrad_rows_reduce.
__global__ void __launch_bounds__(256, 4)
outconv_wgrad_rows(const bf16_t* __restrict__ yv, int64_t ldy, const float* __restrict__ dl, float* __restrict__ part, int N, int D,
                   int H, int W, int ncls, int tilesW, int64_t ntiles, int xcd, const float* __restrict__ bn_scale,
                   const float* __restrict__ bn_shift, const float* __restrict__ slope_p) {
  constexpr int C0 = 32;
  __shared__ __attribute__((aligned(16))) char smem[4][4096];        // per wave: the a tile (2 KB); at the end its 32 x 32 sums
  __shared__ float sbf[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, khalf = lane >> 5;
  char* atile = smem[wave];
  const int64_t Vs = (int64_t)D * H * W;
  const int npair = 9 * ncls;
  // ---- A side: pieces and constants
  const int pv = lane >> 2, chunk = lane & 3;
  float bsc[8], bsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bsc[j] = bn_scale[chunk * 8 + j]; bsh[j] = bn_shift[chunk * 8 + j]; }
  const float bslope = *slope_p;
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int lane_off = (8 * (g4 >> 1) + q4) * 64 + (16 * (g4 & 1) + 4 * p4) * 2;      // tr_frag64: row = channel lane & 31
  // ---- B side: the lane's pair
  const bool pok = r < npair;
  const int tap = pok ? r / ncls : 4, cls = pok ? r % ncls : 0;
  const int kh = tap / 3, kw = tap % 3;
  // source voxel of element j of k-step s: row h - (kh - 1), column w0 + 16 s + 8 khalf + j - (kw - 1)
  const int goff = pok ? ((-(kh - 1)) * W - (kw - 1) + 8 * khalf) * 4 + cls * (int)(Vs * 4) : (int)0x80000000;
  const int64_t gbytes = (int64_t)N * ncls * Vs * 4;          // (the launcher keeps the dlogits below 2 GiB)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dl, 0, (int)gbytes, 0x00020000);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float sb = 0.f;

  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);   // tiles = 4 consecutive 32-voxel segments (one per wave)
  auto sread = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  const unsigned uW = (unsigned)tilesW, uH = (unsigned)H, uD = (unsigned)D;
  unsigned sw_, sh_, sd_, sn_;
  struct Seg { unsigned cw, ch, cd, cn; };
  Seg nx;
  {
    unsigned t = sread((unsigned)tr.step * 4u);
    sw_ = sread(t % uW); t = sread(t / uW);
    sh_ = sread(t % uH); t = sread(t / uH);
    sd_ = sread(t % uD); sn_ = sread(t / uD);
    t = sread((unsigned)tr.first * 4u + (unsigned)wave);
    nx.cw = sread(t % uW); t = sread(t / uW);
    nx.ch = sread(t % uH); t = sread(t / uH);
    nx.cd = sread(t % uD); nx.cn = sread(t / uD);
  }
  auto advance = [&](Seg& g) {
    g.cw += sw_;
    unsigned carry = g.cw >= uW ? 1u : 0u;
    g.cw -= carry ? uW : 0u;
    g.ch += sh_ + carry;
    carry = g.ch >= uH ? 1u : 0u;
    g.ch -= carry ? uH : 0u;
    g.cd += sd_ + carry;
    carry = g.cd >= uD ? 1u : 0u;
    g.cd -= carry ? uD : 0u;
    g.cn += sn_ + carry;
  };
  auto issue = [&](const Seg& g, u32x4e (&gq)[4], u32x4e (&yq)[2]) {
    if (g.cn >= (unsigned)N) return;
    const int n = (int)g.cn, d = (int)g.cd, h = (int)g.ch, w0 = (int)g.cw * 32;
    const int sbase = (((n * ncls * D + d) * H + h) * W + w0) * 4;         // byte offset of the segment in the class-0 plane
    const bool rowok = (unsigned)(h - (kh - 1)) < (unsigned)H;
    const int o = goff + sbase;                          // >= 0, except: -4 where the window starts one column left of the tensor's first row
    if (n == 0 && d == 0 && h <= 1 && w0 == 0) {
      // (the hardware adds offsets without wrapping: a window that starts below the tensor is out of range as a whole.  Such
      // a lane loads from the tensor's first byte and moves the components up; the column mask zeroes component 0 anyway)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ok = o + (k >> 1) * 64 + (k & 1) * 16;
        const bool neg = ok < 0;
        int ol = neg ? 0 : ok;
        asm volatile("" : "+v"(ol));                     // (one opaque offset: no base + immediate split of a negative base)
        const u32x4e q = __builtin_bit_cast(u32x4e, __builtin_amdgcn_raw_buffer_load_b128(rsrc, rowok ? (unsigned)ol : 0x80000000u, 0, 0));
        gq[k] = neg ? u32x4e{0u, q.x, q.y, q.z} : q;
      }
    } else {
      const unsigned vo = rowok ? (unsigned)o : 0x80000000u;
      // (a window that reaches beyond the tensor's last byte returns zeros there: raw buffers are range-checked per dword)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        gq[k] = __builtin_bit_cast(u32x4e, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + (unsigned)((k >> 1) * 64 + (k & 1) * 16), 0, 0));
    }
    const int64_t vrow = (((int64_t)n * D + d) * H + h) * W + w0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bool ok = w0 + pv + 16 * t < W;
      yq[t] = ok ? *reinterpret_cast<const u32x4e*>(yv + (vrow + pv + 16 * t) * ldy + 8 * chunk) : u32x4e{0u, 0u, 0u, 0u};
    }
  };
  auto consume = [&](const Seg& g, const u32x4e (&gq)[4], const u32x4e (&yq)[2]) {
    if (g.cn >= (unsigned)N) return;
    const int w0 = (int)g.cw * 32;
    // ---- a = PReLU(scale y + shift) -> bf16 into the tile (zeros beyond W: such voxels then add nothing to any sum)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      u32x4e o = u32x4e{0u, 0u, 0u, 0u};
      if (w0 + pv + 16 * t < W) {
        bf16x8 v8 = __builtin_bit_cast(bf16x8, yq[t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = fmaf((float)v8[j], bsc[j], bsh[j]);
          z = z > 0.f ? z : z * bslope;
          v8[j] = (bf16_t)z;
        }
        o = __builtin_bit_cast(u32x4e, v8);
      }
      *reinterpret_cast<u32x4e*>(atile + (lane + 64 * t) * 16) = o;
    }
    // ---- g: columns outside the plane are zeros (only the segments that touch a row end have any)
    float gf[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (whole-vector cast: __builtin_bit_cast(float, gq[k][e]) on a vector ELEMENT reads element 0 with this compiler)
      const f32x4e q = __builtin_bit_cast(f32x4e, gq[k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) gf[4 * k + e] = q[e];
    }
    if (!(w0 >= 1 && w0 + 33 <= W)) {
      const int wsrc = w0 + 8 * khalf - (kw - 1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if ((unsigned)(wsrc + 16 * s + j) >= (unsigned)W) gf[8 * s + j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) sb += gf[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 fa = tr_frag64(atile + 16 * s * 64 + lane_off);
      bf16x8 fb;
#pragma unroll
      for (int j = 0; j < 8; ++j) fb[j] = (bf16_t)gf[8 * s + j];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();                   // the tile is rewritten by the next segment
  };
  u32x4e gA[4], gB[4], yA[2], yB[2];
  Seg cur = nx;
  issue(cur, gA, yA);
  for (int64_t tt = tr.first; tt < tr.end; tt += tr.step) {
    nx = cur;
    advance(nx);
    if (tt + tr.step < tr.end) issue(nx, gB, yB);
    consume(cur, gA, yA);
    cur = nx;
#pragma unroll
    for (int k = 0; k < 4; ++k) gA[k] = gB[k];
    yA[0] = yB[0];
    yA[1] = yB[1];
  }
  // ---- fold the four waves in a fixed order: sums[pair r][channel]
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float* fold = reinterpret_cast<float*>(smem[wave]);
#pragma unroll
  for (int i = 0; i < 16; ++i) fold[r * 32 + (i & 3) + 8 * (i >> 2) + 4 * khalf] = acc[i];
  sbf[wave][lane] = sb;
  __syncthreads();
  const int rowf = npair * C0 + ncls;
  float* row = part + (int64_t)blockIdx.x * rowf;
  for (int i = threadIdx.x; i < npair * C0; i += 256) {
    const float* f0 = reinterpret_cast<const float*>(smem[0]);
    row[i] = ((f0[i] + f0[1024 + i]) + f0[2048 + i]) + f0[3072 + i];
  }
  if ((int)threadIdx.x < ncls) {
    const int pl = 4 * ncls + threadIdx.x;              // the centre tap's pair of this class
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) t += sbf[wv][pl] + sbf[wv][32 + pl];
    row[npair * C0 + threadIdx.x] = t;
  }
}

// dw[class][ci][tap], db[class] = sums over the block rows of outconv_wgrad_rows
__global__ void __launch_bounds__(256)
outconv_wgrad_rows_reduce(const float* __restrict__ part, int nblk, int ncls, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[256];
  constexpr int C0 = 32;
  const int total = 9 * ncls * C0 + ncls;
  const int i = blockIdx.x * SP_OUT + (threadIdx.x & (SP_OUT - 1));
  const float t = sum_partials(part, nblk, total, i, red);
  if (threadIdx.x >= SP_OUT || i >= total) return;
  if (i >= 9 * ncls * C0) {
    if (db) db[i - 9 * ncls * C0] = t;
    return;
  }
  const int pair = i >> 5, ci = i & 31, tap = pair / ncls, cls = pair % ncls;
  dw[((int64_t)cls * C0 + ci) * 9 + tap] = t;
}

inline int64_t tiles_of(int n, int d, int h, int w, int* th, int* tw) {
  *th = (h + TH - 1) / TH;
  *tw = (w + TW - 1) / TW;
  return (int64_t)n * d * (*th) * (*tw);                  // < 2^31 for anything that fits in memory (256 voxels per tile)
}
inline int edge_blocks(int64_t ntiles, int mult = 1) {
  const int cap = (int)fplx_knob(FPLX_K_EDGE_BLOCKS);       // tuning knob (benchmarks only)
  const int64_t c = (int64_t)cap * mult;
  return (int)(ntiles < c ? ntiles : c);
}
// the stem weight gradient keeps few bytes per block (27 x 32 partial sums): twice the blocks hide more of its per-tile
// load latency (98 -> 75 us at level 0) and the extra partial rows are negligible; for out_conv (9 x 1024 floats per
// block and ci tile) the reduction would eat the gain
constexpr int STEM_WGRAD_MULT = 2;

}  // namespace

// ---- entry points used by conv_generic.hip; return 1 = handled, 0 = not applicable, <0 = error
extern "C" int fplx_edge_stem_rows(int n, int d, int h, int w, int cin, int cout) {
  if (!(cin == 1 || cin == 4) || cout % 32 != 0) return 0;
  int th, tw;
  return edge_blocks(tiles_of(n, d, h, w, &th, &tw));
}

extern "C" int fplx_edge_stem_fwd(const float* x, const void* wf, const float* bias, void* y, int64_t ldy, int n, int d,
                                  int h, int w, int cin, int cout, float* stats, hipStream_t st) {
  if (!(cin == 1 || cin == 4) || cout % 32 != 0) return 0;
  const int vec_ok = ldy % 8 == 0 && ((uintptr_t)y % 16) == 0;
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  const int nb = edge_blocks(nt);
  if (vec_ok && fplx_knob(FPLX_K_STEM_ROWS) && (cin == 1 || fplx_knob(FPLX_K_STEM_ROWS) != 2) &&
      (int64_t)n * cin * d * h * w < ((int64_t)1 << 29)) {
    // the LDS-free row kernel; the same number of blocks (= statistics rows) as the tile kernel  (knob 2: in_chns = 1 only)
    const int tilesW = (w + 31) / 32;
    const int64_t segs = (int64_t)n * d * h * tilesW, nt4 = (segs + 3) / 4;
    for (int co0 = 0; co0 < cout; co0 += 32) {
      if (cin == 1) stem_fwd_rows<1><<<nb, 256, 0, st>>>(x, (const bf16_t*)wf, bias, (bf16_t*)y, ldy, n, d, h, w, co0, cout, stats, tilesW, nt4, fplx_xcd_on());
      else stem_fwd_rows<4><<<nb, 256, 0, st>>>(x, (const bf16_t*)wf, bias, (bf16_t*)y, ldy, n, d, h, w, co0, cout, stats, tilesW, nt4, fplx_xcd_on());
    }
    int rc0 = fplx_check_launch("edge_stem_fwd_rows");
    return rc0 < 0 ? rc0 : 1;
  }
  for (int co0 = 0; co0 < cout; co0 += 32) {
    if (cin == 1)
      stem_fwd_mfma<1><<<nb, 256, 0, st>>>(x, (const bf16_t*)wf, bias, (bf16_t*)y, ldy, n, d, h, w, co0, cout, stats, nt, th, tw, vec_ok, fplx_xcd_on());
    else
      stem_fwd_mfma<4><<<nb, 256, 0, st>>>(x, (const bf16_t*)wf, bias, (bf16_t*)y, ldy, n, d, h, w, co0, cout, stats, nt, th, tw, vec_ok, fplx_xcd_on());
  }
  int rc = fplx_check_launch("edge_stem_fwd");
  return rc < 0 ? rc : 1;
}

extern "C" size_t fplx_edge_stem_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  if (!(cin == 1 || cin == 4) || cout % 32 != 0) return 0;
  int th, tw;
  const int nb = edge_blocks(tiles_of(n, d, h, w, &th, &tw), STEM_WGRAD_MULT);
  return (size_t)nb * ((27 * cin + 31) / 32) * 1024 * sizeof(float);
}

// y != NULL: dy is d(a), the gradient w.r.t. the site's OUTPUT, and the BatchNorm + PReLU backward apply runs inside the kernel
// (stem_wgrad_mfma<CIN, true>; coef from fplx_bn_act_bwd_finalize)
extern "C" int fplx_edge_stem_wgrad_bn(const float* x, const void* dy, int64_t ldy, float* dw, int n, int d, int h, int w,
                                       int cin, int cout, void* ws, hipStream_t st, const void* y, int64_t ldyy, const float* mean,
                                       const float* rstd, const float* scale, const float* shift, const float* slope,
                                       const float* coef) {
  if (!(cin == 1 || cin == 4) || cout % 32 != 0 || ldy % 8 != 0 || ((uintptr_t)dy % 16)) return 0;
  if (y && (ldyy % 8 != 0 || ((uintptr_t)y % 16))) return 0;
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  const int nb = edge_blocks(nt, STEM_WGRAD_MULT);
  const int rt = (27 * cin + 31) / 32;
  const StemBnBwd bn{(const bf16_t*)y, ldyy, mean, rstd, scale, shift, slope, coef, cout};
  for (int co0 = 0; co0 < cout; co0 += 32) {
    if (y) {
      if (cin == 1) stem_wgrad_mfma<1, true><<<nb, 256, 0, st>>>(x, (const bf16_t*)dy, ldy, (float*)ws, n, d, h, w, co0, nt, th, tw, fplx_xcd_on(), bn);
      else stem_wgrad_mfma<4, true><<<nb, 256, 0, st>>>(x, (const bf16_t*)dy, ldy, (float*)ws, n, d, h, w, co0, nt, th, tw, fplx_xcd_on(), bn);
    }
    else if (cin == 1) stem_wgrad_mfma<1><<<nb, 256, 0, st>>>(x, (const bf16_t*)dy, ldy, (float*)ws, n, d, h, w, co0, nt, th, tw, fplx_xcd_on());
    else stem_wgrad_mfma<4><<<nb, 256, 0, st>>>(x, (const bf16_t*)dy, ldy, (float*)ws, n, d, h, w, co0, nt, th, tw, fplx_xcd_on());
    stem_wgrad_reduce<<<(rt * 1024 + SP_OUT - 1) / SP_OUT, 256, 0, st>>>((const float*)ws, nb, rt, cin, co0, dw);
  }
  int rc = fplx_check_launch("edge_stem_wgrad");
  return rc < 0 ? rc : 1;
}
extern "C" int fplx_edge_stem_wgrad(const float* x, const void* dy, int64_t ldy, float* dw, int n, int d, int h, int w,
                                    int cin, int cout, void* ws, hipStream_t st) {
  return fplx_edge_stem_wgrad_bn(x, dy, ldy, dw, n, d, h, w, cin, cout, ws, st, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr);
}

extern "C" int fplx_edge_outconv_fwd(const void* x, int64_t ldx, const float* wf, const float* bias, float* out, int n,
                                     int d, int h, int w, int cin, int ncls, hipStream_t st) {
  if (!(cin == 16 || cin == 32 || cin == 64) || ncls > 32 || ldx % 8 != 0 || ((uintptr_t)x % 16)) return 0;
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  const int nb = (int)(nt < 2048 ? nt : 2048);
  if ((cin == 32 || cin == 64) && ncls <= 16 && fplx_knob(FPLX_K_OUTCONV_T)) {
    if (cin == 32) outconv_fwd_t<1><<<nb, 256, 0, st>>>((const bf16_t*)x, ldx, wf, bias, out, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
    else outconv_fwd_t<2><<<nb, 256, 0, st>>>((const bf16_t*)x, ldx, wf, bias, out, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
    int rct = fplx_check_launch("edge_outconv_fwd_t");
    return rct < 0 ? rct : 1;
  }
  if (cin == 16) outconv_fwd_mfma<1><<<nb, 256, 0, st>>>((const bf16_t*)x, ldx, wf, bias, out, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
  else if (cin == 32) outconv_fwd_mfma<2><<<nb, 256, 0, st>>>((const bf16_t*)x, ldx, wf, bias, out, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
  else outconv_fwd_mfma<4><<<nb, 256, 0, st>>>((const bf16_t*)x, ldx, wf, bias, out, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
  int rc = fplx_check_launch("edge_outconv_fwd");
  return rc < 0 ? rc : 1;
}

extern "C" int fplx_edge_outconv_dgrad(const float* dl, const void* wb, void* dx, int64_t ldx, int n, int d, int h, int w,
                                       int c0, int ncls, hipStream_t st) {
  if (!(c0 == 16 || c0 == 32 || c0 == 64) || ncls > 4 || ldx % 8 != 0 || ((uintptr_t)dx % 16)) return 0;
  const int64_t V = (int64_t)n * d * h * w;
  if (V >= ((int64_t)1 << 31)) return 0;                      // 32-bit voxel decode in the kernel
  {
    const int kmf = (int)fplx_knob(FPLX_K_OUTCONV_DGRAD_MFMA);   // A/B knob
    if (kmf && (c0 == 32 || c0 == 64) && ncls <= 4 && ldx % 8 == 0 && ((uintptr_t)dx % 16) == 0) {
      int th, tw;
      const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
      const int nbm = (int)(nt < 2048 ? nt : 2048);
      if (c0 == 32) outconv_dgrad_mfma<1><<<nbm, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
      else outconv_dgrad_mfma<2><<<nbm, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
      int rcm = fplx_check_launch("edge_outconv_dgrad_mfma");
      return rcm < 0 ? rcm : 1;
    }
  }
  const unsigned nb = (unsigned)((V + 255) / 256);
  if (c0 == 16) outconv_dgrad_valu<16><<<nb, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, ncls);
  else if (c0 == 32) outconv_dgrad_valu<32><<<nb, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, ncls);
  else outconv_dgrad_valu<64><<<nb, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, ncls);
  int rc = fplx_check_launch("edge_outconv_dgrad");
  return rc < 0 ? rc : 1;
}

// ---- out_conv fused with the BatchNorm + PReLU passes of the convolution site in front of it (round 4)
extern "C" int fplx_edge_outconv_bn_ok(int n, int d, int h, int w, int c0, int ncls) {
  return c0 == 32 && ncls >= 1 && ncls <= 4 && (int64_t)n * d * h * w < ((int64_t)1 << 31);
}
// blocks of the fused data-gradient kernels = partial rows the reduction form writes
// the fused backward as a stream of row segments (outconv_dgrad_rows, knob outconv_dgrad_rows): tiles of four 32-voxel segments
static bool outconv_rows_on(int n, int d, int h, int w, int ncls) {
  return fplx_knob(FPLX_K_OUTCONV_DGRAD_ROWS) != 0 && ncls == 2 && (int64_t)n * ncls * d * h * w < ((int64_t)1 << 29) &&
         (int64_t)n * d * h * w < ((int64_t)1 << 31);
}
static int64_t outconv_rows_tiles(int n, int d, int h, int w) { return ((int64_t)n * d * h * ((w + 31) / 32) + 3) / 4; }
// partial rows of the fused backward's reduction = its blocks
extern "C" int fplx_edge_outconv_bn_rows(int n, int d, int h, int w, int ncls) {
  if (outconv_rows_on(n, d, h, w, ncls)) {
    const int64_t nt4 = outconv_rows_tiles(n, d, h, w);
    return (int)(nt4 < 1024 ? nt4 : 1024);
  }
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  return (int)(nt < 2048 ? nt : 2048);
}
extern "C" int fplx_edge_outconv_fwd_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* slope,
                                        void* a, int64_t lda, const float* wf, const float* bias, float* out, int n, int d,
                                        int h, int w, int c0, int ncls, hipStream_t st) {
  if (!fplx_edge_outconv_bn_ok(n, d, h, w, c0, ncls) || ldy % 8 != 0 || lda % 8 != 0 || ((uintptr_t)y % 16) || ((uintptr_t)a % 16))
    return 0;
  if (!a && fplx_knob(FPLX_K_OUTCONV_FWD_ROWS) == 0) return 0;      // only the march of row segments can leave the activation out
  if (fplx_knob(FPLX_K_OUTCONV_FWD_ROWS) != 0) {
    // the march of row segments (outconv_fwd_rows): strips of 32 voxels x S rows, one per wave
    const int tilesW = (w + 31) / 32, S = 32, hsegs = (h + S - 1) / S;
    const int64_t ntask = (int64_t)n * d * tilesW * hsegs, nblk = (ntask + 3) / 4;
    const int nbr = (int)(nblk < 4096 ? nblk : 4096);
    outconv_fwd_rows<<<nbr, 256, 0, st>>>((const bf16_t*)y, ldy, wf, bias, out, n, d, h, w, ncls, tilesW, hsegs, S, ntask, fplx_xcd_on(),
                                          scale, shift, slope, (bf16_t*)a, lda);
    const int rcr = fplx_check_launch("edge_outconv_fwd_rows");
    return rcr < 0 ? rcr : 1;
  }
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  const int nb = (int)(nt < 2048 ? nt : 2048);
  outconv_fwd_t<1, true><<<nb, 256, 0, st>>>((const bf16_t*)y, ldy, wf, bias, out, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on(),
                                             scale, shift, slope, (bf16_t*)a, lda);
  const int rc = fplx_check_launch("edge_outconv_fwd_bn");
  return rc < 0 ? rc : 1;
}
// mode 1: reduction (part: fplx_edge_outconv_bn_rows rows of 2 c0 + 1 floats), mode 2: apply (coef from the finalize, dy out)
extern "C" int fplx_edge_outconv_dgrad_bn(int mode, const float* dl, const void* wb, const void* y, int64_t ldy,
                                          const float* mean, const float* rstd, const float* scale, const float* shift,
                                          const float* slope, const float* coef, float* part, void* dy, int64_t lddy, int n,
                                          int d, int h, int w, int c0, int ncls, hipStream_t st) {
  if (!fplx_edge_outconv_bn_ok(n, d, h, w, c0, ncls) || ldy % 8 != 0 || ((uintptr_t)y % 16)) return 0;
  if (mode == 2 && (lddy % 8 != 0 || ((uintptr_t)dy % 16))) return 0;
  if (outconv_rows_on(n, d, h, w, ncls)) {
    const int64_t nt4 = outconv_rows_tiles(n, d, h, w);
    const int nbr = fplx_edge_outconv_bn_rows(n, d, h, w, ncls), tilesW = (w + 31) / 32;
    if (mode == 1)
      outconv_dgrad_rows<1, 2><<<nbr, 256, 0, st>>>(dl, (const bf16_t*)wb, nullptr, 0, n, d, h, w, tilesW, nt4, fplx_xcd_on(),
                                                    (const bf16_t*)y, ldy, mean, rstd, scale, shift, slope, nullptr, part);
    else
      outconv_dgrad_rows<2, 2><<<nbr, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dy, lddy, n, d, h, w, tilesW, nt4,
                                                    fplx_xcd_on(), (const bf16_t*)y, ldy, mean, rstd, scale, shift, slope, coef, nullptr);
    const int rcr = fplx_check_launch("edge_outconv_dgrad_rows");
    return rcr < 0 ? rcr : 1;
  }
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  const int nb = (int)(nt < 2048 ? nt : 2048);
  if (mode == 1)
    outconv_dgrad_mfma<1, 1><<<nb, 256, 0, st>>>(dl, (const bf16_t*)wb, nullptr, 0, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on(),
                                                 (const bf16_t*)y, ldy, mean, rstd, scale, shift, slope, nullptr, part);
  else
    outconv_dgrad_mfma<1, 2><<<nb, 256, 0, st>>>(dl, (const bf16_t*)wb, (bf16_t*)dy, lddy, n, d, h, w, ncls, nt, th, tw,
                                                 fplx_xcd_on(), (const bf16_t*)y, ldy, mean, rstd, scale, shift, slope, coef,
                                                 nullptr);
  const int rc = fplx_check_launch("edge_outconv_dgrad_bn");
  return rc < 0 ? rc : 1;
}

// out_conv's weight + bias gradients from the pre-BatchNorm tensor (outconv_wgrad_rows): 0 = not available for this shape
extern "C" size_t fplx_edge_outconv_wgrad_bn_ws_bytes(int n, int d, int h, int w, int c0, int ncls) {
  if (c0 != 32 || ncls < 1 || ncls > 3 || fplx_knob(FPLX_K_OUTCONV_FWD_ROWS) == 0 ||
      (int64_t)n * ncls * d * h * w >= ((int64_t)1 << 29))
    return 0;
  const int64_t nt4 = outconv_rows_tiles(n, d, h, w);
  return (size_t)(nt4 < 1024 ? nt4 : 1024) * (9 * ncls * 32 + ncls) * sizeof(float);
}
extern "C" int fplx_edge_outconv_wgrad_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* slope,
                                          const float* dl, float* dw, float* db, int n, int d, int h, int w, int c0, int ncls,
                                          void* ws, hipStream_t st) {
  if (fplx_edge_outconv_wgrad_bn_ws_bytes(n, d, h, w, c0, ncls) == 0 || ldy % 8 != 0 || ((uintptr_t)y % 16)) return 0;
  const int64_t nt4 = outconv_rows_tiles(n, d, h, w);
  const int nbr = (int)(nt4 < 1024 ? nt4 : 1024), tilesW = (w + 31) / 32;
  outconv_wgrad_rows<<<nbr, 256, 0, st>>>((const bf16_t*)y, ldy, dl, (float*)ws, n, d, h, w, ncls, tilesW, nt4, fplx_xcd_on(), scale,
                                          shift, slope);
  const int total = 9 * ncls * 32 + ncls;
  outconv_wgrad_rows_reduce<<<(total + SP_OUT - 1) / SP_OUT, 256, 0, st>>>((const float*)ws, nbr, ncls, dw, db);
  const int rc = fplx_check_launch("edge_outconv_wgrad_bn");
  return rc < 0 ? rc : 1;
}

extern "C" size_t fplx_edge_outconv_wgrad_ws_bytes(int n, int d, int h, int w, int c0, int ncls) {
  if (c0 % 32 != 0 || ncls > 32) return 0;
  int th, tw;
  const int nb = edge_blocks(tiles_of(n, d, h, w, &th, &tw));
  return (size_t)nb * (c0 / 32) * 9 * 1024 * sizeof(float);
}

extern "C" int fplx_edge_outconv_wgrad(const void* x, int64_t ldx, const float* dl, float* dw, int n, int d, int h, int w,
                                       int c0, int ncls, void* ws, hipStream_t st) {
  if (c0 % 32 != 0 || ncls > 32 || ldx % 8 != 0 || ((uintptr_t)x % 16)) return 0;
  int th, tw;
  const int64_t nt = tiles_of(n, d, h, w, &th, &tw);
  const int nb = edge_blocks(nt);
  dim3 grid(nb, c0 / 32);
  outconv_wgrad_mfma<<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, dl, (float*)ws, n, d, h, w, ncls, nt, th, tw, fplx_xcd_on());
  const int total = (c0 / 32) * 9 * 1024;
  outconv_wgrad_reduce<<<(total + SP_OUT - 1) / SP_OUT, 256, 0, st>>>((const float*)ws, nb, c0 / 32, c0, ncls, dw);
  int rc = fplx_check_launch("edge_outconv_wgrad");
  return rc < 0 ? rc : 1;
}
