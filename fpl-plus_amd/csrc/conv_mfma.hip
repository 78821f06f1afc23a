// MFMA (matrix-core) 3x3x3 convolution kernels for gfx950: bf16 operands, fp32 accumulate,
// NDHWC activations.  v_mfma_f32_32x32x16_bf16 everywhere:
//   A fragment: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7   (8 contiguous bf16 = 16 B)
//   B fragment: lane l holds B[k = 8*(l>>5) + j][col l&31]
//   C/D       : lane l holds D[row (reg&3) + 8*(reg>>2) + 4*(l>>5)][col l&31], reg = 0..15
//
// conv_fwd_direct : implicit GEMM, rows = output voxels, cols = output channels, K = taps x Cin.
//                   Operand fragments are loaded straight from global/L2 (16 B per lane), no LDS:
//                   the universal path (any spatial size, Cin % 16 == 0, Cout % 32 == 0).  With the
//                   mirrored pack it is also the data-gradient kernel.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int DIRECT_THREADS = 256;

// MODE 0: 3x3x3 "same" convolution (27 taps, +-1 shifts, zero padding)
// MODE 1: data gradient of ConvTranspose3d(k=2,s=2): 8 taps, the source voxel of tap (i,j,k) is
//         (2d+i, 2h+j, 2w+k) of the twice-as-large dy grid; rows = INPUT voxels of the deconv
// MODE 2: data gradient of ConvTranspose2d(k=2,s=2) on every depth slice (2.5D levels): 4 taps (j,k), source
//         voxel (d, 2h+j, 2w+k)
template <int MT, int NTL, int MODE>
__global__ void __launch_bounds__(DIRECT_THREADS)
conv_fwd_direct(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                int Cin, int Cout, float* __restrict__ stats, float* __restrict__ partial = nullptr) {
  // split-K: gridDim.z > 1 deals the taps to blockIdx.z (tap % gridDim.z); fp32 partial tiles go to
  // partial[z][voxel][Cout] and splitk_finish_k adds bias, stores bf16 and produces the statistics
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kh8 = (lane >> 5) * 8;
  const int64_t V = (int64_t)N * D * H * W;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * (MT * 32);
  const int n0 = blockIdx.y * (NTL * 32);

  // decode this lane's voxel for every M-tile
  int vn[MT], vd[MT], vh[MT], vw[MT];
  bool vok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int64_t v = m0 + t * 32 + r;
    vok[t] = v < V;
    if (!vok[t]) v = 0;
    unsigned q = (unsigned)v;                    // V < 2^31 (checked by the launchers)
    vw[t] = (int)(q % (unsigned)W); q /= (unsigned)W;
    vh[t] = (int)(q % (unsigned)H); q /= (unsigned)H;
    vd[t] = (int)(q % (unsigned)D); q /= (unsigned)D;
    vn[t] = (int)q;
  }
  f32x16 acc[MT][NTL];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][j][i] = 0.f;

  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int NTAPS = MODE == 0 ? 27 : (MODE == 1 ? 8 : 4), SD = MODE == 2 ? 1 : 2;
  if (MODE != 0 && gridDim.z == 1) {
    // Transposed-convolution data gradient: no padding, so a tap is a block-uniform offset from the lane's tap-0 row and
    // the (tap, 16-channel step) loop is a flat stream of independent loads.  Four steps are kept in flight per wave: the
    // plain loop below waits out a full memory round trip per step (up1 of the benchmark, 327 MB: 102 us = 3.3 TB/s)
    constexpr int P = 4;
    const bf16_t* a0[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int64_t vi = (((int64_t)vn[t] * SD * D + SD * vd[t]) * 2 * H + 2 * vh[t]) * 2 * W + 2 * vw[t];
      a0[t] = x + (vok[t] ? vi : 0) * ldx + kh8;
    }
    const bf16_t* b0[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) b0[j] = wp + ((int64_t)(n0 + j * 32 + r)) * Cin + kh8;
    const int KS = Cin / 16, NIT = NTAPS * KS;
    bf16x8 ra[P][MT], rb[P][NTL];
    auto fetch = [&](int it, int u) {
      const int tap = it / KS, kc = (it - tap * KS) * 16;
      const int64_t toff = ((int64_t)((tap >> 2) * 2 * H + ((tap >> 1) & 1)) * 2 * W + (tap & 1)) * ldx + kc;   // uniform
      const int64_t woff = (int64_t)tap * Cout * Cin + kc;
#pragma unroll
      for (int t = 0; t < MT; ++t) ra[u][t] = vok[t] ? *reinterpret_cast<const bf16x8*>(a0[t] + toff) : zero;
#pragma unroll
      for (int j = 0; j < NTL; ++j) rb[u][j] = *reinterpret_cast<const bf16x8*>(b0[j] + woff);
    };
#pragma unroll
    for (int u = 0; u < P; ++u) fetch(u < NIT ? u : NIT - 1, u);
    for (int it = 0; it < NIT; it += P) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        if (it + u < NIT) {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
              acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[u][t], rb[u][j], acc[t][j], 0, 0, 0);
        }
        const int nx = it + P + u;
        fetch(nx < NIT ? nx : NIT - 1, u);            // past the end: a harmless re-read (keeps the loop branch-free)
      }
    }
  } else
  for (int tap = blockIdx.z; tap < NTAPS; tap += gridDim.z) {
    const bf16_t* ap[MT];
    bool aok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      int64_t vi;
      if (MODE == 0) {
        const int kd = tap / 9 - 1, kh = (tap / 3) % 3 - 1, kw = tap % 3 - 1;
        const int dd = vd[t] + kd, hh = vh[t] + kh, ww = vw[t] + kw;
        aok[t] = vok[t] && dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W;
        vi = (((int64_t)vn[t] * D + dd) * H + hh) * W + ww;
      } else {
        aok[t] = vok[t];
        vi = (((int64_t)vn[t] * SD * D + SD * vd[t] + (tap >> 2)) * 2 * H + 2 * vh[t] + ((tap >> 1) & 1)) * 2 * W +
             2 * vw[t] + (tap & 1);
      }
      ap[t] = x + (aok[t] ? vi : 0) * ldx + kh8;
    }
    const bf16_t* bp[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) bp[j] = wp + ((int64_t)tap * Cout + n0 + j * 32 + r) * Cin + kh8;
    for (int kc = 0; kc < Cin; kc += 16) {
      bf16x8 a[MT], b[NTL];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t] = aok[t] ? *reinterpret_cast<const bf16x8*>(ap[t] + kc) : zero;
#pragma unroll
      for (int j = 0; j < NTL; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bp[j] + kc);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[j], acc[t][j], 0, 0, 0);
    }
  }

  const int rh = (lane >> 5) * 4;
  if (partial) {
    float* pz = partial + (int64_t)blockIdx.z * V * Cout;
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const int co = n0 + j * 32 + r;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t v = m0 + t * 32 + (i & 3) + 8 * (i >> 2) + rh;
          if (v < V) pz[v * Cout + co] = acc[t][j][i];
        }
    }
    return;
  }
  // epilogue: + bias, per-channel statistics of the unrounded outputs, bf16 store.  Aligned outputs go through a 2-KB
  // per-wave LDS transpose and leave as 16-byte stores (2-byte global stores are issue-bound); channel slices of odd
  // buffers keep the element stores.
  float s[NTL], q[NTL];
  const bool vec_ok = ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0;        // uniform
  __shared__ __attribute__((aligned(16))) char stg_all[4][32 * 64];
  char* stg = stg_all[wave];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int co = n0 + j * 32 + r;
    const float bv = bias ? bias[co] : 0.f;
    s[j] = q[j] = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + rh;
        const int64_t v = m0 + t * 32 + row;
        const float o = acc[t][j][i] + bv;
        if (vec_ok) *reinterpret_cast<bf16_t*>(stg + row * 64 + r * 2) = (bf16_t)o;
        if (v < V) {
          if (!vec_ok) y[v * ldy + co] = (bf16_t)o;
          s[j] += o;
          q[j] = fmaf(o, o, q[j]);
        }
      }
      if (vec_ok) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int row = (lane >> 2) + 16 * half;
          const int64_t v = m0 + t * 32 + row;
          const uint4 pk = *reinterpret_cast<const uint4*>(stg + row * 64 + (lane & 3) * 16);
          if (v < V) *reinterpret_cast<uint4*>(y + v * ldy + n0 + j * 32 + (lane & 3) * 8) = pk;
        }
      }
    }
  }
  if (stats) {
    __shared__ float red[4][2][NTL * 32];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const float a = s[j] + __shfl_xor(s[j], 32, 64);
      const float b = q[j] + __shfl_xor(q[j], 32, 64);
      if (lane < 32) { red[wave][0][j * 32 + r] = a; red[wave][1][j * 32 + r] = b; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NTL * 32; i += DIRECT_THREADS) {
      const int which = i / (NTL * 32), c = i % (NTL * 32);
      const float t = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
      stats[((int64_t)blockIdx.x * 2 + which) * Cout + n0 + c] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv_wgrad_stream: dW[tap][ci][co] = sum over voxels of x[v + tap][ci] * dy[v][co] for one 32x32
// (ci, co) tile pair.  A block owns an 8 x TW footprint in (h, w) and marches along d with a ring
// of three x slabs (1-voxel halo, zero filled = the convolution's padding) and one dy slab in LDS.
// Both MFMA operands need 8 consecutive VOXELS per lane for a fixed channel - the transpose of the
// NDHWC image - which ds_read_b64_tr_b16 delivers for free.  The 27 taps are dealt to the 4 waves
// (7/7/7/6), each wave keeps its taps' 32x32 fp32 tiles in registers for the whole march and
// writes them once; a second kernel sums the per-block partials in a fixed order.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int WG_TH = 8;

__device__ __forceinline__ bf16x8 tr_frag(const char* base_lo) {
  // two transposed 4x16 block reads: voxels +0..3 and +4..7 (64 B per voxel row)
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base_lo));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base_lo + 4 * 64));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// one depth step of conv_wgrad_stream for wave WV: its taps WV, WV+4, ... are compile-time constants, so the
// k-loop is straight-line code: 2 + 14 transposed reads for step ks+1 in flight behind the 7 MFMAs of step ks
// TWOD: the layer is a Conv2d per depth slice (2.5D levels): only the middle-plane taps 9..17 exist - wave WV owns
// 9 + WV, 13 + WV (and 17 for wave 0): a third of the MFMAs, the other accumulators stay untouched
// COT = 2: two output-channel tiles per block (dy slab = two 32-channel planes): every x fragment - the bulk of the LDS
// reads, one per tap - feeds two MFMAs: 1.3 transposed reads per MFMA instead of 2.3
template <int WV, int TW, int CIT, int COT, bool TWOD>
__device__ __forceinline__ void wgrad_depth_step(f32x16 (&acc)[7 * CIT * COT], const char* sl0, const char* sl1,
                                                 const char* sl2, const char* dys, int lane_off) {
  constexpr int TH = 8, SW = TW + 2, NKS = TH * TW / 16;
  constexpr int NT = TWOD ? (9 - WV + 3) / 4 : (27 - WV + 3) / 4;     // 3D: 7 (6 for wave 3); 2D: 3 (wave 0) or 2
  constexpr int PLANE = (TH + 2) * SW * 64;           // one ci tile of an x slab: [voxel][32 ch]
  constexpr int DYPLANE = TH * TW * 64;               // one co tile of the dy slab
  bf16x8 fbw[2][COT], faw[2][NT * CIT];
  auto load_dy = [&](int ks, int o) {
    const int hr = ks / (TW / 16), ws = (ks % (TW / 16)) * 16;
    return tr_frag(dys + o * DYPLANE + (hr * TW + ws) * 64 + lane_off);
  };
  auto load_x = [&](int ks, int j) {                  // j = c * NT + i: ci tile c, tap WV + 4 i
    const int c = j / NT, i = j % NT;
    const int hr = ks / (TW / 16), ws = (ks % (TW / 16)) * 16;
    const int tap = (TWOD ? 9 : 0) + WV + 4 * i;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;                // folded: j is unrolled, WV constant
    const char* sl = kd == 0 ? sl0 : (kd == 1 ? sl1 : sl2);
    return tr_frag(sl + c * PLANE + ((hr + kh) * SW + ws + kw) * 64 + lane_off);
  };
#pragma unroll
  for (int o = 0; o < COT; ++o) fbw[0][o] = load_dy(0, o);
#pragma unroll
  for (int j = 0; j < NT * CIT; ++j) faw[0][j] = load_x(0, j);
  // one wave per SIMD: nothing else fills the matrix core while this wave issues a burst of LDS reads, so the
  // fragments of the next k-step are requested one at a time IN the gaps between the MFMAs of the current one
  // (the fences pin that order; the loads complete a whole k-step before their first use)
#pragma unroll 1
  for (int ks = 0; ks < NKS; ks += 2) {
#pragma unroll
    for (int j = 0; j < NT * CIT; ++j) {
      if (j < COT) fbw[1][j] = load_dy(ks + 1, j);
      faw[1][j] = load_x(ks + 1, j);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < COT; ++o)
        acc[(o * CIT + j / NT) * 7 + j % NT] =
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(faw[0][j], fbw[0][o], acc[(o * CIT + j / NT) * 7 + j % NT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int kn = ks + 2 < NKS ? ks + 2 : ks;      // last trip: a harmless re-read instead of branches in the gaps
#pragma unroll
    for (int j = 0; j < NT * CIT; ++j) {
      if (j < COT) fbw[0][j] = load_dy(kn, j);
      faw[0][j] = load_x(kn, j);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < COT; ++o)
        acc[(o * CIT + j / NT) * 7 + j % NT] =
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(faw[1][j], fbw[1][o], acc[(o * CIT + j / NT) * 7 + j % NT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// the whole march of wave WV (its taps are compile-time constants): the wave variants never merge before the end
// of the kernel, so the seven accumulator tiles stay in one register class (no VGPR <-> AGPR copies per depth)
template <int TW, int WV, int CIT, int COT, bool TWOD>
__device__ __forceinline__ void wgrad_march(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                            int64_t ldy, float* __restrict__ part, int N, int D, int H, int W, int Cin,
                                            int Cout, int tilesH, int tilesW, int dsegs, int dlen,
                                            const bf16_t* __restrict__ x1, const FplxBlock bid) {
  constexpr int TH = WG_TH, SW = TW + 2, SH = TH + 2, SLAB = SH * SW;   // voxels per x slab
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PLANE = SLAB * 64, XSLOT = CIT * PLANE;
  char* xs = smem;                                   // [3 slots][CIT ci tiles][SLAB][32] bf16
  char* dys = smem + 3 * XSLOT;                      // [COT co tiles][TH*TW][32] bf16
  constexpr int DYPLANE = TH * TW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int wave = WV;
  int b = bid.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = b;
  const int ncg = Cin / (32 * CIT);                  // groups of CIT ci tiles: one block reads whole CIT*64-byte rows
  const int cot = bid.y / ncg, cg = bid.y % ncg;
  const int h0 = th * TH, w0 = tw * TW, d0 = seg * dlen;
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const bf16_t* xb = x + cg * (32 * CIT);
  const bf16_t* dyb = dy + cot * (32 * COT);

  // async-stage split: the next depth's x slab and dy slab travel global -> registers while the current
  // depth is computed, and are committed to LDS behind the barrier that ends the depth
  constexpr int XCH = 4 * CIT;                        // 16-byte chunks per x voxel
  constexpr int YCH = 4 * COT;                        // 16-byte chunks per dy voxel
  constexpr int NLX = (SLAB * XCH + 255) / 256, NLY = TH * TW * YCH / 256;
  uint4 xreg[NLX], yreg[NLY];
  auto fetch_x = [&](int d) {
    const bool dok = d >= 0 && d < D;
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = tid + k * 256;
      const int vox = i / XCH, ch = i % XCH;
      const int hh = vox / SW + h0 - 1, ww = vox % SW + w0 - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < SLAB * XCH && dok && hh >= 0 && hh < H && ww >= 0 && ww < W)
        v = *reinterpret_cast<const uint4*>(((x1 && ch >= 4) ? x1 - 32 : xb) + ((((int64_t)n * D + d) * H + hh) * W + ww) * ldx + ch * 8);
      xreg[k] = v;
    }
  };
  auto commit_x = [&](int d) {
    char* dst = xs + ((d + 1) % 3) * XSLOT;
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = tid + k * 256;
      const int vox = i / XCH, ch = i % XCH;            // plane ch / 4 keeps 64-byte rows (conflict-free tr reads)
      if (i < SLAB * XCH) *reinterpret_cast<uint4*>(dst + (ch >> 2) * PLANE + vox * 64 + (ch & 3) * 16) = xreg[k];
    }
  };
  auto fetch_dy = [&](int d) {
#pragma unroll
    for (int k = 0; k < NLY; ++k) {
      const int i = tid + k * 256;
      const int vox = i / YCH, ch = i % YCH;
      const int hh = vox / TW + h0, ww = vox % TW + w0;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (d < D && hh < H && ww < W)
        v = *reinterpret_cast<const uint4*>(dyb + ((((int64_t)n * D + d) * H + hh) * W + ww) * ldy + ch * 8);
      yreg[k] = v;
    }
  };
  auto commit_dy = [&]() {
#pragma unroll
    for (int k = 0; k < NLY; ++k) {
      const int i = tid + k * 256;
      const int vox = i / YCH, ch = i % YCH;
      *reinterpret_cast<uint4*>(dys + (ch >> 2) * DYPLANE + vox * 64 + (ch & 3) * 16) = yreg[k];
    }
  };

  // transposed-read lane geometry (see header comment of tr_frag): group g = lane / 16
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;   // bytes

  f32x16 acc[7 * CIT * COT];
#pragma unroll
  for (int i = 0; i < 7 * CIT * COT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (d0 < d1) {
    fetch_x(d0 - 1); commit_x(d0 - 1);
    fetch_x(d0); commit_x(d0);
    fetch_x(d0 + 1); commit_x(d0 + 1);
    fetch_dy(d0); commit_dy();
  }
  __syncthreads();
  for (int d = d0; d < d1; ++d) {
    const bool more = d + 1 < d1;
    if (more) {                                  // in flight during this depth's MFMAs
      fetch_x(d + 2);
      fetch_dy(d + 1);
    }
    {
      const char* sl0 = xs + ((d + 0) % 3) * XSLOT;              // depth d - 1
      const char* sl1 = xs + ((d + 1) % 3) * XSLOT;              // depth d
      const char* sl2 = xs + ((d + 2) % 3) * XSLOT;              // depth d + 1
      wgrad_depth_step<WV, TW, CIT, COT, TWOD>(acc, sl0, sl1, sl2, dys, lane_off);
    }
    __syncthreads();                             // every wave is done with depth d-1's slot and the dy slab
    if (more) {
      commit_x(d + 2);
      commit_dy();
      __syncthreads();
    }
  }
  // partial tiles: part[blockIdx.x][pair][tap][co][ci] - a lane owns 4 consecutive ci per register quad, so the
  // tile leaves as 16-byte stores (4 per tile instead of 16 dword stores: the epilogue is store-issue bound)
  const int co = lane & 31, rbase = (lane >> 5) * 4;
#pragma unroll
  for (int o = 0; o < COT; ++o)
#pragma unroll
    for (int c = 0; c < CIT; ++c) {
      // pair index of (co tile cot * COT + o, ci tile cg * CIT + c) in the [Cout/32][Cin/32] enumeration the reduction uses
      const int pair = (cot * COT + o) * (Cin / 32) + cg * CIT + c;
      float* out = part + ((int64_t)bid.x * ((Cin / 32) * (Cout / 32)) + pair) * (27 * 1024);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int tap = (TWOD ? 9 : 0) + wave + 4 * i;      // TWOD: only taps 9..17 are written (and later reduced)
        if (tap < (TWOD ? 18 : 27)) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<float4*>(out + (tap * 32 + co) * 32 + 8 * g4 + rbase) =
                make_float4(acc[(o * CIT + c) * 7 + i][4 * g4 + 0], acc[(o * CIT + c) * 7 + i][4 * g4 + 1],
                            acc[(o * CIT + c) * 7 + i][4 * g4 + 2], acc[(o * CIT + c) * 7 + i][4 * g4 + 3]);
        }
      }
    }
}

template <int TW, int CIT, int COT, bool TWOD>
__global__ void __launch_bounds__(256)
conv_wgrad_stream(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
                  float* __restrict__ part, int N, int D, int H, int W, int Cin, int Cout, int tilesH, int tilesW,
                  int dsegs, int dlen, const bf16_t* __restrict__ x1, int xcd) {
  const FplxBlock bid = fplx_xcd_block(xcd);
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {            // wave-uniform
    case 0: wgrad_march<TW, 0, CIT, COT, TWOD>(x, ldx, dy, ldy, part, N, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 1: wgrad_march<TW, 1, CIT, COT, TWOD>(x, ldx, dy, ldy, part, N, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    case 2: wgrad_march<TW, 2, CIT, COT, TWOD>(x, ldx, dy, ldy, part, N, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
    default: wgrad_march<TW, 3, CIT, COT, TWOD>(x, ldx, dy, ldy, part, N, D, H, W, Cin, Cout, tilesH, tilesW, dsegs, dlen, x1, bid); break;
  }
}

// dw[co][ci][tap] = sum_b part[b][pair][tap][co%32][ci%32].  A block = 16 outputs (four consecutive ci each, 16-byte loads) x 16
// partial lanes: lane pl adds the blocks b = pl, pl + 16, ... one after the other (four loads in flight), then the 16 lane sums
// are added in index order through LDS - a fixed order for every output.  (Round 3's form had 4 partial lanes: with the
// 200-500 partial tiles of the large levels each lane walked 50-128 dependent rounds of loads - 22-36 us of pure latency per
// launch, 6.5 % of the shipped 2.5D configuration's kernel time.)
__global__ void __launch_bounds__(256)
wgrad_stream_reduce(const float* __restrict__ part, int nblk, int npairs, int Cin, int Cout, float* __restrict__ dw,
                    int mid) {       // mid: only taps 9..17 were produced; dw is the 9-tap tensor [Cout][Cin][3][3]
  __shared__ float4 red[256];
  const int64_t total = (int64_t)npairs * 27 * 1024;
  const int o = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int64_t i = ((int64_t)blockIdx.x * 16 + o) * 4;
  const int tap0 = i < total ? (int)((i >> 10) % 27) : 0;
  const bool live = i < total && (!mid || (tap0 >= 9 && tap0 < 18));     // uniform per 64-element run (= per block)
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* p = part + i;
    int b = pl;
    for (; b + 48 < nblk; b += 64) {                 // four loads in flight, added in order
      const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)b * total);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (int64_t)(b + 16) * total);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(b + 32) * total);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (int64_t)(b + 48) * total);
      t.x += v0.x; t.y += v0.y; t.z += v0.z; t.w += v0.w;
      t.x += v1.x; t.y += v1.y; t.z += v1.z; t.w += v1.w;
      t.x += v2.x; t.y += v2.y; t.z += v2.z; t.w += v2.w;
      t.x += v3.x; t.y += v3.y; t.z += v3.z; t.w += v3.w;
    }
    for (; b < nblk; b += 16) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)b * total);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
  }
  red[pl * 16 + o] = t;
  __syncthreads();
  if (pl != 0 || !live) return;
  float out[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    const float4 r = red[k * 16 + o];
    out[0] += r.x; out[1] += r.y; out[2] += r.z; out[3] += r.w;
  }
  const int ci_l = i & 31, co_l = (i >> 5) & 31, tap = (int)((i >> 10) % 27), pair = (int)(i / (27 * 1024));
  const int ncit = Cin / 32;
  const int co = (pair / ncit) * 32 + co_l, ci = (pair % ncit) * 32 + ci_l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (mid) dw[((int64_t)co * Cin + ci + k) * 9 + tap - 9] = out[k];
    else dw[((int64_t)co * Cin + ci + k) * 27 + tap] = out[k];
  }
}

// The same sum for FEW partial blocks and a large gradient (the deep levels: 7-28 MB of dw, 1-8 partial blocks): above, a
// thread ends with four ci of ONE (co, tap) - four 4-byte stores 108 bytes apart, 7 M scattered stores for a 512 x 512 layer
// (55 us, a third of that layer's weight-gradient time).  Here a block owns one output row (pair, co): its 27 x 32 values are
// 27 runs of 128 bytes in every partial block (coalesced float4 loads, summed over the blocks in index order), transposed
// through LDS into the row's 864 CONTIGUOUS floats of dw[co][ci0 .. ci0 + 31][27] and written as 16-byte stores.
__global__ void __launch_bounds__(256)
wgrad_reduce_rows(const float* __restrict__ part, int nblk, int npairs, int Cin, int Cout, float* __restrict__ dw) {
  __shared__ __attribute__((aligned(16))) float tile[32 * 27 + 4];
  const int pair = blockIdx.x >> 5, co_l = blockIdx.x & 31;
  const int64_t total = (int64_t)npairs * 27 * 1024;
  const int t = threadIdx.x;
  if (t < 216) {
    const int tap = t >> 3, c4 = t & 7;
    const float* p = part + ((int64_t)pair * 27 + tap) * 1024 + co_l * 32 + c4 * 4;
    float4 a = *reinterpret_cast<const float4*>(p);
    for (int b = 1; b < nblk; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)b * total);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    tile[(4 * c4 + 0) * 27 + tap] = a.x;
    tile[(4 * c4 + 1) * 27 + tap] = a.y;
    tile[(4 * c4 + 2) * 27 + tap] = a.z;
    tile[(4 * c4 + 3) * 27 + tap] = a.w;
  }
  __syncthreads();
  if (t < 216) {
    const int ncit = Cin / 32;
    const int co = (pair / ncit) * 32 + co_l, ci0 = (pair % ncit) * 32;
    *reinterpret_cast<float4*>(dw + ((int64_t)co * Cin + ci0) * 27 + 4 * t) = *reinterpret_cast<const float4*>(tile + 4 * t);
  }
}
// partial blocks -> dw: the row kernel for few blocks and many outputs (3x3x3 form, dw 16-byte aligned), else the lane kernel
static inline void wgrad_finish(const float* part, int nblk, int npairs, int cin, int cout, float* dw, int mid, hipStream_t st) {
  const int64_t total = (int64_t)npairs * 27 * 1024;
  if (!mid && nblk <= (int)fplx_knob(FPLX_K_WG_REDUCE_ROWS) && npairs >= 16 && ((uintptr_t)dw % 16) == 0)
    wgrad_reduce_rows<<<(unsigned)(npairs * 32), 256, 0, st>>>(part, nblk, npairs, cin, cout, dw);
  else
    wgrad_stream_reduce<<<(unsigned)((total + 63) / 64), 256, 0, st>>>(part, nblk, npairs, cin, cout, dw, mid);
}

struct WgCfg { int tw, cit, cot, tilesH, tilesW, dsegs, dlen, nblk, npairs; size_t ws; };

static inline int64_t cot_env_min_vox() {
  return fplx_knob(FPLX_K_WG_COT_MINVOX);
}

inline WgCfg wg_cfg(int n, int d, int h, int w, int cin, int cout) {
  WgCfg c;
  // voxels are the K dimension here: padding the width to the tile is wasted MFMA work.  16-wide tiles when they pad
  // less and the rows are short (W = 40 at level 2: 48 instead of 64 columns, measured -24..-30 %); at W = 80 the
  // shorter k-loop per depth step costs as much as the padding saves (measured +-5 %), so 32 stays
  c.tw = (w >= 64 || (w >= 32 && (w + 15) / 16 * 16 >= (w + 31) / 32 * 32)) ? 32 : 16;
  {
    const int ktw = (int)fplx_knob(FPLX_K_WG_TW);   // tuning knob
    if (ktw == 16 || ktw == 32) c.tw = ktw;
  }
  c.tilesH = (h + WG_TH - 1) / WG_TH;
  c.tilesW = (w + c.tw - 1) / c.tw;
  c.npairs = (cin / 32) * (cout / 32);
  {
    const int cit_env = (int)fplx_knob(FPLX_K_WG_CIT);   // tuning knob
    // two ci tiles per block: dy is read once for both and x in whole 128-byte lines (-15 % at level 0/1); the small
    // deep volumes need the block count more (measured: slower below 32 K voxels per sample)
    c.cit = (cin % 64 == 0 && cit_env == 2 && (int64_t)d * h * w >= 32000) ? 2 : 1;
    // two co tiles per block instead where Cout allows: every x fragment then feeds two MFMAs (A/B knob FPLX_WG_COT)
    const int cot_env = (int)fplx_knob(FPLX_K_WG_COT);
    c.cot = 1;
    if (cout % 64 == 0 && cot_env == 2 && (cot_env_min_vox() <= (int64_t)d * h * w)) { c.cot = 2; c.cit = 1; }
  }
  const int tiles = n * c.tilesH * c.tilesW;
  // one block per CU at a time (LDS + 512-register waves): pick the depth split that minimises
  // rounds x (depths per block + per-block overhead).  The overhead - prologue slabs, the 110-KB partial tile
  // and its share of the reduction pass - is worth about nine depths (measured), so few long blocks win even when
  // they leave some CUs idle.
  int ds = 1;
  {
    double best = 1e30;
    const int e = (int)fplx_knob(FPLX_K_WG_DS);          // tuning knob (benchmarks only)
    for (int cand = 1; cand <= d; ++cand) {
      const int dl = (d + cand - 1) / cand;
      if (dl < 4 && cand > 1) break;
      const int segs = (d + dl - 1) / dl;
      const int64_t rounds = ((int64_t)tiles * (c.npairs / (c.cit * c.cot)) * segs + 255) / 256;
      const double cost = (double)rounds * (dl + 9.0);
      if (cost < best - 1e-9) { best = cost; ds = segs; }
    }
    if (e > 0) ds = e;
  }
  c.dlen = (d + ds - 1) / ds;
  c.dsegs = (d + c.dlen - 1) / c.dlen;
  c.nblk = tiles * c.dsegs;
  c.ws = (size_t)c.nblk * c.npairs * 27 * 1024 * sizeof(float);
  return c;
}

// ------------------------------------------------------------------------------------------
// conv_wgrad_vox: the same weight gradient for the SMALL volumes of the deep levels (2 x 20 x 40 x 40 and below), where the
// footprint march above pads its 8 x 16 tiles 1.2-2.6 x and writes one 110-KB partial tile per footprint and channel
// pair (level 4: 112 MB of partials for 2 MB of operands).  Here the reduction dimension is the LINEAR voxel index: a
// block owns a channel-pair group (32 ci x 32 COT co, all 27 taps dealt to its 4 waves exactly as above - same
// accumulators, same partial-tile format, same reduction kernel) and a contiguous voxel range [k0, k1) of the whole
// batch, walked in chunks of 128 voxels = 8 k-steps.  In linear order a tap is a constant row shift
// (kd - 1) H W + (kh - 1) W + (kw - 1); what the shift must not do is wrap around a row, a slice or a sample:
//   * depth: the three kd planes are staged separately (rows [chunk - W - 1, chunk + 128 + W + 1) shifted by -HW, 0, +HW);
//     a source row whose depth cannot belong to a valid pair (d = D - 1 for kd = 0, d = 0 for kd = 2), or that lies outside
//     the tensor, is staged as zeros;
//   * height / width: an x fragment holds 8 consecutive voxels per lane, so the pairs whose OUTPUT voxel sits on the
//     border the tap leaves through (h = 0 for kh = 0, h = H - 1 for kh = 2, likewise w) are removed by one 16-byte AND
//     mask per (tap column, 8-voxel group), built per chunk from ballots over a per-voxel border code (a byte table the
//     block computes once with multiply-high divisions) - 4 VALU per fragment, no second copy of x in LDS.
// No padding waste (every MFMA row is a real voxel), K-split only as far as needed to fill the chip, x and dy of a deep
// level (4-8 MB) stay in L2 / Infinity Cache across the pair groups.
constexpr int VX_KC = 128, VX_MAXW = 40, VX_XR_MAX = VX_KC + 2 * VX_MAXW + 2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;

struct VoxGeo { int V, D, H, W, HW; unsigned mW, mH, mD; int kper, S; };   // mX = ceil(2^32 / X): q = umulhi(u, mX) for u < 2^26

// LDS of conv_wgrad_vox: two slots of {three x planes, COT dy planes, the chunk's masks} + the border-code table.  The
// slot strides are compile-time constants (W enters only the row count inside a plane), so the slot is an immediate in
// every LDS read of the unrolled chunk.
template <int COT>
struct VXL {
  static constexpr int X_SLOT = 3 * VX_XR_MAX * 64, DY_SLOT = COT * VX_KC * 64, MK_SLOT = 9 * 16 * 16;
  static constexpr int X0 = 0, DY0 = 2 * X_SLOT, MK0 = DY0 + 2 * DY_SLOT, VC0 = MK0 + 2 * MK_SLOT;
};

// what the staging of the NEXT chunk needs (wave-uniform unless noted)
struct VoxStage {
  const bf16_t* xg;        // x + this block's 32 input channels
  const bf16_t* dyg;       // dy + this block's 32 COT output channels
  int64_t ldx, ldy;
  const unsigned char* vc; // border codes, entry t <-> voxel tb0 + t
  int tb0, tbn, XR, W, HW, k1;
};

// staging item m (0 .. NLX + NLY - 1) of chunk kn: x rows first (16-byte column tid & 3 of row tid / 4 + 64 m of the three
// planes; a row that must read as zeros loads voxel 0 and is cleared by selects - no divergent control flow, and a plain
// global load: a pointer select with a zero constant would make it a FLAT one), then dy
template <int COT>
struct VXS {
  static constexpr int NLX = (3 * VX_XR_MAX * 4 + 255) / 256, NLY = VX_KC * 4 * COT / 256, NA = 7, NB = NLX + NLY - NA;
  static_assert(NB <= 7 && NB >= 0, "two batches of at most 7 items");
  static __device__ __forceinline__ uint4 load(const VoxStage& sg, int kn, int tid, int m) {
    uint4 v;
    bool ok;
    if (m < NLX) {
      const int rr = (tid >> 2) + m * 64;
      const int pl = (rr >= sg.XR ? 1 : 0) + (rr >= 2 * sg.XR ? 1 : 0), row = rr - pl * sg.XR;
      const int u = kn + row - (sg.W + 1) + (pl - 1) * sg.HW;
      const int t = u - sg.tb0;
      const bool in = rr < 3 * sg.XR && t >= 0 && t < sg.tbn;
      const unsigned c = sg.vc[in ? t : 0];
      ok = in && c != 0xFFu && !((pl == 0 && (c & 2u)) || (pl == 2 && (c & 1u)));
      v = *reinterpret_cast<const uint4*>(sg.xg + (int64_t)(ok ? u : 0) * sg.ldx + (tid & 3) * 8);
    } else {
      const int i = tid + (m - NLX) * 256;
      const int vox = i / (4 * COT), ch = i % (4 * COT);
      const int v_ = kn + vox;
      ok = v_ < sg.k1;
      v = *reinterpret_cast<const uint4*>(sg.dyg + (int64_t)(ok ? v_ : 0) * sg.ldy + ch * 8);
    }
    return ok ? v : make_uint4(0, 0, 0, 0);
  }
  static __device__ __forceinline__ void store(const VoxStage& sg, char* xw_n, char* dy_n, int tid, int m, const uint4& v) {
    if (m < NLX) {
      const int i = tid + m * 256;
      if (i < 3 * sg.XR * 4) *reinterpret_cast<uint4*>(xw_n + i * 16) = v;     // planes are contiguous: row i / 4, column i % 4
    } else {
      const int i = tid + (m - NLX) * 256;
      const int vox = i / (4 * COT), ch = i % (4 * COT);
      *reinterpret_cast<uint4*>(dy_n + (ch >> 2) * (VX_KC * 64) + vox * 64 + (ch & 3) * 16) = v;
    }
  }
  // the chunk's masks: wave wv (0 or 1) takes 64 voxels; lane e -> (combo e / 8, octet e % 8), then combo 8
  static __device__ __forceinline__ void masks(const VoxStage& sg, char* mk_n, int kn, int wv, int lane) {
    const unsigned c = sg.vc[kn - sg.tb0 + wv * 64 + lane];
    const uint64_t h0 = __ballot(c & 4), h2 = __ballot(c & 8), w0 = __ballot(c & 16), w2 = __ballot(c & 32);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      const int e = rnd * 64 + lane;
      if (e < 72) {
        const int combo = e >> 3, jj = e & 7, kh = combo / 3, kw = combo - 3 * kh;
        const uint64_t inv = (kh == 0 ? h0 : (kh == 2 ? h2 : 0)) | (kw == 0 ? w0 : (kw == 2 ? w2 : 0));
        const unsigned by = (unsigned)(inv >> (8 * jj)) & 0xFFu;
        u32x4v m_;
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_)
          m_[q_] = ((by >> (2 * q_)) & 1u ? 0u : 0xFFFFu) | ((by >> (2 * q_ + 1)) & 1u ? 0u : 0xFFFF0000u);
        *reinterpret_cast<u32x4v*>(mk_n + (combo * 16 + wv * 8 + jj) * 16) = m_;
      }
    }
  }
};

// One chunk of 128 voxels for wave WV out of slot SL; meanwhile the next chunk (kn >= 0) travels global -> registers ->
// slot SL ^ 1 in two batches of at most 7 sixteen-byte items per lane, issued and committed at fixed items of the MFMA
// stream (one wave per SIMD: everything that is not an MFMA sits in an MFMA gap), and waves 0 / 1 build its masks.
// NKS: k-steps computed (8 = the whole chunk).  The block's FIRST chunk is staged by a pass with NKS = 1 over a zeroed slot 0
// (14 MFMAs that add zeros): a staging-only prologue beside the march makes hipcc move the accumulators out of the AGPRs
// (160 spills), the same code as one more link of the MFMA chain does not.
template <int WV, int COT, int SL, int NKS = VX_KC / 16>
__device__ __forceinline__ void wgrad_vox_chunk(f32x16 (&acc)[7 * COT], const char* const (&xb)[7], const char* dyb,
                                                const char* mkb, char* smem, const VoxStage& sg, int kn, int tid, int lane) {
  using L = VXL<COT>;
  using S = VXS<COT>;
  constexpr int NT = (27 - WV + 3) / 4;               // 7 taps (6 for wave 3): tap = WV + 4 i
  constexpr int DYPLANE = VX_KC * 64, NI = NKS * NT;
  constexpr int XOFF = SL * L::X_SLOT, DOFF = SL * L::DY_SLOT, MOFF = SL * L::MK_SLOT;
  constexpr int PD = 3, RING = PD + 1;
  constexpr int NA = S::NA, NB = S::NB;
  // commit A + issue B | commit B | masks (A is issued at item 0); positions that exist for the 6-tap wave too
  constexpr int T_A1 = NI < 18 ? NI : 18, T_B1 = NI < 38 ? NI : 38, T_MK = NI < 42 ? NI : 42;
  bf16x8 fbw[2][COT], far[RING];
  u32x4v mkr[RING];
  uint4 st[7];
  const bool more = kn >= 0;
  char* xw_n = smem + L::X0 + (SL ^ 1) * L::X_SLOT;
  char* dy_n = smem + L::DY0 + (SL ^ 1) * L::DY_SLOT;
  char* mk_n = smem + L::MK0 + (SL ^ 1) * L::MK_SLOT;
  auto masked = [](int j) { const int tap = WV + 4 * j; return ((tap / 3) % 3) != 1 || (tap % 3) != 1; };
  auto load_item = [&](int t) {                        // t = ks * NT + j
    const int ks = t / NT, j = t % NT, tap = WV + 4 * j;
    far[t % RING] = tr_frag(xb[j] + XOFF + ks * 1024);
    if (masked(j)) mkr[t % RING] = *reinterpret_cast<const u32x4v*>(mkb + MOFF + ((tap % 9) * 16 + 2 * ks) * 16);
  };
#pragma unroll
  for (int o = 0; o < COT; ++o) fbw[0][o] = tr_frag(dyb + DOFF + o * DYPLANE);
#pragma unroll
  for (int t = 0; t < PD; ++t) load_item(t);
  auto item = [&](int t) {
    const int ks = t / NT, j = t % NT;
    if (t + PD < NI) load_item(t + PD);
    if (j < COT && ks + 1 < NKS) fbw[(ks + 1) & 1][j] = tr_frag(dyb + DOFF + j * DYPLANE + (ks + 1) * 1024);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a = far[t % RING];
    if (masked(j)) a = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4v, a) & mkr[t % RING]);
#pragma unroll
    for (int o = 0; o < COT; ++o)
      acc[o * 7 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fbw[ks & 1][o], acc[o * 7 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // straight-line segments of the item stream (each loop has constant bounds and a small body: fully unrolled, so every
  // ring index, tap and LDS offset is a constant), the staging steps between them
  if (more) {
#pragma unroll
    for (int m = 0; m < NA; ++m) st[m] = S::load(sg, kn, tid, m);
  }
#pragma unroll
  for (int t = 0; t < T_A1; ++t) item(t);
  if (more) {
#pragma unroll
    for (int m = 0; m < NA; ++m) S::store(sg, xw_n, dy_n, tid, m, st[m]);
#pragma unroll
    for (int m = 0; m < NB; ++m) st[m] = S::load(sg, kn, tid, NA + m);
  }
#pragma unroll
  for (int t = T_A1; t < T_B1; ++t) item(t);
  if (more) {
#pragma unroll
    for (int m = 0; m < NB; ++m) S::store(sg, xw_n, dy_n, tid, NA + m, st[m]);
  }
#pragma unroll
  for (int t = T_B1; t < T_MK; ++t) item(t);
  if (more && WV < 2) S::masks(sg, mk_n, kn, WV, lane);
#pragma unroll
  for (int t = T_MK; t < NI; ++t) item(t);
}

template <int WV, int COT>
__device__ __forceinline__ void wgrad_vox_march(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy,
                                                int64_t ldy, float* __restrict__ part, int Cin, int Cout, const VoxGeo g,
                                                const FplxBlock bid) {
  using L = VXL<COT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = g.W, HW = g.HW, V = g.V;
  const int XR = VX_KC + 2 * W + 2, xplane = XR * 64;
  unsigned char* vc = reinterpret_cast<unsigned char*>(smem + L::VC0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int ncit = Cin / 32;
  const int cot = bid.y / ncit, cit = bid.y % ncit;
  const int k0 = bid.x * g.kper;
  const int k1 = (k0 + g.kper < V) ? k0 + g.kper : V;

  // border codes of the voxels this block can touch: bit 0 d == 0, 1 d == D-1, 2 h == 0, 3 h == H-1, 4 w == 0, 5 w == W-1
  const int halo = HW + W + 1;
  const int tb0 = k0 - halo, tbn = (k1 - k0 + VX_KC - 1) / VX_KC * VX_KC + 2 * halo;
  for (int t = tid; t < tbn; t += 256) {
    const int u = tb0 + t;
    unsigned char c = 0xFF;
    if (u >= 0 && u < V) {
      const unsigned q1 = __umulhi((unsigned)u, g.mW), w_ = (unsigned)u - q1 * (unsigned)W;
      const unsigned q2 = __umulhi(q1, g.mH), h_ = q1 - q2 * (unsigned)g.H;
      const unsigned q3 = __umulhi(q2, g.mD), d_ = q2 - q3 * (unsigned)g.D;
      c = (unsigned char)((d_ == 0 ? 1 : 0) | (d_ == (unsigned)g.D - 1 ? 2 : 0) | (h_ == 0 ? 4 : 0) |
                          (h_ == (unsigned)g.H - 1 ? 8 : 0) | (w_ == 0 ? 16 : 0) | (w_ == (unsigned)W - 1 ? 32 : 0));
    }
    vc[t] = c;
  }
  VoxStage sg;
  sg.xg = x + cit * 32; sg.dyg = dy + cot * (32 * COT); sg.ldx = ldx; sg.ldy = ldy; sg.vc = vc;
  sg.tb0 = tb0; sg.tbn = tbn; sg.XR = XR; sg.W = W; sg.HW = HW; sg.k1 = k1;
  for (int i = tid; i < L::VC0 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);   // slot 0 reads as zeros

  // transposed-read lane geometry (tr_frag): group gq = lane / 16 -> k-octet gq >> 1, channel half gq & 1
  const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int lane_off = (8 * (gq >> 1) + q) * 64 + (16 * (gq & 1) + 4 * pp) * 2;
  constexpr int NT = (27 - WV + 3) / 4;
  const char* xb[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = j < NT ? WV + 4 * j : WV;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    xb[j] = smem + L::X0 + kd * xplane + (kh * W + kw) * 64 + lane_off;
  }
  const char* dyb = smem + L::DY0 + lane_off;
  const char* mkb = smem + L::MK0 + (gq >> 1) * 16;

  f32x16 acc[7 * COT];
#pragma unroll
  for (int i = 0; i < 7 * COT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  int kc = k0;
  __syncthreads();                                         // the code table and the zeroed slot are complete
  wgrad_vox_chunk<WV, COT, 0, 1>(acc, xb, dyb, mkb, smem, sg, k0, tid, lane);      // stages chunk k0 into slot 1
  __syncthreads();
  for (;;) {
    {
      const int kn = kc + VX_KC < k1 ? kc + VX_KC : -1;
      wgrad_vox_chunk<WV, COT, 1>(acc, xb, dyb, mkb, smem, sg, kn, tid, lane);
      __syncthreads();
      kc += VX_KC;
      if (kc >= k1) break;
    }
    {
      const int kn = kc + VX_KC < k1 ? kc + VX_KC : -1;
      wgrad_vox_chunk<WV, COT, 0>(acc, xb, dyb, mkb, smem, sg, kn, tid, lane);
      __syncthreads();
      kc += VX_KC;
      if (kc >= k1) break;
    }
  }
  // partial tiles, the format of conv_wgrad_stream: part[split][pair][tap][co][ci]
  const int co = lane & 31, rbase = (lane >> 5) * 4;
#pragma unroll
  for (int o = 0; o < COT; ++o) {
    const int pair = (cot * COT + o) * ncit + cit;
    float* out = part + ((int64_t)bid.x * (ncit * (Cout / 32)) + pair) * (27 * 1024);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int tap = WV + 4 * i;
      if (tap < 27) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(out + (tap * 32 + co) * 32 + 8 * g4 + rbase) =
              make_float4(acc[o * 7 + i][4 * g4 + 0], acc[o * 7 + i][4 * g4 + 1], acc[o * 7 + i][4 * g4 + 2],
                          acc[o * 7 + i][4 * g4 + 3]);
      }
    }
  }
}

template <int COT>
__global__ void __launch_bounds__(256)
conv_wgrad_vox(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
               float* __restrict__ part, int Cin, int Cout, VoxGeo g, int xcd) {
  const FplxBlock bid = fplx_xcd_block(xcd);               // y fastest: the pair groups of one voxel range share an L2
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: wgrad_vox_march<0, COT>(x, ldx, dy, ldy, part, Cin, Cout, g, bid); break;
    case 1: wgrad_vox_march<1, COT>(x, ldx, dy, ldy, part, Cin, Cout, g, bid); break;
    case 2: wgrad_vox_march<2, COT>(x, ldx, dy, ldy, part, Cin, Cout, g, bid); break;
    default: wgrad_vox_march<3, COT>(x, ldx, dy, ldy, part, Cin, Cout, g, bid); break;
  }
}

// ------------------------------------------------------------------------------------------
// conv_wgrad_vox_lw (round 5): the voxel-GEMM weight gradient with DEDICATED LOADER WAVES (the form of conv_fwd_brick_lw).
// conv_wgrad_vox stages the next chunk global -> registers -> LDS from its four computing waves, two batches of seven 16-byte
// loads per lane issued and committed at fixed items of a 112-MFMA chunk: 600 cycles between a load and its commit against an
// L2 latency of 1-2 K cycles under load - the level-4 512 -> 512 layer ran 42 us for 14 us of MFMA issue (0.05-0.12 of peak:
// the lowest row of the round-5 kernel table).  Here a block is 8 waves:
//   waves 0-3 (compute): the item stream of wgrad_vox_chunk - transposed fragment reads, border masks, MFMAs - and nothing
//   else; one co tile per block (7 accumulator tiles per wave: two waves per SIMD have 256 registers), so a layer has twice
//   the blocks of conv_wgrad_vox<2>;
//   waves 4-7 (loaders): the next chunk's three x planes and its dy rows by LDS-DMA through buffer descriptors (a row that
//   must read as zeros - outside the tensor, a depth that cannot pair - gets an out-of-range offset), the chunk's masks
//   (waves 4, 5), s_waitcnt, the chunk's barrier.
// Same partial-tile format and finish (wgrad_finish) as conv_wgrad_vox; sums in the same order per tile (chunks ascending).
template <int WV, int SL>
__device__ __forceinline__ void wgrad_vox_lw_chunk(f32x16 (&acc)[7], const char* const (&xb)[7], const char* dyb, const char* mkb) {
  using L = VXL<1>;
  constexpr int NT = (27 - WV + 3) / 4, NKS = VX_KC / 16, NI = NKS * NT;
  constexpr int XOFF = SL * L::X_SLOT, DOFF = SL * L::DY_SLOT, MOFF = SL * L::MK_SLOT;
  constexpr int PD = 3, RING = PD + 1;
  bf16x8 fbw[2], far[RING];
  u32x4v mkr[RING];
  auto masked = [](int j) { const int tap = WV + 4 * j; return ((tap / 3) % 3) != 1 || (tap % 3) != 1; };
  auto load_item = [&](int t) {
    const int ks = t / NT, j = t % NT, tap = WV + 4 * j;
    far[t % RING] = tr_frag(xb[j] + XOFF + ks * 1024);
    if (masked(j)) mkr[t % RING] = *reinterpret_cast<const u32x4v*>(mkb + MOFF + ((tap % 9) * 16 + 2 * ks) * 16);
  };
  fbw[0] = tr_frag(dyb + DOFF);
#pragma unroll
  for (int t = 0; t < PD; ++t) load_item(t);
#pragma unroll
  for (int t = 0; t < NI; ++t) {
    const int ks = t / NT, j = t % NT;
    if (t + PD < NI) load_item(t + PD);
    if (j == 0 && ks + 1 < NKS) fbw[(ks + 1) & 1] = tr_frag(dyb + DOFF + (ks + 1) * 1024);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a = far[t % RING];
    if (masked(j)) a = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4v, a) & mkr[t % RING]);
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fbw[ks & 1], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int WV>
__device__ __forceinline__ void wgrad_vox_lw_compute(float* __restrict__ part, int Cin, int Cout, const VoxGeo& g, const FplxBlock& bid,
                                                     char* smem, int lane, int k0, int k1) {
  using L = VXL<1>;
  const int W = g.W, XR = VX_KC + 2 * W + 2, xplane = XR * 64;
  const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int lane_off = (8 * (gq >> 1) + q) * 64 + (16 * (gq & 1) + 4 * pp) * 2;
  constexpr int NT = (27 - WV + 3) / 4;
  const char* xb[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = j < NT ? WV + 4 * j : WV;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    xb[j] = smem + L::X0 + kd * xplane + (kh * W + kw) * 64 + lane_off;
  }
  const char* dyb = smem + L::DY0 + lane_off;
  const char* mkb = smem + L::MK0 + (gq >> 1) * 16;
  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // P0: chunk k0 sits in slot 0
  for (int kc = k0;;) {
    wgrad_vox_lw_chunk<WV, 0>(acc, xb, dyb, mkb);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    kc += VX_KC;
    if (kc >= k1) break;
    wgrad_vox_lw_chunk<WV, 1>(acc, xb, dyb, mkb);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    kc += VX_KC;
    if (kc >= k1) break;
  }
  // partial tiles, the format of conv_wgrad_stream: part[split][pair][tap][co][ci]
  const int ncit = Cin / 32;
  const int cot = bid.y / ncit, cit = bid.y % ncit;
  const int co = lane & 31, rbase = (lane >> 5) * 4;
  const int pair = cot * ncit + cit;
  float* out = part + ((int64_t)bid.x * (ncit * (Cout / 32)) + pair) * (27 * 1024);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = WV + 4 * i;
    if (tap < 27) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(out + (tap * 32 + co) * 32 + 8 * g4 + rbase) =
            make_float4(acc[i][4 * g4 + 0], acc[i][4 * g4 + 1], acc[i][4 * g4 + 2], acc[i][4 * g4 + 3]);
    }
  }
}

__global__ void __launch_bounds__(512)
conv_wgrad_vox_lw(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
                  float* __restrict__ part, int Cin, int Cout, const VoxGeo g, int xcd) {
  using L = VXL<1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FplxBlock bid = fplx_xcd_block(xcd);
  const int W = g.W, HW = g.HW, V = g.V;
  const int XR = VX_KC + 2 * W + 2;
  unsigned char* vc = reinterpret_cast<unsigned char*>(smem + L::VC0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncit = Cin / 32;
  const int cot = bid.y / ncit, cit = bid.y % ncit;
  const int k0 = bid.x * g.kper;
  const int k1 = (k0 + g.kper < V) ? k0 + g.kper : V;
  // border codes of the voxels this block can touch (as conv_wgrad_vox), by all 8 waves
  const int halo = HW + W + 1;
  const int tb0 = k0 - halo, tbn = (k1 - k0 + VX_KC - 1) / VX_KC * VX_KC + 2 * halo;
  for (int t = tid; t < tbn; t += 512) {
    const int u = tb0 + t;
    unsigned char c = 0xFF;
    if (u >= 0 && u < V) {
      const unsigned q1 = __umulhi((unsigned)u, g.mW), w_ = (unsigned)u - q1 * (unsigned)W;
      const unsigned q2 = __umulhi(q1, g.mH), h_ = q1 - q2 * (unsigned)g.H;
      const unsigned q3 = __umulhi(q2, g.mD), d_ = q2 - q3 * (unsigned)g.D;
      c = (unsigned char)((d_ == 0 ? 1 : 0) | (d_ == (unsigned)g.D - 1 ? 2 : 0) | (h_ == 0 ? 4 : 0) |
                          (h_ == (unsigned)g.H - 1 ? 8 : 0) | (w_ == 0 ? 16 : 0) | (w_ == (unsigned)W - 1 ? 32 : 0));
    }
    vc[t] = c;
  }
  __syncthreads();                                         // the code table is complete
  if (wave8 < 4) {
    switch (wave8) {
      case 0: wgrad_vox_lw_compute<0>(part, Cin, Cout, g, bid, smem, lane, k0, k1); break;
      case 1: wgrad_vox_lw_compute<1>(part, Cin, Cout, g, bid, smem, lane, k0, k1); break;
      case 2: wgrad_vox_lw_compute<2>(part, Cin, Cout, g, bid, smem, lane, k0, k1); break;
      default: wgrad_vox_lw_compute<3>(part, Cin, Cout, g, bid, smem, lane, k0, k1); break;
    }
    return;
  }
  // ================================================================ loader waves
  const int lw = wave8 - 4;
  const bf16_t* xg = x + cit * 32;
  const bf16_t* dyg = dy + cot * 32;
  u32x4v rx, ry;
  rx[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)xg);
  rx[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)xg >> 32) & 0xFFFFu);
  rx[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)V * ldx * 2));
  rx[3] = 0x00020000u;
  ry[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dyg);
  ry[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)dyg >> 32) & 0xFFFFu);
  ry[2] = __builtin_amdgcn_readfirstlane((unsigned)((int64_t)V * ldy * 2));
  ry[3] = 0x00020000u;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  auto buf_dma = [&](const u32x4v& rsrc, unsigned vo, unsigned dst_off) {
    const unsigned dst = lds0 + dst_off;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(0u), "s"(dst) : "memory");
  };
  const int nxp = (3 * XR + 15) / 16;                      // x pieces of a chunk (16 rows of 64 bytes each), then 8 dy pieces
  auto stage = [&](int kn, int slot) {
    for (int p = lw; p < nxp + VX_KC / 16; p += 4) {
      if (p < nxp) {
        const int rr = p * 16 + (lane >> 2);
        const int pl = (rr >= XR ? 1 : 0) + (rr >= 2 * XR ? 1 : 0), row = rr - pl * XR;
        const int u = kn + row - (W + 1) + (pl - 1) * HW;
        const int t = u - tb0;
        const bool in = t >= 0 && t < tbn;
        const unsigned c = vc[in ? t : 0];
        const bool ok = in && c != 0xFFu && !((pl == 0 && (c & 2u)) || (pl == 2 && (c & 1u)));
        const unsigned vo = ok ? (unsigned)(((int64_t)u * ldx + (lane & 3) * 8) * 2) : 0x80000000u;
        if (rr < 3 * XR) buf_dma(rx, vo, (unsigned)(L::X0 + slot * L::X_SLOT + p * 1024));      // (rows past the planes: not written)
      } else {
        const int v_ = kn + (p - nxp) * 16 + (lane >> 2);
        const unsigned vo = v_ < k1 ? (unsigned)(((int64_t)v_ * ldy + (lane & 3) * 8) * 2) : 0x80000000u;
        buf_dma(ry, vo, (unsigned)(L::DY0 + slot * L::DY_SLOT + (p - nxp) * 1024));
      }
    }
    if (lw < 2) {                                          // the chunk's masks (VXS::masks): wave lw takes 64 voxels
      const unsigned c = vc[kn - tb0 + lw * 64 + lane];
      const uint64_t h0 = __ballot(c & 4), h2 = __ballot(c & 8), w0 = __ballot(c & 16), w2 = __ballot(c & 32);
      char* mk_n = smem + L::MK0 + slot * L::MK_SLOT;
#pragma unroll
      for (int rnd = 0; rnd < 2; ++rnd) {
        const int e = rnd * 64 + lane;
        if (e < 72) {
          const int combo = e >> 3, jj = e & 7, kh = combo / 3, kw = combo - 3 * kh;
          const uint64_t inv = (kh == 0 ? h0 : (kh == 2 ? h2 : 0)) | (kw == 0 ? w0 : (kw == 2 ? w2 : 0));
          const unsigned by = (unsigned)(inv >> (8 * jj)) & 0xFFu;
          u32x4v m_;
#pragma unroll
          for (int q_ = 0; q_ < 4; ++q_)
            m_[q_] = ((by >> (2 * q_)) & 1u ? 0u : 0xFFFFu) | ((by >> (2 * q_ + 1)) & 1u ? 0u : 0xFFFF0000u);
          *reinterpret_cast<u32x4v*>(mk_n + (combo * 16 + lw * 8 + jj) * 16) = m_;
        }
      }
    }
  };
  stage(k0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // P0
  for (int kc = k0, slot = 0;; slot ^= 1) {
    const int kn = kc + VX_KC;
    if (kn < k1) stage(kn, slot ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    kc = kn;
    if (kc >= k1) break;
  }
}

struct VoxCfg { int ok, cot, npairs; VoxGeo g; size_t ws, lds; };
// deep levels only: small volumes (the footprint march wins where its tiles fit and the voxel count amortises its
// partial tiles), W <= 40 (LDS windows), 3 x 3 x 3 in all three dimensions (not the 2.5D middle-plane form)
inline VoxCfg vox_cfg(int n, int d, int h, int w, int cin, int cout) {
  VoxCfg c = {};
  const int64_t V = (int64_t)n * d * h * w;
  const int mode = (int)fplx_knob(FPLX_K_WG_VOX);           // 0: never, 1: the rule below, 2: wherever the kernel can run (tests)
  if (!mode || cin % 32 != 0 || cout % 32 != 0 || w > VX_MAXW || w < 2 || h < 2 || d < 2 || V >= ((int64_t)1 << 24)) return c;
  if (mode == 1 && V > (int64_t)fplx_knob(FPLX_K_WG_VOX_MAXV)) return c;
  // the loader-wave form (conv_wgrad_vox_lw) has one co tile per block: every x fragment and mask then feeds ONE MFMA and
  // the LDS (2 transposed reads + 1 mask read per MFMA and wave) is as busy as the matrix pipe - it wins only where the
  // register-staged kernel is latency-bound outright, the tiny level-4 volumes (512 -> 512: 57 -> 43 us; level 3 512 -> 256:
  // 150 -> 175).  Knob wg_vox_lw: 1 = volumes of at most 2048 voxels, 2 = everywhere (tests), 0 = never
  const int lwk = (int)fplx_knob(FPLX_K_WG_VOX_LW);
  const bool lw = lwk == 2 || (lwk == 1 && V <= 2048);
  c.cot = (cout % 64 == 0 && !lw) ? 2 : 1;
  c.npairs = (cin / 32) * (cout / 32);
  const int groups = c.npairs / c.cot;
  const int chunks = (int)((V + VX_KC - 1) / VX_KC);
  const int cus = (int)fplx_knob(FPLX_K_WG_VOX_CUS);         // blocks the launch aims at (256: every CU one)
  int S = (cus + groups - 1) / groups;                       // just enough voxel ranges to give every CU a block
  if (S > chunks) S = chunks;
  if (S < 1) S = 1;
  const int cper = (chunks + S - 1) / S;
  S = (chunks + cper - 1) / cper;
  VoxGeo& g = c.g;
  g.V = (int)V; g.D = d; g.H = h; g.W = w; g.HW = h * w;
  g.mW = (unsigned)((((uint64_t)1 << 32) + w - 1) / w);
  g.mH = (unsigned)((((uint64_t)1 << 32) + h - 1) / h);
  g.mD = (unsigned)((((uint64_t)1 << 32) + d - 1) / d);
  g.kper = cper * VX_KC; g.S = S;
  c.ws = (size_t)S * c.npairs * 27 * 1024 * sizeof(float);
  c.lds = (size_t)(c.cot == 2 ? VXL<2>::VC0 : VXL<1>::VC0) + (size_t)(g.kper + 2 * (g.HW + w + 1)) + 16;
  c.lds = (c.lds + 15) / 16 * 16;
  if (c.lds > 160 * 1024) return c;
  c.ok = 1;
  return c;
}

// ------------------------------------------------------------------------------------------
// ConvTranspose3d(k=2,s=2) forward: rows = input voxels, K = Cin, one 32x32 accumulator per tap
// (4 taps per block, blockIdx.z picks the half); every result row is scattered to its own output
// voxel (2d+i, 2h+j, 2w+k) of the concat buffer.  HBM-bound (8 output voxels per input voxel).
__global__ void __launch_bounds__(DIRECT_THREADS)
deconv_fwd_mfma(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wf,
                const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                int Cin, int Cout, int sd, int vec_ok, int xcd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kh8 = (lane >> 5) * 8;
  const int64_t V = (int64_t)N * D * H * W;
  const FplxBlock bid = fplx_xcd_block(xcd);               // the cout blocks / depth taps of one voxel tile share an L2
  const int64_t m0 = ((int64_t)bid.x * 4 + wave) * 32;
  const int n0 = bid.y * 32, tap0 = bid.z * 4;
  const int64_t v = m0 + r;
  const bool ok = v < V;
  const bf16_t* ap = x + (ok ? v : 0) * ldx + kh8;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // Cin % 32 == 0 (fplx_mfma_deconv2_fwd): two k-steps per trip, all ten loads issued before the eight MFMAs
  const bf16_t* bp = wf + ((int64_t)tap0 * Cout + n0 + r) * Cin + kh8;
  const int64_t tstride = (int64_t)Cout * Cin;
  for (int kc = 0; kc < Cin; kc += 32) {
    bf16x8 a[2], b[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      a[u] = ok ? *reinterpret_cast<const bf16x8*>(ap + kc + 16 * u) : zero;
#pragma unroll
      for (int t = 0; t < 4; ++t) b[u][t] = *reinterpret_cast<const bf16x8*>(bp + t * tstride + kc + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u][t], acc[t], 0, 0, 0);
  }
  const int co = n0 + r, rh = (lane >> 5) * 4;
  const float bv = bias[co];
  if (vec_ok) {
    // the accumulator layout is lane = channel, registers = voxels: 2-byte pieces per lane.  As global stores that is 64
    // instructions per wave and the store path sets the pace (the kernel ran at 2.4x its memory time); each tap's
    // 32 x 32 tile goes through a 2-KB per-wave LDS transpose instead and leaves as 16-byte stores.
    __shared__ __attribute__((aligned(16))) char stg_all[4][32 * 64];
    char* stg = stg_all[wave];
    int64_t obase[2];
    bool ook[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {                    // this lane stores voxels (lane >> 2) and (lane >> 2) + 16
      int64_t vv = m0 + (lane >> 2) + 16 * half;
      ook[half] = vv < V;
      if (!ook[half]) vv = 0;
      unsigned q = (unsigned)vv;                             // V < 2^31 (host check): 32-bit divisions
      const int w0 = (int)(q % (unsigned)W); q /= (unsigned)W;
      const int h0 = (int)(q % (unsigned)H); q /= (unsigned)H;
      const int d0 = (int)(q % (unsigned)D); q /= (unsigned)D;
      obase[half] = (((int64_t)q * sd * D + sd * d0) * 2 * H + 2 * h0) * 2 * W + 2 * w0;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tap = tap0 + t;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        *reinterpret_cast<bf16_t*>(stg + ((i & 3) + 8 * (i >> 2) + rh) * 64 + r * 2) = (bf16_t)(acc[t][i] + bv);
      const int64_t toff = ((int64_t)(tap >> 2) * 2 * H + ((tap >> 1) & 1)) * 2 * W + (tap & 1);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint4 v = *reinterpret_cast<const uint4*>(stg + ((lane >> 2) + 16 * half) * 64 + (lane & 3) * 16);
        if (ook[half]) *reinterpret_cast<uint4*>(y + (obase[half] + toff) * ldy + n0 + (lane & 3) * 8) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + rh;
    int64_t vv = m0 + row;
    if (vv < V) {
      const int w0 = (int)(vv % W); vv /= W;
      const int h0 = (int)(vv % H); vv /= H;
      const int d0 = (int)(vv % D); vv /= D;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int tap = tap0 + t;
        const int64_t ov = ((vv * sd * D + sd * d0 + (tap >> 2)) * 2 * H + 2 * h0 + ((tap >> 1) & 1)) * 2 * W + 2 * w0 + (tap & 1);
        y[ov * ldy + co] = (bf16_t)(acc[t][i] + bv);
      }
    }
  }
}

// deconv_fwd_rows: ConvTranspose3d(k=2,s=2) forward as a streaming kernel for the shallow levels (Cin <= 128, Cout 32 | 64),
// where the op is pure HBM traffic (up1 of the benchmark: 66 MB in, 262 MB out; deconv_fwd_mfma above: 118 us = 2.8 TB/s,
// short-lived waves with one tile each, half-line stores per tap).  Persistent blocks walk tiles of 32 consecutive input
// voxels; wave (i, j) of a block owns the output rows (2d + i, 2h + j): its taps k = 0, 1 are the two w-neighbours
// (2w, 2w + 1), i.e. one contiguous piece of 2 * Cout channels per input voxel.  The wave's weight fragments (2 taps x
// Cout / 32 x Cin / 16) live in registers for the whole kernel, the next tile's x rows are requested before the current
// tile's MFMAs, and the result leaves through an LDS tile [voxel][k][Cout] as 16-byte stores whose lanes run along
// whole output rows.
template <int KS, int NTC>        // Cin / 16, Cout / 32
__global__ void __launch_bounds__(256)
deconv_fwd_rows(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wf, const float* __restrict__ bias,
                bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int xcd) {
  constexpr int CIN = KS * 16, COUT = NTC * 32, RB = 2 * COUT * 2, CPR = RB / 16;     // row bytes / chunks of the LDS tile
  __shared__ __attribute__((aligned(16))) char tile_all[4][32 * RB];
  __shared__ int64_t obase_all[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, khalf = lane >> 5, kh8 = khalf * 8;
  const int ti = wave >> 1, tj = wave & 1;
  const int64_t V = (int64_t)N * D * H * W, ntiles = (V + 31) / 32;
  bf16x8 bfr[2][NTC][KS];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
      for (int sx = 0; sx < KS; ++sx)
        bfr[k][nt][sx] = *reinterpret_cast<const bf16x8*>(wf + ((int64_t)(ti * 4 + tj * 2 + k) * COUT + nt * 32 + r) * CIN +
                                                          sx * 16 + kh8);
  float bv[NTC];
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) bv[nt] = bias ? bias[nt * 32 + r] : 0.f;
  char* tile = tile_all[wave];
  int64_t* obase = obase_all[wave];
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const FplxTileRange tr = fplx_xcd_tiles(ntiles, xcd);
  bf16x8 an[KS];
  auto fetch = [&](int64_t tt) {
    const int64_t v = tt * 32 + r;
    const bf16_t* ap = x + (v < V ? v : 0) * ldx + kh8;
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) an[sx] = v < V ? *reinterpret_cast<const bf16x8*>(ap + sx * 16) : zero;
  };
  int64_t tt = tr.first;
  if (tt < tr.end) fetch(tt);
  for (; tt < tr.end; tt += tr.step) {
    bf16x8 a[KS];
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) a[sx] = an[sx];
    if (tt + tr.step < tr.end) fetch(tt + tr.step);
    // this lane's voxel -> its first output voxel (2d + i, 2h + j, 2w) of the wave's row pair; -1 past the end
    if (lane < 32) {
      const int64_t v = tt * 32 + r;
      int64_t ob = -1;
      if (v < V) {
        unsigned q = (unsigned)v;
        const int w0 = (int)(q % (unsigned)W); q /= (unsigned)W;
        const int h0 = (int)(q % (unsigned)H); q /= (unsigned)H;
        const int d0 = (int)(q % (unsigned)D); q /= (unsigned)D;
        ob = (((int64_t)q * 2 * D + 2 * d0 + ti) * 2 * H + 2 * h0 + tj) * 2 * W + 2 * w0;
      }
      obase[r] = ob;
    }
    f32x16 acc[2][NTC];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][nt][i] = 0.f;
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sx], bfr[k][nt][sx], acc[k][nt], 0, 0, 0);
      }
    // accumulator: lane = channel r of N-tile nt, registers = voxels  ->  tile[voxel][k][channel]
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          *reinterpret_cast<bf16_t*>(tile + ((i & 3) + 8 * (i >> 2) + 4 * khalf) * RB + k * (COUT * 2) + nt * 64 + r * 2) =
              (bf16_t)(acc[k][nt][i] + bv[nt]);
    // 16-byte stores: consecutive lanes run along a voxel's 2 * Cout channels, then along w
#pragma unroll
    for (int q = 0; q < 32 * CPR / 64; ++q) {
      const int e = lane + 64 * q, vox = e / CPR, c = e % CPR;
      const int k = c / (CPR / 2), cc = c % (CPR / 2);
      const uint4 pk = *reinterpret_cast<const uint4*>(tile + vox * RB + c * 16);
      const int64_t ob = obase[vox];
      if (ob >= 0) *reinterpret_cast<uint4*>(y + (ob + k) * ldy + cc * 8) = pk;
    }
  }
}

// ConvTranspose3d(k=2,s=2) weight gradient: dW[tap][ci][co] = sum_v x[v][ci] * dy[out(v,tap)][co].
// Block = one chunk of 128 consecutive input voxels at a time (grid-strided), CIT ci-tiles x one
// co-tile; x chunk and the 8 parity-gathered dy chunks are staged in LDS and read transposed
// (ds_read_b64_tr_b16); wave w owns taps 2w, 2w+1.  Partials per block, fixed-order reduce.
template <int CIT>
__global__ void __launch_bounds__(256)
deconv_wgrad_mfma(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ dy, int64_t ldy,
                  float* __restrict__ part, float* __restrict__ bpart, int N, int D, int H, int W, int Cin, int Cout,
                  int sd) {        // sd = 1: ConvTranspose2d per depth slice - taps 4..7 do not exist (zero fragments)
  constexpr int KV = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                           // [CIT][KV][32] bf16
  char* dys = smem + CIT * KV * 64;          // [8][KV][32] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncig = Cin / (32 * CIT);
  const int cot = blockIdx.y / ncig, cig = blockIdx.y % ncig;
  const int64_t V = (int64_t)N * D * H * W;
  const int64_t chunks = (V + KV - 1) / KV;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;
  // staging geometry: a thread always handles 16-byte chunk c16 of voxels vx0 and vx0 + 64 of the chunk
  const int c16 = tid & 3, vx0 = tid >> 2;
  const bool want_bias = bpart != nullptr && cig == 0;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  f32x16 acc[2][CIT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < CIT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
  for (int64_t ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const int64_t v0 = ch * KV;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int vox = vx0 + 64 * half;
      int64_t vv = v0 + vox;
      const bool ok = vv < V;
      uint4 xv[CIT], gv[8];
#pragma unroll
      for (int c = 0; c < CIT; ++c) xv[c] = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int tap = 0; tap < 8; ++tap) gv[tap] = make_uint4(0, 0, 0, 0);
      if (ok) {
#pragma unroll
        for (int c = 0; c < CIT; ++c)
          xv[c] = *reinterpret_cast<const uint4*>(x + vv * ldx + (cig * CIT + c) * 32 + c16 * 8);
        const int w0 = (int)(vv % W); vv /= W;
        const int h0 = (int)(vv % H); vv /= H;
        const int d0 = (int)(vv % D); vv /= D;
        const int64_t obase = ((vv * sd * D + sd * d0) * 2 * H + 2 * h0) * 2 * W + 2 * w0;
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
          if (tap < 4 * sd) {
            const int64_t ov = obase + ((int64_t)(tap >> 2) * 2 * H + ((tap >> 1) & 1)) * 2 * W + (tap & 1);
            gv[tap] = *reinterpret_cast<const uint4*>(dy + ov * ldy + cot * 32 + c16 * 8);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CIT; ++c) *reinterpret_cast<uint4*>(xs + (c * KV + vox) * 64 + c16 * 16) = xv[c];
#pragma unroll
      for (int tap = 0; tap < 8; ++tap) {
        *reinterpret_cast<uint4*>(dys + (tap * KV + vox) * 64 + c16 * 16) = gv[tap];
        if (want_bias) {                                    // bias gradient = column sums of dy, for free
          const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(&gv[tap]);
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += (float)b8[j];
        }
      }
    }
    __syncthreads();
    for (int ks = 0; ks < KV / 16; ++ks) {
      bf16x8 bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bfr[t] = tr_frag(dys + ((2 * wave + t) * KV + ks * 16) * 64 + lane_off);
#pragma unroll
      for (int c = 0; c < CIT; ++c) {
        const bf16x8 afr = tr_frag(xs + (c * KV + ks * 16) * 64 + lane_off);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[t], acc[t][c], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // part[blockIdx.x][blockIdx.y][tap][c][ci 32][co 32]
  float* out = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (8 * CIT * 1024);
  const int co = lane & 31, rbase = (lane >> 5) * 4;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < CIT; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2) + rbase;
        out[(((2 * wave + t) * CIT + c) * 32 + ci) * 32 + co] = acc[t][c][i];
      }
  if (want_bias) {                                          // uniform per block
    float* red = reinterpret_cast<float*>(smem);            // [256][8]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 32) {
      const int cc = tid >> 3, j = tid & 7;                 // channel = cc * 8 + j, held by threads with c16 == cc
      float t = 0.f;
      for (int k = 0; k < 64; ++k) t += red[(k * 4 + cc) * 8 + j];
      bpart[(int64_t)blockIdx.x * Cout + cot * 32 + tid] = t;
    }
  }
}

__global__ void __launch_bounds__(64) deconv_bias_reduce(const float* __restrict__ bpart, int rows, int C, float* __restrict__ db) {
  const int c = blockIdx.x;
  float t = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) t += bpart[(int64_t)r * C + c];
  t = wave_sum(t);
  if (threadIdx.x == 0) db[c] = t;
}

// dw[ci][co][tap] (torch ConvTranspose3d layout) = sum_b part[b][pair][tap][c][ci%32][co%32]
__global__ void __launch_bounds__(256)
deconv_wgrad_reduce(const float* __restrict__ part, int nblk, int npairs, int cit, int Cin, int Cout,
                    float* __restrict__ dw, int ntaps) {      // ntaps = 8, or 4 (the partials of taps 4..7 are dropped)
  __shared__ float red[256];
  const int64_t per = (int64_t)8 * cit * 1024, total = npairs * per;
  const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6, o = threadIdx.x & 63;
  float t = 0.f;
  if (i < total)
    for (int b = pl; b < nblk; b += 4) t += part[(int64_t)b * total + i];
  red[pl * 64 + o] = t;
  __syncthreads();
  if (pl != 0 || i >= total) return;
  t = red[o] + red[64 + o] + red[128 + o] + red[192 + o];
  const int co_l = i & 31, ci_l = (i >> 5) & 31, c = (int)((i >> 10) % cit), tap = (int)((i / (1024 * cit)) % 8);
  const int pair = (int)(i / per);
  const int ncig = Cin / (32 * cit);
  const int co = (pair / ncig) * 32 + co_l, ci = ((pair % ncig) * cit + c) * 32 + ci_l;
  if (tap < ntaps) dw[((int64_t)ci * Cout + co) * ntaps + tap] = t;
}

struct DwCfg { int cit, npairs, nblk; size_t ws; };
inline DwCfg dw_cfg(int n, int d, int h, int w, int cin, int cout) {
  DwCfg c;
  c.cit = (cin % 128 == 0) ? 4 : (cin % 64 == 0 ? 2 : 1);
  c.npairs = (cin / (32 * c.cit)) * (cout / 32);
  const int64_t chunks = ((int64_t)n * d * h * w + 127) / 128;
  int nb = 512 / c.npairs;
  if (nb < 1) nb = 1;
  if (nb > chunks) nb = (int)chunks;
  c.nblk = nb;
  c.ws = (size_t)c.nblk * c.npairs * 8 * c.cit * 1024 * sizeof(float) + (size_t)c.nblk * cout * sizeof(float);
  return c;
}

// ------------------------------------------------------------------------------------------
// conv_fwd_stream: the previous-generation slab kernel (Cin in {32, 64}, large H x W); the depth-marching kernels of
// conv_march.hip take these shapes first, this one remains for what they refuse and for A/B runs (FPLX_MARCH=0).
// One block = an 8 x 32
// output footprint in (h, w) x 32 output channels, marching along d:
//   * three input slabs (10 x 34 voxels x Cin, 1-voxel halo, zero fill = padding) form a ring in
//     LDS; every slab is fetched from HBM/L2 once per block and feeds 27 taps x 3 output depths;
//     the next slab travels global -> registers while the current depth is computed and is
//     written to LDS behind the depth's last barrier (async-stage split);
//   * the weights of GT taps at a time stream L2 -> registers -> a two-buffer LDS ring, one
//     barrier per group;
//   * LDS rows are 16-byte-chunk XOR-swizzled so that every ds_read_b128 of an operand fragment
//     (32 voxels / 32 output channels x 16 B) is bank-conflict free;
//   * epilogue per depth: bias, bf16 store, per-channel sum / sum of squares kept in registers and
//     written once per block (DSBN statistics, fixed order).
template <int CIN>
struct StreamGeo {
  static constexpr int ROWB = CIN * 2, CH = ROWB / 16;
  static constexpr int SW = 34, SH = 10, SLAB = SW * SH;
  // Cin = 32: all 27 taps of the block's 32 output channels stay resident in LDS (55 KB) and the slab ring
  // has a 4th slot, so one barrier per depth suffices.  Cin = 64: 3 taps at a time through a 2-buffer ring.
  static constexpr int GT = CIN == 32 ? 27 : 3, NG = 27 / GT;
  static constexpr int NSLOT = CIN == 32 ? 4 : 3, NWBUF = NG == 1 ? 1 : 2;
  static constexpr int SLAB_BYTES = SLAB * ROWB, WBUF_BYTES = GT * 32 * ROWB;
  static constexpr int NLD = (SLAB * CH + 255) / 256, NLW = NG == 1 ? 1 : (GT * 32 * CH + 255) / 256;
  static constexpr int LDS = NSLOT * SLAB_BYTES + NWBUF * WBUF_BYTES;
  static __device__ __forceinline__ int swz(int row) { return (row / (16 / CH)) % CH; }
};

template <int CIN>
__global__ void __launch_bounds__(256)
conv_fwd_stream(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen) {
  typedef StreamGeo<CIN> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabs = smem;
  char* wbuf = smem + G::NSLOT * G::SLAB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, khalf = lane >> 5;
  int b = blockIdx.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = b;
  const int h0 = th * 8, w0 = tw * 32, d0 = seg * dlen;
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = blockIdx.y * 32;

  uint4 sreg[G::NLD], wreg[G::NLW];
  auto slab_fetch = [&](int d) {                 // global -> registers (zero outside the volume)
    const bool dok = d >= 0 && d < D;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
      const int i = tid + k * 256;
      const int vox = i / G::CH, c = i % G::CH;
      const int hh = vox / G::SW + h0 - 1, ww = vox % G::SW + w0 - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < G::SLAB * G::CH && dok && hh >= 0 && hh < H && ww >= 0 && ww < W)
        v = *reinterpret_cast<const uint4*>(x + ((((int64_t)n * D + d) * H + hh) * W + ww) * ldx + c * 8);
      sreg[k] = v;
    }
  };
  auto slab_commit = [&](int d) {                // registers -> LDS slot of depth d
    char* dst = slabs + ((d + 1) % G::NSLOT) * G::SLAB_BYTES;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
      const int i = tid + k * 256;
      const int vox = i / G::CH, c = i % G::CH;
      if (i < G::SLAB * G::CH) *reinterpret_cast<uint4*>(dst + vox * G::ROWB + ((c ^ G::swz(vox)) * 16)) = sreg[k];
    }
  };
  auto w_fetch = [&](int g) {
#pragma unroll
    for (int k = 0; k < G::NLW; ++k) {
      const int i = tid + k * 256;
      const int row = i / G::CH, c = i % G::CH;                 // row = tap_local * 32 + co
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < G::GT * 32 * G::CH)
        v = *reinterpret_cast<const uint4*>(wp + ((int64_t)(g * G::GT + (row >> 5)) * Cout + n0 + (row & 31)) * CIN + c * 8);
      wreg[k] = v;
    }
  };
  auto w_commit = [&](int g) {
    char* dst = wbuf + (g & 1) * G::WBUF_BYTES;
#pragma unroll
    for (int k = 0; k < G::NLW; ++k) {
      const int i = tid + k * 256;
      const int row = i / G::CH, c = i % G::CH;
      if (i < G::GT * 32 * G::CH) *reinterpret_cast<uint4*>(dst + row * G::ROWB + ((c ^ G::swz(row)) * 16)) = wreg[k];
    }
  };

  const int co = n0 + r;
  const float bv = bias ? bias[co] : 0.f;
  float ssum = 0.f, qsum = 0.f;
  int wpar = 0;                                  // weight-ring buffer holding the current group

  // prologue: slabs d0-1, d0 and weight group 0
  slab_fetch(d0 - 1); slab_commit(d0 - 1);
  slab_fetch(d0); slab_commit(d0);
  slab_fetch(d0 + 1);
  if constexpr (G::NG == 1) {                    // resident weights: straight global -> LDS, once per block
    for (int i = tid; i < 27 * 32 * G::CH; i += 256) {
      const int row = i / G::CH, c = i % G::CH;
      const uint4 v = *reinterpret_cast<const uint4*>(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * CIN + c * 8);
      *reinterpret_cast<uint4*>(wbuf + row * G::ROWB + ((c ^ G::swz(row)) * 16)) = v;
    }
  } else {
    w_fetch(0); w_commit(0);
  }
  __syncthreads();
  slab_commit(d0 + 1);
  __syncthreads();

  for (int d = d0; d < d1; ++d) {
    if (d + 1 < d1) slab_fetch(d + 2);           // in flight during this depth's 27 taps
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
#pragma unroll 1
    for (int g = 0; g < G::NG; ++g) {
      const int gn = (g + 1 == G::NG) ? 0 : g + 1;
      if constexpr (G::NG > 1) w_fetch(gn);
      const char* wb_ = wbuf + wpar * G::WBUF_BYTES;
      // software pipeline: the fragments of tap tl+1 are requested before the MFMAs of tap tl are
      // issued, so every ds_read_b128 has a full tap of matrix work (2*KS MFMAs) to land behind
      constexpr int KS = CIN / 16;
      bf16x8 fb[2][KS], fa[2][KS][2];
      auto load_tap = [&](int tl, int buf) {
        const int tap = g * G::GT + tl;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const char* sl = slabs + ((d + kd) % G::NSLOT) * G::SLAB_BYTES;   // depth d + kd - 1
        const int brow = tl * 32 + r;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int c = 2 * s + khalf;
          fb[buf][s] = *reinterpret_cast<const bf16x8*>(wb_ + brow * G::ROWB + ((c ^ G::swz(brow)) * 16));
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int vox = (wave * 2 + m + kh) * G::SW + r + kw;
            fa[buf][s][m] = *reinterpret_cast<const bf16x8*>(sl + vox * G::ROWB + ((c ^ G::swz(vox)) * 16));
          }
        }
      };
      load_tap(0, 0);
#pragma unroll
      for (int tl = 0; tl < G::GT; ++tl) {
        if (tl + 1 < G::GT) load_tap(tl + 1, (tl + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tl & 1][s][m], fb[tl & 1][s], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the ring alternates buffers across the wrap-around into the next depth as well (NG is odd)
      if constexpr (G::NG > 1) {
        wpar ^= 1;
        char* dst = wbuf + wpar * G::WBUF_BYTES;
#pragma unroll
        for (int k = 0; k < G::NLW; ++k) {
          const int i = tid + k * 256;
          const int row = i / G::CH, c = i % G::CH;
          if (i < G::GT * 32 * G::CH) *reinterpret_cast<uint4*>(dst + row * G::ROWB + ((c ^ G::swz(row)) * 16)) = wreg[k];
        }
        __syncthreads();
      }
    }
    // 3 slots: the slot of depth d-1, free since the last group barrier; 4 slots: the slot of depth d-2, free since
    // the barrier that ended depth d-1.  The slab is committed BEFORE the output stores are issued: loads and stores
    // share vmcnt, so waiting for the slab's registers after the stores would wait for the stores as well.
    if (d + 1 < d1) slab_commit(d + 2);
    // epilogue of depth d
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int h = h0 + wave * 2 + m;
      if (h < H) {
        const int64_t vrow = (((int64_t)n * D + d) * H + h) * W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int w = w0 + (i & 3) + 8 * (i >> 2) + 4 * khalf;
          if (w < W) {
            const float o = acc[m][i] + bv;
            y[(vrow + w) * ldy + co] = (bf16_t)o;
            ssum += o;
            qsum = fmaf(o, o, qsum);
          }
        }
      }
    }
    if (d + 1 < d1) {
      // one barrier publishes the new slab; raw (no fence): the output stores stay in flight across it
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  if (stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [4][2][32]
    const float a = ssum + __shfl_xor(ssum, 32, 64), q2 = qsum + __shfl_xor(qsum, 32, 64);
    if (lane < 32) { red[(wave * 2 + 0) * 32 + r] = a; red[(wave * 2 + 1) * 32 + r] = q2; }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      stats[((int64_t)blockIdx.x * 2 + which) * Cout + n0 + c] =
          red[(0 * 2 + which) * 32 + c] + red[(1 * 2 + which) * 32 + c] + red[(2 * 2 + which) * 32 + c] + red[(3 * 2 + which) * 32 + c];
    }
  }
}

struct StreamCfg { int tilesH, tilesW, dsegs, dlen, nblk; };
inline int stream_min_w() {
  return (int)fplx_knob(FPLX_K_STREAM_MIN_W);          // tuning knob (benchmarks only)
}
inline bool stream_ok(int d, int h, int w, int cin, int cout) {
  // below W = 128 (level 1) the LDS-tiled GEMM kernel is faster whenever it applies (Cout % 64 == 0)
  const bool tile_applies = cin % 32 == 0 && cout % 64 == 0;
  return (cin == 32 || cin == 64) && cout % 32 == 0 && h >= 16 && w >= stream_min_w() && d >= 4 &&
         (w >= 128 || !tile_applies);
}
inline StreamCfg stream_cfg(int n, int d, int h, int w, int cout) {
  StreamCfg c;
  c.tilesH = (h + 7) / 8;
  c.tilesW = (w + 31) / 32;
  const int tiles = n * c.tilesH * c.tilesW * (cout / 32);
  int ds = (1024 + tiles / 2) / tiles;
  const int maxds = d / 8 > 0 ? d / 8 : 1;
  if (ds > maxds) ds = maxds;
  if (ds < 1) ds = 1;
  c.dlen = (d + ds - 1) / ds;
  c.dsegs = (d + c.dlen - 1) / c.dlen;
  c.nblk = n * c.tilesH * c.tilesW * c.dsegs;
  return c;
}

// ------------------------------------------------------------------------------------------
// conv_fwd_tile: LDS-tiled implicit GEMM (Cin % KC == 0, Cout % 64 == 0) for every 3x3x3 layer that is not on
// the slab-streaming kernel (levels 1-4).  Block tile = 128 voxels x NT output channels, waves 2 (M) x 2 (N).
// Per K-iteration (one tap, KC = 32 or 64 input channels) the shifted / zero-padded voxel rows (A) and the
// weight rows (B) travel global -> registers while the previous iteration computes out of the other LDS
// buffer, then are committed (16-byte-chunk XOR swizzle: conflict-free ds_read_b128) behind ONE barrier.
// KC = 64 doubles the matrix work per barrier and per global-load round trip.  blockIdx.z deals the taps
// (split-K) exactly like conv_fwd_direct.
// MTL = 256 (8 waves, 4 (M) x 2 (N), one block per CU) keeps the per-wave work and the waves per SIMD of the 128-voxel
// form but shares each weight tile among twice the voxels: 25 % fewer L2 -> LDS bytes per MFMA.  Measured: no gain
// (see direct_cfg), so it is selected only by FPLX_TILE_MT=256.
template <int NT, int KC, int MTL>
__global__ void __launch_bounds__(MTL * 2, MTL == 128 ? 2 : 1)
conv_fwd_tile(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
              bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int Cin, int Cout,
              float* __restrict__ stats, float* __restrict__ partial, int tap_lo, int tap_cnt, int xcd) {
  // tap_lo / tap_cnt: the taps to run - 0 / 27, or 9 / 9 for the 2.5D levels whose 27-tap packs are zero outside the
  // middle depth plane (a third of the work, the same result)
  constexpr int THREADS = MTL * 2, NTW = NT / 64;            // N tiles (32 wide) per wave
  constexpr int ROWB = KC * 2, CH = KC / 8, RP = THREADS / CH;   // row bytes, 16-byte chunks per row, rows per pass
  constexpr int PA = MTL / RP, PB = NT / RP;                 // staging passes (chunks per thread) for A and B
  constexpr int KS = KC / 16;
  constexpr int A_BYTES = MTL * ROWB, B_BYTES = NT * ROWB, BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, khalf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t V = (int64_t)N * D * H * W;
  const FplxBlock bid = fplx_xcd_block(xcd);
  const int64_t m0 = (int64_t)bid.x * MTL;
  const int n0 = bid.y * NT;
  // staging geometry: thread -> 16-byte chunk c16 of rows row0 + RP * u
  const int c16 = tid % CH, row0 = tid / CH;
  // per staged row: pointer to its centre-tap voxel and a 27-bit mask of the taps that stay inside the volume,
  // so a K-iteration costs one 64-bit add and one bit test per row instead of the full index arithmetic
  const bf16_t* abase[PA];
  uint32_t amask[PA];
#pragma unroll
  for (int u = 0; u < PA; ++u) {
    int64_t v = m0 + row0 + RP * u;
    const bool ok = v < V;
    if (!ok) v = 0;
    abase[u] = x + v * ldx + c16 * 8;
    unsigned q = (unsigned)v;                    // V < 2^31 (mfma_applicable): 32-bit divisions, not 64-bit sequences
    const int ww0 = (int)(q % (unsigned)W); q /= (unsigned)W;
    const int hh0 = (int)(q % (unsigned)H); q /= (unsigned)H;
    const int dd0 = (int)(q % (unsigned)D);
    uint32_t m = 0;
    for (int t = 0; t < 27; ++t) {
      const int dd = dd0 + t / 9 - 1, hh = hh0 + (t / 3) % 3 - 1, ww = ww0 + t % 3 - 1;
      if (ok && dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W) m |= 1u << t;
    }
    amask[u] = m;
  }
  const int nkc = Cin / KC;
  const int ntaps = (tap_cnt - bid.z + (int)gridDim.z - 1) / (int)gridDim.z;
  const int niter = ntaps * nkc;
  auto swz = [](int row) { return (row / (16 / CH)) % CH; };
  // named staging registers (runtime-indexed arrays would go to scratch).  The A loads are UNCONDITIONAL: a row
  // whose tap falls outside the volume (padding) reads its centre voxel instead and is replaced by zeros when it is
  // committed (zmask, one bit per staged row).  A conditional load with a zero default makes hipcc drain vmcnt(0)
  // between the A and the B loads of one iteration, i.e. wait a full L2 round trip before the B loads even leave.
  // TWO register sets (suffix 0 / 1): the loads of iteration it + 2 leave before iteration it computes, so every load
  // has a full iteration of matrix work plus the partner blocks' time to come back (an L2 round trip is longer than
  // one iteration's MFMAs).
  uint4 areg00, areg01, areg02, areg03, breg00, breg01, breg02, breg03;
  uint4 areg10, areg11, areg12, areg13, breg10, breg11, breg12, breg13;
  areg02 = areg03 = breg01 = breg02 = breg03 = make_uint4(0, 0, 0, 0);
  areg12 = areg13 = breg11 = breg12 = breg13 = make_uint4(0, 0, 0, 0);
  unsigned zmask0 = 0, zmask1 = 0;
#define TILE_A(U, S)                                                                                                \
  if (PA > (U)) {                                                                                                   \
    constexpr int u_ = (U) < PA ? (U) : 0;                                                                          \
    const bool in_ = (amask[u_] >> tap_) & 1u;                                                                      \
    zmask##S |= in_ ? 0u : (1u << (U));                                                                             \
    areg##S##U = *reinterpret_cast<const uint4*>(abase[u_] + (in_ ? aoff_ : (int64_t)kc_));                         \
  }
#define TILE_B(U, S)                                                                                                \
  if (PB > (U)) breg##S##U = *reinterpret_cast<const uint4*>(wp + ((int64_t)tap_ * Cout + n0 + row0 + RP * (U)) * Cin + kc_ + c16 * 8);
#define TILE_FETCH(IT, S)                                                                                           \
  do {                                                                                                              \
    const int tap_ = tap_lo + bid.z + ((IT) / nkc) * gridDim.z, kc_ = ((IT) % nkc) * KC;                       \
    const int kd_ = tap_ / 9 - 1, kh_ = (tap_ / 3) % 3 - 1, kw_ = tap_ % 3 - 1;                                     \
    const int64_t aoff_ = (((int64_t)kd_ * H + kh_) * W + kw_) * ldx + kc_;     /* wave-uniform */                  \
    zmask##S = 0;                                                                                                   \
    TILE_A(0, S) TILE_A(1, S) TILE_A(2, S) TILE_A(3, S)                                                             \
    TILE_B(0, S) TILE_B(1, S) TILE_B(2, S) TILE_B(3, S)                                                             \
  } while (0)
#define TILE_ST(BASE, ROW, REG) *reinterpret_cast<uint4*>((BASE) + (ROW) * ROWB + ((c16 ^ swz(ROW)) * 16)) = REG
#define TILE_STZ(BASE, ROW, REG, U, S) TILE_ST(BASE, ROW, ((zmask##S >> (U)) & 1u) ? make_uint4(0, 0, 0, 0) : REG)
#define TILE_COMMIT(BUFI, S)                                                                                        \
  do {                                                                                                              \
    char* a_ = smem + (BUFI) * BUF;                                                                                 \
    char* b_ = a_ + A_BYTES;                                                                                        \
    TILE_STZ(a_, row0, areg##S##0, 0, S);                                                                           \
    if (PA > 1) TILE_STZ(a_, row0 + RP, areg##S##1, 1, S);                                                          \
    if (PA > 2) TILE_STZ(a_, row0 + 2 * RP, areg##S##2, 2, S);                                                      \
    if (PA > 3) TILE_STZ(a_, row0 + 3 * RP, areg##S##3, 3, S);                                                      \
    TILE_ST(b_, row0, breg##S##0);                                                                                  \
    if (PB > 1) TILE_ST(b_, row0 + RP, breg##S##1);                                                                 \
    if (PB > 2) TILE_ST(b_, row0 + 2 * RP, breg##S##2);                                                             \
    if (PB > 3) TILE_ST(b_, row0 + 3 * RP, breg##S##3);                                                             \
  } while (0)
  f32x16 acc[2][NTW];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][j][i] = 0.f;

  // one K-iteration out of LDS buffer BUFI
#define TILE_COMPUTE(BUFI)                                                                                          \
  do {                                                                                                              \
    const char* a = smem + (BUFI) * BUF;                                                                            \
    const char* b = a + A_BYTES;                                                                                    \
    bf16x8 fa[KS][2], fb[KS][NTW];                                                                                  \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                \
      const int c = 2 * s + khalf;                                                                                  \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                               \
        const int row = wm * 64 + t * 32 + r;                                                                       \
        fa[s][t] = *reinterpret_cast<const bf16x8*>(a + row * ROWB + ((c ^ swz(row)) * 16));                        \
      }                                                                                                             \
      _Pragma("unroll") for (int j = 0; j < NTW; ++j) {                                                             \
        const int row = wn * (NT / 2) + j * 32 + r;                                                                 \
        fb[s][j] = *reinterpret_cast<const bf16x8*>(b + row * ROWB + ((c ^ swz(row)) * 16));                        \
      }                                                                                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int s = 0; s < KS; ++s)                                                                  \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                 \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                             \
          acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][t], fb[s][j], acc[t][j], 0, 0, 0);              \
  } while (0)

  if (niter > 0) {
    TILE_FETCH(0, 0);
    TILE_COMMIT(0, 0);
  }
  if (niter > 1) TILE_FETCH(1, 1);                   // in flight across the barrier
  __syncthreads();
  for (int it = 0; it < niter; it += 2) {
    // even iteration: LDS buffer 0 holds it, register set 1 holds it + 1
    if (it + 2 < niter) TILE_FETCH(it + 2, 0);
    TILE_COMPUTE(0);
    if (it + 1 < niter) TILE_COMMIT(1, 1);
    __syncthreads();
    if (it + 1 >= niter) break;
    // odd iteration: buffer 1 holds it + 1, register set 0 holds it + 2
    if (it + 3 < niter) TILE_FETCH(it + 3, 1);
    TILE_COMPUTE(1);
    if (it + 2 < niter) TILE_COMMIT(0, 0);
    __syncthreads();
  }
#undef TILE_A
#undef TILE_B
#undef TILE_FETCH
#undef TILE_ST
#undef TILE_STZ
#undef TILE_COMMIT
#undef TILE_COMPUTE

  const int rh = khalf * 4;
  if (partial) {
    // fp32 partial tiles [voxel][Cout]: transposed through a 4-KB per-wave LDS tile, 16-byte stores (the operand tiles
    // are dead after the loop's last barrier)
    float* pz = partial + (int64_t)bid.z * V * Cout;
    float* stf = reinterpret_cast<float*>(smem + 4096 + wave * 4096);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int cb = n0 + wn * (NT / 2) + j * 32;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) stf[((i & 3) + 8 * (i >> 2) + rh) * 32 + r] = acc[t][j][i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // lane -> voxel (lane >> 3) + 8 q, channels 4 (lane & 7) ..
          const int row = (lane >> 3) + 8 * q;
          const int64_t v = m0 + wm * 64 + t * 32 + row;
          const float4 pk = *reinterpret_cast<const float4*>(stf + row * 32 + (lane & 7) * 4);
          if (v < V) *reinterpret_cast<float4*>(pz + v * Cout + cb + (lane & 7) * 4) = pk;
        }
      }
    }
    return;
  }
  float* red = reinterpret_cast<float*>(smem);               // [MTL / 64 (wm)][2][NT]: at most 4 KB
  // bf16 outputs: lane = channel, registers = voxels -> 2-byte pieces; aligned outputs are transposed through a 2-KB
  // per-wave LDS tile (the operand tiles are dead after the loop's last barrier) and leave as 16-byte stores
  const bool vec_ok = ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(y) % 16) == 0;        // uniform
  char* stg = smem + 4096 + wave * 2048;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int cl = wn * (NT / 2) + j * 32 + r, co = n0 + cl;
    const float bv = bias ? bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + rh;
        const int64_t v = m0 + wm * 64 + t * 32 + row;
        const float o = acc[t][j][i] + bv;
        if (vec_ok) *reinterpret_cast<bf16_t*>(stg + row * 64 + r * 2) = (bf16_t)o;
        if (v < V) {
          if (!vec_ok) y[v * ldy + co] = (bf16_t)o;
          s1 += o;
          s2 = fmaf(o, o, s2);
        }
      }
      if (vec_ok) {                                          // 32 x 32 tile: LDS transpose, two 16-byte stores per lane
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int row = (lane >> 2) + 16 * half;
          const int64_t v = m0 + wm * 64 + t * 32 + row;
          const uint4 pk = *reinterpret_cast<const uint4*>(stg + row * 64 + (lane & 3) * 16);
          if (v < V) *reinterpret_cast<uint4*>(y + v * ldy + n0 + wn * (NT / 2) + j * 32 + (lane & 3) * 8) = pk;
        }
      }
    }
    if (stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32) { red[(wm * 2 + 0) * NT + cl] = s1; red[(wm * 2 + 1) * NT + cl] = s2; }
    }
  }
  if (stats) {
    __syncthreads();
    for (int i = tid; i < 2 * NT; i += THREADS) {
      const int which = i / NT, c = i % NT;
      float t = 0.f;
#pragma unroll
      for (int m = 0; m < MTL / 64; ++m) t += red[(m * 2 + which) * NT + c];
      stats[((int64_t)bid.x * 2 + which) * Cout + n0 + c] = t;
    }
  }
}

// split-K finish: y = bf16(sum_z partial[z] + bias), statistics rows; thread = voxel-lane x 8 channels
__global__ void __launch_bounds__(256)
splitk_finish_k(const float* __restrict__ partial, int ks, int64_t V, int Cout, const float* __restrict__ bias,
                bf16_t* __restrict__ y, int64_t ldy, float* __restrict__ stats, const float* __restrict__ slope_p = nullptr) {
  const float slope_v = slope_p ? *slope_p : 1.f;      // inference: PReLU on the finished sum (1 = identity: x > 0 ? x : x * 1)
  const int G = Cout / 8;                      // channel groups per voxel (<= 256 by construction)
  const int VL = 256 / G;
  const int g = threadIdx.x % G, vl = threadIdx.x / G;
  const int c0 = g * 8;
  float s[8], q[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = q[j] = 0.f; bv[j] = bias ? bias[c0 + j] : 0.f; }
  if (vl < VL)
    for (int64_t v = (int64_t)blockIdx.x * VL + vl; v < V; v += (int64_t)gridDim.x * VL) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = bv[j];
      for (int z = 0; z < ks; ++z) {
        const float4 a = *reinterpret_cast<const float4*>(partial + ((int64_t)z * V + v) * Cout + c0);
        const float4 b = *reinterpret_cast<const float4*>(partial + ((int64_t)z * V + v) * Cout + c0 + 4);
        o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; o[4] += b.x; o[5] += b.y; o[6] += b.z; o[7] += b.w;
      }
      bf16x8 ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = o[j] > 0.f ? o[j] : o[j] * slope_v;
        ov[j] = (bf16_t)o[j]; s[j] += o[j]; q[j] = fmaf(o[j], o[j], q[j]);
      }
      *reinterpret_cast<bf16x8*>(y + v * ldy + c0) = ov;
    }
  if (!stats) return;
  __shared__ float red[256][16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = vl < VL ? s[j] : 0.f; red[threadIdx.x][8 + j] = vl < VL ? q[j] : 0.f; }
  __syncthreads();
  if (threadIdx.x < G) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t0 = 0.f, t1 = 0.f;
      for (int k = 0; k < VL; ++k) { t0 += red[k * G + threadIdx.x][j]; t1 += red[k * G + threadIdx.x][8 + j]; }
      stats[((int64_t)blockIdx.x * 2 + 0) * Cout + c0 + j] = t0;
      stats[((int64_t)blockIdx.x * 2 + 1) * Cout + c0 + j] = t1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// deconv_dgrad_rows: data gradient of ConvTranspose3d(k = 2, s = 2) for the shallow levels (Cout = 32 | 64 channels of dy,
// Cin = 64 | 128): dx[parent][ci] = sum over the parent's 8 children and co of dy[child][co] w[ci][co][tap] - no reuse, a pure
// stream of dy (262 MB at level 0).  conv_fwd_direct<.., 1> gathers its fragments straight from global memory, 16 bytes
// per lane at a 128-byte stride: every wave instruction touches 32 cache lines for 1 KiB (96 us = 3.4 TB/s alone, 195 us
// beside the weight-gradient stream).  Here a block takes MB parents of one row at a time; their children are four
// CONTIGUOUS child-row pieces (128 Cout/32 bytes per parent and (i, j)), fetched with fully coalesced 16-byte loads into
// registers one segment ahead and committed to LDS with the 16-byte chunk index XOR-ed by a swizzle of the parent index that
// follows ds_read_b128's lane groups (pswz below: conflict-free fragment reads since round 6).  The product is TRANSPOSED,
// D[ci][parent] = Wb[ci][k] dy^T[k][parent]: the weights are the A operand and stay in registers for the whole kernel, a lane
// ends up with 16 input channels of ONE parent, two v_permlane32_swap exchanges make them two 16-byte stores.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2d;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4d;
template <int CO32, int NTW>       // Cout / 32 (1 | 2), Cin / 32 (2 | 4): four wave tiles = (4 / NTW) parent tiles x NTW channel tiles
__global__ void __launch_bounds__(256)
deconv_dgrad_rows(const bf16_t* __restrict__ dy, int64_t ldy, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx,
                  int64_t ldx, int N, int D, int H, int W, int segsW, int64_t nseg, int xcd) {
  constexpr int COUT = CO32 * 32, CIN = NTW * 32, MB = 32 * 4 / NTW, U = 8 * CO32;        // U: 16-byte chunks per (parent, i, j)
  constexpr int NIT = 4 * MB * U / 256, KS = 2 * CO32;                                    // k-steps of 16 per tap
  __shared__ __attribute__((aligned(16))) char sm[4 * MB * U * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kq = lane >> 5;
  const int mt = wave / NTW, n0 = (wave % NTW) * 32;
  bf16x8 afr[8][KS];                                  // A: row = input channel n0 + r, the lane's 8 output channels of (tap, s)
#pragma unroll
  for (int tap = 0; tap < 8; ++tap)
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
      afr[tap][s_] = *reinterpret_cast<const bf16x8*>(wb + ((int64_t)tap * CIN + n0 + r) * COUT + s_ * 16 + 8 * kq);
  const FplxTileRange tr = fplx_xcd_tiles(nseg, xcd);
  struct Seg { int n, d, h, w0; };
  auto seg_of = [&](int64_t t64) {
    unsigned t = (unsigned)t64;
    Seg o;
    o.w0 = (int)(t % (unsigned)segsW) * MB; t /= (unsigned)segsW;
    o.h = (int)(t % (unsigned)H); t /= (unsigned)H;
    o.d = (int)(t % (unsigned)D);
    o.n = (int)(t / (unsigned)D);
    return o;
  };
  // chunk q of parent p sits at slot q ^ pswz(p) of the parent's U slots.  A fragment read is one chunk index for 32 parents
  // (lane = parent r), and ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): the
  // swizzle must separate the 16 parents of a GROUP, not 16 consecutive ones.  Round 3's p & 7 left parents {0, 24}, {12, 20}, ...
  // of a group on one slot - 2-way conflicts on every read, a third of the kernel's LDS cycles (profiles/r05_pmc_sq_counters.txt:
  // 33 %).  U = 8 (128-byte rows: slot mod 16 = 8 (p & 1) + (q ^ swz)): bits 1, 3, 4 of p take all eight values over the even and
  // over the odd parents of either group; U = 16 (256-byte rows): p & 15 is distinct over both groups.  The staging stores (8
  // contiguous lanes = 8 consecutive q of one parent) stay one contiguous 128-byte run under both.
  auto pswz = [](int p) { return U == 8 ? (((p >> 1) & 1) | (((p >> 3) & 3) << 1)) : (p & 15); };
  uint4 reg[NIT];
  auto fetch = [&](const Seg& g) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int e = tid + 256 * k;
      const int q = e % U, p = (e / U) % MB, ij = e / (U * MB);
      const int kk = q / (4 * CO32), cc = q % (4 * CO32);
      const bool ok = g.w0 + p < W;
      const int64_t child = (((int64_t)g.n * 2 * D + 2 * g.d + (ij >> 1)) * 2 * H + 2 * g.h + (ij & 1)) * 2 * W + 2 * (g.w0 + (ok ? p : 0)) + kk;
      const uint4 v = *reinterpret_cast<const uint4*>(dy + child * ldy + cc * 8);
      reg[k] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  int64_t tt = tr.first;
  Seg gn = seg_of(tt < tr.end ? tt : 0);
  if (tt < tr.end) fetch(gn);
  for (; tt < tr.end; tt += tr.step) {
    const Seg g = gn;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int e = tid + 256 * k;
      const int q = e % U, p = (e / U) % MB, ij = e / (U * MB);
      *reinterpret_cast<uint4*>(sm + (((ij * MB + p) * U) + (q ^ pswz(p))) * 16) = reg[k];
    }
    __syncthreads();
    if (tt + tr.step < tr.end) {
      gn = seg_of(tt + tr.step);
      fetch(gn);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int ij = tap >> 1, kk = tap & 1;
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        const int q = kk * (4 * CO32) + 2 * s_ + kq;
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(sm + (((ij * MB + mt * 32 + r) * U) + (q ^ pswz(r))) * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tap][s_], b, acc, 0, 0, 0);
      }
    }
    // D rows = channels n0 + (i & 3) + 8 (i >> 2) + 4 kq of parent r: quads -> 8-channel runs by two half-wave swaps
    unsigned pk[8];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const bf16_t e0 = (bf16_t)acc[4 * q4], e1 = (bf16_t)acc[4 * q4 + 1], e2 = (bf16_t)acc[4 * q4 + 2], e3 = (bf16_t)acc[4 * q4 + 3];
      pk[2 * q4] = (unsigned)__builtin_bit_cast(unsigned short, e0) | ((unsigned)__builtin_bit_cast(unsigned short, e1) << 16);
      pk[2 * q4 + 1] = (unsigned)__builtin_bit_cast(unsigned short, e2) | ((unsigned)__builtin_bit_cast(unsigned short, e3) << 16);
    }
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const u32x2d sw = __builtin_amdgcn_permlane32_swap(pk[4 * g2 + e], pk[4 * g2 + 2 + e], false, false);
        pk[4 * g2 + e] = sw[0];
        pk[4 * g2 + 2 + e] = sw[1];
      }
    const int wv = g.w0 + mt * 32 + r;
    if (wv < W) {
      bf16_t* dst = dx + ((((int64_t)g.n * D + g.d) * H + g.h) * W + wv) * ldx + n0 + 8 * kq;
      *reinterpret_cast<u32x4d*>(dst) = u32x4d{pk[0], pk[1], pk[2], pk[3]};
      *reinterpret_cast<u32x4d*>(dst + 16) = u32x4d{pk[4], pk[5], pk[6], pk[7]};
    }
  }
}

// ------------------------------------------------------------------------------------------
// deconv_dgrad_small: the same data gradient for the small deep volumes (<= 16 K parents, dy 128 | 256 | ... channels), where
// conv_fwd_direct<1, 2, 1> is ONE latency chain per block: 64 blocks x 128 dependent (tap, k-step) iterations at level 4
// (48 us for 4 GFLOP).  Here a block = 32 parents x 64 input channels and its four waves SPLIT THE EIGHT TAPS (two each):
// a quarter of the chain per wave, four times the blocks (every CU busy), the four partial tiles summed through LDS in a
// fixed order.  Operand fragments straight from global / L2 (16 B per lane), eight k-steps in flight.
__global__ void __launch_bounds__(256)
deconv_dgrad_small(const bf16_t* __restrict__ dy, int64_t ldy, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx,
                   int64_t ldx, int N, int D, int H, int W, int Cout, int Cin) {
  __shared__ float red[4][2][1024];                   // [wave][col tile][row 32 x col 32]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, kh8 = (lane >> 5) * 8;
  const int64_t V = (int64_t)N * D * H * W;
  const int64_t v = (int64_t)blockIdx.x * 32 + r;
  const int n0 = blockIdx.y * 64;
  const bool ok = v < V;
  unsigned q = (unsigned)(ok ? v : 0);
  const int vw = (int)(q % (unsigned)W); q /= (unsigned)W;
  const int vh = (int)(q % (unsigned)H); q /= (unsigned)H;
  const int vd = (int)(q % (unsigned)D);
  const int vn = (int)(q / (unsigned)D);
  const bf16_t* a0 = dy + ((((int64_t)vn * 2 * D + 2 * vd) * 2 * H + 2 * vh) * 2 * W + 2 * vw) * ldy + kh8;
  const bf16_t* b0 = wb + ((int64_t)(n0 + r)) * Cout + kh8;
  const bf16_t* b1 = b0 + (int64_t)32 * Cout;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const int KS = Cout / 16, NIT = 2 * KS;             // this wave's taps: 2 wave, 2 wave + 1
  constexpr int P = 8;
  bf16x8 ra[P], rb0[P], rb1[P];
  auto fetch = [&](int it, int u) {
    const int tap = 2 * wave + it / KS, kc = (it % KS) * 16;
    const int64_t toff = ((int64_t)((tap >> 2) * 2 * H + ((tap >> 1) & 1)) * 2 * W + (tap & 1)) * ldy + kc;      // uniform
    const int64_t woff = (int64_t)tap * Cin * Cout + kc;
    ra[u] = ok ? *reinterpret_cast<const bf16x8*>(a0 + toff) : zero;
    rb0[u] = *reinterpret_cast<const bf16x8*>(b0 + woff);
    rb1[u] = *reinterpret_cast<const bf16x8*>(b1 + woff);
  };
#pragma unroll
  for (int u = 0; u < P; ++u) fetch(u < NIT ? u : NIT - 1, u);
  for (int it = 0; it < NIT; it += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      if (it + u < NIT) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[u], rb0[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[u], rb1[u], acc1, 0, 0, 0);
      }
      const int nx = it + P + u;
      fetch(nx < NIT ? nx : NIT - 1, u);              // past the end: a harmless re-read (keeps the loop branch-free)
    }
  }
  const int rh = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + rh;
    red[wave][0][row * 32 + r] = acc0[i];
    red[wave][1][row * 32 + r] = acc1[i];
  }
  __syncthreads();
  // thread t -> 8 consecutive channels of one parent: row t / 8, channels 8 (t % 8) of the block's 64
  const int row = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
  const int64_t vo = (int64_t)blockIdx.x * 32 + row;
  if (vo < V) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c8 + j;
      const float* p_ = &red[0][c >> 5][row * 32 + (c & 31)];
      o[j] = (bf16_t)((p_[0] + p_[2048]) + (p_[4096] + p_[6144]));
    }
    *reinterpret_cast<bf16x8*>(dx + vo * ldx + n0 + c8) = o;
  }
}

struct DirectCfg { int mt, ntl, ksplit, fin_blocks, tile_nt, tile_mt; int64_t mblocks; };

inline DirectCfg direct_cfg(int64_t V, int cin, int cout, int taps = 27) {
  DirectCfg c;
  if (cout % 64 == 0) { c.mt = 2; c.ntl = 2; }
  else { c.mt = 4; c.ntl = 1; }
  // LDS-tiled kernel (128 voxels x 128|64 channels per block) when the shape allows
  c.tile_nt = (cin % 32 == 0 && cout % 64 == 0) ? (cout % 128 == 0 ? 128 : 64) : 0;
  c.tile_mt = 128;
  if (c.tile_nt && cin % 64 == 0) {
    // 256-voxel tiles: measured +-2 % at level 2 and -15 % on the 64-wide level-1 layer, so they stay a tuning knob
    // (FPLX_TILE_MT=256) - the kernel is not simply L2-bandwidth-bound
    const int kmt = (int)fplx_knob(FPLX_K_TILE_MT);
    if (kmt == 256) c.tile_mt = 256;
  }
  c.mblocks = c.tile_nt ? (V + c.tile_mt - 1) / c.tile_mt : (V + 4 * c.mt * 32 - 1) / (4 * c.mt * 32);
  // small volumes (deep levels) do not fill 256 CUs: deal the 27 taps to 3 / 9 / 27 blocks - the SMALLEST split that
  // gives every CU a block, because each split adds an fp32 partial tensor to write and re-read (measured on the level-3
  // and level-4 shapes: 9 -> 3 and 27 -> 9 save 12-21 us per launch)
  const int64_t blocks = c.mblocks * (c.tile_nt ? cout / c.tile_nt : cout / (c.ntl * 32));
  c.ksplit = 1;
  if (cout % 8 == 0 && cout <= 2048) {
    if (blocks * 9 < 256) c.ksplit = 27;
    else if (blocks * 3 < 256) c.ksplit = 9;
    else if (blocks < 256) c.ksplit = 3;
    if (c.ksplit > taps) c.ksplit = taps;
  }
  if (c.tile_mt == 256) c.ksplit = 1;                        // chosen only where it fills the chip by itself
  {  // tuning knobs: FPLX_TILE_NT=64 forces the narrow tile, FPLX_TILE_KS forces the tap split (1/3/9/27)
    const int knt = (int)fplx_knob(FPLX_K_TILE_NT), kks = (int)fplx_knob(FPLX_K_TILE_KS);
    if (knt == 64 && c.tile_nt == 128) c.tile_nt = 64;
    if (kks == 1 || kks == 3 || kks == 9 || kks == 27) c.ksplit = kks;
  }
  int64_t fb = (V + 7) / 8;
  c.fin_blocks = (int)(fb > 512 ? 512 : (fb < 1 ? 1 : fb));
  return c;
}

inline bool mfma_applicable(int64_t ldx, int64_t ldy, int cin, int cout, const void* x, const void* y, const void* wp) {
  return cin % 16 == 0 && cout % 32 == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)wp % 16 == 0) &&
         ((uintptr_t)y % 2 == 0) && ldy >= cout;
}

}  // namespace

// conv_march.hip
extern "C" int fplx_march_ok(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_march_rows(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_march_variant(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_march_conv3d_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                     int n, int d, int h, int w, int cin, int cout, float* stats, hipStream_t st,
                                     const void* x1, void* y1, int twod);
extern "C" int fplx_march_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                         int n, int d, int h, int w, int cin, int cout, float* stats, hipStream_t st,
                                         const void* x1, void* y1, int twod, const float* slope, int nmod0);
extern "C" int fplx_brick_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                         int n, int d, int h, int w, int cin, int cout, float* stats, float* partial, int geo,
                                         int ksplit, hipStream_t st, const float* slope, const void* x1, int nmod0);
// conv_brick.hip
extern "C" int fplx_brick_ok(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_brick_first(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_brick_plan(int n, int d, int h, int w, int cin, int cout, int* geo, int* ksplit, int* bricks);
extern "C" int fplx_brick_conv3d_fwd_ex(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                        int n, int d, int h, int w, int cin, int cout, float* stats, float* partial, int geo,
                                        int ksplit, hipStream_t st);
static inline int splitk_fin_blocks(int64_t V) {
  const int64_t fb = (V + 7) / 8;
  return (int)(fb > 512 ? 512 : (fb < 1 ? 1 : fb));
}
static inline int brick_stats_rows(int n, int d, int h, int w, int cin, int cout) {
  int geo, ks, bricks;
  fplx_brick_plan(n, d, h, w, cin, cout, &geo, &ks, &bricks);
  return ks > 1 ? splitk_fin_blocks((int64_t)n * d * h * w) : bricks;
}
static inline size_t brick_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  int geo, ks, bricks;
  fplx_brick_plan(n, d, h, w, cin, cout, &geo, &ks, &bricks);
  return ks > 1 ? (size_t)ks * n * d * h * w * cout * sizeof(float) : 0;
}

// mid != 0: the 27-tap pack is zero outside the middle depth plane (a Conv2d per depth slice, 2.5D levels).  Layers that
// would go to the tile kernel run taps 9..17 only (measured 348 -> 155 us on the 128 -> 64 level-1 layer of the shipped
// config); where a march kernel applies it keeps all 27 taps - its 1.0 PFLOP/s on three times the work still beats the
// tile kernel's short-K form (111 vs 117 us at 64 -> 64, 52 vs 70 us at 32 -> 64).  Same result either way.
static inline bool mid_tile(int mid, int n, int d, int h, int w, int cin, int cout) {
  const bool on = fplx_knob(FPLX_K_MID_TILE) != 0;          // A/B knob
  return mid && on && cin % 32 == 0 && cout % 64 == 0 && !fplx_march_ok(n, d, h, w, cin, cout) &&
         !stream_ok(d, h, w, cin, cout);
}

static int stats_rows_impl(int n, int d, int h, int w, int cin, int cout, int mid) {
  if (cin % 16 != 0 || cout % 32 != 0) return 0;
  const int64_t V = (int64_t)n * d * h * w;
  if (!mid && fplx_brick_first(n, d, h, w, cin, cout)) return brick_stats_rows(n, d, h, w, cin, cout);
  if (fplx_march_ok(n, d, h, w, cin, cout)) return fplx_march_rows(n, d, h, w, cin, cout);
  if (stream_ok(d, h, w, cin, cout)) return stream_cfg(n, d, h, w, cout).nblk;
  if (!mid && fplx_brick_ok(n, d, h, w, cin, cout)) return brick_stats_rows(n, d, h, w, cin, cout);
  const DirectCfg c = direct_cfg(V, cin, cout, mid_tile(mid, n, d, h, w, cin, cout) ? 9 : 27);
  if (c.ksplit > 1) return c.fin_blocks;
  return (int)c.mblocks;
}

static size_t fwd_ws_impl(int n, int d, int h, int w, int cin, int cout, int mid) {
  if (cin % 16 != 0 || cout % 32 != 0) return 0;
  if (!mid && fplx_brick_first(n, d, h, w, cin, cout)) return brick_ws_bytes(n, d, h, w, cin, cout);
  if (fplx_march_ok(n, d, h, w, cin, cout) || stream_ok(d, h, w, cin, cout)) return 0;
  if (!mid && fplx_brick_ok(n, d, h, w, cin, cout)) return brick_ws_bytes(n, d, h, w, cin, cout);
  const int64_t V = (int64_t)n * d * h * w;
  const DirectCfg c = direct_cfg(V, cin, cout, mid_tile(mid, n, d, h, w, cin, cout) ? 9 : 27);
  return c.ksplit > 1 ? (size_t)c.ksplit * V * cout * sizeof(float) : 0;
}

// the dispatch order of mfma_fwd_impl as data (fplx_conv3d_plan_query): kernel family, brick geometry, reduction split
extern "C" int fplx_mfma_conv3d_plan(int n, int d, int h, int w, int cin, int cout, int mid, int* kernel, int* geo, int* ksplit) {
  *kernel = FPLX_KERNEL_GENERIC; *geo = -1; *ksplit = 1;
  if (cin % 16 != 0 || cout % 32 != 0 || (int64_t)n * d * h * w >= ((int64_t)1 << 31)) return 0;
  auto brick = [&]() { int b; fplx_brick_plan(n, d, h, w, cin, cout, geo, ksplit, &b); *kernel = FPLX_KERNEL_BRICK; return 1; };
  if (!mid && fplx_brick_first(n, d, h, w, cin, cout)) return brick();
  if (fplx_march_ok(n, d, h, w, cin, cout)) { *kernel = FPLX_KERNEL_MARCH; *geo = fplx_march_variant(n, d, h, w, cin, cout); return 1; }
  if (stream_ok(d, h, w, cin, cout)) { *kernel = FPLX_KERNEL_STREAM; return 1; }
  if (!mid && fplx_brick_ok(n, d, h, w, cin, cout)) return brick();
  const DirectCfg c = direct_cfg((int64_t)n * d * h * w, cin, cout, mid_tile(mid, n, d, h, w, cin, cout) ? 9 : 27);
  *kernel = c.tile_nt ? FPLX_KERNEL_TILE : FPLX_KERNEL_DIRECT;
  *ksplit = c.ksplit;
  return 1;
}

extern "C" int fplx_mfma_conv3d_stats_rows(int n, int d, int h, int w, int cin, int cout) {
  return stats_rows_impl(n, d, h, w, cin, cout, 0);
}
extern "C" int fplx_mfma_conv3d_mid_stats_rows(int n, int d, int h, int w, int cin, int cout) {
  return stats_rows_impl(n, d, h, w, cin, cout, 1);
}
extern "C" size_t fplx_mfma_conv3d_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  return fwd_ws_impl(n, d, h, w, cin, cout, 0);
}
extern "C" size_t fplx_mfma_conv3d_mid_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  return fwd_ws_impl(n, d, h, w, cin, cout, 1);
}

// slope != NULL (inference: eval-mode BatchNorm folded into the pack): PReLU in the kernel's write-out or in the split-K
// finish; the caller has checked fplx_mfma_conv3d_act_ok (the stream / unsplit tile / direct kernels have no such form)
static int mfma_fwd_impl(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy, int n, int d,
                         int h, int w, int cin, int cout, float* stats, void* ws, size_t ws_bytes, int mid,
                         hipStream_t st, const float* slope = nullptr) {
  if (!mfma_applicable(ldx, ldy, cin, cout, x, y, wp) || (int64_t)n * d * h * w >= ((int64_t)1 << 31)) return 0;
  const bool midt = mid_tile(mid, n, d, h, w, cin, cout);
  const int tap_lo = midt ? 9 : 0, tap_cnt = midt ? 9 : 27;
  auto brick_launch = [&]() -> int {
    int geo, ks, bricks;
    fplx_brick_plan(n, d, h, w, cin, cout, &geo, &ks, &bricks);
    const int64_t Vb = (int64_t)n * d * h * w;
    if (ks > 1 && (!ws || ws_bytes < (size_t)ks * Vb * cout * sizeof(float)))
      return fplx_fail(FPLX_E_WORKSPACE, "mfma_conv3d_fwd: split-K needs %zu workspace bytes (fplx_conv3d_fwd_ws_bytes)",
                       (size_t)ks * Vb * cout * sizeof(float));
    const int rb = fplx_brick_conv3d_fwd_act(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats,
                                             (float*)ws, geo, ks, st, slope, nullptr, 0);
    if (rb == 1 && ks > 1) {
      splitk_finish_k<<<splitk_fin_blocks(Vb), 256, 0, st>>>((const float*)ws, ks, Vb, cout, bias, (bf16_t*)y, ldy, stats, slope);
      const int rf = fplx_check_launch("brick_splitk_finish");
      if (rf < 0) return rf;
    }
    return rb;
  };
  if (!mid && fplx_brick_first(n, d, h, w, cin, cout)) return brick_launch();
  if (fplx_march_ok(n, d, h, w, cin, cout))
    return fplx_march_conv3d_fwd_act(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, st, nullptr, nullptr, mid, slope, 0);
  if (stream_ok(d, h, w, cin, cout)) {
    if (slope) return 0;
    const StreamCfg sc = stream_cfg(n, d, h, w, cout);
    dim3 grid(sc.nblk, cout / 32);
    if (cin == 32) {
      (void)hipFuncSetAttribute((const void*)conv_fwd_stream<32>, hipFuncAttributeMaxDynamicSharedMemorySize, StreamGeo<32>::LDS);
      conv_fwd_stream<32><<<grid, 256, StreamGeo<32>::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y,
                                                                ldy, n, d, h, w, cout, stats, sc.tilesH, sc.tilesW,
                                                                sc.dsegs, sc.dlen);
    } else {
      (void)hipFuncSetAttribute((const void*)conv_fwd_stream<64>, hipFuncAttributeMaxDynamicSharedMemorySize, StreamGeo<64>::LDS);
      conv_fwd_stream<64><<<grid, 256, StreamGeo<64>::LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y,
                                                                ldy, n, d, h, w, cout, stats, sc.tilesH, sc.tilesW,
                                                                sc.dsegs, sc.dlen);
    }
    int rc0 = fplx_check_launch("mfma_conv3d_fwd_stream");
    return rc0 < 0 ? rc0 : 1;
  }
  if (!mid && fplx_brick_ok(n, d, h, w, cin, cout)) return brick_launch();
  const int64_t V = (int64_t)n * d * h * w;
  const DirectCfg c = direct_cfg(V, cin, cout, tap_cnt);
  const int ks = c.ksplit;
  float* partial = nullptr;
  if (slope && ks <= 1) return 0;                          // no activation form of the unsplit tile / direct kernels
  if (ks > 1) {
    // the statistics row count was promised for the split-K path: the workspace is mandatory here
    if (!ws || ws_bytes < (size_t)ks * V * cout * sizeof(float))
      return fplx_fail(FPLX_E_WORKSPACE, "mfma_conv3d_fwd: split-K needs %zu workspace bytes (fplx_conv3d_fwd_ws_bytes)",
                       (size_t)ks * V * cout * sizeof(float));
    partial = (float*)ws;
  }
  if (c.tile_nt) {
    dim3 tg((unsigned)c.mblocks, cout / c.tile_nt, ks);
#define LAUNCH_TILE(NT_, KC_, MT_)                                                                                \
  do {                                                                                                              \
    constexpr int LDS = 2 * (MT_ + NT_) * KC_ * 2;                                                                  \
    (void)hipFuncSetAttribute((const void*)conv_fwd_tile<NT_, KC_, MT_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
    conv_fwd_tile<NT_, KC_, MT_><<<tg, MT_ * 2, LDS, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y, ldy, \
                                                           n, d, h, w, cin, cout, stats, partial, tap_lo, tap_cnt, fplx_xcd_on()); \
  } while (0)
    const bool k64 = cin % 64 == 0;
    if (c.tile_mt == 256) { if (c.tile_nt == 128) LAUNCH_TILE(128, 64, 256); else LAUNCH_TILE(64, 64, 256); }
    else if (c.tile_nt == 128) { if (k64) LAUNCH_TILE(128, 64, 128); else LAUNCH_TILE(128, 32, 128); }
    else { if (k64) LAUNCH_TILE(64, 64, 128); else LAUNCH_TILE(64, 32, 128); }
#undef LAUNCH_TILE
    if (ks > 1) splitk_finish_k<<<c.fin_blocks, 256, 0, st>>>(partial, ks, V, cout, bias, (bf16_t*)y, ldy, stats, slope);
    int rct = fplx_check_launch("mfma_conv3d_fwd_tile");
    return rct < 0 ? rct : 1;
  }
  dim3 grid((unsigned)((V + 4 * c.mt * 32 - 1) / (4 * c.mt * 32)), cout / (c.ntl * 32), ks);
  if (c.ntl == 2)
    conv_fwd_direct<2, 2, 0><<<grid, DIRECT_THREADS, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y,
                                                           ldy, n, d, h, w, cin, cout, stats, partial);
  else
    conv_fwd_direct<4, 1, 0><<<grid, DIRECT_THREADS, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wp, bias, (bf16_t*)y,
                                                           ldy, n, d, h, w, cin, cout, stats, partial);
  if (ks > 1)
    splitk_finish_k<<<c.fin_blocks, 256, 0, st>>>(partial, ks, V, cout, bias, (bf16_t*)y, ldy, stats, slope);
  int rc = fplx_check_launch("mfma_conv3d_fwd");
  return rc < 0 ? rc : 1;
}

// returns 1 if handled, 0 if not applicable (caller falls back to the generic kernel), <0 on error
extern "C" int fplx_mfma_conv3d_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                    int n, int d, int h, int w, int cin, int cout, float* stats, void* ws,
                                    size_t ws_bytes, hipStream_t st) {
  return mfma_fwd_impl(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, ws, ws_bytes, 0, st);
}
extern "C" int fplx_mfma_conv3d_mid_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y,
                                        int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats, void* ws,
                                        size_t ws_bytes, hipStream_t st) {
  return mfma_fwd_impl(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, ws, ws_bytes, 1, st);
}

// 1 if the layer's forward kernel has the PReLU write-out (or finishes through splitk_finish_k)
// The launchers decline samples of 1 GiB and more (32-bit buffer offsets, the out-of-range marker 0x40000000) - a limit on
// d h w ldx the plan cannot see; mirrored here for every leading dimension the engine uses (ldx <= 2 cin: a half of a
// concatenation buffer), so that a caller who committed to the fused form on this answer is never refused at launch.
static inline bool act_sample_fits(int d, int h, int w, int cin) {
  return (int64_t)d * h * w * (2 * cin) * 2 < ((int64_t)1 << 30);
}
extern "C" int fplx_mfma_conv3d_act_ok(int n, int d, int h, int w, int cin, int cout, int mid) {
  int kernel, geo, ks;
  if (!act_sample_fits(d, h, w, cin)) return 0;
  if (!fplx_mfma_conv3d_plan(n, d, h, w, cin, cout, mid, &kernel, &geo, &ks)) return 0;
  return kernel == FPLX_KERNEL_BRICK || kernel == FPLX_KERNEL_MARCH || ((kernel == FPLX_KERNEL_TILE || kernel == FPLX_KERNEL_DIRECT) && ks > 1);
}
// the two-tensor input (x0 || x1, Cin / 2 channels each, one leading dimension; x0 read modulo nmod0 samples) of the brick
// kernel's activation form: the 3D layers it takes first, unsplit in Cin
extern "C" int fplx_mfma_conv3d_act_cat2_ok(int n, int d, int h, int w, int cin, int cout, int mid) {
  int kernel, geo, ks;
  if (mid || cin % 64 != 0 || !act_sample_fits(d, h, w, cin) || !fplx_mfma_conv3d_plan(n, d, h, w, cin, cout, mid, &kernel, &geo, &ks)) return 0;
  return kernel == FPLX_KERNEL_BRICK && ks == 1;
}
extern "C" int fplx_mfma_conv3d_fwd_act_cat2(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias,
                                             const float* slope, void* y, int64_t ldy, int n, int d, int h, int w, int cin,
                                             int cout, int nmod0, hipStream_t st) {
  if (!fplx_mfma_conv3d_act_cat2_ok(n, d, h, w, cin, cout, 0) || !mfma_applicable(ldx, ldy, cin, cout, x0, y, wp)) return 0;
  int geo, ks, bricks;
  fplx_brick_plan(n, d, h, w, cin, cout, &geo, &ks, &bricks);
  return fplx_brick_conv3d_fwd_act(x0, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, nullptr, nullptr, geo, ks, st, slope, x1, nmod0);
}
extern "C" int fplx_mfma_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, const float* slope, void* y,
                                        int64_t ldy, int n, int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes,
                                        int mid, hipStream_t st) {
  return mfma_fwd_impl(x, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, nullptr, ws, ws_bytes, mid, st, slope);
}

// conv_wgrad.hip (rolling-window weight gradient): same partial-tile format, reduced by wgrad_stream_reduce
extern "C" int fplx_wgroll_ok(int n, int d, int h, int w, int cin, int cout, int64_t ldx, int64_t ldy);
extern "C" size_t fplx_wgroll_ws_bytes(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_wgroll_conv3d_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n, int d, int h,
                                        int w, int cin, int cout, void* ws, size_t ws_bytes, hipStream_t st, const void* x1, int mid);
extern "C" int fplx_wgrad_reduce_launch(const float* part, int nblk, int npairs, int cin, int cout, float* dw, int mid,
                                        hipStream_t st) {
  const int64_t total = (int64_t)npairs * 27 * 1024;
  wgrad_finish(part, nblk, npairs, cin, cout, dw, mid, st);
  return fplx_check_launch("wgrad_stream_reduce");
}

extern "C" int fplx_mfma_conv3d_wgrad_cit(int n, int d, int h, int w, int cin, int cout) {
  if (cin % 32 != 0 || cout % 32 != 0) return 0;
  return wg_cfg(n, d, h, w, cin, cout).cit;
}

extern "C" size_t fplx_mfma_conv3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  if (cin % 32 != 0 || cout % 32 != 0) return 0;
  const size_t a = wg_cfg(n, d, h, w, cin, cout).ws;
  const VoxCfg v = vox_cfg(n, d, h, w, cin, cout);          // the largest of the three: the 2.5D form of a layer never takes vox
  size_t m = (v.ok && v.ws > a) ? v.ws : a;                 // or the rolling-window kernel, and a knob may switch kernels
  if (fplx_wgroll_ok(n, d, h, w, cin, cout, cin, cout)) { const size_t r = fplx_wgroll_ws_bytes(n, d, h, w, cin, cout); if (r > m) m = r; }
  return m;
}

// returns 1 if handled, 0 if not applicable, <0 on error.  dw fp32 [Cout][Cin][27]
// mid != 0: a Conv2d per depth slice - only the middle-plane taps are computed and dw is [Cout][Cin][3][3]
extern "C" int fplx_mfma_conv3d_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n, int d,
                                      int h, int w, int cin, int cout, void* ws, size_t ws_bytes, hipStream_t st,
                                      const void* x1, int mid) {
  if (cin % 32 != 0 || cout % 32 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16))
    return 0;
  if (!x1 && !mid) {
    const VoxCfg v = vox_cfg(n, d, h, w, cin, cout);
    if (v.ok) {
      if (ws_bytes < v.ws) return fplx_fail(FPLX_E_WORKSPACE, "mfma_conv3d_wgrad: workspace %zu < %zu", ws_bytes, v.ws);
      dim3 grid(v.g.S, v.npairs / v.cot);
      const int lwk = (int)fplx_knob(FPLX_K_WG_VOX_LW);
      if ((lwk == 2 || (lwk == 1 && v.g.V <= 2048)) && v.cot == 1 && (int64_t)v.g.V * ldx * 2 < ((int64_t)1 << 31) &&
          (int64_t)v.g.V * ldy * 2 < ((int64_t)1 << 31)) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_vox_lw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
        conv_wgrad_vox_lw<<<grid, 512, v.lds, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, cin, cout, v.g, fplx_xcd_on());
      } else if (v.cot == 2) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_vox<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
        conv_wgrad_vox<2><<<grid, 256, v.lds, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, cin, cout, v.g, fplx_xcd_on());
      } else {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_vox<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
        conv_wgrad_vox<1><<<grid, 256, v.lds, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, cin, cout, v.g, fplx_xcd_on());
      }
      wgrad_finish((const float*)ws, v.g.S, v.npairs, cin, cout, dw, 0, st);
      int rcv = fplx_check_launch("mfma_conv3d_wgrad_vox");
      return rcv < 0 ? rcv : 1;
    }
  }
  {                                                         // the rolling-window kernels (conv_wgrad.hip) where they apply
    const int rr = fplx_wgroll_conv3d_wgrad(x, ldx, dy, ldy, dw, n, d, h, w, cin, cout, ws, ws_bytes, st, x1, mid);
    if (rr != 0) return rr;
  }
  const WgCfg c = wg_cfg(n, d, h, w, cin, cout);
  if (x1 && (c.cit != 2 || cin != 64 || ((uintptr_t)x1 % 16))) return 0;   // split x: one group of two ci tiles
  if (ws_bytes < c.ws) return fplx_fail(FPLX_E_WORKSPACE, "mfma_conv3d_wgrad: workspace %zu < %zu", ws_bytes, c.ws);
  dim3 grid(c.nblk, c.npairs / (c.cit * c.cot));
#define LAUNCH_WG2(TW_, CIT_, COT_, TWOD_)                                                                          \
  do {                                                                                                              \
    const size_t lds = (size_t)(3 * CIT_ * (WG_TH + 2) * (TW_ + 2) + COT_ * WG_TH * TW_) * 64;                      \
    (void)hipFuncSetAttribute((const void*)conv_wgrad_stream<TW_, CIT_, COT_, TWOD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    conv_wgrad_stream<TW_, CIT_, COT_, TWOD_><<<grid, 256, lds, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, n, d, \
                                                                h, w, cin, cout, c.tilesH, c.tilesW, c.dsegs, c.dlen, (const bf16_t*)x1, fplx_xcd_on()); \
  } while (0)
#define LAUNCH_WG(TW_, CIT_, COT_) do { if (mid) LAUNCH_WG2(TW_, CIT_, COT_, true); else LAUNCH_WG2(TW_, CIT_, COT_, false); } while (0)
  if (c.tw == 32) { if (c.cot == 2) LAUNCH_WG(32, 1, 2); else if (c.cit == 2) LAUNCH_WG(32, 2, 1); else LAUNCH_WG(32, 1, 1); }
  else { if (c.cot == 2) LAUNCH_WG(16, 1, 2); else if (c.cit == 2) LAUNCH_WG(16, 2, 1); else LAUNCH_WG(16, 1, 1); }
#undef LAUNCH_WG
#undef LAUNCH_WG2
  const int64_t total = (int64_t)c.npairs * 27 * 1024;
  wgrad_finish((const float*)ws, c.nblk, c.npairs, cin, cout, dw, mid, st);
  int rc = fplx_check_launch("mfma_conv3d_wgrad");
  return rc < 0 ? rc : 1;
}

// ---- ConvTranspose3d(k=2,s=2) fast paths; same return convention
// sd = 2: ConvTranspose3d(k=2,s=2); sd = 1: ConvTranspose2d(k=2,s=2) on every depth slice (4 taps, packs [4][..][..])
extern "C" int fplx_mfma_deconv2_fwd(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy,
                                     int n, int d, int h, int w, int cin, int cout, int sd, hipStream_t st) {
  if (cin % 32 != 0 || cout % 32 != 0 || ldx % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)wf % 16)) return 0;
  const int64_t V = (int64_t)n * d * h * w;
  if (V >= ((int64_t)1 << 31)) return 0;
  const int vec_ok = ldy % 8 == 0 && ((uintptr_t)y % 16) == 0;
  {
    const int krows = (int)fplx_knob(FPLX_K_DECONV_ROWS);    // A/B knob
    const int ks = cin / 16, ntc = cout / 32;
    if (krows && sd == 2 && vec_ok && V >= 32 * 1024 && ((ks == 4 && ntc == 1) || (ks == 8 && ntc == 2) || (ks == 8 && ntc == 1) ||
                                                         (ks == 4 && ntc == 2))) {
      const int64_t nt = (V + 31) / 32;
      const unsigned nb = (unsigned)(nt < 1024 ? nt : 1024);
#define LAUNCH_DR(KS_, NTC_) deconv_fwd_rows<KS_, NTC_><<<nb, 256, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wf, bias, \
                                                                           (bf16_t*)y, ldy, n, d, h, w, fplx_xcd_on())
      if (ks == 4 && ntc == 1) LAUNCH_DR(4, 1);
      else if (ks == 8 && ntc == 2) LAUNCH_DR(8, 2);
      else if (ks == 8 && ntc == 1) LAUNCH_DR(8, 1);
      else LAUNCH_DR(4, 2);
#undef LAUNCH_DR
      int rc = fplx_check_launch("mfma_deconv2_fwd_rows");
      return rc < 0 ? rc : 1;
    }
  }
  dim3 grid((unsigned)((V + 127) / 128), cout / 32, sd);      // blockIdx.z = depth tap i (4 in-plane taps per block)
  deconv_fwd_mfma<<<grid, DIRECT_THREADS, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wf, bias, (bf16_t*)y, ldy, n, d,
                                                   h, w, cin, cout, sd, vec_ok, fplx_xcd_on());
  int rc = fplx_check_launch("mfma_deconv2_fwd");
  return rc < 0 ? rc : 1;
}

extern "C" int fplx_mfma_deconv2_dgrad(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx, int n, int d,
                                       int h, int w, int cin, int cout, int sd, hipStream_t st) {
  // GEMM view: K = 8 (or 4) taps x Cout (channels of dy), columns = Cin
  if (cout % 16 != 0 || cin % 32 != 0 || ldy % 8 != 0 || ((uintptr_t)dy % 16) || ((uintptr_t)wb % 16)) return 0;
  const int64_t V = (int64_t)n * d * h * w;
  if (V * 8 >= ((int64_t)1 << 31)) return 0;                 // the kernel decodes voxel indices in 32 bits
  // shallow levels: the coalesced row-segment stream (deconv_dgrad_rows)
  if (sd == 2 && fplx_knob(FPLX_K_DECONV_DGRAD_ROWS) && V >= 16 * 1024 && ldx % 8 == 0 && ((uintptr_t)dx % 16) == 0 &&
      ((cout == 32 && cin == 64) || (cout == 64 && cin == 128))) {
    const int mb = cin == 64 ? 64 : 32;
    const int segsW = (w + mb - 1) / mb;
    const int64_t nseg = (int64_t)n * d * h * segsW;
    const unsigned nb = (unsigned)(nseg < 512 ? nseg : 512);           // persistent: two blocks per CU (registers)
    if (cout == 32)
      deconv_dgrad_rows<1, 2><<<nb, 256, 0, st>>>((const bf16_t*)dy, ldy, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, segsW, nseg, fplx_xcd_on());
    else
      deconv_dgrad_rows<2, 4><<<nb, 256, 0, st>>>((const bf16_t*)dy, ldy, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, segsW, nseg, fplx_xcd_on());
    int rcr = fplx_check_launch("mfma_deconv2_dgrad_rows");
    return rcr < 0 ? rcr : 1;
  }
  // small deep volumes: the taps split over the waves of a block (deconv_dgrad_small)
  // (only where the direct kernel's grid leaves half the chip idle: at 8000 parents, 252 blocks, it is the faster one - 28
  // against 38 us - at 1000 parents, 64 blocks, 46 against 16 us)
  if (sd == 2 && fplx_knob(FPLX_K_DECONV_DGRAD_ROWS) && ((V + 127) / 128) * (cin / 64) < 128 && cin % 64 == 0 && cout % 16 == 0 &&
      ldx % 8 == 0 && ((uintptr_t)dx % 16) == 0) {
    dim3 gs((unsigned)((V + 31) / 32), cin / 64);
    deconv_dgrad_small<<<gs, 256, 0, st>>>((const bf16_t*)dy, ldy, (const bf16_t*)wb, (bf16_t*)dx, ldx, n, d, h, w, cout, cin);
    int rcs = fplx_check_launch("mfma_deconv2_dgrad_small");
    return rcs < 0 ? rcs : 1;
  }
#define LAUNCH_DD(MT_, NTL_, MODE_, GRID_)                                                                          \
  conv_fwd_direct<MT_, NTL_, MODE_><<<GRID_, DIRECT_THREADS, 0, st>>>((const bf16_t*)dy, ldy, (const bf16_t*)wb, nullptr, \
                                                                      (bf16_t*)dx, ldx, n, d, h, w, cout, cin, nullptr)
  if (cin % 64 == 0) {
    dim3 grid((unsigned)((V + 255) / 256), cin / 64);
    if ((int64_t)grid.x * grid.y < 192) {                     // deep levels: 128-voxel blocks, twice as many of them
      dim3 g1((unsigned)((V + 127) / 128), cin / 64);
      if (sd == 2) LAUNCH_DD(1, 2, 1, g1); else LAUNCH_DD(1, 2, 2, g1);
    } else if (sd == 2) LAUNCH_DD(2, 2, 1, grid);
    else LAUNCH_DD(2, 2, 2, grid);
  } else {
    dim3 grid((unsigned)((V + 511) / 512), cin / 32);
    if (sd == 2) LAUNCH_DD(4, 1, 1, grid); else LAUNCH_DD(4, 1, 2, grid);
  }
#undef LAUNCH_DD
  int rc = fplx_check_launch("mfma_deconv2_dgrad");
  return rc < 0 ? rc : 1;
}

extern "C" size_t fplx_mfma_deconv2_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  if (cin % 32 != 0 || cout % 32 != 0) return 0;
  return dw_cfg(n, d, h, w, cin, cout).ws;
}

extern "C" int fplx_mfma_deconv2_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db,
                                       int n, int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes,
                                       int sd, hipStream_t st) {
  if (cin % 32 != 0 || cout % 32 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16))
    return 0;
  const DwCfg c = dw_cfg(n, d, h, w, cin, cout);
  if (ws_bytes < c.ws) return fplx_fail(FPLX_E_WORKSPACE, "mfma_deconv2_wgrad: workspace %zu < %zu", ws_bytes, c.ws);
  dim3 grid(c.nblk, c.npairs);
  const size_t lds = (size_t)(c.cit + 8) * 128 * 64;
  float* bpart = db ? (float*)((char*)ws + (size_t)c.nblk * c.npairs * 8 * c.cit * 1024 * sizeof(float)) : nullptr;
#define LAUNCH_DW(CIT)                                                                                              \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)deconv_wgrad_mfma<CIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    deconv_wgrad_mfma<CIT><<<grid, 256, lds, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (float*)ws, bpart, n, d, \
                                                   h, w, cin, cout, sd);                                            \
  } while (0)
  if (c.cit == 4) LAUNCH_DW(4);
  else if (c.cit == 2) LAUNCH_DW(2);
  else LAUNCH_DW(1);
#undef LAUNCH_DW
  const int64_t total = (int64_t)c.npairs * 8 * c.cit * 1024;
  if (db) deconv_bias_reduce<<<cout, 64, 0, st>>>(bpart, c.nblk, cout, db);
  deconv_wgrad_reduce<<<(unsigned)((total + 63) / 64), 256, 0, st>>>((const float*)ws, c.nblk, c.npairs, c.cit, cin, cout, dw,
                                                                      4 * sd);
  int rc = fplx_check_launch("mfma_deconv2_wgrad");
  return rc < 0 ? rc : 1;
}
