// Generic (any shape, any stride, fp32 or bf16) convolution kernels: plain FMA, fp32 accumulate.
// They serve (a) the fp32 parity mode, (b) layers the MFMA kernels do not cover (odd channel
// counts), (c) the in-GPU cross-check of the MFMA kernels.  The bf16 hot path of the 3x3x3
// convolutions is in conv_mfma.hip.
#include "common.h"

namespace {

struct Strides { int64_t n, d, h, w, c; };

constexpr int CO_T = 8;     // output channels per thread (forward / dgrad)
constexpr int FWD_THREADS = 256;
constexpr int MAX_ROWS = 2048;

// ------------------------------------------------------------------------------------------
// forward: thread = one output voxel x CO_T output channels
template <typename TX, typename TW, typename TY>
__global__ void __launch_bounds__(FWD_THREADS)
conv_fwd_generic(const TX* __restrict__ x, Strides xs, const TW* __restrict__ wp, const float* __restrict__ bias,
                 TY* __restrict__ y, Strides ys, int N, int D, int H, int W, int Cin, int Cout,
                 int KD, int KH, int KW, float* __restrict__ stats, int64_t tiles) {
  const int co0 = blockIdx.y * CO_T;
  const int64_t V = (int64_t)N * D * H * W;
  float s[CO_T], q[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) s[j] = q[j] = 0.f;
  const int pd = KD / 2, ph = KH / 2, pw = KW / 2;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t v = tile * FWD_THREADS + threadIdx.x;
    const bool valid = v < V;
    int64_t r = valid ? v : 0;
    const int w0 = r % W; r /= W;
    const int h0 = r % H; r /= H;
    const int d0 = r % D; r /= D;
    const int n0 = (int)r;
    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.f;
    if (valid) {
      for (int kd = 0; kd < KD; ++kd) {
        const int dd = d0 + kd - pd;
        if (dd < 0 || dd >= D) continue;
        for (int kh = 0; kh < KH; ++kh) {
          const int hh = h0 + kh - ph;
          if (hh < 0 || hh >= H) continue;
          for (int kw = 0; kw < KW; ++kw) {
            const int ww = w0 + kw - pw;
            if (ww < 0 || ww >= W) continue;
            const TX* xp = x + n0 * xs.n + dd * xs.d + hh * xs.h + ww * xs.w;
            const TW* wt = wp + ((int64_t)((kd * KH + kh) * KW + kw) * Cout + co0) * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
              const float xv = Act<TX>::ld(xp + ci * xs.c);
#pragma unroll
              for (int j = 0; j < CO_T; ++j)
                if (co0 + j < Cout) acc[j] = fmaf(xv, Act<TW>::ld(wt + (int64_t)j * Cin + ci), acc[j]);
            }
          }
        }
      }
      TY* yp = y + n0 * ys.n + d0 * ys.d + h0 * ys.h + w0 * ys.w;
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
        if (co0 + j < Cout) {
          Act<TY>::st(yp + (co0 + j) * ys.c, acc[j]);
          s[j] += acc[j];
          q[j] += acc[j] * acc[j];
        }
    }
  }
  if (stats) {
    __shared__ float red[FWD_THREADS / 64][2 * CO_T];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CO_T; ++j) {
      const float a = wave_sum(s[j]), b = wave_sum(q[j]);
      if (lane == 0) { red[wv][j] = a; red[wv][CO_T + j] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CO_T) {
      float t = 0.f;
      for (int k = 0; k < FWD_THREADS / 64; ++k) t += red[k][threadIdx.x];
      const int j = threadIdx.x % CO_T, which = threadIdx.x / CO_T;
      if (co0 + j < Cout) stats[((int64_t)blockIdx.x * 2 + which) * Cout + co0 + j] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient: block = (voxel chunk, 8x8 (co,ci) tile, tap); thread = voxel lane with an
// 8x8 register tile; partial tiles [chunk][taps][Cout][Cin] are summed by wgrad_reduce.
constexpr int WG_T = 8;
constexpr int WG_THREADS = 256;

template <typename TX, typename TY, bool DECONV>
__global__ void __launch_bounds__(WG_THREADS)
wgrad_generic(const TX* __restrict__ x, Strides xs, const TY* __restrict__ dy, Strides ys, float* __restrict__ part,
              int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW, int chunks) {
  // D,H,W: for conv the (shared) spatial size; for DECONV the INPUT size (dy is 2x)
  const int taps = KD * KH * KW;
  const int tap = blockIdx.z;
  const int tiles_ci = (Cin + WG_T - 1) / WG_T;
  const int co0 = (blockIdx.y / tiles_ci) * WG_T, ci0 = (blockIdx.y % tiles_ci) * WG_T;
  const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
  const int64_t V = (int64_t)N * D * H * W;
  float acc[WG_T][WG_T];
#pragma unroll
  for (int a = 0; a < WG_T; ++a)
#pragma unroll
    for (int b = 0; b < WG_T; ++b) acc[a][b] = 0.f;
  const int64_t per = (V + chunks - 1) / chunks;
  const int64_t v0 = (int64_t)blockIdx.x * per, v1 = (v0 + per < V) ? v0 + per : V;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += WG_THREADS) {
    int64_t r = v;
    const int w0 = r % W; r /= W;
    const int h0 = r % H; r /= H;
    const int d0 = r % D; r /= D;
    const int n0 = (int)r;
    const TX* xp;
    const TY* gp;
    if (DECONV) {
      xp = x + n0 * xs.n + d0 * xs.d + h0 * xs.h + w0 * xs.w;
      gp = dy + n0 * ys.n + (2 * d0 + kd) * ys.d + (2 * h0 + kh) * ys.h + (2 * w0 + kw) * ys.w;
    } else {
      const int dd = d0 + kd - KD / 2, hh = h0 + kh - KH / 2, ww = w0 + kw - KW / 2;
      if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      xp = x + n0 * xs.n + dd * xs.d + hh * xs.h + ww * xs.w;
      gp = dy + n0 * ys.n + d0 * ys.d + h0 * ys.h + w0 * ys.w;
    }
    float xv[WG_T], gv[WG_T];
#pragma unroll
    for (int b = 0; b < WG_T; ++b) xv[b] = (ci0 + b < Cin) ? Act<TX>::ld(xp + (ci0 + b) * xs.c) : 0.f;
#pragma unroll
    for (int a = 0; a < WG_T; ++a) gv[a] = (co0 + a < Cout) ? Act<TY>::ld(gp + (co0 + a) * ys.c) : 0.f;
#pragma unroll
    for (int a = 0; a < WG_T; ++a)
#pragma unroll
      for (int b = 0; b < WG_T; ++b) acc[a][b] = fmaf(gv[a], xv[b], acc[a][b]);
  }
  __shared__ float red[WG_THREADS / 64][WG_T * WG_T];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < WG_T; ++a)
#pragma unroll
    for (int b = 0; b < WG_T; ++b) {
      const float t = wave_sum(acc[a][b]);
      if (lane == 0) red[wv][a * WG_T + b] = t;
    }
  __syncthreads();
  if (threadIdx.x < WG_T * WG_T) {
    float t = 0.f;
    for (int k = 0; k < WG_THREADS / 64; ++k) t += red[k][threadIdx.x];
    const int a = threadIdx.x / WG_T, b = threadIdx.x % WG_T;
    if (co0 + a < Cout && ci0 + b < Cin)
      part[(((int64_t)blockIdx.x * taps + tap) * Cout + co0 + a) * Cin + ci0 + b] = t;
  }
}

// part [chunks][taps][Cout][Cin] -> conv: dw[Cout][Cin][taps];  deconv: dw[Cin][Cout][taps]
__global__ void wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int chunks, int taps, int Cout,
                             int Cin, int deconv) {
  const int64_t total = (int64_t)taps * Cout * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int c = 0; c < chunks; ++c) t += part[(int64_t)c * total + i];
    const int ci = i % Cin, co = (i / Cin) % Cout, tap = (int)(i / ((int64_t)Cin * Cout));
    if (deconv) dw[((int64_t)ci * Cout + co) * taps + tap] = t;
    else dw[((int64_t)co * Cin + ci) * taps + tap] = t;
  }
}

// bias gradient: db[c] = sum over voxels of dy[v][c]; two-stage, rows partial blocks
template <typename TY>
__global__ void __launch_bounds__(256)
bias_grad_partial(const TY* __restrict__ dy, Strides ys, int N, int D, int H, int W, int C, float* __restrict__ part) {
  const int64_t V = (int64_t)N * D * H * W;
  const int c = blockIdx.y;
  float t = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    int64_t r = v;
    const int w0 = r % W; r /= W;
    const int h0 = r % H; r /= H;
    const int d0 = r % D; r /= D;
    t += Act<TY>::ld(dy + r * ys.n + d0 * ys.d + h0 * ys.h + w0 * ys.w + c * ys.c);
  }
  __shared__ float red[4];
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)blockIdx.x * C + c] = red[0] + red[1] + red[2] + red[3];
}
// channels-last variant (ys.c == 1): lanes run over channels, coalesced; grid (rows, ceil(C/64))
template <typename TY>
__global__ void __launch_bounds__(256)
bias_grad_partial_cl(const TY* __restrict__ dy, int64_t ld, int64_t V, int C, float* __restrict__ part) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int vl = threadIdx.x >> 6;
  float t = 0.f;
  if (c < C)
    for (int64_t v = (int64_t)blockIdx.x * 4 + vl; v < V; v += (int64_t)gridDim.x * 4) t += Act<TY>::ld(dy + v * ld + c);
  __shared__ float red[4][64];
  red[vl][threadIdx.x & 63] = t;
  __syncthreads();
  if (threadIdx.x < 64 && c < C)
    part[(int64_t)blockIdx.x * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void __launch_bounds__(64) bias_grad_reduce(const float* __restrict__ part, int rows, int C, float* __restrict__ db) {
  const int c = blockIdx.x;
  float t = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) t += part[(int64_t)r * C + c];
  t = wave_sum(t);
  if (threadIdx.x == 0) db[c] = t;
}

// ------------------------------------------------------------------------------------------
// transposed conv k2 s2: forward (thread = input voxel x CO_T couts, blockIdx.z = tap)
template <typename T>
__global__ void __launch_bounds__(FWD_THREADS)
deconv_fwd_generic(const T* __restrict__ x, int64_t ldx, const T* __restrict__ wf, const float* __restrict__ bias,
                   T* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int Cin, int Cout, int sd) {
  // sd = 2: ConvTranspose3d(k=2,s=2), 8 taps; sd = 1: ConvTranspose2d(k=2,s=2) on every depth slice, taps 0..3 (i = 0)
  const int tap = blockIdx.z, co0 = blockIdx.y * CO_T;
  const int i = tap >> 2, j = (tap >> 1) & 1, k = tap & 1;
  const int64_t V = (int64_t)N * D * H * W;
  const int64_t v = (int64_t)blockIdx.x * FWD_THREADS + threadIdx.x;
  if (v >= V) return;
  int64_t r = v;
  const int w0 = r % W; r /= W;
  const int h0 = r % H; r /= H;
  const int d0 = r % D; r /= D;
  const T* xp = x + v * ldx;
  const T* wt = wf + ((int64_t)tap * Cout + co0) * Cin;
  float acc[CO_T];
#pragma unroll
  for (int a = 0; a < CO_T; ++a) acc[a] = (co0 + a < Cout) ? bias[co0 + a] : 0.f;
  for (int ci = 0; ci < Cin; ++ci) {
    const float xv = Act<T>::ld(xp + ci);
#pragma unroll
    for (int a = 0; a < CO_T; ++a)
      if (co0 + a < Cout) acc[a] = fmaf(xv, Act<T>::ld(wt + (int64_t)a * Cin + ci), acc[a]);
  }
  T* yp = y + ((((int64_t)r * sd * D + sd * d0 + i) * 2 * H + 2 * h0 + j) * 2 * W + 2 * w0 + k) * ldy;
#pragma unroll
  for (int a = 0; a < CO_T; ++a)
    if (co0 + a < Cout) Act<T>::st(yp + co0 + a, acc[a]);
}

// data gradient: dx[v][ci] = sum_tap sum_co dy[out(v,tap)][co] * wb[tap][ci][co]
template <typename T>
__global__ void __launch_bounds__(FWD_THREADS)
deconv_dgrad_generic(const T* __restrict__ dy, int64_t ldy, const T* __restrict__ wb, T* __restrict__ dx, int64_t ldx,
                     int N, int D, int H, int W, int Cin, int Cout, int sd) {
  const int ci0 = blockIdx.y * CO_T;
  const int64_t V = (int64_t)N * D * H * W;
  const int64_t v = (int64_t)blockIdx.x * FWD_THREADS + threadIdx.x;
  if (v >= V) return;
  int64_t r = v;
  const int w0 = r % W; r /= W;
  const int h0 = r % H; r /= H;
  const int d0 = r % D; r /= D;
  float acc[CO_T];
#pragma unroll
  for (int a = 0; a < CO_T; ++a) acc[a] = 0.f;
  for (int tap = 0; tap < 4 * sd; ++tap) {
    const int i = tap >> 2, j = (tap >> 1) & 1, k = tap & 1;
    const T* gp = dy + ((((int64_t)r * sd * D + sd * d0 + i) * 2 * H + 2 * h0 + j) * 2 * W + 2 * w0 + k) * ldy;
    const T* wt = wb + ((int64_t)tap * Cin + ci0) * Cout;
    for (int co = 0; co < Cout; ++co) {
      const float g = Act<T>::ld(gp + co);
#pragma unroll
      for (int a = 0; a < CO_T; ++a)
        if (ci0 + a < Cin) acc[a] = fmaf(g, Act<T>::ld(wt + (int64_t)a * Cout + co), acc[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < CO_T; ++a)
    if (ci0 + a < Cin) Act<T>::st(dx + v * ldx + ci0 + a, acc[a]);
}

// ------------------------------------------------------------------------------------------
// weight packing
template <typename T>
__global__ void pack_conv_w(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int Cout, int Cin,
                            int taps) {
  const int64_t total = (int64_t)Cout * Cin * taps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = i % taps, ci = (i / taps) % Cin, co = (int)(i / ((int64_t)taps * Cin));
    const float v = w[i];
    Act<T>::st(wf + ((int64_t)tap * Cout + co) * Cin + ci, v);
    if (wb) Act<T>::st(wb + ((int64_t)(taps - 1 - tap) * Cin + ci) * Cout + co, v);
  }
}
// 3x3x3 bf16 pack through LDS: a block owns 16 co x 32 ci x 27 taps.  The fp32 master weights come in as 16 contiguous
// 3456-byte runs (float4 loads); both packed layouts go out as 16-byte stores - wf[tap][co][ci] in 64-byte runs,
// wb[26-tap][ci][co] in 32-byte runs - instead of one scattered 2-byte store per element and layout (the element-wise
// kernel above is store-issue-bound: 0.7 TB/s on the 512x512 layers).
constexpr int PK_CO = 16, PK_CI = 32, PK_ROW = PK_CI * 27 + 2;     // +2 bf16: odd dword stride between co rows
// Stamps (round 6): a tile's pack can be kept across optimiser steps (fplx_adam_pack_step writes it from the updated weights), and
// a writer that bypasses every version counter (p.data.mul_(), an EMA swap, init.*_(w.data), raw pointers) would leave it stale
// without anybody noticing.  So whoever writes a tile's pack also records PK_STAMP of the fp32 master values it was made from
// (two per output-channel row: elements 0 and 432 of the row's 864), and a `verify` pass re-reads those 32 values per tile and
// repacks exactly the tiles whose masters no longer match, bit for bit - any whole-tensor write is caught; single-element
// surgery between the sampled positions is not (such writers call Engine.invalidate()).
constexpr int PK_STAMP = 2 * PK_CO, PK_STAMP_Q = PK_CI * 27 / 4 / 2;     // 32 stamped floats per tile, one every 108 float4
__device__ __forceinline__ void pack27_tile(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wb,
                                            int Cout, int Cin, int blk, float* __restrict__ stamp = nullptr, int verify = 0) {
  __shared__ bf16_t lds[PK_CO * PK_ROW];
  const int ci_tiles = Cin / PK_CI;
  const int co0 = (blk / ci_tiles) * PK_CO, ci0 = (blk % ci_tiles) * PK_CI;
  constexpr int SEG4 = PK_CI * 27 / 4;                               // float4 per co run (216)
  if (stamp && verify) {
    int bad = 0;
    if (threadIdx.x < PK_STAMP) {
      const int co_l = threadIdx.x >> 1, q = (threadIdx.x & 1) * PK_STAMP_Q;
      const float cur = w[((int64_t)(co0 + co_l) * Cin + ci0) * 27 + 4 * q];
      bad = __float_as_uint(cur) != __float_as_uint(stamp[(int64_t)blk * PK_STAMP + threadIdx.x]);
    }
    if (!__syncthreads_or(bad)) return;                              // the pack was made from these masters: nothing to do
  }
  for (int i = threadIdx.x; i < PK_CO * SEG4; i += 256) {
    const int co_l = i / SEG4, q = i % SEG4;
    const float4 v = *reinterpret_cast<const float4*>(w + ((int64_t)(co0 + co_l) * Cin + ci0) * 27 + 4 * q);
    if (stamp && q % PK_STAMP_Q == 0) stamp[(int64_t)blk * PK_STAMP + co_l * 2 + q / PK_STAMP_Q] = v.x;
    bf16_t* d = lds + co_l * PK_ROW + 4 * q;
    Act<bf16_t>::st(d, v.x); Act<bf16_t>::st(d + 1, v.y); Act<bf16_t>::st(d + 2, v.z); Act<bf16_t>::st(d + 3, v.w);
  }
  __syncthreads();
  union Pack8 { bf16_t h[8]; uint4 u; };
  for (int i = threadIdx.x; i < 27 * PK_CO * (PK_CI / 8); i += 256) {          // wf: (tap, co, ci octet)
    const int oct = i % (PK_CI / 8), co_l = (i / (PK_CI / 8)) % PK_CO, tap = i / ((PK_CI / 8) * PK_CO);
    Pack8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p.h[j] = lds[co_l * PK_ROW + (8 * oct + j) * 27 + tap];
    *reinterpret_cast<uint4*>(wf + ((int64_t)tap * Cout + co0 + co_l) * Cin + ci0 + 8 * oct) = p.u;
  }
  if (wb)
    for (int i = threadIdx.x; i < 27 * PK_CI * (PK_CO / 8); i += 256) {        // wb: (tap, ci, co octet), taps flipped
      const int half = i % (PK_CO / 8), ci_l = (i / (PK_CO / 8)) % PK_CI, tap = i / ((PK_CO / 8) * PK_CI);
      Pack8 p;
#pragma unroll
      for (int j = 0; j < 8; ++j) p.h[j] = lds[(8 * half + j) * PK_ROW + ci_l * 27 + tap];
      *reinterpret_cast<uint4*>(wb + ((int64_t)(26 - tap) * Cin + ci0 + ci_l) * Cout + co0 + 8 * half) = p.u;
    }
}
__global__ void __launch_bounds__(256)
pack_conv_w27_tiled(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wb, int Cout, int Cin) {
  pack27_tile(w, wf, wb, Cout, Cin, blockIdx.x);
}
// every 3x3x3 layer of the network in ONE launch (the per-layer launches are 5-18 us each, mostly latency: 17 of them
// cost a training step 0.17 ms right after Adam, where nothing can overlap them): block -> (layer, tile) by a table
constexpr int PK_MAX = 32;
struct PackTable {
  const float* w[PK_MAX];
  bf16_t* wf[PK_MAX];
  bf16_t* wb[PK_MAX];
  float* stamp[PK_MAX];                // per layer: tiles x PK_STAMP floats, or NULL
  int cout[PK_MAX], cin[PK_MAX], first[PK_MAX + 1], n, verify;
};
__global__ void __launch_bounds__(256)
pack_conv_w27_tiled_multi(const PackTable t) {
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.first[e + 1]) ++e;
  pack27_tile(t.w[e], t.wf[e], t.wb[e], t.cout[e], t.cin[e], (int)blockIdx.x - t.first[e], t.stamp[e], t.verify);
}

// Adam + weight pack in ONE launch (round 5): the weights change only in the optimiser step, and every forward opened with
// pack_conv_w27_tiled_multi re-reading all 22 M master weights (48 us at the head of the step, where nothing overlaps it).
// Blocks [0, tiles): a pack tile (16 co x 32 ci x 27 taps) of one 3x3x3 layer - the block applies Adam to ITS 13 824
// elements (p, g, m, v as float4 runs, the arithmetic of adam_k: fplx_adam_elem) and packs the updated values from LDS;
// blocks [tiles, ...): the rest of the flat segment (biases, PReLU slopes, transposed convolutions, out_conv, the stem),
// `gap` ranges of the table, 1024 elements per block.  Every element of the segment is updated exactly once.
constexpr int AP_MAXGAP = PK_MAX + 1, AP_GAPBLK = 1024;
struct AdamPackTable {
  float *p, *m, *v;
  const float* g;
  FplxAdamConst c;
  int64_t off[PK_MAX];                 // element offset of layer e's weight inside the segment
  bf16_t* wf[PK_MAX];
  bf16_t* wb[PK_MAX];
  float* stamp[PK_MAX];                // see pack27_tile
  int cout[PK_MAX], cin[PK_MAX], first[PK_MAX + 1], n;
  int64_t gstart[AP_MAXGAP], glen[AP_MAXGAP];
  int gfirst[AP_MAXGAP + 1], ng;
};
__global__ void __launch_bounds__(256)
adam_pack27_multi(const AdamPackTable t) {
  const int b = (int)blockIdx.x;
  if (b >= t.first[t.n]) {             // ---- a gap block: plain Adam on up to 1024 elements
    const int gb = b - t.first[t.n];
    int e = 0;
    while (e + 1 < t.ng && gb >= t.gfirst[e + 1]) ++e;
    const int64_t lo = (int64_t)(gb - t.gfirst[e]) * AP_GAPBLK, len = t.glen[e];
#pragma unroll
    for (int k = 0; k < AP_GAPBLK / 256; ++k) {
      const int64_t i = lo + threadIdx.x + 256 * k;
      if (i < len) {
        const int64_t x = t.gstart[e] + i;
        float pi = t.p[x], mi = t.m[x], vi = t.v[x];
        fplx_adam_elem(pi, t.g[x], mi, vi, t.c);
        t.m[x] = mi; t.v[x] = vi; t.p[x] = pi;
      }
    }
    return;
  }
  int e = 0;
  while (e + 1 < t.n && b >= t.first[e + 1]) ++e;
  const int Cout = t.cout[e], Cin = t.cin[e], blk = b - t.first[e];
  __shared__ bf16_t lds[PK_CO * PK_ROW];
  const int ci_tiles = Cin / PK_CI;
  const int co0 = (blk / ci_tiles) * PK_CO, ci0 = (blk % ci_tiles) * PK_CI;
  constexpr int SEG4 = PK_CI * 27 / 4;
  for (int i = threadIdx.x; i < PK_CO * SEG4; i += 256) {
    const int co_l = i / SEG4, q = i % SEG4;
    const int64_t x = t.off[e] + ((int64_t)(co0 + co_l) * Cin + ci0) * 27 + 4 * q;
    float4 pv = *reinterpret_cast<const float4*>(t.p + x), mv = *reinterpret_cast<const float4*>(t.m + x);
    float4 vv = *reinterpret_cast<const float4*>(t.v + x);
    const float4 gv = *reinterpret_cast<const float4*>(t.g + x);
    fplx_adam_elem(pv.x, gv.x, mv.x, vv.x, t.c);
    fplx_adam_elem(pv.y, gv.y, mv.y, vv.y, t.c);
    fplx_adam_elem(pv.z, gv.z, mv.z, vv.z, t.c);
    fplx_adam_elem(pv.w, gv.w, mv.w, vv.w, t.c);
    *reinterpret_cast<float4*>(t.m + x) = mv;
    *reinterpret_cast<float4*>(t.v + x) = vv;
    *reinterpret_cast<float4*>(t.p + x) = pv;
    if (t.stamp[e] && q % PK_STAMP_Q == 0) t.stamp[e][(int64_t)blk * PK_STAMP + co_l * 2 + q / PK_STAMP_Q] = pv.x;
    bf16_t* d = lds + co_l * PK_ROW + 4 * q;
    Act<bf16_t>::st(d, pv.x); Act<bf16_t>::st(d + 1, pv.y); Act<bf16_t>::st(d + 2, pv.z); Act<bf16_t>::st(d + 3, pv.w);
  }
  __syncthreads();
  bf16_t* wf = t.wf[e];
  bf16_t* wb = t.wb[e];
  union Pack8 { bf16_t h[8]; uint4 u; };
  for (int i = threadIdx.x; i < 27 * PK_CO * (PK_CI / 8); i += 256) {          // wf: (tap, co, ci octet)
    const int oct = i % (PK_CI / 8), co_l = (i / (PK_CI / 8)) % PK_CO, tap = i / ((PK_CI / 8) * PK_CO);
    Pack8 pk;
#pragma unroll
    for (int j = 0; j < 8; ++j) pk.h[j] = lds[co_l * PK_ROW + (8 * oct + j) * 27 + tap];
    *reinterpret_cast<uint4*>(wf + ((int64_t)tap * Cout + co0 + co_l) * Cin + ci0 + 8 * oct) = pk.u;
  }
  if (wb)
    for (int i = threadIdx.x; i < 27 * PK_CI * (PK_CO / 8); i += 256) {        // wb: (tap, ci, co octet), taps flipped
      const int half = i % (PK_CO / 8), ci_l = (i / (PK_CO / 8)) % PK_CI, tap = i / ((PK_CO / 8) * PK_CI);
      Pack8 pk;
#pragma unroll
      for (int j = 0; j < 8; ++j) pk.h[j] = lds[(8 * half + j) * PK_ROW + ci_l * 27 + tap];
      *reinterpret_cast<uint4*>(wb + ((int64_t)(26 - tap) * Cin + ci0 + ci_l) * Cout + co0 + 8 * half) = pk.u;
    }
}

// the small packs of a step in ONE launch (round 5): stem, transposed convolutions, out_conv - seven launches of 3-6 us each with
// 10-us gaps between them at the head of every step, where nothing overlaps them
constexpr int PS_MAX = 16, PS_BLK = 1024;
struct PackSmallTable {
  const float* w[PS_MAX];
  void* wf[PS_MAX];
  void* wb[PS_MAX];
  int a[PS_MAX], b[PS_MAX], taps[PS_MAX], kind[PS_MAX], dt[PS_MAX], first[PS_MAX + 1], n;
};
template <typename T>
__device__ __forceinline__ void pack_small_elem(const PackSmallTable& t, int e, int64_t i) {
  const int taps = t.taps[e];
  const float v = t.w[e][i];
  T* wf = (T*)t.wf[e];
  T* wb = (T*)t.wb[e];
  if (t.kind[e] == 0) {                 // Conv: w[Cout = a][Cin = b][taps] -> wf[tap][Cout][Cin], wb[taps-1-tap][Cin][Cout]
    const int Cout = t.a[e], Cin = t.b[e];
    const int tap = (int)(i % taps), ci = (int)((i / taps) % Cin), co = (int)(i / ((int64_t)taps * Cin));
    if (wf) Act<T>::st(wf + ((int64_t)tap * Cout + co) * Cin + ci, v);
    if (wb) Act<T>::st(wb + ((int64_t)(taps - 1 - tap) * Cin + ci) * Cout + co, v);
  } else {                              // ConvTranspose: w[Cin = a][Cout = b][taps] -> wf[tap][Cout][Cin], wb[tap][Cin][Cout]
    const int Cin = t.a[e], Cout = t.b[e];
    const int tap = (int)(i % taps), co = (int)((i / taps) % Cout), ci = (int)(i / ((int64_t)taps * Cout));
    if (wf) Act<T>::st(wf + ((int64_t)tap * Cout + co) * Cin + ci, v);
    if (wb) Act<T>::st(wb + ((int64_t)tap * Cin + ci) * Cout + co, v);
  }
}
__global__ void __launch_bounds__(256)
pack_small_multi(const PackSmallTable t) {
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.first[e + 1]) ++e;
  const int64_t total = (int64_t)t.a[e] * t.b[e] * t.taps[e];
  const int64_t lo = (int64_t)((int)blockIdx.x - t.first[e]) * PS_BLK;
#pragma unroll
  for (int k = 0; k < PS_BLK / 256; ++k) {
    const int64_t i = lo + threadIdx.x + 256 * k;
    if (i < total) {
      if (t.dt[e] == FPLX_F32) pack_small_elem<float>(t, e, i);
      else pack_small_elem<bf16_t>(t, e, i);
    }
  }
}

template <typename T>
__global__ void pack_deconv_w(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int Cin, int Cout,
                              int taps) {
  const int64_t total = (int64_t)Cin * Cout * taps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = i % taps, co = (i / taps) % Cout, ci = (int)(i / (taps * (int64_t)Cout));
    const float v = w[i];
    Act<T>::st(wf + ((int64_t)tap * Cout + co) * Cin + ci, v);
    if (wb) Act<T>::st(wb + ((int64_t)tap * Cin + ci) * Cout + co, v);
  }
}

// 2.5D levels (conv_dims = 2): a Conv2d(3x3) on every depth slice IS a Conv3d whose kd = 0 and kd = 2 planes are zero.
// Packing the 9 taps into the middle plane of the 27-tap layouts lets every 3x3x3 kernel (forward, data gradient)
// run unchanged and bit-identically (products with an exact zero add nothing); the weight gradient is the middle plane
// of the 27-tap gradient.
template <typename T>
__global__ void pack_conv2d_w_as3d(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int Cout, int Cin) {
  const int64_t total = (int64_t)Cout * Cin * 27;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = i % 27, ci = (i / 27) % Cin, co = (int)(i / ((int64_t)27 * Cin));
    const float v = (tap >= 9 && tap < 18) ? w[((int64_t)co * Cin + ci) * 9 + tap - 9] : 0.f;
    Act<T>::st(wf + ((int64_t)tap * Cout + co) * Cin + ci, v);
    if (wb) Act<T>::st(wb + ((int64_t)(26 - tap) * Cin + ci) * Cout + co, v);
  }
}
__global__ void extract_mid_plane(const float* __restrict__ dw27, float* __restrict__ dw9, int64_t pairs) {
  const int64_t total = pairs * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    dw9[i] = dw27[(i / 9) * 27 + 9 + i % 9];
}

template <typename TY>
void launch_bias_grad(const TY* dy, Strides ys, int n, int d, int h, int w, int c, float* bpart, float* db, hipStream_t st) {
  const int64_t V = (int64_t)n * d * h * w;
  const int rows = fplx_rows_for(V);
  const bool cl = ys.c == 1 && ys.h == ys.w * w && ys.d == ys.h * h && ys.n == ys.d * d;
  if (cl) {
    dim3 g(rows, (c + 63) / 64);
    bias_grad_partial_cl<TY><<<g, 256, 0, st>>>(dy, ys.w, V, c, bpart);
  } else {
    dim3 g(rows, c);
    bias_grad_partial<TY><<<g, 256, 0, st>>>(dy, ys, n, d, h, w, c, bpart);
  }
  bias_grad_reduce<<<c, 64, 0, st>>>(bpart, rows, c, db);
}

inline int grid_for(int64_t n, int threads, int cap) {
  int64_t g = (n + threads - 1) / threads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline int wgrad_chunks(int64_t V) {
  int64_t c = (V + 8191) / 8192;
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  return (int)c;
}

}  // namespace

// MFMA fast paths (conv_mfma.hip); return 1 if they handled the call, 0 if not applicable, <0 on error
extern "C" int fplx_mfma_conv3d_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                    int n, int d, int h, int w, int cin, int cout, float* stats, void* ws,
                                    size_t ws_bytes, hipStream_t st);
extern "C" size_t fplx_mfma_conv3d_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_mfma_conv3d_act_ok(int n, int d, int h, int w, int cin, int cout, int mid);
extern "C" int fplx_mfma_conv3d_act_cat2_ok(int n, int d, int h, int w, int cin, int cout, int mid);
extern "C" int fplx_mfma_conv3d_fwd_act_cat2(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias,
                                             const float* slope, void* y, int64_t ldy, int n, int d, int h, int w, int cin,
                                             int cout, int nmod0, hipStream_t st);
extern "C" int fplx_mfma_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, const float* slope, void* y,
                                        int64_t ldy, int n, int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes,
                                        int mid, hipStream_t st);
extern "C" int fplx_march_conv3d_fwd_act(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                         int n, int d, int h, int w, int cin, int cout, float* stats, hipStream_t st,
                                         const void* x1, void* y1, int twod, const float* slope, int nmod0);
extern "C" int fplx_mfma_conv3d_plan(int n, int d, int h, int w, int cin, int cout, int mid, int* kernel, int* geo, int* ksplit);
extern "C" int fplx_mfma_conv3d_stats_rows(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_edge_stem_rows(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_edge_stem_fwd(const float* x, const void* wf, const float* bias, void* y, int64_t ldy, int n, int d,
                                  int h, int w, int cin, int cout, float* stats, hipStream_t st);
extern "C" size_t fplx_edge_stem_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_edge_stem_wgrad_bn(const float* x, const void* dy, int64_t ldy, float* dw, int n, int d, int h, int w,
                                       int cin, int cout, void* ws, hipStream_t st, const void* y, int64_t ldyy, const float* mean,
                                       const float* rstd, const float* scale, const float* shift, const float* slope,
                                       const float* coef);
extern "C" int fplx_edge_stem_wgrad(const float* x, const void* dy, int64_t ldy, float* dw, int n, int d, int h, int w,
                                    int cin, int cout, void* ws, hipStream_t st);
extern "C" int fplx_edge_outconv_fwd(const void* x, int64_t ldx, const float* wf, const float* bias, float* out, int n,
                                     int d, int h, int w, int cin, int ncls, hipStream_t st);
extern "C" int fplx_edge_outconv_bn_ok(int n, int d, int h, int w, int c0, int ncls);
extern "C" int fplx_edge_outconv_bn_rows(int n, int d, int h, int w, int ncls);
extern "C" int fplx_edge_outconv_fwd_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* slope,
                                        void* a, int64_t lda, const float* wf, const float* bias, float* out, int n, int d,
                                        int h, int w, int c0, int ncls, hipStream_t st);
extern "C" int fplx_edge_outconv_dgrad_bn(int mode, const float* dl, const void* wb, const void* y, int64_t ldy,
                                          const float* mean, const float* rstd, const float* scale, const float* shift,
                                          const float* slope, const float* coef, float* part, void* dy, int64_t lddy, int n,
                                          int d, int h, int w, int c0, int ncls, hipStream_t st);
extern "C" int fplx_edge_outconv_dgrad(const float* dl, const void* wb, void* dx, int64_t ldx, int n, int d, int h, int w,
                                       int c0, int ncls, hipStream_t st);
extern "C" size_t fplx_edge_outconv_wgrad_ws_bytes(int n, int d, int h, int w, int c0, int ncls);
extern "C" size_t fplx_edge_outconv_wgrad_bn_ws_bytes(int n, int d, int h, int w, int c0, int ncls);
extern "C" int fplx_edge_outconv_wgrad_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* slope,
                                          const float* dl, float* dw, float* db, int n, int d, int h, int w, int c0, int ncls,
                                          void* ws, hipStream_t st);
extern "C" int fplx_edge_outconv_wgrad(const void* x, int64_t ldx, const float* dl, float* dw, int n, int d, int h, int w,
                                       int c0, int ncls, void* ws, hipStream_t st);
extern "C" int fplx_mfma_deconv2_fwd(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy,
                                     int n, int d, int h, int w, int cin, int cout, int sd, hipStream_t st);
extern "C" int fplx_mfma_deconv2_dgrad(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx, int n, int d,
                                       int h, int w, int cin, int cout, int sd, hipStream_t st);
extern "C" size_t fplx_mfma_deconv2_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_mfma_deconv2_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db,
                                       int n, int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes,
                                       int sd, hipStream_t st);
extern "C" size_t fplx_mfma_conv3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_mfma_conv3d_mid_stats_rows(int n, int d, int h, int w, int cin, int cout);
extern "C" size_t fplx_mfma_conv3d_mid_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_mfma_conv3d_mid_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y,
                                        int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats, void* ws,
                                        size_t ws_bytes, hipStream_t st);
extern "C" int fplx_mfma_conv3d_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n, int d,
                                      int h, int w, int cin, int cout, void* ws, size_t ws_bytes, hipStream_t st,
                                      const void* x1, int mid);
extern "C" int fplx_mfma_conv3d_wgrad_cit(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_march_ok(int n, int d, int h, int w, int cin, int cout);
extern "C" int fplx_march_conv3d_fwd(const void* x, int64_t ldx, const void* wp, const float* bias, void* y, int64_t ldy,
                                     int n, int d, int h, int w, int cin, int cout, float* stats, hipStream_t st,
                                     const void* x1, void* y1, int twod);

extern "C" {

int fplx_version(void) { return 1; }

int fplx_last_error(char* buf, size_t n) {
  const char* e = fplx_err_buf();
  size_t l = strlen(e);
  if (buf && n) {
    size_t c = l < n - 1 ? l : n - 1;
    memcpy(buf, e, c);
    buf[c] = 0;
  }
  return (int)l;
}

int fplx_num_partials(int64_t voxels) { return fplx_rows_for(voxels); }

/* ---- tuning table (common.h: FPLX_KNOB_LIST) ---- */
int64_t fplx_knob_values[FPLX_K_COUNT] = {
#define FPLX_KNOB_DEF(id, key, def) def,
    FPLX_KNOB_LIST(FPLX_KNOB_DEF)
#undef FPLX_KNOB_DEF
};
static const char* const knob_keys[FPLX_K_COUNT] = {
#define FPLX_KNOB_KEY(id, key, def) key,
    FPLX_KNOB_LIST(FPLX_KNOB_KEY)
#undef FPLX_KNOB_KEY
};
static int knob_index(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < FPLX_K_COUNT; ++i)
    if (strcmp(key, knob_keys[i]) == 0) return i;
  return -1;
}
int fplx_set_tuning(const char* key, int64_t value) {
  const int i = knob_index(key);
  FPLX_REQUIRE(i >= 0, FPLX_E_BADSHAPE, "set_tuning: unknown key '%s'", key ? key : "(null)");
  __atomic_store_n(&fplx_knob_values[i], value, __ATOMIC_RELAXED);
  return FPLX_OK;
}
int fplx_get_tuning(const char* key, int64_t* value) {
  const int i = knob_index(key);
  FPLX_REQUIRE(i >= 0 && value, FPLX_E_BADSHAPE, "get_tuning: unknown key '%s'", key ? key : "(null)");
  *value = fplx_knob(i);
  return FPLX_OK;
}
int fplx_tuning_key(int index, char* buf, size_t n) {
  if (index < 0 || index >= FPLX_K_COUNT) return -1;
  const size_t l = strlen(knob_keys[index]);
  if (buf && n) {
    const size_t c = l < n - 1 ? l : n - 1;
    memcpy(buf, knob_keys[index], c);
    buf[c] = 0;
  }
  return (int)l;
}

int fplx_pack_conv_weight(const float* w, void* wf, void* wb, int cout, int cin, int kd, int kh, int kw, int dt,
                          fplx_stream_t stream) {
  FPLX_REQUIRE(w && wf, FPLX_E_NULL, "pack_conv_weight: null pointer");
  FPLX_REQUIRE(cout > 0 && cin > 0 && kd > 0 && kh > 0 && kw > 0, FPLX_E_BADSHAPE, "pack_conv_weight: bad shape");
  const int taps = kd * kh * kw;
  const int64_t total = (int64_t)cout * cin * taps;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_F32)
    pack_conv_w<float><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (float*)wf, (float*)wb, cout, cin, taps);
  else if (dt == FPLX_BF16) {
    const bool tiled = fplx_knob(FPLX_K_PACK_TILED) != 0;
    if (tiled && taps == 27 && cin % PK_CI == 0 && cout % PK_CO == 0 && ((uintptr_t)w % 16 == 0) &&
        ((uintptr_t)wf % 16 == 0) && ((uintptr_t)wb % 16 == 0))
      pack_conv_w27_tiled<<<(cout / PK_CO) * (cin / PK_CI), 256, 0, st>>>(w, (bf16_t*)wf, (bf16_t*)wb, cout, cin);
    else
      pack_conv_w<bf16_t><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (bf16_t*)wf, (bf16_t*)wb, cout, cin, taps);
  }
  else
    return fplx_fail(FPLX_E_BADDTYPE, "pack_conv_weight: dtype %d", dt);
  return fplx_check_launch("pack_conv_weight");
}

int fplx_pack_conv_weights_batched(int n, const float* const* w, void* const* wf, void* const* wb, const int* cout,
                                   const int* cin, int dt, float* const* stamp, int verify, fplx_stream_t stream) {
  FPLX_REQUIRE(w && wf && wb && cout && cin, FPLX_E_NULL, "pack_conv_weights_batched: null pointer");
  FPLX_REQUIRE(!verify || stamp, FPLX_E_NULL, "pack_conv_weights_batched: verify needs the stamps of the previous pack");
  FPLX_REQUIRE(n > 0 && n <= PK_MAX, FPLX_E_BADSHAPE, "pack_conv_weights_batched: %d layers (1..%d)", n, PK_MAX);
  const bool tiled = fplx_knob(FPLX_K_PACK_TILED) != 0;
  const bool multi = fplx_knob(FPLX_K_PACK_MULTI) != 0;     // A/B knob
  PackTable t;
  t.n = 0;
  t.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    FPLX_REQUIRE(w[i] && wf[i] && cout[i] > 0 && cin[i] > 0, FPLX_E_NULL, "pack_conv_weights_batched: layer %d", i);
    const bool ok = multi && tiled && dt == FPLX_BF16 && cin[i] % PK_CI == 0 && cout[i] % PK_CO == 0 &&
                    ((uintptr_t)w[i] % 16 == 0) && ((uintptr_t)wf[i] % 16 == 0) && ((uintptr_t)wb[i] % 16 == 0);
    FPLX_REQUIRE(ok || !(stamp && stamp[i]), FPLX_E_BADSHAPE, "pack_conv_weights_batched: layer %d has no tiled pack, hence no stamps "
                 "(fplx_adam_pack_ok)", i);
    if (!ok) {                                                // layers the tiled kernel does not take: one by one
      const int rc = fplx_pack_conv_weight(w[i], wf[i], wb[i], cout[i], cin[i], 3, 3, 3, dt, stream);
      if (rc != FPLX_OK) return rc;
      continue;
    }
    const int k = t.n++;
    t.w[k] = w[i]; t.wf[k] = (bf16_t*)wf[i]; t.wb[k] = (bf16_t*)wb[i]; t.cout[k] = cout[i]; t.cin[k] = cin[i];
    t.stamp[k] = stamp ? stamp[i] : nullptr;
    FPLX_REQUIRE(!verify || t.stamp[k], FPLX_E_NULL, "pack_conv_weights_batched: verify: layer %d has no stamps", i);
    t.first[k + 1] = t.first[k] + (cout[i] / PK_CO) * (cin[i] / PK_CI);
  }
  t.verify = verify ? 1 : 0;
  if (t.n == 0) return FPLX_OK;
  pack_conv_w27_tiled_multi<<<t.first[t.n], 256, 0, (hipStream_t)stream>>>(t);
  return fplx_check_launch("pack_conv_weights_batched");
}

int fplx_pack_weights_multi(int n, const int* kind, const float* const* w, void* const* wf, void* const* wb, const int* a,
                            const int* b, const int* taps, const int* dt, fplx_stream_t stream) {
  FPLX_REQUIRE(kind && w && wf && wb && a && b && taps && dt, FPLX_E_NULL, "pack_weights_multi: null pointer");
  FPLX_REQUIRE(n > 0 && n <= PS_MAX, FPLX_E_BADSHAPE, "pack_weights_multi: %d jobs (1..%d)", n, PS_MAX);
  PackSmallTable t;
  t.n = n;
  t.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    FPLX_REQUIRE(w[i] && (wf[i] || wb[i]) && a[i] > 0 && b[i] > 0 && taps[i] > 0 && (kind[i] == 0 || kind[i] == 1), FPLX_E_BADSHAPE,
                 "pack_weights_multi: job %d", i);
    FPLX_REQUIRE(dt[i] == FPLX_F32 || dt[i] == FPLX_BF16, FPLX_E_BADDTYPE, "pack_weights_multi: dtype %d", dt[i]);
    t.w[i] = w[i]; t.wf[i] = wf[i]; t.wb[i] = wb[i]; t.a[i] = a[i]; t.b[i] = b[i]; t.taps[i] = taps[i]; t.kind[i] = kind[i]; t.dt[i] = dt[i];
    const int64_t total = (int64_t)a[i] * b[i] * taps[i];
    FPLX_REQUIRE(total < ((int64_t)1 << 30), FPLX_E_BADSHAPE, "pack_weights_multi: job %d too large for this entry point", i);
    t.first[i + 1] = t.first[i] + (int)((total + PS_BLK - 1) / PS_BLK);
  }
  pack_small_multi<<<t.first[n], 256, 0, (hipStream_t)stream>>>(t);
  return fplx_check_launch("pack_weights_multi");
}

int fplx_adam_pack_ok(int cout, int cin) {
  return fplx_knob(FPLX_K_PACK_TILED) != 0 && fplx_knob(FPLX_K_PACK_MULTI) != 0 && cin % PK_CI == 0 && cout % PK_CO == 0;
}

int fplx_adam_pack_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int nl, const int64_t* off, const int* cout,
                        const int* cin, void* const* wf, void* const* wb, float* const* stamp, fplx_stream_t stream) {
  FPLX_REQUIRE(p && g && m && v && off && cout && cin && wf && wb, FPLX_E_NULL, "adam_pack_step: null pointer");
  FPLX_REQUIRE(n > 0 && step >= 1 && nl > 0 && nl <= PK_MAX, FPLX_E_BADSHAPE, "adam_pack_step: n=%lld step=%d layers=%d (1..%d)",
               (long long)n, step, nl, PK_MAX);
  FPLX_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
               FPLX_E_BADSHAPE, "adam_pack_step: the flat buffers must be 16-byte aligned");
  AdamPackTable t;
  t.p = p; t.g = g; t.m = m; t.v = v;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  t.c = {(float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale, 1.f - beta1, 1.f - beta2};
  t.n = nl;
  t.first[0] = 0;
  t.ng = 0;
  t.gfirst[0] = 0;
  int64_t pos = 0;
  auto gap = [&](int64_t a, int64_t b_) {
    if (b_ <= a) return;
    t.gstart[t.ng] = a; t.glen[t.ng] = b_ - a;
    t.gfirst[t.ng + 1] = t.gfirst[t.ng] + (int)((b_ - a + AP_GAPBLK - 1) / AP_GAPBLK);
    ++t.ng;
  };
  for (int i = 0; i < nl; ++i) {
    const int64_t len = (int64_t)cout[i] * cin[i] * 27;
    FPLX_REQUIRE(fplx_adam_pack_ok(cout[i], cin[i]) && wf[i], FPLX_E_BADSHAPE, "adam_pack_step: layer %d (%d x %d) not packable here (fplx_adam_pack_ok)", i, cout[i], cin[i]);
    FPLX_REQUIRE(off[i] >= pos && off[i] % 4 == 0 && off[i] + len <= n, FPLX_E_BADSHAPE, "adam_pack_step: layer %d: offsets must ascend, be multiples of 4 and lie inside the segment", i);
    FPLX_REQUIRE(((uintptr_t)wf[i] % 16) == 0 && ((uintptr_t)wb[i] % 16) == 0, FPLX_E_BADSHAPE, "adam_pack_step: packs must be 16-byte aligned");
    gap(pos, off[i]);
    t.off[i] = off[i]; t.cout[i] = cout[i]; t.cin[i] = cin[i]; t.wf[i] = (bf16_t*)wf[i]; t.wb[i] = (bf16_t*)wb[i];
    t.stamp[i] = stamp ? stamp[i] : nullptr;
    t.first[i + 1] = t.first[i] + (cout[i] / PK_CO) * (cin[i] / PK_CI);
    pos = off[i] + len;
  }
  gap(pos, n);
  adam_pack27_multi<<<t.first[t.n] + t.gfirst[t.ng], 256, 0, (hipStream_t)stream>>>(t);
  return fplx_check_launch("adam_pack_step");
}

int fplx_pack_deconv_weight(const float* w, void* wf, void* wb, int cin, int cout, int dt, fplx_stream_t stream) {
  FPLX_REQUIRE(w && wf, FPLX_E_NULL, "pack_deconv_weight: null pointer");
  FPLX_REQUIRE(cout > 0 && cin > 0, FPLX_E_BADSHAPE, "pack_deconv_weight: bad shape");
  const int64_t total = (int64_t)cout * cin * 8;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_F32)
    pack_deconv_w<float><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (float*)wf, (float*)wb, cin, cout, 8);
  else if (dt == FPLX_BF16)
    pack_deconv_w<bf16_t><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (bf16_t*)wf, (bf16_t*)wb, cin, cout, 8);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "pack_deconv_weight: dtype %d", dt);
  return fplx_check_launch("pack_deconv_weight");
}

int fplx_pack_deconv122_weight(const float* w, void* wf, void* wb, int cin, int cout, int dt, fplx_stream_t stream) {
  FPLX_REQUIRE(w && wf, FPLX_E_NULL, "pack_deconv122_weight: null pointer");
  FPLX_REQUIRE(cout > 0 && cin > 0, FPLX_E_BADSHAPE, "pack_deconv122_weight: bad shape");
  const int64_t total = (int64_t)cout * cin * 4;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_F32)
    pack_deconv_w<float><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (float*)wf, (float*)wb, cin, cout, 4);
  else if (dt == FPLX_BF16)
    pack_deconv_w<bf16_t><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (bf16_t*)wf, (bf16_t*)wb, cin, cout, 4);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "pack_deconv122_weight: dtype %d", dt);
  return fplx_check_launch("pack_deconv122_weight");
}

int fplx_pack_conv2d_weight(const float* w, void* wf, void* wb, int cout, int cin, int dt, fplx_stream_t stream) {
  FPLX_REQUIRE(w && wf, FPLX_E_NULL, "pack_conv2d_weight: null pointer");
  FPLX_REQUIRE(cout > 0 && cin > 0, FPLX_E_BADSHAPE, "pack_conv2d_weight: bad shape");
  const int64_t total = (int64_t)cout * cin * 27;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_F32)
    pack_conv2d_w_as3d<float><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (float*)wf, (float*)wb, cout, cin);
  else if (dt == FPLX_BF16)
    pack_conv2d_w_as3d<bf16_t><<<grid_for(total, 256, 1024), 256, 0, st>>>(w, (bf16_t*)wf, (bf16_t*)wb, cout, cin);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "pack_conv2d_weight: dtype %d", dt);
  return fplx_check_launch("pack_conv2d_weight");
}

static int conv2d_wgrad_extract(const float* dw27, float* dw9, int cout, int cin, fplx_stream_t stream) {
  FPLX_REQUIRE(dw27 && dw9, FPLX_E_NULL, "conv2d_wgrad_extract: null pointer");
  FPLX_REQUIRE(cout > 0 && cin > 0, FPLX_E_BADSHAPE, "conv2d_wgrad_extract: bad shape");
  const int64_t pairs = (int64_t)cout * cin;
  extract_mid_plane<<<grid_for(pairs * 9, 256, 1024), 256, 0, (hipStream_t)stream>>>(dw27, dw9, pairs);
  return fplx_check_launch("conv2d_wgrad_extract");
}

static bool is_cl(int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc, int d, int h, int w) {
  return sc == 1 && sh == sw * w && sd == sh * h && sn == sd * d;          // NDHWC with voxel stride sw
}
static bool is_planar(int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc, int c, int d, int h, int w) {
  return sw == 1 && sh == w && sd == (int64_t)h * w && sc == (int64_t)d * h * w && sn == sc * c;   // contiguous NCDHW
}

int fplx_conv3d_stats_rows(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw, int x_dt, int y_dt) {
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3) {
    int r = fplx_mfma_conv3d_stats_rows(n, d, h, w, cin, cout);
    if (r > 0) return r;
  }
  if (x_dt == FPLX_F32 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3) {   // stem: fp32 network input
    int r = fplx_edge_stem_rows(n, d, h, w, cin, cout);
    if (r > 0) return r;
  }
  const int64_t V = (int64_t)n * d * h * w;
  return grid_for(V, FWD_THREADS, MAX_ROWS);
}

size_t fplx_conv3d_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw, int x_dt, int y_dt) {
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3)
    return fplx_mfma_conv3d_fwd_ws_bytes(n, d, h, w, cin, cout);
  return 0;
}

int fplx_conv3d_plan_query(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw, int x_dt, int y_dt,
                           int* kernel, int* geometry, int* ksplit, int* stats_rows) {
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, FPLX_E_BADSHAPE, "conv3d_plan_query: bad shape");
  int k = FPLX_KERNEL_GENERIC, g = -1, ks = 1;
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3) fplx_mfma_conv3d_plan(n, d, h, w, cin, cout, 0, &k, &g, &ks);
  else if (x_dt == FPLX_F32 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3 && fplx_edge_stem_rows(n, d, h, w, cin, cout) > 0)
    k = FPLX_KERNEL_STEM;
  else if (kd == 1 && kh == 3 && kw == 3 && ((x_dt == FPLX_BF16 && y_dt == FPLX_F32 && (cin == 16 || cin == 32 || cin == 64) && cout <= 4) ||
                                             (x_dt == FPLX_F32 && y_dt == FPLX_BF16 && (cout == 16 || cout == 32 || cout == 64) && cin <= 4)))
    k = FPLX_KERNEL_OUTCONV;
  if (kernel) *kernel = k;
  if (geometry) *geometry = g;
  if (ksplit) *ksplit = ks;
  if (stats_rows) *stats_rows = fplx_conv3d_stats_rows(n, d, h, w, cin, cout, kd, kh, kw, x_dt, y_dt);
  return FPLX_OK;
}

int fplx_conv3d_fwd(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc,
                    const void* wp, const float* bias, void* y, int y_dt, int64_t yn, int64_t yd, int64_t yh,
                    int64_t yw, int64_t yc, int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw,
                    float* stats, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  FPLX_REQUIRE(x && wp && y, FPLX_E_NULL, "conv3d_fwd: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, FPLX_E_BADSHAPE,
               "conv3d_fwd: bad shape n=%d d=%d h=%d w=%d cin=%d cout=%d", n, d, h, w, cin, cout);
  FPLX_REQUIRE((kd & 1) && (kh & 1) && (kw & 1) && kd <= 3 && kh <= 3 && kw <= 3, FPLX_E_BADSHAPE,
               "conv3d_fwd: kernel %dx%dx%d unsupported (odd sizes <= 3)", kd, kh, kw);
  hipStream_t st = (hipStream_t)stream;
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3 && sc == 1 && yc == 1 &&
      sh == sw * w && sd == sh * h && sn == sd * d && yh == yw * w && yd == yh * h && yn == yd * d) {
    int r = fplx_mfma_conv3d_fwd(x, sw, wp, bias, y, yw, n, d, h, w, cin, cout, stats, ws, ws_bytes, st);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  if (x_dt == FPLX_F32 && y_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3 &&
      is_planar(sn, sd, sh, sw, sc, cin, d, h, w) && is_cl(yn, yd, yh, yw, yc, d, h, w)) {
    int r = fplx_edge_stem_fwd((const float*)x, wp, bias, y, yw, n, d, h, w, cin, cout, stats, st);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  if (x_dt == FPLX_BF16 && y_dt == FPLX_F32 && kd == 1 && kh == 3 && kw == 3 && !stats &&
      is_cl(sn, sd, sh, sw, sc, d, h, w) && is_planar(yn, yd, yh, yw, yc, cout, d, h, w)) {
    int r = fplx_edge_outconv_fwd(x, sw, (const float*)wp, bias, (float*)y, n, d, h, w, cin, cout, st);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  if (x_dt == FPLX_F32 && y_dt == FPLX_BF16 && kd == 1 && kh == 3 && kw == 3 && !stats && !bias &&
      is_planar(sn, sd, sh, sw, sc, cin, d, h, w) && is_cl(yn, yd, yh, yw, yc, d, h, w)) {
    int r = fplx_edge_outconv_dgrad((const float*)x, wp, y, yw, n, d, h, w, cout, cin, st);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  const int64_t V = (int64_t)n * d * h * w;
  const int64_t tiles = (V + FWD_THREADS - 1) / FWD_THREADS;
  // the statistics row count promised by fplx_conv3d_stats_rows() must hold on this path too
  // (the kernel is grid-strided, any grid.x works)
  dim3 grid(fplx_conv3d_stats_rows(n, d, h, w, cin, cout, kd, kh, kw, x_dt, y_dt), (cout + CO_T - 1) / CO_T);
  Strides xs{sn, sd, sh, sw, sc}, ys{yn, yd, yh, yw, yc};
#define LAUNCH(TX, TW, TY)                                                                                      \
  conv_fwd_generic<TX, TW, TY><<<grid, FWD_THREADS, 0, st>>>((const TX*)x, xs, (const TW*)wp, bias, (TY*)y, ys, \
                                                             n, d, h, w, cin, cout, kd, kh, kw, stats, tiles)
  // packed weight type: that of y when y is an activation tensor in dt, fp32 when y is fp32
  if (x_dt == FPLX_F32 && y_dt == FPLX_F32) LAUNCH(float, float, float);
  else if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16) LAUNCH(bf16_t, bf16_t, bf16_t);
  else if (x_dt == FPLX_F32 && y_dt == FPLX_BF16) LAUNCH(float, bf16_t, bf16_t);
  else if (x_dt == FPLX_BF16 && y_dt == FPLX_F32) LAUNCH(bf16_t, float, float);
  else return fplx_fail(FPLX_E_BADDTYPE, "conv3d_fwd: dtypes %d/%d", x_dt, y_dt);
#undef LAUNCH
  return fplx_check_launch("conv3d_fwd");
}

/* ---- Conv2d(3x3) on every depth slice, given as a 27-tap pack that is zero outside the middle depth plane
 * (fplx_pack_conv2d_weight).  Same arguments and results as fplx_conv3d_* with a 3x3x3 kernel; the hint lets the
 * implicit-GEMM kernel skip the 18 dead taps. */
int fplx_conv2d_stats_rows(int n, int d, int h, int w, int cin, int cout, int x_dt, int y_dt) {
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16) {
    int r = fplx_mfma_conv3d_mid_stats_rows(n, d, h, w, cin, cout);
    if (r > 0) return r;
  }
  return fplx_conv3d_stats_rows(n, d, h, w, cin, cout, 3, 3, 3, x_dt, y_dt);
}

size_t fplx_conv2d_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout, int x_dt, int y_dt) {
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16) return fplx_mfma_conv3d_mid_fwd_ws_bytes(n, d, h, w, cin, cout);
  return 0;
}

int fplx_conv2d_fwd(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc, const void* wp,
                    const float* bias, void* y, int y_dt, int64_t yn, int64_t yd, int64_t yh, int64_t yw, int64_t yc,
                    int n, int d, int h, int w, int cin, int cout, float* stats, void* ws, size_t ws_bytes,
                    fplx_stream_t stream) {
  FPLX_REQUIRE(x && wp && y, FPLX_E_NULL, "conv2d_fwd: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, FPLX_E_BADSHAPE, "conv2d_fwd: bad shape");
  if (x_dt == FPLX_BF16 && y_dt == FPLX_BF16 && sc == 1 && yc == 1 && sh == sw * w && sd == sh * h && sn == sd * d &&
      yh == yw * w && yd == yh * h && yn == yd * d) {
    int r = fplx_mfma_conv3d_mid_fwd(x, sw, wp, bias, y, yw, n, d, h, w, cin, cout, stats, ws, ws_bytes, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  // not on the MFMA path (dtype / layout / alignment): all 27 taps through fplx_conv3d_fwd - the same result; the
  // statistics rows promised by fplx_conv2d_stats_rows must then be the 3x3x3 path's
  FPLX_REQUIRE(!stats || fplx_conv2d_stats_rows(n, d, h, w, cin, cout, x_dt, y_dt) ==
                             fplx_conv3d_stats_rows(n, d, h, w, cin, cout, 3, 3, 3, x_dt, y_dt),
               FPLX_E_BADSHAPE, "conv2d_fwd: operands are not laid out for the MFMA path the statistics rows were sized for");
  return fplx_conv3d_fwd(x, x_dt, sn, sd, sh, sw, sc, wp, bias, y, y_dt, yn, yd, yh, yw, yc, n, d, h, w, cin, cout, 3, 3, 3,
                         stats, ws, ws_bytes, stream);
}

size_t fplx_conv3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw) {
  const int64_t V = (int64_t)n * d * h * w;
  const size_t a = (size_t)wgrad_chunks(V) * kd * kh * kw * cout * cin * sizeof(float);
  const size_t b = (size_t)fplx_rows_for(V) * cout * sizeof(float);
  size_t m = 0;
  if (kd == 3 && kh == 3 && kw == 3) {
    m = fplx_mfma_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout);
    const size_t e = fplx_edge_stem_wgrad_ws_bytes(n, d, h, w, cin, cout);
    if (e > m) m = e;
  }
  if (kd == 1 && kh == 3 && kw == 3) m = fplx_edge_outconv_wgrad_ws_bytes(n, d, h, w, cin, cout);
  const size_t g = a + b + 256;
  return (m + b + 256) > g ? (m + b + 256) : g;
}

/* ---- the channel concatenation of two tensors as a convolution operand, never materialised ---- */
int fplx_conv3d_cat2_ok(int n, int d, int h, int w, int cin, int cout) {
  // two halves of 32 channels: the depth-marching Cin = 64 kernel streams them as its two half-slabs, the data
  // gradient is the Cin = 32 march with two output tensors, the weight gradient the two-ci-tile kernel
  return cin == 64 && cout == 32 && fplx_march_ok(n, d, h, w, 64, cout) && fplx_march_ok(n, d, h, w, cout, 64) &&
         fplx_mfma_conv3d_wgrad_cit(n, d, h, w, 64, cout) == 2;
}

static int fwd_cat2_impl(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias, void* y,
                         int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats, int mid,
                         fplx_stream_t stream) {
  FPLX_REQUIRE(x0 && x1 && wp && y, FPLX_E_NULL, "conv3d_fwd_cat2: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && fplx_conv3d_cat2_ok(n, d, h, w, cin, cout), FPLX_E_BADSHAPE,
               "conv3d_fwd_cat2: shape n=%d d=%d h=%d w=%d cin=%d cout=%d not supported (fplx_conv3d_cat2_ok)", n, d,
               h, w, cin, cout);
  const int r = fplx_march_conv3d_fwd(x0, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, (hipStream_t)stream, x1,
                                      nullptr, mid);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "conv3d_fwd_cat2: pointers / leading dimensions not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}

static int dgrad_split2_impl(const void* dy, int64_t ldy, const void* wb, void* dx0, void* dx1, int64_t ldx, int n,
                             int d, int h, int w, int cin, int cout, int mid, fplx_stream_t stream) {
  FPLX_REQUIRE(dy && wb && dx0 && dx1, FPLX_E_NULL, "conv3d_dgrad_split2: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && fplx_conv3d_cat2_ok(n, d, h, w, cin, cout), FPLX_E_BADSHAPE,
               "conv3d_dgrad_split2: shape not supported (fplx_conv3d_cat2_ok)");
  // the data gradient is the convolution of dy (cout channels) with the mirrored pack, producing cin channels
  const int r = fplx_march_conv3d_fwd(dy, ldy, wb, nullptr, dx0, ldx, n, d, h, w, cout, cin, nullptr,
                                      (hipStream_t)stream, nullptr, dx1, mid);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "conv3d_dgrad_split2: pointers / leading dimensions not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}

/* ---- inference: conv + (eval-mode BatchNorm folded into pack and bias by the caller) + PReLU in one kernel ---- */
int fplx_conv3d_fwd_act_ok(int n, int d, int h, int w, int cin, int cout, int mid, int cat2) {
  if (n <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
  if (cat2)                                     // the Cin = 64 march on two half-slabs; the brick kernel's two-tensor form
    return fplx_conv3d_cat2_ok(n, d, h, w, cin, cout) || fplx_mfma_conv3d_act_cat2_ok(n, d, h, w, cin, cout, mid ? 1 : 0);
  return fplx_mfma_conv3d_act_ok(n, d, h, w, cin, cout, mid ? 1 : 0);
}

int fplx_conv3d_fwd_act(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias, const float* prelu_slope,
                        void* y, int64_t ldy, int n, int d, int h, int w, int cin, int cout, int mid, int n_x0, void* ws,
                        size_t ws_bytes, fplx_stream_t stream) {
  FPLX_REQUIRE(x0 && wp && bias && prelu_slope && y, FPLX_E_NULL, "conv3d_fwd_act: null pointer");
  FPLX_REQUIRE(n_x0 == 0 || (x1 && n_x0 > 0 && n % n_x0 == 0), FPLX_E_BADSHAPE,
               "conv3d_fwd_act: n_x0 = %d needs the two-tensor form and must divide n = %d", n_x0, n);
  FPLX_REQUIRE(fplx_conv3d_fwd_act_ok(n, d, h, w, cin, cout, mid, x1 != nullptr), FPLX_E_BADSHAPE,
               "conv3d_fwd_act: no fused kernel for n=%d d=%d h=%d w=%d cin=%d cout=%d (fplx_conv3d_fwd_act_ok)", n, d, h, w, cin, cout);
  int r;
  if (x1 && fplx_conv3d_cat2_ok(n, d, h, w, cin, cout))
    r = fplx_march_conv3d_fwd_act(x0, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, nullptr, (hipStream_t)stream, x1, nullptr,
                                  mid ? 1 : 0, prelu_slope, n_x0 == n ? 0 : n_x0);
  else if (x1)
    r = fplx_mfma_conv3d_fwd_act_cat2(x0, x1, ldx, wp, bias, prelu_slope, y, ldy, n, d, h, w, cin, cout, n_x0 == n ? 0 : n_x0,
                                      (hipStream_t)stream);
  else r = fplx_mfma_conv3d_fwd_act(x0, ldx, wp, bias, prelu_slope, y, ldy, n, d, h, w, cin, cout, ws, ws_bytes, mid ? 1 : 0,
                                    (hipStream_t)stream);
  if (r == 0)
    return fplx_fail(FPLX_E_BADSHAPE, "conv3d_fwd_act: the kernel's launcher declined (pointers / leading dimensions not 16-byte "
                     "aligned, or a sample of 1 GiB and more: d h w ldx 2 >= 2^30)");
  return r < 0 ? r : FPLX_OK;
}

int fplx_conv3d_fwd_cat2(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias, void* y,
                         int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats,
                         fplx_stream_t stream) {
  return fwd_cat2_impl(x0, x1, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, 0, stream);
}
int fplx_conv3d_dgrad_split2(const void* dy, int64_t ldy, const void* wb, void* dx0, void* dx1, int64_t ldx, int n,
                             int d, int h, int w, int cin, int cout, fplx_stream_t stream) {
  return dgrad_split2_impl(dy, ldy, wb, dx0, dx1, ldx, n, d, h, w, cin, cout, 0, stream);
}
/* the same two operations for a pack that is a Conv2d in the middle depth plane (fplx_conv2d_*) */
int fplx_conv2d_fwd_cat2(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias, void* y,
                         int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats,
                         fplx_stream_t stream) {
  return fwd_cat2_impl(x0, x1, ldx, wp, bias, y, ldy, n, d, h, w, cin, cout, stats, 1, stream);
}
int fplx_conv2d_dgrad_split2(const void* dy, int64_t ldy, const void* wb, void* dx0, void* dx1, int64_t ldx, int n,
                             int d, int h, int w, int cin, int cout, fplx_stream_t stream) {
  return dgrad_split2_impl(dy, ldy, wb, dx0, dx1, ldx, n, d, h, w, cin, cout, 1, stream);
}

static int wgrad_cat2_impl(const void* x0, const void* x1, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n,
                           int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, int mid, fplx_stream_t stream) {
  FPLX_REQUIRE(x0 && x1 && dy && dw && ws, FPLX_E_NULL, "conv3d_wgrad_cat2: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && fplx_conv3d_cat2_ok(n, d, h, w, cin, cout), FPLX_E_BADSHAPE,
               "conv3d_wgrad_cat2: shape not supported (fplx_conv3d_cat2_ok)");
  const size_t m = fplx_mfma_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout);
  FPLX_REQUIRE(ws_bytes >= m, FPLX_E_WORKSPACE, "conv3d_wgrad_cat2: workspace %zu < %zu", ws_bytes, m);
  const int r = fplx_mfma_conv3d_wgrad(x0, ldx, dy, ldy, dw, n, d, h, w, cin, cout, ws, m, (hipStream_t)stream, x1, mid);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "conv3d_wgrad_cat2: pointers / leading dimensions not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}
int fplx_conv3d_wgrad_cat2(const void* x0, const void* x1, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n,
                           int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  return wgrad_cat2_impl(x0, x1, ldx, dy, ldy, dw, n, d, h, w, cin, cout, ws, ws_bytes, 0, stream);
}
/* dw fp32 [Cout][Cin][3][3]: the weight gradient of a Conv2d per depth slice (only the middle-plane taps are computed) */
int fplx_conv2d_wgrad_cat2(const void* x0, const void* x1, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n,
                           int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  return wgrad_cat2_impl(x0, x1, ldx, dy, ldy, dw, n, d, h, w, cin, cout, ws, ws_bytes, 1, stream);
}

/* ---- out_conv (C0 -> classes, 1x3x3) fused with the BatchNorm + PReLU passes of the convolution site in front of it ---- */
static bool outconv_bn_ok(int n, int d, int h, int w, int c0, int ncls) {
  return n > 0 && d > 0 && h > 0 && w > 0 && fplx_edge_outconv_bn_ok(n, d, h, w, c0, ncls);
}
int fplx_outconv_bn_rows(int n, int d, int h, int w, int c0, int ncls) {
  return outconv_bn_ok(n, d, h, w, c0, ncls) ? fplx_edge_outconv_bn_rows(n, d, h, w, ncls) : 0;
}
int fplx_outconv_fwd_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* prelu_slope, void* a,
                        int64_t lda, const float* wf, const float* bias, float* logits, int n, int d, int h, int w, int c0,
                        int ncls, fplx_stream_t stream) {
  FPLX_REQUIRE(y && scale && shift && prelu_slope && wf && logits, FPLX_E_NULL, "outconv_fwd_bn: null pointer");
  FPLX_REQUIRE(a || fplx_edge_outconv_wgrad_bn_ws_bytes(n, d, h, w, c0, ncls) > 0, FPLX_E_NULL,
               "outconv_fwd_bn: a = NULL only where fplx_outconv_wgrad_bn_ws_bytes(...) > 0 (the weight gradient then needs no stored activation)");
  FPLX_REQUIRE(outconv_bn_ok(n, d, h, w, c0, ncls), FPLX_E_BADSHAPE, "outconv_fwd_bn: c0 = %d, classes = %d not supported (fplx_outconv_bn_rows == 0)", c0, ncls);
  const int r = fplx_edge_outconv_fwd_bn(y, ldy, scale, shift, prelu_slope, a, lda, wf, bias, logits, n, d, h, w, c0, ncls, (hipStream_t)stream);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "outconv_fwd_bn: pointers / leading dimensions not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}
int fplx_outconv_dgrad_bn_reduce(const float* dlogits, const void* wb, const void* y, int64_t ldy, const float* mean,
                                 const float* rstd, const float* scale, const float* shift, const float* prelu_slope,
                                 float* part, int n, int d, int h, int w, int c0, int ncls, fplx_stream_t stream) {
  FPLX_REQUIRE(dlogits && wb && y && mean && rstd && scale && shift && prelu_slope && part, FPLX_E_NULL, "outconv_dgrad_bn_reduce: null pointer");
  FPLX_REQUIRE(outconv_bn_ok(n, d, h, w, c0, ncls), FPLX_E_BADSHAPE, "outconv_dgrad_bn_reduce: shape not supported (fplx_outconv_bn_rows == 0)");
  const int r = fplx_edge_outconv_dgrad_bn(1, dlogits, wb, y, ldy, mean, rstd, scale, shift, prelu_slope, nullptr, part, nullptr, 0,
                                           n, d, h, w, c0, ncls, (hipStream_t)stream);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "outconv_dgrad_bn_reduce: pointers / leading dimensions not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}
int fplx_outconv_dgrad_bn_apply(const float* dlogits, const void* wb, const void* y, int64_t ldy, const float* mean,
                                const float* rstd, const float* scale, const float* shift, const float* prelu_slope,
                                const float* coef, void* dy, int64_t lddy, int n, int d, int h, int w, int c0, int ncls,
                                fplx_stream_t stream) {
  FPLX_REQUIRE(dlogits && wb && y && mean && rstd && scale && shift && prelu_slope && coef && dy, FPLX_E_NULL, "outconv_dgrad_bn_apply: null pointer");
  FPLX_REQUIRE(outconv_bn_ok(n, d, h, w, c0, ncls), FPLX_E_BADSHAPE, "outconv_dgrad_bn_apply: shape not supported (fplx_outconv_bn_rows == 0)");
  const int r = fplx_edge_outconv_dgrad_bn(2, dlogits, wb, y, ldy, mean, rstd, scale, shift, prelu_slope, coef, nullptr, dy, lddy,
                                           n, d, h, w, c0, ncls, (hipStream_t)stream);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "outconv_dgrad_bn_apply: pointers / leading dimensions not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}

size_t fplx_outconv_wgrad_bn_ws_bytes(int n, int d, int h, int w, int c0, int ncls) {
  return outconv_bn_ok(n, d, h, w, c0, ncls) ? fplx_edge_outconv_wgrad_bn_ws_bytes(n, d, h, w, c0, ncls) : 0;
}
int fplx_outconv_wgrad_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* prelu_slope,
                          const float* dlogits, float* dw, float* db, int n, int d, int h, int w, int c0, int ncls, void* ws,
                          size_t ws_bytes, fplx_stream_t stream) {
  FPLX_REQUIRE(y && scale && shift && prelu_slope && dlogits && dw && ws, FPLX_E_NULL, "outconv_wgrad_bn: null pointer");
  const size_t need = fplx_outconv_wgrad_bn_ws_bytes(n, d, h, w, c0, ncls);
  FPLX_REQUIRE(need > 0, FPLX_E_BADSHAPE, "outconv_wgrad_bn: c0 = %d, classes = %d not supported (fplx_outconv_wgrad_bn_ws_bytes == 0)", c0, ncls);
  FPLX_REQUIRE(ws_bytes >= need, FPLX_E_WORKSPACE, "outconv_wgrad_bn: workspace %zu < %zu bytes", ws_bytes, need);
  const int r = fplx_edge_outconv_wgrad_bn(y, ldy, scale, shift, prelu_slope, dlogits, dw, db, n, d, h, w, c0, ncls, ws, (hipStream_t)stream);
  if (r == 0) return fplx_fail(FPLX_E_BADSHAPE, "outconv_wgrad_bn: y / its leading dimension not 16-byte aligned");
  return r < 0 ? r : FPLX_OK;
}

/* the stem's weight gradient with the site's BatchNorm + PReLU backward apply formed on the way in (include/fplx.h) */
int fplx_stem_wgrad_bn(const float* x, const void* y, int64_t ldy, const void* dout, int64_t ldd, const float* mean,
                       const float* rstd, const float* scale, const float* shift, const float* slope, const float* coef,
                       float* dw, int n, int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  FPLX_REQUIRE(x && y && dout && mean && rstd && scale && shift && slope && coef && dw && ws, FPLX_E_NULL, "stem_wgrad_bn: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0, FPLX_E_BADSHAPE, "stem_wgrad_bn: bad shape");
  const size_t need = fplx_edge_stem_wgrad_ws_bytes(n, d, h, w, cin, cout);
  FPLX_REQUIRE(need > 0, FPLX_E_BADSHAPE, "stem_wgrad_bn: %d -> %d channels is not a stem the MFMA kernel takes (in_chns 1 | 4, C0 %% 32 == 0)", cin, cout);
  FPLX_REQUIRE(ws_bytes >= need, FPLX_E_WORKSPACE, "stem_wgrad_bn: workspace %zu < %zu (fplx_conv3d_wgrad_ws_bytes)", ws_bytes, need);
  FPLX_REQUIRE(ldy >= cout && ldd >= cout && ldy % 8 == 0 && ldd % 8 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)dout % 16 == 0,
               FPLX_E_BADSHAPE, "stem_wgrad_bn: y and dout must be 16-byte aligned with leading dimensions %% 8 == 0");
  const int r = fplx_edge_stem_wgrad_bn(x, dout, ldd, dw, n, d, h, w, cin, cout, ws, (hipStream_t)stream, y, ldy, mean, rstd, scale,
                                        shift, slope, coef);
  if (r < 0) return r;
  FPLX_REQUIRE(r == 1, FPLX_E_BADSHAPE, "stem_wgrad_bn: operands refused");
  return FPLX_OK;
}

int fplx_conv3d_wgrad(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc,
                      const void* dy, int dy_dt, int64_t yn, int64_t yd, int64_t yh, int64_t yw, int64_t yc,
                      float* dw, float* db, int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw,
                      void* ws, size_t ws_bytes, fplx_stream_t stream) {
  FPLX_REQUIRE(x && dy && dw && ws, FPLX_E_NULL, "conv3d_wgrad: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, FPLX_E_BADSHAPE, "conv3d_wgrad: bad shape");
  FPLX_REQUIRE(ws_bytes >= fplx_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, kd, kh, kw), FPLX_E_WORKSPACE,
               "conv3d_wgrad: workspace %zu < %zu", ws_bytes,
               fplx_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, kd, kh, kw));
  hipStream_t st = (hipStream_t)stream;
  const int64_t V = (int64_t)n * d * h * w;
  const int taps = kd * kh * kw, chunks = wgrad_chunks(V);
  Strides xs{sn, sd, sh, sw, sc}, ys{yn, yd, yh, yw, yc};
  bool done = false;
  size_t used = (size_t)chunks * taps * cout * cin * sizeof(float);
  if (x_dt == FPLX_BF16 && dy_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3 && sc == 1 && yc == 1 &&
      sh == sw * w && sd == sh * h && sn == sd * d && yh == yw * w && yd == yh * h && yn == yd * d) {
    const size_t m = fplx_mfma_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout);
    if (m > 0) {
      int r = fplx_mfma_conv3d_wgrad(x, sw, dy, yw, dw, n, d, h, w, cin, cout, ws, m, st, nullptr, 0);
      if (r < 0) return r;
      if (r == 1) { done = true; used = m; }
    }
  }
  if (!done && x_dt == FPLX_F32 && dy_dt == FPLX_BF16 && kd == 3 && kh == 3 && kw == 3 &&
      is_planar(sn, sd, sh, sw, sc, cin, d, h, w) && is_cl(yn, yd, yh, yw, yc, d, h, w)) {
    const size_t m = fplx_edge_stem_wgrad_ws_bytes(n, d, h, w, cin, cout);
    if (m > 0) {
      int r = fplx_edge_stem_wgrad((const float*)x, dy, yw, dw, n, d, h, w, cin, cout, ws, st);
      if (r < 0) return r;
      if (r == 1) { done = true; used = m; }
    }
  }
  if (!done && x_dt == FPLX_BF16 && dy_dt == FPLX_F32 && kd == 1 && kh == 3 && kw == 3 &&
      is_cl(sn, sd, sh, sw, sc, d, h, w) && is_planar(yn, yd, yh, yw, yc, cout, d, h, w)) {
    const size_t m = fplx_edge_outconv_wgrad_ws_bytes(n, d, h, w, cin, cout);
    if (m > 0) {
      int r = fplx_edge_outconv_wgrad(x, sw, (const float*)dy, dw, n, d, h, w, cin, cout, ws, st);
      if (r < 0) return r;
      if (r == 1) { done = true; used = m; }
    }
  }
  float* part = (float*)ws;
  float* bpart = (float*)((char*)ws + used);
  dim3 grid(chunks, ((cout + WG_T - 1) / WG_T) * ((cin + WG_T - 1) / WG_T), taps);
  if (!done) {
#define LAUNCH(TX, TY)                                                                                         \
  wgrad_generic<TX, TY, false><<<grid, WG_THREADS, 0, st>>>((const TX*)x, xs, (const TY*)dy, ys, part, n, d, h, w, \
                                                            cin, cout, kd, kh, kw, chunks)
  if (x_dt == FPLX_F32 && dy_dt == FPLX_F32) LAUNCH(float, float);
  else if (x_dt == FPLX_BF16 && dy_dt == FPLX_BF16) LAUNCH(bf16_t, bf16_t);
  else if (x_dt == FPLX_F32 && dy_dt == FPLX_BF16) LAUNCH(float, bf16_t);
  else if (x_dt == FPLX_BF16 && dy_dt == FPLX_F32) LAUNCH(bf16_t, float);
  else return fplx_fail(FPLX_E_BADDTYPE, "conv3d_wgrad: dtypes %d/%d", x_dt, dy_dt);
#undef LAUNCH
  wgrad_reduce<<<grid_for((int64_t)taps * cout * cin, 256, 1024), 256, 0, st>>>(part, dw, chunks, taps, cout, cin, 0);
  }
  if (db) {
    if (dy_dt == FPLX_F32) launch_bias_grad<float>((const float*)dy, ys, n, d, h, w, cout, bpart, db, st);
    else launch_bias_grad<bf16_t>((const bf16_t*)dy, ys, n, d, h, w, cout, bpart, db, st);
  }
  return fplx_check_launch("conv3d_wgrad");
}

/* ---- weight gradient of a Conv2d(3x3) per depth slice: dw fp32 [Cout][Cin][3][3], db fp32 [Cout] or NULL.
 * bf16 NDHWC operands: the MFMA stream kernel in its middle-plane mode (9 of 27 taps).  Anything else: the 3x3x3 path
 * into a 27-tap scratch at the end of the workspace, then its middle plane. */
size_t fplx_conv2d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  return fplx_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, 3, 3, 3) + (size_t)27 * cout * cin * sizeof(float) + 256;
}

int fplx_conv2d_wgrad(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc, const void* dy,
                      int dy_dt, int64_t yn, int64_t yd, int64_t yh, int64_t yw, int64_t yc, float* dw, float* db, int n,
                      int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  FPLX_REQUIRE(x && dy && dw && ws, FPLX_E_NULL, "conv2d_wgrad: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, FPLX_E_BADSHAPE, "conv2d_wgrad: bad shape");
  FPLX_REQUIRE(ws_bytes >= fplx_conv2d_wgrad_ws_bytes(n, d, h, w, cin, cout), FPLX_E_WORKSPACE,
               "conv2d_wgrad: workspace %zu < %zu", ws_bytes, fplx_conv2d_wgrad_ws_bytes(n, d, h, w, cin, cout));
  hipStream_t st = (hipStream_t)stream;
  const size_t w3 = fplx_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, 3, 3, 3);
  if (x_dt == FPLX_BF16 && dy_dt == FPLX_BF16 && sc == 1 && yc == 1 && sh == sw * w && sd == sh * h && sn == sd * d &&
      yh == yw * w && yd == yh * h && yn == yd * d) {
    const size_t m = fplx_mfma_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout);
    if (m > 0) {
      int r = fplx_mfma_conv3d_wgrad(x, sw, dy, yw, dw, n, d, h, w, cin, cout, ws, m, st, nullptr, 1);
      if (r < 0) return r;
      if (r == 1) {
        if (db) {
          Strides ys{yn, yd, yh, yw, yc};
          launch_bias_grad<bf16_t>((const bf16_t*)dy, ys, n, d, h, w, cout, (float*)((char*)ws + m), db, st);
        }
        return fplx_check_launch("conv2d_wgrad");
      }
    }
  }
  float* dw27 = (float*)((char*)ws + ((w3 + 255) / 256) * 256);
  int rc = fplx_conv3d_wgrad(x, x_dt, sn, sd, sh, sw, sc, dy, dy_dt, yn, yd, yh, yw, yc, dw27, db, n, d, h, w, cin, cout, 3,
                             3, 3, ws, w3, stream);
  if (rc != FPLX_OK) return rc;
  return conv2d_wgrad_extract(dw27, dw, cout, cin, stream);
}

// sd = 2: ConvTranspose3d(k=2,s=2); sd = 1: ConvTranspose2d(k=2,s=2) on every depth slice (2.5D levels)
static int deconv_fwd_impl(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy, int n, int d,
                           int h, int w, int cin, int cout, int dt, int sd, fplx_stream_t stream) {
  FPLX_REQUIRE(x && wf && bias && y, FPLX_E_NULL, "deconv2_fwd: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && ldx >= cin && ldy >= cout, FPLX_E_BADSHAPE,
               "deconv2_fwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_BF16) {
    int r = fplx_mfma_deconv2_fwd(x, ldx, wf, bias, y, ldy, n, d, h, w, cin, cout, sd, st);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  const int64_t V = (int64_t)n * d * h * w;
  dim3 grid((unsigned)((V + FWD_THREADS - 1) / FWD_THREADS), (cout + CO_T - 1) / CO_T, 4 * sd);
  if (dt == FPLX_F32)
    deconv_fwd_generic<float><<<grid, FWD_THREADS, 0, st>>>((const float*)x, ldx, (const float*)wf, bias, (float*)y,
                                                            ldy, n, d, h, w, cin, cout, sd);
  else if (dt == FPLX_BF16)
    deconv_fwd_generic<bf16_t><<<grid, FWD_THREADS, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)wf, bias,
                                                             (bf16_t*)y, ldy, n, d, h, w, cin, cout, sd);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "deconv2_fwd: dtype %d", dt);
  return fplx_check_launch("deconv2_fwd");
}

static int deconv_dgrad_impl(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx, int n, int d, int h, int w,
                             int cin, int cout, int dt, int sd, fplx_stream_t stream) {
  FPLX_REQUIRE(dy && wb && dx, FPLX_E_NULL, "deconv2_dgrad: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && ldx >= cin && ldy >= cout, FPLX_E_BADSHAPE,
               "deconv2_dgrad: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_BF16) {
    int r = fplx_mfma_deconv2_dgrad(dy, ldy, wb, dx, ldx, n, d, h, w, cin, cout, sd, st);
    if (r != 0) return r < 0 ? r : FPLX_OK;
  }
  const int64_t V = (int64_t)n * d * h * w;
  dim3 grid((unsigned)((V + FWD_THREADS - 1) / FWD_THREADS), (cin + CO_T - 1) / CO_T);
  if (dt == FPLX_F32)
    deconv_dgrad_generic<float><<<grid, FWD_THREADS, 0, st>>>((const float*)dy, ldy, (const float*)wb, (float*)dx, ldx,
                                                              n, d, h, w, cin, cout, sd);
  else if (dt == FPLX_BF16)
    deconv_dgrad_generic<bf16_t><<<grid, FWD_THREADS, 0, st>>>((const bf16_t*)dy, ldy, (const bf16_t*)wb, (bf16_t*)dx,
                                                               ldx, n, d, h, w, cin, cout, sd);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "deconv2_dgrad: dtype %d", dt);
  return fplx_check_launch("deconv2_dgrad");
}

static size_t deconv_wgrad_ws_impl(int n, int d, int h, int w, int cin, int cout, int sd) {
  const int64_t V = (int64_t)n * d * h * w;
  size_t a = (size_t)wgrad_chunks(V) * 4 * sd * cout * cin * sizeof(float);
  const size_t m = fplx_mfma_deconv2_wgrad_ws_bytes(n, d, h, w, cin, cout);     // 8-tap partials for either sd
  if (m > a) a = m;
  return a + (size_t)fplx_rows_for(V * 4 * sd) * cout * sizeof(float) + 256;
}

static int deconv_wgrad_impl(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db, int n, int d,
                             int h, int w, int cin, int cout, int dt, void* ws, size_t ws_bytes, int sd,
                             fplx_stream_t stream) {
  FPLX_REQUIRE(x && dy && dw && ws, FPLX_E_NULL, "deconv2_wgrad: null pointer");
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, FPLX_E_BADSHAPE, "deconv2_wgrad: bad shape");
  FPLX_REQUIRE(ws_bytes >= deconv_wgrad_ws_impl(n, d, h, w, cin, cout, sd), FPLX_E_WORKSPACE,
               "deconv2_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t V = (int64_t)n * d * h * w;
  const int chunks = wgrad_chunks(V);
  const int taps = 4 * sd;
  float* part = (float*)ws;
  size_t used = (size_t)chunks * taps * cout * cin * sizeof(float);
  dim3 grid(chunks, ((cout + WG_T - 1) / WG_T) * ((cin + WG_T - 1) / WG_T), taps);
  Strides xs{(int64_t)d * h * w * ldx, (int64_t)h * w * ldx, (int64_t)w * ldx, ldx, 1};
  // dy is [N, sd D, 2H, 2W]; the kernel addresses depth (2 d + kd): for sd = 1 (kd = 0) half the slice stride does it
  Strides ys{(int64_t)4 * sd * d * h * w * ldy, (int64_t)(sd == 2 ? 4 : 2) * h * w * ldy, (int64_t)2 * w * ldy, ldy, 1};
  bool done = false;
  if (dt == FPLX_BF16) {
    const size_t m = fplx_mfma_deconv2_wgrad_ws_bytes(n, d, h, w, cin, cout);
    if (m > 0) {
      int r = fplx_mfma_deconv2_wgrad(x, ldx, dy, ldy, dw, db, n, d, h, w, cin, cout, ws, m, sd, st);
      if (r < 0) return r;
      if (r == 1) { done = true; used = m; db = nullptr; }        // bias gradient came out of the same pass
    }
  }
  float* bpart = (float*)((char*)ws + used);
  if (done) {
  } else if (dt == FPLX_F32)
    wgrad_generic<float, float, true><<<grid, WG_THREADS, 0, st>>>((const float*)x, xs, (const float*)dy, ys, part, n,
                                                                   d, h, w, cin, cout, sd, 2, 2, chunks);
  else if (dt == FPLX_BF16)
    wgrad_generic<bf16_t, bf16_t, true><<<grid, WG_THREADS, 0, st>>>((const bf16_t*)x, xs, (const bf16_t*)dy, ys, part,
                                                                     n, d, h, w, cin, cout, sd, 2, 2, chunks);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "deconv2_wgrad: dtype %d", dt);
  if (!done) wgrad_reduce<<<grid_for((int64_t)taps * cout * cin, 256, 1024), 256, 0, st>>>(part, dw, chunks, taps, cout, cin, 1);
  if (db) {
    Strides yb{(int64_t)4 * sd * d * h * w * ldy, (int64_t)4 * h * w * ldy, (int64_t)2 * w * ldy, ldy, 1};   // true strides
    if (dt == FPLX_F32) launch_bias_grad<float>((const float*)dy, yb, n, sd * d, 2 * h, 2 * w, cout, bpart, db, st);
    else launch_bias_grad<bf16_t>((const bf16_t*)dy, yb, n, sd * d, 2 * h, 2 * w, cout, bpart, db, st);
  }
  return fplx_check_launch("deconv2_wgrad");
}

int fplx_deconv2_fwd(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy, int n, int d,
                     int h, int w, int cin, int cout, int dt, fplx_stream_t stream) {
  return deconv_fwd_impl(x, ldx, wf, bias, y, ldy, n, d, h, w, cin, cout, dt, 2, stream);
}
int fplx_deconv2_dgrad(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx, int n, int d, int h, int w,
                       int cin, int cout, int dt, fplx_stream_t stream) {
  return deconv_dgrad_impl(dy, ldy, wb, dx, ldx, n, d, h, w, cin, cout, dt, 2, stream);
}
size_t fplx_deconv2_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  return deconv_wgrad_ws_impl(n, d, h, w, cin, cout, 2);
}
int fplx_deconv2_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db, int n, int d,
                       int h, int w, int cin, int cout, int dt, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  return deconv_wgrad_impl(x, ldx, dy, ldy, dw, db, n, d, h, w, cin, cout, dt, ws, ws_bytes, 2, stream);
}
int fplx_deconv122_fwd(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy, int n, int d,
                       int h, int w, int cin, int cout, int dt, fplx_stream_t stream) {
  return deconv_fwd_impl(x, ldx, wf, bias, y, ldy, n, d, h, w, cin, cout, dt, 1, stream);
}
int fplx_deconv122_dgrad(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx, int n, int d, int h, int w,
                         int cin, int cout, int dt, fplx_stream_t stream) {
  return deconv_dgrad_impl(dy, ldy, wb, dx, ldx, n, d, h, w, cin, cout, dt, 1, stream);
}
size_t fplx_deconv122_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout) {
  return deconv_wgrad_ws_impl(n, d, h, w, cin, cout, 1);
}
int fplx_deconv122_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db, int n, int d,
                         int h, int w, int cin, int cout, int dt, void* ws, size_t ws_bytes, fplx_stream_t stream) {
  return deconv_wgrad_impl(x, ldx, dy, ldy, dw, db, n, d, h, w, cin, cout, dt, ws, ws_bytes, 1, stream);
}

}  // extern "C"
