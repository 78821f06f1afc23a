// Segmentation loss (softmax + Dice / weighted CE / image-weighted Dice / entropy term + the
// hard-Dice train metric, one pass), its backward, and the pseudo-label uncertainty filter.
// All tensors here are fp32 planar [N][C][V]: one coalesced stream per class plane.
#include "common.h"

namespace {

constexpr int MAXC = 8;
constexpr int LT = 256;

inline int loss_rows(int64_t v) {
  int64_t r = (v + 4095) / 4096;
  if (r > 512) r = 512;
  if (r < 1) r = 1;
  return (int)r;
}

template <int C>
__device__ __forceinline__ int softmax_argmax(const float (&l)[MAXC], float (&p)[MAXC], bool do_softmax) {
  // scipy.special.softmax / torch.softmax order of operations: max, exp(x - max), sum, divide.
  // Returns argmax (first maximum) of the PROBABILITIES, as np.argmax(prob) does.
  float m = l[0];
#pragma unroll
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  if (do_softmax) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = expf(l[c] - m); s += p[c]; }
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = p[c] / s;
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] = l[c];
  }
  int a = 0;
  float best = p[0];
#pragma unroll
  for (int c = 1; c < C; ++c)
    if (p[c] > best) { best = p[c]; a = c; }
  return a;
}

// part[n][row][6C+3]: per class (Yw, Pw, Iw, Yh, Ph, Ih), then ce numerator, weight sum, entropy sum
template <int C>
__global__ void __launch_bounds__(LT)
seg_loss_fwd_k(const float* __restrict__ logits, const float* __restrict__ label, const float* __restrict__ pw,
               int64_t V, int do_softmax, float* __restrict__ part) {
  constexpr int K = 6 * C + 3;
  const int n = blockIdx.y;
  const float* lg = logits + (int64_t)n * C * V;
  const float* lb = label + (int64_t)n * C * V;
  const float* wp = pw ? pw + (int64_t)n * V : nullptr;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * LT + threadIdx.x; v < V; v += (int64_t)gridDim.x * LT) {
    float l[MAXC], p[MAXC], y[MAXC];
#pragma unroll
    for (int c = 0; c < C; ++c) { l[c] = lg[(int64_t)c * V + v]; y[c] = lb[(int64_t)c * V + v]; }
    const float w = wp ? wp[v] : 1.f;
    // the train metric takes argmax of the raw network output (agent_seg.py:472)
    int am = 0;
    {
      float best = l[0];
#pragma unroll
      for (int c = 1; c < C; ++c)
        if (l[c] > best) { best = l[c]; am = c; }
    }
    softmax_argmax<C>(l, p, do_softmax != 0);
    float ce = 0.f, ent = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      acc[6 * c + 0] += y[c] * w;
      acc[6 * c + 1] += p[c] * w;
      acc[6 * c + 2] += y[c] * p[c] * w;
      const float hc = (am == c) ? 1.f : 0.f;
      acc[6 * c + 3] += y[c];
      acc[6 * c + 4] += hc;
      acc[6 * c + 5] += y[c] * hc;
      ce -= y[c] * logf(p[c] * 0.999f + 5e-4f);
      ent -= p[c] * log2f(p[c] + 1e-10f);
    }
    acc[6 * C + 0] += w * ce;
    acc[6 * C + 1] += w;
    acc[6 * C + 2] += ent;
  }
  __shared__ float red[LT / 64][K];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float t = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    float t = 0.f;
    for (int i = 0; i < LT / 64; ++i) t += red[i][threadIdx.x];
    part[((int64_t)n * gridDim.x + blockIdx.x) * K + threadIdx.x] = t;
  }
}

// one block: reduce the partial rows (double) to per-sample sums [N][K] and their total over the local samples [K]
__global__ void seg_loss_sums_k(const float* __restrict__ part, int rows, int N, int C, double* __restrict__ sums,
                                double* __restrict__ totals) {
  const int K = 6 * C + 3;
  // one wave per (n, k) sum: lanes stride over rows
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = wv; i < N * K; i += nw) {
    const int n = i / K, k = i % K;
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += (double)part[((int64_t)n * rows + r) * K + k];
    s = wave_sum_d(s);
    if (lane == 0) sums[i] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    double t = 0.0;
    for (int n = 0; n < N; ++n) t += sums[n * K + threadIdx.x];      // fixed order
    totals[threadIdx.x] = t;
  }
}

// one thread: the loss terms and the backward coefficient table from the per-sample sums of the LOCAL samples and the
// totals over the WHOLE batch (= the local totals in one process; all-reduced over the ranks under data parallelism, where
// the reference's nn.DataParallel gathers the logits and evaluates ONE loss over the full batch, agent_seg.py:692-698)
__global__ void seg_loss_coef_k(const double* __restrict__ sums, const double* __restrict__ tot, int N, int NG, int C,
                                double V, int has_pw, const float* __restrict__ image_weight, float w_dice, float w_ce,
                                float w_img, float w_ent, float* __restrict__ out, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int K = 6 * C + 3;
  double Ld = 0.0, Limg = 0.0, Lce = 0.0, Lent = 0.0;
  for (int i = 0; i < N * C * 2; ++i) coef[i] = 0.f;
  // global Dice over all voxels of the batch (dice.py:20-57)
  for (int c = 0; c < C; ++c) {
    const double* t = tot + 6 * c;
    const double den = t[0] + t[1] + 1e-5, num = 2.0 * t[2] + 1e-5;
    Ld += num / den;
    out[4 + c] = (float)((2.0 * t[5] + 1e-5) / (t[3] + t[4] + 1e-5));
    for (int n = 0; n < N; ++n) {
      coef[(n * C + c) * 2 + 0] += (float)(w_dice * (-2.0 / (C * den)));
      coef[(n * C + c) * 2 + 1] += (float)(w_dice * (num / (C * den * den)));
    }
  }
  Ld = 1.0 - Ld / C;
  // per-sample Dice times image weight, mean over the batch (dice.py:106-128): this rank's samples only; the terms of the
  // other ranks' samples are theirs (the value is completed by the caller's all-reduce when it wants the number)
  if (w_img != 0.f && image_weight) {
    for (int n = 0; n < N; ++n) {
      double dn = 0.0;
      const double f = (double)image_weight[n] / NG;
      for (int c = 0; c < C; ++c) {
        const double* s = sums + n * K + 6 * c;
        const double den = s[0] + s[1] + 1e-5, num = 2.0 * s[2] + 1e-5;
        dn += num / den;
        coef[(n * C + c) * 2 + 0] += (float)(w_img * f * (-2.0 / (C * den)));
        coef[(n * C + c) * 2 + 1] += (float)(w_img * f * (num / (C * den * den)));
      }
      Limg += f * (1.0 - dn / C);
    }
  }
  const double cenum = tot[6 * C + 0], wsum = tot[6 * C + 1], ent = tot[6 * C + 2];
  const double ce_norm = has_pw ? 1.0 / (wsum + 1e-5) : 1.0 / (NG * V);   // ce.py:39-43
  Lce = cenum * ce_norm;
  Lent = ent / (NG * V);                                                  // agent_seg.py:352-353
  coef[N * C * 2 + 0] = (float)(w_ce * ce_norm);
  coef[N * C * 2 + 1] = (float)(w_ent / (NG * V));
  out[0] = (float)(w_dice * Ld + w_img * Limg + w_ce * Lce + w_ent * Lent);
  out[1] = (float)(w_dice * Ld + w_img * Limg);
  out[2] = (float)Lce;
  out[3] = (float)Lent;
}

template <int C>
__global__ void __launch_bounds__(LT)
seg_loss_bwd_k(const float* __restrict__ logits, const float* __restrict__ label, const float* __restrict__ pw,
               const float* __restrict__ coef, const float* __restrict__ gscale, int N, int64_t V, int do_softmax,
               int use_dice, int use_ce, int use_ent, float* __restrict__ dlogits) {
  const int n = blockIdx.y;
  const float* lg = logits + (int64_t)n * C * V;
  const float* lb = label + (int64_t)n * C * V;
  const float* wp = pw ? pw + (int64_t)n * V : nullptr;
  float* dl = dlogits + (int64_t)n * C * V;
  float A[MAXC], B[MAXC];
#pragma unroll
  for (int c = 0; c < C; ++c) { A[c] = coef[(n * C + c) * 2]; B[c] = coef[(n * C + c) * 2 + 1]; }
  const float cce = coef[N * C * 2], cent = coef[N * C * 2 + 1], gs = *gscale;
  const float inv_ln2 = 1.4426950408889634f;
  for (int64_t v = (int64_t)blockIdx.x * LT + threadIdx.x; v < V; v += (int64_t)gridDim.x * LT) {
    float l[MAXC], p[MAXC], g[MAXC];
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = lg[(int64_t)c * V + v];
    const float w = wp ? wp[v] : 1.f;
    softmax_argmax<C>(l, p, do_softmax != 0);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float y = lb[(int64_t)c * V + v];
      float gc = 0.f;
      if (use_dice) gc += w * fmaf(A[c], y, B[c]);
      if (use_ce) gc -= cce * w * y * 0.999f / (p[c] * 0.999f + 5e-4f);
      if (use_ent) gc -= cent * (log2f(p[c] + 1e-10f) + p[c] * inv_ln2 / (p[c] + 1e-10f));
      g[c] = gc;
      dot = fmaf(gc, p[c], dot);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dl[(int64_t)c * V + v] = gs * (do_softmax ? p[c] * (g[c] - dot) : g[c]);
  }
}

// ------------------------------------------------------------------------------------------
// MC / TTA uncertainty filter (agent_seg.py:911-931) for one volume
constexpr int MAXT = 16;
constexpr uint32_t FPL_CUT_LO = 0x3ACA8577u, FPL_CUT_HI = 0x3F7D6D40u;   // m1 in [0.00154511526, 0.989948273]

template <int C>
__global__ void __launch_bounds__(LT)
mc_filter_k(const float* __restrict__ logits, int T, int64_t V, float thr, uint8_t* __restrict__ hards,
            float* __restrict__ mean_out, float* __restrict__ unc_out, double* __restrict__ part) {
  const bool exact_cut = thr == 0.01f;
  double var_acc = 0.0;
  long long bnd = 0;
  for (int64_t v = (int64_t)blockIdx.x * LT + threadIdx.x; v < V; v += (int64_t)gridDim.x * LT) {
    float sum_c[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum_c[c] = 0.f;
    for (int t = 0; t < T; ++t) {
      float l[MAXC], q[MAXC];
#pragma unroll
      for (int c = 0; c < C; ++c) l[c] = logits[((int64_t)t * C + c) * V + v];
      const int a = softmax_argmax<C>(l, q, true);
      if (hards) hards[(int64_t)t * V + v] = (uint8_t)a;
#pragma unroll
      for (int c = 0; c < C; ++c) sum_c[c] += q[c];   // sequential fp32 sum over T (np.add.reduce, axis 0)
    }
    const float invT = (float)T;
    float mc[C], s2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { mc[c] = sum_c[c] / invT; s2[c] = 0.f; }
    // second sweep (the T logits of this voxel are L1/L2 resident): squared deviations, np.var ddof 0
    for (int t = 0; t < T; ++t) {
      float l[MAXC], q[MAXC];
#pragma unroll
      for (int c = 0; c < C; ++c) l[c] = logits[((int64_t)t * C + c) * V + v];
      softmax_argmax<C>(l, q, true);
#pragma unroll
      for (int c = 0; c < C; ++c) { const float d = q[c] - mc[c]; s2[c] += d * d; }
    }
    float vsum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) vsum += s2[c] / invT;
    var_acc += (double)vsum;
    const float m1 = sum_c[C > 1 ? 1 : 0] / invT;                          // means = mean_T maps[:,1]
    const float u = -1.0f * (m1 * logf(m1 + 1e-6f));
    if (mean_out) mean_out[v] = m1;
    if (unc_out) unc_out[v] = u;
    // `u > 0.01` (the reference's hard-coded threshold) is decided on m1 itself: numpy's float32 u(m) crosses 0.01
    // exactly twice on [0, 1] (tools/filter_cutpoints.py: bisection + exhaustive check of 2^17 patterns around each
    // crossing), so the count does not depend on this device's logf.  Any other threshold uses u.
    const uint32_t mb = __float_as_uint(m1);
    const bool over = exact_cut ? (mb >= FPL_CUT_LO && mb <= FPL_CUT_HI) : (u > thr);
    bnd += over ? 1 : 0;
  }
  __shared__ double red[LT / 64][2];
  const double a = wave_sum_d(var_acc), b = wave_sum_d((double)bnd);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0, s1 = 0;
    for (int i = 0; i < LT / 64; ++i) { s0 += red[i][0]; s1 += red[i][1]; }
    part[(int64_t)blockIdx.x * 2 + 0] = s0;
    part[(int64_t)blockIdx.x * 2 + 1] = s1;
  }
}

// The same filter, four consecutive voxels per thread: 16-byte loads of the logits, one 4-byte store of the four hard labels
// per pass (mc_filter_k's byte stores and its one-voxel-at-a-time dependent loads kept it at 0.7 TB/s).  Per voxel the
// arithmetic - and so every mask, mean and uncertainty - is mc_filter_k's; V % 4 == 0 and 16-byte-aligned pointers (host).
template <int C>
__global__ void __launch_bounds__(LT)
mc_filter_v4_k(const float* __restrict__ logits, int T, int64_t V, float thr, uint8_t* __restrict__ hards,
               float* __restrict__ mean_out, float* __restrict__ unc_out, double* __restrict__ part) {
  const bool exact_cut = thr == 0.01f;
  double var_acc = 0.0;
  long long bnd = 0;
  for (int64_t v = ((int64_t)blockIdx.x * LT + threadIdx.x) * 4; v < V; v += (int64_t)gridDim.x * LT * 4) {
    float sum_c[4][C];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < C; ++c) sum_c[j][c] = 0.f;
    constexpr int TB = 3;                                 // passes loaded together (their loads in flight before the first exp)
    for (int t0 = 0; t0 < T; t0 += TB) {
      float4 lv[TB][C];
#pragma unroll
      for (int k = 0; k < TB; ++k) {
        const int t = t0 + k < T ? t0 + k : T - 1;
#pragma unroll
        for (int c = 0; c < C; ++c) lv[k][c] = *reinterpret_cast<const float4*>(logits + ((int64_t)t * C + c) * V + v);
      }
#pragma unroll
      for (int k = 0; k < TB; ++k) {
        if (t0 + k >= T) break;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float l[MAXC], q[MAXC];
#pragma unroll
          for (int c = 0; c < C; ++c) l[c] = j == 0 ? lv[k][c].x : j == 1 ? lv[k][c].y : j == 2 ? lv[k][c].z : lv[k][c].w;
          const int a = softmax_argmax<C>(l, q, true);
          packed |= (uint32_t)a << (8 * j);
#pragma unroll
          for (int c = 0; c < C; ++c) sum_c[j][c] += q[c];
        }
        if (hards) *reinterpret_cast<uint32_t*>(hards + (int64_t)(t0 + k) * V + v) = packed;
      }
    }
    const float invT = (float)T;
    float mc[4][C], s2[4][C];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < C; ++c) { mc[j][c] = sum_c[j][c] / invT; s2[j][c] = 0.f; }
    for (int t0 = 0; t0 < T; t0 += TB) {                 // second sweep: L2 resident
      float4 lv[TB][C];
#pragma unroll
      for (int k = 0; k < TB; ++k) {
        const int t = t0 + k < T ? t0 + k : T - 1;
#pragma unroll
        for (int c = 0; c < C; ++c) lv[k][c] = *reinterpret_cast<const float4*>(logits + ((int64_t)t * C + c) * V + v);
      }
#pragma unroll
      for (int k = 0; k < TB; ++k) {
        if (t0 + k >= T) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float l[MAXC], q[MAXC];
#pragma unroll
          for (int c = 0; c < C; ++c) l[c] = j == 0 ? lv[k][c].x : j == 1 ? lv[k][c].y : j == 2 ? lv[k][c].z : lv[k][c].w;
          softmax_argmax<C>(l, q, true);
#pragma unroll
          for (int c = 0; c < C; ++c) { const float d = q[c] - mc[j][c]; s2[j][c] += d * d; }
        }
      }
    }
    float m4[4], u4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vsum = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) vsum += s2[j][c] / invT;
      var_acc += (double)vsum;
      const float m1 = sum_c[j][C > 1 ? 1 : 0] / invT;
      const float u = -1.0f * (m1 * logf(m1 + 1e-6f));
      m4[j] = m1; u4[j] = u;
      const uint32_t mb = __float_as_uint(m1);
      const bool over = exact_cut ? (mb >= FPL_CUT_LO && mb <= FPL_CUT_HI) : (u > thr);
      bnd += over ? 1 : 0;
    }
    if (mean_out) *reinterpret_cast<float4*>(mean_out + v) = make_float4(m4[0], m4[1], m4[2], m4[3]);
    if (unc_out) *reinterpret_cast<float4*>(unc_out + v) = make_float4(u4[0], u4[1], u4[2], u4[3]);
  }
  __shared__ double red[LT / 64][2];
  const double a = wave_sum_d(var_acc), b = wave_sum_d((double)bnd);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0, s1 = 0;
    for (int i = 0; i < LT / 64; ++i) { s0 += red[i][0]; s1 += red[i][1]; }
    part[(int64_t)blockIdx.x * 2 + 0] = s0;
    part[(int64_t)blockIdx.x * 2 + 1] = s1;
  }
}

__global__ void mc_filter_finalize_k(const double* __restrict__ part, int rows, double* __restrict__ out) {
  // one wave: lane l adds rows l, l + 64, ... in order, then the 64 lane sums are added in lane order (fixed order; a single
  // thread walking the <= 512 rows took 40-50 us of dependent loads - a third of the whole filter)
  __shared__ double red[64][2];
  double v = 0, b = 0;
  for (int r = threadIdx.x; r < rows; r += 64) { v += part[r * 2]; b += part[r * 2 + 1]; }
  red[threadIdx.x][0] = v; red[threadIdx.x][1] = b;
  __syncthreads();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  v = 0; b = 0;
  for (int l = 0; l < 64; ++l) { v += red[l][0]; b += red[l][1]; }
  out[0] = v;
  out[1] = b;
  // `vars` is a float32 scalar in the reference (maps is float32), boundary an int64
  out[2] = (b < 50.0) ? 1.0 : (double)(float)v / b;
  out[3] = 0.0;
}

template <int C>
__global__ void __launch_bounds__(LT)
hard_label_k(const float* __restrict__ logits, int64_t V, uint8_t* __restrict__ out) {
  const int n = blockIdx.y;
  for (int64_t v = (int64_t)blockIdx.x * LT + threadIdx.x; v < V; v += (int64_t)gridDim.x * LT) {
    float l[MAXC], p[MAXC];
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = logits[((int64_t)n * C + c) * V + v];
    out[(int64_t)n * V + v] = (uint8_t)softmax_argmax<C>(l, p, true);
  }
}

__global__ void __launch_bounds__(LT)
pixel_weight_k(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int64_t V, int apply, float iw,
               float* __restrict__ out) {
  for (int64_t v = (int64_t)blockIdx.x * LT + threadIdx.x; v < V; v += (int64_t)gridDim.x * LT) {
    // get_pixel_weight.py:21-26: both = min(a+b,1); and = a*b; xor = both - and; w = 1 - 0.5*xor
    const int av = a[v], bv = b[v];
    int both = av + bv;
    both = both > 1 ? 1 : both;
    const int x = both - av * bv;
    float w = 1.0f - 0.5f * (float)x;
    if (apply) {                       // nifty_dataset.py:165-168
      if (w < 1.0f) w = 0.f;
      w = w * iw;
    }
    out[v] = w;
  }
}

inline int grid1(int64_t v, int cap) {
  int64_t g = (v + LT - 1) / LT;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define DISPATCH_C(C, KERNEL, ...)                 \
  switch (C) {                                     \
    case 1: KERNEL<1> __VA_ARGS__; break;          \
    case 2: KERNEL<2> __VA_ARGS__; break;          \
    case 3: KERNEL<3> __VA_ARGS__; break;          \
    case 4: KERNEL<4> __VA_ARGS__; break;          \
    case 5: KERNEL<5> __VA_ARGS__; break;          \
    case 6: KERNEL<6> __VA_ARGS__; break;          \
    case 7: KERNEL<7> __VA_ARGS__; break;          \
    default: KERNEL<8> __VA_ARGS__; break;         \
  }

extern "C" {

// rows to ALLOCATE per sample: the partial rows the kernels use plus 5 spare ones - behind the N x rows x K partials of a call
// the spare N x 5 x K floats hold the per-sample sums and the batch totals as doubles ((N + 1) x K doubles = 2 (N + 1) K floats)
// plus the one float the 8-byte alignment of that region may cost (K is odd): 2 (N + 1) K + 1 <= 5 N K for every N >= 1.
// (With 4 spare rows N = 1 and an odd rows x K overran the buffer by one float: ADVICE r02.)
int fplx_loss_rows(int64_t voxels_per_sample) { return loss_rows(voxels_per_sample) + 5; }

static int seg_loss_check(const char* what, const float* logits, const float* label, int n, int c, int64_t v) {
  FPLX_REQUIRE(logits && label, FPLX_E_NULL, "%s: null pointer", what);
  FPLX_REQUIRE(n > 0 && n <= 64 && c >= 1 && c <= MAXC && v > 0, FPLX_E_BADSHAPE, "%s: n=%d (<=64) c=%d (<=%d) v=%lld", what, n,
               c, MAXC, (long long)v);
  return FPLX_OK;
}

int fplx_seg_loss_sums(const float* logits, const float* label, const float* pixel_weight, int n, int c, int64_t v,
                       int softmax, float* part, double* sums, double* totals, fplx_stream_t stream) {
  const int rc = seg_loss_check("seg_loss_sums", logits, label, n, c, v);
  if (rc != FPLX_OK) return rc;
  FPLX_REQUIRE(part && sums && totals, FPLX_E_NULL, "seg_loss_sums: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int rows = loss_rows(v);
  dim3 grid(rows, n);
  DISPATCH_C(c, seg_loss_fwd_k, <<<grid, LT, 0, st>>>(logits, label, pixel_weight, v, softmax, part));
  seg_loss_sums_k<<<1, 1024, 0, st>>>(part, rows, n, c, sums, totals);
  return fplx_check_launch("seg_loss_sums");
}

int fplx_seg_loss_from_sums(const double* sums, const double* totals, const float* image_weight, int n, int n_global, int c,
                            int64_t v, int has_pixel_weight, float w_dice, float w_ce, float w_dice_img, float w_entropy,
                            float* out, float* coef, fplx_stream_t stream) {
  FPLX_REQUIRE(sums && totals && out && coef, FPLX_E_NULL, "seg_loss_from_sums: null pointer");
  FPLX_REQUIRE(n > 0 && n <= 64 && n_global >= n && c >= 1 && c <= MAXC && v > 0, FPLX_E_BADSHAPE, "seg_loss_from_sums: bad shape");
  FPLX_REQUIRE(w_dice_img == 0.f || (image_weight && has_pixel_weight), FPLX_E_NULL,
               "seg_loss_from_sums: image-weighted Dice needs image_weight and pixel_weight");
  seg_loss_coef_k<<<1, 64, 0, (hipStream_t)stream>>>(sums, totals, n, n_global, c, (double)v, has_pixel_weight, image_weight,
                                                     w_dice, w_ce, w_dice_img, w_entropy, out, coef);
  return fplx_check_launch("seg_loss_from_sums");
}

int fplx_seg_loss_fwd(const float* logits, const float* label, const float* pixel_weight, const float* image_weight,
                      int n, int c, int64_t v, float w_dice, float w_ce, float w_dice_img, float w_entropy, int softmax,
                      float* part, float* out, float* coef, fplx_stream_t stream) {
  FPLX_REQUIRE(part && out && coef, FPLX_E_NULL, "seg_loss_fwd: null pointer");
  FPLX_REQUIRE(w_dice_img == 0.f || (image_weight && pixel_weight), FPLX_E_NULL,
               "seg_loss_fwd: image-weighted Dice needs image_weight and pixel_weight");
  // the per-sample sums live in the spare rows of the caller's `part` buffer (fplx_loss_rows)
  const int rows = loss_rows(v), K = 6 * c + 3;
  int rc = seg_loss_check("seg_loss_fwd", logits, label, n, c, v);
  if (rc != FPLX_OK) return rc;
  double* sums = reinterpret_cast<double*>(part + (((size_t)n * rows * K + 1) / 2) * 2);
  double* totals = sums + (size_t)n * K;
  rc = fplx_seg_loss_sums(logits, label, pixel_weight, n, c, v, softmax, part, sums, totals, stream);
  if (rc != FPLX_OK) return rc;
  return fplx_seg_loss_from_sums(sums, totals, image_weight, n, n, c, v, pixel_weight != nullptr, w_dice, w_ce, w_dice_img,
                                 w_entropy, out, coef, stream);
}

int fplx_seg_loss_bwd(const float* logits, const float* label, const float* pixel_weight, const float* coef,
                      const float* gscale, int n, int c, int64_t v, float w_dice, float w_ce, float w_dice_img,
                      float w_entropy, int softmax, float* dlogits, fplx_stream_t stream) {
  FPLX_REQUIRE(logits && label && coef && gscale && dlogits, FPLX_E_NULL, "seg_loss_bwd: null pointer");
  FPLX_REQUIRE(n > 0 && n <= 64 && c >= 1 && c <= MAXC && v > 0, FPLX_E_BADSHAPE, "seg_loss_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(grid1(v, 2048), n);
  const int ud = (w_dice != 0.f || w_dice_img != 0.f), uc = w_ce != 0.f, ue = w_entropy != 0.f;
  DISPATCH_C(c, seg_loss_bwd_k, <<<grid, LT, 0, st>>>(logits, label, pixel_weight, coef, gscale, n, v, softmax, ud, uc,
                                                      ue, dlogits));
  return fplx_check_launch("seg_loss_bwd");
}

int fplx_mc_filter(const float* logits, int t, int c, int64_t v, float thr, uint8_t* hards, float* mean_out,
                   float* unc_out, double* part, double* out, fplx_stream_t stream) {
  FPLX_REQUIRE(logits && part && out, FPLX_E_NULL, "mc_filter: null pointer");
  FPLX_REQUIRE(t >= 1 && t <= MAXT && c >= 2 && c <= MAXC && v > 0, FPLX_E_BADSHAPE,
               "mc_filter: t=%d (<=%d) c=%d (2..%d)", t, MAXT, c, MAXC);
  hipStream_t st = (hipStream_t)stream;
  const int rows = fplx_rows_for(v);
  const bool v4 = v % 4 == 0 && ((uintptr_t)logits % 16) == 0 && ((uintptr_t)hards % 4) == 0 && ((uintptr_t)mean_out % 16) == 0 &&
                  ((uintptr_t)unc_out % 16) == 0;
  if (v4 && c <= 4) {
    switch (c) {
      case 2: mc_filter_v4_k<2><<<rows, LT, 0, st>>>(logits, t, v, thr, hards, mean_out, unc_out, part); break;
      case 3: mc_filter_v4_k<3><<<rows, LT, 0, st>>>(logits, t, v, thr, hards, mean_out, unc_out, part); break;
      default: mc_filter_v4_k<4><<<rows, LT, 0, st>>>(logits, t, v, thr, hards, mean_out, unc_out, part); break;
    }
    mc_filter_finalize_k<<<1, 64, 0, st>>>(part, rows, out);
    return fplx_check_launch("mc_filter");
  }
  switch (c) {
    case 2: mc_filter_k<2><<<rows, LT, 0, st>>>(logits, t, v, thr, hards, mean_out, unc_out, part); break;
    case 3: mc_filter_k<3><<<rows, LT, 0, st>>>(logits, t, v, thr, hards, mean_out, unc_out, part); break;
    case 4: mc_filter_k<4><<<rows, LT, 0, st>>>(logits, t, v, thr, hards, mean_out, unc_out, part); break;
    default:
      return fplx_fail(FPLX_E_BADSHAPE, "mc_filter: class_num %d not instantiated (2..4)", c);
  }
  mc_filter_finalize_k<<<1, 64, 0, st>>>(part, rows, out);
  return fplx_check_launch("mc_filter");
}

int fplx_hard_label(const float* logits, int n, int c, int64_t v, uint8_t* out, fplx_stream_t stream) {
  FPLX_REQUIRE(logits && out, FPLX_E_NULL, "hard_label: null pointer");
  FPLX_REQUIRE(n > 0 && c >= 1 && c <= MAXC && v > 0, FPLX_E_BADSHAPE, "hard_label: bad shape");
  dim3 grid(grid1(v, 2048), n);
  DISPATCH_C(c, hard_label_k, <<<grid, LT, 0, (hipStream_t)stream>>>(logits, v, out));
  return fplx_check_launch("hard_label");
}

int fplx_pixel_weight(const uint8_t* a, const uint8_t* b, int64_t v, int apply_set_weight, float image_weight,
                      float* out, fplx_stream_t stream) {
  FPLX_REQUIRE(a && b && out, FPLX_E_NULL, "pixel_weight: null pointer");
  FPLX_REQUIRE(v > 0, FPLX_E_BADSHAPE, "pixel_weight: empty volume");
  pixel_weight_k<<<grid1(v, 2048), LT, 0, (hipStream_t)stream>>>(a, b, v, apply_set_weight, image_weight, out);
  return fplx_check_launch("pixel_weight");
}

}  // extern "C"
