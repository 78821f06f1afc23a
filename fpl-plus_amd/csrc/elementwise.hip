// HBM-bound passes of the DSBN U-Net: BatchNorm statistics finalisation, the fused
// BN-apply + PReLU + dropout pass and its three-stage backward, MaxPool3d(2) forward/backward,
// fused Adam.  One 16-byte vector per lane per access (8 bf16 / 4 fp32 channels of one voxel),
// fixed-order two-stage reductions (bitwise reproducible).
#include "common.h"
#include "philox.h"

namespace {

constexpr int EW_THREADS = 256;

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void load_nt(const float* p, float (&v)[4]) { load(p, v); }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  typedef __attribute__((ext_vector_type(8))) __bf16 raw;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const raw r = *reinterpret_cast<const raw*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
  }
  // streaming variant for the last read of a tensor (BN-apply forward, second BN-backward pass): no allocation in
  // L2 / Infinity Cache, measured +1.7 % on the whole step; the pooling kernels lose with it and keep plain loads
  static __device__ __forceinline__ void load_nt(const bf16_t* p, float (&v)[8]) {
    const raw r = __builtin_nontemporal_load(reinterpret_cast<const raw*>(p));
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    raw r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    *reinterpret_cast<raw*>(p) = r;
  }
};

// generic VEC-wide access: VEC == Vec<T>::N uses 16-byte accesses, VEC == 1 scalar
template <typename T, int VEC>
__device__ __forceinline__ void ldv(const T* p, float (&v)[VEC]) {
  if constexpr (VEC == 1) v[0] = Act<T>::ld(p);
  else Vec<T>::load(p, v);
}
template <typename T, int VEC>
__device__ __forceinline__ void ldv_nt(const T* p, float (&v)[VEC]) {
  if constexpr (VEC == 1) v[0] = Act<T>::ld(p);
  else Vec<T>::load_nt(p, v);
}
template <typename T, int VEC>
__device__ __forceinline__ void stv(T* p, const float (&v)[VEC]) {
  if constexpr (VEC == 1) Act<T>::st(p, v[0]);
  else Vec<T>::store(p, v);
}

// dropout keep flags for VEC consecutive elements starting at flat index e (e % VEC == 0 when VEC>1)
template <int VEC>
__device__ __forceinline__ void keep_flags(int64_t e, uint32_t thr, uint32_t k0, uint32_t k1, uint32_t sid,
                                           bool (&keep)[VEC]) {
  if constexpr (VEC == 1) {
    const Philox4 r = philox4x32_10((uint32_t)(e >> 2), 0u, sid, 0u, k0, k1);
    keep[0] = r.v[e & 3] >= thr;
  } else {
#pragma unroll
    for (int g = 0; g < VEC / 4; ++g) {
      const Philox4 r = philox4x32_10((uint32_t)((e >> 2) + g), 0u, sid, 0u, k0, k1);
#pragma unroll
      for (int i = 0; i < 4; ++i) keep[g * 4 + i] = r.v[i] >= thr;
    }
  }
}

struct DropCfg { uint32_t thr, k0, k1, sid; float inv_keep; int on; };

inline DropCfg make_drop(float p, uint64_t seed, uint32_t sid) {
  DropCfg d;
  d.on = p > 0.f;
  d.thr = dropout_threshold(p);
  d.k0 = (uint32_t)seed;
  d.k1 = (uint32_t)(seed >> 32);
  d.sid = sid;
  d.inv_keep = d.on ? (float)(1.0 / (1.0 - (double)p)) : 1.f;
  return d;
}

// ------------------------------------------------------------------------------------------
// one block of four waves per channel: threads stride over the partial rows (two rows each at 512 rows instead of
// eight on one wave - the kernel is a latency chain between a convolution and its BN-apply pass), fixed-order sum
// (per thread, wave butterfly, four wave totals)
constexpr int FIN_THREADS = 256;
__global__ void __launch_bounds__(FIN_THREADS)
bn_train_finalize_k(const float* __restrict__ stats, int rows, int C, double count,
                    const float* __restrict__ gamma, const float* __restrict__ beta,
                    float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt,
                    float momentum, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                    float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int r = threadIdx.x; r < rows; r += FIN_THREADS) {
    s1 += (double)stats[((int64_t)r * 2 + 0) * C + c];
    s2 += (double)stats[((int64_t)r * 2 + 1) * C + c];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  __shared__ double wred[FIN_THREADS / 64][2];
  if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = s1; wred[threadIdx.x >> 6][1] = s2; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  s1 = (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]);
  s2 = (wred[0][1] + wred[1][1]) + (wred[2][1] + wred[3][1]);
  if (c == 0 && nbt) *nbt += 1;
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  const float mf = (float)m;
  mean[c] = mf;
  rstd[c] = rs;
  const float sc = gamma[c] * rs;
  scale[c] = sc;
  shift[c] = beta[c] - mf * sc;
  if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * mf;
  if (rv) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_prepare_k(const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                  float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rs = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * rs;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// ------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_fwd_k(const T* __restrict__ y, int64_t ldy, T* __restrict__ out, int64_t ldo, const float* __restrict__ scale,
             const float* __restrict__ shift, const float* __restrict__ slope_p, DropCfg dc, int64_t voxels, int C) {
  const int G = C / VEC;
  const int64_t total = voxels * G;
  const float slope = *slope_p;
  auto one = [&](int64_t v, int c0, float (&a)[VEC]) {
    bool keep[VEC];
    if (dc.on) keep_flags<VEC>(v * C + c0, dc.thr, dc.k0, dc.k1, dc.sid, keep);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float z = fmaf(a[j], scale[c0 + j], shift[c0 + j]);
      z = z > 0.f ? z : z * slope;
      if (dc.on) z = keep[j] ? z * dc.inv_keep : 0.f;
      a[j] = z;
    }
    stv<T, VEC>(out + v * ldo + c0, a);
  };
  const int64_t st = (int64_t)gridDim.x * EW_THREADS;
  int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  for (; i + 3 * st < total; i += 4 * st) {
    float a[4][VEC];
    int64_t v[4];
    int c0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = (i + u * st) / G;
      c0[u] = (int)((i + u * st) % G) * VEC;
      ldv_nt<T, VEC>(y + v[u] * ldy + c0[u], a[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(v[u], c0[u], a[u]);
  }
  for (; i < total; i += st) {
    const int64_t v = i / G;
    const int c0 = (int)(i % G) * VEC;
    float a[VEC];
    ldv_nt<T, VEC>(y + v * ldy + c0, a);
    one(v, c0, a);
  }
}

// dz for one element given y, dout
__device__ __forceinline__ float dz_of(float yv, float dout, float sc, float sh, float slope, bool on, bool keep,
                                       float inv_keep, float& z_out) {
  const float z = fmaf(yv, sc, sh);
  z_out = z;
  float da = dout;
  if (on) da = keep ? dout * inv_keep : 0.f;
  return z > 0.f ? da : da * slope;
}

// Channel-group-stationary forms of the two apply passes (the vector path when C / VEC divides the block): a thread keeps
// ONE group of VEC channels for its whole life, like bn_act_bwd_reduce_k, so the per-channel constants sit in registers
// and the voxel index advances by a constant - the flat-index forms above re-derive (voxel, channel) with a 64-bit
// division and re-load up to six constants per element, which costs the backward apply pass 15 % of its bandwidth.
template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_fwd_g_k(const T* __restrict__ y, int64_t ldy, T* __restrict__ out, int64_t ldo, const float* __restrict__ scale,
               const float* __restrict__ shift, const float* __restrict__ slope_p, DropCfg dc, int64_t voxels, int C) {
  const int G = C / VEC, VL = EW_THREADS / G;            // host: EW_THREADS % G == 0
  const int g = threadIdx.x % G, vl = threadIdx.x / G, c0 = g * VEC;
  const float slope = *slope_p;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
  auto one = [&](int64_t v, float (&a)[VEC]) {
    bool keep[VEC];
    if (dc.on) keep_flags<VEC>(v * C + c0, dc.thr, dc.k0, dc.k1, dc.sid, keep);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float z = fmaf(a[j], sc[j], sh[j]);
      z = z > 0.f ? z : z * slope;
      if (dc.on) z = keep[j] ? z * dc.inv_keep : 0.f;
      a[j] = z;
    }
    stv<T, VEC>(out + v * ldo + c0, a);
  };
  const int64_t st = (int64_t)gridDim.x * VL;
  int64_t v = (int64_t)blockIdx.x * VL + vl;
  for (; v + 3 * st < voxels; v += 4 * st) {
    float a[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u) ldv_nt<T, VEC>(y + (v + u * st) * ldy + c0, a[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(v + u * st, a[u]);
  }
  for (; v < voxels; v += st) {
    float a[VEC];
    ldv_nt<T, VEC>(y + v * ldy + c0, a);
    one(v, a);
  }
}

template <typename T, int VEC, int U = 4, bool DROP = true>
__global__ void __launch_bounds__(EW_THREADS, (U == 1 ? 7 : 1))
bn_act_bwd_apply_g_k(const T* __restrict__ y, int64_t ldy, const T* __restrict__ dout, int64_t ldd, T* __restrict__ dy,
                     int64_t ldo, const float* __restrict__ mean, const float* __restrict__ rstd,
                     const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ slope_p,
                     const float* __restrict__ coef, DropCfg dc, int64_t voxels, int C) {
  const int G = C / VEC, VL = EW_THREADS / G;
  const int g = threadIdx.x % G, vl = threadIdx.x / G, c0 = g * VEC;
  const float slope = *slope_p;
  float sc[VEC], sh[VEC], m[VEC], rs[VEC], k0[VEC], k1[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j];
    k0[j] = coef[c0 + j]; k1[j] = coef[C + c0 + j];
  }
  auto one = [&](int64_t v, float (&a)[VEC], float (&d)[VEC]) {
    bool keep[VEC];
    const bool on = DROP && dc.on;                       // DROP = false: the host saw p == 0 (no Philox code, fewer registers)
    if (on) keep_flags<VEC>(v * C + c0, dc.thr, dc.k0, dc.k1, dc.sid, keep);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float z;
      const float dz = dz_of(a[j], d[j], sc[j], sh[j], slope, on, on ? keep[j] : true, dc.inv_keep, z);
      const float xh = (a[j] - m[j]) * rs[j];
      d[j] = sc[j] * (dz - k0[j] - xh * k1[j]);
    }
    stv<T, VEC>(dy + v * ldo + c0, d);
  };
  const int64_t st = (int64_t)gridDim.x * VL;
  int64_t v = (int64_t)blockIdx.x * VL + vl;
  for (; v + (U - 1) * st < voxels; v += U * st) {         // U voxels (2 U loads) in flight per lane
    float a[U][VEC], d[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ldv_nt<T, VEC>(y + (v + u * st) * ldy + c0, a[u]);
      ldv_nt<T, VEC>(dout + (v + u * st) * ldd + c0, d[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(v + u * st, a[u], d[u]);
  }
  for (; v < voxels; v += st) {
    float a[VEC], d[VEC];
    ldv_nt<T, VEC>(y + v * ldy + c0, a);
    ldv_nt<T, VEC>(dout + v * ldd + c0, d);
    one(v, a, d);
  }
}

// stage 1 of backward: per-channel sums of dz and dz*xhat, and the slope gradient
template <typename T, int VEC, int U = 4, bool DROP = true>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_bwd_reduce_k(const T* __restrict__ y, int64_t ldy, const T* __restrict__ dout, int64_t ldd,
                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ scale,
                    const float* __restrict__ shift, const float* __restrict__ slope_p, DropCfg dc, int64_t voxels,
                    int C, float* __restrict__ part) {
  const int G = C / VEC;                 // channel groups per voxel (host guarantees G <= EW_THREADS)
  const int VL = EW_THREADS / G;         // voxel lanes per block
  const int g = threadIdx.x % G, vl = threadIdx.x / G;
  const bool active = vl < VL;
  const int c0 = g * VEC;
  const float slope = *slope_p;
  float sdz[VEC], sdx[VEC], sds = 0.f;
  float m[VEC], rs[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sdz[j] = sdx[j] = 0.f;
    m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j];
  }
  auto consume = [&](int64_t v, const float (&a)[VEC], const float (&d)[VEC]) {
    bool keep[VEC];
    const bool on = DROP && dc.on;
    if (on) keep_flags<VEC>(v * C + c0, dc.thr, dc.k0, dc.k1, dc.sid, keep);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float z;
      const bool kp = on ? keep[j] : true;
      const float dz = dz_of(a[j], d[j], sc[j], sh[j], slope, on, kp, dc.inv_keep, z);
      float da = d[j];
      if (on) da = kp ? d[j] * dc.inv_keep : 0.f;
      sds += z > 0.f ? 0.f : da * z;
      sdz[j] += dz;
      sdx[j] = fmaf(dz, (a[j] - m[j]) * rs[j], sdx[j]);
    }
  };
  if (active) {
    // four voxels (8 x 16-byte loads) in flight per lane: the pass is pure HBM streaming
    const int64_t st = (int64_t)gridDim.x * VL;
    int64_t v = (int64_t)blockIdx.x * VL + vl;
    for (; v + (U - 1) * st < voxels; v += U * st) {
      float a[U][VEC], d[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ldv<T, VEC>(y + (v + u * st) * ldy + c0, a[u]);
        ldv<T, VEC>(dout + (v + u * st) * ldd + c0, d[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) consume(v + u * st, a[u], d[u]);
    }
    for (; v < voxels; v += st) {
      float a[VEC], d[VEC];
      ldv<T, VEC>(y + v * ldy + c0, a);
      ldv<T, VEC>(dout + v * ldd + c0, d);
      consume(v, a, d);
    }
  }
  __shared__ float red[EW_THREADS][2 * VEC + 1];
  float* row = part + (int64_t)blockIdx.x * (2 * C + 1);
  if ((G & (G - 1)) == 0 && G <= 64) {
    // lanes of one channel group sit G apart: butterfly inside the wave (fixed order), then 4 wave totals through LDS.
    // (The serial form below costs a block ~1000 dependent LDS reads on 4 threads: 10-20 % of this pass at levels 0-2.)
    for (int o = G; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { sdz[j] += __shfl_xor(sdz[j], o, 64); sdx[j] += __shfl_xor(sdx[j], o, 64); }
      sds += __shfl_xor(sds, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < G) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { red[wave * 64 + lane][j] = sdz[j]; red[wave * 64 + lane][VEC + j] = sdx[j]; }
      red[wave * 64 + lane][2 * VEC] = sds;
    }
    __syncthreads();
    if (threadIdx.x < G) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int wv = 0; wv < EW_THREADS / 64; ++wv) { t0 += red[wv * 64 + threadIdx.x][j]; t1 += red[wv * 64 + threadIdx.x][VEC + j]; }
        row[c0 + j] = t0;
        row[C + c0 + j] = t1;
      }
    }
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int wv = 0; wv < EW_THREADS / 64; ++wv)
        for (int k = 0; k < G; ++k) t += red[wv * 64 + k][2 * VEC];
      row[2 * C] = t;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { red[threadIdx.x][j] = sdz[j]; red[threadIdx.x][VEC + j] = sdx[j]; }
  red[threadIdx.x][2 * VEC] = active ? sds : 0.f;
  __syncthreads();
  if (threadIdx.x < G) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t0 = 0.f, t1 = 0.f;
      for (int k = 0; k < VL; ++k) { t0 += red[k * G + threadIdx.x][j]; t1 += red[k * G + threadIdx.x][VEC + j]; }
      row[c0 + j] = t0;
      row[C + c0 + j] = t1;
    }
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < VL * G; ++k) t += red[k][2 * VEC];
    row[2 * C] = t;
  }
}

// one block of four waves per channel (+ one for the PReLU slope); same summation scheme as bn_train_finalize_k
__global__ void __launch_bounds__(FIN_THREADS)
bn_act_bwd_finalize_k(const float* __restrict__ part, int rows, int C, double count, int train,
                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dslope,
                      float* __restrict__ coef) {
  const int c = blockIdx.x;
  const int stride = 2 * C + 1;
  __shared__ double wred[FIN_THREADS / 64][2];
  if (c < C) {
    double s0 = 0.0, s1 = 0.0;
    for (int r = threadIdx.x; r < rows; r += FIN_THREADS) {
      s0 += (double)part[(int64_t)r * stride + c];
      s1 += (double)part[(int64_t)r * stride + C + c];
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = s0; wred[threadIdx.x >> 6][1] = s1; }
    __syncthreads();
    s0 = (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]);
    s1 = (wred[0][1] + wred[1][1]) + (wred[2][1] + wred[3][1]);
    if (threadIdx.x == 0) {
      if (dbeta) dbeta[c] += (float)s0;
      if (dgamma) dgamma[c] += (float)s1;
      coef[c] = train ? (float)(s0 / count) : 0.f;
      coef[C + c] = train ? (float)(s1 / count) : 0.f;
    }
  } else {
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += FIN_THREADS) s += (double)part[(int64_t)r * stride + 2 * C];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6][0] = s;
    __syncthreads();
    s = (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]);
    if (threadIdx.x == 0 && dslope) dslope[0] += (float)s;
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_bwd_apply_k(const T* __restrict__ y, int64_t ldy, const T* __restrict__ dout, int64_t ldd, T* __restrict__ dy,
                   int64_t ldo, const float* __restrict__ mean, const float* __restrict__ rstd,
                   const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ slope_p,
                   const float* __restrict__ coef, DropCfg dc, int64_t voxels, int C) {
  const int G = C / VEC;
  const int64_t total = voxels * G;
  const float slope = *slope_p;
  auto one = [&](int64_t v, int c0, float (&a)[VEC], float (&d)[VEC]) {
    bool keep[VEC];
    if (dc.on) keep_flags<VEC>(v * C + c0, dc.thr, dc.k0, dc.k1, dc.sid, keep);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      float z;
      const float dz = dz_of(a[j], d[j], scale[c], shift[c], slope, dc.on, dc.on ? keep[j] : true, dc.inv_keep, z);
      const float xh = (a[j] - mean[c]) * rstd[c];
      d[j] = scale[c] * (dz - coef[c] - xh * coef[C + c]);
    }
    stv<T, VEC>(dy + v * ldo + c0, d);
  };
  const int64_t st = (int64_t)gridDim.x * EW_THREADS;
  int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  for (; i + 3 * st < total; i += 4 * st) {                  // four vectors (8 loads) in flight per lane
    float a[4][VEC], d[4][VEC];
    int64_t v[4];
    int c0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = (i + u * st) / G;
      c0[u] = (int)((i + u * st) % G) * VEC;
      ldv_nt<T, VEC>(y + v[u] * ldy + c0[u], a[u]);
      ldv_nt<T, VEC>(dout + v[u] * ldd + c0[u], d[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(v[u], c0[u], a[u], d[u]);
  }
  for (; i < total; i += st) {
    const int64_t v = i / G;
    const int c0 = (int)(i % G) * VEC;
    float a[VEC], d[VEC];
    ldv_nt<T, VEC>(y + v * ldy + c0, a);
    ldv_nt<T, VEC>(dout + v * ldd + c0, d);
    one(v, c0, a, d);
  }
}

// ------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
maxpool2_fwd_k(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int C,
               int pd) {      // pd = 2: MaxPool3d(2); pd = 1: MaxPool2d(2) on every depth slice (2.5D levels)
  const int G = C / VEC, Do = D / pd, Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Do * Ho * Wo * G;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    int c0, wo, ho, d_o;
    int64_t vo, r;
    if (total < ((int64_t)1 << 31)) {          // 32-bit index math: a 64-bit division is a ~100-instruction sequence
      unsigned q = (unsigned)i;
      c0 = (int)(q % (unsigned)G) * VEC; q /= (unsigned)G;
      vo = q;
      wo = (int)(q % (unsigned)Wo); q /= (unsigned)Wo;
      ho = (int)(q % (unsigned)Ho); q /= (unsigned)Ho;
      d_o = (int)(q % (unsigned)Do); q /= (unsigned)Do;
      r = q;
    } else {
      c0 = (int)(i % G) * VEC;
      r = i / G;
      vo = r;
      wo = r % Wo; r /= Wo;
      ho = r % Ho; r /= Ho;
      d_o = r % Do; r /= Do;
    }
    float best[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) best[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
      const int64_t vi = (((int64_t)r * D + pd * d_o + (t >> 2)) * H + 2 * ho + ((t >> 1) & 1)) * W + 2 * wo + (t & 1);
      float a[VEC];
      ldv<T, VEC>(x + vi * ldx + c0, a);
#pragma unroll
      for (int j = 0; j < VEC; ++j) best[j] = a[j] > best[j] ? a[j] : best[j];
    }
    stv<T, VEC>(y + vo * ldy + c0, best);
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
maxpool2_bwd_k(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t ldy, const T* __restrict__ dskip,
               int64_t lds, T* __restrict__ dx, int64_t ldo, int N, int D, int H, int W, int C, int pd) {
  const int G = C / VEC, Do = D / pd, Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Do * Ho * Wo * G;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    int c0, wo, ho, d_o;
    int64_t vo, r;
    if (total < ((int64_t)1 << 31)) {          // 32-bit index math: a 64-bit division is a ~100-instruction sequence
      unsigned q = (unsigned)i;
      c0 = (int)(q % (unsigned)G) * VEC; q /= (unsigned)G;
      vo = q;
      wo = (int)(q % (unsigned)Wo); q /= (unsigned)Wo;
      ho = (int)(q % (unsigned)Ho); q /= (unsigned)Ho;
      d_o = (int)(q % (unsigned)Do); q /= (unsigned)Do;
      r = q;
    } else {
      c0 = (int)(i % G) * VEC;
      r = i / G;
      vo = r;
      wo = r % Wo; r /= Wo;
      ho = r % Ho; r /= Ho;
      d_o = r % Do; r /= Do;
    }
    float g[VEC], best[VEC];
    int arg[VEC];
    ldv<T, VEC>(dy + vo * ldy + c0, g);
    float a[8][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    int64_t vis[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
      vis[t] = (((int64_t)r * D + pd * d_o + (t >> 2)) * H + 2 * ho + ((t >> 1) & 1)) * W + 2 * wo + (t & 1);
      ldv<T, VEC>(x + vis[t] * ldx + c0, a[t]);
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (a[t][j] > best[j]) { best[j] = a[t][j]; arg[j] = t; }   // first maximum wins (ATen max_pool3d)
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
      float o[VEC];
      if (dskip) ldv<T, VEC>(dskip + vis[t] * lds + c0, o);
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += (arg[j] == t) ? g[j] : 0.f;
      stv<T, VEC>(dx + vis[t] * ldo + c0, o);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused forms at the end of a DownBlock (second ConvBlockND site, no dropout there: unet2d5_dsbn.py:79-81 then 117):
//   bn_act_pool_fwd_k  y2 -> a2 = PReLU(BN(y2)) (the skip tensor) AND MaxPool(a2) in one pass: the pooling pass no longer
//                      re-reads the activation it was just handed (6.25 -> 4.25 bytes per element);
//   pool_bwd_bn_reduce_k  the pooling gradient + skip gradient -> d(a2) AND the two per-channel sums of the BatchNorm
//                      backward over that d(a2) in one pass; a2 itself is recomputed from y2 (same arithmetic, same bf16
//                      rounding, same first-maximum rule) instead of read: 10.25 -> 6.25 bytes per element.
// A thread owns one group of VEC channels (constants in registers) and strides over POOLED voxels.
template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_pool_fwd_k(const T* __restrict__ y, int64_t ldy, T* __restrict__ out, int64_t ldo, T* __restrict__ pooled, int64_t ldp,
                  const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ slope_p, int N,
                  int D, int H, int W, int C, int pd) {
  const int G = C / VEC, VL = EW_THREADS / G;            // host: EW_THREADS % G == 0
  const int g = threadIdx.x % G, vl = threadIdx.x / G, c0 = g * VEC;
  const int Do = D / pd, Ho = H / 2, Wo = W / 2;
  const int64_t vout = (int64_t)N * Do * Ho * Wo;
  const float slope = *slope_p;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
  for (int64_t vo = (int64_t)blockIdx.x * VL + vl; vo < vout; vo += (int64_t)gridDim.x * VL) {
    int64_t q = vo;
    const int wo = (int)(q % Wo); q /= Wo;
    const int ho = (int)(q % Ho); q /= Ho;
    const int d_o = (int)(q % Do);
    const int64_t n = q / Do;
    float a[8][VEC], best[VEC];
    int64_t vis[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
      vis[t] = ((n * D + pd * d_o + (t >> 2)) * H + 2 * ho + ((t >> 1) & 1)) * W + 2 * wo + (t & 1);
      ldv_nt<T, VEC>(y + vis[t] * ldy + c0, a[t]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) best[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z = fmaf(a[t][j], sc[j], sh[j]);
        z = z > 0.f ? z : z * slope;
        z = (float)(T)z;                                 // what the skip tensor stores and the pooling compares
        a[t][j] = z;
        best[j] = z > best[j] ? z : best[j];
      }
      stv<T, VEC>(out + vis[t] * ldo + c0, a[t]);
    }
    stv<T, VEC>(pooled + vo * ldp + c0, best);
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
pool_bwd_bn_reduce_k(const T* __restrict__ y, int64_t ldy, const T* __restrict__ dy, int64_t lddy, const T* __restrict__ dskip,
                     int64_t lds, T* __restrict__ dx, int64_t ldo, const float* __restrict__ mean,
                     const float* __restrict__ rstd, const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ slope_p, int N, int D, int H, int W, int C, int pd, float* __restrict__ part) {
  const int G = C / VEC, VL = EW_THREADS / G;            // host: G a power of two <= 64 dividing EW_THREADS
  const int g = threadIdx.x % G, vl = threadIdx.x / G, c0 = g * VEC;
  const int Do = D / pd, Ho = H / 2, Wo = W / 2;
  const int64_t vout = (int64_t)N * Do * Ho * Wo;
  const float slope = *slope_p;
  float sdz[VEC], sdx[VEC], sds = 0.f, m[VEC], rs[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sdz[j] = sdx[j] = 0.f;
    m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j];
  }
  for (int64_t vo = (int64_t)blockIdx.x * VL + vl; vo < vout; vo += (int64_t)gridDim.x * VL) {
    int64_t q = vo;
    const int wo = (int)(q % Wo); q /= Wo;
    const int ho = (int)(q % Ho); q /= Ho;
    const int d_o = (int)(q % Do);
    const int64_t n = q / Do;
    float gr[VEC], best[VEC], a[8][VEC], o[8][VEC];
    int arg[VEC];
    int64_t vis[8];
    ldv<T, VEC>(dy + vo * lddy + c0, gr);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
      vis[t] = ((n * D + pd * d_o + (t >> 2)) * H + 2 * ho + ((t >> 1) & 1)) * W + 2 * wo + (t & 1);
      ldv<T, VEC>(y + vis[t] * ldy + c0, a[t]);
      if (dskip) ldv<T, VEC>(dskip + vis[t] * lds + c0, o[t]);
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[t][j] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; arg[j] = 0; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z = fmaf(a[t][j], sc[j], sh[j]);
        z = z > 0.f ? z : z * slope;
        z = (float)(T)z;                                 // the stored activation the forward pooling compared
        if (z > best[j]) { best[j] = z; arg[j] = t; }    // first maximum wins (ATen max_pool3d)
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= 4 * pd) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float dv = (float)(T)(o[t][j] + ((arg[j] == t) ? gr[j] : 0.f));     // d(a2) as stored; the sums use the stored value
        o[t][j] = dv;
        const float z = fmaf(a[t][j], sc[j], sh[j]);
        const float dz = z > 0.f ? dv : dv * slope;
        sds += z > 0.f ? 0.f : dv * z;
        sdz[j] += dz;
        sdx[j] = fmaf(dz, (a[t][j] - m[j]) * rs[j], sdx[j]);
      }
      stv<T, VEC>(dx + vis[t] * ldo + c0, o[t]);
    }
  }
  // per-block partial row, as bn_act_bwd_reduce_k writes it (butterfly over the lanes of a channel group, 4 waves via LDS)
  __shared__ float red[EW_THREADS / 64 * 64][2 * VEC + 1];
  float* row = part + (int64_t)blockIdx.x * (2 * C + 1);
  for (int ofs = G; ofs < 64; ofs <<= 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sdz[j] += __shfl_xor(sdz[j], ofs, 64); sdx[j] += __shfl_xor(sdx[j], ofs, 64); }
    sds += __shfl_xor(sds, ofs, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < G) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[wave * 64 + lane][j] = sdz[j]; red[wave * 64 + lane][VEC + j] = sdx[j]; }
    red[wave * 64 + lane][2 * VEC] = sds;
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int wv = 0; wv < EW_THREADS / 64; ++wv) { t0 += red[wv * 64 + threadIdx.x][j]; t1 += red[wv * 64 + threadIdx.x][VEC + j]; }
      row[c0 + j] = t0;
      row[C + c0 + j] = t1;
    }
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int wv = 0; wv < EW_THREADS / 64; ++wv)
      for (int k = 0; k < G; ++k) t += red[wv * 64 + k][2 * VEC];
    row[2 * C] = t;
  }
}

// Round 3: the same two passes with the lanes laid along the INPUT row.  Above, a thread owns a pooled voxel and reads its 8
// window voxels - consecutive lanes read 64-byte pieces 128 bytes apart (every wave load touches twice the lines it uses, 17
// loads and 200 registers per thread): 1.6 TB/s at level 0 where the plain BatchNorm reduction reads at 5.  Here a thread
// owns one input COLUMN of a window - voxel w of row pair (h, h + 1) of the pd depths, VEC channels - so that a wave load is
// one contiguous kilobyte; the two columns of a window are the lanes l and l ^ G (consecutive w, W even), which exchange
// their candidate maxima by a lane shuffle.  First-maximum rule: window index t = 4 dd + 2 hh + ww, a lane scans its own
// positions in ascending t, the pair keeps the larger value and, on ties, the smaller t - the same winner as the scan over
// all eight.  Same arithmetic and rounding points per element as the kernels above: dx / a2 / pooled are bit-identical,
// the partial rows differ only in how the voxels are dealt to them.
template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
bn_act_pool_fwd_col_k(const T* __restrict__ y, int64_t ldy, T* __restrict__ out, int64_t ldo, T* __restrict__ pooled,
                      int64_t ldp, const float* __restrict__ scale, const float* __restrict__ shift,
                      const float* __restrict__ slope_p, int N, int D, int H, int W, int C, int pd) {
  const int G = C / VEC, VL = EW_THREADS / G;            // host: G a power of two <= 32
  const int g = threadIdx.x % G, vl = threadIdx.x / G, c0 = g * VEC;
  const int Do = D / pd, Ho = H / 2;
  const unsigned units = (unsigned)N * Do * Ho * W;     // (n, d_o, ho, w): host keeps this below 2^31
  const float slope = *slope_p;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
  // the loop bound is rounded up to whole lane PAIRS being active together (units is even, VL is even)
  for (unsigned u = blockIdx.x * VL + vl; u < units; u += gridDim.x * VL) {
    unsigned q = u;
    const int w = (int)(q % (unsigned)W); q /= (unsigned)W;
    const int ho = (int)(q % (unsigned)Ho); q /= (unsigned)Ho;
    const int d_o = (int)(q % (unsigned)Do);
    const int64_t n = q / (unsigned)Do;
    float a[4][VEC], best[VEC];
    int64_t vis[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 * pd) break;
      vis[t] = ((n * D + pd * d_o + (t >> 1)) * H + 2 * ho + (t & 1)) * W + w;
      ldv_nt<T, VEC>(y + vis[t] * ldy + c0, a[t]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) best[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 * pd) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z = fmaf(a[t][j], sc[j], sh[j]);
        z = z > 0.f ? z : z * slope;
        z = (float)(T)z;                                 // what the skip tensor stores and the pooling compares
        a[t][j] = z;
        best[j] = z > best[j] ? z : best[j];
      }
      stv<T, VEC>(out + vis[t] * ldo + c0, a[t]);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float o = __shfl_xor(best[j], G, 64);
      best[j] = o > best[j] ? o : best[j];
    }
    if (!(w & 1)) stv<T, VEC>(pooled + (int64_t)(u >> 1) * ldp + c0, best);     // u / 2 = ((n Do + d_o) Ho + ho) Wo + w / 2
  }
}

template <typename T, int VEC>
__global__ void __launch_bounds__(EW_THREADS)
pool_bwd_bn_reduce_col_k(const T* __restrict__ y, int64_t ldy, const T* __restrict__ dy, int64_t lddy,
                         const T* __restrict__ dskip, int64_t lds, T* __restrict__ dx, int64_t ldo,
                         const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ scale,
                         const float* __restrict__ shift, const float* __restrict__ slope_p, int N, int D, int H, int W, int C,
                         int pd, float* __restrict__ part) {
  const int G = C / VEC, VL = EW_THREADS / G;            // host: G a power of two <= 32
  const int g = threadIdx.x % G, vl = threadIdx.x / G, c0 = g * VEC;
  const int Do = D / pd, Ho = H / 2;
  const unsigned units = (unsigned)N * Do * Ho * W;
  const float slope = *slope_p;
  float sdz[VEC], sdx[VEC], sds = 0.f, m[VEC], rs[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sdz[j] = sdx[j] = 0.f;
    m[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j];
  }
  for (unsigned u = blockIdx.x * VL + vl; u < units; u += gridDim.x * VL) {
    unsigned q = u;
    const int w = (int)(q % (unsigned)W); q /= (unsigned)W;
    const int ho = (int)(q % (unsigned)Ho); q /= (unsigned)Ho;
    const int d_o = (int)(q % (unsigned)Do);
    const int64_t n = q / (unsigned)Do;
    float gr[VEC], best[VEC], a[4][VEC], o[4][VEC];
    int arg[VEC];
    int64_t vis[4];
    ldv<T, VEC>(dy + (int64_t)(u >> 1) * lddy + c0, gr);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 * pd) break;
      vis[t] = ((n * D + pd * d_o + (t >> 1)) * H + 2 * ho + (t & 1)) * W + w;
      ldv<T, VEC>(y + vis[t] * ldy + c0, a[t]);
      if (dskip) ldv<T, VEC>(dskip + vis[t] * lds + c0, o[t]);
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[t][j] = 0.f;
      }
    }
    // own candidates in ascending window index t8 = 4 dd + 2 hh + ww (ww = w & 1)
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; arg[j] = 8; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 * pd) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z = fmaf(a[t][j], sc[j], sh[j]);
        z = z > 0.f ? z : z * slope;
        z = (float)(T)z;                                 // the stored activation the forward pooling compared
        if (z > best[j]) { best[j] = z; arg[j] = 2 * t + (w & 1); }
      }
    }
    // the pair's winner: larger value, on ties the smaller window index (= the first maximum of the scan over all eight)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float ob = __shfl_xor(best[j], G, 64);
      const int oa = __shfl_xor(arg[j], G, 64);
      if (ob > best[j] || (ob == best[j] && oa < arg[j])) arg[j] = -1;       // the other column holds the maximum
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= 2 * pd) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float dv = (float)(T)(o[t][j] + ((arg[j] == 2 * t + (w & 1)) ? gr[j] : 0.f));     // d(a2) as stored
        o[t][j] = dv;
        const float z = fmaf(a[t][j], sc[j], sh[j]);
        const float dz = z > 0.f ? dv : dv * slope;
        sds += z > 0.f ? 0.f : dv * z;
        sdz[j] += dz;
        sdx[j] = fmaf(dz, (a[t][j] - m[j]) * rs[j], sdx[j]);
      }
      stv<T, VEC>(dx + vis[t] * ldo + c0, o[t]);
    }
  }
  // per-block partial row, as bn_act_bwd_reduce_k writes it (butterfly over the lanes of a channel group, 4 waves via LDS)
  __shared__ float red[EW_THREADS / 64 * 64][2 * VEC + 1];
  float* row = part + (int64_t)blockIdx.x * (2 * C + 1);
  for (int ofs = G; ofs < 64; ofs <<= 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sdz[j] += __shfl_xor(sdz[j], ofs, 64); sdx[j] += __shfl_xor(sdx[j], ofs, 64); }
    sds += __shfl_xor(sds, ofs, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < G) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[wave * 64 + lane][j] = sdz[j]; red[wave * 64 + lane][VEC + j] = sdx[j]; }
    red[wave * 64 + lane][2 * VEC] = sds;
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int wv = 0; wv < EW_THREADS / 64; ++wv) { t0 += red[wv * 64 + threadIdx.x][j]; t1 += red[wv * 64 + threadIdx.x][VEC + j]; }
      row[c0 + j] = t0;
      row[C + c0 + j] = t1;
    }
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int wv = 0; wv < EW_THREADS / 64; ++wv)
      for (int k = 0; k < G; ++k) t += red[wv * 64 + k][2 * VEC];
    row[2 * C] = t;
  }
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EW_THREADS)
adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
       float step_size, float b1, float b2, float eps, float wd, float inv_sqrt_bc2, float gscale) {
  const FplxAdamConst c = {step_size, b1, b2, eps, wd, inv_sqrt_bc2, gscale, 1.f - b1, 1.f - b2};
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * EW_THREADS) {
    float pi = p[i], mi = m[i], vi = v[i];
    fplx_adam_elem(pi, g[i], mi, vi, c);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}


// per-channel sum / sum of squares of an NDHWC tensor: part [rows][2][C] (standalone DSBN layer)
template <typename T>
__global__ void __launch_bounds__(EW_THREADS)
channel_stats_k(const T* __restrict__ x, int64_t ld, int64_t V, int C, float* __restrict__ part) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int vl = threadIdx.x >> 6;
  float s = 0.f, q = 0.f;
  if (c < C)
    for (int64_t v = (int64_t)blockIdx.x * 4 + vl; v < V; v += (int64_t)gridDim.x * 4) {
      const float a = Act<T>::ld(x + v * ld + c);
      s += a;
      q = fmaf(a, a, q);
    }
  __shared__ float red[4][2][64];
  red[vl][0][threadIdx.x & 63] = s;
  red[vl][1][threadIdx.x & 63] = q;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, cc = blockIdx.y * 64 + (threadIdx.x & 63), l = threadIdx.x & 63;
    if (cc < C)
      part[((int64_t)blockIdx.x * 2 + which) * C + cc] = red[0][which][l] + red[1][which][l] + red[2][which][l] + red[3][which][l];
  }
}

inline int ew_grid(int64_t total) {
  int64_t g = (total + EW_THREADS - 1) / EW_THREADS;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T>
bool vec_ok(const void* a, int64_t lda, const void* b, int64_t ldb, const void* c, int64_t ldc, int C) {
  constexpr int V = Vec<T>::N;
  auto ok = [&](const void* p, int64_t ld) { return !p || (((uintptr_t)p % 16 == 0) && (ld % V == 0)); };
  return (C % V == 0) && ok(a, lda) && ok(b, ldb) && ok(c, ldc);
}

}  // namespace

static inline bool ew_group_form() {     // A/B knob (benchmarks only): FPLX_EW_GROUP=0 selects the flat-index kernels
  return fplx_knob(FPLX_K_EW_GROUP) != 0;
}

#define DISPATCH_VEC(T, OK, KERNEL, ...)                         \
  do {                                                           \
    if (OK) KERNEL<T, Vec<T>::N> __VA_ARGS__;                    \
    else KERNEL<T, 1> __VA_ARGS__;                               \
  } while (0)

extern "C" {

int fplx_bn_train_finalize(const float* stats, int rows, int c, int64_t count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                           float* mean, float* rstd, float* scale, float* shift, fplx_stream_t stream) {
  FPLX_REQUIRE(stats && gamma && beta && mean && rstd && scale && shift, FPLX_E_NULL, "bn_train_finalize: null pointer");
  FPLX_REQUIRE(rows > 0 && c > 0 && count > 0, FPLX_E_BADSHAPE, "bn_train_finalize: bad shape");
  bn_train_finalize_k<<<c, FIN_THREADS, 0, (hipStream_t)stream>>>(stats, rows, c, (double)count, gamma, beta,
                                                                     running_mean, running_var, nbt, momentum, eps,
                                                                     mean, rstd, scale, shift);
  return fplx_check_launch("bn_train_finalize");
}

int fplx_channel_stats(const void* x, int64_t ldx, int64_t voxels, int c, int dt, float* stats, fplx_stream_t stream) {
  FPLX_REQUIRE(x && stats, FPLX_E_NULL, "channel_stats: null pointer");
  FPLX_REQUIRE(voxels > 0 && c > 0 && ldx >= c, FPLX_E_BADSHAPE, "channel_stats: bad shape");
  dim3 grid(fplx_rows_for(voxels), (c + 63) / 64);
  if (dt == FPLX_F32) channel_stats_k<float><<<grid, EW_THREADS, 0, (hipStream_t)stream>>>((const float*)x, ldx, voxels, c, stats);
  else if (dt == FPLX_BF16)
    channel_stats_k<bf16_t><<<grid, EW_THREADS, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, voxels, c, stats);
  else return fplx_fail(FPLX_E_BADDTYPE, "channel_stats: dtype %d", dt);
  return fplx_check_launch("channel_stats");
}

int fplx_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, int c, float* scale, float* shift, fplx_stream_t stream) {
  FPLX_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, FPLX_E_NULL,
               "bn_eval_prepare: null pointer");
  FPLX_REQUIRE(c > 0, FPLX_E_BADSHAPE, "bn_eval_prepare: bad shape");
  bn_eval_prepare_k<<<(c + 63) / 64, 64, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, c, scale,
                                                                   shift);
  return fplx_check_launch("bn_eval_prepare");
}

int fplx_bn_act_fwd(const void* y, int64_t ldy, void* out, int64_t ldo, const float* scale, const float* shift,
                    const float* slope, float p, uint64_t seed, uint32_t stream_id, int64_t voxels, int c, int dt,
                    fplx_stream_t stream) {
  FPLX_REQUIRE(y && out && scale && shift && slope, FPLX_E_NULL, "bn_act_fwd: null pointer");
  FPLX_REQUIRE(voxels > 0 && c > 0 && ldy >= c && ldo >= c && p >= 0.f && p < 1.f, FPLX_E_BADSHAPE,
               "bn_act_fwd: bad shape/p");
  hipStream_t st = (hipStream_t)stream;
  const DropCfg dc = make_drop(p, seed, stream_id);
  if (dt == FPLX_F32) {
    const bool ok = vec_ok<float>(y, ldy, out, ldo, nullptr, 0, c);
    const int g = ew_grid(voxels * (ok ? c / 4 : c));
    DISPATCH_VEC(float, ok, bn_act_fwd_k, <<<g, EW_THREADS, 0, st>>>((const float*)y, ldy, (float*)out, ldo, scale,
                                                                     shift, slope, dc, voxels, c));
  } else if (dt == FPLX_BF16) {
    const bool ok = vec_ok<bf16_t>(y, ldy, out, ldo, nullptr, 0, c);
    const int g = ew_grid(voxels * (ok ? c / 8 : c));
    if (ok && c / 8 <= EW_THREADS && EW_THREADS % (c / 8) == 0 && ew_group_form())
      bn_act_fwd_g_k<bf16_t, 8><<<g, EW_THREADS, 0, st>>>((const bf16_t*)y, ldy, (bf16_t*)out, ldo, scale, shift, slope, dc,
                                                          voxels, c);
    else
      DISPATCH_VEC(bf16_t, ok, bn_act_fwd_k, <<<g, EW_THREADS, 0, st>>>((const bf16_t*)y, ldy, (bf16_t*)out, ldo, scale,
                                                                        shift, slope, dc, voxels, c));
  } else
    return fplx_fail(FPLX_E_BADDTYPE, "bn_act_fwd: dtype %d", dt);
  return fplx_check_launch("bn_act_fwd");
}

int fplx_bn_act_bwd_reduce(const void* y, int64_t ldy, const void* dout, int64_t ldd, const float* mean,
                           const float* rstd, const float* scale, const float* shift, const float* slope, float p,
                           uint64_t seed, uint32_t stream_id, int64_t voxels, int c, int dt, float* part,
                           fplx_stream_t stream) {
  FPLX_REQUIRE(y && dout && mean && rstd && scale && shift && slope && part, FPLX_E_NULL,
               "bn_act_bwd_reduce: null pointer");
  FPLX_REQUIRE(voxels > 0 && c > 0 && ldy >= c && ldd >= c, FPLX_E_BADSHAPE, "bn_act_bwd_reduce: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const DropCfg dc = make_drop(p, seed, stream_id);
  const int rows = fplx_rows_for(voxels);
  if (dt == FPLX_F32) {
    const bool ok = vec_ok<float>(y, ldy, dout, ldd, nullptr, 0, c) && c / 4 <= EW_THREADS;
    FPLX_REQUIRE(ok || c <= EW_THREADS, FPLX_E_BADSHAPE, "bn_act_bwd_reduce: C=%d unsupported", c);
    DISPATCH_VEC(float, ok, bn_act_bwd_reduce_k, <<<rows, EW_THREADS, 0, st>>>((const float*)y, ldy, (const float*)dout,
                                                                               ldd, mean, rstd, scale, shift, slope,
                                                                               dc, voxels, c, part));
  } else if (dt == FPLX_BF16) {
    const bool ok = vec_ok<bf16_t>(y, ldy, dout, ldd, nullptr, 0, c) && c / 8 <= EW_THREADS;
    FPLX_REQUIRE(ok || c <= EW_THREADS, FPLX_E_BADSHAPE, "bn_act_bwd_reduce: C=%d unsupported", c);
    const int infl = (int)fplx_knob(FPLX_K_EW_INFLIGHT_REDUCE);
#define REDUCE_U(U_, D_) bn_act_bwd_reduce_k<bf16_t, 8, U_, D_><<<rows, EW_THREADS, 0, st>>>((const bf16_t*)y, ldy, (const bf16_t*)dout, \
                         ldd, mean, rstd, scale, shift, slope, dc, voxels, c, part)
    if (ok && dc.on) { if (infl >= 4) REDUCE_U(4, true); else if (infl >= 2) REDUCE_U(2, true); else REDUCE_U(1, true); }
    else if (ok) { if (infl >= 4) REDUCE_U(4, false); else if (infl >= 2) REDUCE_U(2, false); else REDUCE_U(1, false); }
    else
    DISPATCH_VEC(bf16_t, ok, bn_act_bwd_reduce_k, <<<rows, EW_THREADS, 0, st>>>((const bf16_t*)y, ldy,
                                                                                (const bf16_t*)dout, ldd, mean, rstd,
                                                                                scale, shift, slope, dc, voxels, c,
                                                                                part));
#undef REDUCE_U
  } else
    return fplx_fail(FPLX_E_BADDTYPE, "bn_act_bwd_reduce: dtype %d", dt);
  return fplx_check_launch("bn_act_bwd_reduce");
}

int fplx_bn_act_bwd_finalize(const float* part, int rows, int c, int64_t count, int train, float* dgamma, float* dbeta,
                             float* dslope, float* coef, fplx_stream_t stream) {
  FPLX_REQUIRE(part && coef, FPLX_E_NULL, "bn_act_bwd_finalize: null pointer");
  FPLX_REQUIRE(rows > 0 && c > 0 && count > 0, FPLX_E_BADSHAPE, "bn_act_bwd_finalize: bad shape");
  bn_act_bwd_finalize_k<<<c + 1, FIN_THREADS, 0, (hipStream_t)stream>>>(part, rows, c, (double)count, train, dgamma,
                                                                       dbeta, dslope, coef);
  return fplx_check_launch("bn_act_bwd_finalize");
}

int fplx_bn_act_bwd_apply(const void* y, int64_t ldy, const void* dout, int64_t ldd, void* dy, int64_t ldo,
                          const float* mean, const float* rstd, const float* scale, const float* shift,
                          const float* slope, const float* coef, float p, uint64_t seed, uint32_t stream_id,
                          int64_t voxels, int c, int dt, fplx_stream_t stream) {
  FPLX_REQUIRE(y && dout && dy && mean && rstd && scale && shift && slope && coef, FPLX_E_NULL,
               "bn_act_bwd_apply: null pointer");
  FPLX_REQUIRE(voxels > 0 && c > 0 && ldy >= c && ldd >= c && ldo >= c, FPLX_E_BADSHAPE, "bn_act_bwd_apply: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const DropCfg dc = make_drop(p, seed, stream_id);
  if (dt == FPLX_F32) {
    const bool ok = vec_ok<float>(y, ldy, dout, ldd, dy, ldo, c);
    const int g = ew_grid(voxels * (ok ? c / 4 : c));
    DISPATCH_VEC(float, ok, bn_act_bwd_apply_k, <<<g, EW_THREADS, 0, st>>>((const float*)y, ldy, (const float*)dout,
                                                                           ldd, (float*)dy, ldo, mean, rstd, scale,
                                                                           shift, slope, coef, dc, voxels, c));
  } else if (dt == FPLX_BF16) {
    const bool ok = vec_ok<bf16_t>(y, ldy, dout, ldd, dy, ldo, c);
    const int g = ew_grid(voxels * (ok ? c / 8 : c));
    if (ok && c / 8 <= EW_THREADS && EW_THREADS % (c / 8) == 0 && ew_group_form())
    {
      const int infl = (int)fplx_knob(FPLX_K_EW_INFLIGHT);
#define APPLY_G(U_, D_) bn_act_bwd_apply_g_k<bf16_t, 8, U_, D_><<<g, EW_THREADS, 0, st>>>((const bf16_t*)y, ldy, (const bf16_t*)dout, \
                        ldd, (bf16_t*)dy, ldo, mean, rstd, scale, shift, slope, coef, dc, voxels, c)
      if (dc.on) { if (infl >= 4) APPLY_G(4, true); else if (infl >= 2) APPLY_G(2, true); else APPLY_G(1, true); }
      else if (infl >= 4) APPLY_G(4, false); else if (infl >= 2) APPLY_G(2, false); else APPLY_G(1, false);
#undef APPLY_G
    }
    else
      DISPATCH_VEC(bf16_t, ok, bn_act_bwd_apply_k, <<<g, EW_THREADS, 0, st>>>((const bf16_t*)y, ldy, (const bf16_t*)dout,
                                                                              ldd, (bf16_t*)dy, ldo, mean, rstd, scale,
                                                                              shift, slope, coef, dc, voxels, c));
  } else
    return fplx_fail(FPLX_E_BADDTYPE, "bn_act_bwd_apply: dtype %d", dt);
  return fplx_check_launch("bn_act_bwd_apply");
}



static int maxpool_fwd_impl(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c, int dt,
                            int pd, fplx_stream_t stream) {
  FPLX_REQUIRE(x && y, FPLX_E_NULL, "maxpool2_fwd: null pointer");
  FPLX_REQUIRE(n > 0 && c > 0 && d >= pd && h >= 2 && w >= 2 && !(d % pd) && !(h & 1) && !(w & 1) && ldx >= c && ldy >= c,
               FPLX_E_BADSHAPE, "maxpool2_fwd: bad shape (even %sH,W required) %dx%dx%d", pd == 2 ? "D," : "", d, h, w);
  hipStream_t st = (hipStream_t)stream;
  const int64_t vo = (int64_t)n * (d / pd) * (h / 2) * (w / 2);
  if (dt == FPLX_F32) {
    const bool ok = vec_ok<float>(x, ldx, y, ldy, nullptr, 0, c);
    DISPATCH_VEC(float, ok, maxpool2_fwd_k, <<<ew_grid(vo * (ok ? c / 4 : c)), EW_THREADS, 0, st>>>(
                                                (const float*)x, ldx, (float*)y, ldy, n, d, h, w, c, pd));
  } else if (dt == FPLX_BF16) {
    const bool ok = vec_ok<bf16_t>(x, ldx, y, ldy, nullptr, 0, c);
    DISPATCH_VEC(bf16_t, ok, maxpool2_fwd_k, <<<ew_grid(vo * (ok ? c / 8 : c)), EW_THREADS, 0, st>>>(
                                                 (const bf16_t*)x, ldx, (bf16_t*)y, ldy, n, d, h, w, c, pd));
  } else
    return fplx_fail(FPLX_E_BADDTYPE, "maxpool2_fwd: dtype %d", dt);
  return fplx_check_launch("maxpool2_fwd");
}

static int maxpool_bwd_impl(const void* x, int64_t ldx, const void* dy, int64_t ldy, const void* dskip, int64_t lds,
                            void* dx, int64_t ldo, int n, int d, int h, int w, int c, int dt, int pd, fplx_stream_t stream) {
  FPLX_REQUIRE(x && dy && dx, FPLX_E_NULL, "maxpool2_bwd: null pointer");
  FPLX_REQUIRE(n > 0 && c > 0 && d >= pd && h >= 2 && w >= 2 && !(d % pd) && !(h & 1) && !(w & 1), FPLX_E_BADSHAPE,
               "maxpool2_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int64_t vo = (int64_t)n * (d / pd) * (h / 2) * (w / 2);
  if (dt == FPLX_F32) {
    const bool ok = vec_ok<float>(x, ldx, dy, ldy, dx, ldo, c) && vec_ok<float>(dskip, lds, nullptr, 0, nullptr, 0, c);
    DISPATCH_VEC(float, ok, maxpool2_bwd_k, <<<ew_grid(vo * (ok ? c / 4 : c)), EW_THREADS, 0, st>>>(
                                                (const float*)x, ldx, (const float*)dy, ldy, (const float*)dskip, lds,
                                                (float*)dx, ldo, n, d, h, w, c, pd));
  } else if (dt == FPLX_BF16) {
    const bool ok = vec_ok<bf16_t>(x, ldx, dy, ldy, dx, ldo, c) && vec_ok<bf16_t>(dskip, lds, nullptr, 0, nullptr, 0, c);
    DISPATCH_VEC(bf16_t, ok, maxpool2_bwd_k, <<<ew_grid(vo * (ok ? c / 8 : c)), EW_THREADS, 0, st>>>(
                                                 (const bf16_t*)x, ldx, (const bf16_t*)dy, ldy, (const bf16_t*)dskip,
                                                 lds, (bf16_t*)dx, ldo, n, d, h, w, c, pd));
  } else
    return fplx_fail(FPLX_E_BADDTYPE, "maxpool2_bwd: dtype %d", dt);
  return fplx_check_launch("maxpool2_bwd");
}

int fplx_maxpool2_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c, int dt,
                      fplx_stream_t stream) {
  return maxpool_fwd_impl(x, ldx, y, ldy, n, d, h, w, c, dt, 2, stream);
}
int fplx_maxpool2_bwd(const void* x, int64_t ldx, const void* dy, int64_t ldy, const void* dskip, int64_t lds, void* dx,
                      int64_t ldo, int n, int d, int h, int w, int c, int dt, fplx_stream_t stream) {
  return maxpool_bwd_impl(x, ldx, dy, ldy, dskip, lds, dx, ldo, n, d, h, w, c, dt, 2, stream);
}
int fplx_maxpool122_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c, int dt,
                        fplx_stream_t stream) {
  return maxpool_fwd_impl(x, ldx, y, ldy, n, d, h, w, c, dt, 1, stream);
}
int fplx_maxpool122_bwd(const void* x, int64_t ldx, const void* dy, int64_t ldy, const void* dskip, int64_t lds, void* dx,
                        int64_t ldo, int n, int d, int h, int w, int c, int dt, fplx_stream_t stream) {
  return maxpool_bwd_impl(x, ldx, dy, ldy, dskip, lds, dx, ldo, n, d, h, w, c, dt, 1, stream);
}

// 1 if the fused DownBlock-tail kernels take this shape (bf16, 16-byte channel groups that tile a block), else 0
int fplx_bn_pool_fused_ok(int c, int dt) {
  const int g = c / 8;
  return dt == FPLX_BF16 && c % 8 == 0 && g >= 1 && g <= 64 && (g & (g - 1)) == 0 ? 1 : 0;
}

static int bn_pool_args_ok(const char* what, int n, int d, int h, int w, int c, int dt, int pd) {
  FPLX_REQUIRE(n > 0 && c > 0 && d >= pd && h >= 2 && w >= 2 && !(d % pd) && !(h & 1) && !(w & 1) && (pd == 1 || pd == 2),
               FPLX_E_BADSHAPE, "%s: bad shape", what);
  FPLX_REQUIRE(fplx_bn_pool_fused_ok(c, dt), FPLX_E_BADSHAPE, "%s: C=%d dtype %d not supported (fplx_bn_pool_fused_ok)", what, c, dt);
  return FPLX_OK;
}

int fplx_bn_act_pool_fwd(const void* y, int64_t ldy, void* out, int64_t ldo, void* pooled, int64_t ldp, const float* scale,
                         const float* shift, const float* slope, int n, int d, int h, int w, int c, int dt, int pd,
                         fplx_stream_t stream) {
  FPLX_REQUIRE(y && out && pooled && scale && shift && slope, FPLX_E_NULL, "bn_act_pool_fwd: null pointer");
  const int rc = bn_pool_args_ok("bn_act_pool_fwd", n, d, h, w, c, dt, pd);
  if (rc != FPLX_OK) return rc;
  FPLX_REQUIRE(vec_ok<bf16_t>(y, ldy, out, ldo, pooled, ldp, c), FPLX_E_BADSHAPE, "bn_act_pool_fwd: pointers / leading dimensions not 16-byte aligned");
  const int64_t vo = (int64_t)n * (d / pd) * (h / 2) * (w / 2);
  if (fplx_knob(FPLX_K_POOL_COL) && c / 8 <= 32 && 2 * vo < ((int64_t)1 << 31))       // lanes along the input row (round 3)
    bn_act_pool_fwd_col_k<bf16_t, 8><<<ew_grid(2 * vo * (c / 8)), EW_THREADS, 0, (hipStream_t)stream>>>(
        (const bf16_t*)y, ldy, (bf16_t*)out, ldo, (bf16_t*)pooled, ldp, scale, shift, slope, n, d, h, w, c, pd);
  else
  bn_act_pool_fwd_k<bf16_t, 8><<<ew_grid(vo * (c / 8)), EW_THREADS, 0, (hipStream_t)stream>>>(
      (const bf16_t*)y, ldy, (bf16_t*)out, ldo, (bf16_t*)pooled, ldp, scale, shift, slope, n, d, h, w, c, pd);
  return fplx_check_launch("bn_act_pool_fwd");
}

int fplx_pool_bwd_bn_reduce(const void* y, int64_t ldy, const void* dy, int64_t lddy, const void* dskip, int64_t lds, void* dx,
                            int64_t ldo, const float* mean, const float* rstd, const float* scale, const float* shift,
                            const float* slope, int n, int d, int h, int w, int c, int dt, int pd, float* part,
                            fplx_stream_t stream) {
  FPLX_REQUIRE(y && dy && dx && mean && rstd && scale && shift && slope && part, FPLX_E_NULL, "pool_bwd_bn_reduce: null pointer");
  const int rc = bn_pool_args_ok("pool_bwd_bn_reduce", n, d, h, w, c, dt, pd);
  if (rc != FPLX_OK) return rc;
  FPLX_REQUIRE(vec_ok<bf16_t>(y, ldy, dy, lddy, dx, ldo, c) && vec_ok<bf16_t>(dskip, lds, nullptr, 0, nullptr, 0, c), FPLX_E_BADSHAPE,
               "pool_bwd_bn_reduce: pointers / leading dimensions not 16-byte aligned");
  const int rows = fplx_rows_for((int64_t)n * d * h * w);          // the partial rows bn_act_bwd_finalize expects for this tensor
  if (fplx_knob(FPLX_K_POOL_COL) && c / 8 <= 32 && (int64_t)n * (d / pd) * (h / 2) * w < ((int64_t)1 << 31))
    pool_bwd_bn_reduce_col_k<bf16_t, 8><<<rows, EW_THREADS, 0, (hipStream_t)stream>>>(
        (const bf16_t*)y, ldy, (const bf16_t*)dy, lddy, (const bf16_t*)dskip, lds, (bf16_t*)dx, ldo, mean, rstd, scale, shift, slope, n,
        d, h, w, c, pd, part);
  else
  pool_bwd_bn_reduce_k<bf16_t, 8><<<rows, EW_THREADS, 0, (hipStream_t)stream>>>(
      (const bf16_t*)y, ldy, (const bf16_t*)dy, lddy, (const bf16_t*)dskip, lds, (bf16_t*)dx, ldo, mean, rstd, scale, shift, slope, n,
      d, h, w, c, pd, part);
  return fplx_check_launch("pool_bwd_bn_reduce");
}

int fplx_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, fplx_stream_t stream) {
  FPLX_REQUIRE(p && g && m && v, FPLX_E_NULL, "adam_step: null pointer");
  FPLX_REQUIRE(n > 0 && step >= 1, FPLX_E_BADSHAPE, "adam_step: n=%lld step=%d", (long long)n, step);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  adam_k<<<ew_grid(n), EW_THREADS, 0, (hipStream_t)stream>>>(p, g, m, v, n, (float)((double)lr / bc1), beta1, beta2, eps,
                                                            weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale);
  return fplx_check_launch("adam_step");
}

}  // extern "C"
