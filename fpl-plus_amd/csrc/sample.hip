// Training-sample transforms on the GPU: the reference's numpy chain
//   NormalizeWithMeanStd -> Pad(reflect) -> RandomCrop -> RandomFlip -> LabelToProbability
// (PyMIC/pymic/transform/normalize.py:43-68, pad.py:126-163, crop.py:27-49,201-236, flip.py:34-62,
// label_convert.py:82-94).  The random decisions stay on the host (fplx/transform.py draws from Python's `random` in
// the reference's order); these kernels are the data movement and the arithmetic.  Volumes are [C][D][H][W], tiny
// next to the network's activations: one thread per output element, no tuning.
#include "common.h"

namespace {

constexpr int SP_THREADS = 256;
constexpr int SP_BLOCKS = 256;                 // partial rows of the moment reductions

__global__ void __launch_bounds__(SP_THREADS)
moments_sum_k(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
  __shared__ double red[SP_THREADS];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) s += (double)x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = SP_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__device__ __forceinline__ double total_of(const double* __restrict__ part, int rows) {   // fixed order
  double t = 0.0;
  for (int i = 0; i < rows; ++i) t += part[i];
  return t;
}

// second pass (numpy's std: mean first, then the mean of squared deviations)
__global__ void __launch_bounds__(SP_THREADS)
moments_dev_k(const float* __restrict__ x, int64_t n, const double* __restrict__ sums, int rows, double* __restrict__ part) {
  __shared__ double red[SP_THREADS];
  const double mean = total_of(sums, rows) / (double)n;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const double d = (double)x[i] - mean;
    s += d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = SP_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(SP_THREADS)
normalize_k(const float* __restrict__ x, float* __restrict__ y, int64_t n, const double* __restrict__ sums,
            const double* __restrict__ devs, int rows, const float* __restrict__ given, float* __restrict__ out_ms) {
  float mean, sd;
  if (given) { mean = given[0]; sd = given[1]; }
  else {
    // float32 mean / std like numpy on a float32 array, then float32 arithmetic
    mean = (float)(total_of(sums, rows) / (double)n);
    sd = (float)sqrt(total_of(devs, rows) / (double)n);
  }
  if (out_ms && blockIdx.x == 0 && threadIdx.x == 0) { out_ms[0] = mean; out_ms[1] = sd; }
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS)
    y[i] = (x[i] - mean) / sd;
}

// NormalizeWithMeanStd_ignore_non_positive (normalize.py:55-66): moments over the voxels > 0 only, and the others replaced
// by the caller's noise volume (the reference draws numpy.random.normal on the host; the draw stays there)
__global__ void __launch_bounds__(SP_THREADS)
moments_pos_sum_k(const float* __restrict__ x, int64_t n, double* __restrict__ part, double* __restrict__ cnt) {
  __shared__ double red[SP_THREADS], redc[SP_THREADS];
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS)
    if (x[i] > 0.f) { s += (double)x[i]; c += 1.0; }
  red[threadIdx.x] = s; redc[threadIdx.x] = c;
  __syncthreads();
  for (int o = SP_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; redc[threadIdx.x] += redc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[blockIdx.x] = red[0]; cnt[blockIdx.x] = redc[0]; }
}

__global__ void __launch_bounds__(SP_THREADS)
moments_pos_dev_k(const float* __restrict__ x, int64_t n, const double* __restrict__ sums, const double* __restrict__ cnt, int rows,
                  double* __restrict__ part) {
  __shared__ double red[SP_THREADS];
  const double mean = total_of(sums, rows) / total_of(cnt, rows);
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS)
    if (x[i] > 0.f) { const double d = (double)x[i] - mean; s += d * d; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = SP_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(SP_THREADS)
normalize_pos_k(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ y, int64_t n,
                const double* __restrict__ sums, const double* __restrict__ cnt, const double* __restrict__ devs, int rows,
                float* __restrict__ out_ms) {
  const double m = total_of(cnt, rows);
  const float mean = (float)(total_of(sums, rows) / m);
  const float sd = (float)sqrt(total_of(devs, rows) / m);
  if (out_ms && blockIdx.x == 0 && threadIdx.x == 0) { out_ms[0] = mean; out_ms[1] = sd; }
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const float v = x[i];                                    // x may alias y
    y[i] = (v <= 0.f) ? noise[i] : (v - mean) / sd;        // the reference's test (a NaN stays a NaN)
  }
}

// numpy.pad(mode='reflect') index map: mirror without repeating the edge, any number of reflections
__device__ __forceinline__ int reflect_index(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  int m = i % period;
  if (m < 0) m += period;
  return m < n ? m : period - m;
}

template <typename T>
__global__ void __launch_bounds__(SP_THREADS)
pad_reflect_k(const T* __restrict__ x, T* __restrict__ y, int C, int D, int H, int W, int ld, int lh, int lw, int OD,
              int OH, int OW) {
  const int64_t total = (int64_t)C * OD * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SP_THREADS) {
    int64_t r = i;
    const int w = (int)(r % OW); r /= OW;
    const int h = (int)(r % OH); r /= OH;
    const int d = (int)(r % OD); r /= OD;
    const int c = (int)r;
    y[i] = x[(((int64_t)c * D + reflect_index(d - ld, D)) * H + reflect_index(h - lh, H)) * W + reflect_index(w - lw, W)];
  }
}

// crop [cd, cd+OD) x [ch, ch+OH) x [cw, cw+OW) and flip the CROPPED patch along the axes in flip (bit 0 = w, 1 = h, 2 = d)
template <typename T>
__global__ void __launch_bounds__(SP_THREADS)
crop_flip_k(const T* __restrict__ x, T* __restrict__ y, int C, int D, int H, int W, int cd, int ch, int cw, int OD, int OH,
            int OW, int flip) {
  const int64_t total = (int64_t)C * OD * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SP_THREADS) {
    int64_t r = i;
    int w = (int)(r % OW); r /= OW;
    int h = (int)(r % OH); r /= OH;
    int d = (int)(r % OD); r /= OD;
    const int c = (int)r;
    if (flip & 1) w = OW - 1 - w;
    if (flip & 2) h = OH - 1 - h;
    if (flip & 4) d = OD - 1 - d;
    y[i] = x[(((int64_t)c * D + cd + d) * H + ch + h) * W + cw + w];
  }
}

// bounding box of {label in mask_labels}: out = [count, min_c, min_d, min_h, min_w, max_c+1, max_d+1, max_h+1, max_w+1]
__global__ void __launch_bounds__(SP_THREADS)
label_bbox_k(const unsigned char* __restrict__ lab, int C, int D, int H, int W, const int* __restrict__ mask_labels,
             int nmask, int* __restrict__ out) {
  const int64_t total = (int64_t)C * D * H * W;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SP_THREADS) {
    const int v = lab[i];
    bool hit = false;
    for (int k = 0; k < nmask; ++k) hit |= v == mask_labels[k];
    if (!hit) continue;
    int64_t r = i;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D); r /= D;
    const int c = (int)r;
    atomicAdd(out, 1);
    atomicMin(out + 1, c); atomicMin(out + 2, d); atomicMin(out + 3, h); atomicMin(out + 4, w);
    atomicMax(out + 5, c + 1); atomicMax(out + 6, d + 1); atomicMax(out + 7, h + 1); atomicMax(out + 8, w + 1);
  }
}

__global__ void label_bbox_init_k(int* out) {
  if (threadIdx.x == 0) out[0] = 0;
  if (threadIdx.x >= 1 && threadIdx.x <= 4) out[threadIdx.x] = 0x7fffffff;
  if (threadIdx.x >= 5 && threadIdx.x <= 8) out[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(SP_THREADS)
onehot_k(const unsigned char* __restrict__ lab, float* __restrict__ prob, int classes, int64_t voxels) {
  const int64_t total = (int64_t)classes * voxels;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SP_THREADS)
    prob[i] = lab[i % voxels] == (unsigned char)(i / voxels) ? 1.f : 0.f;
}

// NiftyDataset.set_weight_: weights below 1 (the two pseudo-label masks disagree) drop to 0, the rest scale by the image weight
__global__ void __launch_bounds__(SP_THREADS)
set_weight_k(float* __restrict__ pw, int64_t n, float iw) {
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const float v = pw[i];
    pw[i] = (v < 1.f ? 0.f : v) * iw;
  }
}

// evaluation: per label l the voxel counts |s==l & g==l|, |s==l|, |g==l| (fuse: one row, membership in the label list)
constexpr int EV_MAX_LABELS = 16;
__global__ void __launch_bounds__(SP_THREADS)
overlap_counts_k(const unsigned char* __restrict__ s, const unsigned char* __restrict__ g, int64_t n,
                 const int* __restrict__ labels, int nlab, int fuse, unsigned long long* __restrict__ out) {
  __shared__ unsigned red[SP_THREADS / 64][EV_MAX_LABELS * 3];
  unsigned cnt[EV_MAX_LABELS * 3];
  const int rows = fuse ? 1 : nlab;
#pragma unroll
  for (int k = 0; k < EV_MAX_LABELS * 3; ++k) cnt[k] = 0;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const int sv = s[i], gv = g[i];
    if (fuse) {
      bool si = false, gi = false;
      for (int k = 0; k < nlab; ++k) { si |= sv == labels[k]; gi |= gv == labels[k]; }
      cnt[0] += si && gi; cnt[1] += si; cnt[2] += gi;
    } else {
#pragma unroll
      for (int k = 0; k < EV_MAX_LABELS; ++k)
        if (k < nlab) {
          const bool si = sv == labels[k], gi = gv == labels[k];
          cnt[3 * k] += si && gi; cnt[3 * k + 1] += si; cnt[3 * k + 2] += gi;
        }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < EV_MAX_LABELS * 3; ++k) {
    if (k < rows * 3) {                                      // block-uniform
      unsigned v = cnt[k];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) red[wave][k] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < rows * 3) {
    unsigned long long t = 0;
    for (int w = 0; w < SP_THREADS / 64; ++w) t += red[w][threadIdx.x];
    if (t) atomicAdd(out + threadIdx.x, t);
  }
}

// evaluation, surface metrics (binary_assd / binary_hd95): get_edge_points = img minus its erosion with the 6- (2D: 4-)
// neighbour cross, outside the volume counting as background (scipy's binary_erosion, border_value 0)
__global__ void __launch_bounds__(SP_THREADS)
edge_points_k(const unsigned char* __restrict__ img, int D, int H, int W, unsigned char* __restrict__ edge) {
  const int64_t n = (int64_t)D * H * W;
  const int64_t hw = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    unsigned char e = 0;
    if (img[i]) {
      const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / hw);
      bool inner = x > 0 && x < W - 1 && y > 0 && y < H - 1 && img[i - 1] && img[i + 1] && img[i - W] && img[i + W];
      if (D > 1) inner = inner && z > 0 && z < D - 1 && img[i - hw] && img[i + hw];
      e = inner ? 0 : 1;
    }
    edge[i] = e;
  }
}

// Distance from every query voxel to the nearest seed voxel in the metric GeodisTK's raster scan converges to on a
// constant image (lambda = 0): shortest 26-neighbour lattice path with step lengths sqrt(sum (d_axis * spacing_axis)^2).
// For a displacement with per-axis voxel counts a >= b >= c (axes A, B, C) that path is c body diagonals, b - c
// diagonals in the A-B plane and a - b steps along A (replacing two moves by a more diagonal pair never lengthens the
// path: norms are sub-additive and sqrt(s^2 + x) is concave in x).  One thread per query, seeds tiled through LDS.
constexpr int SD_TILE = 1024;
__global__ void __launch_bounds__(SP_THREADS)
surface_min_dist_k(const int* __restrict__ q, int64_t nq, const int* __restrict__ s, int64_t ns, float sz, float sy,
                   float sx, float* __restrict__ out) {
  __shared__ int tile[SD_TILE * 3];
  const int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x;
  const bool live = i < nq;
  const int qz = live ? q[i * 3] : 0, qy = live ? q[i * 3 + 1] : 0, qx = live ? q[i * 3 + 2] : 0;
  const float w3 = sqrtf(sz * sz + sy * sy + sx * sx);
  float best = 1.0e10f;                                      // the raster scan's initial distance (no seed reached)
  for (int64_t t0 = 0; t0 < ns; t0 += SD_TILE) {
    const int cnt = (int)(ns - t0 < SD_TILE ? ns - t0 : SD_TILE);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt * 3; k += SP_THREADS) tile[k] = s[t0 * 3 + k];
    __syncthreads();
    if (!live) continue;
    for (int k = 0; k < cnt; ++k) {
      int a = abs(qz - tile[3 * k]), b = abs(qy - tile[3 * k + 1]), c = abs(qx - tile[3 * k + 2]);
      float sa = sz, sb = sy, sc = sx;
      if (a < b) { const int t = a; a = b; b = t; const float u = sa; sa = sb; sb = u; }
      if (b < c) { const int t = b; b = c; c = t; const float u = sb; sb = sc; sc = u; }
      if (a < b) { const int t = a; a = b; b = t; const float u = sa; sa = sb; sb = u; }
      const float d = (float)c * w3 + (float)(b - c) * sqrtf(sa * sa + sb * sb) + (float)(a - b) * sa;
      best = fminf(best, d);
    }
  }
  if (live) out[i] = best;
}

inline int sp_grid(int64_t total) {
  int64_t g = (total + SP_THREADS - 1) / SP_THREADS;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

size_t fplx_normalize_ws_bytes(void) { return 3 * SP_BLOCKS * sizeof(double); }

int fplx_normalize_mean_std(const float* x, float* y, int64_t n, const float* mean_std, void* ws, size_t ws_bytes,
                            float* out_mean_std, fplx_stream_t stream) {
  FPLX_REQUIRE(x && y && n > 0, FPLX_E_NULL, "normalize_mean_std: null pointer / empty volume");
  hipStream_t st = (hipStream_t)stream;
  double* sums = (double*)ws;
  double* devs = sums + SP_BLOCKS;
  if (!mean_std) {
    FPLX_REQUIRE(ws && ws_bytes >= fplx_normalize_ws_bytes(), FPLX_E_WORKSPACE, "normalize_mean_std: workspace %zu < %zu",
                 ws_bytes, fplx_normalize_ws_bytes());
    moments_sum_k<<<SP_BLOCKS, SP_THREADS, 0, st>>>(x, n, sums);
    moments_dev_k<<<SP_BLOCKS, SP_THREADS, 0, st>>>(x, n, sums, SP_BLOCKS, devs);
  }
  normalize_k<<<sp_grid(n), SP_THREADS, 0, st>>>(x, y, n, sums, devs, SP_BLOCKS, mean_std, out_mean_std);
  return fplx_check_launch("normalize_mean_std");
}

int fplx_normalize_positive(const float* x, const float* noise, float* y, int64_t n, void* ws, size_t ws_bytes,
                            float* out_mean_std, fplx_stream_t stream) {
  FPLX_REQUIRE(x && noise && y && n > 0, FPLX_E_NULL, "normalize_positive: null pointer / empty volume");
  FPLX_REQUIRE(ws && ws_bytes >= 3 * SP_BLOCKS * sizeof(double), FPLX_E_WORKSPACE, "normalize_positive: workspace %zu < %zu", ws_bytes,
               3 * SP_BLOCKS * sizeof(double));
  hipStream_t st = (hipStream_t)stream;
  double* sums = (double*)ws;
  double* devs = sums + SP_BLOCKS;
  double* cnt = devs + SP_BLOCKS;
  moments_pos_sum_k<<<SP_BLOCKS, SP_THREADS, 0, st>>>(x, n, sums, cnt);
  moments_pos_dev_k<<<SP_BLOCKS, SP_THREADS, 0, st>>>(x, n, sums, cnt, SP_BLOCKS, devs);
  normalize_pos_k<<<sp_grid(n), SP_THREADS, 0, st>>>(x, noise, y, n, sums, cnt, devs, SP_BLOCKS, out_mean_std);
  return fplx_check_launch("normalize_positive");
}

int fplx_pad_reflect(const void* x, void* y, int elem_bytes, int c, int d, int h, int w, int lo_d, int lo_h, int lo_w,
                     int od, int oh, int ow, fplx_stream_t stream) {
  FPLX_REQUIRE(x && y, FPLX_E_NULL, "pad_reflect: null pointer");
  FPLX_REQUIRE(c > 0 && d > 0 && h > 0 && w > 0 && od >= d && oh >= h && ow >= w && lo_d >= 0 && lo_h >= 0 && lo_w >= 0 &&
                   lo_d <= od - d && lo_h <= oh - h && lo_w <= ow - w,
               FPLX_E_BADSHAPE, "pad_reflect: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int g = sp_grid((int64_t)c * od * oh * ow);
  if (elem_bytes == 4)
    pad_reflect_k<float><<<g, SP_THREADS, 0, st>>>((const float*)x, (float*)y, c, d, h, w, lo_d, lo_h, lo_w, od, oh, ow);
  else if (elem_bytes == 1)
    pad_reflect_k<unsigned char><<<g, SP_THREADS, 0, st>>>((const unsigned char*)x, (unsigned char*)y, c, d, h, w, lo_d,
                                                          lo_h, lo_w, od, oh, ow);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "pad_reflect: element size %d", elem_bytes);
  return fplx_check_launch("pad_reflect");
}

int fplx_crop_flip(const void* x, void* y, int elem_bytes, int c, int d, int h, int w, int cd, int ch, int cw, int od,
                   int oh, int ow, int flip_mask, fplx_stream_t stream) {
  FPLX_REQUIRE(x && y, FPLX_E_NULL, "crop_flip: null pointer");
  FPLX_REQUIRE(c > 0 && od > 0 && oh > 0 && ow > 0 && cd >= 0 && ch >= 0 && cw >= 0 && cd + od <= d && ch + oh <= h &&
                   cw + ow <= w,
               FPLX_E_BADSHAPE, "crop_flip: crop box outside the volume");
  hipStream_t st = (hipStream_t)stream;
  const int g = sp_grid((int64_t)c * od * oh * ow);
  if (elem_bytes == 4)
    crop_flip_k<float><<<g, SP_THREADS, 0, st>>>((const float*)x, (float*)y, c, d, h, w, cd, ch, cw, od, oh, ow, flip_mask);
  else if (elem_bytes == 1)
    crop_flip_k<unsigned char><<<g, SP_THREADS, 0, st>>>((const unsigned char*)x, (unsigned char*)y, c, d, h, w, cd, ch, cw,
                                                        od, oh, ow, flip_mask);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "crop_flip: element size %d", elem_bytes);
  return fplx_check_launch("crop_flip");
}

int fplx_label_bbox(const unsigned char* label, int c, int d, int h, int w, const int* mask_labels, int nmask, int* out9,
                    fplx_stream_t stream) {
  FPLX_REQUIRE(label && mask_labels && out9, FPLX_E_NULL, "label_bbox: null pointer");
  FPLX_REQUIRE(c > 0 && d > 0 && h > 0 && w > 0 && nmask > 0, FPLX_E_BADSHAPE, "label_bbox: bad shape");
  hipStream_t st = (hipStream_t)stream;
  label_bbox_init_k<<<1, 64, 0, st>>>(out9);
  label_bbox_k<<<sp_grid((int64_t)c * d * h * w), SP_THREADS, 0, st>>>(label, c, d, h, w, mask_labels, nmask, out9);
  return fplx_check_launch("label_bbox");
}

int fplx_label_to_probability(const unsigned char* label, float* prob, int class_num, int64_t voxels, fplx_stream_t stream) {
  FPLX_REQUIRE(label && prob, FPLX_E_NULL, "label_to_probability: null pointer");
  FPLX_REQUIRE(class_num > 0 && class_num <= 255 && voxels > 0, FPLX_E_BADSHAPE, "label_to_probability: bad shape");
  onehot_k<<<sp_grid((int64_t)class_num * voxels), SP_THREADS, 0, (hipStream_t)stream>>>(label, prob, class_num, voxels);
  return fplx_check_launch("label_to_probability");
}

int fplx_set_weight(float* pixel_weight, int64_t n, float image_weight, fplx_stream_t stream) {
  FPLX_REQUIRE(pixel_weight && n > 0, FPLX_E_NULL, "set_weight: null pointer / empty volume");
  set_weight_k<<<sp_grid(n), SP_THREADS, 0, (hipStream_t)stream>>>(pixel_weight, n, image_weight);
  return fplx_check_launch("set_weight");
}

int fplx_overlap_counts(const unsigned char* seg, const unsigned char* gt, int64_t n, const int* labels, int nlabels,
                        int fuse, unsigned long long* out, fplx_stream_t stream) {
  FPLX_REQUIRE(seg && gt && labels && out, FPLX_E_NULL, "overlap_counts: null pointer");
  FPLX_REQUIRE(n > 0 && n < ((int64_t)1 << 40) && nlabels > 0 && nlabels <= EV_MAX_LABELS, FPLX_E_BADSHAPE,
               "overlap_counts: %d labels (1..%d) over %lld voxels", nlabels, EV_MAX_LABELS, (long long)n);
  hipStream_t st = (hipStream_t)stream;
  const int rows = fuse ? 1 : nlabels;
  if (hipMemsetAsync(out, 0, sizeof(unsigned long long) * 3 * rows, st) != hipSuccess)
    return fplx_fail(FPLX_E_HIP, "overlap_counts: memset failed");
  // a thread sees at most n / (grid * 256) voxels: the 32-bit per-thread counters cannot overflow
  overlap_counts_k<<<sp_grid(n), SP_THREADS, 0, st>>>(seg, gt, n, labels, nlabels, fuse, out);
  return fplx_check_launch("overlap_counts");
}

int fplx_surface_edge_points(const unsigned char* img, int d, int h, int w, unsigned char* edge, fplx_stream_t stream) {
  FPLX_REQUIRE(img && edge, FPLX_E_NULL, "edge_points: null pointer");
  FPLX_REQUIRE(d > 0 && h > 0 && w > 0, FPLX_E_BADSHAPE, "edge_points: bad shape %dx%dx%d", d, h, w);
  edge_points_k<<<sp_grid((int64_t)d * h * w), SP_THREADS, 0, (hipStream_t)stream>>>(img, d, h, w, edge);
  return fplx_check_launch("edge_points");
}

int fplx_surface_min_dist(const int* query_zyx, int64_t nq, const int* seed_zyx, int64_t ns, float sz, float sy, float sx,
                          float* out, fplx_stream_t stream) {
  FPLX_REQUIRE(query_zyx && out && (seed_zyx || ns == 0), FPLX_E_NULL, "surface_min_dist: null pointer");
  FPLX_REQUIRE(nq > 0 && ns >= 0 && nq < ((int64_t)1 << 31) * SP_THREADS, FPLX_E_BADSHAPE,
               "surface_min_dist: %lld queries, %lld seeds", (long long)nq, (long long)ns);
  FPLX_REQUIRE(sz > 0.f && sy > 0.f && sx > 0.f, FPLX_E_BADSHAPE, "surface_min_dist: spacing must be positive");
  const unsigned grid = (unsigned)((nq + SP_THREADS - 1) / SP_THREADS);
  surface_min_dist_k<<<grid, SP_THREADS, 0, (hipStream_t)stream>>>(query_zyx, nq, seed_zyx, ns, sz, sy, sx, out);
  return fplx_check_launch("surface_min_dist");
}

}  // extern "C"
