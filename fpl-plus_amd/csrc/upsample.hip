// (Tri / bi)linear x2 upsampling with align_corners = True, forward and backward - the `bilinear = True` form of UpBlock
// (reference PyMIC/pymic/net/net3d/unet2d5_dsbn.py:148-150, 172-176: nn.Upsample(scale_factor=2, mode='trilinear' | 'bilinear',
// align_corners=True) behind a kernel-1 convolution).  NDHWC activations, [voxels][ld]; sd = 2: all three axes doubled
// (trilinear), sd = 1: H and W doubled on every depth slice (bilinear on the depth-folded tensor of a 2.5D level).
// Source coordinates follow ATen: src = o * (in - 1) / (out - 1) in float32, i0 = (int)src, l1 = src - i0, i1 = i0 + (i0 < in - 1).
// No shipped configuration uses this branch: one thread per output (forward) / input (backward, a gather over the few outputs
// that touch the voxel - deterministic, no atomics) group of channels, no tuning.
#include "common.h"

namespace {

constexpr int UP_THREADS = 256;

struct Axis { int i0, i1; float l0, l1; };

__device__ __forceinline__ Axis src_of(int o, int in, int out) {
  Axis a;
  if (out == in) { a.i0 = a.i1 = o; a.l0 = 1.f; a.l1 = 0.f; return a; }
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float s = scale * (float)o;
  a.i0 = (int)s;
  if (a.i0 > in - 1) a.i0 = in - 1;
  a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
  float l1 = s - (float)a.i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  a.l1 = l1;
  a.l0 = 1.f - l1;
  return a;
}

template <typename T>
__global__ void __launch_bounds__(UP_THREADS)
upsample2_fwd_k(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int N, int D, int H, int W, int C, int sd) {
  const int Do = D * sd, Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * Do * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * UP_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * UP_THREADS) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do);
    const int n = (int)(r / Do);
    const Axis ad = src_of(od, D, Do), ah = src_of(oh, H, Ho), aw = src_of(ow, W, Wo);
    auto at = [&](int dd, int hh, int ww) {
      return Act<T>::ld(x + ((((int64_t)n * D + dd) * H + hh) * W + ww) * ldx + c);
    };
    // ATen's order: interpolate along w, then h, then d
    const float r00 = aw.l0 * at(ad.i0, ah.i0, aw.i0) + aw.l1 * at(ad.i0, ah.i0, aw.i1);
    const float r01 = aw.l0 * at(ad.i0, ah.i1, aw.i0) + aw.l1 * at(ad.i0, ah.i1, aw.i1);
    const float r10 = aw.l0 * at(ad.i1, ah.i0, aw.i0) + aw.l1 * at(ad.i1, ah.i0, aw.i1);
    const float r11 = aw.l0 * at(ad.i1, ah.i1, aw.i0) + aw.l1 * at(ad.i1, ah.i1, aw.i1);
    const float v = ad.l0 * (ah.l0 * r00 + ah.l1 * r01) + ad.l1 * (ah.l0 * r10 + ah.l1 * r11);
    Act<T>::st(y + ((((int64_t)n * Do + od) * Ho + oh) * Wo + ow) * ldy + c, v);
  }
}

// weight of output o on input i along one axis (i0 == i1 at the upper border: both terms land on the same voxel)
__device__ __forceinline__ float wt_of(int o, int i, int in, int out) {
  const Axis a = src_of(o, in, out);
  return (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
}

template <typename T>
__global__ void __launch_bounds__(UP_THREADS)
upsample2_bwd_k(const T* __restrict__ dy, int64_t ldy, T* __restrict__ dx, int64_t ldx, int N, int D, int H, int W, int C, int sd) {
  const int Do = D * sd, Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * D * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * UP_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * UP_THREADS) {
    int64_t r = i;
    const int c = (int)(r % C); r /= C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); r /= H;
    const int d = (int)(r % D);
    const int n = (int)(r / D);
    // outputs whose source coordinate lies in (i - 1, i + 1): o in ((i - 1) * (out - 1) / (in - 1), (i + 1) * ...), padded by one
    auto lo_of = [](int i_, int in, int out) { return in == out ? i_ : (in > 1 ? max(0, (int)(((int64_t)(i_ - 1) * (out - 1)) / (in - 1)) - 1) : 0); };
    auto hi_of = [](int i_, int in, int out) { return in == out ? i_ : (in > 1 ? min(out - 1, (int)(((int64_t)(i_ + 1) * (out - 1)) / (in - 1)) + 1) : out - 1); };
    float acc = 0.f;
    for (int od = lo_of(d, D, Do); od <= hi_of(d, D, Do); ++od) {
      const float wd = wt_of(od, d, D, Do);
      if (wd == 0.f) continue;
      for (int oh = lo_of(h, H, Ho); oh <= hi_of(h, H, Ho); ++oh) {
        const float wh = wt_of(oh, h, H, Ho);
        if (wh == 0.f) continue;
        for (int ow = lo_of(w, W, Wo); ow <= hi_of(w, W, Wo); ++ow) {
          const float ww = wt_of(ow, w, W, Wo);
          if (ww == 0.f) continue;
          acc += wd * wh * ww * Act<T>::ld(dy + ((((int64_t)n * Do + od) * Ho + oh) * Wo + ow) * ldy + c);
        }
      }
    }
    Act<T>::st(dx + i / C * ldx + c, acc);
  }
}

inline int up_grid(int64_t total) {
  int64_t g = (total + UP_THREADS - 1) / UP_THREADS;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

int up_check(const char* what, const void* a, const void* b, int64_t lda, int64_t ldb, int n, int d, int h, int w, int c, int sd) {
  FPLX_REQUIRE(a && b, FPLX_E_NULL, "%s: null pointer", what);
  FPLX_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && c > 0 && lda >= c && ldb >= c && (sd == 1 || sd == 2), FPLX_E_BADSHAPE,
               "%s: bad shape", what);
  return FPLX_OK;
}

}  // namespace

extern "C" {

int fplx_upsample2_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c, int dt, int sd,
                       fplx_stream_t stream) {
  const int rc = up_check("upsample2_fwd", x, y, ldx, ldy, n, d, h, w, c, sd);
  if (rc != FPLX_OK) return rc;
  const int64_t total = (int64_t)n * d * sd * 4 * h * w * c;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_F32)
    upsample2_fwd_k<float><<<up_grid(total), UP_THREADS, 0, st>>>((const float*)x, ldx, (float*)y, ldy, n, d, h, w, c, sd);
  else if (dt == FPLX_BF16)
    upsample2_fwd_k<bf16_t><<<up_grid(total), UP_THREADS, 0, st>>>((const bf16_t*)x, ldx, (bf16_t*)y, ldy, n, d, h, w, c, sd);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "upsample2_fwd: dtype %d", dt);
  return fplx_check_launch("upsample2_fwd");
}

int fplx_upsample2_bwd(const void* dy, int64_t ldy, void* dx, int64_t ldx, int n, int d, int h, int w, int c, int dt, int sd,
                       fplx_stream_t stream) {
  const int rc = up_check("upsample2_bwd", dy, dx, ldy, ldx, n, d, h, w, c, sd);
  if (rc != FPLX_OK) return rc;
  const int64_t total = (int64_t)n * d * h * w * c;
  hipStream_t st = (hipStream_t)stream;
  if (dt == FPLX_F32)
    upsample2_bwd_k<float><<<up_grid(total), UP_THREADS, 0, st>>>((const float*)dy, ldy, (float*)dx, ldx, n, d, h, w, c, sd);
  else if (dt == FPLX_BF16)
    upsample2_bwd_k<bf16_t><<<up_grid(total), UP_THREADS, 0, st>>>((const bf16_t*)dy, ldy, (bf16_t*)dx, ldx, n, d, h, w, c, sd);
  else
    return fplx_fail(FPLX_E_BADDTYPE, "upsample2_bwd: dtype %d", dt);
  return fplx_check_launch("upsample2_bwd");
}

}  // extern "C"
