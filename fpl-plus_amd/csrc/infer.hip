// Sliding-window / flip test-time-augmentation plumbing of the Inferer on the GPU.
// Reference semantics (PyMIC/pymic/net_run_dsbn/infer_func.py): tiles start at min(k * stride, size - window) per axis,
// enumerated w outermost, then h, then d (lines 75-84); every tile's prediction is ADDED into the output in that order
// and the sum divided by the number of tiles covering the voxel (lines 96-112); tta_mode 1 averages the predictions of
// the image, its H flip, its W flip and its H+W flip, each flipped back: ((o1 + o2) + o3 + o4) / 4 (lines 199-219).
// Here all tiles of all flips are gathered into ONE batch (fplx_sw_extract), the network runs on chunks of that batch,
// and fplx_sw_merge forms every output voxel with exactly those additions in exactly that order (a gather: one thread
// per output element walks its covering tiles in tile order), so the result does not depend on how the batch was cut.
#include "common.h"

namespace {

constexpr int SW_MAX = 64;          // tiles per axis
constexpr int SW_THREADS = 256;

struct SwPlan {
  int n, c, d, h, w;                // image / output: [n][c][d][h][w]
  int wd, wh, ww;                   // window
  int nd, nh, nw;                   // tiles per axis; tile index = (iw * nh + ih) * nd + id
  int nflips;
  int flips[4];                     // bit0: W axis flipped, bit1: H axis flipped
  int sd[SW_MAX], sh[SW_MAX], sw[SW_MAX];
  int passes, chunk, nb;            // sw_merge: Monte-Carlo passes, patches per network call, patches in all (nflips tiles n)
};

// Where patch j of pass q lies in the prediction buffer (in patches): the network ran on chunks of `chunk` consecutive patches
// and wrote, per chunk, all passes one after the other ([chunks][passes][m_c]; only the last chunk is shorter).
__device__ __forceinline__ int64_t patch_slot(const SwPlan& p, int q, int64_t j) {
  const int64_t c0 = j / p.chunk * p.chunk;
  const int64_t mc = p.nb - c0 < p.chunk ? p.nb - c0 : p.chunk;
  return c0 * p.passes + q * mc + (j - c0);
}

// patches: [nflips][tiles][n][c][wd][wh][ww]
__global__ void __launch_bounds__(SW_THREADS) sw_extract_k(const float* __restrict__ image, float* __restrict__ patches, SwPlan p) {
  const int64_t per = (int64_t)p.wd * p.wh * p.ww;
  const int64_t tiles = (int64_t)p.nd * p.nh * p.nw;
  const int64_t total = (int64_t)p.nflips * tiles * p.n * p.c * per;
  for (int64_t i = (int64_t)blockIdx.x * SW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SW_THREADS) {
    int64_t r = i;
    const int x = (int)(r % p.ww); r /= p.ww;
    const int y = (int)(r % p.wh); r /= p.wh;
    const int z = (int)(r % p.wd); r /= p.wd;
    const int ch = (int)(r % p.c); r /= p.c;
    const int nn = (int)(r % p.n); r /= p.n;
    const int t = (int)(r % tiles);
    const int f = (int)(r / tiles);
    const int id = t % p.nd, ih = (t / p.nd) % p.nh, iw = t / (p.nd * p.nh);
    const int dd = p.sd[id] + z;
    int hh = p.sh[ih] + y, wp = p.sw[iw] + x;          // coordinates in the flipped image
    if (p.flips[f] & 2) hh = p.h - 1 - hh;
    if (p.flips[f] & 1) wp = p.w - 1 - wp;
    patches[i] = image[((((int64_t)nn * p.c + ch) * p.d + dd) * p.h + hh) * p.w + wp];
  }
}

// tiles of one axis that cover position q: a contiguous index range [lo, hi] (starts are non-decreasing)
__device__ __forceinline__ void cover(const int* __restrict__ s, int ns, int win, int q, int& lo, int& hi) {
  lo = ns; hi = -1;
  for (int k = 0; k < ns; ++k)
    if (s[k] <= q && q < s[k] + win) { if (k < lo) lo = k; hi = k; }
}

// patches: patch j = (f tiles + t) n + nn of pass q at patch_slot(q, j), each [c][wd][wh][ww] (c = class channels here)
// -> out [passes][n][c][d][h][w]; blockIdx.y = pass
__global__ void __launch_bounds__(SW_THREADS) sw_merge_k(const float* __restrict__ patches, float* __restrict__ out, SwPlan p) {
  const int64_t per = (int64_t)p.wd * p.wh * p.ww;
  const int64_t tiles = (int64_t)p.nd * p.nh * p.nw;
  const int64_t total = (int64_t)p.n * p.c * p.d * p.h * p.w;
  const int q = blockIdx.y;
  out += q * total;
  for (int64_t i = (int64_t)blockIdx.x * SW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * SW_THREADS) {
    int64_t r = i;
    const int x = (int)(r % p.w); r /= p.w;
    const int y = (int)(r % p.h); r /= p.h;
    const int z = (int)(r % p.d); r /= p.d;
    const int ch = (int)(r % p.c);
    const int nn = (int)(r / p.c);
    float tot = 0.f;
    for (int f = 0; f < p.nflips; ++f) {
      const int yy = (p.flips[f] & 2) ? p.h - 1 - y : y;       // where this voxel sits in the flipped image
      const int xx = (p.flips[f] & 1) ? p.w - 1 - x : x;
      int d0, d1, h0, h1, w0, w1;
      cover(p.sd, p.nd, p.wd, z, d0, d1);
      cover(p.sh, p.nh, p.wh, yy, h0, h1);
      cover(p.sw, p.nw, p.ww, xx, w0, w1);
      float acc = 0.f, cnt = 0.f;
      for (int iw = w0; iw <= w1; ++iw)
        for (int ih = h0; ih <= h1; ++ih)
          for (int id = d0; id <= d1; ++id) {
            // duplicate starts (the clamped last tile) are separate tiles in the reference's list: each adds once
            const int64_t t = ((int64_t)iw * p.nh + ih) * p.nd + id;
            const int64_t base = (patch_slot(p, q, ((int64_t)f * tiles + t) * p.n + nn) * p.c + ch) * per;
            acc += patches[base + ((int64_t)(z - p.sd[id]) * p.wh + (yy - p.sh[ih])) * p.ww + (xx - p.sw[iw])];
            cnt += 1.f;
          }
      const float o = (tiles == 1) ? acc : acc / cnt;          // no window: the prediction itself (no division)
      tot = (f == 0) ? o : tot + o;
    }
    out[i] = p.nflips > 1 ? tot / (float)p.nflips : tot;
  }
}

// The same sums where there is no window (ONE tile = the image, the FPL+ selection's whole-volume forwards) and W % 4 == 0:
// out = ((o1 + o2) + o3 + o4) / nflips with o_f the flip's prediction read back-to-front along the flipped axes - four
// consecutive w per thread as 16-byte loads (a W flip reads the mirrored 16 bytes and reverses them in registers).
__global__ void __launch_bounds__(SW_THREADS) sw_merge_whole_k(const float* __restrict__ patches, float* __restrict__ out, SwPlan p) {
  const int w4 = p.w >> 2;
  const int64_t per = (int64_t)p.d * p.h * p.w;
  const int64_t rows = (int64_t)p.n * p.c * p.d * p.h;          // rows of w4 float4
  const int64_t total4 = rows * w4;
  const int q = blockIdx.y;
  float4* __restrict__ o4 = reinterpret_cast<float4*>(out + q * rows * p.w);
  for (int64_t i = (int64_t)blockIdx.x * SW_THREADS + threadIdx.x; i < total4; i += (int64_t)gridDim.x * SW_THREADS) {
    const int x4 = (int)(i % w4);
    int64_t r = i / w4;
    const int y = (int)(r % p.h); r /= p.h;
    const int z = (int)(r % p.d); r /= p.d;
    const int ch = (int)(r % p.c);
    const int nn = (int)(r / p.c);
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int f = 0; f < p.nflips; ++f) {
      const int yy = (p.flips[f] & 2) ? p.h - 1 - y : y;
      const bool fw = p.flips[f] & 1;
      const int xx = fw ? p.w - 4 - 4 * x4 : 4 * x4;
      const int64_t base = (patch_slot(p, q, (int64_t)f * p.n + nn) * p.c + ch) * per;
      float4 v = *reinterpret_cast<const float4*>(patches + base + ((int64_t)z * p.h + yy) * p.w + xx);
      if (fw) v = make_float4(v.w, v.z, v.y, v.x);
      if (f == 0) tot = v;
      else { tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w; }
    }
    if (p.nflips > 1) {
      const float k = (float)p.nflips;
      tot.x /= k; tot.y /= k; tot.z /= k; tot.w /= k;
    }
    o4[i] = tot;
  }
}

int make_plan(SwPlan& p, int n, int c, int d, int h, int w, const int* sd, int nd, const int* sh, int nh, const int* sw, int nw,
              int wd, int wh, int ww, const int* flips, int nflips, const char* what) {
  FPLX_REQUIRE(n > 0 && c > 0 && d > 0 && h > 0 && w > 0, FPLX_E_BADSHAPE, "%s: bad shape", what);
  FPLX_REQUIRE(sd && sh && sw && flips, FPLX_E_NULL, "%s: null pointer", what);
  FPLX_REQUIRE(nd > 0 && nh > 0 && nw > 0 && nd <= SW_MAX && nh <= SW_MAX && nw <= SW_MAX, FPLX_E_BADSHAPE,
               "%s: 1..%d tiles per axis (got %d, %d, %d)", what, SW_MAX, nd, nh, nw);
  FPLX_REQUIRE(nflips >= 1 && nflips <= 4, FPLX_E_BADSHAPE, "%s: 1..4 flips", what);
  FPLX_REQUIRE(wd > 0 && wh > 0 && ww > 0 && wd <= d && wh <= h && ww <= w, FPLX_E_BADSHAPE, "%s: window larger than the image", what);
  p.n = n; p.c = c; p.d = d; p.h = h; p.w = w; p.wd = wd; p.wh = wh; p.ww = ww; p.nd = nd; p.nh = nh; p.nw = nw; p.nflips = nflips;
  for (int i = 0; i < 4; ++i) p.flips[i] = i < nflips ? flips[i] : 0;
  p.passes = 1; p.nb = nflips * nd * nh * nw * n; p.chunk = p.nb;
  const int* src[3] = {sd, sh, sw};
  int* dst[3] = {p.sd, p.sh, p.sw};
  const int cnt[3] = {nd, nh, nw}, win[3] = {wd, wh, ww}, ext[3] = {d, h, w};
  for (int a = 0; a < 3; ++a) {
    bool covered_end = false;
    for (int i = 0; i < SW_MAX; ++i) dst[a][i] = 0;
    for (int i = 0; i < cnt[a]; ++i) {
      const int s = src[a][i];
      FPLX_REQUIRE(s >= 0 && s + win[a] <= ext[a] && (i == 0 || s >= src[a][i - 1]), FPLX_E_BADSHAPE,
                   "%s: tile starts must be non-decreasing and inside the image", what);
      FPLX_REQUIRE(i == 0 ? s == 0 : s <= src[a][i - 1] + win[a], FPLX_E_BADSHAPE, "%s: tiles leave a gap", what);
      dst[a][i] = s;
      covered_end = s + win[a] == ext[a];
    }
    FPLX_REQUIRE(covered_end, FPLX_E_BADSHAPE, "%s: tiles do not reach the end of the image", what);
  }
  return FPLX_OK;
}

inline int blocks_for(int64_t total) {
  int64_t b = (total + SW_THREADS - 1) / SW_THREADS;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int fplx_sw_extract(const float* image, int n, int c, int d, int h, int w, const int* starts_d, int nd, const int* starts_h,
                    int nh, const int* starts_w, int nw, int wd, int wh, int ww, const int* flips, int nflips, float* patches,
                    fplx_stream_t stream) {
  FPLX_REQUIRE(image && patches, FPLX_E_NULL, "sw_extract: null pointer");
  SwPlan p;
  const int rc = make_plan(p, n, c, d, h, w, starts_d, nd, starts_h, nh, starts_w, nw, wd, wh, ww, flips, nflips, "sw_extract");
  if (rc != FPLX_OK) return rc;
  const int64_t total = (int64_t)nflips * nd * nh * nw * n * c * wd * wh * ww;
  sw_extract_k<<<blocks_for(total), SW_THREADS, 0, (hipStream_t)stream>>>(image, patches, p);
  return fplx_check_launch("sw_extract");
}

int fplx_sw_merge_mc(const float* patches, int passes, int chunk, int n, int c, int d, int h, int w, const int* starts_d, int nd,
                     const int* starts_h, int nh, const int* starts_w, int nw, int wd, int wh, int ww, const int* flips,
                     int nflips, float* out, fplx_stream_t stream) {
  FPLX_REQUIRE(patches && out, FPLX_E_NULL, "sw_merge: null pointer");
  SwPlan p;
  const int rc = make_plan(p, n, c, d, h, w, starts_d, nd, starts_h, nh, starts_w, nw, wd, wh, ww, flips, nflips, "sw_merge");
  if (rc != FPLX_OK) return rc;
  FPLX_REQUIRE(passes >= 1 && passes <= 65535 && chunk >= 1, FPLX_E_BADSHAPE, "sw_merge: passes %d, chunk %d", passes, chunk);
  p.passes = passes;
  p.chunk = chunk < p.nb ? chunk : p.nb;
  const int64_t total = (int64_t)n * c * d * h * w;
  const bool whole = nd * nh * nw == 1 && w % 4 == 0 && ((uintptr_t)patches % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (whole)
    sw_merge_whole_k<<<dim3(blocks_for(total / 4), passes), SW_THREADS, 0, (hipStream_t)stream>>>(patches, out, p);
  else
    sw_merge_k<<<dim3(blocks_for(total), passes), SW_THREADS, 0, (hipStream_t)stream>>>(patches, out, p);
  return fplx_check_launch("sw_merge");
}

int fplx_sw_merge(const float* patches, int n, int c, int d, int h, int w, const int* starts_d, int nd, const int* starts_h,
                  int nh, const int* starts_w, int nw, int wd, int wh, int ww, const int* flips, int nflips, float* out,
                  fplx_stream_t stream) {
  return fplx_sw_merge_mc(patches, 1, nflips * nd * nh * nw * n, n, c, d, h, w, starts_d, nd, starts_h, nh, starts_w, nw, wd, wh,
                          ww, flips, nflips, out, stream);
}

}  // extern "C"
