// Stand-alone timing / stamping harness for conv_fwd_brick (no torch):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DFPLX_STAMP] tools/micro/brick_bench.hip -o gpurun_out/brick_bench
//   gpurun_out/brick_bench CIN COUT N D H W [stats]
// Includes the kernel source itself.  With -DFPLX_STAMP the kernel records shader-clock stamps (s_memtime) around the phases
// of its stage loop (first-half MFMAs | vmcnt wait | barrier | DMA issue | second-half MFMAs | write-out).
#include "../../fpl-plus_amd/csrc/conv_brick.hip"
#include <vector>
#include <algorithm>
#include <random>
#include <string>

// the tuning table of the library (conv_generic.hip) is not linked here: every knob reads its default
int64_t fplx_knob_values[FPLX_K_COUNT] = {
#define FPLX_KNOB_DEF(id, key, def) def,
    FPLX_KNOB_LIST(FPLX_KNOB_DEF)
#undef FPLX_KNOB_DEF
};
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
  const int cin = atoi(argv[1]), cout = atoi(argv[2]), n = atoi(argv[3]), d = atoi(argv[4]), h = atoi(argv[5]), w = atoi(argv[6]);
  const int want_stats = argc > 7 ? atoi(argv[7]) : 0;
  const int64_t V = (int64_t)n * d * h * w;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<unsigned short> hx((size_t)V * cin), hw((size_t)27 * cout * cin);
  for (auto& v : hx) v = f2bf(nd(rng));
  for (auto& v : hw) v = f2bf(0.05f * nd(rng));
  void *x, *wp, *y; float *bias, *stats, *partial;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&wp, hw.size() * 2); hipMalloc(&y, (size_t)V * cout * 2);
  hipMalloc(&bias, cout * 4); hipMemset(bias, 0, cout * 4);
  int geo, ksplit, bricks;
  if (!fplx_brick_plan(n, d, h, w, cin, cout, &geo, &ksplit, &bricks)) { printf("shape not brick-eligible\n"); return 1; }
  hipMalloc(&stats, (size_t)bricks * 2 * cout * 4 + 1024);
  hipMalloc(&partial, (size_t)ksplit * V * cout * 4);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(wp, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
#ifdef FPLX_STAMP
  long long* sb; const size_t nst = (size_t)4096 * 8 * 4 * 10;
  hipMalloc(&sb, nst * 8); hipMemset(sb, 0, nst * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(fplx_brick_stamp_buf), &sb, sizeof(sb));
#endif
  hipStream_t st; hipStreamCreate(&st);
  float* sarg = (want_stats && ksplit == 1) ? stats : nullptr;
  for (int i = 0; i < 3; ++i) fplx_brick_conv3d_fwd_ex(x, cin, wp, bias, y, cout, n, d, h, w, cin, cout, sarg, partial, geo, ksplit, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20;
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) fplx_brick_conv3d_fwd_ex(x, cin, wp, bias, y, cout, n, d, h, w, cin, cout, sarg, partial, geo, ksplit, st);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double fl = 2.0 * V * cin * cout * 27;
  printf("brick cin=%d cout=%d dims=%dx%dx%dx%d geo=%d ksplit=%d bricks=%d: %.1f us  %.1f TF/s (%s)\n", cin, cout, n, d, h, w, geo, ksplit,
         bricks, ms * 1e3, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
  std::vector<unsigned short> hy((size_t)V * cout);
  hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost);
  unsigned long long cs = 0; for (size_t i = 0; i < hy.size(); i += 7) cs = cs * 1315423911ull + hy[i];
  printf("checksum %llx\n", cs);
#ifdef FPLX_STAMP
  std::vector<long long> hs(nst);
  hipMemcpy(hs.data(), sb, nst * 8, hipMemcpyDeviceToHost);
  double s[10] = {0}, l[10] = {0}; size_t cnt = 0, lcnt = 0;
  for (size_t i = 0; i < nst; i += 10) {
    if (hs[i + 9] == 1) { for (int k = 0; k < 10; ++k) s[k] += hs[i + k]; ++cnt; }
    if (hs[i + 9] == 2) { for (int k = 0; k < 10; ++k) l[k] += hs[i + k]; ++lcnt; }      // loader waves of conv_fwd_brick_lw
  }
  if (lcnt) printf("loader waves (avg over %zu): stages %.1f | cycles per stage: issue %.0f  vmcnt wait %.0f  barrier %.0f | total %.0f cycles\n",
                   lcnt, l[6] / lcnt, l[0] / l[6], l[1] / l[6], l[2] / l[6], l[7] / lcnt);
  const double ns = s[6];
  printf("per wave (avg over %zu waves): stages %.1f | cycles per stage: first half %.0f  vmcnt wait %.0f  barrier %.0f  DMA issue %.0f  second half %.0f | "
         "write-out per wave %.0f | kernel total %.0f cycles, in-kernel clock %.3f GHz\n",
         cnt, ns / cnt, s[0] / ns, s[1] / ns, s[2] / ns, s[3] / ns, s[4] / ns, s[5] / cnt, s[7] / cnt, s[7] / s[8] * 0.1);
#endif
  return 0;
}
