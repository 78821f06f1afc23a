// Stand-alone timing / stamping harness for the depth-march kernels (no torch, starts in a second):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DFPLX_STAMP] tools/micro/march_bench.hip -o gpurun_out/march_bench
//   gpurun_out/march_bench CIN COUT [N D H W] [twod]
// Includes the kernel source itself, so every template / lambda is the shipped one.  With -DFPLX_STAMP the Cin = 32 kernel
// records shader-clock stamps (s_memtime) around the phases of its depth loop; the harness prints their distribution.
#include "../../fpl-plus_amd/csrc/conv_march.hip"
#include <vector>
#include <algorithm>
#include <random>
#include <string>

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
  int cin = argc > 1 ? atoi(argv[1]) : 32, cout = argc > 2 ? atoi(argv[2]) : 32;
  int n = argc > 6 ? atoi(argv[3]) : 2, d = argc > 6 ? atoi(argv[4]) : 80, h = argc > 6 ? atoi(argv[5]) : 160, w = argc > 6 ? atoi(argv[6]) : 160;
  int twod = argc > 7 ? atoi(argv[7]) : 0;
  const int64_t V = (int64_t)n * d * h * w;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<unsigned short> hx((size_t)V * cin), hw((size_t)27 * cout * cin);
  const bool zeros = getenv("MB_ZEROS") != nullptr;
  for (auto& v : hx) v = zeros ? 0 : f2bf(nd(rng));
  for (auto& v : hw) v = f2bf(0.05f * nd(rng));
  void *x, *wp, *y; float *bias, *stats;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&wp, hw.size() * 2); hipMalloc(&y, (size_t)V * cout * 2);
  hipMalloc(&bias, cout * 4); hipMemset(bias, 0, cout * 4);
  const int rows = fplx_march_rows(n, d, h, w, cin, cout);
  hipMalloc(&stats, (size_t)rows * 2 * cout * 4);
  float* stats_arg = getenv("MB_NOSTATS") ? nullptr : stats;
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(wp, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  if (!fplx_march_ok(n, d, h, w, cin, cout)) { printf("shape not march-eligible\n"); return 1; }
#ifdef FPLX_STAMP
  long long* sb; const size_t nst = (size_t)rows * (cout / 32) * 8 * 6;
  hipMalloc(&sb, nst * 8); hipMemset(sb, 0, nst * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(fplx_stamp_buf), &sb, sizeof(sb));
#endif
  hipStream_t st; hipStreamCreate(&st);
  for (int i = 0; i < 3; ++i) fplx_march_conv3d_fwd(x, cin, wp, bias, y, cout, n, d, h, w, cin, cout, stats_arg, st, nullptr, nullptr, twod);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20;
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) fplx_march_conv3d_fwd(x, cin, wp, bias, y, cout, n, d, h, w, cin, cout, stats_arg, st, nullptr, nullptr, twod);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double fl = 2.0 * V * cin * cout * (twod ? 9 : 27);
  printf("march cin=%d cout=%d dims=%dx%dx%dx%d twod=%d blocks=%d: %.1f us  %.1f TF/s  %.0f GB/s (%s)\n", cin, cout, n, d, h, w, twod,
         rows * (cout / 32), ms * 1e3, fl / ms / 1e9, (double)V * (cin + cout) * 2 / ms / 1e6, hipGetErrorString(hipGetLastError()));
  // checksum so that variants can be compared
  std::vector<unsigned short> hy((size_t)V * cout);
  hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost);
  unsigned long long cs = 0; for (size_t i = 0; i < hy.size(); i += 7) cs = cs * 1315423911ull + hy[i];
  printf("checksum %llx\n", cs);
  if (const char* f = getenv("MB_DUMP")) { FILE* fp = fopen(f, "wb"); fwrite(hy.data(), 2, hy.size(), fp); fclose(fp);
    std::vector<float> hst((size_t)rows * 2 * cout); hipMemcpy(hst.data(), stats, hst.size() * 4, hipMemcpyDeviceToHost);
    std::string g = std::string(f) + ".stats"; fp = fopen(g.c_str(), "wb"); fwrite(hst.data(), 4, hst.size(), fp); fclose(fp); }
#ifdef FPLX_STAMP
  std::vector<long long> hs(nst);
  hipMemcpy(hs.data(), sb, nst * 8, hipMemcpyDeviceToHost);
  double s[6] = {0, 0, 0, 0, 0, 0}; size_t cnt = 0;
  std::vector<long long> tot;
  for (size_t i = 0; i < nst; i += 6) { if (hs[i + 4] == 0) continue; for (int k = 0; k < 6; ++k) s[k] += hs[i + k]; tot.push_back(hs[i + 3]); ++cnt; }
  std::sort(tot.begin(), tot.end());
  if (getenv("FPLX_MARCH32_V2") == nullptr || atoi(getenv("FPLX_MARCH32_V2")) != 0)
    printf("v2 stamps per wave (avg over %zu waves): fast steps %.1f at %.0f cycles each | flagged steps total %.0f | rotations total %.0f | loop total %.0f | clock %.3f GHz\n",
           cnt, s[4] / cnt, s[0] / s[4], s[1] / cnt, s[2] / cnt, s[3] / cnt, s[3] / s[5] * 0.1);
  else
  printf("per wave (avg over %zu waves): steps %.1f | cycles per step: compute %.0f  vmcnt-wait %.0f  barrier %.0f | loop total %.0f (median %lld, max %lld)  in-kernel clock %.3f GHz\n",
         cnt, s[4] / cnt, s[0] / s[4], s[1] / s[4], s[2] / s[4], s[3] / cnt, tot[tot.size() / 2], tot.back(), s[3] / s[5] * 0.1);
#endif
  return 0;
}
