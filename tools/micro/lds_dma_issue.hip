// What does ONE LDS-DMA wave-instruction cost the wave that issues it?  (tools/micro/brick_bench.hip found about 120 cycles per
// `buffer_load_dwordx4 ... offen lds` piece in conv_fwd_brick, wherever the piece sits.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/lds_dma_issue.hip -o tools/micro/bin/lds_dma_issue
// One block of 4 waves per CU (64 KB LDS each, like the production kernels), every wave issues NP pieces with GAP filler
// v_mfma between two pieces and stamps s_memtime around each piece; the source is a 1-MB buffer (L2 hits after the first pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int GAP>
__global__ void __launch_bounds__(256) k(const char* __restrict__ src, long long* __restrict__ out, float* sink, int np, int waves_on) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
  u32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)src);
  rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)src >> 32) & 0xFFFFu);
  rsrc[2] = 1u << 20;
  rsrc[3] = 0x00020000u;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)1.f; }
  long long tot = 0, mx = 0;
  const long long t_begin = __builtin_amdgcn_s_memtime();
  if (wave < waves_on) {
    for (int p = 0; p < np; ++p) {
      const unsigned vo = (unsigned)(((p * 4 + wave) * 1024 + lane * 16) & ((1 << 20) - 1));
      const unsigned dst = lds0 + (unsigned)(((p & 15) * 4 + wave) * 1024);
      const unsigned so = 0;
      if (MODE == 0) {              // the production statement
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(rsrc), "s"(so), "s"(dst) : "memory");
      } else if (MODE == 1) {       // global_load_lds_dwordx4
        const char* g = src + vo;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(dst) : "memory");
      } else if (MODE == 2) {       // four dword pieces (256 B each) instead of one dwordx4
        for (int q = 0; q < 4; ++q) {
          const unsigned vo4 = (unsigned)(((p * 4 + wave) * 1024 + q * 256 + lane * 4) & ((1 << 20) - 1));
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" :: "v"(vo4), "s"(rsrc), "s"(so), "s"(dst + q * 256) : "memory");
        }
      } else if (MODE == 3) {       // a plain dwordx4 load to registers (no LDS), for comparison
        u32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
        asm volatile("" :: "v"(v));
      }
      if (MODE == 4) { }            // no piece at all: the loop's own cost
      if (MODE == 5 || MODE == 6) {  // the production statement as conv_fwd_brick issues it: scalar offset through VALU + readfirstlane;
                                     // 6: source rows of 64 bytes at a 128-byte stride (a weight piece: 16 rows of 4 lanes)
        unsigned sov = (unsigned)(p * 64) + (unsigned)lane * 0u;
        asm volatile("" : "+v"(sov));
        const unsigned so5 = __builtin_amdgcn_readfirstlane(sov);
        const unsigned vo5 = MODE == 6 ? (unsigned)((((p * 4 + wave) * 16 + (lane >> 2)) * 128 + (lane & 3) * 16) & ((1 << 20) - 1)) : vo;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo5), "s"(rsrc), "s"(so5), "s"(dst) : "memory");
      }
#pragma unroll
      for (int gq = 0; gq < GAP; ++gq) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if ((p & 15) == 15) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t_all = __builtin_amdgcn_s_memtime() - t_begin;
  if (lane == 0) { long long* o = out + (blockIdx.x * 4 + wave) * 4; o[0] = tot; o[1] = mx; o[2] = t_all; o[3] = wave < waves_on; }
  if (acc[0] == 12345.f) sink[0] = acc[1];
}

template <int MODE, int GAP>
static void run(const char* name, const char* src, long long* out, float* sink, int np, int waves_on) {
  hipFuncSetAttribute((const void*)k<MODE, GAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int it = 0; it < 3; ++it) k<MODE, GAP><<<256, 256, 65536>>>(src, out, sink, np, waves_on);
  hipDeviceSynchronize();
  std::vector<long long> h(256 * 4 * 4);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double tot = 0, all = 0; long long mx = 0; int n = 0;
  for (size_t i = 0; i < h.size(); i += 4) if (h[i + 3]) { tot += h[i]; all += h[i + 2]; mx = std::max(mx, h[i + 1]); ++n; }
  printf("%-44s gap %2d MFMAs, %d waves issuing: loop %8.0f cycles = %6.1f per iteration, %6.1f beyond the MFMAs' %d (%s)\n", name, GAP,
         waves_on, all / n, all / n / np, all / n / np - 32.0 * GAP, 32 * GAP, hipGetErrorString(hipGetLastError()));
}

int main() {
  char* src; long long* out; float* sink;
  hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20); hipMalloc(&out, 256 * 4 * 4 * 8); hipMalloc(&sink, 64);
  const int np = 256;
  for (int w = 4; w >= 1; w -= 3) {
    run<4, 0>("nothing", src, out, sink, np, w);
    run<4, 4>("nothing", src, out, sink, np, w);
    run<0, 0>("buffer_load_dwordx4 offen lds (+M0 save)", src, out, sink, np, w);
    run<0, 1>("buffer_load_dwordx4 offen lds (+M0 save)", src, out, sink, np, w);
    run<0, 2>("buffer_load_dwordx4 offen lds (+M0 save)", src, out, sink, np, w);
    run<0, 4>("buffer_load_dwordx4 offen lds (+M0 save)", src, out, sink, np, w);
    run<0, 8>("buffer_load_dwordx4 offen lds (+M0 save)", src, out, sink, np, w);
    run<5, 4>("... with the scalar offset via readfirstlane", src, out, sink, np, w);
    run<6, 4>("... and 64-byte rows at a 128-byte stride", src, out, sink, np, w);
    run<6, 2>("... and 64-byte rows at a 128-byte stride", src, out, sink, np, w);
    run<1, 0>("global_load_lds_dwordx4", src, out, sink, np, w);
    run<1, 4>("global_load_lds_dwordx4", src, out, sink, np, w);
    run<2, 0>("4 x buffer_load_dword offen lds", src, out, sink, np, w);
    run<2, 4>("4 x buffer_load_dword offen lds", src, out, sink, np, w);
    run<3, 0>("buffer_load_dwordx4 offen (to registers)", src, out, sink, np, w);
    run<3, 4>("buffer_load_dwordx4 offen (to registers)", src, out, sink, np, w);
  }
  return 0;
}
