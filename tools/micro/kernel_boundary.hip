// What does the boundary behind a kernel that WROTE a large tensor cost the next kernel of the stream?  (r05 two-stream timeline:
// 6-7 us of device idle time behind every convolution / fill kernel, 0.1 us behind a kernel that wrote a few KB.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/kernel_boundary.hip -o tools/micro/bin/kernel_boundary
// W writes `mb` MB with 16-byte stores (mode 0: plain, 1: nontemporal, 2: sc0 sc1 = write-through), T is a one-wave kernel
// behind it.  Chains of N launches, HIP events around the chain: per-iteration time of [W], [W, T] and [T].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ void __launch_bounds__(256) W(u32x4* __restrict__ dst, size_t n16, unsigned v) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    u32x4 val = {v, v + 1, v + 2, (unsigned)i};
    if (MODE == 0) dst[i] = val;
    else if (MODE == 1) __builtin_nontemporal_store(val, dst + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(val) : "memory");
  }
}
__global__ void T(const unsigned* __restrict__ src, unsigned* __restrict__ out) {
  if (threadIdx.x == 0) out[0] = src[0] + 1;
}

template <int MODE>
static void run(u32x4* buf, size_t n16, unsigned* out, int N, const char* name, double mb) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float ms[3];
  for (int chain = 0; chain < 3; ++chain) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a, 0);
      for (int i = 0; i < N; ++i) {
        if (chain != 2) W<MODE><<<2048, 256>>>(buf, n16, (unsigned)i);
        if (chain != 0) T<<<1, 64>>>((const unsigned*)buf, out);
      }
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms[chain], a, b);
    }
  }
  printf("%-28s %6.0f MB   [W] %7.2f us (%.2f TB/s)   [W, T] %7.2f us   [T] %5.2f us   boundary behind W: %+.2f us\n", name, mb,
         ms[0] * 1e3 / N, mb / 1e6 / (ms[0] * 1e-3 / N), ms[1] * 1e3 / N, ms[2] * 1e3 / N, (ms[1] - ms[0] - ms[2]) * 1e3 / N);
}

int main(int argc, char** argv) {
  const int N = 100;
  unsigned* out;
  hipMalloc(&out, 64);
  for (double mb : {0.25, 4.0, 32.0, 262.0}) {
    const size_t n16 = (size_t)(mb * 1e6 / 16);
    u32x4* buf;
    hipMalloc(&buf, n16 * 16);
    run<0>(buf, n16, out, N, "plain stores", mb);
    run<1>(buf, n16, out, N, "nontemporal stores", mb);
    run<2>(buf, n16, out, N, "sc0 sc1 (write-through)", mb);
    hipFree(buf);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
