// Calibration: sustained v_mfma_f32_32x32x16_bf16 rate with register-resident operands (no memory traffic),
// NACC independent accumulators per wave, WPS waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void __launch_bounds__(512) k(const bf16x8* __restrict__ in, float* out, int iters) {
  bf16x8 a[2], b[3];
  for (int i = 0; i < 2; ++i) a[i] = in[threadIdx.x + 512 * i];
  for (int i = 0; i < 3; ++i) b[i] = in[threadIdx.x + 512 * (2 + i)];
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 1], b[j % 3], acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// same FLOPs per iteration with v_mfma_f32_16x16x32_bf16 (24 accumulators of 4 registers = the 6 x 16 above)
__global__ void __launch_bounds__(512) k16(const bf16x8* __restrict__ in, float* out, int iters) {
  bf16x8 a[2], b[3];
  for (int i = 0; i < 2; ++i) a[i] = in[threadIdx.x + 512 * i];
  for (int i = 0; i < 3; ++i) b[i] = in[threadIdx.x + 512 * (2 + i)];
  f32x4 acc[24];
  for (int j = 0; j < 24; ++j)
    for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 24; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 1], b[j % 3], acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 24; ++j)
    for (int i = 0; i < 4; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0 random data, 1 zeros
  const int threads = argc > 2 ? atoi(argv[2]) : 512;
  const int iters = 20000, nacc = 6;
  size_t n = 512 * 5 * 8;
  unsigned short* h = (unsigned short*)malloc(n * 2);
  srand(1);
  for (size_t i = 0; i < n; ++i) h[i] = mode ? 0 : (unsigned short)((rand() & 0x807f) | 0x3f00 | ((rand() & 1) << 7));
  bf16x8* din; float* dout;
  hipMalloc(&din, n * 2); hipMalloc(&dout, 256 * 8 * 512 * 4);
  hipMemcpy(din, h, n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<6><<<256, threads>>>(din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * (threads / 64) * iters * 4 * nacc * 32768.0;
    printf("32x32x16 mode %d threads %d: %.3f ms  %.1f TFLOP/s\n", mode, threads, ms, fl / ms / 1e9);
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k16<<<256, threads>>>(din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * (threads / 64) * iters * 2 * 24 * 16384.0;
    printf("16x16x32 mode %d threads %d: %.3f ms  %.1f TFLOP/s\n", mode, threads, ms, fl / ms / 1e9);
  }
  return 0;
}
