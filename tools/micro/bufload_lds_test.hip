// What does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) do with out-of-range lanes on gfx950,
// and is the SGPR offset part of the range check?  hipcc -O3 --offload-arch=gfx950 bufload_lds_test.hip -o bin/bufload_lds_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned* x, unsigned nbytes, unsigned soff, const unsigned* voffs, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned smem[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) smem[i] = 0xABABABABu;
  __syncthreads();
  u32x4 rsrc;
  rsrc[0] = (unsigned)(size_t)x; rsrc[1] = (unsigned)((size_t)x >> 32) & 0xFFFFu; rsrc[2] = nbytes; rsrc[3] = 0x00020000u;
  const unsigned voff = voffs[threadIdx.x];
  const unsigned dst = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(dst) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
  const int N = 4096;
  std::vector<unsigned> hx(N);
  for (int i = 0; i < N; ++i) hx[i] = 0x1000 + i;                 // word i holds 0x1000 + i
  unsigned *x, *out, *vo;
  hipMalloc(&x, N * 4); hipMalloc(&out, 1024); hipMalloc(&vo, 256);
  hipMemcpy(x, hx.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> hv(64);
  for (int l = 0; l < 64; ++l) hv[l] = l < 32 ? l * 32 : (l < 48 ? 0x80000000u : 0xFFFFFFF0u);   // in range | halo | -16
  hipMemcpy(vo, hv.data(), 256, hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 256u}) {
    const unsigned nbytes = 1024;                                  // lanes with voff >= 1024 - (soff counted ? soff : 0) are out of range
    k<<<1, 64>>>(x, nbytes, soff, vo, out);
    std::vector<unsigned> ho(256);
    hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
    printf("num_records %u soffset %u:\n", nbytes, soff);
    for (int l = 0; l < 64; l += 1) {
      const unsigned expect_word = (hv[l] + soff) / 4;
      printf("  lane %2d voff %08x -> %08x (in-range data would be %08x)%s", l, hv[l], ho[l * 4], hv[l] < 0x10000000u ? 0x1000 + expect_word : 0,
             (l % 2) ? "\n" : "");
    }
  }
  return 0;
}
