// Philox4x32-10 (Salmon et al., SC'11) - the counter-based dropout stream of fplx.
// keep(i) for the flat NDHWC element index i of a dropout site:
//   word (i & 3) of philox(counter = (i >> 2, 0, stream_id, 0), key = (seed_lo, seed_hi)) >= floor(p * 2^32)
// oracle/np_ref.py:philox_keep_mask reproduces this bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Philox4 { uint32_t v[4]; };

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)c0 * 0xD2511F53u;
    const uint64_t p1 = (uint64_t)c2 * 0xCD9E8D57u;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
