"""DRAFT (not part of the build): source patch that rewrites conv_fwd_march64 (fpl-plus_amd/csrc/conv_march.hip) to
v_mfma_f32_16x16x32_bf16 tiles - the shape that sustains a ~14 % higher clock (tools/micro/mfma_peak.hip).  It contains
the pieces that are settled: the 64-byte-row swizzle 2*((row/4)%2) (conflict-free for the 16-row fragment reads), the
Tile16 accumulator layout, the write-out mapping and the 24-slot stage schedule.  Status: hipcc (ROCm 7.2) aborts on it
with "Illegal instruction detected: Operand has incorrect register class" (459 VGPR+AGPR).  Bisected: the error goes
away when ANY ONE of {side work in the gaps, the ds_write_b16 staging, the inline-asm global stores} is removed, and an
f32x16-with-subvectors formulation fails the same way - it looks like a register-class assignment problem under
pressure around the inline-asm operands, not a problem of the MFMA code itself.  Next round: lower the pressure (one
fragment buffer set less) or issue the stores without inline asm.  Run from fpl-plus_amd/csrc to apply."""
p='conv_march.hip'
s=open(p).read()
a=s.index('// one half-slab -> the three output depths it touches.  FIRST: the kd = 0 accumulators start from zero.\n// side(q, g) runs in gap g (0..5) of stage q (0..17)')
b=s.index('struct MarchCfg { int tilesH, tilesW, dsegs, dlen, nblk; };')
new=r'''// 16 x 16 x 32 MFMA tiles of one 32-voxel x 32-channel output tile: t[mh][nh] = voxels 16 mh .., channels 16 nh ..
//   A fragment: lane l holds A[row l&15][k = 8*(l>>4) + j];  B: B[k = 8*(l>>4) + j][col l&15]
//   D: lane l holds D[row 4*(l>>4) + i][col l&15], i = 0..3
// (the chip sustains a ~14 % higher clock on this shape than on 32x32x16 in an MFMA-bound loop: tools/micro/mfma_peak.hip)
struct Tile16 { f32x4 t[2][2]; };

// one half-slab (32 channels = one K = 32 step per tap) -> the three output depths it touches.
// FIRST: the kd = 0 accumulators start from zero.  A stage = one (kh, kw) tap pair p: 10 fragments, 24 MFMAs.
// side(q, g), q in 0..17, g in 0..5 (the schedule of the 32x32x16 version: 108 slots per half-step) runs in every
// second MFMA gap.
template <int MASK, bool FIRST, class Side>
__device__ __forceinline__ void march64_half(const char* __restrict__ sl, const char* __restrict__ wh, int wave, int r16,
                                             int kq, Tile16& A00, Tile16& A01, Tile16& A10, Tile16& A11, Tile16& A20,
                                             Tile16& A21, Side&& side) {
  bf16x8 fa[2][4], fb[2][6];                      // [buffer][m * 2 + mh], [buffer][kd * 2 + nh]
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  int vb = wave * 2 * MG64::SW + r16, rb = r16;
  asm volatile("" : "+v"(vb), "+v"(rb));         // lane bases re-derived per half-step (no hoisted address zoo)
  // weight rows: swz((tap * 32 + 16 nh + r16)) == swz(r16): a fragment is an immediate offset from one lane base
  const char* wl = wh + rb * MG64::ROWB + ((kq ^ MG64::swz(rb)) << 4);
  auto load_a = [&](int p, int f) {               // f = m * 2 + mh
    const int kh = p / 3, kw = p % 3, m = f >> 1, mh = f & 1;
    const int vox = vb + (m + kh) * MG64::SW + kw + mh * 16;
    return *reinterpret_cast<const bf16x8*>(sl + vox * MG64::ROWB + ((kq ^ MG64::swz(vox)) << 4));
  };
  auto load_b = [&](int p, int f) {               // f = kd * 2 + nh
    const int kd = f >> 1, nh = f & 1;
    return *reinterpret_cast<const bf16x8*>(wl + ((kd * 9 + p) * 32 + nh * 16) * MG64::ROWB);
  };
#pragma unroll
  for (int f = 0; f < 4; ++f) fa[0][f] = load_a(0, f);
#pragma unroll
  for (int f = 0; f < 6; ++f)
    if ((MASK >> (f >> 1)) & 1) fb[0][f] = load_b(0, f);
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    const int b = p & 1, nb = b ^ 1;
    // 24 MFMA slots: slot = (m * 3 + kd) * 4 + mh * 2 + nh; before each: one fragment of tap pair p + 1 (10 of them)
    // and, in every second slot, the side work
#pragma unroll
    for (int slot = 0; slot < 24; ++slot) {
      const int m = slot / 12, kd = (slot / 4) % 3, mh = (slot >> 1) & 1, nh = slot & 1;
      if (p + 1 < 9) {
        if (slot < 4) fa[nb][slot] = load_a(p + 1, slot);
        else if (slot < 10) { if ((MASK >> ((slot - 4) >> 1)) & 1) fb[nb][slot - 4] = load_b(p + 1, slot - 4); }
      }
      if ((slot & 1) == 0) { const int F = (p * 24 + slot) >> 1; side(F / 6, F % 6); }
      __builtin_amdgcn_sched_barrier(0);
      if ((MASK >> kd) & 1) {
        Tile16& A = kd == 0 ? (m ? A01 : A00) : (kd == 1 ? (m ? A11 : A10) : (m ? A21 : A20));
        const bool fresh = FIRST && kd == 0 && p == 0;
        A.t[mh][nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[b][m * 2 + mh], fb[b][kd * 2 + nh],
                                                              fresh ? zero : A.t[mh][nh], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__global__ void __launch_bounds__(MG64::THREADS)
conv_fwd_march64(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                 const float* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy, int N, int D, int H, int W,
                 int Cout, float* __restrict__ stats, int tilesH, int tilesW, int dsegs, int dlen,
                 const bf16_t* __restrict__ x1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabs = smem;                                        // [2 channel halves][SLAB][32]
  char* wbuf = smem + 2 * MG64::SLAB_BYTES;                  // [2 channel halves][27][32 co][32 ci]
  float* bias_s = reinterpret_cast<float*>(wbuf + 2 * MG64::WH_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int seg = b % dsegs; b /= dsegs;
  const int tw = b % tilesW; b /= tilesW;
  const int th = b % tilesH; b /= tilesH;
  const int n = __builtin_amdgcn_readfirstlane(b);
  const int h0 = __builtin_amdgcn_readfirstlane(th * MG64::FH), w0 = __builtin_amdgcn_readfirstlane(tw * MG64::FW);
  const int d0 = __builtin_amdgcn_readfirstlane(seg * dlen);
  const int d1 = (d0 + dlen < D) ? d0 + dlen : D;
  const int n0 = blockIdx.y * 32;

  auto lds_dma = [&](const void* g, const char* l) {         // see conv_fwd_march32
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((const __attribute__((address_space(3))) char*)l));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto block_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  constexpr int NPIECE = MG64::NPIECE;
  // per lane and piece: byte offset of the source chunk (channel half 0) inside one depth slice of x, -1 = zero,
  // -2 = lane past the end of the slab (the last piece is 16 lanes wide)
  int soff[NPIECE];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int i = (wave + 4 * k) * 64 + lane;
    const int vox = i >> 2, c = (i & 3) ^ MG64::swz(vox);
    const int hh = vox / MG64::SW + h0 - 1, ww = vox % MG64::SW + w0 - 1;
    const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
    soff[k] = i >= MG64::SLAB_CHUNKS ? -2 : (in ? (int)((((int64_t)hh * W + ww) * ldx + c * 8) * 2) : -1);
  }
  const int64_t xslice = (int64_t)H * W * ldx * 2;
  const char* xn = reinterpret_cast<const char*>(x) + (int64_t)n * D * xslice;
  // channel half 1: the next 32 channels of x, or a second tensor (torch.cat([x, x1], 1) never materialised)
  const char* xn1 = reinterpret_cast<const char*>(x1 ? x1 : x + 32) + (int64_t)n * D * xslice;
  auto slab_piece = [&](int s, int hf, int k) {             // piece k of channel half hf of slab s -> slot hf
    if (wave + 4 * k < MG64::SLAB_DMA && soff[k] != -2) {
      const char* xs = (hf ? xn1 : xn) + s * xslice;        // uniform
      const void* src = soff[k] >= 0 ? (const void*)(xs + (unsigned)soff[k]) : (const void*)fplx_zero16;
      lds_dma(src, slabs + hf * MG64::SLAB_BYTES + (wave + 4 * k) * 1024);
    }
  };

  Tile16 K0a, K0b, K1a, K1b, K2a, K2b, Ra, Rb;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        K0a.t[u][v][i] = K0b.t[u][v][i] = K1a.t[u][v][i] = K1b.t[u][v][i] = K2a.t[u][v][i] = K2b.t[u][v][i] =
            Ra.t[u][v][i] = Rb.t[u][v][i] = 0.f;

  // prologue: both halves of the first slab, the resident weights (source-side swizzle), bias
  if (d0 - 1 >= 0) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) { slab_piece(d0 - 1, 0, k); }
  }
  for (int j = wave; j < 2 * 27 * 32 * MG64::CH / 64; j += 4) {
    const int i = j * 64 + lane;                             // chunk index over [half][tap][co][4 chunks]
    const int hf = i / (27 * 32 * MG64::CH), ii = i % (27 * 32 * MG64::CH);
    const int row = ii >> 2, c = (ii & 3) ^ MG64::swz(row);
    lds_dma(wp + ((int64_t)(row >> 5) * Cout + n0 + (row & 31)) * 64 + hf * 32 + c * 8, wbuf + j * 1024);
  }
  if (tid < 32) bias_s[tid] = bias ? bias[n0 + tid] : 0.f;
  dma_wait();
  block_sync();

  // write-out: lane = (channel r16 of both 16-channel halves, voxel quad kq); a Tile16 element e = mh * 8 + nh * 4 + i
  // is voxel 16 mh + 4 kq + i, channel 16 nh + r16
  const float bv0 = bias_s[r16], bv1 = bias_s[16 + r16];
  float ssum0 = 0.f, qsum0 = 0.f, ssum1 = 0.f, qsum1 = 0.f;
  char* stg = reinterpret_cast<char*>(bias_s + 32) + wave * MG64::STAGE_BYTES;
  char* stg_w = stg + (4 * kq) * 64 + r16 * 2;
  const char* stg_r = stg + lane * 16;
  unsigned wmask = 0;                                       // bit mh * 4 + i: that voxel of the w-row is inside the volume
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (w0 + (e >> 2) * 16 + 4 * kq + (e & 3) < W) wmask |= 1u << e;
  const bool hok0 = h0 + wave * 2 < H, hok1 = h0 + wave * 2 + 1 < H;
  const unsigned ldy2 = (unsigned)ldy * 2u;
  char* yn = reinterpret_cast<char*>(y) + ((((int64_t)n * D * H + (h0 + wave * 2)) * W + w0) * ldy + n0) * 2;
  const int64_t yslice = (int64_t)H * W * ldy * 2;
  const unsigned soffb = (unsigned)(lane >> 2) * ldy2 + (unsigned)(lane & 3) * 16u;
  const bool sok0 = w0 + (lane >> 2) < W, sok1 = w0 + (lane >> 2) + 16 < W;
  auto retire_elem = [&](Tile16& A, int m, int e) {
    const int mh = e >> 3, nh = (e >> 2) & 1, i = e & 3;
    const float ov = A.t[mh][nh][i] + (nh ? bv1 : bv0);
    *reinterpret_cast<bf16_t*>(stg_w + (mh * 16 + i) * 64 + nh * 32) = (bf16_t)ov;
    if ((m ? hok1 : hok0) && ((wmask >> (mh * 4 + i)) & 1u)) {
      if (nh) { ssum1 += ov; qsum1 = fmaf(ov, ov, qsum1); }
      else { ssum0 += ov; qsum0 = fmaf(ov, ov, qsum0); }
    }
  };
  auto retire_flush = [&](int m, int o) {
    if (m ? hok1 : hok0) {
      unsigned l2 = ldy2;
      asm volatile("" : "+s"(l2));
      char* rowp = yn + o * yslice + (unsigned)(m * W) * l2;
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(stg_r);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(stg_r + 1024);
      if (sok0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(rowp + soffb), "v"(v0) : "memory");
      if (sok1) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(rowp + 16u * l2 + soffb), "v"(v1) : "memory");
    }
  };

  // half-step (t, hf): slab s = d0 - 1 + t, channel half hf, out of slot hf.  In its gaps: the DMA of the NEXT
  // half-slab (the other half of s, or half 0 of s + 1) into the other slot; during half 0 also the write-out of the
  // depth that completed in step t - 1 (R): stages 0-3 M-tile 0 -> LDS tile, flush, stages 4-7 M-tile 1, flush.
  const int nd = d1 - d0;                         // >= 2 (march_cfg)
  for (int t = 0; t < nd + 2; ++t) {
    const int s = d0 - 1 + t;
    const bool live = s >= 0 && s < D;             // a padding slab contributes nothing
    const bool wout = t >= 3;
    const int o = s - 2;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      // next half-slab: (s, 1) after (s, 0); (s + 1, 0) after (s, 1)
      const int ns = hf == 0 ? s : s + 1, nh_ = hf ^ 1;
      const bool fetch = ns >= 0 && ns < D && ns <= d1 && (hf == 0 ? live : true);
      auto side = [&](int q, int g) {
        if (g == 5 && q < NPIECE && fetch) slab_piece(ns, nh_, q < NPIECE ? q : 0);
        if (hf == 0 && wout && q < 8) {
          if (g < 4) {
            if (q < 4) retire_elem(Ra, 0, 4 * q + g);
            else retire_elem(Rb, 1, 4 * (q - 4) + g);
          }
          if (g == 4 && q == 3) retire_flush(0, o);
          if (g == 4 && q == 7) retire_flush(1, o);
        }
      };
      const char* sl = slabs + hf * MG64::SLAB_BYTES;
      const char* wh = wbuf + hf * MG64::WH_BYTES;
#define M64_STEP(MASK)                                                                                              \
  do {                                                                                                              \
    if (hf == 0) march64_half<MASK, true>(sl, wh, wave, r16, kq, K0a, K0b, K1a, K1b, K2a, K2b, side);               \
    else march64_half<MASK, false>(sl, wh, wave, r16, kq, K0a, K0b, K1a, K1b, K2a, K2b, side);                      \
  } while (0)
      if (live) {
        if (t == 0) M64_STEP(1);
        else if (t == 1) M64_STEP(3);
        else if (t < nd) M64_STEP(7);
        else if (t == nd) M64_STEP(6);
        else M64_STEP(4);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int g = 0; g < 6; ++g) side(q, g);
        if (hf == 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
              for (int i = 0; i < 4; ++i) K0a.t[u][v][i] = K0b.t[u][v][i] = 0.f;
        }
      }
#undef M64_STEP
      dma_wait();
      block_sync();
    }
    Ra = K2a; Rb = K2b; K2a = K1a; K2b = K1b; K1a = K0a; K1b = K0b;
  }
  // drain: the last depth completed in the final step
#pragma unroll
  for (int e = 0; e < 16; ++e) retire_elem(Ra, 0, e);
  retire_flush(0, d1 - 1);
#pragma unroll
  for (int e = 0; e < 16; ++e) retire_elem(Rb, 1, e);
  retire_flush(1, d1 - 1);

  if (stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][32]; the slabs are dead
    float v4[4] = {ssum0, ssum1, qsum0, qsum1};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v4[k] += __shfl_xor(v4[k], 16, 64); v4[k] += __shfl_xor(v4[k], 32, 64); }
    if (lane < 16) {
      red[(wave * 2 + 0) * 32 + r16] = v4[0]; red[(wave * 2 + 0) * 32 + 16 + r16] = v4[1];
      red[(wave * 2 + 1) * 32 + r16] = v4[2]; red[(wave * 2 + 1) * 32 + 16 + r16] = v4[3];
    }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      float tt = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) tt += red[(wv * 2 + which) * 32 + c];
      stats[((int64_t)blockIdx.x * 2 + which) * Cout + n0 + c] = tt;
    }
  }
}

'''
s=s[:a]+new+s[b:]
# new swizzle for MG64
i0=s.index('struct MG64 {')
i1=s.index('};', i0)
blk=s[i0:i1]
assert 'return (row >> 2) & 3;' in blk
blk=blk.replace("  static __device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }","  // 16-byte-chunk XOR swizzle of the 64-byte rows: conflict-free for the 16-row x 4-chunk fragment reads of the\n  // 16x16x32 MFMA at every row alignment (found by exhaustive search over the ds_read_b128 lane groups)\n  static __device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }")
s=s[:i0]+blk+s[i1:]
if 'typedef __attribute__((ext_vector_type(4))) float f32x4;' not in s:
    s=s.replace("typedef __attribute__((ext_vector_type(16))) float f32x16;","typedef __attribute__((ext_vector_type(16))) float f32x16;\ntypedef __attribute__((ext_vector_type(4))) float f32x4;",1)
open(p,'w').write(s)
