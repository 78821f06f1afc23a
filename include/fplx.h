/*
 * fplx.h - C ABI of libfplx.so: the MI355X (gfx950 / CDNA4) hot path of FPL+
 * (3D U-Net with domain-specific BatchNorm: forward / backward, segmentation losses,
 * pseudo-label uncertainty filter, fused Adam).
 *
 * The reference (HiLab-git/FPL-plus) is pure Python and has no FFI; every device kernel it
 * runs comes from torch/ATen.  Each entry point below therefore cites the reference call
 * site (path under /root/reference/PyMIC/pymic unless noted) whose ATen op it replaces.
 *
 * Conventions
 *  - plain C: raw device pointers + sizes.  The CALLER owns every buffer (including
 *    workspaces); the library allocates nothing.  Its only mutable state is the table of
 *    integer tuning knobs behind fplx_set_tuning (A/B measurements; the defaults are the shipped
 *    configuration and every knob selects among kernels computing the same function - some in another order of fp32 additions, only which kernel computes it) and the calling
 *    thread's last error message.  The library never reads the environment.
 *  - only the functions declared here are exported (the library is built with
 *    -fvisibility=hidden; tests/test_host_cpu.py compares `nm -D` with this file).
 *  - activations are NDHWC ("channels last"): element (n,d,h,w,c) of a tensor with voxel
 *    stride `ld` (elements, >= C) lives at ((n*D+d)*H+h)*W+w)*ld + c.  ld > C addresses a
 *    channel slice of a wider buffer (the skip/up concat of UpBlock is never materialised).
 *  - `dt` selects the activation storage type: FPLX_F32 or FPLX_BF16.  Parameters,
 *    statistics, reductions and gradients of parameters are always fp32.
 *  - every launch is asynchronous on `stream` (a hipStream_t); no call synchronises.
 *  - return value: 0 on success, negative FPLX_E_* otherwise; fplx_last_error() gives the
 *    message of the calling thread's last failure.
 *  - reductions use fixed-order two-stage trees (no float atomics): bitwise reproducible.
 */
#ifndef FPLX_H
#define FPLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef void* fplx_stream_t; /* hipStream_t */

enum { FPLX_F32 = 0, FPLX_BF16 = 1 };

enum {
  FPLX_OK = 0,
  FPLX_E_BADSHAPE = -1,   /* unsupported / inconsistent shape            -> ValueError   */
  FPLX_E_BADDTYPE = -2,   /* unsupported dtype enum                      -> ValueError   */
  FPLX_E_WORKSPACE = -3,  /* workspace too small                         -> RuntimeError */
  FPLX_E_HIP = -4,        /* HIP launch error                            -> RuntimeError */
  FPLX_E_NULL = -5        /* required pointer is NULL                    -> ValueError   */
};

int fplx_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
int fplx_last_error(char* buf, size_t n);
/* number of partial-sum rows the reduction kernels write for a tensor of `voxels` voxels */
int fplx_num_partials(int64_t voxels);

/* ------------------------------------------------------------------ tuning knobs and plan queries
 * (no reference counterpart: the reference's kernels are cuDNN's, chosen by torch.backends.cudnn.benchmark,
 *  agent_seg.py:737-738 sets deterministic = True / benchmark = False)
 * fplx_set_tuning / fplx_get_tuning: named integer knobs that select among equivalent kernels / launch geometries
 *   ("xcd", "brick", "march", "march32_v2", "wg_cot", "tile_ks", ...; fplx_tuning_key enumerates them: returns the key's
 *   length and copies it, or -1 past the end).  Unknown key: FPLX_E_BADSHAPE.  Benchmarks and tests only.
 * fplx_conv3d_plan_query: which kernel family fplx_conv3d_fwd dispatches this layer to (for 16-byte aligned NDHWC bf16
 *   operands; FPLX_KERNEL_*), the geometry (brick: 0 = 4x8x8, 1 = 5x4x8; depth march: Cin = 32: 2 = the one-wave-per-SIMD
 *   kernels for footprints inside the volume [statistics form / statistics-free form], 0 = the 8-wave kernel; Cin >= 64: the
 *   footprint width 16 | 32; -1 for other families), the split of the reduction dimension (1 = none) and the statistics
 *   rows - pure host code, no launch. */
enum {
  FPLX_KERNEL_GENERIC = 0, FPLX_KERNEL_DIRECT = 1, FPLX_KERNEL_TILE = 2, FPLX_KERNEL_STREAM = 3, FPLX_KERNEL_MARCH = 4,
  FPLX_KERNEL_BRICK = 5, FPLX_KERNEL_STEM = 6, FPLX_KERNEL_OUTCONV = 7
};
int fplx_set_tuning(const char* key, int64_t value);
int fplx_get_tuning(const char* key, int64_t* value);
int fplx_tuning_key(int index, char* buf, size_t n);
int fplx_conv3d_plan_query(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw, int x_dt, int y_dt,
                           int* kernel, int* geometry, int* ksplit, int* stats_rows);

/* ------------------------------------------------------------------ weight packing
 * nn.Conv3d weight [Cout][Cin][KD][KH][KW] fp32 (net/net3d/unet2d5_dsbn.py:54-55, 293-294)
 *   -> wf[tap][Cout][Cin]  (forward operand)
 *   -> wb[tap][Cin][Cout]  with tap mirrored (data-gradient operand; may be NULL)
 * both in dtype dt.  tap = (kd*KH + kh)*KW + kw. */
int fplx_pack_conv_weight(const float* w, void* wf, void* wb, int cout, int cin, int kd, int kh, int kw,
                          int dt, fplx_stream_t stream);
/* the same for n (<= 32) 3x3x3 layers in one launch: host arrays of per-layer pointers / channel counts (what the train
 * step does after every optimizer step for all ConvBlockND convolutions, unet2d5_dsbn.py:54-55).  wb[i] may be NULL.
 * stamp (may be NULL; layers with cout % 16 == 0 and cin % 32 == 0 in bf16 only): stamp[i] = caller-owned buffer of
 * (cout / 16) * (cin / 32) * 32 floats per layer (or NULL) that receives, per 16 x 32-channel pack tile, 32 of the fp32 master
 * values the tile's pack was made from.  verify = 1: the packs and stamps of an earlier call (or of fplx_adam_pack_step) are
 * still in wf / wb / stamp - only the tiles whose masters no longer match their stamps bit for bit are packed again.  This is
 * how a caller that keeps packs across optimiser steps notices writers that bypass it (torch's `.data` edits bump no version
 * counter: agent_seg.py's EMA / init paths); whole-tensor writes are always caught, single-element edits between the sampled
 * positions (elements 0 and 432 of every 864-element row segment) are not. */
int fplx_pack_conv_weights_batched(int n, const float* const* w, void* const* wf, void* const* wb, const int* cout,
                                   const int* cin, int dt, float* const* stamp, int verify, fplx_stream_t stream);
/* n (<= 16) small packs in one launch: job i is a convolution weight [a][b][taps] fp32 (kind 0: a = Cout, b = Cin, layouts of
 * fplx_pack_conv_weight) or a transposed-convolution weight (kind 1: a = Cin, b = Cout, layouts of fplx_pack_deconv_weight /
 * _deconv122_weight) packed into wf[i] / wb[i] (either may be NULL) in dtype dt[i] - the stem, the four transposed convolutions
 * and out_conv of the network (unet2d5_dsbn.py:54, 152, 293): what a train step otherwise launches one by one after Adam */
int fplx_pack_weights_multi(int n, const int* kind, const float* const* w, void* const* wf, void* const* wb, const int* a,
                            const int* b, const int* taps, const int* dt, fplx_stream_t stream);
/* nn.ConvTranspose3d weight [Cin][Cout][2][2][2] fp32 (unet2d5_dsbn.py:152)
 *   -> wf[tap][Cout][Cin] and wb[tap][Cin][Cout] (dtype dt), tap = (i*2+j)*2+k */
int fplx_pack_deconv_weight(const float* w, void* wf, void* wb, int cin, int cout, int dt,
                            fplx_stream_t stream);

/* ------------------------------------------------------------------ convolution
 * Generic "same" convolution, stride 1, zero padding (KD/2, KH/2, KW/2):
 *   nn.Conv3d(k=3,p=1)            unet2d5_dsbn.py:54-55 used at 75,79
 *   nn.Conv3d(k=(1,3,3),p=(0,1,1)) unet2d5_dsbn.py:293-294, 307 (out_conv)
 * and, with the mirrored pack `wb`, the data gradient of either.
 *  x      input, described by explicit element strides (sn,sd,sh,sw,sc) so that both the
 *         fp32 NCDHW network input and NDHWC activations can be read; type x_dt
 *  wp     packed weight [taps][Cout][Cin], type = y's dt for NDHWC outputs, fp32 otherwise
 *  bias   fp32 [Cout] or NULL
 *  y      output, element strides (yn,yd,yh,yw,yc), type y_dt
 *         (i.e. the packed weight has the type of y)
 *  stats  optional fp32 [fplx_conv3d_stats_rows(...)][2][Cout]: per-row sum / sum of squares of
 *         the (unrounded) outputs, for the BatchNorm that follows (dsbn.py:54-57).  The row
 *         count depends on the kernel variant chosen for the shape; ask for it. */
int fplx_conv3d_stats_rows(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw, int x_dt, int y_dt);
/*  ws     workspace of fplx_conv3d_fwd_ws_bytes(...) bytes (0 for most shapes; small deep-level volumes
 *         use a split-K kernel that keeps fp32 partial tiles there).  ws may be NULL when that is 0. */
size_t fplx_conv3d_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw, int x_dt, int y_dt);
int fplx_conv3d_fwd(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc,
                    const void* wp, const float* bias,
                    void* y, int y_dt, int64_t yn, int64_t yd, int64_t yh, int64_t yw, int64_t yc,
                    int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw,
                    float* stats, void* ws, size_t ws_bytes, fplx_stream_t stream);

/* weight + bias gradient of the same convolution.
 *  dw   fp32 [Cout][Cin][KD][KH][KW] (torch layout), db fp32 [Cout] or NULL
 *  ws   workspace of at least fplx_conv3d_wgrad_ws_bytes(...) bytes */
size_t fplx_conv3d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout, int kd, int kh, int kw);
int fplx_conv3d_wgrad(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc,
                      const void* dy, int dy_dt, int64_t yn, int64_t yd, int64_t yh, int64_t yw, int64_t yc,
                      float* dw, float* db, int n, int d, int h, int w, int cin, int cout,
                      int kd, int kh, int kw, void* ws, size_t ws_bytes, fplx_stream_t stream);

/* The stem site's weight gradient fused with the apply pass of its BatchNorm + PReLU backward (round 6).  The gradient w.r.t.
 * the output of the network's first convolution (Conv3d(in_chns -> C0), unet2d5_dsbn.py:54, 74-77) has one consumer - this
 * weight gradient; the network input needs no data gradient - so fplx_bn_act_bwd_apply + fplx_conv3d_wgrad of that site (write dy,
 * read it back) become one call: y = the convolution's stored output (bf16 [V, ldy]), dout = gradient w.r.t. the site's output a
 * (bf16 [V, ldd]), mean .. slope = the site's BatchNorm constants, coef = fplx_bn_act_bwd_finalize's; the kernel forms
 * dy = scale (dz - k0 - x-hat k1) -> bf16 on the pieces it stages (fplx_bn_act_bwd_apply's arithmetic, dropout-free) and dy is
 * never stored.  x: fp32 NCDHW contiguous, in_chns 1 | 4, C0 % 32 == 0 (fplx_conv3d_plan_query answers FPLX_KERNEL_STEM for the
 * forward); dw, ws as fplx_conv3d_wgrad with ws of fplx_conv3d_wgrad_ws_bytes(.., 3, 3, 3) bytes.  dw is bit for bit the
 * two-call result. */
int fplx_stem_wgrad_bn(const float* x, const void* y, int64_t ldy, const void* dout, int64_t ldd, const float* mean,
                       const float* rstd, const float* scale, const float* shift, const float* slope, const float* coef,
                       float* dw, int n, int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream);

/* The first convolution of an UpBlock reads torch.cat([skip, up], dim=1) (unet2d5_dsbn.py:182-183).  Where the two
 * halves are 32 channels wide (level 0) a concat BUFFER would be addressed in half 128-byte lines by every other
 * kernel that touches one half (pooling, transposed convolution), so there the concatenation is never built: these
 * three entry points take / produce the two tensors separately (bf16 NDHWC, both with leading dimension ldx;
 * cin = total input channels = 64).  fplx_conv3d_cat2_ok tells whether a shape is served; stats rows and the
 * workspace size are those of fplx_conv3d_stats_rows / fplx_conv3d_wgrad_ws_bytes for the same (cin, cout). */
int fplx_conv3d_cat2_ok(int n, int d, int h, int w, int cin, int cout);
int fplx_conv3d_fwd_cat2(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias, void* y,
                         int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats,
                         fplx_stream_t stream);
/* dy: [voxels][cout]; wb: the mirrored pack of the same weight; dx0 / dx1: gradients of the two input halves */
int fplx_conv3d_dgrad_split2(const void* dy, int64_t ldy, const void* wb, void* dx0, void* dx1, int64_t ldx, int n,
                             int d, int h, int w, int cin, int cout, fplx_stream_t stream);
int fplx_conv3d_wgrad_cat2(const void* x0, const void* x1, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n,
                           int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream);

/* ------------------------------------------------------------------ inference: convolution + BatchNorm(eval) + PReLU in one kernel
 * ConvBlockND in eval mode (unet2d5_dsbn.py:74-81 with dsbn.py:54-57 on running statistics): BatchNorm is then a fixed
 * per-channel affine map, z = scale (conv(x; w) + b) + shift = conv(x; scale w) + (scale b + shift).  The CALLER folds scale
 * into the weights before packing them (fplx_pack_conv_weight / _conv2d_weight) and passes bias = scale b + shift; the
 * kernel applies PReLU(prelu_slope[0]) in its write-out, so the separate fplx_bn_act_fwd pass disappears.  bf16 NDHWC
 * operands, 3x3x3 (mid != 0: a Conv2d pack in the middle depth plane).  x1 != NULL: the input is cat([x0, x1], channel)
 * of two cin/2-channel tensors (as fplx_conv3d_fwd_cat2; here also the brick kernel's layers, e.g. 64 || 64 -> 64 at level
 * 1).  Only for layers fplx_conv3d_fwd_act_ok accepts (the depth-march,
 * brick and split-K kernels); FPLX_E_BADSHAPE otherwise - callers fall back to fplx_conv3d_fwd + fplx_bn_act_fwd.
 * n_x0 (two-tensor form only; 0 = n): x0 holds n_x0 samples and sample i reads x0[i % n_x0] - the Monte-Carlo passes of
 * test-time dropout (agent_seg.py:898-909) share the encoder levels above the first active dropout, whose skip tensor is
 * then not replicated per pass.  ws: fplx_conv3d_fwd_ws_bytes / fplx_conv2d_fwd_ws_bytes of the layer. */
int fplx_conv3d_fwd_act_ok(int n, int d, int h, int w, int cin, int cout, int mid, int cat2);
int fplx_conv3d_fwd_act(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias,
                        const float* prelu_slope, void* y, int64_t ldy, int n, int d, int h, int w, int cin, int cout,
                        int mid, int n_x0, void* ws, size_t ws_bytes, fplx_stream_t stream);

/* ------------------------------------------------------------------ transposed convolution
 * nn.ConvTranspose3d(k=2,s=2) (unet2d5_dsbn.py:152,181).  x: [N,D,H,W,Cin] ld ldx;
 * y: [N,2D,2H,2W,Cout] ld ldy (normally the upper channel half of the concat buffer, line 182). */
int fplx_deconv2_fwd(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy,
                     int n, int d, int h, int w, int cin, int cout, int dt, fplx_stream_t stream);
int fplx_deconv2_dgrad(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx,
                       int n, int d, int h, int w, int cin, int cout, int dt, fplx_stream_t stream);
size_t fplx_deconv2_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout);
/* dw fp32 [Cin][Cout][2][2][2], db fp32 [Cout] */
int fplx_deconv2_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db,
                       int n, int d, int h, int w, int cin, int cout, int dt,
                       void* ws, size_t ws_bytes, fplx_stream_t stream);

/* ------------------------------------------------------------------ 2.5D levels (conv_dims[l] = 2, the shipped configs:
 * config_dual/data_vs/vs_t1s_g.cfg:58 conv_dims = [2, 2, 3, 3, 3]).  The reference folds the depth axis into the batch and
 * runs 2D modules (unet2d5_dsbn.py:110-127, 160-188); on [N,D,H,W,C] volumes that is
 *  - Conv2d(3x3, pad 1)  = the 3x3x3 kernels with the 9 taps packed into the middle depth plane (pack_conv2d_weight:
 *    w fp32 [Cout][Cin][3][3] -> the same wf / wb layouts as fplx_pack_conv_weight); weight gradient = middle plane of
 *    the 27-tap gradient (dw27 [Cout][Cin][27] -> dw9 [Cout][Cin][9], taps 9..17)
 *  - MaxPool2d(2)        = maxpool122: x [N,D,H,W,C] -> y [N,D,H/2,W/2,C]
 *  - ConvTranspose2d(2,2) = deconv122: x [N,D,H,W,Cin] -> y [N,D,2H,2W,Cout]; weights fp32 [Cin][Cout][2][2] packed to
 *    wf [4][Cout][Cin], wb [4][Cin][Cout]; dw fp32 [Cin][Cout][2][2]
 *  BatchNorm2d over (N D, C, H, W) has the statistics of BatchNorm3d over (N, C, D, H, W): same kernels.
 *  fplx_conv2d_* = fplx_conv3d_* (3x3x3) for such packs, same arguments and results; knowing that only the middle plane
 *  is live, the implicit-GEMM kernel runs 9 taps instead of 27. */
int fplx_pack_conv2d_weight(const float* w, void* wf, void* wb, int cout, int cin, int dt, fplx_stream_t stream);
int fplx_conv2d_stats_rows(int n, int d, int h, int w, int cin, int cout, int x_dt, int y_dt);
size_t fplx_conv2d_fwd_ws_bytes(int n, int d, int h, int w, int cin, int cout, int x_dt, int y_dt);
int fplx_conv2d_fwd(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc,
                    const void* wp, const float* bias, void* y, int y_dt, int64_t yn, int64_t yd, int64_t yh,
                    int64_t yw, int64_t yc, int n, int d, int h, int w, int cin, int cout,
                    float* stats, void* ws, size_t ws_bytes, fplx_stream_t stream);
int fplx_conv2d_fwd_cat2(const void* x0, const void* x1, int64_t ldx, const void* wp, const float* bias, void* y,
                         int64_t ldy, int n, int d, int h, int w, int cin, int cout, float* stats,
                         fplx_stream_t stream);                     /* = fplx_conv3d_fwd_cat2 for such packs */
int fplx_conv2d_dgrad_split2(const void* dy, int64_t ldy, const void* wb, void* dx0, void* dx1, int64_t ldx, int n,
                             int d, int h, int w, int cin, int cout, fplx_stream_t stream);
/* weight gradient of the Conv2d: dw fp32 [Cout][Cin][3][3], db fp32 [Cout] or NULL (9 of 27 taps on the MFMA path) */
size_t fplx_conv2d_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout);
int fplx_conv2d_wgrad(const void* x, int x_dt, int64_t sn, int64_t sd, int64_t sh, int64_t sw, int64_t sc,
                      const void* dy, int dy_dt, int64_t yn, int64_t yd, int64_t yh, int64_t yw, int64_t yc,
                      float* dw, float* db, int n, int d, int h, int w, int cin, int cout,
                      void* ws, size_t ws_bytes, fplx_stream_t stream);
int fplx_conv2d_wgrad_cat2(const void* x0, const void* x1, int64_t ldx, const void* dy, int64_t ldy, float* dw, int n,
                           int d, int h, int w, int cin, int cout, void* ws, size_t ws_bytes, fplx_stream_t stream);
int fplx_maxpool122_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c,
                        int dt, fplx_stream_t stream);
int fplx_maxpool122_bwd(const void* x, int64_t ldx, const void* dy, int64_t ldy, const void* dskip, int64_t lds,
                        void* dx, int64_t ldo, int n, int d, int h, int w, int c, int dt, fplx_stream_t stream);
int fplx_pack_deconv122_weight(const float* w, void* wf, void* wb, int cin, int cout, int dt, fplx_stream_t stream);
int fplx_deconv122_fwd(const void* x, int64_t ldx, const void* wf, const float* bias, void* y, int64_t ldy,
                       int n, int d, int h, int w, int cin, int cout, int dt, fplx_stream_t stream);
int fplx_deconv122_dgrad(const void* dy, int64_t ldy, const void* wb, void* dx, int64_t ldx,
                         int n, int d, int h, int w, int cin, int cout, int dt, fplx_stream_t stream);
size_t fplx_deconv122_wgrad_ws_bytes(int n, int d, int h, int w, int cin, int cout);
int fplx_deconv122_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* db,
                         int n, int d, int h, int w, int cin, int cout, int dt,
                         void* ws, size_t ws_bytes, fplx_stream_t stream);

/* ------------------------------------------------------------------ DSBN + PReLU + dropout
 * Training-mode statistics of nn.BatchNorm3d for the ACTIVE domain (dsbn.py:54-57):
 *   stats [rows][2][C] (from fplx_conv3d_fwd) -> mean, biased var -> scale = gamma*rstd,
 *   shift = beta - mean*scale; running_mean/var updated with momentum (unbiased var),
 *   num_batches_tracked += 1 (int64, may be NULL).  eps as torch (1e-5). */
int fplx_bn_train_finalize(const float* stats, int rows, int c, int64_t count,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float momentum, float eps,
                           float* mean, float* rstd, float* scale, float* shift, fplx_stream_t stream);
/* Statistics of an NDHWC tensor for a stand-alone DomainSpecificBatchNorm3d call (dsbn.py:54-57):
 * stats fp32 [fplx_num_partials(voxels)][2][C], same format fplx_conv3d_fwd emits. */
int fplx_channel_stats(const void* x, int64_t ldx, int64_t voxels, int c, int dt, float* stats,
                       fplx_stream_t stream);
/* Eval mode: scale/shift from the running statistics. */
int fplx_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int c, float* scale, float* shift,
                         fplx_stream_t stream);
/* out = dropout_p(PReLU_slope(y*scale + shift))   (unet2d5_dsbn.py:76-78 / 80-81)
 * dropout keep-mask = counter-based Philox stream (seed, stream_id, flat NDHWC index of `out`
 * assuming ld == C numbering); p == 0 disables it.  y and out may alias. */
int fplx_bn_act_fwd(const void* y, int64_t ldy, void* out, int64_t ldo, const float* scale, const float* shift,
                    const float* slope, float p, uint64_t seed, uint32_t stream_id,
                    int64_t voxels, int c, int dt, fplx_stream_t stream);
/* Backward, stage 1: partial sums over voxels of dz and dz*xhat per channel and of the PReLU
 * slope gradient: part [rows][2*C+1] (rows = fplx_num_partials(voxels)). */
int fplx_bn_act_bwd_reduce(const void* y, int64_t ldy, const void* dout, int64_t ldd,
                           const float* mean, const float* rstd, const float* scale, const float* shift,
                           const float* slope, float p, uint64_t seed, uint32_t stream_id,
                           int64_t voxels, int c, int dt, float* part, fplx_stream_t stream);
/* stage 2: dgamma, dbeta (ACCUMULATED into the fp32 grads), dslope (accumulated), and the
 * two per-channel coefficients coef[2][C] used by stage 3.  train=0: eval-mode BN (statistics
 * are constants): coefficients are zero. */
int fplx_bn_act_bwd_finalize(const float* part, int rows, int c, int64_t count, int train,
                             float* dgamma, float* dbeta, float* dslope, float* coef, fplx_stream_t stream);
/* stage 3: dy = gamma*rstd * (dz - coef0 - xhat*coef1).  dy may alias dout. */
int fplx_bn_act_bwd_apply(const void* y, int64_t ldy, const void* dout, int64_t ldd, void* dy, int64_t ldo,
                          const float* mean, const float* rstd, const float* scale, const float* shift,
                          const float* slope, const float* coef, float p, uint64_t seed, uint32_t stream_id,
                          int64_t voxels, int c, int dt, fplx_stream_t stream);

/* ------------------------------------------------------------------ pooling
 * nn.MaxPool3d(2,2) (unet2d5_dsbn.py:106,117).  x [N,D,H,W,C] ld ldx -> y [N,D/2,H/2,W/2,C] ld ldy */
int fplx_maxpool2_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c,
                      int dt, fplx_stream_t stream);
/* dx = dskip (may be NULL) + scatter(dy to the first maximum of each 2x2x2 window of x) */
int fplx_maxpool2_bwd(const void* x, int64_t ldx, const void* dy, int64_t ldy, const void* dskip, int64_t lds,
                      void* dx, int64_t ldo, int n, int d, int h, int w, int c, int dt, fplx_stream_t stream);

/* Fused tail of a DownBlock (second ConvBlockND site - no dropout there - then the pooling, unet2d5_dsbn.py:79-81, 117):
 *  fplx_bn_act_pool_fwd:   a2 = PReLU(BN(y2)) written to `out` (the skip tensor) and MaxPool(a2) to `pooled` in one pass.
 *  fplx_pool_bwd_bn_reduce: dx = dskip + unpooled(dy) = d(a2) (first-maximum rule on a2 recomputed from y2) and, over that
 *                          dx, the partial rows of fplx_bn_act_bwd_reduce for this site (same layout: fplx_num_partials(voxels)
 *                          rows of 2C+1 floats) - the caller then runs fplx_bn_act_bwd_finalize / _apply as usual.
 * pd = 2: MaxPool3d(2), pd = 1: MaxPool2d(2) per depth slice; dims = the UNPOOLED tensor's; bf16 only, see fplx_bn_pool_fused_ok. */
int fplx_bn_pool_fused_ok(int c, int dt);
int fplx_bn_act_pool_fwd(const void* y, int64_t ldy, void* out, int64_t ldo, void* pooled, int64_t ldp, const float* scale,
                         const float* shift, const float* slope, int n, int d, int h, int w, int c, int dt, int pd,
                         fplx_stream_t stream);
int fplx_pool_bwd_bn_reduce(const void* y, int64_t ldy, const void* dy, int64_t lddy, const void* dskip, int64_t lds, void* dx,
                            int64_t ldo, const float* mean, const float* rstd, const float* scale, const float* shift,
                            const float* slope, int n, int d, int h, int w, int c, int dt, int pd, float* part,
                            fplx_stream_t stream);

/* out_conv (Conv3d C0 -> classes, kernel (1,3,3), unet2d5_dsbn.py:293-294, 307) fused with the BatchNorm + PReLU passes of the
 * convolution site in front of it (ConvBlockND's second site of the last UpBlock, unet2d5_dsbn.py:79-81; dropout-free, bf16
 * NDHWC, C0 = 32, classes <= 4: fplx_outconv_bn_rows > 0).  The out_conv operands are 1 / 8 the size of that site's tensors, so
 *   fplx_outconv_fwd_bn          reads the site's PRE-BatchNorm output y, applies a = PReLU(scale y + shift) on the way into its
 *                                tiles, writes a (unless NULL, see below) and the fp32 planar logits: fplx_bn_act_fwd + fplx_conv3d_fwd
 *                                in one pass over y; a and the logits are the bits of the two-call path;
 *   fplx_outconv_dgrad_bn_reduce / _apply   never store out_conv's data gradient: both RECOMPUTE it from dlogits (fp32 planar)
 *                                and the mirrored pack wb and run fplx_bn_act_bwd_reduce / _apply on it - _reduce writes
 *                                fplx_outconv_bn_rows(...) partial rows of 2 C0 + 1 floats for fplx_bn_act_bwd_finalize,
 *                                _apply takes the finalize's coef and writes dy (gradient w.r.t. y);
 *   fplx_outconv_wgrad_bn        out_conv's weight gradient dw [classes][C0][1][3][3] and bias gradient db [classes] (NULL: not
 *                                wanted) from y as well: a is formed on the way in, exactly as fplx_outconv_fwd_bn forms it.
 *                                With the three of them a has no reader left in a training step, and fplx_outconv_fwd_bn takes
 *                                a = NULL (no activation written) wherever fplx_outconv_wgrad_bn_ws_bytes(...) > 0 - classes <= 3;
 *                                0 = this form is not available, a is required and fplx_conv3d_wgrad takes the gradient from it.
 *                                ws: fplx_outconv_wgrad_bn_ws_bytes(...) bytes.  dw agrees with fplx_conv3d_wgrad on the stored a
 *                                to fp32 summation order (same bf16 operands).
 * mean / rstd / scale / shift: the site's BatchNorm constants (fplx_bn_train_finalize / fplx_bn_eval_prepare). */
/* partial rows the fused backward writes, or 0 where the fused forms do not apply (the caller then runs the separate passes) */
int fplx_outconv_bn_rows(int n, int d, int h, int w, int c0, int ncls);
int fplx_outconv_fwd_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* prelu_slope, void* a,
                        int64_t lda, const float* wf, const float* bias, float* logits, int n, int d, int h, int w, int c0,
                        int ncls, fplx_stream_t stream);
int fplx_outconv_dgrad_bn_reduce(const float* dlogits, const void* wb, const void* y, int64_t ldy, const float* mean,
                                 const float* rstd, const float* scale, const float* shift, const float* prelu_slope,
                                 float* part, int n, int d, int h, int w, int c0, int ncls, fplx_stream_t stream);
int fplx_outconv_dgrad_bn_apply(const float* dlogits, const void* wb, const void* y, int64_t ldy, const float* mean,
                                const float* rstd, const float* scale, const float* shift, const float* prelu_slope,
                                const float* coef, void* dy, int64_t lddy, int n, int d, int h, int w, int c0, int ncls,
                                fplx_stream_t stream);
size_t fplx_outconv_wgrad_bn_ws_bytes(int n, int d, int h, int w, int c0, int ncls);
int fplx_outconv_wgrad_bn(const void* y, int64_t ldy, const float* scale, const float* shift, const float* prelu_slope,
                          const float* dlogits, float* dw, float* db, int n, int d, int h, int w, int c0, int ncls, void* ws,
                          size_t ws_bytes, fplx_stream_t stream);

/* (Tri / bi)linear x2 upsampling, align_corners = True - UpBlock with bilinear = True (unet2d5_dsbn.py:148-150, 172-176:
 * nn.Upsample(scale_factor=2, mode='trilinear' | 'bilinear', align_corners=True) behind a kernel-1 convolution, which runs
 * through fplx_conv3d_fwd / _wgrad with kd = kh = kw = 1).  x [n][d][h][w][ldx] -> y [n][sd*d][2h][2w][ldy]; sd = 2:
 * trilinear, sd = 1: bilinear on every depth slice (2.5D levels).  bwd: dx = the transpose applied to dy (gather, no atomics). */
int fplx_upsample2_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, int n, int d, int h, int w, int c, int dt, int sd,
                       fplx_stream_t stream);
int fplx_upsample2_bwd(const void* dy, int64_t ldy, void* dx, int64_t ldx, int n, int d, int h, int w, int c, int dt, int sd,
                       fplx_stream_t stream);

/* ------------------------------------------------------------------ segmentation loss
 * Fused softmax + Dice (loss/seg/dice.py:20-57, util.py:85-107) + cross entropy
 * (loss/seg/ce.py:23-44) + per-sample image-weighted Dice (dice.py:106-128) + entropy
 * regulariser (net_run_dsbn/agent_seg.py:352-354) + hard-Dice train metric
 * (agent_seg.py:472-476).  logits / label: fp32 [N,C,D,H,W] contiguous (C <= 8);
 * pixel_weight fp32 [N,1,D,H,W] or NULL.
 *   part   fp32 workspace [N][rows][FPLX_LOSS_K(C)], rows = fplx_loss_rows(D*H*W) (partial rows + 5 spare rows that hold the
 *          per-sample sums and batch totals as (N + 1) x K doubles, 8-byte aligned: exactly N*rows*K floats are touched)
 *   cfg    host floats: w_dice, w_ce, w_dice_img (per-sample Dice x image_weight), w_entropy
 *   image_weight fp32 [N] device or NULL (needed iff w_dice_img != 0)
 *   out    fp32 device [4 + C]: total loss, dice term, ce term, entropy term, hard class Dice[C]
 *   coef   fp32 device [N][C][2] + [2]: backward coefficients (written by the finalize kernel)
 * Data parallelism (one process per GPU): the reference's nn.DataParallel gathers the replicas' logits and evaluates ONE
 * loss over the full batch (agent_seg.py:692-698 with get_loss_value 134-142); Dice and the weighted CE are not sums of
 * per-shard losses.  fplx_seg_loss_fwd = fplx_seg_loss_sums + fplx_seg_loss_from_sums; between the two a caller
 * all-reduces `totals` (double [FPLX_LOSS_K(C)], the sums over the local samples) over the ranks and passes the global
 * sample count: loss value, metric and coefficients are then those of the full batch, and the ranks' gradients ADD UP to
 * the full-batch gradient (no division by the world size).
 *   sums   double device [N][FPLX_LOSS_K(C)] per-sample sums;  totals double device [FPLX_LOSS_K(C)] */
#define FPLX_LOSS_K(C) (6 * (C) + 3)
int fplx_loss_rows(int64_t voxels_per_sample);
int fplx_seg_loss_fwd(const float* logits, const float* label, const float* pixel_weight,
                      const float* image_weight, int n, int c, int64_t voxels_per_sample,
                      float w_dice, float w_ce, float w_dice_img, float w_entropy, int softmax,
                      float* part, float* out, float* coef, fplx_stream_t stream);
int fplx_seg_loss_sums(const float* logits, const float* label, const float* pixel_weight, int n, int c,
                       int64_t voxels_per_sample, int softmax, float* part, double* sums, double* totals,
                       fplx_stream_t stream);
int fplx_seg_loss_from_sums(const double* sums, const double* totals, const float* image_weight, int n, int n_global, int c,
                            int64_t voxels_per_sample, int has_pixel_weight, float w_dice, float w_ce, float w_dice_img,
                            float w_entropy, float* out, float* coef, fplx_stream_t stream);
/* dlogits = gscale[0] * dLoss/dlogits ; gscale: device fp32 scalar (upstream gradient) */
int fplx_seg_loss_bwd(const float* logits, const float* label, const float* pixel_weight,
                      const float* coef, const float* gscale, int n, int c, int64_t voxels_per_sample,
                      float w_dice, float w_ce, float w_dice_img, float w_entropy, int softmax,
                      float* dlogits, fplx_stream_t stream);

/* ------------------------------------------------------------------ optimiser
 * torch.optim.Adam(lr, weight_decay) as built by net_run_dsbn/get_optimizer.py:17 on a flat
 * fp32 buffer: g += wd*p; m,v moments; bias correction with `step` (1-based); p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps).
 * grad_scale multiplies g first (1/world_size after an all-reduce sum). */
int fplx_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, fplx_stream_t stream);
/* fplx_adam_step over the flat segment [0, n) AND, in the same launch, the bf16 packs fplx_pack_conv_weights_batched makes
 * of the `nl` (<= 32) 3x3x3 convolution weights that live inside it - from the UPDATED values (the weights change only here;
 * what the train step otherwise re-reads at the head of every forward, unet2d5_dsbn.py:54-55 + get_optimizer.py:17).
 * off[i]: element offset of layer i's [cout][cin][3][3][3] weight in the segment (ascending, multiples of 4); layers must
 * satisfy fplx_adam_pack_ok; wb[i] may be NULL; stamp: as in fplx_pack_conv_weights_batched (may be NULL), written from the
 * updated values.  Same parameters, moments and packs as the two separate calls, bit for bit. */
int fplx_adam_pack_ok(int cout, int cin);
int fplx_adam_pack_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, int nl, const int64_t* off, const int* cout,
                        const int* cin, void* const* wf, void* const* wb, float* const* stamp, fplx_stream_t stream);

/* ------------------------------------------------------------------ pseudo-label filter
 * FPL branch of SegmentationAgent.infer (net_run_dsbn/agent_seg.py:911-931) for one volume:
 * logits fp32 [T][C][V] (T MC/TTA passes).  Per voxel: softmax per pass (scipy.special.softmax,
 * fp32), hard label per pass (argmax, uint8 [T][V], may be NULL), population variance over T
 * summed over classes, mean_T p_1, u = -m*log(m+1e-6), boundary = #(u > thr).
 *   part  workspace fp32/int [rows][2] (rows = fplx_num_partials(V))
 *   out   device double [4]: vars, boundary, uncer_one (= 1 if boundary < 50 else vars/boundary), 0
 *   mean_out / unc_out optional fp32 [V] maps */
int fplx_mc_filter(const float* logits, int t, int c, int64_t v, float thr, uint8_t* hards,
                   float* mean_out, float* unc_out, double* part, double* out, fplx_stream_t stream);
/* save_outputs (agent_seg.py:1049-1050): softmax -> argmax -> uint8.  logits fp32 [N][C][V] */
int fplx_hard_label(const float* logits, int n, int c, int64_t v, uint8_t* out, fplx_stream_t stream);
/* data/get_pixel_weight.py:21-26 (/root/reference): w = 1 - 0.5*xor(a,b) (a,b uint8 {0,1}) as fp32
 * (exact in fp32: values are 1.0 / 0.5), followed, if apply_set_weight, by
 * NiftyDataset.set_weight_ (io/nifty_dataset.py:165-168): w<1 -> 0, w *= image_weight. */
int fplx_pixel_weight(const uint8_t* a, const uint8_t* b, int64_t v, int apply_set_weight, float image_weight,
                      float* out, fplx_stream_t stream);

/* ------------------------------------------------------------------ training-sample transforms
 * The reference's numpy chain train_transform = [NormalizeWithMeanStd, Pad, RandomCrop, RandomFlip,
 * LabelToProbability] (config_dual/data_vs/vs_t1s_g.cfg:21) on device volumes [C][D][H][W]; the random draws stay on
 * the host (fplx/transform.py, Python `random` in the reference's order).
 *  normalize: y = (x - mean) / std per call (one channel); mean_std == NULL -> float32 mean / population std of x
 *             (PyMIC/pymic/transform/normalize.py:43-68); ws of fplx_normalize_ws_bytes() bytes; out_mean_std may be NULL
 *  normalize_positive: NormalizeWithMeanStd_ignore_non_positive (normalize.py:55-66): mean / std over the voxels > 0, y =
 *             (x - mean) / std there and noise[i] elsewhere (noise: the caller's numpy.random.normal(0, 1) draw, as fp32);
 *             y may alias x
 *  pad_reflect: numpy.pad(mode='reflect') with lower margins lo_* (pad.py:126-163); elem_bytes 4 (fp32) or 1 (uint8)
 *  crop_flip: crop box [c*, c*+o*) then flip of the cropped patch, flip_mask bit0 = W, bit1 = H, bit2 = D
 *             (crop.py:27-49, flip.py:34-62)
 *  label_bbox: out9 = [count, min c,d,h,w, max+1 c,d,h,w] of {label in mask_labels} (util/image_process.py:8-34)
 *  label_to_probability: one-hot fp32 [class_num][voxels] (label_convert.py:82-94)
 *  set_weight: in place pw = (pw < 1 ? 0 : pw) * image_weight (PyMIC/pymic/io/nifty_dataset.py:165-168) */
size_t fplx_normalize_ws_bytes(void);
int fplx_normalize_mean_std(const float* x, float* y, int64_t n, const float* mean_std, void* ws, size_t ws_bytes,
                            float* out_mean_std, fplx_stream_t stream);
int fplx_normalize_positive(const float* x, const float* noise, float* y, int64_t n, void* ws, size_t ws_bytes,
                            float* out_mean_std, fplx_stream_t stream);
int fplx_pad_reflect(const void* x, void* y, int elem_bytes, int c, int d, int h, int w, int lo_d, int lo_h, int lo_w,
                     int od, int oh, int ow, fplx_stream_t stream);
int fplx_crop_flip(const void* x, void* y, int elem_bytes, int c, int d, int h, int w, int cd, int ch, int cw, int od,
                   int oh, int ow, int flip_mask, fplx_stream_t stream);
int fplx_label_bbox(const unsigned char* label, int c, int d, int h, int w, const int* mask_labels, int nmask, int* out9,
                    fplx_stream_t stream);
int fplx_label_to_probability(const unsigned char* label, float* prob, int class_num, int64_t voxels,
                              fplx_stream_t stream);
int fplx_set_weight(float* pixel_weight, int64_t n, float image_weight, fplx_stream_t stream);

/* ------------------------------------------------------------------ Inferer: sliding window + flip TTA
 * PyMIC/pymic/net_run_dsbn/infer_func.py:50-112 (tiling, overlap averaging), 188-222 (tta_mode 1).
 * The tile grid is the Cartesian product of per-axis start lists (HOST arrays, at most 64 entries each, non-decreasing,
 * first 0, last start + window == size; a clamped duplicate start is a separate tile, as in the reference's list);
 * tile index = (iw * nh + ih) * nd + id, the reference's enumeration order.  flips: HOST array of 1..4 masks,
 * bit0 = W axis flipped, bit1 = H axis flipped (reference order: 0, 2, 1, 3).
 *  fplx_sw_extract: image fp32 [n][c][d][h][w] -> patches [nflips][tiles][n][c][wd][wh][ww]: tile t of flip f is cut from
 *                   the flipped image.
 *  fplx_sw_merge:   predictions [nflips][tiles][n][c][wd][wh][ww] (c = classes) -> out [n][c][d][h][w]: per flip the
 *                   sum over the covering tiles IN TILE ORDER divided by their number (no division when there is one
 *                   tile), flipped back, then ((o1 + o2) + o3 + o4) / nflips - the reference's additions in its order.
 *  fplx_sw_merge_mc: the same for `passes` Monte-Carlo predictions of every patch in ONE launch (agent_seg.py:898-909 calls
 *                   the inferer once per pass) -> out [passes][n][c][d][h][w], reading the predictions where the network wrote
 *                   them: it ran on chunks of `chunk` consecutive patches (the last one shorter) and each call produced all
 *                   passes of its chunk, pass-major - [chunks][passes][patches of the chunk][c][wd][wh][ww].
 *                   fplx_sw_merge == passes 1, chunk = all patches. */
int fplx_sw_extract(const float* image, int n, int c, int d, int h, int w, const int* starts_d, int nd, const int* starts_h,
                    int nh, const int* starts_w, int nw, int wd, int wh, int ww, const int* flips, int nflips, float* patches,
                    fplx_stream_t stream);
int fplx_sw_merge(const float* patches, int n, int c, int d, int h, int w, const int* starts_d, int nd, const int* starts_h,
                  int nh, const int* starts_w, int nw, int wd, int wh, int ww, const int* flips, int nflips, float* out,
                  fplx_stream_t stream);
int fplx_sw_merge_mc(const float* patches, int passes, int chunk, int n, int c, int d, int h, int w, const int* starts_d, int nd,
                     const int* starts_h, int nh, const int* starts_w, int nw, int wd, int wh, int ww, const int* flips,
                     int nflips, float* out, fplx_stream_t stream);

/* ------------------------------------------------------------------ evaluation
 * Exact voxel counts behind binary_dice / binary_iou / rve / volume (PyMIC/pymic/util/evaluation_seg_train.py:21-50,
 * 68-81, 171-186, 220-225): out[row][3] = {|seg==l & gt==l|, |seg==l|, |gt==l|} (uint64) for each of the nlabels labels
 * (<= 16), or ONE row counting membership in the label list when fuse != 0 (get_multi_class_evaluation_score, 231-262).
 * The scores themselves are float64 host arithmetic on these integers (fplx/evaluation.py). */
int fplx_overlap_counts(const unsigned char* seg, const unsigned char* gt, int64_t n, const int* labels, int nlabels,
                        int fuse, unsigned long long* out, fplx_stream_t stream);

/* Surface metrics behind binary_assd / binary_hd95 (PyMIC/pymic/util/evaluation_seg_train.py:84-99, 101-135, 137-171).
 * fplx_surface_edge_points: get_edge_points - edge = img - binary_erosion(img, cross) of a binary [d][h][w] volume (d = 1: the
 * 2D form, no depth neighbours); outside the volume is background.
 * fplx_surface_min_dist replaces the reference's GeodisTK.geodesic3d_raster_scan(zeros, seeds, spacing, 0.0, 2) (a
 * third-party C++ extension the reference imports, version unpinned, absent here) sampled at the query voxels:
 * out[i] = length of the shortest 26-neighbour lattice path from query i to the nearest seed with Euclidean step
 * lengths under spacing (sz, sy, sx) - the fixed point of that raster scan on a constant image; 1e10 when ns = 0.
 * Coordinates are int32 (z, y, x) triples.  PARITY UNPINNED against GeodisTK itself (see oracle/np_ref.py ev_raster_scan). */
int fplx_surface_edge_points(const unsigned char* img, int d, int h, int w, unsigned char* edge, fplx_stream_t stream);
int fplx_surface_min_dist(const int* query_zyx, int64_t nq, const int* seed_zyx, int64_t ns, float sz, float sy, float sx,
                          float* out, fplx_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* FPLX_H */
