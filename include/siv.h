/*
 * libsiv -- MI355X (gfx950) kernels for the Soft-IntroVAE 3D-MRI training path.
 *
 * C ABI.  Plain pointers and sizes only; no torch types.  The reference
 * (M-hayatooo/Soft-intro-VAE-for-3D-MRI) has no native code and no FFI: its
 * "operator interface" for this path is the set of torch.nn / torch.nn.functional
 * calls its model and trainer make.  Each entry point below names the reference
 * call site(s) it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *  - every function returns 0 on success, a negative SIV_E* code otherwise, never
 *    throws and never exits; siv_last_error() returns a thread-local message.
 *  - all tensors are caller-owned DEVICE pointers to fp32 unless stated.
 *    Activations are NDHWC: [N][D][H][W][C] (the memory of a torch NCDHW tensor in
 *    torch.channels_last_3d format).  3x3x3 weights are [Cout][kd][kh][kw][Cin]
 *    (channels_last_3d memory of the reference's (Cout,Cin,3,3,3) parameter);
 *    1x1x1 weights are [Cout][Cin].
 *  - `stream` is a hipStream_t (0 = the null stream).  Every call is asynchronous
 *    on that stream and performs no device synchronisation and no allocation.
 *  - `ws`/`ws_bytes` is caller-provided scratch (device memory, 256-byte aligned);
 *    siv_workspace_bytes() returns a size that is sufficient for every call.
 *  - one stream at a time per workspace; calls are not re-entrant on one workspace.
 */
#ifndef SIV_H
#define SIV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIV_OK 0
#define SIV_EINVAL (-1)      /* bad argument (shape, null pointer, unsupported option) */
#define SIV_EWORKSPACE (-2)  /* workspace too small */
#define SIV_EHIP (-3)        /* a HIP runtime call or kernel launch failed */

typedef void* siv_stream_t;  /* hipStream_t */

const char* siv_last_error(void);
int siv_version(void);
/* Name of the kernel family the calling thread's last 3x3x3 convolution entry point dispatched to
 * ("wino2d", "wino2x4", "up", "up_wino", "wino", "wino_b3", "mfma", "thin", "generic"): lets tests assert that a
 * shape really ran on the kernel the automatic path is meant to pick.  No reference counterpart. */
const char* siv_last_conv_kernel(void);
/* Upper bound on the scratch any entry point needs for tensors up to the given size. */
size_t siv_workspace_bytes(void);
/* Kernel-path override: 0 = automatic (Winograd F(2x2,3x3) forward / input gradient and F(3x3,2x2)
 * weight gradient for large non-upsampled layers; behind an upsample collapsed taps (in product form forward,
 * box form for the input gradient) and the Winograd weight gradient; direct matrix-core kernels
 * elsewhere), 1 = force the generic (any-shape)
 * kernels, 2 = force the specialised kernels (error when the shape has none), 3 = Winograd F(2,3)
 * along W wherever it is supported, 4 = matrix-core kernels without Winograd, 5 = as 0, but the
 * forward / input-gradient convolutions with Cin % 8 == 0 and Cout % 64 == 0 form their fp32
 * products from three bf16 pieces per operand on the bf16 matrix cores (fp32 accumulate; error of
 * the order of fp32 rounding, not bf16; opt-in), 6 = the large-volume kernels of path 0 (two-dimensional Winograd forward, input gradient and
 * weight gradient, product-form / box-form upsampled forward / input gradient) wherever they are supported, whatever the volume.
 * Environment SIV_W2_W4 (read at every call): on volumes it covers with 8 x 8 brick planes the two-dimensional
 * Winograd kernel can apply F(4,3) instead of F(2,3) along W (9 instead of 12 multiply-adds per output; fp32
 * throughout, rounding error of the order of the direct kernel's, 2-3 x that of the F(2x2) form).  Unset or "2":
 * the input gradients do (default), "1": forward too (opt-in: forward rounding is what LeakyReLU kinks and
 * BatchNorm statistics amplify), "0": neither.  siv_last_conv_kernel() reports "wino2x4" for such a call. */
int siv_set_conv_path(int path);

/* ---- packed-weight cache ---------------------------------------------------------------------
 * The matrix-core convolution kernels read their weights in a packed (and, for the Winograd kernels,
 * transformed) order that a small kernel produces in front of every convolution launch.  Weights change
 * only at an optimiser step (utils/my_trainer.py:288,324) while one iteration convolves with them up to
 * eight times, so a caller may donate device memory in which the packings are kept between calls:
 *   siv_pack_cache_configure(mem, bytes)  donate `mem` (256-byte aligned; NULL = no cache) on the current device;
 *   siv_pack_cache_add_weights(lo, bytes) declare [lo, lo+bytes) as weight memory that changes only at
 *                                         siv_pack_cache_invalidate() calls (weights elsewhere are never cached);
 *   siv_pack_cache_remove_weights(lo)     the range that starts at lo is no longer weight memory (freed);
 *   siv_pack_cache_invalidate()           the weights changed: forget every packing;
 *   siv_pack_cache_invalidate_range(lo, bytes)  only the weights inside [lo, lo+bytes) changed (one optimiser's
 *                                         flat parameter buffer): their packings are marked stale and refilled
 *                                         in place at their next use, packings of other weights stay valid;
 *   siv_pack_cache_hits()                 number of convolution calls served from the cache (diagnostic).
 * The library allocates nothing; a full cache falls back to packing into the workspace.  Cached packings
 * are written and read in stream order: every CONVOLUTION call that may use the cache must be issued on one
 * stream.  The invalidate calls are host-side bookkeeping and may be made while another stream updates the
 * weights (the data-parallel path runs all-reduce + Adam on a side stream and invalidates that optimiser's
 * range at once): what the caller must guarantee is that the convolution stream waits for the update before
 * its next convolution with those weights -- the refill is then ordered behind the update.
 * No reference counterpart. */
int siv_pack_cache_configure(void* mem, size_t bytes);
int siv_pack_cache_add_weights(const void* lo, size_t bytes);
int siv_pack_cache_remove_weights(const void* lo);
int siv_pack_cache_invalidate(void);
int siv_pack_cache_invalidate_range(const void* lo, size_t bytes);
long siv_pack_cache_hits(void);

/* ---- 3x3x3 convolution, stride 1, zero padding 1 --------------------------------
 * Replaces nn.Conv3d(k=3,s=1,p=1) at models/models.py:17,21,55,59,92,137
 * (models/vaemodel.py:16,20,53,57,88,128).  With ups == 2 the input is first
 * nearest-upsampled x2, fusing nn.Upsample(scale_factor=2) of models/models.py:58:
 * x is then [N][D/2][H/2][W/2][Cin].  D,H,W are always the OUTPUT extents.
 * bias may be NULL. */
int siv_conv3d_k3_fwd(const float* x, const float* w, const float* bias, float* y,
                      int N, int D, int H, int W, int Cin, int Cout, int ups,
                      void* ws, size_t ws_bytes, siv_stream_t stream);
/* The same convolution followed by siv_bn_stats of its output (training-mode nn.BatchNorm3d
 * after nn.Conv3d, models/models.py:17-18,21-22,55-56,59-60; models/mymodel.py:56-57 etc.) in one
 * call: where the kernel family allows it the statistics are accumulated in the convolution's
 * epilogue and y is not read again.  mean / var / running_* / num_batches_tracked as in
 * siv_bn_stats.  The epilogue sums are unshifted fp32 per brick (fp64 across bricks): exact to 1e-4 of
 * the variance while |channel mean| stays within a few tens of standard deviations (fp32 cancellation
 * grows with (mean/std)^2 / sqrt(bricks); tested at 5, a bias-free convolution of normalised activations
 * sits near 1); siv_bn_stats, which shifts by a sample first, has no such limit. */
int siv_conv3d_k3_fwd_bnstats(const float* x, const float* w, const float* bias, float* y,
                              int N, int D, int H, int W, int Cin, int Cout, int ups,
                              float* mean, float* var, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float momentum,
                              void* ws, size_t ws_bytes, siv_stream_t stream);
/* Gradient w.r.t. the (non-upsampled) input of the same convolution: autograd of
 * nn.Conv3d.  dx is [N][D][H][W][Cin]; dy is [N][D][H][W][Cout]. */
int siv_conv3d_k3_dgrad(const float* dy, const float* w, float* dx,
                        int N, int D, int H, int W, int Cin, int Cout,
                        void* ws, size_t ws_bytes, siv_stream_t stream);
/* siv_conv3d_k3_dgrad_up: input gradient of siv_conv3d_k3_fwd with ups == 2, i.e. of
 * nn.Upsample(scale_factor=2, mode="nearest") followed by nn.Conv3d(k=3, p=1)
 * (models/models.py:57-59, models/mymodel.py:176-215), with respect to the LOW-RESOLUTION input:
 * dx[N][D/2][H/2][W/2][Cin] = 2x2x2 sum-pool of the gradient with respect to the upsampled tensor,
 * computed in one pass as a stride-2 convolution of dy[N][D][H][W][Cout] with the 4x4x4 kernel the
 * 27 taps collapse to (no full-resolution intermediate).  Needs Cin % 16 == 0 (tiles of 64 input
 * channels, the last one may be partial), Cout % 8 == 0 and 16-byte aligned pointers; siv_conv3d_k3_dgrad_up_supported returns 1 when the current conv path
 * and the channel counts allow it, otherwise callers use siv_conv3d_k3_dgrad + siv_upsample2_bwd. */
int siv_conv3d_k3_dgrad_up_supported(int Cin, int Cout);
int siv_conv3d_k3_dgrad_up(const float* dy, const float* w, float* dx, int N, int D, int H, int W,
                           int Cin, int Cout, void* ws, size_t ws_bytes, siv_stream_t stream);

/* Gradient w.r.t. weight (and bias when dbias != NULL).  x as in _fwd (ups honoured).
 * accumulate != 0 adds into dw/dbias instead of overwriting them. */
int siv_conv3d_k3_wgrad(const float* x, const float* dy, float* dw, float* dbias,
                        int N, int D, int H, int W, int Cin, int Cout, int ups, int accumulate,
                        void* ws, size_t ws_bytes, siv_stream_t stream);
/* The encoder stem as one operator: nn.Conv3d(1, Cout, 3, 1, 1) -> nn.BatchNorm3d -> nn.LeakyReLU ->
 * nn.Dropout (models/models.py:91-96, the stem of ResNetEncoder).  The convolution output costs 27
 * multiply-adds per element from the single-channel input, so it is never stored: the statistics
 * pass, the apply pass and both passes of the backward recompute it on the matrix pipe.
 * _fwd: x[N][D][H][W], w[Cout][27], out[N][D][H][W][Cout]; keep (uint8, layout of out) / p as in
 * siv_bn_act_fwd -- or keep == NULL with p > 0: the kernels draw the mask themselves, element i of out
 * from component i % 4 of Philox(i / 4, mask_stream_id [+ *mask_epoch * 2^20], mask_seed), the bits
 * siv_dropout_keep(_at) would have written, so no mask tensor exists (pass the same three values to
 * _bwd); train != 0: batch statistics are computed, written to mean / var[Cout] and folded
 * into the running statistics (as siv_bn_stats); train == 0: mean / var are inputs.
 * _bwd: dout[N][D][H][W][Cout] -> dy = gradient of the (never stored) convolution output, which is
 * what siv_conv3d_k3_wgrad / _dgrad of the stem take; dgamma / dbeta written or (accumulate = 1) added.
 * Cout % 4 == 0, Cout <= 64 (siv_conv1_bnact_supported). */
int siv_conv1_bnact_supported(int Cout);
int siv_conv1_bnact_fwd(const float* x, const float* w, const float* bias, const float* gamma,
                        const float* beta, const uint8_t* keep, uint64_t mask_seed, uint64_t mask_stream_id,
                        const uint64_t* mask_epoch, float p, float slope, float eps, int train,
                        float* mean, float* var, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float momentum, float* out,
                        int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_conv1_bnact_bwd(const float* dout, const float* x, const float* w, const float* bias,
                        const float* mean, const float* var, const float* gamma, const float* beta,
                        const uint8_t* keep, uint64_t mask_seed, uint64_t mask_stream_id,
                        const uint64_t* mask_epoch, float p, float slope, float eps, float* dy, float* dgamma,
                        float* dbeta, int accumulate, int N, int D, int H, int W, int Cout,
                        void* ws, size_t ws_bytes, siv_stream_t stream);
/* The same stem (models/models.py:91-96) with the dropout and activation decisions of the forward pass kept
 * as two bits per output element, so that the backward passes neither redraw the mask nor read a mask
 * tensor -- and, where the stem's input needs no gradient (it is the data or a detached reconstruction in
 * utils/my_trainer.py:249, 266-267), so that the backward is ONE pass over dout:
 * _fwd_bits = siv_conv1_bnact_fwd_bf16 (out_bf16 may be NULL) that also writes bits[_bits_bytes]: one byte
 *   per channel quad q of a voxel (byte index = element index of out / 4), bit j = channel 4q + j was kept
 *   by the dropout (all set without dropout), bit 4 + j = its BatchNorm output was > 0.
 * _bwd_bits = siv_conv1_bnact_bwd with the mask taken from bits (bit-identical results).
 * _wgrad_bits: dw[Cout][27], dbias[Cout] (of the convolution), dgamma, dbeta -- each optional, written or
 *   (accumulate = 1) added -- straight from dout, x and bits.  The gradient dy of the convolution output is
 *   never formed: with dz = dout * keep / (1 - p) * act'(z) and the input's own 28 x 28 tap correlation,
 *   BatchNorm's backward and the weight gradient are linear maps of T[c][tap] = sum_v dz[v][c] x[v + tap]
 *   (formulas at stem_wg_reduce_k in csrc/conv_thin.hip); sums over blocks and the combination in double
 *   precision.  Equals siv_conv1_bnact_bwd followed by siv_conv3d_k3_wgrad up to fp32 rounding. */
size_t siv_conv1_bnact_bits_bytes(int N, int D, int H, int W, int Cout);
int siv_conv1_bnact_fwd_bits(const float* x, const float* w, const float* bias, const float* gamma,
                             const float* beta, const uint8_t* keep, uint64_t mask_seed, uint64_t mask_stream_id,
                             const uint64_t* mask_epoch, float p, float slope, float eps, int train,
                             float* mean, float* var, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float momentum, float* out, void* out_bf16, void* bits,
                             int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_conv1_bnact_bwd_bits(const float* dout, const float* x, const float* w, const float* bias,
                             const float* mean, const float* var, const float* gamma, const float* beta,
                             const void* bits, float p, float slope, float eps, float* dy, float* dgamma,
                             float* dbeta, int accumulate, int N, int D, int H, int W, int Cout,
                             void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_conv1_bnact_wgrad_bits(const float* dout, const float* x, const float* w, const float* bias,
                               const float* mean, const float* var, const float* gamma, const void* bits,
                               float p, float slope, float eps, float* dw, float* dbias, float* dgamma,
                               float* dbeta, int accumulate, int N, int D, int H, int W, int Cout,
                               void* ws, size_t ws_bytes, siv_stream_t stream);

/* BatchNorm3d + LeakyReLU folded into the Cin -> 1 convolution behind them: the decoder's last block
 * ends in conv -> BatchNorm -> LeakyReLU and feeds the output layer nn.Conv3d(in_ch, 1, 3, 1, 1)
 * (models/models.py:76-78 and :137).  x is the block's RAW convolution output [N][D][H][W][Cin]; the
 * kernels apply z = x*a + b (a = gamma/sqrt(var+eps), b = beta - mean*a), LeakyReLU(slope) and the zero
 * padding of the activated tensor while they load x, so the activated full-resolution tensor is never
 * written or read.  _fwd: y[N][D][H][W] = conv(act(bn(x)), w[1][27][Cin]) + bias.  _wgrad: dw[1][27][Cin]
 * (+ dbias[1]) from dy[N][D][H][W] and the same x.  The gradient with respect to the activation is the
 * plain siv_conv3d_k3_dgrad of dy; BatchNorm's own backward (siv_bn_act_bwd) takes it from there --
 * or _dgrad does both at once: dx = gradient of the RAW tensor x (what siv_bn_act_bwd would return as
 * dy), dgamma / dbeta written or, with accumulate = 1, added to; the gradient of the activation is
 * recomputed in registers by both of its passes (27 multiply-adds per element on the matrix pipe)
 * instead of being written once and read twice.  Cin % 4 == 0, Cin <= 64.
 * _supported: 1 when _fwd has a kernel for the shape (64 channels, D >= 16, H >= 12, W >= 16). */
int siv_bnact_conv3d_k3_to1_supported(int N, int D, int H, int W, int Cin);
int siv_bnact_conv3d_k3_to1_fwd(const float* x, const float* mean, const float* var, const float* gamma,
                                const float* beta, float eps, float slope, const float* w,
                                const float* bias, float* y, int N, int D, int H, int W, int Cin,
                                siv_stream_t stream);
int siv_bnact_conv3d_k3_to1_dgrad(const float* dy, const float* w, const float* x, const float* mean,
                                  const float* var, const float* gamma, const float* beta, float eps,
                                  float slope, float* dx, float* dgamma, float* dbeta, int accumulate,
                                  int N, int D, int H, int W, int Cin, void* ws, size_t ws_bytes,
                                  siv_stream_t stream);
int siv_bnact_conv3d_k3_to1_wgrad(const float* x, const float* mean, const float* var, const float* gamma,
                                  const float* beta, float eps, float slope, const float* dy, float* dw,
                                  float* dbias, int N, int D, int H, int W, int Cin, int accumulate,
                                  void* ws, size_t ws_bytes, siv_stream_t stream);

/* Backward of nn.Upsample(scale_factor=2, nearest): dx[n][d][h][w][c] = sum of the
 * 2x2x2 children of dy.  D,H,W are the extents of dy (even). */
int siv_upsample2_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C,
                      siv_stream_t stream);

/* ---- 3x3x3 transposed convolution, stride 1, padding 1 ---------------------------------------
 * Replaces nn.ConvTranspose3d(k=3, s=1, p=1) at models/model.py:202-205 (Decoder_lucky) and its
 * autograd.  x is [N][D][H][W][Cin], y and dy are [N][D][H][W][Cout]; w is [Cin][kd][kh][kw][Cout]
 * (channels_last_3d memory of the reference's (Cin,Cout,3,3,3) parameter); bias (Cout) may be NULL.
 * With stride 1 the operation is the adjoint of siv_conv3d_k3_fwd, so the three calls run on the
 * same kernels as siv_conv3d_k3_dgrad / _fwd / _wgrad.  accumulate as in siv_conv3d_k3_wgrad. */
int siv_conv_transpose3d_k3_fwd(const float* x, const float* w, const float* bias, float* y,
                                int N, int D, int H, int W, int Cin, int Cout,
                                void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_conv_transpose3d_k3_dgrad(const float* dy, const float* w, float* dx,
                                  int N, int D, int H, int W, int Cin, int Cout,
                                  void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_conv_transpose3d_k3_wgrad(const float* x, const float* dy, float* dw, float* dbias,
                                  int N, int D, int H, int W, int Cin, int Cout, int accumulate,
                                  void* ws, size_t ws_bytes, siv_stream_t stream);

/* ---- 1x1x1 convolution (pointwise linear map over channels) ---------------------
 * Replaces nn.Conv3d(k=1) at models/models.py:35,73 (shortcut projections), :118
 * (decoder stem) and :216-217 (mu / var heads).  M = N*D*H*W rows. */
int siv_conv1x1_fwd(const float* x, const float* w, const float* bias, float* y,
                    long M, int Cin, int Cout, siv_stream_t stream);
int siv_conv1x1_dgrad(const float* dy, const float* w, float* dx,
                      long M, int Cin, int Cout, siv_stream_t stream);
int siv_conv1x1_wgrad(const float* x, const float* dy, float* dw, float* dbias,
                      long M, int Cin, int Cout, int accumulate,
                      void* ws, size_t ws_bytes, siv_stream_t stream);

/* ---- fully connected layers of the FC-latent model ---------------------------------
 * y[b][o] = bias[o] + sum_k x[b][k] w[o][k]: nn.Linear of models/mymodel.py:125 (encoder
 * head, K = forth_ch*150, O = 2*z) and :151 (decoder stem, K = z, O = forth_ch*150).
 * x is [B][K] row-major in the reference's flatten order (channel-major), w is [O][K].
 * Forward and input gradient split the weight matrix over ~4000 waves (K slabs / output-row slabs)
 * and reduce the partial sums in fixed order: both need the workspace. */
int siv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int O,
                   void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int O,
                     void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, int B, int K, int O,
                     int accumulate, siv_stream_t stream);
/* out = LeakyReLU_slope(a + b): the skip connections `self.Leakyrelu5(x + h)` etc. of
 * models/mymodel.py:136,139,222,225; backward d = dout * act'(a + b) is the gradient of both
 * a and b. */
int siv_add_act_fwd(const float* a, const float* b, float slope, float* out, long n,
                    siv_stream_t stream);
int siv_add_act_bwd(const float* dout, const float* a, const float* b, float slope, float* d,
                    long n, siv_stream_t stream);

/* ---- input pipeline ------------------------------------------------------------------
 * siv_preprocess: BrainDataset._preprocess of utils/data_load.py:25-30 for a batch of B raw
 * volumes of n voxels each: v = clip(v, 0, cut_range * std(v)) (population std, cut_range = 4
 * in the reference), then (v - min(v)) / (max(v) - min(v)); a constant volume yields NaN as
 * numpy does.  voxels and out are [B][n] fp32 and must not overlap. */
int siv_preprocess(const float* voxels, float* out, int B, long n, float cut_range,
                   void* ws, size_t ws_bytes, siv_stream_t stream);
/* siv_flip_shift: per-volume axis flips and integer shifts with zero fill, the flip/shift
 * augmentation the 160x192x160 benchmark configuration names.  It stands in the slot of the
 * transform applied before _preprocess at utils/data_load.py:21-22 (the reference fills that slot
 * with torchio's RandomAffine, aug-z-1200main.py:106-119).  params is int[B][6] on the device:
 * flipD, flipH, flipW (0/1), shiftD, shiftH, shiftW (voxels, toward larger indices; the flip is
 * applied first).  in and out are [B][D][H][W] fp32 and must not alias. */
int siv_flip_shift(const float* in, const int* params, float* out, int B, int D, int H, int W,
                   siv_stream_t stream);
/* siv_affine_resample: trilinear resampling of each volume under an affine map of voxel indices,
 * the device form of the transform the reference applies before _preprocess
 * (utils/data_load.py:21-22 with tio.transforms.RandomAffine(degrees=10), aug-z-1200main.py:106-119:
 * small random rotation and scaling about the image centre, linear interpolation, padding with the
 * image minimum).  mats is float[B][12] on the device: per volume the row-major 3x4 matrix that
 * maps an OUTPUT index (d, h, w, 1) to the INPUT index sampled.  Each of the 8 neighbours that
 * falls outside the volume contributes the pad value: pad_value when pad_mode == 0, the minimum of
 * that volume when pad_mode == 1 (ws: B * 64 * 32 bytes, 8-byte aligned; unused for mode 0).
 * in and out are [B][D][H][W] fp32 and must not alias. */
int siv_affine_resample(const float* in, const float* mats, float* out, int B, int D, int H, int W,
                        int pad_mode, float pad_value, void* ws, size_t ws_bytes,
                        siv_stream_t stream);

/* ---- BatchNorm3d (+ residual + activation + AvgPool3d + Dropout) ----------------
 * siv_bn_stats: training-mode batch statistics of nn.BatchNorm3d (models/models.py:18,22,
 * 56,60,93,119): per-channel mean and BIASED variance over the M = N*D*H*W rows of y,
 * and, when running_mean != NULL, the running-statistics update
 *   running = (1-momentum)*running + momentum*stat   (unbiased variance, M/(M-1)),
 *   num_batches_tracked += 1  (int64, may be NULL). */
int siv_bn_stats(const float* y, long M, int C, float* mean, float* var,
                 float* running_mean, float* running_var, int64_t* num_batches_tracked,
                 float momentum, void* ws, size_t ws_bytes, siv_stream_t stream);
/* out = dropout( avgpool_k( act( gamma*(y-mean)/sqrt(var+eps)+beta [+ residual] ) ) )
 *  - act: LeakyReLU(slope) (slope 0.2: models/models.py:15,19,94,121; slope 0: nn.ReLU of
 *    models/vaemodel.py:14,18,90,113)
 *  - pool: 1 or 2 = nn.AvgPool3d(kernel_size=stride) of models/models.py:20; out is then
 *    [N][D/2][H/2][W/2][C]
 *  - residual (NULL or [N][D][H][W][C]): the `+ shortcut` of models/models.py:41,78
 *  - keep (NULL or uint8 0/1 per output element) and p: nn.Dropout(p) of models/models.py:95,122
 *    in training mode, out *= keep/(1-p)
 * Evaluation mode passes the running statistics as mean/var. */
int siv_bn_act_fwd(const float* y, const float* mean, const float* var,
                   const float* gamma, const float* beta, const float* residual,
                   const uint8_t* keep, float p, float slope, int pool, float eps,
                   float* out, int N, int D, int H, int W, int C, siv_stream_t stream);
/* Backward of siv_bn_act_fwd in training mode (batch statistics take part in the
 * gradient, as in autograd of nn.BatchNorm3d).  dy and dresidual may alias nothing;
 * dgamma/dbeta/dresidual may be NULL.  accumulate != 0 adds into dgamma/dbeta. */
int siv_bn_act_bwd(const float* dout, const float* y, const float* mean, const float* var,
                   const float* gamma, const float* beta, const float* residual,
                   const uint8_t* keep, float p, float slope, int pool, float eps,
                   float* dy, float* dgamma, float* dbeta, float* dresidual, int accumulate,
                   int N, int D, int H, int W, int C,
                   void* ws, size_t ws_bytes, siv_stream_t stream);
/* The same three operators for a batch that consists of `groups` equal parts which the reference sends through
 * the layer in SEPARATE calls with the same weights -- model.forward(rec.detach()) then model.forward(fake.detach()),
 * decode(noise) then decode(z), encode(rec) then encode(fake), decode(z_rec) then decode(z_fake)
 * (utils/my_trainer.py:266-267, 297-298, 304-311) -- and which are run here as one batch so that the small deep
 * layers fill the chip: part g (samples [g*N/groups, (g+1)*N/groups)) is normalised with its own batch statistics
 * mean[g][C], var[g][C]; the running statistics receive one update per part, in part order, and
 * num_batches_tracked += groups, exactly as after `groups` consecutive calls; dgamma / dbeta are the sums over the
 * parts.  M of siv_bn_stats_grouped is the number of rows of ONE part; N of the other two is the whole batch
 * (N % groups == 0).  groups == 1 is the plain operator. */
int siv_bn_stats_grouped(const float* y, long M, int C, int groups, float* mean, float* var,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float momentum, void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_bn_act_fwd_grouped(const float* y, const float* mean, const float* var,
                           const float* gamma, const float* beta, const float* residual,
                           const uint8_t* keep, float p, float slope, int pool, float eps,
                           float* out, int N, int D, int H, int W, int C, int groups, siv_stream_t stream);
int siv_bn_act_bwd_grouped(const float* dout, const float* y, const float* mean, const float* var,
                           const float* gamma, const float* beta, const float* residual,
                           const uint8_t* keep, float p, float slope, int pool, float eps,
                           float* dy, float* dgamma, float* dbeta, float* dresidual, int accumulate,
                           int N, int D, int H, int W, int C, int groups,
                           void* ws, size_t ws_bytes, siv_stream_t stream);
/* bf16 activation storage inside the BatchNorm passes (no reference counterpart: the reference trains in fp32,
 * utils/my_trainer.py:242-325; this is the "bf16 activations" of the 160x192x160 configuration).  The two
 * tensors a conv -> BatchNorm -> activation layer keeps for its backward pass are the convolution output y (for
 * the BatchNorm backward) and the activated output (the next convolution's input, for its weight gradient).
 * siv_bn_act_fwd_bf16 is siv_bn_act_fwd_grouped that ALSO writes bf16 copies (round-to-nearest-even) of y
 * (y_bf16, NULL = none) and of out (out_bf16, NULL = none) in the same pass, so that the fp32 tensors need not
 * be kept and no separate conversion pass reads them again; siv_bn_act_bwd_bf16 is siv_bn_act_bwd_grouped
 * reading y from its bf16 copy (half the bytes in both of its passes).  Arithmetic is fp32 throughout.  Both
 * need the float4 kernels: siv_bn_act_bf16_supported(C) (C = 4..1024, a power of two), 16-byte aligned tensors. */
int siv_bn_act_bf16_supported(int C);
int siv_bn_act_fwd_bf16(const float* y, const float* mean, const float* var,
                        const float* gamma, const float* beta, const float* residual,
                        const uint8_t* keep, float p, float slope, int pool, float eps,
                        float* out, void* y_bf16, void* out_bf16, int N, int D, int H, int W, int C,
                        int groups, siv_stream_t stream);
/* The same for the two fused ends of the model.  siv_conv1_bnact_fwd_bf16 = siv_conv1_bnact_fwd that also writes
 * the bf16 copy of its output (out_bf16, NULL = none), which the first wide convolution keeps for its weight
 * gradient.  siv_bnact_conv3d_k3_to1_fwd_bf16 = siv_bnact_conv3d_k3_to1_fwd that also writes the bf16 copy of x, the
 * raw tensor it normalises (x_bf16, NULL = none; every voxel is written once, by the block that owns it);
 * ..._dgrad_bf16 / ..._wgrad_bf16 read x from that copy. */
int siv_conv1_bnact_fwd_bf16(const float* x, const float* w, const float* bias, const float* gamma,
                             const float* beta, const uint8_t* keep, uint64_t mask_seed,
                             uint64_t mask_stream_id, const uint64_t* mask_epoch, float p, float slope,
                             float eps, int train, float* mean, float* var, float* running_mean,
                             float* running_var, int64_t* num_batches_tracked, float momentum,
                             float* out, void* out_bf16, int N, int D, int H, int W, int Cout, void* ws,
                             size_t ws_bytes, siv_stream_t stream);
int siv_bnact_conv3d_k3_to1_fwd_bf16(const float* x, const float* mean, const float* var,
                                     const float* gamma, const float* beta, float eps, float slope,
                                     const float* w, const float* bias, float* y, void* x_bf16, int N,
                                     int D, int H, int W, int Cin, siv_stream_t stream);
int siv_bnact_conv3d_k3_to1_dgrad_bf16(const float* dy, const float* w, const void* x_bf16,
                                       const float* mean, const float* var, const float* gamma,
                                       const float* beta, float eps, float slope, float* dx,
                                       float* dgamma, float* dbeta, int accumulate, int N, int D, int H,
                                       int W, int Cin, void* ws, size_t ws_bytes, siv_stream_t stream);
int siv_bnact_conv3d_k3_to1_wgrad_bf16(const void* x_bf16, const float* mean, const float* var,
                                       const float* gamma, const float* beta, float eps, float slope,
                                       const float* dy, float* dw, float* dbias, int N, int D, int H,
                                       int W, int Cin, int accumulate, void* ws, size_t ws_bytes,
                                       siv_stream_t stream);
int siv_bn_act_bwd_bf16(const float* dout, const void* y_bf16, const float* mean, const float* var,
                        const float* gamma, const float* beta, const float* residual,
                        const uint8_t* keep, float p, float slope, int pool, float eps,
                        float* dy, float* dgamma, float* dbeta, float* dresidual, int accumulate,
                        int N, int D, int H, int W, int C, int groups,
                        void* ws, size_t ws_bytes, siv_stream_t stream);
/* out = dropout(act(x)) without normalisation: the decoder tail nn.ReLU + nn.Dropout(0.35)
 * of models/models.py:139-140, and its backward (dx = dout * keep/(1-p) * act'(x)). */
int siv_act_dropout_fwd(const float* x, const uint8_t* keep, float p, float slope,
                        float* out, long n, siv_stream_t stream);
int siv_act_dropout_bwd(const float* dout, const float* x, const uint8_t* keep, float p,
                        float slope, float* dx, long n, siv_stream_t stream);

/* ---- latent sampling and losses -------------------------------------------------
 * z = mu + eps*exp(0.5*logvar): SoftIntroVAE.reparameterize, models/models.py:263-271.
 * eps == NULL uses the constant eps_const (0.1 in validation, models/models.py:269). */
int siv_reparam_fwd(const float* mu, const float* logvar, const float* eps, float eps_const,
                    float* z, long n, siv_stream_t stream);
int siv_reparam_bwd(const float* dz, const float* logvar, const float* eps, float eps_const,
                    float* dmu, float* dlogvar, long n, siv_stream_t stream);
/* out[b] = -0.5 * sum_i (1 + logvar - mu^2 - exp(logvar)) over the n latent elements of
 * sample b: calc_kl(reduce='none'), utils/my_trainer.py:38-48 (models/lossf.py:14-18). */
int siv_kl_fwd(const float* mu, const float* logvar, float* out, int B, long n, siv_stream_t stream);
/* dmu = g[b]*mu, dlogvar = g[b]*0.5*(exp(logvar)-1). */
int siv_kl_bwd(const float* g, const float* mu, const float* logvar, float* dmu, float* dlogvar,
               int B, long n, siv_stream_t stream);
/* out[b] = sum_i (x-y)^2 over the n elements of sample b: calc_reconstruction_loss with
 * reduction != 'mean', utils/my_trainer.py:62-78 (models/lossf.py:5-12). */
int siv_sse_fwd(const float* x, const float* y, float* out, int B, long n,
                void* ws, size_t ws_bytes, siv_stream_t stream);
/* siv_sse_fwd without a workspace and without any allocation: the per-block partial sums go to a
 * 512 KiB device array built into the library (one copy per GPU, held until the process ends).
 * No counterpart in the reference; it is for callers that cannot allocate or carry a workspace:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same result
 * as siv_sse_fwd bit for bit.  At most 256 samples per call (SIV_EINVAL beyond).  The first call on a
 * device looks the array's address up (make it outside graph capture); calls that overlap in time on
 * one GPU share the array and race, so issue them on one stream. */
int siv_sse_fwd_noalloc(const float* x, const float* y, float* out, int B, long n,
                        siv_stream_t stream);
/* dx = 2 g[b] (x-y), dy = -dx; either output may be NULL. */
int siv_sse_bwd(const float* g, const float* x, const float* y, float* dx, float* dy,
                int B, long n, siv_stream_t stream);

/* ---- the two objectives of one Soft-IntroVAE iteration --------------------------------------------
 * The scalar algebra the reference spells with a dozen torch operations per step, from the per-sample
 * vectors siv_sse_fwd / siv_kl_fwd produce (all of length B, device memory):
 *  E-step, utils/my_trainer.py:260-284:
 *    lossE = 10 * ( scale*(beta_rec*mean(sse_real) + beta_kl*mean(kl_real))
 *                   + 0.5*( mean_b exp(-2 scale (beta_rec*sse_fake + beta_neg*kl_fake))
 *                         + mean_b exp(-2 scale (beta_rec*sse_rec  + beta_neg*kl_rec )) ) )
 *    out4 = {lossE, mean(kl_real), exp_elbo_rec, exp_elbo_fake};  _bwd: d6 = [6][B], g[0] times the
 *    derivative of lossE by sse_real, kl_real, sse_fake, kl_fake, sse_rec, kl_rec.
 *  D-step, utils/my_trainer.py:301-321:
 *    lossD = 10 * scale * ( beta_rec*mean(sse_real) + 0.5*beta_kl*(mean(kl_rec) + mean(kl_fake))
 *                           + gamma_r*0.5*beta_rec*(mean(sse_rec_rec) + mean(sse_fake_rec)) )
 *    out4 = {lossD, mean(kl_fake), mean(kl_rec), mean(sse_real)};  _bwd: d5 = [5][B] in argument order. */
int siv_sintro_loss_e_fwd(const float* sse_real, const float* kl_real, const float* sse_fake,
                          const float* kl_fake, const float* sse_rec, const float* kl_rec, int B,
                          float scale, float beta_rec, float beta_neg, float beta_kl, float* out4,
                          siv_stream_t stream);
int siv_sintro_loss_e_bwd(const float* g, const float* sse_fake, const float* kl_fake,
                          const float* sse_rec, const float* kl_rec, int B, float scale, float beta_rec,
                          float beta_neg, float beta_kl, float* d6, siv_stream_t stream);
int siv_sintro_loss_d_fwd(const float* sse_real, const float* kl_rec, const float* kl_fake,
                          const float* sse_rec_rec, const float* sse_fake_rec, int B, float scale,
                          float beta_rec, float beta_kl, float gamma_r, float* out4, siv_stream_t stream);
int siv_sintro_loss_d_bwd(const float* g, int B, float scale, float beta_rec, float beta_kl, float gamma_r,
                          float* d5, siv_stream_t stream);

/* ---- optimiser -------------------------------------------------------------------
 * One torch.optim.Adam step (lr, betas, eps; no weight decay, no amsgrad) over a flat
 * parameter buffer: replaces optimizer_e.step()/optimizer_d.step(),
 * utils/my_trainer.py:183-184,288,324.  `step` is the 1-based step count. */
int siv_adam_step(float* p, const float* g, float* m, float* v, long n,
                  float lr, float beta1, float beta2, float eps, int step, siv_stream_t stream);

/* siv_adam_step with the learning rate (float) and the 1-based step count (int64) read from device
 * memory at execution time, so that a launch captured in a HIP graph keeps following the schedule
 * (utils/my_trainer.py:185-186, 502-503) and the bias corrections on every replay.
 * siv_counter_add: *counter += inc on the stream (advances such a step count, or the random epoch below,
 * from inside a captured sequence). */
int siv_adam_step_at(float* p, const float* g, float* m, float* v, long n, const float* lr,
                     float beta1, float beta2, float eps, const int64_t* step, siv_stream_t stream);
int siv_counter_add(int64_t* counter, int64_t inc, siv_stream_t stream);

/* ---- device random numbers -------------------------------------------------------
 * Philox-4x32-10 counter RNG keyed by (seed, stream_id); element i of a call uses
 * counter i/4.  Replaces torch.randn / torch.randn_like (utils/my_trainer.py:238,
 * models/models.py:266) and the Bernoulli draw inside nn.Dropout.
 * keep[i] = 1 with probability 1-p. */
int siv_randn(float* out, long n, uint64_t seed, uint64_t stream_id, siv_stream_t stream);
int siv_dropout_keep(uint8_t* keep, long n, float p, uint64_t seed, uint64_t stream_id,
                     siv_stream_t stream);
/* The same with stream_id + (*epoch << 20) as the stream id, *epoch (uint64, device memory, may be NULL)
 * read at execution time: launches captured in a HIP graph draw fresh numbers on every replay once
 * the epoch has been advanced (siv_counter_add). */
int siv_randn_at(float* out, long n, uint64_t seed, uint64_t stream_id, const uint64_t* epoch,
                 siv_stream_t stream);
int siv_dropout_keep_at(uint8_t* keep, long n, float p, uint64_t seed, uint64_t stream_id,
                        const uint64_t* epoch, siv_stream_t stream);

/* ---- helpers ------------------------------------------------------------------------ */
/* y = a*x + b*y elementwise (gradient accumulation on device); x and y must not overlap. */
int siv_axpby(float a, const float* x, float b, float* y, long n, siv_stream_t stream);
/* bf16 storage of the tensors the backward pass keeps (no counterpart in the reference, which trains in
 * fp32 throughout; this is the "bf16 activations" of the 160x192x160 configuration): h[i] = bf16(x[i])
 * with round-to-nearest-even, and back (exact).  n elements in memory order, both buffers 16-byte
 * aligned.  Arithmetic stays fp32 everywhere: only what autograd saves between forward and backward is
 * held at half the size. */
int siv_f32_to_bf16(const float* x, void* h, long n, siv_stream_t stream);
int siv_bf16_to_f32(const void* h, float* x, long n, siv_stream_t stream);
/* x *= a in place: the division by the world size after the gradient all-reduce that replaces
 * nn.DataParallel's reduce_add (main_DataParallel.py:609).  The collective itself is not part of this
 * library: callers run it with torch.distributed (backend "nccl" = RCCL over xGMI) on the same stream. */
int siv_scale(float a, float* x, long n, siv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIV_H */
