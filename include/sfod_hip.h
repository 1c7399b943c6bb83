/* libsfod_hip.so -- C ABI of the MI355X (gfx950) teacher-student Faster R-CNN hot path.
 *
 * The reference (EPFL-IMOS/simple-SFOD) has no FFI: its hot path is Python calling
 * Detectron2 / torchvision / torch device ops.  Each entry point below replaces the device op
 * the reference reaches at the cited site (paths relative to /root/reference; "d2:" / "tv:" =
 * upstream Detectron2 / torchvision semantics restated in SURVEY.md Appendix A).
 *
 * Conventions
 *   - every pointer is a BORROWED device pointer (tensor.data_ptr()); the caller keeps the
 *     memory alive until the stream is synchronised.  The library never allocates.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is enqueued on it,
 *     nothing synchronises.
 *   - return value: 0 on success, negative hipError_t otherwise, -1000 for bad arguments.
 *   - activations are NHWC ("pixel-major rows, channels contiguous"); `dt` selects the storage /
 *     MFMA input type: SFOD_F32 (v_mfma_f32_32x32x2_f32), SFOD_BF16 (reduced precision:
 *     v_mfma_f32_32x32x16_bf16, fp32 accumulate) or SFOD_BF16X3 (fp32-equivalent arithmetic on the
 *     bf16 matrix pipe, the default mode): every MFMA operand element v is stored as the pair
 *     hi = bf16(v), lo = bf16(v - hi) (|v - hi - lo| <= 2^-16 |v|) and a product is accumulated in fp32 as
 *     hi*hi + hi*lo + lo*hi (three v_mfma_f32_32x32x16_bf16; the dropped lo*lo term is <= 2^-16 relative).
 *     Storage of a SFOD_BF16X3 tensor [.., C] (C a multiple of 8): 4 bytes per logical element, every group
 *     of 8 consecutive channels is 32 bytes = 8 x bf16 hi followed by 8 x bf16 lo.  Convolutions / GEMMs take
 *     SFOD_BF16X3 operands and write fp32; the elementwise producers (preprocess, BatchNorm apply, BatchNorm
 *     backward, ROIAlign, cast) write the pairs.
 *     SFOD_F16X3: the same storage and the same three-MFMA product with IEEE half pairs, hi = f16(v),
 *     lo = f16(v - hi) (v_mfma_f32_32x32x16_f16; subnormal inputs are kept): 22 significand bits per operand for
 *     |v| in [2^-3, 65504], an absolute 2^-25 below, +-65504 saturation above -- the operand format of the FORWARD
 *     products of cfg.SFOD.COMPUTE_DTYPE = "f16x3" (activations and weights, whose magnitudes sit in that window;
 *     gradients do not, the backward products of that mode use SFOD_BF16X3).  Served by the forward convolution /
 *     GEMM entry points and by every producer of operand pairs; the weight-gradient entry points take SFOD_BF16X3.
 *   - per-image variable-length results live in fixed-capacity arrays plus an int32 count per
 *     image, so that no entry point needs a host round trip.
 */
#ifndef SFOD_HIP_H
#define SFOD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SFOD_F32 0
#define SFOD_BF16 1
#define SFOD_BF16X3 2
#define SFOD_F16X3 3

int sfod_version(void);
/* last error text of the calling thread ("" if none) */
const char* sfod_last_error(void);

/* ---- K1: preprocess.  d2 GeneralizedRCNN.preprocess_image (normalise, pad) as called at
 * daod/modeling/meta_arch/source_free_adaptive_teacher_rcnn.py:212 ; uint8 CHW BGR -> NHWC.
 * img_ptrs: device array of B pointers to uint8 [3,h_i,w_i]; sizes: int32 [B,2] (h,w);
 * out: [B,Hp,Wp,Cpad] (channels >= 3 are zero, padding pixels are zero). */
int sfod_preprocess(const void* const* img_ptrs, const int32_t* sizes, int B, int Hp, int Wp,
                    int Cpad, const float* mean3, const float* std3, void* out, int dt,
                    void* stream);

/* horizontal flip of a uint8 [C,H,W] image: the RandomFlip(0.5, horizontal) of the weak augmentation
 * (d2 build_augmentation; daod/data/mappers/two_crop_augmentation_mapper.py:73-157), on device */
int sfod_hflip_u8(const void* src, void* dst, int C, int H, int W, void* stream);

/* ResizeShortestEdge on a uint8 [C,H,W] frame -> [C,h,w], bit-exact with Pillow's Image.resize(..., BILINEAR)
 * (what d2's ResizeTransform calls for uint8 images; detection_utils / two_crop_augmentation_mapper.py:73-157):
 * separable support-scaled triangle filter, 22-bit fixed-point coefficients, uint8 clip between the passes.
 * hbounds/vbounds: int32 [out][2] = (first source index, tap count); hcoef/vcoef: int32 [out][ksize] fixed-point
 * weights -- built on the host like Pillow's precompute_coeffs + normalize_coeffs_8bpc.  flip != 0 also mirrors
 * the result horizontally (RandomFlip follows the resize). */
int sfod_resize_bilinear_u8(const void* src, void* dst, int C, int H, int W, int h, int w,
                            const int32_t* hbounds, const int32_t* hcoef, int ksize_h,
                            const int32_t* vbounds, const int32_t* vcoef, int ksize_v, int flip,
                            void* stream);

/* ---- K2/K5/K14/K18: implicit-GEMM convolution / linear layer on MFMA.
 * y[m, n] = act( sum_{tap, c} x[pix(m) + tap][c] * w[n][tap][c] + bias[n] ),  m = (b, oy, ox)
 * x: [B,H,W,Cin] NHWC, w: packed [Cout][KH*KW][Cin] (K contiguous), y: [B,H,W,ldy] with ldy >=
 * Cout.  KH=KW=3 pad 1 (vgg.py:18 conv3x3; d2 StandardRPNHead.conv; dann.py:14-17) or KH=KW=1
 * (1x1 convs and nn.Linear: x [R,K] -> y [R,N]).  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).
 * stats (optional, fp32 [nblk][2][Cout] followed by [nblk] row counts, nblk from
 * sfod_conv_stats_blocks): per-row-block (sum, M2, count) of the pre-activation output for the
 * train-mode BatchNorm that follows (vgg.py:20).  out_dt may differ from dt (e.g. fp32
 * logits out of a bf16 head).  Also used for data-gradient (dgrad) with rotated weights. */
int sfod_conv_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W,
                  int Cin, int Cout, int ksize, int ldy, int act, float* stats, int dt, int out_dt,
                  void* stream);
/* the same for SFOD_F16X3 weights packed with their per-tensor power-of-two scale (sfod_pack_*_ws below): w_absmax is the
 * device word the packer published (bits of max|w|); the kernels multiply their accumulators by the inverse scale before
 * bias / statistics / activation (exact).  w_absmax == NULL: unscaled weights (= sfod_conv_fwd). */
int sfod_conv_fwd_ws(const void* x, const void* w, const uint32_t* w_absmax, const float* bias, void* y, int B, int H,
                     int W, int Cin, int Cout, int ksize, int ldy, int act, float* stats, int dt, int out_dt,
                     void* stream);
/* the same with device scratch the library may use (sfod_conv_fwd_scratch_bytes for the shape; 0 = none wanted, and the
 * call is then sfod_conv_fwd_ws).  Today: linear layers (ksize 1, fp32 output, no statistics) with so few rows that their
 * grid fills less than half the chip -- the ROI head's fc1 at one frame per GPU, 512 x 25088 -> 1024 -- run split along K;
 * the scratch holds the S partial outputs (S x rows x Cout floats), summed in index order with bias and activation by a
 * second launch (run-to-run identical; differs from the unsplit kernel by fp32 summation order).  scratch == NULL or too
 * small: the unsplit kernel.  SFOD_GEMM_SPLITK=0 in the environment switches the split form off (A/B runs). */
int64_t sfod_conv_fwd_scratch_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int dt, int out_dt, int with_stats);
int sfod_conv_fwd_scratch(const void* x, const void* w, const uint32_t* w_absmax, const float* bias, void* y, int B, int H,
                          int W, int Cin, int Cout, int ksize, int ldy, int act, float* stats, int dt, int out_dt,
                          void* scratch, int64_t scratch_bytes, void* stream);
/* number of statistics blocks nblk sfod_conv_fwd writes for this layer shape; `stats` holds
 * nblk * (2*Cout + 1) floats */
int sfod_conv_stats_blocks(int B, int H, int W, int Cin, int Cout, int ksize, int dt);
/* kernel selection for 3x3 bf16 convolutions: 0 auto, 1 generic implicit GEMM (im2col chunks
 * DMA'd per tap), 2 halo-patch kernel (input patch staged once per 32-channel slice and reused by
 * all nine taps) whenever the shape allows, 3 / 4 auto without / with the wide-tile bf16x3
 * weight gradient wherever it applies.  For A/B tests
 * and benchmarks. */
int sfod_set_conv_algo(int algo);
/* Data gradient of a 3x3 convolution with the BatchNorm-backward REDUCTION of the layer below folded into its epilogue
 * (the hand-written backward of vgg.py's conv-BN-ReLU chain; torch autograd runs these as separate kernels).
 * x: the upper layer's dy [B,H,W,Cin] (operand dtype dt), w: its rotated weights, dz [B,H,W,Cout] fp32 (dense): the
 * gradient w.r.t. the lower layer's activated output.  y / mean / invstd / gamma / beta: the lower layer's saved
 * pre-BatchNorm output (fp32 [B,H,W,Cout]) and BatchNorm inputs; red_ws receives sfod_conv_dgrad_bnred_blocks(...) rows
 * of 2 * Cout partial sums for sfod_bn_relu_pool_bwd(reduced_blocks = that count).  _blocks returns 0 when the shape /
 * dtype is not served by the halo-patch kernel: use sfod_conv_fwd and the unfused BatchNorm backward then. */
int sfod_conv_dgrad_bnred_blocks(int B, int H, int W, int Cin, int Cout, int dt);
int sfod_conv_dgrad_bnred(const void* x, const void* w, void* dz, int B, int H, int W, int Cin, int Cout, int dt,
                          const float* y, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, float* red_ws, void* stream);
/* K2 with the PRODUCER's BatchNorm + ReLU folded into the operand path (vgg.py:18-20: conv -> BatchNorm2d -> ReLU -> conv, the
 * activated tensor consumed by nobody else: forward-only passes such as the teacher's, source_free_adaptive_teacher.py:385-390).
 * x_pre: the producer's pre-BatchNorm output, fp32 [B,H,W,Cin]; in_mean / in_invstd / in_gamma / in_beta [Cin]: its batch
 * statistics and affine parameters.  The kernel computes relu((x - mean) * (invstd * gamma) + beta) and the (hi, lo) split of
 * SFOD_BF16X3 -- the arithmetic of sfod_bn_relu_pool_fwd, bit for bit -- on each patch slice in LDS; w / bias / y / stats / act
 * as in sfod_conv_fwd (3x3, y fp32).  _supported: 1 when the shape is served (dt SFOD_BF16X3, Cin % 32 == 0, the halo-patch
 * kernel's 256 x 128 shape); otherwise run sfod_bn_relu_pool_fwd + sfod_conv_fwd.  It also answers 0 -- process-wide
 * settings, not properties of the shape -- when the generic implicit-GEMM path is forced (sfod_set_conv_algo(1)), when the
 * 16x16x32 form is switched off (sfod_set_conv3x3_m16(0) / SFOD_P3_M16=0: the fold exists on that form only) and when a
 * workgroup-shape variant other than auto / 2 / 5 / 6 is forced (sfod_set_conv3x3_variant).  The caller decides WHETHER to
 * use it (the in-LDS transform costs the convolution 12-15 %; backbone_vgg.py takes it in forward-only train-mode passes
 * for producers whose fp32 output is >= 256 MB: conv2_1 from 3 frames per GPU at 600 x 1200, conv3_1 / conv3_2 from 6;
 * conv4_x never, see profiles/r4_bnin_layers.txt). */
int sfod_conv_fwd_bnin_supported(int B, int H, int W, int Cin, int Cout, int dt);
int sfod_conv_fwd_bnin(const float* x_pre, const float* in_mean, const float* in_invstd, const float* in_gamma,
                       const float* in_beta, const void* w, const float* bias, float* y, int B, int H, int W, int Cin,
                       int Cout, int ldy, int act, float* stats, int dt, void* stream);
/* Tile shape of the generic implicit-GEMM kernel behind sfod_conv_fwd (1x1 / linear / first layer): 0 = the planner's choice
 * (environment SFOD_GEMM_TILE), 1 = 128 x 64, 2 = 128 x 128, 3 = 256 x 128, 4 = 256 x 64, 5 = 256 x 256 (pairs, fp32 out),
 * 6 / 7 / 8 = 256 x 128 / 128 x 128 / 256 x 128 (four stages) with 64-byte K stages (pairs, fp32 out: two or three
 * workgroups per CU); shapes a layer cannot take fall back to the planner's.  Every shape computes the same values.  For
 * A/B runs and parity tests. */
int sfod_set_gemm_tile(int tile);
/* workgroup shape of the halo-patch kernel: 0 auto, 1 = 512 px x 128 ch, 2 = 256 x 128, 3 = 256 x 64,
 * 4 = 512 x 64 (applied where the channel counts allow it), 5 = 256 x 128 on v_mfma_f32_16x16x32 (operand pairs with
 * Cin % 32 == 0; otherwise as 2: 8 waves x (64 px x 64 ch), four waves per SIMD), 6 = the same with 4 waves x (128 px x
 * 64 ch), two per SIMD (leaves registers for a co-resident kernel; tools/experiments/corun_conv_bn.py), 7 / 8 = shapes 3 / 4
 * on v_mfma_f32_16x16x32 (operand pairs with Cin % 32 == 0: 4 waves x (64 px x 64 ch) / 8 waves of them; otherwise as 3 / 4).
 * For A/B runs and parity tests. */
int sfod_set_conv3x3_variant(int variant);
/* Deterministic gradients (default 0; environment SFOD_DETERMINISTIC, config SFOD.DETERMINISTIC): the generic weight-gradient
 * kernels (1x1 / linear / first layer / fp32) store every pixel split's partial tile into a slab of the workspace
 * (sfod_conv_wgrad_ws_bytes grows accordingly) and sum the slabs in split order instead of combining them with float atomics,
 * the bias gradient runs one workgroup per 64 columns: bit-identical results from run to run (the halo-patch 3x3 weight
 * gradient, BatchNorm and ROIAlign backward are always order-fixed).  sfod_get_deterministic: the current setting. */
int sfod_set_deterministic(int on);
int sfod_get_deterministic(void);
/* whether the automatic choice runs shape 2 as shape 5 (default 1; environment SFOD_P3_M16).  Same tiles, same products,
 * another fp32 summation order inside a 32-deep k-step. */
int sfod_set_conv3x3_m16(int on);
/* A/B knob of the bf16x3 halo-patch weight gradient (environment SFOD_W3_PIPE): 2 (default) the 64 co x 64 ci block on
 * 128-pixel tiles (k_wgrad3x3_w64; layers with Cin >= 64, else as 1), 1 the pipelined 64 x 32-block loop
 * (k_wgrad3x3_patch<4, true, true>), 0 the round-2 loop.  0 and 1 give bit-identical results, 2 sums the same products over
 * other pixel tiles (equal to fp32 summation order). */
int sfod_set_wgrad3x3_pipe(int on);
/* which kernel sfod_conv_fwd runs for this shape: 1 generic implicit GEMM, 2 halo-patch, 3 first-layer
 * kernel (Cin = one padded 8-channel chunk, Cout = 64), 4 generic implicit GEMM on its 256 x 256 tile (bf16x3 linear
 * layers / 1x1 convolutions with wide outputs) */
int sfod_conv_fwd_algo(int B, int H, int W, int Cin, int Cout, int ksize, int dt);

/* dt: SFOD_BF16 (x, w, y bf16) or SFOD_BF16X3 (x, w operand pairs; y: the activated operand pairs of the next layer).
 * First VGG layer (3 real channels in one 8-wide chunk -> 64, bf16) with the train-mode BatchNorm + ReLU folded in
 * by recomputation -- the layer's K is 27, its cost is writing 64 channels per pixel: pass 1 (y = NULL) produces
 * only the BatchNorm partial statistics (nothing stored), pass 2 (scale = gamma * invstd, shift = beta - mean *
 * scale, act = 1) recomputes the convolution and stores z = relu(scale * (conv + bias) + shift) directly, so
 * neither y nor a separate BatchNorm pass exist.  For forward-only use (the teacher: vgg.py:15-20 under no_grad;
 * a backward needs y).  stats layout / count as sfod_conv_fwd for this shape (sfod_conv_stats_blocks). */
int sfod_conv_first_supported(int B, int H, int W, int Cin, int Cout, int dt, int ldy);
int sfod_conv_first_fused(const void* x, const void* w, const float* bias, const float* scale,
                          const float* shift, void* y, float* stats, int B, int H, int W, int ldy,
                          int act, int dt, void* stream);
int sfod_conv_first_fused_ws(const void* x, const void* w, const uint32_t* w_absmax, const float* bias,
                             const float* scale, const float* shift, void* y, float* stats, int B, int H, int W,
                             int ldy, int act, int dt, void* stream);

/* weight gradient: dw[n][tap][c] += sum_m dy[m][n] * x[pix(m)+tap][c]  (fp32, packed layout, added
 * into the caller's (zero-initialised) buffer).  3x3 bf16 layers run the halo-patch kernel: pixel
 * splits write fp32 slabs into `ws` (sfod_conv_wgrad_ws_bytes, may be 0 -> ws unused) and are summed
 * in a fixed order; other shapes use the generic kernel (split over row chunks, float atomics). */
int64_t sfod_conv_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int lddy, int dt);
int sfod_conv_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin,
                    int Cout, int ksize, int lddy, int dt, void* ws, int64_t ws_bytes, void* stream);
/* the same gradient written straight in the state-dict layout OIHW [Cout][Cin][3][3] (fp32): overwritten
 * (accumulate 0) or added into (accumulate 1, e.g. the parameter's slice of the flat gradient buffer).
 * Only for shapes the halo-patch kernel serves (sfod_conv_wgrad_oihw_supported != 0): its slab
 * reduction does the layout change, so neither a packed temporary nor its zero fill exist. */
int sfod_conv_wgrad_oihw_supported(int B, int H, int W, int Cin, int Cout, int ksize, int lddy, int dt);
int sfod_conv_wgrad_oihw(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin,
                         int Cout, int ksize, int lddy, int dt, int accumulate, void* ws,
                         int64_t ws_bytes, void* stream);

/* weight (re)packing between the reference's state-dict layouts and the kernel layouts.
 * OIHW fp32 [Cout][Cin][KH][KW] -> packed [Cout][KH*KW][CinPad] (dt); rot180=1 additionally
 * swaps in/out channels and flips the taps: the dgrad weight [Cin][KH*KW][Cout]. */
int sfod_pack_conv_weight(const float* w_oihw, void* w_packed, int Cout, int Cin, int ksize,
                          int CinPad, int rot180, int dt, void* stream);
/* SFOD_F16X3 weights with a per-tensor scale: half pairs keep 22 bits only for |v| >= 2^-3 and weights are uniformly
 * small, so the `_ws` packers first reduce max|w| of the fp32 source into the device word(s) `absmax` (bit pattern of a
 * non-negative float; one word per tensor), then store w * s with s the power of two that puts max|w| * s into
 * [2^13, 2^14).  The forward entry points (`sfod_conv_fwd_ws`, `sfod_conv_first_fused_ws`) take the same word and undo
 * the scale on their accumulators.  absmax == NULL: unscaled (any dt). */
int sfod_pack_conv_weight_ws(const float* w_oihw, void* w_packed, uint32_t* absmax, int Cout, int Cin, int ksize,
                             int CinPad, int rot180, int dt, void* stream);
/* every conv weight of a model in one launch.  desc (device, int64): n entries of 8 values
 * {src fp32 OIHW pointer, dst packed pointer, Cout, Cin, ksize, innerPad, rot180, first_block}, where
 * first_block is the running sum of sfod_pack_conv_weights_blocks over the preceding entries;
 * total_blocks = the sum over all entries.  Same layouts as sfod_pack_conv_weight. */
int sfod_pack_conv_weights_blocks(int Cout, int Cin, int ksize, int innerPad, int rot180);
int sfod_pack_conv_weights_multi(const int64_t* desc, int n, int total_blocks, int dt, void* stream);
int sfod_pack_conv_weights_multi_ws(const int64_t* desc, int n, int total_blocks, int dt, uint32_t* absmax /* [n] */,
                                    void* stream);
/* packed fp32 grad [Cout][taps][CinPad] -> OIHW fp32 grad (accumulate=0: overwrite) */
int sfod_unpack_conv_wgrad(const float* dw_packed, float* dw_oihw, int Cout, int Cin, int ksize,
                           int CinPad, int accumulate, void* stream);
/* nn.Linear weight [N][K] fp32 -> [N][Kperm] (dt) where, if chw_c > 0, the K axis is permuted
 * from (c, p) (flatten of [C,7,7], d2 FastRCNNConvFCHead) to (p, c) (our ROIAlign layout);
 * transpose=1 writes [K][N] instead (the dgrad operand). */
int sfod_pack_fc_weight(const float* w, void* out, int N, int K, int chw_c, int transpose, int dt,
                        void* stream);
int sfod_unpack_fc_wgrad(const float* dw_packed, float* dw, int N, int K, int chw_c,
                         int accumulate, void* stream);
/* same with an explicit leading dimension `ld` of the packed matrix (zero-padded inner axis, so
 * that every row is a whole number of 16-byte chunks, e.g. the 41 predictor outputs -> 48) */
int sfod_pack_fc_weight_ld(const float* w, void* out, int N, int K, int chw_c, int transpose, int ld,
                           int dt, void* stream);
int sfod_pack_fc_weight_ld_ws(const float* w, void* out, uint32_t* absmax, int N, int K, int chw_c, int transpose,
                              int ld, int dt, void* stream);
int sfod_unpack_fc_wgrad_ld(const float* dw_packed, float* dw, int N, int K, int chw_c, int ld,
                            int accumulate, void* stream);
/* column sums of a [M, ld] matrix's first N columns: bias gradients.  db[n] (+)= sum_m dy[m][n] */
int sfod_bias_grad(const void* dy, float* db, int M, int N, int ld, int accumulate, int dt,
                   void* stream);

/* ---- K3/K4: train-mode BatchNorm2d + ReLU (+ 2x2 max-pool), NHWC.  vgg.py:15,20; torch
 * BatchNorm2d semantics of SURVEY.md A.14 (biased var to normalise, unbiased var into
 * running_var, momentum 0.1, eps 1e-5).  This is also the AdaBN running-stat refresh
 * (daod/engine/trainers/base.py:270-337): it runs identically under no_grad. */
int sfod_bn_finalize(const float* stats, int nblocks, int M, int C,
                     float* mean, float* invstd, float* running_mean, float* running_var,
                     float momentum, float eps, int update_running, int64_t* num_batches_tracked,
                     float* ws, void* stream);
/* update_running: 0 = leave the running statistics alone; k >= 1 = apply k momentum updates with this batch's
 * statistics (k forward passes over the same batch between two optimiser steps -- the reference's student runs its
 * backbone three times per step when its zero-weighted domain branch is on, source_free_adaptive_teacher.py:527-537).
 * num_batches_tracked: the BatchNorm buffer of that name (device int64 scalar), incremented by k; may be NULL. */
/* Layers whose statistics fit one slice (nblocks <= 64) are finalised by ONE launch instead of two (default 1; environment
 * SFOD_BN_FINALIZE_FUSED): the same sums in the same order, bit-identical outputs.  0 keeps the two launches (A/B runs, the
 * parity test). */
int sfod_set_bn_finalize_fused(int on);
/* size (in floats, 8-byte aligned) of the `ws` scratch of sfod_bn_finalize */
int sfod_bn_finalize_ws_floats(int C);
/* z = relu(gamma*(y-mean)*invstd+beta); `pool` is a flag word: bit 0 additionally 2x2/2 max-pools z
 * (floor), bit 1 drops the ReLU (d2 BottleneckBlock conv3 / shortcut norms, no activation) */
int sfod_bn_relu_pool_fwd(const void* y, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, void* z, int B, int H, int W,
                          int C, int pool, int dt, int out_dt, void* stream);
/* the same with a second output: out_dt SFOD_F16X3 (z: half pairs, the next convolution's forward operand) and z2 (may be
 * NULL) the same values as SFOD_BF16X3 pairs (that convolution's weight-gradient operand) -- "f16x3" mode, student pass */
int sfod_bn_relu_pool_fwd2(const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, void* z, void* z2, int B, int H, int W,
                           int C, int pool, int dt, int out_dt, void* stream);
/* Bottleneck tail of the ResNet path (d2 BottleneckBlock.forward, reached through build_resnet_backbone of the r101
 * yaml): z = relu(bn(y) + residual) in one pass over [rows, C]; same statistics / affine inputs as above.
 * z_pairs (may be NULL; dt SFOD_F32, C % 8 == 0): additionally the same values as operand pairs of type pairs_dt
 * (SFOD_BF16X3 / SFOD_F16X3) -- the block output is both the fp32 residual stream and the next convolution's MFMA
 * operand; z_pairs2 (may be NULL; pairs_dt SFOD_F16X3): once more as SFOD_BF16X3 pairs -- that convolution's
 * weight-gradient operand in "f16x3" mode. */
int sfod_bn_add_relu_fwd(const void* y, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const void* residual, void* z, void* z_pairs, void* z_pairs2, int64_t rows,
                         int C, int dt, int pairs_dt, void* stream);
/* backward of the block above.  dz: grad w.r.t. block output; y: saved conv output; returns dy
 * (grad w.r.t. conv output), dgamma, dbeta.  ws: fp32 workspace [nblk*2*C] (see ws query).
 * dgamma_acc / dbeta_acc (may be NULL): the parameters' gradient accumulators (+= this call's dgamma /
 * dbeta), so the caller needs no separate accumulate pass per BatchNorm layer. */
int sfod_bn_relu_pool_bwd(const void* dz, const void* y, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, void* dy, float* dgamma,
                          float* dbeta, float* dgamma_acc, float* dbeta_acc, float* ws, int B, int H,
                          int W, int C, int pool, int dt, int out_dt, int reduced_blocks, void* stream);
/* reduced_blocks > 0: `ws` already holds that many rows of partial (dbeta | dgamma) sums, written by the epilogue of the
 * data-gradient convolution that produced dz (sfod_conv_dgrad_bnred): the reduction pass over dz and y is skipped.
 * `ws` must then have SFOD_BN_BWD_SCRATCH_ROWS more rows of 2 * C floats behind them (scratch of the two-stage sum). */
#define SFOD_BN_BWD_SCRATCH_ROWS 32
int sfod_bn_bwd_ws_floats(int M, int C);
/* ---- ResNet-101-C4 backbone helpers (d2 build_resnet_backbone selected by the r101 yaml's missing
 * BACKBONE.NAME, configs/r101_c4_cs_foggy_adaptive_teacher_source_free.yaml:1-28) ----
 * out = act(a + b), act 0/1: residual join of BottleneckBlock (relu(conv3(x) + shortcut(x))); out_pairs (may be NULL,
 * fp32 data, n % 8 == 0 with 8-channel groups): the same values as operand pairs of type pairs_dt; out_pairs2 (may be NULL;
 * pairs_dt SFOD_F16X3): once more as SFOD_BF16X3 pairs */
int sfod_add_act(const void* a, const void* b, void* out, void* out_pairs, void* out_pairs2, int64_t n, int act, int dt,
                 int pairs_dt, void* stream);
/* backward=0: dst [B,ceil(H/2),ceil(W/2),C] = src [B,H,W,C] at even pixels (data movement of a 1x1
 * stride-2 conv, STRIDE_IN_1X1); backward=1: the adjoint (dst [B,H,W,C] zero except even pixels) */
int sfod_subsample2(const void* src, void* dst, int B, int H, int W, int C, int backward, int dt,
                    void* stream);
/* ResNet BasicStem as one kernel on operand pairs (d2 build_resnet_backbone: conv 7x7 stride 2 pad 3, 3 -> 64, FrozenBN
 * folded into w / bias, ReLU): replaces sfod_im2col_stem + sfod_conv_fwd for dt SFOD_BF16X3 / SFOD_F16X3 -- the [pixels][160]
 * operand matrix is gathered from an LDS patch of the input in registers and never written (bit-identical output: same k
 * order, same splits, same MFMA sequence).  x fp32 [B,H,W,Cp] (channels 0-2 used, Cp % 4 == 0); w_packed: sfod_pack_fc_weight
 * of the [64][160] matrix (k = (ky*7+kx)*3+c, zero columns from 147), w_absmax its SFOD_F16X3 scale word or NULL;
 * y fp32 [B,Ho,Wo,64], Ho = (H-1)/2+1.  act 0 / 1 (ReLU).  The max-pool that follows stays sfod_maxpool3s2. */
int sfod_stem7x7_supported(int B, int H, int W, int Cp, int dt);
int sfod_stem7x7(const float* x, const void* w_packed, const uint32_t* w_absmax, const float* bias, float* y, int B, int H,
                 int W, int Cp, int act, int dt, void* stream);
/* im2col of BasicStem's 7x7 stride-2 pad-3 conv: out [B,Ho,Wo,Kpad], column (ky*7+kx)*3+c; out_dt = dt, or
 * SFOD_BF16X3 from fp32 input (the stem GEMM's operand pairs, written directly) */
int sfod_im2col_stem(const void* x, void* out, int B, int H, int W, int Cp, int Kpad, int dt, int out_dt,
                     void* stream);
/* BasicStem max_pool2d(kernel 3, stride 2, padding 1), forward (the stem is frozen) */
int sfod_maxpool3s2(const void* x, void* y, int B, int H, int W, int C, int dt, void* stream);
/* elementwise: dx = dy * (y > 0) (ReLU) or dy * (y > 0 ? 1 : 0.2) (LeakyReLU) in place on dy */
int sfod_act_bwd(void* dy, const void* y, int64_t n, int act, int dt, void* stream);
/* a += b  (fp32 or bf16 elementwise) */
int sfod_add_inplace(void* a, const void* b, int64_t n, int dt, void* stream);
/* a = (a + b) * [y > 0]: sum of a residual join's two gradient branches taken through the ReLU of the block below
 * (its output y) in one pass -- sfod_add_inplace followed by sfod_act_bwd(act 1) */
int sfod_add_act_bwd(void* a, const void* b, const void* y, int64_t n, int dt, void* stream);
/* a = mask ? a * scale : 0 (mask: one 0/1 byte per element): F.dropout forward and backward of the
 * instance-level domain discriminator (daod/modeling/dann/dann.py:146-151), scale = 1 / (1 - p) */
int sfod_mul_mask(void* a, const uint8_t* mask, int64_t n, float scale, int dt, void* stream);

/* ---- K6/K7: anchors + proposal decode.  d2 DefaultAnchorGenerator / Box2BoxTransform /
 * find_top_rpn_proposals reached from daod/modeling/proposal_generator/rpn.py:25,54-56.
 * rpn_out: fp32 [B*Hf*Wf, ld]: cols [0,A) objectness, cols [A, 5A) deltas (a*4+j).
 * props: [B,NA,4] decoded+clipped boxes (order y,x,a); scores: [B,NA]; flags[0] |= 1 if any
 * non-finite prediction (d2 raises FloatingPointError). */
int sfod_rpn_decode(const float* rpn_out, int ld, const float* cell_anchors, int A, int B, int Hf,
                    int Wf, int stride, const int32_t* image_sizes, float* props, float* scores,
                    int32_t* flags, void* stream);
/* stable descending segmented sort of B segments of n floats; out_idx int32 [B,n].
 * ws: workspace of sfod_sort_ws_bytes(B,n) bytes. */
int64_t sfod_sort_ws_bytes(int B, int n);
int sfod_segmented_sort_desc(const float* keys, int B, int n, float* out_keys, int32_t* out_idx,
                             void* ws, int64_t ws_bytes, void* stream);
/* gather the first k sorted candidates: boxes[b][j] = props[b][idx[b][j]], valid = nonempty */
int sfod_rpn_gather_topk(const float* props, const float* sorted_scores, const int32_t* sorted_idx,
                         int B, int NA, int k, float* cand_boxes, float* cand_scores,
                         uint8_t* cand_valid, void* stream);

/* ---- K8/K16: greedy NMS.  tv nms / batched_nms via d2 batched_nms (Appendix A.6):
 * boxes already sorted by descending score; suppress j>i when IoU > thr (strict, fp32, no +1).
 * classes (optional int32): pairs of different class never suppress each other ("vanilla"
 * batched_nms); alt_boxes/mode (optional): per image mode[b]!=0 selects alt_boxes (the
 * coordinate-offset boxes) and disables the class test -- torchvision's strategy switch.
 * n_per_image (optional int32 [B]) bounds the live prefix, else n.  mask: u64 [B,n,ceil(n/64)].
 * keep_idx int32 [B,max_keep] (positions in the sorted order), keep_count int32 [B]. */
int64_t sfod_nms_mask_bytes(int B, int n);
int sfod_nms(const float* boxes, const float* alt_boxes, const int32_t* classes,
             const int32_t* mode, const uint8_t* valid, const int32_t* n_per_image, int B, int n,
             float thr, int max_keep, uint64_t* mask, int32_t* keep_idx, int32_t* keep_count,
             void* stream);
/* proposals[b][j] = cand_boxes[b][keep_idx[b][j]] for j < keep_count[b], zero-filled above */
int sfod_gather_kept(const float* cand_boxes, const float* cand_scores, const int32_t* keep_idx,
                     const int32_t* keep_count, int B, int n, int max_keep, float* out_boxes,
                     float* out_scores, void* stream);

/* ---- K9/K12: pairwise IoU + Matcher.  d2 pairwise_iou + Matcher (Appendix A.8) as reached
 * from rpn.py:45 (anchors, thresholds [0.3,0.7], labels [0,-1,1], low-quality on) and
 * source_free_adaptive_teacher_roi_heads.py:179-183 (proposals, [0.5], [0,1]).
 * anchors are generated in-kernel from (Hf,Wf,stride,cell_anchors).  gt: [B,Gcap,4] + count.
 * out: matched int32 [B,NA], labels int8 [B,NA]; gtmax fp32 [B,Gcap] scratch. */
int sfod_anchor_match(const float* cell_anchors, int A, int B, int Hf, int Wf, int stride,
                      const float* gt_boxes, const int32_t* gt_count, int Gcap, float lo, float hi,
                      int32_t* matched, int8_t* labels, float* gtmax, void* stream);
/* ROI version: boxes [B,P,4] with count; one threshold; writes the class target:
 * cls[b][p] = gt_classes[matched] if IoU>=thr else num_classes; -2 for p >= count. */
int sfod_roi_match(const float* boxes, const int32_t* box_count, int B, int P,
                   const float* gt_boxes, const int32_t* gt_classes, const int32_t* gt_count,
                   int Gcap, float thr, int num_classes, int32_t* matched, int32_t* cls,
                   void* stream);
/* ---- K10: subsample_labels (Appendix A.9).  The random choice is an INPUT: one uint32 key per
 * element; the sample is the n candidates with the smallest (key, index).
 * mode 0 (RPN, _subsample_labels): labels int8 in/out [B,n]: positives = 1, negatives = 0,
 *   everything not sampled becomes -1.
 * mode 1 (ROI, _sample_proposals): cls int32 [B,n] in; out_idx int32 [B,num] = sampled fg indices
 *   ascending then bg ascending, out_count [B]. */
int sfod_subsample(void* labels_or_cls, const uint32_t* keys, int B, int n, int num, float pos_frac,
                   int bg_label, int mode, int32_t* out_idx, int32_t* out_count, void* stream);

/* ---- K11: RPN losses (Appendix A.10; rpn.py:46-49).  loss[0]=loss_rpn_cls, loss[1]=
 * loss_rpn_loc, both / (batch_per_image*B).  With grad_scale != NULL (device fp32 [2], the
 * upstream d(total)/d(loss_k)) also writes d_rpn_out (same layout as rpn_out, dense). */
int sfod_rpn_loss(const float* rpn_out, int ld, const float* cell_anchors, int A, int B, int Hf,
                  int Wf, int stride, const int8_t* labels, const int32_t* matched,
                  const float* gt_boxes, const int32_t* gt_count, int Gcap, int batch_per_image,
                  float* loss, const float* grad_scale, float* d_rpn_out, float* ws, void* stream);

/* concat proposals and GT boxes: d2 add_ground_truth_to_proposals (roi_heads.py:170) */
int sfod_append_gt(const float* props, const int32_t* prop_count, int B, int P,
                   const float* gt_boxes, const int32_t* gt_count, int Gcap, float* out_boxes,
                   int32_t* out_count, void* stream);
/* build the sampled ROI batch: rois [B*S,5], gt_cls int32 [B*S] (-1 on padding rows),
 * gt_box [B*S,4]; n_valid int32[1] = number of real rows in the whole batch. */
int sfod_roi_build_samples(const float* boxes, const int32_t* cls, const int32_t* matched,
                           const int32_t* samp_idx, const int32_t* samp_count, int B, int P, int S,
                           const float* gt_boxes, const int32_t* gt_count, int Gcap, float* rois,
                           int32_t* gt_cls, float* gt_box, int32_t* n_valid, void* stream);
/* rois [B*P,5] from per-image proposal arrays (padding rows get batch index -1) */
int sfod_make_rois(const float* props, const int32_t* prop_count, int B, int P, float* rois,
                   void* stream);

/* ---- K13: ROIAlign (tv roi_align aligned=True, adaptive sampling; Appendix A.11) on NHWC
 * features.  out: [R, PH*PW, C] (dt).  rois with batch index < 0 produce zeros.  pooled in [1, 16] (the backward: [1, 8]). */
int sfod_roi_align_fwd(const void* feat, int B, int H, int W, int C, const float* rois, int R,
                       int pooled, float scale, void* out, int dt, void* stream);
/* dfeat fp32 [B,H,W,C] += adjoint of the forward (zero-init by the caller); one workgroup per ROI,
 * separable interpolation weights, one float atomic per footprint pixel and channel */
int sfod_roi_align_bwd(const void* dout, int B, int H, int W, int C, const float* rois, int R,
                       int pooled, float scale, float* dfeat, int dt, void* stream);

/* ---- K15: Fast R-CNN losses (Appendix A.12; roi_heads.py:124).  pred: fp32 [R, ld]: cols
 * [0,K] class scores, cols [K+1, K+1+4K) deltas.  loss[0]=loss_cls, loss[1]=loss_box_reg. */
int sfod_frcnn_loss(const float* pred, int ld, int R, int K, const float* rois,
                    const int32_t* gt_cls, const float* gt_box, const int32_t* n_valid,
                    float* loss, const float* grad_scale, float* d_pred, float* ws, void* stream);

/* ---- K16: teacher inference post-processing (Appendix A.13; roi_heads.py:161) ----
 * step 1: softmax + per-class decode + clip + score filter -> candidates [B, P*K]
 *         (cand_scores = -1 for filtered entries), cand_count[b] = number passing. */
int sfod_frcnn_candidates(const float* pred, int ld, int B, int P, int K, const float* props,
                          const int32_t* prop_count, const int32_t* image_sizes,
                          float score_thresh, float* cand_boxes, float* cand_scores,
                          int32_t* cand_count, void* stream);
/* d2 FastRCNNOutputLayers.predict_probs (reached from daod/modeling/roi_heads/source_free_fast_rcnn.py:16-17): row
 * softmax of scores [R, ld] (K+1 class scores per row) -> probs [R, K+1], in step 1's operation order. */
int sfod_predict_probs(const float* scores, int ld, int R, int K, float* probs, void* stream);
/* step 2 (after sorting scores): gather sorted candidates, classes, coordinate-offset boxes
 * and the torchvision strategy flag (mode[b]=1: coordinate trick, 0: per-class). */
int sfod_frcnn_prepare_nms(const float* cand_boxes, const float* sorted_scores,
                           const int32_t* sorted_idx, const int32_t* cand_count, int B, int n,
                           int K, int numel_limit, float* s_boxes, float* s_alt_boxes,
                           int32_t* s_classes, int32_t* mode, float* maxcoord_ws, void* stream);
/* step 3 (after NMS): top-`max_det` detections + the pseudo-label filter score > thr
 * (daod/engine/trainers/source_free_adaptive_teacher.py:167-181, strict '>'). */
int sfod_frcnn_finalize(const float* s_boxes, const float* sorted_scores, const int32_t* s_classes,
                        const int32_t* keep_idx, const int32_t* keep_count, int B, int n,
                        int max_det, float pseudo_thr, float* det_boxes, float* det_scores,
                        int32_t* det_classes, int32_t* det_count, float* gt_boxes,
                        int32_t* gt_classes, int32_t* gt_count, void* stream);

/* BPC calibration metric of the student's training pass (SURVEY 8a row a6 + 8f rank 4), logged as
 * calibration/bpc_loss and weighted by 0 (source_free_adaptive_teacher.py:549-550,566).  Fuses
 * SourceFreeFastRCNNOutputLayers.convert_bbox_scores (daod/modeling/roi_heads/source_free_fast_rcnn.py:15-36,
 * 82-147: softmax, per-class decode, non-finite rows dropped, clip, score > 0, no NMS) applied -- as the
 * reference does -- to the sampled proposals AFTER their boxes were overwritten by
 * predict_boxes_for_gt_classes (source_free_adaptive_teacher_roi_heads.py:136-143), with bpc_loss
 * (daod/loss/bpc_loss.py:10-262) against the images' (pseudo) ground truth.
 * pred [R,ld] fp32 = (K+1 logits | 4K deltas) per sampled row; rois [R,5] (image, x1,y1,x2,y2; image < 0:
 * padding row); roi_cls [R] the rows' gt classes (K = background); image_sizes [B,2] (h,w);
 * gt_boxes [B,G,4], gt_classes [B,G], gt_count [B].  ws: B*4 doubles.  loss: 1 float. */
int sfod_bpc_loss(const float* pred, int ld, int R, int K, const float* rois, const int32_t* roi_cls,
                  int B, const int32_t* image_sizes, const float* gt_boxes, const int32_t* gt_classes,
                  const int32_t* gt_count, int G, float iou_thresh, float* loss, void* ws, void* stream);

/* Class-wise adaptive pseudo-label threshold: AdaptiveConfidenceBasedSelfTrainingLoss
 * (daod/modeling/adaptive_thresh/adaptive_confidence.py:6-34, "convex" curve) with the trainer's bookkeeping
 * (daod/engine/trainers/source_free_adaptive_teacher.py:282-295 count_label_prediction, :297-309
 * update_adaptive_threshold, :393-404 per-step update, :461-466 selection).  Over the det_* arrays of
 * sfod_frcnn_finalize [B,max_det]: writes the per-class counts of score > thr into reserve[row] ([R][K] fp32
 * ring, caller-owned state), class_acc[K] = counter / max(max(counter), 1) with classes 0 and 2 excluded from
 * the counter and pinned to 1, and, when `select` != 0, replaces the pseudo ground truth of every image by the
 * stable subset score >= thr * (acc[c] / (2 - acc[c])) (gt_boxes/gt_classes/gt_scores [B,max_det], gt_count [B]).
 * K <= 64. */
int sfod_adaptive_pseudo_labels(const float* det_boxes, const float* det_scores,
                                const int32_t* det_classes, const int32_t* det_count, int B,
                                int max_det, int K, float thr, float* reserve, int R, int row,
                                float* class_acc, int select, float* gt_boxes, int32_t* gt_classes,
                                float* gt_scores, int32_t* gt_count, void* stream);

/* ---- Strong augmentation on the device: build_strong_augmentation
 * (daod/data/detection_utils.py:7-36) as applied by DatasetMapperTwoCropSeparate
 * (daod/data/mappers/two_crop_augmentation_mapper.py:141-146), on planar uint8 [3][H][W] frames; bit-exact
 * with the torchvision-on-Pillow path (channels in stored order, as the reference hands its BGR array to
 * Pillow as "RGB").
 * sfod_aug_color: a sequence of <= 8 point operations: 0 brightness, 1 contrast (at most one per call), 2
 * saturation (Pillow ImageEnhance = Image.blend with black / the mean-luma level / the luma image; factor =
 * blend alpha), 3 hue (Pillow HSV round trip; the factor slot carries the wrapped uint8 shift
 * np.uint8(hue_factor * 255)), 4 grayscale (convert("L") replicated; factor ignored).  codes / factors are HOST
 * arrays.  ws: 8 bytes of device memory (the contrast op's luma sum).  in may equal out.
 * sfod_aug_gaussian_blur: ImageFilter.GaussianBlur(radius=sigma) (daod/data/transforms/augmentations.py:6-21):
 * Pillow's three extended-box passes along x then along y; in, out, tmp distinct, C*H*W bytes each.
 * sfod_aug_erase: RandomErasing(value="random") followed by ToPILImage: rectangle (i, j, h, w) of every channel
 * is overwritten by (uint8)(noise * 255) (truncation, low 8 bits); noise [C][h][w] fp32 on the device. */
int sfod_aug_color(const uint8_t* in, uint8_t* out, int H, int W, int n_ops, const int32_t* codes,
                   const float* factors, void* ws, void* stream);
int sfod_aug_gaussian_blur(const uint8_t* in, uint8_t* out, uint8_t* tmp, int C, int H, int W, float sigma,
                           void* stream);
int sfod_aug_erase(uint8_t* img, int C, int H, int W, int i, int j, int h, int w, const float* noise,
                   void* stream);

/* ---- K20/K21: fused SGD(momentum, weight decay) + EMA teacher update over flat fp32 arrays.
 * d2 build_optimizer + torch SGD (Appendix A.15) and _update_teacher_model
 * (source_free_adaptive_teacher.py:583-603).  lr is a device scalar (no host sync on schedule).
 * first_step: momentum buffer is initialised to the gradient (torch SGD semantics).
 * teacher may be NULL (no EMA).
 * ema_one_minus_keep: the caller's (1 - k) computed in double and rounded once (the reference multiplies by the python
 * float 1 - keep_rate); the kernels use two rounded products and a rounded sum like torch's op chain, so the teacher
 * is bit-identical to _update_teacher_model's (tests/golden/glue_ref.npz). */
int sfod_sgd_ema(float* param, const float* grad, float* mom, float* teacher, int64_t n,
                 const float* lr, float momentum, float weight_decay, float grad_scale,
                 float ema_keep, float ema_one_minus_keep, int first_step, void* stream);
/* t = s*(1-k) + t*k  on fp32 buffers (BN running stats) */
int sfod_ema(float* teacher, const float* student, int64_t n, float keep, float one_minus_keep, void* stream);
/* the same update on the int64 buffers (num_batches_tracked): fp32 arithmetic, truncated on the way back like the
 * reference's load_state_dict copy (source_free_adaptive_teacher.py:583-603, SURVEY A.17 iv).  Both factors are passed
 * as the caller rounded them (torch rounds the Python doubles k and 1 - k to fp32 separately). */
int sfod_ema_i64(int64_t* teacher, const int64_t* student, int n, float keep, float one_minus_keep, void* stream);
/* the scalars the trainer logs after the teacher pass (source_free_adaptive_teacher.py:411-423,445-452): out[0] mean over
 * images of the mean detection score, out[1] RPN proposals with logit > thr per image, out[2] mean pseudo-label count.
 * det_scores [B,D] / det_count [B], rpn_logits [B,P] / rpn_count [B], gt_count [B]; out: 3 floats. */
int sfod_teacher_metrics(const float* det_scores, const int* det_count, int D, const float* rpn_logits,
                         const int* rpn_count, int P, const int* gt_count, int B, float thr, float* out,
                         void* stream);

/* fp32 [n] (n % 8 == 0) -> SFOD_F16X3 pairs and SFOD_BF16X3 pairs of the same values in one pass ("f16x3" mode: a tensor
 * that is the operand of a forward product and of a weight gradient; replaces the reference's single fp32 tensor at
 * every such site, e.g. the RPN head input daod/modeling/proposal_generator/rpn.py:25-56) */
int sfod_cast_pairs_both(const float* src, void* dst_f16x3, void* dst_bf16x3, int64_t n, void* stream);
/* SFOD_F16X3 range report: every producer of half pairs raises a device flag when a value beyond +-65504 (an infinity included; NaN stays NaN) had to be
 * clamped (activations or scaled weights outside the window the mode assumes).  This call ORs the flags into the caller's
 * device word (zeroed by the caller once) and clears them; no host synchronisation -- the trainer reads the word at its
 * metrics period, beside the RPN's non-finite flag (d2 raises FloatingPointError inside predict_proposals). */
int sfod_f16x3_poll(uint32_t* word, void* stream);
/* utilities */
int sfod_fill_f32(float* p, int64_t n, float v, void* stream);
int sfod_cast(const void* src, void* dst, int64_t n, int src_dt, int dst_dt, void* stream);

#ifdef __cplusplus
}
#endif
#endif
