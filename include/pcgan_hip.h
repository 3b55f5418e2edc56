/*
 * pcgan_hip.h -- C-ABI of libpcgan_hip.so: the MI355X (gfx950) kernels behind the
 * PC-GAN wsgan_emb training step.
 *
 * The reference (phymhan pc-gan) has no FFI: every op on the hot path is a stock
 * torch.nn module.  Each entry point below therefore names the
 * torch.nn call site in the reference that it replaces (file:line relative to the
 * reference root).  The host side (pc-gan_amd/hip/*.py) binds these with ctypes and
 * wraps them in torch.autograd.Function objects used by the define_G/define_D/
 * define_E/define_IP modules.
 *
 * Conventions
 *   - tensors are NCHW, contiguous, device pointers (tensor.data_ptr()).  ACTIVATION tensors (and their gradients) are
 *     `void*` of the storage type named by the call's `dtype` argument / the descriptor's `dtype` field (PCGAN_F32 or
 *     PCGAN_BF16); everything typed `float*` -- parameters, parameter gradients, biases, statistics, losses, optimizer
 *     state -- is fp32 in either mode, and all arithmetic accumulates in fp32.
 *   - no hidden allocation, no device synchronisation, NO ENVIRONMENT VARIABLES (round 4: the 14 getenv reads of round 3
 *     are gone): the caller provides the workspace (query *_workspace_bytes first) and the hipStream_t
 *     (torch.cuda.current_stream().cuda_stream); every launch is asynchronous on that stream, so calls are
 *     graph-capturable and may run concurrently on distinct streams.  The library's ENTIRE process-global state is the three
 *     items declared at the end of this header, each set only by an explicit call: the routing options (pcgan_set_option:
 *     five integers with measured-best defaults), the non-finite sentinel pointer (pcgan_set_nonfinite_counter) and the
 *     measurement-only kernel timer (pcgan_timer_enable).  Nothing else is mutable between calls.
 *   - return value 0 = ok; non-zero = error, text via pcgan_last_error()
 *     (thread-local).  The Python side turns non-zero into RuntimeError.
 */
#ifndef PCGAN_HIP_H
#define PCGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pcgan_stream_t; /* hipStream_t */

/* ---- status ------------------------------------------------------------------ */
const char* pcgan_last_error(void);
int pcgan_version(void);
/* number of compute units / arch string of device 0 (diagnostics) */
int pcgan_device_info(int* cu_count, char* arch, int arch_len);

/* ---- convolution family ------------------------------------------------------
 * One descriptor covers nn.Conv2d and (with the roles swapped by the caller)
 * nn.ConvTranspose2d.  Replaces:
 *   ResnetGenerator convs      models/networks.py:578-605
 *   ResnetBlock convs          models/networks.py:621-648 (ReflectionPad2d folded in)
 *   NLayerDiscriminator convs  models/networks.py:747-772
 *   ResNet-18 / Elo head convs models/resnet.py:20-28,134 ; models/networks.py:1020-1026
 *   AlexNetFeature convs       models/networks.py:1223-1233
 * pad_mode: 0 = zero padding, 1 = reflection padding (ReflectionPad2d(pad) + conv pad 0).
 * stride must be 1, 2 or 4.
 */
typedef struct {
    int N, C, H, W; /* input  [N][C][H][W]            */
    int K, R, S;    /* weight [K][C][R][S]            */
    int stride, pad, pad_mode;
    int P, Q;       /* output [N][K][P][Q]            */
    int dtype;      /* storage type of x / y / dy / dx: PCGAN_F32 or PCGAN_BF16 (w, bias, dw are fp32)   */
} pcgan_conv_desc;

enum { PCGAN_ACT_NONE = 0, PCGAN_ACT_RELU = 1, PCGAN_ACT_LRELU = 2, PCGAN_ACT_TANH = 3, PCGAN_ACT_SIGMOID = 4 };
enum { PCGAN_PASS_FWD = 0, PCGAN_PASS_BWD_DATA = 1, PCGAN_PASS_BWD_WEIGHT = 2 };
/* storage type of ACTIVATION tensors (and of their gradients); parameters, parameter gradients, statistics, losses and
 * optimizer state are always fp32, accumulation is always fp32 */
enum { PCGAN_F32 = 0, PCGAN_BF16 = 1 };

size_t pcgan_conv2d_workspace_bytes(const pcgan_conv_desc* d, int pass);

/* y = act(conv(x, w) + bias); bias may be NULL. */
int pcgan_conv2d_fwd(const pcgan_conv_desc* d, const void* x, const float* w, const float* bias,
                     void* y, int act, float slope, void* ws, size_t ws_bytes, pcgan_stream_t s);
/* dx = conv^T(dy, w) (+ bias per dx-channel, used when this entry serves as the
 * forward of nn.ConvTranspose2d, models/networks.py:597-600). */
int pcgan_conv2d_bwd_data(const pcgan_conv_desc* d, const void* dy, const float* w, const float* bias,
                          void* dx, void* ws, size_t ws_bytes, pcgan_stream_t s);
/* Weight packing split out of the two calls above.  fwd/bwd_data re-tile w into the implicit-GEMM A
 * operand on every call (a few us, ~2 % of a step); weights only change once per optimizer step while the
 * reference's step runs each net 2-4 times (models/wsgan_emb_model.py:277-330), so the host can pack once
 * per (weight version, pass) and call the *_packed variants.  `packed` holds pcgan_conv2d_packed_bytes(d,
 * pass) bytes (0 for PCGAN_PASS_BWD_WEIGHT) and is valid for any desc that differs only in N (batch). */
size_t pcgan_conv2d_packed_bytes(const pcgan_conv_desc* d, int pass);
int pcgan_conv2d_pack_weights(const pcgan_conv_desc* d, int pass, const float* w, float* packed,
                              pcgan_stream_t s);
int pcgan_conv2d_fwd_packed(const pcgan_conv_desc* d, const void* x, const float* packed, const float* bias,
                            void* y, int act, float slope, void* ws, size_t ws_bytes, pcgan_stream_t s);
int pcgan_conv2d_bwd_data_packed(const pcgan_conv_desc* d, const void* dy, const float* packed,
                                 const float* bias, void* dx, void* ws, size_t ws_bytes, pcgan_stream_t s);
/* dw[K][C][R][S] (+)= sum_{n,p,q} dy * gather(x).  accumulate != 0 adds into dw -- used with dw = the
 * parameter's slice of the optimizer's flat gradient buffer, which fuses autograd's "grad += dw" pass. */
int pcgan_conv2d_bwd_weight(const pcgan_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                            void* ws, size_t ws_bytes, pcgan_stream_t s);

/* ---- per-channel reductions / pointwise ---------------------------------------- */
/* out[c] (+)= sum over n,h,w of x[n][c][h][w]  (bias gradients); scratch_nc: N*C floats. */
int pcgan_channel_sum(const void* x, float* out, float* scratch_nc, int N, int C, int HW, int accumulate, int dtype,
                      pcgan_stream_t s);
/* dx = dy * act'(y)  where y is the activation OUTPUT (relu/lrelu/tanh/sigmoid). */
int pcgan_act_bwd(const void* dy, const void* y, void* dx, size_t n, int act, float slope, int dtype, pcgan_stream_t s);
/* y = act(x) standalone (nn.ReLU in AlexNetFeature, models/networks.py:1224-1234). */
int pcgan_act_fwd(const void* x, void* y, size_t n, int act, float slope, int dtype, pcgan_stream_t s);
/* out[n][0..C)[hw] = img, out[n][C..C+nz)[hw] = z[n or 0][j] broadcast
 * (torch.cat((input, z_img), 1), models/networks.py:610-611, 781-782).  z_batch is 1 or N. */
int pcgan_concat_z(const void* img, const float* z, void* out, int N, int C, int nz, int HW, int z_batch, int dtype,
                   pcgan_stream_t s);
/* y = a + b; y = alpha * scalar_dev[0] * x (scalar_dev may be NULL): the upstream
 * gradient of a scalar loss stays on the device -- helpers for the autograd glue. */
int pcgan_add(const void* a, const void* b, void* y, size_t n, int dtype, pcgan_stream_t s);
int pcgan_scale(const void* x, const float* scalar_dev, float alpha, void* y, size_t n, int dtype, pcgan_stream_t s);
/* y = x converted between storage types (fp32 -> bf16: round to nearest even; the bf16 path's boundary: images in,
 * parity checks out). */
int pcgan_cast(const void* x, int dtype_x, void* y, int dtype_y, size_t n, pcgan_stream_t s);
/* Dropout2d with an explicit keep mask per (n,c) (models/resnet.py:38-51): y = x*mask[nc]*scale. */
int pcgan_channel_scale(const void* x, const float* mask_nc, void* y, int NC, int HW, float scale, int dtype,
                        pcgan_stream_t s);

/* ---- normalisation (+ fused activation / residual) -----------------------------
 * InstanceNorm2d(affine=False, track_running_stats=True)  models/networks.py:26
 * BatchNorm2d(affine=True) in train mode                   models/networks.py:24,
 *                                                          models/resnet.py:47-51,136
 * plane_stats: per (n,c) plane mean and M2 = sum (x-mean)^2 (exact two-pass). */
int pcgan_plane_stats(const void* x, float* mean_nc, float* m2_nc, int NC, int HW, int dtype, pcgan_stream_t s);
/* merge the N per-plane (mean,M2) of each channel (Chan's formula) into batch
 * statistics: mean_c, var_c (biased) and update running stats
 * (momentum m: r = (1-m) r + m stat, unbiased variance), either pointer may be NULL. */
int pcgan_bn_merge(const float* mean_nc, const float* m2_nc, float* mean_c, float* var_c,
                   float* running_mean, float* running_var, int N, int C, int HW, float momentum,
                   pcgan_stream_t s);
/* instance-norm running-stat update: r_mean = (1-m) r_mean + m * mean_n(mean_nc),
 * r_var = (1-m) r_var + m * mean_n(unbiased var_nc). */
int pcgan_in_running_update(const float* mean_nc, const float* m2_nc, float* running_mean,
                            float* running_var, int N, int C, int HW, float momentum, pcgan_stream_t s);
/* y = act( (x - mean[i]) * rsqrt(var[i] + eps) * gamma[c] + beta[c] + residual )
 * with i = n*C+c (per_plane=1, instance norm: `var` then holds the plane M2 from
 * pcgan_plane_stats and the kernel divides by HW) or i = c (per_plane=0, batch norm or
 * eval-mode running statistics: `var` is the variance).  gamma/beta/residual may be NULL. */
int pcgan_norm_act_fwd(const void* x, const float* mean, const float* var, const float* gamma,
                       const float* beta, const void* residual, void* y, float* y_pmax, int N, int C, int HW,
                       int per_plane, float eps, int act, float slope, int dtype, pcgan_stream_t s);
/* y_pmax / dx_pmax ([N*C], may be NULL) here and y_cmax / dx_cmax ([C], may be NULL) of the fused BatchNorm calls below receive
 * the largest magnitude of each output plane / channel: partial maxima for the fp16 route of the convolution that consumes the
 * tensor (pcgan_conv2d_fwd_packed_hsplit: x_amax, n_amax) -- no separate pcgan_absmax pass. */
/* backward statistics per plane: s1[nc] = sum g, s2[nc] = sum g*xhat where
 * g = dy * act'(y) (y = forward output, used only for the activation mask; may be NULL
 * when act == NONE). */
int pcgan_norm_bwd_stats(const void* dy, const void* x, const void* y, const float* mean,
                         const float* var, float* s1_nc, float* s2_nc, int N, int C, int HW,
                         int per_plane, float eps, int act, float slope, int dtype, pcgan_stream_t s);
/* dx = rstd*gamma*( g - s1/cnt - xhat*s2/cnt ); s1/s2 indexed like mean;
 * d_residual (optional) = g.  For batch norm the caller first sums s1/s2 over n
 * (pcgan_bn_bwd_reduce) which also yields dgamma, dbeta. */
int pcgan_norm_bwd_apply(const void* dy, const void* x, const void* y, const float* mean,
                         const float* var, const float* gamma, const float* s1, const float* s2,
                         void* dx, void* d_residual, float* dx_pmax, int N, int C, int HW, int per_plane, float eps,
                         int act, float slope, int dtype, pcgan_stream_t s);
int pcgan_bn_bwd_reduce(const float* s1_nc, const float* s2_nc, float* s1_c, float* s2_c, int N, int C,
                        pcgan_stream_t s);
/* The two-launch pairs above as ONE launch each, for train-mode BatchNorm2d on tensors too large for the fused kernels below
 * (the Elo encoder's 112^2 / 56^2 / 28^2 maps, the PatchGAN's 32^2 map at batch 32: models/resnet.py:47-71, models/networks.py:756-761).
 * pcgan_bn_stats_merged = pcgan_plane_stats + pcgan_bn_merge (+ num_batches_tracked += 1 when `batches` != NULL);
 * pcgan_bn_bwd_stats_reduced = pcgan_norm_bwd_stats(per_plane = 0) + pcgan_bn_bwd_reduce.  The workgroup whose plane completes a
 * channel ("last arriver", counted on ticket[c]) finishes that channel with the stand-alone kernels' arithmetic in their order, so the
 * results are the same bits.  ticket: C unsigned words owned by the caller, zeroed ONCE when the layer is created; the arrival that finds
 * old + 1 == N is the last of its call and stores 0 back, so every call starts from 0 whatever N the previous call had (partial last
 * batch of an epoch, test() at another batch size) and a captured graph replays; forward and backward use separate arrays; calls sharing an array must be
 * ordered (one stream, or stream dependencies) -- a BatchNorm layer's passes are ordered anyway for its running statistics. */
int pcgan_bn_stats_merged(const void* x, float* mean_nc, float* m2_nc, float* mean_c, float* var_c, float* running_mean,
                          float* running_var, long long* batches, unsigned int* ticket, int N, int C, int HW, float momentum, int dtype,
                          pcgan_stream_t s);
int pcgan_bn_bwd_stats_reduced(const void* dy, const void* x, const void* y, const float* mean_c, const float* var_c, float* s1_nc,
                               float* s2_nc, float* s1_c, float* s2_c, unsigned int* ticket, int N, int C, int HW, float eps, int act,
                               float slope, int dtype, pcgan_stream_t s);
/* BatchNorm2d in training mode for small tensors (used up to N * HW = 8192 per channel), ONE launch per pass: batch statistics,
 * running-statistics update (+ num_batches_tracked when `batches` != NULL), normalise + affine (+ residual) + activation
 * -- replaces nn.BatchNorm2d (+ the following ReLU / LeakyReLU) of the PatchGAN and of the Elo encoder's late stages
 * (models/networks.py:24-26, 756-771; models/resnet.py:58-71).  mean_c / var_c (biased) are kept for the backward pass;
 * backward returns s1_c = d(beta), s2_c = d(gamma) and dx (dx / dres may be NULL). */
int pcgan_bn_fwd_fused(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                       float* mean_c, float* var_c, float* running_mean, float* running_var, long long* batches, float* y_cmax,
                       int N, int C, int HW, float momentum, float eps, int act, float slope, int dtype, pcgan_stream_t s);
int pcgan_bn_bwd_fused(const void* dy, const void* x, const void* y, const float* mean_c, const float* var_c,
                       const float* gamma, void* dx, void* dres, float* s1_c, float* s2_c, float* dx_cmax, int N, int C, int HW,
                       float eps, int act, float slope, int dtype, pcgan_stream_t s);

/* Fused instance norm (the generator's 23 norm sites per pass, models/networks.py:580-601,633,646): the
 * (n,c) plane stays in registers, so forward = one read + one write (statistics + normalise + residual +
 * activation; mean / M2 are also returned for the backward and the running-stat update) and backward = one
 * read of dy, x (and y for the activation mask) + one write.  Planes that do not fit (HW % 4 != 0 or
 * HW > 65536) fall back to the two-pass kernels above; ws_s1s2 (2*N*C floats) is only used then. */
int pcgan_instnorm_fwd(const void* x, const void* residual, void* y, float* mean_nc, float* m2_nc, float* y_pmax, int N, int C,
                       int HW, float eps, int act, float slope, int dtype, pcgan_stream_t s);
int pcgan_instnorm_bwd(const void* dy, const void* x, const void* y, const float* mean_nc, const float* m2_nc,
                       void* dx, float* dx_psum, float* dx_pmax, float* ws_s1s2, int N, int C, int HW, float eps, int act,
                       float slope, int dtype, pcgan_stream_t s);
/* y_pmax / dx_pmax (may be NULL; register-resident kernels only, see pcgan_instnorm_fused): [N*C] floats that receive the largest
 * magnitude of each output plane -- the fp16 route of the convolution that consumes the tensor scales by the largest of them
 * (pcgan_conv2d_fwd_hsplit with n_amax = N*C), without a separate pcgan_absmax pass over the tensor. */
/* 1 when a plane of HW elements runs in the register-resident kernels.  Only then may pcgan_instnorm_bwd be given
 * dx_psum[N*C]: the sum of dx over each plane, taken from the registers that store dx.  The convolution in front of the
 * norm has its bias gradient = sum over n of these (reference: autograd of nn.Conv2d(bias=True) + nn.InstanceNorm2d,
 * models/networks.py:580-601, 621-648; true value 0, fp32 noise) -- pcgan_sum_planes finishes it without re-reading dx. */
int pcgan_instnorm_fused(int HW);
/* out[c] (+)= sum over n of part_nc[n*C + c]: second stage of pcgan_channel_sum on plane sums that already exist. */
int pcgan_sum_planes(const float* part_nc, float* out, int N, int C, int accumulate, pcgan_stream_t s);

/* ---- pooling / resize ------------------------------------------------------------
 * MaxPool2d(k, stride, pad)   models/resnet.py:138 ; models/networks.py:1225-1235
 * AvgPool2d(H) / MaxPool2d(H) global pooling  models/networks.py:1056-1059
 * F.interpolate(bilinear, align_corners=True)  util/util.py:111-117 */
int pcgan_maxpool_fwd(const void* x, void* y, int32_t* argmax, int NC, int H, int W, int k, int stride,
                      int pad, int P, int Q, int dtype, pcgan_stream_t s);
int pcgan_maxpool_bwd(const void* dy, const int32_t* argmax, void* dx, int NC, int H, int W, int k, int stride,
                      int pad, int P, int Q, int dtype, pcgan_stream_t s);
int pcgan_global_pool_fwd(const void* x, void* y, int32_t* argmax, int NC, int HW, int is_max, int dtype,
                          pcgan_stream_t s);
int pcgan_global_pool_bwd(const void* dy, const int32_t* argmax, void* dx, int NC, int HW, int is_max, int dtype,
                          pcgan_stream_t s);
int pcgan_bilinear_fwd(const void* x, void* y, int NC, int H, int W, int P, int Q, int dtype, pcgan_stream_t s);
int pcgan_bilinear_bwd(const void* dy, void* dx, int NC, int H, int W, int P, int Q, int dtype, pcgan_stream_t s);

/* ---- losses ----------------------------------------------------------------------
 * nn.BCELoss(mean) against a per-sample target broadcast over the patch map
 * (GANLoss, models/networks.py:386-420; log clamped at -100 like torch);
 * nn.L1Loss / nn.MSELoss (models/wsgan_emb_model.py:141-149).
 * Each writes the scalar loss to loss[0] and, when grad != NULL, dloss/dpred * gscale. */
int pcgan_bce_loss(const void* pred, const float* target_n, float* loss, void* grad, int N, int per_n,
                   float gscale, void* ws, size_t ws_bytes, int dtype, pcgan_stream_t s);
int pcgan_l1_loss(const void* a, const void* b, float* loss, void* grad_a, size_t n, float gscale,
                  void* ws, size_t ws_bytes, int dtype, pcgan_stream_t s);
int pcgan_mse_loss(const void* a, const void* b, float* loss, void* grad_a, size_t n, float gscale,
                   void* ws, size_t ws_bytes, int dtype, pcgan_stream_t s);
size_t pcgan_loss_workspace_bytes(size_t n);

/* ---- optimizer -------------------------------------------------------------------
 * torch.optim.Adam(lr, betas=(beta1, 0.999), eps=1e-8) on a flat fp32 parameter
 * buffer (models/wsgan_emb_model.py:153-156).  step is the 1-based step count. */
int pcgan_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                    float beta1, float beta2, float eps, int step, pcgan_stream_t s);
/* hipGraph-capturable variant: lr_dev[0] (float) and step_dev[0] (int) live in device
 * memory; the step counter is incremented on the stream before the update reads it. */
int pcgan_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                        const float* lr_dev, int* step_dev, float beta1, float beta2, float eps,
                        pcgan_stream_t s);

/* ---- loader image pipeline ----------------------------------
 * transforms.Resize([loadSize, loadSize], Image.BICUBIC) -> RandomCrop(fineSize) ->
 * RandomHorizontalFlip -> ToTensor -> Normalize(.5, .5) of get_transform
 * (data/base_dataset.py:24-64) and the RGB -> gray mix of the pair dataset
 * (data/wsgan_emb_dataset.py:46-49), for n equally sized uint8 RGB images
 * src[n][H][W][3] -> out[dst][out_channels][FH][FW] fp32.  Pillow's two-pass 22-bit
 * fixed-point resampling (horizontal first, uint8 intermediate): the caller builds the
 * coefficient tables kh[RW][ksize_h] / kv[RH][ksize_v] and bounds bh[RW][2] / bv[RH][2]
 * (first source index, tap count) the way Pillow's precompute_coeffs /
 * normalize_coeffs_8bpc do; a dimension that keeps its size gets the identity table
 * (ksize 1, coefficient 1 << 22).  aug[n][4] (device) = crop x0, crop y0, flip, dst index;
 * the caller guarantees 0 <= x0 <= RW - FW, 0 <= y0 <= RH - FH.  Integer arithmetic up to
 * the last three correctly rounded fp32 operations: bit-exact with the PIL path.
 * pcgan_image_transform_band (host only, bv_host in host memory) picks the band of output
 * rows one workgroup takes and the LDS rows it needs. */
typedef struct {
    int H, W;     /* source image                     */
    int RH, RW;   /* after the resize (loadSize)      */
    int FH, FW;   /* after the crop (fineSize)        */
    int ksize_h, ksize_v;
    int out_channels; /* 3, or 1 = gray mix           */
} pcgan_image_desc;
int pcgan_image_transform_band(const pcgan_image_desc* d, const int32_t* bv_host, int* band, int* max_rows);
int pcgan_image_transform(const pcgan_image_desc* d, const uint8_t* src, const int32_t* kh, const int32_t* bh,
                          const int32_t* kv, const int32_t* bv, const int32_t* aug, float* out, int n, int band,
                          int max_rows, pcgan_stream_t s);

/* ---- convolution on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulators) -----------
 * Same call sites as pcgan_conv2d_fwd_packed for stride-1 convolutions with C % 16 == 0, K >= 32, at most 25 taps
 * (the ResnetBlock convolutions, models/networks.py:621-648).
 *   desc.dtype = PCGAN_F32:  every fp32 operand is split exactly into three bf16 pieces and the six piece products that
 *                            matter are accumulated in fp32: error as the fp32 MFMA path (scripts/micro/bf16_split.hip).
 *                            The host routes the residual-block convolutions here by default (PCGAN_BF16X6=0: fp32 MFMA).
 *   desc.dtype = PCGAN_BF16: the bf16 path: bf16 activations as stored, weights rounded to bf16 by the pack call, one product.
 * Pack once per weight version. */
int pcgan_conv2d_bsplit_supported(const pcgan_conv_desc* d);
size_t pcgan_conv2d_bsplit_packed_bytes(const pcgan_conv_desc* d);
int pcgan_conv2d_bsplit_pack(const pcgan_conv_desc* d, const float* w, void* packed, pcgan_stream_t s);
int pcgan_conv2d_fwd_bsplit(const pcgan_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                            int act, float slope, pcgan_stream_t s);
/* the same for the data gradient of nn.ReflectionPad2d(1) + nn.Conv2d(k=3, stride 1) (the ResnetBlock convolutions): row
 * mirrors folded into three per-row-class weight sets by the pack call, column mirrors gathered as a second source. */
int pcgan_conv2d_bsplit_dgrad_supported(const pcgan_conv_desc* d);
size_t pcgan_conv2d_bsplit_dgrad_packed_bytes(const pcgan_conv_desc* d);
int pcgan_conv2d_bsplit_dgrad_pack(const pcgan_conv_desc* d, const float* w, void* packed, pcgan_stream_t s);
int pcgan_conv2d_bwd_data_bsplit(const pcgan_conv_desc* d, const void* dy, const void* packed, void* dx, pcgan_stream_t s);
/* ... and for its weight gradient: the same GEMM with the roles turned (rows = output channels, operand A = dy re-split per
 * call, columns = (c, r, s), reduction over pixels in splits + a fixed-order reduce); ws holds the reflection-padded copy of x,
 * the pieces of dy and the partial sums.  accumulate != 0 adds into dw like pcgan_conv2d_bwd_weight. */
int pcgan_conv2d_bsplit_wgrad_supported(const pcgan_conv_desc* d);
size_t pcgan_conv2d_bsplit_wgrad_workspace_bytes(const pcgan_conv_desc* d);
int pcgan_conv2d_bwd_weight_bsplit(const pcgan_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* ws,
                                   size_t ws_bytes, pcgan_stream_t s);

/* ---- the same residual-block convolutions (nn.ReflectionPad2d(1) + nn.Conv2d(dim, dim, 3), models/networks.py:621-648; fp32
 * tensors) with TWO fp16 pieces per operand and three products -- half the matrix instructions of the three-piece bf16 split at the
 * same measured error (scripts/micro/bf16_split: 5.3e-7 relative L2 at K = 2304, fp32 MFMA 6.1e-7).  fp16 has five exponent
 * bits: every operand tensor is scaled by the power of two that puts its largest magnitude into (2^13, 2^14], and the result is
 * scaled back exactly.  The largest magnitude is a DEVICE value (no host synchronisation): x_amax[0 .. n_amax) holds partial
 * maxima, the kernel takes the largest -- one float written by pcgan_absmax, or the per-plane maxima the producing kernel
 * hands over (pcgan_instnorm_* pmax outputs).  It must be a true bound: a larger element would overflow.
 * pass = PCGAN_PASS_FWD | PCGAN_PASS_BWD_DATA; image width 32 or 64, 128 / width rows dividing the height, gathered channels a
 * multiple of 32, produced channels a multiple of 256.  packed: pcgan_conv2d_hsplit_packed_bytes(d, pass) bytes, valid for any
 * batch size; the pack call also stores the largest magnitude of every weight row in it (one power-of-two scale per row). */
/* out[0 .. slots) = partial maxima of |x| (one per workgroup, 1 <= slots <= 1024; no atomics, nothing to clear beforehand);
 * pcgan_absmax_slots(n) = the count that keeps the pass bandwidth-bound.  Weight tensors use exactly 64 slots. */
int pcgan_absmax_slots(size_t n);
int pcgan_absmax(const void* x, size_t n, int dtype, float* out, int slots, pcgan_stream_t s);
/* Audit of the maxima a tensor carries (the fp16 route trusts them: a claim that is too SMALL overflows an fp16 piece -- the
 * non-finite sentinel below reports that; a claim that is too LARGE, e.g. after the tensor was rewritten with smaller values through
 * `.data` / a raw pointer, pushes the low pieces into subnormals and loses bits silently).  claimed[0 .. n_claimed) = the attached
 * partial maxima, fresh[0 .. n_fresh) = the partials of a pcgan_absmax pass over the tensor as it is now; both are maxima of the same
 * stored values, so they must agree exactly.  counts (three device words owned by the caller): [0] += 1 if the tensor holds more than
 * claimed, [1] += 1 if its largest value is below 2^-8 of the claim, [2] += 1 for any other mismatch.  One tiny launch; the host
 * (hip/ops.py: amax_of) audits every n-th consumption of attached maxima (every one in the test suite). */
int pcgan_amax_audit(const float* claimed, int n_claimed, const float* fresh, int n_fresh, unsigned int* counts, pcgan_stream_t s);
int pcgan_conv2d_hsplit_supported(const pcgan_conv_desc* d, int pass);
size_t pcgan_conv2d_hsplit_packed_bytes(const pcgan_conv_desc* d, int pass);
int pcgan_conv2d_hsplit_pack(const pcgan_conv_desc* d, int pass, const float* w, void* packed, pcgan_stream_t s);
int pcgan_conv2d_fwd_hsplit(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_amax, const void* packed,
                            const float* bias, void* y, int act, float slope, pcgan_stream_t s);
int pcgan_conv2d_bwd_data_hsplit(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax, const void* packed,
                                 void* dx, pcgan_stream_t s);
/* ... the same with dx = conv^T(dy, w) + add: the gradient that reaches a ResnetBlock's input through the skip connection
 * (`x + conv_block(x)`, models/networks.py:650-652) summed in the epilogue instead of by a separate `grad +=` pass.  add may be NULL. */
int pcgan_conv2d_bwd_data_hsplit_add(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax, const void* packed,
                                     const void* add, void* dx, pcgan_stream_t s);
/* Every other convolution whose gathered channel count is a multiple of 16 (<= 25 taps, more than 32 produced channels; any
 * stride; forward with zero / reflection padding, data gradient with zero padding -- incl. nn.ConvTranspose2d forward,
 * models/networks.py:584-602, 734-763; models/resnet.py): the fp16 two-piece form of pcgan_conv2d_fwd_packed /
 * pcgan_conv2d_bwd_data_packed.  Same workspace and semantics; packed weights from pcgan_conv2d_hgemm_pack (desc.dtype = PCGAN_BF16:
 * from pcgan_conv2d_pack_weights, w_amax unused); w_amax = the row maxima pcgan_conv2d_hgemm_pack wrote. */
int pcgan_conv2d_hgemm_supported(const pcgan_conv_desc* d, int pass);
/* packed weights of the two calls below: pcgan_conv2d_pack_weights' image with every group of 4 consecutive k of a row PRE-SPLIT into
 * [4 fp16 high pieces | 4 fp16 low pieces] of the weights scaled by ONE POWER OF TWO PER ROW of the pass's weight matrix (forward: per
 * output channel; data gradient: per input channel) -- the weights change once per optimizer step while each net runs 2-4 times in
 * between, so the kernel's weight path is a plain copy, and a filter row far below the tensor's largest weight keeps its 22 bits.
 * w_rowmax (K floats for PCGAN_PASS_FWD, C for PCGAN_PASS_BWD_DATA) is WRITTEN by the pack call -- the largest magnitude of every row --
 * and must be handed to the convolution calls as `w_amax` (their epilogue divides row m by pow2(w_rowmax[m])).
 * Size of packed: pcgan_conv2d_packed_bytes. */
int pcgan_conv2d_hgemm_pack(const pcgan_conv_desc* d, int pass, const float* w, float* w_rowmax, float* packed, pcgan_stream_t s);
int pcgan_conv2d_fwd_packed_hsplit(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_amax, const float* packed,
                                   const float* w_amax, const float* bias, void* y, int act, float slope, void* ws, size_t ws_bytes,
                                   pcgan_stream_t s);
int pcgan_conv2d_bwd_data_packed_hsplit(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax,
                                        const float* packed, const float* w_amax, const float* bias, void* dx, void* ws,
                                        size_t ws_bytes, pcgan_stream_t s);
/* weight gradient on the same route: stride 1 or 2, reflection (stride 1) or zero padding, <= 49 taps, >= 32 output channels -- the
 * residual blocks, the generator's down / up-sampling layers, the PatchGAN's layers (models/networks.py:584-648, 753-775).  Both
 * operands are split on their way to LDS, so there is no packed copy of dy; ws (pcgan_conv2d_hsplit_wgrad_workspace_bytes) holds the
 * padded copy of x (when one is made) and the partial sums of the splits of the pixel reduction, which are combined in a fixed order.
 * Zero padding <= 1: the padding is applied inside the gather (no padded copy); with fp32 tensors the output width may also be ragged
 * (at least 3/4 of its multiple of 16: the PatchGAN's 15 x 15 layer) and K may exceed 256 (row tiles); otherwise (bf16 tensors, other
 * paddings) the output width is a multiple of 16 and K <= 256.  pcgan_conv2d_hsplit_wgrad_inline: 1 when the call makes no padded copy of x (routing: the
 * copy of a 134 MB input costs more than the matrix pipe saves).  accumulate != 0 adds into dw like pcgan_conv2d_bwd_weight.
 * desc.dtype = PCGAN_BF16: the one-product bf16 form (the maxima pointers may be NULL). */
int pcgan_conv2d_hsplit_wgrad_supported(const pcgan_conv_desc* d);
int pcgan_conv2d_hsplit_wgrad_inline(const pcgan_conv_desc* d);
size_t pcgan_conv2d_hsplit_wgrad_workspace_bytes(const pcgan_conv_desc* d);
int pcgan_conv2d_bwd_weight_hsplit(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_xamax, const void* dy,
                                   const float* dy_amax, int n_dyamax, float* dw, int accumulate, void* ws, size_t ws_bytes,
                                   pcgan_stream_t s);


/* ---- composite: one ResnetBlock per call -----------------------------------------------------------------------------------------
 * `x + conv_block(x)` with conv_block = ReflectionPad2d(1), Conv2d(dim, dim, 3, bias), InstanceNorm2d, ReLU, ReflectionPad2d(1),
 * Conv2d(dim, dim, 3, bias), InstanceNorm2d (models/networks.py:616-652; 9 blocks x 2 generator passes per step).  The per-op entry
 * points above cost the host one call per launch (6 forward, 10 backward per block: ~12 ms of a 32 ms step); these two launch exactly
 * the same kernels with the same arguments in the same order -- results are bit-identical to the per-op sequence -- from ONE call.
 * fp32 tensors on the fp16 two-piece route, or (round 4) bf16 tensors on the one-product forms (pcgan_resblock_supported: the shapes
 * the residual-convolution kernels and pcgan_instnorm_fused take); other shapes use the per-op calls.
 *
 * forward   y1 = conv1(x) ; h = relu(IN(y1)) ; y2 = conv2(h) ; out = IN(y2) + x.  pk1 / pk2: pcgan_conv2d_hsplit_pack(PASS_FWD).
 *           stats[4][N*C] receives mean1, M2_1, mean2, M2_2 (kept for the backward pass); amax[2][N*C] the plane maxima of h and
 *           out (operand maxima of the next convolution); running statistics are updated in place.
 * backward  given dout: dy2 = IN2'(dout) ; dw2 += wgrad(h, dy2), db2 += sum dy2 ; dh = dgrad2(dy2) ; dy1 = (IN1 o relu)'(dh) ;
 *           dw1 += wgrad(x, dy1), db1 += sum dy1 ; dx = dgrad1(dy1) + dout (skip connection).  pk1b / pk2b:
 *           pcgan_conv2d_hsplit_pack(PASS_BWD_DATA).  The weight / bias gradients are launched on `side` (ordered behind the producer of
 *           their operands through `fork_event`, an event the caller owns: pcgan_event_create); the caller joins `side` before it reads
 *           dw / db.  scratch: 5*N*C floats (plane sums and maxima of dy2 / dy1); wgrad_ws: pcgan_resblock_wgrad_workspace_bytes(d),
 *           used on `side` only.  dy2, dh, dy1 are caller-allocated temporaries the size of x. */
typedef struct {
    int N, C, H, W;
    float eps, momentum;
    int dtype; /* PCGAN_F32: the fp16 two-piece route (x_amax / amax used); PCGAN_BF16 (round 4): bf16 tensors on the one-product forms --
                  pk*: pcgan_conv2d_bsplit_pack / pcgan_conv2d_bsplit_dgrad_pack, x_amax / h_amax / amax unused (may be NULL) */
} pcgan_resblock_desc;
typedef void* pcgan_event_t; /* hipEvent_t */
int pcgan_event_create(pcgan_event_t* ev);
int pcgan_event_destroy(pcgan_event_t ev);
int pcgan_resblock_supported(const pcgan_resblock_desc* d);
size_t pcgan_resblock_wgrad_workspace_bytes(const pcgan_resblock_desc* d);
int pcgan_resblock_fwd(const pcgan_resblock_desc* d, const void* x, const float* x_amax, int n_xamax, const void* pk1, const float* b1,
                       const void* pk2, const float* b2, float* rm1, float* rv1, float* rm2, float* rv2, void* y1, void* h, void* y2,
                       void* out, float* stats, float* amax, pcgan_stream_t s);
int pcgan_resblock_bwd(const pcgan_resblock_desc* d, const void* dout, const void* x, const float* x_amax, int n_xamax, const void* y1,
                       const void* h, const float* h_amax, const void* y2, const float* stats, const void* pk1b, const void* pk2b,
                       float* dw1, float* db1, float* dw2, float* db2, void* dy2, void* dh, void* dy1, void* dx, float* scratch,
                       void* wgrad_ws, size_t wgrad_ws_bytes, pcgan_stream_t s, pcgan_stream_t side, pcgan_event_t fork_event);

/* ---- convolutions that gather <= 4 channels, on the same fp16 two-piece route (csrc/thin_conv.hip) --------------------------------------
 * pass = PCGAN_PASS_FWD: C <= 4 input channels, K a multiple of 64 (<= 256), 7x7 stride 1 / 2 or 4x4 stride 2, zero padding or (stride 1)
 * reflection padding -- ReflectionPad2d(3) + Conv2d(input_nc + nz, ngf, 7) of the generator (models/networks.py:578-581), conv1 of the Elo
 * encoder's ResNet-18 (models/resnet.py:108-111), the first PatchGAN layer (models/networks.py:753-755).
 * pass = PCGAN_PASS_BWD_DATA: K <= 4 output channels of the convolution, C a multiple of 64, 7x7 stride 1 -- autograd's data gradient of
 * the generator's head ReflectionPad2d(3) + Conv2d(ngf, output_nc, 7) (models/networks.py:603-605), computed as a forward-form
 * convolution of dy with flipped weights; with reflection padding on the padded grid (ws: pcgan_conv2d_thin_workspace_bytes) and folded.
 * One workgroup = 8 x 32 output pixels x 64 channels; its input window is loaded, scaled and split ONCE into LDS as [position][4 channels],
 * and with k = 4 tap + channel an MFMA lane's 8 consecutive k are two adjacent taps = two 8-byte LDS reads.  packed
 * (pcgan_conv2d_thin_packed_bytes): the weights pre-scaled per output row, pre-split, in MFMA fragment order + the row maxima; x_amax /
 * n_amax as for the other fp16-route calls (partial maxima of |x| on the device).  fp32 tensors only. */
int pcgan_conv2d_thin_supported(const pcgan_conv_desc* d, int pass);
size_t pcgan_conv2d_thin_packed_bytes(const pcgan_conv_desc* d, int pass);
size_t pcgan_conv2d_thin_workspace_bytes(const pcgan_conv_desc* d, int pass);
int pcgan_conv2d_thin_pack(const pcgan_conv_desc* d, int pass, const float* w, void* packed, pcgan_stream_t s);
int pcgan_conv2d_fwd_thin(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_amax, const void* packed, const float* bias,
                          void* y, int act, float slope, pcgan_stream_t s);
int pcgan_conv2d_bwd_data_thin(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax, const void* packed, void* dx,
                               void* ws, size_t ws_bytes, pcgan_stream_t s);

/* A RUN of `nblocks` consecutive ResnetBlocks per call (round 4: the generator's nine blocks, models/networks.py:589-592): the
 * launches of nblocks pcgan_resblock_fwd / _bwd calls, block i reading block i - 1's output and plane maxima, from one host call.
 * Per-block tensors are slices of stacked buffers: y1, h, y2, out [nblocks][N*C*H*W]; stats [nblocks][4*N*C]; amax [nblocks][2*N*C];
 * backward: dy2, dy1 [nblocks][N*C*H*W] (read by `side` after the call), dh [N*C*H*W], dxs [2][N*C*H*W] (gradient handed from block
 * to block), scratch [nblocks][5*N*C]; dx = gradient of the chain's input.  Per-block parameter pointers (packed weights, biases,
 * running statistics, gradient slots; a bias / running-statistics entry may be NULL) come as HOST arrays of nblocks pointers. */
int pcgan_restrunk_fwd(const pcgan_resblock_desc* d, int nblocks, const void* x, const float* x_amax, int n_xamax, const void* const* pk1,
                       const float* const* b1, const void* const* pk2, const float* const* b2, float* const* rm1, float* const* rv1,
                       float* const* rm2, float* const* rv2, void* y1, void* h, void* y2, void* out, float* stats, float* amax,
                       pcgan_stream_t s);
int pcgan_restrunk_bwd(const pcgan_resblock_desc* d, int nblocks, const void* dout, const void* x, const float* x_amax, int n_xamax,
                       const void* y1, const void* h, const void* y2, const void* out, const float* stats, const float* amax,
                       const void* const* pk1b, const void* const* pk2b, float* const* dw1, float* const* db1, float* const* dw2,
                       float* const* db2, void* dy2, void* dh, void* dy1, void* dxs, void* dx, float* scratch, void* wgrad_ws,
                       size_t wgrad_ws_bytes, pcgan_stream_t s, pcgan_stream_t side, pcgan_event_t fork_event);

/* ---- weight gradient of the residual-block convolution, "image-innermost" form (round 4; csrc/wgrad_direct.hip) -------------------------
 * nn.Conv2d(dim, dim, 3) behind nn.ReflectionPad2d(1) in ResnetBlock (models/networks.py:621-648), autograd's weight gradient; fp32 tensors,
 * fp16 two-piece arithmetic as pcgan_conv2d_bwd_weight_hsplit (same operand maxima, same error level).  A transposing pre-pass scales and
 * splits x and dy ONCE into 16-byte records of 8 images per (pixel, channel) -- a filter tap then shifts the pixel, not the position
 * inside a record -- and the main kernel loads every MFMA operand fragment straight from memory: no LDS, no barrier, no split arithmetic
 * in the loop (the per-tap form re-gathers and re-splits x nine times and is LDS-bound).  3x3, stride 1, reflection padding 1,
 * N % 16 == 0, K % 128 == 0, C % 32 == 0.  ws: pcgan_conv2d_wgrad_direct_workspace_bytes(d) bytes (the four piece arrays, the split
 * partials, two scale words).  dw[K][C][3][3], accumulate != 0: dw += (the optimizer's gradient buffer). */
int pcgan_conv2d_wgrad_direct_supported(const pcgan_conv_desc* d);
/* ---- ... and in the "row ring" form (round 4; csrc/wgrad_rowring.hip): a workgroup's column tile is 32 input channels x all nine taps and it
 * walks down a 16-pixel-wide strip of one image, keeping the padded rows in a ring of four LDS slots -- every element of x is loaded and split
 * once per strip instead of once per tap.  Same arguments, workspace rule and results level as pcgan_conv2d_bwd_weight_hsplit, which hands the
 * shapes this form takes to it under option "wgrad_rowring": fp32 or bf16 tensors, 3x3, stride 1, reflection padding 1, W % 16 == 0, K % 128 == 0,
 * C % 32 == 0, H >= 3; bf16 tensors too (one bf16 product per tap, x_amax / dy_amax unused and may be NULL).  Replaces autograd's weight
 * gradient of nn.Conv2d in ResnetBlock (models/networks.py:616-652). */
int pcgan_conv2d_wgrad_rowring_supported(const pcgan_conv_desc* d);
size_t pcgan_conv2d_wgrad_rowring_workspace_bytes(const pcgan_conv_desc* d);
int pcgan_conv2d_bwd_weight_rowring(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_xamax, const void* dy, const float* dy_amax,
                                    int n_dyamax, float* dw, int accumulate, void* ws, size_t ws_bytes, pcgan_stream_t s);
size_t pcgan_conv2d_wgrad_direct_workspace_bytes(const pcgan_conv_desc* d);
int pcgan_conv2d_bwd_weight_direct(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_xamax, const void* dy,
                                   const float* dy_amax, int n_dyamax, float* dw, int accumulate, void* ws, size_t ws_bytes, pcgan_stream_t s);

/* ---- kernel timer (measurement only) -----------------------------------------------------------------------------------------------
 * bench.py's roofline block: HIP events on the launch stream around every launch of the three residual-block convolution kernels
 * (kind 0 forward, 1 data gradient, 2 weight gradient incl. its padded copy and reduce, 3 the weight gradient's main kernel),
 * whichever entry point issued them.  pcgan_timer_enable(capacity) creates `capacity` event pairs per kind and switches recording on
 * (0: off, events destroyed); pcgan_timer_read waits for the recorded pairs of one kind, writes their durations in ms (at most cap) and
 * resets that kind; returns the count or -1.  Off by default, never records inside a graph capture. */
/* Non-finite sentinel of the fp16 two-piece route: the kernels that scale their operands by a device-side maximum count the waves that
 * produced an inf / NaN result into *dev_word (a device word the caller owns and zeroes; NULL switches the sentinel off).  An operand
 * element above the maximum its scale came from -- a stale maximum -- overflows fp16 and poisons everything it touches, so a non-zero
 * count means "stale operand maxima or non-finite inputs"; the host checks it where it synchronises anyway. */
int pcgan_set_nonfinite_counter(unsigned int* dev_word);
/* Routing options below the C-ABI (A/B measurement; the defaults are the measured-best settings and need no call).  Set before the
 * calls they affect; *_supported / *_workspace_bytes queries follow the current values.
 *   "bsplit_halo"   1  residual convolutions (3x3 reflect, width 32 / 64) on the window kernel; 0: per-tap gather kernel
 *   "wgrad_gen"     1  weight gradients with zero padding <= 1 apply the padding inside the gather; 0: padded copy of x
 *   "wgrad_padcopy" 0  1: the residual blocks' weight gradient reads a reflection-padded copy of x (round 2's form)
 *   "wgrad_cw"      0  columns per workgroup of the matrix-pipe weight gradient: 0 = as measured (128 for fp32 tensors, 256 for bf16),
 *                      128 / 256 force one form
 *   "hgemm_bf16"    1  bf16 tensors on the one-product bf16 MFMA form of the packed implicit GEMM; 0: fp32 MFMA kernels
 *   "wgrad_direct"  0  1: pcgan_conv2d_bwd_weight_hsplit hands the shapes pcgan_conv2d_wgrad_direct_supported takes to the
 *                      image-innermost form below (round-4 experiment: same results level, no step gain); the workspace query follows
 *   "wgd_look"      3  K steps the operand loads of that form's main kernel run ahead of its MFMAs (2, 3 or 4)
 *   "wgrad_rowring" 1  pcgan_conv2d_bwd_weight_hsplit hands the shapes pcgan_conv2d_wgrad_rowring_supported takes to the row-ring form
 *                      (fp32 tensors: 0.125-0.137 ms against 0.158-0.166 per residual-block call, step + 0.5 %; bf16 tensors: 0.060
 *                      against 0.119, step + 7 %); 0: the per-tap kernel for every shape; 3: the row-ring form for fp32 tensors only
 *   "hgemm_tile"    0  packed implicit GEMM, layers with > 32 output rows: 0 = the library's tile heuristic; BM * 1000 + BP with
 *                      BM, BP in {64, 128} forces the workgroup tile (measurement: scripts/sweep_hgemm.py)
 *   "hgemm_ks"      0  ... and the K split: 0 = heuristic, 1 .. 8 forced (every layer's workspace query then includes the partial sums)
 * Unknown keys / values return non-zero. */
int pcgan_set_option(const char* key, int value);
int pcgan_get_option(const char* key, int* value);
int pcgan_timer_enable(int capacity);
int pcgan_timer_read(int kind, float* ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* PCGAN_HIP_H */
