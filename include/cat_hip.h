/* cat_hip.h — C-ABI of libcat_hip.so: the MI355X (gfx950) kernels behind CAT's generator-distillation step.
 *
 * snap-research/CAT has no native code: every "kernel" on its hot path is a stock torch.nn op.
 * Each entry point below therefore replaces the torch op(s) a reference call site issues; the reference
 * file:line that issues them is cited per function.  The host binding is ctypes (cat_amd/_lib.py);
 * INTEGRATION.md shows the stub a CAT maintainer would add.
 *
 * Conventions
 *   - All tensors are fp32, resident in HBM, NHWC ("channels last").  An activation is described by
 *     (ptr, N, H, W, C, cs): cs = floats per pixel (>= C, multiple of 4); channels [C, cs) are padding that
 *     kernels keep at 0.0f.  ptr is 16-byte aligned.
 *   - Conv weights are [Cout][kh][kw][wcs] with wcs >= Cin (= torch OIHW in channels_last memory format; cat_amd stores
 *     them with wcs = round_up(Cin, 4) and zero padding so that every filter quad is one aligned float4 load);
 *     ConvTranspose2d weights are [Cin_t][kh][kw][wcs >= Cout_t] (= torch IOHW channels_last).
 *   - Every function enqueues on `stream` and returns immediately: 0 on success, negative on a bad
 *     argument / launch failure (text via cat_hip_last_error()).  Nothing is allocated; scratch is passed in.
 */
#ifndef CAT_HIP_H
#define CAT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* cat_stream_t; /* hipStream_t */

enum { CAT_PAD_ZERO = 0, CAT_PAD_REFLECT = 1 };
enum { CAT_ACT_NONE = 0, CAT_ACT_RELU = 1, CAT_ACT_LRELU = 2, CAT_ACT_TANH = 3, CAT_ACT_RELU6 = 4 /* nn.ReLU6: min(max(x, 0), 6) */ };
enum { CAT_NORM_INSTANCE = 0, CAT_NORM_BATCH = 1 };

const char* cat_hip_last_error(void);
int cat_hip_version(void);

/* Dense convolution, geometry shared by fwd / dgrad / wgrad.
 * Replaces nn.Conv2d (+ the nn.ReflectionPad2d in front of it) at models/modules/inception_modules.py:135-147,
 * inception_architecture/inception_generator.py:37-56,130-131 and models/modules/discriminators.py:40-76. */
typedef struct {
  int N, H, W, Cin, xcs;   /* input  activation (H, W = unpadded input size) */
  int Ho, Wo, Cout, ycs;   /* output activation */
  int kh, kw, stride, pad; /* pad = implicit padding on each side */
  int pad_mode;            /* CAT_PAD_ZERO | CAT_PAD_REFLECT */
  int act;                 /* fused epilogue activation (fwd only) */
  float slope;             /* LeakyReLU slope */
  int ycw;                 /* fwd: channels [Cout, ycw) of y are written as 0 (ycw<=ycs); 0 -> Cout */
  int wcs;                 /* weight floats per (cout, tap): w is [Cout][kh][kw][wcs], channels [Cin, wcs) zero; 0 -> Cin (dense) */
} cat_conv_t;

/* y = act(conv(x, w) + bias); bias may be NULL. */
int cat_conv2d_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y,
                   cat_stream_t stream);
/* dx = conv^T(dy, w) with the geometry of the forward conv g.  For CAT_PAD_REFLECT the gradient w.r.t. the
 * PADDED input is produced (dx has (H+2*pad) x (W+2*pad) pixels); fold it with cat_reflect_pad_bwd.
 * Also serves as nn.ConvTranspose2d forward (inception_generator.py:120-126): bias/act then apply to dx.
 * dxcw: channels [Cin, dxcw) of dx are zeroed. */
int cat_conv2d_dgrad(const cat_conv_t* g, const float* dy, const float* w, const float* bias, float* dx,
                     int dxcs, int dxcw, cat_stream_t stream);
/* Input gradient with a TRANSPOSED copy of the filters, wt = [Cin][kh][kw][Cout] (cat_conv2d_weight_transpose, refreshed whenever the
 * weights change: once per optimizer step for a trainable layer): both MFMA operands of the wide 128 x 128 x 32 tile are then K-contiguous
 * in LDS, like the forward kernel's.  cat_conv2d_dgrad_t_applicable says whether the layer takes that path; otherwise (or with wt = NULL)
 * the call is cat_conv2d_dgrad. */
int cat_conv2d_dgrad_t_applicable(const cat_conv_t* g);
int cat_conv2d_weight_transpose(const cat_conv_t* g, const float* w, float* wt, cat_stream_t stream);
int cat_conv2d_dgrad_t(const cat_conv_t* g, const float* dy, const float* w, const float* wt, const float* bias, float* dx, int dxcs,
                       int dxcw, cat_stream_t stream);
/* dw[Cout][kh][kw][Cin] (+)= sum_pixels dy * im2col(x).  `ws` is scratch of cat_conv2d_wgrad_ws_bytes(g) bytes.
 * accumulate != 0 adds into dw (gradient accumulation over several backward calls). */
size_t cat_conv2d_wgrad_ws_bytes(const cat_conv_t* g);
int cat_conv2d_wgrad(const cat_conv_t* g, const float* x, const float* dy, float* dw, int accumulate, void* ws,
                     cat_stream_t stream);
/* Several weight gradients whose partial sums are reduced by ONE launch (the narrow layers of a fused block: seven 5 - 8 us reduce launches
 * per block were launch latency).  Item i uses its own region of `ws` (cat_conv2d_wgrad_batch_ws_bytes); results are bit-identical to n
 * cat_conv2d_wgrad calls (same kernels, same summation order). */
#define CAT_WGRAD_BATCH_MAX 8
typedef struct {
  cat_conv_t g;
  const float* x;
  const float* dy;
  float* dw;
  int accumulate;
} cat_wgrad_item_t;
size_t cat_conv2d_wgrad_batch_ws_bytes(const cat_wgrad_item_t* items, int n);
int cat_conv2d_wgrad_batch(const cat_wgrad_item_t* items, int n, void* ws, cat_stream_t stream);
/* Split-K variants for layers whose output tile grid cannot fill 256 CUs (few pixels, very deep reduction: the 4x8 .. 32x64
 * blocks of the SPADE generators, inception_spade_generator.py:63-124; PatchGAN's 1024 -> 1 head, discriminators.py:72-74, whose
 * input channels are cut into slices).  *_ws_bytes returns 0 when the plain entry point is the right one; otherwise the caller
 * provides that much scratch and the library reduces the K slices (+ bias, activation). */
size_t cat_conv2d_fwd_ws_bytes(const cat_conv_t* g);
int cat_conv2d_fwd_ws(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y, void* ws,
                      cat_stream_t stream);
size_t cat_conv2d_dgrad_ws_bytes(const cat_conv_t* g, int dxcs);
int cat_conv2d_dgrad_ws(const cat_conv_t* g, const float* dy, const float* w, const float* bias, float* dx, int dxcs,
                        int dxcw, void* ws, cat_stream_t stream);

/* Depthwise conv (groups == C), stride 1, weights [C][kh][kw]; replaces the groups=midp ConvBNReLU conv at
 * inception_modules.py:166-173.  dgrad for reflect padding again yields the padded-input gradient.
 * fwd: g->act is applied in the epilogue; g->ycw > 0 makes the call operate on a channel SLICE of wider buffers (x and y
 * point at the slice's first channel, xcs / ycs are the buffers' pixel strides, ycw = channels incl. padding the slice owns). */
int cat_dwconv2d_fwd(const cat_conv_t* g, const float* x, const float* w, const float* bias, float* y,
                     cat_stream_t stream);
/* Several depthwise convs of different kernel sizes over ADJACENT channel slices of one buffer as one launch (csrc/dwconv.hip): per channel
 * quad a kernel size in {1, 3, 5}, filters in a 5 x 5 frame w25[25][4*nq] (centred, zero outside the k x k window), bias[4*nq] (or NULL),
 * y = act(conv + bias).  The frozen (eval-mode, BatchNorm-folded) teacher's InvertedResidualChannels block: its three groups=midp
 * ConvBNReLU convs (inception_modules.py:166-173) and the copy of the k = 1 residual branch's hidden slice (k = 1, centre weight 1) --
 * torch: 3 x F.conv2d(groups=C) + 3 x F.batch_norm + 3 x relu + a slice copy. */
#define CAT_DWMULTI_MAXQ 64
#define CAT_DWMULTI_MAXRUN 8   /* runs of consecutive quads with one kernel size */
typedef struct {
  int N, H, W;
  int nq;                  /* channel quads */
  int xcs, ycs;            /* pixel strides of x / y (x and y point at the first channel of the slice range) */
  int reflect;             /* source pixels outside the plane: 1 = mirrored, 0 = zero */
  int act;                 /* epilogue activation */
  float slope;
  int ks[CAT_DWMULTI_MAXQ];
} cat_dwmulti_t;
int cat_dwconv2d_multi_fwd(const cat_dwmulti_t* g, const float* x, const float* w25, const float* bias, float* y, cat_stream_t stream);
int cat_dwconv2d_dgrad(const cat_conv_t* g, const float* dy, const float* w, float* dx, int dxcs,
                       cat_stream_t stream);
int cat_dwconv2d_wgrad(const cat_conv_t* g, const float* x, const float* dy, float* dw, int accumulate, void* ws,
                       cat_stream_t stream);
size_t cat_dwconv2d_wgrad_ws_bytes(const cat_conv_t* g);

/* Backward of nn.ReflectionPad2d(pad): dx[N,H,W,C] = fold(dxp[N,H+2p,W+2p,C]). */
int cat_reflect_pad_bwd(const float* dxp, float* dx, int N, int H, int W, int C, int cs, int pad,
                        cat_stream_t stream);

/* nn.ReplicationPad2d(pad) (reference models/modules/inception_modules.py:114-115, padding_type='replicate'): y[N,H+2p,W+2p,C] with the border
 * pixels repeated, and its backward (dx gathers the padded positions that clamp onto each source pixel). */
int cat_replicate_pad_fwd(const float* x, float* y, int N, int H, int W, int C, int cs, int pad, cat_stream_t stream);
int cat_replicate_pad_bwd(const float* dyp, float* dx, int N, int H, int W, int C, int cs, int pad, cat_stream_t stream);

/* Per-channel sums over pixels: out[c] (+)= sum_m x[m*cs + c]  (conv bias gradient). */
int cat_channel_sum(const float* x, int M, int C, int cs, float* out, int accumulate, void* ws, cat_stream_t stream);
size_t cat_channel_sum_ws_bytes(int M, int cs);

/* InstanceNorm2d / BatchNorm2d (training statistics), fused with the activation that follows them.
 * Replaces norm_layer(...) + active_fn() at inception_modules.py:43-44 and networks.py:29-64 semantics:
 * biased variance, eps inside the sqrt, BatchNorm running stats use the unbiased variance.
 *   G = number of statistic groups (N for instance norm, 1 for batch norm); each group spans M/G pixels.
 *   save_mean / save_rstd: [G][C] outputs kept for the backward pass.
 *   running_mean / running_var: BatchNorm only, may be NULL.
 *   num_batches_tracked: BatchNorm only, may be NULL; incremented by one on the device (nn.BatchNorm2d's counter), so that no
 *   stock torch kernel is left in the step.                                                     */
typedef struct {
  int N, HW, C, cs;
  int mode;          /* CAT_NORM_INSTANCE | CAT_NORM_BATCH */
  float eps, momentum;
  int act;           /* CAT_ACT_* applied after the affine */
  float slope;
} cat_norm_t;
size_t cat_norm_ws_bytes(const cat_norm_t* g);
int cat_norm_fwd(const cat_norm_t* g, const float* x, const float* gamma, const float* beta, float* y,
                 float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                 int64_t* num_batches_tracked, void* ws, cat_stream_t stream);
/* dx, dgamma (+)=, dbeta (+)= from dy (gradient w.r.t. the activated output), the saved pre-norm input x and
 * the saved statistics.  gamma/dgamma/dbeta may be NULL (affine=False). */
int cat_norm_bwd(const cat_norm_t* g, const float* x, const float* dy, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_rstd, float* dx, float* dgamma, float* dbeta,
                 int accumulate, void* ws, cat_stream_t stream);
/* Inference-mode norm: y = act(x * scale[c] + shift[c]) (BatchNorm eval with running stats folded by
 * cat_bn_fold; used by the frozen teacher, distillers/base_inception_distiller.py:168). */
int cat_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                float eps, int C, float* scale, float* shift, cat_stream_t stream);
int cat_affine_act_fwd(const float* x, const float* scale, const float* shift, float* y, int64_t M, int C, int cs,
                       int act, float slope, cat_stream_t stream);

/* Elementwise: activations (fwd + bwd from the OUTPUT), n-ary add (the branch sum + residual of
 * InvertedResidualChannels.forward, inception_modules.py:233-236), channel concat (torch.cat(...,1),
 * distillers/base_inception_distiller.py:295-296) and its split backward. */
int cat_act_fwd(const float* x, float* y, int64_t n, int act, float slope, cat_stream_t stream);
int cat_act_bwd(const float* y, const float* dy, float* dx, int64_t n, int act, float slope, cat_stream_t stream);
int cat_add_n(const float* const* srcs, int nsrc, float* dst, int64_t n, cat_stream_t stream);
int cat_concat2(const float* a, int ca, int acs, const float* b, int cb, int bcs, float* y, int ycs, int64_t M,
                cat_stream_t stream);
int cat_slice_channels(const float* x, int xcs, int c0, int c, float* y, int ycs, int64_t M, cat_stream_t stream);

/* Layout transposes at the boundary (images / checkpoints are NCHW in the reference). */
int cat_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int ycs, cat_stream_t stream);
int cat_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int xcs, cat_stream_t stream);

/* Kernel alignment, utils/common.py:38-46.  X: [N][Dx], Y: [N][Dy] row-major (zero padding inside rows is
 * harmless).  ws >= cat_ka_ws_bytes(N).  out[0] = KA; the N x N Grams stay in ws for the backward.
 * cat_ka_bwd: dX = gout[0] * dKA/dX (same layout as X). */
size_t cat_ka_ws_bytes(int N);
int cat_ka_fwd(const float* X, int64_t Dx, const float* Y, int64_t Dy, int N, float* out, void* ws,
               cat_stream_t stream);
int cat_ka_bwd(const float* X, int64_t Dx, int N, const float* gout, const void* ws, float* dX, cat_stream_t stream);

/* Scalar losses with their gradients (models/modules/loss.py:52-99, nn.L1Loss / nn.MSELoss):
 *   kind 0: mean |a-b|      (L1, b tensor)          kind 1: mean (a-t)^2  (lsgan; t scalar target)
 *   kind 2: -mean min(a-1,0)  (hinge D real)        kind 3: -mean min(-a-1,0) (hinge D fake)
 *   kind 4: -mean a           (hinge G / wgangp real) kind 5: mean (a-b)^2 (MSE vs tensor)
 *   kind 6: mean BCE-with-logits(a, target) (vanilla) kind 7: mean a (wgangp fake)
 * Inputs are NHWC with (C, cs); the mean runs over M*C real elements.  out[0] = loss. */
size_t cat_loss_ws_bytes(int64_t M);
int cat_loss_fwd(int kind, const float* a, const float* b, float target, int64_t M, int C, int cs, float* out,
                 void* ws, cat_stream_t stream);
/* da = gout[0] * scale * dloss/da */
int cat_loss_bwd(int kind, const float* a, const float* b, float target, int64_t M, int C, int cs,
                 const float* gout, float scale, float* da, cat_stream_t stream);

/* Adam step over one flat buffer (torch.optim.Adam semantics, base_inception_distiller.py:205-214):
 * p, g, m, v: n floats; step = 1-based step count. */
int cat_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, cat_stream_t stream);
/* Same update with the optimiser scalars resident in HBM: hyper = float[8] {lr, beta1, beta2, eps, weight_decay, step,
 * (derived) lr/bc1, (derived) 1/sqrt(bc2)}.  The call first advances hyper[5] by one and refreshes the derived entries on
 * the device, so a captured hipGraph of the whole training step can be replayed without touching host state. */
int cat_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* hyper, float grad_scale,
                      cat_stream_t stream);
/* Measurement hook (bench.py `roofline`): while enabled, every entry point brackets what it enqueues with a HIP event
 * pair on the caller's stream, tagged with a kernel-family name and its algorithmic FLOPs.  cat_prof_collect()
 * synchronises and folds the records per family; cat_prof_family(i, ...) reads family i (count, total ms, total FLOPs). */
void cat_prof_enable(int on);
int cat_prof_collect(void);
int cat_prof_family(int i, char* name, int cap, int64_t* count, double* ms, double* flops);
double cat_prof_family_bytes(int i);

int cat_fill(float* p, int64_t n, float v, cat_stream_t stream);
int cat_axpy(float* y, const float* x, int64_t n, float a, cat_stream_t stream); /* y += a*x */

/* ------------------------------------------------------------------------------------------------------------------
 * GauGAN / SPADE distillation step (distillers/base_spade_distiller.py:226-234).
 *
 * Split-phase batch normalisation: replaces _SynchronizedBatchNorm.forward / _data_parallel_master / _compute_mean_std
 * (models/modules/sync_batchnorm/batchnorm.py:68-140) and F.batch_norm on one device (:69-72).  The host all-reduces
 * `sums` (2*cs floats: [sum x | sum x^2], or [sum g | sum g*xhat] in the backward) over RCCL between the calls; on one
 * GPU the calls simply follow each other.  a[c] = inv_std, b[c] = -mean*inv_std (xhat = x*a + b), zero on padding.
 * clamp = 1: inv_std = max(var, eps)^-1/2 (batchnorm.py:140, >1 replica); clamp = 0: (var+eps)^-1/2 (F.batch_norm). */
size_t cat_bn_ws_bytes(int64_t M, int cs);
int cat_bn_stats_fwd(const float* x, int64_t M, int C, int cs, float* sums, void* ws, cat_stream_t stream);
int cat_bn_finalize(const float* sums, double count, int C, int cs, float eps, int clamp, float momentum,
                    const float* gamma, const float* beta, float* mean, float* rstd, float* running_mean,
                    float* running_var, float* a, float* b, float* scale, float* shift, cat_stream_t stream);
/* y = act(x*scale + shift) is cat_affine_act_fwd.  Backward: g = dy * act'(gamma*xhat + beta). */
int cat_bn_stats_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* a,
                     const float* b, int64_t M, int C, int cs, int act, float slope, float* sums, void* ws,
                     cat_stream_t stream);
/* dx = gamma*inv_std*(g - sums[0]/count - xhat*sums[1]/count); dgamma (+)= local_sums[1], dbeta (+)= local_sums[0]. */
int cat_bn_apply_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* a,
                     const float* b, const float* sums, double count, const float* local_sums, float* dx,
                     float* dgamma, float* dbeta, int accumulate, int64_t M, int C, int cs, int act, float slope,
                     cat_stream_t stream);

/* InceptionSPADE.forward (models/modules/inception_modules.py:746-762) fused with the activation that
 * SPADEInvertedResidualChannels.forward applies to it (:553):  y = act((x*a + b) * (1 + gamma) + beta), where
 * gb: [M][gcs] holds gamma in channels [0,C) and beta in [C,2C) (the summed branch outputs, :756-759).
 * Backward pass 1 writes dgb (d gamma | d beta), dxh = g*(1+gamma) and sums = [sum dxh | sum dxh*xhat];
 * pass 2 (after the all-reduce of sums) turns dxh into dx in place (param_free_norm backward). */
int cat_spade_fwd(const float* x, const float* a, const float* b, const float* gb, float* y, int64_t M, int C,
                  int cs, int gcs, int act, float slope, cat_stream_t stream);
int cat_spade_bwd_stats(const float* x, const float* a, const float* b, const float* gb, const float* y,
                        const float* dy, float* dgb, float* dxh, float* sums, int64_t M, int C, int cs, int gcs,
                        int act, float slope, void* ws, cat_stream_t stream);
int cat_spade_bwd_apply(const float* x, const float* a, const float* b, const float* sums, double count, float* dx,
                        int64_t M, int C, int cs, cat_stream_t stream);

/* F.interpolate(mode='nearest') of the segmentation map (inception_modules.py:748, inception_spade_generator.py:67)
 * and nn.Upsample(scale_factor=2) (:45): src = min(floor(dst * in/out), in-1).  Backward for integer factor f. */
int cat_interp_nearest_fwd(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, int C, int xcs, int ycs,
                           cat_stream_t stream);
int cat_upsample_nearest_bwd(const float* dy, float* dx, int N, int Hi, int Wi, int f, int C, int cs,
                             cat_stream_t stream);
/* MultiscaleDiscriminator.downsample: F.avg_pool2d(3, stride 2, padding 1, count_include_pad=False)
 * (models/modules/discriminators.py:213-218).  Ho = (H-1)/2+1. */
int cat_avgpool3x3s2_fwd(const float* x, float* y, int N, int H, int W, int C, int cs, cat_stream_t stream);
int cat_avgpool3x3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, int cs, cat_stream_t stream);
/* nn.MaxPool2d(2,2) inside VGG19.features (models/modules/loss.py:151-186). */
int cat_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C, int cs, cat_stream_t stream);
int cat_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int cs,
                       cat_stream_t stream);
/* ---- evaluation path: the FID feature extractor InceptionV3, metric/inception.py:16-150, 177-300 over torchvision
 * 0.8.2's Inception3; called from metric/fid_score.py:152-216 (get_activations_from_ims).  Inference only. ---- */
/* nn.Conv2d with different zero padding along H (g->pad) and W (pad_w): the 1x7 / 7x1 / 1x3 / 3x1 factorised filters of InceptionC / D / E
 * (BasicConv2d, eval-mode BatchNorm folded into w / bias by the host, ReLU = g->act).  y may point at a channel slice of a wider NHWC
 * tensor (g->ycs = its pixel stride, g->ycw = g->Cout): torch.cat(outputs, 1) of an Inception block is never materialised separately. */
int cat_conv2d_fwd_rect(const cat_conv_t* g, int pad_w, const float* x, const float* w, const float* bias, float* y,
                        cat_stream_t stream);
/* nn.MaxPool2d(k, stride, pad) (mode 0: padding = -inf) and F.avg_pool2d(k, stride, pad, count_include_pad=False) (mode 1) on C4
 * channels (multiple of 4); x / y may be channel slices (xcs / ycs pixel strides). */
int cat_pool2d_fwd(const float* x, int xcs, int N, int H, int W, int C4, int k, int stride, int pad, int mode, float* y, int ycs,
                   int Ho, int Wo, cat_stream_t stream);
/* nn.AdaptiveAvgPool2d((1, 1)): y[n][c] = mean over the HW pixels; y row stride ycs. */
int cat_global_avgpool_fwd(const float* x, int xcs, int N, int HW, int C4, float* y, int ycs, cat_stream_t stream);
/* y = a * F.interpolate(x, (Ho, Wo), mode='bilinear', align_corners=False) + b (metric/inception.py:129-136: resize to 299 x 299 and the
 * 2 * x - 1 input normalisation in one pass); padding channels [C, round_up(C, 4)) are written as 0. */
int cat_resize_bilinear_fwd(const float* x, int xcs, int N, int H, int W, int C, float* y, int ycs, int Ho, int Wo, float a, float b,
                            cat_stream_t stream);
/* SPADEModel.preprocess_input + get_edges (models/spade_model.py:142-179): label -> one-hot over nc channels,
 * instance ids -> 4-neighbour edge map in channel nc (inst may be NULL = --no_instance).  y: [N][H][W][cs]. */
int cat_onehot_edges(const int* label, const int* inst, float* y, int N, int H, int W, int nc, int cs,
                     cat_stream_t stream);
/* torch.nn.utils.spectral_norm on the discriminator convs (spade_architecture/normalization.py:17-50): one power
 * iteration (power_iter=1, training) updating u [O] and v [I*taps, torch (i,kh,kw) order] in place, sigma = u.W v,
 * w_sn = w / sigma.  w, w_sn: [O][taps][wcs]; vp: [taps][wcs] scratch kept for the backward.
 * Backward: d weight_orig (+)= (gw - <gw, w_sn> u v^T) / sigma. */
size_t cat_spectral_norm_ws_bytes(int O, int I, int taps, int wcs);
int cat_spectral_norm_fwd(const float* w, int O, int I, int taps, int wcs, float* u, float* v, int power_iter,
                          float eps, float* sigma, float* w_sn, float* vp, void* ws, cat_stream_t stream);
int cat_spectral_norm_bwd(const float* gw, const float* w_sn, const float* u, const float* vp, const float* sigma,
                          int O, int taps, int wcs, float* dw, int accumulate, void* ws, cat_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * LDS-tile convolution with pre-packed filters, stride 1, k in {1,3,5} (csrc/conv_pk.hip):
 *   out[n][oy][ox][co] = act(bias[co] + sum_s sum_(ky,kx) sum_c f_s(src_s[n][oy-padv_s+ky][ox-padv_s+kx][c]) * W_s(tap,c,co))
 * One launch = a single nn.Conv2d (+ the nn.ReflectionPad2d in front of it), its input gradient, or the K-concatenated SUM of up to
 * 8 convolutions of different kernel sizes over different tensors: the branch sum of InvertedResidualChannels.forward
 * (models/modules/inception_modules.py:230-236: sum_k res_k(x) + sum_k dw_k(x)) written once, and in the backward pass the sum of the
 * branches' first-conv input gradients.  f_s = optional per-channel affine + activation applied while the source tile is staged
 * (the normalise + ReLU of a train-mode norm_layer, inception_modules.py:43-44, folded into the conv that consumes it).
 * Filters are consumed from a packed stream (cat_tconv_pack) in the exact order of the kernel's MFMA groups; trainable layers
 * re-pack once per optimizer step. */
#define CAT_TCONV_MAXSEG 8
typedef struct {
  const float* src;    /* [N][H][W][xcs], pointing at the segment's first channel (a channel slice of a wider buffer is fine) */
  const float* scale;  /* optional [c4] staging affine: v = act(src * scale + shift); NULL = none (act still applies) */
  const float* shift;
  int sstride;         /* floats between per-image rows of scale / shift (InstanceNorm statistics); 0 = one row for the batch */
  int xcs, c4;         /* pixel stride of src, channels of this segment (multiple of 4; padding channels must read as 0 after f_s) */
  int cin;             /* valid channels (<= c4): FLOP accounting only */
  int ks, padv;        /* ks x ks taps; output pixel (oy, ox) reads src pixel (oy - padv + ky, ox - padv + kx) */
  int act;             /* CAT_ACT_* applied while staging */
  float slope;
  int reflect;         /* source pixels outside the plane: 1 = mirrored (nn.ReflectionPad2d), 0 = zero */
  int pack_off;        /* float offset of this segment's packed filters inside `pack` */
} cat_tseg_t;
typedef struct {
  const float* res;    /* optional residual added AFTER the epilogue activation: y = act(acc + bias) + res (the block's skip connection) */
  float* stats;        /* optional per-tile statistics of the PRE-activation output for a following train-mode norm layer:
                          stats[(tile*2 + 0)*scs + co] = sum over the tile's valid pixels, [(tile*2 + 1)*scs + co] = sum of squared
                          deviations from the tile mean (combined exactly by cat_tnorm_finalize); tile = image * tiles_per_image + t,
                          8 x 16 pixel tiles in row-major order.  Forces the 8 x 16 tiling. */
  int rcs;             /* pixel stride of res */
  int scs;             /* floats per statistics row (>= the columns this launch owns; `stats` points at its first column) */
  int N, H, W;         /* source planes (all segments) */
  int Ho, Wo;          /* output plane */
  int Nn, ycs, ycw;    /* output channels, pixel stride, channels [Nn, ycw) are written as 0 */
  int nvalid;          /* output channels that are real (N-concatenated launches carry zero-filter padding columns): FLOP accounting only; 0 = Nn */
  int act;             /* epilogue activation */
  float slope;
  int nseg;
  cat_tseg_t seg[CAT_TCONV_MAXSEG];
} cat_tconv_t;
/* floats of packed filters for one segment (ks, c4) and Nn output channels */
size_t cat_tconv_pack_floats(int ks, int c4, int Nn);
/* mode 0 (forward):  W(tap, c, n) = w[n*wn + tap*wcs + c],  w = [Nn][ks*ks][wcs] conv weight, Ck = valid input channels
 * mode 1 (dgrad):    W(tap, c, n) = w[c*wn + (ks*ks-1-tap)*wcs + n], w = [Ck][ks*ks][wcs] conv weight (Ck = its Cout, Nn = its Cin) */
int cat_tconv_pack(const float* w, int mode, int Nn, int Ck, int ks, int wcs, int wn, int c4, float* dst, cat_stream_t stream);
int cat_tconv_fwd(const cat_tconv_t* g, const float* pack, const float* bias, float* y, cat_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K-concatenated SUM of stride-1 "same" convolutions (k in {1,3,5}, padding (k-1)/2, zero or reflect) over different tensors as ONE
 * implicit GEMM on 128 x 128 output tiles with both operands DMA-ed into LDS (csrc/conv_ksum.hip):
 *   y[n][oy][ox][co] = act(bias[co] + sum_s sum_(ky,kx) sum_c src_s[n][oy-pad_s+ky][ox-pad_s+kx][c] * w_s[co][ky][kx][c]) + res[n][oy][ox][co]
 * Replaces, for the eval-mode (frozen, BatchNorm-folded) teacher, the tail of InvertedResidualChannels.forward
 * (models/modules/inception_modules.py:230-236): the six second convs (the four 1 x 1 ones as one K = 176 segment), their sum, pw_bn and
 * the skip connection -- torch: 6 x F.conv2d + 5 x aten::add + F.batch_norm + aten::add.  Filters are read in the product's own layout
 * [Cout][k][k][wcs]; wide N (Cout >= 64) is what it is built for, the narrow layers stay on cat_tconv_fwd. */
#define CAT_KSUM_MAXSEG 4
typedef struct {
  const float* src;    /* [N][H][W][xcs], pointing at the segment's first channel */
  const float* w;      /* [Cout][ks][ks][wcs] */
  int xcs, c4;         /* pixel stride of src, channels read per tap (multiple of 4; channels [cin, c4) must hold zeros or meet zero filters) */
  int cin;             /* valid channels (<= c4): FLOP accounting only; 0 = c4 */
  int ks;              /* ks x ks taps, padding (ks - 1) / 2 */
  int reflect;         /* source pixels outside the plane: 1 = mirrored (nn.ReflectionPad2d), 0 = zero; one mode per launch */
  int wcs;             /* floats per (output channel, tap) of w (multiple of 4, >= c4) */
} cat_ksum_seg_t;
typedef struct {
  int N, H, W;         /* planes (input = output size) */
  int Cout, ycs, ycw;  /* output channels, pixel stride, channels [Cout, ycw) are written as 0 */
  int rcs;             /* pixel stride of res */
  int act;             /* epilogue activation (before the residual) */
  float slope;
  int nseg;
  cat_ksum_seg_t seg[CAT_KSUM_MAXSEG];
} cat_ksum_t;
int cat_conv2d_ksum_supported(const cat_ksum_t* g);
int cat_conv2d_ksum_fwd(const cat_ksum_t* g, const float* bias, const float* res, float* y, cat_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Train-mode norm layers of an InvertedResidualChannels block without their own passes (csrc/block_norm.hip): the producing
 * conv leaves per-tile statistics (cat_tconv_fwd `stats`, cat_dwm_fwd), cat_tnorm_finalize turns the table of ALL branches of a
 * block stage into scale / shift (+ running statistics of every branch's nn.BatchNorm2d, inception_modules.py:43-44,150-173), and
 * the consumer applies them while staging its input.  gamma / beta: the stage's affine parameters concatenated in the table's
 * channel order (cat_prep_run gathers them), NULL = no affine.  scale / shift / mean / rstd: [G][scs]; G = 1 (BatchNorm) or N
 * (InstanceNorm).  Biased variance, eps inside the sqrt, running_var gets the unbiased one (torch semantics). */
#define CAT_TNORM_MAXSLICE 8
typedef struct {
  int c0, c;                 /* channels [c0, c0 + c) of the table belong to this norm module */
  float* running_mean;       /* [c] or NULL */
  float* running_var;
  int64_t* num_batches;      /* nn.BatchNorm2d.num_batches_tracked or NULL */
} cat_nslice_t;
int cat_tnorm_finalize(const float* part, int scs, int G, int N, int Ho, int Wo, const float* gamma, const float* beta, int nslices,
                       const cat_nslice_t* slices, float eps, float momentum, float* scale, float* shift, float* mean, float* rstd,
                       int mstride, cat_stream_t stream);   /* mean / rstd rows are mstride floats apart (scs, or C for cat_norm_bwd) */
/* The same norms as SynchronizedBatchNorm2d over several ranks (models/modules/sync_batchnorm/batchnorm.py:103-140): cat_tnorm_sums folds this
 * rank's tile table (th x tw tiles, ncls entries per tile) to sums = [sum x | sum x^2] (2 * scs floats), the host all-reduces them over RCCL --
 * ONE collective per block stage instead of one per norm layer --, and cat_tnorm_finalize_sums applies the reference's multi-replica formula
 * (clamp = 1: inv_std = max(var, eps)^-1/2; running_var from the unbiased variance) to every norm module of the stage: scale / shift for the
 * consumers' staging, a = inv_std, b = -mean * inv_std for cat_bn_stats_bwd / cat_bn_apply_bwd.  count = pixels over ALL ranks. */
int cat_tnorm_sums(const float* part, int scs, int N, int Ho, int Wo, int th, int tw, int ncls, float* sums, cat_stream_t stream);
int cat_tnorm_finalize_sums(const float* sums, double count, int scs, const float* gamma, const float* beta, int nslices,
                            const cat_nslice_t* slices, float eps, float momentum, int clamp, float* scale, float* shift, float* a, float* b,
                            cat_stream_t stream);
/* cat_reflect_pad_bwd on channel slices: dxp / dx / add have their own pixel strides, C4 channels (multiple of 4) are folded and
 * `add` (optional) is added -- the skip connection's gradient joins the folded first-conv input gradient in the same pass. */
int cat_reflect_pad_bwd2(const float* dxp, int pcs, float* dx, int dcs, const float* add, int acs, int N, int H, int W, int C4, int pad,
                         cat_stream_t stream);
/* y = act(x * scale[g][c] + shift[g][c]) (+ res): the stand-alone apply of such a norm -- the block output
 * x + pw_bn(sum) (inception_modules.py:235-236) and the re-materialisation of a hidden activation in the backward pass.
 * x / y / res: [G][Pg][*cs], C4 channels (multiple of 4) are processed; sstride = floats between the groups' scale rows. */
int cat_affine_res_fwd(const float* x, int xcs, const float* scale, const float* shift, int sstride, const float* res, int rcs, float* y,
                       int ycs, int G, int Pg, int C4, int act, float slope, cat_stream_t stream);
/* All depthwise convs of a block (the groups=midp ConvBNReLU convs, inception_modules.py:166-173; k = 1 / 3 / 5 per channel quad) as one
 * launch over channel slices: input = first-stage pre-norm buffer with that stage's scale / shift + activation applied while staging,
 * output = concatenated pre-norm buffer + its per-tile statistics.  w25: [25][4*nq] filters embedded in a 5 x 5 frame (cat_prep_run). */
#define CAT_DWM_MAXQ 24      /* channel quads of the fused depthwise stage, forward (96 channels: 101 KB of LDS) */
#define CAT_DWM_MAXQ_BWD 18  /* ... and of cat_dwm_bwd (two patches in LDS: 72 channels = 142 KB -- the unpruned gamma|beta nets, 3 x 21 -> 72) */
typedef struct {
  int N, H, W;
  int nq;               /* channel quads */
  int xcs, ycs, scs;    /* pixel strides of x / y, floats per statistics row */
  int sstride;          /* floats between per-image scale rows (InstanceNorm), 0 = one group */
  int reflect, act;
  float slope;
  int ks[CAT_DWM_MAXQ];
} cat_dwm_t;
int cat_dwm_fwd(const cat_dwm_t* g, const float* x, const float* scale, const float* shift, const float* w25, const float* bias, float* y,
                float* stats, cat_stream_t stream);
/* Backward of cat_dwm_fwd for all depthwise convs of a block: da (pixel stride dacs) = gradient w.r.t. the (activated) input slice `a`
 * (pixel stride g->xcs), dz = gradient w.r.t. the pre-norm output (pixel stride g->ycs); reflect padding is folded in the kernel.  Filter
 * gradients go to dw[b] ([c[b]][ks[b]][ks[b]], the parameters' own layout) for the nbranch branches whose channels are [c0[b], c0[b] + c[b])
 * of the concatenation.  ws: cat_dwm_bwd_ws_bytes(g) bytes of scratch.  g->sstride / act / scs are ignored. */
size_t cat_dwm_bwd_ws_bytes(const cat_dwm_t* g);
int cat_dwm_bwd(const cat_dwm_t* g, const float* a, const float* dz, const float* w25, float* da, int dacs, int nbranch, const int* c0,
                const int* c, const int* ks, float* const* dw, int accumulate, void* ws, cat_stream_t stream);
/* Per-step preparation of a block's operands in ONE launch, driven by a job table resident in HBM (built once per network):
 *   kind 0  pack a conv weight for cat_tconv_fwd into columns [col0, col0 + Nn) of a stream with nt_total 16-wide N tiles (N-concatenated
 *           first convs; mode as cat_tconv_pack)
 *   kind 1  dst[i] = sum_k srcs[k][i], i < n     (concatenated gamma / beta / bias vectors; the summed bias of the branch sum)
 *   kind 2  depthwise filter [C][ks][ks] -> dst[tap25][cs] 5 x 5 frame, channels [col0, col0 + C)
 *   kind 3  srcs[1][i] (+)= srcs[0][i], i < n     (scatter a concatenated parameter gradient back; accumulate = cat_prep_run's flag)
 *   kind 4  srcs[1][r*wcs + i] (+)= srcs[0][r*wn + i], i < cs, r < n / cs   (columns of a K-concatenated weight gradient) */
#define CAT_PREP_MAXSRC 8
typedef struct {
  const float* srcs[CAT_PREP_MAXSRC];
  float* dst;
  int kind, nsrc, n;
  int mode, Nn, Ck, ks, wcs, wn, c4, nt_total, col0, cs;
  int block0, nblocks;   /* this job owns blocks [block0, block0 + nblocks) of 256 threads */
} cat_prep_job_t;
int cat_prep_run(const cat_prep_job_t* jobs_dev, int njobs, int total_blocks, int accumulate, cat_stream_t stream);

/* All first convs of a block in one launch (csrc/conv_pk.hip tstage1_kernel): the 5 x 5 conv (slot 0), the 3 x 3 conv (slot 1) and the
 * N-concatenated 1 x 1 convs (slot 2) of InvertedResidualChannels (inception_modules.py:135-147,150-165) share ONE staging of the
 * input tile per 16-channel chunk; each writes its channel slice [col0, col0 + width) of the pre-norm buffer y (no activation) and its
 * per-tile statistics (layout as cat_tconv_fwd `stats`).  packs[k]: the slot's filter stream (cat_prep_run kind 0 / cat_tconv_pack with
 * Nn = width); bias: concatenated over y's columns or NULL.  cat_tstage1_supported tells whether a kernel exists for the widths. */
typedef struct {
  int N, H, W, xcs, cin;
  int reflect;
  int ycs, scs;
  int col0[3], width[3], nvalid[3];
} cat_tstage1_t;
int cat_tstage1_supported(int w5, int w3, int w1);
int cat_tstage1_fwd(const cat_tstage1_t* g, const float* x, const float* const* packs, const float* bias, float* y, float* stats,
                    cat_stream_t stream);

/* Input gradients of the branches' SECOND convs in one launch (autograd of inception_modules.py:471-489 / 549-562, zero padding): the same
 * kernel reading dy -- slot 0 = the 5 x 5 residual branch, slot 1 = the 3 x 3 one, slot 2 = the N-concatenated 1 x 1 second convs of the
 * depthwise branches -- with transposed / flipped filter streams (cat_prep_run kind 0 mode 1); slot k writes columns
 * [col0[k], col0[k] + width[k]) of its OWN buffer dxs[k] with pixel stride dxcs[k] (the hidden-activation gradients live in two tensors).
 * g->ycs / scs / reflect are ignored (no statistics, no bias). */
int cat_tstage1_dgrad_supported(int w5, int w3, int w1);
/* Stage 1 of an EVAL-mode (frozen, BatchNorm-folded) InvertedResidualChannels block in one launch (round 6; cat_amd/frozen.py): the 5 x 5 and
 * 3 x 3 first convs of the residual branches and the N-concatenated 1 x 1 first convs of every other branch (inception_modules.py:135-165) from
 * one staging of the input, y_k = act(conv_k(x) + bias_k) into three buffers -- torch: 6 x F.conv2d + 6 x F.batch_norm + 6 x relu.
 * packs[k]: cat_tconv_pack streams of the (folded) filters for k = 5, 3, 1; the instantiation that exists serves the teacher's widths
 * (cat_tstage1w_supported: 42 / 42 / 176 output channels). */
typedef struct {
  int N, H, W;
  int xcs, cin;        /* pixel stride / channels of x */
  int reflect;         /* padding of the 5 x 5 / 3 x 3 convs: 1 = mirrored, 0 = zero */
  int act;             /* epilogue activation of all three slots */
  float slope;
  int ycs[3];          /* pixel strides of the three outputs (multiples of 4; channels [nvalid, ycs) are written as 0) */
  int nvalid[3];       /* output channels of the 5 x 5 / 3 x 3 / 1 x 1 slot */
} cat_tstage1w_t;
int cat_tstage1w_supported(int w5, int w3, int w1);
int cat_tstage1w_fwd(const cat_tstage1w_t* g, const float* x, const float* const* packs, const float* const* biases, float* const* ys,
                     cat_stream_t stream);
int cat_tstage1_dgrad(const cat_tstage1_t* g, const float* dy, const float* const* packs, float* const* dxs, const int* dxcs,
                      cat_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Quad-granule LDS-tile convolution (csrc/conv_q.hip, round 4): the same K-segment model as cat_tconv_fwd on v_mfma_f32_4x4x1_16B_f32 --
 * N granule 4 output channels, K granule 1 -- with a source stride and sub-pixel output classes, for the generator's ragged edge layers
 * (models/modules/inception_architecture/inception_generator.py:37-56: ReflectionPad2d(3) + Conv2d 7x7 (3 -> ngf), two Conv2d 3x3 stride 2;
 * :118-132: two ConvTranspose2d 3x3 stride 2 padding 1 output_padding 1, ReflectionPad2d(3) + Conv2d 7x7 (ngf -> 3) + Tanh) and any
 * stride-1 "same" convolution of a pruned student (inception_modules.py:135-147):
 *   out[n][cy*OS+py][cx*OS+px][co] = act(bias[co] + sum_s sum_(i<kh_s, j<kw_s) sum_c f_s(src_s[n][cy*S+oy_s+i][cx*S+ox_s+j][c]) * W_s(i*kw_s+j, c, co))
 * over the output LATTICE (cy < Ho, cx < Wo).  ncls = 1: OS = 1, the nseg segments are summed.  ncls = 4: OS = 2, seg[2*py+px] is the ONE
 * segment of sub-pixel class (py, px) -- nn.ConvTranspose2d(k3, s2, p1, op1) as four dense correlations of the coarse input grid with
 * 1 / 2 / 2 / 4 taps.  f_s: optional per-channel affine + activation applied while staging (the normalise + ReLU of the train-mode norm
 * layer in front, inception_generator.py:39-41).  stats: per-tile statistics of the pre-activation output for the norm layer behind
 * (layout as cat_tconv_fwd `stats` with the plan's tile height; entry = (image * tiles + tile) * ncls + class; merged by
 * cat_tnorm_finalize2).  Filters and the per-step LDS offsets come from a packed stream (cat_qconv_pack). */
#define CAT_QCONV_MAXSEG 8
typedef struct {
  const float* src;    /* [N][H][W][xcs] at the segment's first channel */
  const float* scale;  /* optional [c4] staging affine (per image if sstride != 0), NULL = none */
  const float* shift;
  int sstride;
  int xcs, c4, cin;    /* pixel stride, channels rounded up to 4 (padding channels must read as 0 after f_s), valid channels */
  int kh, kw;          /* tap rectangle */
  int oy, ox;          /* tap (i, j) of lattice pixel (cy, cx) reads source pixel (cy*S + oy + i, cx*S + ox + j); -pad for a padded conv */
  int act;             /* CAT_ACT_NONE / RELU / LRELU applied while staging */
  float slope;
  int reflect;         /* source pixels outside the plane: 1 = mirrored (nn.ReflectionPad2d), 0 = zero */
  int pack_off;        /* unused (the stream is laid out in program order by cat_qconv_pack) */
} cat_qseg_t;
typedef struct {
  const float* res;    /* optional residual added after the epilogue activation (ncls = 1 only) */
  float* stats;        /* optional per-tile statistics (see above); requires act = NONE, res = NULL */
  int rcs, scs;
  int N, H, W;         /* source planes (all segments) */
  int Ho, Wo;          /* output lattice; the output plane is (Ho*OS) x (Wo*OS) */
  int S, OS, ncls;     /* source stride 1 | 2; output stride / classes: (1, 1) or (2, 4) */
  int Nn, ycs, ycw;    /* output channels, pixel stride, channels [Nn, ycw) are written as 0 (ycw multiple of 4) */
  int nvalid;          /* FLOP accounting only; 0 = Nn */
  int act;             /* epilogue activation (any CAT_ACT_*) */
  float slope;
  int nseg;
  cat_qseg_t seg[CAT_QCONV_MAXSEG];
} cat_qconv_t;
typedef struct {
  int cs, nq, nsplit;  /* channels per staged chunk, output-channel quads per wave, N splits (waves x grid) */
  int th, tw;          /* output-lattice tile of one workgroup = one statistics entry per class */
  int tiles;           /* statistics entries per image (tiles x ncls) */
  int64_t pack_floats; /* floats of the launch's packed stream: [LDS offset table of every step][filters of every step], all segments */
} cat_qplan_t;
int cat_qconv_plan(const cat_qconv_t* g, cat_qplan_t* plan);
/* fewest 16 x 16 lattice tiles for which narrow outputs (<= 8 quads) take the 16 x 16 tiling (default 1024); v < 0 only queries.  Returns the
 * previous value.  Streams packed before a change stay valid only with the value they were packed under (tests lower it to run both tilings). */
int cat_qconv_min_tiles16(int v);
/* Write segment `seg`'s part of the packed stream `dst` (cat_qplan_t.pack_floats floats, zero-initialised once by the caller): its rows of the
 * offset table and its filters, W(t, c, co) = w[co*s_co + tapsrc[t]*s_tap + c*s_ci] for co < Nn_w (tapsrc NULL = identity).  The geometry must be
 * the one the stream is launched with (pointers inside g are not read).
 *   nn.Conv2d weight [Nn][kh*kw][wcs]:                 s_co = kh*kw*wcs, s_tap = wcs, s_ci = 1
 *   nn.ConvTranspose2d weight [Cin][3*3][wcs >= Nn]:   s_co = 1, s_tap = wcs, s_ci = 9*wcs, tapsrc = the class's (ky, kx) per tap */
int cat_qconv_pack(const cat_qconv_t* g, int seg, const float* w, float* dst, int Nn_w, const int* tapsrc, int s_co, int s_tap, int s_ci,
                   cat_stream_t stream);
int cat_qconv_fwd(const cat_qconv_t* g, const float* pack, const float* bias, float* y, cat_stream_t stream);
/* cat_tnorm_finalize for a statistics table of th x tw lattice tiles with ncls entries per tile (cat_qconv_fwd); Ho x Wo = the LATTICE, the
 * norm's pixel count per image is Ho * Wo * ncls.  (cat_tnorm_finalize = th 8, tw 16, ncls 1.) */
int cat_tnorm_finalize2(const float* part, int scs, int G, int N, int Ho, int Wo, int th, int tw, int ncls, const float* gamma,
                        const float* beta, int nslices, const cat_nslice_t* slices, float eps, float momentum, float* scale, float* shift,
                        float* mean, float* rstd, int mstride, cat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
