/* segland_hip.h -- C ABI of libsegland_hip.so: the MI355X (gfx950) kernels under SegLand's PSPNet-POP hot path.
 *
 * The reference (LiZhuoHong/SegLand) is pure Python/PyTorch: it has no FFI.  Its "plugin surface" for this path is
 * the nn.Module API of networks/pspnet_pop.py (kept by segland_amd/networks/pspnet_pop.py); this header is the new
 * boundary UNDER that module: one entry point per ATen op group the path executes.  Each entry cites
 * the reference call site(s) it replaces.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every device buffer is owned by the caller (PyTorch's allocator); the library
 *    never allocates or frees device memory and keeps no state besides its loaded code object.
 *  - Activations are NHWC ([B][H][W][C], C contiguous) in `dtype` (SL_F32 or SL_BF16); statistics, BN coefficients,
 *    logits, losses and weight gradients are float.  Weights are consumed in the GEMM layouts produced by
 *    sl_weight_prep (K-contiguous); master weights / gradients stay in PyTorch's OIHW float layout.
 *  - All launches are asynchronous on `stream` (a hipStream_t passed as void*); no hidden synchronisation.
 *  - Return value: 0 on success, negative SL_E* for a bad argument, positive hipError_t for a runtime failure.
 *    sl_last_error_string() describes the last failure on the calling thread.
 */
#ifndef SEGLAND_HIP_H
#define SEGLAND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL_F32 0
#define SL_BF16 1

#define SL_EINVAL (-1)      /* bad descriptor / unsupported shape */
#define SL_EWORKSPACE (-2)  /* workspace too small */

typedef void* sl_stream_t;

int sl_version(void);
const char* sl_last_error_string(void);
/* reads and resets the HIP runtime's sticky last error of the calling thread (returns its code): the step drivers call it after a failed graph capture */
int sl_hip_clear_error(void);

/* ------------------------------------------------------------------------------------------------ convolution
 * nn.Conv2d call sites: networks/backbones/resnet.py:44-49,109-110 (Bottleneck 1x1/3x3, stride 1/2, dilation 1/2/4),
 * networks/pspnet_pop.py:19,22,27 (PPM 3x3 4096->512, 1x1 +bias, stage 1x1), :47-51 (classifier 1x1 512->512).
 * Implicit GEMM on MFMA: rows = output pixels, cols = Cout, K = KH*KW*Cin.  The input may be a virtual channel
 * concat [x (C1 channels) | x2 (Cin-C1 channels)] (torch.cat at pspnet_pop.py:33-34 is never materialised).
 * Requirements: Cin, C1, Cin-C1 multiples of 64 (bf16) / 32 (f32); Cout multiple of 64. */
typedef struct SlConvDesc {
  int dtype;             /* SL_F32 | SL_BF16 : x, x2, w, y */
  int B, H, W;           /* input  [B][H][W][Cin]  */
  int Cin, Cout;
  int KH, KW, stride, pad, dil;
  int Ho, Wo;            /* output [B][Ho][Wo][Cout] */
  int C1;                /* channels read from x; the remaining Cin-C1 come from x2 (C1 == Cin: x2 unused) */
} SlConvDesc;

/* which tile kernel a shape is dispatched to (1000000*variant + 1000*BM + BN; variant 4 = 4-stage LDS ring, 2 = two-stage);
 * mode 0 = forward, 1 = data gradient.  Lets a profiler attribute launches to kernel names. */
int sl_conv2d_tile_config(const SlConvDesc* d, int mode);
/* the same for a launch with the given epilogue (bit set of SL_EPI_*): what sl_conv2d_fwd(stat_partial) / sl_conv2d_affine_fwd(_ex) / sl_conv2d_bwd_data(addend, bits) /
 * sl_conv2d_bwd_data_bnstat run on.  family 9 = conv_gemm_sk512_kernel, 8 = conv_gemm_p9_kernel (+ 10000000: split-K), 7 = conv_c64k3_kernel, 6 = conv_gemm_sk_kernel,
 * 5 = conv_gemm_p8_kernel, 4 = conv_gemm_ring_kernel (BM = 64: the few-tile form), 2 = conv_gemm_glds_kernel. */
#define SL_EPI_STATS 1
#define SL_EPI_AFFINE 2
#define SL_EPI_ADDEND 4
#define SL_EPI_ADDEND_BITS 8
#define SL_EPI_GATE 16
#define SL_EPI_SPLITK 32
#define SL_EPI_GELU 64      /* data gradient behind a GELU (sl_conv2d_bwd_data_gelu) */
int sl_conv2d_tile_config_ex(const SlConvDesc* d, int mode, int epi);
/* the same for the weight gradient: 1 = conv_wgrad_c64k3_kernel, 2 = conv_wgrad_c64p_kernel, 3 = conv_wgrad3_kernel (3x3 stride 1, nine taps per block), 10000000 + 1000*BN + BC = conv_wgrad_glds_kernel,
 * 20000000 + ... = conv_wgrad_kernel (+ 500000: rows are pixel pairs); every one is followed by its fixed-order slab reduce. */
int sl_conv2d_wgrad_config(const SlConvDesc* d);

/* number of row-blocks of the forward kernel == rows of the BN partial-statistics buffer */
int sl_conv2d_stat_rows(const SlConvDesc* d);

/* y = conv(x|x2, w) (+bias) (relu).  w: [Cout][KH][KW][Cin] dtype.  bias: float[Cout] or NULL.
 * stat_partial: NULL or float[stat_rows][2][Cout] receiving per-row-block (sum, sum of squares) of the fp32
 * accumulators per output channel (train-mode BatchNorm statistics, F.batch_norm at resnet.py:45-50). */
int sl_conv2d_fwd(const SlConvDesc* d, const void* x, const void* x2, const void* w, const float* bias, int relu,
                  void* y, float* stat_partial, sl_stream_t stream);

/* sl_conv2d_fwd with an extra [B][Ho][Wo][Cout] tensor (dtype) added to the accumulators BEFORE the statistics / bias:
 * the contribution of input channels that are handled outside this launch (factorised PPM priors, see sl_ppm_fact_*). */
int sl_conv2d_fwd_ex(const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend,
                     const float* bias, int relu, void* y, float* stat_partial, sl_stream_t stream);

/* nn.Linear (a 1x1 conv over an NHWC token map) with the elementwise tail of a Swin block in the epilogue (swintransformer.py:36,246-249):
 *   y = row_scale[b] * (x w^T + bias) + residual     row_scale: NULL or float[B] (DropPath), residual: NULL or [B][Ho][Wo][Cout]
 *   gelu_out (NULL or [B][Ho][Wo][Cout]) receives GELU(x w^T + bias) evaluated on the value y stores (Mlp fc1 + act). */
int sl_linear_fwd(const SlConvDesc* d, const void* x, const void* w, const float* bias, const float* row_scale, const void* residual,
                  void* y, void* gelu_out, sl_stream_t stream);

/* Inference / frozen-BN form: y = act(conv(x|x2, w) * scale[c] + shift[c] (+ residual)) in ONE kernel -- eval-mode
 * BatchNorm (running statistics), the shortcut add and the ReLU of resnet.py:60-76 folded into the conv epilogue. */
int sl_conv2d_affine_fwd(const SlConvDesc* d, const void* x, const void* x2, const void* w, const float* scale,
                         const float* shift, const void* residual, int relu, void* y, sl_stream_t stream);
/* The same with (a) `pre_addend` [M][Cout] added to the conv result BEFORE scale / shift (the factorised pyramid priors of pspnet_pop.py:31-35 in a frozen forward: the
 * prior half of the 3x3 bottleneck conv contracted on the pooled grids, see sl_ppm_fact_gather) and (b) an optional split-K workspace: layers with too few 256 x 256 tiles
 * for the chip (the fine-tune pair of ft_pop.py:233-269 is 8 192 pixel rows) are cut along K, the parts' fp32 tiles go through `workspace` and are summed in a fixed
 * order before the epilogue.  sl_conv2d_affine_fwd_workspace: bytes that would be used (0 = the layer is not split); workspace may be NULL (no split). */
size_t sl_conv2d_affine_fwd_workspace(const SlConvDesc* d);
int sl_conv2d_affine_fwd_ex(const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend, const float* scale,
                            const float* shift, const void* residual, int relu, void* y, void* workspace, size_t workspace_bytes, sl_stream_t stream);

/* dx = conv_transpose(dy, w) (+addend [* relu bit of addend_mask]) (masked by mask_src > 0).
 * wt: [Cin][KH][KW][Cout] dtype (sl_weight_prep).  dx: [B][H][W][Cin] (all Cin channels, also for a virtual concat).
 * addend / mask_src: NULL or [B][H][W][Cin]; addend_mask: NULL or the relu_mask bytes (sl_bn_act_fwd) of that tensor --
 * the shortcut gradient `dout * relu'(out)` of a bottleneck is applied here instead of being materialised. */
int sl_conv2d_bwd_data(const SlConvDesc* d, const void* dy, const void* wt, const void* addend, const uint8_t* addend_mask,
                       const void* mask_src, void* dx, sl_stream_t stream);

/* dx = conv_transpose(dy, w) * GELU'(h): the data gradient of an Mlp's fc2 lands behind the activation (networks/backbones/swintransformer.py:26-31, x = fc2(act(fc1(x)))
 * backward) in one launch; h: [B][H][W][Cin], the stored pre-activation.  Same result as sl_conv2d_bwd_data followed by sl_gelu_bwd (the epilogue works on the rounded
 * data gradient), one write and two reads of the block's widest tensor less. */
int sl_conv2d_bwd_data_gelu(const SlConvDesc* d, const void* dy, const void* wt, const void* h, void* dx, sl_stream_t stream);

/* Data gradient + the reduce pass of the BatchNorm backward below it, in one kernel (resnet.py:57-78 backward: conv3 <- bn2/relu, conv2 <- bn1/relu).
 * The result is the gradient wrt a = relu(bn(c)); the epilogue gates it with the ReLU bits of `gate` (1 byte per 16-byte vector of dx), stores the gated
 * gradient g in dx and writes per-row-block column sums (sum g, sum g * (bn_x - mean) * invstd) to stat_partial [rows][2][Cin] -- the partials that
 * sl_bn_bwd_reduce would produce in a separate pass over g and bn_x; sl_bn_bwd_finalize / sl_bn_bwd_apply(relu_mask = NULL) consume them unchanged.
 * sl_conv2d_bwd_data_bnstat_rows: rows of stat_partial, or 0 when the shape is not served (then: sl_conv2d_bwd_data + sl_bn_bwd_reduce). */
int sl_conv2d_bwd_data_bnstat_rows(const SlConvDesc* d);
int sl_conv2d_bwd_data_bnstat(const SlConvDesc* d, const void* dy, const void* wt, const uint8_t* gate, const void* bn_x, const float* bn_mean,
                              const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream);

/* The same across a block boundary: dx = data gradient + `addend` (the shortcut gradient, already gated) is the gradient wrt the PREVIOUS bottleneck's output
 * relu(bn3(c3) + res); the epilogue gates it with that ReLU's bits (`gate`) and reduces it against bn_x = c3: the previous block's bn3 backward needs no reduce
 * pass and receives its gradient gated.  Shapes of the pixel-stationary kernel only (1x1 stride 1, Cout = 64 / 128 / 256, Cin % 128 == 0, Cin <= 1024,
 * B*H*W % 256 == 0, bf16); sl_conv2d_bwd_data_addend_bnstat_rows = rows of stat_partial, 0 = not served. */
int sl_conv2d_bwd_data_addend_bnstat_rows(const SlConvDesc* d);
int sl_conv2d_bwd_data_addend_bnstat(const SlConvDesc* d, const void* dy, const void* wt, const void* addend, const uint8_t* gate, const void* bn_x,
                                     const float* bn_mean, const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream);
/* Data gradient + a half-resolution addend at the even positions: addend_half [B][H/2][W/2][Cin] is the dense data gradient of a 1x1 stride-2 conv on its own output
 * grid (the downsample branch of a stride-2 stage entry, resnet.py:109-110); its zero-filled full-resolution form is never written.  gate .. stat_partial: all NULL, or
 * the cross-block statistics of sl_conv2d_bwd_data_addend_bnstat.  Served where sl_conv2d_bwd_data_addend_half_ok(d) != 0 (pixel-stationary kernel, even H and W). */
int sl_conv2d_bwd_data_addend_half_ok(const SlConvDesc* d);
int sl_conv2d_bwd_data_addend_half(const SlConvDesc* d, const void* dy, const void* wt, const void* addend_half, const uint8_t* gate, const void* bn_x,
                                   const float* bn_mean, const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream);
/* BatchNorm-backward APPLY pass folded into the gradients of the 1x1 conv in front of it (resnet.py:66-70 backward, conv3 -> bn3 of an identity bottleneck whose incoming
 * gradient g arrived gated and reduced): with dc = cA g + cB (c - mean) + cC (sl_bn_bwd_finalize) and c = x W^T,
 *   dx = [g | x] wt_ext^T + bias,  wt_ext [Cin][Cout + Cin] = [diag(cA) W ; W^T diag(cB) W]^T, bias = (cC - cB mean) W = -[mean(g) | mean(x)] wt_ext^T      (sl_bn_fold_weights; w_fwd [Cout][Cin] /
 *        w_bwd [Cin][Cout] are the layer's prepared bf16 weights), gated with the ReLU bits of x's own BatchNorm + its column sums as in sl_conv2d_bwd_data_bnstat;
 *   dW = diag(cA) (g^T x) + diag(cB) W (x^T x) + (cC - cB mean) (x) colsum(x)      (sl_bn_fold_wgrad; g^T x, x^T x [Cin][Cin] and colsum(x) from ONE launch of
 *        sl_conv2d_bwd_weight_dy2).
 * The pass over (g, c) and the tensor dc never exist.  Served (rows > 0): bf16 1x1 stride-1 layers with Cin % 256 == 0 on whole 256-row tiles (the half-tile kernel). */
int sl_conv2d_bwd_data_bnstat_folded_rows(const SlConvDesc* d);
int sl_conv2d_bwd_data_bnstat_folded(const SlConvDesc* d, const void* g, const void* x, const void* wt_ext, const float* bias, const uint8_t* gate, const void* bn_x,
                                     const float* bn_mean, const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream);
int sl_bn_fold_weights(int Cout, int Cin, const void* w_fwd, const void* w_bwd, const float* cA, const float* cB, const float* g_colsum, const float* x_colsum,
                       long long rows, void* wt_ext, float* bias, sl_stream_t stream);       /* g_colsum = dbeta of the BatchNorm, x_colsum = colsum(x): the bias is formed from the two means */
int sl_bn_fold_wgrad(int Cout, int Cin, const float* gtx, float* dw, const float* xtx, const float* x_colsum, const void* w_fwd, const float* cA, const float* cB,
                     const float* cC, const float* mean, sl_stream_t stream);       /* gtx = g^T x [Cout][Cin] (may be dw itself) */
/* [dy1 | dy2]^T x in one launch: dw [d->Cout][d->Cin], d->Cout = Cout1 + the channels of dy2 (g^T x and x^T x of the fold above from one pass over x); colsum_partial
 * [sl_conv2d_bwd_weight_bias_rows(d, 0, 0)][d->Cout] or NULL.  1x1 stride-1 bf16 layers on the LDS-DMA tile kernel, Cout1 a multiple of its row tile. */
int sl_conv2d_bwd_weight_dy2(const SlConvDesc* d, const void* x, const void* dy1, const void* dy2, int Cout1, float* dw, void* workspace, size_t workspace_bytes,
                             float* colsum_partial, sl_stream_t stream);

/* The dual form: the previous block is the FIRST bottleneck of a stage, whose output ReLU sits behind bn3 AND the downsample BatchNorm (resnet.py:71-76): the gated
 * gradient is reduced against both BatchNorm inputs in one store loop.  stat_partial / stat_partial2: [rows][2][Cin] each, (sum g, sum g * xhat) per BatchNorm
 * (what sl_bn_bwd_reduce2 would produce in a pass of its own over g, bn_x and bn_x2).  Same shapes as above. */
int sl_conv2d_bwd_data_addend_bnstat2(const SlConvDesc* d, const void* dy, const void* wt, const void* addend, const uint8_t* gate, const void* bn_x,
                                      const float* bn_mean, const float* bn_invstd, const void* bn_x2, const float* bn_mean2, const float* bn_invstd2, void* dx,
                                      float* stat_partial, float* stat_partial2, sl_stream_t stream);

/* dw (float, OIHW [Cout][Cin][KH][KW]) = sum over pixels of dy (x) x.  Deterministic split-K through `workspace`. */
size_t sl_conv2d_bwd_weight_workspace(const SlConvDesc* d);
int sl_conv2d_bwd_weight(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw,
                         void* workspace, size_t workspace_bytes, sl_stream_t stream);

/* the same, writing into a WIDER gradient tensor dw [Cout][dw_cin_total][KH][KW] at input-channel offset dw_ci_off */
int sl_conv2d_bwd_weight_ex(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total,
                            int dw_ci_off, void* workspace, size_t workspace_bytes, sl_stream_t stream);
/* Weight gradient and the bias gradient's column-sum partials of one nn.Linear / biased conv (swintransformer.py:40-52 Mlp, :95-98 qkv / proj: the autograd of F.linear
 * yields dW = dy^T x and db = sum_rows dy).  colsum_partial: float [sl_conv2d_bwd_weight_bias_rows(d, n_valid, c_valid)][Cout]; finalize with sl_colsum_finalize(_multi).
 * 1x1 layers on the LDS-DMA tile kernel form the column sums inside the weight-gradient kernel (its dy fragments times an all-ones fragment: one partial row per split of
 * the pixel range, no second pass over dy); other shapes run sl_colsum_rows_partial after the weight gradient (sl_colsum_rows_blocks rows). */
int sl_conv2d_bwd_weight_bias_rows(const SlConvDesc* d, int n_valid, int c_valid);      /* n_valid / c_valid: as sl_conv2d_bwd_weight_clip, 0 = all */
int sl_conv2d_bwd_weight_bias(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                              float* colsum_partial, sl_stream_t stream);
/* The same for layers computed at zero-padded channel counts (Swin-T/S carry C = 96 at pitch 128, section 10 of DESIGN.md): the GEMM runs at d->Cout x d->Cin, dw is the
 * PARAMETER's shape [n_valid][c_valid][KH][KW] -- the slab reduce writes only the channels that exist (no slicing copy behind it).  colsum_partial may be NULL. */
int sl_conv2d_bwd_weight_clip(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int n_valid, int c_valid, void* workspace,
                              size_t workspace_bytes, float* colsum_partial, sl_stream_t stream);
/* The slab reduces of several layers in ONE launch (a transformer block's four nn.Linear weight gradients, swintransformer.py:195-250 backward): sl_conv2d_bwd_weight_defer
 * is sl_conv2d_bwd_weight_clip (n_valid / c_valid 0 = all; colsum_partial may be NULL) that runs the split-K kernel and leaves its flat slab reduce -- and the bias
 * column sums that would ride in it -- in *item; sl_wgrad_reduce_multi runs up to SL_WGRAD_BATCH_MAX items at once (same summation order, same bits as the single
 * launches).  item->splits == 0: the shape's path reduced by itself, nothing is deferred.  Every deferred layer needs a workspace of its own until the multi launch. */
#define SL_WGRAD_BATCH_MAX 8
typedef struct SlWgradReduce {
  const float* ws; float* dw; long long total; int splits, Cin, dw_cin_total, dw_ci_off, n_valid, c_valid, dtype, Cout, ncol;
  const void* dy; long long rows, rows_per_block; float* colsum_part;
} SlWgradReduce;
int sl_conv2d_bwd_weight_defer(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int n_valid, int c_valid, void* workspace,
                               size_t workspace_bytes, float* colsum_partial, SlWgradReduce* item, sl_stream_t stream);
int sl_wgrad_reduce_multi(const SlWgradReduce* items, int n, sl_stream_t stream);
/* OIHW float master weight -> w_fwd [Cout][KH][KW][Cin] and/or w_bwd [Cin][KH][KW][Cout] in dtype (either may be NULL) */
int sl_weight_prep(int dtype, const float* w_oihw, int Cout, int Cin, int KH, int KW, void* w_fwd, void* w_bwd,
                   sl_stream_t stream);

/* GEMM layouts of the input-channel slice [ci_off, ci_off+ci_cnt) of an OIHW weight with CinTot input channels */
int sl_weight_prep_slice(int dtype, const float* w_oihw, int Cout, int CinTot, int ci_off, int ci_cnt, int KH, int KW,
                         void* w_fwd, void* w_bwd, sl_stream_t stream);

/* The same for n weights in ONE launch.  table_dev: device array of n 48-byte records
 *   { const float* src; void* w_fwd; void* w_bwd; int O, I, KHW, dtype; long long start; }
 * with `start` the running count of 64 x 32 channel tiles (O*I/2048) of the preceding records and total_tiles their grand total;
 * O % 64 == 0, I % 32 == 0, KHW <= 9. */
int sl_weight_prep_batched(const void* table_dev, int n, long long total_tiles, sl_stream_t stream);

/* ------------------------------------------------------------------------------------------------ batch norm
 * nn.BatchNorm2d call sites: resnet.py:45,48,50,88,111; pspnet_pop.py:20,28 (eps 1e-5, momentum 0.1). */

/* Train mode: reduce the conv kernel's partials to mean / biased var, update running stats (unbiased var),
 * emit mean, invstd and the affine pair scale = gamma*invstd, shift = beta - mean*scale. */
int sl_bn_finalize_train(const float* stat_partial, int stat_rows, int C, long long count, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* mean, float* invstd, float* scale, float* shift, sl_stream_t stream);
/* The same behind a BIASED conv (swin_pop.py:112-131: Conv2d(bias=True) + BatchNorm2d): stat_partial are the statistics of the raw conv output, the bias
 * (bias_n <= C entries, the rest of a padded channel pitch has none) shifts only the mean that enters running_mean; mean / invstd / scale / shift stay those of
 * the raw output (the bias cancels in the normalised result).  Replaces a separate running_mean += momentum * bias pass. */
int sl_bn_finalize_train_bias(const float* stat_partial, int stat_rows, int C, long long count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              float* mean, float* invstd, float* scale, float* shift, const float* conv_bias, int bias_n, sl_stream_t stream);
/* Eval mode: scale/shift (and mean/invstd) from the running statistics. */
int sl_bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* mean, float* invstd, float* scale, float* shift,
                        sl_stream_t stream);
/* y = x*scale[c] + shift[c] (+ residual) (relu)   -- BN affine + `out += residual` + ReLU of resnet.py:60-76.
 * relu_mask (nullable): one byte per 16-byte vector of y, bit k = (y_k > 0): the backward reads this instead of y. */
int sl_bn_act_fwd(int dtype, const void* x, const float* scale, const float* shift, const void* residual, int relu,
                  void* y, uint8_t* relu_mask, long long rows, int C, sl_stream_t stream);
/* Backward, step 1: partial[blk][0][c] = sum g, partial[blk][1][c] = sum g*xhat with g = dy * relu'(y) (mask from
 * relu_mask if given, else from y > 0 if y is given, else none) and
 * xhat = (x - mean)*invstd.  Returns the number of partial rows through *nblk (buffer: float[nblk][2][C]). */
int sl_bn_bwd_reduce_rows(long long rows, int C);
int sl_bn_bwd_reduce(int dtype, const void* dy, const void* y, const uint8_t* relu_mask, const void* x, const float* mean,
                     const float* invstd, float* partial, long long rows, int C, sl_stream_t stream);
/* step 2: dgamma = S2, dbeta = S1 and the per-channel coefficients of dx = cA*g + cB*(x-mean) + cC.
 * train != 0: batch-statistics backward; train == 0: frozen statistics (dx = scale*g). */
int sl_bn_bwd_finalize(const float* partial, int nblk, int C, long long count, const float* gamma, const float* mean,
                       const float* invstd, int train, float* dgamma, float* dbeta, float* cA, float* cB, float* cC,
                       sl_stream_t stream);
/* step 3: dx = cA*g + cB*(x-mean) + cC; if dres != NULL also dres = g (the residual-branch gradient). */
int sl_bn_bwd_apply(int dtype, const void* dy, const void* y, const uint8_t* relu_mask, const void* x, const float* cA,
                    const float* cB, const float* cC, const float* mean, void* dx, void* dres, long long rows, int C,
                    sl_stream_t stream);

/* Two BatchNorms whose outputs were added before ONE ReLU -- bn3 and the downsample BN of the first bottleneck of a stage (resnet.py:71-76) -- share the gated
 * gradient g = dy * relu'(out): one sweep over dy and the ReLU bits serves both reduce passes (partial1 / partial2, each [rows][2][C] like sl_bn_bwd_reduce)
 * and both apply passes (dx1 / dx2).  The largest tensor of the block is read twice instead of four times. */
int sl_bn_bwd_reduce2(int dtype, const void* dy, const uint8_t* relu_mask, const void* x1, const float* mean1, const float* invstd1, float* partial1,
                      const void* x2, const float* mean2, const float* invstd2, float* partial2, long long rows, int C, sl_stream_t stream);
int sl_bn_bwd_apply2(int dtype, const void* dy, const uint8_t* relu_mask, const void* x1, const float* cA1, const float* cB1, const float* cC1,
                     const float* mean1, void* dx1, const void* x2, const float* cA2, const float* cB2, const float* cC2, const float* mean2, void* dx2,
                     long long rows, int C, sl_stream_t stream);

/* ------------------------------------------------------------------------------------------------ stem
 * resnet.py:86-90,124-125: conv 7x7 s2 p3 3->64 on the NCHW float image, BN, ReLU, maxpool 3x3 s2 p1. */
int sl_stem_conv_stat_rows(int B, int H, int W);
size_t sl_stem_conv_fwd_workspace(int dtype);                                 /* bytes of caller-provided scratch (bf16: the weights in MFMA fragment order) */
int sl_stem_conv_fwd(int dtype, const float* img_nchw, const float* w_oihw, void* y, float* stat_partial, int B, int H,
                     int W, void* workspace, sl_stream_t stream);             /* y: [B][H/2][W/2][64] */
/* pooled: [B][Hc/2][Wc/2][64]; argmax (nullable): uint8 window position ky*3+kx of the FIRST maximum in ATen's
 * scan order, same shape as pooled (needed by the backward). */
int sl_stem_bn_relu_pool_fwd(int dtype, const void* c0, const float* scale, const float* shift, void* pooled,
                             uint8_t* argmax, int B, int Hc, int Wc, sl_stream_t stream);
/* g0 = maxpool_bwd(dpooled) masked by relu'(bn(c0)) */
int sl_stem_pool_relu_bwd(int dtype, const void* dpooled, const uint8_t* argmax, const void* c0, const float* scale,
                          const float* shift, void* g0, int B, int Hc, int Wc, sl_stream_t stream);
/* The same + the reduce pass of bn1's backward (resnet.py:124-125 backward) in the same sweep: stat_partial [sl_stem_pool_relu_bwd_bnstat_rows][2][64] receives the
 * per-block column sums (sum g0, sum g0 * (c0 - mean) * invstd) of the stored gradient; sl_bn_bwd_finalize / sl_bn_bwd_apply(relu_mask = NULL, y = NULL) consume them. */
int sl_stem_pool_relu_bwd_bnstat_rows(int B, int Hc, int Wc);
int sl_stem_pool_relu_bwd_bnstat(int dtype, const void* dpooled, const uint8_t* argmax, const void* c0, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, void* g0, float* stat_partial, int B, int Hc, int Wc, sl_stream_t stream);
/* im2col of the 7x7 s2 p3 receptive fields: col [B*H/2*W/2][192] dtype, column t = c*49 + ky*7 + kx (t >= 147: zero).
 * The stem weight gradient is then sl_conv2d_bwd_weight on (col as a 192-channel 1x1 input, dc0) -- an MFMA reduction. */
int sl_stem_im2col(int dtype, const float* img_nchw, void* col, int B, int H, int W, sl_stream_t stream);
/* direct (non-MFMA) variant of the same gradient, kept as a cross-check */
size_t sl_stem_conv_bwd_weight_workspace(int B, int H, int W);
int sl_stem_conv_bwd_weight(int dtype, const float* img_nchw, const void* dc0, float* dw_oihw, void* workspace,
                            size_t workspace_bytes, int B, int H, int W, sl_stream_t stream);

/* ------------------------------------------------------------------------------------------------ pyramid pooling
 * pspnet_pop.py:26 AdaptiveAvgPool2d(1,2,3,6) (bins [floor(i*H/s), ceil((i+1)*H/s))) and :33 bilinear
 * (align_corners=False) upsampling of the four stage outputs.  Level l has s_l*s_l cells; the pooled / stage tensors
 * are stored level after level as rows [sum_l B*s_l^2][C], level-major then (b, i, j), and are ALWAYS float
 * (pooled, dpooled, stage, dstage): train-mode BN over B*s^2 samples must not see bf16-rounded inputs. */
typedef struct SlPpmDesc {
  int dtype;
  int B, H, W, C;        /* feature map x4 [B][H][W][C] */
  int nlevels;           /* <= 4 */
  int sizes[4];
} SlPpmDesc;
size_t sl_ppm_workspace(const SlPpmDesc* d);
int sl_ppm_pool_fwd(const SlPpmDesc* d, const void* x, float* pooled, void* workspace, size_t workspace_bytes,
                    sl_stream_t stream);
/* dx[b,y,x,c] = dcat[b,y,x, cat_off + c] + sum over levels/bins covering (y,x) of dpooled/bin_area.
 * dcat has row pitch cat_pitch channels (the dgrad of the virtual concat). */
int sl_ppm_pool_bwd(const SlPpmDesc* d, const float* dpooled, const void* dcat, int cat_pitch, int cat_off, void* dx,
                    void* workspace, size_t workspace_bytes, sl_stream_t stream);   /* workspace: sl_ppm_workspace(d) */
/* priors[b,y,x, l*Cs + c] = bilinear(stage_l)[b,y,x,c]; stage rows as above with Cs channels */
int sl_ppm_upsample_fwd(const SlPpmDesc* d, int Cs, const float* stage, void* priors, sl_stream_t stream);
int sl_ppm_upsample_bwd(const SlPpmDesc* d, int Cs, const void* dcat, int cat_pitch, float* dstage, void* workspace,
                        size_t workspace_bytes, sl_stream_t stream);
/* BatchNorm + ReLU backward of ALL pyramid stages (pspnet_pop.py:12-16 backward: four nn.BatchNorm2d over B*s*s samples each) in one launch.  dy / y / x / dx:
 * float [rows][C] in the row order above (dy: gradient wrt relu(bn(x)), y = relu(bn(x)) as the gate, x: the stage conv's output); the arrays have d->nlevels entries:
 * mean / invstd / gamma (gamma[l] may be NULL = 1), train[l] (0: running statistics, dx = gamma * invstd * g), dgamma[l] / dbeta[l] (float [C] each, may be NULL). */
#define SL_PPM_MAX_LEVELS 4
int sl_ppm_stage_bn_bwd(const SlPpmDesc* d, int C, const float* dy, const float* y, const float* x, const float* const* mean, const float* const* invstd,
                        const float* const* gamma, const int* train, float* const* dgamma, float* const* dbeta, float* dx, sl_stream_t stream);

/* Grouped skinny 1x1 convolution over the pyramid rows (the four stage convs pspnet_pop.py:12-16 in one launch, and the per-level
 * GEMMs of the factorised prior path):  y[r][n] = sum_k x[r][k] * w[l(r)][n][k]  with x [rows][K], w [nlevels][N][K], y [rows][N], all
 * float, rows ordered as in sl_ppm_pool_fwd; d->C is ignored.  K % 32 == 0, N % 64 == 0.  stat_partial (optional):
 * float [sl_ppm_rows_gemm_stat_rows][2][N], per level ceil(B*s*s/128) consecutive groups of (sum y, sum y^2) for the BN statistics. */
int sl_ppm_rows_gemm_stat_rows(const SlPpmDesc* d);
size_t sl_ppm_rows_gemm_workspace(const SlPpmDesc* d, int K, int N);
int sl_ppm_rows_gemm(const SlPpmDesc* d, int K, int N, const float* x, const float* w, float* y, float* stat_partial,
                     void* workspace, size_t workspace_bytes, sl_stream_t stream);
/* The same with one [N][K] weight tensor per level (w_levels: d->nlevels pointers): the stage convs' own prepared weight copies, no stacked copy per step. */
int sl_ppm_rows_gemm_levels(const SlPpmDesc* d, int K, int N, const float* x, const float* const* w_levels, float* y, float* stat_partial,
                            void* workspace, size_t workspace_bytes, sl_stream_t stream);
/* Weight gradients of those grouped GEMMs, all levels in one launch (the backward of pspnet_pop.py:12-16 wrt the stage conv weights, and of the per-level
 * GEMMs of the factorised prior path):  dw[l][n][k] = sum over the rows r of level l of a[r][n] * x[r][k]  with a [rows][N] (the gradient wrt y), x [rows][K], all
 * float; dw: d->nlevels pointers to float [N][K] tensors (the parameters' own gradient buffers).  N % 64 == 0, K % 64 == 0.  Fixed summation order, no workspace. */
int sl_ppm_rows_wgrad(const SlPpmDesc* d, int N, int K, const float* a, const float* x, float* const* dw, sl_stream_t stream);

/* Factorised prior half of the PPM bottleneck 3x3 conv (pspnet_pop.py:19 applied to the concat of :33-34).  The 3x3 conv of a
 * bilinearly upsampled s x s map is  sum_tap sum_cells u_tap(pixel; cell) * Q[cell][(tap, n)]  with  Q = (1x1 conv of the stage
 * map with the tap's weight slice): exact by linearity, and it removes half of that conv's FLOPs (77 GFLOP/tile forward).
 *   sl_ppm_wq_prep     : W_oihw [N][Ctot][3][3] -> per level l  wq_f [9N][Cs] / wq_b [Cs][9N]  (float; channels l*Cs..)
 *   sl_ppm_fact_gather : gout[b,y,x,n] (dtype) = sum over levels/taps/cells of u * q   (q: float [rows][9N], rows as above)
 *   sl_ppm_fact_scatter: gq[rows][9N] (float) = transpose of the gather applied to dcb [B][H][W][N] (dtype)
 *   sl_ppm_dwq_scatter : dwq [l][(tap,n)][c] -> dw_oihw[n][l*Cs + c][tap] */
int sl_ppm_wq_prep(const float* w_oihw, int N, int Ctot, int Cs, int nlevels, float* wq_f, float* wq_b, sl_stream_t stream);
int sl_ppm_dwq_scatter(const float* dwq, int N, int Ctot, int Cs, int nlevels, float* dw_oihw, sl_stream_t stream);
size_t sl_ppm_fact_workspace(const SlPpmDesc* d, int N);
int sl_ppm_fact_gather(const SlPpmDesc* d, int N, const float* q, void* gout, void* workspace, size_t workspace_bytes,
                       sl_stream_t stream);
int sl_ppm_fact_scatter(const SlPpmDesc* d, int N, const void* dcb, float* gq, void* workspace, size_t workspace_bytes,
                        sl_stream_t stream);

/* ------------------------------------------------------------------------------------------------ POP head
 * pspnet_pop.py:95-121 orthogonal_decompose + :178-182 / :210-219 classifier on the components, evaluated in the
 * exactly equivalent collapsed form: for a class k with unit prototype s_k and p = s_k.q,
 * MLP(p*s_k) = max(p,0)*MLP(s_k) + max(-p,0)*MLP(-s_k) because the MLP is bias-free and ReLU is positively
 * homogeneous; only the residual (background) feature needs the per-pixel MLP. */
/* proj[r][k] = S[k].q[r];  bg[r] = q[r] - sum_k proj[r][k] S[k].   S: float[Kt][C], already L2-normalised. */
int sl_pop_decompose_fwd(int dtype, const void* feats, const float* S, int Kt, float* proj, void* bg, long long R,
                         int C, sl_stream_t stream);
/* rows of +S and -S appended to an MLP input matrix: dst[2*Kt][C] = [S ; -S] in dtype */
/* Prototype preparation (pspnet_pop.py:96-99,170-171,185-186,236-239 + criterion.py:37-43) in one launch: Sa / Sb = L2-normalised rows of Ea [Ka][C] / Eb [Kb][C]
 * (eps 1e-12), G [Ka][Ka+Kb] = Sa [Sa ; Sb]^T, orth[0] = mean |G[i][j]| over j > i, inv_norm [Ka+Kb] for the backward.  Kb may be 0 (base training: G = S S^T). */
int sl_pop_proto_fwd(const float* Ea, int Ka, const float* Eb, int Kb, int C, float* Sa, float* Sb, float* inv_norm, float* G, float* orth,
                     sl_stream_t stream);
/* 1 when sl_pop_proto_fwd AND sl_pop_proto_bwd serve (Ka, Kb, C) -- both are one-block kernels with everything in the LDS, the backward needs about twice the
 * forward's -- else 0: the caller then evaluates pspnet_pop.py:96-99,185-186 / criterion.py:37-43 with its own tensor ops (GFSS_Model._protos). */
int sl_pop_proto_ok(int Ka, int Kb, int C);
/* gradients wrt Ea / Eb (either may be NULL) from dSa / dSb (may be NULL = zero) and the scalar dorth (may be NULL) */
int sl_pop_proto_bwd(const float* Sa, int Ka, const float* Sb, int Kb, int C, const float* inv_norm, const float* G, const float* dSa,
                     const float* dSb, const float* dorth, float* dEa, float* dEb, sl_stream_t stream);
int sl_pop_proto_rows(int dtype, const float* S, int Kt, int C, void* dst, sl_stream_t stream);
/* z[r] = h[r] . w   (classifier.4, Cout = 1; h is the post-ReLU activation) */
int sl_rowdot_fwd(int dtype, const void* h, const float* w, float* z, long long R, int C, sl_stream_t stream);
/* dh[r][c] = dz[r]*w[c]*(h[r][c] > 0);  partial[blk][c] = sum_r dz[r]*h[r][c]  (float[nblk][C]) */
int sl_rowdot_bwd_rows(long long R, int C);
int sl_rowdot_bwd(int dtype, const void* h, const float* w, const float* dz, void* dh, float* partial, long long R,
                  int C, sl_stream_t stream);
/* out[c] = sum_blk partial[blk][c] */
/* per-channel sums over the rows of an NHWC activation tensor [rows][C] (nn.Linear / conv bias gradients): partial[sl_colsum_rows_blocks][C], to be summed by
 * sl_colsum_finalize / sl_colsum_finalize_multi (fixed order: bit-stable) */
int sl_colsum_rows_blocks(long long rows, int C, int dtype);
int sl_colsum_rows_partial(int dtype, const void* x, long long rows, int C, float* partial, sl_stream_t stream);
int sl_colsum_finalize(const float* partial, int nblk, int C, float* out, sl_stream_t stream);
/* the same for up to SL_COLSUM_MAX independent partial buffers in ONE launch (host struct, read during the call) */
#define SL_COLSUM_MAX 12
typedef struct SlColsumBatch {
  int n;
  const float* part[SL_COLSUM_MAX];
  float* out[SL_COLSUM_MAX];
  int nblk[SL_COLSUM_MAX];
  int C[SL_COLSUM_MAX];
} SlColsumBatch;
int sl_colsum_finalize_multi(const SlColsumBatch* batch, sl_stream_t stream);
/* SyncBatchNorm (train_base.py:175-176: nn.SyncBatchNorm under DDP): the per-channel totals of the statistic partials are kept in double
 * for the all-reduce over the process group (out[c] = sum_blk partial[blk][c], double), then handed to sl_bn_finalize_train /
 * sl_bn_bwd_finalize as TWO float partial rows hi_lo[0][i] + hi_lo[1][i] == totals[i] to 48 bits. */
int sl_colsum_f64(const float* partial, int nblk, int C, double* out, sl_stream_t stream);
int sl_f64_split(const double* totals, int n, float* hi_lo, sl_stream_t stream);
/* preds NCHW float [B][1+Kt][N]: channel 0 = z_bg[r], channel 1+k = a[k]*max(p,0) + b[k]*max(-p,0) */
int sl_pop_combine_fwd(const float* proj, const float* z_bg, const float* a, const float* b, int Kt, float* preds,
                       int B, int N, sl_stream_t stream);
/* dz_bg[r], dproj[r][k] = dpred*(a[k]*[p>0] - b[k]*[p<0]),  dab_partial[blk][2*Kt] = partial sums of
 * (dpred*max(p,0), dpred*max(-p,0)) */
int sl_pop_combine_bwd_rows(int B, int N);
int sl_pop_combine_bwd(const float* dpreds, const float* proj, const float* a, const float* b, int Kt, float* dz_bg,
                       float* dproj, float* dab_partial, int B, int N, sl_stream_t stream);
/* t[r][k] = dproj[r][k] - dg[r].S[k];  dq[r] = dg[r] + sum_k t[r][k] S[k];
 * dS_partial[blk][k][c] = sum_r t[r][k] q[r][c] - proj[r][k] dg[r][c] */
int sl_pop_decompose_bwd_rows(long long R);
int sl_pop_decompose_bwd(int dtype, const void* dg, const void* feats, const float* S, const float* proj,
                         const float* dproj, int Kt, void* dq, float* dS_partial, long long R, int C,
                         sl_stream_t stream);

/* ------------------------------------------------------------------------------------------------ loss
 * loss/criterion.py:51-52: F.interpolate(bilinear, align_corners=True) to the label size fused with
 * CrossEntropyLoss(ignore_index, mean); the H x W logits are never written. */
int sl_upsample_ce_rows(int B, int H, int W);
/* partial[blk][2] = (sum of pixel losses, number of valid pixels) */
int sl_upsample_ce_fwd(const float* logits, const int64_t* target, int B, int K, int h, int w, int H, int W,
                       int ignore_index, float* partial, sl_stream_t stream);
int sl_upsample_ce_finalize(const float* partial, int nblk, float* loss_and_count, sl_stream_t stream);
/* dlogits[b][k][i][j] = gscale/nvalid * sum over pixels of weight * (softmax - onehot); loss_and_count from fwd */
int sl_upsample_ce_bwd(const float* logits, const int64_t* target, const float* loss_and_count, const float* gscale,
                       int B, int K, int h, int w, int H, int W, int ignore_index, float* dlogits, sl_stream_t stream);
/* pspnet_pop.py:221-231: argmax of the upsampled (align_corners=True) [K2][h][w] logits of tile b, ids > 0 shifted
 * by n_base, written into mask[b] where mask == 0 (in place, int64). */
int sl_pseudo_label(const float* logits, int K2, int h, int w, int64_t* mask, int B, int H, int W, int n_base,
                    sl_stream_t stream);
/* argmax over channels of the upsampled (align_corners=True) logits -> uint8 labels (eval_base.py:168-169) */
int sl_upsample_argmax(const float* logits, int B, int K, int h, int w, int H, int W, uint8_t* labels,
                       sl_stream_t stream);
/* eval_base.py:168,189-190: the upsampled logits themselves (the 'outputs' array of the per-tile .mat probability dumps), NCHW float */
int sl_upsample_logits(const float* logits, int B, int K, int h, int w, int H, int W, float* out, sl_stream_t stream);
/* fusemat.py:35-52: argmax over classes of the mean of n_models probability maps [K][hw] (mats_dev: device array of n_models float pointers),
 * summed in list order like the reference's in-place `mats[idx] += prob` */
int sl_fuse_argmax(const void* mats_dev, int n_models, int K, long long hw, uint8_t* labels, sl_stream_t stream);
/* utils/pyt_utils.py:293-305 intersectionAndUnionGPU: hist[0..K) inter, [K..2K) pred area, [2K..3K) target area
 * (int64 counts; caller zeroes hist). */
int sl_iou_hist(const uint8_t* pred, const int64_t* target, long long n, int K, int ignore_index, long long* hist,
                sl_stream_t stream);
/* eval_base.py:171-177 / utils/pyt_utils.py:182-200: cm[t*K + p] += 1 over pixels with target != ignore_index (int64 counts, K <= 64;
 * caller zeroes cm). */
int sl_confusion_matrix(const uint8_t* pred, const int64_t* target, long long n, int K, int ignore_index, long long* cm,
                        sl_stream_t stream);
/* networks/pspnet.py:7-15 masked_average_pooling (dead code upstream; prototype-init utility here):
 * proto[c] = mean_b( sum_hw f*m / (sum_hw m + 1e-5) ), m = bilinear(mask, align_corners=True) at feature size.
 * feature: NHWC dtype [B][h][w][C]; mask: float [B][H][W]. */
int sl_masked_avg_pool(int dtype, const void* feature, const float* mask, int B, int h, int w, int C, int H, int W,
                       float* proto, sl_stream_t stream);

/* ------------------------------------------------------------------------------------------------ utilities */
/* NHWC dtype <-> NCHW float (module boundary / tests) */
int sl_nhwc_to_nchw_f32(int dtype, const void* src, float* dst, int B, int H, int W, int C, sl_stream_t stream);
int sl_nchw_f32_to_nhwc(int dtype, const float* src, void* dst, int B, int H, int W, int C, sl_stream_t stream);
/* ---- optimizer ------------------------------------------------------------------------------------------------------------
 * torch.optim.AdamW (train_base.py:197-204; stepped twice per iteration, :262-264) over n parameter tensors in ONE launch.
 * table_dev: device array of n 64-byte records { float* p; const float* g; float* m; float* v; long long numel; float lr, wd;
 * long long start; int group; int pad; } with `start` the running count of 4096-element chunks (ceil(numel/4096)) of the preceding records
 * and total_chunks their grand total.  bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) for this step, the
 * `_next` pair for step t+1 (used when repeat == 2: two consecutive steps on the same gradient in one pass).  grad_scale: optional
 * device scalar multiplied into every gradient (the clip_grad_norm_ coefficient). */
int sl_adamw_multi(const void* table_dev, int n, long long total_chunks, float beta1, float beta2, float eps,
                   float bias_correction1, float bias_correction2_sqrt, float bias_correction1_next, float bias_correction2_sqrt_next,
                   int repeat, const float* grad_scale, sl_stream_t stream);
/* The same step for a captured HIP graph (segland_amd/graph_step.py): every step-dependent scalar lives in device memory.
 * hyper_dev = { bias_correction1, bias_correction2_sqrt, bias_correction1_next, bias_correction2_sqrt_next, then (lr, weight_decay) per
 * parameter group }; a record's `group` selects its pair and the record's own lr / wd are ignored. */
int sl_adamw_multi_dev(const void* table_dev, int n, long long total_chunks, float beta1, float beta2, float eps,
                       const float* hyper_dev, int repeat, const float* grad_scale, sl_stream_t stream);
/* torch.nn.utils.clip_grad_norm_'s total norm and clip coefficient (train_base.py:258-261 through utils/pyt_utils.py:327-347; ft_pop.py:250) over a list of contiguous
 * float gradient tensors: sl_grad_sqnorm_multi (up to SL_NORM_MAX tensors per launch; the addresses are kernel arguments, so a captured step carries them) writes one
 * partial sum of squares per 4096-element chunk to partial[batch->chunk_base + chunk]; sl_grad_norm_finalize sums the nchunks partials in a fixed order and writes
 * out[0] = norm = sqrt(sum) * inv_div and out[1] = min(1, max_norm / (norm + 1e-6)) * inv_div (inv_div = 1 / world size when the gradients hold the sum over ranks).
 * chunk0[i] = first chunk of tensor i within this batch (chunk0[0] = 0, chunk0[n] = the batch's chunk count). */
#define SL_NORM_MAX 64
typedef struct SlNormBatch { const void* grad[SL_NORM_MAX]; long long numel[SL_NORM_MAX]; int chunk0[SL_NORM_MAX + 1]; int n; int chunk_base; } SlNormBatch;
int sl_grad_sqnorm_multi(const SlNormBatch* batch, float* partial, sl_stream_t stream);
int sl_grad_norm_finalize(const float* partial, int nchunks, float max_norm, float inv_div, float* out, sl_stream_t stream);
/* torch.optim.SGD (momentum, weight decay; dampening 0, nesterov off) over all parameters in one launch: ft_pop.py:205-209,252.  Table records as for sl_adamw_multi
 * (64 bytes: p, g, momentum buffer or NULL, unused, numel, lr, weight_decay, chunk start, group, pad); hyper_dev = (lr, weight_decay) per parameter group in device
 * memory or NULL (the records' own values): with it the launch is capturable while the driver changes the learning rate every iteration.  grad_scale: optional
 * 1-element device float multiplied into every gradient. */
int sl_sgd_multi(const void* table_dev, int n, long long total_chunks, float momentum, const float* hyper_dev, const float* grad_scale, sl_stream_t stream);
/* n <= 16 host floats -> device memory as kernel arguments (stream-ordered, no host -> device copy) */
int sl_store_floats(float* dst_dev, int n, const float* values_host, sl_stream_t stream);

/* Many small strided fp32 copies in one launch (the zero-padded staging copies of weights / biases / BatchNorm vectors at the channel pitch, refreshed once per
 * optimizer step).  table_dev: n entries {float* dst; const float* src; int rows, cols, dst_pitch, pad; int64 start} (40 bytes): src is [rows][cols] contiguous,
 * row r goes to dst + r * dst_pitch; start = running count of 1024-element chunks of the preceding entries, total_chunks their grand total. */
int sl_copy2d_multi(const void* table_dev, int n, long long total_chunks, sl_stream_t stream);

/* The relative-position bias tiles of n Swin blocks in one launch: out[h][p] = table[index[p]][h] (swintransformer.py:128-131, the gather + permute of
 * WindowAttention.forward).  table_dev: device array of n entries {float* out; const float* table; const int64_t* index; int heads; int npair;} (32 bytes). */
int sl_relpos_gather_multi(const void* table_dev, int n, sl_stream_t stream);

/* ---- Swin-POP path ---------------------------------------------------------------------------
 * Token maps are NHWC images [B][H][W][pitch] with C real channels and a ZERO channel pad up to `pitch` (a multiple of 64 so that the
 * nn.Linear layers run on sl_conv2d_* as 1x1 convs; Swin-T/S: C = 96, pitch = 128).  Replaces, on the GPU, these call sites of the
 * reference: networks/backbones/swintransformer.py PatchEmbed.forward :413-433, nn.LayerNorm (norm1 / norm2 / PatchMerging.norm /
 * norm{i}, :214,244,287,636), WindowAttention.forward :118-149 together with the pad / roll / window_partition / window_reverse /
 * attention mask of SwinTransformerBlock.forward :208-238 and BasicLayer.forward :363-379, Mlp's nn.GELU :36, PatchMerging's 2x2 gather
 * :280-284, timm DropPath :243-244; networks/swin_pop.py F.interpolate(align_corners=True) :33,150-153,167, nn.Upsample :133-137,
 * nn.Dropout2d :21. */
typedef struct SlWinDesc {
  int dtype, B, H, W;       /* token map (before the padding to multiples of 7, which the kernel does by index arithmetic) */
  int C, heads;             /* C = heads * 32 */
  int qkv_pitch, out_pitch; /* channels per token of the qkv tensor ([q | k | v | pad]) and of the output ([C | pad]) */
  int shift;                /* 0 (W-MSA) or 3 (SW-MSA: cyclic shift + region mask) */
} SlWinDesc;
typedef struct SlResizeDesc {
  int dtype, B, h, w, H, W; /* source h x w -> destination H x W */
  int C;                    /* channels resized */
  int src_pitch, src_off;   /* channel pitch of the source tensor and first channel of the window */
  int dst_pitch, dst_off;
  int align_corners;        /* F.interpolate(mode='bilinear', align_corners=...) */
  int accumulate;           /* add into the output instead of overwriting it (top-down FPN sums, sum of the level heads) */
  int src_f32;              /* the SOURCE-side tensor is float whatever dtype says (the pyramid-pooling stage maps stay fp32) */
} SlResizeDesc;
/* conv 4x4 stride 4 (3 -> C) + bias from the NCHW float image; zero padding to multiples of 4 (:417-421); out [B][ceil(H/4)][ceil(W/4)][pitch] */
int sl_patch_embed_fwd(int dtype, const float* img_nchw, const float* w_oihw, const float* bias, void* out, int B, int H, int W, int C,
                       int out_pitch, sl_stream_t stream);
/* weight gradient of the patch embedding: col [B*ceil(H/4)*ceil(W/4)][64] = the 48 patch values of every token (k = ci*16 + ky*4 + kx) + 16
 * zeros, then dW = sl_conv2d_bwd_weight(col as [1,1,T,64], dy as [1,1,T,pitch]) on the MFMA kernel */
int sl_patch_im2col(int dtype, const float* img_nchw, void* col, int B, int H, int W, sl_stream_t stream);
/* y = (x - mean) * rstd * gamma + beta over the first C channels of each row, pad channels of y zeroed; mean_rstd [rows][2] (may be null) */
int sl_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, long long rows, int C,
                     int x_pitch, int y_pitch, float eps, sl_stream_t stream);
/* dx (+= addend, e.g. the gradient of the residual branch); dgamma_dbeta_partial [sl_layernorm_bwd_rows(dtype, rows, C, dx_pitch)][2][C] or null */
int sl_layernorm_bwd_rows(int dtype, long long rows, int C, int dx_pitch);
int sl_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const void* addend, void* dx,
                     float* dgamma_dbeta_partial, long long rows, int C, int dy_pitch, int x_pitch, int dx_pitch, sl_stream_t stream);
/* The same with a second result dx_scaled = dx (as rounded to dtype) * row_scale[row / rows_per_sample]: the gradient that enters the next residual branch behind
 * DropPath's per-sample factor (swintransformer.py:246-249 backward) -- one launch and one pass over dx less than sl_scale_add on the result, same bits. */
int sl_layernorm_bwd_scaled(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const void* addend, void* dx,
                            void* dx_scaled, const float* row_scale, long long rows_per_sample,
                            float* dgamma_dbeta_partial, long long rows, int C, int dy_pitch, int x_pitch, int dx_pitch, sl_stream_t stream);
/* exact GELU on n contiguous elements (n % 8 == 0) */
int sl_gelu_fwd(int dtype, const void* h, void* y, long long n, sl_stream_t stream);
int sl_gelu_bwd(int dtype, const void* h, const void* dy, void* dh, long long n, sl_stream_t stream);
/* PatchMerging: x [B][H][W][x_pitch] -> xm [B][ceil(H/2)][ceil(W/2)][4*C] in the reference's (x0, x1, x2, x3) order; and its transpose */
int sl_patch_merge_gather(int dtype, const void* x, void* xm, int B, int H, int W, int C, int x_pitch, sl_stream_t stream);
int sl_patch_merge_scatter(int dtype, const void* dxm, void* dx, int B, int H, int W, int C, int dx_pitch, sl_stream_t stream);
/* bilinear resize of a channel window, NHWC; the backward (d(dst) -> d(src)) is a gather over the destination pixels (bit-stable) */
int sl_bilinear_fwd(const SlResizeDesc* d, const void* src, void* dst, sl_stream_t stream);
int sl_bilinear_bwd(const SlResizeDesc* d, const void* ddst, void* dsrc, sl_stream_t stream);
/* dst = base + bilinear(src): the top-down sum of swin_pop.py:150-153 / the level-head sum of :167-169 without a copy of base first (base and dst share
 * d's dst_pitch / dst_off; d->accumulate must be 0) */
int sl_bilinear_fwd_add(const SlResizeDesc* d, const void* src, const void* base, void* dst, sl_stream_t stream);
/* out = (addend ? addend : 0) + x * scale[b] (per_channel 0: DropPath) or x * scale[b][c] (per_channel 1: Dropout2d; pad channels -> 0) */
int sl_scale_add(int dtype, const void* x, const float* scale, const void* addend, void* out, int B, long long rows_per_sample, int C,
                 int pitch, int per_channel, sl_stream_t stream);
/* softmax(q k^T / sqrt(32) + rel_bias[head] + shift mask) v per 7x7 window and head.  qkv [B][H][W][qkv_pitch]; qkv_bias [3C] (the q/k/v of
 * the zero-padded tokens); rel_bias [heads][49][49] (relative_position_bias_table gathered by relative_position_index); out [B][H][W][out_pitch]. */
int sl_window_attention_fwd(const SlWinDesc* d, const void* qkv, const float* qkv_bias, const float* rel_bias, void* out,
                            sl_stream_t stream);
/* backward (probabilities recomputed): dqkv like qkv; drel_partial [sl_window_attention_bwd_chunks()][heads][49*49] (sum the chunks);
 * pad_partial [sl_window_attention_windows()][heads][96]: q/k/v gradients that reach the qkv bias through the pad tokens (sum the windows) */
int sl_window_attention_bwd_chunks(const SlWinDesc* d);
/* d relative_position_bias_table [rows][heads] from d bias [heads][npair] (npair = 49*49): dtable[t][h] = sum_j dbias[h][pairs[t][j]] over the
 * pairs[t][0..m) >= 0 (constant lists of the (query, key) pairs with relative offset t; swintransformer.py:128-131 backward, fixed order) */
int sl_relpos_table_grad(const float* dbias, const int* pairs, int rows, int m, int heads, int npair, float* dtable, sl_stream_t stream);
/* the same + the qkv bias gradient in the launch: dbq [3][heads][32] = dbq_colsum (column sums of dqkv, same layout) + dpad [heads][3][32] (the finalized pad_partial
 * of sl_window_attention_bwd: the pad tokens' k / v are the bias itself, swintransformer.py:208-213) */
int sl_relpos_table_grad_bias(const float* dbias, const int* pairs, int rows, int m, int heads, int npair, float* dtable, const float* dbq_colsum,
                              const float* dpad, float* dbq, sl_stream_t stream);
int sl_window_attention_windows(const SlWinDesc* d);
int sl_window_attention_bwd(const SlWinDesc* d, const void* qkv, const float* qkv_bias, const float* rel_bias, const void* dout, void* dqkv,
                            float* drel_partial, float* pad_partial, sl_stream_t stream);

/* ---- OpenEarthMap tile preparation ------------------------------------------------------------
 * dataset/base_dataset.py crop :140-174, pad :88-104, random_flip :106-110, fixed_random_rotate :134-138, normalize :29-34, totensor
 * :36-43 and the label re-indexing of dataset/oem.py:113-133 / oem_ft.py:197 for a batch of decoded tiles in one launch.
 * tiles_dev: device array of B 32-byte records { const uint8_t* img [H][W][3]; const uint8_t* lbl [H][W] or null; int H, W, h_off, w_off; };
 * flip_rot_dev: device int[2*B] = {flip (0/1), k of np.rot90}; crops must be square when any k is odd.  mean3 / std3: HOST pointers (in the
 * reversed channel order normalize() applies them in).  label_lut_dev: 256-entry table or null (identity).  out_img [B][3][crop_h][crop_w]
 * float, out_lbl [B][crop_h][crop_w] int64 (may be null).  Pixels beyond the tile are the padding: image value 0 BEFORE normalisation,
 * label ignore_label. */
int sl_augment_batch(const void* tiles_dev, const int* flip_rot_dev, int B, int crop_h, int crop_w, const double* mean3_host,
                     const double* std3_host, int ignore_label, const uint8_t* label_lut_dev, float* out_img, long long* out_lbl,
                     sl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
