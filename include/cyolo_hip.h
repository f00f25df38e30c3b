/*
 * cyolo_hip.h -- C ABI of libcyolo_hip.so, the MI355X (gfx950) hot path of Complex-YOLOv4.
 *
 * The reference (maudzung/Complex-YOLOv4-Pytorch) has no native operator API: its hot path is a set
 * of Python call sites.  Every entry point below names the reference interface it
 * stands behind (file:line under /root/reference/src).  Conventions, all entry points:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller
 *     (torch-allocated), unless the name ends in _host;
 *   - stream-ordered on the given hipStream_t, no internal synchronisation, no allocation;
 *   - returns 0 on success, CY_ERR_ARG (-1) for a rejected argument, -(1000+hipError_t) when a
 *     launch failed; never throws;
 *   - activations are NHWC "views": element (n,h,w,c) of a view with channel stride ld lives at
 *     base[((n*H + h)*W + w)*ld + c]; `dtype` selects the storage type of activations/weights
 *     (CY_F16 = performance mode, CY_F32 = parity mode, exact f32 MFMA); accumulation is always f32.
 */
#ifndef CYOLO_HIP_H
#define CYOLO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cy_stream_t; /* a hipStream_t */

enum { CY_F16 = 0, CY_BF16 = 1, CY_F32 = 2 };
enum { CY_ACT_LINEAR = 0, CY_ACT_LEAKY = 1, CY_ACT_MISH = 2 };
enum { CY_ERR_ARG = -1, CY_ERR_UNSUPPORTED = -3 /* a valid call this kernel variant does not take: use the general path */ };
/* cy_conv_igemm flags */
enum { CY_CONV_STATS = 1, CY_CONV_BIAS_F32OUT = 2, CY_CONV_ACCUM = 4, CY_CONV_TRANSPOSED = 8, CY_CONV_AFFINE_ACT = 16,
       CY_CONV_STATS_DET = 32, CY_CONV_BNBWD_SUMS = 64 /* set by cy_conv_dgrad_bn_sums only */,
       CY_CONV_BN_FUSED = 128 /* set by cy_conv_bn_act_train only */ };
/* bits 8-11 of `flags`: kernel / tile hint of the call (0 = library default), see cy_conv_igemm */
enum { CY_CONV_TILE_SHIFT = 8 };
#define CY_CONV_TILE(h) ((h) << CY_CONV_TILE_SHIFT)

int cy_version(void);
/* Number of compute units / wavefront size of the current device (sanity for the loader). */
int cy_device_info(int* cus, int* wave);

/* ------------------------------------------------------------------------------------------------
 * Convolution stack  (reference: nn.Sequential(Conv2d, BatchNorm2d, Mish|LeakyReLU) built at
 * models/darknet2pytorch.py:247-278 and run at :178; backward = torch autograd of the same)
 * ---------------------------------------------------------------------------------------------- */

/* fp32 OIHW torch weights [Co][Ci][ks][ks] -> packed `dtype` matrices
 *   wf [CoPad][ks*ks*CiPad]  (forward / wgrad layout, K index = (kh*ks+kw)*CiPad + ci)
 *   wd [CiPad][ks*ks*CoPad]  (dgrad layout,          K index = (kh*ks+kw)*CoPad + co), may be NULL.
 * Padding rows/columns are written as zero. */
int cy_pack_weights(const float* w, int Co, int Ci, int ks, int CoPad, int CiPad, int dtype, void* wf, void* wd,
                    cy_stream_t s);

/* Table-driven variants: ONE launch for every conv of the network (the per-layer calls above cost a kernel boundary
 * each, ~110 per step).  `desc` is a device array of cy_pack_desc / cy_reduce_desc; `blocks` a device array of
 * (descriptor index, first element / 256) pairs, one per 256-thread block covering CY_MULTI_ELEMS elements; for the
 * pack table an "element" run of CY_MULTI_ELEMS stands for one 64 x 64 (co, ci) tile of the padded weight matrix
 * (tile = second entry / 4, tiles enumerated ci-fastest, ks <= 3); for the reduce table the unit is one (co, ci) pair
 * with all ks*ks taps and a block covers 256 / lanes pairs (second entry = first pair / 32). */
typedef struct {
    const float* w; void* wf; void* wd;
    int Co, Ci, ks, CoPad, CiPad;
    int wd_ld;      /* 0, or the row stride (elements) of a WIDER dgrad matrix this layer fills a column range of (ks = 1) */
} cy_pack_desc;
typedef struct {
    const float* part; float* grad;
    int split, CoRows, CiPad, ks, Co, Ci;
    int lanes, flags;    /* threads sharing one (co, ci) pair, each folding every lanes-th slab: 1, 2, 4 or 8 (ks 1 / 3);
                          * flags bit 0: write zeros back over the slab elements just read (slabs of cy_conv_wgrad's
                          * atomic mode stay resident and must be zero at the start of the next step) */
} cy_reduce_desc;
enum { CY_MULTI_ELEMS = 1024 };
int cy_pack_weights_multi(const cy_pack_desc* desc, const int32_t* blocks, int nblocks, int dtype, cy_stream_t s);
int cy_wgrad_reduce_multi(const cy_reduce_desc* desc, const int32_t* blocks, int nblocks, float scale, int accumulate,
                          cy_stream_t s);

/* Fused multi-tensor Adam (torch.optim.Adam semantics: L2 weight decay added to the gradient, bias-corrected moments,
 * p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)); one launch for every parameter of the model.  Replaces the reference's
 * torch.optim.Adam.step() (src/utils/train_utils.py:21-50 builds the three parameter groups).  zero_grad != 0 clears each
 * gradient after it is consumed. */
typedef struct {
    float* p; float* g; float* m; float* v;
    int64_t n;
    int group, pad_;      /* parameter group: learning rate / weight decay come per call (LR schedulers change them) */
} cy_adam_desc;
int cy_adam_multi(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float beta1, float beta2, float eps,
                  float bias_corr1, float bias_corr2, int zero_grad, const float* group_lr_host,
                  const float* group_wd_host, int ngroups, const int32_t* skip_flag, cy_stream_t s);
/* cy_adam_multi with the step count t kept on the device (dynamic loss scaling, where the host learns about a skipped
 * step one step late): the bias corrections 1 - beta^t are computed in the kernel from t = *step_in + 1, and *step_out
 * receives t -- or *step_in unchanged when *skip_flag is set.  A skipped step therefore never advances the bias correction,
 * as torch.optim.Adam under torch.cuda.amp.GradScaler.  step_in and step_out are two distinct device int32 (ping-pong). */
int cy_adam_multi_dev(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float beta1, float beta2, float eps,
                      const int32_t* step_in, int32_t* step_out, int zero_grad, const float* group_lr_host,
                      const float* group_wd_host, int ngroups, const int32_t* skip_flag, cy_stream_t s);
/* The hipGraph-capturable form of the step: the step count is a device int32 advanced by a one-thread kernel in the same
 * call (not when *skip_flag is set), bias corrections are computed from it in the kernel, and the per-group learning rates /
 * weight decays are read from a device array [lr x 8, wd x 8] -- so a captured launch needs no re-capture when the step
 * number or the learning-rate schedule moves on.  Same arithmetic as cy_adam_multi. */
int cy_adam_multi_graph(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float beta1, float beta2, float eps,
                        int32_t* step_counter, const float* group_lr_wd_dev, int zero_grad, const int32_t* skip_flag,
                        cy_stream_t s);
/* Fused multi-tensor SGD with momentum / Nesterov (torch.optim.SGD semantics, dampening 0): the reference's other
 * optimizer choice (src/utils/train_utils.py:35-37: SGD(lr, momentum, nesterov=True)).  Uses cy_adam_desc with `m` as the
 * momentum buffer (`v` unused, may be NULL); first_step != 0 initialises the buffer with the gradient as torch does. */
int cy_sgd_multi(const cy_adam_desc* desc, const int32_t* blocks, int nblocks, float momentum, int nesterov, int first_step,
                 int zero_grad, const float* group_lr_host, const float* group_wd_host, int ngroups, const int32_t* skip_flag,
                 cy_stream_t s);
/* Dynamic loss scaling (the fp16 counterpart of torch.cuda.amp.GradScaler; the reference trains in fp32 and needs none):
 * *flag = 1 when any of the n fp32 gradient elements (16-byte aligned) is inf or nan, else 0.  The optimizer entry points
 * above skip the whole step when their skip_flag (device pointer, may be NULL) is non-zero -- no host round trip. */
int cy_grad_nonfinite(const float* g, int64_t n, int32_t* flag, cy_stream_t s);


/* NCHW fp32 image batch [N][C][H][W] -> NHWC `dtype` view with CPad channels (extra channels zero).
 * (reference: the imgs tensor handed to Darknet.forward, darknet2pytorch.py:162) */
int cy_nchw_to_nhwc(const float* x, int N, int C, int H, int W, int CPad, int dtype, void* out, cy_stream_t s);

/* Implicit-GEMM convolution on MFMA.  One kernel family serves
 *   forward : out[n,oh,ow,co] = sum_{kh,kw,ci} g[n, oh*stride-pad+kh, ow*stride-pad+kw, ci] * w[co][(kh,kw,ci)]
 *   dgrad   : (CY_CONV_TRANSPOSED) out = dX, g = dY, w = wd:  gh = (oh+pad-kh)/stride when divisible
 * g: view (N,GH,GW,GC,ldg);  out: view (N,OH,OW,OC,ldo);  w: [wrows][ks*ks*GC].
 * flags: CY_CONV_STATS       -> also ADD (fp32 atomics) per-channel partial (sum, sumsq) of the f32 accumulators into
 *                               stats_part[bin][2][OC], bin = tile % 16  (BatchNorm batch statistics).  The table must
 *                               be zero on entry; cy_bn_finalize folds it and leaves it zeroed.
 *        CY_CONV_BIAS_F32OUT -> out is float regardless of dtype and bias[OC] is added (the YOLO head convs)
 *        CY_CONV_ACCUM       -> out += result (gradient fan-in of routes / shortcuts)
 *        CY_CONV_TILE(h)     -> which kernel runs the call (results are the same; a caller that launches the same shape
 *                               every step times the candidates once): 0 library default, 1 the 4-wave kernels, 2-5 the
 *                               8-wave pipelined kernel with a 128 / 192 / 256 / 384-pixel tile, 6 with its own tile policy,
 *                               7-9 its loader / compute split (4 + 8 waves) with a 128 / 192 / 256-pixel tile
 *                               10 the direct streaming kernels of conv_direct.hip also where the library default would not
 *                               pick them (1x1 launches of fewer than 256 k pixels)
 *                               11-13 the slab kernel of conv_pipe.hip (3x3 / stride 1 / pad 1, forward and dgrad, more than 64
 *                               output channels: one halo'd slab of input pixels per 64-channel chunk in LDS serves all nine
 *                               taps, wave pairs split the K step): 11 its own tile policy, 12 / 13 a 192 / 256-pixel tile
 *                               (the hint is ignored where that kernel does not apply: f32, first layers, fp32 output)
 * Returns the number of stats rows written through *stats_rows when non-NULL. */
int cy_conv_igemm(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* out, int OH,
                  int OW, int OC, int ldo, int ks, int stride, int pad, int dtype, int flags, const float* bias,
                  float* stats_part, int* stats_rows_host, cy_stream_t s);
/* TRAINING-mode conv block in ONE kernel (reference work unit darknet2pytorch.py:247-278 under model.train(): Conv2d ->
 * BatchNorm2d with BATCH statistics -> Mish | LeakyReLU (+ the folded [shortcut])): the fused BN + activation epilogue of
 * the north star.  Batch statistics need every tile's sums before any tile can normalise, so the launch is two-phase:
 * every block adds its (sum, sum of squares) into `stats_bins` (the CY_CONV_STATS table), arrives at a grid-wide ticket,
 * stores its pre-BN tile to `raw` (kept for the backward pass) while the others arrive, then folds the bins (double, bin order:
 * what cy_bn_act_fwd_fused does), and writes out = act((T)acc * scale + shift) (+ res) from its accumulators -- the pre-BN
 * tensor is never read back.  Block row 0 also writes `vec` = (mean, invstd, scale, shift), the running statistics and
 * num_batches_tracked (NULL: left alone); the grid zeroes `zero_table` (the other table of the alternating pair).
 * Only for launches whose grid is co-resident (one round of the 8-wave kernel: blocks <= CUs x occupancy, checked here)
 * and shapes the pipelined kernel takes (16-bit types, GC % 64 == 0, OC % 8 == 0, stride-1 or -2 forward): anything else
 * returns CY_ERR_UNSUPPORTED and the caller keeps cy_conv_igemm + cy_bn_act_fwd_fused.  `ticket`: int32[4] zeroed once by the
 * caller (arrivals, departures -- both back at 0 when the launch ends -- and [2] != 0 after a launch whose wait for the
 * grid timed out: its outputs are invalid).  Results equal the two-launch path bit for bit given equal bins. */
int cy_conv_bn_act_train(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* raw,
                         int OH, int OW, int OC, int ldraw, void* out, int ldout, const void* res, int ldres, int ks,
                         int stride, int pad, int dtype, int flags, float* stats_bins, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, void* num_batches_tracked,
                         float momentum, float eps, float* vec, float* zero_table, int zero_n, int act, int32_t* ticket,
                         cy_stream_t s);
/* Eval-mode conv block in ONE kernel: out = act(conv(g, w) * scale[co] + shift[co]) (+ res), scale/shift being the
 * BatchNorm affine of the running statistics (cy_bn_eval_affine).  The pre-BN tensor is never written
 * (reference: the same nn.Sequential under model.eval(), evaluate.py:32-44).  flags: 0 or CY_CONV_TILE(h). */
int cy_conv_bn_act_eval(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* out,
                        int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype, const float* scale,
                        const float* shift, int act, const void* res, int ldres, int flags, cy_stream_t s);

/* Input-gradient conv (dgrad) whose epilogue ALSO accumulates the BatchNorm-backward sums of the layer that produced
 * its output tensor: with out = dL/dy of that layer (after the CY_CONV_ACCUM fan-in, as stored), raw = the layer's
 * pre-BN tensor (same pixels, row stride ldraw), z = raw*scale + shift,
 *     sums_part[bin][0][c] += sum_p out * act'(z)            (= d beta)
 *     sums_part[bin][1][c] += sum_p out * act'(z) * (raw - mean) * invstd      (= d gamma)
 * -- the table cy_bn_act_bwd_reduce would produce from a second pass over (raw, out); cy_bn_act_bwd_apply_fused takes it
 * with rows = cy_conv_stats_rows().  The layer's gradient tensor is then read once (by the apply pass) instead of twice.
 * Reference: the autograd graph of Conv2d -> BatchNorm2d -> Mish, darknet2pytorch.py:247-278.
 * Runs on the 8-wave pipelined kernel only: 16-bit dtype, stride 1, GC % 64 == 0, OC % 8 == 0, ldo % 8 == 0,
 * ldraw % 8 == 0, 16-byte aligned raw; anything else returns CY_ERR_ARG (the caller keeps the separate reduce pass).
 * flags: CY_CONV_TRANSPOSED, CY_CONV_ACCUM, CY_CONV_STATS_DET (one table row per pixel tile), CY_CONV_TILE(2..9, 11..13). */
int cy_conv_dgrad_bn_sums(const void* g, int N, int GH, int GW, int GC, int ldg, const void* w, int wrows, void* out,
                          int OH, int OW, int OC, int ldo, int ks, int stride, int pad, int dtype, int flags,
                          const void* raw, int ldraw, const float* mean, const float* invstd, const float* scale,
                          const float* shift, int act, float* sums_part, cy_stream_t s);

/* Number of kernel launches since load that ran on the 8-wave pipelined kernel (csrc/conv_pipe.hip: CY_F16 / CY_BF16,
 * Cin a multiple of 64, 16-bit output with OC and ldo multiples of 8, enough pixel tiles to fill the chip); every
 * other launch runs on conv_igemm.hip's 4-wave kernels.  Diagnostics for tests and profiles. */
int64_t cy_pipe_launches(void);
/* Number of launches since load that ran on the direct streaming kernels (csrc/conv_direct.hip, 16-bit types): the 3x3
 * forward convs 3(8) -> 32 stride 1 and 32 -> 64 stride 1 / 2 with pad 1, and 1x1 stride-1 convs (forward, eval, dgrad with
 * or without fan-in accumulation) of 64->64, 128->64, 64->128, 64->32, 32->64 channels on launches of >= 256 k pixels;
 * BN statistics into shared bins or the eval-mode epilogue.  CY_CONV_TILE(1) or CY_CONV_DIRECT=0 in the environment keep
 * a call on the implicit-GEMM kernels. */
int64_t cy_direct_launches(void);
/* Test / tool switch, not used on the step path.  mode 0: never use the pipelined kernel, 1: default (hints and the
 * eval-mode epilogue select it), 2: every launch that qualifies; cap x bn (0 x 0 = policy) forces a tile capacity out of
 * {384,256,192,128} x 128 / {384,256,128} x 64 with bm_eff (0 = cap) pixels of it used; variant 0: shipped (3-stage ring,
 * LDS-transposed stores), 1: direct stores from the MFMA layout (what CY_CONV_ACCUM launches use), 2: 2-stage ring,
 * 3: four loader waves + eight compute waves.
 * CY_CONV_PIPE=0 in the environment = mode 0. */
int cy_conv_pipe_config(int mode, int cap, int bn, int variant, int bm_eff);
/* Test / tool switch of the slab kernels (CY_CONV_TILE(11..13)).  mode bit 0: the hints select them (0: hints 11-13 fall
 * through to the library default); + 2: the loader / compute variant (4 + 8 waves) instead of the shipped K-split wave pairs;
 * + 4 / + 8: force the 3- / 4-stage weight ring of the K-split kernel (default: 4 stages where the LDS allows).  bm_eff > 0
 * forces the pixels used of the tile capacity (clamped to it).  CY_CONV_SLAB=0 in the environment = mode 0. */
int cy_conv_slab_config(int mode, int bm_eff);

/* Number of rows (bins) of the stats table cy_conv_igemm adds into (16). */
int cy_conv_stats_rows(int M, int OC);
/* With CY_CONV_STATS | CY_CONV_STATS_DET every pixel tile adds into its OWN row (one add per address onto zero: the
 * table, and everything computed from it, is bit-identical from run to run -- fp32 atomics into shared bins are not).
 * Upper bound of the rows the table needs; cy_bn_finalize takes the same number and folds them in a fixed order. */
int cy_conv_stats_rows_det(int M, int OC);
/* Extra rows a partial table needs behind it (0 since the binned-atomics version; kept for ABI stability). */
int cy_bn_scratch_rows(void);

/* Consumer-side BatchNorm for a 1x1 conv (round 6; VERDICT r5 #6 asked for this to be measured: tools/bn_in_micro.py,
 * profiles/r06_consumer_side_bn.txt).  The reference runs conv -> BatchNorm2d -> Mish -> next conv as separate modules
 * (darknet2pytorch.py:166-205, 247-278); this library normally runs the producer's BatchNorm + activation as one pass
 * (cy_bn_act_fwd_fused) that the next conv then reads.  Here the NEXT conv reads the producer's pre-BN rows x [M][Cin] itself,
 * applies act_in(x * in_scale[c] + in_shift[c]) on their way into LDS (in_scale / in_shift: the producer's folded affine, as
 * cy_bn_finalize writes them), writes the activated rows to act_out (the weight gradient of this conv and the backward pass
 * read them) and convolves: out [M][OC] = activated rows x W^T, W = [wrows >= OC][Cin] as cy_pack_weights lays a 1x1 forward
 * matrix out.  flags: 0 or CY_CONV_STATS (BatchNorm statistics of `out` into the CY_STAT_BINS-row table stats_part, as
 * cy_conv_igemm).  16-bit dtypes; (Cin, OC) = (64, 128) or (64, 64) -- the 304 x 304 stage of complex_yolov4.cfg --, anything else
 * CY_ERR_UNSUPPORTED.  The engine does not call it: measured gain below the box spread. */
int cy_conv1x1_bn_in(const void* x, int64_t M, int Cin, int ldx, const float* in_scale, const float* in_shift, int act_in,
                     void* act_out, int ld_act, const void* w, int wrows, void* out, int OC, int ldo, int dtype, int flags,
                     float* stats_part, cy_stream_t s);

/* BatchNorm backward + weight gradient in ONE kernel, for a conv block WITHOUT an input gradient (the first layer: reference
 * darknet2pytorch.py:247-278, module 0 -- autograd never asks for d(loss)/d(image)).  Equivalent to cy_bn_act_bwd_apply_fused
 * (no shortcut operand, dx NOT written) followed by cy_conv_wgrad on the dRaw that pass would have stored: g = dL/d(activated
 * output), raw = the layer's pre-BN tensor, part_bins = the [16][2][Co] table of (sum dz, sum dz * xhat) (cy_bn_act_bwd_reduce or
 * cy_conv_dgrad_bn_sums), mean / invstd / scale / shift = the layer's batch statistics and affine.  Adds gscale * sums into
 * ggamma / gbeta (NULL: skipped), zeroes zero_table (the other table of the alternating pair), and writes `split` slabs
 * part[sp][Co][ks*ks*Ci] like cy_conv_wgrad.  dRaw is formed in registers (rounded to `dtype` as the stored tensor would be) and
 * goes straight into the kernel's LDS tile: 378 MB less written and read again at 608 x 608 x 32 channels, batch 16.
 * 16-bit dtypes, Co <= 32 and a multiple of 8; anything else returns CY_ERR_UNSUPPORTED (the caller keeps the two launches). */
int cy_conv_wgrad_bn(const void* g, int N, int OH, int OW, int Co, int ldg, const void* raw, int ldraw, const void* x, int XH,
                     int XW, int Ci, int ldx, int ks, int stride, int pad, int dtype, const float* mean, const float* invstd,
                     const float* scale, const float* shift, const float* part_bins, int rows, float* ggamma, float* gbeta,
                     float gscale, float* zero_table, int zero_n, int act, float* part, int split, cy_stream_t s);

/* Weight gradient: part[sp][CoRows][ks*ks*Ci] = sum over the pixels of split sp of dy[p][co] * x[p (+) tap][ci].
 * dy: view (N,OH,OW,Co,lddy) ; x: view (N,XH,XW,Ci,ldx).  `split` partial slabs are written (not accumulated);
 * cy_wgrad_reduce folds them into the torch-layout gradient.
 * use_tr: 1 = the default kernels (LDS transpose reads), 0 / 2 = test / A-B variants; + 4 = ATOMIC mode: the pixel range is
 * still cut into `split` blocks per tile, but every block ADDS its tile into slab 0 with fp32 atomics (part must hold one
 * zeroed slab; fold it with split = 1 and cy_reduce_desc.flags bit 0).  No slab traffic proportional to `split`; the
 * sum's rounding depends on arrival order, so the deterministic mode never uses it.  + 8 = tiles of at most 64 x 64 (16-bit
 * kernels): four times the tiles of the default 128 x 128, i.e. the same number of blocks at a quarter of the split. */
int cy_conv_wgrad(const void* dy, int N, int OH, int OW, int Co, int lddy, const void* x, int XH, int XW, int Ci,
                  int ldx, int ks, int stride, int pad, int dtype, float* part, int split, int use_tr, cy_stream_t s);
/* Recommended split for the given problem (fills the chip, bounded slab memory). */
int cy_conv_wgrad_split(int M, int Co, int Ci, int ks);
/* grad[co][ci][kh][kw] (+)= scale * sum_sp part[sp][co][(kh*ks+kw)*CiPad + ci]   (fp32 OIHW, real Co x Ci) */
int cy_wgrad_reduce(const float* part, int split, int CoRows, int CiPad, int ks, int Co, int Ci, float scale,
                    int accumulate, float* grad, cy_stream_t s);

/* BatchNorm2d training statistics from the conv epilogue partials (torch defaults: biased variance for
 * normalisation, unbiased for running_var; reference darknet2pytorch.py:260).
 * Writes mean, invstd, scale = gamma*invstd, shift = beta - mean*scale and updates running stats
 * (momentum) and num_batches_tracked (int64, may be NULL). */
int cy_bn_finalize(const float* stats_part, int rows, int C, int64_t count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift, cy_stream_t s);
/* Eval-mode affine from running statistics. */
int cy_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      int C, float eps, float* scale, float* shift, cy_stream_t s);
/* y = act(x*scale[c] + shift[c]) (+ res)   -- BN apply + Mish/leaky (+ the [shortcut] add,
 * darknet2pytorch.py:208-219) in one pass.  x: (M pixels, C, ldx); y: ldy; res may be NULL. */
int cy_bn_act_fwd(const void* x, int ldx, void* y, int ldy, const void* res, int ldres, int64_t M, int C,
                  const float* scale, const float* shift, int act, int dtype, cy_stream_t s);
/* cy_bn_finalize + cy_bn_act_fwd in ONE launch (the training forward path; cy_bn_finalize stays for the deterministic
 * mode, whose per-tile tables are too long to fold in every block): every block folds the CY_STAT_BINS-row table of its
 * channel group in its prologue, the first pixel block of a group writes vec_out[4][C] = (mean, invstd, scale, shift) and
 * updates the running statistics, and the launch zeroes zero_table[0:zero_n] -- the OTHER table of an alternating pair
 * (the one this kernel reads is left as it is: it is zeroed by the next layer's launch).  stats_ld > 0: the table's rows are
 * stats_ld channels wide and this layer's C channels start at column stats_c0 (ONE conv launch over two sibling layers --
 * the CSP stages' 1x1 pairs -- leaves one table for both; each keeps its own parameters and its own pass); 0: stats_ld = C.
 * vec_ld > 0: vec_out's four rows are vec_ld floats apart (two layers whose output gradients ONE dgrad launch writes -- the
 * producers of a CSP concatenation -- keep their (mean, invstd, scale, shift) side by side in one [4][C1 + C2] block, so that
 * cy_conv_dgrad_bn_sums can take both layers' BatchNorm-backward sums in that launch); 0: vec_ld = C. */
int cy_bn_act_fwd_fused(const void* x, int ldx, void* y, int ldy, const void* res, int ldres, int64_t M, int C,
                        const float* stats_bins, int rows, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* vec_out,
                        float* zero_table, int zero_n, int act, int dtype, int stats_ld, int stats_c0, int vec_ld, cy_stream_t s);
/* cy_bn_bwd_finalize + cy_bn_act_bwd_apply in one launch, same scheme: ggamma / gbeta += gscale * sums by the first pixel
 * block of every channel group.  bins_ld > 0: the rows of part_bins are bins_ld channels wide and this layer's C channels start at
 * column bins_c0 (a table cy_conv_dgrad_bn_sums filled for two layers at once, see cy_bn_act_fwd_fused's vec_ld); 0: bins_ld = C. */
int cy_bn_act_bwd_apply_fused(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, void* res_grad, int ldrg,
                              int res_accum, int64_t M, int C, const float* mean, const float* invstd, const float* scale,
                              const float* shift, const float* part_bins, int rows, float* ggamma, float* gbeta,
                              float gscale, float* zero_table, int zero_n, int act, int dtype, int bins_ld, int bins_c0, cy_stream_t s);
/* Backward pass 1: per-channel partial sums of dz and dz*xhat, dz = dy*act'(x*scale+shift), ADDED (fp32 atomics) into
 * part[row][2][C]; zero on entry, cy_bn_bwd_finalize(rows) folds it and leaves it zeroed.  rows = cy_bn_bwd_rows()
 * (64 bins shared by the blocks) or cy_bn_bwd_rows_det() (one row per block: run-to-run deterministic). */
int cy_bn_act_bwd_reduce(const void* x, int ldx, const void* dy, int lddy, int64_t M, int C, const float* mean,
                         const float* invstd, const float* scale, const float* shift, int act, int dtype,
                         float* part, int rows, cy_stream_t s);
int cy_bn_bwd_rows(int64_t M, int C, int dtype);
int cy_bn_bwd_rows_det(int64_t M, int C, int dtype);
/* Fold the partials: dgamma_sum[C], dbeta_sum[C] (raw sums, used by the apply pass) and accumulate
 * gscale*sums into the parameter gradients ggamma/gbeta (+=). */
int cy_bn_bwd_finalize(const float* part, int rows, int C, float* dgamma_sum, float* dbeta_sum, float* ggamma,
                       float* gbeta, float gscale, cy_stream_t s);
/* Backward pass 2: dx = scale*(dz - dbeta/M - xhat*dgamma/M), written to dx (may alias dy).  When
 * res_grad is non-NULL the shortcut branch gradient dy is added into it (res_accum) or stored. */
int cy_bn_act_bwd_apply(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, void* res_grad,
                        int ldrg, int res_accum, int64_t M, int C, const float* mean, const float* invstd,
                        const float* scale, const float* shift, const float* dgamma_sum, const float* dbeta_sum,
                        int act, int dtype, cy_stream_t s);

/* Data-movement blocks of Darknet.forward (darknet2pytorch.py:180-219, :64-79, :285). */
/* Max pooling (SPP 5/9/13 stride 1, tiny's 2x2 stride 2) as a row pass and a column pass (k + k loads per output, torch's
 * value and tie rule).  `scratch`: >= N*H*OW*C elements of the tensor dtype (forward) / floats (backward).  `argmax`
 * (training; NULL in eval): cy_maxpool_argmax_bytes bytes holding the two passes' tap indices; the backward gathers
 * through the same two stages (no atomics).  dx: stored or accumulated into. */
int64_t cy_maxpool_argmax_bytes(int N, int H, int OH, int OW, int C);
int cy_maxpool_fwd(const void* x, int N, int H, int W, int C, int ldx, void* y, int OH, int OW, int ldy, int k,
                   int stride, int pad, uint8_t* argmax, void* scratch, int dtype, cy_stream_t s);
int cy_maxpool_bwd(const void* dy, int N, int OH, int OW, int C, int lddy, const uint8_t* argmax, void* dx, int H,
                   int W, int lddx, int k, int stride, int pad, int accumulate, float* scratch, int dtype, cy_stream_t s);
int cy_upsample_fwd(const void* x, int N, int H, int W, int C, int ldx, void* y, int ldy, int stride, int dtype,
                    cy_stream_t s);
int cy_upsample_bwd(const void* dy, int N, int H, int W, int C, int lddy, void* dx, int lddx, int stride,
                    int accumulate, int dtype, cy_stream_t s);
/* y (+)= x over a channel slice view (route concat fallback / shortcut without fusion / grad fan-in). */
int cy_slice_copy(const void* x, int ldx, void* y, int ldy, int64_t M, int C, int accumulate, int dtype,
                  cy_stream_t s);
/* y = a + b (shortcut add when it cannot be fused into bn_act). */
int cy_slice_add(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int64_t M, int C, int dtype,
                 cy_stream_t s);
/* fp32 rows [M][C] (ld = C) -> `dtype` view with CPad channels, scaled; used for d(logits) -> head conv backward */
int cy_f32_to_view(const float* x, int64_t M, int C, float scale, const float* scale_dev, void* y, int ldy, int CPad,
                   int dtype, cy_stream_t s);
/* bias gradient of a head conv (C <= 32): gbias[c] += scale * sum_p dlogits[p][c].
 * In both calls the effective factor is scale * (*scale_dev) when scale_dev is non-NULL (the upstream
 * d(loss) scalar stays on the device: no host synchronisation in backward).  deterministic != 0: one block does the
 * whole sum (no cross-block float atomics; slower). */
int cy_bias_grad(const float* dlogits, int64_t M, int C, float scale, const float* scale_dev, float* gbias,
                 int deterministic, cy_stream_t s);

/* Deterministic mode of the same: per-block partial rows into `scratch` (>= 256 * 32 floats), folded in block order by a
 * second one-block launch -- no float atomics, the sum does not depend on scheduling. */
int cy_bias_grad_det(const float* dlogits, int64_t M, int C, float scale, const float* scale_dev, float* gbias,
                     float* scratch, cy_stream_t s);
/* Deterministic mode, first stage of a two-stage fold of a partial-sum table: bins [rows][W] (W = 2 * C for the BatchNorm
 * tables) -> out [rows_out][W], slice k = rows [k RS, (k+1) RS) with RS = ceil(rows / rows_out), summed in row order; the
 * rows read are zeroed.  rows_out = cy_fold_rows_out(rows) (= rows up to 256, else <= 128).  The finalisers
 * (cy_bn_finalize / cy_bn_bwd_finalize) then take (out, rows_out) instead of (bins, rows). */
int cy_fold_rows(float* bins, int rows, int W, float* out, int rows_out, cy_stream_t s);
int cy_fold_rows_out(int rows);

/* ------------------------------------------------------------------------------------------------
 * YOLO head  (reference models/yolo_layer.py)
 * ---------------------------------------------------------------------------------------------- */

/* Decode (yolo_layer.py:144-189): logits NHWC fp32 [B][G][G][A*(7+C)] -> out rows
 * out[b][row_offset + (a*G+gy)*G+gx][0..6+C] of a [B][rows_total][7+C] fp32 tensor.
 * anchors: A*(w,h) in input pixels (host array). */
int cy_yolo_decode(const float* logits, int B, int G, int A, int C, const float* anchors_host, float img_size,
                   float* out, int rows_total, int row_offset, cy_stream_t s);

/* Scratch (private segment) bytes per lane of the per-target loss kernels in the loaded code object: 0 is the
 * precondition for running cy_yolo_loss concurrently with other kernels (side stream); -2 = no device to ask. */
int cy_head_scratch_bytes(void);
/* The heads of one model in ONE sequence of launches (head = blockIdx.y; decode when out != NULL, then the loss): what
 * cy_yolo_decode + cy_yolo_loss do per head, with identical results, at 9 launches per step instead of 8 per head.  heads_host:
 * host array of nheads (<= 3) entries; anchors_host: A * (w, h, im, re) in input pixels.  One workspace for all heads
 * (cy_yolo_loss_multi_workspace bytes).  Replaces the three YoloLayer.forward calls of Darknet.forward (darknet2pytorch.py:218-226). */
typedef struct {
    const float* logits; float* dlogits; float* metrics; const float* anchors_host;
    int G, row_offset;
} cy_head_in;
int64_t cy_yolo_loss_multi_workspace(int nheads, const int* Gs_host, int B, int A, int C, int nT);
int cy_yolo_loss_multi(int nheads, const cy_head_in* heads_host, int B, int A, int C, const float* targets, int nT,
                       float img_size, float ignore_thresh, int use_giou, void* workspace, float* out, int rows_total,
                       cy_stream_t s);
/* The same with the batch's row count ON THE DEVICE: the launches are sized for nT_cap rows (grid, workspace from
 * cy_yolo_loss_multi_workspace(..., nT_cap)), *nT_dev (0 <= *nT_dev <= nT_cap) of them are live; rows beyond are never read.  All
 * arguments then repeat from step to step for every batch whose count falls into the same bucket, so ONE recorded launch list
 * (cy_run_plan) serves them -- the reference's dataloader yields a different number of boxes almost every step
 * (kitti_dataset.py:110-114, collate_fn :216-223).  Results are identical to cy_yolo_loss_multi(..., nT = *nT_dev, ...). */
int cy_yolo_loss_multi_n(int nheads, const cy_head_in* heads_host, int B, int A, int C, const float* targets, int nT_cap,
                         const int32_t* nT_dev, float img_size, float ignore_thresh, int use_giou, void* workspace, float* out,
                         int rows_total, cy_stream_t s);
/* Workspace size in bytes for cy_yolo_loss. */
int64_t cy_yolo_loss_workspace(int B, int G, int A, int C, int nT);
/* Fused build_targets + loss + metrics + d(loss)/d(logits) (yolo_layer.py:69-142, :199-251, and the
 * autograd of both).  targets: device [nT][8] rows (sample, class, x, y, w, l, im, re).
 * anchors_host: A*(w, h, im, re) in input pixels.  use_giou selects the loss variant (:213-218).
 * Outputs: metrics[20] device floats = the 18 entries of YoloLayer.metrics in the reference's key
 * order + [18] = nObj, [19] = error flag; dlogits fp32 [B][G][G][A*(7+C)] = d(total_loss)/d(logits). */
int cy_yolo_loss(const float* logits, int B, int G, int A, int C, const float* targets, int nT,
                 const float* anchors_host, float img_size, float ignore_thresh, int use_giou, void* workspace,
                 float* metrics, float* dlogits, cy_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Rotated-box geometry  (reference utils/iou_rotated_boxes_utils.py, utils/cal_intersection_rotated_boxes.py)
 * ---------------------------------------------------------------------------------------------- */

/* iou_pred_vs_target_boxes (iou_rotated_boxes_utils.py:98-142): n box pairs (x,y,w,l,im,re).
 * giou=1: float32 clip with the reference's control flow (incl. its disjoint-pair behaviour, SURVEY
 * App. A #0) + hull area; giou=0: float64 exact clip.  ious[n], terms[n] (per-pair loss term; the
 * reference returns their sum), gpred[n][6] = d(term)/d(pred) with the reference's partial gradient. */
int cy_riou_pairs(const float* pred, const float* target, int n, int giou, float* ious, float* terms, float* gpred,
                  cy_stream_t s);
/* iou_rotated_boxes_targets_vs_anchors (:64-95): shapes (w,l,im,re) at a common centre, float64 clip.
 * ious[nA][nT]. */
int cy_riou_anchors(const float* anchors_wlir, int nA, const float* targets_wlir, int nT, float* ious,
                    cy_stream_t s);
/* Pairwise rotated IoU matrix, float64 clip, float32 tail (evaluation_utils.py:193-218, :24-40). */
int cy_riou_matrix(const float* a, int na, const float* b, int nb, float eps, float* iou, cy_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Rotated NMS  (reference utils/evaluation_utils.py)
 * ---------------------------------------------------------------------------------------------- */

/* nms_cpu (:250-276): class-agnostic greedy NMS; keep[0..*count) = original indices, highest
 * confidence first.  workspace bytes: cy_rnms_workspace(1, K) (also sizes cy_pp2_merge with B images, Kmax). */
int64_t cy_rnms_workspace(int B, int K);
int cy_rnms_greedy(const float* boxes, const float* confs, int K, float nms_thresh, void* workspace, int32_t* keep,
                   int32_t* count, cy_stream_t s);
/* post_processing_v2 (:321-357), phase 1: per image, select rows with objectness >= conf_thresh,
 * rank them by obj*max(cls) (ties: lower row first).  cand_idx[B][N] (ranked original rows),
 * cand_count[B]. */
int cy_pp2_select(const float* pred, int B, int N, int C, float conf_thresh, void* workspace /* B*N*8 bytes */,
                  int32_t* cand_idx, int32_t* cand_count, cy_stream_t s);
/* phase 2: same-class merge-NMS over the ranked candidates (at most Kmax per image).
 * det[B][Kmax][9] rows (x,y,w,l,im,re,obj,cls_conf,cls_id), det_src[B][Kmax] original row of each
 * detection, det_count[B]. */
int cy_pp2_merge(const float* pred, int B, int N, int C, const int32_t* cand_idx, const int32_t* cand_count,
                 int Kmax, float nms_thresh, void* workspace, float* det, int32_t* det_src, int32_t* det_count,
                 cy_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * LiDAR -> bird's-eye-view rasteriser (reference removePoints + makeBVFeature,
 * src/data_process/kitti_bev_utils.py:18-76, called from kitti_dataset.py:81-82,105-106)
 * ---------------------------------------------------------------------------------------------- */
/* Bytes of zero-initialised device workspace cy_bev_rasterize needs for an H x W map (it leaves it zeroed). */
int64_t cy_bev_workspace(int H, int W);
/* points: n x (x, y, z, intensity) float32 on the device, 16-byte aligned.  Points outside the closed box
 * [minX,maxX] x [minY,maxY] x [minZ,maxZ] are dropped; pixel (xi, yi) = (floor(x / disc), trunc(floor(y / disc) + (W+1)/2)),
 * float32 arithmetic as numpy does it; bins xi == H / yi == W are cropped like the reference's [:H, :W].
 * out: float32 [3][H][W] = intensity of the highest point, its (z - zshift) / max_height, min(1, log(count+1)/log 64);
 * zshift = minZ and max_height = |maxZ - minZ| reproduce removePoints + makeBVFeature on raw points.
 * Equal heights in one pixel: the point that comes first in `points` wins (the reference's stable sort). */
int cy_bev_rasterize(const float* points, int n, float minX, float maxX, float minY, float maxY, float minZ, float maxZ,
                     float zshift, float max_height, float disc, int H, int W, void* workspace, float* out, cy_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Augmentation on the rasterised maps, SURVEY.md section 8f row 1 (reference src/data_process/kitti_dataset.py:123-173
 * load_mosaic; src/data_process/transformation.py:376-437 Horizontal_Flip, Cutout).  Images are float32 [C][H][W] on
 * the device, targets float32 [nT][8] = (sample, class, x, y, w, l, im, re) normalised to the image, updated in place
 * with the reference's float32 arithmetic (bit-identical).  Random draws stay with the caller (host RNG, same stream
 * as the reference); these entry points take the drawn geometry.
 * ---------------------------------------------------------------------------------------------- */
/* 2*img_size square canvas = `fill` (0.5 in the reference) with the four tiles pasted: rects_host[k] = (x1a, y1a, x2a,
 * y2a, x1b, y1b): canvas[y1a:y2a, x1a:x2a] = tile_k[y1b:, x1b:] (kitti_dataset.py:141-157). */
int cy_bev_mosaic(const float* tile0, const float* tile1, const float* tile2, const float* tile3, int C, int h, int w,
                  int img_size, const int* rects_host, float fill, float* out, cy_stream_t s);
/* targets of tile tile_of_target[i]: x = (x*w + padw)/(2 S), y = (y*h + padh)/(2 S), w *= w/(2 S), l *= h/(2 S), then x, y
 * clamped to [0, 1 - 0.5/S]; pads_host[k] = (padw, padh) = (x1a - x1b, y1a - y1b) (kitti_dataset.py:158-171). */
int cy_bev_mosaic_targets(float* targets, int nT, const int32_t* tile_of_target, int h, int w, const int* pads_host,
                          int img_size, cy_stream_t s);
/* out = src flipped along W when flip != 0 (targets: x = 1 - x, im = -im), with nholes <= 8 rectangles holes_host[k] =
 * (y1, y2, x1, x2) set to fill; keep[i] (optional) = 0 for targets whose centre lies inside a hole (borders included).
 * out must not alias src. */
int cy_bev_flip_cutout(const float* src, int C, int H, int W, int flip, const int* holes_host, int nholes, float fill,
                       float* out, float* targets, int nT, uint8_t* keep, cy_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Stream ordering and recorded launch lists (no reference counterpart: the reference's step is eager PyTorch,
 * src/train.py:205-235; this is how the ~650 calls of a train step leave the host in one call).
 * cy_event_*: a hipEvent (timing disabled) as an opaque handle; record on a stream, make another stream wait for it -- the
 * fork / join of the two-stream backward (weight gradients beside the BatchNorm / dgrad chain), as calls of THIS library so
 * that they can be part of a recorded list.
 * cy_run_plan: `prog` = int64 words [fn, nargs, args...] per call (fn from cy_plan_fn_index(entry point name); pointers and
 * integers as themselves, floats as the bits of a double); re-issues the calls in order.  Returns 0, or the first failing
 * call's status with its ordinal in *failed_op.  Only int-returning entry points can be recorded. */
int cy_event_create(void** ev);
int cy_event_destroy(void* ev);
int cy_event_record(void* ev, cy_stream_t s);
int cy_stream_wait_event(cy_stream_t s, void* ev);
int cy_plan_fn_index(const char* name);
int cy_plan_fn_nargs(int fn);
int cy_run_plan(const int64_t* prog, int64_t nwords, int32_t* failed_op);

#ifdef __cplusplus
}
#endif
#endif
