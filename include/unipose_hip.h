/*
 * unipose_hip.h — C ABI of libunipose_hip.so: hand-written gfx950 (MI355X, CDNA4) kernels for the
 * UniPose / UniPose-LSTM forward+backward hot path.
 *
 * The reference (bmartacho/UniPose) has NO native / FFI layer: its arithmetic is PyTorch ATen ops
 * reached from Python nn.Modules.  Each entry point below therefore cites the
 * reference call site(s) whose ATen op it replaces (paths relative to the reference tree).  The
 * Python binding a maintainer adds is the ctypes table in unipose_amd/_C.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 unless stated), borrowed for the call, never freed;
 *   - activations are NHWC ("pixel-major"): element (n,h,w,c) of a tensor with pixel stride `ld`
 *     lives at ((n*H + h)*W + w)*ld + c; ld >= C lets a producer write into a channel slice of a
 *     wider buffer (this is how torch.cat is eliminated: wasp.py:84, decoder.py:51);
 *   - channel counts seen by the convolution kernels are padded to a multiple of 4 (`Cp`), pad
 *     channels hold zeros; pixel strides and base pointers are multiples of 4 floats (16 B);
 *   - `stream` is a hipStream_t (as void*); all work is enqueued asynchronously on it;
 *   - return value: 0 on success, negative up_status on failure; up_last_error() gives the text.
 *     No C++ exception crosses this boundary.
 *   - host-side global state: the per-thread error string; the development knobs of up_conv_tune (and the UP_* environment
 *     variables read at load time), which select kernels process-wide; per-stream K-split scratch (up_stream_release) and
 *     per-geometry device tables (tap-sort permutations, weight-gradient rectangles) allocated on first use and kept.
 */
#ifndef UNIPOSE_HIP_H
#define UNIPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    UP_OK = 0,
    UP_ERR_INVALID = -1,     /* bad argument (shape, alignment, null pointer) */
    UP_ERR_UNSUPPORTED = -2, /* configuration this library does not implement */
    UP_ERR_LAUNCH = -3,      /* HIP launch / runtime error */
    UP_ERR_WORKSPACE = -4    /* caller-provided workspace too small */
} up_status;

const char* up_last_error(void);
int up_abi_version(void);   /* 10 */

/* Geometry of one 2-D convolution (nn.Conv2d as used at resnet.py:10-16,61,80-84,104-109;
 * wasp.py:9,52,59-60; decoder.py:17,22,26,30; model/uniposeLSTM.py:12-14,30-38,85-89). */
typedef struct {
    int32_t N, H, W;      /* input batch / height / width                                   */
    int32_t C, Cp;        /* real input channels, padded input channels (Cp%4==0, Cp>=C)    */
    int32_t ldx;          /* input pixel stride in floats (>= Cp, %4==0)                     */
    int32_t K;            /* output channels                                                 */
    int32_t R, S;         /* kernel height / width                                           */
    int32_t stride, pad, dil;
    int32_t P, Q;         /* output height / width                                           */
    int32_t ldy;          /* output pixel stride in floats (>= K)                            */
    int32_t Kp;           /* padded output channels for the dgrad weight image (Kp%4==0)     */
} up_conv_desc;

/* Fused epilogue of the forward convolution (all optional, applied in this order):
 *   v = acc;  if(scale) v = v*scale[k] + shift[k];   (folded eval-mode BatchNorm, K7)
 *   if(bias) v += bias[k];  if(residual) v += residual[pixel*ldr + k];  if(relu) v = max(v,0)
 * `stats` != NULL asks for per-(row-tile, channel) Welford partials {count, mean, M2} of the RAW
 * accumulator (train-mode BatchNorm statistics, K7); it excludes scale/bias/residual/relu. */
/* ABI 10: BatchNorm finalize FOLDED into the launch that writes `stats`.  Every workgroup takes a ticket after publishing its
 * partial row; the last arriver merges the partials in a fixed order (deterministic; the same bits as up_bn_finalize on the same
 * stats) and writes mean / invstd / scale / shift (+ the running-statistics update) — no up_bn_finalize launch on the critical
 * path.  `folded` is set by the call: 1 = done by the launch, 0 = the caller runs up_bn_finalize (row groups, fold switched off
 * with up_conv_tune("bn_fold", 0), scratch exhausted).  Scratch (tickets + level-1 rows, ~8 MB) is per stream, like the K-split's. */
typedef struct {
    float eps, momentum;
    float* running_mean;   /* optional pair, updated in place */
    float* running_var;
    const float* gamma;
    const float* beta;
    float* mean;           /* out [K] */
    float* invstd;
    float* scale;
    float* shift;
    int32_t folded;        /* out */
} up_bn_fold;
typedef struct {
    const float* scale;
    const float* shift;
    const float* bias;
    const float* residual;
    int32_t ldr;
    int32_t relu;
    float* stats;         /* [up_conv_stats_tiles(desc)][K][3] or NULL */
    up_bn_fold* fold;     /* optional (ABI 10, needs stats): see above */
} up_conv_epilogue;

/* Re-lay an OIHW fp32 weight (PyTorch layout) for the implicit-GEMM kernels:
 *   w_fwd  [K][R*S][Cp]  (forward, B operand rows = output channel, k-contiguous)
 *   w_dgrad[C ][R*S][Kp] (data-gradient: rows = input channel)           — either may be NULL.
 * The data-gradient image is opaque: for stride-2, dilation-1 convolutions it holds four parity-class sub-images
 * (same total size); only up_conv2d_bwd_data with the SAME descriptor may read it. */
int up_pack_weights(const up_conv_desc* d, const float* w_oihw, float* w_fwd, float* w_dgrad, void* stream);
/* The same for many parameters in ONE launch (every weight changes once per optimizer step): `jobs` is a table in
 * DEVICE memory; w_fwd / w_dgrad may be NULL per job. */
typedef struct {
    const float* w;        /* OIHW */
    float* w_fwd;
    float* w_dgrad;
    int32_t K, C, Cp, Kp, taps;
    int32_t geometry;      /* stride | R << 4 | S << 10 | pad << 16 | dilation << 24: selects the data-gradient image layout
                              (stride-2 convolutions use a parity-class-major one, like up_pack_weights does) */
} up_pack_job;
int up_pack_weights_batched(const up_pack_job* jobs_device, int njobs, void* stream);

/* Forward convolution, implicit GEMM on v_mfma_f32_32x32x2_f32 (replaces aten::convolution, K1-K6,K17). */
int up_conv2d_fwd(const up_conv_desc* d, const float* x, const float* w_fwd, float* y,
                  const up_conv_epilogue* ep, void* stream);
int up_conv_stats_tiles(const up_conv_desc* d);   /* row tiles the fp32 forward kernel will use */
int up_conv_stats_tiles_math(const up_conv_desc* d, int math);   /* ... the forward kernel of arithmetic `math` (up_math) */
/* Load balance: when the tile count leaves a short tail (tiles % CUs small), the forward / data-gradient launch splits
 * each tail tile along K into this many parts (1 = no split), one per CU, and merges them in a fixed order through a
 * per-stream scratch the library allocates on first use (one 128x128 fp32 partial per CU: 16 MB + flags).  UP_TAIL_SPLIT=0 disables it.  Informational. */
int up_conv_split_parts(const up_conv_desc* d);
/* Frees the per-stream scratch of a stream the caller retires (no launch or captured graph of that stream may run afterwards);
 * a stream that never ran a split launch has none: no-op. */
int up_stream_release(void* stream);
/* ABI 10, experiment support: a HIP stream whose kernels run only on the compute units named by `mask` (bit i of `words` 32-bit
 * words = logical CU i; hipExtStreamCreateWithCUMask) — the weight-gradient side stream on a fixed share of the chip
 * (UNIPOSE_SIDE_CUS, profiles/r06_experiments.txt) — and a probe that reports where the workgroups of a stream run:
 * out[blocks][2] = (XCC id, CU | SH << 4 | SE << 5 of HW_ID), device memory. */
int up_stream_create_cu_mask(const uint32_t* mask, int words, void** stream);
int up_stream_destroy(void* stream);
int up_probe_placement(int blocks, int* out_device, void* stream);
/* Development knobs (A/B runs inside one process; each also has an environment variable read at load time).  Twelve keys + five of round 6
 * (round 4 removed the ones whose question is settled: short_k, short_k_mult, db_min_k, wgrad_per_cu, tap_skip, lds_swz and
 * the bf16 forms that lost in round 3):
 * "tiny_k" (UP_TINY_K, round 5: fp32 reductions of at most this length always take 64x64 tiles; default 128),
 * "tile_want" (UP_TILE_WANT; "tile_want_bf16" for the plain-bf16 kernels) workgroups a launch should at least have when the tile
 * size is chosen, "tail_split" (UP_TAIL_SPLIT), "split_per_cu" (UP_SPLIT_PER_CU: launches with fewer tiles than CUs split every tile
 * along K up to this many workgroups per CU), "tap_sort" (UP_TAP_SORT: GEMM rows ordered by their set of live filter taps so that
 * the tile-level tap skipping becomes near exact), "wgrad_rect" (UP_WGRAD_RECT, see up_conv_wgrad_visits).
 * bf16 storage: "glds" (UP_GLDS: direct-to-LDS kernels of bf16s_glds.h, default 1; 0 = the register-staged kernels), "bn_rows"
 * (row-strided BatchNorm kernels).  Round 6, the 8-wave (32 TM) x 256 tiles of bf16s_big.h: "glds_big" (UP_GLDS_BIG, default 1: launches
 * with N % 256 == 0, 64-aligned channels and a reduction of at least "big_min_k" (UP_BIG_MIN_K, 1024) run on igemm_big_kernel),
 * "big_stages" (UP_BIG_STAGES: LDS stages of the 160-row tiles, 3 | 2), "big_rows" (UP_BIG_ROWS: 0 = rows per tile by the fill
 * rule, else only 160 / 192 / 256), "big_dgrad" (UP_BIG_DGRAD: 0 keeps data gradients on igemm_glds_kernel); "stem7" (UP_STEM7, default 0:
 * the 7x7 stride-2 first convolution on stem7_kernel of stem_f32.h — equal bits, faster alone, not faster inside the step).  fp32 (round 4): "glds32" (UP_GLDS32: forward / data gradient on f32_glds.h, default 1),
 * "glds32_epi" (LDS-transposed 16-byte-store epilogue, 1), "glds32_wgrad" (weight gradient on f32_glds.h, 1).  "cu_count" (tests: pretend the chip has this many CUs when planning
 * splits; 0 = the real count).
 * These knobs and the UP_* environment variables they mirror are PROCESS-GLOBAL host state (kernel selection of every later
 * launch on every stream), like the library's per-stream K-split scratch and per-geometry tables; see the note at the top.
 * Change them only between steps: workspace sizes and the BatchNorm partial-row count follow the tile choice. */
int up_conv_tune(const char* key, int value);
/* Diagnostics: fp32 forward / data-gradient launches since load, by kernel family ("big": bf16-storage launches on igemm_big_kernel) — "igemm" (register-staged igemm_kernel),
 * "glds32" (f32_glds.h), "glds32_epi1" (of those, with the LDS-transposed epilogue), "glds32_bnred" (with the fused
 * BatchNorm-backward reduction), "wgrad_glds32" / "wgrad_glds32_st1" (fp32 weight-gradient launches on the direct-to-LDS kernel /
 * of those, the one-stage form); -1 for an unknown name.  Tests use it to prove which kernel a case ran on. */
long long up_conv_counter(const char* name);
/* Analysis (host only, no launch): share of (row tile, filter tap) pairs the K loop of the forward (data_gradient = 0) or
 * data-gradient launch of `d` visits with its rows in image order and in tap-sorted order ("tap_sort" knob), and the share
 * of (pixel, tap) pairs that touch the image at all (`live`: what a perfect skip would visit).  Aligned fast path only. */
int up_conv_tap_visits(const up_conv_desc* d, int data_gradient, double* image_order, double* tap_sorted, double* live);
/* The same for the weight gradient with the "wgrad_rect" knob (UP_WGRAD_RECT, default off until measured): each column
 * tile of the weight-gradient GEMM reduces over the bounding rectangle of the output pixels on which its filter taps read
 * real input (instead of all N*P*Q pixels); *rect_fraction = share of (pixel, column tile) pairs still visited. */
int up_conv_wgrad_visits(const up_conv_desc* d, double* rect_fraction);

/* Data gradient: dx[N,H,W,ldx(:Cp)] from dy[N,P,Q,ldy(:K)] (replaces convolution_backward, input half).
 * Writes all Cp channels of every input pixel (pad channels get 0).  `add` (optional, [N,H,W,ld_add]) is a second
 * gradient of the same input — the identity branch of a residual block, resnet.py:36-40 — summed in the epilogue
 * instead of by a separate add kernel. */
int up_conv2d_bwd_data(const up_conv_desc* d, const float* dy, const float* w_dgrad, float* dx,
                       const float* add, int ld_add, void* stream);
/* Extended data-gradient epilogue (round 4; fp32 and bf16 storage).  Two things a residual network's backward otherwise pays
 * separate full-tensor passes for (resnet.py:25-42):
 *  - `bn`: dx is dz of the layer z = relu(bn(y) (+ res)) that produced this convolution's input (bn1 -> relu -> conv2,
 *    bn2 -> relu -> conv3, a block's output -> the next identity block's conv1): the launch also reduces that layer's two
 *    BatchNorm-backward sums per row tile — partial[tile][c] = {sum g, invstd[c] * sum g * (y - mean[c])}, g = dz * [z > 0] —
 *    so the layer's backward (up_bn_bwd_prereduced_t) needs no reduction pass (native_batch_norm_backward's first read of dz, y);
 *  - `add_relu_bits`: the addend (the skip-connection gradient, resnet.py:36-40) is the UNMASKED dz of the block's last layer
 *    and its ReLU mask is applied here (bit row * C + c of that layer's sign bits), so that layer's backward need not write
 *    dz * [z > 0] as a tensor of its own (threshold_backward's output).
 * up_conv2d_bwd_data_tiles_math(d, math) = rows of `partial`, and > 0 exactly when the launch of `d` runs on a kernel that
 * supports the two extras (math = UP_MATH_F32 or UP_MATH_BF16S); with 0 use the plain entry points (add only). */
/* Row groups (ABI 8): `groups` equal batches stacked along N that must keep separate BatchNorm statistics (the frames of the video
 * model's batched trunk).  Every group is tiled on its own, so no row tile straddles two groups and stats is
 * [groups][up_conv_stats_tiles_grouped(d, groups)][K][3] — the layout up_bn_finalize_groups merges; no extra pass over y
 * (up_bn_batch_stats_t) is needed.  fp32 only, direct-to-LDS kernel with the LDS-transposed epilogue: up_conv_stats_tiles_grouped
 * returns 0 and up_conv2d_fwd_grouped UP_ERR_UNSUPPORTED (nothing launched) otherwise. */
/* ABI 10 (row groups): the statistics pass and the finalize as one launch where the fold applies (else the two launches);
 * and the grouped backward whose sums the data gradient already merged (up_bn_reduce_slot.gsum, folded = 1): the apply pass alone. */
int up_bn_stats_groups_t(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats, float eps,
                         float momentum, float* running_mean, float* running_var, const float* gamma, const float* beta, float* coef,
                         void* stream);
int up_bn_bwd_groups_finalized_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                                 const float* coef, int relu, void* dy, int lddy, void* dres, int lddres, const float* gsum,
                                 int64_t rows_per_group, int C, int groups, int dtype, void* stream);
int up_bn_bwd_groups_prereduced_ok(int64_t rows_per_group, int C, int groups, int ld);
int up_bn_bwd_groups_prereduced_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                                  const float* coef, int relu, void* dy, int lddy, void* dres, int lddres, float* dgamma, float* dbeta,
                                  float* workspace, size_t workspace_bytes, const float* partial, int tiles, int64_t rows_per_group,
                                  int C, int groups, int dtype, void* stream);
int up_conv_stats_tiles_grouped(const up_conv_desc* d, int groups);
int up_conv2d_fwd_grouped(const up_conv_desc* d, const float* x, const float* w_fwd, float* y, float* stats, int groups, void* stream);

typedef struct {
    const void* y;             /* raw convolution output of the producing layer, [rows][ld] (element type of dx)  */
    const uint32_t* relu_bits; /* sign bits of z (bit row * C + c), NULL when the layer has no ReLU               */
    const float* mean;         /* [C] batch (or running) mean / 1 / sqrt(var + eps) of that BatchNorm              */
    const float* invstd;
    float* partial;            /* out: [up_conv2d_bwd_data_tiles_math(d, math)][C][2]                               */
    int32_t ld, C;             /* pixel stride of y; channels (== d->C of this convolution)                        */
    int32_t group_stride;      /* row groups (ABI 8): floats between two groups' mean / invstd vectors, else 0     */
    float* dgamma;             /* ABI 10, optional pair: the launch also FINISHES the reduction (last-arriver ticket, see      */
    float* dbeta;              /* up_bn_fold): [C] sums over all rows; then up_bn_bwd_finalized_t runs the apply pass alone   */
    float* gsum;               /* row groups (ep->groups > 1): [groups][dgamma | dbeta][C], every group's own sums (its data     */
                               /* gradient needs them: up_bn_bwd_groups_finalized_t); NULL: a grouped launch does not fold        */
    int32_t folded;            /* out: 1 = dgamma / dbeta (/ gsum) are final; 0 = call up_bn_bwd_(groups_)prereduced_t on `partial` */
} up_bn_reduce_slot;
typedef struct {
    const void* add;               /* second gradient of the same input (element type of dx), or NULL              */
    const uint32_t* add_relu_bits; /* optional ReLU mask of the addend, see above                                   */
    up_bn_reduce_slot* bn;         /* optional fused BatchNorm-backward reduction                                   */
    int32_t ld_add;
    int32_t groups;                /* row groups (ABI 8; 0 / 1: none): the N images are `groups` equal batches, every one tiled
                                      on its own; bn->partial is then [groups][up_conv2d_bwd_data_tiles_grouped][C][2] and
                                      bn->mean / invstd are per group (group_stride).  fp32 only.                       */
} up_dgrad_epilogue;
int up_conv2d_bwd_data_tiles_grouped(const up_conv_desc* d, int groups);   /* tiles per group, 0: not supported */
int up_conv2d_bwd_data_tiles(const up_conv_desc* d);                 /* = ..._tiles_math(d, UP_MATH_F32) */
int up_conv2d_bwd_data_tiles_math(const up_conv_desc* d, int math);
/* w_dgrad: the fp32 data-gradient image of up_pack_weights (math = UP_MATH_F32) or the bf16 `hi` plane of
 * up_pack_weights_bf16 (UP_MATH_BF16S: dy, dx, add and bn->y are bf16). */
int up_conv2d_bwd_data_ex(const up_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const up_dgrad_epilogue* ep,
                          int math, void* stream);

/* bf16-operand variants of the forward / data-gradient convolution (v_mfma_f32_32x32x16_bf16, fp32 accumulate,
 * fp32 activations in HBM).  math = UP_MATH_BF16X3: every operand is carried as hi = bf16(x), lo = bf16(x - hi)
 * and a*b ~= ah*bh + ah*bl + al*bh (fp32-equivalent, relative error 2^-16 per product);  UP_MATH_BF16: plain bf16
 * operands (BASELINE config 5 arithmetic).  Weights come as bf16 planes made by up_pack_weights_bf16 in the
 * [rows][R*S][padded channels] order of up_pack_weights.  Requires the padded reduction channel count (Cp forward,
 * Kp backward) to be a multiple of 32; otherwise UP_ERR_UNSUPPORTED (use the fp32 entry points). */
typedef enum { UP_MATH_F32 = 0, UP_MATH_BF16X3 = 1, UP_MATH_BF16 = 2, UP_MATH_BF16S = 3, UP_MATH_BF16S_F32OUT = 4 } up_math;
int up_pack_weights_bf16(const up_conv_desc* d, const float* w_oihw, uint16_t* fwd_hi, uint16_t* fwd_lo,
                         uint16_t* dgrad_hi, uint16_t* dgrad_lo, void* stream);
/* The same for many parameters in ONE launch (ABI 9; like up_pack_weights_batched: an optimizer step changes every weight, and
 * 2 x 115 separate 5-us launches per step cost more than the packing itself).  `jobs` is a table in DEVICE memory; the
 * fwd / dgrad plane pairs may be NULL per job, and so may the lo plane of a pair alone (only UP_MATH_BF16X3 reads lo planes). */
typedef struct {
    const float* w;        /* OIHW */
    uint16_t* fwd_hi;
    uint16_t* fwd_lo;
    uint16_t* dgrad_hi;
    uint16_t* dgrad_lo;
    int32_t K, C, Cp, Kp, taps, reserved;
} up_pack_job_bf16;
int up_pack_weights_bf16_batched(const up_pack_job_bf16* jobs_device, int njobs, void* stream);
int up_conv2d_fwd_bf16(const up_conv_desc* d, const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                       const up_conv_epilogue* ep, int math, void* stream);
int up_conv2d_bwd_data_bf16(const up_conv_desc* d, const float* dy, const uint16_t* w_hi, const uint16_t* w_lo,
                            float* dx, const float* add, int ld_add, int math, void* stream);

/* Weight gradient into PyTorch OIHW layout (replaces convolution_backward, weight half); `dbias`
 * (K floats) may be NULL.  Split-K partial slabs live in the caller-provided workspace. */
size_t up_conv2d_bwd_weight_workspace(const up_conv_desc* d);
int up_conv2d_bwd_weight(const up_conv_desc* d, const float* x, const float* dy, float* dw_oihw,
                         float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* The same on v_mfma_f32_32x32x16_bf16 (BASELINE configs[4] arithmetic: operands rounded to bf16 while they are staged,
 * fp32 accumulation, fp32 split-K slabs and result; same workspace).  Falls back to the fp32 kernel for tensors of more
 * than 2^31 elements. */
int up_conv2d_bwd_weight_bf16(const up_conv_desc* d, const float* x, const float* dy, float* dw_oihw,
                              float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* ---- bf16 STORAGE (BASELINE configs[4]: "bf16, 736x736, batch 16/GPU") -------------------------------------------------
 * Activations and activation gradients live in HBM as bf16 (raw 16-bit patterns, NHWC, channel counts and pixel strides
 * multiples of 8 = 16-byte channel groups; convolution outputs are padded to multiples of 32 channels so that every
 * reduction is a whole number of 32-wide K slices); all arithmetic is fp32 (BatchNorm statistics, accumulators) or
 * bf16 MFMA with fp32 accumulation; weights, weight gradients, BatchNorm parameters and the optimizer stay fp32.
 *   - convolutions: up_conv2d_fwd_bf16 / up_conv2d_bwd_data_bf16 with math = UP_MATH_BF16S — x / y / residual / add are
 *     then bf16 tensors passed through the same pointer arguments; up_conv2d_bwd_weight_bf16s below.  up_conv2d_fwd_bf16 also
 *     takes math = UP_MATH_BF16S_F32OUT: bf16 x, fp32 y (no residual / statistics) — the network's LAST convolution
 *     (decoder.py:30), so that the heat-maps and the bilinear up-sampling behind them (model/unipose.py:31-32) are not
 *     rounded to 8-bit mantissas;
 *   - every streaming operator has a `_t` twin taking the element type of its activation tensors (UP_DT_F32 / UP_DT_BF16);
 *     the fp32 entry points above are the `_t` forms with UP_DT_F32.  The max-pool takes two types: it is where the
 *     network leaves the fp32 stem (fp32 in, bf16 out; its backward bf16 in, fp32 out). */
enum { UP_DT_F32 = 0, UP_DT_BF16 = 1 };
int up_conv2d_bwd_weight_bf16s(const up_conv_desc* d, const void* x_bf16, const void* dy_bf16, float* dw_oihw,
                               float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* The three weight-gradient entry points above behind one signature, plus accumulation: `math` = UP_MATH_F32 (fp32 tensors,
 * fp32 MFMA), UP_MATH_BF16 (fp32 tensors, bf16 operands) or UP_MATH_BF16S (bf16 tensors); accumulate != 0 ADDS the result to
 * dw_oihw / dbias instead of overwriting them (dw += sum of the split-K slabs, in one rounding step like a separate add).
 * A weight that is used several times per backward pass — every weight of the video model, once per frame
 * (uniposeLSTM.py:116-133) — is then summed by the reduce pass itself: no extra buffer, no add kernel per use. */
int up_conv2d_bwd_weight_acc(const up_conv_desc* d, const void* x, const void* dy, float* dw_oihw, float* dbias,
                             void* workspace, size_t workspace_bytes, int math, int accumulate, void* stream);

/* ---- BatchNorm (nn.BatchNorm2d, K7; every bnX site, e.g. resnet.py:11,14,16; wasp.py:11,53,61) ---- */
/* eval: scale = g/sqrt(rv+eps), shift = b - rm*scale */
int up_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* train: merge the conv epilogue partials -> batch mean / invstd, update running stats in place
 * (momentum, unbiased variance), emit scale/shift for up_bn_apply and mean/invstd for backward. */
int up_bn_finalize(const float* stats, int tiles, int C, float eps, float momentum,
                   float* running_mean, float* running_var,
                   const float* gamma, const float* beta,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);
/* z = relu?(y*scale[c] + shift[c] (+ residual)).  relu_bits (optional, (rows*C + 31)/32 words): dense bit array,
 * bit row*C + c = (z > 0) — the backward passes read it instead of z (1/32 of the bytes). */
int up_bn_apply(const float* y, int ldy, const float* scale, const float* shift,
                const float* residual, int ldr, int relu, float* z, int ldz, uint32_t* relu_bits,
                int64_t rows, int C, void* stream);
/* backward of z = relu?(bn(y) (+res)):  g = dz * (z>0 if relu);  dgamma = sum g*xhat, dbeta = sum g,
 * dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)  (train)   or   gamma*invstd*g (eval: use_batch_stats=0);
 * dres (optional) = g.  The ReLU mask comes from relu_bits when given (z may then be NULL), else from z. */
int up_bn_bwd(const float* dz, int lddz, const float* z, int ldz, const uint32_t* relu_bits,
              const float* y, int ldy,
              const float* gamma, const float* mean, const float* invstd, int relu, int use_batch_stats,
              float* dy, int lddy, float* dres, int lddres, float* dgamma, float* dbeta,
              float* workspace, size_t workspace_bytes, int64_t rows, int C, void* stream);
size_t up_bn_bwd_workspace(int64_t rows, int C);
int up_bn_apply_t(const void* y, int ldy, const float* scale, const float* shift, const void* residual, int ldr,
                  int relu, void* z, int ldz, uint32_t* relu_bits, int64_t rows, int C, int dtype, void* stream);
/* The same pass in the CENTRED form ATen evaluates (native_batch_norm: (x - mean) * invstd * weight + bias):
 * z = relu?((y - mean[c]) * scale[c] + beta[c] (+ residual)), scale = gamma * invstd from up_bn_finalize, beta = the BatchNorm bias.
 * y * scale + (beta - mean * scale) rounds the product mean * scale: an absolute error of 2^-24 |mean| * scale on z, i.e. a relative
 * |mean| / std ulps — 364 ulps for the reference's BatchNorm behind the global-average-pool branch (wasp.py:53: four samples per
 * channel at B = 4), enough to flip ReLU decisions the reference does not flip: the WASP gradients of G14 moved from 2.7x to within
 * the reference's own fp32-vs-fp64 distance with this form.  ABI version 6. */
int up_bn_apply_centered_t(const void* y, int ldy, const float* mean, const float* scale, const float* beta, const void* residual,
                           int ldr, int relu, void* z, int ldz, uint32_t* relu_bits, int64_t rows, int C, int dtype, void* stream);
int up_bn_bwd_t(const void* dz, int lddz, const void* z, int ldz, const uint32_t* relu_bits, const void* y, int ldy,
                const float* gamma, const float* mean, const float* invstd, int relu, int use_batch_stats,
                void* dy, int lddy, void* dres, int lddres, float* dgamma, float* dbeta,
                float* workspace, size_t workspace_bytes, int64_t rows, int C, int dtype, void* stream);
/* ... and with running sums over several calls: acc_dgamma[c] += this call's dgamma[c] (likewise dbeta), for a BatchNorm that
 * runs once per frame of the video unroll (uniposeLSTM.py:116-133); dgamma / dbeta still receive this call's sums. */
int up_bn_bwd_acc_t(const void* dz, int lddz, const void* z, int ldz, const uint32_t* relu_bits, const void* y, int ldy,
                const float* gamma, const float* mean, const float* invstd, int relu, int use_batch_stats,
                void* dy, int lddy, void* dres, int lddres, float* dgamma, float* dbeta, float* acc_dgamma, float* acc_dbeta,
                float* workspace, size_t workspace_bytes, int64_t rows, int C, int dtype, void* stream);
/* BatchNorm backward whose reduction pass ran inside the data-gradient launch that produced dz (up_conv2d_bwd_data_ex):
 * `partial` = that launch's [chunks][C][2] sums; finalize + apply only. */
int up_bn_bwd_prereduced_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                           const float* mean, const float* invstd, int relu, int use_batch_stats, void* dy, int lddy, void* dres,
                           int lddres, float* dgamma, float* dbeta, float* acc_dgamma, float* acc_dbeta, float* partial, int chunks,
                           int64_t rows, int C, int dtype, void* stream);
/* ABI 10: ... and when that launch also carried the merge (up_bn_reduce_slot.dgamma / dbeta, folded = 1): dgamma / dbeta are
 * final, the apply pass alone (bn_bwd_apply of native_batch_norm_backward). */
int up_bn_bwd_finalized_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                          const float* mean, const float* invstd, int relu, int use_batch_stats, void* dy, int lddy, void* dres,
                          int lddres, const float* dgamma, const float* dbeta, int64_t rows, int C, int dtype, void* stream);

/* ---- grouped BatchNorm: one tensor holds `groups` row groups of equal size (the T frames of a clip batch, frame-major), each
 * normalised with ITS OWN batch statistics — what `groups` separate module calls do in the reference's video loop
 * (uniposeLSTM.py:116-133 runs the trunk once per frame), while every convolution sees one T-times larger batch.
 *   up_bn_batch_stats_t      partial (count, mean, M2) per (group, 256-row chunk, channel): stats[groups][tiles][C][3]
 *   up_bn_finalize_groups    coef[groups][4][C] = mean, invstd, scale, shift; running statistics updated group by group
 *   up_bn_apply_groups_t     z = relu((y - mean_g) * scale_g + beta (+ res)) per group (beta = NULL: y * scale_g + shift_g), relu_bits
 *                            as in up_bn_apply
 *   up_bn_bwd_groups_t       data gradient per group from that group's sums; dgamma / dbeta = sums over all groups */
/* Statistics of SMALL batches (<= 4096 rows per group), float64 two-pass, one tile per group: stats[groups][1][C][3] for
 * up_bn_finalize (tiles = 1) / up_bn_finalize_groups.  The host side uses it below 256 rows per channel — the BatchNorm behind the
 * global-average-pool branch (wasp.py:53), B rows — where the mean must be the correctly rounded one (see the kernel). */
int up_bn_exact_stats_t(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats, void* stream);
int up_bn_batch_stats_tiles(int64_t rows_per_group);
int up_bn_batch_stats_t(const void* y, int ldy, int64_t rows_per_group, int C, int groups, int dtype, float* stats, void* stream);
int up_bn_finalize_groups(const float* stats, int tiles, int C, int groups, int64_t rows_per_group, float eps, float momentum,
                          float* running_mean, float* running_var, const float* gamma, const float* beta, float* coef,
                          void* stream);
int up_bn_apply_groups_t(const void* y, int ldy, const float* coef, const float* beta, const void* res, int ldr, int relu, void* z,
                         int ldz, uint32_t* relu_bits, int64_t rows_per_group, int C, int groups, int dtype, void* stream);
size_t up_bn_bwd_groups_workspace(int64_t rows_per_group, int C, int groups);
int up_bn_bwd_groups_t(const void* dz, int lddz, const uint32_t* relu_bits, const void* y, int ldy, const float* gamma,
                       const float* coef, int relu, void* dy, int lddy, void* dres, int lddres, float* dgamma, float* dbeta,
                       float* workspace, size_t workspace_bytes, int64_t rows_per_group, int C, int groups, int dtype,
                       void* stream);

/* ---- pointwise / data movement ---- */
int up_relu_bwd(const float* dz, const float* z, float* dx, int64_t n, void* stream);              /* K8 */
int up_copy2d(const float* src, int lds, float* dst, int ldd, int64_t rows, int C, void* stream);   /* K13 */
int up_add2d(const float* a, int lda, const float* b, int ldb, float* dst, int ldd, int64_t rows, int C, void* stream);
int up_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int ldy, void* stream);   /* boundary */
int up_nhwc_to_nchw(const float* x, int ldx, float* y, int N, int C, int H, int W, void* stream);
/* nn.MaxPool2d(3,2,1) (resnet.py:65, decoder.py:33): idx (uint8, 0..8 = window tap of the first max). */
int up_maxpool3s2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx,
                      int N, int H, int W, int C, int P, int Q, void* stream);
int up_maxpool3s2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx,
                      int N, int H, int W, int C, int P, int Q, void* stream);
/* F.interpolate(mode='bilinear', align_corners=True) (wasp.py:83, decoder.py:49, model/unipose.py:32) */
int up_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int P, int Q, void* stream);
int up_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int P, int Q, void* stream);
/* nn.AdaptiveAvgPool2d(1) (wasp.py:51) */
int up_gap_fwd(const float* x, int ldx, float* y, int N, int HW, int C, void* stream);
int up_gap_bwd(const float* dy, float* dx, int lddx, int N, int HW, int C, void* stream);
/* nn.AvgPool2d(9,8,1) on the 1-channel centre map (model/uniposeLSTM.py:75,114); NCHW(1ch) in,
 * writes channel `coff` of an NHWC buffer with pixel stride ldy */
int up_avgpool9s8_fwd(const float* x, float* y, int ldy, int coff, int N, int H, int W, int P, int Q, void* stream);
/* nn.Dropout (wasp.py:63, decoder.py:25,29): keep-mask from a counter hash of (seed, element index)
 * or, when ext_mask != NULL, from the caller (float 0/1) — the injectable-RNG hook used for parity. */
int up_dropout_fwd(const float* x, float* y, uint8_t* mask, const float* ext_mask, int64_t n,
                   float p, uint64_t seed, void* stream);
int up_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream);
/* `_t` twins of the data-movement operators (see "bf16 STORAGE" above) */
int up_nchw_to_nhwc_t(const float* x, void* y, int N, int C, int H, int W, int ldy, int dt_out, void* stream);
int up_nhwc_to_nchw_t(const void* x, int ldx, float* y, int N, int C, int H, int W, int dt_in, void* stream);
int up_maxpool3s2_fwd_t(const void* x, int ldx, void* y, int ldy, uint8_t* idx, int N, int H, int W, int C, int P, int Q,
                        int dt_in, int dt_out, void* stream);
int up_maxpool3s2_bwd_t(const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, int N, int H, int W, int C, int P,
                        int Q, int dt_in, int dt_out, void* stream);
int up_bilinear_fwd_t(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int P, int Q, int dtype,
                      void* stream);
int up_bilinear_bwd_t(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, int P, int Q, int dtype,
                      void* stream);
int up_gap_fwd_t(const void* x, int ldx, void* y, int N, int HW, int C, int dtype, void* stream);
int up_gap_bwd_t(const void* dy, void* dx, int lddx, int N, int HW, int C, int dtype, void* stream);
int up_dropout_fwd_t(const void* x, void* y, uint8_t* mask, const float* ext_mask, int64_t n, float p, uint64_t seed,
                     int dtype, void* stream);
/* ABI 10: ... with a step counter in DEVICE memory mixed into the seed (seed + *step_device * 0x9E3779B97F4A7C15): a captured
 * training step (hipGraph, unipose_amd.graph.GraphedTrainStep) replays the launch with unchanged arguments, its masks still
 * differ from step to step.  step_device = NULL: up_dropout_fwd_t. */
int up_dropout_fwd_step_t(const void* x, void* y, uint8_t* mask, const float* ext_mask, int64_t n, float p, uint64_t seed,
                          const uint64_t* step_device, int dtype, void* stream);
int up_dropout_bwd_t(const void* dy, const uint8_t* mask, void* dx, int64_t n, float p, int dtype, void* stream);
/* nn.MSELoss() mean reduction (unipose.py:70,117): loss[0] = mean((y-t)^2); bwd: dy = 2(y-t)/n * dloss[0] */
int up_mse_fwd(const float* y, const float* t, float* loss, float* workspace, int64_t n, void* stream);
int up_mse_bwd(const float* y, const float* t, const float* dloss, float* dy, int64_t n, void* stream);
size_t up_mse_workspace(int64_t n);

/* ---- ConvLSTM gate math (model/uniposeLSTM.py:16-24, 40-64).  `gates` is the fused gate
 * pre-activation tensor [rows][ldg] laid out g|i|o(|f), each Cg wide, produced by ONE convolution
 * over cat(x,h) with the gate weights stacked along K. ---- */
int up_lstm0_fwd(const float* gates, int ldg, float* cell, float* hide, int ldo, int64_t rows, int Cg, void* stream);
int up_lstm0_bwd(const float* gates, int ldg, const float* dcell, const float* dhide, int ldo,
                 float* dgates, int64_t rows, int Cg, void* stream);
int up_lstm_fwd(const float* gates, int ldg, const float* cprev, int ldc, float* cell, float* hide, int ldo,
                int64_t rows, int Cg, void* stream);
int up_lstm_bwd(const float* gates, int ldg, const float* cprev, int ldc, const float* cell,
                const float* dcell, const float* dhide, int ldo,
                float* dgates, float* dcprev, int64_t rows, int Cg, void* stream);

/* ---- heat-map argmax (utils/evaluate.py:32-54 get_max_preds; utils/utils.py:94-106) ----
 * hm: NCHW fp32 (B,J,H,W) as returned by the model.  One wavefront per (b,j): first-max flat index
 * (lowest index wins ties), preds = (idx % W, idx / W) zeroed where max <= 0.  idx may be NULL. */
int up_heatmap_argmax(const float* hm, int B, int J, int H, int W,
                      int32_t* idx, float* preds_xy, float* maxvals, void* stream);

/* ---- training targets and input normalisation (the step BEFORE the path) ----
 * up_make_heatmaps: lsp_lspet_data.py:224-236 / mpii_data.py:165-175.  kpt_xy (B,K,2) float64 pixel coordinates of the
 * input image; joint k of sample b is centred at int(coordinate) / stride on the H x W map; values
 * exp(-D2 / 2 / sigma^2) computed in float64 (utils/utils.py:200-203), clipped to <= 1, < 0.0099 -> 0, stored float32;
 * out (B,K+1,H,W): channel 0 = 1 - max over the joint channels, channel k+1 = joint k.
 * up_make_gaussian_maps: one such map per centre (the centre maps, lsp_lspet_data.py:238-242; centres as given).
 * up_normalize_image: (pixel - mean) / std and HWC -> CHW (Mytransforms.py:10-41 with mean 128, std 256). */
int up_make_heatmaps(const double* kpt_xy, int B, int K, int H, int W, double stride, double sigma, float* out,
                     void* stream);
int up_make_gaussian_maps(const double* center_xy, int N, int H, int W, double sigma, float* out, void* stream);
int up_normalize_image(const float* img_hwc, int B, int H, int W, int C, float mean, float stdv, float* out_chw,
                       void* stream);

/* ---- PCK / PCKh evaluation (utils/evaluate.py:5-29 calc_dists / dist_acc, :58-172 accuracy) ----
 * From the joint coordinates of the predicted and the target heat-maps (two up_heatmap_argmax calls), entirely on
 * the device: per joint the fraction of counted samples (both target coordinates > 1) whose normalised distance is
 * below 0.5 (acc), thr_pck * torso (pck) and thr_pckh * head (pckh); entry 0 of each is replaced by the mean over
 * the joints with any counted sample; visible[j] = 1 for those joints, *cnt = their number.  Head and torso sizes
 * come from the target joints of sample 0, as in the reference.  Arithmetic types follow the reference under
 * NumPy >= 2 (float32 thresholds, float64 distances). */
enum { UP_DS_LSP = 0, UP_DS_COCO = 1, UP_DS_PENN_ACTION = 2, UP_DS_NTID = 3, UP_DS_POSETRACK = 4, UP_DS_BBC = 5,
       UP_DS_MPII = 6 };
int up_pck_accuracy(const float* pred_xy, const float* target_xy, int B, int J, int H, int W, int dataset,
                    double thr_pck, double thr_pckh, double* acc, double* pck, double* pckh, double* visible,
                    int32_t* cnt, void* stream);

/* ---- multi-person decode of the optional box head (utils/uniPose.py:14-200 uniPose_kpts) ----
 * up_peak_mask: mask[e] = 1 where maps[e] > 0 and >= each in-bounds neighbour of its 3x3 neighbourhood — the peaks the
 * reference extracts with scipy's maximum_filter / binary_erosion on the centre and the four corner maps; maps is
 * (nmaps,H,W) fp32, mask (nmaps,H,W) bytes.
 * up_box_argmax: maps (C,H,W) of ONE sample; boxes (P,4) int32 = row0,row1,col0,col1 (half-open, non-empty, inside the
 * map: the caller checks, as numpy would raise); for every person and channel ch0..ch0+nch-1 the position of the first
 * maximum inside the box, relative to the box, row-major: out_hw (P,nch,2) int32 = (row, col). */
int up_peak_mask(const float* maps, int nmaps, int H, int W, uint8_t* mask, void* stream);
int up_box_argmax(const float* maps, int C, int H, int W, const int32_t* boxes, int P, int ch0, int nch,
                  int32_t* out_hw, void* stream);

/* ---- whole-graph inference entry (ABI 9) ----------------------------------------------------------------------------
 * The UniPose image network (model/unipose.py:8-38: ResNet-101 + WASP + decoder) with every BatchNorm folded into its
 * convolution, as ONE call on one stream — what the reference's validation / test loops issue as `heat = model(input)`
 * (unipose.py:150-160).  The plan owns the packed weight images and biases (device memory); activations live in a
 * caller-provided workspace of up_unipose_plan_workspace() bytes (256-byte aligned), reused between calls.
 *   up_unipose_plan_create(&cfg, &plan);
 *   for i in [0, up_unipose_plan_num_convs): set_conv(plan, i, folded weight of <name>.weight, <name>.bias or NULL, stream)
 *   up_unipose_forward(plan, input NCHW (batch, 3, H, W), heat-maps NCHW (batch, out_channels, ceil(H/8), ceil(W/8)), workspace, bytes, stream)
 * Convolution names are the reference's state_dict prefixes ("backbone.layer3.11.conv2", "wasp.aspp2.atrous_conv",
 * "decoder.last_conv.8"); a parameter that is applied twice (wasp.conv2, wasp.py:72-80) appears twice, setting it once suffices.
 * Launches and descriptors are those of the drop-in module's folded inference forward: equal bits.  Training has no
 * whole-graph entry (it runs through autograd). */
typedef struct up_unipose_plan up_unipose_plan;
typedef struct {
    int32_t batch, height, width;   /* input (batch, 3, height, width) */
    int32_t output_stride;          /* 16 or 8 (resnet.py:49-58) */
    int32_t out_channels;           /* channels of decoder.last_conv.8: num_classes + 1 (+ 5 for the multi-person box head) */
} up_unipose_config;
int up_unipose_plan_create(const up_unipose_config* cfg, up_unipose_plan** plan);
void up_unipose_plan_destroy(up_unipose_plan* plan);
int up_unipose_plan_num_convs(const up_unipose_plan* plan);
const char* up_unipose_plan_conv_name(const up_unipose_plan* plan, int i);
int up_unipose_plan_conv_shape(const up_unipose_plan* plan, int i, int32_t* oihw /* [4] */, int32_t* has_bias);
int up_unipose_plan_set_conv(up_unipose_plan* plan, int i, const float* w_oihw, const float* bias, void* stream);
size_t up_unipose_plan_workspace(const up_unipose_plan* plan);
int up_unipose_forward(up_unipose_plan* plan, const float* x_nchw, float* heat_nchw, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- measurement hooks (bench.py roofline leg; no reference counterpart) ----
 * Between begin/end every MFMA convolution launch is bracketed by two hipEvents on its stream;
 * end() returns per kernel variant {launches, total ms, total algorithmic FLOP}. */
int up_profile_variants(void);
const char* up_profile_variant_name(int i);
int up_profile_begin(void);
int up_profile_enable(int on);   /* pause / resume recording between begin and end (sampled profiling) */
int up_profile_end(double* out, int variants);
/* per variant: FLOP of the collection up_profile_end just closed with dilated launches charged for their live (pixel, tap) pairs only */
int up_profile_live_flops(double* out, int variants);

#ifdef __cplusplus
}
#endif
#endif /* UNIPOSE_HIP_H */
