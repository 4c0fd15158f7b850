/*
 * ssdseg.h -- C-ABI of libssdseg_hip.so: the MI355X (gfx950) hot path of `ssdseglib`.
 *
 * The reference (matteo-stat/multi-task-learning-object-detection-semantic-segmentation) is pure Python on
 * TensorFlow/Keras: it has no FFI of its own.  The boundary replaced here is therefore the set of TensorFlow
 * ops its hot path invokes; every entry point below cites the reference call
 * sites (file:line under the reference root) whose arithmetic it replaces.  The Python host
 * (`ssdseglib/_hip.py`, ctypes) is the only caller; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no C++/torch types.  All tensors are dense fp32, NHWC for activations,
 *     HWIO ([kh][kw][cin][cout]) for dense conv kernels, [kh][kw][c] for depthwise kernels, [cin][cout] for
 *     pointwise kernels (== Keras layouts, so get_weights()/set_weights() are memcpy).
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - every function returns 0 on success, -(hipError_t) for a HIP failure, or SSDSEG_EINVAL (-1000-n) for
 *     argument n being invalid.  Nothing throws across the boundary.  `ssdseg_last_error()` returns a
 *     thread-local description of the last failure.
 *   - launches are asynchronous on the ctx stream; `ssdseg_ctx_sync` waits.  A ctx is single-threaded.
 *   - kernels never allocate.  Scratch (BN partial sums, weight-gradient partial slabs, top-k histograms) comes
 *     from a ctx-owned workspace that grows on demand *outside* of launches (ssdseg_ctx_reserve).
 *
 * "Activation view" -- how conv + BatchNorm(train) + ReLU6 are fused:
 *   a conv kernel writes its RAW output y and per-block partial (sum, sum-of-squares) per channel; a tiny
 *   finalize kernel turns those into per-channel (scale, shift) = (gamma*invstd, beta - mean*gamma*invstd);
 *   every CONSUMER applies a = act(scale*y + shift) while loading.  So each activation is written once (raw)
 *   and never re-written normalised.  `ssdseg_view` describes such an input.
 *
 * "Gradient view" -- BatchNorm backward folded into the consumer of dY:
 *   given g = dL/d(act output), raw y and per-channel coefficients, the conv backward kernels form
 *       dy = scale * mask(z) * g + k1 * y + k0,   z = scale*y + shift,  mask = act'(z)
 *   on load (k1, k0 come from ssdseg_bn_bwd_finalize).  With scale == NULL the view is the identity (dy = g).
 */
#ifndef SSDSEG_H
#define SSDSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDSEG_VERSION 1
#define SSDSEG_EINVAL(n) (-1000 - (n))

/* activation applied by a view (Keras ReLU(max_value=...) semantics) */
enum {
    SSDSEG_ACT_NONE = 0,  /* linear (MobileNetV2 project BN: models.py:110-111)                     */
    SSDSEG_ACT_RELU = 1,  /* tf.nn.relu, grad iff z > 0 (ShuffleNetV2 backbone: models.py:529,540)    */
    SSDSEG_ACT_RELU6 = 2, /* tf.nn.relu6, grad iff 0 < z < 6 (models.py:67,90; blocks.py:30,...)      */
    SSDSEG_ACT_ZERO = 3   /* ReLU(max_value=0.0): output 0, grad 0 (quirk Q1: blocks.py:4,76,134)     */
};

typedef struct ssdseg_ctx ssdseg_ctx;

/* input view: a = act(scale[c] * x + shift[c]); scale == NULL -> a = act(x) */
typedef struct {
    const float* x;
    const float* scale;
    const float* shift;
    int32_t act;
    int32_t _pad;
} ssdseg_view;

/* gradient view: dy = scale[c]*mask(scale[c]*y+shift[c])*g + k1[c]*y + k0[c]; scale == NULL -> dy = g */
typedef struct {
    const float* g;
    const float* y;
    const float* scale;
    const float* shift;
    const float* k1;
    const float* k0;
    int32_t act;
    int32_t _pad;
} ssdseg_gview;

/* ---------------------------------------------------------------- context, memory, timing */
const char* ssdseg_last_error(void);
int ssdseg_version(void);
int ssdseg_device_count(int* count_host);
/* stream == NULL: the ctx creates (and owns) a non-blocking stream; otherwise it borrows the caller's hipStream_t. */
int ssdseg_ctx_create(int device, void* stream, ssdseg_ctx** out_host);
int ssdseg_ctx_destroy(ssdseg_ctx* ctx);
int ssdseg_ctx_sync(ssdseg_ctx* ctx);
/* Weight-gradient kernels run on an internal side stream, concurrently with the rest of the backward pass.  Every entry point
 * that synchronises, copies, records an event, all-reduces or runs the optimizer joins it implicitly; a caller that hands the
 * gradient buffers to ANOTHER library on the ctx stream calls this first. */
int ssdseg_ctx_join(ssdseg_ctx* ctx);
/* on != 0: the launches that follow go to the side stream (after everything queued so far); on == 0: back to the ctx stream.
 * For callers that issue a weight-gradient entry point themselves (ssdseg_pwconv_bwd does this internally). */
int ssdseg_ctx_side(ssdseg_ctx* ctx, int on);
/* ssdseg_ctx_side_mark: remembers the current end of the side stream; ssdseg_ctx_side_wait_mark: the ctx stream waits for that point
 * -- and only for it: side-stream work queued after the mark keeps running beside the ctx stream (a join would wait for it too).
 * Used by the Python engine between the detection branch (side stream) and the layers that need its gradients. */
int ssdseg_ctx_side_mark(ssdseg_ctx* ctx);
int ssdseg_ctx_side_wait_mark(ssdseg_ctx* ctx);
/* enabled == 0: everything on the ctx stream from now on (used by bench.py to time a kernel without a co-running neighbour);
 * enabled != 0: side stream back on (if the ctx has one). */
int ssdseg_ctx_side_enable(ssdseg_ctx* ctx, int enabled);
/* Deferred column sums.  Weight-gradient kernels split their reduction over pixel ranges and leave partial slabs that a
 * fixed-order column sum folds into dW; nothing reads dW before the optimizer (Keras `train_step`: gradients are consumed by
 * `optimizer.apply_gradients`, NB03#cell16), so with enabled != 0 those ~80 small launches per backward pass are recorded
 * instead of launched and go out as ONE launch at the next join (ssdseg_ctx_join, and every entry point that joins: sync, copies,
 * all-reduce, Adam).  Same summation order either way: gradients are bit-identical with deferral on or off.
 * enabled == 0 flushes what is pending and returns to one launch per column sum. */
int ssdseg_colsum_defer(ssdseg_ctx* ctx, int enabled);
int ssdseg_ctx_reserve(ssdseg_ctx* ctx, size_t workspace_bytes);
int ssdseg_ctx_device_name(ssdseg_ctx* ctx, char* buf_host, size_t buf_len);
int ssdseg_malloc(ssdseg_ctx* ctx, size_t bytes, void** out_host);
int ssdseg_free(ssdseg_ctx* ctx, void* ptr);
int ssdseg_memcpy_h2d(ssdseg_ctx* ctx, void* dst, const void* src_host, size_t bytes);
int ssdseg_memcpy_d2h(ssdseg_ctx* ctx, void* dst_host, const void* src, size_t bytes);
int ssdseg_memcpy_d2d(ssdseg_ctx* ctx, void* dst, const void* src, size_t bytes);
int ssdseg_memset(ssdseg_ctx* ctx, void* dst, int value, size_t bytes);
/* HIP events on the ctx stream (bench.py times kernels with these, not with torch events) */
int ssdseg_event_create(ssdseg_ctx* ctx, void** out_host);
int ssdseg_event_destroy(ssdseg_ctx* ctx, void* ev);
int ssdseg_event_record(ssdseg_ctx* ctx, void* ev);
int ssdseg_event_elapsed_ms(ssdseg_ctx* ctx, void* ev_start, void* ev_stop, float* ms_host);
/* Per-kernel timing: while enabled, every kernel launch of this library is bracketed by HIP events on the ctx
 * stream and aggregated per kernel symbol together with the launch's ALGORITHMIC bytes / flops.
 * report: one "kernel\tcount\ttotal_ms\tbytes\tflops\n" line per kernel into buf_host. */
int ssdseg_timing_enable(ssdseg_ctx* ctx, int enable);
int ssdseg_timing_reset(ssdseg_ctx* ctx);
/* bracket only launches of one kernel symbol (low overhead inside a timed region); NULL or "" = every kernel */
int ssdseg_timing_filter(ssdseg_ctx* ctx, const char* kernel);
int ssdseg_timing_report(ssdseg_ctx* ctx, char* buf_host, size_t buf_len);
/* Overlapped uploads (new: the reference feeds tf.data batches, NB03#cell8,16).  Pinned host memory and a copy stream
 * per ctx: ssdseg_upload_async enqueues host -> device on the copy stream; ssdseg_upload_fence marks a point on the ctx stream
 * (typically right after the kernels / copies that read a staging buffer) and after_fence != 0 keeps an upload behind the last
 * such point -- but not behind work queued after it, which is what lets the upload of batch i+1 run under step i;
 * ssdseg_upload_join makes the ctx stream wait for the uploads queued so far; ssdseg_upload_sync blocks the host until they are
 * done (before the pinned source is overwritten). */
int ssdseg_host_alloc(ssdseg_ctx* ctx, size_t bytes, void** out_host);
int ssdseg_host_free(ssdseg_ctx* ctx, void* ptr_host);
int ssdseg_upload_fence(ssdseg_ctx* ctx);
/* src_host: pinned (ssdseg_host_alloc: the call returns at once) or ordinary pageable memory (the call returns when the
 * runtime has staged it; the transfer still overlaps the kernels of the ctx stream) */
int ssdseg_upload_async(ssdseg_ctx* ctx, void* dst, const void* src_host, size_t bytes, int after_fence);
int ssdseg_upload_join(ssdseg_ctx* ctx);
int ssdseg_upload_sync(ssdseg_ctx* ctx);

/* ---------------------------------------------------------------- data parallelism (new: SURVEY.md 8(e))
 * The reference has no distributed code (single-process Keras fit, NB03#cell16); the boundary replaced is what
 * tf.distribute.MirroredStrategy would do around NB03#cell14-16: per-replica BatchNormalization statistics and mining pool
 * (losses.py:74-75,113,127), ONE sum-all-reduce of the gradients per step.  One process per GPU; RCCL over xGMI (librccl.so is
 * dlopen'ed at the first call below).  Rank 0 obtains a 128-byte id and hands it to the other ranks by any host-side means
 * (ssdseglib/_parallel.py: a file next to MASTER_PORT); every rank then calls ssdseg_comm_init_rank -- a collective call.
 * (SURVEY.md 8(b2) sketched `ssdseg_comm_init_all`, the single-process / all-devices form; the bench contract is one process
 * per GPU, hence init_rank.) */
#define SSDSEG_COMM_ID_BYTES 128
enum { SSDSEG_COMM_F32 = 0, SSDSEG_COMM_F64 = 1 };
enum { SSDSEG_COMM_SUM = 0, SSDSEG_COMM_MAX = 1 };
int ssdseg_comm_unique_id(void* id_host, size_t id_bytes);
int ssdseg_comm_init_rank(ssdseg_ctx* ctx, const void* id_host, size_t id_bytes, int rank, int world);
int ssdseg_comm_destroy(ssdseg_ctx* ctx);
int ssdseg_comm_info(ssdseg_ctx* ctx, int* rank_host, int* world_host);
/* The step's one collective, stream-ordered behind the backward pass (joins the weight-gradient side stream first):
 * grads[count] <- sum over ranks (ssdseg_adam_step's grad_scale = 1/world makes it the mean);
 * state[state_count] <- MEAN over ranks (BatchNormalization moving_mean / moving_variance, models.py:66,89,111: every replica
 * updates them from its own shard, the average keeps replicas and checkpoints identical).  Either may be NULL / 0. */
int ssdseg_allreduce_grads(ssdseg_ctx* ctx, float* grads, size_t count, float* state, size_t state_count);
/* in-place all-reduce of a small device buffer (timing max over ranks, barriers) and broadcast from `root` */
int ssdseg_allreduce(ssdseg_ctx* ctx, void* buf, size_t count, int dtype, int op);
int ssdseg_broadcast(ssdseg_ctx* ctx, float* buf, size_t count, int root);

/* ---------------------------------------------------------------- K1+K2: stem conv
 * Conv2D 3x3 stride 2 SAME on the rescaled image, tiny Cin (models.py:187 Rescaling x/127.5-1, :196 -> :65;
 * ShuffleNetV2 stem models.py:622,628 with bias).  in = x*in_scale + in_offset fused into the load.
 * y: [n][ho][wo][cout] raw.  stats: [nparts][2][cout] partial (sum, sumsq), may be NULL.
 * ssdseg_stem_conv_parts() tells how many partial rows the launch writes. */
int ssdseg_stem_conv_parts(int n, int h, int w, int cout, int* nparts_host);
int ssdseg_stem_conv_fwd(ssdseg_ctx* ctx, const float* x, const float* w, const float* bias, float* y,
                         int n, int h, int wdt, int cin, int cout, float in_scale, float in_offset,
                         float* stats);
/* dW (and dbias) only: the image needs no gradient.  dw: [3][3][cin][cout]. */
int ssdseg_stem_conv_bwd_weight(ssdseg_ctx* ctx, const float* x, const ssdseg_gview* dy, float* dw,
                                float* dbias, int n, int h, int wdt, int cin, int cout, float in_scale,
                                float in_offset);

/* ---------------------------------------------------------------- K3+K4: depthwise 3x3 (stride 1|2, dilation >= 1)
 * DepthwiseConv2D / depthwise half of SeparableConv2D, SAME padding, depth multiplier 1
 * (models.py:88,236,242,524,533,542,577,586; blocks.py:33,38,43,122,152).
 * TF SAME: out = ceil(in/s), pad_total = max((out-1)*s + (k-1)*d + 1 - in, 0), before = pad_total/2. */
int ssdseg_dwconv_parts(int n, int h, int w, int c, int stride, int dilation, int* nparts_host);
int ssdseg_dwconv_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, float* y, int n, int h, int wdt,
                      int c, int stride, int dilation, float* stats);
/* dx (= dL/d(act output of the producer), same shape as the input) and dw [3][3][c] in one pass.
 * accumulate != 0: dx += result (fan-out taps).  dx may be NULL (only dw wanted). */
int ssdseg_dwconv_bwd(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, const ssdseg_gview* dy, float* dx,
                      float* dw, int n, int h, int wdt, int c, int stride, int dilation, int accumulate);
/* The same plus the BatchNormalization backward of the layer that FEEDS this conv (in = act(scale*x + shift) is a BN view,
 * models.py:66-67 -> :88): the kernel that writes dx also reduces sum(mask*dx) and sum(mask*dx*xhat) over it, so the
 * 6x-wide expand tensors are not re-read by ssdseg_bn_bwd_reduce.  Valid when this conv is the only consumer of the BN
 * output (accumulate = 0) or the LAST of several to contribute (accumulate != 0: dx already holds the other consumers'
 * gradients -- a backbone tap that also feeds the heads -- and the sums are taken over the completed dx).
 * Outputs (dgamma, dbeta may be NULL): as ssdseg_bn_bwd_reduce. */
int ssdseg_dwconv_bwd_bn(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, const ssdseg_gview* dy, float* dx,
                         float* dw, int n, int h, int wdt, int c, int stride, int dilation, int accumulate,
                         const float* in_mean, const float* in_invstd, float* in_dgamma, float* in_dbeta, float* in_k1,
                         float* in_k0);

/* ---------------------------------------------------------------- K5: pointwise 1x1 conv == GEMM [m,k] x [k,n]
 * Conv2D 1x1 / pointwise half of SeparableConv2D (models.py:65,110,527,...; blocks.py:28,58,70,109).
 * fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, fp32 accumulate.
 * ldx/ldy: row strides in floats (>= k / >= n) so a layer can read/write a channel slice of a concat buffer
 * (K10: Concatenate is a write offset, never a copy). */
int ssdseg_pwconv_parts(int m, int n, int* nparts_host);
int ssdseg_pwconv_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int ldy, int m,
                      int k, int n, float* stats);
/* The tile GEMM behind the forward streams the weights with the reduction channel contiguous (Wt[n][k]); ssdseg_pwconv_fwd
 * makes that copy itself, one small launch per call.  A caller that owns all layers of a model (ssdseglib/_engine.py) does it
 * ONCE per step for all of them instead: ssdseg_pwconv_wt_floats() tells whether the default dispatch of a shape uses such a
 * copy (k*n floats, 0 = no), ssdseg_transpose_batch() fills the copies of a whole table of layers in one launch (table on the
 * device: per matrix four 64-bit words {source pointer W[k][n], destination pointer Wt[n][k], k, n}; max_tiles = the largest
 * ceil(k/32)*ceil(n/32) of the table; total_floats = sum of k*n, for the timing registry), and ssdseg_pwconv_fwd_wt() takes the
 * copy (wt may be NULL = as ssdseg_pwconv_fwd; a copy passed for a shape that runs on another kernel family is ignored). */
int ssdseg_pwconv_wt_floats(int m, int ldx, int k, int n, int* floats_host);
int ssdseg_transpose_batch(ssdseg_ctx* ctx, const long long* table, int nmat, int max_tiles, long long total_floats);
int ssdseg_pwconv_fwd_wt(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, const float* wt, float* y, int ldy,
                         int m, int k, int n, float* stats);
/* dx[m][k] = dy[m][n] * w^T; residual != NULL: dx += residual (Add backward, models.py:162);
 * accumulate != 0: dx += previous contents. */
int ssdseg_pwconv_bwd_data(ssdseg_ctx* ctx, const ssdseg_gview* dy, int ldy, const float* w, float* dx, int ldx,
                           int m, int k, int n, const float* residual, int ldr, int accumulate);
/* dw[k][n] = sum_m in[m][k] * dy[m][n] */
int ssdseg_pwconv_bwd_weight(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy,
                             float* dw, int m, int k, int n);
/* Both of the above in one call.  For k <= 32 and n <= 192 (the MBConv expand convs, models.py:65 with 6x expansion) ONE
 * kernel produces dx and dw from a single pass over the gradient view; other shapes run the two kernels above.
 * (dy->y is, in every use the host package makes of this entry point, the convolution's own raw forward output.) */
int ssdseg_pwconv_bwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy, const float* w,
                      float* dx, int lddx, float* dw, int m, int k, int n, const float* residual, int ldr,
                      int accumulate);
/* dx + dw plus the BatchNormalization backward of the layer that FEEDS this conv (in is that BN's view, models.py:89-90 ->
 * :110): the backward-data kernel's float4 epilogue reduces sum(mask*dx) and sum(mask*dx*xhat) while it stores dx.  Only
 * valid when this conv is the ONLY consumer of the BN output.  Outputs as ssdseg_bn_bwd_reduce (dgamma, dbeta may be NULL). */
int ssdseg_pwconv_bwd_bn(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy, const float* w,
                         float* dx, int lddx, float* dw, int m, int k, int n, const float* in_mean,
                         const float* in_invstd, float* in_dgamma, float* in_dbeta, float* in_k1, float* in_k0);

/* ---------------------------------------------------------------- K6: dense 3x3 stride 1 SAME (implicit GEMM)
 * Conv2D 3x3 in the DeepLabV3+ decoder (blocks.py:117,127).  w: [3][3][cin][cout]. */
int ssdseg_conv3x3_parts(int n, int h, int w, int cin, int cout, int* nparts_host);
int ssdseg_conv3x3_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int n, int h,
                       int wdt, int cin, int cout, float* stats);
/* Large layers (the DeepLabV3+ decoder conv) run in the Winograd form, whose weight gradient wants the ACTIVATED input act(scale*x+shift)
 * in a zero-bordered copy [n][h+2][w+2][cin].  ssdseg_conv3x3_saved_floats reports that copy's size for a shape (0: the shape does
 * not use it -- call the plain entry points); ssdseg_conv3x3_fwd_saved writes it (one pass; the forward kernel then reads IT, with
 * the identity view) and ssdseg_conv3x3_bwd_weight_saved consumes it, so the view is applied once per step instead of twice.
 * Same results as ssdseg_conv3x3_fwd / _bwd_weight (dy: the materialised gradient, i.e. an identity gradient view). */
int ssdseg_conv3x3_saved_floats(int n, int h, int w, int cin, int cout, long long* floats_host);
int ssdseg_conv3x3_fwd_saved(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int n, int h, int wdt, int cin,
                             int cout, float* stats, float* xsaved);
/* the same when the channels [0, c_from) of xsaved already hold the activated input (ssdseg_bilinear_fwd_padded wrote them over
 * a border that was zero when the buffer was made): only channels [c_from, cin) of `in` are viewed and copied. */
int ssdseg_conv3x3_fwd_saved_from(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int n, int h,
                                  int wdt, int cin, int cout, float* stats, float* xsaved, int c_from);
int ssdseg_conv3x3_bwd_weight_saved(ssdseg_ctx* ctx, const float* xsaved, const float* dy, float* dw, int n, int h, int wdt, int cin,
                                    int cout);
int ssdseg_conv3x3_bwd_data(ssdseg_ctx* ctx, const ssdseg_gview* dy, const float* w, float* dx, int ldx, int n,
                            int h, int wdt, int cin, int cout, int accumulate);
/* The same plus the BatchNormalization backward of the layer that FEEDS this conv (in = that BN's view of the conv input; the
 * decoder's 256 -> output_channels conv behind sepconv-batchnorm, blocks.py:121-126): for the narrow, tap-expanded form the sums
 * sum(mask*dx), sum(mask*dx*xhat) ride in the epilogue of the GEMM that writes dx; other shapes run ssdseg_conv3x3_bwd_data
 * followed by ssdseg_bn_bwd_reduce.  Only valid when this conv is the ONLY consumer of the BN output (dx is overwritten).
 * Outputs as ssdseg_bn_bwd_reduce (dgamma, dbeta may be NULL). */
int ssdseg_conv3x3_bwd_data_bn(ssdseg_ctx* ctx, const ssdseg_view* in, const ssdseg_gview* dy, const float* w, float* dx, int ldx,
                               int n, int h, int wdt, int cin, int cout, const float* in_mean, const float* in_invstd,
                               float* in_dgamma, float* in_dbeta, float* in_k1, float* in_k0);
int ssdseg_conv3x3_bwd_weight(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, float* dw,
                              int n, int h, int wdt, int cin, int cout);

/* dy->g <- scale*mask(scale*y+shift)*g + k1*y + k0 in place: materialises a BatchNorm-backward gradient view ONCE for a consumer
 * that would otherwise re-form it many times (the dense 3x3 conv reads every dy element 9x in backward-data and again in the
 * weight gradient).  Afterwards the consumer uses the identity gradient view {dy->g}. */
int ssdseg_gview_materialize(ssdseg_ctx* ctx, const ssdseg_gview* dy, int ld, int m, int c);

/* ---------------------------------------------------------------- K7/K8: BatchNormalization (training) + activation
 * Keras BatchNormalization defaults eps 1e-3, momentum 0.99 (models.py:66,89,111,...; blocks.py:29,...).
 * finalize: partial (sum, sumsq)[nparts][2][c] over `count` samples per channel ->
 *   mean, biased var -> scale = gamma*rsqrt(var+eps), shift = beta - mean*scale, invstd;
 *   moving_mean <- mom*mm + (1-mom)*mean; moving_var <- mom*mv + (1-mom)*var*count/(count-1) (Bessel, App. B.3).
 * training == 0: scale/shift from the moving statistics (inference), partials ignored. */
int ssdseg_bn_finalize(ssdseg_ctx* ctx, const float* stats, int nparts, int c, double count, const float* gamma,
                       const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                       float* mean, float* invstd, float* scale, float* shift, int training);
/* plain per-channel (sum, sumsq) of a tensor [m][c] (row stride ld) -> stats[nparts][2][c] (for tensors
 * not produced by one of the conv kernels, e.g. the GAP branch) */
int ssdseg_channel_stats_parts(int m, int c, int* nparts_host);
int ssdseg_channel_stats(ssdseg_ctx* ctx, const float* x, int ld, int m, int c, float* stats);
/* materialise out = view(in) (+ view(residual)): Add (models.py:162,593), taps, concat slices; residual may be NULL */
int ssdseg_bn_apply(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_view* residual, int ldr, float* out,
                    int ldo, int m, int c);
/* BN backward reductions: dbeta = sum mask*g, dgamma = sum mask*g*xhat over [m][c]; writes
 * dgamma, dbeta (each BatchNormalization has one use, so they are stored, not accumulated) and the gview
 * coefficients k1, k0. */
int ssdseg_bn_bwd_reduce(ssdseg_ctx* ctx, const float* g, int ldg, const float* y, int ldy, int m, int c,
                         const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                         float* dgamma, float* dbeta, float* k1, float* k0);
/* dst (+)= src over [m][c] with row strides (gradient fan-in: Concatenate / Add backward) */
int ssdseg_axpby(ssdseg_ctx* ctx, const float* src, int lds, float* dst, int ldd, int m, int c, float a, float b);

/* ---------------------------------------------------------------- K11: GlobalAveragePooling2D keepdims (blocks.py:57) */
int ssdseg_gap_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int n, int hw, int c);
/* dx[n][hw][c] (+)= g[n][c] / hw */
int ssdseg_gap_bwd(ssdseg_ctx* ctx, const float* g, float* dx, int n, int hw, int c, int accumulate);

/* ---------------------------------------------------------------- K12: UpSampling2D bilinear, half-pixel (blocks.py:61,104,129)
 * out[n][h*fy][w*fx][c] written with row stride ldo (concat slice). */
int ssdseg_bilinear_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, float* out, int ldo, int n, int h, int wdt,
                        int c, int fy, int fx);
int ssdseg_bilinear_bwd(ssdseg_ctx* ctx, const float* g, int ldg, float* dx, int ldx, int n, int h, int wdt, int c,
                        int fy, int fx, int accumulate);
/* the forward written into the INTERIOR of a bordered tensor out[n][h*fy + 2][wdt*fx + 2][ldo] (border untouched): the x4
 * up-sampled ASPP output (blocks.py:104) lands directly in the zero-bordered input copy the decoder's 3x3 conv kernels read
 * (blocks.py:117; ssdseg_conv3x3_fwd_saved_from), instead of in a plain concat buffer that a padding pass copies again. */
int ssdseg_bilinear_fwd_padded(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, float* out, int ldo, int n, int h, int wdt,
                               int c, int fy, int fx);

/* ---------------------------------------------------------------- K12+K13: mask head tail
 * logits [n][h][w][c] --x(fy,fx) bilinear--> softmax -> probabilities (output-mask, blocks.py:128-130),
 * optionally fused with the weighted cross-entropy (losses.py:294-303): loss[n] = -sum_c w_c sum_px y*log(clip p).
 * prob may be NULL (training does not need it stored); y_true/loss may be NULL (inference). */
int ssdseg_mask_head_fwd(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx,
                         const float* y_true, const float* class_weights_host, float* prob, float* loss);
/* dlogits (low resolution) = upsample^T( softmax'( dL/dp ) ), dL/dp = -loss_scale * w_c * y / p inside the clip */
int ssdseg_mask_head_bwd(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx,
                         const float* y_true, const float* class_weights_host, float loss_scale, float* dlogits);
/* The same head trained with the reference's dice / dice_square losses (losses.py:175-264; squared != 0: dice_square).  The
 * forward also leaves, per image, the eight coefficients the backward needs (coef [n][8]: A_c = -2 w_c / (T_c + eps),
 * B_c = w_c (2 I_c + eps) / (T_c + eps)^2 with I_c = sum y p, T_c = sum (y + p) [sum (y^2 + p^2)] over the image's pixels), so
 * that dL/dp_c = A_c y_c + B_c [2 p_c] is formed per pixel in the backward kernels exactly like the cross-entropy's.
 * prob, loss may be NULL (not both loss and coef). */
int ssdseg_mask_head_fwd_dice(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx,
                              const float* y_true, const float* class_weights_host, int squared, float* prob, float* loss,
                              float* coef);
int ssdseg_mask_head_bwd_dice(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx,
                              const float* y_true, const float* coef, int squared, float loss_scale, float* dlogits);
/* SSD head plumbing: Reshape(-1, 4) + Concatenate(axis=1) (blocks.py:155; models.py:256,271).  Forward: the activated
 * head tensor of one feature map, in[b][in_img_elems] (channel of element r = r % c), is written to
 * out[b][out_off_elems + r] of a buffer with out_img_elems floats per image.  reverse != 0 copies the other way
 * (gradient of the concat back to the head; `in->x` is then the concat-side buffer, `out` the head-side one). */
int ssdseg_head_gather(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int b, int in_img_elems, int c,
                       int out_off_elems, int out_img_elems, int reverse);

/* ---------------------------------------------------------------- K13..K15: detection losses
 * softmax over the class axis of head logits (models.py:259) */
int ssdseg_softmax_rows(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int rows, int c);
/* confidence_loss (losses.py:52-172) + localization_loss (losses.py:5-49) forward and gradient in one call.
 * y_labels/p_labels: [b][a][c] (p = probabilities); y_boxes/p_boxes: [b][a][4].
 * conf_loss/loc_loss: [b].  d_logits: dL/d(pre-softmax logits) [b][a][c], d_boxes: dL/d p_boxes, both already
 * scaled by loss_scale (= loss_weight / batch for Keras' SUM_OVER_BATCH_SIZE).  Either may be NULL.
 * keep_mask (optional, uint8 [b*a]) receives the hard-negative selection (1 = kept background anchor). */
int ssdseg_det_loss(ssdseg_ctx* ctx, const float* y_labels, const float* p_labels, const float* y_boxes,
                    const float* p_boxes, int b, int a, int c, float loss_scale, float* conf_loss, float* loc_loss,
                    float* d_logits, float* d_boxes, uint8_t* keep_mask);
/* exact top-k selection used by the mining step (tf.math.top_k semantics: larger value first, lower index
 * first among equals, losses.py:131): mask[i] = 1 for the k selected entries of values[n]. */
int ssdseg_topk_mask(ssdseg_ctx* ctx, const float* values, int n, int k, uint8_t* mask);
/* dice / dice_square (losses.py:175-264), API surface only: loss[n] from y_true, p [n][hw][c] */
int ssdseg_dice_loss(ssdseg_ctx* ctx, const float* y_true, const float* p, int n, int hw, int c,
                     const float* class_weights_host, int squared, float* loss);

/* ---------------------------------------------------------------- K16: anchor matching + offset encoding
 * DataEncoderDecoder._encode_ground_truth_labels_boxes (datacoder.py:205-300).
 * anchors_corners [a][4] (xmin,ymin,xmax,ymax); gt [b][gmax][5] (label,xmin,ymin,xmax,ymax), gt_count [b].
 * labels [b][a][c] one-hot, boxes [b][a][4] offsets; match (optional) [b][a] int32 matched gt index or -1. */
int ssdseg_encode_targets(ssdseg_ctx* ctx, const float* anchors_corners, int a, const float* gt,
                          const int32_t* gt_count, int b, int gmax, int c, float iou_threshold, const float* stds4_host,
                          float* labels, float* boxes, int32_t* match);

/* ---------------------------------------------------------------- compact batch -> engine buffers
 * DataEncoderDecoder.read_and_encode (datacoder.py:302-347) on the device: the host uploads what the files hold -- uint8 pixels
 * [b][h][w][3], uint8 class indices [b][h][w], ground-truth rows -- and these two calls (+ ssdseg_encode_targets) produce the
 * float32 image (tf.cast :327), the float32 one-hot mask [b][h][w][c] (tf.one_hot :332: an index >= c gives an all-zero row) and
 * the mirrored ground truth of the samples whose flip[n] != 0 (tf.image.flip_left_right :341-342; boxes xmin' = W - xmax,
 * xmax' = W - xmin :202-203, quirk Q8).  images_u8 or mask_index_u8 may be NULL (only the other is expanded); flip may be NULL. */
int ssdseg_expand_inputs(ssdseg_ctx* ctx, const uint8_t* images_u8, const uint8_t* mask_index_u8, const uint8_t* flip, float* images_f32,
                         float* mask_onehot, int b, int h, int w, int c);
/* gt [b][gmax][5] = (label, xmin, ymin, xmax, ymax), in place, rows g < gt_count[n] of the samples with flip[n] != 0 */
int ssdseg_flip_gt_boxes(ssdseg_ctx* ctx, float* gt, const int32_t* gt_count, const uint8_t* flip, int b, int gmax, float image_width);

/* ---------------------------------------------------------------- training metrics
 * Per-image values of the three metric factories NB03#cell14 passes to compile(metrics=...); Keras averages them.
 * jaccard_iou_segmentation_masks_metric (metrics.py:35-47): SOFT Jaccard, inter = sum t*p, total = sum(t+p) over pixels,
 *   sum_c w_c * inter_c / (total_c - inter_c + 1e-7).  from_logits = 1: src = the low-resolution logits [n][h][w][4], p =
 *   softmax(bilinear x(fy,fx)) recomputed per pixel (the training step never stores output-mask); from_logits = 0:
 *   src = probabilities [n][h][w][4], fy = fx = 1.  y_true [n][h*fy][w*fx][4].  out [n]. */
int ssdseg_metric_mask_iou(ssdseg_ctx* ctx, const float* src, int n, int h, int wdt, int c, int fy, int fx, int from_logits,
                           const float* y_true, const float* class_weights_host, float* out);
/* categorical_accuracy_metric (metrics.py:204-216): per class #anchors with one_hot(argmax p)[c] == y_true[c], / a,
 *   weighted sum over classes.  y_true, y_pred [b][a][4]; out [b]. */
int ssdseg_metric_label_accuracy(ssdseg_ctx* ctx, const float* y_true, const float* y_pred, int b, int a, int c,
                                 const float* class_weights_host, float* out);
/* jaccard_iou_bounding_boxes_metric (metrics.py:76-171): offsets decoded against the default boxes (anchors_centroids
 *   [a][4] = cx, cy, w, h; device), IoU with the reference's conventions, averaged over the non-background anchors of
 *   y_true (NaN for an image without objects, like the reference).  y_true, y_pred [b][a][4]; out [b]. */
int ssdseg_metric_box_iou(ssdseg_ctx* ctx, const float* y_true, const float* y_pred, const float* anchors_centroids,
                          const float* stds4_host, int b, int a, float* out);

/* ---------------------------------------------------------------- K17..K19: inference tail
 * DecodeBoxesCentroidsOffsets.call (layers.py:58-79): offsets [b][a][4] -> corners (ymin,xmin,ymax,xmax) */
int ssdseg_decode_boxes(ssdseg_ctx* ctx, const float* offsets, const float* anchors_centroids, int b, int a,
                        const float* stds4_host, float* corners);
/* NonMaximumSuppression.call (layers.py:141-162) == tf.image.combined_non_max_suppression semantics
 * (App. B.9) + repack: out [b][max_total][6] = (label, prob, xmin, ymin, xmax, ymax), zero padded;
 * valid [b] = number of real detections. */
int ssdseg_combined_nms(ssdseg_ctx* ctx, const float* corners, const float* probs, int b, int a, int c,
                        int max_per_class, int max_total, float iou_threshold, float score_threshold, float* out,
                        int32_t* valid);
/* SegmentationSuppression.call (layers.py:203-210): class-present flags over the WHOLE batch (quirk Q6) */
int ssdseg_seg_suppress(ssdseg_ctx* ctx, const float* mask_prob, int n_pixels_total, int c, const float* probs,
                        int rows, float* probs_out);

/* ---------------------------------------------------------------- K20: Adam (Keras 2.13, NB03#cell14)
 * m += (g-m)(1-b1); v += (g^2-v)(1-b2); p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps), one flat bucket.
 * grad_scale multiplies g first (1/world_size after the RCCL sum). */
int ssdseg_adam_step(ssdseg_ctx* ctx, float* params, const float* grads, float* m, float* v, size_t count,
                     float lr, float beta1, float beta2, float eps, int step, float grad_scale);

/* ---------------------------------------------------------------- K21: ShuffleNetV2-only ops
 * MaxPooling2D 3x3 stride 2 SAME (models.py:629); argmax-free backward (recomputes the window max). */
int ssdseg_maxpool3x3s2_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int n, int h, int wdt, int c);
int ssdseg_maxpool3x3s2_bwd(ssdseg_ctx* ctx, const ssdseg_view* in, const float* g, float* dx, int n, int h, int wdt,
                            int c);
/* channel shuffle (groups) of a concat of two halves: out[.., j*g + i] = view(in)[.., i*(c/g) + j]  (models.py:497-503);
 * the input is a view, so the lazily fused BatchNorm + ReLU of the concatenated branches is applied in the same pass.
 * inverse != 0 applies the inverse permutation (gradient path; pass an identity view). */
int ssdseg_channel_shuffle(ssdseg_ctx* ctx, const ssdseg_view* in, int ldi, float* out, int ldo, int m, int c, int groups,
                           int inverse);
/* General channel re-indexing: out[m][j] = table[j] >= 0 ? view(in)[m][table[j]] : 0 (+ previous contents when accumulate).
 * ShuffleNetV2 '1x' / '2x' split their stage-2 tensors into 58 / 122 channels; inside those units the branch tensors are
 * zero-padded to multiples of 4 and Split (models.py:573), the channel shuffle (models.py:497-503) and their gradients are
 * table lookups between the packed and the padded layouts.  table: device int32 [c_out]. */
int ssdseg_channel_gather(ssdseg_ctx* ctx, const ssdseg_view* in, int ldi, float* out, int ldo, long long m, int c_out,
                          const int32_t* table, int accumulate);
/* rows x cols block copy between row-major matrices of different leading dimension: parameters between their exact Keras
 * shapes (flat bucket) and the zero-padded shapes of those units */
int ssdseg_copy2d(ssdseg_ctx* ctx, float* dst, int ldd, const float* src, int lds, int rows, int cols);
/* The same for a table of blocks in ONE launch.  `table` (device): ncopies rows of six 64-bit words {dst pointer, ldd, src pointer,
 * lds, rows, cols}; max_elems = the largest rows * cols of the table (sizes the grid), total_floats = their sum (timing registry).
 * ShuffleNetV2 '1x' / '2x' (reference models.py:557-603: stage-2 branches of 58 / 122 channels) keep zero-padded copies of ~50
 * weight tensors: refreshing them and folding their gradients back was 160 launches of ~6 us per step. */
int ssdseg_copy2d_batch(ssdseg_ctx* ctx, const long long* table, int ncopies, int max_elems, long long total_floats);
/* g *= act'(x) in place: backward of a ReLU that follows an Add (ShuffleNetV2 basic unit, models.py:593-595) */
int ssdseg_act_bwd(ssdseg_ctx* ctx, float* g, int ldg, const float* x, int ldx, int m, int c, int act);

#ifdef __cplusplus
}
#endif
#endif /* SSDSEG_H */
