/*
 * yolo355.h — C ABI of libyolo355.so: the MI355X (gfx950) YOLOv3 hot path.
 *
 * This is the drop-in boundary.  The reference
 * (wizyoung/YOLOv3_TensorFlow) has no FFI of its own: its boundary is the Python
 * API of model.py and the utils package, whose arithmetic executes inside TensorFlow.  Each
 * entry point below names the reference interface (file:line under /root/reference)
 * whose TensorFlow-side execution it replaces.  The Python mirror of that API lives
 * in yolov3_tensorflow_amd/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain C, no torch/TF types: device pointers + sizes only;
 *   - every call returns 0 on success, a negative Y3_E* code otherwise; the message
 *     is available from y3_last_error() (thread-local);
 *   - the caller owns every device buffer; the library allocates no device memory
 *     and keeps no pointer beyond the call, except the per-layer parameter pointers
 *     registered in a y3_net (which the caller must keep alive);
 *   - activations are NHWC fp32 contiguous; conv kernels in TF layout are HWIO
 *     (utils/misc_utils.py:117-120) and are re-packed once by y3_pack_conv_weights;
 *   - every launch goes to the hipStream_t bound to the context; no call synchronises
 *     the host except y3_nms_counts_to_host-style helpers that say so.
 */
#ifndef YOLO355_H
#define YOLO355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y3_OK 0
#define Y3_EINVAL (-1)   /* bad argument / unsupported shape (maps to ValueError / InvalidArgumentError) */
#define Y3_EHIP (-2)     /* a HIP runtime call failed */
#define Y3_ESTATE (-3)   /* object used before it was fully configured */

#define Y3_ABI_VERSION 3

typedef struct y3_ctx y3_ctx; /* one per (device, stream) */
typedef struct y3_net y3_net; /* the 75-conv YOLOv3 graph bound to caller-owned parameters */

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* y3_last_error(void);
int y3_abi_version(void);

/* stream: a hipStream_t (0 = the null stream).  The library never creates streams.  A context owns one 64-byte block of
 * pinned, device-visible host memory: its error word (below). */
int y3_ctx_create(int device, void* stream, y3_ctx** out);
int y3_ctx_destroy(y3_ctx* ctx);
/* Device-side failures are LOUD.  The stream-K conv schedules (y3_conv2d_fwd*, y3_conv2d_dgrad*, y3_net_forward) finish
 * tiles that were cut between two persistent workgroups inside the kernel: the consumer workgroup polls a flag the
 * producer raises.  The poll is bounded so that a launch can never hang; if it expires the tile's sum
 * is incomplete, and the kernel ORs a code into the context's error word.  From then on every conv / net entry point
 * called on that context returns Y3_EHIP ("a stream-K hand-off timed out ...") WITHOUT launching, until
 * y3_ctx_check — which synchronises the stream, reports the condition once more and clears it — has been called.
 * A caller that wants the guarantee for a specific batch calls y3_ctx_check after it (the Python mirror does so
 * wherever it synchronises anyway: NMS read-back, train.py's loss read-out, bench.py after the timed region).
 * Assumption the protocol rests on, stated here because HIP does not promise it: workgroups of one launch are
 * dispatched in increasing blockIdx order.  A consumer only ever waits for workgroups with SMALLER ids of its own
 * XCD group, so under that order it cannot wait for a workgroup that has not been dispatched; if a future dispatcher
 * broke the order, the bounded poll would expire and the failure would surface here, never as a silent wrong tensor.
 * Test hook: after y3_debug_streamk_fault(1) producers never raise their flag and consumers give up after 2^10 polls,
 * until y3_debug_streamk_fault(0) (process-wide; tests/test_conv_gpu.py::test_streamk_timeout_is_loud).  It is an
 * explicit call on purpose: no environment variable changes what the product library does. */
int y3_ctx_check(y3_ctx* ctx);
void y3_debug_streamk_fault(int on);

/* ---- parameter preparation (one-off, utils/misc_utils.py:114-124 produces HWIO) ---------------
 * w_hwio [k][k][cin][cout]  ->  w_packed [k*k][cout][cin]   (cin contiguous: 16-B loads along K).
 * For the Cin==3 stem conv the kernel consumes HWIO directly and no packing is needed. */
int y3_pack_conv_weights(y3_ctx* ctx, const float* w_hwio, int k, int cin, int cout, float* w_packed);

/* Inference batch-norm folding (model.py:35-41, eps=1e-5):
 *   scale = gamma * rsqrt(var + eps) ; shift = beta - mean * scale                              */
int y3_bn_fold(y3_ctx* ctx, const float* gamma, const float* beta, const float* mean,
               const float* var, float eps, int c, float* scale, float* shift);

/* ---- a1-a4: conv2d + BN + LeakyReLU (+ residual) (+ fused upsample/concat input) ---------------
 * Replaces utils/layer_utils.py:9-22 `conv2d` (slim.conv2d + batch_norm + leaky_relu, model.py:43-49),
 * the residual add of `res_block` (utils/layer_utils.py:25-32), `upsample_layer` (:82-87) and the
 * channel concat of model.py:62,72.
 *
 *   y = act(conv(x, w) * scale + shift) + residual
 *
 * k in {1,3}; stride 1 -> SAME; stride 2 -> explicit zero pad 1 each side then VALID
 * (utils/layer_utils.py:10-21), i.e. input row = oy*stride + ky - (k/2) in both cases.
 * If x_up != NULL (k must be 1): the logical input is concat([nearest_up2x(x_up), x], axis=3),
 * x_up is [N, H/2, W/2, c_up], x is [N, H, W, cin - c_up]; never materialised.
 * act: 0 = linear (detection convs, model.py:55-57), 1 = LeakyReLU(0.1).
 * scale/shift are per-output-channel (folded BN, or scale=1/shift=bias); residual may be NULL. */
typedef struct y3_conv_desc {
    int n, h, w;   /* logical input spatial size (after upsample for the fused-concat case) */
    int cin;       /* logical input channels (c_up + channels of x) */
    int c_up;      /* channels coming from x_up (0 if none) */
    int cout;
    int k;         /* 1 or 3 */
    int stride;    /* 1 or 2 */
    int act;       /* 0 linear, 1 leaky(0.1) */
} y3_conv_desc;

/* Scratch the stream-K schedule of this conv may use (0 if it never does): one accumulator slot per persistent
 * workgroup + one flag word each; uninitialised memory is fine (the library zeroes the words it polls ahead of every
 * launch; y3_net_forward gives each of its layers its own flag words and zeroes them all with one memset per forward)
 * and it must not be shared by launches on different streams.  Passing workspace = NULL to
 * y3_conv2d_fwd is allowed and selects the data-parallel schedule; results of the two schedules differ in
 * the last bits (the K sum of a split tile is associated differently), each is deterministic. */
size_t y3_conv_workspace_bytes(const y3_conv_desc* d);
/* Test hook (host arithmetic only, no device needed): the stream-K work split the kernels evaluate on the device.
 * `units` output tiles (kind 0: direct / split kernels) or Winograd blocks (kind 1) of `ksteps` K-steps each are divided,
 * whole, among 8 workgroup groups (blockIdx %% 8); inside group `group` its workers/8 local workers own equal
 * contiguous ranges of (unit, K-step) items: [*begin, *end) for `local_worker`.  A unit cut by a range boundary is
 * finished inside the kernel by the worker owning its K-step 0, from the partial sums the following local workers
 * publish; tests/test_streamk_partition.py replays that protocol on these ranges.
 * kind 2 = the Winograd kernel's default schedule: of group g's blocks [b0, b1) the first R*G (G = workers/8, R =
 * (b1-b0)/G whole rounds) are computed whole - round r, local worker j: block b0 + r*G + j - and [*begin, *end) is
 * `local_worker`'s range of the remaining (b1 - b0 - R*G) * ksteps items, cut as above. */
int y3_streamk_range(int kind, int units, int ksteps, int workers, int group, int local_worker, long long* begin,
                     long long* end);
int y3_conv2d_fwd(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* x_up,
                  const float* w, const float* scale, const float* shift, const float* residual,
                  float* y, void* workspace, size_t workspace_bytes);

/* The first two convs of darknet53_body in one launch (utils/layer_utils.py:34-40: conv2d(inputs, 32, 3) and
 * conv2d(net, 64, 3, strides=2), each with folded batch norm and LeakyReLU), exact fp32 arithmetic: x = the image [n,h,w,3],
 * w0_hwio = the stem's HWIO kernel [3][3][3][32], w1_packed = the second conv's kernel from y3_pack_conv_weights (k = 3,
 * cin = 32, cout = 64), y = [n,h/2,w/2,64].  The stem's output - the largest tensor of the network - exists only in the LDS.
 * h and w even.  y3_net_forward (dtypes 0 and 4) uses it for its layers 0 and 1. */
int y3_conv2d_fwd_stem_s2(y3_ctx* ctx, int n, int h, int w, const float* x, const float* w0_hwio, const float* scale0,
                          const float* shift0, const float* w1_packed, const float* scale1, const float* shift1, float* y);

/* ---- bf16 storage / fp32 accumulate variant (BASELINE configs[4]: 608x608 bf16 inference) -----------------
 * Same contract as y3_conv2d_fwd (utils/layer_utils.py:9-22) with bf16 (round-to-nearest-even) activations, residual
 * and packed weights; scale/shift stay fp32; the accumulator is fp32; ONE rounding to bf16 at the store.
 * out_f32 != 0 writes fp32 (used for the detection convs so that decode/NMS are unchanged).  The Cin==3 stem
 * takes the fp32 image and the fp32 HWIO kernel and writes bf16.
 * w_packed (k*k*cin*cout bf16) is OPAQUE: y3_pack_conv_weights_bf16 chooses the layout from (k, cin) -
 * [tap][cin/64][cout][64] for the 3x3 convs with cin % 64 == 0 and (round 5) for the 1x1 convs with cin >= 512 and
 * cin % 64 == 0, [tap][cin/32][cout][32] otherwise - and y3_conv2d_fwd_bf16 assumes the same rule.  (Kernels behind it:
 * csrc/y3_conv_bf16x.hip - LDS-DMA 3x3 convs, tile shape per launch from a fitted cost model -, csrc/y3_conv_bf16r.hip - the
 * persistent ring kernel of the deep 1x1 convs -, csrc/y3_conv_bf16.hip - the register-staged kernel of the others.) */
int y3_pack_conv_weights_bf16(y3_ctx* ctx, const float* w_hwio, int k, int cin, int cout, void* w_packed);
/* Host-only: which tile shape y3_conv2d_fwd_bf16 runs this launch on - 'A' 256x256, 'B' 256x128, 'C' 128x128, 'D' 192x256, 'E'
 * 192x128 (3x3 convs, csrc/y3_conv_bf16x.hip); 'a'..'g' (the 1x1 ring kernel, csrc/y3_conv_bf16r.hip); 'x' narrow 3x3 forms; 'o'
 * the register-staged kernel; 's' the stem.  Documentation of the dispatch, not needed to call anything. */
int y3_conv_bf16_tile(const y3_conv_desc* d);
int y3_conv2d_fwd_bf16(y3_ctx* ctx, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                       const float* scale, const float* shift, const void* residual, void* y, int out_f32);

/* The first two convs of darknet53_body in one launch (utils/layer_utils.py:34-40: conv2d(inputs, 32, 3) and
 * conv2d(net, 64, 3, strides=2), each with folded batch norm and LeakyReLU), bf16 storage: x = the fp32 image [n,h,w,3],
 * w0_hwio = the stem's fp32 HWIO kernel [3][3][3][32], w1_packed = the second conv's kernel from y3_pack_conv_weights_bf16
 * (k = 3, cin = 32, cout = 64), y = bf16 [n,h/2,w/2,64].  The stem's output exists only in the LDS; both convs accumulate
 * in fp32 on the bf16 matrix pipe.  The stem does NOT round the image: the fp32 image and the stem's kernel are each split once
 * into a high and a low bf16 part and the conv runs as three products (hi*hi + hi*lo + lo*hi: fp32-grade; a plainly rounded
 * image cost 0.05 of the mAP-style gate of tests/test_bf16_gpu.py); its output is rounded to bf16 like every other activation
 * of this path.  h and w even.
 * y3_net_forward (dtype 1) uses it for its layers 0 and 1. */
int y3_conv2d_fwd_bf16_stem_s2(y3_ctx* ctx, int n, int h, int w, const float* x, const float* w0_hwio, const float* scale0,
                               const float* shift0, const void* w1_packed, const float* scale1, const float* shift1, void* y);

/* The first residual block of darknet53_body in one launch (utils/layer_utils.py:25-32 res_block(net, 32) on a 64-channel map:
 * y = x + conv3x3(conv1x1(x)), both convs with folded batch norm and LeakyReLU), bf16 storage: x, y = bf16 [n,h,w,64];
 * w2_packed / w3_packed = the kernels of the 1x1 (64 -> 32) and the 3x3 (32 -> 64) conv from y3_pack_conv_weights_bf16.  The
 * 32-channel tensor between the convs exists only in the LDS and x is read once (it is the input AND the shortcut).
 * y3_net_forward (dtype 1) uses it for its layers 2 and 3. */
int y3_resblock64_fwd_bf16(y3_ctx* ctx, int n, int h, int w, const void* x, const void* w2_packed, const float* scale2,
                           const float* shift2, const void* w3_packed, const float* scale3, const float* shift3, void* y);

/* ---- Winograd F(2x2,3x3) form of the stride-1 3x3 conv (exact fp32 arithmetic, 2.25x fewer multiplies) ------------
 * Same tensors and epilogue as y3_conv2d_fwd (utils/layer_utils.py:9-22,25-32) for the convs
 * y3_conv_wino_eligible accepts (k = 3, stride 1, no fused upsample input, Cin %% 32 == 0,
 * Cout %% 32 == 0).  w_wino = G g G^T per (cin, cout), packed [16][cin/8][cout][8] fp32 (16*cin*cout floats) by
 * y3_pack_conv_weights_wino.  Results differ from the direct kernel by a few fp32 roundings per term.  With a
 * workspace the kernel may pick a stream-K schedule (persistent grid; cut blocks are summed inside the kernel in a fixed
 * order, so results are run-to-run bit-exact), as y3_conv2d_fwd does.  The workspace needs no initialisation: the
 * library zeroes the words it polls ahead of every launch. */
int y3_conv_wino_eligible(const y3_conv_desc* d);
int y3_pack_conv_weights_wino(y3_ctx* ctx, const float* w_hwio, int cin, int cout, float* w_wino);
size_t y3_conv_wino_workspace_bytes(const y3_conv_desc* d);   /* stream-K scratch; workspace = NULL is allowed */
int y3_conv2d_fwd_wino(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino, const float* scale,
                       const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes);
/* The same conv in its Winograd F(4x4,3x3) form (36 multiplies per 4x4 output tile and channel pair: 1.78x less
 * matrix-pipe work again).  A workgroup owns 16 tiles x 64 channels, two workgroups per CU.  y3_conv_wino44_eligible accepts
 * (k = 3, stride 1, no fused upsample input, Cin %% 32 == 0, Cout %% 64 == 0).  w_wino44 = G g G^T with the 6x3 G of the
 * interpolation points 0, +-1, +-2, inf, packed [18 position pairs][cin/8][cout][4 channel pairs][2 positions][2 channels]
 * fp32 (36*cin*cout floats: an opaque layout, what the kernel's 16-byte fragment loads want) by
 * y3_pack_conv_weights_wino44.  Results differ from the direct kernel by fp32 roundings of the transforms (measured on the
 * whole network: boxes 1.0e-5 of the box scale from the fp64 oracle, 6.3e-6 for the direct sum).
 * WORKSPACE (round 6): y3_conv_wino44_workspace_bytes(d) = the bytes of V = B^T d B for this conv (2.25x the input, rounded up
 * to 16-tile blocks).  With a 16-byte-aligned workspace of at least that size the conv runs as TWO kernels - the input
 * transform written once, then 36 batched GEMMs + the output transform (csrc/y3_conv_wino44.hip, form 1) -; with
 * workspace = NULL (or a smaller one) as ONE kernel that transforms inside its K-loop (form 2: rounds 3-5).  Same arithmetic
 * in the same order either way; the workspace needs no initialisation and carries nothing between calls. */
int y3_conv_wino44_eligible(const y3_conv_desc* d);
int y3_conv_wino44_candidate(const y3_conv_desc* d);   /* by shape: the convs worth an alternative packing (y3_net_set_layer_alt) */
int y3_conv_wino44_preferred(const y3_conv_desc* d);   /* for THIS n, h, w: a candidate with enough blocks to fill the CUs */
int y3_pack_conv_weights_wino44(y3_ctx* ctx, const float* w_hwio, int cin, int cout, float* w_wino44);
size_t y3_conv_wino44_workspace_bytes(const y3_conv_desc* d);   /* bytes of V (two-kernel form); workspace = NULL is allowed */
int y3_conv2d_fwd_wino44(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino44, const float* scale,
                         const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes);
/* Training uses of the F(4x4,3x3) kernel (train.py:105-115; round 4; the workspace arguments round 6, same rule as above):
 *   y3_conv2d_fwd_wino44_stats: the conv plus the column sums of y and y^2 per 16-tile block - stats
 *     [y3_conv_stats_blocks(d, 2)][2][cout] floats, for y3_bn_train_stats_partials (same contract as
 *     y3_conv2d_fwd_wino_stats: no residual);
 *   y3_pack_conv_weights_wino44_dgrad + y3_conv2d_dgrad_wino44: the data gradient of a stride-1 3x3 conv as the same kernel
 *     on dz with the flipped, channel-swapped kernel (same arguments as y3_conv2d_dgrad_wino; w_wino44_d = 36 * dz_stride *
 *     cin floats; needs dz_stride %% 32 == 0 and cin %% 64 == 0; its workspace is that of the conv [n,h,w,dz_stride] ->
 *     [n,h,w,cin]: y3_conv_wino44_workspace_bytes of fwd with cin = dz_stride, cout = fwd->cin). */
int y3_conv2d_fwd_wino44_stats(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino44, const float* scale,
                               const float* shift, float* y, float* stats, void* workspace, size_t workspace_bytes);
int y3_pack_conv_weights_wino44_dgrad(y3_ctx* ctx, const float* w_d, int cin, int dz_stride, float* w_wino44_d);
int y3_conv2d_dgrad_wino44(y3_ctx* ctx, const y3_conv_desc* fwd, const float* dz, int dz_stride, const float* w_wino44_d,
                           const float* ones, const float* zeros, int accumulate, float* dx, void* workspace,
                           size_t workspace_bytes);

/* ---- fp32 on the bf16 matrix pipe ----------------------------------------------------------------------------
 * Same contract and tensors as y3_conv2d_fwd (fp32 NHWC in, fp32 out, same epilogue, same workspace rule); every
 * fp32 product a*b is rebuilt from bf16 plane products with fp32 accumulation: planes = 3 splits each operand
 * exactly into three bf16 values and keeps the 6 products a_i*b_j, i+j <= 2 (dropped terms <= 2^-23 |ab|, i.e.
 * fp32-level accuracy at 6/16 of the fp32-MFMA cost); planes = 2 keeps 3 products (<= 2^-15 |ab|).
 * w_split = [k*k][cin/16][planes][cout][16] bf16 (cin % 16 == 0) from y3_pack_conv_weights_split (the Cin==3 stem takes the fp32 HWIO
 * kernel and runs the exact kernel).  Replaces the same reference code as y3_conv2d_fwd
 * (utils/layer_utils.py:9-22). */
int y3_pack_conv_weights_split(y3_ctx* ctx, const float* w_hwio, int k, int cin, int cout, int planes,
                               void* w_split);
int y3_conv2d_fwd_split(y3_ctx* ctx, const y3_conv_desc* d, int planes, const float* x, const float* x_up,
                        const void* w_split, const float* scale, const float* shift, const float* residual,
                        float* y, void* workspace, size_t workspace_bytes);

/* ---- unfused graph ops, for callers composing the network op by op (utils/layer_utils.py) ---------
 * y3_net_forward never launches these (it fuses them into the neighbouring convs).
 * y3_upsample_nearest : tf.image.resize_nearest_neighbor, align_corners=False (utils/layer_utils.py:82-87)
 * y3_concat_channels  : tf.concat([a, b], axis=3) on NHWC with `rows` = N*H*W (model.py:62,72)
 * y3_add              : net + shortcut (utils/layer_utils.py:30)
 * y3_reorg_boxes      : box part of reorg_layer for one scale (model.py:96-131):
 *                       boxes [N,gh,gw,3,4] = (cx,cy,w,h) in input pixels; anchors3 = the 3 (w,h) pairs. */
int y3_upsample_nearest(y3_ctx* ctx, const float* x, int n, int h, int w, int c, int out_h, int out_w,
                        float* y);
int y3_concat_channels(y3_ctx* ctx, const float* a, int ca, const float* b, int cb, long long rows, float* y);
int y3_add(y3_ctx* ctx, const float* a, const float* b, long long count, float* y);
int y3_reorg_boxes(y3_ctx* ctx, const float* fm, int n, int gh, int gw, int class_num, int img_h, int img_w,
                   const float* anchors3_host, float* boxes);

/* ---- a6-a8: reorg_layer + predict (+ conf*prob) ------------------------------------------------
 * Replaces model.py:82-137 / :140-190 and the score product of test_single_image.py:55.
 * fm_s: [N, H/stride_s, W/stride_s, 3*(5+C)] for strides 32,16,8 (in that order).
 * anchors: 9 (w,h) pairs in data/yolo_anchors.txt order; scale s uses anchors[6-3s .. 8-3s].
 * boxes [N,B,4] = (x_min,y_min,x_max,y_max); confs [N,B,1]; probs [N,B,C]; scores [N,B,C] or NULL.
 * B = 3*(g1^2+g2^2+g3^2), box order = scale (13,26,52), then y, x, anchor (model.py:155-180). */
int y3_decode(y3_ctx* ctx, const float* fm1, const float* fm2, const float* fm3, int n, int h, int w,
              int class_num, const float* anchors_host18, float* boxes, float* confs, float* probs,
              float* scores);

/* ---- a9/a10: per-class NMS ---------------------------------------------------------------------
 * mode Y3_NMS_TF : utils/nms_utils.py:8-48 `gpu_nms` = per class {score >= thresh,
 *                  tf.image.non_max_suppression (IoU without +1, suppress if IoU > thresh)}.
 * mode Y3_NMS_PY : utils/nms_utils.py:51-123 `cpu_nms`/`py_nms` (+1 on intersection w/h only,
 *                  keep while ovr <= thresh).
 * Tie-break in both modes: (score descending, box index ascending).
 * Batched over n images (the reference handles one image per call; n=1 reproduces it).
 * Outputs per image i, concatenated by class ascending, selection order within a class:
 *   out_boxes [n][cap][4], out_scores [n][cap], out_labels [n][cap] (int32),
 *   out_index [n][cap] (int32 box index into the B inputs; NULL to skip), out_counts [n] (int32),
 *   with cap = class_num * max_boxes.  All device pointers; counts are read by the caller (nothing in the call waits for
 *   the host: copy them back asynchronously and read them when the batch is consumed).  The scratch may be reused by the next
 *   call on the same stream.  Classes with up to 16,384 candidates are selected from keys sorted in the LDS, in chunks, with an
 *   early exit at max_boxes; beyond that, and for max_boxes above ~1,600 (the selected boxes of a class live in the LDS too), an
 *   arg-max form on global memory takes over - same selections either way. */
#define Y3_NMS_TF 0
#define Y3_NMS_PY 1
size_t y3_nms_workspace_bytes(int n, int num_boxes, int class_num, int max_boxes);
int y3_nms(y3_ctx* ctx, int mode, const float* boxes, const float* scores, int n, int num_boxes,
           int class_num, int max_boxes, float score_thresh, float iou_thresh, void* workspace,
           size_t workspace_bytes, float* out_boxes, float* out_scores, int32_t* out_labels,
           int32_t* out_index, int32_t* out_counts);

/* ---- a5: yolov3.forward (model.py:30-80) as one call -------------------------------------------
 * The graph is fixed by class_num: 52 backbone convs (utils/layer_utils.py:24-68) + 23 head convs
 * (model.py:53-78), indexed 0..74 in variable-creation order (= darknet file order).
 * Per layer the caller registers device pointers: packed weights (HWIO for layer 0), scale, shift. */
int y3_net_create(y3_ctx* ctx, int class_num, y3_net** out);
int y3_net_destroy(y3_net* net);
/* 0 = fp32 (default), 1 = bf16 storage: layer parameters must then be bf16-packed (fp32 HWIO for layer 0),
 * intermediate activations are bf16, the three feature maps stay fp32.  2 / 3 = fp32 tensors with the products on
 * the bf16 matrix pipe (y3_conv2d_fwd_split with planes = 3 / 2): layer weights from y3_pack_conv_weights_split.
 * 4 = fp32 with the Winograd kernel for the layers y3_conv_wino_eligible accepts (their weights from
 * y3_pack_conv_weights_wino) and the direct kernel for the rest.  A layer y3_conv_wino44_candidate names may also get
 * its F(4x4,3x3) packing (y3_net_set_layer_alt, weights from y3_pack_conv_weights_wino44): y3_net_forward then runs it on
 * that kernel whenever y3_conv_wino44_preferred says the launch is large enough (bs=32 at 416x416: yes; bs=4: no).  (The
 * train step, y3_net_train_*, packs its own kernels every step and uses F(2x2,3x3) throughout.) */
int y3_net_set_dtype(y3_net* net, int dtype);
int y3_net_num_layers(const y3_net* net);
/* geometry of layer i for input-independent fields: k, stride, cin, cout, has_bn */
int y3_net_layer_info(const y3_net* net, int i, int* k, int* stride, int* cin, int* cout, int* has_bn);
int y3_net_set_layer(y3_net* net, int i, const float* w_packed, const float* scale, const float* shift);
int y3_net_set_layer_alt(y3_net* net, int i, const float* w_wino44);   /* optional, dtype 4: see y3_net_set_dtype */
size_t y3_net_workspace_bytes(const y3_net* net, int n, int h, int w);
/* x [n,h,w,3] -> fm1 [n,h/32,w/32,3*(5+C)], fm2 (/16), fm3 (/8).  h,w multiples of 32. */
int y3_net_forward(y3_net* net, const float* x, int n, int h, int w, void* workspace,
                   size_t workspace_bytes, float* fm1, float* fm2, float* fm3);
/* Optional per-layer timing with hipEvents on the context stream.  While enabled, every forward (up to
 * 256) records one event per layer boundary; y3_net_get_layer_ms synchronises on the last event, writes
 * the per-layer elapsed ms averaged over the forwards recorded since the previous call, and resets. */
int y3_net_set_profiling(y3_net* net, int enabled);
int y3_net_get_layer_ms(y3_net* net, float* ms, int count);
int y3_net_layer_is_streamk(const y3_net* net, int i, int n, int h, int w);
/* which launches y3_net_forward fuses at this size (dtype as set): 0 = layer i has its own launch; 1 = it runs inside the NEXT
 * layer's launch (its output tensor never reaches memory; its profiled time is 0); 2 = its launch also runs the layer before it
 * (stem + stride-2 conv in the fp32 and bf16 paths, the first residual block in the bf16 path: model.py:34-40 of the reference) */
int y3_net_layer_fused(const y3_net* net, int i, int n, int h, int w);

/* graph topology of y3_net (tensor ids: 0 = network input, 1.. = conv outputs in creation order; -1 = none) */
int y3_net_layer_graph(const y3_net* net, int i, int* src, int* up, int* resid, int* dst, int* act);
int y3_net_num_tensors(const y3_net* net);
int y3_net_tensor_info(const y3_net* net, int t, int* channels, int* sdiv, int* ext);

/* ==== training path (train.py:72-115 of the reference) ==========================
 * All reductions are two-stage with a fixed combination order (no float atomics): deterministic. */

/* K8: batch norm in batch-statistics mode (model.py:35-41 with is_training=True, train.py:74).
 * z [rows][c] raw conv output.  Computes the batch mean / biased variance over rows, writes mean, inv_std
 * = rsqrt(var+eps), the folded scale = gamma*inv_std and shift = beta - mean*scale for y3_bn_apply_fwd, and
 * (if non-NULL) updates the moving statistics in place: moving <- moving*decay + batch*(1-decay), the
 * UNBIASED variance going into moving_var (TF fused batch norm).  scratch: y3_reduce_scratch_bytes(c). */
size_t y3_reduce_scratch_bytes(int c);
int y3_bn_train_stats(y3_ctx* ctx, const float* z, long long rows, int c, const float* gamma, const float* beta,
                      float eps, float decay, float* mean, float* inv_std, float* scale, float* shift,
                      float* moving_mean, float* moving_var, float* scratch);
/* The same statistics without the extra pass over z: the conv entry points below also write, per row block of their
 * output, the column sums of y and y^2 — stats [y3_conv_stats_blocks(d, wino)][2][cout] floats — taken in the conv's
 * epilogue where the tile is in registers (the training forward calls them with scale = 1, shift = 0, act = 0, so y = z);
 * y3_bn_train_stats_partials then finishes exactly like y3_bn_train_stats (fixed-order fp64 combination: deterministic).
 * y3_conv_stats_blocks returns 0 for convs without this support (the Cin = 3 stem, Cout %% 4 != 0, fused upsample
 * inputs; wino = 1: convs y3_conv_wino_eligible rejects; wino = 2: the F(4x4,3x3) kernel, convs y3_conv_wino44_eligible
 * rejects).  Same arguments as y3_conv2d_fwd / y3_conv2d_fwd_wino otherwise (no x_up, no residual). */
int y3_conv_stats_blocks(const y3_conv_desc* d, int wino);
int y3_conv2d_fwd_stats(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w, const float* scale,
                        const float* shift, float* y, float* stats, void* workspace, size_t workspace_bytes);
int y3_conv2d_fwd_wino_stats(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino, const float* scale,
                             const float* shift, float* y, float* stats, void* workspace, size_t workspace_bytes);
int y3_bn_train_stats_partials(y3_ctx* ctx, const float* partial, int nblocks, long long rows, int c,
                               const float* gamma, const float* beta, float eps, float decay, float* mean,
                               float* inv_std, float* scale, float* shift, float* moving_mean, float* moving_var);
/* y = act(z*scale + shift) + residual   (act: 1 = LeakyReLU(0.1); residual may be NULL) */
int y3_bn_apply_fwd(y3_ctx* ctx, const float* z, const float* scale, const float* shift, const float* residual,
                    long long rows, int c, int act, float* y);
/* Backward of leaky(BN_train(z)): given dy, writes d gamma, d beta and dz (dz may alias dy).
 * scratch: y3_bn_bwd_scratch_bytes(c). */
size_t y3_bn_bwd_scratch_bytes(int c);
int y3_bn_train_bwd(y3_ctx* ctx, const float* z, const float* dy, const float* gamma, const float* scale,
                    const float* shift, const float* mean, const float* inv_std, long long rows, int c,
                    float* dgamma, float* dbeta, float* dz, float* scratch);
/* d bias = sum over rows of dy [rows][c] (detection convs, model.py:55-57); scratch: 1024*c floats */
int y3_bias_grad(y3_ctx* ctx, const float* dy, long long rows, int c, float* dbias, float* scratch);

/* K9: conv backward (TF autodiff of slim.conv2d, train.py:112).  `fwd` describes the FORWARD layer
 * (n,h,w = its input size; c_up must be 0: training materialises the upsample+concat).
 * dgrad: dx [n,h,w,cin] (+)= data gradient.  dz is [n,h/s,w/s] x dz_stride channels (dz_stride >= cout,
 *        multiple of 32: the 3*(5+C) detection convs pad to the next multiple), w_d = the HWIO kernel with
 *        its last axis padded to dz_stride ([k*k][cin][dz_stride]); `ones`/`zeros` are [cin] device vectors.
 * wgrad: dw_hwio [k][k][cin][cout] = weight gradient; scratch: y3_conv_wgrad_scratch_bytes(fwd), 16-byte aligned
 *        (dw_hwio itself: any 4-byte boundary). */
int y3_conv2d_dgrad(y3_ctx* ctx, const y3_conv_desc* fwd, const float* dz, int dz_stride, const float* w_d,
                    const float* ones, const float* zeros, int accumulate, float* dx, void* workspace,
                    size_t workspace_bytes);
/* The same data gradient on the bf16 matrix pipe (stride-1 convs; see y3_conv2d_fwd_split): w_split_d comes from
 * y3_pack_conv_weights_split_dgrad(w_d = the [k*k][cin][dz_stride] kernel above). */
int y3_pack_conv_weights_split_dgrad(y3_ctx* ctx, const float* w_d, int k, int cin, int dz_stride, int planes,
                                     void* w_split_d);
int y3_conv2d_dgrad_split(y3_ctx* ctx, const y3_conv_desc* fwd, int planes, const float* dz, int dz_stride,
                          const void* w_split_d, const float* ones, const float* zeros, int accumulate, float* dx,
                          void* workspace, size_t workspace_bytes);
/* The same data gradient in Winograd F(2x2,3x3) form (stride-1 3x3 convs with Cin %% 32 == 0 and dz_stride %% 32 == 0;
 * the gradient of a SAME stride-1 conv is a SAME stride-1 conv with the flipped, channel-swapped kernel, so the forward
 * Winograd kernel runs it): w_wino_d = 16 * dz_stride * cin floats from y3_pack_conv_weights_wino_dgrad(w_d = the
 * [k*k][cin][dz_stride] kernel above); workspace as y3_conv_wino_workspace_bytes of the [n,h,w,dz_stride]->[..,cin] conv. */
int y3_pack_conv_weights_wino_dgrad(y3_ctx* ctx, const float* w_d, int cin, int dz_stride, float* w_wino_d);
int y3_conv2d_dgrad_wino(y3_ctx* ctx, const y3_conv_desc* fwd, const float* dz, int dz_stride, const float* w_wino_d,
                         const float* ones, const float* zeros, int accumulate, float* dx, void* workspace,
                         size_t workspace_bytes);
size_t y3_conv_wgrad_scratch_bytes(const y3_conv_desc* fwd);
int y3_conv_wgrad(y3_ctx* ctx, const y3_conv_desc* fwd, const float* x, const float* dz, int dz_stride,
                  float* dw_hwio, void* scratch, size_t scratch_bytes);
/* The weight gradient of a stride-1 3x3 conv in Winograd F(2x2,3x3) form (Cin %% 64 == 0, Cout %% 64 == 0, at least
 * 8 / ceil(w/2) + 1 rows of 2x2 tiles per image — y3_conv_wgrad_wino_eligible says):
 *   dw = G^T [ sum over 2x2 output tiles of (B^T d B) .* (A dY A^T) ] G,
 * exact fp32 arithmetic with 16/36 of the multiplies of y3_conv_wgrad; results differ from it by a few fp32 roundings
 * per term; deterministic (the tile range is split over workgroups, partial kernels are added in a fixed order).
 * scratch: y3_conv_wgrad_wino_scratch_bytes(fwd), 16-byte aligned. */
int y3_conv_wgrad_wino_eligible(const y3_conv_desc* fwd);
size_t y3_conv_wgrad_wino_scratch_bytes(const y3_conv_desc* fwd);
int y3_conv_wgrad_wino(y3_ctx* ctx, const y3_conv_desc* fwd, const float* x, const float* dz, int dz_stride,
                       float* dw_hwio, void* scratch, size_t scratch_bytes);
/* backward routing: 2x2 sum of the gradient of a nearest-upsampled tensor (g has g_channels per pixel, the
 * first c belong to the upsampled part), channel-slice (accumulate), zero-extension of the channel axis */
int y3_upsample2x_bwd(y3_ctx* ctx, const float* g, int g_channels, int n, int h, int w, int c, int accumulate,
                      float* dx);
int y3_slice_accumulate(y3_ctx* ctx, const float* src, int src_channels, int offset, long long rows, int c,
                        int accumulate, float* dst);
int y3_pad_channels(y3_ctx* ctx, const float* src, int c_src, long long rows, int c_dst, float* dst);

/* K10: loss_layer forward + backward for one scale (model.py:192-304, box_iou :307-345).
 * feature_map [n,gh,gw,3*(5+C)], y_true [n,gh,gw,3,6+C] (utils/data_utils.py:69-113 layout), anchors3 = the
 * 3 (w,h) pairs of this scale.  loss4 (device, 4 floats: xy, wh, conf, class; each already divided by N)
 * is overwritten or accumulated; grad [n,gh,gw] x grad_stride receives d(sum of the four)/d(feature_map). */
size_t y3_loss_scratch_bytes(int n, int gh, int gw);
int y3_loss_layer(y3_ctx* ctx, const float* feature_map, const float* y_true, int n, int gh, int gw,
                  int class_num, int img_h, int img_w, const float* anchors3_host, int use_label_smooth,
                  int use_focal_loss, int accumulate, float* loss4, float* grad, int grad_stride, void* scratch,
                  size_t scratch_bytes);

/* ---- a14 as ONE call: the train step of train.py:72-115 on the y3_net graph ---------------------------------------
 * The per-op entry points above, sequenced by the library over a caller-owned workspace (whole-graph entries;
 * a host in any language trains through these without re-implementing the backward walk):
 *   y3_net_train_forward   yolov3.forward(inputs, is_training=True) (model.py:30-80): batch-statistics BN in all 72 BN
 *                          layers, moving statistics updated in place (decay = opts->bn_decay, unbiased variance), the
 *                          tensors backward needs are kept in the workspace; *fm1..3 = the feature maps (inside it)
 *   y3_net_train_loss      yolov3.compute_loss(y_pred, y_true) (model.py:348-365): loss5 (device, 5 floats) = [total, xy,
 *                          wh, conf, class]; d total / d feature_map_i stays in the workspace
 *   y3_net_train_backward  the gradients of loss[0] w.r.t. every variable whose g_* offset is >= 0, written to
 *                          flat_grad + offset (train.py:112 compute_gradients; the L2 term, the per-tensor clip and the
 *                          update are y3_clip_update_multi's).  Gradients do not flow below the first layer that holds a
 *                          trainable variable (train.py:82 update_vars).  `ready(user, g_end)` is called right after the
 *                          kernels that complete layer i's gradients have been enqueued (layers are visited last to
 *                          first): the hook a data-parallel caller hangs its bucketed all-reduce on; may be NULL.
 *                          Re-runnable from the same forward / loss state.
 *   y3_net_train_step      forward + loss + backward in one call.
 * The net's dtype picks the kernels: 0 direct fp32, 4 Winograd forms of the stride-1 3x3 convs (forward, data AND weight
 * gradient), 2 / 3 products on the bf16 matrix pipe; 1 (bf16 storage) is rejected.  Variables are plain device pointers
 * in layer order (y3_net_layer_info): HWIO kernel, BN gamma / beta / moving mean / moving variance or the bias.
 * workspace: y3_net_train_workspace_bytes(net, vars, n, h, w) bytes (it depends on which variables are trainable),
 * 256-byte aligned, untouched by the caller between forward and backward.  Deterministic (fixed reduction orders). */
typedef struct y3_train_var {
    float* weights;
    float *gamma, *beta, *moving_mean, *moving_variance;   /* BN layers (NULL otherwise) */
    float* biases;                                         /* detection convs (NULL otherwise) */
    long long g_weights, g_gamma, g_beta, g_biases;        /* element offsets into flat_grad; < 0: not trainable */
    long long g_end;                                       /* passed to `ready` once this layer's gradients are enqueued; < 0: no call */
} y3_train_var;
typedef struct y3_train_opts {
    float bn_decay;                    /* model.py:36 batch_norm_decay */
    int use_label_smooth, use_focal_loss;
    const float* anchors;              /* HOST pointer: the 9 (w,h) anchor pairs, smallest first (utils/misc_utils.py:31-37) */
} y3_train_opts;
typedef void (*y3_grad_ready_fn)(void* user, long long g_end);
size_t y3_net_train_workspace_bytes(y3_net* net, const y3_train_var* vars, int n, int h, int w);
int y3_net_train_forward(y3_net* net, const y3_train_var* vars, const float* x, int n, int h, int w,
                         const y3_train_opts* opts, void* workspace, size_t workspace_bytes, float** fm1, float** fm2,
                         float** fm3);
int y3_net_train_loss(y3_net* net, const float* y_true_1, const float* y_true_2, const float* y_true_3,
                      const y3_train_opts* opts, float* loss5);
int y3_net_train_backward(y3_net* net, const y3_train_var* vars, float* flat_grad, y3_grad_ready_fn ready, void* user);
int y3_net_train_step(y3_net* net, const y3_train_var* vars, const float* x, int n, int h, int w, const float* y_true_1,
                      const float* y_true_2, const float* y_true_3, const y3_train_opts* opts, float* flat_grad,
                      void* workspace, size_t workspace_bytes, float* loss5, y3_grad_ready_fn ready, void* user);
/* Optional second stream for backward: a layer's weight gradient is off the critical path of the pass (only the optimizer
 * reads it), its data gradient and the BN backward below it are on it.  With a stream set here (a hipStream_t of the net's
 * device, not the context's; NULL switches it off) the weight gradients run there, ordered against the context's stream by
 * events, and overlap the HBM-bound BN passes of the next layer.  Everything the caller sees keeps its stream order: `ready`
 * for a layer is called once the context's stream has been made to wait for that layer's weight gradient, and backward
 * returns with the context's stream waiting for all of them.  The workspace is a little larger (one layer's dz lives one
 * layer longer): size it after this call.  Y3_OWN_STREAM: the library creates (once per net, destroyed with it) a stream of
 * the device's LOWEST priority for it - the recommended form: the weight gradient is the work that can wait, and a stream
 * of another priority cannot share the hardware queue of the context's stream (an ordinary stream of the caller's can,
 * once enough streams are alive in the process; the event waits then serialise inside one queue and the step gets slower
 * than on one stream). */
#define Y3_OWN_STREAM ((void*)(size_t)1)
int y3_net_train_set_wgrad_stream(y3_net* net, void* stream);
/* test hook: byte offsets inside the last forward's workspace of layer i's raw conv output z and of its [4][cout]
 * mean / inv_std / folded scale / folded shift (the tensors that fix the LeakyReLU branches); SIZE_MAX for non-BN layers */
int y3_net_train_saved(const y3_net* net, int layer, size_t* z_offset, size_t* stats_offset);

/* ---- next row 8(f)#1: target assignment on the device (utils/data_utils.py:51-115 `process_box`) ------------
 * boxes [n][kmax][5] = (x_min,y_min,x_max,y_max,mix_weight) in resized-image pixels, labels [n][kmax] int32,
 * counts [n] int32 (boxes actually present per image), anchors: 9 (w,h) pairs.  Writes the three y_true
 * tensors [n][g][g][3][6+C] (zero, mix weight 1, then the per-box entries in box order). */
int y3_process_box(y3_ctx* ctx, const float* boxes, const int32_t* labels, const int32_t* counts, int n, int kmax,
                   int class_num, int img_w, int img_h, const float* anchors_host18, float* y_true_13,
                   float* y_true_26, float* y_true_52);

/* ---- row 8(f)#1, the feeder's pixel work on the device (utils/data_utils.py:118-172 parse_data after its draws:
 * mix_up blend, random_color_distort, the crop window of the expanded canvas, cv2.resize with the drawn interpolation,
 * letterbox padding, random_flip, / 255) for a BATCH of samples.  The host half - the draws, the box arithmetic, Pillow's
 * double-precision filter windows and weights - is liby3feed.so's y3f_plan_batch (include/yolo355_feed.h), which writes one
 * relocatable blob per batch; the caller uploads it and passes its device address here, together with the host copy of
 * the n y3f_djob records at its start (launch geometry), the y3f_dtables uploaded once per device, and `scratch_bytes` >=
 * the plan's scratch.  out: [n][out_h][out_w][3] float32, the bytes y3f_sample writes (tests/test_feed_gpu.py).
 * Asynchronous on the context's stream; blob, tables and scratch must stay untouched until it has run. */
struct y3f_djob;
int y3_feed_run(y3_ctx* ctx, const void* blob_dev, const struct y3f_djob* jobs_host, int n, const void* tables_dev,
                void* scratch_dev, size_t scratch_bytes, float* out, int out_h, int out_w);

/* box_iou (model.py:307-345): pred_boxes [num_pred][4], true_boxes [num_true][4], both (cx,cy,w,h);
 * iou [num_pred][num_true] = inter / (area_p + area_t - inter + 1e-10). */
int y3_box_iou(y3_ctx* ctx, const float* pred_boxes, long long num_pred, const float* true_boxes, int num_true,
               float* iou);

/* K11: g <- g*grad_scale + weight_decay*w (slim.l2_regularizer, model.py:49) ; g <- tf.clip_by_norm(g, clip)
 * (train.py:113-114) ; TF1 update rule (utils/misc_utils.py:151-161).  kind: 0 sgd, 1 momentum (slot0 =
 * accumulator), 2 adam (slot0 = m, slot1 = v, decay = beta1, lr = lr_t), 3 rmsprop (slot0 = ms, slot1 = mom).
 * scratch: y3_optimizer_scratch_bytes(). */
size_t y3_optimizer_scratch_bytes(void);
int y3_clip_update(y3_ctx* ctx, int kind, float* w, float* g, float* slot0, float* slot1, long long n,
                   float weight_decay, float grad_scale, float clip_norm, float lr, float momentum, float decay,
                   float beta2, float eps, float* scratch);

/* The same K11 for ALL trainable tensors of a step in three launches (train.py:112-115 applies clip_by_norm and the
 * update to every (gradient, variable) pair): per-tensor norms come from a segmented, fixed-order reduction, so the
 * result is deterministic.  `params` is a HOST array (device pointers inside); it is copied to the scratch with an
 * asynchronous copy on the context's stream.  scratch: y3_clip_update_multi_scratch_bytes(params, count), 16-byte
 * aligned.  w and g of a tensor with n %% 4 == 0 must be 16-byte aligned. */
typedef struct y3_param_desc {
    float* w;            /* the variable */
    float* g;            /* its gradient (updated in place to the clipped gradient) */
    float* slot0;        /* optimizer slots as in y3_clip_update (NULL where the rule has none) */
    float* slot1;
    long long n;         /* elements */
    float weight_decay;  /* l2 coefficient (0 for BN parameters and biases) */
    int reserved;
} y3_param_desc;
size_t y3_clip_update_multi_scratch_bytes(const y3_param_desc* params, int count);
int y3_clip_update_multi(y3_ctx* ctx, int kind, const y3_param_desc* params, int count, float grad_scale,
                         float clip_norm, float lr, float momentum, float decay, float beta2, float eps,
                         void* scratch, size_t scratch_bytes);

#ifdef __cplusplus
}
#endif
#endif /* YOLO355_H */
