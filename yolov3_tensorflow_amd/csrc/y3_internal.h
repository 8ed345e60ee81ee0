// Internal declarations shared by the HIP translation units of libyolo355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/yolo355.h"

struct y3_ctx {
    int device;
    hipStream_t stream;
    unsigned* err_host;   // one pinned, device-visible word: kernels OR a non-zero code into it when a stream-K
                          // hand-off times out; read (no sync) at the next call on the context and by y3_ctx_check
    // pinned staging for small host->device descriptor uploads (y3_clip_update_multi): the buffer is rewritten only
    // after `stage_ev` (recorded behind the previous upload) has completed, so an in-flight copy never reads a
    // buffer the host is rewriting, whatever the runtime does with pageable memory
    void* stage_host = nullptr;
    size_t stage_bytes = 0;
    hipEvent_t stage_ev = nullptr;
    bool stage_busy = false;
};
// returns a pinned buffer of at least `bytes` that no earlier upload is still reading (waits for it if needed)
int y3_ctx_stage_acquire(y3_ctx* ctx, size_t bytes, void** out);
// records that an asynchronous copy out of the staging buffer was just enqueued on ctx->stream
int y3_ctx_stage_release(y3_ctx* ctx);

// stream-K plumbing handed down to the conv launchers by the ctx entry points / y3_net_forward
struct y3_sk_opts {
    unsigned* flags = nullptr;  // pre-zeroed "partial published" words for THIS launch (y3_net_forward zeroes the flag
                                // regions of all its layers with one memset); nullptr: the launcher uses the words
                                // inside the workspace and zeroes them itself ahead of the launch
    unsigned* err = nullptr;    // device-visible error word (y3_ctx::err_host) or nullptr
    float* stats = nullptr;     // [y3_conv_stats_blocks][2][cout] column sums of y / y^2 per output row block, or nullptr
    // y3_launch_conv_dgrad of a 1x1 conv whose output is the dy of a BN layer: that layer's z and its [4][C] mean / inv_std /
    // scale / shift; `stats` then receives [y3_conv_dgrad_stats_blocks][2][C] partial sums of g' and g' * zhat (the BN backward
    // reduction, y3_bn_train_bwd_partials consumes them)
    const float* bwd_z = nullptr;
    const float* bwd_vec = nullptr;
};
// rows of the `stats` output of the exact-fp32 conv launchers for this conv (0: not available); wino != 0: the Winograd kernel
int y3_conv_stats_blocks_impl(const y3_conv_desc* d, int wino);
// rows of the fused BN-backward partial sums of the 1x1 data gradient of forward layer `fwd` (0: not available)
int y3_conv_dgrad_stats_blocks_impl(const y3_conv_desc* fwd);
// BN backward with the reduction already done: partial = [nblocks][2][c] sums of g' and g' * zhat (scratch: y3_bn_bwd_scratch_bytes)
int y3_bn_train_bwd_partials(y3_ctx* ctx, const float* z, const float* dy, const float* gamma, const float* scale, const float* shift,
                             const float* mean, const float* inv_std, long long rows, int c, const float* partial, int nblocks,
                             float* dgamma, float* dbeta, float* dz, float* scratch);
#define Y3_ERR_STREAMK_TIMEOUT 1u
// Test hook: with Y3_STREAMK_FAULT=1 in the environment the producers of a stream-K launch never raise their flag and
// the consumers give up after 2^10 polls, so the time-out path (error word -> Y3_EHIP) can be exercised.
void y3_sk_debug_env(unsigned* spin_limit, int* fault);

// Experiment switches (A/B runs of tools/ and tests/test_conv_variants_gpu.py) exist only in a -DY3_EXPERIMENTS build
// (csrc/libyolo355_exp.so, `python -m yolov3_tensorflow_amd.build --experiments`): in the product library no environment
// variable changes which kernel runs.
#ifdef Y3_EXPERIMENTS
#include <cstdlib>
static inline const char* y3_exp_env(const char* name) { return getenv(name); }
#else
static inline const char* y3_exp_env(const char*) { return nullptr; }
#endif

void y3_set_error(const char* fmt, ...);

// The process may drive several devices: "hipFuncSetAttribute done" flags are kept per (kernel instantiation, device).
constexpr int Y3_MAX_DEVICES = 16;
int y3_current_device();            // hipGetDevice, or -1 (no device / beyond Y3_MAX_DEVICES: callers then set the attribute every time)
size_t y3_device_max_lds();         // dynamic LDS one workgroup may ask for on the current device (cached per device); 0: unknown

#define Y3_CHECK_ARG(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            y3_set_error(__VA_ARGS__);         \
            return Y3_EINVAL;                  \
        }                                      \
    } while (0)

#define Y3_CHECK_HIP(expr)                                                          \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            y3_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                         __FILE__, __LINE__);                                       \
            return Y3_EHIP;                                                         \
        }                                                                           \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));   // a float4 at a 4-byte-aligned address (one dwordx4 access)
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Launchers implemented in the .hip files (all asynchronous on `stream`).
int y3_launch_conv(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* x_up,
                   const float* w, const float* scale, const float* shift, const float* residual,
                   float* y, void* workspace, size_t workspace_bytes, const y3_sk_opts* sk = nullptr);
size_t y3_conv_workspace_bytes_impl(const y3_conv_desc* d);
int y3_conv_schedule_impl(const y3_conv_desc* d);
// host evaluation of the device-side stream-K work split (test hook): kind 0 = sk_range (direct / split kernels,
// y3_conv_common.h), 1 = wk_range (Winograd kernel, every block may be cut), 2 = wk_range hybrid (whole rounds first)
int y3_streamk_range_impl(int kind, int units, int ksteps, int workers, int group, int local_worker, long long* begin,
                          long long* end);
void y3_wino_range_impl(int units, int ksteps, int workers, int group, int local_worker, int hybrid, long long* begin,
                        long long* end);
int y3_launch_conv_dgrad(hipStream_t stream, const y3_conv_desc* fwd, const float* dz, int dz_stride,
                         const float* w_d, const float* ones, const float* zeros, int accumulate, float* dx,
                         void* workspace, size_t workspace_bytes, const y3_sk_opts* sk = nullptr);
int y3_launch_conv_bf16(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                        const float* scale, const float* shift, const void* residual, void* y, int out_f32);
int y3_conv_bf16x_takes(int k, int cin);
int y3_launch_pack_bf16x(hipStream_t stream, const float* w_hwio, int k, int cin, int cout, void* w_packed);
int y3_launch_conv_bf16x(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                         const float* scale, const float* shift, const void* residual, void* y, int out_f32);
// fp32 path: the 3 -> 32 stem and the stride-2 32 -> 64 conv behind it in one kernel (y3_conv_f32s.hip)
int y3_conv_f32_stem_s2_takes(const y3_conv_desc* d0, const y3_conv_desc* d1);
int y3_launch_conv_f32_stem_s2(hipStream_t stream, int n, int h, int w, const float* x, const float* w0, const float* scale0,
                               const float* shift0, int act0, const float* w1_packed, const float* scale1, const float* shift1,
                               int act1, float* y);
// bf16 path: the 3 -> 32 stem and the stride-2 32 -> 64 conv behind it in one kernel (y3_conv_bf16s.hip)
int y3_conv_bf16_stem_s2_takes(const y3_conv_desc* d0, const y3_conv_desc* d1);
int y3_launch_conv_bf16_stem_s2(hipStream_t stream, int n, int h, int w, const float* x, const float* w0, const float* scale0,
                                const float* shift0, int act0, const void* w1_packed, const float* scale1, const float* shift1,
                                int act1, void* y);
// bf16 path: the first residual block (1x1 64 -> 32, 3x3 32 -> 64, + shortcut) in one kernel (y3_conv_bf16b.hip)
int y3_conv_bf16_resblock64_takes(const y3_conv_desc* d2, const y3_conv_desc* d3);
int y3_launch_conv_bf16_resblock64(hipStream_t stream, int n, int h, int w, const void* x, const void* w2_packed,
                                   const float* scale2, const float* shift2, int act2, const void* w3_packed, const float* scale3,
                                   const float* shift3, int act3, void* y);
// persistent LDS-DMA ring kernel for the bf16 1x1 convs (y3_conv_bf16r.hip); weights in the bf16x packing with one tap
int y3_conv_bf16r_takes(int k, int cin);
int y3_conv_bf16r_tile(const y3_conv_desc* d);
int y3_launch_conv_bf16r(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                         const float* scale, const float* shift, const void* residual, void* y, int out_f32);
int y3_launch_pack_split(hipStream_t stream, const float* w_hwio, int k, int cin, int cout, int planes, void* out,
                         int transposed = 0);
int y3_launch_conv_dgrad_split(hipStream_t stream, const y3_conv_desc* fwd, int planes, const float* dz, int dz_stride,
                               const void* w, const float* ones, const float* zeros, int accumulate, float* dx,
                               void* workspace, size_t workspace_bytes, const y3_sk_opts* sk = nullptr);
int y3_launch_conv_split(hipStream_t stream, const y3_conv_desc* d, int planes, const float* x, const float* x_up,
                         const void* w, const float* scale, const float* shift, const float* residual, float* y,
                         void* workspace, size_t workspace_bytes, const y3_sk_opts* sk = nullptr);
int y3_conv_wino_eligible_impl(const y3_conv_desc* d);
// F(4x4,3x3) inference kernel (y3_conv_wino44.hip)
int y3_conv_wino44_eligible_impl(const y3_conv_desc* d);
int y3_conv_wino44_candidate_impl(const y3_conv_desc* d);   // by shape (what to pack an alternative kernel for)
int y3_conv_wino44_preferred_impl(const y3_conv_desc* d);   // for this launch (candidate + enough blocks to fill the CUs)
int y3_launch_pack_wino44(hipStream_t stream, const float* w_hwio, int cin, int cout, float* out, int dgrad = 0);
int y3_conv_wino44_stats_blocks_impl(const y3_conv_desc* d);   // rows of the STATS output (one per 16-tile block)
size_t y3_conv_wino44_workspace_bytes_impl(const y3_conv_desc* d);   // bytes of V = B^T d B (the two-kernel form's workspace)
int y3_conv_wino44_two_pass_impl(const y3_conv_desc* d);          // the form a launch with a workspace takes (1 = two kernels)
int y3_launch_conv_wino44(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* u, const float* scale,
                          const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes,
                          const y3_sk_opts* sk);
int y3_conv_wgrad_wino_eligible_impl(const y3_conv_desc* d);
size_t y3_conv_wgrad_wino_scratch_bytes_impl(const y3_conv_desc* d);
int y3_launch_conv_wgrad_wino(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* dz, int dz_stride,
                              float* dw_hwio, void* scratch, size_t scratch_bytes);
int y3_launch_pack_wino(hipStream_t stream, const float* w_hwio, int cin, int cout, float* out, int dgrad = 0);
size_t y3_conv_wino_workspace_bytes_impl(const y3_conv_desc* d);
int y3_launch_conv_wino(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* u, const float* scale,
                        const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes,
                        const y3_sk_opts* sk = nullptr);
