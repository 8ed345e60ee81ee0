// Internal declarations shared by the HIP translation units of libyolo355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/yolo355.h"

struct y3_ctx {
    int device;
    hipStream_t stream;
};

void y3_set_error(const char* fmt, ...);

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
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Launchers implemented in the .hip files (all asynchronous on `stream`).
int y3_launch_conv(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* x_up,
                   const float* w, const float* scale, const float* shift, const float* residual,
                   float* y, void* workspace, size_t workspace_bytes, hipEvent_t mid_event = nullptr);
size_t y3_conv_workspace_bytes_impl(const y3_conv_desc* d);
int y3_conv_schedule_impl(const y3_conv_desc* d);
// host evaluation of the device-side stream-K work split (test hook): kind 0 = sk_range (direct / split kernels,
// y3_conv_common.h), 1 = wk_range (Winograd kernel)
int y3_streamk_range_impl(int kind, int units, int ksteps, int workers, int group, int local_worker, long long* begin,
                          long long* end);
void y3_wino_range_impl(int units, int ksteps, int workers, int group, int local_worker, long long* begin, long long* end);
int y3_launch_conv_dgrad(hipStream_t stream, const y3_conv_desc* fwd, const float* dz, int dz_stride,
                         const float* w_d, const float* ones, const float* zeros, int accumulate, float* dx,
                         void* workspace, size_t workspace_bytes);
int y3_launch_conv_bf16(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                        const float* scale, const float* shift, const void* residual, void* y, int out_f32);
int y3_launch_pack_split(hipStream_t stream, const float* w_hwio, int k, int cin, int cout, int planes, void* out,
                         int transposed = 0);
int y3_launch_conv_dgrad_split(hipStream_t stream, const y3_conv_desc* fwd, int planes, const float* dz, int dz_stride,
                               const void* w, const float* ones, const float* zeros, int accumulate, float* dx,
                               void* workspace, size_t workspace_bytes);
int y3_launch_conv_split(hipStream_t stream, const y3_conv_desc* d, int planes, const float* x, const float* x_up,
                         const void* w, const float* scale, const float* shift, const float* residual, float* y,
                         void* workspace, size_t workspace_bytes, hipEvent_t mid_event = nullptr);
int y3_conv_wino_eligible_impl(const y3_conv_desc* d);
int y3_launch_pack_wino(hipStream_t stream, const float* w_hwio, int cin, int cout, float* out);
size_t y3_conv_wino_workspace_bytes_impl(const y3_conv_desc* d);
int y3_launch_conv_wino(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* u, const float* scale,
                        const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes,
                        hipEvent_t mid_event = nullptr);
