// Stand-alone (unfused) forms of the small graph ops, for callers that compose the network op by op
// through utils/layer_utils.py the way the reference's model.py does.  y3_net_forward never launches
// these: it folds the upsample/concat into the consuming 1x1 conv's loads and the residual add into the
// producing 3x3 conv's epilogue.  All are HBM-bound streaming kernels with 16-byte-per-lane accesses.
#include "y3_internal.h"

namespace {

// tf.image.resize_nearest_neighbor, align_corners=False (utils/layer_utils.py:82-87):
//   src = min(floor(dst * in / out), in - 1), scale computed in fp32 like TF's kernel.
__global__ void __launch_bounds__(256) upsample_nearest_kernel(const float* __restrict__ x,
                                                               float* __restrict__ y, int n, int h, int w,
                                                               int c4, int oh, int ow, float sy, float sx) {
    const long long total = (long long)n * oh * ow * c4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4);
        long long r = i / c4;
        const int ox = (int)(r % ow); r /= ow;
        const int oy = (int)(r % oh);
        const int b = (int)(r / oh);
        const int iy = min((int)floorf((float)oy * sy), h - 1);
        const int ix = min((int)floorf((float)ox * sx), w - 1);
        y4[i] = x4[(((long long)b * h + iy) * w + ix) * c4 + c];
    }
}

// tf.concat([a, b], axis=3) for NHWC (model.py:62,72)
__global__ void __launch_bounds__(256) concat_channels_kernel(const float* __restrict__ a,
                                                              const float* __restrict__ b,
                                                              float* __restrict__ y, long long rows, int ca4,
                                                              int cb4) {
    const int ct4 = ca4 + cb4;
    const long long total = rows * ct4;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % ct4);
        const long long r = i / ct4;
        y4[i] = c < ca4 ? a4[r * ca4 + c] : b4[r * cb4 + (c - ca4)];
    }
}

// net = net + shortcut (utils/layer_utils.py:30)
__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long n4) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        y4[i] = a4[i] + b4[i];
}

// reorg_layer box part for ONE scale (model.py:96-131): boxes [N,gh,gw,3,4] = (cx,cy,w,h) in input pixels.
__global__ void __launch_bounds__(256) reorg_boxes_kernel(const float* __restrict__ fm, float* __restrict__ boxes,
                                                          long long nboxes, int gh, int gw, int F,
                                                          float ratio_h, float ratio_w, float raw0, float rah0,
                                                          float raw1, float rah1, float raw2, float rah2) {
    for (long long gb = (long long)blockIdx.x * 256 + threadIdx.x; gb < nboxes; gb += (long long)gridDim.x * 256) {
        const int anc = (int)(gb % 3);
        const long long cell = (gb / 3) % ((long long)gh * gw);
        const int gy = (int)(cell / gw), gx = (int)(cell - (long long)gy * gw);
        const float* p = fm + gb * F;
        const float raw = anc == 0 ? raw0 : (anc == 1 ? raw1 : raw2);
        const float rah = anc == 0 ? rah0 : (anc == 1 ? rah1 : rah2);
        f32x4 o;
        o[0] = (1.f / (1.f + expf(-p[0])) + (float)gx) * ratio_w;
        o[1] = (1.f / (1.f + expf(-p[1])) + (float)gy) * ratio_h;
        o[2] = (expf(p[2]) * raw) * ratio_w;
        o[3] = (expf(p[3]) * rah) * ratio_h;
        *reinterpret_cast<f32x4*>(boxes + gb * 4) = o;
    }
}

inline int grid_for(long long work) {
    long long b = (work + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int y3_upsample_nearest(y3_ctx* ctx, const float* x, int n, int h, int w, int c, int out_h,
                                   int out_w, float* y) {
    Y3_CHECK_ARG(ctx && x && y, "y3_upsample_nearest: null argument");
    Y3_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && out_h > 0 && out_w > 0,
                 "y3_upsample_nearest: non-positive dimension");
    Y3_CHECK_ARG(c % 4 == 0, "y3_upsample_nearest: channels must be a multiple of 4 (got %d)", c);
    const float sy = (float)h / (float)out_h, sx = (float)w / (float)out_w;
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid_for((long long)n * out_h * out_w * (c / 4))),
                       dim3(256), 0, ctx->stream, x, y, n, h, w, c / 4, out_h, out_w, sy, sx);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_concat_channels(y3_ctx* ctx, const float* a, int ca, const float* b, int cb, long long rows,
                                  float* y) {
    Y3_CHECK_ARG(ctx && a && b && y, "y3_concat_channels: null argument");
    Y3_CHECK_ARG(ca > 0 && cb > 0 && rows > 0 && ca % 4 == 0 && cb % 4 == 0,
                 "y3_concat_channels: channel counts must be positive multiples of 4");
    hipLaunchKernelGGL(concat_channels_kernel, dim3(grid_for(rows * ((ca + cb) / 4))), dim3(256), 0,
                       ctx->stream, a, b, y, rows, ca / 4, cb / 4);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_add(y3_ctx* ctx, const float* a, const float* b, long long count, float* y) {
    Y3_CHECK_ARG(ctx && a && b && y, "y3_add: null argument");
    Y3_CHECK_ARG(count > 0 && count % 4 == 0, "y3_add: element count must be a positive multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(count / 4)), dim3(256), 0, ctx->stream, a, b, y, count / 4);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_reorg_boxes(y3_ctx* ctx, const float* fm, int n, int gh, int gw, int class_num, int img_h,
                              int img_w, const float* anchors3_host, float* boxes) {
    Y3_CHECK_ARG(ctx && fm && anchors3_host && boxes, "y3_reorg_boxes: null argument");
    Y3_CHECK_ARG(n > 0 && gh > 0 && gw > 0 && class_num > 0 && img_h > 0 && img_w > 0,
                 "y3_reorg_boxes: non-positive dimension");
    const float ratio_h = (float)((double)img_h / (double)gh);
    const float ratio_w = (float)((double)img_w / (double)gw);
    float ra[6];
    for (int k = 0; k < 3; ++k) {
        ra[2 * k] = anchors3_host[2 * k] / ratio_w;
        ra[2 * k + 1] = anchors3_host[2 * k + 1] / ratio_h;
    }
    const long long nboxes = (long long)n * gh * gw * 3;
    hipLaunchKernelGGL(reorg_boxes_kernel, dim3(grid_for(nboxes)), dim3(256), 0, ctx->stream, fm, boxes,
                       nboxes, gh, gw, 5 + class_num, ratio_h, ratio_w, ra[0], ra[1], ra[2], ra[3], ra[4], ra[5]);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
