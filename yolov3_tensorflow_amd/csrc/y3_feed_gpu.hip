// y3_feed_run: the pixel half of the feeder on the device (include/yolo355_feed.h "the device form").
// The reference does this work per image in OpenCV inside tf.data's py_func workers (utils/data_utils.py:118-172); here a
// whole batch is three launches beside the train step:
//   feed_window_kernel      blend (mix-up) + colour jitter of every live pixel of every crop window      -> scratch `win`
//   feed_horizontal_kernel  Pillow's horizontal 8-bit pass for the CUBIC / AREA / LANCZOS4 jobs           -> scratch `tmp`
//   feed_output_kernel      vertical pass / NEAREST / LINEAR, pad, mirror, / 255                           -> the float32 batch
// blockIdx.y = job; a job's pixels are walked by the x-blocks with a grid stride.  Byte gathers from tables that sit in the
// L2 (the conversion tables are 260 KB, a job's coefficient tables a few KB): HBM traffic is the source pixels once and the
// batch once - microseconds; nothing here is worth an LDS stage.  Per-pixel arithmetic: y3_feed_px.h, shared with the host
// build of the tests; compiled without FMA contraction.
#include <algorithm>
#include "y3_internal.h"
#include "y3_feed_px.h"

namespace {

__global__ void __launch_bounds__(256) feed_window_kernel(const uint8_t* __restrict__ blob, const y3f_dtables* __restrict__ T,
                                                          uint8_t* __restrict__ scratch) {
    const y3f_djob& d = reinterpret_cast<const y3f_djob*>(blob)[blockIdx.y];
    const int lw = d.live_x1 - d.live_x0;
    const long long total = (long long)lw * (d.live_y1 - d.live_y0);
    uint8_t* win = scratch + d.win_off;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        uint8_t px[3];
        y3fpx::window_pixel(d, blob, *T, d.live_x0 + (int)(i % lw), d.live_y0 + (int)(i / lw), px);
        win[3 * i] = px[0], win[3 * i + 1] = px[1], win[3 * i + 2] = px[2];
    }
}

__global__ void __launch_bounds__(256) feed_horizontal_kernel(const uint8_t* __restrict__ blob, uint8_t* __restrict__ scratch) {
    const y3f_djob& d = reinterpret_cast<const y3f_djob*>(blob)[blockIdx.y];
    if (d.mode != Y3F_MODE_RESAMPLE || !d.horizontal) return;
    const long long total = (long long)d.tmp_rows * d.res_w;
    const uint8_t* win = scratch + d.win_off;
    uint8_t* tmp = scratch + d.tmp_off;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        uint8_t px[3];
        y3fpx::horizontal_pixel(d, blob, win, (int)(i / d.res_w), (int)(i % d.res_w), px);
        tmp[3 * i] = px[0], tmp[3 * i + 1] = px[1], tmp[3 * i + 2] = px[2];
    }
}

__global__ void __launch_bounds__(256) feed_output_kernel(const uint8_t* __restrict__ blob, const y3f_dtables* __restrict__ T,
                                                          const uint8_t* __restrict__ scratch, float* __restrict__ out) {
    const y3f_djob& d = reinterpret_cast<const y3f_djob*>(blob)[blockIdx.y];
    const long long total = (long long)d.out_h * d.out_w;
    float* o = out + (size_t)blockIdx.y * total * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float px[3];
        y3fpx::output_pixel(d, blob, scratch + d.win_off, scratch + d.tmp_off, *T, (int)(i % d.out_w), (int)(i / d.out_w), px);
        o[3 * i] = px[0], o[3 * i + 1] = px[1], o[3 * i + 2] = px[2];
    }
}

inline unsigned blocks_for(long long work) {
    const long long b = (work + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace

extern "C" int y3_feed_run(y3_ctx* ctx, const void* blob_dev, const y3f_djob* jobs_host, int n, const void* tables_dev,
                           void* scratch_dev, size_t scratch_bytes, float* out, int out_h, int out_w) {
    Y3_CHECK_ARG(ctx && blob_dev && jobs_host && tables_dev && out, "y3_feed_run: null argument");
    Y3_CHECK_ARG(n > 0 && n <= 65535 && out_h > 0 && out_w > 0, "y3_feed_run: bad job count or output size");
    long long win_px = 0, hor_px = 0;
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        const y3f_djob& d = jobs_host[i];
        Y3_CHECK_ARG(d.out_h == out_h && d.out_w == out_w, "y3_feed_run: job %d writes %dx%d, the batch is %dx%d", i, d.out_w,
                     d.out_h, out_w, out_h);
        const long long live = (long long)(d.live_x1 - d.live_x0) * (d.live_y1 - d.live_y0);
        const long long hor = (d.mode == Y3F_MODE_RESAMPLE && d.horizontal) ? (long long)d.tmp_rows * d.res_w : 0;
        win_px = live > win_px ? live : win_px;
        hor_px = hor > hor_px ? hor : hor_px;
        need = std::max(need, std::max((size_t)d.win_off + (size_t)live * 3, (size_t)d.tmp_off + (size_t)hor * 3));
    }
    Y3_CHECK_ARG(need <= scratch_bytes && (need == 0 || scratch_dev), "y3_feed_run: the jobs need %zu bytes of scratch, %zu given",
                 need, scratch_bytes);
    const uint8_t* blob = static_cast<const uint8_t*>(blob_dev);
    const y3f_dtables* T = static_cast<const y3f_dtables*>(tables_dev);
    uint8_t* scratch = static_cast<uint8_t*>(scratch_dev);
    if (win_px > 0) hipLaunchKernelGGL(feed_window_kernel, dim3(blocks_for(win_px), n), dim3(256), 0, ctx->stream, blob, T, scratch);
    if (hor_px > 0) hipLaunchKernelGGL(feed_horizontal_kernel, dim3(blocks_for(hor_px), n), dim3(256), 0, ctx->stream, blob, scratch);
    hipLaunchKernelGGL(feed_output_kernel, dim3(blocks_for((long long)out_h * out_w), n), dim3(256), 0, ctx->stream, blob, T,
                       scratch, out);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
