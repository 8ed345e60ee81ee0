// Test infrastructure: the device functions of the feeder's pixel work (yolov3_tensorflow_amd/csrc/y3_feed_px.h) run on
// the HOST in the order y3_feed_run's three kernels run them, so that the planner (y3f_plan_batch) and the per-pixel
// arithmetic can be compared with y3f_sample without a GPU (tests/test_feed_plan.py builds this file with g++).
// Never part of the product: the package has no CPU form of y3_feed_run.
#include <cstring>
#include "../yolov3_tensorflow_amd/csrc/y3_feed_px.h"

extern "C" int y3f_emulate(const uint8_t* blob, int n, const y3f_dtables* T, uint8_t* scratch, float* out) {
    const y3f_djob* jobs = reinterpret_cast<const y3f_djob*>(blob);
    for (int j = 0; j < n; ++j) {
        const y3f_djob& d = jobs[j];
        uint8_t* win = scratch + d.win_off;
        uint8_t* tmp = scratch + d.tmp_off;
        const int lw = d.live_x1 - d.live_x0, lh = d.live_y1 - d.live_y0;
        for (long long i = 0; i < (long long)lw * lh; ++i)
            y3fpx::window_pixel(d, blob, *T, d.live_x0 + (int)(i % lw), d.live_y0 + (int)(i / lw), win + 3 * i);
        if (d.mode == Y3F_MODE_RESAMPLE && d.horizontal)
            for (long long i = 0; i < (long long)d.tmp_rows * d.res_w; ++i)
                y3fpx::horizontal_pixel(d, blob, win, (int)(i / d.res_w), (int)(i % d.res_w), tmp + 3 * i);
        float* o = out + (size_t)j * d.out_h * d.out_w * 3;
        for (long long i = 0; i < (long long)d.out_h * d.out_w; ++i)
            y3fpx::output_pixel(d, blob, win, tmp, *T, (int)(i % d.out_w), (int)(i / d.out_w), o + 3 * i);
    }
    return 0;
}
