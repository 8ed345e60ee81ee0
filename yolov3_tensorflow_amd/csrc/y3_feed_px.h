// The per-pixel functions of the feeder's device form (include/yolo355_feed.h: y3f_djob): what y3f_sample does to ONE pixel
// of the window, of the horizontal pass and of the output, written once for the GPU kernels (y3_feed_gpu.hip) and for the
// host build tests/test_feed_plan.py runs against y3f_sample without a GPU.  Integer arithmetic, table look-ups and three
// spots of floating point that must round like the host library's (no FMA contraction: both builds pass
// -ffp-contract=off):
//   blend      (uint8)(int)( (float)a * lam1  [+ (float)b * lam2] )                 float32, as numpy multiplies
//   hsv -> rgb (int)( (float)v * (1.0 - fs [* f | * (1.0 - f)]) + 0.5 )             double, as Pillow's C code
//   / 255      a 256-entry table filled by the host
#pragma once
#include <stdint.h>
#include "../../include/yolo355_feed.h"

#ifdef __HIPCC__
#define Y3F_HD __host__ __device__ __forceinline__
#else
#define Y3F_HD inline
#endif

namespace y3fpx {

Y3F_HD uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

constexpr int kCoefBits = 32 - 8 - 2;      // Pillow's PRECISION_BITS

// ---- 1. one live pixel of the window: blend (mix-up), jitter (random_color_distort after its draws) -------------------
// (wx, wy) in window coordinates, inside the live rectangle.  y3_feed.cpp: run_job step 1 + colour_run.
Y3F_HD void window_pixel(const y3f_djob& d, const uint8_t* blob, const y3f_dtables& T, int wx, int wy, uint8_t out[3]) {
    const int ix = wx + d.img_dx, iy = wy + d.img_dy;
    const int ax = ix - d.r1_x0, ay = iy - d.r1_y0;
    const bool in1 = ax >= 0 && ax < d.r1_w && ay >= 0 && ay < d.r1_h;
    const uint8_t* a = blob + d.img1_off + ((size_t)ay * d.r1_w + ax) * 3;
    int r, g, b;
    if (!d.has2) {
        r = a[0], g = a[1], b = a[2];          // (the live rectangle of a single image lies inside it)
    } else {
        const int bx = ix - d.r2_x0, by = iy - d.r2_y0;
        const bool in2 = bx >= 0 && bx < d.r2_w && by >= 0 && by < d.r2_h;
        const uint8_t* p2 = blob + d.img2_off + ((size_t)by * d.r2_w + bx) * 3;
        int px[3];
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            if (in1) acc = (float)a[c] * d.lam1;
            if (in2) acc = acc + (float)p2[c] * d.lam2;
            px[c] = (uint8_t)(int)acc;
        }
        r = px[0], g = px[1], b = px[2];
    }
    if (!d.colour_on) {
        out[0] = (uint8_t)r, out[1] = (uint8_t)g, out[2] = (uint8_t)b;
        return;
    }
    const uint8_t* J = blob + d.jitter_off;     // bright, h, s, v maps
    r = J[r], g = J[g], b = J[b];
    const int mx = r > g ? (r > b ? r : b) : (g > b ? g : b);
    const int mn = r < g ? (r < b ? r : b) : (g < b ? g : b);
    int h = 0, s = 0;
    if (mx != mn) {
        h = r == mx ? T.hue[0][mx - g][mx - b] : (g == mx ? T.hue[1][mx - r][mx - b] : T.hue[2][mx - r][mx - g]);
        s = T.sat[mx][mn];
    }
    h = J[256 + h];
    s = J[512 + s];
    const int v = J[768 + mx];
    if (s == 0) {
        out[0] = out[1] = out[2] = (uint8_t)v;
        return;
    }
    const float f = T.frac[h], fs = T.unit[s];
    const int sec = T.sector[h];
    const uint8_t u = (uint8_t)v;
    const uint8_t p = clamp_u8((int)((float)v * (1.0 - fs) + 0.5));
    const uint8_t qt = (sec & 1) ? clamp_u8((int)((float)v * (1.0 - fs * f) + 0.5))
                                 : clamp_u8((int)((float)v * (1.0 - fs * (1.0 - f)) + 0.5));
    switch (sec) {
        case 0: out[0] = u; out[1] = qt; out[2] = p; break;
        case 1: out[0] = qt; out[1] = u; out[2] = p; break;
        case 2: out[0] = p; out[1] = u; out[2] = qt; break;
        case 3: out[0] = p; out[1] = qt; out[2] = u; break;
        case 4: out[0] = qt; out[1] = p; out[2] = u; break;
        default: out[0] = u; out[1] = p; out[2] = qt; break;
    }
}

// the window as the resize sees it: black canvas outside the live rectangle.  `win` holds the live part, rows packed.
Y3F_HD const uint8_t* live_ptr(const y3f_djob& d, const uint8_t* win, int x, int y) {
    return win + ((size_t)(y - d.live_y0) * (d.live_x1 - d.live_x0) + (x - d.live_x0)) * 3;
}
Y3F_HD bool is_live(const y3f_djob& d, int x, int y) {
    return x >= d.live_x0 && x < d.live_x1 && y >= d.live_y0 && y < d.live_y1;
}
Y3F_HD void win_px(const y3f_djob& d, const uint8_t* win, int x, int y, int px[3]) {
    if (is_live(d, x, y)) {
        const uint8_t* p = live_ptr(d, win, x, y);
        px[0] = p[0], px[1] = p[1], px[2] = p[2];
    } else {
        px[0] = px[1] = px[2] = 0;
    }
}

// ---- 2. one pixel of Pillow's horizontal pass: window row tmp_y0 + t, output column x ----------------------------------
// y3_feed.cpp: resample(), first loop (zero pixels add nothing to the sums: only live columns are visited).
Y3F_HD void horizontal_pixel(const y3f_djob& d, const uint8_t* blob, const uint8_t* win, int t, int x, uint8_t out[3]) {
    const int32_t* tab = reinterpret_cast<const int32_t*>(blob + d.xtab_off);
    const int first = tab[x], count = tab[d.res_w + x];
    const int32_t* k = tab + 2 * (size_t)d.res_w + (size_t)x * d.ksize_x;
    const int lo = first > d.live_x0 ? first : d.live_x0;
    int hi = first + count < d.live_x1 ? first + count : d.live_x1;
    if (hi < lo) hi = lo;
    const uint8_t* p = live_ptr(d, win, lo, d.tmp_y0 + t);
    int32_t s0 = 1 << (kCoefBits - 1), s1 = s0, s2 = s0;
    for (int i = lo - first, n = hi - first; i < n; ++i, p += 3) {
        s0 += p[0] * k[i];
        s1 += p[1] * k[i];
        s2 += p[2] * k[i];
    }
    out[0] = clamp_u8(s0 >> kCoefBits);
    out[1] = clamp_u8(s1 >> kCoefBits);
    out[2] = clamp_u8(s2 >> kCoefBits);
}

// ---- 3. one pixel of the resized image (rx, ry) ------------------------------------------------------------------------
Y3F_HD void resized_pixel(const y3f_djob& d, const uint8_t* blob, const uint8_t* win, const uint8_t* tmp, int rx, int ry,
                          int out[3]) {
    const int32_t* xt = reinterpret_cast<const int32_t*>(blob + d.xtab_off);
    const int32_t* yt = reinterpret_cast<const int32_t*>(blob + d.ytab_off);
    switch (d.mode) {
        case Y3F_MODE_COPY: win_px(d, win, rx, ry, out); return;
        case Y3F_MODE_NEAREST: win_px(d, win, xt[rx], yt[ry], out); return;         // y3_feed.cpp: resize_nearest
        case Y3F_MODE_MEAN2X2: {                                                    // resize_linear's 2x2 reduction
            int a[3], b[3], c[3], e[3];
            win_px(d, win, 2 * rx, 2 * ry, a);
            win_px(d, win, 2 * rx + 1, 2 * ry, b);
            win_px(d, win, 2 * rx, 2 * ry + 1, c);
            win_px(d, win, 2 * rx + 1, 2 * ry + 1, e);
            for (int q = 0; q < 3; ++q) out[q] = (uint8_t)((a[q] + b[q] + c[q] + e[q] + 2) >> 2);
            return;
        }
        case Y3F_MODE_LINEAR: {                                                     // resize_linear: 11-bit weights
            const int n = d.res_w, m = d.res_h;
            const int xl = xt[rx], xh = xt[n + rx], wa = xt[2 * n + rx], wb = xt[3 * n + rx];
            const int yl = yt[ry], yh = yt[m + ry], b0 = yt[2 * m + ry], b1 = yt[3 * m + ry];
            int a0[3], a1[3], c0[3], c1[3];
            win_px(d, win, xl, yl, a0);
            win_px(d, win, xh, yl, a1);
            win_px(d, win, xl, yh, c0);
            win_px(d, win, xh, yh, c1);
            for (int q = 0; q < 3; ++q) {
                const int32_t s0 = a0[q] * wa + a1[q] * wb, s1 = c0[q] * wa + c1[q] * wb;
                out[q] = clamp_u8((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2);
            }
            return;
        }
        default: break;
    }
    // Pillow's vertical pass over the horizontal pass (or over the window itself when the width does not change)
    const size_t stride = (size_t)d.res_w * 3;
    if (!d.vertical) {                      // (horizontal only: the pass IS the result; rows outside it are black canvas)
        const int t = ry - d.tmp_y0;
        if (t >= 0 && t < d.tmp_rows) {
            const uint8_t* p = tmp + (size_t)t * stride + (size_t)rx * 3;
            out[0] = p[0], out[1] = p[1], out[2] = p[2];
        } else {
            out[0] = out[1] = out[2] = 0;
        }
        return;
    }
    const int first = yt[ry], count = yt[d.res_h + ry];
    const int32_t* k = yt + 2 * (size_t)d.res_h + (size_t)ry * d.ksize_y;
    const int lo = first > d.live_y0 ? first : d.live_y0;
    const int hi = first + count < d.live_y1 ? first + count : d.live_y1;
    int32_t s0 = 1 << (kCoefBits - 1), s1 = s0, s2 = s0;
    if (d.horizontal) {
        for (int r = lo; r < hi; ++r) {
            const uint8_t* p = tmp + (size_t)(r - d.tmp_y0) * stride + (size_t)rx * 3;
            const int32_t kt = k[r - first];
            s0 += p[0] * kt, s1 += p[1] * kt, s2 += p[2] * kt;
        }
    } else if (rx >= d.live_x0 && rx < d.live_x1) {
        for (int r = lo; r < hi; ++r) {
            const uint8_t* p = live_ptr(d, win, rx, r);
            const int32_t kt = k[r - first];
            s0 += p[0] * kt, s1 += p[1] * kt, s2 += p[2] * kt;
        }
    }
    out[0] = clamp_u8(s0 >> kCoefBits);
    out[1] = clamp_u8(s1 >> kCoefBits);
    out[2] = clamp_u8(s2 >> kCoefBits);
}

// ---- 4. one pixel of the network's input: pad, mirror, / 255 (y3_feed.cpp: run_job step 3) ------------------------------
Y3F_HD void output_pixel(const y3f_djob& d, const uint8_t* blob, const uint8_t* win, const uint8_t* tmp, const y3f_dtables& T,
                         int x, int y, float out[3]) {
    const int ux = d.flip_x ? d.out_w - 1 - x : x;
    const int rx = ux - d.pad_x, ry = y - d.pad_y;
    int px[3] = {d.pad_value, d.pad_value, d.pad_value};
    if (rx >= 0 && rx < d.res_w && ry >= 0 && ry < d.res_h) resized_pixel(d, blob, win, tmp, rx, ry, px);
    out[0] = T.unit255[px[0]], out[1] = T.unit255[px[1]], out[2] = T.unit255[px[2]];
}

}  // namespace y3fpx
