// liby3feed.so - the per-image CPU work of the feeder behind include/yolo355_feed.h (host code: g++, no HIP).
//
// What it computes is DEFINED by yolov3_tensorflow_amd/utils/data_aug.py + data_utils.py (numpy / Pillow), which in turn
// follow the reference's utils/data_aug.py; this file produces the same bytes in one pass over the pixels that survive the
// crop.  Floating point here must round exactly like numpy's float32 loops and Pillow's C code do, so the build uses
// -ffp-contract=off and no fast-math, and every expression below keeps the operand types of the code it restates.
//
//   resize_nearest / resize_linear   OpenCV's uint8 INTER_NEAREST / INTER_LINEAR as utils/data_utils.py restates them
//   resample (box / bicubic / lanczos)  Pillow's two-pass 8-bit resampling (22-bit fixed-point coefficients)
//   rgb_to_hsv_px / hsv_to_rgb_px    Pillow's Image.convert('HSV') / convert('RGB')
//   colour_run                        random_color_distort after its draws (utils/data_aug.py:228-271 of the reference)
//   y3f_sample                        window of the mixed / expanded canvas -> jitter -> resize -> pad -> flip -> /255
#include "../../include/yolo355_feed.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

// The integer resize loops are compiled three times (AVX2, SSE4.1, baseline x86-64) and picked at load time by the CPU: the
// library is built in one container and runs on another host.  Integer arithmetic only - every version gives the same bytes.
#define Y3F_CLONES __attribute__((target_clones("avx2", "sse4.1", "default")))

static_assert(sizeof(y3f_colour) == 24 && sizeof(y3f_job) == 128, "the ctypes mirrors in feed_native.py assume this layout");

namespace {

thread_local char g_error[256] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---------------------------------------------------------------------------------------------------------------------
// OpenCV INTER_NEAREST: src index = min(floor(dst index * src / dst), src - 1), no half-pixel centre
// ---------------------------------------------------------------------------------------------------------------------
void resize_nearest(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    std::vector<int> col(dw);
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    for (int x = 0; x < dw; ++x) col[x] = 3 * std::min((int)std::floor(x * fx), sw - 1);
    for (int y = 0; y < dh; ++y) {
        const uint8_t* row = src + (size_t)std::min((int)std::floor(y * fy), sh - 1) * sw * 3;
        uint8_t* out = dst + (size_t)y * dw * 3;
        for (int x = 0; x < dw; ++x) {
            const uint8_t* p = row + col[x];
            out[3 * x] = p[0];
            out[3 * x + 1] = p[1];
            out[3 * x + 2] = p[2];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// OpenCV INTER_LINEAR for uint8: 11-bit fixed-point weights, horizontal pass into int32, vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16), + 2, >> 2; an exact 2x2 reduction takes the INTER_AREA fast path
// ---------------------------------------------------------------------------------------------------------------------
struct LinearTaps {
    std::vector<int> lo, hi, wlo, whi;
    LinearTaps(int src, int dst) : lo(dst), hi(dst), wlo(dst), whi(dst) {
        const double scale = 1.0 / ((double)dst / (double)src);
        for (int i = 0; i < dst; ++i) {
            float f = (float)(((double)i + 0.5) * scale - 0.5);
            int s = (int)std::floor(f);
            f = f - (float)s;
            if (s < 0) { s = 0; f = 0.f; }
            if (s >= src - 1) { s = src - 1; f = 0.f; }
            lo[i] = s;
            hi[i] = std::min(s + 1, src - 1);
            whi[i] = (int)std::nearbyintf(f * 2048.f);
            wlo[i] = (int)std::nearbyintf((1.f - f) * 2048.f);
        }
    }
};

Y3F_CLONES void resize_linear(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    if (sh == dh && sw == dw) {
        memcpy(dst, src, (size_t)sh * sw * 3);
        return;
    }
    if (sw == 2 * dw && sh == 2 * dh) {
        for (int y = 0; y < dh; ++y) {
            const uint8_t* r0 = src + (size_t)(2 * y) * sw * 3;
            const uint8_t* r1 = r0 + (size_t)sw * 3;
            uint8_t* out = dst + (size_t)y * dw * 3;
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < 3; ++c)
                    out[3 * x + c] = (uint8_t)((r0[6 * x + c] + r0[6 * x + 3 + c] + r1[6 * x + c] + r1[6 * x + 3 + c] + 2) >> 2);
        }
        return;
    }
    const LinearTaps tx(sw, dw), ty(sh, dh);
    // the horizontal pass of a source row is kept while consecutive output rows use it (two rows live at a time)
    std::vector<int32_t> buf[2] = {std::vector<int32_t>((size_t)dw * 3), std::vector<int32_t>((size_t)dw * 3)};
    int held[2] = {-1, -1};
    auto row_of = [&](int sy, int avoid) -> const int32_t* {
        for (int k = 0; k < 2; ++k)
            if (held[k] == sy) return buf[k].data();
        const int k = (held[0] == avoid) ? 1 : 0;
        const uint8_t* r = src + (size_t)sy * sw * 3;
        int32_t* o = buf[k].data();
        for (int x = 0; x < dw; ++x) {
            const uint8_t* a = r + 3 * tx.lo[x];
            const uint8_t* b = r + 3 * tx.hi[x];
            const int wa = tx.wlo[x], wb = tx.whi[x];
            o[3 * x] = a[0] * wa + b[0] * wb;
            o[3 * x + 1] = a[1] * wa + b[1] * wb;
            o[3 * x + 2] = a[2] * wa + b[2] * wb;
        }
        held[k] = sy;
        return o;
    };
    for (int y = 0; y < dh; ++y) {
        const int32_t* s0 = row_of(ty.lo[y], ty.hi[y]);
        const int32_t* s1 = row_of(ty.hi[y], ty.lo[y]);
        const int b0 = ty.wlo[y], b1 = ty.whi[y];
        uint8_t* out = dst + (size_t)y * dw * 3;
        for (int i = 0; i < dw * 3; ++i)
            out[i] = clamp_u8((((b0 * (s0[i] >> 4)) >> 16) + ((b1 * (s1[i] >> 4)) >> 16) + 2) >> 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pillow's 8-bit resampling: per output index a window [first, first + count) of source indices with normalised filter
// weights in 22-bit fixed point; horizontal pass (rounded to uint8) over the rows the vertical pass needs, then vertical
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCoefBits = 32 - 8 - 2;

double filter_box(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }

double filter_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

double sinc(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return std::sin(x) / x;
}

double filter_lanczos(double x) { return (-3.0 <= x && x < 3.0) ? sinc(x) * sinc(x / 3) : 0.0; }

struct Kernel1D {
    int ksize = 0;
    std::vector<int> first, count;
    std::vector<int32_t> coef;      // [out][ksize]
    Kernel1D(int in_size, int out_size, double (*filter)(double), double filter_support) : first(out_size), count(out_size) {
        const double scale = (double)((float)in_size - 0.f) / out_size;
        const double filterscale = scale < 1.0 ? 1.0 : scale;
        const double support = filter_support * filterscale;
        ksize = (int)std::ceil(support) * 2 + 1;
        coef.assign((size_t)out_size * ksize, 0);
        std::vector<double> w(ksize);
        const double ss = 1.0 / filterscale;
        for (int o = 0; o < out_size; ++o) {
            const double center = 0.f + (o + 0.5) * scale;
            int lo = (int)(center - support + 0.5);
            if (lo < 0) lo = 0;
            int hi = (int)(center + support + 0.5);
            if (hi > in_size) hi = in_size;
            const int n = hi - lo;
            double total = 0.0;
            for (int x = 0; x < n; ++x) {
                w[x] = filter((x + lo - center + 0.5) * ss);
                total += w[x];
            }
            int32_t* k = &coef[(size_t)o * ksize];
            for (int x = 0; x < n; ++x) {
                double v = w[x];
                if (total != 0.0) v /= total;
                k[x] = v < 0 ? (int32_t)(-0.5 + v * (1 << kCoefBits)) : (int32_t)(0.5 + v * (1 << kCoefBits));
            }
            first[o] = lo;
            count[o] = n;
        }
    }
};

// `live`: the caller may know that everything outside a rectangle of the source is zero (the black canvas around an
// expanded image): zero pixels add nothing to the integer sums, so their rows and taps are skipped - same bytes, less work.
struct Rect {
    int x0, y0, x1, y1;
};

Y3F_CLONES void resample(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, double (*filter)(double), double support,
                         Rect live) {
    const bool horizontal = dw != sw, vertical = dh != sh;
    if (!horizontal && !vertical) {
        memcpy(dst, src, (size_t)sh * sw * 3);
        return;
    }
    const Kernel1D kx(sw, dw, filter, support), ky(sh, dh, filter, support);
    const int row_first = ky.first[0], row_last = ky.first[dh - 1] + ky.count[dh - 1];
    const uint8_t* mid = src;          // what the vertical pass reads: rows [row_base, ...), width mid_w
    int row_base = 0;
    std::vector<uint8_t> tmp;
    if (horizontal) {
        const int rows = row_last - row_first;
        uint8_t* out_base = dst;
        if (vertical) {
            tmp.resize((size_t)rows * dw * 3);
            out_base = tmp.data();
        }
        for (int y = 0; y < rows; ++y) {
            const uint8_t* in = src + (size_t)(y + row_first) * sw * 3;
            uint8_t* out = out_base + (size_t)y * dw * 3;
            if (y + row_first < live.y0 || y + row_first >= live.y1) {
                memset(out, 0, (size_t)dw * 3);
                continue;
            }
            for (int x = 0; x < dw; ++x) {
                const int lo = std::max(kx.first[x], live.x0), hi = std::max(lo, std::min(kx.first[x] + kx.count[x], live.x1));
                const int32_t* k = &kx.coef[(size_t)x * kx.ksize];
                const uint8_t* p = in + 3 * lo;
                int32_t s0 = 1 << (kCoefBits - 1), s1 = s0, s2 = s0;
                for (int t = lo - kx.first[x], n = hi - kx.first[x]; t < n; ++t, p += 3) {
                    s0 += p[0] * k[t];
                    s1 += p[1] * k[t];
                    s2 += p[2] * k[t];
                }
                out[3 * x] = clamp_u8(s0 >> kCoefBits);
                out[3 * x + 1] = clamp_u8(s1 >> kCoefBits);
                out[3 * x + 2] = clamp_u8(s2 >> kCoefBits);
            }
        }
        if (!vertical) return;      // (row_first = 0, rows = sh = dh then)
        mid = tmp.data();
        row_base = row_first;
    }
    const size_t stride = (size_t)dw * 3;
    std::vector<int32_t> acc(stride);
    for (int y = 0; y < dh; ++y) {
        const int32_t* k = &ky.coef[(size_t)y * ky.ksize];
        std::fill(acc.begin(), acc.end(), 1 << (kCoefBits - 1));
        const int lo = std::max(ky.first[y], live.y0), hi = std::min(ky.first[y] + ky.count[y], live.y1);
        for (int r = lo; r < hi; ++r) {
            const uint8_t* in = mid + (size_t)(r - row_base) * stride;
            const int32_t kt = k[r - ky.first[y]];
            for (size_t i = 0; i < stride; ++i) acc[i] += in[i] * kt;
        }
        uint8_t* out = dst + (size_t)y * stride;
        for (size_t i = 0; i < stride; ++i) out[i] = clamp_u8(acc[i] >> kCoefBits);
    }
}

int resize_any(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, int interp, const Rect* known = nullptr) {
    if (!src || !dst || sh < 1 || sw < 1 || dh < 1 || dw < 1)
        return fail(Y3F_EINVAL, "resize: empty image or null pointer (%dx%d -> %dx%d)", sw, sh, dw, dh);
    Rect live = known ? *known : Rect{0, 0, sw, sh};
    live = {std::min(std::max(live.x0, 0), sw), std::min(std::max(live.y0, 0), sh), std::min(std::max(live.x1, 0), sw),
            std::min(std::max(live.y1, 0), sh)};
    if (live.x1 <= live.x0 || live.y1 <= live.y0) live = {0, 0, 0, 0};          // nothing but black
    try {
        switch (interp) {
            case Y3F_INTER_NEAREST: resize_nearest(src, sh, sw, dst, dh, dw); return Y3F_OK;
            case Y3F_INTER_LINEAR: resize_linear(src, sh, sw, dst, dh, dw); return Y3F_OK;
            case Y3F_INTER_CUBIC: resample(src, sh, sw, dst, dh, dw, filter_bicubic, 2.0, live); return Y3F_OK;
            case Y3F_INTER_AREA: resample(src, sh, sw, dst, dh, dw, filter_box, 0.5, live); return Y3F_OK;
            case Y3F_INTER_LANCZOS4: resample(src, sh, sw, dst, dh, dw, filter_lanczos, 3.0, live); return Y3F_OK;
            default: return fail(Y3F_EINVAL, "resize: interpolation code %d is not one of 0..4", interp);
        }
    } catch (const std::bad_alloc&) {
        return fail(Y3F_ENOMEM, "resize: out of memory");
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pillow's RGB <-> HSV for 8-bit pixels (H, S, V all on 0..255)
// ---------------------------------------------------------------------------------------------------------------------
inline void rgb_to_hsv_px(int r, int g, int b, uint8_t* out) {
    const int maxc = std::max(r, std::max(g, b)), minc = std::min(r, std::min(g, b));
    out[2] = (uint8_t)maxc;
    if (minc == maxc) {
        out[0] = out[1] = 0;
        return;
    }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + rc - bc);
    else h = (float)(4.0 + gc - rc);
    h = (float)std::fmod(h / 6.0 + 1.0, 1.0);
    out[0] = clamp_u8((int)(h * 255.0));
    out[1] = clamp_u8((int)(s * 255.0));
}

inline void hsv_to_rgb_px(int h, int s, int v, uint8_t* out) {
    if (s == 0) {
        out[0] = out[1] = out[2] = (uint8_t)v;
        return;
    }
    const int i = (int)std::floor((float)h * 6.0 / 255.0);
    const float f = (float)((float)h * 6.0 / 255.0 - (float)i);
    const float fs = (float)(((float)s) / 255.0);
    const uint8_t p = clamp_u8((int)std::round((float)v * (1.0 - fs)));
    const uint8_t q = clamp_u8((int)std::round((float)v * (1.0 - fs * f)));
    const uint8_t t = clamp_u8((int)std::round((float)v * (1.0 - fs * (1.0 - f))));
    const uint8_t u = (uint8_t)v;
    switch (i % 6) {
        case 0: out[0] = u; out[1] = t; out[2] = p; break;
        case 1: out[0] = q; out[1] = u; out[2] = p; break;
        case 2: out[0] = p; out[1] = u; out[2] = t; break;
        case 3: out[0] = p; out[1] = q; out[2] = u; break;
        case 4: out[0] = t; out[1] = p; out[2] = u; break;
        default: out[0] = u; out[1] = p; out[2] = q; break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two conversions behind tables, for the per-pixel loop.  Pillow's RGB -> HSV only looks at differences: with
// d(x) = max - x, the hue is a function of which channel is the maximum and of the other two channels' d (the chroma is
// the larger of those), and the saturation a function of (max, min); HSV -> RGB needs the sector and the fraction of the
// hue, s / 255, and one or two products.  The tables are filled BY the functions above, and the fast forms are compared
// with them over all 2^24 inputs (tests/test_feed_native.py through y3f_colour_distort with an identity jitter).
// ---------------------------------------------------------------------------------------------------------------------
struct ColourTables {
    uint8_t hue[3][256][256];      // [channel that is the maximum][d of the next channel][d of the one after]
    uint8_t sat[256][256];         // [max][min]
    uint8_t sector[256];           // floor(h * 6 / 255) % 6
    float frac[256];               // h * 6 / 255 - floor(.)
    float unit[256];               // s / 255
    ColourTables() {
        uint8_t hsv[3];
        for (int d1 = 0; d1 < 256; ++d1)
            for (int d2 = 0; d2 < 256; ++d2) {
                rgb_to_hsv_px(255, 255 - d1, 255 - d2, hsv);                        // red is the maximum: d(g), d(b)
                hue[0][d1][d2] = hsv[0];
                rgb_to_hsv_px(d1 ? 255 - d1 : 254, 255, 255 - d2, hsv);             // green (red below it): d(r), d(b)
                hue[1][d1][d2] = hsv[0];
                rgb_to_hsv_px(d1 ? 255 - d1 : 254, d2 ? 255 - d2 : 254, 255, hsv);  // blue (red, green below it): d(r), d(g)
                hue[2][d1][d2] = hsv[0];
            }
        for (int mx = 0; mx < 256; ++mx)
            for (int mn = 0; mn < 256; ++mn) {
                if (mn > mx) {
                    sat[mx][mn] = 0;
                    continue;
                }
                rgb_to_hsv_px(mx, mn, mn, hsv);
                sat[mx][mn] = hsv[1];
            }
        for (int h = 0; h < 256; ++h) {
            const int i = (int)std::floor((float)h * 6.0 / 255.0);
            sector[h] = (uint8_t)(i % 6);
            frac[h] = (float)((float)h * 6.0 / 255.0 - (float)i);
            unit[h] = (float)(((float)h) / 255.0);
        }
    }
};

const ColourTables& colour_tables() {
    static const ColourTables tables;
    return tables;
}

// ---------------------------------------------------------------------------------------------------------------------
// random_color_distort after its draws, on a run of pixels in place.  numpy does the jitter on a float32 HSV image whose hue
// is rescaled from Pillow's 0..255 circle to OpenCV's 0..180 and back; every product / sum below is a float32 operation with
// the scalar rounded to float32 first, as numpy does for a float32 array and a Python scalar.  Each of H, S, V is jittered
// on its own, so the jitter is three 256-entry maps, built per call.
// ---------------------------------------------------------------------------------------------------------------------
inline float clip255(float v) { return v < 0.f ? 0.f : (v > 255.f ? 255.f : v); }

struct Jitter {
    uint8_t bright[256], h[256], s[256], v[256];
    explicit Jitter(const y3f_colour& c) {
        const float to_cv = (float)(180.0 / 255.0), to_pil = (float)(255.0 / 180.0);
        const float delta = (float)c.hue_delta;
        for (int x = 0; x < 256; ++x) {
            bright[x] = clamp_u8(x + c.brightness);
            float hue = (float)x * to_cv;
            if (c.hue_on) {
                hue = std::fmod(hue + delta, 180.f);          // numpy's remainder: fmod, then the divisor's sign
                if (hue != 0.f) {
                    if (hue < 0.f) hue += 180.f;
                } else {
                    hue = 0.f;
                }
            }
            h[x] = (uint8_t)(int)std::min(clip255(hue) * to_pil, 255.f);
            s[x] = (uint8_t)(int)clip255((float)x * c.sat_gain);
            v[x] = (uint8_t)(int)clip255((float)x * c.val_gain);
        }
    }
};

void colour_run(uint8_t* px, size_t n, const y3f_colour& c) {
    if (!c.enabled) return;
    const ColourTables& T = colour_tables();
    const Jitter J(c);
    for (size_t i = 0; i < n; ++i, px += 3) {
        const int r = J.bright[px[0]], g = J.bright[px[1]], b = J.bright[px[2]];
        const int mx = std::max(r, std::max(g, b)), mn = std::min(r, std::min(g, b));
        int h = 0, s = 0;
        if (mx != mn) {
            h = r == mx ? T.hue[0][mx - g][mx - b] : (g == mx ? T.hue[1][mx - r][mx - b] : T.hue[2][mx - r][mx - g]);
            s = T.sat[mx][mn];
        }
        h = J.h[h];
        s = J.s[s];
        const int v = J.v[mx];
        if (s == 0) {
            px[0] = px[1] = px[2] = (uint8_t)v;
            continue;
        }
        // Pillow: p = round(v * (1 - fs)), q = round(v * (1 - fs * f)), t = round(v * (1 - fs * (1 - f))); the products are
        // non-negative, where round(x) == (int)(x + 0.5) unless x + 0.5 rounds up across an integer - the all-inputs test
        // is what says it does not for these operands
        const float f = T.frac[h], fs = T.unit[s];
        const int sec = T.sector[h];
        const uint8_t u = (uint8_t)v;
        const uint8_t p = clamp_u8((int)((float)v * (1.0 - fs) + 0.5));
        const uint8_t qt = (sec & 1) ? clamp_u8((int)((float)v * (1.0 - fs * f) + 0.5))
                                     : clamp_u8((int)((float)v * (1.0 - fs * (1.0 - f)) + 0.5));
        switch (sec) {
            case 0: px[0] = u; px[1] = qt; px[2] = p; break;
            case 1: px[0] = qt; px[1] = u; px[2] = p; break;
            case 2: px[0] = p; px[1] = u; px[2] = qt; break;
            case 3: px[0] = p; px[1] = qt; px[2] = u; break;
            case 4: px[0] = qt; px[1] = p; px[2] = u; break;
            default: px[0] = u; px[1] = p; px[2] = qt; break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// one sample
// ---------------------------------------------------------------------------------------------------------------------
const float* unit_table() {          // v / 255 in float32 for v = 0..255
    static const struct Table {
        float t[256];
        Table() { for (int v = 0; v < 256; ++v) t[v] = (float)v / 255.f; }
    } table;
    return table.t;
}

int check_job(const y3f_job& j) {
    if (!j.img1 || j.h1 < 1 || j.w1 < 1) return fail(Y3F_EINVAL, "sample: img1 is empty");
    if (j.img2 && (j.h2 < 1 || j.w2 < 1)) return fail(Y3F_EINVAL, "sample: img2 is empty");
    if (j.win_w < 1 || j.win_h < 1 || j.res_w < 1 || j.res_h < 1 || j.out_w < 1 || j.out_h < 1)
        return fail(Y3F_EINVAL, "sample: empty window (%dx%d), resize target (%dx%d) or output (%dx%d)", j.win_w, j.win_h,
                    j.res_w, j.res_h, j.out_w, j.out_h);
    if (j.pad_x < 0 || j.pad_y < 0 || j.pad_x + j.res_w > j.out_w || j.pad_y + j.res_h > j.out_h)
        return fail(Y3F_EINVAL, "sample: %dx%d at (%d,%d) does not fit the %dx%d output", j.res_w, j.res_h, j.pad_x, j.pad_y,
                    j.out_w, j.out_h);
    const int32_t far = 1 << 28, big = 1 << 20;          // (keeps every sum of two coordinates inside int32)
    if (j.win_w > big || j.win_h > big || j.out_w > big || j.out_h > big || j.h1 > big || j.w1 > big || j.h2 > big || j.w2 > big ||
        std::abs(j.win_x) > far || std::abs(j.win_y) > far || std::abs(j.off_x) > far || std::abs(j.off_y) > far)
        return fail(Y3F_EINVAL, "sample: coordinates out of range");
    return Y3F_OK;
}

// the part of the window that is not black canvas, as run_job hands it to the resampling filters
Rect live_rect(const y3f_job& j) {
    const int mh = j.img2 ? std::max(j.h1, j.h2) : j.h1, mw = j.img2 ? std::max(j.w1, j.w2) : j.w1;
    const int x_lo = std::max(j.win_x, j.off_x), x_hi = std::min(j.win_x + j.win_w, j.off_x + mw);
    Rect live = {std::max(0, x_lo - j.win_x), std::max(0, std::max(j.win_y, j.off_y) - j.win_y), std::max(0, x_hi - j.win_x),
                 std::max(0, std::min(j.win_y + j.win_h, j.off_y + mh) - j.win_y)};
    live = {std::min(live.x0, j.win_w), std::min(live.y0, j.win_h), std::min(live.x1, j.win_w), std::min(live.y1, j.win_h)};
    if (live.x1 <= live.x0 || live.y1 <= live.y0) live = {0, 0, 0, 0};
    return live;
}

int run_job(const y3f_job& j, uint8_t* out_u8, float* out_f32) {
    const int bad = check_job(j);
    if (bad != Y3F_OK) return bad;
    if (!out_u8 && !out_f32) return Y3F_OK;
    try {
        // 1. the window of the canvas: black, except where the (mixed) image lies; those pixels are blended and jittered
        const int mh = j.img2 ? std::max(j.h1, j.h2) : j.h1, mw = j.img2 ? std::max(j.w1, j.w2) : j.w1;
        std::vector<uint8_t> win((size_t)j.win_h * j.win_w * 3, 0);
        const int x_lo = std::max(j.win_x, j.off_x), x_hi = std::min(j.win_x + j.win_w, j.off_x + mw);
        if (x_lo < x_hi) {
            for (int wy = 0; wy < j.win_h; ++wy) {
                const int iy = j.win_y + wy - j.off_y;
                if (iy < 0 || iy >= mh) continue;
                uint8_t* out = &win[((size_t)wy * j.win_w + (x_lo - j.win_x)) * 3];
                const int ix0 = x_lo - j.off_x, n = x_hi - x_lo;
                if (!j.img2) {
                    memcpy(out, j.img1 + ((size_t)iy * j.w1 + ix0) * 3, (size_t)n * 3);
                } else {
                    const uint8_t* a = iy < j.h1 ? j.img1 + (size_t)iy * j.w1 * 3 : nullptr;
                    const uint8_t* b = iy < j.h2 ? j.img2 + (size_t)iy * j.w2 * 3 : nullptr;
                    for (int x = 0; x < n; ++x) {
                        const int ix = ix0 + x;
                        for (int c = 0; c < 3; ++c) {
                            float acc = 0.f;
                            if (a && ix < j.w1) acc = (float)a[3 * ix + c] * j.lam1;
                            if (b && ix < j.w2) acc = acc + (float)b[3 * ix + c] * j.lam2;
                            out[3 * x + c] = (uint8_t)(int)acc;
                        }
                    }
                }
                colour_run(out, (size_t)n, j.colour);
            }
        }
        // 2. resize; 3. place on the padded field, mirror, convert
        const bool plain = j.res_w == j.out_w && j.res_h == j.out_h && !j.flip_x;
        std::vector<uint8_t> res_store;
        uint8_t* res = out_u8;
        if (!plain || !out_u8) {
            res_store.resize((size_t)j.res_h * j.res_w * 3);
            res = res_store.data();
        }
        // everything of the window outside the (mixed) image is the black canvas: the resampling filters skip it
        const Rect live = {std::max(0, x_lo - j.win_x), std::max(0, std::max(j.win_y, j.off_y) - j.win_y),
                           std::max(0, x_hi - j.win_x), std::max(0, std::min(j.win_y + j.win_h, j.off_y + mh) - j.win_y)};
        const int rc = resize_any(win.data(), j.win_h, j.win_w, res, j.res_h, j.res_w, j.interp, &live);
        if (rc != Y3F_OK) return rc;
        const float* unit = unit_table();
        if (plain) {
            if (out_f32)
                for (size_t i = 0, n = (size_t)j.out_h * j.out_w * 3; i < n; ++i) out_f32[i] = unit[res[i]];
            return Y3F_OK;
        }
        const uint8_t pad = clamp_u8(j.pad_value);
        std::vector<uint8_t> line((size_t)j.out_w * 3);
        for (int y = 0; y < j.out_h; ++y) {
            std::fill(line.begin(), line.end(), pad);
            if (y >= j.pad_y && y < j.pad_y + j.res_h)
                memcpy(&line[(size_t)j.pad_x * 3], res + (size_t)(y - j.pad_y) * j.res_w * 3, (size_t)j.res_w * 3);
            if (j.flip_x)
                for (int x = 0, z = j.out_w - 1; x < z; ++x, --z)
                    for (int c = 0; c < 3; ++c) std::swap(line[3 * x + c], line[3 * z + c]);
            if (out_u8) memcpy(out_u8 + (size_t)y * j.out_w * 3, line.data(), line.size());
            if (out_f32) {
                float* o = out_f32 + (size_t)y * j.out_w * 3;
                for (size_t i = 0; i < line.size(); ++i) o[i] = unit[line[i]];
            }
        }
        return Y3F_OK;
    } catch (const std::bad_alloc&) {
        return fail(Y3F_ENOMEM, "sample: out of memory");
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The trial loop of random_crop_with_constraints, on the caller's Python generator: random.Random is MT19937 and
// getstate() hands out its 624 words + position, so the loop can draw exactly what prng.uniform / prng.randrange would.
// ---------------------------------------------------------------------------------------------------------------------
struct PyRandom {
    uint32_t* mt;          // [625]: the state words, then the position
    explicit PyRandom(uint32_t* state) : mt(state) {}
    uint32_t next32() {
        constexpr int N = 624, M = 397;
        constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LOW = 0x7fffffffu;
        uint32_t& pos = mt[N];
        if (pos >= (uint32_t)N) {
            int k = 0;
            for (; k < N - M; ++k) {
                const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LOW);
                mt[k] = mt[k + M] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
            }
            for (; k < N - 1; ++k) {
                const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LOW);
                mt[k] = mt[k + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
            }
            const uint32_t y = (mt[N - 1] & UP) | (mt[0] & LOW);
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double random() {                                   // 53 bits: two words
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    double uniform(double a, double b) { return a + (b - a) * random(); }
    uint32_t below(uint32_t n) {                        // randrange(n), n >= 1: rejection on bit_length(n) bits
        int bits = 0;
        for (uint32_t v = n; v; v >>= 1) ++bits;
        uint32_t r = next32() >> (32 - bits);
        while (r >= n) r = next32() >> (32 - bits);
        return r;
    }
};

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// The device form: the same job, split into what needs the host (double-precision filter windows and weights exactly as
// Pillow computes them, the jitter maps with numpy's float32 remainder, which source pixels a window can see) and what the
// GPU does from tables (csrc/y3_feed_px.h).  plan_geometry fixes sizes and offsets, fill_job writes the blob.
// ---------------------------------------------------------------------------------------------------------------------
static_assert(sizeof(y3f_djob) == 208 && sizeof(y3f_djob) % 16 == 0, "feed_native.py and the device kernels assume this layout");

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct FilterSpec {
    double (*fn)(double);
    double support;
};

bool filter_of(int interp, FilterSpec* f) {
    switch (interp) {
        case Y3F_INTER_CUBIC: *f = {filter_bicubic, 2.0}; return true;
        case Y3F_INTER_AREA: *f = {filter_box, 0.5}; return true;
        case Y3F_INTER_LANCZOS4: *f = {filter_lanczos, 3.0}; return true;
        default: return false;
    }
}

// ksize and the source window [lo, hi) of output index o, as Kernel1D computes them
int kernel_window(int in_size, int out_size, double filter_support, int o, int* lo_out, int* hi_out) {
    const double scale = (double)((float)in_size - 0.f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filter_support * filterscale;
    const double center = 0.f + (o + 0.5) * scale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
    *lo_out = lo;
    *hi_out = hi;
    return (int)std::ceil(support) * 2 + 1;
}

int plan_geometry(const y3f_job& j, y3f_djob& d, size_t& blob, size_t& scratch) {
    const int bad = check_job(j);
    if (bad != Y3F_OK) return bad;
    memset(&d, 0, sizeof(d));
    const int sw = j.win_w, sh = j.win_h, dw = j.res_w, dh = j.res_h;
    const Rect live = live_rect(j);
    d.live_x0 = live.x0, d.live_y0 = live.y0, d.live_x1 = live.x1, d.live_y1 = live.y1;
    d.win_w = sw, d.win_h = sh;
    d.img_dx = j.win_x - j.off_x, d.img_dy = j.win_y - j.off_y;
    d.lam1 = j.lam1, d.lam2 = j.lam2;
    d.has2 = j.img2 != nullptr;
    d.colour_on = j.colour.enabled != 0;
    d.res_w = dw, d.res_h = dh, d.out_w = j.out_w, d.out_h = j.out_h, d.pad_x = j.pad_x, d.pad_y = j.pad_y;
    d.pad_value = clamp_u8(j.pad_value), d.flip_x = j.flip_x != 0;
    // the source pixels the live part of the window can see
    const int ix0 = live.x0 + d.img_dx, ix1 = live.x1 + d.img_dx, iy0 = live.y0 + d.img_dy, iy1 = live.y1 + d.img_dy;
    auto cut = [&](int w, int h, int32_t* r) {
        const int x0 = std::max(ix0, 0), x1 = std::min(ix1, w), y0 = std::max(iy0, 0), y1 = std::min(iy1, h);
        if (x1 <= x0 || y1 <= y0) { r[0] = r[1] = r[2] = r[3] = 0; return; }
        r[0] = x0, r[1] = y0, r[2] = x1 - x0, r[3] = y1 - y0;
    };
    cut(j.w1, j.h1, &d.r1_x0);
    if (d.has2) cut(j.w2, j.h2, &d.r2_x0);
    FilterSpec f;
    switch (j.interp) {
        case Y3F_INTER_NEAREST: d.mode = Y3F_MODE_NEAREST; break;
        case Y3F_INTER_LINEAR:
            d.mode = (sh == dh && sw == dw) ? Y3F_MODE_COPY : (sw == 2 * dw && sh == 2 * dh) ? Y3F_MODE_MEAN2X2 : Y3F_MODE_LINEAR;
            break;
        case Y3F_INTER_CUBIC: case Y3F_INTER_AREA: case Y3F_INTER_LANCZOS4:
            d.horizontal = dw != sw, d.vertical = dh != sh;
            d.mode = (d.horizontal || d.vertical) ? Y3F_MODE_RESAMPLE : Y3F_MODE_COPY;
            break;
        default: return fail(Y3F_EINVAL, "resize: interpolation code %d is not one of 0..4", j.interp);
    }
    size_t xtab = 0, ytab = 0;
    if (d.mode == Y3F_MODE_NEAREST) {
        xtab = (size_t)dw * 4, ytab = (size_t)dh * 4;
    } else if (d.mode == Y3F_MODE_LINEAR) {
        xtab = (size_t)dw * 16, ytab = (size_t)dh * 16;
    } else if (d.mode == Y3F_MODE_RESAMPLE) {
        filter_of(j.interp, &f);
        int lo, hi, row_first = 0, row_last = sh;
        if (d.horizontal) {
            d.ksize_x = kernel_window(sw, dw, f.support, 0, &lo, &hi);
            xtab = (size_t)dw * (2 + d.ksize_x) * 4;
        }
        if (d.vertical) {
            d.ksize_y = kernel_window(sh, dh, f.support, 0, &row_first, &hi);
            kernel_window(sh, dh, f.support, dh - 1, &lo, &row_last);
            ytab = (size_t)dh * (2 + d.ksize_y) * 4;
        }
        if (d.horizontal) {
            d.tmp_y0 = std::max(row_first, live.y0);
            d.tmp_rows = std::max(0, std::min(row_last, live.y1) - d.tmp_y0);
        }
    }
    d.img1_off = blob, blob += align16((size_t)d.r1_w * d.r1_h * 3);
    d.img2_off = blob, blob += align16((size_t)d.r2_w * d.r2_h * 3);
    d.jitter_off = blob, blob += d.colour_on ? 1024 : 0;
    d.xtab_off = blob, blob += align16(xtab);
    d.ytab_off = blob, blob += align16(ytab);
    d.win_off = scratch, scratch += align16((size_t)(live.x1 - live.x0) * (live.y1 - live.y0) * 3);
    d.tmp_off = scratch, scratch += align16((size_t)d.tmp_rows * dw * 3);
    return Y3F_OK;
}

void fill_job(const y3f_job& j, const y3f_djob& d, uint8_t* blob) {
    auto pack = [&](const uint8_t* img, int w, const int32_t* r, uint64_t off) {
        for (int y = 0; y < r[3]; ++y)
            memcpy(blob + off + (size_t)y * r[2] * 3, img + ((size_t)(r[1] + y) * w + r[0]) * 3, (size_t)r[2] * 3);
    };
    pack(j.img1, j.w1, &d.r1_x0, d.img1_off);
    if (d.has2) pack(j.img2, j.w2, &d.r2_x0, d.img2_off);
    if (d.colour_on) {
        const Jitter J(j.colour);
        uint8_t* t = blob + d.jitter_off;
        memcpy(t, J.bright, 256), memcpy(t + 256, J.h, 256), memcpy(t + 512, J.s, 256), memcpy(t + 768, J.v, 256);
    }
    const int sw = j.win_w, sh = j.win_h, dw = j.res_w, dh = j.res_h;
    int32_t* xt = reinterpret_cast<int32_t*>(blob + d.xtab_off);
    int32_t* yt = reinterpret_cast<int32_t*>(blob + d.ytab_off);
    if (d.mode == Y3F_MODE_NEAREST) {
        const double fx = (double)sw / dw, fy = (double)sh / dh;
        for (int x = 0; x < dw; ++x) xt[x] = std::min((int)std::floor(x * fx), sw - 1);
        for (int y = 0; y < dh; ++y) yt[y] = std::min((int)std::floor(y * fy), sh - 1);
    } else if (d.mode == Y3F_MODE_LINEAR) {
        auto taps = [](const LinearTaps& t, int n, int32_t* o) {
            for (int i = 0; i < n; ++i) o[i] = t.lo[i], o[n + i] = t.hi[i], o[2 * n + i] = t.wlo[i], o[3 * n + i] = t.whi[i];
        };
        taps(LinearTaps(sw, dw), dw, xt);
        taps(LinearTaps(sh, dh), dh, yt);
    } else if (d.mode == Y3F_MODE_RESAMPLE) {
        FilterSpec f;
        filter_of(j.interp, &f);
        auto table = [&](int in, int out, int32_t* o) {
            const Kernel1D k(in, out, f.fn, f.support);
            for (int i = 0; i < out; ++i) o[i] = k.first[i], o[out + i] = k.count[i];
            memcpy(o + 2 * (size_t)out, k.coef.data(), k.coef.size() * 4);
        };
        if (d.horizontal) table(sw, dw, xt);
        if (d.vertical) table(sh, dh, yt);
    }
}

// run `work(i)` for i in [0, n) on up to `threads` threads (the caller's included)
template <typename F>
void parallel_jobs(int n, int threads, F work) {
    int workers = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    workers = std::max(1, std::min(workers, n));
    std::atomic<int> next(0);
    auto loop = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) work(i);
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < workers; ++t) pool.emplace_back(loop);
    } catch (...) {      // fewer threads than asked for: the caller's thread takes what is left
    }
    loop();
    for (auto& t : pool) t.join();
}

}  // namespace

extern "C" {

int y3f_plan_batch(const y3f_job* jobs, int n, uint8_t* blob, size_t capacity, size_t* blob_bytes, size_t* scratch_bytes,
                   int threads) {
    if (n < 0 || (n && !jobs) || !blob_bytes || !scratch_bytes) return fail(Y3F_EINVAL, "plan_batch: bad arguments");
    try {
        std::vector<y3f_djob> recs((size_t)n);
        size_t need = align16((size_t)n * sizeof(y3f_djob)), scratch = 0;
        for (int i = 0; i < n; ++i) {
            if (jobs[i].out_w != jobs[0].out_w || jobs[i].out_h != jobs[0].out_h)
                return fail(Y3F_EINVAL, "plan_batch: job %d writes %dx%d, job 0 %dx%d", i, jobs[i].out_w, jobs[i].out_h,
                            jobs[0].out_w, jobs[0].out_h);
            const int rc = plan_geometry(jobs[i], recs[i], need, scratch);
            if (rc != Y3F_OK) {
                char message[sizeof(g_error)];
                snprintf(message, sizeof(message), "job %d: %s", i, g_error);
                snprintf(g_error, sizeof(g_error), "%s", message);
                return rc;
            }
        }
        *blob_bytes = need;
        *scratch_bytes = scratch;
        if (!blob || capacity < need) return Y3F_OK;
        if (n) memcpy(blob, recs.data(), (size_t)n * sizeof(y3f_djob));
        parallel_jobs(n, threads, [&](int i) { fill_job(jobs[i], recs[i], blob); });
        return Y3F_OK;
    } catch (const std::bad_alloc&) {
        return fail(Y3F_ENOMEM, "plan_batch: out of memory");
    }
}

size_t y3f_device_tables(void* dst, size_t capacity) {
    static_assert(sizeof(y3f_dtables) == 3 * 65536 + 65536 + 256 + 3 * 1024, "no padding expected");
    if (dst && capacity >= sizeof(y3f_dtables)) {
        y3f_dtables* t = static_cast<y3f_dtables*>(dst);
        const ColourTables& T = colour_tables();
        memcpy(t->hue, T.hue, sizeof(t->hue));
        memcpy(t->sat, T.sat, sizeof(t->sat));
        memcpy(t->sector, T.sector, sizeof(t->sector));
        memcpy(t->frac, T.frac, sizeof(t->frac));
        memcpy(t->unit, T.unit, sizeof(t->unit));
        memcpy(t->unit255, unit_table(), sizeof(t->unit255));
    }
    return sizeof(y3f_dtables);
}

int y3f_crop_candidates(uint32_t* mt_state, const double* boxes, int n_boxes, int width, int height, double min_scale,
                        double max_scale, double max_aspect_ratio, const double* bands, int n_bands, int max_trial,
                        int32_t* windows, int32_t* n_windows) {
    if (!mt_state || !windows || !n_windows || (n_boxes && !boxes) || (n_bands && !bands) || n_boxes < 0 || n_bands < 0)
        return fail(Y3F_EINVAL, "crop_candidates: null pointer or negative count");
    if (mt_state[624] > 624) return fail(Y3F_EINVAL, "crop_candidates: generator position %u is out of range", mt_state[624]);
    PyRandom prng(mt_state);
    int found = 0;
    for (int band = 0; band < n_bands; ++band) {
        const double floor_iou = bands[2 * band], ceil_iou = bands[2 * band + 1];
        for (int trial = 0; trial < max_trial; ++trial) {
            const double s = prng.uniform(min_scale, max_scale);
            const double root = std::sqrt(prng.uniform(std::max(1 / max_aspect_ratio, s * s), std::min(max_aspect_ratio, 1 / (s * s))));
            const int win_h = (int)(height * s / root), win_w = (int)(width * s * root);
            if (height - win_h < 1 || width - win_w < 1) continue;
            const int y = (int)prng.below((uint32_t)(height - win_h));
            const int x = (int)prng.below((uint32_t)(width - win_w));
            int32_t* out = windows + 4 * found;
            out[0] = x, out[1] = y, out[2] = win_w, out[3] = win_h;
            if (n_boxes == 0) {                         // nothing to constrain: the first proper window is the crop
                *n_windows = -1;
                windows[0] = x, windows[1] = y, windows[2] = win_w, windows[3] = win_h;
                return Y3F_OK;
            }
            double least = INFINITY, most = -INFINITY;
            const double cl = x, ct = y, cr = x + win_w, cb = y + win_h;
            const double area_b = (double)(win_w * win_h);
            for (int i = 0; i < n_boxes; ++i) {
                const double x0 = boxes[4 * i], y0 = boxes[4 * i + 1], x1 = boxes[4 * i + 2], y1 = boxes[4 * i + 3];
                const double tlx = std::max(x0, cl), tly = std::max(y0, ct), brx = std::min(x1, cr), bry = std::min(y1, cb);
                const double area_i = (brx - tlx) * (bry - tly) * ((tlx < brx && tly < bry) ? 1.0 : 0.0);
                const double v = area_i / ((x1 - x0) * (y1 - y0) + area_b - area_i);
                least = std::min(least, v);
                most = std::max(most, v);
            }
            if (floor_iou <= least && most <= ceil_iou) {
                ++found;
                break;
            }
        }
    }
    *n_windows = found;
    return Y3F_OK;
}

const char* y3f_last_error(void) { return g_error; }

int y3f_abi_version(void) { return 2; }

int y3f_resize(const uint8_t* src, int src_h, int src_w, uint8_t* dst, int dst_h, int dst_w, int interp) {
    return resize_any(src, src_h, src_w, dst, dst_h, dst_w, interp);
}

int y3f_rgb_to_hsv(const uint8_t* rgb, uint8_t* hsv, size_t pixels) {
    if (pixels && (!rgb || !hsv)) return fail(Y3F_EINVAL, "rgb_to_hsv: null pointer");
    for (size_t i = 0; i < pixels; ++i) rgb_to_hsv_px(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], hsv + 3 * i);
    return Y3F_OK;
}

int y3f_hsv_to_rgb(const uint8_t* hsv, uint8_t* rgb, size_t pixels) {
    if (pixels && (!rgb || !hsv)) return fail(Y3F_EINVAL, "hsv_to_rgb: null pointer");
    for (size_t i = 0; i < pixels; ++i) hsv_to_rgb_px(hsv[3 * i], hsv[3 * i + 1], hsv[3 * i + 2], rgb + 3 * i);
    return Y3F_OK;
}

int y3f_colour_distort(uint8_t* rgb, size_t pixels, const y3f_colour* colour) {
    if (!colour || (pixels && !rgb)) return fail(Y3F_EINVAL, "colour_distort: null pointer");
    colour_run(rgb, pixels, *colour);
    return Y3F_OK;
}

int y3f_sample(const y3f_job* job, uint8_t* out_u8, float* out_f32) {
    if (!job) return fail(Y3F_EINVAL, "sample: null job");
    return run_job(*job, out_u8, out_f32);
}

int y3f_sample_batch(const y3f_job* jobs, int n, uint8_t* const* outs_u8, float* const* outs_f32, int threads) {
    if (n < 0 || (n && !jobs)) return fail(Y3F_EINVAL, "sample_batch: bad job array");
    if (n == 0) return Y3F_OK;
    int workers = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    workers = std::max(1, std::min(workers, n));
    std::atomic<int> next(0), first_error(Y3F_OK);
    char message[sizeof(g_error)] = "";
    std::atomic<bool> have_message(false);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            const int rc = run_job(jobs[i], outs_u8 ? outs_u8[i] : nullptr, outs_f32 ? outs_f32[i] : nullptr);
            int expected = Y3F_OK;
            if (rc != Y3F_OK && first_error.compare_exchange_strong(expected, rc)) {
                snprintf(message, sizeof(message), "job %d: %s", i, g_error);
                have_message.store(true);
            }
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < workers; ++t) pool.emplace_back(work);
    } catch (...) {      // fewer threads than asked for: the caller's thread takes what is left
    }
    work();
    for (auto& t : pool) t.join();
    if (have_message.load()) snprintf(g_error, sizeof(g_error), "%s", message);
    return first_error.load();
}

}  // extern "C"
