// Parameter preparation (weight re-pack, BN fold) and the anchor decode for gfx950.
//
// y3_decode replaces the ~60 small TensorFlow ops of yolov3.reorg_layer (model.py:82-137),
// yolov3.predict (model.py:140-190) and the conf*prob product (test_single_image.py:55) with ONE
// HBM-bound elementwise pass over the three feature maps: each feature map is a flat array of
// (N*g*g*3) boxes x (5+C) fields, read once with coalesced 4-byte-per-lane loads (the (5+C)=85
// record length defeats wider vectors) and written once.
#include "y3_internal.h"

namespace {

__global__ void pack_weights_kernel(const float* __restrict__ w_hwio, float* __restrict__ w_packed,
                                    int taps, int cin, int cout) {
    // out[t][co][ci] = in[t][ci][co]; one thread per output element, reads go through L2.
    const size_t total = (size_t)taps * cin * cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const size_t r = i / cin;
        const int co = (int)(r % cout);
        const int t = (int)(r / cout);
        w_packed[i] = w_hwio[((size_t)t * cin + ci) * cout + co];
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, int c, float* scale, float* shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) {
        const float s = gamma[i] / sqrtf(var[i] + eps);
        scale[i] = s;
        shift[i] = beta[i] - mean[i] * s;
    }
}

struct DecodeArgs {
    const float* fm[3];
    int g_h[3], g_w[3];
    int box_off[3];        // first box index of each scale inside one image
    long long elem_end[3]; // cumulative element counts over the three scales (all images)
    float ratio_h[3], ratio_w[3];
    float ra_w[3][3], ra_h[3][3];  // rescaled anchors: anchor / ratio  (model.py:94)
    int n, C, F, B;        // F = 5 + C, B = boxes per image
    float* boxes; float* confs; float* probs; float* scores;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256) decode_kernel(const DecodeArgs a) {
    const long long total = a.elem_end[2];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        int s = 0;
        long long le = e;
        if (e >= a.elem_end[1]) { s = 2; le = e - a.elem_end[1]; }
        else if (e >= a.elem_end[0]) { s = 1; le = e - a.elem_end[0]; }
        const float* fm = a.fm[s];
        const int f = (int)(le % a.F);
        const long long gb = le / a.F;                 // box index over (n, y, x, anchor)
        const int per_img = a.g_h[s] * a.g_w[s] * 3;
        const int n = (int)(gb / per_img);
        const int lb = (int)(gb - (long long)n * per_img);
        const size_t ob = (size_t)n * a.B + a.box_off[s] + lb;   // output box slot
        if (f >= 5) {
            const float pr = sigmoidf_(fm[le]);
            a.probs[ob * a.C + (f - 5)] = pr;
            if (a.scores) a.scores[ob * a.C + (f - 5)] = sigmoidf_(fm[le - f + 4]) * pr;
        } else if (f == 4) {
            a.confs[ob] = sigmoidf_(fm[le]);
        } else if (f == 0) {
            const int anc = lb % 3;
            const int cell = lb / 3;
            const int gy = cell / a.g_w[s], gx = cell - gy * a.g_w[s];
            const float tx = fm[le], ty = fm[le + 1], tw = fm[le + 2], th = fm[le + 3];
            // model.py:105-126: (sigmoid + offset) * ratio ; (exp * rescaled_anchor) * ratio
            const float cx = (sigmoidf_(tx) + (float)gx) * a.ratio_w[s];
            const float cy = (sigmoidf_(ty) + (float)gy) * a.ratio_h[s];
            const float bw = (expf(tw) * a.ra_w[s][anc]) * a.ratio_w[s];
            const float bh = (expf(th) * a.ra_h[s][anc]) * a.ratio_h[s];
            // model.py:182-188
            f32x4 o;
            o[0] = cx - bw / 2.f;
            o[1] = cy - bh / 2.f;
            o[2] = cx + bw / 2.f;
            o[3] = cy + bh / 2.f;
            *reinterpret_cast<f32x4*>(a.boxes + ob * 4) = o;
        }
    }
}

}  // namespace

extern "C" int y3_pack_conv_weights(y3_ctx* ctx, const float* w_hwio, int k, int cin, int cout,
                                    float* w_packed) {
    Y3_CHECK_ARG(ctx && w_hwio && w_packed, "y3_pack_conv_weights: null argument");
    Y3_CHECK_ARG(k > 0 && cin > 0 && cout > 0, "y3_pack_conv_weights: non-positive dimension");
    const size_t total = (size_t)k * k * cin * cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, ctx->stream, w_hwio, w_packed,
                       k * k, cin, cout);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_bn_fold(y3_ctx* ctx, const float* gamma, const float* beta, const float* mean,
                          const float* var, float eps, int c, float* scale, float* shift) {
    Y3_CHECK_ARG(ctx && gamma && beta && mean && var && scale && shift, "y3_bn_fold: null argument");
    Y3_CHECK_ARG(c > 0, "y3_bn_fold: non-positive channel count");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, ctx->stream, gamma, beta,
                       mean, var, eps, c, scale, shift);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_decode(y3_ctx* ctx, const float* fm1, const float* fm2, const float* fm3, int n, int h,
                         int w, int class_num, const float* anchors, float* boxes, float* confs,
                         float* probs, float* scores) {
    Y3_CHECK_ARG(ctx && fm1 && fm2 && fm3 && anchors && boxes && confs && probs,
                 "y3_decode: null argument");
    Y3_CHECK_ARG(n > 0 && class_num > 0, "y3_decode: non-positive dimension");
    Y3_CHECK_ARG(h > 0 && w > 0 && h % 32 == 0 && w % 32 == 0,
                 "y3_decode: input size must be a positive multiple of 32 (got %dx%d)", h, w);
    DecodeArgs a;
    a.fm[0] = fm1; a.fm[1] = fm2; a.fm[2] = fm3;
    a.n = n; a.C = class_num; a.F = 5 + class_num;
    const int strides[3] = {32, 16, 8};
    int off = 0;
    long long cum = 0;
    for (int s = 0; s < 3; ++s) {
        a.g_h[s] = h / strides[s];
        a.g_w[s] = w / strides[s];
        // model.py:91: ratio = cast(img_size / grid_size, float32)  (true division, then cast)
        a.ratio_h[s] = (float)((double)h / (double)a.g_h[s]);
        a.ratio_w[s] = (float)((double)w / (double)a.g_w[s]);
        for (int k = 0; k < 3; ++k) {
            const int ai = (2 - s) * 3 + k;  // model.py:147-149: scale 0 -> anchors[6:9]
            // model.py:94: python float anchor / float32 ratio -> float32 division
            a.ra_w[s][k] = anchors[2 * ai] / a.ratio_w[s];
            a.ra_h[s][k] = anchors[2 * ai + 1] / a.ratio_h[s];
        }
        a.box_off[s] = off;
        off += a.g_h[s] * a.g_w[s] * 3;
        cum += (long long)n * a.g_h[s] * a.g_w[s] * 3 * a.F;
        a.elem_end[s] = cum;
    }
    a.B = off;
    a.boxes = boxes; a.confs = confs; a.probs = probs; a.scores = scores;
    long long nb = (cum + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(decode_kernel, dim3((int)nb), dim3(256), 0, ctx->stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
