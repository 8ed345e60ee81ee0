// Parameter preparation (weight re-pack, BN fold) and the anchor decode for gfx950.
//
// y3_decode replaces the ~60 small TensorFlow ops of yolov3.reorg_layer (model.py:82-137),
// yolov3.predict (model.py:140-190) and the conf*prob product (test_single_image.py:55) with ONE
// HBM-bound pass over the three feature maps: each feature map is a flat array of (N*g*g*3) records x (5+C)
// fields.  A workgroup stages 64 consecutive records of one (image, scale) in the LDS with 16-byte coalesced loads
// (the 85-float record is not 16-byte periodic, so the span is read from the enclosing aligned window), then writes
// boxes / confs / probs / scores with 16-byte stores: the outputs of consecutive records are contiguous, and
// 16-byte aligned when C % 4 == 0 (decode_staged_kernel; other class counts use the 4-byte-per-lane kernel).
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

// ---- staged form (C % 4 == 0) --------------------------------------------------------------------------------------
constexpr int DRPB = 64;                 // records per workgroup
struct StagedArgs {
    DecodeArgs d;
    int chunks_per_img[3];               // ceil(g*g*3 / DRPB)
    int chunk_end[3];                    // cumulative chunk counts over the scales (all images)
};

__global__ void __launch_bounds__(256) decode_staged_kernel(const StagedArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [DRPB*F + 8] staged fields, then [DRPB] confs
    const DecodeArgs& a = sa.d;
    const int tid = threadIdx.x;
    int s = 0, ch = blockIdx.x;
    if (ch >= sa.chunk_end[1]) { s = 2; ch -= sa.chunk_end[1]; }
    else if (ch >= sa.chunk_end[0]) { s = 1; ch -= sa.chunk_end[0]; }
    const int cpi = sa.chunks_per_img[s];
    const int n = ch / cpi, j = ch - n * cpi;
    const int per_img = a.g_h[s] * a.g_w[s] * 3;
    const int rec0 = j * DRPB;
    const int nrec = min(DRPB, per_img - rec0);
    const int F = a.F, C = a.C;
    float* conf_s = lds + DRPB * F + 8;
    // stage: the aligned window around elements [e0, e0 + nrec*F) of fm[s]
    const long long e0 = ((long long)n * per_img + rec0) * F;
    const long long a0 = e0 & ~3LL;
    const int lead = (int)(e0 - a0);
    const int nf4 = (lead + nrec * F + 3) >> 2;
    const long long fm_elems = (long long)a.n * per_img * F;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.fm[s] + a0);
    for (int i = tid; i < nf4; i += 256) {
        f32x4 v;
        if (a0 + 4LL * i + 4 <= fm_elems) {
            v = src[i];
        } else {                          // the last window of the tensor may end inside a float4
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (a0 + 4LL * i + q < fm_elems) ? a.fm[s][a0 + 4LL * i + q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + 4 * i) = v;
    }
    __syncthreads();
    const float* rec = lds + lead;
    const size_t ob0 = (size_t)n * a.B + a.box_off[s] + rec0;      // first output box slot of the chunk
    if (tid < nrec) {
        const float* r = rec + tid * F;
        const float conf = sigmoidf_(r[4]);
        conf_s[tid] = conf;
        a.confs[ob0 + tid] = conf;
        const int lb = rec0 + tid;
        const int anc = lb % 3, cell = lb / 3;
        const int gy = cell / a.g_w[s], gx = cell - gy * a.g_w[s];
        // model.py:105-126: (sigmoid + offset) * ratio ; (exp * rescaled_anchor) * ratio
        const float cx = (sigmoidf_(r[0]) + (float)gx) * a.ratio_w[s];
        const float cy = (sigmoidf_(r[1]) + (float)gy) * a.ratio_h[s];
        const float bw = (expf(r[2]) * a.ra_w[s][anc]) * a.ratio_w[s];
        const float bh = (expf(r[3]) * a.ra_h[s][anc]) * a.ratio_h[s];
        f32x4 o;                                                    // model.py:182-188
        o[0] = cx - bw / 2.f;
        o[1] = cy - bh / 2.f;
        o[2] = cx + bw / 2.f;
        o[3] = cy + bh / 2.f;
        *reinterpret_cast<f32x4*>(a.boxes + (ob0 + tid) * 4) = o;
    }
    __syncthreads();
    // class probabilities (and scores): the chunk's outputs are nrec*C contiguous floats, 16-byte aligned
    const int nq = nrec * C / 4;
    f32x4* pout = reinterpret_cast<f32x4*>(a.probs + ob0 * C);
    f32x4* sout = a.scores ? reinterpret_cast<f32x4*>(a.scores + ob0 * C) : nullptr;
    for (int q = tid; q < nq; q += 256) {
        const int e = 4 * q;
        const int r = e / C, c = e - r * C;
        const float* f = rec + r * F + 5 + c;
        f32x4 pr;
#pragma unroll
        for (int k = 0; k < 4; ++k) pr[k] = sigmoidf_(f[k]);
        pout[q] = pr;
        if (sout) {
            const float conf = conf_s[r];
            f32x4 sc;
#pragma unroll
            for (int k = 0; k < 4; ++k) sc[k] = conf * pr[k];
            sout[q] = sc;
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
    const bool aligned = (((uintptr_t)fm1 | (uintptr_t)fm2 | (uintptr_t)fm3 | (uintptr_t)boxes | (uintptr_t)probs |
                           (uintptr_t)scores) & 15) == 0;
    // the staged kernel needs (DRPB*(5+C) + 8 + DRPB) floats of dynamic LDS; it is used while that stays inside the
    // 64 KB every launch may ask for without raising the function's dynamic-LDS limit (C <= 248); larger class counts
    // take the 4-byte-per-lane kernel below, which handles any class count (a 600-class head used to fail)
    const size_t staged_lds = (size_t)(DRPB * a.F + 8 + DRPB) * sizeof(float);
    if (class_num % 4 == 0 && aligned && staged_lds <= (size_t)64 * 1024) {
        StagedArgs sa;
        sa.d = a;
        long long chunks = 0;
        for (int s = 0; s < 3; ++s) {
            sa.chunks_per_img[s] = (a.g_h[s] * a.g_w[s] * 3 + DRPB - 1) / DRPB;
            chunks += (long long)n * sa.chunks_per_img[s];
            Y3_CHECK_ARG(chunks < (1LL << 31), "y3_decode: too many boxes");
            sa.chunk_end[s] = (int)chunks;
        }
        hipLaunchKernelGGL(decode_staged_kernel, dim3((unsigned)chunks), dim3(256), staged_lds, ctx->stream, sa);
        Y3_CHECK_HIP(hipGetLastError());
        return Y3_OK;
    }
    long long nb = (cum + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(decode_kernel, dim3((int)nb), dim3(256), 0, ctx->stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
