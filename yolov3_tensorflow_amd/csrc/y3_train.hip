// Training-path kernels other than the convolutions:
//   batch-norm in batch-statistics mode (forward statistics, apply, backward reduce/apply), the YOLO loss
//   forward+backward, gradient preparation (L2 term + norm), per-tensor clip + optimizer update, and the
//   small routing ops of the backward graph (2x2 upsample-backward, channel-slice accumulate, bias grad).
// All are HBM-bound streaming/reduction kernels: 16-byte-per-lane accesses along the channel axis,
// per-workgroup partial sums written to a scratch array and combined by a finalize kernel in a FIXED order
// (fp64 accumulation) so that every statistic and gradient is run-to-run deterministic (no float atomics).
#include <vector>
#include "y3_internal.h"

namespace {

constexpr int RED_BLOCKS = 512;   // partial-sum rows for the column reductions (2 workgroups per CU)

inline int grid_for(long long work, int cap = 256 * 16) {
    long long b = (work + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- column reductions over an [M][C] matrix -----------------------------------------------------
// MODE 0: (sum z, sum z^2)                                       -> BN forward statistics
// MODE 1: (sum g', sum g'*zhat), g' = dy * leaky'(z*scale+shift) -> BN backward (d beta, d gamma)
// Workgroup b handles rows b, b+gridDim, ...; thread t owns float4 column (t % C4) and row lane t / C4.
template <int MODE>
__global__ void __launch_bounds__(256) col_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ inv_std, long long M, int C,
                                                         float* __restrict__ partial /*[grid][2][C]*/) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [rows_per_pass][2][C]
    const int C4 = C >> 2;
    const int lanes_per_row = C4 < 256 ? C4 : 256;
    const int rows_per_pass = 256 / lanes_per_row;              // >= 1
    const int cols_per_thread = (C4 + 255) / 256;               // > 1 only when C > 1024
    const int tl = threadIdx.x % lanes_per_row, tr = threadIdx.x / lanes_per_row;
    for (int cc = 0; cc < cols_per_thread; ++cc) {
        const int c4 = tl + cc * 256;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        if (c4 < C4 && tr < rows_per_pass) {
            f32x4 sc, sh, mu, is;
            if (MODE == 1) {
                sc = *reinterpret_cast<const f32x4*>(scale + 4 * c4);
                sh = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
                mu = *reinterpret_cast<const f32x4*>(mean + 4 * c4);
                is = *reinterpret_cast<const f32x4*>(inv_std + 4 * c4);
            }
            for (long long r = (long long)blockIdx.x * rows_per_pass + tr; r < M;
                 r += (long long)gridDim.x * rows_per_pass) {
                if (MODE == 0) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(z + r * C + 4 * c4);
                    s0 += v;
                    s1 += v * v;
                } else {
                    // (plain loads: bn_apply_bwd reads both tensors again right behind this kernel, and what the caches keep of
                    // them counts - streaming loads here cost 0.15-0.35 ms per bs=64 step, profiles/r06_bn_nt_ab.txt)
                    const f32x4 v = *reinterpret_cast<const f32x4*>(z + r * C + 4 * c4);
                    f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * C + 4 * c4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float u = v[q] * sc[q] + sh[q];
                        g[q] = u > 0.f ? g[q] : 0.1f * g[q];
                    }
                    s0 += g;
                    s1 += g * ((v - mu) * is);
                }
            }
        }
        // combine the row lanes of this workgroup (fixed order)
        __syncthreads();
        if (c4 < C4 && tr < rows_per_pass) {
            *reinterpret_cast<f32x4*>(red + ((size_t)tr * 2 + 0) * C + 4 * c4) = s0;
            *reinterpret_cast<f32x4*>(red + ((size_t)tr * 2 + 1) * C + 4 * c4) = s1;
        }
        __syncthreads();
        if (c4 < C4 && tr == 0) {
            for (int k = 1; k < rows_per_pass; ++k) {
                s0 += *reinterpret_cast<const f32x4*>(red + ((size_t)k * 2 + 0) * C + 4 * c4);
                s1 += *reinterpret_cast<const f32x4*>(red + ((size_t)k * 2 + 1) * C + 4 * c4);
            }
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 0) * C + 4 * c4) = s0;
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 2 + 1) * C + 4 * c4) = s1;
        }
    }
}

// Sum partial[(b*stride_b) + idx] over b = 0..nblocks-1 in a FIXED order with one 256-thread workgroup:
// thread t accumulates b = t, t+256, ... in fp64, then a binary tree over the 256 lanes (deterministic).
__device__ __forceinline__ double block_sum_fixed(const float* __restrict__ partial, int nblocks, size_t stride_b,
                                                  size_t idx, double* sh /*[256]*/) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partial[(size_t)b * stride_b + idx];
    sh[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// The same for TWO columns at once (idx0, idx1): each sum is formed in exactly the order block_sum_fixed uses - one pass over
// the partial rows and one tree instead of two (the finalize kernels are latency-bound: 72 + 72 launches per train step).
__device__ __forceinline__ void block_sum2_fixed(const float* __restrict__ partial, int nblocks, size_t stride_b, size_t idx0,
                                                 size_t idx1, double* sh /*[512]*/, double& r0, double& r1) {
    double s0 = 0.0, s1 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        s0 += (double)partial[(size_t)b * stride_b + idx0];
        s1 += (double)partial[(size_t)b * stride_b + idx1];
    }
    sh[threadIdx.x] = s0;
    sh[256 + threadIdx.x] = s1;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sh[threadIdx.x] += sh[threadIdx.x + off];
            sh[256 + threadIdx.x] += sh[256 + threadIdx.x + off];
        }
        __syncthreads();
    }
    r0 = sh[0];
    r1 = sh[256];
}

// BN forward finalize: batch mean / biased variance, folded scale & shift for the apply pass, and the
// moving-statistics update  moving <- moving*decay + batch*(1-decay)  with the UNBIASED variance going
// into moving_variance (TF fused batch norm).
__global__ void __launch_bounds__(256) bn_stats_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                                double count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float decay,
                                                                float* __restrict__ mean, float* __restrict__ inv_std,
                                                                float* __restrict__ scale, float* __restrict__ shift,
                                                                float* __restrict__ moving_mean,
                                                                float* __restrict__ moving_var) {
    __shared__ double sh[512];
    const int c = blockIdx.x;   // one workgroup per channel
    double s0, s1;
    block_sum2_fixed(partial, nblocks, (size_t)2 * C, (size_t)c, (size_t)C + c, sh, s0, s1);
    if (threadIdx.x != 0) return;
    const double mu = s0 / count;
    double var = s1 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu, varf = (float)var;
    const float is = 1.0f / sqrtf(varf + eps);
    mean[c] = muf;
    inv_std[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - muf * sc;
    if (moving_mean) {
        const float unbiased = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
        moving_mean[c] = moving_mean[c] * decay + muf * (1.f - decay);
        moving_var[c] = moving_var[c] * decay + unbiased * (1.f - decay);
    }
}

// BN backward finalize: d beta, d gamma (the parameter gradients) and the coefficients of the apply pass.
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ inv_std,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                              float* __restrict__ coef /*[3][C]: a, b, c*/) {
    __shared__ double sh[512];
    const int c = blockIdx.x;
    double s0, s1;
    block_sum2_fixed(partial, nblocks, (size_t)2 * C, (size_t)c, (size_t)C + c, sh, s0, s1);
    if (threadIdx.x != 0) return;
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
    if (coef) {
        coef[c] = gamma[c] * inv_std[c];
        coef[C + c] = (float)(s0 / count);
        coef[2 * C + c] = (float)(s1 / count);
    }
}

// y = leaky(z*scale + shift) (+ residual)
__global__ void __launch_bounds__(256) bn_apply_fwd_kernel(const float* __restrict__ z,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ resid, long long total4,
                                                           int C4, int act, float* __restrict__ y) {
    const f32x4* z4 = reinterpret_cast<const f32x4*>(z);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(resid);
    const f32x4* sc4 = reinterpret_cast<const f32x4*>(scale);
    const f32x4* sh4 = reinterpret_cast<const f32x4*>(shift);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        // (z is not read again before the backward pass: a streaming load leaves the caches to y, which the next conv reads;
        // measured on the bs=64 step together with the two loads of bn_apply_bwd: 81.63 -> 81.22 ms, profiles/r06_bn_nt_ab.txt)
        f32x4 v = __builtin_nontemporal_load(z4 + i) * sc4[c] + sh4[c];
        if (act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
        }
        if (resid) v += __builtin_nontemporal_load(r4 + i);
        y4[i] = v;
    }
}

// dz = a * (g' - b - zhat * c),  g' = dy * leaky'(u), u = z*scale+shift, zhat = (z-mean)*inv_std
__global__ void __launch_bounds__(256) bn_apply_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ inv_std,
                                                           const float* __restrict__ coef, long long total4,
                                                           int C4, float* __restrict__ dz) {
    const f32x4* z4 = reinterpret_cast<const f32x4*>(z);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* o4 = reinterpret_cast<f32x4*>(dz);
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const f32x4 v = __builtin_nontemporal_load(z4 + i);      // (the last reader of z and of dy)
        f32x4 g = __builtin_nontemporal_load(g4 + i);
        f32x4 out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float u = v[q] * scale[c + q] + shift[c + q];
            const float gp = u > 0.f ? g[q] : 0.1f * g[q];
            const float zh = (v[q] - mean[c + q]) * inv_std[c + q];
            out[q] = coef[c + q] * (gp - coef[C + c + q] - zh * coef[2 * C + c + q]);
        }
        o4[i] = out;
    }
}

// ---- backward routing ----------------------------------------------------------------------------------
// dx[n,y,x,c] (+)= sum over the 2x2 block of g[n,2y+dy,2x+dx, c]   (g has row stride gC channels)
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const float* __restrict__ g, int gC, int n, int h,
                                                             int w, int c4n, int accumulate,
                                                             float* __restrict__ dx) {
    const long long total = (long long)n * h * w * c4n;
    f32x4* o4 = reinterpret_cast<f32x4*>(dx);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4n);
        long long r = i / c4n;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const int b = (int)(r / h);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx)
                s += *reinterpret_cast<const f32x4*>(
                    g + (((long long)b * 2 * h + 2 * y + dyy) * 2 * w + 2 * x + dxx) * gC + 4 * c);
        o4[i] = accumulate ? o4[i] + s : s;
    }
}

// dst[r, 0:c] (+)= src[r, off:off+c]   (src row stride sC)
__global__ void __launch_bounds__(256) slice_acc_kernel(const float* __restrict__ src, int sC, int off,
                                                        long long rows, int c4n, int accumulate,
                                                        float* __restrict__ dst) {
    const long long total = rows * c4n;
    f32x4* o4 = reinterpret_cast<f32x4*>(dst);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4n);
        const long long r = i / c4n;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * sC + off + 4 * c);
        o4[i] = accumulate ? o4[i] + v : v;
    }
}

// ---- loss (model.py:192-365) ---------------------------------------------------------------------------
struct LossArgs {
    const float* fm;      // [N,gh,gw,3,5+C] logits
    const float* y_true;  // [N,gh,gw,3,6+C]
    float* grad;          // [N,gh,gw,3,5+C] d(total loss)/d(fm)
    float* gt_boxes;      // scratch [N][cap][4]  (cx,cy,w,h) of the cells with object_mask == 1
    int* gt_count;        // scratch [N]
    float* partial;       // [blocks_x * N][4]
    int N, gh, gw, C, cap, vmax;   // cap = row stride of gt_boxes, vmax = boxes staged in LDS
    float ratio_h, ratio_w, img_h, img_w;
    float anc_w[3], anc_h[3];     // anchors of this scale (pixels)
    float ra_w[3], ra_h[3];       // anchors / ratio
    int label_smooth, focal;
    int grad_stride;              // floats per cell-row of 3 anchors in `grad` (>= 3*(5+C); padded for the dgrad)
};

__global__ void __launch_bounds__(256) loss_collect_gt_kernel(const LossArgs a) {
    const int n = blockIdx.y;
    const int cells = a.gh * a.gw * 3;
    const int T = 6 + a.C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const float* yt = a.y_true + ((size_t)n * cells + i) * T;
        if (yt[4] > 0.5f) {                               // tf.cast(object_mask, 'bool')
            const int slot = atomicAdd(&a.gt_count[n], 1);
            if (slot < a.cap) {
                float* o = a.gt_boxes + ((size_t)n * a.cap + slot) * 4;
                o[0] = yt[0]; o[1] = yt[1]; o[2] = yt[2]; o[3] = yt[3];
            }
        }
    }
}

__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + expf(-x)); }
// tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1 + exp(-|x|))
__device__ __forceinline__ float bce_(float z, float x) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }

// A wave owns 64 consecutive (cell, anchor) records of one image.
//   phase 1, one LANE per record: box decode, ignore mask (best IoU over the image's ground-truth boxes of this scale, read
//            as LDS broadcasts), xy / wh / conf terms and their five gradients, parked in the LDS;
//   phase 2, lanes over the 64 x (5+C) contiguous logits of the chunk: the class terms - only where the record holds an
//            object (object_mask is 0 for all but a few records, whose class logits and targets are then never read) -
//            and ALL the chunk's gradients written out contiguously.
// (The form this replaces ran one wave per record with the box / conf arithmetic on lane 0 alone: 0.84 ms for the
// 52-grid of a bs=64 batch, 530 MB; profiles/r02_train_c4_kernel_stats.csv.)
__global__ void __launch_bounds__(256) loss_kernel(const LossArgs a) {
    extern __shared__ float gts[];                         // [V][4] of this image
    __shared__ float red[4][4];
    __shared__ float rec_g[4][64][5];                      // per wave: the five box / conf gradients of its records
    __shared__ float rec_m[4][64], rec_w[4][64];           // object_mask, mix-up weight
    const int n = blockIdx.y;
    const int cells = a.gh * a.gw * 3;
    const int F = 5 + a.C, T = 6 + a.C;
    const int V = min(a.gt_count[n], a.vmax);
    for (int i = threadIdx.x; i < V * 4; i += 256) gts[i] = a.gt_boxes[(size_t)n * a.cap * 4 + i];
    __syncthreads();
    const float invN = 1.f / (float)a.N;
    const float invF = 1.f / (float)F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float l_xy = 0.f, l_wh = 0.f, l_conf = 0.f, l_cls = 0.f;   // lane-partial sums
    for (int i0 = (blockIdx.x * 4 + wave) * 64; i0 < cells; i0 += gridDim.x * 256) {
        // ---- phase 1 ------------------------------------------------------------------------------------------
        const int i = i0 + lane;
        const bool valid = i < cells;
        float m = 0.f, mixw = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
        if (valid) {
            const int anc = i % 3;
            const int cell = i / 3;
            const int gy = cell / a.gw, gx = cell - gy * a.gw;
            const float* f = a.fm + ((size_t)n * cells + i) * F;
            const float* yt = a.y_true + ((size_t)n * cells + i) * T;
            const float f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], xc = f[4];
            const float y0 = yt[0], y1 = yt[1], y2 = yt[2], y3 = yt[3];
            m = yt[4];                                   // object_mask
            mixw = yt[T - 1];
            // reorg_layer (model.py:96-126)
            const float sx = sigmoid_(f0), sy = sigmoid_(f1);
            const float ex = expf(f2), ey = expf(f3);
            const float px = (sx + (float)gx) * a.ratio_w, py = (sy + (float)gy) * a.ratio_h;
            const float pw = (ex * a.ra_w[anc]) * a.ratio_w, ph = (ey * a.ra_h[anc]) * a.ratio_h;
            // ignore mask (model.py:220-237): best IoU with this image's GT boxes of THIS scale < 0.5
            float best = -INFINITY;
            for (int v = 0; v < V; ++v) {
                const float tx = gts[4 * v], ty = gts[4 * v + 1], tw = gts[4 * v + 2], th = gts[4 * v + 3];
                const float iw = fmaxf(fminf(px + pw / 2.f, tx + tw / 2.f) - fmaxf(px - pw / 2.f, tx - tw / 2.f), 0.f);
                const float ih = fmaxf(fminf(py + ph / 2.f, ty + th / 2.f) - fmaxf(py - ph / 2.f, ty - th / 2.f), 0.f);
                const float inter = iw * ih;
                best = fmaxf(best, inter / (pw * ph + tw * th - inter + 1e-10f));
            }
            const float ignore = best < 0.5f ? 1.f : 0.f;
            const float bls = 2.f - (y2 / a.img_w) * (y3 / a.img_h);
            const float wgt = m * bls * mixw;
            // xy (model.py:248-249,276)
            const float txy0 = y0 / a.ratio_w - (float)gx, txy1 = y1 / a.ratio_h - (float)gy;
            const float pxy0 = px / a.ratio_w - (float)gx, pxy1 = py / a.ratio_h - (float)gy;
            const float d0 = txy0 - pxy0, d1 = txy1 - pxy1;
            l_xy += (d0 * d0 + d1 * d1) * wgt;
            g0 = -2.f * d0 * wgt * sx * (1.f - sx) * invN;
            g1 = -2.f * d1 * wgt * sy * (1.f - sy) * invN;
            // wh (model.py:254-262,277)
            float tt0 = y2 / a.anc_w[anc], tt1 = y3 / a.anc_h[anc];
            float pt0 = pw / a.anc_w[anc], pt1 = ph / a.anc_h[anc];
            tt0 = tt0 == 0.f ? 1.f : tt0; tt1 = tt1 == 0.f ? 1.f : tt1;
            const bool pz0 = pt0 == 0.f, pz1 = pt1 == 0.f;
            pt0 = pz0 ? 1.f : pt0; pt1 = pz1 ? 1.f : pt1;
            const bool in0 = !pz0 && pt0 >= 1e-9f && pt0 <= 1e9f, in1 = !pz1 && pt1 >= 1e-9f && pt1 <= 1e9f;
            const float e0 = logf(fminf(fmaxf(tt0, 1e-9f), 1e9f)) - logf(fminf(fmaxf(pt0, 1e-9f), 1e9f));
            const float e1 = logf(fminf(fmaxf(tt1, 1e-9f), 1e9f)) - logf(fminf(fmaxf(pt1, 1e-9f), 1e9f));
            l_wh += (e0 * e0 + e1 * e1) * wgt;
            g2 = in0 ? -2.f * e0 * wgt * invN : 0.f;     // d log(exp(t)*const)/dt = 1 inside the clip range
            g3 = in1 ? -2.f * e1 * wgt * invN : 0.f;
            // conf (model.py:280-292)
            const float pc = sigmoid_(xc);
            const float cmask = m + (1.f - m) * ignore;
            const float b = bce_(m, xc);
            float lc = cmask * b;
            float gc = cmask * (pc - m);
            if (a.focal) {
                const float dm = m - pc;
                const float fo = dm * dm;                                   // alpha=1, gamma=2
                gc = cmask * ((pc - m) * fo + b * (-2.f * dm * pc * (1.f - pc)));
                lc *= fo;
            }
            l_conf += lc * mixw;
            g4 = gc * mixw * invN;
        }
        rec_g[wave][lane][0] = g0; rec_g[wave][lane][1] = g1; rec_g[wave][lane][2] = g2;
        rec_g[wave][lane][3] = g3; rec_g[wave][lane][4] = g4;
        rec_m[wave][lane] = m; rec_w[wave][lane] = mixw;
        // (a wave reads back only what it wrote itself: no workgroup barrier, the LDS counter wait orders it)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase 2 ------------------------------------------------------------------------------------------
        const int nrec = min(64, cells - i0);
        const float* fchunk = a.fm + ((size_t)n * cells + i0) * F;
        const float* ychunk = a.y_true + ((size_t)n * cells + i0) * T;
        const float delta = 0.01f;
        for (int e = lane; e < nrec * F; e += 64) {
            const int r = (int)(((float)e + 0.5f) * invF);       // e / F (e < 64 * F: the product is exact enough)
            const int field = e - r * F;
            const int rec = i0 + r;
            const int cell = rec / 3, anc = rec - cell * 3;
            float gv;
            if (field < 5) {
                gv = rec_g[wave][r][field];
            } else {
                // class (model.py:296-302)
                const float mr = rec_m[wave][r];
                gv = 0.f;
                if (mr != 0.f) {
                    const float wr = rec_w[wave][r];
                    float tgt = ychunk[(size_t)r * T + field];
                    if (a.label_smooth) tgt = (1.f - delta) * tgt + delta * 1.f / (float)a.C;
                    const float x = fchunk[e];
                    l_cls += mr * bce_(tgt, x) * wr;
                    gv = mr * wr * (sigmoid_(x) - tgt) * invN;
                }
            }
            a.grad[((size_t)n * (cells / 3) + cell) * a.grad_stride + anc * F + field] = gv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // the next chunk overwrites the wave's LDS rows
    }
    // wave reduction, then the four waves, in a fixed order
    float vals[4] = {l_xy, l_wh, l_conf, l_cls};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = vals[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        a.partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = v;
    }
}

__global__ void __launch_bounds__(256) loss_finalize_kernel(const float* __restrict__ partial, int nparts, float invN,
                                                            int accumulate, float* __restrict__ out4) {
    __shared__ double sh[256];
    const int q = blockIdx.x;   // 4 workgroups: xy, wh, conf, class
    const double s = block_sum_fixed(partial, nparts, (size_t)4, (size_t)q, sh);
    if (threadIdx.x == 0) {
        const float v = (float)(s * (double)invN);
        out4[q] = accumulate ? out4[q] + v : v;
    }
}

// ---- gradient preparation and optimizer update ---------------------------------------------------------
// g += wd * w (slim.l2_regularizer gradient) and per-workgroup partial sums of g^2
__global__ void __launch_bounds__(256) grad_prepare_kernel(float* __restrict__ g, const float* __restrict__ w,
                                                           float wd, float gscale, long long n,
                                                           float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = g[i] * gscale;
        if (wd != 0.f) v += wd * w[i];
        g[i] = v;
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) norm_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                            float* __restrict__ norm) {
    __shared__ double sh[256];
    const double s = block_sum_fixed(partial, nparts, (size_t)1, (size_t)0, sh);
    if (threadIdx.x == 0) norm[0] = (float)sqrt(s);
}

// tf.clip_by_norm(g, clip) then the TF1 update rule.  kind: 0 sgd, 1 momentum, 2 adam,
// 3 rmsprop.  hp: lr (or lr_t for adam), momentum, decay/beta1, beta2, eps.
__global__ void __launch_bounds__(256) optimizer_update_kernel(float* __restrict__ w, float* __restrict__ g,
                                                               float* __restrict__ s0, float* __restrict__ s1,
                                                               const float* __restrict__ norm, float clip, int kind,
                                                               float lr, float momentum, float decay, float beta2,
                                                               float eps, long long n) {
    const float nn = norm[0];
    const float factor = clip / fmaxf(nn, clip);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * factor;
        g[i] = gi;                                   // the clipped gradient stays observable
        float wi = w[i];
        if (kind == 0) {
            wi -= lr * gi;
        } else if (kind == 1) {
            const float acc = s0[i] * momentum + gi;
            s0[i] = acc;
            wi -= lr * acc;
        } else if (kind == 2) {
            const float m = s0[i] * decay + (1.f - decay) * gi;          // decay = beta1
            const float v = s1[i] * beta2 + (1.f - beta2) * gi * gi;
            s0[i] = m; s1[i] = v;
            wi -= lr * m / (sqrtf(v) + eps);                              // lr = lr_t
        } else {
            const float ms = s0[i] * decay + (1.f - decay) * gi * gi;
            const float mom = s1[i] * momentum + lr * gi / sqrtf(ms + eps);
            s0[i] = ms; s1[i] = mom;
            wi -= mom;
        }
        w[i] = wi;
    }
}

// ---- multi-tensor form: every trainable tensor of the step in THREE launches (prepare / norms / update) --------------
// A tensor is cut into chunks of MT_CHUNK elements; workgroup b owns global chunk b and finds its tensor by a binary
// search over the tensors' first-chunk indices.  Per-tensor norms are segmented sums of the per-chunk partials in a
// fixed order (deterministic; no float atomics).
constexpr int MT_CHUNK = 8192;          // elements per workgroup: 256 threads x 8 float4
struct MtDesc {                         // device copy of y3_param_desc + its first global chunk
    float* w; float* g; float* s0; float* s1;
    long long n;
    float wd;
    int chunk_begin;
};
static_assert(sizeof(MtDesc) == 48, "MtDesc layout");

__device__ __forceinline__ int mt_find(const MtDesc* __restrict__ d, int count, int chunk) {
    int lo = 0, hi = count - 1;         // last tensor whose chunk_begin <= chunk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].chunk_begin <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// g <- g*gscale + wd*w ; partial[chunk] = sum of g^2 over the chunk
__global__ void __launch_bounds__(256) mt_prepare_kernel(const MtDesc* __restrict__ descs, int count, float gscale,
                                                         float* __restrict__ partial) {
    __shared__ float red[4];
    const int t = mt_find(descs, count, blockIdx.x);
    const MtDesc d = descs[t];
    const long long base = (long long)(blockIdx.x - d.chunk_begin) * MT_CHUNK;
    const long long end = base + MT_CHUNK < d.n ? base + MT_CHUNK : d.n;
    float s = 0.f;
    if ((d.n & 3) == 0) {               // every view is 16-byte aligned and (n % 4 == 0) a whole number of float4
        for (long long i = base + 4 * threadIdx.x; i < end; i += 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(d.g + i) * gscale;
            if (d.wd != 0.f) v += d.wd * *reinterpret_cast<const f32x4*>(d.w + i);
            *reinterpret_cast<f32x4*>(d.g + i) = v;
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
    } else {
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            float v = d.g[i] * gscale;
            if (d.wd != 0.f) v += d.wd * d.w[i];
            d.g[i] = v;
            s += v * v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm[t] = sqrt(sum of tensor t's chunk partials) — one workgroup per tensor, fixed order, fp64
__global__ void __launch_bounds__(256) mt_norms_kernel(const MtDesc* __restrict__ descs, int count, int total_chunks,
                                                       const float* __restrict__ partial, float* __restrict__ norm) {
    __shared__ double sh[256];
    const int t = blockIdx.x;
    const int b = descs[t].chunk_begin, e = t + 1 < count ? descs[t + 1].chunk_begin : total_chunks;
    const double s = block_sum_fixed(partial + b, e - b, (size_t)1, (size_t)0, sh);
    if (threadIdx.x == 0) norm[t] = (float)sqrt(s);
}

__device__ __forceinline__ void mt_update_one(int kind, float& wi, float& gi, float* s0, float* s1, long long i,
                                              float factor, float lr, float momentum, float decay, float beta2,
                                              float eps) {
    gi *= factor;
    if (kind == 0) {
        wi -= lr * gi;
    } else if (kind == 1) {
        const float acc = s0[i] * momentum + gi;
        s0[i] = acc;
        wi -= lr * acc;
    } else if (kind == 2) {
        const float m = s0[i] * decay + (1.f - decay) * gi;          // decay = beta1
        const float v = s1[i] * beta2 + (1.f - beta2) * gi * gi;
        s0[i] = m; s1[i] = v;
        wi -= lr * m / (sqrtf(v) + eps);                              // lr = lr_t
    } else {
        const float ms = s0[i] * decay + (1.f - decay) * gi * gi;
        const float mom = s1[i] * momentum + lr * gi / sqrtf(ms + eps);
        s0[i] = ms; s1[i] = mom;
        wi -= mom;
    }
}

// tf.clip_by_norm per tensor, then the TF1 update rule — the same arithmetic as optimizer_update_kernel
__global__ void __launch_bounds__(256) mt_update_kernel(const MtDesc* __restrict__ descs, int count,
                                                        const float* __restrict__ norm, float clip, int kind, float lr,
                                                        float momentum, float decay, float beta2, float eps) {
    const int t = mt_find(descs, count, blockIdx.x);
    const MtDesc d = descs[t];
    const float factor = clip / fmaxf(norm[t], clip);
    const long long base = (long long)(blockIdx.x - d.chunk_begin) * MT_CHUNK;
    const long long end = base + MT_CHUNK < d.n ? base + MT_CHUNK : d.n;
    for (long long i = base + threadIdx.x; i < end; i += 256) {
        float gi = d.g[i], wi = d.w[i];
        mt_update_one(kind, wi, gi, d.s0, d.s1, i, factor, lr, momentum, decay, beta2, eps);
        d.g[i] = gi;                                  // the clipped gradient stays observable
        d.w[i] = wi;
    }
}

// sum over rows of an [M][C] matrix with arbitrary C (bias gradient of the detection convs, C = 3*(5+classes))
__global__ void __launch_bounds__(256) col_sum_scalar_kernel(const float* __restrict__ x, long long M, int C,
                                                             float* __restrict__ partial /*[grid][C]*/) {
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (long long r = blockIdx.x; r < M; r += gridDim.x) s += x[r * C + c];
        partial[(size_t)blockIdx.x * C + c] = s;
    }
}
// dst[r][0:c_dst] = src[r][0:c_src] zero-extended (c_dst >= c_src): 16-byte-aligned rows for the 3*(5+C)-wide tensors
__global__ void __launch_bounds__(256) pad_channels_kernel(const float* __restrict__ src, int cs, long long rows,
                                                           int cd, float* __restrict__ dst) {
    const long long total = rows * cd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cd);
        const long long r = i / cd;
        dst[i] = c < cs ? src[r * cs + c] : 0.f;
    }
}

__global__ void __launch_bounds__(256) col_sum_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                               float* __restrict__ out) {
    __shared__ double sh[256];
    const double s = block_sum_fixed(partial, nblocks, (size_t)C, (size_t)blockIdx.x, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s;
}

// box_iou (model.py:307-345): IoU of every predicted box (cx,cy,w,h) with every ground-truth box (cx,cy,w,h)
__global__ void __launch_bounds__(256) box_iou_kernel(const float* __restrict__ pred, long long np,
                                                      const float* __restrict__ gt, int v, float* __restrict__ iou) {
    const long long total = np * v;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pi = i / v;
        const int gi = (int)(i - pi * v);
        const float px = pred[pi * 4], py = pred[pi * 4 + 1], pw = pred[pi * 4 + 2], ph = pred[pi * 4 + 3];
        const float tx = gt[gi * 4], ty = gt[gi * 4 + 1], tw = gt[gi * 4 + 2], th = gt[gi * 4 + 3];
        const float iw = fmaxf(fminf(px + pw / 2.f, tx + tw / 2.f) - fmaxf(px - pw / 2.f, tx - tw / 2.f), 0.f);
        const float ih = fmaxf(fminf(py + ph / 2.f, ty + th / 2.f) - fmaxf(py - ph / 2.f, ty - th / 2.f), 0.f);
        const float inter = iw * ih;
        iou[i] = inter / (pw * ph + tw * th - inter + 1e-10f);
    }
}

// ---- target assignment (utils/data_utils.py:51-115 `process_box`) ---------------------------------------
// One thread per image walks its boxes IN ORDER (later boxes overwrite earlier ones in the same cell/anchor;
// class one-hots are not cleared on overwrite, exactly like the reference).  fp32 arithmetic in numpy's order.
struct TargetArgs {
    const float* boxes;   // [N][kmax][5]  x0,y0,x1,y1,mix_w
    const int* labels;    // [N][kmax]
    const int* counts;    // [N]
    float* y[3];          // y_true_13 / 26 / 52 : [N][g][g][3][6+C]
    int N, kmax, C, img_w, img_h;
    float anc_w[9], anc_h[9];
};

__global__ void __launch_bounds__(256) target_fill_kernel(float* __restrict__ y, long long cells, int T) {
    // zeros, with the trailing mix-up weight preset to 1 (utils/data_utils.py:70-77)
    const long long total = cells * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        y[i] = (i % T) == T - 1 ? 1.f : 0.f;
}

__global__ void target_assign_kernel(const TargetArgs a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const int T = 6 + a.C;
    const int K = min(a.counts[n], a.kmax);
    for (int i = 0; i < K; ++i) {
        const float* b = a.boxes + ((size_t)n * a.kmax + i) * 5;
        const float cx = (b[0] + b[2]) / 2.f, cy = (b[1] + b[3]) / 2.f;
        const float bw = b[2] - b[0], bh = b[3] - b[1];
        int best = 0;
        float best_iou = -INFINITY;
        for (int k = 0; k < 9; ++k) {
            const float w = fminf(bw / 2.f, a.anc_w[k] / 2.f) - fmaxf(-bw / 2.f, -a.anc_w[k] / 2.f);
            const float h = fminf(bh / 2.f, a.anc_h[k] / 2.f) - fmaxf(-bh / 2.f, -a.anc_h[k] / 2.f);
            const float iou = (w * h) / (bw * bh + a.anc_w[k] * a.anc_h[k] - w * h + 1e-10f);
            if (iou > best_iou) { best_iou = iou; best = k; }      // np.argmax: first maximum
        }
        const int group = 2 - best / 3;                              // 0 -> 13-grid
        const float stride = best / 3 == 0 ? 8.f : (best / 3 == 1 ? 16.f : 32.f);
        const int gx = (int)floorf(cx / stride), gy = (int)floorf(cy / stride);
        const int gw = a.img_w / (int)stride, gh = a.img_h / (int)stride;
        if (gx < 0 || gy < 0 || gx >= gw || gy >= gh) continue;     // numpy would raise IndexError here
        float* y = a.y[group] + ((((size_t)n * gh + gy) * gw + gx) * 3 + best % 3) * T;
        y[0] = cx; y[1] = cy; y[2] = bw; y[3] = bh; y[4] = 1.f;
        const int c = a.labels[(size_t)n * a.kmax + i];
        if (c >= 0 && c < a.C) y[5 + c] = 1.f;
        y[T - 1] = b[4];
    }
}

}  // namespace




// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
extern "C" size_t y3_reduce_scratch_bytes(int c) { return (size_t)RED_BLOCKS * 2 * (size_t)(c > 0 ? c : 0) * sizeof(float); }

static int reduce_launch(y3_ctx* ctx, int mode, const float* z, const float* dy, const float* scale,
                         const float* shift, const float* mean, const float* inv_std, long long rows, int c,
                         float* scratch, int* nblocks_out) {
    const int C4 = c / 4;
    const int lanes_per_row = C4 < 256 ? C4 : 256;
    const int rows_per_pass = 256 / lanes_per_row;
    long long nb = (rows + rows_per_pass - 1) / rows_per_pass;
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    const size_t lds = (size_t)rows_per_pass * 2 * c * sizeof(float);
    if (mode == 0)
        hipLaunchKernelGGL(col_reduce_kernel<0>, dim3((int)nb), dim3(256), lds, ctx->stream, z, dy, scale, shift,
                           mean, inv_std, rows, c, scratch);
    else
        hipLaunchKernelGGL(col_reduce_kernel<1>, dim3((int)nb), dim3(256), lds, ctx->stream, z, dy, scale, shift,
                           mean, inv_std, rows, c, scratch);
    Y3_CHECK_HIP(hipGetLastError());
    *nblocks_out = (int)nb;
    return Y3_OK;
}

extern "C" int y3_bn_train_stats(y3_ctx* ctx, const float* z, long long rows, int c, const float* gamma,
                                 const float* beta, float eps, float decay, float* mean, float* inv_std,
                                 float* scale, float* shift, float* moving_mean, float* moving_var,
                                 float* scratch) {
    Y3_CHECK_ARG(ctx && z && gamma && beta && mean && inv_std && scale && shift && scratch,
                 "y3_bn_train_stats: null argument");
    Y3_CHECK_ARG(rows > 0 && c > 0 && c % 4 == 0, "y3_bn_train_stats: bad shape rows=%lld c=%d", rows, c);
    Y3_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "y3_bn_train_stats: moving stats must come in pairs");
    int nb = 0;
    if (int rc = reduce_launch(ctx, 0, z, nullptr, nullptr, nullptr, nullptr, nullptr, rows, c, scratch, &nb)) return rc;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c), dim3(256), 0, ctx->stream, scratch, nb, c,
                       (double)rows, gamma, beta, eps, decay, mean, inv_std, scale, shift, moving_mean, moving_var);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_bn_train_stats_partials(y3_ctx* ctx, const float* partial, int nblocks, long long rows, int c,
                                          const float* gamma, const float* beta, float eps, float decay, float* mean,
                                          float* inv_std, float* scale, float* shift, float* moving_mean,
                                          float* moving_var) {
    Y3_CHECK_ARG(ctx && partial && gamma && beta && mean && inv_std && scale && shift,
                 "y3_bn_train_stats_partials: null argument");
    Y3_CHECK_ARG(nblocks > 0 && rows > 0 && c > 0, "y3_bn_train_stats_partials: bad shape");
    Y3_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr),
                 "y3_bn_train_stats_partials: moving stats must come in pairs");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c), dim3(256), 0, ctx->stream, partial, nblocks, c, (double)rows,
                       gamma, beta, eps, decay, mean, inv_std, scale, shift, moving_mean, moving_var);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_bn_apply_fwd(y3_ctx* ctx, const float* z, const float* scale, const float* shift,
                               const float* residual, long long rows, int c, int act, float* y) {
    Y3_CHECK_ARG(ctx && z && scale && shift && y, "y3_bn_apply_fwd: null argument");
    Y3_CHECK_ARG(rows > 0 && c > 0 && c % 4 == 0, "y3_bn_apply_fwd: bad shape");
    const long long total4 = rows * (c / 4);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, ctx->stream, z, scale, shift,
                       residual, total4, c / 4, act, y);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_bn_train_bwd(y3_ctx* ctx, const float* z, const float* dy, const float* gamma,
                               const float* scale, const float* shift, const float* mean, const float* inv_std,
                               long long rows, int c, float* dgamma, float* dbeta, float* dz, float* scratch) {
    Y3_CHECK_ARG(ctx && z && dy && gamma && scale && shift && mean && inv_std && dz && scratch,
                 "y3_bn_train_bwd: null argument");
    Y3_CHECK_ARG(rows > 0 && c > 0 && c % 4 == 0, "y3_bn_train_bwd: bad shape");
    int nb = 0;
    if (int rc = reduce_launch(ctx, 1, z, dy, scale, shift, mean, inv_std, rows, c, scratch, &nb)) return rc;
    float* coef = scratch + (size_t)RED_BLOCKS * 2 * c;   // scratch has room for 3*C more (see y3_bn_bwd_scratch_bytes)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, ctx->stream, scratch, nb, c,
                       (double)rows, gamma, inv_std, dbeta, dgamma, coef);
    Y3_CHECK_HIP(hipGetLastError());
    const long long total4 = rows * (c / 4);
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, ctx->stream, z, dy, scale, shift,
                       mean, inv_std, coef, total4, c / 4, dz);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

int y3_bn_train_bwd_partials(y3_ctx* ctx, const float* z, const float* dy, const float* gamma, const float* scale, const float* shift,
                             const float* mean, const float* inv_std, long long rows, int c, const float* partial, int nblocks,
                             float* dgamma, float* dbeta, float* dz, float* scratch) {
    Y3_CHECK_ARG(ctx && z && dy && gamma && scale && shift && mean && inv_std && dz && scratch && partial && nblocks > 0,
                 "y3_bn_train_bwd_partials: null argument");
    Y3_CHECK_ARG(rows > 0 && c > 0 && c % 4 == 0, "y3_bn_train_bwd_partials: bad shape");
    float* coef = scratch + (size_t)RED_BLOCKS * 2 * c;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, ctx->stream, partial, nblocks, c,
                       (double)rows, gamma, inv_std, dbeta, dgamma, coef);
    Y3_CHECK_HIP(hipGetLastError());
    const long long total4 = rows * (c / 4);
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, ctx->stream, z, dy, scale, shift,
                       mean, inv_std, coef, total4, c / 4, dz);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" size_t y3_bn_bwd_scratch_bytes(int c) { return y3_reduce_scratch_bytes(c) + (size_t)3 * (c > 0 ? c : 0) * sizeof(float); }

extern "C" int y3_bias_grad(y3_ctx* ctx, const float* dy, long long rows, int c, float* dbias, float* scratch) {
    Y3_CHECK_ARG(ctx && dy && dbias && scratch, "y3_bias_grad: null argument");
    Y3_CHECK_ARG(rows > 0 && c > 0, "y3_bias_grad: bad shape");
    const int nb = (int)(rows < RED_BLOCKS ? rows : RED_BLOCKS);
    hipLaunchKernelGGL(col_sum_scalar_kernel, dim3(nb), dim3(256), 0, ctx->stream, dy, rows, c, scratch);
    Y3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(col_sum_finalize_kernel, dim3(c), dim3(256), 0, ctx->stream, scratch, nb, c, dbias);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_upsample2x_bwd(y3_ctx* ctx, const float* g, int g_channels, int n, int h, int w, int c,
                                 int accumulate, float* dx) {
    Y3_CHECK_ARG(ctx && g && dx, "y3_upsample2x_bwd: null argument");
    Y3_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && g_channels >= c && g_channels % 4 == 0,
                 "y3_upsample2x_bwd: bad shape");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 4))), dim3(256), 0,
                       ctx->stream, g, g_channels, n, h, w, c / 4, accumulate, dx);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_slice_accumulate(y3_ctx* ctx, const float* src, int src_channels, int offset, long long rows,
                                   int c, int accumulate, float* dst) {
    Y3_CHECK_ARG(ctx && src && dst, "y3_slice_accumulate: null argument");
    Y3_CHECK_ARG(rows > 0 && c > 0 && c % 4 == 0 && offset % 4 == 0 && src_channels % 4 == 0 &&
                     offset + c <= src_channels, "y3_slice_accumulate: bad shape");
    hipLaunchKernelGGL(slice_acc_kernel, dim3(grid_for(rows * (c / 4))), dim3(256), 0, ctx->stream, src,
                       src_channels, offset, rows, c / 4, accumulate, dst);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

// ---- loss ---------------------------------------------------------------------------------------------
extern "C" size_t y3_loss_scratch_bytes(int n, int gh, int gw) {
    if (n <= 0 || gh <= 0 || gw <= 0) return 0;
    const size_t cells = (size_t)gh * gw * 3;
    const size_t blocks = (cells + 255) / 256;
    return (size_t)n * cells * 4 * sizeof(float) + 256 + (size_t)n * sizeof(int) + 256 +
           blocks * n * 4 * sizeof(float) + 256;
}

extern "C" int y3_loss_layer(y3_ctx* ctx, const float* feature_map, const float* y_true, int n, int gh, int gw,
                             int class_num, int img_h, int img_w, const float* anchors3_host, int use_label_smooth,
                             int use_focal_loss, int accumulate, float* loss4, float* grad, int grad_stride,
                             void* scratch, size_t scratch_bytes) {
    Y3_CHECK_ARG(ctx && feature_map && y_true && anchors3_host && loss4 && grad && scratch,
                 "y3_loss_layer: null argument");
    Y3_CHECK_ARG(n > 0 && gh > 0 && gw > 0 && class_num > 0 && img_h > 0 && img_w > 0, "y3_loss_layer: bad shape");
    Y3_CHECK_ARG(scratch_bytes >= y3_loss_scratch_bytes(n, gh, gw), "y3_loss_layer: scratch too small");
    Y3_CHECK_ARG(grad_stride >= 3 * (5 + class_num), "y3_loss_layer: grad_stride smaller than 3*(5+C)");
    LossArgs a;
    a.fm = feature_map; a.y_true = y_true; a.grad = grad;
    a.N = n; a.gh = gh; a.gw = gw; a.C = class_num;
    const int cells = gh * gw * 3;
    a.cap = cells;
    const int blocks = (cells + 255) / 256;   // also the loss kernel's grid.x: each workgroup walks its records
    char* p = static_cast<char*>(scratch);
    a.gt_boxes = reinterpret_cast<float*>(p); p += (((size_t)n * cells * 4 * sizeof(float)) + 255) & ~(size_t)255;
    a.gt_count = reinterpret_cast<int*>(p);   p += (((size_t)n * sizeof(int)) + 255) & ~(size_t)255;
    a.partial = reinterpret_cast<float*>(p);
    a.ratio_h = (float)((double)img_h / gh); a.ratio_w = (float)((double)img_w / gw);
    a.img_h = (float)img_h; a.img_w = (float)img_w;
    for (int k = 0; k < 3; ++k) {
        a.anc_w[k] = anchors3_host[2 * k]; a.anc_h[k] = anchors3_host[2 * k + 1];
        a.ra_w[k] = a.anc_w[k] / a.ratio_w; a.ra_h[k] = a.anc_h[k] / a.ratio_h;
    }
    a.label_smooth = use_label_smooth; a.focal = use_focal_loss; a.grad_stride = grad_stride;
    Y3_CHECK_HIP(hipMemsetAsync(a.gt_count, 0, (size_t)n * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(loss_collect_gt_kernel, dim3(blocks, n), dim3(256), 0, ctx->stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    // GT boxes of one image are staged in LDS: up to 2048 per scale per image (32 KB) — far above any real
    // annotation count (the reference's datasets have tens of boxes per image)
    a.vmax = cells < 2048 ? cells : 2048;
    const size_t lds = (size_t)a.vmax * 4 * sizeof(float);
    hipLaunchKernelGGL(loss_kernel, dim3(blocks, n), dim3(256), lds, ctx->stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(4), dim3(256), 0, ctx->stream, a.partial, blocks * n,
                       1.f / (float)n, accumulate, loss4);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

// ---- optimizer ----------------------------------------------------------------------------------------
extern "C" size_t y3_optimizer_scratch_bytes(void) { return (size_t)(RED_BLOCKS + 64) * sizeof(float); }

extern "C" int y3_clip_update(y3_ctx* ctx, int kind, float* w, float* g, float* slot0, float* slot1, long long n,
                              float weight_decay, float grad_scale, float clip_norm, float lr, float momentum,
                              float decay, float beta2, float eps, float* scratch) {
    Y3_CHECK_ARG(ctx && w && g && scratch, "y3_clip_update: null argument");
    Y3_CHECK_ARG(n > 0, "y3_clip_update: empty tensor");
    Y3_CHECK_ARG(kind >= 0 && kind <= 3, "y3_clip_update: unknown optimizer kind %d", kind);
    Y3_CHECK_ARG(kind == 0 || slot0, "y3_clip_update: optimizer slot missing");
    Y3_CHECK_ARG(kind < 2 || slot1, "y3_clip_update: second optimizer slot missing");
    int nb = grid_for(n, RED_BLOCKS);
    float* norm = scratch + RED_BLOCKS;
    hipLaunchKernelGGL(grad_prepare_kernel, dim3(nb), dim3(256), 0, ctx->stream, g, w, weight_decay, grad_scale, n, scratch);
    Y3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, scratch, nb, norm);
    Y3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(optimizer_update_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, w, g, slot0, slot1,
                       norm, clip_norm, kind, lr, momentum, decay, beta2, eps, n);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

static long long mt_chunks(long long n) { return (n + MT_CHUNK - 1) / MT_CHUNK; }

extern "C" size_t y3_clip_update_multi_scratch_bytes(const y3_param_desc* params, int count) {
    if (!params || count <= 0) return 0;
    long long chunks = 0;
    for (int i = 0; i < count; ++i) chunks += params[i].n > 0 ? mt_chunks(params[i].n) : 0;
    // [descs][partial per chunk][norm per tensor]
    return (((size_t)count * sizeof(MtDesc) + 255) & ~(size_t)255) + (((size_t)chunks * 4 + 255) & ~(size_t)255) +
           (size_t)count * 4 + 256;
}

extern "C" int y3_clip_update_multi(y3_ctx* ctx, int kind, const y3_param_desc* params, int count, float grad_scale,
                                    float clip_norm, float lr, float momentum, float decay, float beta2, float eps,
                                    void* scratch, size_t scratch_bytes) {
    Y3_CHECK_ARG(ctx && params && scratch, "y3_clip_update_multi: null argument");
    Y3_CHECK_ARG(count > 0 && count <= 65536, "y3_clip_update_multi: bad tensor count %d", count);
    Y3_CHECK_ARG(kind >= 0 && kind <= 3, "y3_clip_update_multi: unknown optimizer kind %d", kind);
    Y3_CHECK_ARG(scratch_bytes >= y3_clip_update_multi_scratch_bytes(params, count) && ((uintptr_t)scratch & 15) == 0,
                 "y3_clip_update_multi: scratch too small or misaligned");
    // host staging of the device descriptors: the context's pinned buffer, handed out only once the previous upload
    // out of it has completed (an event behind that copy), so the asynchronous copy below can never race with this
    // or a later call's rewrite of the buffer
    void* stage = nullptr;
    {
        const int rc = y3_ctx_stage_acquire(ctx, (size_t)count * sizeof(MtDesc), &stage);
        if (rc != Y3_OK) return rc;
    }
    MtDesc* host = static_cast<MtDesc*>(stage);
    long long chunks = 0;
    for (int i = 0; i < count; ++i) {
        const y3_param_desc& q = params[i];
        Y3_CHECK_ARG(q.w && q.g && q.n > 0, "y3_clip_update_multi: tensor %d has a null pointer or no elements", i);
        Y3_CHECK_ARG(kind == 0 || q.slot0, "y3_clip_update_multi: tensor %d: optimizer slot missing", i);
        Y3_CHECK_ARG(kind < 2 || q.slot1, "y3_clip_update_multi: tensor %d: second optimizer slot missing", i);
        Y3_CHECK_ARG((((uintptr_t)q.w | (uintptr_t)q.g) & 15) == 0 || (q.n & 3) != 0,
                     "y3_clip_update_multi: tensor %d: w and g must be 16-byte aligned", i);
        host[i] = MtDesc{q.w, q.g, q.slot0, q.slot1, q.n, q.weight_decay, (int)chunks};
        chunks += mt_chunks(q.n);
        Y3_CHECK_ARG(chunks < (1LL << 30), "y3_clip_update_multi: too many elements");
    }
    char* base = static_cast<char*>(scratch);
    MtDesc* descs = reinterpret_cast<MtDesc*>(base);
    float* partial = reinterpret_cast<float*>(base + (((size_t)count * sizeof(MtDesc) + 255) & ~(size_t)255));
    float* norm = reinterpret_cast<float*>(reinterpret_cast<char*>(partial) + (((size_t)chunks * 4 + 255) & ~(size_t)255));
    hipStream_t st = ctx->stream;
    Y3_CHECK_HIP(hipMemcpyAsync(descs, host, (size_t)count * sizeof(MtDesc), hipMemcpyHostToDevice, st));
    {
        const int rc = y3_ctx_stage_release(ctx);
        if (rc != Y3_OK) return rc;
    }
    hipLaunchKernelGGL(mt_prepare_kernel, dim3((unsigned)chunks), dim3(256), 0, st, descs, count, grad_scale, partial);
    Y3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(mt_norms_kernel, dim3(count), dim3(256), 0, st, descs, count, (int)chunks, partial, norm);
    Y3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(mt_update_kernel, dim3((unsigned)chunks), dim3(256), 0, st, descs, count, norm, clip_norm, kind,
                       lr, momentum, decay, beta2, eps);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_pad_channels(y3_ctx* ctx, const float* src, int c_src, long long rows, int c_dst, float* dst) {
    Y3_CHECK_ARG(ctx && src && dst, "y3_pad_channels: null argument");
    Y3_CHECK_ARG(rows > 0 && c_src > 0 && c_dst >= c_src, "y3_pad_channels: bad shape");
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for(rows * c_dst)), dim3(256), 0, ctx->stream, src, c_src,
                       rows, c_dst, dst);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_box_iou(y3_ctx* ctx, const float* pred_boxes, long long num_pred, const float* true_boxes,
                          int num_true, float* iou) {
    Y3_CHECK_ARG(ctx && pred_boxes && true_boxes && iou, "y3_box_iou: null argument");
    Y3_CHECK_ARG(num_pred > 0 && num_true > 0, "y3_box_iou: empty input");
    hipLaunchKernelGGL(box_iou_kernel, dim3(grid_for(num_pred * num_true)), dim3(256), 0, ctx->stream, pred_boxes,
                       num_pred, true_boxes, num_true, iou);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_process_box(y3_ctx* ctx, const float* boxes, const int32_t* labels, const int32_t* counts, int n,
                              int kmax, int class_num, int img_w, int img_h, const float* anchors_host18,
                              float* y_true_13, float* y_true_26, float* y_true_52) {
    Y3_CHECK_ARG(ctx && boxes && labels && counts && anchors_host18 && y_true_13 && y_true_26 && y_true_52,
                 "y3_process_box: null argument");
    Y3_CHECK_ARG(n > 0 && kmax > 0 && class_num > 0, "y3_process_box: non-positive dimension");
    Y3_CHECK_ARG(img_w > 0 && img_h > 0 && img_w % 32 == 0 && img_h % 32 == 0,
                 "y3_process_box: image size must be a positive multiple of 32");
    TargetArgs a;
    a.boxes = boxes; a.labels = labels; a.counts = counts;
    a.y[0] = y_true_13; a.y[1] = y_true_26; a.y[2] = y_true_52;
    a.N = n; a.kmax = kmax; a.C = class_num; a.img_w = img_w; a.img_h = img_h;
    for (int k = 0; k < 9; ++k) { a.anc_w[k] = anchors_host18[2 * k]; a.anc_h[k] = anchors_host18[2 * k + 1]; }
    const int T = 6 + class_num;
    const int strides[3] = {32, 16, 8};
    for (int s = 0; s < 3; ++s) {
        const long long cells = (long long)n * (img_h / strides[s]) * (img_w / strides[s]) * 3;
        hipLaunchKernelGGL(target_fill_kernel, dim3(grid_for(cells * T)), dim3(256), 0, ctx->stream, a.y[s], cells, T);
        Y3_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(target_assign_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
