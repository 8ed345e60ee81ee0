// bf16 path (BASELINE configs[4]): the first TWO convs of Darknet-53 in one kernel - the 3x3 3->32 stem and the stride-2 3x3
// 32->64 conv behind it (utils/layer_utils.py:34-40: darknet53_body's `conv2d(inputs, 32, 3)` and `conv2d(net, 64, 3,
// strides=2)`, each with folded batch norm + LeakyReLU(0.1); the stride-2 conv pads one pixel on every side,
// utils/layer_utils.py:9-22 `_fixed_padding`).  Round 5.
//
// Why: at 608x608, bs=16 the stem's output is the largest tensor of the network (378 MB in bf16) and exists only to be read
// once by the next layer: stem 0.17 ms + stride-2 conv 0.17 ms for 413 + 567 MB of traffic, where the image is 35 MB and the
// second conv's output 189 MB.  Fused, the stem's pixels live in the LDS only.  Both convs run on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate); the stem, whose 27 x 32 multiply-adds per pixel bound it on the vector ALU
// before (csrc/y3_conv_bf16.hip: conv_stem_bf16_kernel), takes K = 27 padded to 32 with image and kernel each split into TWO
// bf16 parts (x = x_hi + x_lo exactly to 2^-16; three products x_hi w_hi + x_hi w_lo + x_lo w_hi): the stem keeps fp32-grade
// arithmetic on the fp32 image, as the separate kernel had - a plain bf16 stem cost 0.05 of the mAP of
// tests/test_bf16_gpu.py - for six MFMAs per 32 pixels instead of two.
//
// A persistent workgroup (eight waves, one per CU: 123 KB of LDS) walks 16 x 16 tiles of the second conv's output:
//   phase 1  the image patch of the tile (35 x 35 pixels, zeros outside the image; loaded into registers during the previous
//            tile's phase 3) goes to the LDS as TWO bf16 planes (hi, lo) of 4-channel pixels (the fourth channel is zero): each
//            image value is split once and gathered nine times;
//   phase 2  the 33 x 33 stem pixels the tile needs, 32 at a time, K ordered (ky | kx padded to 4 | channel padded to 4) = 3
//            steps of 16: weights are the A operand (rows = the 32 output channels), pixels the B operand - a lane's 8 k-values
//            of a step are two neighbouring 4-channel pixels = two 8-byte LDS reads per plane, no arithmetic - so that a lane
//            ends up with 4 CONSECUTIVE channels of ONE pixel per register quad: scale / shift / LeakyReLU, a zero for stem pixels outside the map (the second conv's padding), bf16, one 8-byte
//            LDS write per quad.  Stem pixels are stored by column parity - plane[x & 1][y][x >> 1][64 B] - because the stride-2
//            conv reads every other column: a tap's 16 neighbouring outputs then read 16 neighbouring rows, and the 16-byte
//            chunk index XOR (row >> 2) & 3 makes those ds_read_b128 conflict-free;
//   phase 3  the stride-2 conv as 9 taps x 2 slices of 16 channels: A = its weights (resident in the LDS for the whole
//            kernel, [tap][64 channels][32] bf16, same swizzle), B = stem pixels; 36 MFMAs per wave; a lane again holds
//            4 consecutive output channels of one output pixel per quad: scale / shift / LeakyReLU, bf16, staged through the
//            (now idle) stem-pixel LDS so that an output pixel's 128 bytes leave as eight 16-byte pieces of one line.
#include <cstdlib>
#include "y3_internal.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct StemS2Args {
    const float* x;       // [N,H,W,3] fp32 image
    const float* w0;      // stem kernel, HWIO fp32 [27][32]
    const float* scale0;  // [32]
    const float* shift0;  // [32]
    const bf16_t* w1;     // second conv, packed [9][64][32] bf16 (y3_pack_conv_weights_bf16 with k = 3, cin = 32)
    const float* scale1;  // [64]
    const float* shift1;  // [64]
    bf16_t* y;            // [N,H/2,W/2,64] bf16
    int N, H, W, act0, act1;
    int tiles_y, tiles_x, ntiles;
};

constexpr int TS = 16;                    // output tile side of the second conv
constexpr int SP = 2 * TS + 1;            // stem pixels per tile side (33)
constexpr int PP = SP + 2;                // image patch side (35)
constexpr int PPITCH = PP + 1;            // pixels per patch row in the LDS (the last one stays zero: kx = 3 of the padded K)
constexpr int PLANE_BYTES = PP * PPITCH * 8;          // one bf16 plane of 4-channel pixels: 10,080
constexpr int SIDX = TS + 1;              // columns per parity plane (17)
constexpr int S_BYTES = 2 * SP * SIDX * 64;          // 71,808
constexpr int W1_BYTES = 9 * 64 * 64;                // 36,864
constexpr int P_BYTES = (2 * PLANE_BYTES + 15) & ~15;
constexpr int CONST_BYTES = (32 + 32 + 64 + 64) * 4;
constexpr int LDS_BYTES = S_BYTES + W1_BYTES + P_BYTES + CONST_BYTES;
constexpr int NTHR = 512;
constexpr int NPRE = (PP * PP + NTHR - 1) / NTHR;          // image pixels each thread prefetches per tile (3)
constexpr int OPITCH = 144;                          // staged output row: 128 bytes + 16 (bank spread)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> packed bf16 pair `hi` (round to nearest even) and the packed bf16 pair `lo` of what is left: a = a_hi + a_lo to 2^-16
__device__ __forceinline__ void split_bf16(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack_bf16(a, b);
    lo = pack_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}

__global__ void __launch_bounds__(NTHR) conv_stem_s2_bf16_kernel(const StemS2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* S = smem;                                   // stem pixels: [2 planes][33 rows][17][64 B]
    unsigned char* W1 = smem + S_BYTES;                        // [9][64][64 B], chunk ^ ((cout >> 2) & 3)
    unsigned char* P = smem + S_BYTES + W1_BYTES;             // image patch: planes hi, lo of [35][36] pixels x 4 bf16
    float* C = reinterpret_cast<float*>(smem + S_BYTES + W1_BYTES + P_BYTES); // scale0[32] shift0[32] scale1[64] shift1[64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int OH = p.H >> 1, OW = p.W >> 1;

    // ---- once per workgroup: the second conv's weights and the constants into the LDS, the stem's weights into registers ----
    for (int c = tid; c < 9 * 64 * 4; c += NTHR) {             // 16-byte chunks
        const int row = c >> 2, ch = c & 3;                    // row = tap * 64 + cout
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.w1 + (size_t)row * 32 + ch * 8);
        *reinterpret_cast<u32x4*>(W1 + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)) = v;
    }
    if (tid < 32) { C[tid] = p.scale0[tid]; C[32 + tid] = p.shift0[tid]; }
    if (tid < 64) { C[64 + tid] = p.scale1[tid]; C[128 + tid] = p.shift1[tid]; }
    for (int c = tid; c < 2 * PLANE_BYTES / 8; c += NTHR) *reinterpret_cast<u32x2*>(P + c * 8) = u32x2{0u, 0u};
    // stem weights as the A operand: lane (row = channel l32, half h) holds, for step s = ky, k = 8 h + j -> (kx = 2 h + j / 4,
    // channel j % 4), zero where kx = 3 or the channel is 3; as two bf16 parts
    bf16x8 wfrag[3], wfrag_lo[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kx = 2 * h + (j >> 2), c = j & 3;
            wv[j] = (kx < 3 && c < 3) ? p.w0[((s * 3 + kx) * 3 + c) * 32 + l32] : 0.f;
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf16(wv[2 * j], wv[2 * j + 1], hi[j], lo[j]);
        wfrag[s] = __builtin_bit_cast(bf16x8, (u32x4{hi[0], hi[1], hi[2], hi[3]}));
        wfrag_lo[s] = __builtin_bit_cast(bf16x8, (u32x4{lo[0], lo[1], lo[2], lo[3]}));
    }
    // the stem's scale / shift of this lane's channels (quads 8 g + 4 h .. + 3)
    f32x4 sc0[4], sh0[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sc0[g] = *reinterpret_cast<const f32x4*>(p.scale0 + 8 * g + 4 * h);
        sh0[g] = *reinterpret_cast<const f32x4*>(p.shift0 + 8 * g + 4 * h);
    }

    // this workgroup's tiles: w, w + G, ... (equal cost per tile)
    const int G = gridDim.x;
    int tile = blockIdx.x;
    float pre[NPRE][3];
    int prow[NPRE], pcol[NPRE];                      // this thread's patch pixels (the same for every tile; row < 0: none)
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int e = tid + NTHR * j;
        prow[j] = e < PP * PP ? e / PP : -1000000;
        pcol[j] = e - (e / PP) * PP;
    }
    auto prefetch = [&](int t) {
        const int tpi = p.tiles_y * p.tiles_x;
        const int n = t / tpi, r = t - n * tpi;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int iy0 = 2 * ty * TS - 2, ix0 = 2 * tx * TS - 2;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int iy = iy0 + prow[j], ix = ix0 + pcol[j];
            const bool ok = t < p.ntiles && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = p.x + ((size_t)(n * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) pre[j][c] = ok ? src[c] : 0.f;
        }
    };
    prefetch(tile);
    __syncthreads();

    for (; tile < p.ntiles; tile += G) {
        const int tpi = p.tiles_y * p.tiles_x;
        const int n = tile / tpi, r = tile - n * tpi;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int oy0 = ty * TS, ox0 = tx * TS;

        // ---- phase 1: the prefetched image patch into the LDS, split into two bf16 planes of 4-channel pixels -----------------
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            if (prow[j] >= 0) {
                unsigned h01, l01, h2, l2;
                split_bf16(pre[j][0], pre[j][1], h01, l01);
                split_bf16(pre[j][2], 0.f, h2, l2);
                unsigned char* dstp = P + (prow[j] * PPITCH + pcol[j]) * 8;
                *reinterpret_cast<u32x2*>(dstp) = u32x2{h01, h2};
                *reinterpret_cast<u32x2*>(dstp + PLANE_BYTES) = u32x2{l01, l2};
            }
        }
        __syncthreads();

        // ---- phase 2: the 33 x 33 stem pixels of the tile, 32 per MFMA pair, px-tiles wave, wave + 4, ... ------------------
        for (int pt = wave; pt * 32 < SP * SP; pt += 8) {
            const int pix = pt * 32 + l32;
            const int pc = pix < SP * SP ? pix : SP * SP - 1;          // (the last, partial px-tile: clamped, never written)
            const int py = pc / SP, px = pc - py * SP;
            const unsigned char* base = P + (py * PPITCH + px + 2 * h) * 8;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const u32x2 h0 = *reinterpret_cast<const u32x2*>(base + s * (PPITCH * 8));
                const u32x2 h1 = *reinterpret_cast<const u32x2*>(base + s * (PPITCH * 8) + 8);
                const u32x2 l0 = *reinterpret_cast<const u32x2*>(base + s * (PPITCH * 8) + PLANE_BYTES);
                const u32x2 l1 = *reinterpret_cast<const u32x2*>(base + s * (PPITCH * 8) + PLANE_BYTES + 8);
                const bf16x8 xh = __builtin_bit_cast(bf16x8, (u32x4{h0[0], h0[1], h1[0], h1[1]}));
                const bf16x8 xl = __builtin_bit_cast(bf16x8, (u32x4{l0[0], l0[1], l1[0], l1[1]}));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag_lo[s], xh, acc, 0, 0, 0);     // (small terms first)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[s], xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[s], xh, acc, 0, 0, 0);
            }
            // D: column = pixel (this lane), rows = channels (q & 3) + 8 * (q >> 2) + 4 * h
            const int sy = 2 * oy0 - 1 + py, sx = 2 * ox0 - 1 + px;
            const bool inside = (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;   // else: the second conv's padding
            const int idx = px >> 1;
            unsigned char* dst = S + (((px & 1) * SP + py) * SIDX + idx) * 64 + h * 8;
            const int sw = (idx >> 2) & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[4 * g + e] * sc0[g][e] + sh0[g][e];
                    o[e] = p.act0 ? fmaxf(t, 0.1f * t) : t;
                }
                const unsigned lo = pack_bf16(o[0], o[1]), hi = pack_bf16(o[2], o[3]);
                if (pix < SP * SP)
                    *reinterpret_cast<u32x2*>(dst + ((g ^ sw) << 4)) = u32x2{inside ? lo : 0u, inside ? hi : 0u};
            }
        }
        __syncthreads();

        // the next tile's image patch: in flight under phase 3
        prefetch(tile + G);

        // ---- phase 3: the stride-2 conv; rows = 64 output channels (2 MFMA tiles), columns = 256 outputs (32 per wave) -------------
        f32x16 acc3[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc3[rt][q] = 0.f;
        const int m = wave * 32 + l32;                         // output pixel of the tile, row-major 16 x 16
        const int boff = (m >> 4) * 2 * SIDX + (m & 15);       // (row 2 * oyl, column oxl) of a parity plane, in pixels
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 a[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int row = tap * 64 + rt * 32 + l32;
                    a[rt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                W1 + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4)));
                }
                const int idx = (m & 15) + (kx >> 1);
                const int pixel = ((kx & 1) * SP + ky) * SIDX + boff + (kx >> 1);
                const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                     S + pixel * 64 + (((2 * s + h) ^ ((idx >> 2) & 3)) << 4)));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    acc3[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt], b, acc3[rt], 0, 0, 0);
            }
        }
        __syncthreads();          // every wave is done with the stem pixels: their LDS stages the output
        // epilogue: a lane holds, per rt, 16 channels of ONE output pixel (quads of 4 consecutive channels) -> the wave's patch
        // [32 pixels][128 B] -> eight 16-byte pieces per pixel, 8 lanes per 128-byte line
        unsigned char* out = S + wave * (32 * OPITCH);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * rt + 8 * g + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(C + 64 + ch);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(C + 128 + ch);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc3[rt][4 * g + e] * sc[e] + sh[e];
                    o[e] = p.act1 ? fmaxf(t, 0.1f * t) : t;
                }
                *reinterpret_cast<u32x2*>(out + l32 * OPITCH + ch * 2) = u32x2{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            }
        // (one wave: its LDS operations execute in order)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = i * 8 + (lane >> 3), piece = lane & 7;      // pixel of the wave's 32, 16-byte piece of its row
            const int mm = wave * 32 + pl;
            const int oy = oy0 + (mm >> 4), ox = ox0 + (mm & 15);
            const u32x4 v = *reinterpret_cast<const u32x4*>(out + pl * OPITCH + piece * 16);
            if (oy < OH && ox < OW)
                *reinterpret_cast<u32x4*>(p.y + ((size_t)(n * OH + oy) * OW + ox) * 64 + piece * 8) = v;
        }
        __syncthreads();          // the next tile's phases 1 / 2 overwrite the patch and the stem pixels
    }
}

}  // namespace

// 1 if the fused kernel takes this pair of layers: the 3x3 stride-1 3 -> 32 stem followed by a 3x3 stride-2 32 -> 64 conv
int y3_conv_bf16_stem_s2_takes(const y3_conv_desc* d0, const y3_conv_desc* d1) {
    if (!d0 || !d1) return 0;
    static int off = -1;
    if (off < 0) {
        const char* e = y3_exp_env("Y3_BF16_FUSED_STEM");
        off = (e && e[0] == '0') ? 1 : 0;
    }
    if (off) return 0;
    if (y3_device_max_lds() < (size_t)LDS_BYTES) return 0;      // (the device must offer the kernel's LDS: the plan then runs the layers unfused)
    return d0->k == 3 && d0->stride == 1 && d0->cin == 3 && d0->cout == 32 && d0->c_up == 0 &&
           d1->k == 3 && d1->stride == 2 && d1->cin == 32 && d1->cout == 64 && d1->c_up == 0 &&
           d1->n == d0->n && d1->h == d0->h && d1->w == d0->w && d0->h % 2 == 0 && d0->w % 2 == 0;
}

int y3_launch_conv_bf16_stem_s2(hipStream_t stream, int n, int h, int w, const float* x, const float* w0, const float* scale0,
                                const float* shift0, int act0, const void* w1_packed, const float* scale1, const float* shift1,
                                int act1, void* y) {
    Y3_CHECK_ARG(x && w0 && scale0 && shift0 && w1_packed && scale1 && shift1 && y, "y3_conv2d_fwd_bf16_stem_s2: null pointer argument");
    Y3_CHECK_ARG(n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "y3_conv2d_fwd_bf16_stem_s2: the image sides must be even");
    Y3_CHECK_ARG((long long)n * h * w * 16 < (1LL << 31), "y3_conv2d_fwd_bf16_stem_s2: tensor too large for 32-bit pixel indices");
    StemS2Args a;
    a.x = x; a.w0 = w0; a.scale0 = scale0; a.shift0 = shift0; a.w1 = static_cast<const bf16_t*>(w1_packed);
    a.scale1 = scale1; a.shift1 = shift1; a.y = static_cast<bf16_t*>(y);
    a.N = n; a.H = h; a.W = w; a.act0 = act0; a.act1 = act1;
    a.tiles_y = (h / 2 + TS - 1) / TS; a.tiles_x = (w / 2 + TS - 1) / TS;
    a.ntiles = n * a.tiles_y * a.tiles_x;
    static bool attr_set[Y3_MAX_DEVICES] = {};     // benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_s2_bf16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(conv_stem_s2_bf16_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_conv2d_fwd_bf16_stem_s2(y3_ctx* ctx, int n, int h, int w, const float* x, const float* w0_hwio,
                                          const float* scale0, const float* shift0, const void* w1_packed,
                                          const float* scale1, const float* shift1, void* y) {
    Y3_CHECK_ARG(ctx, "y3_conv2d_fwd_bf16_stem_s2: null context");
    return y3_launch_conv_bf16_stem_s2(ctx->stream, n, h, w, x, w0_hwio, scale0, shift0, 1, w1_packed, scale1, shift1, 1, y);
}
