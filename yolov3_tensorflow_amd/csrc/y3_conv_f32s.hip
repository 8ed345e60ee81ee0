// fp32 path: the first TWO convs of Darknet-53 in one kernel - the 3x3 3->32 stem and the stride-2 3x3 32->64 conv behind it
// (utils/layer_utils.py:34-40: darknet53_body's `conv2d(inputs, 32, 3)` and `conv2d(net, 64, 3, strides=2)`, each with folded
// batch norm + LeakyReLU(0.1); the stride-2 conv pads one pixel on every side, utils/layer_utils.py:9-22).  Exact fp32
// arithmetic on v_mfma_f32_32x32x2_f32, like the separate kernels it replaces.  Round 5 (the fp32 twin of y3_conv_bf16s.hip).
//
// Why: at 416x416, bs=32 the stem writes 709 MB that only the next layer reads: stem 0.20 ms (memory-bound) + stride-2 conv
// 0.49 ms (matrix-pipe bound, its operand gathered from the L2 nine times over) of a 10.2 ms forward.  Fused, the stem's
// pixels exist only in the LDS, the second conv reads its operand from there, and the stem itself moves from the vector ALU
// (27 x 32 multiply-adds per pixel) to the matrix pipe (K = 27 padded to 28: 14 MFMAs per 32 pixels).
//
// A persistent workgroup (eight waves, one per CU: 153 KB of LDS) walks 8 x 16 tiles of the second conv's output:
//   phase 1  the image patch of the tile (19 x 35 pixels x 3, zeros outside the image; loaded into registers during the
//            previous tile's phase 3) goes to the LDS;
//   phase 2  the 17 x 33 stem pixels the tile needs, 32 at a time: weights are the A operand (rows = the 32 output channels,
//            in registers), pixels the B operand - MFMA j takes k = 2 j + (lane / 32), (tap, channel) = (k / 3, k % 3), one
//            4-byte LDS read through a per-lane offset table - so that a lane ends up with 4 consecutive channels of ONE
//            pixel per register quad: scale / shift / LeakyReLU, zero for stem pixels outside the map (the second conv's
//            padding), one 16-byte LDS write per quad.  Stem pixels are stored by column parity - plane[x & 1][y][x >> 1][128 B]
//            - because the stride-2 conv reads every other column, with the 16-byte chunk index XOR (x >> 2) & 7
//            (conflict-free ds_read_b128 for 16 neighbouring outputs);
//   phase 3  the stride-2 conv: A = its weights (resident in the LDS for the whole kernel, [tap][64][32] fp32), B = stem
//            pixels, one ds_read_b128 per operand and FOUR MFMAs (lane half h holds channels 4 (2 kk + h) + q of both
//            operands); a wave owns 32 outputs x 32 channels = 144 MFMAs per tile.  Scale / shift / LeakyReLU, then through a
//            wave-private patch of the (now idle) stem-pixel LDS so that an output pixel's 128-byte half row leaves as
//            eight 16-byte pieces.
#include <cstdlib>
#include "y3_internal.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct StemS2F32Args {
    const float* x;       // [N,H,W,3] image
    const float* w0;      // stem kernel, HWIO [27][32]
    const float* scale0;  // [32]
    const float* shift0;  // [32]
    const float* w1;      // second conv, packed [9][64][32] (y3_pack_conv_weights with k = 3: [tap][cout][cin])
    const float* scale1;  // [64]
    const float* shift1;  // [64]
    float* y;             // [N,H/2,W/2,64]
    int N, H, W, act0, act1;
    int tiles_y, tiles_x, ntiles;
};

constexpr int TSY = 8, TSX = 16;                      // output tile of the second conv
constexpr int SPY = 2 * TSY + 1, SPX = 2 * TSX + 1;   // stem pixels per tile: 17 x 33
constexpr int NSP = SPY * SPX;                        // 561
constexpr int PPY = SPY + 2, PPX = SPX + 2;           // image patch: 19 x 35
constexpr int PATCH_FLOATS = PPY * PPX * 3;           // 1,995 (+ one zero word behind it)
constexpr int SIDX = TSX + 1;                         // columns per parity plane (17)
constexpr int S_BYTES = 2 * SPY * SIDX * 128;         // 73,984
constexpr int W1_BYTES = 9 * 64 * 128;                // 73,728
constexpr int P_BYTES = ((PATCH_FLOATS + 1) * 4 + 15) & ~15;
constexpr int LDS_BYTES = S_BYTES + W1_BYTES + P_BYTES;
constexpr int NTHR = 512, NW = 8;
constexpr int NPRE = (PPY * PPX + NTHR - 1) / NTHR;   // image pixels each thread prefetches per tile (2)
constexpr int OPITCH = 144;                           // staged half row of an output pixel: 128 bytes + 16 (bank spread)

__global__ void __launch_bounds__(NTHR) conv_stem_s2_f32_kernel(const StemS2F32Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* S = smem;                                   // stem pixels: [2 planes][17 rows][17][128 B]
    unsigned char* W1 = smem + S_BYTES;                        // [9][64][128 B], chunk ^ ((cout >> 1) & 7)
    float* P = reinterpret_cast<float*>(smem + S_BYTES + W1_BYTES);          // image patch [19][35][3], then one zero

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int OH = p.H >> 1, OW = p.W >> 1;

    // ---- once per workgroup: the second conv's weights into the LDS, everything small into registers -------------------------
    for (int c = tid; c < 9 * 64 * 8; c += NTHR) {             // 16-byte chunks
        const int row = c >> 3, ch = c & 7;                    // row = tap * 64 + cout
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.w1 + (size_t)row * 32 + ch * 4);
        *reinterpret_cast<u32x4*>(W1 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
    }
    if (tid == 0) P[PATCH_FLOATS] = 0.f;
    // stem weights as the A operand: MFMA j, lane (row = channel l32, half h): k = 2 j + h
    float wst[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int k = 2 * j + h;
        wst[j] = k < 27 ? p.w0[k * 32 + l32] : 0.f;
    }
    // gather table of the B operand: byte offset of k = 2 j + h inside a pixel's 3 x 3 x 3 window: k + 96 * (k / 9) floats (patch
    // rows are 105 floats apart); k = 27 reads the zero word behind the patch
    int koff[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int k = 2 * j + h;
        koff[j] = k < 27 ? (k + 96 * (k / 9)) * 4 : -1;
    }
    // scale / shift of this lane's channels: stem quads 8 g + 4 h; second conv quads 32 rt + 8 g + 4 h (rt = wave / 4)
    const int rt = wave >> 2, ct = wave & 3;
    f32x4 sc0[4], sh0[4], sc1[4], sh1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sc0[g] = *reinterpret_cast<const f32x4*>(p.scale0 + 8 * g + 4 * h);
        sh0[g] = *reinterpret_cast<const f32x4*>(p.shift0 + 8 * g + 4 * h);
        sc1[g] = *reinterpret_cast<const f32x4*>(p.scale1 + 32 * rt + 8 * g + 4 * h);
        sh1[g] = *reinterpret_cast<const f32x4*>(p.shift1 + 32 * rt + 8 * g + 4 * h);
    }

    // this workgroup's tiles: w, w + G, ... (equal cost per tile)
    const int G = gridDim.x;
    int tile = blockIdx.x;
    float pre[NPRE][3];
    int prow[NPRE], pcol[NPRE];                      // this thread's patch pixels (the same for every tile; row < 0: none)
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int e = tid + NTHR * j;
        prow[j] = e < PPY * PPX ? e / PPX : -1000000;
        pcol[j] = e - (e / PPX) * PPX;
    }
    auto prefetch = [&](int t) {
        const int tpi = p.tiles_y * p.tiles_x;
        const int n = t / tpi, r = t - n * tpi;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int iy0 = 2 * ty * TSY - 2, ix0 = 2 * tx * TSX - 2;
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
        const int oy0 = ty * TSY, ox0 = tx * TSX;

        // ---- phase 1: the prefetched image patch into the LDS --------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (prow[j] >= 0) {
                float* d = P + (prow[j] * PPX + pcol[j]) * 3;
                d[0] = pre[j][0]; d[1] = pre[j][1]; d[2] = pre[j][2];
            }
        __syncthreads();

        // ---- phase 2: the 17 x 33 stem pixels of the tile, 32 per group of 14 MFMAs, px-tiles wave, wave + 8, ... ---------------
        for (int pt = wave; pt * 32 < NSP; pt += NW) {
            const int pix = pt * 32 + l32;
            const int pc = pix < NSP ? pix : NSP - 1;                  // (the last, partial px-tile: clamped, never written)
            const int py = pc / SPX, px = pc - py * SPX;
            const unsigned char* base = reinterpret_cast<const unsigned char*>(P) + (py * PPX + px) * 12;
            const unsigned char* zero = reinterpret_cast<const unsigned char*>(P + PATCH_FLOATS);
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                const float b = *reinterpret_cast<const float*>(koff[j] >= 0 ? base + koff[j] : zero);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wst[j], b, acc, 0, 0, 0);
            }
            // D: column = pixel (this lane), rows = channels (q & 3) + 8 * (q >> 2) + 4 * h
            const int sy = 2 * oy0 - 1 + py, sx = 2 * ox0 - 1 + px;
            const bool inside = (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;   // else: the second conv's padding
            const int idx = px >> 1;
            unsigned char* dst = S + (((px & 1) * SPY + py) * SIDX + idx) * 128;
            const int sw = (idx >> 1) & 7;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[4 * g + e] * sc0[g][e] + sh0[g][e];
                    const float a = p.act0 ? fmaxf(t, 0.1f * t) : t;
                    o[e] = inside ? a : 0.f;
                }
                if (pix < NSP) *reinterpret_cast<f32x4*>(dst + (((2 * g + h) ^ sw) << 4)) = o;
            }
        }
        __syncthreads();

        // the next tile's image patch: in flight under phase 3
        prefetch(tile + G);

        // ---- phase 3: the stride-2 conv; this wave: output pixels 32 ct .. + 31 of the tile, channels 32 rt .. + 31 ---------------
        f32x16 acc3;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc3[q] = 0.f;
        const int m = ct * 32 + l32;                           // output pixel of the tile, row-major 8 x 16
        const int oyl = m >> 4, oxl = m & 15;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int row = tap * 64 + rt * 32 + l32;
            const unsigned char* ap = W1 + row * 128;
            const int asw = (row >> 1) & 7;
            const int idx = oxl + (kx >> 1);
            const unsigned char* bp = S + (((kx & 1) * SPY + 2 * oyl + ky) * SIDX + idx) * 128;
            const int bsw = (idx >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ap + (((2 * kk + h) ^ asw) << 4));
                const f32x4 b = *reinterpret_cast<const f32x4*>(bp + (((2 * kk + h) ^ bsw) << 4));
#pragma unroll
                for (int q = 0; q < 4; ++q) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc3, 0, 0, 0);
            }
        }
        __syncthreads();          // every wave is done with the stem pixels: their LDS stages the output
        // epilogue: a lane holds 16 channels of ONE output pixel (quads of 4 consecutive channels) -> the wave's patch
        // [32 pixels][128 B] -> eight 16-byte pieces per pixel's half row
        unsigned char* out = S + wave * (32 * OPITCH);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = acc3[4 * g + e] * sc1[g][e] + sh1[g][e];
                o[e] = p.act1 ? fmaxf(t, 0.1f * t) : t;
            }
            *reinterpret_cast<f32x4*>(out + l32 * OPITCH + (8 * g + 4 * h) * 4) = o;
        }
        // (one wave: its LDS operations execute in order)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = i * 8 + (lane >> 3), piece = lane & 7;      // pixel of the wave's 32, 16-byte piece of its half row
            const int mm = ct * 32 + pl;
            const int oy = oy0 + (mm >> 4), ox = ox0 + (mm & 15);
            const f32x4 v = *reinterpret_cast<const f32x4*>(out + pl * OPITCH + piece * 16);
            if (oy < OH && ox < OW)
                *reinterpret_cast<f32x4*>(p.y + ((size_t)(n * OH + oy) * OW + ox) * 64 + rt * 32 + piece * 4) = v;
        }
        __syncthreads();          // the next tile's phases 1 / 2 overwrite the patch and the stem pixels
    }
}

}  // namespace

// 1 if the fused kernel takes this pair of layers: the 3x3 stride-1 3 -> 32 stem followed by a 3x3 stride-2 32 -> 64 conv
int y3_conv_f32_stem_s2_takes(const y3_conv_desc* d0, const y3_conv_desc* d1) {
    if (!d0 || !d1) return 0;
    static int off = -1;
    if (off < 0) {
        const char* e = y3_exp_env("Y3_F32_FUSED_STEM");
        off = (e && e[0] == '0') ? 1 : 0;
    }
    if (off) return 0;
    if (y3_device_max_lds() < (size_t)LDS_BYTES) return 0;      // (the device must offer the kernel's LDS: the plan then runs the layers unfused)
    return d0->k == 3 && d0->stride == 1 && d0->cin == 3 && d0->cout == 32 && d0->c_up == 0 &&
           d1->k == 3 && d1->stride == 2 && d1->cin == 32 && d1->cout == 64 && d1->c_up == 0 &&
           d1->n == d0->n && d1->h == d0->h && d1->w == d0->w && d0->h % 2 == 0 && d0->w % 2 == 0;
}

int y3_launch_conv_f32_stem_s2(hipStream_t stream, int n, int h, int w, const float* x, const float* w0, const float* scale0,
                               const float* shift0, int act0, const float* w1_packed, const float* scale1, const float* shift1,
                               int act1, float* y) {
    Y3_CHECK_ARG(x && w0 && scale0 && shift0 && w1_packed && scale1 && shift1 && y, "y3_conv2d_fwd_stem_s2: null pointer argument");
    Y3_CHECK_ARG(n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "y3_conv2d_fwd_stem_s2: the image sides must be even");
    Y3_CHECK_ARG((long long)n * h * w * 16 < (1LL << 31), "y3_conv2d_fwd_stem_s2: tensor too large for 32-bit pixel indices");
    StemS2F32Args a;
    a.x = x; a.w0 = w0; a.scale0 = scale0; a.shift0 = shift0; a.w1 = w1_packed; a.scale1 = scale1; a.shift1 = shift1; a.y = y;
    a.N = n; a.H = h; a.W = w; a.act0 = act0; a.act1 = act1;
    a.tiles_y = (h / 2 + TSY - 1) / TSY; a.tiles_x = (w / 2 + TSX - 1) / TSX;
    a.ntiles = n * a.tiles_y * a.tiles_x;
    static bool attr_set[Y3_MAX_DEVICES] = {};     // benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_s2_f32_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(conv_stem_s2_f32_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_conv2d_fwd_stem_s2(y3_ctx* ctx, int n, int h, int w, const float* x, const float* w0_hwio, const float* scale0,
                                     const float* shift0, const float* w1_packed, const float* scale1, const float* shift1,
                                     float* y) {
    Y3_CHECK_ARG(ctx, "y3_conv2d_fwd_stem_s2: null context");
    return y3_launch_conv_f32_stem_s2(ctx->stream, n, h, w, x, w0_hwio, scale0, shift0, 1, w1_packed, scale1, shift1, 1, y);
}
