// bf16 path (BASELINE configs[4]): the first residual block of Darknet-53 in one kernel - res_block(net, 32) on the 64-channel
// map behind the first stride-2 conv (utils/layer_utils.py:25-32: shortcut = x; conv2d(x, 32, 1); conv2d(., 64, 3); + shortcut,
// each conv with folded batch norm + LeakyReLU(0.1)).  Round 5.
//
// Why: at 608x608, bs=16 this block moves 662 MB for 1.2 GFLOP per image - its 1x1 conv reads the 189 MB map and writes
// 95 MB, its 3x3 conv reads those 95 MB and the 189 MB shortcut and writes 189 MB: 0.067 + 0.139 ms.  Fused, the map is read
// ONCE (it is the 1x1 conv's input and the shortcut) and the 32-channel tensor between the convs never leaves the LDS:
// 378 MB.
//
// A persistent workgroup (eight waves, one per CU) walks 16 x 16 output tiles:
//   phase 1  the 18 x 18 x 64-channel patch of the input around the tile arrives by LDS-DMA (`buffer_load ... lds`, issued
//            during the previous tile's phase 3 into the other of two patch buffers; pixels outside the map are out-of-range
//            offsets = zeros; the XOR swizzle that makes the fragment reads conflict-free sits on the source address);
//   phase 2  the 1x1 conv on the 324 patch pixels, 32 at a time: A = its weights (rows = 32 output channels, held in
//            registers), B = patch pixels (one ds_read_b128 per 16 channels), so that a lane ends up with 4 consecutive
//            channels of one pixel per register quad: scale / shift / LeakyReLU, ZERO for pixels outside the map (they are the
//            3x3 conv's padding), bf16, one 8-byte LDS write per quad ([18][18] pixels x 64 B, chunk ^ (column >> 2) & 3);
//   phase 3  the 3x3 conv as 9 taps x 2 slices of 16 channels: A = its weights (resident in the LDS, [tap][64][32] bf16), B =
//            the pixels of phase 2; the shortcut is read from the patch (the tile's own pixels) BEFORE the patch buffer is
//            reused to stage the output, so that an output pixel's 128 bytes leave as eight 16-byte pieces of one line.
#include <cstdlib>
#include "y3_internal.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_base, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_base, 16, voff, soff, 0, 0);
#endif
}

struct ResBlockArgs {
    const bf16_t* x;      // [N,H,W,64] bf16: input and shortcut
    const bf16_t* w2;     // 1x1 conv, packed [32][64] bf16 (y3_pack_conv_weights_bf16 with k = 1, cin = 64: [cin/32][cout][32])
    const float* scale2;  // [32]
    const float* shift2;  // [32]
    const bf16_t* w3;     // 3x3 conv, packed [9][64][32] bf16 (k = 3, cin = 32)
    const float* scale3;  // [64]
    const float* shift3;  // [64]
    bf16_t* y;            // [N,H,W,64] bf16
    int N, H, W, act2, act3;
    int tiles_y, tiles_x, ntiles;
};

constexpr unsigned OOB = 0x80000000u;
constexpr int TS = 16;                          // output tile side
constexpr int PS = TS + 2;                      // patch side (18)
constexpr int NPIX = PS * PS;                   // 324
constexpr int NDMA = (NPIX + 7) / 8;            // DMA instructions per patch (8 pixels x 128 B each): 41
constexpr int A_BYTES = NDMA * 1024;            // one patch buffer: 41,984
constexpr int M_BYTES = NPIX * 64;              // the 32-channel pixels between the convs: 20,736
constexpr int W3_BYTES = 9 * 64 * 64;           // 36,864
constexpr int CONST_BYTES = (32 + 32 + 64 + 64) * 4;
constexpr int LDS_BYTES = 2 * A_BYTES + M_BYTES + W3_BYTES + CONST_BYTES;
constexpr int NTHR = 512, NW = 8;
constexpr int DPW = (NDMA + NW - 1) / NW;       // DMA instructions per wave and tile (6; the last ones of some waves are dead)
constexpr int OPITCH = 144;                     // staged output row: 128 bytes + 16 (bank spread)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}

__global__ void __launch_bounds__(NTHR) conv_resblock64_bf16_kernel(const ResBlockArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* A = smem;                                   // [2] patch buffers: [324 pixels][128 B], slot = chunk ^ (pixel >> 1) & 7
    unsigned char* Mid = smem + 2 * A_BYTES;                   // [18][18] pixels x 64 B, chunk ^ ((column >> 2) & 3)
    unsigned char* W3 = Mid + M_BYTES;                         // [9][64][64 B], chunk ^ ((cout >> 2) & 3)
    float* C = reinterpret_cast<float*>(W3 + W3_BYTES);        // scale2[32] shift2[32] scale3[64] shift3[64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * 128), 0x00020000);

    // ---- once per workgroup -------------------------------------------------------------------------------------------------
    for (int c = tid; c < 9 * 64 * 4; c += NTHR) {
        const int row = c >> 2, ch = c & 3;                    // row = tap * 64 + cout
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.w3 + (size_t)row * 32 + ch * 8);
        *reinterpret_cast<u32x4*>(W3 + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)) = v;
    }
    if (tid < 32) { C[tid] = p.scale2[tid]; C[32 + tid] = p.shift2[tid]; }
    if (tid < 64) { C[64 + tid] = p.scale3[tid]; C[128 + tid] = p.shift3[tid]; }
    // the 1x1 conv's weights as the A operand: lane (row = output channel l32, half h), step s: input channels 16 s + 8 h ..
    // + 7 = 16 contiguous bytes of the packing [cin / 32][32 cout][32]
    bf16x8 wfrag[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ci = 16 * s + 8 * h;
        wfrag[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p.w2 + ((ci >> 5) * 32 + l32) * 32 + (ci & 31)));
    }
    f32x4 sc2[4], sh2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        sc2[g] = *reinterpret_cast<const f32x4*>(p.scale2 + 8 * g + 4 * h);
        sh2[g] = *reinterpret_cast<const f32x4*>(p.shift2 + 8 * g + 4 * h);
    }

    // DMA of a tile's patch: instruction q = wave + 8 j fills pixels 8 q .. 8 q + 7 (lane -> pixel 8 q + lane / 8, slot lane % 8)
    const int G = gridDim.x;
    auto issue_patch = [&](int t, int buf) {
        const int tpi = p.tiles_y * p.tiles_x;
        const int n = t / tpi, r = t - n * tpi;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int y0 = ty * TS - 1, x0 = tx * TS - 1;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int q = wave + NW * j;
            const int pix = q * 8 + (lane >> 3);
            const int py = pix / PS, px = pix - py * PS;
            const int iy = y0 + py, ix = x0 + px;
            const bool ok = t < p.ntiles && pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int chunk = (lane & 7) ^ ((pix >> 1) & 7);
            const unsigned voff = ok ? (unsigned)(((n * p.H + iy) * p.W + ix) * 128 + chunk * 16) : OOB;
            if (q < NDMA) dma16(rs_x, A + buf * A_BYTES + q * 1024, voff, 0);
        }
    };

    int tile = blockIdx.x, buf = 0;
    issue_patch(tile, 0);

    for (; tile < p.ntiles; tile += G, buf ^= 1) {
        const int tpi = p.tiles_y * p.tiles_x;
        const int n = tile / tpi, r = tile - n * tpi;
        const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
        const int oy0 = ty * TS, ox0 = tx * TS;
        const unsigned char* Ab = A + buf * A_BYTES;

        // ---- phase 1: this tile's patch has landed (and the previous tile's stores have left) ------------------------------------
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        __syncthreads();

        // ---- phase 2: the 1x1 conv on the patch pixels, px-tiles wave, wave + 8 ------------------------------------------------
        for (int pt = wave; pt * 32 < NPIX; pt += NW) {
            const int pix = pt * 32 + l32;
            const int pc = pix < NPIX ? pix : NPIX - 1;                // (the last, partial px-tile: clamped, never written)
            const unsigned char* src = Ab + pc * 128;
            const int sw = (pc >> 1) & 7;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + (((2 * s + h) ^ sw) << 4)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[s], b, acc, 0, 0, 0);
            }
            const int py = pc / PS, px = pc - py * PS;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;     // else: the 3x3 conv's padding
            unsigned char* dst = Mid + pc * 64 + h * 8;
            const int msw = (px >> 2) & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[4 * g + e] * sc2[g][e] + sh2[g][e];
                    o[e] = p.act2 ? fmaxf(t, 0.1f * t) : t;
                }
                const unsigned lo = pack_bf16(o[0], o[1]), hi = pack_bf16(o[2], o[3]);
                if (pix < NPIX)
                    *reinterpret_cast<u32x2*>(dst + ((g ^ msw) << 4)) = u32x2{inside ? lo : 0u, inside ? hi : 0u};
            }
        }
        __syncthreads();

        // the next tile's patch: in flight under phase 3, into the other buffer
        issue_patch(tile + G, buf ^ 1);

        // ---- phase 3: the 3x3 conv; rows = 64 output channels (2 MFMA tiles), columns = 256 outputs (32 per wave) ----------------
        f32x16 acc3[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc3[rt][q] = 0.f;
        const int m = wave * 32 + l32;                         // output pixel of the tile, row-major 16 x 16
        const int oyl = m >> 4, oxl = m & 15;
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
                                W3 + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4)));
                }
                const int col = oxl + kx;
                const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                     Mid + ((oyl + ky) * PS + col) * 64 + (((2 * s + h) ^ ((col >> 2) & 3)) << 4)));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    acc3[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt], b, acc3[rt], 0, 0, 0);
            }
        }
        // epilogue: a lane holds, per rt, 16 channels of ONE output pixel (quads of 4 consecutive channels); the shortcut = the
        // same channels of the patch's centre pixel (oyl + 1, oxl + 1)
        const int cpix = (oyl + 1) * PS + oxl + 1;
        const int csw = (cpix >> 1) & 7;
        u32x2 res[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * rt + 8 * g + 4 * h;                // chunk ch / 8, byte (ch % 8) * 2 inside it
                res[rt][g] = *reinterpret_cast<const u32x2*>(Ab + cpix * 128 + (((ch >> 3) ^ csw) << 4) + (ch & 7) * 2);
            }
        __syncthreads();          // every wave has its shortcut values: the patch buffer stages the output
        unsigned char* out = A + buf * A_BYTES + wave * (32 * OPITCH);
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
                    o[e] = p.act3 ? fmaxf(t, 0.1f * t) : t;
                }
                o[0] += __uint_as_float(res[rt][g][0] << 16);
                o[1] += __uint_as_float(res[rt][g][0] & 0xFFFF0000u);
                o[2] += __uint_as_float(res[rt][g][1] << 16);
                o[3] += __uint_as_float(res[rt][g][1] & 0xFFFF0000u);
                *reinterpret_cast<u32x2*>(out + l32 * OPITCH + ch * 2) = u32x2{pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
            }
        // (one wave: its LDS operations execute in order)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = i * 8 + (lane >> 3), piece = lane & 7;      // pixel of the wave's 32, 16-byte piece of its row
            const int mm = wave * 32 + pl;
            const int oy = oy0 + (mm >> 4), ox = ox0 + (mm & 15);
            const u32x4 v = *reinterpret_cast<const u32x4*>(out + pl * OPITCH + piece * 16);
            if (oy < p.H && ox < p.W)
                *reinterpret_cast<u32x4*>(p.y + ((size_t)(n * p.H + oy) * p.W + ox) * 64 + piece * 8) = v;
        }
        // (the next iteration's vmcnt(0) + barrier order these reads / stores before the buffer's next DMA two tiles on, and
        // the Mid pixels before the next phase 2)
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // (the last, dead DMAs write zeros: let them land)
}

}  // namespace

// 1 if the fused kernel takes this pair of layers: a 1x1 64 -> 32 conv followed by a 3x3 stride-1 32 -> 64 conv whose residual is
// the first conv's input
int y3_conv_bf16_resblock64_takes(const y3_conv_desc* d2, const y3_conv_desc* d3) {
    if (!d2 || !d3) return 0;
    static int off = -1;
    if (off < 0) {
        const char* e = y3_exp_env("Y3_BF16_FUSED_RESBLOCK");
        off = (e && e[0] == '0') ? 1 : 0;
    }
    if (off) return 0;
    // (the kernel reads the 1x1 weights in the [cin/32][cout][32] packing; an experiments build that sends the 64 -> 32 conv to
    // the ring kernel packs them [cin/64][cout][64]: the plan then runs the two layers unfused)
    if (y3_conv_bf16r_takes(1, 64)) return 0;
    if (y3_device_max_lds() < (size_t)LDS_BYTES) return 0;      // (the device must offer the kernel's LDS: the plan then runs the layers unfused)
    return d2->k == 1 && d2->stride == 1 && d2->cin == 64 && d2->cout == 32 && d2->c_up == 0 &&
           d3->k == 3 && d3->stride == 1 && d3->cin == 32 && d3->cout == 64 && d3->c_up == 0 &&
           d3->n == d2->n && d3->h == d2->h && d3->w == d2->w;
}

int y3_launch_conv_bf16_resblock64(hipStream_t stream, int n, int h, int w, const void* x, const void* w2_packed,
                                   const float* scale2, const float* shift2, int act2, const void* w3_packed, const float* scale3,
                                   const float* shift3, int act3, void* y) {
    Y3_CHECK_ARG(x && w2_packed && scale2 && shift2 && w3_packed && scale3 && shift3 && y,
                 "y3_resblock64_fwd_bf16: null pointer argument");
    Y3_CHECK_ARG(n > 0 && h > 0 && w > 0, "y3_resblock64_fwd_bf16: non-positive dimension");
    Y3_CHECK_ARG((long long)n * h * w * 128 < (1LL << 31), "y3_resblock64_fwd_bf16: tensor too large for 32-bit byte offsets");
    ResBlockArgs a;
    a.x = static_cast<const bf16_t*>(x); a.w2 = static_cast<const bf16_t*>(w2_packed); a.scale2 = scale2; a.shift2 = shift2;
    a.w3 = static_cast<const bf16_t*>(w3_packed); a.scale3 = scale3; a.shift3 = shift3; a.y = static_cast<bf16_t*>(y);
    a.N = n; a.H = h; a.W = w; a.act2 = act2; a.act3 = act3;
    a.tiles_y = (h + TS - 1) / TS; a.tiles_x = (w + TS - 1) / TS;
    a.ntiles = n * a.tiles_y * a.tiles_x;
    static bool attr_set[Y3_MAX_DEVICES] = {};     // benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_resblock64_bf16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(conv_resblock64_bf16_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_resblock64_fwd_bf16(y3_ctx* ctx, int n, int h, int w, const void* x, const void* w2_packed, const float* scale2,
                                      const float* shift2, const void* w3_packed, const float* scale3, const float* shift3,
                                      void* y) {
    Y3_CHECK_ARG(ctx, "y3_resblock64_fwd_bf16: null context");
    return y3_launch_conv_bf16_resblock64(ctx->stream, n, h, w, x, w2_packed, scale2, shift2, 1, w3_packed, scale3, shift3, 1, y);
}
