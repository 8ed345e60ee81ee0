// Winograd F(4x4, 3x3) form of the stride-1 3x3 conv (+ folded BN + LeakyReLU + residual; training: + batch-norm column
// sums, and the data gradient as the same conv on dz), fp32 arithmetic (products and sums in fp32; the transforms round, see
// the numerics line below) on v_mfma_f32_16x16x4_f32.  Replaces the same reference code as y3_conv.hip / y3_conv_wino.hip
// (utils/layer_utils.py:9-22,25-32; train.py:105-115 for the training uses): 36 multiplies per 4x4 output tile and channel
// pair instead of 144 (direct) or 64 (F(2x2,3x3)) - 1.78x less MFMA work than y3_conv_wino.hip.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 4x4 output tile (6x6 input patch), summed over input channels,
//   interpolation points 0, +-1, +-2, inf (Lavin & Gray).  Numerics on THIS network (tests/probes/winograd_numerics.py, fp32
//   restatement of the whole forward against the fp64 oracle): boxes 1.0e-5 of the box scale against 7.5e-6 for F(2x2,3x3)
//   and 6.3e-6 for the direct sum - a hundred times inside the 1e-3 of the north star.
//
// TWO forms, one arithmetic (same transforms, same K order, same output tail):
//
//   (1) two kernels (round 6; the one y3_net_forward / y3_net_train_* use wherever the caller hands in a workspace):
//       wino44_input_transform_kernel writes V = B^T d B ONCE per layer, laid out as the byte image of the LDS stages the second
//       kernel wants - [16-tile block][Cin / 16][2 K-steps][18 position pairs][4 lane quarters][16 tiles][2 positions][2 channels]
//       = 36 KB per (tile block, 16 channels) - and conv_wino44v_f32_kernel is then 36 batched GEMMs with nothing in its K-loop
//       but LDS-DMA of those images (linear 1 KB pieces), one ds_read_b128 and one 16-byte weight load per FOUR MFMAs, and
//       one barrier per 16 channels.  The fp32 forward uses 1.2 of the 8 TB/s of the HBM: the 2.25x larger V (100 MB for a
//       52-grid 128-channel layer of the bs=32 batch) buys back the issue slots the in-kernel transform took (rounds 4 / 5:
//       2.4 non-MFMA instructions per 32-cycle MFMA, matrix pipe 44 % busy).
//   (2) one kernel (rounds 3-5): raw 6x6 patches by LDS-DMA, B^T d B inside the K-loop.  Runs when the caller has no
//       workspace, and where policy says the extra pass does not pay (y3_conv_wino44_two_pass_impl).
//
//   * weights U = G g G^T transformed once at load time and packed [18 position pairs][Cin/8][Cout][4 channel pairs][2 positions]
//     [2 channels] (y3_pack_conv_weights_wino44): a lane's fragments of two neighbouring positions are 16 contiguous bytes;
//   * a workgroup = FOUR waves owns 16 tiles x 64 output channels for ALL 36 transform positions (72 KB of LDS, <= 256
//     registers per wave -> TWO workgroups per CU, one's prologue / store tail under the other's MFMAs); wave wn holds 16 tiles
//     x 16 channels of every position as v_mfma_f32_16x16x4_f32 accumulators (36 x 4 = 144 registers), so the 36 position sums
//     of one (tile, channel) sit in ONE lane and A^T M A needs no exchange between waves at all;
//   * lane quarter q = lane / 16 holds channels 2q, 2q+1 of both operands and MFMA m = 0, 1 consumes channel 2q + m - which of
//     the 8 channels plays "k" where is free as long as both operands agree;
//   * weight fragments - a lane's 16 bytes for two positions, 1 KB contiguous per wave load - straight from global memory (L2)
//     through a rolling window of loads that runs on across K-step boundaries; the order is pinned with scheduling barriers:
//     left alone, hipcc moves every fragment read right in front of its MFMAs;
//   * form (2) only: K-step = 8 input channels; thread (tile, channel pair, job) reads the patch rows its job needs from the
//     LDS, transforms them on float2s and writes one or two rows of B^T d B for the NEXT K-step (jobs: row 0 | rows 1,2 | rows
//     3,4 | row 5); V planes are [channel pair][tile][2 channels] with the tile index rotated by 4 per channel pair (v_off):
//     the fragment reads come out as ds_read2st64_b64, served in 16-lane groups over 32 banks (round 4);
//   * form (1): workgroup -> (tile block, Cout block) goes by XCD (blockIdx % 8): the Cout / 64 workgroups that read one tile
//     block's V meet in ONE L2;
//   * tail (shared): A^T M A per accumulator register in registers (120 adds / multiplies by 2, 4, 8 per 4x4 tile), staged
//     through the LDS ([tile][pixel][64 channels]) so that scale / shift, LeakyReLU and the residual run on 16-byte pieces and
//     an output pixel's 64 channels leave as 256 contiguous bytes; the sixteen residual loads of a thread are issued together
//     ahead of the store loop; STATS instantiation: column sums of y and y^2 per block for the training forward's batch norm;
//   * mosaic tiling (w44_tiling): a batch whose map side is not a multiple of 4 is tiled as ONE picture (round 4).
#include <cstdlib>
#include "y3_internal.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct W44Args {
    const float* x;      // [N,H,W,Cin]
    const float* u;      // packed [18 position pairs][Cin/8][Cout][4 channel pairs][2 positions][2 channels]
    const float* scale;  // [Cout]
    const float* shift;  // [Cout]
    const float* resid;  // [N,H,W,Cout] or nullptr
    float* y;            // [N,H,W,Cout]
    int N, H, W, Cin, Cout, act;
    int TH, TW, T;       // 4x4 output tiles per image column / row, and in total (mosaic: of the mosaic, T = TH * TW)
    int mr, mc;          // mosaic tiling (w44_tiling): the N = mr * mc images form ONE picture, mr rows x mc columns of images with a
                         // zero row / column between neighbours; 0 = every image tiled on its own
    float* stats;        // STATS instantiation: [tile blocks][2][Cout] column sums of y and y^2 per 16-tile block, the training
                         // forward's batch-norm statistics (y3_bn_train_stats_partials)
    float* v;            // two-kernel form: V = B^T d B, [tile blocks][Cin / 16][VSTAGE bytes] (written by the transform kernel)
    int xb;              // channel blocks per XCD rectangle (w44_block_of)
};

constexpr int BT = 16, BNC = 64, NTH = BT * 16;  // one wave per 16 tiles x 16 channels
constexpr int NW = NTH / 64;
constexpr int KC = 8;                          // input channels per K-step
constexpr int ROWB = KC * 4;                   // (one-kernel form) LDS bytes per (position, tile) row
constexpr int PLANE = BT * ROWB;               // one position's tiles
constexpr int STAGE = 36 * PLANE;              // 18,432 B: one K-step of V (either form)
constexpr int VSTAGE = 2 * STAGE;              // 36,864 B: the two-kernel form stages TWO K-steps (16 channels) at a time
constexpr int LDS_RING = 4 * STAGE;            // 73,728 B of K-loop buffers either way
constexpr int LDS_BYTES = LDS_RING + 512;      // + the tiles' pixel-index parts: 74,240 B, two workgroups per CU
constexpr unsigned OOB = 0x80000000u;
#ifndef W44_BDEPTH
#define W44_BDEPTH 6
#endif
constexpr int BDEPTH = W44_BDEPTH;             // (one-kernel form) weight fragment loads in flight per wave: one load = one lane's 16
                                               // bytes for a PAIR of positions (divides 18: the window runs on across K-steps)
#ifndef W44V_BDEPTH
#define W44V_BDEPTH 12
#endif
constexpr int VBDEPTH = W44V_BDEPTH;           // (two-kernel form) the same window; deeper: the loads queue behind the V DMAs (far memory)
                                               // in the in-order vmcnt queue
#ifndef W44V_AD
#define W44V_AD 3
#endif
constexpr int VAD = W44V_AD;                   // (two-kernel form) activation fragment reads (16 B = one position pair) ahead

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// 16 bytes per lane, global -> LDS without a register round trip: lane l's bytes land at lds_base + 16*l (lds_base is
// wave-uniform), an out-of-range `voff` writes zeros (see y3_conv_bf16x.hip; the builtin exists in the device pass only).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_base, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_base, 16, voff, soff, 0, 0);
#endif
}

// 1-D input transform B^T v for the 6-point rule; rows 1,2 and 3,4 share their sums
template <typename V> __device__ __forceinline__ V bt0(const V& v0, const V& v2, const V& v4) { return 4.f * v0 - 5.f * v2 + v4; }
template <typename V> __device__ __forceinline__ V bt5(const V& v1, const V& v3, const V& v5) { return 4.f * v1 - 5.f * v3 + v5; }

// all six outputs of B^T applied along a 6-vector
template <typename V> __device__ __forceinline__ void bt_row(const V (&e)[6], V (&o)[6]) {
    o[0] = bt0(e[0], e[2], e[4]);
    const V p = e[4] - 4.f * e[2], q = e[3] - 4.f * e[1];
    o[1] = p + q;
    o[2] = p - q;
    const V p2 = e[4] - e[2], q2 = 2.f * (e[3] - e[1]);
    o[3] = p2 + q2;
    o[4] = p2 - q2;
    o[5] = bt5(e[1], e[3], e[5]);
}

// ... written as row i of V = B^T d B (6 position planes of the one-kernel form's LDS stage)
__device__ __forceinline__ void put_row(const f32x2 (&e)[6], unsigned char* vs, int i) {
    f32x2 o[6];
    bt_row(e, o);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(vs + (i * 6 + j) * PLANE) = o[j];
}

// One staging job: rs / vs point at this thread's (tile, channel pair) inside the raw patch planes [k*6+l] and the V planes.
// JOB 0: row 0 of V (patch rows 0,2,4) | 1: rows 1,2 (patch rows 1..4) | 2: rows 3,4 (patch rows 1..4) | 3: row 5 (1,3,5)
template <int JOB>
__device__ __forceinline__ void transform_job(const unsigned char* rs, unsigned char* vs) {
    auto ld = [&](int k, int l) { return *reinterpret_cast<const f32x2*>(rs + (k * 6 + l) * PLANE); };
    f32x2 e[6];
    if (JOB == 0) {
#pragma unroll
        for (int l = 0; l < 6; ++l) e[l] = bt0(ld(0, l), ld(2, l), ld(4, l));
        put_row(e, vs, 0);
    } else if (JOB == 3) {
#pragma unroll
        for (int l = 0; l < 6; ++l) e[l] = bt5(ld(1, l), ld(3, l), ld(5, l));
        put_row(e, vs, 5);
    } else {
        f32x2 f[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            const f32x2 d1 = ld(1, l), d2 = ld(2, l), d3 = ld(3, l), d4 = ld(4, l);
            const f32x2 pq = JOB == 1 ? d4 - 4.f * d2 : d4 - d2;
            const f32x2 qq = JOB == 1 ? d3 - 4.f * d1 : 2.f * (d3 - d1);
            e[l] = pq + qq;
            f[l] = pq - qq;
        }
        put_row(e, vs, JOB == 1 ? 1 : 3);
        put_row(f, vs, JOB == 1 ? 2 : 4);
    }
}

// Pixel-index parts of tile t's patch / output rows and columns: pixel (row, column) of the tile's grid = pixel index
// rpart + cpart of the [N,H,W] tensor (separable in the mosaic too: image ry * mc + cx, row y, column x ->
// ((ry * mc) * H + y) * W  +  cx * H * W + x); -1 = no such pixel (outside the image: padding, a gap row / column of the
// mosaic, past the edge).  yy / xx are coordinates in the picture the tiling was made for (one image, or the mosaic).
__device__ __forceinline__ int w44_rpart(const W44Args& p, int n, int yy) {
    if (yy < 0) return -1;
    if (p.mr) {
        const int ry = yy / (p.H + 1);
        yy -= ry * (p.H + 1);
        return (ry < p.mr && yy < p.H) ? (ry * p.mc * p.H + yy) * p.W : -1;
    }
    return yy < p.H ? (n * p.H + yy) * p.W : -1;
}
__device__ __forceinline__ int w44_cpart(const W44Args& p, int xx) {
    if (xx < 0) return -1;
    if (p.mr) {
        const int cx = xx / (p.W + 1);
        xx -= cx * (p.W + 1);
        return (cx < p.mc && xx < p.W) ? cx * p.H * p.W + xx : -1;
    }
    return xx < p.W ? xx : -1;
}

#ifdef W44V_PROBE
// Clock probe build (tools/wino44v_probe.py; never in the product library): thread 0 of every workgroup stamps the 100 MHz
// clock at its phase boundaries into g_w44v_probe[block][8] (word 7: HW_ID | XCC_ID << 32).
__device__ unsigned long long* g_w44v_probe = nullptr;
#define W44V_STAMP(i) do { if (g_w44v_probe && tid == 0) g_w44v_probe[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define W44V_STAMP(i) do { } while (0)
#endif

// ---- tail (both forms) -----------------------------------------------------------------------------------------------------
// Pixel-index parts of the block's tiles ([BT][8] ints: 4 output rows, 4 output columns of every tile; -1: no such pixel), written
// by the first 16 threads at kernel START into their own 512 bytes behind the K-loop's buffers (visible after the first barrier):
// the tail's residual loads need them before its first barrier.
constexpr int TINFO_OFF = LDS_RING;
__device__ __forceinline__ void w44_tinfo(const W44Args& p, unsigned char* smem, int bt, int tid) {
    if (tid < BT) {
        int* tinfo = reinterpret_cast<int*>(smem + TINFO_OFF);
        const int t = bt * BT + tid;
        const int n = p.mr ? 0 : t / (p.TH * p.TW);
        const int r = t - n * p.TH * p.TW;
        const int ty = r / p.TW, tx = r - ty * p.TW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tinfo[8 * tid + q] = t < p.T ? w44_rpart(p, n, 4 * ty + q) : -1;
            tinfo[8 * tid + 4 + q] = t < p.T ? w44_cpart(p, 4 * tx + q) : -1;
        }
    }
}

// A^T M A in registers -> LDS ([tile][pixel][64 channels]: the K-loop's buffers are free now) -> scale / shift / LeakyReLU /
// residual on 16-byte pieces, 256 contiguous bytes per output pixel.  The caller has passed a workgroup barrier behind its last
// LDS read.  Order (round 6, from a clock probe of the phases, profiles/r06_w44v_probe.txt): the sixteen residual loads of a thread
// go out FIRST and land under the output transform; ONE wait for them ahead of the store loop, which then issues its sixteen
// stores back to back.  (Before: loads after the staging barrier, and - loads and stores share vmcnt on gfx950 and may complete
// out of order with each other, so hipcc waits vmcnt(0) for a load whenever a store is pending - every iteration of the store
// loop waited for the previous iteration's STORE to complete: 8-13 us per block for 64 KB.)  Out-of-range pixels / channels are
// out-of-range buffer offsets (no branches).
template <bool STATS>
__device__ __forceinline__ void w44_tail(const W44Args& p, const f32x4 (&acc)[36], unsigned char* smem, int bt, int n0, int tid,
                                         int wn) {
    const int lane = tid & 63, row16 = lane & 15, quart = lane >> 4;
    constexpr int RS = BNC + 4;                                      // staged output row stride in floats
    float* cs = reinterpret_cast<float*>(smem);
    const int* tinfo = reinterpret_cast<const int*>(smem + TINFO_OFF);
    const int c4 = (tid & 15) * 4;
    const int co = n0 + c4;
    // this thread's sixteen output pieces: (tile it, pixel tid / 16), channels co .. co + 3
    const int px = tid >> 4;
    unsigned off[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rpart = tinfo[8 * it + (px >> 2)], cpart = tinfo[8 * it + 4 + (px & 3)];
        off[it] = ((rpart | cpart) >= 0 && co < p.Cout) ? (unsigned)(((rpart + cpart) * p.Cout + co) * 4) : OOB;
    }
    f32x4 rv[16];
    if (p.resid) {
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.resid), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);
#pragma unroll
        for (int it = 0; it < 16; ++it)
            rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off[it], 0, 0));
    } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // 16x16 accumulator: row (tile) = 4 * (lane / 16) + r, column (channel) = lane % 16
        float tq[6][4];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float m0 = acc[i * 6 + 0][r], m1 = acc[i * 6 + 1][r], m2 = acc[i * 6 + 2][r], m3 = acc[i * 6 + 3][r],
                        m4 = acc[i * 6 + 4][r], m5 = acc[i * 6 + 5][r];
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            tq[i][0] = m0 + s1 + s2;
            tq[i][1] = d1 + 2.f * d2;
            tq[i][2] = s1 + 4.f * s2;
            tq[i][3] = d1 + 8.f * d2 + m5;
        }
        float* row = cs + ((quart * 4 + r) * 16) * RS + wn * 16 + row16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s1 = tq[1][q] + tq[2][q], d1 = tq[1][q] - tq[2][q], s2 = tq[3][q] + tq[4][q], d2 = tq[3][q] - tq[4][q];
            row[(0 * 4 + q) * RS] = tq[0][q] + s1 + s2;
            row[(1 * 4 + q) * RS] = d1 + 2.f * d2;
            row[(2 * 4 + q) * RS] = s1 + 4.f * s2;
            row[(3 * 4 + q) * RS] = d1 + 8.f * d2 + tq[5][q];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), visible to hipcc's wait-count pass: every rv[] has landed
    __syncthreads();
#ifdef W44V_PROBE
    if (g_w44v_probe && tid == 0) g_w44v_probe[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
#endif
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (co < p.Cout) {
        sc = *reinterpret_cast<const f32x4*>(p.scale + co);
        sh = *reinterpret_cast<const f32x4*>(p.shift + co);
    }
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        p.y, 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rowi = it * 16 + px;                   // (tile, pixel) row: 256 rows
        f32x4 v = *reinterpret_cast<const f32x4*>(cs + rowi * RS + c4);
        v = v * sc + sh;
        if (p.act) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
        }
        v += rv[it];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_y, off[it], 0, 0);
        if (STATS) {
            if (off[it] != OOB) { st1 += v; st2 += v * v; }
        }
    }
    if (STATS) {
        // column sums of this block's outputs: the 16 threads that share a channel quad (one per pixel slot, it = tile)
        // add up through the LDS in a fixed order -> deterministic
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [16 pixel slots][2][64]
        *reinterpret_cast<f32x4*>(red + (px * 2 + 0) * BNC + c4) = st1;
        *reinterpret_cast<f32x4*>(red + (px * 2 + 1) * BNC + c4) = st2;
        __syncthreads();
        if (tid < 16 && co < p.Cout) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a += *reinterpret_cast<const f32x4*>(red + (k * 2 + 0) * BNC + c4);
                b += *reinterpret_cast<const f32x4*>(red + (k * 2 + 1) * BNC + c4);
            }
            float* st = p.stats + (size_t)bt * 2 * p.Cout;
            *reinterpret_cast<f32x4*>(st + co) = a;
            *reinterpret_cast<f32x4*>(st + p.Cout + co) = b;
        }
    }
}

// Workgroup -> (tile block bt, channel block bn).  Workgroup b runs on XCD b % 8 (observed placement; only speed depends on it).
// Each XCD gets a contiguous run of the order "for channel-block group: for tile block: for channel block of the group" (groups
// of p.xb channel blocks): its workgroups then share a rectangle of (tile blocks) x (xb channel blocks), i.e. input patches / V
// images AND weight slices meet in ONE L2.  A weight slice (36 x Cin x 64) is four times a V image (36 x Cin x 16):
// y3_launch_conv_wino44 picks xb ~ sqrt(blocks / 32), which minimises (distinct tile blocks) x image + (distinct channel blocks)
// x slice per XCD.  (One channel block per XCD - what plain dispatch order gives for Cout = 256 / 512 - fetches every input four
// or eight times: 466 MB of fabric traffic per 52-grid launch against 173 MB algorithmic in round 5; all channel blocks on every
// XCD fetches the 75 MB of 13-grid weights 25 times.)  The grid is 8 * ceil(blocks / 8) workgroups; false = nothing to do.
__device__ __forceinline__ bool w44_block_of(const W44Args& p, int nbn, int& bt, int& bn) {
    const int nbt = (p.T + BT - 1) / BT;
    const int nblocks = nbt * nbn, per = (nblocks + 7) >> 3;
    const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (L >= nblocks) return false;
    const int xb = p.xb, grp = L / (nbt * xb), rem = L - grp * (nbt * xb);
    // (the last group may hold fewer than xb channel blocks when xb does not divide Cout / 64)
    const int gb = (grp + 1) * xb <= nbn ? xb : nbn - grp * xb;
    bt = rem / gb;
    bn = grp * xb + (rem - bt * gb);
    return true;
}

// ---- form (2): one kernel, B^T d B inside the K-loop -------------------------------------------------------------------
template <bool STATS>
__global__ void __launch_bounds__(NTH, 2) conv_wino44_f32_kernel(const W44Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // V: [2][36][BT][32 B], then raw patches: the same shape
    constexpr int RAW_OFF = 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;                                 // 4 waves: 16 tiles x 16 channels each
    const int nbn = p.Cout / BNC;
    const int ksteps = p.Cin / KC;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.u), 0, (unsigned)((size_t)36 * p.Cin * p.Cout * 4), 0x00020000);

    // ---- per-thread constants ---------------------------------------------------------------------------------------
    const int unit = tid & (BT * 4 - 1), job = __builtin_amdgcn_readfirstlane(tid / (BT * 4));
    const int st_off = (unit >> 2) * ROWB + (unit & 3) * 8;          // staging job: (tile, channel pair) inside a RAW plane ([tile][8 channels])
    const int row16 = lane & 15, quart = lane >> 4;
    // V planes are [channel pair][16 tiles][2 channels] with the tile index rotated by 4 per channel pair (v_off): the
    // fragment reads come out of hipcc as ds_read2st64_b64, which the LDS serves in groups of 16 lanes over 32 banks -
    // 16 tiles x 8 B of ONE channel pair must be 128 contiguous bytes (as [tile][8 channels] rows they hit every bank
    // four times: 3.5k conflict cycles per K-step, profiles/r03_pmc_layers.txt) - and the rotation keeps the staging
    // writes of a 16-lane group (4 tiles x 4 channel pairs) on 32 different banks as well.
    auto v_off = [](int tile, int cp) { return cp * (BT * 8) + ((tile * 8 + cp * 32) & (BT * 8 - 1)); };
    const int sv_off = v_off(unit >> 2, unit & 3);
    const int a_off = v_off(row16, quart);                           // activation fragment inside a position plane
    const unsigned b_pos_stride = (unsigned)((size_t)ksteps * p.Cout * 2 * KC * 4);         // bytes between position pairs
    const unsigned b_ks_stride = (unsigned)(p.Cout * 2 * KC * 4);

    int bt, bn;
    if (!w44_block_of(p, nbn, bt, bn)) return;
    const int t0 = bt * BT, n0 = bn * BNC;
    w44_tinfo(p, smem, bt, tid);

    // raw patches, global -> LDS by DMA: plane i = patch pixel (k, l) = (i / 6, i % 6) holds [BT tiles][8 channels];
    // one instruction moves 1 KB = PPI planes, wave w issues instructions w, w + NW, ...: lane = (plane, tile, 16-byte half)
    constexpr int PPI = 1024 / PLANE, NDMA = 36 / PPI;
    constexpr int DPW = (NDMA + NW - 1) / NW;
    unsigned dvoff[DPW];
    {
        const int t = t0 + ((lane >> 1) & (BT - 1));
        const bool tok = t < p.T;
        const int n = p.mr ? 0 : t / (p.TH * p.TW);
        const int r = t - n * p.TH * p.TW;
        const int ty = r / p.TW, tx = r - ty * p.TW;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int i = (wave + NW * j) * PPI + lane / (2 * BT);
            const int k = i / 6, l = i - 6 * k;
            const int rpart = w44_rpart(p, n, 4 * ty - 1 + k), cpart = w44_cpart(p, 4 * tx - 1 + l);
            const bool ok = tok && i < 36 && (rpart | cpart) >= 0;
            dvoff[j] = ok ? (unsigned)(((rpart + cpart) * p.Cin) * 4 + (lane & 1) * 16) : OOB;
        }
    }
    // (`live` = false: the same instructions with out-of-range offsets - zeros into a buffer nobody reads, nothing fetched: the
    // DMAs of the K-loop must not sit behind a branch, see dma_piece in conv_wino44v_f32_kernel)
    auto dma_raw = [&](int ks, int buf, bool live) {
        const unsigned so = (unsigned)(ks * KC) * 4u;
#pragma unroll
        for (int j = 0; j < DPW; ++j)
            if (wave + NW * j < NDMA) dma16(rs_x, smem + RAW_OFF + buf * STAGE + (wave + NW * j) * 1024, live ? dvoff[j] : OOB, so);
    };
    auto transform = [&](int bufr, int bufv) {
        const unsigned char* rs = smem + RAW_OFF + bufr * STAGE + st_off;
        unsigned char* vs = smem + bufv * STAGE + sv_off;
        if (job == 0) transform_job<0>(rs, vs);
        else if (job == 1) transform_job<1>(rs, vs);
        else if (job == 2) transform_job<2>(rs, vs);
        else transform_job<3>(rs, vs);
    };

    f32x4 acc[36];
#pragma unroll
    for (int pos = 0; pos < 36; ++pos) acc[pos] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weights are packed [pos / 2][Cin / 8][Cout][lane quarter][pos % 2][2 channels]: a lane's fragments of two
    // neighbouring positions are 16 contiguous bytes, a wave load 1 KB
    const unsigned b_voff = (n0 + wn * 16 + row16 < p.Cout)
        ? (unsigned)(((n0 + wn * 16 + row16) * 2 * KC + quart * 4) * 4) : OOB;
    f32x4 bq[BDEPTH];
    auto issue_b = [&](int slot, int pair, int ks) {
        bq[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs_u, b_voff, (unsigned)pair * b_pos_stride + (unsigned)ks * b_ks_stride, 0));
    };

    // ---- prologue: raw(0), raw(1) by DMA; V(0) = transform(raw(0)); the first weight fragments ---------------------------
    dma_raw(0, 0, true);
    dma_raw(1, 1, 1 < ksteps);
#pragma unroll
    for (int s = 0; s < BDEPTH; ++s) issue_b(s, s, 0);      // pairs 0 .. BDEPTH-1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BDEPTH) : "memory");     // the DMAs are older than the fragment loads
    __builtin_amdgcn_s_barrier();
    transform(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < ksteps;
        // DMA of raw(ks+2) (raw[cur] held raw(ks): consumed a K-step ago) and V(ks+1) from raw(ks+1) (it landed before the
        // last barrier; on the last K-step it transforms stale data into a buffer nobody reads: keeps the K-step's shape)
        dma_raw(ks + 2, cur, ks + 2 < ksteps);
        transform(cur ^ 1, cur ^ 1);
        const unsigned char* vs = smem + cur * STAGE + a_off;
        constexpr int AD = 4;                             // activation fragments read ahead (two pairs)
        f32x2 aq[AD];
#pragma unroll
        for (int s = 0; s < AD; ++s) aq[s] = *reinterpret_cast<const f32x2*>(vs + s * PLANE);
#pragma unroll
        for (int pr = 0; pr < 18; ++pr) {
            const f32x2 a0 = aq[(2 * pr) % AD], a1 = aq[(2 * pr + 1) % AD];
            const f32x4 b = bq[pr % BDEPTH];
            if (2 * pr + AD < 36) {
                aq[(2 * pr) % AD] = *reinterpret_cast<const f32x2*>(vs + (2 * pr + AD) * PLANE);
                aq[(2 * pr + 1) % AD] = *reinterpret_cast<const f32x2*>(vs + (2 * pr + 1 + AD) * PLANE);
            }
            // refill the slot with the pair BDEPTH pairs ahead (it runs on into the next K-step; past the last K-step it
            // re-reads a valid address and is never used)
            {
                const int np = pr + BDEPTH;
                if (np < 18) issue_b(pr % BDEPTH, np, ks);
                else issue_b(pr % BDEPTH, np - 18, more ? ks + 1 : ks);
            }
            acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b[0], acc[2 * pr], 0, 0, 0);
            acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b[2], acc[2 * pr + 1], 0, 0, 0);
            acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b[1], acc[2 * pr], 0, 0, 0);
            acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b[3], acc[2 * pr + 1], 0, 0, 0);
            // keep the software pipeline as written: left alone, hipcc's scheduler moves every fragment read right in
            // front of its MFMAs (lgkmcnt(0) / vmcnt(1..3) ahead of each pair: the LDS and L2 latencies in full, 72 times)
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BDEPTH) : "memory");     // this K-step's DMA has landed (older than the window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the window's last, unused fragments: their registers are reused)
    w44_tail<STATS>(p, acc, smem, bt, n0, tid, wn);
}

// ---- form (1), first kernel: V = B^T d B of every (tile, input channel), written as the second kernel's LDS stage images --------
// A wave = one image (16 tiles x 16 channels x 36 positions = 36 KB); lane (tile = lane % 16, j = lane / 16) holds channels
// 4j .. 4j+3 of the tile's 6x6 patch in registers (36 x 16-byte loads: the four lanes of a tile read 64 contiguous bytes per
// pixel; padding / mosaic gaps = out-of-range buffer offsets = zeros), transforms columns then rows exactly as the one-kernel
// form does, and writes per position pair two 16-byte pieces {position 2pp, 2pp+1} x {channel pair}: 16 lanes = 256
// contiguous bytes.  Channel pair cp = 2j, 2j+1 of the 16 lies in K-step cp / 4 of the image, lane quarter cp % 4.
__global__ void __launch_bounds__(256) wino44_input_transform_kernel(const W44Args p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nkb = p.Cin / 16, nbt = (p.T + BT - 1) / BT;
    const long long item = (long long)blockIdx.x * 4 + wave;          // (tile block, 16-channel block): the channel blocks of a tile block are neighbours
    if (item >= (long long)nbt * nkb) return;
    const int bt = (int)(item / nkb), kb = (int)(item - (long long)bt * nkb);
    const int tl = lane & 15, j = lane >> 4;
    const int t = bt * BT + tl;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const int n = p.mr ? 0 : t / (p.TH * p.TW);
    const int r = t - n * p.TH * p.TW;
    const int ty = r / p.TW, tx = r - ty * p.TW;
    int rpart[6], cpart[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        rpart[k] = t < p.T ? w44_rpart(p, n, 4 * ty - 1 + k) : -1;
        cpart[k] = w44_cpart(p, 4 * tx - 1 + k);
    }
    const unsigned coff = (unsigned)((kb * 16 + j * 4) * 4);
    f32x4 d[6][6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            const bool ok = (rpart[k] | cpart[l]) >= 0;
            d[k][l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                          rs_x, ok ? (unsigned)((rpart[k] + cpart[l]) * p.Cin) * 4u + coff : OOB, 0, 0));
        }
    unsigned char* img = reinterpret_cast<unsigned char*>(p.v) + ((size_t)bt * nkb + kb) * VSTAGE
                       + (j >> 1) * STAGE + (2 * (j & 1)) * 256 + tl * 16;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4 e[6], o[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            if (i == 0) e[l] = bt0(d[0][l], d[2][l], d[4][l]);
            else if (i == 5) e[l] = bt5(d[1][l], d[3][l], d[5][l]);
            else {
                const f32x4 pq = i <= 2 ? d[4][l] - 4.f * d[2][l] : d[4][l] - d[2][l];
                const f32x4 qq = i <= 2 ? d[3][l] - 4.f * d[1][l] : 2.f * (d[3][l] - d[1][l]);
                e[l] = (i & 1) ? pq + qq : pq - qq;
            }
        }
        bt_row(e, o);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            unsigned char* dst = img + (i * 3 + m) * 1024;
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[2 * m][0], o[2 * m][1], o[2 * m + 1][0], o[2 * m + 1][1]};
            *reinterpret_cast<f32x4*>(dst + 256) = f32x4{o[2 * m][2], o[2 * m][3], o[2 * m + 1][2], o[2 * m + 1][3]};
        }
    }
}


// ---- form (1), second kernel: 36 batched GEMMs on the V images + the tail ---------------------------------------------------
// Per 16 channels (two K-steps, one barrier): nine 1 KB DMA pieces per wave for the NEXT 16 channels into the other buffer,
// then per position pair one ds_read_b128 (address = pair * 1 KB + lane * 16: the guide's conflict-free form), one 16-byte
// weight load and four MFMAs.
template <bool STATS>
__global__ void __launch_bounds__(NTH, 2) conv_wino44v_f32_kernel(const W44Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][2 K-steps][18 pairs][1 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;
    const int nbn = p.Cout / BNC;
    const int ksteps = p.Cin / KC, nds = p.Cin / 16;
    int bt, bn;
    if (!w44_block_of(p, nbn, bt, bn)) return;
    const int n0 = bn * BNC;
    const int row16 = lane & 15, quart = lane >> 4;
#ifdef W44V_PROBE
    if (g_w44v_probe && tid == 0)
        g_w44v_probe[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                   ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) << 32);
#endif
    W44V_STAMP(0);
    w44_tinfo(p, smem, bt, tid);
#ifdef W44V_STAGGER
    // probe: spread the starts of the first resident workgroups over W44V_STAGGER us (100 MHz clock), by slot inside the XCD
    if (blockIdx.x < 512) {
#ifdef W44V_STAGGER_PAIR
        const unsigned long long wait = (unsigned long long)W44V_STAGGER * 100ull * (((blockIdx.x >> 3) / W44V_STAGGER_PAIR) & 1);
#else
        const unsigned long long wait = (unsigned long long)W44V_STAGGER * 100ull * ((blockIdx.x >> 3) & 63) / 64;
#endif
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(64);
    }
#endif

    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(p.v) + (size_t)bt * nds * VSTAGE, 0, (unsigned)(nds * VSTAGE), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.u), 0, (unsigned)((size_t)36 * p.Cin * p.Cout * 4), 0x00020000);
    const unsigned b_pos_stride = (unsigned)((size_t)ksteps * p.Cout * 2 * KC * 4);         // bytes between position pairs
    const unsigned b_ks_stride = (unsigned)(p.Cout * 2 * KC * 4);
    const unsigned b_voff = (n0 + wn * 16 + row16 < p.Cout)
        ? (unsigned)(((n0 + wn * 16 + row16) * 2 * KC + quart * 4) * 4) : OOB;
    const unsigned d_voff = (unsigned)lane * 16u;

    // piece j (1 KB) of the 36 KB image of 16 channels: wave w moves pieces w, w + 4, ...; `live` = false: the same instruction
    // with an out-of-range offset (writes zeros into a buffer nobody reads, fetches nothing) - the DMAs of the loop below must
    // be UNCONDITIONAL: hipcc's wait-count pass merges the two sides of a branch to the smaller count, and with the DMAs
    // behind `if (ds + 1 < nds)` every weight-fragment wait of the first iterations also waited for a just-issued DMA
    auto dma_piece = [&](int ds, int buf, int j, bool live) {
        dma16(rs_v, smem + buf * VSTAGE + (wave + NW * j) * 1024, live ? d_voff : OOB, (unsigned)(ds * VSTAGE + (wave + NW * j) * 1024));
    };

    f32x4 acc[36];
#pragma unroll
    for (int pos = 0; pos < 36; ++pos) acc[pos] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bq[VBDEPTH];
    auto issue_b = [&](int slot, int pair, int ks) {
        bq[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs_u, b_voff, (unsigned)pair * b_pos_stride + (unsigned)ks * b_ks_stride, 0));
    };

#pragma unroll
    for (int j = 0; j < 9; ++j) dma_piece(0, 0, j, true);
#pragma unroll
    for (int s = 0; s < VBDEPTH; ++s) issue_b(s, s, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VBDEPTH) : "memory");     // the DMAs are older than the fragment loads
    __builtin_amdgcn_s_barrier();
    W44V_STAMP(1);

    for (int ds = 0; ds < nds; ++ds) {
        const int cur = ds & 1;
        const bool more = ds + 1 < nds;
        const unsigned char* vs = smem + cur * VSTAGE + lane * 16;
        f32x4 aq[VAD];
#pragma unroll
        for (int s = 0; s < VAD; ++s) aq[s] = *reinterpret_cast<const f32x4*>(vs + s * 1024);
#pragma unroll
        for (int ii = 0; ii < 36; ++ii) {                    // (K-step, position pair) = (ii / 18, ii % 18)
            const int ks = 2 * ds + ii / 18, pr = ii % 18;
            // the next 16 channels into the other buffer (read until the barrier just passed), one 1 KB piece per wave every
            // third position pair: weight loads queue behind DMAs in the in-order vmcnt queue, and a burst of nine would hold
            // them back for a memory latency + 36 KB; the last piece leaves 12 weight loads behind it (>= the window)
#ifdef W44V_DMA_TOP
            if (ii == 0) {
#pragma unroll
                for (int j = 0; j < 9; ++j) dma_piece(ds + 1, cur ^ 1, j, more);
            }
#elif !defined(W44V_KO_DMA)
            if (ii % 3 == 0 && ii / 3 < 9) dma_piece(ds + 1, cur ^ 1, ii / 3, more);
#endif
            const f32x4 a = aq[ii % VAD];
            const f32x4 b = bq[ii % VBDEPTH];
#ifndef W44V_KO_A
            if (ii + VAD < 36) aq[ii % VAD] = *reinterpret_cast<const f32x4*>(vs + (ii + VAD) * 1024);
#endif
            {
                // the window runs on into the next K-step; past the last one it re-reads a valid address and is never used
                const int np = pr + VBDEPTH;
                const int nks = ks + np / 18;
#ifndef W44V_KO_B
                issue_b(ii % VBDEPTH, np % 18, nks < ksteps ? nks : ks);
#endif
            }
            acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[2 * pr], 0, 0, 0);
            acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[2 * pr + 1], 0, 0, 0);
            acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[2 * pr], 0, 0, 0);
            acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[2 * pr + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(VBDEPTH <= 12, "the last DMA piece must be older than the whole weight window");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VBDEPTH) : "memory");     // the next 16 channels have landed (older than the window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the window's last, unused fragments: their registers are reused)
    W44V_STAMP(2);
#ifdef W44V_KO_TAIL
    {   // probe build: no output transform / store phase (one never-taken store keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int pos = 0; pos < 36; ++pos) sum += acc[pos][0] + acc[pos][1] + acc[pos][2] + acc[pos][3];
        if (sum == 1.2345e-30f) p.y[tid] = sum;
        return;
    }
#endif
    w44_tail<STATS>(p, acc, smem, bt, n0, tid, wn);
#if defined(W44V_PROBE) || defined(W44V_END_WAIT)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the probe's clock wants the stores' completion; the product ends behind their ISSUE)
#endif
    W44V_STAMP(3);
}

// U = G g G^T for every (ci, co), G the 6x3 matrix of F(4x4,3x3); out[pos/2][ci/8][co][(ci%8)/2][pos%2][ci%2]
// dgrad != 0: w is the FORWARD kernel as [9][cout][cin] (its HWIO layout, the forward's input channels = this conv's cout):
// the data gradient's kernel g'[a][b][ci][co] = w[2-a][2-b][co][ci] (flipped taps, channel axes swapped)
__global__ void __launch_bounds__(256) pack_weights_wino44_kernel(const float* __restrict__ w_hwio, float* __restrict__ out,
                                                                  int cin, int cout, int dgrad) {
    const long long total = (long long)cin * cout;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int cil = (int)(e % KC);
        const long long r = e / KC;
        const int co = (int)(r % cout);
        const int kb = (int)(r / cout);
        const int ci = kb * KC + cil;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g[a][b] = dgrad ? w_hwio[((size_t)((2 - a) * 3 + (2 - b)) * cout + co) * cin + ci]
                                : w_hwio[((size_t)(a * 3 + b) * cin + ci) * cout + co];
        // rows of G: [1/4,0,0], [-1/6,-1/6,-1/6], [-1/6,1/6,-1/6], [1/24,1/12,1/6], [1/24,-1/12,1/6], [0,0,1]
        float t[6][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
            t[0][b] = g0 * (1.f / 4.f);
            t[1][b] = -(g0 + g1 + g2) * (1.f / 6.f);
            t[2][b] = -(g0 - g1 + g2) * (1.f / 6.f);
            t[3][b] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
            t[4][b] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
            t[5][b] = g2;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const float t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
            const float u[6] = {t0 * (1.f / 4.f), -(t0 + t1 + t2) * (1.f / 6.f), -(t0 - t1 + t2) * (1.f / 6.f),
                                t0 * (1.f / 24.f) + t1 * (1.f / 12.f) + t2 * (1.f / 6.f),
                                t0 * (1.f / 24.f) - t1 * (1.f / 12.f) + t2 * (1.f / 6.f), t2};
#pragma unroll
            for (int b = 0; b < 6; ++b)
            {
                const int pos = a * 6 + b;
                // [pos / 2][cin / 8][cout][channel pair = lane quarter][pos % 2][channel % 2]
                out[((((size_t)(pos >> 1) * (cin / KC) + kb) * cout + co) * 2 * KC) + (cil >> 1) * 4 + (pos & 1) * 2 + (cil & 1)] = u[b];
            }
        }
    }
}

}  // namespace

#ifdef W44V_PROBE
extern "C" __attribute__((visibility("default"))) int y3_debug_w44v_probe(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_w44v_probe), &buf, sizeof(buf));
}
#endif

int y3_conv_wino44_eligible_impl(const y3_conv_desc* d) {
    return d && d->k == 3 && d->stride == 1 && d->c_up == 0 && d->cin % 32 == 0 && d->cout % 64 == 0 &&
           d->n > 0 && d->h > 1 && d->w > 1;
}

// The convs y3_net_forward (dtype 4) runs on this kernel instead of the F(2x2,3x3) one, given the alternative packing
// (y3_net_set_layer_alt): where it measured faster inside the bs=32 416x416 forward (tools/layer_profile.py,
// profiles/r04_wino44.txt; ms per layer, F(2x2) | F(4x4) one-kernel form):
//   64->128 @104: 0.280 | 0.260     128->256 @52: 0.244 | 0.186     256->512 @26: 0.222 | 0.205     512->1024 @13: 0.240 | 0.198
//   32->64 @208: 0.361 | 0.372 - four K-steps per block, the block prologue and store tail dominate: stays on F(2x2) at bs=32.
//   candidate (shape only: what a caller packs for): every eligible conv;
//   preferred (this launch): a candidate whose blocks - 16 tiles x 64 channels, one workgroup each - fill at least half of
//     the 512 workgroup slots (below that the F(2x2) kernel's stream-K schedule keeps every CU busy and ties or wins); the
//     four-K-step Cin = 32 shape only up to 4,096 blocks.
// (experiments build only) Y3_WINO44=0 turns the kernel off, =2 takes every eligible conv whatever its size (A/B runs).
static int wino44_mode() {
    static const int mode = y3_exp_env("Y3_WINO44") ? atoi(y3_exp_env("Y3_WINO44")) : 1;
    return mode;
}

int y3_conv_wino44_candidate_impl(const y3_conv_desc* d) {
    if (wino44_mode() == 0 || !y3_conv_wino44_eligible_impl(d)) return 0;
    return 1;
}

// How the output is cut into 4x4 tiles.  A map whose side is a multiple of 4 is tiled image by image.  Otherwise (the 13- and
// 26-grids of a 416-pixel input) every image would pad its last tile row and column with zero work (26: 49 tiles for 42.25
// tiles' worth of pixels; 13: 16 for 10.6), so the batch is tiled as ONE picture instead: the N = mr x mc images laid out
// mr rows by mc columns with one zero row / column between neighbours.  The gap is exactly the zero padding both neighbours
// need, so a tile may straddle two images (its gap outputs are never stored), and only the picture's last tile row / column
// pads.  bs=32 at 26x26: 4 x 8 images = 107 x 215 pixels = 27 x 54 = 1,458 tiles instead of 1,568 - 736 blocks instead of 784,
// which is one block per CU less than three rounds of 256 instead of one more (time goes with ceil(blocks / 256),
// profiles/r04_wino44.txt 5).  mr * mc == N exactly (mr a divisor of N: a prime batch becomes a 1 x N strip).
struct W44Tiling { int TH, TW, T, mr, mc; };
static W44Tiling w44_tiling(const y3_conv_desc* d) {
    W44Tiling t;
    t.mr = t.mc = 0;
    t.TH = (d->h + 3) / 4; t.TW = (d->w + 3) / 4; t.T = d->n * t.TH * t.TW;
    static const bool off = y3_exp_env("Y3_WINO44_MOSAIC") && atoi(y3_exp_env("Y3_WINO44_MOSAIC")) == 0;
    if ((d->h % 4 == 0 && d->w % 4 == 0) || off) return t;
    for (int r = 1; r <= d->n; ++r) {
        if (d->n % r) continue;
        const int c = d->n / r;
        const long long th = ((long long)r * (d->h + 1) - 1 + 3) / 4, tw = ((long long)c * (d->w + 1) - 1 + 3) / 4;
        if (th * tw < t.T) { t.mr = r; t.mc = c; t.TH = (int)th; t.TW = (int)tw; t.T = (int)(th * tw); }
    }
    return t;
}

int y3_conv_wino44_preferred_impl(const y3_conv_desc* d) {
    if (!y3_conv_wino44_candidate_impl(d)) return 0;
    if (wino44_mode() == 2) return 1;
    // (thresholds on the image-by-image tile count: the measurements above were taken on it, and the choice of kernel should not
    // move because the mosaic saves a few blocks)
    const long long tiles = (long long)d->n * ((d->h + 3) / 4) * ((d->w + 3) / 4);
    const long long blocks = ((tiles + BT - 1) / BT) * (d->cout / BNC);
    return blocks >= 128 && (d->cin >= 64 || blocks <= 4096);
}

// rows of the `stats` output of the STATS instantiation: one per 16-tile block
int y3_conv_wino44_stats_blocks_impl(const y3_conv_desc* d) {
    if (!y3_conv_wino44_eligible_impl(d)) return 0;
    const long long tiles = w44_tiling(d).T;
    return (int)((tiles + BT - 1) / BT);
}

int y3_launch_pack_wino44(hipStream_t stream, const float* w_hwio, int cin, int cout, float* out, int dgrad) {
    const long long total = (long long)cin * cout;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_weights_wino44_kernel, dim3(blocks), dim3(256), 0, stream, w_hwio, out, cin, cout, dgrad);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

// Bytes of V for this conv: what the two-kernel form needs as its workspace (0: the conv always runs as one kernel).
size_t y3_conv_wino44_workspace_bytes_impl(const y3_conv_desc* d) {
    if (!y3_conv_wino44_eligible_impl(d)) return 0;
    const long long nbt = (w44_tiling(d).T + BT - 1) / BT;
    return (size_t)nbt * (d->cin / 16) * VSTAGE;
}

// Which form a launch WITH a sufficient workspace takes (1 = two kernels).  The extra pass costs time in proportion to tiles x
// Cin, the batched GEMMs save in proportion to tiles x Cin x Cout: measured (profiles/r06_wino44_forms.txt; us per launch at
// bs=32, one kernel | two kernels) 52-grid 128->256 162 | 176, 26-grid 256->512 157 | 141, 13-grid 512->1024 189 | 163, the data
// gradients 52-grid 256->128 162 | 219, 26-grid 512->256 190 | 182, 13-grid 1024->512 212 | 172; bs=64 alike): Cout >= 512.
// (experiments build only) Y3_WINO44_V=0 / 1 forces one form on every conv.
int y3_conv_wino44_two_pass_impl(const y3_conv_desc* d) {
    if (!y3_conv_wino44_eligible_impl(d) || d->cin % 16) return 0;
    static const int force = y3_exp_env("Y3_WINO44_V") ? atoi(y3_exp_env("Y3_WINO44_V")) : -1;
    if (force >= 0) return force;
    return d->cout >= 512;
}

static int w44_set_lds(const void* kern, int slot) {
    static bool done[4][Y3_MAX_DEVICES] = {};      // per kernel and device (idempotent: a race sets it twice)
    const int dev = y3_current_device();
    if (dev < 0 || !done[slot][dev]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev >= 0) done[slot][dev] = true;
    }
    return Y3_OK;
}

int y3_launch_conv_wino44(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* u, const float* scale,
                          const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes,
                          const y3_sk_opts* sk) {
    Y3_CHECK_ARG(d && x && u && scale && shift && y, "y3_conv2d_fwd_wino44: null pointer argument");
    Y3_CHECK_ARG(y3_conv_wino44_eligible_impl(d),
                 "y3_conv2d_fwd_wino44: needs a 3x3 stride-1 conv with Cin %% 32 == 0 and Cout %% 64 == 0, no fused upsample input");
    Y3_CHECK_ARG((long long)d->n * d->h * d->w * d->cin < (1LL << 29) && (long long)d->n * d->h * d->w * d->cout < (1LL << 29) &&
                     (long long)36 * d->cin * d->cout < (1LL << 29),
                 "y3_conv2d_fwd_wino44: tensor exceeds 2^29 elements (32-bit byte offsets)");
    W44Args a;
    a.x = x; a.u = u; a.scale = scale; a.shift = shift; a.resid = residual; a.y = y;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.act = d->act;
    const W44Tiling til = w44_tiling(d);
    a.TH = til.TH; a.TW = til.TW; a.T = til.T; a.mr = til.mr; a.mc = til.mc;
    a.stats = sk ? sk->stats : nullptr;
    a.v = nullptr; a.xb = 1;
    const int nbt = (a.T + BT - 1) / BT, nbn = d->cout / BNC;
    // channel blocks per XCD rectangle (w44_block_of): the power of two nearest sqrt(blocks / 32) in log scale, at most nbn
    const long long nblocks = (long long)nbt * nbn;
    int xb = 1;
    while (xb * 2 <= nbn && (long long)(xb * 2) * (xb * 2) * 32 <= 2 * nblocks) xb *= 2;
    static const int xb_force = y3_exp_env("Y3_WINO44_XB") ? atoi(y3_exp_env("Y3_WINO44_XB")) : 0;
    if (xb_force > 0) xb = xb_force < nbn ? xb_force : nbn;
    a.xb = xb;
    const unsigned grid = (unsigned)(8 * ((nblocks + 7) / 8));
    const size_t vbytes = y3_conv_wino44_workspace_bytes_impl(d);
    const bool two_pass = workspace != nullptr && workspace_bytes >= vbytes && ((uintptr_t)workspace & 15) == 0 &&
                          y3_conv_wino44_two_pass_impl(d);
    if (two_pass) {
        a.v = static_cast<float*>(workspace);
        const long long items = (long long)nbt * (d->cin / 16);
        hipLaunchKernelGGL(wino44_input_transform_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, a);
        Y3_CHECK_HIP(hipGetLastError());
        auto kern = a.stats ? conv_wino44v_f32_kernel<true> : conv_wino44v_f32_kernel<false>;
        if (int rc = w44_set_lds(reinterpret_cast<const void*>(kern), a.stats ? 3 : 2)) return rc;
#ifdef W44V_LDS      // probe: a larger request keeps the second workgroup off the CU
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W44V_LDS));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), W44V_LDS, stream, a);
#else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), LDS_BYTES, stream, a);
#endif
    } else {
        auto kern = a.stats ? conv_wino44_f32_kernel<true> : conv_wino44_f32_kernel<false>;
        if (int rc = w44_set_lds(reinterpret_cast<const void*>(kern), a.stats ? 1 : 0)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTH), LDS_BYTES, stream, a);
    }
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
