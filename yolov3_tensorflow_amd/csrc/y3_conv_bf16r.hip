// bf16-storage 1x1 convs (BASELINE configs[4]; utils/layer_utils.py:9-22 with kernel_size 1, model.py:55-77 for the head's
// route / detection convs and the fused upsample + concat input): a PERSISTENT kernel whose operands stream through a ring
// of LDS stages filled by LDS-DMA (`buffer_load ... lds`), round 5.
//
// Why a second 1x1 kernel: at bs=16, 608x608 a 1x1 conv is 6 GFLOP over 19-142 MB of tensors - 3-23 us of HBM time - and
// the register-staged kernel of y3_conv_bf16.hip spent 21-24 us on every one of them: one workgroup per 64x64 / 128x128
// tile, a cold prologue per tile, and a K-loop that waits out a global-load latency every one or two 32-element K-steps
// (19-grid 1024->512: 32 K-steps of ~1,450 cycles for 64 cycles of MFMA each, profiles/r03_layers_bs16_608_bf16.csv).
// Here:
//   * a workgroup (four waves) owns a contiguous run of output tiles and walks a stream of (tile, K-step) ITEMS; item i
//     lives in LDS stage i % NSTAGE.  One barrier per item: wait (counted `s_waitcnt vmcnt`) for item i | barrier | issue
//     the DMAs of item i + NSTAGE - 1 into the stage item i - 1 just left | fragment reads + MFMAs of item i.  The ring
//     runs on ACROSS tile boundaries: the next tile's first K-steps land while this tile's epilogue runs, so the
//     load latency is paid once per workgroup, not once per tile or K-step;
//   * K-step = 64 channels = whole 128-byte lines per row; stage image [BM + BN rows][128 B], lane-linear per DMA
//     instruction (8 rows), the XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free sits on the SOURCE
//     address (as in y3_conv_bf16x.hip); ragged rows / Cout tails / dead items past the end are out-of-range offsets =
//     zeros;
//   * weights packed [Cin/64][Cout][64] (y3_launch_pack_bf16x with one tap): the B tile of a K-step is contiguous;
//   * the fused upsample + concat input (model.py:60-62,70-72): a K-step's 64 channels come either from the
//     half-resolution tensor (pixel (y/2, x/2)) or from the route tensor - never both (c_up % 64 == 0);
//   * epilogue per WAVE, no workgroup barrier: 32 accumulator rows at a time through a wave-private LDS patch, fp32
//     scale / shift / LeakyReLU / residual, ONE rounding to bf16, 16 bytes per lane (8 channels) to memory; the detection
//     convs (3*(5+C) = 255 channels, fp32 output) leave as 4-byte-aligned dwordx4 pieces with a per-element tail.
//     The epilogue ends with ONE `vmcnt(0)` (its own loads and stores share the counter with the DMAs; nothing here relies
//     on loads and stores retiring in order relative to each other).
#include <cstdlib>
#include "y3_internal.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_base, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_base, 16, voff, soff, 0, 0);
#endif
}

struct ConvArgsR {
    const bf16_t* x;     // [N,H,W,Cx]
    const bf16_t* xu;    // [N,H/2,W/2,Cu] or nullptr
    const bf16_t* w;     // packed [Cin/64][Cout][64]
    const float* scale;
    const float* shift;
    const bf16_t* resid; // [M,Cout] or nullptr
    void* y;             // [M,Cout] bf16, or fp32 when out_f32
    int N, H, W, Cin, Cu, Cx, Cout;
    int act, out_f32;
    int M;
};

constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round to nearest even (finite inputs)
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// slot of a 128-byte row that holds logical 16-byte chunk c: c ^ swz(row)
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int MI, int NI, int WM, int WN, int NSTAGE, bool UPCAT>
__global__ void __launch_bounds__(256) conv1x1_bf16r_kernel(const ConvArgsR p) {
    constexpr int ROWB = 128, NW = 4;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int AQ = BM / 8, BQ = BN / 8;                   // DMA instructions per item (8 rows each)
    constexpr int ACH = AQ / NW, BCH = BQ / NW;               // per wave: the same count in every wave (counted waits)
    constexpr int DPW = ACH + BCH;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int WTM = MI * 32, WTN = NI * 32;
    constexpr int SROW = WTN + 4;                             // wave-private epilogue patch: 32 rows of WTN floats
    constexpr int EPATCH = 32 * SROW * 4;
    static_assert(WM * WN == NW, "four waves");
    static_assert(AQ % NW == 0 && BQ % NW == 0 && ACH >= 1 && BCH >= 1, "DMA instructions must divide over the waves");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NSTAGE][STAGE] ring, then [NW][EPATCH]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.Cout + BN - 1) / BN;
    const int kchunks = p.Cin / 64;
    const int ntiles = nbm * nbn;

    // This workgroup's contiguous run of tiles.  Tile ids are ROW-major (tile = bm * nbn + bn: neighbours share their A rows)
    // and the runs are handed out XCD by XCD (workgroup b runs on XCD b % 8, observed; placement is a speed matter only): one
    // XCD's workgroups cover a contiguous range of row blocks for every column block, so an activation row is fetched into
    // ONE L2 instead of into nbn of them (19-grid 1024->512: 55 -> 20 MB over the fabric) and only the weights - the small
    // operand of a 1x1 conv - are read by all eight.
    const int G = gridDim.x;
    const int g = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int tq = ntiles / G, tr = ntiles - tq * G;
    const int t0 = g * tq + (g < tr ? g : tr);
    const int t1 = t0 + tq + (g < tr ? 1 : 0);
    if (t0 >= t1) return;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(UPCAT ? p.xu : p.x), 0,
        (unsigned)(UPCAT ? (size_t)p.N * (p.H >> 1) * (p.W >> 1) * p.Cu * 2 : 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.w), 0, (unsigned)((size_t)p.Cout * p.Cin * 2), 0x00020000);

    // ---- loader: runs NSTAGE - 1 items ahead of the compute --------------------------------------------------------------
    const int l_row = lane >> 3, l_slot = lane & 7;
    unsigned a_off[ACH], a_off_u[UPCAT ? ACH : 1], b_off[BCH];
    int ld_tile = t0, ld_ks = 0;
    auto set_tile = [&](int tile) {
        const int bm = tile / nbn, bn = tile - bm * nbn;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int r = (wave * ACH + j) * 8 + l_row;
            const int chunk = l_slot ^ swz(r);
            const int m = bm * BM + r;
            a_off[j] = m < p.M ? (unsigned)(m * p.Cx + chunk * 8) * 2u : OOB;
            if (UPCAT) {
                const int hw = p.H * p.W;
                const int n = m / hw;
                const int rem = m - n * hw;
                const int oy = rem / p.W, ox = rem - oy * p.W;
                a_off_u[j] = m < p.M ? (unsigned)(((n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cu + chunk * 8) * 2u
                                     : OOB;
            }
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int r = (wave * BCH + j) * 8 + l_row;
            const int chunk = l_slot ^ swz(r);
            const int co = bn * BN + r;
            b_off[j] = co < p.Cout ? (unsigned)(co * 64 + chunk * 8) * 2u : OOB;
        }
    };
    auto issue = [&](int stage) {
        const bool live = ld_tile < t1;
        const int c0 = ld_ks * 64;
        const bool from_up = UPCAT && c0 < p.Cu;
        // (the scalar offset is not range-checked: a dead item must not carry one)
        const unsigned soff = live ? (unsigned)(from_up ? c0 : c0 - (UPCAT ? p.Cu : 0)) * 2u : 0u;
        const unsigned wsoff = live ? (unsigned)(ld_ks * p.Cout) * 128u : 0u;
        unsigned char* as = smem + stage * STAGE + (wave * ACH) * (8 * ROWB);
        unsigned char* bs = smem + stage * STAGE + BM * ROWB + (wave * BCH) * (8 * ROWB);
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            if (UPCAT) {
                if (from_up) dma16(rs_u, as + j * (8 * ROWB), live ? a_off_u[j] : OOB, soff);
                else dma16(rs_x, as + j * (8 * ROWB), live ? a_off[j] : OOB, soff);
            } else {
                dma16(rs_x, as + j * (8 * ROWB), live ? a_off[j] : OOB, soff);
            }
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) dma16(rs_w, bs + j * (8 * ROWB), live ? b_off[j] : OOB, wsoff);
        if (live && ++ld_ks == kchunks) {
            ld_ks = 0;
            if (++ld_tile < t1) set_tile(ld_tile);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // lane l feeds row l & 31 of a 32x32 tile with the 8 k-values of chunk 2*kk + (l >> 5); tile row bases are multiples
    // of 32, so the swizzle term depends on the lane only
    const int frag_row = lane & 31, frag_half = lane >> 5, frag_f = swz(frag_row);
    auto compute = [&](int stage) {
        const unsigned char* as = smem + stage * STAGE + (wm * WTM + frag_row) * ROWB;
        const unsigned char* bs = smem + stage * STAGE + BM * ROWB + (wn * WTN + frag_row) * ROWB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = ((2 * kk + frag_half) ^ frag_f) << 4;
            bf16x8 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(as + mi * 32 * ROWB + off));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bs + ni * 32 * ROWB + off));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // ---- epilogue of one tile, per wave ---------------------------------------------------------------------------------
    float* patch = reinterpret_cast<float*>(smem + NSTAGE * STAGE + wave * EPATCH);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    auto epilogue = [&](int tile) {
        const int bm = tile / nbn, bn = tile - bm * nbn;
        const int mw = bm * BM + wm * WTM, nw = bn * BN + wn * WTN;     // this wave's corner
        const bool wide = !p.out_f32 && (p.Cout & 7) == 0;              // bf16 out, 8-channel (16-byte) pieces
        const bool quad = p.out_f32 && !p.resid;                        // fp32 out (detection convs): 4-channel pieces
        if (!wide && !quad) {
            // any other combination (no layer of the network): per element, straight from the accumulators
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = nw + ni * 32 + col_l;
                const bool cok = col < p.Cout;
                const float sc = cok ? p.scale[col] : 0.f, sh = cok ? p.shift[col] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mw + mi * 32 + row_l + (r & 3) + 8 * (r >> 2);
                        if (cok && row < p.M) {
                            float v = acc[mi][ni][r] * sc + sh;
                            if (p.act) v = v > 0.f ? v : 0.1f * v;
                            const size_t o = (size_t)row * p.Cout + col;
                            if (p.resid) v += __uint_as_float((unsigned)p.resid[o] << 16);
                            if (p.out_f32) static_cast<float*>(p.y)[o] = v;
                            else static_cast<bf16_t*>(p.y)[o] = f32_to_bf16(v);
                        }
                    }
            }
            return;
        }
        if (wide) {
            constexpr int P = WTN / 8, RPP = 64 / P, PASSES = 32 / RPP;     // pieces per row, rows per pass
            const int piece = lane % P, rr0 = lane / P;
            const int col = nw + piece * 8;
            const bool cok = col < p.Cout;
            f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
            if (cok) {
                sc0 = *reinterpret_cast<const f32x4*>(p.scale + col);
                sc1 = *reinterpret_cast<const f32x4*>(p.scale + col + 4);
                sh0 = *reinterpret_cast<const f32x4*>(p.shift + col);
                sh1 = *reinterpret_cast<const f32x4*>(p.shift + col + 4);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                u32x4 rv[PASSES];
                if (p.resid) {
#pragma unroll
                    for (int i = 0; i < PASSES; ++i) {
                        const int row = mw + mi * 32 + rr0 + i * RPP;
                        rv[i] = (cok && row < p.M) ? *reinterpret_cast<const u32x4*>(p.resid + (size_t)row * p.Cout + col)
                                                   : u32x4{0u, 0u, 0u, 0u};
                    }
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[(row_l + (r & 3) + 8 * (r >> 2)) * SROW + ni * 32 + col_l] = acc[mi][ni][r];
                // (one wave: its LDS operations execute in order, the reads below see the writes above)
#pragma unroll
                for (int i = 0; i < PASSES; ++i) {
                    const int rr = rr0 + i * RPP;
                    const int row = mw + mi * 32 + rr;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + rr * SROW + piece * 8);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + rr * SROW + piece * 8 + 4);
                    v0 = v0 * sc0 + sh0;
                    v1 = v1 * sc1 + sh1;
                    if (p.act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v0[q] = v0[q] > 0.f ? v0[q] : 0.1f * v0[q];
                            v1[q] = v1[q] > 0.f ? v1[q] : 0.1f * v1[q];
                        }
                    }
                    if (p.resid) {
                        v0[0] += __uint_as_float(rv[i][0] << 16);
                        v0[1] += __uint_as_float(rv[i][0] & 0xFFFF0000u);
                        v0[2] += __uint_as_float(rv[i][1] << 16);
                        v0[3] += __uint_as_float(rv[i][1] & 0xFFFF0000u);
                        v1[0] += __uint_as_float(rv[i][2] << 16);
                        v1[1] += __uint_as_float(rv[i][2] & 0xFFFF0000u);
                        v1[2] += __uint_as_float(rv[i][3] << 16);
                        v1[3] += __uint_as_float(rv[i][3] & 0xFFFF0000u);
                    }
                    if (cok && row < p.M) {
                        u32x4 pk;
                        pk[0] = (unsigned)f32_to_bf16(v0[0]) | ((unsigned)f32_to_bf16(v0[1]) << 16);
                        pk[1] = (unsigned)f32_to_bf16(v0[2]) | ((unsigned)f32_to_bf16(v0[3]) << 16);
                        pk[2] = (unsigned)f32_to_bf16(v1[0]) | ((unsigned)f32_to_bf16(v1[1]) << 16);
                        pk[3] = (unsigned)f32_to_bf16(v1[2]) | ((unsigned)f32_to_bf16(v1[3]) << 16);
                        *reinterpret_cast<u32x4*>(static_cast<bf16_t*>(p.y) + (size_t)row * p.Cout + col) = pk;
                    }
                }
            }
        } else {
            constexpr int P = WTN / 4, RPP = 64 / P, PASSES = 32 / RPP;
            const int piece = lane % P, rr0 = lane / P;
            const int col = nw + piece * 4;
            const bool cok = col < p.Cout;
            const bool full = col + 3 < p.Cout;          // (cok && !full: the quad that crosses an odd Cout)
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (full) {
                sc = *reinterpret_cast<const f32x4_u*>(p.scale + col);
                sh = *reinterpret_cast<const f32x4_u*>(p.shift + col);
            } else if (cok) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col + q < p.Cout) {
                        sc[q] = p.scale[col + q];
                        sh[q] = p.shift[col + q];
                    }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[(row_l + (r & 3) + 8 * (r >> 2)) * SROW + ni * 32 + col_l] = acc[mi][ni][r];
#pragma unroll
                for (int i = 0; i < PASSES; ++i) {
                    const int rr = rr0 + i * RPP;
                    const int row = mw + mi * 32 + rr;
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + rr * SROW + piece * 4);
                    v = v * sc + sh;
                    if (p.act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
                    }
                    if (cok && row < p.M) {
                        float* yp = static_cast<float*>(p.y) + (size_t)row * p.Cout + col;
                        if (full) *reinterpret_cast<f32x4_u*>(yp) = v;
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < p.Cout) yp[q] = v[q];
                        }
                    }
                }
            }
        }
    };

    // ---- the stream of items: one clean K-loop per tile, the ring state carried across tiles ---------------------------------
    set_tile(t0);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s);
    int st = 0, st_ld = NSTAGE - 1;
    for (int tile = t0; tile < t1; ++tile) {
        for (int ks = 0; ks < kchunks; ++ks) {
            // this item has landed for this wave: NSTAGE - 2 younger items may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * DPW) : "memory");
            __builtin_amdgcn_s_barrier();       // ... for every wave, and nobody still reads the stage of the item before
            issue(st_ld);
            compute(st);
            st = st + 1 == NSTAGE ? 0 : st + 1;
            st_ld = st_ld + 1 == NSTAGE ? 0 : st_ld + 1;
        }
        epilogue(tile);
        // the epilogue's loads and stores sit in the same counter as the DMAs: drain once per tile (the DMAs of the next
        // tile's first items were issued before the epilogue and have had its whole length to land), so that the counted
        // wait above never depends on loads and stores retiring in order relative to each other
#ifndef R_NODRAIN
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0); the builtin, not inline asm: hipcc's own wait-count pass must see it
#endif
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the last, dead DMAs write zeros: let them land)
}

// Tile shapes (letter = experiments override Y3_BF16R_TILE): a = 128x32, b = 128x64, c = 128x128, d = 192x128,
// e = 96x128, f = 64x64, g = 64x128.
struct RTile { int bm, bn, lds; };
template <int MI, int NI, int WM, int WN, int NSTAGE>
constexpr RTile rtile() {
    return RTile{WM * MI * 32, WN * NI * 32, NSTAGE * (WM * MI * 32 + WN * NI * 32) * 128 + 4 * 32 * (NI * 32 + 4) * 4};
}

template <int MI, int NI, int WM, int WN, int NSTAGE, bool UPCAT>
int launch_r(hipStream_t stream, const ConvArgsR& a) {
    auto kern = conv1x1_bf16r_kernel<MI, NI, WM, WN, NSTAGE, UPCAT>;
    constexpr RTile t = rtile<MI, NI, WM, WN, NSTAGE>();
    static bool attr_set[Y3_MAX_DEVICES] = {};     // per instantiation; benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, t.lds));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int nbm = (a.M + t.bm - 1) / t.bm, nbn = (a.Cout + t.bn - 1) / t.bn;
    const long long ntiles = (long long)nbm * nbn;
    const int per_cu = 163840 / t.lds < 1 ? 1 : (163840 / t.lds > 4 ? 4 : 163840 / t.lds);
    const long long slots = 256LL * per_cu;
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), t.lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

int forced_rtile() {
    static int v = -2;
    if (v == -2) {
        const char* e = y3_exp_env("Y3_BF16R_TILE");
        v = (e && e[0] >= 'a' && e[0] <= 'g') ? e[0] - 'a' : -1;
    }
    return v;
}

#ifndef R_STAGES
#define R_STAGES 3
#endif
constexpr int RS = R_STAGES;       // ring depth of the tiles whose LDS leaves room for more than three stages (probe builds)

// Tile rule, from the per-layer measurements with every tile forced in turn at bs = 16 and bs = 8, 608x608
// (profiles/r05_bf16_tiles.txt): the largest of 192x128 / 96x128 / 64x64 that still gives ~200 tiles (a tile per CU: the
// workgroups are resident, one or two per CU by their LDS), Cout <= 64 layers on 128x32 / 128x64.
template <bool UPCAT>
int dispatch_r(hipStream_t stream, const ConvArgsR& a) {
    constexpr RTile T[7] = {rtile<1, 1, 4, 1, RS>(), rtile<2, 1, 2, 2, RS>(), rtile<2, 2, 2, 2, 3>(), rtile<3, 2, 2, 2, 3>(),
                            rtile<3, 1, 1, 4, RS>(), rtile<1, 1, 2, 2, RS>(), rtile<1, 2, 2, 2, RS>()};
    int t = forced_rtile();
    if (t < 0 || (T[t].bn > 32 && a.Cout <= 32) || (T[t].bn > 64 && a.Cout <= 64)) {
        auto tiles = [&](int c) { return (long long)((a.M + T[c].bm - 1) / T[c].bm) * ((a.Cout + T[c].bn - 1) / T[c].bn); };
        if (a.Cout <= 32) t = 0;
        else if (a.Cout <= 64) t = 1;
        else if (tiles(3) >= 200) t = 3;
        else if (tiles(4) >= 200) t = 4;
        else t = 5;
    }
    switch (t) {
    case 0: return launch_r<1, 1, 4, 1, RS, UPCAT>(stream, a);
    case 1: return launch_r<2, 1, 2, 2, RS, UPCAT>(stream, a);
    case 2: return launch_r<2, 2, 2, 2, 3, UPCAT>(stream, a);
    case 3: return launch_r<3, 2, 2, 2, 3, UPCAT>(stream, a);
    case 4: return launch_r<3, 1, 1, 4, RS, UPCAT>(stream, a);
    case 5: return launch_r<1, 1, 2, 2, RS, UPCAT>(stream, a);
    default: return launch_r<1, 2, 2, 2, RS, UPCAT>(stream, a);
    }
}

}  // namespace

// 1x1 convs with Cin >= 512 (a multiple of 64): decided by (k, Cin) alone, because the weight packing is chosen when only
// the kernel's shape is known.  Measured per layer at configs[4] (profiles/r05_bf16_tiles.txt, us per launch, this kernel |
// the register-staged kernel of y3_conv_bf16.hip): 38-grid 512->256 19.4 | 23.3, 19-grid 1024->512 17.9 | 21.7, 768->256 with
// the fused upsample 25.7 | 28.8, detection convs 1024->255 16.1 | 18.7, 512->255 21.5 | 26.9 - deep K, few rows: what the
// ring is for.  The wide, shallow layers (76-grid 256->128 26.8 | 23.5, 152-grid 128->64 42 | 33) run at the rate the
// memory system gives a mixed read / write stream either way and stay on the register-staged kernel, whose three
// co-resident workgroups per CU cover one another's epilogues.  (experiments build: Y3_BF16R=0 / 1 = never / every Cin % 64 == 0)
int y3_conv_bf16r_takes(int k, int cin) {
    if (k != 1 || cin % 64 != 0) return 0;
    static int mode = -2;
    if (mode == -2) {
        const char* e = y3_exp_env("Y3_BF16R");
        mode = e ? (e[0] == '0' ? 0 : 1) : -1;
    }
    if (mode >= 0) return mode;
    return cin >= 512 ? 1 : 0;
}
// the tile letter dispatch_r picks for this conv (host-only, see y3_conv_bf16_tile)
int y3_conv_bf16r_tile(const y3_conv_desc* d) {
    const long long M = (long long)d->n * d->h * d->w;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((d->cout + bn - 1) / bn); };
    if (d->cout <= 32) return 'a';
    if (d->cout <= 64) return 'b';
    if (tiles(192, 128) >= 200) return 'd';
    if (tiles(96, 128) >= 200) return 'e';
    return 'f';
}

int y3_launch_conv_bf16r(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                         const float* scale, const float* shift, const void* residual, void* y, int out_f32) {
    const long long M = (long long)d->n * d->h * d->w;
    ConvArgsR a;
    a.x = static_cast<const bf16_t*>(x); a.xu = static_cast<const bf16_t*>(x_up);
    a.w = static_cast<const bf16_t*>(w); a.scale = scale; a.shift = shift;
    a.resid = static_cast<const bf16_t*>(residual); a.y = y;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cu = d->c_up; a.Cx = d->cin - d->c_up; a.Cout = d->cout;
    a.act = d->act; a.out_f32 = out_f32; a.M = (int)M;
    if (x_up) {
        Y3_CHECK_ARG(d->c_up % 64 == 0 && d->c_up < d->cin && d->h % 2 == 0 && d->w % 2 == 0,
                     "y3_conv2d_fwd_bf16: bad fused upsample+concat configuration (c_up must be a multiple of 64)");
        return dispatch_r<true>(stream, a);
    }
    return dispatch_r<false>(stream, a);
}
