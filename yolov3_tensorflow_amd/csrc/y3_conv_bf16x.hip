// bf16-storage conv, second generation (round 3): global -> LDS by LDS-DMA (`buffer_load ... lds`, 16 bytes per lane),
// 64-element K-steps, 256x256 / 256x128 tiles for eight waves or 128x128 tiles two per CU.  Same contract and the same
// arithmetic as y3_conv_bf16.hip (fp32 accumulate in v_mfma_f32_32x32x16_bf16, fp32 epilogue, ONE rounding to bf16), which
// stays as the fall-back for the shapes this kernel does not take (utils/layer_utils.py:9-22 is the reference code both
// replace).
//
// Why: the first kernel stages every operand through registers (buffer_load -> VGPR -> ds_write) on 128x128 tiles with
// 32-element K-steps: 16 KB of loads and a barrier per 8 MFMAs x 32 cycles, 590-650 TF/s on the 3x3 convs = 0.25 of the
// 2.5 PF bf16 peak (profiles/r02_bench_c5_bf16_608.json).  Here a K-step is 64 elements (whole 128-byte lines per row),
// the loads never touch a VGPR or the VALU, and a 256x256 tile does 32 MFMAs per wave and K-step.
//   * LDS image of a stage: [BM + BN rows][128 B]; one LDS-DMA instruction fills 8 rows (lane l -> row l/8, 16-byte slot
//     l%8: the destination is wave-uniform base + lane*16, so the image is lane-linear) and the XOR swizzle that makes
//     the ds_read_b128 fragment reads conflict-free sits on the SOURCE side: slot s of row r holds chunk s ^ ((r>>1)&7)
//     (the 8 lanes of a row still cover the row's whole 128-byte line);
//   * padding, ragged rows and Cout tails are out-of-range buffer offsets: LDS-DMA writes zeros for them
//     (tools/glds_probe.hip, measured on gfx950);
//   * two LDS stages, one barrier per K-step: wait for stage g (vmcnt(0)) | barrier | issue the loads of stage g+1 into
//     the other buffer | 4 x (fragment reads, MFMAs) of stage g.
#include <cstdlib>
#include <type_traits>
#include "y3_internal.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

// 16 bytes per lane, global -> LDS without a register round trip: lane l's bytes land at lds_base + 16*l (lds_base is
// wave-uniform), an out-of-range `voff` writes zeros.  (The builtin exists in the device pass only; hipcc's host pass
// instantiates the kernel templates too and silently drops their launch stubs if it meets it there.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_base, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_base, 16, voff, soff, 0, 0);
#endif
}

struct ConvArgsX {
    const bf16_t* x;     // [N,H,W,Cx]
    const bf16_t* xu;    // [N,H/2,W/2,Cu] or nullptr
    const bf16_t* w;     // packed [taps][Cin/BK][Cout][BK]
    const float* scale;
    const float* shift;
    const bf16_t* resid; // [M,Cout] or nullptr
    void* y;             // [M,Cout] bf16, or fp32 when out_f32
    int N, H, W, Cin, Cu, Cx;
    int Ho, Wo, Cout;
    int stride, pad, act, out_f32;
    int M;
};

constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round to nearest even (finite inputs)
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <int BK>
struct Swz {
    // rows of BK bf16 = 128 B (8 slots) or 64 B (4 slots); the slot of a row is flipped so that the 16 rows a
    // ds_read_b128 lane group reads (one logical slot each) fall on 16 different 16-byte bank groups
    static constexpr int ROWB = BK * 2;
    static constexpr int SLOTS = ROWB / 16;
    __device__ static __forceinline__ int f(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};

// ---- epilogue shared by the kernels of this file (as in y3_conv_bf16.hip): fp32 scale/shift, LeakyReLU, + residual, one
// rounding to bf16; 64 output rows at a time through an fp32 staging tile in the (now idle) tile LDS ----------------------
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_x(const ConvArgsX& p, unsigned char* smem,
                                           f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int m0, int n0, int wm, int wn) {
    constexpr int NT = 64 * WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
    constexpr int LDC = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    if ((p.Cout & 3) != 0 && (!p.out_f32 || p.resid)) {
        // Cout % 4 != 0 with bf16 output or a residual (no layer of the network): scalar stores.  The detection convs
        // (3*(5+C) channels, fp32 output) take the staged path below with 4-byte-aligned dwordx4 stores.
        float* yf = static_cast<float*>(p.y);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * WTN + ni * 32 + col_l;
            const bool cok = col < p.Cout;
            const float sc = cok ? p.scale[col] : 0.f, sh = cok ? p.shift[col] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WTM + mi * 32 + row_l + (r & 3) + 8 * (r >> 2);
                    if (cok && row < p.M) {
                        float v = acc[mi][ni][r] * sc + sh;
                        if (p.act) v = v > 0.f ? v : 0.1f * v;
                        const size_t o = (size_t)row * p.Cout + col;
                        if (p.resid) v += __uint_as_float((unsigned)p.resid[o] << 16);
                        if (p.out_f32) yf[o] = v;
                        else static_cast<bf16_t*>(p.y)[o] = f32_to_bf16(v);
                    }
                }
        }
        return;
    }
    float* cs = reinterpret_cast<float*>(smem);
    if (!p.out_f32 && (p.Cout & 7) == 0) {
        // bf16 output rows of a multiple of 8 channels (every BN layer of the network): 8 channels = 16 bytes per lane for
        // the residual load and the store.  The tail of a tile is bound by the number of store instructions a CU can issue
        // (~7 B per cycle and CU with 8-byte stores, MI355X guide T21): 16-byte pieces halve it.
        constexpr int C8 = BN / 8, RPP8 = NT / C8, PASSES8 = 64 / RPP8, HALVES = BM / 64;
        static_assert(RPP8 >= 1 && PASSES8 >= 1 && RPP8 * C8 == NT, "epilogue pass geometry (8-channel pieces)");
        const int tc = (tid % C8) * 8, tr = tid / C8;
        const int col = n0 + tc;
        const bool cok = col < p.Cout;               // (Cout % 8 == 0: the whole piece is in range)
        f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
        if (cok) {
            sc0 = *reinterpret_cast<const f32x4*>(p.scale + col);
            sc1 = *reinterpret_cast<const f32x4*>(p.scale + col + 4);
            sh0 = *reinterpret_cast<const f32x4*>(p.shift + col);
            sh1 = *reinterpret_cast<const f32x4*>(p.shift + col + 4);
        }
        // Memory order of the tail (round 6; the same finding as csrc/y3_conv_wino44.hip's tail).  On gfx950 loads and stores share
        // vmcnt and may complete out of order with each other, so hipcc waits vmcnt(0) for a LOAD whenever a STORE is pending: with
        // the residual loads of a 64-row pass issued behind the stores of the pass before, every store of the tile waited for the
        // one before it (six stores, five store round trips and three load latencies in a row per 192 x 128 tile: the 8-13 us per
        // round of tiles that nothing hid, profiles/r05_bf16_tiles.txt).  Now: buffer accesses relative to the tile's first row
        // (rows past M and columns past Cout are out-of-range offsets: no branches around memory instructions), the residual
        // pieces of pass h + 1 are requested BEFORE the stores of pass h, and the one wait per pass sits behind the staging
        // barrier, where the stores of the pass before have had the whole staging time to complete.
        const int rows_left = p.M - m0 < BM ? p.M - m0 : BM;
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
            static_cast<bf16_t*>(p.y) + (size_t)m0 * p.Cout, 0, (unsigned)((size_t)rows_left * p.Cout * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(p.resid ? p.resid + (size_t)m0 * p.Cout : static_cast<const bf16_t*>(p.y)), 0,
            p.resid ? (unsigned)((size_t)rows_left * p.Cout * 2) : 0u, 0x00020000);
        auto piece = [&](int half, int i) -> unsigned {      // byte offset of (row 64 * half + tr + i * RPP8, column col) in the tile's rows
            return cok ? (unsigned)(((64 * half + tr + i * RPP8) * p.Cout + col) * 2) : 0x80000000u;
        };
        u32x4 rv[2][PASSES8];
        if (p.resid) {
#pragma unroll
            for (int i = 0; i < PASSES8; ++i)
                rv[0][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, piece(0, i), 0, 0));
        }
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int rbase = wm * WTM + mi * 32;
                if (rbase / 64 == half) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            cs[(rbase - 64 * half + row_l + (r & 3) + 8 * (r >> 2)) * LDC + wn * WTN + ni * 32 + col_l] =
                                acc[mi][ni][r];
                }
            }
            __syncthreads();
            if (p.resid) {
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), visible to hipcc: this pass's residual pieces (and the stores before)
                if (half + 1 < HALVES) {
#pragma unroll
                    for (int i = 0; i < PASSES8; ++i)
                        rv[(half + 1) & 1][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, piece(half + 1, i), 0, 0));
                }
            }
#pragma unroll
            for (int i = 0; i < PASSES8; ++i) {
                const int rr = tr + i * RPP8;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(cs + rr * LDC + tc);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(cs + rr * LDC + tc + 4);
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
                    const u32x4 r4 = rv[half & 1][i];
                    v0[0] += __uint_as_float(r4[0] << 16);
                    v0[1] += __uint_as_float(r4[0] & 0xFFFF0000u);
                    v0[2] += __uint_as_float(r4[1] << 16);
                    v0[3] += __uint_as_float(r4[1] & 0xFFFF0000u);
                    v1[0] += __uint_as_float(r4[2] << 16);
                    v1[1] += __uint_as_float(r4[2] & 0xFFFF0000u);
                    v1[2] += __uint_as_float(r4[3] << 16);
                    v1[3] += __uint_as_float(r4[3] & 0xFFFF0000u);
                }
                u32x4 pk;
                pk[0] = (unsigned)f32_to_bf16(v0[0]) | ((unsigned)f32_to_bf16(v0[1]) << 16);
                pk[1] = (unsigned)f32_to_bf16(v0[2]) | ((unsigned)f32_to_bf16(v0[3]) << 16);
                pk[2] = (unsigned)f32_to_bf16(v1[0]) | ((unsigned)f32_to_bf16(v1[1]) << 16);
                pk[3] = (unsigned)f32_to_bf16(v1[2]) | ((unsigned)f32_to_bf16(v1[3]) << 16);
                __builtin_amdgcn_raw_buffer_store_b128(pk, rs_y, piece(half, i), 0, 0);
            }
            __syncthreads();
        }
        return;
    }
    constexpr int C4 = BN / 4, RPP = NT / C4, PASSES = 64 / RPP;
    static_assert(RPP >= 1 && PASSES >= 1 && RPP * C4 == NT, "epilogue pass geometry");
    const int tc = (tid % C4) * 4, tr = tid / C4;
    const int col = n0 + tc;
    const bool cok = col < p.Cout;
    const bool full = col + 3 < p.Cout;          // (cok && !full: the column quad that crosses an odd Cout)
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (full) {                                  // (one dwordx4 each; no alignment assumed)
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
    for (int half = 0; half < BM / 64; ++half) {
        // residual rows of this pass first: their latency overlaps the staging
        u32x2 rv[PASSES];
        if (p.resid) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int row = m0 + 64 * half + tr + i * RPP;
                rv[i] = (cok && row < p.M) ? *reinterpret_cast<const u32x2*>(p.resid + (size_t)row * p.Cout + col)
                                           : u32x2{0u, 0u};
            }
        }
        // the waves whose rows fall in [64*half, 64*half+64) stage their accumulators (fp32)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rbase = wm * WTM + mi * 32;
            if (rbase / 64 == half) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        cs[(rbase - 64 * half + row_l + (r & 3) + 8 * (r >> 2)) * LDC + wn * WTN + ni * 32 + col_l] =
                            acc[mi][ni][r];
            }
        }
        __syncthreads();
        if (cok) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int rr = tr + i * RPP;
                const int row = m0 + 64 * half + rr;
                if (row < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(cs + rr * LDC + tc);
                    v = v * sc + sh;
                    if (p.act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
                    }
                    const size_t o = (size_t)row * p.Cout + col;
                    if (p.resid) {
                        v[0] += __uint_as_float(rv[i][0] << 16);
                        v[1] += __uint_as_float(rv[i][0] & 0xFFFF0000u);
                        v[2] += __uint_as_float(rv[i][1] << 16);
                        v[3] += __uint_as_float(rv[i][1] & 0xFFFF0000u);
                    }
                    if (p.out_f32) {
                        float* yp = static_cast<float*>(p.y) + o;
                        if (full) *reinterpret_cast<f32x4_u*>(yp) = v;
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < p.Cout) yp[q] = v[q];
                        }
                    } else {
                        u32x2 pk;
                        pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(p.y) + o) = pk;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int BM, int BN, int WM, int WN, int BK, int KS, bool UPCAT>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_bf16x_kernel(const ConvArgsX p) {
    constexpr int NW = WM * WN;
    constexpr int ROWB = BK * 2, LPR = ROWB / 16, RPI = 64 / LPR;     // row bytes, lanes per row, rows per DMA instruction
    constexpr int ACH = BM / RPI / NW, BCH = BN / RPI / NW;           // DMA instructions per thread and K-step
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
    constexpr int LDC = BN + 4;
    static_assert(ACH >= 1 && BCH >= 1 && ACH * RPI * NW == BM && BCH * RPI * NW == BN, "tile / wave count mismatch");
    static_assert(MI >= 1 && NI >= 1, "wave tile must hold a 32x32 MFMA tile");
    static_assert((size_t)64 * LDC * 4 <= (size_t)2 * STAGE, "epilogue staging must fit in the tile LDS");
    using SW = Swz<BK>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nbm = (p.M + BM - 1) / BM;
    const int kchunks = p.Cin / BK;
    const int S = KS * KS * kchunks;

    // XCD-contiguous, column-major tile id (see y3_conv.hip)
    const int nt = gridDim.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = blockIdx.x & 7, kk8 = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk8;
    const int bn = tile / nbm, bm = tile - bn * nbm;
    const int m0 = bm * BM, n0 = bn * BN;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.w), 0, (unsigned)((size_t)KS * KS * p.Cout * p.Cin * 2), 0x00020000);
    const unsigned bytes_x = (unsigned)((size_t)p.N * p.H * p.W * p.Cx * 2);
    const unsigned bytes_u = UPCAT ? (unsigned)((size_t)p.N * (p.H >> 1) * (p.W >> 1) * p.Cu * 2) : 0u;

    // ---- per-thread DMA rows: instruction q = wave*CH + j fills rows q*RPI .. q*RPI + RPI-1, lane -> (row l/LPR, slot l%LPR)
    const int l_row = lane / LPR, l_slot = lane % LPR;
    int a_base[ACH], a_msk[ACH], a_base_u[UPCAT ? ACH : 1];
    unsigned b_voff[BCH];
    {
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int r = (wave * ACH + j) * RPI + l_row;
            const int chunk = l_slot ^ SW::f(r);
            const int m = m0 + r;
            int mk = 0, base = 0, base_u = 0;
            if (m < p.M) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    if ((unsigned)(iy0 + t) < (unsigned)p.H) mk |= 1 << t;
                    if ((unsigned)(ix0 + t) < (unsigned)p.W) mk |= 1 << (4 + t);
                }
                base = ((n * p.H + iy0) * p.W + ix0) * p.Cx + chunk * 8;
                if (UPCAT) base_u = ((n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cu + chunk * 8;
            }
            a_msk[j] = mk; a_base[j] = base;
            if (UPCAT) a_base_u[j] = base_u;
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int r = (wave * BCH + j) * RPI + l_row;
            const int chunk = l_slot ^ SW::f(r);
            const int co = n0 + r;
            b_voff[j] = co < p.Cout ? (unsigned)(co * BK + chunk * 8) * 2u : OOB;
        }
    }
    int ld_tap = 0, ld_cc = 0;
    // LDS-DMA of the prepared K-step into stage `buf`, in KSUB parts (one behind each 16-element slice of the running
    // K-step's MFMAs: ~20 scalar / vector instructions per part sit in the shadow of 8 MFMAs; issued in one piece they
    // were an 80-instruction block between two MFMA groups).  Branch-free: past the last K-step (`live` false) every
    // lane's offset is out of range, so the loads write zeros into a stage nobody reads any more.
    constexpr int KSUB = BK / 16;
    struct Step { int ky, kx, tap_off; bool from_up, live; unsigned soff, wsoff; __amdgpu_buffer_rsrc_t rs_a; };
    auto issue_begin = [&](bool live) {
        Step t;
        t.live = live;
        t.ky = (KS == 1) ? 0 : ld_tap / KS;
        t.kx = (KS == 1) ? 0 : ld_tap - t.ky * KS;
        t.tap_off = (t.ky * p.W + t.kx) * p.Cx;
        const int c0 = ld_cc * BK;
        t.from_up = UPCAT && c0 < p.Cu;
        t.soff = live ? (unsigned)(t.from_up ? c0 : c0 - (UPCAT ? p.Cu : 0)) * 2u : 0u;
        t.rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(t.from_up ? p.xu : p.x), 0,
                                                   t.from_up ? bytes_u : bytes_x, 0x00020000);
        // (the scalar offset is not range-checked: a dead K-step must not carry one)
        t.wsoff = live ? (unsigned)((ld_tap * kchunks + ld_cc) * p.Cout) * (unsigned)(BK * 2) : 0u;
        const bool wrap = ++ld_cc == kchunks;
        ld_cc = wrap ? 0 : ld_cc;
        ld_tap += wrap ? 1 : 0;
        return t;
    };
    auto issue_part = [&](const Step& t, int buf, int part) {
        unsigned char* as = smem + buf * STAGE + (wave * ACH) * (RPI * ROWB);
        unsigned char* bs = smem + buf * STAGE + BM * ROWB + (wave * BCH) * (RPI * ROWB);
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            if (j * KSUB / ACH != part) continue;
            const bool ok = t.live && ((a_msk[j] >> t.ky) & (a_msk[j] >> (4 + t.kx)) & 1) != 0;
            unsigned voff = ok ? (unsigned)(a_base[j] + t.tap_off) * 2u : OOB;
            if (UPCAT) voff = t.from_up ? (ok ? (unsigned)a_base_u[j] * 2u : OOB) : voff;
            dma16(t.rs_a, as + j * (RPI * ROWB), voff, t.soff);
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            if (j * KSUB / BCH != part) continue;
            dma16(rs_w, bs + j * (RPI * ROWB), t.live ? b_voff[j] : OOB, t.wsoff);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // lane l feeds row l&31 of a 32x32 tile with the 8 k-values of slot 2*kk + (l>>5); tile row bases are multiples of
    // 32, so the swizzle term depends on the lane only
    const int frag_row = lane & 31, frag_half = lane >> 5, frag_f = SW::f(frag_row);
    auto compute = [&](int buf, int kk0, int kk1) {
        const unsigned char* as = smem + buf * STAGE + (wm * WTM + frag_row) * ROWB;
        const unsigned char* bs = smem + buf * STAGE + BM * ROWB + (wn * WTN + frag_row) * ROWB;
#pragma unroll
        for (int kk = kk0; kk < kk1; ++kk) {
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

    {
        const Step t = issue_begin(true);
#pragma unroll
        for (int part = 0; part < KSUB; ++part) issue_part(t, 0, part);
    }
    for (int g = 0; g < S; ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of stage g has landed ...
        __syncthreads();                                     // ... everyone's has, and nobody still reads the other stage
        const Step t = issue_begin(g + 1 < S);               // next stage: in flight under this K-step's MFMAs
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
            compute(g & 1, kk, kk + 1);
            issue_part(t, (g + 1) & 1, kk);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the last, dead DMA writes zeros: let it land)
    __syncthreads();                                         // the staging below reuses the tile LDS

    epilogue_x<BM, BN, WM, WN>(p, smem, acc, m0, n0, wm, wn);
}

// ---- the pipelined kernel: 256-row tiles, eight waves, operands of a whole K-tile held in registers ------------------
// The two-stage kernel above drains its DMA queue and crosses a barrier every K-step; measured, that structure gives the
// same ~600 TF/s as the register-staged kernel it replaced (profiles/r03_bf16x_tiles.txt) - the ceiling the CDNA guide
// reports for every "wait for the stage, barrier, compute" loop.  This one never drains:
//   * a K-tile (64 elements of K) is staged as four half-tiles - A rows [0,BM/2), A rows [BM/2,BM), B rows [0,BN/2),
//     B rows [BN/2,BN) - into one of two LDS K-tile buffers; the DMA of a half-tile is issued 1 - 2 K-tiles before its
//     first reader and the only wait in the loop is a COUNTED one at the end of a K-tile (`s_waitcnt vmcnt(BCH)`: the two
//     B half-tiles issued last stay in flight across the barrier);
//   * a wave reads its whole B operand of K-tile t (NI column tiles x 4 K-slices) into registers in the first phase and
//     one 32-row tile of A per phase (double-buffered: the reads of phase p+1 fly under the MFMAs of phase p), so the B
//     half-tiles of a buffer are free again after phase 0 and can be re-staged for K-tile t+2 while K-tile t is still
//     being computed;
//   * schedule of K-tile t (MI phases of 4*NI MFMAs per wave; buffer b = t & 1):
//         phase 0 .. MI/2-1   : DMA A half-tile(s) of t+1 -> buffer 1-b  (free since the barrier that ended K-tile t-1)
//         barrier X           : every wave holds B(t) in registers
//         phase MI/2 .. MI-1  : DMA B half-tile(s) of t+2 -> buffer b
//         s_waitcnt vmcnt(BCH); lgkmcnt(0); barrier Y : A(t+1) has landed for every wave (B(t+1) landed a K-tile ago)
//     two raw s_barriers per K-tile and no vmcnt(0) anywhere in the loop; hipcc's __syncthreads() would drain the queue.
template <int MI, int NI, int WM, int WN, int KS, bool UPCAT>
__global__ void __launch_bounds__(512, 2) conv_bf16p_kernel(const ConvArgsX p) {
    constexpr int BK = 64, ROWB = 128, RPI = 8, NW = 8;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int ACH = BM / RPI / NW, BCH = BN / RPI / NW;           // DMA instructions per thread and K-tile (4; 4 or 2)
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int WTM = MI * 32, WTN = NI * 32;
    constexpr int LDC = BN + 4;
    static_assert(WM * WN == NW && MI >= 2 && MI <= 4, "eight waves, two to four 32-row tiles per wave");
    // phases [0, XP) of a K-tile carry the A DMA of the next K-tile, phases [XP, MI) the B DMA of the one after (MI = 3, the
    // 192-row tiles: one phase of A, two of B)
    constexpr int XP = MI / 2, PA = XP, PB = MI - XP;
    static_assert(ACH >= PA && BCH >= PB, "every DMA phase must carry at least one instruction");
    static_assert((size_t)64 * LDC * 4 <= (size_t)2 * STAGE, "epilogue staging must fit in the tile LDS");
    using SW = Swz<BK>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nbm = (p.M + BM - 1) / BM;
    const int kchunks = p.Cin / BK;
    const int S = KS * KS * kchunks;

    const int nt = gridDim.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = blockIdx.x & 7, kk8 = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk8;
    const int bn = tile / nbm, bm = tile - bn * nbm;
    const int m0 = bm * BM, n0 = bn * BN;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.w), 0, (unsigned)((size_t)KS * KS * p.Cout * p.Cin * 2), 0x00020000);
    const unsigned bytes_x = (unsigned)((size_t)p.N * p.H * p.W * p.Cx * 2);
    const unsigned bytes_u = UPCAT ? (unsigned)((size_t)p.N * (p.H >> 1) * (p.W >> 1) * p.Cu * 2) : 0u;

    // DMA rows of this thread: instruction q = wave*CH + j fills rows 8q .. 8q+7 (lane -> row l/8, slot l%8)
    const int l_row = lane >> 3, l_slot = lane & 7;
    int a_base[ACH], a_msk[ACH], a_base_u[UPCAT ? ACH : 1];
    unsigned b_voff[BCH];
    {
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int r = (wave * ACH + j) * RPI + l_row;
            const int chunk = l_slot ^ SW::f(r);
            const int m = m0 + r;
            int mk = 0, base = 0, base_u = 0;
            if (m < p.M) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    if ((unsigned)(iy0 + t) < (unsigned)p.H) mk |= 1 << t;
                    if ((unsigned)(ix0 + t) < (unsigned)p.W) mk |= 1 << (4 + t);
                }
                base = ((n * p.H + iy0) * p.W + ix0) * p.Cx + chunk * 8;
                if (UPCAT) base_u = ((n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cu + chunk * 8;
            }
            a_msk[j] = mk; a_base[j] = base;
            if (UPCAT) a_base_u[j] = base_u;
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int r = (wave * BCH + j) * RPI + l_row;
            const int chunk = l_slot ^ SW::f(r);
            const int co = n0 + r;
            b_voff[j] = co < p.Cout ? (unsigned)(co * BK + chunk * 8) * 2u : OOB;
        }
    }
    // Two independent loader positions (A runs one K-tile ahead of the compute, B two): (tap, chunk) of the next K-tile
    // each will stage; past the last K-tile the DMA is "dead" (every lane out of range: zeros into a buffer nobody reads).
    struct Pos { int tap, cc, kt; };
    Pos pa = {0, 0, 0}, pb = {0, 0, 0};
    auto advance = [&](Pos& q) {
        const bool wrap = ++q.cc == kchunks;
        q.cc = wrap ? 0 : q.cc;
        q.tap += wrap ? 1 : 0;
        ++q.kt;
    };
    // part `part` of `parts` of the A tile of K-tile pa -> buffer `buf`
    auto dma_a = [&](int buf, int part, int parts) {
        const bool live = pa.kt < S;
        const int ky = (KS == 1) ? 0 : pa.tap / KS;
        const int kx = (KS == 1) ? 0 : pa.tap - ky * KS;
        const int tap_off = (ky * p.W + kx) * p.Cx;
        const int c0 = pa.cc * BK;
        const bool from_up = UPCAT && c0 < p.Cu;
        const unsigned soff = live ? (unsigned)(from_up ? c0 : c0 - (UPCAT ? p.Cu : 0)) * 2u : 0u;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(from_up ? p.xu : p.x), 0, from_up ? bytes_u : bytes_x, 0x00020000);
        unsigned char* as = smem + buf * STAGE + (wave * ACH) * (RPI * ROWB);
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            if (j * parts / ACH != part) continue;
            const bool ok = live && ((a_msk[j] >> ky) & (a_msk[j] >> (4 + kx)) & 1) != 0;
            unsigned voff = ok ? (unsigned)(a_base[j] + tap_off) * 2u : OOB;
            if (UPCAT) voff = from_up ? (ok ? (unsigned)a_base_u[j] * 2u : OOB) : voff;
            dma16(rs_a, as + j * (RPI * ROWB), voff, soff);
        }
    };
    auto dma_b = [&](int buf, int part, int parts) {
        const bool live = pb.kt < S;
        // (the scalar offset is not range-checked: a dead K-tile must not carry one)
        const unsigned wsoff = live ? (unsigned)((pb.tap * kchunks + pb.cc) * p.Cout) * (unsigned)(BK * 2) : 0u;
        unsigned char* bs = smem + buf * STAGE + BM * ROWB + (wave * BCH) * (RPI * ROWB);
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            if (j * parts / BCH != part) continue;
            dma16(rs_w, bs + j * (RPI * ROWB), live ? b_voff[j] : OOB, wsoff);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5, frag_f = SW::f(frag_row);
    int frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = ((2 * kk + frag_half) ^ frag_f) << 4;
    auto read_a = [&](int buf, int mi, u32x4 (&a)[4]) {
        const unsigned char* as = smem + buf * STAGE + (wm * WTM + mi * 32 + frag_row) * ROWB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = *reinterpret_cast<const u32x4*>(as + frag_off[kk]);
    };
    u32x4 breg[NI][4];
    auto read_b = [&](int buf) {
        const unsigned char* bs = smem + buf * STAGE + BM * ROWB + (wn * WTN + frag_row) * ROWB;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) breg[ni][kk] = *reinterpret_cast<const u32x4*>(bs + ni * 32 * ROWB + frag_off[kk]);
    };
    auto mfmas = [&](int mi, const u32x4 (&a)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kk]),
                                                                      __builtin_bit_cast(bf16x8, breg[ni][kk]),
                                                                      acc[mi][ni], 0, 0, 0);
    };

    // ---- prologue: B(0), A(0), B(1) ------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < PB; ++q) dma_b(0, q, PB);
    advance(pb);
#pragma unroll
    for (int q = 0; q < PA; ++q) dma_a(0, q, PA);
    advance(pa);
#pragma unroll
    for (int q = 0; q < PB; ++q) dma_b(1, q, PB);
    advance(pb);
    if (BCH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int t = 0; t < S; ++t) {
        const int b = t & 1;
        u32x4 a0[4], a1[4];
        read_b(b);
        read_a(b, 0, a0);
#pragma unroll
        for (int ph = 0; ph < MI; ++ph) {
            if (ph == XP) {
                // every wave has B(t) in registers: its half-tiles may be re-staged
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (ph < XP) dma_a(b ^ 1, ph, PA);          // A(t+1) -> the other buffer
            else dma_b(b, ph - XP, PB);                 // B(t+2) -> this buffer
            if (ph + 1 < MI) {
                if (ph & 1) read_a(b, ph + 1, a0);
                else read_a(b, ph + 1, a1);
            }
            if (ph & 1) mfmas(ph, a1);
            else mfmas(ph, a0);
        }
        advance(pa);
        advance(pb);
        // A(t+1) has landed for this wave (the B(t+2) DMA issued last stays in flight) ...
        if (BCH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // ... and for every wave; nobody still reads buffer b's A tile
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the last, dead DMAs write zeros: let them land)
    __syncthreads();                                     // the staging below reuses the tile LDS

    epilogue_x<BM, BN, WM, WN>(p, smem, acc, m0, n0, wm, wn);
}

__global__ void pack_weights_bf16x_kernel(const float* __restrict__ w_hwio, bf16_t* __restrict__ w_packed, int taps,
                                          int cin, int cout, int bk) {
    const size_t total = (size_t)taps * cin * cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const size_t r = i / cin;
        const int co = (int)(r % cout);
        const int t = (int)(r / cout);
        // [tap][Cin/bk][Cout][bk]: the B tile of one K-step is contiguous
        w_packed[((((size_t)t * (cin / bk) + ci / bk) * cout + co) * bk) + (ci % bk)] =
            f32_to_bf16(w_hwio[((size_t)t * cin + ci) * cout + co]);
    }
}

template <int BM, int BN, int WM, int WN, int BK, int KS, bool UPCAT>
int launch_x(hipStream_t stream, const ConvArgsX& a) {
    auto kern = conv_bf16x_kernel<BM, BN, WM, WN, BK, KS, UPCAT>;
    constexpr size_t lds = (size_t)2 * (BM + BN) * BK * 2;
    static bool attr_set[Y3_MAX_DEVICES] = {};     // per instantiation; benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(nbm * nbn), dim3(64 * WM * WN), lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

template <int MI, int NI, int WM, int WN, int KS, bool UPCAT>
int launch_p(hipStream_t stream, const ConvArgsX& a) {
    auto kern = conv_bf16p_kernel<MI, NI, WM, WN, KS, UPCAT>;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 128;
    static bool attr_set[Y3_MAX_DEVICES] = {};     // per instantiation; benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(nbm * nbn), dim3(512), lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

// Tile choice.  Candidates: A = 256x256, D = 192x256, B = 256x128, E = 192x128 (the pipelined kernel: eight waves, one
// workgroup per CU), C = 128x128 (two-stage kernel, four waves, two workgroups per CU); narrow Cout: 128x64 / 128x32.
// At configs[4] (bs = 16, 608x608) a layer is 54 GFLOP - 22 us at the bf16 peak - and what decides is how the tile count
// quantises over the 256 CUs and what a tile costs besides its K-tiles.  Fitted to the per-layer measurements with every
// tile forced in turn (profiles/r05_bf16_tiles.txt; S = 9 * Cin / 64 K-tiles per output tile):
//     time = ceil(tiles / slots) * (S * kt + fixed)
//   A 256x256: kt 1.58 us, fixed 12.6 | D 192x256: 1.48, 8.9 | B 256x128: 0.965, 7.7  (one workgroup per CU; fixed = launch
//   ramp, first DMA latency, address set-up, and the output tile leaving through an issue-bound store tail);
//   E 192x128 (80 KB of LDS, 104 registers: TWO workgroups per CU): a lone tile 1.0 us per K-tile, a pair 1.875 for both,
//   and one's prologue / epilogue under the other's K-loop: fixed ~1 us;
//   C 128x128 two-stage kernel (two per CU): 0.876 per pair and K-tile, fixed 10.8.
// 192-row tiles (round 5): 76-grid 128->256 482 tiles = 1.88 rounds instead of 1,444 128x128 tiles in 2.82 rounds of pairs
// (83 -> 70 us), 38-grid 256->512 242 tiles (95 % of the CUs) instead of 182 (70 -> 62 us), 19-grid 512->1024 248 instead of
// 184 (82 -> 72 us), 152-grid 64->128 1,926 pairs-of-E instead of 2,888 128x128 tiles (105 -> 82 us).
// (experiments build: Y3_BF16X_TILE=A|B|C|D|E forces one where it applies, tools/layer_profile.py, tools/r05_c5.sh)
int forced_tile() {
    static int v = -2;
    if (v == -2) {
        const char* e = y3_exp_env("Y3_BF16X_TILE");
        v = (e && e[0] >= 'A' && e[0] <= 'E') ? e[0] - 'A' : -1;
    }
    return v;
}

struct TileCand { int bm, bn, slots; float kt, fixed, kt_lone; };   // kt_lone: per K-tile when at most one tile per CU runs
// index = tile letter - 'A'
constexpr TileCand kTiles[5] = {
    {256, 256, 256, 1.58f, 12.6f, 1.58f},
    {256, 128, 256, 0.965f, 7.7f, 0.965f},
    {128, 128, 512, 0.876f, 10.8f, 0.95f},
    {192, 256, 256, 1.48f, 8.9f, 1.48f},
    {192, 128, 512, 1.875f, 1.0f, 1.0f},
};

int choose_tile(long long M, int cout, int S) {
    int best = 2;
    float best_t = 0.f;
    for (int t = 0; t < 5; ++t) {
        const TileCand& c = kTiles[t];
        if (c.bn > 128 && cout < 256) continue;         // a 256-column tile needs at least 256 output channels
        const long long tiles = ((M + c.bm - 1) / c.bm) * ((cout + c.bn - 1) / c.bn);
        const long long rounds = (tiles + c.slots - 1) / c.slots;
        const float est = (float)rounds * ((float)S * (tiles <= 256 ? c.kt_lone : c.kt) + c.fixed);
        if (best_t == 0.f || est < best_t) { best = t; best_t = est; }
    }
    return best;
}

template <int KS, bool UPCAT>
int dispatch_x(hipStream_t stream, const ConvArgsX& a) {
    if (a.Cin % 64 != 0)            // Cin = 32: 64-byte rows (Cout <= 32 too: the rows past Cout read as zeros)
        return launch_x<128, 64, 4, 1, 32, KS, UPCAT>(stream, a);
    if (a.Cout <= 32) return launch_x<128, 32, 4, 1, 64, KS, UPCAT>(stream, a);
    if (a.Cout <= 64) return launch_x<128, 64, 4, 1, 64, KS, UPCAT>(stream, a);
    int t = forced_tile();
    if (t < 0 || (kTiles[t].bn > 128 && a.Cout < 256)) t = choose_tile(a.M, a.Cout, KS * KS * (a.Cin / 64));
    static int pipe = -1;           // Y3_BF16X_PIPE=0: the two-stage kernel on the 256-row tiles too (A/B runs)
    if (pipe < 0) {
        const char* e = y3_exp_env("Y3_BF16X_PIPE");
        pipe = (e && e[0] == '0') ? 0 : 1;
    }
    switch (t) {
    case 0: return pipe ? launch_p<4, 2, 2, 4, KS, UPCAT>(stream, a) : launch_x<256, 256, 2, 4, 64, KS, UPCAT>(stream, a);
    case 1: return pipe ? launch_p<2, 2, 4, 2, KS, UPCAT>(stream, a) : launch_x<256, 128, 4, 2, 64, KS, UPCAT>(stream, a);
    case 3: return launch_p<3, 2, 2, 4, KS, UPCAT>(stream, a);
    case 4: return launch_p<3, 1, 2, 4, KS, UPCAT>(stream, a);
    default: return launch_x<128, 128, 2, 2, 64, KS, UPCAT>(stream, a);
    }
}

}  // namespace

// Shapes the kernels of this file take: the 3x3 convs with Cin = 32 or a multiple of 64 (64-element K-steps; 32 for the
// two Cin = 32 layers).  Decided by (k, Cin) alone, because the weight packing ([tap][Cin/BK][Cout][BK]) is chosen when
// only the kernel's shape is known.  The 1x1 convs (HBM-bound: three workgroups per CU on 32-element K-steps measured
// 10 % faster than anything here) and everything with Y3_BF16X=0 in the environment (A/B runs) stay on y3_conv_bf16.hip.
int y3_conv_bf16x_takes(int k, int cin) {
    if (k != 3 || (cin != 32 && cin % 64 != 0)) return 0;
    static int off = -1;
    if (off < 0) {
        const char* e = y3_exp_env("Y3_BF16X");
        off = (e && e[0] == '0') ? 1 : 0;
    }
    return off ? 0 : 1;
}

// Which tile the bf16 path runs this conv on (host-only; tests/test_wino44_tiling_cpu.py holds the configs[4] table against it):
// 'A'..'E' = the 3x3 tiles above, 'a'..'g' = the ring kernel's tiles, 'x' = the narrow 3x3 forms (Cin = 32 / Cout <= 64), 'o' = the
// register-staged kernel of y3_conv_bf16.hip, 's' = the Cin = 3 stem
extern "C" int y3_conv_bf16_tile(const y3_conv_desc* d) {
    if (!d) return 0;
    if (d->cin == 3) return 's';
    if (y3_conv_bf16x_takes(d->k, d->cin)) {
        if (d->cin % 64 != 0 || d->cout <= 64) return 'x';
        const long long M = (long long)d->n * (d->h / d->stride) * (d->w / d->stride);
        return 'A' + choose_tile(M, d->cout, d->k * d->k * (d->cin / 64));
    }
    if (y3_conv_bf16r_takes(d->k, d->cin)) return y3_conv_bf16r_tile(d);
    return 'o';
}

int y3_launch_pack_bf16x(hipStream_t stream, const float* w_hwio, int k, int cin, int cout, void* w_packed) {
    const int bk = cin % 64 == 0 ? 64 : 32;
    const size_t total = (size_t)k * k * cin * cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_bf16x_kernel, dim3(blocks), dim3(256), 0, stream, w_hwio,
                       static_cast<bf16_t*>(w_packed), k * k, cin, cout, bk);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

int y3_launch_conv_bf16x(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                         const float* scale, const float* shift, const void* residual, void* y, int out_f32) {
    const int Ho = d->h / d->stride, Wo = d->w / d->stride;
    const long long M = (long long)d->n * Ho * Wo;
    ConvArgsX a;
    a.x = static_cast<const bf16_t*>(x); a.xu = static_cast<const bf16_t*>(x_up);
    a.w = static_cast<const bf16_t*>(w); a.scale = scale; a.shift = shift;
    a.resid = static_cast<const bf16_t*>(residual); a.y = y;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cu = d->c_up; a.Cx = d->cin - d->c_up;
    a.Ho = Ho; a.Wo = Wo; a.Cout = d->cout; a.stride = d->stride; a.pad = d->k / 2; a.act = d->act;
    a.out_f32 = out_f32; a.M = (int)M;
    return dispatch_x<3, false>(stream, a);      // (the launcher in y3_conv_bf16.hip sends only 3x3 convs here)
}
