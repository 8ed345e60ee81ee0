// bf16-storage / fp32-accumulate variant of the fused conv (BASELINE configs[4]: 608x608 bf16 inference).
//
// Same implicit-GEMM structure as y3_conv.hip, re-dimensioned for the 16x faster matrix pipe:
//   * activations, packed weights and residuals are bf16 (2 B/element), scale/shift fp32, accumulation fp32
//     in v_mfma_f32_32x32x16_bf16 (K = 16 per instruction: lane l holds k = 8*(l>>5)..+7 of row l&31, i.e.
//     one 16-byte LDS read per fragment);
//   * BK = 32 elements = 64-byte rows (every Cin of the network is a multiple of 32); LDS rows are unpadded with the
//     16-byte slot index XOR-swizzled by (row/4)%4, which makes the ds_read_b128 fragment reads (16 rows of one
//     slot) and the 64-byte-contiguous staging writes bank-conflict free;
//   * a K-step is only 8 MFMAs x 32 cycles per wave — shorter than the memory latency — so global loads run TWO
//     K-steps ahead in two alternating register sets, and the loader is branch-free (a branch around a load makes
//     hipcc drain vmcnt(0) at its join); weights are packed [tap][Cin/32][Cout][32] so that a K-step's B tile is
//     whole 128-byte lines;
//   * epilogue in fp32 (scale/shift, LeakyReLU, residual), ONE rounding to bf16 (round-to-nearest-even) at the
//     store; the detection convs (linear, 3*(5+C) channels) write fp32 so that decode/NMS are unchanged.
// Data-parallel schedule with XCD-contiguous tile ids (no stream-K: the kernel is not matrix-pipe bound).
#include <cstdlib>
#include <type_traits>
#include "y3_internal.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct ConvArgsB {
    const bf16_t* x;     // [N,H,W,Cx]
    const bf16_t* xu;    // [N,H/2,W/2,Cu] or nullptr
    const bf16_t* w;     // packed [taps][Cout][Cin]
    const float* scale;
    const float* shift;
    const bf16_t* resid; // [M,Cout] or nullptr
    void* y;             // [M,Cout] bf16, or fp32 when out_f32
    int N, H, W, Cin, Cu, Cx;
    int Ho, Wo, Cout;
    int stride, pad, act, out_f32;
    int M;
};

constexpr int BKB = 32;            // K elements per step
constexpr int LDB = 64;            // LDS row stride in bytes (unpadded; 16-byte slots swizzled, see lds_off)
constexpr unsigned OOB = 0x80000000u;

// LDS byte offset of 16-byte slot `slot` (0..3) of row `row`: four rows share one 256-byte bank row, so the slot is
// flipped by (row/4)%4 and 16 consecutive rows of one logical slot land on 16 different bank groups.
__device__ __forceinline__ int lds_off(int row, int slot) { return row * LDB + ((slot ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round to nearest even (finite inputs)
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <int BM, int BN, int WGM, int WGN, int KS, bool UPCAT>
__global__ void __launch_bounds__(256, 3) conv_mfma_bf16_kernel(const ConvArgsB p) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    static_assert(WGM * WGN == 4 && MI >= 1 && NI >= 1, "bad wave layout");
    constexpr int ACH = BM * 4 / 256;                 // 16-byte chunks of A each thread stages (2)
    constexpr int BCH = (BN * 4 + 255) / 256;         // of B (2, 1, 1)
    constexpr int LDC = BN + 4;                       // fp32 epilogue staging stride (floats), 64 rows at a time
    // (the launcher allocates max(2 * TILE_BYTES, 64 * LDC * 4): the epilogue stages 64 fp32 rows at a time)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                        // [2][BM][80 B]
    unsigned char* Bs = smem + 2 * BM * LDB;         // [2][BN][80 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nbm = (p.M + BM - 1) / BM;
    const int kchunks = p.Cin / BKB;
    const int S = KS * KS * kchunks;

    // XCD-contiguous, column-major tile id (see y3_conv.hip)
    const int nt = gridDim.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = blockIdx.x & 7, kk8 = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk8;
    const int bn = tile / nbm, bm = tile - bn * nbm;
    const int m0 = bm * BM, n0 = bn * BN;

    const int ch = tid & 3;          // 16-byte chunk inside the 64-byte row
    const int r0 = tid >> 2;         // rows r0 + 64*j
    const int c8 = ch * 8;           // element offset of the chunk

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(UPCAT ? p.xu : p.x), 0,
        (unsigned)(UPCAT ? (size_t)p.N * (p.H >> 1) * (p.W >> 1) * p.Cu * 2 : 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.w), 0, (unsigned)((size_t)KS * KS * p.Cout * p.Cin * 2), 0x00020000);

    int a_base[ACH], a_msk[ACH], a_base_u[UPCAT ? ACH : 1];
    unsigned b_voff[BCH];
    int ld_tap = 0, ld_cc = 0;
    {
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int m = m0 + r0 + 64 * j;
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
                base = ((n * p.H + iy0) * p.W + ix0) * p.Cx;
                if (UPCAT) base_u = ((n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cu;
            }
            a_msk[j] = mk; a_base[j] = base;
            if (UPCAT) a_base_u[j] = base_u;
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int co = n0 + r0 + 64 * j;
            const bool ok = co < p.Cout && (BN >= 64 || r0 < BN) && (r0 + 64 * j) < BN;
            b_voff[j] = ok ? (unsigned)(co * BKB + c8) * 2u : OOB;
        }
    }
    u32x4 ra[2][ACH], rb[2][BCH];
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // fetch the prepared K-step into register set s and step the loader; branch-free (past the last K-step the
    // loads read in-bounds-or-zero addresses and are never consumed)
    auto issue = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int ky = (KS == 1) ? 0 : ld_tap / KS;
        const int kx = (KS == 1) ? 0 : ld_tap - ky * KS;
        const int tap_off = (ky * p.W + kx) * p.Cx + c8;
        const int c0 = ld_cc * BKB;
        const bool from_up = UPCAT && c0 < p.Cu;
        const unsigned soff = (unsigned)(from_up ? c0 : c0 - (UPCAT ? p.Cu : 0)) * 2u;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const bool ok = ((a_msk[j] >> ky) & (a_msk[j] >> (4 + kx)) & 1) != 0;
            const unsigned voff = ok ? (unsigned)(a_base[j] + tap_off) * 2u : OOB;
            if (UPCAT) {
                const unsigned voff_u = ok ? (unsigned)(a_base_u[j] + c8) * 2u : OOB;
                ra[s][j] = from_up ? __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u, soff, 0)
                                   : __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff, soff, 0);
            } else {
                ra[s][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff, soff, 0);
            }
        }
        // weights: [tap][Cin/32][Cout][32] — the B tile of a K-step is contiguous
        const unsigned wsoff = (unsigned)((ld_tap * kchunks + ld_cc) * p.Cout) * (BKB * 2u);
#pragma unroll
        for (int j = 0; j < BCH; ++j) rb[s][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff[j], wsoff, 0);
        const bool wrap = ++ld_cc == kchunks;
        ld_cc = wrap ? 0 : ld_cc;
        ld_tap += wrap ? 1 : 0;
    };
    auto store = [&](auto sc) {
        constexpr int s = decltype(sc)::value;      // register set s -> LDS buffer s
        unsigned char* as = As + s * BM * LDB;
        unsigned char* bs = Bs + s * BN * LDB;
#pragma unroll
        for (int j = 0; j < ACH; ++j) *reinterpret_cast<u32x4*>(as + lds_off(r0 + 64 * j, ch)) = ra[s][j];
#pragma unroll
        for (int j = 0; j < BCH; ++j)
            if ((r0 + 64 * j) < BN) *reinterpret_cast<u32x4*>(bs + lds_off(r0 + 64 * j, ch)) = rb[s][j];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // lane l feeds row l&31 of each 32x32 MFMA tile with the 8 k-values of slot 2*kk + (l>>5); tile row bases are
    // multiples of 32, so the swizzle term depends on the lane only
    const int frag_row = lane & 31, frag_half = lane >> 5;
    auto compute = [&](int buf) {
        const unsigned char* as = As + buf * BM * LDB + wm * WTM * LDB;
        const unsigned char* bs = Bs + buf * BN * LDB + wn * WTN * LDB;
#pragma unroll
        for (int kk = 0; kk < BKB / 16; ++kk) {
            const int off = lds_off(frag_row, 2 * kk + frag_half);
            bf16x8 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(as + mi * 32 * LDB + off));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bs + ni * 32 * LDB + off));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // K-step g uses LDS buffer / register set g&1; its loads were issued two K-steps earlier
    I0 i0;
    I1 i1;
    issue(i0);
    issue(i1);
    store(i0);
    __syncthreads();
    for (int g = 0; g < S; g += 2) {
        issue(i0);                     // K-step g+2
        compute(0);
        store(i1);                     // K-step g+1
        __syncthreads();
        issue(i1);                     // K-step g+3
        if (g + 1 < S) compute(1);
        store(i0);                     // K-step g+2
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------------
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
    constexpr int C4 = BN / 4, RPP = 256 / C4, PASSES = 64 / RPP;
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
                        const u32x2 rv = *reinterpret_cast<const u32x2*>(p.resid + o);
                        v[0] += __uint_as_float(rv[0] << 16);
                        v[1] += __uint_as_float(rv[0] & 0xFFFF0000u);
                        v[2] += __uint_as_float(rv[1] << 16);
                        v[3] += __uint_as_float(rv[1] & 0xFFFF0000u);
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

// stem: fp32 image [N,H,W,3] -> bf16 [N,H,W,32]; fp32 arithmetic, HWIO fp32 weights (same as the fp32 stem)
__global__ void __launch_bounds__(256) conv_stem_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                             int N, int H, int W, int M, int act) {
    constexpr int COUT = 32;
    __shared__ float ssc[COUT], ssh[COUT];
    __shared__ __attribute__((aligned(16))) float xs[27 * 256];      // [tap*3+ci][thread]; later the output staging
    for (int i = threadIdx.x; i < COUT; i += 256) { ssc[i] = scale[i]; ssh[i] = shift[i]; }
    __syncthreads();
    const int m_raw = blockIdx.x * 256 + threadIdx.x;
    const int m = m_raw < M ? m_raw : M - 1;          // threads past the end recompute the last pixel, store nothing
    const int n = m / (H * W);
    const int rem = m - n * H * W;
    const int oy = rem / W, ox = rem - oy * W;
    // inputs first (one round of load latency), parked in a thread-private LDS column; see conv_stem_kernel
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox - 1 + kx;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float* src = x + ((size_t)(n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = src[ci];
                xs[((ky * 3 + kx) * 3 + ci) * 256 + threadIdx.x] = ok ? v : 0.f;
            }
        }
    }
    // the layer is bound by these 27 x COUT multiply-adds per pixel: packed (two channels per v_pk_fma_f32)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t acc2[COUT / 2];
#pragma unroll
    for (int c = 0; c < COUT / 2; ++c) acc2[c] = f32x2_t{0.f, 0.f};
#pragma unroll 3
    for (int t = 0; t < 27; ++t) {
        const float xv = xs[t * 256 + threadIdx.x];
        const f32x2_t x2 = {xv, xv};
        // the tap's 32 weights: wave-uniform addresses -> scalar loads, the FMAs take them as SGPR operands (read as
        // broadcasts from the LDS they cost one 16-byte LDS read per two packed FMAs and bound the kernel)
        const float* wr = w + t * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 2)
            acc2[c / 2] = __builtin_elementwise_fma(x2, f32x2_t{wr[c], wr[c + 1]}, acc2[c / 2]);
    }
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = acc2[c / 2][c & 1];
    unsigned pk[COUT / 2];
#pragma unroll
    for (int c = 0; c < COUT; c += 2) {
        float t0 = acc[c] * ssc[c] + ssh[c], t1 = acc[c + 1] * ssc[c + 1] + ssh[c + 1];
        if (act) { t0 = t0 > 0.f ? t0 : 0.1f * t0; t1 = t1 > 0.f ? t1 : 0.1f * t1; }
        pk[c / 2] = (unsigned)f32_to_bf16(t0) | ((unsigned)f32_to_bf16(t1) << 16);
    }
    // store through the LDS (see conv_stem_kernel): 64 bytes per pixel, staged rows of 80 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                   // every thread is done with its xs column
    unsigned char* stage = reinterpret_cast<unsigned char*>(xs) + wave * 64 * 80;
#pragma unroll
    for (int c = 0; c < COUT / 8; ++c)
        *reinterpret_cast<u32x4*>(stage + lane * 80 + c * 16) = u32x4{pk[4 * c], pk[4 * c + 1], pk[4 * c + 2], pk[4 * c + 3]};
    __syncthreads();
    const int m_wave = blockIdx.x * 256 + wave * 64;
#pragma unroll
    for (int i = 0; i < COUT / 8; ++i) {
        const int f = i * 64 + lane;                   // 16-byte piece inside the wave's 64 x 64-byte block
        const int pix = f >> 2, piece = f & 3;
        if (m_wave + pix < M)
            *reinterpret_cast<u32x4*>(y + (size_t)(m_wave + pix) * COUT + piece * 8) =
                *reinterpret_cast<const u32x4*>(stage + pix * 80 + piece * 16);
    }
}

__global__ void pack_weights_bf16_kernel(const float* __restrict__ w_hwio, bf16_t* __restrict__ w_packed, int taps,
                                         int cin, int cout) {
    const size_t total = (size_t)taps * cin * cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const size_t r = i / cin;
        const int co = (int)(r % cout);
        const int t = (int)(r / cout);
        // [tap][Cin/32][Cout][32]: the B tile of one K-step (32 input channels) is contiguous
        w_packed[((((size_t)t * (cin / BKB) + ci / BKB) * cout + co) * BKB) + (ci % BKB)] =
            f32_to_bf16(w_hwio[((size_t)t * cin + ci) * cout + co]);
    }
}

template <int BM, int BN, int WGM, int WGN, int KS, bool UPCAT>
int launch_b(hipStream_t stream, const ConvArgsB& a) {
    auto kern = conv_mfma_bf16_kernel<BM, BN, WGM, WGN, KS, UPCAT>;
    constexpr size_t tiles = (size_t)2 * (BM + BN) * LDB, stage = (size_t)64 * (BN + 4) * 4;
    constexpr size_t lds = tiles > stage ? tiles : stage;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(nbm * nbn), dim3(256), lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

template <int KS, bool UPCAT>
int dispatch_b(hipStream_t stream, const ConvArgsB& a) {
    if (a.Cout <= 32) return launch_b<128, 32, 4, 1, KS, UPCAT>(stream, a);
    if (a.Cout <= 64) return launch_b<128, 64, 4, 1, KS, UPCAT>(stream, a);
    if (KS == 1) {
        // 1x1 convs whose 128x128 tiles would not fill two rounds of the 256 CUs (the 19-grid layers and the two route
        // convs at bs=16, 608x608: 184 / 46 tiles): 64x64 tiles, three workgroups per CU - 26.5 -> 21.9 us and 18.9 ->
        // 12.0 us per launch; from 512 tiles up the wide tile wins again (measured, tools/layer_profile.py)
        const long long tiles = (long long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
        if (tiles < 512) return launch_b<64, 64, 2, 2, KS, UPCAT>(stream, a);
    }
    return launch_b<128, 128, 2, 2, KS, UPCAT>(stream, a);
}

}  // namespace

int y3_launch_conv_bf16(hipStream_t stream, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                        const float* scale, const float* shift, const void* residual, void* y, int out_f32) {
    Y3_CHECK_ARG(d && x && w && scale && shift && y, "y3_conv2d_fwd_bf16: null pointer argument");
    Y3_CHECK_ARG(d->k == 1 || d->k == 3, "y3_conv2d_fwd_bf16: kernel_size must be 1 or 3");
    Y3_CHECK_ARG(d->stride == 1 || d->stride == 2, "y3_conv2d_fwd_bf16: stride must be 1 or 2");
    Y3_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "y3_conv2d_fwd_bf16: bad dimension");
    Y3_CHECK_ARG(!(d->stride == 2 && (d->h % 2 || d->w % 2)), "y3_conv2d_fwd_bf16: stride-2 conv needs even H,W");
    Y3_CHECK_ARG((x_up != nullptr) == (d->c_up > 0), "y3_conv2d_fwd_bf16: x_up and c_up must agree");
    const int Ho = d->h / d->stride, Wo = d->w / d->stride;
    const long long M = (long long)d->n * Ho * Wo;
    Y3_CHECK_ARG((long long)d->n * d->h * d->w * d->cin < (1LL << 30) && M * d->cout < (1LL << 30),
                 "y3_conv2d_fwd_bf16: tensor exceeds 2^30 elements (32-bit byte offsets)");
    if (d->cin == 3) {
        Y3_CHECK_ARG(d->k == 3 && d->cout == 32 && d->stride == 1 && !x_up && !residual && !out_f32,
                     "y3_conv2d_fwd_bf16: Cin=3 is supported only as the 3x3 3->32 stem conv (fp32 image in)");
        hipLaunchKernelGGL(conv_stem_bf16_kernel, dim3((int)((M + 255) / 256)), dim3(256), 0, stream,
                           static_cast<const float*>(x), static_cast<const float*>(w), scale, shift,
                           static_cast<bf16_t*>(y), d->n, d->h, d->w, (int)M, d->act);
        Y3_CHECK_HIP(hipGetLastError());
        return Y3_OK;
    }
    Y3_CHECK_ARG(d->cin % BKB == 0, "y3_conv2d_fwd_bf16: Cin must be 3 or a multiple of %d", BKB);
    Y3_CHECK_ARG(out_f32 || d->cout % 4 == 0, "y3_conv2d_fwd_bf16: bf16 output needs Cout %% 4 == 0");
    if (y3_conv_bf16x_takes(d->k, d->cin))     // the LDS-DMA kernels (y3_conv_bf16x.hip) and their weight packing
        return y3_launch_conv_bf16x(stream, d, x, x_up, w, scale, shift, residual, y, out_f32);
    if (y3_conv_bf16r_takes(d->k, d->cin)) {   // the 1x1 convs: persistent LDS-DMA ring kernel (y3_conv_bf16r.hip)
        Y3_CHECK_ARG(d->stride == 1, "y3_conv2d_fwd_bf16: 1x1 conv must have stride 1");
        return y3_launch_conv_bf16r(stream, d, x, x_up, w, scale, shift, residual, y, out_f32);
    }
    ConvArgsB a;
    a.x = static_cast<const bf16_t*>(x); a.xu = static_cast<const bf16_t*>(x_up);
    a.w = static_cast<const bf16_t*>(w); a.scale = scale; a.shift = shift;
    a.resid = static_cast<const bf16_t*>(residual); a.y = y;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cu = d->c_up; a.Cx = d->cin - d->c_up;
    a.Ho = Ho; a.Wo = Wo; a.Cout = d->cout; a.stride = d->stride; a.pad = d->k / 2; a.act = d->act;
    a.out_f32 = out_f32; a.M = (int)M;
    if (x_up) {
        Y3_CHECK_ARG(d->k == 1 && d->stride == 1 && d->c_up % BKB == 0 && d->c_up < d->cin && d->h % 2 == 0 &&
                         d->w % 2 == 0, "y3_conv2d_fwd_bf16: bad fused upsample+concat configuration");
        return dispatch_b<1, true>(stream, a);
    }
    if (d->k == 1) {
        Y3_CHECK_ARG(d->stride == 1, "y3_conv2d_fwd_bf16: 1x1 conv must have stride 1");
        return dispatch_b<1, false>(stream, a);
    }
    return dispatch_b<3, false>(stream, a);
}

extern "C" int y3_pack_conv_weights_bf16(y3_ctx* ctx, const float* w_hwio, int k, int cin, int cout, void* w_packed) {
    Y3_CHECK_ARG(ctx && w_hwio && w_packed, "y3_pack_conv_weights_bf16: null argument");
    Y3_CHECK_ARG(k > 0 && cin > 0 && cout > 0 && cin % BKB == 0,
                 "y3_pack_conv_weights_bf16: cin must be a positive multiple of %d", BKB);
    if (y3_conv_bf16x_takes(k, cin) || y3_conv_bf16r_takes(k, cin))      // [tap][Cin/64][Cout][64]
        return y3_launch_pack_bf16x(ctx->stream, w_hwio, k, cin, cout, w_packed);
    const size_t total = (size_t)k * k * cin * cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, ctx->stream, w_hwio,
                       static_cast<bf16_t*>(w_packed), k * k, cin, cout);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

extern "C" int y3_conv2d_fwd_bf16(y3_ctx* ctx, const y3_conv_desc* d, const void* x, const void* x_up, const void* w,
                                  const float* scale, const float* shift, const void* residual, void* y, int out_f32) {
    Y3_CHECK_ARG(ctx, "y3_conv2d_fwd_bf16: null context");
    return y3_launch_conv_bf16(ctx->stream, d, x, x_up, w, scale, shift, residual, y, out_f32);
}
