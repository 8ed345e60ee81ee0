// Shared pieces of the implicit-GEMM conv kernels (exact-fp32 MFMA kernel in y3_conv.hip, split-bf16 kernel in
// y3_conv_split.hip): argument block, tile geometry, the LDS-staged epilogue, the stream-K partition and its
// fix-up kernel.
#pragma once
#include <cstdlib>
#include "y3_internal.h"

namespace y3conv {

struct ConvArgs {
    const float* x;      // [N,H,W,Cx]  (Cx = Cin - Cu)
    const float* xu;     // [N,H/2,W/2,Cu] or nullptr
    const float* w;      // packed [taps][Cout][Cin]   (stem: HWIO [27][32])
    const float* scale;  // [Cout]
    const float* shift;  // [Cout]
    const float* resid;  // [M,Cout] or nullptr
    float* y;            // [M,Cout]
    float* partial;      // stream-K scratch: [workers][2][BM*BN] raw accumulators
    int N, H, W, Cin, Cu, Cx;
    int Ho, Wo, Cout;
    int stride, pad, act;
    int M;
    int workers;         // stream-K grid size (0 = data-parallel)
    int wrev;            // 1: walk the weight taps in reverse order (data-gradient = conv with the flipped kernel)
    int tmode;           // 1: data gradient of a stride-2 3x3 conv, ONE output parity class (cy,cx) per launch:
                         //    x is the coarse gradient [N,H,W,Cx], the output grid is [N,2H,2W]; rows enumerate
                         //    the class pixels (2y'+cy, 2x'+cx); only the taps whose parity matches contribute
                         //    (1, 2, 2 or 4 of the 9), reading x[y'+dy, x'+dx] with dy,dx in {0,1}
    int cy, cx, ntaps;   // tmode only
    int bkk;             // K elements per K-step of the kernel that produced the stream-K partials (fix-up)
};

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr unsigned OOB = 0x80000000u;  // any offset >= num_records reads as 0 through a buffer load

template <int BM, int BN, int WGM, int WGN>
struct Geo {
    static constexpr int WTM = BM / WGM, WTN = BN / WGN;  // per-wave output tile
    static constexpr int MI = WTM / 32, NI = WTN / 32;    // 32x32 MFMA tiles per wave
    static constexpr int LDC = BN + 4;                    // epilogue staging row stride
    static constexpr size_t LDS_BYTES = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    static_assert(MI >= 1 && NI >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    static_assert((size_t)BM * LDC * sizeof(float) <= LDS_BYTES, "accumulator tile must fit in the LDS");
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// four fp32 values -> NP planes of four bf16 values (8 bytes per plane).  Truncating split: every remainder is
// exactly representable, so x1 + x2 + x3 == x for NP = 3; the last plane of NP = 2 rounds to nearest even.
template <int NP>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&o)[NP]) {
    unsigned h[NP][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float r = v[q];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            unsigned u = __float_as_uint(r);
            if (NP == 2 && pl == 1) u += 0x7fffu + ((u >> 16) & 1u);
            u &= 0xffff0000u;
            h[pl][q] = u;
            if (pl + 1 < NP) r -= __uint_as_float(u);
        }
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        o[pl][0] = (h[pl][0] >> 16) | h[pl][1];
        o[pl][1] = (h[pl][2] >> 16) | h[pl][3];
    }
}
// ---- epilogue shared by the conv kernel and the stream-K fix-up kernel ------------------------------
// D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int BM, int BN, int WGM, int WGN, bool TMODE>
__device__ __forceinline__ void epilogue(const ConvArgs& p, float* smem,
                                         f32x16 (&acc)[Geo<BM, BN, WGM, WGN>::MI][Geo<BM, BN, WGM, WGN>::NI],
                                         int m0, int n0) {
    using G = Geo<BM, BN, WGM, WGN>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    // output pixel of GEMM row `row`: the row itself, or (tmode) pixel (2y'+cy, 2x'+cx) of the 2x finer grid
    auto out_pixel = [&](int row) -> size_t {
        if (!TMODE) return (size_t)row;
        const int hw = p.H * p.W;
        const int n = row / hw;
        const int rem = row - n * hw;
        const int y = rem / p.W, x = rem - y * p.W;
        return ((size_t)(n * 2 * p.H + 2 * y + p.cy)) * (2 * p.W) + 2 * x + p.cx;
    };
    if ((p.Cout & 3) == 0) {
        constexpr int C4 = BN / 4;     // float4 columns per tile row
        constexpr int RPP = 256 / C4;  // rows covered per pass
        constexpr int PASSES = BM / RPP;
        const int tc = (tid % C4) * 4, tr = tid / C4;
        const int col = n0 + tc;
        const bool cok = col < p.Cout;
        // residual tile first: its HBM/L2 latency overlaps the LDS staging below
        f32x4 res[PASSES];
        if (p.resid) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int row = m0 + tr + i * RPP;
                res[i] = (cok && row < p.M)
                             ? *reinterpret_cast<const f32x4*>(p.resid + out_pixel(row) * p.Cout + col)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        float* cs = smem;
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    cs[(wm * G::WTM + mi * 32 + row_l + (r & 3) + 8 * (r >> 2)) * G::LDC + wn * G::WTN +
                       ni * 32 + col_l] = acc[mi][ni][r];
        __syncthreads();
        if (cok) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + col);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + col);
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int rr = tr + i * RPP;
                const int row = m0 + rr;
                if (row < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(cs + rr * G::LDC + tc);
                    v = v * sc + sh;
                    if (p.act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
                    }
                    if (p.resid) v += res[i];
                    *reinterpret_cast<f32x4*>(p.y + out_pixel(row) * p.Cout + col) = v;
                }
            }
        }
        return;
    }
    // Cout not a multiple of 4 (detection heads, 3*(5+C)): rows are not 16-byte aligned -> scalar stores
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni) {
        const int col = n0 + wn * G::WTN + ni * 32 + col_l;
        const bool cok = col < p.Cout;
        const float sc = cok ? p.scale[col] : 0.f;
        const float sh = cok ? p.shift[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
            const int rbase = m0 + wm * G::WTM + mi * 32 + row_l;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (cok && row < p.M) {
                    float v = acc[mi][ni][r] * sc + sh;
                    if (p.act) v = v > 0.f ? v : 0.1f * v;
                    const size_t o = out_pixel(row) * p.Cout + col;
                    if (p.resid) v += p.resid[o];
                    p.y[o] = v;
                }
            }
        }
    }
}

// Balanced contiguous partition of `items` over `workers`: worker w owns [begin(w), begin(w+1)).
__device__ __host__ __forceinline__ long long sk_begin(long long items, int workers, int w) {
    const long long q = items / workers, r = items % workers;
    return (long long)w * q + (w < r ? w : r);
}
__device__ __host__ __forceinline__ int sk_owner(long long items, int workers, long long item) {
    const long long q = items / workers, r = items % workers;
    if (item < r * (q + 1)) return (int)(item / (q + 1));
    return (int)(r + (item - r * (q + 1)) / q);
}
// XCD-aware worker id: workgroup b runs on XCD b%8 (observed, speed only); give each XCD a contiguous
// eighth of the item space so neighbouring tiles share their A halo / B panel in one L2.
__device__ __forceinline__ int sk_worker_id(int b, int workers) {
    return (workers & 7) == 0 ? (b & 7) * (workers >> 3) + (b >> 3) : b;
}

// Stream-K fix-up: one workgroup per output tile; tiles computed whole by one worker exit at once, split
// tiles sum their partials in worker (= K) order and run the common epilogue.
template <int BM, int BN, int WGM, int WGN, int KS, bool TMODE>
__global__ void __launch_bounds__(256, 2) conv_streamk_fixup_kernel(const ConvArgs p) {
    using G = Geo<BM, BN, WGM, WGN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nbn = (p.Cout + BN - 1) / BN;
    const int S = (TMODE ? p.ntaps : KS * KS) * (p.Cin / p.bkk);
    const long long items = (long long)((p.M + BM - 1) / BM) * nbn * S;
    const int tile = blockIdx.x;
    const long long t0 = (long long)tile * S, t1 = t0 + S;
    const int w_lo = sk_owner(items, p.workers, t0), w_hi = sk_owner(items, p.workers, t1 - 1);
    if (w_lo == w_hi) return;
    const int tid = threadIdx.x;
    f32x16 acc[G::MI][G::NI];
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int w = w_lo; w <= w_hi; ++w) {
        const long long wb = sk_begin(items, p.workers, w);
        const int first = (int)(wb / S);
        const float* slot = p.partial + ((size_t)w * 2 + (tile == first ? 0 : 1)) * (BM * BN);
        const f32x4* slot4 = reinterpret_cast<const f32x4*>(slot);
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = slot4[((mi * G::NI + ni) * 4 + g) * 256 + tid];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[mi][ni][4 * g + q] += v[q];
                }
    }
    const int nbm = (p.M + BM - 1) / BM;
    const int bn = tile / nbm, bm = tile - bn * nbm;
    epilogue<BM, BN, WGM, WGN, TMODE>(p, smem, acc, bm * BM, bn * BN);
}

// ---- stem conv: 3x3, Cin = 3 -> COUT (=32), stride 1 ------------------------------------------------
// K = 27 is too short for the implicit-GEMM tile and the layer is HBM-write bound (709 MB out per
// bs=32 batch vs 9.6 GFLOP): one thread per output pixel, weights HWIO [27][COUT] broadcast from LDS.
template <int COUT>
__global__ void __launch_bounds__(256) conv_stem_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float ws[27 * COUT];
    __shared__ float ssc[COUT], ssh[COUT];
    __shared__ __attribute__((aligned(16))) float xs[4 * 64 * (COUT + 4)];   // inputs [tap*3+ci][thread] (27*256
                                                                              // floats), later the output staging
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) ws[i] = p.w[i];
    for (int i = threadIdx.x; i < COUT; i += 256) {
        ssc[i] = p.scale[i];
        ssh[i] = p.shift[i];
    }
    __syncthreads();
    const int m_raw = blockIdx.x * 256 + threadIdx.x;
    const int m = m_raw < p.M ? m_raw : p.M - 1;     // threads past the end recompute the last pixel, store nothing
    const int HoWo = p.Ho * p.Wo;
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    // all 27 input values first (one round of global-load latency instead of nine dependent ones), parked in a
    // thread-private LDS column so that the FMA loop below can stay a real loop (fully unrolled, hipcc hoists all
    // 27 x COUT/4 weight reads ahead of the FMAs and needs ~390 VGPRs)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * p.stride - p.pad + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * p.stride - p.pad + kx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = p.x + ((size_t)(n * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = src[ci];
                xs[((ky * 3 + kx) * 3 + ci) * 256 + threadIdx.x] = ok ? v : 0.f;
            }
        }
    }
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll 3
    for (int t = 0; t < 27; ++t) {
        const float x = xs[t * 256 + threadIdx.x];
        const float* wr = ws + t * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
            acc[c + 0] = fmaf(x, wv[0], acc[c + 0]);
            acc[c + 1] = fmaf(x, wv[1], acc[c + 1]);
            acc[c + 2] = fmaf(x, wv[2], acc[c + 2]);
            acc[c + 3] = fmaf(x, wv[3], acc[c + 3]);
        }
    }
    // Store through the LDS: a thread owns one pixel = COUT contiguous floats, so direct stores would touch 64 cache
    // lines per wave instruction; staged, each store instruction writes 1 KB contiguous.
    static_assert(COUT == 32, "staging layout below assumes 32 output channels");
    constexpr int SROWF = COUT + 4;                      // staged row stride in floats (pad: conflict-free writes)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                     // every thread is done with its xs column
    float* stage = xs + wave * 64 * SROWF;
#pragma unroll
    for (int c = 0; c < COUT; c += 4) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t = acc[c + q] * ssc[c + q] + ssh[c + q];
            if (p.act) t = t > 0.f ? t : 0.1f * t;
            v[q] = t;
        }
        *reinterpret_cast<f32x4*>(stage + lane * SROWF + c) = v;
    }
    __syncthreads();
    const int m_wave = blockIdx.x * 256 + wave * 64;
#pragma unroll
    for (int i = 0; i < COUT / 4; ++i) {
        const int f = i * 64 + lane;                     // float4 index inside the wave's 64 x COUT block
        const int pix = f / (COUT / 4), c4 = f % (COUT / 4);
        if (m_wave + pix < p.M)
            *reinterpret_cast<f32x4*>(p.y + (size_t)(m_wave + pix) * COUT + c4 * 4) =
                *reinterpret_cast<const f32x4*>(stage + pix * SROWF + c4 * 4);
    }
}

template <typename K>
inline int set_lds_attr(K kern, size_t lds) {
    Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return Y3_OK;
}

constexpr int SK_WORKERS = 512;  // 256 CUs x 2 co-resident 128x128 workgroups (73.7 KB LDS, 176 VGPRs)

// Schedule choice: stream-K for 3x3 convs on 128x128 tiles whose tile count is within a few multiples
// of the co-resident workgroup count (tail quantisation dominates there); Y3_CONV_STREAMK=0/1 overrides
// (experiment hook for tools/conv_bench.py).
inline bool use_streamk(const ConvArgs& a, int k, bool has_ws) {
    if (!has_ws || k != 3 || a.Cout < 128 || a.xu) return false;
    static int force = -2;
    if (force == -2) {
        const char* e = getenv("Y3_CONV_STREAMK");
        force = e ? atoi(e) : -1;
    }
    if (force == 0) return false;
    const int tiles = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    if (force == 1) return true;
    return tiles < 4 * SK_WORKERS;
}

}  // namespace y3conv
