// Shared pieces of the implicit-GEMM conv kernels (exact-fp32 MFMA kernel in y3_conv.hip, split-bf16 kernel in
// y3_conv_split.hip): argument block, tile geometry, the LDS-staged epilogue, the stream-K partition and its
// in-kernel hand-off of cut tiles.
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
    float* partial;      // stream-K scratch: [workers][BM*BN] raw accumulators of the cut tiles' later K-ranges
    unsigned* flags;     // stream-K scratch: [workers] "partial published" words, zeroed ahead of every launch
    unsigned* err;       // stream-K: device-visible error word (a consumer whose poll expires ORs a code into it) or null
    unsigned spin_limit; // stream-K: polls per awaited flag before giving up
    int fault;           // stream-K test hook: producers skip raising their flag
    float* stats;        // nullptr, or [ceil(M/BM)][2][Cout]: per row block of the output, the column sums of y and y^2
                         // (batch-norm statistics of the training forward, taken where the tile is already in registers)
    const float* bz;     // BSTATS instantiations (a data gradient whose output IS the dy of a BN layer): that layer's raw conv
    const float* bvec;   // output z [M,Cout] and its [4][Cout] mean / inv_std / folded scale / folded shift; `stats` then gets
                         // the column sums of g' = y * leaky'(z*scale+shift) and g' * (z-mean)*inv_std (the BN backward reduction)
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
// ---- epilogue shared by the conv kernels ------------------------------------------------------------
// D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// STATS (training forward only; a template parameter so that the inference instantiations carry none of it): the column
// sums of y and y^2 over the tile's rows go to p.stats (ConvArgs).
template <int BM, int BN, int WGM, int WGN, bool TMODE, bool STATS = false, bool BSTATS = false>
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
    {
        // Rows of the output are Cout floats: 16-byte aligned only when Cout % 4 == 0.  The detection heads (Cout =
        // 3*(5+C) = 255) take the same staged path with 4-byte-aligned dwordx4 accesses (f32x4_u; gfx950 global memory
        // serves them) and a per-element tail for the one column quad that crosses Cout.
        constexpr int C4 = BN / 4;     // float4 columns per tile row
        constexpr int RPP = 256 / C4;  // rows covered per pass
        constexpr int PASSES = BM / RPP;
        const int tc = (tid % C4) * 4, tr = tid / C4;
        const int col = n0 + tc;
        const bool cok = col < p.Cout;
        if (PASSES <= 8 && (p.Cout & 3) == 0) {
            // Every layer but the detection heads, on the 64x64 / 128x32 / 128x64 tiles (measured on the 128x128 tiles, whose
            // kernels sit at the register limit: 3 % slower, tools/r06_epi_ab.sh - they keep the pass-by-pass path below).  Loads and stores share vmcnt on gfx950 and may complete out of order with
            // each other, so hipcc waits vmcnt(0) for a load whenever a store is pending: with the residual / scale / shift
            // loads consumed pass by pass (the odd-Cout path below) every pass waited for the previous pass's STORE to
            // complete - four store round trips in a row per 64x64 tile, sixteen per 128x128 tile - and so did the first LDS
            // write of the next tile (its operands, prefetched under this epilogue, are loads too).  Here every load goes out
            // ahead of the staging, ONE compiler-visible wait sits behind the staging barrier, and the stores leave back to
            // back; invalid rows / columns are out-of-range buffer offsets (no branches).  Same pattern as w44_tail.  A tile
            // of eight passes WITH a residual takes its residual four passes at a time: one round trip more.
            constexpr int CH = PASSES < 4 ? PASSES : 4;
            const size_t out_rows = TMODE ? (size_t)p.N * 4 * p.H * p.W : (size_t)p.M;
            const unsigned ybytes = (unsigned)(out_rows * p.Cout * 4);
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.resid ? p.resid : p.y), 0, p.resid ? ybytes : 0u, 0x00020000);    // (no residual: every offset is out of range = zeros)
            auto off_of = [&](int i) -> unsigned {
                const int row = m0 + tr + i * RPP;
                return (cok && row < p.M) ? (unsigned)((out_pixel(row) * p.Cout + col) * 4) : OOB;
            };
            f32x4 res[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i)
                res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, off_of(i), 0, 0));
            // BSTATS: the BN layer's z at the same places, and its per-channel vectors
            const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(BSTATS ? p.bz : p.y), 0, BSTATS ? ybytes : 0u, 0x00020000);
            f32x4 zr[BSTATS ? CH : 1];
            f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bsc = bmu, bsh = bmu;
            if (BSTATS) {
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    zr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, off_of(i), 0, 0));
                const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bvec), 0, (unsigned)p.Cout * 16u, 0x00020000);
                const unsigned vo = cok ? (unsigned)col * 4u : OOB;
                bmu = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, 0, 0));
                bis = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, (unsigned)p.Cout * 4u, 0));
                bsc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, (unsigned)p.Cout * 8u, 0));
                bsh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, (unsigned)p.Cout * 12u, 0));
            }
            const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale), 0, (unsigned)p.Cout * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift), 0, (unsigned)p.Cout * 4u, 0x00020000);
            const f32x4 sc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sc, cok ? (unsigned)col * 4u : OOB, 0, 0));
            const f32x4 sh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sh, cok ? (unsigned)col * 4u : OOB, 0, 0));
            float* cs = smem;
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        cs[(wm * G::WTM + mi * 32 + row_l + (r & 3) + 8 * (r >> 2)) * G::LDC + wn * G::WTN +
                           ni * 32 + col_l] = acc[mi][ni][r];
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), visible to hipcc's wait-count pass: every load above has landed
            __syncthreads();
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < PASSES / CH; ++c) {
                if (c > 0 && (p.resid || BSTATS)) {
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, off_of(c * CH + i), 0, 0));
                    if (BSTATS) {
#pragma unroll
                        for (int i = 0; i < CH; ++i)
                            zr[BSTATS ? i : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, off_of(c * CH + i), 0, 0));
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(cs + (tr + (c * CH + i) * RPP) * G::LDC + tc);
                    v = v * sc + sh;
                    if (p.act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
                    }
                    v += res[i];
                    const unsigned off = off_of(c * CH + i);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, off, 0, 0);
                    if (STATS) {
                        if (off != OOB) { s1 += v; s2 += v * v; }
                    }
                    if (BSTATS) {
                        if (off != OOB) {
                            const f32x4 z = zr[BSTATS ? i : 0];
                            f32x4 g = v;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                // the LeakyReLU branch exactly as bn_apply_fwd / bn_apply_bwd take it (y3_train.hip is built
                                // without FMA contraction): a fused multiply-add here flips the elements that sit on the edge
                                float u;
                                {
#pragma clang fp contract(off)
                                    u = z[q] * bsc[q] + bsh[q];
                                }
                                g[q] = u > 0.f ? g[q] : 0.1f * g[q];
                            }
                            s1 += g;
                            s2 += g * ((z - bmu) * bis);
                        }
                    }
                }
            }
            if (STATS || BSTATS) {
                __syncthreads();                           // every thread is done reading the staged tile
                float* red = smem;                         // [RPP][2][BN]
                *reinterpret_cast<f32x4*>(red + (tr * 2 + 0) * BN + tc) = s1;
                *reinterpret_cast<f32x4*>(red + (tr * 2 + 1) * BN + tc) = s2;
                __syncthreads();
                if (tid < C4 && cok) {                     // (tid < C4  <=>  tr == 0: tc = 4 * tid)
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < RPP; ++k) {
                        a += *reinterpret_cast<const f32x4*>(red + (k * 2 + 0) * BN + tc);
                        b += *reinterpret_cast<const f32x4*>(red + (k * 2 + 1) * BN + tc);
                    }
                    float* st = p.stats + (size_t)(m0 / BM) * 2 * p.Cout;
                    *reinterpret_cast<f32x4*>(st + col) = a;
                    *reinterpret_cast<f32x4*>(st + p.Cout + col) = b;
                }
                __syncthreads();                           // (the LDS goes back to the caller)
            }
            return;
        }
        // ---- the 128x128 tiles, and Cout % 4 != 0 (the detection heads, Cout = 3 * (5 + C): 4-byte-aligned accesses, a
        // per-element tail for the quad that crosses Cout) ----
        const bool full = col + 3 < p.Cout;      // (cok && !full: the quad that crosses Cout, odd Cout only)
        // residual tile first: its HBM/L2 latency overlaps the LDS staging below
        f32x4 res[PASSES];
        if (p.resid) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int row = m0 + tr + i * RPP;
                res[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cok && row < p.M) {
                    const float* rp = p.resid + out_pixel(row) * p.Cout + col;
                    if (full) res[i] = *reinterpret_cast<const f32x4_u*>(rp);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col + q < p.Cout) res[i][q] = rp[q];
                    }
                }
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
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};      // column sums of y, y^2 over this thread's rows
        if (cok) {
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (full) {                               // (one dwordx4 each; no alignment assumed)
                sc = *reinterpret_cast<const f32x4_u*>(p.scale + col);
                sh = *reinterpret_cast<const f32x4_u*>(p.shift + col);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col + q < p.Cout) {
                        sc[q] = p.scale[col + q];
                        sh[q] = p.shift[col + q];
                    }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);       // (no load of this path stays "pending" for hipcc past the epilogue: see above)
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
                    float* yp = p.y + out_pixel(row) * p.Cout + col;
                    if (full) *reinterpret_cast<f32x4_u*>(yp) = v;
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col + q < p.Cout) yp[q] = v[q];
                            else v[q] = 0.f;             // (keeps the column sums below clean)
                    }
                    if (STATS) {
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
        }
        if (STATS) {
            // the RPP row lanes of every column quad are combined through the LDS in a fixed order (deterministic)
            __syncthreads();                           // every thread is done reading the staged tile
            float* red = smem;                         // [RPP][2][BN]
            *reinterpret_cast<f32x4*>(red + (tr * 2 + 0) * BN + tc) = s1;
            *reinterpret_cast<f32x4*>(red + (tr * 2 + 1) * BN + tc) = s2;
            __syncthreads();
            if (tid < C4 && cok) {                     // (tid < C4  <=>  tr == 0: tc = 4 * tid)
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < RPP; ++k) {
                    a += *reinterpret_cast<const f32x4*>(red + (k * 2 + 0) * BN + tc);
                    b += *reinterpret_cast<const f32x4*>(red + (k * 2 + 1) * BN + tc);
                }
                float* st = p.stats + (size_t)(m0 / BM) * 2 * p.Cout;
                if (full) {
                    *reinterpret_cast<f32x4_u*>(st + col) = a;
                    *reinterpret_cast<f32x4_u*>(st + p.Cout + col) = b;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < p.Cout) {
                            st[col + q] = a[q];
                            st[p.Cout + col + q] = b[q];
                        }
                }
            }
            __syncthreads();                           // (the LDS goes back to the caller)
        }
    }
}

// ---- stream-K schedule ---------------------------------------------------------------------------------------
// The output tiles are first divided, whole, among the 8 XCD groups (group x = workgroups with blockIdx % 8 == x:
// one L2; a group's tiles are neighbours, so they share A halos / B panels); inside a group its G = workers/8
// workers own equal contiguous ranges of (tile, K-step) items.  A tile cut by a range boundary is finished INSIDE
// the kernel: the worker that owns the tile's later K-steps meets them FIRST in its range, writes its raw
// accumulators write-through and raises a flag; the worker that owns the tile's K-step 0 meets the tile LAST, adds
// the published accumulators in worker order (deterministic) and runs the normal epilogue.
// Local worker j runs in workgroup blockIdx = x + 8*(G-1-j): a consumer j waits only for workers j+1.. of its own
// group, which have SMALLER workgroup ids, i.e. were dispatched earlier - the wait cannot deadlock even when fewer
// than `workers` workgroups are resident.  (workers % 8 == 0, items of a group >= G.)
// (32-bit arithmetic: the launchers keep tiles * S below 2^31; the 64-bit division expansion costs ~40 registers
// where this runs, next to live accumulators)
// balanced contiguous partition of `items` over `workers`: worker w owns [begin(w), begin(w+1))
__device__ __host__ __forceinline__ int sk_begin32(int items, int workers, int w) {
    const int q = items / workers, r = items - q * workers;
    return w * q + (w < r ? w : r);
}
__device__ __host__ __forceinline__ void sk_range(int tiles, int S, int workers, int x, int j, long long& begin,
                                                  long long& end) {
    const int G = workers >> 3;
    const int t0 = sk_begin32(tiles, 8, x), t1 = sk_begin32(tiles, 8, x + 1);
    const int items = (t1 - t0) * S;
    const int q = items / G, r = items - q * G;
    begin = t0 * S + (j * q + (j < r ? j : r));
    end = t0 * S + ((j + 1) * q + (j + 1 < r ? j + 1 : r));
}
struct SkWorker {
    int grp, lw, id;     // XCD group, local worker in the group, global worker index (slot / flag index)
    long long begin, end;
};
__device__ __forceinline__ SkWorker sk_worker(int b, int tiles, int S, int workers) {
    SkWorker w;
    w.grp = b & 7;
    w.lw = (workers >> 3) - 1 - (b >> 3);
    w.id = w.grp * (workers >> 3) + w.lw;
    sk_range(tiles, S, workers, w.grp, w.lw, w.begin, w.end);
    return w;
}

typedef __attribute__((address_space(1))) unsigned gu32;   // flags: agent-scope global atomics only
typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));

// Producer: this worker's accumulators of a cut tile's later K-range -> its slot, 16 B per lane write-through
// (sc1: no release fence needed), every storing wave drains, then ONE lane raises the flag.
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void sk_publish(const ConvArgs& p, int worker,
                                           const f32x16 (&acc)[Geo<BM, BN, WGM, WGN>::MI][Geo<BM, BN, WGM, WGN>::NI]) {
    using G = Geo<BM, BN, WGM, WGN>;
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.partial, 0, (unsigned)((size_t)p.workers * BM * BN * 4), 0x00020000);
    const unsigned base = (unsigned)worker * (unsigned)(BM * BN * 4);
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // 16 bytes per lane: registers 4g..4g+3
                const f32x4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2],
                                 acc[mi][ni][4 * g + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_u32x4, v), rs,
                                                       (unsigned)((((mi * G::NI + ni) * 4 + g) * 256 + tid) * 16),
                                                       base, 16);   // aux 16 = sc1
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && !p.fault) {
        __hip_atomic_store((gu32*)(p.flags + worker), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Consumer: the worker that owns K-steps [0, k) of a cut tile (its LAST segment, ending at item `seg_end` inside
// the tile that ends at `tile_end`).  Counts the following workers of its group whose ranges start inside the tile,
// waits for their flags (one lane polls, one acquire), and adds their slots to `acc` in worker order.
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void sk_consume(const ConvArgs& p, const SkWorker& w, int tiles, int S, long long tile_end,
                                           f32x16 (&acc)[Geo<BM, BN, WGM, WGN>::MI][Geo<BM, BN, WGM, WGN>::NI]) {
    using G = Geo<BM, BN, WGM, WGN>;
    const int tid = threadIdx.x;
    // following workers whose range starts inside the tile; an empty range publishes nothing
    const int G8 = p.workers >> 3;
    const int t0 = sk_begin32(tiles, 8, w.grp), t1 = sk_begin32(tiles, 8, w.grp + 1);
    const int gitems = (t1 - t0) * S, gq = gitems / G8, gr = gitems - gq * G8;
    const int rel_end = (int)(tile_end - (long long)t0 * S);
    int n = 0;
    unsigned live = 0;   // bit e: worker w.id + 1 + e has a non-empty range (n <= 32 by the launcher's size rule)
    for (int jj = w.lw + 1; jj < G8; ++jj) {
        const int b = jj * gq + (jj < gr ? jj : gr), e = (jj + 1) * gq + (jj + 1 < gr ? jj + 1 : gr);
        if (b >= rel_end) break;
        n = jj - w.lw;
        live |= (b < e ? 1u : 0u) << (n - 1);
    }
    if (tid == 0) {
        for (int e = 0; e < n; ++e) {
            if (!((live >> e) & 1)) continue;
            gu32* flag = (gu32*)(p.flags + w.id + 1 + e);
            // bounded: the launch always ends.  On expiry this tile's sum is incomplete, so the failure is made LOUD:
            // a code is ORed (system scope) into the context's error word and the next call on the context, or
            // y3_ctx_check, returns Y3_EHIP
            unsigned spins = 0;
            for (; spins < p.spin_limit; ++spins) {
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (spins == p.spin_limit && p.err)
                __hip_atomic_fetch_or(p.err, Y3_ERR_STREAMK_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
        p.partial, 0, (unsigned)((size_t)p.workers * BM * BN * 4), 0x00020000);
    for (int e = 0; e < n; ++e) {
        if (!((live >> e) & 1)) continue;
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // system-scope loads (aux 17 = sc0 sc1): measured necessary - plain loads after the one-lane
                    // acquire still returned stale lines for consumers that arrive right when the flag flips
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs_c, (unsigned)((((mi * G::NI + ni) * 4 + g) * 256 + tid) * 16),
                        (unsigned)(w.id + 1 + e) * (unsigned)(BM * BN * 4), 17));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[mi][ni][4 * g + q] += v[q];
                    if (g & 1) __builtin_amdgcn_sched_barrier(0);   // two loads in flight: registers are scarce here
                }
    }
}

// ---- stem conv: 3x3, Cin = 3 -> COUT (=32), stride 1 ------------------------------------------------
// K = 27 is too short for the implicit-GEMM tile and the layer is HBM-write bound (709 MB out per bs=32 batch vs
// 9.6 GFLOP): one thread per output pixel.  Its 27 x COUT multiply-adds per pixel are packed (v_pk_fma_f32, two
// channels each) and take the weights as SGPR operands - a tap's 32 weights are wave-uniform scalar loads of the HWIO
// kernel [27][COUT].  (Round-3 measurements, profiles/r03_stem.txt: plain FMAs with the weights broadcast from the LDS
// 0.222 ms; packed FMAs alone 0.222 - the 216 16-byte LDS reads per pixel were the second bound; scalar weights 0.197;
// the same conv as 14 v_mfma_f32_32x32x2_f32 per 32 pixels with per-lane 4-byte patch loads 0.30-0.31.)
template <int COUT>
__global__ void __launch_bounds__(256) conv_stem_kernel(const ConvArgs p) {
    __shared__ float ssc[COUT], ssh[COUT];
    __shared__ __attribute__((aligned(16))) float xs[4 * 64 * (COUT + 4)];   // inputs [tap*3+ci][thread] (27*256
                                                                              // floats), later the output staging
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
        const float* wr = p.w + t * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 2)
            acc2[c / 2] = __builtin_elementwise_fma(x2, f32x2_t{wr[c], wr[c + 1]}, acc2[c / 2]);
    }
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = acc2[c / 2][c & 1];
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
// stream-K scratch: one accumulator slot per worker, then one flag word per worker
constexpr size_t SK_SLOT_BYTES = (size_t)128 * 128 * sizeof(float);
constexpr size_t SK_FLAGS_OFFSET = (size_t)SK_WORKERS * SK_SLOT_BYTES;
constexpr size_t SK_WORKSPACE_BYTES = SK_FLAGS_OFFSET + (size_t)SK_WORKERS * sizeof(unsigned);

// points the kernel arguments at the scratch; every polled word is zero when the kernel starts: either the caller
// hands in pre-zeroed words (sk->flags) or the words inside the workspace are zeroed here, ahead of the launch
inline int sk_prepare(hipStream_t stream, ConvArgs& a, void* workspace, const y3_sk_opts* sk) {
    a.partial = static_cast<float*>(workspace);
    a.workers = SK_WORKERS;
    a.err = sk ? sk->err : nullptr;
    y3_sk_debug_env(&a.spin_limit, &a.fault);
    if (sk && sk->flags) {
        a.flags = sk->flags;
    } else {
        a.flags = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + SK_FLAGS_OFFSET);
        Y3_CHECK_HIP(hipMemsetAsync(a.flags, 0, (size_t)SK_WORKERS * sizeof(unsigned), stream));
    }
    return Y3_OK;
}

// Schedule choice: stream-K for 3x3 convs on 128x128 tiles whose tile count is within a few multiples
// of the co-resident workgroup count (tail quantisation dominates there); Y3_CONV_STREAMK=0/1 overrides
// (experiment hook for tools/conv_bench.py).
inline bool use_streamk(const ConvArgs& a, int k, bool has_ws) {
    if (!has_ws || k != 3 || a.Cout < 128 || a.xu) return false;
    static int force = -2;
    if (force == -2) {
        const char* e = y3_exp_env("Y3_CONV_STREAMK");
        force = e ? atoi(e) : -1;
    }
    if (force == 0) return false;
    const int tiles = ((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    // every worker of every XCD group gets at least one (tile, K-step) item, and a tile is cut into < 32 ranges
    const long long S = (long long)(a.tmode ? a.ntaps : k * k) * (a.Cin / 32);
    if (tiles < 32 || (tiles / 8) * S < SK_WORKERS / 8) return false;   // (tiles/8 >= 4: a tile spans <= 18 ranges)
    if ((long long)tiles * S >= (1LL << 31)) return false;
    if (force == 1) return true;
    return tiles < 4 * SK_WORKERS;
}

}  // namespace y3conv
