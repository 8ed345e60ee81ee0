// Winograd F(2x2, 3x3) form of the stride-1 3x3 conv (+ folded BN + LeakyReLU + residual), exact fp32 arithmetic on
// v_mfma_f32_32x32x2_f32.  Replaces the same reference code as y3_conv.hip (utils/layer_utils.py:9-22,25-32) for the
// layers where the direct kernel is bound by the fp32 matrix pipe: 16 multiplies per 2x2 output tile and channel pair
// instead of 36 (2.25x less MFMA work).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, summed over input channels
//
//   * weights U = G g G^T are transformed once at load time and packed [16][Cin/8][Cout][8]
//     (y3_pack_conv_weights_wino): a K-step's B tile is 16 contiguous [Cout][8] slabs;
//   * a workgroup owns BT = 64 output tiles x 64 output channels for ALL 16 transform positions: each of its four
//     waves accumulates 16 independent 32(tiles) x 32(channels) products = 256 accumulator registers, so the kernel
//     runs one wave per SIMD with the accumulators in AGPRs;
//   * K-step = 8 input channels = 64 MFMAs (4096 cycles) per wave, ONE basic block: every thread stages a 1/256 share
//     of both operands - the 4x4 input patch of one (tile, channel pair) as 16 bounds-checked 8-byte loads (padding =
//     OOB = 0), B^T d B on the float2s (32 packed adds), 16 8-byte LDS writes, and eight 16-byte weight pieces - and
//     sched_group_barrier hints weave the loads of K-step s+1, the transform and the LDS writes into the gaps of the
//     MFMAs of K-step s (register prefetch, double-buffered LDS, fragment reads pipelined across the barrier);
//   * LDS rows are 32 bytes (8 channels) per (position, tile|channel), halves XOR-swizzled as in y3_conv_split.hip;
//   * epilogue: the 16 position sums of one (tile, channel) live in the same lane and register index of the 16
//     accumulator sets, so A^T M A is 24 adds per output tile in registers; then (through an LDS staging tile)
//     scale/shift, LeakyReLU, residual and 16-byte row-contiguous branch-free buffer stores;
//   * block counts that do not fill the last round of the 256 resident workgroups run a persistent stream-K schedule
//     whose cut blocks are finished inside the kernel (wk_range below).
// Numerics: the transforms use only additions and the constants 1/2, so the result differs from the direct sum by a
// few fp32 roundings per term (tests/test_conv_gpu.py holds it to the same 1e-4 tolerance against fp64).
#include <cstdlib>
#include <type_traits>
#include "y3_internal.h"

namespace {

struct WinoArgs {
    const float* x;      // [N,H,W,Cin]
    const float* u;      // packed [16][Cin/8][Cout][8]
    const float* scale;  // [Cout]
    const float* shift;  // [Cout]
    const float* resid;  // [N,H,W,Cout] or nullptr
    float* y;            // [N,H,W,Cout]
    int N, H, W, Cin, Cout, act;
    int TH, TW, T;       // 2x2 output tiles per image column / row, and in total
    float* partial;      // stream-K scratch: [workers][BT*4][BNW] output-space partial sums (pre scale/shift)
    unsigned* flags;     // stream-K scratch: [workers] "partial published" words, zeroed ahead of every launch
    int workers;         // stream-K grid size (0 = one workgroup per block)
    unsigned* err;       // stream-K: device-visible error word (a consumer whose poll expires ORs a code into it) or null
    unsigned spin_limit; // stream-K: polls per awaited flag before giving up
    int fault;           // stream-K test hook: producers skip raising their flag
    float* stats;        // STATS instantiations: [ceil(T/BT)][2][Cout] column sums of y, y^2 per 64-tile block
    int hybrid;          // stream-K: whole rounds of blocks first (see wk_range), only the remainder is cut
    int bn_inner;        // block order: 0 = column block outer (consecutive blocks share a weight panel), 1 = column block
                         // inner (the Cout/64 blocks of one tile block are neighbours and share its activations in the L2)
};

// Balanced contiguous partition of `items` over `workers` (same as y3_conv_common.h)
__device__ __host__ __forceinline__ long long wk_begin(long long items, int workers, int w) {
    const long long q = items / workers, r = items % workers;
    return (long long)w * q + (w < r ? w : r);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Stream-K work split.  The blocks are first divided, whole, among the 8 XCD groups (group x = workgroups with
// blockIdx % 8 == x: they share an L2, and a group's blocks share weight panels); inside a group its G = workers/8
// workers own equal contiguous ranges of (block, K-step) items.  A block cut by a range boundary is finished INSIDE
// the kernel (no fix-up launch): the worker that owns the block's later K-steps meets them first in its range, writes
// its output-space partial sums write-through and publishes a flag; the worker that owns the block's K-step 0 meets
// the block last, adds the published partials in worker order (deterministic) and runs the normal tail.
// Local worker j runs in workgroup blockIdx = x + 8*(G-1-j): every consumer j waits only for workers j+1.. of its own
// group, which have SMALLER workgroup ids, i.e. were dispatched earlier - the wait cannot deadlock even when fewer
// than `workers` workgroups are resident.
//
// HYBRID (p.hybrid; off by default, see the launcher): a group's first R*G blocks (R = its block count / G whole rounds) are NOT cut - in round r
// local worker j computes block b0 + r*G + j, start to end - and only the remaining < G blocks are divided as above.
// The G workers of an XCD then walk neighbouring blocks of one weight panel at the same K-step, so a weight slab
// (and the halo rows two neighbouring tile strips share) is fetched into that XCD's L2 once and hit by the others;
// with every block cut at a different K position (the plain split) each worker streams its own copy of the panel from
// the fabric: 862 MB per launch on the 13-grid layers against 45 MB of tensors (profiles/r02_pmc_layers.txt).
// The cut remainder runs LAST, so a consumer's producers have long published when it looks.
__device__ __host__ __forceinline__ void wk_range(int blocks, int ksteps, int workers, int x, int j, int hybrid,
                                                  long long& begin, long long& end, int* rounds = nullptr,
                                                  int* first_block = nullptr) {
    const int G = workers >> 3;
    const long long b0 = wk_begin(blocks, 8, x), b1 = wk_begin(blocks, 8, x + 1);
    const int R = hybrid ? (int)((b1 - b0) / G) : 0;
    const long long s0 = b0 + (long long)R * G;
    const long long items = (b1 - s0) * ksteps;
    begin = s0 * ksteps + wk_begin(items, G, j);
    end = s0 * ksteps + wk_begin(items, G, j + 1);
    if (rounds) *rounds = R;
    if (first_block) *first_block = (int)b0 + j;
}

typedef __attribute__((address_space(1))) unsigned gu32;   // flags are only ever touched by agent-scope global atomics

constexpr int WKC = 8;                   // input channels per K-step
constexpr int WROW = 32;                 // LDS bytes per row (8 floats)
constexpr unsigned OOB = 0x80000000u;

// Identity the optimiser cannot see through.  The per-block set-up code below is a function of threadIdx only; left
// visible, LICM hoists ~100 partial results out of the block loop and keeps them (spilled) across the K-loop.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ int lds_off(int row, int half) { return row * WROW + ((half ^ ((row >> 3) & 1)) << 4); }

// B^T d B on float4s (4 channels at once).  d[i][j], i = patch row, j = patch column; result v[i*4+j].
template <typename V>
__device__ __forceinline__ void input_transform(const V (&d)[16], V (&v)[16]) {
    V t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // rows: B^T d
        t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // columns: (B^T d) B
        v[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
        v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
        v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
        v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
    }
}

// The kernel's tail, in two steps so that the residual loads fly under the output transform: prepare() computes the output offsets of this thread's
// float4 rows of the staging tile cs[BT*4][LDC] and issues the residual loads; finish() reads the staged rows ->
// scale/shift, LeakyReLU, + residual -> global.  tile_pix / tile_ok describe the workgroup's tiles (see the kernel).
// NT = threads per workgroup; TWO: the staged sums are the element-wise sum of two staging tiles (cs and cs + BT*4*LDC:
// the eight-wave kernel's two position halves).
template <int BT, int BNW, int NT = 256, bool TWO = false>
struct WinoRows {
    static constexpr int LDC = BNW + 4;
    static constexpr int C4 = BNW / 4;            // float4 columns per staged row
    static constexpr int RPP = NT / C4;           // rows per pass
    static constexpr int PASSES = BT * 4 / RPP;
    static_assert(PASSES <= 32, "one validity bit per pass");
    unsigned off[PASSES];                         // element offsets (the launcher bounds M * Cout by 2^29)
    unsigned ok;
    f32x4 res[PASSES];

    __device__ __forceinline__ void prepare(const WinoArgs& p, const int* tile_pix, const int* tile_ok, int n0) {
        const int tid = opaque(threadIdx.x);
        const int co = n0 + (tid % C4) * 4, tr = tid / C4;
        const bool cok = co < p.Cout;           // Cout % 4 == 0
        ok = 0;
        // branch-free residual loads: rows that do not exist (and a null residual: zero records) read as 0
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.resid), 0, p.resid ? (unsigned)((size_t)p.N * p.H * p.W * p.Cout * 4) : 0u, 0x00020000);
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int rr = tr + i * RPP;                       // staged row = tile*4 + dy*2 + dx
            const int tl = rr >> 2, q = rr & 3;
            const int pix = tile_pix[tl];
            const int tok = tile_ok[tl];                       // (no short-circuit: a branch per pass otherwise)
            const bool oki = cok & (((tok >> q) & 1) != 0);
            ok |= (oki ? 1u : 0u) << i;
            off[i] = (unsigned)(pix + (q >> 1) * p.W + (q & 1)) * (unsigned)p.Cout + (unsigned)co;
            res[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, oki ? off[i] * 4u : OOB, 0, 0));
        }
    }

    // between(i) runs after the i-th row's store was issued: the tail is bound by store issue (~500 cycles per 1 KB
    // wave store), so independent register work placed there is free (the kernel resets its accumulators)
    // n_extra consecutive published partial-sum slots ([BT*4][BNW] floats each) starting `extra_base` bytes into
    // p.partial are added to the staged sums, in slot order, before scale/shift (the stream-K consumer; 0 elsewhere).
    // STATS: s1 / s2 receive the sums of y and y^2 over this thread's existing rows (training forward: batch-norm
    // statistics taken where the outputs are already in registers)
    template <bool STATS = false, typename F>
    __device__ __forceinline__ void finish(const WinoArgs& p, const float* cs, int n0, F between,
                                           int n_extra = 0, unsigned extra_base = 0, f32x4* s1 = nullptr,
                                           f32x4* s2 = nullptr) const {
        const int tid = threadIdx.x;
        const int tc = (tid % C4) * 4, tr = tid / C4;
        const int co = n0 + tc;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (co < p.Cout) {
            sc = *reinterpret_cast<const f32x4*>(p.scale + co);
            sh = *reinterpret_cast<const f32x4*>(p.shift + co);
        }
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            p.partial, 0, n_extra ? (unsigned)((size_t)p.workers * BT * 4 * BNW * 4) : 0u, 0x00020000);
        // branch-free stores: rows that do not exist go to an out-of-range offset, which the buffer store drops
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
            p.y, 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            f32x4 v = *reinterpret_cast<const f32x4*>(cs + (tr + i * RPP) * LDC + tc);
            if (TWO) v += *reinterpret_cast<const f32x4*>(cs + (BT * 4 + tr + i * RPP) * LDC + tc);
            for (int e = 0; e < n_extra; ++e)
                v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                         rs_x, (unsigned)((tr + i * RPP) * BNW + tc) * 4u,
                         extra_base + (unsigned)e * (unsigned)(BT * 4 * BNW * 4), 17));   // aux 17 = sc0 sc1:
                // system-scope loads; plain loads after the one-lane acquire were measured to return stale lines
                // when the consumer arrives right as the flag flips (y3_conv_common.h, sk_consume)
            v = v * sc + sh;
            if (p.act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
            }
            v += res[i];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_y,
                                                   ((ok >> i) & 1) ? off[i] * 4u : OOB, 0, 0);
            if (STATS) {
                const float m = ((ok >> i) & 1) ? 1.f : 0.f;      // rows that do not exist contribute nothing
                *s1 += v * m;
                *s2 += (v * v) * m;
            }
            between(i);
        }
    }
};

// t / d for 0 <= t < 2^24 and a quotient below 2^20: float reciprocal estimate (off by at most one) + correction; ~8 instructions
// instead of the ~45 of the integer division expansion (the per-block set-up runs four of these per thread)
__device__ __forceinline__ int fastdiv(int t, int d) {
    int q = (int)((float)t * __frcp_rn((float)d));
    const int r = t - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// pixel index of output (n, 2ty, 2tx) of tile t (-1: no such tile) and which of its 2x2 outputs exist
__device__ __forceinline__ void wino_tile_info(const WinoArgs& p, int t, int& pix, int& okbits, int& n, int& ty,
                                               int& tx) {
    n = ty = tx = 0;
    pix = -1;
    okbits = 0;
    if (t < p.T) {
        n = fastdiv(t, p.TH * p.TW);
        const int rem = t - n * p.TH * p.TW;
        ty = fastdiv(rem, p.TW);
        tx = rem - ty * p.TW;
        pix = (n * p.H + 2 * ty) * p.W + 2 * tx;
        okbits = 1 | ((2 * tx + 1 < p.W) ? 2 : 0) | ((2 * ty + 1 < p.H) ? 4 : 0) |
                 ((2 * tx + 1 < p.W && 2 * ty + 1 < p.H) ? 8 : 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The eight-wave kernel: a 64-tile x 64-channel block for all 16 transform positions, computed by 512
// threads - waves 0-3 own transform positions 0-7 (rows 0,1 of the 4x4 transform), waves 4-7 positions 8-15 - so a
// wave holds 128 accumulator registers and a SIMD runs TWO waves: while one waits for its loads, the barrier or its
// stores, the other one keeps the matrix pipe busy (the four-wave kernel measures 63 % pipe occupancy with one wave
// per SIMD: ~1,000 exposed cycles per 4,096-cycle K-step and a 16k-cycle tail per block).
//   * staging: thread = (position half, tile, channel pair) loads the three patch rows its two transform rows need
//     (12 x 8 bytes), forms X = e0 - e2 and Y = e1 + e2 (rows 0,1: d0 - d2, d1 + d2) or e1 - e0 (rows 2,3: Y = d2 - d1,
//     X = d1 - d3) - the half is the wave's position half, a compile-time constant on either side of the K-step's
//     phase branch -, the column transform, and writes 8 positions; four 16-byte weight pieces per thread;
//   * tail: each half applies A^T . A to its two rows of M (linear), the two partial 2x2 outputs go to two staging
//     tiles and are summed when the rows are read back.
template <bool STREAMK, bool STATS = false>
__global__ void __launch_bounds__(512, 1) conv_wino8_f32_kernel(const WinoArgs p) {
    constexpr int BT = 64, BNW = 64, NT = 512;
    constexpr int PLANE_V = BT * WROW, PLANE_U = BNW * WROW;
    constexpr int STAGE_V = 16 * PLANE_V, STAGE_U = 16 * PLANE_U;
    constexpr int LDC = BNW + 4;
    constexpr int CS_BYTES = BT * 4 * LDC * 4;            // one output staging tile
    static_assert(2 * CS_BYTES >= 2 * (STAGE_V + STAGE_U), "the tile tables sit behind the two staging tiles");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Vs = smem;                    // [2][16][BT][32 B]
    unsigned char* Us = smem + 2 * STAGE_V;      // [2][16][BNW][32 B]
    int* tile_pix = reinterpret_cast<int*>(smem + 2 * CS_BYTES);   // [BT]
    int* tile_ok = tile_pix + BT;                                  // [BT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int nbt = (p.T + BT - 1) / BT;
    const int ksteps = p.Cin / WKC;
    long long item, item_end;
    int worker = 0, grp = 0, lw = 0;
    int dp_left = 0, dp_blk = 0;
    const int nbn_ = (p.Cout + BNW - 1) / BNW;
    const int nblocks = nbt * nbn_;
    {
        const int nt = gridDim.x;
        const int q8 = nt >> 3, r8 = nt & 7, xcd = blockIdx.x & 7, k8 = blockIdx.x >> 3;
        const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
        if (STREAMK) {
            grp = xcd;
            lw = (p.workers >> 3) - 1 - k8;
            worker = grp * (p.workers >> 3) + lw;
            wk_range(nblocks, ksteps, p.workers, grp, lw, p.hybrid, item, item_end, &dp_left, &dp_blk);
        } else {
            item = (long long)id * ksteps;
            item_end = item + ksteps;
        }
    }
    if (dp_left == 0 && item >= item_end) return;
    const int first_blk = dp_left > 0 ? dp_blk : (int)(item / ksteps);
    const int first_ks = dp_left > 0 ? 0 : (int)(item - (long long)first_blk * ksteps);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.u), 0, (unsigned)((size_t)16 * p.Cin * p.Cout * 4), 0x00020000);
    unsigned voff_a[12];          // byte offsets of the 3x4 patch pixels this thread loads (OOB where padded)
    unsigned voff_u[4];           // byte offsets of the 4 weight pieces
    const int a_tile = (tid & 255) >> 2, a_pair = tid & 3, a_half = tid >> 8;
    int t0 = 0, n0 = 0;
    auto setup_tables = [&](int blk) {
        int bn, bt;
        if (p.bn_inner) { bt = fastdiv(blk, nbn_); bn = blk - bt * nbn_; }
        else            { bn = fastdiv(blk, nbt); bt = blk - bn * nbt; }
        t0 = bt * BT;
        n0 = bn * BNW;
        if (a_pair == 0 && a_half == 0) {
            int pix, okbits, n, ty, tx;
            wino_tile_info(p, t0 + a_tile, pix, okbits, n, ty, tx);
            tile_pix[a_tile] = pix;
            tile_ok[a_tile] = okbits;
        }
    };
    auto setup_voff = [&](int blk) {
        int bn, bt;
        if (p.bn_inner) { bt = fastdiv(blk, nbn_); bn = blk - bt * nbn_; }
        else            { bn = fastdiv(blk, nbt); bt = blk - bn * nbt; }
        const int t0 = bt * BT, n0 = bn * BNW;
        const int tid = opaque(threadIdx.x);
        {
            const int a_tile = (tid & 255) >> 2, a_pair = tid & 3, a_half = tid >> 8;
            int pix, okbits, n, ty, tx;
            wino_tile_info(p, t0 + a_tile, pix, okbits, n, ty, tx);
            const bool tok = pix >= 0;
            const int y0 = 2 * ty - 1 + a_half, x0 = 2 * tx - 1;       // patch rows a_half .. a_half + 2
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int yy = y0 + i, xx = x0 + j;
                    const bool ok = tok && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                    voff_a[i * 4 + j] = ok ? (unsigned)(((n * p.H + yy) * p.W + xx) * p.Cin + a_pair * 2) * 4u : OOB;
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + NT * j;
            const int pos = q / (2 * BNW), co = (q >> 1) % BNW, half = q & 1;
            const bool ok = (n0 + co) < p.Cout;
            voff_u[j] = ok ? (unsigned)(((size_t)pos * ksteps * p.Cout + (n0 + co)) * WKC + half * 4) * 4u : OOB;
        }
    };
    f32x2 ra[12];
    f32x4 ru[4];
    auto issue = [&](int ks) {
        const unsigned soff_a = (unsigned)(ks * WKC) * 4u;
        const unsigned soff_u = (unsigned)((size_t)ks * p.Cout * WKC) * 4u;
#pragma unroll
        for (int j = 0; j < 12; ++j)
            ra[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, voff_a[j], soff_a, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ru[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u[j], soff_u, 0));
    };
    // rows of the transform this thread produces: X -> row (half ? 3 : 0), Y -> row (half ? 2 : 1).  The half is the
    // wave's position half (a_half == ph: threads 256.. are waves 4-7), so it is a compile-time constant on each side of
    // the K-step's phase branch: no multiplications by 0 / +-1 are needed to keep the K-step branch-free.
    const int st_a = lds_off(a_tile, a_pair >> 1) + (a_pair & 1) * 8;
    auto store = [&](int buf, auto half_c) {
        constexpr int HALF = decltype(half_c)::value;
        unsigned char* us = Us + buf * STAGE_U;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + NT * j;
            const int pos = q / (2 * BNW), co = (q >> 1) % BNW, half = q & 1;
            *reinterpret_cast<f32x4*>(us + pos * PLANE_U + lds_off(co, half)) = ru[j];
        }
        f32x2 x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = ra[0 * 4 + j] - ra[2 * 4 + j];                                   // rows 0 / 3: d0 - d2 / d1 - d3
            y[j] = HALF ? ra[1 * 4 + j] - ra[0 * 4 + j] : ra[1 * 4 + j] + ra[2 * 4 + j];     // rows 1 / 2: d1 + d2 / d2 - d1
        }
        unsigned char* vx = Vs + buf * STAGE_V + st_a + (HALF ? 12 : 0) * PLANE_V;
        unsigned char* vy = Vs + buf * STAGE_V + st_a + (HALF ? 8 : 4) * PLANE_V;
        *reinterpret_cast<f32x2*>(vx + 0 * PLANE_V) = x[0] - x[2];
        *reinterpret_cast<f32x2*>(vx + 1 * PLANE_V) = x[1] + x[2];
        *reinterpret_cast<f32x2*>(vx + 2 * PLANE_V) = x[2] - x[1];
        *reinterpret_cast<f32x2*>(vx + 3 * PLANE_V) = x[1] - x[3];
        *reinterpret_cast<f32x2*>(vy + 0 * PLANE_V) = y[0] - y[2];
        *reinterpret_cast<f32x2*>(vy + 1 * PLANE_V) = y[1] + y[2];
        *reinterpret_cast<f32x2*>(vy + 2 * PLANE_V) = y[2] - y[1];
        *reinterpret_cast<f32x2*>(vy + 3 * PLANE_V) = y[1] - y[3];
    };

    f32x16 acc[8];
#pragma unroll
    for (int pos = 0; pos < 8; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pos][r] = 0.f;

    const int frag_a = lds_off(wm * 32 + (lane & 31), lane >> 5) + ph * 8 * PLANE_V;
    const int frag_b = lds_off(wn * 32 + (lane & 31), lane >> 5) + ph * 8 * PLANE_U;
    // fragments of position pair g of this wave's half (4 reads) and its 8 MFMAs (k pairs [j0, j1))
    auto frags = [&](int buf, int g, f32x4 (&a)[2], f32x4 (&b)[2]) {
        const unsigned char* vs = Vs + buf * STAGE_V + frag_a;
        const unsigned char* us = Us + buf * STAGE_U + frag_b;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const f32x4*>(vs + (g * 2 + i) * PLANE_V);
            b[i] = *reinterpret_cast<const f32x4*>(us + (g * 2 + i) * PLANE_U);
        }
    };
    auto mfmas = [&](int g, const f32x4 (&a)[2], const f32x4 (&b)[2], int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[g * 2 + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[i][j], acc[g * 2 + i], 0, 0, 0);
    };
    f32x4 a0[2], b0[2];            // pair 0 of the current K-step

    float* cs = reinterpret_cast<float*>(smem);
    setup_voff(first_blk);
    issue(first_ks);
    while (dp_left > 0 || item < item_end) {
        const bool whole = STREAMK && dp_left > 0;
        const int blk = whole ? dp_blk : (int)(item / ksteps);
        const int ks0 = whole ? 0 : (int)(item - (long long)blk * ksteps);
        const long long blk_end = (long long)(blk + 1) * ksteps;
        const long long seg_end = whole ? item : (blk_end < item_end ? blk_end : item_end);
        const int ks1 = whole ? ksteps : ks0 + (int)(seg_end - item);
        const bool whole_next = whole && dp_left > 1;
        const bool has_next = whole_next || seg_end < item_end;
        const int next_blk = whole_next ? blk + (p.workers >> 3) : (int)(seg_end / ksteps);
        const int next_ks = whole_next ? 0 : (int)(seg_end - (long long)next_blk * ksteps);
        setup_tables(blk);
        if (ph) store(0, std::integral_constant<int, 1>());
        else store(0, std::integral_constant<int, 0>());
        __syncthreads();
        frags(0, 0, a0, b0);
        // The two position halves run the K-step in OPPOSITE phase, so that one wave of every SIMD has MFMAs to issue
        // while the other one waits for its loads, transforms and writes the next K-step's tiles:
        //   ph 0: 16 MFMAs(ks) | transform + LDS writes(ks+1) | loads(ks+2) | 12 MFMAs(ks) | barrier | fragments(ks+1), last 4 MFMAs(ks)
        //   ph 1: transform + LDS writes(ks+1) | loads(ks+2) | 32 MFMAs(ks) | barrier | fragments(ks+1)
        // Both fetch a whole K-step ahead (the loads stay in flight across the barrier): L2 misses take 1-2 us here, half
        // a K-step is not enough.  (tools/issue_probe.hip: with two waves per SIMD, LDS and VMEM instructions are free
        // next to a stream of MFMAs, but every VALU instruction takes ~4.6 cycles of the matrix pipe.)
        // (a macro: sched_group_barrier wants literal counts) the wave's 16 loads go out two per MFMA under its first
        // position pair - back to back they hold the wave for ~900 cycles before its first MFMA - and the fragment reads
        // one pair ahead of their MFMAs; TAIL = MFMAs left after the three full pairs
#define W8_WEAVE(TAIL)                                                   \
    do {                                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);           \
            if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
        }                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
            if ((i & 7) >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
        }                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, TAIL, 0);            \
    } while (0)
        if (ph && ks0 + 1 < ks1) issue(ks0 + 1);
        for (int ks = ks0; ks + 1 < ks1; ++ks) {
            const int cur = (ks - ks0) & 1;
            f32x4 a1[2], b1[2];
            if (ph == 0) {
                issue(ks + 1);
                frags(cur, 1, a1, b1);
                mfmas(0, a0, b0, 0, 4);
                frags(cur, 2, a0, b0);
                mfmas(1, a1, b1, 0, 4);
                frags(cur, 3, a1, b1);
                mfmas(2, a0, b0, 0, 4);
                mfmas(3, a1, b1, 0, 2);
                W8_WEAVE(4);
                __builtin_amdgcn_sched_barrier(0);
                store(cur ^ 1, std::integral_constant<int, 0>());
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                frags(cur ^ 1, 0, a0, b0);
                mfmas(3, a1, b1, 2, 4);
            } else {
                store(cur ^ 1, std::integral_constant<int, 1>());
                __builtin_amdgcn_sched_barrier(0);
                // its loads go out under its MFMAs too (unconditional: on the last pass they re-read K-step ks+1 - in
                // range - and are never used; issued back to back after the staging they cost +4..11 %)
                issue(ks + 2 < ks1 ? ks + 2 : ks + 1);
                frags(cur, 1, a1, b1);
                mfmas(0, a0, b0, 0, 4);
                frags(cur, 2, a0, b0);
                mfmas(1, a1, b1, 0, 4);
                frags(cur, 3, a1, b1);
                mfmas(2, a0, b0, 0, 4);
                mfmas(3, a1, b1, 0, 4);
                W8_WEAVE(8);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                frags(cur ^ 1, 0, a0, b0);
            }
        }
        {
            const int cur = (ks1 - 1 - ks0) & 1;
            f32x4 a1[2], b1[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) frags(cur, g + 1, a1, b1);
                mfmas(g, a0, b0, 0, 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) { a0[i] = a1[i]; b0[i] = b1[i]; }
            }
        }

        // (1) this half's share of A^T M A per (tile, channel): 2x2 partial outputs -> its staging tile
        __syncthreads();
        const bool producer = STREAMK && ks0 > 0;
        WinoRows<BT, BNW, NT, true> rows;
        rows.prepare(p, tile_pix, tile_ok, n0);
        {
            const int col = wn * 32 + (lane & 31);
            float* csh = cs + ph * (BT * 4 * LDC);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float m[8];
#pragma unroll
                for (int pos = 0; pos < 8; ++pos) {
                    m[pos] = acc[pos][r];
                }
                // rows 0,1 of M (ph 0): s0 = m0 + m1, s1 = m1;  rows 2,3 (ph 1): s0 = m2, s1 = -m2 - m3
                float s0[4], s1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0[j] = ph ? m[j] : m[j] + m[4 + j];
                    s1[j] = ph ? -m[j] - m[4 + j] : m[4 + j];
                }
                float* row = csh + (tl * 4) * LDC + col;
                row[0 * LDC] = s0[0] + s0[1] + s0[2];
                row[1 * LDC] = s0[1] - s0[2] - s0[3];
                row[2 * LDC] = s1[0] + s1[1] + s1[2];
                row[3 * LDC] = s1[1] - s1[2] - s1[3];
            }
        }
        __syncthreads();
        int n_extra = 0;
        if (STREAMK && ks1 < ksteps) {
            const int G = p.workers >> 3;
            for (int jj = lw + 1; jj < G; ++jj) {
                long long b, e;
                wk_range(nblocks, ksteps, p.workers, grp, jj, p.hybrid, b, e);
                if (b >= blk_end) break;
                ++n_extra;
            }
            if (tid == 0) {
                for (int e = 0; e < n_extra; ++e) {
                    gu32* flag = (gu32*)(p.flags + worker + 1 + e);
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
        }
        if (STREAMK && has_next) {
            setup_voff(next_blk);
            issue(next_ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!producer) {
            static_assert(WinoRows<BT, BNW, NT, true>::PASSES == 8, "one accumulator set is reset per store pass");
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
            rows.template finish<STATS>(p, cs, n0, [&](int i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            }, n_extra, (unsigned)(worker + 1) * (unsigned)(BT * 4 * BNW * 4), &s1, &s2);
            if (STATS) {
                constexpr int C4 = BNW / 4, RPP = NT / C4;
                const int tc = (tid % C4) * 4, tr = tid / C4;
                __syncthreads();
                float* red = cs;                           // [RPP][2][BNW]
                *reinterpret_cast<f32x4*>(red + (tr * 2 + 0) * BNW + tc) = s1;
                *reinterpret_cast<f32x4*>(red + (tr * 2 + 1) * BNW + tc) = s2;
                __syncthreads();
                if (tid < C4 && n0 + tc < p.Cout) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < RPP; ++k) {
                        a += *reinterpret_cast<const f32x4*>(red + (k * 2 + 0) * BNW + tc);
                        b += *reinterpret_cast<const f32x4*>(red + (k * 2 + 1) * BNW + tc);
                    }
                    float* st = p.stats + (size_t)(t0 / BT) * 2 * p.Cout;
                    *reinterpret_cast<f32x4*>(st + n0 + tc) = a;
                    *reinterpret_cast<f32x4*>(st + p.Cout + n0 + tc) = b;
                }
                if (!STREAMK) __syncthreads();
            }
        } else {
#pragma unroll
            for (int pos = 0; pos < 8; ++pos)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pos][r] = 0.f;
            const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
                p.partial, 0, (unsigned)((size_t)p.workers * BT * 4 * BNW * 4), 0x00020000);
            const unsigned slot_off = (unsigned)worker * (unsigned)(BT * 4 * BNW * 4);
            constexpr int C4 = BNW / 4;
            for (int f = tid; f < BT * 4 * C4; f += NT) {
                const int rr = f / C4, c4 = f - rr * C4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(cs + rr * LDC + c4 * 4) +
                                *reinterpret_cast<const f32x4*>(cs + (BT * 4 + rr) * LDC + c4 * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_part,
                                                       slot_off + (unsigned)f * 16u, 0, 16);   // aux 16 = sc1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !p.fault)
                __hip_atomic_store((gu32*)(p.flags + worker), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (STREAMK) __syncthreads();
        if (whole) { --dp_left; dp_blk += p.workers >> 3; }
        item = seg_end;
    }
}

// U = G g G^T for every (ci, co), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; out[pos][ci/8][co][ci%8]
// dgrad != 0: (cin, cout) are those of the GRADIENT conv (cin = dz channels, cout = the forward layer's Cin) and w is
// the forward kernel stored [3*3][cout][cin] (= [tap][fwd Cin][dz_stride]); the gradient conv's kernel is the forward
// one with its taps flipped and its channel axes swapped: g'[a][b](ci, co) = w[(2-a, 2-b)][co][ci].
// A workgroup owns one K-step slab (8 input channels) x 32 output channels: the nine taps go through the LDS so that
// both the reads (runs along the source's contiguous axis) and the writes (1 KB contiguous per transform position:
// [co][8]) are coalesced — the training step re-packs the forward and the data-gradient kernels of 32 layers every step.
__global__ void __launch_bounds__(256) pack_weights_wino_kernel(const float* __restrict__ w_hwio, float* __restrict__ out,
                                                                int cin, int cout, int dgrad) {
    __shared__ float g[9][WKC][33];
    const int tid = threadIdx.x;
    const int ncb = (cout + 31) / 32;
    const int kb = blockIdx.x / ncb, cb = blockIdx.x - kb * ncb;      // K-step slab, 32-channel column block
    const int ci0 = kb * WKC, co0 = cb * 32;
    if (!dgrad) {            // source [tap][cin][cout]: 32 consecutive co per (tap, ci)
        const int col = tid & 31, cil = tid >> 5;
        const bool ok = co0 + col < cout && ci0 + cil < cin;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            g[t][cil][col] = ok ? w_hwio[((size_t)t * cin + ci0 + cil) * cout + co0 + col] : 0.f;
    } else {                 // source [tap][cout][cin], taps flipped: 8 consecutive ci per (tap, co)
        const int cil = tid & 7, col = tid >> 3;
        const bool ok = co0 + col < cout && ci0 + cil < cin;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            g[t][cil][col] = ok ? w_hwio[((size_t)(8 - t) * cout + co0 + col) * cin + ci0 + cil] : 0.f;
    }
    __syncthreads();
    const int cil = tid & 7, col = tid >> 3;           // writer: (channel-in-slab fastest, then co) = the packed order
    if (co0 + col >= cout || ci0 + cil >= cin) return;
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float g0 = g[0 * 3 + b][cil][col], g1 = g[1 * 3 + b][cil][col], g2 = g[2 * 3 + b][cil][col];
        t[0][b] = g0;
        t[1][b] = 0.5f * (g0 + g1 + g2);
        t[2][b] = 0.5f * (g0 - g1 + g2);
        t[3][b] = g2;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                            t[a][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b)
            out[(((size_t)(a * 4 + b) * (cin / WKC) + kb) * cout + co0 + col) * WKC + cil] = u[b];
    }
}

}  // namespace

void y3_wino_range_impl(int units, int ksteps, int workers, int group, int local_worker, int hybrid, long long* begin,
                        long long* end) {
    wk_range(units, ksteps, workers, group, local_worker, hybrid, *begin, *end);
}

int y3_conv_wino_eligible_impl(const y3_conv_desc* d) {
    return d && d->k == 3 && d->stride == 1 && d->c_up == 0 && d->cin % 32 == 0 && d->cout % 32 == 0 &&
           d->n > 0 && d->h > 1 && d->w > 1;
}

int y3_launch_pack_wino(hipStream_t stream, const float* w_hwio, int cin, int cout, float* out, int dgrad) {
    const int blocks = (cin / WKC) * ((cout + 31) / 32);       // cin % 8 == 0 (checked by the callers)
    hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(blocks), dim3(256), 0, stream, w_hwio, out, cin, cout, dgrad);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

constexpr int WK_WORKERS = 256;     // one persistent workgroup per CU (139 KB of LDS, eight waves)

// stream-K scratch: one partial-sum slot per worker, then one flag word per worker
constexpr size_t WK_SLOT_BYTES = (size_t)64 * 4 * 64 * sizeof(float);
constexpr size_t WK_FLAGS_OFFSET = (size_t)WK_WORKERS * WK_SLOT_BYTES;

size_t y3_conv_wino_workspace_bytes_impl(const y3_conv_desc* d) {
    if (!y3_conv_wino_eligible_impl(d)) return 0;
    return WK_FLAGS_OFFSET + (size_t)WK_WORKERS * sizeof(unsigned);
}

int y3_launch_conv_wino(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* u, const float* scale,
                        const float* shift, const float* residual, float* y, void* workspace, size_t workspace_bytes,
                        const y3_sk_opts* sk) {
    Y3_CHECK_ARG(d && x && u && scale && shift && y, "y3_conv2d_fwd_wino: null pointer argument");
    Y3_CHECK_ARG(y3_conv_wino_eligible_impl(d),
                 "y3_conv2d_fwd_wino: needs a 3x3 stride-1 conv with Cin %% 32 == 0, Cout %% 32 == 0 and no "
                 "fused upsample input");
    const long long M = (long long)d->n * d->h * d->w;
    Y3_CHECK_ARG(M * d->cin < (1LL << 29) && M * d->cout < (1LL << 29),
                 "y3_conv2d_fwd_wino: tensor exceeds 2^29 elements (32-bit byte offsets)");
    WinoArgs a;
    a.x = x; a.u = u; a.scale = scale; a.shift = shift; a.resid = residual; a.y = y;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.act = d->act;
    a.TH = (d->h + 1) / 2; a.TW = (d->w + 1) / 2; a.T = d->n * a.TH * a.TW;
    a.partial = nullptr; a.flags = nullptr; a.workers = 0;
    a.err = nullptr; a.spin_limit = 0; a.fault = 0;
    a.stats = sk ? sk->stats : nullptr;
    {
        // Y3_WINO_SK_HYBRID=1: whole rounds of blocks first (wk_range).  Measured (profiles/r02_wino_hybrid_schedule.txt):
        // fabric reads of the 13-grid layers 862 -> 294 MB per launch, time +1..6 % on every stream-K layer (the
        // re-reads are Infinity-Cache hits and off the critical path; workgroups in lock-step collide in their store
        // tails) - so the default stays the plain split.
        static int hyb = -1;
        if (hyb < 0) {
            const char* e = y3_exp_env("Y3_WINO_SK_HYBRID");
            hyb = e ? (atoi(e) != 0) : 0;
        }
        a.hybrid = hyb;
    }
    // Block order.  Column block OUTER streams the whole activation tensor once per 64-channel column block (Cout/64
    // passes that miss the 4 MB L2 of an XCD); column block INNER reads each tile block's activations once and re-reads
    // the weight panels per tile block instead — the better trade only while all panels (16*Cin*Cout floats) are small
    // next to the 4 MB L2: measured (FETCH_SIZE, profiles/r02_wino_block_order.txt) -17 % fabric fetch on the 104-grid
    // layers (0.5 MB of panels), +20 % on the 52-grid layers (2 MB), time unchanged either way.
    // Y3_WINO_ORDER=0/1 overrides (experiment hook).
    {
        static int force = -2;
        if (force == -2) {
            const char* e = y3_exp_env("Y3_WINO_ORDER");
            force = e ? atoi(e) : -1;
        }
        a.bn_inner = force >= 0 ? (force != 0) : ((size_t)16 * d->cin * d->cout * sizeof(float) <= (size_t)(1u << 20));
    }
    constexpr int BT = 64, BNW = 64;
    constexpr int workers = WK_WORKERS;
    constexpr size_t lds = (size_t)2 * BT * 4 * (BNW + 4) * sizeof(float) + 2 * BT * sizeof(int);
    auto kern = a.stats ? conv_wino8_f32_kernel<false, true> : conv_wino8_f32_kernel<false, false>;
    auto kern_sk = a.stats ? conv_wino8_f32_kernel<true, true> : conv_wino8_f32_kernel<true, false>;
    {
        static bool attr_set[2][Y3_MAX_DEVICES] = {};   // per pair of instantiations and device; benign race (idempotent)
        const int slot = a.stats ? 1 : 0;
        const int dev_ = y3_current_device();
        if (dev_ < 0 || !attr_set[slot][dev_]) {
            Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_sk),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev_ >= 0) attr_set[slot][dev_] = true;
        }
    }
    const int nbt = (a.T + BT - 1) / BT, nbn = (a.Cout + BNW - 1) / BNW;
    const int blocks = nbt * nbn;
    // stream-K when the block count is a small non-integer multiple of the resident workgroups (the last round
    // would run partly empty); Y3_CONV_WINO_STREAMK=0/1 overrides (experiment hook)
    static int force = -2;
    if (force == -2) {
        const char* e = y3_exp_env("Y3_CONV_WINO_STREAMK");
        force = e ? atoi(e) : -1;
    }
    const bool has_ws = workspace != nullptr && workspace_bytes >= y3_conv_wino_workspace_bytes_impl(d) &&
                        ((uintptr_t)workspace & 15) == 0;
    bool use_sk = has_ws && blocks > workers && blocks < 8 * workers && blocks % workers != 0;
    if (force >= 0) use_sk = has_ws && force != 0 && blocks >= workers;     // (>= workers: no worker range is empty)
    if (use_sk) {
        a.partial = static_cast<float*>(workspace);
        a.workers = workers;
        a.err = sk ? sk->err : nullptr;
        y3_sk_debug_env(&a.spin_limit, &a.fault);
        if (sk && sk->flags) {
            a.flags = sk->flags;       // pre-zeroed by the caller (y3_net_forward: one memset per forward)
        } else {
            // every polled word is zeroed ahead of the launch (no state is assumed in the caller's workspace)
            a.flags = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + WK_FLAGS_OFFSET);
            Y3_CHECK_HIP(hipMemsetAsync(a.flags, 0, (size_t)workers * sizeof(unsigned), stream));
        }
        hipLaunchKernelGGL(kern_sk, dim3(workers), dim3(512), lds, stream, a);
    } else {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, stream, a);
    }
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
