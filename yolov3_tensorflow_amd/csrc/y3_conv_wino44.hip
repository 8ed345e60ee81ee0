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
//   * weights U = G g G^T transformed once at load time and packed [18 position pairs][Cin/8][Cout][4 channel pairs][2 positions]
//     [2 channels] (y3_pack_conv_weights_wino44): a lane's fragments of two neighbouring positions are 16 contiguous bytes;
//   * a workgroup = FOUR waves owns 16 tiles x 64 output channels for ALL 36 transform positions (BT = 16: 72 KB of LDS, 256
//     registers per wave -> TWO workgroups per CU, one's prologue / transform / store tail under the other's MFMAs: round 4,
//     -9 ... -24 % per layer shape against the 32-tile, eight-wave block of round 3); wave wn holds 16 tiles x 16 channels
//     of every position as v_mfma_f32_16x16x4_f32 accumulators (36 x 4 = 144 registers), so the 36 position sums of one
//     (tile, channel) sit in ONE lane and A^T M A needs no exchange between waves at all;
//   * K-step = 8 input channels = 72 MFMAs per wave, ONE barrier.  Raw 6x6 patches go global -> LDS by DMA
//     (buffer_load ... lds: no registers, padding = out-of-range lanes = zeros), two K-steps ahead; inside a K-step thread
//     (tile, channel pair, job) reads the patch rows its job needs from the LDS, transforms them on float2s and writes one
//     or two rows of B^T d B for the NEXT K-step (jobs: row 0 | rows 1,2 | rows 3,4 | row 5 - paired rows share their first
//     pass); then the MFMAs of THIS K-step: activation fragments from the LDS four positions ahead, weight fragments -
//     a lane's 16 bytes for two positions, 1 KB contiguous per wave load - straight from global memory through a rolling
//     window of six loads (twelve positions) that runs on across K-step boundaries.  The order is pinned with scheduling
//     barriers: left alone, hipcc moves every fragment read right in front of its MFMAs;
//   * V = B^T d B planes are [channel pair][tile][2 channels] with the tile index rotated by 4 per channel pair (v_off):
//     hipcc emits the fragment reads as ds_read2st64_b64, served in 16-lane groups over 32 banks - as [tile][8 channels] rows
//     they hit every bank four times (SQ_LDS_BANK_CONFLICT 3.8e7 per launch in round 3, 1.0e6 now);
//   * lane quarter q = lane / 16 reads channels 2q, 2q+1 of both operands (one ds_read_b64 / one 8-byte load) and MFMA
//     m = 0, 1 consumes channel 2q + m - which of the 8 channels plays "k" where is free as long as both operands agree;
//   * tail: A^T M A per accumulator register in registers (120 adds / multiplies by 2, 4, 8 per 4x4 tile), staged through
//     the LDS ([tile][pixel][64 channels]) so that scale / shift, LeakyReLU and the residual run on 16-byte pieces and an
//     output pixel's 64 channels leave as 256 contiguous bytes; the sixteen residual loads of a thread are issued together
//     ahead of the store loop; STATS instantiation: column sums of y and y^2 per block for the training forward's batch norm;
//   * with a workspace: persistent schedule (whole rounds of blocks, the remaining blocks cut along K and finished inside
//     the kernel, same hand-off as y3_conv_wino.hip).  y3_net_forward does not use it (profiles/r03_wino44.txt and
//     profiles/r04_wino44.txt have the measurements, and those of every variant tried on the way).
//   Build-time switches (-D, tools/build_variant.py; never set in the product): W44_BT=32 (round 3's block), W44_PROBE (s_memtime
//   stamps for tools/wino44_probe.py), W44_KO_* (knock-outs), W44_MIDPOS / W44_BDEPTH / W44_AD / W44_SK_KEEP / W44_DMA_HALF (variants
//   measured in profiles/r04_wino44.txt).
#include <cstdlib>
#include "y3_internal.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    // persistent schedule (workers > 0): whole rounds of blocks first, the remaining < workers blocks cut along K
    float* partial;      // [workers][2 * 256 rows][64] output-space partial sums (pre scale / shift) of the cut blocks' later K-ranges
    unsigned* flags;     // [workers] "partial published" words, zeroed ahead of every launch
    unsigned* err;       // device-visible error word (a consumer whose poll expires ORs a code into it) or null
    unsigned spin_limit; // polls per awaited flag before giving up
    int workers;         // grid size of the persistent schedule (0 = one workgroup per block)
    int fault;           // test hook: producers skip raising their flag
    float* stats;        // STATS instantiation: [tile blocks][2][Cout] column sums of y and y^2 per 16-tile block (one workgroup
                         // per block schedule only), the training forward's batch-norm statistics (y3_bn_train_stats_partials)
};

#ifndef W44_BT
#define W44_BT 16
#endif
constexpr int BT = W44_BT, BNC = 64, NTH = BT * 16;       // one wave per 16 tiles x 16 channels
constexpr int NW = NTH / 64;
constexpr int KC = 8;                          // input channels per K-step
constexpr int ROWB = KC * 4;                   // LDS bytes per (position, tile) row
constexpr int PLANE = BT * ROWB;               // one position's tiles
constexpr int STAGE = 36 * PLANE;              // 36,864 B
constexpr unsigned OOB = 0x80000000u;
#ifndef W44_BDEPTH
#define W44_BDEPTH 6
#endif
constexpr int BDEPTH = W44_BDEPTH;                      // weight fragment loads in flight per wave: one load = one lane's 16 bytes for a PAIR of
                                               // positions (divides 18: the window runs on across K-steps)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) unsigned gu32;   // flags are only ever touched by agent-scope global atomics
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int SLOT_FLOATS = BT * 16 * BNC;      // one partial-sum slot: 32 tiles x 16 pixels x 64 channels

// Balanced contiguous partition of `items` over `parts`
__device__ __host__ __forceinline__ long long part_begin(long long items, int parts, int i) {
    const long long q = items / parts, r = items % parts;
    return (long long)i * q + (i < r ? i : r);
}

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

// all six outputs of B^T applied along a 6-vector, written as row i of V = B^T d B (6 position planes)
__device__ __forceinline__ void put_row(const f32x2 (&e)[6], unsigned char* vs, int i) {
    f32x2 o[6];
    o[0] = bt0(e[0], e[2], e[4]);
    const f32x2 p = e[4] - 4.f * e[2], q = e[3] - 4.f * e[1];
    o[1] = p + q;
    o[2] = p - q;
    const f32x2 p2 = e[4] - e[2], q2 = 2.f * (e[3] - e[1]);
    o[3] = p2 + q2;
    o[4] = p2 - q2;
    o[5] = bt5(e[1], e[3], e[5]);
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

#ifdef W44_PROBE
// Clock probe build (tools/wino44_probe.py; never in the product library): wave 0 of every workgroup stamps s_memtime at
// its phase boundaries into g_w44_probe[block][16].
__device__ unsigned long long* g_w44_probe = nullptr;
#define W44_STAMP(i) do { if (g_w44_probe && tid == 0) g_w44_probe[(size_t)blockIdx.x * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define W44_STAMP(i) do { } while (0)
#endif

template <bool STATS>
__global__ void __launch_bounds__(NTH, BT == 16 ? 2 : 1) conv_wino44_f32_kernel(const W44Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // V: [2][36][BT][32 B], then raw patches: the same shape
    constexpr int RAW_OFF = 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;             // 2 x 4 waves: 16 tiles x 16 channels each
    const int nbn = p.Cout / BNC;
    const int nbt = (p.T + BT - 1) / BT;
    const int ksteps = p.Cin / KC;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.u), 0, (unsigned)((size_t)36 * p.Cin * p.Cout * 4), 0x00020000);

    // ---- schedule ---------------------------------------------------------------------------------------------------
    // workers == 0: workgroup b owns block b.  Persistent schedule: worker w runs the blocks w, w + W, ... of the R whole
    // rounds start to end; the remaining < W blocks are divided into W equal ranges of (block, K-step) items.  Range index
    // W-1-w goes to worker w, so the piece holding a block's K-step 0 - its OWNER, which adds the others' output-space
    // partial sums and runs the tail - has the highest workgroup id of the block's pieces: it waits only for workgroups
    // dispatched before it, and every worker meets its producer piece (the tail of a block) before its owner piece.
    const int W = p.workers;
    const int nblocks = nbt * nbn;
    int whole_left = 1, whole_blk = blockIdx.x;
    long long lo = 0, hi = 0;
    int rem0 = 0;
    if (W > 0) {
#ifdef W44_SK_KEEP
        const int R = nblocks / W > W44_SK_KEEP ? nblocks / W - W44_SK_KEEP : 0;     // (probe) cut the last W44_SK_KEEP whole rounds too
#else
        const int R = nblocks / W;
#endif
        whole_left = R;
        rem0 = R * W;
        const long long items = (long long)(nblocks - rem0) * ksteps;
        const int ri = W - 1 - (int)blockIdx.x;
        lo = part_begin(items, W, ri);
        hi = part_begin(items, W, ri + 1);
    }

    // ---- per-thread constants ---------------------------------------------------------------------------------------
    const int unit = tid & (BT * 4 - 1), job = __builtin_amdgcn_readfirstlane(tid / (BT * 4));
    const int st_off = (unit >> 2) * ROWB + (unit & 3) * 8;          // staging job: (tile, channel pair) inside a RAW plane ([tile][8 channels])
    const int row16 = lane & 15, quart = lane >> 4;
    // V planes are [channel pair][32 tiles][2 channels] with the tile index rotated by 4 per channel pair (v_off): the
    // fragment reads come out of hipcc as ds_read2st64_b64, which the LDS serves in groups of 16 lanes over 32 banks -
    // 16 tiles x 8 B of ONE channel pair must be 128 contiguous bytes (as [tile][8 channels] rows they hit every bank
    // four times: 3.5k conflict cycles per K-step, profiles/r03_pmc_layers.txt) - and the rotation keeps the staging
    // writes of a 16-lane group (4 tiles x 4 channel pairs) on 32 different banks as well.
#ifdef W44_OLD_LAYOUT
    auto v_off = [](int tile, int cp) { return tile * ROWB + cp * 8; };
#else
    auto v_off = [](int tile, int cp) { return cp * (BT * 8) + ((tile * 8 + cp * 32) & (BT * 8 - 1)); };
#endif
    const int sv_off = v_off(unit >> 2, unit & 3);
    const int a_off = v_off(wm * 16 + row16, quart);                 // activation fragment inside a position plane
    const unsigned b_pos_stride = (unsigned)((size_t)ksteps * p.Cout * 2 * KC * 4);         // bytes between position pairs
    const unsigned b_ks_stride = (unsigned)(p.Cout * 2 * KC * 4);
    constexpr int RS = BNC + 4;                                      // staged output row stride in floats
    float* cs = reinterpret_cast<float*>(smem) + wm * (16 * 16 * RS);
    int* tinfo = reinterpret_cast<int*>(smem + (BT / 16) * 16 * 16 * RS * 4);     // [BT][8]: pixel-index parts of the tile's 4 output rows, 4 output columns (-1: none)

    W44_STAMP(0);
    while (whole_left > 0 || lo < hi) {
        // ---- this segment: block, K-range, role ------------------------------------------------------------------------
        int blk, ks0, ks1;
        if (whole_left > 0) {
            blk = whole_blk; ks0 = 0; ks1 = ksteps;
            --whole_left; whole_blk += W;
        } else {
            const int br = (int)(lo / ksteps);
            ks0 = (int)(lo - (long long)br * ksteps);
            const long long left = hi - lo;
            ks1 = (ksteps - ks0 < left) ? ksteps : ks0 + (int)left;
            blk = rem0 + br;
            lo += ks1 - ks0;
        }
        const bool producer = ks0 > 0;
        int n_extra = 0;                                 // pieces of this block other workers publish (owner of a cut block)
        if (W > 0 && !producer && ks1 < ksteps) {
            const long long items = (long long)(nblocks - rem0) * ksteps;
            const long long blk_end = (long long)(blk - rem0 + 1) * ksteps;
            for (int jj = W - (int)blockIdx.x; jj < W; ++jj) {          // range indices after this worker's
                if (part_begin(items, W, jj) >= blk_end) break;
                ++n_extra;
            }
        }
        const int bt = blk / nbn, bn = blk - bt * nbn;     // the Cout/64 blocks of one tile block are neighbours
        const int t0 = bt * BT, n0 = bn * BNC;

        // raw patches, global -> LDS by DMA: plane i = patch pixel (k, l) = (i / 6, i % 6) holds [BT tiles][8 channels];
        // one instruction moves 1 KB = PPI planes, wave w issues instructions w, w + NW, ...: lane = (plane, tile, 16-byte half)
        constexpr int PPI = 1024 / PLANE, NDMA = 36 / PPI;
#ifdef W44_DMA_HALF
        constexpr int DW = NW / 2;        // only the first half of the waves issues DMAs (the half that transforms first)
#else
        constexpr int DW = NW;
#endif
        constexpr int DPW = (NDMA + DW - 1) / DW;
        unsigned dvoff[DPW];
        {
            const int t = t0 + ((lane >> 1) & (BT - 1));
            const bool tok = t < p.T;
            const int n = p.mr ? 0 : t / (p.TH * p.TW);
            const int r = t - n * p.TH * p.TW;
            const int ty = r / p.TW, tx = r - ty * p.TW;
#pragma unroll
            for (int j = 0; j < DPW; ++j) {
                const int i = (wave + DW * j) * PPI + lane / (2 * BT);
                const int k = i / 6, l = i - 6 * k;
                int yy = 4 * ty - 1 + k, xx = 4 * tx - 1 + l, img = n;
                bool ok = tok && i < 36 && yy >= 0 && xx >= 0;
                if (p.mr) {
                    // mosaic row yy = row yy % (H+1) of image row yy / (H+1); row H of every image slot is the zero gap, which
                    // is at once the bottom padding of the image above and the top padding of the image below
                    const int ry = yy / (p.H + 1), cx = xx / (p.W + 1);
                    yy -= ry * (p.H + 1);
                    xx -= cx * (p.W + 1);
                    img = ry * p.mc + cx;
                    ok = ok && ry < p.mr && cx < p.mc;
                }
                ok = ok && yy < p.H && xx < p.W;
                dvoff[j] = ok ? (unsigned)((((img * p.H + yy) * p.W + xx) * p.Cin) * 4 + (lane & 1) * 16) : OOB;
            }
        }
        auto dma_raw = [&](int ks, int buf) {
            const unsigned so = (unsigned)(ks * KC) * 4u;
#pragma unroll
            for (int j = 0; j < DPW; ++j)
                if (wave < DW && wave + DW * j < NDMA) dma16(rs_x, smem + RAW_OFF + buf * STAGE + (wave + DW * j) * 1024, dvoff[j], so);
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

        // ---- prologue: raw(ks0), raw(ks0+1) by DMA; V(ks0) = transform(raw(ks0)); the first weight fragments ---------------
        W44_STAMP(1);
        dma_raw(ks0, 0);
        if (ks0 + 1 < ks1) dma_raw(ks0 + 1, 1);
#pragma unroll
        for (int s = 0; s < BDEPTH; ++s) issue_b(s, s, ks0);      // pairs 0 .. BDEPTH-1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BDEPTH) : "memory");     // the DMAs are older than the fragment loads
        __builtin_amdgcn_s_barrier();
        W44_STAMP(2);
        transform(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        W44_STAMP(3);

        for (int ks = ks0; ks < ks1; ++ks) {
            const int cur = (ks - ks0) & 1;
            const bool more = ks + 1 < ks1;
#ifdef W44_PROBE
            const bool pk = ks == ks0 + 3;
            if (pk) W44_STAMP(16);
#endif
            // DMA of raw(ks+2) (raw[cur] held raw(ks): consumed a K-step ago) and V(ks+1) from raw(ks+1) (it landed before the
            // last barrier; on the last K-step it transforms stale data into a buffer nobody reads: keeps the K-step's shape)
            auto stage_next = [&]() {
#ifndef W44_KO_DMA
                if (ks + 2 < ks1) dma_raw(ks + 2, cur);
#endif
#ifndef W44_KO_TRANSFORM
                transform(cur ^ 1, cur ^ 1);
#endif
            };
#ifndef W44_MIDPOS
#define W44_MIDPOS -1
#endif
            if (W44_MIDPOS < 0) stage_next();
#ifdef W44_PROBE
            if (pk) W44_STAMP(17);
#endif
            const unsigned char* vs = smem + cur * STAGE + a_off;
#ifndef W44_AD
#define W44_AD 4
#endif
            constexpr int AD = W44_AD;                             // activation fragments read ahead (two pairs)
            f32x2 aq[AD];
#pragma unroll
            for (int s = 0; s < AD; ++s) aq[s] = *reinterpret_cast<const f32x2*>(vs + s * PLANE);
#pragma unroll
            for (int pr = 0; pr < 18; ++pr) {
                if (pr == W44_MIDPOS) {
                    stage_next();
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x2 a0 = aq[(2 * pr) % AD], a1 = aq[(2 * pr + 1) % AD];
                const f32x4 b = bq[pr % BDEPTH];
#ifndef W44_KO_AFRAG
                if (2 * pr + AD < 36)
#else
                if (false)
#endif
                {
                    aq[(2 * pr) % AD] = *reinterpret_cast<const f32x2*>(vs + (2 * pr + AD) * PLANE);
                    aq[(2 * pr + 1) % AD] = *reinterpret_cast<const f32x2*>(vs + (2 * pr + 1 + AD) * PLANE);
                }
                // refill the slot with the pair BDEPTH pairs ahead (it runs on into the next K-step; past the last K-step it
                // re-reads a valid address and is never used)
#ifndef W44_KO_B
                {
                    const int np = pr + BDEPTH;
                    if (np < 18) issue_b(pr % BDEPTH, np, ks);
                    else issue_b(pr % BDEPTH, np - 18, more ? ks + 1 : ks);
                }
#endif
                acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b[0], acc[2 * pr], 0, 0, 0);
                acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b[2], acc[2 * pr + 1], 0, 0, 0);
                acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b[1], acc[2 * pr], 0, 0, 0);
                acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b[3], acc[2 * pr + 1], 0, 0, 0);
                // keep the software pipeline as written: left alone, hipcc's scheduler moves every fragment read right in
                // front of its MFMAs (lgkmcnt(0) / vmcnt(1..3) ahead of each pair: the LDS and L2 latencies in full, 72 times)
#ifdef W44_PROBE
                if (pk && pr == 0) W44_STAMP(18);
                if (pk && pr == 5) W44_STAMP(19);
                if (pk && pr == 11) W44_STAMP(20);
                if (pk && pr == 17) W44_STAMP(21);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BDEPTH) : "memory");     // this K-step's DMA has landed (older than the window)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef W44_PROBE
            if (pk) W44_STAMP(22);
#endif
            __builtin_amdgcn_s_barrier();
#ifdef W44_PROBE
            if (pk) W44_STAMP(23);
            if (ks == ks0) W44_STAMP(4);
            if (ks == ks0 + 1) W44_STAMP(5);
            if (ks == ks1 - 2) W44_STAMP(6);
#endif
        }
        W44_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the window's last, unused fragments: their registers are reused)

        // ---- tail: A^T M A in registers -> LDS ([tile][pixel][64 channels] per 16-tile half: the K-loop's buffers are free
        //      now) -> either the partial-sum slot (producer) or (+ the other pieces' sums) scale / shift / LeakyReLU /
        //      residual on 16-byte pieces, 256 contiguous bytes per output pixel ------------------------------------------
        if (tid < BT) {
            // output pixel (q, c) of the tile = pixel index rowpart[q] + colpart[c] of y (separable in the mosaic too: image
            // ry * mc + cx, row y, column x -> ((ry * mc) * H + y) * W  +  cx * H * W + x); -1 = no such pixel (past the
            // edge, a gap row / column of the mosaic, or a tile past T)
            const int t = t0 + tid;
            const int n = p.mr ? 0 : t / (p.TH * p.TW);
            const int r = t - n * p.TH * p.TW;
            const int ty = r / p.TW, tx = r - ty * p.TW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int yy = 4 * ty + q, xx = 4 * tx + q, rpart, cpart;
                if (p.mr) {
                    const int ry = yy / (p.H + 1), cx = xx / (p.W + 1);
                    yy -= ry * (p.H + 1);
                    xx -= cx * (p.W + 1);
                    rpart = (ry < p.mr && yy < p.H) ? (ry * p.mc * p.H + yy) * p.W : -1;
                    cpart = (cx < p.mc && xx < p.W) ? cx * p.H * p.W + xx : -1;
                } else {
                    rpart = yy < p.H ? (n * p.H + yy) * p.W : -1;
                    cpart = xx < p.W ? xx : -1;
                }
                tinfo[8 * tid + q] = t < p.T ? rpart : -1;
                tinfo[8 * tid + 4 + q] = t < p.T ? cpart : -1;
            }
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
        W44_STAMP(8);
        __syncthreads();
        W44_STAMP(9);
        const int gt = tid & 255;                          // thread inside the 16-tile half (four waves)
        const int c4 = (gt & 15) * 4;
        if (producer) {
            // later K-steps of a cut block: publish the output-space sums, write-through, then the flag
            const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
                p.partial, 0, (unsigned)((size_t)W * SLOT_FLOATS * 4), 0x00020000);
            const unsigned slot_off = (unsigned)blockIdx.x * (unsigned)(SLOT_FLOATS * 4);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int rowi = it * 16 + (gt >> 4);
                const f32x4 v = *reinterpret_cast<const f32x4*>(cs + rowi * RS + c4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_part,
                                                       slot_off + (unsigned)(((wm * 256 + rowi) * BNC + c4) * 4), 0, 16);   // aux 16 = sc1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !p.fault)
                __hip_atomic_store((gu32*)(p.flags + blockIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (n_extra > 0) {
                if (tid == 0) {
                    for (int e = 1; e <= n_extra; ++e) {
                        gu32* flag = (gu32*)(p.flags + blockIdx.x - e);
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
            const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(
                p.partial, 0, n_extra ? (unsigned)((size_t)W * SLOT_FLOATS * 4) : 0u, 0x00020000);
            const int co = n0 + c4;
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (co < p.Cout) {
                sc = *reinterpret_cast<const f32x4*>(p.scale + co);
                sh = *reinterpret_cast<const f32x4*>(p.shift + co);
            }
            f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
            // all sixteen residual loads of the thread in flight at once (inside the store loop each one is a dependent
            // load behind a branch: sixteen memory latencies in a row)
            f32x4 rv[16];
            if (p.resid) {
                const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.resid), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int rowi = it * 16 + (gt >> 4);
                    const int tl = rowi >> 4, px = rowi & 15;
                    const int rpart = tinfo[8 * (wm * 16 + tl) + (px >> 2)], cpart = tinfo[8 * (wm * 16 + tl) + 4 + (px & 3)];
                    const bool ok = (rpart | cpart) >= 0 && co < p.Cout;
                    rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                 rs_res, ok ? (unsigned)(((rpart + cpart) * p.Cout + co) * 4) : OOB, 0, 0));
                }
            } else {
#pragma unroll
                for (int it = 0; it < 16; ++it) rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int rowi = it * 16 + (gt >> 4);            // (tile, pixel) row of this half: 256 rows
                const int tl = rowi >> 4, px = rowi & 15;
                const int rpart = tinfo[8 * (wm * 16 + tl) + (px >> 2)], cpart = tinfo[8 * (wm * 16 + tl) + 4 + (px & 3)];
                f32x4 v = *reinterpret_cast<const f32x4*>(cs + rowi * RS + c4);
                for (int e = 1; e <= n_extra; ++e)               // the other pieces, in worker order (deterministic)
                    v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                             rs_part, (unsigned)(((wm * 256 + rowi) * BNC + c4) * 4),
                             (unsigned)(blockIdx.x - e) * (unsigned)(SLOT_FLOATS * 4), 17));   // aux 17 = sc0 sc1 (see y3_conv_wino.hip)
                if ((rpart | cpart) >= 0 && co < p.Cout) {
                    v = v * sc + sh;
                    if (p.act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.1f * v[q];
                    }
                    const size_t o = (size_t)(rpart + cpart) * p.Cout + co;
                    v += rv[it];
                    *reinterpret_cast<f32x4*>(p.y + o) = v;
                    if (STATS) { st1 += v; st2 += v * v; }
                }
            }
            if (STATS) {
                // column sums of this block's outputs: the 16 threads that share a channel quad (one per pixel slot, it = tile)
                // add up through the LDS in a fixed order -> deterministic
                __syncthreads();
                float* red = reinterpret_cast<float*>(smem);      // [wm][16 pixel slots][2][64]
                *reinterpret_cast<f32x4*>(red + ((wm * 16 + (gt >> 4)) * 2 + 0) * BNC + c4) = st1;
                *reinterpret_cast<f32x4*>(red + ((wm * 16 + (gt >> 4)) * 2 + 1) * BNC + c4) = st2;
                __syncthreads();
                if (tid < 16 && co < p.Cout) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < (BT / 16) * 16; ++k) {
                        a += *reinterpret_cast<const f32x4*>(red + (k * 2 + 0) * BNC + c4);
                        b += *reinterpret_cast<const f32x4*>(red + (k * 2 + 1) * BNC + c4);
                    }
                    float* st = p.stats + (size_t)bt * 2 * p.Cout;
                    *reinterpret_cast<f32x4*>(st + co) = a;
                    *reinterpret_cast<f32x4*>(st + p.Cout + co) = b;
                }
            }
        }
        W44_STAMP(10);
        __syncthreads();          // the next segment's DMA overwrites the staging tile
        W44_STAMP(11);
    }
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

#ifdef W44_PROBE
extern "C" __attribute__((visibility("default"))) int y3_debug_w44_probe(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_w44_probe), &buf, sizeof(buf));
}
#endif

int y3_conv_wino44_eligible_impl(const y3_conv_desc* d) {
    return d && d->k == 3 && d->stride == 1 && d->c_up == 0 && d->cin % 32 == 0 && d->cout % 64 == 0 &&
           d->n > 0 && d->h > 1 && d->w > 1;
}

// The convs y3_net_forward (dtype 4) runs on this kernel instead of the F(2x2,3x3) one, given the alternative packing
// (y3_net_set_layer_alt): where it measured faster inside the bs=32 416x416 forward (tools/layer_profile.py,
// profiles/r04_wino44.txt; ms per layer, F(2x2) | F(4x4) with 16-tile blocks, two workgroups per CU):
//   64->128 @104: 0.280 | 0.260     128->256 @52: 0.244 | 0.186     256->512 @26: 0.222 | 0.205     512->1024 @13: 0.240 | 0.198
//   32->64 @208: 0.361 | 0.372 - four K-steps per block, the block prologue and store tail dominate: stays on F(2x2) at bs=32.
// Smaller batches (ms per layer at bs = 4 / 8 / 16, F(2x2) stream-K | F(4x4); blocks = 16-tile x 64-channel workgroups):
//   32->64 0.070 | 0.055 (676 blocks), 0.108 | 0.084 (1,352), 0.188 | 0.160 (2,704);  64->128 0.058 | 0.043 (338), 0.089 | 0.071, 0.161 | 0.117;
//   128->256 0.047 | 0.047 (172), 0.082 | 0.064 (340), 0.134 | 0.104;  256->512 0.076 | 0.078 (104), 0.079 | 0.080 (200), 0.126 | 0.104 (392);
//   512->1024 - | 0.137 (64), 0.137 | 0.137 (128), 0.143 | 0.139 (256)       (profiles/r04_wino44.txt)
//   candidate (shape only: what a caller packs for): every eligible conv;
//   preferred (this launch): a candidate whose blocks - 16 tiles x 64 channels, one workgroup each, no K-split - fill at
//     least half of the 512 workgroup slots (below that the F(2x2) kernel's stream-K schedule keeps every CU busy and ties
//     or wins); the four-K-step Cin = 32 shape only up to 4,096 blocks.
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
    return blocks >= 128 * (32 / BT) && (d->cin >= 64 || blocks <= 4096);
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

constexpr int W44_WORKERS = 256 * (32 / BT);     // one persistent workgroup per CU slot (147 KB of LDS per 32 tiles)
constexpr size_t W44_FLAGS_OFFSET = (size_t)W44_WORKERS * SLOT_FLOATS * sizeof(float);

size_t y3_conv_wino44_workspace_bytes_impl(const y3_conv_desc* d) {
    if (!y3_conv_wino44_eligible_impl(d)) return 0;
    return W44_FLAGS_OFFSET + (size_t)W44_WORKERS * sizeof(unsigned);
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
    a.partial = nullptr; a.flags = nullptr; a.err = nullptr; a.spin_limit = 0; a.workers = 0; a.fault = 0;
    a.stats = sk ? sk->stats : nullptr;
    auto kern = a.stats ? conv_wino44_f32_kernel<true> : conv_wino44_f32_kernel<false>;
    static bool attr_set[2] = {false, false};      // benign race (idempotent)
    if (!attr_set[a.stats ? 1 : 0]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE));
        attr_set[a.stats ? 1 : 0] = true;
    }
    const int nbt = (a.T + BT - 1) / BT, nbn = d->cout / BNC;
    const int blocks = nbt * nbn, ksteps = d->cin / KC;
    // Persistent schedule: only when the caller hands in a workspace, the last round of blocks would run partly empty and
    // every worker gets at least two K-steps of the cut blocks.  (y3_net_forward hands in NONE: inside the bs=32 416x416
    // forward one workgroup per block measured 11.72-11.74 ms per batch against 11.75-11.79 - the second, partly empty
    // round of blocks runs faster than a full one, and a cut block pays a second prologue and a 128 KB hand-off; stand-alone
    // the 26-grid 256->512 conv gains 5 %, the 52-grid one nothing.)  Y3_CONV_WINO44_STREAMK=0 turns it off everywhere.
    static const int force = y3_exp_env("Y3_CONV_WINO44_STREAMK") ? atoi(y3_exp_env("Y3_CONV_WINO44_STREAMK")) : -1;
    const bool has_ws = workspace != nullptr && workspace_bytes >= y3_conv_wino44_workspace_bytes_impl(d) &&
                        ((uintptr_t)workspace & 15) == 0;
    const int rem = blocks % W44_WORKERS;
#ifdef W44_SK_KEEP
    const bool use_sk = force != 0 && has_ws && (long long)blocks * ksteps >= 4LL * W44_WORKERS && blocks % W44_WORKERS != 0;
#else
    const bool use_sk = force != 0 && has_ws && blocks > W44_WORKERS && rem != 0 && (long long)rem * ksteps >= 2LL * W44_WORKERS &&
                        !a.stats;       // (the statistics epilogue runs on whole blocks only)
#endif
    if (use_sk) {
        a.partial = static_cast<float*>(workspace);
        a.workers = W44_WORKERS;
        a.err = sk ? sk->err : nullptr;
        y3_sk_debug_env(&a.spin_limit, &a.fault);
        if (sk && sk->flags) {
            a.flags = sk->flags;       // pre-zeroed by the caller (y3_net_forward: one memset per forward)
        } else {
            a.flags = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + W44_FLAGS_OFFSET);
            Y3_CHECK_HIP(hipMemsetAsync(a.flags, 0, (size_t)W44_WORKERS * sizeof(unsigned), stream));
        }
        hipLaunchKernelGGL(kern, dim3(W44_WORKERS), dim3(NTH), 4 * STAGE, stream, a);
    } else {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTH), 4 * STAGE, stream, a);
    }
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
