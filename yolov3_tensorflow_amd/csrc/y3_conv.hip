// conv2d + folded-BN + LeakyReLU (+ residual) (+ fused nearest-upsample/concat input) for gfx950.
//
// Replaces the TensorFlow execution of utils/layer_utils.py:9-22 (conv2d), :25-32 (res_block add),
// :82-87 (upsample_layer) and model.py:62,72 (concat) of the reference.
//
// Design (MI355X-first, not a port of anything):
//   * implicit GEMM, M = N*Ho*Wo output pixels, N = Cout, K = k*k*Cin, never materialising im2col,
//     the padded tensor, the upsampled tensor or the concat;
//   * fp32-in/fp32-accumulate MFMA v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s chip peak): the fp32
//     forward is matrix-pipe bound, so the tile is sized to keep the matrix pipe busy:
//     128 x {128,64,32} block tile, BK = 32, 4 waves, each wave a grid of 32x32 MFMA tiles;
//   * NHWC activations give 16-byte-per-lane loads along Cin; weights are pre-packed
//     [tap][Cout][Cin] so the B tile is loaded exactly like the A tile; both through bounds-checked
//     buffer loads (padding, ragged M and ragged Cout read as 0 with no branch);
//   * LDS tiles are [rows][BK+4] floats: one ds_read_b128 per lane feeds FOUR k-steps of the MFMA
//     (lane l holds k = 4*(l>>5)+j for step j on both operands) and the +4 pad makes the 16-lane
//     groups of ds_read_b128 bank-conflict free;
//   * double-buffered LDS with register prefetch of K-step t+1 during the MFMAs of K-step t: one
//     barrier per K-step; the prefetch runs across tile boundaries in the persistent form;
//   * epilogue: raw accumulators staged through the (then free) LDS so that scale/shift (folded BN or
//     bias), LeakyReLU(0.1), the residual read and the output write are 16 bytes per lane and
//     row-contiguous;
//   * two schedules over the same code: data-parallel (one workgroup per output tile) for layers with
//     many tiles, and stream-K (a persistent grid of 2 workgroups per CU, each owning an equal contiguous
//     range of (tile, K-step) work items) for the 52x52/26x26/13x13 layers, whose tile counts do not
//     divide the 512 co-resident workgroups (tail quantisation cost 12-33 % there).  Split tiles are
//     combined in a fixed order by a second small kernel, so results are deterministic.
#include <cstdlib>
#include "y3_conv_common.h"

namespace {
using namespace y3conv;

// TMODE (compile time, so that the forward instantiations carry none of its state): data gradient of a stride-2
// conv, one output parity class per launch (see ConvArgs::tmode)
template <int BM, int BN, int WGM, int WGN, int KS, bool UPCAT, bool STREAMK, bool TMODE, bool STATS = false, bool BSTATS = false>
__global__ void __launch_bounds__(256, 2) conv_mfma_f32_kernel(const ConvArgs p) {
    using G = Geo<BM, BN, WGM, WGN>;
    constexpr int MI = G::MI, NI = G::NI, WTM = G::WTM, WTN = G::WTN;
    constexpr int AROWS = BM / 32, BROWS = BN / 32;  // float4 rows each thread stages

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;  // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // Output tiles are enumerated COLUMN-major (tile = bn * nbm + bm) and every XCD works on a contiguous
    // range of tile ids: the workgroups sharing one L2 then read ONE weight panel [taps][BN][Cin] (<= 2.4 MB,
    // L2-resident) instead of cycling through the whole kernel tensor (4.7-18.9 MB for the 26x26/13x13 layers,
    // which row-major order re-fetched from the Infinity Cache for every M tile: ~800 MB per launch measured).
    const int nbn = (p.Cout + BN - 1) / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int kchunks = p.Cin / BK;
    const int taps = TMODE ? p.ntaps : KS * KS;
    const int S = taps * kchunks;  // K-steps per output tile

    // ---- this workgroup's range of work items (item = tile * S + kstep) ----------------------------
    long long item, item_end;
    SkWorker skw = {};
    const int ntiles = nbm * nbn;
    if (STREAMK) {
        skw = sk_worker(blockIdx.x, ntiles, S, p.workers);
        item = skw.begin;
        item_end = skw.end;
    } else {
        // workgroup b runs on XCD b%8 (observed): give each XCD a contiguous eighth of the tile ids, and each of the XCD's
        // workgroups a contiguous, balanced run of WHOLE tiles of it - one tile when the grid has a workgroup per tile; with
        // fewer workgroups than tiles (the resident-grid launch of the 1x1 convs, launch_data_parallel) a workgroup walks
        // its tiles, the next tile's first loads issued under the current tile's epilogue
        const int nt = gridDim.x;
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int wx = (nt >> 3) + ((nt & 7) > x ? 1 : 0);                     // workgroups of this launch on XCD x
        const int t0x = (int)sk_begin32(ntiles, 8, x), t1x = (int)sk_begin32(ntiles, 8, x + 1);
        const int tb = t0x + (int)sk_begin32(t1x - t0x, wx, k), te = t0x + (int)sk_begin32(t1x - t0x, wx, k + 1);
        item = (long long)tb * S;
        item_end = (long long)te * S;
    }
    if (item >= item_end) return;

    // ---- per-thread staging coordinates: float4 column c4 of rows r0 + 32*j ---------------------
    const int c4 = (tid & 7) * 4;
    const int r0 = tid >> 3;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(UPCAT ? p.xu : p.x), 0,
        (unsigned)(UPCAT ? (size_t)p.N * (p.H >> 1) * (p.W >> 1) * p.Cu * 4 : 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (unsigned)((size_t)KS * KS * p.Cout * p.Cin * 4), 0x00020000);

    // loader state: the tile and K-step the NEXT issue_loads() call fetches.
    // K-step order is tap major, channel-chunk minor: inside a tap the chunk advances through the scalar
    // soffset operand of the buffer load (8 loads + a few SALU per K-step); a tap change recomputes the
    // per-lane byte offsets from one base offset and a validity mask per row (bit ky: row tap ky in range,
    // bit 4+kx: column tap kx in range; transposed mode adds the parities).  (Chunk-major order was measured:
    // fabric traffic -30 % but 11 % slower — the per-step tap change sits outside the MFMA shadow.)
    int a_base[AROWS];
    int a_msk[AROWS];
    int a_base_u[UPCAT ? AROWS : 1];
    unsigned a_voff[AROWS], a_voff_u[UPCAT ? AROWS : 1], b_voff[BROWS];
    int ld_tap = 0, ld_cc = 0;

    int ld_wtap = 0;   // index of the weight tap plane of the prepared K-step
    auto set_tap = [&]() {
        int ky, kx, dy, dx;
        if (TMODE) {
            const int nkx = p.cx ? 2 : 1;
            const int ty = ld_tap / nkx, tx = ld_tap - ty * nkx;
            ky = p.cy ? 2 * ty : 1;
            kx = p.cx ? 2 * tx : 1;
            dy = (p.cy + ky - 1) >> 1;     // source row = y' + dy
            dx = (p.cx + kx - 1) >> 1;
        } else {
            ky = (KS == 1) ? 0 : ld_tap / KS;
            kx = (KS == 1) ? 0 : ld_tap - ky * KS;
            dy = ky;
            dx = kx;
        }
        ld_wtap = p.wrev ? KS * KS - 1 - (ky * KS + kx) : ky * KS + kx;
        const int tap_off = (dy * p.W + dx) * p.Cx + c4;
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            const int mk = a_msk[j];
            const bool ok = ((mk >> dy) & (mk >> (4 + dx)) & 1) != 0;
            a_voff[j] = ok ? (unsigned)(a_base[j] + tap_off) * 4u : OOB;
            if (UPCAT) a_voff_u[j] = ok ? (unsigned)(a_base_u[j] + c4) * 4u : OOB;
        }
    };

    // (tile, ks) as 32-bit values: a 64-bit `item / S` here and at the top of the segment loop was ~150 scalar instructions of
    // software division per tile, and hipcc put an s_waitcnt vmcnt(0) in front of it (its VALU temporaries) - which waited for
    // the loads prefetched across the tile boundary: per 64x64 tile of a 1x1 conv (8-32 K-steps) that was the difference
    // between 0.59 and the matrix-pipe bound (round 6, profiles/r06_conv1x1_ko.txt)
    auto set_loader = [&](int tile, int ks) {
        const int bn = tile / nbm, bm = tile - bn * nbm;
        ld_tap = ks / kchunks;
        ld_cc = ks - ld_tap * kchunks;
        const int HoWo = TMODE ? p.H * p.W : p.Ho * p.Wo;
        const int Wrow = TMODE ? p.W : p.Wo;
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            const int m = bm * BM + r0 + 32 * j;
            int mk = 0, base = 0, base_u = 0;
            if (KS == 1 && !TMODE && !UPCAT && p.stride == 1) {
                // 1x1, stride 1: GEMM row m IS pixel m (no padding, no decomposition into image / row / column)
                if (m < p.M) { mk = 0x11; base = m * p.Cx; }
            } else if (m < p.M) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int oy = rem / Wrow;
                const int ox = rem - oy * Wrow;
                // first source pixel of the row (tmode: (y', x'); else (oy*stride - pad, ox*stride - pad))
                const int iy0 = TMODE ? oy : oy * p.stride - p.pad;
                const int ix0 = TMODE ? ox : ox * p.stride - p.pad;
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    if ((unsigned)(iy0 + t) < (unsigned)p.H) mk |= 1 << t;
                    if ((unsigned)(ix0 + t) < (unsigned)p.W) mk |= 1 << (4 + t);
                }
                base = ((n * p.H + iy0) * p.W + ix0) * p.Cx;
                if (UPCAT) base_u = ((n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cu;
            }
            a_msk[j] = mk;
            a_base[j] = base;
            if (UPCAT) a_base_u[j] = base_u;
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int co = bn * BN + r0 + 32 * j;
            b_voff[j] = co < p.Cout ? (unsigned)(co * p.Cin + c4) * 4u : OOB;
        }
        set_tap();
    };

    f32x4 ra[AROWS], rb[BROWS];

    // issue the 8 buffer loads of the prepared K-step (branch-free), then advance the loader by one K-step
    auto issue_loads = [&]() {
        const int c0 = ld_cc * BK;
        if (UPCAT) {
            // channels [0, Cu) come from the half-resolution tensor, the rest from the route tensor
            const bool from_up = c0 < p.Cu;
            const unsigned soff = (unsigned)(from_up ? c0 : c0 - p.Cu) * 4u;
#pragma unroll
            for (int j = 0; j < AROWS; ++j)
                ra[j] = __builtin_bit_cast(
                    f32x4, from_up ? __builtin_amdgcn_raw_buffer_load_b128(rs_u, a_voff_u[j], soff, 0)
                                   : __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j], soff, 0));
        } else {
            const unsigned soff = (unsigned)c0 * 4u;
#pragma unroll
            for (int j = 0; j < AROWS; ++j)
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_voff[j], soff, 0));
        }
        const unsigned wsoff = (unsigned)((ld_wtap * p.Cout) * p.Cin + c0) * 4u;
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff[j], wsoff, 0));
    };
    auto advance = [&]() {
        if (++ld_cc == kchunks) {
            ld_cc = 0;
            ++ld_tap;
            if (KS > 1 && ld_tap < taps) set_tap();
        }
    };

    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * LDK;
        float* bs = Bs + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < AROWS; ++j)
            *reinterpret_cast<f32x4*>(as + (r0 + 32 * j) * LDK + c4) = ra[j];
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            *reinterpret_cast<f32x4*>(bs + (r0 + 32 * j) * LDK + c4) = rb[j];
    };

    f32x16 acc[MI][NI];
    const int frag_row = lane & 31;
    const int frag_k = 4 * (lane >> 5);

    auto compute_tile = [&](int buf) {
        const float* as = As + buf * BM * LDK + (wm * WTM + frag_row) * LDK + frag_k;
        const float* bs = Bs + buf * BN * LDK + (wn * WTN + frag_row) * LDK + frag_k;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = *reinterpret_cast<const f32x4*>(as + mi * 32 * LDK + kk * 8);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = *reinterpret_cast<const f32x4*>(bs + ni * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j],
                                                                           acc[mi][ni], 0, 0, 0);
        }
    };

    // Interleave the memory instructions of one K-step into the shadow of its 64 MFMAs (a wave issues in
    // order and an MFMA occupies the matrix pipe for 64 cycles, so whatever sits BETWEEN two MFMAs is free,
    // whatever sits before the first or after the last one is exposed): loads of K-step t+1 behind the
    // first MFMAs, fragment reads of the second half behind the second quarter, LDS writes behind the last
    // quarter.
    auto pipeline_hint = [&]() {
        constexpr int MF = MI * NI * 4;   // MFMAs per k-group of 8
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);          // fragments of k-groups 0,1
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < AROWS + BROWS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 2 * (MI + NI)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // fragments 2,3
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < AROWS + BROWS) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        }
    };

    // ---- segments: maximal runs of K-steps of one tile inside this workgroup's item range --------
    // Invariant at the top of a segment: ra/rb hold (or are receiving) its first K-step and the loader is
    // prepared for its second one.
    int tile = (int)(item / S);                    // (once per workgroup; tiles advance by one below)
    int ks = (int)(item - (long long)tile * S);
    set_loader(tile, ks);
    issue_loads();
    advance();
    // (the first K-step's operands are needed at once; waiting HERE, visibly to hipcc, keeps the wait out of the loop header,
    // where the back edge would turn it into a wait for the previous tile's output stores: loads and stores share vmcnt)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    while (item < item_end) {
        const long long tile_end = (long long)(tile + 1) * S;
        const long long seg_end = tile_end < item_end ? tile_end : item_end;
        const int nsteps = (int)(seg_end - item);
        const int bn = tile / nbm, bm = tile - bn * nbm;
        const int m0 = bm * BM, n0 = bn * BN;

#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        store_tile(0);
        __syncthreads();
        for (int t = 0; t + 1 < nsteps; ++t) {
            issue_loads();               // K-step t+1
            compute_tile(t & 1);
            store_tile((t + 1) & 1);
            pipeline_hint();
            advance();                   // prepare K-step t+2 (scalar branch on a tap change)
            __syncthreads();
        }
        if (seg_end < item_end) {
            set_loader(tile + 1, 0);     // (a segment that is not the last ends at its tile's end) prefetch across the tile boundary; stored after the epilogue
            issue_loads();
            advance();
        }
        compute_tile((nsteps - 1) & 1);
        __syncthreads();
        // the next tile's first operands (requested above, a K-step ago) are in: said HERE so that no path through the epilogue
        // leaves them "pending" for hipcc - at the loop header that would become a wait for this tile's output stores
        __builtin_amdgcn_s_waitcnt(0x0F70);

        if (STREAMK && ks > 0) {
            // later K-steps of a cut tile (this worker's first segment): publish the raw accumulators
            sk_publish<BM, BN, WGM, WGN>(p, skw.id, acc);
        } else {
            // whole tile, or K-steps [0, k) of a cut tile (this worker's last segment): add what the next workers
            // of the group published for it, then the common epilogue
            if (STREAMK && seg_end < tile_end) sk_consume<BM, BN, WGM, WGN>(p, skw, ntiles, S, tile_end, acc);
            epilogue<BM, BN, WGM, WGN, TMODE, STATS, BSTATS>(p, smem, acc, m0, n0);
        }
        if (STREAMK || seg_end < item_end) __syncthreads();  // the staging LDS is reused by the next segment
        item = seg_end;
        ++tile;
        ks = 0;
    }
}

constexpr int RESIDENT_64 = 1024;      // 64x64-tile workgroups resident at once: four per CU (37 KB of LDS, 72 VGPRs)

template <int BM, int BN, int WGM, int WGN, int KS, bool UPCAT, bool TMODE = false, bool STATS = false, bool BSTATS = false>
int launch_data_parallel(hipStream_t stream, const ConvArgs& a) {
    using G = Geo<BM, BN, WGM, WGN>;
    auto kern = conv_mfma_f32_kernel<BM, BN, WGM, WGN, KS, UPCAT, false, TMODE, STATS, BSTATS>;
    static bool attr_set[Y3_MAX_DEVICES] = {};  // per instantiation; benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        if (int rc = set_lds_attr(kern, G::LDS_BYTES)) return rc;
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int nbm = (a.M + BM - 1) / BM;
    const int nbn = (a.Cout + BN - 1) / BN;
    int grid = nbm * nbn;
    // 64x64 tiles (the 1x1 convs), more tiles than the 1,024 resident workgroups but only a few rounds of them: a RESIDENT
    // grid whose workgroups walk balanced runs of whole tiles.  The hardware dispatcher hands the tiles of a partly filled
    // last round to the first slots that come free - four per CU on a quarter of the CUs while the rest idle (measured on
    // the F(4x4) kernel: 1.5 rounds cost 2, tools/wino44_quant.py); a resident grid spreads them one or two per CU, and
    // a workgroup fetches its next tile under the current one's store tail.
    if (BM == 64 && BN == 64 && grid > RESIDENT_64 && grid < 8 * RESIDENT_64 && y3_exp_env("Y3_CONV_RESIDENT_OFF") == nullptr)
        grid = RESIDENT_64;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

template <int KS, bool TMODE = false, bool STATS = false>
int launch_streamk(hipStream_t stream, const ConvArgs& a) {
    constexpr int BM = 128, BN = 128, WGM = 2, WGN = 2;
    using G = Geo<BM, BN, WGM, WGN>;
    auto kern = conv_mfma_f32_kernel<BM, BN, WGM, WGN, KS, false, true, TMODE, STATS>;
    static bool attr_set[Y3_MAX_DEVICES] = {};
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        if (int rc = set_lds_attr(kern, G::LDS_BYTES)) return rc;
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.workers), dim3(256), G::LDS_BYTES, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

template <int KS, bool UPCAT, bool TMODE = false, bool STATS = false, bool BSTATS = false>
int dispatch_bn(hipStream_t stream, const ConvArgs& a) {
    if (a.Cout <= 32) return launch_data_parallel<128, 32, 4, 1, KS, UPCAT, TMODE, STATS, BSTATS>(stream, a);
    if (a.Cout <= 64) return launch_data_parallel<128, 64, 4, 1, KS, UPCAT, TMODE, STATS, BSTATS>(stream, a);
    // 1x1 layers have 8-32 K-steps per tile: 64x64 tiles (37 KB of LDS, 72 VGPRs -> 4 workgroups per CU) hide one
    // tile's prologue/epilogue under its neighbours' MFMAs and quantise the 172-1352-tile grids of the network 4x
    // finer (measured, batch 32: 52x52 -10 %, 26x26 -19 %, 13x13 -21 % against the 128x128 tile)
    if (KS == 1 && !TMODE) return launch_data_parallel<64, 64, 2, 2, KS, UPCAT, TMODE, STATS, BSTATS>(stream, a);
    static_assert(!BSTATS || KS == 1, "the fused BN backward reduction exists on the small tiles of the 1x1 data gradient only");
    return launch_data_parallel<128, 128, 2, 2, KS, UPCAT, TMODE, STATS, BSTATS>(stream, a);
}

}  // namespace

// 1 if y3_launch_conv would pick the stream-K schedule for this conv when given a workspace.
int y3_conv_schedule_impl(const y3_conv_desc* d) {
    if (!d || d->k != 3 || d->cin == 3 || d->c_up > 0) return 0;
    ConvArgs a;
    a.xu = nullptr;
    a.Cout = d->cout;
    a.M = d->n * (d->h / d->stride) * (d->w / d->stride);
    return use_streamk(a, d->k, true) ? 1 : 0;
}

int y3_streamk_range_impl(int kind, int units, int ksteps, int workers, int group, int local_worker, long long* begin,
                          long long* end) {
    Y3_CHECK_ARG(begin && end && units > 0 && ksteps > 0 && workers >= 8 && workers % 8 == 0 && group >= 0 && group < 8 &&
                     local_worker >= 0 && local_worker < workers / 8 && (long long)units * ksteps < (1LL << 31),
                 "y3_streamk_range: bad argument");
    if (kind == 0) sk_range(units, ksteps, workers, group, local_worker, *begin, *end);
    else y3_wino_range_impl(units, ksteps, workers, group, local_worker, kind == 2, begin, end);
    return Y3_OK;
}

size_t y3_conv_workspace_bytes_impl(const y3_conv_desc* d) {
    if (!d || d->k != 3 || d->cout < 128 || d->c_up > 0) return 0;
    return SK_WORKSPACE_BYTES;
}

int y3_launch_conv(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* x_up,
                   const float* w, const float* scale, const float* shift, const float* residual,
                   float* y, void* workspace, size_t workspace_bytes, const y3_sk_opts* sk) {
    Y3_CHECK_ARG(d && x && w && scale && shift && y, "y3_conv2d_fwd: null pointer argument");
    Y3_CHECK_ARG(d->k == 1 || d->k == 3, "y3_conv2d_fwd: kernel_size must be 1 or 3 (got %d)", d->k);
    Y3_CHECK_ARG(d->stride == 1 || d->stride == 2, "y3_conv2d_fwd: stride must be 1 or 2 (got %d)",
                 d->stride);
    Y3_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0,
                 "y3_conv2d_fwd: non-positive dimension");
    Y3_CHECK_ARG(!(d->stride == 2 && (d->h % 2 || d->w % 2)),
                 "y3_conv2d_fwd: stride-2 conv needs even H,W (got %dx%d)", d->h, d->w);
    Y3_CHECK_ARG((x_up != nullptr) == (d->c_up > 0), "y3_conv2d_fwd: x_up and c_up must agree");
    ConvArgs a;
    a.x = x; a.xu = x_up; a.w = w; a.scale = scale; a.shift = shift; a.resid = residual; a.y = y;
    a.partial = nullptr; a.flags = nullptr; a.workers = 0; a.wrev = 0; a.tmode = 0; a.cy = a.cx = 0; a.ntaps = 0;
    a.err = nullptr; a.spin_limit = 0; a.fault = 0; a.stats = nullptr; a.bz = nullptr; a.bvec = nullptr;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cu = d->c_up; a.Cx = d->cin - d->c_up;
    a.Cout = d->cout; a.stride = d->stride; a.pad = d->k / 2; a.act = d->act;
    a.Ho = d->h / d->stride; a.Wo = d->w / d->stride;
    const long long M = (long long)d->n * a.Ho * a.Wo;
    Y3_CHECK_ARG((long long)d->n * d->h * d->w * d->cin < (1LL << 29) && M * d->cout < (1LL << 29),
                 "y3_conv2d_fwd: tensor exceeds 2^29 elements (32-bit byte offsets)");
    a.M = (int)M;
    a.stats = sk ? sk->stats : nullptr;
    Y3_CHECK_ARG(!a.stats || (d->cin != 3 && !x_up && d->cout % 4 == 0),
                 "y3_conv2d_fwd_stats: statistics need Cin != 3, no fused upsample input and Cout %% 4 == 0");

    if (d->cin == 3) {
        Y3_CHECK_ARG(d->k == 3 && d->cout == 32 && !x_up && !residual,
                     "y3_conv2d_fwd: Cin=3 is supported only as the 3x3 3->32 stem conv");
        auto stem = conv_stem_kernel<32>;
        hipLaunchKernelGGL(stem, dim3((a.M + 255) / 256), dim3(256), 0, stream, a);
        Y3_CHECK_HIP(hipGetLastError());
        return Y3_OK;
    }
    Y3_CHECK_ARG(d->cin % BK == 0, "y3_conv2d_fwd: Cin must be 3 or a multiple of %d (got %d)", BK,
                 d->cin);
    if (x_up) {
        Y3_CHECK_ARG(d->k == 1 && d->stride == 1, "y3_conv2d_fwd: fused upsample+concat needs a 1x1 s1 conv");
        Y3_CHECK_ARG(d->c_up % BK == 0 && d->c_up < d->cin && d->h % 2 == 0 && d->w % 2 == 0,
                     "y3_conv2d_fwd: bad c_up=%d for cin=%d", d->c_up, d->cin);
        return dispatch_bn<1, true>(stream, a);
    }
    if (d->k == 1) {
        Y3_CHECK_ARG(d->stride == 1, "y3_conv2d_fwd: 1x1 conv must have stride 1");
        return a.stats ? dispatch_bn<1, false, false, true>(stream, a) : dispatch_bn<1, false>(stream, a);
    }
    const bool has_ws = workspace != nullptr && workspace_bytes >= y3_conv_workspace_bytes_impl(d) &&
                        ((uintptr_t)workspace & 15) == 0;
    if (use_streamk(a, d->k, has_ws)) {
        if (int rc = sk_prepare(stream, a, workspace, sk)) return rc;
        return a.stats ? launch_streamk<3, false, true>(stream, a) : launch_streamk<3>(stream, a);
    }
    return a.stats ? dispatch_bn<3, false, false, true>(stream, a) : dispatch_bn<3, false>(stream, a);
}

// Row blocks of the `stats` output (= output rows / the BM the dispatch above picks); the Winograd kernel: 64-tile blocks.
int y3_conv_stats_blocks_impl(const y3_conv_desc* d, int wino) {
    if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout % 4 != 0 || d->cin == 3 || d->c_up > 0) return 0;
    if (wino == 2) return y3_conv_wino44_stats_blocks_impl(d);
    if (wino) {
        if (!y3_conv_wino_eligible_impl(d)) return 0;
        const long long T = (long long)d->n * ((d->h + 1) / 2) * ((d->w + 1) / 2);
        return (int)((T + 63) / 64);
    }
    if (!((d->k == 1 && d->stride == 1) || d->k == 3) || (d->stride != 1 && d->stride != 2)) return 0;
    const long long M = (long long)d->n * (d->h / d->stride) * (d->w / d->stride);
    const int bm = (d->k == 1 && d->cout > 64) ? 64 : 128;
    return (int)((M + bm - 1) / bm);
}

int y3_conv_dgrad_stats_blocks_impl(const y3_conv_desc* fwd) {
    if (!fwd || fwd->k != 1 || fwd->stride != 1 || fwd->c_up != 0 || fwd->cin % 4 != 0) return 0;
    const long long M = (long long)fwd->n * fwd->h * fwd->w;
    const int bm = fwd->cin > 64 ? 64 : 128;        // dispatch_bn's tile for Cout' = fwd->cin
    return (int)((M + bm - 1) / bm);
}

// Data gradient of a conv layer as a forward conv over dz:
//   stride 1: dx = conv_same(dz, flip(W)^T)  -> same kernel, weights = the HWIO variable read with reversed taps
//             ([tap][ci][co] is already "[tap][Cout'=ci][Cin'=co]"), no re-packing;
//   stride 2: transposed gather (tmode) on the 2x finer output grid.
// fwd describes the FORWARD layer (n, h, w = its input size).  dz is [N,Ho,Wo] x dz_stride channels
// (dz_stride >= cout: the detection convs pad 255 -> 256), w_d is [k*k][cin][dz_stride].
// accumulate != 0 adds into dx (gradient fan-in) instead of overwriting it.
int y3_launch_conv_dgrad(hipStream_t stream, const y3_conv_desc* fwd, const float* dz, int dz_stride,
                         const float* w_d, const float* ones, const float* zeros, int accumulate, float* dx,
                         void* workspace, size_t workspace_bytes, const y3_sk_opts* sk) {
    Y3_CHECK_ARG(fwd && dz && w_d && ones && zeros && dx, "y3_conv2d_dgrad: null pointer argument");
    Y3_CHECK_ARG(fwd->k == 1 || fwd->k == 3, "y3_conv2d_dgrad: kernel_size must be 1 or 3");
    Y3_CHECK_ARG(fwd->stride == 1 || (fwd->stride == 2 && fwd->k == 3), "y3_conv2d_dgrad: unsupported stride");
    Y3_CHECK_ARG(fwd->c_up == 0, "y3_conv2d_dgrad: fused upsample+concat inputs are not supported");
    Y3_CHECK_ARG(dz_stride >= fwd->cout && dz_stride % BK == 0, "y3_conv2d_dgrad: dz stride must be a multiple of %d", BK);
    Y3_CHECK_ARG(fwd->cin % 4 == 0, "y3_conv2d_dgrad: Cin must be a multiple of 4");
    const int Ho = fwd->h / fwd->stride, Wo = fwd->w / fwd->stride;
    ConvArgs a;
    a.x = dz; a.xu = nullptr; a.w = w_d; a.scale = ones; a.shift = zeros;
    a.resid = accumulate ? dx : nullptr; a.y = dx; a.partial = nullptr; a.flags = nullptr; a.workers = 0;
    a.err = nullptr; a.spin_limit = 0; a.fault = 0; a.stats = nullptr; a.bz = nullptr; a.bvec = nullptr;
    a.wrev = 1; a.tmode = 0; a.cy = a.cx = 0; a.ntaps = 0;
    a.N = fwd->n; a.H = Ho; a.W = Wo; a.Cin = dz_stride; a.Cu = 0; a.Cx = dz_stride;
    a.Cout = fwd->cin; a.stride = 1; a.pad = fwd->k / 2; a.act = 0;
    a.Ho = fwd->h; a.Wo = fwd->w;
    const long long M = (long long)fwd->n * fwd->h * fwd->w;
    Y3_CHECK_ARG(M * fwd->cin < (1LL << 29) && (long long)fwd->n * Ho * Wo * dz_stride < (1LL << 29),
                 "y3_conv2d_dgrad: tensor exceeds 2^29 elements (32-bit byte offsets)");
    a.M = (int)M;
    if (fwd->k == 1) {
        if (sk && sk->bwd_z) {
            // this gradient is the dy of a BN layer: its backward reduction rides in the epilogue (y3_sk_opts::bwd_*)
            Y3_CHECK_ARG(sk->bwd_vec && sk->stats, "y3_conv2d_dgrad: the fused BN reduction needs z, the layer's vectors and the partial rows");
            a.bz = sk->bwd_z; a.bvec = sk->bwd_vec; a.stats = sk->stats;
            return dispatch_bn<1, false, false, false, true>(stream, a);
        }
        return dispatch_bn<1, false>(stream, a);
    }
    const bool has_ws = workspace != nullptr && workspace_bytes >= SK_WORKSPACE_BYTES &&
                        ((uintptr_t)workspace & 15) == 0;
    if (fwd->stride == 2) {
        // four output parity classes, each a dense conv over N*Ho*Wo rows with 1/2/2/4 taps
        a.tmode = 1;
        a.M = (int)((long long)fwd->n * Ho * Wo);
        for (int cls = 0; cls < 4; ++cls) {
            a.cy = cls >> 1; a.cx = cls & 1;
            a.ntaps = (a.cy ? 2 : 1) * (a.cx ? 2 : 1);
            a.partial = nullptr; a.workers = 0;
            int rc;
            if (use_streamk(a, 3, has_ws)) {
                // (the four parity-class launches share the workspace's own flag words: zeroed ahead of each)
                y3_sk_opts o;
                o.err = sk ? sk->err : nullptr;
                rc = sk_prepare(stream, a, workspace, &o);
                if (rc == Y3_OK) rc = launch_streamk<3, true>(stream, a);
            } else {
                rc = dispatch_bn<3, false, true>(stream, a);
            }
            if (rc != Y3_OK) return rc;
        }
        return Y3_OK;
    }
    if (use_streamk(a, 3, has_ws)) {
        if (int rc = sk_prepare(stream, a, workspace, sk)) return rc;
        return launch_streamk<3>(stream, a);
    }
    return dispatch_bn<3, false>(stream, a);
}
