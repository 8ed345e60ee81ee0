// fp32 conv2d on the bf16 matrix pipe: every fp32 operand is split into NP bf16 planes (x = x1 + x2 (+ x3),
// exact for NP = 3: 3 x 8 significant bits cover the 24-bit fp32 significand) and the product a*b is rebuilt
// from the plane products a_i*b_j with i + j <= NP - 1 (6 products for NP = 3, 3 for NP = 2), each computed
// exactly by v_mfma_f32_32x32x16_bf16 and accumulated in fp32.  Dropped terms are <= 2^-23 |a b| (NP = 3) or
// <= 2^-15 |a b| (NP = 2).  Same drop-in contract as y3_conv.hip (fp32 tensors in HBM, same epilogue, same
// stream-K schedule); the weights are pre-split once at load time, the activations are split on the fly between
// the global load and the LDS write.
//
// Why: the exact fp32 MFMA (v_mfma_f32_32x32x2_f32) peaks at 157 TF/s; six bf16 MFMAs per fp32 product run at
// 2.5 PF / 6 = 417 TF/s fp32-equivalent (SURVEY.md hard part 1 names this route).
//
// Tile: 128 x {128,64,32}, 16 K elements per step, 4 waves; LDS rows are 32 bytes per plane, unpadded, with the two
// 16-byte halves XOR-swizzled (conflict-free ds_read_b128 fragments and staging writes).  A K-step is only 24 MFMAs
// x 32 cycles per wave, shorter than the memory latency, so global loads run TWO K-steps ahead in two alternating
// register sets; a work item of the stream-K partition is a PAIR of K-steps so that the set index is static.
// The same kernel computes the stride-1 data gradients of training (wrev: flipped kernel, transposed weight
// packing).  Design notes, measurements and rejected variants: docs/history_r01_r05.md 4.3.
#include <type_traits>
#include "y3_conv_common.h"

namespace {
using namespace y3conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SBK = 16;    // K elements per K-step
constexpr int SROW = 32;   // LDS bytes per tile row and plane: 16 bf16, the two 16-byte halves XOR-swizzled

// LDS byte offset of half `half` (0/1) of row `row` inside one plane tile: rows are 32 B, unpadded; the half index
// is flipped on rows 8..15 (mod 16), which makes both the ds_read_b128 fragment reads (16-lane groups over 16
// rows of one half) and the 128-byte-contiguous staging writes bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int half) { return row * SROW + ((half ^ ((row >> 3) & 1)) << 4); }

template <int BM, int BN, int NP>
constexpr size_t split_lds_bytes() {
    const size_t t = (size_t)2 * NP * (BM + BN) * SROW;
    const size_t e = (size_t)BM * (BN + 4) * sizeof(float);   // epilogue staging
    return t > e ? t : e;
}

template <int BM, int BN, int WGM, int WGN, int KS, bool UPCAT, bool STREAMK, int NP>
__global__ void __launch_bounds__(256, 2) conv_mfma_split_kernel(const ConvArgs p) {
    using G = Geo<BM, BN, WGM, WGN>;
    constexpr int MI = G::MI, NI = G::NI, WTM = G::WTM, WTN = G::WTN;
    constexpr int AROWS = BM / 64;                      // float4 chunks of A each thread stages per K-step
    constexpr int PLANE_A = BM * SROW, PLANE_B = BN * SROW;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* As = smem_raw;                       // [2][NP][BM][SROW]
    unsigned char* Bs = smem_raw + 2 * NP * PLANE_A;    // [2][NP][BN][SROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const int nbn = (p.Cout + BN - 1) / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int kchunks = p.Cin / SBK;       // 16-wide K-steps per tap (even: Cin % 32 == 0)
    constexpr int taps = KS * KS;
    const int S = taps * (kchunks >> 1);   // work items per output tile; one item = a PAIR of K-steps (32 K elements)

    // ---- this workgroup's range of work items (item = tile * S + pair) --------------------------------
    long long item, item_end;
    SkWorker skw = {};
    const int ntiles = nbm * nbn;
    if (STREAMK) {
        skw = sk_worker(blockIdx.x, ntiles, S, p.workers);
        item = skw.begin;
        item_end = skw.end;
    } else {
        const int nt = gridDim.x;
        const int q = nt >> 3, r = nt & 7, x = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int tile_id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
        item = (long long)tile_id * S;
        item_end = item + S;
    }
    if (item >= item_end) return;

    // staging coordinates.  A: float4 column c4 of rows r0 + 64*j.  B: 16-byte half `tid&1` (8 bf16) of row tid>>1.
    const int brow = tid >> 1, bhalf = tid & 1;
    const int c4 = (tid & 3) * 4;
    const int r0 = tid >> 2;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(UPCAT ? p.xu : p.x), 0,
        (unsigned)(UPCAT ? (size_t)p.N * (p.H >> 1) * (p.W >> 1) * p.Cu * 4 : 16), 0x00020000);
    const unsigned wplane = (unsigned)(p.Cout * SBK * 2);   // bytes of one [Cout][16] bf16 plane slice
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (unsigned)((size_t)taps * p.Cout * p.Cin * 2 * NP), 0x00020000);

    // ---- loader: the K-step the next issue() fetches.  issue() is branch-free (a branch around a load makes the
    // compiler drain vmcnt at its join); tile changes happen outside the hot loop through set_loader().
    int a_base[AROWS], a_msk[AROWS], a_base_u[UPCAT ? AROWS : 1];
    unsigned b_voff;
    int ld_tap = 0, ld_cc = 0;

    // (32-bit tile / pair: a 64-bit `item / S` per tile is ~150 scalar instructions of software division, see y3_conv.hip)
    auto set_loader = [&](int tile, int pair) {
        const int ks = 2 * pair;
        const int bn = tile / nbm, bm = tile - bn * nbm;
        ld_tap = ks / kchunks;
        ld_cc = ks - ld_tap * kchunks;
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            const int m = bm * BM + r0 + 64 * j;
            int mk = 0, base = 0, base_u = 0;
            if (m < p.M) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad;
                const int ix0 = ox * p.stride - p.pad;
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    if ((unsigned)(iy0 + t) < (unsigned)p.H) mk |= 1 << t;
                    if ((unsigned)(ix0 + t) < (unsigned)p.W) mk |= 1 << (4 + t);
                }
                base = ((n * p.H + iy0) * p.W + ix0) * p.Cx + c4;
                if (UPCAT) base_u = ((n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.Cu + c4;
            }
            a_msk[j] = mk;
            a_base[j] = base;
            if (UPCAT) a_base_u[j] = base_u;
        }
        const int co = bn * BN + brow;
        b_voff = (brow < BN && co < p.Cout) ? (unsigned)(co * SBK + bhalf * 8) * 2u : OOB;
    };

    f32x4 ra[2][AROWS];
    u32x4 rb[2][NP];

    // fetch the prepared K-step into register set s and step the loader.  Past the end of a tile the state runs
    // on (tap == taps): those loads read in-bounds-or-zero addresses and are never stored.
    auto issue = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int ky = (KS == 1) ? 0 : ld_tap / KS;
        const int kx = (KS == 1) ? 0 : ld_tap - ky * KS;
        const int tap_off = (ky * p.W + kx) * p.Cx;
        const int c0 = ld_cc * SBK;
        unsigned voff[AROWS], voff_u[UPCAT ? AROWS : 1];
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            const int mk = a_msk[j];
            const bool ok = ((mk >> ky) & (mk >> (4 + kx)) & 1) != 0;
            voff[j] = ok ? (unsigned)(a_base[j] + tap_off) * 4u : OOB;
            if (UPCAT) voff_u[j] = ok ? (unsigned)a_base_u[j] * 4u : OOB;
        }
        if (UPCAT) {
            const bool from_up = c0 < p.Cu;
            const unsigned soff = (unsigned)(from_up ? c0 : c0 - p.Cu) * 4u;
#pragma unroll
            for (int j = 0; j < AROWS; ++j)
                ra[s][j] = __builtin_bit_cast(
                    f32x4, from_up ? __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u[j], soff, 0)
                                   : __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff[j], soff, 0));
        } else {
            const unsigned soff = (unsigned)c0 * 4u;
#pragma unroll
            for (int j = 0; j < AROWS; ++j)
                ra[s][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff[j], soff, 0));
        }
        // weights are packed [tap][Cin/16][plane][Cout][16]: one K-step's B tile is contiguous per plane
        const int wtap = p.wrev ? taps - 1 - ld_tap : ld_tap;   // data gradient: the flipped kernel
        const unsigned wsoff = (unsigned)((wtap * kchunks + ld_cc) * NP) * wplane;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            rb[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff, wsoff + pl * wplane, 0);
        const bool wrap = ++ld_cc == kchunks;
        ld_cc = wrap ? 0 : ld_cc;
        ld_tap += wrap ? 1 : 0;
    };

    auto store = [&](auto sc) {
        constexpr int s = decltype(sc)::value;      // register set s -> LDS buffer s
        unsigned char* as = As + s * NP * PLANE_A;
        unsigned char* bs = Bs + s * NP * PLANE_B;
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            u32x2 o[NP];
            split4<NP>(ra[s][j], o);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *reinterpret_cast<u32x2*>(as + pl * PLANE_A + lds_off(r0 + 64 * j, c4 >> 3) + (c4 & 4) * 2) = o[pl];
        }
        if (BN >= 128 || brow < BN) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *reinterpret_cast<u32x4*>(bs + pl * PLANE_B + lds_off(brow, bhalf)) = rb[s][pl];
        }
    };

    f32x16 acc[MI][NI];
    // lane l feeds row l&31 of each 32x32 MFMA tile with the 8 k-values of half l>>5 (tile row bases are multiples
    // of 32, so the swizzle term depends on the lane only)
    const int frag_a = lds_off(wm * WTM + (lane & 31), lane >> 5);
    const int frag_b = lds_off(wn * WTN + (lane & 31), lane >> 5);

    auto compute = [&](int buf) {
        const unsigned char* as = As + buf * NP * PLANE_A + frag_a;
        const unsigned char* bs = Bs + buf * NP * PLANE_B + frag_b;
        bf16x8 a[NP][MI], b[NP][NI];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[pl][mi] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(as + pl * PLANE_A + mi * 32 * SROW));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[pl][ni] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(bs + pl * PLANE_B + ni * 32 * SROW));
        }
        // smallest terms first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = sum; i >= 0; --i)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][mi], b[sum - i][ni],
                                                                              acc[mi][ni], 0, 0, 0);
    };

    // Scheduling hint: issue the K-step's fragment reads together, ahead of its MFMAs; the rest is left to the
    // compiler (measured: interleaving the split arithmetic and the LDS writes between the MFMAs by hand was 5 %
    // slower than hipcc's own order — MFMAs back to back, staging behind them).
    auto pipeline_hint = [&]() { __builtin_amdgcn_sched_group_barrier(0x100, (MI + NI) * NP, 0); };

    // ---- segments: maximal runs of items of one tile inside this workgroup's range ---------------------
    // Invariant at the top of a segment: register sets 0 and 1 hold (or are receiving) its first two K-steps and
    // the loader is prepared for its third.  K-step 2t uses set/buffer 0, K-step 2t+1 set/buffer 1.
    I0 i0;
    I1 i1;
    int tile = (int)(item / S);                    // (once per workgroup; tiles advance by one below)
    int pair0 = (int)(item - (long long)tile * S);
    set_loader(tile, pair0);
    issue(i0);
    issue(i1);
    while (item < item_end) {
        const long long tile_end = (long long)(tile + 1) * S;
        const long long seg_end = tile_end < item_end ? tile_end : item_end;
        const int npairs = (int)(seg_end - item);
        const int bn = tile / nbm, bm = tile - bn * nbm;

#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        store(i0);
        __syncthreads();
        for (int t = 0; t + 1 < npairs; ++t) {
            issue(i0);
            compute(0);
            store(i1);
            pipeline_hint();
            __syncthreads();
            issue(i1);
            compute(1);
            store(i0);
            pipeline_hint();
            __syncthreads();
        }
        // last pair of the segment: its prefetches belong to the next tile (if this workgroup has one)
        if (STREAMK && seg_end < item_end) set_loader(tile + 1, 0);
        issue(i0);
        compute(0);
        store(i1);
        __syncthreads();
        issue(i1);
        compute(1);
        __syncthreads();

        if (STREAMK && pair0 > 0) {
            // later K-steps of a cut tile (this worker's first segment): publish the raw accumulators
            sk_publish<BM, BN, WGM, WGN>(p, skw.id, acc);
        } else {
            // whole tile, or the first K-steps of a cut tile (this worker's last segment): add what the next workers
            // of the group published for it, then the common epilogue
            if (STREAMK && seg_end < tile_end) sk_consume<BM, BN, WGM, WGN>(p, skw, ntiles, S, tile_end, acc);
            epilogue<BM, BN, WGM, WGN, false>(p, reinterpret_cast<float*>(smem_raw), acc, bm * BM, bn * BN);
        }
        if (STREAMK) __syncthreads();  // the staging LDS is reused by the next segment
        item = seg_end;
        ++tile;
        pair0 = 0;
    }
}

template <int BM, int BN, int WGM, int WGN, int KS, bool UPCAT, int NP>
int launch_dp(hipStream_t stream, const ConvArgs& a) {
    auto kern = conv_mfma_split_kernel<BM, BN, WGM, WGN, KS, UPCAT, false, NP>;
    constexpr size_t lds = split_lds_bytes<BM, BN, NP>();
    static bool attr_set[Y3_MAX_DEVICES] = {};
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        if (int rc = set_lds_attr(kern, lds)) return rc;
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int nbm = (a.M + BM - 1) / BM;
    const int nbn = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(nbm * nbn), dim3(256), lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

template <int KS, int NP>
int launch_sk(hipStream_t stream, const ConvArgs& a) {
    constexpr int BM = 128, BN = 128, WGM = 2, WGN = 2;
    auto kern = conv_mfma_split_kernel<BM, BN, WGM, WGN, KS, false, true, NP>;
    constexpr size_t lds = split_lds_bytes<BM, BN, NP>();
    static bool attr_set[Y3_MAX_DEVICES] = {};
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        if (int rc = set_lds_attr(kern, lds)) return rc;
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.workers), dim3(256), lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

template <int KS, bool UPCAT, int NP>
int dispatch_bn(hipStream_t stream, const ConvArgs& a) {
    if (a.Cout <= 32) return launch_dp<128, 32, 4, 1, KS, UPCAT, NP>(stream, a);
    if (a.Cout <= 64) return launch_dp<128, 64, 4, 1, KS, UPCAT, NP>(stream, a);
    return launch_dp<128, 128, 2, 2, KS, UPCAT, NP>(stream, a);
}

template <int NP>
int launch_np(hipStream_t stream, const y3_conv_desc* d, ConvArgs& a, void* workspace, size_t workspace_bytes,
              const y3_sk_opts* sk) {
    if (a.xu) return dispatch_bn<1, true, NP>(stream, a);
    if (d->k == 1) return dispatch_bn<1, false, NP>(stream, a);
    const bool has_ws = workspace != nullptr && workspace_bytes >= y3_conv_workspace_bytes_impl(d) &&
                        ((uintptr_t)workspace & 15) == 0;
    if (use_streamk(a, d->k, has_ws)) {
        if (int rc = sk_prepare(stream, a, workspace, sk)) return rc;
        return launch_sk<3, NP>(stream, a);
    }
    return dispatch_bn<3, false, NP>(stream, a);
}

__global__ void pack_weights_split_kernel(const float* __restrict__ w_hwio, unsigned short* __restrict__ out,
                                          int taps, int cin, int cout, int planes, int s_ci, int s_co) {
    // out[t][ci/16][pl][co][ci%16] = plane pl of in[t*cin*cout + ci*s_ci + co*s_co]: the B tile of one K-step (16
    // input channels) is contiguous per plane, so a workgroup's weight loads are whole cache lines.
    // (s_ci, s_co) = (cout, 1) reads an HWIO kernel; (1, cin) reads it as the data-gradient conv does:
    // [tap][Cout' = forward cin][Cin' = forward cout].
    const size_t total = (size_t)taps * cin * cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const size_t r = i / cin;
        const int co = (int)(r % cout);
        const int t = (int)(r / cout);
        float rem = w_hwio[(size_t)t * cin * cout + (size_t)ci * s_ci + (size_t)co * s_co];
        for (int pl = 0; pl < planes; ++pl) {
            unsigned u = __float_as_uint(rem);
            if (planes == 2 && pl == 1) u += 0x7fffu + ((u >> 16) & 1u);
            u &= 0xffff0000u;
            out[((((size_t)t * (cin / 16) + ci / 16) * planes + pl) * cout + co) * 16 + (ci & 15)] =
                (unsigned short)(u >> 16);
            rem -= __uint_as_float(u);
        }
    }
}

int check_desc(const y3_conv_desc* d, const void* x_up, const char* who) {
    Y3_CHECK_ARG(d->k == 1 || d->k == 3, "%s: kernel_size must be 1 or 3 (got %d)", who, d->k);
    Y3_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride must be 1 or 2 (got %d)", who, d->stride);
    Y3_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "%s: non-positive dimension", who);
    Y3_CHECK_ARG(!(d->stride == 2 && (d->h % 2 || d->w % 2)), "%s: stride-2 conv needs even H,W (got %dx%d)", who,
                 d->h, d->w);
    Y3_CHECK_ARG(!(d->k == 1 && d->stride != 1), "%s: 1x1 conv must have stride 1", who);
    Y3_CHECK_ARG((x_up != nullptr) == (d->c_up > 0), "%s: x_up and c_up must agree", who);
    Y3_CHECK_ARG(d->cin % (2 * SBK) == 0, "%s: Cin must be 3 or a multiple of %d (got %d)", who, 2 * SBK, d->cin);
    if (x_up) {
        Y3_CHECK_ARG(d->k == 1 && d->stride == 1, "%s: fused upsample+concat needs a 1x1 s1 conv", who);
        Y3_CHECK_ARG(d->c_up % (2 * SBK) == 0 && d->c_up < d->cin && d->h % 2 == 0 && d->w % 2 == 0,
                     "%s: bad c_up=%d for cin=%d", who, d->c_up, d->cin);
    }
    const long long M = (long long)d->n * (d->h / d->stride) * (d->w / d->stride);
    Y3_CHECK_ARG((long long)d->n * d->h * d->w * d->cin < (1LL << 29) && M * d->cout < (1LL << 29),
                 "%s: tensor exceeds 2^29 elements (32-bit byte offsets)", who);
    return Y3_OK;
}

void fill_args(ConvArgs& a, const y3_conv_desc* d) {
    a.partial = nullptr; a.flags = nullptr; a.workers = 0; a.wrev = 0; a.tmode = 0; a.cy = a.cx = 0; a.ntaps = 0;
    a.err = nullptr; a.spin_limit = 0; a.fault = 0; a.stats = nullptr; a.bz = nullptr; a.bvec = nullptr;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cu = d->c_up; a.Cx = d->cin - d->c_up;
    a.Cout = d->cout; a.stride = d->stride; a.pad = d->k / 2; a.act = d->act;
    a.Ho = d->h / d->stride; a.Wo = d->w / d->stride;
    a.M = d->n * a.Ho * a.Wo;
}

}  // namespace

// transposed == 0: w is HWIO [k*k][cin][cout].  transposed != 0: w is [k*k][cout][cin] (what the data gradient of a
// forward conv with kernel [k*k][cout][cin'=cin] reads: its "Cin" is the forward Cout).
int y3_launch_pack_split(hipStream_t stream, const float* w_hwio, int k, int cin, int cout, int planes, void* out,
                         int transposed) {
    const size_t total = (size_t)k * k * cin * cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_split_kernel, dim3(blocks), dim3(256), 0, stream, w_hwio,
                       static_cast<unsigned short*>(out), k * k, cin, cout, planes, transposed ? 1 : cout,
                       transposed ? cin : 1);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

// Same contract as y3_launch_conv; `w` is the split-plane packing [taps][Cin/16][planes][Cout][16] bf16 (the 3->32
// stem conv keeps its fp32 HWIO weights and the exact kernel).
int y3_launch_conv_split(hipStream_t stream, const y3_conv_desc* d, int planes, const float* x, const float* x_up,
                         const void* w, const float* scale, const float* shift, const float* residual, float* y,
                         void* workspace, size_t workspace_bytes, const y3_sk_opts* sk) {
    Y3_CHECK_ARG(planes == 2 || planes == 3, "y3_conv2d_fwd_split: planes must be 2 or 3 (got %d)", planes);
    Y3_CHECK_ARG(d && x && w && scale && shift && y, "y3_conv2d_fwd_split: null pointer argument");
    if (d->cin == 3)
        return y3_launch_conv(stream, d, x, x_up, static_cast<const float*>(w), scale, shift, residual, y,
                              workspace, workspace_bytes, sk);
    if (int rc = check_desc(d, x_up, "y3_conv2d_fwd_split")) return rc;
    ConvArgs a;
    a.x = x; a.xu = x_up; a.w = static_cast<const float*>(w); a.scale = scale; a.shift = shift;
    a.resid = residual; a.y = y;
    fill_args(a, d);
    return planes == 3 ? launch_np<3>(stream, d, a, workspace, workspace_bytes, sk)
                       : launch_np<2>(stream, d, a, workspace, workspace_bytes, sk);
}

// Data gradient of a stride-1 conv on the split kernel: dx (+)= conv_same(dz, flipped kernel).  `w` is the
// transposed split packing of the [k*k][cin][dz_stride] kernel (y3_pack_conv_weights_split with transposed = 1:
// its K axis is dz_stride, its output axis the forward cin).  Stride-2 layers use the exact kernel's parity classes.
int y3_launch_conv_dgrad_split(hipStream_t stream, const y3_conv_desc* fwd, int planes, const float* dz, int dz_stride,
                               const void* w, const float* ones, const float* zeros, int accumulate, float* dx,
                               void* workspace, size_t workspace_bytes, const y3_sk_opts* sk) {
    Y3_CHECK_ARG(planes == 2 || planes == 3, "y3_conv2d_dgrad_split: planes must be 2 or 3 (got %d)", planes);
    Y3_CHECK_ARG(fwd && dz && w && ones && zeros && dx, "y3_conv2d_dgrad_split: null pointer argument");
    Y3_CHECK_ARG((fwd->k == 1 || fwd->k == 3) && fwd->stride == 1, "y3_conv2d_dgrad_split: only stride-1 1x1 / 3x3 convs");
    Y3_CHECK_ARG(fwd->c_up == 0, "y3_conv2d_dgrad_split: fused upsample+concat inputs are not supported");
    Y3_CHECK_ARG(dz_stride >= fwd->cout && dz_stride % (2 * SBK) == 0,
                 "y3_conv2d_dgrad_split: dz stride must be a multiple of %d", 2 * SBK);
    Y3_CHECK_ARG(fwd->cin % 4 == 0 && fwd->n > 0 && fwd->h > 0 && fwd->w > 0,
                 "y3_conv2d_dgrad_split: Cin must be a multiple of 4");
    y3_conv_desc d = *fwd;              // the gradient conv: [n,h,w,dz_stride] -> [n,h,w,cin]
    d.cin = dz_stride; d.c_up = 0; d.cout = fwd->cin; d.act = 0;
    const long long M = (long long)fwd->n * fwd->h * fwd->w;
    Y3_CHECK_ARG(M * fwd->cin < (1LL << 29) && M * dz_stride < (1LL << 29),
                 "y3_conv2d_dgrad_split: tensor exceeds 2^29 elements (32-bit byte offsets)");
    ConvArgs a;
    a.x = dz; a.xu = nullptr; a.w = static_cast<const float*>(w); a.scale = ones; a.shift = zeros;
    a.resid = accumulate ? dx : nullptr; a.y = dx;
    fill_args(a, &d);
    a.wrev = 1;
    return planes == 3 ? launch_np<3>(stream, &d, a, workspace, workspace_bytes, sk)
                       : launch_np<2>(stream, &d, a, workspace, workspace_bytes, sk);
}
