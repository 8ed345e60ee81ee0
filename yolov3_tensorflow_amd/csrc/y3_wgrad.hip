// Convolution weight gradient for gfx950 (the TF autodiff of slim.conv2d w.r.t. its kernel,
// train.py:112 `optimizer.compute_gradients`).
//
//   dW[ky][kx][ci][co] = sum over output pixels m of  x[n, oy*s+ky-pad, ox*s+kx-pad, ci] * dz[m, co]
//
// As a GEMM:  D[j][co] = sum_m  P[m][j] * dz[m][co],  j = (ky*k+kx)*Cin + ci  — D is exactly the HWIO layout
// of the kernel variable, so the result needs no transposition.  P (the im2col patch matrix) is never
// materialised: its rows are gathered from x with bounds-checked buffer loads (padding reads 0).
// The reduction dimension is the pixel index m (up to 11 M at bs=64): it is split over `nsplit` workgroups
// per output tile; each split writes its partial tile to scratch and a second kernel adds the splits in a
// fixed order (deterministic, no float atomics).
// Tile: 128 (j) x 128 (co), 32 pixels per K-step, 4 waves x (2x2) 32x32 fp32 MFMA tiles — the same
// matrix-pipe-bound regime as the forward conv (K = M is long, so the prologue/epilogue are negligible).
#include <cstdlib>
#include <cmath>
#include "y3_internal.h"

namespace {

struct WgradArgs {
    const float* x;    // [N,H,W,Cin]
    const float* dz;   // [M][CoP]  (CoP = row stride of dz, >= Cout, multiple of 4)
    float* out;        // nsplit == 1: dW [J][Cout] ; else scratch [nsplit][J][Cout]
    int N, H, W, Cin, Ho, Wo, Cout, CoP;
    int k, stride, pad;
    int M, J;          // J = k*k*Cin
    int chunk;         // pixels per split (multiple of 32)
    int nsplit;
};

constexpr int WBK = 32;          // pixels per K-step
constexpr int WLD = 128;         // LDS row stride (floats): lanes of a half-wave read consecutive floats
constexpr unsigned OOB = 0x80000000u;

// BNT = output-channel tile (128 / 64 / 32), waves arranged WGM x WGN over (j, co)
template <int BNT, int WGM, int WGN>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradArgs p) {
    constexpr int WTM = 128 / WGM, WTN = BNT / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    static_assert(WGM * WGN == 4 && MI >= 1 && NI >= 1, "bad wave layout");
    constexpr int BC4 = BNT / 4;            // float4 columns of a dz row inside the tile
    constexpr int BRP = 256 / BC4;          // dz rows staged per pass
    constexpr int BPASS = WBK / BRP;        // passes (128: 4, 64: 2, 32: 1)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][32][128]  patch rows  P[m][j]
    float* Bs = smem + 2 * WBK * WLD;   // [2][32][BNT]  dz rows     dz[m][co]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nct = (p.Cout + BNT - 1) / BNT;
    const int jt = blockIdx.x / nct, ct = blockIdx.x - jt * nct;
    const int j0 = jt * 128, co0 = ct * BNT;
    const int split = blockIdx.y;
    const int m_begin = split * p.chunk;
    const int m_end = min(m_begin + p.chunk, ((p.M + WBK - 1) / WBK) * WBK);
    const int T = (m_end - m_begin) / WBK;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dz), 0, (unsigned)((size_t)p.M * p.CoP * 4), 0x00020000);

    // A (patch) staging: float4 column c4 (fixed per thread), rows (tid>>5) + 8*i
    const int c4 = (tid & 31) * 4;
    const int r0 = tid >> 5;
    const int j = j0 + c4;
    const bool j_ok = j < p.J;
    const int tap = j_ok ? j / p.Cin : 0;
    const int ci = j - tap * p.Cin;
    const int ky = tap / p.k, kx = tap - ky * p.k;
    // B (dz) staging: float4 column bc4, rows (tid / BC4) + BRP*i
    const int bc4 = (tid % BC4) * 4;
    const int br0 = tid / BC4;
    const bool co_ok = co0 + bc4 < p.CoP;

    // pixel coordinates of this thread's 4 patch rows, advanced incrementally by 32 pixels per K-step
    int pn[4], poy[4], pox[4];
    {
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_begin + r0 + 8 * i;
            pn[i] = m / HoWo;
            const int rem = m - pn[i] * HoWo;
            poy[i] = rem / p.Wo;
            pox[i] = rem - poy[i] * p.Wo;
        }
    }

    f32x4 ra[4], rb[BPASS];
    auto load_tile = [&](int t) {
        const int mb = m_begin + t * WBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned offa = OOB;
            const int iy = poy[i] * p.stride - p.pad + ky, ix = pox[i] * p.stride - p.pad + kx;
            if (j_ok && pn[i] < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                offa = (unsigned)(((pn[i] * p.H + iy) * p.W + ix) * p.Cin + ci) * 4u;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, offa, 0, 0));
            // advance 32 pixels
            pox[i] += WBK;
            while (pox[i] >= p.Wo) {
                pox[i] -= p.Wo;
                if (++poy[i] == p.Ho) { poy[i] = 0; ++pn[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            // rows m >= M lie beyond the end of dz: the buffer load returns 0 for them
            const int m = mb + br0 + BRP * i;
            const unsigned offb = co_ok ? (unsigned)(m * p.CoP + co0 + bc4) * 4u : OOB;
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_z, offb, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + (buf * WBK + r0 + 8 * i) * WLD + c4) = ra[i];
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            *reinterpret_cast<f32x4*>(Bs + (buf * WBK + br0 + BRP * i) * BNT + bc4) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // MFMA operands: A[i][k] = P[m=k][j=i], B[k][n] = dz[m=k][co=n]; lane l holds k = l>>5
    const int fcol = lane & 31, fk = lane >> 5;
    auto compute_tile = [&](int buf) {
        const float* as = As + buf * WBK * WLD + fk * WLD + wm * WTM + fcol;
        const float* bs = Bs + buf * WBK * BNT + fk * BNT + wn * WTN + fcol;
#pragma unroll
        for (int s = 0; s < WBK / 2; ++s) {
            float a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = as[(2 * s) * WLD + mi * 32];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = bs[(2 * s) * BNT + ni * 32];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // Same MFMA-shadow interleave as the forward conv (y3_conv.hip): global loads behind the first MFMAs, one
    // fragment read per MFMA, LDS writes behind the last ones.
    auto pipeline_hint = [&]() {
        constexpr int NM = MI * NI * (WBK / 2);          // MFMAs per K-step
        constexpr int NR = (MI + NI) * (WBK / 2);        // ds_read_b32 per K-step
        constexpr int NV = 4 + BPASS;                    // buffer loads per K-step
        __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (i < NR - (MI + NI)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i >= NM - NV) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    if (T > 0) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int t = 0; t + 1 < T; ++t) {
            load_tile(t + 1);
            compute_tile(t & 1);
            store_tile((t + 1) & 1);
            pipeline_hint();
            __syncthreads();
        }
        compute_tile((T - 1) & 1);
    }

    // D layout: col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (j)
    float* out = p.out + (size_t)split * p.J * p.Cout;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = co0 + wn * WTN + ni * 32 + (lane & 31);
        if (co >= p.Cout) continue;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int jb = j0 + wm * WTM + mi * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = jb + (r & 3) + 8 * (r >> 2);
                if (jj < p.J) out[(size_t)jj * p.Cout + co] = acc[mi][ni][r];
            }
        }
    }
}

// dw = sum over the splits, in a FIXED order: four lanes per element group each add every fourth split, then
// (q0 + q1) + (q2 + q3).  A lane holds FOUR consecutive elements (n is a multiple of 4: J = k*k*Cin with Cin % 4 == 0, or the
// stem's 27 x 32; every split starts 16-byte aligned), 64 such groups per workgroup; element by element the same additions
// in the same order as one float per lane.  `dw` itself may sit at any 4-byte boundary of the caller's flat gradient buffer
// (a 255-element bias tensor before it): f32x4_u = one dwordx4 store at dword alignment.
__global__ void __launch_bounds__(256) wgrad_sum_splits_kernel(const float* __restrict__ scratch, int nsplit,
                                                               long long n4, float* __restrict__ dw) {
    __shared__ f32x4 part[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(scratch);
    for (long long base = (long long)blockIdx.x * 64; base < n4; base += (long long)gridDim.x * 64) {
        const long long i = base + e;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < n4)
            for (int k = q; k < nsplit; k += 4) s += s4[(size_t)k * n4 + i];
        part[q][e] = s;
        __syncthreads();
        if (q == 0 && i < n4)
            *reinterpret_cast<f32x4_u*>(dw + 4 * i) = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
        __syncthreads();
    }
}

// The stem's partial tiles: few elements (27 x 32), many splits (one per workgroup of stem_wgrad_kernel, up to 1,280).  One
// workgroup per kernel row j (32 output channels): lane (co, g) adds the splits k = g, g + 8, ... in order, the eight group
// sums are added pairwise in a fixed order.  (The general kernel above gave this case four workgroups and 512 dependent
// loads per lane: 0.66 ms per bs=64 step, profiles/r05_train_c4_kernel_stats.csv.)
__global__ void __launch_bounds__(256) stem_sum_splits_kernel(const float* __restrict__ part, int nsplit,
                                                              float* __restrict__ dw) {
    __shared__ float red[8][32];
    const int co = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + co;
    float s0 = 0.f, s1 = 0.f;
    int k = g;
    for (; k + 8 < nsplit; k += 16) {                      // two independent chains (k = g mod 16, g + 8 mod 16)
        s0 += part[(size_t)k * (27 * 32) + e];
        s1 += part[(size_t)(k + 8) * (27 * 32) + e];
    }
    if (k < nsplit) s0 += part[(size_t)k * (27 * 32) + e];
    red[g][co] = s0 + s1;
    __syncthreads();
    if (g == 0)
        dw[e] = ((red[0][co] + red[1][co]) + (red[2][co] + red[3][co])) + ((red[4][co] + red[5][co]) + (red[6][co] + red[7][co]));
}

// Stem conv (Cin = 3): D[27][32] = sum over pixels of patch[m][27] * dz[m][32].  One 32x32 MFMA tile (kernel rows j = (ky*3 +
// kx)*3 + ci padded 27 -> 32 with zeros).  A workgroup walks PIECES of 128 consecutive pixels of one image row:
//   * the patch matrix is never built: the three image rows under the piece are staged as they lie in memory - 130 pixels x 3
//     channels = 390 consecutive floats per row, coalesced, zeros outside the image - and lane (j, pixel parity) of an MFMA
//     reads its A value at  ky*XRL + (j % 9) + 3*pixel  (j % 9 = kx*3 + ci: pixel p under tap kx is staged pixel p + kx).
//     XRL = 394 = 10 mod 32: the 27 lanes of a half-wave hit 27 different banks; j >= 27 reads a zero word with stride 0.
//     (The first version gathered 16 patch values per thread and 128 pixels with one bounds-checked 4-byte load each:
//     0.96-1.02 ms per bs=64 step for 1.55 GB of x and dz, 1.6 TB/s; profiles/r05_train_c4_kernel_stats.csv.)
//   * dz rows: 128 x 32 floats, coalesced float4, zeros past the row end (the last piece of a 416-pixel row has 32 pixels);
//   * the loads of piece t + 1 are issued before the MFMAs of piece t and held in registers;
//   * each wave accumulates a quarter of the piece's pixels; the four wave accumulators are summed through LDS at the end.
constexpr int XRL = 394;                       // floats per staged image row (390 used)
constexpr int XZERO = 3 * XRL;                 // index of the zero word behind the three rows
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                         int N, int H, int W, int pieces_per_row, int pieces, int chunk,
                                                         float* __restrict__ partial /*[grid][27][32]*/) {
    constexpr int TP = 128;
    __shared__ __attribute__((aligned(16))) float xs[3 * XRL + 4];
    __shared__ __attribute__((aligned(16))) float zs[TP * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int p_begin = blockIdx.x * chunk, p_end = min(p_begin + chunk, pieces);
    f32x4 zr[4];
    float xr[3][2];
    auto load = [&](int piece) {
        const int row = piece / pieces_per_row, pc = piece - row * pieces_per_row;
        const int n = row / H, oy = row - n * H;
        const int px0 = pc * TP;
        const size_t m0 = ((size_t)n * H + oy) * W + px0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;              // float4 index inside the piece
            const int p = e >> 3, c4 = (e & 7) * 4;
            zr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (px0 + p < W) zr[i] = *reinterpret_cast<const f32x4*>(dz + (m0 + p) * 32 + c4);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy - 1 + r;
            const bool rok = (unsigned)iy < (unsigned)H;
            const float* xrow = x + ((size_t)n * H + (rok ? iy : 0)) * W * 3;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = tid + 256 * u;                   // float index inside the staged row
                const int f = (px0 - 1) * 3 + q;               // float index inside the image row
                xr[r][u] = (rok && q < 390 && f >= 0 && f < W * 3) ? xrow[f] : 0.f;
            }
        }
    };
    if (tid < 4) xs[XZERO + tid] = 0.f;
    if (p_begin < p_end) load(p_begin);
    // A operand: lane (j = lane & 31, h = lane >> 5), MFMA s2 of wave w reads pixel 32 w + 2 s2 + h
    const int j = lane & 31;
    const int a_stride = j < 27 ? 3 : 0;
    const float* as = xs + (j < 27 ? (j / 9) * XRL + (j % 9) : XZERO) + a_stride * (wave * 32 + (lane >> 5));
    const float* bs = zs + (wave * 32 + (lane >> 5)) * 32 + (lane & 31);
    for (int piece = p_begin; piece < p_end; ++piece) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            *reinterpret_cast<f32x4*>(zs + (e >> 3) * 32 + (e & 7) * 4) = zr[i];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            xs[r * XRL + tid] = xr[r][0];
            if (tid < 390 - 256) xs[r * XRL + 256 + tid] = xr[r][1];
        }
        __syncthreads();
        if (piece + 1 < p_end) load(piece + 1);   // in flight under this piece's MFMAs (and the other workgroups of the CU)
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s2 * 2 * a_stride], bs[s2 * 64], acc, 0, 0, 0);
        __syncthreads();
    }
    // sum the four waves: D layout col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (j)
    float* red = zs;                              // [4][32][32] = 16 KB: zs is free now
#pragma unroll
    for (int r = 0; r < 16; ++r)
        red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
    __syncthreads();
    for (int e = tid; e < 27 * 32; e += 256)
        partial[(size_t)blockIdx.x * 27 * 32 + e] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
}

}  // namespace

static int wgrad_co_tile(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

// Split choice: the number of pixel-range splits per output tile that minimises a small cost model
//   rounds(tiles * nsplit over the 512 co-resident workgroups: 256 CUs x 2) x (K-steps per split x t_K + t_fixed)
//   + the write + read of the nsplit partial tiles.
// What matters is the FIRST term's quantisation: the previous rule ("about 1024 workgroups") produced e.g. 18 tiles x 57
// = 1026 workgroups = 2.004 rounds, i.e. three rounds of which the last runs two workgroups.
static void wgrad_split(const y3_conv_desc* d, int* nsplit_out, int* chunk_out) {
    const long long M = (long long)d->n * (d->h / d->stride) * (d->w / d->stride);
    const int J = d->k * d->k * d->cin;
    const int bnt = wgrad_co_tile(d->cout);
    const int tiles = ((J + 127) / 128) * ((d->cout + bnt - 1) / bnt);
    const int ksteps = (int)((M + WBK - 1) / WBK);
    constexpr double SLOTS = 512.0;            // co-resident workgroups
    constexpr double T_K = 3.9, T_FIX = 6.0;   // us per K-step with two workgroups per CU; prologue + epilogue
    const double partial_us = 2.0 * (double)J * d->cout * 4.0 / 3.0e6;   // one partial tile set written + read at ~3 TB/s
    if (const char* e = y3_exp_env("Y3_WGRAD_OLD_SPLIT"); e && e[0] == '1') {   // experiment hook: the round-1 rule
        int nsplit = (1024 + tiles - 1) / tiles;
        if (nsplit > 512) nsplit = 512;
        if (nsplit > ksteps) nsplit = ksteps;
        if (nsplit < 1) nsplit = 1;
        const int chunk = ((ksteps + nsplit - 1) / nsplit) * WBK;
        *chunk_out = chunk;
        *nsplit_out = (int)((M + chunk - 1) / chunk);
        return;
    }
    int best_ns = 1, best_chunk = ksteps;
    double best = 1e300;
    const int ns_max = ksteps < 512 ? ksteps : 512;
    for (int ns = 1; ns <= ns_max; ++ns) {
        const int chunk = (ksteps + ns - 1) / ns;
        const int real = (ksteps + chunk - 1) / chunk;          // splits that actually get pixels
        if (real != ns) continue;
        const double rounds = ceil((double)tiles * real / SLOTS);
        const double t = rounds * (chunk * T_K + T_FIX) + (real > 1 ? real * partial_us : 0.0);
        if (t < best) { best = t; best_ns = real; best_chunk = chunk; }
    }
    *chunk_out = best_chunk * WBK;
    *nsplit_out = best_ns;
}

extern "C" size_t y3_conv_wgrad_scratch_bytes(const y3_conv_desc* d) {
    if (!d || d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->stride <= 0) return 0;
    if (d->cin == 3) return (size_t)4096 * 27 * 32 * sizeof(float);
    int nsplit, chunk;
    wgrad_split(d, &nsplit, &chunk);
    return (size_t)nsplit * d->k * d->k * d->cin * d->cout * sizeof(float) + 256;
}

extern "C" int y3_conv_wgrad(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* dz, int dz_stride,
                             float* dw_hwio, void* scratch, size_t scratch_bytes) {
    Y3_CHECK_ARG(ctx && d && x && dz && dw_hwio && scratch, "y3_conv_wgrad: null argument");
    Y3_CHECK_ARG(d->k == 1 || d->k == 3, "y3_conv_wgrad: kernel_size must be 1 or 3");
    Y3_CHECK_ARG(d->stride == 1 || d->stride == 2, "y3_conv_wgrad: stride must be 1 or 2");
    Y3_CHECK_ARG(d->c_up == 0, "y3_conv_wgrad: fused upsample+concat inputs are not supported (materialise the concat)");
    Y3_CHECK_ARG(dz_stride >= d->cout && dz_stride % 4 == 0, "y3_conv_wgrad: dz row stride must be >= Cout and a multiple of 4");
    Y3_CHECK_ARG(scratch_bytes >= y3_conv_wgrad_scratch_bytes(d), "y3_conv_wgrad: scratch too small");
    Y3_CHECK_ARG((reinterpret_cast<size_t>(scratch) & 15) == 0, "y3_conv_wgrad: scratch must be 16-byte aligned");
    const int Ho = d->h / d->stride, Wo = d->w / d->stride;
    const long long M = (long long)d->n * Ho * Wo;
    Y3_CHECK_ARG((long long)d->n * d->h * d->w * d->cin < (1LL << 29) && M * dz_stride < (1LL << 29),
                 "y3_conv_wgrad: tensor exceeds 2^29 elements (32-bit byte offsets)");
    hipStream_t st = ctx->stream;
    if (d->cin == 3) {
        Y3_CHECK_ARG(d->k == 3 && d->cout == 32 && d->stride == 1 && dz_stride == 32,
                     "y3_conv_wgrad: Cin=3 is supported only as the 3x3 3->32 stem conv");
        // pieces of 128 pixels of one image row; one round of co-resident workgroups (21 KB of LDS each: six per CU)
        const int ppr = (d->w + 127) / 128;
        const long long pieces = (long long)d->n * d->h * ppr;
        Y3_CHECK_ARG(pieces < (1LL << 30), "y3_conv_wgrad: too many pixels");
        int nblk = (int)(pieces < 1536 ? pieces : 1536);
        const int chunk = (int)((pieces + nblk - 1) / nblk);
        nblk = (int)((pieces + chunk - 1) / chunk);
        float* part = static_cast<float*>(scratch);
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, x, dz, d->n, d->h, d->w, ppr, (int)pieces,
                           chunk, part);
        Y3_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(stem_sum_splits_kernel, dim3(27), dim3(256), 0, st, part, nblk, dw_hwio);
        Y3_CHECK_HIP(hipGetLastError());
        return Y3_OK;
    }
    Y3_CHECK_ARG(d->cin % 4 == 0, "y3_conv_wgrad: Cin must be 3 or a multiple of 4");
    WgradArgs a;
    a.x = x; a.dz = dz;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Ho = Ho; a.Wo = Wo; a.Cout = d->cout; a.CoP = dz_stride;
    a.k = d->k; a.stride = d->stride; a.pad = d->k / 2; a.M = (int)M; a.J = d->k * d->k * d->cin;
    const int bnt = wgrad_co_tile(a.Cout);
    const int tiles = ((a.J + 127) / 128) * ((a.Cout + bnt - 1) / bnt);
    int nsplit;
    wgrad_split(d, &nsplit, &a.chunk);
    a.nsplit = nsplit;
    a.out = nsplit == 1 ? dw_hwio : static_cast<float*>(scratch);
    static bool attr_set[Y3_MAX_DEVICES] = {};
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<128, 2, 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(2 * WBK * (WLD + 128) * sizeof(float))));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const size_t lds = (size_t)2 * WBK * (WLD + bnt) * sizeof(float);
    if (bnt == 128)
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 2, 2>), dim3(tiles, nsplit), dim3(256), lds, st, a);
    else if (bnt == 64)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 2, 2>), dim3(tiles, nsplit), dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 4, 1>), dim3(tiles, nsplit), dim3(256), lds, st, a);
    Y3_CHECK_HIP(hipGetLastError());
    if (nsplit > 1) {
        const long long n4 = (long long)a.J * a.Cout / 4;     // J = k*k*Cin, Cin % 4 == 0
        long long nb = (n4 + 63) / 64;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(wgrad_sum_splits_kernel, dim3((int)nb), dim3(256), 0, st, static_cast<float*>(scratch),
                           nsplit, n4, dw_hwio);
        Y3_CHECK_HIP(hipGetLastError());
    }
    return Y3_OK;
}
