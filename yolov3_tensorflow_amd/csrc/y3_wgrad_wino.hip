// Weight gradient of the stride-1 3x3 convs in Winograd F(2x2,3x3) form (train.py:112
// `optimizer.compute_gradients` for the kernels of utils/layer_utils.py:9-22), exact fp32 arithmetic on
// v_mfma_f32_32x32x2_f32 with 16/36 of the direct algorithm's multiplies:
//
//   dg = G^T [ sum over 2x2 output tiles t of (B^T d_t B) .* (A dY_t A^T) ] G        per (ci, co)
//
// d_t = the 4x4 input patch of tile t (channel ci), dY_t = the 2x2 tile of dz (channel co).  The sum over tiles is 16
// independent GEMMs  dU[pos][ci][co] = sum_t V[pos][t][ci] * Z[pos][t][co]  whose reduction axis is the tile index:
//   * a workgroup owns 64 input channels x 64 output channels for ALL 16 transform positions (each of its four waves
//     accumulates 16 independent 32x32 products = 256 accumulator registers, one wave per SIMD — the accumulator shape
//     of the forward Winograd kernel, y3_conv_wino.hip) over a contiguous range of tiles (one of `nsplit` splits);
//   * K-step = 8 tiles = 64 MFMAs per wave.  Staging, per thread and K-step: lane = channel (64 consecutive channels =
//     256 B per pixel: coalesced), wave = tile pair: the two 4x4 patches of its input channel (32 4-byte loads, padding
//     = OOB = 0), B^T d B on each; the two 2x2 dz tiles of its output channel (8 loads), A dY A^T on each; 16 + 16
//     8-byte LDS writes (the two tiles of a pair side by side);
//   * LDS row = (position, channel) = 8 tiles = four 8-byte tile pairs, the pair's slot rotated by (channel >> 2) & 3:
//     both the staging writes (64 lanes = 64 rows, one pair each) and the fragment reads (lanes 0-31 read pairs 0,1 of
//     their row, lanes 32-63 pairs 2,3) spread evenly over the banks.  Which tile plays "k" in which MFMA is free as long
//     as both operands agree: lane half h, step s uses tile 4h + s;
//   * epilogue: the 16 position sums of one (ci, co) live in the same lane and register index of the 16 accumulator
//     sets, so G^T dU G (4x4 -> 3x3) is register arithmetic; the nine taps go straight to the HWIO layout of the kernel
//     variable (or of this split's partial tile, which wgrad_sum_splits adds in a fixed order: deterministic).
#include <cstdlib>
#include "y3_internal.h"

namespace {

struct WgWinoArgs {
    const float* x;     // [N,H,W,Cin]
    const float* dz;    // [N,H,W][CoP]  (CoP = row stride of dz >= Cout)
    float* out;         // nsplit == 1: dW [9][Cin][Cout]; else scratch [nsplit][9][Cin][Cout]
    int N, H, W, Cin, Cout, CoP;
    int TH, TW, T;      // 2x2 tiles per image column / row, and in total
    int adv_r, adv_c;   // 8 tiles further = adv_r tile rows + adv_c tile columns (8 / TW, 8 % TW); adv_r + 1 <= TH
    int ksteps;         // K-steps (8 tiles) per split
    int nsplit;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;
constexpr int WROW = 32;                       // LDS bytes per (position, channel) row: 8 tiles
constexpr int PLANE = 64 * WROW;               // bytes per transform position (64 channels)
constexpr int STAGE = 16 * PLANE;              // one operand, one stage

// byte offset of tile pair `pair` (0..3) inside row `ch`
__device__ __forceinline__ int row_off(int ch, int pair) { return ch * WROW + (((pair + (ch >> 2)) & 3) << 3); }

// B^T d B (d[i*4+j], i = patch row) -> v[pos = i*4+j]
__device__ __forceinline__ void input_transform(const float (&d)[16], float (&v)[16]) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i * 4 + 0] = t[i * 4 + 0] - t[i * 4 + 2];
        v[i * 4 + 1] = t[i * 4 + 1] + t[i * 4 + 2];
        v[i * 4 + 2] = t[i * 4 + 2] - t[i * 4 + 1];
        v[i * 4 + 3] = t[i * 4 + 1] - t[i * 4 + 3];
    }
}
// A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]; y[p*2+q] -> z[pos = i*4+j]
__device__ __forceinline__ void grad_transform(const float (&y)[4], float (&z)[16]) {
    float r[4][2];      // rows: A dY
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        r[0][q] = y[0 * 2 + q];
        r[1][q] = y[0 * 2 + q] + y[1 * 2 + q];
        r[2][q] = y[0 * 2 + q] - y[1 * 2 + q];
        r[3][q] = -y[1 * 2 + q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        z[i * 4 + 0] = r[i][0];
        z[i * 4 + 1] = r[i][0] + r[i][1];
        z[i * 4 + 2] = r[i][0] - r[i][1];
        z[i * 4 + 3] = -r[i][1];
    }
}

__global__ void __launch_bounds__(256, 1) conv_wgrad_wino_kernel(const WgWinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Vs = smem;                   // [2][16][64 ci][32 B]
    unsigned char* Zs = smem + 2 * STAGE;       // [2][16][64 co][32 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nob = p.Cout >> 6;
    const int cb = blockIdx.x / nob, ob = blockIdx.x - cb * nob;     // 64-channel input / output blocks
    const int split = blockIdx.y;
    const int ks_begin = split * p.ksteps;
    const int ks_total = (p.T + 7) >> 3;
    const int ks_end = min(ks_begin + p.ksteps, ks_total);
    if (ks_begin >= ks_end) return;             // (the launcher sizes nsplit so that no split is empty)

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((size_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dz), 0, (unsigned)((size_t)p.N * p.H * p.W * p.CoP * 4), 0x00020000);

    // staging role: lane = channel inside the block, wave = tile pair of the K-step (tiles 2*wave, 2*wave + 1).
    // Everything about the TILE is wave-uniform and kept in scalar registers (readfirstlane makes it provably uniform):
    // its coordinates, the validity of its patch rows / columns, and the pixel part of every address, which goes into
    // the scalar offset of the buffer load; the per-lane part is the channel.  All of it is branch-free bit arithmetic:
    // a uniform `?:` becomes a real branch in hipcc's output and would cut the K-step into dozens of basic blocks.
    const unsigned ci_b = (unsigned)(cb * 64 + lane) * 4u, co_b = (unsigned)(ob * 64 + lane) * 4u;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // tile coordinates of this wave's two tiles: integer divisions once, then advanced by 8 tiles per K-step with
    // compare + select (no division, no vector ALU, no branch in the K-loop)
    int tn[2], tty[2], ttx[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int t = ks_begin * 8 + wv * 2 + e;
        tn[e] = t / (p.TH * p.TW);
        const int rem = t - tn[e] * p.TH * p.TW;
        tty[e] = rem / p.TW;
        ttx[e] = rem - tty[e] * p.TW;
    }
    const unsigned cin4 = (unsigned)p.Cin * 4u, cop4 = (unsigned)p.CoP * 4u, wcin4 = (unsigned)p.W * cin4,
                   wcop4 = (unsigned)p.W * cop4;
    // Two register sets: the loads of K-step s+2 are issued during K-step s and consumed (transform + LDS writes) at the
    // end of K-step s+1 — a whole K-step (~5k cycles) of latency cover; the accumulators live in AGPRs, so the VGPRs are there.
    float rdA[2][16], ryA[2][4], rdB[2][16], ryB[2][4];
    auto issue = [&](float (&rd)[2][16], float (&ry)[2][4]) {   // loads of the K-step the tile coordinates point at; then advance
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int n = tn[e], ty = tty[e], tx = ttx[e];
            const bool tok = n < p.N;                             // tiles past the end read zeros
            {
                int x = tx + p.adv_c;
                const int c1 = (int)(x >= p.TW);
                x -= p.TW & -c1;
                int y = ty + p.adv_r + c1;
                const int c2 = (int)(y >= p.TH);
                y -= p.TH & -c2;
                ttx[e] = x; tty[e] = y; tn[e] = n + c2;
            }
            // (a tile past the end gets a row coordinate no image has: every patch row fails the range test below)
            const int y0 = (2 * ty - 1) | (0x40000000 & -(int)!tok), x0 = 2 * tx - 1;
            // Per patch row / column: its byte offset (scalar) and 0 or OOB (scalar) for "no such pixel".  An invalid
            // pixel sets the top bit of the VGPR offset, which alone fails the buffer's range check (the scalar offset
            // is not part of that check), so the load returns 0 whatever its scalar offset is.
            // (readfirstlane: hipcc evaluates these uniform compares on the vector ALU and would then wrap every load
            // in a waterfall loop to get its scalar offset.)
            unsigned rowx[4], colx[4], rowz[4], colz[4], rowb[4], colb[4];
            const unsigned ro = (unsigned)((n * p.H + y0) * p.W), cx = (unsigned)x0;      // may wrap: masked below
            rowx[0] = ro * cin4; colx[0] = cx * cin4; rowz[0] = ro * cop4; colz[0] = cx * cop4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rowb[i] = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)(OOB & (0u - (unsigned)((unsigned)(y0 + i) >= (unsigned)p.H))));
                colb[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)(OOB & (0u - (unsigned)((unsigned)(x0 + i) >= (unsigned)p.W))));
                if (i > 0) {
                    rowx[i] = rowx[i - 1] + wcin4; colx[i] = colx[i - 1] + cin4;
                    rowz[i] = rowz[i - 1] + wcop4; colz[i] = colz[i - 1] + cop4;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned vrow = ci_b | rowb[i];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rd[e][i * 4 + j] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vrow | colb[j], rowx[i] + colx[j], 0));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {                 // output pixel (2ty+i, 2tx+j) = patch pixel (i+1, j+1)
                const unsigned vrow = co_b | rowb[i + 1];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ry[e][i * 2 + j] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, vrow | colb[j + 1], rowz[i + 1] + colz[j + 1], 0));
            }
        }
    };
    const int st_off = row_off(lane, wave);       // this thread's 8-byte slot inside a (position, channel) row
    auto store = [&](int buf, const float (&rd)[2][16], const float (&ry)[2][4]) {
        float va[16], vb[16];
        input_transform(rd[0], va);
        input_transform(rd[1], vb);
        unsigned char* vs = Vs + buf * STAGE + st_off;
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) *reinterpret_cast<f32x2*>(vs + pos * PLANE) = f32x2{va[pos], vb[pos]};
        float za[16], zb[16];
        grad_transform(ry[0], za);
        grad_transform(ry[1], zb);
        unsigned char* zs = Zs + buf * STAGE + st_off;
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) *reinterpret_cast<f32x2*>(zs + pos * PLANE) = f32x2{za[pos], zb[pos]};
    };

    f32x16 acc[16];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pos][r] = 0.f;

    // fragments: lane (i = lane & 31, h = lane >> 5) reads tile pairs 2h and 2h+1 of row i: tiles 4h .. 4h+3
    const int h = lane >> 5;
    const int fa0 = row_off(wm * 32 + (lane & 31), 2 * h), fa1 = row_off(wm * 32 + (lane & 31), 2 * h + 1);
    const int fb0 = row_off(wn * 32 + (lane & 31), 2 * h), fb1 = row_off(wn * 32 + (lane & 31), 2 * h + 1);
    auto compute = [&](int buf) {
        const unsigned char* vs = Vs + buf * STAGE;
        const unsigned char* zs = Zs + buf * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {             // position groups of four: four independent accumulators in rotation
            f32x2 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pos = g * 4 + i;
                a0[i] = *reinterpret_cast<const f32x2*>(vs + pos * PLANE + fa0);
                a1[i] = *reinterpret_cast<const f32x2*>(vs + pos * PLANE + fa1);
                b0[i] = *reinterpret_cast<const f32x2*>(zs + pos * PLANE + fb0);
                b1[i] = *reinterpret_cast<const f32x2*>(zs + pos * PLANE + fb1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a = s < 2 ? a0[i][s & 1] : a1[i][s & 1];
                    const float b = s < 2 ? b0[i][s & 1] : b1[i][s & 1];
                    acc[g * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g * 4 + i], 0, 0, 0);
                }
        }
    };

    // One K-step = one basic block: 64 MFMAs on stage `cur` | the 40 loads of K-step ks+2 into `ld` (+ their scalar
    // address arithmetic) | 32 fragment reads | transforms + 16 LDS writes of K-step ks+1 (registers `use`, loaded one
    // K-step ago) into the other stage.  The hints pin the classes that have a natural place (position group = 16 MFMAs):
    //   group 0: 16 x (MFMA, load), the fragments of group 1 behind the last 8
    //   group 1: 16 x (MFMA, load), the fragments of group 2 behind the last 8
    //   group 2:  8 x (MFMA, load), 8 x (MFMA, fragment read of group 3)
    //   group 3: 16 x (MFMA, LDS write) — the transforms float in front of the writes
    // (the ~150 scalar instructions of the address arithmetic stay ahead of the first MFMA: SALU / VALU groups in this list
    // were tried and made hipcc bunch the MFMAs instead)
    auto kstep = [&](int cur, float (&ld_d)[2][16], float (&ld_y)[2][4], const float (&use_d)[2][16],
                     const float (&use_y)[2][4]) {
        issue(ld_d, ld_y);
        compute(cur);
        store(cur ^ 1, use_d, use_y);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                 // fragments of group 0
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __syncthreads();
    };

    issue(rdA, ryA);                  // K-step ks_begin
    store(0, rdA, ryA);
    issue(rdB, ryB);                  // K-step ks_begin + 1 (tiles past this split's range are loaded but never staged)
    __syncthreads();
    int ks = ks_begin;
    for (; ks + 1 < ks_end; ++ks) {   // a further K-step remains to be staged
        // rotate the register sets: B was loaded during the previous K-step (a whole K-step ago: no stall), A is free
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rdA[e][i] = rdB[e][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) ryA[e][i] = ryB[e][i];
        }
        kstep((ks - ks_begin) & 1, rdB, ryB, rdA, ryA);
    }
    compute((ks - ks_begin) & 1);

    // G^T dU G per (ci, co) in registers; D layout of the 32x32 MFMA: row (ci) = (r&3) + 8*(r>>2) + 4*(lane>>5),
    // column (co) = lane & 31
    float* out = p.out + (size_t)split * 9 * p.Cin * p.Cout;
    const int co = ob * 64 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci = cb * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float m[16];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            m[pos] = acc[pos][r];
            asm volatile("" : "+v"(m[pos]));      // one AGPR read per value
        }
        float q[3][4];          // G^T dU : q[a][j]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = 0.5f * (m[1 * 4 + j] + m[2 * 4 + j]);
            q[0][j] = m[0 * 4 + j] + s;
            q[1][j] = 0.5f * (m[1 * 4 + j] - m[2 * 4 + j]);
            q[2][j] = s + m[3 * 4 + j];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float s = 0.5f * (q[a][1] + q[a][2]);
            const float g0 = q[a][0] + s, g1 = 0.5f * (q[a][1] - q[a][2]), g2 = s + q[a][3];
            float* o = out + ((size_t)(a * 3) * p.Cin + ci) * p.Cout + co;
            o[0] = g0;
            o[(size_t)p.Cin * p.Cout] = g1;
            o[(size_t)2 * p.Cin * p.Cout] = g2;
        }
        __builtin_amdgcn_sched_barrier(0);       // one accumulator row at a time (as in y3_conv_wino.hip)
    }
}

// dw = sum over the splits in a fixed order (same scheme as y3_wgrad.hip: four lanes per group of four consecutive elements -
// n = 9 * Cin * Cout is a multiple of 4 -, each adds every fourth split, then (q0 + q1) + (q2 + q3); dw at dword alignment)
__global__ void __launch_bounds__(256) wgw_sum_splits_kernel(const float* __restrict__ scratch, int nsplit, long long n4,
                                                             float* __restrict__ dw) {
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

constexpr int WGW_SLOTS = 256;      // one workgroup per CU (128 KB of LDS, 256 accumulator registers per wave)

void wgw_split(const y3_conv_desc* d, int* nsplit, int* ksteps) {
    const long long T = (long long)d->n * ((d->h + 1) / 2) * ((d->w + 1) / 2);
    const int ks_total = (int)((T + 7) / 8);
    const int wt = (d->cin / 64) * (d->cout / 64);
    int ns = WGW_SLOTS / wt;
    if (ns < 1) ns = 1;
    if (ns > ks_total) ns = ks_total;
    const int per = (ks_total + ns - 1) / ns;
    *ksteps = per;
    *nsplit = (ks_total + per - 1) / per;       // every split gets at least one K-step
}

}  // namespace

int y3_conv_wgrad_wino_eligible_impl(const y3_conv_desc* d) {
    if (!(d && d->k == 3 && d->stride == 1 && d->c_up == 0 && d->cin % 64 == 0 && d->cout % 64 == 0 && d->n > 0 &&
          d->h > 1 && d->w > 1))
        return 0;
    // the K-loop advances its tile coordinates by 8 tiles with one wrap per axis: 8 / TW + 1 tile rows must fit an image
    const int th = (d->h + 1) / 2, tw = (d->w + 1) / 2;
    return 8 / tw + 1 <= th;
}

size_t y3_conv_wgrad_wino_scratch_bytes_impl(const y3_conv_desc* d) {
    if (!y3_conv_wgrad_wino_eligible_impl(d)) return 0;
    int ns, ks;
    wgw_split(d, &ns, &ks);
    return (size_t)ns * 9 * d->cin * d->cout * sizeof(float) + 256;
}

int y3_launch_conv_wgrad_wino(hipStream_t stream, const y3_conv_desc* d, const float* x, const float* dz, int dz_stride,
                              float* dw_hwio, void* scratch, size_t scratch_bytes) {
    Y3_CHECK_ARG(d && x && dz && dw_hwio && scratch, "y3_conv_wgrad_wino: null argument");
    Y3_CHECK_ARG(y3_conv_wgrad_wino_eligible_impl(d),
                 "y3_conv_wgrad_wino: needs a 3x3 stride-1 conv with Cin %% 64 == 0, Cout %% 64 == 0 and a map of at least "
                 "8 / ceil(w/2) + 1 tile rows (y3_conv_wgrad_wino_eligible)");
    Y3_CHECK_ARG(dz_stride >= d->cout, "y3_conv_wgrad_wino: dz row stride must be >= Cout");
    Y3_CHECK_ARG(scratch_bytes >= y3_conv_wgrad_wino_scratch_bytes_impl(d), "y3_conv_wgrad_wino: scratch too small");
    Y3_CHECK_ARG((reinterpret_cast<size_t>(scratch) & 15) == 0, "y3_conv_wgrad_wino: scratch must be 16-byte aligned");
    const long long M = (long long)d->n * d->h * d->w;
    Y3_CHECK_ARG(M * d->cin < (1LL << 29) && M * dz_stride < (1LL << 29),
                 "y3_conv_wgrad_wino: tensor exceeds 2^29 elements (32-bit byte offsets)");
    WgWinoArgs a;
    a.x = x; a.dz = dz;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.CoP = dz_stride;
    a.TH = (d->h + 1) / 2; a.TW = (d->w + 1) / 2; a.T = d->n * a.TH * a.TW;
    a.adv_r = 8 / a.TW; a.adv_c = 8 % a.TW;
    Y3_CHECK_ARG(a.T < (1 << 24), "y3_conv_wgrad_wino: too many tiles");
    wgw_split(d, &a.nsplit, &a.ksteps);
    a.out = a.nsplit == 1 ? dw_hwio : static_cast<float*>(scratch);
    constexpr size_t lds = (size_t)4 * STAGE;
    static bool attr_set[Y3_MAX_DEVICES] = {};   // benign race (idempotent)
    const int dev_ = y3_current_device();
    if (dev_ < 0 || !attr_set[dev_]) {
        Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_wino_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev_ >= 0) attr_set[dev_] = true;
    }
    const int wt = (d->cin / 64) * (d->cout / 64);
    hipLaunchKernelGGL(conv_wgrad_wino_kernel, dim3(wt, a.nsplit), dim3(256), lds, stream, a);
    Y3_CHECK_HIP(hipGetLastError());
    if (a.nsplit > 1) {
        const long long n4 = (long long)9 * a.Cin * a.Cout / 4;
        long long nb = (n4 + 63) / 64;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(wgw_sum_splits_kernel, dim3((int)nb), dim3(256), 0, stream, static_cast<float*>(scratch),
                           a.nsplit, n4, dw_hwio);
        Y3_CHECK_HIP(hipGetLastError());
    }
    return Y3_OK;
}

extern "C" int y3_conv_wgrad_wino_eligible(const y3_conv_desc* d) { return y3_conv_wgrad_wino_eligible_impl(d); }
extern "C" size_t y3_conv_wgrad_wino_scratch_bytes(const y3_conv_desc* d) { return y3_conv_wgrad_wino_scratch_bytes_impl(d); }
extern "C" int y3_conv_wgrad_wino(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* dz, int dz_stride,
                                  float* dw_hwio, void* scratch, size_t scratch_bytes) {
    Y3_CHECK_ARG(ctx, "y3_conv_wgrad_wino: null context");
    return y3_launch_conv_wgrad_wino(ctx->stream, d, x, dz, dz_stride, dw_hwio, scratch, scratch_bytes);
}
