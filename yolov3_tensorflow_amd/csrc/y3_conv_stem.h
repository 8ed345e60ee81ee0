// Stem conv (3x3, stride 1, Cin = 3 -> 32; reference model.py:42 via utils/layer_utils.py:9-22: conv + folded BN + LeakyReLU)
// on the fp32 matrix pipe, shared by the fp32 path (fp32 out) and the bf16-storage path (bf16 out, same fp32 arithmetic).
//
// K = 27 is far too short for the implicit-GEMM tiles of y3_conv.hip, and the thread-per-pixel form this replaces was
// bound by the LDS: 216 broadcast 16-byte weight reads per pixel, 0.27 ms for a 16 x 608 x 608 batch against 0.08 ms of
// HBM time.  Here the weights never leave registers:
//
//     C^T [32 channels x 32 pixels] = W^T [32 x 28] * P^T [28 x 32]          (k = tap*3 + ci; k = 27 is a zero row)
//
// as 14 v_mfma_f32_32x32x2_f32 per 32-pixel tile.  A wave keeps W^T as its 14 A operands for its whole life; a B operand is
// one 4-byte load per lane - lane l fetches element k = 2j + l/32 of the patch of pixel l%32, bounds-checked (padding reads
// as 0) - and the accumulator layout leaves every lane with four runs of 4 consecutive channels of ONE pixel: scale, shift,
// LeakyReLU and 16-byte (fp32) or 8-byte (bf16) stores straight from registers.
#pragma once
#include "y3_internal.h"

namespace y3stem {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_bits(float f) {      // round to nearest even (finite inputs)
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

constexpr int TILES_PER_STEP = 2;     // 32-pixel tiles a wave has in flight (28 loads outstanding per lane)

template <bool OUT_BF16>
__global__ void __launch_bounds__(256) conv_stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             void* __restrict__ y, int H, int W, int M, int act) {
    __shared__ __attribute__((aligned(16))) float ssc[32], ssh[32];
    if (threadIdx.x < 32) { ssc[threadIdx.x] = scale[threadIdx.x]; ssh[threadIdx.x] = shift[threadIdx.x]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    // A operands: W^T[ch = col][k = 2j + half] from the HWIO kernel [27][32]
    float a[14];
    int dyx[14];                          // per B operand: (dy + 1) | (dx + 1) << 8 | live << 16, and the element offset below
    int off[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int k = 2 * j + half;
        const bool live = k < 27;
        a[j] = live ? w[k * 32 + col] : 0.f;
        const int tap = live ? k / 3 : 0, ci = live ? k - tap * 3 : 0;
        const int ky = tap / 3, kx = tap - ky * 3;
        dyx[j] = ky | (kx << 8) | ((live ? 1 : 0) << 16);
        off[j] = ((ky - 1) * W + (kx - 1)) * 3 + ci;
    }
    const int ntiles = (M + 31) >> 5;
    const int nsteps = (ntiles + TILES_PER_STEP - 1) / TILES_PER_STEP;
    const int HW = H * W;
    for (int step = blockIdx.x * 4 + wave; step < nsteps; step += gridDim.x * 4) {
        float b[TILES_PER_STEP][14];
        int mpix[TILES_PER_STEP];
#pragma unroll
        for (int t = 0; t < TILES_PER_STEP; ++t) {
            const int m_raw = (step * TILES_PER_STEP + t) * 32 + col;
            const int m = m_raw < M ? m_raw : M - 1;       // lanes past the end recompute the last pixel, store nothing
            mpix[t] = m_raw;
            const int n = m / HW;
            const int rem = m - n * HW;
            const int oy = rem / W, ox = rem - oy * W;
            const float* centre = x + (size_t)m * 3;
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                const int iy = oy - 1 + (dyx[j] & 0xff), ix = ox - 1 + ((dyx[j] >> 8) & 0xff);
                const bool ok = (dyx[j] >> 16) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const float v = centre[ok ? off[j] : 0];
                b[t][j] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < TILES_PER_STEP; ++t) {
            f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 14; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[t][j], c, 0, 0, 0);
            // c[4g + q] = channel 8g + 4*half + q of pixel `col`
            if (mpix[t] < M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 8 * g + 4 * half;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(ssc + ch), sh = *reinterpret_cast<const f32x4*>(ssh + ch);
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r = c[4 * g + q] * sc[q] + sh[q];
                        if (act) r = r > 0.f ? r : 0.1f * r;
                        v[q] = r;
                    }
                    if (OUT_BF16) {
                        const u32x2 pk = {bf16_bits(v[0]) | (bf16_bits(v[1]) << 16), bf16_bits(v[2]) | (bf16_bits(v[3]) << 16)};
                        *reinterpret_cast<u32x2*>(static_cast<unsigned short*>(y) + (size_t)mpix[t] * 32 + ch) = pk;
                    } else {
                        *reinterpret_cast<f32x4*>(static_cast<float*>(y) + (size_t)mpix[t] * 32 + ch) = v;
                    }
                }
            }
        }
    }
}

// x: fp32 [N,H,W,3]; w: HWIO fp32 [27][32]; y: [N,H,W,32] fp32 or bf16
template <bool OUT_BF16>
inline int launch_stem(hipStream_t stream, const float* x, const float* w, const float* scale, const float* shift, void* y,
                       int N, int H, int W, int act) {
    const long long M = (long long)N * H * W;
    const long long steps = ((M + 31) / 32 + TILES_PER_STEP - 1) / TILES_PER_STEP;
    // enough workgroups for ~8 waves per SIMD on the 256 CUs; the waves stride over the steps
    const long long want = (steps + 3) / 4;
    const int blocks = (int)(want < 256 * 8 ? want : 256 * 8);
    auto kern = conv_stem_mfma_kernel<OUT_BF16>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, stream, x, w, scale, shift, y, H, W, (int)M, act);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}

}  // namespace y3stem
