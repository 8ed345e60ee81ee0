// Calibration micro-benchmark: attainable v_mfma_f32_32x32x16_bf16 rate on this chip under sustained load with
// random (non-zero) operand bits — the DVFS-limited ceiling of the split-bf16 conv path.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o /tmp/mfma_peak_bf16 && /tmp/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    u32x4 ua, ub;
    for (int q = 0; q < 4; ++q) {
        h = h * 1664525u + 1013904223u; ua[q] = (h & 0x007f007fu) | 0x3f003f00u | ((h >> 8) & 0x80008000u);
        h = h * 1664525u + 1013904223u; ub[q] = (h & 0x007f007fu) | 0x3b003b00u | ((h >> 8) & 0x80008000u);
    }
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wpb = 1; wpb <= 2; ++wpb) {
        int blocks = 256 * wpb, iters = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 7u + rep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
            printf("blocks/CU=%d: %.2f ms, %.1f TFLOP/s bf16 (= %.1f fp32-equivalent at 6 products)\n", wpb, ms,
                   fl / ms / 1e9, fl / ms / 1e9 / 6);
        }
    }
    return 0;
}
