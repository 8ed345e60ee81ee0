// Micro-benchmark: what does one more VALU / LDS / VMEM instruction cost a wave that is otherwise streaming independent
// 32x32x2 fp32 MFMAs (64 matrix-pipe cycles each), with one and with two waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// Prints cycles per MFMA per SIMD (at 2.4 GHz, from the launch time); 64 = the pipe is the only limit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int NG, int NT>
__global__ void __launch_bounds__(NT) k(float* out, const float* in, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a, b};
    const f32x2 c = {1e-3f, 2e-3f};
    f32x4 l = {0.f, 0.f, 0.f, 0.f};
    f32x2 g = {0.f, 0.f};
    const f32x4* lp = reinterpret_cast<const f32x4*>(smem) + threadIdx.x;
    const f32x2* gp = reinterpret_cast<const f32x2*>(in) + (blockIdx.x * NT + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[(u + q) & 7]) : "v"(c));
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                f32x4 t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)(lp + 64 * q)));
                if (u == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); l += t; }
            }
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                f32x2 t = __builtin_nontemporal_load(gp + ((it * 8 + u) & 1023) * 64);
                if (u == 7) g += t;
            }
        }
    }
    float s = l[0] + g[0];
    for (int i = 0; i < 8; ++i) { s += v[i][0] + v[i][1]; for (int r = 0; r < 16; ++r) s += acc[i][r]; }
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int NV, int NL, int NG, int NT>
void run(float* d, const float* in) {
    auto kern = k<NV, NL, NG, NT>;
    const int lds = 100 * 1024;      // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, 0, d, in, 50, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, 0, d, in, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * (NT / 256);
    printf("waves/SIMD=%d  +%d v_pk_add +%d ds_read_b128 +%d load_b64 per MFMA: %.3f ms, %.1f cycles per MFMA per SIMD (2.4 GHz)\n",
           NT / 256, NV, NL, NG, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
    float *d, *in;
    hipMalloc(&d, 256 * 512 * 4);
    hipMalloc(&in, (size_t)256 * 512 * 8 + 1024 * 64 * 8 + 4096);
    hipMemset(in, 0, (size_t)256 * 512 * 8 + 1024 * 64 * 8 + 4096);
#define BOTH(NV, NL, NG) run<NV, NL, NG, 256>(d, in); run<NV, NL, NG, 512>(d, in);
    BOTH(0, 0, 0) BOTH(1, 0, 0) BOTH(2, 0, 0) BOTH(4, 0, 0) BOTH(8, 0, 0) BOTH(16, 0, 0)
    BOTH(0, 1, 0) BOTH(0, 2, 0) BOTH(0, 0, 1) BOTH(0, 0, 2) BOTH(2, 1, 1) BOTH(4, 1, 1)
    return 0;
}
