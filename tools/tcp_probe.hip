// Micro-benchmark: vector-memory (TA/TCP) throughput of one CU for the Winograd kernel's activation access pattern:
// a wave-level 8-byte load whose 64 lanes touch NL distinct 128-byte lines (4 lanes x 8 B = 32 B per line for NL = 16)
// against a fully contiguous one (512 B = 4 lines), all L1/L2 hits (small footprint), 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/tcp_probe.hip -o /tmp/tcp_probe && /tmp/tcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// STRIDE = bytes between the 32-byte groups of consecutive 4-lane quads (32: contiguous; 512: one line per quad)
// FOOT = footprint in bytes each workgroup cycles through (L1 = 32 KB)
template <int WIDTH>
__global__ void __launch_bounds__(512) k(float* out, const char* in, int iters, int stride, int foot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = in + (size_t)blockIdx.x * (1 << 20);
    const int quad = lane >> 2, sub = lane & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned off = (unsigned)(quad * stride + sub * WIDTH + wave * 16 * stride);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned o = (off + (unsigned)u * 128u * (unsigned)stride) & (unsigned)(foot - 1);
            if (WIDTH == 8) {
                f32x2 t = *reinterpret_cast<const f32x2*>(base + o);
                acc[0] += t[0]; acc[1] += t[1];
            } else {
                f32x4 t = *reinterpret_cast<const f32x4*>(base + o);
                acc += t;
            }
        }
        off += 32;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int WIDTH>
void run(float* d, const char* in, int stride, int foot) {
    const int blocks = 256, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WIDTH>, dim3(blocks), dim3(512), 0, 0, d, in, 20, stride, foot);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<WIDTH>, dim3(blocks), dim3(512), 0, 0, d, in, iters, stride, foot);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)iters * 16 * 8;
    printf("%2d-byte loads, quad stride %4d B (%2d lines per wave load), footprint %4d KB: %.1f cycles per wave load per CU (2.4 GHz)\n",
           WIDTH, stride, stride >= 128 ? 16 : (16 * stride + 127) / 128, foot >> 10, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
    float* d; char* in;
    (void)hipMalloc(&d, 256 * 512 * 4);
    (void)hipMalloc(&in, (size_t)257 << 20);
    (void)hipMemset(in, 0, (size_t)257 << 20);
    for (int foot : {16 << 10, 64 << 10, 512 << 10}) {
        run<8>(d, in, 32, foot);      // contiguous: 4 lines per wave load
        run<8>(d, in, 128, foot);     // 16 lines, 32 B used of each
        run<8>(d, in, 512, foot);     // 16 lines, pixel stride 512 B (Cin = 128)
        run<8>(d, in, 2048, foot);    // pixel stride 2 KB (Cin = 512)
        run<16>(d, in, 64, foot);     // 16-byte loads, contiguous 1 KB: 8 lines
    }
    return 0;
}
