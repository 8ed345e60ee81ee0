#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// each lane loads 16 bytes from src[(lane ^ 1) * 16 ...] (per-lane global address) into LDS at base + lane*16.
// lanes with (lane % 5 == 0) use an out-of-range offset: what lands in LDS?
__global__ void probe(const unsigned* __restrict__ src, unsigned nbytes, unsigned* __restrict__ out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 64 * 4 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += blockDim.x) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    unsigned* dst = lds + wave * 64 * 4;            // wave-uniform base
    if (mode == 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
        unsigned voff = (unsigned)((lane ^ 1) * 16);
        if (lane % 5 == 0) voff = 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    } else {
        const unsigned* g = src + (lane ^ 1) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 64 * 4; i += blockDim.x) out[i] = lds[i];
}
int main() {
    const int n = 64 * 4 * 2;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d, (unsigned)(64 * 16), o, mode);
        std::vector<unsigned> r(n);
        hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
        int ok = 0, zero = 0, dead = 0, other = 0;
        for (int w = 0; w < 2; ++w)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    unsigned v = r[w * 256 + lane * 4 + j], want = 1000 + (lane ^ 1) * 4 + j;
                    if (mode == 0 && lane % 5 == 0) { if (v == 0) ++zero; else if (v == 0xDEADBEEFu) ++dead; else ++other; }
                    else { if (v == want) ++ok; else ++other; }
                }
        printf("mode %d (%s): in-range dwords correct %d, OOB lanes: zero %d untouched %d other %d; wrong %d\n", mode,
               mode == 0 ? "raw_buffer_load_lds b128" : "global_load_lds b128", ok, zero, dead, mode == 0 ? other : 0, mode == 1 ? other : 0);
    }
    return 0;
}
