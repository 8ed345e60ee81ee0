// Where do the first resident workgroups of a 2-per-CU launch land?  Records HW_ID / XCC_ID and the start time of every workgroup
// of a grid shaped like conv_wino44v_f32_kernel's (256 threads, 72 KB of LDS), each spinning ~20 us.
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, unsigned long long* t) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        smem[0] = (unsigned char)hw;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(32);
}
int main() {
    const int nb = 1376;
    unsigned* out; unsigned long long* t;
    hipMalloc(&out, nb * 8); hipMalloc(&t, nb * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 73728, 0, out, t);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(nb * 2); std::vector<unsigned long long> ht(nb);
    hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, nb * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ht[0];
    for (int i = 0; i < nb; ++i) if (ht[i] < t0) t0 = ht[i];
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < 600; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3, wv = hw & 15;
        if (b < 80 || (b >= 256 && b < 300) || b >= 500)
            printf("block %4d  xcc %u se %u sh %u cu %2u simd %u wave %u  t=%6.2f us\n", b, xcc, se, sh, cu, simd, wv, (ht[b] - t0) / 100.0);
        if (b < 512) by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    int n = 0;
    for (auto& kv : by_cu) {
        if (n++ < 40) { printf("cu %06x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    }
    printf("distinct CUs among the first 512 workgroups: %zu\n", by_cu.size());
    return 0;
}
