// Per-class non-maximum suppression for gfx950 — both definitions the reference carries:
//   Y3_NMS_TF : utils/nms_utils.py:8-48 `gpu_nms` (80x {boolean_mask, tf.image.non_max_suppression,
//               gather}); IoU without +1, degenerate boxes give IoU 0, suppress when IoU > thresh.
//   Y3_NMS_PY : utils/nms_utils.py:51-123 `py_nms`/`cpu_nms`; +1 on the intersection w/h only,
//               a candidate survives only while ovr <= thresh (NaN does not survive).
//
// Greedy NMS == visit the candidates of a class in the order (score descending, box index ascending) and keep one iff no
// kept box suppresses it.  The order is a 64-bit key {orderable score bits, ~box index}, which makes the tie-break
// deterministic.  Two forms, one workgroup per (image, class) each:
//   * nms_select_sorted_kernel (every class with up to RCAP = 16,384 candidates): keys sorted in the LDS, candidates
//     visited in chunks - parallel test against the kept boxes, in-order resolution inside the chunk, early exit at
//     max_boxes (round 4; it replaced an arg-max sweep per selected box whose heaviest class took 2.7 ms alone);
//   * nms_select_kernel (more than RCAP candidates in one class): no sort - repeatedly take the best live candidate
//     (wave-shuffle arg-max on the keys) and kill what it overlaps, candidates re-derived from global memory.
// All IoU arithmetic is plain fp32 in the reference's operation order (compiled with -ffp-contract=off) so selections are
// bit-exact against the CPU oracle.
//
// Pipeline: collect (threshold + per-class compaction, LDS-aggregated atomics) -> select -> gather.
#include "y3_internal.h"

namespace {

// classes with up to RCAP candidates run on the sorted forms (nms_select_sorted_kernel: their keys live in the LDS); the
// arg-max form below keeps only what is left: more than RCAP candidates in ONE class (a dense score map at 608x608)
constexpr int RCAP = 16384;

struct NmsWs {
    int32_t* cand_count;  // [n*C]
    int32_t* cand_idx;    // [n*C][B]
    int32_t* sel_count;   // [n*C]
    int32_t* sel_idx;     // [n*C][max_boxes]
    int rcap;             // classes with up to this many candidates run on the sorted forms (0: max_boxes too large for them)
};

__global__ void __launch_bounds__(256) nms_collect_kernel(const float* __restrict__ scores, int B, int C,
                                                          float thresh, int boxes_per_block,
                                                          int32_t* cand_count, int32_t* cand_idx) {
    extern __shared__ int lds_i[];
    int* lcount = lds_i;      // [C]
    int* lbase = lds_i + C;   // [C]
    const int n = blockIdx.y;
    const int b0 = blockIdx.x * boxes_per_block;
    const int nb = min(boxes_per_block, B - b0);
    const float* tile = scores + ((size_t)n * B + b0) * C;
    const int total = nb * C;
    for (int c = threadIdx.x; c < C; c += 256) lcount[c] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 256)
        if (tile[e] >= thresh) atomicAdd(&lcount[e % C], 1);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int k = lcount[c];
        lbase[c] = k ? atomicAdd(&cand_count[n * C + c], k) : 0;
        lcount[c] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 256) {
        if (tile[e] >= thresh) {
            const int c = e % C;
            const int slot = lbase[c] + atomicAdd(&lcount[c], 1);
            cand_idx[((size_t)n * C + c) * B + slot] = b0 + e / C;
        }
    }
}

struct Cand {
    float c0, c1, c2, c3;  // TF mode: normalised (min0,min1,max0,max1); PY mode: x1,y1,x2,y2
    float area;
    float score;
    int idx;               // box index; < 0 once dead (stored as ~idx)
};

template <int MODE>
__device__ __forceinline__ void make_cand(const float* boxes_n, const float* scores_n, int C, int c,
                                          int idx, Cand& k) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(boxes_n + (size_t)idx * 4);
    if (MODE == Y3_NMS_TF) {
        // tf.image.non_max_suppression: coordinates are normalised per box before use
        k.c0 = fminf(b[0], b[2]); k.c1 = fminf(b[1], b[3]);
        k.c2 = fmaxf(b[0], b[2]); k.c3 = fmaxf(b[1], b[3]);
    } else {
        k.c0 = b[0]; k.c1 = b[1]; k.c2 = b[2]; k.c3 = b[3];
    }
    k.area = (k.c2 - k.c0) * (k.c3 - k.c1);   // utils/nms_utils.py:68 (no +1) / TF area
    k.score = scores_n[(size_t)idx * C + c];
    k.idx = idx;
}

// true -> candidate j must be removed after selecting i
template <int MODE>
__device__ __forceinline__ bool suppressed(const Cand& i, const Cand& j, float thr) {
    if (MODE == Y3_NMS_TF) {
        if (i.area <= 0.f || j.area <= 0.f) return false;  // IoU defined as 0
        const float h = fmaxf(fminf(i.c2, j.c2) - fmaxf(i.c0, j.c0), 0.f);
        const float w = fmaxf(fminf(i.c3, j.c3) - fmaxf(i.c1, j.c1), 0.f);
        const float inter = h * w;
        const float iou = inter / (i.area + j.area - inter);
        return iou > thr;
    } else {
        // utils/nms_utils.py:75-85
        const float xx1 = fmaxf(i.c0, j.c0), yy1 = fmaxf(i.c1, j.c1);
        const float xx2 = fminf(i.c2, j.c2), yy2 = fminf(i.c3, j.c3);
        const float w = fmaxf(0.f, xx2 - xx1 + 1.f);
        const float h = fmaxf(0.f, yy2 - yy1 + 1.f);
        const float inter = w * h;
        const float ovr = inter / (i.area + j.area - inter);
        return !(ovr <= thr);
    }
}

__device__ __forceinline__ unsigned long long make_key(float score, int idx) {
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // total order on floats
    return ((unsigned long long)u << 32) | (unsigned)(~idx);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Block-wide arg-max over NW waves.  Each thread passes its local best (key, position); returns the winning position
// (or -1 if none) to every thread.
template <int NW = 4>
__device__ __forceinline__ int block_argmax(unsigned long long key, int pos, unsigned long long* skey,
                                            int* spos) {
    const unsigned long long wmax = wave_max_u64(key);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the lane holding the wave maximum publishes its position (keys are unique per candidate)
    if (key == wmax && key != 0ull) spos[wave] = pos;
    if (lane == 0) {
        skey[wave] = wmax;
        if (wmax == 0ull) spos[wave] = -1;
    }
    __syncthreads();
    unsigned long long best = skey[0];
    int bpos = spos[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if (skey[w] > best) { best = skey[w]; bpos = spos[w]; }
    __syncthreads();
    return bpos;
}

template <int MODE>
__global__ void __launch_bounds__(256) nms_select_kernel(const float* __restrict__ boxes,
                                                         const float* __restrict__ scores, int B, int C,
                                                         int max_boxes, float iou_thr, NmsWs ws) {
    __shared__ unsigned long long skey[4];
    __shared__ int spos[4];
    __shared__ Cand sel;

    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const int K = ws.cand_count[blockIdx.x];
    int32_t* cidx = ws.cand_idx + (size_t)blockIdx.x * B;
    int32_t* sidx = ws.sel_idx + (size_t)blockIdx.x * max_boxes;
    if (K == 0) {
        if (threadIdx.x == 0) ws.sel_count[blockIdx.x] = 0;
        return;
    }
    if (K <= ws.rcap) return;                   // nms_select_sorted_kernel's share
    const float* boxes_n = boxes + (size_t)n * B * 4;
    const float* scores_n = scores + (size_t)n * B * C;
    constexpr bool in_lds = false;              // (candidates are re-derived from global memory on every sweep)
    Cand* lcand = nullptr;

    // ---- load candidates and find the first best ---------------------------------------------------
    unsigned long long lkey = 0ull;
    int lpos = -1;
    for (int j = threadIdx.x; j < K; j += 256) {
        Cand k;
        make_cand<MODE>(boxes_n, scores_n, C, c, cidx[j], k);
        if (in_lds) lcand[j] = k;
        const unsigned long long key = make_key(k.score, k.idx);
        if (key > lkey) { lkey = key; lpos = j; }
    }
    int best = block_argmax(lkey, lpos, skey, spos);

    int nsel = 0;
    while (best >= 0 && nsel < max_boxes) {
        if (threadIdx.x == 0) {
            Cand k;
            if (in_lds) k = lcand[best];
            else make_cand<MODE>(boxes_n, scores_n, C, c, cidx[best], k);
            sel = k;
            sidx[nsel] = k.idx;
            // retire the selected candidate
            if (in_lds) lcand[best].idx = ~k.idx;
            else cidx[best] = ~k.idx;
        }
        ++nsel;
        __syncthreads();
        const Cand s = sel;
        lkey = 0ull;
        lpos = -1;
        if (nsel < max_boxes) {
            for (int j = threadIdx.x; j < K; j += 256) {
                Cand k;
                if (in_lds) {
                    k = lcand[j];
                    if (k.idx < 0) continue;
                } else {
                    const int id = cidx[j];
                    if (id < 0) continue;
                    make_cand<MODE>(boxes_n, scores_n, C, c, id, k);
                }
                if (suppressed<MODE>(s, k, iou_thr)) {
                    if (in_lds) lcand[j].idx = ~k.idx;
                    else cidx[j] = ~k.idx;
                } else {
                    const unsigned long long key = make_key(k.score, k.idx);
                    if (key > lkey) { lkey = key; lpos = j; }
                }
            }
        }
        best = block_argmax(lkey, lpos, skey, spos);
    }
    if (threadIdx.x == 0) ws.sel_count[blockIdx.x] = nsel;
}

// Greedy selection on SORTED candidates, for every class with 1 .. RCAP candidates.  The arg-max forms pay one
// workgroup-wide reduction (two barriers) and one sweep over ALL live candidates per SELECTED box - 190 dependent sweeps
// per class in the bench's detector regime at eval.py's parameters, 400 on its heaviest classes (5-10k candidates:
// 2.7 ms for ONE workgroup, the critical path of the whole batch, profiles/r04_postproc.txt).  Here:
//   1. keys {orderable score bits, ~box index} of the K candidates are sorted descending in the LDS (bitonic); greedy NMS
//      visits candidates in exactly that order - same keys, same tie-break as the arg-max form;
//   2. chunks of NT consecutive candidates, one per thread: (a) every thread tests its candidate against the boxes
//      selected so far (LDS broadcast reads, all threads in parallel); (b) the survivors are resolved in order: wave by
//      wave, and inside a wave the lowest live lane is selected (v_readlane), the lanes behind it that it overlaps die
//      (one ballot per selection); the later waves test their survivors against what the wave before them selected;
//   3. the loop ends as soon as max_boxes are selected - a heavy class never looks at its tail.
// Three instantiations by the class's candidate count, so that the LDS a workgroup holds (its keys) matches what it needs and
// several classes share a CU: ONE wave up to SK1 candidates (4 KB of keys), four waves up to SK4 (32 KB; the bench's
// detector regime at eval.py's threshold has 1,064 of its 2,560 classes here, 2,330 candidates at most), sixteen waves up
// to RCAP (128 KB: one class per CU).  Same arithmetic (suppressed<MODE>, make_cand<MODE>) on the same pairs, same order of
// decisions: the selections are bit-identical to the arg-max kernels' (tests/test_nms_gpu.py against the C oracle).
#ifndef Y3_NMS_SK1
#define Y3_NMS_SK1 256
#endif
constexpr int SK1 = Y3_NMS_SK1, SK4 = 4096;

template <int MODE, int NT>
__global__ void __launch_bounds__(NT) nms_select_sorted_kernel(const float* __restrict__ boxes,
                                                               const float* __restrict__ scores, int B, int C,
                                                               int max_boxes, float iou_thr, NmsWs ws) {
    constexpr int NW = NT / 64;
    constexpr int KMAX = NW == 1 ? SK1 : NW == 4 ? SK4 : RCAP;      // keys the LDS holds (a power of two >= the largest K of this form)
    constexpr int KLO = NW == 1 ? 0 : NW == 4 ? SK1 : SK4;          // this form's classes: KLO < K <= KMAX
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_lds[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(nms_lds);                 // [KMAX]
    f32x4* sbox = reinterpret_cast<f32x4*>(nms_lds + (size_t)KMAX * 8);                         // [max_boxes] selected corners
    float* sarea = reinterpret_cast<float*>(nms_lds + (size_t)KMAX * 8 + (size_t)max_boxes * 16);   // [max_boxes]
    int* s_nsel = reinterpret_cast<int*>(nms_lds + (size_t)KMAX * 8 + (size_t)max_boxes * 20);       // [2]
    const int K = ws.cand_count[blockIdx.x];
    if (K <= KLO || K > KMAX) return;           // (K == 0: nms_select_kernel wrote the zero count)
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const int32_t* cidx = ws.cand_idx + (size_t)blockIdx.x * B;
    int32_t* sidx = ws.sel_idx + (size_t)blockIdx.x * max_boxes;
    const float* boxes_n = boxes + (size_t)n * B * 4;
    const float* scores_n = scores + (size_t)n * B * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int KP = 64;
    while (KP < K) KP <<= 1;
    for (int j = tid; j < KP; j += NT) {
        unsigned long long key = 0ull;          // padding sorts last (a real key is never 0: its low half is ~idx, idx >= 0)
        if (j < K) {
            const int idx = cidx[j];
            key = make_key(scores_n[(size_t)idx * C + c], idx);
        }
        keys[j] = key;
    }
    __syncthreads();
    for (int size = 2; size <= KP; size <<= 1) {             // bitonic sort, descending
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (KP >> 1); t += NT) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const int hi = lo | stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }

    int nsel = 0;                               // uniform over the workgroup
    auto test_against = [&](const Cand& k, bool& alive, int s0, int s1) {
#pragma unroll 4
        for (int s = s0; s < s1; ++s) {
            Cand sc;
            const f32x4 sb = sbox[s];
            sc.c0 = sb[0]; sc.c1 = sb[1]; sc.c2 = sb[2]; sc.c3 = sb[3];
            sc.area = sarea[s];
            if (alive && suppressed<MODE>(sc, k, iou_thr)) alive = false;
        }
    };
    for (int base = 0; base < K && nsel < max_boxes; base += NT) {
        const int j = base + tid;
        bool alive = j < K;
        Cand k;
        k.c0 = k.c1 = k.c2 = k.c3 = k.area = k.score = 0.f;
        k.idx = 0;
        if (alive) {
            const int idx = (int)~(unsigned)(keys[j] & 0xffffffffull);
            make_cand<MODE>(boxes_n, scores_n, C, c, idx, k);
        }
        test_against(k, alive, 0, nsel);        // (a) against the boxes selected from earlier chunks
        for (int w = 0; w < NW && nsel < max_boxes; ++w) {       // (b) in order: wave by wave, lane by lane
            if (base + w * 64 >= K) break;
            int mine = nsel;
            if (wave == w) {
                unsigned long long mask = __ballot(alive);
                while (mask != 0ull && mine < max_boxes) {
                    const int l = __builtin_ctzll(mask);
                    Cand sc;
                    sc.c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k.c0), l));
                    sc.c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k.c1), l));
                    sc.c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k.c2), l));
                    sc.c3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k.c3), l));
                    sc.area = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k.area), l));
                    if (lane == l) {
                        sbox[mine] = f32x4{k.c0, k.c1, k.c2, k.c3};
                        sarea[mine] = k.area;
                        sidx[mine] = k.idx;
                        alive = false;
                    }
                    ++mine;
                    if (alive && lane > l && suppressed<MODE>(sc, k, iou_thr)) alive = false;
                    mask = __ballot(alive);
                }
                if (NW > 1 && lane == 0) s_nsel[w & 1] = mine;
            }
            if (NW > 1) {
                // (the slot of step w is rewritten at step w + 2, which no wave reaches before every wave has passed the
                //  barrier of step w + 1, i.e. after it has read this one)
                __syncthreads();
                const int upto = s_nsel[w & 1];
                if (wave > w) test_against(k, alive, nsel, upto);
                nsel = upto;
            } else {
                nsel = mine;
            }
        }
        __syncthreads();      // this chunk's selections are in the LDS before the next chunk reads them
    }
    if (tid == 0) ws.sel_count[blockIdx.x] = nsel;
}

__global__ void __launch_bounds__(256) nms_gather_kernel(const float* __restrict__ boxes,
                                                         const float* __restrict__ scores, int B, int C,
                                                         int max_boxes, NmsWs ws, float* out_boxes,
                                                         float* out_scores, int32_t* out_labels,
                                                         int32_t* out_index, int32_t* out_counts) {
    extern __shared__ int lds_i[];
    int* off = lds_i;  // [C+1] exclusive prefix of the per-class selection counts
    const int n = blockIdx.x;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) {
            off[c] = acc;
            acc += ws.sel_count[n * C + c];
        }
        off[C] = acc;
        out_counts[n] = acc;
    }
    __syncthreads();
    const size_t cap = (size_t)C * max_boxes;
    for (int e = threadIdx.x; e < C * max_boxes; e += 256) {
        const int c = e / max_boxes, k = e - c * max_boxes;
        if (k < off[c + 1] - off[c]) {
            const int idx = ws.sel_idx[((size_t)n * C + c) * max_boxes + k];
            const size_t o = (size_t)n * cap + off[c] + k;
            *reinterpret_cast<f32x4*>(out_boxes + o * 4) =
                *reinterpret_cast<const f32x4*>(boxes + ((size_t)n * B + idx) * 4);
            out_scores[o] = scores[((size_t)n * B + idx) * C + c];
            out_labels[o] = c;
            if (out_index) out_index[o] = idx;
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t y3_nms_workspace_bytes(int n, int num_boxes, int class_num, int max_boxes) {
    if (n <= 0 || num_boxes <= 0 || class_num <= 0 || max_boxes <= 0) return 0;
    const size_t nc = (size_t)n * class_num;
    return align256(nc * 4) * 2 + align256(nc * num_boxes * 4) + align256(nc * max_boxes * 4);
}

extern "C" int y3_nms(y3_ctx* ctx, int mode, const float* boxes, const float* scores, int n, int num_boxes,
                      int class_num, int max_boxes, float score_thresh, float iou_thresh, void* workspace,
                      size_t workspace_bytes, float* out_boxes, float* out_scores, int32_t* out_labels,
                      int32_t* out_index, int32_t* out_counts) {
    Y3_CHECK_ARG(ctx && boxes && scores && workspace && out_boxes && out_scores && out_labels && out_counts,
                 "y3_nms: null argument");
    Y3_CHECK_ARG(mode == Y3_NMS_TF || mode == Y3_NMS_PY, "y3_nms: unknown mode %d", mode);
    Y3_CHECK_ARG(n > 0 && num_boxes > 0 && class_num > 0 && max_boxes > 0, "y3_nms: non-positive dimension");
    Y3_CHECK_ARG(workspace_bytes >= y3_nms_workspace_bytes(n, num_boxes, class_num, max_boxes),
                 "y3_nms: workspace too small (%zu < %zu)", workspace_bytes,
                 y3_nms_workspace_bytes(n, num_boxes, class_num, max_boxes));
    Y3_CHECK_ARG(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)out_boxes & 15) == 0,
                 "y3_nms: boxes/out_boxes must be 16-byte aligned");
    const size_t nc = (size_t)n * class_num;
    char* p = static_cast<char*>(workspace);
    NmsWs ws;
    ws.cand_count = reinterpret_cast<int32_t*>(p); p += align256(nc * 4);
    ws.sel_count = reinterpret_cast<int32_t*>(p);  p += align256(nc * 4);
    ws.cand_idx = reinterpret_cast<int32_t*>(p);   p += align256(nc * num_boxes * 4);
    ws.sel_idx = reinterpret_cast<int32_t*>(p);
    // the sorted forms keep a class's keys AND its selected boxes in the LDS: beyond ~1,600 boxes per class the arg-max form
    // takes every class
    const bool sorted_ok = (size_t)16384 * 8 + (size_t)max_boxes * 20 + 16 <= (size_t)160 * 1024;
    ws.rcap = sorted_ok ? RCAP : 0;
    hipStream_t st = ctx->stream;
    Y3_CHECK_HIP(hipMemsetAsync(ws.cand_count, 0, nc * 4, st));
    const int bpb = 64;  // boxes per collect block
    hipLaunchKernelGGL(nms_collect_kernel, dim3((num_boxes + bpb - 1) / bpb, n), dim3(256),
                       2 * class_num * sizeof(int), st, scores, num_boxes, class_num, score_thresh, bpb,
                       ws.cand_count, ws.cand_idx);
    Y3_CHECK_HIP(hipGetLastError());
    if (mode == Y3_NMS_TF)
        hipLaunchKernelGGL(nms_select_kernel<Y3_NMS_TF>, dim3((unsigned)nc), dim3(256), 0, st, boxes, scores,
                           num_boxes, class_num, max_boxes, iou_thresh, ws);
    else
        hipLaunchKernelGGL(nms_select_kernel<Y3_NMS_PY>, dim3((unsigned)nc), dim3(256), 0, st, boxes, scores,
                           num_boxes, class_num, max_boxes, iou_thresh, ws);
    Y3_CHECK_HIP(hipGetLastError());
    if (sorted_ok) {
        // every class with 1 .. RCAP candidates on the sorted forms: one wave up to SK1 candidates, sixteen waves above (each
        // kernel returns at once on the others' classes: the counts live on the device)
        const size_t sel_bytes = (size_t)max_boxes * 20 + 16;
        auto launch_tier = [&](auto kern, int threads, size_t keys) -> int {
            const size_t lds = keys * 8 + sel_bytes;
            Y3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)nc), dim3(threads), lds, st, boxes, scores, num_boxes, class_num, max_boxes,
                               iou_thresh, ws);
            Y3_CHECK_HIP(hipGetLastError());
            return Y3_OK;
        };
        const bool tf = mode == Y3_NMS_TF;
        if (int rc = tf ? launch_tier(nms_select_sorted_kernel<Y3_NMS_TF, 64>, 64, SK1)
                        : launch_tier(nms_select_sorted_kernel<Y3_NMS_PY, 64>, 64, SK1)) return rc;
        if (num_boxes > SK1)
            if (int rc = tf ? launch_tier(nms_select_sorted_kernel<Y3_NMS_TF, 256>, 256, SK4)
                            : launch_tier(nms_select_sorted_kernel<Y3_NMS_PY, 256>, 256, SK4)) return rc;
        if (num_boxes > SK4)
            if (int rc = tf ? launch_tier(nms_select_sorted_kernel<Y3_NMS_TF, 1024>, 1024, RCAP)
                            : launch_tier(nms_select_sorted_kernel<Y3_NMS_PY, 1024>, 1024, RCAP)) return rc;
    }
    hipLaunchKernelGGL(nms_gather_kernel, dim3(n), dim3(256), (class_num + 1) * sizeof(int), st, boxes,
                       scores, num_boxes, class_num, max_boxes, ws, out_boxes, out_scores, out_labels,
                       out_index, out_counts);
    Y3_CHECK_HIP(hipGetLastError());
    return Y3_OK;
}
