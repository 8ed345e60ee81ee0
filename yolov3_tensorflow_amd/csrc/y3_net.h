// The 75-conv graph object shared by the inference plan (y3_abi.hip: y3_net_forward) and the train step
// (y3_net_train.hip).  Tensor ids: 0 = network input; 1.. = conv outputs in creation order.
#pragma once
#include <vector>
#include <algorithm>
#include "y3_internal.h"

struct Tensor {
    int c;        // channels
    int sdiv;     // spatial divisor relative to the input (1,2,4,8,16,32)
    int last_use; // index of the last layer reading it (-1: never read)
    int ext;      // >=0: external output slot (fm1..fm3), storage provided by the caller
};

struct Layer {
    int k, stride, cin, cout, bn, act;
    int src, up, resid, dst;  // tensor ids (-1 = none)
    int c_up;
    const float *w, *scale, *shift;
    const float* w_alt = nullptr;   // dtype 4: the F(4x4,3x3) packing of a y3_conv_wino44_candidate layer (y3_net_set_layer_alt)
};

struct y3_train_state;                       // y3_net_train.hip: what a training forward leaves for loss / backward
void y3_train_state_free(y3_train_state* s);

struct y3_net {
    y3_ctx* ctx;
    int class_num;
    y3_train_state* train = nullptr;
    void* wgrad_stream = nullptr;   // y3_net_train_set_wgrad_stream: the weight gradients of backward run on this stream (nullptr: on the context's)
    void* own_stream = nullptr;     // the low-priority stream y3_net_train_set_wgrad_stream(net, Y3_OWN_STREAM) created (destroyed with the net)
    int dtype = 0;            // 0: fp32 (exact fp32 MFMA), 1: bf16 storage with fp32 accumulation,
                              // 2 / 3: fp32 tensors, products rebuilt from 3 / 2 bf16 planes (y3_conv_split.hip)
                              // 4: fp32, Winograd F(2x2,3x3) kernel for the layers y3_conv_wino_eligible accepts
    std::vector<Tensor> tensors;
    std::vector<Layer> layers;
    // cached plan
    int pn = 0, ph = 0, pw = 0;
    std::vector<size_t> offsets;  // byte offset of each tensor in the workspace (SIZE_MAX if external)
    size_t plan_bytes = 0;    // arena + conv scratch + flag regions
    size_t arena_bytes = 0;   // activations only; the conv (stream-K) scratch follows at this offset
    size_t scratch_bytes = 0; // stream-K accumulator slots / the V tensor of the two-kernel F(4x4,3x3) form (shared by all
                              // layers: launches on one stream are ordered)
    size_t flags_bytes = 0;   // one region of FLAG_WORDS "partial published" words per layer, after the scratch:
                              // all regions are zeroed by ONE memset at the start of a forward
    static constexpr size_t FLAG_WORDS = 512;   // >= the largest stream-K grid (512 direct / 256 Winograd workers)
    // profiling: one set of (layers+1) events per profiled forward, averaged by y3_net_get_layer_ms
    bool profiling = false;
    std::vector<std::vector<hipEvent_t>> event_sets;
    size_t sets_used = 0;

    int add_conv(int src, int cout, int k, int stride = 1, bool bn = true, bool act = true, int resid = -1,
                 int up = -1) {
        Layer l;
        l.k = k; l.stride = stride; l.cout = cout; l.bn = bn; l.act = act;
        l.src = src; l.up = up; l.resid = resid;
        l.c_up = up >= 0 ? tensors[up].c : 0;
        l.cin = tensors[src].c + l.c_up;
        l.w = l.scale = l.shift = nullptr;
        Tensor t;
        t.c = cout; t.sdiv = tensors[src].sdiv * stride; t.last_use = -1; t.ext = -1;
        tensors.push_back(t);
        l.dst = (int)tensors.size() - 1;
        const int li = (int)layers.size();
        tensors[src].last_use = li;
        if (up >= 0) tensors[up].last_use = li;
        if (resid >= 0) tensors[resid].last_use = li;
        layers.push_back(l);
        return l.dst;
    }
    // utils/layer_utils.py:25-32
    int res_block(int x, int f) {
        const int a = add_conv(x, f, 1);
        return add_conv(a, 2 * f, 3, 1, true, true, /*resid=*/x);
    }
    // utils/layer_utils.py:71-79 ; `up` >= 0 means the input is concat([upsample(up), x])
    void yolo_block(int x, int f, int up, int* route, int* net) {
        int t = add_conv(x, f, 1, 1, true, true, -1, up);
        t = add_conv(t, 2 * f, 3);
        t = add_conv(t, f, 1);
        t = add_conv(t, 2 * f, 3);
        t = add_conv(t, f, 1);
        *route = t;
        *net = add_conv(t, 2 * f, 3);
    }
    void build() {
        tensors.clear(); layers.clear();
        tensors.push_back(Tensor{3, 1, -1, -1});
        // utils/layer_utils.py:34-68 darknet53_body
        int t = add_conv(0, 32, 3);
        t = add_conv(t, 64, 3, 2);
        t = res_block(t, 32);
        t = add_conv(t, 128, 3, 2);
        for (int i = 0; i < 2; ++i) t = res_block(t, 64);
        t = add_conv(t, 256, 3, 2);
        for (int i = 0; i < 8; ++i) t = res_block(t, 128);
        const int route1 = t;
        t = add_conv(t, 512, 3, 2);
        for (int i = 0; i < 8; ++i) t = res_block(t, 256);
        const int route2 = t;
        t = add_conv(t, 1024, 3, 2);
        for (int i = 0; i < 4; ++i) t = res_block(t, 512);
        const int route3 = t;
        // model.py:53-78 yolov3_head
        const int det = 3 * (5 + class_num);
        int inter1, net1, inter2, net2, inter3, net3;
        yolo_block(route3, 512, -1, &inter1, &net1);
        const int fm1 = add_conv(net1, det, 1, 1, false, false);
        tensors[fm1].ext = 0;
        const int i1 = add_conv(inter1, 256, 1);
        yolo_block(route2, 256, i1, &inter2, &net2);
        const int fm2 = add_conv(net2, det, 1, 1, false, false);
        tensors[fm2].ext = 1;
        const int i2 = add_conv(inter2, 128, 1);
        yolo_block(route1, 128, i2, &inter3, &net3);
        const int fm3 = add_conv(net3, det, 1, 1, false, false);
        tensors[fm3].ext = 2;
    }

    size_t tensor_bytes(int id, int n, int h, int w) const {
        const Tensor& t = tensors[id];
        const size_t esize = (dtype == 1 && t.ext < 0 && id != 0) ? 2 : sizeof(float);
        return (size_t)n * (h / t.sdiv) * (w / t.sdiv) * t.c * esize;
    }

    // Liveness-based arena: a tensor's bytes are recycled after its last reader has been launched
    // (same stream => ordered), keeping the working set small enough to sit in the 256 MB Infinity Cache
    // for the deeper layers.
    void plan(int n, int h, int w) {
        if (n == pn && h == ph && w == pw) return;
        struct Free { size_t off, size; };
        std::vector<Free> freelist;
        std::vector<char> live(tensors.size(), 0);
        size_t top = 0, peak = 0;
        offsets.assign(tensors.size(), SIZE_MAX);
        auto rounded = [&](int id) { return (tensor_bytes(id, n, h, w) + 255) & ~(size_t)255; };
        auto release = [&](size_t off, size_t size) {
            freelist.push_back({off, size});
            std::sort(freelist.begin(), freelist.end(),
                      [](const Free& a, const Free& b) { return a.off < b.off; });
            std::vector<Free> merged;
            for (const Free& f : freelist) {
                if (!merged.empty() && merged.back().off + merged.back().size == f.off)
                    merged.back().size += f.size;
                else
                    merged.push_back(f);
            }
            if (!merged.empty() && merged.back().off + merged.back().size == top) {
                top = merged.back().off;  // give the tail back to the bump pointer
                merged.pop_back();
            }
            freelist.swap(merged);
        };
        for (size_t li = 0; li < layers.size(); ++li) {
            const Layer& l = layers[li];
            for (size_t t = 1; t < tensors.size(); ++t)
                if (live[t] && tensors[t].last_use < (int)li) {
                    release(offsets[t], rounded((int)t));
                    live[t] = 0;
                }
            if (tensors[l.dst].ext >= 0) continue;
            const size_t need = rounded(l.dst);
            size_t best = SIZE_MAX, best_size = SIZE_MAX;
            for (size_t f = 0; f < freelist.size(); ++f)
                if (freelist[f].size >= need && freelist[f].size < best_size) {
                    best = f;
                    best_size = freelist[f].size;
                }
            if (best != SIZE_MAX) {
                offsets[l.dst] = freelist[best].off;
                freelist[best].off += need;
                freelist[best].size -= need;
                if (freelist[best].size == 0) freelist.erase(freelist.begin() + best);
            } else {
                offsets[l.dst] = top;
                top += need;
            }
            live[l.dst] = 1;
            peak = std::max(peak, top);
        }
        arena_bytes = (peak + 255) & ~(size_t)255;
        scratch_bytes = 0;
        for (const Layer& l : layers) {
            y3_conv_desc d;
            d.n = n; d.h = h / tensors[l.src].sdiv; d.w = w / tensors[l.src].sdiv;
            d.cin = l.cin; d.c_up = l.c_up; d.cout = l.cout; d.k = l.k; d.stride = l.stride; d.act = l.act;
            if (dtype == 1) break;   // the bf16 kernels use no stream-K scratch
            scratch_bytes = std::max(scratch_bytes, y3_conv_workspace_bytes_impl(&d));
            if (dtype == 4) scratch_bytes = std::max(scratch_bytes, y3_conv_wino_workspace_bytes_impl(&d));
            // V = B^T d B of the two-kernel F(4x4,3x3) form, for the layers that can run on it at this size
            if (dtype == 4 && l.w_alt && y3_conv_wino44_preferred_impl(&d) && y3_conv_wino44_two_pass_impl(&d))
                scratch_bytes = std::max(scratch_bytes, y3_conv_wino44_workspace_bytes_impl(&d));
        }
        scratch_bytes = (scratch_bytes + 255) & ~(size_t)255;
        flags_bytes = scratch_bytes ? layers.size() * FLAG_WORDS * sizeof(unsigned) : 0;
        plan_bytes = arena_bytes + scratch_bytes + flags_bytes;
        pn = n; ph = h; pw = w;
    }
};

