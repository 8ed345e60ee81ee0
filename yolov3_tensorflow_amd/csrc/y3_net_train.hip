// The train step of train.py:72-115 behind ONE C entry (SURVEY.md §8b asks for whole-graph entries; VERDICT r2 missing #4):
//   y3_net_train_forward   yolov3.forward(inputs, is_training=True)        model.py:30-80 with batch-statistics BN
//   y3_net_train_loss      yolov3.compute_loss(y_pred, y_true)             model.py:192-365
//   y3_net_train_backward  optimizer.compute_gradients(loss[0] + l2_loss)  train.py:112 (the L2 term, the clip and the
//                                                                          update are y3_clip_update_multi's)
//   y3_net_train_step      the three in one call
// Host code only: it sequences the per-op entry points of include/yolo355.h over a caller-owned workspace, exactly the
// calls yolov3_tensorflow_amd/training.py used to make from Python (with torch.empty as the allocator).  Every buffer lives
// in the workspace: the forward's saved tensors are bump-allocated and stay until the next forward; everything else
// (packed kernels, the materialised concat's parts, gradients of activations) comes from a first-fit free list and is
// recycled as soon as its last consumer has been LAUNCHED - all launches go to the context's one stream, so launch order is
// execution order.  y3_net_train_workspace_bytes runs the same allocation sequence without launching anything.
#include <cstring>
#include <vector>
#include <algorithm>
#include "y3_net.h"

namespace {

constexpr float BN_EPS = 1e-5f;      // model.py:37
constexpr int MAXC = 1024;           // widest channel count of the graph (ones / zeros vectors, per-channel scratch)

struct Buf {
    size_t off = SIZE_MAX, bytes = 0;
    bool ok() const { return off != SIZE_MAX; }
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, top = 0, peak = 0;
    bool dry = false;                 // size query: hand out offsets, touch nothing
    bool overflow = false;
    struct Free { size_t off, size; };
    std::vector<Free> fl;
    static size_t round(size_t b) { return (b + 255) & ~(size_t)255; }
    void reset(void* ws, size_t bytes, bool dry_) { base = static_cast<char*>(ws); cap = bytes; top = peak = 0; dry = dry_; overflow = false; fl.clear(); }
    void rewind(size_t to) { top = to; fl.clear(); }
    Buf alloc(size_t bytes) {
        bytes = round(bytes ? bytes : 1);
        size_t best = SIZE_MAX, best_size = SIZE_MAX;
        for (size_t f = 0; f < fl.size(); ++f)
            if (fl[f].size >= bytes && fl[f].size < best_size) { best = f; best_size = fl[f].size; }
        Buf b;
        b.bytes = bytes;
        if (best != SIZE_MAX) {
            b.off = fl[best].off;
            fl[best].off += bytes;
            fl[best].size -= bytes;
            if (fl[best].size == 0) fl.erase(fl.begin() + best);
        } else {
            b.off = top;
            top += bytes;
            peak = std::max(peak, top);
            if (!dry && top > cap) overflow = true;
        }
        return b;
    }
    void release(Buf& b) {
        if (!b.ok()) return;
        fl.push_back({b.off, b.bytes});
        std::sort(fl.begin(), fl.end(), [](const Free& x, const Free& y) { return x.off < y.off; });
        std::vector<Free> merged;
        for (const Free& f : fl) {
            if (!merged.empty() && merged.back().off + merged.back().size == f.off) merged.back().size += f.size;
            else merged.push_back(f);
        }
        if (!merged.empty() && merged.back().off + merged.back().size == top) {
            top = merged.back().off;
            merged.pop_back();
        }
        fl.swap(merged);
        b = Buf();
    }
    float* p(const Buf& b) const { return b.ok() ? reinterpret_cast<float*>(base + b.off) : nullptr; }
};

__global__ void loss_total_kernel(const float* __restrict__ loss4, float* __restrict__ loss5) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // model.py:364: total = xy + wh + conf + class, summed in this order
        loss5[0] = loss4[0] + loss4[1] + loss4[2] + loss4[3];
        loss5[1] = loss4[0]; loss5[2] = loss4[1]; loss5[3] = loss4[2]; loss5[4] = loss4[3];
    }
}

}  // namespace

struct y3_train_state {
    Arena arena;
    int n = 0, h = 0, w = 0;
    const float* x = nullptr;
    bool have_fwd = false, have_loss = false;
    size_t saved_top = 0;                    // arena top after forward (+ loss): backward's temporaries start here
    // persistent scratch (allocated first)
    Buf ones, zeros, sk_ws, reduce_sc, bnbwd_sc, loss_sc, bias_sc, wgrad_sc, loss4;
    std::vector<Buf> tens;                   // activation of every tensor id (0: the caller's x)
    std::vector<Buf> xin;                    // per layer: the materialised concat input (fused-upsample layers), else empty
    std::vector<Buf> z, stats;               // per BN layer: raw conv output, [4][cout] mean / inv_std / scale / shift
    Buf fm_grad[3];                          // d loss / d feature_map_i, [n,g,g,det_pad]
    int fm_tensor[3] = {-1, -1, -1};         // tensor ids of feature maps 1..3 (13-, 26-, 52-grid)
    // the second stream of backward (y3_net_train_set_wgrad_stream): its context and the events that order it against the
    // context's stream - "dz of this layer is ready" (main -> side), "this layer's weight gradient is done" (side -> main)
    y3_ctx* side_ctx = nullptr;
    void* side_stream = nullptr;
    static constexpr int NEV = 8;
    hipEvent_t ev_dz[NEV] = {}, ev_done[NEV] = {};
};

void y3_train_state_free(y3_train_state* s) {
    if (!s) return;
    for (int i = 0; i < y3_train_state::NEV; ++i) {
        if (s->ev_dz[i]) (void)hipEventDestroy(s->ev_dz[i]);
        if (s->ev_done[i]) (void)hipEventDestroy(s->ev_done[i]);
    }
    if (s->side_ctx) (void)y3_ctx_destroy(s->side_ctx);
    delete s;
}

namespace {

int det_pad_of(int class_num) { return ((3 * (5 + class_num) + 31) / 32) * 32; }

y3_conv_desc desc_of(const y3_net* net, const Layer& l, int n, int h, int w) {
    y3_conv_desc d;
    const int sd = net->tensors[l.src].sdiv;
    d.n = n; d.h = h / sd; d.w = w / sd;
    d.cin = l.cin; d.c_up = 0; d.cout = l.cout; d.k = l.k; d.stride = l.stride; d.act = 0;
    return d;
}

bool wino_layer(const y3_net* net, const y3_conv_desc& d) { return net->dtype == 4 && y3_conv_wino_eligible(&d) == 1; }
int split_planes(const y3_net* net, const Layer& l) { return l.cin == 3 ? 0 : (net->dtype == 2 ? 3 : net->dtype == 3 ? 2 : 0); }

#define Y3_TRY(expr)                    \
    do {                                \
        if (!dry) {                     \
            const int rc_ = (expr);     \
            if (rc_ != Y3_OK) return rc_; \
        }                               \
    } while (0)

// sizes the persistent scratch and runs (or, dry, only sizes) the training forward
int forward_impl(y3_net* net, const y3_train_var* vars, const float* x, int n, int h, int w, const y3_train_opts* o,
                 void* ws, size_t ws_bytes, bool dry) {
    if (!net->train) net->train = new y3_train_state;
    y3_train_state& S = *net->train;
    y3_ctx* ctx = net->ctx;
    Arena& A = S.arena;
    A.reset(ws, ws_bytes, dry);
    S.n = n; S.h = h; S.w = w; S.x = x;
    S.have_fwd = S.have_loss = false;
    const size_t nl = net->layers.size(), nt = net->tensors.size();
    S.tens.assign(nt, Buf());
    S.xin.assign(nl, Buf());
    S.z.assign(nl, Buf());
    S.stats.assign(nl, Buf());
    const int C = net->class_num, det_pad = det_pad_of(C);

    // ---- persistent scratch --------------------------------------------------------------------------------------------
    S.ones = A.alloc(MAXC * 4);
    S.zeros = A.alloc(MAXC * 4);
    size_t sk = 0, wg = 0;
    for (const Layer& l : net->layers) {
        y3_conv_desc d = desc_of(net, l, n, h, w);
        sk = std::max(sk, y3_conv_workspace_bytes(&d));
        sk = std::max(sk, y3_conv_wino_workspace_bytes(&d));
        // the data gradient of a stride-1 3x3 conv runs as a conv with the channel axes swapped
        y3_conv_desc g = d;
        g.cin = l.bn ? l.cout : det_pad; g.cout = l.cin;
        if (l.k == 3 && l.stride == 1) sk = std::max(sk, y3_conv_wino_workspace_bytes(&g));
        // V of the two-kernel F(4x4,3x3) form, forward and data gradient
        if (net->dtype == 4 && l.bn && y3_conv_wino44_preferred(&d) == 1 && y3_conv_wino44_two_pass_impl(&d))
            sk = std::max(sk, y3_conv_wino44_workspace_bytes(&d));
        if (net->dtype == 4 && l.k == 3 && l.stride == 1 && l.up < 0 && y3_conv_wino44_preferred(&g) == 1 && y3_conv_wino44_two_pass_impl(&g))
            sk = std::max(sk, y3_conv_wino44_workspace_bytes(&g));
        wg = std::max(wg, y3_conv_wgrad_scratch_bytes(&d));
        if (y3_conv_wgrad_wino_eligible(&d)) wg = std::max(wg, y3_conv_wgrad_wino_scratch_bytes(&d));
    }
    {
        y3_conv_desc d128 = {1, 8, 8, 128, 0, 128, 3, 1, 0};      // the library's stream-K scratch for any 3x3 conv, Cout >= 128
        sk = std::max(sk, y3_conv_workspace_bytes(&d128));
    }
    S.sk_ws = A.alloc(std::max<size_t>(sk, 256));
    S.reduce_sc = A.alloc(y3_reduce_scratch_bytes(MAXC));
    S.bnbwd_sc = A.alloc(y3_bn_bwd_scratch_bytes(MAXC));
    size_t ls = 0;
    for (int s : {32, 16, 8}) ls = std::max(ls, y3_loss_scratch_bytes(n, h / s, w / s));
    S.loss_sc = A.alloc(ls);
    S.bias_sc = A.alloc((size_t)1024 * det_pad * 4 + (size_t)det_pad * 4);
    S.wgrad_sc = A.alloc(std::max<size_t>(wg, 256));
    S.loss4 = A.alloc(64);
    if (!dry) {
        if (A.overflow) { y3_set_error("y3_net_train_forward: workspace too small"); return Y3_EINVAL; }
        void* stage = nullptr;
        if (int rc = y3_ctx_stage_acquire(ctx, MAXC * 4, &stage)) return rc;
        for (int i = 0; i < MAXC; ++i) static_cast<float*>(stage)[i] = 1.f;
        Y3_CHECK_HIP(hipMemcpyAsync(A.p(S.ones), stage, MAXC * 4, hipMemcpyHostToDevice, ctx->stream));
        if (int rc = y3_ctx_stage_release(ctx)) return rc;
        Y3_CHECK_HIP(hipMemsetAsync(A.p(S.zeros), 0, MAXC * 4, ctx->stream));
    }
    void* skp = A.p(S.sk_ws);
    const size_t skb = S.sk_ws.bytes;
    auto tptr = [&](int id) -> const float* { return id == 0 ? x : A.p(S.tens[id]); };

    // ---- layers ---------------------------------------------------------------------------------------------------------
    for (size_t i = 0; i < nl; ++i) {
        const Layer& l = net->layers[i];
        const y3_train_var& v = vars[i];
        y3_conv_desc d = desc_of(net, l, n, h, w);
        const long long in_rows = (long long)n * d.h * d.w;
        const float* xin = tptr(l.src);
        if (l.up >= 0) {      // training materialises concat([upsample(up), route]) (model.py:61-62,71-72)
            const int cu = net->tensors[l.up].c, cx = net->tensors[l.src].c;
            Buf upt = A.alloc((size_t)in_rows * cu * 4);
            S.xin[i] = A.alloc((size_t)in_rows * (cu + cx) * 4);
            Y3_TRY(y3_upsample_nearest(ctx, tptr(l.up), n, d.h / 2, d.w / 2, cu, d.h, d.w, A.p(upt)));
            Y3_TRY(y3_concat_channels(ctx, A.p(upt), cu, xin, cx, in_rows, A.p(S.xin[i])));
            A.release(upt);
            xin = A.p(S.xin[i]);
        }
        const int cout = l.cout, ho = d.h / l.stride, wo = d.w / l.stride;
        const long long rows = (long long)n * ho * wo;
        const bool wino = wino_layer(net, d);
        // the F(4x4,3x3) kernel where it fills the chip (y3_conv_wino44_preferred: the layers with Cin >= 64 at the bench sizes)
        const bool wino44 = wino && l.bn && y3_conv_wino44_preferred(&d) == 1;
        const int planes = split_planes(net, l);
        const int nblk = (l.bn && !planes) ? y3_conv_stats_blocks(&d, wino44 ? 2 : wino ? 1 : 0) : 0;
        Buf part = nblk ? A.alloc((size_t)nblk * 2 * cout * 4) : Buf();
        // the kernel in this step's packing (the variable changes every step)
        const size_t kelems = (size_t)l.k * l.k * l.cin * cout;
        Buf wp;
        const void* wdev = v.weights;                       // Cin = 3 stem: HWIO as it is
        if (wino44) {
            wp = A.alloc((size_t)36 * l.cin * cout * 4);
            Y3_TRY(y3_pack_conv_weights_wino44(ctx, v.weights, l.cin, cout, A.p(wp)));
            wdev = A.p(wp);
        } else if (wino) {
            wp = A.alloc((size_t)16 * l.cin * cout * 4);
            Y3_TRY(y3_pack_conv_weights_wino(ctx, v.weights, l.cin, cout, A.p(wp)));
            wdev = A.p(wp);
        } else if (planes) {
            wp = A.alloc(kelems * 2 * planes);
            Y3_TRY(y3_pack_conv_weights_split(ctx, v.weights, l.k, l.cin, cout, planes, A.p(wp)));
            wdev = A.p(wp);
        } else if (l.cin != 3) {
            wp = A.alloc(kelems * 4);
            Y3_TRY(y3_pack_conv_weights(ctx, v.weights, l.k, l.cin, cout, A.p(wp)));
            wdev = A.p(wp);
        }
        const float* ones = A.p(S.ones);
        const float* zeros = A.p(S.zeros);
        if (l.bn) {
            S.z[i] = A.alloc((size_t)rows * cout * 4);
            float* z = A.p(S.z[i]);
            if (wino44)
                Y3_TRY(y3_conv2d_fwd_wino44_stats(ctx, &d, xin, static_cast<const float*>(wdev), ones, zeros, z, A.p(part), skp, skb));
            else if (wino)
                Y3_TRY(y3_conv2d_fwd_wino_stats(ctx, &d, xin, static_cast<const float*>(wdev), ones, zeros, z, A.p(part), skp, skb));
            else if (planes)
                Y3_TRY(y3_conv2d_fwd_split(ctx, &d, planes, xin, nullptr, wdev, ones, zeros, nullptr, z, skp, skb));
            else if (nblk)
                Y3_TRY(y3_conv2d_fwd_stats(ctx, &d, xin, static_cast<const float*>(wdev), ones, zeros, z, A.p(part), skp, skb));
            else
                Y3_TRY(y3_conv2d_fwd(ctx, &d, xin, nullptr, static_cast<const float*>(wdev), ones, zeros, nullptr, z, skp, skb));
            A.release(wp);
            S.stats[i] = A.alloc((size_t)4 * cout * 4);
            float* st = A.p(S.stats[i]);
            if (nblk)
                Y3_TRY(y3_bn_train_stats_partials(ctx, A.p(part), nblk, rows, cout, v.gamma, v.beta, BN_EPS, o->bn_decay, st,
                                                  st + cout, st + 2 * cout, st + 3 * cout, v.moving_mean, v.moving_variance));
            else
                Y3_TRY(y3_bn_train_stats(ctx, z, rows, cout, v.gamma, v.beta, BN_EPS, o->bn_decay, st, st + cout, st + 2 * cout,
                                         st + 3 * cout, v.moving_mean, v.moving_variance, A.p(S.reduce_sc)));
            A.release(part);
            S.tens[l.dst] = A.alloc((size_t)rows * cout * 4);
            Y3_TRY(y3_bn_apply_fwd(ctx, z, st + 2 * cout, st + 3 * cout, l.resid >= 0 ? tptr(l.resid) : nullptr, rows, cout, 1,
                                   A.p(S.tens[l.dst])));
        } else {              // detection conv: bias, linear (model.py:55-57)
            S.tens[l.dst] = A.alloc((size_t)rows * cout * 4);
            float* y = A.p(S.tens[l.dst]);
            if (planes)
                Y3_TRY(y3_conv2d_fwd_split(ctx, &d, planes, xin, nullptr, wdev, ones, v.biases, nullptr, y, skp, skb));
            else
                Y3_TRY(y3_conv2d_fwd(ctx, &d, xin, nullptr, static_cast<const float*>(wdev), ones, v.biases, nullptr, y, skp, skb));
            A.release(wp);
        }
        const int e = net->tensors[l.dst].ext;
        if (e >= 0) S.fm_tensor[e] = l.dst;
    }
    // the loss gradients live with the saved tensors (backward may be re-run from them)
    for (int i = 0; i < 3; ++i) {
        const int s = 32 >> i;
        S.fm_grad[i] = A.alloc((size_t)n * (h / s) * (w / s) * det_pad * 4);
    }
    S.saved_top = A.top;
    if (!dry && A.overflow) { y3_set_error("y3_net_train_forward: workspace too small"); return Y3_EINVAL; }
    S.have_fwd = !dry;
    return Y3_OK;
}

int loss_impl(y3_net* net, const float* const y_true[3], const y3_train_opts* o, float* loss5, bool dry) {
    y3_train_state& S = *net->train;
    Arena& A = S.arena;
    y3_ctx* ctx = net->ctx;
    const int C = net->class_num, det_pad = det_pad_of(C);
    if (!dry) Y3_CHECK_HIP(hipMemsetAsync(A.p(S.loss4), 0, 16, ctx->stream));
    for (int i = 0; i < 3; ++i) {
        const int s = 32 >> i, gh = S.h / s, gw = S.w / s;
        if (dry) continue;
        Y3_CHECK_HIP(hipMemsetAsync(A.p(S.fm_grad[i]), 0, (size_t)S.n * gh * gw * det_pad * 4, ctx->stream));   // pad lanes stay 0
        // model.py:352-355: anchors [6:9], [3:6], [0:3] for the 13-, 26-, 52-grid maps
        const float* anc = o->anchors + 2 * 3 * (2 - i);
        const int rc = y3_loss_layer(ctx, A.p(S.tens[S.fm_tensor[i]]), y_true[i], S.n, gh, gw, C, S.h, S.w, anc,
                                     o->use_label_smooth, o->use_focal_loss, i > 0, A.p(S.loss4), A.p(S.fm_grad[i]), det_pad,
                                     A.p(S.loss_sc), S.loss_sc.bytes);
        if (rc != Y3_OK) return rc;
    }
    if (!dry && loss5) {
        hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, ctx->stream, A.p(S.loss4), loss5);
        Y3_CHECK_HIP(hipGetLastError());
    }
    S.have_loss = !dry;
    return Y3_OK;
}

int backward_impl(y3_net* net, const y3_train_var* vars, float* flat_grad, y3_grad_ready_fn ready, void* user, bool dry) {
    y3_train_state& S = *net->train;
    Arena& A = S.arena;
    y3_ctx* ctx = net->ctx;
    A.rewind(S.saved_top);
    const int n = S.n, h = S.h, w = S.w;
    const size_t nl = net->layers.size(), nt = net->tensors.size();
    const int C = net->class_num, det_pad = det_pad_of(C);
    auto trainable = [&](long long g) { return g >= 0; };
    auto gptr = [&](long long g) -> float* { return g >= 0 ? flat_grad + g : nullptr; };
    // earliest layer holding a trainable variable: gradients need not flow below it
    int first = -1;
    for (size_t i = 0; i < nl; ++i) {
        const y3_train_var& v = vars[i];
        if (trainable(v.g_weights) || trainable(v.g_gamma) || trainable(v.g_beta) || trainable(v.g_biases)) { first = (int)i; break; }
    }
    if (first < 0) return Y3_OK;
    auto needs = [&](int t) { return t > 0 && (t - 1) >= first; };       // tensor t is produced by layer t-1
    std::vector<Buf> grads(nt);
    std::vector<char> have(nt, 0), own(nt, 0);                            // own: grads[t] is an arena buffer of this pass
    for (int i = 0; i < 3; ++i) { grads[S.fm_tensor[i]] = S.fm_grad[i]; have[S.fm_tensor[i]] = 1; }
    auto tptr = [&](int id) -> const float* { return id == 0 ? S.x : A.p(S.tens[id]); };
    auto tbytes = [&](int id) { return (size_t)n * (h / net->tensors[id].sdiv) * (w / net->tensors[id].sdiv) * net->tensors[id].c * 4; };
    void* skp = A.p(S.sk_ws);
    const size_t skb = S.sk_ws.bytes;
    const float* ones = A.p(S.ones);
    const float* zeros = A.p(S.zeros);

    // Two streams (y3_net_train_set_wgrad_stream).  A layer's weight gradient is off the critical path - nothing of this pass
    // reads it - while its data gradient and the BN backward of the layer below are on it: the weight gradient goes to the
    // side stream behind "dz ready", and the context's stream waits for it only one layer later, just before the buffers it
    // read go back to the arena (their release, and the layer's `ready` call, are deferred by that one layer; the dry run
    // defers alike, so the workspace is sized for it).  What overlaps: a matrix-pipe-bound weight gradient with the HBM-bound
    // BN passes of the next layer.
    const bool two = net->wgrad_stream != nullptr;
    y3_ctx* sctx = ctx;
    if (two && !dry) {
        if (S.side_ctx && S.side_stream != net->wgrad_stream) { (void)y3_ctx_destroy(S.side_ctx); S.side_ctx = nullptr; }
        if (!S.side_ctx) {
            if (int rc = y3_ctx_create(ctx->device, net->wgrad_stream, &S.side_ctx)) return rc;
            S.side_stream = net->wgrad_stream;
        }
        for (int e = 0; e < y3_train_state::NEV; ++e) {
            if (!S.ev_dz[e]) Y3_CHECK_HIP(hipEventCreateWithFlags(&S.ev_dz[e], hipEventDisableTiming));
            if (!S.ev_done[e]) Y3_CHECK_HIP(hipEventCreateWithFlags(&S.ev_done[e], hipEventDisableTiming));
        }
        sctx = S.side_ctx;
    }
#ifndef Y3_WGRAD_DEPTH
#define Y3_WGRAD_DEPTH 1            // layers a weight gradient may lag behind the main stream (measured: profiles/r06_wgrad_stream_ab.txt)
#endif
    static_assert(Y3_WGRAD_DEPTH >= 1 && Y3_WGRAD_DEPTH < y3_train_state::NEV, "one event pair per weight gradient in flight");
    struct Deferred {
        Buf a, b;                       // buffers the weight gradient read: released once the main stream waits for it
        hipEvent_t done = nullptr;
        long long g_end = -1;
    };
    std::vector<Deferred> pend;         // oldest first
    int flip = 0;
    auto flush_oldest = [&]() -> int {
        Deferred q = pend.front();
        pend.erase(pend.begin());
        if (!dry && q.done) Y3_CHECK_HIP(hipStreamWaitEvent(ctx->stream, q.done, 0));
        if (q.a.ok()) A.release(q.a);
        if (q.b.ok()) A.release(q.b);
        if (ready && q.g_end >= 0 && !dry) ready(user, q.g_end);
        return Y3_OK;
    };

    // grads[t] (+)= src[:, offset : offset + c]
    auto accumulate_into = [&](int t, const float* src, int src_channels, int offset, long long rows, int c) -> int {
        if (!have[t]) { grads[t] = A.alloc(tbytes(t)); own[t] = 1; }
        Y3_TRY(y3_slice_accumulate(ctx, src, src_channels, offset, rows, c, have[t] ? 1 : 0, A.p(grads[t])));
        have[t] = 1;
        return Y3_OK;
    };

    // The BN backward reduction (column sums of g' and g' * zhat over dy and z) rides in the epilogue of the data gradient
    // that WRITES that dy, where it can: the gradient of tensor t is complete once its first consumer in forward order - the
    // last one backward visits - has added its part; if that consumer is a stride-1 1x1 conv on the direct kernel, its data
    // gradient's epilogue has the finished dy in registers (conv + what the later consumers had left) and reads z beside it.
    // That is every residual block's output, every yolo-block 3x3 and every stride-2 conv: the large tensors.  The separate
    // reduction pass (col_reduce<1>: z and dy from memory) is then skipped for that layer.
#ifndef Y3_BN_FUSE
#define Y3_BN_FUSE 1
#endif
    std::vector<int> first_consumer(nt, -1);
    for (int li = (int)nl - 1; li >= 0; --li) {
        const Layer& q = net->layers[li];
        first_consumer[q.src] = li;
        if (q.up >= 0) first_consumer[q.up] = li;
        if (q.resid >= 0) first_consumer[q.resid] = li;
    }
    std::vector<Buf> fused_part(nl);
    std::vector<int> fused_nb(nl, 0);

    for (int i = (int)nl - 1; i >= first; --i) {
        const Layer& l = net->layers[i];
        const y3_train_var& v = vars[i];
        const int dst = l.dst;
        if (!have[dst]) continue;
        y3_conv_desc d = desc_of(net, l, n, h, w);
        const int cout = l.cout, cin = l.cin;
        const float* xin = S.xin[i].ok() ? A.p(S.xin[i]) : tptr(l.src);
        Buf dy = grads[dst];
        Buf dz_buf;                      // where dz lives (dy itself, or a fresh buffer)
        bool dy_given_away = false;      // dy became the shortcut's gradient
        int dz_stride;
        const float* w_d = v.weights;    // the kernel as the data gradient reads it: [k*k][cin][dz_stride]
        Buf w_d_buf;
        long long rows;
        if (l.bn) {
            rows = (long long)n * (d.h / l.stride) * (d.w / l.stride);
            dz_buf = dy;                 // BN backward runs in place ...
            if (l.resid >= 0 && needs(l.resid)) {
                if (have[l.resid]) {
                    if (int rc = accumulate_into(l.resid, A.p(dy), cout, 0, rows, cout)) return rc;
                } else {
                    // ... unless dy is also the first contribution to the shortcut's gradient (res_block: net + shortcut,
                    // utils/layer_utils.py:30): then dy itself becomes that gradient (no copy), dz goes elsewhere
                    grads[l.resid] = dy;
                    own[l.resid] = own[dst];
                    have[l.resid] = 1;
                    dy_given_away = true;
                    dz_buf = A.alloc((size_t)rows * cout * 4);
                }
            }
            Buf tmp;
            float *dgam = gptr(v.g_gamma), *dbet = gptr(v.g_beta);
            if (!dgam || !dbet) tmp = A.alloc((size_t)2 * cout * 4);
            const float* st = A.p(S.stats[i]);
            if (fused_nb[i] > 0) {
                Y3_TRY(y3_bn_train_bwd_partials(ctx, A.p(S.z[i]), A.p(dy), v.gamma, st + 2 * cout, st + 3 * cout, st, st + cout, rows,
                                                cout, A.p(fused_part[i]), fused_nb[i], dgam ? dgam : A.p(tmp),
                                                dbet ? dbet : A.p(tmp) + cout, A.p(dz_buf), A.p(S.bnbwd_sc)));
                A.release(fused_part[i]);
            } else {
                Y3_TRY(y3_bn_train_bwd(ctx, A.p(S.z[i]), A.p(dy), v.gamma, st + 2 * cout, st + 3 * cout, st, st + cout, rows, cout,
                                       dgam ? dgam : A.p(tmp), dbet ? dbet : A.p(tmp) + cout, A.p(dz_buf), A.p(S.bnbwd_sc)));
            }
            A.release(tmp);
            dz_stride = cout;
        } else {
            dz_stride = det_pad;
            rows = (long long)n * d.h * d.w;
            if (trainable(v.g_biases)) {
                float* tmp = A.p(S.bias_sc) + (size_t)1024 * det_pad;      // [det_pad] behind the kernel's own scratch
                Y3_TRY(y3_bias_grad(ctx, A.p(dy), rows, dz_stride, tmp, A.p(S.bias_sc)));
                if (!dry) Y3_CHECK_HIP(hipMemcpyAsync(gptr(v.g_biases), tmp, (size_t)cout * 4, hipMemcpyDeviceToDevice, ctx->stream));
            }
            dz_buf = dy;
            // the data gradient reads the kernel as [k*k][cin][dz_stride]: zero-extend its last axis
            w_d_buf = A.alloc((size_t)l.k * l.k * cin * dz_stride * 4);
            Y3_TRY(y3_pad_channels(ctx, v.weights, cout, (long long)l.k * l.k * cin, dz_stride, A.p(w_d_buf)));
            w_d = A.p(w_d_buf);
        }
        const float* dz = A.p(dz_buf);
        const bool on_side = two && trainable(v.g_weights);
        if (trainable(v.g_weights)) {
            if (on_side && !dry) {
                Y3_CHECK_HIP(hipEventRecord(S.ev_dz[flip], ctx->stream));
                Y3_CHECK_HIP(hipStreamWaitEvent(sctx->stream, S.ev_dz[flip], 0));
            }
            y3_ctx* wctx = on_side ? sctx : ctx;
            if (net->dtype == 4 && y3_conv_wgrad_wino_eligible(&d) == 1)
                Y3_TRY(y3_conv_wgrad_wino(wctx, &d, xin, dz, dz_stride, gptr(v.g_weights), A.p(S.wgrad_sc), S.wgrad_sc.bytes));
            else
                Y3_TRY(y3_conv_wgrad(wctx, &d, xin, dz, dz_stride, gptr(v.g_weights), A.p(S.wgrad_sc), S.wgrad_sc.bytes));
            if (on_side && !dry) Y3_CHECK_HIP(hipEventRecord(S.ev_done[flip], sctx->stream));
        }
        // this layer's gradients are complete (enqueued) - with two streams: once the main stream waits for the side one (flush)
        if (!on_side && ready && v.g_end >= 0 && !dry) ready(user, v.g_end);
        const int src = l.src, up = l.up;
        const bool need_src = needs(src), need_up = up >= 0 && needs(up);
        if (need_src || need_up) {
            const int planes = (l.stride == 1 && cin % 4 == 0 && dz_stride % 32 == 0) ? split_planes(net, l) : 0;
            y3_conv_desc g = d;
            g.cin = dz_stride; g.cout = cin; g.k = 3; g.stride = 1;
            const bool wino_d = net->dtype == 4 && up < 0 && l.k == 3 && l.stride == 1 && y3_conv_wino_eligible(&g) == 1;
            const bool wino44_d = wino_d && y3_conv_wino44_preferred(&g) == 1;
            Buf wk;
            if (wino44_d) {
                wk = A.alloc((size_t)36 * cin * dz_stride * 4);
                Y3_TRY(y3_pack_conv_weights_wino44_dgrad(ctx, w_d, cin, dz_stride, A.p(wk)));
            } else if (wino_d) {
                wk = A.alloc((size_t)16 * cin * dz_stride * 4);
                Y3_TRY(y3_pack_conv_weights_wino_dgrad(ctx, w_d, cin, dz_stride, A.p(wk)));
            } else if (planes) {
                wk = A.alloc((size_t)planes * l.k * l.k * cin * dz_stride * 2);
                Y3_TRY(y3_pack_conv_weights_split_dgrad(ctx, w_d, l.k, cin, dz_stride, planes, A.p(wk)));
            }
            auto dgrad = [&](int accumulate, float* dx) -> int {
                if (wino44_d)
                    Y3_TRY(y3_conv2d_dgrad_wino44(ctx, &d, dz, dz_stride, A.p(wk), ones, zeros, accumulate, dx, skp, skb));
                else if (wino_d)
                    Y3_TRY(y3_conv2d_dgrad_wino(ctx, &d, dz, dz_stride, A.p(wk), ones, zeros, accumulate, dx, skp, skb));
                else if (planes)
                    Y3_TRY(y3_conv2d_dgrad_split(ctx, &d, planes, dz, dz_stride, A.p(wk), ones, zeros, accumulate, dx, skp, skb));
                else
                    Y3_TRY(y3_conv2d_dgrad(ctx, &d, dz, dz_stride, w_d, ones, zeros, accumulate, dx, skp, skb));
                return Y3_OK;
            };
            if (up >= 0) {
                const long long in_rows = (long long)n * d.h * d.w;
                Buf dcat = A.alloc((size_t)in_rows * cin * 4);
                if (int rc = dgrad(0, A.p(dcat))) return rc;
                const int cu = net->tensors[up].c;
                if (need_up) {
                    if (!have[up]) { grads[up] = A.alloc(tbytes(up)); own[up] = 1; }
                    Y3_TRY(y3_upsample2x_bwd(ctx, A.p(dcat), cin, n, d.h / 2, d.w / 2, cu, have[up] ? 1 : 0, A.p(grads[up])));
                    have[up] = 1;
                }
                if (need_src)
                    if (int rc = accumulate_into(src, A.p(dcat), cin, cu, in_rows, cin - cu)) return rc;
                A.release(dcat);
            } else {
                if (!have[src]) { grads[src] = A.alloc(tbytes(src)); own[src] = 1; }
                // src is the output of layer src - 1: fuse its BN backward reduction where this is the last contribution
                const int pj = src - 1;
                const int nbf = (!wino44_d && !wino_d && !planes && pj >= first && pj >= 0 && net->layers[pj].bn &&
                                 first_consumer[src] == i && S.z[pj].ok())
                                    ? (Y3_BN_FUSE ? y3_conv_dgrad_stats_blocks_impl(&d) : 0) : 0;
                if (nbf > 0) {
                    fused_part[pj] = A.alloc((size_t)nbf * 2 * cin * 4);
                    fused_nb[pj] = nbf;
                    if (!dry) {
                        y3_sk_opts o;
                        o.err = ctx->err_host;
                        o.stats = A.p(fused_part[pj]);
                        o.bwd_z = A.p(S.z[pj]);
                        o.bwd_vec = A.p(S.stats[pj]);
                        if (int rc = y3_launch_conv_dgrad(ctx->stream, &d, dz, dz_stride, w_d, ones, zeros, have[src] ? 1 : 0,
                                                          A.p(grads[src]), skp, skb, &o))
                            return rc;
                    }
                } else if (int rc = dgrad(have[src] ? 1 : 0, A.p(grads[src]))) {
                    return rc;
                }
                have[src] = 1;
            }
            A.release(wk);
        }
        A.release(w_d_buf);
        // the consumed gradient goes back to the free list (the loss gradients are not this pass's to free) - at once, or, when
        // the side stream may still be reading it, behind the wait of the NEXT layer's end
        while ((int)pend.size() >= Y3_WGRAD_DEPTH)
            if (int rc = flush_oldest()) return rc;
        Buf rel_a = (dz_buf.off != dy.off) ? dz_buf : Buf();
        Buf rel_b = (!dy_given_away && own[dst]) ? grads[dst] : Buf();
        if (on_side) {
            Deferred q;
            q.a = rel_a;
            q.b = rel_b;
            q.done = dry ? nullptr : S.ev_done[flip];
            q.g_end = v.g_end;
            pend.push_back(q);
            flip = (flip + 1) % y3_train_state::NEV;
        } else {
            if (rel_a.ok()) A.release(rel_a);
            if (rel_b.ok()) A.release(rel_b);
        }
        grads[dst] = Buf();
    }
    while (!pend.empty())
        if (int rc = flush_oldest()) return rc;
    if (!dry && A.overflow) { y3_set_error("y3_net_train_backward: workspace too small"); return Y3_EINVAL; }
    return Y3_OK;
}

int check_common(const char* who, y3_net* net, int n, int h, int w) {
    Y3_CHECK_ARG(net, "%s: null net", who);
    Y3_CHECK_ARG(n > 0 && h > 0 && w > 0 && h % 32 == 0 && w % 32 == 0,
                 "%s: the batch must be positive and the input size a positive multiple of 32 (got %d x %dx%d)", who, n, h, w);
    Y3_CHECK_ARG(net->dtype != 1, "%s: the train step is fp32 (net dtype 0, 2, 3 or 4)", who);
    return Y3_OK;
}

int check_vars(const char* who, const y3_net* net, const y3_train_var* vars) {
    Y3_CHECK_ARG(vars, "%s: null variable table", who);
    for (size_t i = 0; i < net->layers.size(); ++i) {
        const Layer& l = net->layers[i];
        const y3_train_var& v = vars[i];
        Y3_CHECK_ARG(v.weights, "%s: layer %zu has no kernel", who, i);
        if (l.bn) Y3_CHECK_ARG(v.gamma && v.beta && v.moving_mean && v.moving_variance, "%s: layer %zu: batch-norm variables missing", who, i);
        else Y3_CHECK_ARG(v.biases, "%s: layer %zu: bias missing", who, i);
    }
    return Y3_OK;
}

}  // namespace

extern "C" int y3_net_train_set_wgrad_stream(y3_net* net, void* stream) {
    Y3_CHECK_ARG(net, "y3_net_train_set_wgrad_stream: null net");
    if (stream == Y3_OWN_STREAM) {
        // A stream of the LOWEST priority, created here: (1) the weight gradient is the work that can wait - where the two
        // streams compete, the critical path should win; (2) a stream of another priority gets a hardware queue of its own.
        // Measured why (2) matters: a caller's ordinary stream can land on the hardware queue of the context's stream once
        // enough streams are alive in the process (bench.py's default line: c2, detect and c5 each leave two behind) - the
        // event waits between the two then serialise INSIDE one queue: 96 ms per step instead of 79.6
        // (profiles/r06_wgrad_stream_ab.txt).
        if (!net->own_stream) {
            if (!net->ctx) { y3_set_error("y3_net_train_set_wgrad_stream: the net was created without a context"); return Y3_ESTATE; }
            int prev = -1;
            Y3_CHECK_HIP(hipGetDevice(&prev));
            Y3_CHECK_HIP(hipSetDevice(net->ctx->device));
            int least = 0, greatest = 0;
            hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            hipStream_t s = nullptr;
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
            (void)hipSetDevice(prev);
            if (e != hipSuccess) { y3_set_error("y3_net_train_set_wgrad_stream: %s", hipGetErrorString(e)); return Y3_EHIP; }
            net->own_stream = s;
        }
        stream = net->own_stream;
    }
    net->wgrad_stream = stream;
    return Y3_OK;
}

extern "C" size_t y3_net_train_workspace_bytes(y3_net* net, const y3_train_var* vars, int n, int h, int w) {
    if (check_common("y3_net_train_workspace_bytes", net, n, h, w) != Y3_OK || !vars) return 0;
    // the same allocation sequence as a real step, nothing launched; the state is rebuilt by the next forward anyway
    y3_train_opts o;
    std::memset(&o, 0, sizeof(o));
    if (forward_impl(net, vars, nullptr, n, h, w, &o, nullptr, 0, true) != Y3_OK) return 0;
    const float* yt[3] = {nullptr, nullptr, nullptr};
    if (loss_impl(net, yt, &o, nullptr, true) != Y3_OK) return 0;
    std::vector<y3_train_var> all(vars, vars + net->layers.size());
    if (backward_impl(net, all.data(), nullptr, nullptr, nullptr, true) != Y3_OK) return 0;
    const size_t peak = net->train->arena.peak;
    net->train->have_fwd = net->train->have_loss = false;
    return (peak + 255) & ~(size_t)255;
}

extern "C" int y3_net_train_forward(y3_net* net, const y3_train_var* vars, const float* x, int n, int h, int w,
                                    const y3_train_opts* opts, void* workspace, size_t workspace_bytes, float** fm1,
                                    float** fm2, float** fm3) {
    if (int rc = check_common("y3_net_train_forward", net, n, h, w)) return rc;
    if (int rc = check_vars("y3_net_train_forward", net, vars)) return rc;
    Y3_CHECK_ARG(x && opts && workspace, "y3_net_train_forward: null argument");
    Y3_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "y3_net_train_forward: workspace must be 256-byte aligned");
    if (!net->ctx) { y3_set_error("y3_net_train_forward: the net was created without a context"); return Y3_ESTATE; }
    // Refuse a workspace that is too small BEFORE anything is launched: the dry run (host only, a few microseconds per layer)
    // replays the allocation sequence of forward + loss + backward for THIS dtype, shape and variable table.  (The arena's
    // own overflow flag is only read after a pass has been enqueued - too late to keep the device from writing past the end.)
    {
        const size_t need = y3_net_train_workspace_bytes(net, vars, n, h, w);
        if (need == 0 || workspace_bytes < need) {
            y3_set_error("y3_net_train_forward: workspace too small (%zu bytes given, %zu needed for this dtype / shape / variable table)",
                         workspace_bytes, need);
            return Y3_EINVAL;
        }
    }
    if (int rc = forward_impl(net, vars, x, n, h, w, opts, workspace, workspace_bytes, false)) return rc;
    y3_train_state& S = *net->train;
    float** out[3] = {fm1, fm2, fm3};
    for (int i = 0; i < 3; ++i)
        if (out[i]) *out[i] = S.arena.p(S.tens[S.fm_tensor[i]]);
    return Y3_OK;
}

extern "C" int y3_net_train_loss(y3_net* net, const float* y_true_1, const float* y_true_2, const float* y_true_3,
                                 const y3_train_opts* opts, float* loss5) {
    Y3_CHECK_ARG(net && y_true_1 && y_true_2 && y_true_3 && opts && opts->anchors, "y3_net_train_loss: null argument");
    if (!net->train || !net->train->have_fwd) { y3_set_error("y3_net_train_loss: call y3_net_train_forward first"); return Y3_ESTATE; }
    const float* yt[3] = {y_true_1, y_true_2, y_true_3};
    return loss_impl(net, yt, opts, loss5, false);
}

extern "C" int y3_net_train_backward(y3_net* net, const y3_train_var* vars, float* flat_grad, y3_grad_ready_fn ready,
                                     void* user) {
    Y3_CHECK_ARG(net && flat_grad, "y3_net_train_backward: null argument");
    if (int rc = check_vars("y3_net_train_backward", net, vars)) return rc;
    if (!net->train || !net->train->have_fwd || !net->train->have_loss) {
        y3_set_error("y3_net_train_backward: call y3_net_train_forward and y3_net_train_loss first");
        return Y3_ESTATE;
    }
    return backward_impl(net, vars, flat_grad, ready, user, false);
}

extern "C" int y3_net_train_step(y3_net* net, const y3_train_var* vars, const float* x, int n, int h, int w,
                                 const float* y_true_1, const float* y_true_2, const float* y_true_3,
                                 const y3_train_opts* opts, float* flat_grad, void* workspace, size_t workspace_bytes,
                                 float* loss5, y3_grad_ready_fn ready, void* user) {
    if (int rc = y3_net_train_forward(net, vars, x, n, h, w, opts, workspace, workspace_bytes, nullptr, nullptr, nullptr)) return rc;
    if (int rc = y3_net_train_loss(net, y_true_1, y_true_2, y_true_3, opts, loss5)) return rc;
    return y3_net_train_backward(net, vars, flat_grad, ready, user);
}

// offsets (bytes into the workspace of the last forward) of the tensors that fix layer i's LeakyReLU branches: the raw conv
// output z [n,ho,wo,cout] and the [4][cout] mean / inv_std / folded scale / folded shift (test hook of tests/test_train_gpu.py)
extern "C" int y3_net_train_saved(const y3_net* net, int layer, size_t* z_offset, size_t* stats_offset) {
    Y3_CHECK_ARG(net && net->train && net->train->have_fwd && layer >= 0 && layer < (int)net->layers.size(),
                 "y3_net_train_saved: no forward state for layer %d", layer);
    const y3_train_state& S = *net->train;
    if (z_offset) *z_offset = S.z[layer].off;
    if (stats_offset) *stats_offset = S.stats[layer].off;
    return Y3_OK;
}
