// Host side of the C ABI: error channel, contexts, the conv entry point and the y3_net graph
// (yolov3.forward, model.py:30-80 of the reference, as a fixed launch plan over caller-owned buffers).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "y3_internal.h"
#include "y3_net.h"

static thread_local char g_err[512] = "";

void y3_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* y3_last_error(void) { return g_err; }

int y3_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= Y3_MAX_DEVICES) return -1;
    return dev;
}

size_t y3_device_max_lds() {
    static size_t cached[Y3_MAX_DEVICES] = {};       // benign race (idempotent)
    const int dev = y3_current_device();
    if (dev >= 0 && cached[dev]) return cached[dev];
    // (the per-block figure is the 64 KB default of static allocations on some runtimes; what a kernel may ask for with
    // hipFuncAttributeMaxDynamicSharedMemorySize is bounded by the CU's LDS: take the largest of the three answers)
    int d = 0, v = 0;
    if (hipGetDevice(&d) != hipSuccess) return 0;
    for (hipDeviceAttribute_t a : {hipDeviceAttributeMaxSharedMemoryPerBlock, hipDeviceAttributeSharedMemPerBlockOptin,
                                   hipDeviceAttributeMaxSharedMemoryPerMultiprocessor}) {
        int q = 0;
        if (hipDeviceGetAttribute(&q, a, d) == hipSuccess && q > v) v = q;
    }
    if (v <= 0) return 0;
    if (dev >= 0) cached[dev] = (size_t)v;
    return (size_t)v;
}
extern "C" int y3_abi_version(void) { return Y3_ABI_VERSION; }

extern "C" int y3_ctx_create(int device, void* stream, y3_ctx** out) {
    Y3_CHECK_ARG(out, "y3_ctx_create: null out pointer");
    int count = 0;
    Y3_CHECK_HIP(hipGetDeviceCount(&count));
    Y3_CHECK_ARG(device >= 0 && device < count, "y3_ctx_create: device %d out of range (have %d)", device,
                 count);
    Y3_CHECK_HIP(hipSetDevice(device));
    void* err = nullptr;
    Y3_CHECK_HIP(hipHostMalloc(&err, 64, hipHostMallocMapped));   // pinned + device-visible: the context's error word
    y3_ctx* c = new y3_ctx;
    c->device = device;
    c->stream = static_cast<hipStream_t>(stream);
    c->err_host = static_cast<unsigned*>(err);
    *c->err_host = 0u;
    *out = c;
    return Y3_OK;
}

extern "C" int y3_ctx_destroy(y3_ctx* ctx) {
    if (ctx) {
        if (ctx->err_host) (void)hipHostFree(ctx->err_host);
        if (ctx->stage_ev) { (void)hipEventSynchronize(ctx->stage_ev); (void)hipEventDestroy(ctx->stage_ev); }
        if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
        delete ctx;
    }
    return Y3_OK;
}

int y3_ctx_stage_acquire(y3_ctx* ctx, size_t bytes, void** out) {
    if (ctx->stage_busy) {                          // the previous upload may still be reading the buffer
        Y3_CHECK_HIP(hipEventSynchronize(ctx->stage_ev));
        ctx->stage_busy = false;
    }
    if (ctx->stage_bytes < bytes) {
        if (ctx->stage_host) Y3_CHECK_HIP(hipHostFree(ctx->stage_host));
        ctx->stage_host = nullptr;
        ctx->stage_bytes = 0;
        const size_t cap = (bytes + 4095) & ~(size_t)4095;
        Y3_CHECK_HIP(hipHostMalloc(&ctx->stage_host, cap, hipHostMallocDefault));
        ctx->stage_bytes = cap;
    }
    if (!ctx->stage_ev) Y3_CHECK_HIP(hipEventCreateWithFlags(&ctx->stage_ev, hipEventDisableTiming));
    *out = ctx->stage_host;
    return Y3_OK;
}

int y3_ctx_stage_release(y3_ctx* ctx) {
    Y3_CHECK_HIP(hipEventRecord(ctx->stage_ev, ctx->stream));
    ctx->stage_busy = true;
    return Y3_OK;
}

// Fault injection for the loud-time-out test (include/yolo355.h, y3_debug_streamk_fault): an explicit call, not an
// environment variable - nothing outside the process can switch it on.
static int g_sk_fault = 0;
extern "C" void y3_debug_streamk_fault(int on) { __atomic_store_n(&g_sk_fault, on ? 1 : 0, __ATOMIC_RELAXED); }
void y3_sk_debug_env(unsigned* spin_limit, int* fault) {
    *fault = __atomic_load_n(&g_sk_fault, __ATOMIC_RELAXED);
    *spin_limit = *fault ? (1u << 10) : (1u << 22);
}

static const char* kStreamKTimeout =
    "a stream-K hand-off timed out in an earlier launch on this context (a consumer workgroup gave up waiting for a "
    "partial sum): the output of that launch is INVALID; y3_ctx_check clears the condition";

// Sticky device-side failure of an earlier launch (no synchronisation: reads the pinned word as it is now).
static int ctx_pending_error(const y3_ctx* ctx) {
    if (ctx && ctx->err_host && __atomic_load_n(ctx->err_host, __ATOMIC_RELAXED) != 0u) {
        y3_set_error("%s", kStreamKTimeout);
        return Y3_EHIP;
    }
    return Y3_OK;
}
#define Y3_CHECK_CTX(ctx, who)                                   \
    do {                                                         \
        Y3_CHECK_ARG(ctx, who ": null context");                 \
        if (int rc_ = ctx_pending_error(ctx)) return rc_;        \
    } while (0)

extern "C" int y3_ctx_check(y3_ctx* ctx) {
    Y3_CHECK_ARG(ctx, "y3_ctx_check: null context");
    Y3_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const unsigned code = __atomic_exchange_n(ctx->err_host, 0u, __ATOMIC_RELAXED);
    if (code != 0u) {
        y3_set_error("%s (code %u)", kStreamKTimeout, code);
        return Y3_EHIP;
    }
    return Y3_OK;
}

extern "C" size_t y3_conv_workspace_bytes(const y3_conv_desc* d) { return y3_conv_workspace_bytes_impl(d); }

extern "C" int y3_streamk_range(int kind, int units, int ksteps, int workers, int group, int local_worker,
                                long long* begin, long long* end) {
    return y3_streamk_range_impl(kind, units, ksteps, workers, group, local_worker, begin, end);
}

extern "C" int y3_conv2d_fwd(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* x_up,
                             const float* w, const float* scale, const float* shift,
                             const float* residual, float* y, void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd");
    y3_sk_opts o;
    o.err = ctx->err_host;
    return y3_launch_conv(ctx->stream, d, x, x_up, w, scale, shift, residual, y, workspace, workspace_bytes, &o);
}

// ---- training forward: the conv + the batch-norm statistics of its output in one pass ---------------------------------
extern "C" int y3_conv_stats_blocks(const y3_conv_desc* d, int wino) { return y3_conv_stats_blocks_impl(d, wino); }

extern "C" int y3_conv2d_fwd_stats(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w,
                                   const float* scale, const float* shift, float* y, float* stats, void* workspace,
                                   size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd_stats");
    Y3_CHECK_ARG(stats && d && y3_conv_stats_blocks_impl(d, 0) > 0,
                 "y3_conv2d_fwd_stats: null stats, or a conv without statistics support (y3_conv_stats_blocks == 0)");
    y3_sk_opts o;
    o.err = ctx->err_host;
    o.stats = stats;
    return y3_launch_conv(ctx->stream, d, x, nullptr, w, scale, shift, nullptr, y, workspace, workspace_bytes, &o);
}

extern "C" int y3_conv2d_fwd_wino_stats(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino,
                                        const float* scale, const float* shift, float* y, float* stats, void* workspace,
                                        size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd_wino_stats");
    Y3_CHECK_ARG(stats && d && y3_conv_stats_blocks_impl(d, 1) > 0,
                 "y3_conv2d_fwd_wino_stats: null stats, or a conv the Winograd kernel does not take");
    y3_sk_opts o;
    o.err = ctx->err_host;
    o.stats = stats;
    return y3_launch_conv_wino(ctx->stream, d, x, w_wino, scale, shift, nullptr, y, workspace, workspace_bytes, &o);
}

extern "C" int y3_pack_conv_weights_split(y3_ctx* ctx, const float* w_hwio, int k, int cin, int cout, int planes,
                                          void* w_split) {
    Y3_CHECK_ARG(ctx && w_hwio && w_split, "y3_pack_conv_weights_split: null argument");
    Y3_CHECK_ARG((k == 1 || k == 3) && cin > 0 && cout > 0 && cin % 16 == 0,
                 "y3_pack_conv_weights_split: k must be 1 or 3 and cin a positive multiple of 16");
    Y3_CHECK_ARG(planes == 2 || planes == 3, "y3_pack_conv_weights_split: planes must be 2 or 3 (got %d)", planes);
    return y3_launch_pack_split(ctx->stream, w_hwio, k, cin, cout, planes, w_split);
}

extern "C" int y3_conv2d_fwd_split(y3_ctx* ctx, const y3_conv_desc* d, int planes, const float* x,
                                   const float* x_up, const void* w_split, const float* scale, const float* shift,
                                   const float* residual, float* y, void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd_split");
    y3_sk_opts o;
    o.err = ctx->err_host;
    return y3_launch_conv_split(ctx->stream, d, planes, x, x_up, w_split, scale, shift, residual, y, workspace,
                                workspace_bytes, &o);
}

extern "C" int y3_conv_wino_eligible(const y3_conv_desc* d) { return y3_conv_wino_eligible_impl(d); }

extern "C" int y3_pack_conv_weights_wino(y3_ctx* ctx, const float* w_hwio, int cin, int cout, float* w_wino) {
    Y3_CHECK_ARG(ctx && w_hwio && w_wino, "y3_pack_conv_weights_wino: null argument");
    Y3_CHECK_ARG(cin > 0 && cout > 0 && cin % 8 == 0, "y3_pack_conv_weights_wino: cin must be a positive multiple of 8");
    return y3_launch_pack_wino(ctx->stream, w_hwio, cin, cout, w_wino);
}

extern "C" size_t y3_conv_wino_workspace_bytes(const y3_conv_desc* d) { return y3_conv_wino_workspace_bytes_impl(d); }

extern "C" int y3_conv2d_fwd_wino(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino,
                                  const float* scale, const float* shift, const float* residual, float* y,
                                  void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd_wino");
    y3_sk_opts o;
    o.err = ctx->err_host;
    return y3_launch_conv_wino(ctx->stream, d, x, w_wino, scale, shift, residual, y, workspace, workspace_bytes, &o);
}

extern "C" int y3_conv_wino44_eligible(const y3_conv_desc* d) { return y3_conv_wino44_eligible_impl(d); }
extern "C" int y3_conv_wino44_candidate(const y3_conv_desc* d) { return y3_conv_wino44_candidate_impl(d); }
extern "C" int y3_conv_wino44_preferred(const y3_conv_desc* d) { return y3_conv_wino44_preferred_impl(d); }

extern "C" int y3_pack_conv_weights_wino44(y3_ctx* ctx, const float* w_hwio, int cin, int cout, float* w_wino44) {
    Y3_CHECK_ARG(ctx && w_hwio && w_wino44, "y3_pack_conv_weights_wino44: null argument");
    Y3_CHECK_ARG(cin > 0 && cout > 0 && cin % 8 == 0, "y3_pack_conv_weights_wino44: cin must be a positive multiple of 8");
    return y3_launch_pack_wino44(ctx->stream, w_hwio, cin, cout, w_wino44);
}

extern "C" size_t y3_conv_wino44_workspace_bytes(const y3_conv_desc* d) { return y3_conv_wino44_workspace_bytes_impl(d); }

extern "C" int y3_conv2d_fwd_wino44(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino44,
                                    const float* scale, const float* shift, const float* residual, float* y,
                                    void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd_wino44");
    y3_sk_opts o;
    o.err = ctx->err_host;
    return y3_launch_conv_wino44(ctx->stream, d, x, w_wino44, scale, shift, residual, y, workspace, workspace_bytes, &o);
}

extern "C" int y3_conv2d_fwd_wino44_stats(y3_ctx* ctx, const y3_conv_desc* d, const float* x, const float* w_wino44,
                                          const float* scale, const float* shift, float* y, float* stats, void* workspace,
                                          size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_fwd_wino44_stats");
    Y3_CHECK_ARG(stats && d && y3_conv_stats_blocks_impl(d, 2) > 0,
                 "y3_conv2d_fwd_wino44_stats: null stats, or a conv the F(4x4,3x3) kernel does not take");
    y3_sk_opts o;
    o.err = ctx->err_host;
    o.stats = stats;
    return y3_launch_conv_wino44(ctx->stream, d, x, w_wino44, scale, shift, nullptr, y, workspace, workspace_bytes, &o);
}

extern "C" int y3_pack_conv_weights_wino44_dgrad(y3_ctx* ctx, const float* w_d, int cin, int dz_stride, float* w_wino44_d) {
    Y3_CHECK_ARG(ctx && w_d && w_wino44_d, "y3_pack_conv_weights_wino44_dgrad: null argument");
    Y3_CHECK_ARG(cin > 0 && dz_stride > 0 && dz_stride % 8 == 0,
                 "y3_pack_conv_weights_wino44_dgrad: dz_stride must be a positive multiple of 8");
    return y3_launch_pack_wino44(ctx->stream, w_d, dz_stride, cin, w_wino44_d, 1);
}

extern "C" int y3_conv2d_dgrad_wino44(y3_ctx* ctx, const y3_conv_desc* fwd, const float* dz, int dz_stride,
                                      const float* w_wino44_d, const float* ones, const float* zeros, int accumulate,
                                      float* dx, void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_dgrad_wino44");
    Y3_CHECK_ARG(fwd && dz && w_wino44_d && ones && zeros && dx, "y3_conv2d_dgrad_wino44: null pointer argument");
    Y3_CHECK_ARG(fwd->k == 3 && fwd->stride == 1 && fwd->c_up == 0 && dz_stride >= fwd->cout,
                 "y3_conv2d_dgrad_wino44: needs a 3x3 stride-1 conv and dz_stride >= Cout");
    y3_conv_desc g = *fwd;
    g.cin = dz_stride; g.cout = fwd->cin; g.act = 0;
    Y3_CHECK_ARG(y3_conv_wino44_eligible_impl(&g), "y3_conv2d_dgrad_wino44: needs dz_stride %% 32 == 0 and Cin %% 64 == 0");
    y3_sk_opts o;
    o.err = ctx->err_host;
    // accumulate: dx is read as the residual and written by the same thread for the same element (no cross-thread hazard)
    return y3_launch_conv_wino44(ctx->stream, &g, dz, w_wino44_d, ones, zeros, accumulate ? dx : nullptr, dx, workspace,
                                 workspace_bytes, &o);
}

// Data gradient of a stride-1 3x3 conv in its Winograd form: dx (+)= conv_same(dz, flipped / channel-swapped kernel) is
// itself a stride-1 3x3 SAME conv [n,h,w,dz_stride] -> [n,h,w,cin], so the forward Winograd kernel runs it unchanged.
static int wino_dgrad_desc(const y3_conv_desc* fwd, int dz_stride, y3_conv_desc* g) {
    Y3_CHECK_ARG(fwd && fwd->k == 3 && fwd->stride == 1 && fwd->c_up == 0, "y3_conv2d_dgrad_wino: needs a 3x3 stride-1 conv");
    Y3_CHECK_ARG(dz_stride >= fwd->cout && dz_stride % 32 == 0 && fwd->cin % 32 == 0,
                 "y3_conv2d_dgrad_wino: dz stride and Cin must be multiples of 32");
    *g = *fwd;
    g->cin = dz_stride; g->cout = fwd->cin; g->act = 0;
    Y3_CHECK_ARG(y3_conv_wino_eligible_impl(g), "y3_conv2d_dgrad_wino: shape not eligible for the Winograd kernel");
    return Y3_OK;
}

extern "C" int y3_pack_conv_weights_wino_dgrad(y3_ctx* ctx, const float* w_d, int cin, int dz_stride, float* w_wino_d) {
    Y3_CHECK_ARG(ctx && w_d && w_wino_d, "y3_pack_conv_weights_wino_dgrad: null argument");
    Y3_CHECK_ARG(cin > 0 && dz_stride > 0 && dz_stride % 8 == 0,
                 "y3_pack_conv_weights_wino_dgrad: dz_stride must be a positive multiple of 8");
    return y3_launch_pack_wino(ctx->stream, w_d, dz_stride, cin, w_wino_d, 1);
}

extern "C" int y3_conv2d_dgrad_wino(y3_ctx* ctx, const y3_conv_desc* fwd, const float* dz, int dz_stride,
                                    const float* w_wino_d, const float* ones, const float* zeros, int accumulate,
                                    float* dx, void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_dgrad_wino");
    Y3_CHECK_ARG(dz && w_wino_d && ones && zeros && dx, "y3_conv2d_dgrad_wino: null pointer argument");
    y3_conv_desc g;
    if (int rc = wino_dgrad_desc(fwd, dz_stride, &g)) return rc;
    y3_sk_opts o;
    o.err = ctx->err_host;
    // accumulate: dx is read as the residual and written by the same thread of the same tile (no cross-thread hazard)
    return y3_launch_conv_wino(ctx->stream, &g, dz, w_wino_d, ones, zeros, accumulate ? dx : nullptr, dx, workspace,
                               workspace_bytes, &o);
}

extern "C" int y3_pack_conv_weights_split_dgrad(y3_ctx* ctx, const float* w_d, int k, int cin, int dz_stride,
                                                int planes, void* w_split) {
    Y3_CHECK_ARG(ctx && w_d && w_split, "y3_pack_conv_weights_split_dgrad: null argument");
    Y3_CHECK_ARG((k == 1 || k == 3) && cin > 0 && dz_stride > 0 && dz_stride % 16 == 0,
                 "y3_pack_conv_weights_split_dgrad: k must be 1 or 3 and dz_stride a positive multiple of 16");
    Y3_CHECK_ARG(planes == 2 || planes == 3, "y3_pack_conv_weights_split_dgrad: planes must be 2 or 3 (got %d)", planes);
    // the gradient conv's K axis is dz_stride, its output axis the forward cin: read [k*k][cin][dz_stride] transposed
    return y3_launch_pack_split(ctx->stream, w_d, k, dz_stride, cin, planes, w_split, 1);
}

extern "C" int y3_conv2d_dgrad_split(y3_ctx* ctx, const y3_conv_desc* fwd, int planes, const float* dz, int dz_stride,
                                     const void* w_split_d, const float* ones, const float* zeros, int accumulate,
                                     float* dx, void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_dgrad_split");
    y3_sk_opts o;
    o.err = ctx->err_host;
    return y3_launch_conv_dgrad_split(ctx->stream, fwd, planes, dz, dz_stride, w_split_d, ones, zeros, accumulate, dx,
                                      workspace, workspace_bytes, &o);
}

extern "C" int y3_conv2d_dgrad(y3_ctx* ctx, const y3_conv_desc* fwd, const float* dz, int dz_stride,
                               const float* w_d, const float* ones, const float* zeros, int accumulate,
                               float* dx, void* workspace, size_t workspace_bytes) {
    Y3_CHECK_CTX(ctx, "y3_conv2d_dgrad");
    y3_sk_opts o;
    o.err = ctx->err_host;
    return y3_launch_conv_dgrad(ctx->stream, fwd, dz, dz_stride, w_d, ones, zeros, accumulate, dx, workspace,
                                workspace_bytes, &o);
}

// ------------------------------------------------------------------------------------------------
// y3_net: the 75-conv graph.  Tensor ids: 0 = network input; 1.. = conv outputs in creation order.
// ------------------------------------------------------------------------------------------------
// (struct Tensor / Layer / y3_net: y3_net.h)

extern "C" int y3_net_create(y3_ctx* ctx, int class_num, y3_net** out) {
    // ctx may be NULL: the graph, layer table and workspace plan are host-only; forward then needs a ctx.
    Y3_CHECK_ARG(out, "y3_net_create: null out pointer");
    Y3_CHECK_ARG(class_num > 0, "y3_net_create: class_num must be positive");
    y3_net* net = new y3_net;
    net->ctx = ctx;
    net->class_num = class_num;
    net->build();
    *out = net;
    return Y3_OK;
}

extern "C" int y3_net_set_dtype(y3_net* net, int dtype) {
    Y3_CHECK_ARG(net, "y3_net_set_dtype: null net");
    Y3_CHECK_ARG(dtype >= 0 && dtype <= 4,
                 "y3_net_set_dtype: dtype must be 0 (fp32), 1 (bf16), 2 (fp32 via bf16x6), 3 (fp32 via bf16x3) or "
                 "4 (fp32, Winograd for the eligible 3x3 convs)");
    net->dtype = dtype;
    net->pn = net->ph = net->pw = 0;   // re-plan
    return Y3_OK;
}

extern "C" int y3_net_destroy(y3_net* net) {
    if (net) {
        for (auto& set : net->event_sets)
            for (hipEvent_t e : set) (void)hipEventDestroy(e);
        if (net->train) y3_train_state_free(net->train);
        if (net->own_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(net->own_stream));
        delete net;
    }
    return Y3_OK;
}

extern "C" int y3_net_num_layers(const y3_net* net) { return net ? (int)net->layers.size() : 0; }

extern "C" int y3_net_layer_info(const y3_net* net, int i, int* k, int* stride, int* cin, int* cout,
                                 int* has_bn) {
    Y3_CHECK_ARG(net && i >= 0 && i < (int)net->layers.size(), "y3_net_layer_info: bad layer index %d", i);
    const Layer& l = net->layers[i];
    if (k) *k = l.k;
    if (stride) *stride = l.stride;
    if (cin) *cin = l.cin;
    if (cout) *cout = l.cout;
    if (has_bn) *has_bn = l.bn;
    return Y3_OK;
}

extern "C" int y3_net_layer_graph(const y3_net* net, int i, int* src, int* up, int* resid, int* dst, int* act) {
    Y3_CHECK_ARG(net && i >= 0 && i < (int)net->layers.size(), "y3_net_layer_graph: bad layer index %d", i);
    const Layer& l = net->layers[i];
    if (src) *src = l.src;
    if (up) *up = l.up;
    if (resid) *resid = l.resid;
    if (dst) *dst = l.dst;
    if (act) *act = l.act;
    return Y3_OK;
}

extern "C" int y3_net_num_tensors(const y3_net* net) { return net ? (int)net->tensors.size() : 0; }

extern "C" int y3_net_tensor_info(const y3_net* net, int t, int* channels, int* sdiv, int* ext) {
    Y3_CHECK_ARG(net && t >= 0 && t < (int)net->tensors.size(), "y3_net_tensor_info: bad tensor id %d", t);
    if (channels) *channels = net->tensors[t].c;
    if (sdiv) *sdiv = net->tensors[t].sdiv;
    if (ext) *ext = net->tensors[t].ext;
    return Y3_OK;
}

extern "C" int y3_net_set_layer(y3_net* net, int i, const float* w_packed, const float* scale,
                                const float* shift) {
    Y3_CHECK_ARG(net && i >= 0 && i < (int)net->layers.size(), "y3_net_set_layer: bad layer index %d", i);
    Y3_CHECK_ARG(w_packed && scale && shift, "y3_net_set_layer: null parameter pointer");
    Layer& l = net->layers[i];
    l.w = w_packed; l.scale = scale; l.shift = shift;
    return Y3_OK;
}

extern "C" int y3_net_set_layer_alt(y3_net* net, int i, const float* w_wino44) {
    Y3_CHECK_ARG(net && i >= 0 && i < (int)net->layers.size(), "y3_net_set_layer_alt: bad layer index %d", i);
    net->layers[i].w_alt = w_wino44;       // NULL: the layer runs on its y3_net_set_layer packing at every size
    net->pn = net->ph = net->pw = 0;       // re-plan: the layer's V scratch (two-kernel form) is part of the workspace
    return Y3_OK;
}

static int check_size(const char* who, int n, int h, int w) {
    Y3_CHECK_ARG(n > 0, "%s: batch must be positive", who);
    Y3_CHECK_ARG(h > 0 && w > 0 && h % 32 == 0 && w % 32 == 0,
                 "%s: input size must be a positive multiple of 32 (got %dx%d)", who, h, w);
    return Y3_OK;
}

extern "C" size_t y3_net_workspace_bytes(const y3_net* net, int n, int h, int w) {
    if (!net || check_size("y3_net_workspace_bytes", n, h, w) != Y3_OK) return 0;
    const_cast<y3_net*>(net)->plan(n, h, w);
    return net->plan_bytes;
}

// Layers that exist to move bytes run fused with their only reader (round 5):
//   bf16 storage: the stem and the stride-2 conv behind it are ONE kernel when nothing else reads the stem's output
//   (y3_conv_bf16s.hip: the 378 MB tensor between them at 608x608, bs=16 never exists); fp32 (dtypes 0 and 4 - both run these
//   two layers on the direct kernels -): the same fusion, y3_conv_f32s.hip;
static bool net_fuses_01(const y3_net* net, int n, int h, int w) {
    const size_t nl = net->layers.size();
    const bool f32_direct01 = net->dtype == 0 || net->dtype == 4;
    if (!((net->dtype == 1 || f32_direct01) && nl >= 2)) return false;
    const Layer &l0 = net->layers[0], &l1 = net->layers[1];
    y3_conv_desc d0 = {n, h, w, l0.cin, l0.c_up, l0.cout, l0.k, l0.stride, l0.act};
    y3_conv_desc d1 = {n, h / net->tensors[l1.src].sdiv, w / net->tensors[l1.src].sdiv, l1.cin, l1.c_up, l1.cout, l1.k,
                       l1.stride, l1.act};
    return l0.src == 0 && l1.src == l0.dst && net->tensors[l0.dst].last_use == 1 && net->tensors[l0.dst].ext < 0 &&
           l0.resid < 0 && l1.resid < 0 && net->tensors[l1.dst].ext < 0 &&
           (net->dtype == 1 ? y3_conv_bf16_stem_s2_takes(&d0, &d1) : y3_conv_f32_stem_s2_takes(&d0, &d1)) == 1;
}
//   ... and, bf16 only, the first residual block (layers 2 and 3: 1x1 64 -> 32, 3x3 32 -> 64 + shortcut) likewise (y3_conv_bf16b.hip).
static bool net_fuses_23(const y3_net* net, int n, int h, int w) {
    if (!(net->dtype == 1 && net->layers.size() >= 4)) return false;
    const Layer &l2 = net->layers[2], &l3 = net->layers[3];
    const int sd = net->tensors[l2.src].sdiv;
    y3_conv_desc d2 = {n, h / sd, w / sd, l2.cin, l2.c_up, l2.cout, l2.k, l2.stride, l2.act};
    y3_conv_desc d3 = {n, h / net->tensors[l3.src].sdiv, w / net->tensors[l3.src].sdiv, l3.cin, l3.c_up, l3.cout, l3.k,
                       l3.stride, l3.act};
    return l3.src == l2.dst && l3.resid == l2.src && l2.resid < 0 && l2.src != 0 && net->tensors[l2.dst].last_use == 3 &&
           net->tensors[l2.dst].ext < 0 && net->tensors[l3.dst].ext < 0 && net->tensors[l2.src].ext < 0 &&
           y3_conv_bf16_resblock64_takes(&d2, &d3) == 1;
}

// 0: layer i has its own launch; 1: it runs inside the NEXT layer's launch (its output tensor never exists; its profiled time
// is 0); 2: its launch also runs the layer before it.
extern "C" int y3_net_layer_fused(const y3_net* net, int i, int n, int h, int w) {
    if (!net || i < 0 || i >= (int)net->layers.size() || n <= 0 || h <= 0 || w <= 0) return 0;
    if (i <= 1 && net_fuses_01(net, n, h, w)) return i == 0 ? 1 : 2;
    if ((i == 2 || i == 3) && net_fuses_23(net, n, h, w)) return i == 2 ? 1 : 2;
    return 0;
}

extern "C" int y3_net_forward(y3_net* net, const float* x, int n, int h, int w, void* workspace,
                              size_t workspace_bytes, float* fm1, float* fm2, float* fm3) {
    Y3_CHECK_ARG(net && x && workspace && fm1 && fm2 && fm3, "y3_net_forward: null argument");
    if (!net->ctx) {
        y3_set_error("y3_net_forward: the net was created without a context");
        return Y3_ESTATE;
    }
    if (int rc = ctx_pending_error(net->ctx)) return rc;
    if (int rc = check_size("y3_net_forward", n, h, w)) return rc;
    net->plan(n, h, w);
    Y3_CHECK_ARG(workspace_bytes >= net->plan_bytes, "y3_net_forward: workspace too small (%zu < %zu)",
                 workspace_bytes, net->plan_bytes);
    Y3_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "y3_net_forward: workspace must be 256-byte aligned");
    for (size_t i = 0; i < net->layers.size(); ++i)
        if (!net->layers[i].w) {
            y3_set_error("y3_net_forward: layer %zu has no parameters (call y3_net_set_layer)", i);
            return Y3_ESTATE;
        }
    float* ext[3] = {fm1, fm2, fm3};
    char* base = static_cast<char*>(workspace);
    auto ptr = [&](int id) -> float* {
        if (id < 0) return nullptr;
        if (id == 0) return const_cast<float*>(x);
        const Tensor& t = net->tensors[id];
        if (t.ext >= 0) return ext[t.ext];
        return reinterpret_cast<float*>(base + net->offsets[id]);
    };
    hipStream_t st = net->ctx->stream;
    const size_t nl = net->layers.size();
    hipEvent_t* ev = nullptr;
    if (net->profiling && net->sets_used < 256) {
        if (net->sets_used == net->event_sets.size()) {
            std::vector<hipEvent_t> set(nl + 1, nullptr);   // nl+1 layer boundaries
            for (size_t i = 0; i < set.size(); ++i) Y3_CHECK_HIP(hipEventCreate(&set[i]));
            net->event_sets.push_back(set);
        }
        ev = net->event_sets[net->sets_used++].data();
        Y3_CHECK_HIP(hipEventRecord(ev[0], st));
    }
    // every stream-K layer polls its own pre-zeroed flag region: ONE memset per forward instead of one per launch
    unsigned* flag_base = reinterpret_cast<unsigned*>(base + net->arena_bytes + net->scratch_bytes);
    if (net->flags_bytes) Y3_CHECK_HIP(hipMemsetAsync(flag_base, 0, net->flags_bytes, st));
    const bool fuse01 = net_fuses_01(net, n, h, w), fuse23 = net_fuses_23(net, n, h, w);
    for (size_t i = 0; i < nl; ++i) {
        const Layer& l = net->layers[i];
        const Tensor& in = net->tensors[l.src];
        y3_conv_desc d;
        d.n = n; d.h = h / in.sdiv; d.w = w / in.sdiv;
        d.cin = l.cin; d.c_up = l.c_up; d.cout = l.cout; d.k = l.k; d.stride = l.stride; d.act = l.act;
        y3_sk_opts o;
        o.err = net->ctx->err_host;
        o.flags = net->flags_bytes ? flag_base + i * y3_net::FLAG_WORDS : nullptr;
        if (fuse23 && (i == 2 || i == 3)) {
            int rc = Y3_OK;
            if (i == 3) {
                const Layer& l2 = net->layers[2];
                const int sd = net->tensors[l2.src].sdiv;
                rc = y3_launch_conv_bf16_resblock64(st, n, h / sd, w / sd, ptr(l2.src), l2.w, l2.scale, l2.shift, l2.act, l.w,
                                                    l.scale, l.shift, l.act, ptr(l.dst));
            }
            if (rc != Y3_OK) return rc;
            if (ev) Y3_CHECK_HIP(hipEventRecord(ev[i + 1], st));
            continue;
        }
        if (fuse01 && i <= 1) {
            int rc = Y3_OK;
            if (i == 1) {
                const Layer& l0 = net->layers[0];
                rc = net->dtype == 1
                    ? y3_launch_conv_bf16_stem_s2(st, n, h, w, x, l0.w, l0.scale, l0.shift, l0.act, l.w, l.scale, l.shift, l.act,
                                                  ptr(l.dst))
                    : y3_launch_conv_f32_stem_s2(st, n, h, w, x, l0.w, l0.scale, l0.shift, l0.act, l.w, l.scale, l.shift, l.act,
                                                 ptr(l.dst));
            }
            if (rc != Y3_OK) return rc;
            if (ev) Y3_CHECK_HIP(hipEventRecord(ev[i + 1], st));
            continue;
        }
        const int rc = net->dtype == 1
            ? y3_launch_conv_bf16(st, &d, ptr(l.src), ptr(l.up), l.w, l.scale, l.shift, ptr(l.resid), ptr(l.dst),
                                  net->tensors[l.dst].ext >= 0 ? 1 : 0)
            : (net->dtype == 4 && l.w_alt && y3_conv_wino44_preferred_impl(&d))
            ? y3_launch_conv_wino44(st, &d, ptr(l.src), l.w_alt, l.scale, l.shift, ptr(l.resid), ptr(l.dst),
                                    base + net->arena_bytes, net->scratch_bytes, &o)
            : (net->dtype == 4 && y3_conv_wino_eligible_impl(&d))
            ? y3_launch_conv_wino(st, &d, ptr(l.src), l.w, l.scale, l.shift, ptr(l.resid), ptr(l.dst),
                                  base + net->arena_bytes, net->scratch_bytes, &o)
            : (net->dtype == 2 || net->dtype == 3)
            ? y3_launch_conv_split(st, &d, net->dtype == 2 ? 3 : 2, ptr(l.src), ptr(l.up), l.w, l.scale, l.shift,
                                   ptr(l.resid), ptr(l.dst), base + net->arena_bytes, net->scratch_bytes, &o)
            : y3_launch_conv(st, &d, ptr(l.src), ptr(l.up), l.w, l.scale, l.shift, ptr(l.resid),
                             ptr(l.dst), base + net->arena_bytes, net->scratch_bytes, &o);
        if (rc != Y3_OK) return rc;
        if (ev) Y3_CHECK_HIP(hipEventRecord(ev[i + 1], st));
    }
    return Y3_OK;
}

extern "C" int y3_net_layer_is_streamk(const y3_net* net, int i, int n, int h, int w) {
    if (!net || i < 0 || i >= (int)net->layers.size() || n <= 0 || h <= 0 || w <= 0 || net->dtype == 1) return 0;
    if (y3_net_layer_fused(net, i, n, h, w)) return 0;      // the fused first layers have no stream-K schedule
    const Layer& l = net->layers[i];
    const Tensor& in = net->tensors[l.src];
    y3_conv_desc d;
    d.n = n; d.h = h / in.sdiv; d.w = w / in.sdiv;
    d.cin = l.cin; d.c_up = l.c_up; d.cout = l.cout; d.k = l.k; d.stride = l.stride; d.act = l.act;
    if (net->dtype == 4 && y3_conv_wino_eligible_impl(&d)) return 0;   // one Winograd kernel, no fix-up
    return y3_conv_schedule_impl(&d);
}

extern "C" int y3_net_set_profiling(y3_net* net, int enabled) {
    Y3_CHECK_ARG(net, "y3_net_set_profiling: null net");
    net->profiling = enabled != 0;     // recorded sets are kept until y3_net_get_layer_ms reads (and clears) them
    return Y3_OK;
}

extern "C" int y3_net_get_layer_ms(y3_net* net, float* ms, int count) {
    Y3_CHECK_ARG(net && ms, "y3_net_get_layer_ms: null argument");
    Y3_CHECK_ARG(count == (int)net->layers.size(), "y3_net_get_layer_ms: count must be %zu",
                 net->layers.size());
    if (net->sets_used == 0) {
        y3_set_error("y3_net_get_layer_ms: no profiled forward has run");
        return Y3_ESTATE;
    }
    for (int i = 0; i < count; ++i) ms[i] = 0.f;
    for (size_t s = 0; s < net->sets_used; ++s) {
        hipEvent_t* ev = net->event_sets[s].data();
        Y3_CHECK_HIP(hipEventSynchronize(ev[count]));
        for (int i = 0; i < count; ++i) {
            float t = 0.f;
            Y3_CHECK_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
            ms[i] += t;
        }
    }
    for (int i = 0; i < count; ++i) ms[i] /= (float)net->sets_used;
    net->sets_used = 0;
    return Y3_OK;
}
