/*
 * oracle/nms_ref.c — plain-C CPU restatement of the two NMS definitions on the reference's hot path.
 * TEST INFRASTRUCTURE (see oracle/__init__.py): used by tests/ and bench.py's cpu_baseline only.
 *
 *   tf_nms        : tf.image.non_max_suppression as driven by utils/nms_utils.py:36-39 (third-party
 *                   TensorFlow kernel non_max_suppression_op.cc, restated from its documented algorithm;
 *                   parity unpinned — no TensorFlow here).  IoU on min/max-normalised corners, no +1,
 *                   IoU = 0 for degenerate boxes, suppress when IoU > threshold.
 *   py_nms        : utils/nms_utils.py:51-88, literally: areas without +1 (:68), +1 on the intersection
 *                   w/h (:80-81), survivors are those with ovr <= thresh (:85), keep[:max_boxes] (:88).
 *                   Pinned against the reference's own py_nms/cpu_nms via tests/golden/.
 *   per_class_nms : the per-class drivers gpu_nms (:8-48) / cpu_nms (:91-123): score >= thresh,
 *                   candidates in ascending box order, classes ascending, max_boxes PER CLASS.
 * Order of candidates: (score descending, index ascending) — the determinism rule of SURVEY.md App. B.3.
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, SSE fp32: every op rounds to float).
 */
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int idx; } item_t;

static int cmp_item(const void* a, const void* b) {
    const item_t* x = (const item_t*)a;
    const item_t* y = (const item_t*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

static float fminf_(float a, float b) { return a < b ? a : b; }
static float fmaxf_(float a, float b) { return a > b ? a : b; }

static float tf_iou(const float* bi, const float* bj) {
    const float ymin_i = fminf_(bi[0], bi[2]), xmin_i = fminf_(bi[1], bi[3]);
    const float ymax_i = fmaxf_(bi[0], bi[2]), xmax_i = fmaxf_(bi[1], bi[3]);
    const float ymin_j = fminf_(bj[0], bj[2]), xmin_j = fminf_(bj[1], bj[3]);
    const float ymax_j = fmaxf_(bj[0], bj[2]), xmax_j = fmaxf_(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iymin = fmaxf_(ymin_i, ymin_j), ixmin = fmaxf_(xmin_i, xmin_j);
    const float iymax = fminf_(ymax_i, ymax_j), ixmax = fminf_(xmax_i, xmax_j);
    const float inter = fmaxf_(iymax - iymin, 0.f) * fmaxf_(ixmax - ixmin, 0.f);
    return inter / (area_i + area_j - inter);
}

/* boxes [K][4], scores [K] -> sel[<=max_out] candidate indices in selection order; returns the count */
int tf_nms(const float* boxes, const float* scores, int K, int max_out, float iou_thr, int* sel) {
    item_t* order = (item_t*)malloc(sizeof(item_t) * (size_t)(K > 0 ? K : 1));
    int nsel = 0;
    for (int i = 0; i < K; ++i) { order[i].score = scores[i]; order[i].idx = i; }
    qsort(order, (size_t)K, sizeof(item_t), cmp_item);
    for (int t = 0; t < K && nsel < max_out; ++t) {
        const int i = order[t].idx;
        int keep = 1;
        for (int s = nsel - 1; s >= 0; --s) {          /* most recently selected first */
            if (tf_iou(boxes + 4 * (size_t)i, boxes + 4 * (size_t)sel[s]) > iou_thr) { keep = 0; break; }
        }
        if (keep) sel[nsel++] = i;
    }
    free(order);
    return nsel;
}

int py_nms(const float* boxes, const float* scores, int K, int max_out, float iou_thr, int* sel) {
    item_t* items = (item_t*)malloc(sizeof(item_t) * (size_t)(K > 0 ? K : 1));
    int* order = (int*)malloc(sizeof(int) * (size_t)(K > 0 ? K : 1));
    float* areas = (float*)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    int n = K, nkeep = 0;
    for (int i = 0; i < K; ++i) {
        items[i].score = scores[i]; items[i].idx = i;
        areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    qsort(items, (size_t)K, sizeof(item_t), cmp_item);
    for (int i = 0; i < K; ++i) order[i] = items[i].idx;
    while (n > 0) {
        const int i = order[0];
        if (nkeep < max_out) sel[nkeep] = i;   /* keep[:max_boxes] */
        ++nkeep;
        int m = 0;
        for (int t = 1; t < n; ++t) {
            const int j = order[t];
            const float xx1 = fmaxf_(boxes[4 * i + 0], boxes[4 * j + 0]);
            const float yy1 = fmaxf_(boxes[4 * i + 1], boxes[4 * j + 1]);
            const float xx2 = fminf_(boxes[4 * i + 2], boxes[4 * j + 2]);
            const float yy2 = fminf_(boxes[4 * i + 3], boxes[4 * j + 3]);
            const float w = fmaxf_(0.f, xx2 - xx1 + 1.f);
            const float h = fmaxf_(0.f, yy2 - yy1 + 1.f);
            const float inter = w * h;
            const float ovr = inter / (areas[i] + areas[j] - inter);
            if (ovr <= iou_thr) order[m++] = j;
        }
        n = m;
        if (nkeep >= max_out) break;           /* later picks would be truncated anyway */
    }
    free(items); free(order); free(areas);
    return nkeep < max_out ? nkeep : max_out;
}

/* mode 0 = gpu_nms/tf semantics, 1 = cpu_nms/py semantics.  boxes [B][4], scores [B][C].
 * Outputs sized C*max_boxes; returns the number of detections. */
int per_class_nms(int mode, const float* boxes, const float* scores, int B, int C, int max_boxes,
                  float score_thr, float iou_thr, float* out_boxes, float* out_scores, int* out_labels,
                  int* out_index) {
    float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)B);
    float* cs = (float*)malloc(sizeof(float) * (size_t)B);
    int* cidx = (int*)malloc(sizeof(int) * (size_t)B);
    int* sel = (int*)malloc(sizeof(int) * (size_t)(max_boxes > 0 ? max_boxes : 1));
    int total = 0;
    for (int c = 0; c < C; ++c) {
        int K = 0;
        for (int b = 0; b < B; ++b) {
            const float s = scores[(size_t)b * C + c];
            if (s >= score_thr) {
                memcpy(cb + 4 * (size_t)K, boxes + 4 * (size_t)b, 4 * sizeof(float));
                cs[K] = s; cidx[K] = b; ++K;
            }
        }
        if (K == 0) continue;
        const int ns = mode == 0 ? tf_nms(cb, cs, K, max_boxes, iou_thr, sel)
                                 : py_nms(cb, cs, K, max_boxes, iou_thr, sel);
        for (int t = 0; t < ns; ++t) {
            memcpy(out_boxes + 4 * (size_t)total, cb + 4 * (size_t)sel[t], 4 * sizeof(float));
            out_scores[total] = cs[sel[t]];
            out_labels[total] = c;
            out_index[total] = cidx[sel[t]];
            ++total;
        }
    }
    free(cb); free(cs); free(cidx); free(sel);
    return total;
}
