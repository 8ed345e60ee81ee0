/* yolo355_feed.h - C ABI of liby3feed.so: the per-image CPU work of the feeder.
 *
 * Host code only (g++, no HIP, no device): it runs in the feeder's workers while the GPU runs the train step.  It
 * replaces what the reference does per image through OpenCV inside tf.data's py_func workers:
 *
 *     utils/data_utils.py:118-172  parse_data: cv2.imread -> mix_up -> random_color_distort -> random_expand ->
 *                                  random_crop_with_constraints -> resize_with_bbox(random interp) -> random_flip -> / 255
 *     utils/data_aug.py:12-39      mix_up                     utils/data_aug.py:228-271  random_color_distort
 *     utils/data_aug.py:349-380    random_expand              utils/data_aug.py:274-320  letterbox_resize / resize_with_bbox
 *     utils/data_aug.py:323-346    random_flip
 *
 * Every RANDOM DRAW and all BOX arithmetic stay on the Python side (yolov3_tensorflow_amd/utils/data_aug.py, pinned draw
 * for draw against the reference module); a job carries their outcome.  The library does the pixel work, and does it in
 * one pass over the pixels that survive: it materialises only the crop window, jitters only those pixels (the jitter is
 * per pixel, so it commutes with expansion and cropping), resizes once, and writes the network's float32 input directly.
 * The result is bit-identical to the numpy / PIL chain of data_aug.py (tests/test_feed_native.py).
 *
 * Images are 8-bit RGB, HWC, rows contiguous (stride = width * 3).  All functions return 0 on success and a negative
 * Y3F_E* code otherwise; y3f_last_error() describes the last failure of the calling thread.  Every function is
 * re-entrant: no global state, the caller's threads may run any number of jobs concurrently.
 */
#ifndef YOLO355_FEED_H
#define YOLO355_FEED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y3F_OK 0
#define Y3F_EINVAL (-1)
#define Y3F_ENOMEM (-2)

/* cv2's interpolation codes, as parse_data draws them (utils/data_utils.py:157 `interp = np.random.randint(0, 5)`).
 * NEAREST and LINEAR restate OpenCV's uint8 arithmetic (no antialiasing); CUBIC / AREA / LANCZOS4 are Pillow's
 * BICUBIC / BOX / LANCZOS resampling (there is no OpenCV in this stack: yolov3_tensorflow_amd/utils/data_aug.py). */
#define Y3F_INTER_NEAREST 0
#define Y3F_INTER_LINEAR 1
#define Y3F_INTER_CUBIC 2
#define Y3F_INTER_AREA 3
#define Y3F_INTER_LANCZOS4 4

const char* y3f_last_error(void);
int y3f_abi_version(void);

/* cv2.resize(src, (dst_w, dst_h), interpolation=interp) for a 3-channel uint8 image (see the codes above). */
int y3f_resize(const uint8_t* src, int src_h, int src_w, uint8_t* dst, int dst_h, int dst_w, int interp);

/* Pillow's 8-bit RGB <-> HSV conversions (Image.convert('HSV') / .convert('RGB')), `pixels` triples. */
int y3f_rgb_to_hsv(const uint8_t* rgb, uint8_t* hsv, size_t pixels);
int y3f_hsv_to_rgb(const uint8_t* hsv, uint8_t* rgb, size_t pixels);

/* The photometric jitter of random_color_distort once its draws are made (utils/data_aug.py:228-271). */
typedef struct y3f_colour {
    int32_t enabled;      /* 0: the pixels pass through untouched (validation mode) */
    int32_t brightness;   /* added to R, G and B, clamped to 0..255 (0: the coin said no) */
    int32_t hue_on;       /* rotate the hue by hue_delta steps of the 180-step circle */
    int32_t hue_delta;
    float sat_gain;       /* S *= sat_gain, V *= val_gain in float32, clamped to 0..255 (1: off) */
    float val_gain;
} y3f_colour;

int y3f_colour_distort(uint8_t* rgb, size_t pixels, const y3f_colour* colour);     /* in place */

/* One sample of parse_data after the draws.  The source image (optionally blended with a mix-up partner on a common
 * top-left anchored canvas) sits at (off_x, off_y) on an unbounded black canvas; the window (win_*) of that canvas is
 * jittered, resized to res_w x res_h with `interp`, placed at (pad_x, pad_y) on an out_h x out_w field of pad_value,
 * and mirrored left-right when flip_x.  Plain resize: res = out, pad 0.  Letterbox: res = the fitted size, pad_value 128. */
typedef struct y3f_job {
    const uint8_t* img1;
    const uint8_t* img2;              /* mix-up partner or NULL */
    int32_t h1, w1, h2, w2;
    float lam1, lam2;                 /* mix-up weights of img1 / img2 (float32, as numpy multiplies them) */
    y3f_colour colour;
    int32_t off_x, off_y;
    int32_t win_x, win_y, win_w, win_h;
    int32_t interp;
    int32_t res_w, res_h;
    int32_t out_w, out_h, pad_x, pad_y, pad_value;
    int32_t flip_x;
} y3f_job;

/* Runs one job.  Writes out_u8 (out_h * out_w * 3 bytes) and / or out_f32 (the same pixels / 255 in float32, the network's
 * input); either may be NULL. */
int y3f_sample(const y3f_job* job, uint8_t* out_u8, float* out_f32);

/* Runs n jobs on up to `threads` threads of the library's own (0: one per job, at most the hardware's); outs_* are arrays of
 * n pointers (or NULL).  Returns the first failing job's code. */
int y3f_sample_batch(const y3f_job* jobs, int n, uint8_t* const* outs_u8, float* const* outs_f32, int threads);

/* The trial loop of random_crop_with_constraints (utils/data_aug.py:128-225 of the reference: per IoU band up to max_trial
 * random windows until one whose IoU with every box lies in the band) drawn from the CALLER's Python generator: mt_state
 * is random.Random.getstate()[1] as 625 uint32 (MT19937 words + position), advanced in place exactly as prng.uniform /
 * prng.randrange would advance it.  boxes: [n_boxes][4] x_min, y_min, x_max, y_max; bands: [n_bands][2] (min_iou, max_iou,
 * +-infinity for an open end); windows: room for n_bands * 4 int32 (x, y, width, height).  *n_windows = the number of
 * windows found, or -1 when there are no boxes: then windows[0..3] is the first proper window and the search ended there. */
int y3f_crop_candidates(uint32_t* mt_state, const double* boxes, int n_boxes, int width, int height, double min_scale,
                        double max_scale, double max_aspect_ratio, const double* bands, int n_bands, int max_trial,
                        int32_t* windows, int32_t* n_windows);

/* ---- the device form of the pixel work ----------------------------------------------------------------------------
 *
 * y3f_sample costs 5 of the 5.9 ms one core spends per image (profiles/r03_feeder_rate.txt): a rank of the bs=64 train step
 * needs four cores for it.  The same pass can run on the GPU beside the train step - its arithmetic is integer and table
 * look-ups, a few hundred microseconds per batch - once the host has done what must be done in double precision exactly as
 * Pillow / OpenCV do it: y3f_plan_batch turns n jobs into ONE relocatable blob (the job records, the source pixels the
 * windows need, the jitter maps, the resampling coefficient tables), the caller uploads the blob, and y3_feed_run
 * (include/yolo355.h, libyolo355.so) produces the float32 batch on the device.  Same bytes as y3f_sample
 * (tests/test_feed_gpu.py; tests/test_feed_plan.py runs the device functions on the host against y3f_sample).
 *
 * All offsets are bytes from the start of the blob (sources, maps, tables) or of the device scratch (win, tmp). */
#define Y3F_MODE_NEAREST 0
#define Y3F_MODE_LINEAR 1
#define Y3F_MODE_MEAN2X2 2      /* INTER_LINEAR's exact 2x2 reduction */
#define Y3F_MODE_COPY 3         /* the window has the resize target's size already */
#define Y3F_MODE_RESAMPLE 4     /* Pillow's two-pass 8-bit filters: CUBIC / AREA / LANCZOS4 */

typedef struct y3f_djob {
    uint64_t img1_off, img2_off;        /* packed sub-rectangles of the sources, row stride r*_w * 3 */
    uint64_t jitter_off;                /* 4 x 256 bytes: brightness, hue, saturation, value maps (colour_on) */
    uint64_t xtab_off, ytab_off;        /* int32 tables.  NEAREST: source index per output index.  LINEAR: [4][n] lo, hi,
                                           weight of lo, weight of hi.  RESAMPLE: first[n], count[n], coef[n][ksize] */
    uint64_t win_off, tmp_off;          /* scratch: the live part of the window (blended, jittered); the horizontal pass */
    int32_t r1_x0, r1_y0, r1_w, r1_h;   /* the part of img1 that was packed, in img1's coordinates */
    int32_t r2_x0, r2_y0, r2_w, r2_h;   /* (r2_w = 0 without a partner) */
    float lam1, lam2;
    int32_t has2, colour_on;
    int32_t img_dx, img_dy;             /* image coordinate = window coordinate + img_d* */
    int32_t live_x0, live_y0, live_x1, live_y1;     /* the part of the window that is not black canvas */
    int32_t win_w, win_h;
    int32_t mode, horizontal, vertical, ksize_x, ksize_y;
    int32_t tmp_y0, tmp_rows;           /* the horizontal pass holds window rows [tmp_y0, tmp_y0 + tmp_rows) */
    int32_t res_w, res_h, out_w, out_h, pad_x, pad_y, pad_value, flip_x;
    int32_t reserved[3];
} y3f_djob;                             /* 208 bytes */

/* Plans n jobs for the device.  Always sets *blob_bytes and *scratch_bytes to what the batch needs; when `blob` is not
 * NULL and `capacity` suffices, also writes the blob (the n y3f_djob records first) on up to `threads` threads (0: the
 * hardware's).  A caller sizes its pinned buffer with a first call, or simply retries when *blob_bytes > capacity.
 * Every job must have the same out_w x out_h. */
int y3f_plan_batch(const y3f_job* jobs, int n, uint8_t* blob, size_t capacity, size_t* blob_bytes, size_t* scratch_bytes,
                   int threads);

/* The constant tables of the colour conversions and of the final / 255, in the layout the device kernels read
 * (y3f_dtables below); uploaded once per device.  Returns the byte count (dst may be NULL). */
typedef struct y3f_dtables {
    uint8_t hue[3][256][256];   /* [channel that is the maximum][d of the next channel][d of the one after] */
    uint8_t sat[256][256];      /* [max][min] */
    uint8_t sector[256];        /* floor(h * 6 / 255) % 6 */
    float frac[256];            /* h * 6 / 255 - floor(.) */
    float unit[256];            /* s / 255, float32 of the double quotient */
    float unit255[256];         /* v / 255.f: the network's input value of byte v */
} y3f_dtables;
size_t y3f_device_tables(void* dst, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
