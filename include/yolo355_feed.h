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

#ifdef __cplusplus
}
#endif
#endif
