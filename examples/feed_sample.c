/* A C host of liby3feed.so (include/yolo355_feed.h), no Python: reads a binary PPM (P6), runs one y3f_sample job - the
 * image on a black canvas twice its size, a crop window across its edge, a colour jitter, Lanczos resize to 416x416,
 * mirrored - and writes the result as a PPM plus the first values of the float32 network input.
 *
 *   gcc -O2 -I include examples/feed_sample.c -o /tmp/feed_sample -L yolov3_tensorflow_amd/csrc -ly3feed \
 *       -Wl,-rpath,$PWD/yolov3_tensorflow_amd/csrc
 *   /tmp/feed_sample in.ppm out.ppm
 *
 * tests/test_feed_native.py builds and runs it and compares out.ppm with the Python binding's result for the same job. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "yolo355_feed.h"

static uint8_t* read_ppm(const char* path, int* w, int* h) {
    FILE* f = fopen(path, "rb");
    int maxval = 0;
    if (!f || fscanf(f, "P6 %d %d %d", w, h, &maxval) != 3 || maxval != 255) return NULL;
    fgetc(f);                                   /* the single whitespace byte after the header */
    size_t n = (size_t)*w * *h * 3;
    uint8_t* px = (uint8_t*)malloc(n);
    if (!px || fread(px, 1, n, f) != n) return NULL;
    fclose(f);
    return px;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.ppm out.ppm\n", argv[0]);
        return 2;
    }
    int w = 0, h = 0;
    uint8_t* img = read_ppm(argv[1], &w, &h);
    if (!img) {
        fprintf(stderr, "cannot read %s as a binary PPM\n", argv[1]);
        return 1;
    }
    y3f_job job;
    memset(&job, 0, sizeof(job));
    job.img1 = img, job.h1 = h, job.w1 = w;
    job.lam1 = 1.0f;
    job.colour.enabled = 1, job.colour.brightness = 9, job.colour.hue_on = 1, job.colour.hue_delta = -7;
    job.colour.sat_gain = 1.25f, job.colour.val_gain = 0.9f;
    job.off_x = w / 2, job.off_y = h / 2;                                   /* the image on a 2w x 2h canvas */
    job.win_x = w / 4, job.win_y = h / 4, job.win_w = w, job.win_h = h;      /* a window across the image's corner */
    job.interp = Y3F_INTER_LANCZOS4;
    job.res_w = job.out_w = 416, job.res_h = job.out_h = 416;
    job.pad_value = 128, job.flip_x = 1;
    uint8_t* out = (uint8_t*)malloc((size_t)416 * 416 * 3);
    float* net = (float*)malloc((size_t)416 * 416 * 3 * sizeof(float));
    int rc = y3f_sample(&job, out, net);
    if (rc != Y3F_OK) {
        fprintf(stderr, "y3f_sample: %s (code %d)\n", y3f_last_error(), rc);
        return 1;
    }
    FILE* f = fopen(argv[2], "wb");
    fprintf(f, "P6\n416 416\n255\n");
    fwrite(out, 1, (size_t)416 * 416 * 3, f);
    fclose(f);
    printf("abi %d; network input[0..2] of the last row: %.6f %.6f %.6f\n", y3f_abi_version(), net[415 * 416 * 3],
           net[415 * 416 * 3 + 1], net[415 * 416 * 3 + 2]);
    return 0;
}
