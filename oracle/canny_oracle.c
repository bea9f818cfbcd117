/*
 * canny_oracle.c -- CPU restatement of the reference Canny hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load or call anything in oracle/.  The shipped HIP
 * path (canny_edge_amd/csrc) never links, loads or falls back to this file.
 *
 * What it restates (paths relative to the reference checkout, StevenChang5/Canny_Edge):
 *   src/utils.cpp:77-95    createGaussianKernel   -> canny_oracle_gaussian_kernel
 *   src/utils.cpp:26-68    gaussian               -> canny_oracle_gaussian
 *   src/utils.cpp:106-187  calculateXYGradient    -> canny_oracle_xy_gradient
 *   src/utils.cpp:201-236  sobelOperator          -> canny_oracle_sobel
 *   src/utils.cpp:248-308  nonmaximalSuppression  -> canny_oracle_nms
 *   src/utils.cpp:322-342  hysteresis             -> canny_oracle_hysteresis
 *   src/utils.cpp:360-427  findEdgePixels         -> canny_oracle_find_edge_pixels
 *   src/utils.cpp:429-492  canny (4 timed stages) -> canny_oracle_canny
 *   src/utils.h:4-6        PI / EDGE / NOEDGE
 *
 * Parity pin: the reference itself cannot be built in this image (src/utils.cpp includes
 * <opencv2/opencv.hpp>, which is absent, and stand-in headers are not allowed), so this
 * restatement is pinned by every known-answer vector of the reference's own test-suite
 * (tests/utils/test_utils.cpp) -- see tests/golden/reference_vectors.json and
 * tests/test_oracle_golden.py.
 *
 * Build flags matter: -O2 -ffp-contract=off and NO -march=native / -ffast-math.  The Gaussian is
 * a chain of separately rounded float multiplies and adds whose quotient is truncated to short;
 * fusing them into FMAs changes pixel values.
 *
 * The arithmetic structure (per-tap bounds tests, double sqrt/atan2 per pixel, FIFO flood fill,
 * single thread) deliberately mirrors the reference so that timing this file is a fair
 * "port" CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORACLE_PI 3.1415926535 /* src/utils.h:4 -- truncated on purpose */
#define ORACLE_EDGE 255        /* src/utils.h:5 */
#define ORACLE_NOEDGE 0        /* src/utils.h:6 */

/* ------------------------------------------------------------------------------------------
 * Gaussian taps.  src/utils.cpp:77-95.
 * window = 1 + 2*ceil(3*sigma) evaluated in float and converted to int.
 * tap    = (float)( (double)expf(-(x*x)/(2*sigma*sigma)) / (sqrt(6.2831853) * (double)sigma) )
 * then a float running sum and a float divide per tap.
 * Returns the window, or -1 if it does not fit in cap.
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_gaussian_window(float sigma)
{
    float w = 1 + 2 * ceilf(3 * sigma);
    return (int)w;
}

int canny_oracle_gaussian_kernel(float sigma, float *taps, int cap)
{
    int window = canny_oracle_gaussian_window(sigma);
    if (window > cap || window < 1)
        return -1;
    int center = window / 2;
    float total = 0.0f;
    for (int i = 0; i < window; i++) {
        float x = (float)(i - center);
        float e = expf(-((x * x) / (2 * sigma * sigma)));
        float tap = (float)((double)e / (sqrt(6.2831853) * (double)sigma));
        taps[i] = tap;
        total += tap;
    }
    for (int i = 0; i < window; i++)
        taps[i] /= total;
    return window;
}

/* ------------------------------------------------------------------------------------------
 * Separable blur with truncated-and-renormalised borders.  src/utils.cpp:26-68.
 * Row pass u8 -> float, column pass float -> short by truncation.  Taps are visited in
 * ascending order and only in-bounds taps contribute to both the sum and the weight.
 * If row_pass_out is non-NULL it receives the float intermediate (test hook).
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_gaussian(const uint8_t *img, float sigma, int height, int width,
                          int16_t *result, float *row_pass_out)
{
    float taps[1024];
    int window = canny_oracle_gaussian_kernel(sigma, taps, 1024);
    if (window < 0)
        return -1;
    int center = window / 2;
    size_t n = (size_t)height * (size_t)width;
    float *tmp = (float *)malloc(n * sizeof(float));
    if (!tmp)
        return -2;

    for (int r = 0; r < height; r++) {
        for (int c = 0; c < width; c++) {
            float acc = 0;
            float wsum = 0;
            for (int k = -center; k < center + 1; k++) {
                if (c + k >= 0 && c + k < width) {
                    acc += (float)img[(size_t)r * width + (c + k)] * taps[center + k];
                    wsum += taps[center + k];
                }
            }
            tmp[(size_t)r * width + c] = acc / wsum;
        }
    }

    for (int c = 0; c < width; c++) {
        for (int r = 0; r < height; r++) {
            float acc = 0;
            float wsum = 0;
            for (int k = -center; k < center + 1; k++) {
                if (r + k >= 0 && r + k < height) {
                    acc += tmp[(size_t)(r + k) * width + c] * taps[center + k];
                    wsum += taps[center + k];
                }
            }
            result[(size_t)r * width + c] = (int16_t)(acc / wsum);
        }
    }
    if (row_pass_out)
        memcpy(row_pass_out, tmp, n * sizeof(float));
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * 3x3 Sobel derivatives.  src/utils.cpp:106-187.
 *   gx: [-1 0 1; -2 0 2; -1 0 1]; column index clamped to the image, rows outside dropped.
 *   gy: (row below) - (row above), weights 1 2 1; row index clamped, columns outside dropped.
 * Values are stored through short (wrap), as in the reference's short grad arrays.
 * Needs height >= 2 and width >= 2 (the reference reads out of bounds below that).
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_xy_gradient(const int16_t *img, int height, int width, int16_t *gx, int16_t *gy)
{
    if (height < 2 || width < 2)
        return -1;
    for (int r = 0; r < height; r++) {
        for (int c = 0; c < width; c++) {
            size_t p = (size_t)r * width + c;
            int cl = c > 0 ? c - 1 : 0;
            int cr = c < width - 1 ? c + 1 : width - 1;
            const int16_t *row = img + (size_t)r * width;
            int v = 2 * row[cr] - 2 * row[cl];
            if (r != height - 1)
                v += row[width + cr] - row[width + cl];
            if (r != 0)
                v += row[cr - width] - row[cl - width];
            gx[p] = (int16_t)v;

            int ru = r > 0 ? r - 1 : 0;
            int rd = r < height - 1 ? r + 1 : height - 1;
            const int16_t *up = img + (size_t)ru * width;
            const int16_t *dn = img + (size_t)rd * width;
            int u = 2 * dn[c] - 2 * up[c];
            if (c != width - 1)
                u += dn[c + 1] - up[c + 1];
            if (c != 0)
                u += dn[c - 1] - up[c - 1];
            gy[p] = (int16_t)u;
        }
    }
    return 0;
}

/* Angle bin of one gradient.  src/utils.cpp:215-231. */
int canny_oracle_angle_bin(int gx, int gy)
{
    float a = atan2((double)gy, (double)gx); /* double atan2 rounded to float */
    a *= (180 / ORACLE_PI);                   /* float *= double: product in double, rounded to float */
    if (a < 0)
        a = 360 + a;
    if ((a >= 22.5 && a < 67.5) || (a >= 202.5 && a < 247.5))
        return 45;
    if ((a >= 112.5 && a < 157.5) || (a >= 292.5 && a < 337.5))
        return 135;
    if ((a >= 67.5 && a < 112.5) || (a >= 247.5 && a < 292.5))
        return 90;
    return 0;
}

/* Magnitude of one gradient.  src/utils.cpp:212: (int)sqrt(double(gx*gx+gy*gy)) stored to short. */
int canny_oracle_magnitude(int gx, int gy)
{
    return (int16_t)(int)sqrt((double)(gx * gx + gy * gy));
}

/* ------------------------------------------------------------------------------------------
 * Gradient magnitude and 4-way direction.  src/utils.cpp:201-236.
 * (The reference also frees its input; ownership is the C++ shim's business, not the oracle's.)
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_sobel(const int16_t *img, int height, int width, int16_t *magnitude, int16_t *angle)
{
    size_t n = (size_t)height * (size_t)width;
    int16_t *gx = (int16_t *)malloc(n * sizeof(int16_t));
    int16_t *gy = (int16_t *)malloc(n * sizeof(int16_t));
    if (!gx || !gy) {
        free(gx);
        free(gy);
        return -2;
    }
    int rc = canny_oracle_xy_gradient(img, height, width, gx, gy);
    if (rc == 0) {
        for (size_t i = 0; i < n; i++) {
            magnitude[i] = (int16_t)canny_oracle_magnitude(gx[i], gy[i]);
            angle[i] = (int16_t)canny_oracle_angle_bin(gx[i], gy[i]);
        }
    }
    free(gx);
    free(gy);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Non-maximal suppression.  src/utils.cpp:248-308.
 * Keep magnitude[i] iff it is strictly greater than both neighbours along its bin; a neighbour
 * outside the image is not compared.  Bin 45 looks up-right / down-left, bin 135 up-left /
 * down-right.  Any other angle value: the reference leaves result[i] unwritten; we write 0.
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_nms(const int16_t *magnitude, const int16_t *angle, int height, int width,
                     int16_t *result)
{
    for (int r = 0; r < height; r++) {
        for (int c = 0; c < width; c++) {
            size_t i = (size_t)r * width + c;
            int has_l = c > 0, has_r = c < width - 1, has_u = r > 0, has_d = r < height - 1;
            int m = magnitude[i];
            int keep = 1;
            switch (angle[i]) {
            case 0:
                if (has_l && m <= magnitude[i - 1]) keep = 0;
                if (has_r && m <= magnitude[i + 1]) keep = 0;
                break;
            case 45:
                if (has_r && has_u && m <= magnitude[i + 1 - width]) keep = 0;
                if (has_l && has_d && m <= magnitude[i - 1 + width]) keep = 0;
                break;
            case 90:
                if (has_u && m <= magnitude[i - width]) keep = 0;
                if (has_d && m <= magnitude[i + width]) keep = 0;
                break;
            case 135:
                if (has_l && has_u && m <= magnitude[i - 1 - width]) keep = 0;
                if (has_r && has_d && m <= magnitude[i + 1 + width]) keep = 0;
                break;
            default:
                keep = 0;
                break;
            }
            result[i] = keep ? (int16_t)m : ORACLE_NOEDGE;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Flood fill from one strong pixel.  src/utils.cpp:360-427.
 * FIFO breadth-first over 8-connectivity; every dequeued pixel becomes EDGE; a neighbour is
 * enqueued when its value >= min_val and it is not yet visited.  `start` itself is enqueued
 * WITHOUT being marked visited (so a neighbour may enqueue it once more).
 * Quirk kept from the reference (lines 378 and 399): the two upward diagonals test
 * `current - width > 0` (strict), so pixel (1,0) never pushes pixel (0,1).
 * fifo must hold height*width + 1 ints.
 * ---------------------------------------------------------------------------------------- */
static void flood_from(int16_t *cand, uint8_t *visited, int start, int min_val, int height,
                       int width, int *fifo)
{
    if (visited[start])
        return;
    long total = (long)height * width;
    size_t head = 0, tail = 0;
    fifo[tail++] = start;
#define TRY_PUSH(idx)                                  \
    do {                                               \
        int q_ = (idx);                                \
        if (cand[q_] >= min_val && !visited[q_]) {     \
            fifo[tail++] = q_;                         \
            visited[q_] = 1;                           \
        }                                              \
    } while (0)
    while (head != tail) {
        int cur = fifo[head];
        cand[cur] = ORACLE_EDGE;
        int col = cur % width;
        if (col > 0) {
            if ((long)cur + width < total) TRY_PUSH(cur + width - 1);
            if (cur - width > 0) TRY_PUSH(cur - width - 1);
            TRY_PUSH(cur - 1);
        }
        if (col < width - 1) {
            if ((long)cur + width < total) TRY_PUSH(cur + width + 1);
            if (cur - width > 0) TRY_PUSH(cur - width + 1);
            TRY_PUSH(cur + 1);
        }
        if ((long)cur + width < total) TRY_PUSH(cur + width);
        if (cur - width >= 0) TRY_PUSH(cur - width);
        head++;
    }
#undef TRY_PUSH
}

int canny_oracle_find_edge_pixels(int16_t *cand, uint8_t *visited, int start, int min_val,
                                  int max_val, int height, int width)
{
    (void)max_val; /* unused by the reference as well */
    /* each pixel is pushed at most once via `visited`, plus `start` once unmarked, plus one
     * possible re-push of `start` */
    int *fifo = (int *)malloc(((size_t)height * width + 2) * sizeof(int));
    if (!fifo)
        return -2;
    flood_from(cand, visited, start, min_val, height, width, fifo);
    free(fifo);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Double-threshold hysteresis, in place.  src/utils.cpp:322-342.
 * One ascending sweep: v < min -> 0, else v >= max -> flood; then v < max -> 0.
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_hysteresis(int16_t *cand, int height, int width, int min_val, int max_val)
{
    size_t n = (size_t)height * (size_t)width;
    uint8_t *visited = (uint8_t *)calloc(n ? n : 1, 1);
    int *fifo = (int *)malloc((n + 2) * sizeof(int));
    if (!visited || !fifo) {
        free(visited);
        free(fifo);
        return -2;
    }
    for (size_t i = 0; i < n; i++) {
        if (cand[i] < min_val)
            cand[i] = ORACLE_NOEDGE;
        else if (cand[i] >= max_val)
            flood_from(cand, visited, (int)i, min_val, height, width, fifo);
    }
    for (size_t i = 0; i < n; i++) {
        if (cand[i] < max_val)
            cand[i] = ORACLE_NOEDGE;
    }
    free(visited);
    free(fifo);
    return 0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------
 * Whole pipeline.  src/utils.cpp:429-492 minus display.  edges receives the final {0,255} map.
 * Optional outputs (may be NULL): smoothed, magnitude, angle, nms planes and stage_seconds[5]
 * = {gaussian, sobel, nms, hysteresis, total}, timed like the reference's chrono bracket
 * (lines 435 and 479: the four stages only).
 * ---------------------------------------------------------------------------------------- */
int canny_oracle_canny(const uint8_t *img, float sigma, int min_val, int max_val, int height,
                       int width, int16_t *edges, int16_t *smoothed_out, int16_t *magnitude_out,
                       int16_t *angle_out, int16_t *nms_out, double *stage_seconds)
{
    size_t n = (size_t)height * (size_t)width;
    int16_t *smoothed = (int16_t *)malloc(n * sizeof(int16_t));
    int16_t *mag = (int16_t *)malloc(n * sizeof(int16_t));
    int16_t *ang = (int16_t *)malloc(n * sizeof(int16_t));
    int rc = -2;
    if (smoothed && mag && ang) {
        double t0 = now_s();
        rc = canny_oracle_gaussian(img, sigma, height, width, smoothed, NULL);
        double t1 = now_s();
        if (rc == 0)
            rc = canny_oracle_sobel(smoothed, height, width, mag, ang);
        double t2 = now_s();
        if (rc == 0)
            rc = canny_oracle_nms(mag, ang, height, width, edges);
        double t3 = now_s();
        if (rc == 0 && nms_out)
            memcpy(nms_out, edges, n * sizeof(int16_t));
        double t3b = now_s();
        if (rc == 0)
            rc = canny_oracle_hysteresis(edges, height, width, min_val, max_val);
        double t4 = now_s();
        if (stage_seconds) {
            stage_seconds[0] = t1 - t0;
            stage_seconds[1] = t2 - t1;
            stage_seconds[2] = t3 - t2;
            stage_seconds[3] = t4 - t3b;
            stage_seconds[4] = (t3 - t0) + (t4 - t3b);
        }
        if (rc == 0) {
            if (smoothed_out) memcpy(smoothed_out, smoothed, n * sizeof(int16_t));
            if (magnitude_out) memcpy(magnitude_out, mag, n * sizeof(int16_t));
            if (angle_out) memcpy(angle_out, ang, n * sizeof(int16_t));
        }
    }
    free(smoothed);
    free(mag);
    free(ang);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Exhaustive tables (test hooks): angle bin and magnitude for every (gx, gy) in [-lim, lim]^2,
 * row-major with gy as the slow index.  lim = 1020 covers every gradient reachable from a
 * smoothed plane in [0,255].  Used to prove the device's integer-only classifier.
 * ---------------------------------------------------------------------------------------- */
void canny_oracle_angle_table(int lim, uint8_t *bins)
{
    int side = 2 * lim + 1;
    for (int gy = -lim; gy <= lim; gy++)
        for (int gx = -lim; gx <= lim; gx++)
            bins[(size_t)(gy + lim) * side + (gx + lim)] = (uint8_t)canny_oracle_angle_bin(gx, gy);
}

void canny_oracle_magnitude_table(int lim, int16_t *mags)
{
    int side = 2 * lim + 1;
    for (int gy = -lim; gy <= lim; gy++)
        for (int gx = -lim; gx <= lim; gx++)
            mags[(size_t)(gy + lim) * side + (gx + lim)] = (int16_t)canny_oracle_magnitude(gx, gy);
}
