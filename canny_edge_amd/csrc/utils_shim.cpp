// utils_shim.cpp -- the reference's C++ stage API (include/utils.h, include/cuda.h) implemented as
// thin shims over the C ABI (include/canny_hip.h).  The shims own nothing but the new[]/delete[]
// contract of the reference (src/utils.cpp:34,107-108,204-205,235,250,306-307) and a per-thread
// context on the device named by $CANNY_HIP_DEVICE (default 0).
#include "utils.h"
#include "cuda.h"

#include "canny_frames.h"
#include "canny_hip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

static_assert(sizeof(bool) == 1, "findEdgePixels hands bool* to the C ABI as one byte per pixel");

namespace {

struct CtxHolder {
    canny_hip_ctx *ctx = nullptr;
    ~CtxHolder() { canny_hip_ctx_destroy(ctx); }
};

canny_hip_ctx *ctx()
{
    thread_local CtxHolder holder;
    if (!holder.ctx) {
        int device = 0;
        if (const char *env = std::getenv("CANNY_HIP_DEVICE")) device = std::atoi(env);
        int st = canny_hip_ctx_create(&holder.ctx, device);
        if (st) throw std::runtime_error(std::string("canny_hip_ctx_create: ") + canny_hip_status_string(st));
    }
    return holder.ctx;
}

void check(int st, const char *what)
{
    if (st == CANNY_HIP_OK) return;
    std::string msg = std::string(what) + ": " + canny_hip_status_string(st);
    const char *detail = canny_hip_last_error(ctx());
    if (detail && *detail) msg += std::string(" (") + detail + ")";
    throw std::runtime_error(msg);
}

size_t count(int height, int width) { return (size_t)height * (size_t)width; }

std::string out_dir()
{
    const char *env = std::getenv("CANNY_OUTPUT_DIR");
    return env && *env ? std::string(env) : std::string(".");
}

// CANNY_OUTPUT_FORMAT=png turns the ".pgm" of every output name into ".png" (Main -p sets it).
void write_pgm(const std::string &name, const unsigned char *px, int height, int width)
{
    std::string path = out_dir() + "/" + name;
    const char *fmt = std::getenv("CANNY_OUTPUT_FORMAT");
    if (fmt && std::string(fmt) == "png" && path.size() > 4) path.replace(path.size() - 4, 4, ".png");
    if (canny_frames_write_gray(path.c_str(), px, height, width)) std::cerr << "WARNING: cannot write " << path << "\n";
}

// The reference shows each plane through cv::normalize(src, dst, 0, 255, NORM_MINMAX) followed by
// convertTo(CV_8U) (src/utils.cpp:440-486): linear min-max stretch, then round-to-nearest-even with
// saturation.  A constant plane maps to 0.
void write_normalised(const std::string &name, const short *plane, int height, int width)
{
    size_t n = count(height, width);
    int lo = std::numeric_limits<int>::max(), hi = std::numeric_limits<int>::min();
    for (size_t i = 0; i < n; i++) {
        lo = plane[i] < lo ? plane[i] : lo;
        hi = plane[i] > hi ? plane[i] : hi;
    }
    std::vector<unsigned char> px(n, 0);
    if (hi > lo) {
        double scale = 255.0 / (double)(hi - lo);
        for (size_t i = 0; i < n; i++) {
            double v = (plane[i] - lo) * scale;
            long r = std::lrint(v);
            px[i] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    write_pgm(name, px.data(), height, width);
}

void run_canny(unsigned char *img, float sigma, int minVal, int maxVal, int height, int width, bool steps,
               const char *tag)
{
    size_t n = count(height, width);
    std::vector<short> edges(n);
    auto start = std::chrono::high_resolution_clock::now();
    if (!steps) {
        check(canny_hip_canny(ctx(), img, sigma, minVal, maxVal, height, width, edges.data()), tag);
    } else {
        std::vector<short> smoothed(n), mag(n), ang(n);
        check(canny_hip_gaussian(ctx(), img, sigma, height, width, smoothed.data()), "gaussian");
        write_normalised("canny_step1_gaussian.pgm", smoothed.data(), height, width);
        check(canny_hip_sobel(ctx(), smoothed.data(), height, width, mag.data(), ang.data()), "sobelOperator");
        write_normalised("canny_step2_gradient.pgm", mag.data(), height, width);
        check(canny_hip_nms(ctx(), mag.data(), ang.data(), height, width, edges.data()), "nonmaximalSuppression");
        write_normalised("canny_step3_nonmaximal.pgm", edges.data(), height, width);
        check(canny_hip_hysteresis(ctx(), edges.data(), height, width, minVal, maxVal), "hysteresis");
    }
    auto stop = std::chrono::high_resolution_clock::now();
    write_normalised("canny_edges.pgm", edges.data(), height, width);
    std::chrono::duration<double> duration = stop - start;
    std::cout << "Execution time: " << duration.count() << " seconds\n";
}

} // namespace

// ---- utils.h ------------------------------------------------------------------------------------
void createGaussianKernel(float *&kernel, float sigma, int *window)
{
    float taps[CANNY_HIP_MAX_WINDOW];
    int w = 0;
    check(canny_hip_gaussian_kernel(sigma, taps, CANNY_HIP_MAX_WINDOW, &w), "createGaussianKernel");
    kernel = new float[w];
    std::memcpy(kernel, taps, (size_t)w * sizeof(float));
    *window = w;
}

// Outputs are owned by unique_ptr until the C-ABI call has succeeded: check() throws on any failure (the reference
// has no error channel), and a throw must not leak the planes or leave the caller with dangling out-parameters.
using plane_ptr = std::unique_ptr<short int[]>;
static plane_ptr new_plane(int height, int width) { return plane_ptr(new short int[count(height, width)]); }

void gaussian(unsigned char *&img, float sigma, int height, int width, short int *&result)
{
    plane_ptr out = new_plane(height, width);
    check(canny_hip_gaussian(ctx(), img, sigma, height, width, out.get()), "gaussian");
    result = out.release();
}

void calculateXYGradient(short int *&img, int height, int width, short int *&grad_x, short int *&grad_y)
{
    plane_ptr gx = new_plane(height, width), gy = new_plane(height, width);
    check(canny_hip_xy_gradient(ctx(), img, height, width, gx.get(), gy.get()), "calculateXYGradient");
    grad_x = gx.release();
    grad_y = gy.release();
}

void sobelOperator(short int *&img, int height, int width, short int *&magnitude, short int *&angle)
{
    plane_ptr mag = new_plane(height, width), ang = new_plane(height, width);
    check(canny_hip_sobel(ctx(), img, height, width, mag.get(), ang.get()), "sobelOperator");
    magnitude = mag.release();
    angle = ang.release();
    delete[] img; // the reference consumes its input (src/utils.cpp:235); on failure the caller still owns it
}

void nonmaximalSuppression(short int *&grad, short int *&angle, int height, int width, short int *&result)
{
    plane_ptr out = new_plane(height, width);
    check(canny_hip_nms(ctx(), grad, angle, height, width, out.get()), "nonmaximalSuppression");
    result = out.release();
    delete[] grad;  // src/utils.cpp:306
    delete[] angle; // src/utils.cpp:307
}

void hysteresis(short int *&edgeCandidates, int height, int width, int minVal, int maxVal)
{
    check(canny_hip_hysteresis(ctx(), edgeCandidates, height, width, minVal, maxVal), "hysteresis");
}

void findEdgePixels(short int *&edgeCandidates, bool *&visited, int start, int minVal, int maxVal, int height,
                    int width)
{
    check(canny_hip_find_edge_pixels(ctx(), edgeCandidates, reinterpret_cast<unsigned char *>(visited), start, minVal,
                                     maxVal, height, width),
          "findEdgePixels");
}

void canny(unsigned char *img, float sigma, int minVal, int maxVal, int height, int width, bool steps)
{
    run_canny(img, sigma, minVal, maxVal, height, width, steps, "canny");
}

short int *cannyEdges(unsigned char *img, float sigma, int minVal, int maxVal, int height, int width)
{
    plane_ptr edges = new_plane(height, width);
    check(canny_hip_canny(ctx(), img, sigma, minVal, maxVal, height, width, edges.get()), "cannyEdges");
    return edges.release();
}

// ---- cuda.h (the reference's GPU-path names; inputs are NOT freed, src/cuda.cu:446-449) -----------
void cuda_gaussian(unsigned char *&img_h, float sigma, int height, int width, short int *&result_h)
{
    plane_ptr out = new_plane(height, width);
    check(canny_hip_gaussian(ctx(), img_h, sigma, height, width, out.get()), "cuda_gaussian");
    result_h = out.release();
}

void cuda_sobel(short int *&img_h, int height, int width, short int *&magnitude_h, short int *&angle_h)
{
    plane_ptr mag = new_plane(height, width), ang = new_plane(height, width);
    check(canny_hip_sobel(ctx(), img_h, height, width, mag.get(), ang.get()), "cuda_sobel");
    magnitude_h = mag.release();
    angle_h = ang.release();
}

void cuda_nonmaixmal_suppression(short int *&magnitude_h, short int *&angle_h, int height, int width,
                                 short int *&result_h)
{
    plane_ptr out = new_plane(height, width);
    check(canny_hip_nms(ctx(), magnitude_h, angle_h, height, width, out.get()), "cuda_nonmaixmal_suppression");
    result_h = out.release();
}

void cuda_canny(unsigned char *img, float sigma, int min_val, int max_val, int height, int width, bool steps)
{
    run_canny(img, sigma, min_val, max_val, height, width, steps, "cuda_canny");
}
