// test_utils_dropin.cpp -- the reference's own unit tests (StevenChang5/Canny_Edge tests/utils/test_utils.cpp)
// replayed against the MI355X drop-in: it includes the drop-in's utils.h / cuda.h, calls the stage functions
// with the reference's reference-to-pointer signatures and new[]/delete[] ownership, and checks the same
// known-answer vectors.  GoogleTest is not available offline, so a minimal CHECK macro stands in for it.
// Vectors are data transcribed from the reference tests (line ranges cited per case); the harness is ours.
//
// Build (done by tests/test_cpp_dropin.py):  hipcc -std=c++14 -Iinclude tests/cpp/test_utils_dropin.cpp
//                                            -Lcanny_edge_amd -lcanny_utils -lcanny_hip
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "utils.h"
#include "cuda.h"
#include "canny_frames.h"

static int failures = 0, checks = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++checks;                                                                    \
        if (!(cond)) {                                                               \
            ++failures;                                                              \
            std::fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);     \
        }                                                                            \
    } while (0)

static std::vector<unsigned char> load_fixture(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<unsigned char> v((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return v;
}

int main(int argc, char **argv)
{
    const char *fixture = argc > 1 ? argv[1] : "tests/golden/test_gray_256x256.u8";

    // Gaussian.KernelSumOne / KernelValues / KernelCreation -- tests/utils/test_utils.cpp:7-45
    {
        float *kernel;
        int window;
        createGaussianKernel(kernel, 0.5, &window);
        float sum = 0;
        for (int i = 0; i < window; i++) sum += kernel[i];
        CHECK(std::fabs(sum - 1) < FLT_EPSILON);
        const float expected[5] = {0.0002638651f, 0.1064507720f, 0.7865707259f, 0.1064507720f, 0.0002638651f};
        CHECK(window == 5);
        for (int i = 0; i < 5 && i < window; i++) CHECK(std::fabs(expected[i] - kernel[i]) < FLT_EPSILON);
        delete[] kernel;
        createGaussianKernel(kernel, 2, &window);
        CHECK(window == 13);
        for (int i = 0; i < 7; i++) CHECK(kernel[i] == kernel[12 - i]);
        delete[] kernel;
    }

    // Gaussian.IsNonzero / InRange / GaussianDimensions -- :47-104 (image decoded once, see make_fixtures.py)
    {
        std::vector<unsigned char> img = load_fixture(fixture);
        CHECK(img.size() == 256u * 256u);
        if (img.size() == 256u * 256u) {
            unsigned char *data = img.data();
            short int *smoothed;
            gaussian(data, 0.5, 256, 256, smoothed);
            long sum = 0;
            bool in_range = true;
            for (int i = 0; i < 256 * 256; i++) {
                sum += smoothed[i];
                in_range = in_range && smoothed[i] <= 255 && smoothed[i] >= 0;
            }
            CHECK(sum != 0);
            CHECK(in_range);
            delete[] smoothed;
            short int *smoothed2;
            cuda_gaussian(data, 0.5, 256, 256, smoothed2); // the reference's GPU entry point, same result
            short int *again;
            gaussian(data, 0.5, 256, 256, again);
            bool same = true;
            for (int i = 0; i < 256 * 256; i++) same = same && smoothed2[i] == again[i];
            CHECK(same);
            delete[] smoothed2;
            delete[] again;
        }
    }

    // Gradient.xOnes / yOnes / xCorrect / yCorrect -- :128-208
    {
        short int *gx, *gy;
        short int *ones = new short int[9]{1, 1, 1, 1, 1, 1, 1, 1, 1};
        calculateXYGradient(ones, 3, 3, gx, gy);
        for (int i = 0; i < 9; i++) {
            CHECK(gx[i] == 0);
            CHECK(gy[i] == 0);
        }
        delete[] gx;
        delete[] gy;
        delete[] ones;
        short int *img = new short int[9]{1, 2, 1, 2, 3, 2, 3, 4, 3};
        calculateXYGradient(img, 3, 3, gx, gy);
        const int ex[9] = {3, 0, -3, 4, 0, -4, 3, 0, -3};
        const int ey[9] = {3, 4, 3, 6, 8, 6, 3, 4, 3};
        for (int i = 0; i < 9; i++) {
            CHECK(gx[i] == ex[i]);
            CHECK(gy[i] == ey[i]);
        }
        delete[] gx;
        delete[] gy;
        delete[] img;
    }

    // SobelOperator.GradientDimensions -- :210-230  (sobelOperator frees its input: no delete[] img here)
    {
        short int *img = new short int[9]{1, 1, 1, 1, 1, 1, 1, 1, 1};
        short int *magnitude, *angle;
        sobelOperator(img, 3, 3, magnitude, angle);
        for (int i = 0; i < 9; i++) {
            CHECK(magnitude[i] == 0);
            CHECK(angle[i] == 0);
        }
        delete[] magnitude;
        delete[] angle;
    }

    // NonmaximalSuppression.SuppressionCalculation0/45/90/135 -- :273-347 (inputs are freed by the callee)
    {
        struct Case {
            short grad[9], angle[9], expect[9];
        };
        const Case cases[4] = {
            {{0, 0, 0, 0, 10, 0, 50, 20, 50}, {0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 10, 0, 50, 0, 50}},
            {{0, 1, 1, 0, 2, 0, 1, 1, 0}, {0, 45, 45, 45, 45, 45, 45, 45, 0}, {0, 1, 0, 0, 2, 0, 0, 1, 0}},
            {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {90, 90, 90, 90, 90, 90, 90, 90, 90}, {1, 0, 0, 0, 1, 0, 0, 0, 1}},
            {{0, 1, 1, 0, 2, 0, 1, 1, 0}, {135, 135, 0, 135, 135, 135, 0, 135, 135}, {0, 1, 0, 0, 2, 0, 0, 1, 0}},
        };
        for (const Case &c : cases) {
            short int *grad = new short int[9], *angle = new short int[9], *suppress;
            for (int i = 0; i < 9; i++) {
                grad[i] = c.grad[i];
                angle[i] = c.angle[i];
            }
            nonmaximalSuppression(grad, angle, 3, 3, suppress);
            for (int i = 0; i < 9; i++) CHECK(suppress[i] == c.expect[i]);
            delete[] suppress;
        }
    }

    // findEdgePixels.CorrectCalculations -- :349-375 (20 initialisers, the fifth row is zero)
    {
        short int *suppress = new short int[25]{5, 6, 0, 5, 5, 4, 1, 0, 1, 4, 1, 3, 7, 0, 0, 10, 9, 8, 0, 0};
        const short expect[25] = {EDGE, EDGE, 0, 5, 5, EDGE, 1, 0, 1, 4, 1, EDGE, EDGE, 0, 0, EDGE, EDGE, EDGE, 0, 0};
        bool *visited = new bool[25];
        for (int i = 0; i < 25; i++) visited[i] = false;
        findEdgePixels(suppress, visited, 1, 2, 10, 5, 5);
        for (int i = 0; i < 25; i++) CHECK(suppress[i] == expect[i]);
        delete[] suppress;
        delete[] visited;
    }

    // Hysteresis.CorrectFunction -- :377-397
    {
        short int *suppress = new short int[25]{5, 6, 0, 5, 10, 4, 1, 0, 1, 4, 1, 3, 7, 0, 0, 10, 9, 8, 0, 0};
        const short expect[25] = {EDGE, EDGE, 0, EDGE, EDGE, EDGE, 0, 0, 0, EDGE, 0, EDGE, EDGE, 0, 0, EDGE, EDGE, EDGE, 0, 0};
        hysteresis(suppress, 5, 5, 2, 10);
        for (int i = 0; i < 25; i++) CHECK(suppress[i] == expect[i]);
        delete[] suppress;
    }

    // whole pipeline through the C++ names: staged calls == cannyEdges() == cuda_* staged calls
    {
        std::vector<unsigned char> img = load_fixture(fixture);
        if (img.size() == 256u * 256u) {
            unsigned char *data = img.data();
            short int *sm, *mag, *ang, *nms;
            gaussian(data, 1.0f, 256, 256, sm);
            sobelOperator(sm, 256, 256, mag, ang); // frees sm
            nonmaximalSuppression(mag, ang, 256, 256, nms); // frees mag, ang
            hysteresis(nms, 256, 256, 50, 150);
            short int *direct = cannyEdges(data, 1.0f, 50, 150, 256, 256);
            short int *gsm, *gmag, *gang, *gnms;
            cuda_gaussian(data, 1.0f, 256, 256, gsm);
            cuda_sobel(gsm, 256, 256, gmag, gang);
            cuda_nonmaixmal_suppression(gmag, gang, 256, 256, gnms);
            hysteresis(gnms, 256, 256, 50, 150);
            long edge_pixels = 0;
            bool same = true;
            for (int i = 0; i < 256 * 256; i++) {
                same = same && nms[i] == direct[i] && gnms[i] == direct[i];
                edge_pixels += direct[i] == EDGE;
            }
            CHECK(same);
            CHECK(edge_pixels == 2448); // the count SURVEY.md 8(c) records for the reference on this decode
            delete[] nms;
            delete[] direct;
            delete[] gsm; // cuda_* do not free their inputs (src/cuda.cu:446-449)
            delete[] gmag;
            delete[] gang;
            delete[] gnms;
        }
    }

    // The same image tests the way the reference runs them: from the JPEG itself.  cv::imread(path, IMREAD_GRAYSCALE)
    // (tests/utils/test_utils.cpp:49) becomes canny_frames_jpeg_decode_gray (include/canny_frames.h), which returns the
    // same bytes (libjpeg's luminance plane).  Gaussian.IsNonzero / InRange -- :47-104, then the whole pipeline.
    if (argc > 2) {
        std::vector<unsigned char> file = load_fixture(argv[2]);
        int height = 0, width = 0;
        CHECK(canny_frames_jpeg_info(file.data(), file.size(), &height, &width) == CANNY_FRAMES_OK);
        CHECK(height == 256 && width == 256);
        std::vector<unsigned char> img((size_t)height * width);
        CHECK(canny_frames_jpeg_decode_gray(file.data(), file.size(), img.data(), img.size(), &height, &width) ==
              CANNY_FRAMES_OK);
        if (height == 256 && width == 256) {
            unsigned char *data = img.data();
            short int *smoothed;
            gaussian(data, 0.5, height, width, smoothed);
            long sum = 0;
            bool in_range = true;
            for (int i = 0; i < height * width; i++) {
                sum += smoothed[i];
                in_range = in_range && smoothed[i] <= 255 && smoothed[i] >= 0;
            }
            CHECK(sum != 0);
            CHECK(in_range);
            delete[] smoothed;
            short int *edges = cannyEdges(data, 1.0f, 50, 150, height, width);
            long edge_pixels = 0;
            for (int i = 0; i < height * width; i++) edge_pixels += edges[i] == EDGE;
            CHECK(edge_pixels == 2445); // the oracle's count on this decode (tests/golden/oracle_stage_hashes.json)
            delete[] edges;
        }
    }

    std::printf("%d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
}
