/*
 * utils.h -- C++ drop-in for the reference's stage API (StevenChang5/Canny_Edge src/utils.h:8-22).
 *
 * Same names, same reference-to-pointer signatures, same ownership rules as the reference's CPU
 * library, so the reference's main.cpp and tests/utils/test_utils.cpp link against
 * libcanny_utils.so unchanged:
 *   - outputs are allocated by the callee with new[] and released by the caller with delete[];
 *   - sobelOperator delete[]s its input plane (src/utils.cpp:235);
 *   - nonmaximalSuppression delete[]s magnitude and angle (src/utils.cpp:306-307);
 *   - hysteresis and findEdgePixels work in place.
 * Every function runs on the MI355X through the C ABI in canny_hip.h (no CPU path).  The reference
 * functions return void and have no error channel; these throw std::runtime_error on failure.
 */
#ifndef UTILS_H
#define UTILS_H

#define PI 3.1415926535
#define EDGE 255
#define NOEDGE 0

void gaussian(unsigned char*& img, float sigma, int height, int width, short int*& result);

void createGaussianKernel(float*& kernel, float sigma, int* window);

void calculateXYGradient(short int*& img, int height, int width, short int*& grad_x, short int*& grad_y);

void sobelOperator(short int*& img, int height, int width, short int*& magnitude, short int*& angle);

void nonmaximalSuppression(short int*& grad, short int*& angle, int height, int width, short int*& result);

void hysteresis(short int*& edgeCandidates, int height, int width, int minVal, int maxVal);

void findEdgePixels(short int*& edgeCandidates, bool*& visited, int start, int minVal, int maxVal, int height, int width);

/* Runs the four stages, prints "Execution time: X seconds" like the reference, and -- because there
 * is no display on a GPU server -- writes the edge map (and with steps=true every intermediate,
 * min-max normalised to 8 bits like the reference's imshow path) as PGM files into the directory
 * named by $CANNY_OUTPUT_DIR (default: current directory). */
void canny(unsigned char* img, float sigma, int minVal, int maxVal, int height, int width, bool steps);

/* Extension: same pipeline, result handed back (new[]-allocated {0,255} map) instead of written. */
short int* cannyEdges(unsigned char* img, float sigma, int minVal, int maxVal, int height, int width);

#endif
