/*
 * cuda.h -- the reference's GPU-path entry points (StevenChang5/Canny_Edge src/cuda.h:4-10), kept
 * under their original names so that main.cpp's `-c` branch links unchanged.  Nothing here is CUDA:
 * the four functions run the hand-written gfx950 HIP kernels through canny_hip.h.  Unlike the
 * utils.h variants they do NOT free their inputs (the reference frees them in cuda_canny,
 * src/cuda.cu:446-449).  Results follow the reference's CPU path (src/utils.cpp) bit for bit, not
 * the divergent arithmetic of src/cuda.cu.
 */
#ifndef CUDA_H
#define CUDA_H

void cuda_gaussian(unsigned char*& img_h, float sigma, int height, int width, short int*& result_h);

void cuda_sobel(short int*& img_h, int height, int width, short int*& magnitude_h, short int*& angle_h);

void cuda_nonmaixmal_suppression(short int*& magnitude_h, short int*& angle_h, int height, int width, short int*& result_h);

void cuda_canny(unsigned char* img, float sigma, int min_val, int max_val, int height, int width, bool steps);

#endif
