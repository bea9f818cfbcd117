/* canny_frames.h -- frame sources next to the hot path.
 *
 * The reference reads its frames through OpenCV: the webcam (src/main.cpp:78-137) and, in its tests,
 * cv::imread(path, IMREAD_GRAYSCALE) of tests/test.jpg (tests/utils/test_utils.cpp:49).  Neither OpenCV nor libjpeg's
 * headers exist in this image, so `Main` and the tests get a small decoder of their own: baseline (sequential Huffman)
 * JPEG -> 8-bit gray, HOST code, no GPU involved.  For a YCbCr or a grayscale file the gray image is the luminance plane
 * reconstructed with the integer "slow" inverse DCT every libjpeg flavour uses by default -- the same bytes
 * IMREAD_GRAYSCALE (libjpeg out_color_space = JCS_GRAYSCALE) hands the reference.  tests/test_jpeg_gray.py holds it
 * to PIL's libjpeg on the same files, byte for byte.
 *
 * Not handled, reported as CANNY_FRAMES_ERR_UNSUPPORTED: progressive / lossless / arithmetic-coded files, 12-bit
 * samples, CMYK / YCCK, RGB-coded files, a luminance plane that is itself subsampled.
 *
 * Exported by canny_edge_amd/libcanny_utils.so.
 */
#ifndef CANNY_FRAMES_H
#define CANNY_FRAMES_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CANNY_FRAMES_OK 0
#define CANNY_FRAMES_ERR_ARG 1          /* null pointer, output buffer too small */
#define CANNY_FRAMES_ERR_FORMAT 2       /* not a JPEG, truncated or damaged stream */
#define CANNY_FRAMES_ERR_UNSUPPORTED 3  /* a JPEG flavour listed above */

/* Size of the image in `data` (reads the headers only). */
int canny_frames_jpeg_info(const void *data, size_t bytes, int *height, int *width);

/* Decodes `data` to height*width gray bytes, row-major, into `out` (capacity `out_bytes`); what
 * cv::imread(..., IMREAD_GRAYSCALE) returns for the same file (tests/utils/test_utils.cpp:49). */
int canny_frames_jpeg_decode_gray(const void *data, size_t bytes, unsigned char *out, size_t out_bytes, int *height,
                                  int *width);

/* Frame sink: writes height*width gray bytes as a PNG when `path` ends in ".png" (8-bit gray, stored -- not compressed --
 * deflate blocks), as a binary PGM otherwise.  What stands in for the reference's cv::imshow windows
 * (src/utils.cpp:440-486) in a headless build.  CANNY_FRAMES_ERR_ARG: bad argument or the file cannot be created;
 * CANNY_FRAMES_ERR_FORMAT: the write failed part-way. */
int canny_frames_write_gray(const char *path, const unsigned char *px, int height, int width);

/* Text of the calling thread's last JPEG failure ("" if none). */
const char *canny_frames_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
