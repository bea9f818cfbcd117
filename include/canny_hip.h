/*
 * canny_hip.h -- C ABI of libcanny_hip.so: the MI355X (gfx950) Canny hot path.
 *
 * This is the drop-in boundary for StevenChang5/Canny_Edge's stage functions.  Every entry point
 * takes plain pointers and sizes (no C++ references, no torch types) so that it can be bound from
 * C, C++ (include/utils.h shims), Python ctypes (canny_edge_amd/capi.py) or any other FFI.
 *
 * Reference interface each function replaces (paths relative to the reference checkout):
 *
 *   canny_hip_gaussian_kernel      createGaussianKernel     src/utils.h:10   src/utils.cpp:77-95
 *   canny_hip_gaussian             gaussian                 src/utils.h:8    src/utils.cpp:26-68
 *                                  cuda_gaussian            src/cuda.h:4     src/cuda.cu:75-102
 *   canny_hip_xy_gradient          calculateXYGradient      src/utils.h:12   src/utils.cpp:106-187
 *   canny_hip_sobel                sobelOperator            src/utils.h:14   src/utils.cpp:201-236
 *                                  cuda_sobel               src/cuda.h:6     src/cuda.cu:220-246
 *   canny_hip_nms                  nonmaximalSuppression    src/utils.h:16   src/utils.cpp:248-308
 *                                  cuda_nonmaixmal_suppression src/cuda.h:8  src/cuda.cu:366-390
 *   canny_hip_hysteresis           hysteresis               src/utils.h:18   src/utils.cpp:322-342
 *   canny_hip_find_edge_pixels     findEdgePixels           src/utils.h:20   src/utils.cpp:360-427
 *   canny_hip_canny                canny / cuda_canny       src/utils.h:22   src/utils.cpp:429-492
 *                                                           src/cuda.h:10    src/cuda.cu:392-450
 *
 * Conventions (same as the reference): images are dense row-major, pixel (r,c) at r*width+c;
 * argument order is (..., height, width, ...); `short` planes are int16; thresholds are ints.
 * Unlike the reference's void functions every call returns a status (0 = CANNY_HIP_OK).
 * Results are bit-identical to the reference's CPU path (src/utils.cpp) on the documented domain.
 *
 * Numeric domain:
 *   - gaussian: any u8 image, sigma finite and > 0, window 1+2*ceil(3*sigma) <= CANNY_HIP_MAX_WINDOW.
 *   - xy_gradient / sobel / sobel_nms: height >= 2 and width >= 2 (the reference reads out of bounds
 *     below that).  Gradients are stored through short exactly like the reference.  Angle bins are
 *     computed with an exact integer rule that is proven equal to the reference's
 *     atan2/float expression for every |gx|,|gy| <= 1020, i.e. for every smoothed plane in [0,255]
 *     (everything gaussian() can produce); outside that range a gradient lying within float rounding
 *     of a bin boundary may be binned differently from the reference.
 *   - hysteresis / canny: if min_val <= 0 the reference's result depends on its scan order whenever a
 *     candidate is below min_val; that case returns CANNY_HIP_ERR_DOMAIN.  So does min_val > 255 >= max_val:
 *     the reference overwrites reached pixels with EDGE = 255 while its scan is still running, and a pixel the
 *     scan has not reached yet then fails `< minVal` and is zeroed again (src/utils.cpp:327-334) -- e.g.
 *     [[300,300,0,0]], min 300, max 100 gives [[255,0,0,0]].  find_edge_pixels rejects min_val > 255 likewise.
 *     (The reference's CLI only admits thresholds in [0,255], src/main.cpp:63-76.)
 *
 * Threading: a context is bound to one device and one stream and must be used by one host thread
 * at a time; different contexts may be used concurrently (one per GPU / per host thread).
 */
#ifndef CANNY_HIP_H
#define CANNY_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CANNY_HIP_VERSION 200        /* 0.2.0: + batch u8 / bit maps, multi-GPU options, host_register, async dev_canny */
#define CANNY_HIP_MAX_WINDOW 129     /* largest Gaussian window (sigma <= 21.33) */

typedef struct canny_hip_ctx canny_hip_ctx;

enum canny_hip_status {
    CANNY_HIP_OK = 0,
    CANNY_HIP_ERR_INVALID = 1,     /* null pointer, non-positive size, bad sigma ... */
    CANNY_HIP_ERR_UNSUPPORTED = 2, /* size or window beyond what the kernels support */
    CANNY_HIP_ERR_NO_DEVICE = 3,   /* no usable HIP device: there is NO CPU fallback */
    CANNY_HIP_ERR_RUNTIME = 4,     /* HIP runtime failure, see canny_hip_last_error() */
    CANNY_HIP_ERR_DOMAIN = 5,      /* input outside the documented numeric domain */
    CANNY_HIP_ERR_NO_CONVERGE = 6  /* hysteresis propagation hit its iteration cap */
};

/* Stages for the per-stage HIP-event profile (canny_hip_profile_*). */
enum canny_hip_stage {
    CANNY_HIP_STAGE_GAUSSIAN = 0,
    CANNY_HIP_STAGE_SOBEL_NMS = 1,      /* the fused Sobel+NMS pass (roofline-graded kernel) */
    CANNY_HIP_STAGE_HYST_CLASSIFY = 2,
    CANNY_HIP_STAGE_HYST_PROPAGATE = 3,
    CANNY_HIP_STAGE_HYST_FINALIZE = 4,
    CANNY_HIP_STAGE_SOBEL = 5,
    CANNY_HIP_STAGE_NMS = 6,
    CANNY_HIP_STAGE_XY_GRADIENT = 7,
    CANNY_HIP_STAGE_COUNT = 8
};

/* ---- library / context ------------------------------------------------------------------- */
int canny_hip_version(void);
const char *canny_hip_status_string(int status);
int canny_hip_device_count(int *count);

/* Creates a context on `device` with its own non-blocking stream. */
int canny_hip_ctx_create(canny_hip_ctx **ctx, int device);
void canny_hip_ctx_destroy(canny_hip_ctx *ctx);
/* Makes the context launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL
 * restores the context's own stream. */
int canny_hip_ctx_set_stream(canny_hip_ctx *ctx, void *hip_stream);
int canny_hip_ctx_device(const canny_hip_ctx *ctx);
/* Kernel-path selection, for A/B measurements and tests; every path gives identical results.
 *   "gaussian_path":  0 auto (default), 1 generic two-pass, 2 wave-marching (window <= 17 and width >= 4)
 *   "sobel_nms_path": 0 auto (default), 1 LDS-tiled, 2 wave-marching
 *   "gaussian_fma_div": 1 (default) / 0 -- single-fma division by the full-window weight (process-wide)
 *   "fuse_classify": 1 (default) / 0 -- canny(): the Sobel+NMS kernel writes the hysteresis bit-planes itself
 *                    (used when width % 8 == 0 and min_val >= 1; otherwise the separate kernels run)
 *   "smoothed_u8": 1 (default since round 3) / 0 -- canny(): the smoothed plane between the Gaussian and the fused Sobel+NMS
 *                    kernel is stored as bytes instead of shorts ((short)(sum/count) lies in [0,255],
 *                    src/utils.cpp:62): 5.25 instead of 7.25 algorithmic bytes per pixel through HBM.
 *                    Used when the fused path and the marching Gaussian apply, else ignored.
 *                    Same results bit for bit
 *   "hysteresis_tail": 1 (default) / 0 -- canny(): after two batch-wide propagation sweeps ONE launch with a workgroup
 *                    per frame runs the remaining sweeps to convergence (frames are independent, so a workgroup
 *                    barrier between a frame's sweeps is all the ordering needed).  The call then queues five
 *                    kernels and returns without waiting: no per-sweep launches, no host round trip.  Used for
 *                    frames of up to 4096 tiles of 64x64 (a 4K frame has 2040); 0 = the multi-launch scheme whose
 *                    host polls for convergence.  "tune_hyst_tail_after": 2 (default), 1..4 -- how many sweeps run
 *                    batch-wide before the tail kernel takes over
 *   "overlap_hysteresis": 0 (default) / 1 -- canny() on 16 or more frames: the propagation sweeps of the first half
 *                    of the batch run on a second stream beside the Sobel+NMS kernel of the second half
 *                    (measured 1.5 % slower on 128 x 4K, kept for A/B)
 *   "tune_batch_workers", "tune_batch_chunk_mb", "tune_batch_chunk_frames", "tune_batch_pipe_mode":
 *                    canny_hip_canny_batch's pipelines (host threads), the chunk size in megabytes of input or in
 *                    frames (frames win) and the stream structure of a pipeline (1 = upload, compute and download
 *                    stream chained by events, 2 = one in-order stream); 0 (default) = automatic: ONE three-stream
 *                    pipeline x 24 MB chunks, for pinned and for ordinary caller memory alike (round 3: pageable input
 *                    is staged by the library's thread pool, the output is written by it: "tune_batch_compact"; with
 *                    tune_batch_compact = 1 pageable buffers fall back to six single-stream pipelines x 8 MB).  HIP multiplexes a process's streams onto 4 hardware queues by
 *                    default (GPU_MAX_HW_QUEUES): a host application with many streams of its own should raise that
 *                    limit, or the three streams of the pipeline end up sharing a queue and serialise
 *   "stream_overlap": 0 (default) / 1 -- canny_hip_dev_canny_stream: the sweeps left in flight run on a second,
 *                    high-priority stream beside the next call's Gaussian instead of in order before it
 *                    (no gain on 128 x 4K batches, kept for A/B)
 *   "tune_sobel_seg": rows per wave segment of the marching Sobel+NMS kernel, 0 = automatic
 *   "tune_batch_compact": 0 (default) the s16 / u8 maps of the batch calls cross PCIe as 1-bit maps and are expanded
 *       into the caller's plane by host threads ("tune_batch_expand_threads": 0 = automatic, up to 8); 1 = the map
 *       itself is downloaded (rounds 1-2).  Same planes bit for bit
 *   "tune_sobel_px": 0 (default) 8 pixels per lane, 1 four pixels per lane (process-wide; the packed-i16 kernel only)
 *   "tune_sobel_variant": 0 (default) automatic -- the f32 marching arithmetic (4 waves per SIMD) for the fused
 *       Sobel+NMS+classify kernel of canny(), round 2's packed-i16 arithmetic (3 waves per SIMD) for the s16 -> s16
 *       stage kernel --, 1 packed-i16 everywhere, 2 f32 everywhere; same results bit for bit; process-wide
 *   "tune_plane_stores": 0 (default) the fused Sobel+NMS kernel parks a segment's plane bytes in LDS and writes
 *                    them as whole words after its last row, 1 direct byte stores (process-wide)
 *   "tune_gaussian_variant": 0 (default) symmetric-tap marching kernel, systolic row pass (the running sums travel
 *                    between lanes), row-pass products looked up in an LDS table; 1 LDS-ring marching kernel; 2
 *                    symmetric-tap kernel that multiplies and fetches its neighbours' products; 3 product-fetching row
 *                    pass with the table (the default of rounds 2-3); 4 systolic row pass that multiplies; same
 *                    results bit for bit (process-wide)
 *   "tune_gaussian_seg": approximate rows per wave segment of the marching Gaussian, 0 = automatic (process-wide)
 *   "tune_finalize_mode": 0 (default) row-major hysteresis finalize, 1 tile-patch finalize (process-wide)
 *   "profile_stage_mask": bit s set = stage s (CANNY_HIP_STAGE_*) gets an event pair while profiling is enabled;
 *                    0 (default) = all stages.  Every pair costs a few microseconds of stream time, so a timed
 *                    region that needs one kernel's duration enables that stage only.
 *   "profile_sample_interval": N >= 1 (default 1): only every N-th launch group of a stage gets its event pair */
int canny_hip_ctx_set_option(canny_hip_ctx *ctx, const char *name, int value);
/* Reads back "smoothed_u8", "fuse_classify", "hysteresis_tail", "gaussian_path", "sobel_nms_path", "tune_batch_compact",
 * the read-only "batch_expand_threads" (threads of the expansion pool once a batch call has created it) and the read-only
 * "last_canny_smoothed_u8": 1 if the context's last canny call really ran on the u8 smoothed plane (the option is a
 * request: windows beyond 17, asymmetric taps and shapes the fused kernel does not take fall back to the s16 plane).
 * bench.py uses it to price the kernel it timed with the bytes that kernel moved. */
int canny_hip_ctx_get_option(const canny_hip_ctx *ctx, const char *name, int *value);
int canny_hip_synchronize(canny_hip_ctx *ctx);
/* Text of the last HIP runtime error seen by this context ("" if none). */
const char *canny_hip_last_error(const canny_hip_ctx *ctx);
/* Number of propagation sweeps that did work in the last hysteresis / canny call (diagnostic).  After a call that only
 * queued its kernels (see canny_hip_dev_canny) the count still lives on the device: the getter then copies it out and
 * SYNCHRONISES the context's stream, hence the non-const context. */
int canny_hip_last_hysteresis_iterations(canny_hip_ctx *ctx);

/* ---- device / pinned memory helpers (for hosts without their own HIP allocator) ----------- */
int canny_hip_malloc(canny_hip_ctx *ctx, void **dev_ptr, size_t bytes);
int canny_hip_free(canny_hip_ctx *ctx, void *dev_ptr);
int canny_hip_host_alloc(canny_hip_ctx *ctx, void **host_ptr, size_t bytes); /* pinned */
int canny_hip_host_free(canny_hip_ctx *ctx, void *host_ptr);
/* Page-locks / releases memory the caller allocated itself (the reference's frames are `new[]` arrays and cv::Mat
 * data, src/main.cpp:111-137): registered buffers are DMA'd in place by the batch entry points, exactly like
 * canny_hip_host_alloc memory.  Register once, not per call (~22 ms per GB). */
int canny_hip_host_register(canny_hip_ctx *ctx, void *host_ptr, size_t bytes);
int canny_hip_host_unregister(canny_hip_ctx *ctx, void *host_ptr);
int canny_hip_memcpy_h2d(canny_hip_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int canny_hip_memcpy_d2h(canny_hip_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);

/* ---- stage entry points on HOST buffers (synchronous; one frame) --------------------------- */
/* createGaussianKernel: host-only.  taps must hold `cap` floats; *window receives 1+2*ceil(3*sigma). */
int canny_hip_gaussian_kernel(float sigma, float *taps, int cap, int *window);
int canny_hip_gaussian(canny_hip_ctx *ctx, const unsigned char *img, float sigma, int height, int width,
                       short *result);
int canny_hip_xy_gradient(canny_hip_ctx *ctx, const short *img, int height, int width, short *grad_x,
                          short *grad_y);
int canny_hip_sobel(canny_hip_ctx *ctx, const short *img, int height, int width, short *magnitude,
                    short *angle);
int canny_hip_nms(canny_hip_ctx *ctx, const short *magnitude, const short *angle, int height, int width,
                  short *result);
/* In place, like the reference. */
int canny_hip_hysteresis(canny_hip_ctx *ctx, short *edge_candidates, int height, int width, int min_val,
                         int max_val);
/* In place on both arrays; visited is one byte per pixel (C++ bool). */
int canny_hip_find_edge_pixels(canny_hip_ctx *ctx, short *edge_candidates, unsigned char *visited, int start,
                               int min_val, int max_val, int height, int width);
/* Whole pipeline; unlike the reference's canny() the {0,255} edge map is returned in `edges`. */
int canny_hip_canny(canny_hip_ctx *ctx, const unsigned char *img, float sigma, int min_val, int max_val,
                    int height, int width, short *edges);
/* n_frames contiguous frames in, n_frames edge maps out, frame i of the output = canny() of frame i (the
 * reference calls canny() once per captured frame, src/main.cpp:120-137).  The batch is cut into chunks; the
 * upload of chunk j+1, the kernels of chunk j and the download of chunk j-1 run concurrently on three streams
 * chained by events (BASELINE config 3).  Input frames in pinned / registered host memory (canny_hip_host_alloc, ...)
 * are DMA'd in place, ordinary (pageable) input is staged through pinned chunk buffers by the library's thread pool.
 * The finished maps cross PCIe as 1-bit maps and the same pool writes the caller's plane from them ("tune_batch_compact",
 * default), so the OUTPUT buffer needs no pinning: 48-53 Gpixel/s on 4K frames, bound by the upload (src/cuda.cu:83-101
 * moves every plane both ways in full). */
int canny_hip_canny_batch(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                          int max_val, int height, int width, short *edges);
/* Same, but the edge maps come back as 8-bit planes (NOEDGE = 0, EDGE = 255: the values of src/utils.h:5-6
 * fit a byte).  Not in the reference: its edge map is the s16 plane hysteresis() works in (src/utils.cpp:478);
 * over PCIe that plane is two thirds of all bytes moved, so batches that only need the final map should take
 * this one. */
int canny_hip_canny_batch_u8(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                             int max_val, int height, int width, unsigned char *edges);
/* Same, with the edge maps as BIT maps: 1 = EDGE, 0 = NOEDGE, rows packed MSB-first (pixel 0 of a row is bit 7 of the
 * row's first byte, as in PBM "P4" files and numpy.packbits) and padded to whole bytes, so frame i occupies
 * height * ((width + 7) / 8) bytes at bits + i * that.  The map only ever holds two values (src/utils.h:5-6), so
 * nothing is lost, and the download shrinks from 2 bytes per pixel to 1/8: the batch then runs at the rate frames can be
 * UPLOADED (one byte per pixel).  Not in the reference; the next step after SURVEY.md 8(f) item 2. */
int canny_hip_canny_batch_bits(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                               int max_val, int height, int width, unsigned char *bits);
/* Shards n_frames by contiguous ranges over n_devices GPUs (devices 0..n_devices-1), one host thread and one
 * context per GPU, each running the batch pipeline above on its shard; no collective (BASELINE config 5: the
 * reference has no multi-GPU path, frames are independent).  n_devices <= 0 = all.  The per-device contexts
 * (pipelines, streams, staging) are created on first use and kept until canny_hip_multi_gpu_release(); each
 * shard's threads are bound to the CPUs local to its GPU (sysfs local_cpulist) for the duration of the call.
 * One sharded call runs at a time per process.  Pinned caller buffers are DMA'd in place (allocate them on
 * the right NUMA node for best results); pageable ones are staged by each shard's thread pool. */
int canny_hip_canny_multi_gpu(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val,
                              int height, int width, short *edges, int n_devices);
int canny_hip_canny_multi_gpu_u8(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val,
                                 int height, int width, unsigned char *edges, int n_devices);
int canny_hip_canny_multi_gpu_bits(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val,
                                   int height, int width, unsigned char *bits, int n_devices);
/* Process-wide options of the sharder: "tune_batch_workers" / "tune_batch_chunk_mb" / "tune_batch_chunk_frames" /
 * "tune_batch_pipe_mode"
 * (applied to every shard's pipeline), "numa_affinity" 1 (default) / 0, "allow_device_reuse" 0 (default) / 1:
 * n_devices may exceed the device count, shard s then runs on device s % count (exercises the sharder with N > 1
 * on a one-GPU box; no use in production). */
int canny_hip_multi_gpu_set_option(const char *name, int value);
/* Destroys the cached per-device contexts. */
int canny_hip_multi_gpu_release(void);
/* CPUs local to `device` in sysfs list form ("0-31,128-159"); CANNY_HIP_ERR_UNSUPPORTED if the platform does not say. */
int canny_hip_device_local_cpus(int device, char *buf, int cap);
/* Frame range [begin, end) of shard `rank` of `world` (what canny_hip_canny_multi_gpu and bench.py use). */
int canny_hip_shard_range(int n_frames, int rank, int world, int *begin, int *end);

/* ---- stage entry points on DEVICE buffers (asynchronous on the context's stream) ----------- */
/* All planes hold n_frames contiguous frames.  Workspace is owned and grown by the context. */
int canny_hip_dev_gaussian(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int height, int width,
                           int n_frames, short *d_result);
int canny_hip_dev_xy_gradient(canny_hip_ctx *ctx, const short *d_img, int height, int width, int n_frames,
                              short *d_grad_x, short *d_grad_y);
int canny_hip_dev_sobel(canny_hip_ctx *ctx, const short *d_img, int height, int width, int n_frames,
                        short *d_magnitude, short *d_angle);
int canny_hip_dev_nms(canny_hip_ctx *ctx, const short *d_magnitude, const short *d_angle, int height, int width,
                      int n_frames, short *d_result);
/* Fused Sobel + NMS: smoothed s16 in, suppressed magnitude s16 out; magnitude and angle never reach
 * HBM (4 algorithmic bytes per pixel).  d_smoothed must lie in [0,255] (gaussian output). */
int canny_hip_dev_sobel_nms(canny_hip_ctx *ctx, const short *d_smoothed, int height, int width, int n_frames,
                            short *d_nms);
/* The two kernels of canny()'s "smoothed_u8" path on their own (tests, A/B): the Gaussian storing bytes
 * and the fused Sobel+NMS reading them (3 algorithmic bytes per pixel).
 * CANNY_HIP_ERR_UNSUPPORTED where the marching kernels do not apply (window > 17, width < 4, asymmetric taps, A/B
 * variants). */
int canny_hip_dev_gaussian_u8(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int height, int width,
                              int n_frames, unsigned char *d_result);
int canny_hip_dev_sobel_nms_u8in(canny_hip_ctx *ctx, const unsigned char *d_smoothed, int height, int width,
                                 int n_frames, short *d_nms);
/* In place.  Blocks the host until propagation has converged (it polls a device flag). */
int canny_hip_dev_hysteresis(canny_hip_ctx *ctx, short *d_edge_candidates, int height, int width, int n_frames,
                             int min_val, int max_val);
/* gaussian -> fused sobel+nms -> hysteresis over n_frames resident frames.
 * COMPLETION CONTRACT: d_edges is complete IN STREAM ORDER on the context's stream when the call has returned (work
 * queued on that stream afterwards sees the final map) and HOST-VISIBLE after canny_hip_synchronize() -- whatever path
 * ran.  Whether the call itself blocks the host depends on the shape: frames of <= 4096 hysteresis tiles (64x64 px)
 * with width % 8 == 0, min_val >= 1 and the option hysteresis_tail = 1 (default) only QUEUE their five kernels and
 * return; every other shape polls the propagation's convergence flag and returns when it has converged.  The context's
 * stream is non-blocking with respect to the legacy default stream: a hipMemcpy on the default stream does NOT wait for
 * it -- copy with canny_hip_memcpy_d2h (same stream), or synchronize first. */
int canny_hip_dev_canny(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                        int height, int width, int n_frames, short *d_edges);
/* canny() for a STREAM of batches -- the reference's capture loop (src/main.cpp:120-137: canny() on one frame
 * after the other) with resident batches in place of frames.  Same results as canny_hip_dev_canny, but the call
 * ALWAYS returns with the batch's hysteresis sweeps still queued, also for the shapes whose plain call has to poll the
 * convergence flag (large frames, hysteresis_tail = 0): that host round trip (which idles the GPU for ~20 us) happens
 * in the next call, after that call has queued its Gaussian.  d_edges of call i
 * is complete -- for work queued on the context's stream and, after a synchronize, for the host -- once call i+1
 * or canny_hip_dev_canny_stream_flush() has returned; until then the caller must neither read nor free it.
 * d_img may be reused as soon as the context's stream has passed the call.  Every other compute entry point of
 * the context, ctx_set_stream, synchronize and destroy flush first, so mixing the two kinds of call is safe
 * (canny_hip_memcpy_* do not flush: copying batch i-1 out while batch i is in flight is the point).  Shapes the fused Sobel+NMS+classify kernel
 * does not take (width % 8 != 0, min_val < 1) run as a plain canny_hip_dev_canny. */
int canny_hip_dev_canny_stream(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                               int height, int width, int n_frames, short *d_edges);
int canny_hip_dev_canny_stream_flush(canny_hip_ctx *ctx);
/* Same with an 8-bit edge map (0 / 255) as output; the s16 map is kept in a context workspace and narrowed
 * by one more elementwise kernel (this entry point exists for transfers, not for speed on the device). */
int canny_hip_dev_canny_u8(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                           int height, int width, int n_frames, unsigned char *d_edges);
/* ... and with a bit map as output (layout as canny_hip_canny_batch_bits: n_frames * height * ((width + 7) / 8) bytes). */
int canny_hip_dev_canny_bits(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                             int height, int width, int n_frames, unsigned char *d_bits);

/* ---- per-stage HIP-event timing (events are recorded on the launch stream) ----------------- */
int canny_hip_profile_enable(canny_hip_ctx *ctx, int on);
int canny_hip_profile_reset(canny_hip_ctx *ctx);
/* Synchronises the stream, then returns accumulated device milliseconds and launch count. */
int canny_hip_profile_get(canny_hip_ctx *ctx, int stage, double *total_ms, long *launches);

/* ---- self-test hooks used by the GPU test-suite -------------------------------------------- */
/* Runs the DEVICE magnitude / angle-bin functions over every (gx,gy) in [-lim,lim]^2 and writes
 * tables indexed [gy+lim][gx+lim] to host memory. */
int canny_hip_selftest_mag_angle(canny_hip_ctx *ctx, int lim, short *magnitudes, unsigned char *bins);
/* Measurement aid (bench.py): a plain device copy of nbytes (a multiple of 16; both pointers 16-byte aligned), launched
 * `launches` times on the context's stream; *avg_ms receives the average device time of one launch (HIP events attached
 * to the dispatch).  It calibrates what a 1:1 read/write stream reaches on THIS device beside the Sobel+NMS pass, whose
 * roofline is quoted against the 8 TB/s spec (the reference has no counterpart: src/cuda.cu:83-101 only ever copies
 * host<->device).  Synchronous. */
int canny_hip_probe_copy(canny_hip_ctx *ctx, const void *d_src, void *d_dst, size_t nbytes, int launches,
                         double *avg_ms);

/* Compares the Gaussian kernels' reciprocal-based division a/divisor with the IEEE divide for EVERY
 * float a in [0, 256] (1.13e9 values) on the device; *mismatches receives the number of differences and
 * *largest_mismatching_dividend the largest a that differed (0 if none). */
int canny_hip_selftest_div(canny_hip_ctx *ctx, float divisor, unsigned long long *mismatches,
                           float *largest_mismatching_dividend);
/* Same comparison for the one-instruction form a/divisor ~ fma(a, c, a) that the interior Gaussian waves
 * use when the full-window weight is within an ulp of 1 (c = 0 when it is exactly 1). */
int canny_hip_selftest_div_fma(canny_hip_ctx *ctx, float divisor, float c, unsigned long long *mismatches,
                               float *largest_mismatching_dividend);
/* Entry `index` of the built-in (divisor, c) table the kernels use; CANNY_HIP_ERR_INVALID past the end. */
int canny_hip_selftest_div_fma_table(int index, float *divisor, float *c);
/* Host-only (needs no device): the expansion step of the batch pipelines' compact transfer -- a bit map (rows MSB-first,
 * padded to bytes: height * ((width + 7) / 8) bytes) becomes the reference's short plane (0 / 255), or with to_u8 != 0 a
 * byte plane, written by n_threads pool threads exactly as canny_hip_canny_batch does it. */
int canny_hip_selftest_expand_bits(const unsigned char *bits, int height, int width, int to_u8, void *out, int n_threads);
/* Host-only: the order in which the waves of a marching launch (Gaussian, Sobel+NMS) take the n_segs x n_strips cells
 * of a frame -- border cells first, see march_cell_of in csrc/canny_kernels.h.  Writes n_segs * n_strips (segment, strip)
 * pairs to out_pairs. */
int canny_hip_selftest_march_order(int n_segs, int n_strips, int *out_pairs);
/* Host-only: number of CPUs in a sysfs-style list ("0-3,8,10-11" -> 7; 0 if malformed) -- the parser behind the
 * sharder's NUMA binding. */
int canny_hip_selftest_cpulist_count(const char *text);

#ifdef __cplusplus
}
#endif
#endif /* CANNY_HIP_H */
