// canny_kernels.h -- launchers for the gfx950 Canny kernels (internal; the public boundary is
// include/canny_hip.h).  All launchers are asynchronous on `stream` and return the launch status.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace canny {

constexpr int kMaxCenter = 64;                 // Gaussian half-window (window <= 129)
constexpr int kMaxWindow = 2 * kMaxCenter + 1;
constexpr int kTile = 64;                      // hysteresis bit-plane tile: 64 rows x 64 columns

// Gaussian taps exactly as the reference computes them on the host (src/utils.cpp:77-95); passed
// to kernels by value so they live in scalar registers / the kernarg segment.
struct GaussTaps {
    int center;
    float tap[kMaxWindow];
};

// Order in which the waves of a marching launch take the (segment, strip) cells of a frame: the frame's BORDER cells
// first (top and bottom segments, then the first and last strip of the segments in between), the interior cells after
// them.  Border cells run the border instantiations, which cost 1.1-1.7 x an interior cell; waves are dealt to SIMD
// slots in index order, so with the plain row-major order the last waves of a launch -- the bottom segment of the last
// frame -- were its most expensive ones and the chip waited for them with most slots empty.  Frames stay contiguous
// (and so do the interior cells of a frame, which share halo rows and columns in L2).  Bijective on [0, n_segs*n_strips).
struct MarchCell {
    int seg, strip;
};
__host__ __device__ inline MarchCell march_cell_of(int k, int n_segs, int n_strips)
{
    MarchCell c;
#ifdef CANNY_MARCH_ROW_MAJOR // A/B builds only: the plain order
    n_segs = 0;
#endif
    if (n_segs < 3 || n_strips < 3) { // every cell is a border cell
        c.seg = k / n_strips;
        c.strip = k % n_strips;
        return c;
    }
    const int sides = 2 * (n_segs - 2);
    if (k < n_strips) {
        c.seg = 0;
        c.strip = k;
    } else if (k < 2 * n_strips) {
        c.seg = n_segs - 1;
        c.strip = k - n_strips;
    } else if (k < 2 * n_strips + sides) {
        const int kk = k - 2 * n_strips;
        c.seg = 1 + (kk >> 1);
        c.strip = (kk & 1) ? n_strips - 1 : 0;
    } else {
        const int kk = k - 2 * n_strips - sides;
        c.seg = 1 + kk / (n_strips - 2);
        c.strip = 1 + kk % (n_strips - 2);
    }
    return c;
}

// Geometry of the hysteresis bit-planes: per frame tiles_y x tiles_x tiles, each 64 words (one
// 64-bit word = 64 consecutive pixels of one row), stored tile-major so that one wave reads a
// whole tile with one coalesced 512-byte load.
struct HystGeom {
    int height, width, n_frames;
    int tiles_x, tiles_y;
    __host__ __device__ size_t words() const { return (size_t)n_frames * tiles_x * tiles_y * kTile; }
    __host__ __device__ int tiles() const { return n_frames * tiles_x * tiles_y; }
};
inline HystGeom make_hyst_geom(int height, int width, int n_frames)
{
    HystGeom g;
    g.height = height;
    g.width = width;
    g.n_frames = n_frames;
    g.tiles_x = (width + kTile - 1) / kTile;
    g.tiles_y = (height + kTile - 1) / kTile;
    return g;
}

// ---- Gaussian (src/utils.cpp:26-68) ---------------------------------------------------------
// General two-pass path: any window up to kMaxWindow; tmp holds n_frames*H*W floats.
hipError_t launch_gaussian_generic(const uint8_t *img, float *tmp, int16_t *out, int height, int width,
                                   int n_frames, const GaussTaps &taps, hipStream_t stream);
// Wave-marching path for center <= 8 (window <= 17), row and column pass in one kernel: the symmetric-tap
// kernel (products shared between the +a / -a taps, column sums in registers) or, for asymmetric taps, the
// LDS-ring kernel (canny_gaussian_march.hip).
bool gaussian_march_supported(int center, int height, int width);
hipError_t launch_gaussian_march(const uint8_t *img, int16_t *out, int height, int width, int n_frames,
                                 const GaussTaps &taps, hipStream_t stream);

// The same kernel storing the smoothed plane as BYTES ((short)(sum/count) lies in [0,255], src/utils.cpp:62):
// half the store traffic, for the u8 form of the fused Sobel+NMS kernel.  Needs bit-symmetric taps and the default
// march variant.
bool gaussian_march_u8_supported(const GaussTaps &taps);
hipError_t launch_gaussian_march_u8(const uint8_t *img, uint8_t *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream);

// (S, c) bit-pattern pairs for which the interior waves divide by the full-window weight S with one
// fma(a, c, a); returns the number of entries.  gaussian_set_fma_div(false) disables the shortcut
// process-wide (A/B measurements only).
int gaussian_fma_div_table(const unsigned (**table)[2]);
void gaussian_set_fma_div(bool on);
// A/B: 0 = symmetric-tap kernel (shared products, register accumulators) with the row-pass products looked up
// in an LDS table (default), 1 = LDS-ring kernel, 2 = symmetric-tap kernel that multiplies.
void gaussian_set_march_variant(int v);
void gaussian_set_seg_target(int rows); // A/B: approximate rows per wave segment, 0 = automatic

// ---- Sobel / NMS (src/utils.cpp:106-308) ----------------------------------------------------
hipError_t launch_xy_gradient(const int16_t *img, int16_t *gx, int16_t *gy, int height, int width, int n_frames,
                              hipStream_t stream);
hipError_t launch_sobel(const int16_t *img, int16_t *mag, int16_t *angle, int height, int width, int n_frames,
                        hipStream_t stream);
hipError_t launch_nms(const int16_t *mag, const int16_t *angle, int16_t *out, int height, int width, int n_frames,
                      hipStream_t stream);
// Fused Sobel+NMS, LDS-tiled.  domain8 = smoothed plane known to lie in [0,255] (float sqrt + 24-bit
// products); otherwise the general path (double sqrt, 64-bit products).
hipError_t launch_sobel_nms(const int16_t *smoothed, int16_t *out, int height, int width, int n_frames,
                            bool domain8, hipStream_t stream);
// Fused Sobel+NMS, wave-marching and LDS-free (canny_sobel_nms_march.hip); smoothed must lie in [0,255].
bool sobel_nms_march_supported(int height, int width);
void sobel_nms_set_px_variant(int v); // A/B: 0 = 8 pixels per lane, 1 = 4 pixels per lane (more resident waves)
void sobel_nms_set_arith_variant(int v); // 0 = automatic (f32 for the fused classify kernel, packed-i16 for s16 -> s16), 1 = packed-i16, 2 = f32
// A/B (fused kernel): 0 = plane bytes staged in LDS for 8 rows and written as 8-byte words, 1 = direct byte stores
void sobel_nms_set_plane_store_variant(int v);
// An event pair attached to one kernel dispatch (hipExtLaunchKernel): that kernel's begin and end timestamps.
// Unlike hipEventRecord before and after the launch it puts no barrier packets into the stream (those cost
// ~30 us of stream time per pair).  Both null: plain launch.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
hipError_t launch_sobel_nms_march(const int16_t *smoothed, int16_t *out, int height, int width, int n_frames,
                                  hipStream_t stream, int tune_seg = 0, const LaunchEvents &ev = {});

// Fused Sobel+NMS+classify: the same marching kernel, but instead of the s16 suppressed magnitudes it
// writes the two hysteresis bit-planes that launch_hyst_classify would derive from them (in-image bytes
// only: pair it with launch_hyst_prepare(..., zero_pad = true)) and the provisional edge map `edges`
// (strong pixels -> edge_value, everything else 0) that launch_hyst_propagate(..., edges, edge_value) completes.
// Needs width % 8 == 0 and min_val >= 1.
bool sobel_nms_classify_supported(int height, int width, int min_val);
hipError_t launch_sobel_nms_classify_march(const int16_t *smoothed, int16_t *edges, uint64_t *strong, uint64_t *conn,
                                           const HystGeom &g, int min_val, int max_val, int edge_value,
                                           hipStream_t stream, int tune_seg = 0, const LaunchEvents &ev = {});

// The two marching kernels reading the smoothed plane as BYTES (8 pixels per lane, LDS-staged planes only).
bool sobel_nms_u8_input_supported();
hipError_t launch_sobel_nms_march_u8in(const uint8_t *smoothed, int16_t *out, int height, int width, int n_frames,
                                       hipStream_t stream, int tune_seg = 0, const LaunchEvents &ev = {});
hipError_t launch_sobel_nms_classify_march_u8in(const uint8_t *smoothed, int16_t *edges, uint64_t *strong,
                                                uint64_t *conn, const HystGeom &g, int min_val, int max_val,
                                                int edge_value, hipStream_t stream, int tune_seg = 0,
                                                const LaunchEvents &ev = {});

// ---- Hysteresis (src/utils.cpp:322-427) -----------------------------------------------------
hipError_t launch_hyst_classify(const int16_t *cand, uint64_t *strong, uint64_t *conn, const HystGeom &g, int min_val,
                                int max_val, unsigned *domain_flag, hipStream_t stream);
// One propagation sweep (`iter` = 0,1,2,...).  sched holds hyst_sched_words(g) words (tile stamps, two batch-wide
// work queues with three counters, two per-frame work queues with four counter words per frame) and, like the
// single word last_change, must be zero before sweep 0.
inline size_t hyst_sched_words(const HystGeom &g) { return 5 * (size_t)g.tiles() + 4 + 4 * (size_t)g.n_frames; }
// First launch of a hysteresis call: zeroes sched (hyst_sched_words(g) words) and flags[0..1] (last_change,
// domain) and, if zero_pad, the plane bits outside the image (tile padding; needed when the planes are filled
// by launch_sobel_nms_classify_march, which writes in-image bytes only; requires width % 8 == 0).
// n_lanes > 1: the frames are propagated as n_lanes independent ranges (each with its own scheduling words, laid
// out back to back -- 3 * tiles + 4 * n_lanes words in all -- and its own pair of flags: 2 * n_lanes words).
hipError_t launch_hyst_prepare(uint64_t *strong, uint64_t *conn, const HystGeom &g, bool zero_pad, unsigned *sched,
                               unsigned *flags, hipStream_t stream, int n_lanes = 1);
// edges != nullptr: an edge map that already holds the initially strong pixels; each sweep writes edge_value
// into the pixels it promotes, so the map is final when propagation has converged (no finalize pass).
// to_frame_queues: the tiles this sweep schedules go into their frame's queue (the sweep before launch_hyst_tail).
hipError_t launch_hyst_propagate(uint64_t *strong, const uint64_t *conn, unsigned *sched, unsigned *last_change,
                                 int iter, const HystGeom &g, hipStream_t stream, int16_t *edges = nullptr,
                                 int edge_value = 0, bool to_frame_queues = false);
// Every sweep from first_iter on, to convergence, in ONE launch: one workgroup per frame walks that frame's queue
// with a workgroup barrier between sweeps (frames are independent; see hyst_tail_kernel).  The sweep first_iter - 1
// must have been launched with to_frame_queues = true.  No host round trip: the propagation is complete when the
// stream has passed this kernel.
hipError_t launch_hyst_tail(uint64_t *strong, const uint64_t *conn, unsigned *sched, unsigned *last_change,
                            int first_iter, const HystGeom &g, hipStream_t stream, int16_t *edges = nullptr,
                            int edge_value = 0);
// s16 edge map (0 / 255) -> u8, n pixels.
hipError_t launch_edges_to_u8(const int16_t *edges, uint8_t *out, size_t n, hipStream_t stream);
// s16 edge map -> packed bit map (1 = pixel != 0), rows MSB-first and padded to bytes: [n][height][(width + 7) / 8]
hipError_t launch_edges_to_bits(const int16_t *edges, uint8_t *bits, int height, int width, int n_frames,
                                hipStream_t stream);
// Copies flags[0..1] (last_change, domain) to host_flags_dev[0..1] and then stores seq to host_flags_dev[2]
// (system-scope release); host_flags_dev is the device view of pinned, mapped host memory.
hipError_t launch_hyst_publish(const unsigned *flags, unsigned *host_flags_dev, unsigned seq, hipStream_t stream);
hipError_t launch_hyst_finalize(int16_t *cand, const uint64_t *strong, const HystGeom &g, int edge_value,
                                hipStream_t stream);
void hyst_set_finalize_mode(int mode); // A/B: 0 = row-major kernel (default), 1 = 8-row patch kernel
// findEdgePixels (single frame): seed = {start}, connectable = cand >= min_val && !visited.
hipError_t launch_fep_classify(const int16_t *cand, const uint8_t *visited, uint64_t *strong, uint64_t *conn,
                               const HystGeom &g, int start, int min_val, hipStream_t stream);
hipError_t launch_fep_finalize(int16_t *cand, uint8_t *visited, const uint64_t *strong, const HystGeom &g, int start,
                               int min_val, hipStream_t stream);

// ---- measurement aid ------------------------------------------------------------------------
// Plain device copy of nbytes (multiple of 16; both pointers 16-byte aligned): what a 1:1 read/write stream reaches.
hipError_t launch_probe_copy(const void *src, void *dst, size_t nbytes, hipStream_t stream, const LaunchEvents &ev = {});

// ---- self-test ------------------------------------------------------------------------------
hipError_t launch_selftest_mag_angle(int lim, int16_t *mags, uint8_t *bins, hipStream_t stream);
// Counts floats a (bit patterns first_bits..last_bits) for which the Gaussian's reciprocal-based
// division a/b differs from the IEEE divide; *d_mismatches must be zero beforehand.
// use_fma != 0 checks the one-instruction form fma(a, c, a) instead of the 5-op reciprocal division.
hipError_t launch_selftest_div(float b, int use_fma, float c, unsigned first_bits, unsigned last_bits,
                               unsigned long long *d_mismatches, hipStream_t stream);

} // namespace canny
