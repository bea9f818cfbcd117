// canny_capi.hip -- implementation of the C ABI declared in include/canny_hip.h.
//
// Host-side runtime around the kernels in canny_kernels.hip: contexts (one device + one stream),
// device workspaces that grow on demand and are reused across calls, HIP-event stage timers, the
// hysteresis convergence loop, the stream-overlapped batch path and the per-GPU sharder.
// There is no CPU implementation behind any of these entry points: without a HIP device every
// call fails with CANNY_HIP_ERR_NO_DEVICE.
#include "canny_hip.h"
#include "canny_kernels.h"

#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

using namespace canny;

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
// queue serialise.  The batch pipeline needs its upload, compute and download streams on separate queues beside
// whatever streams the host application has (measured: 25.3 -> 16.7 Gpix/s with six streams on four queues).  The
// variable belongs to the APPLICATION: it has to be in the environment before the HIP runtime initialises
// (bench.py and Main set it).  The library does not touch the process environment (round 2 did, from
// a constructor: a process-wide side effect that raced getenv() in other threads of a plugin host).

namespace {

// "0-3,8,10-11" -> cpu_set_t; returns the number of CPUs set (0 on a malformed list)
int parse_cpulist(const char *text, cpu_set_t *set)
{
    CPU_ZERO(set);
    int count = 0;
    const char *p = text;
    while (*p) {
        while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') p++;
        if (!*p) break;
        char *end = nullptr;
        long a = std::strtol(p, &end, 10);
        if (end == p || a < 0) return 0;
        long b = a;
        p = end;
        if (*p == '-') {
            p++;
            b = std::strtol(p, &end, 10);
            if (end == p || b < a) return 0;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET((int)c, set);
            count++;
        }
    }
    return count;
}

// CPUs local to a GPU, from sysfs (/sys/bus/pci/devices/<bdf>/local_cpulist, e.g. "0-31,128-159")
bool device_local_cpus(int device, cpu_set_t *set)
{
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    for (char *c = bdf; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = std::fgets(line, sizeof line, f) != nullptr;
    std::fclose(f);
    if (!got) return false;
    return parse_cpulist(line, set) > 0;
}

// hipHostMalloc with the pages next to `device` (see canny_hip_host_alloc)
hipError_t numa_host_malloc(int device, void **host_ptr, size_t bytes)
{
    cpu_set_t local, saved;
    const bool rebind = device >= 0 && device_local_cpus(device, &local) &&
                        sched_getaffinity(0, sizeof saved, &saved) == 0 && sched_setaffinity(0, sizeof local, &local) == 0;
    hipError_t e = hipHostMalloc(host_ptr, bytes ? bytes : 1);
    if (e == hipSuccess && bytes) { // first touch, one byte per page
        volatile unsigned char *p = (volatile unsigned char *)*host_ptr;
        for (size_t off = 0; off < bytes; off += 4096) p[off] = 0;
    }
    if (rebind) (void)sched_setaffinity(0, sizeof saved, &saved);
    return e;
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) return e;
        }
        // grow with a little head-room so slowly growing batches do not reallocate every call
        size_t want = need + need / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct PinBuf { // page-locked host staging
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need, int device = -1)
    {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = numa_host_malloc(device, &p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// ---- host-side expansion of bit maps (the "compact" transfer of the batch pipelines) -------------------------------
// The edge map only ever holds EDGE and NOEDGE (src/utils.h:5-6), so a batch call does not have to move the
// reference's 2 bytes per pixel over PCIe: the device packs the finished map into one bit per pixel (rows MSB-first,
// padded to bytes: launch_edges_to_bits), that travels -- 1/16 of the s16 map --, and a small pool of host threads
// writes the caller's short (or byte) plane from it: one 16-byte (8-byte) streaming store per bit-map byte out of a
// 256-entry table.  The s16 batch was bound by the DOWNLOAD (51 GB/s D2H for 25.5 Gpix/s); this way it runs at the
// rate frames can be UPLOADED, like the bit-map API, and the caller still gets the reference's plane, bit for bit.
class ExpandPool {
public:
    struct Job {
        const uint8_t *bits; // bit rows of the block (row_bytes each) -- or, for a copy job, the source bytes
        void *dst;           // first pixel of the block in the caller's plane -- or the copy's destination
        int rows, width, row_bytes;
        bool to_u8;          // bytes instead of shorts
        std::atomic<int> *left; // jobs of the chunk still to do
        size_t copy_bytes = 0;  // != 0: a plain memcpy of that many bytes (staging of pageable input frames)
    };
    // dst[0, bytes) = src[0, bytes), split over the pool in 1 MB pieces; returns when it is done (the caller works too)
    void parallel_copy(void *dst, const void *src, size_t bytes)
    {
        constexpr size_t kPiece = 1u << 20;
        const int n = (int)((bytes + kPiece - 1) / kPiece);
        if (n <= 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::atomic<int> left{n};
        for (int i = 0; i < n; i++) {
            Job j{};
            j.bits = (const uint8_t *)src + (size_t)i * kPiece;
            j.dst = (uint8_t *)dst + (size_t)i * kPiece;
            j.copy_bytes = std::min(kPiece, bytes - (size_t)i * kPiece);
            j.left = &left;
            submit(j);
        }
        wait(left);
    }
    explicit ExpandPool(int n_threads)
    {
        for (int b = 0; b < 256; b++)
            for (int k = 0; k < 8; k++) {
                const bool on = (b & (0x80 >> k)) != 0; // MSB first: bit 7 is the row's first pixel of this byte
                lut16_[b][k] = on ? 255 : 0;            // EDGE / NOEDGE
                lut8_[b][k] = on ? 255 : 0;
            }
        for (int i = 0; i < n_threads; i++) threads_.emplace_back([this] { loop(); });
    }
    ~ExpandPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (auto &t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    void submit(const Job &j)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(j);
        }
        cv_job_.notify_one();
    }
    // blocks until `left` has reached zero; the waiting thread works on queued jobs meanwhile
    void wait(std::atomic<int> &left)
    {
        while (left.load(std::memory_order_acquire) > 0) {
            Job j;
            bool have = false;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!q_.empty()) {
                    j = q_.front();
                    q_.pop_front();
                    have = true;
                }
            }
            if (have) {
                run(j);
            } else {
                std::unique_lock<std::mutex> lk(mu_);
                cv_done_.wait_for(lk, std::chrono::microseconds(50),
                                  [&] { return left.load(std::memory_order_acquire) <= 0 || !q_.empty(); });
            }
        }
    }

private:
    void loop()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_job_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return; // stop_
                j = q_.front();
                q_.pop_front();
            }
            run(j);
        }
    }
    void run(const Job &j)
    {
        if (j.copy_bytes) std::memcpy(j.dst, j.bits, j.copy_bytes);
        for (int r = 0; r < (j.copy_bytes ? 0 : j.rows); r++) {
            const uint8_t *src = j.bits + (size_t)r * j.row_bytes;
            const int full = j.width / 8, tail = j.width % 8;
            if (j.to_u8) {
                uint8_t *d = (uint8_t *)j.dst + (size_t)r * j.width;
                for (int b = 0; b < full; b++) std::memcpy(d + 8 * b, lut8_[src[b]], 8);
                if (tail) std::memcpy(d + 8 * full, lut8_[src[full]], tail);
            } else {
                short *d = (short *)j.dst + (size_t)r * j.width;
#if defined(__SSE2__)
                if ((((uintptr_t)d) & 15) == 0) { // streaming stores: the plane is written once and not read here
                    for (int b = 0; b < full; b++)
                        _mm_stream_si128((__m128i *)(d + 8 * b), _mm_load_si128((const __m128i *)lut16_[src[b]]));
                } else
#endif
                {
                    for (int b = 0; b < full; b++) std::memcpy(d + 8 * b, lut16_[src[b]], 16);
                }
                if (tail) std::memcpy(d + 8 * full, lut16_[src[full]], 2 * tail);
            }
        }
#if defined(__SSE2__)
        _mm_sfence();
#endif
        if (j.left->fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(mu_); // pairs with wait(): no lost wake-up
            cv_done_.notify_all();
        }
    }
    alignas(16) short lut16_[256][8];
    alignas(8) uint8_t lut8_[256][8];
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    std::deque<Job> q_;
    bool stop_ = false;
};

struct EventPair {
    hipEvent_t a, b;
};

// One propagation in flight: a frame range with its planes, scheduling words, flags and the stream it runs on.
struct PropLane {
    hipStream_t stream = nullptr;
    uint64_t *S = nullptr;
    const uint64_t *C = nullptr;
    unsigned *sched = nullptr, *flags = nullptr; // device: tile stamps + queues + counters; last_change, domain
    unsigned *host = nullptr, *host_dev = nullptr; // 4 pinned words: last_change, domain, sequence number, spare
    hipEvent_t event = nullptr;                    // recorded behind every publish (fallback wait, stream join)
    HystGeom g{};
    short *edges = nullptr;
    int edge_value = 0;
    int iter = 0;       // sweeps launched so far
    unsigned seq = 0;   // sequence number of the last publish
    bool converged = false;
};

} // namespace

struct canny_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    int last_hyst_iters = 0;
    int gaussian_path = 0;   // 0 auto, 1 generic, 2 march
    int sobel_nms_path = 0;  // 0 auto, 1 LDS tile, 2 march
    int tune_sobel_seg = 0;  // A/B knob of the marching Sobel+NMS kernel: rows per segment, 0 = automatic
    int fuse_classify = 1;   // canny(): Sobel+NMS emits the hysteresis bit-planes directly when it can
    // canny(): the smoothed plane between the Gaussian and the fused Sobel+NMS kernel as bytes (its values lie in
    // [0,255], src/utils.cpp:62): 0 = s16 plane, 1 = u8 plane
    int smoothed_u8 = 1; // canny(): the plane between the Gaussian and the fused Sobel+NMS as bytes (round 3 default)
    int last_canny_u8 = 0; // whether the last canny call really used it (window, taps and shape permitting)
    // canny(): after the two batch-wide sweeps, one launch with a workgroup per frame finishes the propagation
    // (launch_hyst_tail): no further launches, no host round trip, the call returns without waiting.
    // 1 (default) = for frames of up to kTailMaxTiles tiles, 0 = never (the multi-launch scheme with its poll)
    int hyst_tail = 1;
    int hyst_tail_after = 2; // batch-wide sweeps before the tail kernel takes over (A/B: 2 or 3)
    bool hyst_iters_async = false; // last_hyst_iters has to be fetched from flags[0] (the tail path does not poll)
    // canny_hip_canny_batch: number of pipelines (host threads, each with an H2D, a compute and a D2H stream and a
    // ring of chunk slots) that each take every n-th chunk, and the chunk size (megabytes of input, or frames).
    // 0 = automatic, see canny_batch_impl.
    int batch_workers = 0;
    int batch_chunk_mb = 0;
    int batch_chunk_frames = 0;
    int batch_pipe_mode = 0; // 0 = automatic, 1 = three streams per pipeline, 2 = one in-order stream per pipeline
    // s16 / u8 batch maps travel as bit maps and are expanded by host threads (ExpandPool): 0 = automatic (on),
    // 1 = off (the map itself is downloaded, as in rounds 1-2); expand_threads: 0 = automatic
    int batch_compact = 0;
    int batch_expand_threads = 0;
    std::unique_ptr<ExpandPool> expand_pool;
    // the pipelines live as long as the context: creating a sub-context with its streams and allocating its
    // staging costs ~10 ms per call, a sixth of a 1024 x 1080p batch
    struct BatchPipe;
    std::vector<BatchPipe *> batch_pool;

    // device workspaces
    DevBuf tmp_f32;   // generic Gaussian row-pass plane
    DevBuf smoothed;  // pipeline: Gaussian output
    DevBuf edges16;   // canny_hip_dev_canny_u8: the s16 edge map before narrowing
    DevBuf plane_s, plane_c, stamps, flags; // hysteresis bit-planes / scheduling words
    DevBuf io[4];     // staging for the host-pointer stage functions
    unsigned *host_flags = nullptr;     // pinned + mapped, 4 words per lane: last_change, domain, sequence number, spare
    unsigned *host_flags_dev = nullptr; // the same memory as the device sees it
    unsigned publish_seq = 0;           // sequence number of the last launch_hyst_publish
    hipEvent_t flag_event = nullptr;    // recorded behind the publish kernel (fallback wait)
    // canny(), optional: the propagation of the first half of a batch runs on a second stream, beside the
    // Sobel+NMS kernel of the second half (the sweeps are latency bound and leave most of the chip idle).  Off by
    // default: 128 x 4K measured 2.76 ms with it against 2.72 ms without -- the Sobel+NMS kernel loses more
    // (two half-size launches, sweeps competing for its CUs) than the hidden sweeps give back.
    int overlap_hysteresis = 0;
    int stream_overlap = 0; // canny_hip_dev_canny_stream: sweeps on the second stream (see dev_canny_stream)
    hipStream_t aux_stream = nullptr;
    hipEvent_t fork_event = nullptr, aux_event = nullptr;
    // canny_hip_dev_canny_stream: the propagation of the batch submitted last, still in flight on aux_stream
    PropLane pend;
    bool has_pend = false;

    // profiling
    bool prof = false;
    unsigned prof_mask = ~0u; // stages whose launches get an event pair (each pair costs a few us of stream time)
    unsigned prof_every = 1;  // ... and only every prof_every-th launch group of a stage gets one
    unsigned prof_seen[CANNY_HIP_STAGE_COUNT] = {0};
    std::vector<EventPair> pending[CANNY_HIP_STAGE_COUNT];
    std::vector<EventPair> pool;
    double total_ms[CANNY_HIP_STAGE_COUNT] = {0};
    long launches[CANNY_HIP_STAGE_COUNT] = {0};
};

// One batch pipeline: three streams and a ring of chunk slots.  Chunk j of the pipeline lives in slot j % kSlots:
//   s_h2d:    host -> d_in                       (ev_h2d)
//   compute:  d_in -> d_out [-> d_out8]           (ev_comp; the sub-context's stream, waits for ev_h2d)
//   s_d2h:    d_out / d_out8 -> host              (ev_d2h; waits for ev_comp)
// so that the upload of chunk j+1, the kernels of chunk j and the download of chunk j-1 are in flight together and
// neither DMA engine waits for a host thread.  Pageable caller buffers go through the slot's pinned staging.
// HIP multiplexes a process's streams onto a handful of hardware queues (4 by default, GPU_MAX_HW_QUEUES), and
// streams that share a queue serialise: measured on 128 x 4K, the same pipeline ran at 25.3 Gpix/s with four streams
// alive in the process and at 16.7 with six.  So streams are created sparingly: pipeline 0 computes on the parent
// context itself (its stream is idle during a batch call), the copy streams exist only in three-stream mode.
struct canny_hip_ctx::BatchPipe {
    static constexpr int kSlots = 3;
    canny_hip_ctx *sub = nullptr; // compute context: the parent itself for pipeline 0, an owned sub-context otherwise
    bool owns_sub = false;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr; // three-stream mode only, created on first use
    struct Slot {
        DevBuf d_in, d_out, d_out8;
        PinBuf pin_in, pin_out;
        hipEvent_t ev_h2d = nullptr, ev_comp = nullptr, ev_d2h = nullptr;
        bool d2h_issued = false;      // ev_d2h has been recorded during the current call
        void *retire_dst = nullptr;   // pageable output: where pin_out goes once ev_d2h has fired
        size_t retire_bytes = 0;
        int retire_frames = 0;        // compact transfer: frames whose bit maps sit in pin_out
        std::atomic<int> expand_left{0}; // compact transfer: expansion jobs of this slot's chunk still running
    } slot[kSlots];
};

namespace {

int fail(canny_hip_ctx *ctx, hipError_t e, const char *where)
{
    if (ctx) {
        ctx->last_error = std::string(where) + ": " + hipGetErrorString(e);
    }
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? CANNY_HIP_ERR_NO_DEVICE : CANNY_HIP_ERR_RUNTIME;
}

#define HIP_TRY(ctx, expr)                                  \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return fail((ctx), e_, #expr); \
    } while (0)

int bind(canny_hip_ctx *ctx)
{
    if (!ctx) return CANNY_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return CANNY_HIP_OK;
}

// RAII stage timer: brackets the launches of one stage with events on the launch stream.
struct StageTimer {
    canny_hip_ctx *ctx;
    int stage;
    EventPair ev{};
    bool on = false;
    hipStream_t stream;
    bool attach; // the events are not recorded around the launches but attached to ONE dispatch (launch_events())
    StageTimer(canny_hip_ctx *c, int s, hipStream_t on_stream = nullptr, bool attached = false)
        : ctx(c), stage(s), stream(on_stream ? on_stream : c->stream), attach(attached)
    {
        if (!ctx->prof || !(ctx->prof_mask >> s & 1u)) return;
        if (ctx->prof_seen[s]++ % ctx->prof_every != 0) return;
        if (!ctx->pool.empty()) {
            ev = ctx->pool.back();
            ctx->pool.pop_back();
        } else {
            if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
        }
        on = attach || hipEventRecord(ev.a, stream) == hipSuccess;
    }
    LaunchEvents launch_events() const { return (on && attach) ? LaunchEvents{ev.a, ev.b} : LaunchEvents{}; }
    ~StageTimer()
    {
        if (!on) return;
        if (!attach) (void)hipEventRecord(ev.b, stream);
        ctx->pending[stage].push_back(ev);
    }
};

// createGaussianKernel, reference src/utils.cpp:77-95, evaluated on the host with the reference's
// exact expression types: float window rule, expf in float, the 1/(sqrt(2pi) sigma) factor in
// double, a float running sum and a float divide per tap.  (Built with -ffp-contract=off.)
int make_taps(float sigma, GaussTaps &t)
{
    if (!(sigma > 0.0f) || !std::isfinite(sigma)) return CANNY_HIP_ERR_INVALID;
    float wf = 1 + 2 * std::ceil(3 * sigma);
    if (!(wf >= 1.0f) || wf > (float)kMaxWindow) return CANNY_HIP_ERR_UNSUPPORTED;
    int window = (int)wf;
    int center = window / 2;
    float total = 0.0f;
    for (int i = 0; i < window; i++) {
        float x = (float)(i - center);
        float e = expf(-((x * x) / (2 * sigma * sigma)));
        float tap = (float)((double)e / (std::sqrt(6.2831853) * (double)sigma));
        t.tap[i] = tap;
        total += tap;
    }
    for (int i = 0; i < window; i++) t.tap[i] /= total;
    for (int i = window; i < kMaxWindow; i++) t.tap[i] = 0.0f;
    t.center = center;
    return CANNY_HIP_OK;
}

int check_dims(int height, int width, int n_frames)
{
    if (height < 1 || width < 1 || n_frames < 1) return CANNY_HIP_ERR_INVALID;
    if ((long long)height * width > 0x7fffffffLL) return CANNY_HIP_ERR_UNSUPPORTED;
    if (n_frames > 65535) return CANNY_HIP_ERR_UNSUPPORTED;
    return CANNY_HIP_OK;
}

// Reached pixels are overwritten with EDGE = 255 while the reference's scan is still running (src/utils.cpp:327-334,
// 367).  With min_val > 255 such a pixel then fails the scan's own `< minVal` test when the scan reaches it and
// is zeroed again -- unless the scan has already passed it: the result depends on the scan order, which the
// parallel formulation does not have.  (If max_val > 255 as well everything ends as 0 either way.)
bool hysteresis_order_dependent(int min_val, int max_val) { return min_val > 255 && max_val <= 255; }

size_t npx(int height, int width, int n_frames) { return (size_t)height * (size_t)width * (size_t)n_frames; }

// ---- device-level stages ------------------------------------------------------------------------
// Can the Gaussian hand its result to Sobel+NMS as bytes?  (marching symmetric-tap kernel on both sides)
bool gaussian_u8_possible(const canny_hip_ctx *ctx, const GaussTaps &taps, int h, int w)
{
    float min_tap = 1.0f;
    for (int k = 0; k <= 2 * taps.center; k++)
        if (taps.tap[k] > 0.0f && taps.tap[k] < min_tap) min_tap = taps.tap[k];
    return ctx->gaussian_path != 1 && gaussian_march_supported(taps.center, h, w) && min_tap >= 0x1p-48f &&
           gaussian_march_u8_supported(taps) && sobel_nms_u8_input_supported();
}

// u8_mode: 0 = s16 plane in d_out; 1 = u8 plane in d_out, only if gaussian_u8_possible()
int dev_gaussian(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int h, int w, int n, void *d_out_any,
                 int u8_mode = 0)
{
    short *d_out = (short *)d_out_any;
    GaussTaps taps;
    int rc = make_taps(sigma, taps);
    if (rc) return rc;
    StageTimer tm(ctx, CANNY_HIP_STAGE_GAUSSIAN);
    if (u8_mode) {
        if (!gaussian_u8_possible(ctx, taps, h, w)) return CANNY_HIP_ERR_UNSUPPORTED;
        HIP_TRY(ctx, launch_gaussian_march_u8(d_img, (uint8_t *)d_out_any, h, w, n, taps, ctx->stream));
        return CANNY_HIP_OK;
    }
    // The marching kernel divides through a precomputed reciprocal, which equals the IEEE quotient for every
    // dividend >= 2^-102 (canny_hip_selftest_div).  Non-zero dividends are >= min_tap (row sums, pixels >= 1)
    // and >= ~min_tap^2 (column sums), so min_tap >= 2^-48 keeps them all above 2^-97; narrower taps
    // (sigma < ~0.13) take the generic kernels, which use the IEEE divide itself.
    float min_tap = 1.0f;
    for (int k = 0; k <= 2 * taps.center; k++)
        if (taps.tap[k] > 0.0f && taps.tap[k] < min_tap) min_tap = taps.tap[k];
    const bool can_march = gaussian_march_supported(taps.center, h, w) && min_tap >= 0x1p-48f;
    if (ctx->gaussian_path == 2 && !can_march) return CANNY_HIP_ERR_UNSUPPORTED;
    if (can_march && ctx->gaussian_path != 1) {
        HIP_TRY(ctx, launch_gaussian_march(d_img, d_out, h, w, n, taps, ctx->stream));
    } else {
        HIP_TRY(ctx, ctx->tmp_f32.ensure(npx(h, w, n) * sizeof(float)));
        HIP_TRY(ctx, launch_gaussian_generic(d_img, (float *)ctx->tmp_f32.p, d_out, h, w, n, taps, ctx->stream));
    }
    return CANNY_HIP_OK;
}

int ensure_hyst(canny_hip_ctx *ctx, const HystGeom &g)
{
    HIP_TRY(ctx, ctx->plane_s.ensure(g.words() * sizeof(uint64_t)));
    HIP_TRY(ctx, ctx->plane_c.ensure(g.words() * sizeof(uint64_t)));
    HIP_TRY(ctx, ctx->stamps.ensure((hyst_sched_words(g) + 4) * sizeof(unsigned))); // room for two lanes
    HIP_TRY(ctx, ctx->flags.ensure(4 * sizeof(unsigned)));
    if (!ctx->host_flags) {
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->host_flags, 8 * sizeof(unsigned), hipHostMallocMapped));
        std::memset(ctx->host_flags, 0, 8 * sizeof(unsigned));
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->host_flags_dev, ctx->host_flags, 0));
    }
    if (!ctx->flag_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->flag_event, hipEventDisableTiming));
    return CANNY_HIP_OK;
}

int ensure_aux_stream(canny_hip_ctx *ctx)
{
    if (!ctx->aux_stream) {
        // highest priority: what runs here is short and latency bound, and the main stream's kernels fill the chip
        int least = 0, greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, greatest));
    }
    if (!ctx->fork_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
    if (!ctx->aux_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming));
    return CANNY_HIP_OK;
}

// First launch of a hysteresis call: clears the scheduling words and the flags (flags[0] = last_change,
// flags[1] = domain) and, for planes filled by the Sobel+NMS kernel, the tile padding.
int prepare_hyst(canny_hip_ctx *ctx, const HystGeom &g, bool zero_pad)
{
    HIP_TRY(ctx, launch_hyst_prepare((uint64_t *)ctx->plane_s.p, (uint64_t *)ctx->plane_c.p, g, zero_pad,
                                     (unsigned *)ctx->stamps.p, (unsigned *)ctx->flags.p, ctx->stream));
    return CANNY_HIP_OK;
}

// Runs propagation sweeps until no tile was re-stamped, then (or meanwhile) the consumer of the strong plane.
// Sweeps are launched eight at a time before the host looks at the flag: natural images converge in 5-10
// sweeps, and a sweep with an empty queue costs ~4 us while a host round trip costs ~25 us.
// speculative: the consumer is a pure function of the strong plane that overwrites all of its output (the
// finalize kernel), so it is launched right behind each chunk of sweeps, BEFORE the host has seen the flag;
// the host waits on an event recorded behind the flag copy only, and the GPU never idles for the round trip.
// If the flag says "not converged" (rare) the consumer simply runs again behind the next chunk.
// edges != nullptr: the sweeps write the pixels they promote straight into that edge map (which must already
// hold the strong pixels); there is then nothing left for a consumer to do.
constexpr int kTailMaxTiles = 4096; // per frame (a 4K frame has 2040): beyond that one workgroup per frame is too few
constexpr int kSweepChunk = 8;
constexpr int kMaxSweeps = 1 << 22;

// Launches the next chunk of sweeps of a lane and, behind them, the one-thread kernel that publishes the two
// flag words and a sequence number in pinned host memory (the host spins on the sequence number: ~5 us from the
// last sweep to the host knowing, against ~35 us for an async copy plus an event wait).
int lane_launch_chunk(canny_hip_ctx *ctx, PropLane &L)
{
    {
        StageTimer tm(ctx, CANNY_HIP_STAGE_HYST_PROPAGATE, L.stream);
        for (int k = 0; k < kSweepChunk; k++)
            HIP_TRY(ctx, launch_hyst_propagate(L.S, L.C, L.sched, L.flags, L.iter + k, L.g, L.stream, L.edges,
                                               L.edge_value));
    }
    L.iter += kSweepChunk;
    L.seq = ++ctx->publish_seq;
    HIP_TRY(ctx, launch_hyst_publish(L.flags, L.host_dev, L.seq, L.stream));
    HIP_TRY(ctx, hipEventRecord(L.event, L.stream));
    return CANNY_HIP_OK;
}

// Waits for the chunk launched last and reads its verdict into L.converged.
int lane_wait(canny_hip_ctx *ctx, PropLane &L)
{
    volatile unsigned *hf = L.host;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&hf[2], __ATOMIC_ACQUIRE) != L.seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            HIP_TRY(ctx, hipEventSynchronize(L.event)); // errors surface here
            if (__atomic_load_n(&hf[2], __ATOMIC_ACQUIRE) != L.seq) return CANNY_HIP_ERR_RUNTIME;
        }
    }
    if (hf[1]) return CANNY_HIP_ERR_DOMAIN;
    L.converged = hf[0] != (unsigned)L.iter; // nothing was scheduled for sweep `iter`
    if (!L.converged && L.iter >= kMaxSweeps) return CANNY_HIP_ERR_NO_CONVERGE;
    if (L.converged) ctx->last_hyst_iters = std::max(ctx->last_hyst_iters, (int)hf[0] + 1);
    return CANNY_HIP_OK;
}

// The whole frame range of a call as one lane on the context's stream.
PropLane main_lane(canny_hip_ctx *ctx, const HystGeom &g, short *edges, int edge_value)
{
    PropLane L;
    L.stream = ctx->stream;
    L.S = (uint64_t *)ctx->plane_s.p;
    L.C = (const uint64_t *)ctx->plane_c.p;
    L.sched = (unsigned *)ctx->stamps.p;
    L.flags = (unsigned *)ctx->flags.p;
    L.host = ctx->host_flags;
    L.host_dev = ctx->host_flags_dev;
    L.event = ctx->flag_event;
    L.g = g;
    L.edges = edges;
    L.edge_value = edge_value;
    return L;
}

template <class Consumer>
int run_propagation(canny_hip_ctx *ctx, const HystGeom &g, bool speculative, Consumer &&consumer,
                    short *edges = nullptr, int edge_value = 0)
{
    PropLane L = main_lane(ctx, g, edges, edge_value);
    ctx->last_hyst_iters = 0;
    ctx->hyst_iters_async = false;
    int rc;
    do {
        if ((rc = lane_launch_chunk(ctx, L))) return rc;
        if (speculative && (rc = consumer())) return rc;
        if ((rc = lane_wait(ctx, L))) return rc;
    } while (!L.converged);
    return speculative ? CANNY_HIP_OK : consumer();
}

// Second half of hysteresis, from filled bit-planes to the s16 edge map (every pixel of d_out is written).
int propagate_and_finalize(canny_hip_ctx *ctx, const HystGeom &g, short *d_out, int hi)
{
    return run_propagation(ctx, g, /*speculative=*/true, [&]() -> int {
        StageTimer tm(ctx, CANNY_HIP_STAGE_HYST_FINALIZE);
        // reached pixels hold EDGE=255 and survive the final `< max_val -> 0` sweep only if 255 >= max_val
        HIP_TRY(ctx, launch_hyst_finalize(d_out, (const uint64_t *)ctx->plane_s.p, g, 255 >= hi ? 255 : 0,
                                          ctx->stream));
        return CANNY_HIP_OK;
    });
}

int finish_pending(canny_hip_ctx *ctx);

int dev_hysteresis(canny_hip_ctx *ctx, short *d_cand, int h, int w, int n, int lo, int hi)
{
    if (hysteresis_order_dependent(lo, hi)) return CANNY_HIP_ERR_DOMAIN;
    HystGeom g = make_hyst_geom(h, w, n);
    int rc = finish_pending(ctx); // a streamed call's sweeps own the planes until they are done
    if (rc || (rc = ensure_hyst(ctx, g))) return rc;
    if ((rc = prepare_hyst(ctx, g, /*zero_pad=*/false))) return rc; // the classify kernels write whole tiles
    {
        StageTimer tm(ctx, CANNY_HIP_STAGE_HYST_CLASSIFY);
        HIP_TRY(ctx, launch_hyst_classify(d_cand, (uint64_t *)ctx->plane_s.p, (uint64_t *)ctx->plane_c.p, g, lo, hi,
                                          (unsigned *)ctx->flags.p + 1, ctx->stream));
    }
    return propagate_and_finalize(ctx, g, d_cand, hi);
}

// Fused Sobel+NMS on a smoothed plane in [0,255].
int dev_sobel_nms(canny_hip_ctx *ctx, const short *d_smoothed, int h, int w, int n, short *d_out)
{
    if (ctx->sobel_nms_path != 1 && sobel_nms_march_supported(h, w)) {
        StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS, nullptr, /*attached=*/true);
        HIP_TRY(ctx, launch_sobel_nms_march(d_smoothed, d_out, h, w, n, ctx->stream, ctx->tune_sobel_seg,
                                            tm.launch_events()));
    } else {
        StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS);
        HIP_TRY(ctx, launch_sobel_nms(d_smoothed, d_out, h, w, n, /*domain8=*/true, ctx->stream));
    }
    return CANNY_HIP_OK;
}

int dev_canny(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int lo, int hi, int h, int w, int n,
              short *d_edges)
{
    if (h < 2 || w < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    if (hysteresis_order_dependent(lo, hi)) return CANNY_HIP_ERR_DOMAIN;
    int rc = finish_pending(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, ctx->smoothed.ensure(npx(h, w, n) * sizeof(short)));
    short *sm = (short *)ctx->smoothed.p;
    const bool fused = ctx->fuse_classify && ctx->sobel_nms_path != 1 && sobel_nms_classify_supported(h, w, lo);
    int sm_u8 = 0; // the smoothed plane as bytes: only between the two marching kernels of the fused path
    if (fused && ctx->smoothed_u8) {
        GaussTaps taps;
        if ((rc = make_taps(sigma, taps))) return rc;
        if (gaussian_u8_possible(ctx, taps, h, w)) sm_u8 = ctx->smoothed_u8;
    }
    ctx->last_canny_u8 = sm_u8;
    if ((rc = dev_gaussian(ctx, d_img, sigma, h, w, n, sm, sm_u8))) return rc;
    // Sobel+NMS+classify on the s16 or the u8 smoothed plane
    auto fused_sobel = [&](const short *smp, short *edges, uint64_t *S, uint64_t *C, const HystGeom &gg, int ev,
                           const LaunchEvents &le) -> hipError_t {
        return sm_u8 ? launch_sobel_nms_classify_march_u8in((const uint8_t *)smp, edges, S, C, gg, lo, hi, ev,
                                                            ctx->stream, ctx->tune_sobel_seg, le)
                     : launch_sobel_nms_classify_march(smp, edges, S, C, gg, lo, hi, ev, ctx->stream,
                                                       ctx->tune_sobel_seg, le);
    };
    if (fused) {
        // Sobel+NMS writes the hysteresis bit-planes directly: the suppressed magnitudes never reach memory
        // and the classify pass disappears (canny() does not return them; the stage API still does).
        HystGeom g = make_hyst_geom(h, w, n);
        if ((rc = ensure_hyst(ctx, g))) return rc;
        uint64_t *S = (uint64_t *)ctx->plane_s.p, *C = (uint64_t *)ctx->plane_c.p;
        // reached pixels hold EDGE=255 and survive the reference's final `< max_val -> 0` sweep only if 255 >= max_val
        const int edge_value = 255 >= hi ? 255 : 0;
        if (ctx->overlap_hysteresis && n >= 16) {
            // Two halves.  The sweeps of half A run on a second stream while the main stream does Sobel+NMS of
            // half B: the sweeps are bound by launch and tile-load latency and leave most of the chip idle, the
            // Sobel+NMS kernel fills it.  Only half B's propagation remains exposed at the end of the call.
            if ((rc = ensure_aux_stream(ctx))) return rc;
            const int nA = n / 2, nB = n - nA;
            const HystGeom gA = make_hyst_geom(h, w, nA), gB = make_hyst_geom(h, w, nB);
            const size_t px_a = npx(h, w, nA);
            HIP_TRY(ctx, launch_hyst_prepare(S, C, g, /*zero_pad=*/true, (unsigned *)ctx->stamps.p,
                                             (unsigned *)ctx->flags.p, ctx->stream, /*n_lanes=*/2));
            PropLane A = main_lane(ctx, gA, d_edges, edge_value), B = main_lane(ctx, gB, d_edges + px_a, edge_value);
            A.stream = ctx->aux_stream;
            A.event = ctx->aux_event;
            B.S += gA.words();
            B.C += gA.words();
            B.sched += hyst_sched_words(gA);
            B.flags += 2;
            B.host += 4;
            B.host_dev += 4;
            {
                StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS, nullptr, /*attached=*/true);
                HIP_TRY(ctx, fused_sobel(sm, d_edges, A.S, (uint64_t *)A.C, gA, edge_value, tm.launch_events()));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->fork_event, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->fork_event, 0));
            ctx->last_hyst_iters = 0;
            if ((rc = lane_launch_chunk(ctx, A))) return rc;
            {
                StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS, nullptr, /*attached=*/true);
                const short *smB = sm_u8 ? (const short *)((const uint8_t *)sm + px_a) : sm + px_a;
                HIP_TRY(ctx, fused_sobel(smB, d_edges + px_a, B.S, (uint64_t *)B.C, gB, edge_value, tm.launch_events()));
            }
            if ((rc = lane_launch_chunk(ctx, B))) return rc;
            for (PropLane *L : {&A, &B}) {
                if ((rc = lane_wait(ctx, *L))) return rc;
                while (!L->converged) {
                    if ((rc = lane_launch_chunk(ctx, *L)) || (rc = lane_wait(ctx, *L))) return rc;
                }
            }
            // whatever the caller queues on the context's stream next must see half A's edge map complete
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, A.event, 0));
            return CANNY_HIP_OK;
        }
        if ((rc = prepare_hyst(ctx, g, /*zero_pad=*/true))) return rc; // the kernel below writes in-image bytes only
        {
            StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS, nullptr, /*attached=*/true);
            HIP_TRY(ctx, fused_sobel(sm, d_edges, S, C, g, edge_value, tm.launch_events()));
        }
        // d_edges now holds the strong pixels; the sweeps add every pixel they promote: no finalize pass
        if (ctx->hyst_tail && g.tiles_x * g.tiles_y <= kTailMaxTiles) {
            // two batch-wide sweeps (all tiles; then the tiles those re-scheduled), then one workgroup per frame
            // runs the rest to convergence: everything is queued, nothing is waited for
            StageTimer tm(ctx, CANNY_HIP_STAGE_HYST_PROPAGATE);
            unsigned *sched = (unsigned *)ctx->stamps.p, *flags = (unsigned *)ctx->flags.p;
            const int wide = ctx->hyst_tail_after;
            for (int k = 0; k < wide; k++)
                HIP_TRY(ctx, launch_hyst_propagate(S, C, sched, flags, k, g, ctx->stream, d_edges, edge_value,
                                                   /*to_frame_queues=*/k == wide - 1));
            HIP_TRY(ctx, launch_hyst_tail(S, C, sched, flags, wide, g, ctx->stream, d_edges, edge_value));
            ctx->hyst_iters_async = true;
            return CANNY_HIP_OK;
        }
        ctx->hyst_iters_async = false;
        return run_propagation(ctx, g, /*speculative=*/false, []() -> int { return CANNY_HIP_OK; }, d_edges, edge_value);
    }
    if ((rc = dev_sobel_nms(ctx, sm, h, w, n, d_edges))) return rc;
    return dev_hysteresis(ctx, d_edges, h, w, n, lo, hi);
}

// Completes the propagation canny_hip_dev_canny_stream left in flight and joins it into the context's stream.
int finish_pending(canny_hip_ctx *ctx)
{
    if (!ctx->has_pend) return CANNY_HIP_OK;
    ctx->has_pend = false; // also on errors: the lane is not resumable
    PropLane &L = ctx->pend;
    int rc;
    if ((rc = lane_wait(ctx, L))) return rc;
    while (!L.converged)
        if ((rc = lane_launch_chunk(ctx, L)) || (rc = lane_wait(ctx, L))) return rc;
    if (L.stream != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, L.event, 0));
    return CANNY_HIP_OK;
}

// canny() for a stream of batches (the reference's capture loop, src/main.cpp:120-137, with batches for frames).
// A plain call ends with a host round trip: the host learns whether the sweeps converged before it returns, and
// the GPU idles until the next call's first kernel arrives (~20 us per call; a third of the time of a single 4K
// frame).  Here the call returns with its chunk of sweeps queued, and the NEXT call first queues its Gaussian
// (which touches none of the hysteresis state) and only then looks at the flag:
//   stream:  ... S(i-1) | P(i-1) sweeps, publish | G(i) ............ | prepare, S(i) | P(i) sweeps, publish |
//   host:                  call i: enqueue G(i); read P(i-1)'s flag (already there, or soon); enqueue the rest
// If the flag says "not converged" (rare: more than 8 sweeps), further chunks simply run behind G(i).
// stream_overlap = 1 puts the sweeps on a second, high-priority stream instead, so that G(i) runs BESIDE P(i-1):
// the sweeps are latency bound and G is VALU bound, but sweep 0's waves take slots from G's 5 waves/SIMD, and a
// 128-frame batch gains nothing (G 1.20 -> 1.37 ms for 0.26 ms of hidden sweeps).
// The bit-planes, scheduling words and the smoothed plane stay single-buffered: everything that writes them is
// ordered behind P(i-1), and the Gaussian writes only the smoothed plane, which S(i-1) has finished reading.
int dev_canny_stream(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int lo, int hi, int h, int w, int n,
                     short *d_edges)
{
    if (h < 2 || w < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    if (hysteresis_order_dependent(lo, hi)) return CANNY_HIP_ERR_DOMAIN;
    if (!(ctx->fuse_classify && ctx->sobel_nms_path != 1 && sobel_nms_classify_supported(h, w, lo)) ||
        (ctx->hyst_tail && make_hyst_geom(h, w, n).tiles_x * make_hyst_geom(h, w, n).tiles_y <= kTailMaxTiles)) {
        // shapes the fused kernel does not take: plain call, nothing left in flight.  With the per-frame tail kernel
        // a plain call does not wait for anything either, so there is nothing to defer.
        int rc = finish_pending(ctx);
        return rc ? rc : dev_canny(ctx, d_img, sigma, lo, hi, h, w, n, d_edges);
    }
    HIP_TRY(ctx, ctx->smoothed.ensure(npx(h, w, n) * sizeof(short)));
    short *sm = (short *)ctx->smoothed.p;
    int sm_u8 = 0;
    if (ctx->smoothed_u8) {
        GaussTaps taps;
        int rt = make_taps(sigma, taps);
        if (rt) return rt;
        if (gaussian_u8_possible(ctx, taps, h, w)) sm_u8 = ctx->smoothed_u8;
    }
    ctx->last_canny_u8 = sm_u8;
    int rc = dev_gaussian(ctx, d_img, sigma, h, w, n, sm, sm_u8);
    if (rc) return rc;
    if ((rc = finish_pending(ctx))) return rc; // host waits here while the Gaussian runs
    HystGeom g = make_hyst_geom(h, w, n);
    if ((rc = ensure_hyst(ctx, g)) || (ctx->stream_overlap && (rc = ensure_aux_stream(ctx)))) return rc;
    const int edge_value = 255 >= hi ? 255 : 0;
    if ((rc = prepare_hyst(ctx, g, /*zero_pad=*/true))) return rc;
    {
        StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS, nullptr, /*attached=*/true);
        if (sm_u8)
            HIP_TRY(ctx, launch_sobel_nms_classify_march_u8in((const uint8_t *)sm, d_edges, (uint64_t *)ctx->plane_s.p,
                                                              (uint64_t *)ctx->plane_c.p, g, lo, hi, edge_value,
                                                              ctx->stream, ctx->tune_sobel_seg, tm.launch_events()));
        else
            HIP_TRY(ctx, launch_sobel_nms_classify_march(sm, d_edges, (uint64_t *)ctx->plane_s.p,
                                                         (uint64_t *)ctx->plane_c.p, g, lo, hi, edge_value, ctx->stream,
                                                         ctx->tune_sobel_seg, tm.launch_events()));
    }
    ctx->pend = main_lane(ctx, g, d_edges, edge_value);
    if (ctx->stream_overlap) {
        HIP_TRY(ctx, hipEventRecord(ctx->fork_event, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->fork_event, 0));
        ctx->pend.stream = ctx->aux_stream;
        ctx->pend.event = ctx->aux_event;
    }
    ctx->last_hyst_iters = 0;
    ctx->hyst_iters_async = false; // this lane reports through its host flags (finish_pending), not through flags[0]
    if ((rc = lane_launch_chunk(ctx, ctx->pend))) return rc;
    ctx->has_pend = true;
    return CANNY_HIP_OK;
}

int h2d(canny_hip_ctx *ctx, DevBuf &b, const void *src, size_t bytes)
{
    HIP_TRY(ctx, b.ensure(bytes));
    HIP_TRY(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CANNY_HIP_OK;
}
int d2h_sync(canny_hip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANNY_HIP_OK;
}

void destroy_batch_pipe(canny_hip_ctx::BatchPipe *w)
{
    if (!w) return;
    if (w->sub) (void)hipSetDevice(w->sub->device);
    if (w->s_h2d) (void)hipStreamSynchronize(w->s_h2d);
    if (w->s_d2h) (void)hipStreamSynchronize(w->s_d2h);
    for (auto &sl : w->slot) {
        sl.d_in.release();
        sl.d_out.release();
        sl.d_out8.release();
        sl.pin_in.release();
        sl.pin_out.release();
        for (hipEvent_t e : {sl.ev_h2d, sl.ev_comp, sl.ev_d2h})
            if (e) (void)hipEventDestroy(e);
    }
    if (w->s_h2d) (void)hipStreamDestroy(w->s_h2d);
    if (w->s_d2h) (void)hipStreamDestroy(w->s_d2h);
    if (w->owns_sub) canny_hip_ctx_destroy(w->sub);
    delete w;
}

int create_batch_pipe(canny_hip_ctx *ctx, bool first, canny_hip_ctx::BatchPipe **out)
{
    auto *w = new (std::nothrow) canny_hip_ctx::BatchPipe();
    if (!w) return CANNY_HIP_ERR_RUNTIME;
    int st = CANNY_HIP_OK;
    if (first) {
        w->sub = ctx;
    } else {
        st = canny_hip_ctx_create(&w->sub, ctx->device);
        w->owns_sub = st == CANNY_HIP_OK;
    }
    hipError_t e = hipSuccess;
    if (!st) {
        for (auto &sl : w->slot)
            for (hipEvent_t *ev : {&sl.ev_h2d, &sl.ev_comp, &sl.ev_d2h})
                if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (e != hipSuccess) st = fail(ctx, e, "batch pipeline events");
    }
    if (st) {
        destroy_batch_pipe(w);
        return st;
    }
    *out = w;
    return CANNY_HIP_OK;
}

// the copy streams of three-stream mode
int ensure_copy_streams(canny_hip_ctx *ctx, canny_hip_ctx::BatchPipe &w)
{
    if (!w.s_h2d) HIP_TRY(ctx, hipStreamCreateWithFlags(&w.s_h2d, hipStreamNonBlocking));
    if (!w.s_d2h) HIP_TRY(ctx, hipStreamCreateWithFlags(&w.s_d2h, hipStreamNonBlocking));
    return CANNY_HIP_OK;
}

} // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int canny_hip_version(void) { return CANNY_HIP_VERSION; }

const char *canny_hip_status_string(int status)
{
    switch (status) {
    case CANNY_HIP_OK: return "ok";
    case CANNY_HIP_ERR_INVALID: return "invalid argument";
    case CANNY_HIP_ERR_UNSUPPORTED: return "unsupported size or window";
    case CANNY_HIP_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case CANNY_HIP_ERR_RUNTIME: return "HIP runtime error";
    case CANNY_HIP_ERR_DOMAIN: return "input outside the documented numeric domain";
    case CANNY_HIP_ERR_NO_CONVERGE: return "hysteresis propagation did not converge";
    default: return "unknown status";
    }
}

int canny_hip_device_count(int *count)
{
    if (!count) return CANNY_HIP_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return CANNY_HIP_ERR_NO_DEVICE;
    }
    *count = n;
    return CANNY_HIP_OK;
}

int canny_hip_ctx_create(canny_hip_ctx **out, int device)
{
    if (!out) return CANNY_HIP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return CANNY_HIP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return CANNY_HIP_ERR_INVALID;
    canny_hip_ctx *ctx = new (std::nothrow) canny_hip_ctx();
    if (!ctx) return CANNY_HIP_ERR_RUNTIME;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete ctx;
        return CANNY_HIP_ERR_NO_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return CANNY_HIP_OK;
}

void canny_hip_ctx_destroy(canny_hip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)finish_pending(ctx);
    ctx->expand_pool.reset(); // joins its threads; no job can be queued here (batch calls are synchronous)
    for (auto *w : ctx->batch_pool) destroy_batch_pipe(w);
    ctx->batch_pool.clear();
    (void)hipSetDevice(ctx->device);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->tmp_f32.release();
    ctx->smoothed.release();
    ctx->plane_s.release();
    ctx->plane_c.release();
    ctx->edges16.release();
    ctx->stamps.release();
    ctx->flags.release();
    for (auto &b : ctx->io) b.release();
    if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
    if (ctx->flag_event) (void)hipEventDestroy(ctx->flag_event);
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->aux_event) (void)hipEventDestroy(ctx->aux_event);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    for (auto &v : ctx->pending)
        for (auto &e : v) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
    for (auto &e : ctx->pool) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int canny_hip_ctx_set_stream(canny_hip_ctx *ctx, void *hip_stream)
{
    int rc = bind(ctx);
    if (rc || (rc = finish_pending(ctx))) return rc; // joins the old stream
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return CANNY_HIP_OK;
}

int canny_hip_ctx_device(const canny_hip_ctx *ctx) { return ctx ? ctx->device : -1; }

int canny_hip_ctx_get_option(const canny_hip_ctx *ctx, const char *name, int *value)
{
    if (!ctx || !name || !value) return CANNY_HIP_ERR_INVALID;
    if (!std::strcmp(name, "smoothed_u8")) *value = ctx->smoothed_u8;
    else if (!std::strcmp(name, "last_canny_smoothed_u8")) *value = ctx->last_canny_u8; // read-only
    else if (!std::strcmp(name, "fuse_classify")) *value = ctx->fuse_classify;
    else if (!std::strcmp(name, "hysteresis_tail")) *value = ctx->hyst_tail;
    else if (!std::strcmp(name, "gaussian_path")) *value = ctx->gaussian_path;
    else if (!std::strcmp(name, "sobel_nms_path")) *value = ctx->sobel_nms_path;
    else if (!std::strcmp(name, "tune_batch_compact")) *value = ctx->batch_compact;
    else if (!std::strcmp(name, "batch_expand_threads")) *value = ctx->expand_pool ? ctx->expand_pool->size() : 0; // read-only
    else return CANNY_HIP_ERR_INVALID;
    return CANNY_HIP_OK;
}

int canny_hip_ctx_set_option(canny_hip_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name || value < 0) return CANNY_HIP_ERR_INVALID;
    if (!std::strcmp(name, "gaussian_path") && value <= 2) ctx->gaussian_path = value;
    else if (!std::strcmp(name, "sobel_nms_path") && value <= 2) ctx->sobel_nms_path = value;
    else if (!std::strcmp(name, "tune_sobel_seg") && value <= 4096) ctx->tune_sobel_seg = value;
    else if (!std::strcmp(name, "fuse_classify") && value <= 1) ctx->fuse_classify = value;
    else if (!std::strcmp(name, "smoothed_u8") && value <= 1) ctx->smoothed_u8 = value;
    else if (!std::strcmp(name, "hysteresis_tail") && value <= 1) ctx->hyst_tail = value;
    else if (!std::strcmp(name, "tune_hyst_tail_after") && value >= 1 && value <= 4) ctx->hyst_tail_after = value;
    else if (!std::strcmp(name, "overlap_hysteresis") && value <= 1) ctx->overlap_hysteresis = value;
    else if (!std::strcmp(name, "tune_batch_workers") && value <= 16) ctx->batch_workers = value;
    else if (!std::strcmp(name, "tune_batch_chunk_mb") && value <= 1024) ctx->batch_chunk_mb = value;
    else if (!std::strcmp(name, "tune_batch_chunk_frames") && value <= 65535) ctx->batch_chunk_frames = value;
    else if (!std::strcmp(name, "tune_batch_pipe_mode") && value <= 2) ctx->batch_pipe_mode = value;
    else if (!std::strcmp(name, "tune_batch_compact") && value <= 1) ctx->batch_compact = value;
    else if (!std::strcmp(name, "tune_batch_expand_threads") && value <= 64) {
        if (value != ctx->batch_expand_threads) ctx->expand_pool.reset();
        ctx->batch_expand_threads = value;
    }
    else if (!std::strcmp(name, "stream_overlap") && value <= 1) {
        int rc = bind(ctx);
        if (rc || (rc = finish_pending(ctx))) return rc;
        ctx->stream_overlap = value;
    } else if (!std::strcmp(name, "profile_sample_interval") && value >= 1) {
        ctx->prof_every = (unsigned)value;
        for (auto &n : ctx->prof_seen) n = 0;
    } else if (!std::strcmp(name, "profile_stage_mask")) ctx->prof_mask = value ? (unsigned)value : ~0u;
    else if (!std::strcmp(name, "tune_sobel_px") && value <= 1) sobel_nms_set_px_variant(value); // process-wide
    else if (!std::strcmp(name, "tune_sobel_variant") && value <= 2) sobel_nms_set_arith_variant(value); // process-wide
    else if (!std::strcmp(name, "tune_plane_stores") && value <= 1) sobel_nms_set_plane_store_variant(value); // process-wide
    else if (!std::strcmp(name, "gaussian_fma_div") && value <= 1) gaussian_set_fma_div(value != 0); // process-wide
    else if (!std::strcmp(name, "tune_finalize_mode") && value <= 1) hyst_set_finalize_mode(value);   // process-wide
    else if (!std::strcmp(name, "tune_gaussian_variant") && value <= 4) gaussian_set_march_variant(value); // process-wide
    else if (!std::strcmp(name, "tune_gaussian_seg") && value <= 8192) gaussian_set_seg_target(value);      // process-wide
    else return CANNY_HIP_ERR_INVALID;
    return CANNY_HIP_OK;
}

int canny_hip_synchronize(canny_hip_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc || (rc = finish_pending(ctx))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANNY_HIP_OK;
}

const char *canny_hip_last_error(const canny_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
int canny_hip_last_hysteresis_iterations(canny_hip_ctx *ctx)
{
    if (!ctx) return 0;
    if (ctx->hyst_iters_async && ctx->flags.p) {
        // the tail path never reports to the host: fetch the last sweep that scheduled work (diagnostic, synchronises)
        unsigned last = 0;
        if (hipSetDevice(ctx->device) == hipSuccess &&
            hipMemcpyAsync(&last, ctx->flags.p, sizeof last, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
            hipStreamSynchronize(ctx->stream) == hipSuccess)
            ctx->last_hyst_iters = (int)last + 1;
        else
            (void)hipGetLastError();
        ctx->hyst_iters_async = false;
    }
    return ctx->last_hyst_iters;
}

// ---- memory helpers -------------------------------------------------------------------------------
int canny_hip_malloc(canny_hip_ctx *ctx, void **dev_ptr, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!dev_ptr) return CANNY_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipMalloc(dev_ptr, bytes ? bytes : 1));
    return CANNY_HIP_OK;
}
int canny_hip_free(canny_hip_ctx *ctx, void *dev_ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipFree(dev_ptr));
    return CANNY_HIP_OK;
}
// Pinned memory NEXT TO THE GPU: the pages of a hipHostMalloc come from the NUMA node of the CPU the calling thread
// happens to run on, and a buffer on the other socket costs the batch pipeline a third to a half of the PCIe rate
// (measured on a two-socket MI355X host: 11.5-19 instead of 25.4 Gpix/s, differing from run to run with the
// scheduler's choice).  So the allocation and the first touch of every page happen with the thread bound to the
// GPU's local CPUs (sysfs local_cpulist); the thread's own affinity is restored afterwards.
int canny_hip_host_alloc(canny_hip_ctx *ctx, void **host_ptr, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!host_ptr) return CANNY_HIP_ERR_INVALID;
    hipError_t e = numa_host_malloc(ctx->device, host_ptr, bytes);
    HIP_TRY(ctx, e);
    return CANNY_HIP_OK;
}
int canny_hip_host_free(canny_hip_ctx *ctx, void *host_ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipHostFree(host_ptr));
    return CANNY_HIP_OK;
}
// Page-locks memory the caller allocated itself (new[], malloc, a cv::Mat's data ...), so that the batch entry
// points DMA it in place like memory from canny_hip_host_alloc.  ~22 ms per GB the first time on an MI355X host.
int canny_hip_host_register(canny_hip_ctx *ctx, void *host_ptr, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!host_ptr || !bytes) return CANNY_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return CANNY_HIP_OK;
}
int canny_hip_host_unregister(canny_hip_ctx *ctx, void *host_ptr)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!host_ptr) return CANNY_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipHostUnregister(host_ptr));
    return CANNY_HIP_OK;
}
int canny_hip_memcpy_h2d(canny_hip_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANNY_HIP_OK;
}
int canny_hip_memcpy_d2h(canny_hip_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    return d2h_sync(ctx, host_dst, dev_src, bytes);
}

// ---- host-pointer stage functions -----------------------------------------------------------------
int canny_hip_gaussian_kernel(float sigma, float *taps, int cap, int *window)
{
    if (!taps || !window) return CANNY_HIP_ERR_INVALID;
    GaussTaps t;
    int rc = make_taps(sigma, t);
    if (rc) return rc;
    int w = 2 * t.center + 1;
    *window = w;
    if (cap < w) return CANNY_HIP_ERR_INVALID;
    std::memcpy(taps, t.tap, (size_t)w * sizeof(float));
    return CANNY_HIP_OK;
}

int canny_hip_gaussian(canny_hip_ctx *ctx, const unsigned char *img, float sigma, int height, int width, short *result)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!img || !result) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    size_t n = npx(height, width, 1);
    if ((rc = h2d(ctx, ctx->io[0], img, n))) return rc;
    HIP_TRY(ctx, ctx->io[1].ensure(n * 2));
    if ((rc = dev_gaussian(ctx, (const unsigned char *)ctx->io[0].p, sigma, height, width, 1, (short *)ctx->io[1].p)))
        return rc;
    return d2h_sync(ctx, result, ctx->io[1].p, n * 2);
}

int canny_hip_xy_gradient(canny_hip_ctx *ctx, const short *img, int height, int width, short *grad_x, short *grad_y)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!img || !grad_x || !grad_y) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    if (height < 2 || width < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    size_t n = npx(height, width, 1);
    if ((rc = h2d(ctx, ctx->io[0], img, n * 2))) return rc;
    HIP_TRY(ctx, ctx->io[1].ensure(n * 2));
    HIP_TRY(ctx, ctx->io[2].ensure(n * 2));
    {
        StageTimer tm(ctx, CANNY_HIP_STAGE_XY_GRADIENT);
        HIP_TRY(ctx, launch_xy_gradient((const int16_t *)ctx->io[0].p, (int16_t *)ctx->io[1].p, (int16_t *)ctx->io[2].p,
                                        height, width, 1, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(grad_x, ctx->io[1].p, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    return d2h_sync(ctx, grad_y, ctx->io[2].p, n * 2);
}

int canny_hip_sobel(canny_hip_ctx *ctx, const short *img, int height, int width, short *magnitude, short *angle)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!img || !magnitude || !angle) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    if (height < 2 || width < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    size_t n = npx(height, width, 1);
    if ((rc = h2d(ctx, ctx->io[0], img, n * 2))) return rc;
    HIP_TRY(ctx, ctx->io[1].ensure(n * 2));
    HIP_TRY(ctx, ctx->io[2].ensure(n * 2));
    {
        StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL);
        HIP_TRY(ctx, launch_sobel((const int16_t *)ctx->io[0].p, (int16_t *)ctx->io[1].p, (int16_t *)ctx->io[2].p,
                                  height, width, 1, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(magnitude, ctx->io[1].p, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    return d2h_sync(ctx, angle, ctx->io[2].p, n * 2);
}

int canny_hip_nms(canny_hip_ctx *ctx, const short *magnitude, const short *angle, int height, int width, short *result)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!magnitude || !angle || !result) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    size_t n = npx(height, width, 1);
    if ((rc = h2d(ctx, ctx->io[0], magnitude, n * 2))) return rc;
    if ((rc = h2d(ctx, ctx->io[1], angle, n * 2))) return rc;
    HIP_TRY(ctx, ctx->io[2].ensure(n * 2));
    {
        StageTimer tm(ctx, CANNY_HIP_STAGE_NMS);
        HIP_TRY(ctx, launch_nms((const int16_t *)ctx->io[0].p, (const int16_t *)ctx->io[1].p, (int16_t *)ctx->io[2].p,
                                height, width, 1, ctx->stream));
    }
    return d2h_sync(ctx, result, ctx->io[2].p, n * 2);
}

int canny_hip_hysteresis(canny_hip_ctx *ctx, short *edge_candidates, int height, int width, int min_val, int max_val)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!edge_candidates) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    size_t n = npx(height, width, 1);
    if ((rc = h2d(ctx, ctx->io[0], edge_candidates, n * 2))) return rc;
    if ((rc = dev_hysteresis(ctx, (short *)ctx->io[0].p, height, width, 1, min_val, max_val))) return rc;
    return d2h_sync(ctx, edge_candidates, ctx->io[0].p, n * 2);
}

int canny_hip_find_edge_pixels(canny_hip_ctx *ctx, short *edge_candidates, unsigned char *visited, int start,
                               int min_val, int max_val, int height, int width)
{
    (void)max_val; // unused by the reference too (src/utils.cpp:360-427)
    int rc = bind(ctx);
    if (rc) return rc;
    if (!edge_candidates || !visited) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    size_t n = npx(height, width, 1);
    if (start < 0 || (size_t)start >= n) return CANNY_HIP_ERR_INVALID;
    if (visited[start]) return CANNY_HIP_OK; // src/utils.cpp:361
    // min_val > EDGE: a popped pixel (now 255) fails the `>= minVal` test of its neighbours' checks, which changes
    // the reference's visited bookkeeping of the start pixel with the pop order
    if (min_val > 255) return CANNY_HIP_ERR_DOMAIN;
    HystGeom g = make_hyst_geom(height, width, 1);
    if ((rc = finish_pending(ctx))) return rc; // a streamed call's sweeps own the planes until they are done
    if ((rc = ensure_hyst(ctx, g))) return rc;
    if ((rc = h2d(ctx, ctx->io[0], edge_candidates, n * 2))) return rc;
    if ((rc = h2d(ctx, ctx->io[1], visited, n))) return rc;
    if ((rc = prepare_hyst(ctx, g, /*zero_pad=*/false))) return rc;
    HIP_TRY(ctx, launch_fep_classify((const int16_t *)ctx->io[0].p, (const uint8_t *)ctx->io[1].p,
                                     (uint64_t *)ctx->plane_s.p, (uint64_t *)ctx->plane_c.p, g, start, min_val,
                                     ctx->stream));
    // fep_finalize updates its inputs in place, so it runs once, after convergence (not speculatively)
    rc = run_propagation(ctx, g, /*speculative=*/false, [&]() -> int {
        HIP_TRY(ctx, launch_fep_finalize((int16_t *)ctx->io[0].p, (uint8_t *)ctx->io[1].p,
                                         (const uint64_t *)ctx->plane_s.p, g, start, min_val, ctx->stream));
        return CANNY_HIP_OK;
    });
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(edge_candidates, ctx->io[0].p, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    return d2h_sync(ctx, visited, ctx->io[1].p, n);
}

enum MapFormat { kMapS16 = 0, kMapU8 = 1, kMapBits = 2 };
static int canny_batch_impl(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                            int max_val, int height, int width, void *edges, MapFormat fmt);

int canny_hip_canny(canny_hip_ctx *ctx, const unsigned char *img, float sigma, int min_val, int max_val, int height,
                    int width, short *edges)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!img || !edges) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    size_t n = npx(height, width, 1);
    // Frames of a megapixel and more go through the batch pipeline as a batch of one: pinned staging for the upload and
    // the compact transfer for the map (1/16 of the bytes down, host threads write the caller's plane) -- the caller's
    // buffers here are ordinary new[] / cv::Mat memory, for which a plain 2-byte-per-pixel download is slowest of all.
    if (n >= (1u << 20) && height >= 2 && width >= 2 && ctx->batch_compact != 1)
        return canny_batch_impl(ctx, img, 1, sigma, min_val, max_val, height, width, edges, kMapS16);
    if ((rc = h2d(ctx, ctx->io[0], img, n))) return rc;
    HIP_TRY(ctx, ctx->io[1].ensure(n * 2));
    if ((rc = dev_canny(ctx, (const unsigned char *)ctx->io[0].p, sigma, min_val, max_val, height, width, 1,
                        (short *)ctx->io[1].p)))
        return rc;
    return d2h_sync(ctx, edges, ctx->io[1].p, n * 2);
}

int canny_hip_shard_range(int n_frames, int rank, int world, int *begin, int *end)
{
    if (!begin || !end || world < 1 || rank < 0 || rank >= world || n_frames < 0) return CANNY_HIP_ERR_INVALID;
    // contiguous ranges, the first (n_frames % world) shards one frame longer
    long long q = n_frames / world, r = n_frames % world;
    long long b = rank * q + std::min<long long>(rank, r);
    long long e = b + q + (rank < r ? 1 : 0);
    *begin = (int)b;
    *end = (int)e;
    return CANNY_HIP_OK;
}

// Stream-overlapped batch (BASELINE config 3).  The frames are cut into chunks; every pipeline (BatchPipe: one host
// thread, an upload, a compute and a download stream, three chunk slots) takes every n-th chunk and runs
//     upload(j+1) | kernels(j) | download(j-1)
// concurrently, chained by events only: the host thread never waits for a copy, it only blocks where
// canny() itself does (the hysteresis convergence poll at the end of a chunk's kernels), by which time the next
// chunk's upload has long been queued.  On an MI355X the kernels run at ~450 Gpix/s and a PCIe 5 x16 link moves
// ~56 GB/s per direction (48 + 48 GB/s with both directions busy): the link is the bound, and the pipeline's job is
// to keep both DMA engines busy all the time.
//   pinned caller buffers (canny_hip_host_alloc, hipHostMalloc, hipHostRegister): DMA'd in place, one pipeline;
//   pageable caller buffers: staged through the slot's pinned buffers by the pipeline's own thread (memcpy), so
//   several pipelines run side by side to get enough copy bandwidth.
// What a batch call returns per frame: the reference's short plane (0 / 255), the same as bytes, or one bit per pixel
// (rows MSB-first, padded to whole bytes -- launch_edges_to_bits).
static size_t map_frame_bytes(MapFormat fmt, int height, int width)
{
    if (fmt == kMapBits) return (size_t)height * (size_t)((width + 7) / 8);
    return npx(height, width, 1) * (fmt == kMapU8 ? 1 : sizeof(short));
}

static int canny_batch_impl(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                            int max_val, int height, int width, void *edges, MapFormat fmt)
{
    using Pipe = canny_hip_ctx::BatchPipe;
    int rc = bind(ctx);
    if (rc) return rc;
    if (!imgs || !edges) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, 1))) return rc;
    if (n_frames < 1) return CANNY_HIP_ERR_INVALID;
    if (height < 2 || width < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    GaussTaps probe;
    if ((rc = make_taps(sigma, probe))) return rc;
    const size_t frame_px = npx(height, width, 1);
    const int device = ctx->device;
    // Buffers from canny_hip_host_alloc (or any hipHostMalloc / hipHostRegister'd memory) need no staging.
    auto is_pinned = [](const void *p) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    };
    // Compact transfer (s16 and u8 maps): the device packs the map into bits, the bits travel, host threads expand them
    // into the caller's plane (ExpandPool).  The caller's output buffer is then written by CPU stores: it needs no
    // pinning and no staging copy.
    const bool compact = fmt != kMapBits && ctx->batch_compact != 1;
    if (ctx->batch_compact != 1 && !ctx->expand_pool) { // (also stages pageable input frames, see below)
        // 8 threads (profiles/r03/compact_transfer_threads_chunks.txt, 128 x 4K: 2 / 4 / 8 / 16 / 24 threads 52.7 / 53.0 /
        // 53.1 / 52.7 / 52.4 Gpix/s, 12 threads 49.8 both times it was measured, the plain download 25.4): even two
        // keep up with one GPU's upload -- streaming stores, and the pipeline thread works on blocks while it waits --,
        // so this is headroom for slower hosts.  Never more than the CPUs this thread may run on (the sharder and
        // bench.py bind themselves to the GPU's local CPUs first, and the pool's threads inherit that mask)
        int n = ctx->batch_expand_threads;
        if (n <= 0) {
            cpu_set_t set;
            const int avail = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 1;
            n = std::max(1, std::min(8, avail - 1));
        }
        ctx->expand_pool.reset(new (std::nothrow) ExpandPool(n));
        if (!ctx->expand_pool) return CANNY_HIP_ERR_RUNTIME;
    }
    const bool in_pinned = is_pinned(imgs), out_pinned = compact ? false : is_pinned(edges);
    // Pageable INPUT frames are staged into the chunk slot's pinned buffer by the same pool (1 MB pieces, the
    // pipeline thread works too): ~90 GB/s of memcpy against the link's 54, so ordinary caller memory runs through
    // the same single three-stream pipeline as pinned memory.  (Rounds 1-2: six single-stream pipelines whose
    // threads each copied their own chunks -- 20-35 Gpix/s and bimodal from call to call.)
    const bool pool_staging = !in_pinned && ctx->expand_pool != nullptr;
    const bool all_pinned = (in_pinned || pool_staging) && (out_pinned || compact);
    // Defaults from the sweep on an MI355X box (tools/probe_batch_sweep.py, 128 x 4K and 256 x 1080p):
    //   pinned buffers:   ONE three-stream pipeline, 24 MB chunks -- s16 maps 25.3 Gpix/s (D2H 50.6 GB/s: the link's
    //                     rate with both directions busy), u8 maps 45.4 Gpix/s; a second pipeline only adds streams
    //                     that fight for hardware queues (20.9 / 33.2);
    //   pageable buffers: six single-stream pipelines, 8 MB chunks (the staging memcpys need the threads): 21.9 /
    //                     32.4 Gpix/s.
    // "tune_batch_pipe_mode": 1 = three streams, 2 = one in-order stream per pipeline, 0 = automatic.
    const bool three_streams = ctx->batch_pipe_mode ? ctx->batch_pipe_mode == 1 : all_pinned;
    size_t chunk_frames = ctx->batch_chunk_frames
                              ? (size_t)ctx->batch_chunk_frames
                              : ((size_t)(ctx->batch_chunk_mb ? ctx->batch_chunk_mb : (all_pinned ? 24 : 8)) << 20) / frame_px;
    // never more than the 65535 frames a launch takes
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)n_frames, 65535), chunk_frames));
    const int n_chunks = (n_frames + chunk - 1) / chunk;
    const int n_workers = std::min(ctx->batch_workers ? ctx->batch_workers : (all_pinned ? 1 : 6), n_chunks);
    std::vector<int> status(n_workers, CANNY_HIP_OK);
    std::vector<std::string> errors(n_workers);
    while ((int)ctx->batch_pool.size() < n_workers) { // (sub-contexts are created here, on the caller's thread)
        Pipe *w = nullptr;
        if ((rc = create_batch_pipe(ctx, ctx->batch_pool.empty(), &w))) return rc;
        ctx->batch_pool.push_back(w);
    }
    if (three_streams)
        for (int i = 0; i < n_workers; i++)
            if ((rc = ensure_copy_streams(ctx, *ctx->batch_pool[i]))) return rc;
    const size_t out_frame = map_frame_bytes(fmt, height, width); // bytes of one frame's map as the caller gets it
    const size_t bits_frame = map_frame_bytes(kMapBits, height, width);
    const size_t wire_frame = compact ? bits_frame : out_frame;   // ... and as it crosses the link
    const int row_bytes = (width + 7) / 8;

    auto worker = [&](int wid) {
        Pipe &P = *ctx->batch_pool[wid];
        canny_hip_ctx *sub = P.sub;
        int &st = status[wid];
        // single-stream mode: uploads and downloads in order on the compute stream (overlap only between pipelines)
        const hipStream_t s_h2d = three_streams ? P.s_h2d : sub->stream;
        const hipStream_t s_d2h = three_streams ? P.s_d2h : sub->stream;
        if (hipSetDevice(device) != hipSuccess) { // a new thread starts on device 0
            st = CANNY_HIP_ERR_RUNTIME;
            return;
        }
        const int my_chunks = (n_chunks - wid + n_workers - 1) / n_workers;
        const size_t in_bytes = frame_px * chunk, out_bytes = wire_frame * chunk;
        hipError_t e = hipSuccess;
        constexpr int n_slots = Pipe::kSlots;
        for (int k = 0; k < std::min(my_chunks, n_slots) && e == hipSuccess; k++) {
            Pipe::Slot &S = P.slot[k];
            S.d2h_issued = false;
            S.retire_dst = nullptr;
            S.expand_left.store(0);
            // pageable caller buffers are staged through pinned memory; pinned ones are DMA'd in place
            if (!in_pinned) e = S.pin_in.ensure(in_bytes, device);
            if (e == hipSuccess && !out_pinned) e = S.pin_out.ensure(out_bytes, device);
            if (e == hipSuccess) e = S.d_in.ensure(in_bytes);
            if (e == hipSuccess) e = S.d_out.ensure(frame_px * chunk * sizeof(short));
            if (e == hipSuccess && (fmt != kMapS16 || compact)) e = S.d_out8.ensure(out_bytes);
        }
        if (e != hipSuccess) {
            st = fail(sub, e, "batch staging allocation");
            errors[wid] = sub->last_error;
            return;
        }
        auto chunk_range = [&](int j, int &f0, int &nf) {
            const int c = wid + j * n_workers;
            f0 = c * chunk;
            nf = std::min(chunk, n_frames - f0);
        };
        // host -> d_in of chunk j
        auto upload = [&](int j) -> hipError_t {
            Pipe::Slot &S = P.slot[j % n_slots];
            int f0, nf;
            chunk_range(j, f0, nf);
            const unsigned char *src = imgs + (size_t)f0 * frame_px;
            hipError_t err = hipSuccess;
            if (j >= n_slots) {
                // the slot's previous user (chunk j - kSlots): its kernels must have read d_in (dev_canny no longer
                // blocks the host), and its upload must have left pin_in
                err = hipStreamWaitEvent(s_h2d, S.ev_comp, 0);
                if (err == hipSuccess && !in_pinned) err = hipEventSynchronize(S.ev_h2d);
                if (err != hipSuccess) return err;
            }
            if (!in_pinned) {
                if (pool_staging)
                    ctx->expand_pool->parallel_copy(S.pin_in.p, src, frame_px * nf);
                else
                    std::memcpy(S.pin_in.p, src, frame_px * nf);
                src = (const unsigned char *)S.pin_in.p;
            }
            err = hipMemcpyAsync(S.d_in.p, src, frame_px * nf, hipMemcpyHostToDevice, s_h2d);
            if (err == hipSuccess) err = hipEventRecord(S.ev_h2d, s_h2d);
            return err;
        };
        // the staged output of chunk j reaches the caller's pageable buffer
        auto retire = [&](int j) -> hipError_t {
            Pipe::Slot &S = P.slot[j % n_slots];
            if (!S.retire_dst) return hipSuccess;
            hipError_t err = hipEventSynchronize(S.ev_d2h);
            if (err == hipSuccess && compact) {
                // the chunk's bit maps have landed in pin_out: hand them to the expansion pool in blocks of rows and
                // go on (the pool's threads write the caller's plane while the next chunks move)
                constexpr int kBlockRows = 128;
                const int blocks_per_frame = (height + kBlockRows - 1) / kBlockRows;
                S.expand_left.store(S.retire_frames * blocks_per_frame, std::memory_order_release);
                const size_t px_bytes = fmt == kMapU8 ? 1 : sizeof(short);
                for (int f = 0; f < S.retire_frames; f++)
                    for (int b = 0; b < blocks_per_frame; b++) {
                        const int r0 = b * kBlockRows;
                        ExpandPool::Job job;
                        job.bits = (const uint8_t *)S.pin_out.p + (size_t)f * bits_frame + (size_t)r0 * row_bytes;
                        job.dst = (unsigned char *)S.retire_dst + ((size_t)f * frame_px + (size_t)r0 * width) * px_bytes;
                        job.rows = std::min(kBlockRows, height - r0);
                        job.width = width;
                        job.row_bytes = row_bytes;
                        job.to_u8 = fmt == kMapU8;
                        job.left = &S.expand_left;
                        ctx->expand_pool->submit(job);
                    }
            } else if (err == hipSuccess) {
                std::memcpy(S.retire_dst, S.pin_out.p, S.retire_bytes);
            }
            S.retire_dst = nullptr;
            return err;
        };
        const char *where = "batch H2D";
        e = upload(0);
        for (int j = 0; j < my_chunks && e == hipSuccess && st == CANNY_HIP_OK; j++) {
            Pipe::Slot &S = P.slot[j % n_slots];
            int f0, nf;
            chunk_range(j, f0, nf);
            if (j + 1 < my_chunks && (e = upload(j + 1)) != hipSuccess) break;
            // kernels of chunk j: behind its upload, and behind the download that last read this slot's outputs
            where = "batch compute";
            if ((e = hipStreamWaitEvent(sub->stream, S.ev_h2d, 0)) != hipSuccess) break;
            if (S.d2h_issued && (e = hipStreamWaitEvent(sub->stream, S.ev_d2h, 0)) != hipSuccess) break;
            st = dev_canny(sub, (const unsigned char *)S.d_in.p, sigma, min_val, max_val, height, width, nf,
                           (short *)S.d_out.p);
            if (st) break;
            const void *d_res = S.d_out.p;
            if (fmt != kMapS16 || compact) { // narrow on the device: the D2H copy is what these variants are for
                e = (fmt == kMapU8 && !compact)
                        ? launch_edges_to_u8((const int16_t *)S.d_out.p, (uint8_t *)S.d_out8.p, frame_px * nf, sub->stream)
                        : launch_edges_to_bits((const int16_t *)S.d_out.p, (uint8_t *)S.d_out8.p, height, width, nf,
                                               sub->stream);
                if (e != hipSuccess) break;
                d_res = S.d_out8.p;
            }
            if ((e = hipEventRecord(S.ev_comp, sub->stream)) != hipSuccess) break;
            // The HOST waits for chunk j's kernels here (its next upload is already queued).  canny() itself no longer
            // blocks, and a host that runs ahead by the whole batch -- a thousand queued copies, kernels and event
            // waits -- makes the runtime slower per chunk and erratic: 1024 x 1080p in 24 MB chunks 18.7 Gpix/s
            // against 25.5 with a third as many 64 MB chunks, 128 x 4K anywhere between 19 and 25.4 from run to run.
            // One chunk of run-ahead is what the blocking canny() of the first version of this pipeline gave it,
            // and what measured best and steadiest (profiles/r02/c3_sweep_*.txt, pipe_patterns.txt "P2b"); letting
            // the host run five chunks ahead kept the s16 rate but made the u8 rate erratic (27-44 ms per 128 x 4K).
            if ((e = hipEventSynchronize(S.ev_comp)) != hipSuccess) break;
            // download of chunk j (pin_out of this slot was retired kSlots - 1 iterations ago)
            where = "batch D2H";
            unsigned char *dst = (unsigned char *)edges + (size_t)f0 * out_frame;
            const size_t bytes = wire_frame * nf;
            // compact transfer: the expansion of the chunk that used this slot's pin_out before must have read it
            if (compact) ctx->expand_pool->wait(S.expand_left);
            if ((e = hipStreamWaitEvent(s_d2h, S.ev_comp, 0)) != hipSuccess) break;
            if ((e = hipMemcpyAsync(out_pinned ? (void *)dst : S.pin_out.p, d_res, bytes, hipMemcpyDeviceToHost,
                                    s_d2h)) != hipSuccess)
                break;
            if ((e = hipEventRecord(S.ev_d2h, s_d2h)) != hipSuccess) break;
            S.d2h_issued = true;
            if (!out_pinned) {
                S.retire_dst = dst;
                S.retire_bytes = bytes;
                S.retire_frames = nf;
            }
            if (j > 0 && (e = retire(j - 1)) != hipSuccess) break;
        }
        if (e == hipSuccess && st == CANNY_HIP_OK && my_chunks > 0) e = retire(my_chunks - 1);
        if (compact) // every plane block of this pipeline has been written when the call returns (also after errors)
            for (auto &sl : P.slot) ctx->expand_pool->wait(sl.expand_left);
        // nothing of this call may still be in flight when it returns (also after an error: the slots are reused)
        hipError_t e2 = hipStreamSynchronize(s_h2d);
        hipError_t e3 = hipStreamSynchronize(sub->stream);
        hipError_t e4 = hipStreamSynchronize(s_d2h);
        if (e == hipSuccess) e = e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : e4);
        if (e != hipSuccess && st == CANNY_HIP_OK) st = fail(sub, e, where);
        if (st != CANNY_HIP_OK) errors[wid] = sub->last_error;
    };
    std::vector<std::thread> threads;
    for (int i = 1; i < n_workers; i++) threads.emplace_back(worker, i);
    worker(0);
    for (auto &t : threads) t.join();
    for (int i = 0; i < n_workers; i++)
        if (status[i]) {
            ctx->last_error = errors[i];
            return status[i];
        }
    return CANNY_HIP_OK;
}

int canny_hip_canny_batch(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                          int max_val, int height, int width, short *edges)
{
    return canny_batch_impl(ctx, imgs, n_frames, sigma, min_val, max_val, height, width, edges, kMapS16);
}

int canny_hip_canny_batch_u8(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                             int max_val, int height, int width, unsigned char *edges)
{
    return canny_batch_impl(ctx, imgs, n_frames, sigma, min_val, max_val, height, width, edges, kMapU8);
}

int canny_hip_canny_batch_bits(canny_hip_ctx *ctx, const unsigned char *imgs, int n_frames, float sigma, int min_val,
                               int max_val, int height, int width, unsigned char *bits)
{
    return canny_batch_impl(ctx, imgs, n_frames, sigma, min_val, max_val, height, width, bits, kMapBits);
}

// ---- multi-GPU sharder (BASELINE config 5) -----------------------------------------------------------
// One context per shard, kept between calls (a context with its pipelines, streams and staging costs ~10-20 ms to
// build); one host thread per shard per call, bound to the CPUs that are local to the shard's GPU.
namespace {
struct MultiGpuState {
    std::mutex mu;
    std::vector<canny_hip_ctx *> shard_ctx; // index = shard
    std::vector<int> shard_dev;
    int batch_workers = 0, batch_chunk_mb = 0, batch_chunk_frames = 0, batch_pipe_mode = 0, batch_compact = 0;
    int allow_device_reuse = 0; // shards beyond the device count wrap around (testing the sharder on a small box)
    int numa_affinity = 1;
};
MultiGpuState g_mgpu;

} // namespace

static int multi_gpu_impl(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val, int height,
                          int width, void *edges, int n_devices, MapFormat fmt)
{
    if (!imgs || !edges || n_frames < 1 || height < 1 || width < 1) return CANNY_HIP_ERR_INVALID;
    int avail = 0;
    int rc = canny_hip_device_count(&avail);
    if (rc) return rc;
    if (avail < 1) return CANNY_HIP_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_mgpu.mu); // one sharded call at a time per process
    if (n_devices <= 0) n_devices = avail;
    if (n_devices > avail && !g_mgpu.allow_device_reuse) n_devices = avail;
    if (n_devices > 64) return CANNY_HIP_ERR_INVALID;
    const int n_shards = n_devices;
    const size_t frame_px = npx(height, width, 1);
    const size_t out_frame = map_frame_bytes(fmt, height, width);
    // contexts are created here, on the caller's thread, and kept for the next call
    while ((int)g_mgpu.shard_ctx.size() < n_shards) {
        const int shard = (int)g_mgpu.shard_ctx.size();
        canny_hip_ctx *c = nullptr;
        if ((rc = canny_hip_ctx_create(&c, shard % avail))) return rc;
        g_mgpu.shard_ctx.push_back(c);
        g_mgpu.shard_dev.push_back(shard % avail);
    }
    std::vector<int> status(n_shards, CANNY_HIP_OK);
    auto worker = [&](int shard, bool own_thread) {
        int b = 0, e = 0;
        canny_hip_shard_range(n_frames, shard, n_shards, &b, &e);
        if (e <= b) return; // more shards than frames
        canny_hip_ctx *ctx = g_mgpu.shard_ctx[shard];
        ctx->batch_workers = g_mgpu.batch_workers;
        ctx->batch_chunk_mb = g_mgpu.batch_chunk_mb;
        ctx->batch_chunk_frames = g_mgpu.batch_chunk_frames;
        ctx->batch_pipe_mode = g_mgpu.batch_pipe_mode;
        ctx->batch_compact = g_mgpu.batch_compact;
        // run next to the GPU: this thread and the pipeline threads it starts inherit the mask.  The caller's own
        // thread (shard 0) gets its mask back afterwards.
        cpu_set_t local, saved;
        bool restore = false;
        if (g_mgpu.numa_affinity && device_local_cpus(g_mgpu.shard_dev[shard], &local)) {
            if (!own_thread) restore = sched_getaffinity(0, sizeof saved, &saved) == 0;
            if (own_thread || restore) (void)sched_setaffinity(0, sizeof local, &local); // best effort
        }
        status[shard] = canny_batch_impl(ctx, imgs + (size_t)b * frame_px, e - b, sigma, min_val, max_val, height,
                                         width, (unsigned char *)edges + (size_t)b * out_frame, fmt);
        if (restore) (void)sched_setaffinity(0, sizeof saved, &saved);
    };
    std::vector<std::thread> threads;
    for (int d = 1; d < n_shards; d++) threads.emplace_back(worker, d, true);
    worker(0, false);
    for (auto &t : threads) t.join();
    for (int s : status)
        if (s) return s;
    return CANNY_HIP_OK;
}

int canny_hip_canny_multi_gpu(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val,
                              int height, int width, short *edges, int n_devices)
{
    return multi_gpu_impl(imgs, n_frames, sigma, min_val, max_val, height, width, edges, n_devices, kMapS16);
}

int canny_hip_canny_multi_gpu_u8(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val,
                                 int height, int width, unsigned char *edges, int n_devices)
{
    return multi_gpu_impl(imgs, n_frames, sigma, min_val, max_val, height, width, edges, n_devices, kMapU8);
}

int canny_hip_canny_multi_gpu_bits(const unsigned char *imgs, int n_frames, float sigma, int min_val, int max_val,
                                   int height, int width, unsigned char *bits, int n_devices)
{
    return multi_gpu_impl(imgs, n_frames, sigma, min_val, max_val, height, width, bits, n_devices, kMapBits);
}

int canny_hip_multi_gpu_set_option(const char *name, int value)
{
    if (!name || value < 0) return CANNY_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lock(g_mgpu.mu);
    if (!std::strcmp(name, "tune_batch_workers") && value <= 16) g_mgpu.batch_workers = value;
    else if (!std::strcmp(name, "tune_batch_chunk_mb") && value <= 1024) g_mgpu.batch_chunk_mb = value;
    else if (!std::strcmp(name, "tune_batch_chunk_frames") && value <= 65535) g_mgpu.batch_chunk_frames = value;
    else if (!std::strcmp(name, "tune_batch_pipe_mode") && value <= 2) g_mgpu.batch_pipe_mode = value;
    else if (!std::strcmp(name, "tune_batch_compact") && value <= 1) g_mgpu.batch_compact = value;
    else if (!std::strcmp(name, "allow_device_reuse") && value <= 1) g_mgpu.allow_device_reuse = value;
    else if (!std::strcmp(name, "numa_affinity") && value <= 1) g_mgpu.numa_affinity = value;
    else return CANNY_HIP_ERR_INVALID;
    return CANNY_HIP_OK;
}

int canny_hip_multi_gpu_release(void)
{
    std::lock_guard<std::mutex> lock(g_mgpu.mu);
    for (canny_hip_ctx *c : g_mgpu.shard_ctx) canny_hip_ctx_destroy(c);
    g_mgpu.shard_ctx.clear();
    g_mgpu.shard_dev.clear();
    return CANNY_HIP_OK;
}

// Number of CPUs in a sysfs-style list ("0-3,8,10-11" -> 7), 0 if malformed: the parser behind the NUMA binding.
int canny_hip_selftest_cpulist_count(const char *text)
{
    if (!text) return 0;
    cpu_set_t set;
    return parse_cpulist(text, &set);
}

// CPUs local to `device` as sysfs lists them ("0-31,128-159"); what canny_hip_canny_multi_gpu binds a shard's
// threads to.  CANNY_HIP_ERR_UNSUPPORTED when the platform does not say.
int canny_hip_device_local_cpus(int device, char *buf, int cap)
{
    if (!buf || cap < 2) return CANNY_HIP_ERR_INVALID;
    int n = 0;
    int rc = canny_hip_device_count(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) return CANNY_HIP_ERR_INVALID;
    cpu_set_t set;
    if (!device_local_cpus(device, &set)) return CANNY_HIP_ERR_UNSUPPORTED;
    std::string out;
    for (int c = 0; c < CPU_SETSIZE;) {
        if (!CPU_ISSET(c, &set)) {
            c++;
            continue;
        }
        int e = c;
        while (e + 1 < CPU_SETSIZE && CPU_ISSET(e + 1, &set)) e++;
        if (!out.empty()) out += ",";
        out += std::to_string(c);
        if (e > c) out += "-" + std::to_string(e);
        c = e + 1;
    }
    if ((int)out.size() + 1 > cap) return CANNY_HIP_ERR_INVALID;
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return CANNY_HIP_OK;
}

// ---- device-pointer stage functions ---------------------------------------------------------------
int canny_hip_dev_gaussian(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int height, int width,
                           int n_frames, short *d_result)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_result) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    return dev_gaussian(ctx, d_img, sigma, height, width, n_frames, d_result);
}

int canny_hip_dev_xy_gradient(canny_hip_ctx *ctx, const short *d_img, int height, int width, int n_frames,
                              short *d_grad_x, short *d_grad_y)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_grad_x || !d_grad_y) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    if (height < 2 || width < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    StageTimer tm(ctx, CANNY_HIP_STAGE_XY_GRADIENT);
    HIP_TRY(ctx, launch_xy_gradient(d_img, d_grad_x, d_grad_y, height, width, n_frames, ctx->stream));
    return CANNY_HIP_OK;
}

int canny_hip_dev_sobel(canny_hip_ctx *ctx, const short *d_img, int height, int width, int n_frames, short *d_magnitude,
                        short *d_angle)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_magnitude || !d_angle) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    if (height < 2 || width < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL);
    HIP_TRY(ctx, launch_sobel(d_img, d_magnitude, d_angle, height, width, n_frames, ctx->stream));
    return CANNY_HIP_OK;
}

int canny_hip_dev_nms(canny_hip_ctx *ctx, const short *d_magnitude, const short *d_angle, int height, int width,
                      int n_frames, short *d_result)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_magnitude || !d_angle || !d_result) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    StageTimer tm(ctx, CANNY_HIP_STAGE_NMS);
    HIP_TRY(ctx, launch_nms(d_magnitude, d_angle, d_result, height, width, n_frames, ctx->stream));
    return CANNY_HIP_OK;
}

int canny_hip_dev_sobel_nms(canny_hip_ctx *ctx, const short *d_smoothed, int height, int width, int n_frames,
                            short *d_nms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_smoothed || !d_nms) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    if (height < 2 || width < 2) return CANNY_HIP_ERR_UNSUPPORTED;
    return dev_sobel_nms(ctx, d_smoothed, height, width, n_frames, d_nms);
}

int canny_hip_dev_gaussian_u8(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int height, int width,
                              int n_frames, unsigned char *d_result)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_result) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    return dev_gaussian(ctx, d_img, sigma, height, width, n_frames, d_result, 1);
}

int canny_hip_dev_sobel_nms_u8in(canny_hip_ctx *ctx, const unsigned char *d_smoothed, int height, int width,
                                 int n_frames, short *d_nms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_smoothed || !d_nms) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    if (height < 2 || width < 2 || !sobel_nms_u8_input_supported()) return CANNY_HIP_ERR_UNSUPPORTED;
    StageTimer tm(ctx, CANNY_HIP_STAGE_SOBEL_NMS, nullptr, /*attached=*/true);
    HIP_TRY(ctx, launch_sobel_nms_march_u8in(d_smoothed, d_nms, height, width, n_frames, ctx->stream,
                                             ctx->tune_sobel_seg, tm.launch_events()));
    return CANNY_HIP_OK;
}

int canny_hip_dev_hysteresis(canny_hip_ctx *ctx, short *d_edge_candidates, int height, int width, int n_frames,
                             int min_val, int max_val)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_edge_candidates) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    return dev_hysteresis(ctx, d_edge_candidates, height, width, n_frames, min_val, max_val);
}

int canny_hip_dev_canny(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                        int height, int width, int n_frames, short *d_edges)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_edges) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    return dev_canny(ctx, d_img, sigma, min_val, max_val, height, width, n_frames, d_edges);
}

int canny_hip_dev_canny_stream(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                               int height, int width, int n_frames, short *d_edges)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_edges) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    return dev_canny_stream(ctx, d_img, sigma, min_val, max_val, height, width, n_frames, d_edges);
}

int canny_hip_dev_canny_stream_flush(canny_hip_ctx *ctx)
{
    int rc = bind(ctx);
    return rc ? rc : finish_pending(ctx);
}

int canny_hip_dev_canny_u8(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                           int height, int width, int n_frames, unsigned char *d_edges)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_edges) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    const size_t n = npx(height, width, n_frames);
    HIP_TRY(ctx, ctx->edges16.ensure(n * sizeof(short)));
    if ((rc = dev_canny(ctx, d_img, sigma, min_val, max_val, height, width, n_frames, (short *)ctx->edges16.p)))
        return rc;
    HIP_TRY(ctx, launch_edges_to_u8((const int16_t *)ctx->edges16.p, d_edges, n, ctx->stream));
    return CANNY_HIP_OK;
}

int canny_hip_dev_canny_bits(canny_hip_ctx *ctx, const unsigned char *d_img, float sigma, int min_val, int max_val,
                             int height, int width, int n_frames, unsigned char *d_bits)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_img || !d_bits) return CANNY_HIP_ERR_INVALID;
    if ((rc = check_dims(height, width, n_frames))) return rc;
    const size_t n = npx(height, width, n_frames);
    HIP_TRY(ctx, ctx->edges16.ensure(n * sizeof(short)));
    if ((rc = dev_canny(ctx, d_img, sigma, min_val, max_val, height, width, n_frames, (short *)ctx->edges16.p)))
        return rc;
    HIP_TRY(ctx, launch_edges_to_bits((const int16_t *)ctx->edges16.p, d_bits, height, width, n_frames, ctx->stream));
    return CANNY_HIP_OK;
}

// ---- profiling --------------------------------------------------------------------------------------
int canny_hip_profile_enable(canny_hip_ctx *ctx, int on)
{
    if (!ctx) return CANNY_HIP_ERR_INVALID;
    ctx->prof = on != 0;
    return CANNY_HIP_OK;
}

static int profile_collect(canny_hip_ctx *ctx)
{
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < CANNY_HIP_STAGE_COUNT; s++) {
        for (auto &e : ctx->pending[s]) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
                ctx->total_ms[s] += ms;
                ctx->launches[s] += 1;
            } else {
                (void)hipGetLastError();
            }
            ctx->pool.push_back(e);
        }
        ctx->pending[s].clear();
    }
    return CANNY_HIP_OK;
}

int canny_hip_profile_reset(canny_hip_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = profile_collect(ctx))) return rc;
    for (int s = 0; s < CANNY_HIP_STAGE_COUNT; s++) {
        ctx->total_ms[s] = 0.0;
        ctx->launches[s] = 0;
    }
    return CANNY_HIP_OK;
}

int canny_hip_profile_get(canny_hip_ctx *ctx, int stage, double *total_ms, long *launches)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (stage < 0 || stage >= CANNY_HIP_STAGE_COUNT || !total_ms || !launches) return CANNY_HIP_ERR_INVALID;
    if ((rc = profile_collect(ctx))) return rc;
    *total_ms = ctx->total_ms[stage];
    *launches = ctx->launches[stage];
    return CANNY_HIP_OK;
}

// ---- measurement aid ------------------------------------------------------------------------------------
int canny_hip_probe_copy(canny_hip_ctx *ctx, const void *d_src, void *d_dst, size_t nbytes, int launches,
                         double *avg_ms)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_src || !d_dst || !avg_ms || launches < 1 || launches > 10000 || nbytes < 16 || (nbytes & 15) ||
        ((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 15))
        return CANNY_HIP_ERR_INVALID;
    hipEvent_t a = nullptr, b = nullptr;
    HIP_TRY(ctx, hipEventCreate(&a));
    {
        const hipError_t eb = hipEventCreate(&b);
        if (eb != hipSuccess) {
            (void)hipEventDestroy(a);
            return fail(ctx, eb, "hipEventCreate");
        }
    }
    double total = 0.0;
    hipError_t e = hipSuccess;
    for (int k = 0; k < launches && e == hipSuccess; k++) {
        // the event pair is attached to the dispatch: the kernel's own begin / end timestamps
        e = launch_probe_copy(d_src, d_dst, nbytes, ctx->stream, LaunchEvents{a, b});
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        total += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIP_TRY(ctx, e);
    *avg_ms = total / launches;
    return CANNY_HIP_OK;
}

// Host-only: the cell order of the marching launches (march_cell_of), for the bijectivity test.
int canny_hip_selftest_march_order(int n_segs, int n_strips, int *out_pairs)
{
    if (!out_pairs || n_segs < 1 || n_strips < 1 || (long long)n_segs * n_strips > (1 << 24)) return CANNY_HIP_ERR_INVALID;
    for (int k = 0; k < n_segs * n_strips; k++) {
        const MarchCell c = march_cell_of(k, n_segs, n_strips);
        out_pairs[2 * k] = c.seg;
        out_pairs[2 * k + 1] = c.strip;
    }
    return CANNY_HIP_OK;
}

// Host-only: the batch pipelines' bit-map expansion (ExpandPool) on a caller-supplied bit map; needs no device.
int canny_hip_selftest_expand_bits(const unsigned char *bits, int height, int width, int to_u8, void *out, int n_threads)
{
    if (!bits || !out || height < 1 || width < 1 || n_threads < 1 || n_threads > 64) return CANNY_HIP_ERR_INVALID;
    ExpandPool pool(n_threads);
    constexpr int kBlockRows = 128;
    const int row_bytes = (width + 7) / 8, blocks = (height + kBlockRows - 1) / kBlockRows;
    std::atomic<int> left{blocks};
    for (int b = 0; b < blocks; b++) {
        ExpandPool::Job job;
        const int r0 = b * kBlockRows;
        job.bits = bits + (size_t)r0 * row_bytes;
        job.dst = (unsigned char *)out + (size_t)r0 * width * (to_u8 ? 1 : sizeof(short));
        job.rows = std::min(kBlockRows, height - r0);
        job.width = width;
        job.row_bytes = row_bytes;
        job.to_u8 = to_u8 != 0;
        job.left = &left;
        pool.submit(job);
    }
    pool.wait(left);
    return CANNY_HIP_OK;
}

// ---- self-test ----------------------------------------------------------------------------------------
int canny_hip_selftest_mag_angle(canny_hip_ctx *ctx, int lim, short *magnitudes, unsigned char *bins)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!magnitudes || !bins || lim < 0 || lim > 1020) return CANNY_HIP_ERR_INVALID;
    size_t total = (size_t)(2 * lim + 1) * (2 * lim + 1);
    HIP_TRY(ctx, ctx->io[0].ensure(total * 2));
    HIP_TRY(ctx, ctx->io[1].ensure(total));
    HIP_TRY(ctx, launch_selftest_mag_angle(lim, (int16_t *)ctx->io[0].p, (uint8_t *)ctx->io[1].p, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(magnitudes, ctx->io[0].p, total * 2, hipMemcpyDeviceToHost, ctx->stream));
    return d2h_sync(ctx, bins, ctx->io[1].p, total);
}

static int selftest_div_common(canny_hip_ctx *ctx, float divisor, int use_fma, float c,
                               unsigned long long *mismatches, float *largest_mismatching_dividend)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!mismatches || !largest_mismatching_dividend || !(divisor > 0.0f) || !std::isfinite(divisor))
        return CANNY_HIP_ERR_INVALID;
    HIP_TRY(ctx, ctx->io[0].ensure(2 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->io[0].p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    const unsigned last = 0x43800000u; // bit pattern of 256.0f; non-negative floats are ordered like their bits
    HIP_TRY(ctx, launch_selftest_div(divisor, use_fma, c, 0u, last, (unsigned long long *)ctx->io[0].p, ctx->stream));
    unsigned long long res[2] = {0, 0};
    if ((rc = d2h_sync(ctx, res, ctx->io[0].p, sizeof(res)))) return rc;
    *mismatches = res[0];
    unsigned bits = (unsigned)res[1];
    std::memcpy(largest_mismatching_dividend, &bits, sizeof(float));
    return CANNY_HIP_OK;
}

int canny_hip_selftest_div(canny_hip_ctx *ctx, float divisor, unsigned long long *mismatches,
                           float *largest_mismatching_dividend)
{
    return selftest_div_common(ctx, divisor, 0, 0.0f, mismatches, largest_mismatching_dividend);
}

int canny_hip_selftest_div_fma_table(int index, float *divisor, float *c)
{
    const unsigned(*table)[2] = nullptr;
    const int n = gaussian_fma_div_table(&table);
    if (index < 0 || index >= n || !divisor || !c) return CANNY_HIP_ERR_INVALID;
    std::memcpy(divisor, &table[index][0], sizeof(float));
    std::memcpy(c, &table[index][1], sizeof(float));
    return CANNY_HIP_OK;
}

int canny_hip_selftest_div_fma(canny_hip_ctx *ctx, float divisor, float c, unsigned long long *mismatches,
                               float *largest_mismatching_dividend)
{
    return selftest_div_common(ctx, divisor, 1, c, mismatches, largest_mismatching_dividend);
}

} // extern "C"
