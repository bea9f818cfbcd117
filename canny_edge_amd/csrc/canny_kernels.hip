// canny_kernels.hip -- hand-written HIP kernels for the Canny hot path on MI355X (gfx950, wave64).
//
// Bit-exactness rules that shape this file:
//   * The Gaussian is a chain of separately rounded f32 multiplies and adds followed by an IEEE
//     divide and a truncating cast (reference src/utils.cpp:37-64).  Everything uses __fmul_rn /
//     __fadd_rn / __fdiv_rn and the file is built with -ffp-contract=off: an FMA would change pixels.
//   * Magnitude is floor(sqrt(gx^2+gy^2)) and the angle bin is decided with integers only; both
//     device rules are proven equal to the reference's libm expressions over the whole reachable
//     domain (tests/test_oracle_golden.py on the CPU, canny_hip_selftest_mag_angle on the GPU).
//   * Three different border conventions (H4 in SURVEY.md): Gaussian = truncate + renormalise,
//     Sobel = clamp in-axis + drop off-axis, NMS / hysteresis = skip.
#include "canny_kernels.h"

#include <hip/hip_ext.h>

namespace canny {

// ================================================================================================
// Small device helpers
// ================================================================================================
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_up(lo, 1);
    hi = __shfl_up(hi, 1);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_down(lo, 1);
    hi = __shfl_down(hi, 1);
    return ((uint64_t)hi << 32) | lo;
}

// the same for a wave-uniform lane number: two v_readlane_b32, no LDS crossbar
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int uniform_lane)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, uniform_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), uniform_lane);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t lane_u64(uint64_t v, int src_lane)
{
    unsigned lo = __shfl((unsigned)v, src_lane), hi = __shfl((unsigned)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

// ---- gradient -> magnitude / angle bin --------------------------------------------------------
// General domain: gx, gy are any values a short can hold (reference src/utils.cpp:212-231).
__device__ __forceinline__ int magnitude_general(int gx, int gy)
{
    long long n = (long long)gx * gx + (long long)gy * gy;
    return (int)(int16_t)(int)sqrt((double)n); // f64 sqrt is correctly rounded, like the host's
}
// Exact-math binning with A=gx^2, B=gy^2, P=gx*gy:
//   bin 0  iff 2|P| <= A-B  (|gy| <= |gx| tan 22.5deg; equality only at gx=gy=0)
//   bin 90 iff 2|P| <  B-A  (|gy| >  |gx| tan 67.5deg)
//   else 45 when gx,gy have the same sign, 135 otherwise.
__device__ __forceinline__ int angle_bin_general(int gx, int gy)
{
    long long A = (long long)gx * gx, B = (long long)gy * gy, P = (long long)gx * gy;
    long long twoP = 2 * (P < 0 ? -P : P);
    if (twoP <= A - B) return 0;
    if (twoP < B - A) return 90;
    return P > 0 ? 45 : 135;
}
// 8-bit domain: |gx|,|gy| <= 1020 so every product fits 24-bit multipliers and f32 is exact.
// v_sqrt_f32 is a 1-ulp approximation; trunc(sqrt(n + 0.5)) is still exactly floor(sqrt(n)) for
// n <= 2*1020^2 (margin 1.42 ulp, proven in tests/test_oracle_golden.py).
__device__ __forceinline__ int magnitude_d8(int gx, int gy)
{
    int n = __mul24(gx, gx) + __mul24(gy, gy);
    return (int)__builtin_amdgcn_sqrtf((float)n + 0.5f);
}
__device__ __forceinline__ int angle_bin_d8(int gx, int gy)
{
    int A = __mul24(gx, gx), B = __mul24(gy, gy), P = __mul24(gx, gy);
    int twoP = 2 * (P < 0 ? -P : P);
    int D = A - B;
    return (twoP <= D) ? 0 : ((twoP < -D) ? 90 : (P > 0 ? 45 : 135));
}

// Sobel derivatives of pixel (r,c) of one frame read straight from global memory.
//   gx: columns clamped to the image, rows outside the image dropped   (src/utils.cpp:114-149)
//   gy: rows clamped to the image, columns outside the image dropped   (src/utils.cpp:155-186)
// Both are stored through short by the reference, hence the int16 wrap.
__device__ __forceinline__ void sobel_at(const int16_t *__restrict__ f, int H, int W, int r, int c, int &gx, int &gy)
{
    int cl = c > 0 ? c - 1 : 0, cr = c < W - 1 ? c + 1 : W - 1;
    int ru = r > 0 ? r - 1 : 0, rd = r < H - 1 ? r + 1 : H - 1;
    const int16_t *row = f + (size_t)r * W;
    const int16_t *up = f + (size_t)ru * W;
    const int16_t *dn = f + (size_t)rd * W;
    int v = 2 * row[cr] - 2 * row[cl];
    if (r < H - 1) v += dn[cr] - dn[cl];
    if (r > 0) v += up[cr] - up[cl];
    int u = 2 * dn[c] - 2 * up[c];
    if (c < W - 1) u += dn[c + 1] - up[c + 1];
    if (c > 0) u += dn[c - 1] - up[c - 1];
    gx = (int16_t)v;
    gy = (int16_t)u;
}

// ================================================================================================
// Gaussian, general two-pass path (any window <= 129).  One thread per pixel, taps visited in
// ascending order, out-of-image taps skipped in both the sum and the weight.
// ================================================================================================
__global__ __launch_bounds__(256) void gauss_rows_generic_kernel(const uint8_t *__restrict__ img,
                                                                 float *__restrict__ tmp, int W, size_t total,
                                                                 GaussTaps t)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % (size_t)W);
        const uint8_t *row = img + (i - c);
        float acc = 0.0f, wsum = 0.0f;
        for (int k = -t.center; k <= t.center; k++) {
            int cc = c + k;
            if (cc >= 0 && cc < W) {
                float w = t.tap[t.center + k];
                acc = __fadd_rn(acc, __fmul_rn((float)row[cc], w));
                wsum = __fadd_rn(wsum, w);
            }
        }
        tmp[i] = __fdiv_rn(acc, wsum);
    }
}

__global__ __launch_bounds__(256) void gauss_cols_generic_kernel(const float *__restrict__ tmp,
                                                                 int16_t *__restrict__ out, int H, int W,
                                                                 size_t total, GaussTaps t)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t rowidx = i / (size_t)W;
        int r = (int)(rowidx % (size_t)H);
        float acc = 0.0f, wsum = 0.0f;
        for (int k = -t.center; k <= t.center; k++) {
            int rr = r + k;
            if (rr >= 0 && rr < H) {
                float w = t.tap[t.center + k];
                acc = __fadd_rn(acc, __fmul_rn(tmp[i + (ptrdiff_t)k * W], w));
                wsum = __fadd_rn(wsum, w);
            }
        }
        out[i] = (int16_t)__fdiv_rn(acc, wsum); // float -> short truncates toward zero
    }
}

static inline unsigned grid_for(size_t total, int block, unsigned cap = 256u * 32u)
{
    size_t g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

hipError_t launch_gaussian_generic(const uint8_t *img, float *tmp, int16_t *out, int height, int width,
                                   int n_frames, const GaussTaps &taps, hipStream_t stream)
{
    size_t total = (size_t)n_frames * height * width;
    unsigned grid = grid_for(total, 256);
    hipLaunchKernelGGL(gauss_rows_generic_kernel, dim3(grid), dim3(256), 0, stream, img, tmp, width, total, taps);
    hipLaunchKernelGGL(gauss_cols_generic_kernel, dim3(grid), dim3(256), 0, stream, tmp, out, height, width, total,
                       taps);
    return hipGetLastError();
}

// (the wave-marching Gaussian for windows <= 17 lives in canny_gaussian_march.hip)

// ================================================================================================
// Stand-alone Sobel / NMS stage kernels (general domain, one thread per pixel)
// ================================================================================================
__global__ __launch_bounds__(256) void xy_gradient_kernel(const int16_t *__restrict__ img, int16_t *__restrict__ gxo,
                                                          int16_t *__restrict__ gyo, int H, int W, size_t total)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t f = i / plane, p = i - f * plane;
        int r = (int)(p / (size_t)W), c = (int)(p - (size_t)r * W);
        int gx, gy;
        sobel_at(img + f * plane, H, W, r, c, gx, gy);
        gxo[i] = (int16_t)gx;
        gyo[i] = (int16_t)gy;
    }
}

__global__ __launch_bounds__(256) void sobel_kernel(const int16_t *__restrict__ img, int16_t *__restrict__ mag,
                                                    int16_t *__restrict__ ang, int H, int W, size_t total)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t f = i / plane, p = i - f * plane;
        int r = (int)(p / (size_t)W), c = (int)(p - (size_t)r * W);
        int gx, gy;
        sobel_at(img + f * plane, H, W, r, c, gx, gy);
        mag[i] = (int16_t)magnitude_general(gx, gy);
        ang[i] = (int16_t)angle_bin_general(gx, gy);
    }
}

// Keep mag iff strictly greater than both neighbours along its bin; neighbours outside the image
// are skipped; unknown angle values give 0 (the reference leaves them unwritten).
__global__ __launch_bounds__(256) void nms_kernel(const int16_t *__restrict__ mag, const int16_t *__restrict__ ang,
                                                  int16_t *__restrict__ out, int H, int W, size_t total)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t p = i % plane;
        int r = (int)(p / (size_t)W), c = (int)(p - (size_t)r * W);
        bool hl = c > 0, hr = c < W - 1, hu = r > 0, hd = r < H - 1;
        int m = mag[i];
        int a = ang[i];
        bool keep = true;
        if (a == 0) {
            if (hl && m <= mag[i - 1]) keep = false;
            if (hr && m <= mag[i + 1]) keep = false;
        } else if (a == 45) {
            if (hr && hu && m <= mag[i + 1 - W]) keep = false;
            if (hl && hd && m <= mag[i - 1 + W]) keep = false;
        } else if (a == 90) {
            if (hu && m <= mag[i - W]) keep = false;
            if (hd && m <= mag[i + W]) keep = false;
        } else if (a == 135) {
            if (hl && hu && m <= mag[i - 1 - W]) keep = false;
            if (hr && hd && m <= mag[i + 1 + W]) keep = false;
        } else {
            keep = false;
        }
        out[i] = keep ? (int16_t)m : (int16_t)0;
    }
}

hipError_t launch_xy_gradient(const int16_t *img, int16_t *gx, int16_t *gy, int height, int width, int n_frames,
                              hipStream_t stream)
{
    size_t total = (size_t)n_frames * height * width;
    hipLaunchKernelGGL(xy_gradient_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, img, gx, gy, height,
                       width, total);
    return hipGetLastError();
}
hipError_t launch_sobel(const int16_t *img, int16_t *mag, int16_t *angle, int height, int width, int n_frames,
                        hipStream_t stream)
{
    size_t total = (size_t)n_frames * height * width;
    hipLaunchKernelGGL(sobel_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, img, mag, angle, height, width,
                       total);
    return hipGetLastError();
}
hipError_t launch_nms(const int16_t *mag, const int16_t *angle, int16_t *out, int height, int width, int n_frames,
                      hipStream_t stream)
{
    size_t total = (size_t)n_frames * height * width;
    hipLaunchKernelGGL(nms_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, mag, angle, out, height, width,
                       total);
    return hipGetLastError();
}

// ================================================================================================
// Fused Sobel + NMS, LDS-tiled.  One 256-thread workgroup per 64x32 output tile:
//   phase A  stage the smoothed tile with a 2-pixel halo in LDS
//   phase B  magnitude + bin for the tile with a 1-pixel halo, kept in LDS (never written to HBM)
//   phase C  directional strict-max test, one coalesced s16 store per pixel
// HBM traffic = 2 B/px in + 2 B/px out (+ halo re-reads, which the per-XCD L2 absorbs).
// ================================================================================================
constexpr int SN_TW = 64, SN_TH = 32;

template <bool D8>
__global__ __launch_bounds__(256) void sobel_nms_tile_kernel(const int16_t *__restrict__ in,
                                                             int16_t *__restrict__ out, int H, int W)
{
    __shared__ int16_t sm[SN_TH + 4][SN_TW + 4];
    __shared__ int16_t mg[SN_TH + 2][SN_TW + 2];
    __shared__ uint8_t bn[SN_TH + 2][SN_TW + 2];

    const size_t plane = (size_t)H * W;
    const int16_t *f = in + (size_t)blockIdx.z * plane;
    int16_t *o = out + (size_t)blockIdx.z * plane;
    const int x0 = blockIdx.x * SN_TW, y0 = blockIdx.y * SN_TH;
    const int tid = threadIdx.x;

    for (int i = tid; i < (SN_TH + 4) * (SN_TW + 4); i += 256) {
        int ly = i / (SN_TW + 4), lx = i - ly * (SN_TW + 4);
        int r = y0 - 2 + ly, c = x0 - 2 + lx;
        int16_t v = 0;
        if (r >= 0 && r < H && c >= 0 && c < W) v = f[(size_t)r * W + c];
        sm[ly][lx] = v;
    }
    __syncthreads();

    for (int i = tid; i < (SN_TH + 2) * (SN_TW + 2); i += 256) {
        int my = i / (SN_TW + 2), mx = i - my * (SN_TW + 2);
        int r = y0 - 1 + my, c = x0 - 1 + mx;
        int m = 0, b = 0;
        if (r >= 0 && r < H && c >= 0 && c < W) {
            // LDS coordinates of (r,c) are (my+1, mx+1); clamp/drop decided in image coordinates
            int ly = my + 1, lx = mx + 1;
            int lxl = c > 0 ? lx - 1 : lx, lxr = c < W - 1 ? lx + 1 : lx;
            int lyu = r > 0 ? ly - 1 : ly, lyd = r < H - 1 ? ly + 1 : ly;
            int v = 2 * sm[ly][lxr] - 2 * sm[ly][lxl];
            if (r < H - 1) v += sm[ly + 1][lxr] - sm[ly + 1][lxl];
            if (r > 0) v += sm[ly - 1][lxr] - sm[ly - 1][lxl];
            int u = 2 * sm[lyd][lx] - 2 * sm[lyu][lx];
            if (c < W - 1) u += sm[lyd][lx + 1] - sm[lyu][lx + 1];
            if (c > 0) u += sm[lyd][lx - 1] - sm[lyu][lx - 1];
            int gx = (int16_t)v, gy = (int16_t)u;
            if (D8) {
                m = magnitude_d8(gx, gy);
                b = angle_bin_d8(gx, gy);
            } else {
                m = magnitude_general(gx, gy);
                b = angle_bin_general(gx, gy);
            }
        }
        mg[my][mx] = (int16_t)m;
        bn[my][mx] = (uint8_t)b;
    }
    __syncthreads();

    const int tx = tid & 63, ty = tid >> 6;
    const int c = x0 + tx;
    if (c < W) {
        const bool hl = c > 0, hr = c < W - 1;
        for (int yy = ty; yy < SN_TH; yy += 4) {
            int r = y0 + yy;
            if (r >= H) break;
            const bool hu = r > 0, hd = r < H - 1;
            int my = yy + 1, mx = tx + 1;
            int m = mg[my][mx];
            int b = bn[my][mx];
            bool keep = true;
            if (b == 0) {
                if (hl && m <= mg[my][mx - 1]) keep = false;
                if (hr && m <= mg[my][mx + 1]) keep = false;
            } else if (b == 45) {
                if (hr && hu && m <= mg[my - 1][mx + 1]) keep = false;
                if (hl && hd && m <= mg[my + 1][mx - 1]) keep = false;
            } else if (b == 90) {
                if (hu && m <= mg[my - 1][mx]) keep = false;
                if (hd && m <= mg[my + 1][mx]) keep = false;
            } else {
                if (hl && hu && m <= mg[my - 1][mx - 1]) keep = false;
                if (hr && hd && m <= mg[my + 1][mx + 1]) keep = false;
            }
            o[(size_t)r * W + c] = keep ? (int16_t)m : (int16_t)0;
        }
    }
}

hipError_t launch_sobel_nms(const int16_t *smoothed, int16_t *out, int height, int width, int n_frames, bool domain8,
                            hipStream_t stream)
{
    dim3 grid((width + SN_TW - 1) / SN_TW, (height + SN_TH - 1) / SN_TH, n_frames);
    if (domain8)
        hipLaunchKernelGGL(sobel_nms_tile_kernel<true>, grid, dim3(256), 0, stream, smoothed, out, height, width);
    else
        hipLaunchKernelGGL(sobel_nms_tile_kernel<false>, grid, dim3(256), 0, stream, smoothed, out, height, width);
    return hipGetLastError();
}

// ================================================================================================
// Hysteresis on bit-planes.
//   classify : one wave per 64-pixel word; __ballot turns the two threshold tests into the
//              `connectable` (v >= min) and `strong` (connectable && v >= max) bit-planes.
//   propagate: one wave per 64x64 tile, lane = row.  Bit-parallel 8-neighbour dilation plus a
//              Kogge-Stone fill along each row, repeated until the tile is stable (wave ballot).
//              Tiles whose border changed stamp their neighbours for the next sweep.
//   finalize : reached -> EDGE (255), everything else -> 0.
// The fixed point ("connectable pixels reachable from a strong pixel") is order independent, so it
// equals the reference's FIFO flood (src/utils.cpp:322-427) as long as the one missing directed
// edge (1,0)->(0,1) (src/utils.cpp:378,399: `current - width > 0`) is honoured.
// ================================================================================================
__device__ __forceinline__ size_t word_index(const HystGeom &g, int f, int y, int wx)
{
    return ((((size_t)f * g.tiles_y + (y >> 6)) * g.tiles_x + wx) << 6) + (y & 63);
}

__global__ __launch_bounds__(256) void hyst_classify_kernel(const int16_t *__restrict__ cand,
                                                            uint64_t *__restrict__ strong,
                                                            uint64_t *__restrict__ conn, HystGeom g, int lo, int hi,
                                                            unsigned *domain_flag)
{
    const int lane = threadIdx.x & 63;
    const size_t rows_padded = (size_t)g.tiles_y * kTile;
    const size_t n_words = (size_t)g.n_frames * rows_padded * g.tiles_x;
    const size_t wave_stride = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t wv = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < n_words; wv += wave_stride) {
        int wx = (int)(wv % (size_t)g.tiles_x);
        size_t t = wv / (size_t)g.tiles_x;
        int y = (int)(t % rows_padded);
        int f = (int)(t / rows_padded);
        int x = wx * kTile + lane;
        bool inside = (y < g.height) && (x < g.width);
        int v = 0;
        if (inside) v = cand[((size_t)f * g.height + y) * g.width + x];
        bool c = inside && v >= lo;
        bool s = c && v >= hi;
        if (inside && lo <= 0 && v < lo) atomicOr(domain_flag, 1u);
        uint64_t mc = __ballot(c), ms = __ballot(s);
        if (lane == 0) {
            size_t idx = word_index(g, f, y, wx);
            conn[idx] = mc;
            strong[idx] = ms;
        }
    }
}

// row above / below (lane = row) through wave-wide DPP shifts; the wave's end lanes read 0
__device__ __forceinline__ uint64_t row_above_u64(uint64_t v)
{
    unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t row_below_u64(uint64_t v)
{
    unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), 0x130, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}

// Row flood: every run of ones in p that contains a bit of g (g is a subset of p) becomes selected entirely.
// The carry chain of an adder does the walk: adding a seed to its run of ones ripples upward to the run's end,
// so the bits that p + g flips inside p are exactly [lowest seed of the run .. end of the run] minus the other
// seeds (which are in g anyway); the carry stops in the zero above the run, so runs do not disturb each other,
// and a carry out of bit 63 is simply lost.  The downward direction is the same on bit-reversed words.
// ~20 32-bit instructions against ~100 (24 of them 64-bit shifts) for the two Kogge-Stone ladders this replaces:
// sweep 0 floods 170 k tiles per 128-frame batch and is bound by exactly this arithmetic.
__device__ __forceinline__ uint64_t fill_runs(uint64_t p, uint64_t g)
{
    const uint64_t up = ((p + g) ^ p) & p;
    const uint64_t pr = __brevll(p), gr = __brevll(g);
    const uint64_t down = __brevll(((pr + gr) ^ pr) & pr);
    return g | up | down;
}

// Scheduling words of one hysteresis call (all zero before sweep 0):
//   stamp[tiles]    sweep at which a tile was last queued (dedupes pushes within a sweep)
//   queue[2][tiles] tiles to run in sweep k are queue[k & 1][0 .. count[k % 3])
//   count[3]        sweep k reads count[k%3], appends to count[(k+1)%3] and clears count[(k+2)%3]
//   fq0/fq1[tiles]  the same two queues PER FRAME (frame f owns [f * tiles_per_frame, (f+1) * tiles_per_frame)),
//   fcount[4 * frames] ... and their counters (3 used per frame): what hyst_tail_kernel walks, one workgroup per frame
struct HystSched {
    unsigned *stamp, *queue0, *queue1, *count;
    unsigned *fq0, *fq1, *fcount;
};
constexpr int kSweep0Tiles = 2; // tiles per wave in sweep 0
__device__ __forceinline__ HystSched make_sched(unsigned *words, int tiles)
{
    HystSched s;
    s.stamp = words;
    s.queue0 = words + tiles;
    s.queue1 = words + 2 * (size_t)tiles;
    s.count = words + 3 * (size_t)tiles;
    s.fq0 = words + 3 * (size_t)tiles + 4;
    s.fq1 = words + 4 * (size_t)tiles + 4;
    s.fcount = words + 5 * (size_t)tiles + 4;
    return s;
}

// edges (may be null): edge map that already holds the initially strong pixels (written by the kernel that
// filled the planes); every pixel this sweep promotes is written there at once, so no finalize pass is needed.
// A tile's own strong words and what it sees of its eight neighbours, as loaded (nothing here waits for memory).
struct TileIn {
    uint64_t s0, su, sd;          // own word; bottom row of the tile above; top row of the tile below
    unsigned lb, rb;              // this row's column 63 of the left tile / column 0 of the right tile
    unsigned ul, ur, dl, dr;      // the four corner pixels
};

// c: the tile's connectable word of this lane if the caller already has it (kAllHalo otherwise).  A neighbour's pixels
// can only ever pull pixels of this tile that are connectable AND lie on the border facing that neighbour (gsel =
// s | (c & nb), and a halo bit enters nb only in the first / last column or row): when the connectable plane has no
// pixel on a border -- most borders of most tiles: 64 of a tile's 4096 pixels, weak candidates are a few per cent --
// that neighbour is not read at all.  The left and right neighbours are the expensive ones: a whole tile (512 bytes)
// each for one bit per row, as much as the tile's own two words.
constexpr uint64_t kAllHalo = ~0ull;
__device__ __forceinline__ TileIn load_tile(int t, int lane, const uint64_t *__restrict__ strong, const HystGeom &g,
                                            uint64_t c = kAllHalo)
{
    const int tpf = g.tiles_x * g.tiles_y;
    const int tt = t % tpf;
    const int ty = tt / g.tiles_x, tx = tt - ty * g.tiles_x;
    const uint64_t c_top = readlane_u64(c, 0), c_bot = readlane_u64(c, 63); // wave-uniform
    const bool needL = __any((c & 1ull) != 0), needR = __any((c >> 63) != 0);
    const bool hasU = ty > 0 && c_top != 0, hasD = ty < g.tiles_y - 1 && c_bot != 0;
    const bool hasL = tx > 0 && needL, hasR = tx < g.tiles_x - 1 && needR;
    const size_t base = (size_t)t * kTile;
    const size_t rowstep = (size_t)g.tiles_x * kTile; // words between vertically adjacent tiles
    TileIn in;
    in.s0 = strong[base + lane];
    // halo from the eight neighbouring tiles (read once per sweep; a change made there during
    // this sweep re-stamps us for the next one)
    in.su = hasU ? strong[base - rowstep + 63] : 0;
    in.sd = hasD ? strong[base + rowstep] : 0;
    in.lb = hasL ? (unsigned)(strong[base - kTile + lane] >> 63) : 0u;
    in.rb = hasR ? (unsigned)(strong[base + kTile + lane] & 1u) : 0u;
    in.ul = (hasU && hasL && (c_top & 1ull)) ? (unsigned)(strong[base - rowstep - kTile + 63] >> 63) : 0u;
    in.ur = (hasU && hasR && (c_top >> 63)) ? (unsigned)(strong[base - rowstep + kTile + 63] & 1u) : 0u;
    in.dl = (hasD && hasL && (c_bot & 1ull)) ? (unsigned)(strong[base + rowstep - kTile] >> 63) : 0u;
    in.dr = (hasD && hasR && (c_bot >> 63)) ? (unsigned)(strong[base + rowstep + kTile] & 1u) : 0u;
    return in;
}

// c = this lane's word of the tile's connectable plane (the caller has checked that the tile has any), in = the
// tile as load_tile() returned it.
// to_frame: neighbours are queued in their FRAME's queue (for hyst_tail_kernel) instead of the batch-wide one.
// push(nb): schedules tile nb for sweep iter + 1 (once per sweep); called by up to eight lanes at once, each with a
// different neighbour tile.
template <class Push>
__device__ __forceinline__ void process_tile_with(int t, int lane, uint64_t *__restrict__ strong,
                                                  unsigned *__restrict__ last_change, int iter, const HystGeom &g,
                                                  int16_t *__restrict__ edges, int edge_value, uint64_t c,
                                                  const TileIn &in, Push push_tile)
{
    const int tpf = g.tiles_x * g.tiles_y;
    const int tt = t % tpf;
    const int ty = tt / g.tiles_x, tx = tt - ty * g.tiles_x;
    const bool hasU = ty > 0, hasD = ty < g.tiles_y - 1, hasL = tx > 0, hasR = tx < g.tiles_x - 1;
    const size_t base = (size_t)t * kTile;
    const uint64_t s0 = in.s0, su = in.su, sd = in.sd;
    const unsigned lb = in.lb, rb = in.rb, ul = in.ul, ur = in.ur, dl = in.dl, dr = in.dr;
    unsigned lb_up = __shfl_up(lb, 1), lb_dn = __shfl_down(lb, 1);
    unsigned rb_up = __shfl_up(rb, 1), rb_dn = __shfl_down(rb, 1);
    if (lane == 0) { lb_up = ul; rb_up = ur; }
    if (lane == 63) { lb_dn = dl; rb_dn = dr; }
    const uint64_t in_left = (uint64_t)(lb | lb_up | lb_dn);         // enters at bit 0
    const uint64_t in_right = (uint64_t)(rb | rb_up | rb_dn) << 63;  // enters at bit 63
    // image row 0, tile column 0: pixel (0,1) must not pull from pixel (1,0)
    const bool quirk = (ty == 0 && tx == 0 && lane == 0);

    uint64_t s = s0;
    for (;;) {
        uint64_t up = row_above_u64(s), dn = row_below_u64(s);
        if (lane == 0) up = su;
        if (lane == 63) dn = sd;
        uint64_t d = up | s | dn;
        uint64_t d_from_left = quirk ? (up | s | (dn & ~1ull)) : d; // what column c may pull from column c-1
        uint64_t nb = d | (d_from_left << 1) | (d >> 1) | in_left | in_right;
        uint64_t gsel = s | (c & nb);
        // Nothing to add from the 8-neighbourhood: done.  (The row flood below only adds connectable pixels that
        // are horizontal neighbours of selected ones, i.e. a subset of what a later round's `nb` would add, so it
        // cannot find anything either; every tile's last round ends here, and so does the only round of the two
        // thirds of sweep 0's tiles that have no weak pixel next to a strong one.)
        if (!__any(gsel != s)) break;
        // flood along the row through runs of connectable pixels, both directions
        s = fill_runs(c, gsel);
    }

    const uint64_t chg = s ^ s0;
    const uint64_t rows_changed = __ballot(chg != 0); // bit r: row r of the tile gained pixels
    if (rows_changed != 0) {
        strong[base + lane] = s;
        if (edges) {
            // (padding bits are never connectable, so every set bit lies inside the image; the row and column tests
            // only keep a corrupted plane from turning into an out-of-bounds store)
            const int f = t / tpf;
            const int cols = min(kTile, g.width - tx * kTile), rows = min(kTile, g.height - ty * kTile);
            int16_t *const tile0 = edges + ((size_t)f * g.height + (size_t)ty * kTile) * g.width + tx * kTile;
            if (__popcll(rows_changed) <= 16) {
                // few rows, possibly long runs in them (a horizontal edge): one store instruction per changed row,
                // lane = column.  With lane = row a run of 60 promoted pixels is 60 dependent loop trips of one
                // lane -- 9 k cycles of the 14 k a sweep of hyst_tail_kernel took.
                for (uint64_t m = rows_changed; m != 0; m &= m - 1) { // wave-uniform
                    const int r = __builtin_ctzll(m);
                    const uint64_t word = readlane_u64(chg, r);
                    if (((word >> lane) & 1ull) != 0 && lane < cols && r < rows)
                        tile0[(size_t)r * g.width + lane] = (int16_t)edge_value;
                }
            } else if (lane < rows) { // many rows (a vertical or diagonal edge: a pixel or two per row): lane = row
                int16_t *const rowp = tile0 + (size_t)lane * g.width;
                for (uint64_t m = chg; m != 0; m &= m - 1) {
                    const int b = __builtin_ctzll(m);
                    if (b < cols) rowp[b] = (int16_t)edge_value;
                }
            }
        }
        // Queue every neighbour whose facing border changed for the next sweep (once per sweep).  Lanes 0..7 take
        // one neighbour each -- up, down, left, right, then the four corners -- so that the stamp exchange and the
        // queue append of all of them are in flight together instead of one returning atomic after the other.
        const uint64_t top64 = readlane_u64(chg, 0);  // changes in the tile's first row
        const uint64_t bot = readlane_u64(chg, 63);   // ... and in its last row
        const bool anyL = __any((chg & 1ull) != 0), anyR = __any((chg >> 63) != 0);
        const int dy = (lane == 0 || lane == 4 || lane == 5) ? -1 : ((lane == 1 || lane == 6 || lane == 7) ? 1 : 0);
        const int dx = (lane == 2 || lane == 4 || lane == 6) ? -1 : ((lane == 3 || lane == 5 || lane == 7) ? 1 : 0);
        const uint64_t facing = dy < 0 ? top64 : bot;
        const bool hit = dy == 0 ? (dx < 0 ? anyL : anyR)
                                 : (dx == 0 ? facing != 0 : (dx < 0 ? (facing & 1ull) != 0 : (facing >> 63) != 0));
        const bool there = (dy >= 0 || hasU) && (dy <= 0 || hasD) && (dx >= 0 || hasL) && (dx <= 0 || hasR);
        const bool want = lane < 8 && hit && there;
        if (want) push_tile(t + dy * g.tiles_x + dx);
        if (__any(want) && lane == 0) atomicMax(last_change, (unsigned)iter + 1u);
    }
}

// ... with the queues in global memory (batch-wide, or per frame for the sweep that hands over to hyst_tail_kernel)
__device__ __forceinline__ void process_tile(int t, int lane, uint64_t *__restrict__ strong, const HystSched &sch,
                                             unsigned *__restrict__ last_change, int iter, const HystGeom &g,
                                             int16_t *__restrict__ edges, int edge_value, uint64_t c,
                                             const TileIn &in, bool to_frame = false)
{
    const unsigned nxt = (unsigned)iter + 1u;
    unsigned *q = (nxt & 1u) ? sch.queue1 : sch.queue0;
    unsigned *cnt = sch.count + nxt % 3u;
    if (to_frame) {
        const int tpf = g.tiles_x * g.tiles_y;
        const int f = t / tpf;
        q = ((nxt & 1u) ? sch.fq1 : sch.fq0) + (size_t)f * tpf;
        cnt = sch.fcount + 4 * (size_t)f + nxt % 3u;
    }
    process_tile_with(t, lane, strong, last_change, iter, g, edges, edge_value, c, in, [&](int nb) {
        if (atomicExch(&sch.stamp[nb], nxt) != nxt) q[atomicAdd(cnt, 1u)] = (unsigned)nb;
    });
}

__device__ __forceinline__ void propagate_tile(int t, int lane, uint64_t *__restrict__ strong,
                                               const uint64_t *__restrict__ conn, const HystSched &sch,
                                               unsigned *__restrict__ last_change, int iter, const HystGeom &g,
                                               int16_t *__restrict__ edges, int edge_value, uint64_t c,
                                               bool to_frame = false)
{
    // c = this lane's word of the tile's connectable plane, loaded by the caller.  Only connectable pixels can
    // ever be added, so a tile without a single one (flat regions: most tiles of a natural frame) cannot
    // change: leave before the strong plane and the nine halo loads are touched.
    if (!__any(c != 0)) return;
    const TileIn in = load_tile(t, lane, strong, g, c);
    process_tile(t, lane, strong, sch, last_change, iter, g, edges, edge_value, c, in, to_frame);
}

// Sweep 0 visits every tile (grid = tiles / (4 kSweep0Tiles) workgroups); later sweeps are launched with a small fixed
// grid whose waves walk the work queue, so a sweep with little or nothing to do costs one short launch
// instead of 130 k waves that each read a stamp and exit.
__global__ __launch_bounds__(256) void hyst_propagate_kernel(uint64_t *__restrict__ strong,
                                                             const uint64_t *__restrict__ conn,
                                                             unsigned *__restrict__ sched_words,
                                                             unsigned *__restrict__ last_change, int iter, HystGeom g,
                                                             int16_t *__restrict__ edges, int edge_value, int to_frame)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int n_waves = gridDim.x * 4;
    const int tiles = g.tiles();
    const HystSched sch = make_sched(sched_words, tiles);
    if (wave == 0 && lane == 0) sch.count[(iter + 2) % 3] = 0; // the slot sweep iter+1 will append to
    if (iter == 0) {
        // kSweep0Tiles tiles per wave, their connectable words loaded up front: a third of the tiles are empty
        // and cost exactly this one load.  Measured per 128 x 4K batch (all sweeps): 1 tile per wave 0.254 ms,
        // 2: 0.231, 3: 0.229, 4: 0.244, 8: 0.249 -- more tiles per wave mean fewer, longer-lived waves whose
        // tile chains (connectable word -> strong word + halo -> flood -> store) run one after the other.
        for (int t0 = wave * kSweep0Tiles; t0 < tiles; t0 += n_waves * kSweep0Tiles) {
            uint64_t c[kSweep0Tiles];
#pragma unroll
            for (int k = 0; k < kSweep0Tiles; k++)
                c[k] = (t0 + k < tiles) ? conn[(size_t)(t0 + k) * kTile + lane] : 0ull;
#pragma unroll
            for (int k = 0; k < kSweep0Tiles; k++)
                propagate_tile(t0 + k, lane, strong, conn, sch, last_change, iter, g, edges, edge_value, c[k],
                               to_frame != 0);
        }
        return;
    }
    const unsigned n = __hip_atomic_load(sch.count + iter % 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned *q = (iter & 1) ? sch.queue1 : sch.queue0;
    for (unsigned i = (unsigned)wave; i < n; i += (unsigned)n_waves) {
        const int t = (int)q[i];
        propagate_tile(t, lane, strong, conn, sch, last_change, iter, g, edges, edge_value,
                       conn[(size_t)t * kTile + lane], to_frame != 0);
    }
}

// The tail of a propagation: ONE workgroup per frame runs every remaining sweep of its frame, until the frame's
// queue stays empty.  Frames never interact (the reference processes one frame per call, src/main.cpp:120-137), so a
// frame's sweeps only have to be ordered among the waves that work on that frame -- a workgroup barrier -- and no
// grid-wide barrier, no cross-workgroup visibility protocol and no host round trip is needed: after the two
// batch-wide sweeps that do the bulk of the work (sweep 1 queues into the per-frame queues), this kernel replaces
// the 6+ nearly empty launches and the host's convergence poll of the multi-launch scheme.
// The frame's two queues, their counters and the tile stamps live in LDS (48 KB: a frame has at most kTailTiles
// tiles): a sweep of this kernel is a chain of dependent steps -- how many tiles, which tile, its words, the flood,
// the pushes -- and with the scheduling words in global memory five of them were L2 round trips or returning atomics
// (~8 us per sweep, 59 us for the seven tail sweeps of a 4K frame); now the tile's own words are the only ones.
// Termination: strong bits only ever get set, a tile is queued only when a neighbour's facing border changed, so
// every frame's queue runs dry after finitely many sweeps; the loop has no other exit and needs none.
// Visibility: all waves of a workgroup run on one CU and share its L1; __syncthreads() orders their stores
// before the next sweep's loads.
constexpr int kTailTiles = 4096; // = kTailMaxTiles of canny_capi.hip (checked by launch_hyst_tail)
__global__ __launch_bounds__(1024) void hyst_tail_kernel(uint64_t *__restrict__ strong, const uint64_t *__restrict__ conn,
                                                         unsigned *__restrict__ sched_words,
                                                         unsigned *__restrict__ last_change, int first_iter, HystGeom g,
                                                         int16_t *__restrict__ edges, int edge_value)
{
    __shared__ unsigned s_queue[2][kTailTiles]; // tile numbers within the frame
    __shared__ unsigned s_stamp[kTailTiles];    // sweep for which a tile was last queued (dedupes pushes)
    __shared__ unsigned s_count[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    const int f = blockIdx.x;
    const int tpf = g.tiles_x * g.tiles_y;
    const int t_base = f * tpf;
    const HystSched sch = make_sched(sched_words, g.tiles());
    // what sweep first_iter - 1 (batch-wide, to_frame_queues) left for this frame
    int cur = first_iter & 1;
    {
        const unsigned n0 = __hip_atomic_load(sch.fcount + 4 * (size_t)f + first_iter % 3, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
        const unsigned *q0 = (cur ? sch.fq1 : sch.fq0) + (size_t)t_base;
        for (unsigned i = threadIdx.x; i < n0; i += blockDim.x)
            s_queue[cur][i] = __hip_atomic_load(q0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)t_base;
        for (int i = threadIdx.x; i < tpf; i += blockDim.x) s_stamp[i] = 0; // sweeps here are numbered >= 2
        if (threadIdx.x == 0) {
            s_count[cur] = n0;
            s_count[cur ^ 1] = 0;
        }
    }
    for (int iter = first_iter;; iter++, cur ^= 1) {
        __syncthreads(); // the previous sweep's stores and pushes (or the set-up above) are complete
        const unsigned n = s_count[cur];
        __syncthreads(); // everybody has read n
        if (n == 0) break; // uniform
        if (threadIdx.x == 0) s_count[cur] = 0; // this sweep appends to the other queue, sweep iter + 1 to this one
        unsigned *const q_next = s_queue[cur ^ 1];
        unsigned *const n_next = &s_count[cur ^ 1];
        const unsigned nxt = (unsigned)iter + 1u;
        for (unsigned i = (unsigned)wave; i < n; i += (unsigned)n_waves) {
            const int t = t_base + (int)s_queue[cur][i];
            // the tile's connectable word and its strong words + halo are requested together (a queued tile nearly
            // always has connectable pixels: it was queued because a neighbour's border towards it changed)
            const uint64_t c = conn[(size_t)t * kTile + lane];
            const TileIn in = load_tile(t, lane, strong, g);
            if (!__any(c != 0)) continue;
            process_tile_with(t, lane, strong, last_change, iter, g, edges, edge_value, c, in, [&](int nb) {
                const unsigned loc = (unsigned)(nb - t_base);
                if (atomicExch(&s_stamp[loc], nxt) != nxt) q_next[atomicAdd(n_next, 1u)] = loc;
            });
        }
    }
}

__global__ __launch_bounds__(256) void hyst_finalize_kernel(int16_t *__restrict__ cand,
                                                            const uint64_t *__restrict__ strong, HystGeom g,
                                                            int edge_value)
{
    const size_t plane = (size_t)g.height * g.width;
    const size_t total = plane * g.n_frames;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t f = i / plane, p = i - f * plane;
        int y = (int)(p / (size_t)g.width), x = (int)(p - (size_t)y * g.width);
        uint64_t w = strong[word_index(g, (int)f, y, x >> 6)];
        cand[i] = ((w >> (x & 63)) & 1ull) ? (int16_t)edge_value : (int16_t)0;
    }
}

// ---- findEdgePixels (one frame) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void fep_classify_kernel(const int16_t *__restrict__ cand,
                                                           const uint8_t *__restrict__ visited,
                                                           uint64_t *__restrict__ strong, uint64_t *__restrict__ conn,
                                                           HystGeom g, int start, int lo)
{
    const int lane = threadIdx.x & 63;
    const size_t rows_padded = (size_t)g.tiles_y * kTile;
    const size_t n_words = rows_padded * g.tiles_x;
    const size_t wave_stride = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t wv = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wv < n_words; wv += wave_stride) {
        int wx = (int)(wv % (size_t)g.tiles_x);
        int y = (int)(wv / (size_t)g.tiles_x);
        int x = wx * kTile + lane;
        bool inside = (y < g.height) && (x < g.width);
        bool c = false, s = false;
        if (inside) {
            size_t i = (size_t)y * g.width + x;
            s = ((long long)i == (long long)start);
            c = s || (cand[i] >= lo && !visited[i]);
        }
        uint64_t mc = __ballot(c), ms = __ballot(s);
        if (lane == 0) {
            size_t idx = word_index(g, 0, y, wx);
            conn[idx] = mc;
            strong[idx] = ms;
        }
    }
}

__device__ __forceinline__ bool bit_at(const uint64_t *planeS, const HystGeom &g, int y, int x)
{
    if (y < 0 || y >= g.height || x < 0 || x >= g.width) return false;
    return (planeS[word_index(g, 0, y, x >> 6)] >> (x & 63)) & 1ull;
}

// Reached pixels become EDGE and visited, except `start`, which the reference enqueues without
// marking it: it only becomes visited when a reached neighbour pushes it again, which needs the
// directed edge neighbour->start and EDGE >= min_val.
__global__ __launch_bounds__(256) void fep_finalize_kernel(int16_t *__restrict__ cand, uint8_t *__restrict__ visited,
                                                           const uint64_t *__restrict__ strong, HystGeom g, int start,
                                                           int lo)
{
    const size_t total = (size_t)g.height * g.width;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int y = (int)(i / (size_t)g.width), x = (int)(i - (size_t)y * g.width);
        if (!bit_at(strong, g, y, x)) continue;
        cand[i] = 255;
        if ((long long)i != (long long)start) {
            visited[i] = 1;
        } else if (255 >= lo) {
            bool pushed = false;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dy && !dx) continue;
                    int py = y + dy, px = x + dx;
                    if (!bit_at(strong, g, py, px)) continue;
                    if (py == 1 && px == 0 && y == 0 && x == 1) continue; // (1,0) never pushes (0,1)
                    pushed = true;
                }
            if (pushed) visited[i] = 1;
        }
    }
}

// saturating per-half a - b: the sign of each half is the sign of the true difference for any shorts
__device__ __forceinline__ uint32_t pk_sub_sat_i16(uint32_t a, uint32_t b)
{
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, a),
                                                                      __builtin_bit_cast(s16x2_t, b)));
}

// Vectorised classify / finalize for widths that are a multiple of 8: one lane = 8 pixels = one
// 16-byte load (store) and one BYTE of each bit-plane (byte k of a word holds pixels 8k..8k+7).
// Thread i owns byte i of the tile-major planes, i.e. tile i/512, row (i%512)/8, 8-pixel group i%8:
// a wave covers a 64-pixel x 8-row patch, reads eight 128-byte row segments and writes 64 CONTIGUOUS
// plane bytes (byte stores at the old row-major mapping hit eight 512-byte-strided words per wave and
// cost 4.6x write amplification in WRITE_SIZE).
// Launched on a 3-D grid (ceil(tiles_x/2), tiles_y, n_frames) of 256-thread workgroups: a workgroup owns two
// horizontally adjacent tiles and every thread four 8-pixel groups (k = 0..3), so no thread ever divides
// (a flat index needed two 64-bit divisions per thread) and each thread keeps four 16-byte accesses in flight
// (one group per thread meant a million single-load waves per launch: wave-launch bound).
constexpr int kPatchItems = 4;
__device__ __forceinline__ bool patch_coords(const HystGeom &g, int k, size_t &i, int &f, int &y, int &x0)
{
    const int tx = (int)blockIdx.x * 2 + (k >> 1), ty = (int)blockIdx.y;
    const int within = (k & 1) * 256 + (int)threadIdx.x;
    f = (int)blockIdx.z;
    i = ((((size_t)f * g.tiles_y + ty) * g.tiles_x + tx) << 9) + within;
    y = ty * kTile + (within >> 3);
    x0 = tx * kTile + (within & 7) * 8;
    return tx < g.tiles_x && y < g.height && x0 < g.width; // width % 8 == 0: a group is all inside or all outside
}
__device__ __forceinline__ bool patch_tile_exists(const HystGeom &g, int k)
{
    return (int)blockIdx.x * 2 + (k >> 1) < g.tiles_x;
}

__global__ __launch_bounds__(256) void hyst_classify8_kernel(const int16_t *__restrict__ cand,
                                                             uint8_t *__restrict__ strong, uint8_t *__restrict__ conn,
                                                             HystGeom g, int lo, int hi, unsigned *domain_flag)
{
    const bool fast = lo >= -32768 && lo <= 32767 && hi >= -32768 && hi <= 32767; // wave-uniform
    const uint32_t lo2 = ((uint32_t)lo & 0xffffu) * 0x10001u, hi2 = ((uint32_t)hi & 0xffffu) * 0x10001u;
    uint4 px4[kPatchItems];
    size_t idx[kPatchItems];
    bool inside[kPatchItems];
#pragma unroll
    for (int k = 0; k < kPatchItems; k++) { // all loads first
        int f, y, x0;
        inside[k] = patch_coords(g, k, idx[k], f, y, x0);
        px4[k] = make_uint4(0u, 0u, 0u, 0u);
        if (inside[k]) __builtin_memcpy(&px4[k], cand + ((size_t)f * g.height + y) * g.width + x0, 16);
    }
#pragma unroll
    for (int k = 0; k < kPatchItems; k++) {
        if (!patch_tile_exists(g, k)) continue; // odd tiles_x: the second tile of the last workgroup
        const size_t i = idx[k];
        unsigned cbits = 0, sbits = 0;
        if (inside[k]) {
            const uint4 v = px4[k];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            if (fast) {
                // Packed path (thresholds fit a short): 4.3 VALU ops per pixel instead of 18.7.
                // Regroup so that the low halves hold pixels 0..3 and the high halves pixels 4..7; a
                // saturating packed subtract leaves the sign bit of each half clear iff px >= threshold;
                // shifting register k right by 3-k lines the eight sign bits up as two nibbles.
                const uint32_t r[4] = {__builtin_amdgcn_perm(w[2], w[0], 0x05040100u),  // (px0, px4)
                                       __builtin_amdgcn_perm(w[2], w[0], 0x07060302u),  // (px1, px5)
                                       __builtin_amdgcn_perm(w[3], w[1], 0x05040100u),  // (px2, px6)
                                       __builtin_amdgcn_perm(w[3], w[1], 0x07060302u)}; // (px3, px7)
                uint32_t dc[4], ds[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t a = pk_sub_sat_i16(r[k], lo2); // sign set  <=>  px <  lo
                    const uint32_t b = pk_sub_sat_i16(r[k], hi2); // sign set  <=>  px <  hi
                    dc[k] = a & 0x80008000u;
                    ds[k] = (a | b) & 0x80008000u;
                }
                const uint32_t xc = (dc[0] >> 3) | (dc[1] >> 2) | (dc[2] >> 1) | dc[3];
                const uint32_t xs = (ds[0] >> 3) | (ds[1] >> 2) | (ds[2] >> 1) | ds[3];
                cbits = (((xc >> 12) & 0xfu) | ((xc >> 24) & 0xf0u)) ^ 0xffu;
                sbits = (((xs >> 12) & 0xfu) | ((xs >> 24) & 0xf0u)) ^ 0xffu;
                if (lo <= 0 && cbits != 0xffu) atomicOr(domain_flag, 1u);
            } else {
                bool below = false;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int px = (e & 1) ? ((int)w[e >> 1] >> 16) : (int)(short)(w[e >> 1] & 0xffffu);
                    const bool c = px >= lo;
                    cbits |= (unsigned)c << e;
                    sbits |= (unsigned)(c && px >= hi) << e;
                    below |= px < lo;
                }
                if (lo <= 0 && below) atomicOr(domain_flag, 1u);
            }
        }
        conn[i] = (uint8_t)cbits;
        strong[i] = (uint8_t)sbits;
    }
}

__global__ __launch_bounds__(256) void hyst_finalize8_kernel(int16_t *__restrict__ cand,
                                                             const uint8_t *__restrict__ strong, HystGeom g,
                                                             int edge_value)
{
    const uint32_t ev = (uint32_t)(uint16_t)edge_value;
#pragma unroll
    for (int k = 0; k < kPatchItems; k++) {
        int f, y, x0;
        size_t i;
        if (!patch_coords(g, k, i, f, y, x0)) continue;
        const unsigned b = strong[i];
        // bit 2k -> bit 0 and bit 2k+1 -> bit 16 of register k, then one 24-bit multiply by the edge value
        const uint32_t t = b | (b << 15);
        uint4 v;
        v.x = __umul24(t & 0x00010001u, ev);
        v.y = __umul24((t >> 2) & 0x00010001u, ev);
        v.z = __umul24((t >> 4) & 0x00010001u, ev);
        v.w = __umul24((t >> 6) & 0x00010001u, ev);
        __builtin_memcpy(cand + ((size_t)f * g.height + y) * g.width + x0, &v, 16);
    }
}

hipError_t launch_hyst_classify(const int16_t *cand, uint64_t *strong, uint64_t *conn, const HystGeom &g, int min_val,
                                int max_val, unsigned *domain_flag, hipStream_t stream)
{
    size_t n_words = (size_t)g.n_frames * g.tiles_y * kTile * g.tiles_x;
    if (g.width % 8 == 0)
        hipLaunchKernelGGL(hyst_classify8_kernel, dim3((g.tiles_x + 1) / 2, g.tiles_y, g.n_frames), dim3(256), 0, stream,
                           cand, (uint8_t *)strong, (uint8_t *)conn, g, min_val, max_val, domain_flag);
    else
        hipLaunchKernelGGL(hyst_classify_kernel, dim3(grid_for(n_words * 64, 256)), dim3(256), 0, stream, cand,
                           strong, conn, g, min_val, max_val, domain_flag);
    return hipGetLastError();
}
// Start of every hysteresis call.  With zero_pad it zeroes the parts of both planes that lie outside the image
// (rows >= H of the last tile row, columns >= W of the last tile column).  The fused Sobel+NMS+classify kernel writes in-image plane bytes only; the
// classify kernels above write whole tiles and do not need this.  Writes pad bytes only, so it may run
// before, after or concurrently with the kernel that fills the image part.  Requires width % 8 == 0.
// The same kernel clears the scheduling words and the two flag words of the propagation (one launch instead
// of two memsets and a kernel: each of those costs ~10 us of launch gap on the stream).
__global__ __launch_bounds__(256) void hyst_prepare_kernel(uint64_t *__restrict__ strong, uint64_t *__restrict__ conn,
                                                           HystGeom g, int pad_waves, unsigned *__restrict__ sched,
                                                           unsigned n_sched, unsigned *__restrict__ flags,
                                                           int n_flags)
{
    const int lane = (int)(threadIdx.x & 63);
    const int wave = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (wave < pad_waves) {
        // one wave per edge tile and frame (k < tiles_x: last tile row, else: last tile column), lane = row
        const int per_frame = g.tiles_x + g.tiles_y;
        const int f = wave / per_frame, k = wave - f * per_frame;
        const int tx = k < g.tiles_x ? k : g.tiles_x - 1;
        const int ty = k < g.tiles_x ? g.tiles_y - 1 : k - g.tiles_x;
        const size_t w = ((((size_t)f * g.tiles_y + ty) * g.tiles_x + tx) << 6) + lane;
        const int y = ty * kTile + lane;
        const int valid_px = min(kTile, g.width - tx * kTile); // in-image pixels of this word (multiple of 8)
        if (y >= g.height) {
            strong[w] = 0;
            conn[w] = 0;
        } else if (valid_px < kTile) {
            uint8_t *sb = (uint8_t *)(strong + w), *cb = (uint8_t *)(conn + w);
            for (int b = valid_px / 8; b < 8; b++) sb[b] = cb[b] = 0;
        }
        return;
    }
    const size_t i0 = ((size_t)(wave - pad_waves) * 64 + lane) * 4;
    for (int j = 0; j < 4; j++)
        if (i0 + j < n_sched) sched[i0 + j] = 0u;
    if (wave == pad_waves && lane < n_flags) flags[lane] = 0u;
}

hipError_t launch_hyst_prepare(uint64_t *strong, uint64_t *conn, const HystGeom &g, bool zero_pad, unsigned *sched,
                               unsigned *flags, hipStream_t stream, int n_lanes)
{
    if (n_lanes < 1 || n_lanes > 16) return hipErrorInvalidValue;
    const bool pad = zero_pad && (g.height % kTile != 0 || g.width % kTile != 0);
    const long long pad_waves = pad ? (long long)(g.tiles_x + g.tiles_y) * g.n_frames : 0;
    // n_lanes independent propagations over disjoint frame ranges: their scheduling words lie back to back
    // (hyst_sched_words() of each range: 5 words per tile + 4 per frame + 4) and so do their flag pairs
    const size_t n_sched = 5 * (size_t)g.tiles() + 4 * (size_t)n_lanes + 4 * (size_t)g.n_frames;
    const long long waves = pad_waves + (long long)((n_sched + 255) / 256);
    if (pad_waves > 0x3fffffffLL || n_sched > 0xffffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hyst_prepare_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, strong, conn, g,
                       (int)pad_waves, sched, (unsigned)n_sched, flags, 2 * n_lanes);
    return hipGetLastError();
}

// s16 edge map (values 0 / 255) -> u8: the low byte of every pixel.  16 pixels per thread where alignment
// allows (two 16-byte loads, one 16-byte store), scalar head and tail otherwise.
__global__ __launch_bounds__(256) void edges_to_u8_kernel(const int16_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          size_t n)
{
    const size_t groups = n / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15u) == 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        if (aligned) {
            const uint4 a = *reinterpret_cast<const uint4 *>(in + g * 16);
            const uint4 b = *reinterpret_cast<const uint4 *>(in + g * 16 + 8);
            // bytes 0 and 2 of each dword are the low bytes of its two pixels
            uint4 r;
            r.x = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u);
            r.y = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
            r.z = __builtin_amdgcn_perm(b.y, b.x, 0x06040200u);
            r.w = __builtin_amdgcn_perm(b.w, b.z, 0x06040200u);
            *reinterpret_cast<uint4 *>(out + g * 16) = r;
        } else {
            for (int k = 0; k < 16; k++) out[g * 16 + k] = (uint8_t)in[g * 16 + k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < n % 16) out[groups * 16 + threadIdx.x] = (uint8_t)in[groups * 16 + threadIdx.x];
}

hipError_t launch_edges_to_u8(const int16_t *edges, uint8_t *out, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(edges_to_u8_kernel, dim3(grid_for(n / 16 + 1, 256)), dim3(256), 0, stream, edges, out, n);
    return hipGetLastError();
}

// s16 edge map -> 1 bit per pixel (pixel != 0), rows packed MSB-first (pixel 0 of a row is bit 7 of the row's byte 0,
// as in PBM "P4" and numpy.packbits), every row padded to whole bytes: [n_frames][height][(width + 7) / 8] bytes.
// One byte -- 8 pixels, one 16-byte load where the row pitch allows it -- per thread and step.
__global__ __launch_bounds__(256) void edges_to_bits_kernel(const int16_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            int width, int row_bytes, size_t n_bytes, int vec_ok)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_bytes; b += stride) {
        const size_t row = b / (size_t)row_bytes;
        const int xb = (int)(b - row * (size_t)row_bytes);
        const int16_t *px = in + row * (size_t)width + (size_t)xb * 8;
        unsigned byte = 0;
        if (vec_ok) { // width % 8 == 0 and a 16-byte aligned plane: every group of 8 pixels is one aligned uint4
            const uint4 v = *reinterpret_cast<const uint4 *>(px);
            const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++)
                byte |= (((d[k] & 0xffffu) != 0 ? 2u : 0u) | ((d[k] >> 16) != 0 ? 1u : 0u)) << (6 - 2 * k);
        } else {
            const int left = width - xb * 8; // pixels of this row from px on
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < left && px[k] != 0) byte |= 0x80u >> k;
        }
        out[b] = (uint8_t)byte;
    }
}

hipError_t launch_edges_to_bits(const int16_t *edges, uint8_t *bits, int height, int width, int n_frames,
                                hipStream_t stream)
{
    const int row_bytes = (width + 7) / 8;
    const size_t n_bytes = (size_t)n_frames * (size_t)height * (size_t)row_bytes;
    if (n_bytes == 0) return hipSuccess;
    const int vec_ok = (width % 8 == 0) && (((uintptr_t)edges & 15u) == 0);
    hipLaunchKernelGGL(edges_to_bits_kernel, dim3(grid_for(n_bytes, 256)), dim3(256), 0, stream, edges, bits, width,
                       row_bytes, n_bytes, vec_ok);
    return hipGetLastError();
}

// Publishes flags[0..1] and a sequence number in host-visible (pinned, mapped) memory: the host polls host[2].
__global__ void hyst_publish_kernel(const unsigned *__restrict__ flags, unsigned *host, unsigned seq)
{
    __hip_atomic_store(&host[0], flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[1], flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_hyst_publish(const unsigned *flags, unsigned *host_flags_dev, unsigned seq, hipStream_t stream)
{
    hipLaunchKernelGGL(hyst_publish_kernel, dim3(1), dim3(1), 0, stream, flags, host_flags_dev, seq);
    return hipGetLastError();
}

hipError_t launch_hyst_tail(uint64_t *strong, const uint64_t *conn, unsigned *sched, unsigned *last_change,
                            int first_iter, const HystGeom &g, hipStream_t stream, int16_t *edges, int edge_value)
{
    if (g.n_frames < 1 || first_iter < 1 || g.tiles_x * g.tiles_y > kTailTiles) return hipErrorInvalidValue;
    // one workgroup per frame; 16 waves while a frame has enough tiles to keep them busy
    const int tpf = g.tiles_x * g.tiles_y;
    const unsigned threads = tpf >= 256 ? 1024u : (tpf >= 64 ? 512u : 256u);
    hipLaunchKernelGGL(hyst_tail_kernel, dim3((unsigned)g.n_frames), dim3(threads), 0, stream, strong, conn, sched,
                       last_change, first_iter, g, edges, edge_value);
    return hipGetLastError();
}

hipError_t launch_hyst_propagate(uint64_t *strong, const uint64_t *conn, unsigned *stamp, unsigned *last_change,
                                 int iter, const HystGeom &g, hipStream_t stream, int16_t *edges, int edge_value,
                                 bool to_frame_queues)
{
    unsigned blocks = (unsigned)((g.tiles() + 4 * kSweep0Tiles - 1) / (4 * kSweep0Tiles)); // sweep 0: all tiles
    if (iter > 0) { // queue walkers: at most 16 waves per CU, and no more than a wave per 4 tiles (a single frame's
                    // 2040 tiles: 128 workgroups instead of 1024, whose dispatch alone took 4 of a sweep's 9 us)
        const unsigned want = (unsigned)((g.tiles() + 15) / 16);
        blocks = want < 8u ? 8u : (want > 1024u ? 1024u : want);
    }
    hipLaunchKernelGGL(hyst_propagate_kernel, dim3(blocks), dim3(256), 0, stream, strong, conn, stamp, last_change,
                       iter, g, edges, edge_value, to_frame_queues ? 1 : 0);
    return hipGetLastError();
}
// Row-major finalize: a wave writes 512 consecutive pixels of ONE row (1 KB contiguous, four rows per
// thread) and gathers its 64 plane bytes from eight tiles (8 x 8 contiguous bytes, L2 resident).
__global__ __launch_bounds__(256) void hyst_finalize_rows_kernel(int16_t *__restrict__ cand,
                                                                 const uint8_t *__restrict__ strong, HystGeom g,
                                                                 int edge_value, int groups_per_row)
{
    // grid: x = 512-pixel chunks of a row (one wave each, 4 per workgroup), y = groups of 4 rows, z = frame
    const uint32_t ev = (uint32_t)(uint16_t)edge_value;
    const int chunk = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int gx = chunk * 64 + (int)(threadIdx.x & 63); // 8-pixel group index within the row
    if (gx >= groups_per_row) return;
    const int f = (int)blockIdx.z;
    const int x0 = gx * 8;
    const size_t tile_col = ((size_t)f * g.tiles_y) * g.tiles_x + (gx >> 3);
    unsigned b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = (int)blockIdx.y * 4 + k;
        b[k] = 0;
        if (y < g.height)
            b[k] = strong[((tile_col + (size_t)(y >> 6) * g.tiles_x) << 9) + (size_t)(y & 63) * 8 + (gx & 7)];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = (int)blockIdx.y * 4 + k;
        if (y >= g.height) break;
        const uint32_t t = b[k] | (b[k] << 15);
        uint4 v;
        v.x = __umul24(t & 0x00010001u, ev);
        v.y = __umul24((t >> 2) & 0x00010001u, ev);
        v.z = __umul24((t >> 4) & 0x00010001u, ev);
        v.w = __umul24((t >> 6) & 0x00010001u, ev);
        // (non-temporal stores measured the same 0.218 ms)
        __builtin_memcpy(cand + ((size_t)f * g.height + y) * g.width + x0, &v, 16);
    }
}

static int finalize_mode = 0; // A/B switch: 0 = rows kernel, 1 = 8-row patch kernel
void hyst_set_finalize_mode(int m) { finalize_mode = m; }

hipError_t launch_hyst_finalize(int16_t *cand, const uint64_t *strong, const HystGeom &g, int edge_value,
                                hipStream_t stream)
{
    size_t total = (size_t)g.n_frames * g.height * g.width;
    if (g.width % 8 == 0 && finalize_mode != 1) {
        const int groups = g.width / 8;
        hipLaunchKernelGGL(hyst_finalize_rows_kernel, dim3((groups + 255) / 256, (g.height + 3) / 4, g.n_frames),
                           dim3(256), 0, stream, cand, (const uint8_t *)strong, g, edge_value, groups);
    } else if (g.width % 8 == 0)
        hipLaunchKernelGGL(hyst_finalize8_kernel, dim3((g.tiles_x + 1) / 2, g.tiles_y, g.n_frames), dim3(256), 0, stream, cand,
                           (const uint8_t *)strong, g, edge_value);
    else
        hipLaunchKernelGGL(hyst_finalize_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, cand, strong, g,
                           edge_value);
    return hipGetLastError();
}
hipError_t launch_fep_classify(const int16_t *cand, const uint8_t *visited, uint64_t *strong, uint64_t *conn,
                               const HystGeom &g, int start, int min_val, hipStream_t stream)
{
    size_t n_words = (size_t)g.tiles_y * kTile * g.tiles_x;
    hipLaunchKernelGGL(fep_classify_kernel, dim3(grid_for(n_words * 64, 256)), dim3(256), 0, stream, cand, visited,
                       strong, conn, g, start, min_val);
    return hipGetLastError();
}
hipError_t launch_fep_finalize(int16_t *cand, uint8_t *visited, const uint64_t *strong, const HystGeom &g, int start,
                               int min_val, hipStream_t stream)
{
    size_t total = (size_t)g.height * g.width;
    hipLaunchKernelGGL(fep_finalize_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, cand, visited, strong, g,
                       start, min_val);
    return hipGetLastError();
}

// ================================================================================================
// Self-test: the device's 8-bit-domain magnitude and angle rules over every (gx,gy) in [-lim,lim]^2
// ================================================================================================
__global__ __launch_bounds__(256) void selftest_mag_angle_kernel(int lim, int16_t *__restrict__ mags,
                                                                 uint8_t *__restrict__ bins)
{
    const int side = 2 * lim + 1;
    const size_t total = (size_t)side * side;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int gy = (int)(i / side) - lim, gx = (int)(i % side) - lim;
        mags[i] = (int16_t)magnitude_d8(gx, gy);
        bins[i] = (uint8_t)angle_bin_d8(gx, gy);
    }
}
hipError_t launch_selftest_mag_angle(int lim, int16_t *mags, uint8_t *bins, hipStream_t stream)
{
    size_t total = (size_t)(2 * lim + 1) * (2 * lim + 1);
    hipLaunchKernelGGL(selftest_mag_angle_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, lim, mags, bins);
    return hipGetLastError();
}


// ---- measurement aid: a plain device copy -----------------------------------------------------------------------
// What does a stream that reads N bytes and writes N bytes reach on THIS device?  The roofline of the Sobel+NMS pass
// is quoted against the 8 TB/s spec; boxes differ by up to 14 % in what a copy reaches (tools/probe_march_pattern.hip),
// so bench.py times this kernel beside the pass it grades.  Workgroup b moves the 4 KB chunk b (one 16-byte load and one
// 16-byte store per thread): the fastest of the copy shapes tried (grid-stride loops, 4-64 KB chunks, non-temporal).
__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) out[i] = in[i];
}
hipError_t launch_probe_copy(const void *src, void *dst, size_t nbytes, hipStream_t stream, const LaunchEvents &ev)
{
    const size_t n16 = nbytes / 16;
    if (n16 == 0) return hipSuccess;
    const size_t blocks = (n16 + 255) / 256;
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
    if (ev.start && ev.stop)
        hipExtLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ev.start, ev.stop, 0,
                              (const uint4 *)src, (uint4 *)dst, n16);
    else
        hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint4 *)src,
                           (uint4 *)dst, n16);
    return hipGetLastError();
}

} // namespace canny
