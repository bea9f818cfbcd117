// canny_sobel_nms_march.hip -- fused Sobel + non-maximal suppression, wave-marching, LDS-free.
//
// The roofline-graded pass: s16 smoothed plane in, s16 suppressed magnitude out, 4 algorithmic bytes
// per pixel; gradient, magnitude and angle never leave registers.
//
// One WAVE owns a strip of 62*8 = 496 output columns (lanes 0 and 63 are halo lanes) and marches down
// a segment of rows.  Each lane holds 8 adjacent pixels of the current row as four packed s16 pairs,
// loaded with one 16-byte global_load and stored with one 16-byte global_store:
//   horizontal step (packed 16-bit math, v_pk_*):  d = s[c+1]-s[c-1],  t = s[c-1]+2s[c]+s[c+1]
//       neighbours across the lane boundary come from wave-wide DPP shifts (no LDS, no barrier);
//   vertical step (packed):  gx = d[y-1]+2d[y]+d[y+1],  gy = t[y+1]-t[y-1]
//   per pixel (f32, exact: every value is an integer < 2^24):
//       n = gx^2+gy^2,  mag = trunc(sqrt(n+0.5)),  P = gx*gy,  q = (gx^2-gy^2)/2
//       bin 0 iff |P| <= q, bin 90 iff |P| < -q, else 45 (P > 0) / 135   [same rule as angle_bin_d8]
//   NMS one row later, when the magnitudes of the row below exist: strict max against the two
//       neighbours of the bin; neighbours outside the image carry magnitude -1 (= "skip").
// Three rows of d/t and of magnitudes rotate through registers (the row loop is unrolled by 6 so
// that every rotation is a compile-time renaming).
//
// Border conventions of the reference (src/utils.cpp:114-186, 248-308), all under wave-uniform branches:
//   gx: column clamp  -> zero-filled neighbours plus a +-s fix-up at columns 0 and W-1; rows dropped
//       (virtual rows are zero, which is exactly "dropped");
//   gy: columns dropped (zero fill), row clamp -> t[-1]:=t[0], t[H]:=t[H-1];
//   NMS: out-of-image neighbours skipped -> magnitude -1.
// Precondition: smoothed values in [0,255] (what gaussian() produces), so |gx|,|gy| <= 1020.
#include "canny_kernels.h"

#include <type_traits>

namespace canny {

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_v(uint32_t u) { return __builtin_bit_cast(s16x2, u); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_v(a) + as_v(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_v(a) - as_v(b)); }
// a*2 + b per half
__device__ __forceinline__ uint32_t pk_mad2(uint32_t a, uint32_t b)
{
    const s16x2 two = {2, 2};
    return as_u(as_v(a) * two + as_v(b));
}

// value held by the lane to the left / right (0 at the wave's ends)
__device__ __forceinline__ uint32_t from_left(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t from_right(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

constexpr int SNM_PX = 8;                 // pixels per lane
constexpr int SNM_SW = 62 * SNM_PX;       // output columns per strip
constexpr int SNM_WPB = 4;                // waves per workgroup (independent of each other)

template <int N>
using IC = std::integral_constant<int, N>;

} // namespace

__global__ __launch_bounds__(SNM_WPB * 64) void sobel_nms_march_kernel(const int16_t *__restrict__ in,
                                                                       int16_t *__restrict__ out, int H, int W,
                                                                       int n_strips, int n_segs, int seg_rows,
                                                                       int total_waves)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * SNM_WPB + (threadIdx.x >> 6);
    if (wave >= total_waves) return;
    const int s = wave % n_strips;
    const int g = (wave / n_strips) % n_segs;
    const int f = wave / (n_strips * n_segs);
    const int ybeg = g * seg_rows;
    const int yend = min(H, ybeg + seg_rows);
    const int x0 = s * SNM_SW + (lane - 1) * SNM_PX; // column of this lane's pixel 0
    const bool full8 = x0 >= 0 && x0 + 7 < W;
    const bool owner = lane >= 1 && lane <= 62 && x0 < W;
    const bool first_strip = (s == 0);                                // holds column 0 (lane 1, pixel 0)
    const bool last_strip = ((s + 1) * SNM_SW + SNM_PX >= W);          // holds column W-1 and/or columns >= W
    const int16_t *fin = in + (size_t)f * H * W;
    int16_t *fout = out + (size_t)f * H * W;

    // lane-varying border masks (only consulted in the first / last strip)
    const uint32_t fix_l = (x0 == 0) ? 0x0000ffffu : 0u; // column 0 = low half of pair 0
    uint32_t fix_r[4];
    unsigned oob = 0; // bit e set: column x0+e is outside the image
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t m = 0;
        if (x0 + 2 * i == W - 1) m |= 0x0000ffffu;
        if (x0 + 2 * i + 1 == W - 1) m |= 0xffff0000u;
        fix_r[i] = m;
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (x0 + e < 0 || x0 + e >= W) oob |= 1u << e;

    auto load_row = [&](int r, uint32_t (&p)[4]) {
        p[0] = p[1] = p[2] = p[3] = 0u;
        if (r < 0 || r >= H) return; // wave-uniform: virtual rows are zero
        const int16_t *src = fin + (size_t)r * W + x0;
        if (full8) {
            uint4 v;
            __builtin_memcpy(&v, src, 16);
            p[0] = v.x;
            p[1] = v.y;
            p[2] = v.z;
            p[3] = v.w;
        } else if (x0 + 7 >= 0 && x0 < W) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                int x = x0 + e;
                if (x >= 0 && x < W) p[e >> 1] |= (uint32_t)(uint16_t)src[e] << (16 * (e & 1));
            }
        }
    };

    // rotating state (all indices are compile-time after unrolling)
    uint32_t d[3][4], t[3][4]; // horizontal difference / smooth of three consecutive rows
    int M[3][10];              // magnitudes of three consecutive rows; [0] and [9] are the neighbours' edge pixels
    float cP[2][8], cQ[2][8];  // bin discriminants of the row awaiting NMS
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int i = 0; i < 4; i++) d[a][i] = t[a][i] = 0u;
#pragma unroll
        for (int e = 0; e < 10; e++) M[a][e] = -1;
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int e = 0; e < 8; e++) cP[a][e] = cQ[a][e] = 0.0f;

    const int rfirst = ybeg - 2, rlast = yend + 1;
    uint32_t pa[4], pb[4], pc[4]; // software prefetch: rows r, r+1, r+2
    load_row(rfirst, pa);
    load_row(rfirst + 1, pb);

    // One input row: PH = (r - rfirst) mod 6 selects the register roles.
    auto step = [&](auto ph, int r, const uint32_t (&p)[4]) {
        constexpr int PH = decltype(ph)::value;
        constexpr int k2 = PH % 3, k1 = (PH + 2) % 3, k0 = (PH + 1) % 3; // rows r, r-1, r-2 in d/t
        constexpr int m2 = PH % 3, m1 = (PH + 2) % 3, m0 = (PH + 1) % 3; // rows r-1, r-2, r-3 in M
        constexpr int cn = PH % 2, co = (PH + 1) % 2;                     // bins of row r-1 (new) / r-2 (old)

        // ---- horizontal step for row r ---------------------------------------------------------
        {
            const uint32_t lp = from_left(p[3]), rp = from_right(p[0]);
            uint32_t sh[5]; // sh[i] = (pixel 2i-1, pixel 2i)
            sh[0] = __builtin_amdgcn_alignbit(p[0], lp, 16);
            sh[1] = __builtin_amdgcn_alignbit(p[1], p[0], 16);
            sh[2] = __builtin_amdgcn_alignbit(p[2], p[1], 16);
            sh[3] = __builtin_amdgcn_alignbit(p[3], p[2], 16);
            sh[4] = __builtin_amdgcn_alignbit(rp, p[3], 16);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                d[k2][i] = pk_sub(sh[i + 1], sh[i]);
                t[k2][i] = pk_mad2(p[i], pk_add(sh[i], sh[i + 1]));
            }
            if (first_strip) d[k2][0] = pk_sub(d[k2][0], p[0] & fix_l); // clamp at column 0
            if (last_strip) {                                            // clamp at column W-1
#pragma unroll
                for (int i = 0; i < 4; i++) d[k2][i] = pk_add(d[k2][i], p[i] & fix_r[i]);
            }
        }

        // ---- gradient, magnitude and bin discriminants for row y1 = r-1 ------------------------
        const int y1 = r - 1;
        if (y1 >= 0 && y1 < H) {
            uint32_t tu[4], td[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                tu[i] = (y1 == 0) ? t[k1][i] : t[k0][i];      // row clamp at the top
                td[i] = (y1 == H - 1) ? t[k1][i] : t[k2][i];  // ... and at the bottom
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t gx = pk_add(pk_mad2(d[k1][i], d[k0][i]), d[k2][i]);
                const uint32_t gy = pk_sub(td[i], tu[i]);
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const int e = 2 * i + hf;
                    const float fx = hf ? (float)((int)gx >> 16) : (float)(int)(short)(gx & 0xffffu);
                    const float fy = hf ? (float)((int)gy >> 16) : (float)(int)(short)(gy & 0xffffu);
                    const float A = __fmul_rn(fx, fx);
                    const float n = __fmaf_rn(fy, fy, A);
                    M[m2][e + 1] = (int)__builtin_amdgcn_sqrtf(__fadd_rn(n, 0.5f));
                    cP[cn][e] = __fmul_rn(fx, fy);
                    cQ[cn][e] = __fmaf_rn(n, -0.5f, A);
                }
            }
            if (first_strip || last_strip) { // columns outside the image never win a comparison
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (oob & (1u << e)) M[m2][e + 1] = -1;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) M[m2][e + 1] = -1; // rows outside the image are skipped by NMS
        }
        // edge pixels of the neighbouring lanes (lanes 0 and 63 get 0 here; they are halo lanes whose
        // own NMS results are never stored, and their neighbours only read M[8] / M[1] from them)
        M[m2][0] = (int)from_left((uint32_t)M[m2][8]);
        M[m2][9] = (int)from_right((uint32_t)M[m2][1]);

        // ---- NMS for row y2 = r-2 ---------------------------------------------------------------
        const int y2 = r - 2;
        if (y2 >= ybeg && y2 < yend) {
            int res[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int c = e + 1;
                const int mc = M[m1][c];
                const int n0 = max(M[m1][c - 1], M[m1][c + 1]);
                const int n90 = max(M[m0][c], M[m2][c]);
                const int n45 = max(M[m0][c + 1], M[m2][c - 1]);  // up-right, down-left
                const int n135 = max(M[m0][c - 1], M[m2][c + 1]); // up-left, down-right
                const float aP = __builtin_fabsf(cP[co][e]), q = cQ[co][e];
                const int ndiag = (cP[co][e] > 0.0f) ? n45 : n135;
                const int nv = (aP < -q) ? n90 : ndiag;
                const int nsel = (aP <= q) ? n0 : nv;
                res[e] = (mc > nsel) ? mc : 0;
            }
            if (owner) {
                int16_t *dst = fout + (size_t)y2 * W + x0;
                if (full8) {
                    uint4 v;
                    v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                    v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                    v.z = (uint32_t)res[4] | ((uint32_t)res[5] << 16);
                    v.w = (uint32_t)res[6] | ((uint32_t)res[7] << 16);
                    __builtin_memcpy(dst, &v, 16);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (x0 + e < W) dst[e] = (int16_t)res[e];
                }
            }
        }
    };

    for (int r = rfirst; r <= rlast; r += 6) {
        // the prefetch registers rotate with period 3, the compute state with period 6
        load_row(r + 2, pc);
        step(IC<0>{}, r, pa);
        if (r + 1 > rlast) break;
        load_row(r + 3, pa);
        step(IC<1>{}, r + 1, pb);
        if (r + 2 > rlast) break;
        load_row(r + 4, pb);
        step(IC<2>{}, r + 2, pc);
        if (r + 3 > rlast) break;
        load_row(r + 5, pc);
        step(IC<3>{}, r + 3, pa);
        if (r + 4 > rlast) break;
        load_row(r + 6, pa);
        step(IC<4>{}, r + 4, pb);
        if (r + 5 > rlast) break;
        load_row(r + 7, pb);
        step(IC<5>{}, r + 5, pc);
    }
}

bool sobel_nms_march_supported(int height, int width) { return height >= 2 && width >= 2; }

hipError_t launch_sobel_nms_march(const int16_t *smoothed, int16_t *out, int height, int width, int n_frames,
                                  hipStream_t stream)
{
    int n_strips = (width + SNM_SW - 1) / SNM_SW;
    int seg = 256;
    while (seg > 32 && (long long)n_frames * n_strips * ((height + seg - 1) / seg) < 16384) seg >>= 1;
    int n_segs = (height + seg - 1) / seg;
    long long waves = (long long)n_frames * n_strips * n_segs;
    if (waves > 0x7fffffffLL) return hipErrorInvalidValue;
    unsigned blocks = (unsigned)((waves + SNM_WPB - 1) / SNM_WPB);
    hipLaunchKernelGGL(sobel_nms_march_kernel, dim3(blocks), dim3(SNM_WPB * 64), 0, stream, smoothed, out, height,
                       width, n_strips, n_segs, seg, (int)waves);
    return hipGetLastError();
}

} // namespace canny
