// canny_gaussian_march.hip -- separable Gaussian, wave-marching path (window <= 17).
//
// Two kernels with the same decomposition (one WAVE owns a strip of (64-2*HL)*4 output columns and marches
// down a segment of rows, row and column pass in one kernel, the f32 intermediate never leaves the CU):
//   * gauss_sym_kernel  (default, see the comment above gauss_sym_strip): needs bit-symmetric taps; each
//     rounded product serves two outputs, the row pass looks its products up in an LDS table and -- since
//     round 3, "SYS" -- passes the running sums from lane to lane with DPP adds (the earlier form, which
//     fetches the neighbours' products instead, is still there for A/B), the column pass keeps 2C+1 running
//     sums in registers;
//   * gauss_march_kernel (below; fallback for asymmetric taps and A/B): LDS row buffer + LDS column ring.
//
// gauss_march_kernel: the waves of a workgroup are independent (no __syncthreads anywhere).  Per input row:
//   1. each lane loads 4 u8 pixels (one dword), converts them once and publishes the 4 floats in a
//      per-wave LDS row buffer; neighbours' pixels come back as aligned ds_read_b128 (halo exchange
//      through LDS, wave-scope fences only);
//   2. row pass: 4 outputs per lane, taps in ascending order, separately rounded mul/add, exact
//      division by the per-column weight -> written to the lane's private column ring in LDS;
//   3. once 2C+1 rows are in the ring the column pass produces one output row: taps in ascending
//      row order out of the ring, exact division by the per-row weight, truncate, one 8-byte store.
//
// Bit-exactness (reference src/utils.cpp:37-64):
//   * Out-of-image pixels/rows enter the sums as +0.0f products, which leaves every partial sum
//     bit-identical to the reference's "skip the tap" (x + 0 == x exactly for x >= 0); only the weight
//     has to be the sum over the in-image taps, accumulated in the reference's ascending order.
//   * a / b is computed as the tail of the hardware's own IEEE-754 division expansion
//     (q0 = a*y, two residual corrections with fma) with y = RN(1/b) obtained once from a true IEEE
//     division; a in [0,256], b in (0,1.01] so no scaling is needed.  canny_hip_selftest_div checks it
//     against __fdiv_rn over every float in [0,256] for the weights of a sigma sweep.
// HBM traffic: 1 B/px in + 2 B/px out; the f32 intermediate never leaves the CU.
//
// COL_EDGE / ROW_EDGE instantiations: waves whose strip touches column 0 / W-1 or whose segment touches
// row 0 / H-1 carry the border logic; all others run straight-line code with unconditional loads and a
// single wave-uniform weight.  The wave index goes through readfirstlane so that rows and ring slots live
// in SGPRs.
#include "canny_kernels.h"

#include <cstring>
#include <type_traits>

// The file is compiled FOUR times (Makefile: -DCANNY_GAUSS_PART=0..3 -> canny_gaussian_march_p<k>.o) so that the eight
// window instantiations, which dominate the library's build time, compile in parallel: part 0 holds the host-side
// entry points, the switches and windows 3..9 (half-windows 1..4), part 1 half-windows 5 and 6, part 2 half-window 7,
// part 3 half-window 8.  Without the macro (tools/gauss_isa.sh) everything is one translation unit.
#ifndef CANNY_GAUSS_PART
#define CANNY_GAUSS_PART -1
#endif
#define CANNY_GAUSS_HAS_HOST (CANNY_GAUSS_PART <= 0)

namespace canny {

// A/B switches shared by the parts (defined in part 0, set through canny_hip_ctx_set_option)
extern bool g_gauss_fma_div_enabled;
extern int g_gauss_march_variant;
extern int g_gauss_seg_target;
// launchers of the half-windows that live in parts 1..3
hipError_t launch_gauss_march_part1(int center, const uint8_t *img, void *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream, int out_u8);
hipError_t launch_gauss_march_part2(int center, const uint8_t *img, void *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream, int out_u8);
hipError_t launch_gauss_march_part3(int center, const uint8_t *img, void *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream, int out_u8);

namespace {

template <int V>
using IC = std::integral_constant<int, V>;

template <int C>
struct MarchCfg {
    static constexpr int HL = (C + 3) / 4;           // halo lanes per side (4 px each)
    static constexpr int RING = 2 * C + 1;           // rows the column pass needs (= window)
    static constexpr int REGROWS = 3;                // the newest rows stay in registers (loop is unrolled by 3)
    static constexpr int LROWS = RING - REGROWS;     // older rows live in the LDS ring
    static constexpr int WIN = 4 + 8 * HL;           // floats a lane reads back per row
    static constexpr int SW = (64 - 2 * HL) * 4;     // output columns per strip
    static constexpr int ROWBUF = (64 + 2 * HL) * 4; // floats
    static constexpr int WAVE_FLOATS = ROWBUF + (LROWS > 0 ? LROWS : 1) * 256;
    static constexpr int WPB = (C <= 6) ? 4 : 2;     // waves per workgroup (LDS budget)
};
// LDS per wave = 1.1 KB row buffer + LROWS KB: 9.1 KB at window 11 -> 16 waves per CU (4 per SIMD).  With
// the whole ring in LDS (12.4 KB) only 3 waves per SIMD fit, and this kernel is bound by per-wave issue
// latency, not by the VALU pipe (55 % busy at 3 waves), so resident waves are what buys speed.

// Orders this wave's LDS accesses in program order for the COMPILER (the hardware already executes one
// wave's DS instructions in order).  A wavefront-scope fence emits no instruction and, unlike
// __builtin_amdgcn_wave_barrier(), does not stop the scheduler from overlapping ALU work across it.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Correctly rounded a / b given y = RN(1/b): the refinement tail of the IEEE division sequence.
__device__ __forceinline__ float div_by(float a, float b, float y)
{
    float q = __fmul_rn(a, y);
    float r = __fmaf_rn(-q, b, a);
    q = __fmaf_rn(r, y, q);
    r = __fmaf_rn(-q, b, a);
    return __fmaf_rn(r, y, q);
}

struct GaussJob {
    const uint8_t *fimg;
    int16_t *fout;     // s16 plane ...
    uint8_t *fout8;    // ... or, in the OUT_U8 kernels, the u8 plane (same values: the quotients lie in [0,255])
    int H, W, ybeg, yend, x0, lane;
};

// FMA_DIV (interior waves only, where every divisor is the full-window weight S): a / S as the single
// instruction fma(a, fma_c, a).  The host enables it only for the (S, c) pairs of kFmaDivTable, each of
// which canny_hip_selftest_div_fma has shown to equal the IEEE quotient for every float a in [0, 256].
template <int C, bool COL_EDGE, bool ROW_EDGE, bool FMA_DIV = false>
__device__ __forceinline__ void gauss_march_strip(const GaussJob &jb, const GaussTaps &t, float *rowbuf, float *colring,
                                                  float fma_c = 0.0f)
{
    static_assert(!FMA_DIV || (!COL_EDGE && !ROW_EDGE), "FMA_DIV needs a single wave-uniform divisor");
    using K = MarchCfg<C>;
    constexpr int HL = K::HL, RING = K::RING, WIN = K::WIN;
    const int H = jb.H, W = jb.W, x0 = jb.x0, ybeg = jb.ybeg, yend = jb.yend, lane = jb.lane;
    const bool owner = lane >= HL && lane < 64 - HL && x0 < W;
    const bool full4 = x0 >= 0 && x0 + 3 < W;

    // weights: full window (wave-uniform) and, at the column borders, this lane's four own weights
    float cnt_full = t.tap[0];
#pragma unroll
    for (int k = 1; k < RING; k++) cnt_full = __fadd_rn(cnt_full, t.tap[k]);
    const float inv_full = __fdiv_rn(1.0f, cnt_full);
    float cnt_h[4], inv_h[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        cnt_h[j] = cnt_full;
        inv_h[j] = inv_full;
    }
    if (COL_EDGE) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = x0 + j;
            float c = 0.0f;
#pragma unroll
            for (int k = 0; k < RING; k++) {
                int xx = x + k - C;
                if (xx >= 0 && xx < W) c = __fadd_rn(c, t.tap[k]);
            }
            cnt_h[j] = (x >= 0 && x < W) ? c : 1.0f;
            inv_h[j] = __fdiv_rn(1.0f, cnt_h[j]);
        }
    }

    auto load_row = [&](int r) -> uint32_t {
        if (ROW_EDGE && (r < 0 || r >= H)) return 0u; // wave-uniform
        const uint8_t *p = jb.fimg + (size_t)r * W;
        uint32_t v = 0u;
        if (!COL_EDGE || full4) {
            __builtin_memcpy(&v, p + x0, 4);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int x = x0 + j;
                if (x >= 0 && x < W) v |= (uint32_t)p[x] << (8 * j);
            }
        }
        return v;
    };

    constexpr int LROWS = K::LROWS;
    int wslot = 0; // LDS ring slot the next evicted row goes to (row i of this segment -> slot i mod LROWS)
    int oslot = 0; // LDS ring slot holding tap 0 of the next output row
    float4 recent[3]; // row-pass results of rows r, r-1, r-2; slot = (row - rfirst) mod 3
    recent[0] = recent[1] = recent[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int rfirst0 = ybeg - C;

    auto step = [&](auto ph, int r, uint32_t cur) {
        constexpr int PH = decltype(ph)::value; // (r - rfirst) mod 3
        // ---- the row that leaves the register window (r-3) moves to the LDS ring ----------------------
        if (LROWS > 0 && r - rfirst0 >= 3) {
            *reinterpret_cast<float4 *>(colring + wslot * 256) = recent[PH];
            wslot = (wslot + 1 == LROWS) ? 0 : wslot + 1;
        }
        // ---- row pass of input row r -> recent[PH] --------------------------------------------------
        float4 tmp = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!ROW_EDGE || (r >= 0 && r < H)) {
            float4 own = make_float4((float)(cur & 0xffu), (float)((cur >> 8) & 0xffu), (float)((cur >> 16) & 0xffu),
                                     (float)(cur >> 24));
            *reinterpret_cast<float4 *>(rowbuf + (lane + HL) * 4) = own;
            wave_lds_fence();
            float wv[WIN];
#pragma unroll
            for (int q = 0; q < WIN / 4; q++) {
                // Not volatile on purpose.  The compiler narrows these reads to the elements it needs
                // (ds_read2_b32 at a 16-byte lane stride: 4-way bank conflicts, half of this kernel's LDS
                // cycles), but forcing five full ds_read_b128 measured 20 % SLOWER (1.16 vs 0.93 ms per
                // 64 x 4K): the kernel is VALU bound and the wider reads only add LDS traffic and waits.
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rowbuf + lane * 4 + q * 4);
                wv[4 * q + 0] = v[0];
                wv[4 * q + 1] = v[1];
                wv[4 * q + 2] = v[2];
                wv[4 * q + 3] = v[3];
            }
            wave_lds_fence();
            float res[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float acc = __fmul_rn(wv[4 * HL + j - C], t.tap[0]);
#pragma unroll
                for (int k = 1; k < RING; k++) acc = __fadd_rn(acc, __fmul_rn(wv[4 * HL + j - C + k], t.tap[k]));
                res[j] = FMA_DIV ? __fmaf_rn(acc, fma_c, acc) : div_by(acc, cnt_h[j], inv_h[j]);
            }
            tmp = make_float4(res[0], res[1], res[2], res[3]);
        }
        recent[PH] = tmp;

        // ---- column pass of output row y = r - C (its last tap just arrived) -------------------------
        // taps 0 .. LROWS-1 come from the LDS ring (rows r-2C .. r-3), the last three from registers
        const int y = r - C;
        if (y >= ybeg && y < yend) {
            float cnt_v = cnt_full, inv_v = inv_full;
            if (ROW_EDGE && (y < C || y + C >= H)) { // top/bottom border rows renormalise (wave-uniform)
                cnt_v = 0.0f;
                for (int k = 0; k < RING; k++) {
                    int yy = y + k - C;
                    if (yy >= 0 && yy < H) cnt_v = __fadd_rn(cnt_v, t.tap[k]);
                }
                inv_v = __fdiv_rn(1.0f, cnt_v);
            }
            float acc[4];
#pragma unroll
            for (int k = 0; k < RING; k++) {
                float4 v;
                if (k < LROWS) {
                    int slot = oslot + k;
                    slot = slot >= LROWS ? slot - LROWS : slot;
                    v = *reinterpret_cast<const float4 *>(colring + slot * 256);
                } else {
                    v = recent[(PH + 1 + (k - LROWS)) % 3]; // k = LROWS, +1, +2  ->  rows r-2, r-1, r
                }
                if (k == 0) {
                    acc[0] = __fmul_rn(v.x, t.tap[0]);
                    acc[1] = __fmul_rn(v.y, t.tap[0]);
                    acc[2] = __fmul_rn(v.z, t.tap[0]);
                    acc[3] = __fmul_rn(v.w, t.tap[0]);
                } else {
                    acc[0] = __fadd_rn(acc[0], __fmul_rn(v.x, t.tap[k]));
                    acc[1] = __fadd_rn(acc[1], __fmul_rn(v.y, t.tap[k]));
                    acc[2] = __fadd_rn(acc[2], __fmul_rn(v.z, t.tap[k]));
                    acc[3] = __fadd_rn(acc[3], __fmul_rn(v.w, t.tap[k]));
                }
            }
            if (owner) {
                // float -> short truncates toward zero (src/utils.cpp:62)
                auto quot = [&](float a) { return FMA_DIV ? __fmaf_rn(a, fma_c, a) : div_by(a, cnt_v, inv_v); };
                const int o0 = (int)quot(acc[0]), o1 = (int)quot(acc[1]);
                const int o2 = (int)quot(acc[2]), o3 = (int)quot(acc[3]);
                int16_t *dst = jb.fout + (size_t)y * W + x0;
                if (!COL_EDGE || full4) {
                    uint2 pk;
                    pk.x = (uint32_t)(uint16_t)o0 | ((uint32_t)(uint16_t)o1 << 16);
                    pk.y = (uint32_t)(uint16_t)o2 | ((uint32_t)(uint16_t)o3 << 16);
                    __builtin_memcpy(dst, &pk, 8);
                } else {
                    dst[0] = (int16_t)o0;
                    if (x0 + 1 < W) dst[1] = (int16_t)o1;
                    if (x0 + 2 < W) dst[2] = (int16_t)o2;
                }
            }
        }
        if (LROWS > 0 && y >= ybeg) oslot = (oslot + 1 == LROWS) ? 0 : oslot + 1;
    };

    // Rows ybeg-C .. yend-1+C, count rounded up to a multiple of 3 so that the three prefetch registers
    // rotate by renaming (copying a register whose load is still in flight would force vmcnt(0)).
    const int rfirst = ybeg - C;
    const int rlast = rfirst + 3 * ((yend - 1 + C - rfirst + 3) / 3) - 1;
    uint32_t pa = load_row(rfirst), pb = load_row(rfirst + 1), pc;
    for (int r = rfirst; r <= rlast; r += 3) {
        pc = load_row(r + 2);
        step(IC<0>{}, r, pa);
        pa = load_row(r + 3);
        step(IC<1>{}, r + 1, pb);
        pb = load_row(r + 4);
        step(IC<2>{}, r + 2, pc);
    }
}

// ---- symmetric-tap variant: shared products, register accumulators, no LDS ---------------------------
// The reference's taps are symmetric bit for bit (tap[C-a] == tap[C+a]: both come from the same
// exp(-(a*a)/...) expression, src/utils.cpp:77-95; the launcher re-checks the bit patterns).  The rounded
// product RN(v * tap) of one value v is therefore needed by TWO outputs, at distance +a and -a, and only
// C+1 instead of 2C+1 multiplies per value and pass are distinct.  The sums keep the reference's order
// (ascending tap index, every add rounded); what changes is who computes each product:
//   row pass     each lane multiplies its own 4 pixels by the C+1 distinct taps; outputs pick the products
//                of neighbouring pixels out of neighbouring lanes with DPP wave shifts that fold into the
//                v_add_f32 itself (no LDS row buffer, no extra instruction);
//   column pass  scatter form: the 2C+1 output rows a row-pass result contributes to are all open at once,
//                each with its partial sum in registers (slot = phase of the output's first row, the loop
//                is unrolled by 2C+1 so slots are compile-time).  Row r is tap k of output r+C-k, and the
//                rows before it have already been added, so the order is still ascending k.
// VALU work per pixel drops from 2*(2C+1) multiplies to 2*(C+1) (44 -> 24 at window 11; adds unchanged).
__device__ __forceinline__ float lane_shr1(float v) // lane L <- lane L-1, lane 0 <- 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_shl1(float v) // lane L <- lane L+1, lane 63 <- 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

template <class F, int... I>
__device__ __forceinline__ void for_each_phase(F &&f, std::integer_sequence<int, I...>)
{
    (f(IC<I>{}), ...);
}

// lut (may be null): (C+1) x 256 floats in LDS, lut[a*256 + v] = RN(float(v) * tap[C-a]), filled by the kernel.
// With it the row pass LOOKS UP its products (the input is 8 bits, so each tap has only 256 of them) instead
// of multiplying: ds_read_b32 issues on the LDS port beside the VALU work of other waves, and this kernel is
// bound by VALU issue.  The lookups of one instruction hit 64 pixels 4 columns apart -- nearly equal values,
// i.e. neighbouring or identical words: few bank conflicts on natural images.
// OUT_U8: the smoothed plane is stored as bytes (src/utils.cpp:62: (short)(sum/count) always lies in [0,255]), one
// dword store per lane and row instead of an 8-byte one.  (v_cvt_pk_u8_f32 would convert and pack in one
// instruction, but it rounds to nearest -- measured on the device: 41.9 M of the 1.13 G floats in [0,256] differ
// from the truncating cast -- so the conversion stays v_cvt_i32_f32.)
//
// SYS ("systolic" row pass, the default): the running SUM travels instead of the products.  A lane loads the four
// pixels x0+C .. x0+C+3 -- the LAST terms of its own four outputs x0 .. x0+3 --, starts the sums of the four outputs
// whose FIRST term is one of its pixels, and hands every unfinished sum to its right neighbour, which adds its own
// pixels' products one by one (the first of them in the same instruction that takes the sum over: a DPP add) and
// hands it on.  A sum visits floor((j+2C)/4)+1 lanes, so a row costs exactly 4*2C adds per lane of which
// sum_j floor((j+2C)/4) cross a lane (window 11: 40 adds, 10 of them DPP) where the product-fetching form above needs
// 44 with 19 DPP; every add has the reference's operands in the reference's order, whichever lane executes it.  The
// halo is on the left only (SysCfg::NL lanes), and the loads sit C bytes right of the (dword aligned) stores.
#ifndef GAUSS_ROTATE_LOOKUPS
#define GAUSS_ROTATE_LOOKUPS 1
#endif

template <int C>
struct SysCfg {
    static constexpr int NL = (3 + 2 * C) / 4;       // lanes a sum crosses at most = halo lanes (all on the left)
    static constexpr int SW = (64 - NL) * 4;         // output columns per strip
};

template <int C, bool COL_EDGE, bool ROW_EDGE, bool FMA_DIV, bool USE_LUT, bool OUT_U8 = false, bool SYS = false>
__device__ __forceinline__ void gauss_sym_strip(const GaussJob &jb, const GaussTaps &t, const float *lut, float *wts,
                                                float fma_c = 0.0f)
{
    static_assert(!FMA_DIV || (!COL_EDGE && !ROW_EDGE), "FMA_DIV needs a single wave-uniform divisor");
    constexpr int HL = MarchCfg<C>::HL, RING = 2 * C + 1;
    static_assert(HL <= 2, "products travel at most two lanes");
    const int H = jb.H, W = jb.W, x0 = jb.x0, ybeg = jb.ybeg, yend = jb.yend, lane = jb.lane;
    const int xin = x0 + (SYS ? C : 0);                  // first of the four columns this lane LOADS
    const bool owner = (SYS ? lane >= SysCfg<C>::NL : (lane >= HL && lane < 64 - HL)) && x0 < W;
    const bool full4 = x0 >= 0 && x0 + 3 < W;            // the four columns stored

    float T[C + 1]; // taps by distance from the centre
#pragma unroll
    for (int a = 0; a <= C; a++) {
        T[a] = t.tap[C - a];
        // The multiplying variant keeps the taps in VGPRs: a multiply with an SGPR operand is a slow-class instruction
        // (tools/valu_issue_bench.hip) and that variant does 2(C+1) of them per value: 1.180 -> 1.086 ms per
        // 128 x 4K together with the phased row pass.  The table variant multiplies only in the column pass and
        // needs the six registers more: pinning them there cost a wave per SIMD in round 2 (1.11 -> 1.25 ms), and in
        // the systolic form, where four or (with two prologue spills) all six fit, it buys nothing: 0.886 / 0.888
        // against 0.869 ms (profiles/r03/ab13_*.txt).
        if (!USE_LUT) asm volatile("" : "+v"(T[a]));
    }

    // weights: full window (wave-uniform) and, at the column borders, this lane's four own weights
    float cnt_full = t.tap[0];
#pragma unroll
    for (int k = 1; k < RING; k++) cnt_full = __fadd_rn(cnt_full, t.tap[k]);
    const float inv_full = __fdiv_rn(1.0f, cnt_full);
    // At the column borders every lane has four weights of its own.  They (and their reciprocals) live in the lane's
    // LDS slot `wts`, not in eight registers: the border strips are the register-hungriest instantiation and the one
    // that spilled (private segment -> every wave of the launch pays for scratch).
    if (COL_EDGE) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = x0 + j;
            float c = 0.0f;
#pragma unroll
            for (int k = 0; k < RING; k++) {
                int xx = x + k - C;
                if (xx >= 0 && xx < W) c = __fadd_rn(c, t.tap[k]);
            }
            const float cnt = (x >= 0 && x < W) ? c : 1.0f;
            wts[j] = cnt;
            wts[4 + j] = __fdiv_rn(1.0f, cnt);
        }
    }
    uint32_t wts_off = 0; // opaque zero added to the slot's address at every use, so that the reads stay in the loop
    auto row_quot = [&](float (&v)[4]) { // v[i] / (weight of output column i), in place
        if (FMA_DIV) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = __fmaf_rn(v[i], fma_c, v[i]);
        } else if (COL_EDGE) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            asm volatile("" : "+v"(wts_off));
            const float *w = reinterpret_cast<const float *>(reinterpret_cast<const char *>(wts) + wts_off);
            const f32x4 cw = *reinterpret_cast<const f32x4 *>(w), iw = *reinterpret_cast<const f32x4 *>(w + 4);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = div_by(v[i], cw[i], iw[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = div_by(v[i], cnt_full, inv_full);
        }
    };

    // Lane offsets of the row loads and stores, passed through an empty asm at every use: the optimiser then cannot
    // fold them into loop-invariant 64-bit VGPR pointers, and the addresses stay "uniform row base (SGPR pair) +
    // 32-bit lane offset" (saddr form).  Because the variable itself is what the asm "changes", its old value is
    // dead at that point and no register copy is needed (opaque_offset() on a loop-invariant value costs a v_mov
    // per use).  Both are only used where x0 >= 0.
    // (The systolic form loads C bytes right of x0: the constant goes into the uniform base.)
    uint32_t ld_off = (uint32_t)x0, st_off = (OUT_U8 ? 1u : 2u) * (uint32_t)x0;
    // Border strips (W >= 4, the launcher sees to that): a lane whose four columns hang over the left or right image
    // border loads the nearest dword that lies inside the row and shifts the outside bytes away -- zeros come in, which
    // is what an out-of-image pixel has to be (see "Bit-exactness" above).  One load per lane and row as in the
    // interior, no per-byte loads with 64-bit lane addresses (those were what made this instantiation spill).
    uint32_t sh_r = 0, sh_l = 0;
    const bool in_any = xin > -4 && xin < W; // at least one column inside
    if (COL_EDGE) {
        const int xc = min(max(xin, 0), W - 4);
        ld_off = (uint32_t)xc; // from the row's first byte: the offset register is UNSIGNED (xc - C would wrap)
        if (in_any) {
            sh_l = (uint32_t)(8 * (xc - xin)) & 31u;     // xin < 0: pixel 0 moves up to byte -xin
            sh_r = (uint32_t)(8 * (xin - xc)) & 31u;     // xin > W - 4: byte xin - (W - 4) moves down to byte 0
            if (xin >= 0) sh_l = 0;
            if (xin <= W - 4) sh_r = 0;
        }
    }

    auto load_row = [&](int r) -> uint32_t {
        if (ROW_EDGE && (r < 0 || r >= H)) return 0u; // wave-uniform
        const uint8_t *p = jb.fimg + (size_t)r * W; // wave-uniform
        uint32_t v = 0u;
        asm volatile("" : "+v"(ld_off)); // see ld_off
        __builtin_memcpy(&v, p + (COL_EDGE ? 0 : xin - x0) + ld_off, 4);
        return v;
    };
    // applied where the row is USED (two rows after the load): shifting at the load would wait for it there
    auto fix_row = [&](uint32_t v) -> uint32_t {
        if (!COL_EDGE) return v;
        v = (v >> sh_r) << sh_l;
        return in_any ? v : 0u;
    };

    float acc[RING][4]; // open column sums; slot = phase of the row that is the output's tap 0
#pragma unroll
    for (int s = 0; s < RING; s++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[s][i] = 0.0f;

    // ROT (systolic row pass with the product table): the products of row r+1 are looked up BETWEEN the row pass and
    // the column pass of row r -- their registers are free then, and the table's latency (and its bank conflicts) pass
    // under the ~70 instructions of the column pass instead of in front of the next row pass.  Rows outside the image
    // arrive as zero bytes, whose products are the table's exact zeros, so the look-ups need no row test.
    // (Not in the border strips: with their shift amounts and weight slot the 24 products alive across the column pass
    // no longer fit the registers of five waves per SIMD.)
    constexpr bool ROT = SYS && USE_LUT && GAUSS_ROTATE_LOOKUPS && !COL_EDGE;
    float Q[4][C + 1]; // products of the row the next row pass works on
    auto products = [&](uint32_t cur) {
        if (USE_LUT) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float *row = lut + ((cur >> (8 * i)) & 0xffu);
#pragma unroll
                for (int a = 0; a <= C; a++) Q[i][a] = row[a * 256];
            }
        } else {
            const float v[4] = {(float)(cur & 0xffu), (float)((cur >> 8) & 0xffu), (float)((cur >> 16) & 0xffu),
                                (float)(cur >> 24)};
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int a = 0; a <= C; a++) Q[i][a] = __fmul_rn(v[i], T[a]);
        }
    };

    auto step = [&](auto ph, int r, uint32_t cur_raw, uint32_t next_raw) {
        constexpr int PH = decltype(ph)::value; // (r - rfirst) mod RING
        // ---- row pass of input row r ----------------------------------------------------------------
        float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!ROT) {
            if (!ROW_EDGE || (r >= 0 && r < H)) products(fix_row(cur_raw));
        }
        if (!ROW_EDGE || (r >= 0 && r < H)) {
            if constexpr (SYS) {
                constexpr int NL = SysCfg<C>::NL;
                // ch[j]: the sum whose first term is pixel j of the lane it started in; after t hand-overs pixel e of
                // the lane it is in is its tap k = e - j + 4t.  As above the DPP instructions are kept in runs.
                float ch[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { // stage 0 (plain): open the four sums that start here
                    ch[j] = Q[j][C];
#pragma unroll
                    for (int e = j + 1; e < 4 && e - j <= 2 * C; e++) {
                        const int k = e - j;
                        ch[j] = __fadd_rn(ch[j], Q[e][k < C ? C - k : k - C]);
                    }
                    if ((j + 2 * C) / 4 == 0) res[(j + 2 * C) % 4] = ch[j]; // window 3: pixels j .. j+2 are all mine
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 1; tt <= NL; tt++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) { // (DPP) take the sum over from the left with my pixel 0 added
                        const int k = 4 * tt - j;
                        if (tt > (j + 2 * C) / 4) continue; // finished in an earlier lane
                        ch[j] = __fadd_rn(lane_shr1(ch[j]), Q[0][k < C ? C - k : k - C]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) { // (plain) my other pixels
                        if (tt > (j + 2 * C) / 4) continue;
#pragma unroll
                        for (int e = 1; e < 4; e++) {
                            const int k = e - j + 4 * tt;
                            if (k > 2 * C) continue;
                            ch[j] = __fadd_rn(ch[j], Q[e][k < C ? C - k : k - C]);
                        }
                        if (tt == (j + 2 * C) / 4) // its last term was one of mine: output (j + 2C) mod 4 of this lane
                            res[(j + 2 * C) % 4] = ch[j];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                row_quot(res);
            } else {
            // The leading terms of output i -- the pixels left of this lane -- are summed by the LEFT neighbour
            // (same operands, same order, so the same bits) and arrive as one value whose wave shift folds into
            // the next add: every lane therefore computes pre[i] for its right neighbour.  Without this each of the
            // 4 outputs starts with a bare v_mov_b32_dpp and the two-lane hop costs another.
            //
            // Instruction ORDER matters as much as the count here: a DPP instruction between plain v_add_f32 costs
            // the SIMD ~11 cycles, in a run of DPP instructions ~5, a plain add ~2.3 (tools/valu_issue_bench.hip:
            // "52 plain then 12 DPP adds" 2.84 cycles per instruction against 3.93 with the same 12 spread singly).
            // So the four output chains advance in lock step through phases -- all their DPP operations together,
            // all their plain adds together -- with scheduling barriers between the phases.
            float pre[4];
            // phase 1 (DPP): the terms the right neighbour's prefix takes from MY left neighbour
#pragma unroll
            for (int i = 0; i < 4; i++) {
                pre[i] = 0.0f;
                bool open = false;
#pragma unroll
                for (int d = -C; d <= C; d++) {
                    const int e = i + d + 4, a = d < 0 ? -d : d; // the right neighbour's pixel i+d is my pixel e
                    if (e >= 0) continue;
                    const float term = lane_shr1(Q[e + 4][a]);
                    pre[i] = open ? __fadd_rn(pre[i], term) : term;
                    open = true;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 2 (plain): ... and from my own pixels
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int d = -C; d <= C; d++) {
                    const int e = i + d + 4, a = d < 0 ? -d : d;
                    if (e < 0 || e >= 4) continue;
                    pre[i] = (d == -C) ? Q[e][a] : __fadd_rn(pre[i], Q[e][a]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 3 (DPP): each output starts from the prefix its left neighbour computed, folded into the add of
            // its first own term
            float sum[4];
            bool started[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                started[i] = i - C < 0;
                sum[i] = 0.0f;
                if (started[i]) {
                    const int d0 = -i, a0 = i;      // first own pixel: e = 0
                    sum[i] = __fadd_rn(lane_shr1(pre[i]), Q[0][a0]);
                    (void)d0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 4 (plain): the rest of the own pixels
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int d = -C; d <= C; d++) {
                    const int e = i + d, a = d < 0 ? -d : d; // pixel e (relative to x0) at distance a
                    if (e < 0 || e >= 4) continue;
                    if (started[i] && e == 0) continue; // added in phase 3
                    sum[i] = (!started[i] && d == -C) ? Q[e][a] : __fadd_rn(sum[i], Q[e][a]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 5 (DPP): the right neighbours' pixels, the four chains interleaved (distance by distance)
#pragma unroll
            for (int e = 4; e <= 3 + C; e++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int d = e - i;
                    if (d > C) continue;
                    const int hop = e / 4;
                    float term = lane_shl1(Q[e - 4 * hop][d]);
                    if (hop == 2) term = lane_shl1(term);
                    sum[i] = __fadd_rn(sum[i], term);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) res[i] = sum[i];
            row_quot(res);
            }
        }

        if (ROT) {
            __builtin_amdgcn_sched_barrier(0);
            products(fix_row(next_raw));
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- column pass: row r is tap k of output row r + C - k --------------------------------------
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float P[C + 1];
#pragma unroll
            for (int a = 0; a <= C; a++) P[a] = __fmul_rn(res[i], T[a]);
            acc[PH][i] = P[C]; // k = 0 opens output row r + C
#pragma unroll
            for (int k = 1; k < RING; k++) {
                const int slot = (PH + RING - k) % RING;
                acc[slot][i] = __fadd_rn(acc[slot][i], P[k < C ? C - k : k - C]);
            }
        }

        // ---- output row y = r - C just received its last tap -------------------------------------------
        // Pin every open sum in a register HERE.  Nothing in this row uses the sums it has just advanced, so
        // the optimiser otherwise sinks those adds (and their products) down to the row that finally stores
        // them and carries 2C rows of row-pass results instead -- twice the registers, spills.
#pragma unroll
        for (int s = 0; s < RING; s++)
            asm volatile("" : "+v"(acc[s][0]), "+v"(acc[s][1]), "+v"(acc[s][2]), "+v"(acc[s][3]));

        // The quotient is taken unconditionally too (also for the 2C warm-up rows, whose sums are incomplete
        // and never stored), so that no add can move into the store's branch.
        const int y = r - C;
        constexpr int DONE = (PH + 1) % RING;
        const bool in_seg = y >= ybeg && y < yend; // wave-uniform
        float cnt_v = cnt_full, inv_v = inv_full;
        if (ROW_EDGE && in_seg && (y < C || y + C >= H)) { // top/bottom border rows renormalise
            cnt_v = 0.0f;
            for (int k = 0; k < RING; k++) {
                int yy = y + k - C;
                if (yy >= 0 && yy < H) cnt_v = __fadd_rn(cnt_v, t.tap[k]);
            }
            inv_v = __fdiv_rn(1.0f, cnt_v);
        }
        // float -> short truncates toward zero (src/utils.cpp:62)
        auto quot = [&](float a) { return FMA_DIV ? __fmaf_rn(a, fma_c, a) : div_by(a, cnt_v, inv_v); };
        uint2 pk;
        {
            const int o0 = (int)quot(acc[DONE][0]), o1 = (int)quot(acc[DONE][1]);
            const int o2 = (int)quot(acc[DONE][2]), o3 = (int)quot(acc[DONE][3]);
            if (OUT_U8) {
                // quotients of non-negative sums <= 255 * (1 + ulps): 0 <= o <= 255
                pk.x = ((uint32_t)o0 | ((uint32_t)o1 << 8)) | (((uint32_t)o2 | ((uint32_t)o3 << 8)) << 16);
                pk.y = 0u;
            } else {
                // quotients of non-negative sums: 0 <= o < 65536, so the low halves need no mask (one v_lshl_or_b32 per pair)
                pk.x = (uint32_t)o0 | ((uint32_t)o1 << 16);
                pk.y = (uint32_t)o2 | ((uint32_t)o3 << 16);
            }
        }
        asm volatile("" : "+v"(pk.x), "+v"(pk.y)); // materialise here, whatever the branch below does
        if (in_seg && owner) {
            // owner lanes have x0 >= 0; byte offset so that no 64-bit shift is needed per lane
            asm volatile("" : "+v"(st_off));
            if (OUT_U8) {
                uint8_t *dst = jb.fout8 + (size_t)y * W + st_off;
                if (!COL_EDGE || full4) {
                    __builtin_memcpy(dst, &pk.x, 4);
                } else {
                    dst[0] = (uint8_t)pk.x;
                    if (x0 + 1 < W) dst[1] = (uint8_t)(pk.x >> 8);
                    if (x0 + 2 < W) dst[2] = (uint8_t)(pk.x >> 16);
                }
            } else {
                int16_t *dst = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(jb.fout + (size_t)y * W) + st_off);
                if (!COL_EDGE || full4) {
                    __builtin_memcpy(dst, &pk, 8);
                } else {
                    dst[0] = (int16_t)pk.x;
                    if (x0 + 1 < W) dst[1] = (int16_t)(pk.x >> 16);
                    if (x0 + 2 < W) dst[2] = (int16_t)pk.y;
                }
            }
        }
    };

    // Rows ybeg-C .. yend-1+C, count rounded up to a multiple of RING: accumulator slots and the prefetch
    // registers (two rows ahead) then rotate by renaming only.
    const int rfirst = ybeg - C;
    const int rows = yend - ybeg + 2 * C;
    const int rlast = rfirst + RING * ((rows + RING - 1) / RING) - 1;
    // (ROT: three rows ahead -- the look-ups need a row half a step earlier)
    constexpr int AHEAD = ROT ? 3 : 2;
    uint32_t pf[RING];
    pf[0] = load_row(rfirst);
    pf[1] = load_row(rfirst + 1);
    if (ROT) {
        pf[2] = load_row(rfirst + 2);
        products(fix_row(pf[0]));
    }
    for (int r = rfirst; r <= rlast; r += RING) {
        for_each_phase(
            [&](auto ph) {
                constexpr int PH = decltype(ph)::value;
                pf[(PH + AHEAD) % RING] = load_row(r + PH + AHEAD);
                step(ph, r + PH, pf[PH], pf[(PH + 1) % RING]);
                // keep the scheduler from interleaving rows: one row's products are all the registers allow
                __builtin_amdgcn_sched_barrier(0);
            },
            std::make_integer_sequence<int, RING>{});
    }
}

} // namespace

// Live state per lane: 4(2C+1) open sums + 4(C+1) products + ~30; without an occupancy target the scheduler
// interleaves several rows' products and doubles that.
template <int C, bool USE_LUT, bool OUT_U8 = false, bool SYS = false>
// (6 waves per SIMD spill: 2.05 ms against 1.05; 4 compile to the same code as 5 -- profiles/r02/ab4_*.txt)
// (Window 17 -- 68 open sums + 36 products -- does not fit the 128 registers of four waves: 193 spilled registers, and
// three waves without spills are 3 % faster, 0.945 against 0.971 ms per 64 x 4K.  Window 15 spills five registers at
// four waves and is still 6 % faster there than at three, 0.772 against 0.825 ms: profiles/r03/ab24_*.txt.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C <= 5 ? 5 : C <= 7 ? 4 : 3)))
void gauss_sym_kernel(const uint8_t *__restrict__ img, void *__restrict__ out, int H, int W, int n_strips,
                      int n_segs, int seg_rows, int total_waves, GaussTaps t, int use_fma_div, float fma_c)
{
    using K = MarchCfg<C>;
    // product table of the row pass (USE_LUT): the only LDS use and the only workgroup barrier of the kernel
    __shared__ float lut_mem[USE_LUT ? (C + 1) * 256 : 1];
    if (USE_LUT) {
#pragma unroll
        for (int a = 0; a <= C; a++) lut_mem[a * 256 + threadIdx.x] = __fmul_rn((float)threadIdx.x, t.tap[C - a]);
        __syncthreads();
    }
    const float *lut = lut_mem;
    // per-lane column-border weights (border strips only, see gauss_sym_strip)
    __shared__ __attribute__((aligned(16))) float wts_mem[256 * 8];
    float *wts = wts_mem + threadIdx.x * 8;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform
    if (wave >= total_waves) return;

    const int f = wave / (n_strips * n_segs);
    const MarchCell cell = march_cell_of(wave - f * (n_strips * n_segs), n_segs, n_strips); // border cells first
    const int s = cell.strip, g = cell.seg;
    GaussJob jb;
    jb.lane = lane;
    jb.H = H;
    jb.W = W;
    jb.ybeg = g * seg_rows;
    jb.yend = min(H, jb.ybeg + seg_rows);
    constexpr int SW = SYS ? SysCfg<C>::SW : K::SW, LEFT = SYS ? SysCfg<C>::NL : K::HL;
    jb.x0 = s * SW + (lane - LEFT) * 4; // first of this lane's 4 output columns (halo lanes may be outside)
    jb.fimg = img + (size_t)f * H * W;
    jb.fout = (int16_t *)out + (size_t)f * H * W;
    jb.fout8 = (uint8_t *)out + (size_t)f * H * W;

    // the strip's lanes load columns [s*SW - 4HL, s*SW + SW + 4HL); systolic: [s*SW - 4NL + C, s*SW + SW + C)
    const bool col_edge = SYS ? (s * SW - 4 * LEFT + C < 0) || (s * SW + SW + C > W)
                              : (s * SW - 4 * LEFT < 0) || (s * SW + SW + 4 * LEFT > W);
    // rows loaded: ybeg-C .. yend-1+C, up to 2C more for the rounding to whole loop trips, +2 prefetched
    const bool row_edge = (jb.ybeg - C < 0) || (jb.yend + C + K::RING + 1 >= H); // (+3 with rotated look-ups)
    if (col_edge) {
        if (row_edge)
            gauss_sym_strip<C, true, true, false, USE_LUT, OUT_U8, SYS>(jb, t, lut, wts);
        else
            gauss_sym_strip<C, true, false, false, USE_LUT, OUT_U8, SYS>(jb, t, lut, wts);
    } else {
        if (row_edge)
            gauss_sym_strip<C, false, true, false, USE_LUT, OUT_U8, SYS>(jb, t, lut, wts);
        else if (use_fma_div)
            gauss_sym_strip<C, false, false, true, USE_LUT, OUT_U8, SYS>(jb, t, lut, wts, fma_c);
        else
            gauss_sym_strip<C, false, false, false, USE_LUT, OUT_U8, SYS>(jb, t, lut, wts);
    }
}

template <int C>
__global__ __launch_bounds__(MarchCfg<C>::WPB * 64) void gauss_march_kernel(
    const uint8_t *__restrict__ img, int16_t *__restrict__ out, int H, int W, int n_strips, int n_segs, int seg_rows,
    int total_waves, GaussTaps t, int use_fma_div, float fma_c)
{
    using K = MarchCfg<C>;
    __shared__ __attribute__((aligned(16))) float lds[K::WPB * K::WAVE_FLOATS];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * K::WPB + wib; // wave-uniform
    if (wave >= total_waves) return;
    float *rowbuf = lds + wib * K::WAVE_FLOATS;
    float *colring = rowbuf + K::ROWBUF + lane * 4; // this lane's 4 columns, slot stride 256 floats

    const int s = wave % n_strips;
    const int g = (wave / n_strips) % n_segs;
    const int f = wave / (n_strips * n_segs);
    GaussJob jb;
    jb.lane = lane;
    jb.H = H;
    jb.W = W;
    jb.ybeg = g * seg_rows;
    jb.yend = min(H, jb.ybeg + seg_rows);
    jb.x0 = s * K::SW + (lane - K::HL) * 4; // first of this lane's 4 columns (halo lanes may be outside)
    jb.fimg = img + (size_t)f * H * W;
    jb.fout = out + (size_t)f * H * W;
    jb.fout8 = nullptr;

    if (lane < 4 * K::HL) { // the pads of the row buffer are only ever read by halo lanes; keep them finite
        rowbuf[lane] = 0.0f;
        rowbuf[(64 + K::HL) * 4 + lane] = 0.0f;
    }

    // the strip's lanes span columns [s*SW - 4HL, s*SW + SW + 4HL); taps reach C <= 4HL beyond owned columns
    const bool col_edge = (s * K::SW - 4 * K::HL < 0) || (s * K::SW + K::SW + 4 * K::HL > W);
    // rows touched: ybeg-C .. yend-1+C (+2 rounding, +2 prefetch)
    const bool row_edge = (jb.ybeg - C < 0) || (jb.yend + C + 3 >= H);
    if (col_edge) {
        if (row_edge)
            gauss_march_strip<C, true, true>(jb, t, rowbuf, colring);
        else
            gauss_march_strip<C, true, false>(jb, t, rowbuf, colring);
    } else {
        if (row_edge)
            gauss_march_strip<C, false, true>(jb, t, rowbuf, colring);
        else if (use_fma_div)
            gauss_march_strip<C, false, false, true>(jb, t, rowbuf, colring, fma_c);
        else
            gauss_march_strip<C, false, false>(jb, t, rowbuf, colring);
    }
}

// Full-window weights S (bit patterns) for which a / S == fma(a, c, a) for EVERY float a in [0, 256];
// found and re-verified exhaustively on the device (tools/probe_div_fma.py, tests/test_gpu_numerics.py).
// S = 1 - 4..+2 ulps covers what the normalised taps sum to in practice (sigma 0.8/1.0/2.5 give exactly 1,
// 0.5/1.2/1.6 give 1 - 2^-24, 1.4/2.0 give 1 + 2^-23); any other S keeps the 5-op division.
static const unsigned kFmaDivTable[][2] = {
    {0x3f800000u, 0x00000000u}, // S = 1                c = 0
    {0x3f800001u, 0xb3fffffeu}, // S = 1 + 2^-23        c = -(2^-23 - 2^-46)
    {0x3f800002u, 0xb47ffffcu}, // S = 1 + 2^-22        c = -(2^-22 - 2^-44)
    {0x3f7fffffu, 0x33800001u}, // S = 1 - 2^-24        c = 2^-24 + 2^-47
    {0x3f7ffffeu, 0x34000001u}, // S = 1 - 2^-23        c = 2^-23 + 2^-46
    {0x3f7ffffdu, 0x34400003u}, // S = 1 - 3*2^-24
    {0x3f7ffffcu, 0x34800002u}, // S = 1 - 2^-22        c = 2^-22 + 2^-44
};

#if CANNY_GAUSS_HAS_HOST
int gaussian_fma_div_table(const unsigned (**table)[2])
{
    *table = kFmaDivTable;
    return (int)(sizeof(kFmaDivTable) / sizeof(kFmaDivTable[0]));
}

bool g_gauss_fma_div_enabled = true; // A/B switch (canny_hip_ctx_set_option "gaussian_fma_div")
void gaussian_set_fma_div(bool on) { g_gauss_fma_div_enabled = on; }
// A/B switch "tune_gaussian_variant": 0 = symmetric-tap kernel, systolic row pass, products looked up in an LDS
// table (default), 1 = LDS ring kernel, 2 = symmetric-tap kernel that multiplies and fetches products (round 1),
// 3 = product-fetching row pass with the table (rounds 2-3 default), 4 = systolic row pass that multiplies
int g_gauss_march_variant = 0;
void gaussian_set_march_variant(int v) { g_gauss_march_variant = v; }
int g_gauss_seg_target = 0; // A/B switch "tune_gaussian_seg": approximate rows per wave segment, 0 = automatic
void gaussian_set_seg_target(int rows) { g_gauss_seg_target = rows; }
#endif

#if CANNY_GAUSS_HAS_HOST
// ---- exhaustive check of div_by against the IEEE divide (test hook) -----------------------------------
// fma_c == 0: the 5-op div_by;  fma_c != 0 (passed as a bit pattern so that c = 0.0f is expressible through
// use_fma): the one-instruction candidate a/b ~ fma(a, c, a) used for the full-window weight b = 1 +- ulp.
__global__ __launch_bounds__(256) void selftest_div_kernel(float b, int use_fma, float c, unsigned first_bits,
                                                           unsigned last_bits, unsigned long long *mismatches)
{
    const float y = __fdiv_rn(1.0f, b);
    unsigned long long bad = 0, worst = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long u = first_bits + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= last_bits;
         u += stride) {
        const float a = __uint_as_float((unsigned)u);
        const float want = __fdiv_rn(a, b);
        const float got = use_fma ? __fmaf_rn(a, c, a) : div_by(a, b, y);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            bad++;
            worst = u; // u ascends within a thread
        }
    }
    if (bad) {
        atomicAdd(mismatches, bad);
        atomicMax(mismatches + 1, worst); // bit pattern of the largest mismatching dividend
    }
}

hipError_t launch_selftest_div(float b, int use_fma, float c, unsigned first_bits, unsigned last_bits,
                               unsigned long long *d_mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_div_kernel, dim3(256 * 16), dim3(256), 0, stream, b, use_fma, c, first_bits, last_bits,
                       d_mismatches);
    return hipGetLastError();
}

#endif // CANNY_GAUSS_HAS_HOST

// out_u8: 0 = s16 plane, 1 = u8 plane; the u8 form exists for the symmetric-tap kernel with the product table only
template <int C>
static hipError_t launch_march_c(const uint8_t *img, void *out, int height, int width, int n_frames,
                                 const GaussTaps &taps, hipStream_t stream, int out_u8)
{
    using K = MarchCfg<C>;
    if (width < 4) return hipErrorInvalidValue; // the border strips load whole dwords inside a row
    // the symmetric-tap kernel needs tap[C-a] == tap[C+a] bit for bit (true for the reference's taps)
    bool symmetric = g_gauss_march_variant != 1;
    for (int a = 1; a <= C && symmetric; a++)
        symmetric = std::memcmp(&taps.tap[C - a], &taps.tap[C + a], sizeof(float)) == 0;
    const bool systolic = symmetric && (g_gauss_march_variant == 0 || g_gauss_march_variant == 4);
    const int strip_w = systolic ? SysCfg<C>::SW : K::SW;
    int n_strips = (width + strip_w - 1) / strip_w;
    // longest segments that still give the chip a few thousand waves; the symmetric kernel processes
    // rows in trips of 2C+1, so its segments are sized to make seg + 2C a whole number of trips
    auto seg_for = [&](int target) {
        return symmetric ? std::max(1, (target + 2 * C + K::RING / 2) / K::RING) * K::RING - 2 * C : target;
    };
    int target = 256;
    while (target > 32 &&
           (long long)n_frames * n_strips * ((height + seg_for(target) - 1) / seg_for(target)) < 16384)
        target >>= 1;
    // a single frame leaves most SIMDs without a wave even then: shorter segments still (down to one loop trip,
    // i.e. up to half the rows of a wave are halo) as long as that gives idle SIMDs something to do
    while (target > 8 &&
           (long long)n_frames * n_strips * ((height + seg_for(target) - 1) / seg_for(target)) < 2048)
        target >>= 1;
    int seg = seg_for(g_gauss_seg_target >= 8 ? g_gauss_seg_target : target);
    // Batches that fill the chip several times over: the waves of a launch run in "rounds" of as many waves as the
    // chip holds (all waves do the same amount of work), and the last round costs a whole round however few waves are
    // left for it.  Pick the segment length that minimises rounds x rows per wave (halo rows included) -- e.g.
    // 128 x 4K at window 11: 144-row segments are 15 per frame = 30720 waves = exactly 6 rounds of 5120, where the
    // 133-row default of the rule above needs 7 rounds (17 segments per frame, the last one 32 rows): 1.069 -> 1.032 ms
    // (profiles/r03/ab4_segments_whole_rounds.txt; 166 rows: 1.120, 100 rows: 1.050, as the model ranks them).
    if (symmetric && g_gauss_seg_target < 8) {
        static int n_simds = 0; // SIMDs of the device (4 per CU); one device family, so one value per process
        if (n_simds == 0) {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            n_simds = 4 * cus;
        }
        // waves per SIMD the kernel's registers allow (gauss_sym_kernel<C, true>: 38/49/68/76/95/105/125/128 VGPRs)
        static const int kWavesPerSimd[9] = {0, 8, 8, 7, 6, 5, 4, 4, 3};
        const long long slots = (long long)n_simds * kWavesPerSimd[C];
        const long long frames_strips = (long long)n_frames * n_strips;
        if (frames_strips * ((height + seg - 1) / seg) >= 2 * slots) {
            long long best_cost = -1;
            for (int m = 4; m * K::RING - 2 * C <= 400; m++) {
                const int s_rows = m * K::RING - 2 * C;
                if (s_rows < 48) continue;
                const long long w = frames_strips * ((height + s_rows - 1) / s_rows);
                const long long cost = ((w + slots - 1) / slots) * (long long)(s_rows + 2 * C);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    seg = s_rows;
                }
            }
        }
    }
    int n_segs = (height + seg - 1) / seg;
    long long waves = (long long)n_frames * n_strips * n_segs;
    if (waves > 0x7fffffffLL) return hipErrorInvalidValue;
    const int wpb = symmetric ? 4 : K::WPB;
    unsigned blocks = (unsigned)((waves + wpb - 1) / wpb);
    // full-window weight, summed exactly like the kernel (and the reference) does: ascending float adds
    volatile float s = taps.tap[0]; // volatile: keep the host compiler from re-associating / widening
    for (int k = 1; k < K::RING; k++) s = s + taps.tap[k];
    const float full = s;
    unsigned full_bits;
    std::memcpy(&full_bits, &full, sizeof(full_bits));
    int use_fma = 0;
    float fma_c = 0.0f;
    for (const auto &e : kFmaDivTable)
        if (e[0] == full_bits && g_gauss_fma_div_enabled) {
            use_fma = 1;
            std::memcpy(&fma_c, &e[1], sizeof(fma_c));
        }
    const bool table = g_gauss_march_variant == 0 || g_gauss_march_variant == 3;
    if (out_u8 && !(symmetric && table)) return hipErrorNotSupported;
#define CANNY_LAUNCH_SYM(LUT, U8, SYS)                                                                                  \
    hipLaunchKernelGGL((gauss_sym_kernel<C, LUT, U8, SYS>), dim3(blocks), dim3(256), 0, stream, img, out, height,     \
                       width, n_strips, n_segs, seg, (int)waves, taps, use_fma, fma_c)
    if (symmetric && table && out_u8) {
        if (systolic) CANNY_LAUNCH_SYM(true, true, true); else CANNY_LAUNCH_SYM(true, true, false);
    } else if (symmetric && table) {
        if (systolic) CANNY_LAUNCH_SYM(true, false, true); else CANNY_LAUNCH_SYM(true, false, false);
    } else if (symmetric) {
        if (systolic) CANNY_LAUNCH_SYM(false, false, true); else CANNY_LAUNCH_SYM(false, false, false);
    } else {
        hipLaunchKernelGGL(gauss_march_kernel<C>, dim3(blocks), dim3(K::WPB * 64), 0, stream, img, (int16_t *)out,
                           height, width, n_strips, n_segs, seg, (int)waves, taps, use_fma, fma_c);
    }
#undef CANNY_LAUNCH_SYM
    return hipGetLastError();
}

#define CANNY_GAUSS_CASE(C) \
    case C: return launch_march_c<C>(img, out, height, width, n_frames, taps, stream, out_u8)
#if CANNY_GAUSS_PART == 1
hipError_t launch_gauss_march_part1(int center, const uint8_t *img, void *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream, int out_u8)
{
    switch (center) {
        CANNY_GAUSS_CASE(5);
        CANNY_GAUSS_CASE(6);
    default: return hipErrorNotSupported;
    }
}
#elif CANNY_GAUSS_PART == 2
hipError_t launch_gauss_march_part2(int center, const uint8_t *img, void *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream, int out_u8)
{
    switch (center) {
        CANNY_GAUSS_CASE(7);
    default: return hipErrorNotSupported;
    }
}
#elif CANNY_GAUSS_PART == 3
hipError_t launch_gauss_march_part3(int center, const uint8_t *img, void *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream, int out_u8)
{
    switch (center) {
        CANNY_GAUSS_CASE(8);
    default: return hipErrorNotSupported;
    }
}
#endif

#if CANNY_GAUSS_HAS_HOST
// (width >= 4: the border strips load whole dwords that must lie inside the row; narrower images take the generic path)
bool gaussian_march_supported(int center, int, int width) { return center >= 1 && center <= 8 && width >= 4; }

static hipError_t launch_march_any(const uint8_t *img, void *out, int height, int width, int n_frames,
                                   const GaussTaps &taps, hipStream_t stream, int out_u8)
{
    switch (taps.center) {
#ifdef CANNY_GAUSS_ONLY_C // development aid: compile one window only (-DCANNY_GAUSS_ONLY_C=5) to read its ISA quickly
        CANNY_GAUSS_CASE(CANNY_GAUSS_ONLY_C);
#else
        CANNY_GAUSS_CASE(1);
        CANNY_GAUSS_CASE(2);
        CANNY_GAUSS_CASE(3);
        CANNY_GAUSS_CASE(4);
#if CANNY_GAUSS_PART == 0
    case 5:
    case 6: return launch_gauss_march_part1(taps.center, img, out, height, width, n_frames, taps, stream, out_u8);
    case 7: return launch_gauss_march_part2(taps.center, img, out, height, width, n_frames, taps, stream, out_u8);
    case 8: return launch_gauss_march_part3(taps.center, img, out, height, width, n_frames, taps, stream, out_u8);
#else
        CANNY_GAUSS_CASE(5);
        CANNY_GAUSS_CASE(6);
        CANNY_GAUSS_CASE(7);
        CANNY_GAUSS_CASE(8);
#endif
#endif
    default: return hipErrorNotSupported;
    }
}

hipError_t launch_gaussian_march(const uint8_t *img, int16_t *out, int height, int width, int n_frames,
                                 const GaussTaps &taps, hipStream_t stream)
{
    return launch_march_any(img, out, height, width, n_frames, taps, stream, 0);
}

bool gaussian_march_u8_supported(const GaussTaps &taps)
{
    if (taps.center < 1 || taps.center > 8 || (g_gauss_march_variant != 0 && g_gauss_march_variant != 3)) return false;
    for (int a = 1; a <= taps.center; a++)
        if (std::memcmp(&taps.tap[taps.center - a], &taps.tap[taps.center + a], sizeof(float)) != 0) return false;
    return true;
}

hipError_t launch_gaussian_march_u8(const uint8_t *img, uint8_t *out, int height, int width, int n_frames,
                                    const GaussTaps &taps, hipStream_t stream)
{
    return launch_march_any(img, out, height, width, n_frames, taps, stream, 1);
}
#endif // CANNY_GAUSS_HAS_HOST
#undef CANNY_GAUSS_CASE

} // namespace canny
