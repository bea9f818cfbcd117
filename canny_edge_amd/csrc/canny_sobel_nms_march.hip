// canny_sobel_nms_march.hip -- fused Sobel + non-maximal suppression, wave-marching; the stencils use no LDS
// (the variant canny() runs stages its hysteresis plane bytes there on their way out).
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
//       neighbours of the bin; neighbours outside the image carry the lowest magnitude there is (0, or the
//       threshold floor of the PLANES kernel), which never changes an output (= "skip", see `skip` below).
// Three rows of d/t, of magnitudes and of bin discriminants rotate through registers; the row loop is
// unrolled by 3 so that every rotation is a compile-time renaming and the three prefetched rows never
// have to be copied while their loads are in flight.  A row's output pixels are stored one step late, at the
// top of the next row's step (DEFER below: loads and stores share the vmcnt counter), and the launch takes a
// frame's border cells first (march_cell_of, canny_kernels.h).  The f32 arithmetic canny() runs since round 3
// is fmarch_strip, further down; the description here is the packed-i16 form of the stage kernel.
//
// Border conventions of the reference (src/utils.cpp:114-186, 248-308):
//   gx: column clamp  -> zero-filled neighbours plus a +-s fix-up at columns 0 and W-1; rows dropped
//       (virtual rows are zero, which is exactly "dropped");
//   gy: columns dropped (zero fill), row clamp -> t[-1]:=t[0], t[H]:=t[H-1];
//   NMS: out-of-image neighbours skipped -> lowest magnitude.
// Column borders exist only in the first and last strip: those waves run the COL_EDGE=true
// instantiation, every other wave runs code with no lane-varying border logic at all.  Row borders are
// wave-uniform (the wave index goes through readfirstlane) and sit behind scalar branches.
// Precondition: smoothed values in [0,255] (what gaussian() produces), so |gx|,|gy| <= 1020.
#include "canny_kernels.h"

#include <hip/hip_ext.h>

#include <type_traits>

namespace canny {

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_v(uint32_t u) { return __builtin_bit_cast(s16x2, u); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_v(a) + as_v(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_v(a) - as_v(b)); }
// a*2 + b per half as ONE v_pk_mad_i16 (hipcc otherwise emits v_pk_mul_lo_u16 / v_pk_lshlrev_b16 plus
// an add; this kernel is VALU-issue bound, so every instruction per pixel counts).
__device__ __forceinline__ uint32_t pk_mad2(uint32_t a, uint32_t b)
{
    uint32_t r;
    const uint32_t two = 0x00020002u;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(two), "v"(b));
    return r;
}

// value held by the lane to the left / right (0 at the wave's ends)
// bound_ctrl: a lane whose source lies outside the wave reads 0, so no `old` value has to be
// materialised (with update_dpp(0, ...) every shift cost an extra v_mov_b32 v, 0).
__device__ __forceinline__ uint32_t from_left(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t from_right(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
// pair.hi16 = (a > b) ? a : 0, low half untouched: the SDWA form of v_cndmask writes the kept magnitude
// straight into the upper half of the output pair, which saves the v_cvt_pk_u16_u32 per pixel pair.
__device__ __forceinline__ void keep_gt_hi(uint32_t &pair, int a, int b, int zero)
{
    // (gfx940+: a VALU that reads an SGPR or VCC another VALU has just written needs two wait states in between;
    // nothing inside an asm string is padded by the compiler)
    asm("v_cmp_gt_i32 vcc, %1, %2\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_sdwa %0, %3, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        : "+v"(pair)
        : "v"(a), "v"(b), "v"(zero)
        : "vcc");
}

// cbits = 2*cbits + (a > b), sbits = 2*sbits + (a > c) with a wave-uniform c: v_cmp + v_addc per bit.  (Written
// in C the compiler builds each bit with v_cndmask and merges them with shifts and v_or3: three instructions per
// bit instead of two.)  The two compares come first, so that each carry-in is read two wait states after the
// compare that wrote it (gfx940+ hazard: VALU writes SGPR/VCC -> VALU reads it) at the price of one s_nop.
__device__ __forceinline__ void push_gt2(unsigned &cbits, unsigned &sbits, int a, int b, int c)
{
    unsigned long long m; // the second compare's lane mask
    asm("v_cmp_gt_i32 vcc, %3, %4\n\t"
        "v_cmp_gt_i32 %2, %3, %5\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
        "v_addc_co_u32 %1, %2, %1, %1, %2"
        : "+v"(cbits), "+v"(sbits), "=&s"(m)
        : "v"(a), "v"(b), "s"(c)
        : "vcc");
}

// NP = packed s16 pairs per lane: 4 (8 pixels, 16-byte accesses, ~146 VGPRs: 3 waves per SIMD) or 2 (4 pixels,
// 8-byte accesses, fewer registers: more resident waves to hide the store and load latency behind).
template <int NP>
struct SnmCfg {
    static constexpr int PX = 2 * NP;  // pixels per lane
    static constexpr int SW = 62 * PX; // output columns per strip (lanes 0 and 63 are halo lanes)
};
constexpr int SNM_STAGE_ROWS = 64; // rows of plane bytes a wave can park in LDS = longest segment of the fused kernel
constexpr int SNM_WPB = 4;          // waves per workgroup (independent of each other)

template <int N>
using IC = std::integral_constant<int, N>;
#ifndef FLT_WAVES
#define FLT_WAVES 4
#endif
// The f32 kernel's phases are fenced for the instruction scheduler: left alone it interleaves the eight pixels of a row
// (and the production of one row with the suppression of the previous one) for instruction-level parallelism that a
// wave on this hardware cannot use (one VALU instruction per ~6 cycles per wave whatever the dependences),
// at 150-170 VGPRs; in groups of FLT_GROUP pixels the same code needs under 128.
#ifndef FLT_GROUP
#define FLT_GROUP 2
#endif
#ifndef FLT_EARLY_RELOAD
#define FLT_EARLY_RELOAD 1 // f32 strips: a row buffer is reloaded (row r+3) as soon as row r is converted
#endif
#ifndef FLT_DEFER_STORE
#define FLT_DEFER_STORE 1 // a row's output pixels are stored at the top of the NEXT row's step (see march_strip; 0: at once, for A/B)
#endif
#ifndef FLT_CARRIERS
#define FLT_CARRIERS 1 // 1: one v_perm_b32-packed bin carrier per pixel; 2: the two floats themselves
#endif
#define FLT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct StripJob {
    const int16_t *fin;
    const uint8_t *fin8; // IN_U8 kernels: the smoothed plane as bytes (its values lie in [0,255])
    int16_t *fout;
    int H, W, ybeg, yend, x0, lane;
    // PLANES only: this frame's hysteresis bit-planes (tile-major, see HystGeom) and the thresholds
    uint8_t *pconn, *pstrong;
    int tiles_x, lo1, hi1; // lo1 = min_val - 1, hi1 = max(min_val, max_val) - 1
    int edge_value;        // what a strong pixel becomes in the edge map (EDGE, or 0 if max_val > EDGE)
};

// COL_EDGE: the strip touches column 0 or W-1 (lane-varying masks, partial lanes).
// ROW_EDGE: the segment touches row 0 or H-1 (virtual rows, row clamp).  With both false the body is
// straight-line code with unconditional 16-byte loads, so the compiler can keep two rows of loads in
// flight behind counted s_waitcnt vmcnt(N).
// PLANES: instead of the s16 suppressed magnitudes, emit what hysteresis' classify step would derive from
// them (src/utils.cpp:331-351): bit-plane `connectable` (v >= min_val) and `strong` (v >= min_val and
// v >= max_val), one byte per lane and row.  Requires min_val >= 1 (then v >= min_val implies the pixel
// survived NMS, so the threshold folds into the NMS comparison: mc > max(neighbour, min_val - 1)) and
// W % 8 == 0 (a lane's 8 pixels are one plane byte).  The s16 plane it writes is not the suppressed magnitude
// but the provisional EDGE MAP (strong -> edge value, else 0), which the propagation sweeps complete in place:
// the classify pass (2 B/px of loads) and the finalize pass (2 B/px of stores in an HBM-bound kernel) both
// disappear from the pipeline.  Stores are what this kernel's row loop stalls on ("store latency"),
// so the plane bytes do not go out row by row: see LDS_PLANES below.
// max of two magnitudes as ONE v_max_u16.  Magnitudes are 0..1442, so the unsigned 16-bit maximum of the low
// halves is the maximum, and the instruction zeroes the upper half of its result (gfx9 VOP2 16-bit rule).  It is
// here for its issue cost: v_max_u16 with VGPR operands issues every ~2.3 cycles per SIMD like v_add_f32, while
// v_max_i32 / v_max3_i32 take ~4.2 (tools/valu_issue_bench.hip, DESIGN.md "instruction classes").
__device__ __forceinline__ int max_u16(int a, int b)
{
    int r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One row of results of a lane (16 or 8 bytes).  (A non-temporal hint on these stores measured no different:
// profiles/r02/ab5_nt_stores.txt.)
template <int NP>
__device__ __forceinline__ void store_row(int16_t *dst, const uint32_t (&outp)[NP])
{
    __builtin_memcpy(dst, outp, 4 * NP);
}

template <bool COL_EDGE, bool ROW_EDGE, bool PLANES, int NP, bool LDS_PLANES, bool IN_U8 = false>
__device__ __forceinline__ void march_strip(const StripJob &jb, uint8_t *stage_mem, const uint4 *edge_lut = nullptr)
{
    static_assert(!IN_U8 || NP == 4, "the u8 input form exists for 8 pixels per lane only");
    constexpr int PX = 2 * NP;
    const int H = jb.H, W = jb.W, x0 = jb.x0, ybeg = jb.ybeg, yend = jb.yend;
    const bool full8 = x0 >= 0 && x0 + PX - 1 < W; // all of this lane's pixels are inside the image
    const bool owner = jb.lane >= 1 && jb.lane <= 62 && x0 < W;

    // lane-varying border masks (COL_EDGE only)
    uint32_t fix_l = 0, fix_r[NP] = {};
    unsigned oob = 0; // bit e set: column x0+e is outside the image
    if (COL_EDGE) {
        fix_l = (x0 == 0) ? 0x0000ffffu : 0u; // column 0 = low half of pair 0
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (x0 + 2 * i == W - 1) fix_r[i] |= 0x0000ffffu;
            if (x0 + 2 * i + 1 == W - 1) fix_r[i] |= 0xffff0000u;
        }
#pragma unroll
        for (int e = 0; e < PX; e++)
            if (x0 + e < 0 || x0 + e >= W) oob |= 1u << e;
    }

    auto load_row = [&](int r, uint32_t (&p)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; i++) p[i] = 0u;
        if (ROW_EDGE && (r < 0 || r >= H)) return; // wave-uniform: virtual rows are zero
        if (IN_U8) {
            // p[0..1] receive the eight raw bytes; step() spreads them into s16 pairs when the row is USED, two rows
            // later (expanding them here would make the load's consumer follow it directly: no prefetch left)
            const uint8_t *src8 = jb.fin8 + (size_t)r * W + x0;
            if (!COL_EDGE || full8) {
                __builtin_memcpy(p, src8, 8); // one 8-byte load
            } else if (x0 + PX - 1 >= 0 && x0 < W) {
#pragma unroll
                for (int e = 0; e < PX; e++) {
                    int x = x0 + e;
                    if (x >= 0 && x < W) p[e >> 2] |= (uint32_t)src8[e] << (8 * (e & 3));
                }
            }
            return;
        }
        const int16_t *src = jb.fin + (size_t)r * W + x0;
        if (!COL_EDGE || full8) {
            __builtin_memcpy(p, src, 4 * NP); // one 16-byte (8-byte) load
        } else {
            if (x0 + PX - 1 >= 0 && x0 < W) {
#pragma unroll
                for (int e = 0; e < PX; e++) {
                    int x = x0 + e;
                    if (x >= 0 && x < W) p[e >> 1] |= (uint32_t)(uint16_t)src[e] << (16 * (e & 1));
                }
            }
        }
    };

    // Magnitude of a pixel NMS must skip (outside the image), and the floor every stored magnitude is raised to:
    //   plain kernel:  0.  The reference skips the comparison; comparing against 0 instead gives the same OUTPUT,
    //                  because the only case that differs (mc == 0 "survives" a skipped neighbour) writes 0 either way.
    //   PLANES kernel: min_val - 1 (>= 0).  A pixel is connectable iff mc > max(neighbours, min_val - 1); with every
    //                  magnitude m replaced by m' = max(m, min_val - 1) that is mc' > max(neighbours') -- for
    //                  mc <= min_val - 1 both sides are false, otherwise mc' = mc and the floor under the neighbours
    //                  is the third operand the old v_max3_i32 carried.  One v_max_u16 per pixel instead of a wider
    //                  maximum in each of the four neighbour pairs.
    int skip = PLANES ? jb.lo1 : 0; // in a VGPR on purpose: an SGPR operand makes the maximum a slow-class instruction
    asm volatile("" : "+v"(skip));
    // rotating state (all indices are compile-time after unrolling by 3)
    uint32_t d[3][NP], t[3][NP]; // horizontal difference / smooth of rows r, r-1, r-2
    int M[3][PX + 2];            // magnitudes of rows r-1, r-2, r-3; [0] and [PX+1] are the neighbours' edge pixels
    float cP[3][PX], cQ[3][PX];  // bin discriminants (slot of the row they belong to)
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int i = 0; i < NP; i++) d[a][i] = t[a][i] = 0u;
#pragma unroll
        for (int e = 0; e < PX + 2; e++) M[a][e] = skip;
#pragma unroll
        for (int e = 0; e < PX; e++) cP[a][e] = cQ[a][e] = 0.0f;
    }

    // ---- plane bytes through LDS (the 8-pixel PLANES kernel) ---------------------------------------------------
    // A lane's plane byte belongs to a 64-bit plane word shared with seven other lanes, and consecutive rows of
    // a tile are consecutive words: stored directly, every row costs two byte-store instructions that hit eight
    // or nine tiles, and stores are what this kernel's row loop stalls on.  Instead the wave parks the
    // bytes of its whole segment in LDS (up to 64 rows x 72 bytes per plane; byte column = position in the strip's
    // first plane word + lane) and writes them out once, after the last row, with LANE = ROW: an 8-byte store per
    // complete word column then covers 64 consecutive words of one tile -- 512 contiguous bytes -- and the strip's
    // ragged ends go out as byte pairs (interior strips: the strip starts on an even byte) or single bytes (border
    // strips).  ~20 store instructions per segment, none of them inside the row loop.
    constexpr bool STAGE = LDS_PLANES && PLANES && NP == 4;
    constexpr unsigned kStagePitch = 72, kStageRows = SNM_STAGE_ROWS, kStagePlane = kStageRows * kStagePitch;
    uint8_t *const stage = stage_mem;
    unsigned stage_col = 0;
    int st_o = 0, st_e = 0, st_w0 = 0; // owned byte columns [st_o, st_e) of a staged row; tile column of its first word
    if (STAGE) {
        const int strip_b0 = (x0 - (jb.lane - 1) * PX) >> 3; // byte column of lane 1 (wave-uniform, even)
        const int nb = COL_EDGE ? max(0, min(62, (W >> 3) - strip_b0)) : 62; // bytes owned (lanes 1..nb)
        st_o = strip_b0 & 7;
        st_w0 = strip_b0 >> 3;
        st_e = st_o + nb;
        stage_col = jb.lane == 0 ? 71u : (unsigned)(st_o + jb.lane - 1); // non-owner lanes land in bytes nobody reads
    }
    auto stage_flush_segment = [&]() { // after the segment's last row
        const int y = ybeg + jb.lane;  // lane = row
        const bool row_ok = y < yend;
        const unsigned lrow = (unsigned)jb.lane * kStagePitch;
        const unsigned grow = (unsigned)(y >> 6) * (unsigned)jb.tiles_x * 512u + (unsigned)(y & 63) * 8u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // compiler: the rows' byte writes come first
        const int wf0 = (st_o + 7) >> 3, wf1 = st_e >> 3; // plane words [wf0, wf1) lie entirely inside [st_o, st_e)
        for (int k = wf0; k < wf1; k++) {                 // wave-uniform, at most 8 trips
            if (row_ok) {
                const unsigned g = grow + (unsigned)(st_w0 + k) * 512u;
                *reinterpret_cast<uint64_t *>(jb.pconn + g) = *reinterpret_cast<const uint64_t *>(stage + lrow + 8 * k);
                *reinterpret_cast<uint64_t *>(jb.pstrong + g) =
                    *reinterpret_cast<const uint64_t *>(stage + kStagePlane + lrow + 8 * k);
            }
        }
        auto leftover = [&](int i) { // byte column i (and i + 1 in interior strips, where everything is even)
            if (!row_ok) return;
            const unsigned g = grow + (unsigned)(st_w0 + (i >> 3)) * 512u + (unsigned)(i & 7);
            if (!COL_EDGE) {
                *reinterpret_cast<uint16_t *>(jb.pconn + g) = *reinterpret_cast<const uint16_t *>(stage + lrow + i);
                *reinterpret_cast<uint16_t *>(jb.pstrong + g) =
                    *reinterpret_cast<const uint16_t *>(stage + kStagePlane + lrow + i);
            } else {
                jb.pconn[g] = stage[lrow + i];
                jb.pstrong[g] = stage[kStagePlane + lrow + i];
            }
        };
        constexpr int kStep = COL_EDGE ? 1 : 2;
        const int head_end = min(8 * wf0, st_e), tail_beg = max(8 * wf1, head_end);
        for (int i = st_o; i < head_end; i += kStep) leftover(i);
        for (int i = tail_beg; i < st_e; i += kStep) leftover(i);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    // DEFER: a row's output pixels are parked in NP registers and stored at the TOP of the next row's step, right
    // behind that step's wait for its input row.  The counter of that wait (vmcnt) counts stores as well as loads: a
    // store issued at the end of a row was waited for -- acknowledged by memory -- a few instructions later at the top
    // of the next row, every row; issued behind the wait it has a whole row's time.  (The table-fed kernels also hide
    // their LDS read this way.)
    constexpr bool DEFER = FLT_DEFER_STORE != 0;
    // (No state beside the pixels: step r parks row r-2 when that row belongs to the segment, so what step r finds
    // parked is row r-3 -- a test on scalar registers.)
    uint32_t pend[NP] = {};
    auto park = [&](const uint32_t (&outp)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; q++) pend[q] = outp[q];
    };
    // the instantiations that park: the staged-plane kernels, and the plain kernels away from the column borders (there
    // a lane may own part of a row and stores pixel by pixel, at once)
    constexpr bool PARKS = DEFER && ((PLANES && STAGE) || (!PLANES && !COL_EDGE));
    auto store_pending = [&](int y) { // y: the row that is parked if any row is
        if (PARKS && y >= ybeg && y < yend) {
            if (owner) store_row<NP>(jb.fout + (size_t)y * W + x0, pend);
        }
    };

    // One input row r; PH = (r - rfirst) mod 3 selects the register roles.
    auto step = [&](auto ph, int r, const uint32_t (&praw)[NP]) {
        constexpr int PH = decltype(ph)::value;
        uint32_t p[NP];
        if (IN_U8) {
            // selector bytes: 0-3 pick a byte of the source, 0x0c yields 0x00
            p[0] = __builtin_amdgcn_perm(praw[0], praw[0], 0x0c010c00u);
            p[1] = __builtin_amdgcn_perm(praw[0], praw[0], 0x0c030c02u);
            p[NP > 2 ? 2 : 0] = __builtin_amdgcn_perm(praw[1], praw[1], 0x0c010c00u);
            p[NP > 2 ? 3 : 1] = __builtin_amdgcn_perm(praw[1], praw[1], 0x0c030c02u);
        } else {
#pragma unroll
            for (int i = 0; i < NP; i++) p[i] = praw[i];
        }
        if (DEFER) { // the row's pixels have arrived (pinned: the wait stays in front of the store)
#pragma unroll
            for (int i = 0; i < NP; i++) asm volatile("" : "+v"(p[i]));
            __builtin_amdgcn_sched_barrier(0);
            store_pending(r - 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int k2 = PH % 3, k1 = (PH + 2) % 3, k0 = (PH + 1) % 3; // d/t of rows r, r-1, r-2
        constexpr int m2 = PH % 3, m1 = (PH + 2) % 3, m0 = (PH + 1) % 3; // M and bins of rows r-1, r-2, r-3

        // ---- horizontal step for row r ---------------------------------------------------------
        {
            const uint32_t lp = from_left(p[NP - 1]), rp = from_right(p[0]);
            uint32_t sh[NP + 1]; // sh[i] = (pixel 2i-1, pixel 2i)
            sh[0] = __builtin_amdgcn_alignbit(p[0], lp, 16);
#pragma unroll
            for (int i = 1; i < NP; i++) sh[i] = __builtin_amdgcn_alignbit(p[i], p[i - 1], 16);
            sh[NP] = __builtin_amdgcn_alignbit(rp, p[NP - 1], 16);
#pragma unroll
            for (int i = 0; i < NP; i++) {
                d[k2][i] = pk_sub(sh[i + 1], sh[i]);
                t[k2][i] = pk_mad2(p[i], pk_add(sh[i], sh[i + 1]));
            }
            if (COL_EDGE) {
                d[k2][0] = pk_sub(d[k2][0], p[0] & fix_l); // clamp at column 0
#pragma unroll
                for (int i = 0; i < NP; i++) d[k2][i] = pk_add(d[k2][i], p[i] & fix_r[i]); // ... and at column W-1
            }
        }

        // ---- gradient, magnitude and bin discriminants for row y1 = r-1 ------------------------
        const int y1 = r - 1;
        if (!ROW_EDGE || (y1 >= 0 && y1 < H)) {
            uint32_t gy[NP];
            if (ROW_EDGE && (y1 == 0 || y1 == H - 1)) { // row clamp: t[-1] := t[0], t[H] := t[H-1]
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const uint32_t tu = (y1 == 0) ? t[k1][i] : t[k0][i];
                    const uint32_t td = (y1 == H - 1) ? t[k1][i] : t[k2][i];
                    gy[i] = pk_sub(td, tu);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NP; i++) gy[i] = pk_sub(t[k2][i], t[k0][i]);
            }
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const uint32_t gx = pk_add(pk_mad2(d[k1][i], d[k0][i]), d[k2][i]);
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const int e = 2 * i + hf;
                    const float fx = hf ? (float)((int)gx >> 16) : (float)(int)(short)(gx & 0xffffu);
                    const float fy = hf ? (float)((int)gy[i] >> 16) : (float)(int)(short)(gy[i] & 0xffffu);
                    // all exact in f32 (integers and halves below 2^24):
                    //   Ah = gx^2 + 1/2,  nh = gx^2 + gy^2 + 1/2,  P = gx*gy,  Q = Ah - nh/2 = (gx^2-gy^2)/2 + 1/4
                    const float Ah = __fmaf_rn(fx, fx, 0.5f);
                    const float nh = __fmaf_rn(fy, fy, Ah);
                    const int mag = (int)__builtin_amdgcn_sqrtf(nh); // floor(sqrt(n)), see magnitude_d8
                    M[m2][e + 1] = PLANES ? max_u16(mag, skip) : mag;
                    cP[m2][e] = __fmul_rn(fx, fy);
                    cQ[m2][e] = __fmaf_rn(nh, -0.5f, Ah);
                }
            }
            if (COL_EDGE) { // columns outside the image never win a comparison
#pragma unroll
                for (int e = 0; e < PX; e++)
                    if (oob & (1u << e)) M[m2][e + 1] = skip;
            }
        } else {
#pragma unroll
            for (int e = 0; e < PX; e++) M[m2][e + 1] = skip; // rows outside the image are skipped by NMS
        }
        // edge pixels of the neighbouring lanes (lanes 0 and 63 get 0 here -- below the PLANES floor, harmless: they
        // are halo lanes whose own NMS results are never stored, and their neighbours only read M[PX] / M[1] from them)
        M[m2][0] = (int)from_left((uint32_t)M[m2][PX]);
        M[m2][PX + 1] = (int)from_right((uint32_t)M[m2][1]);

        // ---- NMS for row y2 = r-2 ---------------------------------------------------------------
        const int y2 = r - 2;
        if (y2 >= ybeg && y2 < yend) {
            uint32_t outp[NP] = {}; // suppressed magnitudes as s16 pairs (pixel 2i, pixel 2i+1)
            unsigned cbits = 0, sbits = 0;
#pragma unroll
            for (int ee = 0; ee < PX; ee++) {
                const int e = PLANES ? PX - 1 - ee : ee; // planes: last pixel first, so that pixel e ends up in bit e
                const int c = e + 1;
                const int mc = M[m1][c];
                auto nmax = [&](int a, int b) { return max_u16(a, b); };
                const int n0 = nmax(M[m1][c - 1], M[m1][c + 1]);
                const int n90 = nmax(M[m0][c], M[m2][c]);
                const int n45 = nmax(M[m0][c + 1], M[m2][c - 1]);  // up-right, down-left
                const int n135 = nmax(M[m0][c - 1], M[m2][c + 1]); // up-left, down-right
                // Bin from two sign tests.  With X = gx^2-gy^2 and Y = 2 gx gy (the doubled angle) the bins are
                // the quadrants of (X+Y, X-Y):  0: both >= 0,  90: both < 0,  45: X+Y >= 0 > X-Y,  135: the rest.
                // P and X/2 are multiples of 1/2 and Q = X/2 + 1/4, so  X-Y >= 0  <=>  P <= Q  and
                // X+Y >= 0  <=>  P > -Q  exactly (same rule as angle_bin_d8, checked exhaustively on the device).
                const float P = cP[m1][e], Q = cQ[m1][e];
                const bool v_ge0 = P <= Q;  // X - Y >= 0
                const bool u_ge0 = P > -Q;  // X + Y >= 0
                const int na = v_ge0 ? n0 : n45;
                const int nb = v_ge0 ? n135 : n90;
                const int nsel = u_ge0 ? na : nb;
                if (PLANES) {
                    // cbits: survived NMS and mc >= min_val;  sbits: mc >= max(min_val, max_val), ANDed with cbits below
                    push_gt2(cbits, sbits, mc, nsel, jb.hi1);
                } else if ((e & 1) == 0) {
                    outp[e >> 1] = (uint32_t)((mc > nsel) ? mc : 0); // magnitudes are < 65536: upper half zero
                } else {
                    keep_gt_hi(outp[e >> 1], mc, nsel, 0);
                }
            }
            if (PLANES) {
                // Provisional edge map: strong pixels already carry their final value, everything else 0; the
                // propagation sweeps add the weak pixels they promote (this replaces the finalize pass).
                unsigned sb = sbits & cbits;
                if (STAGE) {
                    // the 8 strong bits -> 8 s16 pixels: one 16-byte read of a 256-entry LDS table on the otherwise
                    // idle LDS port instead of 13 VALU instructions (shift, mask, multiply per pixel pair)
                    const uint4 v = edge_lut[sb];
                    outp[0] = v.x;
                    outp[1] = v.y;
                    outp[NP > 2 ? 2 : 0] = v.z;
                    outp[NP > 2 ? 3 : 1] = v.w;
                } else {
                    const uint32_t tb = sb | (sb << 15); // bit 2k -> bit 0, bit 2k+1 -> bit 16 of pair k
#pragma unroll
                    for (int i = 0; i < NP; i++)
                        outp[i] = __umul24((tb >> (2 * i)) & 0x00010001u, (uint32_t)jb.edge_value);
                }
                if (NP == 2) {
                    // 4 pixels are half a plane byte: odd lanes hold the low nibble (x0 % 8 == 0) and fetch the high
                    // one from their right neighbour; lanes 1..62 pair up exactly (1,2) .. (61,62)
                    cbits |= from_right(cbits) << 4;
                    sb |= from_right(sb) << 4;
                }
                if (STAGE) {
                    // plane bytes: parked in LDS until the segment is done (see stage_flush_segment)
                    const unsigned rowoff = (unsigned)(y2 - ybeg) * kStagePitch;
                    stage[rowoff + stage_col] = (uint8_t)cbits; // halo lanes write pad columns nobody reads
                    stage[kStagePlane + rowoff + stage_col] = (uint8_t)sb;
                    if (DEFER)
                        park(outp);
                    else if (owner)
                        store_row<NP>(jb.fout + (size_t)y2 * W + x0, outp);
                } else if (owner) { // W % 8 == 0: an owner lane's pixels are all inside the image
                    if (NP == 4 || (jb.lane & 1)) {
                        const unsigned bx = (unsigned)x0 >> 3;
                        const unsigned off = ((unsigned)(y2 >> 6) * (unsigned)jb.tiles_x + (bx >> 3)) * 512u +
                                             (unsigned)(y2 & 63) * 8u + (bx & 7u);
                        jb.pconn[off] = (uint8_t)cbits;
                        jb.pstrong[off] = (uint8_t)sb;
                    }
                    store_row<NP>(jb.fout + (size_t)y2 * W + x0, outp);
                }
            } else if (DEFER && !COL_EDGE) {
                park(outp); // interior strips: every owner lane stores whole rows
            } else if (owner) {
                int16_t *dst = jb.fout + (size_t)y2 * W + x0;
                if (!COL_EDGE || full8) {
                    store_row<NP>(dst, outp);
                } else {
#pragma unroll
                    for (int e = 0; e < PX; e++)
                        if (x0 + e < W) dst[e] = (int16_t)(outp[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
    };

    // Rows ybeg-2 .. yend+1 are needed; the count is rounded up to a multiple of 3 so that the loop
    // body has no early exit (the extra rows produce no output: y2 >= yend).  For ROW_EDGE=false the
    // dispatcher guarantees that even the extra rows and the two prefetched ones lie inside the image.
    const int rfirst = ybeg - 2;
    const int rlast = rfirst + 3 * ((yend + 1 - rfirst + 3) / 3) - 1;
    uint32_t pa[NP], pb[NP], pc[NP]; // software prefetch: rows r, r+1, r+2 -- renamed, never copied
    load_row(rfirst, pa);
    load_row(rfirst + 1, pb);
    for (int r = rfirst; r <= rlast; r += 3) {
        load_row(r + 2, pc);
        step(IC<0>{}, r, pa);
        load_row(r + 3, pa);
        step(IC<1>{}, r + 1, pb);
        load_row(r + 4, pb);
        step(IC<2>{}, r + 2, pc);
    }
    store_pending(rlast - 2);
    // Every row below yend has been staged by now; the row loop's state is dead here, so the flush's own
    // registers come for free (called from inside the loop it pushed the kernel from 160 to 215 VGPRs).
    if (STAGE) stage_flush_segment();
}

// ---- f32 marching variant (round 3; the default) ----------------------------------------------------------------
// Same strip / segment / lane geometry, same loads, stores and plane staging as march_strip above; what differs is the
// arithmetic between the row load and the row store, rebuilt around what a vector instruction COSTS on gfx950
// (tools/valu_issue_bench.hip, DESIGN.md "instruction classes"): plain f32 add/mul/fma, 32-bit integer add/sub,
// and/or/xor, v_ashrrev_i32 and the 16-bit max issue every ~2.3 cycles per SIMD; everything packed (v_pk_*), every
// SDWA / DPP form, v_cvt_*, v_cmp_*, v_cndmask, v_addc and the VOP3-only integer ops take ~4.2.  The packed-i16 kernel
// spends ~20 of its 28.7 instructions per pixel in the slow class; this one ~7 of 30:
//   * one v_cvt_f32_ubyteN per pixel (the smoothed plane lies in [0,255]: byte 0 / byte 2 of an s16 pair, or the four
//     bytes of a dword in the u8 form), then every Sobel step is an exact f32 add or fma on integers < 2^11:
//       vertical first (no lane crossing):  a = s[y-1] + 2 s[y] + s[y+1],  b = s[y+1] - s[y-1]
//       horizontal:                         gx = a[x+1] - a[x-1],          gy = b[x-1] + 2 b[x] + b[x+1]
//     (the neighbouring lanes' a / b arrive through DPP shifts folded into the four adds that need them);
//   * magnitude: floor(sqrt(n)) = low 16 bits of (sqrt(n + 1/2) + (2^23 - 1/2)) -- the float add puts the integer part
//     into the mantissa (the sum lies in [2^23, 2^24), where one ulp is 1, and sqrt(n + 1/2) is never within rounding
//     of an integer: DESIGN.md 4.3), the 16-bit maximum that raises the magnitude to the threshold floor discards the
//     exponent bits: no v_cvt_i32_f32;
//   * bins: the signs of Q - P and Q + P (never zero: Q has a fractional part of 1/4) travel to the NMS step as the two
//     floats themselves; there v_ashrrev_i32 turns each into a lane mask IN A VGPR and three v_bfi_b32 select
//     the neighbour maximum -- no v_cmp, no v_cndmask, no SGPR lane masks, no s_nop hazard pads;
//   * classify: (nsel - mc) and (hi1 - mc) are negative iff the pixel is connectable / strong; v_alignbit_b32 shifts
//     each sign bit into its plane byte (one instruction per bit instead of v_cmp + v_addc).
// State per lane: 16 floats of input rows, 30 magnitudes, 16 bin floats, 12 prefetch registers: under 128 VGPRs, so a
// SIMD holds four waves instead of three.
__device__ __forceinline__ float f_from_left(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float f_from_right(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
// (m & a) | (~m & b) as ONE v_bfi_b32 (the compiler's pattern match is not guaranteed for constants it can fold)
__device__ __forceinline__ int bfi(int m, int a, int b)
{
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
// acc = 2 * acc + (d < 0): the sign bit of d shifted into acc
__device__ __forceinline__ unsigned push_sign(unsigned acc, int d)
{
    return __builtin_amdgcn_alignbit(acc, (unsigned)d, 31);
}

template <bool COL_EDGE, bool ROW_EDGE, bool PLANES, bool LDS_PLANES, bool IN_U8>
__device__ __forceinline__ void fmarch_strip(const StripJob &jb, uint8_t *stage_mem, const uint4 *edge_lut = nullptr)
{
    constexpr int NP = 4, PX = 8;
    const int H = jb.H, W = jb.W, x0 = jb.x0, ybeg = jb.ybeg, yend = jb.yend;
    const bool full8 = x0 >= 0 && x0 + PX - 1 < W; // all of this lane's pixels are inside the image
    const bool owner = jb.lane >= 1 && jb.lane <= 62 && x0 < W;

    // lane-varying border constants (COL_EDGE only).  Out-of-image columns load as zero, so a = b = 0 there:
    // "columns dropped" (gy) needs nothing, "column clamp" (gx) is gx -= a[0] at column 0 and gx += a[e] at column W-1.
    // elast = index of column W-1 relative to this lane's pixel 0 (>= 7: every pixel inside; -1: every pixel outside,
    // also for the halo lane left of column 0): "pixel e is inside" is the sign of e - 1 - elast.
    const float fix_l = (COL_EDGE && x0 == 0) ? 1.0f : 0.0f; // column 0 is always pixel 0 of a lane (x0 % 8 == 0)
    const int elast = (x0 < 0) ? -1 : W - 1 - x0;
    // The nine masks live in registers for the whole segment (border strips have the room: 115 of 128 VGPRs);
    // recomputing them per row cost two instructions per pixel in a quarter of a 4K frame's waves.
    int in_m[PX + 1];
#pragma unroll
    for (int e = 0; e <= PX; e++) in_m[e] = COL_EDGE ? (e - 1 - elast) >> 31 : -1;
    auto inside = [&](int e) { return in_m[e]; }; // all ones iff x0 + e lies inside the image

    // Lane offset (in pixels) of the interior strips' row loads and stores, passed through an empty asm at every use so
    // that the optimiser cannot fold it into loop-invariant 64-bit VGPR pointers (as in the Gaussian: the addresses stay
    // "uniform row base + 32-bit lane offset").  Interior strips have x0 >= 0.
    uint32_t lane_off = (uint32_t)x0;
    auto load_row = [&](int r, uint32_t (&p)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; i++) p[i] = 0u;
        if (ROW_EDGE && (r < 0 || r >= H)) return; // wave-uniform: virtual rows are zero
        if (IN_U8) {
            if (!COL_EDGE) { // uniform row base (SGPR pair) + 32-bit lane offset: no 64-bit address arithmetic per lane
                asm volatile("" : "+v"(lane_off));
                __builtin_memcpy(p, jb.fin8 + (size_t)r * W + lane_off, 8);
                return;
            }
            const uint8_t *src8 = jb.fin8 + (size_t)r * W + x0;
            if (!COL_EDGE || full8) {
                __builtin_memcpy(p, src8, 8); // one 8-byte load
            } else if (x0 + PX - 1 >= 0 && x0 < W) {
#pragma unroll
                for (int e = 0; e < PX; e++) {
                    int x = x0 + e;
                    if (x >= 0 && x < W) p[e >> 2] |= (uint32_t)src8[e] << (8 * (e & 3));
                }
            }
            return;
        }
        if (!COL_EDGE) {
            asm volatile("" : "+v"(lane_off));
            __builtin_memcpy(p, reinterpret_cast<const char *>(jb.fin + (size_t)r * W) + 2 * lane_off, 4 * NP);
            return;
        }
        const int16_t *src = jb.fin + (size_t)r * W + x0;
        if (!COL_EDGE || full8) {
            __builtin_memcpy(p, src, 4 * NP); // one 16-byte load
        } else if (x0 + PX - 1 >= 0 && x0 < W) {
#pragma unroll
            for (int e = 0; e < PX; e++) {
                int x = x0 + e;
                if (x >= 0 && x < W) p[e >> 1] |= (uint32_t)(uint16_t)src[e] << (16 * (e & 1));
            }
        }
    };

    // floor of every stored magnitude and the value of a pixel NMS must skip: see march_strip
    int skip = PLANES ? jb.lo1 : 0;
    int hi1v = PLANES ? jb.hi1 : 0; // VGPR copies: an SGPR operand would make the instruction slow class
    asm volatile("" : "+v"(skip), "+v"(hi1v));

    float F[3][PX];      // input rows r, r-1, r-2 as floats
    int M[3][PX + 2];    // magnitudes of rows r-1, r-2, r-3; [0] and [PX+1] are the neighbours' edge pixels
    int cz[3][PX];       // bin carriers (signs of Q - P and Q + P) of the rows r-1 (being produced) and r-2 (consumed)
#if FLT_CARRIERS != 1
    int cy[3][PX];
#endif
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int e = 0; e < PX; e++) {
            F[a][e] = 0.0f;
            cz[a][e] = 0;
        }
#pragma unroll
        for (int e = 0; e < PX + 2; e++) M[a][e] = skip;
    }

    // ---- plane bytes through LDS: identical to march_strip ---------------------------------------------------------
    constexpr bool STAGE = LDS_PLANES && PLANES;
    constexpr unsigned kStagePitch = 72, kStageRows = SNM_STAGE_ROWS, kStagePlane = kStageRows * kStagePitch;
    uint8_t *const stage = stage_mem;
    unsigned stage_col = 0;
    int st_o = 0, st_e = 0, st_w0 = 0;
    if (STAGE) {
        const int strip_b0 = (x0 - (jb.lane - 1) * PX) >> 3;
        const int nb = COL_EDGE ? max(0, min(62, (W >> 3) - strip_b0)) : 62;
        st_o = strip_b0 & 7;
        st_w0 = strip_b0 >> 3;
        st_e = st_o + nb;
        stage_col = jb.lane == 0 ? 71u : (unsigned)(st_o + jb.lane - 1);
    }
    auto stage_flush_segment = [&]() {
        const int y = ybeg + jb.lane; // lane = row
        const bool row_ok = y < yend;
        const unsigned lrow = (unsigned)jb.lane * kStagePitch;
        const unsigned grow = (unsigned)(y >> 6) * (unsigned)jb.tiles_x * 512u + (unsigned)(y & 63) * 8u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int wf0 = (st_o + 7) >> 3, wf1 = st_e >> 3;
        for (int k = wf0; k < wf1; k++) {
            if (row_ok) {
                const unsigned g = grow + (unsigned)(st_w0 + k) * 512u;
                *reinterpret_cast<uint64_t *>(jb.pconn + g) = *reinterpret_cast<const uint64_t *>(stage + lrow + 8 * k);
                *reinterpret_cast<uint64_t *>(jb.pstrong + g) =
                    *reinterpret_cast<const uint64_t *>(stage + kStagePlane + lrow + 8 * k);
            }
        }
        auto leftover = [&](int i) {
            if (!row_ok) return;
            const unsigned g = grow + (unsigned)(st_w0 + (i >> 3)) * 512u + (unsigned)(i & 7);
            if (!COL_EDGE) {
                *reinterpret_cast<uint16_t *>(jb.pconn + g) = *reinterpret_cast<const uint16_t *>(stage + lrow + i);
                *reinterpret_cast<uint16_t *>(jb.pstrong + g) =
                    *reinterpret_cast<const uint16_t *>(stage + kStagePlane + lrow + i);
            } else {
                jb.pconn[g] = stage[lrow + i];
                jb.pstrong[g] = stage[kStagePlane + lrow + i];
            }
        };
        constexpr int kStep = COL_EDGE ? 1 : 2;
        const int head_end = min(8 * wf0, st_e), tail_beg = max(8 * wf1, head_end);
        for (int i = st_o; i < head_end; i += kStep) leftover(i);
        for (int i = tail_beg; i < st_e; i += kStep) leftover(i);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    // DEFER: see march_strip -- a row's output pixels are parked and stored at the top of the next row's step.
    constexpr bool DEFER = FLT_DEFER_STORE != 0;
    // (No state beside the pixels: step r parks row r-2 when that row belongs to the segment, so what step r finds
    // parked is row r-3 -- a test on scalar registers.)
    uint32_t pend[NP] = {};
    auto park = [&](const uint32_t (&outp)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; q++) pend[q] = outp[q];
    };
    // the instantiations that park: the staged-plane kernels, and the plain kernels away from the column borders (there
    // a lane may own part of a row and stores pixel by pixel, at once)
    constexpr bool PARKS = DEFER && ((PLANES && STAGE) || (!PLANES && !COL_EDGE));
    auto store_pending = [&](int y) { // y: the row that is parked if any row is
        if (PARKS && y >= ybeg && y < yend) {
            if (!COL_EDGE) {
                asm volatile("" : "+v"(lane_off));
                if (owner)
                    store_row<NP>(reinterpret_cast<int16_t *>(reinterpret_cast<char *>(jb.fout + (size_t)y * W) +
                                                              2 * lane_off), pend);
            } else if (owner) {
                store_row<NP>(jb.fout + (size_t)y * W + x0, pend);
            }
        }
    };

    // One input row r; PH = (r - rfirst) mod 3 selects the register roles.
    auto step = [&](auto ph, int r, uint32_t (&praw)[NP]) {
        constexpr int PH = decltype(ph)::value;
        constexpr int k2 = PH % 3, k1 = (PH + 2) % 3, k0 = (PH + 1) % 3; // F of rows r, r-1, r-2
        constexpr int m2 = PH % 3, m1 = (PH + 2) % 3, m0 = (PH + 1) % 3; // M of rows r-1, r-2, r-3; carriers alike

        // ---- row r as floats (exact: the plane's values are 0..255) ----------------------------------------------
#pragma unroll
        for (int e = 0; e < PX; e++) {
            const uint32_t w = IN_U8 ? praw[e >> 2] >> (8 * (e & 3)) : praw[e >> 1] >> (16 * (e & 1));
            F[k2][e] = (float)(w & 0xffu); // v_cvt_f32_ubyteN
        }
        // The previous row's edge-map pixels (DEFER) go out HERE, right behind the wait for this row's pixels: the
        // counter that wait uses (vmcnt) also counts stores, so a store issued late in a row is waited for at the top
        // of the next one; issued here it has a whole row's time to be acknowledged.  (The conversions are pinned in
        // front of it -- sunk behind the store, their wait would cover the row load issued a moment ago as well.)
        if (DEFER) {
#pragma unroll
            for (int e = 0; e < PX; e++) asm volatile("" : "+v"(F[k2][e]));
            FLT_SCHED_FENCE();
            store_pending(r - 3);
            FLT_SCHED_FENCE();
        }
        // EARLY: the row's raw pixels are converted, so their registers can take row r+3 at once -- three rows in
        // flight with the same three buffers (the plain loop reloads a buffer at the top of the NEXT step: two rows).
        if (FLT_EARLY_RELOAD) {
#pragma unroll
            for (int e = 0; e < PX; e++) asm volatile("" : "+v"(F[k2][e]));
            load_row(r + 3, praw);
            FLT_SCHED_FENCE();
        }

        // ---- gradient, magnitude and bin carriers for row y1 = r-1 -----------------------------------------------
        const int y1 = r - 1;
        if (!ROW_EDGE || (y1 >= 0 && y1 < H)) {
            float a[PX], b[PX];
#pragma unroll
            for (int e = 0; e < PX; e++) {
                // gx: rows outside the image are dropped (their floats are zero); gy: the row is clamped
                a[e] = __fadd_rn(__fmaf_rn(2.0f, F[k1][e], F[k0][e]), F[k2][e]);
                const float up = (ROW_EDGE && y1 == 0) ? F[k1][e] : F[k0][e];
                const float dn = (ROW_EDGE && y1 == H - 1) ? F[k1][e] : F[k2][e];
                b[e] = __fsub_rn(dn, up);
            }
            const float a_l = f_from_left(a[PX - 1]), a_r = f_from_right(a[0]);
            const float b_l = f_from_left(b[PX - 1]), b_r = f_from_right(b[0]);
            FLT_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < PX; e++) {
                if (e && e % FLT_GROUP == 0) FLT_SCHED_FENCE();
                const float al = e ? a[e - 1] : a_l, ar = e < PX - 1 ? a[e + 1] : a_r;
                const float bl = e ? b[e - 1] : b_l, br = e < PX - 1 ? b[e + 1] : b_r;
                float gx = __fsub_rn(ar, al);
                if (COL_EDGE) {
                    if (e == 0) gx = __fmaf_rn(-fix_l, a[0], gx); // column clamp at 0 ...
                    // ... and at W-1: pixel e is the last column iff it is inside and pixel e + 1 is not
                    gx = __fadd_rn(gx, __int_as_float(__float_as_int(a[e]) & (inside(e) ^ inside(e + 1))));
                }
                const float gy = __fadd_rn(__fmaf_rn(2.0f, b[e], bl), br);
                // all exact in f32 (integers and quarters below 2^24):
                //   Ah = gx^2 + 1/2,  nh = gx^2 + gy^2 + 1/2,  Q = (gx^2 - gy^2)/2 + 1/4,  P = gx gy
                const float Ah = __fmaf_rn(gx, gx, 0.5f);
                const float nh = __fmaf_rn(gy, gy, Ah);
                const float mb = __fadd_rn(__builtin_amdgcn_sqrtf(nh), 8388607.5f); // 2^23 + floor(sqrt(n))
                int m = __float_as_int(mb);
                if (COL_EDGE) m &= inside(e); // out-of-image magnitudes become the floor
                M[m2][e + 1] = PLANES ? max_u16(m, skip) : (m & 0xffff);
                const float Q = __fmaf_rn(nh, -0.5f, Ah);
                const float qm = __fmaf_rn(-gx, gy, Q); // Q - P < 0  <=>  not (X - Y >= 0)
                const float qp = __fmaf_rn(gx, gy, Q);  // Q + P < 0  <=>  not (X + Y >= 0)
#if FLT_CARRIERS == 1
                // one carrier per pixel: byte 3 = sign of Q - P, bytes 0-2 = sign of Q + P, eight copies each
                // (v_perm_b32 selectors 11 / 9 replicate bit 31 of the first / second source)
                cz[m2][e] = (int)__builtin_amdgcn_perm(__float_as_uint(qm), __float_as_uint(qp), 0x0b090909u);
                // Materialise the carrier HERE: its only use is in the next row's step, and the optimiser otherwise
                // sinks its computation behind that row's NMS branch and keeps gx, gy, Ah and nh alive instead.
                asm volatile("" : "+v"(cz[m2][e]));
#else
                cz[m2][e] = __float_as_int(qm);
                cy[m2][e] = __float_as_int(qp);
                asm volatile("" : "+v"(cz[m2][e]), "+v"(cy[m2][e]));
#endif
            }
        } else {
#pragma unroll
            for (int e = 0; e < PX; e++) M[m2][e + 1] = skip; // rows outside the image are skipped by NMS
        }
        M[m2][0] = (int)from_left((uint32_t)M[m2][PX]);
        M[m2][PX + 1] = (int)from_right((uint32_t)M[m2][1]);
        FLT_SCHED_FENCE();

        // ---- NMS for row y2 = r-2 --------------------------------------------------------------------------------
        const int y2 = r - 2;
        if (y2 >= ybeg && y2 < yend) {
            uint32_t outp[NP] = {};
            unsigned cacc = 0, sacc = 0;
            int keep[PX];
#pragma unroll
            for (int ee = 0; ee < PX; ee++) {
                if (ee && ee % FLT_GROUP == 0) FLT_SCHED_FENCE();
                const int e = PX - 1 - ee; // last pixel first: pixel e ends up in bit e
                const int c = e + 1;
                const int mc = M[m1][c];
                const int n0 = max_u16(M[m1][c - 1], M[m1][c + 1]);
                const int n90 = max_u16(M[m0][c], M[m2][c]);
                const int n45 = max_u16(M[m0][c + 1], M[m2][c - 1]);  // up-right, down-left
                const int n135 = max_u16(M[m0][c - 1], M[m2][c + 1]); // up-left, down-right
                // bins are the quadrants of (X + Y, X - Y), see march_strip: 0: both >= 0, 90: both < 0,
                // 45: X + Y >= 0 > X - Y, 135: the rest.  mv / mu are all ones where the test FAILS.
                const int mv = cz[m1][e] >> 31; // all 32 bits
#if FLT_CARRIERS == 1
                const int mu = cz[m1][e];       // its low 16 bits are the mask; the selected values are 16 bits wide
#else
                const int mu = cy[m1][e] >> 31;
#endif
                const int na = bfi(mv, n45, n0);   // X - Y >= 0 ? n0 : n45
                const int nb = bfi(mv, n90, n135); // X - Y >= 0 ? n135 : n90
                const int nsel = bfi(mu, nb, na);  // X + Y >= 0 ? na : nb
                const int dc = nsel - mc;          // < 0  <=>  mc > nsel
                if (PLANES) {
                    cacc = push_sign(cacc, dc);
                    sacc = push_sign(sacc, hi1v - mc); // < 0  <=>  mc > hi1; ANDed with cacc below
                } else {
                    keep[e] = mc & (dc >> 31);
                }
            }
            if (PLANES) {
                const unsigned cbits = cacc & 0xffu, sb = sacc & cbits;
                if (STAGE) {
                    const uint4 v = edge_lut[sb];
                    outp[0] = v.x;
                    outp[1] = v.y;
                    outp[2] = v.z;
                    outp[3] = v.w;
                    const unsigned rowoff = (unsigned)(y2 - ybeg) * kStagePitch;
                    stage[rowoff + stage_col] = (uint8_t)cbits; // halo lanes write pad columns nobody reads
                    stage[kStagePlane + rowoff + stage_col] = (uint8_t)sb;
                    if (DEFER)
                        park(outp);
                    else if (owner)
                        store_row<NP>(jb.fout + (size_t)y2 * W + x0, outp);
                } else {
                    const uint32_t tb = sb | (sb << 15);
#pragma unroll
                    for (int i = 0; i < NP; i++)
                        outp[i] = __umul24((tb >> (2 * i)) & 0x00010001u, (uint32_t)jb.edge_value);
                    if (owner) { // W % 8 == 0: an owner lane's pixels are all inside the image
                        const unsigned bx = (unsigned)x0 >> 3;
                        const unsigned off = ((unsigned)(y2 >> 6) * (unsigned)jb.tiles_x + (bx >> 3)) * 512u +
                                             (unsigned)(y2 & 63) * 8u + (bx & 7u);
                        jb.pconn[off] = (uint8_t)cbits;
                        jb.pstrong[off] = (uint8_t)sb;
                        store_row<NP>(jb.fout + (size_t)y2 * W + x0, outp);
                    }
                }
            } else if (DEFER && !COL_EDGE) {
#pragma unroll
                for (int i = 0; i < NP; i++) outp[i] = (uint32_t)keep[2 * i] | ((uint32_t)keep[2 * i + 1] << 16);
                park(outp); // interior strips: every owner lane stores whole rows
            } else if (owner) {
#pragma unroll
                for (int i = 0; i < NP; i++) outp[i] = (uint32_t)keep[2 * i] | ((uint32_t)keep[2 * i + 1] << 16);
                int16_t *dst = jb.fout + (size_t)y2 * W + x0;
                if (!COL_EDGE || full8) {
                    store_row<NP>(dst, outp);
                } else {
#pragma unroll
                    for (int e = 0; e < PX; e++)
                        if (x0 + e < W) dst[e] = (int16_t)keep[e];
                }
            }
        }
    };

    const int rfirst = ybeg - 2;
    const int rlast = rfirst + 3 * ((yend + 1 - rfirst + 3) / 3) - 1;
    uint32_t pa[NP], pb[NP], pc[NP];
    load_row(rfirst, pa);
    load_row(rfirst + 1, pb);
    if (FLT_EARLY_RELOAD) {
        load_row(rfirst + 2, pc);
        for (int r = rfirst; r <= rlast; r += 3) {
            step(IC<0>{}, r, pa);
            step(IC<1>{}, r + 1, pb);
            step(IC<2>{}, r + 2, pc);
        }
    } else {
        for (int r = rfirst; r <= rlast; r += 3) {
            load_row(r + 2, pc);
            step(IC<0>{}, r, pa);
            load_row(r + 3, pa);
            step(IC<1>{}, r + 1, pb);
            load_row(r + 4, pb);
            step(IC<2>{}, r + 2, pc);
        }
    }
    store_pending(rlast - 2);
    if (STAGE) stage_flush_segment();
}

} // namespace

struct PlaneArgs { // PLANES instantiation only
    uint8_t *conn, *strong;
    int tiles_x, tiles_y, lo1, hi1, edge_value;
};

// FLT: the f32 marching arithmetic (fmarch_strip, 8 pixels per lane only) instead of the packed-i16 one.
template <bool PLANES, int NP, bool LDS_PLANES, bool IN_U8 = false, bool FLT = false>
__global__ __launch_bounds__(SNM_WPB * 64) __attribute__((amdgpu_waves_per_eu(FLT ? FLT_WAVES : 1)))
void sobel_nms_march_kernel(const void *__restrict__ in,
                                                                       int16_t *__restrict__ out, int H, int W,
                                                                       int n_strips, int n_segs, int seg_rows,
                                                                       int total_waves, PlaneArgs pl)
{
    // readfirstlane tells the compiler what it cannot prove: everything derived from the wave index is
    // wave-uniform, so rows, segments and border tests live in SGPRs and branch with s_cbranch.
    // per-wave staging area of the plane bytes (LDS_PLANES kernels only): 2 planes x 64 rows x 72 bytes
    __shared__ __attribute__((aligned(8))) uint8_t stage_lds[LDS_PLANES ? SNM_WPB * 2 * SNM_STAGE_ROWS * 72 : 8];
    uint8_t *stage_mem = stage_lds + (LDS_PLANES ? (threadIdx.x >> 6) * (2 * SNM_STAGE_ROWS * 72) : 0);
    // edge-map table of the LDS_PLANES kernel: entry b = the eight s16 pixels of strong-bit byte b
    __shared__ uint4 edge_lut[LDS_PLANES ? 256 : 1];
    if (LDS_PLANES) {
        static_assert(SNM_WPB * 64 == 256, "one table entry per thread");
        const unsigned b = threadIdx.x, ev = (unsigned)pl.edge_value;
        uint4 v;
        v.x = ((b >> 0) & 1u) * ev | (((b >> 1) & 1u) * ev) << 16;
        v.y = ((b >> 2) & 1u) * ev | (((b >> 3) & 1u) * ev) << 16;
        v.z = ((b >> 4) & 1u) * ev | (((b >> 5) & 1u) * ev) << 16;
        v.w = ((b >> 6) & 1u) * ev | (((b >> 7) & 1u) * ev) << 16;
        edge_lut[b] = v;
        __syncthreads(); // the kernel's only workgroup barrier; before any wave can leave
    }
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): remapped so that every XCD gets a
    // contiguous range of (frame, segment, strip) jobs, the rows and columns neighbouring jobs share are read
    // from HBM once per XCD range instead of once per job.  (Bijective for any grid size.)
    unsigned bid = blockIdx.x;
    {
        const unsigned q = gridDim.x / 8u, r = gridDim.x % 8u, xcd = bid % 8u;
        bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + bid / 8u;
    }
    const int wave = __builtin_amdgcn_readfirstlane(bid * SNM_WPB + (threadIdx.x >> 6));
    if (wave >= total_waves) return;
    const int f = wave / (n_strips * n_segs);
    const MarchCell cell = march_cell_of(wave - f * (n_strips * n_segs), n_segs, n_strips); // border cells first
    const int s = cell.strip, g = cell.seg;
    StripJob jb;
    jb.lane = threadIdx.x & 63;
    jb.H = H;
    jb.W = W;
    jb.ybeg = g * seg_rows;
    jb.yend = min(H, jb.ybeg + seg_rows);
    jb.x0 = s * SnmCfg<NP>::SW + (jb.lane - 1) * SnmCfg<NP>::PX; // column of this lane's pixel 0
    jb.fin = (const int16_t *)in + (size_t)f * H * W;
    jb.fin8 = (const uint8_t *)in + (size_t)f * H * W;
    jb.fout = out + (size_t)f * H * W; // PLANES: the provisional edge map
    if (PLANES) {
        jb.edge_value = pl.edge_value;
        const size_t frame_bytes = (size_t)pl.tiles_y * pl.tiles_x * 512;
        jb.pconn = pl.conn + (size_t)f * frame_bytes;
        jb.pstrong = pl.strong + (size_t)f * frame_bytes;
        jb.tiles_x = pl.tiles_x;
        jb.lo1 = pl.lo1;
        jb.hi1 = pl.hi1;
    }
    // first strip: column 0 and the out-of-image halo lane; last strip: column W-1 and columns >= W
    const bool col_edge = (s == 0) || ((s + 1) * SnmCfg<NP>::SW + SnmCfg<NP>::PX >= W);
    // rows touched: ybeg-2 .. (rounded-up last row) + 2 prefetched  <=  yend + 5 (+ 3: yend + 6 in the f32 strips)
    const bool row_edge = (jb.ybeg < 2) || (jb.yend + (FLT && FLT_EARLY_RELOAD ? 6 : 5) >= H);
    if constexpr (FLT) {
        static_assert(NP == 4, "the f32 variant processes 8 pixels per lane");
#ifdef PROBE_VARIANT
        fmarch_strip<(PROBE_VARIANT & 1) != 0, (PROBE_VARIANT & 2) != 0, PLANES, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
        return;
#endif
        if (col_edge) {
            if (row_edge)
                fmarch_strip<true, true, PLANES, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
            else
                fmarch_strip<true, false, PLANES, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
        } else {
            if (row_edge)
                fmarch_strip<false, true, PLANES, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
            else
                fmarch_strip<false, false, PLANES, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
        }
    } else if (col_edge) {
        if (row_edge)
            march_strip<true, true, PLANES, NP, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
        else
            march_strip<true, false, PLANES, NP, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
    } else {
        if (row_edge)
            march_strip<false, true, PLANES, NP, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
        else
            march_strip<false, false, PLANES, NP, LDS_PLANES, IN_U8>(jb, stage_mem, edge_lut);
    }
}

bool sobel_nms_march_supported(int height, int width) { return height >= 2 && width >= 2; }

static int px_variant = 0; // A/B switch "tune_sobel_px": 0 = 8 pixels per lane, 1 = 4 pixels per lane
void sobel_nms_set_px_variant(int v) { px_variant = v; }
// "tune_sobel_variant": 0 = automatic (default): the f32 marching arithmetic for the fused Sobel+NMS+classify kernel
// canny() runs, round 2's packed-i16 arithmetic for the s16 -> s16 stage-API kernel -- what three interleaved A/B
// sessions on three boxes measured (profiles/r03/ab_session_s{1,2,3}.txt: fused kernel on the u8 plane 0.87-0.90 ms
// f32 against 0.92-0.95 packed; stage kernel 0.86-0.90 packed against 0.89-0.96 f32); 1 = packed-i16 everywhere,
// 2 = f32 everywhere (parity tests run all of them)
static int arith_variant = 0;
void sobel_nms_set_arith_variant(int v) { arith_variant = v; }
// A/B switch "tune_plane_stores": 0 = plane bytes staged in LDS and written as words, 1 = direct byte stores
static int plane_store_variant = 0;
void sobel_nms_set_plane_store_variant(int v) { plane_store_variant = v; }

// tune_seg: 0 = automatic, else rows per segment (A/B knob).
// ev: optional event pair ATTACHED to the dispatch (hipExtLaunchKernel): the kernel's own begin/end timestamps,
// without the two barrier packets that hipEventRecord before and after a launch puts into the stream.
template <class K, class... A>
static void launch_timed(K kernel, dim3 grid, dim3 block, hipStream_t stream, const LaunchEvents &ev, A... args)
{
    if (ev.start && ev.stop)
        hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev.start, ev.stop, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
}

static hipError_t launch_march(const void *smoothed, int16_t *out, const PlaneArgs *planes, int height, int width,
                               int n_frames, hipStream_t stream, int tune_seg, const LaunchEvents &ev,
                               bool in_u8 = false)
{
    const int np = px_variant == 1 ? 2 : 4;
    if (in_u8 && !(np == 4 && (!planes || plane_store_variant == 0))) return hipErrorNotSupported;
    const int sw = 62 * 2 * np;
    int n_strips = (width + sw - 1) / sw;
    // 64-row segments (6 % halo rows) beat longer ones at every batch size measured (64 x 4K: 0.462 ms against
    // 0.490 at 128 rows and 0.589 at 256); 32 rows only when that is what it takes to give the chip enough waves
    int seg = 64;
    while (seg > 32 && (long long)n_frames * n_strips * ((height + seg - 1) / seg) < 16384) seg >>= 1;
    // a single frame: fewer waves than SIMDs even then -- latency, not throughput, is what is left to win
    while (seg > 8 && (long long)n_frames * n_strips * ((height + seg - 1) / seg) < 2048) seg >>= 1;
    if (tune_seg >= 8) seg = tune_seg;
    if (planes && np == 4 && plane_store_variant == 0 && seg > SNM_STAGE_ROWS) seg = SNM_STAGE_ROWS; // a segment's plane bytes are parked in LDS
    int n_segs = (height + seg - 1) / seg;
    long long waves = (long long)n_frames * n_strips * n_segs;
    if (waves > 0x7fffffffLL) return hipErrorInvalidValue;
    unsigned blocks = (unsigned)((waves + SNM_WPB - 1) / SNM_WPB);
    const PlaneArgs pl = planes ? *planes : PlaneArgs{};
    const dim3 grid(blocks), block(SNM_WPB * 64);
    const bool use_f32 = arith_variant == 2 || (arith_variant == 0 && planes != nullptr);
    if (use_f32 && np == 4) { // the f32 arithmetic: every 8-pixel form
        const bool lds = planes && plane_store_variant == 0;
#define CANNY_FLT_LAUNCH(P, L, U)                                                                                      \
    launch_timed(sobel_nms_march_kernel<P, 4, L, U, true>, grid, block, stream, ev, smoothed, out, height, width,      \
                 n_strips, n_segs, seg, (int)waves, pl)
        if (planes && lds && in_u8) CANNY_FLT_LAUNCH(true, true, true);
        else if (planes && lds) CANNY_FLT_LAUNCH(true, true, false);
        else if (planes) CANNY_FLT_LAUNCH(true, false, false);
        else if (in_u8) CANNY_FLT_LAUNCH(false, false, true);
        else CANNY_FLT_LAUNCH(false, false, false);
#undef CANNY_FLT_LAUNCH
        return hipGetLastError();
    }
    if (in_u8 && planes)
        launch_timed(sobel_nms_march_kernel<true, 4, true, true>, grid, block, stream, ev, smoothed, out, height, width,
                     n_strips, n_segs, seg, (int)waves, pl);
    else if (in_u8)
        launch_timed(sobel_nms_march_kernel<false, 4, false, true>, grid, block, stream, ev, smoothed, out, height,
                     width, n_strips, n_segs, seg, (int)waves, pl);
    else if (planes && np == 4 && plane_store_variant == 0)
        launch_timed(sobel_nms_march_kernel<true, 4, true>, grid, block, stream, ev, smoothed, out, height, width,
                     n_strips, n_segs, seg, (int)waves, pl);
    else if (planes && np == 4)
        launch_timed(sobel_nms_march_kernel<true, 4, false>, grid, block, stream, ev, smoothed, out, height, width,
                     n_strips, n_segs, seg, (int)waves, pl);
    else if (planes)
        launch_timed(sobel_nms_march_kernel<true, 2, false>, grid, block, stream, ev, smoothed, out, height, width,
                     n_strips, n_segs, seg, (int)waves, pl);
    else if (np == 4)
        launch_timed(sobel_nms_march_kernel<false, 4, false>, grid, block, stream, ev, smoothed, out, height, width,
                     n_strips, n_segs, seg, (int)waves, pl);
    else
        launch_timed(sobel_nms_march_kernel<false, 2, false>, grid, block, stream, ev, smoothed, out, height, width,
                     n_strips, n_segs, seg, (int)waves, pl);
    return hipGetLastError();
}

hipError_t launch_sobel_nms_march(const int16_t *smoothed, int16_t *out, int height, int width, int n_frames,
                                  hipStream_t stream, int tune_seg, const LaunchEvents &ev)
{
    return launch_march(smoothed, out, nullptr, height, width, n_frames, stream, tune_seg, ev);
}

bool sobel_nms_classify_supported(int height, int width, int min_val)
{
    return sobel_nms_march_supported(height, width) && width % 8 == 0 && min_val >= 1;
}

bool sobel_nms_u8_input_supported() { return px_variant == 0 && plane_store_variant == 0; }

hipError_t launch_sobel_nms_march_u8in(const uint8_t *smoothed, int16_t *out, int height, int width, int n_frames,
                                       hipStream_t stream, int tune_seg, const LaunchEvents &ev)
{
    return launch_march(smoothed, out, nullptr, height, width, n_frames, stream, tune_seg, ev, /*in_u8=*/true);
}

static hipError_t classify_march_any(const void *smoothed, bool in_u8, int16_t *edges, uint64_t *strong, uint64_t *conn,
                                     const HystGeom &g, int min_val, int max_val, int edge_value, hipStream_t stream,
                                     int tune_seg, const LaunchEvents &ev);

hipError_t launch_sobel_nms_classify_march(const int16_t *smoothed, int16_t *edges, uint64_t *strong, uint64_t *conn,
                                           const HystGeom &g, int min_val, int max_val, int edge_value,
                                           hipStream_t stream, int tune_seg, const LaunchEvents &ev)
{
    return classify_march_any(smoothed, false, edges, strong, conn, g, min_val, max_val, edge_value, stream, tune_seg, ev);
}

hipError_t launch_sobel_nms_classify_march_u8in(const uint8_t *smoothed, int16_t *edges, uint64_t *strong,
                                                uint64_t *conn, const HystGeom &g, int min_val, int max_val,
                                                int edge_value, hipStream_t stream, int tune_seg,
                                                const LaunchEvents &ev)
{
    return classify_march_any(smoothed, true, edges, strong, conn, g, min_val, max_val, edge_value, stream, tune_seg, ev);
}

static hipError_t classify_march_any(const void *smoothed, bool in_u8, int16_t *edges, uint64_t *strong, uint64_t *conn,
                                     const HystGeom &g, int min_val, int max_val, int edge_value, hipStream_t stream,
                                     int tune_seg, const LaunchEvents &ev)
{
    if (!sobel_nms_classify_supported(g.height, g.width, min_val)) return hipErrorNotSupported;
    if (edge_value < 0 || edge_value > 32767) return hipErrorInvalidValue;
    // magnitudes are <= 1442, so thresholds beyond that all mean "never"; clamping keeps hi1 from overflowing
    const int lo = min_val > 4096 ? 4096 : min_val;
    const int hi = max_val > 4096 ? 4096 : max_val;
    PlaneArgs pl;
    pl.conn = (uint8_t *)conn;
    pl.strong = (uint8_t *)strong;
    pl.tiles_x = g.tiles_x;
    pl.tiles_y = g.tiles_y;
    pl.lo1 = lo - 1;
    pl.hi1 = (hi > lo ? hi : lo) - 1;
    pl.edge_value = edge_value;
    return launch_march(smoothed, edges, &pl, g.height, g.width, g.n_frames, stream, tune_seg, ev, in_u8);
}

} // namespace canny
