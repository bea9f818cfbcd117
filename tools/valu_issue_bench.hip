// valu_issue_bench.hip -- how many cycles does one SIMD of gfx950 need per wave64 vector instruction?
//
// DESIGN.md (round 1) priced both hot kernels on "every VALU instruction costs 4 cycles per wave64";
// MI355X_MICROARCH.md says 2 (SIMD-32) with 4 only for a wave that is alone on its SIMD.  This tool measures it
// for the instruction forms the Canny kernels are made of, at 1/2/4/8 waves per SIMD:
//
//   * each wave executes ITERS x 64 copies of one instruction inside one asm block (16 independent accumulators,
//     or fewer for the dependent-chain variants) between two s_memtime stamps;
//   * waves per SIMD are forced by workgroup size and an LDS allocation that admits exactly the intended number
//     of workgroups per CU, and CHECKED by a census of HW_REG_HW_ID / HW_REG_XCC_ID (the table prints the
//     waves-per-SIMD the census saw);
//   * reported: cycles per instruction as one wave sees it (median over waves) and the SIMD's issue interval
//     = that / waves-per-SIMD, i.e. cycles of SIMD time per wave-instruction.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue_bench valu_issue_bench.hip ; run: ./valu_issue_bench [out.json]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                            \
        }                                                                                            \
    } while (0)

struct Stamp {
    unsigned long long t0, t1;   // s_memtime (shader clock)
    unsigned long long r0, r1;   // s_memrealtime (constant 100 MHz)
    unsigned hw_id, xcc_id;
};

// 16 accumulators a0..a15 ("+v"), two vector sources s0, s1 ("v"), one scalar source ("s").
// X(d) expands to one instruction on accumulator d.
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP8x2(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP4x4(X) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#define REP2x8(X) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1)
#define REP1x16(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)

#define OPERANDS                                                                                                  \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), \
      "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])                    \
    : "v"(s0), "v"(s1), "s"(sc), "v"(lds_addr)                                                                    \
    : "vcc", "memory"
// operand numbers: %0..%15 accumulators, %16 s0, %17 s1, %18 scalar, %19 LDS byte address

#define DEF_KERNEL(NAME, REP, X, TAIL)                                                               \
    __global__ void __launch_bounds__(1024) k_##NAME(Stamp *stamps, unsigned *sink, int iters)       \
    {                                                                                                \
        extern __shared__ unsigned lds[];                                                            \
        unsigned a[16];                                                                              \
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 16 + i;                                    \
        unsigned s0 = threadIdx.x | 1u, s1 = threadIdx.x * 3u + 7u;                                  \
        unsigned sc = (unsigned)iters | 3u;                                                          \
        unsigned lds_addr = (threadIdx.x & 63u) * 4u;                                                \
        for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;                        \
        __syncthreads();                                                                             \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < iters; it++) {                                                         \
            asm volatile(REP(X) REP(X) REP(X) REP(X) TAIL OPERANDS);                                 \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned acc = 0;                                                                            \
        for (int i = 0; i < 16; i++) acc ^= a[i];                                                    \
        if (acc == 0x12345u) sink[0] = acc;                                                          \
        if ((threadIdx.x & 63) == 0) {                                                               \
            unsigned hw, xcc;                                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
            Stamp s{t0, t1, r0, r1, hw, xcc};                                                                \
            stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                           \
        }                                                                                            \
    }

// ---- instruction forms ---------------------------------------------------------------------------------------
#define I_ADD_F32(d) "v_add_f32 %" #d ", %" #d ", %16\n\t"
#define I_MUL_F32(d) "v_mul_f32 %" #d ", %" #d ", %16\n\t"
#define I_FMA_F32(d) "v_fma_f32 %" #d ", %" #d ", %16, %17\n\t"
#define I_ADD_DPP_WSHR(d) "v_add_f32_dpp %" #d ", %16, %" #d " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD_DPP_RSHR(d) "v_add_f32_dpp %" #d ", %16, %" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MOV_DPP_WSHR(d) "v_mov_b32_dpp %" #d ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MOV_DPP_RSHR(d) "v_mov_b32_dpp %" #d ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_PK_MAD_I16(d) "v_pk_mad_i16 %" #d ", %" #d ", %16, %17\n\t"
#define I_PK_SUB_I16(d) "v_pk_sub_i16 %" #d ", %" #d ", %16\n\t"
#define I_PK_MAX_I16(d) "v_pk_max_i16 %" #d ", %" #d ", %16\n\t"
#define I_PK_ADD_F32(d) "v_pk_add_f32 %[p" #d "], %[p" #d "], %[ps]\n\t"
#define I_CNDMASK_SDWA(d) \
    "v_cndmask_b32_sdwa %" #d ", %16, %17, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
#define I_CNDMASK(d) "v_cndmask_b32 %" #d ", %16, %" #d ", vcc\n\t"
#define I_CVT_F32_I32(d) "v_cvt_f32_i32 %" #d ", %" #d "\n\t"
#define I_CVT_F32_UBYTE1(d) "v_cvt_f32_ubyte1 %" #d ", %16\n\t"
#define I_CVT_I32_F32(d) "v_cvt_i32_f32 %" #d ", %" #d "\n\t"
#define I_MAX3_I32(d) "v_max3_i32 %" #d ", %" #d ", %16, %17\n\t"
#define I_ADD_U32(d) "v_add_u32 %" #d ", %" #d ", %16\n\t"
#define I_AND_OR(d) "v_and_or_b32 %" #d ", %" #d ", %16, %17\n\t"
#define I_PERM(d) "v_perm_b32 %" #d ", %" #d ", %16, %17\n\t"
#define I_MAD_U32_U24(d) "v_mad_u32_u24 %" #d ", %" #d ", %16, %17\n\t"
#define I_MUL_LO_U32(d) "v_mul_lo_u32 %" #d ", %" #d ", %16\n\t"
#define I_SQRT_F32(d) "v_sqrt_f32 %" #d ", %" #d "\n\t"
#define I_RCP_F32(d) "v_rcp_f32 %" #d ", %" #d "\n\t"
#define I_CMP_ADDC(d) "v_cmp_gt_i32 vcc, %" #d ", %16\n\ts_nop 1\n\tv_addc_co_u32 %" #d ", vcc, %" #d ", %" #d ", vcc\n\t"
#define I_CMP_ONLY(d) "v_cmp_gt_i32 vcc, %" #d ", %16\n\t"
#define I_ADD_SGPR(d) "v_add_f32 %" #d ", %18, %" #d "\n\t"
#define I_ADD_F32_E64(d) "v_add_f32_e64 %" #d ", %" #d ", |%16|\n\t"
#define I_ADD_SDWA(d) "v_add_f32_sdwa %" #d ", %" #d ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define I_DS_READ2ST64(d) "ds_read2st64_b32 %[p" #d "], %19 offset0:0 offset1:1\n\t"
#define I_DS_READ_B32(d) "ds_read_b32 %" #d ", %19\n\t"
#define I_DS_READ_B128(d) "ds_read_b128 %[q" #d "], %19\n\t"
#define I_ADD_THEN_DSREAD(d) "v_add_f32 %" #d ", %" #d ", %16\n\t"
#define I_NOP(d) "s_nop 0\n\t"
#define WAIT_LGKM "s_waitcnt lgkmcnt(0)\n\t"

DEF_KERNEL(add_f32, REP16, I_ADD_F32, "")
DEF_KERNEL(add_f32_dep8, REP8x2, I_ADD_F32, "")
DEF_KERNEL(add_f32_dep4, REP4x4, I_ADD_F32, "")
DEF_KERNEL(add_f32_dep2, REP2x8, I_ADD_F32, "")
DEF_KERNEL(add_f32_dep1, REP1x16, I_ADD_F32, "")
DEF_KERNEL(mul_f32, REP16, I_MUL_F32, "")
DEF_KERNEL(fma_f32, REP16, I_FMA_F32, "")
DEF_KERNEL(add_f32_sgpr, REP16, I_ADD_SGPR, "")
DEF_KERNEL(add_f32_e64, REP16, I_ADD_F32_E64, "")
DEF_KERNEL(add_f32_sdwa, REP16, I_ADD_SDWA, "")
DEF_KERNEL(add_f32_dpp_wave_shr, REP16, I_ADD_DPP_WSHR, "")
DEF_KERNEL(add_f32_dpp_wave_shr_dep4, REP4x4, I_ADD_DPP_WSHR, "")
DEF_KERNEL(add_f32_dpp_row_shr, REP16, I_ADD_DPP_RSHR, "")
DEF_KERNEL(mov_dpp_wave_shr, REP16, I_MOV_DPP_WSHR, "")
DEF_KERNEL(mov_dpp_row_shr, REP16, I_MOV_DPP_RSHR, "")
DEF_KERNEL(pk_mad_i16, REP16, I_PK_MAD_I16, "")
DEF_KERNEL(pk_sub_i16, REP16, I_PK_SUB_I16, "")
DEF_KERNEL(pk_max_i16, REP16, I_PK_MAX_I16, "")
DEF_KERNEL(cndmask_sdwa, REP16, I_CNDMASK_SDWA, "")
DEF_KERNEL(cndmask, REP16, I_CNDMASK, "")
DEF_KERNEL(cvt_f32_i32, REP16, I_CVT_F32_I32, "")
DEF_KERNEL(cvt_f32_ubyte1, REP16, I_CVT_F32_UBYTE1, "")
DEF_KERNEL(cvt_i32_f32, REP16, I_CVT_I32_F32, "")
DEF_KERNEL(max3_i32, REP16, I_MAX3_I32, "")
DEF_KERNEL(add_u32, REP16, I_ADD_U32, "")
DEF_KERNEL(and_or_b32, REP16, I_AND_OR, "")
DEF_KERNEL(perm_b32, REP16, I_PERM, "")
DEF_KERNEL(mad_u32_u24, REP16, I_MAD_U32_U24, "")
DEF_KERNEL(mul_lo_u32, REP16, I_MUL_LO_U32, "")
DEF_KERNEL(sqrt_f32, REP16, I_SQRT_F32, "")
DEF_KERNEL(rcp_f32, REP16, I_RCP_F32, "")
DEF_KERNEL(cmp_nop_addc, REP16, I_CMP_ADDC, "")
DEF_KERNEL(cmp_gt_i32, REP16, I_CMP_ONLY, "")
DEF_KERNEL(s_nop0, REP16, I_NOP, "")
DEF_KERNEL(ds_read_b32, REP16, I_DS_READ_B32, WAIT_LGKM)

// Kernels whose operands are register PAIRS / QUADS (packed f32, ds_read2, ds_read_b128) use named operands.
#define DEF_KERNEL_WIDE(NAME, TYPE, N, PFX, X, TAIL, NREP)                                                          \
    __global__ void __launch_bounds__(1024) k_##NAME(Stamp *stamps, unsigned *sink, int iters)                      \
    {                                                                                                               \
        extern __shared__ unsigned lds[];                                                                           \
        typedef unsigned TYPE __attribute__((ext_vector_type(N)));                                                  \
        TYPE p[8];                                                                                                  \
        for (int i = 0; i < 8; i++)                                                                                 \
            for (int j = 0; j < N; j++) p[i][j] = threadIdx.x * 16 + i * N + j;                                     \
        typedef unsigned pair_t __attribute__((ext_vector_type(2)));                                                \
        pair_t ps = {threadIdx.x | 1u, threadIdx.x * 3u + 7u};                                                      \
        unsigned s0 = threadIdx.x | 1u;                                                                             \
        unsigned lds_addr = (threadIdx.x & 63u) * (4u * (N == 4 ? 4u : 1u));                                        \
        for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;                                       \
        __syncthreads();                                                                                            \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
        for (int it = 0; it < iters; it++) {                                                                        \
            asm volatile(NREP TAIL                                                                                  \
                         : [PFX##0] "+v"(p[0]), [PFX##1] "+v"(p[1]), [PFX##2] "+v"(p[2]), [PFX##3] "+v"(p[3]),      \
                           [PFX##4] "+v"(p[4]), [PFX##5] "+v"(p[5]), [PFX##6] "+v"(p[6]), [PFX##7] "+v"(p[7])       \
                         : [ps] "v"(ps), [s0] "v"(s0), [la] "v"(lds_addr)                                           \
                         : "vcc", "memory");                                                                        \
        }                                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                   \
        unsigned acc = 0;                                                                                           \
        for (int i = 0; i < 8; i++)                                                                                 \
            for (int j = 0; j < N; j++) acc ^= p[i][j];                                                             \
        if (acc == 0x12345u) sink[0] = acc;                                                                         \
        if ((threadIdx.x & 63) == 0) {                                                                              \
            unsigned hw, xcc;                                                                                       \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                      \
            Stamp s{t0, t1, r0, r1, hw, xcc};                                                                               \
            stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                                          \
        }                                                                                                           \
    }

#define REPW8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define W_PK_ADD_F32(d) "v_pk_add_f32 %[p" #d "], %[p" #d "], %[ps]\n\t"
#define W_PK_MUL_F32(d) "v_pk_mul_f32 %[p" #d "], %[p" #d "], %[ps]\n\t"
#define W_PK_FMA_F32(d) "v_pk_fma_f32 %[p" #d "], %[p" #d "], %[ps], %[ps]\n\t"
#define W_DS_READ2ST64(d) "ds_read2st64_b32 %[p" #d "], %[la] offset0:0 offset1:1\n\t"
#define W_DS_READ_B64(d) "ds_read_b64 %[p" #d "], %[la]\n\t"
#define W_DS_READ_B128(d) "ds_read_b128 %[q" #d "], %[la]\n\t"
// one LDS read followed by three plain VALU adds on the halves of OTHER registers: does the LDS port run beside the VALU?
#define W_DS2_PLUS_3ADD(d)                                \
    "ds_read2st64_b32 %[p" #d "], %[la] offset0:0 offset1:1\n\t" \
    "v_add_f32 %[s0], %[s0], %[s0]\n\t"                   \
    "v_add_f32 %[s0], %[s0], %[s0]\n\t"                   \
    "v_add_f32 %[s0], %[s0], %[s0]\n\t"
#define R8x8(X) REPW8(X) REPW8(X) REPW8(X) REPW8(X) REPW8(X) REPW8(X) REPW8(X) REPW8(X)
#define R8x2(X) REPW8(X) REPW8(X)

DEF_KERNEL_WIDE(pk_add_f32, pairv, 2, p, W_PK_ADD_F32, "", R8x8(W_PK_ADD_F32))
DEF_KERNEL_WIDE(pk_mul_f32, pairv, 2, p, W_PK_MUL_F32, "", R8x8(W_PK_MUL_F32))
DEF_KERNEL_WIDE(pk_fma_f32, pairv, 2, p, W_PK_FMA_F32, "", R8x8(W_PK_FMA_F32))
DEF_KERNEL_WIDE(ds_read2st64_b32, pairv, 2, p, W_DS_READ2ST64, WAIT_LGKM, R8x8(W_DS_READ2ST64))
DEF_KERNEL_WIDE(ds_read_b64, pairv, 2, p, W_DS_READ_B64, WAIT_LGKM, R8x8(W_DS_READ_B64))
DEF_KERNEL_WIDE(ds_read_b128, quadv, 4, q, W_DS_READ_B128, WAIT_LGKM, R8x8(W_DS_READ_B128))

// mixed stream: 16 x (1 ds_read2st64 + 3 v_add) = 64 instructions, counted as 64
__global__ void __launch_bounds__(1024) k_ds2_plus_3add(Stamp *stamps, unsigned *sink, int iters)
{
    extern __shared__ unsigned lds[];
    typedef unsigned pairv __attribute__((ext_vector_type(2)));
    pairv p[8];
    for (int i = 0; i < 8; i++) p[i] = pairv{threadIdx.x, threadIdx.x + i};
    unsigned b[3] = {threadIdx.x | 1u, threadIdx.x + 5u, threadIdx.x + 9u};
    unsigned s0 = threadIdx.x | 1u;
    unsigned lds_addr = (threadIdx.x & 63u) * 4u;
    for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;
    __syncthreads();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define MIX(d)                                                  \
    "ds_read2st64_b32 %[p" #d "], %[la] offset0:0 offset1:1\n\t" \
    "v_add_f32 %[b0], %[b0], %[s0]\n\t"                         \
    "v_add_f32 %[b1], %[b1], %[s0]\n\t"                         \
    "v_add_f32 %[b2], %[b2], %[s0]\n\t"
    for (int it = 0; it < iters; it++) {
        asm volatile(REPW8(MIX) REPW8(MIX) WAIT_LGKM
                     : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]),
                       [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7]), [b0] "+v"(b[0]), [b1] "+v"(b[1]),
                       [b2] "+v"(b[2])
                     : [s0] "v"(s0), [la] "v"(lds_addr)
                     : "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = b[0] ^ b[1] ^ b[2];
    for (int i = 0; i < 8; i++) acc ^= p[i][0] ^ p[i][1];
    if (acc == 0x12345u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Stamp s{t0, t1, r0, r1, hw, xcc};
        stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;
    }
}


// ---- mixes shaped like the hot kernels' inner loops ---------------------------------------------------------
// Gaussian row+column pass per lane-pixel (window 11): 6 v_mul_f32, 13 plain v_add_f32, 7 v_add_f32_dpp, 2 v_fma_f32,
// 1 v_cvt, 2 misc (v_lshl_or / v_add_u32) = 31 VALU + 3 ds_read2st64_b32.  Two pixels per asm block = 62 VALU + 6 DS.
#define GMIX(a, b, c, d)                                                                  \
    "ds_read2st64_b32 %[p0], %[la] offset0:0 offset1:1\n\t"                               \
    "v_mul_f32 %" #a ", %" #a ", %[s0]\n\t"                                                  \
    "v_mul_f32 %" #b ", %" #b ", %[s0]\n\t"                                                  \
    "v_add_f32 %" #c ", %" #c ", %[s0]\n\t"                                                  \
    "v_add_f32_dpp %" #d ", %[s0], %" #d " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"        \
    "v_add_f32 %" #a ", %" #a ", %[s1]\n\t"                                                  \
    "v_add_f32 %" #b ", %" #b ", %[s1]\n\t"                                                  \
    "v_add_f32_dpp %" #c ", %[s1], %" #c " wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"        \
    "v_add_f32 %" #d ", %" #d ", %[s0]\n\t"                                                  \
    "v_mul_f32 %" #a ", %" #a ", %[s1]\n\t"                                                  \
    "v_add_f32 %" #b ", %" #b ", %[s0]\n\t"
// 10 VALU (3 mul, 5 add, 2 dpp) + 1 DS per GMIX; x6 = 60 VALU + 6 DS, plus 2 fma + 1 cvt + 1 misc = 64 VALU
__global__ void __launch_bounds__(1024) k_gauss_mix(Stamp *stamps, unsigned *sink, int iters)
{
    extern __shared__ unsigned lds[];
    typedef unsigned pairv __attribute__((ext_vector_type(2)));
    unsigned a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 16 + i;
    pairv p0 = {threadIdx.x, threadIdx.x + 1};
    unsigned s0 = threadIdx.x | 1u, s1 = threadIdx.x * 3u + 7u;
    unsigned lds_addr = (threadIdx.x & 63u) * 4u;
    for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;
    __syncthreads();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        asm volatile(GMIX(0, 1, 2, 3) GMIX(4, 5, 6, 7) GMIX(8, 9, 10, 11) GMIX(12, 13, 14, 15) GMIX(0, 5, 10, 15)
                         GMIX(1, 6, 11, 12) "v_fma_f32 %0, %0, %[s0], %[s1]\n\t"
                                            "v_fma_f32 %1, %1, %[s0], %[s1]\n\t"
                                            "v_cvt_i32_f32 %2, %2\n\t"
                                            "v_lshl_or_b32 %3, %3, 16, %[s1]\n\t"
                                            "s_waitcnt lgkmcnt(0)\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                       "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]),
                       "+v"(a[15]), [p0] "+v"(p0)
                     : [s0] "v"(s0), [s1] "v"(s1), [la] "v"(lds_addr)
                     : "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = p0[0] ^ p0[1];
    for (int i = 0; i < 16; i++) acc ^= a[i];
    if (acc == 0x12345u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Stamp s{t0, t1, r0, r1, hw, xcc};
        stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;
    }
}

// the same VALU mix without the LDS reads, and with the DPP adds replaced by plain adds (what do DPP and DS cost?)
#define GMIX_NODS(a, b, c, d)                                                             \
    "v_mul_f32 %" #a ", %" #a ", %16\n\t"                                                  \
    "v_mul_f32 %" #b ", %" #b ", %16\n\t"                                                  \
    "v_add_f32 %" #c ", %" #c ", %16\n\t"                                                  \
    "v_add_f32_dpp %" #d ", %16, %" #d " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"        \
    "v_add_f32 %" #a ", %" #a ", %17\n\t"                                                  \
    "v_add_f32 %" #b ", %" #b ", %17\n\t"                                                  \
    "v_add_f32_dpp %" #c ", %17, %" #c " wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"        \
    "v_add_f32 %" #d ", %" #d ", %16\n\t"                                                  \
    "v_mul_f32 %" #a ", %" #a ", %17\n\t"                                                  \
    "v_add_f32 %" #b ", %" #b ", %16\n\t"
#define GMIX_PLAIN(a, b, c, d)                                                            \
    "v_mul_f32 %" #a ", %" #a ", %16\n\t"                                                  \
    "v_mul_f32 %" #b ", %" #b ", %16\n\t"                                                  \
    "v_add_f32 %" #c ", %" #c ", %16\n\t"                                                  \
    "v_add_f32 %" #d ", %16, %" #d "\n\t"                                                  \
    "v_add_f32 %" #a ", %" #a ", %17\n\t"                                                  \
    "v_add_f32 %" #b ", %" #b ", %17\n\t"                                                  \
    "v_add_f32 %" #c ", %17, %" #c "\n\t"                                                  \
    "v_add_f32 %" #d ", %" #d ", %16\n\t"                                                  \
    "v_mul_f32 %" #a ", %" #a ", %17\n\t"                                                  \
    "v_add_f32 %" #b ", %" #b ", %16\n\t"
#define G6(M) M(0, 1, 2, 3) M(4, 5, 6, 7) M(8, 9, 10, 11) M(12, 13, 14, 15) M(0, 5, 10, 15) M(1, 6, 11, 12)
#define GTAIL "v_add_f32 %0, %0, %16\n\tv_add_f32 %1, %1, %16\n\tv_add_f32 %2, %2, %16\n\tv_add_f32 %3, %3, %16\n\t"
#define REP_G_NODS(X) G6(GMIX_NODS) GTAIL
#define REP_G_PLAIN(X) G6(GMIX_PLAIN) GTAIL
#define I_UNUSED(d) ""
#define DEF_KERNEL1(NAME, BODY)                                                                      \
    __global__ void __launch_bounds__(1024) k_##NAME(Stamp *stamps, unsigned *sink, int iters)       \
    {                                                                                                \
        extern __shared__ unsigned lds[];                                                            \
        unsigned a[16];                                                                              \
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 16 + i;                                    \
        unsigned s0 = threadIdx.x | 1u, s1 = threadIdx.x * 3u + 7u;                                  \
        unsigned sc = (unsigned)iters | 3u;                                                          \
        unsigned lds_addr = (threadIdx.x & 63u) * 4u;                                                \
        for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;                        \
        __syncthreads();                                                                             \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < iters; it++) {                                                         \
            asm volatile(BODY OPERANDS);                                                             \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned acc = 0;                                                                            \
        for (int i = 0; i < 16; i++) acc ^= a[i];                                                    \
        if (acc == 0x12345u) sink[0] = acc;                                                          \
        if ((threadIdx.x & 63) == 0) {                                                               \
            unsigned hw, xcc;                                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
            Stamp s{t0, t1, r0, r1, hw, xcc};                                                        \
            stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                           \
        }                                                                                            \
    }
DEF_KERNEL1(gauss_mix_nods, G6(GMIX_NODS) GTAIL)
DEF_KERNEL1(gauss_mix_plain, G6(GMIX_PLAIN) GTAIL)
// alternating classes: does a "slow-class" instruction between two plain ones cost its own price or more?
#define ALT_ADD_DPP(d) "v_add_f32 %" #d ", %" #d ", %16\n\tv_add_f32_dpp %" #d ", %17, %" #d " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define ALT_ADD_FMA(d) "v_add_f32 %" #d ", %" #d ", %16\n\tv_fma_f32 %" #d ", %" #d ", %16, %17\n\t"
DEF_KERNEL1(alt_add_dpp, REP16(ALT_ADD_DPP) REP16(ALT_ADD_DPP))
DEF_KERNEL1(alt_add_fma, REP16(ALT_ADD_FMA) REP16(ALT_ADD_FMA))
// distinct source registers (VGPR bank effects): a[d] += a[(d+5)%16]
#define ADD_X(d, e) "v_add_f32 %" #d ", %" #d ", %" #e "\n\t"
#define REP_X ADD_X(0, 5) ADD_X(1, 6) ADD_X(2, 7) ADD_X(3, 8) ADD_X(4, 9) ADD_X(5, 10) ADD_X(6, 11) ADD_X(7, 12) ADD_X(8, 13) ADD_X(9, 14) ADD_X(10, 15) ADD_X(11, 0) ADD_X(12, 1) ADD_X(13, 2) ADD_X(14, 3) ADD_X(15, 4)
DEF_KERNEL1(add_f32_distinct_src, REP_X REP_X REP_X REP_X)
#define FMA_X(d, e, f) "v_fma_f32 %" #d ", %" #d ", %" #e ", %" #f "\n\t"
#define REP_FX FMA_X(0, 5, 10) FMA_X(1, 6, 11) FMA_X(2, 7, 12) FMA_X(3, 8, 13) FMA_X(4, 9, 14) FMA_X(5, 10, 15) FMA_X(6, 11, 0) FMA_X(7, 12, 1) FMA_X(8, 13, 2) FMA_X(9, 14, 3) FMA_X(10, 15, 4) FMA_X(11, 0, 5) FMA_X(12, 1, 6) FMA_X(13, 2, 7) FMA_X(14, 3, 8) FMA_X(15, 4, 9)
DEF_KERNEL1(fma_f32_distinct_src, REP_FX REP_FX REP_FX REP_FX)
#define I_ADD_LIT(d) "v_add_f32 %" #d ", 0x3f801234, %" #d "\n\t"
#define I_ADD_INL(d) "v_add_f32 %" #d ", 1.0, %" #d "\n\t"
#define I_MAC(d) "v_fmac_f32 %" #d ", %16, %17\n\t"
#define I_SUB_U16(d) "v_sub_u16 %" #d ", %" #d ", %16\n\t"
#define I_LSHL_ADD(d) "v_lshl_add_u32 %" #d ", %" #d ", 1, %16\n\t"
#define I_LSHLREV(d) "v_lshlrev_b32 %" #d ", 1, %" #d "\n\t"
#define I_MAX_I32(d) "v_max_i32 %" #d ", %" #d ", %16\n\t"
#define I_AND(d) "v_and_b32 %" #d ", %" #d ", %16\n\t"
#define I_MOV(d) "v_mov_b32 %" #d ", %16\n\t"
#define I_ALIGNBIT(d) "v_alignbit_b32 %" #d ", %" #d ", %16, 16\n\t"
#define I_MUL_I32_I24(d) "v_mul_i32_i24 %" #d ", %" #d ", %16\n\t"
#define I_CVT_PK_U8(d) "v_cvt_pk_u8_f32 %" #d ", %16, 1, %" #d "\n\t"
#define I_SUBREV(d) "v_subrev_f32 %" #d ", %" #d ", %16\n\t"
DEF_KERNEL(add_f32_literal, REP16, I_ADD_LIT, "")
DEF_KERNEL(add_f32_inline1, REP16, I_ADD_INL, "")
DEF_KERNEL(fmac_f32, REP16, I_MAC, "")
DEF_KERNEL(sub_u16, REP16, I_SUB_U16, "")
DEF_KERNEL(lshl_add_u32, REP16, I_LSHL_ADD, "")
DEF_KERNEL(lshlrev_b32, REP16, I_LSHLREV, "")
DEF_KERNEL(max_i32, REP16, I_MAX_I32, "")
DEF_KERNEL(and_b32, REP16, I_AND, "")
DEF_KERNEL(mov_b32, REP16, I_MOV, "")
DEF_KERNEL(alignbit_b32, REP16, I_ALIGNBIT, "")
DEF_KERNEL(mul_i32_i24, REP16, I_MUL_I32_I24, "")
DEF_KERNEL(cvt_pk_u8_f32, REP16, I_CVT_PK_U8, "")


// ---- round 2, second batch: what makes an instruction "slow class", and what do switches between classes cost? ----
#define I_MAX_F32(d) "v_max_f32 %" #d ", %" #d ", %16\n\t"
#define I_SUB_F32(d) "v_sub_f32 %" #d ", %" #d ", %16\n\t"
#define I_OR_B32(d) "v_or_b32 %" #d ", %" #d ", %16\n\t"
#define I_XOR_B32(d) "v_xor_b32 %" #d ", %" #d ", %16\n\t"
#define I_MIN_U32(d) "v_min_u32 %" #d ", %" #d ", %16\n\t"
#define I_ASHRREV(d) "v_ashrrev_i32 %" #d ", 3, %" #d "\n\t"
#define I_BFE_U32(d) "v_bfe_u32 %" #d ", %" #d ", 8, 8\n\t"
#define I_LSHL_OR(d) "v_lshl_or_b32 %" #d ", %" #d ", 16, %16\n\t"
#define I_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %16, %17\n\t"
#define I_BFI(d) "v_bfi_b32 %" #d ", %16, %17, %" #d "\n\t"
#define I_MED3_I32(d) "v_med3_i32 %" #d ", %" #d ", %16, %17\n\t"
#define I_CVT_F32_U32(d) "v_cvt_f32_u32 %" #d ", %" #d "\n\t"
#define I_SUBREV_U32(d) "v_subrev_u32 %" #d ", %" #d ", %16\n\t"
#define I_MUL_U32_U24(d) "v_mul_u32_u24 %" #d ", %" #d ", %16\n\t"
#define I_PK_ADD_U16(d) "v_pk_add_u16 %" #d ", %" #d ", %16\n\t"
#define I_MAX_U16(d) "v_max_u16 %" #d ", %" #d ", %16\n\t"
#define I_SWIZZLE(d) "ds_swizzle_b32 %" #d ", %" #d " offset:swizzle(SWAP,1)\n\t"
#define I_BPERMUTE(d) "ds_bpermute_b32 %" #d ", %19, %" #d "\n\t"
#define I_CMP_CNDMASK(d) "v_cmp_gt_i32 vcc, %" #d ", %16\n\tv_cndmask_b32 %" #d ", %17, %" #d ", vcc\n\t"
#define I_CMP_NOP_CNDMASK(d) "v_cmp_gt_i32 vcc, %" #d ", %16\n\ts_nop 1\n\tv_cndmask_b32 %" #d ", %17, %" #d ", vcc\n\t"
#define I_CNDMASK_E64(d) "v_cndmask_b32_e64 %" #d ", %16, %" #d ", s[20:21]\n\t"
#define I_CMPE64_CNDMASK(d) "v_cmp_gt_i32_e64 s[20:21], %" #d ", %16\n\ts_nop 1\n\tv_cndmask_b32_e64 %" #d ", %17, %" #d ", s[20:21]\n\t"
#define I_CMP_SDST(d) "v_cmp_gt_i32_e64 s[20:21], %" #d ", %16\n\t"
#define I_ADDC(d) "v_addc_co_u32 %" #d ", vcc, %" #d ", %16, vcc\n\t"
#define I_MUL_SGPR(d) "v_mul_f32 %" #d ", %18, %" #d "\n\t"
DEF_KERNEL(max_f32, REP16, I_MAX_F32, "")
DEF_KERNEL(sub_f32, REP16, I_SUB_F32, "")
DEF_KERNEL(or_b32, REP16, I_OR_B32, "")
DEF_KERNEL(xor_b32, REP16, I_XOR_B32, "")
DEF_KERNEL(min_u32, REP16, I_MIN_U32, "")
DEF_KERNEL(ashrrev_i32, REP16, I_ASHRREV, "")
DEF_KERNEL(bfe_u32, REP16, I_BFE_U32, "")
DEF_KERNEL(lshl_or_b32, REP16, I_LSHL_OR, "")
DEF_KERNEL(add3_u32, REP16, I_ADD3, "")
DEF_KERNEL(bfi_b32, REP16, I_BFI, "")
DEF_KERNEL(med3_i32, REP16, I_MED3_I32, "")
DEF_KERNEL(cvt_f32_u32, REP16, I_CVT_F32_U32, "")
DEF_KERNEL(subrev_u32, REP16, I_SUBREV_U32, "")
DEF_KERNEL(mul_u32_u24, REP16, I_MUL_U32_U24, "")
DEF_KERNEL(pk_add_u16, REP16, I_PK_ADD_U16, "")
DEF_KERNEL(max_u16, REP16, I_MAX_U16, "")
DEF_KERNEL(ds_swizzle, REP16, I_SWIZZLE, WAIT_LGKM)
DEF_KERNEL(ds_bpermute, REP16, I_BPERMUTE, WAIT_LGKM)
DEF_KERNEL(cmp_cndmask, REP16, I_CMP_CNDMASK, "")
DEF_KERNEL(cmp_nop_cndmask, REP16, I_CMP_NOP_CNDMASK, "")
DEF_KERNEL(addc_only, REP16, I_ADDC, "")
DEF_KERNEL(mul_f32_sgpr, REP16, I_MUL_SGPR, "")
// SGPR-pair masks: s[20:21] is clobbered explicitly
#define OPERANDS_S20                                                                                              \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), \
      "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])                    \
    : "v"(s0), "v"(s1), "s"(sc), "v"(lds_addr)                                                                    \
    : "vcc", "memory", "s20", "s21"
#define DEF_KERNEL_S20(NAME, BODY)                                                                   \
    __global__ void __launch_bounds__(1024) k_##NAME(Stamp *stamps, unsigned *sink, int iters)       \
    {                                                                                                \
        extern __shared__ unsigned lds[];                                                            \
        unsigned a[16];                                                                              \
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 16 + i;                                    \
        unsigned s0 = threadIdx.x | 1u, s1 = threadIdx.x * 3u + 7u;                                  \
        unsigned sc = (unsigned)iters | 3u;                                                          \
        unsigned lds_addr = (threadIdx.x & 63u) * 4u;                                                \
        for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;                        \
        __syncthreads();                                                                             \
        asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");                                 \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < iters; it++) {                                                         \
            asm volatile(BODY OPERANDS_S20);                                                         \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned acc = 0;                                                                            \
        for (int i = 0; i < 16; i++) acc ^= a[i];                                                    \
        if (acc == 0x12345u) sink[0] = acc;                                                          \
        if ((threadIdx.x & 63) == 0) {                                                               \
            unsigned hw, xcc;                                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
            Stamp s{t0, t1, r0, r1, hw, xcc};                                                        \
            stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                           \
        }                                                                                            \
    }
DEF_KERNEL_S20(cndmask_e64_sgpr, REP16(I_CNDMASK_E64) REP16(I_CNDMASK_E64) REP16(I_CNDMASK_E64) REP16(I_CNDMASK_E64))
DEF_KERNEL_S20(cmp_e64_nop_cndmask_e64, REP16(I_CMPE64_CNDMASK) REP16(I_CMPE64_CNDMASK) REP16(I_CMPE64_CNDMASK) REP16(I_CMPE64_CNDMASK))
DEF_KERNEL_S20(cmp_e64_sdst, REP16(I_CMP_SDST) REP16(I_CMP_SDST) REP16(I_CMP_SDST) REP16(I_CMP_SDST))
// DPP grouped: 52 plain adds then 12 DPP adds (the Gaussian mix's ratio), and 2 groups of 26 + 6
#define PLAIN4(a, b, c, d) "v_add_f32 %" #a ", %" #a ", %16\n\tv_mul_f32 %" #b ", %" #b ", %17\n\tv_add_f32 %" #c ", %" #c ", %17\n\tv_add_f32 %" #d ", %" #d ", %16\n\t"
#define DPP4(a, b, c, d)                                                         \
    "v_add_f32_dpp %" #a ", %16, %" #a " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %" #b ", %17, %" #b " wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %" #c ", %16, %" #c " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %" #d ", %17, %" #d " wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
#define P13 PLAIN4(0, 1, 2, 3) PLAIN4(4, 5, 6, 7) PLAIN4(8, 9, 10, 11) PLAIN4(12, 13, 14, 15) PLAIN4(0, 5, 10, 15) PLAIN4(1, 6, 11, 12) PLAIN4(2, 7, 8, 13) PLAIN4(3, 4, 9, 14) PLAIN4(0, 1, 2, 3) PLAIN4(4, 5, 6, 7) PLAIN4(8, 9, 10, 11) PLAIN4(12, 13, 14, 15) PLAIN4(0, 5, 10, 15)
DEF_KERNEL1(dpp_grouped_52_12, P13 DPP4(0, 1, 2, 3) DPP4(4, 5, 6, 7) DPP4(8, 9, 10, 11))
// spread: one DPP add after every 4-5 plain ones (13 groups of 4 plain, 12 DPP singles)
#define D1(a) "v_add_f32_dpp %" #a ", %16, %" #a " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
DEF_KERNEL1(dpp_spread_52_12, PLAIN4(0, 1, 2, 3) D1(4) PLAIN4(5, 6, 7, 8) D1(9) PLAIN4(10, 11, 12, 13) D1(14) PLAIN4(15, 0, 1, 2) D1(3) PLAIN4(4, 5, 6, 7) D1(8) PLAIN4(9, 10, 11, 12) D1(13) PLAIN4(14, 15, 0, 1) D1(2) PLAIN4(3, 4, 5, 6) D1(7) PLAIN4(8, 9, 10, 11) D1(12) PLAIN4(13, 14, 15, 0) D1(1) PLAIN4(2, 3, 4, 5) D1(6) PLAIN4(7, 8, 9, 10) D1(11) PLAIN4(12, 13, 14, 15))
// SGPR operands: every 4th instruction multiplies by an SGPR (taps as kernel arguments) vs all-VGPR
#define SG4(a, b, c, d) "v_add_f32 %" #a ", %" #a ", %16\n\tv_mul_f32 %" #b ", %18, %" #b "\n\tv_add_f32 %" #c ", %" #c ", %17\n\tv_add_f32 %" #d ", %" #d ", %16\n\t"
DEF_KERNEL1(sgpr_every_4th, SG4(0, 1, 2, 3) SG4(4, 5, 6, 7) SG4(8, 9, 10, 11) SG4(12, 13, 14, 15) SG4(0, 5, 10, 15) SG4(1, 6, 11, 12) SG4(2, 7, 8, 13) SG4(3, 4, 9, 14) SG4(0, 1, 2, 3) SG4(4, 5, 6, 7) SG4(8, 9, 10, 11) SG4(12, 13, 14, 15) SG4(0, 5, 10, 15) SG4(1, 6, 11, 12) SG4(2, 7, 8, 13) SG4(3, 4, 9, 14))
// cvt every 8th
#define CV8(a, b, c, d, e, f, g, h) PLAIN4(a, b, c, d) "v_add_f32 %" #e ", %" #e ", %16\n\tv_mul_f32 %" #f ", %" #f ", %17\n\tv_add_f32 %" #g ", %" #g ", %17\n\tv_cvt_i32_f32 %" #h ", %" #h "\n\t"
DEF_KERNEL1(cvt_every_8th, CV8(0, 1, 2, 3, 4, 5, 6, 7) CV8(8, 9, 10, 11, 12, 13, 14, 15) CV8(0, 1, 2, 3, 4, 5, 6, 7) CV8(8, 9, 10, 11, 12, 13, 14, 15) CV8(0, 1, 2, 3, 4, 5, 6, 7) CV8(8, 9, 10, 11, 12, 13, 14, 15) CV8(0, 1, 2, 3, 4, 5, 6, 7) CV8(8, 9, 10, 11, 12, 13, 14, 15))
// LDS shuffles beside plain VALU: 1 ds_swizzle + 3 plain adds
#define SW4(a, b, c, d) "ds_swizzle_b32 %" #a ", %" #a " offset:swizzle(SWAP,1)\n\tv_mul_f32 %" #b ", %" #b ", %17\n\tv_add_f32 %" #c ", %" #c ", %17\n\tv_add_f32 %" #d ", %" #d ", %16\n\t"
DEF_KERNEL1(swizzle_plus_3_plain, SW4(0, 1, 2, 3) SW4(4, 5, 6, 7) SW4(8, 9, 10, 11) SW4(12, 13, 14, 15) SW4(0, 5, 10, 15) SW4(1, 6, 11, 12) SW4(2, 7, 8, 13) SW4(3, 4, 9, 14) SW4(0, 1, 2, 3) SW4(4, 5, 6, 7) SW4(8, 9, 10, 11) SW4(12, 13, 14, 15) SW4(0, 5, 10, 15) SW4(1, 6, 11, 12) SW4(2, 7, 8, 13) SW4(3, 4, 9, 14) "s_waitcnt lgkmcnt(0)\n\t")
// 1 ds_swizzle + 7 plain
#define SW8(a, b, c, d, e, f, g, h) "ds_swizzle_b32 %" #a ", %" #a " offset:swizzle(SWAP,1)\n\tv_mul_f32 %" #b ", %" #b ", %17\n\tv_add_f32 %" #c ", %" #c ", %17\n\tv_add_f32 %" #d ", %" #d ", %16\n\t" PLAIN4(e, f, g, h)
DEF_KERNEL1(swizzle_plus_7_plain, SW8(0, 1, 2, 3, 4, 5, 6, 7) SW8(8, 9, 10, 11, 12, 13, 14, 15) SW8(0, 1, 2, 3, 4, 5, 6, 7) SW8(8, 9, 10, 11, 12, 13, 14, 15) SW8(0, 1, 2, 3, 4, 5, 6, 7) SW8(8, 9, 10, 11, 12, 13, 14, 15) SW8(0, 1, 2, 3, 4, 5, 6, 7) SW8(8, 9, 10, 11, 12, 13, 14, 15) "s_waitcnt lgkmcnt(0)\n\t")


// ---- third batch: explicit registers.  Is the 2.3-cycle rate tied to the in-place form (dst == src0) or to the VGPR
// banks (register number mod 4) of the operands?  Registers v64..v127 are clobbered explicitly.
#define CLOB64 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
#define DEF_KERNEL_X(NAME, BODY)                                                                     \
    __global__ void __launch_bounds__(1024) k_##NAME(Stamp *stamps, unsigned *sink, int iters)       \
    {                                                                                                \
        extern __shared__ unsigned lds[];                                                            \
        for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;                        \
        __syncthreads();                                                                             \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < iters; it++) {                                                         \
            asm volatile(BODY BODY BODY BODY ::: "memory", CLOB64);                                  \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                    \
        unsigned acc;                                                                                \
        asm volatile("v_mov_b32 %0, v64" : "=v"(acc)::CLOB64);                                       \
        if (acc == 0x12345u) sink[0] = acc;                                                          \
        if ((threadIdx.x & 63) == 0) {                                                               \
            unsigned hw, xcc;                                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                         \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                       \
            Stamp s{t0, t1, r0, r1, hw, xcc};                                                        \
            stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;                           \
        }                                                                                            \
    }
// 16 instructions per BODY; dst v64+4k+b..., sources chosen per test.  R(op, d, a, b) -> "op vD, vA, vB"
#define R3(op, d, a, b) op " v" #d ", v" #a ", v" #b "\n\t"
// in place, sources in different banks:        d = d op (reg in another bank)
#define X_INPLACE(op) R3(op,64,64,81) R3(op,65,65,82) R3(op,66,66,83) R3(op,67,67,80) R3(op,68,68,85) R3(op,69,69,86) R3(op,70,70,87) R3(op,71,71,84) R3(op,72,72,89) R3(op,73,73,90) R3(op,74,74,91) R3(op,75,75,88) R3(op,76,76,93) R3(op,77,77,94) R3(op,78,78,95) R3(op,79,79,92)
// three different registers, three different banks (dst bank k, src0 bank k+1, src1 bank k+2)
#define X_3REG_3BANK(op) R3(op,64,81,98) R3(op,65,82,99) R3(op,66,83,96) R3(op,67,80,97) R3(op,68,85,102) R3(op,69,86,103) R3(op,70,87,100) R3(op,71,84,101) R3(op,72,89,106) R3(op,73,90,107) R3(op,74,91,104) R3(op,75,88,105) R3(op,76,93,110) R3(op,77,94,111) R3(op,78,95,108) R3(op,79,92,109)
// three different registers, the two SOURCES in the same bank
#define X_SRC_SAME_BANK(op) R3(op,64,81,97) R3(op,65,82,98) R3(op,66,83,99) R3(op,67,80,96) R3(op,68,85,101) R3(op,69,86,102) R3(op,70,87,103) R3(op,71,84,100) R3(op,72,89,105) R3(op,73,90,106) R3(op,74,91,107) R3(op,75,88,104) R3(op,76,93,109) R3(op,77,94,110) R3(op,78,95,111) R3(op,79,92,108)
// three different registers, dst in the same bank as src0
#define X_DST_SRC0_BANK(op) R3(op,64,80,97) R3(op,65,81,98) R3(op,66,82,99) R3(op,67,83,96) R3(op,68,84,101) R3(op,69,85,102) R3(op,70,86,103) R3(op,71,87,100) R3(op,72,88,105) R3(op,73,89,106) R3(op,74,90,107) R3(op,75,91,104) R3(op,76,92,109) R3(op,77,93,110) R3(op,78,94,111) R3(op,79,95,108)
// dependent on the instruction just before (dst of i is src0 of i+1), not in place
#define X_CHAIN(op) R3(op,64,79,81) R3(op,65,64,82) R3(op,66,65,83) R3(op,67,66,80) R3(op,68,67,85) R3(op,69,68,86) R3(op,70,69,87) R3(op,71,70,84) R3(op,72,71,89) R3(op,73,72,90) R3(op,74,73,91) R3(op,75,74,88) R3(op,76,75,93) R3(op,77,76,94) R3(op,78,77,95) R3(op,79,78,92)
DEF_KERNEL_X(x_add_inplace, X_INPLACE("v_add_f32"))
DEF_KERNEL_X(x_add_3reg_3bank, X_3REG_3BANK("v_add_f32"))
DEF_KERNEL_X(x_add_src_same_bank, X_SRC_SAME_BANK("v_add_f32"))
DEF_KERNEL_X(x_add_dst_src0_bank, X_DST_SRC0_BANK("v_add_f32"))
DEF_KERNEL_X(x_add_chain, X_CHAIN("v_add_f32"))
DEF_KERNEL_X(x_mul_3reg_3bank, X_3REG_3BANK("v_mul_f32"))
DEF_KERNEL_X(x_maxu16_3reg_3bank, X_3REG_3BANK("v_max_u16"))
DEF_KERNEL_X(x_maxu16_src_same_bank, X_SRC_SAME_BANK("v_max_u16"))
DEF_KERNEL_X(x_maxi32_3reg_3bank, X_3REG_3BANK("v_max_i32"))
DEF_KERNEL_X(x_pksub_3reg_3bank, X_3REG_3BANK("v_pk_sub_i16"))
DEF_KERNEL_X(x_and_3reg_3bank, X_3REG_3BANK("v_and_b32"))
DEF_KERNEL_X(x_addu32_3reg_3bank, X_3REG_3BANK("v_add_u32"))
// 4-operand forms: d, a, b, c
#define R4(op, d, a, b, c) op " v" #d ", v" #a ", v" #b ", v" #c "\n\t"
#define X4_4BANK(op) R4(op,64,81,98,115) R4(op,65,82,99,112) R4(op,66,83,96,113) R4(op,67,80,97,114) R4(op,68,85,102,119) R4(op,69,86,103,116) R4(op,70,87,100,117) R4(op,71,84,101,118) R4(op,72,89,106,123) R4(op,73,90,107,120) R4(op,74,91,104,121) R4(op,75,88,105,122) R4(op,76,93,110,127) R4(op,77,94,111,124) R4(op,78,95,108,125) R4(op,79,92,109,126)
#define X4_ACC(op) R4(op,64,81,98,64) R4(op,65,82,99,65) R4(op,66,83,96,66) R4(op,67,80,97,67) R4(op,68,85,102,68) R4(op,69,86,103,69) R4(op,70,87,100,70) R4(op,71,84,101,71) R4(op,72,89,106,72) R4(op,73,90,107,73) R4(op,74,91,104,74) R4(op,75,88,105,75) R4(op,76,93,110,76) R4(op,77,94,111,77) R4(op,78,95,108,78) R4(op,79,92,109,79)
DEF_KERNEL_X(x_fma_4reg_4bank, X4_4BANK("v_fma_f32"))
DEF_KERNEL_X(x_fma_acc, X4_ACC("v_fma_f32"))
DEF_KERNEL_X(x_max3_4reg_4bank, X4_4BANK("v_max3_i32"))
DEF_KERNEL_X(x_pkmad_4reg_4bank, X4_4BANK("v_pk_mad_i16"))
DEF_KERNEL_X(x_pkfma_4bank, "v_pk_fma_f32 v[64:65], v[82:83], v[100:101], v[118:119]\n\tv_pk_fma_f32 v[66:67], v[84:85], v[102:103], v[120:121]\n\tv_pk_fma_f32 v[68:69], v[86:87], v[104:105], v[122:123]\n\tv_pk_fma_f32 v[70:71], v[88:89], v[106:107], v[124:125]\n\tv_pk_fma_f32 v[72:73], v[90:91], v[108:109], v[126:127]\n\tv_pk_fma_f32 v[74:75], v[92:93], v[110:111], v[112:113]\n\tv_pk_fma_f32 v[76:77], v[94:95], v[96:97], v[114:115]\n\tv_pk_fma_f32 v[78:79], v[80:81], v[98:99], v[116:117]\n\tv_pk_fma_f32 v[64:65], v[82:83], v[100:101], v[118:119]\n\tv_pk_fma_f32 v[66:67], v[84:85], v[102:103], v[120:121]\n\tv_pk_fma_f32 v[68:69], v[86:87], v[104:105], v[122:123]\n\tv_pk_fma_f32 v[70:71], v[88:89], v[106:107], v[124:125]\n\tv_pk_fma_f32 v[72:73], v[90:91], v[108:109], v[126:127]\n\tv_pk_fma_f32 v[74:75], v[92:93], v[110:111], v[112:113]\n\tv_pk_fma_f32 v[76:77], v[94:95], v[96:97], v[114:115]\n\tv_pk_fma_f32 v[78:79], v[80:81], v[98:99], v[116:117]\n\t")

typedef void (*kernel_t)(Stamp *, unsigned *, int);
struct Test {
    const char *name;
    kernel_t fn;
    int instr_per_iter;
    const char *note;
};

static const Test kTests[] = {
    {"v_add_f32 (16 indep.)", k_add_f32, 64, ""},
    {"v_add_f32 (8 chains)", k_add_f32_dep8, 64, "dependency distance 8"},
    {"v_add_f32 (4 chains)", k_add_f32_dep4, 64, "dependency distance 4"},
    {"v_add_f32 (2 chains)", k_add_f32_dep2, 64, "dependency distance 2"},
    {"v_add_f32 (1 chain)", k_add_f32_dep1, 64, "fully dependent"},
    {"v_mul_f32", k_mul_f32, 64, ""},
    {"v_fma_f32", k_fma_f32, 64, ""},
    {"v_add_f32 v,s,v (SGPR src)", k_add_f32_sgpr, 64, ""},
    {"v_add_f32_e64 |abs|", k_add_f32_e64, 64, "VOP3 encoding"},
    {"v_add_f32_sdwa (dword sel)", k_add_f32_sdwa, 64, ""},
    {"v_add_f32_dpp wave_shr:1", k_add_f32_dpp_wave_shr, 64, "Gaussian row pass"},
    {"v_add_f32_dpp wave_shr:1 (4 chains)", k_add_f32_dpp_wave_shr_dep4, 64, ""},
    {"v_add_f32_dpp row_shr:1", k_add_f32_dpp_row_shr, 64, ""},
    {"v_mov_b32_dpp wave_shr:1", k_mov_dpp_wave_shr, 64, "Sobel halo"},
    {"v_mov_b32_dpp row_shr:1", k_mov_dpp_row_shr, 64, ""},
    {"v_pk_mad_i16", k_pk_mad_i16, 64, "Sobel"},
    {"v_pk_sub_i16", k_pk_sub_i16, 64, "Sobel"},
    {"v_pk_max_i16", k_pk_max_i16, 64, ""},
    {"v_pk_add_f32", k_pk_add_f32, 64, ""},
    {"v_pk_mul_f32", k_pk_mul_f32, 64, ""},
    {"v_pk_fma_f32", k_pk_fma_f32, 64, ""},
    {"v_cndmask_b32", k_cndmask, 64, ""},
    {"v_cndmask_b32_sdwa WORD_1", k_cndmask_sdwa, 64, "Sobel+NMS output select"},
    {"v_cvt_f32_i32", k_cvt_f32_i32, 64, ""},
    {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte1, 64, ""},
    {"v_cvt_i32_f32", k_cvt_i32_f32, 64, ""},
    {"v_max3_i32", k_max3_i32, 64, "NMS"},
    {"v_add_u32", k_add_u32, 64, ""},
    {"v_and_or_b32", k_and_or_b32, 64, ""},
    {"v_perm_b32", k_perm_b32, 64, ""},
    {"v_mad_u32_u24", k_mad_u32_u24, 64, ""},
    {"v_mul_lo_u32", k_mul_lo_u32, 64, ""},
    {"v_sqrt_f32", k_sqrt_f32, 64, "magnitude"},
    {"v_rcp_f32", k_rcp_f32, 64, ""},
    {"v_cmp_gt_i32 (VCC)", k_cmp_gt_i32, 64, ""},
    {"v_cmp + s_nop 1 + v_addc", k_cmp_nop_addc, 64, "per TRIPLE (classify bit shift-in)"},
    {"s_nop 0", k_s_nop0, 64, ""},
    {"ds_read_b32", k_ds_read_b32, 64, "wait at the end of 64"},
    {"ds_read2st64_b32", k_ds_read2st64_b32, 64, "Gaussian product table"},
    {"ds_read_b64", k_ds_read_b64, 64, ""},
    {"ds_read_b128", k_ds_read_b128, 64, "edge-map table"},
    {"1 ds_read2st64 + 3 v_add_f32", k_ds2_plus_3add, 64, "per instruction of the mix"},
    {"v_add_f32 distinct src regs", k_add_f32_distinct_src, 64, "VGPR banks"},
    {"v_fma_f32 distinct src regs", k_fma_f32_distinct_src, 64, "VGPR banks"},
    {"v_add_f32 literal", k_add_f32_literal, 64, "8-byte VOP2"},
    {"v_add_f32 inline 1.0", k_add_f32_inline1, 64, ""},
    {"v_fmac_f32 (VOP2)", k_fmac_f32, 64, "3 reads, 4-byte encoding"},
    {"v_sub_u16", k_sub_u16, 64, ""},
    {"v_lshl_add_u32", k_lshl_add_u32, 64, ""},
    {"v_lshlrev_b32", k_lshlrev_b32, 64, ""},
    {"v_max_i32", k_max_i32, 64, ""},
    {"v_and_b32", k_and_b32, 64, ""},
    {"v_mov_b32", k_mov_b32, 64, ""},
    {"v_alignbit_b32", k_alignbit_b32, 64, "Sobel"},
    {"v_mul_i32_i24", k_mul_i32_i24, 64, ""},
    {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32, 64, ""},
    {"alternating v_add_f32 / v_add_f32_dpp", k_alt_add_dpp, 64, ""},
    {"alternating v_add_f32 / v_fma_f32", k_alt_add_fma, 64, ""},
    {"Gaussian mix: 64 VALU + 6 ds_read2st64", k_gauss_mix, 64, "per VALU instruction"},
    {"Gaussian mix without the LDS reads", k_gauss_mix_nods, 64, ""},
    {"Gaussian mix, DPP adds made plain, no LDS", k_gauss_mix_plain, 64, ""},
    // ---- second batch (index 62 on) ----
    {"v_max_f32", k_max_f32, 64, ""},
    {"v_sub_f32", k_sub_f32, 64, ""},
    {"v_or_b32", k_or_b32, 64, ""},
    {"v_xor_b32", k_xor_b32, 64, ""},
    {"v_min_u32", k_min_u32, 64, ""},
    {"v_ashrrev_i32", k_ashrrev_i32, 64, ""},
    {"v_bfe_u32", k_bfe_u32, 64, ""},
    {"v_lshl_or_b32", k_lshl_or_b32, 64, ""},
    {"v_add3_u32", k_add3_u32, 64, ""},
    {"v_bfi_b32", k_bfi_b32, 64, "select without a lane mask"},
    {"v_med3_i32", k_med3_i32, 64, ""},
    {"v_cvt_f32_u32", k_cvt_f32_u32, 64, ""},
    {"v_subrev_u32", k_subrev_u32, 64, ""},
    {"v_mul_u32_u24", k_mul_u32_u24, 64, ""},
    {"v_pk_add_u16", k_pk_add_u16, 64, ""},
    {"v_max_u16", k_max_u16, 64, ""},
    {"v_mul_f32 v,s,v (SGPR src)", k_mul_f32_sgpr, 64, ""},
    {"v_addc_co_u32 (VCC in/out)", k_addc_only, 64, ""},
    {"v_cmp + v_cndmask (vcc), per PAIR", k_cmp_cndmask, 64, "per pair"},
    {"v_cmp + s_nop 1 + v_cndmask (vcc), per TRIPLE", k_cmp_nop_cndmask, 64, "per triple"},
    {"v_cndmask_b32_e64 with SGPR-pair mask", k_cndmask_e64_sgpr, 64, ""},
    {"v_cmp_e64 sdst + s_nop 1 + v_cndmask_e64, per TRIPLE", k_cmp_e64_nop_cndmask_e64, 64, "per triple"},
    {"v_cmp_gt_i32_e64 s[20:21]", k_cmp_e64_sdst, 64, ""},
    {"ds_swizzle_b32", k_ds_swizzle, 64, ""},
    {"ds_bpermute_b32", k_ds_bpermute, 64, ""},
    {"1 ds_swizzle + 3 plain VALU", k_swizzle_plus_3_plain, 64, "per instruction of the mix"},
    {"1 ds_swizzle + 7 plain VALU", k_swizzle_plus_7_plain, 64, "per instruction of the mix"},
    {"52 plain then 12 DPP adds (grouped)", k_dpp_grouped_52_12, 64, ""},
    {"52 plain, 12 DPP adds spread singly", k_dpp_spread_52_12, 64, ""},
    {"plain VALU, every 4th with an SGPR operand", k_sgpr_every_4th, 64, ""},
    {"plain VALU, every 8th a v_cvt_i32_f32", k_cvt_every_8th, 64, ""},
    // ---- third batch (explicit registers) ----
    {"v_max_f32", k_max_f32, 64, ""},
    {"v_add_f32 in place, other bank", k_x_add_inplace, 64, "d = d + x"},
    {"v_add_f32 3 regs, 3 banks", k_x_add_3reg_3bank, 64, "d = a + b"},
    {"v_add_f32 3 regs, sources share a bank", k_x_add_src_same_bank, 64, ""},
    {"v_add_f32 3 regs, dst shares src0's bank", k_x_add_dst_src0_bank, 64, ""},
    {"v_add_f32 3 regs, chain (src0 = previous dst)", k_x_add_chain, 64, ""},
    {"v_mul_f32 3 regs, 3 banks", k_x_mul_3reg_3bank, 64, ""},
    {"v_max_u16 3 regs, 3 banks", k_x_maxu16_3reg_3bank, 64, ""},
    {"v_max_u16 3 regs, sources share a bank", k_x_maxu16_src_same_bank, 64, ""},
    {"v_max_i32 3 regs, 3 banks", k_x_maxi32_3reg_3bank, 64, ""},
    {"v_pk_sub_i16 3 regs, 3 banks", k_x_pksub_3reg_3bank, 64, ""},
    {"v_and_b32 3 regs, 3 banks", k_x_and_3reg_3bank, 64, ""},
    {"v_add_u32 3 regs, 3 banks", k_x_addu32_3reg_3bank, 64, ""},
    {"v_fma_f32 4 regs, 4 banks", k_x_fma_4reg_4bank, 64, ""},
    {"v_fma_f32 accumulate (d = a*b + d)", k_x_fma_acc, 64, ""},
    {"v_max3_i32 4 regs, 4 banks", k_x_max3_4reg_4bank, 64, ""},
    {"v_pk_mad_i16 4 regs, 4 banks", k_x_pkmad_4reg_4bank, 64, ""},
    {"v_pk_fma_f32 register pairs", k_x_pkfma_4bank, 64, ""},
};

struct Result {
    double cyc_wave;   // s_memtime ticks per instruction as one wave sees it (median over waves)
    double ns_simd;    // wall-clock nanoseconds of SIMD time per wave-instruction = launch time x SIMDs / wave-instructions
    double clock_ghz;  // in-kernel clock: d(s_memtime) / d(s_memrealtime) x 100 MHz (median over waves)
    double resident;   // waves per SIMD that really ran together (time-weighted overlap of the stamp intervals)
    double cyc_simd() const { return ns_simd * clock_ghz; }
};

struct Shape {
    int waves_per_simd, threads, blocks_per_cu;
    size_t lds;
};
// one block per CU needs more than half of the CU's 160 KB of LDS; two blocks per CU just under half
static const Shape kShapes[] = {
    {1, 256, 1, 100 * 1024}, {2, 512, 1, 100 * 1024}, {3, 768, 1, 100 * 1024},
    {4, 1024, 1, 100 * 1024}, {6, 768, 2, 72 * 1024}, {8, 1024, 2, 72 * 1024},
};
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

static Result run(const Test &t, const Shape &sh, int iters, Stamp *d_stamps, unsigned *d_sink, int cus)
{
    const int grid = cus * sh.blocks_per_cu;
    CHECK(hipFuncSetAttribute((const void *)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds));
    const int n_waves = grid * sh.threads / 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(t.fn, dim3(grid), dim3(sh.threads), sh.lds, 0, d_stamps, d_sink, iters / 4 + 1); // warm
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(t.fn, dim3(grid), dim3(sh.threads), sh.lds, 0, d_stamps, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> st(n_waves);
    CHECK(hipMemcpy(st.data(), d_stamps, sizeof(Stamp) * n_waves, hipMemcpyDeviceToHost));
    std::vector<double> per, clk;
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> by_simd;
    for (auto &s : st) {
        per.push_back((double)(s.t1 - s.t0) / ((double)iters * t.instr_per_iter));
        if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
        // (xcc, se, sh, cu, simd): HW_ID bits 15:8 and 5:4
        unsigned long long key = ((unsigned long long)(s.xcc_id & 0xf) << 32) | (s.hw_id & 0xff30u);
        by_simd[key].push_back({s.r0, s.r1});
    }
    std::sort(per.begin(), per.end());
    std::sort(clk.begin(), clk.end());
    // residency: sum of interval lengths / length of their union, per SIMD (realtime stamps share one clock)
    double res_sum = 0;
    for (auto &kv : by_simd) {
        auto v = kv.second;
        std::sort(v.begin(), v.end());
        unsigned long long total = 0, uni = 0, cur_b = v[0].first, cur_e = v[0].second;
        for (auto &iv : v) {
            total += iv.second - iv.first;
            if (iv.first > cur_e) {
                uni += cur_e - cur_b;
                cur_b = iv.first;
                cur_e = iv.second;
            } else if (iv.second > cur_e) {
                cur_e = iv.second;
            }
        }
        uni += cur_e - cur_b;
        res_sum += uni ? (double)total / (double)uni : 0.0;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    Result r;
    r.cyc_wave = per[per.size() / 2];
    r.clock_ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    r.resident = res_sum / by_simd.size();
    const double wave_instrs = (double)n_waves * iters * t.instr_per_iter;
    r.ns_simd = (double)ms * 1e6 * (double)by_simd.size() / wave_instrs;
    return r;
}

int main(int argc, char **argv)
{
    int iters = 12000; // 768 k instructions per wave: 1.5 - 8 ms per launch
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    Stamp *d_stamps;
    unsigned *d_sink;
    CHECK(hipMalloc(&d_stamps, sizeof(Stamp) * cus * 2 * 16));
    CHECK(hipMalloc(&d_sink, 64));
    std::printf("%s, %d CUs; per shape: [SIMD cycles per wave-instruction | ns of SIMD time per wave-instruction | "
                "in-kernel GHz | waves per SIMD really resident]\n", prop.name, cus);
    std::printf("%-42s", "instruction (wave64) \\ waves per SIMD");
    for (auto &sh : kShapes) std::printf(" | %-25d", sh.waves_per_simd);
    std::printf("\n");
    std::string json = "{\n  \"_shapes_waves_per_simd\": [1, 2, 3, 4, 6, 8],\n";
    const int first = argc > 2 ? std::atoi(argv[2]) : 0;
    int index = 0;
    for (const Test &t : kTests) {
        if (index++ < first) continue;
        Result r[kNumShapes];
        for (int i = 0; i < kNumShapes; i++) r[i] = run(t, kShapes[i], iters, d_stamps, d_sink, cus);
        std::printf("%-42s", t.name);
        for (int i = 0; i < kNumShapes; i++)
            std::printf(" | %5.2f %5.2fns %4.2fG %4.2f", r[i].cyc_simd(), r[i].ns_simd, r[i].clock_ghz, r[i].resident);
        std::printf("  %s\n", t.note);
        std::fflush(stdout);
        json += std::string("  \"") + t.name + "\": {";
        const char *keys[] = {"simd_cycles_per_instr", "simd_ns_per_instr", "clock_ghz", "resident_waves_per_simd",
                              "wave_cycles_per_instr"};
        for (int k = 0; k < 5; k++) {
            json += std::string("\"") + keys[k] + "\": [";
            for (int i = 0; i < kNumShapes; i++) {
                char buf[64];
                double v = k == 0 ? r[i].cyc_simd() : k == 1 ? r[i].ns_simd : k == 2 ? r[i].clock_ghz
                           : k == 3 ? r[i].resident : r[i].cyc_wave;
                std::snprintf(buf, sizeof buf, "%s%.3f", i ? ", " : "", v);
                json += buf;
            }
            json += k < 4 ? "], " : "]";
        }
        json += "},\n";
    }
    json += "  \"_note\": \"see tools/valu_issue_bench.hip\"\n}\n";
    if (argc > 1) {
        FILE *f = std::fopen(argv[1], "w");
        if (f) {
            std::fputs(json.c_str(), f);
            std::fclose(f);
        }
    }
    return 0;
}
