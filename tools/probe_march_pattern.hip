// probe_march_pattern.hip -- what does the MEMORY ACCESS PATTERN of the marching Sobel+NMS kernel cost by itself?
//
// The fused Sobel+NMS kernel reads 2 B/px and writes 2 B/px; round 3 found that replacing two thirds of its slow-class
// VALU instructions and raising its occupancy from 3 to 4 waves per SIMD changed nothing.  This probe runs kernels
// that do the same loads and stores with (almost) no arithmetic, in several geometries, so that the pattern's own
// ceiling is known:
//   strip geometry   "62x8": 62 owner lanes + 2 halo lanes, 8 px (16 B) per lane -- strips advance by 992 B, so
//                            neither a wave's 1 KB row load nor its 992 B row store is aligned to 128 B lines
//                    "64x8": 64 owner lanes, no halo lanes: 1 KB aligned loads and stores (what a kernel that got its
//                            two halo columns some other way would do)
//                    "60x8": 60 owner lanes + 2 halo lanes each side: strips advance by 960 B (64 B aligned)
//   march            every wave walks `seg` rows (+4 halo rows) of its strip, one row per iteration, prefetching
//                    `ahead` rows; the stored row is the OR of the three last loaded rows (a dependency like the
//                    stencil's, no VALU cost to speak of)
//   tile             for comparison: a plain grid-stride copy of the same bytes (16 B per lane, fully coalesced)
// Usage: probe_march_pattern [frames=128] [H=2160] [W=3840] [reps=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                            \
        }                                                                                            \
    } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Geo {
    int H, W, n_strips, n_segs, seg, total_waves, sw, hl, xcd_remap;
};

// OWN = owner lanes per wave, HL = halo lanes per side, AHEAD = rows prefetched (1..3), STORE = 0 none / 1 all
template <int AHEAD, bool STORE, bool LOAD, int WPB = 4, int SYNC = 0>
__global__ __launch_bounds__(WPB * 64) void march_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, Geo g)
{
    unsigned bid = blockIdx.x;
    if (g.xcd_remap) {
        const unsigned q = gridDim.x / 8u, r = gridDim.x % 8u, xcd = bid % 8u;
        bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + bid / 8u;
    }
    const int wave = __builtin_amdgcn_readfirstlane(bid * WPB + (threadIdx.x >> 6));
    if (wave >= g.total_waves) return;
    const int lane = threadIdx.x & 63;
    const int s = wave % g.n_strips;
    const int sg = (wave / g.n_strips) % g.n_segs;
    const int f = wave / (g.n_strips * g.n_segs);
    const int ybeg = sg * g.seg, yend = min(g.H, ybeg + g.seg);
    const int x0 = s * g.sw + (lane - g.hl) * 8;
    const bool in_img = x0 >= 0 && x0 + 7 < g.W;
    const bool owner = lane >= g.hl && lane < 64 - g.hl && in_img;
    const size_t fbase = (size_t)f * g.H * g.W;
    const int16_t *src = (const int16_t *)in + fbase;
    int16_t *dst = (int16_t *)out + fbase;
    auto load = [&](int r) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        r = min(max(r, 0), g.H - 1);
        if (LOAD && in_img) v = *reinterpret_cast<const uint4 *>(src + (size_t)r * g.W + x0);
        return v;
    };
    uint4 q[AHEAD + 1];
    uint4 p1 = make_uint4(0, 0, 0, 0), p2 = p1;
    const int rfirst = ybeg - 2;
#pragma unroll
    for (int k = 0; k < AHEAD; k++) q[k] = load(rfirst + k);
    for (int r = rfirst; r <= yend + 1; r++) {
        if (SYNC && ((r - rfirst) % SYNC) == 0) __syncthreads(); // keeps the workgroup's waves on the same rows
        q[AHEAD] = load(r + AHEAD);
        const uint4 cur = q[0];
#pragma unroll
        for (int k = 0; k < AHEAD; k++) q[k] = q[k + 1];
        uint4 o;
        o.x = cur.x | p1.x | p2.x;
        o.y = cur.y | p1.y | p2.y;
        o.z = cur.z | p1.z | p2.z;
        o.w = cur.w | p1.w | p2.w;
        p2 = p1;
        p1 = cur;
        const int y = r - 2;
        if (y >= ybeg && y < yend) {
            if (STORE) {
                if (owner) *reinterpret_cast<uint4 *>(dst + (size_t)y * g.W + x0) = o;
            } else {
                asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
            }
        }
    }
}

// The marching pattern (62x8, 2 rows ahead) plus OPS VALU instructions per pixel and row, KIND 0: v_fma_f32 (plain f32,
// the "fast class" of tools/valu_issue_bench.hip), 1: v_bfi_b32 (VOP3 integer, "slow class"), 2: v_pk_add_u16;
// WAVES = waves per SIMD, enforced by the launch's dynamic LDS size (160 KB / WAVES per 4-wave workgroup).
template <int OPS, int KIND, int WAVES>
__global__ __launch_bounds__(256) void march_valu_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, Geo g)
{
    unsigned bid = blockIdx.x;
    {
        const unsigned q = gridDim.x / 8u, r = gridDim.x % 8u, xcd = bid % 8u;
        bid = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + bid / 8u;
    }
    const int wave = __builtin_amdgcn_readfirstlane(bid * 4 + (threadIdx.x >> 6));
    if (wave >= g.total_waves) return;
    const int lane = threadIdx.x & 63;
    const int s = wave % g.n_strips;
    const int sg = (wave / g.n_strips) % g.n_segs;
    const int f = wave / (g.n_strips * g.n_segs);
    const int ybeg = sg * g.seg, yend = min(g.H, ybeg + g.seg);
    const int x0 = s * g.sw + (lane - g.hl) * 8;
    const bool in_img = x0 >= 0 && x0 + 7 < g.W;
    const bool owner = lane >= g.hl && lane < 64 - g.hl && in_img;
    const size_t fbase = (size_t)f * g.H * g.W;
    const int16_t *src = (const int16_t *)in + fbase;
    int16_t *dst = (int16_t *)out + fbase;
    auto load = [&](int r) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        r = min(max(r, 0), g.H - 1);
        if (in_img) v = *reinterpret_cast<const uint4 *>(src + (size_t)r * g.W + x0);
        return v;
    };
    uint4 q0 = load(ybeg - 2), q1 = load(ybeg - 1);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 1.0f + e;
    for (int r = ybeg - 2; r <= yend + 1; r++) {
        const uint4 q2 = load(r + 2);
        const uint4 cur = q0;
        q0 = q1;
        q1 = q2;
        const float m = __uint_as_float((cur.x & 0x007fffffu) | 0x3f800000u); // some value in [1, 2) that depends on the load
#pragma unroll
        for (int k = 0; k < OPS; k++) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (KIND == 0) {
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc[e]) : "v"(m));
                } else if (KIND == 1) {
                    asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(acc[e]) : "v"(m));
                } else {
                    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(acc[e]) : "v"(m));
                }
            }
        }
        const int y = r - 2;
        if (y >= ybeg && y < yend && owner) {
            uint4 o;
            o.x = __float_as_uint(acc[0]) ^ __float_as_uint(acc[1]);
            o.y = __float_as_uint(acc[2]) ^ __float_as_uint(acc[3]);
            o.z = __float_as_uint(acc[4]) ^ __float_as_uint(acc[5]);
            o.w = __float_as_uint(acc[6]) ^ __float_as_uint(acc[7]);
            *reinterpret_cast<uint4 *>(dst + (size_t)y * g.W + x0) = o;
        }
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}

// each thread moves U x 16 B per trip, the U loads issued before the first store; NT: non-temporal accesses
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_unrolled_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) {
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + i + k * stride));
                v[k] = make_uint4(t[0], t[1], t[2], t[3]);
            } else {
                v[k] = in[i + k * stride];
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) {
                u32x4 t;
                t[0] = v[k].x; t[1] = v[k].y; t[2] = v[k].z; t[3] = v[k].w;
                __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(out + i + k * stride));
            } else
                out[i + k * stride] = v[k];
        }
    }
    for (; i < n16; i += stride) out[i] = in[i];
}
// a workgroup owns a contiguous chunk (consecutive blocks = consecutive chunks): the layout of a tiled kernel
template <int U>
__global__ __launch_bounds__(256) void copy_chunk_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16)
{
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = (i + k * 256 < n16) ? in[i + k * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < U; k++)
        if (i + k * 256 < n16) out[i + k * 256] = v[k];
}
__global__ __launch_bounds__(256) void read_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = in[i];
        acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc; // never true for the fill pattern
}
__global__ __launch_bounds__(256) void write_kernel(uint4 *__restrict__ out, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = v;
}

template <class L>
static double time_ms(L &&launch, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <class K>
static void allow_big_lds(K kern)
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

int main(int argc, char **argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 2160, W = argc > 3 ? atoi(argv[3]) : 3840;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const size_t px = (size_t)F * H * W, bytes = px * 2;
    uint4 *d_in, *d_out;
    CHECK(hipMalloc(&d_in, bytes + 4096));
    CHECK(hipMalloc(&d_out, bytes + 4096));
    CHECK(hipMemset(d_in, 1, bytes));
    CHECK(hipMemset(d_out, 0, bytes));
    // spin the clocks up
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, d_in, d_out, bytes / 16);
    CHECK(hipDeviceSynchronize());
    const double alg = 4.0 * px;
    auto report = [&](const char *name, double ms, double frac_bytes = 1.0) {
        std::printf("%-58s %8.4f ms  %7.1f GB/s of 4 B/px  (%4.1f %% of 8 TB/s)\n", name, ms, alg * frac_bytes / ms / 1e6,
                    alg * frac_bytes / ms / 8e9 * 100);
        std::fflush(stdout);
    };
    for (int blocks : {2048, 4096, 8192})
        {
            char nm[96];
            std::snprintf(nm, sizeof nm, "grid-stride copy, %d blocks", blocks);
            report(nm, time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_out, bytes / 16); }, reps));
        }
    {
        const size_t n16 = bytes / 16;
        report("copy, 4 x 16 B in flight per thread, 2048 blocks", time_ms([&] { hipLaunchKernelGGL((copy_unrolled_kernel<4, false>), dim3(2048), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, 4 x 16 B in flight per thread, 1024 blocks", time_ms([&] { hipLaunchKernelGGL((copy_unrolled_kernel<4, false>), dim3(1024), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, 8 x 16 B in flight per thread, 2048 blocks", time_ms([&] { hipLaunchKernelGGL((copy_unrolled_kernel<8, false>), dim3(2048), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, 4 x 16 B, non-temporal, 2048 blocks", time_ms([&] { hipLaunchKernelGGL((copy_unrolled_kernel<4, true>), dim3(2048), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, block = contiguous 16 KB chunk", time_ms([&] { hipLaunchKernelGGL((copy_chunk_kernel<4>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, block = contiguous 32 KB chunk", time_ms([&] { hipLaunchKernelGGL((copy_chunk_kernel<8>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, block = contiguous 4 KB chunk", time_ms([&] { hipLaunchKernelGGL((copy_chunk_kernel<1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, block = contiguous 8 KB chunk", time_ms([&] { hipLaunchKernelGGL((copy_chunk_kernel<2>), dim3((unsigned)((n16 + 511) / 512)), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("copy, block = contiguous 64 KB chunk", time_ms([&] { hipLaunchKernelGGL((copy_chunk_kernel<16>), dim3((unsigned)((n16 + 4095) / 4096)), dim3(256), 0, 0, d_in, d_out, n16); }, reps));
        report("read only (2 B/px)", time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, d_in, d_out, n16); }, reps), 0.5);
        report("write only (2 B/px)", time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, d_out, n16); }, reps), 0.5);
    }
    {
        Geo g;
        g.H = H; g.W = W; g.sw = 496; g.hl = 1; g.seg = 64;
        g.n_strips = (W + g.sw - 1) / g.sw;
        g.n_segs = (H + g.seg - 1) / g.seg;
        g.total_waves = F * g.n_strips * g.n_segs;
        g.xcd_remap = 1;
        const unsigned blocks = (g.total_waves + 3) / 4;
        char nm[128];
#define RUN_VALU(OPS, KIND, WAVES)                                                                                     \
    allow_big_lds(march_valu_kernel<OPS, KIND, WAVES>);                                                                \
    std::snprintf(nm, sizeof nm, "march 62x8 seg 64 + %2d %s per px, %d waves/SIMD", OPS,                              \
                  KIND == 0 ? "v_fma_f32" : KIND == 1 ? "v_bfi_b32" : "v_pk_add_u16", WAVES);                          \
    report(nm, time_ms([&] { hipLaunchKernelGGL((march_valu_kernel<OPS, KIND, WAVES>), dim3(blocks), dim3(256), (160 * 1024 / WAVES) & ~1023, 0, d_in, d_out, g); }, reps));
        RUN_VALU(0, 0, 4) RUN_VALU(8, 0, 4) RUN_VALU(16, 0, 4) RUN_VALU(24, 0, 4) RUN_VALU(32, 0, 4) RUN_VALU(40, 0, 4)
        RUN_VALU(8, 1, 4) RUN_VALU(16, 1, 4) RUN_VALU(24, 1, 4) RUN_VALU(32, 1, 4)
        RUN_VALU(16, 2, 4) RUN_VALU(32, 2, 4)
        RUN_VALU(16, 0, 3) RUN_VALU(24, 0, 3) RUN_VALU(32, 0, 3)
        RUN_VALU(16, 0, 8) RUN_VALU(24, 0, 8) RUN_VALU(32, 0, 8)
        RUN_VALU(16, 0, 2) RUN_VALU(32, 0, 2)
#undef RUN_VALU
    }
    {   // workgroups that span the whole row: 8 strips of 496 px = 3968 >= 3840
        Geo g;
        g.H = H; g.W = W; g.sw = 496; g.hl = 1;
        g.n_strips = (W + g.sw - 1) / g.sw;
        g.xcd_remap = 1;
        char nm[128];
        for (int seg : {8, 16, 32, 64, 128}) {
            g.seg = seg;
            g.n_segs = (H + seg - 1) / seg;
            g.total_waves = F * g.n_strips * g.n_segs;
#define RUN_WPB(WPB, SYNC)                                                                                             \
    if (g.total_waves % WPB == 0) {                                                                                    \
        std::snprintf(nm, sizeof nm, "march 62x8 seg %3d ld+st ahead2, %2d waves per workgroup, barrier every %d rows", seg, WPB, SYNC); \
        report(nm, time_ms([&] { hipLaunchKernelGGL((march_kernel<2, true, true, WPB, SYNC>), dim3(g.total_waves / WPB), dim3(WPB * 64), 0, 0, d_in, d_out, g); }, reps)); \
    }
            RUN_WPB(4, 0) RUN_WPB(8, 0) RUN_WPB(8, 1) RUN_WPB(8, 4) RUN_WPB(8, 16) RUN_WPB(16, 0) RUN_WPB(16, 4) RUN_WPB(2, 0) RUN_WPB(1, 0)
#undef RUN_WPB
        }
    }
    struct Cfg {
        const char *name;
        int sw, hl;
    } cfgs[] = {{"62x8 (992 B strips, 2 halo lanes)", 496, 1}, {"64x8 (1 KB strips, no halo)", 512, 0},
                {"60x8 (960 B strips, 4 halo lanes)", 480, 2}};
    for (const Cfg &c : cfgs) {
        for (int seg : {32, 64, 128, 256}) {
            for (int remap : {1, 0}) {
                if (remap == 0 && seg != 64) continue;
                Geo g;
                g.H = H;
                g.W = W;
                g.sw = c.sw;
                g.hl = c.hl;
                g.seg = seg;
                g.n_strips = (W + c.sw - 1) / c.sw;
                g.n_segs = (H + seg - 1) / seg;
                g.total_waves = F * g.n_strips * g.n_segs;
                g.xcd_remap = remap;
                const unsigned blocks = (g.total_waves + 3) / 4;
                char nm[128];
                auto run = [&](const char *what, auto kern, double fb) {
                    std::snprintf(nm, sizeof nm, "%s seg %3d %s %s", c.name, seg, remap ? "xcd" : "rr ", what);
                    report(nm, time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_in, d_out, g); }, reps), fb);
                };
                run("ld+st ahead2", march_kernel<2, true, true>, 1.0);
                if (seg == 64 && remap) {
                    run("ld+st ahead1", march_kernel<1, true, true>, 1.0);
                    run("ld+st ahead3", march_kernel<3, true, true>, 1.0);
                    run("ld only     ", march_kernel<2, false, true>, 0.5);
                    run("st only     ", march_kernel<2, true, false>, 0.5);
                }
            }
        }
    }
    return 0;
}
