// How does buffer_store_format_xyzw convert f32 -> 8_8_8_8 USCALED / UINT on gfx950?  (Could the Gaussian's
// v_cvt_i32_f32 + byte packing be left to the store?)  Exhaustive over every float in [0, 256): compares the stored
// byte with the truncating cast the reference needs (src/utils.cpp:62).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_store_format.hip -o tools/bin/probe_store_format
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void store_fmt(const float *in, uint8_t *out, unsigned n4, unsigned word3)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = {in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]};
    i32x4 rsrc;
    const uint64_t base = (uint64_t)out;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)); // stride 0
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)(4 * n4));               // num_records (bytes, stride 0)
    rsrc[3] = __builtin_amdgcn_readfirstlane((int)word3);
    const unsigned off = 4 * i;
    asm volatile("buffer_store_format_xyzw %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(rsrc) : "memory");
}

int main()
{
    // all floats in [0, 256): bit patterns 0 .. 0x43800000 (exclusive), plus a few above
    const unsigned first = 0x3a000000u, last = 0x43800000u; // below 2^-11 everything is 0 anyway; checked separately
    const unsigned n = last - first, n4 = n / 4;
    std::vector<float> h(n);
    for (unsigned k = 0; k < n; k++) {
        unsigned b = first + k;
        std::memcpy(&h[k], &b, 4);
    }
    float *d_in;
    uint8_t *d_out;
    hipMalloc(&d_in, (size_t)n * 4);
    hipMalloc(&d_out, n);
    hipMemcpy(d_in, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    const unsigned dst_sel = 4 | (5 << 3) | (6 << 6) | (7 << 9);
    struct { const char *name; unsigned nf; } fmts[] = {{"USCALED", 2}, {"UINT", 4}, {"UNORM", 0}};
    for (auto &f : fmts) {
        hipMemset(d_out, 0xEE, n);
        const unsigned word3 = dst_sel | (f.nf << 12) | (10u << 15);
        hipLaunchKernelGGL(store_fmt, dim3((n4 + 255) / 256), dim3(256), 0, 0, d_in, d_out, n4, word3);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", f.name); return 1; }
        std::vector<uint8_t> o(n);
        hipMemcpy(o.data(), d_out, n, hipMemcpyDeviceToHost);
        unsigned long long bad_trunc = 0, bad_rne = 0;
        unsigned first_bad = 0;
        for (unsigned k = 0; k < 4 * n4; k++) {
            const float x = h[k];
            const unsigned t = (unsigned)x;
            unsigned r = (unsigned)__builtin_rintf(x);
            if (r > 255) r = 255;
            if (o[k] != t) { if (!bad_trunc) first_bad = k; bad_trunc++; }
            if (o[k] != r) bad_rne++;
        }
        printf("%-8s word3=%#x: differs from truncation %llu, from round-to-nearest-even %llu of %u", f.name, word3,
               bad_trunc, bad_rne, 4 * n4);
        if (bad_trunc) printf("  (first: x=%.9g stored %u trunc %u)", h[first_bad], o[first_bad], (unsigned)h[first_bad]);
        printf("\n");
    }
    return 0;
}
