// probe_host_link.hip -- what the host side of the batch pipeline can reach on this box:
//   * PCIe H2D / D2H rates from pinned memory, alone and both directions at once (chunk sizes 8..256 MB)
//   * hipHostRegister / hipHostUnregister cost per GB of a touched pageable buffer (whole buffer and per chunk)
//   * host memcpy into pinned staging with 1..8 threads
//   * hipMemcpyAsync straight from pageable memory (the runtime's own staging)
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_host_link probe_host_link.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                            \
        }                                                                                            \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const size_t GB = 1ull << 30, MB = 1ull << 20;
    const size_t total = 2 * GB;
    void *d_a, *d_b;
    CHECK(hipMalloc(&d_a, total));
    CHECK(hipMalloc(&d_b, total));
    char *pin_a, *pin_b;
    CHECK(hipHostMalloc((void **)&pin_a, total));
    CHECK(hipHostMalloc((void **)&pin_b, total));
    std::memset(pin_a, 1, total);
    std::memset(pin_b, 2, total);
    hipStream_t s0, s1;
    CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));

    std::printf("threads available: %u\n", std::thread::hardware_concurrency());
    for (size_t chunk : {8 * MB, 24 * MB, 64 * MB, 256 * MB}) {
        const int n = (int)(total / chunk);
        // warm
        CHECK(hipMemcpyAsync(d_a, pin_a, chunk, hipMemcpyHostToDevice, s0));
        CHECK(hipMemcpyAsync(pin_b, d_b, chunk, hipMemcpyDeviceToHost, s1));
        CHECK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < n; i++)
            CHECK(hipMemcpyAsync((char *)d_a + i * chunk, pin_a + i * chunk, chunk, hipMemcpyHostToDevice, s0));
        CHECK(hipStreamSynchronize(s0));
        double t_h2d = now() - t0;
        t0 = now();
        for (int i = 0; i < n; i++)
            CHECK(hipMemcpyAsync(pin_b + i * chunk, (char *)d_b + i * chunk, chunk, hipMemcpyDeviceToHost, s1));
        CHECK(hipStreamSynchronize(s1));
        double t_d2h = now() - t0;
        t0 = now();
        for (int i = 0; i < n; i++) {
            CHECK(hipMemcpyAsync((char *)d_a + i * chunk, pin_a + i * chunk, chunk, hipMemcpyHostToDevice, s0));
            CHECK(hipMemcpyAsync(pin_b + i * chunk, (char *)d_b + i * chunk, chunk, hipMemcpyDeviceToHost, s1));
        }
        CHECK(hipDeviceSynchronize());
        double t_both = now() - t0;
        std::printf("pinned, %4zu MB chunks: H2D %.1f GB/s  D2H %.1f GB/s  both at once %.1f + %.1f GB/s\n", chunk / MB,
                    total / t_h2d / 1e9, total / t_d2h / 1e9, total / t_both / 1e9, total / t_both / 1e9);
        std::fflush(stdout);
    }
    // D2H twice the bytes of H2D (the s16 pipeline): 1 GB in, 2 GB out
    {
        const size_t chunk = 24 * MB;
        const int n = (int)(GB / chunk);
        double t0 = now();
        for (int i = 0; i < n; i++) {
            CHECK(hipMemcpyAsync((char *)d_a + i * chunk, pin_a + i * chunk, chunk, hipMemcpyHostToDevice, s0));
            CHECK(hipMemcpyAsync(pin_b + 2 * i * chunk, (char *)d_b + 2 * i * chunk, 2 * chunk, hipMemcpyDeviceToHost, s1));
        }
        CHECK(hipDeviceSynchronize());
        double t = now() - t0;
        std::printf("pinned, 1:2 mix (24 MB in, 48 MB out per step): %.1f GB/s in + %.1f GB/s out\n", n * chunk / t / 1e9,
                    2.0 * n * chunk / t / 1e9);
    }

    // pageable buffer, touched
    char *pg = (char *)std::malloc(total);
    std::memset(pg, 3, total);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CHECK(hipHostRegister(pg, total, hipHostRegisterDefault));
        double t_reg = now() - t0;
        t0 = now();
        CHECK(hipMemcpyAsync(d_a, pg, total, hipMemcpyHostToDevice, s0));
        CHECK(hipStreamSynchronize(s0));
        double t_cp = now() - t0;
        t0 = now();
        CHECK(hipHostUnregister(pg));
        double t_unreg = now() - t0;
        std::printf("hipHostRegister 2 GB: %.1f ms (%.1f GB/s), H2D from it %.1f GB/s, unregister %.1f ms\n", t_reg * 1e3,
                    total / t_reg / 1e9, total / t_cp / 1e9, t_unreg * 1e3);
    }
    for (size_t chunk : {16 * MB, 64 * MB}) {
        const int n = (int)(total / chunk);
        double t0 = now();
        for (int i = 0; i < n; i++) CHECK(hipHostRegister(pg + i * chunk, chunk, hipHostRegisterDefault));
        double t_reg = now() - t0;
        t0 = now();
        for (int i = 0; i < n; i++) CHECK(hipHostUnregister(pg + i * chunk));
        double t_unreg = now() - t0;
        std::printf("hipHostRegister per %zu MB chunk: %.2f ms each (%.1f GB/s), unregister %.2f ms each\n", chunk / MB,
                    t_reg / n * 1e3, total / t_reg / 1e9, t_unreg / n * 1e3);
    }
    {
        double t0 = now();
        CHECK(hipMemcpyAsync(d_a, pg, total, hipMemcpyHostToDevice, s0));
        CHECK(hipStreamSynchronize(s0));
        double t1 = now() - t0;
        t0 = now();
        CHECK(hipMemcpyAsync(pg, d_b, total, hipMemcpyDeviceToHost, s0));
        CHECK(hipStreamSynchronize(s0));
        double t2 = now() - t0;
        std::printf("hipMemcpyAsync from/to PAGEABLE memory (runtime staging): H2D %.1f GB/s, D2H %.1f GB/s\n",
                    total / t1 / 1e9, total / t2 / 1e9);
    }
    for (int nt : {1, 2, 4, 8}) {
        std::vector<std::thread> th;
        double t0 = now();
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                size_t part = total / nt;
                std::memcpy(pin_a + t * part, pg + t * part, part);
            });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        std::printf("memcpy pageable -> pinned, %d threads: %.1f GB/s\n", nt, total / dt / 1e9);
    }
    // single-frame latencies: 4K u8 in (8.3 MB), s16 out (16.6 MB), u8 out (8.3 MB)
    {
        const size_t in = 3840 * 2160, out16 = in * 2;
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now();
            CHECK(hipMemcpyAsync(d_a, pin_a, in, hipMemcpyHostToDevice, s0));
            CHECK(hipStreamSynchronize(s0));
            double a = now() - t0;
            t0 = now();
            CHECK(hipMemcpyAsync(pin_b, d_b, out16, hipMemcpyDeviceToHost, s0));
            CHECK(hipStreamSynchronize(s0));
            double b = now() - t0;
            t0 = now();
            CHECK(hipMemcpyAsync(pin_b, d_b, in, hipMemcpyDeviceToHost, s0));
            CHECK(hipStreamSynchronize(s0));
            double c = now() - t0;
            std::printf("one 4K frame, pinned: H2D u8 %.3f ms, D2H s16 %.3f ms, D2H u8 %.3f ms\n", a * 1e3, b * 1e3, c * 1e3);
        }
    }
    return 0;
}
