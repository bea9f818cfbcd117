// probe_pipe_patterns.hip -- which stream structure keeps both PCIe directions busy around a short kernel?
// Chunk = H2D of `in_mb`, a kernel that spins for ~kernel_us, D2H of `out_mb`; 64 chunks; pinned host memory.
//   P0  one stream, everything in order (serial reference)
//   P1  three streams (h2d / compute / d2h) chained by events, fully asynchronous enqueue, 3 slots
//   P2  like P1, but the host blocks until each chunk's kernel has finished before it enqueues that chunk's D2H
//       (what canny() forces: the hysteresis convergence poll)
//   P3  W host threads, one in-order stream each (round 1's structure), host blocks on its kernel
//   P4  like P2 with W threads (each its own three streams)
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_pipe_patterns probe_pipe_patterns.hip -lpthread
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

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin_kernel(const unsigned char *in, short *out, long long cycles)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (in && out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = in[0];
}

struct Pipe {
    hipStream_t h2d, comp, d2h;
    static constexpr int K = 3;
    void *d_in[K], *d_out[K];
    hipEvent_t e_h2d[K], e_comp[K], e_d2h[K];
};

static void make_pipe(Pipe &p, size_t in_b, size_t out_b)
{
    CHECK(hipStreamCreateWithFlags(&p.h2d, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&p.comp, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&p.d2h, hipStreamNonBlocking));
    for (int k = 0; k < Pipe::K; k++) {
        CHECK(hipMalloc(&p.d_in[k], in_b));
        CHECK(hipMalloc(&p.d_out[k], out_b));
        CHECK(hipEventCreateWithFlags(&p.e_h2d[k], hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&p.e_comp[k], hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&p.e_d2h[k], hipEventDisableTiming));
    }
}

int main(int argc, char **argv)
{
    const size_t MB = 1ull << 20;
    const int n_chunks = 64;
    const double kernel_us = argc > 1 ? atof(argv[1]) : 150.0;
    const long long cycles = (long long)(kernel_us * 100.0); // clock64 ~ 100 MHz constant counter on gfx9
    for (int cfg = 0; cfg < 2; cfg++) {
        const size_t in_b = 16 * MB, out_b = cfg == 0 ? 32 * MB : 16 * MB;
        std::printf("---- chunk: %zu MB in, %zu MB out, kernel ~%.0f us, %d chunks ----\n", in_b / MB, out_b / MB, kernel_us,
                    n_chunks);
        char *h_in, *h_out;
        CHECK(hipHostMalloc((void **)&h_in, in_b * n_chunks));
        CHECK(hipHostMalloc((void **)&h_out, out_b * n_chunks));
        std::memset(h_in, 1, in_b * n_chunks);
        std::memset(h_out, 2, out_b * n_chunks);
        std::vector<Pipe> pipes(4);
        for (auto &p : pipes) make_pipe(p, in_b, out_b);
        auto report = [&](const char *name, double t) {
            std::printf("%-58s %7.1f ms   H2D %5.1f GB/s  D2H %5.1f GB/s\n", name, t * 1e3, in_b * n_chunks / t / 1e9,
                        out_b * n_chunks / t / 1e9);
            std::fflush(stdout);
        };
        // kernel time check
        {
            Pipe &p = pipes[0];
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, p.comp, nullptr, nullptr, cycles);
            CHECK(hipStreamSynchronize(p.comp));
            double t0 = now();
            for (int i = 0; i < 10; i++) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, p.comp, nullptr, nullptr, cycles);
                CHECK(hipStreamSynchronize(p.comp));
            }
            std::printf("kernel + sync round trip: %.1f us\n", (now() - t0) / 10 * 1e6);
        }
        for (int rep = 0; rep < 2; rep++) {
            // P0
            {
                Pipe &p = pipes[0];
                double t0 = now();
                for (int c = 0; c < n_chunks; c++) {
                    CHECK(hipMemcpyAsync(p.d_in[0], h_in + c * in_b, in_b, hipMemcpyHostToDevice, p.comp));
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, p.comp, (unsigned char *)p.d_in[0],
                                       (short *)p.d_out[0], cycles);
                    CHECK(hipMemcpyAsync(h_out + c * out_b, p.d_out[0], out_b, hipMemcpyDeviceToHost, p.comp));
                }
                CHECK(hipStreamSynchronize(p.comp));
                report("P0 one stream, in order", now() - t0);
            }
            // P1 / P2 on one pipe, P4 on W pipes
            auto run3 = [&](Pipe &p, int first, int step, bool host_blocks) {
                bool used[Pipe::K] = {false, false, false};
                int j = 0;
                for (int c = first; c < n_chunks; c += step, j++) {
                    const int k = j % Pipe::K;
                    if (used[k]) CHECK(hipStreamWaitEvent(p.h2d, p.e_comp[k], 0)); // d_in free again
                    CHECK(hipMemcpyAsync(p.d_in[k], h_in + c * in_b, in_b, hipMemcpyHostToDevice, p.h2d));
                    CHECK(hipEventRecord(p.e_h2d[k], p.h2d));
                    CHECK(hipStreamWaitEvent(p.comp, p.e_h2d[k], 0));
                    if (used[k]) CHECK(hipStreamWaitEvent(p.comp, p.e_d2h[k], 0)); // d_out free again
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, p.comp, (unsigned char *)p.d_in[k],
                                       (short *)p.d_out[k], cycles);
                    CHECK(hipEventRecord(p.e_comp[k], p.comp));
                    if (host_blocks) CHECK(hipEventSynchronize(p.e_comp[k]));
                    CHECK(hipStreamWaitEvent(p.d2h, p.e_comp[k], 0));
                    CHECK(hipMemcpyAsync(h_out + c * out_b, p.d_out[k], out_b, hipMemcpyDeviceToHost, p.d2h));
                    CHECK(hipEventRecord(p.e_d2h[k], p.d2h));
                    used[k] = true;
                }
                CHECK(hipStreamSynchronize(p.d2h));
            };
            {
                double t0 = now();
                run3(pipes[0], 0, 1, false);
                report("P1 three streams + events, fully async", now() - t0);
            }
            {
                double t0 = now();
                run3(pipes[0], 0, 1, true);
                report("P2 three streams + events, host waits for each kernel", now() - t0);
            }
            // P2b: host blocks, but the NEXT upload is enqueued before the wait (prefetch distance 1), as the library does
            {
                Pipe &p = pipes[0];
                double t0 = now();
                auto upload = [&](int c) {
                    const int k = c % Pipe::K;
                    CHECK(hipMemcpyAsync(p.d_in[k], h_in + c * in_b, in_b, hipMemcpyHostToDevice, p.h2d));
                    CHECK(hipEventRecord(p.e_h2d[k], p.h2d));
                };
                upload(0);
                for (int c = 0; c < n_chunks; c++) {
                    const int k = c % Pipe::K;
                    if (c + 1 < n_chunks) upload(c + 1);
                    CHECK(hipStreamWaitEvent(p.comp, p.e_h2d[k], 0));
                    if (c >= Pipe::K) CHECK(hipStreamWaitEvent(p.comp, p.e_d2h[k], 0));
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, p.comp, (unsigned char *)p.d_in[k],
                                       (short *)p.d_out[k], cycles);
                    CHECK(hipEventRecord(p.e_comp[k], p.comp));
                    CHECK(hipEventSynchronize(p.e_comp[k]));
                    CHECK(hipStreamWaitEvent(p.d2h, p.e_comp[k], 0));
                    CHECK(hipMemcpyAsync(h_out + c * out_b, p.d_out[k], out_b, hipMemcpyDeviceToHost, p.d2h));
                    CHECK(hipEventRecord(p.e_d2h[k], p.d2h));
                }
                CHECK(hipStreamSynchronize(p.d2h));
                report("P2b as P2, next upload enqueued before the wait", now() - t0);
            }
            // P2c: as P2b but the D2H is enqueued on the COMPUTE stream's own order: no event between kernel and D2H
            //      (d2h stream = compute stream), uploads on their own stream
            {
                Pipe &p = pipes[0];
                double t0 = now();
                auto upload = [&](int c) {
                    const int k = c % Pipe::K;
                    CHECK(hipMemcpyAsync(p.d_in[k], h_in + c * in_b, in_b, hipMemcpyHostToDevice, p.h2d));
                    CHECK(hipEventRecord(p.e_h2d[k], p.h2d));
                };
                upload(0);
                for (int c = 0; c < n_chunks; c++) {
                    const int k = c % Pipe::K;
                    if (c + 1 < n_chunks) upload(c + 1);
                    CHECK(hipStreamWaitEvent(p.comp, p.e_h2d[k], 0));
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, p.comp, (unsigned char *)p.d_in[k],
                                       (short *)p.d_out[k], cycles);
                    CHECK(hipEventRecord(p.e_comp[k], p.comp));
                    CHECK(hipEventSynchronize(p.e_comp[k]));
                    CHECK(hipMemcpyAsync(h_out + c * out_b, p.d_out[k], out_b, hipMemcpyDeviceToHost, p.comp));
                }
                CHECK(hipStreamSynchronize(p.comp));
                report("P2c uploads on their own stream, kernel+D2H in one stream", now() - t0);
            }
            for (int W : {2, 3, 4}) {
                double t0 = now();
                std::vector<std::thread> th;
                for (int w = 0; w < W; w++)
                    th.emplace_back([&, w] {
                        Pipe &p = pipes[w];
                        for (int c = w; c < n_chunks; c += W) {
                            CHECK(hipMemcpyAsync(p.d_in[0], h_in + c * in_b, in_b, hipMemcpyHostToDevice, p.comp));
                            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, p.comp, (unsigned char *)p.d_in[0],
                                               (short *)p.d_out[0], cycles);
                            CHECK(hipEventRecord(p.e_comp[0], p.comp));
                            CHECK(hipEventSynchronize(p.e_comp[0]));
                            CHECK(hipMemcpyAsync(h_out + c * out_b, p.d_out[0], out_b, hipMemcpyDeviceToHost, p.comp));
                        }
                        CHECK(hipStreamSynchronize(p.comp));
                    });
                for (auto &t : th) t.join();
                char name[96];
                std::snprintf(name, sizeof name, "P3 %d threads, one in-order stream each", W);
                report(name, now() - t0);
            }
            for (int W : {2, 3}) {
                double t0 = now();
                std::vector<std::thread> th;
                for (int w = 0; w < W; w++) th.emplace_back([&, w] { run3(pipes[w], w, W, true); });
                for (auto &t : th) t.join();
                char name[96];
                std::snprintf(name, sizeof name, "P4 %d threads, three streams each, host waits per kernel", W);
                report(name, now() - t0);
            }
        }
        CHECK(hipHostFree(h_in));
        CHECK(hipHostFree(h_out));
    }
    return 0;
}
