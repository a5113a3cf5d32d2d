// host -> device copy costs on this box: hipMemcpyAsync from pageable memory (first touch / reused buffer) by size, pinned
// allocation cost, pinned copies.  hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d tools/bench_extra/h2d_micro.hip && /tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    char* d; hipMalloc(&d, 64 << 20);
    hipMemset(d, 0, 64 << 20); hipDeviceSynchronize();
    for (size_t mb : {0, 1, 2, 4, 8, 16, 32}) {
        const size_t n = mb ? mb << 20 : 256 << 10;
        char* h = (char*)malloc(n); memset(h, 1, n);
        double t0 = now(); hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); double t1 = now(); hipStreamSynchronize(s); double t2 = now();
        double best = 1e9, bestc = 1e9;
        for (int r = 0; r < 5; ++r) { double a = now(); hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); double b = now(); hipStreamSynchronize(s); double c = now(); if (c - a < best) { best = c - a; bestc = b - a; } }
        printf("pageable %5.2f MB: first call %.3f ms (+sync %.3f), reused buffer best %.3f ms (call %.3f) = %.1f GB/s\n", n / 1048576.0, t1 - t0, t2 - t0, best, bestc, n / best / 1e6);
        // fresh buffer every time (what a parser thread's first use of a buffer sees)
        double tot = 0;
        for (int r = 0; r < 4; ++r) { char* f = (char*)malloc(n); memset(f, 2, n); double a = now(); hipMemcpyAsync(d, f, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); tot += now() - a; free(f); }
        printf("          fresh buffers: %.3f ms per copy\n", tot / 4);
        free(h);
    }
    for (size_t mb : {1, 4, 16, 64}) {
        const size_t n = mb << 20;
        double a = now(); char* p; hipHostMalloc((void**)&p, n, hipHostMallocDefault); double b = now(); memset(p, 1, n); double c = now();
        hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double e = now();
        double best = 1e9; for (int r = 0; r < 5; ++r) { double x = now(); hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double y = now(); if (y - x < best) best = y - x; }
        double f0 = now(); hipHostFree(p); double f1 = now();
        printf("pinned %3zu MB: hipHostMalloc %.3f ms, first touch %.3f ms, first copy %.3f ms, best copy %.3f ms = %.1f GB/s, free %.3f ms\n", mb, b - a, c - b, e - c, best, n / best / 1e6, f1 - f0);
    }
    // hipHostRegister of existing malloc'd memory
    for (size_t mb : {4, 32}) {
        const size_t n = mb << 20; char* h = (char*)aligned_alloc(4096, n); memset(h, 1, n);
        double a = now(); hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double b = now();
        hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double c = now();
        hipHostUnregister(h); double u = now();
        printf("hipHostRegister %zu MB: %s %.3f ms, copy %.3f ms, unregister %.3f ms\n", mb, hipGetErrorString(e), b - a, c - b, u - c);
        free(h);
    }
    return 0;
}
