// probe_small_d2h.hip -- what a small device -> host read-back costs on this stack: hipMemcpyAsync of 8 / 2048 bytes +
// hipStreamSynchronize into pageable and into pinned host memory, right after a small kernel (the pattern of every
// "how many did the kernel produce" round trip in the library).   hipcc -O3 --offload-arch=gfx950 tools/probe_small_d2h.hip -o tools/probe_small_d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void bump(unsigned long long* p) { if (threadIdx.x == 0) p[0] += 1; }
static double ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    unsigned long long* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* pinned; hipHostMalloc((void**)&pinned, 4096, hipHostMallocDefault);
    unsigned long long* pageable = (unsigned long long*)malloc(4096);
    const int iters = 2000;
    for (size_t bytes : {8ul, 2048ul})
        for (int kind = 0; kind < 2; ++kind) {
            unsigned long long* h = kind ? pinned : pageable;
            for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, d); hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            const double t0 = ms();
            for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, d); hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            printf("kernel + %4zu-byte D2H + sync into %-8s host memory: %.1f us per round trip\n", bytes, kind ? "pinned" : "pageable", (ms() - t0) * 1000 / iters);
        }
    // the same with a memset in front (reset of a cursor) as the library does
    {
        const double t0 = ms();
        for (int i = 0; i < iters; ++i) { hipMemsetAsync(d, 0, 64, s); hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, d); hipMemcpyAsync(pinned, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
        printf("memset + kernel + 8-byte D2H + sync (pinned): %.1f us per round trip\n", (ms() - t0) * 1000 / iters);
    }
    return 0;
}
