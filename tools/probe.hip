// Diagnostic probe (not part of the library): launch overhead, shader clock under light load,
// dependent-load latency.  Build: hipcc -O3 --offload-arch=gfx950 tools/probe.hip -o tools/probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}

__global__ void k_clock(unsigned long long* out, int spin) {
    unsigned long long c0 = clock64(), r0 = wall_clock64();
    while (clock64() - c0 < (unsigned long long)spin) {}
    unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

__global__ void k_chase(const int* next, int steps, int* out, unsigned long long* t) {
    int i = threadIdx.x;
    unsigned long long r0 = wall_clock64(), c0 = clock64();
    for (int s = 0; s < steps; ++s) i = next[i];
    unsigned long long r1 = wall_clock64(), c1 = clock64();
    out[threadIdx.x] = i;
    if (threadIdx.x == 0) { t[0] = r1 - r0; t[1] = c1 - c0; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned long long* d; CK(hipMalloc(&d, 64));
    unsigned long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) k_empty<<<16, 256, 0, s>>>();
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::high_resolution_clock::now();
        const int N = 2000;
        for (int i = 0; i < N; ++i) k_empty<<<16, 256, 0, s>>>();
        auto t1 = std::chrono::high_resolution_clock::now();
        CK(hipStreamSynchronize(s));
        auto t2 = std::chrono::high_resolution_clock::now();
        printf("empty kernel x%d: host issue %.2f us/launch, total %.2f us/launch\n", N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    // graph of 128 empty kernels
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 128; ++i) k_empty<<<16, 256, 0, s>>>();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t2 = std::chrono::high_resolution_clock::now();
        printf("graph of 128 empty kernels: %.2f us/kernel\n", std::chrono::duration<double, std::micro>(t2 - t0).count() / 50 / 128);
    }
    for (int spin : {2000, 20000, 200000, 2000000}) {
        k_clock<<<1, 64, 0, s>>>(d, spin);
        CK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        printf("spin %d cycles: memtime %llu, realtime %llu ticks (100 MHz) -> %.0f MHz\n", spin, h[0], h[1], (double)h[0] / h[1] * 100.0);
    }
    // pointer chase over 1 MiB (L2) and 64 MiB (MALL), stride 4 KiB-ish random
    for (size_t n : {(size_t)1 << 18, (size_t)1 << 24}) {
        std::vector<int> nx(n);
        for (size_t i = 0; i < n; ++i) nx[i] = (int)((i * 7919 + 12345) % n);
        int* dn; int* dout; CK(hipMalloc(&dn, n * 4)); CK(hipMalloc(&dout, 256));
        CK(hipMemcpy(dn, nx.data(), n * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            k_chase<<<1, 1, 0, s>>>(dn, 2000, dout, d);
            CK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            printf("chase %zu KiB: %.1f ns/load, %.0f cycles/load\n", n * 4 / 1024, h[0] * 10.0 / 2000, (double)h[1] / 2000);
        }
        CK(hipFree(dn)); CK(hipFree(dout));
    }
    return 0;
}
