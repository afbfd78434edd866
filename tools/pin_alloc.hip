// How long does pinned host memory take?  hipcc -O2 tools/pin_alloc.hip -o tools/bin/pin_alloc -lpthread && tools/bin/pin_alloc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now();
    hipSetDevice(0); hipFree(0);
    printf("runtime start %.3f s\n", now() - t0);
    const size_t CH = (size_t)17 << 20; const int N = 64;
    for (unsigned flags : {0u, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocWriteCombined, (unsigned)hipHostMallocNumaUser}) {
        void *p = nullptr; t0 = now();
        hipError_t e = hipHostMalloc(&p, CH * N, flags);
        double t1 = now();
        if (e == hipSuccess) { hipHostFree(p); }
        printf("one block of %.2f GB flags %#x: alloc %.3f s free %.3f s (%s)\n", CH * N / 1e9, flags, t1 - t0, now() - t1, hipGetErrorString(e));
    }
    {
        std::vector<void *> ps(N, nullptr); t0 = now();
        for (int i = 0; i < N; i++) hipHostMalloc(&ps[i], CH, 0);
        double t1 = now();
        for (int i = 0; i < N; i++) hipHostFree(ps[i]);
        printf("%d blocks one after the other: alloc %.3f s free %.3f s\n", N, t1 - t0, now() - t1);
    }
    {
        std::vector<void *> ps(N, nullptr); std::vector<std::thread> th; t0 = now();
        for (int i = 0; i < N; i++) th.emplace_back([&, i]() { hipSetDevice(0); hipHostMalloc(&ps[i], CH, 0); });
        for (auto &x : th) x.join();
        double t1 = now();
        for (int i = 0; i < N; i++) hipHostFree(ps[i]);
        printf("%d blocks from %d threads: alloc %.3f s free %.3f s\n", N, N, t1 - t0, now() - t1);
    }
    {
        char *m = (char *)aligned_alloc(4096, CH * N); t0 = now();
        std::vector<std::thread> th;
        for (int i = 0; i < N; i++) th.emplace_back([&, i]() { memset(m + (size_t)i * CH, 1, CH); });
        for (auto &x : th) x.join();
        double t1 = now();
        hipError_t e = hipHostRegister(m, CH * N, hipHostRegisterDefault);
        double t2 = now();
        if (e == hipSuccess) hipHostUnregister(m);
        printf("malloc + parallel first touch %.3f s, hipHostRegister %.3f s, unregister %.3f s (%s)\n", t1 - t0, t2 - t1, now() - t2, hipGetErrorString(e));
        free(m);
    }
    {   // H2D from pageable memory, for comparison
        const size_t B = (size_t)1 << 30; char *m = (char *)malloc(B); memset(m, 1, B); void *d = nullptr; hipMalloc(&d, B);
        hipMemcpy(d, m, B, hipMemcpyHostToDevice); t0 = now(); hipMemcpy(d, m, B, hipMemcpyHostToDevice); hipDeviceSynchronize();
        printf("H2D of 1 GiB pageable: %.3f s\n", now() - t0);
        void *p = nullptr; hipHostMalloc(&p, B, 0); memset(p, 1, B); hipMemcpy(d, p, B, hipMemcpyHostToDevice); t0 = now(); hipMemcpy(d, p, B, hipMemcpyHostToDevice); hipDeviceSynchronize();
        printf("H2D of 1 GiB pinned: %.3f s\n", now() - t0);
    }
    return 0;
}
