// Random-read rates of the memory system, measured: N independent 4-byte reads at hashed addresses of an array of S bytes,
// one per thread at full occupancy (a), and chains of two dependent reads (b).  What an index lookup per request can cost at best.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_rate tools/gather_rate.hip && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int CHAIN, int PER>
__global__ void k_gather(const uint32_t *__restrict__ a, uint64_t nwords, uint64_t n, uint32_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const uint64_t q = t * PER + r;
        if (q >= n) break;
        uint64_t i = mix(q) % nwords;
        uint32_t v = a[i];
        if (CHAIN >= 2) { i = mix(q ^ v ^ 0x1234567ull) % nwords; v += a[i]; }
        if (CHAIN >= 3) { i = mix(q ^ v ^ 0x7654321ull) % nwords; v += a[i]; }
        acc += v;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const uint64_t n = 300000000ull;
    uint32_t *out; hipMalloc(&out, 4);
    for (uint64_t mb : {16ull, 64ull, 256ull, 1024ull, 2900ull, 8000ull}) {
        const uint64_t nwords = mb * 1000000ull / 4;
        uint32_t *a; hipMalloc(&a, nwords * 4); hipMemset(a, 0, nwords * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](auto kern, int per, const char *name) {
            const uint64_t threads = (n + per - 1) / per;
            kern<<<(unsigned)((threads + 255) / 256), 256>>>(a, nwords, n, out);
            hipEventRecord(e0);
            kern<<<(unsigned)((threads + 255) / 256), 256>>>(a, nwords, n, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %-22s %7.2f ms  %6.1f G requests/s\n", name, ms, n / ms / 1e6);
        };
        printf("array %llu MB, %llu requests\n", (unsigned long long)mb, (unsigned long long)n);
        run(k_gather<1, 1>, 1, "1 read, 1 per thread");
        run(k_gather<1, 4>, 4, "1 read, 4 per thread");
        run(k_gather<2, 1>, 1, "chain of 2, 1/thread");
        run(k_gather<2, 4>, 4, "chain of 2, 4/thread");
        run(k_gather<3, 4>, 4, "chain of 3, 4/thread");
        hipFree(a);
    }
    return 0;
}
