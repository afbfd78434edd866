// Diagnostic: how well do minimizer partitions balance?  Counts the bench sample, then histograms the DISTINCT k-mers
// (and their occurrences) per partition id = mix(min over the k-m+1 canonical m-mers of fmix32(m-mer)) >> (32-B).
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -o tools/bin/skm_stats tools/skm_stats.hip -Lmetafast_amd/lib -lmetafast_hip
//   LD_LIBRARY_PATH=metafast_amd/lib tools/bin/skm_stats [reads] [m]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "metafast_hip.h"

__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__device__ __forceinline__ uint32_t remix32(uint32_t h) { h *= 0x9E3779B1u; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; return h; }
__device__ __forceinline__ uint32_t minimizer_ph(uint64_t x, int k, int m) {
    // forward m-mers of x; rc m-mer via 2-bit reversal of the complement
    const uint32_t mm = (m == 16) ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    uint32_t best = 0xFFFFFFFFu;
    for (int j = 0; j + m <= k; j++) {
        uint32_t f = (uint32_t)(x >> (2 * (k - m - j))) & mm;
        uint32_t r = __brev(~f);                  // reverse bits, then swap within pairs
        r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
        r >>= (32 - 2 * m);
        uint32_t c = f < r ? f : r;
        uint32_t h = fmix32(c);
        best = h < best ? h : best;
    }
    return remix32(best);
}
__global__ void k_hist(const uint64_t *keys, const uint16_t *cnt, uint64_t n, int k, int m, int B, uint32_t *hd, unsigned long long *ho) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = minimizer_ph(keys[i], k, m) >> (32 - B);
    atomicAdd(&hd[p], 1u);
    atomicAdd(&ho[p], (unsigned long long)cnt[i]);
}
int main(int argc, char **argv) {
    uint64_t n_reads = argc > 1 ? strtoull(argv[1], 0, 10) : 20000000ull;
    int m = argc > 2 ? atoi(argv[2]) : 15;
    uint64_t gscale = argc > 3 ? strtoull(argv[3], 0, 10) : 1000000ull;
    int rl = 150, k = 31;
    mf_ctx *ctx; if (mf_ctx_create(0, 0, &ctx)) { printf("ctx: %s\n", mf_last_error()); return 1; }
    mf_ctx_set_stream(ctx, nullptr);
    uint8_t *bases; uint64_t *offs;
    hipMalloc(&bases, n_reads * rl + 64); hipMalloc(&offs, (n_reads + 1) * 8);
    hipMemset(bases, 0, n_reads * rl + 64);
    if (mf_synth_reads_device(ctx, 0x4D45544146415354ull, 0, 0, n_reads, rl, gscale, bases, offs)) { printf("synth: %s\n", mf_last_error()); return 1; }
    mf_table *t;
    if (mf_count_device(ctx, bases, offs, n_reads, n_reads * rl, k, 0, &t)) { printf("count: %s\n", mf_last_error()); return 1; }
    const void *dk, *dc; uint64_t n;
    mf_table_device_view(t, &dk, &dc, &n);
    uint64_t nocc = n_reads * (rl - k + 1);
    int B = 0; while ((3072ull << B) < nocc) B++;
    printf("reads %llu distinct %llu occ %llu B %d m %d\n", (unsigned long long)n_reads, (unsigned long long)n, (unsigned long long)nocc, B, m);
    uint32_t *hd; unsigned long long *ho; size_t np = (size_t)1 << B;
    hipMalloc(&hd, np * 4); hipMalloc(&ho, np * 8); hipMemset(hd, 0, np * 4); hipMemset(ho, 0, np * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_hist<<<(unsigned)((n + 255) / 256), 256>>>((const uint64_t *)dk, (const uint16_t *)dc, n, k, m, B, hd, ho);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> d(np); std::vector<unsigned long long> o(np);
    hipMemcpy(d.data(), hd, np * 4, hipMemcpyDeviceToHost); hipMemcpy(o.data(), ho, np * 8, hipMemcpyDeviceToHost);
    std::sort(d.begin(), d.end()); std::sort(o.begin(), o.end());
    auto q = [&](auto &v, double f) { return (unsigned long long)v[(size_t)(f * (np - 1))]; };
    printf("hist kernel %.2f ms (%.1f ns/key incl. atomics)\n", ms, ms * 1e6 / n);
    printf("distinct/partition: mean %.1f  p50 %llu p99 %llu p99.9 %llu p99.99 %llu max %llu   >3600: %zu  >2048: %zu\n", (double)n / np, q(d, .5), q(d, .99), q(d, .999), q(d, .9999),
           (unsigned long long)d.back(), (size_t)(d.end() - std::upper_bound(d.begin(), d.end(), 3600u)), (size_t)(d.end() - std::upper_bound(d.begin(), d.end(), 2048u)));
    printf("occ/partition:      mean %.1f  p50 %llu p99 %llu p99.9 %llu p99.99 %llu max %llu\n", (double)nocc / np, q(o, .5), q(o, .99), q(o, .999), q(o, .9999), o.back());
    return 0;
}
