// Diagnostic microbenchmark: LDS-array cost of the wave-instructions a hash-count insert is made of, with random and
// with conflict-free addresses (12 waves per CU as in k_count).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_ops tools/lds_ops.hip && /tmp/lds_ops
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define SLOTS 4096
#define ITER 512

template <int OP>
__global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ addr, uint32_t *out, int random, uint64_t emask) {
    __shared__ __attribute__((aligned(16))) uint64_t tab[SLOTS + 64];
    for (int i = threadIdx.x; i < SLOTS + 64; i += 256) tab[i] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)tab;
    uint32_t a[8];
    for (int j = 0; j < 8; j++) {
        uint32_t s = random ? addr[(blockIdx.x * 8 + j) * 256 + threadIdx.x] & (SLOTS - 1) : (uint32_t)((threadIdx.x & 63) + 64 * j);
        a[j] = base + 8u * s;
    }
    uint64_t acc = 0;
    uint32_t acc32 = 0;
    long long t0 = clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a[j])); acc32 += r; }
            if (OP == 1) { uint64_t r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a[j])); acc += r; }
            if (OP == 2) { asm volatile("ds_add_u32 %0, %1" ::"v"(a[j]), "v"(1u)); }
            if (OP == 3) { uint32_t r; asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r) : "v"(a[j]), "v"(1u)); acc32 += r; }
            if (OP == 4) { uint32_t r; asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a[j]), "v"(0xFFFFFFFFu), "v"(5u)); acc32 += r; }
            if (OP == 5) { uint64_t r; asm volatile("ds_cmpst_rtn_b64 %0, %1, %2, %3" : "=v"(r) : "v"(a[j]), "v"(~0ull), "v"(5ull)); acc += r; }
            if (OP == 6) { uint64_t r; asm volatile("ds_add_rtn_u64 %0, %1, %2" : "=v"(r) : "v"(a[j]), "v"(1ull)); acc += r; }
            if (OP == 7) { asm volatile("ds_write_b64 %0, %1" ::"v"(a[j]), "v"(acc)); }
            if (OP == 8) { asm volatile("ds_write_b32 %0, %1" ::"v"(a[j]), "v"(acc32)); }
            if (OP == 9) { asm volatile("ds_add_u64 %0, %1" ::"v"(a[j]), "v"(1ull)); }
            // exec-masked forms: only the lanes of `emask` take part
            if (OP == 10) { uint64_t r = 0, sv; asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, exec, %3\n ds_read_b64 %0, %2\n s_mov_b64 exec, %1" : "+v"(r), "=&s"(sv) : "v"(a[j]), "s"(emask) : "scc"); acc += r; }
            if (OP == 11) { uint64_t sv; asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %3\n ds_add_u32 %1, %2\n s_mov_b64 exec, %0" : "=&s"(sv) : "v"(a[j]), "v"(1u), "s"(emask) : "scc"); }
            if (OP == 12) { uint64_t r = 0, sv; asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, exec, %5\n ds_cmpst_rtn_b64 %0, %2, %3, %4\n s_mov_b64 exec, %1" : "+v"(r), "=&s"(sv) : "v"(a[j]), "v"(~0ull), "v"(5ull), "s"(emask) : "scc"); acc += r; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x * 2] = (uint32_t)(t1 - t0);
    if (acc + acc32 == 0x123456789ull) out[1] = 1;
}

template <int OP>
static void run(const char *name, const uint32_t *d_addr, uint32_t *d_out, int nblk, uint64_t emask = ~0ull) {
    for (int random = 0; random < 2; random++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<nblk, 256>>>(d_addr, d_out, random, emask);
        hipEventRecord(e0);
        k<OP><<<nblk, 256>>>(d_addr, d_out, random, emask);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 12 waves x ITER x 8 wave-instructions; s_memtime-independent: use wall time at ~2.1 GHz nominal
        double winstr_per_cu = 12.0 * ITER * 8;
        printf("%-18s %s  %.3f ms  %.1f ns per wave-instr per CU\n", name, random ? "random" : "linear", ms, ms * 1e6 / winstr_per_cu);
    }
}

int main() {
    int ncu = 256, nblk = ncu * 3;
    std::vector<uint32_t> h((size_t)nblk * 8 * 256);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 20); }
    uint32_t *d_addr, *d_out;
    hipMalloc(&d_addr, h.size() * 4); hipMalloc(&d_out, nblk * 8 + 64);
    hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("ds_read_b32", d_addr, d_out, nblk);
    run<1>("ds_read_b64", d_addr, d_out, nblk);
    run<8>("ds_write_b32", d_addr, d_out, nblk);
    run<7>("ds_write_b64", d_addr, d_out, nblk);
    run<2>("ds_add_u32", d_addr, d_out, nblk);
    run<3>("ds_add_rtn_u32", d_addr, d_out, nblk);
    run<9>("ds_add_u64", d_addr, d_out, nblk);
    run<6>("ds_add_rtn_u64", d_addr, d_out, nblk);
    run<4>("ds_cmpst_rtn_b32", d_addr, d_out, nblk);
    run<5>("ds_cmpst_rtn_b64", d_addr, d_out, nblk);
    const uint64_t m8 = 0x0101010101010101ull, m16 = 0x1111111111111111ull, m1 = 1ull << 17, m32 = 0x5555555555555555ull;
    run<10>("read_b64 x32", d_addr, d_out, nblk, m32);
    run<10>("read_b64 x16", d_addr, d_out, nblk, m16);
    run<10>("read_b64 x8", d_addr, d_out, nblk, m8);
    run<10>("read_b64 x1", d_addr, d_out, nblk, m1);
    run<11>("add_u32 x32", d_addr, d_out, nblk, m32);
    run<11>("add_u32 x16", d_addr, d_out, nblk, m16);
    run<11>("add_u32 x8", d_addr, d_out, nblk, m8);
    run<11>("add_u32 x1", d_addr, d_out, nblk, m1);
    run<12>("cmpst_b64 x32", d_addr, d_out, nblk, m32);
    run<12>("cmpst_b64 x16", d_addr, d_out, nblk, m16);
    run<12>("cmpst_b64 x8", d_addr, d_out, nblk, m8);
    run<12>("cmpst_b64 x1", d_addr, d_out, nblk, m1);
    return 0;
}
