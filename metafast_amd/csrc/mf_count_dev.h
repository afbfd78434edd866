// mf_count_dev.h -- device helpers shared by the two counting paths (mf_count.hip: one record per k-mer;
// mf_skm.hip: super-k-mer records).
#ifndef MF_COUNT_DEV_H
#define MF_COUNT_DEV_H
#include "mf_common.h"
#ifdef __HIPCC__
#define MF_WG __HIP_MEMORY_SCOPE_WORKGROUP

// 4 ASCII bases (byte 0 = first base) -> 8 bits, first base most significant, code A0 G1 C2 T3.
// (c>>1)&3 maps A,C,T,G (either case) to 0,1,2,3; the reference order needs f(x)=((x0^x1)<<1)|x1.
__device__ __forceinline__ uint32_t mf_dec4(uint32_t w) {
    uint32_t t = (w >> 1) & 0x03030303u;
    uint32_t x1 = (t >> 1) & 0x01010101u;
    uint32_t x0 = t & 0x01010101u;
    uint32_t c = ((x0 ^ x1) << 1) | x1;
    return (c * 0x40100401u) >> 24;   // gathers the four 2-bit fields into one byte
}
__device__ __forceinline__ uint32_t mf_dec16(uint4 v) {
    return (mf_dec4(v.x) << 24) | (mf_dec4(v.y) << 16) | (mf_dec4(v.z) << 8) | mf_dec4(v.w);
}


// LDS byte address of a pointer into __shared__ memory (operand of the ds_* instructions below)
__device__ __forceinline__ uint32_t mf_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// Four LDS operations issued back to back, ONE s_waitcnt for all of them.  These are inline asm because hipcc turns the
// equivalent C++ (atomics that add 0 on dummy slots) back into per-element exec branches with a wait after each atomic.
// LDS instructions of one wave execute in order, so a write issued before the commit atomic is visible to whoever
// observes that commit; the "memory" clobbers keep the compiler from moving other accesses across.
__device__ __forceinline__ void mf_lds_read4(const uint32_t (&a)[4], uint32_t (&v)[4]) {
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
                 : "memory");
}
__device__ __forceinline__ void mf_lds_add_rtn4(const uint32_t (&a)[4], const uint32_t (&inc)[4], uint32_t (&old)[4]) {
    asm volatile("ds_add_rtn_u32 %0, %4, %8\n\tds_add_rtn_u32 %1, %5, %9\n\tds_add_rtn_u32 %2, %6, %10\n\t"
                 "ds_add_rtn_u32 %3, %7, %11\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3])
                 : "memory");
}

// ---- four-wide LDS steps of the counting table (inline asm for the same reason as in the staging protocol) ----
__device__ __forceinline__ void mf_lds_read4_b64(const uint32_t (&a)[4], uint64_t (&v)[4]) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
                 : "memory");
}
// ds_cmpst_rtn_b64 vdst, vaddr, vcmp, vnew : MEM = (MEM == cmp) ? new : MEM, returns the old value
__device__ __forceinline__ void mf_lds_cmpst4_b64(const uint32_t (&a)[4], uint64_t cmp, const uint64_t (&nv)[4], uint64_t (&old)[4]) {
    asm volatile("ds_cmpst_rtn_b64 %0, %4, %8, %9\n\tds_cmpst_rtn_b64 %1, %5, %8, %10\n\tds_cmpst_rtn_b64 %2, %6, %8, %11\n\t"
                 "ds_cmpst_rtn_b64 %3, %7, %8, %12\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(cmp), "v"(nv[0]), "v"(nv[1]), "v"(nv[2]), "v"(nv[3])
                 : "memory");
}
__device__ __forceinline__ void mf_lds_add4(const uint32_t (&a)[4], const uint32_t (&inc)[4]) {
    asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %5\n\tds_add_u32 %2, %6\n\tds_add_u32 %3, %7"
                 :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3])
                 : "memory");
}
// Insert four keys per lane: probe (4 reads in flight) -> claim empty slots (4 CAS in flight) -> count hits (4 adds, not
// waited for).  Lanes with nothing to do in a step use a private dummy slot / counter.  The loop is wave-uniform; a
// miss moves that key to the next slot (linear probing).  A `volatile` C++ probe here compiled to flat_load sc0 sc1 +
// a full vmcnt/lgkmcnt wait, and the CAS path ran serially with a few active lanes: 45 % of the kernel.
__device__ __forceinline__ void mf_count_insert4(uint32_t tk0, uint32_t tc0, uint32_t dummy_k, uint32_t dummy_c, uint32_t mask,
                                                 uint32_t slots, const uint64_t (&key)[4], unsigned int *overflow) {
    uint32_t s[4]; bool pend[4];
#pragma unroll
    for (int b = 0; b < 4; b++) { pend[b] = key[b] != MF_EMPTY; s[b] = mf_pslot(mf_phash(key[b])) & mask; }
    for (uint32_t probes = 0;; probes++) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < 4; b++) any |= pend[b];
        if (__ballot(any) == 0ull) break;
        if (probes > slots) { if (mf_lane() == 0) atomicExch(overflow, 1u); break; }
        uint32_t ka[4], ca[4], aa[4], inc[4]; uint64_t cur[4], ret[4]; bool need[4]; bool anyneed = false;
#pragma unroll
        for (int b = 0; b < 4; b++) ka[b] = pend[b] ? tk0 + 8u * s[b] : dummy_k;
        mf_lds_read4_b64(ka, cur);
#pragma unroll
        for (int b = 0; b < 4; b++) { need[b] = pend[b] && cur[b] == MF_EMPTY; ca[b] = need[b] ? ka[b] : dummy_k; anyneed |= need[b]; }
        if (__ballot(anyneed) != 0ull) {
            mf_lds_cmpst4_b64(ca, MF_EMPTY, key, ret);
#pragma unroll
            for (int b = 0; b < 4; b++) if (need[b]) cur[b] = ret[b] == MF_EMPTY ? key[b] : ret[b];
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const bool hit = pend[b] && cur[b] == key[b];
            aa[b] = hit ? tc0 + 4u * s[b] : dummy_c;
            inc[b] = hit ? 1u : 0u;
            if (hit) pend[b] = false;
            else s[b] = (s[b] + 1) & mask;
        }
        mf_lds_add4(aa, inc);
    }
}

// dense output: partition p's d distinct entries go to [doff[p], doff[p]+d)
static __global__ __launch_bounds__(256) void k_gather(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnt,
                                                const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ dcount,
                                                const uint64_t *__restrict__ doff, uint32_t np,
                                                uint64_t *__restrict__ dk, uint16_t *__restrict__ dc,
                                                uint32_t p0 = 0, uint64_t sbase = 0) {
    // partitions [p0, np); sbase: index of keys[0] / cnt[0] in pstart's numbering (a buffer that holds one batch of slices)
    for (uint32_t p = p0 + blockIdx.x; p < np; p += gridDim.x) {
        uint64_t s = pstart[p] - sbase, o = doff[p];
        uint32_t d = dcount[p];
        for (uint32_t j = threadIdx.x; j < d; j += blockDim.x) { dk[o + j] = keys[s + j]; dc[o + j] = cnt[s + j]; }
    }
}


#endif  // __HIPCC__
#endif
