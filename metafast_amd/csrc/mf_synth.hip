// mf_synth.hip -- deterministic synthetic metagenome reads; integer arithmetic only,
// so the device kernel and the host mirror produce identical bytes.
//
//  pool     128 genomes, length = (scale/2 + h % (scale/2)) << (0..3)  (log-uniform-ish over [0.5,8) x scale);
//           2 % of every genome is a copy of a stretch of another pool member (shared repeats)
//  sample s uses genomes (32 s + j) mod 128, j = 0..63 (neighbouring samples share 32 genomes) with
//           log-normal-like abundances: exponent = Irwin-Hall(4 x 16 bit) * 2.164 / sigma, 2^x piecewise linear
//  read     genome by abundance x length, uniform start, strand flip p = 1/2, substitutions p = 82/16384 (0.5 %;
//           the _ex entry points take the numerator: 164 = 1 %, BASELINE config 5)
//  alphabet reference coding A0 G1 C2 T3, complement = 3 - c; no N
#include "mf_common.h"

#define MF_SYNTH_NG 128
#define MF_SYNTH_NS 64

struct mf_synth_genome { uint64_t len, rs, rl, ro; uint32_t src; uint32_t pad; };
struct mf_synth_tables {
    mf_synth_genome g[MF_SYNTH_NG];
    uint64_t cum[MF_SYNTH_NS + 1];
    uint32_t member[MF_SYNTH_NS];
    uint64_t seed_pool, seed_reads;
    uint32_t sub_thr, pad_;                 // substitution when a 14-bit draw < sub_thr
};

__host__ __device__ __forceinline__ uint64_t mf_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t mf_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
__host__ __device__ __forceinline__ uint32_t mf_synth_raw(const mf_synth_tables &T, uint32_t g, uint64_t i) {
    uint64_t h = mf_splitmix64(T.seed_pool + ((uint64_t)g << 40) + (i >> 5));
    return (uint32_t)(h >> (2 * (i & 31))) & 3u;
}
__host__ __device__ __forceinline__ uint32_t mf_synth_base(const mf_synth_tables &T, uint32_t g, uint64_t i) {
    const mf_synth_genome &G = T.g[g];
    if (i >= G.rs && i < G.rs + G.rl) return mf_synth_raw(T, G.src, G.ro + (i - G.rs));
    return mf_synth_raw(T, g, i);
}
// writes read r (index within the sample) to out[0..L)
__host__ __device__ __forceinline__ void mf_synth_read(const mf_synth_tables &T, uint64_t r, int L, uint8_t *out) {
    const uint64_t s0 = mf_splitmix64(T.seed_reads ^ (r * 0x9E3779B97F4A7C15ULL));
    const uint64_t h0 = mf_splitmix64(s0 + 1), h1 = mf_splitmix64(s0 + 2), h2 = mf_splitmix64(s0 + 3);
    const uint64_t u = mf_mulhi64(h0, T.cum[MF_SYNTH_NS]);
    int lo = 0, hi = MF_SYNTH_NS;               // cum[lo] <= u < cum[hi]
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (T.cum[mid] <= u) lo = mid; else hi = mid; }
    const uint32_t g = T.member[lo];
    const uint64_t glen = T.g[g].len;
    const uint64_t start = mf_mulhi64(h1, glen - (uint64_t)L + 1);
    const bool flip = h2 & 1;
    uint64_t e = 0;
    for (int i = 0; i < L; i++) {
        if ((i & 3) == 0) e = mf_splitmix64(s0 + 16 + (uint64_t)(i >> 2));
        uint32_t x = (uint32_t)(e >> (16 * (i & 3))) & 0xFFFFu;
        uint32_t c = flip ? 3u - mf_synth_base(T, g, start + (uint64_t)(L - 1 - i)) : mf_synth_base(T, g, start + (uint64_t)i);
        if ((x >> 2) < T.sub_thr) c = (c + 1u + (x & 3u) % 3u) & 3u;
        out[i] = (uint8_t)("AGCT"[c]);
    }
}

static int mf_synth_make_tables(uint64_t seed, int sample, int read_len, uint64_t scale, int sub_per_16384, mf_synth_tables *T) {
    if (sub_per_16384 < 0 || sub_per_16384 > 16384) return mf_set_error("mf_synth: substitutions per 16384 bases out of [0,16384]");
    T->sub_thr = (uint32_t)sub_per_16384; T->pad_ = 0;
    if (scale < 64 || (scale / 2) < (uint64_t)read_len) return mf_set_error("mf_synth: genome_scale_bp too small for read_len");
    if (scale > (1ull << 24)) return mf_set_error("mf_synth: genome_scale_bp too large");
    T->seed_pool = mf_splitmix64(seed ^ 0x504F4F4CULL);                                  // "POOL"
    T->seed_reads = mf_splitmix64(seed ^ 0x5245414453ULL ^ ((uint64_t)(uint32_t)sample << 48));   // "READS"
    for (uint32_t g = 0; g < MF_SYNTH_NG; g++) {
        uint64_t h = mf_splitmix64(seed + 0x1000 + g);
        uint64_t base = scale / 2 + h % (scale / 2);
        T->g[g].len = base << ((h >> 40) & 3);
    }
    for (uint32_t g = 0; g < MF_SYNTH_NG; g++) {
        uint64_t h2 = mf_splitmix64(seed + 0x2000 + g), h3 = mf_splitmix64(seed + 0x3000 + g), h4 = mf_splitmix64(seed + 0x4000 + g);
        mf_synth_genome &G = T->g[g];
        G.src = (g + 1 + (uint32_t)(h2 % (MF_SYNTH_NG - 1))) % MF_SYNTH_NG;
        G.rl = G.len / 50;
        if (G.rl > T->g[G.src].len) G.rl = T->g[G.src].len;
        G.rs = h3 % (G.len - G.rl + 1);
        G.ro = h4 % (T->g[G.src].len - G.rl + 1);
        G.pad = 0;
    }
    uint64_t acc = 0;
    for (int j = 0; j < MF_SYNTH_NS; j++) {
        uint32_t g = (uint32_t)((32 * (uint32_t)sample + (uint32_t)j) % MF_SYNTH_NG);
        T->member[j] = g;
        uint64_t h = mf_splitmix64(seed + 0x5000 + ((uint64_t)(uint32_t)sample << 20) + (uint64_t)j);
        int64_t z = (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) + (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48);
        int64_t e = (z - 131070) * 15353 / 32768;          // x * (1.5 / ln 2) / sigma_IH in Q13
        int64_t t = 8 * 8192 + e;
        uint64_t w = (uint64_t)(8192 + (t & 8191)) << (t >> 13);
        T->cum[j] = acc;
        acc += w * T->g[g].len;
    }
    T->cum[MF_SYNTH_NS] = acc;
    return MF_OK;
}

__global__ void k_synth_reads(const mf_synth_tables *__restrict__ T, uint64_t first_read, uint64_t n_reads, int L,
                              uint8_t *__restrict__ bases, uint64_t *__restrict__ offsets) {
    __shared__ mf_synth_tables S;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(T);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&S);
        for (unsigned i = threadIdx.x; i < sizeof(mf_synth_tables) / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) offsets[n_reads] = n_reads * (uint64_t)L;
    if (r >= n_reads) return;
    offsets[r] = r * (uint64_t)L;
    mf_synth_read(S, first_read + r, L, bases + r * (uint64_t)L);
}

extern "C" int mf_synth_reads_device_ex(mf_ctx *ctx, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                                        int read_len, uint64_t genome_scale_bp, int sub_per_16384, void *d_bases, void *d_offsets) {
    if (!ctx || !d_bases || !d_offsets) return mf_set_error("mf_synth_reads_device: NULL argument");
    if (read_len < 1 || read_len > 100000) return mf_set_error("mf_synth: bad read_len");
    MF_HIP(hipSetDevice(ctx->device));
    mf_synth_tables T;
    MF_TRY(mf_synth_make_tables(seed, sample, read_len, genome_scale_bp, sub_per_16384, &T));
    mf_buf<mf_synth_tables> dT; MF_TRY(dT.alloc(ctx, 1));
    MF_HIP(hipMemcpyAsync(dT.p, &T, sizeof T, hipMemcpyHostToDevice, ctx->stream));
    if (n_reads) {
        k_synth_reads<<<(unsigned)((n_reads + 255) / 256), 256, 0, ctx->stream>>>(dT.p, first_read, n_reads, read_len,
                                                                                   (uint8_t *)d_bases, (uint64_t *)d_offsets);
    } else {
        uint64_t z = 0;
        MF_HIP(hipMemcpyAsync(d_offsets, &z, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    MF_HIP(hipGetLastError());
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

extern "C" int mf_synth_reads_device(mf_ctx *ctx, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                                     int read_len, uint64_t genome_scale_bp, void *d_bases, void *d_offsets) {
    return mf_synth_reads_device_ex(ctx, seed, sample, first_read, n_reads, read_len, genome_scale_bp, 82, d_bases, d_offsets);
}

extern "C" int mf_synth_reads_host_ex(uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads, int read_len,
                                      uint64_t genome_scale_bp, int sub_per_16384, uint8_t *bases, uint64_t *offsets) {
    if (!bases || !offsets) return mf_set_error("mf_synth_reads_host: NULL argument");
    if (read_len < 1 || read_len > 100000) return mf_set_error("mf_synth: bad read_len");
    mf_synth_tables T;
    MF_TRY(mf_synth_make_tables(seed, sample, read_len, genome_scale_bp, sub_per_16384, &T));
    for (uint64_t r = 0; r < n_reads; r++) {
        offsets[r] = r * (uint64_t)read_len;
        mf_synth_read(T, first_read + r, read_len, bases + r * (uint64_t)read_len);
    }
    offsets[n_reads] = n_reads * (uint64_t)read_len;
    return MF_OK;
}

extern "C" int mf_synth_reads_host(uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads, int read_len,
                                   uint64_t genome_scale_bp, uint8_t *bases, uint64_t *offsets) {
    return mf_synth_reads_host_ex(seed, sample, first_read, n_reads, read_len, genome_scale_bp, 82, bases, offsets);
}
