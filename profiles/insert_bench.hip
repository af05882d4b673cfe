// Micro-benchmark behind round 4's work on k_insert_multi ("Next round" item 3): what do the requests of the
// table insertion cost on this part, and which arrangement of them is worth building?
//   hipcc --offload-arch=gfx950 -O3 -o profiles/insert_bench profiles/insert_bench.hip && profiles/insert_bench
// The insertion of a round at configs[2]: 9.69 M samples (31 targets x 5 Mbp / 16), each reads its bucket of the 2 GiB table
// (8-byte entries, 2^28 buckets) and about a third go on to a 64-bit atomicMax. Variants (requests as in the product: random
// buckets, a third of the samples reach the atomic, decided by the value read):
//   read_only          every sample reads its bucket, nothing else
//   atomic_all         every sample issues the atomic without a read              (round 2's A/B: 0.60 against 0.49 ms)
//   atomic_third       a third of the samples issue the atomic, nobody reads      (the atomics' own rate)
//   fused              read, then atomicMax for a third                           (k_insert_multi as built)
//   fused_x2 / _x4     two / four samples per thread, all reads in flight, then the atomics
//   fused_ret_log      the atomic returns the displaced entry, which is appended to a log (what an insertion that can be
//                      undone needs: wave-aggregated append of {bucket, old})
//   split              pass 1 reads and lists the survivors (wave-aggregated append of {bucket, key}), pass 2 issues the atomics
//   fused_sliced       as fused, but thread t only meets the 32 MB slice t * 64 / N of the table (the access pattern after a
//                      multisplit of the samples by table slice; the multisplit itself not included)
//   fused_src          as fused with the K-mer hashed from a 155 MB text (7 dwords per sample at stride 16), bucket = that hash
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool third(u64 old, uint32_t t) { return ((uint32_t) old + mix(t ^ 0x9e3779b9u)) % 3u == 0u; }
__device__ __forceinline__ u64 key_of(uint32_t t, uint32_t epoch) { return ((u64) epoch << 42) | ((u64) t << 10) | (mix(t) & 1023u); }

__global__ void __launch_bounds__(256) k_read_only(const u64 *__restrict__ ht, uint32_t mask, uint32_t n, uint32_t seed, u64 *sink) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const u64 v = ht[mix(t + seed) & mask];
    if (v == 0x123456789ull) *sink = v;
}
__global__ void __launch_bounds__(256) k_atomic(u64 *__restrict__ ht, uint32_t mask, uint32_t n, uint32_t seed, uint32_t epoch, int everyThird) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    if (everyThird && !third(0, t)) return;
    atomicMax(&ht[mix(t + seed) & mask], key_of(t, epoch));
}
template <int PER>
__global__ void __launch_bounds__(256) k_fused(u64 *__restrict__ ht, uint32_t mask, uint32_t n, uint32_t seed, uint32_t epoch) {
    const uint32_t t0 = (blockIdx.x * 256u + threadIdx.x) * PER;
    uint32_t b[PER]; u64 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { b[k] = mix(t0 + k + seed) & mask; v[k] = t0 + k < n ? ht[b[k]] : ~0ull; }
#pragma unroll
    for (int k = 0; k < PER; k++) if (t0 + k < n && third(v[k], t0 + k) && v[k] < key_of(t0 + k, epoch)) atomicMax(&ht[b[k]], key_of(t0 + k, epoch));
}
__global__ void __launch_bounds__(256) k_fused_sliced(u64 *__restrict__ ht, uint32_t sliceBits, uint32_t n, uint32_t seed, uint32_t epoch) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t slice = (uint32_t) ((u64) t * 64ull / n);
    const uint32_t b = (slice << sliceBits) | (mix(t + seed) & ((1u << sliceBits) - 1u));
    const u64 v = ht[b];
    if (third(v, t) && v < key_of(t, epoch)) atomicMax(&ht[b], key_of(t, epoch));
}
__global__ void __launch_bounds__(256) k_fused_ret_log(u64 *__restrict__ ht, uint32_t mask, uint32_t n, uint32_t seed, uint32_t epoch,
                                                        u64 *__restrict__ log, unsigned int *__restrict__ logN) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    bool app = false;
    uint32_t b = 0; u64 old = 0;
    if (t < n) {
        b = mix(t + seed) & mask;
        const u64 v = ht[b], key = key_of(t, epoch);
        if (third(v, t) && v < key) { old = atomicMax(&ht[b], key); app = old < key; }
    }
    const u64 m = __ballot(app);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(logN, (unsigned int) __popcll(m));
        base = (uint32_t) __builtin_amdgcn_readlane((int) base, leader);
        if (app) { const uint32_t at = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull)); log[2 * (u64) at] = b; log[2 * (u64) at + 1] = old; }
    }
}
__global__ void __launch_bounds__(256) k_split_read(const u64 *__restrict__ ht, uint32_t mask, uint32_t n, uint32_t seed, uint32_t epoch,
                                                     u64 *__restrict__ list, unsigned int *__restrict__ listN) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    bool app = false;
    uint32_t b = 0; u64 key = 0;
    if (t < n) {
        b = mix(t + seed) & mask;
        const u64 v = ht[b];
        key = key_of(t, epoch);
        app = third(v, t) && v < key;
    }
    const u64 m = __ballot(app);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(listN, (unsigned int) __popcll(m));
        base = (uint32_t) __builtin_amdgcn_readlane((int) base, leader);
        if (app) { const uint32_t at = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull)); list[2 * (u64) at] = b; list[2 * (u64) at + 1] = key; }
    }
}
__global__ void __launch_bounds__(256) k_split_atomic(u64 *__restrict__ ht, const u64 *__restrict__ list, const unsigned int *__restrict__ listN) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= *listN) return;
    atomicMax(&ht[list[2 * (u64) t]], list[2 * (u64) t + 1]);
}
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__global__ void __launch_bounds__(256) k_fused_src(u64 *__restrict__ ht, uint32_t mask, uint32_t n, const uint8_t *__restrict__ text, uint32_t epoch,
                                                    uint16_t *__restrict__ tags) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint8_t *s = text + 16ull * (n - 1 - t);                       // newest first, as the product
    uint32_t h = 28u, f = 0x811C9DC5u;
#pragma unroll
    for (int j = 0; j < 7; j++) { const uint32_t w = ld32(s + 4 * j); h = (h ^ (w + (uint32_t) j)) * 171717u; f = (f ^ w) * 0x9E3779B1u; }
    tags[n - 1 - t] = (uint16_t) epoch;
    const uint32_t b = h & mask;
    const u64 key = ((u64) epoch << 42) | ((u64) (n - 1 - t) << 10) | (f >> 22);
    const u64 v = ht[b];
    if (third(v, t) && v < key) atomicMax(&ht[b], key);
}
__global__ void k_fill_text(uint8_t *p, u64 n) {
    const u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = "ACGT"[mix((uint32_t) i) & 3u];
}

int main() {
    const uint32_t buckets = 1u << 28, mask = buckets - 1, N = 9690000;
    u64 *ht, *log, *sink; unsigned int *cnt; uint8_t *text; uint16_t *tags;
    CHECK(hipMalloc(&ht, (size_t) buckets * 8)); CHECK(hipMalloc(&log, (size_t) N * 16)); CHECK(hipMalloc(&sink, 8)); CHECK(hipMalloc(&cnt, 4));
    CHECK(hipMalloc(&text, (size_t) N * 16 + 64)); CHECK(hipMalloc(&tags, (size_t) N * 2));
    CHECK(hipMemset(ht, 0, (size_t) buckets * 8));
    k_fill_text<<<dim3((unsigned) (((u64) N * 16 + 64 + 255) / 256)), dim3(256)>>>(text, (u64) N * 16 + 64);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const dim3 blk(256), grid((N + 255) / 256);
    uint32_t epoch = 1;
    printf("{\"samples\": %u, \"table_bytes\": %llu, \"variants\": {\n", N, (u64) buckets * 8);
    auto run = [&](const char *name, auto launch, bool last = false) {
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipMemsetAsync(cnt, 0, 4));
            CHECK(hipEventRecord(a));
            launch(rep * 7919u + 13u, epoch++);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep) { sum += ms; if (ms < best) best = ms; }
        }
        printf("  \"%s\": {\"ms_best\": %.4f, \"ms_mean\": %.4f, \"G_samples_per_s\": %.2f}%s\n", name, best, sum / 5, N / best / 1e6, last ? "" : ",");
        fflush(stdout);
    };
    run("read_only", [&](uint32_t s, uint32_t) { k_read_only<<<grid, blk>>>(ht, mask, N, s, sink); });
    run("atomic_all", [&](uint32_t s, uint32_t e) { k_atomic<<<grid, blk>>>(ht, mask, N, s, e, 0); });
    run("atomic_third", [&](uint32_t s, uint32_t e) { k_atomic<<<grid, blk>>>(ht, mask, N, s, e, 1); });
    run("fused", [&](uint32_t s, uint32_t e) { k_fused<1><<<grid, blk>>>(ht, mask, N, s, e); });
    run("fused_x2", [&](uint32_t s, uint32_t e) { k_fused<2><<<dim3((N / 2 + 255) / 256), blk>>>(ht, mask, N, s, e); });
    run("fused_x4", [&](uint32_t s, uint32_t e) { k_fused<4><<<dim3((N / 4 + 255) / 256), blk>>>(ht, mask, N, s, e); });
    run("fused_ret_log", [&](uint32_t s, uint32_t e) { k_fused_ret_log<<<grid, blk>>>(ht, mask, N, s, e, log, cnt); });
    run("split", [&](uint32_t s, uint32_t e) { k_split_read<<<grid, blk>>>(ht, mask, N, s, e, log, cnt); k_split_atomic<<<dim3((N / 2 + 255) / 256), blk>>>(ht, log, cnt); });
    run("fused_sliced", [&](uint32_t s, uint32_t e) { k_fused_sliced<<<grid, blk>>>(ht, 22, N, s, e); });
    run("fused_src", [&](uint32_t, uint32_t e) { k_fused_src<<<grid, blk>>>(ht, mask, N, text, e, tags); }, true);
    printf("}}\n");
    return 0;
}
