// Micro-benchmark behind DESIGN "occupancy filter": what do independent random gathers cost on this part, by table size?
//   hipcc --offload-arch=gfx950 -O3 -o profiles/gather_bench profiles/gather_bench.hip && profiles/gather_bench
// (1) N independent random 4-byte / 8-byte gathers over tables of 32 MB .. 2 GiB (L2 4 MiB per XCD, Infinity Cache 256 MiB,
//     HBM beyond): requests per second.
// (2) the two-step lookup of the resolve kernel: a 1-bit-per-bucket filter (T/64 bytes for a table of T bytes of 8-byte
//     entries) is gathered first, the 8-byte entry only when the bit is set, for set fractions 0.05 .. 1.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <typename T, int PER>
__global__ void __launch_bounds__(256) k_gather(const T *__restrict__ tab, uint32_t mask, uint32_t seed, unsigned long long *__restrict__ sink) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) acc += (unsigned long long) tab[mix(t * PER + k + seed) & mask];
    if (acc == 0x1234567ull) *sink = acc;
}

// filter bit first, entry only when set: bits[b >> 5] >> (b & 31)
template <int PER>
__global__ void __launch_bounds__(256) k_two_step(const uint32_t *__restrict__ bits, const unsigned long long *__restrict__ tab, uint32_t mask, uint32_t seed,
                                                  unsigned long long *__restrict__ sink) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    unsigned long long acc = 0;
    uint32_t b[PER], w[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { b[k] = mix(t * PER + k + seed) & mask; w[k] = bits[b[k] >> 5]; }
#pragma unroll
    for (int k = 0; k < PER; k++) if ((w[k] >> (b[k] & 31u)) & 1u) acc += tab[b[k]];
    if (acc == 0x1234567ull) *sink = acc;
}

__global__ void k_fill_bits(uint32_t *bits, uint64_t words, uint32_t thresh) {
    const uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    uint32_t w = 0;
    for (int j = 0; j < 32; j++) if (mix((uint32_t) (i * 32 + j) ^ 0x5bd1e995u) < thresh) w |= 1u << j;
    bits[i] = w;
}

int main() {
    const uint64_t maxBytes = 2ull << 30;
    void *tab = nullptr;
    uint32_t *bits = nullptr;
    unsigned long long *sink = nullptr;
    CHECK(hipMalloc(&tab, maxBytes));
    CHECK(hipMalloc(&bits, maxBytes / 64));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(tab, 1, maxBytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const uint32_t N = 1u << 26;                       // gathers per launch
    constexpr int PER = 4;
    const dim3 grid(N / PER / 256), blk(256);
    printf("{\"gathers_per_launch\": %u, \"independent\": [\n", N);
    const uint64_t sizes[] = {4ull << 20, 32ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 1ull << 30, 2ull << 30};
    bool first = true;
    for (uint64_t sz : sizes)
        for (int width : {4, 8}) {
            const uint32_t mask = (uint32_t) (sz / width - 1);
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(a));
                if (width == 4) k_gather<uint32_t, PER><<<grid, blk>>>((const uint32_t *) tab, mask, rep * 7919u, sink);
                else k_gather<unsigned long long, PER><<<grid, blk>>>((const unsigned long long *) tab, mask, rep * 7919u, sink);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf("%s  {\"table_bytes\": %llu, \"width\": %d, \"ms\": %.4f, \"G_per_s\": %.2f}", first ? "" : ",\n", (unsigned long long) sz, width, best, N / best / 1e6);
            first = false;
        }
    printf("\n], \"filter_then_entry\": [\n");
    first = true;
    for (uint64_t sz : {1ull << 30, 2ull << 30})
        for (double frac : {0.05, 0.17, 0.3, 0.5, 1.0}) {
            const uint64_t buckets = sz / 8, words = buckets / 32;
            k_fill_bits<<<dim3((unsigned) ((words + 255) / 256)), blk>>>(bits, words, frac >= 1.0 ? 0xFFFFFFFFu : (uint32_t) (frac * 4294967296.0));
            CHECK(hipDeviceSynchronize());
            const uint32_t mask = (uint32_t) (buckets - 1);
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(a));
                k_two_step<PER><<<grid, blk>>>(bits, (const unsigned long long *) tab, mask, rep * 104729u, sink);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf("%s  {\"table_bytes\": %llu, \"filter_bytes\": %llu, \"set_fraction\": %.2f, \"ms\": %.4f, \"G_lookups_per_s\": %.2f}", first ? "" : ",\n",
                   (unsigned long long) sz, (unsigned long long) (words * 4), frac, best, N / best / 1e6);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
