// Shared device-side definitions of the MI355X match-finding path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace swk {

constexpr int WAVE = 64;
constexpr int RBU = 256;              // unit of a resolve block's length: a block scans rb * RBU query positions (fine enough for the blocks of a
                                      // batch to fill the wave slots they are sized for: 6 656 positions at configs[2], not 6 144 or 7 168)
constexpr int OVERLAP_MATCH_MAX_LENGTH = 1 << 13;   // SlidingWindowSparseEMMatcher.h:18

// Hash-table entry: (epoch << 32) | (pos >> k1ord). The reference's table is "last writer wins" in
// sample order (SlidingWindowSparseEMMatcher.cpp:146-171); every insertion phase gets a fresh,
// larger epoch and positions ascend inside a phase, so a 64-bit atomicMax reproduces exactly the
// image a single CPU thread leaves behind. The low word alone is the reference's 32-bit entry.
typedef unsigned long long ht_entry;

constexpr uint32_t HIT_CAPL = 1u, HIT_CAPR = 2u;   // the left / right run is known only up to the value held

struct Match {         // == swsem_match_t
    uint64_t posSrc, len, posDest;
};

// per-contig description of a round
struct Contig {
    uint64_t qoff;     // byte offset of the contig in the query buffer
    uint64_t n;        // contig length
    uint64_t lock;     // matchingLockPos or UINT64_MAX
    uint64_t matchBase;// first row of this contig in the batch match array
    uint32_t rb0;      // first resolve block of this contig
    uint32_t nrb;      // resolve blocks (= ceil(positions / (rb * RBU)))
};

// frozen matcher state seen by the kernels of one round
struct RefView {
    const uint8_t *ref;
    const ht_entry *ht;
    uint64_t pos1, refLength, maxRefLength;
    uint32_t mask;
    int fpBits;               // low bits of a table entry that hold the K-mer's fingerprint
    int fpCheck;              // fingerprints may be used to reject entries (see ht_value): 0 no, 1 every entry is as hashed, 2 by epoch
    uint32_t eCur, ePrev;     // first epoch of the current / of the previous lap of the circular buffer (fpCheck == 2)
    uint32_t curMax, prevMin; // entry values (position >> k1ord) up to curMax lie wholly below the loading position, from prevMin on at or above it
    // per sampling slot (position >> k1ord): the lap (as lap_tag) in which the slot was last sampled ON the grid, 0: never, or
    // its K-mer's bytes changed afterwards. tagCur / tagPrev: the tags of the current / previous lap (0: there is none).
    const uint16_t *tags;
    uint32_t tagCur, tagPrev;
    int K, k1ord, skipMargin;
    uint32_t minLen;
};

// little-endian loads at arbitrary byte addresses. gfx950 under HSA runs global memory in unaligned
// access mode, so hipcc lowers these to single global_load_dword / _dwordx4; callers only ask for
// ranges whose bytes are all valid.
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint4 ld_u128(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// maRushPrime1HashSimplified<K>, utils/Hashes.h:28-40, one step (pure u32 arithmetic is exact)
__device__ __forceinline__ uint32_t hash_step(uint32_t h, uint32_t k, uint32_t j) {
    return (h ^ (k + j)) * 171717u;
}

// The fingerprint's own hash of the same dwords (fp_step per dword from FP_SEED; its top F bits are the fingerprint): the
// reference's hash has 32 bits and the bucket index takes 27-29 of them, which would leave a fingerprint of 3-5 bits — and
// a visit of the reference for every 8th-32nd lookup that finds a foreign K-mer's entry, more with every lap as the table
// fills (measured: the step grew from 3.3 to 4.4 ms over three laps). A function of the K-mer's bytes like the other one.
constexpr uint32_t FP_SEED = 0x811C9DC5u;
__device__ __forceinline__ uint32_t fp_step(uint32_t f, uint32_t k) {
    return (f ^ k) * 0x9E3779B1u;
}

// Table entry: (epoch << (32 + F)) | (position >> k1ord) << F | fingerprint, F = fpBits. atomicMax on it
// gives the reference's last-writer-wins (later load phases carry larger epochs, later positions of one
// phase larger values). The fingerprint is the top F bits of the K-mer's second hash (fp_step) — the bucket index
// is the low bits of the first — so an entry left by a different K-mer is told apart without touching the reference
// (the reference finds out with memcmp, .cpp:298, and moves on without any other effect). That is exact
// as long as the bytes an entry was hashed from are still the bytes at its position, i.e. until the
// buffer wraps (or a separator replaces an already hashed byte): from then on a stale entry may point
// at new text that does equal the query, the reference would match it, and the filter is switched off
// (fpCheck = 0; entries keep their layout).
// (`hash` here and in ht_value: the K-mer's SECOND hash)
__device__ __forceinline__ ht_entry ht_key(uint32_t epoch, uint32_t value, uint32_t hash, int fpBits) {
    return ((ht_entry) epoch << (32 + fpBits)) | ((ht_entry) value << fpBits) | (ht_entry) (fpBits ? hash >> (32 - fpBits) : 0u);
}
// stored position of an entry, or 0 when the bucket is empty or holds another K-mer's fingerprint. A fingerprint
// mismatch only proves anything while the entry's bytes are still the bytes it was hashed from. Once the circular
// buffer has wrapped that is decided per entry: the loader writes sequentially, so the text below the loading
// position belongs to the current lap (epochs >= eCur) and the text above it to the previous one (epochs in
// [ePrev, eCur)); an entry whose epoch is older than its region's lap, or whose K-mer straddles the loading
// position, may describe overwritten text — the reference would still follow it and compare bytes (golden case
// rounds3_wrap has such matches) — so it is passed on whatever its fingerprint. Epoch 0 marks the one entry whose
// K-mer a separator overwrote after it had been hashed (k_mark_stale). (Positions stored after a wrap are one off
// the sample's, .cpp quirk: the straddle test leaves a byte of margin.)
// LAPS = false: the caller knows the buffer has not wrapped (fpCheck <= 1; the lap epochs then cost the kernel no registers).
template <bool LAPS = true>
__device__ __forceinline__ uint32_t ht_value(const RefView &v, ht_entry e, uint32_t hash) {
    const uint32_t fp = (uint32_t) e & ((1u << v.fpBits) - 1u);
    if (v.fpCheck && fp != (hash >> (32 - v.fpBits))) {
        const uint32_t epoch = (uint32_t) (e >> (32 + v.fpBits));
        bool trusted = epoch != 0;
        if (LAPS && v.fpCheck == 2 && trusted) {
            const uint32_t val = (uint32_t) (e >> v.fpBits);
            trusted = val <= v.curMax ? epoch >= v.eCur : (val >= v.prevMin && epoch >= v.ePrev);
        }
        if (trusted) return 0u;
    }
    return (uint32_t) (e >> v.fpBits);
}

// An untrusted entry (stale: its text has been overwritten since it was written) must be followed — the reference compares
// bytes — unless the comparison is known to fail: when the slot it points at was sampled again, on the grid, by the very
// load that wrote the text now standing there, that sample went to the bucket of the NEW K-mer with a newer epoch; had that
// been this bucket, the stale entry would be gone. So the text there hashes elsewhere, hence differs from every K-mer that
// hashes here, and memcmp (.cpp:298) fails without any other effect. lap_want gives the tag such a sampling left: the
// current lap's for slots wholly below the loading position, the previous lap's for slots at or above it, none for a slot
// that straddles it. (After many laps about 40 % of the buckets hold stale entries: each costs one 2-byte lookup here
// instead of a 256-byte visit.)
__host__ __device__ __forceinline__ uint32_t lap_tag(int lap) { return 1u + (uint32_t) lap % 65535u; }
__device__ __forceinline__ uint32_t lap_want(const RefView &v, uint32_t val) {
    return val <= v.curMax ? v.tagCur : (val >= v.prevMin ? v.tagPrev : 0u);
}

// value e of entry hte has passed the window test: is it a stale entry whose slot's tag settles that it cannot verify?
template <bool LAPS>
__device__ __forceinline__ bool stale_settled(const RefView &v, ht_entry hte, uint32_t e) {
    if (!LAPS || !v.tags) return false;
    const uint32_t ep = (uint32_t) (hte >> (32 + v.fpBits));
    if (ep == 0 || (e <= v.curMax ? ep >= v.eCur : (e >= v.prevMin && ep >= v.ePrev))) return false;   // marked by k_mark_stale / as young as the text
    const uint32_t want = lap_want(v, e);
    return want != 0 && v.tags[e] == want;
}

// window test of SlidingWindowSparseEMMatcher.cpp:212-222. Returns false when the entry is rejected.
__device__ __forceinline__ bool window_ok(const RefView &v, uint64_t lock, uint64_t c, uint64_t &lo, uint64_t &hi) {
    const uint64_t swStart = v.pos1;
    const uint64_t swStop = lock != UINT64_MAX ? lock : v.pos1;
    const bool endsBefore = c + (uint64_t) v.K < swStart;
    const bool startsBefore = c < swStop;
    if (swStart <= swStop) {
        if (!endsBefore && startsBefore) return false;
    } else if (!endsBefore || startsBefore)
        return false;
    hi = endsBefore ? swStart : v.refLength;
    lo = startsBefore ? 0 : swStop;
    return true;
}

}  // namespace swk
