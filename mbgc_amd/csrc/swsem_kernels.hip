// Hand-written CDNA4 (gfx950) kernels of the MBGC match-finding hot path: reference loading
// (copy / reverse complement / sparse k-mer insertion), hash-table probing of every query position,
// and the greedy match resolution with wave-cooperative exact extension. Integer/byte work,
// HBM-bound: no MFMA.
#include "swsem_device.h"

namespace swk {

// ------------------------------------------------------------------------------------------------
// reference loading: SlidingWindowSparseEMMatcher::loadRef (private), .cpp:402-437
// ------------------------------------------------------------------------------------------------

// PgHelpers::upperReverseComplement, utils/helper.cpp:405-410: dst[n-1-i] = LUT[src[i]].
// Each thread produces 4 consecutive destination bytes (one dword store when aligned).
__global__ void __launch_bounds__(256) k_load_rc(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                 uint64_t n, const uint8_t *__restrict__ lut) {
    __shared__ uint8_t slut[256];
    slut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x * 4;
    for (uint64_t d = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4; d < n; d += stride) {
        const uint64_t m = n - d < 4 ? n - d : 4;
#pragma unroll
        for (uint64_t k = 0; k < 4; k++)
            if (k < m) dst[d + k] = slut[src[n - 1 - d - k]];
    }
}

__global__ void k_set_byte(uint8_t *p, uint8_t v) { *p = v; }
__global__ void k_touch() {}                        // a stream's first launch (it is given its hardware queue then)
// keeps being dispatched for as long as it runs (more workgroups than the device holds, each `ticks` of the 100 MHz clock long):
// what a stream that shares its dispatch pipe has to wait for (swsem_runtime.hip, side_streams)
__global__ void __launch_bounds__(256) k_hog(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// The byte writes of a whole finalize call (swsem_finalize_targets): every extension copied to its place
// in the reference buffer by one launch (block = 4096 bytes of one piece, 16 bytes per thread, any
// alignment), then the separator bytes in program order by one thread.
struct CopyPiece { uint64_t dst; const uint8_t *src; uint64_t len; };
struct BytePiece { uint64_t off, val; };
// gate: when not null the launch belongs to a speculative finalize (swsem_emit_batch_begin_spec) and does nothing
// unless the device-side check of the prediction stored a 1 there.
__global__ void __launch_bounds__(256) k_copy_multi(uint8_t *__restrict__ ref, const CopyPiece *__restrict__ pieces,
                                                    const uint64_t *__restrict__ first, int np, const uint32_t *__restrict__ gate) {
    if (gate && *gate == 0) return;
    // (a grid no larger than the device holds at once, walking the blocks: a launch that keeps being dispatched while it runs
    // holds its hardware queue's pipe, and whatever shares that pipe waits for its last workgroup — profiles/queue_pipes.hip)
    for (uint64_t blk = blockIdx.x; blk < first[np]; blk += gridDim.x) {
        int lo = 0, hi = np - 1;                                        // piece p with first[p] <= blk < first[p+1]
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (first[mid] <= blk) lo = mid; else hi = mid - 1;
        }
        const CopyPiece pc = pieces[lo];
        const uint64_t o = (blk - first[lo]) * 4096 + 16 * (uint64_t) threadIdx.x;
        if (o >= pc.len) continue;
        uint8_t *d = ref + pc.dst + o;
        const uint8_t *s = pc.src + o;
        if (o + 16 <= pc.len) {
            uint4 t;
            memcpy(&t, s, 16);
            memcpy(d, &t, 16);
        } else
            for (uint64_t k = 0; k < pc.len - o; k++) d[k] = s[k];
    }
}
// host tables staged in pinned memory -> device (a plain kernel: the runtime's own host-to-device copies can
// block the calling thread for milliseconds on a side stream)
__global__ void __launch_bounds__(256) k_upload(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n) {
    const uint64_t o = 16 * ((uint64_t) blockIdx.x * 256 + threadIdx.x);
    if (o + 16 <= n) {
        uint4 t;
        memcpy(&t, src + o, 16);
        memcpy(dst + o, &t, 16);
    } else
        for (uint64_t k = o; k < n; k++) dst[k] = src[k];
}
// several such copies in one launch (every launch costs the stream 5-10 us however small): block b serves the
// segment whose block range holds it; a segment without a source is zero-filled
struct CopySegs { static constexpr int MAX = 8; uint8_t *dst[MAX]; const uint8_t *src[MAX]; uint64_t bytes[MAX]; uint32_t first[MAX + 1]; int n; };
__global__ void __launch_bounds__(256) k_copy_segs(CopySegs sg) {
    int k = 0;
    while (k + 1 < sg.n && blockIdx.x >= sg.first[k + 1]) k++;
    const uint64_t o = 16 * ((uint64_t) (blockIdx.x - sg.first[k]) * 256 + threadIdx.x), n = sg.bytes[k];
    uint8_t *dst = sg.dst[k];
    const uint8_t *src = sg.src[k];
    if (o + 16 <= n) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (src) memcpy(&t, src + o, 16);
        memcpy(dst + o, &t, 16);
    } else
        for (uint64_t i = o; i < n; i++) dst[i] = src ? src[i] : (uint8_t) 0;
}
__global__ void k_set_bytes(uint8_t *__restrict__ ref, const BytePiece *__restrict__ b, int n, const uint32_t *__restrict__ gate) {
    if (gate && *gate == 0) return;
    for (int i = 0; i < n; i++) ref[b[i].off] = (uint8_t) b[i].val;
}


// A table entry stores its sample's position in units of k1 (htEncodePos, .h:132), so a lookup is sent to the slot's FIRST
// byte whatever byte the sample started at. After a wrap the loader samples at 1 + 16 n until the next call's tail is back
// on the grid (.cpp:407-411 vs :503): such an entry sits in the bucket of the K-mer at 16 n + 1 and sends its
// lookups to 16 n, where the reference compares bytes (memcmp, .cpp:298) — and finds them equal in a run of one letter (an
// N run, a homopolymer), a match it then reports. The fingerprint stands for what that comparison will see, so for a sample
// off the grid it is taken from the K-mer at the slot's first byte, not from the sample's own: a lookup whose fingerprint
// differs from it differs from the bytes at 16 n and may be dropped; one that equals them goes on to the comparison.
__device__ __forceinline__ uint32_t fp_at(const uint8_t *s, int nw) {
    uint32_t f = FP_SEED;
    for (int j = 0; j < nw; j++) f = fp_step(f, ld_u32(s + 4 * j));
    return f;
}

// processIgnoreCollisionsRef, .cpp:146-171. Thread t < nMain inserts the main-loop sample at
// S + t*k1 with epoch `epoch`; thread nMain + u inserts the tail sample T + u*k1 with epoch + 1
// (the tail runs after the main loop on the CPU, so it wins collisions against it).
__global__ void __launch_bounds__(256) k_insert(const uint8_t *__restrict__ ref, ht_entry *__restrict__ ht,
                                                uint64_t S, uint64_t nMain, uint64_t T, uint64_t nTail, int k1,
                                                int k1ord, int K, uint32_t mask, uint32_t epoch, int fpBits,
                                                uint16_t *__restrict__ tags, uint32_t tag) {
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nMain + nTail) return;
    const bool tail = t >= nMain;
    const uint64_t p = tail ? T + (t - nMain) * (uint64_t) k1 : S + t * (uint64_t) k1;
    if (tags && (p & ((1ull << k1ord) - 1)) == 0) tags[p >> k1ord] = (uint16_t) tag;      // sampled on the grid in this lap (lap_want)
    const uint8_t *s = ref + p;
    uint32_t h = (uint32_t) K, f = FP_SEED;
    const int nw = K / 4;
    for (int j = 0; j < nw; j++) { const uint32_t w = ld_u32(s + 4 * j); h = hash_step(h, w, (uint32_t) j); f = fp_step(f, w); }
    const uint32_t off = (uint32_t) (p & ((1ull << k1ord) - 1));
    if (off && p >= off) f = fp_at(s - off, nw);                   // a sample off the grid (see fp_at)
    const ht_entry key = ht_key(epoch + (tail ? 1u : 0u), (uint32_t) (p >> k1ord), f, fpBits);
    atomicMax(&ht[h & mask], key);
}

// The same for several loadRef pieces in one launch. Epochs make the result independent of the order in
// which the samples arrive, so all pieces of a round are inserted concurrently once their bytes (and the
// separators) are in place. first[p] = index of piece p's first thread (prefix sums, first[np] = total).
//
// The bytes a piece's samples hash are, for all but a few of them, bytes of the text the piece's loadRef step copies into
// the buffer: reference positions [lo, hi) hold src[0 .. hi - lo). FROM_SRC launches hash those samples from the text
// itself, so the insertion need not wait for the copy (which runs beside it on a stream of its own); the samples whose
// window reaches outside [lo, hi) — into the previous text, a separator — are left to a small launch over the buffer
// once the copy has landed (the host lists them as runs of their own).
struct InsertPiece { uint64_t S, nMain, T, nTail; uint32_t epoch, tag; const uint8_t *src; uint64_t lo, hi; };   // tag: lap_tag of the lap being loaded
__device__ __forceinline__ bool insert_sample(const InsertPiece *__restrict__ pieces, const uint64_t *__restrict__ first, int np, int k1, uint64_t g,
                                              InsertPiece &pc, uint64_t &p, bool &tail) {
    int lo = 0, hi = np;                               // largest p with first[p] <= g
    while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (first[mid] <= g) lo = mid; else hi = mid; }
    pc = pieces[lo];
    const uint64_t t = g - first[lo];
    tail = t >= pc.nMain;
    p = tail ? pc.T + (t - pc.nMain) * (uint64_t) k1 : pc.S + t * (uint64_t) k1;
    return true;
}
template<bool FROM_SRC>
__global__ void __launch_bounds__(256) k_insert_multi(const uint8_t *__restrict__ ref, ht_entry *__restrict__ ht,
                                                      const InsertPiece *__restrict__ pieces, const uint64_t *__restrict__ first,
                                                      int np, int k1, int k1ord, int K, uint32_t mask, int fpBits,
                                                      const uint32_t *__restrict__ gate, uint16_t *__restrict__ tags) {
    if (gate && *gate == 0) return;
    const uint64_t total = first[np];
    // (the grid may be smaller than the samples: see k_copy_multi)
    for (uint64_t g0 = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += (uint64_t) gridDim.x * blockDim.x) {
        // The newest samples first: the table keeps the largest key of a bucket, a round's targets share most of their
        // K-mers, and a sample that finds its bucket already holding a larger key (checked with a plain read below) has
        // nothing to do — scattered atomics are what this kernel's time is made of (0.28 -> 0.23 ms).
        const uint64_t g = total - 1 - g0;
        InsertPiece pc; uint64_t p; bool tail;
        insert_sample(pieces, first, np, k1, g, pc, p, tail);
        const uint8_t *s = ref + p;
        if (FROM_SRC) {
            if (!pc.src || p < pc.lo || p + (uint64_t) K > pc.hi) continue;
            s = pc.src + (p - pc.lo);
        }
        if (tags && (p & ((1ull << k1ord) - 1)) == 0) tags[p >> k1ord] = (uint16_t) pc.tag;    // sampled on the grid in this lap (lap_want)
        uint32_t h = (uint32_t) K, f = FP_SEED;
        const int nw = K / 4;
        for (int j = 0; j < nw; j++) { const uint32_t w = ld_u32(s + 4 * j); h = hash_step(h, w, (uint32_t) j); f = fp_step(f, w); }
        const uint32_t off = (uint32_t) (p & ((1ull << k1ord) - 1));
        if (off && p >= off && (!FROM_SRC || p - off >= pc.lo)) f = fp_at(s - off, nw);   // a sample off the grid (see fp_at)
        const ht_entry key = ht_key(pc.epoch + (tail ? 1u : 0u), (uint32_t) (p >> k1ord), f, fpBits);
        const ht_entry seen = ht[h & mask];
        if (seen >= key) continue;                         // (entries only grow: a stale read shows a smaller one at worst, then the atomic decides)
        atomicMax(&ht[h & mask], key);
    }
}
// The K-mer starting at p is about to lose its last byte to a separator (loadSeparator at the window's end, .cpp:439-451)
// after it may have been hashed: if the table still holds its sample, the entry keeps its position — the reference
// would still follow it — but its epoch becomes 0 = "do not trust the fingerprint" (ht_value).
__global__ void k_mark_stale(const uint8_t *__restrict__ ref, ht_entry *__restrict__ ht, uint64_t p, int K, int k1ord, uint32_t mask, int fpBits,
                             uint16_t *__restrict__ tags) {
    if (tags) tags[p >> k1ord] = 0;                    // that slot's bytes are about to change without a new sample
    uint32_t h = (uint32_t) K;
    for (int j = 0; j < K / 4; j++) h = hash_step(h, ld_u32(ref + p + 4 * j), (uint32_t) j);
    const ht_entry e = ht[h & mask];
    if ((uint32_t) (e >> fpBits) == (uint32_t) (p >> k1ord) && e != 0)
        ht[h & mask] = e & ((((ht_entry) 1) << (32 + fpBits)) - 1);
}

__global__ void __launch_bounds__(256) k_ht_low_words(const ht_entry *__restrict__ ht, uint32_t *__restrict__ out, uint64_t n, int fpBits) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t) (ht[i] >> fpBits);
}

// ------------------------------------------------------------------------------------------------
// greedy resolution: the match list semantics of .cpp:250-315, replayed on the candidate array.
// One wave per chain; every lane runs the same (wave-uniform) automaton, 64 positions are fetched per
// step and skipped positions are jumped over with a ballot. Reference and query bytes are only touched
// unless a capped run has to be continued.
// ------------------------------------------------------------------------------------------------
struct Chain {
    int32_t scan;          // next query position the sequential loop would visit (contigs stay below 2^31 bytes: the
                           // whole automaton runs on 32-bit scalar arithmetic, 64-bit compares would go through the VALU)
    int32_t minTouched;    // lowest stack index examined since the last reset (-1: walked off the bottom)
    int32_t minKeep;       // lowest keepCount of an emission since the last reset
    int32_t visited;       // hits visited since the last reset
    uint32_t cands;        // candidates fetched (diagnostics)
    bool emitted;          // set by an emission (cleared by whoever watches for it)
#ifdef SWSEM_DIAG_PHASES
    uint64_t tRefill, tVisit, tLcp;   // shader-clock ticks between issuing a window's table gathers / a visit's loads / a run continuation and having the data
    uint32_t nRefill, nLcp;
#endif
};
#ifdef SWSEM_DIAG_PHASES
__device__ unsigned long long g_diag[8];   // sums over the blocks of a launch: total, refill wait, visit wait, refills, visits, lcp wait, lcp steps, blocks
#endif

constexpr int RING = 64;

// wave-uniform values belong in SGPRs: lane reads with a uniform lane index / first-lane broadcasts keep
// the whole automaton on the scalar unit and out of the vector register file
__device__ __forceinline__ uint32_t rl32(uint32_t x, int lane) { return (uint32_t) __builtin_amdgcn_readlane((int) x, lane); }
__device__ __forceinline__ uint32_t rfl32(uint32_t x) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) x); }
__device__ __forceinline__ uint64_t rfl64(uint64_t x) { return ((uint64_t) rfl32((uint32_t) (x >> 32)) << 32) | rfl32((uint32_t) x); }
constexpr int OVERLAP_MAX = 1024;      // warm-up positions of a speculative block chain, at most; the launch says how many (run_batch adapts it)
constexpr int SNAP = 4;                // stack elements snapshotted at a block boundary / end

// A row of a block chain's stack: the match plus the scan position the chain had right after emitting it.
// Two chains that emit the same row and have the same scan position afterwards are in the same state as
// far as everything above that row is concerned (k_stitch re-synchronises a replay on that).
struct __attribute__((aligned(16))) Row {
    uint64_t posSrc, len, posDest;
    int64_t scanAfter;
};
__device__ __forceinline__ void put_row(Match *dst, const Match &m, int64_t) { *dst = m; }
__device__ __forceinline__ void put_row(Row *dst, const Match &m, int64_t scan) {
    Row r;
    r.posSrc = m.posSrc; r.len = m.len; r.posDest = m.posDest; r.scanAfter = scan;
    *dst = r;
}

// A chain's match stack held in one array (global memory) with the top RING entries mirrored in LDS.
template <class R>
struct ArrayStack {
    R *st;
    uint2 *ring;
    int32_t sp, ringLow;
    __device__ __forceinline__ int size() const { return sp; }
    __device__ __forceinline__ void get(int idx, int32_t &posDest, int32_t &len) {
        if (idx >= ringLow) {
            const uint2 e = ring[idx & (RING - 1)];
            posDest = (int32_t) rfl32(e.x); len = (int32_t) rfl32(e.y);
        } else {
            posDest = (int32_t) rfl32((uint32_t) st[idx].posDest); len = (int32_t) rfl32((uint32_t) st[idx].len);
        }
    }
    // resMatches.resize(keep); resMatches.push_back(m)   (.cpp:299-300)
    __device__ __forceinline__ void truncate_push(int keep, const Match &m, int32_t scanAfter) {
        sp = keep;
        if ((threadIdx.x & (WAVE - 1)) == 0) put_row(st + sp, m, scanAfter);
        ring[sp & (RING - 1)] = make_uint2((uint32_t) m.posDest, (uint32_t) m.len);
        if (sp - (RING - 1) > ringLow) ringLow = sp - (RING - 1);
        if (ringLow > sp) ringLow = sp;
        sp++;
    }
};

// The true match list of a contig while the stitch kernel assembles it: a chain of per-block
// segments (block b contributes region_b[segStart[b] .. +keepN[b])), newest segment on top, plus the
// pushes of the block being replayed ("own"). Elements are only ever consulted from the top down.
// The newest segment's row count and predecessor live in registers (topKeep, topPrev); keepN[] in
// memory is valid for every segment below it.
struct VirtStack {
    Row *region;              // block regions of this contig
    uint32_t cap;             // rows per block region
    uint32_t *segStart, *keepN;
    int32_t *prev;
    int32_t segTop;           // newest non-empty segment, -1 if none
    int32_t topKeep, topPrev; // rows / predecessor of segTop
    int32_t size_;            // rows in the whole list
    Row *own; int32_t ownN;   // pushes of the block being replayed
    int32_t curSeg, curLocal, curR;   // cursor: element curLocal of segment curSeg is curR rows below the top of the segments
    __device__ __forceinline__ int size() const { return size_; }
    __device__ __forceinline__ int keep_of(int seg) const { return seg == segTop ? topKeep : (int) rfl32(keepN[seg]); }
    __device__ __forceinline__ int prev_of(int seg) const { return seg == segTop ? topPrev : (int) rfl32((uint32_t) prev[seg]); }
    __device__ const Row *at(int idx) {
        const int below = size_ - ownN;              // rows held by the segments
        if (idx >= below) return own + (idx - below);
        const int r = below - 1 - idx;
        if (curSeg < 0 || curR > r) { curSeg = segTop; curLocal = topKeep - 1; curR = 0; }
        while (curR < r) {
            const int step = r - curR < curLocal ? r - curR : curLocal;
            if (step > 0) { curLocal -= step; curR += step; }
            else { curSeg = prev_of(curSeg); curLocal = keep_of(curSeg) - 1; curR++; }
        }
        return region + (uint64_t) curSeg * cap + rfl32(segStart[curSeg]) + curLocal;
    }
    __device__ __forceinline__ void get(int idx, int32_t &posDest, int32_t &len) {
        const Row *m = at(idx);
        posDest = (int32_t) rfl32((uint32_t) m->posDest); len = (int32_t) rfl32((uint32_t) m->len);
    }
    __device__ void pop_segments(int p) {
        const bool l0 = (threadIdx.x & (WAVE - 1)) == 0;
        while (p > 0) {
            const int take = topKeep < p ? topKeep : p;
            topKeep -= take;
            p -= take;
            if (topKeep == 0) {                      // the segment is gone: its predecessor becomes the top
                if (l0) keepN[segTop] = 0;
                const int nt = topPrev;
                if (nt >= 0) { topKeep = (int) rfl32(keepN[nt]); topPrev = (int) rfl32((uint32_t) prev[nt]); }
                segTop = nt;
            }
        }
        curSeg = -1;
    }
    __device__ void truncate_push(int keep, const Match &m, int32_t scanAfter) {
        int p = size_ - keep;
        const int t = p < ownN ? p : ownN;
        ownN -= t; p -= t;
        if (p > 0) pop_segments(p);
        if ((threadIdx.x & (WAVE - 1)) == 0) put_row(own + ownN, m, scanAfter);
        ownN++;
        size_ = keep + 1;
    }
    __device__ void push_segment(int b, uint32_t start, uint32_t n) {
        if (n == 0) return;
        if ((threadIdx.x & (WAVE - 1)) == 0) {
            if (segTop >= 0) keepN[segTop] = (uint32_t) topKeep;
            segStart[b] = start; prev[b] = segTop;
        }
        topPrev = segTop;
        segTop = b;
        topKeep = (int32_t) n;
        size_ += (int32_t) n;
        curSeg = -1;
    }
    __device__ void flush() { if (segTop >= 0 && (threadIdx.x & (WAVE - 1)) == 0) keepN[segTop] = (uint32_t) topKeep; }
};

// Wave-cooperative exact comparison, 256 bytes per step: equal bytes of a[n..limit) vs b[n..limit)
// given that the first n are equal. steps > 0 bounds the number of wave steps; *more tells the caller
// the run may continue past the value returned.
__device__ uint32_t wave_lcp_fwd(const uint8_t *a, const uint8_t *b, uint32_t n, uint32_t limit, int steps, bool &more) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    more = false;
    while (n < limit) {
        const uint32_t off = n + 4 * lane;
        uint32_t eq = 4;
        bool stop = false;
        if (off + 4 <= limit) {
            const uint32_t x = ld_u32(a + off) ^ ld_u32(b + off);
            if (x) { eq = (uint32_t) (__builtin_ctz(x) >> 3); stop = true; }
        } else {
            eq = 0; stop = true;
            if (off < limit) {
                const uint32_t avail = limit - off;
                while (eq < avail && a[off + eq] == b[off + eq]) eq++;
            }
        }
        const unsigned long long bal = __ballot(stop);
        if (bal) {
            const int l = __builtin_ctzll(bal);
            return n + 4 * (uint32_t) l + rl32(eq, l);
        }
        n += 4 * WAVE;
        if (steps && --steps == 0 && n < limit) { more = true; return n; }
    }
    return limit;
}
// the same to the left: equal bytes of a[-1-k] vs b[-1-k], k in [n, limit)
__device__ uint32_t wave_lcp_bwd(const uint8_t *a, const uint8_t *b, uint32_t n, uint32_t limit, int steps, bool &more) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    more = false;
    while (n < limit) {
        const uint32_t off = n + 4 * lane;
        uint32_t eq = 4;
        bool stop = false;
        if (off + 4 <= limit) {
            const uint32_t x = ld_u32(a - off - 4) ^ ld_u32(b - off - 4);
            if (x) { eq = (uint32_t) (__builtin_clz(x) >> 3); stop = true; }
        } else {
            eq = 0; stop = true;
            if (off < limit) {
                const uint32_t avail = limit - off;
                while (eq < avail && a[-(int64_t) (off + eq) - 1] == b[-(int64_t) (off + eq) - 1]) eq++;
            }
        }
        const unsigned long long bal = __ballot(stop);
        if (bal) {
            const int l = __builtin_ctzll(bal);
            return n + 4 * (uint32_t) l + rl32(eq, l);
        }
        n += 4 * WAVE;
        if (steps && --steps == 0 && n < limit) { more = true; return n; }
    }
    return limit;
}

// Processes one visited hit. All arguments are wave-uniform. Returns true when a match was emitted.
template <class Stack>
__device__ bool process_hit(const RefView &v, const Contig &cg, const uint8_t *q, Stack &stk, Chain &ch,
                            uint64_t c, int32_t i, int32_t ell, int32_t rext, int32_t loDist, uint32_t flags) {
    const int32_t K = v.K;
    uint64_t lo = 0, hi = 0;
    if (flags) window_ok(v, cg.lock, c, lo, hi);
    // true left run on demand: the automaton below only ever asks "is ell >= x", so a capped run is
    // continued the first time it could matter
    auto need_ell = [&](int32_t x) {
        if ((flags & HIT_CAPL) && x > ell) {
            const uint64_t d = c - lo;
            const uint32_t jmax = (uint64_t) i < d ? (uint32_t) i : (uint32_t) d;
            bool more;
#ifdef SWSEM_DIAG_PHASES
            const uint64_t dl0 = __builtin_amdgcn_s_memtime();
#endif
            ell = (int32_t) wave_lcp_bwd(v.ref + c, q + i, (uint32_t) ell, jmax, 0, more);
#ifdef SWSEM_DIAG_PHASES
            ch.tLcp += __builtin_amdgcn_s_memtime() - dl0; ch.nLcp++;
#endif
            flags &= ~HIT_CAPL;
        }
    };
    int32_t s = 0;                                   // (p1, p2) = (c - s, i - s)
    int keep = stk.size();                           // resSizeWithoutOverlapped
    bool brokeOut = false;
    while (keep-- > 0) {                             // .cpp:253
        int32_t mPos, mLen;
        stk.get(keep, mPos, mLen);
        const int32_t mEnd = mPos + mLen;
        if (mEnd < i - s) {                          // .cpp:255
            const int32_t g = mEnd;
            need_ell((loDist < i - g + 1 ? loDist : i - g + 1) - 1);
            int32_t t = loDist < i - g + 1 ? loDist : i - g + 1;
            if (ell + 1 < t) t = ell + 1;
            if (t > s) s = t;                        // .cpp:257-259
            if (i - s > g - 1) { brokeOut = true; break; }   // .cpp:260-261
            s -= 1;                                  // .cpp:262
        }
        const int32_t d = (i - s) - mPos;            // lastDelta, .cpp:264
        bool fail = (loDist - s < d) || (mLen > OVERLAP_MATCH_MAX_LENGTH);
        if (!fail) {
            need_ell(s + d);
            fail = ell < s + d;                      // strcmplcp(...) != 0, .cpp:266
        }
        if (fail) { s += 1; brokeOut = true; break; }        // .cpp:267-268
        s += d;                                      // .cpp:270
    }
    if (!brokeOut) keep = -1;
    if (keep < ch.minTouched) ch.minTouched = keep;
    ch.visited++;
    if (keep < 0) {                                  // .cpp:277-280
        need_ell((loDist < i + 1 ? loDist : i + 1) - 1);
        int32_t t = loDist < i + 1 ? loDist : i + 1;
        if (ell + 1 < t) t = ell + 1;
        if (t > s) s = t;
    } else {                                         // .cpp:285-289
        int32_t mPos, mLen;
        stk.get(keep, mPos, mLen);
        const int32_t overlap = (mPos + mLen) - (i - s + 1);
        if (overlap > 0) s -= overlap;
    }
    ++keep;
    // right1 - p1 > minMatchLength, .cpp:298 (the K-mer itself was verified by the probe kernel)
    if (K + rext + s > (int32_t) v.minLen || (flags & HIT_CAPR)) {
        if (flags & HIT_CAPR) {
            const uint64_t a = hi - (c + K);
            const uint32_t b = (uint32_t) cg.n - (uint32_t) (i + K);
            bool more;
#ifdef SWSEM_DIAG_PHASES
            const uint64_t dl0 = __builtin_amdgcn_s_memtime();
#endif
            rext = (int32_t) wave_lcp_fwd(v.ref + c + K, q + i + K, (uint32_t) rext, a < b ? (uint32_t) a : b, 0, more);
#ifdef SWSEM_DIAG_PHASES
            ch.tLcp += __builtin_amdgcn_s_memtime() - dl0; ch.nLcp++;
#endif
            flags &= ~HIT_CAPR;
        }
        if (K + rext + s > (int32_t) v.minLen) {
            Match m;
            m.posSrc = c - (uint64_t) (int64_t) (s - 1);
            m.len = (uint64_t) (uint32_t) (K + rext + s - 1);
            m.posDest = (uint64_t) (uint32_t) (i - s + 1);
            int32_t skip = K + rext;                 // (matchEnd - i2), k2 == 1, .cpp:308
            skip -= skip > v.skipMargin ? v.skipMargin : skip;
            ch.scan = skip ? i + skip : i + 1;       // .cpp:310-313 then the loop's i2 += k2
            stk.truncate_push(keep, m, ch.scan);
            if (keep < ch.minKeep) ch.minKeep = keep;
            ch.emitted = true;
            return true;
        }
    }
    ch.scan = i + 1;
    return false;
}

// A visited candidate: verify the K-mer (memcmp(curr1, curr2, K), .cpp:298 — a failed candidate changes
// no state in the reference either) and take the first step of the right run (.cpp:227-246) and of the
// left run. All of it is ONE dword load per lane from the reference and one from the query: the wave
// reads the 256 bytes starting LEFTW bytes before the candidate, so lanes [0, LEFTW/4) hold the left
// run, the next K/4 lanes the K-mer and the rest the first 256 - LEFTW - K bytes of the right run. (The
// chains keep every CU's vector-memory pipeline busy — a wave-wide load occupies it for 16 cycles
// whatever its width — so a visit is priced in load instructions, not in bytes.) A lane whose dword
// crosses a limit of its run is not loaded and ends the run there; if the run really reaches it the last
// < 4 bytes are compared one by one. Longer runs continue 256 bytes per step (wave_lcp_fwd / _bwd).
constexpr int LEFTW = 64;

template <class Stack>
__device__ void visit(const RefView &v, const Contig &cg, const uint8_t *q, Stack &stk, Chain &ch, int32_t i, uint32_t val) {
    const int32_t lane = (int32_t) (threadIdx.x & (WAVE - 1));
    ch.cands++;
    const uint64_t c = (uint64_t) val << v.k1ord;                     // htDecodePos, .h:133
    uint64_t lo, hi;
    window_ok(v, cg.lock, c, lo, hi);
    const int32_t K = v.K;
    const uint8_t *r0 = v.ref + c, *q0 = q + i;
    const uint64_t ra = hi - (c + K);
    const uint32_t rb = (uint32_t) cg.n - (uint32_t) (i + K);
    const int32_t limR = (int32_t) (ra < rb ? (uint32_t) ra : rb);
    const uint64_t d = c - lo;
    const int32_t limL = (uint64_t) i < d ? i : (int32_t) d;
    constexpr int LL = LEFTW / 4;                                     // lanes of the left run
    const int32_t rel = 4 * lane - LEFTW;                             // first byte of the lane's dword, relative to the candidate
    const bool full = lane < LL ? -rel <= limL : rel + 4 <= K + limR; // the dword lies inside its run's limit
    uint32_t x = 0;
#ifdef SWSEM_DIAG_PHASES
    const uint64_t dv0 = __builtin_amdgcn_s_memtime();
#endif
    if (full) x = ld_u32(r0 + rel) ^ ld_u32(q0 + rel);
    const unsigned long long stopAll = __ballot(!full || x != 0);
#ifdef SWSEM_DIAG_PHASES
    ch.tVisit += __builtin_amdgcn_s_memtime() - dv0;
#endif
    const unsigned long long fwd = stopAll >> LL;                     // bit t: lane LL + t
    if (fwd & ((1ull << (K / 4)) - 1)) { ch.scan = i + 1; return; }   // K-mer differs (its lanes are always inside the limits)
    const unsigned long long fullAll = __ballot(full);
    int32_t rext = 4 * (WAVE - LL) - K;
    const bool capR = fwd == 0;
    if (!capR) {
        const int l = LL + __builtin_ctzll(fwd);
        rext = 4 * (l - LL) - K;
        if ((fullAll >> l) & 1) rext += __builtin_ctz(rl32(x, l)) >> 3;
        else while (rext < limR && r0[K + rext] == q0[K + rext]) rext++;      // the lane that holds the limit
    }
    const uint32_t bwd = (uint32_t) stopAll & ((1u << LL) - 1);
    int32_t ell = LEFTW;
    const bool capL = bwd == 0;
    if (!capL) {
        const int l = 31 - __builtin_clz(bwd);                        // the stopping lane nearest to the candidate
        ell = 4 * (LL - 1 - l);
        if ((fullAll >> l) & 1) ell += __builtin_clz(rl32(x, l)) >> 3;
        else while (ell < limL && r0[-ell - 1] == q0[-ell - 1]) ell++;
    }
    // loDist only ever competes with distances inside the contig (< 2^31)
    process_hit(v, cg, q, stk, ch, c, i, ell, rext, (int32_t) (d > 0x7FFFFFFFull ? 0x7FFFFFFFull : d),
                (capL ? HIT_CAPL : 0u) | (capR ? HIT_CAPR : 0u));
}

// Runs the chain over the candidates at query positions [p0, p1) of one contig (p1 <= positions).
// Four 64-position batches of the candidate array are fetched per round trip; a jump that lands inside
// the fetched window (the common case: matches of ~100 bases) costs no further load.
struct NoStop { __device__ __forceinline__ bool operator()() { return false; } };

// The hash table is consulted only for the WL positions from the scan position on — the loop visits a
// few positions after every match and jumps ~100 ahead, so most buckets never need to be fetched.
#ifndef SWSEM_WL
#define SWSEM_WL 32
#endif
constexpr int WL = SWSEM_WL;
// Scalar registers of the block-resolve kernel. A SIMD has 800 of them and a wave is given its count plus ~20,
// rounded up to 16: the 106 the compiler takes when left alone allow 6 waves per SIMD, 88 allow 7, 72 allow 8
// (measured with a spinning kernel: 6144 / 7168 / 8192 resident waves chip-wide). The chains are latency-bound,
// so resident waves are throughput — but every register taken away is a spill to a vector lane in the
// automaton's inner loop (124 at 72, the launch 1.12 ms; at 88: 1.07 ms), and a round's blocks are sized to the
// slots there are (run_batch), so 7 per SIMD it is. RESOLVE_WAVES_PER_SIMD must say what this cap allows.
#ifndef SWSEM_RESOLVE_SGPRS
#define SWSEM_RESOLVE_SGPRS 88
#endif
#ifndef SWSEM_RESOLVE_WAVES
#define SWSEM_RESOLVE_WAVES 7
#endif
constexpr int RESOLVE_WAVES_PER_SIMD = SWSEM_RESOLVE_WAVES;
// K-mer hashes of the scan window [s, s + cnt) (cnt <= WL), one position per lane, from the query bytes: the
// window's bytes are loaded once as aligned dwords (one per lane), every lane picks the nine it needs with
// ds_bpermute and shifts them into place. The chains wait on memory most of the time, so the multiplications
// are free there, while a hash array written ahead by its own kernel costs that kernel (0.27 ms per round of
// 80 M positions, at the multiplier's quarter rate) and 4 bytes of HBM traffic per position each way.
__device__ __forceinline__ uint32_t window_hash(const RefView &v, const uint8_t *q, int32_t s, int32_t cnt, int32_t lane, uint32_t &fpHash) {
    const uintptr_t A = (uintptr_t) (q + s);
    const uint32_t sh = (uint32_t) (A & 3);
    const uint32_t *A0 = (const uint32_t *) (A & ~(uintptr_t) 3);
    const uint32_t ndw = (sh + (uint32_t) cnt + (uint32_t) v.K - 1u + 3u) >> 2;     // <= 64 (swsem_create checks K)
    const uint32_t a = (uint32_t) lane < ndw ? A0[lane] : 0u;
    const uint32_t o = sh + (uint32_t) lane, wi = o >> 2, sft = o & 3u;
    const int nw = v.K / 4;
    uint32_t h = (uint32_t) v.K, f = FP_SEED;
    uint32_t lo = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (wi << 2), (int) a);
    for (int x = 0; x < nw; x++) {
        const uint32_t hi = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((wi + (uint32_t) x + 1u) << 2), (int) a);
        const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sft);
        h = hash_step(h, w, (uint32_t) x);
        f = fp_step(f, w);
        lo = hi;
    }
    fpHash = f;
    return h;
}

template <bool LAPS, class Stack, class Stop = NoStop>
__device__ void run_chain_lazy(const RefView &v, const Contig &cg, const uint8_t *q,
                               int32_t p0, int32_t p1, Stack &stk, Chain &ch, Stop stop = Stop()) {
    const int32_t lane = (int32_t) (threadIdx.x & (WAVE - 1));
    int32_t wb = -0x40000000;                                         // window [wb, wb + WL)
    uint32_t w = 0;
    unsigned long long m = 0;
    uint32_t prioTick = 0;
    while (true) {
        const int32_t s = ch.scan > p0 ? ch.scan : p0;
        if (s >= p1) break;
        ch.scan = s;
        if (s < wb || s >= wb + WL) {
            // The SIMD's arbiter serves its oldest wave first, and with every CU's memory pipeline saturated that
            // is a lasting advantage: blocks dispatched first ran 25 % faster than the last ones, and the launch
            // lasts as long as its slowest wave. Each wave therefore walks through the four priority levels,
            // one scan window at a time (spread 761-986 us -> 828-897 us per contig, the launch 1.10 -> 1.02 ms).
            switch ((prioTick++ + blockIdx.x) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
            wb = s;
#ifdef SWSEM_DIAG_PHASES
            const uint64_t dt0 = __builtin_amdgcn_s_memtime();
#endif
            const int32_t pos = s + lane;
            uint32_t e = 0;
            uint32_t hf;
            const uint32_t hv = window_hash(v, q, s, p1 - s < WL ? p1 - s : WL, lane, hf);
            if (lane < WL && pos < p1) {
                const ht_entry hte = v.ht[hv & v.mask];
                e = ht_value<LAPS>(v, hte, hf);
                if (e != 0) {
                    uint64_t lo, hi;
                    if (!window_ok(v, cg.lock, (uint64_t) e << v.k1ord, lo, hi)) e = 0;
                    else if (stale_settled<LAPS>(v, hte, e)) e = 0;
                }
            }
            w = e;
            m = __ballot(w != 0);
#ifdef SWSEM_DIAG_PHASES
            ch.tRefill += __builtin_amdgcn_s_memtime() - dt0; ch.nRefill++;
#endif
        }
        const unsigned long long mk = m & ~((1ull << (s - wb)) - 1);
        if (!mk) { ch.scan = wb + WL < p1 ? wb + WL : p1; continue; }
        const int l = __builtin_ctzll(mk);
        visit(v, cg, q, stk, ch, wb + l, rl32(w, l));
        if (stop()) return;
    }
}


// LAPS: the circular buffer has wrapped (ht_value then tells stale entries by their epochs)
template <bool LAPS, class Stack, class Stop = NoStop>
__device__ __forceinline__ void chain_run(const RefView &v, const Contig &cg, const uint8_t *q, int32_t p0, int32_t p1, Stack &stk, Chain &ch,
                                          Stop stop = Stop()) {
    run_chain_lazy<LAPS>(v, cg, q, p0, p1, stk, ch, stop);
}

// Sequential resolution (one wave replays a whole contig): the simple form, kept as the cross-check
// of the block-parallel path (SWSEM_RESOLVE=seq).
template <bool LAPS>
__global__ void __launch_bounds__(WAVE) k_resolve_seq(RefView v, const uint8_t *__restrict__ qbuf,
                                                      const Contig *__restrict__ contigs,
                                                      Match *__restrict__ matches, uint32_t *__restrict__ matchCount) {
    __shared__ uint2 ring[RING];
    const Contig cg = contigs[blockIdx.x];
    Chain ch;
    ch.scan = 0; ch.minTouched = 0; ch.minKeep = 0; ch.visited = 0; ch.cands = 0; ch.emitted = false;
    ArrayStack<Match> stk;
    stk.st = matches + cg.matchBase; stk.ring = ring; stk.sp = 0; stk.ringLow = 0;
    const int32_t npos = cg.n >= (uint64_t) v.K ? (int32_t) (cg.n - v.K + 1) : 0;
    chain_run<LAPS>(v, cg, qbuf + cg.qoff, 0, npos, stk, ch);
    if (threadIdx.x == 0) matchCount[blockIdx.x] = (uint32_t) stk.sp;
}

// ------------------------------------------------------------------------------------------------
// Block-parallel resolution. The greedy chain is sequential, but its state at a query position is
// tiny (scan position + the few newest matches), and chains started from different states fall into
// step after a few emissions. So every tile gets its own wave: it warms up on the last `overlap`
// positions of the previous tile from an empty stack (speculation), snapshots its state at the tile
// boundary, then replays its own tile. The stitch kernel walks the tiles of a contig in order with
// the TRUE state, accepts a block when its boundary snapshot equals the true state as deep as the
// block ever looked (then everything it did is exactly what the sequential loop does), and replays
// the block from the true state otherwise. Results are therefore identical to the sequential loop by
// construction; speculation only decides how much of the work ran in parallel.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) BlockRec {
    int64_t scanB;         // scan position at the block boundary (clamped to >= block start)
    int64_t scanF;         // scan position after the block
    int32_t spB, spF;      // stack size at the boundary / at the end
    int32_t minTouched;    // lowest index examined while replaying the own tiles (INT_MAX: nothing visited)
    int32_t minKeep;       // rows of the boundary stack that survived the own tiles (<= spB)
    uint64_t cycles;       // s_memtime ticks spent by the block's wave (diagnostics: swsem_debug_block_times)
    uint32_t visits, emits;
    Match bTop[SNAP];      // newest rows at the boundary, newest first
    Match fTop[SNAP];      // newest rows at the end, newest first
};

__device__ __forceinline__ void snapshot_top(const Row *st, int sp, Match *out) {
#pragma unroll
    for (int j = 0; j < SNAP; j++) {
        if (j < sp) { out[j].posSrc = st[sp - 1 - j].posSrc; out[j].len = st[sp - 1 - j].len; out[j].posDest = st[sp - 1 - j].posDest; }
        else { out[j].posSrc = 0; out[j].len = 0; out[j].posDest = 0; }
    }
}

// resolve block rbIdx of a contig = tiles [rbIdx*rb, (rbIdx+1)*rb) of it
template <bool LAPS>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_num_sgpr(SWSEM_RESOLVE_SGPRS))) k_resolve_blocks(RefView v, const uint8_t *__restrict__ qbuf,
                                                         const Contig *__restrict__ contigs,
                                                         const uint32_t *__restrict__ rbContig,
                                                         const uint32_t *__restrict__ order,
                                                         Row *__restrict__ regions, uint32_t cap, uint32_t rb,
                                                         BlockRec *__restrict__ recs, uint32_t overlap) {
    __shared__ uint2 ring[RING];
    // launch slot -> block (run_batch: blocks that scan the same offsets of different contigs sit in slots that are
    // equal mod 8, i.e. on one XCD, next to each other; 0xFFFFFFFF pads the lists of the eight XCDs to one length)
    const uint32_t g = order[blockIdx.x];
    if (g == 0xFFFFFFFFu) return;
    const Contig cg = contigs[rbContig[g]];
    const uint32_t b = g - cg.rb0;
    const int32_t npos = cg.n >= (uint64_t) v.K ? (int32_t) (cg.n - v.K + 1) : 0;
    const int32_t w0 = (int32_t) (b * rb * RBU);
    const int32_t w1 = w0 + (int32_t) (rb * RBU) < npos ? w0 + (int32_t) (rb * RBU) : npos;
    Chain ch;
    const int32_t warm0 = w0 > (int32_t) overlap ? w0 - (int32_t) overlap : 0;   // (from the contig's start the empty state IS the true one)
    ch.scan = b ? warm0 : 0; ch.minTouched = 0x7fffffff; ch.minKeep = 0x7fffffff; ch.visited = 0; ch.cands = 0; ch.emitted = false;
#ifdef SWSEM_DIAG_PHASES
    ch.tRefill = ch.tVisit = ch.tLcp = 0; ch.nRefill = ch.nLcp = 0;
#endif
    const uint64_t tstart = __builtin_amdgcn_s_memtime();
#ifdef SWSEM_DIAG_T0
    const uint64_t rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    ArrayStack<Row> stk;
    stk.st = regions + (uint64_t) g * cap; stk.ring = ring; stk.sp = 0; stk.ringLow = 0;
    const uint8_t *q = qbuf + cg.qoff;
    if (b) chain_run<LAPS>(v, cg, q, warm0, w0, stk, ch);   // warm-up on the previous block's tail
    BlockRec r;
    r.scanB = ch.scan > w0 ? ch.scan : w0;
    r.spB = stk.sp;
    __builtin_amdgcn_s_waitcnt(0);            // the wave's own stack rows are read back below
    snapshot_top(stk.st, stk.sp, r.bTop);
    ch.minTouched = 0x7fffffff; ch.minKeep = stk.sp; ch.visited = 0;
    chain_run<LAPS>(v, cg, q, w0, w1, stk, ch);
    r.scanF = ch.scan;
    r.spF = stk.sp;
    r.minTouched = ch.visited ? ch.minTouched : 0x7fffffff;
    r.minKeep = ch.minKeep < r.spB ? ch.minKeep : r.spB;
    __builtin_amdgcn_s_waitcnt(0);
    snapshot_top(stk.st, stk.sp, r.fTop);
    r.cycles = __builtin_amdgcn_s_memtime() - tstart;
    r.visits = ch.cands; r.emits = (uint32_t) (stk.sp);
#ifdef SWSEM_DIAG_T0
    r.visits = (uint32_t) rt0;                            // diagnostics build: the wave's start and end on the 100 MHz clock
    r.emits = (uint32_t) __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x == 0) recs[g] = r;
#ifdef SWSEM_DIAG_PHASES
    if (threadIdx.x == 0) {
        atomicAdd(&g_diag[0], (unsigned long long) r.cycles); atomicAdd(&g_diag[1], (unsigned long long) ch.tRefill);
        atomicAdd(&g_diag[2], (unsigned long long) ch.tVisit); atomicAdd(&g_diag[3], (unsigned long long) ch.nRefill);
        atomicAdd(&g_diag[4], (unsigned long long) ch.cands); atomicAdd(&g_diag[5], (unsigned long long) ch.tLcp);
        atomicAdd(&g_diag[6], (unsigned long long) ch.nLcp); atomicAdd(&g_diag[7], 1ull);
    }
#endif
}

__device__ __forceinline__ bool same_match(const Match &a, const Match &b) {
    return a.posSrc == b.posSrc && a.len == b.len && a.posDest == b.posDest;
}

// The stitch's acceptance test for block b only looks at the block's own record and — when the block before
// it was accepted as speculated — at that block's final state, which its record also holds. So the test is
// evaluated for every block in parallel, "assuming the predecessor is accepted": the boundary scan position
// equals the predecessor's final one, and the boundary rows the block looked at equal the predecessor's
// newest rows, all of them pushed by the predecessor itself (then they are true rows once it is accepted).
// k_stitch walks the blocks in order as before, but for a block that passes, after a predecessor accepted in
// the same way, only the list bookkeeping is left; everything else takes the complete test below.
struct __attribute__((aligned(16))) FastRec {
    int32_t scanF;         // scan position after the block
    int32_t minKeep;       // first row of the block's region that belongs to the list
    int32_t npush;         // rows the block adds
    int32_t popB;          // rows of the predecessor's it removes; -1: take the complete test
};
__global__ void __launch_bounds__(256) k_stitch_pre(const Contig *__restrict__ contigs, const uint32_t *__restrict__ rbContig,
                                                    const BlockRec *__restrict__ recs, uint32_t rb, uint32_t nblocks,
                                                    FastRec *__restrict__ fast) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nblocks) return;
    const Contig cg = contigs[rbContig[g]];
    const uint32_t b = g - cg.rb0;
    const BlockRec &cur = recs[g];
    FastRec f;
    f.scanF = (int32_t) cur.scanF; f.minKeep = cur.minKeep; f.npush = cur.spF - cur.minKeep; f.popB = -1;
    const bool visited = cur.minTouched != 0x7fffffff;
    if (b == 0) {
        if (visited) f.popB = 0;                                      // started from the true (empty) state: spB = minKeep = 0
    } else if (visited && cur.minTouched >= 0) {
        const BlockRec &prv = recs[g - 1];
        const int32_t span = (int32_t) (rb * RBU), w0 = (int32_t) b * span;
        const int32_t pScanF = (int32_t) prv.scanF;
        const int cmp = cur.spB - cur.minTouched;
        const int prevPush = prv.spF - prv.minKeep;
        bool ok = prv.minTouched != 0x7fffffff && pScanF < w0 + span && (pScanF > w0 ? pScanF : w0) == (int32_t) cur.scanB &&
                  cmp >= 0 && cmp <= SNAP && cmp <= prevPush;
        for (int j = 0; ok && j < cmp; j++)
            ok = cur.bTop[j].posSrc == prv.fTop[j].posSrc && cur.bTop[j].len == prv.fTop[j].len && cur.bTop[j].posDest == prv.fTop[j].posDest;
        if (ok) f.popB = cur.spB - cur.minKeep;                      // <= cmp <= prevPush: stays inside the predecessor's segment
    }
    fast[g] = f;
}

// One wave per contig: walk the resolve blocks with the true state (see above). Outputs per block the
// segment (segStart, keepN) of its region that belongs to the final list, the row offsets and the
// contig's match count. The walk is a dependent chain executed by a single wave, so everything in it
// is either scalar (readfirstlane'd record fields) or one lane-parallel LDS operation: the newest
// SNAP true rows live in LDS as 3*SNAP u64 words and are compared / rebuilt by 3*SNAP lanes at once.
template <bool LAPS>
__global__ void __launch_bounds__(WAVE) k_stitch(RefView v, const uint8_t *__restrict__ qbuf,
                                                 const Contig *__restrict__ contigs,
                                                 Row *__restrict__ regions, Row *__restrict__ replayArea,
                                                 uint32_t cap, uint32_t rb, const BlockRec *__restrict__ recs,
                                                 const FastRec *__restrict__ fast,
                                                 uint32_t *__restrict__ segStart, uint32_t *__restrict__ keepN,
                                                 int32_t *__restrict__ prev, uint32_t *__restrict__ dstOff,
                                                 uint32_t *__restrict__ matchCount,
                                                 unsigned long long *__restrict__ stats) {
    const Contig cg = contigs[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const uint8_t *q = qbuf + cg.qoff;
    VirtStack vs;
    vs.region = regions + (uint64_t) cg.rb0 * cap; vs.cap = cap;
    vs.segStart = segStart + cg.rb0; vs.keepN = keepN + cg.rb0; vs.prev = prev + cg.rb0;
    vs.segTop = -1; vs.topKeep = 0; vs.topPrev = -1; vs.size_ = 0; vs.own = nullptr; vs.ownN = 0; vs.curSeg = -1; vs.curLocal = 0; vs.curR = 0;
    for (uint32_t b = lane; b < cg.nrb; b += WAVE) { vs.keepN[b] = 0; vs.segStart[b] = 0; }
    __builtin_amdgcn_s_waitcnt(0);
    int32_t scanT = 0;                 // true scan position
    constexpr int TW = 3 * SNAP;       // u64 words of the newest-rows window
    __shared__ uint64_t ptop[TW];      // newest true rows, newest first, {posSrc, len, posDest} each
    int known = 0;                     // rows of ptop that are valid
    uint32_t replayed = 0, nFast = 0, nSlow = 0, nSkip = 0;   // (diagnostics: blocks accepted in runs / tested one by one / jumped over)
    const BlockRec *rc = recs + cg.rb0;
    const int32_t span = (int32_t) (rb * RBU);
    const int32_t npos = cg.n >= (uint64_t) v.K ? (int32_t) (cg.n - v.K + 1) : 0;
    __shared__ uint4 srec[sizeof(BlockRec) / 16];
    bool prevPlain = true;             // the block before was accepted as speculated (block 0: the empty state is what it assumed)
    uint4 fr = make_uint4(0, 0, 0, 0); // FastRec of block (b & ~63) + lane
    for (uint32_t b = 0; b < cg.nrb; b++) {
        if ((b & (WAVE - 1)) == 0) fr = b + lane < cg.nrb ? ((const uint4 *) (fast + cg.rb0))[b + lane] : make_uint4(0, 0, 0, 0xFFFFFFFFu);
        const int32_t w0 = (int32_t) b * span;
        if (scanT >= w0 + span) { prevPlain = false; nSkip++; continue; }  // the sequential loop jumped over this block
        const int fl = (int) (b & (WAVE - 1));
        const int32_t fPop = (int32_t) rl32(fr.w, fl);
        if (prevPlain && fPop >= 0) {
            // Accepted as speculated, and so is every block after it whose quick test passed, up to the end of this
            // batch of 64 records: only the list bookkeeping is left, and for such a run it is done by all lanes at
            // once (one block each) instead of block after block. A block of the run removes rows only from the
            // segment of the block right before it (k_stitch_pre: popB <= the predecessor's pushes), and a block
            // that pushes nothing removes nothing, so the rows a segment keeps are its pushes minus the next
            // block's pops, and its predecessor in the list is the nearest earlier block that pushed anything.
            const unsigned long long fastMask = __ballot((int32_t) fr.w >= 0);
            const unsigned long long rest = ~(fastMask >> fl);
            const int L = rest ? __builtin_ctzll(rest) : WAVE - fl;                  // blocks b .. b + L - 1 (>= 1)
            const bool inRun = (int) lane >= fl && (int) lane < fl + L;
            const int32_t myPop = inRun ? (int32_t) fr.w : 0, myPush = inRun ? (int32_t) fr.z : 0;
            int32_t nextPop = __shfl_down(myPop, 1);
            if ((int) lane + 1 >= fl + L) nextPop = 0;
            if (fPop > 0) vs.pop_segments(fPop);                                       // the first block's pops hit the list as it stands
            const unsigned long long segMask = __ballot(inRun && myPush > 0);
            int32_t delta = myPush - myPop;
            for (int d = WAVE / 2; d > 0; d >>= 1) delta += __shfl_xor(delta, d);
            vs.size_ += delta;
            if (segMask) {
                if (lane == 0 && vs.segTop >= 0) vs.keepN[vs.segTop] = (uint32_t) vs.topKeep;
                const int32_t b0 = (int32_t) b - fl;                                  // block of lane 0
                const unsigned long long below = segMask & ((1ull << lane) - 1ull);
                const int32_t pv = below ? b0 + (63 - __builtin_clzll(below)) : vs.segTop;
                const int32_t keepMine = myPush - nextPop;
                if (inRun && myPush > 0) {
                    vs.segStart[b0 + (int32_t) lane] = (uint32_t) fr.y;
                    vs.prev[b0 + (int32_t) lane] = pv;
                    vs.keepN[b0 + (int32_t) lane] = (uint32_t) keepMine;
                }
                const int top = 63 - __builtin_clzll(segMask);
                vs.segTop = b0 + top;
                vs.topKeep = (int32_t) rl32((uint32_t) keepMine, top);
                vs.topPrev = (int32_t) rl32((uint32_t) pv, top);
                vs.curSeg = -1;
            }
            scanT = (int32_t) rl32(fr.x, fl + L - 1);
            known = 0;                                          // the newest-rows window is rebuilt from the list if ever needed
            b += (uint32_t) (L - 1);
            nFast += (uint32_t) L;
            continue;
        }
        nSlow++;
        {                                                       // the complete test needs the whole record
            const uint4 *src = (const uint4 *) (rc + b);
            for (uint32_t i = lane; i < sizeof(BlockRec) / 16; i += WAVE) srec[i] = src[i];
            __builtin_amdgcn_s_waitcnt(0);
        }
        const BlockRec *r = (const BlockRec *) srec;
        const int32_t rScanB = (int32_t) rfl32((uint32_t) r->scanB), rScanF = (int32_t) rfl32((uint32_t) r->scanF);
        const int spB = (int) rfl32((uint32_t) r->spB), spF = (int) rfl32((uint32_t) r->spF);
        const int minTouched = (int) rfl32((uint32_t) r->minTouched), minKeep = (int) rfl32((uint32_t) r->minKeep);
        const uint64_t *bw = (const uint64_t *) r->bTop, *fw = (const uint64_t *) r->fTop;
        const bool visited = minTouched != 0x7fffffff;
        bool ok;
        if (b == 0) ok = true;                                  // block 0 started from the true (empty) state
        else {
            ok = (scanT > w0 ? scanT : w0) == rScanB;
            if (ok && visited) {
                // rows of the boundary stack the block looked at (all of them + "nothing below" when it
                // walked off the bottom)
                const int cmp = minTouched < 0 ? spB : spB - minTouched;
                if (minTouched < 0) ok = vs.size_ == spB && spB <= SNAP;
                else ok = cmp <= SNAP && cmp <= vs.size_;
                if (ok && cmp > known) {                        // refresh the true newest rows from the list
                    __builtin_amdgcn_s_waitcnt(0);
                    const int n = vs.size_ < SNAP ? vs.size_ : SNAP;
                    for (int j = 0; j < n; j++) {
                        const Row *m = vs.at(vs.size_ - 1 - j);
                        if (lane < 3) ptop[3 * j + lane] = ((const uint64_t *) m)[lane];
                    }
                    known = n;
                }
                if (ok) {
                    const bool diff = (int) lane < 3 * cmp && bw[lane < TW ? lane : 0] != ptop[lane < TW ? lane : 0];
                    ok = __ballot(diff) == 0;
                }
            }
        }
        if (ok) {
            if (visited || b == 0) {
                const int popB = spB - minKeep;
                const int npush = spF - minKeep;
                if (popB > 0) vs.pop_segments(popB);
                vs.size_ -= popB;
                vs.push_segment((int) b, (uint32_t) minKeep, (uint32_t) npush);
                // newest rows after the block: its pushes first, then what is left of the old window
                const int keepOld = known > popB ? known - popB : 0;
                if (lane < TW) {
                    const int row = lane / 3, fld = lane % 3;
                    uint64_t x = 0;
                    if (row < npush) x = fw[lane];
                    else if (row - npush < keepOld) x = ptop[3 * (row - npush + popB) + fld];
                    ptop[lane] = x;
                }
                known = npush + keepOld < SNAP ? npush + keepOld : SNAP;
                scanT = rScanF;
                prevPlain = true;                               // its rows and final state are now known to be the true ones
            } else
                prevPlain = false;
        } else {
            prevPlain = false;
            // Replay the block from the true state. The replay writes its rows to a scratch area and
            // watches the speculative chain's final rows: as soon as it emits a row the speculative chain
            // also ended up with, with the same scan position right after it, the two chains are in the
            // same state for everything above that row (a row that survived to the end of the
            // speculative chain was never looked beneath after it was pushed), so the speculative rows
            // above it are the true continuation and the replay stops.
            Chain ch;
            ch.scan = scanT; ch.minTouched = 0x7fffffff; ch.minKeep = 0x7fffffff; ch.visited = 0; ch.cands = 0; ch.emitted = false;
            Row *own = replayArea + (uint64_t) blockIdx.x * cap;
            const Row *spec = vs.region + (uint64_t) b * cap;
            vs.own = own; vs.ownN = 0;
            int sp = minKeep < 0 ? 0 : minKeep;                 // speculative rows pushed while replaying the own block
            int syncAt = -1;
            auto stop = [&]() -> bool {
                if (!ch.emitted) return false;
                ch.emitted = false;
                if (vs.ownN == 0) return false;
                __builtin_amdgcn_s_waitcnt(0);
                const uint64_t ePos = rfl64(own[vs.ownN - 1].posDest), eLen = rfl64(own[vs.ownN - 1].len), eSrc = rfl64(own[vs.ownN - 1].posSrc);
                while (sp < spF && rfl64(spec[sp].posDest) < ePos) sp++;
                if (sp < spF && rfl64(spec[sp].posDest) == ePos && rfl64(spec[sp].len) == eLen && rfl64(spec[sp].posSrc) == eSrc &&
                    (int32_t) rfl32((uint32_t) spec[sp].scanAfter) == ch.scan) { syncAt = sp; return true; }
                return false;
            };
            chain_run<LAPS>(v, cg, q, w0, w0 + span < npos ? w0 + span : npos, vs, ch, stop);
            int n = vs.ownN;
            vs.size_ -= n; vs.ownN = 0; vs.own = nullptr;
            __builtin_amdgcn_s_waitcnt(0);
            Row *dst = vs.region + (uint64_t) b * cap;
            if (syncAt >= 0) {                                    // append the speculative continuation
                const int tail = spF - syncAt - 1;
                for (int k = lane; k < tail; k += WAVE) own[n + k] = spec[syncAt + 1 + k];
                n += tail;
                scanT = rScanF;
                __builtin_amdgcn_s_waitcnt(0);
            } else
                scanT = ch.scan;
            for (int k = lane; k < n; k += WAVE) dst[k] = own[k];
            __builtin_amdgcn_s_waitcnt(0);
            vs.push_segment((int) b, 0, (uint32_t) n);
            known = 0;
            replayed++;
        }
    }
    // row offsets of the surviving segments (ascending block order = list order)
    vs.flush();
    __builtin_amdgcn_s_waitcnt(0);
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < cg.nrb; b0 += WAVE) {
        const uint32_t b = b0 + lane;
        const uint32_t k = b < cg.nrb ? vs.keepN[b] : 0;
        uint32_t x = k;                                          // inclusive wave scan
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t y = (uint32_t) __shfl_up((int) x, d);
            if ((int) lane >= d) x += y;
        }
        if (b < cg.nrb) dstOff[cg.rb0 + b] = run + x - k;
        run += (uint32_t) __shfl((int) x, WAVE - 1);
    }
    if (lane == 0) {
        matchCount[blockIdx.x] = run;
        atomicAdd(&stats[3], (unsigned long long) replayed);
        atomicAdd(&stats[5], (unsigned long long) nFast);
        atomicAdd(&stats[6], (unsigned long long) nSlow);
        atomicAdd(&stats[7], (unsigned long long) nSkip);
    }
}

// copy every block's surviving rows to their place in the contig's match array
__global__ void __launch_bounds__(WAVE) k_gather(const Contig *__restrict__ contigs, const uint32_t *__restrict__ rbContig,
                                                 const Row *__restrict__ regions, uint32_t cap,
                                                 const uint32_t *__restrict__ segStart, const uint32_t *__restrict__ keepN,
                                                 const uint32_t *__restrict__ dstOff, Match *__restrict__ matches) {
    const uint32_t g = blockIdx.x;
    const uint32_t n = keepN[g];
    if (n == 0) return;
    const Contig cg = contigs[rbContig[g]];
    const Row *src = regions + (uint64_t) g * cap + segStart[g];
    Match *dst = matches + cg.matchBase + dstOff[g];
    for (uint32_t k = threadIdx.x; k < n; k += WAVE) { Match m; m.posSrc = src[k].posSrc; m.len = src[k].len; m.posDest = src[k].posDest; dst[k] = m; }
}

// order-sensitive fingerprint of the whole batch, single thread: test hook only
__global__ void k_fingerprint(const Contig *__restrict__ contigs, int n, const Match *__restrict__ matches,
                              const uint32_t *__restrict__ matchCount, unsigned long long *out) {
    unsigned long long fp = 0xcbf29ce484222325ull, tot = 0, len = 0;
    for (int c = 0; c < n; c++) {
        const Match *m = matches + contigs[c].matchBase;
        for (uint32_t k = 0; k < matchCount[c]; k++) {
            fp ^= m[k].posSrc; fp *= 0x100000001b3ull;
            fp ^= m[k].len; fp *= 0x100000001b3ull;
            fp ^= m[k].posDest; fp *= 0x100000001b3ull;
            tot++; len += m[k].len;
        }
    }
    out[0] = fp; out[1] = tot; out[2] = len;
}

}  // namespace swk
