// Per-contig stream emission on the GPU: MBGC_Encoder::processMatches / extendMatchRight /
// extendMatchLeft (mbgccoder/MBGC_Encoder.cpp:137-427) and ContextAwareMismatchesCoder::mismatch2code
// (coders/ContextAwareMismatchesCoder.cpp:65-70). The reference's loop is split by what is truly
// sequential in it:
//   k_emit_pass1  gap-breaking removal (:153-199). "removed[j]" only depends on removed[j-1] and a
//                 locally computable predicate, so it is resolved per run of that predicate; the kept
//                 rows are compacted and unmatchedChars / totalMatched reduced.      [1024 threads/contig]
//   k_emit_meta_* the pairing ring, gap deltas and gap bookkeeping (:229-278) — a chain over matches
//                 that never looks at sequence bytes. The 64-deep look-ahead of every match is a mask made
//                 lane-parallel from keys in LDS; the chain itself runs one LANE per block of 16 matches,
//                 speculatively, and is stitched with the true state.        [256 blocks/workgroup]
//   k_emit_sizes / k_emit_place / k_emit_write
//                 everything that touches bytes is local to the gap between two consecutive matches:
//                 right extension of the left match, then left extension of the right match, then the
//                 plain literals. One thread per gap sizes its output, a block scan per contig places
//                 every piece, the same automata run again and write.              [one thread/gap]
#include "swsem_device.h"
#include "../../include/mbgc_swsem.h"

namespace swk {

constexpr int EMIT_THREADS = 1024;
constexpr uint8_t MATCH_MARK = 0xA5;                 // MBGC_Params.h:45
constexpr int MAX_EXTEND_MATCH_LEFT_LENGTH = 1 << 24;// MBGC_Params.h:55

struct EMatch {                                      // TextMatch incl. nextSrcRegionLoadingPos scratch
    uint64_t posSrc, len, posDest, lp;               // lp = getMatchLoadedPos(posSrc)
};

struct EmitContig {
    uint64_t qoff, n;                                // contig bytes
    uint64_t matchBase;                              // rows of the match-finding result
    uint64_t lock;
    uint64_t scratchBase;                            // per-match scratch rows (cap rows)
    uint64_t streamBase[SWSEM_NSTREAMS];             // byte offsets into the stream arena
    uint32_t cap;                                    // rows reserved (>= matches + 2)
    uint32_t chunk0;                                 // first of the contig's ceil(cap / CH) chunks in the chunk grid
    int32_t factor;                                  // unmatchedFractionFactor
    uint32_t span0;                                  // first of the contig's ceil(cap / MSPAN) spans in the span grid (pairing chain)
    int64_t processed, targetIdx;                    // processedTargetsCount / targetIdx
};

struct EmitOut {                                     // per contig, read back by the host
    uint64_t size[SWSEM_NSTREAMS];
    uint64_t unmatchedChars;                         // or UINT64_MAX = skipped as dissimilar
    uint64_t extMatched, extMismatches, totalMatched, removed, nmatches;
};

struct LongCopy { uint8_t *dst; const uint8_t *src; uint64_t len; };
constexpr uint32_t LONG_COPY_CAP = 4096;             // entries of the list (a run that finds it full is copied by its wave)
constexpr uint32_t LONG_COPY_MIN = 32768;            // bytes from which a run goes to the list

struct EmitView {
    const uint8_t *ref, *qbuf;
    const Match *matches;
    const uint32_t *matchCount;
    uint64_t pos1, refLength, maxRefLength;
    const uint64_t *loaded; uint32_t nLoaded;       // refExtLoadedPosArr
    swsem_emit_params_t p;
    // scratch (rows indexed by EmitContig::scratchBase + t)
    EMatch *em;                                      // compacted matches
    uint64_t *next0;                                 // upper_bound(loaded, lp)
    uint8_t *removed;
    uint32_t *keepIdx;
    uint32_t *meta;                                  // per kept match, see META_*
    uint32_t *corr;                                  // gapStartIdx when in a gap
    unsigned long long *pairMask;                    // per kept match: which of the next 64 it can be paired with (k_emit_meta_masks)
    unsigned long long *litBits;                     // per 64 kept matches (word chunk0 * 4 + m / 64): no literal follows the match
    unsigned long long *metaBad;                     // per 64 blocks of the pairing chain (word chunk0 + b / 64): not acceptable as they are
    uint32_t *sz;                                    // 6 u32 per gap task
    uint32_t *ofs;                                   // 6 u32 per iteration: start offsets in the six streams
    uint32_t *chunkCnt;                              // [chunk] kept-match counts / offsets, 6 sums per chunk for placement
    const uint32_t *chunkOwner;                      // [chunk] -> contig of the batch (chunks of CH rows, per contig)
    const uint32_t *spanOwner;                       // [span] -> contig of the batch (spans of the pairing chain)
    uint32_t ncontigs;
    uint64_t *packBase;                              // [contig][stream] start of the stream in the packed arena
    uint8_t *arena;                                  // streams
    struct LongCopy *longCopies;                     // runs of plain literals too long for one wave (k_emit_write lists, k_emit_copy_long copies)
    uint32_t *longCount;
    EmitOut *out;
};

constexpr uint32_t META_SKIPOFF = 1, META_HASGAP = 2, META_ISGAP = 4, META_GSTART = 8, META_GMID = 16, META_GEND = 32;
// gap delta byte in bits 8..15

__device__ __forceinline__ bool paired(uint64_t aSrc, uint64_t aDst, uint64_t bSrc, uint64_t bDst) {   // TextMatchers.h:42-44
    return aSrc + bDst == bSrc + aDst;
}

// coders/ContextAwareMismatchesCoder.h:13-17, .cpp:65-70 (bytes >= 0x80 index the reference's table out
// of range; they are treated as outside ACGTN, like the oracle does)
__device__ __forceinline__ int sym5(uint8_t c) {
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : -1;
}
__device__ __forceinline__ uint8_t mismatch2code(uint8_t actual, uint8_t mismatch) {
    const int a = sym5(actual), b = sym5(mismatch);
    if (actual == mismatch || a < 0 || b < 0) return mismatch;
    // rows A,C,G,T,N of mis2code packed 3 bits per entry (diagonal unused)
    const uint32_t rows[5] = {0 | 2u << 3 | 0u << 6 | 1u << 9 | 3u << 12, 1 | 0u << 3 | 2u << 6 | 0u << 9 | 3u << 12,
                              0 | 2u << 3 | 0u << 6 | 1u << 9 | 3u << 12, 1 | 0u << 3 | 2u << 6 | 0u << 9 | 3u << 12,
                              1 | 2u << 3 | 3u << 6 | 0u << 9 | 0u << 12};
    return (uint8_t) ((rows[a] >> (3 * b)) & 7u);
}

// getMatchLoadedPos, MBGC_Encoder.cpp:137-141
__device__ __forceinline__ uint64_t loaded_pos(const EmitView &v, uint64_t pos) {
    const uint64_t span = v.refLength - 1;
    const uint64_t back = v.loaded[v.nLoaded - 1];
    while (pos + span < back) pos += span;
    return pos;
}

// block-wide exclusive scan of one u32 per thread (NT threads); returns the exclusive prefix,
// *total = sum over the block. lds: NT / WAVE + 1 words.
template <int NT>
__device__ uint32_t block_scan(uint32_t x, uint32_t *lds, uint32_t *total) {
    const uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    uint32_t inc = x;
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t y = (uint32_t) __shfl_up((int) inc, d);
        if ((int) lane >= d) inc += y;
    }
    __syncthreads();
    if (lane == WAVE - 1) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < NT / WAVE; i++) { const uint32_t t = lds[i]; lds[i] = run; run += t; }
        lds[NT / WAVE] = run;
    }
    __syncthreads();
    *total = lds[NT / WAVE];
    return lds[w] + inc - x;
}

// ------------------------------------------------------------------------------------------------
// pass 1, MBGC_Encoder.cpp:153-205. Grid kernels: one block per chunk of CH rows; chunkOwner maps the
// block to its contig (gk) and gx is the chunk index inside the contig, so ragged batches launch no empty blocks.
// ------------------------------------------------------------------------------------------------
constexpr int CH = 256;

// (a) the locally computable part of the removal test (:176-178), taking j-1 as the kept predecessor
__device__ __forceinline__ bool p1_flag(const EmitView &v, const Match *__restrict__ M, int64_t n, int64_t j) {
    if (!v.p.enableExtensionsWithMismatches || j < 1 || j + 1 >= n) return false;
    const Match a = M[j - 1], b = M[j], c = M[j + 1];
    return paired(c.posSrc, c.posDest, a.posSrc, a.posDest) && !paired(b.posSrc, b.posDest, a.posSrc, a.posDest) &&
           b.len < v.p.gapBreakingMatchMinLength;
}

// (b) a removed match keeps its successor (the successor is then paired with the kept predecessor), so
// inside a run of the predicate the matches are removed alternately, starting with the first. Also
// counts the kept matches of the chunk.
__global__ void __launch_bounds__(CH) k_emit_p1_removed(EmitView v, const EmitContig *__restrict__ cgs, const int *__restrict__ which) {
    __shared__ uint32_t cnt;
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const int64_t n = v.matchCount[which[gk]];
    if ((int64_t) gx * CH >= n) { if (threadIdx.x == 0) v.chunkCnt[(size_t) blockIdx.x] = 0; return; }
    const int64_t j = (int64_t) gx * CH + threadIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const Match *M = v.matches + cg.matchBase;
    bool keep = false;
    if (j < n) {
        bool r = false;
        if (p1_flag(v, M, n, j)) {
            int64_t k = 0;
            while (j - 1 - k >= 0 && p1_flag(v, M, n, j - 1 - k)) k++;
            r = (k & 1) == 0;
        }
        v.removed[cg.scratchBase + j] = r;
        keep = !r;
    }
    const unsigned long long bal = __ballot(keep);
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(&cnt, (uint32_t) __popcll(bal));
    __syncthreads();
    if (threadIdx.x == 0) v.chunkCnt[(size_t) blockIdx.x] = cnt;
}

// (c) per contig: exclusive scan of the chunk counts; resets the per-contig accumulators
__global__ void __launch_bounds__(CH) k_emit_p1_scan(EmitView v, const EmitContig *__restrict__ cgs, const int *__restrict__ which) {
    __shared__ uint32_t lds[CH / WAVE + 2];
    const int64_t n = v.matchCount[which[blockIdx.x]];
    const uint32_t nch = (uint32_t) ((n + CH - 1) / CH);
    uint32_t *cc = v.chunkCnt + cgs[blockIdx.x].chunk0;
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < nch; c0 += CH) {
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t x = c < nch ? cc[c] : 0;
        uint32_t tot;
        const uint32_t ex = block_scan<CH>(x, lds, &tot);
        if (c < nch) cc[c] = base + ex;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        EmitOut o;
        for (int s = 0; s < SWSEM_NSTREAMS; s++) o.size[s] = 0;
        o.unmatchedChars = 0; o.extMatched = 0; o.extMismatches = 0; o.totalMatched = 0;
        o.removed = (uint64_t) (n - base);
        o.nmatches = base;
        v.out[blockIdx.x] = o;
    }
}

// (d) compaction; an abutting successor of a removed match is extended to the left (:180-186). Also (e): unmatchedChars /
// totalMatched with the reference's integer types (uint32 pos, :145,:193-196) as per-chunk partial sums — what lies between a
// kept match and the kept match in front of it, which is its predecessor or, behind a removed one, the match before that (two
// removed matches never follow each other; a left extension moves a match's start, not its end) — two u64 at the chunk's slot
// of the size scratch, which nothing else uses before the second phase.
__global__ void __launch_bounds__(CH) k_emit_p1_compact(EmitView v, const EmitContig *__restrict__ cgs, const int *__restrict__ which) {
    __shared__ uint32_t lds[CH / WAVE + 2];
    __shared__ unsigned long long part[2 * (CH / WAVE)];
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const int64_t n = v.matchCount[which[gk]];
    if ((int64_t) gx * CH >= n) return;
    const int64_t j = (int64_t) gx * CH + threadIdx.x;
    const Match *M = v.matches + cg.matchBase;
    const uint8_t *q = v.qbuf + cg.qoff, *rm = v.removed + cg.scratchBase;
    const uint32_t keep = (j < n && !rm[j]) ? 1u : 0u;
    uint32_t tot;
    const uint32_t ex = block_scan<CH>(keep, lds, &tot);
    unsigned long long um = 0, tm = 0;
    if (keep) {
        const uint32_t t = v.chunkCnt[(size_t) blockIdx.x] + ex;
        EMatch e;
        e.posSrc = M[j].posSrc; e.len = M[j].len; e.posDest = M[j].posDest;
        const bool behindRemoved = j >= 1 && rm[j - 1];
        if (behindRemoved && M[j - 1].posDest + M[j - 1].len == M[j].posDest) {
            int64_t s = (int64_t) e.posSrc, d = (int64_t) e.posDest;
            uint64_t x = 0;
            // (eight bytes a step where there are eight: a dependent load per byte of a run of dozens is what this kernel's
            // slowest threads would be made of)
            while (d >= 8 && s >= 8) {
                uint64_t a, b;
                __builtin_memcpy(&a, q + d - 8, 8); __builtin_memcpy(&b, v.ref + s - 8, 8);
                if (a != b) break;
                d -= 8; s -= 8; x += 8;
            }
            while (d - 1 >= 0 && s - 1 >= 0 && q[d - 1] == v.ref[s - 1]) { d--; s--; x++; }
            e.posSrc -= x; e.posDest -= x; e.len += x;               // shiftStartPos(-leftExtension)
        }
        e.lp = 0;                                                     // (k_emit_meta_regions: nothing in this pass asks for it)
        v.em[cg.scratchBase + t] = e;
        v.keepIdx[cg.scratchBase + t] = (uint32_t) j;
        const int64_t jp = behindRemoved ? j - 2 : j - 1;            // the kept match in front
        const uint32_t pos = jp >= 0 ? (uint32_t) (M[jp].posDest + M[jp].len) : 0u;
        um = e.posDest - (uint64_t) pos;
        tm = (uint32_t) e.len;
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        um += (unsigned long long) __shfl_down((long long) um, d);
        tm += (unsigned long long) __shfl_down((long long) tm, d);
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) { part[2 * (threadIdx.x / WAVE)] = um; part[2 * (threadIdx.x / WAVE) + 1] = tm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < CH / WAVE; w++) { a += part[2 * w]; b += part[2 * w + 1]; }
        unsigned long long *o = (unsigned long long *) (v.sz + (size_t) blockIdx.x * 6);
        o[0] = a; o[1] = b;
    }
}

// (f) the chunks' partial sums per contig (one block each), the tail literal and the dissimilarity early-out (:200-205,
// isContigDissimilar MGMP_Params.h:193-196)
__global__ void __launch_bounds__(CH) k_emit_p1_finish(EmitView v, const EmitContig *__restrict__ cgs) {
    __shared__ unsigned long long part[2 * (CH / WAVE)];
    const int k = blockIdx.x;
    const EmitContig cg = cgs[k];
    EmitOut *o = v.out + k;
    const uint32_t nk = (uint32_t) o->nmatches;
    const uint64_t nAll = o->nmatches + o->removed;                  // the chunks k_emit_p1_compact ran over: matches before the removal
    const uint32_t nch = (uint32_t) ((nAll + CH - 1) / CH);
    unsigned long long um = 0, tm = 0;
    for (uint32_t c = threadIdx.x; c < nch; c += CH) {
        const unsigned long long *p = (const unsigned long long *) (v.sz + ((size_t) cg.chunk0 + c) * 6);
        um += p[0]; tm += p[1];
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        um += (unsigned long long) __shfl_down((long long) um, d);
        tm += (unsigned long long) __shfl_down((long long) tm, d);
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) { part[2 * (threadIdx.x / WAVE)] = um; part[2 * (threadIdx.x / WAVE) + 1] = tm; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    um = 0; tm = 0;
    for (int w = 0; w < CH / WAVE; w++) { um += part[2 * w]; tm += part[2 * w + 1]; }
    const EMatch *E = v.em + cg.scratchBase;
    const uint32_t pos = nk ? (uint32_t) (E[nk - 1].posDest + E[nk - 1].len) : 0u;
    const int64_t unmatched = (int64_t) (um + (cg.n - (uint64_t) pos));
    o->totalMatched = (uint32_t) tm;
    o->unmatchedChars = (uint64_t) unmatched;
    if (cg.processed < cg.targetIdx - v.p.allowedTargetsOutrunForDissimilarContigs &&
        cg.n > v.p.minimalLengthForDissimilarContigs &&
        (uint64_t) (unmatched * (int64_t) (cg.factor / v.p.unmatchedFractionFactorTweakForDissimilarContigs)) > cg.n)
        o->unmatchedChars = UINT64_MAX;
}

// ------------------------------------------------------------------------------------------------
// the pairing / gap chain, MBGC_Encoder.cpp:229-278 (no sequence bytes involved). The chain's state at
// match j is small — which of the next 64 matches are already paired, the open gap, and the inherited
// nextSrcRegionLoadingPos values of the paired ones — and claims reach at most 64 matches ahead, so
// chains started from an empty state fall into step with the true one after a short warm-up.
//   k_emit_meta_spec    every LANE runs the chain of its own block of MB matches after MWARM warm-up matches
//                       (speculation), 256 blocks per workgroup: the chain is scalar work, and a wave that runs
//                       one chain issues one instruction in four cycles whatever it does — 64 chains per wave
//                       cost the same. The look-ahead masks ("which of my next 64 matches are pairedWith me")
//                       of the workgroup's span are computed first, lane-parallel, from keys staged in LDS.
//   k_emit_meta_stitch  accepts a block when the state it started from equals the state its predecessor
//                       ended in (the first block of a contig starts from the true, empty state) and replays
//                       the group of 64 matches around a block that fails, from the true state, with the
//                       wave-wide form of the chain (meta_run) — identical results by construction.
// ------------------------------------------------------------------------------------------------
constexpr int META_GROUP = WAVE;                     // matches per replay of the stitch (one chunk of meta_run)
constexpr int META_LDS = META_GROUP + WAVE;          // staged matches of a replay: the group + its look-ahead
constexpr int MB = 16, MWARM = 64, MLANES = 256;     // matches per speculative block, warm-up, blocks per workgroup
constexpr int MSPAN = MLANES * MB;                   // matches per workgroup
constexpr int MNX = 4;                               // inherited boundaries a lane keeps (more: the block is replayed)
static_assert(MSPAN % CH == 0 && META_GROUP % MB == 0, "spans are whole chunks of the emission grid, groups whole blocks");
constexpr int MCHUNKS = MSPAN / CH;                  // chunks of the emission grid per span
constexpr int MPAD_N = MSPAN + MWARM + WAVE;         // staged matches: warm-up + span + look-ahead
// one pad word per 16: lanes 16 matches apart (the chains) and lanes one match apart (staging) both spread over the banks
__device__ __forceinline__ int mpad(int i) { return i + (i >> 4); }
constexpr int MPAD_SIZE = MPAD_N + (MPAD_N >> 4) + 2;

// The chain's state at a block boundary J, in canonical form (entries sorted by rel, empty ones zeroed): two states are
// equal iff every field is.
struct MetaRec {
    unsigned long long claimed;      // bit g-1: match J+g is already paired (the pairedGap ring)
    int64_t gapStartIdx, gapEndIdx;
    uint32_t flags;                  // bit 0: pairedGap[gapCurIdx] (match J is paired); bit 1: more than MNX inherited values
    uint8_t rel[MNX];                // match J + rel inherited a nextSrcRegionLoadingPos that is not its own (0xFF: no entry)
    uint64_t val[MNX];
};
static_assert(sizeof(MetaRec) == 64, "one record, one 64-byte line");
constexpr uint32_t MREC_CUR = 1, MREC_OVERFLOW = 2;

struct MetaLds {
    int64_t sdiag[META_LDS];
    uint64_t ssrc[META_LDS], slp[META_LDS], snx0[META_LDS];
    uint32_t slen[META_LDS];
};

// The chain's state between two matches. The inherited nextSrcRegionLoadingPos values of the next 65
// matches live in two vector registers: lane k of nxA belongs to match (chunk start + k), lane k of nxB
// to match (chunk start + 64 + k).
struct MetaRun {
    unsigned long long claimed;
    bool curClaimed;
    int64_t gapStartIdx, gapEndIdx;
    uint64_t nxA, nxB;
};

__device__ __forceinline__ uint64_t rl64(uint64_t x, int lane) {
    return ((uint64_t) rl32((uint32_t) (x >> 32), lane) << 32) | rl32((uint32_t) x, lane);
}
// write a wave-uniform value into one lane of a vector register
__device__ __forceinline__ uint32_t wl32(uint32_t val, int lane, uint32_t vec) {
    return (int) (threadIdx.x & (WAVE - 1)) == lane ? val : vec;
}
__device__ __forceinline__ uint64_t wl64(uint64_t val, int lane, uint64_t vec) {
    return (int) (threadIdx.x & (WAVE - 1)) == lane ? val : vec;
}

// stage matches [from, to) of the compacted list
__device__ __forceinline__ void meta_load(const EmitView &v, const EmitContig &cg, MetaLds &L, int64_t from, int64_t to, bool lazy) {
    const EMatch *E = v.em + cg.scratchBase;
    for (int64_t t = from + threadIdx.x; t < to; t += WAVE) {
        const EMatch e = E[t];
        const int64_t k = t - from;
        L.sdiag[k] = (int64_t) (e.posSrc - e.posDest);
        L.ssrc[k] = e.posSrc;
        L.slp[k] = e.lp;
        L.slen[k] = (uint32_t) e.len;
        L.snx0[k] = lazy ? v.next0[cg.scratchBase + t] : 0;
    }
    __builtin_amdgcn_s_waitcnt(0);
}

// Matches [j0, j1) of the chain, j0 a multiple of 64; LDS holds matches [base, ...) covering j1 + 64.
// Per chunk of 64 matches the lanes first compute everything that does not depend on the chain's state
// (lane k works for match chunk+k: which of its next 64 matches are pairedWith it, which the lazy rule
// rejects with its own region boundary, whether no literal follows it); the chain itself then runs on the
// scalar unit, one match after the other, reading those per-match words with v_readlane.
template <bool WRITE>
__device__ void meta_run(const EmitView &v, const EmitContig &cg, MetaLds &L, MetaRun &st, int64_t base, int64_t j0, int64_t j1, int64_t n) {
    const int lane = (int) threadIdx.x;
    const bool lazy = v.p.lazyDecompressionSupport != 0, ext = v.p.enableExtensionsWithMismatches != 0;
    const int depth = v.p.gapDepthOffsetEncoding;
    for (int64_t jb = j0; jb < j1; jb += WAVE) {
        const int cnt = (int) (j1 - jb < WAVE ? j1 - jb : WAVE);
        // ---- lane-parallel part
        const int64_t jl = jb + lane, lk = jl - base;
        unsigned long long pm = 0, lm = 0;
        uint64_t own = 0;
        bool lit0 = false;
        if (lane < cnt) {
            const int gCnt = (int) (n - jl - 1 < depth ? n - jl - 1 : depth);
            const int64_t dj = L.sdiag[lk];
            const uint64_t sj = L.ssrc[lk];
            own = L.snx0[lk];
            const bool above = sj > cg.lock, below = sj < cg.lock;
            // branch-free and unrolled, so the LDS reads of several look-ahead steps are in flight together
#pragma unroll 8
            for (int g = 1; g <= WAVE; g++) {
                const int64_t x = g <= gCnt ? lk + g : lk;
                const int64_t dg = L.sdiag[x];
                const uint64_t sg = L.ssrc[x], lg = L.slp[x];
                const bool in = g <= gCnt;
                const bool pw = in & (dg == dj) & ((above & (sg > cg.lock)) | (below & (sg < cg.lock)));    // TextMatchers.h:46-50
                pm |= (unsigned long long) pw << (g - 1);
                lm |= (unsigned long long) (in & lazy & (lg >= own)) << (g - 1);         // :258-259 with the match's own boundary
            }
            const uint64_t endj = (uint64_t) ((int64_t) sj - dj) + L.slen[lk];
            const uint64_t nxt = jl + 1 < n ? (uint64_t) ((int64_t) L.ssrc[lk + 1] - L.sdiag[lk + 1]) : cg.n;
            lit0 = nxt - (uint32_t) endj == 0;                                           // :242 (uint32 pos)
        }
        const unsigned long long litZero = __ballot(lit0);
        uint32_t metaV = 0, corrV = 0;
        // ---- the chain (wave-uniform)
        for (int k = 0; k < cnt; k++) {
            const int64_t j = jb + k;
            if (ext && j == st.gapEndIdx) { st.gapStartIdx = -1; st.gapEndIdx = -1; }    // :222-225
            const bool skipOffset = st.curClaimed;                                       // :229
            const int gCnt = (int) (n - j - 1 < depth ? n - j - 1 : depth);
            const bool rule = !lazy && st.gapStartIdx == -1 && ((litZero >> k) & 1);      // :247, applies to g == 1
            const unsigned long long taken = st.claimed | (rule ? 1ull : 0ull);
            unsigned long long em = rl64(pm, k) & ~taken;                                 // paired and not yet claimed
            uint64_t nextj = 0;
            if (lazy && em) {
                nextj = rl64(st.nxA, k);
                const uint64_t mine = rl64(own, k);
                if (!nextj || nextj == mine) { nextj = mine; em &= ~rl64(lm, k); }         // :253-259
                else {                                                                     // inherited boundary: evaluate on the lanes
                    const int g = lane + 1;
                    const bool rej = g <= gCnt && L.slp[j - base + g] >= nextj;
                    em &= ~__ballot(rej);
                }
            }
            uint32_t gapByte = 0;
            if (em) {
                const int gf = __builtin_ctzll(em) + 1;
                const unsigned long long below = gf > 1 ? ((1ull << (gf - 1)) - 1) : 0ull;
                gapByte = (uint32_t) (gf - __popcll(taken & below));
                st.claimed |= 1ull << (gf - 1);                                             // :262
                if (lazy) {                                                                 // :260
                    const int t = k + gf;
                    if (t < WAVE) st.nxA = wl64(nextj, t, st.nxA);
                    else st.nxB = wl64(nextj, t - WAVE, st.nxB);
                }
                if (ext && st.gapEndIdx <= j + gf && gf <= v.p.gapDepthMismatchesEncoding) { st.gapStartIdx = j; st.gapEndIdx = j + gf; }
            }
            st.curClaimed = st.claimed & 1ull;                                              // :272-273: advance the ring
            st.claimed >>= 1;
            if (WRITE) {
                const bool gs = st.gapStartIdx == j, ge = st.gapEndIdx == j + 1, gm = st.gapStartIdx < j && j + 1 < st.gapEndIdx;
                const bool isGap = gs || gm || ge;
                const uint32_t mw = (skipOffset ? META_SKIPOFF : 0) | (gCnt ? META_HASGAP : 0) | (isGap ? META_ISGAP : 0) |
                                    (gs ? META_GSTART : 0) | (gm ? META_GMID : 0) | (ge ? META_GEND : 0) | (gapByte << 8);
                metaV = wl32(mw, k, metaV);
                corrV = wl32(isGap ? (uint32_t) st.gapStartIdx : (uint32_t) j, k, corrV);
            }
        }
        if (WRITE && lane < cnt) {
            v.meta[cg.scratchBase + jl] = metaV;
            v.corr[cg.scratchBase + jl] = corrV;
        }
        // the next chunk: its inherited values were collected in nxB
        st.nxA = st.nxB;
        st.nxB = 0;
    }
}


__device__ __forceinline__ bool mrec_equal(const MetaRec &a, const MetaRec &b) {
    uint32_t ra, rb;
    __builtin_memcpy(&ra, a.rel, 4); __builtin_memcpy(&rb, b.rel, 4);
    uint64_t d = (a.claimed ^ b.claimed) | (uint64_t) (a.gapStartIdx ^ b.gapStartIdx) | (uint64_t) (a.gapEndIdx ^ b.gapEndIdx) |
                 (uint64_t) (a.flags ^ b.flags) | (uint64_t) (ra ^ rb);
#pragma unroll
    for (int e = 0; e < MNX; e++) d |= a.val[e] ^ b.val[e];
    return d == 0;
}

// a lane's chain state -> canonical record relative to boundary J
__device__ __forceinline__ void mrec_store(MetaRec *dst, unsigned long long claimed, bool cur, int gs, int ge, const int (&nxPos)[MNX],
                                           const uint64_t (&nxVal)[MNX], bool overflow, int J) {
    uint32_t r[MNX]; uint64_t x[MNX];
#pragma unroll
    for (int e = 0; e < MNX; e++) { r[e] = nxPos[e] >= J ? (uint32_t) (nxPos[e] - J) : 0xFFu; x[e] = nxPos[e] >= J ? nxVal[e] : 0; }
    auto cx = [&](int a, int b) {
        if (r[a] > r[b]) { const uint32_t t = r[a]; r[a] = r[b]; r[b] = t; const uint64_t u = x[a]; x[a] = x[b]; x[b] = u; }
    };
    cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
    MetaRec R;
    R.claimed = claimed; R.gapStartIdx = gs; R.gapEndIdx = ge;
    R.flags = (cur ? MREC_CUR : 0u) | (overflow ? MREC_OVERFLOW : 0u);
#pragma unroll
    for (int e = 0; e < MNX; e++) { R.rel[e] = (uint8_t) r[e]; R.val[e] = x[e]; }
    *dst = R;
}

// Span = MSPAN matches of one contig, one workgroup each: the span grid (spanOwner, EmitContig::span0) is laid out like the
// chunk grid, for the rows a contig has reserved — a span beyond the matches it really has ends at once.
struct MetaSpan { uint32_t gk; int n, B0; bool live; };
__device__ __forceinline__ MetaSpan meta_span(const EmitView &v, const EmitContig *__restrict__ cgs) {
    MetaSpan sp;
    sp.gk = v.spanOwner[blockIdx.x];
    const EmitOut o = v.out[sp.gk];
    sp.n = (int) o.nmatches;
    sp.B0 = (int) (blockIdx.x - cgs[sp.gk].span0) * MSPAN;
    sp.live = o.unmatchedChars != UINT64_MAX && sp.B0 < sp.n;
    return sp;
}

// the source region of every kept match: getMatchLoadedPos (:137-141) and the loading position of the region behind it
// (std::upper_bound, :254-256) — what the pairing chain's lazy-decompression rule compares. Off pass 1's path: the
// extension policy does not wait for a binary search per match.
__global__ void __launch_bounds__(CH) k_emit_meta_regions(EmitView v, const EmitContig *__restrict__ cgs) {
    if (!v.p.lazyDecompressionSupport) return;
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const EmitOut o = v.out[gk];
    if (o.unmatchedChars == UINT64_MAX) return;
    const uint64_t t = (uint64_t) gx * CH + threadIdx.x;
    if (t >= o.nmatches) return;
    const uint64_t lp = loaded_pos(v, v.em[cg.scratchBase + t].posSrc);
    v.em[cg.scratchBase + t].lp = lp;
    uint32_t lo = 0, hi = v.nLoaded;
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (v.loaded[mid] <= lp) lo = mid + 1; else hi = mid; }
    v.next0[cg.scratchBase + t] = lo == v.nLoaded ? UINT64_MAX : v.loaded[lo];
}

// look-ahead masks of a span's matches: bit g-1 of match m = match m+g is within the depth, pairedWith m
// (TextMatchers.h:42-50: same diagonal, same side of the lock — two matches of one contig on one diagonal cannot both
// start AT the lock, so equal keys say it all) and, under lazy decompression, not beyond m's OWN region boundary
// (:253-259 when nothing else was inherited). Plus one bit per match: no literal follows it (:242).
__global__ void __launch_bounds__(MLANES) k_emit_meta_masks(EmitView v, const EmitContig *__restrict__ cgs) {
    __shared__ uint64_t skey[MPAD_SIZE], slp[MPAD_SIZE];
    const MetaSpan sp = meta_span(v, cgs);
    if (!sp.live) return;
    const EmitContig cg = cgs[sp.gk];
    const int n = sp.n, B0 = sp.B0, tid = (int) threadIdx.x;
    const int P1 = B0 + MSPAN < n ? B0 + MSPAN : n;                      // masks for [B0, P1)
    const int PK = P1 + WAVE < n ? P1 + WAVE : n;                        // from the keys of [B0, PK)
    const EMatch *E = v.em + cg.scratchBase;
    const bool lazy = v.p.lazyDecompressionSupport != 0;
    const int depth = v.p.gapDepthOffsetEncoding;
    for (int i = tid; i < ((PK - B0 + WAVE - 1) & ~(WAVE - 1)); i += MLANES) {
        const int m = B0 + i;
        bool lit0 = false;
        if (m < PK) {
            const EMatch e = E[m];
            const uint64_t side = e.posSrc > cg.lock ? 1u : (e.posSrc < cg.lock ? 2u : 0u);
            skey[mpad(i)] = ((e.posSrc - e.posDest) << 2) | side;
            slp[mpad(i)] = e.lp;
            const uint64_t nxt = m + 1 < n ? E[m + 1].posDest : cg.n;
            lit0 = nxt - (uint32_t) (e.posDest + e.len) == 0;                       // (uint32 pos)
        }
        const unsigned long long lb = __ballot(lit0);
        if ((tid & (WAVE - 1)) == 0 && m < P1) v.litBits[(size_t) cg.chunk0 * (CH / WAVE) + (size_t) (m / WAVE)] = lb;
    }
    __syncthreads();
    for (int i = tid; i < P1 - B0; i += MLANES) {
        const int m = B0 + i;
        const int gCnt = n - m - 1 < depth ? n - m - 1 : depth;
        const uint64_t kj = skey[mpad(i)];
        const uint64_t own = lazy ? v.next0[cg.scratchBase + m] : UINT64_MAX;
        unsigned long long pm = 0;
#pragma unroll 8
        for (int g = 1; g <= WAVE; g++) {                    // branch-free and unrolled: the LDS reads of several steps are in flight together
            const int x = mpad(g <= gCnt ? i + g : i);
            const bool ok = (g <= gCnt) & (skey[x] == kj) & ((slp[x] < own) | !lazy);
            pm |= (unsigned long long) ok << (g - 1);
        }
        v.pairMask[cg.scratchBase + m] = pm;
    }
}

// the chains: lane = block [j0, j1) of a span, warmed up on the MWARM matches in front of it
__global__ void __launch_bounds__(MLANES) k_emit_meta_spec(EmitView v, const EmitContig *__restrict__ cgs, MetaRec *__restrict__ recs, int warm,
                                                          unsigned long long *__restrict__ stats) {
    __shared__ uint64_t spm[MPAD_SIZE], sown[MPAD_SIZE];     // masks; the matches' own region boundaries
    __shared__ unsigned long long slit[MPAD_N / WAVE + 2];
    const MetaSpan sp = meta_span(v, cgs);
    if (!sp.live) return;
    const EmitContig cg = cgs[sp.gk];
    const int n = sp.n, B0 = sp.B0, tid = (int) threadIdx.x;
    const int P0 = B0 >= MWARM ? B0 - MWARM : 0;                         // first staged match
    const int P1 = B0 + MSPAN < n ? B0 + MSPAN : n;
    const EMatch *E = v.em + cg.scratchBase;
    const bool lazy = v.p.lazyDecompressionSupport != 0, ext = v.p.enableExtensionsWithMismatches != 0;
    const int depth = v.p.gapDepthOffsetEncoding, depthMism = v.p.gapDepthMismatchesEncoding;
    const int PK = P1 + WAVE < n ? P1 + WAVE : n;                        // own boundaries for [P0, PK): a claim looks at its target's
    for (int i = tid; i < PK - P0; i += MLANES) {
        if (P0 + i < P1) spm[mpad(i)] = v.pairMask[cg.scratchBase + P0 + i];
        sown[mpad(i)] = lazy ? v.next0[cg.scratchBase + P0 + i] : 0;
    }
    if (tid < MPAD_N / WAVE + 2) {
        const int w = P0 / WAVE + tid;                                    // (P0 is a multiple of 64)
        slit[tid] = w * WAVE < P1 ? v.litBits[(size_t) cg.chunk0 * (CH / WAVE) + (size_t) w] : 0;
    }
    __syncthreads();
    const int j0 = B0 + tid * MB;
    const bool active = j0 < n;
    const int j1 = j0 + MB < n ? j0 + MB : n;
    unsigned long long claimed = 0;
    bool cur = false, overflow = false;
    int gs = -1, ge = -1;
    int nxPos[MNX]; uint64_t nxVal[MNX];
#pragma unroll
    for (int e = 0; e < MNX; e++) { nxPos[e] = -1; nxVal[e] = 0; }
    MetaRec *myRecs = recs + ((size_t) cg.chunk0 * (CH / MB) + (size_t) (j0 / MB)) * 2;
    for (int s = 0; s < MWARM + MB; s++) {
        const int j = j0 - MWARM + s;
        if (s == MWARM && active) mrec_store(myRecs, claimed, cur, gs, ge, nxPos, nxVal, overflow, j0);
        if (!active || j < j0 - warm || j < 0 || j >= j1) continue;         // (warm < MWARM: A/B switch and the tests' way to make blocks fail)
        const int i = j - P0;
        if (ext && j == ge) { gs = -1; ge = -1; }                                           // :222-225
        const bool skipOffset = cur;                                                        // :229
        const int gCnt = n - j - 1 < depth ? n - j - 1 : depth;
        const bool rule = !lazy && gs == -1 && ((slit[i / WAVE] >> (i & (WAVE - 1))) & 1);  // :247, applies to g == 1
        const unsigned long long taken = claimed | (rule ? 1ull : 0ull);
        unsigned long long em = spm[mpad(i)] & ~taken;                                      // paired (within the own boundary) and not yet claimed
        uint64_t inherited = 0;
#pragma unroll
        for (int e = 0; e < MNX; e++) if (nxPos[e] == j) { inherited = nxVal[e]; nxPos[e] = -1; }
        uint64_t nextj = 0;
        if (lazy) {                                                                         // :253-259
            nextj = sown[mpad(i)];
            if (inherited && inherited != nextj) {
                // an inherited boundary that is not the match's own (a pair across two source regions): the mask does not
                // hold for it — from the matches themselves
                nextj = inherited;
                atomicAdd(&stats[4], 1ull);
                const EMatch mj = E[j];
                em = 0;
                for (int g = 1; g <= gCnt; g++) {
                    const EMatch c = E[j + g];
                    const bool pw = paired(mj.posSrc, mj.posDest, c.posSrc, c.posDest) &&
                                    ((mj.posSrc > cg.lock && c.posSrc > cg.lock) || (mj.posSrc < cg.lock && c.posSrc < cg.lock));
                    if (pw && c.lp < nextj) em |= 1ull << (g - 1);
                }
                em &= ~taken;
            }
        }
        uint32_t gapByte = 0;
        if (em) {
            const int gf = __builtin_ctzll(em) + 1;
            const unsigned long long below = gf > 1 ? ((1ull << (gf - 1)) - 1) : 0ull;
            gapByte = (uint32_t) (gf - __popcll(taken & below));
            claimed |= 1ull << (gf - 1);                                                    // :262
            if (lazy && nextj != sown[mpad(i + gf)]) {                                      // :260 (its own boundary: nothing to carry, :254)
                bool put = false;
#pragma unroll
                for (int e = 0; e < MNX; e++) if (!put && nxPos[e] < 0) { nxPos[e] = j + gf; nxVal[e] = nextj; put = true; }
                overflow |= !put;
            }
            if (ext && ge <= j + gf && gf <= depthMism) { gs = j; ge = j + gf; }
        }
        cur = claimed & 1ull;                                                               // :272-273: advance the ring
        claimed >>= 1;
        if (j >= j0) {
            const bool s0 = gs == j, e0 = ge == j + 1, m0 = gs < j && j + 1 < ge;
            const bool isGap = s0 || m0 || e0;
            const uint32_t mw = (skipOffset ? META_SKIPOFF : 0) | (gCnt ? META_HASGAP : 0) | (isGap ? META_ISGAP : 0) |
                                (s0 ? META_GSTART : 0) | (m0 ? META_GMID : 0) | (e0 ? META_GEND : 0) | (gapByte << 8);
            v.meta[cg.scratchBase + j] = mw;
            v.corr[cg.scratchBase + j] = (uint32_t) (isGap ? gs : j);
        }
    }
    if (active) mrec_store(myRecs + 1, claimed, cur, gs, ge, nxPos, nxVal, overflow, j1);
    if (active && overflow) atomicAdd(&stats[7], 1ull);
}

// which blocks cannot be accepted as they are: one bit per block, one word per 64 blocks of a contig (word chunk0 + b / 64).
// A block is held against its predecessor's end; the first block of a contig started from the true, empty state.
__global__ void __launch_bounds__(WAVE) k_emit_meta_check(EmitView v, const EmitContig *__restrict__ cgs, const MetaRec *__restrict__ recs,
                                                          unsigned long long *__restrict__ stats) {
    constexpr int CPW = WAVE * MB / CH;                              // chunks of the emission grid per word of blocks
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    if (gx % CPW) return;
    const EmitOut o = v.out[gk];
    if (o.unmatchedChars == UINT64_MAX) return;
    const int64_t n = (int64_t) o.nmatches, nblk = (n + MB - 1) / MB;
    const int64_t b = (int64_t) (gx / CPW) * WAVE + threadIdx.x;
    if (b - (int64_t) threadIdx.x >= nblk) return;
    const MetaRec *R0 = recs + (size_t) cg.chunk0 * (CH / MB) * 2;
    bool bad = false;
    if (b < nblk) {
        const MetaRec S = R0[2 * b], F = R0[2 * b + 1];
        bad = ((S.flags | F.flags) & MREC_OVERFLOW) != 0;
        if (b > 0 && !bad) bad = !mrec_equal(S, R0[2 * b - 1]);
    }
    const unsigned long long mb = __ballot(bad);
    if (threadIdx.x == 0) {
        v.metaBad[cg.chunk0 + gx / CPW] = mb;
        if (mb) atomicAdd(&stats[5], (unsigned long long) __popcll(mb));
    }
}

// a record (wave-uniform) in the wave-wide form of the chain's state
__device__ __forceinline__ void mrec_expand(const MetaRec &R, MetaRun &st) {
    st.claimed = rfl64(R.claimed); st.curClaimed = (rfl32(R.flags) & MREC_CUR) != 0;
    st.gapStartIdx = (int64_t) rfl64((uint64_t) R.gapStartIdx); st.gapEndIdx = (int64_t) rfl64((uint64_t) R.gapEndIdx);
    st.nxA = 0; st.nxB = 0;
#pragma unroll
    for (int e = 0; e < MNX; e++) {
        const int rel = (int) rfl32(R.rel[e]);
        const uint64_t val = rfl64(R.val[e]);
        if (rel < WAVE) st.nxA = wl64(val, rel, st.nxA);
        else if (rel == WAVE) st.nxB = wl64(val, 0, st.nxB);
    }
}

__global__ void __launch_bounds__(WAVE) k_emit_meta_stitch(EmitView v, const EmitContig *__restrict__ cgs, const MetaRec *__restrict__ recs,
                                                           unsigned long long *__restrict__ stats) {
    __shared__ MetaLds L;
    const EmitContig cg = cgs[blockIdx.x];
    const EmitOut o = v.out[blockIdx.x];
    if (o.unmatchedChars == UINT64_MAX) return;
    const int64_t n = (int64_t) o.nmatches;
    const int lane = (int) threadIdx.x;
    const bool lazy = v.p.lazyDecompressionSupport != 0;
    const int64_t nblk = (n + MB - 1) / MB, ngrp = (n + META_GROUP - 1) / META_GROUP;
    constexpr int BPG = META_GROUP / MB;                             // blocks per group
    const MetaRec *R0 = recs + (size_t) cg.chunk0 * (CH / MB) * 2;   // block b: R0[2b] where it started, R0[2b + 1] where it ended
    MetaRun st;                                                      // the true state at the start of group g
    st.claimed = 0; st.curClaimed = false; st.gapStartIdx = -1; st.gapEndIdx = -1; st.nxA = 0; st.nxB = 0;
    uint32_t replayed = 0;
    const unsigned long long *BW = v.metaBad + cg.chunk0;           // k_emit_meta_check's bits, word b / 64
    const int64_t nwords = (nblk + WAVE - 1) / WAVE;
    for (int64_t g = 0; g < ngrp;) {
        // the first block from BPG * g on that cannot be accepted. The group's first block is held against the true state
        // (behind a replay its predecessor's record is not the truth any more); every later one was held against its
        // predecessor's end by k_emit_meta_check — on similar genomes no bit is set and the kernel ends here.
        const int64_t h = BPG * g;
        int64_t first = nblk;
        {
            const MetaRec S = R0[2 * h], F = R0[2 * h + 1];
            bool bad = ((S.flags | F.flags) & MREC_OVERFLOW) != 0;
            if (!bad && g > 0) {
                // (records carry an inherited boundary only where it differs from the match's own, :254 treats both alike)
                MetaRun t;
                mrec_expand(S, t);
                const int64_t J = h * MB;
                const uint64_t ownA = lazy && J + lane < n ? v.next0[cg.scratchBase + J + lane] : 0;
                const uint64_t ownB = lazy && J + WAVE < n ? v.next0[cg.scratchBase + J + WAVE] : 0;
                if (st.nxA == ownA) st.nxA = 0;
                if (st.nxB == ownB) st.nxB = 0;
                const bool d = t.nxA != st.nxA || (lane == 0 && (t.nxB != st.nxB || t.claimed != st.claimed || t.gapStartIdx != st.gapStartIdx ||
                                                                 t.gapEndIdx != st.gapEndIdx || t.curClaimed != st.curClaimed));
                bad = __ballot(d) != 0;
            }
            if (bad) first = h;
        }
        for (int64_t w0 = (h + 1) / WAVE; w0 < nwords && first == nblk; w0 += WAVE) {
            const int64_t w = w0 + lane;
            unsigned long long bits = w < nwords ? BW[w] : 0ull;
            if (w == (h + 1) / WAVE) bits &= ~0ull << ((h + 1) & (WAVE - 1));              // (blocks up to h are settled)
            const unsigned long long any = __ballot(bits != 0);
            if (any) {
                const int l = __builtin_ctzll(any);
                first = (w0 + l) * WAVE + __builtin_ctzll(rl64(bits, l));
            }
        }
        if (first >= nblk) break;
        const int64_t gb = first / BPG;
        if (gb > g) mrec_expand(R0[2 * (BPG * gb - 1) + 1], st);     // (the block in front of group gb was accepted: its end is the true state)
        g = gb;
        const int64_t j0 = g * META_GROUP, j1 = j0 + META_GROUP < n ? j0 + META_GROUP : n;
        meta_load(v, cg, L, j0, j1 + WAVE < n ? j1 + WAVE : n, lazy);
        meta_run<true>(v, cg, L, st, j0, j0, j1, n);                 // replay the group from the true state
        replayed++;
        g++;
    }
    if (lane == 0 && replayed) atomicAdd(&stats[6], (unsigned long long) replayed);
}

// ------------------------------------------------------------------------------------------------
// byte-level automata. W = false: only count; W = true: also write.
// ------------------------------------------------------------------------------------------------
struct ExtRes { uint32_t consumed, nlit, nfl, matched, mism; };

// The automata walk the reference and the query one byte at a time, and every byte used to be a dependent
// global load (a microsecond each once the GPU is busy). ByteWin keeps the 8 bytes around the last index in
// a register: one load per 8 steps, forward (window starts at the index) or backward (window ends at it).
__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }
// index of the first zero byte of w counted from the low end (8: none) / from the high end
__device__ __forceinline__ int zero_lo(uint64_t w) {
    const uint64_t t = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;
    return t ? (__builtin_ctzll(t) >> 3) : 8;
}
__device__ __forceinline__ int zero_hi(uint64_t w) {
    // (the borrow trick above is only exact for the lowest zero byte: test the bytes from the top one by one)
    for (int k = 0; k < 8; k++) if (((w >> (56 - 8 * k)) & 0xFF) == 0) return k;
    return 8;
}

template <bool FWD>
struct ByteWin {
    const uint8_t *__restrict__ base; // (never the arena the automata write to)
    int64_t hi;                       // bytes [0, hi) of base may be read
    int64_t at;
    uint64_t w;
    __device__ __forceinline__ ByteWin(const uint8_t *b, int64_t limit) : base(b), hi(limit), at(INT64_MIN / 2), w(0) {}
    __device__ __forceinline__ uint8_t get(int64_t i) {
        if (hi < 8) return base[i];
        if (i < at || i >= at + 8) {
            at = FWD ? i : i - 7;
            if (at < 0) at = 0;
            if (at + 8 > hi) at = hi - 8;
            __builtin_memcpy(&w, base + at, 8);
        }
        return (uint8_t) (w >> (8 * (int) (i - at)));
    }
};

// extendMatchRight, MBGC_Encoder.cpp:310-371
template <bool W>
__device__ ExtRes ext_right(const EmitView &v, const uint8_t *__restrict__ gapPtr, int64_t gapAvail, int64_t src, uint64_t length, bool isGap, bool gapStart,
                            bool gapMiddle, bool gapEnd, uint8_t *__restrict__ lit, uint8_t *__restrict__ fl) {
    ExtRes r = {0, 0, 0, 0, 0};
    const swsem_emit_params_t &p = v.p;
    if (length == 0) {
        if (gapMiddle) { if (W) fl[0] = 1; r.nfl = 1; }
        return r;
    }
    ByteWin<true> ref(v.ref, (int64_t) v.maxRefLength + 8), gap(gapPtr, gapAvail);
    const bool lazy = p.lazyDecompressionSupport != 0;
    const int64_t loading = (int64_t) v.pos1;
    const int64_t srcGuard = src + (int64_t) length;
    int64_t valid = src + (int64_t) length;
    if (src == loading) valid = src;
    if (!isGap) {
        const int64_t srcEnd = (int64_t) v.maxRefLength;
        if (valid > srcEnd) valid = srcEnd;
        if (src <= loading && loading < valid) valid = loading;
    }
    uint32_t g = 0;
    if (gapStart || !isGap) {
        if (lazy && ref.get(src) == 0) valid = src;
        r.mism++;
        const uint8_t b = p.mismatchesWithExclusion && src < valid ? mismatch2code(ref.get(src), gap.get(0)) : gap.get(0);
        if (W) lit[r.nlit] = b;
        r.nlit++;
        g++;
    } else
        src--;
    int score = p.mmsMismatchesInitialScore;
    while (++src < valid && (!lazy || ref.get(src) != 0) && (isGap || score < p.mmsMismatchesScoreThreshold)) {
        // A stretch of matching bytes changes nothing but counters (flag 0 each, the score only falls): 8 bytes at a
        // time up to the first mismatch / separator byte. What is left takes the byte-wise step below.
        while (src + 8 <= valid && (int64_t) g + 8 <= gapAvail) {
            const uint64_t rw = ld_u64(v.ref + src), x = rw ^ ld_u64(gapPtr + g);
            int c = x ? (__builtin_ctzll(x) >> 3) : 8;
            if (lazy) { const int z = zero_lo(rw); if (z < c) c = z; }
            if (c == 0) break;
            if (W) for (int k = 0; k < c; k++) fl[r.nfl + k] = 0;
            r.nfl += c; r.matched += c; g += c; src += c;
            score -= p.mmsMatchBonus * c;
            if (score < 0) score = 0;
            if (c < 8) break;
        }
        if (!(src < valid && (!lazy || ref.get(src) != 0))) break;   // (the score test still holds: it only fell)
        const bool mm = gap.get(g) != ref.get(src);
        if (W) fl[r.nfl] = mm ? 1 : 0;
        r.nfl++;
        if (mm) {
            r.mism++;
            score += p.mmsMismatchPenalty;
            const uint8_t b = p.mismatchesWithExclusion ? mismatch2code(ref.get(src), gap.get(g)) : gap.get(g);
            if (W) lit[r.nlit] = b;
            r.nlit++;
        } else {
            r.matched++;
            score -= p.mmsMatchBonus;
            if (score < 0) score = 0;
        }
        g++;
    }
    while (src++ < srcGuard && (isGap || score < p.mmsMismatchesScoreThreshold)) {
        if (W) { fl[r.nfl] = 1; lit[r.nlit] = gap.get(g); }
        r.nfl++; r.nlit++; g++;
        r.mism++;
        score += p.mmsMismatchPenalty;
    }
    if ((isGap && !gapEnd) || (!isGap && score < p.mmsMismatchesScoreThreshold)) { if (W) fl[r.nfl] = 1; r.nfl++; }
    r.consumed = g;
    return r;
}

// A long extension by a whole wave. One thread takes 0.6 us per byte, and extensions are long exactly where they matter:
// between two paired matches the right extension walks the whole stretch whatever its length (isGap, :338-371), and in a
// collection that diverges by a few percent the score of an ordinary extension (+penalty on a mismatch, -bonus but not below
// zero on a match, :346-365, :404-421) rarely reaches the threshold, so it runs from one match to the next — kilobytes.
// What byte k contributes depends on the byte alone (in front of position e: compared with the reference; behind it: a
// mismatch taken as it is) — except where the walk stops: behind the first byte that lifts the score to the threshold. The
// score after every byte of a row of 64 is a prefix composition of x -> max(x + a, b) steps, which compose: the wave scans
// them, finds the first lane at the threshold by a ballot and places the codes by another. Arguments are wave-uniform.
constexpr uint32_t EXT_WIDE_MIN = 192;                                 // shorter stretches stay with their thread
constexpr int SCORE_NEG = -(1 << 28);
struct WideOut { uint32_t nlit, matched, consumed; bool crossed; };    // (flags written = consumed)
// bytes k0 .. n-1: byte k = q[qdir * k] against ref[rdir * k] while k < e, where e is `valid` or — under lazy decompression —
// the first k with a separator byte (0) in the reference, found on the way; flags to fl[k - k0], codes to lit[nlit0 ...]
template <bool W>
__device__ WideOut ext_wide(const uint8_t *ref, int rdir, const uint8_t *q, int qdir, uint32_t k0, uint32_t n, uint32_t valid, bool lazy, bool excl,
                            bool useScore, int score, int pen, int bonus, int thr, uint8_t *lit, uint32_t nlit0, uint8_t *fl) {
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    WideOut r = {nlit0, 0, 0, false};
    if (useScore && score >= thr) { r.crossed = true; return r; }
    uint32_t e = valid > k0 ? valid : k0;
    constexpr int U = 4;                                               // rows of 64 bytes whose loads are in flight together
    uint8_t nq[U], nr[U];                                              // the next group's bytes: loaded before this group's codes are stored
    auto load_group = [&](uint32_t b0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = b0 + (uint32_t) u * WAVE + lane;
            nq[u] = k < n ? q[(int64_t) qdir * (int64_t) k] : 0;
            nr[u] = k < n && k < e ? ref[(int64_t) rdir * (int64_t) k] : 1;
        }
    };
    load_group(k0);
    for (uint32_t base0 = k0; base0 < n && !r.crossed; base0 += U * WAVE) {
        uint8_t qbs[U], rbs[U];
#pragma unroll
        for (int u = 0; u < U; u++) { qbs[u] = nq[u]; rbs[u] = nr[u]; }
        if (base0 + U * WAVE < n) load_group(base0 + U * WAVE);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t base = base0 + (uint32_t) u * WAVE;
            if (base >= n || r.crossed) break;
            const uint32_t k = base + lane;
            const bool live = k < n;
            const uint8_t qb = qbs[u], rb = rbs[u];
            if (lazy) {                                                // the comparison ends at the first separator byte
                const unsigned long long z = __ballot(live && k < e && rb == 0);
                if (z) e = base + (uint32_t) __builtin_ctzll(z);
            }
            const bool cmp = k < e;
            const bool mm = live && (!cmp || qb != rb);
            uint32_t last = (n - base < (uint32_t) WAVE ? n - base : (uint32_t) WAVE) - 1;    // last lane of the row that is walked
            int after = score;
            if (useScore) {
                int fa = !live ? 0 : (mm ? pen : -bonus), fb = !live || mm ? SCORE_NEG : 0;     // this byte's step x -> max(x + fa, fb)
                for (int d = 1; d < WAVE; d <<= 1) {                                            // inclusive scan: steps of lanes 0 .. lane, composed
                    const int pa = __shfl_up(fa, d), pb = __shfl_up(fb, d);
                    if ((int) lane >= d) { const int nb = pb + fa > fb ? pb + fa : fb; fa = pa + fa; fb = nb < SCORE_NEG ? SCORE_NEG : nb; }
                }
                after = score + fa > fb ? score + fa : fb;
                const unsigned long long cm = __ballot(live && after >= thr);
                if (cm) { r.crossed = true; last = (uint32_t) __builtin_ctzll(cm); }
            }
            const bool proc = live && lane <= last;
            const unsigned long long mmMask = __ballot(proc && mm);
            if (W && proc) {
                fl[k - k0] = mm ? 1 : 0;
                if (mm) lit[r.nlit + (uint32_t) __popcll(mmMask & ((1ull << lane) - 1ull))] = cmp && excl ? mismatch2code(rb, qb) : qb;
            }
            const uint32_t nmm = (uint32_t) __popcll(mmMask);
            r.nlit += nmm;
            r.matched += last + 1 - nmm;
            r.consumed += last + 1;
            if (useScore) score = (int) rl32((uint32_t) after, (int) last);
        }
    }
    return r;
}

// extendMatchRight (:310-371) of a stretch of `length` bytes by the wave
template <bool W>
__device__ ExtRes ext_right_wide(const EmitView &v, const uint8_t *gapPtr, int64_t src0, uint32_t length, bool isGap, bool gapStart, bool gapEnd,
                                 uint8_t *lit, uint8_t *fl) {
    const swsem_emit_params_t &p = v.p;
    const bool lazy = p.lazyDecompressionSupport != 0, excl = p.mismatchesWithExclusion != 0;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint8_t *ref = v.ref + src0;
    const int64_t loading = (int64_t) v.pos1;
    int64_t valid = src0 + (int64_t) length;                            // :318-335
    if (src0 == loading) valid = src0;
    if (!isGap) {
        if (valid > (int64_t) v.maxRefLength) valid = (int64_t) v.maxRefLength;
        if (src0 <= loading && loading < valid) valid = loading;
    }
    uint32_t validLen = valid > src0 ? (uint32_t) (valid - src0) : 0u;
    const uint32_t first = gapStart || !isGap ? 1u : 0u;                // the stretch opens with a literal that has no flag
    if (first && lazy && ref[0] == 0) validLen = 0;
    uint32_t nlit = 0;
    if (first) {
        if (W && lane == 0) lit[0] = excl && validLen ? mismatch2code(ref[0], gapPtr[0]) : gapPtr[0];
        nlit = 1;
    }
    const WideOut o = ext_wide<W>(ref, 1, gapPtr, 1, first, length, validLen, lazy, excl, !isGap, p.mmsMismatchesInitialScore, p.mmsMismatchPenalty, p.mmsMatchBonus,
                                  p.mmsMismatchesScoreThreshold, lit, nlit, fl);
    ExtRes r;
    r.nlit = o.nlit; r.nfl = o.consumed; r.matched = o.matched; r.mism = o.nlit; r.consumed = first + o.consumed;
    if ((isGap && !gapEnd) || (!isGap && !o.crossed)) { if (W && lane == 0) fl[r.nfl] = 1; r.nfl++; }     // :372
    return r;
}

// extendMatchLeft (:373-427) of at most `length` bytes in front of match m by the wave
template <bool W>
__device__ ExtRes ext_left_wide(const EmitView &v, const uint8_t *dest, uint64_t length, uint64_t mPosSrc, uint64_t mPosDest, uint64_t lockPos,
                                uint8_t *lit, uint8_t *fl) {
    ExtRes r = {0, 0, 0, 0, 0};
    const swsem_emit_params_t &p = v.p;
    const bool lazy = p.lazyDecompressionSupport != 0, excl = p.mismatchesWithExclusion != 0;
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const int64_t srcMatch = (int64_t) mPosSrc;
    int64_t guard = 1;
    if (guard < srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH) guard = srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH;
    const int64_t srcLock = (int64_t) lockPos;
    if (guard < srcLock && srcLock <= srcMatch) guard = srcLock;
    bool guardKnown = true;
    if (guard < srcMatch - (int64_t) length) { guard = srcMatch - (int64_t) length; guardKnown = false; }
    if (guard == srcMatch) return r;
    const uint32_t maxBytes = (uint32_t) (srcMatch - guard);            // bytes srcMatch-1 .. guard, walked downwards
    const uint8_t *ref = v.ref + srcMatch - 1, *q = dest + mPosDest - 1;
    const bool validRegion = !lazy || ref[0] != 0;
    if (W && lane == 0) lit[0] = excl && validRegion ? mismatch2code(ref[0], q[0]) : q[0];
    // compared bytes: [1, maxBytes) up to a separator byte — it and what lies below it are taken as they are
    const WideOut o = ext_wide<W>(ref, -1, q, -1, 1, maxBytes, validRegion ? maxBytes : 1u, lazy, excl, true, p.mmsMismatchesInitialScore, p.mmsMismatchPenalty, p.mmsMatchBonus,
                                  p.mmsMismatchesScoreThreshold, lit, 1, fl);
    r.nlit = o.nlit; r.nfl = o.consumed; r.matched = o.matched; r.mism = o.nlit; r.consumed = 1 + o.consumed;
    const bool atGuard = r.consumed == maxBytes;
    if ((!atGuard || !guardKnown) && !o.crossed) { if (W && lane == 0) fl[r.nfl] = 1; r.nfl++; }             // :424
    return r;
}

// extendMatchLeft, MBGC_Encoder.cpp:373-427
template <bool W>
__device__ ExtRes ext_left(const EmitView &v, const uint8_t *__restrict__ dest, int64_t destLen, uint64_t length, const EMatch &m, uint64_t lockPos,
                           uint8_t *__restrict__ lit, uint8_t *__restrict__ fl) {
    ExtRes r = {0, 0, 0, 0, 0};
    const swsem_emit_params_t &p = v.p;
    ByteWin<false> ref(v.ref, (int64_t) v.maxRefLength + 8), gq(dest, destLen);
    const bool lazy = p.lazyDecompressionSupport != 0;
    const int64_t srcMatch = (int64_t) m.posSrc;
    int64_t guard = 1;
    if (guard < srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH) guard = srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH;
    const int64_t srcLock = (int64_t) lockPos;           // SIZE_MAX behaves like -1 here, as ref + SIZE_MAX does there
    if (guard < srcLock && srcLock <= srcMatch) guard = srcLock;
    bool guardKnown = true;
    if (guard < srcMatch - (int64_t) length) { guard = srcMatch - (int64_t) length; guardKnown = false; }
    if (guard == srcMatch) return r;
    int64_t src = srcMatch - 1;
    int64_t gp = (int64_t) m.posDest - 1;                 // index into the contig
    bool validRegion = !lazy || ref.get(src) != 0;
    r.mism++;
    {
        const uint8_t b = p.mismatchesWithExclusion && validRegion ? mismatch2code(ref.get(src), gq.get(gp)) : gq.get(gp);
        if (W) lit[r.nlit] = b;
        r.nlit++;
    }
    int score = p.mmsMismatchesInitialScore;
    while (validRegion && src > guard && score < p.mmsMismatchesScoreThreshold) {
        // matching stretches 8 bytes at a time (see ext_right), walking down: bytes src-1 .. src-8 against gp-1 .. gp-8
        while (src - 8 >= guard && gp >= 8) {
            const uint64_t rw = ld_u64(v.ref + src - 8), x = rw ^ ld_u64(dest + gp - 8);
            int c = x ? (__builtin_clzll(x) >> 3) : 8;
            if (lazy) { const int z = zero_hi(rw); if (z < c) c = z; }
            if (c == 0) break;
            if (W) for (int k = 0; k < c; k++) fl[r.nfl + k] = 0;
            r.nfl += c; r.matched += c; gp -= c; src -= c;
            score -= p.mmsMatchBonus * c;
            if (score < 0) score = 0;
            if (c < 8) break;
        }
        if (!(src > guard)) break;
        --gp; --src;
        const bool mm = gq.get(gp) != ref.get(src);
        if (lazy && ref.get(src) == 0) { validRegion = false; src++; gp++; break; }
        if (W) fl[r.nfl] = mm ? 1 : 0;
        r.nfl++;
        if (mm) {
            score += p.mmsMismatchPenalty;
            r.mism++;
            const uint8_t b = p.mismatchesWithExclusion ? mismatch2code(ref.get(src), gq.get(gp)) : gq.get(gp);
            if (W) lit[r.nlit] = b;
            r.nlit++;
        } else {
            score -= p.mmsMatchBonus;
            if (score < 0) score = 0;
            r.matched++;
        }
    }
    while (!validRegion && src > guard && score < p.mmsMismatchesScoreThreshold) {
        src--;
        --gp;
        if (W) { fl[r.nfl] = 1; lit[r.nlit] = gq.get(gp); }
        r.nfl++; r.nlit++;
        r.mism++;
        score += p.mmsMismatchPenalty;
    }
    if ((src != guard || !guardKnown) && score < p.mmsMismatchesScoreThreshold) { if (W) fl[r.nfl] = 1; r.nfl++; }
    r.consumed = (uint32_t) (srcMatch - src);
    return r;
}

__device__ __forceinline__ uint32_t frugal_size(uint64_t v) {        // writeUInt64Frugal, utils/helper.cpp:237-246
    return v < 0xFFFFu ? 2u : (v < 0xFFFFFFFFull ? 6u : 14u);
}
__device__ __forceinline__ void put_bytes(uint8_t *d, uint64_t v, int n) {
    for (int i = 0; i < n; i++) d[i] = (uint8_t) (v >> (8 * i));
}
__device__ void frugal_write(uint8_t *d, uint64_t v) {
    put_bytes(d, v < 0xFFFFu ? v : 0xFFFFu, 2);
    if (v >= 0xFFFFu) {
        put_bytes(d + 2, v < 0xFFFFFFFFull ? v : 0xFFFFFFFFull, 4);
        if (v >= 0xFFFFFFFFull) put_bytes(d + 6, v, 8);
    }
}

// Gap task t (0..n): the bytes between kept match t-1 and kept match t (contig start / end at the rims).
// It owns: the right extension of match t-1, the left extension of match t and the plain literals in
// front of match t (or the contig tail).
struct GapSizes { uint32_t rLit, rFl, lLit, lFl, plain, pos; };

// right extension of match t-1 (t >= 1), :279-286: where it starts and what it may consume
struct RightTask { uint32_t pos; uint64_t litLeft; int64_t src; bool isGap, gStart, gMid, gEnd; };
__device__ __forceinline__ RightTask right_task(const EmitView &v, const EmitContig &cg, const EMatch *E, int64_t n, int64_t t) {
    RightTask k;
    const EMatch mp = E[t - 1];
    const uint32_t meta = v.meta[cg.scratchBase + t - 1];
    k.pos = (uint32_t) (mp.posDest + mp.len);                                         // uint32_t pos, :145,:240
    k.litLeft = (t < n ? E[t].posDest : cg.n) - k.pos;                                // :242
    k.isGap = (meta & META_ISGAP) != 0;
    k.gStart = (meta & META_GSTART) != 0; k.gMid = (meta & META_GMID) != 0; k.gEnd = (meta & META_GEND) != 0;
    k.src = 0;
    if (v.p.enableExtensionsWithMismatches) {
        const EMatch core = E[v.corr[cg.scratchBase + t - 1]];
        k.src = k.isGap ? (int64_t) (core.posSrc + (mp.posDest + mp.len) - core.posDest) : (int64_t) (mp.posSrc + mp.len);
    }
    return k;
}
// a stretch long enough for the whole wave (ext_right_wide)
__device__ __forceinline__ bool right_task_wide(const EmitView &v, const RightTask &k) {
    return v.p.enableExtensionsWithMismatches && k.litLeft > EXT_WIDE_MIN && k.litLeft <= UINT32_MAX;
}
// Returns the position after it.
template <bool W>
__device__ uint32_t gap_right(const EmitView &v, const EmitContig &cg, const RightTask &k, const uint8_t *q,
                              uint8_t *rLit, uint8_t *rFl, GapSizes &s, uint32_t *counters) {
    uint32_t pos = k.pos;
    if (v.p.enableExtensionsWithMismatches) {
        const ExtRes r = ext_right<W>(v, q + pos, (int64_t) cg.n - (int64_t) pos, k.src, k.litLeft, k.isGap, k.gStart, k.gMid, k.gEnd, rLit, rFl);
        s.rLit = r.nlit; s.rFl = r.nfl;
        pos += r.consumed;
        if (!W) { counters[0] += r.matched; counters[1] += r.mism; }
    }
    return pos;
}

// The task of a thread in a block that serves one chunk of 256 gap tasks with 256 * S threads: a wave holds 64 / S tasks in
// its first lanes. A wave walks the long stretches of its tasks one after the other (all lanes on each), so on sequence that
// diverges by a few percent — every gap a stretch of hundreds of bytes — a wave's time is its number of tasks: a batch that
// cannot fill the chip with waves of 64 tasks (one contig of the sequential schedule, a unit of two targets) is launched with
// S = 4, four times the waves and a quarter of the stretches each; -1: a lane without a task.
template <int S>
__device__ __forceinline__ int64_t chunk_task(uint32_t gx) {
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    return lane < (uint32_t) (WAVE / S) ? (int64_t) gx * 256 + wave * (WAVE / S) + lane : -1;
}

// sizes of every gap task, one block per chunk of 256 tasks
template <int S>
__global__ void __launch_bounds__(256 * S) k_emit_sizes(EmitView v, const EmitContig *__restrict__ cgs) {
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const EmitOut o = v.out[gk];
    if (o.unmatchedChars == UINT64_MAX) return;
    const int64_t n = (int64_t) o.nmatches;
    if ((int64_t) gx * 256 > n) return;
    const int64_t tt = chunk_task<S>(gx);
    const int64_t t = tt < 0 ? n + 1 : tt;                          // (a lane without a task: beyond the last one)
    uint32_t cloc[2] = {0, 0};
    const EMatch *E = v.em + cg.scratchBase;
    const uint8_t *q = v.qbuf + cg.qoff;
    GapSizes s = {0, 0, 0, 0, 0, 0};
    RightTask k = {0, 0, 0, false, false, false, false};
    uint32_t pos = 0;
    bool wide = false;
    if (t <= n && t >= 1) {
        k = right_task(v, cg, E, n, t);
        wide = right_task_wide(v, k);
        pos = wide ? k.pos : gap_right<false>(v, cg, k, q, nullptr, nullptr, s, cloc);
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    for (unsigned long long todo = __ballot(wide); todo; todo &= todo - 1) {      // the long stretches of this wave's tasks, one after the other, all lanes on each
        const int l = __builtin_ctzll(todo);
        const uint32_t wpos = rl32(k.pos, l);
        const ExtRes r = ext_right_wide<false>(v, q + wpos, (int64_t) rl64((uint64_t) k.src, l), rl32((uint32_t) k.litLeft, l), rl32(k.isGap, l) != 0,
                                               rl32(k.gStart, l) != 0, rl32(k.gEnd, l) != 0, nullptr, nullptr);
        if ((int) lane == l) {
            s.rLit = r.nlit; s.rFl = r.nfl;
            pos = k.pos + r.consumed;
            cloc[0] += r.matched; cloc[1] += r.mism;
        }
    }
    // the left extension of match t (:218-221) and the plain literals in front of it
    uint64_t litLeft = 0;
    EMatch m = {0, 0, 0, 0};
    bool wideL = false;
    if (t <= n) {
        if (t < n) {
            m = E[t];
            litLeft = m.posDest - pos;                                                      // :216
            if (v.p.enableExtensionsWithMismatches && !k.isGap && litLeft) {
                wideL = litLeft > EXT_WIDE_MIN;
                if (!wideL) {
                    const ExtRes r = ext_left<false>(v, q, (int64_t) cg.n, litLeft, m, cg.lock, nullptr, nullptr);
                    s.lLit = r.nlit; s.lFl = r.nfl;
                    litLeft -= r.consumed;
                    cloc[0] += r.matched; cloc[1] += r.mism;
                }
            }
        } else
            litLeft = cg.n - pos;                                                           // :288-289
    }
    for (unsigned long long todo = __ballot(wideL); todo; todo &= todo - 1) {
        const int l = __builtin_ctzll(todo);
        const ExtRes r = ext_left_wide<false>(v, q, rl64(litLeft, l), rl64(m.posSrc, l), rl64(m.posDest, l), cg.lock, nullptr, nullptr);
        if ((int) lane == l) {
            s.lLit = r.nlit; s.lFl = r.nfl;
            litLeft -= r.consumed;
            cloc[0] += r.matched; cloc[1] += r.mism;
        }
    }
    if (t <= n) {
        s.plain = (uint32_t) litLeft; s.pos = pos;
        uint32_t *z = v.sz + (cg.scratchBase + t) * 6;
        z[0] = s.rLit; z[1] = s.rFl; z[2] = s.lLit; z[3] = s.lFl; z[4] = s.plain; z[5] = s.pos;
    }
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        cloc[0] += (uint32_t) __shfl_down((int) cloc[0], d);
        cloc[1] += (uint32_t) __shfl_down((int) cloc[1], d);
    }
    // one pair of atomics per block: every block of a contig adds to the same two words
    __shared__ uint32_t part[2 * (256 * S / WAVE)];
    if ((threadIdx.x & (WAVE - 1)) == 0) { part[2 * (threadIdx.x / WAVE)] = cloc[0]; part[2 * (threadIdx.x / WAVE) + 1] = cloc[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < 256 * S / WAVE; w++) { a += part[2 * w]; b += part[2 * w + 1]; }
        if (a | b) {
            atomicAdd((unsigned long long *) &v.out[gk].extMatched, a);
            atomicAdd((unsigned long long *) &v.out[gk].extMismatches, b);
        }
    }
}

// placement. Iteration t of the reference's loop emits, in this order:
//   literals: [left codes t][plain t] MATCH_MARK [right codes t]      flags: [left flags t][right flags t]
// where "right codes t" are produced by gap task t+1. Three steps: per-chunk sums of what every
// iteration adds to each of the six streams, a per-contig scan over the chunks, and the start offsets
// of every iteration inside its chunk.
struct IterSizes { uint32_t s[6]; };
__device__ __forceinline__ IterSizes iter_sizes(const EmitView &v, const EmitContig &cg, int64_t n, int64_t t) {
    IterSizes r;
    const uint32_t *z = v.sz + (cg.scratchBase + t) * 6;
    r.s[0] = z[2] + z[4]; r.s[1] = z[3]; r.s[2] = 0; r.s[3] = 0; r.s[4] = 0; r.s[5] = 0;
    if (t < n) {
        const uint32_t meta = v.meta[cg.scratchBase + t];
        r.s[0] += 1 + z[6];
        r.s[1] += z[7];
        if (!(meta & META_SKIPOFF)) { r.s[2] = 4; r.s[3] = v.p.enable40bitReference ? 1 : 0; }
        r.s[4] = v.p.frugal64bitLenEncoding ? frugal_size(v.em[cg.scratchBase + t].len) : 4u;
        r.s[5] = (meta & META_HASGAP) ? 1u : 0u;
    }
    return r;
}

__global__ void __launch_bounds__(CH) k_emit_place_sums(EmitView v, const EmitContig *__restrict__ cgs) {
    __shared__ uint32_t acc[6];
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const EmitOut o = v.out[gk];
    if (o.unmatchedChars == UINT64_MAX) return;
    const int64_t n = (int64_t) o.nmatches;
    if ((int64_t) gx * CH > n) return;
    if (threadIdx.x < 6) acc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t t = (int64_t) gx * CH + threadIdx.x;
    IterSizes r = {{0, 0, 0, 0, 0, 0}};
    if (t <= n) r = iter_sizes(v, cg, n, t);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t x = r.s[k];
        for (int d = WAVE / 2; d > 0; d >>= 1) x += (uint32_t) __shfl_down((int) x, d);
        if ((threadIdx.x & (WAVE - 1)) == 0 && x) atomicAdd(&acc[k], x);
    }
    __syncthreads();
    if (threadIdx.x < 6) v.chunkCnt[((size_t) blockIdx.x) * 6 + threadIdx.x] = acc[threadIdx.x];
}

__global__ void __launch_bounds__(CH) k_emit_place_scan(EmitView v, const EmitContig *__restrict__ cgs) {
    __shared__ uint32_t lds[CH / WAVE + 2];
    const EmitOut o = v.out[blockIdx.x];
    if (o.unmatchedChars == UINT64_MAX) return;
    const uint32_t nch = (uint32_t) ((o.nmatches + 1 + CH - 1) / CH);
    uint32_t *cc = v.chunkCnt + (size_t) cgs[blockIdx.x].chunk0 * 6;
    uint32_t base[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t c0 = 0; c0 < nch; c0 += CH) {
        const uint32_t c = c0 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint32_t x = c < nch ? cc[(size_t) c * 6 + k] : 0;
            uint32_t tot;
            const uint32_t ex = block_scan<CH>(x, lds, &tot);
            if (c < nch) cc[(size_t) c * 6 + k] = base[k] + ex;
            base[k] += tot;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        EmitOut *op = v.out + blockIdx.x;
        op->size[SWSEM_LIT] = base[0]; op->size[SWSEM_FLAGS] = base[1]; op->size[SWSEM_OFF] = base[2]; op->size[SWSEM_OFF5] = base[3];
        op->size[SWSEM_LEN] = base[4]; op->size[SWSEM_GAP] = base[5];
    }
}

// the six streams of all contigs back to back, (contig, stream) major: the arena is written packed
__global__ void __launch_bounds__(CH) k_emit_packoffs(EmitView v) {
    __shared__ uint32_t lds[CH / WAVE + 2];
    if (threadIdx.x == 0) *v.longCount = 0;                          // (k_emit_write, behind this kernel, fills the list)
    uint64_t run = 0;
    for (uint32_t k0 = 0; k0 < v.ncontigs; k0 += CH) {
        const uint32_t k = k0 + threadIdx.x;
        const bool live = k < v.ncontigs && v.out[k].unmatchedChars != UINT64_MAX;
        uint32_t mine = 0;                                           // a contig's streams stay below 4 GiB (32-bit offsets)
        if (live) for (int s = 0; s < SWSEM_NSTREAMS; s++) mine += (uint32_t) v.out[k].size[s];
        // 64-bit running base, 32-bit scan inside a group of CH contigs would overflow for 256 x 4 GiB: scan halves
        uint32_t totLo, totHi;
        const uint32_t exLo = block_scan<CH>(mine & 0xFFFFu, lds, &totLo);
        __syncthreads();
        const uint32_t exHi = block_scan<CH>(mine >> 16, lds, &totHi);
        __syncthreads();
        if (k < v.ncontigs) {
            uint64_t at = run + exLo + ((uint64_t) exHi << 16);
            for (int s = 0; s < SWSEM_NSTREAMS; s++) {
                v.packBase[(size_t) k * SWSEM_NSTREAMS + s] = at;
                at += live ? v.out[k].size[s] : 0;
            }
        }
        run += totLo + ((uint64_t) totHi << 16);
    }
}

__global__ void __launch_bounds__(CH) k_emit_place_final(EmitView v, const EmitContig *__restrict__ cgs) {
    __shared__ uint32_t lds[CH / WAVE + 2];
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const EmitOut o = v.out[gk];
    if (o.unmatchedChars == UINT64_MAX) return;
    const int64_t n = (int64_t) o.nmatches;
    if ((int64_t) gx * CH > n) return;
    const int64_t t = (int64_t) gx * CH + threadIdx.x;
    IterSizes r = {{0, 0, 0, 0, 0, 0}};
    if (t <= n) r = iter_sizes(v, cg, n, t);
    const uint32_t *cb = v.chunkCnt + ((size_t) blockIdx.x) * 6;
    uint32_t out[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t tot;
        out[k] = cb[k] + block_scan<CH>(r.s[k], lds, &tot);
        __syncthreads();
    }
    if (t <= n) {
        uint32_t *w = v.ofs + (cg.scratchBase + t) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) w[k] = out[k];
    }
}

// write: one thread per iteration; the extension codes are produced by re-running the automata that
// belong to the iteration (left codes: gap task t, right codes: first half of gap task t+1). A long run of plain
// literals — the unmatched stretch of a divergent contig can be megabytes — is not one thread's work: the wave
// copies it together, 16 bytes per lane and round.
constexpr uint32_t PLAIN_INLINE = 48;                                  // plain literals a thread copies itself
template <int S>
__global__ void __launch_bounds__(256 * S) k_emit_write(EmitView v, const EmitContig *__restrict__ cgs) {
    const uint32_t gk = v.chunkOwner[blockIdx.x];
    const EmitContig cg = cgs[gk];
    const uint32_t gx = blockIdx.x - cg.chunk0;
    const EmitOut o = v.out[gk];
    if (o.unmatchedChars == UINT64_MAX) return;
    const int64_t n = (int64_t) o.nmatches;
    const int64_t tt = chunk_task<S>(gx);
    const int64_t t = tt < 0 ? n + 1 : tt;                          // (a lane without a task: beyond the last one)
    const uint8_t *q = v.qbuf + cg.qoff;
    uint32_t longLen = 0;
    const uint8_t *longSrc = nullptr;
    uint8_t *longDst = nullptr;
    RightTask rk = {0, 0, 0, false, false, false, false};
    bool wide = false, wideL = false;
    uint8_t *wLit = nullptr, *wFl = nullptr, *wlLit = nullptr, *wlFl = nullptr;
    EMatch lm = {0, 0, 0, 0};
    uint64_t leftLen = 0;
    if (t <= n) {
        const EMatch *E = v.em + cg.scratchBase;
        const uint32_t *z = v.sz + (cg.scratchBase + t) * 6;
        const uint32_t *w = v.ofs + (cg.scratchBase + t) * 6;
        const bool bit40 = v.p.enable40bitReference != 0, frugal = v.p.frugal64bitLenEncoding != 0;
        const uint64_t *pb = v.packBase + (size_t) gk * SWSEM_NSTREAMS;
        uint8_t *lLit = v.arena + pb[SWSEM_LIT] + w[0], *lFl = v.arena + pb[SWSEM_FLAGS] + w[1];
        uint8_t *plainDst = lLit + z[2];
        if (z[4] <= PLAIN_INLINE) for (uint32_t k = 0; k < z[4]; k++) plainDst[k] = q[z[5] + k];
        else {
            uint32_t slot = LONG_COPY_CAP;
            if (z[4] >= LONG_COPY_MIN) slot = atomicAdd(v.longCount, 1u);                   // the unmatched stretch of a contig without a relative: megabytes
            if (slot < LONG_COPY_CAP) { LongCopy lc; lc.dst = plainDst; lc.src = q + z[5]; lc.len = z[4]; v.longCopies[slot] = lc; }
            else { longLen = z[4]; longSrc = q + z[5]; longDst = plainDst; }
        }
        if (t < n) {
            const uint32_t *zn = z + 6;
            uint8_t *rLit = plainDst + z[4] + 1, *rFl = lFl + z[3];
            if (z[2] | z[3]) {
                lm = E[t];
                leftLen = lm.posDest - z[5];
                wideL = leftLen > EXT_WIDE_MIN;
                wlLit = lLit; wlFl = lFl;
                if (!wideL) ext_left<true>(v, q, (int64_t) cg.n, leftLen, lm, cg.lock, lLit, lFl);
            }
            plainDst[z[4]] = MATCH_MARK;
            if (zn[0] | zn[1]) {
                rk = right_task(v, cg, E, n, t + 1);
                wide = right_task_wide(v, rk);
                wLit = rLit; wFl = rFl;
                if (!wide) {
                    GapSizes tmp = {0, 0, 0, 0, 0, 0};
                    uint32_t dummy[2];
                    gap_right<true>(v, cg, rk, q, rLit, rFl, tmp, dummy);
                }
            }
            const uint32_t meta = v.meta[cg.scratchBase + t];
            if (!(meta & META_SKIPOFF)) {
                put_bytes(v.arena + pb[SWSEM_OFF] + w[2], (uint32_t) E[t].posSrc, 4);
                if (bit40) v.arena[pb[SWSEM_OFF5] + w[3]] = (uint8_t) (E[t].posSrc >> 32);
            }
            if (frugal) frugal_write(v.arena + pb[SWSEM_LEN] + w[4], E[t].len);
            else put_bytes(v.arena + pb[SWSEM_LEN] + w[4], (uint32_t) E[t].len, 4);
            if (meta & META_HASGAP) v.arena[pb[SWSEM_GAP] + w[5]] = (uint8_t) (meta >> 8);
        }
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    for (unsigned long long todo = __ballot(wideL); todo; todo &= todo - 1) {     // long extensions: the codes of one by all lanes
        const int l = __builtin_ctzll(todo);
        ext_left_wide<true>(v, q, rl64(leftLen, l), rl64(lm.posSrc, l), rl64(lm.posDest, l), cg.lock, (uint8_t *) rl64((uint64_t) wlLit, l),
                            (uint8_t *) rl64((uint64_t) wlFl, l));
    }
    for (unsigned long long todo = __ballot(wide); todo; todo &= todo - 1) {
        const int l = __builtin_ctzll(todo);
        ext_right_wide<true>(v, q + rl32(rk.pos, l), (int64_t) rl64((uint64_t) rk.src, l), rl32((uint32_t) rk.litLeft, l), rl32(rk.isGap, l) != 0,
                             rl32(rk.gStart, l) != 0, rl32(rk.gEnd, l) != 0, (uint8_t *) rl64((uint64_t) wLit, l), (uint8_t *) rl64((uint64_t) wFl, l));
    }
    for (unsigned long long todo = __ballot(longLen != 0); todo; todo &= todo - 1) {
        const int l = __builtin_ctzll(todo);
        const uint32_t len = rl32(longLen, l);
        const uint8_t *src = (const uint8_t *) rl64((uint64_t) longSrc, l);
        uint8_t *dst = (uint8_t *) rl64((uint64_t) longDst, l);
        for (uint32_t at = 16 * lane; at < len; at += 16 * WAVE) {
            if (at + 16 <= len) {
                uint4 x;
                memcpy(&x, src + at, 16);
                memcpy(dst + at, &x, 16);
            } else
                for (uint32_t k = at; k < len; k++) dst[k] = src[k];
        }
    }
}

// the listed runs, 16 KB pieces dealt over all blocks
__global__ void __launch_bounds__(256) k_emit_copy_long(EmitView v) {
    const uint32_t n = *v.longCount < LONG_COPY_CAP ? *v.longCount : LONG_COPY_CAP;
    constexpr uint64_t PIECE = 16384;
    for (uint32_t e = 0; e < n; e++) {
        const LongCopy lc = v.longCopies[e];
        for (uint64_t at = (uint64_t) blockIdx.x * PIECE; at < lc.len; at += (uint64_t) gridDim.x * PIECE) {
            const uint64_t end = at + PIECE < lc.len ? at + PIECE : lc.len;
            for (uint64_t o = at + 16 * (uint64_t) threadIdx.x; o < end; o += 16 * 256) {
                if (o + 16 <= end) { uint4 x; memcpy(&x, lc.src + o, 16); memcpy(lc.dst + o, &x, 16); }
                else for (uint64_t k = o; k < end; k++) lc.dst[k] = lc.src[k];
            }
        }
    }
}

// The prediction a speculative finalize was queued on (swsem_emit_batch_begin_spec): every contig's reference
// extension decision — isContigProperForRefExtension / isContigProperForRefRCExtension, MGMP_Params.h:178-190,
// as MGMP.cpp:389-398 applies them — must come out as predicted, and no contig may have been given up as
// dissimilar. gate[0] = 1 lets the finalize kernels queued behind this one run.
__global__ void k_spec_verify(const EmitOut *__restrict__ out, const EmitContig *__restrict__ cgs, int n, const uint8_t *__restrict__ predExt,
                              const uint8_t *__restrict__ predRC, int factor, int rcFactor, uint32_t *__restrict__ gate) {
    __shared__ uint32_t bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const uint64_t un = out[k].unmatchedChars, len = cgs[k].n;
        const bool ext = un * (uint64_t) factor > len, rc = un * (uint64_t) rcFactor > len;
        if (un == UINT64_MAX || ext != (predExt[k] != 0) || rc != (predRC[k] != 0)) atomicOr(&bad, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) gate[0] = bad ? 0u : 1u;
}

}  // namespace swk
