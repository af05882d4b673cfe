// The decoder's per-contig automaton on the device: the exact inverse of processMatches —
// MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars (mbgccoder/MBGC_Decoder.cpp:319-432), extendMatchRight (:434-460),
// extendMatchLeft (:462-523), one entry of the mapLen stream (decodeMapLenStream :966-986 with readUInt64Frugal,
// utils/helper.h:256-272) and ContextAwareMismatchesCoder::code2mismatch (coders/ContextAwareMismatchesCoder.cpp:8-17,72-77).
//
// The six streams of a contig are consumed in lock-step and carry no synchronisation points (how many flag bytes an
// extension reads depends on the flags themselves), so WHERE every match's bytes lie in the six streams and in the contig
// is one sequential chain per contig — but only that: which bytes they are depends on nothing the chain computes. Two passes:
//   k_decode_plan   one wave per contig walks the automaton without producing a byte: marks, lengths, offsets, the pairing
//                   ring, the score automaton over the flags — every stream read through a register window (64 bytes or 64
//                   dwords per load instead of one dependent load per byte; flags and marks as ballot masks) — and leaves one
//                   record per match: its place in every stream and in the contig, the parameters of its two extensions;
//   k_decode_fill   one thread per record produces the bytes: the plain literals, the left extension (written backwards), the
//                   match copied out of the reference buffer, the right extension — dec_left_write / dec_extend_right, the
//                   same functions the one-wave form runs, now thousands at a time;
//   k_decode_check  (verification) the contig against the bytes it was encoded from.
// The contigs of a round are independent (each stands against the reference as its lock position froze it) and are decoded
// side by side. Used as the device-side check of an emission (swsem_emit_verify: what was just emitted must decode to the
// query, with no encoder logic in the loop) and as the building block of an accelerated `mbgc d`.
#include "swsem_device.h"
#include "../../include/mbgc_swsem.h"

namespace swk {

constexpr uint8_t DEC_MATCH_MARK = 0xA5;                  // MBGC_Params.h:45
constexpr int DEC_MAX_GAP_DEPTH = 128;                    // MBGC_Params.h:50
constexpr int64_t DEC_MAX_EXT_LEFT = 1 << 24;             // MAX_EXTEND_MATCH_LEFT_LENGTH, MBGC_Params.h:55
constexpr uint64_t DEC_NPOS = UINT64_MAX;
constexpr uint64_t DEC_WIDE_MIN = 192;                   // k_decode_fill: longer copies and gap extensions are a wave's work, not a thread's
__device__ __forceinline__ uint64_t rl64d(uint64_t x, int lane) {
    return ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (x >> 32), lane) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) x, lane);
}

struct DecodeJob {                                        // == swsem_decode_job_t (device pointers)
    const uint8_t *stream[SWSEM_NSTREAMS];
    uint64_t size[SWSEM_NSTREAMS];
    uint64_t refLockPos;
    uint8_t *dest;
    uint64_t destCap;
    const uint8_t *expect;                                // when not null: the contig the streams must give back
};
struct DecodeOut { uint64_t destLen; int64_t unmatched; uint64_t firstDiff; };   // unmatched -1: malformed streams; firstDiff: DEC_NPOS = equal

struct Dec {
    const uint8_t *ref;
    const uint8_t *lit, *off, *off5, *len, *gap, *flags;
    uint64_t nLit, nOff, nOff5, nLen, nGap, nFlags;
    uint64_t litPos, offPos, off5Pos, lenPos, gapPos, flPos;
    uint8_t *dest;
    uint64_t destLen, destCap;
    int bad;
    int initialScore, penalty, bonus, threshold;
    bool solo;                                            // one THREAD runs this automaton (k_decode_fill): every lane stores its own bytes
    uint64_t refBytes;                                    // size of the reference buffer: no stream, however malformed, sends a read beyond it
};

__device__ __forceinline__ int dec_sym5(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : -1; }
// code2mis[actual][code], ContextAwareMismatchesCoder.cpp:8-17,72-77 (the inverse of mismatch2code's table)
__device__ uint8_t dec_code2mismatch(Dec &d, uint8_t actual, uint8_t code) {
    const int8_t mis2code[5][5] = {{-1, 2, 0, 1, 3}, {1, -1, 2, 0, 3}, {0, 2, -1, 1, 3}, {1, 0, 2, -1, 3}, {1, 2, 3, 0, -1}};
    if (code >= 5) return code;
    const int a = dec_sym5(actual);
    if (a < 0) { d.bad = 1; return 0; }
    const char sym[6] = "ACGTN";
    for (int j = 0; j < 5; j++)
        if (mis2code[a][j] == (int8_t) code) return (uint8_t) sym[j];
    d.bad = 1;
    return 0;
}
__device__ __forceinline__ uint8_t dec_lit_next(Dec &d) { if (d.litPos >= d.nLit) { d.bad = 1; return 0; } return d.lit[d.litPos++]; }
__device__ __forceinline__ uint8_t dec_flag_at(Dec &d, uint64_t i) { if (i >= d.nFlags) { d.bad = 1; return 0; } return d.flags[i]; }
__device__ __forceinline__ uint8_t dec_ref_at(Dec &d, int64_t i) { if (i < 0 || (uint64_t) i >= d.refBytes) { d.bad = 1; return 0; } return d.ref[i]; }
__device__ __forceinline__ void dec_push(Dec &d, uint8_t c) {
    if (d.destLen >= d.destCap) { d.bad = 1; return; }
    if (d.solo || (threadIdx.x & (WAVE - 1)) == 0) d.dest[d.destLen] = c;
    d.destLen++;
}
// extendMatchRight, :434-460
__device__ uint64_t dec_extend_right(Dec &d, int64_t offsetDelta, bool isGap, bool gapStart, bool gapMiddle, bool gapEnd, uint64_t guardLitPos) {
    if (d.litPos == guardLitPos && !gapMiddle) return 0;
    const uint64_t destStart = d.destLen;
    int64_t src = (int64_t) d.destLen + offsetDelta;
    if (gapStart || !isGap) dec_push(d, dec_code2mismatch(d, dec_ref_at(d, src), dec_lit_next(d)));
    else src--;
    int score = d.initialScore;
    while (!d.bad && (!gapEnd || d.litPos != guardLitPos) && (isGap || score < d.threshold)) {
        const bool mismatch = dec_flag_at(d, d.flPos++) != 0;
        if (mismatch && d.litPos == guardLitPos) break;
        if (mismatch) score += d.penalty;
        else { score -= d.bonus; if (score < 0) score = 0; }
        ++src;
        dec_push(d, mismatch ? dec_code2mismatch(d, dec_ref_at(d, src), dec_lit_next(d)) : dec_ref_at(d, src));
    }
    return d.destLen - destStart;
}

// extendMatchLeft, :462-523. The reference writes the extension backwards into a scratch buffer and appends it behind the
// plain literals that follow its codes in the literal stream; here its length and the number of literal codes it will take
// are found first (plan_left_measure: the same walk, nothing written), which fixes its place in the contig, then the same
// walk writes every byte to dest[at + len - 1 - k].
struct LeftExt { int64_t srcMatch, srcGuard; uint64_t len, codes; };   // (codes = literal bytes the extension takes)
__device__ void dec_left_write(Dec &d, const LeftExt &e, uint64_t at, uint64_t markPos) {
    if (at + e.len > d.destCap) { d.bad = 1; return; }
    const bool l0 = d.solo || (threadIdx.x & (WAVE - 1)) == 0;
    uint64_t k = 0;
    int64_t src = e.srcMatch - 1;
    {
        const uint8_t c = dec_code2mismatch(d, dec_ref_at(d, src), dec_lit_next(d));
        if (l0) d.dest[at + e.len - 1 - k] = c;
        k++;
    }
    int score = d.initialScore;
    while (!d.bad && --src >= e.srcGuard && score < d.threshold) {
        const bool mismatch = dec_flag_at(d, d.flPos++) != 0;
        if (mismatch && d.litPos == markPos) break;
        if (mismatch) score += d.penalty;
        else { score -= d.bonus; if (score < 0) score = 0; }
        const uint8_t c = mismatch ? dec_code2mismatch(d, dec_ref_at(d, src), dec_lit_next(d)) : dec_ref_at(d, src);
        if (k >= e.len) { d.bad = 1; return; }
        if (l0) d.dest[at + e.len - 1 - k] = c;
        k++;
    }
    if (k != e.len) d.bad = 1;
}

// ------------------------------------------------------------------------------------------------------------------
// pass 1: the plan — decodeSequenceAndReturnUnmatchedChars (:319-432) with extendMatchLeft / extendMatchRight walked for
// their lengths only. One wave per contig, every value wave-uniform.
// ------------------------------------------------------------------------------------------------------------------
struct DecRec {                                           // one match as the plan leaves it: where, in every stream and in the contig
    uint64_t litFrom;                                     // literal stream at the top of the iteration: the left extension's codes, then the plain literals
    uint64_t mark;                                        // the match's mark (the tail record: the end of the stream)
    uint64_t destAt;                                      // contig position at the top of the iteration
    uint64_t src;                                         // matchSrcPos (after the left extension's adjustment, :462-523)
    int64_t leftSrcMatch, leftSrcGuard;                   // the left extension's walk (dec_left_write)
    uint64_t flLeft, flRight;                             // flag stream: where the two extensions' walks start
    int64_t offsetDelta;                                  // the right extension's source relative to the contig position (:434-460)
    uint64_t guardLit;                                    // ... and the literal position it must not pass
    uint32_t len, leftLen, leftCodes, rightLen;
    uint32_t flags, pad;
};
enum { REC_GAP = 1, REC_GAP_START = 2, REC_GAP_MIDDLE = 4, REC_GAP_END = 8, REC_TAIL = 16, REC_RIGHT = 32 };
struct DecPlanOut { uint64_t nrec, destLen; int64_t unmatched; uint64_t pad; };    // unmatched -1: malformed streams

// 64 bytes of a stream from `base` on, as the ballot mask of one property (a match mark / a set flag); bytes past the end: clear
struct DecByteWin { const uint8_t *p; uint64_t n, base; unsigned long long mask; };
template <bool MARKS>
__device__ __forceinline__ void bw_load(DecByteWin &w, uint64_t pos) {
    const uint64_t i = pos + (threadIdx.x & (WAVE - 1));
    const uint8_t b = i < w.n ? w.p[i] : (uint8_t) 0;
    w.base = pos;
    w.mask = __ballot(MARKS ? b == DEC_MATCH_MARK : b != 0);
}
// 256 bytes of a stream from `base` on, a dword per lane
struct DecDwWin { const uint8_t *p; uint64_t n, base; uint32_t v; };
__device__ __forceinline__ void dw_load(DecDwWin &w, uint64_t pos) {
    const uint64_t o = pos + 4ull * (threadIdx.x & (WAVE - 1));
    uint32_t x = 0;
    if (o + 4 <= w.n) x = ld_u32(w.p + o);
    else
        for (int k = 0; k < 4; k++) if (o + k < w.n) x |= (uint32_t) w.p[o + k] << (8 * k);
    w.v = x;
    w.base = pos;
}
// `width` (1, 2 or 4) bytes at pos, little endian; the caller has checked pos + width <= n
__device__ __forceinline__ uint32_t dw_get(DecDwWin &w, uint64_t pos, uint32_t width) {
    if (pos < w.base || pos + width > w.base + 4 * WAVE) dw_load(w, pos);
    const uint32_t rel = (uint32_t) (pos - w.base), q = rel >> 2, r = rel & 3u;
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) w.v, __builtin_amdgcn_readfirstlane((int) q));
    const uint32_t hi = q + 1 < (uint32_t) WAVE ? (uint32_t) __builtin_amdgcn_readlane((int) w.v, __builtin_amdgcn_readfirstlane((int) (q + 1))) : 0u;
    const uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, r);
    return width == 4 ? x : (width == 2 ? x & 0xFFFFu : x & 0xFFu);
}

struct Plan {
    DecByteWin lit, fl;
    DecDwWin len, off, off5, gap;
    uint64_t nLit, nOff, nOff5, nLen, nGap, nFlags;
    uint64_t litPos, offPos, off5Pos, lenPos, gapPos, flPos;
    uint64_t destLen, destCap;
    int bad;
    int initialScore, penalty, bonus, threshold;
};
__device__ __forceinline__ bool plan_flag_at(Plan &d, uint64_t i) {
    if (i >= d.nFlags) { d.bad = 1; return false; }
    if (i - d.fl.base >= (uint64_t) WAVE) bw_load<false>(d.fl, i);
    return (d.fl.mask >> (i - d.fl.base)) & 1ull;
}
// clear flags from position i on, as far as the window and the stream reach (0: the flag at i is set, or the stream ends there).
// The score automaton of an extension only falls over them and none of its other conditions moves: they are taken in one step.
__device__ __forceinline__ uint64_t plan_zero_run(Plan &d, uint64_t i) {
    if (i >= d.nFlags) return 0;
    if (i - d.fl.base >= (uint64_t) WAVE) bw_load<false>(d.fl, i);
    const uint64_t rel = i - d.fl.base;
    const unsigned long long m = d.fl.mask >> rel;
    const uint64_t z = m ? (uint64_t) __builtin_ctzll(m) : (uint64_t) WAVE - rel;
    return z < d.nFlags - i ? z : d.nFlags - i;
}
__device__ __forceinline__ void plan_lit_skip(Plan &d) { if (d.litPos >= d.nLit) d.bad = 1; else d.litPos++; }
__device__ __forceinline__ void plan_grow(Plan &d, uint64_t n) { if (d.destLen + n > d.destCap) d.bad = 1; else d.destLen += n; }
__device__ uint64_t plan_find_mark(Plan &d, uint64_t from) {
    while (from < d.nLit) {
        if (from - d.lit.base >= (uint64_t) WAVE) bw_load<true>(d.lit, from);
        const unsigned long long m = d.lit.mask >> (from - d.lit.base);
        if (m) return from + (uint64_t) __builtin_ctzll(m);
        from = d.lit.base + WAVE;
    }
    return DEC_NPOS;
}
// extendMatchRight, :434-460, for its length
__device__ uint64_t plan_extend_right(Plan &d, bool isGap, bool gapStart, bool gapMiddle, bool gapEnd, uint64_t guardLitPos) {
    if (d.litPos == guardLitPos && !gapMiddle) return 0;
    uint64_t n = 0;
    if (gapStart || !isGap) { plan_lit_skip(d); n++; }
    if (isGap) {
        // In a gap the score plays no part (:441): the walk passes every flag up to the mismatch that takes the last literal in
        // front of the mark — and, in a stretch that is not the gap's last, up to the set flag behind it, which finds none left
        // (:447) and is not a byte. That is a count of set flags, taken a window of 64 at a time; a gap is as long as the
        // stretch between two matches on one diagonal (megabytes in a divergent region).
        uint64_t need = (guardLitPos - d.litPos) + (gapEnd ? 0 : 1), passed = 0;
        while (need && !d.bad) {
            if (d.flPos >= d.nFlags) { d.bad = 1; break; }
            if (d.flPos - d.fl.base >= (uint64_t) WAVE) bw_load<false>(d.fl, d.flPos);
            const uint64_t rel = d.flPos - d.fl.base;
            unsigned long long m = d.fl.mask >> rel;                           // (flags past the end of the stream: clear)
            const uint64_t avail = (uint64_t) WAVE - rel < d.nFlags - d.flPos ? (uint64_t) WAVE - rel : d.nFlags - d.flPos;
            const uint64_t c = (uint64_t) __popcll(m);
            if (c < need) { need -= c; d.flPos += avail; passed += avail; }
            else {
                for (uint64_t q = 1; q < need; q++) m &= m - 1;
                const uint64_t take = (uint64_t) __builtin_ctzll(m) + 1;
                d.flPos += take; passed += take; need = 0;
            }
        }
        if (!d.bad) { d.litPos = guardLitPos; n += passed - (gapEnd ? 0 : 1); }
        plan_grow(d, n);
        return n;
    }
    int score = d.initialScore;
    while (!d.bad && (!gapEnd || d.litPos != guardLitPos) && (isGap || score < d.threshold)) {
        const uint64_t z = plan_zero_run(d, d.flPos);
        if (z) {                                                              // z matching positions: the score falls, nothing else moves
            const int64_t sc = (int64_t) score - (int64_t) d.bonus * (int64_t) z;
            score = sc < 0 ? 0 : (int) sc;
            n += z; d.flPos += z;
            continue;
        }
        const bool mismatch = plan_flag_at(d, d.flPos++);                     // (set, or past the end of the stream: malformed)
        if (mismatch && d.litPos == guardLitPos) break;
        if (mismatch) { score += d.penalty; plan_lit_skip(d); }
        else { score -= d.bonus; if (score < 0) score = 0; }
        n++;
    }
    plan_grow(d, n);
    return n;
}
// extendMatchLeft, :462-523, for its length, the literal codes it takes and the flag position behind it (dec_left_measure with
// the flags read through the window)
__device__ LeftExt plan_left_measure(Plan &d, uint64_t *matchSrcPos, bool skipOffset, uint64_t refLockPos, uint64_t markPos, uint64_t *flEnd) {
    LeftExt e; e.len = 0; e.codes = 0;
    *flEnd = d.flPos;
    int64_t srcMatch = (int64_t) *matchSrcPos;
    int64_t srcGuard = srcMatch - DEC_MAX_EXT_LEFT;
    if (!skipOffset) {
        if (srcGuard < 1) srcGuard = 1;                                       // REF_SHIFT
        const int64_t srcLock = (int64_t) refLockPos;                         // SIZE_MAX: one before the buffer, as there
        if (srcGuard < srcLock && srcLock <= srcMatch) srcGuard = srcLock;
    }
    e.srcMatch = srcMatch; e.srcGuard = srcGuard;
    if (srcGuard == srcMatch) return e;
    if (skipOffset) {                                                         // the match position was given relative to the extension's end
        int64_t src = srcMatch - 1;
        uint64_t length = 0, mismatches = 0;
        bool known = true;
        int score = d.initialScore;
        while (--src > srcGuard && score < d.threshold) {
            const bool mismatch = plan_flag_at(d, d.flPos + length++);
            if (d.bad) return e;
            if (mismatch && d.litPos + ++mismatches == markPos) { known = false; break; }
            if (mismatch) score += d.penalty;
            else { score -= d.bonus; if (score < 0) score = 0; }
        }
        if (src == srcGuard && known) { mismatches++; length++; }
        const uint64_t matchingChars = length - mismatches;
        *matchSrcPos += matchingChars;
        srcGuard += (int64_t) matchingChars;
        srcMatch += (int64_t) matchingChars;
        e.srcMatch = srcMatch; e.srcGuard = srcGuard;
    }
    if (d.litPos >= d.nLit) { d.bad = 1; return e; }
    uint64_t len = 1, codes = 1;                                              // the first byte is always a coded mismatch
    int64_t src = srcMatch - 1;
    uint64_t lp = d.litPos + 1, fp = d.flPos;
    int score = d.initialScore;
    while (--src >= srcGuard && score < d.threshold) {
        if ((int64_t) len >= DEC_MAX_EXT_LEFT) { d.bad = 1; return e; }
        {   // matching positions ahead, as far as the guard and the length limit allow: one step (this iteration and z - 1 more)
            uint64_t z = plan_zero_run(d, fp);
            const uint64_t room = (uint64_t) (src - srcGuard) + 1, cap = (uint64_t) (DEC_MAX_EXT_LEFT - (int64_t) len);
            if (z > room) z = room;
            if (z > cap) z = cap;
            if (z) {
                const int64_t sc = (int64_t) score - (int64_t) d.bonus * (int64_t) z;
                score = sc < 0 ? 0 : (int) sc;
                len += z; fp += z; src -= (int64_t) z - 1;
                continue;
            }
        }
        const bool mismatch = plan_flag_at(d, fp++);
        if (d.bad) return e;
        if (mismatch && lp == markPos) break;
        if (mismatch) { score += d.penalty; if (lp >= d.nLit) { d.bad = 1; return e; } lp++; codes++; }
        else { score -= d.bonus; if (score < 0) score = 0; }
        len++;
    }
    e.len = len; e.codes = codes;
    *flEnd = fp;
    return e;
}
// one entry of the mapLen stream, decodeMapLenStream :966-986
__device__ uint32_t plan_next_len(Plan &d, bool frugal) {
    if (!frugal) {
        if (d.lenPos + 4 > d.nLen) { d.bad = 1; return 0; }
        const uint32_t v = dw_get(d.len, d.lenPos, 4); d.lenPos += 4;
        return v;
    }
    if (d.lenPos + 2 > d.nLen) { d.bad = 1; return 0; }
    const uint32_t y16 = dw_get(d.len, d.lenPos, 2); d.lenPos += 2;
    if (y16 < 0xFFFFu) return y16;
    if (d.lenPos + 4 > d.nLen) { d.bad = 1; return 0; }
    const uint32_t y32 = dw_get(d.len, d.lenPos, 4); d.lenPos += 4;
    if (y32 < 0xFFFFFFFFu) return y32;
    if (d.lenPos + 8 > d.nLen) { d.bad = 1; return 0; }
    const uint32_t lo = dw_get(d.len, d.lenPos, 4); d.lenPos += 8;            // readUInt64Frugal<uint32_t>: the low word
    return lo;
}

__global__ void __launch_bounds__(WAVE) k_decode_plan(swsem_emit_params_t p, const DecodeJob *__restrict__ jobs, DecRec *__restrict__ recs,
                                                      const uint64_t *__restrict__ recBase, DecPlanOut *__restrict__ outs, uint64_t refBytes) {
    __shared__ int64_t paired[DEC_MAX_GAP_DEPTH];
    const DecodeJob jb = jobs[blockIdx.x];
    DecRec *out = recs + recBase[blockIdx.x];
    const uint64_t recCap = recBase[blockIdx.x + 1] - recBase[blockIdx.x];
    Plan d;
    d.lit.p = jb.stream[SWSEM_LIT]; d.lit.n = d.nLit = jb.size[SWSEM_LIT];
    d.fl.p = jb.stream[SWSEM_FLAGS]; d.fl.n = d.nFlags = jb.size[SWSEM_FLAGS];
    d.off.p = jb.stream[SWSEM_OFF]; d.off.n = d.nOff = jb.size[SWSEM_OFF];
    d.off5.p = jb.stream[SWSEM_OFF5]; d.off5.n = d.nOff5 = jb.size[SWSEM_OFF5];
    d.len.p = jb.stream[SWSEM_LEN]; d.len.n = d.nLen = jb.size[SWSEM_LEN];
    d.gap.p = jb.stream[SWSEM_GAP]; d.gap.n = d.nGap = jb.size[SWSEM_GAP];
    d.litPos = d.offPos = d.off5Pos = d.lenPos = d.gapPos = d.flPos = 0;
    d.destLen = 0; d.destCap = jb.destCap;
    d.bad = 0;
    d.initialScore = p.mmsMismatchesInitialScore; d.penalty = p.mmsMismatchPenalty; d.bonus = p.mmsMatchBonus; d.threshold = p.mmsMismatchesScoreThreshold;
    bw_load<true>(d.lit, 0); bw_load<false>(d.fl, 0);
    dw_load(d.len, 0); dw_load(d.off, 0); dw_load(d.off5, 0); dw_load(d.gap, 0);
    for (int i = threadIdx.x; i < DEC_MAX_GAP_DEPTH; i += WAVE) paired[i] = INT64_MAX;
    __builtin_amdgcn_s_waitcnt(0);
    const bool l0 = threadIdx.x == 0;
    const uint64_t seqEnd = d.nLit;
    uint32_t unmatchedChars = 0;
    int64_t gapStartIdx = -1, gapEndIdx = -1;
    int gapCurIdx = 0;
    uint64_t matchSrcPos = 0, prevMatchDestPos = 0;
    int64_t offsetDelta = -1;
    uint64_t extLeftLen = 0, extRightLen = 0;
    bool isGap = false;
    int64_t j = 0;
    uint64_t markPos = plan_find_mark(d, d.litPos);
    while (!d.bad && markPos != DEC_NPOS && markPos < seqEnd) {
        if ((uint64_t) j + 1 >= recCap) { d.bad = 1; break; }                 // (more marks than mapLen entries: malformed)
        DecRec rec;
        rec.litFrom = d.litPos; rec.mark = markPos; rec.destAt = d.destLen; rec.flLeft = d.flPos;
        rec.leftLen = 0; rec.leftCodes = 0; rec.leftSrcMatch = 0; rec.leftSrcGuard = 0; rec.pad = 0;
        const uint64_t literalsLeft = markPos - d.litPos;
        matchSrcPos = 0;
        const int64_t pv = paired[gapCurIdx];
        const bool skipOffset = pv != INT64_MAX;
        if (skipOffset) {
            matchSrcPos = (uint64_t) (pv + (int64_t) d.destLen + (int64_t) literalsLeft);
            if (l0) paired[gapCurIdx] = INT64_MAX;
        } else {
            if (d.offPos + 4 > d.nOff) { d.bad = 1; break; }
            matchSrcPos = dw_get(d.off, d.offPos, 4); d.offPos += 4;
            if (p.enable40bitReference) {                                     // :356-359
                if (d.off5Pos >= d.nOff5) { d.bad = 1; break; }
                matchSrcPos += (uint64_t) dw_get(d.off5, d.off5Pos, 1) << 32; d.off5Pos++;
            }
        }
        extLeftLen = 0;
        if (p.enableExtensionsWithMismatches) {
            if (!isGap && literalsLeft) {
                uint64_t flEnd = d.flPos;
                const LeftExt e = plan_left_measure(d, &matchSrcPos, skipOffset, jb.refLockPos, markPos, &flEnd);
                if (!d.bad && e.len) {
                    if (e.codes > literalsLeft) { d.bad = 1; break; }
                    rec.leftLen = (uint32_t) e.len; rec.leftCodes = (uint32_t) e.codes; rec.leftSrcMatch = e.srcMatch; rec.leftSrcGuard = e.srcGuard;
                    d.litPos += e.codes; d.flPos = flEnd;                     // what dec_left_write consumes
                    extLeftLen = e.len;
                }
            }
            if (gapEndIdx == j) { gapStartIdx = -1; gapEndIdx = -1; }
        }
        if (d.bad) break;
        const uint64_t literalLen = markPos - d.litPos + extLeftLen + extRightLen;
        plan_grow(d, markPos - d.litPos);                                     // the plain literals ...
        plan_grow(d, extLeftLen);                                             // ... and, behind them, the left extension
        if (d.bad) break;
        unmatchedChars += (uint32_t) literalLen;
        d.litPos = markPos + 1;
        const uint32_t matchLength = plan_next_len(d, p.frugal64bitLenEncoding != 0);
        if (d.bad) break;
        prevMatchDestPos = d.destLen;
        plan_grow(d, matchLength);
        if (matchSrcPos > refBytes || matchLength > refBytes - matchSrcPos) d.bad = 1;   // (a match outside the reference buffer: malformed)
        if (d.bad) break;
        rec.src = matchSrcPos; rec.len = matchLength;
        markPos = plan_find_mark(d, d.litPos);
        uint32_t gapDelta = 0;
        if (p.gapDepthOffsetEncoding && markPos != DEC_NPOS && markPos < seqEnd) {
            if (d.gapPos >= d.nGap) { d.bad = 1; break; }
            gapDelta = dw_get(d.gap, d.gapPos, 1); d.gapPos++;
        }
        if (gapDelta) {
            int gapIdx = gapCurIdx;
            int g = (int) gapDelta;
            if (!p.lazyDecompressionSupport && gapStartIdx == -1 && markPos - d.litPos == 0) {
                gapIdx = (gapIdx + 1) % DEC_MAX_GAP_DEPTH;
                g++;
            }
            int guard = 0;
            while (gapDelta) {
                gapIdx = (gapIdx + 1) % DEC_MAX_GAP_DEPTH;
                if (paired[gapIdx] == INT64_MAX) gapDelta--;
                else g++;
                if (++guard > 4 * DEC_MAX_GAP_DEPTH) { d.bad = 1; break; }    // (a ring without free slots: malformed stream)
            }
            if (d.bad) break;
            if (l0) paired[gapIdx] = (int64_t) matchSrcPos - (int64_t) prevMatchDestPos;
            if (p.enableExtensionsWithMismatches && gapEndIdx <= j + g && g <= p.gapDepthMismatchesEncoding) {
                gapStartIdx = j;
                gapEndIdx = j + g;
            }
        }
        gapCurIdx = (gapCurIdx + 1) % DEC_MAX_GAP_DEPTH;
        const bool gapStart = gapStartIdx == j;
        const bool gapEnd = gapEndIdx == j + 1;
        const bool gapMiddle = gapStartIdx < j && j + 1 < gapEndIdx;
        isGap = gapStart || gapMiddle || gapEnd;
        extRightLen = 0;
        rec.flags = (isGap ? REC_GAP : 0) | (gapStart ? REC_GAP_START : 0) | (gapMiddle ? REC_GAP_MIDDLE : 0) | (gapEnd ? REC_GAP_END : 0);
        rec.flRight = d.flPos; rec.offsetDelta = 0; rec.guardLit = 0; rec.rightLen = 0;
        if (p.enableExtensionsWithMismatches) {
            if (!isGap || gapStart) offsetDelta = (int64_t) matchSrcPos + (int64_t) matchLength - (int64_t) d.destLen;
            const uint64_t guardLit = markPos != DEC_NPOS && markPos < seqEnd ? markPos : seqEnd;
            rec.flags |= REC_RIGHT; rec.offsetDelta = offsetDelta; rec.guardLit = guardLit;
            extRightLen = plan_extend_right(d, isGap, gapStart, gapMiddle, gapEnd, guardLit);
            rec.rightLen = (uint32_t) extRightLen;
        }
        if (l0) out[j] = rec;
        j++;
    }
    if (!d.bad) {                                                             // the literals behind the last match
        const uint64_t literalLen = seqEnd - d.litPos + extRightLen;
        DecRec rec = {};
        rec.litFrom = d.litPos; rec.mark = seqEnd; rec.destAt = d.destLen; rec.flags = REC_TAIL;
        plan_grow(d, seqEnd - d.litPos);
        unmatchedChars += (uint32_t) literalLen;
        d.litPos = seqEnd;
        if (l0 && !d.bad) out[j] = rec;
        j++;
    }
    if (!d.bad && (d.offPos != d.nOff || d.off5Pos != d.nOff5 || d.lenPos != d.nLen || d.gapPos != d.nGap || d.flPos != d.nFlags)) d.bad = 1;
    if (l0) {
        DecPlanOut o;
        o.nrec = d.bad ? 0 : (uint64_t) j; o.destLen = d.destLen; o.unmatched = d.bad ? -1 : (int64_t) unmatchedChars; o.pad = 0;
        outs[blockIdx.x] = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// pass 2: the bytes — one thread per record
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_decode_fill(const uint8_t *__restrict__ ref, swsem_emit_params_t p, const DecodeJob *__restrict__ jobs,
                                                     const DecRec *__restrict__ recs, const uint64_t *__restrict__ recBase,
                                                     const DecPlanOut *__restrict__ plans, uint32_t *__restrict__ badFlags, uint64_t refBytes, uint32_t c0) {
    const uint32_t c = c0 + blockIdx.y;                                     // (grid.y holds at most 65 535: the host launches slices of contigs)
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (plans[c].unmatched < 0 || (uint64_t) blockIdx.x * blockDim.x >= plans[c].nrec) return;
    // (a lane without a record of its own stays: the long pieces below are the work of all 64 lanes of a wave)
    DecRec r = {};
    r.flags = REC_TAIL;
    if (i < plans[c].nrec) r = recs[recBase[c] + i];
    const DecodeJob &jb = jobs[c];
    Dec d;
    d.ref = ref; d.solo = true; d.refBytes = refBytes;
    d.lit = jb.stream[SWSEM_LIT]; d.nLit = jb.size[SWSEM_LIT];
    d.flags = jb.stream[SWSEM_FLAGS]; d.nFlags = jb.size[SWSEM_FLAGS];
    d.off = nullptr; d.off5 = nullptr; d.len = nullptr; d.gap = nullptr; d.nOff = d.nOff5 = d.nLen = d.nGap = 0;
    d.litPos = r.litFrom; d.flPos = r.flLeft; d.offPos = d.off5Pos = d.lenPos = d.gapPos = 0;
    d.dest = jb.dest; d.destLen = r.destAt; d.destCap = jb.destCap;
    d.bad = 0;
    d.initialScore = p.mmsMismatchesInitialScore; d.penalty = p.mmsMismatchPenalty; d.bonus = p.mmsMatchBonus; d.threshold = p.mmsMismatchesScoreThreshold;
    const uint64_t plain = (r.mark - r.litFrom) - r.leftCodes;
    if (r.leftLen) {                                                          // (its codes come first in the literal stream, its bytes behind the plain literals)
        LeftExt e; e.srcMatch = r.leftSrcMatch; e.srcGuard = r.leftSrcGuard; e.len = r.leftLen; e.codes = r.leftCodes;
        dec_left_write(d, e, r.destAt + plain, r.mark);
    }
    // what is long goes to the whole wave (below): a run of plain literals, the match itself, the right extension across a gap
    uint64_t copyLen[2] = {0, 0};
    const uint8_t *copySrc[2] = {nullptr, nullptr};
    uint8_t *copyDst[2] = {nullptr, nullptr};
    bool wideRight = false;
    {
        const uint8_t *s = d.lit + r.litFrom + r.leftCodes;
        uint8_t *t = d.dest + r.destAt;
        if (plain <= DEC_WIDE_MIN) for (uint64_t k = 0; k < plain; k++) t[k] = s[k];
        else { copyLen[0] = plain; copySrc[0] = s; copyDst[0] = t; }
    }
    const uint64_t at = r.destAt + plain + r.leftLen;
    if (!(r.flags & REC_TAIL)) {
        const uint8_t *s = ref + r.src;
        uint8_t *t = d.dest + at;
        if (r.len <= DEC_WIDE_MIN) {
            uint64_t k = 0;
            for (; k + 16 <= r.len; k += 16) { const uint4 x = ld_u128(s + k); __builtin_memcpy(t + k, &x, 16); }
            for (; k < r.len; k++) t[k] = s[k];
        } else { copyLen[1] = r.len; copySrc[1] = s; copyDst[1] = t; }
        if (r.flags & REC_RIGHT) {
            wideRight = (r.flags & REC_GAP) != 0 && r.rightLen > DEC_WIDE_MIN;
            if (!wideRight) {
                d.litPos = r.mark + 1; d.flPos = r.flRight; d.destLen = at + r.len;
                const uint64_t n = dec_extend_right(d, r.offsetDelta, (r.flags & REC_GAP) != 0, (r.flags & REC_GAP_START) != 0, (r.flags & REC_GAP_MIDDLE) != 0,
                                                    (r.flags & REC_GAP_END) != 0, r.guardLit);
                if (n != r.rightLen) d.bad = 1;
            }
        }
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int w = 0; w < 2; w++)
        for (unsigned long long todo = __ballot(copyLen[w] != 0); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const uint64_t len = rl64d(copyLen[w], l);
            const uint8_t *src = (const uint8_t *) rl64d((uint64_t) copySrc[w], l);
            uint8_t *dst = (uint8_t *) rl64d((uint64_t) copyDst[w], l);
            for (uint64_t o = 16 * (uint64_t) lane; o < len; o += 16 * WAVE) {
                if (o + 16 <= len) { const uint4 x = ld_u128(src + o); __builtin_memcpy(dst + o, &x, 16); }
                else for (uint64_t k = o; k < len; k++) dst[k] = src[k];
            }
        }
    for (unsigned long long todo = __ballot(wideRight); todo; todo &= todo - 1) {
        // extendMatchRight across a gap (:434-460) by the wave: byte i of the extension is the reference byte at its place or,
        // where its flag is set, the mismatch the next literal code stands for — the code's place is the count of set flags
        // in front of it
        const int l = __builtin_ctzll(todo);
        const uint64_t n = rl64d(r.rightLen, l), dst0 = rl64d(at + r.len, l), fl0 = rl64d(r.flRight, l), lit0 = rl64d(r.mark + 1, l);
        const int64_t src0 = (int64_t) rl64d((uint64_t) ((int64_t) (at + r.len) + r.offsetDelta), l);
        const uint32_t s0 = (rl64d(r.flags, l) & REC_GAP_START) ? 1u : 0u;
        int bad = dst0 + n > d.destCap;
        uint64_t rank = s0;                                                    // literal codes taken so far
        if (!bad && s0 && lane == 0) {
            Dec t = d;
            t.bad = 0; t.litPos = lit0;
            d.dest[dst0] = dec_code2mismatch(t, dec_ref_at(t, src0), dec_lit_next(t));
            bad |= t.bad;
        }
        bad = __ballot(bad != 0) != 0;
        for (uint64_t i0 = s0; i0 < n && !bad; i0 += WAVE) {
            const uint64_t i = i0 + lane;
            const bool live = i < n;
            Dec t = d;
            t.bad = 0;
            const uint8_t f = live ? dec_flag_at(t, fl0 + (i - s0)) : (uint8_t) 0;
            const unsigned long long set = __ballot(live && f != 0);
            if (live) {
                const uint8_t rb = dec_ref_at(t, src0 + (int64_t) i);
                uint8_t b = rb;
                if (f) {
                    t.litPos = lit0 + rank + (uint64_t) __popcll(set & ((1ull << lane) - 1ull));
                    b = dec_code2mismatch(t, rb, dec_lit_next(t));
                }
                d.dest[dst0 + i] = b;
            }
            rank += (uint64_t) __popcll(set);
            bad |= __ballot(t.bad != 0) != 0;
        }
        bad = __ballot(bad != 0) != 0;
        if (bad && (int) lane == l) d.bad = 1;
    }
    if (d.bad) atomicOr(&badFlags[c], 1u);
}

// the contig the streams gave back against the one that was encoded: first differing byte per contig (DEC_NPOS: equal)
__global__ void __launch_bounds__(256) k_decode_check(const DecodeJob *__restrict__ jobs, const DecPlanOut *__restrict__ plans, unsigned long long *__restrict__ firstDiff, uint32_t c0) {
    const uint32_t c = c0 + blockIdx.y;
    const DecodeJob &jb = jobs[c];
    if (!jb.expect || plans[c].unmatched < 0) return;
    const uint64_t n = plans[c].destLen;
    const uint64_t i0 = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= n) return;
    const uint64_t i1 = i0 + 16 < n ? i0 + 16 : n;
    for (uint64_t i = i0; i < i1; i++)
        if (jb.dest[i] != jb.expect[i]) { atomicMin(&firstDiff[c], (unsigned long long) i); break; }
}

}  // namespace swk
