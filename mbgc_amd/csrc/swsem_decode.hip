// The decoder's per-contig automaton on the device: the exact inverse of processMatches —
// MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars (mbgccoder/MBGC_Decoder.cpp:319-432), extendMatchRight (:434-460),
// extendMatchLeft (:462-523), one entry of the mapLen stream (decodeMapLenStream :966-986 with readUInt64Frugal,
// utils/helper.h:256-272) and ContextAwareMismatchesCoder::code2mismatch (coders/ContextAwareMismatchesCoder.cpp:8-17,72-77).
//
// The six streams of a contig are consumed in lock-step and carry no synchronisation points (how many flag bytes an
// extension reads depends on the flags themselves), so a contig is one sequential chain: one wave per contig, the chain's
// values wave-uniform, the lanes sharing what can be shared — the search for the next match mark (64 literal bytes per
// step) and every copy (plain literals, the match bytes out of the reference buffer). The contigs of a round are
// independent (each stands against the reference as its lock position froze it) and are decoded side by side. Used as the
// device-side check of an emission (swsem_emit_verify: what was just emitted must decode to the query, with no encoder
// logic in the loop) and as the building block of an accelerated `mbgc d`.
#include "swsem_device.h"
#include "../../include/mbgc_swsem.h"

namespace swk {

constexpr uint8_t DEC_MATCH_MARK = 0xA5;                  // MBGC_Params.h:45
constexpr int DEC_MAX_GAP_DEPTH = 128;                    // MBGC_Params.h:50
constexpr int64_t DEC_MAX_EXT_LEFT = 1 << 24;             // MAX_EXTEND_MATCH_LEFT_LENGTH, MBGC_Params.h:55
constexpr uint64_t DEC_NPOS = UINT64_MAX;

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
__device__ __forceinline__ uint8_t dec_ref_at(Dec &d, int64_t i) { if (i < 0) { d.bad = 1; return 0; } return d.ref[i]; }
__device__ __forceinline__ void dec_push(Dec &d, uint8_t c) {
    if (d.destLen >= d.destCap) { d.bad = 1; return; }
    if ((threadIdx.x & (WAVE - 1)) == 0) d.dest[d.destLen] = c;
    d.destLen++;
}
// all lanes copy; source and destination never overlap (the reference buffer / the literal stream -> the contig)
__device__ void dec_append(Dec &d, const uint8_t *s, uint64_t n) {
    if (d.destLen + n > d.destCap) { d.bad = 1; return; }
    uint8_t *t = d.dest + d.destLen;
    for (uint64_t i = threadIdx.x & (WAVE - 1); i < n; i += WAVE) t[i] = s[i];
    d.destLen += n;
}
// memchr(lit + from, MATCH_MARK, nLit - from), 64 bytes per step
__device__ uint64_t dec_find_mark(const Dec &d, uint64_t from) {
    const uint64_t lane = threadIdx.x & (WAVE - 1);
    for (uint64_t base = from; base < d.nLit; base += WAVE) {
        const uint64_t i = base + lane;
        const unsigned long long m = __ballot(i < d.nLit && d.lit[i] == DEC_MATCH_MARK);
        if (m) return base + (uint64_t) __builtin_ctzll(m);
    }
    return DEC_NPOS;
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
// are found first (the same walk, nothing consumed or written), which fixes its place in the contig, then the same walk
// writes every byte to dest[at + len - 1 - k].
struct LeftExt { int64_t srcMatch, srcGuard; uint64_t len, codes; };
__device__ LeftExt dec_left_measure(Dec &d, uint64_t *matchSrcPos, bool skipOffset, uint64_t refLockPos, uint64_t markPos) {
    LeftExt e; e.len = 0; e.codes = 0;
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
            const bool mismatch = dec_flag_at(d, d.flPos + length++) != 0;
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
        if (fp >= d.nFlags) { d.bad = 1; return e; }
        const bool mismatch = d.flags[fp++] != 0;
        if (mismatch && lp == markPos) break;
        if (mismatch) { score += d.penalty; if (lp >= d.nLit) { d.bad = 1; return e; } lp++; codes++; }
        else { score -= d.bonus; if (score < 0) score = 0; }
        len++;
    }
    e.len = len; e.codes = codes;
    return e;
}
__device__ void dec_left_write(Dec &d, const LeftExt &e, uint64_t at, uint64_t markPos) {
    if (at + e.len > d.destCap) { d.bad = 1; return; }
    const bool l0 = (threadIdx.x & (WAVE - 1)) == 0;
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

// one entry of the mapLen stream, decodeMapLenStream :966-986
__device__ uint32_t dec_next_len(Dec &d, bool frugal) {
    if (!frugal) {
        if (d.lenPos + 4 > d.nLen) { d.bad = 1; return 0; }
        const uint32_t v = ld_u32(d.len + d.lenPos); d.lenPos += 4;
        return v;
    }
    if (d.lenPos + 2 > d.nLen) { d.bad = 1; return 0; }
    const uint32_t y16 = (uint32_t) d.len[d.lenPos] | ((uint32_t) d.len[d.lenPos + 1] << 8); d.lenPos += 2;
    if (y16 < 0xFFFFu) return y16;
    if (d.lenPos + 4 > d.nLen) { d.bad = 1; return 0; }
    const uint32_t y32 = ld_u32(d.len + d.lenPos); d.lenPos += 4;
    if (y32 < 0xFFFFFFFFu) return y32;
    if (d.lenPos + 8 > d.nLen) { d.bad = 1; return 0; }
    const uint32_t lo = ld_u32(d.len + d.lenPos); d.lenPos += 8;             // readUInt64Frugal<uint32_t>: the low word
    return lo;
}

// decodeSequenceAndReturnUnmatchedChars, :319-432: one wave per contig
__global__ void __launch_bounds__(WAVE) k_decode_contigs(const uint8_t *__restrict__ ref, swsem_emit_params_t p, const DecodeJob *__restrict__ jobs,
                                                         DecodeOut *__restrict__ outs) {
    __shared__ int64_t paired[DEC_MAX_GAP_DEPTH];
    const DecodeJob jb = jobs[blockIdx.x];
    Dec d;
    d.ref = ref;
    d.lit = jb.stream[SWSEM_LIT]; d.nLit = jb.size[SWSEM_LIT];
    d.off = jb.stream[SWSEM_OFF]; d.nOff = jb.size[SWSEM_OFF];
    d.off5 = jb.stream[SWSEM_OFF5]; d.nOff5 = jb.size[SWSEM_OFF5];
    d.len = jb.stream[SWSEM_LEN]; d.nLen = jb.size[SWSEM_LEN];
    d.gap = jb.stream[SWSEM_GAP]; d.nGap = jb.size[SWSEM_GAP];
    d.flags = jb.stream[SWSEM_FLAGS]; d.nFlags = jb.size[SWSEM_FLAGS];
    d.litPos = d.offPos = d.off5Pos = d.lenPos = d.gapPos = d.flPos = 0;
    d.dest = jb.dest; d.destLen = 0; d.destCap = jb.destCap;
    d.bad = 0;
    d.initialScore = p.mmsMismatchesInitialScore; d.penalty = p.mmsMismatchPenalty; d.bonus = p.mmsMatchBonus; d.threshold = p.mmsMismatchesScoreThreshold;
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
    uint64_t markPos = dec_find_mark(d, d.litPos);
    while (!d.bad && markPos != DEC_NPOS && markPos < seqEnd) {
        const uint64_t literalsLeft = markPos - d.litPos;
        matchSrcPos = 0;
        const int64_t pv = paired[gapCurIdx];
        const bool skipOffset = pv != INT64_MAX;
        if (skipOffset) {
            matchSrcPos = (uint64_t) (pv + (int64_t) d.destLen + (int64_t) literalsLeft);
            if (l0) paired[gapCurIdx] = INT64_MAX;
        } else {
            if (d.offPos + 4 > d.nOff) { d.bad = 1; break; }
            matchSrcPos = ld_u32(d.off + d.offPos); d.offPos += 4;
            if (p.enable40bitReference) {                                     // :356-359
                if (d.off5Pos >= d.nOff5) { d.bad = 1; break; }
                matchSrcPos += (uint64_t) d.off5[d.off5Pos++] << 32;
            }
        }
        extLeftLen = 0;
        if (p.enableExtensionsWithMismatches) {
            if (!isGap && literalsLeft) {
                const LeftExt e = dec_left_measure(d, &matchSrcPos, skipOffset, jb.refLockPos, markPos);
                if (!d.bad && e.len) {
                    dec_left_write(d, e, d.destLen + (literalsLeft - e.codes), markPos);    // behind the plain literals its codes leave over
                    extLeftLen = e.len;
                }
            }
            if (gapEndIdx == j) { gapStartIdx = -1; gapEndIdx = -1; }
        }
        if (d.bad) break;
        const uint64_t literalLen = markPos - d.litPos + extLeftLen + extRightLen;
        dec_append(d, d.lit + d.litPos, markPos - d.litPos);                  // the plain literals ...
        if (d.destLen + extLeftLen > d.destCap) { d.bad = 1; break; }
        d.destLen += extLeftLen;                                              // ... and, already in place behind them, the left extension
        unmatchedChars += (uint32_t) literalLen;
        d.litPos = markPos + 1;
        const uint32_t matchLength = dec_next_len(d, p.frugal64bitLenEncoding != 0);
        if (d.bad) break;
        prevMatchDestPos = d.destLen;
        dec_append(d, ref + matchSrcPos, matchLength);
        markPos = dec_find_mark(d, d.litPos);
        uint32_t gapDelta = 0;
        if (p.gapDepthOffsetEncoding && markPos != DEC_NPOS && markPos < seqEnd) {
            if (d.gapPos >= d.nGap) { d.bad = 1; break; }
            gapDelta = d.gap[d.gapPos++];
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
        if (p.enableExtensionsWithMismatches) {
            if (!isGap || gapStart) offsetDelta = (int64_t) matchSrcPos + (int64_t) matchLength - (int64_t) d.destLen;
            extRightLen = dec_extend_right(d, offsetDelta, isGap, gapStart, gapMiddle, gapEnd,
                                           markPos != DEC_NPOS && markPos < seqEnd ? markPos : seqEnd);
        }
        j++;
    }
    if (!d.bad) {
        const uint64_t literalLen = seqEnd - d.litPos + extRightLen;
        dec_append(d, d.lit + d.litPos, seqEnd - d.litPos);
        unmatchedChars += (uint32_t) literalLen;
        d.litPos = seqEnd;
    }
    if (!d.bad && (d.offPos != d.nOff || d.off5Pos != d.nOff5 || d.lenPos != d.nLen || d.gapPos != d.nGap || d.flPos != d.nFlags)) d.bad = 1;
    // the device-side check: the contig the streams gave back against the one that was encoded
    uint64_t firstDiff = DEC_NPOS;
    if (jb.expect && !d.bad) {
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t lane = threadIdx.x;
        for (uint64_t base = 0; base < d.destLen; base += WAVE) {
            const uint64_t i = base + lane;
            const unsigned long long m = __ballot(i < d.destLen && d.dest[i] != jb.expect[i]);
            if (m) { firstDiff = base + (uint64_t) __builtin_ctzll(m); break; }
        }
    }
    if (l0) {
        DecodeOut o;
        o.destLen = d.destLen; o.unmatched = d.bad ? -1 : (int64_t) unmatchedChars; o.firstDiff = firstDiff;
        outs[blockIdx.x] = o;
    }
}

}  // namespace swk
