/* TEST INFRASTRUCTURE ONLY — CPU restatement (plain C) of the decoder's per-contig automaton, the exact inverse
 * of processMatches: MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars (mbgccoder/MBGC_Decoder.cpp:319-432),
 * extendMatchRight (:434-460), extendMatchLeft (:462-523), the mapLen stream (decodeMapLenStream :966-986 with
 * PgHelpers::readUInt64Frugal, utils/helper.h:256-272) and ContextAwareMismatchesCoder::code2mismatch
 * (coders/ContextAwareMismatchesCoder.cpp:8-17,72-77).
 *
 * It is used by the tests for the size-independent round-trip property: the six streams of a contig, decoded
 * against the reference buffer the encoder matched against, give the contig back. Parity status: PINNED through
 * that property — it must hold for streams produced by the reference's own encoder (oracle/_ref, in
 * tests/test_oracle_vs_ref.py) and for the golden stream fixtures under tests/golden/.
 *
 * The six streams are those of ONE contig (what processMatches appends); the SEQ_SEPARATOR_MARK the encoder's
 * caller puts behind a contig's literals is the end of the literal buffer here.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define MATCH_MARK 0xA5                        /* '%'+128, MBGC_Params.h:45 */
#define MAX_GAP_DEPTH 128                      /* MBGC_Params.h:50 */
#define MAX_EXTEND_MATCH_LEFT_LENGTH (1 << 24) /* MBGC_Params.h:55 */
#define REF_SHIFT 1
#define NPOS UINT64_MAX
#define NO_GAP (-1)

typedef struct {
    const uint8_t *ref;
    const orc_emit_params *p;
    const uint8_t *lit, *off, *off5, *len, *gap, *flags;
    uint64_t nLit, nOff, nOff5, nLen, nGap, nFlags;
    uint64_t litPos, offPos, off5Pos, lenPos, gapPos, flPos;
    uint8_t *dest;
    uint64_t destLen, destCap;
    int bad;                                   /* a stream ran out / an index left its buffer */
} dec_t;

/* code2mis[i][mis2code[i][j]] = val2sym[j], ContextAwareMismatchesCoder.cpp:8-17 */
static int sym5(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : -1; }
static uint8_t code2mismatch(dec_t *d, uint8_t actual, uint8_t code) {   /* :72-77 */
    static const int8_t mis2code[5][5] = {{-1, 2, 0, 1, 3}, {1, -1, 2, 0, 3}, {0, 2, -1, 1, 3},
                                          {1, 0, 2, -1, 3}, {1, 2, 3, 0, -1}};
    if (code >= 5) return code;
    const int a = sym5(actual);
    if (a < 0) { d->bad = 1; return 0; }       /* (the reference indexes its table out of range here) */
    for (int j = 0; j < 5; j++)
        if (mis2code[a][j] == (int8_t) code) return (uint8_t) "ACGTN"[j];
    d->bad = 1;
    return 0;
}

static uint8_t lit_next(dec_t *d) { if (d->litPos >= d->nLit) { d->bad = 1; return 0; } return d->lit[d->litPos++]; }
static uint8_t flag_at(dec_t *d, uint64_t i) { if (i >= d->nFlags) { d->bad = 1; return 0; } return d->flags[i]; }
static uint8_t ref_at(dec_t *d, int64_t i) { if (i < 0) { d->bad = 1; return 0; } return d->ref[i]; }
static void push(dec_t *d, uint8_t c) { if (d->destLen >= d->destCap) { d->bad = 1; return; } d->dest[d->destLen++] = c; }
static void append(dec_t *d, const uint8_t *s, uint64_t n) {
    if (d->destLen + n > d->destCap) { d->bad = 1; return; }
    memcpy(d->dest + d->destLen, s, n);
    d->destLen += n;
}
static uint64_t find_mark(const dec_t *d, uint64_t from) {
    if (from >= d->nLit) return NPOS;
    const uint8_t *q = memchr(d->lit + from, MATCH_MARK, d->nLit - from);
    return q ? (uint64_t) (q - d->lit) : NPOS;
}

/* extendMatchRight, :434-460 */
static uint64_t extend_right(dec_t *d, int64_t offsetDelta, int isGap, int gapStart, int gapMiddle, int gapEnd, uint64_t guardLitPos) {
    const orc_emit_params *p = d->p;
    if (d->litPos == guardLitPos && !gapMiddle) return 0;
    const uint64_t destStart = d->destLen;
    int64_t src = (int64_t) d->destLen + offsetDelta;
    if (gapStart || !isGap) push(d, code2mismatch(d, ref_at(d, src), lit_next(d)));
    else src--;
    int score = p->mmsMismatchesInitialScore;
    while (!d->bad && (!gapEnd || d->litPos != guardLitPos) && (isGap || score < p->mmsMismatchesScoreThreshold)) {
        const int mismatch = flag_at(d, d->flPos++) != 0;
        if (mismatch && d->litPos == guardLitPos) break;
        if (mismatch) score += p->mmsMismatchPenalty;
        else { score -= p->mmsMatchBonus; if (score < 0) score = 0; }
        ++src;
        push(d, mismatch ? code2mismatch(d, ref_at(d, src), lit_next(d)) : ref_at(d, src));
    }
    return d->destLen - destStart;
}

/* extendMatchLeft, :462-523. Writes the extension backwards from extEnd; returns its length. */
static uint64_t extend_left(dec_t *d, uint8_t *extEnd, uint64_t *matchSrcPos, int skipOffset, uint64_t refLockPos, uint64_t markPos) {
    const orc_emit_params *p = d->p;
    uint8_t *ptr = extEnd;
    int64_t srcMatch = (int64_t) *matchSrcPos;
    int64_t srcGuard = srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH;
    if (!skipOffset) {
        if (srcGuard < REF_SHIFT) srcGuard = REF_SHIFT;
        const int64_t srcLock = (int64_t) refLockPos;                 /* SIZE_MAX: one before the buffer, as there */
        if (srcGuard < srcLock && srcLock <= srcMatch) srcGuard = srcLock;
    }
    if (srcGuard == srcMatch) return 0;
    if (skipOffset) {                                                 /* the match position was given relative to the extension's end */
        int64_t src = srcMatch - 1;
        uint64_t length = 0, mismatches = 0;
        int known = 1;
        int score = p->mmsMismatchesInitialScore;
        while (--src > srcGuard && score < p->mmsMismatchesScoreThreshold) {
            const int mismatch = flag_at(d, d->flPos + length++) != 0;
            if (d->bad) return 0;
            if (mismatch && d->litPos + ++mismatches == markPos) { known = 0; break; }
            if (mismatch) score += p->mmsMismatchPenalty;
            else { score -= p->mmsMatchBonus; if (score < 0) score = 0; }
        }
        if (src == srcGuard && known) { mismatches++; length++; }
        const uint64_t matchingChars = length - mismatches;
        *matchSrcPos += matchingChars;
        srcGuard += (int64_t) matchingChars;
        srcMatch += (int64_t) matchingChars;
    }
    int64_t src = srcMatch - 1;
    *(--ptr) = code2mismatch(d, ref_at(d, src), lit_next(d));
    int score = p->mmsMismatchesInitialScore;
    while (!d->bad && --src >= srcGuard && score < p->mmsMismatchesScoreThreshold) {
        if ((uint64_t) (extEnd - ptr) >= MAX_EXTEND_MATCH_LEFT_LENGTH) { d->bad = 1; break; }
        const int mismatch = flag_at(d, d->flPos++) != 0;
        if (mismatch && d->litPos == markPos) break;
        if (mismatch) score += p->mmsMismatchPenalty;
        else { score -= p->mmsMatchBonus; if (score < 0) score = 0; }
        *(--ptr) = mismatch ? code2mismatch(d, ref_at(d, src), lit_next(d)) : ref_at(d, src);
    }
    return (uint64_t) (extEnd - ptr);
}

/* one entry of the mapLen stream, decodeMapLenStream :966-986 */
static uint32_t next_len(dec_t *d) {
    if (!d->p->frugal64bitLenEncoding) {
        uint32_t v = 0;
        if (d->lenPos + 4 > d->nLen) { d->bad = 1; return 0; }
        memcpy(&v, d->len + d->lenPos, 4); d->lenPos += 4;
        return v;
    }
    uint16_t y16 = 0;
    if (d->lenPos + 2 > d->nLen) { d->bad = 1; return 0; }
    memcpy(&y16, d->len + d->lenPos, 2); d->lenPos += 2;
    if (y16 < UINT16_MAX) return y16;
    uint32_t y32 = 0;
    if (d->lenPos + 4 > d->nLen) { d->bad = 1; return 0; }
    memcpy(&y32, d->len + d->lenPos, 4); d->lenPos += 4;
    if (y32 < UINT32_MAX) return y32;
    uint64_t y64 = 0;
    if (d->lenPos + 8 > d->nLen) { d->bad = 1; return 0; }
    memcpy(&y64, d->len + d->lenPos, 8); d->lenPos += 8;
    return (uint32_t) y64;                                            /* readUInt64Frugal<uint32_t> */
}

/* decodeSequenceAndReturnUnmatchedChars, :319-432. streams[] in the ORC_* order. Returns unmatchedChars, or -1
 * when a stream ran out, an index left its buffer or not every stream byte was consumed. */
int64_t orc_decode_contig(const uint8_t *ref, const orc_emit_params *p, const uint8_t *const streams[ORC_NSTREAMS],
                          const uint64_t sizes[ORC_NSTREAMS], uint64_t refLockPos, uint8_t *dest, uint64_t destCap,
                          uint64_t *destLen) {
    dec_t D;
    memset(&D, 0, sizeof D);
    dec_t *d = &D;
    d->ref = ref; d->p = p;
    d->lit = streams[ORC_LIT]; d->nLit = sizes[ORC_LIT];
    d->off = streams[ORC_OFF]; d->nOff = sizes[ORC_OFF];
    d->off5 = streams[ORC_OFF5]; d->nOff5 = sizes[ORC_OFF5];
    d->len = streams[ORC_LEN]; d->nLen = sizes[ORC_LEN];
    d->gap = streams[ORC_GAP]; d->nGap = sizes[ORC_GAP];
    d->flags = streams[ORC_FLAGS]; d->nFlags = sizes[ORC_FLAGS];
    d->dest = dest; d->destCap = destCap;
    uint8_t *ext = p->enableExtensionsWithMismatches ? malloc((size_t) MAX_EXTEND_MATCH_LEFT_LENGTH + 16) : NULL;
    uint8_t *extEnd = ext ? ext + MAX_EXTEND_MATCH_LEFT_LENGTH + 16 : NULL;
    const uint64_t seqEnd = d->nLit;
    uint32_t unmatchedChars = 0;
    const uint32_t minMatchLength = 0;
    int64_t paired[MAX_GAP_DEPTH];
    for (int i = 0; i < MAX_GAP_DEPTH; i++) paired[i] = INT64_MAX;
    int64_t gapStartIdx = NO_GAP, gapEndIdx = NO_GAP;
    int gapCurIdx = 0;
    uint64_t matchSrcPos = 0, prevMatchDestPos = 0;
    int64_t offsetDelta = -1;
    uint64_t extLeftLen = 0, extRightLen = 0;
    int isGap = 0;
    int64_t j = 0;
    uint64_t markPos = find_mark(d, d->litPos);
    while (!d->bad && markPos != NPOS && markPos < seqEnd) {
        const uint64_t literalsLeft = markPos - d->litPos;
        matchSrcPos = 0;
        const int skipOffset = paired[gapCurIdx] != INT64_MAX;
        if (skipOffset) {
            matchSrcPos = (uint64_t) (paired[gapCurIdx] + (int64_t) d->destLen + (int64_t) literalsLeft);
            paired[gapCurIdx] = INT64_MAX;
        } else {
            uint32_t lo = 0;
            if (d->offPos + 4 > d->nOff) { d->bad = 1; break; }
            memcpy(&lo, d->off + d->offPos, 4); d->offPos += 4;
            matchSrcPos = lo;
            if (p->enable40bitReference) {                           /* refTotalLength > UINT32_MAX, :356-359 */
                if (d->off5Pos >= d->nOff5) { d->bad = 1; break; }
                matchSrcPos += (uint64_t) d->off5[d->off5Pos++] << 32;
            }
        }
        extLeftLen = 0;
        if (p->enableExtensionsWithMismatches) {
            if (!isGap && literalsLeft) extLeftLen = extend_left(d, extEnd, &matchSrcPos, skipOffset, refLockPos, markPos);
            if (gapEndIdx == j) { gapStartIdx = NO_GAP; gapEndIdx = NO_GAP; }
        }
        if (d->bad) break;
        const uint64_t literalLen = markPos - d->litPos + extLeftLen + extRightLen;
        append(d, d->lit + d->litPos, markPos - d->litPos);
        if (extLeftLen) append(d, extEnd - extLeftLen, extLeftLen);
        unmatchedChars += (uint32_t) literalLen;
        d->litPos = markPos + 1;
        uint32_t matchLength = next_len(d);
        matchLength += minMatchLength;
        if (d->bad) break;
        prevMatchDestPos = d->destLen;
        append(d, ref + matchSrcPos, matchLength);
        markPos = find_mark(d, d->litPos);
        uint8_t gapDelta = 0;
        if (p->gapDepthOffsetEncoding && markPos != NPOS && markPos < seqEnd) {
            if (d->gapPos >= d->nGap) { d->bad = 1; break; }
            gapDelta = d->gap[d->gapPos++];
        }
        if (gapDelta) {
            int gapIdx = gapCurIdx;
            int g = gapDelta;
            if (!p->lazyDecompressionSupport && gapStartIdx == NO_GAP && markPos - d->litPos == 0) {
                gapIdx = (gapIdx + 1) % MAX_GAP_DEPTH;
                g++;
            }
            while (gapDelta) {
                gapIdx = (gapIdx + 1) % MAX_GAP_DEPTH;
                if (paired[gapIdx] == INT64_MAX) gapDelta--;
                else g++;
            }
            paired[gapIdx] = (int64_t) matchSrcPos - (int64_t) prevMatchDestPos;
            if (p->enableExtensionsWithMismatches && gapEndIdx <= j + g && g <= p->gapDepthMismatchesEncoding) {
                gapStartIdx = j;
                gapEndIdx = j + g;
            }
        }
        gapCurIdx = (gapCurIdx + 1) % MAX_GAP_DEPTH;
        const int gapStart = gapStartIdx == j;
        const int gapEnd = gapEndIdx == j + 1;
        const int gapMiddle = gapStartIdx < j && j + 1 < gapEndIdx;
        isGap = gapStart || gapMiddle || gapEnd;
        extRightLen = 0;
        if (p->enableExtensionsWithMismatches) {
            if (!isGap || gapStart) offsetDelta = (int64_t) matchSrcPos + (int64_t) matchLength - (int64_t) d->destLen;
            extRightLen = extend_right(d, offsetDelta, isGap, gapStart, gapMiddle, gapEnd,
                                       markPos != NPOS && markPos < seqEnd ? markPos : seqEnd);
        }
        j++;
    }
    if (!d->bad) {
        const uint64_t literalLen = seqEnd - d->litPos + extRightLen;
        append(d, d->lit + d->litPos, seqEnd - d->litPos);
        unmatchedChars += (uint32_t) literalLen;
        d->litPos = seqEnd;
    }
    free(ext);
    *destLen = d->destLen;
    if (d->bad) return -1;
    /* every byte of every stream belongs to exactly one step of the automaton */
    if (d->offPos != d->nOff || d->off5Pos != d->nOff5 || d->lenPos != d->nLen || d->gapPos != d->nGap || d->flPos != d->nFlags) return -1;
    return (int64_t) unmatchedChars;
}
