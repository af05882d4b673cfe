/* TEST INFRASTRUCTURE ONLY — see oracle.h. Plain-C restatement of the per-contig stream emission:
 * MBGC_Encoder::processMatches / extendMatchRight / extendMatchLeft (mbgccoder/MBGC_Encoder.cpp:
 * 137-427), ContextAwareMismatchesCoder::mismatch2code (coders/ContextAwareMismatchesCoder.cpp:65-70)
 * and writeUInt64Frugal (utils/helper.cpp:237-246). Release-build semantics (DEVELOPER_BUILD off);
 * processLiteral is a no-op there because refLiteralMinimalLengthExt = SIZE_MAX
 * (matching/MGMP_Params.h:45,175-177). Parity: PINNED against oracle/_ref and tests/golden.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define MAX_GAP_DEPTH 128                 /* MBGC_Params.h:50 */
#define MATCH_MARK 0xA5                   /* '%'+128, MBGC_Params.h:45 */
#define REF_REGION_SEPARATOR 0            /* MGMP_Params.h:14 */
#define MAX_EXTEND_MATCH_LEFT_LENGTH (1 << 24) /* MBGC_Params.h:55 */

void orc_emit_params_default(orc_emit_params *p, int mode) {
    memset(p, 0, sizeof(*p));
    p->enableExtensionsWithMismatches = 1;
    p->mismatchesWithExclusion = 1;
    p->lazyDecompressionSupport = 1;
    p->enable40bitReference = 0;
    p->frugal64bitLenEncoding = 1;
    p->gapDepthOffsetEncoding = 64;
    p->gapDepthMismatchesEncoding = 2;
    p->gapBreakingMatchMinLength = 256;
    /* initMismatchesMatchingScoreParams, MBGC_Params.h:92-97 with the defaults of :57-59 */
    p->mmsMatchBonus = 50;
    p->mmsMismatchPenalty = 100 - p->mmsMatchBonus;
    p->mmsMismatchesScoreThreshold = 10 * p->mmsMismatchPenalty;
    p->mmsMismatchesInitialScore = p->mmsMismatchesScoreThreshold * 25 / 100;
    p->allowedTargetsOutrunForDissimilarContigs = 1;
    p->minimalLengthForDissimilarContigs = 1024;
    p->unmatchedFractionFactorTweakForDissimilarContigs = 16;
    if (mode == 0) {                      /* setCompressionMode, MBGC_Params.h:893-902 */
        p->allowedTargetsOutrunForDissimilarContigs = 4;
        p->unmatchedFractionFactorTweakForDissimilarContigs = 32;
        p->frugal64bitLenEncoding = 0;
    }
    if (mode == 2) {                      /* :903-906 */
        p->allowedTargetsOutrunForDissimilarContigs = 0;
        p->unmatchedFractionFactorTweakForDissimilarContigs = 2;
    }
}

void orc_buf_put(orc_buf *b, const void *p, uint64_t n) {
    if (b->size + n > b->cap) {
        uint64_t nc = b->cap ? b->cap : 4096;
        while (nc < b->size + n) nc *= 2;
        b->data = (uint8_t *) realloc(b->data, nc);
        b->cap = nc;
    }
    if (n) memcpy(b->data + b->size, p, n);
    b->size += n;
}
static inline void put1(orc_buf *b, uint8_t v) { orc_buf_put(b, &v, 1); }

/* utils/helper.cpp:237-246 */
void orc_write_frugal64(orc_buf *b, uint64_t value) {
    uint16_t y16 = value < UINT16_MAX ? (uint16_t) value : UINT16_MAX;
    orc_buf_put(b, &y16, 2);
    if (value >= UINT16_MAX) {
        uint32_t y32 = value < UINT32_MAX ? (uint32_t) value : UINT32_MAX;
        orc_buf_put(b, &y32, 4);
        if (value >= UINT32_MAX) orc_buf_put(b, &value, 8);
    }
}

void orc_streams_init(orc_streams *s) { memset(s, 0, sizeof(*s)); }
void orc_streams_free(orc_streams *s) {
    for (int i = 0; i < ORC_NSTREAMS; i++) free(s->s[i].data);
    memset(s, 0, sizeof(*s));
}

/* coders/ContextAwareMismatchesCoder.h:10-17, .cpp:65-70. sym2val is indexed by a (signed) char in
 * the reference, so bytes >= 0x80 read outside the table (undefined); they are treated as
 * "not in ACGTN" here, i.e. the raw byte is emitted. */
uint8_t orc_mismatch2code(uint8_t actual, uint8_t mismatch) {
    static const int8_t mis2code[5][5] = {{-1, 2, 0, 1, 3}, {1, -1, 2, 0, 3}, {0, 2, -1, 1, 3},
                                          {1, 0, 2, -1, 3}, {1, 2, 3, 0, -1}};
    int a = -1, b = -1;
    switch (actual) { case 'A': a = 0; break; case 'C': a = 1; break; case 'G': a = 2; break; case 'T': a = 3; break; case 'N': a = 4; break; default: break; }
    switch (mismatch) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; case 'N': b = 4; break; default: break; }
    if (actual == mismatch || a < 0 || b < 0) return mismatch;
    return (uint8_t) mis2code[a][b];
}

typedef struct {
    const orc_matcher *m;
    const orc_emit_params *p;
    const uint8_t *ref;
    orc_streams *out;
    const uint64_t *loaded; uint64_t nLoaded;
    uint32_t extMatched, extMismatches;
} ctx_t;

static inline int paired(const orc_match *a, const orc_match *b) {                 /* TextMatchers.h:42-44 */
    return a->posSrcText + b->posDestText == b->posSrcText + a->posDestText;
}
static inline int paired_lock(const orc_match *a, const orc_match *b, uint64_t brk) { /* TextMatchers.h:46-50 */
    return a->posSrcText + b->posDestText == b->posSrcText + a->posDestText &&
           ((a->posSrcText > brk && b->posSrcText > brk) || (a->posSrcText < brk && b->posSrcText < brk));
}

/* getMatchLoadedPos, MBGC_Encoder.cpp:137-141 */
static uint64_t match_loaded_pos(const ctx_t *c, uint64_t pos) {
    const uint64_t span = orc_ref_length(c->m) - 1;
    while (pos + span < c->loaded[c->nLoaded - 1]) pos += span;
    return pos;
}

/* extendMatchRight, MBGC_Encoder.cpp:310-371. gap = contig bytes right after the match. */
static uint64_t extend_right(ctx_t *c, const uint8_t *gapStart, const orc_match *core, const orc_match *match,
                             uint64_t length, int isGap, int gapStartF, int gapMiddle, int gapEnd) {
    const orc_emit_params *p = c->p;
    orc_buf *lit = &c->out->s[ORC_LIT], *fl = &c->out->s[ORC_FLAGS];
    if (length == 0) {
        if (gapMiddle) put1(fl, 1);
        return 0;
    }
    const uint8_t *ref = c->ref;
    const uint8_t *gapPtr = gapStart;
    int64_t src = (int64_t) (isGap ? core->posSrcText + (match->posDestText + match->length) - core->posDestText
                                   : match->posSrcText + match->length);
    const int64_t srcLoadingPos = (int64_t) orc_loading_position(c->m);
    const int64_t srcGuard = src + (int64_t) length;
    int64_t validSrcGuard = src + (int64_t) length;
    if (src == srcLoadingPos) validSrcGuard = src;
    if (!isGap) {
        const int64_t srcEnd = (int64_t) orc_max_ref_length(c->m);
        if (validSrcGuard > srcEnd) validSrcGuard = srcEnd;
        if (src <= srcLoadingPos && srcLoadingPos < validSrcGuard) validSrcGuard = srcLoadingPos;
    }
    if (gapStartF || !isGap) {
        if (p->lazyDecompressionSupport && ref[src] == REF_REGION_SEPARATOR) validSrcGuard = src;
        c->extMismatches++;
        put1(lit, p->mismatchesWithExclusion && src < validSrcGuard ? orc_mismatch2code(ref[src], *gapPtr) : *gapPtr);
        gapPtr++;
    } else
        src--;
    int score = p->mmsMismatchesInitialScore;
    while (++src < validSrcGuard && (!p->lazyDecompressionSupport || ref[src] != REF_REGION_SEPARATOR) &&
           (isGap || score < p->mmsMismatchesScoreThreshold)) {
        const int mismatch = *gapPtr != ref[src];
        put1(fl, mismatch ? 1 : 0);
        if (mismatch) {
            c->extMismatches++;
            score += p->mmsMismatchPenalty;
            put1(lit, p->mismatchesWithExclusion ? orc_mismatch2code(ref[src], *gapPtr) : *gapPtr);
        } else {
            c->extMatched++;
            score -= p->mmsMatchBonus;
            if (score < 0) score = 0;
        }
        gapPtr++;
    }
    while (src++ < srcGuard && (isGap || score < p->mmsMismatchesScoreThreshold)) {
        put1(fl, 1);
        put1(lit, *gapPtr++);
        c->extMismatches++;
        score += p->mmsMismatchPenalty;
    }
    if ((isGap && !gapEnd) || (!isGap && score < p->mmsMismatchesScoreThreshold)) put1(fl, 1);
    return (uint64_t) (gapPtr - gapStart);
}

/* extendMatchLeft, MBGC_Encoder.cpp:373-427 */
static uint64_t extend_left(ctx_t *c, const uint8_t *dest, uint64_t length, const orc_match *match, uint64_t lockPos) {
    const orc_emit_params *p = c->p;
    orc_buf *lit = &c->out->s[ORC_LIT], *fl = &c->out->s[ORC_FLAGS];
    const uint8_t *ref = c->ref;
    const int64_t srcMatch = (int64_t) match->posSrcText;
    int64_t srcGuard = 1;
    if (srcGuard < srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH) srcGuard = srcMatch - MAX_EXTEND_MATCH_LEFT_LENGTH;
    /* `ref + matchingLockPos` with the default SIZE_MAX wraps to ref - 1 in the reference, which is
     * below srcGuard: the test is then false, exactly like a signed -1 here. */
    const int64_t srcLock = (int64_t) lockPos;
    if (srcGuard < srcLock && srcLock <= srcMatch) srcGuard = srcLock;
    int guardKnown = 1;
    if (srcGuard < srcMatch - (int64_t) length) {
        srcGuard = srcMatch - (int64_t) length;
        guardKnown = 0;
    }
    if (srcGuard == srcMatch) return 0;
    int64_t src = srcMatch - 1;
    const uint8_t *gapPtr = dest + match->posDestText - 1;
    int validSrcRegion = !p->lazyDecompressionSupport || ref[src] != REF_REGION_SEPARATOR;
    c->extMismatches++;
    put1(lit, p->mismatchesWithExclusion && validSrcRegion ? orc_mismatch2code(ref[src], *gapPtr) : *gapPtr);
    int score = p->mmsMismatchesInitialScore;
    while (validSrcRegion && src > srcGuard && score < p->mmsMismatchesScoreThreshold) {
        --gapPtr; --src;
        const int mismatch = *gapPtr != ref[src];
        if (p->lazyDecompressionSupport && ref[src] == REF_REGION_SEPARATOR) {
            validSrcRegion = 0;
            src++;
            gapPtr++;
            break;
        }
        put1(fl, mismatch ? 1 : 0);
        if (mismatch) {
            score += p->mmsMismatchPenalty;
            c->extMismatches++;
            put1(lit, p->mismatchesWithExclusion ? orc_mismatch2code(ref[src], *gapPtr) : *gapPtr);
        } else {
            score -= p->mmsMatchBonus;
            if (score < 0) score = 0;
            c->extMatched++;
        }
    }
    while (!validSrcRegion && src > srcGuard && score < p->mmsMismatchesScoreThreshold) {
        src--;
        put1(fl, 1);
        put1(lit, *--gapPtr);
        c->extMismatches++;
        score += p->mmsMismatchPenalty;
    }
    if ((src != srcGuard || !guardKnown) && score < p->mmsMismatchesScoreThreshold) put1(fl, 1);
    return (uint64_t) (srcMatch - src);
}

/* processMatches, MBGC_Encoder.cpp:143-308 */
uint64_t orc_process_matches(const orc_matcher *m, const orc_emit_params *p, orc_match *tm, uint64_t *nmatches,
                             const uint8_t *dest, uint64_t destLen, uint64_t lockPos, int unmatchedFractionFactor,
                             int64_t processedTargetsCount, int64_t targetIdx,
                             const uint64_t *refExtLoadedPos, uint64_t nLoaded, orc_streams *out) {
    ctx_t c = {m, p, orc_ref(m), out, refExtLoadedPos, nLoaded, 0, 0};
    const uint8_t *ref = c.ref;
    int64_t n = (int64_t) *nmatches;
    uint32_t pos = 0;
    int64_t unmatchedChars = 0;
    uint32_t totalMatched = 0, removed = 0;
    int64_t jj = 0, j;
    for (j = 0; j < n; j++, jj++) {                                             /* pass 1, :153-198 */
        orc_match *match = &tm[j];
        if (p->enableExtensionsWithMismatches && j + 1 < n) {
            if (jj > 0 && paired(&tm[j + 1], &tm[jj - 1]) && !paired(match, &tm[jj - 1]) &&
                match->length < p->gapBreakingMatchMinLength) {
                int64_t leftExtension = 0;
                if (match->posDestText + match->length == tm[j + 1].posDestText) {
                    /* :181-185 is unbounded in the reference (it relies on meeting a mismatch);
                     * bounded here by both buffer starts, identical whenever the reference is defined */
                    int64_t s = (int64_t) tm[j + 1].posSrcText, d = (int64_t) tm[j + 1].posDestText;
                    while (d - 1 >= 0 && s - 1 >= 0 && dest[d - 1] == ref[s - 1]) { d--; s--; leftExtension++; }
                    tm[j + 1].posSrcText -= (uint64_t) leftExtension;           /* shiftStartPos(-x) */
                    tm[j + 1].posDestText -= (uint64_t) leftExtension;
                    tm[j + 1].length += (uint64_t) leftExtension;
                }
                removed++;
                jj--;
                continue;
            }
        }
        totalMatched += (uint32_t) match->length;
        unmatchedChars += (int64_t) (match->posDestText - pos);
        pos = (uint32_t) (match->posDestText + match->length);
        tm[jj] = tm[j];
    }
    n = jj;
    *nmatches = (uint64_t) n;
    uint64_t literalsLeft = destLen - pos;
    unmatchedChars += (int64_t) literalsLeft;
    /* :203-205, isContigDissimilar MGMP_Params.h:193-196 */
    if (processedTargetsCount < targetIdx - p->allowedTargetsOutrunForDissimilarContigs &&
        destLen > p->minimalLengthForDissimilarContigs &&
        (uint64_t) (unmatchedChars * (int64_t) (unmatchedFractionFactor / p->unmatchedFractionFactorTweakForDissimilarContigs)) > destLen)
        return ORC_SKIPPED;
    orc_buf *lit = &out->s[ORC_LIT];
    int pairedGap[MAX_GAP_DEPTH] = {0};
    const int64_t NO_GAP = -1;
    int64_t gapStartIdx = NO_GAP, gapEndIdx = NO_GAP;
    int gapCurIdx = 0;
    pos = 0;
    int isGap = 0;
    for (j = 0; j < n; j++) {                                                   /* pass 2, :214-287 */
        orc_match *match = &tm[j];
        literalsLeft = match->posDestText - pos;
        if (p->enableExtensionsWithMismatches) {
            if (!isGap && literalsLeft) literalsLeft -= extend_left(&c, dest, literalsLeft, match, lockPos);
            if (j == gapEndIdx) { gapStartIdx = NO_GAP; gapEndIdx = NO_GAP; }
        }
        orc_buf_put(lit, dest + pos, literalsLeft);
        if (!pairedGap[gapCurIdx]) {
            uint32_t off = (uint32_t) match->posSrcText;
            orc_buf_put(&out->s[ORC_OFF], &off, 4);
            if (p->enable40bitReference) put1(&out->s[ORC_OFF5], (uint8_t) (match->posSrcText >> 32));
        }
        if (p->frugal64bitLenEncoding)
            orc_write_frugal64(&out->s[ORC_LEN], match->length);
        else {
            uint32_t l32 = (uint32_t) match->length;
            orc_buf_put(&out->s[ORC_LEN], &l32, 4);
        }
        put1(lit, MATCH_MARK);
        pos = (uint32_t) (match->posDestText + match->length);
        literalsLeft = (j + 1 < n ? tm[j + 1].posDestText : destLen) - pos;
        int gapIdx = (gapCurIdx + 1) % MAX_GAP_DEPTH;
        int gCnt = (int) (n - j - 1) < p->gapDepthOffsetEncoding ? (int) (n - j - 1) : p->gapDepthOffsetEncoding;
        int reduce = 0, g;
        for (g = 1; g <= gCnt; g++, gapIdx = (gapIdx + 1) % MAX_GAP_DEPTH) {
            if (pairedGap[gapIdx] || (!p->lazyDecompressionSupport && g == 1 && gapStartIdx == NO_GAP && literalsLeft == 0)) {
                reduce++;
                continue;
            }
            if (paired_lock(match, &tm[j + g], lockPos)) {
                if (p->lazyDecompressionSupport) {
                    if (!match->nextSrcRegionLoadingPos) {
                        /* std::upper_bound over refExtLoadedPosArr, :254-256 */
                        const uint64_t key = match_loaded_pos(&c, match->posSrcText);
                        uint64_t lo = 0, hi = nLoaded;
                        while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (refExtLoadedPos[mid] <= key) lo = mid + 1; else hi = mid; }
                        match->nextSrcRegionLoadingPos = lo == nLoaded ? UINT64_MAX : refExtLoadedPos[lo];
                    }
                    if (match_loaded_pos(&c, tm[j + g].posSrcText) >= match->nextSrcRegionLoadingPos) continue;
                    tm[j + g].nextSrcRegionLoadingPos = match->nextSrcRegionLoadingPos;
                }
                pairedGap[gapIdx] = 1;
                if (p->enableExtensionsWithMismatches && gapEndIdx <= j + g && g <= p->gapDepthMismatchesEncoding) {
                    gapStartIdx = j;
                    gapEndIdx = j + g;
                }
                break;
            }
        }
        if (gCnt) put1(&out->s[ORC_GAP], (uint8_t) (g <= gCnt ? g - reduce : 0));
        pairedGap[gapCurIdx] = 0;
        gapCurIdx = (gapCurIdx + 1) % MAX_GAP_DEPTH;
        const int gapStartF = gapStartIdx == j;
        const int gapEnd = gapEndIdx == j + 1;
        const int gapMiddle = gapStartIdx < j && j + 1 < gapEndIdx;
        isGap = gapStartF || gapMiddle || gapEnd;
        if (p->enableExtensionsWithMismatches) {
            const orc_match *corr = isGap ? &tm[gapStartIdx] : match;
            uint64_t ext = extend_right(&c, dest + pos, corr, match, literalsLeft, isGap, gapStartF, gapMiddle, gapEnd);
            literalsLeft -= ext;
            pos += (uint32_t) ext;
        }
    }
    literalsLeft = destLen - pos;
    orc_buf_put(lit, dest + pos, literalsLeft);
    out->unmatchedChars += (uint64_t) unmatchedChars;
    out->extensionsMatchedChars += c.extMatched;
    out->extensionsMismatches += c.extMismatches;
    out->totalMatched += totalMatched;
    out->totalDestLen += destLen;
    out->removedGapBreakingMatches += removed;
    return (uint64_t) unmatchedChars;
}
