/* TEST INFRASTRUCTURE ONLY — CPU restatement of the `-m3` reverse-complement pass over the literal stream
 * (SURVEY.md §8(f) row 2):
 *
 *   MBGC_Encoder::prepareAndCompressStreams           mbgccoder/MBGC_Encoder.cpp:636-638
 *     SimpleSequenceMatcher::rcMatchSequence          matching/SimpleSequenceMatcher.cpp:165-176
 *       CopMEMMatcher (index over the sequence)       matching/copmem/CopMEMMatcher.cpp:68-144 (parameters),
 *                                                     :146-225 (genCumm + processRef, the single-thread form)
 *       markAndRemoveExactMatches(destIsRef, rc)      SimpleSequenceMatcher.cpp:68-148
 *         exactMatchSequence                          :26-55  (query = reverseComplement(sequence), utils/helper.cpp:429-437)
 *           CopMEMMatcher::matchTexts                 CopMEMMatcher.cpp:519-540 -> processExactMatchQueryTight :349-495
 *         correctDestPositionDueToRevComplMatching    :59-62
 *         resolveMappingCollisionsInTheSameText       :150-163
 *   hash: maRushPrime1HashSparsified<K>               utils/Hashes.h:42-68
 *
 * Single-thread semantics are the oracle: with more than one thread the reference fills its buckets in the order its
 * threads arrive (processRefMultithreaded, :279-335) and the matches it finds depend on that order.
 * Parity status: PINNED against the reference's own classes compiled into oracle/_ref/libswsem_ref.so
 * (ref_harness.cpp: refrc_*, PgHelpers::numberOfThreads = 1), tests/test_oracle_vs_ref.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle.h"

#define COLLISIONS_LIMIT 12              /* HASH_COLLISIONS_PER_POSITION_LIMIT, CopMEMMatcher.h:11 */
#define RC_MATCH_MARK ((uint8_t) ('$' + 128))   /* MBGC_Params.h:48 */

typedef struct {
    const uint8_t *start1;
    uint64_t N;
    int L, K, k1, k2, LK2, K_PLUS_LK24;
    uint32_t hash_size, mask;
    uint64_t *cumm;                      /* hash_size + 2 */
    uint64_t *sampled;
} copmem;

/* maRushPrime1HashSparsified<K>, utils/Hashes.h:47-68 */
static uint32_t hash_sparsified(const uint8_t *s, int K) {
    uint64_t h = (uint64_t) K;
    uint32_t j = 0, k;
    while (j < 3) {                                     /* SPARSIFY_MASK_A_COUNT */
        memcpy(&k, s, 4);
        k &= 0x00FFFFFFu;
        k += j++;
        h ^= k; h *= 171717; s += 4;
    }
    while (j < (uint32_t) K / 4) {
        memcpy(&k, s, 4);
        k &= 0x0000FFFFu;
        k += j++;
        h ^= k; h *= 171717; s += 4;
    }
    return (uint32_t) h;
}
uint32_t orc_hash_sparsified(const uint8_t *s, int K) { return hash_sparsified(s, K); }

/* initParams + calcCoprimes, CopMEMMatcher.cpp:68-144. Returns 0 when the reference would exit. */
static int copmem_params(copmem *c, uint32_t minMatchLength) {
    const int L = c->L;
    if (L > 110) c->K = 56;
    else if (L > 62) c->K = 44;
    else if (L > 53) c->K = 40;
    else if (L > 46) c->K = 36;
    else if (L > 42) c->K = 32;
    else if (L > 32) c->K = 28;
    else c->K = (L / 4 - 1) * 4;
    if (minMatchLength < 24) return 0;                  /* "Minimal matching length too short!" */
    const int KmmL = (int) (minMatchLength / 4 - 1) * 4;
    if (KmmL < c->K) c->K = KmmL;
    const int t = L - c->K + 1;
    if (t <= 0) return 0;                               /* "L and K mismatch." */
    if (t >= 20) {
        c->k1 = (int) pow((double) t, 0.5) + 1;
        c->k2 = c->k1 - 1;
        if (c->k1 * c->k2 > t) { --c->k2; --c->k1; }
    } else if (t >= 15) { c->k1 = 5; c->k2 = 3; }
    else if (t >= 12) { c->k1 = 4; c->k2 = 3; }
    else if (t >= 10) { c->k1 = 5; c->k2 = 2; }
    else if (t >= 6) { c->k1 = 3; c->k2 = 2; }
    else { c->k1 = t; c->k2 = 1; }
    c->LK2 = (L - c->K) / 2;
    c->K_PLUS_LK24 = c->K + c->LK2 - 4;
    uint8_t i = 24;                                     /* HASH_SIZE_MIN_ORDER .. MAX_ORDER */
    do {
        c->hash_size = ((uint32_t) 1) << (i++);
    } while (i <= 31 && c->hash_size < c->N / (uint64_t) c->k1);
    c->mask = c->hash_size - 1;
    return 1;
}

/* genCumm + processRef (single thread), :146-225: every bucket keeps the first 13 sampled positions in text order */
static void copmem_index(copmem *c) {
    const uint64_t N = c->N;
    const int K = c->K, k1 = c->k1;
    uint64_t *cumm = calloc((size_t) c->hash_size + 2, sizeof(uint64_t));
    uint8_t *skipped = calloc((size_t) (N / (uint64_t) k1 + 2), 1);      /* skippedList as a bitmap over sample numbers */
    for (uint64_t i = 0; i + (uint64_t) K <= N; i += (uint64_t) k1) {    /* both loops of genCumm visit i = 0, k1, 2 k1 ... < N - K + 1 */
        const uint32_t h = (hash_sparsified(c->start1 + i, K) & c->mask) + 2;
        if (cumm[h] <= COLLISIONS_LIMIT) ++cumm[h];
        else skipped[i / (uint64_t) k1] = 1;
    }
    for (uint64_t h = 1; h < (uint64_t) c->hash_size + 2; h++) cumm[h] += cumm[h - 1];   /* partial_sum */
    const uint64_t hashCount = cumm[c->hash_size + 1];
    uint64_t *sampled = malloc((size_t) (hashCount + 2) * sizeof(uint64_t));
    for (uint64_t i = 0; i + (uint64_t) K <= N; i += (uint64_t) k1) {
        if (skipped[i / (uint64_t) k1]) continue;
        const uint32_t h = (hash_sparsified(c->start1 + i, K) & c->mask) + 1;
        sampled[cumm[h]] = i;
        ++cumm[h];
    }
    free(skipped);
    c->cumm = cumm; c->sampled = sampled;               /* now: bucket h = sampled[cumm[h] .. cumm[h + 1]) */
}

typedef struct { orc_match *m; uint64_t n, cap; } mvec;
static void mpush(mvec *v, uint64_t src, uint64_t len, uint64_t dest) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->m = realloc(v->m, (size_t) v->cap * sizeof(orc_match)); }
    v->m[v->n].posSrcText = src; v->m[v->n].length = len; v->m[v->n].posDestText = dest; v->m[v->n].nextSrcRegionLoadingPos = 0;
    v->n++;
}

/* processExactMatchQueryTight, :349-495: blocks of 256 query samples (a skip ends with its block), then the rest one by one.
 * The 4-byte pre-filter (:404-407,:460-463) is restated with the stale values the reference leaves in l/r when a read
 * would leave a text; the tail loop's unconditional reads (:460-461) are guarded the same way (the reference reads out of
 * bounds there — undefined; the filter is result-neutral: a match longer than L covers one of the two windows). */
static void copmem_query(const copmem *c, mvec *res, const uint8_t *start2, uint64_t N2, int destIsSrc, int revCompl,
                         uint32_t minMatchLength, uint64_t *charExtensions) {
    const int K = c->K, k1 = c->k1, k2 = c->k2, LK2 = c->LK2, KL = c->K_PLUS_LK24;
    const uint64_t MULTI = 256, k2MULTI = (uint64_t) k2 * MULTI;
    const uint8_t *start1 = c->start1, *end1 = start1 + c->N, *end2 = start2 + N2;
    uint32_t l1 = 0, l2 = 0, r1 = 0, r2 = 0;
    const int skip = K / k1 - 1;
    const uint64_t skipK2 = (uint64_t) skip * (uint64_t) k2;
    uint64_t i1 = 0, ext = 0;
    for (int tail = 0; tail < 2; tail++) {
        for (;;) {
            uint64_t nsamples;
            if (!tail) { if (!(i1 + (uint64_t) K + k2MULTI < N2 + 1)) break; nsamples = MULTI; }
            else { if (!(i1 + (uint64_t) K < N2 + 1)) break; nsamples = 1; }
            const uint8_t *curr2 = start2 + i1;
            for (uint64_t i2 = 0; i2 < nsamples; ++i2) {
                const uint32_t h = hash_sparsified(curr2, K) & c->mask;
                const uint64_t b0 = c->cumm[h], b1 = c->cumm[h + 1];
                if (b0 == b1) { curr2 += k2; continue; }
                if (curr2 - LK2 >= start2) memcpy(&l2, curr2 - LK2, 4);
                if (curr2 + KL + 4 <= end2) memcpy(&r2, curr2 + KL, 4);
                for (uint64_t j = b0; j < b1; ++j) {
                    ++ext;
                    const uint8_t *curr1 = start1 + c->sampled[j];
                    const uint64_t tmpSrc = c->sampled[j], tmpDest = (uint64_t) (curr2 - start2);
                    if (destIsSrc && (revCompl ? N2 - tmpSrc < tmpDest : tmpDest >= tmpSrc)) continue;
                    if (res->n > 0) {
                        const orc_match *bk = &res->m[res->n - 1];
                        if (tmpDest - tmpSrc == bk->posDestText - bk->posSrcText && tmpDest + (uint64_t) K < bk->posDestText + bk->length) {
                            curr2 += skipK2;
                            if (!tail) i2 += (uint64_t) skip; else i1 += skipK2;
                            break;
                        }
                    }
                    if (curr1 - LK2 >= start1) memcpy(&l1, curr1 - LK2, 4);
                    if (curr1 + KL + 4 <= end1) memcpy(&r1, curr1 + KL, 4);
                    if (r1 == r2 || l1 == l2) {
                        const uint8_t *p1 = curr1 + K - 1, *p2 = curr2 + K - 1;
                        while (++p1 != end1 && ++p2 != end2 && *p1 == *p2);
                        const uint8_t *right = p1;
                        p1 = curr1; p2 = curr2;
                        while (p1 != start1 && p2 != start2 && *p1 == *p2) { p1--; p2--; }
                        if ((uint64_t) (right - p1) > minMatchLength && memcmp(curr1, curr2, (size_t) K) == 0) {
                            mpush(res, (uint64_t) (p1 + 1 - start1), (uint64_t) (right - p1 - 1), (uint64_t) (p2 + 1 - start2));
                            curr2 += skipK2;
                            if (!tail) i2 += (uint64_t) skip; else i1 += skipK2;
                            break;
                        }
                    }
                }
                curr2 += k2;
            }
            i1 += tail ? (uint64_t) k2 : k2MULTI;
        }
    }
    if (charExtensions) *charExtensions = ext;
}

/* PgHelpers::complementsLUT, utils/helper.cpp:312-361: the upper LUT, then the lower-case letters map to lower case and U/u
 * to themselves. (Entry 127 stays 0: the constructor's loops stop at i < CHAR_MAX.) */
static void complements_lut(uint8_t *lut) {
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t) i;
    lut[127] = 0;
    const char *from = "AaCcGgTtNnUuYyRrKkMmBbDdHhVvWwSs", *to = "TTGGCCAANNAARRYYMMKKVVHHDDBBSSWW";
    for (int i = 0; from[i]; i++) lut[(uint8_t) from[i]] = (uint8_t) to[i];
    lut['U'] = 'U'; lut['u'] = 'u';
    const char *lf = "acgtnyrkmbdhvws", *lt = "tgcanrymkvhdbsw";
    for (int i = 0; lf[i]; i++) lut[(uint8_t) lf[i]] = (uint8_t) lt[i];
}
void orc_reverse_complement(const uint8_t *src, uint64_t n, uint8_t *dst) {
    uint8_t lut[256];
    complements_lut(lut);
    for (uint64_t i = 0; i < n; i++) dst[n - 1 - i] = lut[src[i]];
}

static int match_less(const void *a, const void *b) {    /* TextMatch::operator<, TextMatchers.h:30-40 */
    const orc_match *x = a, *y = b;
    if (x->posDestText != y->posDestText) return x->posDestText < y->posDestText ? -1 : 1;
    if (x->posSrcText != y->posSrcText) return x->posSrcText < y->posSrcText ? -1 : 1;
    if (x->length != y->length) return x->length < y->length ? -1 : 1;
    return 0;
}

static void put_byte_frugal(orc_buf *b, uint64_t v) {    /* writeUIntByteFrugal, utils/helper.cpp:217-225 */
    while (v >= 128) { const uint8_t y = (uint8_t) (128 + v % 128); orc_buf_put(b, &y, 1); v /= 128; }
    const uint8_t y = (uint8_t) v;
    orc_buf_put(b, &y, 1);
}

/* The matches CopMEMMatcher::matchTexts(textMatches, reverseComplement(seq), destIsSrc = true, revCompl = true, minLen)
 * pushes, in push order, before any post-processing (rows {posSrcText, length, posDestText} in the coordinates of the
 * query = the reverse-complemented text). *out is malloc'd. Returns the count, or UINT64_MAX where the reference exits. */
uint64_t orc_rc_find_matches(const uint8_t *seq, uint64_t n, uint32_t targetMatchLength, uint32_t minMatchLength,
                             orc_match **out, int *params /* K, k1, k2, log2(hash_size) */, uint64_t *charExtensions) {
    *out = NULL;
    if (n < targetMatchLength) return 0;                                    /* SimpleSequenceMatcher.cpp:16: no matcher */
    copmem c;
    memset(&c, 0, sizeof c);
    c.start1 = seq; c.N = n; c.L = (int) targetMatchLength;
    uint32_t mm = minMatchLength > targetMatchLength ? targetMatchLength : minMatchLength;   /* CopMEMMatcher.cpp:500-501 */
    if (!copmem_params(&c, mm)) return UINT64_MAX;
    if (params) { params[0] = c.K; params[1] = c.k1; params[2] = c.k2; params[3] = 31 - __builtin_clz(c.hash_size); }
    copmem_index(&c);
    uint8_t *q = malloc((size_t) n + 1);
    orc_reverse_complement(seq, n, q);
    mvec res = {0, 0, 0};
    uint32_t qmin = minMatchLength == UINT32_MAX ? targetMatchLength : minMatchLength;     /* SimpleSequenceMatcher.cpp:80-81 */
    if (qmin < (uint32_t) c.K) { free(q); free(c.cumm); free(c.sampled); return UINT64_MAX; }   /* CopMEMMatcher.cpp:522-525 */
    copmem_query(&c, &res, q, n, 1, 1, qmin, charExtensions);
    free(q); free(c.cumm); free(c.sampled);
    *out = res.m;
    return res.n;
}

/* markAndRemoveExactMatches(destSeqIsRef = true, ..., revComplMatching = true) from the matches on: SimpleSequenceMatcher.cpp
 * :59-62 (dest positions back to the forward text), :150-163 (collisions in the same text), :93-147 (sort, unique, trim
 * overlaps, cut the matched parts out of the sequence, write the two maps). seq is rewritten in place; returns its new length. */
uint64_t orc_rc_apply_matches(uint8_t *seq, uint64_t n, orc_match *m, uint64_t nm, uint32_t targetMatchLength, uint32_t minMatchLength,
                              orc_buf *mapOff, orc_buf *mapLen, uint64_t *stats /* [0] unique matches, [1] matched, [2] overlapped */) {
    if (minMatchLength == UINT32_MAX) minMatchLength = targetMatchLength;
    for (uint64_t i = 0; i < nm; i++) m[i].posDestText = n - (m[i].posDestText + m[i].length);          /* :59-62 */
    for (uint64_t i = 0; i < nm; i++) {                                                                  /* :150-163 */
        if (m[i].posSrcText > m[i].posDestText) { const uint64_t t = m[i].posSrcText; m[i].posSrcText = m[i].posDestText; m[i].posDestText = t; }
        if (m[i].posSrcText + m[i].length > m[i].posDestText) {
            const uint64_t margin = (m[i].posSrcText + m[i].length - m[i].posDestText + 1) / 2;
            m[i].length -= margin;
            m[i].posDestText += margin;
        }
    }
    put_byte_frugal(mapLen, minMatchLength);                                                             /* :91 */
    qsort(m, (size_t) nm, sizeof(orc_match), match_less);                                                /* :93 */
    uint64_t u = 0;
    for (uint64_t i = 0; i < nm; i++)                                                                    /* :94 unique */
        if (u == 0 || match_less(&m[u - 1], &m[i]) != 0) m[u++] = m[i];
    uint64_t pos = 0, nPos = 0, overlap = 0, matched = 0;
    const int std32 = n <= UINT32_MAX;                                                                   /* :102 */
    for (uint64_t i = 0; i < u; i++) {                                                                   /* :103-131 */
        orc_match *t = &m[i];
        if (t->posDestText < pos) {
            const uint64_t over = pos - t->posDestText;
            if (over >= t->length) { overlap += t->length; t->length = 0; continue; }
            overlap += over;
            t->length -= over;
            t->posDestText += over;
        }
        if (t->length < minMatchLength) { overlap += t->length; continue; }
        matched += t->length;
        const uint64_t len = t->posDestText - pos;
        memmove(seq + nPos, seq + pos, (size_t) len);
        nPos += len;
        seq[nPos++] = RC_MATCH_MARK;
        if (std32) { const uint32_t v = (uint32_t) t->posSrcText; orc_buf_put(mapOff, &v, 4); }
        else orc_buf_put(mapOff, &t->posSrcText, 8);
        put_byte_frugal(mapLen, t->length - minMatchLength);
        pos = t->posDestText + t->length;
    }
    memmove(seq + nPos, seq + pos, (size_t) (n - pos));
    nPos += n - pos;
    if (stats) { stats[0] = u; stats[1] = matched; stats[2] = overlap; }
    return nPos;
}

/* SimpleSequenceMatcher::rcMatchSequence(sequence, rcMapOff, rcMapLen, targetMatchLength, minMatchLength), :165-176.
 * Returns the new length of seq (rewritten in place), or UINT64_MAX where the reference exits. */
uint64_t orc_rc_match_sequence(uint8_t *seq, uint64_t n, uint32_t targetMatchLength, uint32_t minMatchLength,
                               orc_buf *mapOff, orc_buf *mapLen, uint64_t *stats) {
    if (n < targetMatchLength) return n;                                    /* :68-73: no matcher, the maps stay empty */
    orc_match *m = NULL;
    const uint64_t nm = orc_rc_find_matches(seq, n, targetMatchLength, minMatchLength, &m, NULL, NULL);
    if (nm == UINT64_MAX) return UINT64_MAX;
    const uint64_t r = orc_rc_apply_matches(seq, n, m, nm, targetMatchLength, minMatchLength, mapOff, mapLen, stats);
    free(m);
    return r;
}
