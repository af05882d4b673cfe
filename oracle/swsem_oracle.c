/* TEST INFRASTRUCTURE ONLY — see oracle.h. Plain-C, single-thread restatement of
 * matching/SlidingWindowSparseEMMatcher.{h,cpp} (Exp variant, H=1 hash, 32-bit table entries).
 * Citations are file:line under /root/reference. Parity: PINNED against oracle/_ref and tests/golden.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_SHIFT 1                      /* SlidingWindowSparseEMMatcher.h:14 */
#define OVERLAP_MATCH_MAX_LENGTH (1 << 13) /* .h:18 */
#define HASH_SIZE_MIN_ORDER 24           /* .h:19 */
#define HASH_SIZE_MAX_ORDER 31           /* .h:20 */
#define SW_WIDTH_FACTOR 16               /* .h:47 */
#define MAX_LOCKS 4096
#define REF_SLACK 64                     /* the reference over-reads a few bytes past the buffer (:224,:337) */

struct orc_matcher {
    uint8_t *ref;                        /* start1 */
    int64_t pos1;
    uint64_t maxRefLength;
    int laps;                            /* reachedRefLengthCount */
    int L, K, k1, k2, skipMargin, k1ord;
    int LK2, K_PLUS_LK24;
    uint32_t hash_size, mask;
    uint32_t *ht;
    uint64_t samplingPos;
    uint64_t swSize, swEnd;
    int circular;
    uint64_t locks[MAX_LOCKS];           /* workersSwEndPositions (deque) */
    int lockHead, lockCount;
    int prefilter;
};

/* utils/Hashes.h:28-40 — the u64 accumulator's low 32 bits depend only on low 32 bits of every
 * step (xor, wrap-around multiply), so u32 arithmetic is exact. */
uint32_t orc_hash(const uint8_t *s, int K) {
    uint32_t h = (uint32_t) K;
    for (uint32_t j = 0; j < (uint32_t) K / 4; j++) {
        uint32_t k;
        memcpy(&k, s + 4 * j, 4);
        k += j;
        h ^= k;
        h *= 171717u;
    }
    return h;
}

static inline uint32_t hash_masked(const orc_matcher *m, const uint8_t *s) {
    return orc_hash(s, m->K) & m->mask;
}

/* initParams, .cpp:74-104 (minMatchLength defaults to L, .cpp:342-343) */
static void init_params(orc_matcher *m) {
    int L = m->L;
    if (L > 110) m->K = 56;
    else if (L > 62) m->K = 44;
    else if (L > 53) m->K = 40;
    else if (L > 46) m->K = 36;
    else if (L > 42) m->K = 32;
    else if (L > 32) m->K = 28;
    else m->K = (L / 4 - 1) * 4;
    int KmmL = (L / 4 - 1) * 4;
    if (KmmL < m->K) m->K = KmmL;
    m->LK2 = (L - m->K) / 2;
    m->K_PLUS_LK24 = m->K + m->LK2 - 4;
    uint8_t i = HASH_SIZE_MIN_ORDER;
    do {
        m->hash_size = ((uint32_t) 1) << (i++);
    } while (i <= HASH_SIZE_MAX_ORDER && m->hash_size < m->maxRefLength / (uint64_t) m->k1);
    m->mask = m->hash_size - 1;
}

orc_matcher *orc_matcher_create(uint64_t maxRefLength, int L, int k1, int k2, int skipMargin) {
    if (k1 <= 0 || L < 16) return NULL;                       /* :82-85; an odd k1 is the base class (identity encoding), MGMP.cpp:170-176 */
    orc_matcher *m = (orc_matcher *) calloc(1, sizeof(*m));
    m->maxRefLength = maxRefLength;
    m->L = L; m->k1 = k1; m->k2 = k2; m->skipMargin = skipMargin;
    m->k1ord = __builtin_ctz((unsigned) k1);                  /* .cpp:498; 0 for an odd k1 = htEncodePos / htDecodePos the identity, .h:74-76 */
    m->ref = (uint8_t *) calloc(maxRefLength + REF_SLACK, 1);
    m->ref[0] = 0;                                            /* .cpp:335 */
    m->pos1 = REF_SHIFT;                                      /* .cpp:337 */
    m->swEnd = maxRefLength;                                  /* .cpp:338 */
    m->swSize = maxRefLength / SW_WIDTH_FACTOR;               /* .cpp:339 */
    m->circular = 1;
    init_params(m);
    m->samplingPos = (k1 % 2) ? REF_SHIFT : (uint64_t) k1;    /* .cpp:503 (Exp variant); the base class keeps its initialiser, .h:78 */
    m->ht = (uint32_t *) calloc(m->hash_size, sizeof(uint32_t));
    m->prefilter = 1;
    return m;
}

void orc_matcher_destroy(orc_matcher *m) {
    if (!m) return;
    free(m->ht); free(m->ref); free(m);
}

void orc_disable_sliding_window(orc_matcher *m) { m->swSize = 0; m->swEnd = m->circular ? 0 : m->maxRefLength; }
void orc_set_sliding_window_size(orc_matcher *m, int f) { m->swSize = m->maxRefLength / (uint64_t) (uint8_t) f; }
void orc_disable_circular_buffer(orc_matcher *m) { m->circular = 0; m->swEnd = m->maxRefLength; }
uint64_t orc_ref_length(const orc_matcher *m) { return m->laps ? m->maxRefLength : (uint64_t) m->pos1; }
uint64_t orc_loading_position(const orc_matcher *m) { return (uint64_t) m->pos1; }
uint64_t orc_loaded_ref_length(const orc_matcher *m) {
    return (uint64_t) m->laps * (m->maxRefLength - REF_SHIFT) + ((uint64_t) m->pos1 - REF_SHIFT);
}
uint64_t orc_max_ref_length(const orc_matcher *m) { return m->maxRefLength; }
void orc_set_position(orc_matcher *m, uint64_t pos, int laps) { m->pos1 = (int64_t) pos; m->laps = laps; }
uint32_t orc_hash_size(const orc_matcher *m) { return m->hash_size; }
const uint32_t *orc_ht(const orc_matcher *m) { return m->ht; }
const uint8_t *orc_ref(const orc_matcher *m) { return m->ref; }
int orc_K(const orc_matcher *m) { return m->K; }
void orc_set_prefilter(orc_matcher *m, int on) { m->prefilter = on; }
void orc_free(void *p) { free(p); }

/* acquireWorkerMatchingLockPos, .cpp:361-378 */
uint64_t orc_acquire_lock(orc_matcher *m) {
    if (m->swSize == 0 || !m->circular) return m->swEnd;
    uint64_t w = (uint64_t) m->pos1 + m->swSize;
    if (m->laps || w > m->maxRefLength) {
        if (w > m->maxRefLength) w -= m->maxRefLength - REF_SHIFT;
    } else
        w = m->maxRefLength;
    if (m->lockCount == 0) m->swEnd = w;
    if (m->lockCount == MAX_LOCKS) { fprintf(stderr, "oracle: too many locks\n"); abort(); }
    m->locks[(m->lockHead + m->lockCount++) % MAX_LOCKS] = w;
    return w;
}

/* releaseWorkerMatchingLockPos, .cpp:380-400 */
int orc_release_lock(orc_matcher *m, uint64_t v) {
    if (m->swSize == 0 || !m->circular) return 0;
    int i = 0;
    while (i < m->lockCount && m->locks[(m->lockHead + i) % MAX_LOCKS] != v) i++;
    if (i == m->lockCount) return -1;                         /* reference: message + exit(EXIT_FAILURE) */
    if (i == 0) {
        do {
            m->lockHead = (m->lockHead + 1) % MAX_LOCKS; m->lockCount--;
        } while (m->lockCount && m->locks[m->lockHead] == ORC_NO_LOCK);
        if (m->lockCount) m->swEnd = m->locks[m->lockHead];
    } else
        m->locks[(m->lockHead + i) % MAX_LOCKS] = ORC_NO_LOCK;
    return 0;
}

/* processIgnoreCollisionsRef, .cpp:146-171, single-thread order = ascending position inside the
 * main loop, then the grid-aligned tail; later writers overwrite earlier ones. */
static void insert_samples(orc_matcher *m) {
    const int64_t STEP = (int64_t) m->k1 * 128;
    const int64_t E = m->pos1 - m->K;
    int64_t i1;
    for (i1 = (int64_t) m->samplingPos; i1 < E - STEP; i1 += STEP) {
        int64_t i2 = i1;
        for (int t = 0; t < 128; t++, i2 += m->k1)
            m->ht[hash_masked(m, m->ref + i2)] = (uint32_t) ((uint64_t) i2 >> m->k1ord);
    }
    for (i1 = m->k1 + ((E - 1) / STEP) * STEP; i1 < E + 1; i1 += m->k1)
        m->ht[hash_masked(m, m->ref + i1)] = (uint32_t) ((uint64_t) i1 >> m->k1ord);
    m->samplingPos = (uint64_t) i1;
}

/* utils/helper.cpp:312-338: identity except the listed symbols; the constructor's loops run
 * i < CHAR_MAX, so entry 127 of the (static, zero-initialised) table stays 0. */
static uint8_t g_uclut[256];
static int g_uclut_ready = 0;
static void init_uclut(void) {
    for (int i = 0; i < 256; i++) g_uclut[i] = (uint8_t) i;
    g_uclut[127] = 0;
    const char *from = "AaCcGgTtNnUuYyRrKkMmBbDdHhVvWwSs";
    const char *to   = "TTGGCCAANNAARRYYMMKKVVHHDDBBSSWW";
    for (int i = 0; from[i]; i++) g_uclut[(uint8_t) from[i]] = (uint8_t) to[i];
    g_uclut_ready = 1;
}

/* utils/helper.cpp:405-410 */
void orc_upper_reverse_complement(const uint8_t *src, uint64_t n, uint8_t *dst) {
    if (!g_uclut_ready) init_uclut();
    for (uint64_t i = 0; i < n; i++) dst[n - 1 - i] = g_uclut[src[i]];
}

/* private loadRef, .cpp:402-437 (tail recursion unrolled into a loop) */
static void load_ref_piecewise(orc_matcher *m, const uint8_t *text, uint64_t len, int rc, int addSep, int sep) {
    while (len != 0) {
        if ((uint64_t) m->pos1 == m->maxRefLength && m->swEnd != m->maxRefLength) {
            m->laps++;
            m->pos1 = REF_SHIFT;
            m->samplingPos = REF_SHIFT;
        }
        uint64_t tmpEnd = m->swEnd;
        uint64_t tmpLength = len;
        uint64_t tmpMax = tmpEnd < (uint64_t) m->pos1 ? m->maxRefLength : tmpEnd;
        if ((uint64_t) m->pos1 + tmpLength > tmpMax) tmpLength = tmpMax - (uint64_t) m->pos1;
        if (rc)
            orc_upper_reverse_complement(text + len - tmpLength, tmpLength, m->ref + m->pos1);
        else
            memcpy(m->ref + m->pos1, text, tmpLength);
        if (addSep && (uint64_t) m->pos1 + tmpLength == m->swEnd) m->ref[m->swEnd - 1] = (uint8_t) sep;
        m->pos1 += (int64_t) tmpLength;
        insert_samples(m);
        text += rc ? 0 : tmpLength;
        len = (uint64_t) m->pos1 == tmpEnd ? 0 : len - tmpLength;
    }
}

/* public loadRef, .cpp:453-458 */
void orc_load_ref(orc_matcher *m, const uint8_t *t, uint64_t n, int loadRC, int addSep, int sep) {
    load_ref_piecewise(m, t, n, 0, addSep, sep);
    if (loadRC) load_ref_piecewise(m, t, n, 1, addSep, sep);
}

/* loadSeparator, .cpp:439-451 */
void orc_load_separator(orc_matcher *m, int sep) {
    if ((uint64_t) m->pos1 == m->maxRefLength && m->swEnd != m->maxRefLength) {
        m->laps++;
        m->pos1 = REF_SHIFT;
        m->samplingPos = REF_SHIFT;
    }
    if ((uint64_t) m->pos1 == m->maxRefLength) return;
    if ((uint64_t) m->pos1 == m->swEnd)
        m->ref[m->pos1 - 1] = (uint8_t) sep;
    else
        m->ref[m->pos1++] = (uint8_t) sep;
}

/* processExactMatchQueryIgnoreCollisionsTight<uint32_t,true>, .cpp:181-321.
 * PAIRED_MATCH_LENGTH_LOSS_LIMIT = 0 (:250) makes every `replacedMatches` statement dead. */
uint64_t orc_match_texts(orc_matcher *m, const uint8_t *q, uint64_t N2, uint32_t minMatchLength,
                         uint64_t lockPos, orc_match **out, uint64_t *stats) {
    const int K = m->K, k2 = m->k2;
    const uint8_t *ref = m->ref;
    orc_match *res = NULL;
    uint64_t nres = 0, cap = 0;
    uint64_t probes = 0, hits = 0;
    uint32_t l1 = 0, l2 = 0, r1 = 0, r2 = 0;
    *out = NULL;
    if (minMatchLength < (uint32_t) K) return 0;                               /* :480-483: reference exits */
    for (int64_t i2 = 0; (uint64_t) (i2 + K) < N2 + 1; i2 += k2) {             /* :200 */
        uint32_t j = hash_masked(m, q + i2);
        if (i2 - m->LK2 >= 0) memcpy(&l2, q + i2 - m->LK2, 4);                 /* :203 */
        if ((uint64_t) i2 + (uint64_t) m->K_PLUS_LK24 + 4 <= N2) memcpy(&r2, q + i2 + m->K_PLUS_LK24, 4); /* :204 */
        probes++;
        if (m->ht[j] == 0) continue;                                           /* :208 */
        const int64_t c = (int64_t) ((uint64_t) m->ht[j] << m->k1ord);         /* :211 */
        const int64_t swStart = m->pos1;
        const int64_t swStop = lockPos != ORC_NO_LOCK ? (int64_t) lockPos : m->pos1;
        const int endsBefore = c + K < swStart;
        const int startsBefore = c < swStop;
        if (swStart <= swStop) {
            if (!endsBefore && startsBefore) continue;
        } else if (!endsBefore || startsBefore)
            continue;
        const int64_t tmpEnd1 = endsBefore ? swStart : (int64_t) orc_ref_length(m);
        const int64_t tmpStart1 = startsBefore ? 0 : swStop;
        hits++;
        memcpy(&l1, ref + c - m->LK2, 4);                                      /* :223 */
        memcpy(&r1, ref + c + m->K_PLUS_LK24, 4);                              /* :224 */
        if (m->prefilter && !(r1 == r2 || l1 == l2)) continue;                 /* :226 */
        /* right extension, :227-246 == plain LCP bounded by tmpEnd1 and the query end */
        int64_t p1 = c + K, p2 = i2 + K;
        while (p1 != tmpEnd1 && (uint64_t) p2 != N2 && ref[p1] == q[p2]) { p1++; p2++; }
        const int64_t right1 = p1;
        p1 = c; p2 = i2;
        int64_t keep = (int64_t) nres;                                         /* resSizeWithoutOverlapped */
        while (keep--) {                                                       /* :253 */
            const orc_match *lm = &res[keep];
            const int64_t lmEnd = (int64_t) (lm->posDestText + lm->length);
            if (lmEnd < p2) {                                                  /* :255 */
                const int64_t g = lmEnd;                                       /* tmpguard2 */
                while (p1 != tmpStart1 && p2 > g - 1 && ref[p1] == q[p2]) { p1--; p2--; }
                if (p2 > g - 1) break;                                         /* :260-261 */
                p1++; p2++;
            }
            const int64_t lastDelta = p2 - (int64_t) lm->posDestText;          /* :264 */
            if (p1 - tmpStart1 < lastDelta || lm->length > OVERLAP_MATCH_MAX_LENGTH ||
                memcmp(q + p2 - lastDelta, ref + p1 - lastDelta, (size_t) lastDelta) != 0) {   /* strcmplcp != 0 <=> differ */
                p1--; p2--;
                break;
            }
            p2 -= lastDelta; p1 -= lastDelta;
        }
        if (keep < 0) {                                                        /* :277-280 */
            while (p1 != tmpStart1 && p2 > -1 && ref[p1] == q[p2]) { p1--; p2--; }
        } else {
            const int64_t overlap = (int64_t) (res[keep].posDestText + res[keep].length) - (p2 + 1);   /* :285 */
            if (overlap > 0) { p1 += overlap; p2 += overlap; }
        }
        ++keep;
        if (right1 - p1 > (int64_t) minMatchLength && memcmp(ref + c, q + i2, (size_t) K) == 0) {     /* :298 */
            nres = (uint64_t) keep;
            if (nres == cap) { cap = cap ? cap * 2 : 1024; res = (orc_match *) realloc(res, cap * sizeof(*res)); }
            res[nres].posSrcText = (uint64_t) (p1 + 1);
            res[nres].length = (uint64_t) (right1 - p1 - 1);
            res[nres].posDestText = (uint64_t) (p2 + 1);
            res[nres].nextSrcRegionLoadingPos = 0;
            nres++;
            int skip = (int) (((p2 + right1 - p1) - i2) / k2 * k2);            /* :308 */
            skip -= skip > m->skipMargin ? m->skipMargin : skip;
            if (skip) i2 += skip - k2;
        }
    }
    if (stats) {
        uint64_t tot = 0;
        for (uint64_t i = 0; i < nres; i++) tot += res[i].length;
        stats[0] = probes; stats[1] = hits; stats[2] = tot;
    }
    *out = res;
    return nres;
}
