/* TEST INFRASTRUCTURE ONLY — CPU restatement (plain C) of MBGC's reference-based match-finding hot
 * path. It is the checker for the HIP path and the `cpu_baseline` leg of bench.py; the product
 * (mbgc_amd/) never links, imports or calls anything in this directory.
 *
 * Parity status: PINNED. Every function is checked against the reference itself compiled from
 * /root/reference into oracle/_ref (tests/test_oracle_vs_ref.py, run in the build container) and
 * against the golden fixtures under tests/golden/ generated from that reference build
 * (tests/golden/make_golden.py), which travel to the GPU box. decode_oracle.c (the decoder's per-contig automaton,
 * used for the round-trip property) is pinned by decoding streams written by that reference build's encoder
 * (tests/test_decode_roundtrip.py).
 *
 * All file:line citations are relative to /root/reference (MBGC v2.1.5).
 */
#ifndef MBGC_ORACLE_H
#define MBGC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NO_LOCK UINT64_MAX          /* SW_END_ERASED_FLAG, SlidingWindowSparseEMMatcher.h:46 */
#define ORC_SKIPPED UINT64_MAX          /* PROCESSING_MATCHES_SKIPPED_..., MultipleGenomeMatchingProcessor.h:97 */

/* TextMatch, matching/TextMatchers.h:9-16 */
typedef struct {
    uint64_t posSrcText, length, posDestText, nextSrcRegionLoadingPos;
} orc_match;

typedef struct orc_matcher orc_matcher;

/* maRushPrime1HashSimplified<K>, utils/Hashes.h:28-40 (unmasked) */
uint32_t orc_hash(const uint8_t *s, int K);

/* SlidingWindowExpSparseEMMatcher ctor, SlidingWindowSparseEMMatcher.cpp:325-359,494-519.
 * Sizes are explicit: no RAM probing (helper.h:324-339 is host policy). k1 must be even. */
orc_matcher *orc_matcher_create(uint64_t maxRefLength, int L, int k1, int k2, int skipMargin);
void orc_matcher_destroy(orc_matcher *m);
void orc_disable_sliding_window(orc_matcher *m);             /* .h:93 */
void orc_set_sliding_window_size(orc_matcher *m, int factor);/* .h:97 */
void orc_disable_circular_buffer(orc_matcher *m);            /* .h:95 */
void orc_load_ref(orc_matcher *m, const uint8_t *t, uint64_t n, int loadRC, int addSep, int sep); /* .cpp:453-458 */
void orc_load_separator(orc_matcher *m, int sep);            /* .cpp:439-451 */
uint64_t orc_ref_length(const orc_matcher *m);               /* .h:106 */
uint64_t orc_loading_position(const orc_matcher *m);         /* .h:107 */
uint64_t orc_loaded_ref_length(const orc_matcher *m);        /* .h:108 */
uint64_t orc_max_ref_length(const orc_matcher *m);           /* .h:105 */
void orc_set_position(orc_matcher *m, uint64_t pos, int laps);/* .h:110-113 */
uint64_t orc_acquire_lock(orc_matcher *m);                   /* .cpp:361-378 */
int orc_release_lock(orc_matcher *m, uint64_t v);            /* .cpp:380-400; -1 = invalid lock value */
uint32_t orc_hash_size(const orc_matcher *m);
const uint32_t *orc_ht(const orc_matcher *m);
const uint8_t *orc_ref(const orc_matcher *m);
int orc_K(const orc_matcher *m);
void orc_set_prefilter(orc_matcher *m, int on);              /* .cpp:203-204,223-226 (result-neutral) */

/* matchTexts, .cpp:478-492 -> :181-321. Returns the number of matches; *out is malloc'd (free with
 * orc_free). stats (may be NULL): [0] probes, [1] non-empty buckets that passed the window test,
 * [2] sum of match lengths. */
uint64_t orc_match_texts(orc_matcher *m, const uint8_t *q, uint64_t n, uint32_t minMatchLength,
                         uint64_t lockPos, orc_match **out, uint64_t *stats);
void orc_free(void *p);

/* upperReverseComplement, utils/helper.cpp:405-410 with the LUT of :312-338 */
void orc_upper_reverse_complement(const uint8_t *src, uint64_t n, uint8_t *dst);
/* ContextAwareMismatchesCoder::mismatch2code, coders/ContextAwareMismatchesCoder.cpp:65-70 */
uint8_t orc_mismatch2code(uint8_t actual, uint8_t mismatch);

/* ---- emission: MBGC_Encoder::processMatches & friends, mbgccoder/MBGC_Encoder.cpp:128-427 ---- */
typedef struct {
    int enableExtensionsWithMismatches;        /* MBGC_Params.h:76 (true) */
    int mismatchesWithExclusion;               /* :77 (true) */
    int lazyDecompressionSupport;              /* :38 (true unless single-file mode) */
    int enable40bitReference;                  /* MGMP_Params.h:208, cleared by MGMP.cpp:159-160 */
    int frugal64bitLenEncoding;                /* MBGC_Params.h:73 (false in -m0, :901) */
    int gapDepthOffsetEncoding;                /* :79 (64) */
    int gapDepthMismatchesEncoding;            /* :80 (2) */
    uint64_t gapBreakingMatchMinLength;        /* :81 (256) */
    int mmsMatchBonus, mmsMismatchPenalty, mmsMismatchesScoreThreshold, mmsMismatchesInitialScore; /* :92-97 */
    int allowedTargetsOutrunForDissimilarContigs;          /* MGMP_Params.h:78 */
    uint64_t minimalLengthForDissimilarContigs;            /* :79 */
    int unmatchedFractionFactorTweakForDissimilarContigs;  /* :80 */
} orc_emit_params;

void orc_emit_params_default(orc_emit_params *p, int mode /* -m 0..3 */);

typedef struct {
    uint8_t *data;
    uint64_t size, cap;
} orc_buf;

/* The six per-target streams, in the order the encoder enrols them (MBGC_Encoder.cpp:779-787). */
enum { ORC_LIT = 0, ORC_OFF = 1, ORC_OFF5 = 2, ORC_LEN = 3, ORC_GAP = 4, ORC_FLAGS = 5, ORC_NSTREAMS = 6 };

typedef struct {
    orc_buf s[ORC_NSTREAMS];
    /* counters, MBGC_Encoder.cpp:293-306 */
    uint64_t unmatchedChars, extensionsMatchedChars, extensionsMismatches, totalMatched,
             totalDestOverlap, totalDestLen, removedGapBreakingMatches;
} orc_streams;

void orc_streams_init(orc_streams *s);
void orc_streams_free(orc_streams *s);

/* processMatches. `matches` is modified in place like the reference's vector (gap-breaking matches
 * removed, *nmatches updated, nextSrcRegionLoadingPos filled). refExtLoadedPos/nLoaded is
 * refExtLoadedPosArr (lazy mode only). Returns unmatchedChars or ORC_SKIPPED. */
uint64_t orc_process_matches(const orc_matcher *m, const orc_emit_params *p, orc_match *matches,
                             uint64_t *nmatches, const uint8_t *dest, uint64_t destLen,
                             uint64_t lockPos, int unmatchedFractionFactor, int64_t processedTargetsCount,
                             int64_t targetIdx, const uint64_t *refExtLoadedPos, uint64_t nLoaded,
                             orc_streams *out);

/* The inverse (decode_oracle.c): MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars with extendMatchLeft/Right,
 * mbgccoder/MBGC_Decoder.cpp:319-523, for the six streams of ONE contig (ORC_* order) against the reference buffer
 * `ref` the encoder matched against. Writes the contig to dest; returns unmatchedChars, or -1 if a stream ran
 * out, an index left its buffer, dest is too small or stream bytes were left over. */
int64_t orc_decode_contig(const uint8_t *ref, const orc_emit_params *p, const uint8_t *const streams[ORC_NSTREAMS],
                          const uint64_t sizes[ORC_NSTREAMS], uint64_t refLockPos, uint8_t *dest, uint64_t destCap,
                          uint64_t *destLen);

/* writeUInt64Frugal, utils/helper.cpp:237-246 */
void orc_write_frugal64(orc_buf *b, uint64_t v);
void orc_buf_put(orc_buf *b, const void *p, uint64_t n);

/* ---- the -m3 reverse-complement pass over the literal stream (rcmatch_oracle.c): SimpleSequenceMatcher::rcMatchSequence
 * on CopMEMMatcher, matching/SimpleSequenceMatcher.cpp:165-176, matching/copmem/CopMEMMatcher.cpp (single-thread form) */
uint32_t orc_hash_sparsified(const uint8_t *s, int K);                         /* maRushPrime1HashSparsified<K>, utils/Hashes.h:47-68 */
void orc_reverse_complement(const uint8_t *src, uint64_t n, uint8_t *dst);     /* PgHelpers::reverseComplement, utils/helper.cpp:429-437 */
uint64_t orc_rc_find_matches(const uint8_t *seq, uint64_t n, uint32_t targetMatchLength, uint32_t minMatchLength,
                             orc_match **out, int *params, uint64_t *charExtensions);
uint64_t orc_rc_apply_matches(uint8_t *seq, uint64_t n, orc_match *m, uint64_t nm, uint32_t targetMatchLength, uint32_t minMatchLength,
                              orc_buf *mapOff, orc_buf *mapLen, uint64_t *stats);
uint64_t orc_rc_match_sequence(uint8_t *seq, uint64_t n, uint32_t targetMatchLength, uint32_t minMatchLength,
                               orc_buf *mapOff, orc_buf *mapLen, uint64_t *stats);

#ifdef __cplusplus
}
#endif
/* ---- input stage (fasta_oracle.c): kseq_read_lossless_fasta over a whole file, utils/kseq.h:233-274 */
typedef struct { uint64_t headerOff, headerLen, seqOff, seqLen; } orc_fasta_record;
uint64_t orc_fasta_parse(const uint8_t *file, uint64_t n, int uppercase, uint8_t *seqOut, orc_fasta_record *rec, uint64_t recCap,
                         uint64_t *seqBytes, uint64_t *dnaLineLen, int *status);

#endif
