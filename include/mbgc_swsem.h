/* mbgc_swsem.h — C ABI of the MI355X-native MBGC match-finding hot path (libmbgc_hip.so).
 *
 * Drop-in boundary: the reference has no FFI; its seam is the C++ class SlidingWindowSparseEMMatcher
 * (constructed at matching/MultipleGenomeMatchingProcessor.cpp:170-176) plus
 * MBGC_Encoder::processMatches (mbgccoder/MBGC_Encoder.cpp:143-308). Every entry point below names
 * the reference member it replaces (file:line under the reference tree). The C++ facade in
 * mbgc_amd/host/ keeps the reference's class and method names and forwards here; INTEGRATION.md
 * shows the binding a maintainer adds to the reference.
 *
 * Conventions: plain pointers and sizes, no exceptions, no torch types. Every function that can fail
 * returns 0 on success and a negative SWSEM_E* code otherwise; swsem_last_error() gives the message
 * the reference would have printed before exit(EXIT_FAILURE) — the facade performs the exit.
 * Pointers named *_dev are HIP device pointers on the handle's device; all others are host pointers.
 * All calls on one handle must be serialised by the caller (the reference guards the same state with
 * `omp critical`, SlidingWindowSparseEMMatcher.cpp:365,383); different handles are independent.
 */
#ifndef MBGC_SWSEM_H
#define MBGC_SWSEM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWSEM_NO_LOCK   UINT64_MAX   /* SW_END_ERASED_FLAG, SlidingWindowSparseEMMatcher.h:46 */
#define SWSEM_SKIPPED   UINT64_MAX   /* PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY, MGMP.h:97 */

#define SWSEM_OK         0
#define SWSEM_EINVAL    -1   /* bad argument (reference: message + exit) */
#define SWSEM_ENOMEM    -2   /* device allocation failed */
#define SWSEM_EHIP      -3   /* HIP runtime error */
#define SWSEM_ELOCK     -4   /* invalid worker lock value, SlidingWindowSparseEMMatcher.cpp:388-391 */
#define SWSEM_ENODEV    -5   /* no usable gfx950 device: the product never falls back to a CPU path */

typedef struct swsem swsem_t;

/* TextMatch, matching/TextMatchers.h:9-16 (nextSrcRegionLoadingPos is emission scratch, kept device-side) */
typedef struct {
    uint64_t posSrcText, length, posDestText;
} swsem_match_t;

const char *swsem_last_error(void);
int swsem_device_count(void);
/* NUMA node of the host memory closest to the device (its PCIe function's numa_node in sysfs), -1 when unknown: a host
 * that reads files into page-locked memory for this device keeps its threads there */
int swsem_device_numa_node(int device);

/* SlidingWindowExpSparseEMMatcher::SlidingWindowExpSparseEMMatcher, SlidingWindowSparseEMMatcher.cpp:494-519
 * (+ base ctor :325-359, initParams :74-104). maxRefLength is explicit: the 60 %-of-RAM cap of
 * utils/helper.h:324-339 is host policy. k1 must be even, k2 must be 1 (the only values MBGC uses). */
int swsem_create(swsem_t **out, uint64_t maxRefLength, int L, int k1, int k2, int skipMargin, int device);
void swsem_destroy(swsem_t *h);                                   /* ~SlidingWindowSparseEMMatcher, .cpp:460-467 */
/* Run every launch of this handle on an existing HIP stream (hipStream_t), e.g. the caller's. */
int swsem_set_stream(swsem_t *h, void *hip_stream);
int swsem_synchronize(swsem_t *h);

void swsem_disable_sliding_window(swsem_t *h);                    /* .h:93 */
void swsem_set_sliding_window_size(swsem_t *h, int factor);       /* .h:97 */
void swsem_disable_circular_buffer(swsem_t *h);                   /* .h:95 */
uint64_t swsem_get_ref_length(const swsem_t *h);                  /* getRefLength, .h:106 */
uint64_t swsem_get_loading_position(const swsem_t *h);            /* getLoadingPosition, .h:107 */
uint64_t swsem_get_loaded_ref_length(const swsem_t *h);           /* getLoadedRefLength, .h:108 */
uint64_t swsem_get_max_ref_length(const swsem_t *h);              /* getMaxRefLength, .h:105 */
/* swSize (.h:47,97; 0: no window — sequential matching or a buffer that is not circular): what the targets that hold their
 * lock positions together may load between them once the buffer has wrapped; the caller sizes its rounds by it */
uint64_t swsem_get_sliding_window_size(const swsem_t *h);
/* extension bytes loadRef has given up so far because the loader reached the oldest outstanding lock position
 * (.cpp:412-417,433: `seqLength = pos1 == tmpEnd ? 0 : ...`) — 0 for rounds that fit the window */
uint64_t swsem_get_dropped_bytes(const swsem_t *h);
void swsem_set_position(swsem_t *h, uint64_t refPos, int reachedRefLengthCount);  /* setPosition, .h:110-113 */
uint64_t swsem_acquire_lock(swsem_t *h);                          /* acquireWorkerMatchingLockPos, .cpp:361-378 */
int swsem_release_lock(swsem_t *h, uint64_t lockValue);           /* releaseWorkerMatchingLockPos, .cpp:380-400 */
int swsem_get_K(const swsem_t *h);
uint32_t swsem_get_hash_size(const swsem_t *h);

/* loadRef(refText, refLength, loadRCRef, addRegionSeparators, regionSeparator), .cpp:453-458 (-> :402-437,
 * :146-171, utils/helper.cpp:405-410). Must be applied identically, in call order, on every replica. */
int swsem_load_ref(swsem_t *h, const uint8_t *text, uint64_t len, int loadRC, int addSep, int sep);
int swsem_load_ref_dev(swsem_t *h, const uint8_t *text_dev, uint64_t len, int loadRC, int addSep, int sep);
int swsem_load_separator(swsem_t *h, int sep);                    /* loadSeparator, .cpp:439-451 */
/* finalizeParallelProcessingOfTarget for n targets in target order (MGMP.cpp:440-457 + MBGC_Encoder.cpp:557-562):
 * per target loadRef(ext, len, false, addSep, sep), then — lazy mode — loadSeparator(sep), then
 * releaseWorkerMatchingLockPos(lockPos[i]) (lockPos may be NULL). loadedAfter[i] receives
 * getLoadedRefLength() after target i. One call per round instead of 3 n. The copies are queued on the
 * handle's stream and the call returns: the ext_dev buffers must stay valid until swsem_synchronize() or any
 * later call that hands results back to the host (swsem_batch_counts, swsem_emit_batch, swsem_match ...). */
int swsem_finalize_targets(swsem_t *h, int n, const uint8_t *const *ext_dev, const uint64_t *ext_len, int addSep, int sep,
                           int lazySeparator, const uint64_t *lockPos, uint64_t *loadedAfter);
/* PgHelpers::upperReverseComplement(src, n, dst) on device buffers, utils/helper.cpp:405-410 — what
 * processTarget uses to append a contig's reverse complement to the target's extension string
 * (MGMP.cpp:393-398) */
int swsem_revcomp_dev(swsem_t *h, const uint8_t *src_dev, uint64_t n, uint8_t *dst_dev);

/* matchTexts(resMatches, destText, destLen, false, false, minMatchLength, matchingLockPos), .cpp:478-492.
 * *matches points into a handle-owned buffer that stays valid until the next match call on the
 * handle (the reference fills a caller-owned vector it clears on entry, .cpp:484). */
int swsem_match(swsem_t *h, const uint8_t *query, uint64_t len, uint32_t minMatchLength, uint64_t lockPos,
                const swsem_match_t **matches, uint64_t *nmatches);

/* A round: n contigs matched against the frozen reference in one pass (what the reference's worker
 * threads do concurrently, MGMP.cpp:340-403). queries_dev holds the contigs back to back, contig c at
 * [offsets[c], offsets[c+1]); offsets/lockPos are host arrays. Results stay in HBM (for swsem_emit_batch)
 * and can be fetched per contig. */
int swsem_match_batch_dev(swsem_t *h, const uint8_t *queries_dev, const uint64_t *offsets, int n,
                          uint32_t minMatchLength, const uint64_t *lockPos);
int swsem_batch_counts(swsem_t *h, uint64_t *nmatches /* [n] */);
int swsem_batch_matches(swsem_t *h, int contig, swsem_match_t *out, uint64_t cap);
/* order-sensitive fingerprint of all match rows of the batch, computed on the device copy */
int swsem_batch_fingerprint(swsem_t *h, uint64_t *fp, uint64_t *total_matches, uint64_t *total_length);

/* ---- emission: MBGC_Encoder::processMatches (+extendMatchLeft/Right, mismatch2code), MBGC_Encoder.cpp:143-427 */
typedef struct {
    int enableExtensionsWithMismatches;   /* MBGC_Params.h:76 */
    int mismatchesWithExclusion;          /* :77 */
    int lazyDecompressionSupport;         /* :38 */
    int enable40bitReference;             /* MGMP_Params.h:208 / MGMP.cpp:159-166 */
    int frugal64bitLenEncoding;           /* MBGC_Params.h:73 */
    int gapDepthOffsetEncoding;           /* :79 */
    int gapDepthMismatchesEncoding;       /* :80 */
    uint64_t gapBreakingMatchMinLength;   /* :81 */
    int mmsMatchBonus, mmsMismatchPenalty, mmsMismatchesScoreThreshold, mmsMismatchesInitialScore; /* :92-97 */
    int allowedTargetsOutrunForDissimilarContigs;           /* MGMP_Params.h:78 */
    uint64_t minimalLengthForDissimilarContigs;             /* :79 */
    int unmatchedFractionFactorTweakForDissimilarContigs;   /* :80 */
} swsem_emit_params_t;
void swsem_emit_params_default(swsem_emit_params_t *p, int mode /* -m 0..3, MBGC_Params.h:886-922 */);

enum { SWSEM_LIT = 0, SWSEM_OFF = 1, SWSEM_OFF5 = 2, SWSEM_LEN = 3, SWSEM_GAP = 4, SWSEM_FLAGS = 5, SWSEM_NSTREAMS = 6 };

typedef struct {
    const uint8_t *data[SWSEM_NSTREAMS];  /* host pointers into a handle-owned buffer (valid until the next emit call) */
    uint64_t size[SWSEM_NSTREAMS];
    uint64_t unmatchedChars;              /* return value of processMatches, or SWSEM_SKIPPED */
    /* the counters processMatches adds to atomically, MBGC_Encoder.cpp:293-306 */
    uint64_t extensionsMatchedChars, extensionsMismatches, totalMatched, removedGapBreakingMatches;
    uint64_t nmatches;                    /* matches left after the gap-breaking pass */
} swsem_streams_t;

/* processMatches for contig `contig` of the last swsem_match_batch_dev (or contig 0 of the last swsem_match).
 * refExtLoadedPos/nLoaded = the encoder's refExtLoadedPosArr (lazy mode). processedTargetsCount/targetIdx/
 * unmatchedFractionFactor feed the dissimilarity early-out (:202-205). */
int swsem_emit(swsem_t *h, const swsem_emit_params_t *p, int contig, uint64_t lockPos,
               int unmatchedFractionFactor, int64_t processedTargetsCount, int64_t targetIdx,
               const uint64_t *refExtLoadedPos, uint64_t nLoaded, swsem_streams_t *out);

/* The same for several contigs of the batch in one pass; contigIdx == NULL means contigs 0..n-1; the
 * per-contig arrays may be NULL (no lock, factor 128, processed 0, target 0). Fetch with swsem_emit_result. */
int swsem_emit_batch(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                     const int *unmatchedFractionFactor, const int64_t *processedTargetsCount, const int64_t *targetIdx,
                     const uint64_t *refExtLoadedPos, uint64_t nLoaded);
/* swsem_emit_batch in two steps, for callers that overlap. _begin returns as soon as the part the caller's
 * control flow depends on is known — processMatches' return value per contig incl. the dissimilarity verdict
 * (swsem_emit_unmatched), i.e. what MGMP.cpp:382-399 decides the retry and the reference extension on — with
 * the byte-level work (pairing, mismatch extensions, stream bytes) queued on a second stream. Between _begin
 * and _end the caller may finalize the round (swsem_finalize_targets) and start the next one
 * (swsem_match_batch_dev): the reference's workers run processMatches next to the finalizer in the same way
 * (MGMP.cpp:520-555). The query buffer of the emitted batch must stay untouched until _end. Once the
 * reference buffer has wrapped, loads wait for the running emission (they would overwrite text it reads).
 * _end (also implied by swsem_emit_result / swsem_emit_pack_dev) waits for the streams. */
int swsem_emit_batch_begin(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                           const int *unmatchedFractionFactor, const int64_t *processedTargetsCount, const int64_t *targetIdx,
                           const uint64_t *refExtLoadedPos, uint64_t nLoaded);
/* swsem_emit_batch_begin with a speculative finalize: the caller predicts, per emitted contig, how
 * isContigProperForRefExtension / isContigProperForRefRCExtension (MGMP_Params.h:178-190, factors as in MGMP.cpp:389-398)
 * will come out, and hands over the swsem_finalize_targets arguments that follow from the prediction. The copies
 * and the table insertion are queued directly behind pass 1, gated by a device-side check of the prediction, so
 * they need not wait for the host's round trip. *applied = 1: every decision came out as predicted (and no contig
 * was given up as dissimilar): the finalize has been done — loadedAfter is filled, the locks are released — and
 * must not be repeated. *applied = 0: nothing has changed, device or host; finalize as usual. */
typedef struct {
    int ntargets;
    const uint8_t *const *ext_dev; const uint64_t *ext_len;   /* as swsem_finalize_targets */
    int addSep, sep, lazySeparator;
    const uint64_t *lockPos;
    uint64_t *loadedAfter;
    const uint8_t *predExt, *predRC;                            /* [n emitted contigs] predicted decisions */
    int factor, rcFactor;                                       /* unmatchedFractionFactor, unmatchedFractionRCFactor */
    /* Several replicas (one handle per GPU, the round's targets sharded over them): the finalize covers the targets of
     * ALL replicas, so it may only run when EVERY replica's pass 1 confirms its part of the prediction. gate_dev: the
     * device word (1 = confirmed) the check writes and the gated launches read — caller's memory, valid until they have
     * run (NULL = the handle's own). exchange(ctx, 0, gate_dev, stream) is called once, after the check has been queued
     * on `stream` (a hipStream_t) and before any gated launch: it queues, on that stream, the reduction of the word to
     * its minimum over the replicas in place (ncclAllReduce / its torch.distributed form) and whatever else the gated
     * launches must wait for (the all-gather that delivers ext_dev); non-zero return = error. exchange(ctx, 1, …) is
     * called later, if the finalize could be queued, and returns the reduced word (it may block until it is known).
     * veto: this replica cannot take part (its share of ext_dev is not what was predicted): its word is 0. */
    uint32_t *gate_dev;
    int (*exchange)(void *ctx, int phase, void *gate_dev, void *stream);
    void *exchange_ctx;
    int veto;
} swsem_spec_finalize_t;
int swsem_emit_batch_begin_spec(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                                const int *unmatchedFractionFactor, const int64_t *processedTargetsCount, const int64_t *targetIdx,
                                const uint64_t *refExtLoadedPos, uint64_t nLoaded, const swsem_spec_finalize_t *spec, int *applied);
int swsem_emit_batch_end(swsem_t *h);
/* Two emissions can be in flight: _begin only waits for the one before the previous (whose buffers it takes
 * over). swsem_emit_result / swsem_emit_pack_dev read the latest emission, or — after swsem_emit_select(h, 1) —
 * the one before it, which lets a caller begin round r+1 before it collects the streams of round r.
 * swsem_emit_batch_end waits for both. The pointers swsem_emit_result hands out (page-locked host memory of the library) stay
 * valid until the emission FOUR _begin calls later is taken: a caller may copy them out on a thread of its own while it
 * begins, runs and takes the next emissions (every slot keeps two host buffers and uses them in turn). */
int swsem_emit_select(swsem_t *h, int previous);
int swsem_emit_result(swsem_t *h, int k, swsem_streams_t *out);
/* Keep the streams in HBM (no host copy inside swsem_emit_batch; swsem_emit_result then copies on demand)
 * and pack them, (result, stream) major, into one device buffer — the unit the multi-GPU path gathers to
 * the rank that feeds the host-side PPMd/LZMA backend. sizes[n*6] and *total are host outputs; a NULL
 * dst_dev only reports sizes. The copy has completed when the call returns (any stream may consume it). */
void swsem_emit_set_host_copy(swsem_t *h, int on);
int swsem_emit_unmatched(swsem_t *h, uint64_t *unmatched /* [n of the last swsem_emit_batch] */);
int swsem_emit_pack_dev(swsem_t *h, uint8_t *dst_dev, uint64_t cap, uint64_t *sizes, uint64_t *total);
/* The same copy queued on `stream` (a hipStream_t; NULL = the handle's main stream) instead of waited for: the caller's
 * consumer must be ordered behind it on that stream. For callers that must not block while the device is busy with the
 * round's finalize (the copy would wait for compute units, and the host with it). */
int swsem_emit_pack_dev_on(swsem_t *h, uint8_t *dst_dev, uint64_t cap, void *stream);
/* the counters processMatches adds up (MBGC_Encoder.cpp:293-306), per contig of the selected emission, without its bytes:
 * unmatchedChars, extensionsMatchedChars, extensionsMismatches, totalMatched, removedGapBreakingMatches, matches. Waits
 * for the emission like swsem_emit_result. */
int swsem_emit_counters(swsem_t *h, uint64_t *out /* [n * 6] */);

/* ---- the decoder's per-contig automaton on the device: the exact inverse of processMatches,
 * MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars + extendMatchLeft/Right (mbgccoder/MBGC_Decoder.cpp:319-523), one
 * wave per contig against the handle's reference buffer (the contigs of a call are independent: each stands against the
 * reference as its lock position froze it). All pointers are device pointers. destLen[k] = bytes written, unmatched[k] =
 * the function's return value or -1 for a malformed stream set (a stream ran out, bytes were left over, dest too small). */
typedef struct {
    const uint8_t *stream_dev[SWSEM_NSTREAMS];
    uint64_t size[SWSEM_NSTREAMS];
    uint64_t refLockPos;                  /* the target's matching-lock position, SWSEM_NO_LOCK if none */
    uint8_t *dest_dev;
    uint64_t destCap;
} swsem_decode_job_t;
int swsem_decode_contigs_dev(swsem_t *h, const swsem_emit_params_t *p, int n, const swsem_decode_job_t *jobs,
                             uint64_t *destLen, int64_t *unmatched);
/* Device-side check of the selected emission (see swsem_emit_select): every emitted contig's six streams, still packed in
 * HBM, are decoded by that automaton — no encoder logic involved — and compared with the query bytes they were emitted
 * for and with processMatches' return value. *nbad = contigs that fail; *firstBad / *firstDiff (may be NULL) = the first
 * of them and the first differing byte. Valid while the reference text the emission could match is still in the buffer,
 * i.e. before the rounds after it start overwriting it (a circular buffer that has wrapped). Waits for the emission. */
int swsem_emit_verify(swsem_t *h, int *nbad, int *firstBad, uint64_t *firstDiff);

/* ---- device-memory plumbing for host code that stays free of HIP headers (the C++ facade) */
int swsem_dev_malloc(swsem_t *h, uint64_t bytes, void **out_dev);
int swsem_dev_free(swsem_t *h, void *p_dev);
int swsem_dev_upload(swsem_t *h, void *dst_dev, const void *src_host, uint64_t bytes);   /* synchronous */
int swsem_dev_download(swsem_t *h, void *dst_host, const void *src_dev, uint64_t bytes); /* synchronous */
int swsem_dev_copy(swsem_t *h, void *dst_dev, const void *src_dev, uint64_t bytes);      /* on the handle's stream */

/* ---- test / measurement hooks (not part of the reference surface) */
int swsem_debug_copy_ref(swsem_t *h, uint64_t from, uint64_t n, uint8_t *out);      /* getRef() bytes, .h:104 */
/* overwrites n reference bytes from `from` (tests: what an emission must not depend on is filled with garbage) */
int swsem_debug_write_ref(swsem_t *h, uint64_t from, uint64_t n, const uint8_t *in);
int swsem_debug_copy_ht(swsem_t *h, uint32_t *out /* [hash_size] 32-bit image as on the CPU */);
enum { SWSEM_K_LOAD = 0, SWSEM_K_INSERT = 1, SWSEM_K_PROBE = 2 /* unused: the chains hash their own scan windows */, SWSEM_K_EMIT2 = 3, SWSEM_K_RESOLVE = 4,
       SWSEM_K_STITCH = 5, SWSEM_K_EMIT = 6, SWSEM_K_COUNT = 7 };
/* When enabled every kernel family's launches are bracketed by HIP events on the stream they run on; the
 * accumulated device time (ms) and bracket count per family are read back with swsem_profile_get.
 * EMIT = pass 1 of processMatches (main stream), EMIT2 = its second phase (second stream; the bracket also
 * covers whatever the main stream runs beside it, so it is an elapsed time, not a sum of kernel times). */
/* counters of the emission's pairing chain since swsem_create: out[4] = steps taken with an inherited region boundary that
   was not the match's own, out[5] = speculative blocks that were not accepted, out[6] = groups of 64 matches replayed from
   the true state, out[7] = blocks given up because they carried too many such boundaries (diagnostics; synchronises the device) */
int swsem_debug_emit_stats(swsem_t *h, uint64_t out[8]);
int swsem_debug_block_times(swsem_t *h, uint64_t *out /* [cap][3]: ticks, candidates visited, stack rows */, uint64_t cap, uint64_t *nblocks);
int swsem_profile_enable(swsem_t *h, int on);
int swsem_profile_get(swsem_t *h, double ms[SWSEM_K_COUNT], uint64_t launches[SWSEM_K_COUNT]);
/* counters of the last batch (after swsem_batch_counts): [0] query bases, [1] hash-table probes,
 * [2] verified hits, [3] matches, [4] sum of match lengths (after swsem_batch_fingerprint),
 * [5] resolve blocks whose speculation was rejected and that were replayed from the true state */
int swsem_batch_stats(swsem_t *h, uint64_t stats[6]);

#ifdef __cplusplus
}
#endif
#endif
