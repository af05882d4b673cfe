/* mbgc_backend.h — C ABI of the backend's job table and container framing.
 *
 * What stands between the streams the match-finding path produces and the archive bytes, in the reference:
 *
 *   MBGC_Encoder::prepareAndCompressStreams            mbgccoder/MBGC_Encoder.cpp:641-710   which stream goes to which coder, with
 *                                                      which order / memory / lc-lp-pb / word size, in how many parallel blocks,
 *                                                      per compression mode (-m0..3) — the "job table"
 *   getDefaultCoderProps / getCompoundCoderProps       coders/PropsLibrary.cpp:8-59         coder level -> parameters
 *   ParallelBlocksCoderProps::prepare, parallelBlocksCompress   coders/CodersLib.h:141-173, CodersLib.cpp:292-314
 *   Compress (compound coder), writeHeader, CompressionJob::writeCompressedCollectiveParallel   CodersLib.cpp:53-132, 202-218, 372-415
 *
 * The entropy coders themselves (PPMd7 and LZMA of the 7-zip SDK, coders/PpmdCoder.cpp, coders/LzmaCoder.cpp) stay what they
 * are — north_star: "the unchanged host-side PPMd/LZMA backend" — and enter here as ONE callback: "compress this buffer with
 * this leaf coder". With the reference's own Ppmd7Compress / LzmaCompress behind the callback the bytes this call returns are
 * the bytes writeCompressedCollectiveParallel writes (tests/test_backend_jobs.py, against the reference compiled into
 * oracle/_ref). Jobs and blocks run on host threads, so the backend of one collection can run beside the matching of the next. */
#ifndef MBGC_BACKEND_H
#define MBGC_BACKEND_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MBGC_NO_CODER = 0, MBGC_LZMA_CODER = 1, MBGC_PPMD7_CODER = 3, MBGC_COMPOUND_CODER = 77, MBGC_PARALLEL_BLOCKS_CODER = 88 };   /* CodersLib.h:14-20 */

/* a leaf coder with its parameters: LzmaCoderProps (coders/LzmaCoder.h:79-104) or PpmdCoderProps (coders/PpmdCoder.h:6-19) */
typedef struct {
    int coder;                                           /* MBGC_LZMA_CODER or MBGC_PPMD7_CODER */
    int level, lc, lp, pb, fb, algo, numThreads;         /* LZMA */
    uint32_t dictSize;
    uint32_t memSize; int order;                         /* PPMd7 */
} mbgc_leaf_coder_t;

/* Compress src with the leaf coder into dest (capacity destCap = srcLen + srcLen / 3 + 256, the reference's own bound for
 * estimated_compression = 1); *destLen = bytes written, in the coder's self-contained format (LzmaCompress: 5 props bytes +
 * stream; Ppmd7Compress: its header + stream). Return 0, or non-zero when the coder fails — the whole call then fails as
 * the reference exits. May be called from several threads at once. */
typedef int (*mbgc_leaf_compress_fn)(void *ctx, const mbgc_leaf_coder_t *coder, const uint8_t *src, uint64_t srcLen,
                                     uint8_t *dest, uint64_t destCap, uint64_t *destLen);

/* the streams in the order the reference enrols them (MBGC_Encoder.cpp:648-709) */
enum {
    MBGC_ST_NAMES = 0, MBGC_ST_SEQ_COUNTS, MBGC_ST_HEADER_TEMPLATES, MBGC_ST_HEADERS, MBGC_ST_DNA_LINE_LENGTHS,
    MBGC_ST_UNMATCHED_FRACTION_FACTORS, MBGC_ST_LITERALS, MBGC_ST_RC_MAP_OFF, MBGC_ST_RC_MAP_LEN, MBGC_ST_LOCKS_POS,
    MBGC_ST_GAP_DELTAS, MBGC_ST_GAP_MISMATCHES_FLAGS, MBGC_ST_MAP_OFF, MBGC_ST_MAP_OFF_5TH_BYTE, MBGC_ST_MAP_LEN,
    MBGC_ST_REF_EXT_SIZE, MBGC_ST_COUNT
};

typedef struct {
    int coderMode;                                       /* 0 speed, 1 default, 2 repo, 3 max (MBGC_Params.h:24-27) */
    int ultraStreamsCompression;                         /* MBGC_Params.h:76 */
    int k;                                               /* 16 = the proteins profile (MGMP_Params.h:27) */
    int enableExtensionsWithMismatches, mismatchesWithExclusion, sequentialMatching, rcRedundancyRemoval,
        frugal64bitLenEncoding, lazyDecompressionSupport;
    uint64_t refFinalTotalLength;                        /* > UINT32_MAX: the 5th-byte stream is enrolled (:698-699) */
    int numberOfThreads;                                 /* PgHelpers::numberOfThreads: > 1 gives LZMA two threads (PropsLibrary.cpp:9) */
    /* 0 or 1: the reference's block counts (the section then equals the reference's byte for byte). k > 1: every stream that
     * the table splits is split into k times as many blocks (still at least 2^20 bytes each) — the container carries the
     * count, so the reference's reader takes the section as it is; the coders restart their models k times as often
     * (+0.1 % of section on 1000 genomes at k = 8) and the two 200 MB blocks of the flags stream no longer set the time */
    int blocksScale;
} mbgc_backend_params_t;

const char *mbgc_backend_last_error(void);
/* the coder of stream `st` under these parameters: *blocks = 0 when the stream is not split (no ParallelBlocksCoderProps
 * around it), else the requested block count; primary->coder = 0 unless the coder is the compound one (then: secondary over
 * primary). Returns 0, or -1 when the stream is not enrolled under these parameters. */
int mbgc_backend_job(const mbgc_backend_params_t *p, int st, int *blocks, mbgc_leaf_coder_t *coder, mbgc_leaf_coder_t *primary);
/* CompressionJob::writeCompressedCollectiveParallel over the enrolled streams: *out = malloc'd archive bytes of the
 * collective section (free with mbgc_backend_free). threads <= 0: one per job. */
int mbgc_backend_compress_streams(const mbgc_backend_params_t *p, const uint8_t *const data[MBGC_ST_COUNT], const uint64_t size[MBGC_ST_COUNT],
                                  mbgc_leaf_compress_fn leaf, void *ctx, int threads, uint8_t **out, uint64_t *outLen);
void mbgc_backend_free(uint8_t *p);

/* The incremental form: the same section while the streams are still growing — the backend beside the matching (SURVEY.md
 * §8(f) row 3's reason to exist). Streams are fed in any number of pieces and in any interleaving; a stream that the job
 * table splits (ParallelBlocksCoderProps) gives up a block of `blockBytes` to the pool of `threads` coder threads as soon
 * as more than a block has arrived, every other stream is coded at the end. The container is parallelBlocksCompress's
 * (CodersLib.cpp:292-314) with the blocks as they were cut — other block boundaries than the reference chooses, so other
 * bytes, and the same streams out of the reference's reader, which takes every block's length from the block's own header
 * (parallelBlocksDecompress, CodersLib.cpp:316-345). finish() codes what is left, writes the section (malloc'd, free with
 * mbgc_backend_free) and tells how many blocks were coded before it was called; it takes the one parameter that is only known
 * when the matching is over (the final reference length: the 5th-byte stream is enrolled beyond 2^32, MBGC_Encoder.cpp:698-699;
 * bytes fed for a stream that turns out not to be enrolled are dropped, as the one-shot call ignores them). close() frees the
 * object (also without finish). */
typedef struct mbgc_backend_stream mbgc_backend_stream_t;
mbgc_backend_stream_t *mbgc_backend_stream_open(const mbgc_backend_params_t *p, mbgc_leaf_compress_fn leaf, void *ctx, int threads, uint64_t blockBytes);
int mbgc_backend_stream_feed(mbgc_backend_stream_t *s, int st, const uint8_t *data, uint64_t n);
int mbgc_backend_stream_finish(mbgc_backend_stream_t *s, uint64_t refFinalTotalLength, uint8_t **out, uint64_t *outLen, uint64_t *blocksCodedEarly);
void mbgc_backend_stream_close(mbgc_backend_stream_t *s);

#ifdef __cplusplus
}
#endif
#endif
