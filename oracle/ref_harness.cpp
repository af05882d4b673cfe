// TEST INFRASTRUCTURE ONLY (oracle/_ref). A thin extern "C" harness around the *reference's own*
// classes, compiled against the headers where they lie under /root/reference and linked with the
// reference's own object files (see oracle/Makefile). Nothing of the reference is copied: this file
// only calls it. It exists so that the repo's C restatement (oracle/*.c) and the HIP path can be
// compared with the real implementation on arbitrary schedules (lock positions, wraps, RC loads),
// which the `mbgc` CLI cannot be driven to reproduce deterministically.
//
// Reference entry points driven here:
//   SlidingWindowExpSparseEMMatcher            matching/SlidingWindowSparseEMMatcher.h:129-138
//   loadRef / loadSeparator / matchTexts       matching/SlidingWindowSparseEMMatcher.cpp:439-492
//   acquire/releaseWorkerMatchingLockPos       matching/SlidingWindowSparseEMMatcher.cpp:361-400
//   MBGC_Encoder::processMatches (+extend*)    mbgccoder/MBGC_Encoder.cpp:143-427
//
// The encoder's emission routines are private members; this translation unit (and only this one)
// sees the class with its access specifiers opened so a test can call them directly. Layout is
// unaffected, the reference objects themselves are compiled untouched.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>
#include <numeric>
#include <map>
#include <set>
#include <list>
#include <functional>
#include <memory>
#include <cmath>
#include <chrono>
#include <omp.h>

#define private public
#define protected public
#include "mbgccoder/MBGC_Encoder.h"
#include "matching/SimpleSequenceMatcher.h"
#include "matching/copmem/CopMEMMatcher.h"
#include "coders/LzmaCoder.h"
#include "coders/PpmdCoder.h"
#include "coders/PropsLibrary.h"
#undef private
#undef protected

namespace {
struct NullBuf : std::streambuf { int overflow(int c) override { return c; } };
NullBuf g_nullbuf;
std::ostream g_null(&g_nullbuf);
void quiet() {
    PgHelpers::devout = &g_null;
    PgHelpers::appout = &g_null;
    PgHelpers::numberOfThreads = 1;
    omp_set_num_threads(1);      // single-thread semantics are the oracle
}
}

extern "C" {

// ---------------------------------------------------------------- matcher
void *refm_create(uint64_t maxRefLen, int L, int k1, int k2, int skipMargin) {
    quiet();
    if (__builtin_ctz((uint32_t) k1) == 0)                                   // MGMP.cpp:170-176: an odd k1 takes the base class
        return new SlidingWindowSparseEMMatcher(maxRefLen, L, k1, k2, skipMargin);
    return new SlidingWindowExpSparseEMMatcher(maxRefLen, L, k1, k2, skipMargin);
}
void refm_destroy(void *h) { delete (SlidingWindowSparseEMMatcher *) h; }
void refm_disable_sliding_window(void *h) { ((SlidingWindowSparseEMMatcher *) h)->disableSlidingWindow(); }
void refm_set_sliding_window_size(void *h, int f) { ((SlidingWindowSparseEMMatcher *) h)->setSlidingWindowSize(f); }
void refm_disable_circular_buffer(void *h) { ((SlidingWindowSparseEMMatcher *) h)->disableCircularBuffer(); }
void refm_load_ref(void *h, const char *t, uint64_t n, int rc, int addSep, int sep) {
    ((SlidingWindowSparseEMMatcher *) h)->loadRef(t, n, rc != 0, addSep != 0, (char) sep);
}
void refm_load_separator(void *h, int sep) { ((SlidingWindowSparseEMMatcher *) h)->loadSeparator((char) sep); }
uint64_t refm_ref_length(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->getRefLength(); }
uint64_t refm_loading_position(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->getLoadingPosition(); }
uint64_t refm_loaded_ref_length(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->getLoadedRefLength(); }
uint64_t refm_max_ref_length(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->getMaxRefLength(); }
uint64_t refm_acquire_lock(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->acquireWorkerMatchingLockPos(); }
void refm_release_lock(void *h, uint64_t v) { ((SlidingWindowSparseEMMatcher *) h)->releaseWorkerMatchingLockPos(v); }
void refm_set_position(void *h, uint64_t pos, int laps) { ((SlidingWindowSparseEMMatcher *) h)->setPosition(pos, laps); }
uint32_t refm_hash_size(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->hash_size; }
const uint32_t *refm_ht(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->ht32bit; }
const char *refm_ref(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->getRef(); }
int refm_K(void *h) { return ((SlidingWindowSparseEMMatcher *) h)->K; }

// matches are returned as rows of 3 x u64 {posSrcText, length, posDestText}; returns the count
// (which may exceed cap: then only cap rows were written).
uint64_t refm_match(void *h, const char *q, uint64_t n, uint32_t minLen, uint64_t lockPos,
                    uint64_t *out, uint64_t cap) {
    std::vector<TextMatch> res;
    ((SlidingWindowSparseEMMatcher *) h)->matchTexts(res, q, n, false, false, minLen, lockPos);
    for (size_t i = 0; i < res.size() && i < cap; i++) {
        out[3 * i] = res[i].posSrcText; out[3 * i + 1] = res[i].length; out[3 * i + 2] = res[i].posDestText;
    }
    return res.size();
}

// ---------------------------------------------------------------- encoder emission
struct RefEnc {
    MBGC_Params params;
    MBGC_Encoder *enc;
};

// mode: the `-m` preset (0..3). lazy: lazyDecompressionSupport. bit40: enable40bitReference.
void *refe_create(void *matcher, int mode, int lazy, int bit40, int nTargets) {
    quiet();
    RefEnc *r = new RefEnc();
    r->params.setCompressionMode(mode);
    r->params.lazyDecompressionSupport = lazy != 0;
    r->params.enable40bitReference = bit40 != 0;
    r->params.initMismatchesMatchingScoreParams();
    r->enc = new MBGC_Encoder(&r->params);
    MBGC_Encoder *e = r->enc;
    e->matcher = (SlidingWindowSparseEMMatcher *) matcher;
    e->targetsCount = nTargets;
    e->targetsCountShift = 0;
    e->processedTargetsCount = 0;
    e->unmatchedFractionFactors.assign(2 * nTargets, 0);
    for (int i = 0; i < nTargets; i++) {
        e->unmatchedFractionFactors[2 * i] = r->params.currentUnmatchedFractionFactor < 256 ? r->params.currentUnmatchedFractionFactor : 0;
        e->unmatchedFractionFactors[2 * i + 1] = r->params.unmatchedFractionRCFactor;
    }
    e->targetRefExtensions.resize(nTargets);
    e->targetLiterals.resize(nTargets);
    e->targetMapOffDests.resize(nTargets);
    e->targetMapOff5thByte.resize(nTargets);
    e->targetMapLenDests.resize(nTargets);
    e->targetGapDeltas.resize(nTargets);
    e->targetGapMismatchesFlags.resize(nTargets);
    if (lazy)
        e->refExtLoadedPosArr.emplace_back(e->matcher->getLoadingPosition());   // MBGC_Encoder.cpp:789-791
    return r;
}
void refe_destroy(void *h) { RefEnc *r = (RefEnc *) h; delete r->enc; delete r; }
void refe_set_processed_targets(void *h, int64_t n) { ((RefEnc *) h)->enc->processedTargetsCount = n; }
// what finalizeParallelProcessingOfTarget does to the lazy bookkeeping (MBGC_Encoder.cpp:557-562)
void refe_push_loaded_pos(void *h, uint64_t v) { ((RefEnc *) h)->enc->refExtLoadedPosArr.emplace_back(v); }

// Runs processMatches on rows of 3 x u64; returns unmatchedChars (SIZE_MAX = skipped as dissimilar).
// Streams of target `t` accumulate inside the encoder; fetch them with refe_stream.
uint64_t refe_process_matches(void *h, const uint64_t *m, uint64_t n, char *dest, uint64_t destLen,
                              int t, uint64_t lockPos) {
    RefEnc *r = (RefEnc *) h;
    std::vector<TextMatch> v;
    v.reserve(n);
    for (uint64_t i = 0; i < n; i++) v.emplace_back(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
    return r->enc->processMatches(v, dest, destLen, t, lockPos);
}
void refe_after_sequence(void *h, int t) { ((RefEnc *) h)->enc->processAfterSequence(t); }
void refe_after_target(void *h, int t) { ((RefEnc *) h)->enc->processAfterTarget(t); }
// target t as a worker that has not started finds it: what processMatches had appended for it is dropped (the round drive of
// tests/_driver.py voids the first pass over a round's later targets when a contig was given up as dissimilar)
void refe_reset_target(void *h, int t) {
    MBGC_Encoder *e = ((RefEnc *) h)->enc;
    e->targetRefExtensions[t].clear();
    e->targetLiterals[t].clear();
    e->targetMapOffDests[t].str(""); e->targetMapOffDests[t].clear();
    e->targetMapOff5thByte[t].clear();
    e->targetMapLenDests[t].str(""); e->targetMapLenDests[t].clear();
    e->targetGapDeltas[t].clear();
    e->targetGapMismatchesFlags[t].clear();
}

// which: 0 literals, 1 mapOff, 2 mapOff5thByte, 3 mapLen, 4 gapDeltas, 5 gapMismatchesFlags.
// Copies up to cap bytes, returns the full size.
uint64_t refe_stream(void *h, int t, int which, char *out, uint64_t cap) {
    MBGC_Encoder *e = ((RefEnc *) h)->enc;
    std::string s;
    switch (which) {
        case 0: s = e->targetLiterals[t]; break;
        case 1: s = e->targetMapOffDests[t].str(); break;
        case 2: s = e->targetMapOff5thByte[t]; break;
        case 3: s = e->targetMapLenDests[t].str(); break;
        case 4: s = e->targetGapDeltas[t]; break;
        case 5: s = e->targetGapMismatchesFlags[t]; break;
        default: return 0;
    }
    memcpy(out, s.data(), std::min<uint64_t>(cap, s.size()));
    return s.size();
}

// ---------------------------------------------------------------- -m3 reverse-complement pass over the literal stream
// SimpleSequenceMatcher::rcMatchSequence (matching/SimpleSequenceMatcher.cpp:165-176) as MBGC_Encoder.cpp:637-638 calls it,
// with one thread (the multi-threaded index build fills its buckets in arrival order). seq is rewritten in place; the maps
// are copied out (up to cap bytes each, full sizes returned through mapOffLen / mapLenLen). Returns the new length.
uint64_t refrc_match_sequence(char *seq, uint64_t n, uint64_t targetMatchLength, uint32_t minMatchLength,
                              char *mapOff, uint64_t *mapOffLen, char *mapLen, uint64_t *mapLenLen, uint64_t cap) {
    quiet();
    std::string s(seq, n), off, len;
    PgTools::SimpleSequenceMatcher::rcMatchSequence(s, off, len, targetMatchLength, minMatchLength);
    memcpy(seq, s.data(), s.size());
    memcpy(mapOff, off.data(), std::min<uint64_t>(cap, off.size()));
    memcpy(mapLen, len.data(), std::min<uint64_t>(cap, len.size()));
    *mapOffLen = off.size(); *mapLenLen = len.size();
    return s.size();
}
// the matches CopMEMMatcher::matchTexts pushes for (reverseComplement(seq), destIsSrc, revComplMatching), in push order
uint64_t refrc_find_matches(const char *seq, uint64_t n, uint32_t targetMatchLength, uint32_t minMatchLength, uint64_t *out, uint64_t cap) {
    quiet();
    std::string s(seq, n);
    CopMEMMatcher m(s.data(), s.size(), targetMatchLength, minMatchLength);
    std::string q = PgHelpers::reverseComplement(s);
    std::vector<TextMatch> res;
    m.matchTexts(res, q, true, true, minMatchLength == UINT32_MAX ? targetMatchLength : minMatchLength);
    for (size_t i = 0; i < res.size() && i < cap; i++) { out[3 * i] = res[i].posSrcText; out[3 * i + 1] = res[i].length; out[3 * i + 2] = res[i].posDestText; }
    return res.size();
}

// ---------------------------------------------------------------- backend: leaf coders and the collective section
// One leaf coder call as Compress() makes it (coders/CodersLib.cpp:53-66): LzmaCompress / Ppmd7Compress with the given
// properties. Returns 0 and the coder's bytes, or non-zero.
int refbk_leaf(int coder, int level, uint32_t dictSize, int lc, int lp, int pb, int fb, int algo, int numThreads, uint32_t memSize,
               int order, const unsigned char *src, uint64_t n, unsigned char *dest, uint64_t cap, uint64_t *destLen) {
    quiet();
    std::unique_ptr<CoderProps> props;
    if (coder == LZMA_CODER) props.reset(new LzmaCoderProps(level, dictSize, lc, lp, pb, fb, algo, numThreads));
    else if (coder == PPMD7_CODER) props.reset(new PpmdCoderProps(memSize, order));
    else return -1;
    size_t len = 0;
    unsigned char *out = Compress(len, src, n, props.get(), 1, &null_stream);
    if (len > cap) { delete[] out; return -2; }
    memcpy(dest, out, len);
    delete[] out;
    *destLen = len;
    return 0;
}

// MBGC_Encoder::prepareAndCompressStreams (mbgccoder/MBGC_Encoder.cpp:615-732) on streams given from outside: writes the
// archive (MBGC_Params::write, writeStats, then the collective section) to `path`. *prefix = bytes in front of the
// collective section. headers / headerTemplates go in as the one file's (prepareHeadersStreams appends the file separator
// to the templates). flags: bit0 ultraStreamsCompression, bit1 lazyDecompressionSupport, bit2 sequentialMatching.
int refbk_archive(const char *path, int mode, int flags, int threads, uint64_t refFinalTotalLength, const unsigned char *const *data,
                  const uint64_t *size, uint64_t *prefix) {
    quiet();
    MBGC_Params params;
    params.setCompressionMode(mode);
    if (flags & 1) params.setUltraStreamsCompression();
    params.lazyDecompressionSupport = (flags & 2) != 0;
    params.sequentialMatching = (flags & 4) != 0;
    params.backendThreads = threads;
    params.outArchiveFileName = path;
    params.forceOverwrite = true;
    MBGC_Encoder e(&params);
    auto str = [&](int i) { return std::string((const char *) data[i], size[i]); };
    e.filesCount = 1;
    e.namesStr = str(0);
    e.seqsCountsDest << str(1);
    e.fileHeadersTemplates.assign(1, str(2));
    e.fileHeaders.assign(1, str(3));
    e.dnaLineLengthsDest << str(4);
    e.unmatchedFractionFactors.assign(data[5], data[5] + size[5]);
    e.targetLiterals.assign(1, str(6));
    e.locksPosStream = str(9);
    e.targetGapDeltas.assign(1, str(10));
    e.targetGapMismatchesFlags.assign(1, str(11));
    e.mapOffStream = str(12);
    e.targetMapOff5thByte.assign(1, str(13));
    e.mapLenStream = str(14);
    e.refExtSizeDest << str(15);
    e.refFinalTotalLength = refFinalTotalLength;
    std::ostringstream head;
    params.write(head);
    e.writeStats(head);
    *prefix = head.str().size();
    e.prepareAndCompressStreams();
    return 0;
}

// the leaf callback of include/mbgc_backend.h served by the reference's coders: what a maintainer's build passes to
// mbgc_backend_compress_streams, (the tool itself is given oracle/_ref/libmbgc_coders.so: the coders alone, ref_coders.cpp)
struct RefLeafCoder { int coder, level, lc, lp, pb, fb, algo, numThreads; uint32_t dictSize, memSize; int order; };
int mbgc_leaf_compress(void *ctx, const RefLeafCoder *c, const unsigned char *src, uint64_t n, unsigned char *dest, uint64_t cap, uint64_t *destLen) {
    return refbk_leaf(c->coder, c->level, c->dictSize, c->lc, c->lp, c->pb, c->fb, c->algo, c->numThreads, c->memSize, c->order, src, n, dest, cap, destLen);
}

// readCompressedCollectiveParallel (coders/CodersLib.cpp:417-478) over a collective section in memory: the reference's
// own reader. sizes[i] = length of stream i; out receives the streams back to back (capacity cap). Returns total bytes.
uint64_t refbk_read_collective(const unsigned char *section, uint64_t n, int nStreams, uint64_t *sizes, unsigned char *out, uint64_t cap) {
    quiet();
    std::istringstream in(std::string((const char *) section, n));
    std::vector<std::string> strs(nStreams);
    std::vector<std::string *> ptrs;
    for (auto &x : strs) ptrs.push_back(&x);
    readCompressedCollectiveParallel(in, ptrs);
    uint64_t at = 0;
    for (int i = 0; i < nStreams; i++) {
        sizes[i] = strs[i].size();
        if (at + strs[i].size() <= cap) memcpy(out + at, strs[i].data(), strs[i].size());
        at += strs[i].size();
    }
    return at;
}

}  // extern "C"
