// C++ facade with the reference's class and method names over the C ABI (include/mbgc_swsem.h).
//
// Mirrors matching/SlidingWindowSparseEMMatcher.h:88-124 (the Exp variant that
// MultipleGenomeMatchingProcessor::initMatcher constructs, MGMP.cpp:170-172) and PgTools::TextMatch
// (matching/TextMatchers.h:9-81). Errors the reference reports with a message on stderr followed by
// exit(EXIT_FAILURE) do exactly that here: the C ABI only returns codes, the exit happens in this layer.
// getRef() has no equivalent: the reference bytes live in HBM, and their only consumer outside the
// matcher — MBGC_Encoder::processMatches / extendMatchLeft / extendMatchRight — runs on the device
// (processMatches below, forwarding to swsem_emit).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mbgc_swsem.h"

#ifndef MBGC_HIP_USE_REFERENCE_TEXTMATCH   // (a build inside the reference's tree brings its own matching/TextMatchers.h: oracle/dropin)
namespace PgTools {

struct TextMatch {                      // TextMatchers.h:9-16
    uint64_t posSrcText, length, posDestText, nextSrcRegionLoadingPos;
    TextMatch(uint64_t s = 0, uint64_t l = 0, uint64_t d = 0) : posSrcText(s), length(l), posDestText(d), nextSrcRegionLoadingPos(0) {}
    bool operator==(const TextMatch &r) const { return posSrcText == r.posSrcText && length == r.length && posDestText == r.posDestText; }
    bool pairedWith(const TextMatch &r) const { return posSrcText + r.posDestText == r.posSrcText + posDestText; }     // :42-44
    uint64_t endPosSrcText() const { return posSrcText + length; }
    uint64_t endPosDestText() const { return posDestText + length; }
};

}  // namespace PgTools
#endif

struct EmittedStreams {                 // the six byte streams of one processMatches call + its counters
    std::string s[SWSEM_NSTREAMS];
    uint64_t unmatchedChars = 0, extensionsMatchedChars = 0, extensionsMismatches = 0, totalMatched = 0,
             removedGapBreakingMatches = 0, nmatches = 0;
};

class SlidingWindowSparseEMMatcher {
public:
    const size_t REF_SHIFT = 1;                                             // .h:14
    static const size_t SW_END_ERASED_FLAG = SIZE_MAX;                      // .h:46

    // SlidingWindowExpSparseEMMatcher(refLengthLimit, targetMatchLength, k1, k2, skipMargin), .cpp:494-519
    SlidingWindowSparseEMMatcher(size_t refLengthLimit, uint32_t targetMatchLength, int k1, int k2, int skipMargin, int device = 0);
    virtual ~SlidingWindowSparseEMMatcher();

    void disableSlidingWindow();                                            // .h:93
    void disableCircularBuffer();                                           // .h:95
    void setSlidingWindowSize(uint8_t factor);                              // .h:97
    void loadRef(const char *refText, size_t refLength, bool loadRCRef, bool addRegionSeparators, char regionSeparator);   // .h:99
    void loadSeparator(char regionSeparator);                               // .h:102
    size_t getMaxRefLength() const;                                         // .h:105
    size_t getSlidingWindowSize() const;                                    // swSize, .h:47 (0: no window)
    size_t getDroppedBytes() const;                                         // extension bytes given up at the window's end so far (.cpp:433)
    size_t getRefLength() const;                                            // .h:106
    size_t getLoadingPosition() const;                                      // .h:107
    size_t getLoadedRefLength() const;                                      // .h:108
    void setPosition(size_t refPos, int reachedRefLengthCount);             // .h:110
    size_t acquireWorkerMatchingLockPos();                                  // .h:115
    void releaseWorkerMatchingLockPos(size_t lockValue);                    // .h:116
    void matchTexts(std::vector<PgTools::TextMatch> &resMatches, const std::string &destText, bool destIsRef, bool revComplMatching,
                    uint32_t minMatchLength, size_t matchingLockPos = SW_END_ERASED_FLAG);                                  // .h:120
    void matchTexts(std::vector<PgTools::TextMatch> &resMatches, const char *destText, size_t destLen, bool destIsRef,
                    bool revComplMatching, uint32_t minMatchLength, size_t matchingLockPos = SW_END_ERASED_FLAG);           // .h:123

    // ---- device-side continuation of the hot path (no counterpart in the class above because the
    // reference does this on the CPU with getRef())
    // MBGC_Encoder::processMatches for the contig matched by the last matchTexts call. Returns
    // unmatchedChars or SIZE_MAX (PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY).
    size_t processMatches(const swsem_emit_params_t &p, size_t matchingLockPos, int unmatchedFractionFactor,
                          int64_t processedTargetsCount, int64_t targetIdx, const std::vector<size_t> &refExtLoadedPosArr,
                          EmittedStreams &out);
    // a whole round (device-resident contigs): matchTexts + processMatches for n contigs at once
    void matchRound(const uint8_t *contigsDev, const std::vector<uint64_t> &offsets, uint32_t minMatchLength,
                    const std::vector<uint64_t> &lockPos, std::vector<uint64_t> &matchCounts);
    void emitRound(const swsem_emit_params_t &p, const std::vector<uint64_t> &lockPos, const std::vector<int> &factors,
                   const std::vector<int64_t> &processed, const std::vector<int64_t> &targetIdx,
                   const std::vector<size_t> &refExtLoadedPosArr, std::vector<EmittedStreams> &out);
    void matchRoundBegin(const uint8_t *contigsDev, const std::vector<uint64_t> &offsets, uint32_t minMatchLength,
                         const std::vector<uint64_t> &lockPos);          // matchRound without the wait for the counts
    // the same in the overlapped form the round pipeline uses (include/mbgc_swsem.h: swsem_emit_batch_begin[_spec], two
    // emissions in flight, swsem_finalize_targets): emitRoundBegin returns with processMatches' return values known and
    // the stream bytes still being produced; spec != nullptr carries the predicted finalize (true: it has been applied)
    bool emitRoundBegin(const swsem_emit_params_t &p, const std::vector<uint64_t> &lockPos, const std::vector<int> &factors,
                        const std::vector<int64_t> &processed, const std::vector<int64_t> &targetIdx,
                        const std::vector<size_t> &refExtLoadedPosArr, const swsem_spec_finalize_t *spec,
                        std::vector<uint64_t> &unmatched, std::vector<uint64_t> &matchCounts);
    void setEmitHostCopy(bool on);                                          // off: the streams stay in HBM until emitTake asks for them
    void emitSelect(bool previous);
    void emitTake(int k, EmittedStreams &out);                              // streams of contig k of the selected emission (waits for it)
    void emitView(int k, swsem_streams_t &view);                            // the same without a copy: pointers into the handle's host buffer, valid until the next emission call
    // the selected emission as one packed device blob ((contig, stream) major; sizes[n*6]) and its counters without the bytes
    // (n*6: unmatchedChars, extensionsMatchedChars, extensionsMismatches, totalMatched, removedGapBreakingMatches, matches)
    uint64_t emitPack(uint8_t *dstDev, uint64_t cap, std::vector<uint64_t> *sizes, int n);
    void emitCounters(std::vector<uint64_t> &out, int n);
    void emitEnd();
    // the selected emission decoded again on the device by the decoder's automaton (MBGC_Decoder.cpp:319-523; include/mbgc_swsem.h:
    // swsem_emit_verify) and compared with the contigs it was emitted for: contigs that fail, the first of them, its first differing byte
    int emitVerify(int *firstBad = nullptr, uint64_t *firstDiff = nullptr);
    void finalizeTargets(const std::vector<const uint8_t *> &extDev, const std::vector<uint64_t> &extLen, bool addSep, char sep,
                         bool lazySeparator, const std::vector<uint64_t> &lockPos, std::vector<uint64_t> &loadedAfter);
    void synchronize();
    void loadRefDev(const uint8_t *textDev, size_t len, bool loadRC, bool addSep, char sep);
    void devDownload(void *dst, const uint8_t *srcDev, size_t bytes);
    uint8_t *devAlloc(size_t bytes);
    void devFree(uint8_t *p);
    void devUpload(uint8_t *dst, const void *src, size_t bytes);
    void devCopy(uint8_t *dst, const uint8_t *src, size_t bytes);
    void devRevComp(const uint8_t *src, size_t n, uint8_t *dst);
    swsem_t *handle() { return h; }

private:
    swsem_t *h = nullptr;
    [[noreturn]] void die(const char *what) const;
    void check(int rc, const char *what) const { if (rc != SWSEM_OK) die(what); }
};
