#include "sw_matcher.h"

#include <atomic>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

using PgTools::TextMatch;

void SlidingWindowSparseEMMatcher::die(const char *what) const {
    // the reference prints its message and exits (e.g. SlidingWindowSparseEMMatcher.cpp:82-85,186-187,389-390,481-482)
    fprintf(stderr, "%s: %s\n\n", what, swsem_last_error());
    exit(EXIT_FAILURE);
}

SlidingWindowSparseEMMatcher::SlidingWindowSparseEMMatcher(size_t refLengthLimit, uint32_t targetMatchLength, int k1, int k2,
                                                           int skipMargin, int device) {
    check(swsem_create(&h, refLengthLimit, (int) targetMatchLength, k1, k2, skipMargin, device), "Error initializing ExpSparseMEM");
}

SlidingWindowSparseEMMatcher::~SlidingWindowSparseEMMatcher() { swsem_destroy(h); }

void SlidingWindowSparseEMMatcher::disableSlidingWindow() { swsem_disable_sliding_window(h); }
void SlidingWindowSparseEMMatcher::disableCircularBuffer() { swsem_disable_circular_buffer(h); }
void SlidingWindowSparseEMMatcher::setSlidingWindowSize(uint8_t factor) { swsem_set_sliding_window_size(h, factor); }
size_t SlidingWindowSparseEMMatcher::getMaxRefLength() const { return swsem_get_max_ref_length(h); }
size_t SlidingWindowSparseEMMatcher::getSlidingWindowSize() const { return swsem_get_sliding_window_size(h); }
size_t SlidingWindowSparseEMMatcher::getDroppedBytes() const { return swsem_get_dropped_bytes(h); }
size_t SlidingWindowSparseEMMatcher::getRefLength() const { return swsem_get_ref_length(h); }
size_t SlidingWindowSparseEMMatcher::getLoadingPosition() const { return swsem_get_loading_position(h); }
size_t SlidingWindowSparseEMMatcher::getLoadedRefLength() const { return swsem_get_loaded_ref_length(h); }
void SlidingWindowSparseEMMatcher::setPosition(size_t refPos, int laps) { swsem_set_position(h, refPos, laps); }
size_t SlidingWindowSparseEMMatcher::acquireWorkerMatchingLockPos() { return swsem_acquire_lock(h); }

void SlidingWindowSparseEMMatcher::releaseWorkerMatchingLockPos(size_t lockValue) {
    check(swsem_release_lock(h, lockValue), "ERROR");
}

void SlidingWindowSparseEMMatcher::loadRef(const char *refText, size_t refLength, bool loadRCRef, bool addRegionSeparators,
                                           char regionSeparator) {
    check(swsem_load_ref(h, (const uint8_t *) refText, refLength, loadRCRef, addRegionSeparators, (unsigned char) regionSeparator), "loadRef");
}

void SlidingWindowSparseEMMatcher::loadRefDev(const uint8_t *textDev, size_t len, bool loadRC, bool addSep, char sep) {
    check(swsem_load_ref_dev(h, textDev, len, loadRC, addSep, (unsigned char) sep), "loadRef");
}

void SlidingWindowSparseEMMatcher::loadSeparator(char regionSeparator) {
    check(swsem_load_separator(h, (unsigned char) regionSeparator), "loadSeparator");
}

void SlidingWindowSparseEMMatcher::matchTexts(std::vector<TextMatch> &res, const std::string &destText, bool destIsRef,
                                              bool revComplMatching, uint32_t minMatchLength, size_t matchingLockPos) {
    matchTexts(res, destText.data(), destText.size(), destIsRef, revComplMatching, minMatchLength, matchingLockPos);
}

void SlidingWindowSparseEMMatcher::matchTexts(std::vector<TextMatch> &res, const char *destText, size_t destLen, bool destIsRef,
                                              bool revComplMatching, uint32_t minMatchLength, size_t matchingLockPos) {
    if (destIsRef || revComplMatching) {        // SlidingWindowSparseEMMatcher.cpp:185-188
        fprintf(stderr, "Source as destination and reverse-complement matching modes unsupported\n\n");
        exit(EXIT_FAILURE);
    }
    // The handle serves one matchTexts / processMatches pair at a time (the rows stay on it between the two calls). A second
    // thread inside matchTexts — the reference's default schedule on this class (MGMP.cpp:520-555) — would launch on the same
    // buffers: refused before it touches the device, without the exit handlers (the first thread is inside the HIP runtime).
    static std::atomic<int> inside(0);
    if (inside.fetch_add(1) != 0) {
        fprintf(stderr, "mbgc (HIP matcher): the device matcher serves one matchTexts/processMatches pair at a time: run the sequential "
                        "schedule (-t1), or the round schedule of mbgc-hip\n");
        fflush(stderr);
        _exit(EXIT_FAILURE);
    }
    res.clear();                                // .cpp:484
    const swsem_match_t *m = nullptr;
    uint64_t n = 0;
    check(swsem_match(h, (const uint8_t *) destText, destLen, minMatchLength, matchingLockPos, &m, &n), "matchTexts");
    res.reserve(n);
    for (uint64_t i = 0; i < n; i++) res.emplace_back(m[i].posSrcText, m[i].length, m[i].posDestText);
    inside.fetch_sub(1);
}

static void take(const swsem_streams_t &st, EmittedStreams &out) {
    for (int s = 0; s < SWSEM_NSTREAMS; s++) out.s[s].assign((const char *) st.data[s], st.size[s]);
    out.unmatchedChars = st.unmatchedChars;
    out.extensionsMatchedChars = st.extensionsMatchedChars;
    out.extensionsMismatches = st.extensionsMismatches;
    out.totalMatched = st.totalMatched;
    out.removedGapBreakingMatches = st.removedGapBreakingMatches;
    out.nmatches = st.nmatches;
}

size_t SlidingWindowSparseEMMatcher::processMatches(const swsem_emit_params_t &p, size_t matchingLockPos, int factor,
                                                    int64_t processed, int64_t targetIdx, const std::vector<size_t> &loaded,
                                                    EmittedStreams &out) {
    swsem_streams_t st;
    std::vector<uint64_t> ld(loaded.begin(), loaded.end());
    check(swsem_emit(h, &p, 0, matchingLockPos, factor, processed, targetIdx, ld.data(), ld.size(), &st), "processMatches");
    take(st, out);
    return st.unmatchedChars;
}

void SlidingWindowSparseEMMatcher::matchRound(const uint8_t *contigsDev, const std::vector<uint64_t> &offsets, uint32_t minLen,
                                              const std::vector<uint64_t> &lockPos, std::vector<uint64_t> &counts) {
    const int n = (int) offsets.size() - 1;
    check(swsem_match_batch_dev(h, contigsDev, offsets.data(), n, minLen, lockPos.empty() ? nullptr : lockPos.data()), "matchTexts");
    counts.assign(n, 0);
    check(swsem_batch_counts(h, counts.data()), "matchTexts");
}

void SlidingWindowSparseEMMatcher::emitRound(const swsem_emit_params_t &p, const std::vector<uint64_t> &lockPos,
                                             const std::vector<int> &factors, const std::vector<int64_t> &processed,
                                             const std::vector<int64_t> &targetIdx, const std::vector<size_t> &loaded,
                                             std::vector<EmittedStreams> &out) {
    const int n = (int) lockPos.size();
    std::vector<uint64_t> ld(loaded.begin(), loaded.end());
    check(swsem_emit_batch(h, &p, n, nullptr, lockPos.data(), factors.data(), processed.data(), targetIdx.data(), ld.data(), ld.size()),
          "processMatches");
    out.assign(n, EmittedStreams());
    for (int k = 0; k < n; k++) {
        swsem_streams_t st;
        check(swsem_emit_result(h, k, &st), "processMatches");
        take(st, out[k]);
    }
}

void SlidingWindowSparseEMMatcher::matchRoundBegin(const uint8_t *contigsDev, const std::vector<uint64_t> &offsets, uint32_t minLen,
                                                   const std::vector<uint64_t> &lockPos) {
    const int n = (int) offsets.size() - 1;
    check(swsem_match_batch_dev(h, contigsDev, offsets.data(), n, minLen, lockPos.empty() ? nullptr : lockPos.data()), "matchTexts");
}

bool SlidingWindowSparseEMMatcher::emitRoundBegin(const swsem_emit_params_t &p, const std::vector<uint64_t> &lockPos,
                                                  const std::vector<int> &factors, const std::vector<int64_t> &processed,
                                                  const std::vector<int64_t> &targetIdx, const std::vector<size_t> &loaded,
                                                  const swsem_spec_finalize_t *spec, std::vector<uint64_t> &unmatched,
                                                  std::vector<uint64_t> &counts) {
    const int n = (int) lockPos.size();
    std::vector<uint64_t> ld(loaded.begin(), loaded.end());
    int applied = 0;
    if (spec)
        check(swsem_emit_batch_begin_spec(h, &p, n, nullptr, lockPos.data(), factors.data(), processed.data(), targetIdx.data(), ld.data(),
                                          ld.size(), spec, &applied), "processMatches");
    else
        check(swsem_emit_batch_begin(h, &p, n, nullptr, lockPos.data(), factors.data(), processed.data(), targetIdx.data(), ld.data(), ld.size()),
              "processMatches");
    unmatched.assign(n, 0);
    check(swsem_emit_unmatched(h, unmatched.data()), "processMatches");
    counts.assign(n, 0);
    check(swsem_batch_counts(h, counts.data()), "matchTexts");            // (after the emission's launches: no round trip in between)
    return applied != 0;
}

void SlidingWindowSparseEMMatcher::setEmitHostCopy(bool on) { swsem_emit_set_host_copy(h, on ? 1 : 0); }
void SlidingWindowSparseEMMatcher::emitSelect(bool previous) { check(swsem_emit_select(h, previous ? 1 : 0), "processMatches"); }

void SlidingWindowSparseEMMatcher::emitTake(int k, EmittedStreams &out) {
    swsem_streams_t st;
    check(swsem_emit_result(h, k, &st), "processMatches");
    take(st, out);
}

uint64_t SlidingWindowSparseEMMatcher::emitPack(uint8_t *dstDev, uint64_t cap, std::vector<uint64_t> *sizes, int n) {
    uint64_t total = 0;
    if (sizes) sizes->assign((size_t) n * SWSEM_NSTREAMS, 0);
    check(swsem_emit_pack_dev(h, dstDev, cap, sizes ? sizes->data() : nullptr, &total), "processMatches");
    return total;
}

void SlidingWindowSparseEMMatcher::emitCounters(std::vector<uint64_t> &out, int n) {
    out.assign((size_t) n * 6, 0);
    check(swsem_emit_counters(h, out.data()), "processMatches");
}

void SlidingWindowSparseEMMatcher::emitView(int k, swsem_streams_t &view) { check(swsem_emit_result(h, k, &view), "processMatches"); }

void SlidingWindowSparseEMMatcher::emitEnd() { check(swsem_emit_batch_end(h), "processMatches"); }
int SlidingWindowSparseEMMatcher::emitVerify(int *firstBad, uint64_t *firstDiff) {
    int nbad = 0;
    check(swsem_emit_verify(h, &nbad, firstBad, firstDiff), "decodeSequence");
    return nbad;
}

void SlidingWindowSparseEMMatcher::finalizeTargets(const std::vector<const uint8_t *> &extDev, const std::vector<uint64_t> &extLen,
                                                   bool addSep, char sep, bool lazySeparator, const std::vector<uint64_t> &lockPos,
                                                   std::vector<uint64_t> &loadedAfter) {
    loadedAfter.assign(extLen.size(), 0);
    check(swsem_finalize_targets(h, (int) extLen.size(), extDev.data(), extLen.data(), addSep, (unsigned char) sep, lazySeparator,
                                 lockPos.data(), loadedAfter.data()), "loadRef");
}

void SlidingWindowSparseEMMatcher::synchronize() { check(swsem_synchronize(h), "synchronize"); }
void SlidingWindowSparseEMMatcher::devDownload(void *dst, const uint8_t *srcDev, size_t bytes) { check(swsem_dev_download(h, dst, srcDev, bytes), "download"); }

uint8_t *SlidingWindowSparseEMMatcher::devAlloc(size_t bytes) {
    void *p = nullptr;
    check(swsem_dev_malloc(h, bytes, &p), "device allocation");
    return (uint8_t *) p;
}
void SlidingWindowSparseEMMatcher::devFree(uint8_t *p) { check(swsem_dev_free(h, p), "device free"); }
void SlidingWindowSparseEMMatcher::devUpload(uint8_t *dst, const void *src, size_t bytes) { check(swsem_dev_upload(h, dst, src, bytes), "upload"); }
void SlidingWindowSparseEMMatcher::devCopy(uint8_t *dst, const uint8_t *src, size_t bytes) { check(swsem_dev_copy(h, dst, src, bytes), "copy"); }
void SlidingWindowSparseEMMatcher::devRevComp(const uint8_t *src, size_t n, uint8_t *dst) { check(swsem_revcomp_dev(h, src, n, dst), "revcomp"); }
