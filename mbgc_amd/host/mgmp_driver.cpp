#include "mgmp_driver.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

using PgTools::TextMatch;

static const char SEQ_SEPARATOR_MARK = (char) ('"' + 128);     // MBGC_Params.h:46
static const char FILE_SEPARATOR_MARK = (char) (';' + 128);    // MBGC_Params.h:47
static const char REF_REGION_SEPARATOR = 0;                    // MGMP_Params.h:14

void MBGC_Params::setCompressionMode(int mode) {                // MBGC_Params.h:886-922
    if (mode < 0 || mode > 3) {
        fprintf(stderr, "Compression mode should be between %d and %d.\n", 0, 3);
        exit(EXIT_FAILURE);
    }
    coderMode = (uint8_t) mode;
    swsem_emit_params_default(&emit, mode);
    if (mode >= 2) {
        bigReferenceCompressorRatio = 4;
        skipMargin = 24;
        unmatchedFractionRCFactor = 128;
    }
    if (mode == 3) sequentialMatching = true;
}

// PgHelpers::writeUInt64Frugal, utils/helper.cpp:237-246
static void writeUInt64Frugal(std::string &dest, uint64_t value) {
    uint16_t y16 = value < UINT16_MAX ? (uint16_t) value : UINT16_MAX;
    dest.append((const char *) &y16, 2);
    if (value >= UINT16_MAX) {
        uint32_t y32 = value < UINT32_MAX ? (uint32_t) value : UINT32_MAX;
        dest.append((const char *) &y32, 4);
        if (value >= UINT32_MAX) dest.append((const char *) &value, 8);
    }
}

// Whole-file read + lossless FASTA split (the role of mgmpInOpen + kseq_read_lossless_fasta, MGMP.cpp:7-14,
// utils/kseq.h:233-274): headers without '>', sequence lines joined, line ends dropped.
bool readFastaFile(const std::string &path, std::vector<Contig> &out, uint64_t *fileSize) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (fileSize) *fileSize = data.size();
    out.clear();
    size_t p = 0;
    const size_t n = data.size();
    if (n && data[0] != '>') {
        fprintf(stderr, "Error parsing file %s - expected FASTA format.\n", path.c_str());
        exit(EXIT_FAILURE);
    }
    while (p < n) {
        Contig c;
        size_t e = data.find('\n', p);
        if (e == std::string::npos) e = n;
        c.header = data.substr(p + 1, e - p - 1);
        if (!c.header.empty() && c.header.back() == '\r') c.header.pop_back();
        p = e < n ? e + 1 : n;
        while (p < n && data[p] != '>') {
            e = data.find('\n', p);
            if (e == std::string::npos) e = n;
            size_t le = e;
            if (le > p && data[le - 1] == '\r') le--;
            c.seq.append(data, p, le - p);
            p = e < n ? e + 1 : n;
        }
        out.push_back(std::move(c));
    }
    return true;
}

// ---------------------------------------------------------------- MultipleGenomeMatchingProcessor

void MultipleGenomeMatchingProcessor::initMatcher(const char *refStr, size_t refStrSize, size_t basicRefLength) {
    size_t refLengthLimit = (size_t) params->referenceFactor * basicRefLength;                 // MGMP.cpp:154
    refLengthLimit *= 2;                                                                       // :157-158 (RC kept in the same buffer)
    if (refLengthLimit <= UINT32_MAX) params->enable40bitReference = false;                    // :159-160
    else if (!params->enable40bitReference) refLengthLimit = UINT32_MAX;
    else if (refLengthLimit > MGMP_Params::REFERENCE_LENGTH_LIMIT) refLengthLimit = MGMP_Params::REFERENCE_LENGTH_LIMIT;
    if (refLengthLimit > UINT32_MAX)
        refLengthLimit = UINT32_MAX + (refLengthLimit - UINT32_MAX) / params->bigReferenceCompressorRatio;   // :165-166
    if (refLengthLimit < refStrSize) refLengthLimit = refStrSize;
    matcher = new SlidingWindowSparseEMMatcher(refLengthLimit, params->k, params->k1, params->k2, params->skipMargin, device);
    if (params->sequentialMatching) matcher->disableSlidingWindow();                           // :177-180
    else matcher->setSlidingWindowSize((uint8_t) params->referenceSlidingWindowFactor);
    if (!params->circularReference) matcher->disableCircularBuffer();
    matcher->loadRef(refStr, refStrSize, params->rcInReference, params->refRegionSeparators, REF_REGION_SEPARATOR);   // :183
}

void MultipleGenomeMatchingProcessor::loadG0Ref(const std::string &refName) {
    std::vector<Contig> contigs;
    uint64_t fileSize = 0;
    if (!readFastaFile(refName, contigs, &fileSize)) {
        fprintf(stderr, "cannot open file %s\n", refName.c_str());
        exit(EXIT_FAILURE);
    }
    std::string refStr;
    initStreamsForG0Ref();
    for (const Contig &c : contigs) {                                                          // MGMP.cpp:82-105
        largestContigSize = std::max<uint64_t>(largestContigSize, c.seq.size());
        refStr.append(c.seq);
        processG0RefContig(c.seq.data(), c.seq.size());
        if (params->sequentialMatching) break;                                                 // :91-100: first contig only
    }
    refG0InitPos = refStr.size();
    size_t basicRefLength = params->sequentialMatching ? fileSize : refStr.size();             // :109
    basicRefLength = std::max<size_t>(basicRefLength, MGMP_Params::MIN_BASIC_BLOCK_SIZE);
    targetsCount = params->sequentialMatching ? filesCount : filesCount - 1;                   // :117-118
    if (params->referenceFactor < 1) {                                                         // :130-134
        int tmp = 15 - (__builtin_clz((unsigned) filesCount) / 3);
        tmp = tmp < 5 ? 5 : (tmp > 12 ? 12 : tmp);
        params->referenceFactor = 1 << tmp;
    }
    initMatcher(refStr.data(), refStr.size(), basicRefLength);
}

void MultipleGenomeMatchingProcessor::processTargetsWithParallelIO() {
    std::vector<TextMatch> resMatches;
    for (uint32_t i = 0; i < filesCount; i++) {
        std::vector<Contig> contigs;
        uint64_t fileSize = 0;
        if (!readFastaFile(fileNames[i], contigs, &fileSize)) { fprintf(stderr, "cannot open file %s\n", fileNames[i].c_str()); exit(EXIT_FAILURE); }
        const size_t startPos = matcher->getLoadedRefLength();                                 // MGMP.cpp:251
        unmatchedFractionFactors.push_back(params->currentUnmatchedFractionFactor < 256 ? params->currentUnmatchedFractionFactor : 0);
        unmatchedFractionFactors.push_back((uint8_t) params->unmatchedFractionRCFactor);
        totalFilesLength += fileSize;
        for (const Contig &c : contigs) {
            largestContigSize = std::max<uint64_t>(largestContigSize, c.seq.size());
            const size_t bSize = c.seq.size();
            matcher->matchTexts(resMatches, c.seq.data(), bSize, false, false, params->k);     // :274
            resCount += resMatches.size();
            const size_t currentUnmatched = processMatches(bSize, 0, SIZE_MAX);                // :276
            const bool loadContigToRef = params->isContigProperForRefExtension(bSize, currentUnmatched, params->currentUnmatchedFractionFactor);
            const bool loadContigRCToRef = params->rcInReference &&
                                           params->isContigProperForRefRCExtension(bSize, currentUnmatched, params->unmatchedFractionRCFactor);
            // :281-287: the contig, or the (always empty in release builds) literal extension
            matcher->loadRef(c.seq.data(), loadContigToRef ? bSize : 0, loadContigRCToRef, params->refRegionSeparators, REF_REGION_SEPARATOR);
            processAfterSequence(0);
        }
        processAfterTargetWithParallelIO(startPos);                                            // :306
    }
}

void MultipleGenomeMatchingProcessor::processTargetsRounds() {
    initParallelProcessing();
    matchingLocksPos.assign(targetsCount, SIZE_MAX);
    unmatchedFractionFactors.assign(2 * (size_t) targetsCount, 0);
    const int R = std::max(1, params->roundSize);
    for (uint32_t r0 = 0; r0 < targetsCount; r0 += R) {
        const uint32_t r1 = std::min<uint32_t>(targetsCount, r0 + R);
        // read the round's files and put their contigs back to back in HBM
        std::vector<std::vector<Contig>> files(r1 - r0);
        std::vector<uint64_t> offsets(1, 0);
        std::vector<uint32_t> targetOf;
        std::string all;
        for (uint32_t t = r0; t < r1; t++) {
            uint64_t fileSize = 0;
            if (!readFastaFile(fileNames[t + 1], files[t - r0], &fileSize)) { fprintf(stderr, "cannot open file %s\n", fileNames[t + 1].c_str()); exit(EXIT_FAILURE); }
            totalFilesLength += fileSize;
            for (const Contig &c : files[t - r0]) {
                largestContigSize = std::max<uint64_t>(largestContigSize, c.seq.size());
                all.append(c.seq);
                offsets.push_back(all.size());
                targetOf.push_back(t);
            }
            unmatchedFractionFactors[2 * t] = params->currentUnmatchedFractionFactor < 256 ? params->currentUnmatchedFractionFactor : 0;   // :351-352
            unmatchedFractionFactors[2 * t + 1] = (uint8_t) params->unmatchedFractionRCFactor;
            matchingLocksPos[t] = matcher->acquireWorkerMatchingLockPos();                      // :353-358
        }
        const size_t ncont = targetOf.size();
        uint8_t *dev = matcher->devAlloc(all.size() + 64);
        matcher->devUpload(dev, all.data(), all.size());
        std::vector<int> pending(ncont);
        for (size_t c = 0; c < ncont; c++) pending[c] = (int) c;
        std::vector<uint64_t> unmatched(ncont, SIZE_MAX);
        std::vector<EmittedStreams> emitted(ncont);
        uint32_t finalized = r0;                                                                // == processedTargetsCount
        while (true) {
            int cut = (int) ncont;                                                              // first contig that has to be retried
            if (!pending.empty()) {
                // contigs from the first pending one on are consecutive in the buffer (everything after a cut is redone)
                const int c0 = pending.front();
                std::vector<uint64_t> offs, locks, counts;
                std::vector<int> factors;
                std::vector<int64_t> processed, tidx;
                for (int c : pending) {
                    offs.push_back(offsets[c] - offsets[c0]);
                    locks.push_back(matchingLocksPos[targetOf[c]]);
                    factors.push_back(unmatchedFractionFactors[2 * targetOf[c]]);
                    processed.push_back(processedTargetsCount);
                    tidx.push_back(targetOf[c]);
                }
                offs.push_back(offsets[pending.back() + 1] - offsets[c0]);
                matcher->matchRound(dev + offsets[c0], offs, params->k, locks, counts);        // :379
                std::vector<EmittedStreams> out;
                matcher->emitRound(emitParams(), locks, factors, processed, tidx, loadedPositions(), out);   // :381
                for (size_t k = 0; k < pending.size(); k++) {
                    const int c = pending[k];
                    if (out[k].unmatchedChars == PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY) {   // :382-388
                        cut = std::min(cut, c);
                        continue;
                    }
                    if (c < cut) {
                        unmatched[c] = out[k].unmatchedChars;
                        resCount += counts[k];
                        emitted[c] = std::move(out[k]);
                    }
                }
            }
            // targets before the one holding the cut are complete: load their extensions in order (:433-468)
            const uint32_t upto = cut < (int) ncont ? targetOf[cut] : r1;
            for (uint32_t t = finalized; t < upto; t++) {
                const size_t startPos = matcher->getLoadedRefLength();
                size_t extLen = 0;
                for (size_t c = 0; c < ncont; c++)
                    if (targetOf[c] == t) {
                        const size_t len = offsets[c + 1] - offsets[c];
                        if (params->isContigProperForRefExtension(len, unmatched[c], unmatchedFractionFactors[2 * t])) extLen += len;     // :389-392
                        if (params->rcInReference && params->isContigProperForRefRCExtension(len, unmatched[c], params->unmatchedFractionRCFactor)) extLen += len;
                    }
                if (extLen) {
                    uint8_t *ext = matcher->devAlloc(extLen);
                    size_t pos = 0;
                    for (size_t c = 0; c < ncont; c++)
                        if (targetOf[c] == t) {
                            const size_t len = offsets[c + 1] - offsets[c];
                            if (params->isContigProperForRefExtension(len, unmatched[c], unmatchedFractionFactors[2 * t])) {
                                matcher->devCopy(ext + pos, dev + offsets[c], len);
                                pos += len;
                            }
                            if (params->rcInReference && params->isContigProperForRefRCExtension(len, unmatched[c], params->unmatchedFractionRCFactor)) {
                                matcher->devRevComp(dev + offsets[c], len, ext + pos);          // :393-398
                                pos += len;
                            }
                        }
                    matcher->loadRefDev(ext, extLen, false, params->refRegionSeparators, REF_REGION_SEPARATOR);   // :441-443
                    matcher->devFree(ext);
                }
                // the target's streams: contig by contig, then the target separator
                for (size_t c = 0; c < ncont; c++)
                    if (targetOf[c] == t) {
                        takeRoundStreams(t, emitted[c]);
                        processAfterSequence(t);
                    }
                processAfterTarget(t);
                finalizeParallelProcessingOfTarget(t, startPos);                                // :455
                matcher->releaseWorkerMatchingLockPos(matchingLocksPos[t]);                     // :456
                processedTargetsCount = t + 1;
            }
            finalized = upto;
            if (cut >= (int) ncont) break;
            pending.clear();
            for (int c = cut; c < (int) ncont; c++) pending.push_back(c);
        }
        matcher->devFree(dev);
    }
}

void MultipleGenomeMatchingProcessor::performMatching() {
    if (params->sequentialMatching) processTargetsWithParallelIO();
    else if (targetsCount) processTargetsRounds();
    else {
        fprintf(stderr, "Error selecting processing mode (no targets for parallel matching?)!\n");
        exit(EXIT_FAILURE);
    }
    refFinalTotalLength = matcher->getRefLength();
}

// ---------------------------------------------------------------- MBGC_Encoder

void MBGC_Encoder::initStreamsForG0Ref() { literals.clear(); }
void MBGC_Encoder::processG0RefContig(const char *seq, size_t len) {
    literals.append(seq, len);
    literals.push_back(SEQ_SEPARATOR_MARK);
}

static void appendStreams(MBGC_Encoder &e, const EmittedStreams &s) {
    e.literals.append(s.s[SWSEM_LIT]);
    e.mapOff.append(s.s[SWSEM_OFF]);
    e.mapOff5thByte.append(s.s[SWSEM_OFF5]);
    e.mapLen.append(s.s[SWSEM_LEN]);
    e.gapDeltas.append(s.s[SWSEM_GAP]);
    e.gapMismatchesFlags.append(s.s[SWSEM_FLAGS]);
}

size_t MBGC_Encoder::processMatches(size_t destLen, int targetIdx, size_t matchingLockPos) {
    EmittedStreams s;
    const int factor = unmatchedFractionFactors[2 * (size_t) targetIdx];                        // ENC.cpp:202
    const size_t um = matcher->processMatches(params->emit, matchingLockPos, factor, processedTargetsCount, targetIdx, refExtLoadedPosArr, s);
    if (um == PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY) return um;
    appendStreams(*this, s);
    unmatchedCharsAll += um;                                                                    // :293-306
    extensionsMatchedCharsAll += s.extensionsMatchedChars;
    extensionsMismatchesAll += s.extensionsMismatches;
    totalMatchedAll += s.totalMatched;
    totalDestLenAll += destLen;
    removedGapBreakingMatchesAll += s.removedGapBreakingMatches;
    return um;
}

void MBGC_Encoder::takeRoundStreams(uint32_t targetIdx, EmittedStreams &s) {
    EmittedStreams &t = targetStreams[targetIdx];
    for (int i = 0; i < SWSEM_NSTREAMS; i++) t.s[i].append(s.s[i]);
    unmatchedCharsAll += s.unmatchedChars;
    extensionsMatchedCharsAll += s.extensionsMatchedChars;
    extensionsMismatchesAll += s.extensionsMismatches;
    totalMatchedAll += s.totalMatched;
    removedGapBreakingMatchesAll += s.removedGapBreakingMatches;
}

void MBGC_Encoder::processAfterSequence(uint32_t targetIdx) {
    if (params->sequentialMatching) literals.push_back(SEQ_SEPARATOR_MARK);
    else targetStreams[targetIdx].s[SWSEM_LIT].push_back(SEQ_SEPARATOR_MARK);
}

void MBGC_Encoder::processAfterTarget(uint32_t targetIdx) {
    if (!params->emit.enableExtensionsWithMismatches) return;
    if (params->sequentialMatching) gapMismatchesFlags.push_back(FILE_SEPARATOR_MARK);
    else targetStreams[targetIdx].s[SWSEM_FLAGS].push_back(FILE_SEPARATOR_MARK);
}

void MBGC_Encoder::processAfterTargetWithParallelIO(size_t matcherLoaderStartPos) {
    processAfterTarget(0);
    if (params->lazyDecompressionSupport) {
        matcher->loadSeparator(REF_REGION_SEPARATOR);
        const size_t refExtSize = matcher->getLoadedRefLength() - matcherLoaderStartPos;
        writeUInt64Frugal(refExtSizeStream, refExtSize);
        refExtLoadedPosArr.emplace_back(refExtLoadedPosArr.back() + refExtSize);
    }
    const size_t tmp = matcher->acquireWorkerMatchingLockPos();
    locksPosStream.append((const char *) &tmp, sizeof(tmp));
    matcher->releaseWorkerMatchingLockPos(tmp);
}

void MBGC_Encoder::initParallelProcessing() { targetStreams.assign(targetsCount, EmittedStreams()); }

void MBGC_Encoder::finalizeParallelProcessingOfTarget(uint32_t targetIdx, size_t matcherLoaderStartPos) {
    appendStreams(*this, targetStreams[targetIdx]);                                             // ENC.cpp:543-556
    targetStreams[targetIdx] = EmittedStreams();
    if (params->lazyDecompressionSupport) {
        matcher->loadSeparator(REF_REGION_SEPARATOR);
        const size_t refExtSize = matcher->getLoadedRefLength() - matcherLoaderStartPos;
        writeUInt64Frugal(refExtSizeStream, refExtSize);
        refExtLoadedPosArr.emplace_back(matcher->getLoadedRefLength());
    }
    locksPosStream.append((const char *) &matchingLocksPos[targetIdx], sizeof(size_t));         // :563
}

void MBGC_Encoder::encode(const std::vector<std::string> &files) {
    fileNames = files;
    filesCount = (uint32_t) fileNames.size();
    if (!filesCount) {
        fprintf(stderr, "ERROR: filelist is empty.\n");
        exit(EXIT_FAILURE);
    }
    if (filesCount == 1) params->sequentialMatching = true;                                     // no targets for the round loop
    params->emit.lazyDecompressionSupport = params->lazyDecompressionSupport;
    loadG0Ref(fileNames[0]);
    params->emit.enable40bitReference = params->enable40bitReference;
    if (params->lazyDecompressionSupport) refExtLoadedPosArr.emplace_back(matcher->getLoadingPosition());   // ENC.cpp:789-791
    performMatching();
}
