// Host-side drivers of the compress hot path with the reference's class names and hook surface:
//   MultipleGenomeMatchingProcessor  matching/MultipleGenomeMatchingProcessor.{h,cpp}  (file list, G0
//       reference, matcher sizing, the two target loops, in-order reference extension)
//   MBGC_Encoder                     mbgccoder/MBGC_Encoder.{h,cpp}:128-564,759-807      (the hooks
//       that collect the match/literal byte streams, and prepareAndCompressStreams' job table + container framing
//       (:641-710, include/mbgc_backend.h) around the PPMd/LZMA coders, which are NOT part of this repo: they are the
//       reference's unchanged backend, handed in as a callback)
// Policy (which contigs extend the reference, dissimilar-contig retry, lock bookkeeping) stays on the
// host exactly as in the reference; everything that touches sequence bytes runs through the C ABI.
#pragma once
#include <cstdint>
#include <future>
#include <string>
#include <vector>

#include "sw_matcher.h"
#include "../../include/mbgc_fasta.h"
#include "../../include/mbgc_exchange.h"
#include "../../include/mbgc_backend.h"

struct MGMP_Params {                                   // matching/MGMP_Params.h (only what this path reads)
    int k = 32, k1 = 16, k2 = 1;                       // :199-202
    uint8_t skipMargin = 16;                           // :203 (24 in -m2/-m3, MBGC_Params.h:908-909)
    int referenceFactor = -1;                          // ADJUSTED_REFERENCE_FACTOR_FLAG, :24,:205
    int referenceSlidingWindowFactor = 16;             // :206
    bool enable40bitReference = true;                  // :208
    bool circularReference = true;                     // :209
    uint8_t bigReferenceCompressorRatio = 16;          // :210 (4 in -m2/-m3)
    bool refRegionSeparators = true;                   // :12
    int currentUnmatchedFractionFactor = 128;          // :71
    int unmatchedFractionRCFactor = 8;                 // :73 (128 in -m2/-m3)
    bool rcInReference = true;                         // !isRCinReferenceDisabled(), :82-84
    bool sequentialMatching = false;                   // :218
    // Targets per round and GPU: the targets of a round hold their lock positions together (the deterministic stand-in for the
    // reference's worker threads, :16-22). 0 = as many as the reference's own rules let be in flight: what they load must fit the
    // sliding window once the buffer has wrapped (their shared lock stands one window ahead of the loading position and loadRef
    // drops what goes beyond it, SlidingWindowSparseEMMatcher.cpp:361-378,412-417,433), and at most matcherWorkingThreads x
    // allowedTargetsOutrunFactor = 64 targets run ahead of the finalizer (MGMP.cpp:374-375,532, :16,:57) — divided by the GPUs.
    int roundSize = 0;
    static constexpr int MAX_TARGETS_IN_FLIGHT = 64;
    bool uppercaseDNA = false;                         // :196 (-U)
    // mbgc-hip c --bench: every round's contigs are put into HBM first, the rounds after `benchWarmup` are timed and the
    // emitted streams stay packed in HBM (what bench.py measures, from the C++ host)
    bool benchMode = false;
    // mbgc-hip c --verify: every emission is decoded again on the device (SlidingWindowSparseEMMatcher::emitVerify) before the
    // reference moves on, and the run fails on the first contig that does not come back; rounds then load on the host's decision
    bool verifyEmissions = false;
    int verifyEvery = 1;                               // --verify-every K: only the emissions of every K-th target file (sequential schedule) / round
    uint64_t verifiedContigs = 0, verifiedBases = 0;
    int benchWarmup = 0;
    double benchSeconds = 0; uint64_t benchBases = 0; int benchRounds = 0;   // filled by processTargetsRounds
    // several GPUs: this process is one rank of `exchange`; a round then holds roundSize targets PER RANK
    mbgc_xchg_t *exchange = nullptr;
    int specRounds = 0;                                // rounds whose finalize ran on the ranks' device-side verdicts (diagnostics)
    int headRounds = 0;                                // rounds whose extension exchange was cut to the loadable head (diagnostics)
    static constexpr uint64_t MIN_BASIC_BLOCK_SIZE = 1 << 21;             // :48
    static constexpr uint64_t REFERENCE_LENGTH_LIMIT = (uint64_t) UINT32_MAX << 8;   // :55
    bool isContigProperForRefExtension(uint64_t len, uint64_t unmatched, int f) const { return unmatched * f > len; }      // :179-186
    bool isContigProperForRefRCExtension(uint64_t len, uint64_t unmatched, int f) const { return unmatched * f > len; }    // :188-190
};

struct MBGC_Params : MGMP_Params {                     // mbgccoder/MBGC_Params.h (only what this path reads)
    uint8_t coderMode = 1;
    bool lazyDecompressionSupport = true;              // :38
    uint64_t rcMatchMinLength = 0;                     // :99 (55 in -m3, :61,:917-918)
    bool rcRedundancyRemoval = false;                  // :101
    swsem_emit_params_t emit;
    int device = 0;
    MBGC_Params() { setCompressionMode(1); }
    void setCompressionMode(int mode);                 // :886-922
};

struct Contig { std::string header, seq; };
// A round's files parsed by the device input stage (include/mbgc_fasta.h: kseq_read_lossless_fasta, MGMP.cpp:349-372):
// their contigs back to back in HBM, the layout matchRound takes.
struct RoundBatch {
    uint8_t *seqDev = nullptr; size_t seqCap = 0;      // grow-only device buffer
    uint64_t bytes = 0;
    std::vector<uint64_t> offsets;                     // ncont + 1
    std::vector<uint32_t> targetOf;                    // target index of every contig
    uint32_t t0 = 0, t1 = 0;                           // targets [t0, t1)
};

class MultipleGenomeMatchingProcessor {
protected:
    MGMP_Params *params;
    SlidingWindowSparseEMMatcher *matcher = nullptr;
    std::vector<std::string> fileNames;
    uint32_t filesCount = 0, targetsCount = 0;
    int64_t processedTargetsCount = 0;
    uint64_t refG0InitPos = 0, refFinalTotalLength = 0, totalFilesLength = 0, largestContigSize = 0;
    size_t resCount = 0, unmatchedCharsAll = 0, totalMatchedAll = 0, totalDestLenAll = 0;
    std::vector<size_t> matchingLocksPos;
    std::vector<uint8_t> unmatchedFractionFactors;
    int device = 0;

    const size_t PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY = SIZE_MAX;     // MGMP.h:97

    void loadG0Ref(const std::string &refName);                                        // MGMP.cpp:66-150
    void initMatcher(const char *refStr, size_t refStrSize, size_t basicRefLength);     // :152-192
    size_t refLengthLimitFor(size_t basicRefLength, bool *bit40) const;                 // :154-168
    uint32_t windowRoundSize(uint64_t window, int gpus) const;                          // MGMP_Params::roundSize == 0
    void performMatching();                                                             // :568-606
    void processTargetsWithParallelIO();                                                // :232-313  (-t1)
    void processTargetsRounds();                                                        // :340-468 as deterministic rounds
    void verifyEmission(size_t contigs, size_t bases);                                  // --verify: the selected emission through the device decoder, exit on a contig that does not come back
    void appendHeldTargets(std::vector<std::vector<EmittedStreams>> &targets);       // streams of targets that kept a round's first pass (:382-388), in target order
    void processTargetsRoundsSharded();                                                 // the rounds with their targets sharded over the ranks of params->exchange (mgmp_sharded.cpp)
    // input stage (MGMP.cpp:7-35,349-372 on the device parser)
    mbgc_fasta_t *fasta = nullptr;
    // The files of a round in page-locked host memory, read (and inflated) by several threads. The round named as "next"
    // is read, uploaded and parsed by a thread of its own while the GPU and the host work on the current one (the
    // reference reads ahead with its IO threads too, MGMP.cpp:232-313 "WithParallelIO").
    // The thread that uploads and parses round r + 1 has the files of round r + 2 read meanwhile (two staging buffers).
    struct StagedFiles {
        uint8_t *pin = nullptr; size_t cap = 0;
        std::vector<uint64_t> fileOff;
        std::string error;
    } staged[2];
    struct Ahead { std::future<void> done; uint32_t f0 = 0, f1 = 0; RoundBatch *B = nullptr; bool active = false; } ahead;
    struct ReadAhead { std::future<void> done; uint32_t f0 = 0, f1 = 0; int slot = 0; bool active = false; } readAhead;
    void readFiles(StagedFiles &S, uint32_t f0, uint32_t f1);
    void startReadAhead(uint32_t f0, uint32_t f1, int slot, bool bothBuffers = false);                            // files [f0, f1) into staged[slot], on a thread of its own
    // read (unless read ahead) + upload + device parse, synchronous; [afterF0, afterF1) = the files to read meanwhile
    void prepareRound(uint32_t f0, uint32_t f1, RoundBatch &B, uint32_t afterF0 = 0, uint32_t afterF1 = 0);
    uint8_t *rawDev = nullptr; size_t rawCap = 0;
    uint8_t *extScratch = nullptr; size_t extScratchCap = 0;                            // extension strings that are not a span of the round's buffer
    void extensionStrings(const RoundBatch &B, const std::vector<char> &ext, const std::vector<char> &rc, uint32_t ta, uint32_t tb,
                          std::vector<const uint8_t *> &extDev, std::vector<uint64_t> &extLen);
    std::vector<mbgc_fasta_record_t> records;
    void openInputStage();
    void readG0(const std::string &path, std::vector<Contig> &out, uint64_t *fileSize);
    // files [f0, f1) parsed into B; [nextF0, nextF1) into *nextB = what will be asked for next (prepared meanwhile)
    void loadRound(uint32_t f0, uint32_t f1, RoundBatch &B, uint32_t nextF0 = 0, uint32_t nextF1 = 0, RoundBatch *nextB = nullptr);

    // hooks, MGMP.h:79-115
    virtual void initStreamsForG0Ref() = 0;
    virtual void processG0RefContig(const char *seq, size_t len) = 0;
    virtual size_t processMatches(size_t destLen, int targetIdx, size_t matchingLockPos) = 0;   // consumes the handle's last match
    virtual void processAfterSequence(uint32_t targetIdx) = 0;
    virtual void processAfterTarget(uint32_t targetIdx) = 0;
    virtual void processAfterTargetWithParallelIO(size_t matcherLoaderStartPos) = 0;
    virtual void afterTargetWithParallelIO(size_t matcherLoaderStartPos) = 0;          // the same without processAfterTarget
    virtual void initParallelProcessing() = 0;
    virtual void finalizeParallelProcessingOfTarget(uint32_t targetIdx, size_t matcherLoaderStartPos) = 0;
    // the same in two halves, for the round pipeline: what is known when the target's extension has been queued (the
    // bytes of refExtSize / locksPos, refExtLoadedPosArr), and the append of its streams once they have arrived
    virtual void noteTargetLoaded(uint32_t targetIdx, size_t matcherLoaderStartPos, size_t loadedRefLengthAfter) = 0;
    virtual void appendTargetStreams(uint32_t targetIdx) = 0;
    virtual void takeRoundStreams(uint32_t targetIdx, EmittedStreams &s) = 0;           // per-target stream append in round mode
    // the same when the round's targets arrive in target order (the round loop's collection): a contig's streams straight
    // from the handle's buffer to the collection's streams, then processAfterSequence / processAfterTarget's marks
    virtual void appendContigInOrder(const swsem_streams_t &st) = 0;
    virtual void endTargetInOrder() = 0;
    virtual const swsem_emit_params_t &emitParams() const = 0;
    virtual const std::vector<size_t> &loadedPositions() const = 0;
    virtual bool lazyMode() const = 0;

public:
    explicit MultipleGenomeMatchingProcessor(MGMP_Params *p) : params(p) {}
    // the calling thread (and the threads it starts from now on) onto the CPUs of the NUMA node the device hangs on:
    // the files travel page cache -> page-locked memory -> device, and page-locked memory is allocated next to the device
    // (MBGC_HIP_NUMA=0 leaves the threads where they are)
    static void bindHostThreadsToDeviceNode(int device);
    virtual ~MultipleGenomeMatchingProcessor();
};

class MBGC_Encoder : public MultipleGenomeMatchingProcessor {
    MBGC_Params *params;
    std::vector<size_t> refExtLoadedPosArr;                                             // MBGC_Encoder.h:48
    std::vector<EmittedStreams> targetStreams;                                          // per-target streams (round mode)
    uint32_t targetsAppended = 0;                                                       // targets whose streams arrived in order
    uint32_t sequentialTargetsDone = 0;
    size_t backendFed[MBGC_ST_COUNT] = {};                                              // bytes of every stream the backend has been handed

    void initStreamsForG0Ref() override;                                                // ENC.cpp:25-32
    void processG0RefContig(const char *seq, size_t len) override;                      // :34-37
    size_t processMatches(size_t destLen, int targetIdx, size_t matchingLockPos) override;      // :143-308 on the device
    void processAfterSequence(uint32_t targetIdx) override;                             // :489-491
    void processAfterTarget(uint32_t targetIdx) override;                               // :493-496
    void processAfterTargetWithParallelIO(size_t matcherLoaderStartPos) override;       // :498-509
    void afterTargetWithParallelIO(size_t matcherLoaderStartPos) override;
    void initParallelProcessing() override;                                             // :516-528
    void finalizeParallelProcessingOfTarget(uint32_t targetIdx, size_t matcherLoaderStartPos) override;   // :542-564
    void noteTargetLoaded(uint32_t targetIdx, size_t matcherLoaderStartPos, size_t loadedRefLengthAfter) override;   // :557-563
    void appendTargetStreams(uint32_t targetIdx) override;                              // :543-556
    void takeRoundStreams(uint32_t targetIdx, EmittedStreams &s) override;
    void appendContigInOrder(const swsem_streams_t &st) override;
    void endTargetInOrder() override;
    const swsem_emit_params_t &emitParams() const override { return params->emit; }
    const std::vector<size_t> &loadedPositions() const override { return refExtLoadedPosArr; }
    bool lazyMode() const override { return params->lazyDecompressionSupport; }

public:
    // the streams the reference enrols at ENC.cpp:779-787 (+ the two it writes beside them)
    std::string literals, mapOff, mapOff5thByte, mapLen, gapDeltas, gapMismatchesFlags, locksPosStream, refExtSizeStream;
    std::string rcMapOff, rcMapLen;                                                     // ENC.cpp:636-638 (-m3)
    size_t extensionsMatchedCharsAll = 0, extensionsMismatchesAll = 0, removedGapBreakingMatchesAll = 0;

    explicit MBGC_Encoder(MBGC_Params *p) : MultipleGenomeMatchingProcessor(p), params(p) { device = p->device; }
    void encode(const std::vector<std::string> &files);                                 // :759-807 up to performMatching()
    // prepareAndCompressStreams' collective section (:641-710) for the streams this path produces — the file-name / header /
    // line-length streams belong to the part of the tool that is not rebuilt here and go in empty (eight zero bytes each) —
    // with the caller's leaf coders; what CompressionJob::writeCompressedCollectiveParallel would write for them
    // threads: the job pool of this call; numberOfThreads: what the reference's -t would be (it only decides LZMA's numThreads; 0: = threads).
    // The header-side streams (names, sequence counts, header templates, headers, line lengths) are the CLI's and go in empty:
    // the section is what the reference's reader takes for the matcher-side streams, not a complete archive.
    std::string compressStreams(mbgc_leaf_compress_fn leaf, void *ctx, int threads, int blocksScale = 1, int numberOfThreads = 0);
    // The backend beside the matching (include/mbgc_backend.h, the incremental form): set before encode(), the encoder hands
    // over what its streams have grown by every few targets, so that the blocks of the split streams are coded while the rounds
    // go on; finishBackendStream() hands over the rest and returns the section. The literals wait for the end under -m3 (the
    // reverse-complement pass rewrites them, ENC.cpp:636-638).
    mbgc_backend_stream_t *backendStream = nullptr;
    void feedBackendStream(bool everything);
    std::string finishBackendStream(uint64_t *blocksCodedEarly);
    void backendParams(mbgc_backend_params_t &bp, int blocksScale, int numberOfThreads) const;
    size_t exactMatches() const { return resCount; }
    size_t finalReferenceLength() const { return refFinalTotalLength; }          // writeStats' refFinalTotalLength, ENC.cpp:734-743
    size_t droppedExtensionBytes() const { return matcher ? matcher->getDroppedBytes() : 0; }
    size_t maxReferenceLength() const { return matcher ? matcher->getMaxRefLength() : 0; }
    size_t unmatchedChars() const { return unmatchedCharsAll; }
};
