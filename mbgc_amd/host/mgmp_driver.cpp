#include "mgmp_driver.h"
#include "simple_sequence_matcher.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <fstream>
#include <functional>
#include <thread>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

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
    if (mode == 3) { sequentialMatching = true; rcMatchMinLength = 55; rcRedundancyRemoval = true; }   // :915-920, DEFAULT_RC_MATCH_MINIMUM_LENGTH :61
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

static double nowSeconds() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
// where the host's wall time goes (MBGC_HIP_TIMES=1 prints it with "matching finished"): reading + inflating files,
// upload + device parse, taking the streams over
static uint64_t g_retryPasses = 0, g_retryContigs = 0;          // rounds whose first pass gave contigs up, and the contigs of the targets matched again in units
static double g_tRead = 0, g_tParse = 0, g_tCollect = 0, g_tWait = 0, g_tCollectWait = 0, g_tAppend = 0, g_tAppendWait = 0, g_tPrepareSync = 0, g_tReadWait = 0, g_tMatch = 0, g_tEmit = 0, g_tFinalize = 0;

// ---------------------------------------------------------------- input stage
// mgmpInOpen / whole-file read (MGMP.cpp:7-14; gz inflate is the host's libdeflate in the reference and is not part
// of this repo) + kseq_read_lossless_fasta on the device (include/mbgc_fasta.h), status handling of
// validate_kseq_status (MGMP.cpp:16-35).
// mgmpInOpen (matching/input_with_libdeflate_wrapper.cpp:51-124): the whole file, and when it starts with the gzip magic,
// its members inflated one after the other (the reference: libdeflate_gzip_decompress_ex in a loop until the input is
// used up, output buffer sized by the ISIZE trailer and doubled when short; here: zlib on the host — DEFLATE's bit-serial
// Huffman decoding has no place on the device, and a round's files inflate on the host while the GPU matches the round
// before)
static void inflateGzip(const std::string &gz, std::string &dest) {
    const size_t at = dest.size();
    uint32_t isize;
    memcpy(&isize, gz.data() + gz.size() - 4, 4);
    size_t cap = isize ? isize : gz.size() * 4, out = 0;
    dest.resize(at + cap);
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) { fprintf(stderr, "Cannot allocate decompressor.\n"); exit(EXIT_FAILURE); }
    z.next_in = (Bytef *) gz.data();
    size_t inLeft = gz.size();
    while (true) {
        z.avail_in = (uInt) std::min<size_t>(inLeft, 1u << 30);
        const size_t inGiven = z.avail_in;
        z.next_out = (Bytef *) &dest[at + out];
        z.avail_out = (uInt) std::min<size_t>(cap - out, 1u << 30);
        const size_t outGiven = z.avail_out;
        const int res = inflate(&z, Z_NO_FLUSH);
        inLeft -= inGiven - z.avail_in;
        out += outGiven - z.avail_out;
        if (res == Z_STREAM_END) {
            if (inLeft == 0) break;
            if (inflateReset(&z) != Z_OK) { fprintf(stderr, "Error decompressing gz file: %d.\n", res); exit(EXIT_FAILURE); }   // the next member
        } else if (res == Z_OK || res == Z_BUF_ERROR) {
            if (out == cap) { cap *= 2; dest.resize(at + cap); }
            else if (inLeft == 0) { fprintf(stderr, "Error decompressing gz file: %d.\n", res); exit(EXIT_FAILURE); }            // truncated
        } else {
            fprintf(stderr, "Error decompressing gz file: %d.\n", res);
            exit(EXIT_FAILURE);
        }
    }
    inflateEnd(&z);
    dest.resize(at + out);
}

static bool isGzip(const uint8_t *p, size_t n) { return n >= 18 && p[0] == 0x1f && p[1] == 0x8b; }     // GZIP_ID1, GZIP_ID2

// mgmpInOpen (matching/input_with_libdeflate_wrapper.cpp:51-124): the whole file, and when it starts with the gzip magic,
// its members inflated one after the other (the reference: libdeflate_gzip_decompress_ex in a loop until the input is
// used up, output buffer sized by the ISIZE trailer and doubled when short; here: zlib on the host — DEFLATE's bit-serial
// Huffman decoding has no place on the device, and a round's files inflate on the host while the GPU matches the round
// before)
static bool readWholeFile(const std::string &path, std::string &dest) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamoff n = f.tellg();
    f.seekg(0);
    const size_t at = dest.size();
    dest.resize(at + (size_t) n);
    if (n) f.read(&dest[at], n);
    if (!f) return false;
    if (!isGzip((const uint8_t *) dest.data() + at, (size_t) n)) return true;
    const std::string gz(dest, at);
    dest.resize(at);
    inflateGzip(gz, dest);
    return true;
}

static void validate_kseq_status(const std::string &fileName, int status) {
    if (status == MBGC_FASTA_OK) return;
    fprintf(stderr, "Error parsing file %s", fileName.c_str());
    if (status == MBGC_FASTA_ENOTFASTA) fprintf(stderr, " - expected FASTA format.\n");
    else if (status == MBGC_FASTA_ELINES) fprintf(stderr, "\nDetected inconsistent line length in sequences.\n");
    else fprintf(stderr, " - unknown error (code %d).\n", status);
    exit(EXIT_FAILURE);
}

MultipleGenomeMatchingProcessor::~MultipleGenomeMatchingProcessor() {
    if (matcher && rawDev) matcher->devFree(rawDev);
    if (matcher && extScratch) matcher->devFree(extScratch);
    if (ahead.active) ahead.done.wait();
    if (readAhead.active) readAhead.done.wait();
    for (StagedFiles &S : staged) if (fasta && S.pin) mbgc_fasta_host_free(fasta, S.pin);
    if (fasta) mbgc_fasta_destroy(fasta);
    delete matcher;
}

void MultipleGenomeMatchingProcessor::openInputStage() {
    if (!fasta && mbgc_fasta_create(&fasta, device) != 0) {
        fprintf(stderr, "input stage: %s\n", mbgc_fasta_last_error());
        exit(EXIT_FAILURE);
    }
}

// the first file: its contigs are needed on the host (initial reference string, head of the literal stream)
void MultipleGenomeMatchingProcessor::readG0(const std::string &path, std::vector<Contig> &out, uint64_t *fileSize) {
    openInputStage();
    std::string data;
    if (!readWholeFile(path, data)) { fprintf(stderr, "cannot open file %s\n", path.c_str()); exit(EXIT_FAILURE); }
    if (fileSize) *fileSize = data.size();
    std::string seq(data.size(), '\0');
    uint64_t seqBytes = 0, nrec = 0, lineLen = 0;
    int status = 0;
    // (the record table starts small and grows to what the parser asks for: one entry per two input bytes, the bound,
    // is gigabytes of zeroed memory for a round's files)
    if (records.size() < 4096) records.resize(4096);
    int rc = mbgc_fasta_parse_host(fasta, (const uint8_t *) data.data(), data.size(), params->uppercaseDNA, (uint8_t *) &seq[0], &seqBytes,
                                   records.data(), records.size(), &nrec, &lineLen, &status);
    if (rc == -104) {
        records.resize(nrec + nrec / 4 + 16);
        rc = mbgc_fasta_parse_host(fasta, (const uint8_t *) data.data(), data.size(), params->uppercaseDNA, (uint8_t *) &seq[0], &seqBytes,
                                   records.data(), records.size(), &nrec, &lineLen, &status);
    }
    if (rc != 0) {
        fprintf(stderr, "input stage: %s\n", mbgc_fasta_last_error());
        exit(EXIT_FAILURE);
    }
    validate_kseq_status(path, status);
    out.clear();
    for (uint64_t k = 0; k < nrec; k++) {
        Contig c;
        c.header.assign(data, records[k].headerOff, records[k].headerLen);
        c.seq.assign(seq, records[k].seqOff, records[k].seqLen);
        out.push_back(std::move(c));
    }
}

// files [f0, f1) of the list, each whole (inflated when gzip), back to back in page-locked memory: the sizes first (a
// gzip file's is known once it is inflated), then every plain file read straight to its place, by up to 8 threads
void MultipleGenomeMatchingProcessor::readFiles(StagedFiles &S, uint32_t f0, uint32_t f1) {
    const double tRead0 = nowSeconds();
    S.error.clear();
    const size_t nf = f1 - f0;
    std::vector<int> fd(nf, -1);
    std::vector<uint64_t> size(nf, 0);
    std::vector<std::string> inflated(nf);
    std::vector<std::string> errors(nf);
    auto parallel = [&](const std::function<void(size_t)> &body) {
        std::atomic<size_t> next{0};
        auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < nf;) body(i); };
        std::vector<std::thread> pool;
        static const size_t readers = getenv("MBGC_HIP_READERS") ? (size_t) std::max(1, atoi(getenv("MBGC_HIP_READERS"))) : 8;
        for (size_t t = 1; t < std::min<size_t>(readers, nf); t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    };
    parallel([&](size_t i) {
        const std::string &path = fileNames[f0 + i];
        fd[i] = open(path.c_str(), O_RDONLY);
        struct stat st;
        if (fd[i] < 0 || fstat(fd[i], &st) != 0) { errors[i] = "cannot open file " + path; return; }
        size[i] = (uint64_t) st.st_size;
        uint8_t head[18];
        if (size[i] >= 18 && pread(fd[i], head, 18, 0) == 18 && isGzip(head, 18)) {
            std::string gz(size[i], '\0');
            size_t got = 0;
            while (got < gz.size()) { const ssize_t k = pread(fd[i], &gz[got], gz.size() - got, (off_t) got); if (k <= 0) break; got += (size_t) k; }
            if (got != gz.size()) { errors[i] = "Problem reading from file: " + path; return; }
            inflateGzip(gz, inflated[i]);
            size[i] = inflated[i].size();
            close(fd[i]); fd[i] = -1;
        }
    });
    for (size_t i = 0; i < nf; i++) if (!errors[i].empty()) { S.error = errors[i]; break; }
    S.fileOff.assign(1, 0);
    for (size_t i = 0; i < nf; i++) S.fileOff.push_back(S.fileOff.back() + size[i]);
    const size_t n = S.fileOff.back();
    if (S.error.empty() && n + 64 > S.cap) {                 // grow-only
        if (S.pin) mbgc_fasta_host_free(fasta, S.pin);
        S.pin = nullptr;
        S.cap = n + n / 4 + 64;
        void *p = nullptr;
        if (mbgc_fasta_host_alloc(fasta, S.cap, &p) != 0) { S.error = std::string("input stage: ") + mbgc_fasta_last_error(); S.cap = 0; }
        S.pin = (uint8_t *) p;
    }
    if (S.error.empty())
        parallel([&](size_t i) {
            uint8_t *dst = S.pin + S.fileOff[i];
            if (fd[i] < 0) { memcpy(dst, inflated[i].data(), inflated[i].size()); return; }
            size_t got = 0;
            while (got < size[i]) { const ssize_t k = pread(fd[i], dst + got, size[i] - got, (off_t) got); if (k <= 0) break; got += (size_t) k; }
            if (got != size[i]) errors[i] = "Problem reading from file: " + fileNames[f0 + i];
        });
    for (size_t i = 0; i < nf; i++) {
        if (fd[i] >= 0) close(fd[i]);
        if (S.error.empty() && !errors[i].empty()) S.error = errors[i];
    }
    g_tRead += nowSeconds() - tRead0;                        // (one reader at a time)
}

static bool readBesideUpload() {
    // MBGC_HIP_READ_BESIDE=0: a round's files are read by the thread that then uploads and parses them, nothing beside it
    static const bool on = !getenv("MBGC_HIP_READ_BESIDE") || atoi(getenv("MBGC_HIP_READ_BESIDE"));
    return on;
}

void MultipleGenomeMatchingProcessor::startReadAhead(uint32_t f0, uint32_t f1, int slot, bool bothBuffers) {
    openInputStage();
    readAhead.f0 = f0; readAhead.f1 = f1; readAhead.slot = slot; readAhead.active = true;
    StagedFiles *S = &staged[slot], *O = &staged[1 - slot];
    readAhead.done = std::async(std::launch::async, [this, S, O, f0, f1, bothBuffers] {
        const double ta = nowSeconds();
        readFiles(*S, f0, f1);
        const double tb = nowSeconds();
        // (page-locking a round's worth of memory takes tens of milliseconds and holds up the uploads queued meanwhile: the
        // second staging buffer is made here, beside the matcher's construction, not beside the first round)
        if (bothBuffers && S->error.empty() && O->cap < S->cap) {
            void *p = nullptr;
            if (mbgc_fasta_host_alloc(fasta, S->cap, &p) == 0) { O->pin = (uint8_t *) p; O->cap = S->cap; }
        }
        if (getenv("MBGC_HIP_TIMES") && f0 <= 1)
            fprintf(stderr, "  first read-ahead: files %u..%u read in %.0f ms (their page-locked buffer included), the second buffer in %.0f ms\n", f0, f1, (tb - ta) * 1e3, (nowSeconds() - tb) * 1e3);
    });
}

void MultipleGenomeMatchingProcessor::bindHostThreadsToDeviceNode(int device) {
    if (getenv("MBGC_HIP_NUMA") && !atoi(getenv("MBGC_HIP_NUMA"))) return;
    const int node = swsem_device_numa_node(device);
    if (node < 0) return;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) { fclose(f); return; }
    int a, b, any = 0;
    while (fscanf(f, "%d", &a) == 1) {                       // "0-63,128-191"
        b = a;
        int c = fgetc(f);
        if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
        for (int k = a; k <= b && k < CPU_SETSIZE; k++) if (CPU_ISSET(k, &allowed)) { CPU_SET(k, &want); any++; }
        if (c != ',') break;
    }
    fclose(f);
    if (any >= 4) sched_setaffinity(0, sizeof want, &want);  // (never down to a handful of CPUs: the readers need theirs)
}

void MultipleGenomeMatchingProcessor::loadRound(uint32_t f0, uint32_t f1, RoundBatch &B, uint32_t nextF0, uint32_t nextF1, RoundBatch *nextB) {
    const double t0 = nowSeconds();
    bool have = false;
    if (ahead.active) {
        ahead.done.wait();
        ahead.active = false;
        have = ahead.f0 == f0 && ahead.f1 == f1 && ahead.B == &B;
    }
    g_tWait += nowSeconds() - t0;
    if (!have) { const double tp0 = nowSeconds(); prepareRound(f0, f1, B, nextF0, nextF1); g_tPrepareSync += nowSeconds() - tp0; }
    if (nextB && nextF1 > nextF0) {
        // the round after the next one: the callers' rounds are ranges of equal length at equal distance (a wrong guess
        // costs a read that nobody takes)
        const uint64_t stride = nextF0 - f0, count = nextF1 - nextF0;
        const uint32_t afterF0 = (uint32_t) std::min<uint64_t>(filesCount, nextF0 + stride);
        const uint32_t afterF1 = (uint32_t) std::min<uint64_t>(filesCount, nextF0 + stride + count);
        ahead.f0 = nextF0; ahead.f1 = nextF1; ahead.B = nextB; ahead.active = true;
        ahead.done = std::async(std::launch::async, [this, nextF0, nextF1, nextB, afterF0, afterF1] { prepareRound(nextF0, nextF1, *nextB, afterF0, afterF1); });
    }
}

// files [f0, f1) of the list -> their contigs back to back in B.seqDev; contig c belongs to target targetBase + (file - f0)
void MultipleGenomeMatchingProcessor::prepareRound(uint32_t f0, uint32_t f1, RoundBatch &B, uint32_t afterF0, uint32_t afterF1) {
    openInputStage();
    const int nf = (int) (f1 - f0);
    if (nf <= 0) {                                           // (a rank without targets in a short last round)
        B.offsets.assign(1, 0); B.targetOf.clear(); B.bytes = 0;
        return;
    }
    const double tWait0 = nowSeconds();
    StagedFiles *S = nullptr;
    if (readAhead.active) {
        readAhead.done.wait();
        readAhead.active = false;
        if (readAhead.f0 == f0 && readAhead.f1 >= f1) S = &staged[readAhead.slot];             // (the first read-ahead works from a guessed round size: a file or two too many are read again later)
    }
    g_tReadWait += nowSeconds() - tWait0;
    if (!S) { S = &staged[staged[1].cap > staged[0].cap ? 1 : 0]; readFiles(*S, f0, f1); }   // (the buffer that has been allocated)
    if (!S->error.empty()) { fprintf(stderr, "%s\n", S->error.c_str()); exit(EXIT_FAILURE); }
    if (readBesideUpload() && afterF1 > afterF0 && afterF0 >= f1) {  // the files after these, into the other staging buffer, meanwhile
        startReadAhead(afterF0, afterF1, S == &staged[0] ? 1 : 0);
    }
    const std::vector<uint64_t> &fileOff = S->fileOff;
    const double tParse0 = nowSeconds();
    const size_t n = fileOff[nf];                            // (the staged files may be more than this round's)
    totalFilesLength += n;
    if (n + 64 > rawCap) {                                   // grow-only: freeing device memory waits for the whole device
        if (rawDev) matcher->devFree(rawDev);
        rawCap = n + n / 4 + 64;
        rawDev = matcher->devAlloc(rawCap);
    }
    if (n + 64 > B.seqCap) {
        if (B.seqDev) matcher->devFree(B.seqDev);
        B.seqCap = n + n / 4 + 64;
        B.seqDev = matcher->devAlloc(B.seqCap);
    }
    if (mbgc_fasta_upload(fasta, rawDev, S->pin, n) != 0) { fprintf(stderr, "input stage: %s\n", mbgc_fasta_last_error()); exit(EXIT_FAILURE); }
    std::vector<uint64_t> seqBase(nf + 1), recBase(nf + 1), lineLen(nf);
    std::vector<int> status(nf);
    if (records.size() < 4096) records.resize(4096);
    int rc = mbgc_fasta_parse_batch_dev(fasta, rawDev, fileOff.data(), nf, params->uppercaseDNA, B.seqDev, B.seqCap, seqBase.data(),
                                        records.data(), records.size(), recBase.data(), lineLen.data(), status.data());
    if (rc == -104) {                                        // the table was too small: the parser said how many records there are
        records.resize(recBase[nf] + recBase[nf] / 4 + 16);
        rc = mbgc_fasta_parse_batch_dev(fasta, rawDev, fileOff.data(), nf, params->uppercaseDNA, B.seqDev, B.seqCap, seqBase.data(),
                                        records.data(), records.size(), recBase.data(), lineLen.data(), status.data());
    }
    if (rc != 0) {
        fprintf(stderr, "input stage: %s\n", mbgc_fasta_last_error());
        exit(EXIT_FAILURE);
    }
    B.offsets.assign(1, 0);
    B.targetOf.clear();
    for (int f = 0; f < nf; f++) {
        validate_kseq_status(fileNames[f0 + f], status[f]);
        for (uint64_t k = recBase[f]; k < recBase[f + 1]; k++) {
            // (the contigs of a file, and the files, follow each other without gaps: seqBase[f] + seqOff == the running offset)
            largestContigSize = std::max<uint64_t>(largestContigSize, records[k].seqLen);
            B.offsets.push_back(seqBase[f] + records[k].seqOff + records[k].seqLen);
            B.targetOf.push_back((uint32_t) f);
        }
    }
    B.bytes = seqBase[nf];
    g_tParse += nowSeconds() - tParse0;
}

// ---------------------------------------------------------------- MultipleGenomeMatchingProcessor

size_t MultipleGenomeMatchingProcessor::refLengthLimitFor(size_t basicRefLength, bool *bit40) const {
    size_t refLengthLimit = (size_t) params->referenceFactor * basicRefLength;                 // MGMP.cpp:154
    refLengthLimit *= 2;                                                                       // :157-158 (RC kept in the same buffer)
    bool b40 = params->enable40bitReference;
    if (refLengthLimit <= UINT32_MAX) b40 = false;                                             // :159-160
    else if (!b40) refLengthLimit = UINT32_MAX;
    else if (refLengthLimit > MGMP_Params::REFERENCE_LENGTH_LIMIT) refLengthLimit = MGMP_Params::REFERENCE_LENGTH_LIMIT;
    if (refLengthLimit > UINT32_MAX)
        refLengthLimit = UINT32_MAX + (refLengthLimit - UINT32_MAX) / params->bigReferenceCompressorRatio;   // :165-166
    if (bit40) *bit40 = b40;
    return refLengthLimit;
}

// MGMP_Params::roundSize == 0: the targets that may hold their lock positions together. Sized by the largest target file (an
// upper bound of the bytes a target adds to the reference without its reverse complement; a gzip file is taken for five times
// its size) — a round whose extensions outgrow the window anyway (reverse complements of dissimilar contigs) loses the excess
// as the reference's workers would, and the tool reports the bytes.
uint32_t MultipleGenomeMatchingProcessor::windowRoundSize(uint64_t window, int gpus) const {
    uint64_t largest = 0;
    for (uint32_t f = 1; f < filesCount; f++) {
        struct stat st;
        if (stat(fileNames[f].c_str(), &st) != 0) continue;
        const std::string &n = fileNames[f];
        const bool gz = n.size() > 3 && n.compare(n.size() - 3, 3, ".gz") == 0;
        largest = std::max<uint64_t>(largest, (uint64_t) st.st_size * (gz ? 5 : 1));
    }
    uint64_t inFlight = MGMP_Params::MAX_TARGETS_IN_FLIGHT;
    if (window && largest) inFlight = std::min<uint64_t>(inFlight, std::max<uint64_t>(1, window / (largest + 1)));
    return (uint32_t) std::max<uint64_t>(1, inFlight / (uint64_t) std::max(1, gpus));
}

void MultipleGenomeMatchingProcessor::initMatcher(const char *refStr, size_t refStrSize, size_t basicRefLength) {
    size_t refLengthLimit = refLengthLimitFor(basicRefLength, &params->enable40bitReference);
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
    readG0(refName, contigs, &fileSize);
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

// The sequential schedule (MGMP.cpp:232-313): every contig is matched against a reference that already holds the one
// before it, so the contigs run one after the other — but only up to what the next one needs: a contig's match-finding
// and processMatches' first pass (its return value decides the loadRef), then the loadRef; the stream bytes of contig c
// are produced beside contig c + 1 (two emissions in flight) and appended, in order, one contig later.
void MultipleGenomeMatchingProcessor::verifyEmission(size_t contigs, size_t bases) {
    int firstBad = -1;
    uint64_t firstDiff = 0;
    const int nbad = matcher->emitVerify(&firstBad, &firstDiff);
    if (nbad) {
        fprintf(stderr, "verify: %d of %zu contigs do not decode back to their bytes (first: contig %d of the batch, byte %llu; %llu contigs verified before)\n",
                nbad, contigs, firstBad, (unsigned long long) firstDiff, (unsigned long long) params->verifiedContigs);
        exit(EXIT_FAILURE);
    }
    params->verifiedContigs += contigs; params->verifiedBases += bases;
}

void MultipleGenomeMatchingProcessor::processTargetsWithParallelIO() {
    RoundBatch three[3];                   // the file being matched, the file arriving, the file whose last contig's emission still reads its bytes
    struct { bool valid = false; int fileSeps = 0; uint32_t buf = 0; } prev;                   // the contig whose streams are still to be taken
    // the appends run on a thread of their own, one set at a time (a divergent contig's streams are megabytes: 0.2 ms of memcpy
    // between a contig's first pass and its loadRef otherwise); the views stay valid meanwhile — every emission slot keeps two
    // host buffers in turn (include/mbgc_swsem.h)
    std::future<void> appending;
    auto appendsDone = [&] { if (appending.valid()) appending.get(); };
    auto collectPrev = [&](bool newerBegun) {
        appendsDone();
        swsem_streams_t st = {};
        const bool has = prev.valid;
        if (prev.valid) {
            if (newerBegun) matcher->emitSelect(true);
            matcher->emitView(0, st);
            if (newerBegun) matcher->emitSelect(false);
        }
        const int seps = prev.fileSeps;
        appending = std::async(std::launch::async, [this, st, has, seps] {
            if (has) appendContigInOrder(st);                                                   // processMatches' appends + processAfterSequence (:276, :288)
            for (int k = 0; k < seps; k++) endTargetInOrder();                                  // processAfterTarget (:306)
        });
        prev.valid = false; prev.fileSeps = 0;
    };
    const std::vector<uint64_t> noLock(1, UINT64_MAX);                                          // :274 (no lock in this mode)
    for (uint32_t i = 0; i < filesCount; i++) {
        RoundBatch &B = three[i % 3];
        if (prev.valid && prev.buf == (i + 1) % 3) { matcher->emitEnd(); collectPrev(false); }  // (files without records in between: the arriving file would overwrite what that emission reads)
        loadRound(i, i + 1, B, i + 1, std::min(filesCount, i + 2), &three[(i + 1) % 3]);        // MGMP.cpp:247-250
        const size_t startPos = matcher->getLoadedRefLength();                                 // :251
        unmatchedFractionFactors.push_back(params->currentUnmatchedFractionFactor < 256 ? params->currentUnmatchedFractionFactor : 0);
        unmatchedFractionFactors.push_back((uint8_t) params->unmatchedFractionRCFactor);
        for (size_t c = 0; c + 1 < B.offsets.size(); c++) {
            const size_t bSize = B.offsets[c + 1] - B.offsets[c];
            const uint8_t *seq = B.seqDev + B.offsets[c];
            std::vector<uint64_t> un, counts;
            matcher->matchRoundBegin(seq, {0, bSize}, params->k, {});
            matcher->emitRoundBegin(emitParams(), noLock, {(int) unmatchedFractionFactors[0]}, {processedTargetsCount}, {0}, loadedPositions(),
                                    nullptr, un, counts);                                        // :276 (targetIdx 0 in this mode, ENC.cpp:202)
            resCount += counts[0];
            totalDestLenAll += bSize;
            if (params->verifyEmissions && i % (uint32_t) std::max(1, params->verifyEvery) == 0) verifyEmission(1, bSize);
            collectPrev(true);
            const size_t currentUnmatched = un[0];
            const bool loadContigToRef = params->isContigProperForRefExtension(bSize, currentUnmatched, params->currentUnmatchedFractionFactor);
            const bool loadContigRCToRef = params->rcInReference &&
                                           params->isContigProperForRefRCExtension(bSize, currentUnmatched, params->unmatchedFractionRCFactor);
            // :281-287: the contig, or the (always empty in release builds) literal extension
            matcher->loadRefDev(seq, loadContigToRef ? bSize : 0, loadContigRCToRef, params->refRegionSeparators, REF_REGION_SEPARATOR);
            prev.valid = true; prev.buf = i % 3;
        }
        prev.fileSeps++;                                                                        // (behind the file's last contig)
        afterTargetWithParallelIO(startPos);                                                    // :306 without processAfterTarget
    }
    if (prev.valid) matcher->emitEnd();
    collectPrev(false);
    appendsDone();
    if (ahead.active) { ahead.done.wait(); ahead.active = false; }
    for (auto &B : three) if (B.seqDev) matcher->devFree(B.seqDev);
}

// The extension strings of the targets [ta, tb) of a round (indices inside the round), :389-398: a target's contigs (and
// reverse complements) that extend the reference, in order. A target loaded whole and without reverse complements is the
// span of its contigs in the round's buffer; the others are put together in a grow-only scratch buffer.
void MultipleGenomeMatchingProcessor::extensionStrings(const RoundBatch &B, const std::vector<char> &ext, const std::vector<char> &rc,
                                                       uint32_t ta, uint32_t tb, std::vector<const uint8_t *> &extDev, std::vector<uint64_t> &extLen) {
    const size_t ncont = B.targetOf.size();
    const uint32_t T = tb - ta;
    extDev.assign(T, nullptr); extLen.assign(T, 0);
    std::vector<char> whole(T, 1), has(T, 0);
    std::vector<uint64_t> beg(T, 0), end(T, 0);
    size_t need = 0;
    for (size_t c = 0; c < ncont; c++) {
        const uint32_t t = B.targetOf[c];
        if (t < ta || t >= tb) continue;
        if (!has[t - ta]) { beg[t - ta] = B.offsets[c]; has[t - ta] = 1; }
        end[t - ta] = B.offsets[c + 1];
        if (!ext[c] || rc[c]) whole[t - ta] = 0;
    }
    for (size_t c = 0; c < ncont; c++) {
        const uint32_t t = B.targetOf[c];
        if (t >= ta && t < tb && !whole[t - ta]) need += ((ext[c] ? 1 : 0) + (rc[c] ? 1 : 0)) * (B.offsets[c + 1] - B.offsets[c]);
    }
    if (need > extScratchCap) {                              // grow-only: freeing device memory waits for the whole device
        if (extScratch) matcher->devFree(extScratch);
        extScratchCap = need + need / 2 + 64;
        extScratch = matcher->devAlloc(extScratchCap);
    }
    size_t at = 0, c = 0;
    for (uint32_t t = ta; t < tb; t++) {
        const uint32_t k = t - ta;
        if (!has[k]) continue;
        if (whole[k]) { if (end[k] > beg[k]) { extDev[k] = B.seqDev + beg[k]; extLen[k] = end[k] - beg[k]; } continue; }
        const size_t start = at;
        for (c = 0; c < ncont; c++)
            if (B.targetOf[c] == t) {
                const size_t len = B.offsets[c + 1] - B.offsets[c];
                if (ext[c]) { matcher->devCopy(extScratch + at, B.seqDev + B.offsets[c], len); at += len; }
                if (rc[c]) { matcher->devRevComp(B.seqDev + B.offsets[c], len, extScratch + at); at += len; }    // :393-398
            }
        if (at > start) { extDev[k] = extScratch + start; extLen[k] = at - start; }
    }
}

// streams the host holds already, target by target (targets of a round that kept what its first pass found while others
// were matched again, processTargetsRounds): contig by contig, then the target separator — the targets arrive in order, so the
// bytes go straight to where appendTargetStreams (ENC.cpp:543-556) would put them
void MultipleGenomeMatchingProcessor::appendHeldTargets(std::vector<std::vector<EmittedStreams>> &targets) {
    for (auto &contigs : targets) {
        for (EmittedStreams &e : contigs) {
            swsem_streams_t view = {};
            for (int i = 0; i < SWSEM_NSTREAMS; i++) { view.data[i] = (const uint8_t *) e.s[i].data(); view.size[i] = e.s[i].size(); }
            view.unmatchedChars = e.unmatchedChars; view.extensionsMatchedChars = e.extensionsMatchedChars;
            view.extensionsMismatches = e.extensionsMismatches; view.totalMatched = e.totalMatched;
            view.removedGapBreakingMatches = e.removedGapBreakingMatches; view.nmatches = e.nmatches;
            appendContigInOrder(view);
            e = EmittedStreams();
        }
        endTargetInOrder();
    }
}

// processTarget + finalizeParallelProcessingOfTarget (MGMP.cpp:340-468) as deterministic rounds, pipelined the way the
// reference's workers and finalizer overlap (:520-555): a round's contigs are matched and processMatches' first pass
// returns what the extension policy needs; the round's loadRefs are queued at once — speculatively behind that first
// pass when the last round's decisions were unanimous (swsem_emit_batch_begin_spec) — and the next round is started
// while the stream bytes of this one are still being produced on the second stream; they are taken a round later.
void MultipleGenomeMatchingProcessor::processTargetsRounds() {
    initParallelProcessing();
    matchingLocksPos.assign(targetsCount, SIZE_MAX);
    unmatchedFractionFactors.assign(2 * (size_t) targetsCount, 0);
    const uint32_t R = (uint32_t) std::max(1, params->roundSize);
    const uint32_t nRounds = (targetsCount + R - 1) / R;
    const bool bench = params->benchMode;
    // round r lives in slot r % 3: its bytes are read by the emission's second phase while round r + 1 is matched, and
    // round r + 2 may already be arriving (bench: every round resident before the clock starts)
    std::vector<RoundBatch> slots(bench ? nRounds : 3);
    if (bench) matcher->setEmitHostCopy(false);
    if (bench)
        for (uint32_t r = 0; r < nRounds; r++) {
            slots[r].t0 = r * R; slots[r].t1 = std::min(targetsCount, (r + 1) * R);
            loadRound(1 + slots[r].t0, 1 + slots[r].t1, slots[r], 1 + std::min(targetsCount, (r + 1) * R), 1 + std::min(targetsCount, (r + 2) * R),
                      r + 1 < nRounds ? &slots[r + 1] : nullptr);
        }
    // an emission whose streams have not been taken yet: targets [u0, u1) of batch B, its contigs [c0, c1). `tail`: targets that
    // follow it and whose streams the host holds already (they kept what the batch's first pass found, see below) — appended
    // behind the unit's, target by target
    typedef std::vector<std::vector<EmittedStreams>> HeldTargets;
    struct Deferred { bool valid = false; RoundBatch *B = nullptr; uint32_t u0 = 0, u1 = 0; size_t c0 = 0, c1 = 0; std::shared_ptr<HeldTargets> tail; } prev;
    int predicted = -1;                                                     // what every contig of the last round decided (-1: no prediction)
    // per-target stream merge, ENC.cpp:542-556: the views of a round's streams (page-locked memory of the emission slot)
    // are appended to the collection's strings by a thread of its own, while the next round is matched — the slot's
    // memory belongs to the emission after the next one, so the appends are waited for before that one begins
    std::future<void> appending;
    auto appendsDone = [&] { if (appending.valid()) appending.get(); };
    auto collect = [&](const Deferred &D, bool newerBegun) {
        RoundBatch &B = *D.B;
        const double tc0 = nowSeconds();
        appendsDone();                                                      // (one set of appends outstanding at a time)
        const double tc1 = nowSeconds();
        g_tAppendWait += tc1 - tc0;
        if (newerBegun) matcher->emitSelect(true);
        if (!bench && D.c1 > D.c0) { swsem_streams_t st = {}; matcher->emitView(0, st); }   // (waits for the emission and its copy)
        g_tCollectWait += nowSeconds() - tc1;
        auto views = std::make_shared<std::vector<swsem_streams_t>>();
        auto perTarget = std::make_shared<std::vector<uint32_t>>();
        size_t c = D.c0;                                                    // (the contigs of a round are in target order)
        for (uint32_t u = D.u0; u < D.u1; u++) {
            uint32_t k = 0;
            for (; c < D.c1 && B.targetOf[c] == u; c++) {
                if (bench) continue;                                        // bench: the bytes stay packed in HBM, as in bench.py
                swsem_streams_t st = {};
                matcher->emitView((int) (c - D.c0), st);
                views->push_back(st);
                k++;
            }
            perTarget->push_back(k);
        }
        if (newerBegun) matcher->emitSelect(false);
        std::shared_ptr<HeldTargets> tail = D.tail;
        if (!bench)
            appending = std::async(std::launch::async, [this, views, perTarget, tail] {
                const double ta0 = nowSeconds();
                size_t v = 0;
                for (uint32_t k : *perTarget) {
                    for (uint32_t i = 0; i < k; i++) appendContigInOrder((*views)[v++]);
                    endTargetInOrder();
                }
                if (tail) appendHeldTargets(*tail);
                g_tAppend += nowSeconds() - ta0;
            });
        g_tCollect += nowSeconds() - tc0;
    };
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    double tStart = 0;
    const char *perRound = getenv("MBGC_HIP_TIMES");
    double tRound = nowSeconds();
    for (uint32_t r = 0; r < nRounds; r++) {
        if (perRound && perRound[0] == '2') {
            const double t = nowSeconds();
            if (r) fprintf(stderr, "  round %u: %.1f ms\n", r - 1, (t - tRound) * 1e3);
            tRound = t;
        }
        RoundBatch &B = slots[bench ? r : r % 3];
        if (!bench) {
            B.t0 = r * R; B.t1 = std::min(targetsCount, (r + 1) * R);
            loadRound(1 + B.t0, 1 + B.t1, B, 1 + std::min(targetsCount, (r + 1) * R), 1 + std::min(targetsCount, (r + 2) * R),
                      r + 1 < nRounds ? &slots[(r + 1) % 3] : nullptr);
        } else if ((int) r == params->benchWarmup) {
            if (prev.valid) { matcher->emitEnd(); collect(prev, false); prev.valid = false; }
            matcher->synchronize();
            if (getenv("MBGC_HIP_PROFILE")) swsem_profile_enable(matcher->handle(), 1);   // (the kernel profile covers what the clock covers)
            tStart = now();
        }
        const size_t ncont = B.targetOf.size();
        const uint32_t T = B.t1 - B.t0;
        for (uint32_t t = B.t0; t < B.t1; t++) {
            unmatchedFractionFactors[2 * t] = params->currentUnmatchedFractionFactor < 256 ? params->currentUnmatchedFractionFactor : 0;   // :351-352
            unmatchedFractionFactors[2 * t + 1] = (uint8_t) params->unmatchedFractionRCFactor;
            matchingLocksPos[t] = matcher->acquireWorkerMatchingLockPos();                      // :353-358
        }
        if (ncont == 0) {                                                                       // files without a record
            if (prev.valid) { matcher->emitEnd(); collect(prev, false); prev.valid = false; }
            appendsDone();
            for (uint32_t t = B.t0; t < B.t1; t++) {
                const size_t startPos = matcher->getLoadedRefLength();
                processAfterTarget(t);
                finalizeParallelProcessingOfTarget(t, startPos);
                matcher->releaseWorkerMatchingLockPos(matchingLocksPos[t]);
                processedTargetsCount = t + 1;
            }
            continue;
        }
        // The batch as one unit first — every target's worker against the reference as the round found it. If that first pass
        // gives contigs up as dissimilar (:382-388), the targets that hold one are void, whole (a worker that starts its target
        // again when its turn has come); the others keep what was found. The finalizer then takes the targets in order: a run of
        // kept targets is loaded from what the first pass found; the stopped targets that follow each other — at most
        // allowedTargetsOutrunForDissimilarContigs + 1 of them, a unit — are matched again with every target in front of the unit
        // loaded (processMatches then gives nothing up, ENC.cpp:203) — and a unit is a round of the pipeline below like any
        // other: its loads queued behind its first pass, its streams taken while the next unit is matched. (Until round 4 a
        // stopped target was matched again in blocking calls, and every later target with it: tests/_driver.py tells that story.)
        const uint32_t unitTargets = (uint32_t) std::max(0, emitParams().allowedTargetsOutrunForDissimilarContigs) + 1;
        std::vector<uint64_t> un, counts;
        // targets [u0, u1) of the batch as one round of the pipeline; false: its first pass gave a contig up (nothing was loaded)
        auto runUnit = [&](uint32_t u0, uint32_t u1) -> bool {
            const uint32_t Tu = u1 - u0;
            size_t c0 = 0, c1;
            while (c0 < ncont && B.targetOf[c0] < u0) c0++;
            for (c1 = c0; c1 < ncont && B.targetOf[c1] < u1; c1++) {}
            const size_t nc = c1 - c0;
            std::vector<uint64_t> locks(nc), tlocks(Tu), offs(nc + 1);
            std::vector<int> factors(nc);
            std::vector<int64_t> processed(nc, processedTargetsCount), tidx(nc);
            for (size_t c = c0; c < c1; c++) {
                const uint32_t t = B.t0 + B.targetOf[c];
                locks[c - c0] = matchingLocksPos[t]; factors[c - c0] = unmatchedFractionFactors[2 * t]; tidx[c - c0] = t;
                offs[c - c0] = B.offsets[c] - B.offsets[c0];
            }
            offs[nc] = B.offsets[c1] - B.offsets[c0];
            for (uint32_t t = 0; t < Tu; t++) tlocks[t] = matchingLocksPos[B.t0 + u0 + t];
            // span of every target's contigs in the buffer (they follow each other)
            std::vector<uint64_t> tBeg(Tu, 0), tEnd(Tu, 0);
            std::vector<char> tHas(Tu, 0);
            for (size_t c = c0; c < c1; c++) {
                const uint32_t lt = B.targetOf[c] - u0;
                if (!tHas[lt]) { tBeg[lt] = B.offsets[c]; tHas[lt] = 1; }
                tEnd[lt] = B.offsets[c + 1];
            }
            const double tm0 = nowSeconds();
            matcher->matchRoundBegin(B.seqDev + B.offsets[c0], offs, params->k, locks);         // :379
            g_tMatch += nowSeconds() - tm0;
            // the unit's finalize under the prediction "every contig decides as the last round's did"
            std::vector<const uint8_t *> extDev(Tu, nullptr);
            std::vector<uint64_t> extLen(Tu, 0), loadedAfter(Tu, 0);
            std::vector<uint8_t> predExt(nc, predicted == 1), predRC(nc, 0);
            swsem_spec_finalize_t spec = {};
            const bool useSpec = predicted >= 0 && !params->verifyEmissions;   // (verifying: the round's loads wait for the check)
            if (useSpec) {
                if (predicted == 1)
                    for (uint32_t t = 0; t < Tu; t++)
                        if (tHas[t] && tEnd[t] > tBeg[t]) { extDev[t] = B.seqDev + tBeg[t]; extLen[t] = tEnd[t] - tBeg[t]; }
                spec.ntargets = (int) Tu; spec.ext_dev = extDev.data(); spec.ext_len = extLen.data();
                spec.addSep = params->refRegionSeparators; spec.sep = REF_REGION_SEPARATOR; spec.lazySeparator = lazyMode();
                spec.lockPos = tlocks.data(); spec.loadedAfter = loadedAfter.data();
                spec.predExt = predExt.data(); spec.predRC = predRC.data();
                spec.factor = params->currentUnmatchedFractionFactor; spec.rcFactor = params->rcInReference ? params->unmatchedFractionRCFactor : 0;
            }
            const size_t before = matcher->getLoadedRefLength();
            // (the appends of the emission before the last one may still be running: their views point into the host buffer this
            // slot used LAST time, and the library gives every slot two of them in turn — include/mbgc_swsem.h; collect() below
            // waits for them before it queues the next ones, so at most one set of appends is ever outstanding)
            const double te0 = nowSeconds();
            const bool applied = matcher->emitRoundBegin(emitParams(), locks, factors, processed, tidx, loadedPositions(),
                                                         useSpec ? &spec : nullptr, un, counts);   // :381
            g_tEmit += nowSeconds() - te0;
            for (size_t c = 0; c < nc; c++)
                if (un[c] == PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY) return false;   // (nothing was applied: the library checks the same condition)
            for (size_t c = 0; c < nc; c++) resCount += counts[c];
            if (params->verifyEmissions && (Tu != T || (B.t0 / std::max<uint32_t>(1, T)) % (uint32_t) std::max(1, params->verifyEvery) == 0))
                verifyEmission(nc, offs[nc]);
            // extension policy, :389-398
            std::vector<char> ext(ncont, 0), rc(ncont, 0);
            bool allExt = true, noneExt = true, anyRC = false;
            for (size_t c = c0; c < c1; c++) {
                const uint32_t t = B.t0 + B.targetOf[c];
                const size_t len = B.offsets[c + 1] - B.offsets[c];
                ext[c] = params->isContigProperForRefExtension(len, un[c - c0], unmatchedFractionFactors[2 * t]);
                rc[c] = params->rcInReference && params->isContigProperForRefRCExtension(len, un[c - c0], params->unmatchedFractionRCFactor);
                allExt &= (bool) ext[c]; noneExt &= !ext[c]; anyRC |= (bool) rc[c];
            }
            if (!applied) {
                extensionStrings(B, ext, rc, u0, u1, extDev, extLen);
                const double tf0 = nowSeconds();
                matcher->finalizeTargets(extDev, extLen, params->refRegionSeparators, REF_REGION_SEPARATOR, lazyMode(), tlocks, loadedAfter);   // :440-457
                g_tFinalize += nowSeconds() - tf0;
            }
            size_t startPos = before;
            for (uint32_t t = 0; t < Tu; t++) {
                noteTargetLoaded(B.t0 + u0 + t, startPos, loadedAfter[t]);                      // ENC.cpp:557-563
                startPos = loadedAfter[t];
            }
            processedTargetsCount = B.t0 + u1;
            // the previous unit's streams: its second phase ran beside everything above
            if (prev.valid) collect(prev, true);
            prev = Deferred();
            prev.valid = true; prev.B = &B; prev.u0 = u0; prev.u1 = u1; prev.c0 = c0; prev.c1 = c1;
            predicted = (!anyRC && (allExt || noneExt)) ? (allExt ? 1 : 0) : -1;
            return true;
        };
        if (runUnit(0, T)) continue;
        // ---- the first pass gave contigs up
        std::vector<char> stopped(T, 0);
        for (size_t c = 0; c < ncont; c++)
            if (un[c] == PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY) stopped[B.targetOf[c]] = 1;
        if (prev.valid) { collect(prev, true); prev.valid = false; }
        appendsDone();
        matcher->emitEnd();
        predicted = -1;
        if (params->verifyEmissions) verifyEmission(ncont, B.offsets[ncont] - B.offsets[0]);
        // what the kept targets' workers found: decisions now, streams to the host (the units take the emission's buffers over)
        const std::vector<uint64_t> un1 = un, counts1 = counts;
        std::vector<EmittedStreams> held(ncont);
        g_retryPasses++;
        for (size_t c = 0; c < ncont; c++) {
            if (stopped[B.targetOf[c]]) { g_retryContigs++; continue; }
            resCount += counts1[c];
            if (!bench) matcher->emitTake((int) c, held[c]);
        }
        uint32_t t = 0;
        while (t < T) {
            if (stopped[t]) {
                uint32_t u1 = t + 1;
                while (u1 < T && u1 - t < unitTargets && stopped[u1]) u1++;
                if (!runUnit(t, u1)) {
                    fprintf(stderr, "internal error: a contig was given up although every target in front of its unit had been loaded\n");
                    exit(EXIT_FAILURE);
                }
                t = u1;
                continue;
            }
            uint32_t t2 = t + 1;
            while (t2 < T && !stopped[t2]) t2++;
            // targets [t, t2) keep the first pass: their extensions in one call (:433-468), their streams behind the unit in flight
            std::vector<char> ext(ncont, 0), rc(ncont, 0);
            auto tail = std::make_shared<HeldTargets>(t2 - t);
            for (size_t c = 0; c < ncont; c++) {
                const uint32_t lt = B.targetOf[c];
                if (lt < t || lt >= t2) continue;
                const size_t len = B.offsets[c + 1] - B.offsets[c];
                ext[c] = params->isContigProperForRefExtension(len, un1[c], unmatchedFractionFactors[2 * (B.t0 + lt)]);         // :389-392
                rc[c] = params->rcInReference && params->isContigProperForRefRCExtension(len, un1[c], params->unmatchedFractionRCFactor);
                (*tail)[lt - t].push_back(std::move(held[c]));
            }
            std::vector<const uint8_t *> extDev;
            std::vector<uint64_t> extLen, loadedAfter, tlocks;
            extensionStrings(B, ext, rc, t, t2, extDev, extLen);
            for (uint32_t x = t; x < t2; x++) tlocks.push_back(matchingLocksPos[B.t0 + x]);
            size_t startPos = matcher->getLoadedRefLength();
            const double tf0 = nowSeconds();
            matcher->finalizeTargets(extDev, extLen, params->refRegionSeparators, REF_REGION_SEPARATOR, lazyMode(), tlocks, loadedAfter);   // :440-457
            g_tFinalize += nowSeconds() - tf0;
            for (uint32_t x = t; x < t2; x++) {
                noteTargetLoaded(B.t0 + x, startPos, loadedAfter[x - t]);                       // ENC.cpp:557-563
                startPos = loadedAfter[x - t];
            }
            processedTargetsCount = B.t0 + t2;
            if (bench) {}                                                   // (bench: nobody takes the bytes)
            else if (prev.valid) prev.tail = tail;                          // (a unit is never followed by two kept runs)
            else { appendsDone(); appendHeldTargets(*tail); }
            t = t2;
        }
    }
    if (prev.valid) { matcher->emitEnd(); collect(prev, false); }
    appendsDone();
    matcher->synchronize();
    if (bench) {
        params->benchSeconds = now() - tStart;
        params->benchRounds = (int) nRounds - params->benchWarmup;
        params->benchBases = 0;
        for (uint32_t r = (uint32_t) params->benchWarmup; r < nRounds; r++) params->benchBases += slots[r].bytes;
    }
    for (auto &B : slots) if (B.seqDev) matcher->devFree(B.seqDev);
}


void MultipleGenomeMatchingProcessor::performMatching() {
    const double t0 = nowSeconds();
    // MBGC_HIP_PROFILE=1: the library's per-family kernel times (HIP events on the streams the kernels run on) over the matching
    // phase, one JSON line on stderr — what bench.py prices its C++-host lines' rooflines with
    const bool profile = getenv("MBGC_HIP_PROFILE") != nullptr;
    if (profile) swsem_profile_enable(matcher->handle(), 1);
    if (params->sequentialMatching) processTargetsWithParallelIO();
    else if (targetsCount && params->exchange) processTargetsRoundsSharded();
    else if (targetsCount) processTargetsRounds();
    else {
        fprintf(stderr, "Error selecting processing mode (no targets for parallel matching?)!\n");
        exit(EXIT_FAILURE);
    }
    refFinalTotalLength = matcher->getRefLength();
    matcher->synchronize();
    // (the reference prints its clock since start here, MGMP.cpp:604; this is the matching phase alone: reading, inflating
    // and parsing the target files, match-finding, processMatches, loadRef)
    if (!params->benchMode && (!params->exchange || mbgc_xchg_rank(params->exchange) == 0))
        fprintf(stderr, "matching finished - %.0f [ms]\n", (nowSeconds() - t0) * 1e3);
    if (profile) {
        static const char *fam[SWSEM_K_COUNT] = {"load", "insert", "probe", "emit2", "resolve", "stitch", "emit"};
        double ms[SWSEM_K_COUNT]; uint64_t nl[SWSEM_K_COUNT];
        swsem_profile_get(matcher->handle(), ms, nl);
        swsem_profile_enable(matcher->handle(), 0);
        fprintf(stderr, "kernel profile: {");
        for (int k = 0; k < SWSEM_K_COUNT; k++) fprintf(stderr, "%s\"%s\": {\"ms\": %.3f, \"launches\": %llu}", k ? ", " : "", fam[k], ms[k], (unsigned long long) nl[k]);
        fprintf(stderr, "}\n");
    }
    if (getenv("MBGC_HIP_TIMES"))
        fprintf(stderr, "  reader threads: reading files %.0f ms; input thread: waiting for them %.0f ms, upload + parse %.0f ms; main thread: waiting for it %.0f ms, taking the streams over %.0f ms"
                        " (%.0f of them waiting for the bytes; appends on their thread %.0f ms, waited for %.0f ms); rounds prepared by the main thread itself %.0f ms; match-finding calls %.0f ms, processMatches calls %.0f ms, loadRef calls %.0f ms; rounds whose first pass gave contigs up as dissimilar: %llu, contigs matched again in units: %llu\n",
                g_tRead * 1e3, g_tReadWait * 1e3, g_tParse * 1e3, g_tWait * 1e3, g_tCollect * 1e3, g_tCollectWait * 1e3, g_tAppend * 1e3, g_tAppendWait * 1e3, g_tPrepareSync * 1e3, g_tMatch * 1e3, g_tEmit * 1e3, g_tFinalize * 1e3,
                (unsigned long long) g_retryPasses, (unsigned long long) g_retryContigs);
}

// ---------------------------------------------------------------- MBGC_Encoder

void MBGC_Encoder::initStreamsForG0Ref() { literals.clear(); }
void MBGC_Encoder::processG0RefContig(const char *seq, size_t len) {
    literals.append(seq, len);
    literals.push_back(SEQ_SEPARATOR_MARK);
}

static const uint32_t BACKEND_FEED_EVERY = 8;                      // targets between two hand-overs to a backend that runs beside the matching

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

// takeRoundStreams + processAfterSequence, then processAfterTarget + appendTargetStreams, for targets that arrive in target
// order: the bytes are appended where appendTargetStreams would put them (ENC.cpp:543-556, :489-496) without the stop in
// the per-target strings
// (room for what is coming: a stream's string grows to what the targets so far project for the whole collection —
// doubling through hundreds of megabytes copies them and touches fresh pages again and again)
static const uintptr_t HUGE_PAGE = 2u << 20;
static void makeRoom(std::string &s, size_t add, uint32_t targetsDone, uint32_t targetsAll) {
    const size_t need = s.size() + add;
    if (need <= s.capacity()) return;
    size_t want = std::max(need, 2 * s.capacity());
    if (targetsDone && targetsAll > targetsDone) {
        const size_t projected = need / targetsDone * targetsAll * 3 / 2;              // (untouched pages cost nothing)
        want = std::max(want, std::min(projected, (targetsDone < 8 ? 16 : 64) * need));   // (one target says little about a thousand)
    }
    s.reserve(want);
    // (fresh memory in 2 MB pages where the system lets a program ask for them: 4 KB page faults cost more than the copy)
    const uintptr_t a = ((uintptr_t) s.data() + s.size() + (HUGE_PAGE - 1)) & ~(uintptr_t) (HUGE_PAGE - 1), b = ((uintptr_t) s.data() + s.capacity()) & ~(uintptr_t) (HUGE_PAGE - 1);
    if (b > a) madvise((void *) a, b - a, MADV_HUGEPAGE);
}

void MBGC_Encoder::appendContigInOrder(const swsem_streams_t &st) {
    makeRoom(literals, st.size[SWSEM_LIT] + 1, targetsAppended + 1, targetsCount);
    makeRoom(mapOff, st.size[SWSEM_OFF], targetsAppended + 1, targetsCount);
    makeRoom(mapLen, st.size[SWSEM_LEN], targetsAppended + 1, targetsCount);
    makeRoom(gapDeltas, st.size[SWSEM_GAP], targetsAppended + 1, targetsCount);
    makeRoom(gapMismatchesFlags, st.size[SWSEM_FLAGS] + 1, targetsAppended + 1, targetsCount);
    literals.append((const char *) st.data[SWSEM_LIT], st.size[SWSEM_LIT]);
    literals.push_back(SEQ_SEPARATOR_MARK);
    mapOff.append((const char *) st.data[SWSEM_OFF], st.size[SWSEM_OFF]);
    mapOff5thByte.append((const char *) st.data[SWSEM_OFF5], st.size[SWSEM_OFF5]);
    mapLen.append((const char *) st.data[SWSEM_LEN], st.size[SWSEM_LEN]);
    gapDeltas.append((const char *) st.data[SWSEM_GAP], st.size[SWSEM_GAP]);
    gapMismatchesFlags.append((const char *) st.data[SWSEM_FLAGS], st.size[SWSEM_FLAGS]);
    unmatchedCharsAll += st.unmatchedChars;
    extensionsMatchedCharsAll += st.extensionsMatchedChars;
    extensionsMismatchesAll += st.extensionsMismatches;
    totalMatchedAll += st.totalMatched;
    removedGapBreakingMatchesAll += st.removedGapBreakingMatches;
}

void MBGC_Encoder::endTargetInOrder() {
    targetsAppended++;
    if (params->emit.enableExtensionsWithMismatches) gapMismatchesFlags.push_back(FILE_SEPARATOR_MARK);
    if (backendStream && targetsAppended % BACKEND_FEED_EVERY == 0) feedBackendStream(false);
}

void MBGC_Encoder::processAfterSequence(uint32_t targetIdx) {
    if (params->sequentialMatching) literals.push_back(SEQ_SEPARATOR_MARK);
    else targetStreams[targetIdx].s[SWSEM_LIT].push_back(SEQ_SEPARATOR_MARK);
}

void MBGC_Encoder::processAfterTarget(uint32_t targetIdx) {
    if (params->emit.enableExtensionsWithMismatches) {
        if (params->sequentialMatching) gapMismatchesFlags.push_back(FILE_SEPARATOR_MARK);
        else targetStreams[targetIdx].s[SWSEM_FLAGS].push_back(FILE_SEPARATOR_MARK);
    }
    if (backendStream && params->sequentialMatching && ++sequentialTargetsDone % BACKEND_FEED_EVERY == 0) feedBackendStream(false);
}

void MBGC_Encoder::processAfterTargetWithParallelIO(size_t matcherLoaderStartPos) {
    processAfterTarget(0);
    afterTargetWithParallelIO(matcherLoaderStartPos);
}

// ... its part that does not touch the literal / flag streams (the pipelined sequential loop appends those a contig later)
void MBGC_Encoder::afterTargetWithParallelIO(size_t matcherLoaderStartPos) {
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
    targetsAppended++;
    if (backendStream && targetsAppended % BACKEND_FEED_EVERY == 0) feedBackendStream(false);
    if (params->lazyDecompressionSupport) {
        matcher->loadSeparator(REF_REGION_SEPARATOR);
        const size_t refExtSize = matcher->getLoadedRefLength() - matcherLoaderStartPos;
        writeUInt64Frugal(refExtSizeStream, refExtSize);
        refExtLoadedPosArr.emplace_back(matcher->getLoadedRefLength());
    }
    locksPosStream.append((const char *) &matchingLocksPos[targetIdx], sizeof(size_t));         // :563
}

void MBGC_Encoder::noteTargetLoaded(uint32_t targetIdx, size_t matcherLoaderStartPos, size_t loadedRefLengthAfter) {
    if (params->lazyDecompressionSupport) {                                                     // ENC.cpp:557-562 (loadSeparator: done by finalizeTargets)
        writeUInt64Frugal(refExtSizeStream, loadedRefLengthAfter - matcherLoaderStartPos);
        refExtLoadedPosArr.emplace_back(loadedRefLengthAfter);
    }
    locksPosStream.append((const char *) &matchingLocksPos[targetIdx], sizeof(size_t));         // :563
}

void MBGC_Encoder::appendTargetStreams(uint32_t targetIdx) {
    appendStreams(*this, targetStreams[targetIdx]);                                             // ENC.cpp:543-556
    targetStreams[targetIdx] = EmittedStreams();
    targetsAppended++;
    if (backendStream && targetsAppended % BACKEND_FEED_EVERY == 0) feedBackendStream(false);
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
    // the first round's files are read (and their page-locked buffer allocated) while the reference file is parsed and the
    // matcher's reference buffer and table are set up
    const int gpus = params->exchange ? mbgc_xchg_world(params->exchange) : 1;
    const bool autoRound = params->roundSize <= 0 && !params->sequentialMatching;
    if (autoRound) {                                                                            // a first guess from the file's size (the sequence is a little shorter): what the read-ahead starts with
        struct stat st;
        const uint64_t g0 = stat(fileNames[0].c_str(), &st) == 0 ? (uint64_t) st.st_size : 0;
        MBGC_Params guessParams = *params;
        if (guessParams.referenceFactor < 1) {
            int tmp = 15 - (__builtin_clz((unsigned) filesCount) / 3);
            guessParams.referenceFactor = 1 << (tmp < 5 ? 5 : (tmp > 12 ? 12 : tmp));
        }
        // (refLengthLimitFor is the base class's and reads ITS parameter pointer: until round 4 only this class's was pointed at the
        // guess, the limit came out 0 and the first read-ahead fetched 64 files — read again, 28 of them, by the first round)
        MGMP_Params *keep = MultipleGenomeMatchingProcessor::params;
        MultipleGenomeMatchingProcessor::params = &guessParams;
        const size_t lim = refLengthLimitFor(std::max<size_t>(g0, MGMP_Params::MIN_BASIC_BLOCK_SIZE), nullptr);
        MultipleGenomeMatchingProcessor::params = keep;
        params->roundSize = (int) windowRoundSize(params->circularReference ? lim / (size_t) params->referenceSlidingWindowFactor : 0, gpus);
    }
    if (!params->sequentialMatching && !params->exchange)
        startReadAhead(1, std::min<uint32_t>(filesCount, 1 + (uint32_t) std::max(1, params->roundSize)), 1, readBesideUpload());
    const double tG0 = nowSeconds();
    loadG0Ref(fileNames[0]);
    if (getenv("MBGC_HIP_TIMES")) fprintf(stderr, "  loadG0Ref (the matcher's buffers and table, the first file): %.0f ms\n", (nowSeconds() - tG0) * 1e3);
    const int guessed = params->roundSize;
    if (autoRound) params->roundSize = (int) windowRoundSize(matcher->getSlidingWindowSize(), gpus);   // (the window as the matcher has it)
    if (getenv("MBGC_HIP_TIMES") && autoRound) fprintf(stderr, "  round size: %d (guessed %d for the first read-ahead)\n", params->roundSize, guessed);
    params->emit.enable40bitReference = params->enable40bitReference;
    if (params->lazyDecompressionSupport) refExtLoadedPosArr.emplace_back(matcher->getLoadingPosition());   // ENC.cpp:789-791
    performMatching();
    // prepareAndCompressStreams' first step on this path, ENC.cpp:636-638: the reverse-complement pass over the literals
    if (params->rcRedundancyRemoval && !params->benchMode && (!params->exchange || mbgc_xchg_rank(params->exchange) == 0))
        PgTools::SimpleSequenceMatcher::rcMatchSequence(literals, rcMapOff, rcMapLen, params->rcMatchMinLength, UINT32_MAX, device);
}

void MBGC_Encoder::backendParams(mbgc_backend_params_t &bp, int blocksScale, int numberOfThreads) const {
    bp = {};
    bp.coderMode = params->coderMode; bp.k = params->k;
    bp.enableExtensionsWithMismatches = params->emit.enableExtensionsWithMismatches;
    bp.mismatchesWithExclusion = params->emit.mismatchesWithExclusion;
    bp.sequentialMatching = params->sequentialMatching; bp.rcRedundancyRemoval = params->rcRedundancyRemoval;
    bp.frugal64bitLenEncoding = params->emit.frugal64bitLenEncoding; bp.lazyDecompressionSupport = params->lazyDecompressionSupport;
    bp.refFinalTotalLength = refFinalTotalLength; bp.blocksScale = blocksScale;
    bp.numberOfThreads = numberOfThreads;                                       // PgHelpers::numberOfThreads (the reference's -t): > 1 gives LZMA two threads, PropsLibrary.cpp:9
}

std::string MBGC_Encoder::compressStreams(mbgc_leaf_compress_fn leaf, void *ctx, int threads, int blocksScale, int numberOfThreads) {
    mbgc_backend_params_t bp;
    backendParams(bp, blocksScale, numberOfThreads > 0 ? numberOfThreads : threads);
    const std::string factors((const char *) unmatchedFractionFactors.data(), unmatchedFractionFactors.size());
    const std::string *src[MBGC_ST_COUNT] = {};
    src[MBGC_ST_UNMATCHED_FRACTION_FACTORS] = &factors; src[MBGC_ST_LITERALS] = &literals; src[MBGC_ST_RC_MAP_OFF] = &rcMapOff;
    src[MBGC_ST_RC_MAP_LEN] = &rcMapLen; src[MBGC_ST_LOCKS_POS] = &locksPosStream; src[MBGC_ST_GAP_DELTAS] = &gapDeltas;
    src[MBGC_ST_GAP_MISMATCHES_FLAGS] = &gapMismatchesFlags; src[MBGC_ST_MAP_OFF] = &mapOff; src[MBGC_ST_MAP_OFF_5TH_BYTE] = &mapOff5thByte;
    src[MBGC_ST_MAP_LEN] = &mapLen; src[MBGC_ST_REF_EXT_SIZE] = &refExtSizeStream;
    const uint8_t *data[MBGC_ST_COUNT] = {};
    uint64_t size[MBGC_ST_COUNT] = {};
    for (int st = 0; st < MBGC_ST_COUNT; st++)
        if (src[st]) { data[st] = (const uint8_t *) src[st]->data(); size[st] = src[st]->size(); }
    uint8_t *out = nullptr;
    uint64_t n = 0;
    if (mbgc_backend_compress_streams(&bp, data, size, leaf, ctx, threads, &out, &n) != 0) {
        fprintf(stderr, "%s\n", mbgc_backend_last_error());
        exit(EXIT_FAILURE);
    }
    std::string res((const char *) out, n);
    mbgc_backend_free(out);
    return res;
}

// what the streams have grown by since the last call (every stream at the end; while the rounds go on only those that the job
// table splits into blocks and that nothing rewrites later)
void MBGC_Encoder::feedBackendStream(bool everything) {
    if (!backendStream) return;
    const std::string factors = everything ? std::string((const char *) unmatchedFractionFactors.data(), unmatchedFractionFactors.size()) : std::string();
    const std::string *src[MBGC_ST_COUNT] = {};
    src[MBGC_ST_GAP_DELTAS] = &gapDeltas; src[MBGC_ST_GAP_MISMATCHES_FLAGS] = &gapMismatchesFlags; src[MBGC_ST_MAP_OFF] = &mapOff; src[MBGC_ST_MAP_LEN] = &mapLen;
    if (!params->rcRedundancyRemoval || everything) src[MBGC_ST_LITERALS] = &literals;
    if (everything) {
        src[MBGC_ST_UNMATCHED_FRACTION_FACTORS] = &factors; src[MBGC_ST_RC_MAP_OFF] = &rcMapOff; src[MBGC_ST_RC_MAP_LEN] = &rcMapLen;
        src[MBGC_ST_LOCKS_POS] = &locksPosStream; src[MBGC_ST_MAP_OFF_5TH_BYTE] = &mapOff5thByte; src[MBGC_ST_REF_EXT_SIZE] = &refExtSizeStream;
    }
    for (int st = 0; st < MBGC_ST_COUNT; st++) {
        if (!src[st] || src[st]->size() <= backendFed[st]) continue;
        if (mbgc_backend_stream_feed(backendStream, st, (const uint8_t *) src[st]->data() + backendFed[st], src[st]->size() - backendFed[st]) != 0) {
            fprintf(stderr, "%s\n", mbgc_backend_last_error());
            exit(EXIT_FAILURE);
        }
        backendFed[st] = src[st]->size();
    }
}

std::string MBGC_Encoder::finishBackendStream(uint64_t *blocksCodedEarly) {
    feedBackendStream(true);
    uint8_t *out = nullptr;
    uint64_t n = 0;
    if (mbgc_backend_stream_finish(backendStream, refFinalTotalLength, &out, &n, blocksCodedEarly) != 0) {
        fprintf(stderr, "%s\n", mbgc_backend_last_error());
        exit(EXIT_FAILURE);
    }
    std::string res((const char *) out, n);
    mbgc_backend_free(out);
    return res;
}
