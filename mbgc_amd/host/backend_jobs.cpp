// The backend's job table and container framing (include/mbgc_backend.h). Host code: which
// stream goes to which coder is policy, the coders are the reference's own behind a callback.
#include "../../include/mbgc_backend.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const char *fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

enum { LEVEL_FAST = 1, LEVEL_NORMAL = 2, LEVEL_MAX = 3 };                  // CodersLib.h:28-30
enum { PERIOD_8 = 0, PERIOD_16 = 1, PERIOD_32 = 2 };                       // LZMA_DATAPERIODCODE_*, CodersLib.h:22-26

// a coder as the job table names it: a leaf, the compound of two leaves, or either in parallel blocks
struct Coder {
    mbgc_leaf_coder_t leaf{};
    bool compound = false;
    mbgc_leaf_coder_t primary{};                       // compound: `leaf` is the secondary coder, run over primary's output
    int blocks = 0;                                    // > 0: ParallelBlocksCoderProps(blocks, this coder)
};

mbgc_leaf_coder_t lzma(int level, int period, int threads) {               // getDefaulLzmaCoderProps, PropsLibrary.cpp:8-22
    mbgc_leaf_coder_t c{};
    c.coder = MBGC_LZMA_CODER;
    c.lc = 3; c.lp = period; c.pb = period; c.algo = -1; c.numThreads = threads > 1 ? 2 : 1;
    if (level == LEVEL_FAST) { c.level = 5; c.dictSize = 1u << 24; c.fb = 6; }
    else { c.level = 9; c.dictSize = 3u << 29; c.fb = level == LEVEL_NORMAL ? 128 : 273; }
    return c;
}

mbgc_leaf_coder_t ppmd(int level, int order) {                             // getDefaultPpmdCoderProps, PropsLibrary.cpp:24-36
    mbgc_leaf_coder_t c{};
    c.coder = MBGC_PPMD7_CODER;
    if (level == LEVEL_FAST) { c.memSize = 16u << 20; c.order = 2; }
    else { c.memSize = 192u << 20; c.order = level == LEVEL_NORMAL && order > 2 ? order / 2 + 1 : order; }
    return c;
}

// MBGC_Encoder::prepareAndCompressStreams, MBGC_Encoder.cpp:641-710
bool jobFor(const mbgc_backend_params_t &p, int st, Coder &c) {
    const bool fast = p.coderMode == 0, repo = p.coderMode == 2, best = p.coderMode >= 3;
    const int th = p.numberOfThreads;
    const int level = best ? LEVEL_MAX : (fast ? LEVEL_FAST : LEVEL_NORMAL);                    // defaultCoderLevel, :689
    c = Coder();
    switch (st) {
    case MBGC_ST_NAMES: c.leaf = ppmd(LEVEL_MAX, 6); return true;                              // :647-648
    case MBGC_ST_SEQ_COUNTS: c.leaf = ppmd(LEVEL_MAX, 4); return true;                         // :649-650
    case MBGC_ST_HEADER_TEMPLATES: c.leaf = lzma(LEVEL_NORMAL, PERIOD_8, th); return true;     // :651-652
    case MBGC_ST_HEADERS:                                                                      // :653-660
        if (p.ultraStreamsCompression) { c.compound = true; c.primary = lzma(LEVEL_NORMAL, PERIOD_8, th); c.leaf = ppmd(LEVEL_NORMAL, 3); }
        else c.leaf = ppmd(LEVEL_MAX, fast ? 8 : 16);
        c.blocks = best ? 1 : 2;
        return true;
    case MBGC_ST_DNA_LINE_LENGTHS: c.leaf = ppmd(LEVEL_MAX, 8); return true;                   // :661-662
    case MBGC_ST_UNMATCHED_FRACTION_FACTORS: c.leaf = ppmd(LEVEL_MAX, 4); return true;         // :663-665
    case MBGC_ST_LITERALS:                                                                     // :666-676
        if (p.ultraStreamsCompression) c.leaf = lzma(LEVEL_MAX, PERIOD_8, th);
        else if (fast || p.k == 16) c.leaf = lzma(LEVEL_FAST, PERIOD_8, th);
        else c.leaf = ppmd(LEVEL_MAX, p.enableExtensionsWithMismatches ? (p.mismatchesWithExclusion ? 5 : 7) : (p.sequentialMatching ? 6 : 7));
        c.blocks = best ? 1 : (fast ? 4 : (repo ? 3 : 2));
        return true;
    case MBGC_ST_RC_MAP_OFF: c.leaf = lzma(LEVEL_MAX, PERIOD_32, th); return p.rcRedundancyRemoval != 0;    // :677-682
    case MBGC_ST_RC_MAP_LEN: c.leaf = lzma(LEVEL_MAX, PERIOD_8, th); return p.rcRedundancyRemoval != 0;
    case MBGC_ST_LOCKS_POS: c.leaf = ppmd(LEVEL_MAX, 8); return true;                          // :683-684
    case MBGC_ST_GAP_DELTAS:                                                                   // :685-689
        c.leaf = ppmd(LEVEL_MAX, best ? 12 : (fast ? 2 : 8));
        c.blocks = repo ? 4 : (best ? 1 : 2);
        return true;
    case MBGC_ST_GAP_MISMATCHES_FLAGS:                                                         // :690-696
        c.leaf = ppmd(LEVEL_MAX, fast ? 8 : 14);
        c.blocks = fast ? 3 : 2;
        return p.enableExtensionsWithMismatches != 0;
    case MBGC_ST_MAP_OFF:                                                                      // :697-700
        c.leaf = lzma(level, PERIOD_32, th);
        c.blocks = best ? 1 : (fast ? 4 : 3);
        return true;
    case MBGC_ST_MAP_OFF_5TH_BYTE: c.leaf = ppmd(LEVEL_MAX, 4); return p.refFinalTotalLength > UINT32_MAX;   // :701-703
    case MBGC_ST_MAP_LEN:                                                                      // :704-708
        c.leaf = lzma(level, p.frugal64bitLenEncoding ? PERIOD_16 : PERIOD_32, th);
        c.blocks = best ? 2 : (fast ? 10 : (repo ? 3 : 5));
        return true;
    case MBGC_ST_REF_EXT_SIZE: c.leaf = ppmd(LEVEL_MAX, 6); return p.lazyDecompressionSupport != 0;          // :709-712
    default: return false;
    }
}

struct Sink {
    std::string s;
    template<typename T> void put(T v) { s.append((const char *) &v, sizeof v); }
    void put(const uint8_t *p, size_t n) { s.append((const char *) p, n); }
};

struct Job { const uint8_t *src; size_t n; Coder coder; std::string packed; uint64_t compLen = 0; bool failed = false; };

struct Ctx { mbgc_leaf_compress_fn leaf; void *ctx; int blocksScale = 1; };

bool leafCompress(const Ctx &x, const mbgc_leaf_coder_t &c, const uint8_t *src, size_t n, std::string &out) {
    out.resize(n + n / 3 + 256);
    uint64_t len = 0;
    if (x.leaf(x.ctx, &c, src, n, (uint8_t *) &out[0], out.size(), &len) != 0 || len > out.size()) return false;
    out.resize(len);
    return true;
}

bool compress(const Ctx &x, Job &j, bool inBlocks);

// header + payload of one compressed job, writeHeader (CodersLib.cpp:202-218) + the bytes; a job that does not shrink is stored raw
void writeJob(Sink &out, Job &j) {
    if (j.n == 0) { out.put<uint64_t>(0); return; }
    const size_t destLen = j.packed.size();
    out.put<uint64_t>(j.n);
    if (j.coder.compound && !j.coder.blocks) {
        out.put<uint64_t>(destLen + 17);                                // + primaryCoder->getHeaderLen()
        out.put<uint8_t>(MBGC_COMPOUND_CODER);
        out.put<uint64_t>(j.compLen);
        out.put<uint8_t>(j.compLen == j.n ? MBGC_NO_CODER : (uint8_t) j.coder.primary.coder);
        out.put<uint64_t>(j.compLen);                                   // the secondary coder's own header over the primary's output
        if (destLen >= j.compLen) { out.put<uint64_t>(j.compLen); out.put<uint8_t>(MBGC_NO_CODER); }
        else { out.put<uint64_t>(destLen); out.put<uint8_t>((uint8_t) j.coder.leaf.coder); }
    } else if (destLen >= j.n) {
        out.put<uint64_t>(j.n); out.put<uint8_t>(MBGC_NO_CODER);
    } else {
        out.put<uint64_t>(destLen);
        out.put<uint8_t>(j.coder.blocks ? (uint8_t) MBGC_PARALLEL_BLOCKS_CODER : (uint8_t) j.coder.leaf.coder);
    }
    if (destLen < j.n) out.put((const uint8_t *) j.packed.data(), destLen);
    else out.put(j.src, j.n);
    j.packed.clear(); j.packed.shrink_to_fit();
}

// CompressionJob::writeCompressedCollectiveParallel, CodersLib.cpp:372-415: every job compressed (in parallel), then
// header + payload per job in order
bool collective(const Ctx &x, std::vector<Job> &jobs, Sink &out, int threads) {
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < jobs.size();) if (jobs[i].n) jobs[i].failed = !compress(x, jobs[i], false); };
    const size_t nt = threads > 0 ? std::min<size_t>(threads, jobs.size()) : jobs.size();
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (Job &j : jobs) if (j.failed) return false;
    for (Job &j : jobs) writeJob(out, j);
    return true;
}

// Compress, CodersLib.cpp:53-132
bool compress(const Ctx &x, Job &j, bool inBlocks) {
    if (j.coder.blocks && !inBlocks) {
        // parallelBlocksCompress, CodersLib.cpp:292-314 (+ ParallelBlocksCoderProps::prepare, CodersLib.h:163-171)
        int nb = j.coder.blocks * (x.blocksScale > 1 ? x.blocksScale : 1);
        const int maxBlocks = (int) (j.n / (1u << 20));
        if (nb > maxBlocks) nb = maxBlocks;
        if (nb == 0) nb = 1;
        const size_t blockSize = ((j.n / nb) / 16) * 16;
        Coder inner = j.coder;
        inner.blocks = 0;
        std::vector<Job> blocks;
        size_t off = 0;
        for (int i = 0; i < nb - 1; i++) { blocks.push_back(Job{j.src + off, blockSize, inner}); off += blockSize; }
        blocks.push_back(Job{j.src + off, j.n - off, inner});
        Sink s;
        s.put<int32_t>(nb);
        if (!collective(x, blocks, s, 0)) return false;
        j.packed = std::move(s.s);
        return true;
    }
    if (j.coder.compound) {
        std::string first;
        if (!leafCompress(x, j.coder.primary, j.src, j.n, first)) return false;
        if (first.size() >= j.n) first.assign((const char *) j.src, j.n);
        j.compLen = first.size();
        if (!leafCompress(x, j.coder.leaf, (const uint8_t *) first.data(), first.size(), j.packed)) return false;
        if (j.packed.size() >= j.compLen) j.packed = std::move(first);
        return true;
    }
    return leafCompress(x, j.coder.leaf, j.src, j.n, j.packed);
}


// ---- the incremental form: blocks of the split streams are coded while the streams still grow ---------------------------
struct Piece { std::string raw; Job job; bool freed = false; };

}  // namespace

struct mbgc_backend_stream {
    mbgc_backend_params_t p{};
    Ctx x{};
    uint64_t blockBytes = 0;
    struct St { bool enrolled = false; Coder coder; std::string pending; uint64_t total = 0; std::deque<std::unique_ptr<Piece>> pieces; } st[MBGC_ST_COUNT];
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<Piece *> queue;
    size_t running = 0, codedEarly = 0;
    bool closing = false, failed = false, finishing = false;
    std::vector<std::thread> pool;

    void work() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return closing || !queue.empty(); });
            if (queue.empty()) return;
            Piece *pc = queue.front();
            queue.pop_front();
            running++;
            lk.unlock();
            const bool ok = compress(x, pc->job, true);                // (a block is coded by the stream's coder itself, CodersLib.cpp:300-306)
            // A block that shrank is written from its packed bytes: its raw bytes go now, not when the section is written (a stream
            // of gigabytes would otherwise stand in memory three times: with the caller, in the pieces, in the raw fallback's copy).
            // "Shrank" with the container's own bytes to spare (17 of header per block, 4 for the count), so that a stream whose
            // blocks all went this way is certain to be smaller as a container than raw — the one case that needs the raw bytes back.
            if (ok && pc->job.packed.size() + 32 <= pc->raw.size()) {
                std::string().swap(pc->raw);
                pc->job.src = nullptr;
                pc->freed = true;
            }
            lk.lock();
            running--;
            if (!ok) failed = true;
            if (!finishing) codedEarly++;
            if (queue.empty() && running == 0) idle.notify_all();
        }
    }
    void enqueue(St &t, std::string &&raw) {                            // mu held
        t.pieces.emplace_back(new Piece());
        Piece *pc = t.pieces.back().get();
        pc->raw = std::move(raw);
        Coder inner = t.coder;
        inner.blocks = 0;
        pc->job = Job{(const uint8_t *) pc->raw.data(), pc->raw.size(), inner};
        queue.push_back(pc);
        cv.notify_one();
    }
};

namespace {
}  // namespace

extern "C" {

const char *mbgc_backend_last_error(void) { return g_err.c_str(); }

int mbgc_backend_job(const mbgc_backend_params_t *p, int st, int *blocks, mbgc_leaf_coder_t *coder, mbgc_leaf_coder_t *primary) {
    Coder c;
    if (!p || !jobFor(*p, st, c)) return -1;
    if (blocks) *blocks = c.blocks;
    if (coder) *coder = c.leaf;
    if (primary) { *primary = c.primary; if (!c.compound) primary->coder = 0; }
    return 0;
}

int mbgc_backend_compress_streams(const mbgc_backend_params_t *p, const uint8_t *const data[MBGC_ST_COUNT], const uint64_t size[MBGC_ST_COUNT],
                                  mbgc_leaf_compress_fn leaf, void *ctx, int threads, uint8_t **out, uint64_t *outLen) {
    if (!p || !data || !size || !leaf || !out || !outLen) return fail("mbgc_backend_compress_streams: null argument");
    std::vector<Job> jobs;
    for (int st = 0; st < MBGC_ST_COUNT; st++) {
        Coder c;
        if (!jobFor(*p, st, c)) continue;
        if (size[st] && !data[st]) return fail("mbgc_backend_compress_streams: stream %d has a size and no bytes", st);
        jobs.push_back(Job{data[st], (size_t) size[st], c});
    }
    Sink s;
    const Ctx x{leaf, ctx, p->blocksScale};
    if (!collective(x, jobs, s, threads)) return fail("Error during compression.");       // (the reference exits, CodersLib.cpp:103-108)
    *out = (uint8_t *) malloc(s.s.size() ? s.s.size() : 1);
    if (!*out) return fail("out of memory");
    memcpy(*out, s.s.data(), s.s.size());
    *outLen = s.s.size();
    return 0;
}

void mbgc_backend_free(uint8_t *p) { free(p); }

mbgc_backend_stream_t *mbgc_backend_stream_open(const mbgc_backend_params_t *p, mbgc_leaf_compress_fn leaf, void *ctx, int threads, uint64_t blockBytes) {
    if (!p || !leaf) { fail("mbgc_backend_stream_open: null argument"); return nullptr; }
    if (blockBytes < (1u << 20) || blockBytes % 16) { fail("mbgc_backend_stream_open: blocks are multiples of 16 bytes, 2^20 at least (CodersLib.h:163-171)"); return nullptr; }
    auto *s = new mbgc_backend_stream();
    s->p = *p;
    s->x = Ctx{leaf, ctx, 1};
    s->blockBytes = blockBytes;
    for (int st = 0; st < MBGC_ST_COUNT; st++) s->st[st].enrolled = jobFor(*p, st, s->st[st].coder);
    for (int t = 0; t < (threads > 0 ? threads : 1); t++) s->pool.emplace_back([s] { s->work(); });
    return s;
}

int mbgc_backend_stream_feed(mbgc_backend_stream_t *s, int st, const uint8_t *data, uint64_t n) {
    if (!s || st < 0 || st >= MBGC_ST_COUNT || (n && !data)) return fail("mbgc_backend_stream_feed: bad argument");
    auto &t = s->st[st];
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->finishing) return fail("mbgc_backend_stream_feed: the section is being written");
    t.total += n;
    if (!t.coder.blocks) { t.pending.append((const char *) data, n); return 0; }
    // a split stream gives up a block as soon as more than a block is waiting (the rest stays: the last block is what is left);
    // every byte is copied once, into the block it will be coded from
    while (t.pending.size() + n > s->blockBytes) {
        const size_t take = s->blockBytes - t.pending.size();
        t.pending.append((const char *) data, take);
        data += take; n -= take;
        std::string blk;
        blk.swap(t.pending);
        s->enqueue(t, std::move(blk));
        t.pending.reserve((size_t) std::min<uint64_t>(s->blockBytes, 1u << 26));
    }
    t.pending.append((const char *) data, n);
    return 0;
}

int mbgc_backend_stream_finish(mbgc_backend_stream_t *s, uint64_t refFinalTotalLength, uint8_t **out, uint64_t *outLen, uint64_t *blocksCodedEarly) {
    if (!s || !out || !outLen) return fail("mbgc_backend_stream_finish: null argument");
    {
        std::unique_lock<std::mutex> lk(s->mu);
        if (s->finishing) return fail("mbgc_backend_stream_finish: called twice");
        s->finishing = true;
        if (blocksCodedEarly) *blocksCodedEarly = s->codedEarly;
        s->p.refFinalTotalLength = refFinalTotalLength;                 // (known when the matching is over; it decides the 5th-byte stream, :698-699)
        for (int st = 0; st < MBGC_ST_COUNT; st++) {
            Coder c;
            const bool en = jobFor(s->p, st, c);
            if (s->st[st].pieces.empty()) { s->st[st].enrolled = en; s->st[st].coder = c; }
        }
        for (auto &t : s->st)
            if (t.enrolled && !t.pending.empty()) s->enqueue(t, std::move(t.pending));
        s->idle.wait(lk, [&] { return s->queue.empty() && s->running == 0; });
        if (s->failed) return fail("Error during compression.");
    }
    Sink sec;
    for (auto &t : s->st) {
        if (!t.enrolled) continue;
        if (t.pieces.empty()) { sec.put<uint64_t>(0); continue; }
        if (!t.coder.blocks) { writeJob(sec, t.pieces.front()->job); continue; }
        // parallelBlocksCompress's container (CodersLib.cpp:292-314) over the blocks as they were cut: the reader takes every
        // block's length from the block's own header (parallelBlocksDecompress, :316-345)
        Sink blocks;
        blocks.put<int32_t>((int32_t) t.pieces.size());
        bool anyFreed = false;
        for (auto &pc : t.pieces) anyFreed |= pc->freed;
        // (the raw fallback of the whole stream — a container that is no smaller than the bytes, CodersLib.cpp:202-218 — needs them
        // all: only a stream none of whose blocks shrank can get there, and then every block still has its bytes)
        std::string whole;
        if (!anyFreed) for (auto &pc : t.pieces) whole += pc->raw;
        for (auto &pc : t.pieces) { writeJob(blocks, pc->job); std::string().swap(pc->raw); }
        if (anyFreed && blocks.s.size() >= t.total) {                  // (cannot happen when every block was freed; blocks that did not shrink beside ones that did)
            sec.put<uint64_t>(t.total); sec.put<uint64_t>(blocks.s.size()); sec.put<uint8_t>((uint8_t) MBGC_PARALLEL_BLOCKS_CODER);
            sec.put((const uint8_t *) blocks.s.data(), blocks.s.size());
            continue;
        }
        Job outer{(const uint8_t *) whole.data(), (size_t) t.total, t.coder};
        outer.packed = std::move(blocks.s);
        writeJob(sec, outer);
    }
    *out = (uint8_t *) malloc(sec.s.size() ? sec.s.size() : 1);
    if (!*out) return fail("out of memory");
    memcpy(*out, sec.s.data(), sec.s.size());
    *outLen = sec.s.size();
    return 0;
}

void mbgc_backend_stream_close(mbgc_backend_stream_t *s) {
    if (!s) return;
    { std::lock_guard<std::mutex> lk(s->mu); s->closing = true; s->queue.clear(); }
    s->cv.notify_all();
    for (auto &t : s->pool) t.join();
    delete s;
}


}  // extern "C"
