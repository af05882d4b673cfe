// Host runtime + C ABI (include/mbgc_swsem.h) of the MI355X match-finding path.
// The host keeps exactly the scalar state the reference keeps in SlidingWindowSparseEMMatcher
// (pos1, reachedRefLengthCount, samplingPos, swEnd, the worker-lock deque); the reference bytes,
// the hash table and every per-round intermediate live in HBM.
#include "../../include/mbgc_swsem.h"
#include "swsem_kernels.hip"
#include "swsem_resolve4.hip"
#include "swsem_emit.hip"
#include "swsem_decode.hip"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

using namespace swk;

#define SWSEM_ESPEC (-100)   /* internal: a speculative finalize cannot be queued (it would need an ungated write) */

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SWSEM_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

constexpr uint64_t REF_SHIFT = 1;          // SlidingWindowSparseEMMatcher.h:14
constexpr int SW_WIDTH_FACTOR = 16;        // .h:47
constexpr uint64_t REF_SLACK = 256;

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return SWSEM_OK;
        if (p) (void) hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        if (hipMalloc((void **) &p, want * sizeof(T)) != hipSuccess) {
            p = nullptr;
            return fail(SWSEM_ENOMEM, "device allocation of %zu bytes failed", want * sizeof(T));
        }
        cap = want;
        return SWSEM_OK;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
};

struct ProfEvent { hipEvent_t a, b; int fam; };

// ------------------------------------------------------------------------------------------------------------------
// The side streams: a process-wide pool per device, dealt to a handle by MEASUREMENT.
// A HIP stream's hardware queue is served by one of the device's four dispatch pipes (the k-th queue a process makes
// goes to pipe k mod 4, whatever its priority: profiles/queue_pipes.hip), and a pipe works on one launch at a time: a
// kernel with more workgroups than the device holds keeps its pipe until its last workgroup has been dispatched, and a
// kernel queued meanwhile on another stream of the same pipe starts behind it. Which pipe the caller's stream sits on
// depends on how many queues its framework made before — the step time of a round moved between 2.6 and 3.0 ms with
// nothing but that (profiles/r04_stream_pipes.md). So the library makes eight candidate streams once, finds out which
// of them get in each other's way (a long-dispatch kernel on one, a one-workgroup kernel on the other), and gives every
// handle streams that do not share a pipe with its main stream or with each other where both are busy at once:
//   stream2     low     the byte automata of an emission's second phase (beside the next batch's chains, then the stitch)
//   streamAux   normal  the stitch (beside the chains' tail), later the emission's pairing kernels (beside the insertion)
//   streamLoad  high    the finalize's copies (beside the insertion)
//   streamUp    normal  table uploads (a few microseconds at a batch's start)
// SWSEM_STREAM_CALIB=0: no measurement, the candidates in the order they were made.
struct SidePool {
    static constexpr int NC = 8;
    bool made = false, ok = false;
    hipStream_t cand[NC] = {};
    int cls[NC] = {1, 1, 1, 1, 0, 0, 2, 2};     // 0 low, 1 normal, 2 high priority
    int label[NC] = {};                          // candidates with one label get in each other's way
    unsigned wgs = 8192;
};
SidePool g_pools[16];
std::mutex g_poolMu;

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// does a one-workgroup kernel on b wait for a long-dispatch kernel on a?
bool streams_collide(hipStream_t a, hipStream_t b, unsigned wgs) {
    int hits = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void) hipStreamSynchronize(a); (void) hipStreamSynchronize(b);
        const double t0 = now_us();
        k_hog<<<dim3(wgs), dim3(256), 0, a>>>(1500);                 // 15 us per workgroup, four generations of them
        k_touch<<<1, 1, 0, b>>>();
        (void) hipStreamSynchronize(b);
        const double t1 = now_us();
        (void) hipStreamSynchronize(a);
        const double t2 = now_us();
        if (t1 - t0 > 0.6 * (t2 - t0)) hits++;
    }
    return hits == 2;
}

SidePool *side_pool(int device, int prioLow, int prioHigh) {
    if (device < 0 || device >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_poolMu);
    SidePool &P = g_pools[device];
    if (P.made) return P.ok ? &P : nullptr;
    P.made = true;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) P.wgs = (unsigned) pr.multiProcessorCount * 8u * 4u;
    for (int i = 0; i < SidePool::NC; i++) {
        const int pv = P.cls[i] == 0 ? prioLow : (P.cls[i] == 2 ? prioHigh : (prioLow + prioHigh) / 2);
        if (hipStreamCreateWithPriority(&P.cand[i], hipStreamNonBlocking, pv) != hipSuccess) return nullptr;
        k_touch<<<1, 1, 0, P.cand[i]>>>();                            // first use: the stream is given its hardware queue now
        if (hipStreamSynchronize(P.cand[i]) != hipSuccess) return nullptr;
    }
    const char *e = getenv("SWSEM_STREAM_CALIB");
    const bool measure = !(e && atoi(e) == 0);
    for (int i = 0; i < SidePool::NC; i++) {
        P.label[i] = i;
        for (int j = 0; measure && j < i; j++)
            if (P.label[j] == j && streams_collide(P.cand[j], P.cand[i], P.wgs)) { P.label[i] = j; break; }
    }
    (void) hipGetLastError();
    P.ok = true;
    return &P;
}

}  // namespace

struct swsem {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // --- reference state (names follow SlidingWindowSparseEMMatcher.h:29-49,78)
    uint8_t *ref = nullptr;                // start1
    ht_entry *ht = nullptr;
    uint8_t *lut = nullptr;                // upper-complement LUT, utils/helper.cpp:312-338
    int64_t pos1 = REF_SHIFT;
    uint64_t maxRefLength = 0;
    int laps = 0;                          // reachedRefLengthCount
    int L = 0, K = 0, k1 = 0, k2 = 1, skipMargin = 0, k1ord = 0;
    uint32_t hash_size = 0, mask = 0;
    uint64_t samplingPos = 0, swSize = 0, swEnd = 0;
    bool circular = true;
    std::deque<uint64_t> locks;            // workersSwEndPositions
    uint32_t epoch = 1;
    uint32_t eCur = 0, ePrev = 0;          // first epoch of the current / previous lap (ht_value's staleness test)
    // the separator at the window's end (the byte before the loading position, when the loader stands at the window's
    // end) has been written with this value at this position in this lap: writing it again changes nothing
    int64_t sepEndPos = -1; int sepEndLaps = -1, sepEndVal = -1;
    bool sep_end_done(int sep) const { return pos1 == sepEndPos && laps == sepEndLaps && sep == sepEndVal; }
    void sep_end_set(int64_t at, int sep) { sepEndPos = at; sepEndLaps = laps; sepEndVal = sep; }
    bool pristine = true;                  // the loader has only moved forward (wraps included: told by epochs); false after swsem_set_position
    int fpBits = 0;                        // fingerprint bits of a table entry
    uint64_t droppedBytes = 0;             // extension bytes loadRef gave up at the window's end (.cpp:433: the rest of a text is dropped when the loader reaches swEnd)
    uint64_t hostProbes = 0;               // query positions of the batch
    bool deferInserts = false;             // collect the insertion phases of a finalize call into one launch
    bool specMode = false;                 // a speculative finalize is being queued: nothing may be written outside its gated launches
    DevBuf<uint32_t> dGate;
    DevBuf<uint8_t> dPred;
    std::vector<InsertPiece> pendingPieces;
    std::vector<InsertPiece> edgePieces;   // flush_inserts: the samples left to the launch behind the copies
    bool insertBeside = true;              // insertion hashes from the copies' sources, the copies run beside it (SWSEM_INSERT_BESIDE=0: one after the other)
    hipStream_t streamLoad = nullptr;      // ... on this stream
    int prioLow = 0, prioHigh = 0;         // the device's stream priority range
    hipEvent_t evLoadFork = nullptr, evLoadDone = nullptr;
    std::vector<CopyPiece> pendingCopies;    // ... and its byte writes: device-to-device copies,
    std::vector<BytePiece> pendingBytes;     // then single bytes (separators), each list in program order
    DevBuf<uint64_t> dTables;                // one upload: insert pieces, their prefix, copy pieces, their prefix, bytes
    struct HostTab {                         // pinned staging for that upload; reused once its copy has completed
        uint64_t *p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false;
    } hostTables[2];
    int hostTableSel = 0;
    // what prepare_inserts has uploaded for the launches that follow (launch_insert_early / launch_inserts)
    struct PreparedInserts {
        size_t np = 0, nc = 0, nb = 0, ne = 0;
        bool beside = false;
        uint64_t nSamples = 0, nEdge = 0, copyBlocks = 0;
        const InsertPiece *dPieces = nullptr, *dEdge = nullptr;
        const uint64_t *dFirst = nullptr, *dCFirst = nullptr, *dEFirst = nullptr;
        const CopyPiece *dCopies = nullptr;
        const BytePiece *dBytes = nullptr;
    } prep;
    uint64_t copyWgs = 2048;               // workgroups of the copies (a grid that is resident at once: k_copy_multi)
    hipStream_t streamUp = nullptr, streamAux = nullptr;      // (dealt from the device's pool, like stream2 and streamLoad: deal_streams) table uploads; the stitch and the pairing kernels
    hipEvent_t evStitched = nullptr, evTables = nullptr, evRoundTop = nullptr;
    bool roundTopFresh = false;            // evRoundTop was recorded by the batch this emission belongs to (run_batch), not an older one
    // --- per-round scratch
    DevBuf<uint8_t> stage;                 // host text / host query staging
    DevBuf<Contig> dContigs;
    DevBuf<uint32_t> dMatchCount, dRbContig, dRbOrder;
    std::vector<uint32_t> rbOrderHost, rbOrderKey;   // the table on the device is kept while the batches keep their shape (rbOrderKey)
    bool simt = true;                      // four chains per wave (k_resolve_blocks4); SWSEM_CHAINS=1: one chain per wave (k_resolve_blocks)
    uint32_t chainsPerWave = 1;            // of the last batch
    hipStream_t stream3 = nullptr;         // device-to-host copies of emitted streams (end_slot): made when first needed (see s3)
    hipStream_t s3() {
        if (!stream3 && hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking) != hipSuccess) stream3 = stream;
        return stream3;
    }
    hipEvent_t evMatched = nullptr;        // chains of the last batch done
    std::vector<uint32_t> rbContigHost;
    DevBuf<Match> dMatches;
    DevBuf<Row> dRegions, dReplay;
    DevBuf<BlockRec> dRecs;
    DevBuf<FastRec> dFast;
    DevBuf<uint32_t> dSegStart, dKeepN, dDstOff;
    DevBuf<int32_t> dPrev;
    DevBuf<unsigned long long> dStats;
    DevBuf<uint8_t> dDecode;                 // contigs given back by the device decoder (swsem_emit_verify)
    DevBuf<DecodeJob> dJobs;
    DevBuf<DecRec> dDecRecs;                 // the plan pass's records, contig after contig (swsem_decode.hip)
    DevBuf<DecPlanOut> dDecPlan;
    DevBuf<uint64_t> dDecAux;                // per contig: record base (n + 1), first differing byte (n), malformed flag (n, as u32 pairs)
    // --- emission: two slots, so that the second phase of one batch can still be running while the next is begun
    struct EmitSlot {
        DevBuf<EmitContig> dECg;
        DevBuf<EmitOut> dEOut;
        DevBuf<int> dEWhich;
        DevBuf<uint32_t> dEOwner, dESpanOwner;
        DevBuf<EMatch> dEM;
        DevBuf<uint64_t> dENext0, dELoaded, dEPack;
        DevBuf<uint8_t> dERm, dEArena;
        DevBuf<uint32_t> dEKeep, dEMeta, dECorr, dESz, dEOfs, dEChunk;
        DevBuf<MetaRec> dEStates;
        DevBuf<unsigned long long> dEStat, dEPm, dELit, dEBad;
        DevBuf<LongCopy> dELong;
        DevBuf<uint32_t> dELongCount;
        bool statZeroed = false;
        std::vector<EmitContig> ecg;
        std::vector<uint32_t> chunkOwner, spanOwner;
        std::vector<int> ewhich;
        std::vector<uint64_t> eloaded;
        std::vector<EmitOut> eout;
        hipEvent_t evDone = nullptr;
        hipEvent_t evMetaDone = nullptr;     // behind the pairing kernels (the byte automata wait for it)
        bool outstanding = false, refGuarded = false;
        bool donePending = false;            // evDone has not been recorded for this emission yet (its byte automata are queued behind the speculative finalize)
        // the byte automata of the second phase wait to be queued: at the next batch's resolve launch, gated on that kernel's
        // start (run_phase2b), or by whoever needs this emission's results first
        bool deferred2b = false, waitFin2b = false;
        EmitView v2b; uint32_t grid2b = 0; int n2b = 0;
        const uint8_t *qdev = nullptr;       // query buffer the emission reads
        swsem_emit_params_t params;          // its parameters
        uint64_t emitPos1 = 0;               // loading position the emission started at
        uint64_t lockMin = UINT64_MAX;       // lowest matching-lock position of its contigs (UINT64_MAX: some contig had none)
        int emitLaps = 0;                    // laps of the buffer when it started
        int emitN = 0;
        uint8_t *pinE = nullptr; size_t pinECap = 0;
        // the streams on the host: page-locked (the copy runs at the link's rate and nothing is zero-filled), grow-only
        struct PinBytes {
            uint8_t *p = nullptr; size_t cap = 0;
            uint8_t *data() { return p; }
            int reserve(size_t n) {
                if (n <= cap) return 0;
                if (p) (void) hipHostFree(p);
                p = nullptr; cap = 0;
                const size_t want = n + n / 2 + 4096;
                if (hipHostMalloc((void **) &p, want, hipHostMallocDefault) != hipSuccess) return -1;
                cap = want;
                return 0;
            }
            void release() { if (p) (void) hipHostFree(p); p = nullptr; cap = 0; }
        } hostHalf[2];
        // Two host buffers per slot, used in turn: the views of an emission (swsem_emit_view) stay valid while the NEXT emission
        // of the same slot is begun, runs and is taken — a caller that copies the bytes out on a thread of its own has two
        // emissions' time for it, not the gap between taking one emission and beginning the next (mgmp_driver.cpp: the large
        // literal and flag streams of divergent collections, 0.4 bytes per base, were waited for there).
        int hostAt = 0;
        PinBytes &hostStreams() { return hostHalf[hostAt]; }
        std::vector<uint64_t> hostStreamOff; // [k * NSTREAMS + s] offset into hostStreams()
        bool hostStreamsValid = false;
        uint64_t packedBytes = 0;
        void release() {
            dECg.release(); dEOut.release(); dEWhich.release(); dEOwner.release(); dESpanOwner.release(); dEM.release(); dENext0.release(); dELoaded.release();
            dEPack.release(); dERm.release(); dEArena.release(); dEKeep.release(); dEMeta.release(); dECorr.release();
            dESz.release(); dEOfs.release(); dEChunk.release(); dEStates.release(); dEStat.release(); dEPm.release(); dELit.release(); dEBad.release(); dELong.release(); dELongCount.release();
            if (pinE) { (void) hipHostFree(pinE); pinE = nullptr; pinECap = 0; }
            hostHalf[0].release(); hostHalf[1].release();
            if (evDone) { (void) hipEventDestroy(evDone); evDone = nullptr; }
            if (evMetaDone) { (void) hipEventDestroy(evMetaDone); evMetaDone = nullptr; }
        }
    } slot[2];
    int latest = 0;                          // slot of the last swsem_emit_batch_begin
    // largest request seen so far: a slot is always sized for it, so the second slot does not regrow (= hipFree +
    // hipMalloc, a device-wide stall) the first time it meets a full-size batch
    uint64_t capN = 0, capRows = 0, capArena = 0, capLoaded = 0, capChunks = 0;
    int selected = -1;                       // slot the result calls read (-1: the latest), swsem_emit_select
    EmitSlot &sel() { return slot[selected < 0 ? latest : selected]; }
    uint8_t *pin = nullptr; size_t pinCap = 0, pinExtraAt = 0;
    // small host tables travel through a pinned ring: an asynchronous copy from pageable memory is staged by the
    // runtime and can block the calling thread for milliseconds when its staging pool is busy
    uint8_t *ring = nullptr; size_t ringCap = 0, ringAt = 0;
    // emission in two phases: pass 1 (what the extension policy needs) on `stream`, the rest on `stream2` behind evP1,
    // so that the caller can queue the round's finalize and the next round's match-finding next to it
    hipStream_t stream2 = nullptr;
    hipEvent_t evP1 = nullptr;
    hipEvent_t evFin = nullptr;            // behind the speculative finalize (see emit_begin_impl)
    hipEvent_t evMeta = nullptr;           // behind the last emission's k_emit_meta_spec
    bool metaPending = false;
    int metaWarm = swk::MWARM;             // warm-up matches of the pairing chain's speculative blocks (SWSEM_META_WARM: fewer, so that blocks fail)
    bool phase2Behind = true;              // the second phase's byte automata are handed over behind the speculative finalize
    bool emitHostCopy = true;              // copy the streams to the host inside swsem_emit_batch
    bool seqResolve = false;               // SWSEM_RESOLVE=seq: one wave per contig (cross-check path)
    // Warm-up positions of a speculative block chain (at most OVERLAP_MAX): a chain started from the empty state falls into step with the
    // true one after a few emissions, and how many positions that takes depends on the collection (on how far apart its matches
    // lie). Too short and blocks are replayed by the stitch, one after the other; too long and every block scans positions twice.
    // Adapted from the share of replayed blocks the last full batch reported (take_counts): the results never depend on it.
    uint32_t overlap = 1024, overlapFixed = 0, batchBlocks = 0;
    uint32_t rb = 8;                       // length of a resolve block in units of RBU positions: chosen per batch (run_batch) unless SWSEM_RB fixes it
    uint32_t rbFixed = 0;
    uint64_t stitchDiag[4] = {0, 0, 0, 0};  // over the handle's life: resolve blocks replayed / accepted in runs / tested one by one / jumped over (SWSEM_DEBUG_STATS)
    uint32_t emitThinMax = 512;            // emissions of at most this many chunks of 256 gap tasks run their byte automata with 16 tasks per wave (SWSEM_EMIT_THIN_MAX)
    uint32_t rbMin = 2048 / RBU;           // shortest resolve block (units of RBU positions); SWSEM_RB_MIN=positions for A/B runs
    uint32_t slotPercent = 95;             // share of the wave slots a launch's blocks are sized for (80 %: +5 % on the 4.35e9-byte sizing, -3 % on configs[2]'s)
    uint32_t waveSlots = 256 * 4 * RESOLVE_WAVES_PER_SIMD;   // resolve waves the device holds at once (CUs x SIMDs x waves)
    std::vector<Contig> contigs;
    std::vector<uint32_t> matchCount;
    std::vector<swsem_match_t> hostMatches;
    const uint8_t *qdev = nullptr;         // query buffer of the last batch
    uint32_t minLen = 0;
    bool batchValid = false;
    uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
    // --- profiling
    bool prof = false;
    uint32_t profMask = ~0u;               // families that get event brackets (SWSEM_PROF_FAMS: every bracket is two markers in the queue)
    std::deque<ProfEvent> events;
    double profMs[SWSEM_K_COUNT] = {0};
    uint64_t profN[SWSEM_K_COUNT] = {0};

    CopySegs segs;                         // small copies staged for one launch (stage_copy / flush_copies)
    hipStream_t segStream = nullptr;

    uint16_t *tags = nullptr;              // per sampling slot: lap_tag of its last on-grid sampling (swsem_device.h, lap_want)
    bool useTags = true;                   // SWSEM_LAP_TAGS=0: every stale entry is visited (the table image and the results are the same)
    uint64_t refLength() const { return laps ? maxRefLength : (uint64_t) pos1; }
    RefView view() const {
        RefView v;
        v.ref = ref; v.ht = ht; v.pos1 = (uint64_t) pos1; v.refLength = refLength(); v.maxRefLength = maxRefLength;
        v.mask = mask; v.fpBits = fpBits; v.fpCheck = (fpBits && pristine) ? (laps ? 2 : 1) : 0; v.eCur = eCur; v.ePrev = ePrev;
        // (position << k1ord) + K + 1 <= pos1  <=>  value <= curMax;   (position << k1ord) >= pos1  <=>  value >= prevMin
        v.curMax = pos1 >= (int64_t) K + 1 ? (uint32_t) (((uint64_t) pos1 - K - 1) >> k1ord) : 0u;
        v.prevMin = (uint32_t) ((((uint64_t) pos1) + (1ull << k1ord) - 1) >> k1ord); v.K = K; v.k1ord = k1ord; v.skipMargin = skipMargin; v.minLen = minLen;
        v.tags = useTags ? tags : nullptr; v.tagCur = swk::lap_tag(laps); v.tagPrev = laps ? swk::lap_tag(laps - 1) : 0u;
        return v;
    }
    // event pairs are recycled: creating events by the hundred makes the runtime grow its signal pool now and
    // then, which can stall the calling thread in the middle of a measurement
    std::vector<ProfEvent> idle;
    void account(const ProfEvent &e) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, e.a, e.b);
        profMs[e.fam] += ms; profN[e.fam]++;
        idle.push_back(e);
    }
    void mark(int fam, bool begin, hipStream_t on = nullptr) {
        if (!prof || !((profMask >> fam) & 1u)) return;
        if (!on) on = stream;
        if (begin) {
            while (events.size() > 1 && hipEventQuery(events.front().b) == hipSuccess) {   // harvest what has finished
                account(events.front());
                events.pop_front();
            }
            ProfEvent e;
            if (!idle.empty()) { e = idle.back(); idle.pop_back(); }
            else { (void) hipEventCreate(&e.a); (void) hipEventCreate(&e.b); }
            e.fam = fam;
            (void) hipEventRecord(e.a, on);
            events.push_back(e);
        } else
            (void) hipEventRecord(events.back().b, on);
    }
    void drain_events() {
        for (auto &e : events) {
            (void) hipEventSynchronize(e.b);
            account(e);
        }
        events.clear();
    }
};

namespace {

// initParams, SlidingWindowSparseEMMatcher.cpp:74-104
void init_params(swsem *h) {
    const int L = h->L;
    if (L > 110) h->K = 56;
    else if (L > 62) h->K = 44;
    else if (L > 53) h->K = 40;
    else if (L > 46) h->K = 36;
    else if (L > 42) h->K = 32;
    else if (L > 32) h->K = 28;
    else h->K = (L / 4 - 1) * 4;
    const int KmmL = (L / 4 - 1) * 4;
    if (KmmL < h->K) h->K = KmmL;
    uint8_t i = 24;
    do {
        h->hash_size = ((uint32_t) 1) << (i++);
    } while (i <= 31 && h->hash_size < h->maxRefLength / (uint64_t) h->k1);
    h->mask = h->hash_size - 1;
    h->fpBits = 10;                        // of the K-mer's second hash (swsem_device.h, fp_step); 22 bits are left for the epoch
}

void build_lut(uint8_t *lut) {
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t) i;
    lut[127] = 0;   // the reference's table constructor stops at i < CHAR_MAX, utils/helper.cpp:321-322
    const char *from = "AaCcGgTtNnUuYyRrKkMmBbDdHhVvWwSs";
    const char *to = "TTGGCCAANNAARRYYMMKKVVHHDDBBSSWW";
    for (int i = 0; from[i]; i++) lut[(uint8_t) from[i]] = (uint8_t) to[i];
}

// processIgnoreCollisionsRef (.cpp:146-171): derive the two sample sets and launch one insertion.
int insert_samples(swsem *h, const uint8_t *src = nullptr, uint64_t lo = 0, uint64_t hi = 0) {
    const int64_t STEP = (int64_t) h->k1 * 128;
    const int64_t E = h->pos1 - h->K;
    const int64_t S = (int64_t) h->samplingPos;
    uint64_t nMain = 0;
    if (S < E - STEP) {
        const int64_t blocks = ((E - STEP) - S + STEP - 1) / STEP;
        nMain = (uint64_t) blocks * 128;
    }
    const int64_t T = h->k1 + ((E - 1) / STEP) * STEP;
    uint64_t nTail = 0;
    if (T < E + 1) nTail = (uint64_t) ((E - T) / h->k1 + 1);
    const uint64_t total = nMain + nTail;
    if (total && h->deferInserts) {
        InsertPiece pc;
        pc.S = (uint64_t) S; pc.nMain = nMain; pc.T = (uint64_t) T; pc.nTail = nTail; pc.epoch = h->epoch; pc.tag = swk::lap_tag(h->laps);
        pc.src = src; pc.lo = lo; pc.hi = hi;                       // reference positions [lo, hi) will hold src[0 .. hi - lo)
        h->pendingPieces.push_back(pc);
    } else if (total) {
        h->mark(SWSEM_K_INSERT, true);
        k_insert<<<dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, h->stream>>>(
            h->ref, h->ht, (uint64_t) S, nMain, (uint64_t) T, nTail, h->k1, h->k1ord, h->K, h->mask, h->epoch, h->fpBits, h->tags, swk::lap_tag(h->laps));
        h->mark(SWSEM_K_INSERT, false);
        HIPCHK(hipGetLastError());
    }
    h->epoch += 2;
    if (h->epoch >= (1u << (32 - h->fpBits)) - 2) return fail(SWSEM_EINVAL, "too many load phases for the table's epoch field");
    h->samplingPos = (uint64_t) (T + (int64_t) nTail * h->k1);
    return SWSEM_OK;
}

int flush_copies(swsem *h);
int download(swsem *h, void *dstPinned, const void *srcDev, size_t bytes, hipStream_t st);

// The byte automata of an emission's second phase (sizes .. write), the copy of its results and its completion event, on
// the second stream. Queued with the emission they become ready when the finalize ends — the moment the next batch's resolve
// does — and whichever was dealt the wave slots first ran at the other's expense: the resolve took 2.8 ms instead of 2.1
// when it lost (steps of 3.1 and 3.9 ms, the slow kind in 40 % of the steps on the 4.35e9-byte sizing). So they are kept
// back until the next resolve kernel has been launched (run_batch; `gated`: the caller has made the second stream wait for
// the event recorded just before that launch), or until somebody needs the emission's results.
int run_phase2b(swsem *h, swsem::EmitSlot &E, bool gated) {
    if (!E.deferred2b) return SWSEM_OK;
    E.deferred2b = false;
    if (E.waitFin2b && !gated) {                                     // behind the finalize's last kernel (everything queued so far)
        if (!h->evFin) HIPCHK(hipEventCreateWithFlags(&h->evFin, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->evFin, h->stream));
        HIPCHK(hipStreamWaitEvent(h->stream2, h->evFin, 0));
    }
    HIPCHK(hipStreamWaitEvent(h->stream2, E.evMetaDone, 0));          // the pairing kernels' results (their own stream)
    const EmitView &v = E.v2b;
    const dim3 grid2(E.grid2b);
    const int n = E.n2b;
    h->mark(SWSEM_K_EMIT2, true, h->stream2);
    // (few chunks: a quarter of the tasks per wave, four times the waves — swsem_emit.hip, chunk_task)
    const bool thin = E.grid2b <= h->emitThinMax;
    if (thin) k_emit_sizes<4><<<grid2, dim3(1024), 0, h->stream2>>>(v, E.dECg.p);
    else k_emit_sizes<1><<<grid2, dim3(256), 0, h->stream2>>>(v, E.dECg.p);
    k_emit_place_sums<<<grid2, dim3(CH), 0, h->stream2>>>(v, E.dECg.p);
    k_emit_place_scan<<<dim3(n), dim3(CH), 0, h->stream2>>>(v, E.dECg.p);
    k_emit_packoffs<<<1, dim3(CH), 0, h->stream2>>>(v);
    k_emit_place_final<<<grid2, dim3(CH), 0, h->stream2>>>(v, E.dECg.p);
    if (thin) k_emit_write<4><<<grid2, dim3(1024), 0, h->stream2>>>(v, E.dECg.p);
    else k_emit_write<1><<<grid2, dim3(256), 0, h->stream2>>>(v, E.dECg.p);
    k_emit_copy_long<<<dim3(512), dim3(256), 0, h->stream2>>>(v);
    h->mark(SWSEM_K_EMIT2, false, h->stream2);
    HIPCHK(hipGetLastError());
    int r2;
    if ((r2 = flush_copies(h)) || (r2 = download(h, E.pinE, E.dEOut.p, (size_t) n * sizeof(EmitOut), h->stream2)) || (r2 = flush_copies(h))) return r2;
    HIPCHK(hipEventRecord(E.evDone, h->stream2));
    E.donePending = false;
    return SWSEM_OK;
}

// An emission whose second phase is still running reads reference bytes next to its matches. It never reads inside
// its own lock window [loading position it started at, its matching-lock position): candidates there were refused
// at match time (.cpp:212-220), pairs do not span the lock (TextMatchers.h:46-50), the right extension stops at the
// loading position and the left one at the lock (ENC.cpp:318-335, :379-384) — that window exists so that the
// reference's loader can write while its workers read, and loadRef never writes beyond it (.cpp:408-414). So a write
// that stays inside an emission's window runs beside it, wrap or not (tests/test_gpu_lock_window.py fills the window
// with garbage and emits again: same bytes); any other write — the separator that replaces the last loaded byte at
// the window's end, a write outside the window of an older emission, contigs without a lock — waits for the emission.
int ref_write_guard(swsem *h, uint64_t firstByte, uint64_t lastByte) {
    for (auto &E : h->slot) {
        if (!E.outstanding || E.refGuarded) continue;
        bool inside = false;
        if (E.lockMin != UINT64_MAX && firstByte >= REF_SHIFT && lastByte >= firstByte) {
            if (E.lockMin > E.emitPos1)                              // window [emitPos1, lockMin)
                inside = h->laps == E.emitLaps && firstByte >= E.emitPos1 && lastByte < E.lockMin;
            else                                                     // it wraps: [emitPos1, end) and, a lap later, [1, lockMin)
                inside = (h->laps == E.emitLaps && firstByte >= E.emitPos1) || (h->laps == E.emitLaps + 1 && lastByte < E.lockMin);
        }
        const bool appendOnly = h->laps == 0 && firstByte >= E.emitPos1;   // nothing was ever written there: nothing to read
        if (!inside && !appendOnly) {
            if (E.deferred2b) { int d = run_phase2b(h, E, false); if (d) return d; }   // (its automata had not been queued yet)
            if (E.donePending) return SWSEM_ESPEC;                   // (only while a speculative finalize is being queued: it is given up)
            HIPCHK(hipStreamWaitEvent(h->stream, E.evDone, 0));
            E.refGuarded = true;
        }
    }
    return SWSEM_OK;
}

// private loadRef, .cpp:402-437, on a device-resident text
int load_pieces(swsem *h, const uint8_t *text, uint64_t len, bool rc, bool addSep, int sep) {
    while (len != 0) {
        if ((uint64_t) h->pos1 == h->maxRefLength && h->swEnd != h->maxRefLength) {
            h->laps++;
            h->ePrev = h->eCur; h->eCur = h->epoch;                    // (entries of older laps are told by their epochs, ht_value)
            h->pos1 = REF_SHIFT;
            h->samplingPos = REF_SHIFT;
        }
        const uint64_t tmpEnd = h->swEnd;
        uint64_t tmpLength = len;
        const uint64_t tmpMax = tmpEnd < (uint64_t) h->pos1 ? h->maxRefLength : tmpEnd;
        if ((uint64_t) h->pos1 + tmpLength > tmpMax) tmpLength = tmpMax - (uint64_t) h->pos1;
        // (a loader that stands at the window's end with that byte already the separator writes nothing — every further target of a
        // round whose loads have filled the window: no emission has to be waited for then; on collections that load every contig
        // with its reverse complement that is the second half of most rounds)
        const bool sepWrite = addSep && (uint64_t) h->pos1 + tmpLength == h->swEnd && !(tmpLength == 0 && h->sep_end_done(sep));
        if (tmpLength || sepWrite) {
            // bytes this step writes: the copy, and the separator at the window's end when the copy reaches it
            // (a window that wraps has its end BELOW the loading position: only a loader that stands AT the end writes the byte before it)
            const uint64_t first = (uint64_t) h->pos1 == h->swEnd ? h->swEnd - 1 : (uint64_t) h->pos1;
            const uint64_t last = tmpLength ? (uint64_t) h->pos1 + tmpLength - 1 : first;
            int g = ref_write_guard(h, first, last);
            if (g) return g;
        }
        if (tmpLength && !rc && h->deferInserts) {                  // (nothing is launched here: no profiling bracket)
            CopyPiece cp; cp.dst = (uint64_t) h->pos1; cp.src = text; cp.len = tmpLength;
            h->pendingCopies.push_back(cp);
        } else if (tmpLength) {
            h->mark(SWSEM_K_LOAD, true);
            if (rc) {
                const uint64_t thr = (tmpLength + 3) / 4;
                const unsigned blocks = (unsigned) std::min<uint64_t>((thr + 255) / 256, 8192);
                k_load_rc<<<dim3(blocks), dim3(256), 0, h->stream>>>(text + len - tmpLength, h->ref + h->pos1, tmpLength, h->lut);
            } else
                HIPCHK(hipMemcpyAsync(h->ref + h->pos1, text, tmpLength, hipMemcpyDeviceToDevice, h->stream));
            h->mark(SWSEM_K_LOAD, false);
        }
        if (sepWrite) {
            // (with the window full every target of a round comes by here and through loadSeparator's same case: once is enough)
            if (h->deferInserts) { BytePiece bp; bp.off = h->swEnd - 1; bp.val = (uint64_t) (uint8_t) sep; h->pendingBytes.push_back(bp); }
            else k_set_byte<<<1, 1, 0, h->stream>>>(h->ref + h->swEnd - 1, (uint8_t) sep);
            h->sep_end_set((int64_t) h->swEnd, sep);
        }
        const bool viaTable = tmpLength && !rc && h->deferInserts;
        const uint64_t copiedTo = (uint64_t) h->pos1;
        h->pos1 += (int64_t) tmpLength;
        int r = viaTable ? insert_samples(h, text, copiedTo, copiedTo + tmpLength) : insert_samples(h);
        if (r) return r;
        text += rc ? 0 : tmpLength;
        if ((uint64_t) h->pos1 == tmpEnd) h->droppedBytes += len - tmpLength;
        len = (uint64_t) h->pos1 == tmpEnd ? 0 : len - tmpLength;
    }
    HIPCHK(hipGetLastError());
    return SWSEM_OK;
}

// the handle's side streams out of the device's pool (see SidePool): by what shares a dispatch pipe with its main stream
int deal_streams(swsem *h) {
    SidePool *P = side_pool(h->device, h->prioLow, h->prioHigh);
    if (!P) return fail(SWSEM_EHIP, "cannot make the side streams");
    constexpr int NC = SidePool::NC;
    const char *e = getenv("SWSEM_STREAM_CALIB");
    const bool measure = !(e && atoi(e) == 0);
    bool mainHits[NC] = {};
    for (int i = 0; measure && i < NC; i++)
        if (P->label[i] == i) mainHits[i] = streams_collide(h->stream, P->cand[i], P->wgs);
    (void) hipGetLastError();
    int chosen[4] = {-1, -1, -1, -1};                                // stream2, load, aux, up
    // penalties: sharing a pipe with the main stream, with a role that is busy at the same time, being another role's stream
    const int cls[4] = {0, 2, 1, 1};
    const int clash[4][4] = {{0, 0, 0, 0}, {60, 0, 0, 0}, {60, 10, 0, 0}, {60, 0, 5, 0}};   // [role][earlier role]
    const int withMain[4] = {100, 100, 100, 20};
    for (int r = 0; r < 4; r++) {
        int best = -1, bestCost = 1 << 30;
        for (int i = 0; i < NC; i++) {
            if (P->cls[i] != cls[r]) continue;
            int cost = mainHits[P->label[i]] ? withMain[r] : 0;
            for (int q = 0; q < r; q++) {
                if (chosen[q] == i) cost += 1000;
                else if (P->label[chosen[q]] == P->label[i]) cost += clash[r][q];
            }
            if (cost < bestCost) { bestCost = cost; best = i; }
        }
        if (best < 0) return fail(SWSEM_EHIP, "no side stream of the class wanted");
        chosen[r] = best;
    }
    h->stream2 = P->cand[chosen[0]]; h->streamLoad = P->cand[chosen[1]]; h->streamAux = P->cand[chosen[2]]; h->streamUp = P->cand[chosen[3]];
    if (getenv("SWSEM_STREAM_DEBUG")) {
        fprintf(stderr, "swsem side streams: labels");
        for (int i = 0; i < NC; i++) fprintf(stderr, " %d%s", P->label[i], mainHits[P->label[i]] ? "*" : "");
        fprintf(stderr, " (* shares the main stream's pipe); stream2 %d, load %d, aux %d, up %d\n", chosen[0], chosen[1], chosen[2], chosen[3]);
    }
    return SWSEM_OK;
}

int flush_inserts(swsem *h, const uint32_t *gate = nullptr);
int prepare_inserts(swsem *h, hipStream_t upStream);
int launch_inserts(swsem *h, const uint32_t *gate);
}  // namespace
// (defined below, after run_batch's helpers)
namespace {
// Everything collected while deferInserts was set: all copies in one launch, the separator bytes in one (in program
// order; no copy of a round lands on a byte written by an earlier separator of the same round), then every insertion
// phase in one launch. The tables travel in a single upload. Two steps:
//   prepare_inserts      the host tables (pinned) and their upload on `upStream` (the speculative finalize's travel with the
//                        emission's own tables while the chains still run, not between pass 1 and the copies)
//   launch_inserts       the launches on the main stream: copies (on their stream), separators, the insertion, the samples at
//                        the pieces' edges
int prepare_inserts(swsem *h, hipStream_t upStream) {
    swsem::PreparedInserts &P = h->prep;
    P = swsem::PreparedInserts();
    const size_t np = h->pendingPieces.size(), nc = h->pendingCopies.size(), nb = h->pendingBytes.size();
    P.np = np; P.nc = nc; P.nb = nb;
    if (!np && !nc && !nb) return SWSEM_OK;
    // (every copy and byte collected here went through ref_write_guard when it was collected — load_pieces,
    // swsem_load_separator — with the lap count of that moment; one test of the whole span would take the two halves of a
    // round that wraps for a write across the whole buffer and give the speculative finalize up once per lap)
    constexpr uint64_t CHUNK = 256 * 16;                 // bytes per copy block
    // Insertion beside the copies: a sample whose K bytes all come out of its own piece's copy is hashed from the copy's
    // source (k_insert_multi<true>), while the copies run on a stream of their own; what is left — windows that reach into
    // the previous text or a separator, pieces without a copy — is listed as runs of its own and inserted from the buffer
    // once the copies have landed. A byte a separator of this flush overwrites is not "the copy's" any more.
    std::vector<InsertPiece> &edge = h->edgePieces;
    edge.clear();
    bool beside = h->insertBeside && np && nc;
    if (beside) {
        for (size_t i = 0; i < nc && beside; i++)        // (two copies of one flush over the same bytes: only in order)
            for (size_t j = i + 1; j < nc && beside; j++)
                beside = h->pendingCopies[i].dst + h->pendingCopies[i].len <= h->pendingCopies[j].dst || h->pendingCopies[j].dst + h->pendingCopies[j].len <= h->pendingCopies[i].dst;
    }
    if (beside) {
        const int64_t k1 = h->k1, K = h->K;
        for (auto &pc : h->pendingPieces) {
            if (pc.src)
                for (auto &b : h->pendingBytes)
                    if (b.off >= pc.lo && b.off < pc.hi) { if (b.off - pc.lo < pc.hi - b.off) { pc.src += b.off + 1 - pc.lo; pc.lo = b.off + 1; } else pc.hi = b.off; }
            // samples p = base + t*k1, t < n, that k_insert_multi<true> does not take: p < lo or p + K > hi
            auto runs = [&](uint64_t base, uint64_t n, uint32_t epoch) {
                if (!n) return;
                int64_t a = 0, b = -1;                              // taken from the source: t in [a, b]
                if (pc.src && (int64_t) pc.hi - K >= (int64_t) base) {
                    a = (int64_t) pc.lo > (int64_t) base ? ((int64_t) pc.lo - (int64_t) base + k1 - 1) / k1 : 0;
                    b = std::min<int64_t>((int64_t) n - 1, ((int64_t) pc.hi - K - (int64_t) base) / k1);
                }
                auto push = [&](int64_t t0, int64_t t1) {           // [t0, t1)
                    if (t1 <= t0) return;
                    InsertPiece e = {};
                    e.S = base + (uint64_t) t0 * (uint64_t) k1; e.nMain = (uint64_t) (t1 - t0); e.epoch = epoch; e.tag = pc.tag;
                    edge.push_back(e);
                };
                if (b < a) push(0, (int64_t) n);
                else { push(0, a); push(b + 1, (int64_t) n); }
            };
            runs(pc.S, pc.nMain, pc.epoch);
            runs(pc.T, pc.nTail, pc.epoch + 1);
        }
    } else
        for (auto &pc : h->pendingPieces) pc.src = nullptr;
    const size_t ne = edge.size();
    const size_t wPieces = np * (sizeof(InsertPiece) / 8), wCopies = nc * (sizeof(CopyPiece) / 8), wBytes = nb * (sizeof(BytePiece) / 8), wEdge = ne * (sizeof(InsertPiece) / 8);
    // host table: a member (two alternating ones), so the upload needs no wait before returning
    swsem::HostTab &ht = h->hostTables[h->hostTableSel ^= 1];
    const size_t words = wPieces + (np + 1) + wCopies + (nc + 1) + wBytes + wEdge + (ne + 1);
    if (ht.pending) { HIPCHK(hipEventSynchronize(ht.ev)); ht.pending = false; }
    if (ht.cap < words) {
        if (ht.p) HIPCHK(hipHostFree(ht.p));
        ht.p = nullptr; ht.cap = 0;
        // generous: (re)allocating pinned memory synchronises the whole device, it must not recur in steady state
        const size_t want = std::max<size_t>(2 * words, 1 << 16);
        if (hipHostMalloc((void **) &ht.p, want * 8, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return fail(SWSEM_ENOMEM, "cannot pin %zu B of host memory", want * 8);
        ht.cap = want;
    }
    if (!ht.ev) HIPCHK(hipEventCreateWithFlags(&ht.ev, hipEventDisableTiming));
    uint64_t *tPieces = ht.p, *tFirst = tPieces + wPieces, *tCopies = tFirst + np + 1, *tCFirst = tCopies + wCopies, *tBytes = tCFirst + nc + 1,
             *tEdge = tBytes + wBytes, *tEFirst = tEdge + wEdge;
    if (np) memcpy(tPieces, h->pendingPieces.data(), np * sizeof(InsertPiece));
    tFirst[0] = 0;
    for (size_t i = 0; i < np; i++) tFirst[i + 1] = tFirst[i] + h->pendingPieces[i].nMain + h->pendingPieces[i].nTail;
    if (nc) memcpy(tCopies, h->pendingCopies.data(), nc * sizeof(CopyPiece));
    tCFirst[0] = 0;
    for (size_t i = 0; i < nc; i++) tCFirst[i + 1] = tCFirst[i] + (h->pendingCopies[i].len + CHUNK - 1) / CHUNK;
    if (nb) memcpy(tBytes, h->pendingBytes.data(), nb * sizeof(BytePiece));
    if (ne) memcpy(tEdge, edge.data(), ne * sizeof(InsertPiece));
    tEFirst[0] = 0;
    for (size_t i = 0; i < ne; i++) tEFirst[i + 1] = tEFirst[i] + edge[i].nMain;
    int r;
    if ((r = h->dTables.reserve(std::max<size_t>(2 * words, 1 << 16)))) return r;   // regrowing = hipFree = a device-wide wait
    // (a kernel reading the pinned table: a runtime copy here costs an engine switch in the middle of the main stream)
    k_upload<<<dim3((unsigned) ((words * 8 + 4095) / 4096)), dim3(256), 0, upStream>>>((uint8_t *) h->dTables.p, (const uint8_t *) ht.p, words * 8);
    HIPCHK(hipEventRecord(ht.ev, upStream));
    ht.pending = true;
    const uint64_t *d = h->dTables.p;
    P.beside = beside; P.ne = ne;
    P.nSamples = tFirst[np]; P.nEdge = tEFirst[ne]; P.copyBlocks = tCFirst[nc];
    P.dPieces = (const InsertPiece *) (d + (tPieces - ht.p)); P.dFirst = d + (tFirst - ht.p);
    P.dCopies = (const CopyPiece *) (d + (tCopies - ht.p)); P.dCFirst = d + (tCFirst - ht.p);
    P.dBytes = (const BytePiece *) (d + (tBytes - ht.p));
    P.dEdge = (const InsertPiece *) (d + (tEdge - ht.p)); P.dEFirst = d + (tEFirst - ht.p);
    h->pendingPieces.clear(); h->pendingCopies.clear(); h->pendingBytes.clear();
    HIPCHK(hipGetLastError());
    return SWSEM_OK;
}

int launch_inserts(swsem *h, const uint32_t *gate) {
    hipStream_t sV = h->stream;
    swsem::PreparedInserts &P = h->prep;
    const size_t np = P.np, nc = P.nc, nb = P.nb, ne = P.ne;
    if (!np && !nc && !nb) return SWSEM_OK;
    hipStream_t cs = sV;                                           // the copies' stream
    if (P.beside) {
        // (its own priority class: the runtime deals the streams of one class over a handful of hardware queues, and a copy
        // that lands on the queue of the emission's second phase runs behind 2 ms of its kernels — seen in a kernel trace)
        cs = h->streamLoad;
        HIPCHK(hipEventRecord(h->evLoadFork, sV));                   // (behind the tables, the gate and every wait the writes were given)
        HIPCHK(hipStreamWaitEvent(cs, h->evLoadFork, 0));
    }
    if (nc) {
        h->mark(SWSEM_K_LOAD, true, cs);
        k_copy_multi<<<dim3((unsigned) std::min<uint64_t>(P.copyBlocks, h->copyWgs)), dim3(256), 0, cs>>>(h->ref, P.dCopies, P.dCFirst, (int) nc, gate);
        h->mark(SWSEM_K_LOAD, false, cs);
    }
    if (nb) k_set_bytes<<<1, 1, 0, cs>>>(h->ref, P.dBytes, (int) nb, gate);
    if (P.beside) HIPCHK(hipEventRecord(h->evLoadDone, cs));
    if (np && P.nSamples) {
        h->mark(SWSEM_K_INSERT, true);
        const dim3 grid((unsigned) ((P.nSamples + 255) / 256));
        if (P.beside) k_insert_multi<true><<<grid, dim3(256), 0, h->stream>>>(h->ref, h->ht, P.dPieces, P.dFirst, (int) np, h->k1, h->k1ord, h->K, h->mask, h->fpBits, gate, h->tags);
        else k_insert_multi<false><<<grid, dim3(256), 0, h->stream>>>(h->ref, h->ht, P.dPieces, P.dFirst, (int) np, h->k1, h->k1ord, h->K, h->mask, h->fpBits, gate, h->tags);
        h->mark(SWSEM_K_INSERT, false);
    }
    if (P.beside) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->evLoadDone, 0));
        if (ne && P.nEdge)
            k_insert_multi<false><<<dim3((unsigned) ((P.nEdge + 255) / 256)), dim3(256), 0, h->stream>>>(
                h->ref, h->ht, P.dEdge, P.dEFirst, (int) ne, h->k1, h->k1ord, h->K, h->mask, h->fpBits, gate, h->tags);
    }
    HIPCHK(hipGetLastError());
    return SWSEM_OK;
}

int flush_inserts(swsem *h, const uint32_t *gate) {
    int r = prepare_inserts(h, h->stream);
    return r ? r : launch_inserts(h, gate);
}

// Small copies between host and device go through pinned host memory that is mapped into the device's address
// space and are made by a kernel (the runtime's own small copies can block the calling thread for milliseconds on a
// side stream, and switch engines in the middle of the main one). They are staged and leave in one launch per
// flush_copies(): up to CopySegs::MAX segments, zero-fills among them.
int flush_copies(swsem *h) {
    CopySegs &sg = h->segs;
    if (sg.n == 0) return SWSEM_OK;
    const uint32_t blocks = sg.first[sg.n];
    k_copy_segs<<<dim3(blocks), dim3(256), 0, h->segStream>>>(sg);
    sg.n = 0;
    HIPCHK(hipGetLastError());
    return SWSEM_OK;
}
int stage_copy(swsem *h, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (!bytes) return SWSEM_OK;
    CopySegs &sg = h->segs;
    if (sg.n && (h->segStream != st || sg.n == CopySegs::MAX)) { int r = flush_copies(h); if (r) return r; }
    if (sg.n == 0) { h->segStream = st; sg.first[0] = 0; }
    sg.dst[sg.n] = (uint8_t *) dst; sg.src[sg.n] = (const uint8_t *) src; sg.bytes[sg.n] = bytes;
    sg.first[sg.n + 1] = sg.first[sg.n] + (uint32_t) ((bytes + 4095) / 4096);
    sg.n++;
    return SWSEM_OK;
}
// device results -> pinned host memory; visible to the host once an event recorded behind the flush has completed
int download(swsem *h, void *dstPinned, const void *srcDev, size_t bytes, hipStream_t st) { return stage_copy(h, dstPinned, srcDev, bytes, st); }
int zero_dev(swsem *h, void *dst, size_t bytes, hipStream_t st) { return stage_copy(h, dst, nullptr, bytes, st); }

// host data -> device through the pinned ring (staged: flush_copies() launches)
int upload(swsem *h, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (!bytes) return SWSEM_OK;
    const size_t need = (bytes + 255) & ~(size_t) 255;
    if (need * 4 > h->ringCap) {                       // (re)allocation: rare, and the only place that waits
        { int r = flush_copies(h); if (r) return r; }
        HIPCHK(hipDeviceSynchronize());
        if (h->ring) HIPCHK(hipHostFree(h->ring));
        h->ring = nullptr; h->ringCap = 0; h->ringAt = 0;
        const size_t want = std::max<size_t>(need * 8, 8u << 20);
        if (hipHostMalloc((void **) &h->ring, want, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return fail(SWSEM_ENOMEM, "cannot pin %zu B of host memory", want);
        h->ringCap = want;
    }
    if (h->ringAt + need > h->ringCap) {               // wrap: everything staged a lap ago has long been copied, but make sure
        { int r = flush_copies(h); if (r) return r; }
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipStreamSynchronize(h->stream2));
        if (h->stream3) HIPCHK(hipStreamSynchronize(h->s3()));
        HIPCHK(hipStreamSynchronize(h->streamUp));
        HIPCHK(hipStreamSynchronize(h->streamAux));
        h->ringAt = 0;
    }
    uint8_t *slot = h->ring + h->ringAt;
    h->ringAt += need;
    memcpy(slot, src, bytes);
    return stage_copy(h, dst, slot, bytes, st);
}

// Launch order of the resolve blocks (h->contigs filled in). The genomes of a collection resemble each other, so the
// blocks that scan the same offsets of a round's contigs look up the same buckets and compare against the same
// reference windows. Workgroups are dealt round-robin over the eight XCDs (slot s -> XCD s mod 8, MI355X_MICROARCH.md
// "Workgroup dispatch": observed, for speed only) and each XCD has its own L2: such a group of blocks is given
// consecutive slots of ONE XCD, so one of them fetches a sector from HBM and the others find it in that L2. Groups
// larger than 64 blocks are cut (a batch of many one-block contigs must still spread over the chip), every piece goes
// to the XCD with the shortest list so far, and the lists are padded to one length with empty slots.
bool build_resolve_order(swsem *h, uint32_t rblocks, uint32_t per) {
    std::vector<uint32_t> key;
    key.reserve(h->contigs.size() + 3);
    key.push_back(rblocks); key.push_back(per);
    for (auto &cg : h->contigs) key.push_back(cg.nrb);
    if (key == h->rbOrderKey && !h->rbOrderHost.empty()) return false;       // same shape as the last batch: the device table stands
    h->rbOrderKey.swap(key);
    std::vector<uint32_t> &order = h->rbOrderHost;
    order.clear();
    uint32_t maxNrb = 0;
    for (auto &cg : h->contigs) maxNrb = std::max(maxNrb, cg.nrb);
    // contigs by descending block count: the contigs that still have a block at offset b are a prefix
    std::vector<uint32_t> byLen(h->contigs.size());
    for (uint32_t c = 0; c < byLen.size(); c++) byLen[c] = c;
    std::stable_sort(byLen.begin(), byLen.end(), [&](uint32_t a, uint32_t b) { return h->contigs[a].nrb > h->contigs[b].nrb; });
    std::vector<uint32_t> lists[8];                                          // wave slots (per block ids each) of every XCD
    std::vector<uint32_t> grp;
    size_t alive = byLen.size();
    const size_t piece = 64;                                                 // blocks of one offset kept together on an XCD
    for (uint32_t b = 0; b < maxNrb; b++) {
        while (alive && h->contigs[byLen[alive - 1]].nrb <= b) alive--;
        grp.assign(byLen.begin(), byLen.begin() + alive);
        if (!std::is_sorted(grp.begin(), grp.end())) std::sort(grp.begin(), grp.end());   // contig order inside a group
        for (size_t i = 0; i < grp.size(); i += piece) {
            int best = 0;
            for (int x = 1; x < 8; x++) if (lists[x].size() < lists[best].size()) best = x;
            const size_t e = std::min(grp.size(), i + piece);
            for (size_t k = i; k < e; k++) lists[best].push_back(h->contigs[grp[k]].rb0 + b);
            while (lists[best].size() % per) lists[best].push_back(0xFFFFFFFFu);          // the last wave of the piece may run fewer chains
        }
    }
    size_t len = 0;
    for (auto &l : lists) len = std::max(len, l.size() / per);
    order.assign(len * 8 * per, 0xFFFFFFFFu);
    for (int x = 0; x < 8; x++)
        for (size_t j = 0; j < lists[x].size() / per; j++)
            for (uint32_t k = 0; k < per; k++) order[(j * 8 + x) * per + k] = lists[x][j * per + k];
    return true;
}

int run_batch(swsem *h, const uint8_t *qdev, const uint64_t *offsets, int n, uint32_t minLen, const uint64_t *lockPos) {
    if (n <= 0) return fail(SWSEM_EINVAL, "empty batch");
    if (minLen < (uint32_t) h->K)   // SlidingWindowSparseEMMatcher.cpp:480-483
        return fail(SWSEM_EINVAL, "Minimal matching length cannot be smaller than K (%u < %d)", minLen, h->K);
    h->batchValid = false;
    h->matchCount.clear();
    h->minLen = minLen;
    h->contigs.assign(n, Contig());
    std::vector<uint32_t> &rbContig = h->rbContigHost;   // uploaded asynchronously
    rbContig.clear();
    uint64_t matchRows = 0, bases = 0, probes = 0;
    uint32_t rblocks = 0;
    {
        // Block chains are latency-bound and a launch lasts as long as its slowest wave: the blocks are sized so
        // that all of them are resident at once and there are as many as that allows.
        // Fewer, longer blocks leave wave slots empty; more of them run in two generations and lengthen the
        // sequential stitch. At least 2048 positions: a small batch (one target of the sequential schedule) fills few wave slots
        // whatever the block length, and then short chains are what is fast (the warm-up positions per 2048 of its own).
        uint64_t allUnits = 0;                          // in units of RBU positions
        for (int c = 0; c < n; c++) {
            const uint64_t len = offsets[c + 1] - offsets[c];
            allUnits += len >= (uint64_t) h->K ? (len - h->K + 1 + RBU - 1) / RBU : 0;
        }
        h->chainsPerWave = (h->simt && !h->seqResolve && h->K <= K_MAX4) ? (uint32_t) GC : 1u;
        const uint64_t waves = h->chainsPerWave > 1 ? (uint64_t) h->waveSlots / RESOLVE_WAVES_PER_SIMD * RESOLVE4_WAVES_PER_SIMD : h->waveSlots;
        const uint64_t slots = std::max<uint64_t>(1, waves * h->chainsPerWave * h->slotPercent / 100);
        h->rb = h->rbFixed ? h->rbFixed : (uint32_t) std::min<uint64_t>(65536 / RBU, std::max<uint64_t>(h->rbMin, (allUnits + slots - 1) / slots));   // (a small batch — one target of the sequential schedule — runs short chains: it is their length that takes the time)
    }
    for (int c = 0; c < n; c++) {
        Contig &cg = h->contigs[c];
        cg.qoff = offsets[c];
        cg.n = offsets[c + 1] - offsets[c];
        // query positions are 32-bit signed in the resolve automaton (and uint32 in processMatches, MBGC_Encoder.cpp:145)
        if (cg.n >= (1ull << 31) - (1ull << 20)) return fail(SWSEM_EINVAL, "contig %d longer than 2^31 - 2^20 bytes", c);
        cg.lock = lockPos ? lockPos[c] : UINT64_MAX;
        const uint64_t npos = cg.n >= (uint64_t) h->K ? cg.n - h->K + 1 : 0;
        probes += npos;
        cg.matchBase = matchRows;
        matchRows += cg.n / minLen + 2;
        cg.rb0 = rblocks;
        cg.nrb = (uint32_t) ((npos + (uint64_t) h->rb * RBU - 1) / ((uint64_t) h->rb * RBU));
        for (uint32_t t = 0; t < cg.nrb; t++) rbContig.push_back((uint32_t) c);
        rblocks += cg.nrb;
        bases += cg.n;
    }
    int r;
    if ((r = h->dContigs.reserve(n))) return r;
    if ((r = h->dMatchCount.reserve(n))) return r;
    if ((r = h->dStats.reserve(8))) return r;
    if ((r = h->dMatches.reserve(matchRows))) return r;
    if ((r = upload(h, h->dContigs.p, h->contigs.data(), n * sizeof(Contig), h->stream))) return r;
    if ((r = h->dRbContig.reserve(std::max<uint32_t>(rblocks, 1)))) return r;
    uint32_t rslots = 0;
    if (rblocks) {
        const uint32_t *had = h->dRbOrder.p;
        const bool fresh = build_resolve_order(h, rblocks, h->chainsPerWave);
        rslots = (uint32_t) (h->rbOrderHost.size() / h->chainsPerWave);
        if ((r = h->dRbOrder.reserve(h->rbOrderHost.size() + h->rbOrderHost.size() / 4 + 64))) return r;
        if ((r = upload(h, h->dRbContig.p, rbContig.data(), rblocks * sizeof(uint32_t), h->stream))) return r;
        if ((fresh || had != h->dRbOrder.p) && (r = upload(h, h->dRbOrder.p, h->rbOrderHost.data(), h->rbOrderHost.size() * sizeof(uint32_t), h->stream)))
            return r;
    }
    if ((r = zero_dev(h, h->dStats.p, 8 * sizeof(unsigned long long), h->stream)) || (r = flush_copies(h))) return r;
    const RefView v = h->view();
    const bool wrapped = v.fpCheck == 2;                // kernels instantiated with / without the lap epochs (ht_value)
    if (h->seqResolve || rblocks == 0)
        for (auto &E : h->slot) if ((r = run_phase2b(h, E, false))) return r;
    // "this batch begins here": everything queued on the main stream before it has finished when this event has (the uploads
    // of the emission that follows wait for nothing else, emit_begin_impl)
    HIPCHK(hipEventRecord(h->evRoundTop, h->stream));
    h->roundTopFresh = true;
    if (h->seqResolve || rblocks == 0) {
        h->mark(SWSEM_K_RESOLVE, true);
        if (wrapped) k_resolve_seq<true><<<dim3(n), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dMatches.p, h->dMatchCount.p);
        else k_resolve_seq<false><<<dim3(n), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dMatches.p, h->dMatchCount.p);
        h->mark(SWSEM_K_RESOLVE, false);
        HIPCHK(hipEventRecord(h->evStitched, h->stream));
    } else {
        // rows a block chain can hold: disjoint matches, each containing the K-mer of a distinct visited hit
        const uint32_t cap = (uint32_t) ((h->rb * RBU + OVERLAP_MAX + h->K) / h->K + 8);
        h->batchBlocks = rblocks;
        if ((r = h->dRegions.reserve((size_t) rblocks * cap))) return r;
        if ((r = h->dReplay.reserve((size_t) n * cap))) return r;
        if ((r = h->dRecs.reserve(rblocks))) return r;
        if ((r = h->dFast.reserve(rblocks))) return r;
        if ((r = h->dSegStart.reserve(rblocks))) return r;
        if ((r = h->dKeepN.reserve(rblocks))) return r;
        if ((r = h->dDstOff.reserve(rblocks))) return r;
        if ((r = h->dPrev.reserve(rblocks))) return r;
        // An emission whose byte automata wait to be queued (run_phase2b): they are handed to the second stream AFTER the
        // resolve kernel has been handed to the first, behind an event recorded just before it — whatever hardware queues
        // the two streams share, the resolve is dealt its wave slots first.
        if (h->metaPending) { HIPCHK(hipStreamWaitEvent(h->stream, h->evMeta, 0)); h->metaPending = false; }
        bool anyDeferred = false;
        for (auto &E : h->slot) anyDeferred |= E.deferred2b;
        if (anyDeferred) {
            if (!h->evFin) HIPCHK(hipEventCreateWithFlags(&h->evFin, hipEventDisableTiming));
            HIPCHK(hipEventRecord(h->evFin, h->stream));
        }
        h->mark(SWSEM_K_RESOLVE, true);
        if (h->chainsPerWave == (uint32_t) GC) {
            if (wrapped) k_resolve_blocks4<true><<<dim3(rslots), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dRbContig.p, h->dRbOrder.p, h->dRegions.p, cap, h->rb, h->dRecs.p, h->overlap);
            else k_resolve_blocks4<false><<<dim3(rslots), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dRbContig.p, h->dRbOrder.p, h->dRegions.p, cap, h->rb, h->dRecs.p, h->overlap);
        } else {
            if (wrapped) k_resolve_blocks<true><<<dim3(rslots), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dRbContig.p, h->dRbOrder.p, h->dRegions.p, cap, h->rb, h->dRecs.p, h->overlap);
            else k_resolve_blocks<false><<<dim3(rslots), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dRbContig.p, h->dRbOrder.p, h->dRegions.p, cap, h->rb, h->dRecs.p, h->overlap);
        }
        h->mark(SWSEM_K_RESOLVE, false);
        if (anyDeferred) {
            HIPCHK(hipStreamWaitEvent(h->stream2, h->evFin, 0));
            for (auto &E : h->slot) if ((r = run_phase2b(h, E, true))) return r;
        }
        h->mark(SWSEM_K_STITCH, true);
        k_stitch_pre<<<dim3((rblocks + 255) / 256), dim3(256), 0, h->stream>>>(h->dContigs.p, h->dRbContig.p, h->dRecs.p, h->rb, rblocks, h->dFast.p);
        if (wrapped) k_stitch<true><<<dim3(n), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dRegions.p, h->dReplay.p, cap, h->rb, h->dRecs.p, h->dFast.p, h->dSegStart.p,
                                                                          h->dKeepN.p, h->dPrev.p, h->dDstOff.p, h->dMatchCount.p, h->dStats.p);
        else k_stitch<false><<<dim3(n), dim3(WAVE), 0, h->stream>>>(v, qdev, h->dContigs.p, h->dRegions.p, h->dReplay.p, cap, h->rb, h->dRecs.p, h->dFast.p, h->dSegStart.p,
                                                                    h->dKeepN.p, h->dPrev.p, h->dDstOff.p, h->dMatchCount.p, h->dStats.p);
        HIPCHK(hipEventRecord(h->evStitched, h->stream));             // nothing reads the table any more (the stitch's replays were the last)
        k_gather<<<dim3(rblocks), dim3(WAVE), 0, h->stream>>>(h->dContigs.p, h->dRbContig.p, h->dRegions.p, cap, h->dSegStart.p,
                                                            h->dKeepN.p, h->dDstOff.p, h->dMatches.p);
        h->mark(SWSEM_K_STITCH, false);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->evMatched, h->stream));
    h->qdev = qdev;
    h->stats[0] = bases;
    h->hostProbes = probes;
    h->batchValid = true;
    return SWSEM_OK;
}

// pinned landing zone for everything a batch hands back to the host: {stats[8] | match counts | emit results}:
// the copies queue up back to back and one wait serves them all
int pin_reserve(swsem *h, size_t bytes) {
    if (h->pinCap >= bytes) return SWSEM_OK;
    if (h->pin) HIPCHK(hipHostFree(h->pin));
    h->pin = nullptr; h->pinCap = 0;
    const size_t want = std::max<size_t>(2 * bytes, 1 << 20);          // see flush_inserts: no regrowth in steady state
    if (hipHostMalloc((void **) &h->pin, want, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return fail(SWSEM_ENOMEM, "cannot pin %zu B of host memory", want);
    h->pinCap = want;
    return SWSEM_OK;
}

// queues the copies of the match counts and statistics (no wait)
int queue_counts(swsem *h, size_t extraBytes, hipStream_t st = nullptr) {
    if (!st) st = h->stream;
    const size_t n = h->contigs.size();
    const size_t countsAt = 64, extraAt = (countsAt + n * sizeof(uint32_t) + 63) & ~(size_t) 63;
    int r = pin_reserve(h, extraAt + extraBytes);
    if (r) return r;
    if ((r = download(h, h->pin, h->dStats.p, 8 * sizeof(unsigned long long), st)) ||
        (r = download(h, h->pin + countsAt, h->dMatchCount.p, n * sizeof(uint32_t), st)))
        return r;
    h->pinExtraAt = extraAt;
    return SWSEM_OK;                                   // (staged: the caller adds what it wants beside them and flushes)
}

// after the wait: pinned block -> host state
void take_counts(swsem *h) {
    const size_t n = h->contigs.size();
    const unsigned long long *st = (const unsigned long long *) h->pin;
    h->matchCount.assign((const uint32_t *) (h->pin + 64), (const uint32_t *) (h->pin + 64) + n);
    h->stats[1] = h->hostProbes; h->stats[2] = st[2]; h->stats[5] = st[3];
    h->stitchDiag[0] += st[3]; h->stitchDiag[1] += st[5]; h->stitchDiag[2] += st[6]; h->stitchDiag[3] += st[7];
    if (!h->overlapFixed && h->batchBlocks >= 2048) {              // (a batch large enough for the share to mean something)
        const uint64_t replayed = st[3];
        if (replayed * 400 > h->batchBlocks) h->overlap = std::min<uint32_t>((uint32_t) OVERLAP_MAX, h->overlap + 128);         // > 0.25 %: longer
        else if (replayed * 2000 < h->batchBlocks) h->overlap = std::max<uint32_t>(640u, h->overlap - 128);          // < 0.05 %: shorter
        h->batchBlocks = 0;                                          // (these counts are taken once per batch)
    }
    uint64_t tot = 0;
    for (size_t c = 0; c < n; c++) tot += h->matchCount[c];
    h->stats[3] = tot;
}

int fetch_counts(swsem *h) {
    int r = queue_counts(h, 0);
    if (r || (r = flush_copies(h))) return r;
    HIPCHK(hipStreamSynchronize(h->stream));
    take_counts(h);
    return SWSEM_OK;
}


// waits for the second phase of the emission in slot si, if one is running, and takes its results
int end_slot(swsem *h, int si) {
    swsem::EmitSlot &E = h->slot[si];
    if (!E.outstanding) return SWSEM_OK;
    HIPCHK(hipSetDevice(h->device));
    { int d = run_phase2b(h, E, false); if (d) return d; }            // (nobody launched a resolve since: nothing to wait for)
    HIPCHK(hipEventSynchronize(E.evDone));
    E.outstanding = false;
    const int n = E.emitN;
    E.eout.assign((const EmitOut *) E.pinE, (const EmitOut *) E.pinE + n);
    uint64_t tot = 0;
    E.hostStreamOff.assign((size_t) n * SWSEM_NSTREAMS, 0);
    for (int k = 0; k < n; k++)
        for (int st = 0; st < SWSEM_NSTREAMS; st++) {
            if (E.eout[k].unmatchedChars == UINT64_MAX) E.eout[k].size[st] = 0;
            E.hostStreamOff[(size_t) k * SWSEM_NSTREAMS + st] = tot;       // == packBase on the device
            tot += E.eout[k].size[st];
        }
    E.packedBytes = tot;
    E.hostStreamsValid = false;
    if (h->emitHostCopy) {
        E.hostAt ^= 1;                                                 // (the other half may still be read through the views of this slot's last emission)
        if (E.hostStreams().reserve(tot + 1)) return fail(SWSEM_ENOMEM, "cannot pin %llu B of host memory", (unsigned long long) tot);
        if (tot) HIPCHK(hipMemcpyAsync(E.hostStreams().data(), E.dEArena.p, tot, hipMemcpyDeviceToHost, h->s3()));
        HIPCHK(hipStreamSynchronize(h->s3()));
        E.hostStreamsValid = true;
    }
    return SWSEM_OK;
}

}  // namespace

extern "C" {

const char *swsem_last_error(void) { return g_err.c_str(); }

int swsem_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int swsem_device_numa_node(int device) {
    char bus[64] = {0};
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count ||
        hipDeviceGetPCIBusId(bus, (int) sizeof bus, device) != hipSuccess) {
        (void) hipGetLastError();                                      // (a failed query must not be the next launch's "last error")
        return -1;
    }
    for (char *c = bus; *c; c++) *c = (char) tolower((unsigned char) *c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

int swsem_create(swsem_t **out, uint64_t maxRefLength, int L, int k1, int k2, int skipMargin, int device) {
    *out = nullptr;
    if (k1 <= 0) return fail(SWSEM_EINVAL, "s - reference sampling step - should be a positive integer.");   // MBGC_Params.h:593-597
    if (k2 != 1) return fail(SWSEM_EINVAL, "k2 = %d unsupported (MBGC always uses k2 = 1, MGMP_Params.h:202)", k2);
    if (L < 16) return fail(SWSEM_EINVAL, "Error: Minimal matching length too short!");
    if (maxRefLength < 64 || (maxRefLength >> __builtin_ctz((unsigned) k1)) >= (1ull << 32))
        return fail(SWSEM_EINVAL, "reference length limit %llu out of range", (unsigned long long) maxRefLength);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev)
        return fail(SWSEM_ENODEV, "no HIP device %d (the HIP path has no CPU fallback)", device);
    HIPCHK(hipSetDevice(device));
    swsem *h = new swsem();
    h->device = device;
    h->maxRefLength = maxRefLength;
    h->L = L; h->k1 = k1; h->k2 = k2; h->skipMargin = skipMargin;
    h->k1ord = __builtin_ctz((unsigned) k1);
    h->swEnd = maxRefLength;
    h->swSize = maxRefLength / SW_WIDTH_FACTOR;
    init_params(h);
    // an even k1: SlidingWindowExpSparseEMMatcher — entries hold position >> ctz(k1), sampling starts at k1 (.cpp:494-503); an odd
    // one: the base class (MGMP.cpp:170-176) — htEncodePos / htDecodePos are the identity (.h:74-76: k1ord = 0 here), sampling
    // starts at REF_SHIFT (.h:78), which is also where a wrap puts it back, so the off-grid samples of the Exp variant do not occur
    h->samplingPos = (k1 % 2) ? REF_SHIFT : (uint64_t) k1;
    if (h->k1ord == 0) h->useTags = false;                  // (a lap tag per sampling slot would be two bytes per reference byte: stale entries are visited instead, same results)
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(SWSEM_EHIP, "hipStreamCreate failed"); }
    h->ownStream = true;
    {
        int least = 0, greatest = 0;                        // numerically: least >= greatest
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) { h->prioLow = least; h->prioHigh = greatest; }
    }
    if (hipEventCreateWithFlags(&h->evP1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->slot[0].evDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->slot[1].evDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->slot[0].evMetaDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->slot[1].evMetaDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evMatched, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evStitched, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evTables, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evRoundTop, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evLoadFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evLoadDone, hipEventDisableTiming) != hipSuccess) { delete h; return fail(SWSEM_EHIP, "hipEventCreate failed"); }
    { int d = deal_streams(h); if (d) { delete h; return d; } }
    if (const char *e = getenv("SWSEM_COPY_WGS")) h->copyWgs = (uint64_t) std::max(1, atoi(e));
    if (const char *e = getenv("SWSEM_RESOLVE")) h->seqResolve = strcmp(e, "seq") == 0;
    if (const char *e = getenv("SWSEM_PROF_FAMS")) h->profMask = (uint32_t) strtoul(e, nullptr, 0);
    if (const char *e = getenv("SWSEM_CHAINS")) h->simt = atoi(e) != 1;
    if (const char *e = getenv("SWSEM_INSERT_BESIDE")) h->insertBeside = atoi(e) != 0;
    if (const char *e = getenv("SWSEM_META_WARM")) h->metaWarm = std::min(swk::MWARM, std::max(0, atoi(e)));
    if (const char *e = getenv("SWSEM_OVERLAP")) { int x = atoi(e); if (x >= 0 && x <= OVERLAP_MAX) h->overlap = h->overlapFixed = (uint32_t) std::max(1, x); }
    if (const char *e = getenv("SWSEM_RB")) { int x = atoi(e); if (x >= 1 && x <= 256) h->rbFixed = (uint32_t) x * (1024 / RBU); }   // (in units of 1024 positions)
    if (const char *e = getenv("SWSEM_EMIT_THIN_MAX")) h->emitThinMax = (uint32_t) std::max(0, atoi(e));
    if (const char *e = getenv("SWSEM_RB_MIN")) { int x = atoi(e); if (x >= RBU && x <= 65536) h->rbMin = (uint32_t) x / RBU; }
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) h->waveSlots = (uint32_t) pr.multiProcessorCount * 4u * RESOLVE_WAVES_PER_SIMD; }
    const size_t nSlots = (size_t) ((maxRefLength + REF_SLACK) >> h->k1ord) + 2;
    if (const char *e = getenv("SWSEM_LAP_TAGS")) h->useTags = h->useTags && atoi(e) != 0;
    if (hipMalloc((void **) &h->ref, maxRefLength + REF_SLACK) != hipSuccess ||
        (h->useTags && hipMalloc((void **) &h->tags, nSlots * sizeof(uint16_t)) != hipSuccess) ||
        hipMalloc((void **) &h->ht, (size_t) h->hash_size * sizeof(ht_entry)) != hipSuccess ||
        hipMalloc((void **) &h->lut, 256) != hipSuccess) {
        swsem_destroy(h);
        return fail(SWSEM_ENOMEM, "cannot allocate %llu B reference + %llu B hash table in HBM",
                    (unsigned long long) maxRefLength, (unsigned long long) h->hash_size * 8ull);
    }
    uint8_t lut[256];
    build_lut(lut);
    HIPCHK(hipMemcpy(h->lut, lut, 256, hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(h->ht, 0, (size_t) h->hash_size * sizeof(ht_entry), h->stream));
    if (h->tags) HIPCHK(hipMemsetAsync(h->tags, 0, nSlots * sizeof(uint16_t), h->stream));
    // start1[0] = 0 (.cpp:335); the rest of the buffer is written before it is ever read, the slack
    // past the end is zeroed because the reference's own reads run a few bytes over (:224, ENC:337)
    // the whole buffer starts out as zeros (the reference reads — harmlessly — bytes it has not loaded yet, e.g. the one at the
    // loading position in extendMatchRight; left as allocated they would be whatever an earlier process had there)
    HIPCHK(hipMemsetAsync(h->ref, 0, maxRefLength, h->stream));
    HIPCHK(hipMemsetAsync(h->ref + maxRefLength, 0, REF_SLACK, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = h;
    return SWSEM_OK;
}

void swsem_destroy(swsem_t *h) {
    if (!h) return;
    (void) hipSetDevice(h->device);
    for (auto &E : h->slot) E.deferred2b = false;                    // (automata that were never queued: nobody wants their bytes any more)
    if (h->stream) (void) hipStreamSynchronize(h->stream);
    if (h->stream2) (void) hipStreamSynchronize(h->stream2);
    if (h->stream3) (void) hipStreamSynchronize(h->stream3);
    if (h->streamUp) (void) hipStreamSynchronize(h->streamUp);
    if (h->streamAux) (void) hipStreamSynchronize(h->streamAux);
    h->drain_events();
    if (getenv("SWSEM_DEBUG_STATS"))
        fprintf(stderr, "swsem stitch: blocks replayed %llu, accepted in runs %llu, tested one by one %llu, jumped over %llu\n",
                (unsigned long long) h->stitchDiag[0], (unsigned long long) h->stitchDiag[1], (unsigned long long) h->stitchDiag[2], (unsigned long long) h->stitchDiag[3]);
    if (getenv("SWSEM_DEBUG_STATS")) {                               // diagnostics: the pairing chain's counters of this handle
        uint64_t t[8];
        if (swsem_debug_emit_stats(h, t) == SWSEM_OK && (t[4] | t[5] | t[6] | t[7]))
            fprintf(stderr, "swsem pairing chain: foreign-boundary steps %llu, blocks not accepted %llu, groups replayed %llu, blocks given up %llu\n",
                    (unsigned long long) t[4], (unsigned long long) t[5], (unsigned long long) t[6], (unsigned long long) t[7]);
    }
    if (h->ref) (void) hipFree(h->ref);
    if (h->tags) (void) hipFree(h->tags);
    if (h->ht) (void) hipFree(h->ht);
    if (h->lut) (void) hipFree(h->lut);
    h->stage.release(); h->dContigs.release(); 
    h->dMatchCount.release(); h->dMatches.release(); h->dStats.release();
    h->dRegions.release(); h->dReplay.release(); h->dRecs.release(); h->dFast.release(); h->dSegStart.release(); h->dKeepN.release(); h->dDstOff.release();
    h->dPrev.release(); h->dRbContig.release(); h->dRbOrder.release(); h->dDecode.release(); h->dJobs.release(); h->dDecRecs.release(); h->dDecPlan.release(); h->dDecAux.release();
    for (auto &E : h->slot) E.release();
    h->dTables.release(); h->dGate.release(); h->dPred.release();
    for (hipEvent_t e : {h->evStitched, h->evTables, h->evRoundTop}) if (e) (void) hipEventDestroy(e);
    if (h->pin) { (void) hipHostFree(h->pin); h->pin = nullptr; h->pinCap = 0; }
    if (h->ring) { (void) hipHostFree(h->ring); h->ring = nullptr; h->ringCap = 0; }
    for (auto &t : h->hostTables) { if (t.p) (void) hipHostFree(t.p); if (t.ev) (void) hipEventDestroy(t.ev); t = swsem::HostTab(); }
    if (h->stream2) (void) hipStreamSynchronize(h->stream2);          // (the side streams are the pool's: never destroyed)
    if (h->streamLoad) (void) hipStreamSynchronize(h->streamLoad);
    if (h->evLoadFork) (void) hipEventDestroy(h->evLoadFork);
    if (h->evLoadDone) (void) hipEventDestroy(h->evLoadDone);
    h->drain_events();
    for (auto &e : h->idle) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    h->idle.clear();
    if (h->stream3) { (void) hipStreamSynchronize(h->stream3); (void) hipStreamDestroy(h->stream3); }
    if (h->evMatched) (void) hipEventDestroy(h->evMatched);
    if (h->evP1) (void) hipEventDestroy(h->evP1);
    if (h->evFin) (void) hipEventDestroy(h->evFin);
    if (h->evMeta) (void) hipEventDestroy(h->evMeta);
    if (h->ownStream && h->stream) (void) hipStreamDestroy(h->stream);
    delete h;
}

int swsem_set_stream(swsem_t *h, void *s) {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->ownStream) (void) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t) s;
    h->ownStream = false;
    return deal_streams(h);                                           // (another main stream, maybe another pipe)
}
int swsem_synchronize(swsem_t *h) { HIPCHK(hipStreamSynchronize(h->stream)); return SWSEM_OK; }

void swsem_disable_sliding_window(swsem_t *h) { h->swSize = 0; h->swEnd = h->circular ? 0 : h->maxRefLength; }
void swsem_set_sliding_window_size(swsem_t *h, int f) { h->swSize = h->maxRefLength / (uint64_t) (uint8_t) f; }
void swsem_disable_circular_buffer(swsem_t *h) { h->circular = false; h->swEnd = h->maxRefLength; }
uint64_t swsem_get_ref_length(const swsem_t *h) { return h->refLength(); }
uint64_t swsem_get_loading_position(const swsem_t *h) { return (uint64_t) h->pos1; }
uint64_t swsem_get_loaded_ref_length(const swsem_t *h) {
    return (uint64_t) h->laps * (h->maxRefLength - REF_SHIFT) + ((uint64_t) h->pos1 - REF_SHIFT);
}
uint64_t swsem_get_max_ref_length(const swsem_t *h) { return h->maxRefLength; }
uint64_t swsem_get_dropped_bytes(const swsem_t *h) { return h->droppedBytes; }
uint64_t swsem_get_sliding_window_size(const swsem_t *h) { return h->circular ? h->swSize : 0; }
void swsem_set_position(swsem_t *h, uint64_t p, int laps) { h->pos1 = (int64_t) p; h->laps = laps; h->pristine = false; }
int swsem_get_K(const swsem_t *h) { return h->K; }
uint32_t swsem_get_hash_size(const swsem_t *h) { return h->hash_size; }

// acquireWorkerMatchingLockPos, .cpp:361-378
uint64_t swsem_acquire_lock(swsem_t *h) {
    if (h->swSize == 0 || !h->circular) return h->swEnd;
    uint64_t w = (uint64_t) h->pos1 + h->swSize;
    if (h->laps || w > h->maxRefLength) {
        if (w > h->maxRefLength) w -= h->maxRefLength - REF_SHIFT;
    } else
        w = h->maxRefLength;
    if (h->locks.empty()) h->swEnd = w;
    h->locks.push_back(w);
    return w;
}

// releaseWorkerMatchingLockPos, .cpp:380-400
int swsem_release_lock(swsem_t *h, uint64_t v) {
    if (h->swSize == 0 || !h->circular) return SWSEM_OK;
    size_t i = 0;
    while (i < h->locks.size() && h->locks[i] != v) i++;
    if (i == h->locks.size()) return fail(SWSEM_ELOCK, "ERROR: Invalid worker lock value (%llu)", (unsigned long long) v);
    if (i == 0) {
        do {
            h->locks.pop_front();
        } while (!h->locks.empty() && h->locks.front() == UINT64_MAX);
        if (!h->locks.empty()) h->swEnd = h->locks.front();
    } else
        h->locks[i] = UINT64_MAX;
    return SWSEM_OK;
}

int swsem_load_ref_dev(swsem_t *h, const uint8_t *t, uint64_t len, int loadRC, int addSep, int sep) {
    HIPCHK(hipSetDevice(h->device));
    int r = load_pieces(h, t, len, false, addSep != 0, sep);
    if (r) return r;
    if (loadRC) r = load_pieces(h, t, len, true, addSep != 0, sep);
    return r;
}

int swsem_load_ref(swsem_t *h, const uint8_t *t, uint64_t len, int loadRC, int addSep, int sep) {
    HIPCHK(hipSetDevice(h->device));
    if (len == 0) return SWSEM_OK;
    int r = h->stage.reserve(len + 64);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(h->stage.p, t, len, hipMemcpyHostToDevice, h->stream));
    r = swsem_load_ref_dev(h, h->stage.p, len, loadRC, addSep, sep);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(h->stream));   // the staging buffer is reused by the next call
    return SWSEM_OK;
}

// loadSeparator, .cpp:439-451
int swsem_load_separator(swsem_t *h, int sep) {
    HIPCHK(hipSetDevice(h->device));
    if ((uint64_t) h->pos1 == h->maxRefLength && h->swEnd != h->maxRefLength) {
        h->laps++;
        h->ePrev = h->eCur; h->eCur = h->epoch;
        h->pos1 = REF_SHIFT;
        h->samplingPos = REF_SHIFT;
    }
    if ((uint64_t) h->pos1 == h->maxRefLength) return SWSEM_OK;
    if ((uint64_t) h->pos1 == h->swEnd && h->sep_end_done(sep)) return SWSEM_OK;    // that byte already is this separator
    { const uint64_t at = (uint64_t) h->pos1 == h->swEnd ? (uint64_t) h->pos1 - 1 : (uint64_t) h->pos1; int g = ref_write_guard(h, at, at); if (g) return g; }
    if ((uint64_t) h->pos1 == h->swEnd) {
        // this overwrites the last byte already loaded: insertion phases still pending hashed it as it was, and so was
        // the one sample whose K-mer ends there, if it has been inserted: its entry stops being trusted (k_mark_stale)
        if (h->specMode) return SWSEM_ESPEC;                      // an ungated write in the middle: give the speculation up
        if (h->deferInserts) { int r = flush_inserts(h); if (r) return r; }
        if (h->pos1 >= (int64_t) h->K + REF_SHIFT)
            k_mark_stale<<<1, 1, 0, h->stream>>>(h->ref, h->ht, (uint64_t) (h->pos1 - h->K), h->K, h->k1ord, h->mask, h->fpBits, h->tags);
        k_set_byte<<<1, 1, 0, h->stream>>>(h->ref + h->pos1 - 1, (uint8_t) sep);
        h->sep_end_set(h->pos1, sep);
    } else if (h->deferInserts) {
        BytePiece bp; bp.off = (uint64_t) h->pos1++; bp.val = (uint64_t) (uint8_t) sep;
        h->pendingBytes.push_back(bp);
    } else
        k_set_byte<<<1, 1, 0, h->stream>>>(h->ref + h->pos1++, (uint8_t) sep);
    HIPCHK(hipGetLastError());
    return SWSEM_OK;
}

// finalizeParallelProcessingOfTarget for n targets in order (MGMP.cpp:440-457, MBGC_Encoder.cpp:557-562):
// loadRef of the target's extension, the lazy-mode region separator, release of its lock position.
// loadedAfter[i] = getLoadedRefLength() after target i (what the encoder appends to refExtLoadedPosArr).
static int finalize_impl(swsem_t *h, int n, const uint8_t *const *ext_dev, const uint64_t *ext_len, int addSep, int sep,
                         int lazySeparator, const uint64_t *lockPos, uint64_t *loadedAfter, const uint32_t *gate, bool planOnly = false) {
    HIPCHK(hipSetDevice(h->device));
    // all byte writes of the round first (copies, region separators), then every insertion phase in one
    // launch: hashing a window needs its bytes — including a separator written by a later step — in place
    h->deferInserts = true;
    int r = SWSEM_OK;
    for (int i = 0; i < n && !r; i++) {
        if (ext_len[i]) r = load_pieces(h, ext_dev[i], ext_len[i], false, addSep != 0, sep);
        if (!r && lazySeparator) r = swsem_load_separator(h, sep);
        if (!r && loadedAfter) loadedAfter[i] = swsem_get_loaded_ref_length(h);
        if (!r && lockPos) r = swsem_release_lock(h, lockPos[i]);
    }
    h->deferInserts = false;
    if (r == SWSEM_ESPEC) { h->pendingPieces.clear(); h->pendingCopies.clear(); h->pendingBytes.clear(); return r; }
    // planOnly (the speculative finalize): the host's bookkeeping is done and the launches are listed; the caller queues them
    // (flush_inserts) once every replica's verdict has been reduced into the gate — and knows by now whether THIS replica can
    if (planOnly) return r;
    const int r2 = flush_inserts(h, gate);
    return r ? r : r2;
}

int swsem_finalize_targets(swsem_t *h, int n, const uint8_t *const *ext_dev, const uint64_t *ext_len, int addSep, int sep,
                           int lazySeparator, const uint64_t *lockPos, uint64_t *loadedAfter) {
    return finalize_impl(h, n, ext_dev, ext_len, addSep, sep, lazySeparator, lockPos, loadedAfter, nullptr);
}

int swsem_match_batch_dev(swsem_t *h, const uint8_t *q, const uint64_t *offsets, int n, uint32_t minLen, const uint64_t *lockPos) {
    HIPCHK(hipSetDevice(h->device));
    return run_batch(h, q, offsets, n, minLen, lockPos);
}

int swsem_batch_counts(swsem_t *h, uint64_t *nm) {
    if (!h->batchValid) return fail(SWSEM_EINVAL, "no batch results");
    if (h->matchCount.size() != h->contigs.size()) { int r = fetch_counts(h); if (r) return r; }
    for (size_t c = 0; c < h->contigs.size(); c++) nm[c] = h->matchCount[c];
    return SWSEM_OK;
}

int swsem_batch_matches(swsem_t *h, int c, swsem_match_t *out, uint64_t cap) {
    if (!h->batchValid || c < 0 || c >= (int) h->contigs.size()) return fail(SWSEM_EINVAL, "no such contig in the batch");
    if (h->matchCount.size() != h->contigs.size()) { int r = fetch_counts(h); if (r) return r; }
    const uint64_t n = std::min<uint64_t>(cap, h->matchCount[c]);
    if (n) HIPCHK(hipMemcpy(out, h->dMatches.p + h->contigs[c].matchBase, n * sizeof(Match), hipMemcpyDeviceToHost));
    return SWSEM_OK;
}

int swsem_batch_fingerprint(swsem_t *h, uint64_t *fp, uint64_t *tot, uint64_t *len) {
    if (!h->batchValid) return fail(SWSEM_EINVAL, "no batch results");
    k_fingerprint<<<1, 1, 0, h->stream>>>(h->dContigs.p, (int) h->contigs.size(), h->dMatches.p, h->dMatchCount.p, h->dStats.p + 4);
    unsigned long long o[3];
    HIPCHK(hipMemcpyAsync(o, h->dStats.p + 4, sizeof o, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *fp = o[0]; *tot = o[1]; *len = o[2];
    h->stats[3] = o[1]; h->stats[4] = o[2];
    return SWSEM_OK;
}

int swsem_match(swsem_t *h, const uint8_t *query, uint64_t len, uint32_t minLen, uint64_t lockPos,
                const swsem_match_t **matches, uint64_t *nmatches) {
    HIPCHK(hipSetDevice(h->device));
    *matches = nullptr; *nmatches = 0;
    int r = swsem_emit_batch_end(h);                   // an emission still running reads the staged query
    if (r) return r;
    if ((r = h->stage.reserve(len + 64))) return r;
    if (len) HIPCHK(hipMemcpyAsync(h->stage.p, query, len, hipMemcpyHostToDevice, h->stream));
    const uint64_t offs[2] = {0, len};
    if ((r = run_batch(h, h->stage.p, offs, 1, minLen, &lockPos))) return r;
    if ((r = fetch_counts(h))) return r;
    h->hostMatches.resize(h->matchCount[0]);
    if (h->matchCount[0])
        HIPCHK(hipMemcpy(h->hostMatches.data(), h->dMatches.p, h->matchCount[0] * sizeof(Match), hipMemcpyDeviceToHost));
    *matches = h->hostMatches.data();
    *nmatches = h->matchCount[0];
    return SWSEM_OK;
}

// plain device-memory helpers so that host code above this ABI needs no HIP headers
int swsem_dev_malloc(swsem_t *h, uint64_t bytes, void **out) {
    HIPCHK(hipSetDevice(h->device));
    *out = nullptr;
    if (hipMalloc(out, bytes ? bytes : 1) != hipSuccess) return fail(SWSEM_ENOMEM, "device allocation of %llu bytes failed", (unsigned long long) bytes);
    return SWSEM_OK;
}
int swsem_dev_free(swsem_t *h, void *p) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (p) HIPCHK(hipFree(p));
    return SWSEM_OK;
}
int swsem_dev_upload(swsem_t *h, void *dst_dev, const void *src, uint64_t bytes) {
    HIPCHK(hipSetDevice(h->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SWSEM_OK;
}
int swsem_dev_download(swsem_t *h, void *dst, const void *src_dev, uint64_t bytes) {
    HIPCHK(hipSetDevice(h->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SWSEM_OK;
}
int swsem_dev_copy(swsem_t *h, void *dst_dev, const void *src_dev, uint64_t bytes) {
    HIPCHK(hipSetDevice(h->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, h->stream));
    return SWSEM_OK;
}

// PgHelpers::upperReverseComplement on device buffers (utils/helper.cpp:405-410): lets the caller build a
// target's extension string "contig + RC(contig)" (MGMP.cpp:389-398) without leaving HBM.
int swsem_revcomp_dev(swsem_t *h, const uint8_t *src_dev, uint64_t n, uint8_t *dst_dev) {
    HIPCHK(hipSetDevice(h->device));
    if (n == 0) return SWSEM_OK;
    const uint64_t thr = (n + 3) / 4;
    const unsigned blocks = (unsigned) std::min<uint64_t>((thr + 255) / 256, 8192);
    h->mark(SWSEM_K_LOAD, true);
    k_load_rc<<<dim3(blocks), dim3(256), 0, h->stream>>>(src_dev, dst_dev, n, h->lut);
    h->mark(SWSEM_K_LOAD, false);
    HIPCHK(hipGetLastError());
    return SWSEM_OK;
}

int swsem_debug_copy_ref(swsem_t *h, uint64_t from, uint64_t n, uint8_t *out) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->ref + from, n, hipMemcpyDeviceToHost));
    return SWSEM_OK;
}

int swsem_debug_write_ref(swsem_t *h, uint64_t from, uint64_t n, const uint8_t *in) {
    if (from + n > h->maxRefLength) return fail(SWSEM_EINVAL, "swsem_debug_write_ref: beyond the buffer");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->stream2));
    HIPCHK(hipMemcpy(h->ref + from, in, n, hipMemcpyHostToDevice));
    {   // the bytes of these slots changed without a sample: their tags say so
        const uint64_t s0 = from >= (uint64_t) h->K ? (from - h->K + 1) >> h->k1ord : 0, s1 = (from + n + ((1ull << h->k1ord) - 1)) >> h->k1ord;
        if (h->tags && s1 > s0) HIPCHK(hipMemset(h->tags + s0, 0, (s1 - s0) * sizeof(uint16_t)));
    }
    return SWSEM_OK;
}

int swsem_debug_copy_ht(swsem_t *h, uint32_t *out) {
    DevBuf<uint32_t> tmp;
    int r = tmp.reserve(h->hash_size);
    if (r) return r;
    k_ht_low_words<<<dim3((h->hash_size + 255) / 256), dim3(256), 0, h->stream>>>(h->ht, tmp.p, h->hash_size, h->fpBits);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, tmp.p, (size_t) h->hash_size * 4, hipMemcpyDeviceToHost));
    tmp.release();
    return SWSEM_OK;
}

// diagnostics: per resolve block of the last batch {ticks, candidates visited, rows on the stack}
int swsem_debug_block_times(swsem_t *h, uint64_t *out, uint64_t cap, uint64_t *n) {
    uint64_t nb = 0;
    for (auto &c : h->contigs) nb += c.nrb;
    *n = nb;
    if (nb > cap) nb = cap;
    std::vector<BlockRec> tmp(nb);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nb) HIPCHK(hipMemcpy(tmp.data(), h->dRecs.p, nb * sizeof(BlockRec), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < nb; i++) { out[3 * i] = tmp[i].cycles; out[3 * i + 1] = tmp[i].visits; out[3 * i + 2] = tmp[i].emits; }
    return SWSEM_OK;
}

// counters of the emission's pairing chain since the handle was made, summed over the emission slots:
// out[4] steps that went by an inherited boundary other than the match's own, out[5] blocks of the speculative pass that were
// not accepted, out[6] groups of 64 matches the stitch replayed, out[7] blocks given up for too many inherited boundaries
int swsem_debug_emit_stats(swsem_t *h, uint64_t out[8]) {
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < 8; k++) out[k] = 0;
    for (auto &E : h->slot) {
        if (!E.dEStat.p || !E.statZeroed) continue;
        unsigned long long t[8];
        HIPCHK(hipMemcpy(t, E.dEStat.p, sizeof t, hipMemcpyDeviceToHost));
        for (int k = 0; k < 8; k++) out[k] += t[k];
    }
    return SWSEM_OK;
}

#ifdef SWSEM_DIAG_PHASES
// diagnostics build only: phase sums of every resolve launch since the last call (g_diag), then reset
int swsem_debug_phases(swsem_t *h, uint64_t *out) {
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long z[8] = {0};
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(swk::g_diag), sizeof z));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(swk::g_diag), z, sizeof z));
    return SWSEM_OK;
}
#endif

int swsem_profile_enable(swsem_t *h, int on) {
    h->drain_events();
    h->prof = on != 0;
    memset(h->profMs, 0, sizeof h->profMs);
    memset(h->profN, 0, sizeof h->profN);
    return SWSEM_OK;
}

int swsem_profile_get(swsem_t *h, double ms[SWSEM_K_COUNT], uint64_t n[SWSEM_K_COUNT]) {
    h->drain_events();
    for (int i = 0; i < SWSEM_K_COUNT; i++) { ms[i] = h->profMs[i]; n[i] = h->profN[i]; }
    return SWSEM_OK;
}

int swsem_batch_stats(swsem_t *h, uint64_t s[6]) {
    if (!h->batchValid) return fail(SWSEM_EINVAL, "no batch results");
    for (int i = 0; i < 6; i++) s[i] = h->stats[i];
    return SWSEM_OK;
}

void swsem_emit_params_default(swsem_emit_params_t *p, int mode) {
    memset(p, 0, sizeof(*p));
    p->enableExtensionsWithMismatches = 1;
    p->mismatchesWithExclusion = 1;
    p->lazyDecompressionSupport = 1;
    p->enable40bitReference = 0;
    p->frugal64bitLenEncoding = 1;
    p->gapDepthOffsetEncoding = 64;                     // MBGC_Params.h:51
    p->gapDepthMismatchesEncoding = 2;                  // :52
    p->gapBreakingMatchMinLength = 256;                 // :53
    p->mmsMatchBonus = 50;                              // initMismatchesMatchingScoreParams, :92-97
    p->mmsMismatchPenalty = 50;
    p->mmsMismatchesScoreThreshold = 500;
    p->mmsMismatchesInitialScore = 125;
    p->allowedTargetsOutrunForDissimilarContigs = 1;    // MGMP_Params.h:60
    p->minimalLengthForDissimilarContigs = 1024;        // :64
    p->unmatchedFractionFactorTweakForDissimilarContigs = 16;   // :61
    if (mode == 0) {                                    // MBGC_Params.h:893-902
        p->allowedTargetsOutrunForDissimilarContigs = 4;
        p->unmatchedFractionFactorTweakForDissimilarContigs = 32;
        p->frugal64bitLenEncoding = 0;
    }
    if (mode == 2) {                                    // :903-906
        p->allowedTargetsOutrunForDissimilarContigs = 0;
        p->unmatchedFractionFactorTweakForDissimilarContigs = 2;
    }
}

// processMatches for `n` contigs of the last batch in one pass (the reference runs it per contig on the
// worker thread that matched it, MGMP.cpp:381). Results stay on the handle until the next emit call.
static int finalize_impl(swsem_t *h, int n, const uint8_t *const *ext_dev, const uint64_t *ext_len, int addSep, int sep,
                         int lazySeparator, const uint64_t *lockPos, uint64_t *loadedAfter, const uint32_t *gate);

static int emit_begin_impl(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                           const int *factor, const int64_t *processed, const int64_t *targetIdx,
                           const uint64_t *refExtLoadedPos, uint64_t nLoaded, const swsem_spec_finalize_t *spec, int *applied) {
    HIPCHK(hipSetDevice(h->device));
    if (applied) *applied = 0;
    const int si = h->latest ^ 1;                                     // the slot not used by the previous emission
    { int e = end_slot(h, si); if (e) return e; }                    // its scratch is about to be reused
    swsem::EmitSlot &E = h->slot[si];
    if (!h->batchValid) return fail(SWSEM_EINVAL, "swsem_emit: no match results on the handle");
    if (n <= 0) return fail(SWSEM_EINVAL, "swsem_emit: empty request");
    if (p->lazyDecompressionSupport && nLoaded == 0) return fail(SWSEM_EINVAL, "swsem_emit: lazy mode needs refExtLoadedPosArr");
    if (p->gapDepthOffsetEncoding > 64 || p->gapDepthOffsetEncoding < 0)
        return fail(SWSEM_EINVAL, "gapDepthOffsetEncoding %d out of range (MAX_GAP_DEPTH / 2)", p->gapDepthOffsetEncoding);
    int r;
    E.ecg.assign(n, EmitContig());
    std::vector<int> &which = E.ewhich;       // uploaded asynchronously: must outlive this call
    which.assign(n, 0);
    uint64_t rows = 0, arena = 0;
    E.chunkOwner.clear(); E.spanOwner.clear();
    for (int k = 0; k < n; k++) {
        const int c = contigIdx ? contigIdx[k] : k;
        if (c < 0 || c >= (int) h->contigs.size()) return fail(SWSEM_EINVAL, "swsem_emit: no contig %d in the batch", c);
        which[k] = c;
        EmitContig &e = E.ecg[k];
        const Contig &cg = h->contigs[c];
        // rows are reserved for the most matches a contig can have, so no round trip to the host is needed
        // between match-finding and emission
        const uint64_t nm = cg.n / (h->minLen ? h->minLen : 1) + 2;
        e.qoff = cg.qoff; e.n = cg.n; e.matchBase = cg.matchBase;
        e.lock = lockPos ? lockPos[k] : UINT64_MAX;
        e.scratchBase = rows;
        e.cap = (uint32_t) (nm + 2);
        rows += e.cap;
        e.chunk0 = (uint32_t) E.chunkOwner.size();
        E.chunkOwner.insert(E.chunkOwner.end(), (e.cap + CH - 1) / CH, (uint32_t) k);
        e.span0 = (uint32_t) E.spanOwner.size();
        E.spanOwner.insert(E.spanOwner.end(), (e.cap + MSPAN - 1) / MSPAN, (uint32_t) k);
        e.factor = factor ? factor[k] : 128;
        e.processed = processed ? processed[k] : 0;
        e.targetIdx = targetIdx ? targetIdx[k] : 0;
        const uint64_t szs[SWSEM_NSTREAMS] = {cg.n + nm + 16, 4 * nm + 16, nm + 16, 14 * nm + 16, nm + 16, cg.n + 2 * nm + 16};
        for (int st = 0; st < SWSEM_NSTREAMS; st++) { e.streamBase[st] = arena; arena += (szs[st] + 15) & ~15ull; }
    }
    const uint32_t chunks = (uint32_t) E.chunkOwner.size(), spans = (uint32_t) E.spanOwner.size();
    h->capN = std::max<uint64_t>(h->capN, (uint64_t) n); h->capRows = std::max(h->capRows, rows); h->capArena = std::max(h->capArena, arena);
    if (nLoaded + 1 > h->capLoaded) h->capLoaded = std::max<uint64_t>(4096, 2 * (nLoaded + 1));   // (regrowing a buffer waits for the whole device: rarely)
    h->capChunks = std::max<uint64_t>(h->capChunks, chunks);
    {
        const uint64_t N = h->capN, R = h->capRows, A = h->capArena, Cn = h->capChunks;
        if ((r = E.dECg.reserve(N)) || (r = E.dEOut.reserve(N)) || (r = E.dEWhich.reserve(N)) || (r = E.dEM.reserve(R)) ||
            (r = E.dENext0.reserve(R)) || (r = E.dERm.reserve(R)) ||
            (r = E.dEKeep.reserve(R)) || (r = E.dEMeta.reserve(R)) || (r = E.dECorr.reserve(R)) ||
            (r = E.dESz.reserve(R * 6)) || (r = E.dEOfs.reserve(R * 6)) || (r = E.dEArena.reserve(A)) || (r = E.dELoaded.reserve(h->capLoaded)) ||
            (r = E.dEStat.reserve(8)) || (r = E.dELong.reserve(LONG_COPY_CAP)) || (r = E.dELongCount.reserve(4)) || (r = E.dEPm.reserve(R)) || (r = E.dELit.reserve((size_t) Cn * (CH / WAVE))) || (r = E.dEBad.reserve(Cn)) || (r = E.dEOwner.reserve(Cn)) || (r = E.dESpanOwner.reserve(Cn)) || (r = E.dEStates.reserve((size_t) Cn * (CH / MB) * 2)) || (r = E.dEChunk.reserve((size_t) Cn * 6)) ||
            (r = E.dEPack.reserve((size_t) N * SWSEM_NSTREAMS)))
            return r;
    }
    if (!E.statZeroed) { HIPCHK(hipMemsetAsync(E.dEStat.p, 0, 8 * sizeof(unsigned long long), h->stream)); E.statZeroed = true; }
    // The emission's tables (and, below, the speculative finalize's) travel on a stream of their own that waits for nothing
    // but the point where this batch began: they land while the chains are still running, and neither they nor their
    // launch gaps sit between the stitch and the first pass. (An emission that is not the first of its batch — a retry pass
    // over some of its contigs — has no such point: its uploads are ordered behind everything queued so far.)
    hipStream_t up = h->streamUp;
    if (!h->roundTopFresh) HIPCHK(hipEventRecord(h->evRoundTop, h->stream));
    h->roundTopFresh = false;
    HIPCHK(hipStreamWaitEvent(up, h->evRoundTop, 0));
    if ((r = upload(h, E.dEOwner.p, E.chunkOwner.data(), chunks * sizeof(uint32_t), up))) return r;
    if ((r = upload(h, E.dESpanOwner.p, E.spanOwner.data(), spans * sizeof(uint32_t), up))) return r;
    if ((r = upload(h, E.dECg.p, E.ecg.data(), n * sizeof(EmitContig), up)) || (r = upload(h, E.dEWhich.p, which.data(), n * sizeof(int), up))) return r;
    E.eloaded.assign(refExtLoadedPos, refExtLoadedPos + nLoaded);
    if ((r = upload(h, E.dELoaded.p, E.eloaded.data(), nLoaded * sizeof(uint64_t), up))) return r;
    if (spec && spec->ntargets > 0) {                                // the prediction k_spec_verify checks pass 1 against
        if ((r = h->dGate.reserve(4)) || (r = h->dPred.reserve(2 * (size_t) n + 64))) return r;
        if ((r = upload(h, h->dPred.p, spec->predExt, n, up)) || (r = upload(h, h->dPred.p + n, spec->predRC, n, up))) return r;
    }
    if ((r = flush_copies(h))) return r;
    // no synchronisation here: the kernels below queue up behind match-finding while it is still running
    EmitView v;
    v.ref = h->ref; v.qbuf = h->qdev; v.matches = h->dMatches.p; v.matchCount = h->dMatchCount.p;
    v.pos1 = (uint64_t) h->pos1; v.refLength = h->refLength(); v.maxRefLength = h->maxRefLength;
    v.loaded = E.dELoaded.p; v.nLoaded = (uint32_t) nLoaded; v.p = *p;
    v.em = E.dEM.p; v.next0 = E.dENext0.p; v.removed = E.dERm.p; v.keepIdx = E.dEKeep.p;
    v.meta = E.dEMeta.p; v.corr = E.dECorr.p; v.sz = E.dESz.p; v.arena = E.dEArena.p; v.out = E.dEOut.p;
    v.packBase = E.dEPack.p;
    v.pairMask = E.dEPm.p; v.litBits = E.dELit.p; v.metaBad = E.dEBad.p;
    v.longCopies = E.dELong.p; v.longCount = E.dELongCount.p;
    v.ofs = E.dEOfs.p;
    v.chunkCnt = E.dEChunk.p;
    v.chunkOwner = E.dEOwner.p; v.spanOwner = E.dESpanOwner.p;
    v.ncontigs = (uint32_t) n;
    const dim3 grid2(chunks);
    // the emission is on the books from here on (the plan of the speculative finalize below asks ref_write_guard about it)
    E.v2b = v; E.grid2b = chunks; E.n2b = n;
    E.donePending = true;
    h->latest = si; h->selected = -1;
    E.outstanding = true; E.refGuarded = false; E.emitN = n; E.emitPos1 = (uint64_t) h->pos1; E.qdev = h->qdev; E.params = *p;
    E.emitLaps = h->laps;
    E.lockMin = UINT64_MAX;
    {
        uint64_t lm = UINT64_MAX; bool all = true;
        for (int k = 0; k < n; k++) { if (E.ecg[k].lock == UINT64_MAX) all = false; else lm = std::min(lm, E.ecg[k].lock); }
        if (all) E.lockMin = lm;
    }
    E.packedBytes = 0; E.hostStreamsValid = false;
    E.deferred2b = false;
    // Speculative finalize: the round's loadRef / loadSeparator / lock releases are worked out on the host now, under
    // the caller's prediction of every contig's extension decision, and queued behind a device-side check of that
    // prediction — so the copies start the moment pass 1 ends instead of after the host's round trip, and the table
    // insertion, which can be taken back (InsertLog), the moment the stitch has ended. The host comes to the same verdict
    // from the values pass 1 hands back and keeps or undoes its bookkeeping accordingly; when the prediction fails nothing
    // on the device has changed.
    struct { int64_t pos1, sepEndPos; int laps, sepEndLaps, sepEndVal; uint64_t samplingPos, swEnd; uint32_t epoch, eCur, ePrev; bool pristine; std::deque<uint64_t> locks; uint64_t dropped; } snap;
    auto restore = [&]() {
        h->pos1 = snap.pos1; h->laps = snap.laps; h->samplingPos = snap.samplingPos; h->swEnd = snap.swEnd; h->epoch = snap.epoch; h->eCur = snap.eCur; h->ePrev = snap.ePrev;
        h->sepEndPos = snap.sepEndPos; h->sepEndLaps = snap.sepEndLaps; h->sepEndVal = snap.sepEndVal;
        h->pristine = snap.pristine; h->locks = snap.locks; h->droppedBytes = snap.dropped;
    };
    bool queued = false, exchanged = false, planned = false;
    uint32_t *gate = nullptr;
    if (spec && spec->ntargets > 0) {
        snap.pos1 = h->pos1; snap.laps = h->laps; snap.samplingPos = h->samplingPos; snap.swEnd = h->swEnd; snap.epoch = h->epoch; snap.eCur = h->eCur; snap.ePrev = h->ePrev; snap.sepEndPos = h->sepEndPos; snap.sepEndLaps = h->sepEndLaps; snap.sepEndVal = h->sepEndVal;
        snap.pristine = h->pristine; snap.locks = h->locks; snap.dropped = h->droppedBytes;
        gate = spec->gate_dev ? spec->gate_dev : h->dGate.p;
        // The host's half first (lock window, piece schedule, separators: load_pieces and its callees, nothing launched): it can
        // find that this finalize cannot be queued behind a gate at all (SWSEM_ESPEC: a write that would have to wait for an older
        // emission, a separator over an already hashed byte). With several replicas that has to be known BEFORE the verdicts are
        // reduced: a replica that cannot apply the round must say so in the reduction, or the others apply it without it.
        h->specMode = true;
        r = finalize_impl(h, spec->ntargets, spec->ext_dev, spec->ext_len, spec->addSep, spec->sep, spec->lazySeparator, spec->lockPos,
                          spec->loadedAfter, gate, true);
        h->specMode = false;
        planned = r == SWSEM_OK;
        if (r == SWSEM_ESPEC) r = SWSEM_OK;                         // not possible this time: nothing will be queued
        else if (r) return r;
        if (planned && (r = prepare_inserts(h, up))) return r;     // the finalize's tables, on the uploads' stream too
    }
    HIPCHK(hipEventRecord(h->evTables, up));
    HIPCHK(hipStreamWaitEvent(h->stream, h->evTables, 0));
    hipStream_t sP = h->stream;
    h->mark(SWSEM_K_EMIT, true, sP);
    k_emit_p1_removed<<<grid2, dim3(CH), 0, sP>>>(v, E.dECg.p, E.dEWhich.p);
    k_emit_p1_scan<<<dim3(n), dim3(CH), 0, sP>>>(v, E.dECg.p, E.dEWhich.p);
    k_emit_p1_compact<<<grid2, dim3(CH), 0, sP>>>(v, E.dECg.p, E.dEWhich.p);
    k_emit_p1_finish<<<dim3(n), dim3(CH), 0, sP>>>(v, E.dECg.p);
    h->mark(SWSEM_K_EMIT, false, sP);
    HIPCHK(hipGetLastError());
    // pass-1 results (unmatchedChars, the dissimilarity verdict), match counts and statistics: one pinned block, one wait
    const bool needCounts = h->matchCount.size() != h->contigs.size();
    if ((r = queue_counts(h, n * sizeof(EmitOut), sP))) return r;
    if ((r = download(h, h->pin + h->pinExtraAt, E.dEOut.p, n * sizeof(EmitOut), sP)) || (r = flush_copies(h))) return r;
    HIPCHK(hipEventRecord(h->evP1, sP));
    // the rest runs on the second stream behind pass 1
    if (E.pinECap < n * sizeof(EmitOut)) {
        if (E.pinE) HIPCHK(hipHostFree(E.pinE));
        E.pinE = nullptr; E.pinECap = 0;
        const size_t want = std::max<size_t>(2 * n * sizeof(EmitOut), 1 << 20);
        if (hipHostMalloc((void **) &E.pinE, want, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return fail(SWSEM_ENOMEM, "cannot pin host memory");
        E.pinECap = want;
    }
    HIPCHK(hipStreamWaitEvent(h->streamAux, h->evP1, 0));
    h->mark(SWSEM_K_EMIT2, true, h->streamAux);
    k_emit_meta_regions<<<grid2, dim3(CH), 0, h->streamAux>>>(v, E.dECg.p);
    k_emit_meta_masks<<<dim3(spans), dim3(MLANES), 0, h->streamAux>>>(v, E.dECg.p);
    k_emit_meta_spec<<<dim3(spans), dim3(MLANES), 0, h->streamAux>>>(v, E.dECg.p, E.dEStates.p, h->metaWarm, E.dEStat.p);
    // (the next batch's resolve is launched behind this kernel, run_batch: a launch of thousands of waves that is still
    // running takes the slots the resolve's blocks are sized for, and the blocks that have to wait double its time)
    if (!h->evMeta) HIPCHK(hipEventCreateWithFlags(&h->evMeta, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->evMeta, h->streamAux));
    h->metaPending = true;
    k_emit_meta_check<<<grid2, dim3(WAVE), 0, h->streamAux>>>(v, E.dECg.p, E.dEStates.p, E.dEStat.p);
    k_emit_meta_stitch<<<dim3(n), dim3(WAVE), 0, h->streamAux>>>(v, E.dECg.p, E.dEStates.p, E.dEStat.p);
    h->mark(SWSEM_K_EMIT2, false, h->streamAux);
    HIPCHK(hipEventRecord(E.evMetaDone, h->streamAux));
    HIPCHK(hipGetLastError());
    // The byte automata (sizes .. write) are queued later (run_phase2b): behind the speculative finalize, and — when the
    // device can make a stream wait for a word in memory — not before the next batch's resolve kernel has started.
    auto phase2b = [&](bool behindFinalize) -> int {
        E.deferred2b = true; E.waitFin2b = behindFinalize;
        return SWSEM_OK;
    };
    const bool specAsked = spec && spec->ntargets > 0 && h->phase2Behind;
    if (!specAsked) { if ((r = phase2b(false))) return r; }
    if (spec && spec->ntargets > 0) {
        k_spec_verify<<<1, 256, 0, sP>>>(E.dEOut.p, E.dECg.p, n, h->dPred.p, h->dPred.p + n, spec->factor, spec->rcFactor, gate);
        if (spec->veto || !planned) HIPCHK(hipMemsetAsync(gate, 0, sizeof(uint32_t), sP));
        // several replicas: the word becomes the minimum over all of them before anything gated by it is queued
        if (spec->exchange) {
            if (spec->exchange(spec->exchange_ctx, 0, gate, (void *) h->stream)) return fail(SWSEM_EHIP, "speculative finalize: the exchange between the replicas failed");
            exchanged = true;
        }
        if (planned) {
            if ((r = launch_inserts(h, gate))) return r;
            queued = true;
        }
        if (!queued) restore();
    }
    if (specAsked) { if ((r = phase2b(queued))) return r; }
    // While pass 1 runs: the emission before this one — its second phase ran beside this batch's match-finding — is taken now, its
    // streams copied to the host (end_slot), instead of when the caller asks for them right after this call returns: on divergent
    // collections that copy is megabytes per emission and stood between one unit's loads and the next unit's launch.
    if (h->emitHostCopy) { int e = end_slot(h, si ^ 1); if (e) return e; }
    HIPCHK(hipEventSynchronize(h->evP1));                           // pass 1 and its copies to the host (not what was queued after them)
    if (needCounts) take_counts(h);
    E.eout.assign((const EmitOut *) (h->pin + h->pinExtraAt), (const EmitOut *) (h->pin + h->pinExtraAt) + n);
    if (queued) {
        bool ok = !spec->veto;                                      // the same test k_spec_verify makes
        for (int k = 0; k < n && ok; k++) {
            const uint64_t un = E.eout[k].unmatchedChars, len = E.ecg[k].n;
            ok = un != UINT64_MAX && (un * (uint64_t) spec->factor > len) == (spec->predExt[k] != 0) &&
                 (un * (uint64_t) spec->rcFactor > len) == (spec->predRC[k] != 0);
        }
        if (exchanged) ok = spec->exchange(spec->exchange_ctx, 1, nullptr, (void *) h->stream) == 1 && ok;   // ... and every other replica's
        if (ok) { if (applied) *applied = 1; }
        else restore();
    } else if (exchanged)
        (void) spec->exchange(spec->exchange_ctx, 1, nullptr, (void *) h->stream);   // (this replica said no in the reduction: the word is 0 everywhere; taken so that the exchange's state is the same on every rank)
    return SWSEM_OK;
}

int swsem_emit_batch_begin(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                           const int *factor, const int64_t *processed, const int64_t *targetIdx,
                           const uint64_t *refExtLoadedPos, uint64_t nLoaded) {
    return emit_begin_impl(h, p, n, contigIdx, lockPos, factor, processed, targetIdx, refExtLoadedPos, nLoaded, nullptr, nullptr);
}

int swsem_emit_batch_begin_spec(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                                const int *factor, const int64_t *processed, const int64_t *targetIdx,
                                const uint64_t *refExtLoadedPos, uint64_t nLoaded, const swsem_spec_finalize_t *spec, int *applied) {
    return emit_begin_impl(h, p, n, contigIdx, lockPos, factor, processed, targetIdx, refExtLoadedPos, nLoaded, spec, applied);
}

// waits for every emission still in its second phase (oldest first); afterwards their streams can be fetched
int swsem_emit_batch_end(swsem_t *h) {
    int r = end_slot(h, h->latest ^ 1);
    return r ? r : end_slot(h, h->latest);
}

// result calls read the latest emission (previous = 0) or the one before it (previous = 1), which may have been
// left running across the next swsem_emit_batch_begin
int swsem_emit_select(swsem_t *h, int previous) {
    h->selected = previous ? (h->latest ^ 1) : -1;
    return SWSEM_OK;
}

int swsem_emit_batch(swsem_t *h, const swsem_emit_params_t *p, int n, const int *contigIdx, const uint64_t *lockPos,
                     const int *factor, const int64_t *processed, const int64_t *targetIdx,
                     const uint64_t *refExtLoadedPos, uint64_t nLoaded) {
    int r = swsem_emit_batch_begin(h, p, n, contigIdx, lockPos, factor, processed, targetIdx, refExtLoadedPos, nLoaded);
    return r ? r : swsem_emit_batch_end(h);
}

void swsem_emit_set_host_copy(swsem_t *h, int on) { h->emitHostCopy = on != 0; }

// unmatchedChars (the return value of processMatches, SWSEM_SKIPPED when skipped) of every result
int swsem_emit_unmatched(swsem_t *h, uint64_t *unmatched) {
    swsem::EmitSlot &E = h->slot[h->latest];
    for (size_t k = 0; k < E.eout.size(); k++) unmatched[k] = E.eout[k].unmatchedChars;
    return SWSEM_OK;
}

// The arena is written packed — every stream of the last emit batch back to back, (result, stream)
// major — so handing it on is one device-to-device copy.
int swsem_emit_pack_dev(swsem_t *h, uint8_t *dst_dev, uint64_t cap, uint64_t *sizes, uint64_t *total) {
    HIPCHK(hipSetDevice(h->device));
    swsem::EmitSlot &E = h->sel();
    { int e = end_slot(h, (int) (&E - h->slot)); if (e) return e; }
    if (sizes)
        for (size_t k = 0; k < E.eout.size(); k++)
            for (int st = 0; st < SWSEM_NSTREAMS; st++) sizes[k * SWSEM_NSTREAMS + st] = E.eout[k].size[st];
    if (dst_dev && E.packedBytes) {
        if (E.packedBytes > cap) return fail(SWSEM_EINVAL, "swsem_emit_pack_dev: buffer too small");
        // on the emission's own stream, and waited for: the consumer may be on any stream, and the main stream
        // may already hold the next round's match-finding
        HIPCHK(hipMemcpyAsync(dst_dev, E.dEArena.p, E.packedBytes, hipMemcpyDeviceToDevice, h->s3()));
        HIPCHK(hipStreamSynchronize(h->s3()));
    }
    if (total) *total = E.packedBytes;
    return SWSEM_OK;
}

int swsem_emit_pack_dev_on(swsem_t *h, uint8_t *dst_dev, uint64_t cap, void *stream) {
    HIPCHK(hipSetDevice(h->device));
    swsem::EmitSlot &E = h->sel();
    { int e = end_slot(h, (int) (&E - h->slot)); if (e) return e; }       // (the emission has finished: its event was waited for)
    if (!E.packedBytes) return SWSEM_OK;
    if (!dst_dev || E.packedBytes > cap) return fail(SWSEM_EINVAL, "swsem_emit_pack_dev_on: buffer too small");
    HIPCHK(hipMemcpyAsync(dst_dev, E.dEArena.p, E.packedBytes, hipMemcpyDeviceToDevice, stream ? (hipStream_t) stream : h->stream));
    return SWSEM_OK;
}

int swsem_emit_counters(swsem_t *h, uint64_t *out) {
    swsem::EmitSlot &E = h->sel();
    { int e = end_slot(h, (int) (&E - h->slot)); if (e) return e; }
    for (size_t k = 0; k < E.eout.size(); k++) {
        const EmitOut &o = E.eout[k];
        const uint64_t v[6] = {o.unmatchedChars, o.extMatched, o.extMismatches, o.totalMatched, o.removed, o.nmatches};
        memcpy(out + k * 6, v, sizeof v);
    }
    return SWSEM_OK;
}

int swsem_emit_result(swsem_t *h, int k, swsem_streams_t *out) {
    swsem::EmitSlot &E = h->sel();
    { int e = end_slot(h, (int) (&E - h->slot)); if (e) return e; }
    if (k < 0 || k >= (int) E.eout.size()) return fail(SWSEM_EINVAL, "swsem_emit_result: no result %d", k);
    if (!E.hostStreamsValid) {
        E.hostAt ^= 1;
        if (E.hostStreams().reserve(E.packedBytes + 1)) return fail(SWSEM_ENOMEM, "cannot pin %llu B of host memory", (unsigned long long) E.packedBytes);
        if (E.packedBytes) HIPCHK(hipMemcpyAsync(E.hostStreams().data(), E.dEArena.p, E.packedBytes, hipMemcpyDeviceToHost, h->s3()));
        HIPCHK(hipStreamSynchronize(h->s3()));
        E.hostStreamsValid = true;
    }
    const EmitOut &o = E.eout[k];
    for (int st = 0; st < SWSEM_NSTREAMS; st++) {
        out->data[st] = E.hostStreams().data() + E.hostStreamOff[(size_t) k * SWSEM_NSTREAMS + st];
        out->size[st] = o.size[st];
    }
    out->unmatchedChars = o.unmatchedChars;
    out->extensionsMatchedChars = o.extMatched;
    out->extensionsMismatches = o.extMismatches;
    out->totalMatched = o.totalMatched;
    out->removedGapBreakingMatches = o.removed;
    out->nmatches = o.nmatches;
    return SWSEM_OK;
}

// ---- the decoder's automaton on the device (swsem_decode.hip)
static int decode_jobs(swsem_t *h, const swsem_emit_params_t *p, int n, const std::vector<DecodeJob> &jobs, std::vector<DecodeOut> &outs) {
    int r;
    // records a contig can need: one per mapLen entry (two bytes at least, four without the frugal encoding) + the tail
    std::vector<uint64_t> aux(3 * (size_t) n + 1, 0);
    uint64_t *recBase = aux.data(), *firstDiff = recBase + n + 1;
    uint32_t *badFlags = (uint32_t *) (firstDiff + n);
    for (int k = 0; k < n; k++) {
        recBase[k + 1] = recBase[k] + jobs[k].size[SWSEM_LEN] / (p->frugal64bitLenEncoding ? 2 : 4) + 2;
        firstDiff[k] = UINT64_MAX;
    }
    if ((r = h->dJobs.reserve(n)) || (r = h->dDecPlan.reserve(n)) || (r = h->dDecAux.reserve(aux.size())) || (r = h->dDecRecs.reserve(recBase[n]))) return r;
    HIPCHK(hipMemcpyAsync(h->dJobs.p, jobs.data(), (size_t) n * sizeof(DecodeJob), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->dDecAux.p, aux.data(), aux.size() * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    const uint64_t *dRecBase = h->dDecAux.p;
    unsigned long long *dFirstDiff = (unsigned long long *) (h->dDecAux.p + n + 1);
    uint32_t *dBad = (uint32_t *) (h->dDecAux.p + 2 * (size_t) n + 1);
    k_decode_plan<<<dim3(n), dim3(WAVE), 0, h->stream>>>(*p, h->dJobs.p, h->dDecRecs.p, dRecBase, h->dDecPlan.p, h->maxRefLength + REF_SLACK);
    HIPCHK(hipGetLastError());
    std::vector<DecPlanOut> plans(n);
    HIPCHK(hipMemcpyAsync(plans.data(), h->dDecPlan.p, (size_t) n * sizeof(DecPlanOut), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // grid.y holds at most 65 535 blocks: contigs go in slices of that many, each slice sized by its own longest contig
    // (a batch of draft assemblies — tens of targets of a thousand contigs each — has more)
    constexpr int YMAX = 65535;
    for (int c0 = 0; c0 < n; c0 += YMAX) {
        const int cn = std::min(YMAX, n - c0);
        uint64_t maxRec = 0, maxLen = 0;
        bool anyExpect = false;
        for (int k = c0; k < c0 + cn; k++) {
            if (plans[k].unmatched < 0) continue;
            maxRec = std::max(maxRec, plans[k].nrec); maxLen = std::max(maxLen, plans[k].destLen);
            anyExpect |= jobs[k].expect != nullptr;
        }
        if ((maxRec + 255) / 256 > 0x7FFFFFFFull || (maxLen + 4095) / 4096 > 0x7FFFFFFFull) return fail(SWSEM_EINVAL, "swsem decode: a contig too long for one launch");
        if (maxRec) k_decode_fill<<<dim3((unsigned) ((maxRec + 255) / 256), (unsigned) cn), dim3(256), 0, h->stream>>>(h->ref, *p, h->dJobs.p, h->dDecRecs.p, dRecBase, h->dDecPlan.p, dBad, h->maxRefLength + REF_SLACK, (uint32_t) c0);
        if (anyExpect && maxLen) k_decode_check<<<dim3((unsigned) ((maxLen + 4095) / 4096), (unsigned) cn), dim3(256), 0, h->stream>>>(h->dJobs.p, h->dDecPlan.p, dFirstDiff, (uint32_t) c0);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(aux.data(), h->dDecAux.p, aux.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    outs.resize(n);
    for (int k = 0; k < n; k++) {
        const bool bad = plans[k].unmatched < 0 || badFlags[k] != 0;
        outs[k].destLen = plans[k].destLen; outs[k].unmatched = bad ? -1 : plans[k].unmatched; outs[k].firstDiff = bad ? UINT64_MAX : firstDiff[k];
    }
    return SWSEM_OK;
}

int swsem_decode_contigs_dev(swsem_t *h, const swsem_emit_params_t *p, int n, const swsem_decode_job_t *jobs, uint64_t *destLen, int64_t *unmatched) {
    HIPCHK(hipSetDevice(h->device));
    if (n <= 0) return SWSEM_OK;
    std::vector<DecodeJob> jb(n);
    for (int k = 0; k < n; k++) {
        for (int st = 0; st < SWSEM_NSTREAMS; st++) { jb[k].stream[st] = jobs[k].stream_dev[st]; jb[k].size[st] = jobs[k].size[st]; }
        jb[k].refLockPos = jobs[k].refLockPos; jb[k].dest = jobs[k].dest_dev; jb[k].destCap = jobs[k].destCap; jb[k].expect = nullptr;
    }
    std::vector<DecodeOut> outs;
    int r = decode_jobs(h, p, n, jb, outs);
    if (r) return r;
    for (int k = 0; k < n; k++) { destLen[k] = outs[k].destLen; unmatched[k] = outs[k].unmatched; }
    return SWSEM_OK;
}

int swsem_emit_verify(swsem_t *h, int *nbad, int *firstBad, uint64_t *firstDiff) {
    HIPCHK(hipSetDevice(h->device));
    swsem::EmitSlot &E = h->sel();
    { int e = end_slot(h, (int) (&E - h->slot)); if (e) return e; }
    const int n = (int) E.eout.size();
    *nbad = 0; if (firstBad) *firstBad = -1; if (firstDiff) *firstDiff = UINT64_MAX;
    if (n == 0) return SWSEM_OK;
    uint64_t total = 0;
    for (int k = 0; k < n; k++) total += E.ecg[k].n + 16;
    int r;
    if ((r = h->dDecode.reserve(total))) return r;
    std::vector<DecodeJob> jb;
    std::vector<int> which;
    uint64_t at = 0;
    for (int k = 0; k < n; k++) {
        if (E.eout[k].unmatchedChars == UINT64_MAX) { at += E.ecg[k].n + 16; continue; }   // given up as dissimilar: nothing was emitted
        DecodeJob j;
        for (int st = 0; st < SWSEM_NSTREAMS; st++) { j.stream[st] = E.dEArena.p + E.hostStreamOff[(size_t) k * SWSEM_NSTREAMS + st]; j.size[st] = E.eout[k].size[st]; }
        j.refLockPos = E.ecg[k].lock; j.dest = h->dDecode.p + at; j.destCap = E.ecg[k].n; j.expect = E.qdev + E.ecg[k].qoff;
        at += E.ecg[k].n + 16;
        jb.push_back(j); which.push_back(k);
    }
    if (jb.empty()) return SWSEM_OK;
    std::vector<DecodeOut> outs;
    if ((r = decode_jobs(h, &E.params, (int) jb.size(), jb, outs))) return r;
    for (size_t i = 0; i < jb.size(); i++) {
        const int k = which[i];
        const bool ok = outs[i].unmatched >= 0 && outs[i].destLen == E.ecg[k].n && outs[i].firstDiff == UINT64_MAX &&
                        (uint32_t) outs[i].unmatched == (uint32_t) E.eout[k].unmatchedChars;
        if (!ok) {
            if (*nbad == 0) { if (firstBad) *firstBad = k; if (firstDiff) *firstDiff = outs[i].unmatched < 0 ? outs[i].destLen : outs[i].firstDiff; }
            (*nbad)++;
        }
    }
    return SWSEM_OK;
}

int swsem_emit(swsem_t *h, const swsem_emit_params_t *p, int contig, uint64_t lockPos, int factor, int64_t processed,
               int64_t targetIdx, const uint64_t *loaded, uint64_t nLoaded, swsem_streams_t *out) {
    int r = swsem_emit_batch(h, p, 1, &contig, &lockPos, &factor, &processed, &targetIdx, loaded, nLoaded);
    if (r) return r;
    return swsem_emit_result(h, 0, out);
}

}  // extern "C"
