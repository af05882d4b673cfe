// processTargetsRounds with the round's targets sharded over the GPUs of one node: rank g of N
// matches and emits targets [base + g·R, base + (g+1)·R) of a round of N·R against its own replica of the reference,
// every replica then loads every target's extension in target order (so the replicas stay bit-identical), and the
// emitted streams travel to rank 0, which merges them in target order as MBGC_Encoder.cpp:542-564 does. It is the
// reference's parallel mode (MGMP.cpp:340-468, :520-555) with N·R workers and the deterministic schedule of
// processTargetsRounds; results equal `mbgc-hip c -R N·R` on one GPU byte for byte.
//
// The exchange (include/mbgc_exchange.h) per round, in the common case (every target of the last round was loaded
// whole, without reverse complement, on every rank): one small all-gather of the targets' sizes at the top, the
// all-gather of the round's contig bytes started right after it — beside match-finding —, and the reduction of the
// ranks' device-side verdicts inside swsem_emit_batch_begin_spec, on the stream, between pass 1 and the gated finalize
// of ALL the round's targets. No host round trip between pass 1 and the finalize. The streams of a round are gathered
// one round later, like the single-GPU loop collects them. Otherwise (first rounds, reverse-complement extensions,
// contigs given up as dissimilar): the decisions are exchanged after pass 1, the extension strings after that, and a
// round whose first pass gave a contig up goes on, from the first such target, in units of
// allowedTargetsOutrunForDissimilarContigs + 1 targets matched with every earlier target loaded — as processTargetsRounds does.
#include "mgmp_driver.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <time.h>

namespace {

struct SpecExchange { mbgc_xchg_t *x; };

int specExchange(void *ctx, int phase, void *gateDev, void *stream) {
    mbgc_xchg_t *x = ((SpecExchange *) ctx)->x;
    if (phase == 0) {
        // the gated launches copy out of the extension all-gather's output and obey every rank's verdict
        if (mbgc_xchg_stream_wait_bytes(x, stream)) return -1;
        return mbgc_xchg_allreduce_min_u32(x, (uint32_t *) gateDev, stream);
    }
    uint32_t v = 0;
    if (mbgc_xchg_reduced_u32(x, &v)) return 0;
    return (int) v;
}

const int META = 12;      // per contig: local target, six stream sizes, five counters

}  // namespace

void MultipleGenomeMatchingProcessor::processTargetsRoundsSharded() {
    mbgc_xchg_t *X = params->exchange;
    const uint32_t N = (uint32_t) mbgc_xchg_world(X), g = (uint32_t) mbgc_xchg_rank(X);
    const uint32_t R = (uint32_t) std::max(1, params->roundSize), perRound = N * R;
    const uint32_t nRounds = (targetsCount + perRound - 1) / perRound;
    const bool bench = params->benchMode, root = g == 0;
    initParallelProcessing();
    matchingLocksPos.assign(targetsCount, SIZE_MAX);
    unmatchedFractionFactors.assign(2 * (size_t) targetsCount, 0);
    auto xc = [&](int rc) { if (rc) { fprintf(stderr, "exchange (rank %u): %s\n", g, mbgc_xchg_last_error()); exit(EXIT_FAILURE); } };
    auto gatherInts = [&](const std::vector<int64_t> &mine, std::vector<int64_t> &all) {
        all.assign(mine.size() * N, 0);
        xc(mbgc_xchg_allgather_i64(X, mine.data(), mine.size(), all.data()));
    };
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    struct DevBuf {
        uint8_t *p = nullptr; size_t cap = 0;
    };
    auto ensure = [&](DevBuf &b, size_t need) {               // grow-only (freeing device memory waits for the whole device)
        if (need <= b.cap) return;
        if (b.p) matcher->devFree(b.p);
        b.cap = need + need / 4 + 64;
        b.p = matcher->devAlloc(b.cap);
    };
    DevBuf extAll[2], extTmp, packDev, gatherDev, padDev;
    uint8_t *gateDev = matcher->devAlloc(64);
    SpecExchange specCtx{X};

    std::vector<RoundBatch> slots(bench ? nRounds : 3);
    auto rankRange = [&](uint32_t q, uint32_t r, uint32_t &a, uint32_t &b) {
        a = std::min<uint64_t>(targetsCount, (uint64_t) q * perRound + (uint64_t) r * R);
        b = std::min<uint64_t>(targetsCount, (uint64_t) q * perRound + (uint64_t) (r + 1) * R);
    };
    if (bench) {
        matcher->setEmitHostCopy(false);
        for (uint32_t q = 0; q < nRounds; q++) {
            rankRange(q, g, slots[q].t0, slots[q].t1);
            uint32_t n0, n1;
            rankRange(q + 1, g, n0, n1);
            loadRound(1 + slots[q].t0, 1 + slots[q].t1, slots[q], 1 + n0, 1 + n1, q + 1 < nRounds ? &slots[q + 1] : nullptr);
        }
    }

    // a round whose streams have not been collected yet
    struct Deferred {
        bool valid = false, onDevice = false;
        RoundBatch *B = nullptr;
        uint32_t q = 0;
        std::vector<EmittedStreams> host;                   // per local contig, when not on the device any more
    } prev;
    // per-target stream merge on rank 0 (ENC.cpp:542-556); newerBegun: this rank has begun an emission since
    auto collect = [&](Deferred &D, bool newerBegun) {
        RoundBatch &B = *D.B;
        const size_t ncont = B.targetOf.size();
        const uint32_t base = D.q * perRound, ntot = std::min(perRound, targetsCount - base);
        std::vector<int64_t> meta(ncont * META, 0);
        uint64_t total = 0;
        std::string hostBlob;                               // this rank's bytes when they are host-side already
        if (ncont && D.onDevice) {
            if (newerBegun) matcher->emitSelect(true);
            std::vector<uint64_t> sizes, cnt;
            total = matcher->emitPack(nullptr, 0, &sizes, (int) ncont);
            matcher->emitCounters(cnt, (int) ncont);
            if (!root) { ensure(packDev, total); matcher->emitPack(packDev.p, packDev.cap, nullptr, (int) ncont); }
            for (size_t c = 0; c < ncont; c++) {
                meta[c * META] = B.targetOf[c];
                for (int st = 0; st < SWSEM_NSTREAMS; st++) meta[c * META + 1 + st] = (int64_t) sizes[c * SWSEM_NSTREAMS + st];
                for (int i = 0; i < 5; i++) meta[c * META + 7 + i] = (int64_t) cnt[c * 6 + i];
            }
            if (root && !bench) {
                D.host.assign(ncont, EmittedStreams());
                for (size_t c = 0; c < ncont; c++) matcher->emitTake((int) c, D.host[c]);
            }
            if (newerBegun) matcher->emitSelect(false);
        } else if (ncont) {
            for (size_t c = 0; c < ncont; c++) {
                const EmittedStreams &es = D.host[c];
                meta[c * META] = B.targetOf[c];
                for (int st = 0; st < SWSEM_NSTREAMS; st++) { meta[c * META + 1 + st] = (int64_t) es.s[st].size(); total += es.s[st].size(); }
                const uint64_t cnt[5] = {es.unmatchedChars, es.extensionsMatchedChars, es.extensionsMismatches, es.totalMatched, es.removedGapBreakingMatches};
                for (int i = 0; i < 5; i++) meta[c * META + 7 + i] = (int64_t) cnt[i];
            }
            if (!root) {
                hostBlob.reserve(total);
                for (size_t c = 0; c < ncont; c++)
                    for (int st = 0; st < SWSEM_NSTREAMS; st++) hostBlob.append(D.host[c].s[st]);
                ensure(packDev, total);
                matcher->devUpload(packDev.p, hostBlob.data(), total);
            }
        }
        std::vector<int64_t> head;
        gatherInts({(int64_t) ncont, (int64_t) total}, head);
        size_t maxC = 0;
        for (uint32_t r = 0; r < N; r++) maxC = std::max<size_t>(maxC, (size_t) head[2 * r]);
        std::vector<int64_t> allMeta;
        if (maxC) { meta.resize(maxC * META, 0); gatherInts(meta, allMeta); }
        std::vector<uint64_t> bytesOf(N, 0);
        uint64_t sum = 0;
        for (uint32_t r = 1; r < N; r++) { bytesOf[r] = (uint64_t) head[2 * r + 1]; sum += bytesOf[r]; }     // (rank 0 holds its own)
        if (root) ensure(gatherDev, sum);
        xc(mbgc_xchg_gather_to_root(X, packDev.p, bytesOf.data(), gatherDev.p));
        if (!root) return;
        std::string remote;
        if (!bench && sum) { remote.resize(sum); matcher->devDownload(&remote[0], gatherDev.p, sum); }
        // target order = rank-major inside the round; the contigs of a target in their rank's order
        std::vector<size_t> at(N, 0);                       // read position in every rank's bytes
        { size_t o = 0; for (uint32_t r = 1; r < N; r++) { at[r] = o; o += bytesOf[r]; } }
        std::vector<size_t> nextC(N, 0);
        for (uint32_t j = 0; j < ntot; j++) {
            const uint32_t r = j / R, lt = j % R, t = base + j;
            const size_t nc = (size_t) head[2 * r];
            while (nextC[r] < nc && (uint32_t) allMeta[(r * maxC + nextC[r]) * META] == lt) {
                const int64_t *m = &allMeta[(r * maxC + nextC[r]) * META];
                EmittedStreams es;
                if (r == 0) {
                    if (!bench) es = std::move(D.host[nextC[r]]);
                } else if (!bench) {
                    for (int st = 0; st < SWSEM_NSTREAMS; st++) { es.s[st].assign(remote, at[r], (size_t) m[1 + st]); at[r] += (size_t) m[1 + st]; }
                }
                es.unmatchedChars = (uint64_t) m[7]; es.extensionsMatchedChars = (uint64_t) m[8]; es.extensionsMismatches = (uint64_t) m[9];
                es.totalMatched = (uint64_t) m[10]; es.removedGapBreakingMatches = (uint64_t) m[11];
                takeRoundStreams(t, es);
                processAfterSequence(t);
                nextC[r]++;
            }
            processAfterTarget(t);
            appendTargetStreams(t);
        }
    };

    bool gpred = false;              // last round: every target of every rank was loaded whole, without reverse complement
    double tStart = 0;
    for (uint32_t q = 0; q < nRounds; q++) {
        RoundBatch &B = slots[bench ? q : q % 3];
        if (!bench) {
            rankRange(q, g, B.t0, B.t1);
            uint32_t n0, n1;
            rankRange(q + 1, g, n0, n1);
            loadRound(1 + B.t0, 1 + B.t1, B, 1 + n0, 1 + n1, q + 1 < nRounds ? &slots[(q + 1) % 3] : nullptr);
        } else if ((int) q == params->benchWarmup) {
            if (prev.valid) { if (prev.onDevice) matcher->emitEnd(); collect(prev, false); prev.valid = false; }
            matcher->synchronize();
            std::vector<int64_t> all;
            gatherInts({0}, all);                           // every rank starts its clock together
            tStart = now();
        }
        const uint32_t base = q * perRound, ntot = std::min(perRound, targetsCount - base);
        const size_t ncont = B.targetOf.size();
        const uint32_t T = B.t1 - B.t0;
        for (uint32_t j = 0; j < ntot; j++) {               // the same bookkeeping on every rank, in target order
            const uint32_t t = base + j;
            unmatchedFractionFactors[2 * t] = params->currentUnmatchedFractionFactor < 256 ? params->currentUnmatchedFractionFactor : 0;   // :351-352
            unmatchedFractionFactors[2 * t + 1] = (uint8_t) params->unmatchedFractionRCFactor;
            matchingLocksPos[t] = matcher->acquireWorkerMatchingLockPos();                      // :353-358
        }
        std::vector<uint64_t> roundLocks(ntot);
        for (uint32_t j = 0; j < ntot; j++) roundLocks[j] = matchingLocksPos[base + j];
        // span of every local target's contigs in the buffer (they follow each other)
        std::vector<uint64_t> tBeg(R, 0), tEnd(R, 0);
        std::vector<char> tHas(R, 0);
        for (size_t c = 0; c < ncont; c++) {
            const uint32_t lt = B.targetOf[c];
            if (!tHas[lt]) { tBeg[lt] = B.offsets[c]; tHas[lt] = 1; }
            tEnd[lt] = B.offsets[c + 1];
        }
        // what every target weighs when it is loaded whole
        std::vector<int64_t> mine(R, 0), whole;
        for (uint32_t lt = 0; lt < T; lt++) mine[lt] = tHas[lt] ? (int64_t) (tEnd[lt] - tBeg[lt]) : 0;
        gatherInts(mine, whole);
        bool canSpec = gpred && ntot == perRound;
        for (uint32_t j = 0; j < ntot && canSpec; j++) canSpec = whole[j] > 0;
        uint64_t mx = 0;
        std::vector<uint64_t> rankBytes(N, 0);
        for (uint32_t j = 0; j < ntot; j++) rankBytes[j / R] += (uint64_t) whole[j];
        for (uint32_t r = 0; r < N; r++) mx = std::max(mx, rankBytes[r]);
        DevBuf &ext = extAll[q & 1];
        bool veto = false;
        if (canSpec) {
            // the extension all-gather under the prediction: this rank's contribution is its round buffer as it is
            bool contiguous = B.bytes == rankBytes[g] && tBeg[0] == 0;
            for (uint32_t lt = 0; lt + 1 < T && contiguous; lt++) contiguous = tEnd[lt] == tBeg[lt + 1];
            veto = !contiguous;
            const uint8_t *src = B.seqDev;
            if (B.seqCap < mx) { ensure(padDev, mx); matcher->devCopy(padDev.p, B.seqDev, B.bytes); matcher->synchronize(); src = padDev.p; }
            ensure(ext, (size_t) N * mx);
            // After the wrap the round's locks stand one window ahead of the loading position and loadRef clips there
            // (SlidingWindowSparseEMMatcher.cpp:361-378, :412-417): what lies beyond that head of the round — target after
            // target — is never read, and only the ranks that hold a part of the head send it.
            const uint64_t lock = roundLocks[0], maxRef = matcher->getMaxRefLength(), pos1 = matcher->getLoadingPosition();
            bool sameLocks = true;
            for (uint32_t j = 1; j < ntot; j++) sameLocks &= roundLocks[j] == lock;
            std::vector<uint64_t> need(N, 0);
            uint64_t wanted = 0, all = 0;
            if (sameLocks && lock < maxRef) {
                const uint64_t cap = (lock > pos1 ? lock - pos1 : lock + (maxRef - 1) - pos1) + 4096;
                uint64_t start = 0;
                for (uint32_t r = 0; r < N; r++) {
                    need[r] = start < cap ? std::min<uint64_t>(rankBytes[r], cap - start) : 0;
                    start += rankBytes[r];
                    wanted += need[r]; all += rankBytes[r];
                }
            }
            // MBGC_HIP_GATHER_ALL=1 (diagnostics): every round's extensions travel whole — the A/B for the head-only exchange,
            // whose safety rests on `cap` covering exactly what loadRef's clipping lets through
            static const bool gatherAll = getenv("MBGC_HIP_GATHER_ALL") != nullptr && atoi(getenv("MBGC_HIP_GATHER_ALL")) != 0;
            if (all && wanted < all && !gatherAll) {
                xc(mbgc_xchg_bcast_heads_begin(X, src, need.data(), mx, ext.p));
                params->headRounds++;
            } else
                xc(mbgc_xchg_allgather_bytes_begin(X, src, mx, ext.p));
        }
        std::vector<uint64_t> locks(ncont), un, counts;
        std::vector<int> factors(ncont);
        std::vector<int64_t> processed(ncont, processedTargetsCount), tidx(ncont);
        for (size_t c = 0; c < ncont; c++) {
            const uint32_t t = B.t0 + B.targetOf[c];
            locks[c] = matchingLocksPos[t]; factors[c] = unmatchedFractionFactors[2 * t]; tidx[c] = t;
        }
        bool applied = false;
        std::vector<uint64_t> loadedAfter(ntot, 0);
        const size_t before = matcher->getLoadedRefLength();
        if (ncont) {
            matcher->matchRoundBegin(B.seqDev, B.offsets, params->k, locks);                    // :379
            std::vector<const uint8_t *> extDev(ntot, nullptr);
            std::vector<uint64_t> extLen(ntot, 0);
            std::vector<uint8_t> predExt(ncont, 1), predRC(ncont, 0);
            swsem_spec_finalize_t spec = {};
            if (canSpec) {
                std::vector<uint64_t> cur(N, 0);
                for (uint32_t j = 0; j < ntot; j++) {
                    const uint32_t r = j / R;
                    extDev[j] = ext.p + (size_t) r * mx + cur[r]; extLen[j] = (uint64_t) whole[j];
                    cur[r] += extLen[j];
                }
                spec.ntargets = (int) ntot; spec.ext_dev = extDev.data(); spec.ext_len = extLen.data();
                spec.addSep = params->refRegionSeparators; spec.sep = 0; spec.lazySeparator = lazyMode();
                spec.lockPos = roundLocks.data(); spec.loadedAfter = loadedAfter.data();
                spec.predExt = predExt.data(); spec.predRC = predRC.data();
                spec.factor = params->currentUnmatchedFractionFactor; spec.rcFactor = params->rcInReference ? params->unmatchedFractionRCFactor : 0;
                spec.gate_dev = (uint32_t *) gateDev; spec.exchange = specExchange; spec.exchange_ctx = &specCtx; spec.veto = veto;
            }
            applied = matcher->emitRoundBegin(emitParams(), locks, factors, processed, tidx, loadedPositions(),
                                              canSpec ? &spec : nullptr, un, counts);           // :381
        }
        if (getenv("MBGC_HIP_SHARD_DEBUG")) {
            fprintf(stderr, "[rank %u round %u] canSpec %d applied %d veto %d targets %u..%u un:", g, q, (int) canSpec, (int) applied, (int) veto, B.t0, B.t1);
            for (size_t c = 0; c < ncont && c < un.size(); c++) fprintf(stderr, " %llu/%llu", (unsigned long long) un[c], (unsigned long long) (B.offsets[c + 1] - B.offsets[c]));
            fprintf(stderr, " loaded %zu\n", matcher->getLoadedRefLength());
        }
        if (applied) {
            for (size_t c = 0; c < ncont; c++) resCount += counts[c];
            size_t startPos = before;
            for (uint32_t j = 0; j < ntot; j++) {
                noteTargetLoaded(base + j, startPos, loadedAfter[j]);                           // ENC.cpp:557-563
                startPos = loadedAfter[j];
            }
            processedTargetsCount = base + ntot;
            if (prev.valid) collect(prev, true);
            prev = Deferred();
            prev.valid = true; prev.onDevice = true; prev.B = &B; prev.q = q;
            params->specRounds++;
            continue;                                       // (gpred stays: that is what was just confirmed)
        }
        if (canSpec) xc(mbgc_xchg_wait_bytes(X));           // the all-gather that was not used: its buffers are free again

        // ---- the general form: decisions after pass 1, extension strings after them, retries
        std::vector<int> pending(ncont);
        for (size_t c = 0; c < ncont; c++) pending[c] = (int) c;
        std::vector<uint64_t> unmatched(ncont, SIZE_MAX), cnt(ncont, 0);
        std::vector<EmittedStreams> hostStreams(ncont);
        auto J = [&](size_t c) { return (uint32_t) (B.t0 - base) + B.targetOf[c]; };       // position of contig c's target in the round
        uint32_t finalized = 0;
        bool firstPass = true, onDevice = ncont > 0, everyWhole = true, retried = false;
        std::map<uint32_t, int> cutOf;                      // local targets still to be matched again (-> their first contig)
        std::vector<char> stoppedAll;                       // the round's targets that gave a contig up in the first pass
        while (true) {
            if (!pending.empty()) {
                std::vector<EmittedStreams> out;
                if (!firstPass) {
                    const int c0 = pending.front();         // (the rest of one target: consecutive in the buffer)
                    std::vector<uint64_t> offs, lk;
                    std::vector<int> fa;
                    std::vector<int64_t> pr, ti;
                    for (int c : pending) {
                        offs.push_back(B.offsets[c] - B.offsets[c0]);
                        lk.push_back(locks[c]); fa.push_back(factors[c]); pr.push_back(processedTargetsCount); ti.push_back(tidx[c]);
                    }
                    offs.push_back(B.offsets[pending.back() + 1] - B.offsets[c0]);
                    matcher->matchRound(B.seqDev + B.offsets[c0], offs, params->k, lk, counts);
                    matcher->emitRound(emitParams(), lk, fa, pr, ti, loadedPositions(), out);
                    un.assign(pending.size(), 0);
                    for (size_t k = 0; k < pending.size(); k++) un[k] = out[k].unmatchedChars;
                }
                // contigs given up as dissimilar (:382-388): only the first pass over the whole round meets them
                std::map<uint32_t, int> fresh;
                for (size_t k = 0; k < pending.size(); k++)
                    if (un[k] == PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY) fresh.emplace(B.targetOf[pending[k]], pending[k]);
                if (!firstPass && !fresh.empty()) { fprintf(stderr, "internal error: a contig was given up although every target in front of its unit had been loaded\n"); exit(EXIT_FAILURE); }
                for (size_t k = 0; k < pending.size(); k++) {
                    const int c = pending[k];
                    if (un[k] == PROCESSING_MATCHES_SKIPPED_DUE_TO_CONTIG_DISSIMILARITY) continue;
                    unmatched[c] = un[k]; cnt[c] = counts[k];
                    if (!firstPass) hostStreams[c] = std::move(out[k]);
                }
                cutOf.insert(fresh.begin(), fresh.end());
            }
            // the first target (in target order) still to be matched (again): the finalizer gets that far. (The first pass also
            // tells which of the round's targets gave a contig up: R flags per rank.)
            int64_t mineSkip = ntot;
            for (const auto &kv : cutOf) mineSkip = std::min<int64_t>(mineSkip, (int64_t) (B.t0 - base) + kv.first);
            std::vector<int64_t> mine(1, mineSkip), all;
            if (firstPass)
                for (uint32_t lt = 0; lt < R; lt++) mine.push_back(cutOf.count(lt) ? 1 : 0);
            gatherInts(mine, all);
            const size_t per = mine.size();
            uint32_t firstSkip = ntot;
            for (uint32_t r = 0; r < N; r++) firstSkip = std::min<uint32_t>(firstSkip, (uint32_t) all[r * per]);
            if (firstPass && firstSkip < ntot) {
                // the first pass met dissimilar contigs: the targets that hold one are void, whole (a worker that starts its target
                // again when its turn has come); the others keep what was found. The finalizer takes the targets in order; the
                // stopped targets that follow each other — at most allowedTargetsOutrunForDissimilarContigs + 1, a unit — are
                // matched again with every target in front of the unit loaded (processMatches then gives nothing up, ENC.cpp:203)
                // and loaded before the finalizer goes on: the schedule of processTargetsRounds and tests/_driver.py, so that N
                // ranks write what one GPU writes.
                stoppedAll.assign(ntot, 0);
                for (uint32_t j = 0; j < ntot; j++) stoppedAll[j] = all[(j / R) * per + 1 + j % R] != 0;
                cutOf.clear();
                for (size_t c = 0; c < ncont; c++)
                    if (stoppedAll[J(c)]) { unmatched[c] = SIZE_MAX; cnt[c] = 0; cutOf.emplace(B.targetOf[c], (int) c); }
            }
            if (firstSkip < ntot && !retried) {
                // a retry follows: its emissions take over the buffers — the streams still on the device are taken now, and
                // the round before this one is collected first (by every rank: it holds collectives)
                retried = true;
                if (prev.valid) { collect(prev, ncont > 0); prev.valid = false; }
                if (onDevice) {
                    matcher->emitEnd();
                    for (size_t c = 0; c < ncont; c++)
                        if (unmatched[c] != SIZE_MAX) matcher->emitTake((int) c, hostStreams[c]);
                    onDevice = false;
                }
            }
            // the next unit: this rank's share of it
            std::vector<int> redo;
            if (firstSkip < ntot) {
                const uint32_t unit = (uint32_t) std::max(0, emitParams().allowedTargetsOutrunForDissimilarContigs) + 1;
                uint32_t end = firstSkip + 1;
                while (end < ntot && end - firstSkip < unit && stoppedAll[end]) end++;
                for (uint32_t j = firstSkip; j < end; j++) {
                    if (j / R != g || !cutOf.count(j % R)) continue;
                    for (size_t c = 0; c < ncont; c++)
                        if (B.targetOf[c] == j % R) redo.push_back((int) c);
                    cutOf.erase(j % R);
                }
            }
            // targets [finalized, firstSkip) are complete on every rank: their extension strings (contig, then its reverse
            // complement, :389-398) are exchanged and loaded in target order by every replica (:433-468)
            if (firstSkip > finalized) {
                std::vector<int64_t> lens(R + 1, 0), allLens;
                size_t need = 0;
                bool localWhole = true;
                for (size_t c = 0; c < ncont; c++) {
                    const uint32_t j = J(c);
                    if (j < finalized || j >= firstSkip) continue;
                    const size_t len = B.offsets[c + 1] - B.offsets[c];
                    const bool e = params->isContigProperForRefExtension(len, unmatched[c], factors[c]);
                    const bool rc = params->rcInReference && params->isContigProperForRefRCExtension(len, unmatched[c], params->unmatchedFractionRCFactor);
                    lens[B.targetOf[c]] += (int64_t) ((e ? len : 0) + (rc ? len : 0));
                    need += (e ? len : 0) + (rc ? len : 0);
                    localWhole = localWhole && e && !rc;
                }
                lens[R] = localWhole ? 1 : 0;
                gatherInts(lens, allLens);
                uint64_t most = need;
                for (uint32_t r = 0; r < N; r++) {
                    uint64_t s = 0;
                    for (uint32_t lt = 0; lt < R; lt++) s += (uint64_t) allLens[r * (R + 1) + lt];
                    most = std::max(most, s);
                    everyWhole = everyWhole && allLens[r * (R + 1) + R] == 1;
                }
                ensure(extTmp, most);
                size_t at = 0;
                for (size_t c = 0; c < ncont; c++) {
                    const uint32_t j = J(c);
                    if (j < finalized || j >= firstSkip) continue;
                    const size_t len = B.offsets[c + 1] - B.offsets[c];
                    if (params->isContigProperForRefExtension(len, unmatched[c], factors[c])) { matcher->devCopy(extTmp.p + at, B.seqDev + B.offsets[c], len); at += len; }
                    if (params->rcInReference && params->isContigProperForRefRCExtension(len, unmatched[c], params->unmatchedFractionRCFactor)) {
                        matcher->devRevComp(B.seqDev + B.offsets[c], len, extTmp.p + at);
                        at += len;
                    }
                }
                matcher->synchronize();                     // the strings are in place (and nothing reads this round's buffer any more)
                // THIS round's buffer — the one its own speculative all-gather may have used (not applied: its gated launches
                // have found their gate shut) — never the other one: the finalize queued below reads the buffer while the
                // NEXT round's all-gather, which starts at that round's top on a stream of its own, writes the other. (With the
                // other one here, a replica that ran late loaded the next round's targets in place of this round's — and
                // then found one of them in its reference, whole: four and five ranks on one GPU showed it, round 4.)
                DevBuf &dst = extAll[q & 1];
                ensure(dst, (size_t) N * most);
                xc(mbgc_xchg_allgather_bytes_begin(X, extTmp.p, most, dst.p));
                xc(mbgc_xchg_wait_bytes(X));
                const uint32_t nfin = firstSkip - finalized;
                std::vector<const uint8_t *> extDev(nfin, nullptr);
                std::vector<uint64_t> extLen(nfin, 0), tl(nfin), after;
                std::vector<uint64_t> cur(N, 0);
                for (uint32_t j = finalized; j < firstSkip; j++) {
                    const uint32_t r = j / R, lt = j % R;
                    const uint64_t ln = (uint64_t) allLens[r * (R + 1) + lt];
                    if (ln) extDev[j - finalized] = dst.p + (size_t) r * most + cur[r];
                    extLen[j - finalized] = ln; cur[r] += ln;
                    tl[j - finalized] = roundLocks[j];
                }
                size_t startPos = matcher->getLoadedRefLength();
                matcher->finalizeTargets(extDev, extLen, params->refRegionSeparators, 0, lazyMode(), tl, after);   // :440-457
                for (uint32_t j = finalized; j < firstSkip; j++) {
                    noteTargetLoaded(base + j, startPos, after[j - finalized]);
                    startPos = after[j - finalized];
                }
                processedTargetsCount = base + firstSkip;
                finalized = firstSkip;
            }
            if (finalized >= ntot) break;
            pending = redo;
            for (int c : pending) { unmatched[c] = SIZE_MAX; cnt[c] = 0; }
            firstPass = false;
        }
        for (size_t c = 0; c < ncont; c++) resCount += cnt[c];
        if (!retried && prev.valid) collect(prev, ncont > 0);
        prev = Deferred();
        prev.valid = true; prev.onDevice = onDevice; prev.B = &B; prev.q = q;
        if (!onDevice) prev.host = std::move(hostStreams);
        gpred = everyWhole && !retried && ntot == perRound;
    }
    if (prev.valid) { if (prev.onDevice) matcher->emitEnd(); collect(prev, false); }
    matcher->synchronize();
    {
        // totals over the ranks (rank 0 reports them); the clock stops when the slowest rank has finished
        std::vector<int64_t> all;
        gatherInts({(int64_t) resCount}, all);
        if (root) { resCount = 0; for (uint32_t r = 0; r < N; r++) resCount += (size_t) all[r]; }
    }
    if (bench) {
        params->benchSeconds = now() - tStart;
        params->benchRounds = (int) nRounds - params->benchWarmup;
        uint64_t bases = 0;
        for (uint32_t q = (uint32_t) params->benchWarmup; q < nRounds; q++) bases += slots[q].bytes;
        std::vector<int64_t> all;
        gatherInts({(int64_t) bases}, all);
        params->benchBases = 0;
        for (uint32_t r = 0; r < N; r++) params->benchBases += (uint64_t) all[r];
    }
    for (auto &B : slots) if (B.seqDev) matcher->devFree(B.seqDev);
    for (DevBuf *b : {&extAll[0], &extAll[1], &extTmp, &packDev, &gatherDev, &padDev}) if (b->p) matcher->devFree(b->p);
    matcher->devFree(gateDev);
}
