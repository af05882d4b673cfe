// mbgc-hip: the compress hot path of `mbgc c` (file list -> match/literal byte streams) on one MI355X,
// driven by the C++ host classes that keep the reference's names. The streams are written raw, one file
// per stream, exactly what the reference's developer build dumps with `mbgc-dev v -D` (main.cpp:575-577;
// stream order of MBGC_Decoder.cpp:1085-1112): literals(13) locksPos(14) gapDelta(15) flags(16)
// mapOff(17) mapLen(18) refExtSize(19). Entropy coding (PPMd/LZMA) is the unchanged host backend of the
// reference and is not part of this tool.
//
// --gpus N: one process per GPU, forked here BEFORE anything touches a GPU; a round then holds -R targets per GPU, the
// replicas exchange the round's reference extensions and their verdicts over RCCL (include/mbgc_exchange.h) and rank 0
// takes the streams and writes them. Results equal -R (N x R) on one GPU. --exchange hostmem puts every rank on device
// -d and moves the bytes through host shared memory: a rehearsal of the protocol on a one-GPU box, not a speed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <atomic>
#include <thread>
#include <dlfcn.h>
#include <time.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "mgmp_driver.h"

// what the ranks share before their exchange exists: the RCCL ids rank 0 makes, then (host-memory transport) its mapping
struct Bootstrap {
    std::atomic<uint32_t> idsReady, failed;
    uint32_t pad[14];
    uint8_t ids[2 * MBGC_XCHG_ID_BYTES];
};
// A rank that gives up (the reference's way: message + exit(EXIT_FAILURE)) must not leave the others waiting inside a
// collective: it raises the flag on its way out, the others watch it.
static Bootstrap *g_boot = nullptr;
static std::atomic<bool> g_done{false};
static void raiseFailure() { if (g_boot && !g_done.load()) g_boot->failed.store(1); }
// Cooperative only as far as it goes: a rank killed by a signal (a fault, the OOM killer) raises nothing. Rank 0 therefore polls
// its children (waitpid WNOHANG) and raises the flag for any that ended abnormally; the children watch their parent.
static void watchOtherRanks(const std::vector<pid_t> *children, pid_t parent) {
    std::thread([children, parent] {
        while (!g_boot->failed.load()) {
            usleep(50000);
            if (children)
                for (pid_t pid : *children) {
                    siginfo_t info;
                    info.si_pid = 0;                                             // (peek: finish() still reaps the child)
                    if (waitid(P_PID, (id_t) pid, &info, WEXITED | WNOHANG | WNOWAIT) == 0 && info.si_pid == pid &&
                        !(info.si_code == CLD_EXITED && info.si_status == 0) && !g_done.load())
                        g_boot->failed.store(1);
                }
            else if (parent && getppid() != parent) g_boot->failed.store(1);       // (rank 0 is gone)
        }
        if (!g_done.load()) { fprintf(stderr, "mbgc-hip: another rank failed\n"); _exit(EXIT_FAILURE); }
    }).detach();
}

static void dump(const std::string &prefix, const char *name, const std::string &data) {
    std::ofstream f(prefix + "." + name, std::ios::binary | std::ios::trunc);
    f.write(data.data(), (std::streamsize) data.size());
}

int main(int argc, char **argv) {
    MBGC_Params params;
    std::vector<std::string> pos;
    int gpus = 1;
    std::string transport = "rccl";
    size_t shmMb = 64;
    std::string backend;                                                        // a library with the reference's leaf coders (mbgc_leaf_compress)
    int backendThreads = 8, backendBlocksScale = 1, coderThreads = 0;
    uint64_t backendOverlapBlock = 0;                                           // > 0: the backend runs beside the matching, blocks of this many bytes           // coderThreads: the reference's -t as the coders see it (0: the pool's size)
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "c") continue;
        if (a == "-t1") params.sequentialMatching = true;                        // like `mbgc c -t1` (MBGC_Params.h:867-869)
        else if (a == "-m" && i + 1 < argc) params.setCompressionMode(atoi(argv[++i]));
        else if (a == "-R" && i + 1 < argc) params.roundSize = atoi(argv[++i]);
        else if (a == "-s" && i + 1 < argc) {                                    // reference sampling step (MBGC_Params.h:593-600); an odd one: identity-encoded table entries
            params.k1 = atoi(argv[++i]);
            if (params.k1 <= 0) { fprintf(stderr, "s - reference sampling step - should be a positive integer.\n\n"); return EXIT_FAILURE; }
        }
        else if (a == "-d" && i + 1 < argc) params.device = atoi(argv[++i]);
        else if (a == "-L") params.lazyDecompressionSupport = false;             // disable lazy decompression support
        else if (a == "-U") params.uppercaseDNA = true;                          // MBGC_Params.h: converts bases to uppercase
        else if (a == "--bench") params.benchMode = true;                        // rounds timed with every contig resident in HBM (no streams written)
        else if (a == "--verify-every" && i + 1 < argc) { params.verifyEmissions = true; params.verifyEvery = atoi(argv[++i]); }
        else if (a == "--verify") params.verifyEmissions = true;                  // every emission decoded again on the device before the reference moves on
        else if (a == "--warmup" && i + 1 < argc) params.benchWarmup = atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (a == "--exchange" && i + 1 < argc) transport = argv[++i];
        else if (a == "--shm-mb" && i + 1 < argc) shmMb = (size_t) atol(argv[++i]);
        else if (a == "--ref-factor" && i + 1 < argc) params.referenceFactor = atoi(argv[++i]);   // MGMP_Params.h:205 (tests: a buffer small enough to wrap)
        else if (a == "--backend" && i + 1 < argc) backend = argv[++i];
        else if (a == "--backend-threads" && i + 1 < argc) backendThreads = atoi(argv[++i]);
        else if (a == "--backend-blocks" && i + 1 < argc) backendBlocksScale = atoi(argv[++i]);
        else if (a == "--backend-overlap" && i + 1 < argc) backendOverlapBlock = (uint64_t) atoll(argv[++i]) << 20;
        else if (a == "--coder-threads" && i + 1 < argc) coderThreads = atoi(argv[++i]);
        else pos.push_back(a);
    }
    if (pos.size() != 2) {
        fprintf(stderr, "usage: mbgc-hip c [-t1] [-m mode] [-s samplingStep] [-R targetsPerRound] [-d device] [-U] [--verify | --verify-every K] [--ref-factor F] [--bench [--warmup rounds]] "
                        "[--gpus N [--exchange rccl|hostmem] [--shm-mb M]] [--backend coders.so [--backend-threads T] [--backend-blocks K | --backend-overlap MiB] [--coder-threads t]] <sequencesListFile> <outputPrefix>\n"
                        "  --backend writes <outputPrefix>.collective: the collective section of the matcher-side streams (the header-side streams are the CLI's and\n"
                        "  go in empty); --coder-threads = the reference's -t as its coders see it (LZMA runs two threads when it is > 1)\n");
        return EXIT_FAILURE;
    }
    if (gpus < 1 || (transport != "rccl" && transport != "hostmem") || (gpus > 1 && params.sequentialMatching)) {
        fprintf(stderr, "--gpus needs a positive count, --exchange rccl or hostmem, and the round mode (not -t1 / -m 3: those match sequentially)\n");
        return EXIT_FAILURE;
    }
    if (gpus > 1 && params.verifyEmissions) {
        // (the sharded round loop has no verification step: every rank would have to decode its own emissions before the round's
        // finalize — accepted silently before, the run then ended with "verified on the device: 0 contigs" and exit code 0)
        fprintf(stderr, "--verify / --verify-every check the emissions of the one-GPU schedules; with --gpus N verify a single-GPU run of the same rounds (-R N x r writes the same streams)\n");
        return EXIT_FAILURE;
    }
    std::vector<std::string> files;
    {
        std::ifstream lst(pos[0]);
        if (!lst) { fprintf(stderr, "cannot open sequences list file %s\n", pos[0].c_str()); return EXIT_FAILURE; }
        std::string line;
        while (std::getline(lst, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (!line.empty()) files.push_back(line);
        }
    }
    // ---- the ranks: forked before the first GPU call; the parent is rank 0 and reports for all
    int rank = 0;
    std::vector<pid_t> children;
    Bootstrap *boot = nullptr;
    size_t sharedBytes = 0;
    const bool explicitExchange = gpus > 1 || transport == "hostmem";
    if (explicitExchange || getenv("MBGC_HIP_EXCHANGE")) {
        sharedBytes = sizeof(Bootstrap) + (transport == "hostmem" ? std::max<size_t>(shmMb << 20, mbgc_xchg_hostmem_min_bytes(gpus))
                                                                   : std::max<size_t>(1 << 20, mbgc_xchg_hostmem_min_bytes(gpus)));   // rccl: the hosts' control exchanges
        void *m = mmap(nullptr, sharedBytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);      // (zero-filled)
        if (m == MAP_FAILED) { perror("mmap"); return EXIT_FAILURE; }
        boot = g_boot = (Bootstrap *) m;
        atexit(raiseFailure);
        fflush(stdout); fflush(stderr);
        const pid_t parent = getpid();
        for (int r = 1; r < gpus; r++) {
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return EXIT_FAILURE; }
            if (pid == 0) { rank = r; children.clear(); break; }
            children.push_back(pid);
        }
        static std::vector<pid_t> watched;                                      // (outlives main's frame for the detached watcher)
        watched = children;
        watchOtherRanks(rank == 0 ? &watched : nullptr, rank == 0 ? 0 : parent);
        if (transport == "rccl") {
            params.device = rank;                                                // one rank per GPU
            if (rank == 0) {
                if (mbgc_xchg_unique_ids(boot->ids)) { fprintf(stderr, "exchange: %s\n", mbgc_xchg_last_error()); return EXIT_FAILURE; }
                boot->idsReady.store(1);
            } else {
                while (!boot->idsReady.load() && !boot->failed.load()) usleep(1000);
            }
            if (mbgc_xchg_create_rccl(&params.exchange, boot->ids, rank, gpus, params.device) ||
                mbgc_xchg_set_host_control(params.exchange, (uint8_t *) m + sizeof(Bootstrap), sharedBytes - sizeof(Bootstrap))) {
                fprintf(stderr, "exchange (rank %d): %s\n", rank, mbgc_xchg_last_error());
                return EXIT_FAILURE;
            }
        } else if (mbgc_xchg_create_hostmem(&params.exchange, (uint8_t *) m + sizeof(Bootstrap), sharedBytes - sizeof(Bootstrap), rank, gpus, params.device)) {
            fprintf(stderr, "exchange (rank %d): %s\n", rank, mbgc_xchg_last_error());
            return EXIT_FAILURE;
        }
    }
    auto finish = [&](int rc) {
        g_done.store(rc == 0);
        if (params.exchange) { mbgc_xchg_destroy(params.exchange); params.exchange = nullptr; }
        if (rank != 0) { fflush(stdout); fflush(stderr); _exit(rc); }
        for (pid_t pid : children) {
            int st = 0;
            if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : EXIT_FAILURE;
        }
        return rc;
    };
    MultipleGenomeMatchingProcessor::bindHostThreadsToDeviceNode(params.device);
    MBGC_Encoder enc(&params);
    mbgc_leaf_compress_fn leaf = nullptr;
    if (!backend.empty() && rank == 0 && !params.benchMode) {
        // the reference's unchanged PPMd7 / LZMA: a library whose symbol mbgc_leaf_compress has the callback's signature
        void *lib = dlopen(backend.c_str(), RTLD_NOW | RTLD_LOCAL);
        leaf = lib ? (mbgc_leaf_compress_fn) dlsym(lib, "mbgc_leaf_compress") : nullptr;
        if (!leaf) { fprintf(stderr, "cannot load the leaf coders from %s: %s\n", backend.c_str(), dlerror()); return finish(EXIT_FAILURE); }
        if (backendOverlapBlock) {
            mbgc_backend_params_t bp;
            enc.backendParams(bp, 1, coderThreads > 0 ? coderThreads : backendThreads);
            enc.backendStream = mbgc_backend_stream_open(&bp, leaf, nullptr, backendThreads, backendOverlapBlock);
            if (!enc.backendStream) { fprintf(stderr, "%s\n", mbgc_backend_last_error()); return finish(EXIT_FAILURE); }
        }
    }
    struct timespec tEnc0, tEnc1;
    clock_gettime(CLOCK_MONOTONIC, &tEnc0);
    enc.encode(files);
    clock_gettime(CLOCK_MONOTONIC, &tEnc1);
    if (rank != 0) return finish(0);
    if (params.benchMode) {
        // the C++ host's own measurement of the hot path (BASELINE.json metric): inputs resident in HBM, rounds of -R targets
        printf("{\"metric\": \"input Gbases/s (compress hot path, C++ host)\", \"value\": %.4f, \"unit\": \"Gbases/s\", \"rounds\": %d, "
               "\"warmup_rounds\": %d, \"targets_per_round\": %d, \"bases\": %llu, \"seconds\": %.6f, \"ms_per_round\": %.4f, "
               "\"n_gpus\": %d, \"exchange\": \"%s\", \"rounds_finalized_on_device_verdicts\": %d, \"extension_bytes_dropped\": %zu, "
               "\"max_ref_len\": %zu}\n",
               params.benchBases / params.benchSeconds / 1e9, params.benchRounds, params.benchWarmup, params.roundSize * gpus,
               (unsigned long long) params.benchBases, params.benchSeconds, params.benchSeconds * 1e3 / std::max(1, params.benchRounds),
               gpus, params.exchange ? transport.c_str() : "none", params.specRounds, enc.droppedExtensionBytes(), enc.maxReferenceLength());
        return finish(0);
    }
    dump(pos[1], "literals", enc.literals);
    if (params.rcRedundancyRemoval) { dump(pos[1], "rcMapOff", enc.rcMapOff); dump(pos[1], "rcMapLen", enc.rcMapLen); }
    dump(pos[1], "locksPos", enc.locksPosStream);
    dump(pos[1], "gapDelta", enc.gapDeltas);
    dump(pos[1], "flags", enc.gapMismatchesFlags);
    dump(pos[1], "mapOff", enc.mapOff);
    dump(pos[1], "mapOff5th", enc.mapOff5thByte);
    dump(pos[1], "mapLen", enc.mapLen);
    dump(pos[1], "refExtSize", enc.refExtSizeStream);
    if (!backend.empty()) {
        // the streams through the backend's job table and container framing (include/mbgc_backend.h), entropy-coded by the
        // library given
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        uint64_t early = 0;
        const std::string section = enc.backendStream ? enc.finishBackendStream(&early)
                                                      : enc.compressStreams(leaf, nullptr, backendThreads, backendBlocksScale, coderThreads);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (enc.backendStream) { mbgc_backend_stream_close(enc.backendStream); enc.backendStream = nullptr; }
        dump(pos[1], "collective", section);
        const size_t raw = enc.literals.size() + enc.rcMapOff.size() + enc.rcMapLen.size() + enc.locksPosStream.size() + enc.gapDeltas.size() +
                           enc.gapMismatchesFlags.size() + enc.mapOff.size() + enc.mapOff5thByte.size() + enc.mapLen.size() + enc.refExtSizeStream.size();
        const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
        if (backendOverlapBlock)
            printf("backend: %zu stream bytes to %zu, %.0f ms after the matching's %.0f ms (%d threads, blocks of %llu MiB, %llu of them coded while the matching ran)\n",
                   raw, section.size(), ms, (tEnc1.tv_sec - tEnc0.tv_sec) * 1e3 + (tEnc1.tv_nsec - tEnc0.tv_nsec) * 1e-6, backendThreads,
                   (unsigned long long) (backendOverlapBlock >> 20), (unsigned long long) early);
        else
            printf("backend: %zu stream bytes to %zu in %.0f ms (%d threads, %d x the reference's blocks)\n", raw, section.size(), ms, backendThreads,
                   std::max(1, backendBlocksScale));
    }
    if (params.verifyEmissions) {
        printf("verified on the device: %llu contigs, %llu bases decoded back to their bytes\n",
               (unsigned long long) params.verifiedContigs, (unsigned long long) params.verifiedBases);
        if (params.verifiedContigs == 0 && files.size() > 1) {
            fprintf(stderr, "--verify was asked for and no emission was verified\n");
            return finish(EXIT_FAILURE);
        }
    }
    if (!params.sequentialMatching)
        printf("rounds of %d targets%s; reference extension bytes dropped at the sliding window's end: %zu\n", params.roundSize,
               gpus > 1 ? " per GPU" : "", enc.droppedExtensionBytes());
    printf("final reference length: %zu\n", enc.finalReferenceLength());
    printf("exact matches total: %zu\n", enc.exactMatches());
    printf("removed matches breaking gaps total: %zu\n", enc.removedGapBreakingMatchesAll);
    printf("swsMEM unmatched chars: %zu\n", enc.unmatchedChars());
    printf("extensions matched chars: %zu\n", enc.extensionsMatchedCharsAll);
    printf("final unmatched chars: %zu\n", enc.unmatchedChars() - enc.extensionsMatchedCharsAll);
    if (params.exchange) printf("rounds finalized on the ranks' device-side verdicts: %d\n", params.specRounds);
    if (params.exchange) printf("rounds whose extension exchange carried only the loadable head: %d\n", params.headRounds);
    return finish(0);
}
