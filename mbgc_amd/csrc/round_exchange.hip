// The exchange step between the replicas of one node (include/mbgc_exchange.h): RCCL over xGMI, or host shared memory
// for rehearsing several ranks on one GPU. Built into libmbgc_xchg.so (links librccl) — only the C++ host loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <sched.h>

#include "../../include/mbgc_exchange.h"

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

#define HIPX(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define NCCLX(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail("%s: %s", #x, ncclGetErrorString(r_)); } while (0)

// header of the shared mapping (host-memory transport); everything behind it is the staging area
struct Shared {
    std::atomic<uint32_t> arrived, generation;
    std::atomic<uint32_t> failed;
    uint32_t pad[13];
};
static_assert(sizeof(Shared) == 64, "header layout");
static_assert(std::atomic<uint32_t>::is_always_lock_free, "the barrier lives in shared memory");

double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

}  // namespace

struct mbgc_xchg {
    int rank = 0, world = 1, device = 0;
    bool rccl = false;
    // RCCL
    ncclComm_t bulk = nullptr, ctl = nullptr;
    hipStream_t sBulk = nullptr, sCtl = nullptr;
    hipEvent_t evBytes = nullptr, evWord = nullptr;
    uint32_t *hWord = nullptr;                                             // pinned: the last reduced word
    int64_t *dInts = nullptr, *hInts = nullptr; uint64_t intsCap = 0;      // [k | world*k] on the device, pinned mirror
    // host memory
    Shared *sh = nullptr; uint8_t *area = nullptr; uint64_t areaBytes = 0;

    int barrier() {
        // sense-reversing barrier over the shared header; gives up (all ranks) when a rank failed, or after MBGC_XCHG_BARRIER_SECONDS
        // (default 120; 0 = never: a rank reading from cold storage is slow, not dead)
        static const double limit = getenv("MBGC_XCHG_BARRIER_SECONDS") ? atof(getenv("MBGC_XCHG_BARRIER_SECONDS")) : 120.0;
        const uint32_t g = sh->generation.load();
        if (sh->arrived.fetch_add(1) + 1 == (uint32_t) world) { sh->arrived.store(0); sh->generation.fetch_add(1); return 0; }
        const double t0 = now();
        for (unsigned spin = 0; sh->generation.load() == g; spin++) {
            if (sh->failed.load()) return fail("exchange: another rank failed");
            if ((spin & 1023) == 1023) {
                sched_yield();
                if (limit > 0 && now() - t0 > limit) { sh->failed.store(1); return fail("exchange: a rank did not arrive within %.0f s", limit); }
            }
        }
        return 0;
    }
    int ints(uint64_t n) {
        if (n <= intsCap) return 0;
        if (dInts) (void) hipFree(dInts);
        if (hInts) (void) hipHostFree(hInts);
        intsCap = n + n / 2 + 64;
        HIPX(hipMalloc(&dInts, intsCap * sizeof(int64_t)));
        HIPX(hipHostMalloc(&hInts, intsCap * sizeof(int64_t), hipHostMallocDefault));
        return 0;
    }
};

extern "C" {

const char *mbgc_xchg_last_error(void) { return g_err.c_str(); }

int mbgc_xchg_unique_ids(uint8_t ids[2 * MBGC_XCHG_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == MBGC_XCHG_ID_BYTES, "id size");
    for (int i = 0; i < 2; i++) {
        ncclUniqueId id;
        NCCLX(ncclGetUniqueId(&id));
        memcpy(ids + i * MBGC_XCHG_ID_BYTES, &id, sizeof id);
    }
    return 0;
}

int mbgc_xchg_create_rccl(mbgc_xchg_t **out, const uint8_t ids[2 * MBGC_XCHG_ID_BYTES], int rank, int world, int device) {
    if (!out || world < 1 || rank < 0 || rank >= world) return fail("mbgc_xchg_create_rccl: bad arguments");
    HIPX(hipSetDevice(device));
    mbgc_xchg *x = new mbgc_xchg;
    x->rank = rank; x->world = world; x->device = device; x->rccl = true;
    ncclUniqueId id[2];
    memcpy(&id[0], ids, sizeof(ncclUniqueId));
    memcpy(&id[1], ids + MBGC_XCHG_ID_BYTES, sizeof(ncclUniqueId));
    // (on any failure what has been created so far is given back: mbgc_xchg_destroy takes a half-built object)
    if (ncclCommInitRank(&x->bulk, world, id[0], rank) != ncclSuccess || ncclCommInitRank(&x->ctl, world, id[1], rank) != ncclSuccess) {
        mbgc_xchg_destroy(x);
        return fail("ncclCommInitRank failed (rank %d of %d on device %d)", rank, world, device);
    }
    if (hipStreamCreateWithFlags(&x->sBulk, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&x->sCtl, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&x->evBytes, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&x->evWord, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(&x->hWord, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        mbgc_xchg_destroy(x);
        return fail("exchange: stream creation failed");
    }
    *out = x;
    return 0;
}

int mbgc_xchg_set_host_control(mbgc_xchg_t *x, void *shared, uint64_t sharedBytes) {
    if (!x || !shared || sharedBytes < mbgc_xchg_hostmem_min_bytes(x->world)) return fail("mbgc_xchg_set_host_control: bad arguments");
    x->sh = (Shared *) shared;
    x->area = (uint8_t *) shared + sizeof(Shared);
    x->areaBytes = sharedBytes - sizeof(Shared);
    return x->barrier();
}

uint64_t mbgc_xchg_hostmem_min_bytes(int world) { return sizeof(Shared) + (uint64_t) world * 4096; }

int mbgc_xchg_create_hostmem(mbgc_xchg_t **out, void *shared, uint64_t sharedBytes, int rank, int world, int device) {
    if (!out || !shared || world < 1 || rank < 0 || rank >= world || sharedBytes < mbgc_xchg_hostmem_min_bytes(world))
        return fail("mbgc_xchg_create_hostmem: bad arguments");
    HIPX(hipSetDevice(device));
    mbgc_xchg *x = new mbgc_xchg;
    x->rank = rank; x->world = world; x->device = device;
    x->sh = (Shared *) shared;
    x->area = (uint8_t *) shared + sizeof(Shared);
    x->areaBytes = sharedBytes - sizeof(Shared);
    if (hipStreamCreateWithFlags(&x->sCtl, hipStreamNonBlocking) != hipSuccess || hipHostMalloc(&x->hWord, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        delete x;
        return fail("exchange: stream creation failed");
    }
    *out = x;
    return x->barrier();                                  // (every rank is up)
}

void mbgc_xchg_destroy(mbgc_xchg_t *x) {
    if (!x) return;
    (void) hipSetDevice(x->device);
    if (x->sBulk) (void) hipStreamSynchronize(x->sBulk);
    if (x->sCtl) (void) hipStreamSynchronize(x->sCtl);
    if (x->bulk) ncclCommDestroy(x->bulk);
    if (x->ctl) ncclCommDestroy(x->ctl);
    if (x->evBytes) (void) hipEventDestroy(x->evBytes);
    if (x->evWord) (void) hipEventDestroy(x->evWord);
    if (x->hWord) (void) hipHostFree(x->hWord);
    if (x->sBulk) (void) hipStreamDestroy(x->sBulk);
    if (x->sCtl) (void) hipStreamDestroy(x->sCtl);
    if (x->dInts) (void) hipFree(x->dInts);
    if (x->hInts) (void) hipHostFree(x->hInts);
    delete x;
}

int mbgc_xchg_rank(const mbgc_xchg_t *x) { return x->rank; }
int mbgc_xchg_world(const mbgc_xchg_t *x) { return x->world; }

int mbgc_xchg_allgather_i64(mbgc_xchg_t *x, const int64_t *mine, uint64_t k, int64_t *all) {
    if (k == 0) return 0;
    if (x->rccl && !x->sh) {
        HIPX(hipSetDevice(x->device));
        if (x->ints(k * (x->world + 1))) return -1;
        memcpy(x->hInts, mine, k * sizeof(int64_t));
        HIPX(hipMemcpyAsync(x->dInts, x->hInts, k * sizeof(int64_t), hipMemcpyHostToDevice, x->sCtl));
        NCCLX(ncclAllGather(x->dInts, x->dInts + k, k, ncclInt64, x->ctl, x->sCtl));
        HIPX(hipMemcpyAsync(x->hInts + k, x->dInts + k, x->world * k * sizeof(int64_t), hipMemcpyDeviceToHost, x->sCtl));
        HIPX(hipStreamSynchronize(x->sCtl));
        memcpy(all, x->hInts + k, x->world * k * sizeof(int64_t));
        return 0;
    }
    // host memory: in pieces that fit the staging area
    const uint64_t per = x->areaBytes / x->world / sizeof(int64_t);
    for (uint64_t at = 0; at < k; at += per) {
        const uint64_t n = k - at < per ? k - at : per;
        memcpy(x->area + (uint64_t) x->rank * per * sizeof(int64_t), mine + at, n * sizeof(int64_t));
        if (x->barrier()) return -1;
        for (int r = 0; r < x->world; r++) memcpy(all + (uint64_t) r * k + at, x->area + (uint64_t) r * per * sizeof(int64_t), n * sizeof(int64_t));
        if (x->barrier()) return -1;
    }
    return 0;
}

int mbgc_xchg_allgather_bytes_begin(mbgc_xchg_t *x, const uint8_t *src_dev, uint64_t bytesPerRank, uint8_t *dst_dev) {
    HIPX(hipSetDevice(x->device));
    if (x->rccl) {
        if (bytesPerRank) NCCLX(ncclAllGather(src_dev, dst_dev, bytesPerRank, ncclUint8, x->bulk, x->sBulk));
        HIPX(hipEventRecord(x->evBytes, x->sBulk));
        return 0;
    }
    const uint64_t per = x->areaBytes / x->world;
    for (uint64_t at = 0; at < bytesPerRank; at += per) {
        const uint64_t n = bytesPerRank - at < per ? bytesPerRank - at : per;
        HIPX(hipMemcpyAsync(x->area + (uint64_t) x->rank * per, src_dev + at, n, hipMemcpyDeviceToHost, x->sCtl));
        HIPX(hipStreamSynchronize(x->sCtl));
        if (x->barrier()) return -1;
        for (int r = 0; r < x->world; r++)
            HIPX(hipMemcpyAsync(dst_dev + (uint64_t) r * bytesPerRank + at, x->area + (uint64_t) r * per, n, hipMemcpyHostToDevice, x->sCtl));
        HIPX(hipStreamSynchronize(x->sCtl));
        if (x->barrier()) return -1;
    }
    return 0;
}

int mbgc_xchg_bcast_heads_begin(mbgc_xchg_t *x, const uint8_t *src_dev, const uint64_t *need, uint64_t stride, uint8_t *dst_dev) {
    HIPX(hipSetDevice(x->device));
    if (x->rccl) {
        NCCLX(ncclGroupStart());
        for (int r = 0; r < x->world; r++)
            if (need[r]) NCCLX(ncclBroadcast(src_dev, dst_dev + (uint64_t) r * stride, need[r], ncclUint8, r, x->bulk, x->sBulk));
        NCCLX(ncclGroupEnd());
        HIPX(hipEventRecord(x->evBytes, x->sBulk));
        return 0;
    }
    uint64_t most = 0;
    for (int r = 0; r < x->world; r++) most = need[r] > most ? need[r] : most;
    const uint64_t per = x->areaBytes / x->world;
    for (uint64_t at = 0; at < most; at += per) {
        const uint64_t mineLeft = need[x->rank] > at ? need[x->rank] - at : 0, n = mineLeft < per ? mineLeft : per;
        if (n) {
            HIPX(hipMemcpyAsync(x->area + (uint64_t) x->rank * per, src_dev + at, n, hipMemcpyDeviceToHost, x->sCtl));
            HIPX(hipStreamSynchronize(x->sCtl));
        }
        if (x->barrier()) return -1;
        for (int r = 0; r < x->world; r++) {
            const uint64_t left = need[r] > at ? need[r] - at : 0, m = left < per ? left : per;
            if (m) HIPX(hipMemcpyAsync(dst_dev + (uint64_t) r * stride + at, x->area + (uint64_t) r * per, m, hipMemcpyHostToDevice, x->sCtl));
        }
        HIPX(hipStreamSynchronize(x->sCtl));
        if (x->barrier()) return -1;
    }
    return 0;
}

int mbgc_xchg_stream_wait_bytes(mbgc_xchg_t *x, void *stream) {
    if (x->rccl) HIPX(hipStreamWaitEvent((hipStream_t) stream, x->evBytes, 0));
    return 0;                                             // (host memory: the all-gather had completed when _begin returned)
}

int mbgc_xchg_wait_bytes(mbgc_xchg_t *x) {
    if (x->rccl) HIPX(hipEventSynchronize(x->evBytes));
    return 0;
}

int mbgc_xchg_allreduce_min_u32(mbgc_xchg_t *x, uint32_t *word_dev, void *stream) {
    HIPX(hipSetDevice(x->device));
    if (x->rccl) {
        NCCLX(ncclAllReduce(word_dev, word_dev, 1, ncclUint32, ncclMin, x->ctl, (hipStream_t) stream));
        // the host will want it too (mbgc_xchg_reduced_u32), without waiting for what is queued behind it
        HIPX(hipMemcpyAsync(x->hWord, word_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t) stream));
        HIPX(hipEventRecord(x->evWord, (hipStream_t) stream));
        return 0;
    }
    uint32_t v = 0;
    HIPX(hipMemcpyAsync(&v, word_dev, sizeof v, hipMemcpyDeviceToHost, (hipStream_t) stream));
    HIPX(hipStreamSynchronize((hipStream_t) stream));
    ((uint32_t *) x->area)[x->rank] = v;
    if (x->barrier()) return -1;
    for (int r = 0; r < x->world; r++) v = ((uint32_t *) x->area)[r] < v ? ((uint32_t *) x->area)[r] : v;
    if (x->barrier()) return -1;
    HIPX(hipMemcpyAsync(word_dev, &v, sizeof v, hipMemcpyHostToDevice, (hipStream_t) stream));
    HIPX(hipStreamSynchronize((hipStream_t) stream));
    *x->hWord = v;
    return 0;
}

int mbgc_xchg_reduced_u32(mbgc_xchg_t *x, uint32_t *out) {
    if (x->rccl) HIPX(hipEventSynchronize(x->evWord));
    *out = *x->hWord;
    return 0;
}

int mbgc_xchg_gather_to_root(mbgc_xchg_t *x, const uint8_t *src_dev, const uint64_t *bytesOfRank, uint8_t *dst_dev) {
    HIPX(hipSetDevice(x->device));
    std::vector<uint64_t> off(x->world + 1, 0);
    for (int r = 0; r < x->world; r++) off[r + 1] = off[r] + bytesOfRank[r];
    if (x->rccl) {
        NCCLX(ncclGroupStart());
        if (x->rank == 0) {
            for (int r = 1; r < x->world; r++)
                if (bytesOfRank[r]) NCCLX(ncclRecv(dst_dev + off[r], bytesOfRank[r], ncclUint8, r, x->bulk, x->sBulk));
        } else if (bytesOfRank[x->rank]) {
            NCCLX(ncclSend(src_dev, bytesOfRank[x->rank], ncclUint8, 0, x->bulk, x->sBulk));
        }
        NCCLX(ncclGroupEnd());
        if (x->rank == 0 && bytesOfRank[0]) HIPX(hipMemcpyAsync(dst_dev, src_dev, bytesOfRank[0], hipMemcpyDeviceToDevice, x->sBulk));
        HIPX(hipStreamSynchronize(x->sBulk));
        return 0;
    }
    uint64_t most = 0;
    for (int r = 0; r < x->world; r++) most = bytesOfRank[r] > most ? bytesOfRank[r] : most;
    const uint64_t per = x->areaBytes / x->world;
    for (uint64_t at = 0; at < most; at += per) {
        const uint64_t mineLeft = bytesOfRank[x->rank] > at ? bytesOfRank[x->rank] - at : 0, n = mineLeft < per ? mineLeft : per;
        if (n) {
            HIPX(hipMemcpyAsync(x->area + (uint64_t) x->rank * per, src_dev + at, n, hipMemcpyDeviceToHost, x->sCtl));
            HIPX(hipStreamSynchronize(x->sCtl));
        }
        if (x->barrier()) return -1;
        if (x->rank == 0) {
            for (int r = 0; r < x->world; r++) {
                const uint64_t left = bytesOfRank[r] > at ? bytesOfRank[r] - at : 0, m = left < per ? left : per;
                if (m) HIPX(hipMemcpyAsync(dst_dev + off[r] + at, x->area + (uint64_t) r * per, m, hipMemcpyHostToDevice, x->sCtl));
            }
            HIPX(hipStreamSynchronize(x->sCtl));
        }
        if (x->barrier()) return -1;
    }
    return 0;
}

}  // extern "C"
