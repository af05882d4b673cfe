/* mbgc_exchange.h — C ABI of the exchange step between the GPUs of one node, for the C++ host.
 *
 * The reference has no counterpart: its workers and its finalizer are threads over ONE matcher in one address space
 * (MultipleGenomeMatchingProcessor::processTargetsParallel, matching/MultipleGenomeMatchingProcessor.cpp:520-555 — a worker's
 * contig reaches the shared reference through loadRef at :441-443, its streams reach the encoder through
 * finalizeParallelProcessingOfTarget at :455 = MBGC_Encoder.cpp:542-564). With one replica of the reference per GPU
 * those two hand-overs become collectives, and this is all of them:
 *
 *   allgather_bytes   the round's reference extensions: every replica loads every target's extension (:441-443)
 *   allgather_i64     the few numbers of a round the ranks must agree on (sizes, first given-up target :382-388)
 *   allreduce_min_u32 the device-side verdicts of the speculative finalize (mbgc_swsem.h, swsem_spec_finalize_t.exchange)
 *   gather_to_root    the emitted streams, to the rank that feeds the host backend (MBGC_Encoder.cpp:542-556)
 *
 * Two transports behind the same calls: RCCL over xGMI (one rank per GPU; two communicators, so that a small exchange
 * never queues behind a bulk one), and host shared memory for rehearsing the protocol with several ranks on ONE GPU
 * (RCCL refuses two ranks on a device) — the bytes and the order of calls are the same, the speed is not. mbgc-hip
 * forks its ranks before anything touches a GPU and hands each a piece of one shared mapping for the bootstrap. */
#ifndef MBGC_EXCHANGE_H
#define MBGC_EXCHANGE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbgc_xchg mbgc_xchg_t;
#define MBGC_XCHG_ID_BYTES 128                       /* sizeof(ncclUniqueId) */

const char *mbgc_xchg_last_error(void);
/* RCCL: rank 0 makes two ids (bulk and control communicator) and gets them to the others by any means */
int mbgc_xchg_unique_ids(uint8_t ids[2 * MBGC_XCHG_ID_BYTES]);
int mbgc_xchg_create_rccl(mbgc_xchg_t **out, const uint8_t ids[2 * MBGC_XCHG_ID_BYTES], int rank, int world, int device);
/* ranks that share a node may give the RCCL exchange a shared mapping as well (as for mbgc_xchg_create_hostmem): the small
 * host-to-host exchanges (mbgc_xchg_allgather_i64) then go through it — a collective on the device would wait for compute
 * units whenever the round's finalize holds them, and the host reading its result with it. Call on every rank, after create. */
int mbgc_xchg_set_host_control(mbgc_xchg_t *x, void *shared, uint64_t sharedBytes);
/* host shared memory: `shared` = a MAP_SHARED mapping of sharedBytes >= mbgc_xchg_hostmem_min_bytes(world), zero-filled
 * before the first rank is created, the same pages in every rank; larger mappings move more bytes per step */
uint64_t mbgc_xchg_hostmem_min_bytes(int world);
int mbgc_xchg_create_hostmem(mbgc_xchg_t **out, void *shared, uint64_t sharedBytes, int rank, int world, int device);
void mbgc_xchg_destroy(mbgc_xchg_t *x);
int mbgc_xchg_rank(const mbgc_xchg_t *x);
int mbgc_xchg_world(const mbgc_xchg_t *x);

/* every rank gives k values, every rank receives world*k (rank-major). Host memory in and out; returns when done.
 * Runs on the exchange's own stream: it does not wait for anything the matcher has queued. */
int mbgc_xchg_allgather_i64(mbgc_xchg_t *x, const int64_t *mine, uint64_t k, int64_t *all);
/* every rank gives bytesPerRank device bytes, dst_dev receives world*bytesPerRank (rank-major). Asynchronous, on the
 * exchange's bulk stream; src and dst must stay untouched until one of the waits below. */
int mbgc_xchg_allgather_bytes_begin(mbgc_xchg_t *x, const uint8_t *src_dev, uint64_t bytesPerRank, uint8_t *dst_dev);
/* the same when only part of every rank's bytes is wanted: rank r gives the first need[r] of its bytes (0: none), dst_dev
 * receives them at r * stride — broadcasts from the ranks that have something to give. After the buffer has wrapped a round
 * can load one window of bytes (its locks stand there, loadRef clips: SlidingWindowSparseEMMatcher.cpp:361-378, :412-417),
 * so only the head of the round's extensions need travel. Asynchronous like _allgather_bytes_begin; same waits. */
int mbgc_xchg_bcast_heads_begin(mbgc_xchg_t *x, const uint8_t *src_dev, const uint64_t *need, uint64_t stride, uint8_t *dst_dev);
int mbgc_xchg_stream_wait_bytes(mbgc_xchg_t *x, void *stream);      /* `stream` (hipStream_t) waits for the last all-gather */
int mbgc_xchg_wait_bytes(mbgc_xchg_t *x);                           /* the host does */
/* *word_dev := min over the ranks, queued on `stream` (a hipStream_t of the caller's) behind what it holds */
int mbgc_xchg_allreduce_min_u32(mbgc_xchg_t *x, uint32_t *word_dev, void *stream);
/* the word the last mbgc_xchg_allreduce_min_u32 produced (waits for the reduction, not for what was queued behind it) */
int mbgc_xchg_reduced_u32(mbgc_xchg_t *x, uint32_t *out);
/* rank r gives bytesOfRank[r] device bytes; on rank 0 dst_dev receives them back to back in rank order. Returns when done. */
int mbgc_xchg_gather_to_root(mbgc_xchg_t *x, const uint8_t *src_dev, const uint64_t *bytesOfRank, uint8_t *dst_dev);

#ifdef __cplusplus
}
#endif
#endif
