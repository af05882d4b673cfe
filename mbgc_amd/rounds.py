"""Deterministic round schedule of the match-finding path, single- and multi-GPU.

The reference's worker threads (MultipleGenomeMatchingProcessor::processTarget, MGMP.cpp:340-429)
match several targets concurrently against one shared reference and a finalizer loads their
extensions in target order (:433-468). Here a *round* makes that schedule explicit: the targets of a
round take their lock positions at the same pos1, are matched against the frozen reference — shard
`rank` of them on GPU `rank`, no collective — and then every replica loads every extension of the
round, in target order, so all replicas stay bit-identical. The only exchange step is the all-gather
of the extension bytes (RCCL over xGMI; gloo in the CPU tests).

PyTorch is plumbing only: device buffers, the process group and the collective."""
import numpy as np
import torch

NO_LOCK = 2 ** 64 - 1


class RoundRunner:
    def __init__(self, matcher, rank=0, world=1, group=None, device="cuda:0", lazy=True):
        self.m, self.rank, self.world, self.group = matcher, rank, world, group
        self.device = torch.device(device)
        self.lazy = lazy
        self.targets_done = 0

    # ---- exchange -----------------------------------------------------------------------------
    def _allgather_bytes(self, local):
        """local: 1-D uint8 tensor on self.device -> list (by rank) of 1-D uint8 tensors."""
        if self.world == 1:
            return [local]
        import torch.distributed as dist
        n = torch.tensor([local.numel()], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad = torch.zeros(mx, dtype=torch.uint8, device=self.device)
        pad[: local.numel()] = local
        out = torch.empty(self.world * mx, dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return [out[r * mx: r * mx + sizes[r]] for r in range(self.world)]

    # ---- one round ----------------------------------------------------------------------------
    def run_round(self, qbuf, offsets, ext_of=None, min_len=32):
        """qbuf: uint8 device tensor with this rank's contigs back to back (one target = one entry of
        `offsets`; a multi-contig target is passed as consecutive entries by the caller through
        `ext_of`). Every rank must pass the same number of targets. Returns this rank's match counts.

        ext_of(rank_local_index, counts) -> (start, end) byte range of qbuf to append to the reference
        for that target, or None; default: the whole contig (the 99 %-identity regime, where every
        contig passes isContigProperForRefExtension, MGMP_Params.h:179-186)."""
        m = self.m
        nloc = len(offsets) - 1
        ntot = nloc * self.world
        # lock positions: all targets of the round are acquired at the same pos1 (MGMP.cpp:353-358)
        locks = [m.acquire_lock() for _ in range(ntot)]
        mine = [locks[self.rank * nloc + i] for i in range(nloc)]
        m.match_batch_dev(qbuf.data_ptr(), offsets, min_len, mine)
        counts = m.batch_counts()
        # extensions of this rank, then the exchange
        spans = []
        for i in range(nloc):
            sp = (int(offsets[i]), int(offsets[i + 1])) if ext_of is None else ext_of(i, counts)
            spans.append(sp)
        if self.world == 1:
            for i, sp in enumerate(spans):
                self._finalize(qbuf, sp, locks[i])
        else:
            parts = [qbuf[s:e] for (s, e) in [sp for sp in spans if sp is not None]]
            local = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=self.device)
            lens = torch.tensor([0 if sp is None else sp[1] - sp[0] for sp in spans], dtype=torch.int64, device=self.device)
            all_ext = self._allgather_bytes(local)
            all_lens = self._allgather_bytes(lens.view(torch.uint8))
            for r in range(self.world):                       # target order = rank-major inside the round
                ln = all_lens[r].view(torch.int64).tolist()
                pos = 0
                for i in range(nloc):
                    sp = None if ln[i] == 0 else (pos, pos + ln[i])
                    self._finalize(all_ext[r], sp, locks[r * nloc + i])
                    pos += ln[i]
        self.targets_done += ntot
        return counts

    def _finalize(self, buf, span, lock):
        """finalizeParallelProcessingOfTarget for one target (MGMP.cpp:440-457, MBGC_Encoder.cpp:557-562)."""
        m = self.m
        if span is not None and span[1] > span[0]:
            m.load_ref_dev(buf.data_ptr() + span[0], span[1] - span[0], False, True, 0)
        if self.lazy:
            m.load_separator(0)
        m.release_lock(lock)


def round_schedule(n_targets, per_rank, world):
    """Target indices of every round: round r holds targets [r*per_rank*world, ...); rank g owns the g-th
    block of per_rank consecutive targets of the round (file-per-GPU sharding)."""
    per_round = per_rank * world
    rounds = []
    for r0 in range(0, n_targets, per_round):
        rounds.append([list(range(r0 + g * per_rank, min(r0 + (g + 1) * per_rank, n_targets))) for g in range(world)])
    return rounds
