"""Deterministic round schedule of the compress hot path, single- and multi-GPU.

The reference's worker threads (MultipleGenomeMatchingProcessor::processTarget, MGMP.cpp:340-429)
match several targets concurrently against one shared reference and a finalizer loads their
extensions in target order (:433-468). A *round* makes that schedule explicit: the targets of a round
take their lock positions at the same pos1 (:353-358), are matched and emitted against the frozen
reference — shard `rank` of them on GPU `rank`, no collective — and then every replica loads every
extension of the round, in target order, so all replicas stay bit-identical. The exchange steps are
the all-gather of the extension bytes and the gather of the emitted stream bytes to rank 0, which
feeds the (unchanged, host-side) PPMd/LZMA backend: RCCL over xGMI on GPUs, gloo in the CPU tests.

PyTorch is plumbing only: device buffers, the process group and the collectives."""
import contextlib
import os
import sys
import time

import numpy as np
import torch

NO_LOCK = 2 ** 64 - 1
SKIPPED = 2 ** 64 - 1
SEQ_SEPARATOR = 0xA2        # MBGC_Params.h:46, appended to the literals after every contig (MBGC_Encoder.cpp:489-491)
FILE_SEPARATOR = 0xBB       # MBGC_Params.h:47, appended to the flags after every target (:493-496)
STREAMS = ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags")


class Policy:
    """Reference-extension predicates, matching/MGMP_Params.h:175-196, with the -m presets of
    mbgccoder/MBGC_Params.h:886-922."""

    def __init__(self, mode=1):
        self.mode = mode
        self.factor = 128                            # currentUnmatchedFractionFactor
        self.rc_factor = 128 if mode >= 2 else 8     # unmatchedFractionRCFactor

    def proper_for_ext(self, n, unmatched): return unmatched * self.factor > n
    def proper_for_rc_ext(self, n, unmatched): return unmatched * self.rc_factor > n


def frugal64(v):
    """PgHelpers::writeUInt64Frugal, utils/helper.cpp:237-246."""
    out = int(min(v, 0xFFFF)).to_bytes(2, "little")
    if v >= 0xFFFF:
        out += int(min(v, 0xFFFFFFFF)).to_bytes(4, "little")
        if v >= 0xFFFFFFFF:
            out += int(v).to_bytes(8, "little")
    return out


class _Works:
    """several asynchronous collectives waited for as one"""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class RoundRunner:
    """matcher: mbgc_amd.binding.SlidingWindowSparseEMMatcher (or any object with the same surface)."""

    def __init__(self, matcher, rank=0, world=1, group=None, device="cuda:0", lazy=True, emit_params=None,
                 policy=None, keep_streams=True):
        self.m, self.rank, self.world, self.group = matcher, rank, world, group
        self.device = torch.device(device)
        # the exchange path is taken with several ranks — or with one, on request (MBGC_ROUNDS_FORCE_EXCHANGE=1): every
        # collective of the protocol then runs, over the backend at hand, with this rank as the only party (how RCCL itself
        # is exercised on a one-GPU box)
        self.multi = world > 1 or os.environ.get("MBGC_ROUNDS_FORCE_EXCHANGE", "0") == "1"
        self.trace = {} if os.environ.get("MBGC_ROUNDS_TRACE") else None      # host seconds per part of run_round (diagnostics)
        # The handful of numbers a round's ranks exchange travel between the HOSTS (a gloo group over loopback: the ranks of
        # this protocol share a node): a collective on the device would have to wait for compute units whenever the round's
        # finalize holds them, and the host reading its result with it (1.1 ms per round, measured with one rank).
        self._ctl_group, self._ctl_dev = group, self.device
        if self.multi and self.device.type == "cuda" and os.environ.get("MBGC_ROUNDS_HOST_CONTROL", "1") != "0":
            import torch.distributed as dist
            try:
                if dist.get_backend(group) != "gloo":
                    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                    self._ctl_group = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group), backend="gloo")
                self._ctl_dev = torch.device("cpu")
            except Exception as e:                   # (no host-side group to be had: the device-side exchanges still work)
                print("mbgc_amd.rounds: no gloo control group (%s); small exchanges stay on the device" % e, file=sys.stderr)
                self._ctl_group, self._ctl_dev = group, self.device
        # The protocol mixes the handle's launches with torch operations on the same buffers (reverse complements
        # written by the handle and concatenated by torch, concatenations read by the handle's copies): both must
        # run on one stream, or they race.
        if self.device.type == "cuda" and hasattr(matcher, "set_stream"):
            matcher.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.lazy = lazy
        self.p = emit_params                         # None: matcher only, every contig extends the reference
        self.policy = policy or Policy()
        self.keep_streams = keep_streams
        self.targets_done = 0                        # processedTargetsCount
        self.loaded = None                           # refExtLoadedPosArr, MBGC_Encoder.cpp:789-791
        self.streams = {k: bytearray() for k in STREAMS}     # merged in target order on rank 0 (:542-556)
        self.locks_stream = bytearray()              # :563
        self.ref_ext_sizes = bytearray()             # :559-560
        self.stream_bytes = 0
        self._keep = []                              # temporaries handed to finalize_targets: alive until the next host wait
        self._deferred = None                        # emission whose streams have not been collected yet
        self._pack_buf = None
        self._pred_ext = None                        # what every contig of the last round decided about extending the reference
        self._gpred = False                          # several ranks: last round every target on every rank was loaded whole, without RC
        self._pre = None                             # extension all-gather started ahead under that prediction (see _pregather)
        self.pregathers = [0, 0]                     # started / used (diagnostics)
        self.head_gathers = [0, 0, 0]                # extension exchanges cut to the round's loadable head: how many, bytes asked for, bytes of the whole round
        self._gathers = []                           # stream gathers still running (work, output, input)
        self._next_key = None                        # the buffer this rank announced for the next round (see _post_announce)
        self._ann = None                             # the announcement exchange in flight
        self._ctl = None                             # its stream
        self._bulk = None                            # the stream early extension all-gathers are issued from
        self._ev_tops = [None, None]                 # main-stream events: where the last round / this round began
        self._gate = self._gate_host = self._gate_ev = None     # the speculative finalize's word, several ranks (see _world_speculation)
        self.spec_rounds = [0, 0]                    # several ranks: speculative finalizes tried / applied (diagnostics)
        self.spec_local = [0, 0, []]                 # one rank: the same, and the rounds in which it was not applied
        if self.p is not None:
            matcher.emit_set_host_copy(False)

    def start(self):
        """call after G0 has been loaded (encode(), MBGC_Encoder.cpp:789-791)"""
        self.loaded = [self.m.loading_position()]

    # ---- exchange -----------------------------------------------------------------------------
    def _allgather_bytes(self, local):
        """local: 1-D uint8 tensor on self.device -> list (by rank) of 1-D uint8 tensors."""
        if not self.multi:
            return [local]
        import torch.distributed as dist
        n = torch.tensor([local.numel()], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        mx = max(max(sizes), 1)
        pad = torch.zeros(mx, dtype=torch.uint8, device=self.device)
        pad[: local.numel()] = local
        out = torch.empty(self.world * mx, dtype=torch.uint8, device=self.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return [out[r * mx: r * mx + sizes[r]] for r in range(self.world)]

    def _allgather_ints(self, vals, fixed=False):
        """fixed: every rank is known to pass the same number of values (one collective instead of sizes + data). The
        values come from the host and go back to it: over the control group (host to host) when there is one, else on the
        side stream, so that reading the result waits for this exchange alone and not for what the main stream holds."""
        import torch.distributed as dist
        with self._side():
            t = torch.tensor(list(vals), dtype=torch.int64, device=self._ctl_dev)
            if not self.multi:
                return [t.tolist()]
            k = t.numel()
            if not fixed:
                n = torch.tensor([k], dtype=torch.int64, device=self._ctl_dev)
                ns = torch.empty(self.world, dtype=torch.int64, device=self._ctl_dev)
                dist.all_gather_into_tensor(ns, n, group=self._ctl_group)
                ns = ns.tolist()
                k = max(max(ns), 1)
                pad = torch.zeros(k, dtype=torch.int64, device=self._ctl_dev)
                pad[: t.numel()] = t
                t = pad
            else:
                ns = [k] * self.world
            out = torch.empty(self.world * k, dtype=torch.int64, device=self._ctl_dev)
            dist.all_gather_into_tensor(out, t, group=self._ctl_group)
            flat = out.tolist()
            return [flat[r * k: r * k + ns[r]] for r in range(self.world)]

    def _side(self):
        """the stream of the small exchanges that must not wait for what the main stream has queued"""
        if self.device.type != "cuda":
            return contextlib.nullcontext()
        if self._ctl is None:
            self._ctl = torch.cuda.Stream(self.device)
        return torch.cuda.stream(self._ctl)

    def _post_announce(self, next_batch, T):
        """Several ranks: what this rank's NEXT round looks like — bytes of its query buffer, bytes of every target —
        travels now, on a stream of its own, so that nothing at the top of the next round has to wait for an exchange:
        its extension all-gather and its speculative finalize are set up from these numbers. Every rank takes part in
        every round (-1 = nothing to say); next_batch = (buffer, offsets[, target of every contig])."""
        self._next_key = None
        if not self.multi:
            return
        import torch.distributed as dist
        vals = [-1] * (2 + T)
        if next_batch is not None:
            nb, no = next_batch[0], next_batch[1]
            nt = list(next_batch[2]) if len(next_batch) > 2 and next_batch[2] is not None else list(range(len(no) - 1))
            if int(no[0]) == 0 and int(no[-1]) == nb.numel() and all(nt[c] <= nt[c + 1] for c in range(len(nt) - 1)):
                vals[0], self._next_key = int(nb.numel()), (nb.data_ptr(), nb.numel())
                tn = max(nt) + 1 if nt else 0
                if tn <= T:
                    lens = [0] * tn
                    for c, t in enumerate(nt):
                        lens[t] += int(no[c + 1]) - int(no[c])
                    vals[1], vals[2: 2 + tn] = tn, lens
        with self._side():
            t = torch.tensor(vals, dtype=torch.int64, device=self._ctl_dev)
            out = torch.empty(self.world * len(vals), dtype=torch.int64, device=self._ctl_dev)
            work = dist.all_gather_into_tensor(out, t, group=self._ctl_group, async_op=True)
        self._ann = (work, out, t, len(vals))

    def _take_announce(self):
        """-> per rank (bytes of its buffer or -1, bytes per target or None) as announced for the round that starts now"""
        a, self._ann = self._ann, None
        if a is None:
            return None
        work, out, _, k = a
        with self._side():
            work.wait()
            flat = out.tolist()
        return [(flat[r * k], flat[r * k + 2: r * k + 2 + flat[r * k + 1]] if flat[r * k + 1] >= 0 else None)
                for r in range(self.world)]

    def _loadable(self, locks):
        """bytes of the round's extensions (target after target) that can be loaded at all, or None = all of them: once the
        buffer has wrapped every target of the round holds the lock value loading position + window (.cpp:361-378), loadRef
        clips at the oldest outstanding lock (:412-417, the rest of a text is dropped, :433) and the round's locks are
        released one by one as its targets are loaded — so whatever lies beyond `window` bytes of the round is never read.
        The same on every rank (the replicas' states are)."""
        if not locks or os.environ.get("MBGC_ROUNDS_GATHER_ALL", "0") == "1":
            return None
        m = self.m
        mx, pos1, lock = int(m.max_ref_length()), int(m.loading_position()), int(min(locks))
        if lock >= mx or int(max(locks)) != lock:
            return None
        cap = lock - pos1 if lock > pos1 else lock + (mx - 1) - pos1
        return cap + 4096 if cap > 0 else None

    def _pregather(self, qbuf, offsets, targets, T, ann, loadable=None):
        """Several ranks: in a collection nearly every target ends up loaded into the reference whole, so the bytes the
        round's all-gather will carry are known before the round starts — they are the queries. When the last round went
        that way on every rank (a fact all ranks hold, so all of them decide alike), the all-gather is started here,
        asynchronously, and runs beside match-finding instead of after it; _finalize_range uses its result if this
        round's decisions come out the same on every rank, and falls back to the ordinary exchange otherwise."""
        self._pre = None
        announced = [a[0] for a in ann] if ann is not None and min(a[0] for a in ann) >= 0 else None
        if not self.multi or not self._gpred or self.p is None or os.environ.get("MBGC_ROUNDS_PREGATHER", "1") == "0":
            return
        ncont = len(offsets) - 1
        lens = None
        usable = not (int(offsets[0]) != 0 or int(offsets[-1]) != qbuf.numel() or
                      any(targets[c] > targets[c + 1] for c in range(ncont - 1)))
        import torch.distributed as dist
        poisoned = False
        if announced is not None:
            # every rank told the others last round how many bytes its next buffer holds: no exchange (and no wait for
            # the device) at the top of the round. A rank whose buffer is not the announced one still takes part in the
            # collective — the others will — and says so in the length exchange, which sends everybody down the ordinary path.
            sizes = announced
            poisoned = not usable or self._next_key != (qbuf.data_ptr(), qbuf.numel()) or qbuf.numel() != sizes[self.rank]
            # every rank's bytes per target, if all of them named T targets that add up to their buffers: what the
            # speculative finalize of the round needs (_world_speculation)
            if all(a[1] is not None and len(a[1]) == T and sum(a[1]) == a[0] and min(a[1], default=0) > 0 for a in ann):   # (a target without bytes: its rank may have nothing to emit, and then cannot take part)
                lens = [a[1] for a in ann]
                mine = [0] * T
                for c, t in enumerate(targets):
                    mine[t] += int(offsets[c + 1]) - int(offsets[c])
                poisoned = poisoned or mine != lens[self.rank]
        else:
            # (whether this rank's buffer has the expected layout is local knowledge: it travels with the sizes, so that
            # every rank enters the big collective or none does)
            n = torch.tensor([qbuf.numel() if usable else -1], dtype=torch.int64, device=self.device)
            sizes = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(sizes, n, group=self.group)
            sizes = [int(x.item()) for x in sizes]
            if min(sizes) < 0:
                return
        mx = max(max(sizes), 1)
        if announced is None:
            lens = None
        # A buffer that was named a round ago existed then: its all-gather need not queue behind what the main stream
        # holds now (the last round's finalize) — it is issued from a stream that only waits for the point where the last
        # round began, and runs beside that finalize and this round's match-finding from the start.
        early = (self.device.type == "cuda" and announced is not None and not poisoned and self._ev_tops[0] is not None and
                 os.environ.get("MBGC_ROUNDS_EARLY_GATHER", "1") != "0")
        if early:
            if self._bulk is None:
                self._bulk = torch.cuda.Stream(self.device)
            self._bulk.wait_event(self._ev_tops[0])
        with (torch.cuda.stream(self._bulk) if early else contextlib.nullcontext()):
            if qbuf.numel() == mx and not poisoned:
                pad = qbuf
            else:
                pad = torch.zeros(mx, dtype=torch.uint8, device=self.device)
                if not poisoned:
                    pad[: qbuf.numel()] = qbuf
            out = torch.empty(self.world * mx, dtype=torch.uint8, device=self.device)
            need = None
            if loadable is not None and announced is not None:
                # only the head of the round can be loaded (see _loadable): rank r's bytes are wanted as far as they lie
                # inside it, and travel as broadcasts from the few ranks that hold them instead of an all-gather of all
                start, need = 0, []
                for r in range(self.world):
                    need.append(max(0, min(sizes[r], loadable - start)))
                    start += sizes[r]
            if need is not None and sum(need) < sum(sizes):
                works = []
                for r in range(self.world):
                    if need[r] == 0:
                        continue
                    piece = out[r * mx: r * mx + need[r]]
                    if r == self.rank:
                        piece.copy_(pad[: need[r]])
                    works.append(dist.broadcast(piece, src=r if self.group is None else dist.get_global_rank(self.group, r),
                                                group=self.group, async_op=True))
                work = _Works(works)
                self.head_gathers[0] += 1
                self.head_gathers[1] += sum(need)
                self.head_gathers[2] += sum(sizes)
            else:
                work = dist.all_gather_into_tensor(out, pad, group=self.group, async_op=True)
        if early:                                       # (allocated under the side stream, consumed on the main one)
            main = torch.cuda.current_stream(self.device)
            out.record_stream(main)
            if pad is not qbuf:
                pad.record_stream(main)
        self._pre = dict(work=work, out=out, pad=pad, mx=mx, sizes=sizes, poisoned=poisoned, lens=lens)
        self.pregathers[0] += 1

    # ---- one round ----------------------------------------------------------------------------
    def run_round(self, qbuf, offsets, targets=None, min_len=32, next_batch=None):
        """qbuf: uint8 device tensor holding this rank's contigs of the round back to back, contig c at
        [offsets[c], offsets[c+1]). targets[c] = index (0..T-1) of the local target contig c belongs to
        (default: one contig per target). Every rank passes the same number of targets T; globally the
        round's targets are ordered rank-major. Returns this rank's match counts per contig.
        next_batch = (qbuf, offsets) of the round that follows, if its bytes are already in HBM: with several ranks its
        size rides on this round's length exchange, so that the next round's extension all-gather starts without one."""
        m = self.m
        ncont = len(offsets) - 1
        targets = list(range(ncont)) if targets is None else list(targets)
        T = max(targets) + 1 if targets else 0
        ntot = T * self.world
        first = self.targets_done                   # global index of the round's first target
        locks = [m.acquire_lock() for _ in range(ntot)]                     # MGMP.cpp:353-358
        if self.multi and self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()                                 # "this round begins here" on the main stream (see _pregather)
            self._ev_tops = [self._ev_tops[1], ev]
        tr = self.trace
        t0 = time.perf_counter() if tr is not None else 0
        self._pregather(qbuf, offsets, targets, T, self._take_announce(), self._loadable(locks))
        self._post_announce(next_batch, T)
        if tr is not None:
            tr["top"] = tr.get("top", 0) + time.perf_counter() - t0
        lock_of = [locks[self.rank * T + targets[c]] for c in range(ncont)]
        pending = list(range(ncont))                # contigs still to be matched + emitted
        counts = np.zeros(ncont, dtype=np.uint64)
        unmatched = [None] * ncont
        packs = []                                  # (contig ids, sizes, device tensor) of accepted emissions
        finalized = 0                               # targets of the round already loaded into the reference
        emitted_here = False                        # an emission of THIS round has been begun (the deferred one is then "previous")
        ext_done = {}
        first_pass = True                           # rank-invariant: every rank is in its first pass over the whole round
        first_pass_local = True
        spec_applied = False
        cuts = {}                                   # local targets still to be matched again (-> their first contig)
        stopped_all = set()                         # the round's targets (global index) that gave a contig up in the first pass
        any_cut = False
        while True:
            if pending:
                t0 = time.perf_counter() if tr is not None else 0
                self._match(qbuf, [(int(offsets[c]), int(offsets[c + 1])) for c in pending],
                            [lock_of[c] for c in pending], min_len)
                if tr is not None:
                    t1 = time.perf_counter()
                    tr["match"] = tr.get("match", 0) + t1 - t0
                spec_applied = False
                if self.p is not None:
                    tgt = [first + self.rank * T + targets[c] for c in pending]
                    spec = self._speculation(qbuf, offsets, targets, T, pending, finalized, ncont)
                    wspec = self._world_speculation(T) if first_pass else None
                    if wspec is not None:
                        # several ranks: the finalize of ALL the round's targets, read from the extension all-gather that
                        # was started ahead, is queued behind pass 1 on every rank and runs iff every rank's pass 1 finds
                        # what was predicted — the ranks' verdicts are reduced on the stream, no host in between
                        before = m.loaded_ref_length()
                        self.spec_rounds[0] += 1
                        spec_applied, after = m.emit_batch_begin_spec(
                            self.p, [lock_of[c] for c in pending], [self.policy.factor] * ncont, [self.targets_done] * ncont, tgt,
                            self.loaded, ncont, wspec[0], wspec[1], locks, [True] * ncont, [False] * ncont,
                            self.policy.factor, self.policy.rc_factor, self.lazy, gate=self._gate.data_ptr(),
                            reduce=self._reduce_gate, verdict=self._gate_verdict, veto=self._pre["poisoned"])
                        if spec_applied:
                            self.spec_rounds[1] += 1
                            self.pregathers[1] += 1
                            self._keep.append(self._pre["out"])
                            self._pre = None
                    elif spec is not None:
                        # the round's finalize is queued behind pass 1 under the prediction "every contig decides as the
                        # last round's did"; the library applies it only if that is what pass 1 finds
                        before = m.loaded_ref_length()
                        spec_applied, after = m.emit_batch_begin_spec(
                            self.p, [lock_of[c] for c in pending], [self.policy.factor] * ncont, [self.targets_done] * ncont, tgt,
                            self.loaded, ncont, spec[0], spec[1], locks, [self._pred_ext] * ncont, [False] * ncont,
                            self.policy.factor, self.policy.rc_factor, self.lazy)
                        self.spec_local[0] += 1
                        if spec_applied:
                            self.spec_local[1] += 1
                        else:
                            self.spec_local[2].append(self.spec_local[0])
                    else:
                        m.emit_batch_begin(self.p, None, [lock_of[c] for c in pending], [self.policy.factor] * len(pending),
                                           [self.targets_done + finalized] * len(pending), tgt, self.loaded, n=len(pending))
                    un = m.emit_unmatched(len(pending))
                    if tr is not None:
                        tr["emit_begin"] = tr.get("emit_begin", 0) + time.perf_counter() - t1
                    emitted_here = True
                else:
                    un = [int(offsets[c + 1] - offsets[c]) for c in pending]     # matcher only: always extend
                cnt = m.batch_counts()          # after the emission launches: no host round trip in between
                self._keep.clear()              # the waits above cover every copy queued by earlier finalize calls
                for k, c in enumerate(pending):
                    counts[c] = cnt[k]
                # contigs given up as dissimilar (MGMP.cpp:382-388): only the first pass over the whole round meets them — the
                # round then goes on in units whose workers start with every earlier target loaded (see below)
                new_cuts = {}
                for k, c in enumerate(pending):
                    if int(un[k]) == SKIPPED:
                        new_cuts.setdefault(targets[c], c)
                assert first_pass_local or not new_cuts, "a contig was given up although every target in front of its unit had been loaded"
                for k, c in enumerate(pending):
                    if int(un[k]) != SKIPPED:
                        unmatched[c] = int(un[k])
                cuts.update(new_cuts)
                any_cut = any_cut or bool(new_cuts)
                emitted_now = (list(pending), len(pending)) if self.p is not None else None
            else:
                emitted_now = None
            # the first target (global order) still to be matched (again): the finalizer gets that far
            first_skip_local = min([self.rank * T + lt for lt in cuts], default=ntot)
            merged = None
            if self.multi and first_pass and spec_applied:
                first_skip = ntot                   # every rank's pass 1 came out as predicted: there is nothing to tell
            elif self.multi and first_pass:
                # first pass over the whole round: the skip index travels together with what this rank would load if
                # nobody skips (one exchange instead of two on the path between pass 1 and the round's finalize)
                if cuts:
                    pieces, whole = None, False
                else:
                    pieces, whole = self._build_pieces(qbuf, offsets, targets, T, unmatched, 0, ntot)
                got = self._allgather_ints([first_skip_local] + ([x[1] for x in pieces] if pieces is not None else [0] * T) +
                                           [1 if whole else 0] + [1 if lt in cuts else 0 for lt in range(T)], fixed=True)
                first_skip = min(v[0] for v in got)
                stopped_all = {r * T + lt for r, v in enumerate(got) for lt in range(T) if v[2 + T + lt]}
                if first_skip == ntot:
                    merged = (pieces, [v[1: 1 + T] for v in got], [v[1 + T] for v in got])
            elif self.multi:
                first_skip = min(x[0] for x in self._allgather_ints([first_skip_local], fixed=True))
            else:
                first_skip = first_skip_local
                if first_pass_local:
                    stopped_all = set(cuts)
            if first_pass_local and first_skip < ntot:
                # the first pass met dissimilar contigs: the targets that hold one are void, whole (a worker that starts its
                # target again when its turn has come); the others keep what was found. The finalizer takes the targets in order;
                # the stopped targets that follow each other — at most allowedTargetsOutrunForDissimilarContigs + 1, a unit — are
                # matched again with every target in front of the unit loaded (processMatches then gives nothing up,
                # MBGC_Encoder.cpp:203) and loaded before the finalizer goes on.
                cuts = {}
                for c in range(ncont):
                    if self.rank * T + targets[c] in stopped_all:
                        unmatched[c] = None
                        counts[c] = 0
                        cuts.setdefault(targets[c], c)
            first_pass_local = False
            first_pass = False
            upto = first_skip                       # targets [finalized, upto) are complete on every rank
            last = None
            if emitted_now is not None:
                kept = [(k, c) for k, c in enumerate(emitted_now[0]) if unmatched[c] is not None]
                if kept:
                    last = ([k for k, _ in kept], [c for _, c in kept], emitted_now[1])
            if last is not None and first_skip < ntot:
                packs.append(self._pack(*last))     # another pass follows: it reuses the emission's buffers, take the streams now
                last = None
            redo = []
            if first_skip < ntot:
                unit = (self.p.allowedTargetsOutrunForDissimilarContigs if self.p is not None else 0) + 1
                run = [first_skip]
                while len(run) < unit and run[-1] + 1 in stopped_all:
                    run.append(run[-1] + 1)
                for j in run:
                    lt = j - self.rank * T
                    if j // T == self.rank and lt in cuts:
                        redo += [c for c in range(ncont) if targets[c] == lt]
                        del cuts[lt]
            if pending and spec_applied:            # (implies: no skip, the whole round) only the bookkeeping is left
                self._note_finalized(locks, after, before)
            else:
                self._finalize_range(qbuf, offsets, targets, T, locks, unmatched, finalized, upto, ext_done, merged=merged)
            finalized = upto
            self._learn(offsets, unmatched, any_cut)
            # the previous round's streams: the second phase of its emission ran beside everything above (two
            # emissions may be in flight) and is collected only now, with this round's finalize already queued
            if self._deferred is not None:
                t0 = time.perf_counter() if tr is not None else 0
                self.flush(previous=self.p is not None and emitted_here)
                if tr is not None:
                    tr["flush"] = tr.get("flush", 0) + time.perf_counter() - t0
            if finalized >= ntot:
                break
            # the next unit
            pending = sorted(redo)
            for c in pending:
                unmatched[c] = None
            packs = [self._drop(pk, pending) for pk in packs]
        if self.p is not None:
            # the streams of the last emission are taken later (flush): until then its second phase runs
            # beside the finalize queued above and the next round's match-finding
            self._deferred = (packs, last, targets, T, offsets, qbuf)      # qbuf: read by the emission until then
        self.targets_done += ntot
        return counts

    def _wait_gathers(self):
        for work, _, _ in self._gathers:
            work.wait()
        self._gathers = []

    def flush(self, previous=False):
        """waits for the emission whose streams are still to be collected (if any) and merges them; call after
        the last round. previous: a newer emission has been begun since (run_round's own call)."""
        d, self._deferred = self._deferred, None
        if d is None:
            if not previous:
                self._wait_gathers()
            return
        packs, last, targets, T, offsets, _ = d
        tr = self.trace
        t0 = time.perf_counter() if tr is not None else 0
        if last is not None:
            if previous:
                self.m.emit_select(True)
            packs.append(self._pack(*last, reuse=(not self.multi and not packs)))
            if previous:
                self.m.emit_select(False)
        if tr is not None:
            tr["flush.pack"] = tr.get("flush.pack", 0) + time.perf_counter() - t0
        self._collect_streams(packs, targets, T, offsets)
        if not previous:
            self._wait_gathers()

    def _match(self, qbuf, spans, locks, min_len):
        """match contigs given as byte spans of qbuf. match_batch_dev takes offsets of back-to-back
        contigs; arbitrary spans are expressed relative to the lowest start."""
        contiguous = all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        if contiguous:
            base = spans[0][0]
            offs = np.array([s - base for s, _ in spans] + [spans[-1][1] - base], dtype=np.uint64)
            self.m.match_batch_dev(qbuf.data_ptr() + base, offs, min_len, locks)
        else:
            parts = [qbuf[s:e] for s, e in spans]
            self._tmp = torch.cat(parts)
            offs = np.zeros(len(spans) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([e - s for s, e in spans])
            self.m.match_batch_dev(self._tmp.data_ptr(), offs, min_len, locks)

    def _pack(self, ks, cs, n_emitted, reuse=False):
        sizes, total = self.m.emit_pack_sizes(n_emitted)
        if not self.multi and not self.keep_streams:
            # nobody takes the bytes over: they stay where the emission packed them (the handle's arena), only
            # their sizes are accounted
            return dict(cs=cs, ks=ks, sizes=sizes, starts=None, buf=None)
        if reuse:
            # one grow-only buffer for the common case (one emission per round, consumed before the next one is
            # packed): allocating per round goes through the caching allocator, whose occasional hipMalloc stalls
            # the whole device for milliseconds
            if self._pack_buf is None or self._pack_buf.numel() < max(total, 1):
                self._pack_buf = torch.empty(max(int(total * 1.5), 1 << 20), dtype=torch.uint8, device=self.device)
            buf = self._pack_buf
        else:
            buf = torch.empty(max(total, 1), dtype=torch.uint8, device=self.device)
        if self.multi and self.device.type == "cuda":
            # queued on the main stream, where the gather that consumes it is ordered: a copy waited for here would wait for
            # compute units while the round's finalize holds them, and the host with it
            self.m.emit_pack_dev(buf.data_ptr(), buf.numel(), stream=torch.cuda.current_stream(self.device).cuda_stream)
        else:
            self.m.emit_pack_dev(buf.data_ptr(), buf.numel())
        starts = np.zeros(n_emitted * 6 + 1, dtype=np.int64)
        starts[1:] = np.cumsum(sizes.reshape(-1))
        return dict(cs=cs, ks=ks, sizes=sizes, starts=starts, buf=buf)

    @staticmethod
    def _drop(pk, redo):
        keep = [i for i, c in enumerate(pk["cs"]) if c not in redo]
        pk["cs"] = [pk["cs"][i] for i in keep]
        pk["ks"] = [pk["ks"][i] for i in keep]
        return pk

    def _build_pieces(self, qbuf, offsets, targets, T, unmatched, lo, hi):
        """extension string of every local target in [lo, hi): contig, then its reverse complement (MGMP.cpp:389-398).
        Returns (pieces, whole): per target (device pointer, bytes, tensor keeping them alive or None, offset in qbuf or -1);
        whole = the extensions are exactly this rank's queries, target after target (what _pregather assumed)."""
        m = self.m
        my = [t for t in range(lo, hi) if t // T == self.rank] if self.multi else range(lo, hi)
        by_target = {}
        for c, lt in enumerate(targets):
            by_target.setdefault(lt, []).append(c)
        base_ptr = qbuf.data_ptr()
        pieces = []
        for t in my:
            parts = []
            for c in by_target.get(t - self.rank * T, ()):
                s, e = int(offsets[c]), int(offsets[c + 1])
                n, un = e - s, unmatched[c]
                if self.policy.proper_for_ext(n, un):
                    parts.append((base_ptr + s, n, None, s))
                if self.p is not None and self.policy.proper_for_rc_ext(n, un):
                    rc = torch.empty(n, dtype=torch.uint8, device=self.device)
                    m.revcomp_dev(base_ptr + s, n, rc.data_ptr())
                    parts.append((rc.data_ptr(), n, rc, -1))
            if len(parts) == 1:
                pieces.append(parts[0])
            elif not parts:
                pieces.append((0, 0, None, -1))
            else:                                   # several strings of one target are loaded as one text
                ext = torch.cat([x[2] if x[2] is not None else qbuf[x[3]: x[3] + x[1]] for x in parts])
                pieces.append((ext.data_ptr(), ext.numel(), ext, -1))
        whole = (self.multi and lo == 0 and hi == T * self.world and len(pieces) == T and
                 all(x[2] is None and x[1] > 0 for x in pieces) and sum(x[1] for x in pieces) == qbuf.numel() and
                 all(pieces[i][3] + pieces[i][1] == pieces[i + 1][3] for i in range(len(pieces) - 1)) and pieces[0][3] == 0)
        if self._pre is not None and self._pre.get("poisoned"):
            whole = False
        return pieces, whole

    def _finalize_range(self, qbuf, offsets, targets, T, locks, unmatched, lo, hi, ext_done, merged=None):
        """finalizeParallelProcessingOfTarget for the round's targets [lo, hi) in order (MGMP.cpp:433-468). merged: the
        whole round's pieces and what every rank said about its own (run_round exchanged them together with the skip
        index), instead of an exchange here."""
        if hi <= lo:
            return
        if merged is not None:
            pieces, all_lens, flags = merged
        else:
            pieces, whole = self._build_pieces(qbuf, offsets, targets, T, unmatched, lo, hi)
        # finalize_targets returns with its copies queued: their sources must outlive this function
        self._keep.extend(x[2] for x in pieces if x[2] is not None)
        if not self.multi:
            self._finalize_many([x[0] for x in pieces], [x[1] for x in pieces], [locks[t] for t in range(lo, hi)])
            return
        whole_round = lo == 0 and hi == T * self.world                                     # then: T targets on every rank
        if merged is None:
            got = self._allgather_ints([x[1] for x in pieces] + [1 if whole else 0, -1],
                                       fixed=whole_round)                                  # (-1 keeps the tensor non-empty)
            flags = [l[-2] for l in got]
            all_lens = [l[:-2] for l in got]
        pre, self._pre = self._pre, None
        self._gpred = all(flags) and whole_round
        if pre is not None:
            pre["work"].wait()                       # (always: the collective was entered by every rank)
        if pre is not None and all(flags) and all(sum(l) == pre["sizes"][r] for r, l in enumerate(all_lens)):
            all_ext = [pre["out"][r * pre["mx"]: r * pre["mx"] + pre["sizes"][r]] for r in range(self.world)]
            self._keep.append(pre["out"])
            self.pregathers[1] += 1
        else:
            tens = [x[2] if x[2] is not None else qbuf[x[3]: x[3] + x[1]] for x in pieces if x[1]]
            local = torch.cat(tens) if tens else torch.empty(0, dtype=torch.uint8, device=self.device)
            all_ext = self._allgather_bytes(local)
        self._keep.extend(all_ext)
        cur = [0] * self.world
        idx = [0] * self.world
        ptrs, lens = [], []
        for t in range(lo, hi):                                # global target order = rank-major in the round
            r = t // T
            ln = all_lens[r][idx[r]]
            ptrs.append(all_ext[r].data_ptr() + cur[r] if ln else 0)
            lens.append(ln)
            cur[r] += ln
            idx[r] += 1
        self._finalize_many(ptrs, lens, [locks[t] for t in range(lo, hi)])

    def _speculation(self, qbuf, offsets, targets, T, pending, finalized, ncont):
        """(ext pointers, ext lengths) per target under the prediction, or None when this emission cannot carry a
        speculative finalize: not the whole round on a single GPU, no uniform prediction yet, or targets whose
        contigs are not one contiguous span of qbuf."""
        if (self.multi or self._pred_ext is None or finalized or len(pending) != ncont or
                not hasattr(self.m, "emit_batch_begin_spec")):
            return None
        ptrs, lens, base = [0] * T, [0] * T, qbuf.data_ptr()
        span = {}
        for c, t in enumerate(targets):
            s, e = int(offsets[c]), int(offsets[c + 1])
            if t in span:
                if span[t][1] != s:
                    return None
                span[t][1] = e
            else:
                span[t] = [s, e]
        if self._pred_ext:
            for t, (s, e) in span.items():
                ptrs[t], lens[t] = (base + s, e - s) if e > s else (0, 0)
        return ptrs, lens

    def _world_speculation(self, T):
        """Several ranks: (ext pointers, ext lengths) of all the round's targets, rank-major, inside the output of the
        extension all-gather started ahead (_pregather) — under the prediction that made it start: every target of
        every rank is loaded whole, without reverse complement. None when the round cannot carry a speculative finalize;
        the conditions are facts every rank holds, so that all of them enter the verdict's reduction or none does (a
        rank whose own buffer is not what it announced takes part with a veto)."""
        pre = self._pre
        if not self.multi or pre is None or pre["lens"] is None or not hasattr(self.m, "emit_batch_begin_spec"):
            return None
        if self._gate is None:
            self._gate = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._gate_host = torch.zeros(1, dtype=torch.int32)
            if self.device.type == "cuda":
                self._gate_host = self._gate_host.pin_memory()
        ptrs, lens, base = [], [], pre["out"].data_ptr()
        for r in range(self.world):
            cur = base + r * pre["mx"]
            for ln in pre["lens"][r]:
                ptrs.append(cur if ln else 0)
                lens.append(ln)
                cur += ln
        return ptrs, lens

    def _reduce_gate(self, stream):
        """called by the library between the device-side check of this rank's prediction and the launches it gates:
        they must see every rank's verdict, and the bytes they copy — both are made to precede them on the stream"""
        import torch.distributed as dist
        cuda = self.device.type == "cuda"
        on = (torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device))
              if cuda and stream and stream != torch.cuda.current_stream(self.device).cuda_stream else contextlib.nullcontext())
        with on:
            self._pre["work"].wait()
            dist.all_reduce(self._gate, op=dist.ReduceOp.MIN, group=self.group)
            self._gate_host.copy_(self._gate, non_blocking=True)
            if cuda:
                self._gate_ev = torch.cuda.Event()
                self._gate_ev.record()

    def _gate_verdict(self):
        if self._gate_ev is not None:
            self._gate_ev.synchronize()
        return int(self._gate_host[0])

    def _learn(self, offsets, unmatched, skipped):
        """prediction for the next round: what every contig of this one decided, if they all decided alike"""
        if self.p is None or self.multi:
            return
        ext, rc = set(), False
        for c, un in enumerate(unmatched):
            if un is None:
                continue
            n = int(offsets[c + 1] - offsets[c])
            ext.add(bool(self.policy.proper_for_ext(n, un)))
            rc = rc or bool(self.policy.proper_for_rc_ext(n, un))
        self._pred_ext = ext.pop() if len(ext) == 1 and not rc and not skipped else None

    def _note_finalized(self, locks, after, before):
        """the bookkeeping of _finalize_many for a finalize the library has already applied"""
        for lk, a in zip(locks, after.tolist()):
            if self.lazy:
                self.ref_ext_sizes += frugal64(a - before)
                if self.loaded is not None:
                    self.loaded.append(a)
                before = a
            self.locks_stream += int(lk).to_bytes(8, "little")

    def _finalize_many(self, ptrs, lens, locks):
        """finalizeParallelProcessingOfTarget for consecutive targets in one call into the library:
        loadRef(ext), lazy-mode separator, lock release (MGMP.cpp:440-457, MBGC_Encoder.cpp:557-563)"""
        m = self.m
        before = m.loaded_ref_length()
        after = m.finalize_targets(ptrs, lens, locks, self.lazy)
        for lk, a in zip(locks, after.tolist()):
            if self.lazy:
                self.ref_ext_sizes += frugal64(a - before)
                if self.loaded is not None:
                    self.loaded.append(a)
                before = a
            self.locks_stream += int(lk).to_bytes(8, "little")

    def _collect_streams(self, packs, targets, T, offsets):
        """per-target stream merge in target order on rank 0 (MBGC_Encoder.cpp:542-556)."""
        if not self.multi and not self.keep_streams:
            self.stream_bytes += sum(int(pk["sizes"][pk["ks"]].sum()) for pk in packs if pk["ks"])
            return
        # order this rank's emitted contigs by contig index, pack their six streams into one tensor
        items = []
        for pk in packs:
            for k, c in zip(pk["ks"], pk["cs"]):
                items.append((c, pk, k))
        items.sort(key=lambda x: x[0])
        chunks, meta = [], []
        for c, pk, k in items:
            s0, s1 = int(pk["starts"][k * 6]), int(pk["starts"][k * 6 + 6])
            chunks.append(pk["buf"][s0:s1])
            meta.extend([c, targets[c]] + [int(x) for x in pk["sizes"][k]])      # contig, ITS rank's local target, six sizes
        if len(packs) == 1 and len(items) == len(packs[0]["starts"]) // 6 and all(k == j for j, (_, _, k) in enumerate(items)):
            local = packs[0]["buf"][: int(packs[0]["starts"][-1])]     # one emission, nothing dropped: already packed in order
        else:
            local = torch.cat(chunks) if chunks else torch.empty(0, dtype=torch.uint8, device=self.device)
        if self.multi:
            # one exchange of the per-contig sizes, which also carries the byte count the big gather needs; the gather
            # itself is asynchronous unless this rank wants the bytes now (it is waited for at the next collection)
            import torch.distributed as dist
            t0 = time.perf_counter() if self.trace is not None else 0
            all_meta = self._allgather_ints(meta + [int(local.numel()), -1])
            if self.trace is not None:
                self.trace["flush.meta"] = self.trace.get("flush.meta", 0) + time.perf_counter() - t0
            sizes = [m_[-2] for m_ in all_meta]
            all_meta = [m_[:-2] + [-1] for m_ in all_meta]
            self._wait_gathers()
            mx = max(max(sizes), 1)
            pad = torch.zeros(mx, dtype=torch.uint8, device=self.device)
            pad[: local.numel()] = local
            # a gather towards the rank that feeds the host backend (MBGC_Encoder.cpp:542-564): nobody else needs the bytes
            outs = [torch.empty(mx, dtype=torch.uint8, device=self.device) for _ in range(self.world)] if self.rank == 0 else None
            want_now = self.rank == 0 and self.keep_streams
            work = dist.gather(pad, outs, dst=0, group=self.group, async_op=not want_now)
            if not want_now:
                self._gathers.append((work, outs, pad))
            all_bytes = [outs[r][: sizes[r]] for r in range(self.world)] if self.rank == 0 else []
            self.stream_bytes += sum(sizes)
            if self.rank != 0 or not self.keep_streams:
                return
        else:
            all_bytes, all_meta = [local], [meta + [-1]]
            self.stream_bytes += int(local.numel())
            if not self.keep_streams:
                return
        for r in range(self.world):
            host = all_bytes[r].cpu().numpy().tobytes()
            mt = all_meta[r][:-1]
            pos = 0
            per_target = {}
            for i in range(0, len(mt), 8):
                lt_of_c, sizes = mt[i + 1], mt[i + 2: i + 8]           # (contig indices and layouts are local to rank r)
                st = {}
                for name, sz in zip(STREAMS, sizes):
                    st[name] = host[pos: pos + sz]
                    pos += sz
                per_target.setdefault(lt_of_c, []).append(st)
            for lt in range(T):
                for st in per_target.get(lt, []):
                    for name in STREAMS:
                        self.streams[name] += st[name]
                    self.streams["literals"].append(SEQ_SEPARATOR)
                self.streams["flags"].append(FILE_SEPARATOR)


def round_schedule(n_targets, per_rank, world):
    """Target indices of every round: round r holds targets [r*per_rank*world, ...); rank g owns the g-th
    block of per_rank consecutive targets of the round (file-per-GPU sharding)."""
    per_round = per_rank * world
    rounds = []
    for r0 in range(0, n_targets, per_round):
        rounds.append([list(range(r0 + g * per_rank, min(r0 + (g + 1) * per_rank, n_targets))) for g in range(world)])
    return rounds
