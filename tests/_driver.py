"""Backend-agnostic restatement of the reference's target loops, used by the tests to drive the
oracle, the reference harness and the HIP path through the *same* schedule.

  encode_sequential  MultipleGenomeMatchingProcessor::processTargetsWithParallelIO
                     (matching/MultipleGenomeMatchingProcessor.cpp:232-313) = `mbgc c -t1`
  encode_rounds      processTarget + finalizeParallelProcessingOfTarget (:340-468) with the
                     deterministic round schedule of SURVEY.md §8e: the targets of a round acquire
                     their lock positions at the same pos1, are matched against the frozen
                     reference, then their extensions are loaded in target order.

`matcher` needs the SlidingWindowSparseEMMatcher surface (load_ref, match, acquire_lock, ...);
`emit(matches, contig, lock, factor, processed, target_idx, loaded)` appends to per-target streams and
returns unmatchedChars (or SKIPPED)."""
import numpy as np

SKIPPED = 2 ** 64 - 1
NO_LOCK = 2 ** 64 - 1
SEQ_SEPARATOR = 0xA2   # MBGC_Params.h:46
FILE_SEPARATOR = 0xBB  # MBGC_Params.h:47


def frugal64(v):
    """PgHelpers::writeUInt64Frugal, utils/helper.cpp:237-246."""
    out = int(min(v, 0xFFFF)).to_bytes(2, "little")
    if v >= 0xFFFF:
        out += int(min(v, 0xFFFFFFFF)).to_bytes(4, "little")
        if v >= 0xFFFFFFFF:
            out += int(v).to_bytes(8, "little")
    return out


def ref_length_limit(files_count, basic_len, k1=16, mode=1):
    """loadG0Ref :130-134 + initMatcher :152-168 (RC in reference enabled, circular)."""
    clz = 32 - int(files_count).bit_length()
    tmp = min(12, max(5, 15 - clz // 3))
    factor = 1 << tmp
    basic = max(basic_len, 1 << 21)
    lim = factor * basic * 2
    bit40 = True
    if lim <= 0xFFFFFFFF:
        bit40 = False
    elif lim > (0xFFFFFFFF << 8):
        lim = 0xFFFFFFFF << 8
    if lim > 0xFFFFFFFF:
        ratio = 4 if mode >= 2 else 16
        lim = 0xFFFFFFFF + (lim - 0xFFFFFFFF) // ratio
    return lim, bit40


class Policy:
    """MGMP_Params.h:175-196 with the -m presets of MBGC_Params.h:886-922."""

    def __init__(self, mode=1):
        self.factor = 128
        self.rc_factor = 128 if mode >= 2 else 8
        self.outrun = {0: 4, 2: 0}.get(mode, 1)                        # allowedTargetsOutrunForDissimilarContigs, MGMP_Params.h:58-62
        self.lazy = True

    def proper_for_ext(self, n, unmatched): return unmatched * self.factor > n
    def proper_for_rc_ext(self, n, unmatched): return unmatched * self.rc_factor > n


def encode_sequential(matcher, emitter, files, policy=None, lazy=True):
    """files: list of lists of contigs (uint8 arrays). Returns dict(locks=..., refExtSize=..., matches=[...])."""
    pol = policy or Policy()
    matcher.disable_sliding_window()                                   # MGMP.cpp:177-178
    matcher.load_ref(files[0][0], load_rc=True, add_sep=True, sep=0)   # :91-100 (first contig only), :183
    loaded = [matcher.loading_position()]                              # MBGC_Encoder.cpp:789-791
    locks, ref_ext, all_matches = b"", b"", []
    for fi, contigs in enumerate(files):
        start_pos = matcher.loaded_ref_length()
        for contig in contigs:
            m = matcher.match(contig, 32, NO_LOCK)
            all_matches.append(m)
            unmatched = emitter.process(m, contig, NO_LOCK, pol.factor, 0, 0, loaded)
            if pol.proper_for_ext(contig.size, unmatched):
                matcher.load_ref(contig, load_rc=pol.proper_for_rc_ext(contig.size, unmatched), add_sep=True, sep=0)
            emitter.put(0, bytes([SEQ_SEPARATOR]))                     # processAfterSequence
        emitter.put(5, bytes([FILE_SEPARATOR]))                        # processAfterTarget
        if lazy:                                                       # MBGC_Encoder.cpp:498-509
            matcher.load_separator(0)
            size = matcher.loaded_ref_length() - start_pos
            ref_ext += frugal64(size)
            loaded.append(loaded[-1] + size)
        lk = matcher.acquire_lock()
        locks += int(lk).to_bytes(8, "little")
        matcher.release_lock(lk)
    return dict(locks=locks, refExtSize=ref_ext, matches=all_matches, loaded=loaded)


def encode_rounds(matcher, make_emitter, g0, targets, round_size, policy=None, lazy=True, sw_factor=16, threads=1,
                  keep_matches=True):
    """g0: list of contigs of the first file (reference only). targets: list of lists of contigs.
    make_emitter() -> fresh per-target emitter. Returns per-target streams merged in target order.

    The schedule (one admissible run of the reference's parallel mode, MGMP.cpp:340-468, made deterministic): the targets of a
    round take their lock positions together (:353-358); every target's worker then runs against the reference as the round
    found it — contig after contig, until it meets a contig that processMatches gives up as dissimilar (:382-388), where it
    stops. The finalizer then takes the targets in order (:433-468): a target whose worker got through keeps what it found; the
    stopped targets that follow each other — at most allowedTargetsOutrunForDissimilarContigs + 1 of them, a UNIT — are matched
    again, whole, when every target in front of the unit is loaded (processMatches then gives nothing up, MBGC_Encoder.cpp:203),
    and loaded before the finalizer goes on. (The reference's stopped workers wait for exactly that, :385-386; that they start
    their target again instead of going on behind the given-up contig is the run in which they had started late. Rounds 1-3 of
    this repo re-matched every LATER target of the round as well, stopped or not: on collections of unrelated species, where
    most targets hold a dissimilar contig, a pass over the rest of the round per stopped target.)
    threads > 1: the run-ahead part of a round goes to a thread pool (the reference is frozen meanwhile and the backends
    release the GIL: same results). keep_matches=False drops the match rows of finished targets (a 1000-genome run holds
    48 M of them)."""
    pol = policy or Policy()
    matcher.set_sliding_window_size(sw_factor)                         # MGMP.cpp:179-180
    g0cat = np.concatenate(g0)
    matcher.load_ref(g0cat, load_rc=True, add_sep=True, sep=0)
    loaded = [matcher.loading_position()]
    merged = {k: b"" for k in ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags")}
    all_matches, unm = [], []
    state = dict(processed=0, ref_ext=b"", locks=b"")
    lock, exts, ems = {}, {}, {}

    def finalize(t):                                                   # in target order, :433-468
        start_pos = matcher.loaded_ref_length()
        # targetRefExtensions[e] is ONE string (contig, then its RC), loaded without RC (:441-443)
        parts = [contig if kind == "fw" else revcomp(contig) for kind, contig in exts[t]]
        if parts:
            matcher.load_ref(np.concatenate(parts), load_rc=False, add_sep=True, sep=0)
        s = ems[t].streams()
        for k in merged:
            merged[k] += s[k]
        if lazy:
            matcher.load_separator(0)
            state["ref_ext"] += frugal64(matcher.loaded_ref_length() - start_pos)
            loaded.append(matcher.loaded_ref_length())                 # MBGC_Encoder.cpp:561
        state["locks"] += int(lock[t]).to_bytes(8, "little")
        matcher.release_lock(lock[t])
        state["processed"] += 1

    pool = None
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(threads)

    def worker(t, em, processed, loaded_now):                          # target t's contigs, until one is given up
        out = []
        for ci in range(len(targets[t])):
            contig = targets[t][ci]
            m = matcher.match(contig, 32, lock[t])
            unmatched = em.process(m, contig, lock[t], pol.factor, processed, t, loaded_now)
            if unmatched == SKIPPED:
                return out, ci
            out.append((m, unmatched))
            em.put(0, bytes([SEQ_SEPARATOR]))                          # processAfterSequence
        em.put(5, bytes([FILE_SEPARATOR]))                             # processAfterTarget
        return out, None

    def run_workers(ts):                                               # against the reference as it is now
        p0, l0 = state["processed"], list(loaded)
        run = (lambda t: worker(t, ems[t], p0, l0))
        return dict(zip(ts, pool.map(run, ts) if pool is not None else map(run, ts)))

    def decide(t, out):                                                # the extension policy, :389-398
        for (m, unmatched), contig in zip(out, targets[t]):
            all_matches.append(m if keep_matches else len(m))
            unm.append(unmatched)
            if pol.proper_for_ext(contig.size, unmatched):
                exts[t].append(("fw", contig))
            if pol.proper_for_rc_ext(contig.size, unmatched):
                exts[t].append(("rc", contig))

    unit = getattr(pol, "outrun", 1) + 1
    for r0 in range(0, len(targets), round_size):
        rnd = list(range(r0, min(r0 + round_size, len(targets))))
        for t in rnd:
            lock[t] = matcher.acquire_lock()                           # :353-358, same pos1 for the round
            ems[t], exts[t] = make_emitter(), []
        got = run_workers(rnd)
        stopped = {t for t in rnd if got[t][1] is not None}
        t = rnd[0]
        while t <= rnd[-1]:
            if t not in stopped:
                decide(t, got[t][0])
                finalize(t)
                t += 1
                continue
            ts = [t]
            while len(ts) < unit and ts[-1] + 1 in stopped:
                ts.append(ts[-1] + 1)
            for u in ts:
                # (what the first pass put there is void; an emitter bound to its target — the reference's — is reset)
                ems[u], exts[u] = (ems[u].reset() if hasattr(ems[u], "reset") else make_emitter()), []
            again = run_workers(ts)
            for u in ts:
                assert again[u][1] is None, "a contig was given up although every target in front of its unit had been loaded"
                decide(u, again[u][0])
            for u in ts:
                finalize(u)
            t = ts[-1] + 1
        if not keep_matches:
            for t in rnd:
                ems.pop(t, None), exts.pop(t, None)
    if pool is not None:
        pool.shutdown()
    locks_stream, ref_ext = state["locks"], state["ref_ext"]
    return dict(streams=merged, locks=locks_stream, refExtSize=ref_ext, matches=all_matches, unmatched=unm)


_UC = np.arange(256, dtype=np.uint8)
_UC[127] = 0
for _a, _b in zip(b"AaCcGgTtNnUuYyRrKkMmBbDdHhVvWwSs", b"TTGGCCAANNAARRYYMMKKVVHHDDBBSSWW"):
    _UC[_a] = _b


def revcomp(seq):
    """PgHelpers::upperReverseComplement, utils/helper.cpp:312-338,405-410 (numpy form for the drivers)."""
    return _UC[np.asarray(seq, dtype=np.uint8)][::-1].copy()


def parse_fasta(path):
    """Minimal stand-in for kseq_read_lossless_fasta (utils/kseq.h:233-274): contigs as uint8 arrays."""
    contigs, cur = [], []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    contigs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
                cur = []
            else:
                cur.append(line.rstrip(b"\r\n"))
    if cur:
        contigs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
    return contigs
