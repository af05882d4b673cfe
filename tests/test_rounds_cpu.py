"""The round protocol (mbgc_amd.rounds.RoundRunner) on CPU: single process and world_size-2 gloo, with
the oracle behind the device surface. Checks the host logic the GPUs run under: lock positions,
extension policy (incl. reverse-complement extensions), the dissimilar-contig retry, the exchange of
extensions, replica consistency and the target-order merge of the streams on rank 0."""
import os
import socket

import numpy as np
import pytest
import torch

import _driver
import _orc
import _orc_backend
from mbgc_amd import synth
from mbgc_amd.rounds import RoundRunner, round_schedule

LIM = 3_000_000


def collection(n, length, div, seed):
    base = synth.base_codes(length, seed)
    if div == "mixed":      # every third genome far from the rest: rounds of 6 hold a stopped target with kept ones behind it
        return [synth.genome(base, i, 0.07 if i % 3 == 2 else 0.004) for i in range(n)]
    return [synth.genome(base, i, div) for i in range(n)]


def reference_result(gs, round_size, contigs_per_target=1, lim=LIM):
    o = _orc.OracleMatcher(lim)
    targets = [split(g, _cpt(contigs_per_target, t)) for t, g in enumerate(gs[1:])]
    res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], targets, round_size)
    return res, o.ht(), o.loaded_ref_length()


def _cpt(contigs_per_target, t):
    """contigs of target t: a number, or "ragged" = 1, 2, 3, 1, 3, 2 ... by target (the ranks of a round then hold
    different contig layouts — contig indices are local to a rank)"""
    return (1, 2, 3, 1, 3, 2, 2)[t % 7] if contigs_per_target == "ragged" else contigs_per_target


def split(g, k):
    cuts = [0] + [g.size * i // k + (7 * i) % 13 for i in range(1, k)] + [g.size]
    return [g[cuts[i]:cuts[i + 1]] for i in range(k)]


def run_rank(rank, world, gs, per_rank, contigs_per_target, group=None, announce=0, lim=LIM):
    """announce: 0 = rounds are passed one by one; 1 = every round names the next one's buffer (next_batch), so the ranks
    tell each other its size ahead; 2 = rank 1 then passes a different buffer than it named (the others must cope)"""
    m = _orc_backend.OracleDeviceMatcher(lim)
    m.set_sliding_window_size(16)
    m.load_ref(gs[0], load_rc=True)
    runner = RoundRunner(m, rank, world, group, "cpu", lazy=True, emit_params=_orc.emit_params(1))
    runner.start()
    rounds = []
    for rnd in round_schedule(len(gs) - 1, per_rank, world):
        mine = rnd[rank]
        contigs, tg = [], []
        for lt, t in enumerate(mine):
            for c in split(gs[1 + t], _cpt(contigs_per_target, t)):
                contigs.append(c)
                tg.append(lt)
        buf = torch.from_numpy(np.concatenate(contigs).copy())
        offs = np.zeros(len(contigs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([c.size for c in contigs])
        rounds.append((buf, offs, tg))
    for i, (buf, offs, tg) in enumerate(rounds):
        nxt = rounds[i + 1] if announce and i + 1 < len(rounds) else None
        if announce == 2 and rank == 1 and i > 0:
            buf = buf.clone()                           # not the buffer named in the round before
        runner.run_round(buf, offs, tg, next_batch=nxt)
    runner.flush()
    return runner, m


@pytest.mark.parametrize("div,cpt", [(0.012, 1), (0.012, 3), (0.06, 2), (0.012, "ragged")])
def test_single_process_runner_equals_reference_loop(div, cpt):
    gs = collection(9, 60_000, div, seed=17)      # 6 % divergence forces dissimilar-contig retries
    runner, m = run_rank(0, 1, gs, 4, cpt)
    res, ht, loaded = reference_result(gs, 4, cpt)
    for k, v in res["streams"].items():
        assert bytes(runner.streams[k]) == v, k
    assert bytes(runner.locks_stream) == res["locks"]
    assert bytes(runner.ref_ext_sizes) == res["refExtSize"]
    assert m.loaded_ref_length() == loaded and np.array_equal(m.ht(), ht)


@pytest.mark.parametrize("cpt", [1, "ragged"])
def test_kept_targets_behind_a_stopped_one(cpt):
    """rounds of 6 in which the fifth target gives a contig up and the sixth does not: the sixth keeps what the first pass found
    (the reference's workers do not wait for each other's dissimilar contigs), the fifth is matched again by itself"""
    gs = collection(13, 60_000, "mixed", seed=17)
    runner, m = run_rank(0, 1, gs, 6, cpt)
    res, ht, loaded = reference_result(gs, 6, cpt)
    for k, v in res["streams"].items():
        assert bytes(runner.streams[k]) == v, k
    assert bytes(runner.locks_stream) == res["locks"]
    assert bytes(runner.ref_ext_sizes) == res["refExtSize"]
    assert m.loaded_ref_length() == loaded and np.array_equal(m.ht(), ht)
    # (the drive itself: some target was matched twice, and some target behind it was not)
    o = _orc.OracleMatcher(LIM)
    calls = []
    real = o.match
    o.match = lambda contig, *a: (calls.append(contig.size), real(contig, *a))[1]
    _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [split(g, _cpt(cpt, t)) for t, g in enumerate(gs[1:])], 6)
    total = sum(len(split(g, _cpt(cpt, t))) for t, g in enumerate(gs[1:]))
    assert total < len(calls) < 2 * total - 4, (total, len(calls))


def _worker_n(rank, world, port, outdir, div, per_rank, announce, n, lim=LIM):
    _worker(rank, world, port, outdir, div, 1, announce, n, per_rank, lim)


def _worker(rank, world, port, outdir, div, cpt, announce=0, n=9, per_rank=2, lim=LIM):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    gs = collection(n, 60_000, div, seed=17)
    runner, m = run_rank(rank, world, gs, per_rank, cpt, announce=announce, lim=lim)
    np.save(os.path.join(outdir, "ht%d.npy" % rank), m.ht())
    open(os.path.join(outdir, "pregathers%d" % rank), "w").write("%d %d" % tuple(runner.pregathers))
    open(os.path.join(outdir, "spec%d" % rank), "w").write("%d %d" % tuple(runner.spec_rounds))
    open(os.path.join(outdir, "head%d" % rank), "w").write("%d %d %d" % tuple(runner.head_gathers))
    if rank == 0:
        for k, v in runner.streams.items():
            open(os.path.join(outdir, k), "wb").write(bytes(v))
        open(os.path.join(outdir, "locks"), "wb").write(bytes(runner.locks_stream))
        open(os.path.join(outdir, "refext"), "wb").write(bytes(runner.ref_ext_sizes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("div,cpt", [(0.012, 1), (0.06, 2), (0.002, 1), (0.002, "ragged"), (0.06, "ragged")])
def test_world_size_2_gloo(tmp_path, div, cpt):
    """2 ranks x 2 targets per round == one process with rounds of 4; replicas bit-identical"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), div, cpt), nprocs=2, join=True)
    gs = collection(9, 60_000, div, seed=17)
    res, ht, _ = reference_result(gs, 4, cpt)
    for k, v in res["streams"].items():
        assert (tmp_path / k).read_bytes() == v, k
    assert (tmp_path / "locks").read_bytes() == res["locks"]
    assert (tmp_path / "refext").read_bytes() == res["refExtSize"]
    h0, h1 = np.load(tmp_path / "ht0.npy"), np.load(tmp_path / "ht1.npy")
    assert np.array_equal(h0, ht) and np.array_equal(h1, ht)
    # the extension all-gather started ahead of the round (RoundRunner._pregather): both ranks decide alike; on the
    # similar collection the second round starts one and uses it
    pg = [(tmp_path / ("pregathers%d" % r)).read_text() for r in range(2)]
    assert pg[0] == pg[1]
    if div < 0.005 and cpt == 1:                       # (at 1.2 % every round of this small collection has a retry)
        assert pg[0] == "1 1", pg


@pytest.mark.parametrize("announce", [1, 2])
def test_world_size_2_gloo_with_announced_buffers(tmp_path, announce):
    """the next round's buffer is named ahead (next_batch): its size and its targets' sizes travel during the round
    before, the extension all-gather starts without an exchange of its own and the round's finalize is queued behind
    pass 1 on every rank, gated by the reduction of the ranks' verdicts; a rank that then passes another buffer makes all
    ranks fall back — same bytes either way"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), 0.002, 1, announce, 17), nprocs=2, join=True)
    gs = collection(17, 60_000, 0.002, seed=17)
    res, ht, _ = reference_result(gs, 4, 1)
    for k, v in res["streams"].items():
        assert (tmp_path / k).read_bytes() == v, k
    assert (tmp_path / "locks").read_bytes() == res["locks"]
    assert (tmp_path / "refext").read_bytes() == res["refExtSize"]
    h0, h1 = np.load(tmp_path / "ht0.npy"), np.load(tmp_path / "ht1.npy")
    assert np.array_equal(h0, ht) and np.array_equal(h1, ht)
    pg = [(tmp_path / ("pregathers%d" % r)).read_text() for r in range(2)]
    assert pg[0] == pg[1]
    started, used = (int(x) for x in pg[0].split())
    sp = [(tmp_path / ("spec%d" % r)).read_text() for r in range(2)]
    assert sp[0] == sp[1]
    if announce == 1:
        assert (started, used) == (3, 3)               # rounds 2..4 of 4
        # ... each of them with the whole round's finalize queued behind pass 1 and applied on both ranks' say-so
        assert sp[0] == "3 3", sp
    else:
        assert sp[0] == "2 0", sp                      # tried with rank 1's veto: applied on neither rank
        # round 2: started, found poisoned, everybody falls back — and nobody predicts for round 3; round 4: the same again
        assert (started, used) == (2, 0)


@pytest.mark.parametrize("world,per_rank,n,div,announce", [(3, 2, 13, 0.002, 1), (4, 1, 13, 0.06, 0), (3, 1, 10, 0.012, 1), (2, 3, 13, "mixed", 0),
                                                          (3, 2, 13, "mixed", 1)])
def test_more_ranks_gloo(tmp_path, world, per_rank, n, div, announce):
    """3 and 4 ranks (rank-major target order inside a round, the veto-free speculative path with three parties, retries
    that cut a round between two ranks) == one process with rounds of world x per_rank"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_n, args=(world, port, str(tmp_path), div, per_rank, announce, n), nprocs=world, join=True)
    gs = collection(n, 60_000, div, seed=17)
    res, ht, _ = reference_result(gs, world * per_rank, 1)
    for k, v in res["streams"].items():
        assert (tmp_path / k).read_bytes() == v, k
    assert (tmp_path / "locks").read_bytes() == res["locks"]
    assert (tmp_path / "refext").read_bytes() == res["refExtSize"]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("ht%d.npy" % r)), ht), r
    sp = [(tmp_path / ("spec%d" % r)).read_text() for r in range(world)]
    assert len(set(sp)) == 1
    if div == 0.002:
        assert int(sp[0].split()[1]) >= 1, sp


@pytest.mark.parametrize("world,per_rank", [(2, 2), (3, 2)])
def test_after_the_wrap_only_the_loadable_head_of_a_round_travels(tmp_path, world, per_rank):
    """a small buffer that wraps early: from then on a round's locks stand one window ahead of the loading position and
    loadRef clips there (SlidingWindowSparseEMMatcher.cpp:361-378, :412-417), so the extension exchange asks every rank only
    for the bytes that lie inside that head of the round (broadcasts instead of the all-gather) — same streams, same tables"""
    import torch.multiprocessing as mp
    n, lim = 1 + 6 * world * per_rank, 900_000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_n, args=(world, port, str(tmp_path), 0.002, per_rank, 1, n, lim), nprocs=world, join=True)
    gs = collection(n, 60_000, 0.002, seed=17)
    res, ht, _ = reference_result(gs, world * per_rank, 1, lim)
    for k, v in res["streams"].items():
        assert (tmp_path / k).read_bytes() == v, k
    assert (tmp_path / "locks").read_bytes() == res["locks"]
    assert (tmp_path / "refext").read_bytes() == res["refExtSize"]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("ht%d.npy" % r)), ht), r
    hd = [(tmp_path / ("head%d" % r)).read_text() for r in range(world)]
    assert len(set(hd)) == 1
    k, asked, whole = (int(x) for x in hd[0].split())
    assert k >= 1 and asked < whole, hd


@pytest.mark.parametrize("per_rank,n,div,announce,lim", [(1, 25, 0.002, 1, LIM), (1, 25, 0.06, 0, LIM), (1, 33, 0.002, 1, 700_000)])
def test_eight_ranks_gloo(tmp_path, per_rank, n, div, announce, lim):
    """BASELINE configs[3]'s shape — eight ranks, file-per-rank, rounds of eight targets — as far as a CPU goes: == one
    process with rounds of 8 in every stream, lock positions, refExtSize and on every replica's table; with the finalize queued
    on all eight verdicts (similar collection), with dissimilar-contig retries that cut a round between ranks (6 %), and on a
    buffer that wraps several times, where only the loadable head of a round travels"""
    import torch.multiprocessing as mp
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_n, args=(world, port, str(tmp_path), div, per_rank, announce, n, lim), nprocs=world, join=True)
    gs = collection(n, 60_000, div, seed=17)
    res, ht, _ = reference_result(gs, world * per_rank, 1, lim)
    for k, v in res["streams"].items():
        assert (tmp_path / k).read_bytes() == v, k
    assert (tmp_path / "locks").read_bytes() == res["locks"]
    assert (tmp_path / "refext").read_bytes() == res["refExtSize"]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("ht%d.npy" % r)), ht), r
    sp = [(tmp_path / ("spec%d" % r)).read_text() for r in range(world)]
    assert len(set(sp)) == 1
    if div == 0.002 and lim == LIM:
        assert int(sp[0].split()[1]) >= 1, sp                        # rounds finalized on the eight ranks' device-side verdicts
    if lim < LIM:
        hd = [(tmp_path / ("head%d" % r)).read_text() for r in range(world)]
        assert len(set(hd)) == 1 and int(hd[0].split()[0]) >= 1 and int(hd[0].split()[1]) < int(hd[0].split()[2]), hd
