"""The decoder's per-contig automaton on the device (mbgc_amd/csrc/swsem_decode.hip, SURVEY.md §8(f) row 4:
MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars with extendMatchLeft/Right, MBGC_Decoder.cpp:319-523): against the
oracle's restatement of it (oracle/decode_oracle.c, pinned on streams of the reference's own encoder) and as the
device-side check of an emission (swsem_emit_verify)."""
import numpy as np
import pytest

import _orc
from mbgc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding():
    from mbgc_amd import binding
    return binding


def small_collection(n, length, div=0.01, seed=7):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


@pytest.mark.parametrize("mode,lazy,lim,div", [(1, True, 4_000_000, 0.012), (2, False, 4_000_000, 0.012), (0, True, 4_000_000, 0.012),
                                               (1, True, 600_000, 0.012), (1, False, 4_000_000, 0.05), (1, True, 4_000_000, 0.0005)])
def test_emitted_streams_decode_on_the_device(binding, mode, lazy, lim, div):
    """every emission verified on the device right after it was made (lim = 600 000: the circular buffer wraps), and the
    same streams decoded through swsem_decode_contigs_dev from buffers of their own: the contig, the byte count and the
    return value the oracle's decoder gives"""
    import torch
    gs = small_collection(9 if lim < 1_000_000 else 5, 100_000, div, seed=61 + mode)
    p = binding.emit_params(mode)
    p.lazyDecompressionSupport = int(lazy)
    po = _orc.emit_params(mode)
    po.lazyDecompressionSupport = int(lazy)
    h = binding.SlidingWindowSparseEMMatcher(lim, skip_margin=24 if mode >= 2 else 16)
    h.load_ref(gs[0], load_rc=True)
    loaded = [h.loaded_ref_length()]
    nm = 0
    for t, g in enumerate(gs[1:]):
        for c in (g[:40_000], g[40_000:]):
            rows = h.match(c)
            nm += len(rows)
            un, streams, _ = h.emit(p, 0, binding.NO_LOCK, 128, t, t, loaded)
            assert h.emit_verify() == (0, -1, 2 ** 64 - 1)
            # the same through the general entry point
            bufs = [torch.from_numpy(np.frombuffer(streams[k], dtype=np.uint8).copy()).to("cuda:0") if len(streams[k]) else
                    torch.empty(1, dtype=torch.uint8, device="cuda:0") for k in binding.STREAM_NAMES]
            dest = torch.zeros(c.size + 16, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            job = ([(b.data_ptr(), len(streams[k])) for b, k in zip(bufs, binding.STREAM_NAMES)], binding.NO_LOCK, dest.data_ptr(), c.size)
            dl, un2 = h.decode_contigs_dev(p, [job])
            back, un3 = _orc.decode_contig(h.ref(h.max_ref_length()), po, streams, _orc.NO_LOCK)
            assert int(dl[0]) == c.size and np.array_equal(dest.cpu().numpy()[:c.size], c) and np.array_equal(back, c)
            assert int(un2[0]) == un3 == (un & 0xFFFFFFFF)
            h.load_ref(c)
            if lazy:
                h.load_separator(0)
            loaded.append(h.loaded_ref_length())
    assert nm > (100 if div > 0.02 or div < 0.001 else 400)
    h.close()


def test_verify_notices_a_wrong_byte_and_malformed_streams(binding):
    import torch
    gs = small_collection(3, 120_000, 0.01, seed=71)
    h = binding.SlidingWindowSparseEMMatcher(4_000_000)
    h.load_ref(gs[0], load_rc=True)
    p = binding.emit_params(1)
    buf = torch.from_numpy(np.concatenate(gs[1:])).to("cuda:0")
    torch.cuda.synchronize()
    offs = np.array([0, 120_000, 240_000], dtype=np.uint64)
    h.match_batch_dev(buf.data_ptr(), offs, 32, None)
    h.emit_batch(p, None, None, None, None, None, [h.loaded_ref_length()], n=2)
    assert h.emit_verify() == (0, -1, 2 ** 64 - 1)
    buf[120_000 + 77_777] = ord("N")                                # the query changes under the emission: contig 1, byte 77 777
    torch.cuda.synchronize()
    assert h.emit_verify() == (1, 1, 77_777)
    # truncated / padded streams are refused (a stream that runs out, stream bytes left over)
    un, streams, _ = h.emit_result(0)
    c = gs[1]
    dest = torch.zeros(c.size + 16, dtype=torch.uint8, device="cuda:0")
    for name, cut in (("literals", -1), ("mapLen", -2), ("flags", -1), ("gapDelta", +1), ("mapOff", -4)):
        t = dict(streams)
        t[name] = t[name][:cut] if cut < 0 else t[name] + b"\x01"
        bufs = [torch.from_numpy(np.frombuffer(t[k], dtype=np.uint8).copy()).to("cuda:0") if len(t[k]) else
                torch.empty(1, dtype=torch.uint8, device="cuda:0") for k in binding.STREAM_NAMES]
        torch.cuda.synchronize()
        job = ([(b.data_ptr(), len(t[k])) for b, k in zip(bufs, binding.STREAM_NAMES)], binding.NO_LOCK, dest.data_ptr(), c.size)
        dl, un2 = h.decode_contigs_dev(p, [job])
        ok = int(un2[0]) >= 0 and int(dl[0]) == c.size and np.array_equal(dest.cpu().numpy()[:c.size], c)
        assert not ok, name
    h.close()


def test_rounds_at_full_genome_size_verify_on_the_device(binding):
    """three rounds of 16 x 5 Mbp with lock positions: every emission checked by the device decoder before its round is
    loaded (80 Mbases decoded by 16 waves: the measure of what one sequential chain per contig costs)"""
    import time
    import torch
    L, R = 5_000_000, 16
    base = synth.base_codes(L)
    h = binding.SlidingWindowSparseEMMatcher(1_280_000_000)
    h.set_sliding_window_size(16)
    g0 = torch.from_numpy(synth.genome(base, 0)).to("cuda:0")
    torch.cuda.synchronize()
    h.load_ref_dev(g0.data_ptr(), g0.numel(), True, True, 0)
    p = binding.emit_params(1)
    loaded = [h.loaded_ref_length()]
    done = 0
    for rnd in range(3):
        gs = synth.genomes(base, [1 + rnd * R + t for t in range(R)], fork=False)
        buf = torch.from_numpy(np.concatenate(gs)).to("cuda:0")
        offs = np.arange(R + 1, dtype=np.uint64) * L
        torch.cuda.synchronize()
        locks = [h.acquire_lock() for _ in range(R)]
        h.match_batch_dev(buf.data_ptr(), offs, 32, locks)
        h.emit_set_host_copy(False)
        h.emit_batch(p, None, locks, [128] * R, [done + t for t in range(R)], [done + t for t in range(R)], loaded, n=R)
        t0 = time.perf_counter()
        assert h.emit_verify() == (0, -1, 2 ** 64 - 1)
        print("device decode + compare of %d x %d bases: %.1f ms" % (R, L, (time.perf_counter() - t0) * 1e3))
        after = h.finalize_targets([buf.data_ptr() + c * L for c in range(R)], [L] * R, locks, lazy=True)
        loaded += [int(x) for x in after]
        done += R
        torch.cuda.synchronize()
    h.close()


def test_many_contigs_decode_side_by_side(binding):
    """the plan pass is one sequential chain per contig (a wave each): a batch of many contigs is what fills the device — 256
    contigs of 500 kbp in one emission, decoded and compared on the device (the figure printed is DESIGN's "decoder throughput")"""
    import time
    import torch
    L, R = 500_000, 256
    base = synth.base_codes(L, 17)
    h = binding.SlidingWindowSparseEMMatcher(400_000_000)
    h.set_sliding_window_size(16)
    g0 = torch.from_numpy(synth.genome(base, 0)).to("cuda:0")
    torch.cuda.synchronize()
    h.load_ref_dev(g0.data_ptr(), g0.numel(), True, True, 0)
    for t in range(1, 9):                                              # a few genomes in the reference: matches as in a collection
        g = torch.from_numpy(synth.genome(base, t)).to("cuda:0")
        torch.cuda.synchronize()
        h.load_ref_dev(g.data_ptr(), g.numel(), False, True, 0)
    gs = synth.genomes(base, [9 + t for t in range(R)], fork=False)
    buf = torch.from_numpy(np.concatenate(gs)).to("cuda:0")
    offs = np.arange(R + 1, dtype=np.uint64) * L
    torch.cuda.synchronize()
    locks = [h.acquire_lock() for _ in range(R)]
    h.match_batch_dev(buf.data_ptr(), offs, 32, locks)
    h.emit_set_host_copy(False)
    h.emit_batch(binding.emit_params(1), None, locks, [128] * R, list(range(R)), list(range(R)), [h.loaded_ref_length()], n=R)
    assert h.emit_verify() == (0, -1, 2 ** 64 - 1)                     # (first call: the scratch buffers)
    t0 = time.perf_counter()
    assert h.emit_verify() == (0, -1, 2 ** 64 - 1)
    dt = time.perf_counter() - t0
    print("device decode + compare of %d x %d bases: %.1f ms = %.2f Gbases/s" % (R, L, dt * 1e3, R * L / dt / 1e9))
    h.close()


@pytest.mark.parametrize("mode,lazy", [(1, 1), (0, 1), (1, 0)])
def test_long_gaps_long_matches_and_long_literal_runs_decode_back(binding, mode, lazy):
    """what k_decode_fill hands to a whole wave: right extensions across gaps of 193 B .. 300 kB (the plan counts their set flags
    a window at a time), matches of hundreds of kilobytes, runs of plain literals as long — the emission of such a contig
    must decode back to it, in one contig and side by side with an ordinary one"""
    rng = np.random.default_rng(5 + mode)
    base = synth.ACGT[rng.integers(0, 4, 1_500_000)]
    h = binding.SlidingWindowSparseEMMatcher(32_000_000)
    h.disable_sliding_window()
    h.load_ref(base, load_rc=False, add_sep=True, sep=0)
    loaded = [h.loading_position()]
    t = base.copy()
    for a, n in ((50_000, 193), (100_000, 2_000), (200_000, 300_000), (700_000, 64), (900_000, 40_000)):
        t[a:a + n] = synth.ACGT[rng.integers(0, 4, n)]                       # same length: the flanks stay on one diagonal (a gap)
    t = np.concatenate([t[:1_200_000], synth.ACGT[rng.integers(0, 4, 150_000)], t[1_200_000:]])      # an insertion: plain literals
    p = binding.emit_params(mode, lazyDecompressionSupport=lazy)
    m = h.match(t, 32, binding.NO_LOCK)
    un, streams, st = h.emit(p, 0, binding.NO_LOCK, 128, 0, 0, loaded)
    assert len(m) < 40 and len(streams["flags"]) > 300_000 and len(streams["literals"]) > 350_000
    assert h.emit_verify() == (0, -1, 2 ** 64 - 1)
    # the oracle's decoder on the same streams (the CPU restatement of MBGC_Decoder.cpp:319-523)
    back, un2 = _orc.decode_contig(h.ref(h.max_ref_length()), _orc.emit_params(mode, lazyDecompressionSupport=lazy), streams, _orc.NO_LOCK, t.size + 16)
    assert un2 == un and np.array_equal(back, t)


def test_more_contigs_than_one_grid_dimension_holds(binding):
    """70 000 contigs in one batch (draft assemblies: tens of targets of a thousand contigs each): the decoder's fill and
    check passes index contigs through grid.y, which holds 65 535 — they go in slices; every contig is given back, and a
    byte changed in one beyond the first slice is reported with its contig and position"""
    import torch
    n, step = 70_000, 64
    base = synth.genome(synth.base_codes(200_000, 5), 0, 0.0)
    h = binding.SlidingWindowSparseEMMatcher(8_000_000)
    h.load_ref(base, load_rc=False)
    rng = np.random.default_rng(11)
    starts = rng.integers(0, base.size - step, n)
    q = np.concatenate([base[s: s + step] for s in starts])
    q[rng.integers(0, q.size, q.size // 100)] = ord("A")           # some mismatches, so that extensions and literals occur
    buf = torch.from_numpy(q).to("cuda:0")
    torch.cuda.synchronize()
    offs = np.arange(n + 1, dtype=np.uint64) * step
    h.match_batch_dev(buf.data_ptr(), offs, 32, None)
    h.emit_set_host_copy(False)
    h.emit_batch(binding.emit_params(1), None, None, None, None, None, [h.loaded_ref_length()], n=n)
    assert h.emit_verify() == (0, -1, 2 ** 64 - 1)
    buf[69_000 * step + 5] = ord("N") if q[69_000 * step + 5] != ord("N") else ord("C")
    torch.cuda.synchronize()
    assert h.emit_verify() == (1, 69_000, 5)
    h.close()
