"""Several laps of the circular buffer (SlidingWindowSparseEMMatcher.cpp:402-437): after the first wrap the table fills
with entries of older laps, which the reference follows whatever they point at. The device path drops the ones whose slot
was sampled again by the load that wrote its present text (the lap tags) — exact, so over five laps of a
small buffer every stream, lock, extension size and the table image must equal the oracle-driven reference loop's, with
the tags and without them."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _driver
import _orc
from mbgc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIM = 2_400_000


def collection(n, length, div, seed):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


def run(tags, round_size=1):
    import torch
    from mbgc_amd import binding
    from mbgc_amd.rounds import RoundRunner, round_schedule
    gs = collection(121, 100_000, 0.01, seed=71)
    h = binding.SlidingWindowSparseEMMatcher(LIM)
    h.set_sliding_window_size(16)
    h.load_ref(gs[0], load_rc=True)
    runner = RoundRunner(h, 0, 1, None, "cuda:0", lazy=True, emit_params=binding.emit_params(1))
    runner.start()
    for rnd in round_schedule(len(gs) - 1, round_size, 1):
        mine = [gs[1 + t] for t in rnd[0]]
        buf = torch.from_numpy(np.concatenate(mine)).to("cuda:0")
        offs = np.zeros(len(mine) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([c.size for c in mine])
        torch.cuda.synchronize()
        runner.run_round(buf, offs)
    runner.flush()
    o = _orc.OracleMatcher(LIM)
    res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [[g] for g in gs[1:]], round_size)
    for k, v in res["streams"].items():
        assert bytes(runner.streams[k]) == v, (k, tags)
    assert bytes(runner.locks_stream) == res["locks"] and bytes(runner.ref_ext_sizes) == res["refExtSize"]
    assert np.array_equal(h.ht(), o.ht())
    assert o.loaded_ref_length() > (5 if round_size == 1 else 2) * LIM      # (rounds of 4: the lock window clips most of every extension)
    o.close()
    h.close()


@pytest.mark.parametrize("round_size", [1, 4])
def test_five_laps_equal_the_oracle_loop(round_size):
    run(True, round_size)


def test_five_laps_without_lap_tags_in_a_child():
    """SWSEM_LAP_TAGS=0 is read when a handle is created: a child process runs the same comparison with every stale entry
    visited"""
    env = dict(os.environ, SWSEM_LAP_TAGS="0", PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    r = subprocess.run([sys.executable, "-c", "import test_gpu_laps as t; t.run(False)"], cwd=os.path.join(ROOT, "tests"), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
