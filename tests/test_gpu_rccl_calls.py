"""The torch.distributed calls the N > 1 round protocol (mbgc_amd/rounds.py) makes, on the backend it makes them on:
nccl = RCCL. The test box has one GPU and RCCL refuses two ranks on a device, so this is a one-rank group — it cannot
show the exchange working between GPUs, but it does run every collective, dtype, reduction and stream arrangement of the
protocol through RCCL itself (the two-rank tests run them over gloo)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, port, out):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    dev = torch.device("cuda:0")
    # the announcement of the next round's sizes: a small all-gather on a stream of its own, taken a round later
    ctl = torch.cuda.Stream(dev)
    with torch.cuda.stream(ctl):
        t = torch.tensor([7, 2, 3, 4], dtype=torch.int64, device=dev)
        o = torch.empty(4, dtype=torch.int64, device=dev)
        work = dist.all_gather_into_tensor(o, t, async_op=True)
    with torch.cuda.stream(ctl):
        work.wait()
        assert o.tolist() == [7, 2, 3, 4]
    # the extension all-gather started ahead, waited for on the handle's stream, and the verdict's reduction behind it
    main = torch.cuda.Stream(dev)                       # stands for the handle's stream (a raw hipStream_t to the callback)
    q = torch.arange(1 << 20, dtype=torch.int64, device=dev).to(torch.uint8)
    big = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    pre = dist.all_gather_into_tensor(big, q, async_op=True)
    gate = torch.ones(1, dtype=torch.int32, device=dev)
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    with torch.cuda.stream(torch.cuda.ExternalStream(main.cuda_stream, device=dev)):
        pre.wait()
        dist.all_reduce(gate, op=dist.ReduceOp.MIN)
        host.copy_(gate, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    ev.synchronize()
    assert int(host[0]) == 1 and torch.equal(big, q)
    # sizes with all_gather on a list, and the stream gather to rank 0
    n = torch.tensor([5], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n)]
    dist.all_gather(sizes, n)
    assert int(sizes[0].item()) == 5
    pad = torch.full((1000,), 9, dtype=torch.uint8, device=dev)
    outs = [torch.empty(1000, dtype=torch.uint8, device=dev)]
    w = dist.gather(pad, outs, dst=0, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], pad)
    dist.barrier()
    dist.destroy_process_group()
    open(out, "w").write("ok")


def test_the_protocols_collectives_run_on_rccl(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(port, str(tmp_path / "ok")), nprocs=1, join=True)
    assert (tmp_path / "ok").read_text() == "ok"
