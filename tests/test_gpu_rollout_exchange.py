"""RecordExchange's side-stream path (events, record_stream, the consumer event) on a CUDA device with world > 1:
two gloo ranks share the one GPU (the path had only ever run with ``side is None``)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, steps, q):
    import torch.distributed as dist

    from occlusionenv_amd import rollout

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    xch = rollout.RecordExchange(n, dev, world)
    assert xch.side is not None  # the overlapped path, not the inline one
    g = torch.Generator(device=dev).manual_seed(50 + rank)
    ok = True
    for t in range(steps):
        obs = torch.rand(n, 4, 64, 64, device=dev, generator=g)
        act = torch.randn(n, 2, device=dev, generator=g)
        lp = torch.randn(n, device=dev, generator=g)
        rew = torch.randn(n, device=dev, generator=g) + 100.0 * rank
        dn = torch.rand(n, device=dev, generator=g) < 0.3
        expect = rollout.pack_records(obs, act, lp, rew, dn).clone()  # synchronous pack of the same tensors
        xch.submit(obs, act, lp, rew, dn)
        # what SimpleVecEnv's in-place reset fallback does: wait for the consumer event, then overwrite obs
        torch.cuda.current_stream(dev).wait_event(xch.ready)
        obs.fill_(-7.0)
        got = xch.wait()
        mine = got[rank * n:(rank + 1) * n]
        ok = ok and torch.equal(mine, expect) and got.shape == (world * n, rollout.RECORD_FLOATS)
        other = got[(1 - rank) * n:(2 - rank) * n]
        ok = ok and bool(((other[:, 259] > 50.0) == (rank == 0)).all())  # the other rank's rewards carry its offset
    torch.cuda.synchronize()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_record_exchange_side_stream_world2_gloo_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 96, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
