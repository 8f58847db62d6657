"""N>1 path on CPU: env sharding + the single all-gather of rollout records, world_size 2 over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from occlusionenv_amd import rollout


def test_env_shard_partitions_contiguously():
    for n, w in [(8192, 8), (10, 3), (7, 7), (1024, 1)]:
        spans = [rollout.env_shard(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_pack_records_layout():
    obs = torch.rand(3, 4, 16, 16)
    act, lp, rw = torch.rand(3, 2), torch.rand(3), torch.rand(3)
    dn = torch.tensor([True, False, True])
    rec = rollout.pack_records(obs, act, lp, rw, dn)
    assert rec.shape == (3, rollout.RECORD_FLOATS) and rollout.RECORD_FLOATS * 4 == 1044
    assert torch.allclose(rec[:, :256], torch.nn.functional.adaptive_avg_pool2d(obs, 8).reshape(3, 256))
    assert torch.equal(rec[:, 256:258], act) and torch.equal(rec[:, 258], lp) and torch.equal(rec[:, 259], rw)
    assert rec[:, 260].tolist() == [1.0, 0.0, 1.0]
    assert rollout.all_gather_records(rec) is rec  # single process: no collective


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rollout.env_shard(n_total, rank, world)
    n = hi - lo
    g = torch.Generator().manual_seed(100 + rank)
    obs = torch.rand(n, 4, 16, 16, generator=g)
    rec = rollout.pack_records(obs, torch.full((n, 2), float(rank)), torch.zeros(n),
                               torch.arange(lo, hi, dtype=torch.float32), torch.zeros(n, dtype=torch.bool))
    out = rollout.all_gather_records(rec)
    q.put((rank, out[:, 259].tolist(), out[:, 256].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_records_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_total, world = 12, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rewards, a0 in res:
        assert rewards == [float(i) for i in range(n_total)]  # rank-major == global env order
        assert a0 == [0.0] * 6 + [1.0] * 6


class _StubVecEnv:
    """Deterministic stand-in with SimpleVecEnv.step's return contract (no GPU): obs depends on (global env id, step)."""

    def __init__(self, lo, hi, img=16):
        self.lo, self.n, self.img, self.t = lo, hi - lo, img, 0
        self.obs_consumer_event = None

    def step(self, actions):
        ids = torch.arange(self.lo, self.lo + self.n, dtype=torch.float32)
        obs = (ids[:, None, None, None] + 0.01 * self.t) * torch.ones(self.n, 4, self.img, self.img)
        rewards = (actions * actions).sum(1) + ids  # differentiable in the actions like the real reward
        dones = (ids.long() + self.t) % 3 == 0
        self.t += 1
        return obs, rewards, dones, None


def _bench_step_worker(rank, world, port, n_total, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rollout.env_shard(n_total, rank, world)
    venv = _StubVecEnv(lo, hi)
    xch = rollout.RecordExchange(hi - lo, "cpu", world)
    gen = torch.Generator().manual_seed(7 + rank)
    out = []
    for _ in range(steps):  # bench.py: one_step()
        actions = torch.randn(hi - lo, 2, generator=gen, requires_grad=True)
        obs, rewards, dones, _ = venv.step(actions)
        rewards.sum().backward()
        xch.submit(obs, actions, torch.zeros(hi - lo), rewards, dones)
        venv.obs_consumer_event = xch.ready
        rec = xch.wait().clone()
        out.append((rec[:, 0].tolist(), rec[:, 256:258].tolist(), rec[:, 259].tolist(), rec[:, 260].tolist(),
                    actions.grad.tolist(), actions.detach().tolist()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_sequence_world2_gloo():
    """bench.py's one_step() (step -> backward -> RecordExchange.submit -> gathered records) on two gloo ranks with a
    stub env: every rank ends up with ALL ranks' records in global env order, every step."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_total, world, steps = 10, 2, 3
    procs = [ctx.Process(target=_bench_step_worker, args=(r, world, port, n_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for t in range(steps):
        f0, act, rew, dn, _, _ = res[0][t]
        assert res[1][t][:4] == (f0, act, rew, dn)  # both ranks hold the same gathered table
        assert [round(x - 0.01 * t, 4) for x in f0] == [float(i) for i in range(n_total)]  # pooled obs = global env id
        assert dn == [1.0 if (i + t) % 3 == 0 else 0.0 for i in range(n_total)]
        # the action columns are each rank's own actions, in shard order; gradients stayed local (2 * action)
        mine = res[0][t][5] + res[1][t][5]
        assert all(abs(a - b) < 1e-6 for ra, rb in zip(act, mine) for a, b in zip(ra, rb))
        for r in range(world):
            assert all(abs(g - 2 * a) < 1e-5 for rg, ra in zip(res[r][t][4], res[r][t][5]) for g, a in zip(rg, ra))


class _StubResetVecEnv(_StubVecEnv):
    """... plus reset() and the time-limit knobs train_rollouts touches."""

    def __init__(self, lo, hi, img=16):
        super().__init__(lo, hi, img)
        self.max_ep_len, self.staggered = None, False

    def reset(self):
        ids = torch.arange(self.lo, self.lo + self.n, dtype=torch.float32)
        return (ids[:, None, None, None, None] * torch.ones(self.n, 1, 4, self.img, self.img))

    def stagger_ages(self, seed=None):
        self.staggered = True


def _ppo_rollout_worker(rank, world, port, n_total, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from occlusionenv_amd import ppo

    lo, hi = rollout.env_shard(n_total, rank, world)
    venv = _StubResetVecEnv(lo, hi)
    agent = ppo.BatchedPPO(K_epochs=3, seed=0, fused=False)  # replicated learner: same seed on every rank
    stored = []
    orig_store = agent.store
    agent.store = lambda rec: (stored.append(rec.clone()), orig_store(rec))[1]
    stats = ppo.train_rollouts(venv, agent, n_updates=1, T=T, with_action_grad=True, generator=torch.Generator().manual_seed(3 + rank),
                               max_ep_len=4)
    q.put((rank, [r.tolist() for r in stored], {k: v.tolist() for k, v in agent.policy.state_dict().items()}, stats[0]["samples"],
           venv.max_ep_len, venv.staggered))
    dist.barrier()
    dist.destroy_process_group()


def test_ppo_rollout_sequence_world2_gloo():
    """ppo.train_rollouts (bench.py --workload ppo_rollout runs the same sequence): per step act -> env step -> backward ->
    RecordExchange.submit (side-stream exchange on the GPU, inline here), the record of step t stored at step t + 1, the
    last one before the update.  Both ranks must store the SAME T gathered tables in step order - every rank's rows in
    global env order - and end up with identical heads (replicated learner, no gradient collective)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_total, world, T = 6, 2, 5
    procs = [ctx.Process(target=_ppo_rollout_worker, args=(r, world, port, n_total, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (rec0, heads0, n0, mel0, stag0), (rec1, heads1, n1, mel1, stag1) = res[0], res[1]
    assert len(rec0) == T and n0 == n1 == T * n_total and mel0 == mel1 == 4 and stag0 and stag1
    assert rec0 == rec1 and heads0 == heads1
    for t, table in enumerate(rec0):
        assert len(table) == n_total
        for i, row in enumerate(table):
            assert abs(row[259] - (row[256] ** 2 + row[257] ** 2 + i)) < 1e-4  # reward of global env i, its own action
            assert row[260] == (1.0 if (i + t) % 3 == 0 else 0.0)            # done flag of step t: the tables are in step order
