"""Rollout records and their exchange between the per-GPU env shards.

Environments are independent (no cross-env term anywhere in /root/reference/environment.py:352-396), so the
N-GPU design shards contiguous env ranges over ranks with no data-path collective; the only exchange is ONE
all-gather per step of the small per-env rollout record the PPO learner consumes
(/root/reference/PPO.py:157-162, trainRL.py:203-204):

    256 f32 pooled features | 2 action | 1 logprob | 1 reward | 1 done   = 261 f32 = 1044 B / env-step

``torch.distributed`` backend "nccl" is RCCL on ROCm; on the xGMI full mesh a 1 MB-per-rank all-gather is a
latency-bound direct peer write.  Raw observations (262 KB/env) are never gathered.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

RECORD_FLOATS = 261


def env_shard(n_total: int, rank: int, world: int):
    """Contiguous env range of ``rank``: [rank*n/world, (rank+1)*n/world)."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def pooled_features(obs: torch.Tensor) -> torch.Tensor:
    """Stand-in for the frozen encoder's 256-d pooled feature (PPO.py:155-157 takes FullNetwork features;
    the network itself is out of scope): 4 channels x 8x8 adaptive average pool."""
    if obs.is_cuda and obs.dtype == torch.float32 and obs.dim() == 4 and obs.shape[1] == 4 and obs.shape[2] == obs.shape[3] \
            and obs.shape[2] % 8 == 0:
        # one pass at memory speed in the library (csrc/occ_ppo.hpp: occ_pool8_kernel); no fallback if it is missing
        import ctypes as C

        from . import _native as nat

        o = obs.detach().contiguous()
        feats = torch.empty(o.shape[0], 256, dtype=torch.float32, device=o.device)
        nat.check(nat.load().occ_pool8(C.c_void_p(o.data_ptr()), int(o.shape[0]), int(o.shape[2]), C.c_void_p(feats.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(o.device).cuda_stream)), "occ_pool8")
        return feats
    return F.adaptive_avg_pool2d(obs, 8).reshape(obs.shape[0], 256)


def pack_records(obs, actions, logprob, rewards, dones, features=None) -> torch.Tensor:
    """(n,4,S,S), (n,2), (n,), (n,), (n,) -> (n, 261) f32.  ``features`` (n,256): the features of the state that
    PRODUCED the action, when the caller already has them (PPO.py:157-158 stores those; the pooling pass over the
    observation - it reads every pixel - is then skipped)."""
    n = actions.shape[0]
    rec = torch.empty(n, RECORD_FLOATS, dtype=torch.float32, device=actions.device)
    rec[:, :256] = pooled_features(obs) if features is None else features
    rec[:, 256:258] = actions.detach()
    rec[:, 258] = logprob.detach()
    rec[:, 259] = rewards.detach()
    rec[:, 260] = dones.to(torch.float32)
    return rec


def all_gather_records(rec: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """One all-gather of the local (n,261) records -> (world*n, 261), rank-major (= global env order)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * rec.shape[0], rec.shape[1], dtype=rec.dtype, device=rec.device)
    if dist.get_backend() == "gloo" and rec.is_cuda:
        # CPU rehearsal of the N>1 path on a box without N GPUs: stage through host memory
        host = torch.empty(out.shape, dtype=rec.dtype)
        dist.all_gather_into_tensor(host, rec.detach().cpu().contiguous())
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


class RecordExchange:
    """The per-step record exchange of the N-GPU rollout, off the step's critical path.

    ``submit`` packs this rank's 261-float records and all-gathers them on a SECOND HIP stream, behind an event
    recorded on the producing stream after the step's launches: the bandwidth-bound 8x8 pooling and the latency-bound
    collective then overlap the next step's VALU-bound raster kernel instead of extending the step.  ``ready`` is an
    event recorded on the side stream once the step's tensors have been consumed: whoever is about to overwrite
    them in place (SimpleVecEnv's synchronous reset fallback writes into ``obs``) waits on it first.  On CPU tensors
    (gloo rehearsal, tests) the same sequence runs inline."""

    def __init__(self, n_local: int, device, world: int, keep: bool = False):
        """``keep``: every submit gathers into a tensor of its own (a learner stores the records of T steps:
        ``ppo.train_rollouts``); otherwise one buffer is reused and ``wait()`` hands out the last step's view of it."""
        self.device = torch.device(device)
        self.world = int(world)
        self.n_local = int(n_local)
        self.keep = bool(keep)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if (self.cuda and self.world > 1) else None
        self.gathered = torch.empty(self.world * n_local, RECORD_FLOATS, device=self.device) if (self.world > 1 and not keep) else None
        self.ready = None
        self.last = None

    def submit(self, obs, actions, logprob, rewards, dones, features=None) -> None:
        """``features`` (n,256): the acting state's features when the caller has them (``pack_records``)."""
        rew, act = rewards.detach(), actions.detach()
        dst = self.gathered
        if self.keep and self.world > 1:  # allocated on the producing stream, filled on the side stream, read after wait()
            dst = torch.empty(self.world * self.n_local, RECORD_FLOATS, device=self.device)
        if self.side is None:
            rec = pack_records(obs, act, logprob, rew, dones, features=features)
            self.last = all_gather_records(rec, dst) if self.world > 1 else rec
            return
        produced = torch.cuda.Event()
        produced.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(produced)
            rec = pack_records(obs, act, logprob, rew, dones, features=features)
            self.ready = torch.cuda.Event()
            self.ready.record(self.side)  # obs / actions / rewards / dones have been read
            self.last = all_gather_records(rec, dst)
        for t in (obs, act, logprob, rew, dones, features, dst):  # used on the side stream: keep their memory until it is done
            if t is not None:
                t.record_stream(self.side)

    def wait(self) -> torch.Tensor:
        """The gathered (world * n, 261) records of the last submit, visible to the current stream."""
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return self.last
