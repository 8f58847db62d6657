"""Batched counterpart of the reference's PPO learner for the vectorised env.

The reference (/root/reference/PPO.py) drives ONE env: ``select_action`` stores the frozen encoder's 256-d pooled
feature, the action and its log-probability per step (PPO.py:152-164); ``update`` turns the reward list into
Monte-Carlo returns that restart at terminals (PPO.py:178-188), normalises them, and runs K epochs of the clipped
surrogate over the ACTION and VALUE heads only (optimizer over ``action_head`` / ``value_head``, PPO.py:113-116;
``FullNetwork.act`` detaches the features, model.py:168-171).

Here the buffer is the (T, N, 261) rollout-record tensor of ``rollout.pack_records`` (features | action | logprob |
reward | done), every env is its own trajectory, and the update is one batched pass over T*N samples.  With several
GPUs every rank all-gathers the records (``rollout.all_gather_records``) and runs the SAME update on the same
data with the same seed: the learner is replicated, no gradient collective is needed (the heads are 771 floats).
The encoder itself is out of scope: ``rollout.pooled_features`` stands in for it.

On the GPU the K epochs of an update run in the library (``occ_ppo_update``, csrc/occ_ppo.hpp): one launch per epoch
does forward, loss, backward and the Adam step over the 771 parameters (80 epochs over 12 800 samples: 1.3 ms instead
of 35 ms of framework launches).  ``fused=False`` keeps the torch implementation (the reference for the tests, and the
CPU path of the gloo rehearsals).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _native as nat
from . import rollout

LOG_2PI = math.log(2.0 * math.pi)


def mc_returns(rewards: torch.Tensor, dones: torch.Tensor, gamma: float) -> torch.Tensor:
    """Monte-Carlo returns per env, restarting at terminals (PPO.py:178-185, one reversed scan per env).
    rewards, dones: (T, N) -> (T, N)."""
    T = rewards.shape[0]
    out = torch.empty_like(rewards)
    running = torch.zeros_like(rewards[0])
    not_done = 1.0 - dones.to(rewards.dtype)
    for t in range(T - 1, -1, -1):
        running = rewards[t] + gamma * running * not_done[t]
        out[t] = running
    return out


class ActorCriticHeads(nn.Module):
    """``FullNetwork.action_head`` / ``value_head`` (model.py:153-154,168-171) on 256-d features, with the fixed
    diagonal Gaussian of ``ActorCritic`` (PPO.py:43-45,62-80,82-104)."""

    def __init__(self, feat_dim: int = 256, action_dim: int = 2, action_std_init: float = 0.6):
        super().__init__()
        self.action_dim = action_dim
        self.action_head = nn.Linear(feat_dim, action_dim)
        self.value_head = nn.Linear(feat_dim, 1)
        self.register_buffer("action_var", torch.full((action_dim,), action_std_init * action_std_init))

    def set_action_std(self, new_action_std: float) -> None:
        self.action_var.fill_(new_action_std * new_action_std)

    def _logprob_entropy(self, mean, action):
        # MultivariateNormal(mean, diag(var)): log N and entropy in closed form
        var = self.action_var
        logdet = torch.log(var).sum()
        lp = -0.5 * (((action - mean) ** 2) / var).sum(-1) - 0.5 * (self.action_dim * LOG_2PI + logdet)
        ent = 0.5 * (self.action_dim * (1.0 + LOG_2PI) + logdet)
        return lp, ent.expand(mean.shape[:-1])

    @torch.no_grad()
    def act(self, features: torch.Tensor, generator: Optional[torch.Generator] = None):
        mean = self.action_head(features)
        eps = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        action = mean + eps * self.action_var.sqrt()
        lp, _ = self._logprob_entropy(mean, action)
        return action, lp

    def evaluate(self, features: torch.Tensor, action: torch.Tensor):
        feats = features.detach()
        mean = self.action_head(feats)
        value = self.value_head(feats).squeeze(-1)
        lp, ent = self._logprob_entropy(mean, action)
        return lp, value, ent


class BatchedPPO:
    """PPO (PPO.py:107-223) over batched rollouts.  Hyper-parameters default to trainRL.py:42-56.
    ``fused`` (default on the GPU): the epochs of ``update`` run in the library (``occ_ppo_update``); the Adam moments and
    step count then live in ``_fused_state`` (``optimizer`` only supplies the learning rates, betas and eps)."""

    def __init__(self, lr_actor: float = 3e-4, lr_critic: float = 1e-3, gamma: float = 0.99, K_epochs: int = 80,
                 eps_clip: float = 0.2, action_std_init: float = 0.6, device=None, seed: Optional[int] = None,
                 graph_epochs: bool = True, fused: Optional[bool] = None, encoder=None):
        """``encoder``: any callable obs (N,4,S,S) -> features (N,256), run under no_grad - the socket for the frozen
        encoder whose pooled features the reference stores (PPO.py:47,155-157: ``FullNetwork.forward``'s first output,
        model.py:157-166).  Default: ``rollout.pooled_features`` (8x8 average pooling: the network is out of scope)."""
        self.encoder = encoder
        self.gamma, self.eps_clip, self.K_epochs = gamma, eps_clip, K_epochs
        self.action_std = action_std_init
        if seed is not None:
            torch.manual_seed(seed)
        self.policy = ActorCriticHeads(action_std_init=action_std_init).to(device)
        self.policy_old = ActorCriticHeads(action_std_init=action_std_init).to(device)
        self.policy_old.load_state_dict(self.policy.state_dict())
        on_gpu = next(self.policy.parameters()).is_cuda
        # capturable: the optimizer keeps its step count on the device, so that an epoch can be captured in a HIP graph
        self.optimizer = torch.optim.Adam([
            {"params": self.policy.action_head.parameters(), "lr": lr_actor},
            {"params": self.policy.value_head.parameters(), "lr": lr_critic},
        ], capturable=on_gpu)
        # The K epochs of an update are one fixed launch sequence (~35 small kernels) over fixed-size tensors: on the GPU
        # epochs 4..K are replays of ONE captured HIP graph (the first three run eagerly, as the capture's warm-up).
        self.graph_epochs = bool(graph_epochs) and on_gpu
        self._graph = None  # (key, graph, static inputs, static loss outputs)
        # On the GPU the epochs run in the HIP library by default (no silent fallback: a missing library raises here).
        self.fused = on_gpu if fused is None else bool(fused)
        self._fused_state = None
        if self.fused:
            if not on_gpu:
                raise nat.NativeError("the fused PPO update needs CUDA/ROCm tensors; there is no CPU fallback (fused=False: torch)")
            nat.load()
            dev = next(self.policy.parameters()).device
            f32 = dict(dtype=torch.float32, device=dev)
            self._fused_state = dict(m=torch.zeros(nat.PPO_PARAMS, **f32), v=torch.zeros(nat.PPO_PARAMS, **f32),
                                     step=torch.zeros(1, **f32), scratch=torch.empty(nat.ppo_scratch_floats(), **f32),
                                     counter=torch.zeros(1, dtype=torch.int32, device=dev))
        self.records = []  # list of (N_total, 261) tensors, one per step

    # ---- acting (PPO.py:152-164) ------------------------------------------------------------
    def select_action(self, obs: torch.Tensor, generator: Optional[torch.Generator] = None):
        """obs (N,4,S,S) -> (features, action, logprob) from the OLD policy."""
        if self.encoder is None:
            feats = rollout.pooled_features(obs)
        else:
            with torch.no_grad():  # PPO.py:154: the encoder is frozen
                feats = self.encoder(obs)
            if feats.shape != (obs.shape[0], 256):
                raise ValueError(f"the encoder must map (N,4,S,S) to (N,256) features, got {tuple(feats.shape)}")
            feats = feats.detach().to(torch.float32).contiguous()
        action, logprob = self.policy_old.act(feats, generator)
        return feats, action, logprob

    def store(self, record: torch.Tensor) -> None:
        self.records.append(record)

    def set_action_std(self, new_action_std: float) -> None:
        self.action_std = new_action_std
        self.policy.set_action_std(new_action_std)
        self.policy_old.set_action_std(new_action_std)

    def decay_action_std(self, rate: float, min_std: float) -> None:
        self.set_action_std(max(round(self.action_std - rate, 4), min_std))  # PPO.py:136-149

    # ---- learning (PPO.py:176-217) ----------------------------------------------------------
    def _epoch_graph(self, epoch, feats, actions, old_lp, returns):
        """The captured epoch for inputs of this shape: static copies of the four inputs, the graph, its two loss outputs.
        Built after three eager epochs of the first update (optimizer state and library workspaces exist by then);
        later updates of the same size copy their data into the static inputs and replay."""
        key = (tuple(feats.shape), feats.device)
        if self._graph is None or self._graph[0] != key:
            static = [t.clone() for t in (feats, actions, old_lp, returns)]
            side = torch.cuda.Stream(device=feats.device)
            side.wait_stream(torch.cuda.current_stream(feats.device))
            g = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    out = epoch(*static)
            torch.cuda.current_stream(feats.device).wait_stream(side)
            self._graph = (key, g, static, out)
        else:
            for dst, src in zip(self._graph[2], (feats, actions, old_lp, returns)):
                dst.copy_(src)
        return self._graph

    def update(self) -> dict:
        rec = torch.stack(self.records)  # (T, N, 261)
        feats, actions = rec[..., :256], rec[..., 256:258]
        old_lp, rewards, dones = rec[..., 258], rec[..., 259], rec[..., 260]
        returns = mc_returns(rewards, dones, self.gamma)
        returns = (returns - returns.mean()) / (returns.std() + 1e-7)
        feats, actions = feats.reshape(-1, 256), actions.reshape(-1, 2)
        old_lp, returns = old_lp.reshape(-1), returns.reshape(-1)
        if self.fused:
            return self._finish_update(*self._fused_epochs(feats, actions, old_lp, returns), int(feats.shape[0]))
        losses, vlosses = [], []

        def epoch(feats, actions, old_lp, returns):
            lp, value, ent = self.policy.evaluate(feats, actions)
            ratios = torch.exp(lp - old_lp)
            adv = returns - value.detach()
            surr1 = ratios * adv
            surr2 = torch.clamp(ratios, 1 - self.eps_clip, 1 + self.eps_clip) * adv
            vloss = torch.mean((value - returns) ** 2)
            loss = (-torch.min(surr1, surr2) + 0.5 * vloss - 0.01 * ent).mean()
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            self.optimizer.step()
            return loss.detach(), vloss.detach()

        n_eager = self.K_epochs
        if self.graph_epochs and self.K_epochs > 4:
            n_eager = 3
        for _ in range(n_eager):
            l, v = epoch(feats, actions, old_lp, returns)
            losses.append(l)
            vlosses.append(v)
        if n_eager < self.K_epochs:
            try:
                g = self._epoch_graph(epoch, feats, actions, old_lp, returns)
                for _ in range(self.K_epochs - n_eager):
                    g[1].replay()
                    losses.append(g[3][0].clone())
                    vlosses.append(g[3][1].clone())
            except Exception as e:  # noqa: BLE001 - capture unsupported here: the same epochs, launched one by one
                import warnings

                warnings.warn(f"HIP-graph capture of the PPO epoch failed ({e!r}); running the epochs eagerly")
                self.graph_epochs, self._graph = False, None
                for _ in range(self.K_epochs - len(losses)):
                    l, v = epoch(feats, actions, old_lp, returns)
                    losses.append(l)
                    vlosses.append(v)
        return self._finish_update(torch.stack(losses), torch.stack(vlosses), int(feats.shape[0]))

    def _fused_epochs(self, feats, actions, old_lp, returns):
        """The K epochs in the library (csrc/occ_ppo.hpp): parameters and Adam state are updated in place."""
        pol, st = self.policy, self._fused_state
        if feats.shape[1] != nat.PPO_FEATURES or pol.action_dim != 2:
            raise nat.NativeError("occ_ppo_update is built for 256 features and 2 action components (PPO.py / model.py)")
        feats, actions = feats.contiguous().float(), actions.contiguous().float()
        old_lp, returns = old_lp.contiguous().float(), returns.contiguous().float()
        ps = nat.OccPpoState()
        for name, t in (("w_a", pol.action_head.weight), ("b_a", pol.action_head.bias), ("w_v", pol.value_head.weight),
                        ("b_v", pol.value_head.bias), ("adam_m", st["m"]), ("adam_v", st["v"]), ("adam_step", st["step"])):
            assert t.is_contiguous() and t.dtype == torch.float32
            setattr(ps, name, t.data_ptr())
        losses = torch.empty(self.K_epochs, 2, dtype=torch.float32, device=feats.device)
        g_a, g_v = self.optimizer.param_groups
        b1, b2 = g_a["betas"]
        stream = C.c_void_p(torch.cuda.current_stream(feats.device).cuda_stream)
        with torch.no_grad():
            nat.check(nat.load().occ_ppo_update(
                C.c_void_p(feats.data_ptr()), C.c_void_p(actions.data_ptr()), C.c_void_p(old_lp.data_ptr()),
                C.c_void_p(returns.data_ptr()), int(feats.shape[0]), float(self.action_std) ** 2, float(self.eps_clip),
                float(g_a["lr"]), float(g_v["lr"]), float(b1), float(b2), float(g_a["eps"]), C.byref(ps), int(self.K_epochs),
                C.c_void_p(losses.data_ptr()), C.c_void_p(st["scratch"].data_ptr()), C.c_void_p(st["counter"].data_ptr()),
                stream), "occ_ppo_update")
        self._keep = (feats, actions, old_lp, returns)  # alive until the stream has run the launches
        return losses[:, 0], losses[:, 1]

    def _finish_update(self, losses, vlosses, samples: int) -> dict:
        # one host sync for the whole update (the reference syncs nowhere inside its epoch loop either, PPO.py:196-217)
        losses, vlosses = losses.cpu(), vlosses.cpu()
        self.policy_old.load_state_dict(self.policy.state_dict())
        self.records = []
        # NB the total is not monotone: the advantages (returns - value) are re-evaluated with the improving critic
        return dict(loss_first=float(losses[0]), loss_last=float(losses[-1]), value_loss_first=float(vlosses[0]),
                    value_loss_last=float(vlosses[-1]), samples=samples)


    # ---- checkpoints (PPO.py:225-230: the old policy's weights, loaded into both) -------------
    def save(self, checkpoint_path) -> None:
        torch.save(self.policy_old.state_dict(), checkpoint_path)

    def load(self, checkpoint_path) -> None:
        dev = next(self.policy.parameters()).device
        sd = torch.load(checkpoint_path, map_location=dev)
        self.policy_old.load_state_dict(sd)
        self.policy.load_state_dict(sd)
        self.action_std = float(self.policy.action_var[0].sqrt())


def train_rollouts(venv, agent: BatchedPPO, n_updates: int = 1, T: int = 50, with_action_grad: bool = False,
                   generator: Optional[torch.Generator] = None, on_step=None, max_ep_len: Optional[int] = None,
                   stagger: bool = True) -> list:
    """trainRL.py:189-229, batched: T vectorised steps (auto-reset inside the env) -> one PPO update; repeated.
    Returns the per-update stats (mean reward, loss before/after).  ``on_step(action, rewards)`` is called after every
    step (and its backward): the hook of callers that consume ``action.grad`` (train_predict.py:52-58).

    ``max_ep_len`` (trainRL.py:22: 50): the reference ends every episode in ``env.reset()`` after max_ep_len steps,
    done or not, and records ``is_terminal = done`` - False for such a reset, so the Monte-Carlo return of PPO.py:178-185
    runs on across it.  Here the env counts every env's steps since its reset on the device and resets the expired ones
    from its reserve (``SimpleVecEnv.max_ep_len``); ``stagger`` spreads the initial ages so that the envs, which all
    start together, do not all expire in the same step (``SimpleVecEnv.stagger_ages``).

    The per-step record exchange runs on a side stream (``rollout.RecordExchange``): the record of step t is stored
    while step t + 1 renders."""
    obs = venv.reset()[:, 0]
    if max_ep_len is not None:
        venv.max_ep_len = int(max_ep_len)
        if stagger:
            venv.stagger_ages()
    import torch.distributed as dist

    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    xch = rollout.RecordExchange(obs.shape[0], obs.device, world, keep=True)
    stats = []
    for _ in range(n_updates):
        rew_sum = torch.zeros((), device=obs.device)
        pending = False
        for _t in range(T):
            feats, action, logprob = agent.select_action(obs, generator)
            action = action.detach().requires_grad_(with_action_grad)
            obs, rewards, dones, _infos = venv.step(action)
            if with_action_grad:
                rewards.sum().backward()
            if on_step is not None:
                on_step(action, rewards)
            if pending:
                agent.store(xch.wait())  # the previous step's gathered records
            xch.submit(obs, action, logprob, rewards, dones, features=feats)  # PPO.py:158: the acting state's features
            venv.obs_consumer_event = xch.ready
            pending = True
            rew_sum = rew_sum + rewards.detach().mean()
        agent.store(xch.wait())
        st = agent.update()
        st["mean_reward"] = float(rew_sum) / T
        stats.append(st)
    return stats
