"""Harness counterparts of the reference's callers of the hot path -- NOT a port of
PPO.py / trainRL.py (out of scope), only the loops that drive ``step()`` and consume its outputs:

  * ``gradient_ascent``  -- demo.py:80-114 / iterator.py:112-138: ``action = nn.Parameter(zeros(2))``,
    ``reward.backward()``, skip NaN gradients, ``action += lr * action.grad``.
  * ``collect_rollout``  -- trainRL.py:189-229 + PPO.py:152-164 in batched form: T steps of a policy over N envs,
    one rollout record per env-step (256 pooled features, action, logprob, reward, done = 1044 B) and one
    all-gather of the records per step when torch.distributed is initialised (rollout.py).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import rollout


def gradient_ascent(env, steps: int = 20, lr: float = 0.01, reset_kwargs: Optional[dict] = None):
    """Gradient ascent on the viewpoint action through env.step (single OcclusionEnv).  Returns the list of
    (reward, full_reward, done) per step and the final action."""
    env.reset(**(reset_kwargs or {}))
    action = torch.nn.Parameter(torch.zeros(2, device=env.device))
    log = []
    for _ in range(steps):
        if action.grad is not None:
            action.grad = None
        obs, reward, done, info = env.step(action)
        reward.backward()
        if torch.isnan(action.grad).any():  # demo.py:87-89
            continue
        with torch.no_grad():
            action += lr * action.grad
        log.append((float(reward), float(info["full_reward"]), bool(done)))
        if done:
            break
    return log, action.detach()


def gaussian_policy(std: float = 0.6) -> Callable:
    """Stand-in for ActorCritic.act (PPO.py:62-80): diagonal Gaussian around a linear read-out of the pooled
    features; returns (action, logprob).  The real actor is the frozen FullNetwork encoder + a linear head."""
    w = None

    def act(features: torch.Tensor):
        nonlocal w
        if w is None:
            g = torch.Generator(device="cpu").manual_seed(0)
            w = (torch.randn(256, 2, generator=g) * 0.05).to(features.device)
        mean = features @ w
        eps = torch.randn_like(mean)
        action = mean + std * eps
        logprob = (-0.5 * eps.pow(2) - torch.log(torch.tensor(std, device=mean.device)) - 0.9189385).sum(1)
        return action, logprob

    return act


def collect_rollout(venv, T: int = 50, policy: Optional[Callable] = None, with_grad: bool = True):
    """T batched steps.  Returns dict(records (T, world*N, 261), action_grads (T, N, 2) or None, obs)."""
    policy = policy or gaussian_policy()
    obs = venv.reset()[:, 0]
    recs, grads = [], []
    for _ in range(T):
        feats = rollout.pooled_features(obs)
        action, logprob = policy(feats)
        action = action.detach().requires_grad_(with_grad)
        obs, rewards, dones, infos = venv.step(action)
        if with_grad:
            rewards.sum().backward()  # train_predict.py:52
            grads.append(action.grad.detach().clone())
        rec = rollout.pack_records(obs, action, logprob, rewards, dones)
        recs.append(rollout.all_gather_records(rec))
    return dict(records=torch.stack(recs), action_grads=torch.stack(grads) if grads else None, obs=obs)
