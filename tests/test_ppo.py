"""Batched PPO learner counterpart against the reference's algorithm restated inline
(/root/reference/PPO.py:62-104,176-217)."""
import math

import torch
from torch.distributions import MultivariateNormal

from occlusionenv_amd import ppo, rollout


def _returns_reference_style(rewards, terminals, gamma):
    """PPO.py:178-185 for ONE env: reversed scan, restart at terminals."""
    out, running = [], 0.0
    for r, d in zip(reversed(rewards), reversed(terminals)):
        if d:
            running = 0.0
        running = r + gamma * running
        out.insert(0, running)
    return out


def test_mc_returns_match_per_env_scan():
    g = torch.Generator().manual_seed(0)
    T, N = 17, 5
    rewards = torch.randn(T, N, generator=g)
    dones = torch.rand(T, N, generator=g) < 0.2
    got = ppo.mc_returns(rewards, dones, 0.99)
    for n in range(N):
        exp = _returns_reference_style(rewards[:, n].tolist(), dones[:, n].tolist(), 0.99)
        assert torch.allclose(got[:, n], torch.tensor(exp), atol=1e-5)


def test_logprob_and_entropy_equal_multivariate_normal():
    torch.manual_seed(1)
    heads = ppo.ActorCriticHeads(action_std_init=0.6)
    feats = torch.randn(7, 256)
    actions = torch.randn(7, 2)
    lp, value, ent = heads.evaluate(feats, actions)
    mean = heads.action_head(feats)
    dist = MultivariateNormal(mean, torch.diag_embed(heads.action_var.expand_as(mean)))  # PPO.py:87-90
    assert torch.allclose(lp, dist.log_prob(actions), atol=1e-5)
    assert torch.allclose(ent, dist.entropy(), atol=1e-5)
    assert value.shape == (7,)
    a, alp = heads.act(feats)
    assert torch.allclose(alp, dist.log_prob(a), atol=1e-5)
    heads.set_action_std(0.3)
    assert torch.allclose(heads.action_var, torch.full((2,), 0.09))


def test_update_trains_heads_and_syncs_old_policy():
    agent = ppo.BatchedPPO(K_epochs=20, seed=0)
    g = torch.Generator().manual_seed(2)
    T, N = 12, 16
    w_true = torch.randn(256, generator=g) * 0.1
    for _ in range(T):
        obs = torch.rand(N, 4, 32, 32, generator=g)
        feats, action, logprob = agent.select_action(obs)
        assert feats.shape == (N, 256) and action.shape == (N, 2) and logprob.shape == (N,)
        rewards = feats @ w_true + 0.1 * action[:, 0]
        dones = torch.rand(N, generator=g) < 0.1
        rec = rollout.pack_records(obs, action, logprob, rewards, dones)
        agent.store(rec)
    before = {k: v.clone() for k, v in agent.policy.state_dict().items()}
    st = agent.update()
    assert st["samples"] == T * N and st["loss_last"] < st["loss_first"]
    after = agent.policy.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before if "head" in k)
    for k, v in agent.policy_old.state_dict().items():
        assert torch.equal(v, after[k])
    assert agent.records == []
    agent.decay_action_std(0.05, 0.1)
    assert abs(agent.action_std - 0.55) < 1e-6 and torch.allclose(agent.policy_old.action_var, torch.full((2,), 0.55 ** 2))


def test_save_and_load_round_trip(tmp_path):
    """PPO.save / PPO.load (PPO.py:225-230): the old policy's weights, restored into both policies."""
    a = ppo.BatchedPPO(K_epochs=2, seed=0)
    a.set_action_std(0.4)
    with torch.no_grad():
        for p in a.policy_old.parameters():
            p.add_(0.25)
    path = tmp_path / "ppo.pth"
    a.save(path)
    b = ppo.BatchedPPO(K_epochs=2, seed=1)
    b.load(path)
    for k, v in a.policy_old.state_dict().items():
        assert torch.equal(b.policy.state_dict()[k], v) and torch.equal(b.policy_old.state_dict()[k], v)
    assert abs(b.action_std - 0.4) < 1e-6


def test_closed_form_epoch_gradients_equal_autograd():
    """The formulas that csrc/occ_ppo.hpp implements (its header comment), written out in numpy-style torch without
    autograd, against torch autograd of the loss of PPO.py:199-212 - on the CPU, so that the kernel's arithmetic is pinned
    independently of the GPU test that compares the kernel itself with the torch epochs.  Includes samples whose ratio is
    clipped from above and from below with either sign of the advantage (the four branches of min / clamp)."""
    g = torch.Generator().manual_seed(11)
    M, F_ = 400, 256
    feats = torch.rand(M, F_, generator=g)
    heads = ppo.ActorCriticHeads(action_std_init=0.5)
    with torch.no_grad():
        heads.action_head.weight.mul_(3.0)
    actions = torch.randn(M, 2, generator=g) * 0.7
    old_lp = -1.2 + 0.8 * torch.randn(M, generator=g)        # wide spread: many ratios outside [0.8, 1.2]
    returns = torch.randn(M, generator=g)
    eps_clip = 0.2
    # autograd
    lp, value, ent = heads.evaluate(feats, actions)
    ratios = torch.exp(lp - old_lp)
    adv = returns - value.detach()
    surr1, surr2 = ratios * adv, torch.clamp(ratios, 1 - eps_clip, 1 + eps_clip) * adv
    vloss = torch.mean((value - returns) ** 2)
    loss = (-torch.min(surr1, surr2) + 0.5 * vloss - 0.01 * ent).mean()
    loss.backward()
    # closed form
    with torch.no_grad():
        var = float(heads.action_var[0])
        mean = feats @ heads.action_head.weight.t() + heads.action_head.bias
        val = feats @ heads.value_head.weight.t().squeeze(1) + heads.value_head.bias
        e = actions - mean
        lp_c = -0.5 * (e * e).sum(1) / var - 0.5 * (2 * ppo.LOG_2PI + 2 * math.log(var))
        ratio = torch.exp(lp_c - old_lp)
        adv_c = returns - val
        lo, hi = 1 - eps_clip, 1 + eps_clip
        s1, s2 = ratio * adv_c, ratio.clamp(lo, hi) * adv_c
        through = ((ratio >= lo) & (ratio <= hi)) | (s1 < s2)
        assert int((~through).sum()) > 20 and int(((ratio > hi) & through).sum()) > 5 and int(((ratio < lo) & through).sum()) > 5
        dlp = torch.where(through, -adv_c * ratio / M, torch.zeros(()))
        gm = dlp[:, None] * e / var                           # d loss / d mean
        gval = (val - returns) / M
        loss_c = (-torch.minimum(s1, s2)).mean() + 0.5 * ((val - returns) ** 2).mean() - 0.01 * 0.5 * (2 * (1 + ppo.LOG_2PI) + 2 * math.log(var))
    assert abs(float(loss_c) - float(loss.detach())) < 1e-5
    assert torch.allclose(gm.t() @ feats, heads.action_head.weight.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(gm.sum(0), heads.action_head.bias.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose((gval[:, None] * feats).sum(0, keepdim=True), heads.value_head.weight.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(gval.sum().reshape(1), heads.value_head.bias.grad, rtol=1e-4, atol=1e-6)


def test_encoder_hook_feeds_the_buffer():
    """PPO.py:47,155-157: the buffer holds the FROZEN ENCODER's pooled features.  BatchedPPO(encoder=...) takes any
    callable obs (N,4,S,S) -> (N,256); it runs under no_grad and its output is what select_action returns and acts on."""
    calls = []
    conv = torch.nn.Sequential(torch.nn.Conv2d(4, 16, 3, stride=2, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(4),
                               torch.nn.Flatten())  # (N, 16 * 4 * 4) = (N, 256)

    def encoder(obs):
        calls.append((tuple(obs.shape), torch.is_grad_enabled()))
        return conv(obs)

    agent = ppo.BatchedPPO(K_epochs=1, seed=0, fused=False, encoder=encoder)
    obs = torch.rand(5, 4, 32, 32)
    feats, action, logprob = agent.select_action(obs)
    assert calls == [((5, 4, 32, 32), False)] and not feats.requires_grad
    assert torch.allclose(feats, conv(obs).detach()) and feats.shape == (5, 256)
    lp, _, _ = agent.policy_old.evaluate(feats, action)
    assert torch.allclose(lp, logprob, atol=1e-6)
    # default: the 8x8 pooling stand-in
    f2, _, _ = ppo.BatchedPPO(K_epochs=1, seed=0, fused=False).select_action(obs)
    assert torch.allclose(f2, rollout.pooled_features(obs))
    import pytest

    with pytest.raises(ValueError):
        ppo.BatchedPPO(K_epochs=1, seed=0, fused=False, encoder=lambda o: torch.zeros(o.shape[0], 7)).select_action(obs)


def test_returns_run_on_across_a_time_limit_reset():
    """trainRL.py:191-229 resets the env after max_ep_len steps WITHOUT a terminal (is_terminal = done = False), so the
    Monte-Carlo return of PPO.py:178-185 keeps flowing across that reset; a real terminal cuts it."""
    rewards = torch.tensor([[1.0], [1.0], [1.0], [1.0]])
    trunc = ppo.mc_returns(rewards, torch.zeros(4, 1, dtype=torch.bool), 0.5)           # time-limit reset after step 1: not a terminal
    term = ppo.mc_returns(rewards, torch.tensor([[False], [True], [False], [False]]), 0.5)
    assert trunc[:, 0].tolist() == [1.875, 1.75, 1.5, 1.0]
    assert term[:, 0].tolist() == [1.5, 1.0, 1.5, 1.0]
    assert trunc[:, 0].tolist() == _returns_reference_style([1.0] * 4, [False] * 4, 0.5)
