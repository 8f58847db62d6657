"""The learner against fixtures the REFERENCE's own PPO.py produced (tests/golden/ppo_golden.npz, generator
tests/golden/make_ppo_golden.py: /root/reference/PPO.py:62-104 ``act`` / ``evaluate``, :176-223 ``update``, run on the CPU of
the authoring container).  Two buffers of T = 200 single-env steps: ``enc`` - filled by the reference's ``select_action``
through its (random-init) FullNetwork encoder; ``dir`` - O(1) features handed to ``policy_old.act`` directly, so that the
actor moves and the clip branches are taken.  K in {1, 5, 80} epochs.

CPU: ``BatchedPPO(fused=False)`` (the torch path) with N = 1.  GPU (-m gpu): ``occ_ppo_update`` through the C ABI.
"""
import ast
import os

import numpy as np
import pytest
import torch

from occlusionenv_amd import ppo

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_golden.npz"))
HYPER = ast.literal_eval(str(GOLD["hyper"]))  # a dict literal written by the generator
HEADS = ("w_a", "b_a", "w_v", "b_v")


def gold(scen, name, device="cpu"):
    return torch.from_numpy(np.asarray(GOLD[f"{scen}_{name}"])).to(device)


def head_params(policy):
    return dict(w_a=policy.action_head.weight, b_a=policy.action_head.bias, w_v=policy.value_head.weight,
                b_v=policy.value_head.bias)


def agent_from_fixture(scen, K, device="cpu", **kw):
    a = ppo.BatchedPPO(lr_actor=HYPER["lr_actor"], lr_critic=HYPER["lr_critic"], gamma=HYPER["gamma"], K_epochs=K,
                       eps_clip=HYPER["eps_clip"], action_std_init=HYPER["action_std"], device=device, **kw)
    with torch.no_grad():
        for pol in (a.policy, a.policy_old):
            for k, p in head_params(pol).items():
                p.copy_(gold(scen, f"init_{k}", device))
    return a


def fill_from_fixture(agent, scen, device="cpu"):
    """The reference's buffer as T records of ONE env: features | action | logprob | reward | done."""
    rec = torch.cat([gold(scen, "features", device), gold(scen, "actions", device), gold(scen, "logprobs", device)[:, None],
                     gold(scen, "rewards", device)[:, None], gold(scen, "terminals", device).float()[:, None]], 1)
    for t in range(rec.shape[0]):
        agent.store(rec[t:t + 1])
    return rec


@pytest.mark.parametrize("scen", ["enc", "dir"])
def test_act_and_evaluate_equal_the_references(scen):
    """PPO.py:62-75 (log-probability of the sampled action under the old policy) and :82-104 (evaluate)."""
    a = agent_from_fixture(scen, 1, fused=False)
    lp, value, ent = a.policy.evaluate(gold(scen, "features"), gold(scen, "actions"))
    assert torch.allclose(lp, gold(scen, "eval_logprobs"), atol=2e-6, rtol=0)
    assert torch.allclose(lp, gold(scen, "logprobs"), atol=2e-6, rtol=0)      # what select_action stored
    assert torch.allclose(value, gold(scen, "eval_values"), atol=1e-6, rtol=0)
    assert torch.allclose(ent, gold(scen, "eval_entropy"), atol=1e-6, rtol=0)


@pytest.mark.parametrize("scen", ["enc", "dir"])
def test_normalised_returns_equal_the_references(scen):
    """PPO.py:178-188: reversed scan restarting at terminals, then (r - mean) / (std + 1e-7)."""
    r = ppo.mc_returns(gold(scen, "rewards")[:, None], gold(scen, "terminals")[:, None], HYPER["gamma"])
    r = (r - r.mean()) / (r.std() + 1e-7)
    for K in (1, 5, 80):
        assert torch.allclose(r[:, 0], gold(scen, f"returns_norm_K{K}"), atol=1e-6, rtol=0)


@pytest.mark.parametrize("scen", ["enc", "dir"])
@pytest.mark.parametrize("K", [1, 5, 80])
def test_torch_update_reproduces_the_references_heads(scen, K):
    """PPO.py:196-217 over K epochs: both heads to 1e-6 (measured 6e-8 at K = 80), the old policy synchronised."""
    a = agent_from_fixture(scen, K, fused=False, graph_epochs=False)
    fill_from_fixture(a, scen)
    st = a.update()
    assert st["samples"] == 200
    for k, p in head_params(a.policy).items():
        ref, ini = gold(scen, f"final_{k}_K{K}"), gold(scen, f"init_{k}")
        assert float((ref - ini).abs().max()) > 1e-4                      # the reference moved it
        assert float((p.detach() - ref).abs().max()) <= 1e-6, (k, K)
        assert torch.equal(head_params(a.policy_old)[k], p)
    vl = GOLD[f"{scen}_vloss_K{K}"]
    assert abs(st["value_loss_first"] - vl[0]) <= 1e-6 * max(1.0, abs(vl[0]))
    assert abs(st["value_loss_last"] - vl[-1]) <= 1e-6 * max(1.0, abs(vl[-1]))


@pytest.mark.gpu
@pytest.mark.parametrize("scen", ["enc", "dir"])
@pytest.mark.parametrize("K", [1, 5, 80])
def test_fused_update_reproduces_the_references_heads(scen, K):
    """``occ_ppo_update`` (csrc/occ_ppo.hpp, through the C ABI) on the reference's buffer: heads to 1e-6 at K <= 5.  At
    K = 80 the trajectory passes through the clip boundary of PPO.py:207, where a sample's last bit switches its gradient
    on or off - the reference's OWN f32 run ends up to 4.7 % of the distance moved away from its float64 run (the
    ``final64`` fixtures: the same update() on .double() heads, tests/golden/make_ppo_golden.py).  The float64 trajectory
    arbitrates: the fused kernel may be no farther from it than TWICE what the reference's f32 run is (floor 1e-6), and
    the critic - which no clip touches - stays within 1e-5 of the reference's f32 heads."""
    dev = "cuda:0"
    a = agent_from_fixture(scen, K, device=dev, fused=True)
    fill_from_fixture(a, scen, dev)
    st = a.update()
    for k, p in head_params(a.policy).items():
        ref, ini = gold(scen, f"final_{k}_K{K}", dev), gold(scen, f"init_{k}", dev)
        err, moved = float((p.detach() - ref).abs().max()), float((ref - ini).abs().max())
        if K <= 5:
            assert err <= 1e-6, (k, K, err)
            continue
        f64 = gold(scen, f"final64_{k}_K{K}", dev)
        e_fused = float((p.detach().double() - f64).abs().max())
        e_ref32 = float((ref.double() - f64).abs().max())
        assert e_fused <= 2.0 * e_ref32 + 1e-6, (k, K, e_fused, e_ref32, moved)
        if k in ("w_v", "b_v"):
            assert err <= 1e-5, (k, K, err)
    vl = GOLD[f"{scen}_vloss_K{K}"]
    assert abs(st["value_loss_first"] - vl[0]) <= 2e-6 * max(1.0, abs(vl[0]))
    assert abs(st["value_loss_last"] - vl[-1]) <= 1e-5 * max(1.0, abs(vl[-1]))
