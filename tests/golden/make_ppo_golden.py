"""Generates tests/golden/ppo_golden.npz by RUNNING the reference's own learner (/root/reference/PPO.py) on the CPU of the
authoring container (PPO.py and model.py import fine; only the weight blob of PPO.py:48 is missing, and
that blob is data: a random-init ``FullNetwork(8, dilation=2, separable=True)`` saved under the file name PPO.py expects).

    python tests/golden/make_ppo_golden.py

What is recorded is what the reference computed, nothing restated here:
  * the buffer its ``select_action`` filled over T steps (PPO.py:152-164): pooled features of ITS encoder, the sampled
    actions, their log-probabilities; rewards / is_terminals drawn here (trainRL.py:203-204 appends them);
  * ``policy.evaluate(old_states, old_actions)`` (PPO.py:82-104) before the update;
  * inside ``update()`` (PPO.py:176-223), per K in {1, 5, 80}: the normalised returns and every epoch's value loss
    (captured by wrapping ``ppo.MseLoss``, which update() calls with (state_values, rewards)), and the resulting
    action_head / value_head parameters;
  * for K = 80 the same ``update()`` once more with the two heads, the buffer and the action variance converted to
    float64 (``.double()``; every line executed is still the reference's): the DOUBLE-PRECISION trajectory's heads,
    ``<scen>_final64_<head>_K80`` - the arbiter of how far an f32 implementation may drift over 80 epochs through the
    clip boundary of PPO.py:207 (tests/test_ppo_golden.py).
The reference never travels to the GPU box; only the .npz does.
"""
import os
import sys
import tempfile

import numpy as np

os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"  # PPO.py:48 loads a whole pickled module
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402

T, S = 200, 64
HYPER = dict(lr_actor=3e-4, lr_critic=1e-3, gamma=0.99, eps_clip=0.2, action_std=0.6)  # trainRL.py:42-56


class _Spy(torch.nn.Module):
    """Stands where PPO.MseLoss stands; records update()'s own (state_values, rewards) per epoch."""

    def __init__(self):
        super().__init__()
        self.inner = torch.nn.MSELoss()
        self.returns, self.vloss, self.values = None, [], []

    def forward(self, values, returns):
        out = self.inner(values, returns)
        self.returns = returns.detach().clone()
        self.values.append(values.detach().clone())
        self.vloss.append(float(out.detach()))
        return out


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.makedirs("models")
        import model as ref_model  # noqa: E402
        import PPO as ref_ppo  # noqa: E402

        torch.manual_seed(20261004)
        torch.save(ref_model.FullNetwork(8, dilation=2, separable=True), "./models/bestSegModel_final2_dice_l1_dilated_res_sep.pt")

        def make(K):
            return ref_ppo.PPO(256, 2, HYPER["lr_actor"], HYPER["lr_critic"], HYPER["gamma"], K, HYPER["eps_clip"], True,
                               HYPER["action_std"])

        g = torch.Generator().manual_seed(7)
        obs = torch.rand(T, 1, 4, S, S, generator=g)
        # rewards on the scale of the env's (environment.py:382-392: -0.2 per step, +5 on success), terminals as the env's
        # `done`; the time-limit resets of trainRL.py:191-229 are NOT terminals there, so there are long stretches without one
        rewards = (-0.2 + 0.3 * torch.randn(T, generator=g)).tolist()
        terminals = (torch.rand(T, generator=g) < 0.04).tolist()
        for t in range(T):
            if terminals[t]:
                rewards[t] += 5.0

        # second scenario: the buffer's states given directly (features of O(1), as a trained encoder's post-ReLU pooled
        # features would be: the random-init encoder's are ~0.01 and barely move the actor), everything after
        # extract_features still the reference's own code (policy_old.act, PPO.py:62-75)
        direct = torch.randn(T, 256, generator=g).abs()
        for scen, K in [(s, k) for s in ("enc", "dir") for k in (1, 5, 80)]:
            _run(out, scen, K, make, obs, direct, rewards, terminals)
        for scen in ("enc", "dir"):
            _run(out, scen, 80, make, obs, direct, rewards, terminals, f64=True)
        out["hyper"] = np.array(repr(dict(HYPER, T=T, obs_side=S)))
        os.chdir(HERE)
    np.savez_compressed(os.path.join(HERE, "ppo_golden.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})


def _heads(policy):
    net = policy.model
    return dict(w_a=net.action_head.weight, b_a=net.action_head.bias, w_v=net.value_head.weight, b_v=net.value_head.bias)


def _run(out, scen, K, make, obs, direct, rewards, terminals, f64=False):
    """One reference agent: fill its buffer, evaluate(), update(); everything lands in `out` under `<scen>_...`.
    ``f64``: the update runs on float64 copies of the heads / buffer (same f32 inputs); only the final heads are recorded."""
    agent = make(K)
    agent.policy.eval(), agent.policy_old.eval()
    torch.manual_seed(99)  # the sampling noise of the old policy: the same buffer for every K
    for t in range(T):
        if scen == "enc":
            agent.select_action(obs[t])
        else:  # select_action (PPO.py:155-162) after its extract_features line
            with torch.no_grad():
                action, action_logprob = agent.policy_old.act(direct[t])
            agent.buffer.states.append(direct[t])
            agent.buffer.actions.append(action)
            agent.buffer.logprobs.append(action_logprob)
        agent.buffer.rewards.append(torch.tensor(rewards[t]))  # trainRL.py:203: 0-dim tensors
        agent.buffer.is_terminals.append(terminals[t])
    states = torch.squeeze(torch.stack(agent.buffer.states, dim=0)).detach()
    actions = torch.squeeze(torch.stack(agent.buffer.actions, dim=0)).detach()
    logprobs = torch.squeeze(torch.stack(agent.buffer.logprobs, dim=0)).detach()
    init = {k: v.detach().clone().numpy() for k, v in _heads(agent.policy).items()}
    with torch.no_grad():
        ev_lp, ev_val, ev_ent = agent.policy.evaluate(states, actions)
    buf = dict(features=states.numpy(), actions=actions.numpy(), logprobs=logprobs.numpy(),
               rewards=np.asarray(rewards, np.float32), terminals=np.asarray(terminals, bool),
               eval_logprobs=ev_lp.numpy(), eval_values=ev_val.squeeze(-1).numpy(), eval_entropy=ev_ent.numpy(),
               **{f"init_{k}": v for k, v in init.items()})
    for k, v in buf.items():
        if f"{scen}_{k}" in out:  # same weights, same noise: the same buffer for every K
            assert np.array_equal(out[f"{scen}_{k}"], v), k
        out[f"{scen}_{k}"] = v
    if f64:
        for pol in (agent.policy, agent.policy_old):
            pol.model.action_head.double()
            pol.model.value_head.double()
            pol.action_var = pol.action_var.double()
        agent.buffer.states = [x.double() for x in agent.buffer.states]
        agent.buffer.actions = [x.double() for x in agent.buffer.actions]
        agent.buffer.logprobs = [x.double() for x in agent.buffer.logprobs]
        agent.buffer.rewards = [x.double() for x in agent.buffer.rewards]

        class _Mse64(torch.nn.Module):  # update() builds its returns as float32 (PPO.py:187): MSELoss wants one dtype
            def forward(self, values, returns):
                return torch.nn.functional.mse_loss(values, returns.to(values.dtype))

        agent.MseLoss = _Mse64()
        agent.update()
        for k, v in _heads(agent.policy).items():
            assert v.dtype == torch.float64
            out[f"{scen}_final64_{k}_K{K}"] = v.detach().clone().numpy()
        return
    spy = _Spy()
    agent.MseLoss = spy
    agent.update()
    assert len(spy.vloss) == K and len(agent.buffer.states) == 0
    out[f"{scen}_returns_norm_K{K}"] = spy.returns.numpy()
    out[f"{scen}_vloss_K{K}"] = np.asarray(spy.vloss, np.float64)
    out[f"{scen}_values_last_K{K}"] = spy.values[-1].numpy()  # the critic's outputs entering the LAST epoch
    old = _heads(agent.policy_old)
    for k, v in _heads(agent.policy).items():
        out[f"{scen}_final_{k}_K{K}"] = v.detach().clone().numpy()
        assert torch.equal(old[k], v)  # PPO.py:220: the old policy is synchronised


if __name__ == "__main__":
    main()
