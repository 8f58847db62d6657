"""Fused (occ_ppo_update) against torch PPO epochs: parameter and loss differences after K epochs (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from occlusionenv_amd import ppo, rollout

g = torch.Generator(device="cuda").manual_seed(7)
r = torch.randn(20, 128, rollout.RECORD_FLOATS, device="cuda", generator=g)
r[..., :256] = r[..., :256].abs() * 0.5
r[..., 256:258] *= 0.6
r[..., 258] = -1.5 + 0.3 * r[..., 258]
r[..., 260] = (torch.rand(20, 128, device="cuda", generator=g) < 0.05).float()
for K in (1, 2, 3, 5, 10, 20, 40, 80):
    res = []
    for fused in (True, False):
        agent = ppo.BatchedPPO(device="cuda", seed=3, K_epochs=K, graph_epochs=False, fused=fused)
        for t in range(r.shape[0]):
            agent.store(r[t])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = agent.update()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.append(([p.detach().clone() for p in agent.policy.parameters()], st, dt))
    d = [float((a - b).abs().max()) for a, b in zip(res[0][0], res[1][0])]
    print("K %2d  max |param diff| W_a %.2e b_a %.2e W_v %.2e b_v %.2e   loss_last %.7f / %.7f  vloss_last %.7f / %.7f  loss_first %.7f / %.7f  time fused %.2f ms torch %.2f ms" % (
        K, d[0], d[1], d[2], d[3], res[0][1]["loss_last"], res[1][1]["loss_last"], res[0][1]["value_loss_last"], res[1][1]["value_loss_last"],
        res[0][1]["loss_first"], res[1][1]["loss_first"], res[0][2] * 1e3, res[1][2] * 1e3), flush=True)
