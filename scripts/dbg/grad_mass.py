"""Diagnostic (GPU box): fp32 noise scale of the action gradient = eps * L1 mass of its per-pixel terms.
   python scripts/dbg/grad_mass.py seed[:mesh:img:az:radius] ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scripts.parity_sweep import case_of
from tests import parity_utils as PU
from occlusionenv_amd import _native as nat

for arg in sys.argv[1:]:
    parts = arg.split(":")
    seed = int(parts[0])
    c = case_of(seed)
    if len(parts) > 1:
        c = dict(n_env=2, img=int(parts[2]), seed=seed, mesh=parts[1], az_range=float(parts[3]), radius=float(parts[4]))
    case = PU.make_case(c["n_env"], seed, c["mesh"], c["az_range"])
    got = PU.run_engine(case, c["img"], radius=c["radius"])
    eng = got["engine"]
    N, S = c["n_env"], c["img"]
    og = eng._ws_tensors["obj_grad"].view(torch.float32)[: N * 3 * S * S * 2].view(N, 3, S, S, 2).cpu()
    al = got["alphas"]
    for i in range(N):
        a = al[i]
        I = a[0] * a[1] + a[1] * a[2] + a[0] * a[2]
        gsum = torch.stack([a[1] + a[2], a[0] + a[2], a[0] + a[1]])
        valid = (a > 0)
        term = (2 * I)[None, :, :, None] * gsum[..., None] * torch.where(valid[..., None], og[i], torch.zeros(()))
        mass = term.abs().sum((0, 1, 2))          # (2,) over el, az
        net = term.sum((0, 1, 2))
        J = eng.cam[i, nat.C_J:nat.C_J + 4].cpu().reshape(2, 2)
        om = float(eng.object_mass[i])
        Ma = (J.abs().t() @ mass) / om
        ga = -(J.t() @ net) / om
        env = PU.oracle_env(case, i, S)
        env.reset(radius=c["radius"], azimuth=float(case["az"][i]))
        ao = case["actions"][i].clone().requires_grad_(True)
        _, r, _, _ = env.step(ao)
        r.backward()
        g64 = PU._oracle_grad64(case, i, S, c["radius"], torch.ones(S, S))
        e_gpu = float((got["grad"][i].double() - g64).norm())
        e_orc = float((ao.grad.double() - g64).norm())
        print("seed %d env %d |g| %.3e  M_a %.3e (mass/|g| %.0f)  e_gpu %.2e = %.1f eps*M  e_orc32 %.2e = %.1f eps*M   recomposed-vs-gpu %.1e" % (
            seed, i, float(g64.norm()), float(Ma.norm()), float(Ma.norm() / g64.norm()), e_gpu, e_gpu / (6e-8 * float(Ma.norm())),
            e_orc, e_orc / (6e-8 * float(Ma.norm())), float((ga - got["grad"][i]).norm())), flush=True)
