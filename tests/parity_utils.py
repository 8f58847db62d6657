"""Shared helpers of the GPU parity tests, smoke() and bench.py's cpu_baseline leg: run the same seeded
scenes through the HIP engine and through the CPU oracle (oracle/p3d_restate.py) and report differences.
This is checker code: it is the only place (with tests/) where oracle/ and the product meet."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_case(n_env, seed, mesh="teapot", az_range=0.6, pool=None, device="cuda"):
    """Seeded scenes as in SURVEY.md §8d config 2: x2 ~ N(0,1), az ~ U(-az_range, az_range), el = 0,
    action ~ N(0,1)^2.  Returns dict with the pool, mesh ids, offsets, az, actions (CPU tensors)."""
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet, load_obj

    g = torch.Generator().manual_seed(seed)
    pool = pool or MeshPool(device)
    if mesh == "teapot":
        v, f = load_obj(os.path.join(ROOT, "tests", "golden", "teapot.obj"))
        ids = [pool.add(v, f, key="teapot")]
    else:
        n_models, mixed = (8, True) if mesh == "mixed" else (6, False)
        ds = SyntheticShapeNet(n_models=n_models, seed=1234 + seed, mixed=mixed, textured=(mesh == "textured"))
        ids = [pool.add(*ds.models[i], key=("syn", mesh, seed, i), atlas=ds.atlases[i]) for i in range(n_models)]
    x2 = torch.randn(n_env, generator=g)
    az = (torch.rand(n_env, generator=g) * 2 - 1) * az_range
    actions = torch.randn(n_env, 2, generator=g)
    pick = torch.randint(0, len(ids), (n_env, 3), generator=g)
    mesh_ids = torch.tensor(ids)[pick]
    offsets = torch.zeros(n_env, 3, 3)
    offsets[:, 1, 0], offsets[:, 1, 2] = x2, 1.0
    offsets[:, 2, 0], offsets[:, 2, 2] = -x2, 2.0
    return dict(pool=pool, mesh_ids=mesh_ids, offsets=offsets, az=az, actions=actions)


def oracle_env(case, i, img):
    from oracle import p3d_restate as O

    objs, atl = [], []
    for o in range(3):
        mid = int(case["mesh_ids"][i, o])
        v, f = case["pool"].get(mid)
        objs.append((v + case["offsets"][i, o], f))
        atl.append(case["pool"].get_atlas(mid))
    return O.OracleEnv(objs, img, atlases=atl if all(a is not None for a in atl) else None)


def run_engine(case, img, n_env=None, faces_per_pixel=100, radius=4.0):
    from occlusionenv_amd.engine import OcclusionEngine

    n = n_env or case["mesh_ids"].shape[0]
    eng = OcclusionEngine(case["pool"], n, img, faces_per_pixel=faces_per_pixel)
    eng.set_scene(list(range(n)), case["mesh_ids"][:n], case["offsets"][:n])
    obs0, loss0, fs0 = eng.reset_render(None, radius, case["az"][:n], 0.0)
    a = case["actions"][:n].to(eng.device).requires_grad_(True)
    obs, reward, done, fs, loss = eng.step(a)
    reward.sum().backward()
    eng.check_status()
    return dict(engine=eng, obs0=obs0.cpu(), loss0=loss0.cpu(), fs0=fs0.cpu(), obs=obs.cpu(), reward=reward.detach().cpu(),
                done=done.cpu(), fs=fs.cpu(), loss=loss.cpu(), grad=a.grad.cpu(), alphas=eng.alphas.cpu(),
                campos=eng.camera_position.cpu())


def run_parity_case(n_env=2, img=64, seed=0, mesh="teapot", az_range=0.6, check_envs=None, radius=4.0, mutate=None):
    case = make_case(n_env, seed, mesh, az_range)
    if mutate is not None:
        mutate(case)  # e.g. push an object out of view
    got = run_engine(case, img, radius=radius)
    res = dict(alpha_flip_frac=0.0, obs_texel_mismatch=0.0, obs_maxabs=0.0, alpha_maxabs=0.0, fs_maxabs=0.0, loss_rel=0.0, reward_abs=0.0, grad_rel=0.0,
               obs0_maxabs=0.0, loss0_rel=0.0, depth_mismatch=0.0)
    for i in (check_envs if check_envs is not None else range(n_env)):
        env = oracle_env(case, i, img)
        obs0 = env.reset(radius=radius, azimuth=float(case["az"][i]))
        a = case["actions"][i].clone().requires_grad_(True)
        obs, reward, done, info = env.step(a)
        reward.backward()
        # pixels where the nearest face differs (depth jump) are counted, not diffed
        d_or, d_hip = obs[0, 3], got["obs"][i, 3]
        mism = (d_or - d_hip).abs() > 1e-3
        res["depth_mismatch"] = max(res["depth_mismatch"], float(mism.float().mean()))
        ok = ~mism
        dobs = ((obs[0] - got["obs"][i]).abs() * ok).detach()
        if mesh == "textured":  # count pixels whose atlas texel differs (boundary flips) instead of diffing them
            bad = dobs[:3].max(0).values > 1e-4
            res["obs_texel_mismatch"] = max(res["obs_texel_mismatch"], float(bad.float().mean()))
            dobs = dobs * (~bad)
        res["obs_maxabs"] = max(res["obs_maxabs"], float(dobs.max()))
        d0 = (obs0[0] - got["obs0"][i]).abs()
        res["obs0_maxabs"] = max(res["obs0_maxabs"], float((d0 * ((obs0[0, 3] - got["obs0"][i, 3]).abs() <= 1e-3)).max()))
        al = torch.stack([im[0, ..., 3] for im in env.alphas]).detach()
        dal = (al - got["alphas"][i]).abs()
        res["alpha_maxabs"] = max(res["alpha_maxabs"], float(dal.max()))
        # pixels beyond the tolerance: a candidate flipped at the blur boundary (|d alpha| <= 1e-4) or the K-th and
        # (K+1)-th nearest faces swapped on a rounding-level depth near-tie (dense meshes) - counted, like depth flips
        res["alpha_flip_frac"] = max(res["alpha_flip_frac"], float((dal > 1e-4).float().mean()))
        res["fs_maxabs"] = max(res["fs_maxabs"], float((info["full_state"][0].detach() - got["fs"][i]).abs().max()))
        lo = float(info["full_reward"])
        res["loss_rel"] = max(res["loss_rel"], abs(lo - float(got["loss"][i])) / max(abs(lo), 1.0))
        res["loss0_rel"] = max(res["loss0_rel"], abs(float(env.objectMass) - 1 - float(got["loss0"][i])) / max(abs(lo), 1.0))
        res["reward_abs"] = max(res["reward_abs"], abs(float(reward) - float(got["reward"][i])))
        g = a.grad
        res["grad_rel"] = max(res["grad_rel"], float((g - got["grad"][i]).norm() / g.norm().clamp(min=1e-6)))
        assert bool(done) == bool(got["done"][i])
    return res
