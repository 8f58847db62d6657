"""GPU tests of the drop-in surface (OcclusionEnv / SimpleVecEnv) and of size-independent properties at the
bench workload's full size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _deterministic_scenes():
    from occlusionenv_amd import environment

    environment.seed_scene_rng(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    yield
    environment.seed_scene_rng(None)


@pytest.fixture(scope="module")
def ds():
    from occlusionenv_amd.meshes import SyntheticShapeNet

    return SyntheticShapeNet(n_models=8, seed=1234)


def test_single_env_contract():
    from environment import OcclusionEnv  # the reference's module name (trainRL.py:9)

    env = OcclusionEnv(img_size=64)  # default scene: three teapots
    env.seed(3)
    obs = env.reset()
    assert obs.shape == (1, 4, 64, 64) and obs.is_cuda and obs.dtype == torch.float32
    assert env.elevation.shape == (1,) and env.azimuth.shape == (1,) and env.radius.shape == (1,)
    assert env.camera_position.shape == (3,) and float(env.camera_position.abs().sum()) == 0.0  # reset leaves it 0
    assert float(env.objectMass) == pytest.approx(float(env.fullReward) + 1.0)
    assert env.observation_space.shape == (4, 64, 64) and env.action_space.shape == (2,)
    assert len(env.meshes) == 4 and env.meshes[0].faces_list()[0].shape[0] == 3 * 2464
    action = torch.nn.Parameter(torch.zeros(2, device="cuda"))  # demo.py:80
    obs, reward, done, info = env.step(action)
    assert obs.shape == (1, 4, 64, 64) and reward.dim() == 0 and done.dim() == 0 and done.dtype == torch.bool
    assert set(info) == {"full_state", "position", "full_reward"} and info["full_state"].shape == (1, 64, 64, 4)
    reward.backward()
    assert action.grad.shape == (2,) and torch.isfinite(action.grad).all()
    if done:  # usable in `if` like the reference (environment.py:389)
        pass
    with torch.no_grad():
        action += 0.1 * action.grad
    obs2, reward2, _, info2 = env.step(action.detach())
    assert not reward2.requires_grad
    assert torch.allclose(info2["position"].norm(), torch.tensor(4.0, device="cuda"), atol=1e-4)
    image, depth = env.render()
    assert image.shape == (1, 64, 64, 4) and depth.shape == (1, 64, 64, 1)
    # render() runs the hard-only kernel variant: same arithmetic, not necessarily the same FMA contraction
    assert torch.allclose(depth[0, ..., 0], obs2[0, 3], atol=1e-5)
    assert torch.allclose(image[0, ..., :3], obs2[0, :3].permute(1, 2, 0), atol=1e-5)
    assert torch.equal(image[0, ..., 3] == 1.0, obs2[0, 3] != -1.0)
    env.close()


def test_vecenv_contract_and_batched_equals_single(ds):
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv  # train_predict.py:3

    np.random.seed(0)
    N, S = 6, 64
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    venv.seed(0)
    obs0 = venv.reset()
    assert obs0.shape == (N, 1, 4, S, S)  # the reference stacks (1,4,S,S) observations (SubProcVecEnv.py:230-235)
    step = torch.nn.Parameter(torch.randn(N, 2, device="cuda"))  # train_predict.py:48
    eng = venv.engine
    state = {k: getattr(eng, k).clone() for k in ("elevation", "azimuth", "radius", "full_reward", "object_mass")}
    scene = (eng.scene_mesh.clone(), eng.scene_offset.clone())
    obs, rewards, finished, info = venv.step(step)
    rewards.sum().backward()  # train_predict.py:52
    assert obs.shape == (N, 4, S, S) and rewards.shape == (N,) and finished.shape == (N,) and finished.dtype == torch.bool
    assert step.grad.shape == (N, 2) and len(info) == N and set(info[0]) >= {"full_state", "position", "full_reward"}
    # the same envs one at a time through single-env engines: bitwise equal (env-independent, fixed-order sums)
    from occlusionenv_amd.engine import OcclusionEngine

    for i in range(N):
        if bool(finished[i]):
            continue  # was auto-reset
        e1 = OcclusionEngine(eng.pool, 1, S)
        e1.set_scene([0], scene[0][i:i + 1].cpu(), scene[1][i:i + 1].cpu())
        for k, v in state.items():
            getattr(e1, k)[0] = v[i]
        a = step.detach()[i:i + 1].clone().requires_grad_(True)
        o1, r1, d1, fs1, l1 = e1.step(a)
        r1.sum().backward()
        assert torch.equal(o1[0], obs[i]) and torch.equal(r1.detach()[0], rewards.detach()[i])
        assert torch.equal(a.grad[0], step.grad[i])


def test_auto_reset_on_done(ds):
    from occlusionenv_amd.engine import OcclusionEngine
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    np.random.seed(1)
    N, S = 4, 64
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    # reset() itself draws unseeded azimuths in +-40 rad (SubProcVecEnv.py:233): a camera on the x axis would see
    # the objects moved below lined up behind object 1 - fix the views instead
    venv._reset_envs(list(range(N)), torch.zeros(N))
    eng = venv.engine
    # push env 2's objects far apart: no occlusion -> loss < 0.1 -> done -> auto reset
    off = eng.scene_offset[2].clone()
    off[1, 0], off[2, 0] = 50.0, -50.0
    eng.scene_offset[2] = off
    obs, rewards, dones, infos = venv.step(torch.randn(N, 2, device="cuda"))
    assert bool(dones[2]) and "terminal_observation" in infos[2] and infos[2]["terminal_observation"].shape == (1, 4, S, S)
    assert float(rewards[2]) > 4.0  # +5 bonus (environment.py:389-390)
    assert float(eng.azimuth[2]) == 0.0 and float(eng.camera_position[2].abs().sum()) == 0.0  # reset() defaults
    assert float(eng.object_mass[2]) == pytest.approx(float(eng.full_reward[2]) + 1.0)
    for i in (0, 1, 3):
        if not bool(dones[i]):
            assert "terminal_observation" not in infos[i] and float(rewards[i]) < 4.0


def test_auto_reset_from_the_speculative_reserve(ds):
    """N = 16 -> 4 reserve scenes ride along with every step; a finished env takes one without an extra render.
    The installed state must be exactly what a synchronous reset() of that scene computes."""
    from occlusionenv_amd.engine import OcclusionEngine
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    np.random.seed(2)
    N, S = 16, 64
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    venv._reset_envs(list(range(N)), torch.zeros(N))  # reset() itself draws unseeded azimuths (SubProcVecEnv.py:233)
    venv._warm_reserve()
    eng = venv.engine
    assert eng.R == 4 and bool((venv._rs_state == 2).all()) and eng.rs_state.tolist() == [2, 2, 2, 2]
    off = eng.scene_offset[5].clone()
    off[1, 0], off[2, 0] = 50.0, -50.0  # no occlusion left -> env 5 finishes
    eng.scene_offset[5] = off
    ready_scenes = [venv._rs_scene[r] for r in range(4)]
    # zero actions leave every camera where it is (environment.py:358): only env 5 can finish
    obs, rewards, dones, infos = venv.step(torch.zeros(N, 2, device="cuda"))
    assert bool(dones[5]) and "terminal_observation" in infos[5]
    assert int(dones.sum()) <= 4, "at most the four reserve slots are needed"
    assert venv.envs[5]._scene in ready_scenes  # taken from the reserve
    ids, offs = venv.envs[5]._scene
    ref = eng.evaluate_scenes([ids], [offs], 4.0, 0.0, 0.0)
    assert torch.equal(ref["obs"][0], obs[5])
    assert float(eng.full_reward[5]) == float(ref["loss"][0]) and float(eng.object_mass[5]) == float(ref["loss"][0] + 1.0)
    assert float(eng.azimuth[5]) == 0.0 and float(eng.elevation[5]) == 0.0 and float(eng.radius[5]) == 4.0
    assert float(eng.camera_position[5].abs().sum()) == 0.0
    assert torch.equal(eng.scene_offset[5].cpu(), torch.tensor(offs, dtype=torch.float32))
    # the next step runs on the new scene
    obs2, r2, d2, _ = venv.step(torch.randn(N, 2, device="cuda"))
    assert torch.isfinite(r2).all()


def test_norm_with_object_size(ds):
    """normWithObjectSize = True (environment.py:208,320,324): reset() sets objectMass = sum_px (a1 + a2 + a3)^2 + 1 instead of
    loss + 1, and step() divides the reward by it (:387).  (a) single env against the oracle's restatement of the same
    lines; (b) batched: the synchronous reset, the device-side auto-reset from the reserve (the slot's stored sum) and a
    mixed batch in which only some envs have the flag."""
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv
    from tests.parity_utils import make_case, oracle_env, run_engine

    # (a) one env, engine against oracle on the same scene and action
    case = make_case(2, 41, "synthetic")
    from occlusionenv_amd.engine import OcclusionEngine

    S = 64
    eng = OcclusionEngine(case["pool"], 2, S)
    eng.set_scene([0, 1], case["mesh_ids"], case["offsets"])
    eng.set_norm_with_object_size(1, True)  # env 1 only
    eng.reset_render(None, 4.0, case["az"], 0.0)
    al = eng.alphas[1].clone()  # the reset render's silhouettes
    assert float(eng.object_mass[1]) == pytest.approx(float((al.sum(0) ** 2).sum()) + 1.0, rel=1e-5)
    assert float(eng.object_mass[0]) == pytest.approx(float(eng.full_reward[0]) + 1.0)
    a = case["actions"].cuda().requires_grad_(True)
    _, reward, _, _, _ = eng.step(a)
    reward.sum().backward()
    for i, norm in ((0, False), (1, True)):
        env = oracle_env(case, i, S)
        env.normWithObjectSize = norm
        env.reset(azimuth=float(case["az"][i]))
        ao = case["actions"][i].clone().requires_grad_(True)
        _, r, _, _ = env.step(ao)
        r.backward()
        assert float(eng.object_mass[i]) == pytest.approx(float(env.objectMass), rel=1e-5), (i, norm)
        assert float(reward[i]) == pytest.approx(float(r), abs=1e-4)
        assert float((a.grad[i].cpu() - ao.grad).norm()) <= 1e-4 * max(float(ao.grad.norm()), 1e-3) + 1e-7

    # (b) the batched env: flags follow the env objects' attribute through every reset path
    np.random.seed(4)
    N = 16
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    venv.set_attr("normWithObjectSize", True, indices=[3, 5, 6])
    assert venv.get_attr("normWithObjectSize", indices=[3, 4]) == [True, False]
    venv._reset_envs(list(range(N)), torch.zeros(N))
    venv._warm_reserve()
    e2 = venv.engine

    def expect(i):
        al = e2.alphas[i]
        return float((al.sum(0) ** 2).sum()) + 1.0

    for i in range(N):
        want = expect(i) if i in (3, 5, 6) else float(e2.full_reward[i]) + 1.0
        assert float(e2.object_mass[i]) == pytest.approx(want, rel=1e-5), i
    # env 5 (flag on) and env 7 (flag off) finish: both are reset on the device from the reserve
    for i in (5, 7):
        off = e2.scene_offset[i].clone()
        off[1, 0], off[2, 0] = 50.0, -50.0
        e2.scene_offset[i] = off
    obs, rewards, dones, infos = venv.step(torch.zeros(N, 2, device="cuda"))
    assert bool(dones[5]) and bool(dones[7]) and "terminal_observation" in infos[5]
    assert float(e2.object_mass[5]) == pytest.approx(expect(5), rel=1e-5)          # the slot's stored silhouette mass
    assert float(e2.object_mass[7]) == pytest.approx(float(e2.full_reward[7]) + 1.0)  # the slot's stored loss + 1
    assert expect(5) != pytest.approx(float(e2.full_reward[5]) + 1.0, rel=1e-3)   # and the two really differ
    # the slots refilled after that go through the step launches (PENDING -> READY on the device) and are used too
    for t in range(6):
        venv.step(torch.zeros(N, 2, device="cuda"))
    venv._drain()
    off = e2.scene_offset[3].clone()
    off[1, 0], off[2, 0] = 50.0, -50.0
    e2.scene_offset[3] = off
    _, _, dones, _ = venv.step(torch.zeros(N, 2, device="cuda"))
    assert bool(dones[3])
    assert float(e2.object_mass[3]) == pytest.approx(expect(3), rel=1e-5)
    e2.check_status()


def test_auto_reset_deferred_report_and_dry_reserve(ds):
    """The auto-reset runs on the device; the host reads its report at the NEXT step (or when infos are read).
    Five envs finish at once with four reserve slots: four are reset from the reserve by the device, the fifth by
    the synchronous fallback when the report is read; the emptied slots get new scenes and become READY again."""
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    np.random.seed(3)
    N, S = 16, 64
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    venv._reset_envs(list(range(N)), torch.zeros(N))
    venv._warm_reserve()
    eng = venv.engine
    fin = (3, 5, 7, 9, 11)
    for i in fin:
        off = eng.scene_offset[i].clone()
        off[1, 0], off[2, 0] = 50.0, -50.0
        eng.scene_offset[i] = off
    old_scenes = {i: venv.envs[i]._scene for i in fin}
    obs, rewards, dones, infos = venv.step(torch.zeros(N, 2, device="cuda"))
    assert venv._pending is not None, "report not read yet: the host ran ahead"
    assert dones[list(fin)].all()
    # device state already reset for the four lowest finished envs (pairing is in index order)
    assert all(float(eng.camera_position[i].abs().sum()) == 0.0 for i in fin[:4])
    assert sorted(eng.rs_state.tolist()) == [0, 0, 0, 0]
    # reading infos forces the bookkeeping, incl. the fallback reset of env 11
    for i in fin:
        assert infos[i]["terminal_observation"].shape == (1, 4, S, S)
        assert venv.envs[i]._scene is not old_scenes[i]
        assert float(eng.object_mass[i]) == pytest.approx(float(eng.full_reward[i]) + 1.0)
        ids, offs = venv.envs[i]._scene
        ref = eng.evaluate_scenes([ids], [offs], 4.0, 0.0, 0.0)
        assert torch.equal(ref["obs"][0], obs[i])
    assert venv._pending is None and eng.rs_state.tolist() == [1, 1, 1, 1]  # refilled, under test
    # a few more steps: the rejection loop advances on the device until both slots are READY again
    for _ in range(12):
        venv.step(torch.zeros(N, 2, device="cuda"))
    venv._drain()
    assert eng.rs_state.tolist() == [2, 2, 2, 2] and venv._rs_state.tolist() == [2, 2, 2, 2]


def test_z_clipped_batch_properties():
    """Cameras inside the scenes (radius 0.8 .. 1.6): many faces cross z = 0.5 (clip cases 3 and 4, split quads).
    No oracle at this size: status clean, rects in range, alphas in [0,1], finite loss and gradients."""
    from tests.parity_utils import make_case
    from occlusionenv_amd.engine import OcclusionEngine

    N, S = 256, 128
    case = make_case(N, 21, "mixed")
    eng = OcclusionEngine(case["pool"], N, S)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    radius = torch.linspace(0.8, 1.6, N)
    eng.reset_render(None, radius, case["az"] * 3.0, 0.0)
    for _ in range(3):
        a = torch.randn(N, 2, device="cuda", requires_grad=True)
        obs, reward, done, fs, loss = eng.step(a)
        reward.sum().backward()
        eng.check_status()
        assert torch.isfinite(loss).all() and torch.isfinite(a.grad).all() and torch.isfinite(obs).all()
        al = eng.alphas
        assert float(al.min()) >= 0.0 and float(al.max()) <= 1.0
        rect = eng._ws_tensors["objrect"][: 12 * N].view(-1, 4)
        vis = eng._ws_tensors["nrec"][: 3 * N] > 0
        assert int(rect[vis].min()) >= 0 and int(rect[vis].max()) < S // 4
        fg = obs[:, 3] != -1.0
        assert float(obs[:, 3][fg].min()) >= 0.5 - 1e-4  # nothing in front of the clip plane is ever drawn


def test_soak_many_steps_with_auto_reset(ds):
    """A long batched rollout (512 envs x 120 steps, random actions): auto-resets from the reserve keep flowing,
    every work-item rect stays inside the image, no status bit is raised, outputs stay finite."""
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    N, S = 512, 128
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    az0 = (torch.rand(N, generator=torch.Generator().manual_seed(1)) * 2 - 1) * 0.6
    venv._reset_envs(list(range(N)), az0)
    venv._warm_reserve()
    eng = venv.engine
    gen = torch.Generator(device="cuda").manual_seed(3)
    n_done = 0
    for step in range(120):
        a = torch.randn(N, 2, device="cuda", generator=gen, requires_grad=True)
        obs, rewards, dones, infos = venv.step(a)
        rewards.sum().backward()
        if step % 10 == 9:
            n_done += int(dones.sum())
            assert torch.isfinite(rewards).all() and torch.isfinite(a.grad).all() and torch.isfinite(obs).all()
            rect = eng._ws_tensors["objrect"][: 12 * eng.NT].view(-1, 4)
            vis = eng._ws_tensors["nrec"][: 3 * eng.NT] > 0
            assert int(rect[vis].min()) >= 0 and int(rect[vis].max()) < S // 4
    venv._drain()
    eng.check_status()
    assert n_done > 0, "some envs must have finished and been reset along the way"


def test_harness_gradient_ascent_and_rollout(ds):
    """H1 counterparts: demo.py's gradient-ascent loop reduces the occlusion; a T-step batched rollout yields
    well-formed 261-float records and finite action gradients."""
    from occlusionenv_amd import harness, rollout
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    np.random.seed(5)
    env = OcclusionEnv(ds, img_size=64)
    log, action = harness.gradient_ascent(env, steps=12, lr=0.05)
    assert len(log) >= 1 and all(np.isfinite(r) for r, _, _ in log)
    fulls = [f for _, f, _ in log]
    assert all(np.isfinite(f) and f >= 0 for f in fulls)
    N, T = 8, 6
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=64) for _ in range(N)])
    out = harness.collect_rollout(venv, T=T)
    assert out["records"].shape == (T, N, rollout.RECORD_FLOATS) and out["action_grads"].shape == (T, N, 2)
    assert torch.isfinite(out["records"]).all() and torch.isfinite(out["action_grads"]).all()
    assert set(out["records"][..., 260].unique().tolist()) <= {0.0, 1.0}


def test_ppo_rollout_and_update_end_to_end(ds):
    """trainRL.py:189-229 batched: T vectorised steps into the features-only buffer, then one PPO update."""
    from occlusionenv_amd import ppo
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    N, T = 8, 6
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=64) for _ in range(N)])
    agent = ppo.BatchedPPO(K_epochs=5, device="cuda", seed=0)
    stats = ppo.train_rollouts(venv, agent, n_updates=2, T=T, with_action_grad=True)
    assert len(stats) == 2 and all(st["samples"] == T * N for st in stats)
    assert all(np.isfinite(st["loss_first"]) and np.isfinite(st["loss_last"]) and np.isfinite(st["mean_reward"]) for st in stats)


def test_config5_per_rank_ppo_rollout_and_update():
    """BASELINE config 5 at its per-rank size (2 048 envs over 8 GPUs = 256 envs/GPU, 256x256): one rollout of T = 50
    vectorised steps with the differentiable-reward backward to the action, then one heads-only PPO update
    (/root/reference/trainRL.py:189-229, PPO.py:152-223)."""
    from occlusionenv_amd import ppo
    from occlusionenv_amd.meshes import SyntheticShapeNet
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    N, T, S = 256, 50, 256
    pool = SyntheticShapeNet(n_models=64, seed=1234)
    venv = SimpleVecEnv([lambda: OcclusionEnv(pool, img_size=S) for _ in range(N)])
    agent = ppo.BatchedPPO(device="cuda", seed=0)  # trainRL.py hyper-parameters: 80 epochs, clip 0.2, gamma 0.99
    gen = torch.Generator(device="cuda").manual_seed(11)
    grads = []
    stats = ppo.train_rollouts(venv, agent, n_updates=1, T=T, with_action_grad=True, generator=gen,
                               on_step=lambda action, rewards: grads.append(action.grad.detach().clone()))
    assert len(stats) == 1 and stats[0]["samples"] == T * N
    st = stats[0]
    assert np.isfinite(st["loss_first"]) and np.isfinite(st["loss_last"]) and np.isfinite(st["mean_reward"])
    # 80 Adam epochs on the value head shrink its regression loss on the (fixed) normalised returns; the TOTAL is not
    # monotone, because the advantages returns - value are re-evaluated with the improving critic every epoch
    assert np.isfinite(st["value_loss_last"]) and st["value_loss_last"] < st["value_loss_first"]
    g = torch.stack(grads)
    assert g.shape == (T, N, 2) and torch.isfinite(g).all() and float(g.abs().max()) > 0.0
    venv._drain()
    venv.engine.check_status()  # no status word set by any kernel of the 50 steps
    assert int(venv.engine.status.abs().max()) == 0
    for p_new, p_old in zip(agent.policy.parameters(), agent.policy_old.parameters()):
        assert torch.equal(p_new, p_old) and torch.isfinite(p_new).all()


def test_dataset_generator_and_no_grad_steps(ds, tmp_path):
    """datasetGenerator.py:76-124 batched (4 runs x 3 frames) + the reader; steps without autograd use the
    gradient-free kernel variants."""
    from occlusionenv_amd import dataset_io
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    N = 16
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=64) for _ in range(N)])
    n = dataset_io.generate(venv, str(tmp_path), num_frames=3)
    assert n == N
    data = dataset_io.OcclusionDataset(str(tmp_path), size=(32, 32))
    assert len(data) == 3 * N
    img, label, pos, grad = data[5]
    assert img.shape == (4, 32, 32) and label.shape == (1, 32, 32) and np.isfinite(grad.numpy()).all()
    with torch.no_grad():
        for _ in range(3):
            obs, rewards, dones, infos = venv.step(torch.randn(N, 2, device="cuda"))
    assert not rewards.requires_grad and torch.isfinite(rewards).all() and obs.shape == (N, 4, 64, 64)
    assert set(infos[0].keys()) >= {"full_state", "position", "full_reward"}


def test_dataset_generator_keeps_a_run_on_its_scene_when_envs_finish(tmp_path):
    """datasetGenerator.py:80-117 never resets on ``finished``: all frames of a run come from one scene and one
    camera trajectory.  Scenes whose objects do not overlap report finished = True at every step; the runs must
    still walk on in 0.05-steps (an auto-reset would put elevation / azimuth back to 0)."""
    import pickle

    from occlusionenv_amd import dataset_io
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    N, F = 4, 5
    venv = SimpleVecEnv([lambda: OcclusionEnv(None, img_size=64) for _ in range(N)])
    eng = venv.engine
    real_reset = venv.reset

    def reset_far_apart():
        obs = real_reset()
        off = torch.zeros(N, 3, 3)
        off[:, 1, 0], off[:, 1, 2] = 3.5, 1.0   # x2 = 3.5: the three teapots do not overlap on screen
        off[:, 2, 0], off[:, 2, 2] = -3.5, 2.0
        eng.set_scene(list(range(N)), eng.scene_mesh.cpu(), off)
        eng.reset_render(None, 4.0, 0.0, 0.0)
        return obs

    venv.reset = reset_far_apart
    assert dataset_io.generate(venv, str(tmp_path), num_frames=F) == N
    for i in range(N):
        arr = np.asarray(pickle.load(open(tmp_path / ("run_%d" % i) / "params.pickle", "rb"))).reshape(-1, 5)
        assert arr.shape == (F, 5) and np.array_equal(arr[:, 0], np.arange(F))
        traj = np.vstack([[0.0, 0.0], arr[:, 1:3]])
        steps = np.linalg.norm(np.diff(traj, axis=0), axis=1)
        assert np.allclose(steps, 0.05, atol=1e-5), steps  # one unit-normalised action per frame, no reset in between
        assert np.isfinite(arr[:, 3:5]).all()
    assert float(eng.full_reward.max()) < 0.1  # the scenes really were "finished" all along


def test_record_arrays_follow_the_scenes_not_the_largest_mesh():
    """Variable record layout: every (env, object) gets room for ITS mesh.  A pool that also holds a 20 480-face model
    must not make every slot pay for it, and a scene that does use it still renders (the arrays grow on demand)."""
    from tests.parity_utils import make_case
    from occlusionenv_amd.engine import OcclusionEngine

    N, S = 64, 64
    case = make_case(N, 31, "mixed")
    pool = case["pool"]
    faces = [pool.num_faces(m) for m in range(len(pool))]
    small = [m for m in range(len(pool)) if faces[m] == min(faces)]
    big = faces.index(max(faces))
    assert max(faces) >= 8 * min(faces)
    eng = OcclusionEngine(pool, N, S)
    ids = torch.tensor([[small[i % len(small)]] * 3 for i in range(N)])
    eng.set_scene(list(range(N)), ids, case["offsets"])
    eng.reset_render(None, 4.0, case["az"], 0.0)
    eng.check_status()
    need_small = N * 3 * 2 * min(faces)
    assert need_small <= eng._rec_total < 2 * need_small, (eng._rec_total, need_small)
    ids[5, 1] = big  # one object of one env is the big model
    eng.set_scene([5], ids[5:6], case["offsets"][5:6])
    obs, loss, fs = eng.reset_render(None, 4.0, case["az"], 0.0)
    eng.check_status()
    assert eng._rec_total >= need_small + 2 * max(faces) - 2 * min(faces) and eng._rec_total < N * 3 * 2 * max(faces) // 2
    assert torch.isfinite(loss).all() and float(eng.alphas[5, 1].max()) > 0.5  # the big object is there


def test_batched_512_reference_default_size():
    """img_size = 512 is the reference's default and what trainRL.py runs (environment.py:202, trainRL.py:75); here it is
    BATCHED: 64 envs x 512 x 512, three 5 120-face objects each.  Workspace sizes as occ_workspace_query lays them out,
    the same size-independent properties as the 128 x 128 full-size test, batched == single env bitwise, and the batched
    VecEnv path (reserve, recycled outputs, auto-reset) at this size."""
    import ctypes as C

    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv
    from occlusionenv_amd import _native as nat
    from occlusionenv_amd.engine import OcclusionEngine
    from occlusionenv_amd.meshes import SyntheticShapeNet
    from tests.parity_utils import make_case

    N, S = 64, 512
    case = make_case(N, 23, "synthetic")
    eng = OcclusionEngine(case["pool"], N, S)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    obs0, loss0, fs0 = eng.reset_render(None, 4.0, case["az"], 0.0)
    fr0, om = eng.full_reward.clone(), eng.object_mass.clone()
    a = case["actions"].cuda().requires_grad_(True)
    obs, reward, done, fs, loss = eng.step(a)
    reward.sum().backward()
    eng.check_status()
    # workspace: what occ_workspace_query says for (N, S), byte for byte
    sizes = nat.OccWorkspaceSizes()
    sc = eng._scene_struct(N, eng._mesh_all, eng._off_all)
    n_slots = eng.lib.occ_device_cu_count() * eng.waves_per_cu
    nat.check(eng.lib.occ_workspace_query(C.byref(sc), n_slots, C.byref(sizes)), "query")
    S2 = S * S
    assert sizes.obj_alpha_bytes == N * 3 * S2 * 4 and sizes.obj_grad_bytes == N * 3 * S2 * 8 and sizes.obj_hrec_bytes == N * 3 * S2 * 4
    assert sizes.partials_bytes == N * (S2 // 256) * 16 and sizes.lists_bytes == n_slots * nat.LOG_CAP * nat.LOG_ENTRY_BYTES
    T = (S // 8) ** 2
    assert sizes.order_bytes >= (N * 3 * T * 3) * 4  # per-tile (rank, class) words + 8-byte items
    for k in ("obj_alpha", "obj_grad", "obj_hz", "obj_hrec", "partials", "lists", "order"):
        assert eng._ws_tensors[k].numel() * 4 >= getattr(sizes, k + "_bytes"), k
    assert eng._rec_total >= N * 3 * 2 * 5120 and eng._rec_total < 2 * N * 3 * 2 * 5120  # records follow the scenes
    # properties
    al = eng.alphas
    assert float(al.min()) >= 0.0 and float(al.max()) <= 1.0
    I = al[:, 0] * al[:, 1] + al[:, 1] * al[:, 2] + al[:, 0] * al[:, 2]
    assert torch.allclose(fs[..., 3], I, atol=1e-6) and bool((fs[..., :3] == 3).all())
    assert torch.allclose(loss, (fs[..., 3].double() ** 2).sum((1, 2)).float(), rtol=1e-4, atol=1e-3)
    bg = obs[:, 3] == -1.0
    assert bool((obs[:, :3].permute(0, 2, 3, 1)[bg] == 1.0).all()) and float(obs[:, 3][~bg].min()) > 0.5
    exp = (fr0 - loss) / om + torch.where(loss < 0.1, torch.tensor(5.0, device="cuda"), torch.tensor(-0.2, device="cuda"))
    assert torch.allclose(reward.detach(), exp, atol=1e-5) and torch.equal(done, loss < 0.1)
    assert torch.isfinite(a.grad).all() and float(a.grad.abs().max()) > 0
    rect = eng._ws_tensors["objrect"][: 4 * 3 * N].view(-1, 4)
    vis = eng._ws_tensors["nrec"][: 3 * N] > 0
    assert int(rect[vis].min()) >= 0 and int(rect[vis].max()) < S // 4
    # batched == one env at a time, bitwise (two envs)
    for i in (3, 40):
        e1 = OcclusionEngine(case["pool"], 1, S)
        e1.set_scene([0], case["mesh_ids"][i:i + 1], case["offsets"][i:i + 1])
        e1.reset_render(None, 4.0, case["az"][i:i + 1], 0.0)
        a1 = case["actions"][i:i + 1].cuda().requires_grad_(True)
        o1, r1, d1, f1, l1 = e1.step(a1)
        r1.sum().backward()
        assert torch.equal(o1[0], obs[i]) and torch.equal(f1[0], fs[i]) and torch.equal(r1.detach()[0], reward.detach()[i])
        assert torch.equal(a1.grad[0], a.grad[i])
        del e1
    del eng, obs, fs, obs0, fs0
    torch.cuda.empty_cache()
    # the VecEnv path at this size: reserve of 16 scenes, recycled outputs, a forced auto-reset
    ds512 = SyntheticShapeNet(n_models=8, seed=1234)
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds512, img_size=S) for _ in range(N)])
    venv._reset_envs(list(range(N)), torch.zeros(N))
    venv._warm_reserve()
    e2 = venv.engine
    assert e2.R == 16 and e2.output_recycle >= 2
    off = e2.scene_offset[9].clone()
    off[1, 0], off[2, 0] = 50.0, -50.0
    e2.scene_offset[9] = off
    ptrs = set()
    for t in range(4):
        act = torch.randn(N, 2, device="cuda", requires_grad=True)
        o, r, d, infos = venv.step(act)
        r.sum().backward()
        ptrs.add(o.data_ptr())
        assert o.shape == (N, 4, S, S) and torch.isfinite(r).all() and torch.isfinite(act.grad).all()
        if t == 0:
            assert bool(d[9]) and infos[9]["terminal_observation"].shape == (1, 4, S, S)
    venv._drain()
    e2.check_status()
    assert len(ptrs) <= 3  # recycled output sets, not 4 fresh 268 MB allocations


def test_full_size_properties(ds):
    """BASELINE config 3 size (1024 envs, 128x128, ~5k-face meshes): properties that need no oracle."""
    from tests.parity_utils import make_case
    from occlusionenv_amd.engine import OcclusionEngine

    N, S = 1024, 128
    case = make_case(N, 11, "synthetic")
    eng = OcclusionEngine(case["pool"], N, S)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    obs0, loss0, fs0 = eng.reset_render(None, 4.0, case["az"], 0.0)
    fr0, om = eng.full_reward.clone(), eng.object_mass.clone()
    a = case["actions"].cuda().requires_grad_(True)
    obs, reward, done, fs, loss = eng.step(a)
    reward.sum().backward()
    eng.check_status()
    al = eng.alphas
    assert float(al.min()) >= 0.0 and float(al.max()) <= 1.0
    I = al[:, 0] * al[:, 1] + al[:, 1] * al[:, 2] + al[:, 0] * al[:, 2]
    assert torch.allclose(fs[..., 3], I, atol=1e-6) and bool((fs[..., :3] == 3).all())
    assert torch.allclose(loss, (fs[..., 3].double() ** 2).sum((1, 2)).float(), rtol=1e-4, atol=1e-3)
    bg = obs[:, 3] == -1.0
    assert bool((obs[:, :3].permute(0, 2, 3, 1)[bg] == 1.0).all()) and float(obs[:, 3][~bg].min()) > 0.5
    assert float(obs[:, :3].max()) <= 1.0 + 1e-5 and float(obs[:, :3].min()) >= 0.5 - 1e-5  # ambient floor
    exp = (fr0 - loss) / om + torch.where(loss < 0.1, torch.tensor(5.0, device="cuda"), torch.tensor(-0.2, device="cuda"))
    assert torch.allclose(reward.detach(), exp, atol=1e-5) and torch.equal(done, loss < 0.1)
    assert torch.isfinite(a.grad).all() and float(a.grad.abs().max()) > 0
    # block rect of every (env, object) stays inside the image (the raster kernel's work items come from it)
    rect = eng._ws_tensors["objrect"][: 4 * 3 * N].view(-1, 4)
    vis = eng._ws_tensors["nrec"][: 3 * N] > 0
    assert int(rect[vis].min()) >= 0 and int(rect[vis].max()) < S // 4
    # the gradient is orthogonal to the action (reward depends on a / |a| only, environment.py:356-358)
    rad = (a.grad * a.detach()).sum(1).abs() / (a.grad.norm(dim=1) * a.detach().norm(dim=1)).clamp(min=1e-12)
    assert float(rad.max()) < 1e-3
    # permutation of the batch permutes the results bitwise (envs are independent; reductions are fixed-order)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0))
    eng2 = OcclusionEngine(case["pool"], N, S)
    eng2.set_scene(list(range(N)), case["mesh_ids"][perm], case["offsets"][perm])
    eng2.reset_render(None, 4.0, case["az"][perm], 0.0)
    a2 = case["actions"][perm].cuda().requires_grad_(True)
    obs2, reward2, done2, fs2, loss2 = eng2.step(a2)
    reward2.sum().backward()
    p = perm.cuda()
    assert torch.equal(loss2, loss[p]) and torch.equal(obs2, obs[p]) and torch.equal(a2.grad, a.grad[p])


def test_round1_faulting_launch_inputs_under_the_bounds_checked_build():
    """Regression fixture for the GPU memory fault of round 1 (bench soak, step ~200-300; commit 1e5d0e4): the scene
    rows and cameras of the launch that was about to run when the fault came (OCC_DEBUG_DUMP), replayed through
    occ_render with the OCC_DBG_BOUNDS build, whose raster kernel checks every index it forms and records the first
    violation.  Root cause then: a face counted as visible and re-judged invisible left a garbage block rect; the
    setup kernel now takes ONE visibility decision per face (occ_setup.hpp).  Child process: the library is chosen
    at load time."""
    import os
    import subprocess
    import sys

    from tests.parity_utils import ROOT

    lib = os.path.join(ROOT, "occlusionenv_amd", "libocc_hip_bounds.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    code = r"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from occlusionenv_amd import _native as nat
from occlusionenv_amd.engine import OcclusionEngine, _p
from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet
z = np.load(os.path.join(%r, "tests", "golden", "fault_scene_r01.npz"))
NT = int(z["n_rows"])
ds = SyntheticShapeNet(n_models=64, seed=1234)          # bench.py's round-1 pool
pool = MeshPool("cuda")
ids = [pool.add(*ds.models[i], key=("syn", i)) for i in range(64)]
assert ids == list(range(64))
eng = OcclusionEngine(pool, NT, 128)
eng.set_scene(list(range(NT)), torch.from_numpy(z["mesh"]), torch.from_numpy(z["off"]))
eng.cam.copy_(torch.from_numpy(z["cam"]).cuda())
ws = eng._ensure_workspace()
S = 128
obs = torch.empty(NT, 4, S, S, device="cuda"); fs = torch.empty(NT, S, S, 4, device="cuda")
loss = torch.empty(NT, device="cuda"); g = torch.empty(NT, 2, device="cuda")
ro = nat.OccRenderOut()
ro.obs, ro.full_state, ro.loss, ro.alphas, ro.grad_elaz = obs.data_ptr(), fs.data_ptr(), loss.data_ptr(), eng.alphas.data_ptr(), g.data_ptr()
sc = eng._scene_struct(NT, eng.scene_mesh, eng.scene_offset)
for _ in range(2):
    nat.check(eng.lib.occ_render(C.byref(sc), _p(eng.cam), C.byref(ws), C.byref(ro),
                                 nat.RENDER_SOFT | nat.RENDER_HARD | nat.RENDER_GRAD, 100, eng._stream()), "occ_render")
torch.cuda.synchronize()
eng.check_status()
buf = (C.c_int * 8)()
eng.lib.occ_debug_fault.argtypes = [C.POINTER(C.c_int)]
assert eng.lib.occ_debug_fault(buf) == 0
print("FAULT", list(buf), "finite", bool(torch.isfinite(loss).all() and torch.isfinite(g).all() and torch.isfinite(obs).all()),
      "loss_sum", float(loss.sum()))
""" % (ROOT, ROOT)
    env = dict(os.environ, OCC_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("FAULT")][-1]
    assert "FAULT [0, 0, 0, 0, 0, 0, 0, 0] finite True" in line, line


def test_record_arrays_are_sized_after_the_pre_launch_hook():
    """Advisor finding (round 1): a reset that runs inside step()'s pre_launch hook (SimpleVecEnv's synchronous
    fallback when the reserve is dry) can hand envs LARGER models that are already in the pool - pool.version does not
    move - so the variable-layout record arrays must be sized AFTER the hook, not before.  Here the hook switches every
    env from 1 280-face to 20 480-face meshes: without the fix the tail objects get an empty record span, raise
    OCC_STATUS_REC_OVERFLOW and render nothing."""
    from occlusionenv_amd.engine import OcclusionEngine
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet

    ds = SyntheticShapeNet(n_models=12, seed=77, mixed=True)
    pool = MeshPool("cuda")
    ids = [pool.add(*ds.models[i], key=("m", i)) for i in range(len(ds.models))]
    faces = [pool.num_faces(i) for i in ids]
    small, big = faces.index(min(faces)), faces.index(max(faces))
    assert faces[big] >= 8 * faces[small]
    N, S = 6, 64
    eng = OcclusionEngine(pool, N, S, reserve=2)
    off = torch.zeros(N, 3, 3)
    off[:, 1, 0], off[:, 1, 2] = 0.4, 1.0
    off[:, 2, 0], off[:, 2, 2] = -0.4, 2.0
    eng.set_scene(list(range(N)), torch.full((N, 3), small), off)
    eng.reset_render(None, 4.0, 0.0, 0.0)
    small_total = eng._rec_total
    version = pool.version

    def hook():  # what _drain()'s fallback reset does: new scenes for (here: all) envs, same pool
        eng.set_scene(list(range(N)), torch.full((N, 3), big), off)

    a = torch.randn(N, 2, device="cuda", requires_grad=True)
    obs, reward, done, fs, loss, out = eng.step(a, with_reserve=True, pre_launch=hook)
    reward.sum().backward()
    eng.check_status()  # raises on OCC_STATUS_REC_OVERFLOW
    assert pool.version == version and eng._rec_total > small_total
    assert torch.isfinite(obs).all() and torch.isfinite(a.grad).all()
    # the big meshes really were rendered: compare with a fresh engine that had them from the start
    eng2 = OcclusionEngine(pool, N, S)
    eng2.set_scene(list(range(N)), torch.full((N, 3), big), off)
    eng2.elevation.copy_(eng.elevation - 0.0)
    assert float(eng.alphas.sum()) > 0
    obs2 = eng2.render_hard()  # camera_position = 0 there: only checks the pool/record path runs at this size
    assert torch.isfinite(obs2).all()


@pytest.mark.parametrize("mesh,S", [("mixed", 128), ("teapot", 64), ("synthetic", 256)])
def test_work_item_order_is_a_cost_sorted_permutation_of_all_tiles(mesh, S):
    """OccWorkspace.order after a render (occ_setup_kernel + occ_order_kernel): the item list holds every tile of every
    object's rect exactly once, queue by queue (env % 8), cost classes never increasing inside a queue - and the class
    of a tile really bounds the face records whose pixel bbox touches it, which is what lets occ_raster2_kernel skip
    the log for tiles of at most K faces."""
    from tests.parity_utils import make_case
    from occlusionenv_amd.engine import OcclusionEngine

    N = 40
    case = make_case(N, 77, mesh, az_range=2.0)
    eng = OcclusionEngine(case["pool"], N, S)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    eng.reset_render(None, 4.0, case["az"], 0.0)
    eng.check_status()
    torch.cuda.synchronize()
    order = eng._ws_tensors["order"].cpu().numpy().view(np.uint32)
    nrec = eng._ws_tensors["nrec"].cpu().numpy()[: N * 3]
    rect = eng._ws_tensors["objrect"].cpu().numpy()[: N * 12].reshape(N * 3, 4)
    G = S // 8
    T = G * G
    tiles_w = 512 + N * 3 * 32
    items_w = tiles_w + N * 3 * T
    qoff = order[:9].astype(np.int64)
    expect = {}  # (eo, local) -> (tx, ty)
    for eo in range(N * 3):
        x0, y0, x1, y1 = rect[eo]
        if nrec[eo] <= 0 or x1 < x0 or y1 < y0:
            continue
        tx0, ty0, tw, th = x0 >> 1, y0 >> 1, (x1 >> 1) - (x0 >> 1) + 1, (y1 >> 1) - (y0 >> 1) + 1
        for local in range(tw * th):
            expect[(eo, local)] = (tx0 + local % tw, ty0 + local // tw)
    assert qoff[0] == 0 and np.all(np.diff(qoff) >= 0) and qoff[8] == len(expect)
    items = order[items_w: items_w + 2 * qoff[8]].reshape(-1, 2)
    eos, local, cls = items[:, 0].astype(np.int64), (items[:, 1] & 0xFFFFFF).astype(np.int64), (items[:, 1] >> 24).astype(np.int64)
    assert sorted(zip(eos.tolist(), local.tolist())) == sorted(expect)  # a permutation: nothing lost, nothing twice
    for q in range(8):
        sl = slice(qoff[q], qoff[q + 1])
        assert np.all((eos[sl] // 3) % 8 == q)
        assert np.all(np.diff(cls[sl]) <= 0)  # heaviest class first
    assert cls.max() > cls.min()
    # the class bounds the faces from above: count, per tile, the records whose pixel bbox touches it
    rec_off = eng._rec_tensors["rec_off"].cpu().numpy().view(np.int64)
    bbox = eng._rec_tensors["scan"].cpu().numpy().view(np.uint32).reshape(-1, 4)  # rows in face order: (pixel bbox, depth key, index)

    def bound(c):  # occ_common.hpp: ord_class_bound
        if c <= 0:
            return 1
        if c >= 31:
            return 1 << 40
        fl, half = (c - 1) >> 1, (c - 1) & 1
        return 2 if fl == 0 else ((2 << fl) if half else (3 << (fl - 1)))

    for k in np.random.default_rng(0).choice(len(items), size=min(400, len(items)), replace=False).tolist():
        eo, (tx, ty) = int(eos[k]), expect[(int(eos[k]), int(local[k]))]
        bb = bbox[rec_off[eo]: rec_off[eo] + nrec[eo]]
        xl, yl, xh, yh = bb[:, 0] & 0xFFFF, bb[:, 0] >> 16, bb[:, 1] & 0xFFFF, (bb[:, 1] >> 16) & 0x0FFF  # (corner-cut bits above)
        touching = int(np.count_nonzero((xl <= tx * 8 + 7) & (xh >= tx * 8) & (yl <= ty * 8 + 7) & (yh >= ty * 8)))
        assert touching < bound(int(cls[k])), (eo, tx, ty, touching, int(cls[k]))


def test_env_on_a_shapenetcore_directory(tmp_path):
    """trainRL.py:66-75 with the directory reader instead of PyTorch3D's ShapeNetCore: OcclusionEnv(dataset) resets,
    steps and renders textured models read from <synset>/<model>/models/model_normalized.obj (+ .mtl + image)."""
    from occlusionenv_amd.shapenet import ShapeNetCoreDir
    from tests.test_shapenet_dir import _write_model
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    root = str(tmp_path / "shapenetcore")
    for k, syn in enumerate(["02691156", "03001627", "04379243"]):
        for m in range(2):
            _write_model(root, syn, f"m{k}{m}", scale=0.25 + 0.05 * m)
    ds = ShapeNetCoreDir(root, version=2)
    assert len(ds) == 6 and len(ds.synset_dict) == 3
    env = OcclusionEnv(ds, img_size=64)
    obs = env.reset()
    assert obs.shape == (1, 4, 64, 64) and torch.isfinite(obs).all()
    a = torch.nn.Parameter(torch.tensor([0.3, -0.2], device="cuda"))
    obs, reward, done, info = env.step(a)
    reward.backward()
    assert torch.isfinite(a.grad).all() and torch.isfinite(obs).all()
    fg = obs[0, 3] > 0
    assert bool(fg.any())
    rgb = obs[0, :3][:, fg]
    # the cubes carry a red Kd material, an image-textured face and grey faces: the observation is not all white/grey
    assert float((rgb.max(0).values - rgb.min(0).values).max()) > 0.2
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=64) for _ in range(4)])
    o = venv.reset()
    assert o.shape == (4, 1, 4, 64, 64)
    o2, r2, d2, _ = venv.step(torch.zeros(4, 2, device="cuda"))
    assert o2.shape == (4, 4, 64, 64) and torch.isfinite(r2).all()


@pytest.mark.parametrize("S", [24, 64, 72, 128, 256])
def test_pool8_equals_adaptive_avg_pool(S):
    """rollout.pooled_features on the GPU is the library's one-pass kernel (occ_pool8): the 4 x 8 x 8 cell means in the
    order of F.adaptive_avg_pool2d(obs, 8).reshape(n, 256); cells of 3, 8, 9, 16 and 32 pixels (16-byte and scalar loads)."""
    import torch.nn.functional as F

    from occlusionenv_amd import rollout

    g = torch.Generator(device="cuda").manual_seed(S)
    obs = torch.rand(5, 4, S, S, device="cuda", generator=g) * 3.0 - 1.0
    want = F.adaptive_avg_pool2d(obs.double(), 8).reshape(5, 256)
    got = rollout.pooled_features(obs)
    assert got.shape == (5, 256) and got.dtype == torch.float32
    assert float((got.double() - want).abs().max()) < 1e-6
    # a non-contiguous view (channels-last memory) goes through the same kernel
    got2 = rollout.pooled_features(obs.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))
    assert torch.equal(got2, got)


def test_fused_ppo_update_equals_the_torch_one():
    """occ_ppo_update (csrc/occ_ppo.hpp: forward, clipped-surrogate loss, backward and Adam of one epoch in ONE launch)
    against the torch implementation of the same epochs (autograd + torch.optim.Adam, PPO.py:196-217), over two updates,
    the second with a decayed action std (PPO.py:136-149).  Five epochs per update: every parameter to 1e-6 (measured
    6e-8).  Eighty (trainRL.py): the critic to 1e-5; the ACTOR's trajectory amplifies rounding differences through the
    clip boundary (a sample crossing it switches its whole gradient on or off: 4e-9 after one epoch, 3e-6 after 20,
    5e-4 after 80, scripts/dbg/ppo_fused_diff.py) - held to 5 % of the distance the update moved it."""
    from occlusionenv_amd import ppo, rollout

    g = torch.Generator(device="cuda").manual_seed(7)
    recs = []
    for _u in range(2):
        r = torch.randn(20, 128, rollout.RECORD_FLOATS, device="cuda", generator=g)
        r[..., :256] = r[..., :256].abs() * 0.5                 # pooled features of an image: non-negative
        r[..., 256:258] *= 0.6                                  # actions
        r[..., 258] = -1.5 + 0.3 * r[..., 258]                  # old log-probabilities
        r[..., 260] = (torch.rand(20, 128, device="cuda", generator=g) < 0.05).float()
        recs.append(r)
    init = [p.detach().clone() for p in ppo.BatchedPPO(device="cuda", seed=3, fused=False).policy.parameters()]
    for K in (5, 80):
        out = []
        for fused in (True, False):
            agent = ppo.BatchedPPO(device="cuda", seed=3, K_epochs=K, graph_epochs=False, fused=fused)
            assert agent.fused == fused
            stats = []
            for u, r in enumerate(recs):
                if u == 1:
                    agent.decay_action_std(0.05, 0.1)
                for t in range(r.shape[0]):
                    agent.store(r[t])
                stats.append(agent.update())
            out.append(([p.detach().clone() for p in agent.policy.parameters()], stats,
                        [p.detach().clone() for p in agent.policy_old.parameters()]))
        names = [n for n, _ in ppo.BatchedPPO(device="cuda", seed=3, fused=False).policy.named_parameters()]
        for n, a, b, p0 in zip(names, out[0][0], out[1][0], init):
            moved, diff = float((b - p0).abs().max()), float((a - b).abs().max())
            assert moved > 1e-4                                    # the Adam steps did move the heads
            bound = 1e-6 if K == 5 else (1e-5 if n.startswith("value_head") else 5e-2 * moved)
            assert diff < bound, (K, n, diff, moved)
        for a, b in zip(out[0][2], out[0][0]):
            assert torch.equal(a, b)                            # policy_old follows the updated policy
        for sa, sb in zip(out[0][1], out[1][1]):
            assert sa["samples"] == sb["samples"] == 20 * 128
            for k in ("loss_first", "loss_last", "value_loss_first", "value_loss_last"):
                assert abs(sa[k] - sb[k]) < (1e-5 if K == 5 else 1e-3) * max(1.0, abs(sb[k])), (K, k, sa[k], sb[k])


def test_graphed_ppo_epochs_equal_the_eager_ones():
    """BatchedPPO.update on the GPU replays ONE captured HIP graph for epochs 4..K: the heads it leaves behind must be
    those of the same epochs launched one by one, over two updates (the second reuses the graph with new data)."""
    from occlusionenv_amd import ppo, rollout

    g = torch.Generator(device="cuda").manual_seed(5)
    recs = []
    for _u in range(2):
        r = torch.randn(12, 96, rollout.RECORD_FLOATS, device="cuda", generator=g)
        r[..., 258] = -2.0 + 0.1 * r[..., 258]          # old log-probabilities
        r[..., 260] = (torch.rand(12, 96, device="cuda", generator=g) < 0.1).float()
        recs.append(r)
    heads = []
    for graph in (True, False):
        agent = ppo.BatchedPPO(device="cuda", seed=3, K_epochs=20, graph_epochs=graph, fused=False)
        stats = []
        for r in recs:
            for t in range(r.shape[0]):
                agent.store(r[t])
            stats.append(agent.update())
        assert agent.graph_epochs == graph  # the capture did not fall back
        heads.append(([p.detach().clone() for p in agent.policy.parameters()], stats))
    for a, b in zip(heads[0][0], heads[1][0]):
        assert torch.allclose(a, b, rtol=0, atol=1e-6), float((a - b).abs().max())
    for sa, sb in zip(heads[0][1], heads[1][1]):
        assert sa["samples"] == sb["samples"] == 12 * 96
        assert abs(sa["loss_last"] - sb["loss_last"]) < 1e-5 and abs(sa["value_loss_last"] - sb["value_loss_last"]) < 1e-5
