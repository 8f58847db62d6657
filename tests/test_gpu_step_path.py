"""GPU tests of round 4's step path: the fused launch sequence (occ_step: camera in the prologue, reward bookkeeping in
the reduction), the opt-in output ring with region tracking, and the episode time limit of the reference's training loop
(/root/reference/trainRL.py:22,189-229)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ds():
    from occlusionenv_amd.meshes import SyntheticShapeNet

    return SyntheticShapeNet(n_models=8, seed=1234)


def _seed(k):
    from occlusionenv_amd import environment

    environment.seed_scene_rng(k)
    np.random.seed(k)
    torch.manual_seed(k)


def _make_venv(ds, N, S, ring=0, seed=77, recycle=None):
    """``recycle``: None = the default (recycled outputs, SimpleVecEnv docstring), 0 = plain allocations every step."""
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    _seed(seed)
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    if recycle is not None:
        venv.use_recycled_outputs(recycle)
    if ring:
        venv.use_output_ring(ring)
    az0 = (torch.rand(N, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * 0.6
    venv._reset_envs(list(range(N)), az0)
    venv._warm_reserve()
    return venv


def _rollout(venv, steps, push=None):
    """Random-action rollout; ``push[t]`` = envs whose objects are pulled apart before step t (they finish).  Returns a
    snapshot (clones: the ring overwrites its sets) of everything a caller can see, per step."""
    eng, N = venv.engine, venv.num_envs
    gen = torch.Generator(device="cuda").manual_seed(5)
    snaps = []
    for t in range(steps):
        for i in (push or {}).get(t, []):
            venv._drain()
            off = eng.scene_offset[i].clone()
            off[1, 0], off[2, 0] = 50.0, -50.0
            eng.scene_offset[i] = off
        a = torch.randn(N, 2, device="cuda", generator=gen, requires_grad=True)
        obs, rewards, dones, infos = venv.step(a)
        rewards.sum().backward()
        fs = torch.cat([infos[i]["full_state"] for i in range(N)])
        term = {i: infos[i]["terminal_observation"].clone() for i in range(N) if "terminal_observation" in infos[i]}
        snaps.append(dict(obs=obs.clone(), rewards=rewards.detach().clone(), dones=dones.clone(), grad=a.grad.clone(),
                          fs=fs.clone(), alphas=eng.alphas.clone(), term=term,
                          pos=torch.stack([infos[i]["position"] for i in range(N)]).clone(),
                          images=torch.cat([venv.envs[i].image for i in range(N)]).clone()))
    venv._drain()
    eng.check_status()
    return snaps


def test_output_ring_never_changes_a_bit(ds):
    """The same seeded rollout with freshly allocated outputs, with recycled outputs (the default) and with a ring of 3
    persistent output sets (the combine kernel then writes only the pixel blocks that meet this step's object rects or
    what the set held before): every observation, occlusion image, alpha plane, reward, done flag, gradient, terminal
    observation and reset image is bit-identical - through auto-resets from the reserve (full-frame commits into a set)
    and through the synchronous fallback (22 envs finish at once against 16 reserve slots)."""
    N, S, T = 64, 64, 30
    push = {4: [3], 9: [10, 11], 15: list(range(20, 42)), 22: [5]}
    ref = _rollout(_make_venv(ds, N, S, ring=0, recycle=0), T, push)
    for mode in (dict(ring=3), dict(recycle=None), dict(recycle=2)):
        got = _rollout(_make_venv(ds, N, S, **mode), T, push)
        n_done = 0
        for t, (a, b) in enumerate(zip(ref, got)):
            for k in ("obs", "rewards", "dones", "grad", "fs", "alphas", "pos", "images"):
                assert torch.equal(a[k], b[k]), (mode, t, k)
            assert set(a["term"]) == set(b["term"])
            for i in a["term"]:
                assert torch.equal(a["term"][i], b["term"][i]), (mode, t, i)
            n_done += int(a["dones"].sum())
        assert n_done >= 26
    # the ring really is a ring: the set of step t is the set of step t + 3
    venv = _make_venv(ds, 16, 64, ring=3)
    ptrs = [venv.step(torch.zeros(16, 2, device="cuda"))[0].data_ptr() for _ in range(6)]
    assert ptrs[0] == ptrs[3] and ptrs[1] == ptrs[4] and len(set(ptrs)) == 3
    with pytest.raises(ValueError):
        venv.use_output_ring(1)


def test_recycled_outputs_never_touch_a_tensor_somebody_still_holds(ds):
    """Recycled outputs (the default): a step's tensors come from a pool of persistent sets, and a set is reused only once
    no view of it is alive.  (a) a rollout loop that rebinds obs / infos runs on two sets; (b) an observation and an
    occlusion image the caller KEEPS are never overwritten - their values stay what a fresh-tensor run returned for that
    step - while later steps move on to other sets and, at the pool's cap, to plain allocations."""
    N, S = 32, 64
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = [torch.randn(N, 2, device="cuda", generator=gen) for _ in range(13)]
    # the reference values of every step: the same seeded env with plain allocations
    ref_env = _make_venv(ds, N, S, recycle=0)
    ref = []
    for t in range(12):
        o, _, _, inf = ref_env.step(acts[t])
        ref.append((o.clone(), inf[0]["full_state"].clone()))
    del ref_env, o, inf
    venv = _make_venv(ds, N, S)
    eng = venv.engine
    assert eng.output_recycle >= 2 and not eng.output_ring
    ptrs = []
    for t in range(6):  # (a) nothing kept: obs / infos of step t die when step t + 1 returns
        obs, _, _, infos = venv.step(acts[t])
        assert torch.equal(obs, ref[t][0]), t
        ptrs.append(obs.data_ptr())
    del obs, infos
    assert len(set(ptrs)) <= 3 and len(eng._ring) <= 3, (ptrs, len(eng._ring))
    kept = []
    for t in range(6, 12):  # (b) every step's obs and one full_state view are KEPT
        obs, _, _, infos = venv.step(acts[t])
        kept.append((obs, infos[0]["full_state"]))
    torch.cuda.synchronize()
    assert len({o.data_ptr() for o, _ in kept}) == 6  # six live observations: six different buffers
    assert len(eng._ring) == eng.output_recycle       # the pool stopped growing at its cap ...
    for t, (o, f) in zip(range(6, 12), kept):          # ... and nothing that is held was overwritten
        assert torch.equal(o, ref[t][0]) and torch.equal(f, ref[t][1]), t
    del kept, obs, infos, o, f
    venv.step(acts[12])
    assert len(eng._ring) == eng.output_recycle  # released sets are found again: no growth, no leak


def test_ring_sets_hold_background_outside_their_rects(ds):
    """The invariant region tracking rests on (OccRenderOut.rect_prev): outside rect[e] a set's obs / full_state rows are
    background, and the alphas state is zero outside its own rects - checked on the device after a rollout with resets."""
    N, S = 32, 64
    venv = _make_venv(ds, N, S, ring=2)
    _rollout(venv, 9, {2: [1, 2, 3], 6: [7]})
    eng = venv.engine
    ys, xs = torch.meshgrid(torch.arange(S, device="cuda"), torch.arange(S, device="cuda"), indexing="ij")

    def outside(rect):
        r = rect.long()
        return ~((xs[None] >= r[:, 0, None, None]) & (xs[None] <= r[:, 2, None, None]) & (ys[None] >= r[:, 1, None, None]) &
                 (ys[None] <= r[:, 3, None, None]))

    for slot in eng._ring:
        out = outside(slot["rect"][slot["cur"]])  # (NT,S,S)
        obs, fs = slot["obs"], slot["fs"]
        assert bool((obs[:, :3].permute(0, 2, 3, 1)[out] == 1.0).all()) and bool((obs[:, 3][out] == -1.0).all())
        assert bool((fs[out] == torch.tensor([3.0, 3.0, 3.0, 0.0], device="cuda")).all())
        assert int((~out).sum()) < out.numel()  # and the rects are not simply everything
    outa = outside(eng._arect[eng._arect_cur])
    assert bool((eng._alphas_all.permute(0, 2, 3, 1)[outa] == 0.0).all())


def test_fused_step_equals_the_separate_calls(ds):
    """occ_step (camera in the prologue launch, reward rule in the reduction launch) against occ_camera + occ_render +
    occ_step_finish on the same state: cameras, outputs, rewards, done flags and action gradients bit for bit."""
    from occlusionenv_amd import _native as nat
    from occlusionenv_amd.engine import OcclusionEngine, _p
    from tests.parity_utils import make_case

    N, S = 24, 64
    case = make_case(N, 31, "synthetic")
    eng = OcclusionEngine(case["pool"], N, S)
    eng.set_scene(list(range(N)), case["mesh_ids"], case["offsets"])
    eng.reset_render(None, 4.0, case["az"], 0.0)
    st0 = dict(el=eng.elevation.clone(), az=eng.azimuth.clone(), fr=eng.full_reward.clone())
    a = case["actions"].cuda().requires_grad_(True)
    obs, reward, done, fs, loss = eng.step(a)  # the fused path
    reward.sum().backward()
    cam_f, el_f, az_f, pos_f, fr_f = eng.cam.clone(), eng.elevation.clone(), eng.azimuth.clone(), eng.camera_position.clone(), eng.full_reward.clone()
    # the same step from the separate entry points
    lib, stream = eng.lib, eng._stream()
    el, az, fr = st0["el"].clone(), st0["az"].clone(), st0["fr"].clone()
    cam = torch.empty(N, nat.CAM_STRIDE, device="cuda")
    pos = torch.empty(N, 3, device="cuda")
    ad = a.detach().contiguous()
    nat.check(lib.occ_camera(nat.CAM_STEP, _p(ad), _p(el), _p(az), _p(eng.radius), _p(cam), _p(pos), N, stream), "occ_camera")
    ro = nat.OccRenderOut()
    o2, f2, l2, g2 = torch.empty_like(obs), torch.empty_like(fs), torch.empty_like(loss), torch.empty(N, 2, device="cuda")
    al2 = torch.empty(N, 3, S, S, device="cuda")
    ro.obs, ro.full_state, ro.loss, ro.alphas, ro.grad_elaz = o2.data_ptr(), f2.data_ptr(), l2.data_ptr(), al2.data_ptr(), g2.data_ptr()
    ws = eng._ensure_workspace()
    sc = eng._scene_struct(N, eng.scene_mesh, eng.scene_offset)
    flags = nat.RENDER_SOFT | nat.RENDER_HARD | nat.RENDER_GRAD
    nat.check(lib.occ_render(C.byref(sc), _p(cam), C.byref(ws), C.byref(ro), flags, eng.K, stream), "occ_render")
    rw, dn, ga = torch.empty(N, device="cuda"), torch.empty(N, dtype=torch.uint8, device="cuda"), torch.empty(N, 2, device="cuda")
    nat.check(lib.occ_step_finish(_p(l2), _p(g2), _p(cam), _p(fr), _p(eng.object_mass), _p(rw), _p(dn), _p(ga), N, stream), "finish")
    for name, x, y in (("cam", cam_f, cam), ("el", el_f, el), ("az", az_f, az), ("pos", pos_f, pos), ("obs", obs, o2), ("fs", fs, f2),
                       ("loss", loss, l2), ("alphas", eng.alphas, al2), ("reward", reward.detach(), rw), ("done", done, dn.bool()),
                       ("grad", a.grad, ga), ("full_reward", fr_f, fr)):
        assert torch.equal(x, y), name
    # argument checks of the new pieces: a finish block without the soft pass, rect arrays that are one and the same
    fin = nat.OccStepFinish()
    fin.full_reward, fin.object_mass, fin.reward, fin.done, fin.n_step = fr.data_ptr(), eng.object_mass.data_ptr(), rw.data_ptr(), dn.data_ptr(), N
    ro.finish = C.pointer(fin)
    assert lib.occ_render(C.byref(sc), _p(cam), C.byref(ws), C.byref(ro), nat.RENDER_HARD, eng.K, stream) == 1
    ro.finish = None
    rect = torch.zeros(N, 4, dtype=torch.int32, device="cuda")
    ro.rect_prev = ro.rect_next = rect.data_ptr()
    assert lib.occ_render(C.byref(sc), _p(cam), C.byref(ws), C.byref(ro), flags, eng.K, stream) == 1
    torch.cuda.synchronize()


def test_time_limit_resets_without_done(ds):
    """trainRL.py:22,191-229: an episode ends in env.reset() after max_ep_len steps, done or not, and is_terminal stays
    False.  16 envs that cannot finish (zero actions) all reach max_ep_len = 4 together: four are reset from the reserve
    on the device, twelve by the synchronous fallback; every reset observation equals a reset render of the env's new
    scene, ``dones`` stays False, infos carry TimeLimit.truncated + terminal_observation, the counters restart."""
    N, S = 16, 64
    venv = _make_venv(ds, N, S)
    venv.max_ep_len = 4
    eng = venv.engine
    zero = torch.zeros(N, 2, device="cuda")
    scenes0 = [venv.envs[i]._scene for i in range(N)]
    for t in range(3):
        obs, rewards, dones, infos = venv.step(zero)
        assert not bool(dones.any()) and all("TimeLimit.truncated" not in infos[i] for i in range(N))
        assert eng.age.tolist() == [t + 1] * N
    last = obs.clone()
    obs, rewards, dones, infos = venv.step(zero)
    assert not bool(dones.any()), "a time-limit reset is not a terminal (PPO.py:181-183 keeps the return flowing)"
    for i in range(N):
        assert infos[i]["TimeLimit.truncated"] is True
        assert torch.equal(infos[i]["terminal_observation"][0], last[i])  # zero actions: the final view is the last one
        assert venv.envs[i]._scene is not scenes0[i]
        ids, offs = venv.envs[i]._scene
        ref = eng.evaluate_scenes([ids], [offs], 4.0, 0.0, 0.0)
        assert torch.equal(ref["obs"][0], obs[i]), i
        assert float(eng.full_reward[i]) == float(ref["loss"][0]) and float(eng.azimuth[i]) == 0.0
        assert torch.equal(venv.envs[i].image[0], ref["full_state"][0])
    assert eng.age.tolist() == [0] * N
    assert float(rewards.max()) < 0.0  # no +5 bonus: nobody finished
    venv.step(zero)
    assert eng.age.tolist() == [1] * N
    # staggered ages: about N / max_ep_len envs expire per step instead of all of them every max_ep_len steps
    venv.max_ep_len = 8
    venv.stagger_ages(seed=3)
    ages = eng.age.cpu().numpy().copy()
    assert ages.min() >= 0 and ages.max() < 8 and len(set(ages.tolist())) > 3
    expect = int((ages == 7).sum())
    obs, rewards, dones, infos = venv.step(zero)
    assert sum(1 for i in range(N) if "TimeLimit.truncated" in infos[i]) == expect


def test_time_limit_on_the_host_driven_path(ds):
    """Fewer than 16 envs have no reserve (flags hand-off, synchronous reset): the same rule, counted on the host."""
    from environment import OcclusionEnv
    from SubProcVecEnv import SimpleVecEnv

    _seed(5)
    N, S = 4, 64
    venv = SimpleVecEnv([lambda: OcclusionEnv(ds, img_size=S) for _ in range(N)])
    venv._reset_envs(list(range(N)), torch.zeros(N))
    assert venv.engine.R == 0
    venv.max_ep_len = 3
    zero = torch.zeros(N, 2, device="cuda")
    for t in range(2):
        _, _, dones, infos = venv.step(zero)
        assert not bool(dones.any()) and "terminal_observation" not in infos[0]
    scenes0 = [venv.envs[i]._scene for i in range(N)]
    obs, _, dones, infos = venv.step(zero)
    assert not bool(dones.any())
    for i in range(N):
        assert infos[i]["TimeLimit.truncated"] is True and "terminal_observation" in infos[i]
        assert venv.envs[i]._scene is not scenes0[i]
    assert venv._age_host.tolist() == [0] * N


def test_encoder_hook_on_the_gpu_rollout(ds):
    """BatchedPPO(encoder=...) in the batched rollout (PPO.py:47,155-157: the buffer stores the frozen encoder's pooled
    features): a small conv stub with the shape contract of the H1 fixture (obs (N,4,S,S) -> (N,256), captured from the
    reference's FullNetwork in tests/golden/vecenv_golden.npz) feeds select_action, the records and the fused update."""
    import os

    from occlusionenv_amd import ppo, rollout

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vecenv_golden.npz"))
    shapes = eval(str(gold["h1_shapes"]))  # noqa: S307 - [obs, pooled_features, ...] per case, from the reference
    assert [s[0][1:] for s in shapes] == [[4, 64, 64], [4, 128, 128], [4, 128, 128]] and shapes[2][1] == [8, 256]
    N, S, T = 16, 64, 6
    venv = _make_venv(ds, N, S)
    torch.manual_seed(0)
    enc = torch.nn.Sequential(torch.nn.Conv2d(4, 16, 3, stride=2, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(4),
                              torch.nn.Flatten()).cuda()
    seen = []

    def encoder(obs):
        seen.append((tuple(obs.shape), torch.is_grad_enabled()))
        return enc(obs)

    agent = ppo.BatchedPPO(device="cuda", seed=0, K_epochs=4, encoder=encoder)
    recs = []
    orig = agent.store
    agent.store = lambda r: (recs.append(r.clone()), orig(r))[1]
    stats = ppo.train_rollouts(venv, agent, n_updates=1, T=T, with_action_grad=True, max_ep_len=4)
    assert len(seen) == T and all(s == ((N, 4, S, S), False) for s in seen)
    assert len(recs) == T and recs[0].shape == (N, rollout.RECORD_FLOATS)
    assert stats[0]["samples"] == T * N and np.isfinite(stats[0]["loss_last"])
    # the stored features are the encoder's, not the 8x8 pooling stand-in's
    obs0 = venv.reset()[:, 0]
    f_enc = enc(obs0).detach()
    feats, _, _ = agent.select_action(obs0)
    assert torch.allclose(feats, f_enc, atol=1e-6) and not torch.allclose(feats, rollout.pooled_features(obs0), atol=1e-3)
    assert venv.max_ep_len == 4  # the time limit train_rollouts switched on


def test_ring_set_waits_for_its_side_stream_consumer(ds):
    """With the output ring a step's ``obs`` is overwritten k steps later.  A consumer on another stream (bench.py's
    RecordExchange packs step t's records while step t + 1 renders) announces itself through ``obs_consumer_event``;
    the env keeps that event with the output set and makes the step that reuses the set wait for it.  Here the consumer
    is artificially slow (it sleeps on the device before it reads): what it reads must still be step t's observation."""
    N, S, k = 16, 64, 2
    venv = _make_venv(ds, N, S, ring=k)
    ref = _make_venv(ds, N, S, ring=0)
    side = torch.cuda.Stream()
    gen = torch.Generator(device="cuda").manual_seed(9)
    copies, expect = [], []
    alive = torch.ones(N, dtype=torch.bool, device="cuda")  # (an env that finishes draws its next scene from the RNG the two
    for t in range(6):                                       #  envs share: only envs that never finished are compared)
        a = torch.randn(N, 2, device="cuda", generator=gen)
        obs, _, d1, _ = venv.step(a)
        o2, _, d2, _ = ref.step(a)
        alive &= ~(d1 | d2)
        expect.append((o2.clone(), alive.clone()))
        produced = torch.cuda.Event()
        produced.record()
        with torch.cuda.stream(side):
            side.wait_event(produced)
            torch.cuda._sleep(20_000_000)  # ~10 ms: far longer than a step - the ring set comes round again before this ends
            copies.append(obs.clone())
            ev = torch.cuda.Event()
            ev.record(side)
        obs.record_stream(side)
        venv.obs_consumer_event = ev
    torch.cuda.synchronize()
    assert int(alive.sum()) >= N // 2
    for t, (c, (e, ok)) in enumerate(zip(copies, expect)):
        assert torch.equal(c[ok], e[ok]), t
        assert t == 0 or not torch.equal(c[ok], expect[t - 1][0][ok])  # (the observation does change from step to step)
