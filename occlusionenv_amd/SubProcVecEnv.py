"""Drop-in ``SubProcVecEnv`` module: ``SimpleVecEnv(env_fns)`` with the reference's signature and return
contract (/root/reference/SubProcVecEnv.py:189-285), but BATCHED: instead of the sequential
``for env_idx in range(num_envs): envs[env_idx].step(...)`` loop (SubProcVecEnv.py:209-218), the N
environments' meshes and cameras are packed into one launch sequence on one GPU
(``OcclusionEngine.step``).

Return contract kept:
  * ``step``  -> ``obs (N,4,S,S)``, ``rewards (N,)`` autograd-attached to ``actions`` (so that
    ``rewards.sum().backward()`` fills ``actions.grad (N,2)``, train_predict.py:51-52), ``dones (N,) bool``,
    ``infos`` = sequence of N dicts with ``full_state``, ``position``, ``full_reward`` and, for finished envs,
    ``terminal_observation``; finished envs are reset (new random scene, default azimuth 0).
  * ``reset`` -> ``(N,1,4,S,S)``: the reference stacks each env's ``(1,4,S,S)`` observation
    (SubProcVecEnv.py:230-235), azimuth ~ U(-40, 40) *radians* per env.

Deviation: scene rejection sampling in ``reset`` runs in rounds over the whole batch (one batched render per
round) rather than env by env, so the order of ``np.random.randn()`` draws differs from the sequential loop
when an env has to re-draw its scene.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import numpy as np
import torch

from . import _native as nat
from .baseVecEnv import VecEnv
from .engine import OcclusionEngine
from .environment import shared_pool

RECYCLE_SETS = 4  # persistent output sets of the recycled-outputs pool (engine.OcclusionEngine.output_recycle)


def _structure(space):
    """How a gym-style space is laid out, told by duck-typing (gym itself may be absent): a mapping of named
    sub-spaces ("dict"), a sequence of them ("tuple"), or a single array space (None)."""
    sub = getattr(space, "spaces", None)
    if sub is None:
        return None, {None: space}
    if hasattr(sub, "items"):
        return "dict", OrderedDict(sub.items())
    return "tuple", OrderedDict(enumerate(sub))


def obs_space_info(obs_space):
    """``(keys, shapes, dtypes)`` of an observation space, one entry per array it holds; a plain space has the single
    key ``None`` (interface of /root/reference/SubProcVecEnv.py:43-70)."""
    _, parts = _structure(obs_space)
    keys = list(parts)
    return keys, {k: parts[k].shape for k in keys}, {k: parts[k].dtype for k in keys}


def dict_to_obs(space, obs_dict):
    """Keyed buffers -> the container the space's structure implies: the dict itself, a tuple in sub-space order, or
    the lone array (interface of /root/reference/SubProcVecEnv.py:21-40)."""
    kind, parts = _structure(space)
    if set(obs_dict) != set(parts):
        raise AssertionError("observation keys %r do not match the observation space %r" % (sorted(map(str, obs_dict)), sorted(map(str, parts))))
    if kind == "dict":
        return obs_dict
    if kind == "tuple":
        return tuple(obs_dict[k] for k in parts)
    return obs_dict[None]


def copy_obs_dict(obs):
    """New OrderedDict over the same arrays (interface of /root/reference/SubProcVecEnv.py:11-18)."""
    if not isinstance(obs, OrderedDict):
        raise AssertionError("observations must be an OrderedDict, got %s" % type(obs).__name__)
    return OrderedDict(obs.items())


class _LazyInfos(Sequence):
    """``list[dict]`` look-alike whose dicts are built on first access (1024 dicts of tensor views per step
    would cost more host time than the render)."""

    def __init__(self, position, full_state, loss, resolve=None):
        # position: (N,3) camera positions AS OF THIS STEP (a snapshot: the auto-reset zeroes the rows of finished
        # envs and the next step overwrites all of them; the reference's reset() rebinds camera_position instead,
        # environment.py:302, so its info dict keeps the terminal step's tensor)
        self._pos, self._fs, self._loss = position, full_state, loss
        self._extra = {}
        self._made = {}
        self._resolve = resolve  # called before the first read: lets the env finish its deferred bookkeeping

    def __len__(self):
        return self._fs.shape[0]

    def set(self, i, key, value):
        self._extra.setdefault(i, {})[key] = value
        if i in self._made:
            self._made[i][key] = value

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if self._resolve is not None:
            r, self._resolve = self._resolve, None
            r()
        if i not in self._made:
            d = {"full_state": self._fs[i:i + 1], "position": self._pos[i],
                 "full_reward": self._loss[i]}
            d.update(self._extra.get(i, {}))
            self._made[i] = d
        return self._made[i]


class SimpleVecEnv(VecEnv):
    def __init__(self, env_fns):
        """``env_fns`` as the reference (SubProcVecEnv.py:191); the signature is the reference's.  Two opt-in extras are
        attributes, both off by default:

        ``venv.max_ep_len = 50``: the episode time limit of the reference's training loop (trainRL.py:22,191-229: an
        episode ends in ``env.reset()`` after max_ep_len steps whether done or not, and ``is_terminal`` stays False):
        an env that has stepped max_ep_len times since its last reset is reset like a finished one, with ``dones`` left
        False and ``infos[i]["TimeLimit.truncated"] = True`` beside ``terminal_observation``.

        ``venv.use_output_ring(k)`` (k >= 2): the engine's output ring (engine.OcclusionEngine): ``obs`` /
        ``infos[i]["full_state"]`` of a step are views of one of k persistent output sets and are overwritten k steps later.

        On by default (batched path with a reserve): RECYCLED OUTPUTS - the tensors of a step come from a small pool of
        persistent output sets, and a set is reused only once no view of it is alive anywhere (the caller dropped that
        step's ``obs`` / ``infos``).  What a caller holds is never overwritten, exactly as with the reference's fresh
        tensors per step (SubProcVecEnv.py:215-219); ``venv.use_recycled_outputs(0)`` switches to plain allocations."""
        self.envs = [fn() for fn in env_fns]
        env = self.envs[0]
        VecEnv.__init__(self, len(env_fns), env.observation_space, env.action_space)
        obs_space = env.observation_space
        self.keys, shapes, dtypes = obs_space_info(obs_space)
        self.actions = None
        dev = torch.device(f"cuda:{torch.cuda.current_device()}") if torch.cuda.is_available() else env.device
        # reserve slots: speculative auto-reset scenes rendered inside every batched step (see engine.py)
        same_data = all(e.shapenet_dataset is env.shapenet_dataset for e in self.envs)
        # ~0.8 % of the envs finish per step and ~58 % of the candidates pass; with the reference's episode time limit
        # (max_ep_len = 50, trainRL.py:22) another 2 % expire per step, and a slot takes three to four steps from taken
        # to READY again.  Only slots under test are rendered, so generous N/4 slots cost next to nothing and keep the
        # reserve from running dry (with N/8 the time-limited PPO rollout fell back to reading the report at the end of
        # most steps: 3.31 instead of 2.87 ms per step at 256 envs)
        reserve = min(512, self.num_envs // 4) if (same_data and self.num_envs >= 16) else 0
        self.engine = OcclusionEngine(shared_pool(dev), self.num_envs, env.img_size, device=dev, reserve=reserve,
                                      output_recycle=RECYCLE_SETS if reserve else 0)
        self._age_host = np.zeros(self.num_envs, dtype=np.int64)  # the time limit's counters of the host-driven path (no reserve)
        self.max_ep_len = None
        for i, e in enumerate(self.envs):
            e._attach(self.engine, i)
        self._rs_scene = [None] * reserve   # scene assigned to each reserve slot
        # slot states live on the device (occ_auto_reset); this is the host's copy as of the last report it read
        self._rs_state = np.zeros(reserve, dtype=np.int32)
        self._pending = None                # (report handle, obs, out, infos) of the last step, not yet read
        self._late = []                     # slots refilled AFTER the pairing of the step whose report is pending
        self._half = None                   # (report handle, infos, taken slots, their envs) of a report read by the
        #                                     fast half of _drain whose per-env bookkeeping has not run yet
        self._fin_recent = 0.0              # decaying maximum of the number of envs that finished in one step
        self._warm = False
        # optional torch.cuda.Event: recorded by an asynchronous consumer of the last step's ``obs`` (another stream)
        # once it has read it; the in-place reset fallback below waits on it before overwriting rows of ``obs``
        self.obs_consumer_event = None
        # output ring / recycled outputs: every output set remembers the consumer event of the step that last used it (a
        # set is not overwritten before the side stream that still reads it - RecordExchange packs step k's records while
        # step k + 1 renders - is done); _last_set = the set of the previous step
        self._last_set = None

    def step_async(self, actions):
        self.actions = actions

    def use_output_ring(self, k: int) -> None:
        """Switch the engine to k >= 2 persistent output sets (0: back to fresh tensors every step).  Needs the reserve
        (>= 16 envs sharing one dataset): the ring lives on the whole-batch step path."""
        self._drain()
        eng = self.engine
        if k and not eng.R:
            raise ValueError("the output ring needs the batched step path with a reserve (>= 16 envs on one dataset)")
        if k == 1 or k < 0:
            raise ValueError("output ring: 0 (fresh outputs) or k >= 2 sets")
        eng.output_ring, eng._ring, eng._ring_pos, eng._picked = int(k), None, 0, None
        self._last_set = None

    def use_recycled_outputs(self, max_sets: int = 4) -> None:
        """Recycled outputs (class docstring) with at most ``max_sets`` persistent output sets; 0 = every step allocates
        its tensors (and the combine kernel writes every pixel of them)."""
        self._drain()
        eng = self.engine
        if max_sets < 0:
            raise ValueError("max_sets must be >= 0")
        eng.output_recycle, eng._picked = (int(max_sets) if eng.R else 0), None
        if not eng.output_ring:
            eng._ring = None
        self._last_set = None

    def stagger_ages(self, seed=None) -> None:
        """Spread the envs' episode ages uniformly over [0, max_ep_len): envs that were reset together would otherwise
        all reach the time limit in the same step (N simultaneous resets against a reserve of N/8 slots, i.e. the
        synchronous fallback every max_ep_len steps); afterwards about N / max_ep_len expire per step.  A deviation of
        the batched loop only: every env's FIRST episode is shorter than max_ep_len."""
        if not self.max_ep_len:
            return
        self._drain()
        g = torch.Generator().manual_seed(int(seed)) if seed is not None else None
        ages = torch.randint(0, int(self.max_ep_len), (self.num_envs,), generator=g, dtype=torch.int32)
        self.engine.age.copy_(ages)
        self._age_host[:] = ages.numpy()

    def _refill_reserve(self, slots):
        """Draw a new candidate scene for the given (EMPTY) reserve slots on the host and hand them to the device."""
        if not len(slots):
            return
        from .environment import MAX_MESH_FACES, sample_scene

        ds = self.envs[0].shapenet_dataset
        pool = self.engine.pool
        for r in slots:
            while True:
                try:
                    ids, offs = sample_scene(ds, pool)
                except (IndexError, KeyError, ValueError, OSError):
                    continue
                if max(pool.num_faces(m) for m in ids) <= MAX_MESH_FACES:  # environment.py:296-298
                    break
            self._rs_scene[r] = (ids, offs)
            self._rs_state[r] = nat.RS_PENDING
        self.engine.refill_reserve(slots, [self._rs_scene[r][0] for r in slots], [self._rs_scene[r][1] for r in slots])

    def _warm_reserve(self):
        """Run the reset rejection loop for every reserve slot that does not hold an accepted scene yet (batched,
        synchronous, host-driven).  Called from reset() and before the first step so that steady state starts at
        once; afterwards the loop advances one try per step inside the step launches, on the device."""
        self._drain()
        eng, R = self.engine, self.engine.R
        pending = [r for r in range(R) if self._rs_state[r] != nat.RS_READY]
        self._refill_reserve([r for r in pending if self._rs_scene[r] is None or self._rs_state[r] == nat.RS_EMPTY])
        tries = {r: 0 for r in pending}
        while pending:
            res = eng.evaluate_scenes([self._rs_scene[r][0] for r in pending], [self._rs_scene[r][1] for r in pending],
                                      4.0, 0.0, 0.0)
            ok = (res["loss"] > 0.1).cpu().tolist()
            redraw, acc_slots, acc_cand = [], [], []
            for j, r in enumerate(pending):
                tries[r] += 1
                if ok[j] or tries[r] >= 10:
                    acc_slots.append(r)
                    acc_cand.append(j)
                else:
                    redraw.append(r)
            eng.install_reserve(acc_slots, res, acc_cand)  # READY slots are never rendered by a step: store this render
            self._refill_reserve(redraw)
            pending = redraw
        self._rs_state[:] = nat.RS_READY
        eng.set_reserve_state(self._rs_state, np.zeros(R, dtype=np.int32))
        self._warm = True

    def _drain(self, defer_refill=False, finish=True):
        """Read the auto-reset report of the last step (the ONE host sync per batched step, taken as late as
        possible: at the start of the next step or when infos are first read) and do the host's share: scene
        bookkeeping of the envs that were reset, terminal observations, new candidate scenes for the slots the
        device emptied, and the synchronous fallback if the reserve ran dry.  With ``defer_refill`` the EMPTY
        slots are returned instead of refilled (step_wait refills them after it has launched the next step).

        The GPU is idle from the moment the report arrives until the next step's first launch, so the work is split:
        this half does only what that launch depends on (status, the host copy of the scene mesh ids that sizes the
        record arrays, the fallback reset, the slot states); with ``finish=False`` the per-env bookkeeping
        (terminal observations, scenes, stored renders) is left to ``_drain_finish``, which step_wait calls right
        after the launch."""
        self._drain_finish()  # bookkeeping left over from an earlier fast drain
        if self._pending is None:
            return []
        pend, obs, out, infos = self._pending
        self._pending = None
        infos._resolve = None
        eng, N, R = self.engine, self.num_envs, self.engine.R
        pend["event"].synchronize()
        rep = pend["report_host"].numpy()
        if rep[N + 2 * R]:
            eng.check_status()
        state = rep[N:N + R].copy()
        assign = rep[N + R:N + 2 * R]
        self._fin_recent = max(0.9 * self._fin_recent, float(np.count_nonzero(rep[:N])))
        taken = np.nonzero(assign >= 0)[0]
        took = assign[taken].astype(np.int64)  # env that took slot taken[j]
        eng.note_commits(took, taken)
        self._half = (pend, infos, taken.tolist(), took.tolist())
        for i in np.nonzero(rep[:N] == 2)[0].tolist():  # reset by the time limit, not done (trainRL.py:191-229)
            infos.set(i, "TimeLimit.truncated", True)
        if rep[N + 2 * R + 1]:  # reserve exhausted: synchronous batched reset for the rest
            self._drain_finish()
            done_envs = set(np.nonzero(rep[:N])[0].tolist())
            left = sorted(done_envs - set(took.tolist()))
            term = obs[left].clone()
            for j, i in enumerate(left):
                infos.set(i, "terminal_observation", term[j:j + 1])
            if self.obs_consumer_event is not None:
                torch.cuda.current_stream(eng.device).wait_event(self.obs_consumer_event)
            obs[left] = self._reset_envs(left, torch.zeros(len(left)))[:, 0]
            if out.get("rect") is not None:  # output ring: these rows of the set now hold whole frames
                out["rect"][left] = torch.tensor([0, 0, eng.S - 1, eng.S - 1], dtype=torch.int32, device=eng.device)
        # slots refilled after this report's pairing ran still read EMPTY in it: they are PENDING by now
        state[self._late] = nat.RS_PENDING
        self._late = []
        self._rs_state = state
        empty = np.nonzero(state == nat.RS_EMPTY)[0].tolist()
        if finish:
            self._drain_finish()
        if defer_refill:
            return empty
        self._refill_reserve(empty)
        return []

    def _drain_sync(self) -> bool:
        """Wait for the last step's auto-reset report (the one host sync per batched step) and tell whether it needs the
        host BEFORE the next launch: a finished env left without a reserve slot (synchronous fallback reset) or a status
        bit.  Everything else the report says is bookkeeping that can follow the launch (``_drain``): the engine sizes
        its record arrays for any commits it has not heard of (``_records_needed_all(ahead=True)``)."""
        if self._pending is None:
            return False
        pend = self._pending[0]
        pend["event"].synchronize()
        rep = pend["report_host"].numpy()
        N, R = self.num_envs, self.engine.R
        return bool(rep[N + 2 * R] or rep[N + 2 * R + 1])

    def _drain_finish(self):
        """Second half of _drain: what the envs that took a reserve slot get from it.  Must run before the slots are
        refilled (``_rs_scene``) and before the launch after next (which may overwrite their stored renders)."""
        if self._half is None:
            return
        pend, infos, taken, took = self._half
        self._half = None
        if not taken:
            return
        eng = self.engine
        for r, i in zip(taken, took):
            # save final observation where user can get it, then reset (SubProcVecEnv.py:211-214)
            infos.set(i, "terminal_observation", pend["term"][r:r + 1])
            self.envs[i]._scene = self._rs_scene[r]
            # the slot's stored occlusion image, copied out by the commit itself (OccAutoResetOpts.reset_full_state)
            self.envs[i].image = pend["reset_fs"][r:r + 1]

    def step_wait(self):
        eng = self.engine
        actions = self.actions
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32)
        if actions.device != eng.device:
            actions = actions.to(eng.device)
        R, N = eng.R, self.num_envs
        if R:
            if not self._warm:
                self._warm_reserve()
            # the report of the previous step is read as late as possible: after this step's outputs are allocated
            # and its launch arguments are built, right before its first kernel launch
            if self._last_set is not None:  # what the caller attached after the previous step belongs to ITS set
                self._last_set["event"] = self.obs_consumer_event
            _, oset = eng.pick_output_set()  # the set this step writes (None: freshly allocated tensors)
            if oset is not None and oset["event"] is not None:
                torch.cuda.current_stream(eng.device).wait_event(oset["event"])
                oset["event"] = None
            self._last_set = oset
            empty = []

            def pre_launch():  # the GPU idles from the report's arrival to this step's first launch: only what must precede it
                if self._drain_sync():
                    empty.extend(self._drain(defer_refill=True, finish=False))

            obs, rewards, dones, full_state, loss, out = eng.step(actions, with_reserve=True, pre_launch=pre_launch)
            # the rest of the previous step's report, now that this step is on its way
            empty.extend(self._drain(defer_refill=True, finish=False))
            self._drain_finish()
            # finished envs are reset ON THE DEVICE from the reserve (pairing + commit); the host reads the
            # report later (_drain).  NB out["obs_all"][:N] IS obs: the commit writes the reset observation in place
            eng.max_ep_len = int(self.max_ep_len or 0)
            pend = eng.auto_reset(out)
            infos = _LazyInfos(out["pos"], full_state, loss, resolve=self._drain)
            self._pending = (pend, obs, out, infos)
            # new candidate scenes for the slots emptied one step ago: off the critical path (the GPU is busy
            # with this step); they are rendered from the next step on
            self._refill_reserve(empty)
            self._late = empty
            # few accepted scenes left (as far as the host knows, one step old) compared with how many envs have
            # been finishing per step lately: do not run ahead, a synchronous fallback would have to come first
            if int((self._rs_state == nat.RS_READY).sum()) < max(2, int(3 * self._fin_recent) + 2):
                self._drain()
            return obs, rewards, dones, infos
        obs, rewards, dones, full_state, loss = eng.step(actions)
        flags = eng.step_flags(dones.to(torch.uint8), None)
        infos = _LazyInfos(eng.camera_position.clone(), full_state, loss)
        fl = flags.cpu().numpy()
        if fl[-1]:
            eng.check_status()
        fin = fl[:N] != 0
        self._age_host += 1
        if self.max_ep_len:  # the time limit (trainRL.py:191-229): reset, not done
            expired = (self._age_host >= int(self.max_ep_len)) & ~fin
            for i in np.nonzero(expired)[0].tolist():
                infos.set(i, "TimeLimit.truncated", True)
            fin = fin | expired
        fin_l = np.nonzero(fin)[0].tolist()
        if fin_l:
            # save final observation where user can get it, then reset (SubProcVecEnv.py:211-214)
            term = obs[fin_l].clone()
            for j, i in enumerate(fin_l):
                infos.set(i, "terminal_observation", term[j:j + 1])
            obs[fin_l] = self._reset_envs(fin_l, torch.zeros(len(fin_l)))[:, 0]
        return obs, rewards, dones, infos

    def seed(self, seed=None):
        return [env.seed(seed + idx) for idx, env in enumerate(self.envs)]

    def _reset_envs(self, indices, az):
        """reset() of the listed envs, batched and speculative.  The reference retries scene draws one at a time
        until the initial occlusion loss exceeds 0.1, at most 10 times, and keeps the 10th regardless
        (environment.py:288-327).  Here every round draws several candidate scenes per pending env on the host
        (in try order), renders ALL candidates in one launch sequence, and each env takes its first accepted
        candidate -- the same outcome per env as trying them one by one, in far fewer GPU round trips."""
        self._drain()
        eng, N = self.engine, self.num_envs
        self._age_host[list(indices)] = 0  # (the device-side counters: OcclusionEngine.commit_reset)
        az = torch.as_tensor(az, dtype=torch.float32).reshape(-1)
        pos = {i: j for j, i in enumerate(indices)}
        pending = list(indices)
        tries = {i: 0 for i in indices}
        max_resets = 10
        obs_all = torch.empty(len(indices), 1, 4, eng.S, eng.S, dtype=torch.float32, device=eng.device)
        while pending:
            per_env = max(1, min(4, N // len(pending)))
            cand_env, cand_scene = [], []
            for i in pending:
                for _ in range(min(per_env, max_resets - tries[i])):
                    fails = 0
                    while not self.envs[i]._new_scene(upload=False):
                        fails += 1
                        if fails >= 1000:
                            raise RuntimeError("reset(): could not load a scene")
                    cand_env.append(i)
                    cand_scene.append(self.envs[i]._scene)
            caz = az[torch.tensor([pos[i] for i in cand_env])]
            res = eng.evaluate_scenes([sc[0] for sc in cand_scene], [sc[1] for sc in cand_scene], 4.0, caz, 0.0)
            ok = (res["loss"] > 0.1).cpu().tolist()
            eng.check_status()
            chosen_env, chosen_cand, still = [], [], []
            k = 0
            while k < len(cand_env):
                i = cand_env[k]
                pick = None
                k0 = k
                while k < len(cand_env) and cand_env[k] == i:
                    tries_now = tries[i] + (k - k0) + 1
                    if pick is None and (ok[k] or tries_now >= max_resets):
                        pick = k
                    k += 1
                tries[i] += k - k0
                if pick is None:
                    still.append(i)
                else:
                    chosen_env.append(i)
                    chosen_cand.append(pick)
                    self.envs[i]._scene = cand_scene[pick]
            if chosen_env:
                eng.commit_reset(chosen_env, chosen_cand, res)
                cc = torch.tensor(chosen_cand, device=eng.device)
                obs_all[torch.tensor([pos[i] for i in chosen_env], device=eng.device), 0] = res["obs"][cc]
                for i, c in zip(chosen_env, chosen_cand):
                    self.envs[i].image = res["full_state"][c:c + 1]
            pending = still
        return obs_all

    def reset(self):
        az = [np.random.default_rng().uniform(low=-40, high=40) for _ in range(self.num_envs)]
        obs = self._reset_envs(list(range(self.num_envs)), az)
        if self.engine.R:
            self._warm_reserve()
        return obs

    def close(self):
        for env in self.envs:
            env.close()

    def get_images(self) -> Sequence[np.ndarray]:
        return [env.render(mode="rgb_array")[0][0, ..., :3].detach().cpu().numpy() for env in self.envs]

    def render(self, mode: str = "human"):
        if self.num_envs == 1:
            return self.envs[0].render(mode=mode)
        return super().render(mode=mode)

    def get_attr(self, attr_name, indices=None):
        return [getattr(env_i, attr_name) for env_i in self._get_target_envs(indices)]

    def set_attr(self, attr_name, value, indices=None):
        for env_i in self._get_target_envs(indices):
            setattr(env_i, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(env_i, method_name)(*method_args, **method_kwargs) for env_i in self._get_target_envs(indices)]

    def _get_target_envs(self, indices):
        self._drain()  # the env objects' scene bookkeeping must be current
        return [self.envs[i] for i in self._get_indices(indices)]
