"""Drop-in ``environment`` module: ``OcclusionEnv`` with the reference's gym-style surface
(/root/reference/environment.py:201-402), running on the batched HIP engine.

Kept from the reference: constructor ``OcclusionEnv(data=None, img_size=512)``, attributes
(``observation_space``, ``action_space``, ``renderMode``, ``step_size``, ``metadata``, ``device``,
``elevation``/``azimuth``/``radius`` (1,), ``camera_position`` (3,), ``fullReward``, ``objectMass``,
``image``, ``meshes``), methods ``seed, createRenderers, reset, step, render, close, detach``, return
shapes and the reward / done rules.

Documented deviations (all from SURVEY.md §0):
  * ``data=None``: the reference's teapot scene cannot run (3 meshes, ``self.meshes[3]`` indexed,
    environment.py:88 vs :318 -> infinite loop).  Here the default scene is three teapots laid out
    like the ShapeNet scene: offsets (0,0,0), (x2,0,1), (-x2,0,2), x2 ~ np.random.randn().
  * ``reset`` only swallows data-loading errors, not every exception (reference: bare ``except``).
  * gradients reach ``action`` through ``reward`` only (what every reference caller uses:
    demo.py:86, iterator.py:117, datasetGenerator.py:92, train_predict.py:52); ``observation`` and
    ``info['full_state']`` are returned detached.
"""
from __future__ import annotations

import itertools
import os
import random
import weakref
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native as nat
from .engine import OcclusionEngine
from .meshes import MeshPool, default_teapot_path, load_obj
from .spaces import Box

# The reference draws category and model from a FRESH unseeded ``np.random.default_rng()`` per draw
# (environment.py:106,119); one process-wide unseeded generator has the same distribution and none of the
# ~20 us construction cost per draw.
_SCENE_RNG = np.random.default_rng()

#: scenes containing a mesh with more faces than this are drawn again (the reference: 250 000,
#: environment.py:296-298).  Every (env, object) slot of the render workspace holds records for the largest mesh in
#: the pool, so lowering this bounds the memory of runs on datasets with a few huge models.
MAX_MESH_FACES = int(os.environ.get("OCC_MAX_MESH_FACES", 250000))
_OVERSIZE = set()  # (dataset token, model index) of models found to exceed it
_DS_TOKENS: Dict[int, tuple] = {}  # id(dataset) -> (weak reference, token)
_DS_COUNTER = itertools.count(1)


def _dataset_token(dataset) -> int:
    """A number that identifies ``dataset`` for as long as the process lives.  ``id()`` alone does not: the mesh pool
    is shared by all envs of the process and outlives datasets, and a NEW dataset object can be given the address of
    one that was garbage-collected - its models would then be served from the old one's pool entries."""
    ent = _DS_TOKENS.get(id(dataset))
    if ent is not None and ent[0]() is dataset:
        return ent[1]
    tok = next(_DS_COUNTER)
    try:
        ref = weakref.ref(dataset)
    except TypeError:  # not weak-referenceable: keep it alive, its address can then never be handed out again
        ref = (lambda d=dataset: d)
    _DS_TOKENS[id(dataset)] = (ref, tok)
    return tok


def seed_scene_rng(seed=None) -> None:
    """Test hook: make the category/model draws reproducible (the reference's are not)."""
    global _SCENE_RNG
    _SCENE_RNG = np.random.default_rng(seed)

# one pool per device, shared by every env/VecEnv in the process (a ShapeNet model is uploaded once)
_POOLS: Dict[str, MeshPool] = {}


def shared_pool(device) -> MeshPool:
    key = str(torch.device(device))
    if key not in _POOLS:
        _POOLS[key] = MeshPool(device)
    return _POOLS[key]


class SceneMesh:
    """Tiny stand-in for the PyTorch3D ``Meshes`` objects the reference keeps in ``env.meshes``
    (environment.py:193): exposes verts/faces of one mesh (world space) for inspection."""

    def __init__(self, verts: torch.Tensor, faces: torch.Tensor):
        self._v, self._f = verts, faces

    def verts_list(self):
        return [self._v]

    def faces_list(self):
        return [self._f]

    def verts_packed(self):
        return self._v

    def faces_packed(self):
        return self._f

    def clone(self):
        return SceneMesh(self._v.clone(), self._f.clone())


def sample_scene(dataset, pool: MeshPool, num_objects: int = 3) -> Tuple[List[int], List[List[float]]]:
    """Scene choice of ``load_shapenet_meshes`` / default scene: returns (pool mesh ids, world offsets).
    environment.py:102-119 (category / model draw from a fresh ``np.random.default_rng()``), :147-148,:171
    (x2 = np.random.randn(); offsets (x2,0,distance/2), (-x2,0,distance), distance=2)."""
    distance = 2
    ids: List[int] = []
    if dataset is None:
        path = default_teapot_path()
        key = ("obj", path)
        if key not in pool._keys:
            v, f = load_obj(path)
            pool.add(v, f, key=key)
        ids = [pool._keys[key]] * num_objects
    else:
        for _ in range(num_objects):
            category_randn = _SCENE_RNG.integers(low=len(dataset.synset_dict))
            category_id = list(dataset.synset_dict.keys())[category_randn]
            low_idx = dataset.synset_start_idxs[category_id]
            high_idx = low_idx + dataset.synset_num_models[category_id]
            model_idx = int(_SCENE_RNG.integers(low=low_idx, high=high_idx))
            key = (_dataset_token(dataset), model_idx)
            if key in _OVERSIZE:
                raise ValueError("scene contains a mesh above MAX_MESH_FACES")  # callers draw again
            if key not in pool._keys:
                obj = dataset[model_idx]
                if int(obj["faces"].shape[0]) > MAX_MESH_FACES:
                    # the reference loads it and then discards the scene (environment.py:296-298); here it never
                    # enters the pool - the pool's largest mesh sizes every slot of the render workspace
                    _OVERSIZE.add(key)
                    raise ValueError("scene contains a mesh above MAX_MESH_FACES")
                # environment.py:126-129: TexturesAtlas when the model has textures, else white TexturesVertex
                pool.add(obj["verts"], obj["faces"], key=key, atlas=obj.get("textures"))
            ids.append(pool._keys[key])
        # environment.py:126-129 gives a model TexturesAtlas when it has textures and white TexturesVertex when it has
        # none; environment.py:191 then joins the three meshes with [P3D] join_meshes_as_scene, which RAISES when the
        # texture types differ ("all meshes must have the same type of texture"), and reset()'s bare except draws the
        # scene again (:329-330).  A mixed scene therefore never reaches the renderer in the reference: same here.
        if len({pool.get_atlas(m) is not None for m in ids}) > 1:
            raise ValueError("scene mixes textured and untextured models (the reference's join_meshes_as_scene raises)")
    x2 = float(np.random.randn())
    offsets = [[0.0, 0.0, 0.0], [x2, 0.0, distance / 2], [-x2, 0.0, float(distance)]][:num_objects]
    return ids, offsets


class OcclusionEnv:
    def __init__(self, data=None, img_size=512):
        self.metadata = "Blablabla"
        self._engine: Optional[OcclusionEngine] = None
        self._slot = 0
        self._norm_with_object_size = False
        self.img_size = img_size
        self.device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
        self.shapenet_dataset = data
        self.step_size = 0.05
        self.observation_space = Box(0, 1, shape=(4, img_size, img_size))
        self.action_space = Box(low=-0.1, high=0.1, shape=(2,))
        self.renderMode = ""  # 'human'
        self.image = None
        self._scene: Optional[Tuple[List[int], List[List[float]]]] = None
        self.faces_per_bin = 10000

    # ---- engine plumbing ------------------------------------------------------------------
    def _attach(self, engine: OcclusionEngine, slot: int) -> None:
        """Called by the batched VecEnv: this env becomes slot ``slot`` of a shared engine."""
        if engine.S != self.img_size:
            raise ValueError("all envs of a VecEnv must share img_size")
        self._engine, self._slot = engine, slot
        engine.set_norm_with_object_size(slot, self._norm_with_object_size)

    def _eng(self) -> OcclusionEngine:
        if self._engine is None:
            dev = torch.device(f"cuda:{torch.cuda.current_device()}") if torch.cuda.is_available() else self.device
            self._engine = OcclusionEngine(shared_pool(dev), 1, self.img_size, device=dev)
            self._slot = 0
            self._engine.set_norm_with_object_size(0, self._norm_with_object_size)
        return self._engine

    def _ids(self):
        return None if self._eng().N == 1 else [self._slot]

    # ---- state exposed like the reference's attributes --------------------------------------
    @property
    def normWithObjectSize(self) -> bool:
        """environment.py:208,324: True = reset() sets objectMass = sum_px (a1 + a2 + a3)^2 + 1 (the silhouettes' own mass)
        instead of loss + 1; takes effect at the env's next reset, like the reference's attribute."""
        return self._norm_with_object_size

    @normWithObjectSize.setter
    def normWithObjectSize(self, on) -> None:
        self._norm_with_object_size = bool(on)
        if self._engine is not None:
            self._engine.set_norm_with_object_size(self._slot, self._norm_with_object_size)

    @property
    def elevation(self):
        return self._eng().elevation[self._slot:self._slot + 1]

    @property
    def azimuth(self):
        return self._eng().azimuth[self._slot:self._slot + 1]

    @property
    def radius(self):
        return self._eng().radius[self._slot:self._slot + 1]

    @property
    def camera_position(self):
        return self._eng().camera_position[self._slot]

    @property
    def fullReward(self):
        return self._eng().full_reward[self._slot]

    @property
    def objectMass(self):
        return self._eng().object_mass[self._slot]

    @property
    def objects(self):
        """image1 + image2 + image3 (environment.py:320): RGB = 3, alpha = a1 + a2 + a3."""
        a = self._eng().alphas[self._slot].sum(0)
        return torch.cat([torch.full_like(a, 3.0)[..., None].expand(-1, -1, 3), a[..., None]], -1)[None]

    @property
    def meshes(self):
        """[full scene, obj1, obj2, obj3] in world space (environment.py:193)."""
        if self._scene is None:
            raise AttributeError("meshes: call reset() first")
        ids, offs = self._scene
        pool = self._eng().pool
        objs, vs, fs, base = [], [], [], 0
        for m, o in zip(ids, offs):
            v, f = pool.get(m)
            v = v + torch.tensor(o, dtype=torch.float32)
            objs.append(SceneMesh(v, f))
            vs.append(v)
            fs.append(f + base)
            base += v.shape[0]
        return [SceneMesh(torch.cat(vs), torch.cat(fs))] + objs

    # ---- gym surface ----------------------------------------------------------------------
    def seed(self, seed):
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)

    # ``shader`` picks the shader of phong_renderer, which the reference does by (un)commenting a line of
    # createRenderers (environment.py:281-283): "flat" = HardFlatShader (the one it runs), "hard_phong" =
    # HardPhongShader, "soft_phong" = SoftPhongShader.  Engine-wide: the envs of one VecEnv share it.
    _SHADERS = {"flat": nat.SHADER_FLAT, "hard_phong": nat.SHADER_HARD_PHONG, "soft_phong": nat.SHADER_SOFT_PHONG}

    @property
    def shader(self) -> str:
        code = self._eng().shader
        return next(k for k, v in self._SHADERS.items() if v == code)

    @shader.setter
    def shader(self, name: str) -> None:
        if name not in self._SHADERS:
            raise ValueError(f"shader must be one of {sorted(self._SHADERS)}")
        self._eng().shader = self._SHADERS[name]

    def createRenderers(self, mesh_size):
        """Renderer constants only (environment.py:234-284): sigma, blur radius, K=100/1, lights and camera
        defaults are compiled into the kernels (csrc/occ_constants.h); ``max_faces_per_bin`` has no
        counterpart because the tile kernel never drops faces."""
        self.faces_per_bin = max(mesh_size, 10000)

    def _new_scene(self, upload: bool = True) -> bool:
        eng = self._eng()
        try:
            ids, offs = sample_scene(self.shapenet_dataset, eng.pool)
        except (IndexError, KeyError, ValueError, OSError):
            return False
        if max(eng.pool.num_faces(m) for m in ids) > MAX_MESH_FACES:  # environment.py:296-298
            return False
        self._scene = (ids, offs)
        if upload:
            eng.set_scene([self._slot], [ids], [offs])
        self.createRenderers(max(eng.pool.num_faces(m) for m in ids) * 3)
        return True

    def reset(self, new_scene=True, radius=4.0, azimuth=0.0, elevation=0.0):
        eng = self._eng()
        max_resets = 10
        resets = 0
        while True:
            resets += 1
            if new_scene or self._scene is None:
                if not self._new_scene():
                    if resets >= 1000:
                        raise RuntimeError("reset(): could not load a scene")
                    continue
            obs, loss, full_state = eng.reset_render(self._ids(), radius, azimuth, elevation)
            self.image = full_state
            eng.check_status()
            if float(loss[0]) > 0.1 or resets >= max_resets:  # environment.py:327
                return obs

    def render(self, mode=None):
        eng = self._eng()
        obs = eng.render_hard(self._ids())
        depth = obs[:, 3:4].permute(0, 2, 3, 1).contiguous()
        mask = (depth != -1.0).to(obs.dtype)
        observation = torch.cat([obs[:, :3].permute(0, 2, 3, 1), mask], dim=-1)
        if self.renderMode == "human":
            import cv2  # noqa: WPS433 - optional, like the reference (environment.py:338-345)

            obs_img = observation.detach().squeeze().cpu().numpy()[..., :3]
            obs_depth = depth.detach().squeeze().cpu().numpy()
            obs_depth[obs_depth == -1] = 0
            obs_depth *= 51
            cv2.imshow("Environment", obs_img)
            cv2.imshow("Environment Depth", obs_depth.astype("uint8"))
            cv2.waitKey(25)
            return None
        return observation, depth

    def close(self):
        pass

    def step(self, action):
        eng = self._eng()
        if self._scene is None:
            raise RuntimeError("step() before reset()")
        act = action.reshape(1, 2)
        if act.device != eng.device:
            act = act.to(eng.device)
        obs, reward, done, full_state, loss = eng.step(act, self._ids())
        self.image = full_state
        info = {"full_state": self.image, "position": self.camera_position, "full_reward": loss[0]}
        return obs, reward[0], done[0], info

    def detach(self):
        """State tensors never carry autograd history here; kept for API parity (environment.py:398-402)."""
        return None
