"""Mesh sources for the OcclusionEnv hot path: OBJ reader, procedural ShapeNet-sized meshes,
and the GPU-resident packed mesh pool the kernels index.

Reference behaviour mirrored here:
  * ``load_obj("./data/teapot.obj")`` verts / faces.verts_idx (/root/reference/environment.py:56-57);
    the teapot file uses ``f a//na b//nb c//nc`` records.
  * the ShapeNetCore duck-type consumed by ``load_shapenet_meshes`` (environment.py:106-135):
    ``synset_dict``, ``synset_inv``, ``synset_start_idxs``, ``synset_num_models`` and
    ``dataset[i] -> {"verts","faces","textures","synset_id","label","model_id"}``.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def default_teapot_path() -> str:
    """``./data/teapot.obj`` like the reference (environment.py:56), else the copy shipped in this repo's data/."""
    for p in ("./data/teapot.obj", os.path.join(_HERE, "..", "data", "teapot.obj")):
        if os.path.exists(p):
            return os.path.abspath(p)
    raise FileNotFoundError("teapot.obj not found (looked in ./data and <repo>/data)")


def load_obj(path: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """Minimal Wavefront reader: returns (verts (V,3) f32, faces (F,3) int64, 0-based).
    Handles ``v``, ``f a``, ``f a/b``, ``f a//c``, ``f a/b/c``, negative indices; polygons are fan-triangulated."""
    verts: List[List[float]] = []
    faces: List[List[int]] = []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return torch.tensor(verts, dtype=torch.float32), torch.tensor(faces, dtype=torch.int64)


# ---- procedural meshes (SURVEY.md §8d "ShapeNet-size synthetic pool") -------------------------
def icosphere(subdiv: int = 4) -> Tuple[np.ndarray, np.ndarray]:
    """Unit icosphere; subdiv=4 -> V=2562, F=5120."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    verts = [tuple(x) for x in v]
    for _ in range(subdiv):
        cache: Dict[Tuple[int, int], int] = {}
        nf = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = (np.array(verts[a]) + np.array(verts[b])) / 2.0
                m /= np.linalg.norm(m)
                cache[key] = len(verts)
                verts.append(tuple(m))
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = np.array(nf, dtype=np.int64)
    return np.array(verts, dtype=np.float64), f


def torus(nu: int = 64, nv: int = 40, R: float = 1.0, r: float = 0.4) -> Tuple[np.ndarray, np.ndarray]:
    """Torus; 64x40 -> V=2560, F=5120."""
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    x = (R + r * np.cos(vv)) * np.cos(uu)
    y = (R + r * np.cos(vv)) * np.sin(uu)
    z = r * np.sin(vv)
    verts = np.stack([x, y, z], -1).reshape(-1, 3)
    faces = []
    for i in range(nu):
        for j in range(nv):
            a = i * nv + j
            b = ((i + 1) % nu) * nv + j
            c = ((i + 1) % nu) * nv + (j + 1) % nv
            d = i * nv + (j + 1) % nv
            faces += [[a, b, c], [a, c, d]]
    return verts, np.array(faces, dtype=np.int64)


def _orient_outward(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Flip the winding if the signed volume is negative (PyTorch3D culls by screen-space winding)."""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum()
    return faces if vol > 0 else faces[:, ::-1].copy()


def _normalise(verts: np.ndarray) -> np.ndarray:
    """Centre and scale to unit bounding-box diagonal (ShapeNet convention)."""
    lo, hi = verts.min(0), verts.max(0)
    verts = verts - (lo + hi) / 2.0
    return verts / np.linalg.norm(hi - lo)


def synthetic_mesh(rng: np.random.Generator, faces_level: int = 5120) -> Tuple[torch.Tensor, torch.Tensor]:
    """One seeded watertight mesh: icosphere or torus with smooth radial noise and anisotropic scale.
    faces_level in {1280, 5120, 20480}."""
    kind = rng.integers(0, 2)
    if kind == 0 or faces_level != 5120:
        sub = {1280: 3, 5120: 4, 20480: 5}[faces_level]
        v, f = icosphere(sub)
        # smooth radial noise from a few random low-order lobes
        k = rng.normal(size=(4, 3))
        amp = 0.1 * rng.normal(size=4)
        rad = 1.0 + sum(a * np.sin(2.0 * v @ kk) for a, kk in zip(amp, k))
        v = v * rad[:, None]
    else:
        v, f = torus(64, 40, 1.0, float(rng.uniform(0.25, 0.5)))
    v = v * rng.uniform(0.4, 1.0, size=3)[None, :]
    # random rotation
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    v = _normalise(v @ Rm.T)
    f = _orient_outward(v, f)
    return torch.tensor(v, dtype=torch.float32), torch.tensor(f, dtype=torch.int64)


class SyntheticShapeNet:
    """In-memory stand-in with the ShapeNetCore duck-type ``load_shapenet_meshes`` consumes
    (environment.py:106-135).  ``mixed=True`` draws face counts {1280, 5120, 20480} with weights
    {0.25, 0.6, 0.15}; otherwise every mesh has 5120 faces (the headline workload)."""

    def __init__(self, n_models: int = 64, seed: int = 1234, mixed: bool = False, n_categories: int = 4,
                 textured: bool = False, atlas_res: int = 4, cache_dir: Optional[str] = None):
        """``cache_dir``: keep the generated (untextured) meshes in a file there, keyed by the arguments and by the
        generator's own source, and load them from it next time (building 1 024 meshes takes ~26 s of one core; eight
        ranks of one node start together - the first to finish writes the file, atomically, the rest of the runs read it)."""
        self.models: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.atlases: List[Optional[torch.Tensor]] = []
        cache = None
        if cache_dir is not None and not textured:
            import hashlib
            import inspect

            tag = hashlib.sha1((inspect.getsource(synthetic_mesh) + inspect.getsource(icosphere) + inspect.getsource(torus)
                                + inspect.getsource(_normalise) + inspect.getsource(_orient_outward)).encode()).hexdigest()[:12]
            cache = os.path.join(cache_dir, f"occ_synth_{n_models}_{seed}_{int(mixed)}_{tag}.pt")
            if os.path.exists(cache):
                try:
                    loaded = torch.load(cache, weights_only=True)
                    if len(loaded) == n_models:
                        self.models = [(v, f) for v, f in loaded]
                except Exception:  # noqa: BLE001 - a torn or foreign file: build instead
                    self.models = []
        if not self.models:
            rng = np.random.default_rng(seed)
            for _ in range(n_models):
                level = int(rng.choice([1280, 5120, 20480], p=[0.25, 0.6, 0.15])) if mixed else 5120
                self.models.append(synthetic_mesh(rng, level))
                # ShapeNetCore(load_textures=True) hands out a per-face (F, R, R, 3) atlas, R = 4 (environment.py:127)
                self.atlases.append(torch.tensor(rng.random((level, atlas_res, atlas_res, 3)), dtype=torch.float32)
                                    if textured else None)
            if cache is not None:
                try:
                    tmp = f"{cache}.{os.getpid()}.tmp"
                    torch.save([(v, f) for v, f in self.models], tmp)
                    os.replace(tmp, cache)  # atomic: a reader sees the whole file or none
                except OSError:
                    pass
        if not self.atlases:
            self.atlases = [None] * n_models
        n_categories = max(1, min(n_categories, n_models))
        per = n_models // n_categories
        self.synset_dict = {f"{i:08d}": f"synthetic_{i}" for i in range(n_categories)}
        self.synset_inv = {v: k for k, v in self.synset_dict.items()}
        self.synset_start_idxs = {k: i * per for i, k in enumerate(self.synset_dict)}
        self.synset_num_models = {k: (per if i < n_categories - 1 else n_models - per * (n_categories - 1))
                                  for i, k in enumerate(self.synset_dict)}

    def __len__(self):
        return len(self.models)

    def __getitem__(self, idx):
        idx = int(idx)
        v, f = self.models[idx]
        cat = max(k for k, s in self.synset_start_idxs.items() if s <= idx)
        return {"verts": v, "faces": f, "textures": self.atlases[idx], "synset_id": cat, "label": self.synset_dict[cat],
                "model_id": f"model_{idx:05d}"}


class _GrowBuf:
    """Append-only device array with geometric capacity growth: appending a mesh uploads that mesh only (the pool
    of a ShapeNet-size dataset grows by a few models per step for a long time; re-packing everything on every
    addition would make a step cost O(pool size))."""

    def __init__(self, device, dtype, tail=(), min_cap=1 << 14):
        self.device, self.dtype, self.tail, self.min_cap = device, dtype, tuple(tail), min_cap
        self.t: Optional[torch.Tensor] = None
        self.n = 0

    def append(self, host: torch.Tensor) -> None:
        k = int(host.shape[0])
        need = self.n + k
        cap = 0 if self.t is None else int(self.t.shape[0])
        if need > cap:
            grown = torch.empty((max(need, 2 * cap, self.min_cap),) + self.tail, dtype=self.dtype, device=self.device)
            if self.n:
                grown[: self.n].copy_(self.t[: self.n])  # device-to-device, stream-ordered
            self.t = grown
        if k:
            self.t[self.n:need].copy_(host.to(self.dtype))
        self.n = need


class MeshPool:
    """Packed, GPU-resident pool: ``verts (sumV,3) f32``, ``faces (sumF,3) i32`` (vertex ids local to the mesh),
    ``vert_off (M+1) i32``, ``face_off (M+1) i32`` (+ optional per-face texture atlases).  Meshes can be appended at any
    time; ``device_tensors`` uploads the meshes added since the last call (only those) into growable device arrays."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._verts: List[torch.Tensor] = []
        self._faces: List[torch.Tensor] = []
        self._atlas: List[Optional[torch.Tensor]] = []  # per mesh (F,R,R,3) f32 or None (white vertices)
        self.atlas_res = 0
        self._keys: Dict[object, int] = {}
        self.version = 0
        self._max_faces = 0
        self._max_verts = 0
        self._flushed = 0  # meshes already on the device
        self._sum_v = self._sum_f = self._sum_a = 0
        self._d_verts = _GrowBuf(self.device, torch.float32, (3,))
        self._d_faces = _GrowBuf(self.device, torch.int32, (3,))
        self._d_voff = _GrowBuf(self.device, torch.int32, (), 1 << 10)
        self._d_foff = _GrowBuf(self.device, torch.int32, (), 1 << 10)
        self._d_atlas = _GrowBuf(self.device, torch.float32, (), 1 << 16)
        self._d_aoff = _GrowBuf(self.device, torch.int64, (), 1 << 10)
        self._any_atlas = False

    def __len__(self):
        return len(self._verts)

    @property
    def max_faces(self) -> int:
        return self._max_faces

    @property
    def max_verts(self) -> int:
        return self._max_verts

    def num_faces(self, mesh_id: int) -> int:
        return int(self._faces[mesh_id].shape[0])

    def add(self, verts: torch.Tensor, faces: torch.Tensor, key=None, atlas: Optional[torch.Tensor] = None) -> int:
        if key is not None and key in self._keys:
            return self._keys[key]
        if atlas is not None:
            atlas = torch.as_tensor(atlas, dtype=torch.float32).detach().cpu().contiguous()
            if atlas.ndim != 4 or atlas.shape[0] != faces.shape[0] or atlas.shape[1] != atlas.shape[2] or atlas.shape[3] != 3:
                raise ValueError("atlas must be (F, R, R, 3)")
            if self.atlas_res not in (0, int(atlas.shape[1])):
                raise ValueError("all texture atlases of a pool must share one resolution R")
            self.atlas_res = int(atlas.shape[1])
        verts = torch.as_tensor(verts, dtype=torch.float32).detach().cpu().contiguous()
        faces = torch.as_tensor(faces).detach().cpu().to(torch.int32).contiguous()
        if verts.ndim != 2 or verts.shape[1] != 3 or faces.ndim != 2 or faces.shape[1] != 3:
            raise ValueError("verts must be (V,3) and faces (F,3)")
        if faces.numel() and (int(faces.min()) < 0 or int(faces.max()) >= verts.shape[0]):
            raise ValueError("face index out of range")
        self._verts.append(verts)
        self._faces.append(faces)
        self._atlas.append(atlas)
        self._max_faces = max(self._max_faces, int(faces.shape[0]))
        self._max_verts = max(self._max_verts, int(verts.shape[0]))
        mid = len(self._verts) - 1
        if key is not None:
            self._keys[key] = mid
        self.version += 1
        return mid

    def get(self, mesh_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._verts[mesh_id], self._faces[mesh_id].long()

    def get_atlas(self, mesh_id: int) -> Optional[torch.Tensor]:
        return self._atlas[mesh_id]

    def _flush(self) -> None:
        """Upload the meshes added since the last flush."""
        m0, m1 = self._flushed, len(self._verts)
        if m0 == m1:
            return
        if m0 == 0:  # offsets arrays start with the leading zero
            self._d_voff.append(torch.zeros(1, dtype=torch.int32))
            self._d_foff.append(torch.zeros(1, dtype=torch.int32))
        voff, foff, aoff = [], [], []
        for m in range(m0, m1):
            self._sum_v += int(self._verts[m].shape[0])
            self._sum_f += int(self._faces[m].shape[0])
            voff.append(self._sum_v)
            foff.append(self._sum_f)
            a = self._atlas[m]
            if a is None:
                aoff.append(-1)
            else:
                aoff.append(self._sum_a)
                self._sum_a += a.numel()
                self._any_atlas = True
        self._d_verts.append(torch.cat(self._verts[m0:m1]))
        self._d_faces.append(torch.cat(self._faces[m0:m1]))
        self._d_voff.append(torch.tensor(voff, dtype=torch.int32))
        self._d_foff.append(torch.tensor(foff, dtype=torch.int32))
        self._d_aoff.append(torch.tensor(aoff, dtype=torch.int64))
        chunks = [self._atlas[m].reshape(-1) for m in range(m0, m1) if self._atlas[m] is not None]
        if chunks:
            self._d_atlas.append(torch.cat(chunks))
        self._flushed = m1

    def vertex_normals_tensor(self) -> torch.Tensor:
        """(sumV,3) f32 on the device: [P3D] ``Meshes.verts_normals_packed()`` of every pool mesh (each face adds its
        area-weighted normal to its three corners, sums normalised with eps 1e-6) - what HardPhongShader /
        SoftPhongShader interpolate (environment.py:281-282).  Built on first use and when the pool has grown."""
        self._flush()
        if getattr(self, "_vn_version", -1) != self._flushed:
            out = []
            for v, f in zip(self._verts[: self._flushed], self._faces[: self._flushed]):
                fl = f.long()
                vf = v[fl]
                n = torch.zeros_like(v)
                n.index_add_(0, fl[:, 1], torch.cross(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1], dim=1))
                n.index_add_(0, fl[:, 2], torch.cross(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2], dim=1))
                n.index_add_(0, fl[:, 0], torch.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0], dim=1))
                out.append(torch.nn.functional.normalize(n, eps=1e-6, dim=1))
            self._d_vnorm = torch.cat(out).to(self.device).contiguous()
            self._vn_version = self._flushed
        return self._d_vnorm

    def atlas_tensors(self):
        """(packed atlas floats, per-mesh float offsets int64 with -1 = untextured) on the device, or (None, None)."""
        self._flush()
        if not self._any_atlas:
            return None, None
        return self._d_atlas.t[: self._sum_a], self._d_aoff.t[: self._flushed]

    def device_tensors(self):
        """(verts, faces, vert_off, face_off) device arrays (views of the growable buffers)."""
        if not self._verts:
            raise ValueError("empty mesh pool")
        self._flush()
        m = self._flushed
        return (self._d_verts.t[: self._sum_v], self._d_faces.t[: self._sum_f], self._d_voff.t[: m + 1],
                self._d_foff.t[: m + 1])
