"""Directory-backed ShapeNetCore reader that needs no PyTorch3D.

The reference opens the dataset with PyTorch3D's ``ShapeNetCore(root, version=2)`` (/root/reference/trainRL.py:66-71)
and ``load_shapenet_meshes`` consumes it through a small duck type (/root/reference/environment.py:106-135):

    dataset.synset_dict            {synset id: label}, iterated in insertion order (category draw, :107-108)
    dataset.synset_inv             {label: synset id} (:111)
    dataset.synset_start_idxs[id]  first model index of the category (:113)
    dataset.synset_num_models[id]  its model count (:116)
    dataset[i] -> {"verts" (V,3) f32, "faces" (F,3) i64, "textures" (F,R,R,3) f32 or None,
                   "synset_id", "model_id", "label"}                                             (:123-135)

[P3D; recalled from pytorch3d/datasets/shapenet/shapenet_core.py, pytorch3d/io/obj_io.py and pytorch3d/io/mtl_io.py,
v0.6.2 - not executable here, parity unpinned like the rest of Appendix A]:

* layout: version 1 ``<root>/<synset>/<model>/model.obj``, version 2 ``<root>/<synset>/<model>/models/model_normalized.obj``;
  categories = sub-directories of the root whose name is a known synset id, in sorted order; models of a category in
  sorted order; a model without its .obj is skipped with a warning.
* ``load_textures=True, texture_resolution=4`` are the defaults (what ``ShapeNetCore(dir, version=2)`` uses): the OBJ is
  read with ``create_texture_atlas=True`` and the item's "textures" is the per-face (F, R, R, 3) atlas:
  every face starts at 0.5 grey, takes its material's ``Kd`` when it has one, and - when the material has a ``map_Kd``
  image and the face has ``vt`` indices - the image sampled at the barycentric centres of an R x R grid over the
  face's uv triangle (lower-left cells ``(i + 1/3) / R``, upper-right cells mirrored ``(R - 1 - i + 2/3) / R``;
  bilinear, ``align_corners=True``, image flipped vertically; when any uv of the mesh leaves [0, 1] all are wrapped by ``% 1``).
* polygons are fan-triangulated (v0, v_k, v_k+1); negative indices count from the end of the list read so far.

Labels come from the dataset's own ``taxonomy.json`` when the root has one (ShapeNetCore v2 ships it), else from the
table below (the 55 ShapeNetCore synsets), else the synset id itself.
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

# ShapeNetCore synset ids -> first lemma (the label PyTorch3D's shapenet_synset_dict_v2.json carries)
SYNSET_LABELS = {
    "02691156": "airplane", "02747177": "trash bin", "02773838": "bag", "02801938": "basket", "02808440": "bathtub",
    "02818832": "bed", "02828884": "bench", "02843684": "birdhouse", "02871439": "bookshelf", "02876657": "bottle",
    "02880940": "bowl", "02924116": "bus", "02933112": "cabinet", "02942699": "camera", "02946921": "can",
    "02954340": "cap", "02958343": "car", "02992529": "cellphone", "03001627": "chair", "03046257": "clock",
    "03085013": "keyboard", "03207941": "dishwasher", "03211117": "display", "03261776": "earphone", "03325088": "faucet",
    "03337140": "file cabinet", "03467517": "guitar", "03513137": "helmet", "03593526": "jar", "03624134": "knife",
    "03636649": "lamp", "03642806": "laptop", "03691459": "loudspeaker", "03710193": "mailbox", "03759954": "microphone",
    "03761084": "microwaves", "03790512": "motorbike", "03797390": "mug", "03928116": "piano", "03938244": "pillow",
    "03948459": "pistol", "03991062": "flowerpot", "04004475": "printer", "04074963": "remote", "04090263": "rifle",
    "04099429": "rocket", "04225987": "skateboard", "04256520": "sofa", "04330267": "stove", "04379243": "table",
    "04401088": "telephone", "04460130": "tower", "04468005": "train", "04530566": "watercraft", "04554684": "washer",
}


# ---- Wavefront OBJ / MTL ----------------------------------------------------------------------------------------------
def parse_mtl(path: str) -> Tuple[Dict[str, dict], Dict[str, str]]:
    """``newmtl`` blocks -> ({name: {"diffuse_color": (3,) f32, ...}}, {name: image path of map_Kd})."""
    props: Dict[str, dict] = {}
    images: Dict[str, str] = {}
    name = None
    keys = {"Kd": "diffuse_color", "Ka": "ambient_color", "Ks": "specular_color"}
    if not os.path.isfile(path):
        return props, images
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            tok = line.strip().split()
            if not tok:
                continue
            if tok[0] == "newmtl":
                name = line.strip()[len("newmtl"):].strip()
                props[name] = {}
            elif name is None:
                continue
            elif tok[0] in keys and len(tok) >= 4:
                props[name][keys[tok[0]]] = torch.tensor([float(tok[1]), float(tok[2]), float(tok[3])], dtype=torch.float32)
            elif tok[0] == "Ns" and len(tok) >= 2:
                props[name]["shininess"] = torch.tensor([float(tok[1])], dtype=torch.float32)
            elif tok[0] == "map_Kd":
                # the file name is everything after the keyword (names with spaces occur in ShapeNet)
                fname = line.strip()[len("map_Kd"):].strip()
                images[name] = os.path.join(os.path.dirname(path), fname)
    return props, images


def read_image(path: str) -> Optional[torch.Tensor]:
    """RGB image -> (H, W, 3) f32 in [0, 1]; None when unreadable (a missing texture leaves the Kd colour in place)."""
    try:
        from PIL import Image

        with Image.open(path) as im:
            return torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0)
    except Exception as e:  # noqa: BLE001
        warnings.warn(f"texture image {path} could not be read: {e}")
        return None


def load_obj_full(path: str, load_textures: bool = True):
    """Wavefront reader with texture coordinates and materials.  Returns dict(verts (V,3) f32, faces (F,3) i64,
    verts_uvs (T,2) f32, faces_uvs (F,3) i64 [-1 = none], face_materials list[str or None] per face,
    material_props, material_images).  ``f`` tokens: ``a``, ``a/b``, ``a//c``, ``a/b/c``."""
    verts: List[List[float]] = []
    uvs: List[List[float]] = []
    faces: List[List[int]] = []
    faces_uv: List[List[int]] = []
    face_mat: List[Optional[str]] = []
    mtl_files: List[str] = []
    current = None
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("vt "):
                p = line.split()
                uvs.append([float(p[1]), float(p[2])])
            elif line.startswith("f "):
                vi, ti = [], []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    vi.append(i - 1 if i > 0 else len(verts) + i)
                    if len(parts) > 1 and parts[1] != "":
                        t = int(parts[1])
                        ti.append(t - 1 if t > 0 else len(uvs) + t)
                    else:
                        ti.append(-1)
                for k in range(1, len(vi) - 1):
                    faces.append([vi[0], vi[k], vi[k + 1]])
                    faces_uv.append([ti[0], ti[k], ti[k + 1]])
                    face_mat.append(current)
            elif line.startswith("usemtl"):
                current = line.strip()[len("usemtl"):].strip()
            elif line.startswith("mtllib"):
                mtl_files.append(line.strip()[len("mtllib"):].strip())
    props: Dict[str, dict] = {}
    images: Dict[str, str] = {}
    if load_textures:
        for m in mtl_files:
            p, im = parse_mtl(os.path.join(os.path.dirname(path), m))
            props.update(p)
            images.update(im)
    return dict(
        verts=torch.tensor(verts, dtype=torch.float32).reshape(-1, 3),
        faces=torch.tensor(faces, dtype=torch.int64).reshape(-1, 3),
        verts_uvs=torch.tensor(uvs, dtype=torch.float32).reshape(-1, 2),
        faces_uvs=torch.tensor(faces_uv, dtype=torch.int64).reshape(-1, 3),
        face_materials=face_mat, material_props=props, material_images=images)


# ---- per-face texture atlas ([P3D] mtl_io.make_mesh_texture_atlas / make_material_atlas) --------------------------
def atlas_barycentrics(R: int) -> torch.Tensor:
    """(R, R, 3) barycentric weights of the atlas cell centres: cell (i, j) = grid (x = j, y = i); below the diagonal
    (x + y < R) (w0, w1) = ((x, y) + 1/3) / R, above it ((R - 1 - (x, y)) + 2/3) / R; w2 = 1 - w0 - w1."""
    rng = torch.arange(R, dtype=torch.float32)
    Y, X = torch.meshgrid(rng, rng, indexing="ij")
    grid = torch.stack([X, Y], dim=-1)
    below = grid.sum(-1) < R
    bary = torch.zeros(R, R, 3)
    lo = (grid + 1.0 / 3.0) / R
    hi = ((R - 1.0 - grid) + 2.0 / 3.0) / R
    bary[..., :2] = torch.where(below[..., None], lo, hi)
    bary[..., 2] = 1.0 - bary[..., :2].sum(-1)
    return bary


def material_atlas(image: torch.Tensor, face_uvs: torch.Tensor, R: int) -> torch.Tensor:
    """image (H, W, 3) already flipped vertically, face_uvs (F, 3, 2) in [0, 1] -> (F, R, R, 3): the image sampled
    bilinearly (align_corners=True) at the cell centres of every face's uv triangle."""
    bary = atlas_barycentrics(R)
    uv = (face_uvs[:, None, None] * bary[None, ..., None]).sum(-2)  # (F, R, R, 2)
    grid = uv * 2.0 - 1.0
    img = image.permute(2, 0, 1)[None]
    out = []
    for s in range(0, uv.shape[0], 65536):  # bounded temporaries for very large models
        g = grid[s:s + 65536]
        smp = torch.nn.functional.grid_sample(img, g.reshape(1, -1, R, 2), mode="bilinear", align_corners=True)
        out.append(smp[0].permute(1, 2, 0).reshape(-1, R, R, 3))
    return torch.cat(out) if out else torch.zeros(0, R, R, 3)


def mesh_texture_atlas(obj: dict, R: int = 4) -> torch.Tensor:
    """(F, R, R, 3) atlas of a ``load_obj_full`` result: 0.5 grey -> material Kd -> map_Kd image where the face has uvs."""
    F = obj["faces"].shape[0]
    atlas = torch.full((F, R, R, 3), 0.5, dtype=torch.float32)
    if F == 0:
        return atlas
    names = obj["face_materials"]
    by_mat: Dict[str, List[int]] = {}
    for i, n in enumerate(names):
        if n is not None:
            by_mat.setdefault(n, []).append(i)
    for name, p in obj["material_props"].items():
        if name in by_mat and "diffuse_color" in p:
            atlas[torch.tensor(by_mat[name])] = p["diffuse_color"][None, None, None, :]
    has_uv = (obj["faces_uvs"] >= 0).all(1) if obj["verts_uvs"].shape[0] else torch.zeros(F, dtype=torch.bool)
    # texture_wrap = "repeat": when ANY uv of the mesh lies outside [0, 1] the integer part of ALL of them is dropped
    # (GL_REPEAT; exact 1.0 then becomes 0.0, as in [P3D]); a mesh that stays inside [0, 1] is left alone
    used = obj["verts_uvs"][obj["faces_uvs"][has_uv]] if bool(has_uv.any()) else torch.zeros(0, 3, 2)
    wrap = bool(((used > 1) | (used < 0)).any())
    for name, path in obj["material_images"].items():
        if name not in by_mat:
            continue
        idx = torch.tensor(by_mat[name])
        idx = idx[has_uv[idx]]
        if idx.numel() == 0:
            continue
        image = read_image(path)
        if image is None:
            continue
        uv = obj["verts_uvs"][obj["faces_uvs"][idx]]
        if wrap:
            uv = uv % 1.0
        atlas[idx] = material_atlas(torch.flip(image, [0]), uv, R)
    return atlas


# ---- dataset ----------------------------------------------------------------------------------------------------------
class ShapeNetCoreDir:
    """``ShapeNetCore(data_dir, synsets=None, version=2, load_textures=True, texture_resolution=4)`` over a directory
    tree, without PyTorch3D (module docstring).  Usable wherever the reference passes its ``shapenet_dataset``
    (``OcclusionEnv(shapenet_dataset)``, /root/reference/trainRL.py:75)."""

    def __init__(self, data_dir: str, synsets=None, version: int = 2, load_textures: bool = True,
                 texture_resolution: int = 4):
        if version not in (1, 2):
            raise ValueError("Version number must be either 1 or 2.")
        if not os.path.isdir(data_dir):
            raise FileNotFoundError(f"ShapeNetCore directory {data_dir} does not exist")
        self.shapenet_dir = data_dir
        self.load_textures = load_textures
        self.texture_resolution = int(texture_resolution)
        self.model_dir = "model.obj" if version == 1 else os.path.join("models", "model_normalized.obj")
        labels = dict(SYNSET_LABELS)
        labels.update(self._taxonomy_labels(data_dir))
        present = sorted(d for d in os.listdir(data_dir) if os.path.isdir(os.path.join(data_dir, d)) and (d in labels or d.isdigit()))
        if synsets is not None:
            inv = {v: k for k, v in labels.items()}
            want = set()
            for s in synsets:
                sid = s if s in labels or s.isdigit() else inv.get(s)
                if sid is None or sid not in present:
                    warnings.warn(f"synset {s} is not in the dataset directory")
                else:
                    want.add(sid)
            present = [d for d in present if d in want]
        self.synset_ids: List[str] = []
        self.model_ids: List[str] = []
        self.synset_dict: Dict[str, str] = {}
        self.synset_start_idxs: Dict[str, int] = {}
        self.synset_num_models: Dict[str, int] = {}
        for sid in present:
            start = len(self.model_ids)
            for model in sorted(os.listdir(os.path.join(data_dir, sid))):
                if not os.path.isfile(os.path.join(data_dir, sid, model, self.model_dir)):
                    if os.path.isdir(os.path.join(data_dir, sid, model)):
                        warnings.warn(f"object file not found in the model directory {model} under synset directory {sid}")
                    continue
                self.synset_ids.append(sid)
                self.model_ids.append(model)
            n = len(self.model_ids) - start
            if n:
                self.synset_dict[sid] = labels.get(sid, sid)
                self.synset_start_idxs[sid] = start
                self.synset_num_models[sid] = n
        self.synset_inv = {label: sid for sid, label in self.synset_dict.items()}

    @staticmethod
    def _taxonomy_labels(root: str) -> Dict[str, str]:
        path = os.path.join(root, "taxonomy.json")
        if not os.path.isfile(path):
            return {}
        try:
            return {e["synsetId"]: e["name"].split(",")[0] for e in json.load(open(path)) if "synsetId" in e and "name" in e}
        except Exception:  # noqa: BLE001
            return {}

    def __len__(self) -> int:
        return len(self.model_ids)

    def __getitem__(self, idx: int) -> dict:
        idx = int(idx)
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        sid, mid = self.synset_ids[idx], self.model_ids[idx]
        obj = load_obj_full(os.path.join(self.shapenet_dir, sid, mid, self.model_dir), load_textures=self.load_textures)
        textures = mesh_texture_atlas(obj, self.texture_resolution) if self.load_textures else None
        return {"synset_id": sid, "model_id": mid, "verts": obj["verts"], "faces": obj["faces"], "textures": textures,
                "label": self.synset_dict[sid]}
