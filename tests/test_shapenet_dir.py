"""Directory-backed ShapeNetCore reader (occlusionenv_amd/shapenet.py) on a synthetic two-synset tree written by the
test: the duck type ``load_shapenet_meshes`` consumes (/root/reference/environment.py:106-135) and the per-face
texture atlas of ``ShapeNetCore(dir, version=2)`` (/root/reference/trainRL.py:66-71)."""
import os

import numpy as np
import pytest
import torch

from occlusionenv_amd import shapenet

CUBE_V = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
CUBE_Q = [(1, 2, 3, 4), (5, 8, 7, 6), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 8, 4), (5, 1, 4, 8)]  # quads, 1-based


def _write_model(root, synset, model, version=2, textured=True, scale=0.3):
    d = os.path.join(root, synset, model, "models" if version == 2 else "")
    os.makedirs(d, exist_ok=True)
    name = "model_normalized" if version == 2 else "model"
    lines = ["mtllib %s.mtl" % name]
    lines += ["v %f %f %f" % tuple(scale * c for c in v) for v in CUBE_V]
    lines += ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1"]
    lines.append("usemtl red")          # Kd only
    lines.append("f 1/1 2/2 3/3 4/4")   # a quad -> two triangles
    lines.append("usemtl checker" if textured else "usemtl red")
    lines.append("f 5/1 8/4 7/3 6/2")
    lines.append("f -8//1 -4//1 -3//1 -7//1")  # negative indices, v//vn form: no uvs
    lines.append("usemtl nowhere")      # not in the .mtl: stays grey
    lines += ["f %d %d %d %d" % q for q in CUBE_Q[3:]]
    open(os.path.join(d, name + ".obj"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(d, name + ".mtl"), "w").write(
        "newmtl red\nKd 1.0 0.0 0.25\nKa 0 0 0\nNs 10\n\nnewmtl checker\nKd 0.2 0.2 0.2\nmap_Kd tex image.png\n")
    from PIL import Image

    img = np.zeros((2, 2, 3), dtype=np.uint8)
    img[0, 0] = (255, 0, 0)      # top-left
    img[0, 1] = (0, 255, 0)      # top-right
    img[1, 0] = (0, 0, 255)      # bottom-left
    img[1, 1] = (255, 255, 255)  # bottom-right
    Image.fromarray(img).save(os.path.join(d, "tex image.png"))


@pytest.fixture()
def tree(tmp_path):
    root = str(tmp_path / "shapenetcore")
    _write_model(root, "02691156", "aaa")
    _write_model(root, "02691156", "bbb", textured=False)
    _write_model(root, "03001627", "ccc", scale=0.2)
    os.makedirs(os.path.join(root, "03001627", "broken"))  # a model directory without its .obj: skipped
    os.makedirs(os.path.join(root, "not_a_synset"))
    return root


def test_duck_type_contract(tree):
    with pytest.warns(UserWarning, match="object file not found"):
        ds = shapenet.ShapeNetCoreDir(tree, version=2)
    assert len(ds) == 3
    assert list(ds.synset_dict.items()) == [("02691156", "airplane"), ("03001627", "chair")]
    assert ds.synset_inv == {"airplane": "02691156", "chair": "03001627"}
    assert ds.synset_start_idxs == {"02691156": 0, "03001627": 2} and ds.synset_num_models == {"02691156": 2, "03001627": 1}
    it = ds[0]
    assert set(it) == {"verts", "faces", "textures", "synset_id", "model_id", "label"}
    assert (it["synset_id"], it["model_id"], it["label"]) == ("02691156", "aaa", "airplane")
    assert it["verts"].shape == (8, 3) and it["verts"].dtype == torch.float32
    assert it["faces"].shape == (12, 3) and it["faces"].dtype == torch.int64  # six quads, fan-triangulated
    assert it["textures"].shape == (12, 4, 4, 3) and it["textures"].dtype == torch.float32  # texture_resolution = 4
    assert ds[2]["model_id"] == "ccc" and float(ds[2]["verts"].abs().max()) == pytest.approx(0.2)
    with pytest.raises(IndexError):
        ds[3]
    # fan triangulation (v0, v_k, v_k+1) and negative indices (-8 == vertex 1 of 8)
    assert it["faces"][0].tolist() == [0, 1, 2] and it["faces"][1].tolist() == [0, 2, 3]
    assert it["faces"][4].tolist() == [0, 4, 5] and it["faces"][5].tolist() == [0, 5, 1]


def test_texture_atlas_known_answers(tree):
    ds = shapenet.ShapeNetCoreDir(tree, version=2, texture_resolution=2)
    at = ds[0]["textures"]
    assert at.shape == (12, 2, 2, 3)
    # faces 0,1: material with Kd only -> its diffuse colour everywhere
    assert torch.allclose(at[:2], torch.tensor([1.0, 0.0, 0.25]).expand(2, 2, 2, 3))
    # faces 4,5 use the textured material but carry no vt indices -> Kd of that material; faces 6.. unknown material -> grey
    assert torch.allclose(at[4:6], torch.full((2, 2, 2, 3), 0.2))
    assert torch.allclose(at[6:], torch.full((6, 2, 2, 3), 0.5))
    # faces 2,3: map_Kd sampled at the cell centres of the uv triangle.  Face 2 = uv (0,0), (0,1), (1,1);
    # cell (row i = y, col j = x): below the diagonal (x + y < R) w = ((x, y) + 1/3) / R, else mirrored.
    bary = shapenet.atlas_barycentrics(2)
    assert torch.allclose(bary[0, 0], torch.tensor([1 / 6, 1 / 6, 2 / 3]))
    assert torch.allclose(bary[0, 1], torch.tensor([2 / 3, 1 / 6, 1 / 6]))
    assert torch.allclose(bary[1, 1], torch.tensor([1 / 3, 1 / 3, 1 / 3]))
    assert torch.allclose(bary.sum(-1), torch.ones(2, 2))
    uv_tri = torch.tensor([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    # the image file's top row is red, green; bottom row blue, white; uv origin is the bottom-left corner
    corner = {(0, 0): (0, 0, 1), (1, 0): (1, 1, 1), (0, 1): (1, 0, 0), (1, 1): (0, 1, 0)}

    def bilinear(u, v):
        c = np.zeros(3)
        for (cu, cv), col in corner.items():
            c += (u if cu else 1 - u) * (v if cv else 1 - v) * np.asarray(col, dtype=np.float64)
        return c

    for i in range(2):
        for j in range(2):
            u, v = (bary[i, j][:, None] * uv_tri).sum(0).tolist()
            assert np.allclose(at[2, i, j].numpy(), bilinear(u, v), atol=1e-6), (i, j)
    assert not torch.allclose(at[2], at[3])  # the second triangle of the quad covers the other half of the image


def test_version_1_layout_synset_filter_and_no_textures(tmp_path):
    root = str(tmp_path / "v1")
    _write_model(root, "04379243", "t1", version=1)
    _write_model(root, "02958343", "c1", version=1)
    ds = shapenet.ShapeNetCoreDir(root, version=1, load_textures=False)
    assert list(ds.synset_dict) == ["02958343", "04379243"] and ds[0]["textures"] is None and ds[0]["label"] == "car"
    only = shapenet.ShapeNetCoreDir(root, synsets=["table"], version=1)
    assert list(only.synset_dict.values()) == ["table"] and len(only) == 1
    with pytest.raises(ValueError):
        shapenet.ShapeNetCoreDir(root, version=3)
    with pytest.raises(FileNotFoundError):
        shapenet.ShapeNetCoreDir(str(tmp_path / "missing"))


def test_taxonomy_json_overrides_labels(tree):
    import json

    json.dump([{"synsetId": "03001627", "name": "chair,seat", "children": [], "numInstances": 1}], open(os.path.join(tree, "taxonomy.json"), "w"))
    ds = shapenet.ShapeNetCoreDir(tree)
    assert ds.synset_dict["03001627"] == "chair"


def test_scene_sampler_consumes_the_directory_dataset(tree):
    """load_shapenet_meshes' draw (environment.py:102-135) through the build's sampler: models enter the mesh pool with
    their atlases."""
    from occlusionenv_amd import environment
    from occlusionenv_amd.meshes import MeshPool

    ds = shapenet.ShapeNetCoreDir(tree)
    pool = MeshPool("cpu")
    environment.seed_scene_rng(3)
    ids, offs = environment.sample_scene(ds, pool)
    assert len(ids) == 3 and offs[0] == [0.0, 0.0, 0.0] and offs[1][2] == 1.0 and offs[2][2] == 2.0 and offs[1][0] == -offs[2][0]
    for m in ids:
        v, f = pool.get(m)
        assert v.shape == (8, 3) and f.shape == (12, 3) and pool.get_atlas(m).shape == (12, 4, 4, 3)
    environment.seed_scene_rng(None)


def test_uv_wrap_only_when_some_uv_leaves_the_unit_square():
    obj = dict(faces=torch.tensor([[0, 1, 2]]), verts_uvs=torch.tensor([[0.25, 0.25], [1.25, 0.25], [0.25, 1.25]]),
               faces_uvs=torch.tensor([[0, 1, 2]]), face_materials=["m"], material_props={"m": {}}, material_images={})
    # no image: nothing to sample, the atlas stays grey whatever the uvs
    assert torch.allclose(shapenet.mesh_texture_atlas(obj, 2), torch.full((1, 2, 2, 3), 0.5))
