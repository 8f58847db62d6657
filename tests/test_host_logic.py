"""CPU tests of the host side: VecEnv API contract vs fixtures captured from the importable parts of the
reference (tests/golden/make_golden.py), mesh pool / loaders, scene sampling, lazy infos, Box shim."""
import inspect
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "vecenv_golden.npz"))


def test_tile_images_matches_reference_fixture():
    from occlusionenv_amd.baseVecEnv import tile_images

    for n in (1, 4, 5, 7, 9):
        out = tile_images(G[f"tile_in_{n}"])
        assert out.shape == G[f"tile_out_{n}"].shape
        assert np.array_equal(out, G[f"tile_out_{n}"])


def test_vecenv_contract_matches_reference_fixture():
    from occlusionenv_amd import baseVecEnv as B

    assert sorted(B.VecEnv.__abstractmethods__) == list(G["vecenv_abstract"])
    mine = {n for n, _ in inspect.getmembers(B.VecEnv) if not n.startswith("__")}
    assert set(G["vecenv_methods"]) <= mine
    mine_w = {n for n, _ in inspect.getmembers(B.VecEnvWrapper) if not n.startswith("__")}
    assert set(G["wrapper_methods"]) <= mine_w
    assert str(inspect.signature(B.VecEnv.step)) == str(G["step_sig"])
    assert str(inspect.signature(B.VecEnv.__init__)) == str(G["init_sig"])
    assert str(B.AlreadySteppingError()) == str(G["err_already"])
    assert str(B.NotSteppingError()) == str(G["err_not"])


def test_vecenv_step_is_async_plus_wait_and_wrapper_delegates():
    from occlusionenv_amd.baseVecEnv import VecEnv, VecEnvWrapper

    class Dummy(VecEnv):
        def __init__(self):
            VecEnv.__init__(self, 3, "obs", "act")
            self.log = []
            self.special = 7

        def reset(self): return "r"
        def step_async(self, actions): self.log.append(("async", actions))
        def step_wait(self): self.log.append("wait"); return 1, 2, 3, 4
        def close(self): self.log.append("close")
        def get_attr(self, attr_name, indices=None): return [attr_name] * len(list(self._get_indices(indices)))
        def set_attr(self, attr_name, value, indices=None): pass
        def env_method(self, method_name, *a, indices=None, **k): return [method_name]
        def seed(self, seed=None): return [seed]

    d = Dummy()
    assert d.step("a") == (1, 2, 3, 4) and d.log == [("async", "a"), "wait"]
    assert list(d._get_indices(None)) == [0, 1, 2] and d._get_indices(1) == [1] and d._get_indices([0, 2]) == [0, 2]
    assert d.unwrapped is d

    class W(VecEnvWrapper):
        def reset(self): return self.venv.reset()
        def step_wait(self): return self.venv.step_wait()

    w = W(d)
    assert w.num_envs == 3 and w.observation_space == "obs" and w.special == 7 and w.unwrapped is d
    assert w.get_attr("x", [0, 1]) == ["x", "x"] and w.seed(5) == [5]
    with pytest.raises(AttributeError):
        _ = w.does_not_exist


def test_obs_space_helpers_and_box():
    from collections import OrderedDict

    from occlusionenv_amd.spaces import Box
    from occlusionenv_amd.SubProcVecEnv import copy_obs_dict, dict_to_obs, obs_space_info

    b = Box(0, 1, shape=(4, 8, 8))
    assert b.shape == (4, 8, 8) and b.low.shape == (4, 8, 8) and float(b.high.max()) == 1.0
    assert b.contains(b.sample())
    keys, shapes, dtypes = obs_space_info(b)
    assert keys == [None] and shapes[None] == (4, 8, 8)
    assert dict_to_obs(b, {None: 5}) == 5
    d = OrderedDict(a=1)
    assert copy_obs_dict(d) == d and copy_obs_dict(d) is not d
    a = Box(low=-0.1, high=0.1, shape=(2,))  # environment.py:222
    assert a.shape == (2,) and np.allclose(a.low, -0.1) and np.allclose(a.high, 0.1)


def test_obj_loader_variants(tmp_path):
    from occlusionenv_amd.meshes import load_obj

    p = tmp_path / "m.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
                 "f 1//1 2//1 3//1\nf 1/1/1 3/1/1 4/1/1\nf -4 -3 -2 -1\n")
    v, f = load_obj(str(p))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3]]


def test_mesh_pool_packing_and_validation():
    from occlusionenv_amd.meshes import MeshPool

    pool = MeshPool("cpu")
    a = pool.add(torch.zeros(4, 3), torch.tensor([[0, 1, 2], [0, 2, 3]]), key="a")
    b = pool.add(torch.ones(3, 3), torch.tensor([[0, 1, 2]]))
    assert (a, b) == (0, 1) and pool.add(torch.zeros(4, 3), torch.tensor([[0, 1, 2]]), key="a") == 0
    pv, pf, vo, fo = pool.device_tensors()
    assert pv.shape == (7, 3) and pf.shape == (3, 3) and pf.dtype == torch.int32
    assert vo.tolist() == [0, 4, 7] and fo.tolist() == [0, 2, 3] and pool.max_faces == 2
    with pytest.raises(ValueError):
        pool.add(torch.zeros(3, 3), torch.tensor([[0, 1, 3]]))
    with pytest.raises(ValueError):
        pool.add(torch.zeros(3, 2), torch.tensor([[0, 1, 2]]))


def test_synthetic_shapenet_duck_type_and_meshes():
    from occlusionenv_amd.meshes import SyntheticShapeNet, icosphere, torus

    v, f = icosphere(4)
    assert v.shape == (2562, 3) and f.shape == (5120, 3)
    v, f = torus(64, 40)
    assert v.shape == (2560, 3) and f.shape == (5120, 3)
    ds = SyntheticShapeNet(n_models=8, seed=3, mixed=True, n_categories=3)
    assert len(ds) == 8 and sum(ds.synset_num_models.values()) == 8
    for cat, start in ds.synset_start_idxs.items():
        it = ds[start]
        assert set(it) >= {"verts", "faces", "textures", "synset_id", "label", "model_id"} and it["synset_id"] == cat
    for v, f in ds.models:
        # watertight, outward-oriented, unit bbox diagonal
        assert f.shape[0] in (1280, 5120, 20480)
        assert abs(float((v.max(0).values - v.min(0).values).norm()) - 1.0) < 1e-5
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        assert float((a * torch.cross(b, c, dim=1)).sum()) > 0
    ds2 = SyntheticShapeNet(n_models=8, seed=3, mixed=True, n_categories=3)
    assert all(torch.equal(x[0], y[0]) for x, y in zip(ds.models, ds2.models))


def test_sample_scene_follows_reference_layout():
    from occlusionenv_amd.environment import sample_scene
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet

    ds = SyntheticShapeNet(n_models=6, seed=1)
    pool = MeshPool("cpu")
    np.random.seed(11)
    x2_expected = np.random.randn()
    np.random.seed(11)
    ids, offs = sample_scene(ds, pool)
    assert len(ids) == 3 and all(0 <= i < len(pool) for i in ids)
    # environment.py:148,171: (x2, 0, distance/2) and (-x2, 0, distance), distance = 2
    assert offs[0] == [0.0, 0.0, 0.0] and offs[1] == [x2_expected, 0.0, 1.0] and offs[2] == [-x2_expected, 0.0, 2.0]
    ids_t, offs_t = sample_scene(None, pool)  # default scene: three teapots
    assert len(set(ids_t)) == 1 and pool.num_faces(ids_t[0]) == 2464


def test_lazy_infos_behaves_like_list_of_dicts():
    from occlusionenv_amd.SubProcVecEnv import _LazyInfos

    pos = torch.arange(12.0).reshape(4, 3)
    fs, loss = torch.zeros(4, 2, 2, 4), torch.arange(4.0)
    infos = _LazyInfos(pos, fs, loss)
    assert len(infos) == 4 and set(infos[1]) == {"full_state", "position", "full_reward"}
    assert infos[1]["position"].tolist() == [3.0, 4.0, 5.0]
    assert infos[2]["full_state"].shape == (1, 2, 2, 4) and float(infos[3]["full_reward"]) == 3.0
    infos.set(1, "terminal_observation", "x")
    assert infos[1]["terminal_observation"] == "x" and "terminal_observation" not in infos[0]
    assert [float(d["full_reward"]) for d in infos] == [0.0, 1.0, 2.0, 3.0]
    with pytest.raises(IndexError):
        infos[4]


def test_product_path_fails_loudly_without_gpu_or_extension(monkeypatch, tmp_path):
    from occlusionenv_amd import _native
    from occlusionenv_amd.engine import OcclusionEngine
    from occlusionenv_amd.meshes import MeshPool

    if not torch.cuda.is_available():
        pool = MeshPool("cpu")
        pool.add(torch.zeros(3, 3), torch.tensor([[0, 1, 2]]))
        with pytest.raises(_native.NativeError):
            OcclusionEngine(pool, 1, 64)
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_native.NativeError):
        _native.load()


def test_product_never_imports_oracle():
    import re

    pkg = os.path.join(ROOT, "occlusionenv_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn
                assert "liborc" not in src, fn


def test_root_aliases_export_reference_names():
    import baseVecEnv
    import environment
    import SubProcVecEnv

    assert hasattr(environment, "OcclusionEnv") and hasattr(SubProcVecEnv, "SimpleVecEnv")
    for n in ("VecEnv", "VecEnvWrapper", "tile_images", "CloudpickleWrapper", "AlreadySteppingError", "NotSteppingError"):
        assert hasattr(baseVecEnv, n)
    sig = inspect.signature(environment.OcclusionEnv.__init__)
    assert list(sig.parameters) == ["self", "data", "img_size"] and sig.parameters["img_size"].default == 512
    sig = inspect.signature(environment.OcclusionEnv.reset)
    assert [(k, v.default) for k, v in list(sig.parameters.items())[1:]] == [
        ("new_scene", True), ("radius", 4.0), ("azimuth", 0.0), ("elevation", 0.0)]
    assert list(inspect.signature(SubProcVecEnv.SimpleVecEnv.__init__).parameters) == ["self", "env_fns"]


def test_mesh_pool_incremental_appends_keep_earlier_data():
    """Meshes are appended for as long as the env runs: each flush uploads the new meshes only, earlier data and
    offsets stay put, capacity growth copies what is already there."""
    from occlusionenv_amd.meshes import MeshPool, icosphere

    pool = MeshPool("cpu")
    v, f = icosphere(2)
    v = torch.as_tensor(v, dtype=torch.float32)
    f = torch.as_tensor(f)
    snapshots = []
    for i in range(120):  # 120 x 162 verts crosses the initial capacity of 16384 rows
        mid = pool.add(v + float(i), f, key=i, atlas=(torch.full((f.shape[0], 2, 2, 3), float(i)) if i % 3 == 0 else None))
        assert mid == i
        if i % 17 == 0 or i == 119:
            pv, pf, vo, fo = pool.device_tensors()
            assert len(vo) == i + 2 and int(vo[-1]) == (i + 1) * v.shape[0] and int(fo[-1]) == (i + 1) * f.shape[0]
            snapshots.append(i)
    pv, pf, vo, fo = pool.device_tensors()
    for i in (0, 16, 17, 100, 119):
        assert torch.equal(pv[vo[i]:vo[i + 1]], v + float(i)) and torch.equal(pf[fo[i]:fo[i + 1]].long(), f)
    atlas, aoff = pool.atlas_tensors()
    assert int(aoff[1]) == -1 and int(aoff[3]) == f.shape[0] * 12 and float(atlas[int(aoff[117])]) == 117.0
    assert pool.max_faces == f.shape[0] and pool.version == 120


def test_oversize_models_never_enter_the_pool(monkeypatch):
    """environment.py:296-298 discards scenes with a mesh above 250 000 faces; here such a model is rejected before it
    reaches the pool (the pool's largest mesh sizes the render workspace)."""
    from occlusionenv_amd import environment
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet

    ds = SyntheticShapeNet(n_models=6, seed=3, mixed=True)
    sizes = sorted({int(m[1].shape[0]) for m in ds.models})
    assert len(sizes) >= 2
    monkeypatch.setattr(environment, "MAX_MESH_FACES", sizes[0])
    environment._OVERSIZE.clear()
    environment.seed_scene_rng(0)
    pool = MeshPool("cpu")
    ok = rejected = 0
    for _ in range(200):
        try:
            ids, offs = environment.sample_scene(ds, pool)
            ok += 1
            assert all(pool.num_faces(m) <= sizes[0] for m in ids)
        except ValueError:
            rejected += 1
    assert ok > 0 and rejected > 0 and pool.max_faces <= sizes[0]
    environment._OVERSIZE.clear()


def test_h1_record_layout_matches_reference_fullnetwork_shapes():
    """SURVEY §8a row H1 pin: the rollout record's feature block and the harness's batched / single-env conventions
    match what the reference's own agent produces from an observation.  Shapes captured by importing
    /root/reference/model.py (FullNetwork(8, dilation=2, separable=True), PPO.py:47,155-162) in the authoring
    container: tests/golden/make_golden.py."""
    import ast

    import torch

    from occlusionenv_amd import rollout

    assert int(G["h1_param_count"]) == 1020902
    cases = ast.literal_eval(str(G["h1_shapes"]))
    assert [c[0] for c in cases] == [[1, 4, 64, 64], [1, 4, 128, 128], [8, 4, 128, 128]]
    for obs_shape, feats, segm, gradp, act, val in cases:
        n, _, S, _ = obs_shape
        obs = torch.rand(*obs_shape)
        pooled = rollout.pooled_features(obs)
        # the reference squeezes the batch axis of a single observation (model.py:162); the record keeps (n, 256)
        assert list(pooled.shape) == [n, 256] and feats == ([256] if n == 1 else [n, 256])
        assert segm == [n, 1, S, S] and gradp == ([2] if n == 1 else [n, 2])  # full-res segmentation, 2-d gradient head
        assert act == ([2] if n == 1 else [n, 2]) and val == ([1] if n == 1 else [n, 1])
        rec = rollout.pack_records(obs, torch.zeros(n, 2), torch.zeros(n), torch.zeros(n), torch.zeros(n, dtype=torch.bool))
        # 256 features + action (action_scores' width) + logprob + reward + done (PPO.py:157-162, trainRL.py:203-204)
        assert rec.shape == (n, rollout.RECORD_FLOATS) and rollout.RECORD_FLOATS == feats[-1] + act[-1] + 1 + 1 + 1


def test_obs_space_helpers_cover_plain_dict_and_tuple_spaces():
    """obs_space_info / dict_to_obs / copy_obs_dict (interface of /root/reference/SubProcVecEnv.py:11-70)."""
    from collections import OrderedDict

    from occlusionenv_amd.spaces import Box
    from occlusionenv_amd.SubProcVecEnv import copy_obs_dict, dict_to_obs, obs_space_info

    box = Box(0.0, 1.0, (4, 8, 8))
    keys, shapes, dtypes = obs_space_info(box)
    assert keys == [None] and shapes[None] == (4, 8, 8) and dtypes[None] == box.dtype
    assert dict_to_obs(box, {None: "arr"}) == "arr"

    class Dict:
        def __init__(self, spaces):
            self.spaces = OrderedDict(spaces)

    class Tuple:
        def __init__(self, spaces):
            self.spaces = tuple(spaces)

    small = Box(-1.0, 1.0, (2,))
    d = Dict([("img", box), ("vec", small)])
    keys, shapes, _ = obs_space_info(d)
    assert keys == ["img", "vec"] and shapes == {"img": (4, 8, 8), "vec": (2,)}
    buf = OrderedDict([("img", 1), ("vec", 2)])
    assert dict_to_obs(d, buf) is buf
    t = Tuple([box, small])
    keys, shapes, _ = obs_space_info(t)
    assert keys == [0, 1] and shapes[1] == (2,)
    assert dict_to_obs(t, {0: "a", 1: "b"}) == ("a", "b")
    with pytest.raises(AssertionError):
        dict_to_obs(t, {0: "a"})
    with pytest.raises(AssertionError):
        dict_to_obs(box, {None: 1, "x": 2})
    c = copy_obs_dict(buf)
    assert c == buf and c is not buf
    with pytest.raises(AssertionError):
        copy_obs_dict({"img": 1})


def test_tie_classifier_bands_are_frozen():
    """The near-tie classifier is what stands between a real bug and a green parity test: its bands may be
    tightened, never loosened, without this test being changed on purpose."""
    from tests import parity_utils as pu

    frozen = dict(TOL=1e-4, TZ_REL=1e-6, TB_REL=1e-4, TPAIR_REL=1e-4, TEDGE=5e-7, TAREA=2e-9, TVERT=2.5e-7, TVIEW=5e-7, TTEXEL=1e-3,
                  GRAD_NOISE_ULPS=256.0)
    for name, bound in frozen.items():
        assert 0 < getattr(pu, name) <= bound, (name, getattr(pu, name), bound)
    # round 5: the gradient's own near-tie rules (tests/grad_explain.py) and the measured noise floor of the arbiter
    from tests import grad_explain as gx

    assert 0 < pu.GRAD_ORC32_FACTOR <= 2.0
    assert 0 < gx.TCENTRE <= 1.2e-7 and 0 < gx.TGRAD_PIX <= 1e-3 and gx.TGRAD_SHARE >= 0.1 and 0 < gx.GRAD_TIE_FRAC <= 2e-4
    for img in (64, 128, 256, 512):
        assert pu.max_grad_tie_pixels(img) <= max(4, int(2e-4 * img * img * 3))
    # the per-env budgets of accepted pixels: tie decisions, pixels under the near / z-clipped rule, and the footprint
    # allowance of hair-flipped needles (these three were still loose)
    assert 0 < pu.TIE_FRAC <= 2e-4 and 0 < pu.UPSTREAM_FRAC <= 2e-3 and 0 < pu.FOOTPRINT_FACTOR <= 8
    for img in (64, 128, 256, 512):
        assert pu.max_tie_pixels(img) <= max(4, int(2e-4 * img * img * 3))
        assert pu.max_upstream_pixels(img) <= max(8, int(2e-3 * img * img * 3))
    # violations() applies exactly these budgets
    res = dict(unexplained=[], tie_pixels=8 * pu.max_tie_pixels(64) + pu.max_upstream_pixels(64) + 1, tie_decisions=0, img=64,
               upstream_pixels=0, fs_arith=0.0, grad_excess=0.0, grad_arbiter=[],
               **{k: 0.0 for k in ("obs_maxabs", "obs0_maxabs", "alpha_maxabs", "alpha0_maxabs", "fs_maxabs", "render_maxabs",
                                   "loss_rel", "loss0_rel", "reward_abs")})
    assert any("too many tie pixels" in v for v in pu.violations(res))
    res["tie_pixels"] -= 1
    assert not pu.violations(res)
    res["upstream_pixels"] = pu.max_upstream_pixels(64) + 1
    assert any("near / z-clipped" in v for v in pu.violations(res))


def test_hair_band_scales_with_the_perimeter_and_needle_depth_bound():
    """The two conditioning rules of the classifier on the faces that motivated them (parity sweep seeds 2352, 2084):
    a needle's area moves by vertex noise x perimeter, its barycentrics by that / area."""
    from oracle import p3d_restate as O
    from tests import parity_utils as pu

    # like seed 2352, object 1, face 664 (signed area 7.5e-9: culled by the oracle, kept by the engine at 1.3e-8;
    # perimeter 0.18): a needle 0.09 long and 8e-8 high
    fv = torch.tensor([[[0.02, 0.0, 2.22], [0.11, 0.0, 2.23], [0.065, -8.33e-8, 2.21]]])
    x, y = fv[0, :, 0].double(), fv[0, :, 1].double()
    area = float((x[2] - x[0]) * (y[1] - y[0]) - (y[2] - y[0]) * (x[1] - x[0]))
    assert 7e-9 < area < 8e-9 and abs(area - 1e-8) > pu.TAREA
    S, yi, xi = 64, 31, 30
    plain = O.pixel_candidates(fv, S, yi, xi, O.BLUR_RADIUS, band=1e-3, area_band=pu.TAREA)
    wide = O.pixel_candidates(fv, S, yi, xi, O.BLUR_RADIUS, band=1e-3, area_band=pu.TAREA, vert_band=pu.TVERT)
    assert plain["f"].size == 0  # culled, and not within the plain band
    assert wide["f"].size == 1 and (wide["flags"] & 8).all()
    # a compact face of the same area class is NOT given the wide band: perimeter 3e-4 adds 7.5e-11
    small = torch.tensor([[[0.0, 0.0, 2.0], [0.0, 1e-4, 2.0], [-1.3e-4, 0.0, 2.0]]])
    c = O.pixel_candidates(small, S, 32, 32, O.BLUR_RADIUS, band=1e-3, area_band=pu.TAREA, vert_band=pu.TVERT)
    assert not (c["flags"] & 8).any()
    # seed 2084, scene face 5797: area 1.05e-5, edges 0.159 / 0.095 / 0.064, depth range 0.106 -> bound ~1.6e-3;
    # an ordinary face (area 1e-3, edges 0.05) stays far below TOL
    needle = np.array([[0.0, 0.0, 1.566], [0.159, 0.0, 1.488], [0.095, 6.6e-5, 1.459]])
    b = pu.sliver_depth_bound(needle)
    assert 5e-4 < b < 5e-3
    fat = np.array([[0.0, 0.0, 1.5], [0.05, 0.0, 1.52], [0.0, 0.04, 1.48]])
    assert pu.sliver_depth_bound(fat) < 0.05 * pu.TOL


def test_vertex_noise_bounds_and_record_matching():
    """The upstream rule of the classifier (parity_utils: NEAR AND Z-CLIPPED FACES) on the oracle's own geometry: a far
    face keeps the flat TVERT, a near one grows with 1 / z, a face cut by the clip plane along an edge that runs nearly
    parallel to it grows with 1 / |z_a - z_b|; records made from the oracle's clipped faces match within the bound with
    noise below it and fail above it - also for a lone half of a split face that carries no pair flag."""
    from oracle import p3d_restate as O
    from tests import parity_utils as pu

    s = float(O.proj_scale())
    def ndc(view):  # (3,3) view-space -> (x_ndc, y_ndc, z_view)
        v = torch.tensor(view, dtype=torch.float32)
        return torch.stack([v[:, 0] * s / v[:, 2], v[:, 1] * s / v[:, 2], v[:, 2]], 1)

    far = ndc([[0.0, 0.0, 4.0], [0.3, 0.0, 4.1], [0.0, 0.3, 4.0]])
    near = ndc([[0.0, 0.0, 0.6], [0.3, 0.0, 0.7], [0.0, 0.3, 0.6]])
    cut1 = ndc([[0.0, 0.0, 0.45], [0.4, 0.0, 0.9], [0.0, 0.4, 0.9]])       # one vertex behind: split in two
    flat = ndc([[0.0, 0.0, 0.4995], [0.5, 0.0, 0.5005], [0.0, 0.4, 0.9]])   # an edge almost inside the clip plane
    fv = torch.stack([far, near, cut1, flat])
    b = pu.face_noise_bounds(fv)
    assert b[0] == pu.TVERT
    assert pu.TVERT < b[1] < 4 * pu.TVERT * 4.0 / 0.6
    assert b[2] > b[1] and b[3] > 50 * b[2]

    class F:
        pass

    faces = F()
    faces.fv_unclipped = fv
    faces.fv, faces.c2u, faces.nb, _, _ = O.clip_faces(fv, O.Z_CLIP, True)
    c2u = faces.c2u.numpy()
    assert c2u.tolist().count(2) == 2 and c2u.tolist().count(3) == 2   # both straddling faces: one vertex behind -> two pieces
    flags = np.zeros(len(c2u), dtype=np.int32)
    for k in (2, 3):
        i0 = int(np.nonzero(c2u == k)[0][0])
        flags[i0], flags[i0 + 1] = 1 | 4, 2 | 4
    rec = dict(fv=faces.fv.clone(), ids=c2u.astype(np.int32), flags=flags)
    ok, worst, _ = pu.upstream_check(faces, rec)
    assert ok and worst == 0.0
    noisy = dict(rec, fv=faces.fv.clone())
    noisy["fv"][0, 0, 0] += 0.5 * pu.TVERT
    assert pu.upstream_check(faces, noisy)[0]
    noisy["fv"][0, 0, 0] += 2.0 * pu.TVERT
    assert not pu.upstream_check(faces, noisy)[0]
    # the second half of face 2 alone, without pair flags (its partner invisible): matched to whichever piece it is
    i1 = int(np.nonzero(c2u == 2)[0][1])
    lone = dict(fv=faces.fv[i1:i1 + 1].clone(), ids=np.array([2], dtype=np.int32), flags=np.array([4], dtype=np.int32))
    assert pu.upstream_check(faces, lone)[0]
    lone["ids"][0] = 0   # ... but not to another face
    assert not pu.upstream_check(faces, lone)[0]


def test_pool_keys_survive_dataset_address_reuse():
    """The shared mesh pool outlives datasets: a new dataset object that is handed the address of a collected one
    (observed as a flaky GPU test: cubes rendered as the previous test's white meshes) must not be served its entries."""
    import gc

    from occlusionenv_amd import environment
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet

    pool = MeshPool("cpu")
    seen = {}
    for rep in range(6):
        ds = SyntheticShapeNet(n_models=2, seed=100 + rep)
        tok = environment._dataset_token(ds)
        assert tok == environment._dataset_token(ds)        # stable while the object lives
        assert tok not in seen.values()                      # never handed out twice, whatever id() does
        seen[id(ds)] = tok
        environment.seed_scene_rng(rep)
        ids, _ = environment.sample_scene(ds, pool)
        for m in ids:
            v, _f = pool.get(m)
            assert any(torch.equal(v, mv) for mv, _ in ds.models)  # the pool entry is THIS dataset's model
        del ds
        gc.collect()
    environment.seed_scene_rng(None)


def test_mixed_texture_scenes_are_drawn_again_like_the_reference():
    """environment.py:126-129 wraps a textured model in TexturesAtlas and an untextured one in TexturesVertex; [P3D]
    join_meshes_as_scene (:191) raises on a mix and reset()'s bare except draws again (:329-330): a scene whose three
    models do not share the texture type never reaches the renderer.  sample_scene raises ValueError for such a draw -
    every caller (OcclusionEnv._new_scene, SimpleVecEnv._refill_reserve) treats that as "draw again" - so all scenes
    that ARE returned are uniformly textured or uniformly white."""
    import numpy as np
    import torch

    from occlusionenv_amd import environment
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet

    ds = SyntheticShapeNet(n_models=8, seed=5, textured=True)
    for i in (1, 4, 6):
        ds.atlases[i] = None  # three of the eight models have no textures
    pool = MeshPool("cpu")
    environment.seed_scene_rng(11)
    np.random.seed(11)
    kinds, rejected = set(), 0
    for _ in range(200):
        try:
            ids, _ = environment.sample_scene(ds, pool)
        except ValueError as e:
            assert "textured and untextured" in str(e)
            rejected += 1
            continue
        tex = {pool.get_atlas(m) is not None for m in ids}
        assert len(tex) == 1
        kinds |= tex
    assert kinds == {True, False} and rejected > 50  # both uniform kinds occur; most draws of this pool are mixed
    environment.seed_scene_rng(None)


def test_closest_edge_tie_rule_on_a_corner_bisector():
    """tests/grad_explain.py, CLOSEST-EDGE TIE, on the configuration that motivated it (parity sweep seed 9276): one large
    face, a pixel centre on the bisector of its corner at v0 - the squared distances to edges (v0,v1) and (v0,v2) are equal
    there and the gradient of dists jumps between the two edges' normals.  The rule fires on the bisector (and within the
    positional noise of it), not a pixel away, and the forward sweep's d alpha on either side differs by the jump."""
    import numpy as np

    from tests import grad_explain as gx

    S = 64
    pitch = 2.0 / S

    def centre(i):
        return -1.0 + (2.0 * (S - 1 - i) + 1.0) / S

    yi, xi = 30, 40
    px, py = centre(xi), centre(yi)
    # corner v0 below-left of the pixel, edges at +-30 degrees around the direction to the pixel: the pixel is ON the bisector
    d = 0.6 * pitch  # (close to the corner: deeper inside, 1 - p underflows and there is no gradient to compare)
    v0 = np.array([px - d * np.cos(0.3), py - d * np.sin(0.3)])
    e1 = np.array([np.cos(0.3 + 0.5), np.sin(0.3 + 0.5)])
    e2 = np.array([np.cos(0.3 - 0.5), np.sin(0.3 - 0.5)])
    fv = np.zeros((1, 3, 3))
    fv[0, :, 2] = 1.0
    fv[0, 0, :2], fv[0, 1, :2], fv[0, 2, :2] = v0, v0 + 0.6 * e1, v0 + 0.6 * e2
    tan = np.zeros((2, 1, 3, 3))
    tan[0, 0, :, 0] = 1.0  # every vertex moves along +x with el, along +y with az
    tan[1, 0, :, 1] = 1.0
    p2f = np.full((S, S, 1), -1, dtype=np.int64)
    p2f[yi, xi, 0] = 0
    p2f[yi, xi + 3, 0] = 0  # three pixels to the side: clearly nearer to one edge
    _, dal, tie = gx.forward_planes(fv, tan, p2f, S)
    assert bool(tie[yi, xi]) and not bool(tie[yi, xi + 3]) and int(tie.sum()) == 1
    # a hair off the bisector - less than the positional noise - still ties; the two sides' gradients differ by the jump
    for shift, expect in ((5e-8, True), (1e-5, False)):
        fv2 = fv.copy()
        fv2[0, :, :2] += shift * np.array([-np.sin(0.3), np.cos(0.3)])  # across the bisector
        _, _, t2 = gx.forward_planes(fv2, tan, p2f, S)
        assert bool(t2[yi, xi]) == expect, shift
    up, dn = fv.copy(), fv.copy()
    up[0, :, :2] += 1e-5 * np.array([-np.sin(0.3), np.cos(0.3)])
    dn[0, :, :2] -= 1e-5 * np.array([-np.sin(0.3), np.cos(0.3)])
    da = gx.forward_planes(up, tan, p2f, S)[1][:, yi, xi]
    db = gx.forward_planes(dn, tan, p2f, S)[1][:, yi, xi]
    assert np.abs(da - db).max() > 0.2 * max(np.abs(da).max(), np.abs(db).max())  # a jump, not a slope
