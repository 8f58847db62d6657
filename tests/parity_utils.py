"""Shared helpers of the GPU parity tests, smoke(), scripts/parity_sweep.py and bench.py's cpu_baseline leg: run the
same seeded scenes through the HIP engine and through the CPU oracle (oracle/p3d_restate.py) and compare.
This is checker code: it is the only place (with tests/) where oracle/ and the product meet.

Tolerance: 1e-4 fp32 (BASELINE.json north_star) on EVERY pixel of the observation, the three silhouette alphas and
the occlusion image, on loss and reward (relative to max(1, |value|) for the occlusion image, 0 ... 3, and the reward,
which is a loss of ~100 when objectMass = 1), and relative 1e-4 (L2) on d reward / d action -- with documented
exceptions, checked pixel by pixel instead of masked wholesale:

EXACT TIES.  The oracle and the HIP path reach a pixel with vertex coordinates that differ in the last bits (torch's
CPU matmul / libm vs the setup kernel's own arithmetic), so a DISCRETE decision can fall either way when it is
decided by less than rounding: which of the K-th / (K+1)-th nearest faces of a pixel is kept (depths equal to a few
ulp), whether a face sits inside the blur radius (|d - blur| < 1e-4 blur), which half of a z-clipped pair is closer,
whether a pixel centre lies on a face edge (hard pass), which of two coincident faces is nearer (hard pass), whether
an edge-on sliver's signed area is above the kEpsilon visibility threshold (the area of a long sliver moves by
vertex noise x perimeter: band TAREA + TVERT * perimeter), which texel cell a barycentric on a cell boundary picks.  Every pixel beyond tolerance is handed to the oracle's per-pixel
candidate dump (raster_naive.c: orc_pixel_candidates) and must be EXPLAINED by one of those near-ties; anything
unexplained fails the test.  Explained pixels are few (bounded below), get weight 0 in the loss on BOTH sides
(OccScene.pix_weight / OracleEnv.pixel_weight) and loss, reward and gradient are then compared at full tolerance
over all remaining pixels.

SLIVER DEPTH.  One continuous quantity is ill-conditioned rather than tied: the depth channel of a pixel whose visible
face is a needle (area << perimeter^2).  Barycentrics are edge functions divided by the area, so vertex noise delta moves
them by up to 2 delta perimeter / |area| and the interpolated depth by that times the face's depth range; such a pixel
is accepted only while its depth error stays within TOL + that first-order bound AND its colour is within TOL
(parity sweep seed 2084: a 0.004-pixel-wide needle over a pixel centre, depth off by 1.9e-4, bound 1.6e-3).

NEAR AND Z-CLIPPED FACES.  The projection divides by the view depth: a view-space coordinate noise TVIEW becomes
TVIEW (s + |x_ndc|) / z in NDC (s = 1 / tan 30 deg), and the cut vertex of a face that straddles the clip plane
z = 0.5 moves by (TVIEW / 0.5) (s + 2 |s (X_b - X_a)| / |z_a - z_b|) - unbounded for an edge parallel to the plane.
With the camera inside an object (radius 2.5 in the wide sweep) such faces move an alpha by up to 6e-4 with nothing
tied.  A pixel beyond tolerance is then accepted only if ALL of this holds, machine-checked: (1) every face record of
the engine's setup kernel for that object equals the oracle's (clipped) face within the bound above (vertex by vertex);
(2) the ORACLE's rasteriser run on the ENGINE's own records reproduces the engine's alpha at that pixel within TOL
(i.e. the raster stage agrees on identical geometry); (3) a face whose bound exceeds TVERT is a candidate there.

ACTION GRADIENT.  d reward / d action is a sum of ~1e5 signed fp32 per-pixel terms (2 I dI/d alpha_o * d alpha_o/d theta,
each itself -(A/sigma) * a sum of ~100 signed terms).  When the pixel terms nearly cancel (|g| is 1/30 ... 1/200 of
their L1 mass M in a fair share of scenes) no fp32 evaluation can deliver 1e-4 of the RESULT: measured against the
oracle's f64 build, the f32 oracle (torch autograd + the C backward) is off by 1 ... 180 eps*M and the HIP path by
2 ... 155 eps*M (eps = 2^-24; gpurun_out/r2e/mass.log, both raster kernels alike).  Criterion: relative L2 <= 1e-4
against the f32 oracle; where that fails, the f64 build arbitrates with the fp32 noise floor of this very sum:
    |g_hip - g_f64| <= 1e-4 |g_f64| + 256 eps M,
M = the L1 mass of the per-pixel terms, pushed through |J| / objectMass like the gradient itself, computed from the
engine's per-object alpha / d alpha planes (recomposing the NET sum from the same planes reproduces the engine's
gradient to 1e-8, so the planes are what the gradient is made of).  Round 5: (i) where the f32 ORACLE is itself farther from
the f64 one than that model floor, the floor is GRAD_ORC32_FACTOR x the f32 oracle's own distance (grad_check); (ii) a
gradient that fails is examined PIXEL BY PIXEL (tests/grad_explain.py): pixels whose d alpha difference matters must be
closest-edge ties or near / z-clipped faces by the oracle's own numbers, are weighted out on both sides like image ties
(budget max_grad_tie_pixels) and the same criterion is applied again.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOL = 1e-4          # BASELINE.json north_star: "within 1e-4 fp32"
TZ_REL = 1e-6       # two depths closer than this (relative, ~8 ulp at z = 4) can swap order
TB_REL = 1e-4       # |dist - blur| <= TB_REL * blur: membership of the blur disc can flip
TPAIR_REL = 1e-4    # |d1 - d2| <= TPAIR_REL * max(d): the halves of a z-clipped pair can swap
TEDGE = 5e-7        # pixel centre within this (NDC units, ~8 ulp of a coordinate) of a face edge: inside test can flip
TAREA = 2e-9        # |signed area - kEpsilon(1e-8)| <= TAREA + TVERT * perimeter: the face is visible / culled by a hair
TVERT = 2.5e-7      # vertex-coordinate noise between the two fp32 projections (measured max 2.4e-7 = 1 ulp at |view coord| in [2,4))
TVIEW = 5e-7        # view-space coordinate noise between the two fp32 camera transforms (1 ulp at |coordinate| in [4, 8))
TTEXEL = 1e-3       # barycentric * R within this of a texel-cell boundary
GRAD_NOISE_ULPS = 256.0  # fp32 noise floor of the action gradient, in units of eps * (L1 mass of its pixel terms)
GRAD_ORC32_FACTOR = 2.0  # ... or, where the f32 ORACLE itself is farther than that from the f64 one, this x ITS distance (round 5)
TEAPOT = os.path.join(ROOT, "data", "teapot.obj")


_BENCH_POOL = None


def _bench_pool():
    global _BENCH_POOL
    if _BENCH_POOL is None:
        from occlusionenv_amd.meshes import SyntheticShapeNet

        _BENCH_POOL = SyntheticShapeNet(n_models=1024, seed=1234)
    return _BENCH_POOL


def make_case(n_env, seed, mesh="teapot", az_range=0.6, pool=None, device="cuda"):
    """Seeded scenes as in SURVEY.md §8d config 2: x2 ~ N(0,1), az ~ U(-az_range, az_range), el = 0,
    action ~ N(0,1)^2.  Returns dict with the pool, mesh ids, offsets, az, actions (CPU tensors)."""
    from occlusionenv_amd.meshes import MeshPool, SyntheticShapeNet, load_obj

    g = torch.Generator().manual_seed(seed)
    pool = pool or MeshPool(device)
    if mesh == "teapot":
        v, f = load_obj(TEAPOT)
        ids = [pool.add(v, f, key="teapot")]
    elif mesh == "benchpool":
        # the bench workload's own pool (bench.py: SyntheticShapeNet(n_models=1024, seed=1234)); only the models the
        # case draws are uploaded
        ds = _bench_pool()
        pick = torch.randint(0, len(ds.models), (n_env, 3), generator=torch.Generator().manual_seed(seed + 77))
        ids = {int(m): pool.add(*ds.models[int(m)], key=("benchpool", int(m))) for m in pick.reshape(-1).tolist()}
        x2 = torch.randn(n_env, generator=g)
        az = (torch.rand(n_env, generator=g) * 2 - 1) * az_range
        actions = torch.randn(n_env, 2, generator=g)
        mesh_ids = torch.tensor([[ids[int(m)] for m in row] for row in pick.tolist()])
        offsets = torch.zeros(n_env, 3, 3)
        offsets[:, 1, 0], offsets[:, 1, 2] = x2, 1.0
        offsets[:, 2, 0], offsets[:, 2, 2] = -x2, 2.0
        return dict(pool=pool, mesh_ids=mesh_ids, offsets=offsets, az=az, actions=actions)
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


def oracle_env(case, i, img, shader="flat", faces_per_pixel=100):
    from oracle import p3d_restate as O

    objs, atl = [], []
    for o in range(3):
        mid = int(case["mesh_ids"][i, o])
        v, f = case["pool"].get(mid)
        objs.append((v + case["offsets"][i, o], f))
        atl.append(case["pool"].get_atlas(mid))
    env = O.OracleEnv(objs, img, atlases=atl if all(a is not None for a in atl) else None)
    env.shader = shader
    env.faces_per_pixel = faces_per_pixel
    return env


def run_engine(case, img, n_env=None, faces_per_pixel=100, radius=4.0, pixel_weight=None, render_too=False, shader="flat",
               cost_order=True):
    from occlusionenv_amd import _native as nat
    from occlusionenv_amd.engine import OcclusionEngine

    n = n_env or case["mesh_ids"].shape[0]
    eng = OcclusionEngine(case["pool"], n, img, faces_per_pixel=faces_per_pixel, cost_order=cost_order)
    eng.shader = {"flat": nat.SHADER_FLAT, "hard_phong": nat.SHADER_HARD_PHONG, "soft_phong": nat.SHADER_SOFT_PHONG}[shader]
    eng.set_scene(list(range(n)), case["mesh_ids"][:n], case["offsets"][:n])
    if pixel_weight is not None:
        eng.pixel_weight = pixel_weight.to(eng.device, torch.float32).contiguous()
    obs0, loss0, fs0 = eng.reset_render(None, radius, case["az"][:n], 0.0)
    alphas0 = eng.alphas.clone()
    records0 = snapshot_records(eng)
    a = case["actions"][:n].to(eng.device).requires_grad_(True)
    obs, reward, done, fs, loss = eng.step(a)
    reward.sum().backward()
    eng.check_status()
    out = dict(engine=eng, obs0=obs0.cpu(), loss0=loss0.cpu(), fs0=fs0.cpu(), alphas0=alphas0.cpu(), obs=obs.cpu(),
               reward=reward.detach().cpu(), done=done.cpu(), fs=fs.cpu(), loss=loss.cpu(), grad=a.grad.cpu(),
               campos=eng.camera_position.cpu(), records0=records0, records=snapshot_records(eng))
    out.update(engine_grad_parts(eng))
    if render_too:  # OcclusionEnv.render() at the camera position the step left behind (environment.py:332-347)
        out["render"] = eng.render_hard().cpu()
    return out


def engine_grad_parts(eng):
    """What the last step's action gradient is made of (gradient_mass): per-object d alpha / d(el, az) planes, alphas,
    the action Jacobian and objectMass, copied to the host."""
    from occlusionenv_amd import _native as nat

    n, S = eng.N, eng.S
    og = eng._ws_tensors["obj_grad"].view(torch.float32)[: n * 3 * S * S * 2].view(n, 3, S, S, 2).cpu().clone()
    jac = eng.cam[:, nat.C_J:nat.C_J + 4].cpu().reshape(n, 2, 2).clone()
    return dict(obj_grad=og, jac=jac, object_mass=eng.object_mass.cpu().clone(), alphas=eng.alphas.cpu().clone())


def grad_check(g_gpu, g32, g64_fn, mass_fn):
    """THE criterion for d reward / d action (module docstring): relative L2 <= TOL against the f32 oracle, else the
    f64 oracle (``g64_fn()``, evaluated only then) arbitrates with the fp32 noise floor GRAD_NOISE_ULPS * eps * M
    (``mass_fn()``).  Returns dict(ok, rel32, ...)."""
    grel = float((g32 - g_gpu).norm() / g32.norm().clamp(min=1e-6))
    if grel < TOL:
        return dict(ok=True, rel32=grel)
    g64 = g64_fn()
    mass = mass_fn()
    e_gpu = float((g_gpu.double() - g64).norm())
    e_orc = float((g32.double() - g64).norm())
    # The noise floor of an f32 evaluation of this sum: the model (GRAD_NOISE_ULPS eps M) - or what the reference's own
    # precision MEASURABLY does: where the f32 oracle (torch autograd + the C backward, the reference's arithmetic) is itself
    # farther from the f64 oracle than the model allows, the model is not the floor there (round 5's wide sweep: seeds 9140
    # and 9215, f32 oracle at 1 060 and 28 255 eps M, the engine at 2 454 and 27 840), and the engine is held to
    # GRAD_ORC32_FACTOR x the f32 oracle's own distance.  `floor` says which of the two applied.
    model = GRAD_NOISE_ULPS * 2.0 ** -24 * mass
    measured = GRAD_ORC32_FACTOR * e_orc
    bound = TOL * float(g64.norm()) + max(model, measured)
    return dict(ok=e_gpu <= bound, rel32=grel, g64=float(g64.norm()), mass=mass, e_gpu=e_gpu, e_orc32=e_orc, bound=bound,
                floor="model" if model >= measured else "f32 oracle")


# ---- tie classifier -----------------------------------------------------------------------------------------------
class _Faces:
    """Clipped NDC face list of one mesh under one camera, as the oracle rasterises it (A.2, A.3)."""

    def __init__(self, verts, faces, R, T):
        from oracle import p3d_restate as O

        ndc = O.world_to_ndc(verts, R, T).detach()
        self.fv_unclipped = ndc[faces].contiguous()
        self.fv, self.c2u, self.nb, _, _ = O.clip_faces(ndc[faces], O.Z_CLIP, True)
        self.fv = self.fv.detach().contiguous()


PAIR = "clipped-pair distance tie"


def explain_soft(faces: _Faces, S, yi, xi, K, hair_faces=None, pair_faces=None):
    """Near-ties of the soft rasterisation at one pixel; returns a list of reasons (empty = decision is robust).
    ``hair_faces`` (a set) collects the faces whose visibility is the near-tie: ONE decision per face, however many
    pixels of its blur footprint it moves.  ``pair_faces`` likewise the z-clipped pairs whose two halves are equally far
    from the pixel: ONE geometric coincidence per pair (the band of pixels nearest to the edge or vertex the halves
    share), each pixel of which is then decided by rounding."""
    from oracle import p3d_restate as O

    c = O.pixel_candidates(faces.fv, S, yi, xi, O.BLUR_RADIUS, band=10 * TB_REL, area_band=TAREA, vert_band=TVERT)
    inside = (c["flags"] & 1) != 0
    cand = (c["flags"] & 2) != 0
    why = []
    unc = (~inside) & (np.abs(c["dist"] - O.BLUR_RADIUS) <= TB_REL * O.BLUR_RADIUS)
    if unc.any():
        why.append("blur-boundary")
    hair = ((c["flags"] & 8) != 0) & (inside | (c["dist"] < O.BLUR_RADIUS * (1 + TB_REL)))
    if hair.any():
        why.append(HAIR)
        if hair_faces is not None:
            hair_faces.update(int(f) for f in c["f"][hair])
    unc = unc | hair  # either kind of membership flip also moves the K boundary
    z = np.sort(c["z"][cand])
    n, u = z.size, int(unc.sum())
    # the K nearest are z[0..K-1]; u membership flips move that boundary by up to u places either way.  Two
    # depths closer than rounding on either side of a reachable boundary can swap.
    for k in range(max(K - 1 - u, 0), min(K + u, n - 1)):
        if z[k + 1] - z[k] <= TZ_REL * max(1.0, abs(float(z[k + 1]))):
            why.append("K-boundary depth tie")
            break
    if faces.nb is not None:
        fl = {int(f): j for j, f in enumerate(c["f"])}
        for f, j in fl.items():
            p = int(faces.nb[f])
            if p > f and p in fl:
                d1, d2 = float(c["dist"][j]), float(c["dist"][fl[p]])
                if abs(d1 - d2) <= TPAIR_REL * max(d1, d2, 1e-12):
                    why.append(PAIR)
                    if pair_faces is not None:
                        pair_faces.add(f)
                    break
    if ((c["flags"] & 4) != 0).any() or (np.abs(c["z"]) <= 1e-6).any():
        why.append("pz ~ 0")
    return why


def sliver_depth_bound(fv_face) -> float:
    """First-order bound on the depth-interpolation error of one face under vertex noise TVERT (module docstring,
    SLIVER DEPTH): 2 TVERT perimeter / |area| * (z_max - z_min).  fv_face (3,3) = (x_ndc, y_ndc, z_view) per vertex."""
    v = np.asarray(fv_face, dtype=np.float64)
    area = abs((v[2, 0] - v[0, 0]) * (v[1, 1] - v[0, 1]) - (v[2, 1] - v[0, 1]) * (v[1, 0] - v[0, 0]))
    perim = sum(float(np.hypot(*(v[(k + 1) % 3, :2] - v[k, :2]))) for k in range(3))
    return 2.0 * TVERT * perim / max(area, 1e-30) * float(v[:, 2].max() - v[:, 2].min())


def snapshot_records(eng):
    """The face records the setup kernel left for the LAST render, per (env, object): NDC vertices (n,3,3), original
    face ids, flags (occ_constants.h: 1 / 2 = first / second half of a z-clipped pair, 4 = z-clipped piece)."""
    torch.cuda.synchronize()
    nrec = eng._ws_tensors["nrec"].cpu().numpy()[: eng.NT * 3]
    rec_off = eng._rec_tensors["rec_off"].cpu().numpy().view(np.int64)
    rec = eng._rec_tensors["rec"].view(torch.float32)
    out = []
    for eo in range(eng.N * 3):
        n, base = int(nrec[eo]), int(rec_off[eo])
        r = rec[base * 32: (base + n) * 32].cpu().numpy().reshape(n, 32).copy()
        out.append(dict(fv=torch.from_numpy(r[:, :9].reshape(n, 3, 3).copy()), ids=r[:, 9].view(np.int32).copy(),
                        flags=r[:, 10].view(np.int32).copy(),
                        tan=r[:, 20:32].reshape(n, 3, 4).copy()))  # per vertex: d x/d el, d y/d el, d x/d az, d y/d az
    return out


def face_noise_bounds(fv_unclipped) -> np.ndarray:
    """Per ORIGINAL face: bound on the NDC displacement of any vertex of the face (or of its z-clipped pieces) under
    view-space coordinate noise TVIEW (module docstring, NEAR AND Z-CLIPPED FACES); never below TVERT."""
    from oracle import p3d_restate as O

    v = fv_unclipped.double().numpy()  # (F,3,3): x_ndc, y_ndc, z_view
    s = float(O.proj_scale(torch.float64))
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    front = z >= O.Z_CLIP
    with np.errstate(divide="ignore", invalid="ignore"):
        own = np.where(front, TVIEW * (s + np.maximum(np.abs(x), np.abs(y))) / np.maximum(z, O.Z_CLIP), 0.0).max(1)
        bound = np.maximum(own, TVERT)
        for a, b in ((0, 1), (1, 2), (2, 0)):
            cut = front[:, a] != front[:, b]
            dxy = np.maximum(np.abs(x[:, b] * z[:, b] - x[:, a] * z[:, a]), np.abs(y[:, b] * z[:, b] - y[:, a] * z[:, a]))
            bc = (TVIEW / O.Z_CLIP) * (s + 2.0 * dxy / np.maximum(np.abs(z[:, a] - z[:, b]), 1e-30))
            bound = np.where(cut, np.maximum(bound, bc), bound)
    return bound


def upstream_check(faces: _Faces, rec):
    """(1) of NEAR AND Z-CLIPPED FACES: every engine record against the oracle's face of the same id / half.  Returns
    (ok, worst |difference| / bound, per-original-face bounds)."""
    bounds = face_noise_bounds(faces.fv_unclipped)
    ofv = faces.fv.double().numpy()
    # oracle pieces of every original face (one; two for a face split by the clip plane; none if all of it is behind)
    if faces.c2u is None:
        pieces = {k: [k] for k in range(ofv.shape[0])}
    else:
        pieces = {}
        for c, k in enumerate(faces.c2u.numpy().tolist()):
            if k >= 0:
                pieces.setdefault(k, []).append(c)
    worst = 0.0
    ev = rec["fv"].double().numpy()
    for j in range(ev.shape[0]):
        k = int(rec["ids"][j])
        cs = pieces.get(k, [])
        if len(cs) == 2 and rec["flags"][j] & 3:  # a half that knows its partner: first / second as the oracle lists them
            cs = [cs[1] if rec["flags"][j] & 2 else cs[0]]
        if not cs:
            return False, float("inf"), bounds
        best = float("inf")
        for c in cs:  # (a half whose partner is invisible carries no pair flag: whichever piece it is)
            # vertex by vertex, whatever the order the two sides list them in
            dm = np.abs(ev[j, :, None, :] - ofv[c, None, :, :])  # (3 engine, 3 oracle, xyz)
            m = dm[..., :2].max(-1).argmin(1)
            d = dm[np.arange(3), m, :2].max()
            dz = dm[np.arange(3), m, 2].max()
            best = min(best, max(d / bounds[k], dz / (2.0 * TVIEW)))
        worst = max(worst, best)
    return worst <= 1.0, worst, bounds


class RecordFaces:
    """The engine's own face records of one object (``snapshot_records``) in the shape ``explain_soft`` expects of the
    oracle's face list: the tie classifier then runs on the geometry BOTH sides of a raster-stage comparison share."""

    def __init__(self, rec):
        self.fv, self.c2u = rec["fv"].float().contiguous(), None
        fl, idx = rec["flags"], np.arange(rec["fv"].shape[0])
        nb = np.full(idx.shape[0], -1, dtype=np.int64)
        nb[(fl & 1) != 0] = idx[(fl & 1) != 0] + 1
        nb[(fl & 2) != 0] = idx[(fl & 2) != 0] - 1
        self.nb = torch.from_numpy(nb)


def alpha_of_records(rec, S, K):
    """(2): the ORACLE's naive rasteriser + sigmoid blend on the engine's own records (identical geometry)."""
    from oracle import p3d_restate as O

    fv = rec["fv"].float().contiguous()
    if fv.shape[0] == 0:
        return torch.zeros(S, S)
    fl = rec["flags"]
    nb = np.full(fv.shape[0], -1, dtype=np.int64)
    idx = np.arange(fv.shape[0])
    nb[(fl & 1) != 0] = idx[(fl & 1) != 0] + 1
    nb[(fl & 2) != 0] = idx[(fl & 2) != 0] - 1
    p2f, _, _, dists = O._Rasterize.apply(fv, torch.from_numpy(nb), S, float(O.BLUR_RADIUS), K, True, True, True)
    return O.sigmoid_alpha_blend(dists, p2f)[..., 3]


class _Upstream:
    """Second chance for an unexplained alpha pixel (module docstring, NEAR AND Z-CLIPPED FACES); caches per object."""

    def __init__(self, records_env, S, K):
        self.rec, self.S, self.K, self.cache = records_env, S, K, {}

    def explain(self, faces: _Faces, o, yi, xi, got_alpha):
        from oracle import p3d_restate as O

        if self.rec is None:
            return []
        if o not in self.cache:
            ok, worst, bounds = upstream_check(faces, self.rec[o])
            self.cache[o] = (ok, worst, bounds, alpha_of_records(self.rec[o], self.S, self.K) if ok else None)
        ok, worst, bounds, al = self.cache[o]
        if not ok or abs(float(al[yi, xi]) - float(got_alpha)) > TOL:
            return []
        c = O.pixel_candidates(faces.fv, self.S, yi, xi, O.BLUR_RADIUS, band=10 * TB_REL, area_band=TAREA, vert_band=TVERT)
        orig = c["f"] if faces.c2u is None else faces.c2u.numpy()[c["f"]]
        if not (bounds[orig] > TVERT).any():
            return []
        return ["near / z-clipped face: vertex noise upstream, the raster of the engine's own records agrees"]


def explain_hard(faces: _Faces, S, yi, xi, err_rgb=None, err_depth=None, hair_faces=None):
    """Near-ties of the hard (K = 1) rasterisation at one pixel.  With the pixel's colour / depth errors given, the
    ill-conditioned depth of a needle face is accepted within its bound as well."""
    from oracle import p3d_restate as O

    c = O.pixel_candidates(faces.fv, S, yi, xi, 0.0, band=0.0, area_band=TAREA, vert_band=TVERT)
    if c["f"].size == 0:
        return []
    inside = ((c["flags"] & 1) != 0) & ((c["flags"] & 2) != 0)
    if err_depth is not None and inside.any() and err_rgb <= TOL:
        front = int(c["f"][inside][np.argmin(c["z"][inside])])
        if err_depth <= TOL + sliver_depth_bound(faces.fv[front].numpy()):
            return ["needle face: depth within its conditioning bound"]
    zin = np.sort(c["z"][inside])
    zfront = float(zin[0]) if zin.size else float("inf")
    why = []
    edge = np.sqrt(c["dist"]) <= TEDGE
    tz = TZ_REL * max(1.0, abs(zfront)) if np.isfinite(zfront) else 0.0
    if (edge & (c["z"] <= zfront + tz)).any():
        why.append("pixel centre on a face edge")
    hair = ((c["flags"] & 8) != 0) & (((c["flags"] & 1) != 0) | edge)
    if (hair & (c["z"] <= zfront + tz)).any():
        why.append(HAIR)
        if hair_faces is not None:
            hair_faces.update(int(f) for f in c["f"][hair & (c["z"] <= zfront + tz)])
    if zin.size >= 2 and zin[1] - zin[0] <= TZ_REL * max(1.0, abs(float(zin[1]))):
        why.append("coincident nearest faces")
    return why


def explain_texel(env, S, yi, xi):
    """Barycentric of the visible face on a texel-cell boundary (TexturesAtlas lookup, A.7)?"""
    from oracle import p3d_restate as O

    if env.atlas is None:
        return []
    ndc = O.world_to_ndc(env.scene[0], env.R[0], env.T[0]).detach()
    p2f, _, bary, _ = O.rasterize_meshes(ndc[env.scene[1]], S, 0.0, 1)
    if int(p2f[yi, xi, 0]) < 0:
        return []
    Rr = env.atlas.shape[1]
    w = bary[yi, xi, 0, :2].double().numpy() * Rr
    near = np.abs(w - np.round(w)).min() <= TTEXEL
    diag = abs((w.sum() - np.floor(w).sum()) - 1.0) <= TTEXEL
    return ["texel-cell boundary"] if (near or diag) else []


HAIR = "face visible / culled by a hair (area ~ kEpsilon)"
REASON_LOG = None  # diagnostics (scripts/dbg): set to a list to collect (kind, obj, y, x, reasons) of every explained pixel


def _classify(env, got_alphas, or_alphas, got_obs, or_obs, S, K, textured, decisions=None, records=None):
    """Pixels beyond tolerance -> (tie mask (S,S) bool, list of unexplained (kind, obj, y, x, err)).  ``decisions`` (a
    set) collects what max_tie_pixels bounds: one entry per tie pixel, except that the pixels whose reason is the
    visibility of a needle face share one entry per such face (its whole blur footprint flips with it), and the pixels
    whose only reason is the equal distance of the two halves of a z-clipped pair one entry per pair."""
    ties = torch.zeros(S, S, dtype=torch.bool)
    unexplained = []
    decisions = set() if decisions is None else decisions
    upstream = _Upstream(records, S, K)
    R, T = env.R[0], env.T[0]
    dal = (or_alphas - got_alphas).abs()
    faces_cache = {}
    for o, y, x in torch.nonzero(dal > TOL).tolist():
        if o not in faces_cache:
            faces_cache[o] = _Faces(env.objs[o][0], env.objs[o][1], R, T)
        hf, pf = set(), set()
        why = explain_soft(faces_cache[o], S, y, x, K, hair_faces=hf, pair_faces=pf)
        up = False
        if not why:
            why = upstream.explain(faces_cache[o], o, y, x, got_alphas[o, y, x])
            up = bool(why)
        if why:
            ties[y, x] = True
            if REASON_LOG is not None:
                REASON_LOG.append(("alpha", o, y, x, tuple(why)))
            if up:
                decisions.add(("upstream", o, y, x))
            elif HAIR in why:
                decisions.update(("face", o, f) for f in hf)
            elif why == [PAIR]:
                decisions.update(("pair", o, f) for f in pf)
            else:
                decisions.add(("pixel", y, x))
        else:
            unexplained.append(("alpha", o, y, x, float(dal[o, y, x])))
    dch = (or_obs - got_obs).abs()
    dob = dch.max(0).values  # (S,S) over the 4 channels
    scene = None
    for y, x in torch.nonzero(dob > TOL).tolist():
        if scene is None:
            scene = _Faces(env.scene[0], env.scene[1], R, T)
        hf = set()
        why = explain_hard(scene, S, y, x, err_rgb=float(dch[:3, y, x].max()), err_depth=float(dch[3, y, x]), hair_faces=hf)
        if not why and textured:
            why = explain_texel(env, S, y, x)
        if why:
            ties[y, x] = True
            if REASON_LOG is not None:
                REASON_LOG.append(("obs", -1, y, x, tuple(why)))
            decisions.update(("face", -1, f) for f in hf) if HAIR in why else decisions.add(("pixel", y, x))
        else:
            unexplained.append(("obs", -1, y, x, float(dob[y, x])))
    return ties, unexplained


def run_parity_case(n_env=2, img=64, seed=0, mesh="teapot", az_range=0.6, check_envs=None, radius=4.0, mutate=None,
                    faces_per_pixel=100, check_render=False, shader="flat"):
    """One seeded batch through the HIP engine and the oracle.  Returns the worst differences over the checked envs
    (all pixels that are not explained exact ties; loss / reward / gradient with the ties weighted out on both
    sides) plus ``unexplained`` (must be empty), ``tie_pixels`` (weighted out) and ``tie_decisions`` (bounded by
    max_tie_pixels: pixels, with the footprint of one hair-flipped needle face counted once)."""
    from oracle import p3d_restate as O

    case = make_case(n_env, seed, mesh, az_range)
    if mutate is not None:
        mutate(case)  # e.g. push an object out of view
    got = run_engine(case, img, radius=radius, faces_per_pixel=faces_per_pixel, render_too=check_render, shader=shader)
    S, K = img, faces_per_pixel
    envs = list(check_envs if check_envs is not None else range(n_env))
    textured = mesh == "textured"
    orc, weights, unexplained, n_ties, n_dec, n_up = {}, torch.ones(n_env, S, S), [], 0, 0, 0
    for i in envs:
        dec = set()
        env = oracle_env(case, i, img, shader, faces_per_pixel)
        obs0 = env.reset(radius=radius, azimuth=float(case["az"][i]))
        al0 = torch.stack([im[0, ..., 3] for im in env.alphas]).detach()
        img0 = env.image.detach()
        t0, u0 = _classify(env, got["alphas0"][i], al0, got["obs0"][i], obs0[0].detach(), S, K, textured, dec,
                         records=got["records0"][3 * i: 3 * i + 3])
        a = case["actions"][i].clone().requires_grad_(True)
        obs, reward, done, info = env.step(a)
        al = torch.stack([im[0, ..., 3] for im in env.alphas]).detach()
        t1, u1 = _classify(env, got["alphas"][i], al, got["obs"][i], obs[0].detach(), S, K, textured, dec,
                         records=got["records"][3 * i: 3 * i + 3])
        rnd = None
        if check_render:
            rimg, rdepth = env.render()
            rnd = torch.cat([rimg[0, ..., :3].permute(2, 0, 1), rdepth[0].permute(2, 0, 1)]).detach()
        ties = t0 | t1
        unexplained += [(i, "reset") + u for u in u0] + [(i, "step") + u for u in u1]
        weights[i] = (~ties).float()
        n_ties = max(n_ties, int(ties.sum()))
        n_dec = max(n_dec, sum(1 for d in dec if d[0] != "upstream"))
        n_up = max(n_up, sum(1 for d in dec if d[0] == "upstream"))
        orc[i] = dict(env=env, obs0=obs0[0].detach(), al0=al0, img0=img0, a=a, obs=obs[0].detach(), al=al, ties=ties,
                      t0=t0, t1=t1, render=rnd)
    if n_ties:  # leave the tie pixels out of the loss on the GPU side too
        got_w = run_engine(case, img, radius=radius, faces_per_pixel=faces_per_pixel, pixel_weight=weights, shader=shader)
    else:
        got_w = got
    res = dict(obs_maxabs=0.0, obs0_maxabs=0.0, alpha_maxabs=0.0, alpha0_maxabs=0.0, fs_maxabs=0.0, loss_rel=0.0,
               loss0_rel=0.0, reward_abs=0.0, grad_rel=0.0, grad_excess=0.0, grad_arbiter=[], render_maxabs=0.0,
               tie_pixels=n_ties, tie_decisions=n_dec, upstream_pixels=n_up, fs_arith=0.0, unexplained=unexplained, img=img)
    for i in envs:
        o = orc[i]
        env, keep0, keep1, keep = o["env"], ~o["t0"], ~o["t1"], ~o["ties"]
        res["obs_maxabs"] = max(res["obs_maxabs"], float(((o["obs"] - got["obs"][i]).abs() * keep1).max()))
        res["obs0_maxabs"] = max(res["obs0_maxabs"], float(((o["obs0"] - got["obs0"][i]).abs() * keep0).max()))
        res["alpha_maxabs"] = max(res["alpha_maxabs"], float(((o["al"] - got["alphas"][i]).abs() * keep1).max()))
        res["alpha0_maxabs"] = max(res["alpha0_maxabs"], float(((o["al0"] - got["alphas0"][i]).abs() * keep0).max()))
        # The occlusion image I = a1 a2 + a2 a3 + a1 a3 (channel 3; RGB = 3) is DERIVED from the alphas: alphas within
        # 1e-4 only bound it to sum_i |d a_i| (a_j + a_k) <= 1e-4 * 2 (a1 + a2 + a3).  Two checks instead of one blunt one:
        # (a) the combine kernel's arithmetic - the engine's image against I formed from the ENGINE's own alphas, every
        # pixel, 1e-6; (b) against the oracle with that first-order bound of a 1e-4 error per alpha.
        def occl(a):
            return a[0] * a[1] + a[1] * a[2] + a[0] * a[2]

        fs_or = env.image[0].detach()
        for fs_g, al_g, fs_o, al_o, keep_ in ((got["fs"][i], got["alphas"][i], fs_or, o["al"], keep1),
                                              (got["fs0"][i], got["alphas0"][i], o["img0"][0], o["al0"], keep0)):
            res["fs_arith"] = max(res["fs_arith"], float((fs_g[..., 3] - occl(al_g)).abs().max()),
                                  float((fs_g[..., :3] - 3.0).abs().max()))
            scale = (2.0 * al_o.sum(0)).clamp(min=1.0)
            res["fs_maxabs"] = max(res["fs_maxabs"], float(((fs_o[..., 3] - fs_g[..., 3]).abs() / scale * keep_).max()))
        if o["render"] is not None:
            res["render_maxabs"] = max(res["render_maxabs"], float(((o["render"] - got["render"][i]).abs() * keep1).max()))
        # loss / reward / gradient with the tie pixels weighted out (environment.py:381-392 restated on the images)
        w = keep.to(torch.float32)
        loss0 = torch.sum(w * o["img0"][0, ..., 3] ** 2)
        loss = torch.sum(w * env.image[0, ..., 3] ** 2)
        reward = (loss0 - loss) / (loss0 + 1)
        finished = bool(loss.detach() < 0.1)
        reward = reward + 5 if finished else reward - 0.2
        reward.backward()
        g = o["a"].grad
        lo = float(loss)
        res["loss_rel"] = max(res["loss_rel"], abs(lo - float(got_w["loss"][i])) / max(abs(lo), 1.0))
        res["loss0_rel"] = max(res["loss0_rel"], abs(float(loss0) - float(got_w["loss0"][i])) / max(abs(float(loss0)), 1.0))
        # (1e-4 of the reward's magnitude where that exceeds 1: with objectMass = 1 the reward IS a loss of ~100)
        res["reward_abs"] = max(res["reward_abs"], abs(float(reward) - float(got_w["reward"][i])) / max(abs(float(reward)), 1.0))
        # fp32 cancellation noise or a real error?  beyond 1e-4 the f64 oracle arbitrates (module docstring)
        gc = grad_check(got_w["grad"][i], g, lambda: _oracle_grad64(case, i, img, radius, w, faces_per_pixel),
                        lambda: gradient_mass(got_w, i, w))
        if not gc["ok"] and shader == "flat":
            # GRADIENT TIES (tests/grad_explain.py): which pixels make the gradient differ, and is each of them a near-tie
            # by the oracle's own numbers - the closest edge of a face undecided within the positional noise, or a near /
            # z-clipped face whose gradient the engine's own records reproduce?  Those pixels are weighted out on both
            # sides, like image ties, and the SAME criterion is applied again.
            from tests import grad_explain as GX

            ex = GX.explain_gradient(case, i, img, faces_per_pixel, radius, got_w, w)
            assert ex["selfcheck"] < 1e-8, ("the f64 forward sweep does not reproduce the f64 oracle's autograd", ex["selfcheck"])
            n_gt = int(ex["ties"].sum())
            res["grad_tie_pixels"] = max(res.get("grad_tie_pixels", 0), n_gt)
            res.setdefault("grad_tie_reasons", {}).update({k: res.get("grad_tie_reasons", {}).get(k, 0) + v for k, v in ex["reasons"].items()})
            if n_gt:
                if REASON_LOG is not None:
                    REASON_LOG.extend(("grad", -1, int(y), int(x), ("gradient: " + "+".join(sorted(ex["reasons"])),)) for y, x in torch.nonzero(ex["ties"]).tolist())
                w2 = w * (~ex["ties"]).float()
                wts2 = weights.clone()
                wts2[i] = w2
                got2 = run_engine(case, img, radius=radius, faces_per_pixel=faces_per_pixel, pixel_weight=wts2, shader=shader)
                env2 = oracle_env(case, i, img, shader, faces_per_pixel)
                env2.reset(radius=radius, azimuth=float(case["az"][i]))
                l0 = torch.sum(w2 * env2.image[0, ..., 3].detach() ** 2)
                a2 = case["actions"][i].clone().requires_grad_(True)
                env2.step(a2)
                ((l0 - torch.sum(w2 * env2.image[0, ..., 3] ** 2)) / (l0 + 1)).backward()
                gc = grad_check(got2["grad"][i], a2.grad, lambda: _oracle_grad64(case, i, img, radius, w2, faces_per_pixel),
                                lambda: gradient_mass(got2, i, w2))
        res["grad_rel"] = max(res["grad_rel"], gc["rel32"])
        if "g64" in gc:
            res["grad_arbiter"].append(dict(gc, env=i))
            if not gc["ok"]:
                res["grad_excess"] = max(res["grad_excess"], gc["e_gpu"] / max(gc["g64"], 1e-12))
        assert finished == bool(got_w["done"][i]), (i, lo, float(got_w["loss"][i]))
    return res


def gradient_mass(got, i, w):
    """L1 mass of the per-pixel terms of d reward / d action of env i (norm over the two action components):
    sum_px w |2 I dI/d alpha_o| |d alpha_o / d(el, az)| through |J| / objectMass (occ_combine.hpp, occ_finish_kernel)."""
    og = got["obj_grad"][i]
    a = got["alphas"][i]
    I = a[0] * a[1] + a[1] * a[2] + a[0] * a[2]
    gsum = torch.stack([a[1] + a[2], a[0] + a[2], a[0] + a[1]])
    dal = torch.where((a > 0)[..., None], og, torch.zeros(()))  # the planes are only defined where the object is
    mass = ((2 * I * w)[None, :, :, None] * gsum[..., None] * dal).abs().sum((0, 1, 2))  # (2,): el, az
    return float((got["jac"][i].abs().t() @ mass).norm() / float(got["object_mass"][i]))


def _oracle_grad64(case, i, img, radius, w, faces_per_pixel=100):
    """d reward / d action of env i from the f64 build of the oracle, loss weighted by w like the f32 comparison."""
    from oracle import p3d_restate as O

    e32 = oracle_env(case, i, img)
    env = O.OracleEnv([(v.double(), f) for v, f in e32.objs], img, dtype=torch.float64)
    env.faces_per_pixel = faces_per_pixel
    env.reset(radius=radius, azimuth=float(case["az"][i]))
    wd = w.double()
    loss0 = torch.sum(wd * env.image[0, ..., 3].detach() ** 2)
    a = case["actions"][i].clone().double().requires_grad_(True)
    env.step(a)
    loss = torch.sum(wd * env.image[0, ..., 3] ** 2)
    reward = (loss0 - loss) / (loss0 + 1)
    reward.backward()
    return a.grad


TIE_FRAC = 2e-4          # explained tie DECISIONS per env: at most this share of the object-pixels (floor 4)
UPSTREAM_FRAC = 2e-3     # pixels accepted through the near / z-clipped rule per env: at most this share (floor 8)
FOOTPRINT_FACTOR = 8     # tie PIXELS per env: at most this many per allowed decision (the blur footprint of a hair-flipped needle)


def max_tie_pixels(img, n_objects=3):
    """Bound on explained tie decisions per env: near-ties are rounding coincidences, a handful per image."""
    return max(4, int(TIE_FRAC * img * img * n_objects))


def max_upstream_pixels(img, n_objects=3):
    """Bound on the pixels accepted through the near / z-clipped rule per env (each passed the rule's three machine
    checks): with the camera inside an object a cut edge parallel to the clip plane moves along its whole length."""
    return max(8, int(UPSTREAM_FRAC * img * img * n_objects))


def max_grad_tie_pixels(img, n_objects=3):
    """Bound on the pixels weighted out as gradient ties per env (tests/grad_explain.py): near-ties are rounding
    coincidences, a handful per image."""
    from tests.grad_explain import GRAD_TIE_FRAC

    return max(4, int(GRAD_TIE_FRAC * img * img * n_objects))


def violations(res, tol=TOL):
    """What the parity tests, smoke() and scripts/parity_sweep.py all check; returns a list of failures (empty = ok)."""
    bad = []
    if res["unexplained"]:
        bad.append("unexplained pixels: %s" % (res["unexplained"][:6],))
    if res.get("tie_decisions", res["tie_pixels"]) > max_tie_pixels(res["img"]):
        bad.append("too many tie pixels: %d (%d decisions)" % (res["tie_pixels"], res.get("tie_decisions", -1)))
    if res.get("upstream_pixels", 0) > max_upstream_pixels(res["img"]):
        bad.append("too many pixels under the near / z-clipped rule: %d" % res["upstream_pixels"])
    if res["tie_pixels"] > FOOTPRINT_FACTOR * max_tie_pixels(res["img"]) + max_upstream_pixels(res["img"]):  # footprints of hair-flipped needles included
        bad.append("too many tie pixels: %d" % res["tie_pixels"])
    if not res.get("fs_arith", 0.0) < 1e-6:
        bad.append("fs_arith = %.3e" % res["fs_arith"])
    for k in ("obs_maxabs", "obs0_maxabs", "alpha_maxabs", "alpha0_maxabs", "fs_maxabs", "render_maxabs", "loss_rel",
              "loss0_rel", "reward_abs"):
        if not res[k] < tol:
            bad.append("%s = %.3e" % (k, res[k]))
    if res.get("grad_tie_pixels", 0) > max_grad_tie_pixels(res["img"]):
        bad.append("too many gradient-tie pixels: %d (%s)" % (res["grad_tie_pixels"], res.get("grad_tie_reasons")))
    if res["grad_excess"] > 0.0:  # beyond 1e-4 of the f32 oracle AND rejected by the f64 arbiter (module docstring)
        bad.append("grad: %s" % [a for a in res["grad_arbiter"] if not a["ok"]])
    return bad


def check_result(res, tol=TOL):
    bad = violations(res, tol)
    assert not bad, (bad, {k: v for k, v in res.items() if k != "unexplained"})
