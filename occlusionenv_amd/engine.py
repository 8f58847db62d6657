"""Batched in-process engine: N environments' meshes and cameras packed into ONE launch sequence
(camera -> setup -> tile raster -> reduce -> finish) on one GPU.

It replaces the sequential per-env Python loop of ``SimpleVecEnv.step_wait``
(/root/reference/SubProcVecEnv.py:209-218) and the four PyTorch3D renders + autograd graph of
``OcclusionEnv.step`` (/root/reference/environment.py:352-396).  All arithmetic happens in the HIP
kernels behind the C ABI (include/occlusionenv_amd.h); torch is used for device memory, streams and
the autograd hook only.  No CPU fallback exists: constructing an engine without the HIP extension
or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _native as nat
from .meshes import MeshPool


WAVES_PER_CU = 12  # resident persistent waves per CU of occ_raster2_kernel (3 per SIMD at <= 168 VGPRs, 12.2 KB LDS each)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class _RewardGrad(torch.autograd.Function):
    """Attaches d reward / d action (computed in forward mode by the render sweep) to autograd, so
    ``reward.backward()`` / ``rewards.sum().backward()`` fill ``action.grad`` like the reference
    (demo.py:86, train_predict.py:52)."""

    @staticmethod
    def forward(ctx, actions, reward, grad_action):
        ctx.save_for_backward(grad_action)
        return reward.view_as(reward)  # a view, not a copy: one launch less per step

    @staticmethod
    def backward(ctx, g):
        (grad_action,) = ctx.saved_tensors
        return g.reshape(-1, 1) * grad_action, None, None


class OcclusionEngine:
    """State + workspace of N environments on one GPU."""

    def __init__(self, pool: MeshPool, n_env: int, img_size: int, device=None, faces_per_pixel: int = 100,
                 waves_per_cu: Optional[int] = None, reserve: int = 0, cost_order: bool = True, output_ring: int = 0,
                 output_recycle: int = 0):
        self.lib = nat.load()
        if not torch.cuda.is_available():
            raise nat.NativeError("OcclusionEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                                  "there is no CPU fallback")
        if img_size % nat.TILE or img_size < nat.TILE or img_size > 2048:
            raise ValueError(f"img_size must be a multiple of {nat.TILE} in [8, 2048]")
        if not (0 < faces_per_pixel <= nat.MAX_K):
            raise ValueError(f"faces_per_pixel must be in 1..{nat.MAX_K}")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.pool = pool
        self.N = int(n_env)
        self.S = int(img_size)
        self.K = int(faces_per_pixel)
        # persistent waves per CU of the raster kernel: what its LDS / VGPR budget admits (occ_raster2.hpp)
        self.waves_per_cu = int(waves_per_cu or os.environ.get("OCC_WAVES_PER_CU", WAVES_PER_CU))
        # work items heaviest first (occ_order_kernel); False = plain rect order (OccWorkspace.order = NULL)
        self.cost_order = bool(cost_order)
        # OUTPUT RING (opt-in; whole-batch steps with the reserve only).  0: every step returns freshly allocated
        # obs / full_state tensors, every pixel of which the combine kernel writes (three quarters of a 128 x 128 frame
        # are background constants).  k >= 2: k persistent output sets are used in turn and the kernel writes only the
        # pixel blocks that meet this step's object rects or what the set held before (OccRenderOut.rect_prev): the
        # tensors a step returns are overwritten k steps later - fine for rollout loops, which consume an observation
        # before the next step (PPO.py:152-164), not for callers that keep observations around.
        if output_ring == 1 or output_ring < 0:
            raise ValueError("output_ring must be 0 (fresh outputs) or >= 2")
        self.output_ring = int(output_ring)
        self._ring, self._ring_pos = None, 0
        # RECYCLED OUTPUTS (whole-batch steps with the reserve; what SimpleVecEnv switches on by default).  Up to
        # ``output_recycle`` persistent output sets, like the ring - but a set is handed out again only once NOTHING outside
        # the engine refers to its tensors any more (no live view of its obs / full_state storage: the caller dropped the
        # step's ``obs`` and ``infos``, as a rollout loop does by rebinding them).  While somebody still holds them a step
        # takes another free set, a new one, or - at the cap - freshly allocated tensors.  Nobody can therefore see a
        # tensor change under them: the lifetime contract is the reference's (every step returns tensors of its own,
        # /root/reference/SubProcVecEnv.py:215-219), only the allocator differs, and with it what the combine kernel has
        # to write (the set still holds its background outside the tracked rects).
        if output_recycle < 0:
            raise ValueError("output_recycle must be >= 0 (0 = off)")
        self.output_recycle = int(output_recycle)
        self._picked = None  # the output set chosen for the next whole-batch step (pick_output_set), or "fresh"
        # the setup kernel stages every object's vertices in LDS once (False: three global gathers per face; same results)
        self.setup_vertex_lds = bool(int(os.environ.get("OCC_SETUP_VERTEX_LDS", "1")))
        d = self.device
        f32 = dict(dtype=torch.float32, device=d)
        N = self.N
        # ``reserve`` extra scene slots ride along with every batched step: candidate scenes for the auto-reset
        # of finished envs, rendered with the reset() camera in the SAME launch sequence (no extra launch, no
        # extra host sync when an env finishes).  They are invisible through the public (N-sized) views.
        self.R = int(reserve)
        self._rs_mesh_host = None
        self._rs_off_host = None
        NT = self.NT = N + self.R
        # environment state (environment.py:302-306,323-324); rows N.. belong to the reserve
        self._el_all = torch.zeros(NT, **f32)
        self._az_all = torch.zeros(NT, **f32)
        self._rad_all = torch.full((NT,), 4.0, **f32)
        self._mesh_all = torch.zeros(NT, 3, dtype=torch.int32, device=d)
        self._off_all = torch.zeros(NT, 3, 3, **f32)
        self._cam_all = torch.zeros(NT, nat.CAM_STRIDE, **f32)
        self._alphas_all = torch.zeros(NT, 3, self.S, self.S, **f32)
        self.elevation, self.azimuth, self.radius = self._el_all[:N], self._az_all[:N], self._rad_all[:N]
        self.camera_position = torch.zeros(N, 3, **f32)
        self.full_reward = torch.zeros(N, **f32)
        self.object_mass = torch.ones(N, **f32)
        # episode time limit (trainRL.py:22,191-229): steps since the env's last reset, counted on the device by
        # occ_auto_reset; max_ep_len = 0: no limit (the reference's SimpleVecEnv has none)
        self.age = torch.zeros(N, dtype=torch.int32, device=d)
        self.max_ep_len = 0
        # normWithObjectSize (environment.py:208,320,324), per env: reset() then sets objectMass = sum_px (a1+a2+a3)^2 + 1
        # instead of loss + 1.  Off for every env by default, like the reference; nothing is computed while none has it on.
        self.norm_flags = torch.zeros(N, dtype=torch.int32, device=d)
        self._norm_host = np.zeros(N, dtype=bool)
        # region tracking of the persistent alphas state (OccRenderOut.arect_prev / arect_next): outside rect[e] the
        # alphas of row e are zero.  Two arrays, flipped by every tracked launch; untracked writers mark rows full-frame.
        self._arect = [torch.tensor([self.S, self.S, -1, -1], dtype=torch.int32, device=d).repeat(NT, 1).contiguous()
                       for _ in range(2)]
        self._arect_cur = 0
        self._full_rect = None
        # scene description
        self.scene_mesh, self.scene_offset = self._mesh_all[:N], self._off_all[:N]
        # internal buffers
        self.cam, self.alphas = self._cam_all[:N], self._alphas_all[:N]
        self._ws_key = None
        self._ws = None
        self._ws_tensors = None
        self._rec_tensors = None   # rec / rec_bbox / scan / rec_cbox / rec_off: sized by the records actually needed
        self._rec_total = 0
        self._mesh_host = np.zeros((NT, 3), dtype=np.int64)  # host copy of _mesh_all (kept current before every launch)
        self._need_all = None        # cached _records_needed(_mesh_host) (see _put_mesh_rows)
        self._need_res = 0           # ... of the reserve rows alone
        self._scene_cache = None     # (key, OccScene) of the last whole-table scene struct
        self._faces_np = np.zeros(0, dtype=np.int64)
        self._faces_ver = -1
        self._reserve_cam_done = False
        # optional (N,S,S) weight of every pixel's loss term (OccScene.pix_weight); None = 1 (environment.py:381)
        self.pixel_weight: Optional[torch.Tensor] = None
        # shader of the RGB-D observation: nat.SHADER_FLAT (HardFlatShader, environment.py:283) or one of the two
        # alternatives the reference keeps commented out (environment.py:281-282)
        self.shader = nat.SHADER_FLAT
        if self.R:
            # device-side auto-reset state (include/occlusionenv_amd.h: occ_auto_reset)
            i32 = dict(dtype=torch.int32, device=d)
            self.rs_state = torch.zeros(self.R, **i32)   # OCC_RS_EMPTY
            self.rs_tries = torch.zeros(self.R, **i32)
            self._pairs = torch.zeros(2 + 3 * self.R, **i32)
            # pinned rows the refill kernel reads in place; four buffers in turn (a step refills at most twice, and the host
            # waits for the step's report event - later on the stream than those kernels - before the next step's refills)
            self._refill_host = [torch.zeros(self.R, 13, dtype=torch.int32).pin_memory() for _ in range(4)]
            self._report_host = [torch.zeros(N + 2 * self.R + 2, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._flip = 0
            # what the last render of every slot produced: a slot is rendered only while PENDING (skip mask), a
            # READY slot keeps its stored observation / occlusion image / loss until an env takes it
            S = self.S
            self._res_obs = torch.zeros(self.R, 4, S, S, **f32)
            self._res_fs = torch.zeros(self.R, S, S, 4, **f32)
            self._res_loss = torch.zeros(self.R, **f32)
            self._res_objsum = torch.zeros(self.R, **f32)  # sum_px (a1+a2+a3)^2 of every slot's stored render (normWithObjectSize)
            self._skip = torch.zeros(NT, **i32)
            self._skip[N:] = 1  # EMPTY slots are not rendered

    # ---- normWithObjectSize ----------------------------------------------------------------
    def set_norm_with_object_size(self, env_id: int, on: bool) -> None:
        on = bool(on)
        if self._norm_host[env_id] != on:
            self._norm_host[env_id] = on
            self.norm_flags[env_id] = int(on)

    def _object_sum(self, alphas: torch.Tensor, gate=None, gate_value=0, out=None) -> torch.Tensor:
        """sum_px (a1 + a2 + a3)^2 per row of ``alphas`` (n,3,S,S) (environment.py:320,324; occ_object_mass)."""
        a = alphas.contiguous()
        n = a.shape[0]
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=self.device)
        nat.check(self.lib.occ_object_mass(_p(a), n, self.S, _p(gate), int(gate_value), _p(out), self._stream()), "occ_object_mass")
        return out

    def _reset_mass(self, loss: torch.Tensor, alphas: torch.Tensor, env_ids) -> torch.Tensor:
        """objectMass of reset() for the envs ``env_ids`` (tensor of indices or None = all), environment.py:324."""
        if not self._norm_host.any():
            return loss + 1.0
        flags = self.norm_flags if env_ids is None else self.norm_flags[env_ids]
        return torch.where(flags != 0, self._object_sum(alphas), loss) + 1.0

    # ---- scenes ---------------------------------------------------------------------------
    def set_scene(self, env_ids, mesh_ids, offsets) -> None:
        """mesh_ids (n,3) pool ids of object 1..3, offsets (n,3,3) world offsets (environment.py:148,171)."""
        idx = torch.as_tensor(env_ids, dtype=torch.long, device=self.device).reshape(-1)
        m = torch.as_tensor(mesh_ids, dtype=torch.int32).reshape(-1, 3)
        if int(m.min()) < 0 or int(m.max()) >= len(self.pool):
            raise ValueError("mesh id outside the pool")
        self.scene_mesh[idx] = m.to(self.device)
        self.scene_offset[idx] = torch.as_tensor(offsets, dtype=torch.float32).reshape(-1, 3, 3).to(self.device)
        self._put_mesh_rows(env_ids, m.numpy())

    def _alpha_touch(self, rows) -> None:
        """Rows of the alphas state were written as whole frames by something other than a tracked launch."""
        if not self.R:  # no reserve = no tracked launch ever reads the rects (a single env: nothing extra on its step)
            return
        if self._full_rect is None:
            self._full_rect = torch.tensor([0, 0, self.S - 1, self.S - 1], dtype=torch.int32, device=self.device)
        self._arect[self._arect_cur][rows] = self._full_rect

    @staticmethod
    def _storage_refs(t: torch.Tensor) -> int:
        """How many tensors share ``t``'s storage (views included), as the allocator counts them."""
        return int(torch._C._storage_Use_Count(t.untyped_storage()._cdata))

    def _new_output_set(self) -> dict:
        """One persistent output set: obs / full_state filled with background, two rect arrays saying where it is not."""
        d, S, NT = self.device, self.S, self.NT
        f32 = dict(dtype=torch.float32, device=d)
        obs = torch.empty(NT, 4, S, S, **f32)
        obs[:, :3] = 1.0
        obs[:, 3] = -1.0   # white background, depth -1 (environment.py:378)
        fs = torch.empty(NT, S, S, 4, **f32)
        fs[..., :3] = 3.0
        fs[..., 3] = 0.0   # environment.py:373 with all three alphas 0
        rect = [torch.tensor([S, S, -1, -1], dtype=torch.int32, device=d).repeat(NT, 1).contiguous() for _ in range(2)]
        slot = dict(obs=obs, fs=fs, rect=rect, cur=0, event=None)
        slot["refs0"] = (self._storage_refs(obs), self._storage_refs(fs))  # with nobody but this dict holding them
        return slot

    def _set_is_free(self, slot) -> bool:
        return (self._storage_refs(slot["obs"]), self._storage_refs(slot["fs"])) == slot["refs0"]

    def pick_output_set(self):
        """Choose the output set of the NEXT whole-batch step: (index, set) - or (None, None) for freshly allocated tensors
        (no ring, no recycling, or every set of a full recycling pool still has a live view somewhere).  The choice
        sticks until that step has launched."""
        if self._picked is not None:
            return self._picked
        if self.output_ring:
            if self._ring is None or len(self._ring) != self.output_ring:
                self._ring = [self._new_output_set() for _ in range(self.output_ring)]
                self._ring_pos = 0
            k = self._ring_pos
            self._ring_pos = (self._ring_pos + 1) % self.output_ring
            self._picked = (k, self._ring[k])
        elif self.output_recycle and self.R:
            if self._ring is None:
                self._ring = []
            self._picked = (None, None)
            for k, slot in enumerate(self._ring):
                if self._set_is_free(slot):
                    self._picked = (k, slot)
                    break
            else:
                if len(self._ring) < self.output_recycle:
                    self._ring.append(self._new_output_set())
                    self._picked = (len(self._ring) - 1, self._ring[-1])
        else:
            self._picked = (None, None)
        return self._picked

    # ---- workspace ------------------------------------------------------------------------
    def _scene_struct(self, n, scene_mesh, scene_offset, skip=None, pix_weight=None) -> nat.OccScene:
        pv, pf, vo, fo = self.pool.device_tensors()
        sc = nat.OccScene()
        sc.pool_verts, sc.pool_faces = pv.data_ptr(), pf.data_ptr()
        sc.mesh_vert_off, sc.mesh_face_off = vo.data_ptr(), fo.data_ptr()
        sc.scene_mesh, sc.scene_offset = scene_mesh.data_ptr(), scene_offset.data_ptr()
        sc.n_meshes, sc.n_env, sc.img = len(self.pool), n, self.S
        sc.rec_cap = self._rec_cap()
        sc.max_mesh_verts = self.pool.max_verts if self.setup_vertex_lds else 0
        atlas, aoff = self.pool.atlas_tensors()
        if atlas is not None:
            sc.pool_atlas, sc.mesh_atlas_off, sc.atlas_res = atlas.data_ptr(), aoff.data_ptr(), self.pool.atlas_res
        if skip is not None:
            sc.skip = skip.data_ptr()
        if self.shader != nat.SHADER_FLAT:
            sc.shader = int(self.shader)
            sc.pool_vnormals = self.pool.vertex_normals_tensor().data_ptr()
        if pix_weight is not None:
            if pix_weight.shape != (n, self.S, self.S) or pix_weight.dtype != torch.float32 or not pix_weight.is_contiguous():
                raise ValueError(f"pixel weights must be a contiguous float32 ({n}, {self.S}, {self.S}) tensor")
            sc.pix_weight = pix_weight.data_ptr()
        return sc

    def _put_mesh_rows(self, rows, mesh_ids) -> None:
        """Write rows of the host copy of the scene mesh ids and keep the cached record need of the whole table
        current (whole-batch launches ask for it every step: the sum is maintained, not recomputed)."""
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        m = np.asarray(mesh_ids, dtype=np.int64).reshape(-1, 3)
        if self._need_all is not None and self._faces_ver == self.pool.version and rows.size <= 64 \
                and np.unique(rows).size == rows.size:
            f_old = self._faces_np[self._mesh_host[rows].reshape(-1)]
            f_new = self._faces_np[m.reshape(-1)]
            d = ((((2 * f_new + 63) >> 6) << 6) - (((2 * f_old + 63) >> 6) << 6)).reshape(-1, 3).sum(1)
            self._need_all += int(d.sum())
            self._need_res += int(d[rows >= self.N].sum())
        else:
            self._need_all = None  # recomputed on demand
        self._mesh_host[rows] = m

    def _records_needed_all(self, ahead: bool = False) -> int:
        """Records the whole table (envs + reserve rows) needs.  ``ahead``: an upper bound that also holds if the DEVICE
        has meanwhile installed reserve scenes in env rows that the host has not heard of yet (auto-reset commits of the
        last step, report not processed): every such commit replaces an env's scene by a slot's, so counting the
        reserve rows twice covers any number of them."""
        if self._need_all is None or self._faces_ver != self.pool.version:
            self._need_all = self._records_needed(self._mesh_host)
            self._need_res = self._records_needed(self._mesh_host[self.N:]) if self.R else 0
        return self._need_all + (self._need_res if ahead else 0)

    def _rec_cap(self) -> int:
        # a z-clipped face can split in two: worst case 2 records per face
        return 2 * max(self.pool.max_faces, 1)

    def _records_needed(self, mesh_ids) -> int:
        """Records a launch over these scenes needs: 2 x faces of every object's mesh, rounded up to 64
        (occ_recoff_kernel lays the spans out the same way)."""
        if self._faces_ver != self.pool.version:
            self._faces_np = np.array([self.pool.num_faces(m) for m in range(len(self.pool))], dtype=np.int64)
            self._faces_ver = self.pool.version
        f = self._faces_np[np.asarray(mesh_ids, dtype=np.int64).reshape(-1)]
        return int((((2 * f + 63) >> 6) << 6).sum())

    def note_commit(self, env_id: int, slot: int) -> None:
        """The device installed reserve ``slot``'s scene in ``env_id`` (occ_auto_reset): keep the host copy current."""
        self._put_mesh_rows([env_id], self._mesh_host[self.N + slot].copy())

    def note_commits(self, env_ids, slots) -> None:
        """``note_commit`` for a batch (index arrays; an env takes at most one slot per step)."""
        if len(env_ids):
            self._put_mesh_rows(env_ids, self._mesh_host[self.N + np.asarray(slots)].copy())

    def _ensure_workspace(self, need_records: Optional[int] = None) -> nat.OccWorkspace:
        """Fixed-size scratch (planes, K-buffers, ...) once; the record arrays hold ``rec_total`` records in a
        variable layout (every object gets room for ITS mesh) and grow geometrically when a launch needs more."""
        key = (self.NT, self.S)
        d = self.device

        def buf(nbytes):
            return torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=d)

        if self._ws_key != key:
            sc = self._scene_struct(self.NT, self._mesh_all, self._off_all)
            sizes = nat.OccWorkspaceSizes()
            cus = self.lib.occ_device_cu_count()
            if cus <= 0:
                raise nat.NativeError("occ_device_cu_count failed")
            n_slots = cus * self.waves_per_cu
            nat.check(self.lib.occ_workspace_query(C.byref(sc), n_slots, C.byref(sizes)), "occ_workspace_query")
            self._ws_tensors = None
            t = dict(nrec=buf(sizes.nrec_bytes), objrect=buf(sizes.objrect_bytes), queue=buf(sizes.queue_bytes),
                     lists=buf(sizes.lists_bytes), partials=buf(sizes.partials_bytes), status=buf(sizes.status_bytes),
                     offsets=buf(sizes.offsets_bytes), obj_alpha=buf(sizes.obj_alpha_bytes),
                     obj_grad=buf(sizes.obj_grad_bytes), obj_hz=buf(sizes.obj_hz_bytes), obj_hrec=buf(sizes.obj_hrec_bytes),
                     order=buf(sizes.order_bytes))
            ws = nat.OccWorkspace()
            for k, v in t.items():
                if k != "order" or self.cost_order:
                    setattr(ws, k, v.data_ptr())
            ws.n_slots = sizes.n_slots
            self._ws, self._ws_tensors, self._ws_key = ws, t, key
            self._rec_tensors, self._rec_total = None, 0
        if need_records is None:
            need_records = self._records_needed_all()
        if need_records > self._rec_total or self._rec_tensors is None:
            total = max(need_records, int(self._rec_total * 1.5), 64)
            total = ((total + 63) >> 6) << 6
            sizes = nat.OccWorkspaceSizes()
            nat.check(self.lib.occ_record_sizes(total, self.NT, C.byref(sizes)), "occ_record_sizes")
            nbytes = sizes.rec_bytes + sizes.rec_bbox_bytes + sizes.scan_bytes + sizes.rec_cbox_bytes + sizes.rec_off_bytes
            self._rec_tensors = None  # release the old arrays first
            free = torch.cuda.mem_get_info(d)[0]
            if nbytes > free:
                raise nat.NativeError(
                    f"the face records of this batch need {nbytes / 2**30:.1f} GiB ({total} records of 160 B) but only "
                    f"{free / 2**30:.1f} GiB are free: use fewer envs per GPU or lower OCC_MAX_MESH_FACES")
            r = dict(rec=buf(sizes.rec_bytes), rec_bbox=buf(sizes.rec_bbox_bytes), scan=buf(sizes.scan_bytes),
                     rec_cbox=buf(sizes.rec_cbox_bytes), rec_off=buf(sizes.rec_off_bytes))
            for k, v in r.items():
                setattr(self._ws, k, v.data_ptr())
            self._ws.rec_total = total
            self._rec_tensors, self._rec_total = r, total
        return self._ws

    @property
    def status(self) -> torch.Tensor:
        self._ensure_workspace()
        return self._ws_tensors["status"][: self.NT]

    def check_status(self) -> None:
        """Raise if any kernel reported a data-dependent failure (host sync)."""
        st = int(self.status.max().item())
        if st:
            bad = torch.nonzero(self.status).reshape(-1)[:8].tolist()
            self.status.zero_()
            raise nat.NativeError(
                f"kernel status {st} for envs {bad}: "
                f"{'internal error: a tile classed as overflow-free held a pixel with more than K candidates; ' if st & nat.STATUS_LIST_OVERFLOW else ''}"
                f"{'face-record capacity exceeded' if st & nat.STATUS_REC_OVERFLOW else ''}")

    # ---- launches -------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _camera_args(self, mode, action, el, az, rad, pos_out, n, pos2=None) -> nat.OccCameraArgs:
        ca = nat.OccCameraArgs()
        ca.mode, ca.n = mode, n
        for name, t in (("action", action), ("el", el), ("az", az), ("radius", rad), ("cam_pos_out", pos_out), ("cam_pos_out2", pos2)):
            setattr(ca, name, None if t is None else t.data_ptr())
        return ca

    def _render(self, idx: Optional[torch.Tensor], cam_mode: int, cam_input: Optional[torch.Tensor], flags: int, finish=None):
        """Camera + render for all envs (idx None) or the compact subset idx, one occ_step call.  ``finish``: an
        OccStepFinish whose reward bookkeeping the launch does as well.  Returns a dict of fresh tensors."""
        rows = self._mesh_host[: self.N] if idx is None else self._mesh_host[idx.cpu().numpy()]
        ws = self._ensure_workspace(self._records_needed(rows))
        d, S = self.device, self.S
        f32 = dict(dtype=torch.float32, device=d)
        if idx is None:
            n = self.N
            el, az, rad = self.elevation, self.azimuth, self.radius
            cam, campos, alphas = self.cam, self.camera_position, self.alphas
            smesh, soff = self.scene_mesh, self.scene_offset
        else:
            n = int(idx.numel())
            el, az, rad = self.elevation[idx], self.azimuth[idx], self.radius[idx]
            cam = torch.empty(n, nat.CAM_STRIDE, **f32)
            campos = torch.empty(n, 3, **f32)
            alphas = torch.empty(n, 3, S, S, **f32)
            smesh, soff = self.scene_mesh[idx].contiguous(), self.scene_offset[idx].contiguous()
        st = self._stream()
        cam_in = None if cam_input is None else cam_input.contiguous()
        # reset() leaves camera_position untouched (environment.py:302 zeros, never written by reset)
        write_pos = campos if cam_mode == nat.CAM_STEP else None
        out = {}
        if cam_mode == nat.CAM_STEP:
            out["pos"] = torch.empty(n, 3, **f32)  # this step's camera positions (info["position"] snapshot)
        ca = self._camera_args(cam_mode, cam_in, el, az, rad, write_pos, n, out.get("pos"))
        ro = nat.OccRenderOut()
        if finish is not None:
            ro.finish = C.pointer(finish)
        if flags & nat.RENDER_HARD:
            out["obs"] = torch.empty(n, 4, S, S, **f32)
            ro.obs = out["obs"].data_ptr()
        if flags & nat.RENDER_SOFT:
            out["full_state"] = torch.empty(n, S, S, 4, **f32)
            out["loss"] = torch.empty(n, **f32)
            ro.full_state, ro.loss, ro.alphas = out["full_state"].data_ptr(), out["loss"].data_ptr(), alphas.data_ptr()
        if flags & nat.RENDER_GRAD:
            out["grad_elaz"] = torch.empty(n, 2, **f32)
            ro.grad_elaz = out["grad_elaz"].data_ptr()
        pw = None
        if self.pixel_weight is not None:
            pw = self.pixel_weight if idx is None else self.pixel_weight[idx].contiguous()
        sc = self._scene_struct(n, smesh, soff, pix_weight=pw)
        nat.check(self.lib.occ_step(C.byref(sc), C.byref(ca), _p(cam), C.byref(ws), C.byref(ro), flags, self.K, st), "occ_step")
        if idx is not None:
            if cam_mode == nat.CAM_STEP:
                self.elevation[idx], self.azimuth[idx] = el, az
                self.camera_position[idx] = campos
            self.cam[idx] = cam
            if flags & nat.RENDER_SOFT:
                self.alphas[idx] = alphas
                self._alpha_touch(idx)
        elif flags & nat.RENDER_SOFT:
            self._alpha_touch(slice(0, n))  # an untracked launch wrote whole frames
        out["cam"] = cam
        out["_keep"] = (smesh, soff, el, az, rad, cam_in, pw)  # keep temporaries alive until the stream is done with them
        return out

    def reset_render(self, env_ids=None, radius=4.0, azimuth=0.0, elevation=0.0):
        """State init + initial renders of reset() (environment.py:302-324).  Returns (obs, loss) for the subset."""
        idx = None if env_ids is None else torch.as_tensor(env_ids, dtype=torch.long, device=self.device).reshape(-1)
        sel = slice(None) if idx is None else idx

        def put(dst, val):
            dst[sel] = torch.as_tensor(val, dtype=torch.float32, device=self.device) if not torch.is_tensor(val) \
                else val.to(self.device, torch.float32)

        put(self.radius, radius)
        put(self.elevation, elevation)
        put(self.azimuth, azimuth)
        self.camera_position[sel] = 0.0
        out = self._render(idx, nat.CAM_LOOKAT, None, nat.RENDER_SOFT | nat.RENDER_HARD)
        loss = out["loss"]
        self.full_reward[sel] = loss
        self.object_mass[sel] = self._reset_mass(loss, self.alphas[sel] if self._norm_host.any() else None, idx)  # environment.py:324
        return out["obs"], loss, out["full_state"]

    def evaluate_scenes(self, mesh_ids, offsets, radius, azimuth, elevation):
        """Render m candidate scenes (NOT bound to env slots) with the reset() camera: the speculative half of
        the batched reset.  m <= N (the workspace is sized for N envs).  Returns a dict of per-candidate tensors."""
        d = self.device
        f32 = dict(dtype=torch.float32, device=d)
        smesh = torch.as_tensor(mesh_ids, dtype=torch.int32).reshape(-1, 3).to(d).contiguous()
        m = smesh.shape[0]
        if m > self.NT:
            raise ValueError("more candidate scenes than env slots")
        if int(smesh.min()) < 0 or int(smesh.max()) >= len(self.pool):
            raise ValueError("mesh id outside the pool")
        soff = torch.as_tensor(offsets, dtype=torch.float32).reshape(m, 3, 3).to(d).contiguous()

        def vec(v):
            t = torch.as_tensor(v, dtype=torch.float32).to(d)
            return t.expand(m).contiguous() if t.ndim == 0 else t.reshape(m).contiguous()

        rad, az, el = vec(radius), vec(azimuth), vec(elevation)
        mesh_host = np.asarray(mesh_ids, dtype=np.int64).reshape(-1, 3)
        ws = self._ensure_workspace(self._records_needed(mesh_host))
        S = self.S
        cam = torch.empty(m, nat.CAM_STRIDE, **f32)
        st = self._stream()
        ca = self._camera_args(nat.CAM_LOOKAT, None, el, az, rad, None, m)
        out = dict(obs=torch.empty(m, 4, S, S, **f32), full_state=torch.empty(m, S, S, 4, **f32),
                   loss=torch.empty(m, **f32), alphas=torch.empty(m, 3, S, S, **f32), cam=cam,
                   radius=rad, azimuth=az, elevation=el, scene_mesh=smesh, scene_offset=soff, mesh_host=mesh_host)
        ro = nat.OccRenderOut()
        ro.obs, ro.full_state, ro.loss, ro.alphas = (out["obs"].data_ptr(), out["full_state"].data_ptr(),
                                                     out["loss"].data_ptr(), out["alphas"].data_ptr())
        sc = self._scene_struct(m, smesh, soff)
        nat.check(self.lib.occ_step(C.byref(sc), C.byref(ca), _p(cam), C.byref(ws), C.byref(ro), nat.RENDER_SOFT | nat.RENDER_HARD,
                                    self.K, st), "occ_step")
        return out

    def commit_reset(self, env_ids, cand_ids, res) -> None:
        """Install candidates ``cand_ids`` of an evaluate_scenes() result as the fresh state of ``env_ids``
        (environment.py:302-306,323-324)."""
        e = torch.as_tensor(env_ids, dtype=torch.long, device=self.device).reshape(-1)
        c = torch.as_tensor(cand_ids, dtype=torch.long, device=self.device).reshape(-1)
        self.scene_mesh[e] = res["scene_mesh"][c]
        self.scene_offset[e] = res["scene_offset"][c]
        self._put_mesh_rows(env_ids, res["mesh_host"][np.asarray(cand_ids, dtype=np.int64).reshape(-1)])
        self.radius[e] = res["radius"][c]
        self.azimuth[e] = res["azimuth"][c]
        self.elevation[e] = res["elevation"][c]
        self.camera_position[e] = 0.0
        self.cam[e] = res["cam"][c]
        self.alphas[e] = res["alphas"][c]
        self._alpha_touch(e)
        self.age[e] = 0
        loss = res["loss"][c]
        self.full_reward[e] = loss
        self.object_mass[e] = self._reset_mass(loss, res["alphas"][c] if self._norm_host.any() else None, e)

    def render_hard(self, env_ids=None):
        """render() (environment.py:332-347): hard RGB-D at the current camera_position."""
        idx = None if env_ids is None else torch.as_tensor(env_ids, dtype=torch.long, device=self.device).reshape(-1)
        pos = self.camera_position if idx is None else self.camera_position[idx]
        out = self._render(idx, nat.CAM_POSITION, pos, nat.RENDER_HARD)
        return out["obs"]

    def set_reserve_state(self, state, tries) -> None:
        """Overwrite the device-side slot states (host-driven warm-up only; no step may be in flight)."""
        st = torch.as_tensor(np.asarray(state), dtype=torch.int32)
        self.rs_state.copy_(st)
        self.rs_tries.copy_(torch.as_tensor(np.asarray(tries), dtype=torch.int32))
        self._skip[self.N:].copy_((st != nat.RS_PENDING).to(torch.int32))

    def install_reserve(self, slots, res, cand) -> None:
        """Store candidates ``cand`` of an evaluate_scenes() result as the last render of reserve ``slots`` (the
        host-driven warm-up accepted them; they will be READY without ever being rendered by a step launch)."""
        if not len(slots):
            return
        sl = torch.as_tensor(slots, dtype=torch.long, device=self.device)
        c = torch.as_tensor(cand, dtype=torch.long, device=self.device)
        self._res_obs[sl] = res["obs"][c]
        self._res_fs[sl] = res["full_state"][c]
        self._res_loss[sl] = res["loss"][c]
        if self._norm_host.any():
            self._res_objsum[sl] = self._object_sum(res["alphas"][c])
        self._alphas_all[self.N + sl] = res["alphas"][c]
        self._alpha_touch(self.N + sl)
        self._cam_all[self.N + sl] = res["cam"][c]

    def refill_reserve(self, slots, mesh_ids, offsets) -> None:
        """Hand new candidate scenes to EMPTY reserve slots: one packed H2D copy + occ_reserve_refill (scatter,
        marks the slots PENDING).  Stream-ordered: the next step's launch renders them."""
        n = len(slots)
        if not n:
            return
        m = np.asarray(mesh_ids, dtype=np.int32).reshape(n, 3)
        if int(m.min()) < 0 or int(m.max()) >= len(self.pool):
            raise ValueError("mesh id outside the pool")
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.float32).reshape(n, 9))
        sl = np.asarray(slots, dtype=np.int64)
        if self._rs_mesh_host is None:
            self._rs_mesh_host = torch.zeros(self.R, 3, dtype=torch.int32).pin_memory()
            self._rs_off_host = torch.zeros(self.R, 3, 3, dtype=torch.float32).pin_memory()
        self._rs_mesh_host.numpy()[sl] = m
        self._rs_off_host.numpy()[sl] = off.reshape(n, 3, 3)
        self._put_mesh_rows(self.N + sl, m)
        self._flip = (self._flip + 1) % len(self._refill_host)
        host = self._refill_host[self._flip]
        h = host.numpy()
        h[:n, 0] = sl
        h[:n, 1:4] = m
        h[:n, 4:13] = off.view(np.int32)
        # the kernel reads the rows straight from the pinned buffer when it runs (no copy launch): the buffers alternate, and
        # before this one is written again the host has waited for a later event on the stream (the next step's report)
        nat.check(self.lib.occ_reserve_refill(C.c_void_p(host.data_ptr()), n, self.N, self.R, _p(self._mesh_all), _p(self._off_all),
                                              _p(self.rs_state), _p(self._skip), self._stream()), "occ_reserve_refill")

    def auto_reset(self, out) -> dict:
        """Device-side auto-reset of the envs that finished in the step which produced ``out`` (pairing with READY
        reserve slots + commit, occ_auto_reset) and an asynchronous copy of its report to pinned host memory.
        Nothing here waits for the GPU; ``event.synchronize()`` before reading ``report_host``."""
        N, R, S = self.N, self.R, self.S
        term = torch.empty(R, 4, S, S, dtype=torch.float32, device=self.device)
        reset_fs = torch.empty(R, S, S, 4, dtype=torch.float32, device=self.device)  # occlusion image of every slot taken now
        report = torch.empty(N + 2 * R + 2, dtype=torch.int32, device=self.device)
        opts = nat.OccAutoResetOpts()
        opts.age, opts.max_ep_len = self.age.data_ptr(), int(self.max_ep_len)
        opts.reset_full_state = reset_fs.data_ptr()
        opts.arect = self._arect[self._arect_cur].data_ptr()
        if out.get("rect") is not None:
            opts.rect = out["rect"].data_ptr()
        if self._norm_host.any():
            # the slots rendered by this step (PENDING until the pairing below runs): their sum_px (a1+a2+a3)^2
            self._object_sum(self._alphas_all[N:], gate=self.rs_state, gate_value=nat.RS_PENDING, out=self._res_objsum)
            opts.norm_flags, opts.slot_objsum = self.norm_flags.data_ptr(), self._res_objsum.data_ptr()
        st = nat.OccEnvState()
        st.el, st.az, st.radius = self._el_all.data_ptr(), self._az_all.data_ptr(), self._rad_all.data_ptr()
        st.campos, st.cam, st.alphas = self.camera_position.data_ptr(), self._cam_all.data_ptr(), self._alphas_all.data_ptr()
        st.full_reward, st.object_mass = self.full_reward.data_ptr(), self.object_mass.data_ptr()
        st.scene_mesh, st.scene_offset = self._mesh_all.data_ptr(), self._off_all.data_ptr()
        store = nat.OccReserveStore()
        store.obs, store.full_state, store.loss = self._res_obs.data_ptr(), self._res_fs.data_ptr(), self._res_loss.data_ptr()
        store.skip = self._skip.data_ptr()
        # the pairing launch writes the report into pinned host memory itself (no copy launch); two buffers alternate
        self._rflip = getattr(self, "_rflip", 0) ^ 1
        host = self._report_host[self._rflip]
        host.numpy()[:N] = 0  # the kernel writes only the entries of the envs it resets (last read two steps ago)
        opts.report_host = host.data_ptr()
        nat.check(self.lib.occ_auto_reset(_p(out["done_u8"]), _p(out["loss_all"]), _p(self.status), N, R, _p(self.rs_state),
                                          _p(self.rs_tries), C.byref(st), _p(out["obs_all"]), _p(out["full_state_all"]),
                                          C.byref(store), _p(term), S, _p(self._pairs), _p(report), C.byref(opts), self._stream()),
                  "occ_auto_reset")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return dict(report_host=host, event=ev, term=term, report=report, reset_fs=reset_fs)

    def step_flags(self, done_u8, loss_all) -> torch.Tensor:
        """(N + R + 1) int32 on the device: done | reserve scene accepted | any status bit (one D2H copy later)."""
        flags = torch.empty(self.NT + 1, dtype=torch.int32, device=self.device)
        nat.check(self.lib.occ_step_flags(_p(done_u8), _p(loss_all), _p(self.status), self.N, self.R if loss_all is not None else 0,
                                          _p(flags), self._stream()), "occ_step_flags")
        return flags

    def step(self, actions: torch.Tensor, env_ids=None, with_reserve: bool = False, pre_launch=None):
        """Batched step(): returns (obs (n,4,S,S), reward (n,) [autograd-attached], done (n,) bool, full_state, loss).
        ``with_reserve`` (whole-batch steps only) also renders the reserve scenes in the same launch sequence and
        returns a sixth item: the dict of whole-batch tensors (``obs_all``, ``loss_all`` ...).  ``pre_launch`` (with
        the reserve only) is called after every output is allocated and every argument struct is built, right
        before the first kernel launch: the place for a host sync that must precede this step."""
        idx = None if env_ids is None else torch.as_tensor(env_ids, dtype=torch.long, device=self.device).reshape(-1)
        n = self.N if idx is None else int(idx.numel())
        if actions.shape != (n, 2):
            raise ValueError(f"actions must have shape ({n}, 2), got {tuple(actions.shape)}")
        need_grad = bool(actions.requires_grad and torch.is_grad_enabled())
        a = actions.detach().to(self.device, torch.float32).contiguous()
        flags = nat.RENDER_SOFT | nat.RENDER_HARD | (nat.RENDER_GRAD if need_grad else 0)
        d = self.device
        reward = torch.empty(n, dtype=torch.float32, device=d)
        done = torch.empty(n, dtype=torch.uint8, device=d)
        grad_action = torch.empty(n, 2, dtype=torch.float32, device=d) if need_grad else None
        # the reward bookkeeping (environment.py:381-392) rides on the launch that reduces the loss (OccStepFinish)
        fr = self.full_reward if idx is None else self.full_reward[idx]
        om = self.object_mass if idx is None else self.object_mass[idx]
        fin = nat.OccStepFinish()
        fin.full_reward, fin.object_mass, fin.reward, fin.done = fr.data_ptr(), om.data_ptr(), reward.data_ptr(), done.data_ptr()
        fin.grad_action = None if grad_action is None else grad_action.data_ptr()
        fin.n_step = n
        if with_reserve and idx is None and self.R > 0:
            out = self._render_with_reserve(a, flags, pre_launch, fin)
        else:
            with_reserve = False
            if pre_launch is not None:
                pre_launch()
            out = self._render(idx, nat.CAM_STEP, a, flags, fin)
        if idx is not None:
            self.full_reward[idx] = fr
        if need_grad:
            reward = _RewardGrad.apply(actions, reward, grad_action)
        # occ_step_finish writes 0 / 1 bytes: reinterpret them as bool (no conversion launch)
        res = (out["obs"], reward, done.view(torch.bool), out["full_state"], out["loss"])
        out["done_u8"] = done
        return res + (out,) if with_reserve else res

    def _render_with_reserve(self, actions, flags, pre_launch=None, finish=None):
        """One launch sequence (one occ_step call) over N stepping envs (OCC_CAM_STEP) + R reserve scenes (OCC_CAM_LOOKAT)."""
        d, S, N, NT = self.device, self.S, self.N, self.NT
        f32 = dict(dtype=torch.float32, device=d)
        st = self._stream()
        ro = nat.OccRenderOut()
        out = {}
        _, slot = self.pick_output_set()
        self._picked = None
        if slot is not None:
            obs, fs = slot["obs"], slot["fs"]
            rp, rn = slot["rect"][slot["cur"]], slot["rect"][slot["cur"] ^ 1]
            slot["cur"] ^= 1
            ro.rect_prev, ro.rect_next = rp.data_ptr(), rn.data_ptr()
            out["rect"] = rn  # describes the set's content once this launch has run (auto-reset / fallback mark rows here)
        else:
            obs = torch.empty(NT, 4, S, S, **f32)
            fs = torch.empty(NT, S, S, 4, **f32)
        loss = torch.empty(NT, **f32)
        ro.obs, ro.full_state, ro.loss, ro.alphas = obs.data_ptr(), fs.data_ptr(), loss.data_ptr(), self._alphas_all.data_ptr()
        if finish is not None:
            ro.finish = C.pointer(finish)
        if flags & nat.RENDER_GRAD:
            g = torch.empty(NT, 2, **f32)
            ro.grad_elaz = g.data_ptr()
            out["grad_elaz"] = g[:N]
        pw = None
        if self.pixel_weight is not None:  # reserve rows (reset candidates) are scored unweighted
            pw = torch.ones(NT, S, S, **f32)
            pw[:N] = self.pixel_weight
        out["pos"] = torch.empty(N, 3, **f32)  # this step's camera positions (info["position"] snapshot)
        ca = self._camera_args(nat.CAM_STEP, actions, self._el_all, self._az_all, self._rad_all, self.camera_position, N, out["pos"])
        if pre_launch is not None:
            pre_launch()
        # pre_launch may have installed other scenes (auto-reset commits; the synchronous fallback reset may pick
        # larger models or grow the pool): size the record arrays and build the scene struct only now
        ws = self._ensure_workspace(self._records_needed_all(ahead=True))
        # the alphas state is persistent: tracked like a one-set ring (rows the launch skips keep their rect; the fallback
        # reset above writes whole alpha frames and marks their rects in the CURRENT array: flip only now)
        ro.arect_prev, ro.arect_next = self._arect[self._arect_cur].data_ptr(), self._arect[self._arect_cur ^ 1].data_ptr()
        self._arect_cur ^= 1
        # the scene struct of the whole table only changes with the pool, the shader or the weights: kept between steps
        key = (self.pool.version, int(self.shader), None if pw is None else pw.data_ptr(), self._rec_cap(), self.setup_vertex_lds)
        if self._scene_cache is None or self._scene_cache[0] != key:
            self._scene_cache = (key, self._scene_struct(NT, self._mesh_all, self._off_all, self._skip, pix_weight=pw))
        sc = self._scene_cache[1]
        if not self._reserve_cam_done:  # reset() camera of the reserve rows: radius 4, az = el = 0, never changes
            nat.check(self.lib.occ_camera(nat.CAM_LOOKAT, None, _p(self._el_all[N:]), _p(self._az_all[N:]),
                                          _p(self._rad_all[N:]), _p(self._cam_all[N:]), None, self.R, st), "occ_camera")
            self._reserve_cam_done = True
        dump = os.environ.get("OCC_DEBUG_DUMP")
        if dump:  # diagnostics: the inputs of the launch that is about to run (host sync; cam / el / az as BEFORE this step's camera update)
            torch.save(dict(mesh=self._mesh_all.cpu(), off=self._off_all.cpu(), el=self._el_all.cpu(), az=self._az_all.cpu(),
                            rad=self._rad_all.cpu(), cam=self._cam_all.cpu(), actions=actions.cpu(), N=N, NT=NT), dump)
        nat.check(self.lib.occ_step(C.byref(sc), C.byref(ca), _p(self._cam_all), C.byref(ws), C.byref(ro), flags, self.K, st),
                  "occ_step")
        out.update(obs=obs[:N], full_state=fs[:N], loss=loss[:N], cam=self._cam_all, obs_all=obs, loss_all=loss,
                   full_state_all=fs, _keep=pw)
        return out
