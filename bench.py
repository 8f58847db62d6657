#!/usr/bin/env python
"""bench.py -- env steps/sec of the batched OcclusionEnv step() on MI355X.

A "step" = one batched pass of the hot path: SimpleVecEnv.step(actions) (camera -> setup -> tile raster ->
reduce -> finish for every env of this rank) + rewards.sum().backward() (action gradients), + for N>1 GPUs the
single RCCL all-gather of rollout records.  value = (envs of all ranks x steps) / max-over-ranks wall time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--img S]
                  [--workload shapenet5k|mixed|teapot|ppo_rollout]

``--gpus N`` with N > 1 from a plain shell: this process starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``
as a CHILD process (never an exec, and before this process has imported torch or touched the GPU), relays the
ranks' output (rank 0 prints the one JSON line) and exits with the child's code.  When torchrun itself started
bench.py (WORLD_SIZE is set) the ranks run directly.

``--workload ppo_rollout`` = BASELINE config 5 at its per-rank size (256 envs/GPU, 256x256): every step is one step
of a PPO rollout (/root/reference/trainRL.py:189-229, PPO.py:152-223) - act from the old policy on the pooled
features, env step, differentiable-reward backward to the action, rollout record, all-gather - and every T = 50
steps the heads-only clipped-surrogate update runs inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4        # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # peak engine clock (MI355X_MICROARCH.md)
VALU_CYCLES = 2        # issue cycles of one wave64 VALU instruction on a SIMD (same guide)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # warm-up 10, >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs", type=int, default=None, help="envs PER GPU (weak scaling); default 1024 (ppo_rollout: 256)")
    ap.add_argument("--img", type=int, default=None, help="image side; default 128 (ppo_rollout: 256)")
    ap.add_argument("--workload", default="shapenet5k", choices=["shapenet5k", "mixed", "teapot", "ppo_rollout"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on fewer GPUs than ranks (ranks share devices)")
    ap.add_argument("--cpu-sample", type=int, default=16)
    ap.add_argument("--pool-models", type=int, default=1024, help="synthetic mesh pool size (SURVEY 8d: 1024, seed 1234)")
    ap.add_argument("--rollout-T", type=int, default=50,
                    help="ppo_rollout: steps between policy updates.  Default 50 = BASELINE config 5's rollout length (SURVEY 8d: "
                         "'PPO-style rollout T=50' = one episode of max_ep_len steps, trainRL.py:22); the reference's own cadence is "
                         "--rollout-T 200 (update_timestep = max_ep_len * 4, trainRL.py:46)")
    ap.add_argument("--ppo-epochs", type=int, default=80, help="ppo_rollout: K_epochs (trainRL.py:49)")
    ap.add_argument("--max-ep-len", type=int, default=50, help="ppo_rollout: episode time limit (trainRL.py:22), 0 = none")
    ap.add_argument("--output-ring", type=int, default=0,
                    help="opt-in: k >= 2 persistent output sets used in turn (the step's obs / full_state are OVERWRITTEN k steps "
                         "later whoever holds them - not the reference's tensor lifetime).  Default 0: recycled outputs, "
                         "SimpleVecEnv's default - a set is reused only once no view of it is alive, so no caller can see a "
                         "tensor change (reference semantics)")
    ap.add_argument("--no-recycle", action="store_true",
                    help="every step allocates its obs / full_state (and writes every pixel of them) instead of recycling "
                         "released output sets; the 'fresh_outputs' leg of the JSON line measures this mode as well")
    ap.add_argument("--fresh-steps", type=int, default=40, help="steps of the fresh_outputs leg (N = 1 only; 0 = skip)")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    args = ap.parse_args(argv)
    if args.envs is None:
        args.envs = 256 if args.workload == "ppo_rollout" else 1024
    if args.img is None:
        args.img = 256 if args.workload == "ppo_rollout" else 128
    return args


# ---- self-launch (N > 1 from a plain shell) ------------------------------------------------------------------------
def free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(gpus: int, argv, port: int, environ=None):
    """The child process that runs the ranks: (argv list, environment).  ``argv`` are bench.py's own arguments
    (sys.argv[1:]), passed through unchanged except for --master-port."""
    passthrough, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--master-port":
            skip = True
            continue
        if a.startswith("--master-port="):
            continue
        passthrough.append(a)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + passthrough
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return cmd, env


def self_launch(args, argv) -> int:
    """Start the N ranks as a child process and relay its output and exit code.  Touches no GPU."""
    cmd, env = launcher_command(args.gpus, argv, args.master_port or free_port())
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    saw_json = False
    for line in proc.stdout:
        saw_json = saw_json or line.startswith("{")
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and not saw_json:
        print("bench.py: the ranks exited 0 but rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return rc


# ---- workload ---------------------------------------------------------------------------------------------------------
def b_alg_bytes(vf_sum: float, S: int) -> float:
    """SURVEY.md §8d canonical algorithmic bytes per env-step (fwd+bwd): 24*sum_i(V_i+F_i) + 56*S^2."""
    return 24.0 * vf_sum + 56.0 * S * S


def build_env(workload: str, n_env: int, img: int, seed: int, pool_models: int = 1024):
    import numpy as np

    from occlusionenv_amd.environment import OcclusionEnv, seed_scene_rng
    from occlusionenv_amd.meshes import SyntheticShapeNet
    from occlusionenv_amd.SubProcVecEnv import SimpleVecEnv

    np.random.seed(seed)
    seed_scene_rng(seed)  # model draws of the auto-reset scenes: reproducible runs
    if workload == "teapot":
        ds = None
    else:
        # pool of 1 024 procedural ShapeNet-size meshes, seed 1234 (94 MB on the device: beyond L2)
        # (cached under the temp dir: eight ranks of a node start together, and ~26 s of mesh generation per rank and run
        # is start-up nobody needs twice)
        import tempfile

        ds = SyntheticShapeNet(n_models=pool_models, seed=1234, mixed=(workload == "mixed"), cache_dir=tempfile.gettempdir())
    venv = SimpleVecEnv([(lambda: OcclusionEnv(ds, img_size=img)) for _ in range(n_env)])
    venv.seed(seed)
    return venv, ds


# ---- CPU baseline (the oracle as the checker's port of the reference's CPU path; after the timed region) -----------
def _oracle_scene_payloads(venv, idx, img):
    """Plain numpy description of the engine's current scenes ``idx`` (picklable: the all-cores leg runs in
    worker PROCESSES, which never see the GPU)."""
    eng = venv.engine
    az = eng.azimuth.cpu().numpy()
    out = []
    for i in idx:
        ids, offs = venv.envs[i]._scene
        objs = []
        for m, o in zip(ids, offs):
            v, f = eng.pool.get(m)
            objs.append((v.cpu().numpy(), f.cpu().numpy(), [float(x) for x in o]))
        out.append(dict(objs=objs, img=img, az=float(az[i]), seed=1000 + int(i)))
    return out


def _oracle_env(payload):
    import torch as T

    from oracle import p3d_restate as O

    objs = [(T.from_numpy(v) + T.tensor(o, dtype=T.float32), T.from_numpy(f)) for v, f, o in payload["objs"]]
    env = O.OracleEnv(objs, payload["img"])
    env.reset(azimuth=payload["az"])  # not timed: state init only
    a = T.randn(2, generator=T.Generator().manual_seed(payload["seed"])).requires_grad_(True)
    return env, a


def _oracle_step(pair):
    env, a = pair
    t0 = time.perf_counter()
    _, r, _, _ = env.step(a)
    r.backward()
    return time.perf_counter() - t0


_CPU_BARRIER = None


def _cpu_worker_init(barrier):
    global _CPU_BARRIER
    _CPU_BARRIER = barrier
    os.environ["CUDA_VISIBLE_DEVICES"] = ""  # a worker must never initialise the GPU
    os.environ["HIP_VISIBLE_DEVICES"] = ""


def _cpu_worker(job):
    """One worker process: build its envs (untimed), meet the others at the barrier, step every env once on
    ``threads`` threads (the C rasteriser runs outside the GIL; the torch glue around it is what a process's GIL
    serialises, hence several processes)."""
    import concurrent.futures as cf

    import torch as T

    payloads, threads = job
    T.set_num_threads(1)
    pairs = [_oracle_env(p) for p in payloads]
    _CPU_BARRIER.wait(timeout=600)
    t0 = time.time()
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(_oracle_step, pairs))
    return t0, time.time(), len(pairs)


CPU_WORKER_PROCESSES = 4  # a GPU box lets at most 6 processes hold the GPU device open, and ``import torch`` opens it


def cpu_cores() -> int:
    """Host cores this process may run on (a 1-GPU box's share is 16 of the host's cores), at most 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_baseline(venv, n_sample: int, img: int, budget_s: float = 12.0):
    """Oracle (CPU restatement of the reference's PyTorch3D CPU path) timed on a bounded sample of the SAME scenes:
    reset-free step + backward.  Two legs: (i) 1 thread - the reference's own execution model
    (naive rasteriser, serial SimpleVecEnv loop); (ii) env-parallel over the host cores this process may use:
    CPU_WORKER_PROCESSES worker processes x threads, one env per thread."""
    import multiprocessing as mp

    import torch as T

    T.set_num_threads(1)
    payloads = _oracle_scene_payloads(venv, range(min(n_sample, venv.num_envs)), img)
    t_tot, done_n = 0.0, 0
    for p in payloads:
        t_tot += _oracle_step(_oracle_env(p))
        done_n += 1
        if t_tot > budget_s:
            break
    one = dict(value=done_n / t_tot, n=done_n)
    cores = min(cpu_cores(), venv.num_envs)
    workers = max(1, min(CPU_WORKER_PROCESSES, cores))
    threads = max(1, cores // workers)
    per = threads * (2 if venv.num_envs >= 2 * workers * threads else 1)
    payloads = _oracle_scene_payloads(venv, range(workers * per), img)
    ctx = mp.get_context("spawn")  # never fork a process that holds a GPU context
    barrier = ctx.Barrier(workers)
    par = None
    try:
        with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=(barrier,)) as pool:
            res = pool.map(_cpu_worker, [(payloads[w * per:(w + 1) * per], threads) for w in range(workers)], chunksize=1)
        wall = max(r[1] for r in res) - min(r[0] for r in res)
        n_par = sum(r[2] for r in res)
        used = workers * threads
        par = dict(value=n_par / wall, n=n_par, cores=used, processes=workers, efficiency=(n_par / wall) / (used * one["value"]))
    except Exception as e:  # noqa: BLE001 - the baseline is a report, never a reason to lose the bench line
        print(f"[bench] all-cores CPU leg failed: {e!r}", file=sys.stderr)
    return one, par


def committed_profile(args, raster_short):
    """The committed PMC summary of the latest round's rocprofv3 passes over this same command line (HBM traffic and
    SQ_INSTS_VALU of the dominant kernel are NOT measured in this run: PMC counters need passes of their own)."""
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None, None
    for prof in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            pj = json.load(open(os.path.join(pdir, prof)))
            if pj.get("workload") == args.workload and pj.get("envs") == args.envs and pj.get("img") == args.img \
                    and pj.get("pool_models", 64) == args.pool_models and pj.get("kernel_short") == raster_short:
                return pj, prof
        except Exception:  # noqa: BLE001
            continue
    return None, None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))  # parent of the ranks: no torch import, no GPU call

    import numpy as np  # noqa: F401
    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {local_rank} has no GPU ({ndev} visible); use --dist-backend gloo to rehearse")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group("gloo")

    from occlusionenv_amd import _native as nat
    from occlusionenv_amd import rollout

    lib = nat.load()
    ppo_mode = args.workload == "ppo_rollout"
    scene_workload = "shapenet5k" if ppo_mode else args.workload
    venv, ds = build_env(scene_workload, args.envs, args.img, seed=42 + rank, pool_models=args.pool_models)
    eng = venv.engine
    if args.no_recycle:
        venv.use_recycled_outputs(0)
    if args.output_ring and eng.R:
        venv.use_output_ring(args.output_ring)
    # SURVEY.md §8d scene distribution: x2 ~ N(0,1) (np.random, seeded), az ~ U(-0.6, 0.6), el = 0; scenes pass
    # the reference's reset rejection loop (loss > 0.1, environment.py:327).  VecEnv.reset() itself draws
    # az ~ U(-40, 40) rad (SubProcVecEnv.py:233), which mostly yields non-occluding views that finish at once.
    az0 = (torch.rand(args.envs, generator=torch.Generator().manual_seed(42 + rank)) * 2 - 1) * 0.6
    obs0 = venv._reset_envs(list(range(args.envs)), az0)[:, 0]
    if eng.R:
        venv._warm_reserve()
    dev = eng.device
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    # The per-step record exchange (pack 261 floats/env + one all-gather over RCCL) runs on its own HIP stream,
    # behind an event recorded after the step's launches: it overlaps the NEXT step's render (bandwidth-bound
    # pooling + a latency-bound collective next to a VALU-bound raster kernel).  rollout.RecordExchange.
    xch = rollout.RecordExchange(args.envs, dev, world) if (world > 1 and not ppo_mode) else None
    zeros_lp = torch.zeros(args.envs, device=dev)

    def one_step():
        actions = torch.randn(args.envs, 2, device=dev, generator=gen, requires_grad=True)
        obs, rewards, dones, infos = venv.step(actions)
        rewards.sum().backward()
        if xch is not None:
            xch.submit(obs, actions, zeros_lp, rewards, dones)
            # the env's synchronous reset fallback writes reset observations into ``obs`` in place at the start of
            # the next step: it must not overtake the side stream's read of it
            venv.obs_consumer_event = xch.ready
        return actions.grad

    ppo_state = {}
    if ppo_mode:
        from occlusionenv_amd import ppo as ppo_mod

        # replicated learner: same seed on every rank, same gathered records -> identical heads, no gradient collective
        agent = ppo_mod.BatchedPPO(device=dev, seed=0, K_epochs=args.ppo_epochs)
        ppo_state.update(obs=obs0, updates=0, stats=None, rew=[], pending=False, t=0)
        if args.max_ep_len:  # trainRL.py:22,191-229: every episode ends in reset() after max_ep_len steps, done or not
            venv.max_ep_len = args.max_ep_len
            venv.stagger_ages(seed=11 + rank)
        # the per-step record exchange on its own stream: the record of step t is stored while step t + 1 renders
        pxch = rollout.RecordExchange(args.envs, dev, world, keep=True)

        def one_step():  # noqa: F811 - trainRL.py:196-216, batched
            feats, action, logprob = agent.select_action(ppo_state["obs"], gen)
            action = action.detach().requires_grad_(True)
            obs, rewards, dones, _infos = venv.step(action)
            rewards.sum().backward()  # differentiable-reward backward to the action (train_predict.py:52)
            if ppo_state["pending"]:
                agent.store(pxch.wait())
            pxch.submit(obs, action, logprob, rewards, dones, features=feats)  # PPO.py:158: the acting state's features
            venv.obs_consumer_event = pxch.ready
            ppo_state["pending"] = True
            ppo_state["obs"] = obs
            ppo_state["t"] += 1
            if ppo_state["t"] >= args.rollout_T:
                agent.store(pxch.wait())
                ppo_state["pending"], ppo_state["t"] = False, 0
                ppo_state["stats"] = agent.update()
                ppo_state["updates"] += 1
            return action.grad

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    if ppo_mode:
        # warm-up of the learner too (untimed, on a throw-away agent fed random records): the first update of a process
        # pays one-time costs - optimizer state, the GEMM library's kernel selection - that took 0.8 s against 72 ms for
        # every later update
        warm = ppo_mod.BatchedPPO(device=dev, seed=1, K_epochs=args.ppo_epochs)
        for _ in range(args.rollout_T):
            r = torch.randn(args.envs * world, rollout.RECORD_FLOATS, device=dev)
            r[:, 260] = 0.0
            warm.store(r)
        warm.update()
        del warm
        agent.records = []  # the timed region starts a fresh rollout
        ppo_state["pending"], ppo_state["t"] = False, 0
    nat.check(lib.occ_profile_enable(1), "occ_profile_enable")
    barrier()
    trace = os.environ.get("OCC_BENCH_TRACE")
    stamps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = one_step()
        if trace:
            stamps.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    if trace and rank == 0:  # host-side issue time of every step (diagnostics)
        print("[trace] per-step host ms:", " ".join("%.2f" % ((t - p) * 1e3) for p, t in zip([t0] + stamps[:-1], stamps)),
              file=sys.stderr)
    import ctypes as C

    ms_sum, launches = C.c_double(0.0), C.c_int(0)
    nat.check(lib.occ_profile_read(C.byref(ms_sum), C.byref(launches)), "occ_profile_read")
    lib.occ_profile_enable(0)
    eng.check_status()
    assert torch.isfinite(g).all()
    # fresh_outputs leg (N = 1): the same step with every output tensor allocated anew and written in full - what the
    # reference's allocator does - timed after the main region on the same env, for the reader who wants that number
    fresh = None
    if world == 1 and args.fresh_steps > 0 and not ppo_mode and eng.R and (not args.no_recycle or args.output_ring):
        del g
        venv.use_output_ring(0)
        venv.use_recycled_outputs(0)
        for _ in range(3):
            one_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.fresh_steps):
            one_step()
        barrier()
        dtf = time.perf_counter() - t1
        fresh = {"value": args.envs * args.fresh_steps / dtf, "unit": "env-steps/s", "ms_per_step": dtf / args.fresh_steps * 1e3,
                 "steps": args.fresh_steps,
                 "mode": "every step allocates obs / full_state and the combine kernel writes every pixel (no recycling, no ring)"}
    if world > 1:
        tmax = torch.tensor([dt], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    raster_short = "occ_raster2_kernel"
    raster_name = raster_short + "<soft,hard,grad>"
    if rank == 0:
        total_envs = args.envs * world
        value = total_envs * args.steps / dt
        # algorithmic bytes of one launch of the dominant kernel (= one batched step of this rank)
        vf = 0.0
        for e in venv.envs:
            for m in e._scene[0]:
                v, f = eng.pool.get(m)
                vf += v.shape[0] + f.shape[0]
        b_launch = b_alg_bytes(vf / args.envs, args.img) * args.envs
        # mean over the full-batch launches of the timed region (auto-resets render tiny batches of their own)
        avg_ms = ms_sum.value / max(launches.value, 1)
        achieved = b_launch / (avg_ms * 1e-3) / 1e9 if launches.value else None
        pj, prof = committed_profile(args, raster_short)
        traffic = traffic_src = valu = None
        if pj is not None:
            traffic = pj.get("hbm_bytes_per_launch")
            traffic_src = "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                          "2*FETCH + WRITE; not collected in this run)" % prof
            nv = pj.get("counters", {}).get("SQ_INSTS_VALU")
            if nv and launches.value:
                issue_ms = nv * VALU_CYCLES / (SIMDS * CLOCK_HZ) * 1e3
                valu = {"insts_per_launch": nv, "issue_ms": issue_ms, "frac_of_launch": issue_ms / avg_ms,
                        "model": f"SQ_INSTS_VALU x {VALU_CYCLES} cycles / ({SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz)",
                        "source": "profiles/%s (not collected in this run)" % prof}
        what = (f"PPO rollout (T={args.rollout_T}, heads-only update of {args.ppo_epochs} epochs inside the timed region"
                + (f", episode time limit {args.max_ep_len} steps" if args.max_ep_len else "") + "), " if ppo_mode else "")
        meshes = {"teapot": "3 x data/teapot.obj (2 464 faces) per env", "mixed": "3 meshes per env from a mixed 1 280 / 5 120 / 20 480-face pool",
                  "shapenet5k": "3 ShapeNet-size (~5k-face) meshes per env"}[scene_workload]
        collective = {"nccl": "RCCL", "gloo": "gloo (rehearsal: the ranks share devices, no RCCL)"}[args.dist_backend]
        # SURVEY 8d "secondary accounting" (reported, not canonical): the fragments PyTorch3D's design materialises per
        # env-step - 28 B x K x S^2 per soft render x 3 + 28 B x S^2 hard forward, 12 B x K x S^2 x 3 re-read backward
        frag_step = (28.0 * 100 * 3 + 28.0 + 12.0 * 100 * 3) * args.img * args.img
        ring_on = bool(args.output_ring and eng.R)
        recycle_on = bool(eng.R and not args.no_recycle and not ring_on)
        out_mode = (f"outputs in a ring of {args.output_ring} persistent sets (overwritten {args.output_ring} steps later)" if ring_on else
                    "recycled outputs (a released output set is reused; nothing a caller holds is ever overwritten)" if recycle_on else
                    "fresh output tensors every step")
        ms_step = dt / args.steps * 1e3
        step_frac = b_launch / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        out = {
            # BASELINE.json's metric is quoted at 128x128 (the default --img); other sizes say so
            "metric": f"env steps/sec (batched renders) @{args.img}x{args.img}, {meshes}"
                      + (f", outputs in an overwriting ring of {args.output_ring}" if ring_on else ""),
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {what}{args.envs} envs/GPU x {world} GPU, {args.img}x{args.img}, "
                                   f"3 objects/env drawn from a pool of {args.pool_models if ds is not None else 1} meshes "
                                   f"(seed 1234), K=100 soft x3 + hard RGB-D, forward + action gradient"
                                   + (f", + {collective} all-gather of 1044-B rollout records" if world > 1 else "")
                                   + "; " + out_mode,
                       "envs_per_gpu": args.envs, "img": args.img, "faces_per_pixel": 100, "pool_models": args.pool_models,
                       "output_ring": args.output_ring if ring_on else 0, "recycled_outputs": recycle_on, "collective_backend": args.dist_backend if world > 1 else None,
                       "sharding": f"env-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": raster_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         # the same algorithmic bytes over the WHOLE step (all launches, host gaps) / peak
                         "step_frac": step_frac,
                         # what the counters say the dominant kernel really moves (2 FETCH + WRITE) / its duration / peak
                         "counter_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and launches.value) else None,
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": b_launch, "avg_launch_ms": avg_ms,
                         "launches": launches.value, "valu": valu,
                         "secondary": {"label": "materialised-fragments accounting of SURVEY 8d (what PyTorch3D's design moves; "
                                                "NOT canonical, the fused path never materialises them)",
                                       "bytes_per_env_step": frag_step, "bytes_per_launch": frag_step * args.envs,
                                       "achieved": (frag_step * args.envs / (avg_ms * 1e-3) / 1e9) if launches.value else None,
                                       "unit": "GB/s", "frac": (frag_step * args.envs / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if launches.value else None},
                         "note": "VALU/latency-bound rasterisation; compulsory bytes are ~1.47 MB/env-step (SURVEY 8d)"},
        }
        if fresh is not None:
            out["fresh_outputs"] = fresh
        if ppo_mode:
            out["ppo"] = {"updates": ppo_state["updates"], "T": args.rollout_T, "epochs": args.ppo_epochs,
                          "last_update": ppo_state["stats"]}
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            one, par = cpu_baseline(venv, args.cpu_sample, args.img)
            out["cpu_baseline"] = {"value": one["value"], "unit": "env-steps/s", "cores": 1, "kind": "port",
                                   "sample": f"{one['n']} env-steps (step + backward) of the same scenes, oracle/ C naive "
                                             f"rasteriser + torch-CPU, 1 thread (the reference's execution model)"}
            if par is not None:
                out["cpu_baseline_all_cores"] = {
                    "value": par["value"], "unit": "env-steps/s", "cores": par["cores"], "kind": "port",
                    "parallel_efficiency": par["efficiency"],
                    "sample": f"{par['n']} env-steps on {par['processes']} worker processes x {par['cores'] // par['processes']} "
                              f"threads ({par['cores']} of os.cpu_count() = {os.cpu_count()} cores), one env per thread, wall "
                              f"time between the first start and the last end"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
