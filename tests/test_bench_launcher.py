"""bench.py's self-launch for --gpus N > 1: the parent builds a torch.distributed.run child
process, never touches torch or the GPU itself, relays the ranks' stdout and their exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench

    return bench


def test_defaults_follow_survey_8d():
    b = _bench()
    a = b.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.envs, a.img, a.workload) == (1, 200, 10, 1024, 128, "shapenet5k")
    a = b.parse_args(["--workload", "ppo_rollout"])  # BASELINE config 5 at its per-rank size
    assert (a.envs, a.img, a.rollout_T, a.ppo_epochs) == (256, 256, 50, 80)
    a = b.parse_args(["--workload", "ppo_rollout", "--envs", "64", "--img", "128"])
    assert (a.envs, a.img) == (64, 128)


def test_launcher_command_argv_and_env():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "50", "--warmup", "5", "--master-port", "1234", "--dist-backend", "gloo"]
    env_in = {"PATH": "/usr/bin", "RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "4", "MASTER_ADDR": "x", "MASTER_PORT": "1"}
    cmd, env = b.launcher_command(8, argv, 29511, environ=env_in)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    # bench.py's own arguments follow the script unchanged, minus the launcher's port
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "50", "--warmup", "5", "--dist-backend", "gloo"]
    assert cmd.count("--master-port") == 1
    # a stale rendezvous of an enclosing launcher must not leak into the child; dmabuf IPC for RCCL
    assert not {"RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"} & set(env)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    cmd2, _ = b.launcher_command(2, ["--gpus=2", "--master-port=77"], 5, environ={})
    assert cmd2[cmd2.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus=2"]


def test_self_launch_relays_output_and_exit_code(monkeypatch, capsys):
    b = _bench()
    line = json.dumps({"metric": "x", "value": 1.0})

    def fake(gpus, argv, port, environ=None):
        prog = f"import sys; print('noise'); print({line!r}); sys.stderr.write('warn\\n'); sys.exit(%d)"
        return [sys.executable, "-c", prog % fake.rc], dict(os.environ)

    monkeypatch.setattr(b, "launcher_command", fake)
    fake.rc = 0
    assert b.self_launch(b.parse_args(["--gpus", "2"]), ["--gpus", "2"]) == 0
    out = capsys.readouterr().out.splitlines()
    assert out == ["noise", line]
    fake.rc = 3  # a failing rank: torchrun exits non-zero, so does bench.py
    assert b.self_launch(b.parse_args(["--gpus", "2"]), ["--gpus", "2"]) == 3


def test_self_launch_without_json_line_is_an_error(monkeypatch, capsys):
    b = _bench()
    monkeypatch.setattr(b, "launcher_command", lambda g, a, p, environ=None: ([sys.executable, "-c", "print('hi')"], dict(os.environ)))
    assert b.self_launch(b.parse_args(["--gpus", "2"]), ["--gpus", "2"]) == 1
    capsys.readouterr()


def test_parent_never_imports_torch_and_starts_a_child():
    """python bench.py --gpus 2 from a cold shell: the parent must not import torch (no GPU call is possible then) and
    must start the ranks as a child; a stub torch.distributed.run on PYTHONPATH records what it was started with."""
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        pkg = os.path.join(tmp, "torch", "distributed")
        os.makedirs(pkg)
        open(os.path.join(tmp, "torch", "__init__.py"), "w").close()
        open(os.path.join(pkg, "__init__.py"), "w").close()
        with open(os.path.join(pkg, "run.py"), "w") as fh:
            fh.write("import json, os, sys\n"
                     "print(json.dumps({'argv': sys.argv[1:], 'ppid': os.getppid(), 'pid': os.getpid(),\n"
                     "                  'ws': os.environ.get('WORLD_SIZE'), 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}))\n")
        probe = ("import sys, os, runpy\n"
                 "sys.argv = ['bench.py', '--gpus', '2', '--dist-backend', 'gloo', '--envs', '512']\n"
                 "try:\n"
                 "    runpy.run_path(%r, run_name='__main__')\n"
                 "except SystemExit as e:\n"
                 "    rc = e.code\n"
                 "sys.stderr.write('PARENT %%d torch=%%s rc=%%s\\n' %% (os.getpid(), 'torch' in sys.modules, rc))\n"
                 % os.path.join(ROOT, "bench.py"))
        env = dict(os.environ, PYTHONPATH=tmp)
        env.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        parent = [l for l in r.stderr.splitlines() if l.startswith("PARENT")][-1].split()
        assert parent[2] == "torch=False" and parent[3] == "rc=0"
        assert rec["ppid"] == int(parent[1]) and rec["pid"] != rec["ppid"]  # a child process, not an exec
        assert rec["ws"] is None and rec["ipc"] == "0"
        assert "--nproc-per-node=2" in rec["argv"] and rec["argv"][-6:] == ["--gpus", "2", "--dist-backend", "gloo", "--envs", "512"]
