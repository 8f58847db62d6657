"""Bitwise A/B of two library builds on a few scenes (GPU box): python scripts/dbg/ab_outputs.py libA.so libB.so
Every output of reset + step (+ render) must be identical."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(2, 64, 5, "teapot", 4.0), (3, 128, 6, "synthetic", 4.0), (1, 72, 7, "teapot", 4.0), (2, 256, 8, "synthetic", 4.0), (2, 96, 9, "textured", 1.3)]
KEYS = ("obs0", "alphas0", "fs0", "loss0", "obs", "alphas", "fs", "loss", "reward", "grad", "render")
CODE = ("import sys, torch; sys.path.insert(0, %r); from tests.parity_utils import make_case, run_engine\n"
        "res = []\n"
        "for n, img, seed, mesh, radius in %r:\n"
        "    got = run_engine(make_case(n, seed, mesh), img, radius=radius, render_too=True)\n"
        "    res.append({k: got[k].detach().cpu() for k in %r})\n"
        "torch.save(res, sys.argv[1])\n" % (ROOT, CASES, KEYS))
import torch
outs = []
with tempfile.TemporaryDirectory() as td:
    for i, lib in enumerate(sys.argv[1:3]):
        path = os.path.join(td, "o%d.pt" % i)
        r = subprocess.run([sys.executable, "-c", CODE, path], env=dict(os.environ, OCC_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
bad = 0
for case, a, b in zip(CASES, outs[0], outs[1]):
    for k in KEYS:
        if not torch.equal(a[k], b[k]):
            bad += 1
            print("DIFFERENT", case, k, float((a[k] - b[k]).abs().max()))
print("bitwise A/B over %d cases x %d outputs: %s" % (len(CASES), len(KEYS), "IDENTICAL" if not bad else "%d differ" % bad))
