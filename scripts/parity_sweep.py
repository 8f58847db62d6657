"""Randomised parity sweep (GPU box): many seeded scenes through the HIP engine and the CPU oracle.
    python scripts/parity_sweep.py [n_seeds [first_seed [wide]]]     -> one line per case + a summary; exit 1 on a violation.
"wide" also varies K (100 / 50 / 8 faces per pixel), the image side (64 ... 160) and the camera distance (1.3 ... 6).
"near" = wide cases with the camera distance restricted to 2.5 / 1.3 / 2.0 / 1.6 (the camera inside the scene: near,
z-clipped faces - where round 4's one violation lived).
"stage" (wide cases) checks the RASTER STAGE ALONE: the oracle's rasteriser + blend run on the engine's own face records
(identical geometry, no projection noise between the sides) against the engine's silhouettes, at 1e-5, no classifier.
A case is "ok" under EXACTLY the criterion of the parity tests (tests/parity_utils.py: violations): 1e-4 everywhere,
every pixel beyond it an oracle-verified exact tie, loss / reward / gradient at 1e-4 with the ties weighted out."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.parity_utils import run_parity_case, violations  # noqa: E402


def case_of(seed):
    mesh = ("teapot", "synthetic", "mixed", "textured")[seed % 4]
    img = (64, 96, 128)[seed % 3] if mesh != "mixed" else 64
    az = (0.6, 3.0)[seed % 2]
    radius = (4.0, 4.0, 4.0, 1.3)[(seed // 4) % 4] if mesh in ("teapot", "textured") else 4.0
    return dict(n_env=2, img=img, seed=seed, mesh=mesh, az_range=az, radius=radius)


def case_of_wide(seed):
    """More of the parameter space per case than case_of: K, image side and camera distance vary independently."""
    mesh = ("teapot", "synthetic", "mixed", "textured")[seed % 4]
    img = (64, 96, 128, 160)[(seed // 4) % 4] if mesh != "mixed" else (64, 96)[(seed // 4) % 2]
    K = (100, 50, 8)[(seed // 16) % 3]
    radius = (4.0, 2.5, 1.3, 6.0)[(seed // 48) % 4] if mesh != "mixed" else (4.0, 6.0)[(seed // 48) % 2]
    return dict(n_env=2, img=img, seed=seed, mesh=mesh, az_range=(0.6, 3.0)[seed % 2], radius=radius, faces_per_pixel=K)


def case_of_near(seed):
    """case_of_wide with the camera always close: radius 2.5 / 1.3 / 2.0 / 1.6 (objects reach out to z = 2)."""
    c = case_of_wide(seed)
    c["radius"] = (2.5, 1.3, 2.0, 1.6)[(seed // 48) % 4]
    return c


def stage_sweep(n, base):
    import torch

    import collections

    from tests.parity_utils import RecordFaces as _RecFaces, alpha_of_records, explain_soft, make_case, run_engine

    reasons = collections.Counter()
    t0, tot_beyond, tot_pix, worst_rest, cases_beyond = time.time(), 0, 0, 0.0, 0
    for seed in range(base, base + n):
        c = case_of_wide(seed)
        K = c["faces_per_pixel"]
        got = run_engine(make_case(c["n_env"], seed, c["mesh"], c["az_range"]), c["img"], radius=c["radius"], faces_per_pixel=K)
        beyond, rest, npx = 0, 0.0, 0
        for phase, al in (("records0", got["alphas0"]), ("records", got["alphas"])):
            for eo, rec in enumerate(got[phase]):
                d = (alpha_of_records(rec, c["img"], K) - al[eo // 3, eo % 3]).abs()
                beyond += int((d > 1e-5).sum())
                for y, x in torch.nonzero(d > 1e-5).tolist():  # every outlier must be a near-tie of the records' own geometry
                    why = explain_soft(_RecFaces(rec), c["img"], y, x, K)
                    reasons[tuple(why) if why else ("UNEXPLAINED",)] += 1
                rest = max(rest, float(torch.where(d > 1e-5, torch.zeros(()), d).max()))
                npx += d.numel()
        tot_beyond, tot_pix, worst_rest = tot_beyond + beyond, tot_pix + npx, max(worst_rest, rest)
        cases_beyond += 1 if beyond else 0
        print("seed %d %-9s %3d K %3d r %.1f  pixels beyond 1e-5: %d of %d  (largest of the others %.1e)" % (
            seed, c["mesh"], c["img"], K, c["radius"], beyond, npx, rest), flush=True)
    print("stage sweep: cases %d  object-pixels %d  beyond 1e-5: %d (in %d cases: depth ties at a K boundary)  largest of the others %.2e  %.0f s" % (
        n, tot_pix, tot_beyond, cases_beyond, worst_rest, time.time() - t0))
    print("reasons of the pixels beyond 1e-5 (tie classifier of tests/parity_utils.py on the records' own geometry):", dict(reasons))
    return 0 if tot_beyond <= 3 * n and not reasons.get(("UNEXPLAINED",), 0) else 1


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    base = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    if len(sys.argv) > 3 and sys.argv[3] == "stage":
        sys.exit(stage_sweep(n, base))
    wide = len(sys.argv) > 3 and sys.argv[3] in ("wide", "near")
    near = len(sys.argv) > 3 and sys.argv[3] == "near"
    import collections

    from tests import parity_utils as pu

    pu.REASON_LOG = []  # (kind, object, y, x, reasons) of every pixel the classifier accepted
    worst, bad, ties, t0 = {}, 0, 0, time.time()
    tot_pix = 0
    for seed in range(base, base + n):
        c = (case_of_near(seed) if near else case_of_wide(seed)) if wide else case_of(seed)
        res = run_parity_case(**c)
        tot_pix += 2 * c["n_env"] * 3 * c["img"] * c["img"]  # reset render + step render, three objects
        v = violations(res)
        bad += 1 if v else 0
        ties += 1 if res["tie_pixels"] else 0
        for k, val in res.items():
            if isinstance(val, float):
                worst[k] = max(worst.get(k, 0.0), val)
        print("seed %d %-9s %3d K %3d az %.1f r %.1f  %s  alpha %.1e obs %.1e loss %.1e reward %.1e grad %.1e ties %d %s" % (
            seed, c["mesh"], c["img"], c.get("faces_per_pixel", 100), c["az_range"], c["radius"], "BAD" if v else "ok ", res["alpha_maxabs"],
            res["obs_maxabs"], res["loss_rel"], res["reward_abs"], res["grad_rel"], res["tie_pixels"],
            ("arbiter " + " ".join("gpu %.0f orc32 %.0f eps*M (floor: %s)" % (a["e_gpu"] / max(2.0 ** -24 * a["mass"], 1e-300), a["e_orc32"] / max(2.0 ** -24 * a["mass"], 1e-300), a.get("floor", "model")) for a in res["grad_arbiter"]) + " " if res["grad_arbiter"] else "")
            + ("gradient ties %d %s " % (res["grad_tie_pixels"], res.get("grad_tie_reasons")) if res.get("grad_tie_pixels") else "")
            + "; ".join(v)), flush=True)
    print("cases %d  violations %d  cases with tie pixels %d  %.0f s  worst: %s" % (
        n, bad, ties, time.time() - t0, " ".join("%s=%.2e" % kv for kv in sorted(worst.items()))))
    # the exception rate: accepted pixels per acceptance rule (a pixel may carry several reasons; each is counted)
    per_rule, per_kind = collections.Counter(), collections.Counter()
    for kind, _o, _y, _x, reasons in pu.REASON_LOG:
        per_kind[kind] += 1
        for r in reasons:
            per_rule[r] += 1
    print("accepted pixels: %d of %d object-pixels (%.2e) by comparison: %s" % (
        len(pu.REASON_LOG), tot_pix, len(pu.REASON_LOG) / max(tot_pix, 1), dict(per_kind)))
    for r, n_r in per_rule.most_common():
        print("  rule %-90s %6d pixels" % (r, n_r))
    sys.exit(1 if bad else 0)
