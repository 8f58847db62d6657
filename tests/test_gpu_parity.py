"""GPU parity: HIP engine vs the CPU oracle on identical seeded scenes (tolerance 1e-4 fp32, BASELINE.json)."""
import pytest
import torch

from tests.parity_utils import run_parity_case

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _check(res, grad_tol=1e-3):
    assert res["depth_mismatch"] < 2e-3, res
    assert res["obs_maxabs"] < TOL and res["obs0_maxabs"] < TOL, res
    assert res["alpha_maxabs"] < TOL and res["fs_maxabs"] < 3 * TOL, res
    assert res["loss_rel"] < TOL and res["loss0_rel"] < TOL, res
    assert res["reward_abs"] < TOL, res
    assert res["grad_rel"] < grad_tol, res


def test_teapot_64():
    _check(run_parity_case(n_env=3, img=64, seed=0, mesh="teapot"))


def test_teapot_128():
    _check(run_parity_case(n_env=2, img=128, seed=1, mesh="teapot"))


def test_synthetic_5k_64_topk_overflow():
    # 5k-face meshes on a small screen: interior pixels have > K=100 candidates (exact top-K path)
    _check(run_parity_case(n_env=2, img=64, seed=2, mesh="synthetic"))


def test_synthetic_5k_128():
    _check(run_parity_case(n_env=2, img=128, seed=3, mesh="synthetic"))
