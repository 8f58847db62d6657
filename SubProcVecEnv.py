"""Top-level alias so the reference's callers (`from SubProcVecEnv import ...`: trainRL.py:9, train_predict.py:2-3)
run unchanged against the MI355X-native implementation in ``occlusionenv_amd.SubProcVecEnv``."""
from occlusionenv_amd.SubProcVecEnv import *  # noqa: F401,F403
from occlusionenv_amd import SubProcVecEnv as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
