"""Top-level alias so the reference's callers (`from baseVecEnv import ...`: trainRL.py:9, train_predict.py:2-3)
run unchanged against the MI355X-native implementation in ``occlusionenv_amd.baseVecEnv``."""
from occlusionenv_amd.baseVecEnv import *  # noqa: F401,F403
from occlusionenv_amd import baseVecEnv as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
