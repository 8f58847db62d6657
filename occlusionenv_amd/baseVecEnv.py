"""Drop-in ``baseVecEnv`` module: the abstract vectorised-environment contract the reference takes from
stable-baselines (/root/reference/baseVecEnv.py:57-356).  Same class and method names, argument meaning
and error types; the bodies are this project's own.
"""
from __future__ import annotations

import inspect
import pickle
from abc import ABC, abstractmethod
from typing import List, Optional, Sequence, Union

import numpy as np


def tile_images(img_nhwc):
    """Arrange N images (N,h,w,c) on a P x Q grid, P = ceil(sqrt(N)), Q = ceil(N / P); unused cells are
    zero.  Returns (P*h, Q*w, c).  (baseVecEnv.py:9-32)"""
    imgs = np.asarray(img_nhwc)
    n, h, w, c = imgs.shape
    rows = int(np.ceil(np.sqrt(n)))
    cols = int(np.ceil(float(n) / rows))
    canvas = np.zeros((rows * cols, h, w, c), dtype=imgs.dtype)
    canvas[:n] = imgs
    grid = canvas.reshape(rows, cols, h, w, c).transpose(0, 2, 1, 3, 4)
    return grid.reshape(rows * h, cols * w, c)


class AlreadySteppingError(Exception):
    """step_async() called while a step is already pending (baseVecEnv.py:35-43)."""

    def __init__(self):
        super().__init__("already running an async step")


class NotSteppingError(Exception):
    """step_wait() called with no pending step (baseVecEnv.py:46-54)."""

    def __init__(self):
        super().__init__("not running an async step")


class VecEnv(ABC):
    """Abstract asynchronous vectorised environment (baseVecEnv.py:57-172)."""

    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        """Reset every environment; returns the stacked observations."""

    @abstractmethod
    def step_async(self, actions):
        """Start a step with the given actions; collect it with step_wait()."""

    @abstractmethod
    def step_wait(self):
        """Returns (observations, rewards, dones, infos) of the pending step."""

    @abstractmethod
    def close(self):
        """Release resources."""

    @abstractmethod
    def get_attr(self, attr_name, indices=None):
        """List of ``attr_name`` of the selected envs."""

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None):
        """Assign ``attr_name`` in the selected envs."""

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """Call a method of the selected envs; list of results."""

    @abstractmethod
    def seed(self, seed: Optional[int] = None) -> List[Union[None, int]]:
        """Seed env i with seed + i."""

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self, *args, **kwargs) -> Sequence[np.ndarray]:
        raise NotImplementedError

    def render(self, mode: str, *args, **kwargs):
        try:
            frames = self.get_images(*args, **kwargs)
        except NotImplementedError:
            print("Render not defined for {}".format(self))
            return None
        mosaic = tile_images(frames)
        if mode == "rgb_array":
            return mosaic
        if mode != "human":
            raise NotImplementedError
        import cv2  # optional dependency, as in the reference

        cv2.imshow("vecenv", mosaic[:, :, ::-1])  # RGB -> BGR
        cv2.waitKey(1)
        return None

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    def getattr_depth_check(self, name, already_found):
        return _qualified(self) if (already_found and hasattr(self, name)) else None

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices


def _qualified(obj) -> str:
    cls = type(obj)
    return "%s.%s" % (cls.__module__, cls.__name__)


class VecEnvWrapper(VecEnv):
    """Base class of wrappers around another VecEnv held in ``self.venv`` (baseVecEnv.py:226-340): everything
    that is not overridden is forwarded; attribute lookups fall through the wrapper chain, and a name that two
    levels of the chain both define is reported as ambiguous instead of silently shadowed."""

    #: methods forwarded verbatim to the wrapped env (generated below)
    _FORWARDED = ("step_async", "seed", "close", "render", "get_images", "get_attr", "set_attr", "env_method")

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(venv.num_envs,
                         venv.observation_space if observation_space is None else observation_space,
                         venv.action_space if action_space is None else action_space)
        self.class_attributes = dict(inspect.getmembers(type(self)))

    @abstractmethod
    def reset(self):
        """Wrappers define what a reset returns."""

    @abstractmethod
    def step_wait(self):
        """Wrappers define what a finished step returns."""

    # ---- attribute resolution through the chain ------------------------------------------------
    def _get_all_attributes(self):
        merged = dict(self.__dict__)
        merged.update(self.class_attributes)
        return merged

    def __getattr__(self, name):
        # only reached when normal lookup failed on this wrapper
        owner = self.getattr_depth_check(name, already_found=False)
        if owner is not None:
            raise AttributeError("Error: Recursive attribute lookup for {0} from {1} is ambiguous and hides "
                                 "attribute from {2}".format(name, _qualified(self), owner))
        return self.getattr_recursive(name)

    def getattr_recursive(self, name):
        if name in self._get_all_attributes():
            return getattr(self, name)
        inner = self.venv
        return inner.getattr_recursive(name) if hasattr(inner, "getattr_recursive") else getattr(inner, name)

    def getattr_depth_check(self, name, already_found):
        here = name in self._get_all_attributes()
        if here and already_found:
            return _qualified(self)
        return self.venv.getattr_depth_check(name, already_found or here)


def _make_forwarder(method_name):
    def forward(self, *args, **kwargs):
        return getattr(self.venv, method_name)(*args, **kwargs)

    forward.__name__ = method_name
    forward.__doc__ = "Forwarded to ``self.venv.%s``." % method_name
    return forward


for _name in VecEnvWrapper._FORWARDED:
    setattr(VecEnvWrapper, _name, _make_forwarder(_name))
VecEnvWrapper.__abstractmethods__ = frozenset({"reset", "step_wait"})
del _name


class CloudpickleWrapper(object):
    """Carries ``var`` across process boundaries with cloudpickle (closures pickle; baseVecEnv.py:343-356)."""

    def __init__(self, var):
        self.var = var

    def __getstate__(self):
        from cloudpickle import dumps

        return dumps(self.var)

    def __setstate__(self, blob):
        self.var = pickle.loads(blob)
