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
            imgs = self.get_images(*args, **kwargs)
        except NotImplementedError:
            print("Render not defined for {}".format(self))
            return None
        big = tile_images(imgs)
        if mode == "human":
            import cv2  # noqa: WPS433 - optional dependency, as in the reference

            cv2.imshow("vecenv", big[:, :, ::-1])
            cv2.waitKey(1)
            return None
        if mode == "rgb_array":
            return big
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    def getattr_depth_check(self, name, already_found):
        if hasattr(self, name) and already_found:
            return "{0}.{1}".format(type(self).__module__, type(self).__name__)
        return None

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices


class VecEnvWrapper(VecEnv):
    """Wrapper base class delegating to ``self.venv`` (baseVecEnv.py:226-340)."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        VecEnv.__init__(self, num_envs=venv.num_envs,
                        observation_space=observation_space or venv.observation_space,
                        action_space=action_space or venv.action_space)
        self.class_attributes = dict(inspect.getmembers(self.__class__))

    def step_async(self, actions):
        self.venv.step_async(actions)

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def close(self):
        return self.venv.close()

    def render(self, *args, **kwargs):
        return self.venv.render(*args, **kwargs)

    def get_images(self):
        return self.venv.get_images()

    def get_attr(self, attr_name, indices=None):
        return self.venv.get_attr(attr_name, indices)

    def set_attr(self, attr_name, value, indices=None):
        return self.venv.set_attr(attr_name, value, indices)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return self.venv.env_method(method_name, *method_args, indices=indices, **method_kwargs)

    def __getattr__(self, name):
        blocked = self.getattr_depth_check(name, already_found=False)
        if blocked is not None:
            own = "{0}.{1}".format(type(self).__module__, type(self).__name__)
            raise AttributeError("Error: Recursive attribute lookup for {0} from {1} is ambiguous and hides "
                                 "attribute from {2}".format(name, own, blocked))
        return self.getattr_recursive(name)

    def _get_all_attributes(self):
        attrs = self.__dict__.copy()
        attrs.update(self.class_attributes)
        return attrs

    def getattr_recursive(self, name):
        if name in self._get_all_attributes():
            return getattr(self, name)
        if hasattr(self.venv, "getattr_recursive"):
            return self.venv.getattr_recursive(name)
        return getattr(self.venv, name)

    def getattr_depth_check(self, name, already_found):
        mine = name in self._get_all_attributes()
        if mine and already_found:
            return "{0}.{1}".format(type(self).__module__, type(self).__name__)
        return self.venv.getattr_depth_check(name, True if mine else already_found)


class CloudpickleWrapper(object):
    """Serialise ``var`` with cloudpickle when pickled (baseVecEnv.py:343-356)."""

    def __init__(self, var):
        self.var = var

    def __getstate__(self):
        import cloudpickle

        return cloudpickle.dumps(self.var)

    def __setstate__(self, obs):
        self.var = pickle.loads(obs)
