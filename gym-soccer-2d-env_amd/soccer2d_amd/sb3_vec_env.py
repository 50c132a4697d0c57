"""SB3 ``VecEnv``-shaped adapter.

stable_baselines3 is not installable in the build image, so the adapter is duck-typed: it
exposes exactly the members SB3's algorithms and the reference's InfoCollectorCallback touch
(num_envs, observation_space, action_space, reset, step_async, step_wait, step, close,
get_attr, set_attr, env_method, env_is_wrapped, seed) with SB3's types: numpy float32
observations [N,10], float32 rewards, bool dones and a list of per-env info dicts that
always has ``'result'`` (falsy when nothing happened, info_collector_callback.py:26) and, for
finished envs, ``'terminal_observation'`` and ``'TimeLimit.truncated'``.  When SB3 is
importable the class also inherits from its VecEnv so isinstance checks pass.
"""
import numpy as np

from . import _capi
from .vec_env import Soccer2DVecEnv

try:  # pragma: no cover - SB3 absent in the build image
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv  # type: ignore
except Exception:  # noqa: BLE001
    _SB3VecEnv = object

_NONE_INFO = {'result': None}


class LazyInfos:
    """Sequence of info dicts built on demand: O(#finished envs) instead of O(N) per step."""

    def __init__(self, result_codes, terminal_obs):
        self._codes = result_codes            # np.uint8 [N]
        self._term = terminal_obs             # np.float32 [N,10] or None
        self._cache = {}

    def __len__(self):
        return len(self._codes)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        d = self._cache.get(i)
        if d is None:
            c = int(self._codes[i])
            d = {'result': _capi.RESULT_NAMES[c]}
            if c:
                d['terminal_observation'] = self._term[i]
                d['TimeLimit.truncated'] = c == _capi.RESULT_TIMEOUT
            self._cache[i] = d
        return d

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def finished(self):
        """Only the info dicts of envs that finished this step (what InfoCollectorCallback keeps)."""
        return [self[int(i)] for i in np.nonzero(self._codes)[0]]


class S2DSB3VecEnv(_SB3VecEnv):
    def __init__(self, num_envs, device='cuda:0', lazy_infos=False, **kwargs):
        self.venv = Soccer2DVecEnv(num_envs, device=device, auto_reset=True, **kwargs)
        self.num_envs = self.venv.num_envs
        self.observation_space = self.venv.observation_space
        self.action_space = self.venv.action_space
        self.lazy_infos = lazy_infos
        self.render_mode = None
        self._actions = None
        if _SB3VecEnv is not object:  # pragma: no cover
            super().__init__(self.num_envs, self.observation_space, self.action_space)

    def reset(self):
        return self.venv.reset().cpu().numpy()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        import torch
        a = self._actions
        if not torch.is_tensor(a):
            a = np.asarray(a)
        obs, reward, done, info = self.venv.step(a)
        codes = info['result'].cpu().numpy()
        done_np = done.cpu().numpy().astype(bool)
        term = info['terminal_observation'].cpu().numpy() if done_np.any() else None
        lazy = LazyInfos(codes, term)
        infos = lazy if self.lazy_infos else list(lazy)
        return obs.cpu().numpy(), reward.cpu().numpy(), done_np, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.venv.close()

    def seed(self, seed=None):
        return [None] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        v = getattr(self.venv, attr_name)
        return [v] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.venv, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        r = getattr(self.venv, method_name)(*args, **kwargs)
        return [r] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices
