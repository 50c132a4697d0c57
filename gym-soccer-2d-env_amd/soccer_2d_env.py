"""Drop-in for the reference module ``soccer_2d_env`` (same class name, same gym surface).

Reference: Soccer2DEnv spawns rcssserver + a C++ proxy + a gRPC server and exchanges one
protobuf State per cycle over 4 queues (soccer_2d_env.py:30-95, 226-269).  Here the same
``reset() / step() / close() / render()`` surface sits directly on the in-process HIP engine
(libs2d_hip.so): no child processes, no port 50051, no sleeps.

Two ways to define a task, chosen by what the subclass does:
  * the built-in tasks (sample_environments/reach_ball_env.py) set ``task_kwargs`` and override nothing: the arithmetic of the
    reference's four task hooks is fused into the kernels (the fast path, batched as ``Soccer2DVecEnv``);
  * a subclass that OVERRIDES the reference's hooks -- action_to_rpc_actions / state_to_observation /
    check_trainer_observation / trainer_reset_actions (soccer_2d_env.py:317-354) -- runs them as the reference does: called per
    cycle with pb2.State-like ``StateView`` objects, returning ``service_pb2`` messages (soccer2d_amd/hook_env.py; the slow
    path: one launch plus host-side Python per cycle).  A task env written for the reference imports unchanged.
"""
import logging

import numpy as np

from soccer2d_amd import _capi
from soccer2d_amd.vec_env import Soccer2DVecEnv

try:  # gym is optional (absent in the build image); SB3 only needs the duck-typed surface
    import gymnasium as _gym  # type: ignore
    _Base = _gym.Env
except Exception:  # noqa: BLE001
    try:
        import gym as _gym  # type: ignore
        _Base = _gym.Env
    except Exception:  # noqa: BLE001
        _Base = object


_HOOK_NAMES = ('action_to_rpc_actions', 'state_to_observation', 'check_trainer_observation', 'trainer_reset_actions')
_defer_hook_runtime = [False]       # set by soccer2d_amd.hook_env.HookVecEnv while it builds the instances of ONE shared engine


class Soccer2DEnv(_Base):
    """Single-env view (num_envs = 1) of the batched engine with the reference's return types:
    ``reset() -> np.ndarray[10]`` (obs only, soccer_2d_env.py:218-224) and
    ``step(a) -> (obs, float reward, bool done, {'result': None|'Goal'|'Out'|'Timeout'})``."""
    metadata = {'render.modes': ['human']}
    task_kwargs = {}

    def __init__(self, render_mode=None, run_grpc_server=True, run_rcssserver=True, run_trainer_player=True,
                 logger=None, log_dir=None, device='cuda:0', seed=0x5EED, noise=True, server_params=None,
                 **kwargs):
        # run_* switches are accepted for signature compatibility; there is nothing to spawn.
        self.render_mode = render_mode
        self.log_dir = log_dir
        self.logger = logger or logging.getLogger(type(self).__name__)
        self.logger.info('Initializing %s on the in-process HIP engine...', type(self).__name__)
        self._latest_player_state = None
        self._latest_trainer_state = None
        self._hooks, self._hook_index, self.vec = None, 0, None
        if self.overrides_task_hooks():                    # the reference's plugin protocol: Python hooks per cycle
            if kwargs:
                raise TypeError(f"unexpected keyword arguments for a hook-based env: {sorted(kwargs)}")
            if not _defer_hook_runtime[0]:
                from soccer2d_amd.hook_env import HookRuntime
                self._hooks = HookRuntime([self], device=device, seed=seed, noise=noise, server_params=server_params)
            return
        kw = dict(self.task_kwargs)
        kw.update(kwargs)
        # reference flow: the caller resets after done (dqn_stable_baselines3.py:52-56)
        self.vec = Soccer2DVecEnv(1, device=device, seed=seed, auto_reset=False, noise=noise,
                                  server_params=server_params, **kw)
        self.action_space = self.vec.action_space
        self.observation_space = self.vec.observation_space

    @classmethod
    def overrides_task_hooks(cls):
        return any(getattr(cls, h) is not getattr(Soccer2DEnv, h) for h in _HOOK_NAMES)

    # ---- the reference's task hooks (soccer_2d_env.py:317-354); a subclass that overrides them selects the hook path
    def action_to_rpc_actions(self, action, player_state):
        raise NotImplementedError('built-in tasks decode actions in the kernel; override the four hooks for a custom task')

    def state_to_observation(self, state):
        raise NotImplementedError

    def check_trainer_observation(self, observation):
        raise NotImplementedError

    def trainer_reset_actions(self):
        raise NotImplementedError

    def env_reset(self):
        """soccer_2d_env.py:179-206 (hook path): trainer reset actions, one command-less cycle -> (player observation, trainer state)"""
        if self._hooks is None:
            raise RuntimeError('env_reset() belongs to the hook path (a subclass overriding the four task hooks)')
        return self._hooks.env_reset(self._hook_index)

    def abs_reset(self):
        """soccer_2d_env.py:208-216"""
        player_observation, _trainer_observation = self.env_reset()
        return player_observation

    # reference: float64 ndarray although the space says float32 (reach_ball_env.py:98-111)
    def _obs(self, t):
        return t[0].detach().cpu().numpy().astype(np.float64)

    def reset(self):
        if self.vec is None:                               # hook path: soccer_2d_env.py:218-224
            return self.abs_reset()
        obs = self._obs(self.vec.reset())
        self._latest_player_state = self._latest_trainer_state = None
        return obs

    def step(self, action):
        if isinstance(action, tuple):            # (action, state) from model.predict (dqn_stable_baselines3.py:48-49)
            action = action[0]
        if self.vec is None:                               # hook path: soccer_2d_env.py:226-269
            return self._hooks.step({self._hook_index: action})[self._hook_index]
        a = np.asarray(action)
        t = self.vec.engine.cfg.task
        if not t.use_continuous_action:
            a = a.reshape(-1)[:1].astype(np.int64)
        else:
            a = a.reshape(1, -1).astype(np.float32)
        obs, reward, done, info = self.vec.step(a)
        code = int(info['result'][0].item())
        self._latest_player_state = self._latest_trainer_state = None
        return self._obs(obs), float(reward[0].item()), bool(done[0].item()), {'result': _capi.RESULT_NAMES[code]}

    def render(self, mode='human'):
        return None

    def close(self):
        self.logger.info('Closing %s...', type(self).__name__)
        if self.vec is not None:
            self.vec.close()
        elif self._hooks is not None and self._hooks.envs[0] is self and len(self._hooks.envs) == 1:
            self._hooks.close()

    # pb2.State-like snapshots (the reference keeps _latest_player_state/_latest_trainer_state)
    @property
    def latest_player_state(self):
        if self._latest_player_state is None and self.vec is not None:
            self._latest_player_state = self.vec.state(0)
        return self._latest_player_state

    latest_trainer_state = latest_player_state
