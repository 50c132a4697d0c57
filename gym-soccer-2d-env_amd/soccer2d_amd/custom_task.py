"""User-defined tasks at tensor level -- the reference's extension point restated for a batched engine.

The reference's way to define a task is to subclass ``Soccer2DEnv`` and fill in four Python hooks that are called once
per env and cycle (soccer_2d_env.py:317-354):

    action_to_rpc_actions(action, player_state)   ->  the body command of this cycle
    state_to_observation(player_state)            ->  the observation row
    check_trainer_observation(trainer_state)      ->  (done, reward, info)
    trainer_reset_actions()                       ->  trainer commands that place ball / player for a new episode

For the built-in tasks that arithmetic is fused into the HIP kernels.  For a NEW task the same four hooks exist here, but
each is called ONCE PER STEP FOR THE WHOLE BATCH and works on device tensors: the engine simulates (rcssserver's dynamics,
``s2d_step``), hands out the protobuf-mirroring ``world_model()`` tensors (idl/service.proto field paths -> ``Tensor[N]``),
and the hooks are a handful of torch ops.  That keeps a custom task at ~0.3-0.5 G env-steps/s on 65 536 envs (a dozen small
torch launches per step) -- three orders of magnitude above the reference chain; a task that has to run at the built-in
tasks' speed gets its hooks fused as a kernel specialisation like ``reach_ball`` (csrc/s2d_device.h: action_map / observe /
judge / reward_of / reset_sample).

    env = TensorTaskEnv(65536, state_to_observation=my_obs, check_trainer_observation=my_check,
                        trainer_reset_actions=my_reset, use_continuous_action=True)       # + any ReachBallEnv kwarg
    obs = env.reset()
    obs, reward, done, info = env.step(actions)                  # SB3 VecEnv convention: finished envs are reset in the step
"""
import torch

from .vec_env import Soccer2DVecEnv



class TensorTaskEnv:
    """hooks (all optional except the two that define the task):

    * ``action_to_rpc_actions(actions, wm) -> Tensor``: caller's actions -> the engine's action tensor (``int [N]`` for a
      Discrete(n) dash-direction space, ``float [N,1]`` relative direction / 180, or ``float [N,4]`` turn / dash logits and
      angles, reach_ball_env.py:53-85).  Default: identity.
    * ``state_to_observation(wm) -> Tensor[N, obs_dim]``
    * ``check_trainer_observation(wm, env) -> (done bool[N], reward float[N], result uint8[N])``; per-env carry lives in
      tensors the hook keeps on ``env`` (``env.carry`` dict), like ``distance_to_ball`` in reach_ball_env.py:158-159.
    * ``trainer_reset_actions(env, mask)``: called after the engine has reset the masked envs with its own sampler
      (reach_ball_env.py:170-218); may overwrite state tensors in place (``env.engine.ball_x[mask] = ...`` etc.: the
      DoMoveBall / DoMovePlayer / DoRecover of idl/service.proto:1393-1433).
    """

    def __init__(self, num_envs, state_to_observation, check_trainer_observation, action_to_rpc_actions=None,
                 trainer_reset_actions=None, device='cuda:0', server_params=None, **kwargs):
        # auto_reset off: the built-in task's own done / reward outputs are simply not read; its kwargs still parameterise the
        # engine's reset sampler (reach_ball_env.py:170-218: ball grids, max_steps in the velocity test) and action space
        self.vec = Soccer2DVecEnv(num_envs, device=device, auto_reset=False, server_params=server_params, **kwargs)
        self.engine, self.num_envs, self.device = self.vec.engine, self.vec.num_envs, self.vec.device
        self.action_space = self.vec.action_space
        self._obs_fn, self._check_fn = state_to_observation, check_trainer_observation
        self._act_fn, self._reset_fn = action_to_rpc_actions, trainer_reset_actions
        self.carry = {}
        self.episode_step = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)

    def world_model(self):
        return self.vec.world_model(derived=True)

    def _after_reset(self, mask):
        if self._reset_fn is not None:
            self._reset_fn(self, mask)
        if mask is None:
            self.episode_step.zero_()
        else:
            self.episode_step[mask] = 0

    def reset(self, mask=None):
        self.vec.reset(mask)
        self._after_reset(None if mask is None else torch.as_tensor(mask, device=self.device).bool())
        wm = self.world_model()
        self._check_fn(wm, self)                          # seeds the hook's carry; outputs dropped (reach_ball_env.py:166)
        return self._obs_fn(wm)

    def step(self, actions):
        wm = None
        if self._act_fn is not None:
            wm = self.world_model()
            actions = self._act_fn(actions, wm)
        self.vec.step(actions)
        self.episode_step += 1
        wm = self.world_model()
        obs = self._obs_fn(wm)
        done, reward, result = self._check_fn(wm, self)
        info = {'result': result, 'terminal_observation': obs}
        if bool(done.any()):                              # SB3 VecEnv convention: reset inside the step
            info['terminal_observation'] = obs.clone()
            self.vec.reset(done.to(torch.uint8))
            self._after_reset(done)
            wm2 = self.world_model()
            keep = {k: v.clone() for k, v in self.carry.items()}
            self._check_fn(wm2, self)                     # seeds the carry of the new episodes ...
            for k, v in keep.items():                     # ... and only of those
                self.carry[k] = torch.where(done.reshape((-1,) + (1,) * (v.dim() - 1)), self.carry[k], v)
            obs = torch.where(done[:, None], self._obs_fn(wm2), obs)
        return obs, reward, done, info

    def close(self):
        self.vec.close()
