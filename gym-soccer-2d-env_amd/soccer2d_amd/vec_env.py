"""Soccer2DVecEnv -- the vectorised, device-tensor form of the reference's gym surface.

Mirrors soccer_2d_env.Soccer2DEnv.step/reset/close (soccer_2d_env.py:218-299) for N
independent reach_ball matches in lockstep on one MI355X: ``reset() -> obs[N,10]``,
``step(actions[N]) -> (obs[N,10], reward[N], done[N], info)``.  Everything stays on the
device; ``info`` is a dict of TENSORS (``result`` codes, ``terminal_observation``) -- the
reference's per-env ``{'result': 'Goal'|'Out'|'Timeout'|None}`` dicts are materialised
lazily (``infos()``), because building 65 536 Python dicts per step would dominate.
Auto-reset follows the SB3 VecEnv convention (the row returned for a finished env is the
first observation of its next episode; the last one of the old episode is in
``info['terminal_observation']``).
"""
import torch

from . import _capi
from .engine import Engine, make_config
from .spaces import reach_ball_spaces
from .state_view import StateView, world_model_tensors

RESULT_NAMES = _capi.RESULT_NAMES


class Soccer2DVecEnv:
    metadata = {'render.modes': ['human']}

    def __init__(self, num_envs, device='cuda:0', seed=0x5EED, env_id_offset=0, auto_reset=True, noise=True,
                 server_params=None, clone_outputs=False, **kwargs):
        cfg = make_config(seed=seed, env_id_offset=env_id_offset, auto_reset=auto_reset, noise=noise,
                          server_params=server_params, **kwargs)
        self.engine = Engine(num_envs, device, cfg=cfg)
        self.num_envs = self.engine.num_envs
        self.device = self.engine.device
        t = cfg.task
        self.action_space, self.observation_space = reach_ball_spaces(
            bool(t.use_continuous_action), bool(t.use_turning), int(t.action_space_size))
        self.clone_outputs = clone_outputs
        self.auto_reset = bool(auto_reset)
        self._closed = False

    # -- gym surface ----------------------------------------------------------------------
    def reset(self, mask=None):
        """Tensor[N,10] float32 (view of engine memory, valid until the next step/reset)."""
        obs = self.engine.reset(mask)
        return obs.clone() if self.clone_outputs else obs

    def step(self, actions=None):
        """actions: int tensor [N] (Discrete) | float [N,1] | float [N,4]; None = random policy."""
        if isinstance(actions, tuple):        # model.predict() returns (action, state): dqn_stable_baselines3.py:48-49
            actions = actions[0]
        obs, reward, done, result = self.engine.step(actions)
        info = {'result': result, 'terminal_observation': self.engine.terminal_obs}
        if self.clone_outputs:
            obs, reward, done = obs.clone(), reward.clone(), done.clone()
            info = {k: v.clone() for k, v in info.items()}
        return obs, reward, done, info

    def rollout(self, n_steps, actions=None, out=None, with_obs=True):
        return self.engine.rollout(n_steps, actions=actions, out=out, with_obs=with_obs)

    def render(self, mode='human'):      # soccer_2d_env.py:271-278: no-op
        return None

    def close(self):
        if not self._closed:
            self.engine.close()
            self._closed = True

    def seed(self, seed=None):
        """gym's env.seed(): a new Philox key for every later draw (resets, in-engine policy, noise), followed by a reset of
        all envs so that the next episodes start from the new stream.  The reference never seeds its `random` / `np.random`
        (reach_ball_env.py:71, 173-205); here the same seed gives the same trajectories on any shard layout."""
        if seed is None:
            seed = int(torch.seed()) & 0xFFFFFFFFFFFFFFFF
        self.engine.set_seed(seed)
        # the draw counters are part of the stream position (episode index: reset sampler; policy_step: policy / noise)
        self.engine.episode.zero_()
        self.engine.policy_step.zero_()
        self.reset()
        return [int(seed)]

    # -- reference-style views -------------------------------------------------------------
    def infos(self, result=None):
        """List of the reference's info dicts for the LAST step (O(N) Python; use sparingly)."""
        r = (self.engine.result if result is None else result).cpu().tolist()
        return [{'result': RESULT_NAMES[c]} for c in r]

    def world_model(self, derived=True):
        """dict: protobuf field path -> device tensor (idl/service.proto State.world_model.*)."""
        return world_model_tensors(self.engine, derived=derived)

    def state(self, index=0):
        """pb2.State-like snapshot of one env (attribute access, host scalars)."""
        return StateView(self.world_model(), index)

    @property
    def stats(self):
        s = self.engine.stats.cpu().tolist()
        return {'env_steps': s[0], 'Goal': s[1], 'Out': s[2], 'Timeout': s[3]}

    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, sd):
        self.engine.load_state_dict(sd)
