"""The reference's task-hook protocol on the HIP engine: a ``Soccer2DEnv`` subclass that overrides
``action_to_rpc_actions / state_to_observation / check_trainer_observation / trainer_reset_actions``
(soccer_2d_env.py:317-354) runs unchanged -- its hooks are called with pb2.State-like ``StateView`` objects and return
``service_pb2`` messages, exactly as in the reference -- while the cycle itself is one ``s2d_step`` launch
(``S2D_ACT_COMMAND``: the decoded body command of every env, executed as it is).

This is the SLOW path, for task envs written against the reference: every step crosses the host (one launch, one
world-model kernel, ~40 small device-to-host copies, then Python hooks per env) -- a few thousand env-steps/s, against
~10^10 for the fused tasks.  ``HookRuntime`` drives 1..64 instances of such an env on ONE engine (env i = instance i);
an env that resets consumes its command-less cycle (soccer_2d_env.py:186-197) while the others are frozen
(``S2D_CMD_FREEZE``), so instances keep their own episode clocks like separate reference processes would.

What a hook may return: ``PlayerAction{dash | turn}`` (executed), ``body_hold_ball`` or an empty action (no body command:
the ball is never kickable for the filler, soccer_2d_env.py:190), a list of those (the first body command counts: one
command per cycle, server.py:56-60); kick / tackle / catch / move and the helios behaviours need the 11v11 engine or the
C++ proxy and raise.  ``TrainerAction``: do_move_ball, do_move_player (our_side, unum 1), do_recover, do_change_mode
(PlayOn is forced anyway, soccer_2d_env.py:242), do_kick_off (ignored: no referee in coach mode).
"""
import numpy as np
import torch

from . import _capi
from .engine import Engine, make_config
from .state_view import StateView, world_model_tensors

MAX_HOOK_ENVS = 64


def command_of(action):
    """(S2D_CMD_*, power, relative direction) of what a hook returned (PlayerAction, list of them, or None)."""
    if action is None:
        return (_capi.CMD_NONE, 0.0, 0.0)
    if isinstance(action, (list, tuple)):
        for a in action:
            c = command_of(a)
            if c[0] != _capi.CMD_NONE:
                return c
        return (_capi.CMD_NONE, 0.0, 0.0)
    which = action.WhichOneof('action')
    if which is None or which == 'body_hold_ball':
        return (_capi.CMD_NONE, 0.0, 0.0)
    if which == 'dash':
        return (_capi.CMD_DASH, float(action.dash.power), float(action.dash.relative_direction))
    if which == 'turn':
        return (_capi.CMD_TURN, 0.0, float(action.turn.relative_direction))
    if which in ('turn_neck', 'change_view'):            # vision commands: a full-state engine has no vision model
        return (_capi.CMD_NONE, 0.0, 0.0)
    raise NotImplementedError(f"PlayerAction.{which}: the one-player reach engine executes dash / turn; kick, tackle, catch and move "
                              f"are commands of the 11v11 engine (soccer2d_amd.match), helios behaviours need the C++ proxy")


class HookRuntime:
    """1..64 hook-based env instances on one engine."""

    def __init__(self, envs, device='cuda:0', seed=0x5EED, noise=True, server_params=None):
        self.envs = list(envs)
        n = len(self.envs)
        if not 1 <= n <= MAX_HOOK_ENVS:
            raise ValueError(f"the hook path runs 1..{MAX_HOOK_ENVS} envs, got {n}")
        # the engine's own task logic is not used: no auto-reset, no episode end of its own making
        cfg = make_config(seed=seed, auto_reset=False, noise=noise, server_params=server_params, use_continuous_action=False,
                          action_space_size=16, max_steps=2000000000, min_distance_to_ball=0.0)
        self.engine = Engine(n, device, cfg=cfg)
        # Connect state of every instance = what s2d_create leaves behind (s2d_init_kernel): a player that has just joined -- full
        # stamina, effort / recovery / capacity at their initial values (rcssserver's state after a (recover)), at rest at the
        # centre spot with body 0, the ball beside it.  A task whose trainer_reset_actions() omits do_recover or do_move_player
        # is legal against rcssserver and starts from this state here (tests/test_gpu_dropin.py::test_hook_env_without_recover).
        self.n = n
        self._cmd = np.zeros((n, 4), dtype=np.float32)
        self.states = [None] * n

    def close(self):
        self.engine.close()

    # ------------------------------------------------------------------ state snapshots
    def snapshot(self):
        """pb2.State-like views of every env, from ONE pass over the device state (host copies of each field)."""
        wm = {k: v.detach().cpu().numpy() for k, v in world_model_tensors(self.engine).items()}
        self.states = [StateView(wm, i) for i in range(self.n)]
        return self.states

    # ------------------------------------------------------------------ trainer
    def apply_trainer_actions(self, i, actions):
        e, sp = self.engine, self.engine.cfg.sp
        if not isinstance(actions, (list, tuple)):
            actions = [actions]
        for a in actions:
            which = a.WhichOneof('action')
            if which == 'do_move_ball':                    # (move (ball) x y 0 vx vy)
                m = a.do_move_ball
                e.ball_x[i], e.ball_y[i] = float(m.position.x), float(m.position.y)
                e.ball_vx[i], e.ball_vy[i] = float(m.velocity.x), float(m.velocity.y)
            elif which == 'do_move_player':                # (move (player T U) x y dir): position, body; at rest
                m = a.do_move_player
                if not m.our_side or int(m.uniform_number) != 1:
                    raise ValueError("the reach engine has one player: our_side=True, uniform_number=1 (reach_ball_env.py:190-192)")
                body = float(np.float32(m.body_direction))
                if body < -360.0 or body > 360.0:
                    body = float(np.fmod(np.float32(body), np.float32(360.0)))
                body = body + 360.0 if body < -180.0 else (body - 360.0 if body > 180.0 else body)
                e.player_x[i], e.player_y[i], e.player_body[i] = float(m.position.x), float(m.position.y), body
                e.player_vx[i], e.player_vy[i] = 0.0, 0.0
            elif which == 'do_recover':                    # (recover)
                e.stamina[i], e.effort[i] = float(sp.stamina_max), float(sp.effort_init)
                e.recovery[i], e.stamina_capacity[i] = float(sp.recover_init), float(sp.stamina_capacity)
            elif which in ('do_change_mode', 'do_kick_off', None):
                pass
            else:
                raise NotImplementedError(f"TrainerAction.{which} has no counterpart in the reach engine")

    # ------------------------------------------------------------------ cycles
    def cycle(self, commands):
        """one launch; commands = {env index: (cmd, power, dir)}, every other env is frozen"""
        self._cmd[:] = (_capi.CMD_FREEZE, 0.0, 0.0, 0.0)
        for i, c in commands.items():
            self._cmd[i, :3] = c
        self.engine.step_commands(torch.from_numpy(self._cmd))
        return self.snapshot()

    def env_reset(self, i):
        """Soccer2DEnv.env_reset (soccer_2d_env.py:179-206) for instance i: trainer actions, then ONE command-less cycle."""
        env = self.envs[i]
        self.apply_trainer_actions(i, env.trainer_reset_actions())
        st = self.cycle({i: (_capi.CMD_NONE, 0.0, 0.0)})[i]
        env._latest_player_state = env._latest_trainer_state = st
        return env.state_to_observation(st), st

    def step(self, actions):
        """actions = {env index: action}: Soccer2DEnv.step (soccer_2d_env.py:226-269) for those instances, one launch."""
        cmds = {i: command_of(self.envs[i].action_to_rpc_actions(a, self.envs[i]._latest_player_state)) for i, a in actions.items()}
        states = self.cycle(cmds)
        out = {}
        for i in actions:
            env, st = self.envs[i], states[i]
            env._latest_player_state = env._latest_trainer_state = st
            obs = env.state_to_observation(st)
            done, reward, info = env.check_trainer_observation(st)
            out[i] = (obs, reward, done, info)
        return out


class HookVecEnv:
    """N instances of a hook-based env class behind a list-style vector surface (``reset() -> [obs]``,
    ``step([a]) -> ([obs], [reward], [done], [info])``, auto-reset with ``info['terminal_observation']``)."""

    def __init__(self, env_cls, num_envs, device='cuda:0', seed=0x5EED, noise=True, server_params=None, **kwargs):
        if not 1 <= int(num_envs) <= MAX_HOOK_ENVS:
            raise ValueError(f"the hook path runs 1..{MAX_HOOK_ENVS} envs, got {num_envs}")
        import soccer_2d_env
        if not (isinstance(env_cls, type) and issubclass(env_cls, soccer_2d_env.Soccer2DEnv) and env_cls.overrides_task_hooks()):
            raise TypeError('HookVecEnv wants a Soccer2DEnv subclass that overrides the four task hooks')
        soccer_2d_env._defer_hook_runtime[0] = True        # the instances share ONE engine, created below
        try:
            self.envs = [env_cls(**kwargs) for _ in range(int(num_envs))]
        finally:
            soccer_2d_env._defer_hook_runtime[0] = False
        self.runtime = HookRuntime(self.envs, device, seed, noise, server_params)
        for i, e in enumerate(self.envs):
            e._hooks, e._hook_index = self.runtime, i
        self.num_envs = len(self.envs)
        self.action_space, self.observation_space = self.envs[0].action_space, self.envs[0].observation_space

    def reset(self):
        return [e.reset() for e in self.envs]

    def step(self, actions):
        res = self.runtime.step({i: a for i, a in enumerate(actions)})
        obs, rew, done, info = [], [], [], []
        for i in range(self.num_envs):
            o, r, d, inf = res[i]
            if d:
                inf = dict(inf); inf['terminal_observation'] = o
                o = self.envs[i].reset()
            obs.append(o); rew.append(r); done.append(d); info.append(inf)
        return obs, rew, done, info

    def close(self):
        self.runtime.close()
