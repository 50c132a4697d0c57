"""Protobuf-mirroring views of the engine state.

``world_model_tensors(engine)`` returns a dict of DEVICE tensors keyed by the dotted proto
path of idl/service.proto (State.world_model....): plain state words are zero-copy views of
the arena, derived fields (RpcVector2D.dist/.angle, dist_from_self, ...) come from the
s2d_world_model kernel.  ``StateView`` gives attribute access for ONE env with the same
spelling the reference code uses on ``pb2.State`` (e.g.
``state.world_model.teammates[0].position.x``, reach_ball_env.py:89-94, 115-122) --
the wire format itself is out of scope.
"""
import torch

from . import _capi


def world_model_tensors(engine, derived=True):
    n, dev = engine.num_envs, engine.device
    e = engine
    wm = {
        'world_model.cycle': e.cycle,                                   # idl/service.proto:326
        'world_model.ball.position.x': e.ball_x, 'world_model.ball.position.y': e.ball_y,
        'world_model.ball.velocity.x': e.ball_vx, 'world_model.ball.velocity.y': e.ball_vy,
        'world_model.self.position.x': e.player_x, 'world_model.self.position.y': e.player_y,
        'world_model.self.velocity.x': e.player_vx, 'world_model.self.velocity.y': e.player_vy,
        'world_model.self.body_direction': e.player_body,
        'world_model.self.stamina': e.stamina, 'world_model.self.effort': e.effort,
        'world_model.self.recovery': e.recovery, 'world_model.self.stamina_capacity': e.stamina_capacity,
    }
    # constants of the scenario (one player, left side, uniform number 1, PlayOn forced every
    # cycle by soccer_2d_env.py:242, no stoppage, score 0-0)
    const_i32 = {
        'world_model.stoped_cycle': 0, 'world_model.game_mode_type': _capi.MODE_PLAY_ON,
        'world_model.left_team_score': 0, 'world_model.right_team_score': 0,
        'world_model.our_team_score': 0, 'world_model.their_team_score': 0,
        'world_model.our_side': _capi.SIDE_LEFT, 'world_model.self.side': _capi.SIDE_LEFT,
        'world_model.self.uniform_number': 1, 'world_model.self.type_id': 0,
    }
    for k, v in const_i32.items():
        wm[k] = torch.full((n,), v, dtype=torch.int32, device=dev)
    if derived:
        d = engine.world_model_derived()
        wm.update({
            'world_model.ball.dist_from_self': d['ball_dist_from_self'],
            'world_model.ball.angle_from_self': d['ball_angle_from_self'],
            'world_model.ball.relative_position.x': d['ball_relative_x'],
            'world_model.ball.relative_position.y': d['ball_relative_y'],
            'world_model.ball.position.dist': d['ball_pos_dist'], 'world_model.ball.position.angle': d['ball_pos_angle'],
            'world_model.ball.velocity.dist': d['ball_vel_dist'], 'world_model.ball.velocity.angle': d['ball_vel_angle'],
            'world_model.self.position.dist': d['self_pos_dist'], 'world_model.self.position.angle': d['self_pos_angle'],
            'world_model.self.velocity.dist': d['self_vel_dist'], 'world_model.self.velocity.angle': d['self_vel_angle'],
            'world_model.self.dist_from_ball': d['self_dist_from_ball'],
            'world_model.self.angle_from_ball': d['self_angle_from_ball'],
        })
    # the trainer sees the same player as teammates[0] (reach_ball_env.py:117-122): table [N,11,F]
    # layout of the 11v11 engine, one row filled
    for src, dst in (('self.position.x', 'teammates.position.x'), ('self.position.y', 'teammates.position.y'),
                     ('self.body_direction', 'teammates.body_direction')):
        wm['world_model.' + dst] = wm['world_model.' + src].unsqueeze(1)
    return wm


class _Node:
    def __init__(self):
        object.__setattr__(self, '_kids', {})

    def __getattr__(self, k):
        try:
            return self._kids[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __repr__(self):
        return f'<{", ".join(self._kids)}>'


class StateView(_Node):
    """Host-side snapshot of one env with pb2.State-like attribute paths."""

    def __init__(self, wm, index=0):
        super().__init__()
        for path, t in wm.items():
            parts = path.split('.')
            node = self
            for i, p in enumerate(parts[:-1]):
                if p == 'teammates':
                    lst = node._kids.setdefault(p, [_Node()])
                    node = lst[0]
                    continue
                node = node._kids.setdefault(p, _Node())
            v = t[index]
            if getattr(v, 'ndim', 0) > 0:                  # player tables [N, 11, ...]: the one filled row
                v = v[0]
            node._kids[parts[-1]] = v.item()
