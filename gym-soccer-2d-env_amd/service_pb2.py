"""``service_pb2`` for task envs written against the reference: the message classes its task hooks build and read
(idl/service.proto), as light Python objects -- no protobuf runtime, no gRPC.

The reference's hooks speak protobuf objects: ``action_to_rpc_actions`` returns
``pb2.PlayerAction(dash=pb2.Dash(power=100, relative_direction=d))`` (reach_ball_env.py:75-85), ``trainer_reset_actions`` a list of
``pb2.TrainerAction(do_move_ball=pb2.DoMoveBall(position=pb2.RpcVector2D(x=.., y=..), ...))`` (reach_ball_env.py:187-197), and the
state arguments are read by attribute (``state.world_model.ball.position.x``).  The engine has no transport, so the messages never
reach a wire here: they are plain records with the reference's field names and keyword constructors, ``HasField`` /
``WhichOneof`` for the oneof members, and proto3 defaults (0 / False / empty message) for fields that were not set.  A hook-based
env imports this module as ``import service_pb2 as pb2`` exactly as it does with the reference.  (Bytes on a wire, where a caller
wants them, come from ``soccer2d_amd.wire``, which is checked against the reference's generated code.)

Only the messages on the task-hook path are declared: low-level body commands (idl/service.proto:380-419), the PlayerAction /
TrainerAction oneofs (:1291-1300, :1393-1433), RpcVector2D (:22-27), the enums GameModeType / Side (:88-92, :267-301).  The 58
high-level helios behaviours of the PlayerAction oneof (:684-1289) are accepted by name -- a hook may build them -- but only
``body_hold_ball`` has a meaning without the C++ proxy (no body command, soccer_2d_env.py:190).
"""


class _Message:
    """keyword-constructed record; FIELDS = {name: default or message class}"""
    FIELDS = {}
    ONEOF = None            # (group name, tuple of member names)
    OPEN = False            # accept unknown field names (the long tail of a oneof)

    def __init__(self, **kw):
        object.__setattr__(self, '_set', {})
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if k not in self.FIELDS and not self.OPEN:
            raise AttributeError(f"{type(self).__name__} has no field {k!r}")
        if self.ONEOF and (k in self.ONEOF[1] or (self.OPEN and k not in self.FIELDS)):
            for other in list(self._set):                  # setting one member of a oneof clears the others
                if other != k and (other in self.ONEOF[1] or other not in self.FIELDS):
                    del self._set[other]
        self._set[k] = v

    def __getattr__(self, k):
        s = object.__getattribute__(self, '_set')
        if k in s:
            return s[k]
        f = type(self).FIELDS
        if k in f:
            d = f[k]
            return d() if isinstance(d, type) else d
        raise AttributeError(k)

    def HasField(self, k):
        return k in self._set

    def WhichOneof(self, group):
        if not self.ONEOF or group != self.ONEOF[0]:
            raise ValueError(f"{type(self).__name__} has no oneof {group!r}")
        for k in self._set:
            if k in self.ONEOF[1] or k not in self.FIELDS:
                return k
        return None

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in self._set.items())})"

    def __eq__(self, o):
        return type(o) is type(self) and o._set == self._set


class RpcVector2D(_Message):                    # idl/service.proto:22-27
    FIELDS = {'x': 0.0, 'y': 0.0, 'dist': 0.0, 'angle': 0.0}


class Dash(_Message):                           # :380-383
    FIELDS = {'power': 0.0, 'relative_direction': 0.0}


class Turn(_Message):                           # :390-392
    FIELDS = {'relative_direction': 0.0}


class Kick(_Message):                           # :394-397
    FIELDS = {'power': 0.0, 'relative_direction': 0.0}


class Tackle(_Message):                         # :399-402
    FIELDS = {'power_or_dir': 0.0, 'foul': False}


class Catch(_Message):                          # :404
    FIELDS = {}


class Move(_Message):                           # :408-411
    FIELDS = {'x': 0.0, 'y': 0.0}


class TurnNeck(_Message):                       # :413-415
    FIELDS = {'moment': 0.0}


class ChangeView(_Message):                     # :417-419
    FIELDS = {'view_width': 0}


class Body_HoldBall(_Message):                  # :748-752 (the reference's no-op filler, soccer_2d_env.py:97, 190)
    FIELDS = {'do_turn': False, 'turn_target_point': RpcVector2D, 'kick_target_point': RpcVector2D}


class PlayerAction(_Message):                   # :1291-1361
    FIELDS = {'dash': Dash, 'turn': Turn, 'kick': Kick, 'tackle': Tackle, 'catch': Catch, 'move': Move,
              'turn_neck': TurnNeck, 'change_view': ChangeView, 'body_hold_ball': Body_HoldBall}
    ONEOF = ('action', tuple(FIELDS))
    OPEN = True


class PlayerActions(_Message):                  # :1363-1366
    FIELDS = {'actions': list, 'ignore_preprocess': False, 'ignore_doforcekick': False, 'ignore_doHeardPassRecieve': False,
              'ignore_doIntention': False}


class DoKickOff(_Message):                      # :1393
    FIELDS = {}


class DoMoveBall(_Message):                     # :1395-1398
    FIELDS = {'position': RpcVector2D, 'velocity': RpcVector2D}


class DoMovePlayer(_Message):                   # :1400-1405
    FIELDS = {'our_side': False, 'uniform_number': 0, 'position': RpcVector2D, 'body_direction': 0.0}


class DoRecover(_Message):                      # :1407
    FIELDS = {}


class DoChangeMode(_Message):                   # :1409-1412
    FIELDS = {'game_mode_type': 0, 'side': 0}


class DoChangePlayerType(_Message):             # :1414-1418
    FIELDS = {'our_side': False, 'uniform_number': 0, 'type': 0}


class TrainerAction(_Message):                  # :1423-1433
    FIELDS = {'do_kick_off': DoKickOff, 'do_move_ball': DoMoveBall, 'do_move_player': DoMovePlayer, 'do_recover': DoRecover,
              'do_change_mode': DoChangeMode, 'do_change_player_type': DoChangePlayerType}
    ONEOF = ('action', tuple(FIELDS))


class TrainerActions(_Message):
    FIELDS = {'actions': list}


class Side:                                     # :88-92
    UNKNOWN, LEFT, RIGHT = 0, 1, 2


class GameModeType:                             # :267-301
    BeforeKickOff, TimeOver, PlayOn, KickOff_, KickIn_, FreeKick_, CornerKick_, GoalKick_, AfterGoal_, OffSide_ = range(10)
    PenaltyKick_, FirstHalfOver, Pause, Human, FoulCharge_, FoulPush_, FoulMultipleAttacker_, FoulBallOut_ = range(10, 18)
    BackPass_, FreeKickFault_, CatchFault_, IndFreeKick_ = 18, 19, 20, 21
    PenaltySetup_, PenaltyReady_, PenaltyTaken_, PenaltyMiss_, PenaltyScore_ = 22, 23, 24, 25, 26
    IllegalDefense_, PenaltyOnfield_, PenaltyFoul_, GoalieCatch_, ExtendHalf, MODE_MAX = 27, 28, 29, 30, 31, 32


class State:
    """type name only: hooks annotate their arguments with ``pb2.State``; what they receive is a
    ``soccer2d_amd.state_view.StateView`` with the same attribute paths."""
