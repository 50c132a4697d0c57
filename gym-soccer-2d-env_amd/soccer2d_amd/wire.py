"""Protobuf wire format of the reference's schema (idl/service.proto), without generated code
(SURVEY.md 8f rank 3).  Lets one env of the batched engine be exported as real ``State``
messages for agents written against the reference's gRPC service (server.py:49-103), and lets
``PlayerAction`` / ``TrainerAction`` bytes from such agents be turned into engine commands.

Only the fields the path touches are modelled; field numbers are cited from idl/service.proto.
proto3 rules kept: scalars equal to zero are not written, fields are written in field-number
order (what the reference's generated code emits), floats are 32-bit little-endian (wire type
5), int32/enum/bool are varints (negative int32 as 10-byte two's complement).
"""
import struct

# ---- field numbers (idl/service.proto) ---------------------------------------------------
VEC = {'x': 1, 'y': 2, 'dist': 3, 'angle': 4}                                   # RpcVector2D :22-27
BALL = {'position': 1, 'relative_position': 2, 'velocity': 5, 'dist_from_self': 16, 'angle_from_self': 17}   # :68-86
PLAYER = {'position': 1, 'velocity': 4, 'side': 15, 'uniform_number': 16, 'body_direction': 19,
          'is_tackling': 29, 'type_id': 30}                                     # :144-175
SELF = {'position': 1, 'velocity': 4, 'side': 13, 'uniform_number': 14, 'body_direction': 17,
        'dist_from_ball': 24, 'angle_from_ball': 25, 'stamina': 29, 'type_id': 35, 'recovery': 37,
        'stamina_capacity': 38, 'effort': 41}                                   # :181-223
WM = {'our_side': 4, 'self': 6, 'ball': 7, 'teammates': 8, 'opponents': 9, 'cycle': 21, 'game_mode_type': 22,
      'left_team_score': 23, 'right_team_score': 24, 'stoped_cycle': 27, 'our_team_score': 28,
      'their_team_score': 29, 'game_mode_side': 42}                             # :306-349
STATE = {'world_model': 2}                                                      # :354-359
PLAYER_ACTION = {'dash': 1, 'turn': 2, 'kick': 3, 'tackle': 4}                  # :1291-1300
TRAINER_ACTION = {'do_kick_off': 1, 'do_move_ball': 2, 'do_move_player': 3, 'do_recover': 4, 'do_change_mode': 5}  # :1423-1433


# ---- primitives ---------------------------------------------------------------------------
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _tag(field, wt):
    return _varint((field << 3) | wt)


def _f32(field, v):
    v = float(v)
    if v == 0.0 and struct.pack('<f', v) == b'\x00\x00\x00\x00':
        return b''
    return _tag(field, 5) + struct.pack('<f', v)


def _int(field, v):
    v = int(v)
    return b'' if v == 0 else _tag(field, 0) + _varint(v)


def _msg(field, payload, present=True):
    if not present:
        return b''
    return _tag(field, 2) + _varint(len(payload)) + payload


def _vec(field, x, y, dist=0.0, angle=0.0):
    return _msg(field, _f32(1, x) + _f32(2, y) + _f32(3, dist) + _f32(4, angle))


# ---- encoders ------------------------------------------------------------------------------
def encode_player(p):
    """p: dict with x,y,vx,vy,side,uniform_number,body_direction[,is_tackling,type_id]."""
    return (_vec(1, p['x'], p['y']) + _vec(4, p.get('vx', 0.0), p.get('vy', 0.0)) + _int(15, p.get('side', 0))
            + _int(16, p.get('uniform_number', 0)) + _int(18, 1 if p.get('is_goalie') else 0)
            + _f32(19, p.get('body_direction', 0.0))
            + _int(29, 1 if p.get('is_tackling') else 0) + _int(30, p.get('type_id', 0)))


def encode_self(s):
    return (_vec(1, s['x'], s['y']) + _vec(4, s.get('vx', 0.0), s.get('vy', 0.0)) + _int(13, s.get('side', 0))
            + _int(14, s.get('uniform_number', 0)) + _f32(17, s.get('body_direction', 0.0))
            + _f32(24, s.get('dist_from_ball', 0.0)) + _f32(25, s.get('angle_from_ball', 0.0))
            + _f32(29, s.get('stamina', 0.0)) + _int(35, s.get('type_id', 0)) + _f32(37, s.get('recovery', 0.0))
            + _f32(38, s.get('stamina_capacity', 0.0)) + _f32(41, s.get('effort', 0.0)))


def encode_ball(b):
    return (_vec(1, b['x'], b['y']) + _vec(2, b.get('rel_x', 0.0), b.get('rel_y', 0.0))
            + _vec(5, b.get('vx', 0.0), b.get('vy', 0.0)) + _f32(16, b.get('dist_from_self', 0.0))
            + _f32(17, b.get('angle_from_self', 0.0)))


def encode_world_model(wm):
    """wm: dict(our_side, self, ball, teammates[list], opponents[list], cycle, game_mode_type,
    left_team_score, right_team_score, stoped_cycle, our_team_score, their_team_score, game_mode_side, is_penalty_kick_mode,
    penalty_kick_state = dict(on_field_side, current_taker_side, our_taker_counter, their_taker_counter, our_score, their_score,
    is_kick_taker))."""
    out = _int(4, wm.get('our_side', 0))
    if wm.get('self') is not None:
        out += _msg(6, encode_self(wm['self']))
    if wm.get('ball') is not None:
        out += _msg(7, encode_ball(wm['ball']))
    for p in wm.get('teammates', ()):
        out += _msg(8, encode_player(p))
    for p in wm.get('opponents', ()):
        out += _msg(9, encode_player(p))
    out += (_int(21, wm.get('cycle', 0)) + _int(22, wm.get('game_mode_type', 0)) + _int(23, wm.get('left_team_score', 0))
            + _int(24, wm.get('right_team_score', 0)) + _int(27, wm.get('stoped_cycle', 0))
            + _int(28, wm.get('our_team_score', 0)) + _int(29, wm.get('their_team_score', 0))
            + _int(30, 1 if wm.get('is_penalty_kick_mode') else 0))
    pk = wm.get('penalty_kick_state')
    if pk is not None:                                     # PenaltyKickState (idl/service.proto:130-138; WorldModel field 38)
        out += _msg(38, _int(1, pk.get('on_field_side', 0)) + _int(2, pk.get('current_taker_side', 0))
                    + _int(3, pk.get('our_taker_counter', 0)) + _int(4, pk.get('their_taker_counter', 0))
                    + _int(5, pk.get('our_score', 0)) + _int(6, pk.get('their_score', 0)) + _int(7, 1 if pk.get('is_kick_taker') else 0))
    out += _int(42, wm.get('game_mode_side', 0))
    return out


def encode_state(wm):
    """Bytes of a `State` message (idl/service.proto:354-359) carrying `world_model`."""
    return _msg(2, encode_world_model(wm))


def encode_player_action(cmd, a=0.0, b=0.0):
    """PlayerAction bytes for the engine's command ISA: 'dash'(power,dir) 'turn'(dir) 'kick'(power,dir)
    'tackle'(power_or_dir) 'catch'() 'move'(x,y)."""
    if cmd == 'dash':
        return _msg(1, _f32(1, a) + _f32(2, b))
    if cmd == 'turn':
        return _msg(2, _f32(1, a))
    if cmd == 'kick':
        return _msg(3, _f32(1, a) + _f32(2, b))
    if cmd == 'tackle':
        return _msg(4, _f32(1, a))
    if cmd == 'catch':                     # Catch{} is an empty message: the oneof member must still be present
        return _tag(5, 2) + _varint(0)
    if cmd == 'move':
        return _msg(6, _f32(1, a) + _f32(2, b))
    raise ValueError(cmd)


# ---- generic decoder ------------------------------------------------------------------------
def decode(buf):
    """Wire-level decode: list of (field, wire_type, value); value = int | float(bits as f32) | bytes."""
    out, i, n = [], 0, len(buf)
    while i < n:
        key, i = _read_varint(buf, i)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
            if v >= 1 << 63:
                v -= 1 << 64
        elif wt == 5:
            v = struct.unpack_from('<f', buf, i)[0]
            i += 4
        elif wt == 1:
            v = struct.unpack_from('<d', buf, i)[0]
            i += 8
        elif wt == 2:
            ln, i = _read_varint(buf, i)
            v = bytes(buf[i:i + ln])
            i += ln
        else:
            raise ValueError(f'unsupported wire type {wt}')
        out.append((field, wt, v))
    return out


def _read_varint(buf, i):
    shift = v = 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, i
        shift += 7


def _vec_dict(payload):
    d = {1: 0.0, 2: 0.0}
    for f, _wt, v in decode(payload):
        d[f] = v
    return d[1], d[2]


def decode_player_action(buf):
    """PlayerAction bytes -> (cmd, a, b) for the commands the engine executes, else (None, 0, 0).
    Reference agents also send high-level helios behaviours (idl/service.proto:684-1289): those are
    AI, not simulation, and are reported as None."""
    for field, wt, v in decode(buf):
        if wt != 2:
            continue
        vals = {f: x for f, _w, x in decode(v)}
        if field == 1:
            return 'dash', float(vals.get(1, 0.0)), float(vals.get(2, 0.0))
        if field == 2:
            return 'turn', float(vals.get(1, 0.0)), 0.0
        if field == 3:
            return 'kick', float(vals.get(1, 0.0)), float(vals.get(2, 0.0))
        if field == 4:
            return 'tackle', float(vals.get(1, 0.0)), 0.0
        if field == 5:                     # Catch{}: the engine's catch direction is 0 (straight ahead)
            return 'catch', 0.0, 0.0
        if field == 6:                     # Move{x, y}: team-frame coordinates (S2D_MCMD_MOVE)
            return 'move', float(vals.get(1, 0.0)), float(vals.get(2, 0.0))
    return None, 0.0, 0.0


def decode_player_actions(buf):
    """PlayerActions (repeated PlayerAction actions = 1, idl/service.proto:1363-1370) -> list of commands."""
    return [decode_player_action(v) for f, wt, v in decode(buf) if f == 1 and wt == 2]


def decode_trainer_action(buf):
    """TrainerAction bytes -> dict (the reset ISA of reach_ball_env.py:187-195 and soccer_2d_env.py:242)."""
    for field, wt, v in decode(buf):
        if wt != 2:
            continue
        if field == 2:
            d = {f: x for f, _w, x in decode(v)}
            return {'do_move_ball': {'position': _vec_dict(d.get(1, b'')), 'velocity': _vec_dict(d.get(2, b''))}}
        if field == 3:
            d = {f: x for f, _w, x in decode(v)}
            return {'do_move_player': {'our_side': bool(d.get(1, 0)), 'uniform_number': int(d.get(2, 0)),
                                       'position': _vec_dict(d.get(3, b'')), 'body_direction': float(d.get(4, 0.0))}}
        if field == 4:
            return {'do_recover': {}}
        if field == 5:
            d = {f: x for f, _w, x in decode(v)}
            return {'do_change_mode': {'game_mode_type': int(d.get(1, 0)), 'side': int(d.get(2, 0))}}
        if field == 1:
            return {'do_kick_off': {}}
    return {}


# ---- engine views -> messages ----------------------------------------------------------------
def reach_ball_state_bytes(vec_env, index=0):
    """`State` of the reach_ball player (what GetPlayerActions receives, server.py:49-53) for env `index`."""
    wm = vec_env.world_model()
    g = lambda k: wm['world_model.' + k][index].item()  # noqa: E731
    me = dict(x=g('self.position.x'), y=g('self.position.y'), vx=g('self.velocity.x'), vy=g('self.velocity.y'),
              side=1, uniform_number=1, body_direction=g('self.body_direction'), stamina=g('self.stamina'),
              recovery=g('self.recovery'), stamina_capacity=g('self.stamina_capacity'), effort=g('self.effort'),
              dist_from_ball=g('self.dist_from_ball'), angle_from_ball=g('self.angle_from_ball'))
    ball = dict(x=g('ball.position.x'), y=g('ball.position.y'), vx=g('ball.velocity.x'), vy=g('ball.velocity.y'),
                rel_x=g('ball.relative_position.x'), rel_y=g('ball.relative_position.y'),
                dist_from_self=g('ball.dist_from_self'), angle_from_self=g('ball.angle_from_self'))
    mate = dict(x=me['x'], y=me['y'], vx=me['vx'], vy=me['vy'], side=1, uniform_number=1, body_direction=me['body_direction'])
    return encode_state(dict(our_side=1, self=me, ball=ball, teammates=[mate], cycle=int(g('cycle')),
                             game_mode_type=int(g('game_mode_type')), stoped_cycle=0))


def match_state_bytes(engine, index, player):
    """`State` seen by `player` (0..21) of match `index`: full-state world model, own team as teammates."""
    x, y, vx, vy, body = (t[index].tolist() for t in (engine.x, engine.y, engine.vx, engine.vy, engine.body))
    st, ef, rc, cp, tk = (t[index].tolist() for t in (engine.stamina, engine.effort, engine.recovery,
                                                       engine.stamina_capacity, engine.tackle_cycles))
    left = player < 11
    mine, theirs = (range(0, 11), range(11, 22)) if left else (range(11, 22), range(0, 11))
    types = list(engine.cfg.player_type_id)

    def pl(i):
        return dict(x=x[i], y=y[i], vx=vx[i], vy=vy[i], side=1 if i < 11 else 2, uniform_number=i % 11 + 1,
                    body_direction=body[i], is_tackling=tk[i] > 0, is_goalie=i % 11 == 0, type_id=types[i])
    me = dict(pl(player), stamina=st[player], effort=ef[player], recovery=rc[player], stamina_capacity=cp[player])
    ball = dict(x=x[22], y=y[22], vx=vx[22], vy=vy[22], rel_x=x[22] - x[player], rel_y=y[22] - y[player])
    sl, sr = int(engine.score_left[index]), int(engine.score_right[index])
    mode, side, w = int(engine.mode[index]), int(engine.mode_side[index]), int(engine.set_play_taker[index])
    pen = 22 <= mode <= 29                                 # the shoot-out: its state is in the set-play word (include/s2d_match.h)
    pk = None
    if pen:
        kicks, goals = ((w >> 12) & 15, (w >> 16) & 15), ((w >> 20) & 15, (w >> 24) & 15)
        mine_i = 0 if left else 1
        pk = dict(on_field_side=2, current_taker_side=0 if mode == 28 else side, our_taker_counter=kicks[mine_i],
                  their_taker_counter=kicks[1 - mine_i], our_score=goals[mine_i], their_score=goals[1 - mine_i],
                  is_kick_taker=mode != 28 and (w & 0xff) - 1 == player)
    return encode_state(dict(our_side=1 if left else 2, self=me, ball=ball, is_penalty_kick_mode=pen, penalty_kick_state=pk,
                             stoped_cycle=int(engine.stopped_cycle[index]),
                             teammates=[pl(i) for i in mine if i != player], opponents=[pl(i) for i in theirs],
                             cycle=int(engine.cycle[index]), game_mode_type=int(engine.mode[index]),
                             game_mode_side=int(engine.mode_side[index]), left_team_score=sl, right_team_score=sr,
                             our_team_score=sl if left else sr, their_team_score=sr if left else sl))
