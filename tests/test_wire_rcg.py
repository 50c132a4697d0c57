"""protobuf wire format and .rcg game-log writer.

tests/golden/wire.json holds bytes produced by the REFERENCE's generated protobuf classes
(service_pb2, via tests/golden/make_golden.py): soccer2d_amd/wire.py must emit exactly those
bytes for the same field values and decode the reference's action messages."""
import json
import os

import numpy as np
import pytest

from soccer2d_amd import rcg, wire

G = os.path.join(os.path.dirname(__file__), 'golden', 'wire.json')


def test_state_bytes_equal_reference_pb2():
    g = json.load(open(G))
    assert len(g['states']) >= 6
    for row in g['states']:
        assert wire.encode_state(row['fields']).hex() == row['hex']


def test_decode_reference_player_actions():
    g = json.load(open(G))
    for row in g['player_actions']:
        buf = bytes.fromhex(row['hex'])
        cmd, a, b = wire.decode_player_action(buf)
        assert (cmd, a, b) == (row['cmd'], row['a'], row['b'])
        assert wire.encode_player_action(row['cmd'], row['a'], row['b']).hex() == row['hex']
        assert wire.decode_player_actions(bytes.fromhex(row['list_hex'])) == [(row['cmd'], row['a'], row['b'])]


def test_decode_reference_trainer_reset_actions():
    g = json.load(open(G))
    kinds = [r['kind'] for r in g['trainer_actions']]
    assert kinds == ['do_move_ball', 'do_move_player', 'do_recover', 'do_change_mode']     # reach_ball_env.py:187-197 + soccer_2d_env.py:242
    mb, mp_, rec, cm = (wire.decode_trainer_action(bytes.fromhex(r['hex'])) for r in g['trainer_actions'])
    f = g['trainer_actions'][0]['fields']
    assert list(mb['do_move_ball']['position']) == f['ball_pos'] and list(mb['do_move_ball']['velocity']) == f['ball_vel']
    f = g['trainer_actions'][1]['fields']
    assert list(mp_['do_move_player']['position']) == f['player_pos'] and mp_['do_move_player']['body_direction'] == f['body']
    assert mp_['do_move_player']['uniform_number'] == 1 and mp_['do_move_player']['our_side'] is True
    assert rec == {'do_recover': {}} and cm == {'do_change_mode': {'game_mode_type': 2, 'side': 1}}


def test_proto3_zero_fields_are_omitted_and_negative_ints_roundtrip():
    assert wire.encode_world_model(dict(cycle=0, game_mode_type=0)) == b''
    b = wire.encode_world_model(dict(cycle=-5))
    assert wire.decode(b) == [(21, 0, -5)]
    s = wire.encode_state(dict(self=dict(x=0.0, y=0.0)))
    # nested messages are present even when all their scalars are default
    assert wire.decode(wire.decode(s)[0][2])[0][0] == 6


def test_penalty_kick_state_on_the_wire():
    """WorldModel.is_penalty_kick_mode (field 30) and .penalty_kick_state (38; idl/service.proto:130-138, 336, 344): proto3 wire
    bytes, fields in number order, zero members omitted (no reference fixture holds these fields: checked with the decoder)."""
    b = wire.encode_world_model(dict(cycle=8000, game_mode_type=24, game_mode_side=1, is_penalty_kick_mode=True,
                                     penalty_kick_state=dict(on_field_side=2, current_taker_side=1, our_taker_counter=3, their_taker_counter=2,
                                                             our_score=2, their_score=0, is_kick_taker=True)))
    f = wire.decode(b)
    assert [x[0] for x in f] == [21, 22, 30, 38, 42] and dict((x[0], x[2]) for x in f)[30] == 1
    pk = dict((x[0], x[2]) for x in wire.decode(dict((x[0], x[2]) for x in f)[38]))
    assert pk == {1: 2, 2: 1, 3: 3, 4: 2, 5: 2, 7: 1}                    # their_score = 0 is not on the wire
    assert 30 not in [x[0] for x in wire.decode(wire.encode_world_model(dict(cycle=5, game_mode_type=2)))]


def test_rcg_round_trip(tmp_path):
    path = tmp_path / 'm.rcg'
    rs = np.random.RandomState(0)
    frames = []
    with rcg.RcgWriter(path) as w:
        w.header({'ball_decay': 0.94, 'effort_dec': 0.005}, [{'player_decay': 0.4}, {'player_decay': 0.4321987, 'kick_rand': 1e-05}])
        for c in range(1, 6):
            mode, side = (3, 1) if c < 3 else (2, 0)
            w.playmode(c, mode, side)
            w.team(c, 0 if c < 4 else 1, 0)
            ball = tuple(round(float(v), 4) for v in rs.uniform(-30, 30, 4))
            pl = [dict(side='l' if i < 11 else 'r', unum=i % 11 + 1, x=round(float(rs.uniform(-50, 50)), 4), y=round(float(rs.uniform(-30, 30)), 4),
                       vx=0.25, vy=-0.5, body=round(float(rs.uniform(-180, 180)), 4), stamina=7945.0, effort=1.0, recovery=1.0,
                       capacity=130555.0, tackling=(i == 4), type=i % 7, goalie=(i % 11 == 0)) for i in range(22)]
            w.show(c, ball, pl)
            frames.append((c, ball, pl))
    recs = rcg.read_rcg(path)
    assert [r for r in recs if r[0] == 'playmode'] == [('playmode', 1, 'kick_off_l'), ('playmode', 3, 'play_on')]
    assert [r for r in recs if r[0] == 'team'] == [('team', 1, 's2d_left', 's2d_right', 0, 0), ('team', 4, 's2d_left', 's2d_right', 1, 0)]
    shows = [r for r in recs if r[0] == 'show']
    assert len(shows) == 5
    for (c, ball, pl), (_k, c2, ball2, pl2) in zip(frames, shows):
        assert c == c2 and ball == pytest.approx(ball2) and len(pl2) == 22
        for a, b in zip(pl, pl2):
            assert (a['side'], a['unum']) == (b['side'], b['unum']) and a['x'] == pytest.approx(b['x']) and a['body'] == pytest.approx(b['body'])
            assert bool(b['state'] & 0x1000) == a['tackling'] and bool(b['state'] & 0x8) == a['goalie'] and b['type'] == a['type']
    assert recs[0] == ('server_param', {'ball_decay': 0.94, 'effort_dec': 0.005})
    assert recs[1] == ('player_type', {'id': 0.0, 'player_decay': 0.4})
    assert recs[2] == ('player_type', {'id': 1.0, 'player_decay': 0.4321987, 'kick_rand': 1e-05})
    assert open(path).readline().strip() == 'ULG5'


@pytest.mark.gpu
def test_engine_states_as_protobuf_and_rcg(tmp_path):
    torch = pytest.importorskip('torch')
    from soccer2d_amd.match import MatchEngine
    from soccer2d_amd.vec_env import Soccer2DVecEnv
    env = Soccer2DVecEnv(64, use_continuous_action=False, change_ball_velocity=True)
    env.reset(); env.rollout(7, with_obs=False)
    buf = wire.reach_ball_state_bytes(env, 5)
    wm = dict((f, v) for f, _w, v in wire.decode(wire.decode(buf)[0][2]))
    assert wm[21] == int(env.engine.cycle[5]) and wm[22] == 2
    me = dict((f, v) for f, _w, v in wire.decode(wm[6]))
    assert wire._vec_dict(me[1]) == (env.engine.player_x[5].item(), env.engine.player_y[5].item())
    assert me[29] == env.engine.stamina[5].item() and me[17] == env.engine.player_body[5].item()
    rcg.record_reach_ball(env, 5, 12, tmp_path / 'reach.rcg')
    assert sum(r[0] == 'show' for r in rcg.read_rcg(tmp_path / 'reach.rcg')) == 12
    eng = MatchEngine(16, half_time_cycles=40)
    rcg.record_match(eng, 3, 60, tmp_path / 'match.rcg')
    recs = rcg.read_rcg(tmp_path / 'match.rcg')
    shows = [r for r in recs if r[0] == 'show']
    assert len(shows) == 60 and all(len(r[3]) == 22 for r in shows)
    assert recs[0][0] == 'server_param' and recs[0][1]['ball_decay'] == pytest.approx(eng.cfg.sp.ball_decay)
    assert sum(r[0] == 'player_type' for r in recs) == 18
    assert [p['type'] for p in shows[0][3]] == list(eng.cfg.player_type_id)[:22] and shows[0][3][11]['state'] & 0x8
    assert ('playmode', 1, 'kick_off_l') in recs or any(r[0] == 'playmode' for r in recs)
    sb = wire.match_state_bytes(eng, 3, 15)
    wmm = wire.decode(wire.decode(sb)[0][2])
    assert sum(1 for f, _w, _v in wmm if f == 8) == 10 and sum(1 for f, _w, _v in wmm if f == 9) == 11
    assert dict((f, v) for f, w, v in wmm if w == 0)[4] == 2      # our_side = RIGHT for player 15
