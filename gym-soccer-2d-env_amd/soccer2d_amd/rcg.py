"""rcssserver game-log (.rcg, text format version 5, header ``ULG5``) writer.

The reference asks rcssserver to write .rcg/.rcl logs into ``log_dir`` (soccer_2d_env.py:367-368)
so that episodes can be replayed in rcssmonitor / soccerwindow2; this module produces the same
kind of file from engine state: ``(show T ((b) x y vx vy) ((l u) type state x y vx vy body neck
(v h 90) (s stamina effort recovery capacity) (c 0 ...)) ...)``, ``(playmode T name)`` and
``(team T l r score_l score_r)`` lines.  Format restated from rcssserver's logger (EXT): viewers
are not available offline, so the writer is checked by its own reader (round trip) only.
"""
PLAYMODE_NAMES = {1: 'time_over', 2: 'play_on', 3: 'kick_off_{s}', 4: 'kick_in_{s}', 5: 'free_kick_{s}',
                  6: 'corner_kick_{s}', 7: 'goal_kick_{s}', 8: 'goal_{s}', 9: 'offside_{s}', 0: 'before_kick_off',
                  10: 'penalty_kick_{s}', 11: 'first_half_over', 12: 'pause', 13: 'human_judge', 14: 'foul_charge_{s}', 15: 'foul_push_{s}', 16: 'foul_multiple_attack_{s}', 17: 'foul_ballout_{s}', 18: 'back_pass_{s}',
                  19: 'free_kick_fault_{s}', 20: 'catch_fault_{s}', 21: 'indirect_free_kick_{s}', 27: 'illegal_defense_{s}', 30: 'goalie_catch_ball_{s}',
                  31: 'time_extended'}


def playmode_name(mode, side):
    return PLAYMODE_NAMES.get(int(mode), 'play_on').format(s='l' if int(side) == 1 else 'r')


def _f(v):
    return f'{float(v):.4f}'.rstrip('0').rstrip('.') if v else '0'


class RcgWriter:
    def __init__(self, path, team_left='s2d_left', team_right='s2d_right'):
        self.f = open(path, 'w')
        self.team_left, self.team_right = team_left, team_right
        self.f.write('ULG5\n')
        self._mode = None
        self._score = None

    def header(self, server_params=None, player_types=None):
        """The parameter records rcssserver puts in front of a version-5 log: ``(server_param (name value)...)``
        and one ``(player_type (id N)(name value)...)`` per heterogeneous type (idl/service.proto:1435-1732 names).
        server_params: dict name -> value; player_types: list of dicts (index = type id)."""
        if server_params:
            self.f.write('(server_param ' + ''.join(f'({k} {float(v):.8g})' for k, v in server_params.items()) + ')\n')
        for tid, t in enumerate(player_types or []):
            self.f.write(f'(player_type (id {tid})' + ''.join(f'({k} {float(v):.8g})' for k, v in t.items()) + ')\n')

    def playmode(self, cycle, mode, side=0):
        name = playmode_name(mode, side)
        if name != self._mode:
            self.f.write(f'(playmode {int(cycle)} {name})\n')
            self._mode = name

    def team(self, cycle, score_left, score_right):
        if (score_left, score_right) != self._score:
            self.f.write(f'(team {int(cycle)} {self.team_left} {self.team_right} {int(score_left)} {int(score_right)})\n')
            self._score = (score_left, score_right)

    def show(self, cycle, ball, players):
        """ball = (x,y,vx,vy); players = iterable of dicts(side 'l'|'r', unum, x,y,vx,vy,body,stamina,effort,
        recovery,capacity[,tackling,type,goalie])."""
        parts = [f'(show {int(cycle)} ((b) {_f(ball[0])} {_f(ball[1])} {_f(ball[2])} {_f(ball[3])})']
        for p in players:
            state = 0x1 | (0x8 if p.get('goalie') else 0) | (0x1000 if p.get('tackling') else 0)   # STAND [| GOALIE] [| TACKLE]
            parts.append(f"(({p['side']} {int(p['unum'])}) {int(p.get('type', 0))} {hex(state)} {_f(p['x'])} {_f(p['y'])} {_f(p['vx'])} {_f(p['vy'])} "
                         f"{_f(p['body'])} 0 (v h 90) (s {_f(p['stamina'])} {_f(p['effort'])} {_f(p['recovery'])} "
                         f"{_f(p['capacity'])}) (c 0 0 0 0 0 0 0 0 0 0 0))")
        self.f.write(' '.join(parts) + ')\n')

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_rcg(path):
    """Minimal reader for the writer's own output: list of ('show', cycle, ball, players) /
    ('playmode', cycle, name) / ('team', cycle, l, r, sl, sr)."""
    import re
    out = []
    with open(path) as f:
        assert f.readline().strip() == 'ULG5'
        for line in f:
            line = line.strip()
            if line.startswith('(playmode'):
                _, c, name = line.strip('()').split()
                out.append(('playmode', int(c), name))
            elif line.startswith('(server_param') or line.startswith('(player_type'):
                kind = line[1:line.index(' ')]
                out.append((kind, {k: float(v) for k, v in re.findall(r'\((\w+) ([-+\d.e]+)\)', line)}))
            elif line.startswith('(team'):
                _, c, l, r, sl, sr = line.strip('()').split()
                out.append(('team', int(c), l, r, int(sl), int(sr)))
            elif line.startswith('(show'):
                c = int(line.split()[1])
                b = re.search(r'\(\(b\) ([^)]*)\)', line).group(1).split()
                pl = []
                for m in re.finditer(r'\(\(([lr]) (\d+)\) (\d+) (0x[0-9a-f]+) ([-\d.e]+) ([-\d.e]+) ([-\d.e]+) ([-\d.e]+) ([-\d.e]+) '
                                     r'[-\d.e]+ \(v h 90\) \(s ([-\d.e]+) ([-\d.e]+) ([-\d.e]+) ([-\d.e]+)\)', line):
                    g = m.groups()
                    pl.append(dict(side=g[0], unum=int(g[1]), type=int(g[2]), state=int(g[3], 16), x=float(g[4]), y=float(g[5]),
                                   vx=float(g[6]), vy=float(g[7]), body=float(g[8]), stamina=float(g[9]), effort=float(g[10]),
                                   recovery=float(g[11]), capacity=float(g[12])))
                out.append(('show', c, tuple(float(v) for v in b), pl))
    return out


def record_match(engine, index, n_cycles, path, actions=None):
    """Step `engine` (MatchEngine) n_cycles times (random policy unless `actions(t)` returns a tensor) and log
    match `index` to `path`.  Host copies of ONE match per cycle: a debugging / viewing aid, not a hot path."""
    from . import _capi_match as M
    cfg = engine.cfg
    types = list(cfg.player_type_id)[:22]
    with RcgWriter(path) as w:
        w.header({n: getattr(cfg.sp, n) for n, _ in cfg.sp._fields_},
                 [{f: getattr(cfg.player_types[t], f) for f in M.PLAYER_TYPE_FIELDS} for t in range(M.MATCH_PLAYER_TYPES)])
        for t in range(n_cycles):
            engine.step(actions(t) if actions else None)
            x, y, vx, vy, body = (a[index].tolist() for a in (engine.x, engine.y, engine.vx, engine.vy, engine.body))
            st, ef, rc, cp, tk = (a[index].tolist() for a in (engine.stamina, engine.effort, engine.recovery,
                                                               engine.stamina_capacity, engine.tackle_cycles))
            cyc = int(engine.cycle[index])
            w.playmode(cyc, int(engine.mode[index]), int(engine.mode_side[index]))
            w.team(cyc, int(engine.score_left[index]), int(engine.score_right[index]))
            w.show(cyc, (x[22], y[22], vx[22], vy[22]),
                   [dict(side='l' if i < 11 else 'r', unum=i % 11 + 1, x=x[i], y=y[i], vx=vx[i], vy=vy[i], body=body[i],
                         stamina=st[i], effort=ef[i], recovery=rc[i], capacity=cp[i], tackling=tk[i] > 0, type=types[i],
                         goalie=i % 11 == 0) for i in range(22)])


def record_reach_ball(vec_env, index, n_cycles, path, actions=None):
    """Same for one reach_ball env (one left player + the ball)."""
    e = vec_env.engine
    with RcgWriter(path) as w:
        w.playmode(int(e.cycle[index]), 2)
        for t in range(n_cycles):
            vec_env.step(actions(t) if actions else None)
            w.show(int(e.cycle[index]), (e.ball_x[index].item(), e.ball_y[index].item(), e.ball_vx[index].item(), e.ball_vy[index].item()),
                   [dict(side='l', unum=1, x=e.player_x[index].item(), y=e.player_y[index].item(), vx=e.player_vx[index].item(),
                         vy=e.player_vy[index].item(), body=e.player_body[index].item(), stamina=e.stamina[index].item(),
                         effort=e.effort[index].item(), recovery=e.recovery[index].item(), capacity=e.stamina_capacity[index].item())])
