"""Hand-built scenarios for the 11v11 match oracle (rcssserver rules restated, EXT /
parity-unpinned): kick, tackle, collisions, goals, restarts, offside,
half time, time over, determinism."""
import numpy as np
import pytest

import match_oracle as MO
from soccer2d_amd._capi_match import (GM_AFTER_GOAL, GM_CORNER_KICK, GM_GOAL_KICK, GM_KICK_IN, GM_KICK_OFF, GM_OFF_SIDE, GM_PLAY_ON,
                                      GM_TIME_OVER, MCMD_DASH, MCMD_KICK, MCMD_MOVE, MCMD_NONE, MCMD_TACKLE, MCMD_TURN)

LEFT, RIGHT = 1, 2


def fresh(n=1, **kw):
    return MO.MatchOracle(MO.make_match_config(**kw), n)


def acts(n=1, **per_player):
    a = np.zeros((n, 22, 3), dtype=np.float32)
    for k, v in per_player.items():
        a[:, int(k[1:])] = v
    return a


def play_on(m, e=0):
    m.set_game(e, mode=GM_PLAY_ON, mode_side=0)


def test_reset_state():
    m = fresh()
    assert m.get('mode')[0] == GM_KICK_OFF and m.get('mode_side')[0] == LEFT and m.get('cycle')[0] == 0
    x, y, body = m.get('x')[0], m.get('y')[0], m.get('body')[0]
    assert x[0] == -50 and x[11] == 50 and body[0] == 0 and body[11] == 180
    assert (x[:10] < 0).all() and (x[11:22] > 0).all() and x[10] == pytest.approx(-0.4) and x[22] == 0 and y[22] == 0
    assert (m.get('stamina')[0][:22] == 8000).all()
    # opponents of the taking side start outside the centre circle
    assert (np.hypot(x[11:22], y[11:22]) > 9.15).all()


def test_kickoff_kick_starts_play_and_ball_moves():
    m = fresh()
    m.step(acts(p10=[MCMD_KICK, 100, 0]))          # left #11 stands 0.4 m behind the ball, body 0
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('last_touch_side')[0] == LEFT
    # eff = 100 * .027 * (1 - .25*0 - .25*(0.4-0.385)/0.7) -> ball speed 2.6855, then decay .94
    eff = 100 * 0.027 * (1 - 0.25 * (0.4 - 0.385) / 0.7)
    assert m.get('x')[0][22] == pytest.approx(eff, rel=1e-5) and m.get('vx')[0][22] == pytest.approx(eff * 0.94, rel=1e-5)
    assert m.stats()[4] == 1


def test_kick_needs_kickable_ball_and_side_in_set_play():
    m = fresh()
    m.step(acts(p21=[MCMD_KICK, 100, 0]))          # right player is not the taker: ignored even if close
    assert m.get('mode')[0] == GM_KICK_OFF and m.get('x')[0][22] == 0 and m.get('setplay_timer')[0] == 1
    m.step(acts(p5=[MCMD_KICK, 100, 0]))           # far away: not kickable
    assert m.get('x')[0][22] == 0 and m.stats()[4] == 0


def test_drop_ball_after_timeout():
    m = fresh(drop_ball_time=5)
    for _ in range(5):
        m.step(acts())
        assert m.get('mode')[0] == GM_KICK_OFF
    m.step(acts())
    assert m.get('mode')[0] == GM_PLAY_ON


def test_goal_and_kickoff_for_conceding_side():
    # after_goal_wait = 0: the kick-off formation follows the goal at once
    m = fresh(after_goal_wait=0)
    play_on(m)
    m.set_obj(0, 22, x=52.0, y=1.0, vx=1.5, vy=0.0)
    m.step(acts())
    assert m.get('score_left')[0] == 1 and m.get('reward_left')[0] == 1.0
    assert m.get('mode')[0] == GM_KICK_OFF and m.get('mode_side')[0] == RIGHT
    assert m.get('x')[0][22] == 0 and m.get('x')[0][21] == pytest.approx(0.4) and m.get('x')[0][10] == pytest.approx(-10.5)
    m2 = fresh(after_goal_wait=0); play_on(m2)
    m2.set_obj(0, 22, x=-52.0, y=-6.9, vx=-1.5, vy=0.0)
    m2.step(acts())
    assert m2.get('score_right')[0] == 1 and m2.get('reward_left')[0] == -1.0 and m2.get('mode_side')[0] == LEFT


def test_after_goal_pause():
    # AfterGoal_ (idl/service.proto:276) for after_goal_wait cycles: mode side = the scorer, the ball rests in the net and
    # is dead for both sides, players may walk and Move inside their own half; then formation + kick-off for the conceding side
    m = fresh(after_goal_wait=4)
    play_on(m)
    m.set_obj(0, 22, x=52.0, y=1.0, vx=1.5, vy=0.0)
    m.set_obj(0, 9, x=52.9, y=1.0, body=0.0)                # a left attacker next to where the ball comes to rest
    m.step(acts())
    assert m.get('score_left')[0] == 1 and m.get('reward_left')[0] == 1.0
    assert m.get('mode')[0] == GM_AFTER_GOAL and m.get('mode_side')[0] == LEFT
    assert m.get('x')[0][22] == pytest.approx(53.5) and m.get('vx')[0][22] == 0 and m.get('x')[0][10] != pytest.approx(-10.5)
    for k in range(3):
        # the scorer's kick is dropped (dead ball); a Move of the conceding side inside its own half (own frame) is executed
        m.step(acts(p9=(MCMD_KICK, 100.0, 0.0), p15=(MCMD_MOVE, -20.0, 5.0)))
        assert m.get('mode')[0] == GM_AFTER_GOAL and m.get('reward_left')[0] == 0.0 and m.stats()[4] == 0
        assert m.get('x')[0][22] == pytest.approx(53.5) and m.get('vx')[0][22] == 0
        assert m.get('x')[0][15] == pytest.approx(20.0) and m.get('y')[0][15] == pytest.approx(-5.0)
    m.step(acts())
    assert m.get('mode')[0] == GM_KICK_OFF and m.get('mode_side')[0] == RIGHT and m.get('score_left')[0] == 1
    assert m.get('x')[0][22] == 0 and m.get('x')[0][21] == pytest.approx(0.4) and m.get('x')[0][10] == pytest.approx(-10.5)
    # the default is rcssserver's 50 cycles
    d = fresh(); play_on(d)
    d.set_obj(0, 22, x=-52.0, y=-6.9, vx=-1.5, vy=0.0)
    for k in range(50):
        d.step(acts())
        assert d.get('mode')[0] == GM_AFTER_GOAL and d.get('mode_side')[0] == RIGHT and d.get('score_right')[0] == 1
    d.step(acts())
    assert d.get('mode')[0] == GM_KICK_OFF and d.get('mode_side')[0] == LEFT


def test_ball_out_restarts():
    # side line -> kick-in for the side that did not touch last
    m = fresh(); play_on(m); m.set_game(0, last_touch_side=LEFT)
    m.set_obj(0, 22, x=10.0, y=33.5, vx=0.0, vy=1.0)
    m.step(acts())
    assert m.get('mode')[0] == GM_KICK_IN and m.get('mode_side')[0] == RIGHT
    assert m.get('x')[0][22] == 10.0 and m.get('y')[0][22] == 34.0 and m.get('vx')[0][22] == 0
    # goal line wide of the posts, last touched by the defender (right defends +x) -> corner for left
    m = fresh(); play_on(m); m.set_game(0, last_touch_side=RIGHT)
    m.set_obj(0, 22, x=52.0, y=20.0, vx=1.0, vy=0.0)
    m.step(acts())
    assert m.get('mode')[0] == GM_CORNER_KICK and m.get('mode_side')[0] == LEFT
    assert m.get('x')[0][22] == 51.5 and m.get('y')[0][22] == 33.0
    # ... last touched by the attacker -> goal kick for the defender
    m = fresh(); play_on(m); m.set_game(0, last_touch_side=LEFT)
    m.set_obj(0, 22, x=52.0, y=-20.0, vx=1.0, vy=0.0)
    m.step(acts())
    assert m.get('mode')[0] == GM_GOAL_KICK and m.get('mode_side')[0] == RIGHT
    assert m.get('x')[0][22] == 47.0 and m.get('y')[0][22] == pytest.approx(-9.16)
    assert m.stats()[7] == 1


def test_free_kick_distance_is_enforced():
    m = fresh(); play_on(m); m.set_game(0, last_touch_side=LEFT)
    m.set_obj(0, 22, x=10.0, y=33.5, vx=0.0, vy=1.0)
    m.set_obj(0, 3, x=10.0, y=30.0)                # left player near the spot: must retreat (right takes)
    m.step(acts())
    m.step(acts())
    d = np.hypot(m.get('x')[0][3] - 10.0, m.get('y')[0][3] - 34.0)
    assert d == pytest.approx(9.15, rel=1e-6)


def test_tackle_freezes_and_back_tackle_fails():
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-19.0, y=-22.0)             # 1 m in front of left #6 (index 5, body 0)
    succ = 0
    for s in range(40):
        mm = fresh(seed=s); play_on(mm)
        mm.set_obj(0, 22, x=-19.0, y=-22.0)
        mm.step(acts(p5=[MCMD_TACKLE, 0, 0]))
        assert mm.get('tackle_cycles')[0][5] == 10
        succ += mm.get('vx')[0][22] > 0
        mm.step(acts(p5=[MCMD_DASH, 100, 0]))      # frozen: the dash is ignored
        assert mm.get('vx')[0][5] == 0 and mm.get('tackle_cycles')[0][5] == 9
    # fail probability (1/2)^6 + 0 = 1.6 %: nearly always succeeds
    assert succ >= 36
    mm = fresh(); play_on(mm)
    mm.set_obj(0, 22, x=-21.0, y=-22.0)            # behind the player: tackle_back_dist = 0 -> never
    mm.step(acts(p5=[MCMD_TACKLE, 0, 0]))
    assert mm.get('vx')[0][22] == 0 and mm.get('tackle_cycles')[0][5] == 10


def test_player_player_and_player_ball_collisions():
    m = fresh(); play_on(m)
    m.set_obj(0, 1, x=0.0, y=20.0); m.set_obj(0, 12, x=0.4, y=20.0)      # overlap (0.4 < 0.6)
    m.step(acts())
    x = m.get('x')[0]
    assert x[12] - x[1] == pytest.approx(0.6, rel=1e-6) and (x[1] + x[12]) / 2 == pytest.approx(0.2, abs=1e-6)
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=5.0, y=5.0, vx=0.5); m.set_obj(0, 15, x=5.6, y=5.0)
    m.step(acts())
    assert m.get('last_touch_side')[0] == RIGHT and m.get('vx')[0][22] == pytest.approx(0.5 * -0.1 * 0.94, rel=1e-5)
    assert m.get('x')[0][15] - m.get('x')[0][22] == pytest.approx(0.385, rel=1e-5)


def test_offside_is_called():
    m = fresh(); play_on(m)
    # left #10 (index 9) passes from midfield; left #11 (index 10) waits behind the last-but-one defender
    for i in range(11, 22):
        m.set_obj(0, i, x=30.0 - (i - 11), y=-30.0 + 2 * (i - 11))       # defenders at x = 30 .. 20
    m.set_obj(0, 9, x=5.0, y=0.0, body=0.0)
    m.set_obj(0, 22, x=5.5, y=0.0)
    m.set_obj(0, 10, x=35.0, y=0.0)                                      # beyond 29 (second-last) and the ball
    m.step(acts(p9=[MCMD_KICK, 100, 0]))
    assert m.get('offside_mask')[0] == 1 << 10 and m.get('mode')[0] == GM_PLAY_ON
    m.set_obj(0, 22, x=34.0, y=0.5, vx=0.0, vy=0.0)                      # ball arrives at the flagged player
    m.step(acts())
    assert m.get('mode')[0] == GM_OFF_SIDE and m.get('mode_side')[0] == LEFT and m.stats()[6] == 1      # offside_l: named after the offender
    assert m.get('x')[0][22] == pytest.approx(m.get('x')[0][10]) and m.get('offside_mask')[0] == 0
    # the announcement: 30 cycles of dead ball with the clock stopped, then the free kick for the other side
    from soccer2d_amd._capi_match import GM_FREE_KICK
    c0 = int(m.get('cycle')[0])
    for k in range(29):
        m.step(acts(p12=[MCMD_KICK, 100, 0], p10=[MCMD_KICK, 100, 0]))
        assert m.get('mode')[0] == GM_OFF_SIDE and m.get('cycle')[0] == c0 and m.get('stopped_cycle')[0] == k + 1
        # the offending side is cleared from the ball (free_kick_distance), the side that will take the free kick is not
        bx, by = m.get('x')[0][22], m.get('y')[0][22]
        assert np.hypot(m.get('x')[0][10] - bx, m.get('y')[0][10] - by) == pytest.approx(9.15, rel=1e-6)
    m.step(acts())
    assert m.get('mode')[0] == GM_FREE_KICK and m.get('mode_side')[0] == RIGHT and m.get('cycle')[0] == c0 and m.get('stopped_cycle')[0] == 30
    m.step(acts())
    assert m.get('cycle')[0] == c0 + 1 and m.get('stopped_cycle')[0] == 0 and m.get('tick')[0] == c0 + 31
    # not offside from a kick-in
    m = fresh(); m.set_game(0, mode=GM_KICK_IN, mode_side=LEFT)
    for i in range(11, 22):
        m.set_obj(0, i, x=30.0 - (i - 11), y=-30.0 + 2 * (i - 11))
    m.set_obj(0, 9, x=5.0, y=34.0, body=-90.0); m.set_obj(0, 22, x=5.0, y=33.5); m.set_obj(0, 10, x=35.0, y=0.0)
    m.step(acts(p9=[MCMD_KICK, 50, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('offside_mask')[0] == 0


def test_half_time_and_time_over():
    m = fresh(half_time_cycles=20, auto_reset=0, nr_extra_halfs=0, penalty_shoot_outs=0)     # (test_extra_time_after_a_draw, test_penalty_shoot_out)
    play_on(m)
    for _ in range(19):
        m.step(acts(p3=[MCMD_DASH, 100, 0]))
    assert m.get('stamina')[0][3] < 8000 and m.get('cycle')[0] == 19
    from soccer2d_amd._capi_match import GM_FIRST_HALF_OVER
    m.step(acts())
    # half time: one cycle of FirstHalfOver (clock stopped), then the kick-off of the side that did not start the match
    assert m.get('cycle')[0] == 20 and m.get('mode')[0] == GM_FIRST_HALF_OVER and m.get('mode_side')[0] == RIGHT
    assert m.get('stamina')[0][3] == 8000 and m.get('stamina_capacity')[0][3] < 130600      # capacity is not restored
    m.step(acts())
    assert m.get('cycle')[0] == 20 and m.get('stopped_cycle')[0] == 1 and m.get('mode')[0] == GM_KICK_OFF and m.get('mode_side')[0] == RIGHT
    for _ in range(20):
        m.step(acts())
    assert m.get('mode')[0] == GM_TIME_OVER and m.get('done')[0] == 1 and m.get('cycle')[0] == 40
    m.step(acts(p3=[MCMD_DASH, 100, 0]))
    assert m.get('done')[0] == 0 and m.get('vx')[0][3] == 0            # time over: commands ignored
    assert m.get('cycle')[0] == 40 and m.get('stopped_cycle')[0] == 1  # ... and the clock stands
    m = fresh(half_time_cycles=5, auto_reset=1, nr_extra_halfs=0, penalty_shoot_outs=0)
    for _ in range(11):                                                 # 5 + FirstHalfOver + 5
        m.step(acts())
    assert m.get('done')[0] == 1 and m.get('cycle')[0] == 0 and m.get('mode')[0] == GM_KICK_OFF and m.stats()[3] == 1
    assert m.get('tick')[0] == 11                                       # the draws of the next match continue the sequence
    # stopped_clock = 0: the round-2 behaviour, time runs in every mode
    m = fresh(half_time_cycles=20, auto_reset=0, stopped_clock=0); play_on(m)
    for _ in range(22):
        m.step(acts())
    assert m.get('cycle')[0] == 22 and m.get('stopped_cycle')[0] == 0 and m.get('mode')[0] == GM_KICK_OFF


def test_extra_time_after_a_draw():
    """ServerParam.nr_extra_halfs / extra_half_time / golden_goal (idl/service.proto:1601, 1622, 1635) and GameModeType.ExtendHalf
    (:299; rcssserver's "time_extended"): a draw after the normal time is extended by nr_extra_halfs halves -- one stopped cycle
    of ExtendHalf, a kick-off for the side that started the match, FirstHalfOver between the extra halves, TimeOver after the last
    one whatever the score (no shoot-out); a decided match ends with the normal time.  Rules restated: parity unpinned."""
    from soccer2d_amd._capi_match import GM_EXTEND_HALF, GM_FIRST_HALF_OVER, GM_AFTER_GOAL
    m = fresh(half_time_cycles=10, extra_half_cycles=6, auto_reset=0, penalty_shoot_outs=0)  # stock nr_extra_halfs = 2; no shoot-out here
    seen = []
    for _ in range(60):
        m.step(acts())
        seen.append((int(m.get('cycle')[0]), int(m.get('mode')[0]), int(m.get('mode_side')[0])))
        if m.get('done')[0]:
            break
    ends = [(c, md, sd) for c, md, sd in seen if md in (GM_FIRST_HALF_OVER, GM_EXTEND_HALF, GM_TIME_OVER)]
    assert ends == [(10, GM_FIRST_HALF_OVER, RIGHT), (20, GM_EXTEND_HALF, LEFT), (26, GM_FIRST_HALF_OVER, RIGHT), (32, GM_TIME_OVER, 0)]
    assert seen[seen.index((20, GM_EXTEND_HALF, LEFT)) + 1] == (20, GM_KICK_OFF, LEFT)        # one stopped cycle, then the kick-off
    # a decided match ends with the normal time
    m = fresh(half_time_cycles=10, extra_half_cycles=6, auto_reset=0); m.set_game(0, score_left=1)
    for _ in range(21):                                                  # 10 + FirstHalfOver + 10
        m.step(acts())
    assert m.get('mode')[0] == GM_TIME_OVER and m.get('done')[0] == 1 and m.get('cycle')[0] == 20
    # extra halves are played in full: a goal in extra time does not end the match ...
    def goal_in_extra_time(**kw):
        m = fresh(half_time_cycles=10, extra_half_cycles=6, auto_reset=0, after_goal_wait=2, **kw)
        for _ in range(23):                                              # ... 20, ExtendHalf, kick-off cycle -> cycle 21
            m.step(acts())
        assert m.get('cycle')[0] == 21 and m.get('mode')[0] in (GM_KICK_OFF, GM_PLAY_ON)
        play_on(m)
        m.set_obj(0, 22, x=52.0, y=0.0, vx=2.0, vy=0.0)                  # the ball rolls over the right goal line
        m.step(acts())
        return m
    m = goal_in_extra_time()
    assert m.get('score_left')[0] == 1 and m.get('mode')[0] == GM_AFTER_GOAL and m.get('done')[0] == 0
    for _ in range(40):
        m.step(acts())
        if m.get('done')[0]:
            break
    assert m.get('mode')[0] == GM_TIME_OVER and m.get('cycle')[0] == 32
    # ... unless golden_goal
    m = goal_in_extra_time(golden_goal=1)
    assert m.get('score_left')[0] == 1 and m.get('mode')[0] == GM_TIME_OVER and m.get('done')[0] == 1 and m.get('cycle')[0] == 22


def _pen_word(m, e=0):
    w = int(m.get('set_play_taker')[e])
    return dict(taker=(w & 0xff) - 1, kicks=((w >> 12) & 15, (w >> 16) & 15), goals=((w >> 20) & 15, (w >> 24) & 15))


def test_penalty_shoot_out():
    """ServerParam.penalty_shoot_outs / pen_* (idl/service.proto:1602-1613), GameModeType PenaltySetup_ ... PenaltyOnfield_ (:290-297),
    PenaltyKickState (:130-138): a draw after the last period goes to the shoot-out -- PenaltyOnfield_, then per kick PenaltySetup_
    (everybody placed by the referee), PenaltyReady_, PenaltyTaken_ (taker against goalie; nobody else moves), PenaltyScore_ /
    PenaltyMiss_; the left team first, takers from index 10 downwards, pen_nr_kicks each, decided early when one side cannot catch
    up, then pairs of extra kicks; the state is in the set-play word.  rcssserver's PenaltyRef restated: parity unpinned."""
    from soccer2d_amd._capi_match import (GM_PENALTY_MISS, GM_PENALTY_ONFIELD, GM_PENALTY_READY, GM_PENALTY_SCORE, GM_PENALTY_SETUP,
                                          GM_PENALTY_TAKEN, MCMD_CATCH)
    kw = dict(half_time_cycles=6, nr_extra_halfs=0, auto_reset=0, pen_before_setup_wait=2, pen_ready_wait=3, pen_taken_wait=30,
              pen_nr_kicks=2, pen_max_extra_kicks=1)

    def to_ready(m, side, taker):
        """from the verdict (or PenaltyOnfield_) through PenaltySetup_ to PenaltyReady_"""
        for _ in range(2):
            m.step(acts(p3=[MCMD_DASH, 100, 0]))
        assert m.get('mode')[0] == GM_PENALTY_SETUP and m.get('mode_side')[0] == side and _pen_word(m)['taker'] == taker
        goalie = 11 if side == LEFT else 0
        x, y, body = m.get('x')[0], m.get('y')[0], m.get('body')[0]
        assert (x[22], y[22]) == (10.0, 0.0) and x[taker] == np.float32(10.0 - 0.7) and y[taker] == 0 and body[taker] == 0
        assert (x[goalie], y[goalie], body[goalie]) == (51.5, 0.0, 180.0)
        rest = [i for i in range(22) if i not in (taker, goalie)]
        assert (np.hypot(x[rest], y[rest]) < 9.15).all() and (m.get('vx')[0][:23] == 0).all()
        m.step(acts(p3=[MCMD_DASH, 100, 0], **{f'p{goalie}': [MCMD_DASH, 100, 0]}))
        assert m.get('mode')[0] == GM_PENALTY_READY and m.get('vx')[0][3] == 0 and m.get('vx')[0][goalie] == 0   # nobody but the taker acts

    def kick_and(m, taker, ball, expect):
        m.step(acts(**{f'p{taker}': [MCMD_KICK, 100, 0]}))
        assert m.get('mode')[0] == GM_PENALTY_TAKEN and m.get('vx')[0][22] > 1.0
        m.set_obj(0, 22, **ball)
        m.step(acts(p3=[MCMD_DASH, 100, 0]))
        assert m.get('mode')[0] == expect and m.get('vx')[0][3] == 0 and m.get('vx')[0][22] == 0

    m = fresh(**kw)
    for _ in range(13):                                                 # 6 + FirstHalfOver + 6: 0-0
        m.step(acts())
    assert m.get('mode')[0] == GM_PENALTY_ONFIELD and m.get('mode_side')[0] == RIGHT and m.get('done')[0] == 0 and m.get('cycle')[0] == 12
    to_ready(m, LEFT, 10)
    kick_and(m, 10, dict(x=52.0, y=1.0, vx=2.0, vy=0.0), GM_PENALTY_SCORE)
    assert _pen_word(m) == dict(taker=10, kicks=(1, 0), goals=(1, 0)) and m.get('reward_left')[0] == 1 and m.get('score_left')[0] == 0
    to_ready(m, RIGHT, 21)
    for _ in range(3):                                                  # the taker lets pen_ready_wait pass
        m.step(acts())
    assert m.get('mode')[0] == GM_PENALTY_MISS and _pen_word(m)['kicks'] == (1, 1)
    to_ready(m, LEFT, 9)
    kick_and(m, 9, dict(x=30.0, y=33.9, vx=0.0, vy=1.0), GM_PENALTY_MISS)             # over the side line
    to_ready(m, RIGHT, 20)
    kick_and(m, 20, dict(x=52.4, y=-6.9, vx=1.0, vy=0.0), GM_PENALTY_SCORE)
    assert _pen_word(m) == dict(taker=20, kicks=(2, 2), goals=(1, 1)) and m.get('reward_left')[0] == -1
    to_ready(m, LEFT, 8)                                                # 1-1 after the regular kicks: one pair of extra kicks
    m.step(acts(p8=[MCMD_KICK, 30, 0]))
    assert m.get('mode')[0] == GM_PENALTY_TAKEN
    m.set_obj(0, 22, x=50.9, y=0.0, vx=0.0, vy=0.0)                      # in front of the goalie, who catches it
    m.step(acts(p11=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_PENALTY_MISS
    to_ready(m, RIGHT, 19)
    m.step(acts(p19=[MCMD_KICK, 100, 0]))
    for _ in range(31):                                                 # ... and this one runs out of time
        assert m.get('mode')[0] == GM_PENALTY_TAKEN
        m.set_obj(0, 22, x=20.0, y=0.0, vx=0.0, vy=0.0)
        m.step(acts())
    assert m.get('mode')[0] == GM_PENALTY_MISS and _pen_word(m)['kicks'] == (3, 3) and m.get('cycle')[0] == 12   # the clock stood
    for _ in range(2):
        m.step(acts())
    assert m.get('mode')[0] == GM_TIME_OVER and m.get('done')[0] == 1 and _pen_word(m)['goals'] == (1, 1)      # used up: the draw stands
    # decided early: 2-0 after three kicks
    m = fresh(**kw)
    for _ in range(13):
        m.step(acts())
    to_ready(m, LEFT, 10); kick_and(m, 10, dict(x=52.0, y=0.0, vx=2.0, vy=0.0), GM_PENALTY_SCORE)
    to_ready(m, RIGHT, 21); kick_and(m, 21, dict(x=52.0, y=20.0, vx=2.0, vy=0.0), GM_PENALTY_MISS)   # over the goal line, beside the goal
    to_ready(m, LEFT, 9); kick_and(m, 9, dict(x=52.0, y=0.0, vx=2.0, vy=0.0), GM_PENALTY_SCORE)
    for _ in range(2):
        m.step(acts())
    assert m.get('mode')[0] == GM_TIME_OVER and m.get('done')[0] == 1 and _pen_word(m) == dict(taker=9, kicks=(2, 1), goals=(2, 0))
    # pen_allow_mult_kicks = 0 (idl/service.proto:1611): a second touch of the kicker is PenaltyFoul_ (:297) and the kick is missed
    from soccer2d_amd._capi_match import GM_PENALTY_FOUL
    for allow in (1, 0):
        m = fresh(pen_allow_mult_kicks=allow, **kw)
        for _ in range(13):
            m.step(acts())
        to_ready(m, LEFT, 10)
        m.step(acts(p10=[MCMD_KICK, 20, 0]))
        assert m.get('mode')[0] == GM_PENALTY_TAKEN
        m.set_obj(0, 22, x=float(m.get('x')[0][10]) + 0.6, y=0.0, vx=0.0, vy=0.0)      # the ball at his feet again
        m.step(acts(p10=[MCMD_KICK, 100, 0]))
        assert m.get('mode')[0] == (GM_PENALTY_TAKEN if allow else GM_PENALTY_FOUL)
        assert _pen_word(m)['kicks'] == ((0, 0) if allow else (1, 0)) and _pen_word(m)['goals'] == (0, 0)
        if not allow:
            assert m.get('vx')[0][22] == 0
            to_ready(m, RIGHT, 21)                                       # the verdict stands, then the other side's kick
    # penalty_shoot_outs = 0: the draw stands at once
    m = fresh(penalty_shoot_outs=0, **{k: v for k, v in kw.items()})
    for _ in range(13):
        m.step(acts())
    assert m.get('mode')[0] == GM_TIME_OVER and m.get('done')[0] == 1


def test_pen_random_winner_tosses_a_coin():
    """ServerParam.pen_random_winner (idl/service.proto:1610): a shoot-out that ends level after every kick is decided by a coin --
    here bits 28-29 of the set-play word (1 = left, 2 = right) in the cycle the match ends; the score is untouched.  One kick each,
    nobody plays the ball: every match ends 0-0.  rcssserver's PenaltyRef restated (drand < 0.5 = left): parity unpinned."""
    kw = dict(half_time_cycles=4, nr_extra_halfs=0, auto_reset=0, pen_before_setup_wait=1, pen_ready_wait=2, pen_taken_wait=5,
              pen_nr_kicks=1, pen_max_extra_kicks=0)
    n = 400
    for on in (0, 1):
        m = fresh(n, pen_random_winner=on, **kw)
        for _ in range(40):
            m.step(acts(n))
        assert (m.get('mode') == GM_TIME_OVER).all()
        w = m.get('set_play_taker').astype(np.int64)
        assert (((w >> 12) & 15) == 1).all() and (((w >> 16) & 15) == 1).all() and (((w >> 20) & 255) == 0).all()
        win = (w >> 28) & 3
        if not on:
            assert (win == 0).all()                                    # the draw stands
        else:
            left, right = int((win == 1).sum()), int((win == 2).sum())
            assert left + right == n and abs(left - n / 2) < 4 * (n / 4) ** 0.5      # a fair coin, four sigma
    # a shoot-out that the kicks decide is not tossed for
    m = fresh(1, pen_random_winner=1, **kw)
    for _ in range(9):
        m.step(acts())
    from soccer2d_amd._capi_match import GM_PENALTY_READY, GM_PENALTY_TAKEN, GM_PENALTY_SCORE
    while m.get('mode')[0] != GM_PENALTY_READY:
        m.step(acts())
    m.step(acts(p10=[MCMD_KICK, 100, 0]))
    assert m.get('mode')[0] == GM_PENALTY_TAKEN
    m.set_obj(0, 22, x=52.0, y=0.0, vx=2.0, vy=0.0)
    m.step(acts())
    assert m.get('mode')[0] == GM_PENALTY_SCORE
    for _ in range(30):
        m.step(acts())
    assert m.get('mode')[0] == GM_TIME_OVER and _pen_word(m)['goals'] == (1, 0) and (int(m.get('set_play_taker')[0]) >> 28) == 0


def test_operator_called_fouls_are_played_like_announcements():
    """FoulPush_ / FoulMultipleAttacker_ / FoulBallOut_ (idl/service.proto:283-285): rcssserver defines them, its referees call only
    FoulCharge_, and so does this engine's.  Written into the mode word with the offending side they are played as an announcement:
    dead ball, the offenders cleared from it, after announce_wait a FreeKick_ for the other side."""
    from soccer2d_amd._capi_match import GM_FOUL_BALL_OUT, GM_FOUL_MULTIPLE_ATTACKER, GM_FOUL_PUSH, GM_FREE_KICK
    for md in (GM_FOUL_PUSH, GM_FOUL_MULTIPLE_ATTACKER, GM_FOUL_BALL_OUT):
        m = fresh(auto_reset=0, announce_wait=4); play_on(m)
        m.set_obj(0, 22, x=10.0, y=5.0, vx=0.0, vy=0.0); m.set_obj(0, 14, x=10.5, y=5.0); m.set_obj(0, 9, x=9.0, y=5.0)
        m.set_game(0, mode=md, mode_side=RIGHT)
        cyc = int(m.get('cycle')[0])
        for k in range(3):
            m.step(acts(p14=[MCMD_KICK, 100, 0], p9=[MCMD_KICK, 100, 180]))
            assert m.get('mode')[0] == md and m.get('cycle')[0] == cyc and m.get('vx')[0][22] == 0       # nobody plays a dead ball
        assert np.hypot(m.get('x')[0][14] - 10.0, m.get('y')[0][14] - 5.0) >= 9.15 - 1e-3                  # the offenders keep away
        m.step(acts())
        assert m.get('mode')[0] == GM_FREE_KICK and m.get('mode_side')[0] == LEFT


def test_pause_and_human_hold_a_match():
    """Pause / Human (idl/service.proto:280-281): modes only an operator sets.  Written into the mode word they hold the match -- commands
    ignored, nothing decided, the clock stands, no timer counts -- until another mode is written."""
    from soccer2d_amd._capi_match import GM_HUMAN, GM_PAUSE
    for held in (GM_PAUSE, GM_HUMAN):
        m = fresh(auto_reset=0); play_on(m)
        for _ in range(5):
            m.step(acts(p3=[MCMD_DASH, 100, 0]))
        x3, cyc = float(m.get('x')[0][3]), int(m.get('cycle')[0])
        m.set_game(0, mode=held)
        m.set_obj(0, 3, vx=0.0, vy=0.0)
        for k in range(7):
            m.step(acts(p3=[MCMD_DASH, 100, 0], p9=[MCMD_KICK, 100, 0]))
            assert m.get('mode')[0] == held and m.get('cycle')[0] == cyc and m.get('stopped_cycle')[0] == k + 1 and m.get('done')[0] == 0
        assert float(m.get('x')[0][3]) == x3 and m.get('setplay_timer')[0] == 0
        play_on(m)
        m.step(acts(p3=[MCMD_DASH, 100, 0]))
        assert m.get('cycle')[0] == cyc + 1 and float(m.get('x')[0][3]) > x3


def test_illegal_defense():
    """IllegalDefense_ (idl/service.proto:295; ServerParam.illegal_defense_number / _duration / _dist_x / _width :1637-1640): off in the
    stock server (number = 0).  Switched on: a team that packs its own goal mouth while the other team has the ball is called after
    `duration` cycles on end -- announcement named after it, ball on that half's penalty spot, then a FreeKick_ for the others; the
    count starts again when the defenders play the ball or leave.  rcssserver's IllegalDefenseRef restated: parity unpinned."""
    from soccer2d_amd._capi_match import GM_FREE_KICK, GM_ILLEGAL_DEFENSE
    def scene(**kw):
        m = fresh(auto_reset=0, announce_wait=3, **kw); play_on(m)
        for i in range(22):                                             # everybody far from the ball and from both goal mouths
            m.set_obj(0, i, x=(-20.0 if i < 11 else 20.0) - (i % 11), y=-25.0 + 2.0 * (i % 11), vx=0.0, vy=0.0)
        for k, i in enumerate((1, 2, 3, 4)):                            # four left defenders inside their own strip
            m.set_obj(0, i, x=-45.0 - k, y=-6.0 + 4.0 * k)
        m.set_obj(0, 22, x=0.0, y=30.0, vx=0.0, vy=0.0)
        m.set_game(0, last_touch_side=RIGHT)
        return m
    m = scene()                                                         # stock: the rule is off
    for _ in range(30):
        m.step(acts())
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('setplay_timer')[0] == 0
    m = scene(illegal_defense_number=4, illegal_defense_duration=5)
    for k in range(4):
        m.step(acts())
        assert m.get('mode')[0] == GM_PLAY_ON and m.get('setplay_timer')[0] == k + 1          # the left team's count (bits 0-7)
    m.step(acts())
    assert m.get('mode')[0] == GM_ILLEGAL_DEFENSE and m.get('mode_side')[0] == LEFT and m.get('setplay_timer')[0] == 0
    assert (m.get('x')[0][22], m.get('y')[0][22]) == (np.float32(-41.5), 0.0)
    for _ in range(3):
        m.step(acts())
    assert m.get('mode')[0] == GM_FREE_KICK and m.get('mode_side')[0] == RIGHT
    # three defenders are not enough; neither are four while their own team was the last to play the ball
    m = scene(illegal_defense_number=4, illegal_defense_duration=5); m.set_obj(0, 4, x=-20.0, y=0.0)
    for _ in range(12):
        m.step(acts())
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('setplay_timer')[0] == 0
    m = scene(illegal_defense_number=4, illegal_defense_duration=5); m.set_game(0, last_touch_side=LEFT)
    for _ in range(12):
        m.step(acts())
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('setplay_timer')[0] == 0
    # the count starts again when a defender leaves the strip for a cycle; the right team is counted in bits 8-15
    m = scene(illegal_defense_number=4, illegal_defense_duration=5)
    for _ in range(3):
        m.step(acts())
    m.set_obj(0, 1, x=-30.0, y=0.0); m.step(acts()); assert m.get('setplay_timer')[0] == 0
    m.set_obj(0, 1, x=-45.0, y=-6.0); m.step(acts()); assert m.get('setplay_timer')[0] == 1
    m = scene(illegal_defense_number=2, illegal_defense_duration=9); m.set_game(0, last_touch_side=LEFT)
    m.set_obj(0, 12, x=50.0, y=3.0); m.set_obj(0, 13, x=40.0, y=-19.0)
    m.step(acts()); m.step(acts())
    assert m.get('setplay_timer')[0] == 2 << 8 and m.get('mode')[0] == GM_PLAY_ON


def test_random_matches_are_deterministic_and_eventful():
    n = 64
    kw = dict(half_time_cycles=400, extra_half_cycles=50, pen_before_setup_wait=2, pen_taken_wait=20, pen_nr_kicks=1, pen_max_extra_kicks=1)
    a, b = fresh(n, **kw), fresh(n, **kw)                  # (short extra halves and a short shoot-out: most random matches are draws)
    for _ in range(1300):                                  # 800 cycles of play + the stopped ones (after goals, offside calls, half time)
        a.step(None); b.step(None)
    for f in MO.OBJ_FIELDS + MO.ENV_FIELDS:
        assert np.array_equal(a.get(f), b.get(f)), f
    st = a.stats()
    assert st[0] == n * 1300 and st[3] >= n // 2 and st[4] > 0 and st[5] > 0 and st[7] > 0
    assert np.isfinite(a.get('x')).all() and np.abs(a.get('x')[:, :22]).max() < 80
    acts0 = a.random_actions()
    assert set(np.unique(acts0[..., 0])) <= {1.0, 2.0, 3.0, 4.0}


def test_nearest_player_reduction():
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-34.0, y=6.0)
    m.step(acts())
    assert m.get('nearest_left')[0] == 3 and m.get('nearest_right')[0] in range(11, 22)
    d = np.hypot(m.get('x')[0][11:22] - m.get('x')[0][22], m.get('y')[0][11:22] - m.get('y')[0][22])
    assert m.get('nearest_right')[0] == 11 + int(np.argmin(d))


# ------------------------------------------------------------------ goalie catch (Catch{}, idl/service.proto:404)
def test_goalie_catch_gives_free_kick_and_bans_catching():
    from soccer2d_amd._capi_match import GM_FREE_KICK, MCMD_CATCH
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-49.2, y=0.3, vx=-1.0, vy=0.0)      # ball 0.8 m in front of the left goalie (-50, 0), moving in
    from soccer2d_amd._capi_match import GM_GOALIE_CATCH
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_GOALIE_CATCH and m.get('mode_side')[0] == LEFT and m.get('last_touch_side')[0] == LEFT
    # held: the ball rests where it was caught, no goal is scored, the goalie is banned for 5 cycles
    assert m.get('vx')[0][22] == 0 and m.get('x')[0][22] == pytest.approx(-49.2) and m.get('score_right')[0] == 0
    assert m.get('catch_ban')[0][0] == 5 and m.stats()[4] == 1
    # one cycle of GoalieCatch_ (the clock runs), then the goalie's free kick: opponents are kept 9.15 m away; his kick resumes play
    m.set_obj(0, 21, x=-48.0, y=0.3)
    c0 = int(m.get('cycle')[0])
    m.step(acts(p0=[MCMD_KICK, 100, 0]))                    # not yet: the ball is dead for this one cycle
    assert m.get('mode')[0] == GM_FREE_KICK and m.get('mode_side')[0] == LEFT and m.get('cycle')[0] == c0 + 1 and m.get('vx')[0][22] == 0
    assert np.hypot(m.get('x')[0][21] + 49.2, m.get('y')[0][21] - 0.3) == pytest.approx(9.15, rel=1e-6)
    m.step(acts(p0=[MCMD_KICK, 100, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('vx')[0][22] > 0


def test_catch_needs_goalie_rectangle_and_no_ban():
    from soccer2d_amd._capi_match import GM_FREE_KICK, MCMD_CATCH
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-33.8, y=-20.0)                     # next to left #2 (index 1, a field player)
    m.step(acts(p1=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('catch_ban')[0][1] == 0
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-48.5, y=0.0)                       # 1.5 m ahead: beyond catchable_area_l = 1.2
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('catch_ban')[0][0] == 5   # the attempt still starts the ban
    m.set_obj(0, 22, x=-49.5, y=0.0, vx=0.0, vy=0.0)
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))                    # banned: ignored although the ball is now in reach
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('catch_ban')[0][0] == 4
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-50.0, y=0.9)                       # beside the goalie: needs the catch direction
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-50.0, y=0.9)
    m.step(acts(p0=[MCMD_CATCH, 90, 0]))                   # dir = +90 deg turns the rectangle towards +y
    from soccer2d_amd._capi_match import GM_GOALIE_CATCH
    assert m.get('mode')[0] == GM_GOALIE_CATCH


def test_catch_outside_the_penalty_area_is_a_fault():
    from soccer2d_amd._capi_match import GM_FREE_KICK, MCMD_CATCH
    m = fresh(); play_on(m)
    m.set_obj(0, 0, x=-30.0, y=0.0)                        # goalie far out of his area (x > -36)
    m.set_obj(0, 22, x=-29.2, y=0.0)
    from soccer2d_amd._capi_match import GM_CATCH_FAULT
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_CATCH_FAULT and m.get('mode_side')[0] == LEFT           # catch_fault_l: the left goalie's fault
    for _ in range(30):
        m.step(acts())
    assert m.get('mode')[0] == GM_FREE_KICK and m.get('mode_side')[0] == RIGHT and m.get('ball_holder')[0] == 0


# ------------------------------------------------------------------ heterogeneous PlayerTypes
def test_player_types_change_the_dynamics_of_their_players_only():
    fast = {3: dict(dash_power_rate=0.0068, player_decay=0.5, effort_max=0.9, effort_min=0.5, extra_stamina=80.0,
                    stamina_inc_max=40.0, player_size=0.25, kickable_margin=0.8, kick_power_rate=0.03,
                    inertia_moment=7.5, player_speed_max=1.2)}
    ids = [0] * 22; ids[5] = 3
    m = MO.MatchOracle(MO.make_match_config(player_types=fast, player_type_id=ids), 1)
    play_on(m)
    assert m.get('effort')[0][5] == pytest.approx(0.9) and m.get('effort')[0][6] == 1.0     # recover -> effort_max of the type
    m.step(acts(p5=[MCMD_DASH, 100, 0], p6=[MCMD_DASH, 100, 0]))
    # acc = effort * power * dash_power_rate ; vel after decay
    assert m.get('vx')[0][5] == pytest.approx(0.9 * 100 * 0.0068 * 0.5, rel=1e-6)
    assert m.get('vx')[0][6] == pytest.approx(1.0 * 100 * 0.006 * 0.4, rel=1e-6)
    # stamina: 8000 - 100 + recovery * stamina_inc_max, clipped at stamina_max
    assert m.get('stamina')[0][5] == pytest.approx(8000 - 100 + 40.0) and m.get('stamina')[0][6] == pytest.approx(8000 - 100 + 45.0)
    # kickable area of the type: 0.25 + 0.085 + 0.8 = 1.135 (default 1.085)
    m = MO.MatchOracle(MO.make_match_config(player_types=fast, player_type_id=ids), 1); play_on(m)
    m.set_obj(0, 22, x=-20.0 + 1.12, y=-22.0)
    m.step(acts(p5=[MCMD_KICK, 50, 0]))
    assert m.get('vx')[0][22] > 0
    m = MO.MatchOracle(MO.make_match_config(), 1); play_on(m)
    m.set_obj(0, 22, x=-20.0 + 1.12, y=-22.0)
    m.step(acts(p5=[MCMD_KICK, 50, 0]))
    assert m.get('vx')[0][22] == 0


# ------------------------------------------------------------------ Move{x, y} (idl/service.proto:408-411)
def test_move_before_kick_off_and_with_a_caught_ball():
    from soccer2d_amd._capi_match import GM_FREE_KICK, MCMD_CATCH, MCMD_MOVE
    m = fresh()                                            # KickOff for the left side
    m.step(acts(p3=[MCMD_MOVE, -12.0, 5.0], p14=[MCMD_MOVE, -30.0, -4.0], p5=[MCMD_MOVE, 20.0, 50.0]))
    x, y = m.get('x')[0], m.get('y')[0]
    assert (x[3], y[3]) == (-12.0, 5.0)                    # left team: as given
    assert (x[14], y[14]) == (30.0, 4.0)                   # right team: mirrored frame
    assert (x[5], y[5]) == (0.0, 34.0)                     # clamped to the own half / the pitch
    assert m.get('mode')[0] == GM_KICK_OFF and (m.get('vx')[0][[3, 14, 5]] == 0).all()
    play_on(m)
    m.step(acts(p3=[MCMD_MOVE, -40.0, 0.0]))               # not legal in play_on: ignored
    assert m.get('x')[0][3] == -12.0
    # goalie catches, then carries the ball twice inside his area; the third move is refused
    m = fresh(); play_on(m)
    m.set_obj(0, 22, x=-49.3, y=0.0)
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))
    assert m.get('ball_holder')[0] == 1 and m.get('goalie_moves')[0] == 2
    m.step(acts(p0=[MCMD_MOVE, -40.0, 10.0]))              # GoalieCatch_ cycle: not yet
    assert m.get('mode')[0] == GM_FREE_KICK and m.get('x')[0][0] == -50.0 and m.get('goalie_moves')[0] == 2
    m.step(acts(p0=[MCMD_MOVE, -40.0, 10.0]))
    assert (m.get('x')[0][0], m.get('y')[0][0]) == (-40.0, 10.0) and m.get('goalie_moves')[0] == 1
    assert m.get('x')[0][22] == pytest.approx(-40.0 + 0.485) and m.get('y')[0][22] == pytest.approx(10.0)   # in front of the body (0 deg)
    m.step(acts(p0=[MCMD_MOVE, -20.0, 30.0]))              # clamped into the penalty area
    assert (m.get('x')[0][0], m.get('y')[0][0]) == (-36.0, pytest.approx(20.16)) and m.get('goalie_moves')[0] == 0
    m.step(acts(p0=[MCMD_MOVE, -45.0, 0.0]))
    assert m.get('x')[0][0] == -36.0
    m.step(acts(p0=[MCMD_KICK, 60, 0]))                    # the kick puts the ball in play and ends the hold
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('ball_holder')[0] == 0 and m.get('vx')[0][22] > 0


# ---- round 2: BeforeKickOff, free-kick fault, back pass (rcssserver rules restated; parity unpinned) ----------------------
def test_before_kick_off_mode_waits_and_lets_players_move():
    from soccer2d_amd._capi_match import GM_BEFORE_KICK_OFF
    m = fresh(kick_off_wait=4, half_time_cycles=20)
    assert m.get('mode')[0] == GM_BEFORE_KICK_OFF and m.get('mode_side')[0] == LEFT
    # nobody may play the ball (not even the side that will kick off); Move inside the own half is allowed
    m.step(acts(p10=[MCMD_KICK, 100, 0], p3=[MCMD_MOVE, -20.0, 5.0], p14=[MCMD_MOVE, -12.0, -3.0]))
    assert m.get('x')[0][22] == 0 and m.get('mode')[0] == GM_BEFORE_KICK_OFF and m.stats()[4] == 0
    assert m.get('x')[0][3] == pytest.approx(-20.0) and m.get('y')[0][3] == pytest.approx(5.0)
    assert m.get('x')[0][14] == pytest.approx(12.0) and m.get('y')[0][14] == pytest.approx(3.0)      # right team: mirrored frame
    for _ in range(2):
        m.step(acts())
        assert m.get('mode')[0] == GM_BEFORE_KICK_OFF
    m.step(acts())                                  # 4th cycle: the wait is over
    assert m.get('mode')[0] == GM_KICK_OFF and m.get('mode_side')[0] == LEFT and m.get('setplay_timer')[0] == 0
    m.step(acts(p10=[MCMD_KICK, 100, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON
    # the clock stood still during the wait: WorldModel.cycle 0, stoped_cycle counted the four cycles
    assert m.get('cycle')[0] == 1 and m.get('tick')[0] == 5
    # half time goes through FirstHalfOver and BeforeKickOff again, for the side that kicks off the second half
    from soccer2d_amd._capi_match import GM_FIRST_HALF_OVER
    while m.get('cycle')[0] < 20:
        m.step(acts())
    assert m.get('mode')[0] == GM_FIRST_HALF_OVER and m.get('mode_side')[0] == RIGHT
    m.step(acts())
    assert m.get('mode')[0] == GM_BEFORE_KICK_OFF and m.get('mode_side')[0] == RIGHT and m.get('cycle')[0] == 20
    for k in range(4):
        assert m.get('stopped_cycle')[0] == k + 1
        m.step(acts())
    assert m.get('mode')[0] == GM_KICK_OFF and m.get('mode_side')[0] == RIGHT


def test_free_kick_fault_on_a_second_touch_by_the_taker():
    from soccer2d_amd._capi_match import GM_FREE_KICK_FAULT
    m = fresh()
    m.step(acts(p10=[MCMD_KICK, 20, 0]))           # the kick-off taker plays the ball softly (0.54 m/cycle) ...
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('set_play_taker')[0] == 11
    m.step(acts(p10=[MCMD_DASH, 100, 0]))          # ... runs after it ...
    assert m.get('set_play_taker')[0] == 11 and m.get('mode')[0] == GM_PLAY_ON
    bx = m.get('x')[0][22]
    m.step(acts(p10=[MCMD_KICK, 50, 0]))           # ... and kicks again before anybody else touched it: fault
    assert m.get('mode')[0] == GM_FREE_KICK_FAULT and m.get('mode_side')[0] == LEFT and m.get('set_play_taker')[0] == 0   # free_kick_fault_l
    assert m.get('vx')[0][22] == 0 and m.get('x')[0][22] > bx          # ball placed where it was after the kick, at rest
    # the announcement is a dead ball for both sides; after announce_wait cycles: indirect free kick for the other side
    from soccer2d_amd._capi_match import GM_IND_FREE_KICK
    m.set_obj(0, 15, x=float(m.get('x')[0][22]) + 0.5, y=0.0, body=180.0)
    for _ in range(29):
        m.step(acts(p10=[MCMD_KICK, 50, 0], p15=[MCMD_KICK, 50, 0]))
        assert m.get('mode')[0] == GM_FREE_KICK_FAULT and m.get('vx')[0][22] == 0
    m.step(acts())
    assert m.get('mode')[0] == GM_IND_FREE_KICK and m.get('mode_side')[0] == RIGHT
    m.step(acts(p10=[MCMD_KICK, 50, 0]))                    # the offender's side cannot play it, the right team can
    assert m.get('mode')[0] == GM_IND_FREE_KICK
    m.step(acts(p15=[MCMD_KICK, 50, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON and m.get('set_play_taker')[0] == (16 | 0x100)      # taker #16 of an INDIRECT free kick


def test_no_fault_after_another_touch_and_switch():
    m = fresh()
    m.step(acts(p10=[MCMD_KICK, 20, 0]))
    # a team-mate touches the ball in between: the taker may play it again
    m.set_obj(0, 9, x=float(m.get('x')[0][22]) - 0.5, y=0.0, body=0.0)
    m.step(acts(p9=[MCMD_KICK, 10, 0]))
    assert m.get('set_play_taker')[0] == 0
    m.set_obj(0, 10, x=float(m.get('x')[0][22]) - 0.5, y=0.0, body=0.0)
    m.step(acts(p10=[MCMD_KICK, 10, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON
    # free_kick_faults = 0 switches the rule off
    m = fresh(free_kick_faults=0)
    m.step(acts(p10=[MCMD_KICK, 20, 0])); m.step(acts(p10=[MCMD_DASH, 100, 0])); m.step(acts(p10=[MCMD_KICK, 50, 0]))
    assert m.get('mode')[0] == GM_PLAY_ON


def test_back_pass_to_the_goalie_is_an_indirect_free_kick():
    from soccer2d_amd._capi_match import GM_BACK_PASS, GM_FREE_KICK, MCMD_CATCH
    m = fresh(); play_on(m)
    # left #3 (index 2) kicks the ball towards his own goalie ...
    m.set_obj(0, 2, x=-45.0, y=0.3, body=180.0)
    m.set_obj(0, 22, x=-45.5, y=0.3, vx=0.0, vy=0.0)
    m.step(acts(p2=[MCMD_KICK, 100, 0]))
    assert m.get('last_kicker')[0] == 3 and m.get('vx')[0][22] < -2.0
    m.step(acts())
    assert m.get('last_kicker')[0] == 3                    # nobody else touched it on the way
    m.set_obj(0, 22, x=-49.2, y=0.3, vx=-1.0, vy=0.0)      # (the rolling ball, placed in front of the goalie)
    # ... who catches it inside his penalty area: BackPass_ (named after the offending side), ball at the nearer front corner,
    # then an indirect free kick for the RIGHT side
    from soccer2d_amd._capi_match import GM_GOALIE_CATCH, GM_IND_FREE_KICK
    m.step(acts(p0=[MCMD_CATCH, 0, 0]))
    assert m.get('mode')[0] == GM_BACK_PASS and m.get('mode_side')[0] == LEFT
    assert m.get('x')[0][22] == pytest.approx(-36.0) and m.get('y')[0][22] == pytest.approx(20.16) and m.get('vx')[0][22] == 0
    assert m.get('ball_holder')[0] == 0 and m.get('last_kicker')[0] == 0
    w = fresh(announce_wait=3); play_on(w)
    w.set_game(0, last_kicker=3)
    w.set_obj(0, 22, x=-49.2, y=0.3, vx=-1.0, vy=0.0)
    w.step(acts(p0=[MCMD_CATCH, 0, 0]))
    for _ in range(3):
        assert w.get('mode')[0] == GM_BACK_PASS
        w.step(acts())
    assert w.get('mode')[0] == GM_IND_FREE_KICK and w.get('mode_side')[0] == RIGHT
    # an opponent's kick before the catch is no back pass; nor is the goalie's own kick; nor with back_passes = 0
    for kicker, kw in ((13, {}), (0, {}), (2, dict(back_passes=0))):
        m = fresh(**kw); play_on(m)
        m.set_game(0, last_kicker=kicker + 1)
        m.set_obj(0, 22, x=-49.2, y=0.3, vx=-1.0, vy=0.0)
        m.step(acts(p0=[MCMD_CATCH, 0, 0]))
        assert m.get('mode')[0] == GM_GOALIE_CATCH and m.get('mode_side')[0] == LEFT and m.get('ball_holder')[0] == 1
    # a tackle or a collision with another player in between ends the back pass
    m = fresh(); play_on(m)
    m.set_game(0, last_kicker=3)
    m.set_obj(0, 22, x=-30.0, y=0.0, vx=0.0, vy=0.0)
    m.set_obj(0, 15, x=-30.2, y=0.0)                # an opponent standing on the ball: collision touch
    m.step(acts())
    assert m.get('last_kicker')[0] == 0


# ---- round 3: fouls and cards (Tackle.foul, idl/service.proto:399-402; FoulCharge_ :282) -- rcssserver's rule restated, parity unpinned
def _foul_scene(**kw):
    from soccer2d_amd._capi_match import GM_PLAY_ON
    m = fresh(**kw); play_on(m)
    m.set_obj(0, 5, x=0.0, y=0.0, body=0.0)                 # left #6 faces +x ...
    m.set_obj(0, 15, x=1.0, y=0.2, body=180.0)              # ... a right player stands 1 m in front of him with the ball at his feet
    m.set_obj(0, 22, x=0.7, y=0.1, vx=0.0, vy=0.0)
    return m


def test_no_goal_directly_from_an_indirect_free_kick():
    """IndFreeKick_ (idl/service.proto:289): the taker shoots straight into the goal -> no goal, a goal kick for the defenders (the ball
    went over the goal line off an attacker); once another player has touched the ball, a goal counts.  A direct FreeKick_ scores.
    (rcssserver's rule as we restate it: parity unpinned.)"""
    from soccer2d_amd._capi_match import GM_AFTER_GOAL, GM_FREE_KICK, GM_GOAL_KICK, GM_IND_FREE_KICK, GM_PLAY_ON
    for mode, scores in ((GM_IND_FREE_KICK, False), (GM_FREE_KICK, True)):
        m = fresh(); play_on(m)
        m.set_game(0, mode=mode, mode_side=LEFT)
        m.set_obj(0, 9, x=45.5, y=0.2, body=0.0)             # left #10, seven metres in front of the right goal
        m.set_obj(0, 22, x=46.0, y=0.2, vx=0.0, vy=0.0)
        m.step(acts(p9=[MCMD_KICK, 100, 0]))
        assert m.get('mode')[0] == GM_PLAY_ON and (m.get('set_play_taker')[0] & 0xff) == 10
        assert bool(m.get('set_play_taker')[0] & 0x100) == (mode == GM_IND_FREE_KICK)
        for _ in range(6):
            if m.get('mode')[0] != GM_PLAY_ON:
                break
            m.step(acts())
        if scores:
            assert m.get('mode')[0] == GM_AFTER_GOAL and m.get('score_left')[0] == 1
        else:
            assert m.get('mode')[0] == GM_GOAL_KICK and m.get('mode_side')[0] == RIGHT and m.get('score_left')[0] == 0
            assert m.get('set_play_taker')[0] == 0
    # an indirect free kick that a team mate deflects in: the second touch clears the flag, the goal counts
    m = fresh(use_offside=0); play_on(m)                    # (the team mate in front of the goal would be offside)
    m.set_game(0, mode=GM_IND_FREE_KICK, mode_side=LEFT)
    m.set_obj(0, 9, x=44.0, y=0.2, body=0.0)
    m.set_obj(0, 22, x=44.5, y=0.2, vx=0.0, vy=0.0)
    m.set_obj(0, 10, x=48.0, y=0.2, body=0.0)                # left #11 waits on the way
    m.step(acts(p9=[MCMD_KICK, 40, 0]))
    assert m.get('set_play_taker')[0] == (10 | 0x100)
    for _ in range(12):
        m.step(acts(p10=[MCMD_KICK, 100, 0]))
        if m.get('mode')[0] != GM_PLAY_ON:
            break
    assert m.get('mode')[0] == GM_AFTER_GOAL and m.get('score_left')[0] == 1


def test_intentional_foul_brings_the_victim_down_and_may_be_carded():
    from soccer2d_amd._capi_match import CARD_RED, CARD_YELLOW, GM_FOUL_CHARGE, GM_FREE_KICK
    seen = {True: 0, False: 0}
    for seed in range(40):
        m = _foul_scene(seed=seed)
        m.step(acts(p5=[MCMD_TACKLE, 0, 1]))                # Tackle(power_or_dir = 0, foul = true)
        if m.get('tackle_cycles')[0][15] == 0:              # the tackle itself failed (exponent 10: rarely): no foul
            assert m.get('mode')[0] == GM_PLAY_ON and m.get('card')[0][5] == 0
            continue
        assert m.get('tackle_cycles')[0][15] == 5            # the victim stays down for foul_cycles (this cycle's tick included)
        called = m.get('mode')[0] == GM_FOUL_CHARGE
        seen[bool(called)] += 1
        if called:                                          # foul_charge_l + a yellow card; 30 stopped cycles; free kick for the victim's side
            assert m.get('mode_side')[0] == LEFT and m.get('card')[0][5] == CARD_YELLOW and m.get('vx')[0][22] == 0
            c0 = int(m.get('cycle')[0])
            for _ in range(30):
                m.step(acts())
            assert m.get('mode')[0] == GM_FREE_KICK and m.get('mode_side')[0] == RIGHT and m.get('cycle')[0] == c0
        else:                                               # the referee did not see it: play goes on, no card
            assert m.get('mode')[0] == GM_PLAY_ON and m.get('card')[0][5] == 0
    assert seen[True] >= 8 and seen[False] >= 8             # foul_detect_probability = 0.5
    # a clean tackle (foul = false) through the same opponent is no foul; nor is an intentional one with nobody on the ball
    m = _foul_scene(); m.step(acts(p5=[MCMD_TACKLE, 0, 0]))
    assert m.get('card')[0][5] == 0 and m.get('tackle_cycles')[0][15] == 0 and m.get('mode')[0] == GM_PLAY_ON
    m = _foul_scene(); m.set_obj(0, 15, x=20.0, y=20.0)
    m.step(acts(p5=[MCMD_TACKLE, 0, 1]))
    assert m.get('card')[0][5] == 0 and m.get('mode')[0] == GM_PLAY_ON
    # the second card is a red one: the player is parked beside the pitch, his commands are ignored, formations leave him there
    m = _foul_scene(foul_detect_probability=1.0, half_time_cycles=50, auto_reset=0)
    m.L.s2dmo_set_card(m.h, 0, 5, CARD_YELLOW)
    m.step(acts(p5=[MCMD_TACKLE, 0, 1]))
    assert m.get('card')[0][5] == CARD_RED and m.get('mode')[0] == GM_FOUL_CHARGE
    assert m.get('x')[0][5] == 0.0 and m.get('y')[0][5] == pytest.approx(-(34.0 + 6.0 + 1.5 * 5))
    for _ in range(100):
        m.step(acts(p5=[MCMD_DASH, 100, 0]))
    assert m.get('x')[0][5] == 0.0 and m.get('y')[0][5] == pytest.approx(-(34.0 + 6.0 + 1.5 * 5)) and m.get('cycle')[0] >= 50   # through half time


def test_foul_inside_the_own_penalty_area_is_a_penalty_kick():
    """PenaltyKick_ (idl/service.proto:278): FoulCharge_ called with the ball inside the offender's own penalty area -> after the
    announcement the other side restarts from the penalty spot of that half (11 m from the goal line), defenders cleared
    free_kick_distance from it; a kick by the taker's side puts the ball into play.  Outside the area: FreeKick_ where it happened."""
    from soccer2d_amd._capi_match import GM_FOUL_CHARGE, GM_FREE_KICK, GM_PENALTY_KICK
    seen = 0
    for seed in range(12):
        m = fresh(seed=seed, foul_detect_probability=1.0); play_on(m)
        m.set_obj(0, 5, x=-45.0, y=3.0, body=0.0)            # left #6 inside his own penalty area, facing +x
        m.set_obj(0, 15, x=-44.0, y=3.2, body=180.0)         # a right forward 1 m in front of him, the ball at his feet
        m.set_obj(0, 22, x=-44.3, y=3.1, vx=0.0, vy=0.0)
        m.set_obj(0, 2, x=-40.0, y=1.0)                      # a left defender who stands 1.8 m from the penalty spot
        m.step(acts(p5=[MCMD_TACKLE, 0, 1]))
        if m.get('mode')[0] != GM_FOUL_CHARGE:               # the tackle itself failed
            continue
        seen += 1
        assert m.get('mode_side')[0] == LEFT
        for _ in range(30):
            m.step(acts())
        assert m.get('mode')[0] == GM_PENALTY_KICK and m.get('mode_side')[0] == RIGHT
        assert m.get('x')[0][22] == -41.5 and m.get('y')[0][22] == 0.0 and m.get('vx')[0][22] == 0.0
        m.step(acts())                                       # the kept-away side (left) is cleared from the spot
        d = np.hypot(m.get('x')[0][:11] + 41.5, m.get('y')[0][:11])
        assert d.min() >= 9.15 - 1e-4
        m.set_obj(0, 15, x=-41.0, y=0.0, body=180.0)         # the fouled forward takes it
        m.step(acts(p15=[MCMD_KICK, 60, 0]))
        assert m.get('mode')[0] == GM_PLAY_ON and m.get('vx')[0][22] < 0
    assert seen >= 6
    # the same foul at midfield stays a free kick where it happened
    m = _foul_scene(foul_detect_probability=1.0)
    m.step(acts(p5=[MCMD_TACKLE, 0, 1]))
    if m.get('mode')[0] == GM_FOUL_CHARGE:
        for _ in range(30):
            m.step(acts())
        assert m.get("mode")[0] == GM_FREE_KICK and abs(m.get("x")[0][22]) < 8.0
