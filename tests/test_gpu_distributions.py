"""Distributional tests of the engine's Philox streams at N = 65 536 ("noise mode = Philox with distributional
tests").  The reference draws from unseeded `random` / `np.random`, so sequences cannot be matched -- distributions can:

  * reset grids (reach_ball_env.py:173-181): chi-square of player x / y / body and ball x / y against the uniform integer grids
    randint produces (body 0 and 360 fold onto the same angle);
  * rejection-sampled ball velocity (:202-212): two-sample chi-square of the speed and direction marginals against
    tests/golden/reset_dist.json -- 120 000 runs of the REFERENCE's own trainer_reset_actions (tests/golden/make_golden.py);
  * the in-engine uniform policy: chi-square over the 16 actions;
  * `_inc` noise of the ball: magnitude / (ball_rand * |v|) uniform on [0, 1), direction uniform, never beyond the bound.

Thresholds: every statistic must stay below the chi-square quantile at 1 - 1e-6 of its degrees of freedom (a correct sampler
fails such a test once in a million seeds; the seed is fixed, so the tests are deterministic)."""
import json
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
chi2 = pytest.importorskip('scipy.stats').chi2

N = 65536
P_TAIL = 1e-6
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reset_dist.json')


def _engine(**kw):
    from soccer2d_amd.engine import Engine, make_config
    kw.setdefault('noise', False)
    return Engine(N, 'cuda:0', cfg=make_config(**kw))


def chi_square_uniform(counts, probs=None):
    counts = np.asarray(counts, dtype=np.float64)
    n = counts.sum()
    exp = n / len(counts) if probs is None else n * np.asarray(probs)
    return float(((counts - exp) ** 2 / exp).sum()), len(counts) - 1


def chi_square_two_sample(a, b):
    """Homogeneity test of two histograms over the same bins."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    keep = (a + b) > 0
    a, b = a[keep], b[keep]
    k1, k2 = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    return float((((k1 * a - k2 * b) ** 2) / (a + b)).sum()), int(keep.sum()) - 1


def check(stat_df, what):
    stat, df = stat_df
    limit = float(chi2.ppf(1.0 - P_TAIL, df))
    assert stat < limit, f'{what}: chi2 = {stat:.1f} with {df} degrees of freedom exceeds the 1 - {P_TAIL} quantile {limit:.1f}'
    return stat, limit


def test_reset_grids_are_uniform():
    # ball at rest (change_ball_velocity False): the command-less cycle of a reset leaves the ball on its grid point
    eng = _engine(use_continuous_action=False, change_ball_position=True, change_ball_velocity=False)
    hist = {k: np.zeros(m, np.int64) for k, m in (('px', 101), ('py', 61), ('bx', 101), ('by', 61), ('body', 360))}
    collided = 0
    for _ in range(4):                                     # 4 x 65 536 episodes per env id: different episode indices
        eng.reset()
        torch.cuda.synchronize()
        planes = {key: plane.cpu().numpy() for key, plane in (('px', eng.player_x), ('py', eng.player_y), ('bx', eng.ball_x), ('by', eng.ball_y))}
        # player and ball drawn onto the same grid point (1 in 6 161) collide in the reset's cycle and leave the grid
        on_grid = np.ones(N, bool)
        for v in planes.values():
            on_grid &= v == np.rint(v)
        collided += int((~on_grid).sum())
        for key, off in (('px', 50), ('py', 30), ('bx', 50), ('by', 30)):
            hist[key] += np.bincount((planes[key][on_grid] + off).astype(np.int64), minlength=len(hist[key]))
        b = eng.player_body.cpu().numpy()
        assert np.array_equal(b, np.rint(b)) and b.min() > -180.0 and b.max() <= 180.0
        hist['body'] += np.bincount((b + 179).astype(np.int64), minlength=360)       # -179 .. 180
    assert collided < 4 * N * 4 / 6161, collided           # expected 4 N / 6 161 = 43
    for key in ('px', 'py', 'bx', 'by'):
        check(chi_square_uniform(hist[key]), f'reset {key}')
    p_body = np.full(360, 1.0 / 361.0)
    p_body[179] = 2.0 / 361.0                              # randint(0, 360): 0 and 360 are the same angle
    check(chi_square_uniform(hist['body'], p_body), 'reset body')


def test_ball_velocity_matches_the_reference_sampler():
    ref = json.load(open(GOLD))
    assert ref['runs'] >= 100000
    eng = _engine(use_continuous_action=False, change_ball_position=True, change_ball_velocity=True, max_steps=200)
    decay = 0.94                                           # the reset's command-less cycle has decayed the velocity once
    speed_h, dir_h = np.zeros(30, np.int64), np.zeros(36, np.int64)
    for _ in range(3):
        eng.reset()
        torch.cuda.synchronize()
        vx, vy = eng.ball_vx.cpu().numpy().astype(np.float64) / decay, eng.ball_vy.cpu().numpy().astype(np.float64) / decay
        sp = np.hypot(vx, vy)
        assert sp.max() < 3.0 + 1e-5
        speed_h += np.bincount(np.minimum(29, (sp / 0.1).astype(np.int64)), minlength=30)
        moving = sp > 0
        deg = np.rint(np.degrees(np.arctan2(vy[moving], vx[moving]))).astype(np.int64) % 360
        dir_h += np.bincount(((deg + 5) % 360) // 10, minlength=36)
    # the rejection loop shapes both marginals (fast balls towards a near line are rejected): plain uniforms would fail
    s_uni, _ = chi_square_uniform(speed_h)
    assert s_uni > 10 * chi2.ppf(1 - P_TAIL, 29), 'the speed marginal looks uniform: is the rejection test running?'
    check(chi_square_two_sample(speed_h, ref['hist']['speed']), 'ball speed marginal vs the reference sampler')
    check(chi_square_two_sample(dir_h, ref['hist']['dir']), 'ball direction marginal vs the reference sampler')
    # acceptance rate: the reference accepts the first candidate with probability tries[0] / runs; the CPU oracle (bit-equal
    # to the device) reports its number of tries
    tries = np.array(ref['hist']['tries'], np.float64)
    p_first = tries[0] / tries.sum()
    orc = O.OracleEngine(O.make_config(noise=0, use_continuous_action=False, change_ball_velocity=True), 20000, 'f32')
    t = orc.reset_tries()
    se = np.sqrt(p_first * (1 - p_first) * (1 / len(t) + 1 / tries.sum()))
    assert abs((t == 1).mean() - p_first) < 5 * se, ((t == 1).mean(), p_first)
    check(chi_square_two_sample(np.bincount(np.minimum(15, t - 1), minlength=16), tries), 'number of velocity candidates')


def test_in_engine_policy_is_uniform_over_the_actions():
    eng = _engine(use_continuous_action=False, action_space_size=16, change_ball_velocity=True)
    eng.reset()
    out = eng.rollout(64, with_obs=False)
    torch.cuda.synchronize()
    counts = torch.bincount(out['action'].reshape(-1).long(), minlength=16).cpu().numpy()
    assert counts.sum() == 64 * N and len(counts) == 16
    check(chi_square_uniform(counts), 'policy actions')
    # and per step, so that a defect in one word of the four-step Philox block cannot hide in the total
    for t in (0, 1, 2, 3, 63):
        check(chi_square_uniform(torch.bincount(out['action'][t].long(), minlength=16).cpu().numpy()), f'policy actions at step {t}')


def test_ball_noise_magnitude_and_direction():
    eng = _engine(use_continuous_action=False, change_ball_velocity=True, noise=True, min_distance_to_ball=0.0)
    eng.reset()
    for _ in range(3):
        eng.step(None)
    mag_h, dir_h, seen = np.zeros(20, np.int64), np.zeros(36, np.int64), 0
    rand, decay = 0.05, 0.94                               # ball_rand, ball_decay (stock rcssserver values of s2d_default_config)
    for _ in range(6):
        torch.cuda.synchronize()
        v0 = np.stack([eng.ball_vx.cpu().numpy(), eng.ball_vy.cpu().numpy()]).astype(np.float64)
        p0 = np.stack([eng.ball_x.cpu().numpy(), eng.ball_y.cpu().numpy()]).astype(np.float64)
        ep0 = eng.episode.cpu().numpy()
        eng.step(None)
        torch.cuda.synchronize()
        v1 = np.stack([eng.ball_vx.cpu().numpy(), eng.ball_vy.cpu().numpy()]).astype(np.float64) / decay
        p1 = np.stack([eng.ball_x.cpu().numpy(), eng.ball_y.cpu().numpy()]).astype(np.float64)
        s0 = np.hypot(*v0)
        # same episode, ball fast enough for fp32 to resolve the noise, and no collision (position advanced by the noisy velocity)
        ok = (eng.episode.cpu().numpy() == ep0) & (s0 > 0.3) & (np.abs(p1 - p0 - v1).max(axis=0) < 1e-4)
        noise = (v1 - v0)[:, ok]
        bound = rand * s0[ok]
        m = np.hypot(*noise) / bound
        assert m.max() < 1.0 + 1e-3, m.max()               # magnitude in [0, rand * |v|)
        mag_h += np.bincount(np.minimum(19, (m * 20).astype(np.int64)), minlength=20)
        big = m > 0.05                                     # direction of a vanishing noise vector is rounding
        dir_h += np.bincount(((np.degrees(np.arctan2(noise[1][big], noise[0][big])) + 180.0) // 10).astype(np.int64) % 36, minlength=36)
        seen += int(ok.sum())
    assert seen > 100000
    check(chi_square_uniform(mag_h), 'ball noise magnitude / (ball_rand * |v|)')
    check(chi_square_uniform(dir_h), 'ball noise direction')


def test_noise_draw_lattice():
    """The movement-noise draw as the spec defines it (round-3 respecification, parity UNPINNED against
    rcssserver's continuous drand pair): ONE Philox word per object and cycle -- magnitude uniform = (word >> 16) / 65536, direction
    = the WHOLE degree ((word & 0xffff) * 360) >> 16 - 180, whose sine / cosine are sincos_deg of that degree; one block per two
    cycles (words x, y for even k, z, w for odd k of block 0 at counter k >> 1, stream 3).  The device's draw (s2d_debug_eval op 10)
    equals that restatement bit for bit -- so a later coarsening of the lattice cannot pass -- and the 360 direction cells and the
    64 leading magnitude cells are uniform."""
    from soccer2d_amd import _capi
    lib = _capi.load_library()
    n, seed = 1 << 16, 0x5EED
    gid = np.arange(n, dtype=np.uint32) * 3 + 11
    dir_h, mag_h = np.zeros(360, np.int64), np.zeros(64, np.int64)
    for k in (0, 1, 6, 7, 1001):
        x = torch.from_numpy(np.stack([gid, np.full(n, 7, np.uint32), np.full(n, k, np.uint32), np.full(n, seed, np.uint32)], axis=1).astype(np.int64)).to(torch.int32).cuda().contiguous()
        y = torch.empty((n, 6), dtype=torch.float32, device='cuda:0')
        _capi.check(lib, lib.s2d_debug_eval(10, x.data_ptr(), y.data_ptr(), n, None), 's2d_debug_eval')
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        sub = np.arange(0, n, 97)                          # the CPU restatement on a subset (ctypes Philox per draw)
        for i in sub:
            w = O.philox([int(gid[i]), 7, k >> 1, (3 << 16) | 0], [seed, 0])
            wp, wb = (w[2], w[3]) if (k & 1) else (w[0], w[1])
            exp = []
            for word in (wp, wb):
                s, c = O.sincos_deg(float((((word & 0xffff) * 360) >> 16) - 180))
                exp += [np.float32((word >> 16) * 2.0 ** -16), np.float32(s), np.float32(c)]
            assert np.array_equal(got[i].view(np.int32), np.array(exp, np.float32).view(np.int32)), (k, i, got[i], exp)
        for m, sn, cs in ((got[:, 0], got[:, 1], got[:, 2]), (got[:, 3], got[:, 4], got[:, 5])):
            assert np.array_equal(m * 65536.0, np.rint(m * 65536.0)) and m.min() >= 0.0 and m.max() < 1.0      # k / 65536
            deg = np.degrees(np.arctan2(sn.astype(np.float64), cs.astype(np.float64)))
            assert np.abs(deg - np.rint(deg)).max() < 1e-4                                                      # whole degrees
            dir_h += np.bincount(np.rint(deg).astype(np.int64) % 360, minlength=360)
            mag_h += np.bincount((m * 64).astype(np.int64), minlength=64)
    check(chi_square_uniform(dir_h), 'noise direction over the 360 whole degrees')
    check(chi_square_uniform(mag_h), 'noise magnitude uniform')
