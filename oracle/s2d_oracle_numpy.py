"""NumPy-vectorised restatement of the reach_ball step -- the "what a Python user could do on the
CPU" baseline of SURVEY.md 8(d), row CPU-2.

TEST / BENCH INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by
bench.py's cpu_baseline leg, never by the product.  It follows the same algorithm as
oracle/s2d_oracle.c (float64, libm), written independently as whole-array NumPy expressions:

  A2 action map          reach_ball_env.py:53-85   (discrete and 1-D continuous: always Dash(100, dir))
  A3 observation         reach_ball_env.py:87-111
  A4 reward/done/result  reach_ball_env.py:113-161
  A5/A6 reset            reach_ball_env.py:170-218, soccer_2d_env.py:179-206
  S  one rcssserver cycle (EXT): dash, stamina, integrate, collision
  randomness             Philox4x32-10 with the counter layout of DESIGN.md section 5

Not covered (the C oracle is the checker for those): the 4-D turning action space, noise.
It is pinned by tests/test_oracle_numpy.py against the float64 build of the C oracle.
"""
import numpy as np

U32 = np.uint32
U64 = np.uint64
M32 = U64(0xFFFFFFFF)
ST_RESET, ST_POLICY = 0, 1


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; counter words are uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=U64) & M32 for c in (c0, c1, c2, c3))
    k0, k1 = U64(k0), U64(k1)
    for _ in range(10):
        p0 = U64(0xD2511F53) * c0
        p1 = U64(0xCD9E8D57) * c2
        n0 = (p1 >> U64(32)) ^ c1 ^ k0
        n2 = (p0 >> U64(32)) ^ c3 ^ k1
        c0, c1, c2, c3 = n0, p1 & M32, n2, p0 & M32
        k0 = (k0 + U64(0x9E3779B9)) & M32
        k1 = (k1 + U64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def rnd_below(w, span):
    return ((w * U64(span)) >> U64(32)).astype(np.int64)


def rnd_u01(w):
    return (w >> U64(8)).astype(np.float64) * 5.9604644775390625e-8


def norm_deg(d):
    d = np.where((d < -360.0) | (d > 360.0), np.fmod(d, 360.0), d)
    d = np.where(d < -180.0, d + 360.0, d)
    return np.where(d > 180.0, d - 360.0, d)


def atan2_deg(y, x):
    return np.where((x == 0.0) & (y == 0.0), 0.0, np.degrees(np.arctan2(y, x)))


class NumpyReachBall:
    """n reach_ball envs advanced in lockstep with whole-array NumPy operations."""

    def __init__(self, n, server, task, seed=0x5EED, env_id_offset=0, auto_reset=True):
        if task.get('use_continuous_action', True) and task.get('use_turning', False):
            raise NotImplementedError("turning action space: use the C oracle")
        self.n, self.sp, self.tk = int(n), dict(server), dict(task)
        self.seed_lo, self.seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        gid = np.arange(n, dtype=np.uint64) + np.uint64(env_id_offset)
        self.gid_lo, self.gid_hi = gid & M32, gid >> U64(32)
        self.auto_reset = bool(auto_reset)
        z = lambda: np.zeros(n)
        self.px, self.py, self.vx, self.vy, self.body = z(), z(), z(), z(), z()
        self.bx, self.by, self.bvx, self.bvy = z(), z(), z(), z()
        self.prev_dist, self.prev_angle = z(), z()
        self.stamina = np.full(n, float(server['stamina_max']))
        self.recovery = np.full(n, float(server['recover_init']))
        self.effort = np.full(n, float(server['effort_init']))
        self.capacity = np.full(n, float(server['stamina_capacity']))
        self.step_number = np.zeros(n, dtype=np.int64)
        self.cycle = np.zeros(n, dtype=np.int64)
        self.policy_step = np.zeros(n, dtype=np.uint64)
        self.episode = np.zeros(n, dtype=np.int64)
        self.obs = np.zeros((n, 10))
        self.terminal_obs = np.zeros((n, 10))
        self.reward, self.done, self.result = z(), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        self.stats = np.zeros(4, dtype=np.int64)
        d = float(task.get('reset_ball_decay', 0.96))
        self.travel_factor = (1.0 - d ** int(task['max_steps'])) / (1.0 - d)

    # ------------------------------------------------------------------ randomness
    def _draw(self, idx, c, stream, block):
        c = (np.asarray(c, dtype=np.int64) & 0xFFFFFFFF).astype(U64)   # counter word = low 32 bits (two's complement)
        return philox4x32_10(self.gid_lo[idx], self.gid_hi[idx], c,
                             np.full(len(idx), (stream << 16) | block, dtype=U64), self.seed_lo, self.seed_hi)

    # ------------------------------------------------------------------ A5: reset sample
    def _reset_sample(self, idx):
        tk, sp = self.tk, self.sp
        self.episode[idx] += 1
        c0 = self.episode[idx]                                    # index of the episode the reset starts
        w = self._draw(idx, c0, ST_RESET, 0)
        w1 = self._draw(idx, c0, ST_RESET, 1)
        px = (-50 + rnd_below(w[0], 101)).astype(np.float64)
        py = (-30 + rnd_below(w[1], 61)).astype(np.float64)
        body = norm_deg(rnd_below(w[2], 361).astype(np.float64))
        m = len(idx)
        if tk['change_ball_position']:
            bx = (-50 + rnd_below(w[3], 101)).astype(np.float64)
            by = (-30 + rnd_below(w1[0], 61)).astype(np.float64)
        else:
            bx = np.full(m, float(tk['ball_position_x'])); by = np.full(m, float(tk['ball_position_y']))
        bvx, bvy = np.zeros(m), np.zeros(m)
        if tk['change_ball_velocity']:
            todo = np.ones(m, dtype=bool)
            ws, wd = w1[1], w1[2]
            wb = None
            for k in range(255):
                if not todo.any():
                    break
                if k >= 1:
                    if (k - 1) & 1:
                        ws, wd = wb[2], wb[3]
                    else:
                        wb = self._draw(idx, c0, ST_RESET, 2 + ((k - 1) >> 1)); ws, wd = wb[0], wb[1]
                speed = rnd_u01(ws) * 3.0
                ang = np.radians(rnd_below(wd, 361).astype(np.float64))
                cs, sn = np.cos(ang), np.sin(ang)
                travel = speed * self.travel_factor
                ok = (np.abs(bx + travel * cs) <= sp['pitch_half_length']) & (np.abs(by + travel * sn) <= sp['pitch_half_width'])
                take = todo & ok
                bvx = np.where(take, speed * cs, bvx); bvy = np.where(take, speed * sn, bvy)
                todo &= ~ok
        else:
            ang = np.radians(float(tk['ball_direction']))
            bvx[:] = float(tk['ball_speed']) * np.cos(ang); bvy[:] = float(tk['ball_speed']) * np.sin(ang)
        return px, py, body, bx, by, bvx, bvy

    # ------------------------------------------------------------------ S: one cycle
    def _cycle(self, idx, ax, ay):
        """integrate + collision + decay + stamina for the envs in idx (ax/ay None = no command)"""
        sp = self.sp
        vx, vy, px, py = self.vx[idx], self.vy[idx], self.px[idx], self.py[idx]
        bvx, bvy, bx, by = self.bvx[idx], self.bvy[idx], self.bx[idx], self.by[idx]
        def clamp_scale(m2, vmax):                                  # 1 where |v| <= vmax, else vmax / |v|
            return np.where(m2 > vmax * vmax, vmax / np.sqrt(np.where(m2 > 0.0, m2, 1.0)), 1.0)
        if ax is not None:
            k = clamp_scale(ax * ax + ay * ay, sp['player_accel_max'])
            vx = vx + ax * k; vy = vy + ay * k
        k = clamp_scale(vx * vx + vy * vy, sp['player_speed_max'])
        vx, vy = vx * k, vy * k
        px = px + vx; py = py + vy
        k = clamp_scale(bvx * bvx + bvy * bvy, sp['ball_speed_max'])
        bvx, bvy = bvx * k, bvy * k
        bx = bx + bvx; by = by + bvy
        rsum = sp['player_size'] + sp['ball_size']
        dx, dy = bx - px, by - py
        d2 = dx * dx + dy * dy
        hit = d2 < rsum * rsum
        if hit.any():
            d = np.sqrt(d2)
            ux = np.where(d > 0.0, dx / np.where(d > 0.0, d, 1.0), 1.0); uy = np.where(d > 0.0, dy / np.where(d > 0.0, d, 1.0), 0.0)
            mx, my, h, cv = (px + bx) * 0.5, (py + by) * 0.5, rsum * 0.5, sp['collision_vel_rate']
            px = np.where(hit, mx - ux * h, px); py = np.where(hit, my - uy * h, py)
            bx = np.where(hit, mx + ux * h, bx); by = np.where(hit, my + uy * h, by)
            vx = np.where(hit, vx * cv, vx); vy = np.where(hit, vy * cv, vy)
            bvx = np.where(hit, bvx * cv, bvx); bvy = np.where(hit, bvy * cv, bvy)
        self.cycle[idx] += 1
        self.vx[idx], self.vy[idx] = vx * sp['player_decay'], vy * sp['player_decay']
        self.bvx[idx], self.bvy[idx] = bvx * sp['ball_decay'], bvy * sp['ball_decay']
        self.px[idx], self.py[idx], self.bx[idx], self.by[idx] = px, py, bx, by
        # Player::updateStamina
        st, rec, eff, cap = self.stamina[idx], self.recovery[idx], self.effort[idx], self.capacity[idx]
        smax = sp['stamina_max']
        rec = np.where((st <= sp['recover_dec_thr'] * smax) & (rec > sp['recover_min']),
                       np.maximum(rec - sp['recover_dec'], sp['recover_min']), rec)
        eff = np.where((st <= sp['effort_dec_thr'] * smax) & (eff > sp['effort_min']),
                       np.maximum(eff - sp['effort_dec'], sp['effort_min']), eff)
        eff = np.where((st >= sp['effort_inc_thr'] * smax) & (eff < sp['effort_init']),
                       np.minimum(eff + sp['effort_inc'], sp['effort_init']), eff)
        inc = np.minimum(rec * sp['stamina_inc_max'], smax - st)
        capped = sp['stamina_capacity'] >= 0.0
        if capped:
            inc = np.minimum(inc, cap)
            cap = np.maximum(cap - inc, 0.0)
        self.stamina[idx] = np.minimum(st + inc, smax)
        self.recovery[idx], self.effort[idx], self.capacity[idx] = rec, eff, cap

    # ------------------------------------------------------------------ A3 + A4
    def _observe(self, idx):
        sp = self.sp
        px, py, body = self.px[idx], self.py[idx], self.body[idx]
        bx, by, bvx, bvy = self.bx[idx], self.by[idx], self.bvx[idx], self.bvy[idx]
        rel = norm_deg(atan2_deg(by - py, bx - px) - body)
        o = np.stack([rel / 180.0, body / 180.0, px / sp['pitch_half_length'], py / sp['pitch_half_width'],
                      bx / sp['pitch_half_length'], by / sp['pitch_half_width'], np.sqrt(bvx * bvx + bvy * bvy) / 3.0,
                      atan2_deg(bvy, bvx) / 360.0, bvx / 3.0, bvy / 3.0], axis=1)
        dist = np.sqrt((bx - px) ** 2 + (by - py) ** 2)
        return o, dist, rel

    def _seed_carry(self, idx):
        o, dist, rel = self._observe(idx)
        self.obs[idx] = o
        self.prev_dist[idx], self.prev_angle[idx] = dist, rel

    # ------------------------------------------------------------------ public API
    def reset(self, mask=None):
        idx = np.arange(self.n) if mask is None else np.nonzero(np.asarray(mask))[0]
        if len(idx) == 0:
            return self.obs
        self._apply_reset(idx)
        self.reward[idx] = 0.0; self.done[idx] = 0; self.result[idx] = 0
        return self.obs

    def _apply_reset(self, idx):
        sp = self.sp
        px, py, body, bx, by, bvx, bvy = self._reset_sample(idx)
        self.step_number[idx] = 0
        self.px[idx], self.py[idx], self.body[idx] = px, py, body
        self.vx[idx] = 0.0; self.vy[idx] = 0.0
        self.bx[idx], self.by[idx], self.bvx[idx], self.bvy[idx] = bx, by, bvx, bvy
        self.stamina[idx] = sp['stamina_max']; self.recovery[idx] = sp['recover_init']
        self.effort[idx] = sp['effort_init']; self.capacity[idx] = sp['stamina_capacity']
        self._cycle(idx, None, None)                               # soccer_2d_env.py:186-197
        self._seed_carry(idx)

    def step(self, actions=None):
        sp, tk, n = self.sp, self.tk, self.n
        allidx = np.arange(n)
        if actions is None:                                        # in-engine random policy
            k = self.policy_step
            q = philox4x32_10(self.gid_lo, self.gid_hi, k >> U64(2), np.full(n, ST_POLICY << 16, dtype=U64),
                              self.seed_lo, self.seed_hi)
            j = (k & U64(3)).astype(np.int64)
            w = np.choose(j, q)
            a = rnd_below(w, tk['action_space_size']).astype(np.float64) if not tk['use_continuous_action'] \
                else rnd_u01(w) * 2.0 - 1.0
            self.policy_step = (k + U64(1)) & M32
        else:
            a = np.asarray(actions, dtype=np.float64).reshape(n)
        self.last_action = a
        if not tk['use_continuous_action']:                        # reach_ball_env.py:84
            dirn = np.mod(a * 360.0 / tk['action_space_size'], 360.0) - 180.0
        else:
            dirn = a * 180.0                                       # :81-82, not clipped
        self.step_number += 1
        # Player::dash(100, dir)
        power = np.clip(100.0, sp['min_dash_power'], sp['max_dash_power'])
        dirn = np.clip(dirn, sp['min_dash_angle'], sp['max_dash_angle'])
        if sp['dash_angle_step'] > 0.0:
            dirn = sp['dash_angle_step'] * np.rint(dirn / sp['dash_angle_step'])
        back = power < 0.0
        need = np.minimum(np.full(n, power * -2.0 if back else power), self.stamina + sp['extra_stamina'])
        self.stamina = np.maximum(self.stamina - need, 0.0)
        eff_power = need / -2.0 if back else need
        ad = np.abs(dirn)
        rate = np.where(ad > 90.0,
                        sp['back_dash_rate'] - ((sp['back_dash_rate'] - sp['side_dash_rate']) * (1.0 - (ad - 90.0) / 90.0)),
                        sp['side_dash_rate'] + ((1.0 - sp['side_dash_rate']) * (1.0 - ad / 90.0)))
        rate = np.clip(rate, 0.0, 1.0)
        acc = np.abs(self.effort * eff_power * rate * sp['dash_power_rate'])
        ang = np.radians(norm_deg(self.body + (dirn + 180.0 if back else dirn)))
        self._cycle(allidx, acc * np.cos(ang), acc * np.sin(ang))
        # observation + reward / done / result
        o, dist, rel = self._observe(allidx)
        r = (self.prev_dist - dist) + (np.abs(self.prev_angle) - np.abs(rel)) / 180.0
        goal = dist < tk['min_distance_to_ball']
        out = (np.abs(self.px) > sp['pitch_half_length']) | (np.abs(self.py) > sp['pitch_half_width'])
        tmo = self.step_number > tk['max_steps']
        r = r + np.where(goal, 10.0, 0.0) - np.where(out, -10.0, 0.0) - np.where(tmo, 5.0, 0.0)
        res = np.where(tmo, 3, np.where(out, 2, np.where(goal, 1, 0))).astype(np.uint8)
        self.prev_dist, self.prev_angle = dist, rel
        self.obs, self.reward, self.result = o, r, res
        self.done = (res != 0).astype(np.uint8)
        self.stats += np.array([n, (res == 1).sum(), (res == 2).sum(), (res == 3).sum()])
        if self.auto_reset and self.done.any():
            idx = np.nonzero(self.done)[0]
            self.terminal_obs[idx] = o[idx]
            self._apply_reset(idx)
        return self.obs, self.reward, self.done, self.result
