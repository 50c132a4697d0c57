"""League self-play on the 11v11 match engine (BASELINE.json configs[4]).

Every rank (one process per GPU) simulates its own batch of matches between policies drawn
from a shared league table; nothing is exchanged per cycle.  After a round the per-match
results -- and, when a learner wants them, the rollout buffers -- are all-gathered
(`torch.distributed`: RCCL over xGMI on GPUs, gloo in the CPU tests) and every rank applies
the same deterministic Elo update, so the table stays replicated without a parameter server.
"""
import math

import torch

from . import _capi_match as M


# ---------------------------------------------------------------- policies (device tensors)
def team_view(engine, side):
    """(x, y, body, ball_x, ball_y) of one team, mirrored so that the team always attacks +x."""
    sl = slice(0, 11) if side == 1 else slice(11, 22)
    sgn = 1.0 if side == 1 else -1.0
    x, y = engine.x[:, sl] * sgn, engine.y[:, sl] * sgn
    body = engine.body[:, sl] if side == 1 else torch.remainder(engine.body[:, sl] + 360.0, 360.0) - 180.0
    bx, by = engine.x[:, M.MATCH_BALL:M.MATCH_BALL + 1] * sgn, engine.y[:, M.MATCH_BALL:M.MATCH_BALL + 1] * sgn
    return x, y, body, bx, by


def _wrap(a):
    return torch.remainder(a + 180.0, 360.0) - 180.0


def _barred(engine, side):
    """bool[N,11]: the player who put the ball into play from the last set play and whom nobody else has touched the ball
    after -- he must not play it again (FreeKickFault_, idl/service.proto:287), so the scripted policies let a team-mate go."""
    taker = (engine.set_play_taker.view(-1, 1).long() & 0xff) - 1 - (0 if side == 1 else 11)   # index inside the team, or out of range (bit 8: the set play was an indirect free kick)
    in_play = (engine.mode == M.GM_PLAY_ON).view(-1, 1)
    return (torch.arange(11, device=engine.device).view(1, 11) == taker) & in_play


def chaser_policy(engine, side, kick_power=100.0):
    """Scripted baseline: the player nearest to the ball chases it (turn, then dash) and kicks it
    towards the opponent goal; the others hold position.  Returns actions float[N,11,3]."""
    x, y, body, bx, by = team_view(engine, side)
    n = x.shape[0]
    act = torch.zeros((n, 11, 3), device=x.device)
    dx, dy = bx - x, by - y
    dist = torch.hypot(dx, dy)
    to_ball = _wrap(torch.rad2deg(torch.atan2(dy, dx)) - body)
    to_goal = _wrap(torch.rad2deg(torch.atan2(-y, 52.5 - x)) - body)
    barred = _barred(engine, side)
    nearest = torch.where(barred, torch.full_like(dist, 1e9), dist).argmin(dim=1, keepdim=True)
    is_chaser = torch.zeros_like(dist, dtype=torch.bool).scatter_(1, nearest, True)
    kickable = (dist <= 1.0) & ~barred
    turn = is_chaser & ~kickable & (to_ball.abs() > 15.0)
    dash = is_chaser & ~kickable & ~turn
    kick = kickable
    act[..., 0] = torch.where(kick, 3.0, torch.where(turn, 2.0, torch.where(dash, 1.0, 0.0)))
    act[..., 1] = torch.where(kick, torch.full_like(dist, kick_power), torch.where(turn, to_ball, torch.where(dash, 100.0, 0.0)))
    act[..., 2] = torch.where(kick, to_goal, torch.zeros_like(dist))
    return act


def idle_policy(engine, side):
    return torch.zeros((engine.num_envs, 11, 3), device=engine.device)


def random_policy(seed=0):
    g = {}

    def pol(engine, side):
        gen = g.setdefault(engine.device, torch.Generator(device=engine.device).manual_seed(seed + side))
        n = engine.num_envs
        act = torch.zeros((n, 11, 3), device=engine.device)
        act[..., 0] = torch.randint(1, 5, (n, 11), device=engine.device, generator=gen).float()
        act[..., 1] = torch.rand((n, 11), device=engine.device, generator=gen) * 200.0 - 100.0
        act[..., 2] = torch.rand((n, 11), device=engine.device, generator=gen) * 360.0 - 180.0
        return act
    return pol


# ---------------------------------------------------------------- league bookkeeping
class League:
    """Replicated Elo table over `n_policies`; deterministic pairing per (round, global match id)."""

    def __init__(self, n_policies, k=16.0, seed=0):
        self.n = int(n_policies)
        self.k = float(k)
        self.seed = int(seed)
        self.elo = torch.full((self.n,), 1000.0, dtype=torch.float64)
        self.games = torch.zeros(self.n, dtype=torch.int64)

    def pairing(self, round_idx, first_match, n_matches):
        """Policy ids (left, right) of global matches [first_match, first_match + n_matches)."""
        ids = torch.arange(first_match, first_match + n_matches, dtype=torch.int64)
        h = (ids * 2654435761 + (round_idx + 1) * 40503 + self.seed * 97) % 2147483647
        left = h % self.n
        right = (left + 1 + (h // self.n) % max(1, self.n - 1)) % self.n
        return left, right

    def update(self, left, right, goals_left, goals_right):
        """Sequential Elo update in global match order (identical on every rank)."""
        for l, r, gl, gr in zip(left.tolist(), right.tolist(), goals_left.tolist(), goals_right.tolist()):
            ea = 1.0 / (1.0 + math.pow(10.0, (float(self.elo[r]) - float(self.elo[l])) / 400.0))
            sa = 1.0 if gl > gr else (0.0 if gl < gr else 0.5)
            d = self.k * (sa - ea)
            self.elo[l] += d
            self.elo[r] -= d
            self.games[l] += 1
            self.games[r] += 1


def play_round(engine, policies, left_ids, right_ids, n_cycles):
    """Simulate `n_cycles` of every match of `engine`; match i is policies[left_ids[i]] (left) vs
    policies[right_ids[i]] (right).  Returns (goals_left, goals_right) int64[N] for the round."""
    engine.reset()
    n, dev = engine.num_envs, engine.device
    left_ids, right_ids = left_ids.to(dev), right_ids.to(dev)
    s0l, s0r = engine.score_left.clone(), engine.score_right.clone()
    act = torch.zeros((n, 22, 3), device=dev)
    for _ in range(n_cycles):
        for pid, pol in enumerate(policies):
            ml = (left_ids == pid).view(n, 1, 1)
            mr = (right_ids == pid).view(n, 1, 1)
            if bool(ml.any()):
                act[:, :11] = torch.where(ml, pol(engine, 1), act[:, :11])
            if bool(mr.any()):
                act[:, 11:] = torch.where(mr, pol(engine, 2), act[:, 11:])
        engine.step(act)
    return (engine.score_left - s0l).to(torch.int64), (engine.score_right - s0r).to(torch.int64)


def exchange_results(left, right, goals_left, goals_right, group=None):
    """All-gather one round's results so that every rank can apply the same Elo update.
    Inputs are per-rank tensors of equal length; output tensors are ordered by rank."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return left, right, goals_left, goals_right
    world = dist.get_world_size(group)
    packed = torch.stack([left, right, goals_left, goals_right]).to(torch.int64).contiguous()
    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out.view(-1), packed.view(-1), group=group)
    out = out.permute(1, 0, 2).reshape(4, -1)
    return out[0], out[1], out[2], out[3]


# ---------------------------------------------------------------- a learner: population-based search
class ParamChaser:
    """Chaser policy with four learnable numbers (a league member's genome):
    kick_power [0,100], aim_y (where on the goal line it shoots, metres), turn_tol (degrees of
    misalignment it accepts before dashing), n_chasers (how many nearest players go for the ball)."""
    LOW = torch.tensor([5.0, -30.0, 2.0, 1.0])
    HIGH = torch.tensor([100.0, 30.0, 90.0, 4.0])

    def __init__(self, theta):
        self.theta = torch.as_tensor(theta, dtype=torch.float32).clone()

    def __call__(self, engine, side):
        kp, aim_y, tol, nch = (float(v) for v in self.theta)
        x, y, body, bx, by = team_view(engine, side)
        act = torch.zeros((x.shape[0], 11, 3), device=x.device)
        dx, dy = bx - x, by - y
        dist = torch.hypot(dx, dy)
        to_ball = _wrap(torch.rad2deg(torch.atan2(dy, dx)) - body)
        to_goal = _wrap(torch.rad2deg(torch.atan2(aim_y - y, 52.5 - x)) - body)
        barred = _barred(engine, side)
        rank = torch.where(barred, torch.full_like(dist, 1e9), dist).argsort(dim=1).argsort(dim=1)
        is_chaser = (rank < int(round(nch))) & ~barred
        kickable = (dist <= 1.0) & ~barred
        turn = is_chaser & ~kickable & (to_ball.abs() > tol)
        dash = is_chaser & ~kickable & ~turn
        act[..., 0] = torch.where(kickable, 3.0, torch.where(turn, 2.0, torch.where(dash, 1.0, 0.0)))
        act[..., 1] = torch.where(kickable, torch.full_like(dist, kp), torch.where(turn, to_ball, torch.where(dash, 100.0, 0.0)))
        act[..., 2] = torch.where(kickable, to_goal, torch.zeros_like(dist))
        return act

    def mutated(self, gen, scale=0.15):
        span = self.HIGH - self.LOW
        noise = torch.randn(4, generator=gen) * scale * span
        return ParamChaser(torch.minimum(torch.maximum(self.theta + noise, self.LOW), self.HIGH))


def evolve_league(engine, population, rounds, n_cycles, seed=0, first_match=0, total_matches=None, group=None):
    """Population-based training on the league: each round every rank plays its shard of the matches
    (pairing by global match id), results are all-gathered, the replicated Elo table is updated, and the
    weakest member is replaced by a mutation of the strongest -- drawn from a generator seeded by
    (seed, round), so every rank makes the same replacement without exchanging parameters.
    Returns (league, population, history of (round, champion id, champion theta))."""
    league = League(len(population), seed=seed)
    total = total_matches if total_matches is not None else engine.num_envs
    history = []
    for r in range(rounds):
        left, right = league.pairing(r, first_match, engine.num_envs)
        gl, gr = play_round(engine, population, left, right, n_cycles)
        L, R, GL, GR = exchange_results(left.to(engine.device), right.to(engine.device), gl, gr, group=group)
        league.update(L.cpu(), R.cpu(), GL.cpu(), GR.cpu())
        assert L.numel() >= engine.num_envs and (total is None or L.numel() <= max(total, engine.num_envs))
        best, worst = int(league.elo.argmax()), int(league.elo.argmin())
        gen = torch.Generator().manual_seed(seed * 1000003 + r)
        if best != worst:
            population[worst] = population[best].mutated(gen)
            league.elo[worst] = league.elo.mean()          # the newcomer starts from the table's mean
        history.append((r, best, population[best].theta.clone()))
    return league, population, history
