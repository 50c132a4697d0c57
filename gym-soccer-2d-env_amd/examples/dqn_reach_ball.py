#!/usr/bin/env python3
"""DQN on N vectorised reach_ball envs, everything on the GPU.

Mirror of the reference's dqn_stable_baselines3.py (same env kwargs :18-31, same train / test
structure :40-71, same Goal/Out/Timeout bookkeeping) with stable-baselines3 -- which cannot be
installed offline -- replaced by a ~100-line DQN that consumes the engine's DEVICE tensors
directly: observations never leave HBM, the replay buffer is a device tensor, one vector step
feeds N transitions.

    python examples/dqn_reach_ball.py --envs 4096 --iters 10 --train-steps 200 --test-steps 250
"""
import argparse
import copy
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_environments.environment_factory import EnvironmentFactory  # noqa: E402

kewargs = {                      # dqn_stable_baselines3.py:18-31
    'change_ball_position': True, 'change_ball_velocity': True,
    'ball_position_x': 0, 'ball_position_y': 0, 'ball_speed': 0, 'ball_direction': 0,
    'min_distance_to_ball': 5.0, 'max_steps': 200,
    'use_continuous_action': False, 'action_space_size': 16, 'use_turning': False,
}


class QNet(nn.Module):           # SB3 DQN "MlpPolicy" default: two hidden layers of 64
    def __init__(self, n_obs=10, n_act=16, hidden=64):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(n_obs, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(),
                                 nn.Linear(hidden, n_act))

    def forward(self, x):
        return self.net(x)


class DeviceReplay:
    def __init__(self, capacity, n_obs, device):
        self.cap, self.pos, self.full = capacity, 0, False
        self.obs = torch.empty((capacity, n_obs), device=device)
        self.next_obs = torch.empty((capacity, n_obs), device=device)
        self.act = torch.empty((capacity,), dtype=torch.int64, device=device)
        self.rew = torch.empty((capacity,), device=device)
        self.term = torch.empty((capacity,), device=device)

    def add(self, obs, act, rew, next_obs, term):
        n = obs.shape[0]
        idx = (torch.arange(n, device=obs.device) + self.pos) % self.cap
        self.obs[idx], self.act[idx], self.rew[idx], self.next_obs[idx], self.term[idx] = obs, act, rew, next_obs, term
        self.full |= self.pos + n >= self.cap
        self.pos = (self.pos + n) % self.cap

    def sample(self, batch):
        hi = self.cap if self.full else self.pos
        i = torch.randint(0, hi, (batch,), device=self.obs.device)
        return self.obs[i], self.act[i], self.rew[i], self.next_obs[i], self.term[i]


class DeviceDQN:
    def __init__(self, env, lr=1e-3, gamma=0.99, buffer=1 << 20, batch=4096, target_every=50, grad_steps=4,
                 eps_start=1.0, eps_end=0.05, eps_decay_steps=300, seed=0):
        torch.manual_seed(seed)
        self.env, self.dev = env, env.device
        self.n_act = env.action_space.n
        self.q = QNet(env.observation_space.shape[0], self.n_act).to(self.dev)
        self.q_target = copy.deepcopy(self.q)
        self.opt = torch.optim.Adam(self.q.parameters(), lr=lr)
        self.rb = DeviceReplay(buffer, env.observation_space.shape[0], self.dev)
        self.gamma, self.batch, self.target_every, self.grad_steps = gamma, batch, target_every, grad_steps
        self.eps_start, self.eps_end, self.eps_decay = eps_start, eps_end, eps_decay_steps
        self.steps = 0
        self.obs = env.reset().clone()

    def epsilon(self):
        f = min(1.0, self.steps / self.eps_decay)
        return self.eps_start + f * (self.eps_end - self.eps_start)

    @torch.no_grad()
    def predict(self, obs, eps=0.0):
        greedy = self.q(obs).argmax(dim=1)
        if eps <= 0:
            return greedy
        rnd = torch.randint(0, self.n_act, greedy.shape, device=self.dev)
        return torch.where(torch.rand(greedy.shape, device=self.dev) < eps, rnd, greedy)

    def learn(self, vec_steps, on_result=None):
        for _ in range(vec_steps):
            act = self.predict(self.obs, self.epsilon())
            nobs, rew, done, info = self.env.step(act)
            res = info['result']
            # bootstrap through time-limit truncation (Timeout) with the terminal observation
            next_obs = torch.where(done.bool().unsqueeze(1), info['terminal_observation'], nobs)
            term = ((res == 1) | (res == 2)).float()          # Goal / Out are true terminations
            self.rb.add(self.obs, act, rew, next_obs, term)
            if on_result is not None:
                on_result(res)
            self.obs = nobs.clone()
            self.steps += 1
            if self.rb.full or self.rb.pos >= self.batch:
                for _g in range(self.grad_steps):
                    o, a, r, no, t = self.rb.sample(self.batch)
                    with torch.no_grad():
                        tgt = r + self.gamma * (1 - t) * self.q_target(no).max(dim=1).values
                    loss = nn.functional.smooth_l1_loss(self.q(o).gather(1, a.unsqueeze(1)).squeeze(1), tgt)
                    self.opt.zero_grad(set_to_none=True)
                    loss.backward()
                    nn.utils.clip_grad_norm_(self.q.parameters(), 10.0)
                    self.opt.step()
            if self.steps % self.target_every == 0:
                self.q_target.load_state_dict(self.q.state_dict())


def test(env, model, vec_steps):
    """dqn_stable_baselines3.py:44-62 -- greedy policy, count info['result'] of finished episodes."""
    obs = env.reset()
    counts = torch.zeros(4, dtype=torch.int64, device=env.device)
    ret = torch.zeros(env.num_envs, device=env.device)
    ep_ret_sum = torch.zeros((), device=env.device)
    for _ in range(vec_steps):
        act = model.predict(obs) if model is not None else None
        obs, rew, done, info = env.step(act)
        ret += rew
        d = done.bool()
        ep_ret_sum += ret[d].sum()
        ret[d] = 0
        counts += torch.bincount(info['result'].long(), minlength=4)
    c = counts.cpu().tolist()
    n = max(1, c[1] + c[2] + c[3])
    return {'Goal': c[1] / n, 'Out': c[2] / n, 'Timeout': c[3] / n, 'episodes': n,
            'mean_return': float(ep_ret_sum.item()) / n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--train-steps', type=int, default=200)
    ap.add_argument('--test-steps', type=int, default=250)
    ap.add_argument('--device', default='cuda:0')
    args = ap.parse_args()
    env = EnvironmentFactory().create_vec('reachball', args.envs, device=args.device, **kewargs)
    test_env = EnvironmentFactory().create_vec('reachball', args.envs, device=args.device, seed=1234, **kewargs)
    model = DeviceDQN(env)
    print('random policy:', test(test_env, None, args.test_steps))
    for i in range(args.iters):
        t0 = time.time()
        model.learn(args.train_steps)
        torch.cuda.synchronize()
        dt = time.time() - t0
        r = test(test_env, model, args.test_steps)
        print(f'iter {i}: {args.train_steps * args.envs / dt / 1e6:.2f} M env-steps/s incl. learning  eps={model.epsilon():.2f}  {r}')
    env.close(); test_env.close()


if __name__ == '__main__':
    main()
