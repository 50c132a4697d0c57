"""Drop-in for the ``GoToCenterEnv`` class of the reference's python_sample_soccer_env.py
(:46-255): same constructor (continuous, turn, actor_out_size, use_turn -- the script's argparse
defaults are True / True / 4 / True, :356-359), same 4-float observation, same reward /
termination rules and the same gymnasium-style return shapes
(reset -> (obs, {}); step -> (obs, reward, done, done, {'result': ''|'Goal'|'Out'|'Timeout'})),
computed by the HIP engine (num_envs = 1 view of soccer2d_amd.gtc.GoToCenterVecEnv)."""
import numpy as np

from soccer2d_amd.gtc import GoToCenterVecEnv

_NAMES = ('', 'Goal', 'Out', 'Timeout')


class GoToCenterEnv:
    metadata = {'render.modes': ['human']}

    def __init__(self, continuous=False, turn=False, actor_out_size=1, use_turn=False, device='cuda:0', seed=0x5EED):
        self.continuous, self.turn, self.use_turn = bool(continuous), bool(turn), bool(use_turn)
        turn_mode = self.turn and self.continuous                       # :70 -- turn without continuous is the discrete env
        self.vec = GoToCenterVecEnv(1, device=device, continuous=int(self.continuous), turn=int(self.turn),
                                    use_turn=int(self.use_turn), actor_out_size=int(actor_out_size) if turn_mode else 1,
                                    auto_reset=False, seed=seed)
        self.action_space, self.observation_space = self.vec.action_space, self.vec.observation_space
        self.max_steps, self.min_distance_to_center = 200, 5.0

    def reset(self, seed=None, options=None):
        return self.vec.reset()[0].cpu().numpy().astype(np.float32), {}

    def step(self, action):
        a = np.asarray(action).reshape(-1)[:self.vec.action_dim]
        obs, reward, done, info = self.vec.step(a.astype(np.float32) if self.continuous else a.astype(np.int32))
        d = bool(done[0].item())
        return (obs[0].cpu().numpy().astype(np.float32), float(reward[0].item()), d, d,
                {'result': _NAMES[int(info['result'][0].item())]})

    def render(self, mode='human'):
        return None

    def close(self):
        self.vec.close()
