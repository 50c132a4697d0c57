"""Drop-in for ``sample_environments.reach_ball_env`` (reference reach_ball_env.py:17-218).

Same class name, same kwargs and defaults (reach_ball_env.py:26-36), same spaces (:39-48),
same 10-float observation (:98-107), reward / done / info['result'] rules (:113-161) and
reset sampler (:170-218) -- computed by the HIP kernels instead of Python + rcssserver.
"""
from soccer_2d_env import Soccer2DEnv
from soccer2d_amd.engine import TASK_KWARGS
from soccer2d_amd.vec_env import Soccer2DVecEnv


class ReachBallEnv(Soccer2DEnv):
    def __init__(self, render_mode=None, logger=None, log_dir=None, **kwargs):
        unknown = [k for k in kwargs if k not in TASK_KWARGS and k not in ('device', 'seed', 'noise', 'server_params')]
        # the reference silently ignores unknown kwargs (kwargs.get with defaults); keep that
        for k in unknown:
            kwargs.pop(k)
        super().__init__(render_mode, logger=logger, log_dir=log_dir, **kwargs)
        t = self.vec.engine.cfg.task
        for k in TASK_KWARGS:                     # reference exposes them as attributes (:26-36)
            setattr(self, k, type(TASK_KWARGS[k])(getattr(t, k)))

    # the reference tracks these three as Python attributes (:49-51); here they are state words
    @property
    def step_number(self):
        return int(self.vec.engine.step_number[0].item())

    @property
    def distance_to_ball(self):
        return float(self.vec.engine.prev_dist[0].item())

    @property
    def body_ball_angle_diff(self):
        return float(self.vec.engine.prev_angle[0].item())


class ReachBallVecEnv(Soccer2DVecEnv):
    """N reach_ball envs on one GPU (device tensors in / out)."""
