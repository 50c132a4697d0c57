"""Drop-in for ``sample_environments.environment_factory`` (reference environment_factory.py:
14-28): ``EnvironmentFactory().create("reachball", render_mode, logger, log_dir, **kwargs)``.
``create_vec`` is the batched entry point this framework adds."""
from sample_environments.reach_ball_env import ReachBallEnv, ReachBallVecEnv

_SINGLE = {'reachball': ReachBallEnv}
_VECTOR = {'reachball': ReachBallVecEnv}


class EnvironmentFactory:
    def create(self, env_name, render_mode=None, logger=None, log_dir=None, **kwargs):
        cls = _SINGLE.get(str(env_name).lower())
        if cls is None:
            raise ValueError(f"Environment {env_name} not found.")
        return cls(render_mode=render_mode, logger=logger, log_dir=log_dir, **kwargs)

    def create_vec(self, env_name, num_envs, device='cuda:0', **kwargs):
        cls = _VECTOR.get(str(env_name).lower())
        if cls is None:
            raise ValueError(f"Environment {env_name} not found.")
        return cls(num_envs, device=device, **kwargs)
