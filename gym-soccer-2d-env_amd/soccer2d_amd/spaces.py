"""Action/observation space objects.  The reference uses ``gym.spaces`` (reach_ball_env.py:
39-48); gym is an optional dependency here: when gymnasium or gym is importable their Box /
Discrete are used (so SB3 accepts the env unchanged), otherwise these minimal stand-ins with
the same attributes (shape, dtype, n, low, high, sample, contains)."""
import numpy as np

try:  # pragma: no cover - depends on the installation
    from gymnasium.spaces import Box, Discrete  # type: ignore
    BACKEND = 'gymnasium'
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        from gym.spaces import Box, Discrete  # type: ignore
        BACKEND = 'gym'
    except Exception:  # noqa: BLE001
        BACKEND = 'builtin'

        class Discrete:  # noqa: D101
            def __init__(self, n):
                self.n = int(n)
                self.shape = ()
                self.dtype = np.dtype(np.int64)

            def sample(self):
                return int(np.random.randint(self.n))

            def contains(self, x):
                try:
                    return 0 <= int(x) < self.n
                except (TypeError, ValueError):
                    return False

            def __repr__(self):
                return f'Discrete({self.n})'

        class Box:  # noqa: D101
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.dtype = np.dtype(dtype)
                if shape is None:
                    shape = np.shape(low)
                self.shape = tuple(shape)
                self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
                self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

            def sample(self):
                return np.random.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'


def reach_ball_spaces(use_continuous_action=True, use_turning=False, action_space_size=16):
    """Spaces of ReachBallEnv, reach_ball_env.py:39-48."""
    if use_continuous_action:
        if use_turning:
            act = Box(low=np.array([-1, -1, -1, -1], dtype=np.float32),
                      high=np.array([1, 1, 1, 1], dtype=np.float32), dtype=np.float32)
        else:
            act = Box(low=-1, high=1, shape=(1,), dtype=np.float32)
    else:
        act = Discrete(action_space_size)
    obs = Box(low=-1, high=1, shape=(10,), dtype=np.float32)
    return act, obs
