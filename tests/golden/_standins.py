"""Stand-in modules used ONLY by tests/golden/make_golden.py (fixture generation).

The reference (`/root/reference`) imports two third-party packages that are not
installed in this container and cannot be installed (no network):

* ``gym==0.26.2``      -- only ``gym.Env`` (as a base class) and ``gym.spaces.Box`` /
  ``gym.spaces.Discrete`` (as attribute holders) are touched on the path.
* ``pyrusgeom==0.1.2`` -- ``Vector2D`` / ``AngleDeg`` (requirements.txt:6 of the
  reference).  Its published algorithm (a Python transliteration of librcsc's
  ``rcsc/geom/vector_2d.h`` and ``angle_deg.h``) is restated here:

  - ``AngleDeg(d)`` keeps degrees normalised: ``fmod(d, 360)`` when ``|d| > 360``,
    then ``+360`` if ``< -180`` and ``-360`` if ``> 180``  (so the closed range is
    ``[-180, 180]``; ``-180`` itself is left alone -- edge "by convention").
  - ``a - b`` -> ``AngleDeg(a.degree() - b.degree())``; ``.abs()`` -> ``|degree|``.
  - ``Vector2D.th()`` -> ``AngleDeg(atan2(y, x) in degrees)``, ``0`` for the exact
    zero vector; ``.r()`` -> ``sqrt(x*x + y*y)``; ``.dist(o)`` likewise;
    ``Vector2D.from_polar(r, deg)`` -> ``(r cos, r sin)``.

Nothing here is shipped with the product; the product never imports it.
"""
import math
import sys
import types

RAD2DEG = 180.0 / math.pi
DEG2RAD = math.pi / 180.0


class AngleDeg:
    def __init__(self, degree=0.0):
        if isinstance(degree, AngleDeg):
            degree = degree.degree()
        self._degree = float(degree)
        self._normal()

    def _normal(self):
        if self._degree < -360.0 or 360.0 < self._degree:
            self._degree = math.fmod(self._degree, 360.0)
        if self._degree < -180.0:
            self._degree += 360.0
        if self._degree > 180.0:
            self._degree -= 360.0

    def degree(self):
        return self._degree

    def abs(self):
        return math.fabs(self._degree)

    def radian(self):
        return self._degree * DEG2RAD

    def __sub__(self, other):
        o = other.degree() if isinstance(other, AngleDeg) else float(other)
        return AngleDeg(self._degree - o)

    def __add__(self, other):
        o = other.degree() if isinstance(other, AngleDeg) else float(other)
        return AngleDeg(self._degree + o)

    def __neg__(self):
        return AngleDeg(-self._degree)

    def __repr__(self):
        return str(self._degree)

    @staticmethod
    def atan2_deg(y, x):
        if x == 0.0 and y == 0.0:
            return 0.0
        return math.atan2(y, x) * RAD2DEG


class Vector2D:
    def __init__(self, x=0.0, y=0.0):
        self._x = float(x)
        self._y = float(y)

    def x(self):
        return self._x

    def y(self):
        return self._y

    def abs_x(self):
        return math.fabs(self._x)

    def abs_y(self):
        return math.fabs(self._y)

    def r(self):
        return math.sqrt(self._x * self._x + self._y * self._y)

    def th(self):
        return AngleDeg(AngleDeg.atan2_deg(self._y, self._x))

    def dist(self, o):
        dx = self._x - o._x
        dy = self._y - o._y
        return math.sqrt(dx * dx + dy * dy)

    def __sub__(self, o):
        return Vector2D(self._x - o._x, self._y - o._y)

    def __add__(self, o):
        return Vector2D(self._x + o._x, self._y + o._y)

    def __repr__(self):
        return f"({self._x}, {self._y})"

    @staticmethod
    def from_polar(r, deg):
        if isinstance(deg, AngleDeg):
            deg = deg.degree()
        return Vector2D(r * math.cos(deg * DEG2RAD), r * math.sin(deg * DEG2RAD))

    @staticmethod
    def polar2vector(r, deg):
        return Vector2D.from_polar(r, deg)


class _Space:
    pass


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=None):
        self.low, self.high, self.dtype = low, high, dtype
        if shape is None:
            shape = getattr(low, "shape", ())
        self.shape = tuple(shape)


class Discrete(_Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()


class Env:
    metadata = {}

    def reset(self, seed=None, options=None):      # gym.Env.reset: seeds self.np_random only; nothing the path reads
        return None


def install_sb3():
    """``stable_baselines3`` is imported at module level by python_sample_soccer_env.py (:9) and by
    utils/info_collector_callback.py; GoToCenterEnv itself never touches it.  Attribute holders only."""
    if "stable_baselines3" in sys.modules:
        return
    sb3 = types.ModuleType("stable_baselines3")

    class _Algo:
        def __init__(self, *a, **k):
            raise RuntimeError("stand-in: stable_baselines3 is not installed")

    class BaseCallback:
        def __init__(self, verbose=0):
            self.verbose = verbose

    common = types.ModuleType("stable_baselines3.common")
    cb = types.ModuleType("stable_baselines3.common.callbacks")
    cb.BaseCallback = BaseCallback
    sb3.DQN = sb3.DDPG = _Algo
    sb3.common, common.callbacks = common, cb
    sys.modules["stable_baselines3"] = sb3
    sys.modules["stable_baselines3.common"] = common
    sys.modules["stable_baselines3.common.callbacks"] = cb


def install():
    """Put the stand-ins into sys.modules (idempotent)."""
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        spaces = types.ModuleType("gym.spaces")
        spaces.Box, spaces.Discrete, spaces.Space = Box, Discrete, _Space
        gym.Env, gym.spaces = Env, spaces
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces
    if "pyrusgeom" not in sys.modules:
        pg = types.ModuleType("pyrusgeom")
        g2 = types.ModuleType("pyrusgeom.geom_2d")
        g2.Vector2D, g2.AngleDeg = Vector2D, AngleDeg
        pg.geom_2d = g2
        sys.modules["pyrusgeom"] = pg
        sys.modules["pyrusgeom.geom_2d"] = g2
