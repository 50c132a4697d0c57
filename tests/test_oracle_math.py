"""The deterministic fp32 math spec as implemented by the oracle:
accuracy against libm double, exact special values, and Philox4x32-10 known answers."""
import math

import numpy as np

import oracle as O


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_sincos_deg_accuracy():
    rs = np.random.RandomState(0)
    degs = np.concatenate([rs.uniform(-720, 720, 20000), np.arange(-720, 721, 0.5)]).astype(np.float32)
    worst = 0.0
    for d in degs:
        s, c = O.sincos_deg(float(d))
        rs_, rc_ = math.sin(math.radians(float(d))), math.cos(math.radians(float(d)))
        worst = max(worst, abs(s - rs_), abs(c - rc_))
    assert worst < 2.5e-7, worst


def test_sincos_deg_exact_cardinals():
    for d, (s, c) in {0: (0, 1), 90: (1, 0), 180: (0, -1), -90: (-1, 0), 270: (-1, 0), 360: (0, 1), -180: (0, -1)}.items():
        assert O.sincos_deg(d) == (s, c)
    s, c = O.sincos_deg(45.0)
    assert abs(s - math.sqrt(0.5)) < 1e-7 and abs(c - math.sqrt(0.5)) < 1e-7
    # 22.5-degree grid of the 16 discrete actions: sin^2 + cos^2 == 1 to fp32 rounding
    for k in range(16):
        s, c = O.sincos_deg(22.5 * k - 180)
        assert abs(s * s + c * c - 1) < 3e-7


def test_atan2_deg_accuracy_and_range():
    rs = np.random.RandomState(1)
    worst = 0.0
    for _ in range(30000):
        y, x = (float(np.float32(v)) for v in rs.uniform(-110, 110, 2))
        a = O.atan2_deg(y, x)
        r = math.degrees(math.atan2(y, x))
        e = abs(a - r)
        e = min(e, abs(e - 360))
        worst = max(worst, e)
        assert -180.0 <= a <= 180.0
    assert worst < 3e-5, worst      # ~2 ulp of 180 in fp32
    assert O.atan2_deg(0, 0) == 0.0 and O.atan2_deg(0, 1) == 0.0 and O.atan2_deg(1, 0) == 90.0
    assert O.atan2_deg(0, -1) == 180.0 and O.atan2_deg(-1, 0) == -90.0
    assert O.atan2_deg(1, 1) == 45.0 and O.atan2_deg(-1, -1) == -135.0


def test_exp_accuracy():
    for x in np.linspace(-3, 3, 2001):
        x = float(np.float32(x))
        assert abs(O.exp(x) / math.exp(x) - 1) < 3e-7
    assert O.exp(0.0) == 1.0


def test_f64_build_is_libm():
    assert O.lib('f64').s2do_real_bytes() == 8 and O.lib('f32').s2do_real_bytes() == 4
    s, c = O.sincos_deg(33.3, 'f64')
    assert s == math.sin(33.3 * (math.pi / 180)) and c == math.cos(33.3 * (math.pi / 180))
    assert O.atan2_deg(1.5, -2.5, 'f64') == math.atan2(1.5, -2.5) * (180 / math.pi)
