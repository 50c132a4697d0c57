"""The DEVICE code of the three ReachBallEnv hooks against the golden vectors produced by running the
reference's own Python (tests/golden/*.json, rows A2-A4 of SURVEY.md 8(a)) -- directly, not through the
oracle: s2d_debug_eval ops 6-8 evaluate action_map / observe / observe_and_check of csrc/s2d_device.h on
the GPU for the fixture inputs.

Tolerances (the engine computes in fp32, the reference in float64 on float32-rounded inputs): observations
2e-6, action directions 3e-5 deg, rewards 1e-4 * max(1, |r|), carry distance 1e-4, carry angle 1e-3 deg;
done / result / command are compared exactly.  The same golden files pin the CPU oracle in
tests/test_oracle_golden.py with the same tolerances, and the GPU equals the fp32 oracle bit for bit.
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

G = os.path.join(os.path.dirname(__file__), 'golden')
CMD = {'dash': 1, 'turn': 2}
RES = {None: 0, 'Goal': 1, 'Out': 2, 'Timeout': 3}


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def run(op, rows, out_cols):
    import ctypes as C
    from soccer2d_amd import _capi
    lib = _capi.load_library()
    x = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.float32), device='cuda:0')
    y = torch.zeros((x.shape[0], out_cols), dtype=torch.float32, device='cuda:0')
    _capi.check(lib, lib.s2d_debug_eval(op, x.data_ptr(), y.data_ptr(), x.shape[0], None), 's2d_debug_eval')
    torch.cuda.synchronize()
    return y.cpu().numpy().astype(np.float64)


def test_observation_hook_on_device_vs_reference():
    g = load('obs.json')
    ins = np.array([r['in'] for r in g['rows']])                       # bx, by, bvx, bvy, px, py, body
    # the engine stores body directions normalised (AngleDeg.normal(), Appendix C), so the device hook takes them so
    b = ins[:, 6]
    b = np.where(np.abs(b) > 360.0, np.fmod(b, 360.0), b)
    ins[:, 6] = np.where(b < -180.0, b + 360.0, np.where(b > 180.0, b - 360.0, b))
    inv = np.tile([np.float32(1.0 / 52.5), np.float32(1.0 / 34.0)], (len(ins), 1))
    got = run(6, np.hstack([ins, inv]), 10)
    ref = np.array([r['obs'] for r in g['rows']])
    err = np.abs(got - ref)
    err[:, 0] = np.where(err[:, 0] > 1.0, np.abs(err[:, 0] - 2.0), err[:, 0])    # +-180 deg are the same direction
    err[:, 7] = np.where(err[:, 7] > 0.5, np.abs(err[:, 7] - 1.0), err[:, 7])
    assert len(ins) >= 256 and err.max() <= 2e-6, err.max()


def test_action_map_hook_on_device_vs_reference():
    g = load('action_map.json')
    rows, want = [], []
    for blk in g['discrete']:
        for r in blk['rows']:
            rows.append([r['a'], 0, 0, 0, 0, 0, np.float32(360.0 / blk['n']), 0]); want.append((CMD[r['type']], r['power'], r['dir']))
    for r in g['continuous']:
        rows.append([r['a'], 0, 0, 0, 0, 1, 0, 0]); want.append((1, 100.0, r['dir']))
    n_turn = 0
    for r in g['turning']:
        c = np.clip(np.asarray(r['a']), -1, 1)
        p0 = math.exp(c[2]) / (math.exp(c[2]) + math.exp(c[0]))
        if abs(r['u'] - p0) < 1e-6:          # the uniform sits within fp32 rounding of the softmax threshold
            continue
        rows.append(list(r['a']) + [r['u'], 2, 0, 0]); want.append((CMD[r['type']], r['power'], r['dir']))
        n_turn += r['type'] == 'turn'
    got = run(7, rows, 3)
    want = np.array(want)
    assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 1], want[:, 1])
    assert np.abs(got[:, 2] - want[:, 2]).max() <= 3e-5
    assert n_turn > 10 and len(rows) > 140


def test_reward_hook_on_device_vs_reference():
    seqs = load('reward.json')
    seen, worst = set(), 0.0
    for s in seqs:
        carry = [0.0, 0.0]
        # the hook carries state from row to row: feed the device its OWN previous outputs, like the engine does
        for row in s['rows']:
            bx, by, px, py, body = row['in']
            body = math.fmod(body, 360.0) if abs(body) > 360.0 else body
            body = body + 360.0 if body < -180.0 else (body - 360.0 if body > 180.0 else body)
            got = run(8, [[bx, by, px, py, body, row['step_number'], carry[0], carry[1], s['min_distance_to_ball'],
                           s['max_steps'], 52.5, 34.0]], 5)[0]
            done, rw, res, carry[0], carry[1] = bool(got[0]), got[1], int(got[2]), float(np.float32(got[3])), float(np.float32(got[4]))
            assert done == row['done'] and res == RES[row['result']], (s['note'], row)
            assert abs(rw - row['reward']) <= 1e-4 * max(1.0, abs(row['reward'])), (s['note'], row, rw)
            assert abs(carry[0] - row['carry_dist']) <= 1e-4 and abs(carry[1] - row['carry_angle']) <= 1e-3
            worst = max(worst, abs(rw - row['reward']))
            seen.add((row['done'], row['result']))
    assert len(seqs) >= 64 and {(True, 'Goal'), (True, 'Out'), (True, 'Timeout'), (False, None)} <= seen
