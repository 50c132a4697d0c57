"""SURVEY section 5 (race detection / sanitizers): the three CPU oracles -- what the HIP kernels are checked against -- run under
AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle sanitize`): ragged env counts, masked resets, rollouts with
records, the 11v11 rules and the GoToCenter turn mode.  (GPU AddressSanitizer is not available on this pool.)"""
import ctypes as C
import os
import subprocess

import oracle as O
import match_oracle as MO
from test_gtc import gtc_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracles_under_asan_and_ubsan(tmp_path):
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'sanitize'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    cfgs = {'reach.cfg': O.make_config(noise=1, **dict(O.DQN_KWARGS, max_steps=40)),
            'match.cfg': MO.make_match_config(noise=1),
            'gtc.cfg': gtc_cfg(continuous=1, turn=1, use_turn=1, actor_out_size=4, max_steps=30)}
    for name, cfg in cfgs.items():
        (tmp_path / name).write_bytes(bytes(memoryview(cfg)))
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    r = subprocess.run([os.path.join(ROOT, 'oracle', '_build', 'sanitize_driver')] + [str(tmp_path / n) for n in cfgs],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert r.returncode == 0 and 'sanitize ok' in r.stdout, r.stdout[-4000:]
    assert 'runtime error' not in r.stdout and 'AddressSanitizer' not in r.stdout, r.stdout[-4000:]
