"""CPU-side checks of the drop-in boundary: libs2d_hip.so loads, exports every symbol that
include/s2d.h declares, the ctypes mirror matches the C struct sizes, and the product path
fails loudly (never falls back) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'include', 's2d.h')


MATCH_HDR = os.path.join(ROOT, 'include', 's2d_match.h')


def declared_functions(path=None):
    src = open(path or HDR).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(s2d_[a-z_0-9]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build_hip()
    from soccer2d_amd import _capi
    return _capi.load_library()


def test_header_functions_all_exported(lib):
    from soccer2d_amd import _capi
    names = declared_functions()
    assert len(names) >= 15
    bound = {p[0] for p in _capi.PROTOTYPES}
    assert set(names) == bound, (set(names) ^ bound)
    for n in names:
        assert hasattr(lib, n), n


def test_match_header_functions_all_exported(lib):
    from soccer2d_amd import _capi_match as M
    names = [n for n in declared_functions(MATCH_HDR) if n.startswith('s2d_match_')]
    assert set(names) == {p[0] for p in M.MATCH_PROTOTYPES}
    M.bind(lib)
    for n in names:
        assert hasattr(lib, n), n
    import ctypes as C
    import match_oracle as MO
    cfg = M.S2DMatchConfig()
    lib.s2d_match_default_config(C.byref(cfg))
    assert bytes(memoryview(cfg)) == bytes(memoryview(MO.make_match_config()))
    assert lib.s2d_match_validate_config(C.byref(cfg)) == 0
    assert lib.s2d_match_arena_bytes(C.byref(cfg), 8192) >= 8192 * (24 * 10 * 4 + 12 * 4)


def test_gtc_header_functions_all_exported(lib):
    from soccer2d_amd import gtc
    names = [n for n in declared_functions(os.path.join(ROOT, 'include', 's2d_gtc.h')) if n.startswith('s2d_gtc_')]
    assert set(names) == {p[0] for p in gtc.GTC_PROTOTYPES}
    gtc.bind(lib)
    for n in names:
        assert hasattr(lib, n), n
    cfg = gtc.S2DGtcConfig()
    lib.s2d_gtc_default_config(C.byref(cfg))
    assert (cfg.x_min, cfg.x_max, cfg.y_min, cfg.y_max, cfg.max_steps, cfg.min_distance_to_center) == (-52.5, 52.5, -34.0, 34.0, 200, 5.0)
    assert lib.s2d_gtc_arena_bytes(C.byref(cfg), 1000) > 1000 * 7 * 4


def test_match_struct_sizes_match_c(tmp_path):
    from soccer2d_amd import _capi_match as M
    prog = tmp_path / 'szm.c'
    prog.write_text('#include <stdio.h>\n#include "s2d_match.h"\n#include "s2d_gtc.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                    'sizeof(S2DMatchConfig),sizeof(S2DMatchParams),sizeof(S2DMatchBuffers),sizeof(S2DMatchRollout),sizeof(S2DGtcConfig));return 0;}\n')
    exe = tmp_path / 'szm'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(prog), '-o', str(exe)], check=True)
    got = list(map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()))
    from soccer2d_amd.gtc import S2DGtcConfig
    want = [C.sizeof(x) for x in (M.S2DMatchConfig, M.S2DMatchParams, M.S2DMatchBuffers, M.S2DMatchRollout, S2DGtcConfig)]
    assert got == want


def test_struct_sizes_match_c(tmp_path):
    from soccer2d_amd import _capi
    prog = tmp_path / 'sz.c'
    prog.write_text('#include <stdio.h>\n#include "s2d.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(S2DConfig),sizeof(S2DServerParams),sizeof(S2DReachBallParams),sizeof(S2DBuffers),'
                    'sizeof(S2DRollout),sizeof(S2DWorldModel));return 0;}\n')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(prog), '-o', str(exe)], check=True)
    got = list(map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()))
    want = [C.sizeof(x) for x in (_capi.S2DConfig, _capi.S2DServerParams, _capi.S2DReachBallParams,
                                  _capi.S2DBuffers, _capi.S2DRollout, _capi.S2DWorldModel)]
    assert got == want


def test_default_config_and_validation(lib):
    from soccer2d_amd import _capi
    import oracle as O
    cfg = _capi.S2DConfig()
    lib.s2d_default_config(C.byref(cfg))
    assert bytes(memoryview(cfg)) == bytes(memoryview(O.make_config()))      # product defaults == test table
    assert lib.s2d_validate_config(C.byref(cfg)) == 0
    assert lib.s2d_arena_bytes(C.byref(cfg), 65536) >= 65536 * (17 * 4 + 80 + 4 + 4 + 3)
    assert lib.s2d_arena_bytes(C.byref(cfg), 0) == 0
    cfg.abi_version = 99
    assert lib.s2d_validate_config(C.byref(cfg)) == _capi.S2D_EINVAL
    assert b'abi_version' in lib.s2d_last_error()
    assert lib.s2d_version().startswith(b's2d-hip') and b'abi %d)' % _capi.S2D_ABI_VERSION in lib.s2d_version()


def test_no_gpu_no_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from soccer2d_amd.engine import Engine
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Engine(4, 'cuda:0', use_continuous_action=False)
    # the C ABI itself reports the missing device instead of computing anything
    from soccer2d_amd import _capi
    cfg = _capi.S2DConfig()
    lib.s2d_default_config(C.byref(cfg))
    h = C.c_void_p()
    rc = lib.s2d_create(C.byref(cfg), 4, 0, None, 0, None, C.byref(h))
    assert rc in (_capi.S2D_ENODEV, _capi.S2D_EHIP) and not h.value


def test_missing_library_is_an_error(tmp_path):
    from soccer2d_amd import _capi
    with pytest.raises(_capi.S2DLibraryError):
        _capi.load_library(str(tmp_path / 'nope.so'))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, 'gym-soccer-2d-env_amd')
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f), errors='ignore').read()
                for needle in ('s2do_', 'import oracle', 'from oracle', 'oracle/', 'libs2d_oracle', 'oracle.py'):
                    assert needle not in txt, (os.path.join(dp, f), needle)
