"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/rl4rs_hip.h declares;
without a GPU the product path fails loudly (no CPU fallback)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from rl4rs_amd.build import build_lib
    build_lib()
    from rl4rs_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(REPO, 'include', 'rl4rs_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rl4rs_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound(lib):
    from rl4rs_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), 'include/rl4rs_hip.h declares %s but the .so does not export it' % name
    # the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.rl4rs_abi_version() == 1


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(REPO, 'include', 'rl4rs_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)       # code only, not the comments
    for banned in ('torch', 'at::', 'std::', 'hipStream_t', '#include <hip'):
        assert banned not in text, banned


def test_fails_loudly_without_a_gpu(lib, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from rl4rs_amd import _lib, synth
    from rl4rs_amd.data import CatalogTables
    from rl4rs_amd.device import DeviceEnv
    assert lib.rl4rs_device_count() <= 0
    p = str(tmp_path / 'item_info.csv')
    synth.write_text(p, synth.make_catalog_text())
    cfg = {"batch_size": 4, "max_steps": 9, "action_size": 284, "maxlen": 64, "dense_feature_num": 432,
           "category_feature_num": 21}
    with pytest.raises(_lib.Rl4rsHipError):
        DeviceEnv(cfg, CatalogTables(p, 284), False, 9, True)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under rl4rs_amd/ or rl4rs/ may reference it."""
    bad = []
    for pkg in ('rl4rs_amd', 'rl4rs'):
        for root, _, files in os.walk(os.path.join(REPO, pkg)):
            for f in files:
                if f.endswith(('.py', '.hip', '.hpp', '.h')):
                    src = open(os.path.join(root, f)).read()
                    if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'oracle/' in src:
                        bad.append(os.path.join(root, f))
    assert not bad, bad


def test_library_missing_is_an_error(monkeypatch):
    from rl4rs_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/librl4rs_hip.so')
    with pytest.raises(_lib.Rl4rsHipError):
        _lib.load()


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/rl4rs_hip.h must compile as C99 on its own (no C++, no torch types)."""
    import subprocess
    src = tmp_path / 'hdr.c'
    src.write_text('#include "rl4rs_hip.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-I', inc, str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
