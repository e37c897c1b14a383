"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly
the symbols include/shg_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, 'include', 'shg_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(shg_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
    from solex_ser_recon_en_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(_lib.lib, name), 'libshg_hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, 'no ctypes signature for %s' % name
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib.shg_abi_version() == _lib.ABI_VERSION == 16
    assert isinstance(_lib.last_error(), str)


def test_workspace_queries_need_no_gpu():
    from solex_ser_recon_en_amd._lib import lib
    assert lib.shg_accumulate_workspace_bytes(2000, 200, 2000, 2) >= 400000 * 6
    assert lib.shg_accumulate_workspace_bytes(0, 200, 2000, 2) == 0
    assert lib.shg_accumulate_workspace_bytes(10, 200, 2000, 3) == 0
    assert lib.shg_clahe_workspace_bytes(2, 2) == 4 * 65536 * 6
    assert lib.shg_clahe_workspace_bytes(3, 1) == 9 * 256 * 6
    assert lib.shg_clahe_workspace_bytes(0, 2) == 0
    # the roomier 16-bit layout: [hist | lut], then one 128 KiB slice histogram per run of whole tile rows holding at most 32768
    # pixels (1000 x 1048 tiles: 31 rows a slice, 33 slices), chunk sums, clipped totals (128 words a tile: one per 512 bins for the
    # saturated slices, 32 x 2 for the u16 ones)
    assert lib.shg_clahe_workspace_bytes_for(2000, 2096, 2, 2) == 4 * 65536 * 6 + 4 * 33 * 131072 + 4 * 1024 * 4 + 4 * 128 * 4
    assert lib.shg_clahe_workspace_bytes_for(77, 50, 4, 1) == lib.shg_clahe_workspace_bytes(4, 1)
    assert lib.shg_contrast_stats_workspace_bytes_for(2000, 2096, 2) == lib.shg_contrast_stats_workspace_bytes(2) + 4 * 33 * 131072 + 4 * 1024 * 4 + 4 * 128 * 4


def test_argument_errors_are_reported_not_thrown():
    """Bad arguments are rejected before any HIP call, so this runs without a GPU."""
    from solex_ser_recon_en_amd import _lib
    lib = _lib.lib
    assert lib.shg_accumulate_sum_max(None, 1, 1, 1, 2, 0, None, None, None, 0, None) == -1
    assert 'null pointer' in _lib.last_error()
    one = ctypes.c_void_p(16)
    assert lib.shg_extract_columns(one, 0, 4, 4, 2, 0, one, one, one, 1, one, 4, 16, 4, 0, 0, None) == -1
    assert lib.shg_frame_pitch_bytes(800000) == 802816 and lib.shg_frame_pitch_bytes(1310720) == 1310720
    assert lib.shg_box_blur_u16(one, 10, 10, 25, 0, one, one, None) == -1      # cv2.blur rejects a zero kernel too
    assert 'must be positive' in _lib.last_error()
    assert lib.shg_rescale_u16(one, 4, 4, 4, 5.0, 5.0, 1.0, one, 4, None) == -1   # assert(sat >= hi > lo)
    with pytest.raises(RuntimeError):
        _lib.check(-1, 'unit test')


def test_ops_refuse_cpu_tensors():
    import torch
    from solex_ser_recon_en_amd import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.accumulate_sum_max(torch.zeros((2, 4, 4), dtype=torch.uint8))


def test_header_compiles_as_plain_c(tmp_path):
    """include/shg_hip.h is the drop-in boundary: it must be valid C99 (no C++, no torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = os.path.join(os.path.dirname(__file__), 'c_abi', 'header_is_c.c')
    inc = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'include')
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', inc, '-c', src, '-o', str(tmp_path / 'h.o')],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_caller_builds_against_the_library(tmp_path):
    """tests/c_abi/abi_smoke.cpp (a torch-free, Python-free caller) compiles and links against libshg_hip.so here;
    it RUNS in the gpu test of the same name."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(__file__))
    lib_dir = os.path.join(root, 'solex_ser_recon_en_amd', 'csrc')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-I', os.path.join(root, 'include'),
                        os.path.join(root, 'tests', 'c_abi', 'abi_smoke.cpp'), '-L', lib_dir, '-lshg_hip',
                        '-Wl,-rpath,' + lib_dir, '-o', str(tmp_path / 'abi_smoke')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
