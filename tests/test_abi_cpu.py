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
    assert _lib.lib.shg_abi_version() == _lib.ABI_VERSION == 14
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
    # pixels (1000 x 1048 tiles: 31 rows a slice, 33 slices), chunk sums, clipped totals
    assert lib.shg_clahe_workspace_bytes_for(2000, 2096, 2, 2) == 4 * 65536 * 6 + 4 * 33 * 131072 + 4 * 1024 * 4 + 4 * 32 * 2 * 4
    assert lib.shg_clahe_workspace_bytes_for(77, 50, 4, 1) == lib.shg_clahe_workspace_bytes(4, 1)
    assert lib.shg_contrast_stats_workspace_bytes_for(2000, 2096, 2) == lib.shg_contrast_stats_workspace_bytes(2) + 4 * 33 * 131072 + 4 * 1024 * 4 + 4 * 32 * 2 * 4


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



def test_every_plain_launch_says_so():
    """csrc/launch.h: on a thread of a scan pool a kernel launch is RECORDED (SHG_LAUNCH) and joins the other scans' launches of the
    same kernel in one dispatch later; code that still launches the plain way on the scan's own stream must first hand over what has
    been recorded and make later recordings wait for that stream (SHG_DIRECT) -- or two kernels of one scan run out of order.  A
    launch that forgets is a race only the rare branch it sits in would show, so the sources are held to the rule here: every
    `<<<` and every hipMemset / hipMemcpy on a stream is preceded, since the last recorded launch of its function, by SHG_DIRECT."""
    import glob
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, '..', 'solex_ser_recon_en_amd', 'csrc')
    # launches that are not part of a scan's chain on its own stream: pass A on the frame-pass lane, the upload service's copies,
    # the bandwidth probes
    exempt = {'accumulate.hip': ('k_accumulate_vec', 'k_accumulate_scalar'), 'decode.hip': ('hipMemcpy2DAsync',), 'probe.hip': None, 'combine.hip': None}
    offenders = []
    for path in sorted(glob.glob(os.path.join(csrc, '*.hip'))):
        name = os.path.basename(path)
        if name in exempt and exempt[name] is None:
            continue
        state = 'start'                                      # since the function began: nothing yet / SHG_DIRECT seen / a recorded launch seen
        for n, line in enumerate(open(path).read().split('\n'), 1):
            code = line.split('//')[0]
            if line and not line[0].isspace() and line[0] not in '}#/' and code.rstrip().endswith('{'):
                state = 'start'
            if 'SHG_DIRECT(' in code:
                state = 'direct'
            elif 'SHG_LAUNCH' in code and '#define' not in code:
                state = 'recorded'
            plain = '<<<' in code or re.search(r'hipMem(set|cpy|cpy2D)Async\(', code)
            if plain and not any(tag in code for tag in exempt.get(name, ())):
                if state != 'direct':
                    offenders.append('%s:%d: %s' % (name, n, line.strip()[:100]))
    assert not offenders, '\n'.join(offenders)
