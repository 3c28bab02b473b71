"""CPU-only: the C-ABI shared library builds (cross-compiled for gfx950), loads,
and exports every symbol include/luminoth_hip.h declares; the ctypes binding
table covers exactly that set.  No kernel is launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'luminoth_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(lmh_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_symbols():
    syms = header_symbols()
    assert 'lmh_conv2d_fwd' in syms and 'lmh_rpn_proposal' in syms and len(syms) >= 25


def test_library_exports_every_declared_symbol():
    from luminoth_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), 'libluminoth_hip.so lacks %s' % s
    assert sorted(_lib.SIGNATURES) == header_symbols()
    loaded = _lib.load()
    assert loaded.lmh_version() == 101
    assert loaded.lmh_last_error() is not None


def test_product_path_refuses_cpu_tensors():
    import torch
    from luminoth_amd import _lib, kernels
    with pytest.raises(_lib.LuminothHipError):
        kernels.softmax(torch.zeros(4, 3))


def test_host_io_library_exports_every_declared_symbol():
    """include/luminoth_io.h <-> libluminoth_io.so (host C: TFRecord framing + CRC-32C)."""
    from luminoth_amd import _lib
    from luminoth_amd.datasets import tfrecord
    src = open(os.path.join(ROOT, 'include', 'luminoth_io.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    syms = sorted(set(re.findall(r'\b(lmh_io_[a-z0-9_]+)\s*\(', src)))
    assert len(syms) == 6, syms
    io_path = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libluminoth_io.so')
    if not os.path.exists(io_path):
        _lib.build()
    lib = ctypes.CDLL(io_path)
    for s in syms:
        assert hasattr(lib, s), 'libluminoth_io.so lacks %s' % s
    assert tfrecord.crc32c(b'123456789') == 0xE3069283


def test_tuning_options_registry_and_no_getenv_in_the_library():
    """the shipped library read ~10 LMH_* environment variables (one of them produced wrong results).  Now
    it reads none: options go through lmh_set_option, unknown names are an error, defaults are restored here."""
    import glob
    from luminoth_amd import _lib
    lib = _lib.load()
    v = ctypes.c_int(0)
    assert lib.lmh_get_option(b'wg_slots', ctypes.byref(v)) == 0 and v.value == 512
    assert lib.lmh_set_option(b'wg_slots', 256) == 0
    assert lib.lmh_get_option(b'wg_slots', ctypes.byref(v)) == 0 and v.value == 256
    assert lib.lmh_set_option(b'wg_slots', 512) == 0
    assert lib.lmh_set_option(b'nms_dbg', 1) != 0 and b'unknown option' in lib.lmh_last_error()
    for f in glob.glob(os.path.join(ROOT, 'luminoth_amd', 'csrc', '*.hip')) + \
            glob.glob(os.path.join(ROOT, 'luminoth_amd', 'csrc', '*.h')):
        assert 'getenv' not in open(f).read(), f
