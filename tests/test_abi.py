"""The C-ABI library loads and exports exactly the symbols include/rewriting_hip.h declares
(no compute calls: there is no GPU in the CPU test environment)."""
import ctypes
import os
import re

from rewriting_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'rewriting_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rw_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_what_python_binds():
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_is_built_and_exports_every_symbol():
    assert os.path.isfile(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().rw_abi_version() == _lib.ABI_VERSION
    assert b'success' in _lib.load().rw_error_string(0)


def test_every_header_entry_cites_the_reference():
    text = open(os.path.join(ROOT, 'include', 'rewriting_hip.h')).read()
    assert text.count('.py:') + text.count('.cu:') + text.count('.cpp:') >= 20


def test_struct_layouts_match_c():
    # sizes computed by hand from include/rewriting_hip.h (LP64): 5 pointers + int (+pad)
    assert ctypes.sizeof(_lib.ConvEpilogue) == 48
    # 5 ints (+4 pad) + 20 pointers + int + 4 floats + int + 2 floats + int(+0 pad) + ptr + int(+4) + ptr
    assert ctypes.sizeof(_lib.SolveProblem) == 24 + 160 + 32 + 8 + 8 + 8 + 8 - 0


def test_missing_gpu_tensor_fails_loudly():
    import pytest
    import torch
    from rewriting_amd import hip
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip.pixel_norm(torch.zeros(2, 8))
