"""CPU: the C-ABI library loads and exports every symbol include/sc_hip.h declares; with no
GPU the compute entries fail loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name='sc_hip.h'):
    text = open(os.path.join(ROOT, 'include', name)).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sc_hip_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from libsafecrypto_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(engine.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/sc_hip.h but not exported'
    assert sorted(engine.SYMBOLS) == syms          # the Python binding covers the whole header
    bsyms = _header_symbols('sc_hip_bench.h')
    assert sorted(engine.BENCH_SYMBOLS) == bsyms
    for s in bsyms:
        assert hasattr(lib, s), f'{s} declared in include/sc_hip_bench.h but not exported'
    vsyms = [s for s in _header_symbols('sc_hip_vtable.h') if s not in ('sc_hip_ntt_params', 'sc_hip_ntt32_vtable')]
    assert sorted(engine.VTABLE_SYMBOLS) == vsyms
    for s in vsyms:
        assert hasattr(lib, s), f'{s} declared in include/sc_hip_vtable.h but not exported'


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from libsafecrypto_amd import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(0)


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under libsafecrypto_amd/ or include/ may name it"""
    bad = []
    for base in ('libsafecrypto_amd', 'include'):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith(('.py', '.h', '.hip', '.cpp', '.c', 'Makefile')):
                    t = open(os.path.join(dp, f), errors='ignore').read()
                    if re.search(r'sc_oracle|oracle/|libsc_oracle|orc_', t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
