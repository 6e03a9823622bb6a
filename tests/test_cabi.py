"""CPU-side checks of the C-ABI boundary: the library builds/loads here (hipcc cross-compiles)
and exports exactly what include/pt_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from point_teacher_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    txt = open(hip.HEADER_PATH).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    declared = set(re.findall(r'\b(pt_\w+)\s*\(', txt))
    assert declared == set(hip.PROTOS), declared ^ set(hip.PROTOS)
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in pt_hip.h but not exported'
    out = subprocess.run(['nm', '-D', '--defined-only', hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (pt_\w+)', out))
    assert exported == declared, exported ^ declared
    assert hip.ABI_VERSION == 1


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing."""
    import torch
    from point_teacher_amd import functional as F
    with pytest.raises(RuntimeError, match='no CPU path'):
        F.bbox_overlaps(torch.zeros(2, 4), torch.zeros(2, 4), is_aligned=True)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'point_teacher_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'ref_ops' not in src, f
    for f in os.listdir(os.path.join(ROOT, 'tools')):            # tools are product-side too: only tests, smoke() and bench's cpu_baseline may
        if f.endswith('.py'):
            src = open(os.path.join(ROOT, 'tools', f)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
