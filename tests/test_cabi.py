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
    assert hip.ABI_VERSION == 7


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


def _integration_stub():
    """The python code block of INTEGRATION.md section 1, pointed at the built library."""
    from point_teacher_amd import hip
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', md, flags=re.S)
    assert len(blocks) == 1
    return blocks[0].replace('/path/to/point_teacher_amd/libpt_hip.so', hip.LIB_PATH)


def test_integration_stub_matches_header():
    """Every `argtypes` list the document tells a maintainer to write has the header's arity and kinds, and every
    call in the stub passes that many arguments."""
    from point_teacher_amd import hip
    src = _integration_stub()
    ns = {}
    exec(compile(src, 'INTEGRATION.md', 'exec'), ns)             # defines the wrappers; nothing is launched
    lib = ns['_lib']
    bound = re.findall(r'_lib\.(pt_\w+)\.argtypes', src)
    assert set(bound) >= {'pt_topk_assign', 'pt_roi_align_fwd'}
    for name in bound:
        want = [t for t, _ in hip.PROTOS[name][1]]
        got = list(getattr(lib, name).argtypes)
        assert len(got) == len(want), (name, len(got), len(want))
        for g, w in zip(got, want):
            assert g is w, (name, g, w)
        call = re.search(r'_lib\.' + name + r'\((.*?)\),\s*\n?\s*\'' + name, src, flags=re.S).group(1)
        depth, nargs = 0, 1
        for ch in call:
            depth += ch in '([' 
            depth -= ch in ')]'
            nargs += (ch == ',' and depth == 0)
        assert nargs == len(want), (name, nargs, len(want))


@pytest.mark.gpu
def test_integration_stub_runs_on_gpu():
    """The documented binding, executed as written, equals the package's own wrappers."""
    import torch
    from point_teacher_amd import functional as F
    ns = {}
    exec(compile(_integration_stub(), 'INTEGRATION.md', 'exec'), ns)
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(20.), torch.arange(20.), indexing='ij')
    pts = (torch.stack((xs.reshape(-1) * 8, ys.reshape(-1) * 8), -1) + 4).to(dev)
    gt = (torch.rand(11, 2, generator=g) * 150 + 3.3).to(dev)
    off, _ = F.make_offsets([6, 5], dev)
    a = ns['topk_assign'](pts, gt, off, 2, 3)
    b = F.topk_assign(pts, gt, off, 2, 3)
    assert torch.equal(a.reshape(-1), b.reshape(-1).to(a.dtype))
    feat = torch.randn(2, 16, 20, 20, generator=g).to(dev)
    rois = torch.tensor([[0, 8., 8., 40., 56.], [1, 20.3, 11.1, 23.9, 14.2], [1, 1., 1., 150., 150.]], device=dev)
    o = ns['roi_align_forward'](feat, rois, 7, 0.125)
    torch.testing.assert_close(o, F.roi_align(feat, rois, 7, 0.125), rtol=1e-6, atol=1e-6)


def test_documented_build_command_works_from_a_clean_copy(tmp_path):
    """`python -m point_teacher_amd.build` - the command the missing-library error prints - on a copy of the sources WITHOUT
    libpt_hip.so and without object files (round-3 verdict P6): the package import must not need the library it is about to
    build; any other import of the package without the library still fails loudly."""
    import shutil
    import sys
    src = os.path.join(ROOT, 'point_teacher_amd')
    dst = tmp_path / 'point_teacher_amd'
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns('*.so', 'build', '__pycache__', '*.o'))
    shutil.copytree(os.path.join(ROOT, 'include'), tmp_path / 'include')
    # only the two translation units that share no heavy templates are compiled here (the full build is __graft_entry__.build());
    # the build module takes its source list from a variable this test narrows through the environment
    env = dict(os.environ, PT_BUILD_SOURCES='glue.hip,image_prep.hip')
    r = subprocess.run([sys.executable, '-m', 'point_teacher_amd.build'], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (dst / 'libpt_hip.so').exists()
    shutil.rmtree(dst / 'build')
    os.remove(dst / 'libpt_hip.so')
    r = subprocess.run([sys.executable, '-c', 'import point_teacher_amd'], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'python -m point_teacher_amd.build' in r.stderr


def test_graft_entry_build_accepts_the_current_abi():
    """`__graft_entry__.build()` is the driver's build check: its ABI assertion must follow the library's version."""
    import re
    src = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    m = re.search(r'assert hip\.ABI_VERSION == (\d+)', src)
    from point_teacher_amd import hip
    assert m and int(m.group(1)) == hip.ABI_VERSION
