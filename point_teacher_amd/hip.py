"""ctypes binding of libpt_hip.so - the ONLY place the native library is loaded.

The prototypes are read from include/pt_hip.h, so the Python side can never drift from
the C ABI.  There is no CPU fallback: if the library is missing, importing this module
raises, and every wrapper refuses non-CUDA tensors.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpt_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'pt_hip.h')

_CT = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'int64_t': ctypes.c_int64, 'double': ctypes.c_double,
    'void': None,
}


PTR_DTYPES = {}      # {fn: {argname: C element type}} filled by parse_header


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [(ctype, argname), ...])} for every function the header declares."""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    protos = {}
    for m in re.finditer(r'(const\s+char\s*\*|int64_t|int)\s+(pt_\w+)\s*\(([^)]*)\)\s*;', txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if 'char' in ret else (ctypes.c_int64 if ret == 'int64_t' else ctypes.c_int)
        argl = []
        for a in [x.strip() for x in args.split(',') if x.strip() and x.strip() != 'void']:
            if '*' in a:
                base = a.split('*')[0].replace('const', '').strip()
                PTR_DTYPES.setdefault(name, {})[a.split('*')[-1].strip()] = base
                argl.append((ctypes.c_void_p, a.split('*')[-1].strip()))
            else:
                typ, an = a.rsplit(None, 1)
                argl.append((_CT[typ.replace('const', '').strip()], an))
        protos[name] = (restype, argl)
    return protos


def parse_structs(path=HEADER_PATH):
    """-> {name: ctypes.Structure subclass} for every `typedef struct { ... } name;` of the header (host-side descriptors
    such as pt_conv_desc: plain scalars and device pointers)."""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    scalars = {'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'int': ctypes.c_int, 'double': ctypes.c_double}
    out = {}
    for m in re.finditer(r'typedef\s+struct\s*\{([^}]*)\}\s*(pt_\w+)\s*;', txt):
        fields = []
        for decl in [d.strip() for d in m.group(1).split(';') if d.strip()]:
            if '*' in decl:
                fields.append((decl.split('*')[-1].strip(), ctypes.c_void_p))
            else:
                typ, names = decl.split(None, 1)
                for n in names.split(','):
                    fields.append((n.strip(), scalars[typ]))
        out[m.group(2)] = type(m.group(2), (ctypes.Structure,), {'_fields_': fields})
    return out


PROTOS = parse_header()
STRUCTS = parse_structs()

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f'{LIB_PATH} not found: build it with `python -m point_teacher_amd.build` '
        '(hipcc --offload-arch=gfx950).  point_teacher_amd has no CPU fallback.')
_lib = ctypes.CDLL(LIB_PATH)
for _n, (_r, _a) in PROTOS.items():
    _f = getattr(_lib, _n)          # AttributeError here == header/library mismatch
    _f.restype = _r
    _f.argtypes = [t for t, _ in _a]

ABI_VERSION = _lib.pt_abi_version()


def header_constant(name):
    """A numeric `#define` of include/pt_hip.h (one source for constants the host side and the kernels share)."""
    m = re.search(r'^#define\s+' + name + r'\s+([0-9.eE+-]+)f?\b', open(HEADER_PATH).read(), re.M)
    if not m:
        raise KeyError(f'{name} is not defined in {HEADER_PATH}')
    return float(m.group(1))


def last_error():
    return _lib.pt_last_error().decode()


_DT = {'float': torch.float32, 'int32_t': torch.int32, 'uint8_t': torch.uint8, 'uint64_t': torch.int64, 'int64_t': torch.int64, 'double': torch.float64,
       'uint16_t': torch.bfloat16}      # uint16_t* in the header = bf16 storage


def _ptr(x, name, ctype=None):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError(f'{name}: expected a CUDA/HIP tensor (point_teacher_amd has no CPU path)')
        if not x.is_contiguous():
            raise RuntimeError(f'{name}: tensor must be contiguous')
        want = _DT.get(ctype)
        if want is not None and x.dtype != want:      # a wrong element size would read out of bounds
            raise TypeError(f'{name}: expected {want}, got {x.dtype}')
        return x.data_ptr()
    if isinstance(x, (ctypes.Array,)):
        return ctypes.cast(x, ctypes.c_void_p)
    if isinstance(x, ctypes.Structure):
        return ctypes.addressof(x)
    if isinstance(x, int):
        return x
    raise TypeError(f'{name}: cannot pass {type(x)} as a pointer')


_NO_RC_CHECK = ('pt_abi_version', 'pt_focal_nblocks', 'pt_sqnorm_nblocks', 'pt_affine_train_rows', 'pt_gemm_bf16x6_tile_rows',
                'pt_conv3x3_wgrad_bf16x6_splits', 'pt_conv_wgrad_bf16x6_splits', 'pt_conv_bf16x6_splits', 'pt_conv_bf16x6_plan')
_PLANS = {}


def _plan(fn):
    """Per entry point, once: the ctypes function, which arguments are pointers (with the element type the header declares), whether a
    trailing `stream` may be filled in, whether the return code is an error code."""
    restype, proto = PROTOS[fn]
    dts = PTR_DTYPES.get(fn, {})
    ptrs = tuple((f'{fn}.{an}', dts.get(an)) if ct is ctypes.c_void_p else None for ct, an in proto)
    pl = _PLANS[fn] = (getattr(_lib, fn), ptrs, len(proto), proto[-1][1] == 'stream' if proto else False,
                       restype is ctypes.c_int and fn not in _NO_RC_CHECK)
    return pl


def current_stream_handle():
    """torch's current HIP stream on the current device as a raw handle (what every libpt_hip launch goes to)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def call(fn, *args):
    """Call a libpt_hip entry point.  Tensors become device pointers; the trailing `stream`
    argument is filled with torch's current stream when omitted."""
    pl = _PLANS.get(fn) or _plan(fn)
    cf, ptrs, n, has_stream, check = pl
    auto_stream = len(args) != n
    if auto_stream and not (has_stream and len(args) == n - 1):
        raise TypeError(f'{fn}: expected {n} arguments, got {len(args)}')
    conv = [a if (p is None or a is None) else _ptr(a, p[0], p[1]) for a, p in zip(args, ptrs)]      # (a CPU tensor fails here, loudly)
    if auto_stream:
        conv.append(current_stream_handle())
    rc = cf(*conv)
    if check and rc != 0:
        raise RuntimeError(f'{fn} failed (code {rc}): {last_error()}')
    return rc


def dptr(t, ctype=None):
    """Device pointer of a tensor (or 0 for None) for a descriptor field, with the checks `call` applies to pointer arguments."""
    return _ptr(t, 'descriptor field', ctype) or 0


def host_doubles(vals):
    arr = (ctypes.c_double * max(len(vals), 1))()
    for i, v in enumerate(vals):
        arr[i] = float(v)
    return arr


def host_floats(vals):
    arr = (ctypes.c_float * max(len(vals), 1))()
    for i, v in enumerate(vals):
        arr[i] = float(v)
    return arr
