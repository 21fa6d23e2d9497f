"""ctypes binding of libcfn_hip.so (the gfx950 kernels; C ABI declared in include/cfn_hip.h).

The product path has no CPU fallback: if the library is missing, or a tensor is not a contiguous
fp32 CUDA(HIP) tensor, the ops raise.  Prototypes are generated from the header so that the header
stays the single source of truth for the ABI.
"""
import ctypes
import threading
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB_PATH = os.environ.get('CFN_HIP_LIB') or os.path.join(_HERE, 'libcfn_hip.so')     # CFN_HIP_LIB: a variant build for a same-box A/B (tools/variant_lib.sh)
HEADER = os.path.join(_ROOT, 'include', 'cfn_hip.h')

ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID = 0, 1, 2, 3
FAMILIES = {'dwconv_fwd': 0, 'dwconv_bwd': 1, 'pwconv_fwd': 2, 'pwconv_bwd': 3, 'gridpool': 4, 'elementwise': 5,
            'stem': 6, 'fusion': 7, 'pwconv_wgrad': 8, 'dwconv_wgrad': 9, 'dense_fwd': 10, 'gridpool_bwd': 11}

_lib = None
_protos = None


def header_prototypes(path=HEADER):
    """{name: (restype, [argtype, ...])} parsed from the C header."""
    with open(path) as fh:
        txt = fh.read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for m in re.finditer(r'(const\s+char\s*\*|int|long)\s+(cfn_\w+)\s*\(([^)]*)\)\s*;', txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        at, dt = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                dt.append(None)
                if '*' in a:
                    at.append(ctypes.c_void_p)
                    base = a.replace('const', '').split('*')[0].strip()
                    dt[-1] = {'float': torch.float32, 'double': torch.float64, 'long': torch.int64, 'int': torch.int32,
                               'unsigned short': torch.bfloat16}.get(base)
                elif a.startswith('long'):
                    at.append(ctypes.c_long)
                elif a.startswith('double'):
                    at.append(ctypes.c_double)
                elif a.startswith('int'):
                    at.append(ctypes.c_int)
                else:
                    raise ValueError('unhandled C type in %s: %r' % (name, a))
        if name.endswith('_f16'):                # the fp16 twins of the bf16 entry points: `unsigned short*` = IEEE half there
            dt = [torch.float16 if d == torch.bfloat16 else d for d in dt]
        out[name] = (ctypes.c_char_p if 'char' in ret else (ctypes.c_long if ret == 'long' else ctypes.c_int), at, dt)
    return out


def load():
    """Load the library (once) and attach the prototypes; raises if anything is missing."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libcfn_hip.so not found at %s -- run `python __graft_entry__.py` (hipcc, gfx950) first; '
                           'there is no CPU fallback for the Coarse-Fine HIP ops' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    protos = header_prototypes()
    for name, (ret, at, _dt) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError('libcfn_hip.so does not export %s declared in include/cfn_hip.h' % name)
        fn.restype = ret
        fn.argtypes = at
    _lib, _protos = lib, protos
    if os.environ.get('CFN_DETERMINISTIC', '0') not in ('', '0') and torch.cuda.is_available():
        if lib.cfn_deterministic(1) < 0:                  # include/cfn_hip.h: order-independent commits of every cross-workgroup accumulation
            raise RuntimeError(lib.cfn_last_error().decode())
        _det_on[0] = True
    return lib


# Deterministic mode records every accumulation of an entry point into ONE buffer per process and commits it (device synchronisation, sort) before the
# entry point returns: two threads inside entry points at the same time would commit each other's half-written records (round 6: the two-thread model
# test aborted under CFN_DETERMINISTIC=1).  While the mode is on, callers of this binding are serialised -- the mode synchronises the device per entry
# point anyway.  (A C++ caller of the ABI has to do the same.)
_det_on = [False]
_det_lock = threading.Lock()


def deterministic(on=None):
    """switch / query the library's deterministic mode (include/cfn_hip.h cfn_deterministic); returns the previous setting"""
    with _det_lock:                          # (not in the middle of another thread's entry point)
        rc = load().cfn_deterministic(-1 if on is None else (1 if on else 0))
        if rc < 0:
            raise RuntimeError(last_error())
        if on is not None:
            _det_on[0] = bool(on)
    return bool(rc)


def last_error():
    return load().cfn_last_error().decode()


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('cfn_hip ops need device tensors (got %s); there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('cfn_hip ops need contiguous tensors')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def _marshal(name, args):
    """tensors -> device pointers; checks element types against the header and that all tensors share ONE device.
    Returns (converted args, that device or None)."""
    dts = _protos[name][2]
    dev = None
    conv = []
    for i, a in enumerate(args):
        if isinstance(a, torch.Tensor):
            if dts[i] is not None and a.dtype != dts[i]:   # element types are part of the ABI: refuse a mismatch
                raise RuntimeError('%s: argument %d must be a %s tensor, got %s' % (name, i, dts[i], a.dtype))
            if a.device.type != 'cuda':             # a host pointer handed to a kernel is a device fault, not an exception
                raise RuntimeError('%s: argument %d is a %s tensor; the Coarse-Fine HIP ops run on the GPU only (no CPU fallback): '
                                   'move the module and its inputs to a cuda device' % (name, i, a.device.type))
            if dev is None:
                dev = a.device
            elif a.device != dev:
                raise RuntimeError('%s: tensors on different devices (%s and %s)' % (name, dev, a.device))
            conv.append(_ptr(a))
        else:
            conv.append(a)
    return conv, dev


def _launch(name, args):
    """Enqueue one entry point on the CURRENT stream of the device that owns the tensors (not of whatever device
    happens to be current: the reference's caller is N threads in one process, one per device)."""
    lib = load()
    conv, dev = _marshal(name, args)
    fn = getattr(lib, name)
    if _det_on[0]:
        with _det_lock:
            return _launch_now(fn, conv, dev)
    return _launch_now(fn, conv, dev)


def _launch_now(fn, conv, dev):
    if dev is None or dev.index == torch.cuda.current_device():
        return fn(*conv, torch.cuda.current_stream().cuda_stream)
    with torch.cuda.device(dev):
        return fn(*conv, torch.cuda.current_stream(dev).cuda_stream)


def call(name, *args):
    """Invoke one C entry point: tensors -> device pointers, appends the tensors' device's current HIP stream."""
    rc = _launch(name, args)
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, last_error()))


def call_try(name, *args):
    """like call(), for entry points that may decline a geometry: returns False (nothing was launched) on -1"""
    rc = _launch(name, args)
    if rc > 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, last_error()))
    return rc == 0


def query(name, *args):
    """host-only helper entry points (no stream argument), e.g. workspace sizes"""
    return getattr(load(), name)(*args)


def check(t, dtype=torch.float32):
    """activation tensors are fp32, bf16 or fp16 (the *_bf16 / *_f16 entry points); anything else is refused"""
    if t is not None and t.dtype != dtype and not (dtype == torch.float32 and t.dtype in (torch.bfloat16, torch.float16)):
        raise RuntimeError('expected %s tensor, got %s' % (dtype, t.dtype))
    return t


def prof_enable(family, on=True):
    rc = load().cfn_prof_enable(FAMILIES[family], 1 if on else 0)
    if rc:
        raise RuntimeError(last_error())


def prof_collect(family):
    ms, n, by = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    rc = load().cfn_prof_collect(FAMILIES[family], ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
    if rc:
        raise RuntimeError(last_error())
    return ms.value, n.value, by.value


def device_info():
    cus, lds = ctypes.c_int(), ctypes.c_int()
    name = ctypes.create_string_buffer(64)
    rc = load().cfn_device_info(ctypes.byref(cus), ctypes.byref(lds), name, 64)
    if rc:
        raise RuntimeError(last_error())
    return {'cus': cus.value, 'lds_per_cu': lds.value, 'arch': name.value.decode()}
