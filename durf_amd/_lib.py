"""ctypes binding of libdurf_hip.so (C ABI in include/durf_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared
object is missing this module raises at import of any op (run `python -c "import
__graft_entry__ as g; g.build()"` or `make -C durf_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DURF_LIB_PATH: kernel-tuning A/B builds (tools/build_variant.sh); the default is the in-tree build
LIB_PATH = os.environ.get('DURF_LIB_PATH') or os.path.join(_HERE, 'libdurf_hip.so')

_lib = None

from ._sigs import SIGS as _SIGS       # generated from include/durf_hip.h (tools/gen_integration_stub.py)


def symbols():
    """Every symbol include/durf_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libdurf_hip.so not built (%s): the HIP extension is required, there is no '
                'fallback path.  Run __graft_entry__.build().' % LIB_PATH)
        # PyTorch first: the process must hold ONE HIP runtime -- the copy torch brings.  Loaded before torch, this library
        # would bind /opt/rocm's libamdhip64 and torch would then load its own next to it; launches from here then fail with
        # "no ROCm-capable device is detected" (seen with __graft_entry__.build() followed by smoke() in one process).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class DurfError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise DurfError('%s failed (%d): %s' % (what, rc, lib().durf_last_error().decode()))
