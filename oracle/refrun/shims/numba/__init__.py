"""Stand-in for `numba` used ONLY by oracle/refrun/make_golden.py in the build container.

numba is not installable here (no network; the conda copy is broken against NumPy 1.26).
The reference decorates plain-Python/NumPy functions with ``@nb.njit(...)``; executed
without the JIT they compute the same values (modulo fastmath reassociation), only slower.
This module makes every decorator the identity and ``prange`` a ``range``.
Test infrastructure — never imported by the product package.
"""
import types as _types


def _identity_decorator(*args, **kwargs):
    # @njit                      -> called with the function
    # @njit("sig", parallel=...) -> called with a signature, returns a decorator
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn
    return wrap


njit = _identity_decorator
jit = _identity_decorator
vectorize = _identity_decorator
prange = range

# `from numba import cuda` is executed at import time by the reference's cuda/ package.
cuda = _types.ModuleType("numba.cuda")


def _cuda_jit(*args, **kwargs):
    return _identity_decorator(*args, **kwargs)


def _unavailable(*a, **k):
    raise RuntimeError("numba.cuda stand-in: no CUDA device in the oracle harness")


cuda.jit = _cuda_jit
cuda.grid = _unavailable
cuda.synchronize = _unavailable
cuda.managed_array = _unavailable
cuda.is_available = lambda: False

import sys as _sys
_sys.modules["numba.cuda"] = cuda
