#!/usr/bin/env python3
"""Diagnostic: does a page-locked H2D copy overlap with compute on another stream on this box?"""
import time
import torch
n = 237 * 1024 * 1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
a = torch.randn(8192, 8192, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

def copy():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)

def compute():
    with torch.cuda.stream(s2):
        for _ in range(60):
            torch.sin_(a)

def both():
    copy(); compute()

tc, tk, tb = t(copy), t(compute), t(both)
print("H2D %d MB alone %.2f ms (%.1f GB/s); elementwise kernels alone %.2f ms; both on two streams %.2f ms (serial sum %.2f)" % (n >> 20, tc, n / tc / 1e6, tk, tb, tc + tk))

# host-side duration of the asynchronous copy call itself
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s1):
    d.copy_(h, non_blocking=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("pinned non_blocking copy: call returned after %.3f ms, done after %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
import ctypes, sys
sys.path.insert(0, "dense-visual-odometry_amd")
import numpy as np
from dvo_amd import _cabi
hip = ctypes.CDLL("libamdhip64.so")
arr = _cabi.pinned_array((n,), np.uint8)
arr[:] = 1
stream = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1)
for rep in range(3):
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    rc = hip.hipMemcpyAsync(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(n), 1, stream)
    t1 = time.perf_counter()
    hip.hipStreamSynchronize(stream)
    t2 = time.perf_counter()
    print("dvo_host_alloc memory, hipMemcpyAsync on a non-blocking stream: rc %d, call %.3f ms, done %.3f ms" % (rc, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
