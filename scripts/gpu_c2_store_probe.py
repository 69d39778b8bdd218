#!/usr/bin/env python3
"""C2 (OU GuidedBridge, 65 536 fresh proposals): the kernel with and without the store of the paths -- is it the stores that bind?"""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, bench, bridgehip as bh
ctx = bh.Context(0)
Po = bench._ou(ctx)
n = 65536
def t(fn, k=200):
    for _ in range(50): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
import ctypes as C
X = bh.EnsemblePath(Po.tt, 1, n, ctx); ll = ctx.empty(n)
x0 = (C.c_double * 1)(0.5)
def run(store):
    ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, x0, None, None, C.c_long(0), C.c_void_p(X.data.data_ptr()) if store else None, C.c_long(X.ld), C.c_void_p(ll.data_ptr()), 0, C.c_long(n), C.c_uint64(1), 0, 0))
try:
    print("with X store   ", t(lambda: run(True)))
    print("without X store", t(lambda: run(False)))
except Exception as e:
    print("direct call failed:", e)
    print("with X store   ", t(lambda: bh.sample_solve((0.5,), Po, n, seed=1, store_X=True)))
    print("without X store", t(lambda: bh.sample_solve((0.5,), Po, n, seed=1, store_X=False)))
