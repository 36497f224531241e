#!/usr/bin/env python3
"""Does the eigen kernel of one plan overlap the boundary-condition kernel of another on two streams?
(eigen: VALU-bound; BC: latency-bound -- co-resident wavefronts of both kinds could fill each other's bubbles.)
Measured (2 048 columns each): eigen alone 4.82-5.02 ms, BC alone 4.33-4.71 ms, one after the other 9.32-9.75 ms, on two
streams 9.21-9.35 ms: the dispatcher drains one grid before the other gets wavefronts.  A single launch whose blocks
alternate between the two stages (five eigen blocks per BC chain, LDS as a union, 158 VGPRs: a round-2 experiment, not
kept) DOES mix them on every SIMD -- and takes 11.26 ms: the two hot loops together (~60 KB of code) no longer fit the
64 KB instruction cache a pair of CUs shares.
Usage (GPU box): python tools/stage_overlap.py [columns]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
plans = []
for k in range(2):
    cfg = synthetic.cfg4_columns(C)
    _, sol = pydisort_amd.pydisort_batch(**cfg, _defer_solve=True)
    plans.append(sol.plan)
A, B = plans
L = 20
for p in plans:
    p.solve_layers(0, L); p.solve_bc(); p.synchronize()

def timed(fn, n=5):
    fn(); A.synchronize(); B.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    A.synchronize(); B.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def eig_only():
    A.solve_layers(0, L); A.synchronize()
def bc_only():
    B.solve_bc(); B.synchronize()
def both_serial():
    A.solve_layers(0, L); A.synchronize(); B.solve_bc(); B.synchronize()
def both_parallel():
    A.solve_layers(0, L); B.solve_bc(); A.synchronize(); B.synchronize()
def both_parallel_bc_first():
    B.solve_bc(); A.solve_layers(0, L); A.synchronize(); B.synchronize()
for name, fn in (("eigen alone", eig_only), ("BC alone", bc_only), ("serial", both_serial), ("two streams", both_parallel),
                 ("two streams, BC first", both_parallel_bc_first)):
    print(f"{name:24s} {timed(fn):7.3f} ms", flush=True)

