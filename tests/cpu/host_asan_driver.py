#!/usr/bin/env python3
"""Driver of the host-only sanitizer build of librtd (tests/test_host_asan.py runs it under LD_PRELOAD=libasan with
RTD_LIB=tests/cpu/librtd_host_asan.so): the REAL Python front end and the REAL host code of csrc/rtd_api.hip -- plan arenas, window
offsets, hand-off slots, retained / lean forms, chunk lists, pinned staging slabs, the pool, evaluation-buffer growth, tensor
export -- over a fake HIP runtime whose "device" memory is heap memory and whose kernels are shadow launchers that touch the
extents the real kernels touch (tests/cpu/host_asan_shadow.cpp).  Values are constants; what is checked is that no call sequence
reads or writes outside what it allocated.  Prints one line per scenario."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd as amd  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402
from pydisort_amd._engine import Plan  # noqa: E402

rng = np.random.default_rng(1)
phi = np.array([0.0, 1.0, 2.5])


def pts(cfg, n):
    C = cfg["tau_arr"].shape[0]
    return np.sort(rng.uniform(0.0, 1.0, (C, n)), axis=1) * cfg["tau_arr"][:, -1:]


def exercise(sol, cfg, label, tensors=True):
    C = cfg["tau_arr"].shape[0]
    for n in (1, 2, 5):
        t = pts(cfg, n)
        for anti in (False, True):
            sol.u(t, phi, anti)
            sol.u0(t, anti)
            sol.flux_up(t, anti)
            sol.flux_down(t, anti)
    sol.u(pts(cfg, 9), phi[:1])                       # evaluation buffers grow
    if tensors:
        for c in {0, C - 1}:
            sol.plan.tensors(c)
    iface = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    sol.plan.set_eval_points(iface, phi)
    for _ in range(2):
        sol.plan.run()
    sol.plan.fetch()
    sol.plan.run_fetch()
    sol.plan.invalidate_tables()
    sol.plan.run()
    sol.plan.synchronize()
    sol.plan.column_status()
    sol.plan.max_sweeps()
    sol.plan.pivoted_chains()
    sol.plan.device_bytes()
    sol.u(pts(cfg, 1), phi)
    sol.plan.close()
    print("ok", label, flush=True)


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    # every padding of the per-hemisphere stream count (N -> NP in 4, 8, 16, 32, 64), one window
    for nq in (2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 100, 128):
        cfg = synthetic.cfg4_columns(5, L=3, NQuad=nq)
        exercise(amd.pydisort_batch(**cfg)[1], cfg, f"one window, NQuad {nq}")
    # windows: plain (solves again per evaluation), retained full, retained lean; pipelined and RTD_NO_PIPELINE
    for nopipe in (False, True):
        if nopipe:
            os.environ["RTD_NO_PIPELINE"] = "1"
        for maker, kw, cols, win in (("cfg4_columns", dict(L=20, NQuad=32), 11, 3), ("cfg3_columns", dict(big=True), 9, 4),
                                     ("cfg3_columns", dict(big=False), 20, 8), ("cfg5_columns", dict(L=7, NQuad=64), 5, 2),
                                     ("cfg4_columns", dict(L=3, NQuad=96), 4, 1), ("cfg5_columns", dict(L=11, NQuad=48), 7, 7)):
            cfg = getattr(synthetic, maker)(cols, **kw)
            for retain in (False, "full", "lean", "auto"):
                sol = amd.pydisort_batch(work_columns=win, retain=retain, **cfg)[1]
                exercise(sol, cfg, f"{maker} {kw} {cols} columns in windows of {win}, retain {retain}, no-pipeline {nopipe}")
        os.environ.pop("RTD_NO_PIPELINE", None)
    # Nakajima-Tanaka corrections, windowed and retained
    cfg = synthetic.cfg4_columns(7, L=5, NQuad=16)
    for retain in (False, "full", "lean"):
        exercise(amd.pydisort_batch(NT_cor=True, work_columns=3, retain=retain, **cfg)[1], cfg, f"NT corrections, retain {retain}")
    # raw inputs prepared on the device, the streamed host-to-host form, only_flux
    cfg = synthetic.cfg5_columns(9, L=6, NQuad=32)
    exercise(amd.pydisort_batch(device_prepare=True, work_columns=4, **cfg)[1], cfg, "device_prepare")
    tau = np.concatenate((np.zeros((9, 1)), cfg["tau_arr"]), axis=1)
    amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=4)
    amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=4, only_flux=True)
    out = amd.solve_columns_streamed(cfg, tau, phi)
    amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=2, out=out)
    print("ok streamed", flush=True)
    exercise(amd.pydisort_batch(only_flux=True, work_columns=4, **cfg)[1], cfg, "only_flux", tensors=False)
    # BDRF samples on the device, Fourier-mode shards, layer shards (eigen stage of some layers, then the boundary-condition solve)
    cfg = synthetic.cfg4_columns(4, L=8, NQuad=16)
    N = 8
    rho = rng.uniform(0.1, 0.3, (4, N, N, 12))
    exercise(amd.pydisort_batch(bdrf_samples=(rho, rng.uniform(0.1, 0.3, (4, N, 12))), NBDRF=3, **cfg)[1], cfg, "bdrf samples")
    for shard in ((0, 3), (2, 3)):
        exercise(amd.pydisort_batch(mode_shard=shard, **cfg)[1], cfg, f"mode shard {shard}")
    sol = amd.pydisort_batch(_defer_solve=True, **cfg)[1]
    for first, count in ((0, 3), (3, 5), (0, 8)):
        sol.plan.solve_layers(first, count)
    sol.plan.solve_bc()
    sol.plan.evaluate(pts(cfg, 2), phi)
    sol.plan.close()
    print("ok layer shards", flush=True)
    # a plan reused for other batches; one-column calls reusing idle plans
    cfg_a, cfg_b = synthetic.cfg4_columns(12, L=6, NQuad=32), synthetic.cfg4_columns(12, first=50, L=6, NQuad=32)
    sol = amd.pydisort_batch(work_columns=4, retain="lean", **cfg_a)[1]
    fresh = amd.pydisort_batch(work_columns=4, retain="lean", _defer_solve=True, **cfg_b)[1]
    sol.u(pts(cfg_a, 1), phi)
    sol.plan.set_columns(fresh.plan.prep)
    sol.plan.solve()
    sol.plan.evaluate(pts(cfg_b, 1), phi)
    sol.plan.close()
    fresh.plan.close()
    for k in range(6):
        kw = synthetic.column_kwargs(synthetic.cfg4_columns(1, first=k, L=4, NQuad=8 if k % 2 else 16), 0)
        res = amd.pydisort(**kw)
        res[4](np.array([0.1, 0.2]), phi)
        res[1](np.array([0.1]))
    print("ok plan reuse", flush=True)
    # the pool of large blocks: off by default, opted in, limit lowered, trimmed
    big = synthetic.cfg4_columns(40, L=20, NQuad=32)
    assert amd.pool_bytes() <= 600 << 20
    amd.pool_trim()
    s1 = amd.pydisort_batch(**big)[1]
    assert s1.plan.device_bytes() > 64 << 20
    s1.plan.close()
    assert amd.pool_bytes() <= 64 << 20, amd.pool_bytes()
    with amd.pooled(1 << 30):
        for _ in range(3):
            amd.pydisort_batch(**big)[1].plan.close()
            amd.pydisort_batch(work_columns=16, **big)[1].plan.close()
        assert amd.pool_bytes() > 64 << 20
        amd.pool_set_limit(100 << 20)
        assert amd.pool_bytes() <= (100 << 20) + (512 << 20)
    amd.pool_trim()
    assert amd.pool_bytes() == 0
    print("ok pool", flush=True)
print("ALL SCENARIOS PASSED")
