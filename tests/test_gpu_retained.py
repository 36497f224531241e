"""Evaluators that can be called again after a solve without solving again, on batches of several windows
(include/rtd.h: rtd_plan_create_retained) -- the contract of the reference's closures, which keep GC_collect, K_collect,
B_collect and evaluate any (tau, phi) from them (_assemble_intensity_and_fluxes.py:170-262).

* a retained windowed plan returns, bit for bit, what the plain windowed plan (which solves its windows again for every
  evaluation) and the one-window plan return -- general points, antiderivatives, Nakajima-Tanaka corrections, thermal +
  BDRF columns at 64 streams, the exported tensors GC, K, B of a column of a later window;
* solving again after an evaluation (new inputs on the same plan) is ordered behind it;
* the cost: on a 20 000-column cfg4 batch (79 windows) a second call of `sol.u` costs a small fraction of the solve;
* round 6, the LEAN retained form (coefficients, k, E, B for every column; Y, A recomputed per evaluation for the layers the points
  touch, in the solve's wavefront composition; never the boundary-condition solve): the same bits as the full form on every kernel
  family from 10 streams up, and on BASELINE's literal 10^5-column batch -- where the full form would need 310 GB -- `sol.u` at one
  depth per column for well under 0.3 of a solve."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]


def _points(cfg, rng, n=5):
    C = cfg["tau_arr"].shape[0]
    top = cfg["tau_arr"][:, -1:]
    return np.sort(rng.uniform(0.0, 1.0, (C, n)), axis=1) * top


@pytest.mark.parametrize("maker,kw,cols,win", [("cfg4_columns", dict(L=6, NQuad=32), 11, 3), ("cfg3_columns", dict(big=True), 9, 4),
                                               ("cfg3_columns", dict(big=False), 20, 8), ("cfg5_columns", dict(L=7, NQuad=64), 5, 2),
                                               ("cfg4_columns", dict(L=3, NQuad=96), 3, 1), ("cfg4_columns", dict(L=20, NQuad=32), 7, 3),
                                               ("cfg5_columns", dict(L=11, NQuad=48), 5, 2)])
def test_retained_windowed_plan_equals_resolving_and_one_window_plans(maker, kw, cols, win):
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    cfg = getattr(synthetic, maker)(cols, **kw)
    rng = np.random.default_rng(11)
    tau, phi = _points(cfg, rng), np.array([0.0, 0.7, 3.0])
    _, one = amd.pydisort_batch(**cfg)
    _, again = amd.pydisort_batch(work_columns=win, retain=False, **cfg)
    _, kept = amd.pydisort_batch(work_columns=win, retain=True, **cfg)
    assert one.plan.retained() and kept.plan.retained() and not again.plan.retained()
    assert (one.plan.retained_form(), kept.plan.retained_form(), again.plan.retained_form()) == ("full", "full", None)
    assert kept.plan.windows()[1] > 1 and again.plan.windows() == kept.plan.windows()
    # the lean form: from 10 streams up (below, the eigen kernel's wavefronts span columns and the matrices are 128 bytes)
    _, lean = amd.pydisort_batch(work_columns=win, retain="lean", **cfg)
    assert lean.plan.retained_form() == ("lean" if cfg["NQuad"] > 8 else None) and lean.plan.windows() == kept.plan.windows()
    # (what the lean form saves shows on real batches -- 54 GB instead of 310 GB for 10^5 cfg4 columns, tested below --, not on these:
    #  two hand-off slots of a 3-column window are as large as 7 retained columns, and pooled blocks are larger than asked for)
    for anti in (False, True):
        for pts in (tau, tau[:, 2:3], tau[:, :2]):  # many points (every chunk recomputed), one and two per column (chunk lists)
            assert np.array_equal(lean.u(pts, phi, anti), one.u(pts, phi, anti))
            assert np.array_equal(lean.u0(pts, anti), one.u0(pts, anti))
            assert np.array_equal(lean.flux_up(pts, anti), one.flux_up(pts, anti))
        assert np.array_equal(lean.u(tau[:, 2:3], phi, anti), one.u(tau[:, 2:3], phi, anti))  # and again
    for k in ("GC", "K", "B", "G"):
        assert np.array_equal(lean.plan.tensors(cols - 1)[k], one.plan.tensors(cols - 1)[k]), k
    assert np.array_equal(lean.u(tau, phi), one.u(tau, phi))  # (the tensors pass left the retained state alone)
    lean.plan.close()
    for anti in (False, True):
        want = one.u(tau, phi, anti)
        assert np.array_equal(again.u(tau, phi, anti), want)
        assert np.array_equal(kept.u(tau, phi, anti), want)
        assert np.array_equal(kept.u(tau, phi, anti), want)       # and again: nothing was consumed by the first evaluation
        assert np.array_equal(kept.u0(tau, anti), one.u0(tau, anti))
        assert np.array_equal(kept.flux_up(tau, anti), one.flux_up(tau, anti))
        for a, b in zip(kept.flux_down(tau, anti), one.flux_down(tau, anti)):
            assert np.array_equal(a, b)
    # other points, after the first ones
    tau2 = _points(cfg, rng, 3)
    assert np.array_equal(kept.u(tau2, phi[:1]), one.u(tau2, phi[:1]))
    # the reference's tensors of a column of the LAST window, from the retained state
    c = cols - 1
    a, b = kept.plan.tensors(c), one.plan.tensors(c)
    for k in ("GC", "K", "B", "G"):
        assert np.array_equal(a[k], b[k]), k
    for s in (one, again, kept):
        s.plan.close()


def test_retained_plan_with_nt_corrections_and_new_inputs():
    """Nakajima-Tanaka corrections on a retained plan; then new columns on the same plan: the next solve (eigen stage on the second
    stream) must wait for the evaluation pass that still reads the retained arrays, and its results are the new columns'."""
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    cfg = synthetic.cfg4_columns(10, L=5, NQuad=16)
    rng = np.random.default_rng(5)
    tau, phi = _points(cfg, rng), np.array([0.3, 2.0])
    _, one = amd.pydisort_batch(NT_cor=True, **cfg)
    _, kept = amd.pydisort_batch(NT_cor=True, work_columns=3, retain=True, **cfg)
    assert kept.plan.retained() and kept.plan.windows()[1] == 4
    want = one.u(tau, phi)
    assert np.array_equal(kept.u(tau, phi), want) and np.array_equal(kept.u(tau, phi), want)
    one.plan.close()
    kept.plan.close()
    # same plan, new inputs, evaluation -> solve -> evaluation back to back
    a, b = synthetic.cfg4_columns(12, L=6, NQuad=32), synthetic.cfg4_columns(12, first=100, L=6, NQuad=32)
    _, solb = amd.pydisort_batch(**b)
    _, kept = amd.pydisort_batch(work_columns=4, retain=True, **a)
    tb = _points(b, rng)
    for _ in range(3):
        kept.u(_points(a, rng), phi)                       # evaluation passes on the plan's stream ... then the other batch through the same plan
    _, fresh = amd.pydisort_batch(work_columns=4, retain=True, _defer_solve=True, **b)
    kept.plan.set_columns(fresh.plan.prep)
    kept.plan.solve()
    got = kept.plan.evaluate(tb, phi, want=("u",))["u"]
    assert np.array_equal(got, solb.u(tb, phi))
    for s in (solb, kept, fresh):
        s.plan.close()


def test_lean_plan_with_nt_corrections_new_inputs_and_invalidated_tables():
    """The lean form through the same life cycle: Nakajima-Tanaka corrections, evaluations, then new columns on the same plan and
    a solve behind the evaluation passes (the next solve's eigen stream must wait for the pass that still reads hand-off slot 0),
    and an evaluation after rtd_plan_invalidate_tables (the tables at -mu0 are rebuilt before the eigen stage runs again)."""
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    rng = np.random.default_rng(6)
    phi = np.array([0.3, 2.0])
    cfg = synthetic.cfg4_columns(10, L=9, NQuad=16)
    tau = _points(cfg, rng, 1)
    _, one = amd.pydisort_batch(NT_cor=True, **cfg)
    _, lean = amd.pydisort_batch(NT_cor=True, work_columns=3, retain="lean", **cfg)
    assert lean.plan.retained_form() == "lean" and lean.plan.windows()[1] == 4
    want = one.u(tau, phi)
    assert np.array_equal(lean.u(tau, phi), want) and np.array_equal(lean.u(tau, phi), want)
    lean.plan.invalidate_tables()
    assert np.array_equal(lean.u(tau, phi), want)
    one.plan.close()
    lean.plan.close()
    a, b = synthetic.cfg4_columns(12, L=12, NQuad=32), synthetic.cfg4_columns(12, first=100, L=12, NQuad=32)
    _, solb = amd.pydisort_batch(**b)
    _, lean = amd.pydisort_batch(work_columns=4, retain="lean", **a)
    tb = _points(b, rng, 1)
    for _ in range(3):
        lean.u(_points(a, rng, 1), phi)
    _, fresh = amd.pydisort_batch(work_columns=4, retain="lean", _defer_solve=True, **b)
    lean.plan.set_columns(fresh.plan.prep)
    lean.plan.solve()
    assert np.array_equal(lean.plan.evaluate(tb, phi, want=("u",))["u"], solb.u(tb, phi))
    lean.plan.set_eval_points(np.concatenate((np.zeros((12, 1)), b["tau_arr"]), axis=1), phi)
    lean.plan.run()                      # the throughput form on a lean plan: solve + fused interface evaluation, pipelined
    solb.plan.set_eval_points(np.concatenate((np.zeros((12, 1)), b["tau_arr"]), axis=1), phi)
    solb.plan.run()
    assert np.array_equal(lean.plan.fetch()["u"], solb.plan.fetch()["u"])
    assert np.array_equal(lean.plan.evaluate(tb, phi, want=("u",))["u"], solb.u(tb, phi))
    for s in (solb, lean, fresh):
        s.plan.close()


def test_auto_retention_takes_the_lean_form_when_the_full_one_does_not_fit():
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    cfg = synthetic.cfg4_columns(40, L=20, NQuad=32)
    full_bytes = 40 * 3.2e6
    _, sol = amd.pydisort_batch(work_columns=8, retain="auto", retain_bytes=int(0.5 * full_bytes), **cfg)
    assert sol.plan.retained_form() == "lean"
    _, full = amd.pydisort_batch(work_columns=8, retain="auto", retain_bytes=int(2 * full_bytes), **cfg)
    assert full.plan.retained_form() == "full"
    _, none = amd.pydisort_batch(work_columns=8, retain="full", retain_bytes=int(0.5 * full_bytes), **cfg)
    assert none.plan.retained_form() is None
    tau = _points(cfg, np.random.default_rng(4), 1)
    want = full.u(tau, np.array([1.0]))
    assert np.array_equal(sol.u(tau, np.array([1.0])), want) and np.array_equal(none.u(tau, np.array([1.0])), want)
    with pytest.raises(ValueError):
        amd.pydisort_batch(retain="everything", **cfg)
    for s in (sol, full, none):
        s.plan.close()


def test_a_batch_that_does_not_fit_the_budget_is_a_plain_windowed_plan():
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    cfg = synthetic.cfg4_columns(8, L=5, NQuad=16)
    _, sol = amd.pydisort_batch(work_columns=3, retain=4096, **cfg)   # 4 KB: nothing fits
    _, one = amd.pydisort_batch(**cfg)
    assert not sol.plan.retained()
    tau = _points(cfg, np.random.default_rng(2))
    assert np.array_equal(sol.u(tau, np.array([1.0])), one.u(tau, np.array([1.0])))
    sol.plan.close()
    one.plan.close()


def test_second_evaluation_of_a_20000_column_batch_costs_an_evaluation_not_a_solve():
    """round-4 verdict, item 6: `sol.u` twice on a 20 000-column cfg4 batch.  The evaluator state of all columns (62 GB) is
    retained; the second call is an evaluation pass (one rtd_eval_kernel launch per window), not 79 windows solved again."""
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    C = 20_000
    cfg = synthetic.cfg4_columns_block(C, first=0)
    tau = np.stack([0.3 * cfg["tau_arr"][:, -1], 0.8 * cfg["tau_arr"][:, -1]], axis=1)
    phi = np.array([0.5])
    amd.pydisort_batch(**{k: (v[:512] if isinstance(v, np.ndarray) else v) for k, v in cfg.items()})[1].plan.close()  # warm the library
    t0 = time.perf_counter()
    _, sol = amd.pydisort_batch(work_columns=256, retain_bytes=80 << 30, **cfg)  # (the default budget is 16 GiB: the caller gives 80)
    first = sol.u(tau, phi)
    t_one = time.perf_counter() - t0
    assert sol.plan.retained_form() == "full" and sol.plan.windows()[1] == 79
    t0 = time.perf_counter()
    second = sol.u(tau, phi)
    t_second = time.perf_counter() - t0
    t0 = time.perf_counter()
    third = sol.u(0.5 * tau, phi)
    t_third = time.perf_counter() - t0
    assert np.array_equal(first, second) and np.all(np.isfinite(third))
    # the same batch without retention: every call solves the 79 windows again
    _, plain = amd.pydisort_batch(work_columns=256, retain=False, **cfg)
    plain.u(tau, phi)
    t0 = time.perf_counter()
    again = plain.u(tau, phi)
    t_again = time.perf_counter() - t0
    assert np.array_equal(again, first)
    print(f"\nretained 20 000-column batch: solve + first u {t_one:.3f} s, second u {t_second:.3f} s, u at new points {t_third:.3f} s; "
          f"without retention a second u costs {t_again:.3f} s")
    assert t_second < 0.2 * t_one and t_third < 0.2 * t_one       # two calls < 1.2 x one
    assert t_second < 0.5 * t_again
    sol.plan.close()
    plain.plan.close()


def test_lean_retention_on_the_literal_100000_column_batch():
    """round-5 verdict, item 6: BASELINE's literal batch -- 10^5 columns x 20 layers x 32 streams x 32 modes -- with evaluators that
    outlive the solve.  The full form would hold 3.1 MB per column (310 GB); the lean form holds 0.5 MB (50 GB): `sol.u(tau, phi)`
    at one non-interface depth per column then costs the eigen stage of ONE wavefront chunk per (column, mode) (4 of the 20
    layers) and an evaluation pass -- measured against the solve of the same batch: < 0.3 -- and returns, bit for bit, what the
    full-retention form returns (checked on two 2 048-column slices of the batch held in full-form plans of their own)."""
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    C = 100_000
    cfg = synthetic.cfg4_columns_block(C, first=0)
    rng = np.random.default_rng(17)
    tau = rng.uniform(0.02, 0.98, (C, 1)) * cfg["tau_arr"][:, -1:]
    phi = np.array([0.5, 2.0])
    amd.pydisort_batch(**{k: (v[:512] if isinstance(v, np.ndarray) else v) for k, v in cfg.items()})[1].plan.close()  # warm the library
    _, sol = amd.pydisort_batch(work_columns=256, retain="auto", retain_bytes=60 << 30, **cfg)  # (solves)
    assert sol.plan.retained_form() == "lean" and sol.plan.windows() == (256, 391)
    held = sol.plan.device_bytes()
    assert held < 75e9
    sol.plan.synchronize()
    t0 = time.perf_counter()
    sol.plan.solve()                      # the solve of the batch on its own (inputs resident): what an evaluation is measured against
    sol.plan.synchronize()
    t_solve = time.perf_counter() - t0
    first = sol.u(tau, phi)               # (first call: grows the evaluation buffers)
    t0 = time.perf_counter()
    second = sol.u(tau, phi)
    t_eval = time.perf_counter() - t0
    tau2 = rng.uniform(0.02, 0.98, (C, 1)) * cfg["tau_arr"][:, -1:]
    t0 = time.perf_counter()
    third = sol.u(tau2, phi)
    t_new = time.perf_counter() - t0
    assert np.array_equal(first, second) and np.all(np.isfinite(third))
    print(f"\nlean 10^5-column batch: {held / 1e9:.1f} GB on the device; solve {t_solve:.3f} s; u at one depth per column {t_eval:.3f} s "
          f"({t_eval / t_solve:.2f} of a solve), at new depths {t_new:.3f} s")
    assert t_eval < 0.3 * t_solve and t_new < 0.3 * t_solve
    for lo in (0, 70_000):
        sub = {k: (v[lo:lo + 2048] if isinstance(v, np.ndarray) else v) for k, v in cfg.items()}
        _, full = amd.pydisort_batch(work_columns=256, retain="full", retain_bytes=8 << 30, **sub)
        assert full.plan.retained_form() == "full"
        assert np.array_equal(full.u(tau[lo:lo + 2048], phi), first[lo:lo + 2048])
        assert np.array_equal(full.u(tau2[lo:lo + 2048], phi), third[lo:lo + 2048])
        full.plan.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "lean_retention_100000.txt"), "w") as f:
        f.write(f"lean retained plan, 100000 cfg4 columns (391 windows of 256): {held / 1e9:.1f} GB on the device; solve {t_solve:.4f} s; "
                f"sol.u at one non-interface depth per column, 2 azimuths: {t_eval:.4f} s = {t_eval / t_solve:.3f} of a solve; at new depths "
                f"{t_new:.4f} s; bit-identical to full-retention plans of columns [0, 2048) and [70000, 72048)\n")
    sol.plan.close()
