"""Evaluators that can be called again after a solve without solving again, on batches of several windows
(include/rtd.h: rtd_plan_create_retained) -- the contract of the reference's closures, which keep GC_collect, K_collect,
B_collect and evaluate any (tau, phi) from them (_assemble_intensity_and_fluxes.py:170-262).

* a retained windowed plan returns, bit for bit, what the plain windowed plan (which solves its windows again for every
  evaluation) and the one-window plan return -- general points, antiderivatives, Nakajima-Tanaka corrections, thermal +
  BDRF columns at 64 streams, the exported tensors GC, K, B of a column of a later window;
* solving again after an evaluation (new inputs on the same plan) is ordered behind it;
* the cost: on a 20 000-column cfg4 batch (79 windows) a second call of `sol.u` costs a small fraction of the solve."""
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
                                               ("cfg4_columns", dict(L=3, NQuad=96), 3, 1)])
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
    assert kept.plan.windows()[1] > 1 and again.plan.windows() == kept.plan.windows()
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
    _, sol = amd.pydisort_batch(work_columns=256, **cfg)
    first = sol.u(tau, phi)
    t_one = time.perf_counter() - t0
    assert sol.plan.retained() and sol.plan.windows()[1] == 79
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
