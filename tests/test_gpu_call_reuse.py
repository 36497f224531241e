"""One-column pydisort() calls reuse idle device plans (pydisort_amd/pydisort.py: _plan_for / _release): a plan serves the next call
of the same shape once no closure of an earlier call refers to it -- never while one does.  The reference's closures stay valid for
as long as the caller keeps them (_assemble_intensity_and_fluxes.py:170-613 close over GC_collect, K_collect, B_collect)."""
import gc
import os
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import goldens  # noqa: E402


def _kw(scale=1.0):
    kw = dict(goldens.load("9c")[0]["kwargs"])
    kw["omega_arr"] = np.asarray(kw["omega_arr"]) * scale
    return kw


def test_closures_of_an_earlier_call_stay_valid_and_idle_plans_are_reused():
    import pydisort_amd as amd
    tau, phi = np.array([0.0, 0.7, 3.1]), np.array([0.0, 1.0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = amd.pydisort(**_kw(1.0))
        ua = a[4](tau, phi)
        b = amd.pydisort(**_kw(0.5))                       # a's closures are alive: b must get a plan of its own
        plan_a, plan_b = a[1].__self__.plan, b[1].__self__.plan
        assert plan_a is not plan_b
        assert np.array_equal(a[4](tau, phi), ua)           # ... and a's results are untouched by b's solve
        ub = b[4](tau, phi)
        assert not np.array_equal(ua, ub)
        del a
        gc.collect()
        c = amd.pydisort(**_kw(0.5))                       # a's plan is idle now: the same shape takes it
        assert c[1].__self__.plan is plan_a
        assert np.array_equal(c[4](tau, phi), ub)           # same inputs as b, solved on the reused plan: the same bits
        assert np.array_equal(b[4](tau, phi), ub)
        # Nakajima-Tanaka state does not leak from one tenant of a plan to the next
        kw5 = dict(goldens.load("5a")[0]["kwargs"])
        n1 = amd.pydisort(**kw5)
        un = n1[4](np.array([0.0, 1.0]), phi)
        p5 = n1[1].__self__.plan
        del n1
        gc.collect()
        kw5_plain = dict(kw5, NT_cor=False)
        n2 = amd.pydisort(**kw5_plain)
        assert n2[1].__self__.plan is p5
        fresh = amd.pydisort(**kw5_plain)                   # (n2 alive: a new plan)
        assert fresh[1].__self__.plan is not p5
        assert np.array_equal(n2[4](np.array([0.0, 1.0]), phi), fresh[4](np.array([0.0, 1.0]), phi))
        assert not np.array_equal(un, n2[4](np.array([0.0, 1.0]), phi))


def test_a_plan_closed_by_hand_is_not_handed_out_again():
    import pydisort_amd as amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = amd.pydisort(**_kw(0.9))
        plan = a[1].__self__.plan
        plan.close()
        del a
        gc.collect()
        b = amd.pydisort(**_kw(0.9))
        assert b[1].__self__.plan is not plan
        assert np.all(np.isfinite(b[1](np.array([0.5]))))


def test_device_memory_of_closed_batch_plans_serves_the_next_plan():
    """The device arena of a closed plan -- whatever a previous tenant left in it -- is what the next plan of the same shape is
    built in, ONCE THE CALLER HAS OPTED IN (include/rtd.h: rtd_pool_set_limit; pydisort_amd.pooled()): same bits as from fresh
    memory, for a retained windowed plan with thermal sources and a BDRF and for a plain one; rtd_pool_trim gives everything back.
    By default (round 6) a closed batch plan leaves nothing large behind: the library is a guest in someone else's process."""
    import pydisort_amd as amd
    from pydisort_amd import synthetic
    amd.pool_trim()
    assert amd.pool_bytes() == 0
    if "RTD_POOL_BYTES" not in os.environ:
        cfg = synthetic.cfg4_columns(300)
        _, sol = amd.pydisort_batch(work_columns=128, **cfg)
        assert sol.plan.device_bytes() > (64 << 20)
        sol.plan.close()
        assert amd.pool_bytes() <= (64 << 20)        # default: only small blocks (the evaluation buffers) stay
        amd.pool_trim()
    with amd.pooled():
        _pooled_plans_serve_the_next(amd, synthetic)
        assert amd.pool_bytes() > (64 << 20)
    if "RTD_POOL_BYTES" not in os.environ:
        assert amd.pool_bytes() <= (64 << 20)        # leaving the block restores the limit and frees what no longer fits
    amd.pool_trim()
    assert amd.pool_bytes() == 0


def _pooled_plans_serve_the_next(amd, synthetic):
    rng = np.random.default_rng(3)
    for maker, kw, cols, win in (("cfg5_columns", dict(L=12, NQuad=64), 24, 8), ("cfg4_columns", {}, 300, 128)):
        cfg = getattr(synthetic, maker)(cols, **kw)
        other = getattr(synthetic, maker)(cols, first=500, **kw)
        tau = np.sort(rng.uniform(0, 1, (cols, 4)), axis=1) * cfg["tau_arr"][:, -1:]
        tau_o = np.sort(rng.uniform(0, 1, (cols, 4)), axis=1) * other["tau_arr"][:, -1:]
        phi = np.array([0.2, 2.9])
        _, first = amd.pydisort_batch(work_columns=win, **cfg)           # fresh memory
        want = first.u(tau, phi), first.flux_up(tau)
        nbytes = first.plan.device_bytes()
        assert nbytes > (64 << 20) and first.plan.retained()
        first.plan.close()
        held = amd.pool_bytes()
        assert held >= 0.9 * nbytes                                       # the arena (and the evaluation buffers) are kept
        _, dirty = amd.pydisort_batch(work_columns=win, **other)         # a tenant that leaves other columns' state behind
        assert amd.pool_bytes() < held                                    # ... it was built in the kept memory
        dirty.u(tau_o, phi)
        dirty.plan.close()
        _, again = amd.pydisort_batch(work_columns=win, **cfg)
        got = again.u(tau, phi), again.flux_up(tau)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        again.plan.close()
