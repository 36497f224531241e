"""pydisort_amd -- MI355X-native discrete-ordinate radiative-transfer solver.

Drop-in for the hot path of LDEO-CREW/Pythonic-DISORT: ``pydisort`` keeps the reference's signature and
returned callables; ``pydisort_batch`` solves many independent atmospheric columns per call.  All
numerics run in hand-written HIP kernels (librtd.so) reached through the C ABI of include/rtd.h.
"""
from .pydisort import pydisort  # noqa: F401
from .batch import pydisort_batch, BatchSolution, solve_columns_streamed  # noqa: F401
from . import subroutines  # noqa: F401
from ._engine import Plan as _Plan

import contextlib as _contextlib

# Device memory of closed plans (include/rtd.h: rtd_pool_*).  Large blocks are NOT kept by default: a loop of batch calls that
# creates and destroys plans of gigabytes opts in -- `with pydisort_amd.pooled(): ...` or pool_set_limit() -- and thereby avoids
# the runtime's seconds-long stalls on later allocations (profiles/r05_alloc_outliers.txt); a one-off call leaves nothing behind.
pool_bytes, pool_trim, pool_set_limit = _Plan.pool_bytes, _Plan.pool_trim, _Plan.pool_set_limit


@_contextlib.contextmanager
def pooled(nbytes=-1, device=0):
    """Within the block, closed plans' large device blocks (up to nbytes per device; default an eighth of the device's memory)
    are kept for the next plan of about the same size; on exit the previous limit is restored and what no longer fits is given
    back to the runtime."""
    prev = pool_set_limit(nbytes, device)
    try:
        yield
    finally:
        pool_set_limit(prev, device)


__all__ = ["pydisort", "pydisort_batch", "BatchSolution", "solve_columns_streamed", "subroutines", "pool_bytes", "pool_trim",
           "pool_set_limit", "pooled"]
