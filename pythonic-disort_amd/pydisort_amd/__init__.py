"""pydisort_amd -- MI355X-native discrete-ordinate radiative-transfer solver.

Drop-in for the hot path of LDEO-CREW/Pythonic-DISORT: ``pydisort`` keeps the reference's signature and
returned callables; ``pydisort_batch`` solves many independent atmospheric columns per call.  All
numerics run in hand-written HIP kernels (librtd.so) reached through the C ABI of include/rtd.h.
"""
from .pydisort import pydisort  # noqa: F401
from .batch import pydisort_batch, BatchSolution, solve_columns_streamed  # noqa: F401
from . import subroutines  # noqa: F401
from ._engine import Plan as _Plan

pool_bytes, pool_trim = _Plan.pool_bytes, _Plan.pool_trim  # device memory of closed plans kept for the next one (include/rtd.h)

__all__ = ["pydisort", "pydisort_batch", "BatchSolution", "solve_columns_streamed", "subroutines", "pool_bytes", "pool_trim"]
