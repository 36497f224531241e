#!/usr/bin/env python3
"""NumPy experiment (CPU), round 6: would WARM-STARTING the eigen kernel's one-sided Jacobi iteration from a neighbouring Fourier
mode pay?  The problems of one (column, layer) at modes m and m + 1 / m + 2 share their table rows and differ by one or two
rank-1 terms of the phase matrix (_solve_for_gen_and_part_sols.py:123-135); F_m V_{m+s}, with V the right singular vectors of
the neighbour, starts closer to orthogonal columns than F_m does.  Replay with the kernel's stop rule (tools/jacobi_convergence.py):

    cfg4 (32 streams): 3.19 sweeps cold, 2.82 from m + 1, 2.98 from m + 2
    cfg5 (64 streams): 2.60 sweeps cold, 2.25 from m + 1, 2.38 from m + 2

A third of a sweep (11-13 %) -- against one more N^3 product per problem, the accumulation of V (which the kernel does not form:
it keeps k Z = F V only) and a work mapping that walks the modes of a (column, layer) one after the other instead of handing out
204 800 independent wavefronts per window.  Not built.  Usage: python3 tools/jacobi_warm_start.py   (about 2 minutes)"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pythonic-disort_amd'), os.path.join(ROOT, 'tools')]
from jacobi_convergence import F_of_column, sweep_maxima
from pydisort_amd import synthetic
import warnings; warnings.simplefilter("ignore")
for name, cfg, NQ in (("cfg4", synthetic.cfg4_columns(3), 32), ("cfg5", synthetic.cfg5_columns(1, L=12), 64)):
    N = NQ//2
    res = {}
    for i in range(cfg["tau_arr"].shape[0]):
        kw = synthetic.column_kwargs(cfg, i)
        kw.pop("BDRF_Fourier_modes", None); kw.pop("s_poly_coeffs", None); kw.pop("b_pos", None)
        F, p = F_of_column(kw)            # [M, L, N, N]
        M, L = F.shape[:2]
        cold = sweep_maxima(F.reshape(-1, N, N), 9)
        cold_sw = (np.argmax(cold <= 1e-14, axis=0) + 1).reshape(M, L)
        for step in (1, 2):
            warm_sw = np.zeros((M, L))
            for m in range(M):
                for l in range(L):
                    if m + step < M:
                        V = np.linalg.svd(F[m + step, l])[2].T
                        W0 = F[m, l] @ V
                    else:
                        W0 = F[m, l]
                    T = sweep_maxima(W0[None], 9)
                    warm_sw[m, l] = np.argmax(T[:, 0] <= 1e-14) + 1
            res.setdefault(step, []).append(warm_sw.mean())
        res.setdefault(0, []).append(cold_sw.mean())
        # also: warm start from the adjacent layer of the same mode (sorted by omega)
    print(name, "cold sweeps", np.mean(res[0]), "warm from m+1", np.mean(res[1]), "warm from m+2", np.mean(res[2]))
