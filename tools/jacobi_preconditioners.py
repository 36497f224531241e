#!/usr/bin/env python3
"""NumPy replay (CPU): does any preconditioning of the one-sided Jacobi iteration of the eigen kernel save sweeps?  DESIGN.md 7(c).
Usage: python3 tools/jacobi_preconditioners.py"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pythonic-disort_amd'), os.path.join(ROOT, 'tools')]
from jacobi_convergence import F_of_column, sweep_maxima
from pydisort_amd import synthetic
C=6
cfg=synthetic.cfg4_columns(C)
F=np.concatenate([F_of_column(synthetic.column_kwargs(cfg,i))[0].reshape(-1,16,16) for i in range(C)])
def sweeps(W):
    T=sweep_maxima(W,12); return (np.argmax(T<=1e-14,axis=0)+1)
def report(name,W):
    s=sweeps(W); s3=s.reshape(C,32,20)
    print(f"{name:40s} mean {s.mean():.2f}  m=0: {s3[:,0].mean():.2f} m=8: {s3[:,8].mean():.2f} m=31: {s3[:,31].mean():.2f}")
report("F = L^T R (kernel)",F)
report("F^T",np.swapaxes(F,1,2))
H=F@np.swapaxes(F,1,2)
Cc=np.linalg.cholesky(H)
report("chol(F F^T) lower",Cc)
report("chol(F F^T)^T upper",np.swapaxes(Cc,1,2))
H2=np.swapaxes(F,1,2)@F
C2=np.linalg.cholesky(H2)
report("chol(F^T F) lower",C2)
report("chol(F^T F)^T",np.swapaxes(C2,1,2))
nrm=np.sum(F*F,1)
idx=np.argsort(-nrm,axis=1)
Fs=np.take_along_axis(F,idx[:,None,:],axis=2)
report("F, columns sorted by norm descending",Fs)
idx=np.argsort(nrm,axis=1)
report("F, columns sorted ascending",np.take_along_axis(F,idx[:,None,:],axis=2))
# QR with column pivoting then transpose: X = R1^T where F P = Q R1
import scipy.linalg as sl
X=np.empty_like(F)
for i in range(len(F)):
    q,r,p=sl.qr(F[i],pivoting=True)
    X[i]=r.T
report("R1^T from pivoted QR of F (Drmac-Veselic)",X)
Xt=np.empty_like(F)
for i in range(len(F)):
    q,r,p=sl.qr(F[i].T,pivoting=True)
    Xt[i]=r.T
report("R1^T from pivoted QR of F^T",Xt)
