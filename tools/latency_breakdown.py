import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import goldens, pydisort_amd
from pydisort_amd import synthetic, _engine, _prepare
warnings.simplefilter("ignore")
kw = goldens.load("9c")[0]["kwargs"]
pydisort_amd.pydisort(**kw)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    res = pydisort_amd.pydisort(**kw)
    res[1](0.5)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
