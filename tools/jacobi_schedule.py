#!/usr/bin/env python3
"""Replays the pair-layout Jacobi schedule of csrc/rtd_eig.hip (struct JSched) on the host and checks that every one of
the NP (NP - 1) / 2 column pairs meets exactly once per sweep, for consecutive sweeps and for any starting arrangement,
and that every move is one of the lane permutations a single DPP instruction expresses on gfx950.

State: NP/2 pair slots, each holding a column X and a column Y.  A step rotates (X, Y) of every slot; then the slots
with (p & sw) swap the roles of X and Y (folded into the rotation on the device) and Y moves to slot p ^ mk."""
import random

DPP_XOR_MASKS = {1, 2, 3, 7, 8, 15}  # quad_perm x3, row_half_mirror, row_ror:8, row_mirror


def build(NP):
    """(sw, mk) per step -- the same construction as `JSched<NP>`."""
    LPP = NP // 2
    sw, mk = [], []
    g = LPP
    while g >= 1:
        for b in range(g):
            if b < g - 1:
                low = (b + 1) & -(b + 1)
                sw.append(0)
                mk.append(7 if low == 4 else low)
            elif g > 1:
                h = g >> 1
                sw.append(h)
                mk.append(7 if h == 4 else h)
            else:
                sw.append(0)
                mk.append(LPP - 1 if LPP > 1 else 0)
        g >>= 1
    return sw, mk


def replay(NP, nsweeps=3, start=None):
    """Runs `nsweeps` sweeps; raises AssertionError if a pair meets twice in a sweep or is missed.  Returns the
    arrangement (X, Y) after the last sweep."""
    LPP = NP // 2
    sw, mk = build(NP)
    assert len(sw) == NP - 1
    X, Y = (list(range(LPP)), list(range(LPP, NP))) if start is None else (list(start[0]), list(start[1]))
    for _ in range(nsweeps):
        met = set()
        for sb, mu in zip(sw, mk):
            for p in range(LPP):
                pair = frozenset((X[p], Y[p]))
                assert pair not in met, (NP, sb, mu, sorted(pair))
                met.add(pair)
            for p in range(LPP):
                if sb and (p & sb):
                    X[p], Y[p] = Y[p], X[p]
            Y = [Y[p ^ mu] for p in range(LPP)]
        assert len(met) == NP * (NP - 1) // 2
    return X, Y


if __name__ == "__main__":
    for NP in (4, 8, 16, 32, 64):
        sw, mk = build(NP)
        # (NP = 64, the untuned 128-stream instance: masks 16 and 31 go through ds_swizzle, all stay inside 32 lanes)
        assert set(mk) <= DPP_XOR_MASKS | {0} | ({16, 31} if NP == 64 else set()), (NP, set(mk))
        replay(NP)
        for _ in range(5):
            perm = list(range(NP))
            random.shuffle(perm)
            replay(NP, 4, (perm[: NP // 2], perm[NP // 2:]))
        print(f"NP = {NP:2d}: {NP - 1} steps, masks {sorted(set(mk))}, every pair once per sweep: ok")
