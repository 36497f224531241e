"""One rank of the multi-GPU parity tests (tests/test_gpu_multi_gpu.py): a process of its own per GPU, RCCL through librtd,
the unique id handed over through a file (no torch).  Not a test module.

  python tests/dist_worker.py --mode allgather|root|modes|layers --rank R --world W --dir D [--device X]

Every rank writes D/result_<R>.json: {"ok": bool, "checks": {...}}.  What is compared (the reference has no multi-process
form; the partitions are SURVEY 8(e)'s -- independent loops _solve_for_gen_and_part_sols.py:88-91, _solve_for_coeffs.py:110-111,
meeting point _assemble_intensity_and_fluxes.py:256-260):
  allgather / root : W column shards -> ncclAllGather / ncclSend+ncclRecv -> the gathered arrays against ONE plan's solve
                     of the union of the columns on this rank's GPU, bit for bit (every rank / the root);
  modes            : W Fourier-mode shards of the same columns -> ncclAllReduce -> the unsharded solve (1e-13);
  layers           : W layer shards of the eigen stage -> ONE ncclAllGather stitches the boundary-condition system ->
                     boundary-condition solve on every rank -> the one-piece solve (1e-11).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]


def exchange_id(a, Plan, tag="0"):
    """A fresh RCCL unique id per communicator (an id serves one ncclCommInitRank round), published by rank 0."""
    path = os.path.join(a.dir, f"uid_{tag}.bin")
    if a.rank == 0:
        uid = Plan.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise RuntimeError("rank 0 never published the RCCL unique id")
        time.sleep(0.02)
    with open(path, "rb") as f:
        return f.read()


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=("allgather", "root", "modes", "layers"))
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--device", type=int, default=None)
    a = ap.parse_args()
    import pydisort_amd
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    dev = a.rank if a.device is None else a.device
    W, R = a.world, a.rank
    Plan.comm_preload()
    checks = {}
    ok = True
    phi = np.array([0.0, 0.9, 2.5])
    if a.mode in ("allgather", "root"):
        per = 7  # columns per rank: 32 streams (the fused kernel, several windows of 3) and 8 streams (row-per-lane kernels)
        for name, kw, win in (("q32", dict(L=6, NQuad=32), 3), ("q8", dict(L=5, NQuad=8), 0)):
            whole = synthetic.cfg4_columns(per * W, **kw)
            mine = synthetic.cfg4_columns(per, first=per * R, **kw)
            _, sol = pydisort_amd.pydisort_batch(device=dev, work_columns=win, _defer_solve=True, **mine)
            plan = sol.plan
            plan.set_eval_points(np.concatenate((np.zeros((per, 1)), mine["tau_arr"]), axis=1), phi)
            plan.comm_init(exchange_id(a, Plan, name), R, W)  # one communicator per plan (the library's model)
            checks[name + "_rccl_says"] = list(plan.comm_size())  # ncclCommCount / ncclCommUserRank / ncclCommCuDevice
            checks["transport"] = plan.comm_transport()
            ok = ok and plan.comm_size() == (W, R, dev)
            for _ in range(2):  # the second step's gather overlaps nothing stale: its snapshot waits for the first gather
                plan.run()
                plan.allgather_results() if a.mode == "allgather" else plan.gather_results(0)
            plan.run()  # a further step in flight beside the collective
            plan.synchronize()
            holds = a.mode == "allgather" or R == 0
            if holds:
                _, ref = pydisort_amd.pydisort_batch(device=dev, _defer_solve=True, **whole)
                ref.plan.set_eval_points(np.concatenate((np.zeros((per * W, 1)), whole["tau_arr"]), axis=1), phi)
                ref.plan.run()
                want = ref.plan.fetch()
                ref.plan.close()
                gu, gf = plan.fetch_gathered_results()
                fl = np.concatenate([gf[r] for r in range(W)], axis=1)  # [3][W * per][ntau]
                same = (np.array_equal(gu, want["u"]) and np.array_equal(fl[0], want["flux_up"])
                        and np.array_equal(fl[1], want["flux_down_diffuse"]) and np.array_equal(fl[2], want["flux_down_direct"]))
                su, sf = plan.fetch_gathered_columns(W - 1, per - 2, 2)  # the slice form, last rank's last columns
                same_slice = np.array_equal(su, want["u"][-2:]) and np.array_equal(sf[0], want["flux_up"][-2:])
                checks[name] = {"bit_equal": bool(same), "slice_bit_equal": bool(same_slice), "max_rel": rel(gu, want["u"])}
                ok = ok and same and same_slice
            else:
                try:  # a non-root rank holds no gathered arrays: must be refused, not answered with stale data
                    plan.fetch_gathered_results()
                    checks[name] = {"non_root_fetch_refused": False}
                    ok = False
                except RuntimeError:
                    checks[name] = {"non_root_fetch_refused": True}
            plan.close()
    elif a.mode == "modes":
        for name, cfg in (("cfg3", synthetic.cfg3_columns(3, big=True)), ("cfg4", synthetic.cfg4_columns(2))):
            C = cfg["tau_arr"].shape[0]
            tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
            _, full = pydisort_amd.pydisort_batch(device=dev, _defer_solve=True, **cfg)
            full.plan.set_eval_points(tau, phi)
            full.plan.run()
            want = full.plan.fetch()
            full.plan.close()
            _, part = pydisort_amd.pydisort_batch(device=dev, mode_shard=(R, W), _defer_solve=True, **cfg)
            part.plan.set_eval_points(tau, phi)
            part.plan.comm_init(exchange_id(a, Plan, name), R, W)
            checks["transport"] = part.plan.comm_transport()
            part.plan.run()
            part.plan.allreduce_results()
            got = part.plan.fetch()
            part.plan.close()
            errs = {k: rel(got[k], want[k]) for k in ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct")}
            checks[name] = errs
            ok = ok and all(v <= 1e-13 for v in errs.values())
    else:  # layers
        uid = exchange_id(a, Plan)
        Ltot = 4 * W
        cfg = synthetic.cfg4_columns(3, L=Ltot)
        cfg.update(s_poly_coeffs=np.tile(np.array([[0.3, 0.02]]), (3, Ltot, 1)), b_pos=0.1)  # thermal vectors travel too
        tau = np.concatenate((np.zeros((3, 1)), cfg["tau_arr"], 0.5 * cfg["tau_arr"][:, :1]), axis=1)
        _, sol = pydisort_amd.pydisort_batch(device=dev, **cfg)
        want = sol.plan.evaluate(tau, phi)
        sol.plan.close()
        _, sh = pydisort_amd.pydisort_batch(device=dev, _defer_solve=True, **cfg)
        plan = sh.plan
        plan.comm_init(uid, R, W)
        checks["transport"] = plan.comm_transport()
        cnt = Ltot // W
        plan.solve_layers(R * cnt, cnt)   # this rank's layers only: the others hold nothing until the gather
        plan.allgather_layers(cnt)        # ONE collective stitches the boundary-condition system
        plan.solve_bc()
        got = plan.evaluate(tau, phi)
        plan.close()
        errs = {k: rel(got[k], want[k]) for k in ("u", "u0", "flux_up", "flux_down_diffuse")}
        checks["layers"] = errs
        ok = ok and all(v <= (0.0 if W == 1 else 1e-11) for v in errs.values())
    with open(os.path.join(a.dir, f"result_{R}.json"), "w") as f:
        json.dump({"ok": bool(ok), "checks": checks}, f)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
