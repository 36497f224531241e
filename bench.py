#!/usr/bin/env python3
"""Headline benchmark: column-solves/sec for 20-layer, 32-stream Henyey-Greenstein atmospheres
(BASELINE.json configs[3], "cfg4" of SURVEY section 8(d)) on N GPUs of one node.

A step = one pass of the whole hot path (Legendre tables, eigen stage, boundary-condition solve,
evaluation of u at the 21 layer interfaces x 3 azimuths plus fluxes) over one batch of
`--columns` synthetic columns per GPU, inputs already resident in HBM.  Columns are independent, so
ranks shard them with no data-path exchange during the solve (weak scaling: per-GPU work fixed); one
RCCL all-gather of the flux results per step stitches the outputs.

Prints ONE JSON line on rank 0 (contract in the build prompt) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak (vendor figure)
L, NQUAD, NTAU, NPHI = 20, 32, 21, 3


def algorithmic_flops():
    """Per-column algorithmic FLOPs per kernel (F_col = L[2N^2 P(P+1) + 70.3 N^3 M] = 195 MFLOP
    for cfg4; per (m, l): assembly 4N^2(P-m), product 2N^3, eigen-decomposition 25N^3, U = (alpha+beta)V/k 2N^3,
    particular solve 5.33N^3, BC solve 36N^3 per layer)."""
    N, P, M = NQUAD // 2, NQUAD, NQUAD
    ml = L * M
    fl = dict(asm=L * 2 * N * N * P * (P + 1) + 2 * N**3 * ml, jacobi=25.0 * N**3 * ml, post=(2 + 5.33) * N**3 * ml,
              bc=36.0 * N**3 * ml, eval=NTAU * M * (2 * N) * (2 * N) * 2.0)
    fl["total"] = fl["asm"] + fl["jacobi"] + fl["post"] + fl["bc"]
    return fl


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json;
    FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as read), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def extra_measurements(device):
    """Secondary numbers of SURVEY section 8(d) on rank 0: max |dI| of the HIP path against the oracle on the sample
    columns of the cpu_baseline leg, and the only_flux (one Fourier mode) throughput."""
    import pydisort_amd
    from pydisort_amd import synthetic
    out = {}
    if _ORACLE_SAMPLES:
        worst_abs = worst_rel = 0.0
        phi = np.array([0.0, np.pi / 2, np.pi])
        for first, want in _ORACLE_SAMPLES:
            cfg = synthetic.cfg4_columns(1, first=first)
            _, sol = pydisort_amd.pydisort_batch(device=device, **cfg)
            tau = np.concatenate((np.zeros((1, 1)), cfg["tau_arr"]), axis=1)
            got = sol.u(tau, phi)[0]
            diff = np.abs(got - want)
            sig = np.abs(want) > 1e-8 * np.max(np.abs(want))
            worst_abs = max(worst_abs, float(diff.max()))
            worst_rel = max(worst_rel, float((diff[sig] / np.abs(want[sig])).max()))
            sol.plan.close()
        out["parity"] = {"max_abs_dI": worst_abs, "max_rel_dI": worst_rel, "columns_checked": len(_ORACLE_SAMPLES),
                         "against": "CPU oracle (pinned to the reference) on the same seeded cfg4 columns"}
    C = 16384
    cfg = synthetic.cfg4_columns(C, first=50_000)
    _, sol = pydisort_amd.pydisort_batch(only_flux=True, device=device, **cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0]))
    plan.run()
    plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        plan.run()
    plan.synchronize()
    out["only_flux"] = {"value": 5 * C / (time.perf_counter() - t0), "unit": "column-solves/sec",
                        "workload": "cfg4 with only_flux=True (one Fourier mode), 16384 columns per pass"}
    plan.close()
    return out


def shard_columns(rank, world, columns_per_gpu):
    """Weak-scaling column shard of a rank: global column indices [first, first + count)."""
    return rank * columns_per_gpu, columns_per_gpu


def reduce_max_seconds(dist, seconds):
    """Max over ranks of a host-side duration through the (gloo) control plane."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def _cpu_worker(args):
    first, n = args
    from threadpoolctl import threadpool_limits
    from oracle import disort_oracle as O
    from pydisort_amd import synthetic
    with threadpool_limits(1):
        cfg = synthetic.cfg4_columns(n, first=first)
        phi = np.array([0.0, np.pi / 2, np.pi])
        keep = None
        for i in range(n):
            res = O.pydisort(**synthetic.column_kwargs(cfg, i))
            tau = np.concatenate(([0.0], cfg["tau_arr"][i]))
            u = res[4](tau, phi)
            res[1](tau), res[2](tau)
            if i == 0:
                keep = u
    return n, first, keep


_ORACLE_SAMPLES = []


def cpu_baseline(cols_per_core=6):
    """Oracle (NumPy/SciPy port of the reference, same LAPACK calls) on every host core, 1 BLAS thread
    per process, same synthetic inputs; bounded sample.  Runs BEFORE the GPU is initialised (fork)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    ctx = mp.get_context("fork")
    jobs = [(10_000 + k * cols_per_core, cols_per_core) for k in range(cores)]
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(0, 1)] * cores)  # warm imports
        t0 = time.perf_counter()
        results = pool.map(_cpu_worker, jobs)
        dt = time.perf_counter() - t0
    done = sum(r[0] for r in results)
    global _ORACLE_SAMPLES  # (global column index, oracle u[Q, 21, 3]) of a few columns, for the parity field
    _ORACLE_SAMPLES = [(r[1], r[2]) for r in results[:8]]
    return dict(value=done / dt, unit="column-solves/sec", cores=cores, kind="port",
                sample=f"{done} cfg4 columns (L=20, NQuad=32, 32 Fourier modes, u at 21 tau x 3 phi + fluxes), "
                       f"{cores} processes x 1 BLAS thread, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--columns", type=int, default=2048, help="columns per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the parity and only_flux legs (profiling runs: every kernel launch is then the workload)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the multi-rank code path even with one rank")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    from pydisort_amd._prepare import prepare_columns

    first, C = shard_columns(rank, world, a.columns)
    cfg = synthetic.cfg4_columns(C, first=first)
    N = NQUAD // 2
    prep = prepare_columns(cfg["tau_arr"], cfg["omega_arr"], NQUAD, cfg["Leg_coeffs_all"], cfg["mu0"], cfg["I0"],
                           cfg["phi0"], NQUAD, NQUAD, np.zeros((C, N, NQUAD)), np.zeros((C, N, NQUAD)),
                           cfg["f_arr"], np.zeros((C, L, 0)), np.zeros((C, 0, N, N)), np.zeros((C, 0, N)))
    if world > 1 or a.force_dist:
        Plan.comm_preload()  # bind RCCL to librtd's HIP runtime before torch (gloo control plane) is imported
    plan = Plan(prep, device=local)  # uploads: inputs now resident in HBM
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))

    dist = None
    gather = None
    comm_thread = None
    collective = "none (single rank)"
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        dist.init_process_group("gloo")  # control plane: barriers and the max-over-ranks of the time
        collective = "none (nccl unavailable)"
        # data plane: one RCCL all-gather of the flux results per step (ncclAllGather inside librtd)
        uid = [None]
        if rank == 0:
            try:
                uid = [Plan.comm_unique_id()]
            except Exception as e:
                print(f"[bench] RCCL unavailable: {e!r}", file=sys.stderr)
        dist.broadcast_object_list(uid, src=0)
        ok = 0
        comm_thread = None
        if uid[0] is not None:
            # RCCL bootstrap in a watchdog thread: a rank that cannot reach its peers must not hang the benchmark
            import threading
            state = {}

            def bootstrap():
                try:
                    plan.comm_init(uid[0], rank, world)
                    plan.run()
                    plan.allgather_fluxes()
                    plan.synchronize()
                    state["ok"] = True
                except Exception as e:  # keep the benchmark alive on a misconfigured node
                    state["err"] = e

            comm_thread = threading.Thread(target=bootstrap, daemon=True)
            comm_thread.start()
            comm_thread.join(timeout=float(os.environ.get("RTD_RCCL_TIMEOUT", "120")))
            if state.get("ok"):
                ok = 1
            else:
                print(f"[bench] rank {rank}: RCCL data plane unavailable: {state.get('err', 'bootstrap timed out')!r}",
                      file=sys.stderr)
        import torch
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            gather = plan.allgather_fluxes
            collective = "rccl all_gather (fluxes)"

    def barrier():
        plan.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        plan.run()
        if gather:
            gather()
    plan.enable_timing(True)
    plan.timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plan.run()
        if gather:
            gather()
    plan.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        elapsed = reduce_max_seconds(dist, elapsed)
    stage = plan.timing(reset=True)
    sweeps = plan.max_sweeps()

    extras = {}
    if rank == 0 and world == 1 and not a.no_extras:
        extras = extra_measurements(local)
    if rank == 0:
        fl = algorithmic_flops()
        ms = {k: (v[0] / max(v[1], 1)) for k, v in stage.items()}
        ms["bc"] = ms["iface"] + ms["sweep"]
        ms["eigen"] = ms["asm"] + ms["jacobi"] + ms["post"]  # one fused kernel at NQuad = 32 (timed in the jacobi slot)
        dom = max(("eigen", "iface", "sweep"), key=lambda k: ms[k])
        dom_flops = fl["bc"] if dom in ("iface", "sweep") else fl["asm"] + fl["jacobi"] + fl["post"]
        dom_ms = ms["bc"] if dom in ("iface", "sweep") else ms["eigen"]
        fused_bc = ms["iface"] < 0.05 * ms["sweep"]  # NQuad = 32: one fused kernel, timed in the sweep slot
        kname = ("rtd_eigen_kernel<16>" if dom == "eigen" else
                 "rtd_bc_mfma_kernel" if fused_bc else "rtd_iface_mfma_kernel+rtd_sweep_kernel<16>")
        achieved = dom_flops * C / (dom_ms * 1e-3) / 1e12
        value = world * C * a.steps / elapsed
        traffic = measured_traffic(kname.split("+")[-1].split("<")[0]) if C == 2048 else None
        out = {
            "metric": "column-solves/sec (32 streams, 20 layers)", "value": value, "unit": "column-solves/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg4: synthetic Henyey-Greenstein, 20 layers, 32 streams, 32 Fourier modes, "
                                   "delta-M on, beam source, u at 21 interfaces x 3 azimuths + fluxes",
                       "columns_per_gpu_per_step": C, "global_columns_per_step": world * C,
                       "parallelism": f"column-sharded x{world}", "collective": collective,
                       "max_jacobi_sweeps": sweeps},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "kernel": kname,
                         "note": "FP64 path (SURVEY 8(d): compute-bound, not HBM-bound): peak = MI355X FP64 vector = matrix "
                                 "peak; achieved = algorithmic FLOPs of the kernel x columns / its HIP-event duration; "
                                 "traffic = measured HBM bytes per launch of that kernel (profiles/r01_pmc_traffic.json)",
                         "kernel_ms_per_step": ms,
                         "whole_path_tflops": fl["total"] * C * a.steps / elapsed / 1e12,
                         "whole_path_frac": fl["total"] * C * a.steps / elapsed / 1e12 / FP64_PEAK_TFLOPS},
            "cpu_baseline": cpu,
        }
        out.update(extras)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
        if comm_thread is not None and comm_thread.is_alive():
            sys.stdout.flush()
            os._exit(0)  # a rank stuck inside the RCCL bootstrap cannot be joined


if __name__ == "__main__":
    main()
