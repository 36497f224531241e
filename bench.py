#!/usr/bin/env python3
"""Headline benchmark: column-solves/sec for 20-layer, 32-stream Henyey-Greenstein atmospheres
(BASELINE.json configs[3], "cfg4" of SURVEY section 8(d)) on N GPUs of one node.

A step = one pass of the whole hot path (Legendre tables, eigen stage, boundary-condition solve,
evaluation of u at the 21 layer interfaces x 3 azimuths plus fluxes) over one batch of synthetic
columns per GPU, inputs already resident in HBM.  Columns are independent (the reference's loops
_solve_for_gen_and_part_sols.py:88-91 and _solve_for_coeffs.py:110-111 carry no state), so ranks shard
them with no exchange during the solve; one RCCL all-gather of u and the fluxes per step stitches the
outputs of all ranks, on its own stream so that it overlaps the next step.

  strong scaling (default): --total-columns T (default 100000: BASELINE's literal batch) is split over the GPUs and
                            solved in windows of --columns columns (256; the eigen kernel of window w + 1 runs beside the boundary-
                            condition kernel of window w on two streams): a step = the whole batch, 100000/N per GPU
  weak scaling            : --total-columns 0: every GPU solves --columns columns per step
  --gather auto|all|root|none : what happens to the results of a step when N > 1 -- ncclAllGather to every rank, ncclSend/ncclRecv to
                            rank 0 only, or nothing (compute scaling alone); auto (default) takes the all-gather unless it is >= 10 %
                            slower than root-only.  Whatever is gathered is VERIFIED before the line is printed (every rank's slot
                            against a local solve of the same columns, bit for bit; exit 4 on a mismatch), and the line carries the
                            compute-only / all-gather / root-only rates of the same run (`gather_rates`).

Launch: `python bench.py --gpus N ...` starts N fresh rank processes itself (one per GPU, before anything in
this process has touched a GPU) unless it already runs under torch.distributed.run (RANK / WORLD_SIZE set),
in which case WORLD_SIZE must equal --gpus.  A rank that cannot join, or an RCCL failure, makes the whole
run exit non-zero.

Prints ONE JSON line on rank 0 (contract in the build prompt) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak (vendor figure)
L, NQUAD, NTAU, NPHI = 20, 32, 21, 3
EXIT_RANKS, EXIT_RCCL, EXIT_VERIFY, EXIT_TIMEOUT = 2, 3, 4, 124


def algorithmic_flops(nlayers=L, nquad=NQUAD, nmodes=None, ntau=NTAU):
    """Per-column algorithmic FLOPs per kernel (F_col = L[2N^2 P(P+1) + 70.3 N^3 M] = 195 MFLOP
    for cfg4; per (m, l): assembly 4N^2(P-m), product 2N^3, eigen-decomposition 25N^3, U = (alpha+beta)V/k 2N^3,
    particular solve 5.33N^3, BC solve 36N^3 per layer)."""
    N, P = nquad // 2, nquad
    M = nquad if nmodes is None else nmodes
    ml = nlayers * M
    asm_gemm = nlayers * sum(4 * N * N * (P - m) for m in range(M))  # = L 2N^2 P(P+1) when M = P
    fl = dict(asm=asm_gemm + 2 * N**3 * ml, jacobi=25.0 * N**3 * ml, post=(2 + 5.33) * N**3 * ml,
              bc=36.0 * N**3 * ml, eval=ntau * M * (2 * N) * (2 * N) * 2.0)
    fl["total"] = fl["asm"] + fl["jacobi"] + fl["post"] + fl["bc"]
    return fl


def measured_traffic(kernel, columns_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this round (profiles/r06_pmc_traffic.json:
    FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as read; separate passes), or None when that file was taken at
    another window size than this run's."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")) as f:
            rec = json.load(f)
        if int(rec.get("columns_per_launch", 0)) != int(columns_per_launch):
            return None
        return rec["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def north_star_evidence(columns_per_launch, seconds_per_window, live=None):
    """The north star asks for "rocprof HBM GB/s and MFMA utilisation against peak": MEASURED traffic (not algorithmic: the
    path is compute-bound) of all kernels of a window from the committed PMC passes over this run's time per
    window, and the matrix-pipe / vector-issue busy fractions of the two main kernels from the same passes."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")) as f:
            rec = json.load(f)
        if int(rec.get("columns_per_launch", 0)) != int(columns_per_launch):
            return None
        total = float(sum(live.values())) if live else float(rec["total_hbm_bytes_per_step"])
        out = {"hbm_bytes_per_window": total, "traffic_measured_in_this_run": bool(live), "hbm_GB_per_s": total / seconds_per_window / 1e9,
               "frac_of_8_TB_per_s": total / seconds_per_window / 8e12,
               "what": "measured traffic (FETCH_SIZE x 2 + WRITE_SIZE of every kernel of a window, profiles/r06_pmc_traffic.json) / "
                       "this run's time per window; NOT algorithmic bytes (22.5 KB per column: 0.003 % of 8 TB/s)"}
        for k, name in (("rtd_eigen_kernel", "eigen"), ("rtd_bc_mfma_kernel", "bc")):
            v = rec["kernels"][k]
            simd_cycles = v["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
            out[name + "_kernel"] = {"mfma_busy_frac": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles,
                                     "valu_issue_frac": 4.0 * v["SQ_INSTS_VALU"] / simd_cycles,
                                     "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]}
        return out
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (NumPy/SciPy restatement of the reference, same LAPACK calls) on the host cores
# ---------------------------------------------------------------------------------------------------------
def physical_cores():
    """One logical CPU per physical core among the CPUs this process may run on (SMT siblings dropped)."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, picked = set(), []
    for cpu in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as f:
                sib = f.read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            picked.append(cpu)
    return picked


def _cpu_worker(args):
    cpu, first, seconds, min_cols = args
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    from threadpoolctl import threadpool_limits
    from oracle import disort_oracle as O
    from pydisort_amd import synthetic
    phi = np.array([0.0, np.pi / 2, np.pi])
    keep = None
    n = 0
    with threadpool_limits(1):
        t0 = time.perf_counter()
        while n < min_cols or time.perf_counter() - t0 < seconds:
            cfg = synthetic.cfg4_columns(1, first=first + n)
            res = O.pydisort(**synthetic.column_kwargs(cfg, 0))
            tau = np.concatenate(([0.0], cfg["tau_arr"][0]))
            u = res[4](tau, phi)
            res[1](tau), res[2](tau)
            if n == 0:
                keep = u
            n += 1
        dt = time.perf_counter() - t0
    return n, dt, first, keep


_ORACLE_SAMPLES = []


def cpu_baseline(seconds=20.0, min_cols=16):
    """Oracle on every physical host core: one process pinned to each, 1 BLAS thread, same synthetic inputs as the GPU
    leg, every worker solving columns for `seconds` (at least `min_cols`).  value = sum of the workers' own rates.
    Runs BEFORE the GPU is initialised (fork)."""
    import multiprocessing as mp
    cpus = physical_cores()
    ctx = mp.get_context("fork")
    with ctx.Pool(len(cpus)) as pool:
        pool.map(_cpu_worker, [(c, 0, 0.0, 1) for c in cpus])  # warm imports
        t0 = time.perf_counter()
        results = pool.map(_cpu_worker, [(c, 10_000 + 1000 * k, seconds, min_cols) for k, c in enumerate(cpus)], chunksize=1)
        wall = time.perf_counter() - t0
    rates = [n / dt for n, dt, _, _ in results]
    done = sum(r[0] for r in results)
    global _ORACLE_SAMPLES  # (global column index, oracle u[Q, 21, 3]) of a few columns, for the parity field
    _ORACLE_SAMPLES = [(r[2], r[3]) for r in results[:8]]
    out = dict(value=float(sum(rates)), unit="column-solves/sec", cores=len(cpus), kind="port",
               per_process=dict(mean=float(np.mean(rates)), min=float(min(rates)), max=float(max(rates))),
               logical_cpus=os.cpu_count(),
               sample=f"{done} cfg4 columns (L=20, NQuad=32, 32 Fourier modes, u at 21 tau x 3 phi + fluxes), "
                      f"{len(cpus)} processes pinned one per physical core x 1 BLAS thread, {wall:.1f} s")
    try:  # ratio oracle / reference on identical hardware and inputs, measured in the build container
        with open(os.path.join(ROOT, "profiles", "archive", "r02_cpu_calibration.json")) as f:
            cal = json.load(f)
        out["calibration"] = dict(r=cal["r"], meaning="oracle rate / reference (PythonicDISORT) rate, same inputs, same core",
                                  source="profiles/archive/r02_cpu_calibration.json (tools/calibrate_cpu_baseline.py)")
    except Exception:
        pass
    return out


def cpu_baseline_subprocess():
    """The CPU leg in a process of its own (started before this one touches the GPU): its pool of forked workers and the
    memory they churn stay out of the benchmark process."""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        print(f"[bench] cpu baseline leg failed: {r.stderr[-500:]}", file=sys.stderr)
        return None
    res = json.loads(line)
    global _ORACLE_SAMPLES
    _ORACLE_SAMPLES = [(int(f), np.array(u)) for f, u in res.pop("_samples")]
    return res


# ---------------------------------------------------------------------------------------------------------
# live HBM traffic: rocprofv3 --pmc passes over a short child run of the same workload (before this process touches a GPU)
# ---------------------------------------------------------------------------------------------------------
FILLED_COLUMNS = 65536  # cfg3 at a chip-filling batch (the 1 024-column BASELINE batch is 768 ... 2 048 wavefronts for 1 024 SIMDs)


def pmc_child_filled():
    """Child of filled_counters(): the two cfg3 workloads at FILLED_COLUMNS columns, one warm-up and one measured pass each."""
    import pydisort_amd
    from pydisort_amd import synthetic
    for big in (True, False):
        cfg = synthetic.cfg3_columns(FILLED_COLUMNS, big=big)
        _, sol = pydisort_amd.pydisort_batch(device=0, work_columns=FILLED_COLUMNS, _defer_solve=True, **cfg)  # (one window: 27 GB at 16 streams)
        sol.plan.set_eval_points(np.concatenate((np.zeros((FILLED_COLUMNS, 1)), cfg["tau_arr"]), axis=1), np.array([0.0, np.pi / 2, np.pi]))
        for _ in range(2):
            sol.plan.run()
        sol.plan.synchronize()
        sol.plan.close()


def pmc_child(columns):
    """The workload of the traffic passes: for the headline config (32 windows of `columns` cfg4 columns) and for BASELINE's other
    configs (cfg5: 2 windows of 128 columns; cfg3 at both sizes: 1 024 columns) one warm-up and one measured pass each, windows one
    after the other (RTD_NO_PIPELINE: a kernel's counters are its own).  The configs use different kernel instances (NP = 16,
    32, 8, 4), so one profiler pass serves them all.  No output."""
    import pydisort_amd
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    C = 32 * columns
    cfg = synthetic.cfg4_columns_block(C, first=0)
    plan = Plan(prepare_cfg4(cfg), device=0, work_columns=columns)
    plan.set_eval_points(np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1), np.array([0.0, np.pi / 2, np.pi]))
    for _ in range(2):
        plan.run()
    plan.synchronize()
    plan.close()
    for maker, kw, cols, win in (("cfg5_columns", {}, 256, 128), ("cfg3_columns", {"big": True}, 1024, 0), ("cfg3_columns", {"big": False}, 1024, 0)):
        cfg = getattr(synthetic, maker)(cols, **kw)
        _, sol = pydisort_amd.pydisort_batch(device=0, work_columns=win, _defer_solve=True, **cfg)
        sol.plan.set_eval_points(np.concatenate((np.zeros((cols, 1)), cfg["tau_arr"]), axis=1), np.array([0.0, np.pi / 2, np.pi]))
        for _ in range(2):
            sol.plan.run()
        sol.plan.synchronize()
        sol.plan.close()


FP64_COUNTERS = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64",
                 "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
LIVE_COUNTERS = {}  # {kernel instance: {counter: mean per launch, "seconds": mean duration under the profiler}} of the FP64 pass


def executed_flops(c):
    """FP64 operations a launch EXECUTED, from the SQ instruction counters (rocprofiler-sdk's own FLOP formula for gfx9:
    counter_defs.yaml `TOTAL_64_OPS`): every FMA wave-instruction = 2 x 64, ADD / MUL / transcendental = 64, an MFMA 'MOP' = 512.
    Wave-instructions count all 64 lanes whatever the EXEC mask and the padding streams: an upper bound on the useful ones."""
    return 64.0 * (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0)
                   + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) + 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)


def executed_roofline(kernel, seconds_per_launch, counters=None):
    """What a kernel really executed against the FP64 peak -- beside `frac`, which prices LAPACK's operation count for the
    reference's algorithm and is therefore a normalised throughput, not a utilisation.  From the live FP64 counter
    pass of this bench.py invocation; None when no profiler ran.
      executed_tflops / executed_frac : counted FP64 operations per launch / the kernel's HIP-event duration (/ 78.6 TFLOP/s)
      valu_issue_frac                 : 4 cycles x wave-level VALU instructions / SIMD cycles while the kernel ran (FP64 issue-bound at 1)
      sustained_clock_ghz             : GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration in the profiled pass
      peak_at_sustained_clock         : 78.6 TFLOP/s x sustained clock / 2.4 GHz -- the peak the chip offers at the clock it holds"""
    c = (LIVE_COUNTERS if counters is None else counters).get(kernel.replace(", ", ","))
    if not c or not seconds_per_launch:
        return None
    fl = executed_flops(c)
    out = {"executed_flop_per_launch": fl, "executed_tflops": fl / seconds_per_launch / 1e12,
           "executed_frac": fl / seconds_per_launch / 1e12 / FP64_PEAK_TFLOPS,
           "executed_what": "FP64 operations counted by SQ_INSTS_VALU_{FMA x 2, ADD, MUL, TRANS}_F64 x 64 lanes + SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 "
                            "(rocprofv3 --pmc pass of this run) / the kernel's HIP-event duration: a UTILISATION of the FP64 units; "
                            "`frac` is LAPACK's operation count for the reference's algorithm over the same time: a normalised throughput"}
    if c.get("SQ_BUSY_CYCLES"):
        out["valu_issue_frac"] = 4.0 * c.get("SQ_INSTS_VALU", 0.0) / (c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
    if c.get("GRBM_GUI_ACTIVE") and c.get("seconds") and c["seconds"] >= 3e-4:
        # (MI355X_MICROARCH.md, DVFS: the quotient reads high on dispatches shorter than about 0.3 ms -- no clock is reported for those)
        ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / c["seconds"] / 1e9
        out["sustained_clock_ghz"] = ghz
        out["peak_at_sustained_clock"] = FP64_PEAK_TFLOPS * ghz / 2.4
        out["executed_frac_of_peak_at_sustained_clock"] = out["executed_tflops"] / out["peak_at_sustained_clock"]
    return out


def _profiler():
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None  # this process is being profiled itself: no profiler inside a profiler
    return exe


def _pmc_pass(exe, counters, child_args, timeout):
    """One rocprofv3 --pmc pass over `bench.py <child_args>` -> ({(kernel instance, counter): [value per launch]},
    {kernel instance: [seconds per launch]}) or None when the pass failed."""
    import csv
    import re
    import shutil
    import tempfile
    env = dict(os.environ, RTD_NO_PIPELINE="1", TMPDIR="/tmp")
    out = tempfile.mkdtemp(prefix="rtd_pmc_", dir="/tmp")
    try:
        r = subprocess.run([exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "--",
                            sys.executable, os.path.abspath(__file__), *child_args],
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
        if r.returncode != 0 or not files:
            print(f"[bench] live counter pass {counters[0]} ... failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
            return None
        acc, dur = {}, {}
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                m = re.search(r"rtd_\w+(<[^>]*>)?", row["Kernel_Name"])
                if m and row["Counter_Name"] in counters:
                    k = m.group(0).replace(", ", ",")
                    acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
                    if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                        dur.setdefault(k, {})[row.get("Dispatch_Id")] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9
        if not dur:  # timestamps in the kernel trace of the same pass
            for path in [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("kernel_trace.csv")]:
                with open(path) as f:
                    for row in csv.DictReader(f):
                        m = re.search(r"rtd_\w+(<[^>]*>)?", row.get("Kernel_Name", ""))
                        if m:
                            dur.setdefault(m.group(0).replace(", ", ","), {})[row.get("Dispatch_Id")] = \
                                (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9
        return acc, {k: list(d.values()) for k, d in dur.items()}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _fold_counters(passed, into):
    """Mean over the second (measured) half of the launches of every kernel instance -> into[kernel][counter], into[kernel]["seconds"]."""
    acc, dur = passed
    for (k, name), v in acc.items():
        v = v[len(v) // 2:]
        into.setdefault(k, {})[name] = sum(v) / len(v)
    for k, v in dur.items():
        v = v[len(v) // 2:]
        if k in into:
            into[k]["seconds"] = sum(v) / len(v)


def filled_counters(timeout=240):
    """The FP64 instruction counters of the <= 16-stream kernels on a batch that FILLS the chip (FILLED_COLUMNS cfg3 columns, same
    kernels as the 1 024-column BASELINE batch): {kernel instance: {counter: mean per launch}} or None.  With them the line
    separates fill from kernel efficiency for cfg3 (round-5 verdict, item 7)."""
    exe = _profiler()
    if not exe:
        return None
    try:
        passed = _pmc_pass(exe, FP64_COUNTERS, ["--pmc-child-filled"], timeout)
    except Exception as e:
        print(f"[bench] filled-batch counters unavailable: {e!r}", file=sys.stderr)
        return None
    if not passed:
        return None
    out = {}
    _fold_counters(passed, out)
    return out or None


def live_traffic(columns, timeout=240):
    """HBM bytes per launch of every kernel of the headline config and of BASELINE's other configs, measured NOW: two rocprofv3
    passes (FETCH_SIZE and WRITE_SIZE cannot share one) of `bench.py --pmc-child`, corrected as MI355X_MICROARCH.md prescribes
    (KiB units; FETCH_SIZE doubled on gfx950), and a third pass with the FP64 instruction counters (FP64_COUNTERS -> LIVE_COUNTERS,
    read by executed_roofline).  Returns {kernel name with its template arguments: bytes per launch} -- the
    plain name too where only one instance of a kernel ran -- or None when rocprofv3 is not usable here."""
    exe = _profiler()
    if not exe:
        return None
    got = {}
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), FP64_COUNTERS):
            passed = _pmc_pass(exe, counters, ["--pmc-child", "--columns", str(columns)], timeout)
            if not passed:
                if counters is FP64_COUNTERS:
                    continue  # the traffic passes stand on their own
                return None
            _fold_counters(passed, LIVE_COUNTERS if counters is FP64_COUNTERS else got)
    except Exception as e:  # profiler missing, timeout, unreadable output: the committed passes are used instead
        print(f"[bench] live traffic unavailable: {e!r}", file=sys.stderr)
        return None
    # (the table kernels run once per change of the inputs, not per window: they are not part of a window's traffic)
    res = {k: (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for k, v in got.items()
           if "FETCH_SIZE" in v and "WRITE_SIZE" in v and not k.startswith("rtd_tables")}
    return res or None


def traffic_of(live, names):
    """Sum of the measured bytes per launch of the kernel instances in `names` ("rtd_eigen_kernel<16,2>" ...), None if unmeasured."""
    if not live:
        return None
    vals = [live.get(n.replace(", ", ",")) for n in names]
    return None if any(v is None for v in vals) else float(sum(vals))


# ---------------------------------------------------------------------------------------------------------
# secondary measurements on rank 0 at N = 1
# ---------------------------------------------------------------------------------------------------------
def roofline_of(stage, fl, cols_per_launch, kernel_names, counters=None):
    """roofline sub-object of one workload from the plan's HIP-event stage times (ms, launches) per slot."""
    ms = {k: (v[0] / max(v[1], 1)) for k, v in stage.items()}  # per launch = per window
    ms["bc"] = ms["iface"] + ms["sweep"]
    ms["eigen"] = ms["asm"] + ms["jacobi"] + ms["post"]  # one fused kernel (timed in the jacobi slot)
    dom = "eigen" if ms["eigen"] >= ms["bc"] else "bc"
    dom_flops = fl["bc"] if dom == "bc" else fl["asm"] + fl["jacobi"] + fl["post"]
    achieved = dom_flops * cols_per_launch / (ms[dom] * 1e-3) / 1e12
    roof = {"bound": "fp64-valu", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_PEAK_TFLOPS, "kernel": kernel_names[dom], "kernel_ms_per_launch": ms,
            "columns_per_launch": cols_per_launch,
            "frac_is": "a normalised throughput: LAPACK's operation count for the reference's algorithm (SURVEY 8(d)) over the kernel's "
                       "time; the utilisation of the FP64 units is `executed_frac`"}
    ex = executed_roofline(kernel_names[dom], ms[dom] * 1e-3, counters)
    if ex:
        roof.update(ex)
    other = "bc" if dom == "eigen" else "eigen"
    ex2 = executed_roofline(kernel_names[other], ms[other] * 1e-3, counters) if ms.get(other) else None
    if ex2:
        roof["other_kernel"] = dict(kernel=kernel_names[other], ms_per_launch=ms[other], **{k: v for k, v in ex2.items() if k != "executed_what"})
    return roof, ms


def golden_parity(name, maker, kwargs, device):
    """max scale-relative / pointwise error of the HIP path on the reference-computed golden columns of a synthetic config
    (tests/golden/synth/<name>.npz: outputs of PythonicDISORT itself, generated in the build container)."""
    import pydisort_amd
    from pydisort_amd import synthetic
    z = np.load(os.path.join(ROOT, "tests", "golden", "synth", name + ".npz"))
    ncol = int(z["ncol"])
    cfg = getattr(synthetic, maker)(ncol, **kwargs)
    _, sol = pydisort_amd.pydisort_batch(device=device, **cfg)
    tau = np.stack([z[f"c{i}.tau_pts"] for i in range(ncol)])
    u = sol.u(tau, z["phi"])
    worst = worst_pw = 0.0
    for i in range(ncol):
        want = z[f"c{i}.u"]
        diff = np.abs(u[i] - want)
        sig = np.abs(want) > 1e-8 * np.max(np.abs(want))
        worst = max(worst, float(diff.max() / np.max(np.abs(want))))
        worst_pw = max(worst_pw, float((diff[sig] / np.abs(want[sig])).max()))
    sol.plan.close()
    return {"max_scale_rel": worst, "max_rel_dI": worst_pw, "columns_checked": ncol,
            "against": f"reference-computed goldens tests/golden/synth/{name}.npz"}


def config_leg(name, golden, maker, kwargs, columns, window, device, passes, live=None, counters=None):
    """One of BASELINE's other configs through the same path: resident rate (plan.run over all windows), host-to-host
    rate (run_fetch: D2H of a window overlapped with the next window's kernels), HIP-event kernel times -> roofline,
    parity of the first columns against the reference-computed goldens."""
    import pydisort_amd
    from pydisort_amd import synthetic
    cfg = getattr(synthetic, maker)(columns, **kwargs)
    nl, nq = cfg["tau_arr"].shape[1], cfg["NQuad"]
    tau = np.concatenate((np.zeros((columns, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, np.pi / 2, np.pi])
    _, sol = pydisort_amd.pydisort_batch(device=device, work_columns=window, _defer_solve=True, **cfg)
    plan = sol.plan
    plan.set_eval_points(tau, phi)
    plan.run()
    plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        plan.run()
    plan.synchronize()
    rate = passes * columns / (time.perf_counter() - t0)
    out_arrays = plan.run_fetch()  # warm (pinned staging)
    t0 = time.perf_counter()
    out_arrays = plan.run_fetch()
    e2e = columns / (time.perf_counter() - t0)
    assert np.all(np.isfinite(out_arrays["flux_up"])) and np.all(np.isfinite(out_arrays["u"]))
    # the batch's first columns ARE the reference-computed golden columns (the generators are deterministic per column): the
    # results of the timed, windowed, pipelined pass itself against the reference at the interfaces
    z = np.load(os.path.join(ROOT, "tests", "golden", "synth", golden + ".npz"))
    in_batch = in_batch_pw = 0.0
    for i in range(int(z["ncol"])):
        pts = np.searchsorted(z[f"c{i}.tau_pts"], tau[i])
        assert np.array_equal(z[f"c{i}.tau_pts"][pts], tau[i])
        want = z[f"c{i}.u"][:, pts, :3]
        diff = np.abs(out_arrays["u"][i] - want)
        sig = np.abs(want) > 1e-8 * np.max(np.abs(want))
        in_batch = max(in_batch, float(diff.max() / np.max(np.abs(want))))
        in_batch_pw = max(in_batch_pw, float((diff[sig] / np.abs(want[sig])).max()))
    plan.enable_timing(True)
    plan.timing(reset=True)
    for _ in range(2):
        plan.run()
    stage = plan.timing(reset=True)
    plan.enable_timing(False)
    cw, nwin = plan.windows()
    fl = algorithmic_flops(nl, nq, nq, nl + 1)
    np_ = 4 if nq <= 8 else 8 if nq <= 16 else 16 if nq <= 32 else 32
    names = {"eigen": "rtd_eigen_lane_kernel<4>" if np_ == 4 else f"rtd_eigen_kernel<{np_}, 2>",
             "bc": "rtd_bc_tile2_kernel" if np_ == 32 else "rtd_bc_mfma_kernel" if np_ == 16 else f"rtd_bc_small_kernel<{np_}>"}
    roof, ms = roofline_of(stage, fl, columns / nwin, names, counters)
    roof["whole_path_tflops"] = fl["total"] * rate / 1e12
    roof["whole_path_frac"] = roof["whole_path_tflops"] / FP64_PEAK_TFLOPS
    # measured HBM bytes per launch of the dominant kernel(s) (the live rocprofv3 passes of this run; cfg5's child runs windows of
    # the same 128 columns, cfg3's the same 1 024 columns in one launch)
    dom_names = [k.strip() for k in roof["kernel"].split("+")]
    roof["traffic"] = traffic_of(live, dom_names) if counters is None else None
    roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation" if roof["traffic"] is not None
                              else "not measured" + (" (no profiler in this run)" if counters is None else " at this batch size"))
    plan.close()
    return {"value": rate, "unit": "column-solves/sec", "workload": name, "columns": columns, "columns_per_window": cw,
            "windows": nwin, "host_to_host": e2e, "mflop_per_column": fl["total"] / 1e6, "roofline": roof,
            "parity": golden_parity(golden, maker, kwargs, device),
            "parity_in_batch": {"max_scale_rel": in_batch, "max_rel_dI": in_batch_pw, "columns_checked": int(z["ncol"]),
                                "against": f"the same goldens, taken from the results of the full {columns}-column windowed pass (run_fetch) at the interfaces"}}


def single_column_leg(device, calls=30):
    """BASELINE configs[1] as worded: Test Problem 5 (Cloud C.1, 300 moments, one layer of optical depth 64, beam source) at
    32 streams with delta-M and the Nakajima-Tanaka corrections, ONE column per call: latency of a whole `pydisort()` call
    with the intensity evaluated at the golden points (host to host: checks, plan, upload, kernels, D2H), and parity against
    the reference's output at that stream count (tests/golden/synth/cfg2_q32_cloud_b.npz)."""
    import warnings
    import pydisort_amd
    z = np.load(os.path.join(ROOT, "tests", "golden", "synth", "cfg2_q32_cloud_b.npz"))
    leg = z["Leg_coeffs_all"]
    kw = dict(tau_arr=np.array([64.0]), omega_arr=z["omega"], NQuad=32, Leg_coeffs_all=leg, mu0=1.0, I0=np.pi, phi0=np.pi,
              f_arr=np.array([leg[0, 32]]), NT_cor=True, device=device)
    times = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(calls + 3):
            t0 = time.perf_counter()
            res = pydisort_amd.pydisort(**kw)
            u = res[4](z["tau_pts"], z["phi"])
            times.append(time.perf_counter() - t0)
            # (`res` is rebound by the next call: its closures go, and their plan serves that call -- pydisort.py: _plan_for)
    times = sorted(times[3:])
    want = z["u"]
    diff = np.abs(u - want)
    sig = np.abs(want) > 1e-8 * np.max(np.abs(want))
    return {"value": 1.0 / times[len(times) // 2], "unit": "column-solves/sec", "latency_ms_median": 1e3 * times[len(times) // 2],
            "latency_ms_min": 1e3 * times[0], "calls": calls,
            "workload": "cfg2: Test Problem 5 (Cloud C.1, 300 moments, tau = 64, omega = 0.9, beam) at 32 streams, delta-M + "
                        "Nakajima-Tanaka corrections, one column per pydisort() call, u at 7 depths x 3 azimuths, host to host",
            "parity": {"max_scale_rel": float(diff.max() / np.max(np.abs(want))), "max_rel_dI": float((diff[sig] / np.abs(want[sig])).max()),
                       "columns_checked": 1, "against": "reference-computed golden tests/golden/synth/cfg2_q32_cloud_b.npz"}}


def many_stream_leg(device, columns=32):
    """The other reading of BASELINE configs[4]: 128 streams, 50 layers, Fourier modes capped at 64
    -- beyond that the reference's own Legendre tables overflow -- on the NP = 64 instances (four-wavefronts-per-chain
    boundary-condition kernels, csrc/rtd_bc_wide.hip; readlane Cholesky and DPP-broadcast assembly in the eigen kernel).  Parity:
    the first two columns of the TIMED batch are the reference-computed golden columns (tests/golden/synth/q128_L50.npz: the same
    128 x 50 x 64 workload at its full depth) and are compared at the interfaces in the timed pass's own results."""
    import pydisort_amd
    from pydisort_amd import synthetic
    maker_kw, nf, ncol = synthetic.many_stream_deep_cases()["q128_L50"]
    cfg = synthetic.cfg4_columns(columns, **maker_kw)
    cfg["NFourier"] = nf
    _, sol = pydisort_amd.pydisort_batch(device=device, _defer_solve=True, **cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((columns, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
    plan.run()
    plan.synchronize()
    t0 = time.perf_counter()
    plan.run()
    plan.synchronize()
    rate = columns / (time.perf_counter() - t0)
    got = plan.fetch()
    cw, nwin = plan.windows()
    plan.close()
    z = np.load(os.path.join(ROOT, "tests", "golden", "synth", "q128_L50.npz"))
    worst = worst_pw = 0.0
    for i in range(ncol):
        pts = np.searchsorted(z[f"c{i}.tau_pts"], tau[i])
        assert np.array_equal(z[f"c{i}.tau_pts"][pts], tau[i])
        want = z[f"c{i}.u"][:, pts, :3]
        diff = np.abs(got["u"][i] - want)
        sig = np.abs(want) > 1e-8 * np.max(np.abs(want))
        worst = max(worst, float(diff.max() / np.max(np.abs(want))))
        worst_pw = max(worst_pw, float((diff[sig] / np.abs(want[sig])).max()))
    return {"value": rate, "unit": "column-solves/sec", "columns": columns, "columns_per_window": cw, "windows": nwin,
            "workload": "128 streams, 50 layers, 64 Fourier modes (Henyey-Greenstein, g up to 0.9, delta-M): rtd_eigen_kernel<64, 2> + rtd_iface_mfma_kernel + rtd_sweep_wide_kernel",
            "parity_in_batch": {"max_scale_rel": worst, "max_rel_dI": worst_pw, "columns_checked": ncol,
                                "against": "reference-computed goldens tests/golden/synth/q128_L50.npz (128 streams, 50 layers, 64 modes: this workload), "
                                           "taken from the results of the timed pass at the 51 interfaces x 3 azimuths"}}


def all_cloud_leg(device, columns=16384, window=256, passes=3):
    """cfg4 with a conservative cloud layer (omega = 1 - 1e-6) in EVERY column: the near-conservative regime the benchmark
    distribution (omega <= 0.99) never touches.  Every Fourier-mode-0 chain then takes the column-pivoted elimination of the
    boundary-condition kernel (register-resident since round 4); the rate beside the headline makes that cost driver-visible,
    and two columns with 40-digit solutions (tests/golden/hp/synth_cfg4cloud_*.npz), spliced into different windows of the
    timed batch, are checked at the interfaces."""
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    cfg = synthetic.cfg4_cloud_columns(columns)
    at = (100, 3000)
    truths = []
    for k, pos in enumerate(at):
        one = synthetic.cfg4_cloud_columns(k + 1)
        for key, v in one.items():
            if isinstance(v, np.ndarray) and v.shape[:1] == (k + 1,):
                cfg[key][pos] = v[k]
        truths.append(np.load(os.path.join(ROOT, "tests", "golden", "hp", f"synth_cfg4cloud_{k}.npz")))
    plan = Plan(prepare_cfg4(cfg), device=device, work_columns=window)
    tau = np.concatenate((np.zeros((columns, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
    plan.run()
    plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        plan.invalidate_tables()
        plan.run()
    plan.synchronize()
    rate = passes * columns / (time.perf_counter() - t0)
    res = plan.fetch()
    plan.close()
    worst = worst_pw = ref = ref_pw = 0.0
    for z, pos in zip(truths, at):
        pts = np.searchsorted(z["tau"], tau[pos])
        assert np.array_equal(z["tau"][pts], tau[pos])
        want = z["u"][:, pts, :3]
        diff = np.abs(res["u"][pos] - want)
        sig = np.abs(want) > 1e-8 * np.max(np.abs(want))
        worst = max(worst, float(diff.max() / np.max(np.abs(want))))
        worst_pw = max(worst_pw, float((diff[sig] / np.abs(want[sig])).max()))
        ref, ref_pw = max(ref, float(z["oracle_u_scale_rel"])), max(ref_pw, float(z["oracle_u_pointwise_rel"]))
    return {"value": rate, "unit": "column-solves/sec", "columns": columns, "columns_per_window": window,
            "workload": "cfg4 with an omega = 1 - 1e-6 layer in every column (fresh inputs every pass): every mode-0 chain is pivoted throughout",
            "parity": {"max_scale_rel": worst, "max_rel_dI": worst_pw, "columns_checked": len(at),
                       "against": "40-digit solutions tests/golden/hp/synth_cfg4cloud_{0,1}.npz, taken from the timed batch at the interfaces",
                       "reference_algorithm_vs_truth": {"max_scale_rel": ref, "max_rel_dI": ref_pw,
                                                        "note": "the float64 oracle (= the reference's algorithm) on the same two columns"}}}


def lean_retention_leg(device, cfg):
    """Evaluators that outlive the solve on the headline batch itself (round-5 verdict, item 6): the lean retained form keeps the
    boundary-condition coefficients, k, E, B and the thermal vectors of EVERY column (0.5 MB per cfg4 column; the full evaluator
    state would be 3.1 MB: 310 GB for 10^5 columns) and re-runs the eigen stage -- never the boundary-condition solve -- for the
    layers a requested depth touches.  Timed: the solve of the batch, then `sol.u(tau, phi)` at ONE non-interface depth per column;
    checked: bit-identity with a full-retention plan on a slice of the batch.  The reference's closures re-evaluate from
    GC_collect, K_collect, B_collect (_assemble_intensity_and_fluxes.py:170-262)."""
    import pydisort_amd
    C = cfg["tau_arr"].shape[0]
    rng = np.random.default_rng(17)
    tau = rng.uniform(0.02, 0.98, (C, 1)) * cfg["tau_arr"][:, -1:]
    phi = np.array([0.5, 2.0])
    _, sol = pydisort_amd.pydisort_batch(device=device, work_columns=256, retain="auto", retain_bytes=64 << 30, **cfg)
    form, held = sol.plan.retained_form(), sol.plan.device_bytes()
    sol.plan.synchronize()
    t0 = time.perf_counter()
    sol.plan.solve()
    sol.plan.synchronize()
    t_solve = time.perf_counter() - t0
    first = sol.u(tau, phi)  # (grows the evaluation buffers)
    t0 = time.perf_counter()
    again = sol.u(tau, phi)
    t_eval = time.perf_counter() - t0
    n = min(C, 1024)
    sub = {k: (v[:n] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == C else v) for k, v in cfg.items()}
    _, full = pydisort_amd.pydisort_batch(device=device, work_columns=256, retain="full", retain_bytes=8 << 30, **sub)
    same = bool(np.array_equal(full.u(tau[:n], phi), first[:n]) and np.array_equal(first, again))
    full_form = full.plan.retained_form()
    full.plan.close()
    sol.plan.close()
    return {"columns": C, "retained_form": form, "device_bytes": held, "solve_seconds": t_solve, "evaluate_seconds": t_eval,
            "evaluate_over_solve": t_eval / t_solve, "bit_identical_to_full_retention": same, "full_form_of_the_slice": full_form,
            "what": "sol.u(tau, phi) at one non-interface depth per column and 2 azimuths on the whole batch, from the lean retained state "
                    "(coefficients, k, E, B per column; Y, A of the touched wavefront chunks recomputed); compared bit for bit with a "
                    f"full-retention plan of the first {n} columns"}


def n_rank_check(ranks=4, columns=4096):
    """NOT a measurement: a check, inside the N = 1 run, that the N-rank data plane of THIS build executes -- `bench.py --gpus 4` as
    a child process with all rank processes on device 0 over the tests' stand-in transport (tests/stub/librccl_stub.so, selected by
    RTD_RCCL_STUB; RCCL itself refuses two ranks on one GPU): socket control plane, communicator of 4, both collectives tried, every
    rank's slot of the gathered arrays verified bit for bit.  Only what it proved is kept; its rate is not (ranks sharing one GPU
    over a synchronous transport)."""
    stub_lib = os.path.join(ROOT, "tests", "stub", "librccl_stub.so")
    if not os.path.exists(stub_lib):
        return {"ran": False, "why": "tests/stub/librccl_stub.so is not built (python tests/stub/build_stub.py)"}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTD_RCCL_STUB=stub_lib, RTD_BENCH_TIMEOUT="240")
    env.setdefault("RCCL_STUB_TIMEOUT_S", "120")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-extras", "--total-columns", str(columns)], env=env, capture_output=True, text=True, timeout=300)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
        if r.returncode != 0 or line is None:
            return {"ran": True, "ok": False, "exit_code": r.returncode, "stderr_tail": r.stderr[-600:]}
        d = json.loads(line)
        return {"ran": True, "ok": bool(d["config"].get("gather_verified")) and d["config"].get("ranks_verified") == ranks,
                "ranks": d["n_gpus"], "devices_used": d.get("devices_used"), "transport": d.get("transport"), "not_a_rate": d.get("not_a_rate"),
                "rccl_nranks": d["config"].get("rccl_nranks"), "gather_verified": d["config"].get("gather_verified"),
                "ranks_verified": d["config"].get("ranks_verified"), "collective_chosen": d.get("gather_rates", {}).get("chosen"),
                "columns": columns, "seconds": round(time.perf_counter() - t0, 1),
                "what": f"bench.py --gpus {ranks} as a child of this run, {ranks} rank processes on device 0 over the tests' stand-in transport: "
                        "the N-rank control and data plane executed and every gathered slot equals a local solve, bit for bit"}
    except Exception as e:
        return {"ran": True, "ok": False, "error": repr(e)}


def extra_measurements(device, main_cfg=None, window=2048, live=None, filled=None):
    """max |dI| of the HIP path against the oracle on the sample columns of the cpu_baseline leg, the only_flux
    throughput, the host-to-host rate of the main batch, and BASELINE's other configs."""
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
    out["parity_goldens"] = golden_parity("cfg4", "cfg4_columns", {}, device)
    C = 16384
    cfg = synthetic.cfg4_columns_block(C, first=50_000)
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
    out["e2e"] = end_to_end(device, main_cfg, window)
    if main_cfg is not None:
        try:
            out["retention"] = lean_retention_leg(device, main_cfg)
        except Exception as e:  # (a box with less free memory than the leg wants: reported, not fatal)
            out["retention"] = {"error": repr(e)}
    out["other_configs"] = {
        "cfg2_cloudC1_Q32_single_column": single_column_leg(device),
        "cfg3_L6_Q8_x1024": config_leg("cfg3: Test Problem 9c (6 layers, 8 streams, thermal + beam + Lambertian surface) x 1024 perturbed columns",
                                       "cfg3_small", "cfg3_columns", {"big": False}, 1024, 0, device, 50, live),
        "cfg3_L8_Q16_x1024": config_leg("cfg3 at BASELINE's size (8 layers, 16 streams) x 1024 perturbed columns",
                                        "cfg3_big", "cfg3_columns", {"big": True}, 1024, 0, device, 50, live),
        # the SAME kernels on a batch that fills the chip: how much of the 1 024-column figures is fill, how much the kernels
        "cfg3_L6_Q8_x65536_chip_filling": config_leg("cfg3 (6 layers, 8 streams) x 65536 perturbed columns: the 1 024-column BASELINE batch's kernels on a "
                                                    "batch that fills the chip (one window, one launch per kernel)",
                                                    "cfg3_small", "cfg3_columns", {"big": False}, FILLED_COLUMNS, FILLED_COLUMNS, device, 5, None, filled or {}),
        "cfg3_L8_Q16_x65536_chip_filling": config_leg("cfg3 at BASELINE's size (8 layers, 16 streams) x 65536 perturbed columns: chip-filling batch",
                                                     "cfg3_big", "cfg3_columns", {"big": True}, FILLED_COLUMNS, FILLED_COLUMNS, device, 5, None, filled or {}),
        "cfg5_L50_Q64_x10000": config_leg("cfg5 at BASELINE's literal size: 50 layers, 64 streams, 64 Fourier modes, 2-mode BDRF surface, thermal "
                                          "source; 10^4 columns in 79 windows of 128",
                                          "cfg5", "cfg5_columns", {}, 10_000, 128, device, 2, live),
        "cfg5alt_L50_Q128_M64_x32": many_stream_leg(device),
        "cfg4_all_cloud_x16384": all_cloud_leg(device),
    }
    return out


def end_to_end(device, cfg=None, window=2048):
    """Host arrays in -> host arrays out for the main batch (BASELINE's literal 10^5 cfg4 columns by default): input
    checks, upload of the raw inputs, delta-M scaling on the device, windowed solve, evaluation, and the device-to-host
    copies overlapped with the next window's kernels."""
    import pydisort_amd
    from pydisort_amd import synthetic
    if cfg is None:
        cfg = synthetic.cfg4_columns_block(100_000, first=0)
    columns = cfg["tau_arr"].shape[0]
    tau = np.concatenate((np.zeros((columns, 1)), cfg["tau_arr"]), axis=1)
    phi = np.array([0.0, np.pi / 2, np.pi])
    nwarm = min(columns, 2 * window)
    pydisort_amd.solve_columns_streamed({k: (v[:nwarm] if isinstance(v, np.ndarray) else v) for k, v in cfg.items()},
                                        tau[:nwarm], phi, chunk_columns=window, device=device)  # warm-up
    # result arrays are the caller's (a serving loop reuses them): allocated and touched once, outside the timed call
    out = dict(u=np.zeros((columns, NQUAD, NTAU, NPHI)), u0=np.zeros((columns, NQUAD, NTAU)), flux_up=np.zeros((columns, NTAU)),
               flux_down_diffuse=np.zeros((columns, NTAU)), flux_down_direct=np.zeros((columns, NTAU)))
    for a in out.values():
        a.fill(0.0)  # (np.zeros maps pages lazily: touch them here, not inside the first timed call)
    calls = []
    with pydisort_amd.pooled(device=device):  # a serving loop opts in: the arena of a call serves the next (include/rtd.h: rtd_pool_set_limit)
        for _ in range(3):
            t0 = time.perf_counter()
            res = pydisort_amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=window, device=device, out=out)
            calls.append(time.perf_counter() - t0)
    best = min(calls)
    assert np.all(np.isfinite(res["flux_up"])) and res["u"] is out["u"]
    # (the host side of a call -- input checks, pageable H2D, the copy out of the pinned staging buffers -- shares the
    #  box's CPU cores and memory bus with whatever else runs there: calls of 1.45 s next to the usual 0.53 s have been seen
    #  on a busy host, hence every call's time is reported)
    return {"value": columns / best, "unit": "column-solves/sec", "columns": columns, "seconds": best,
            "seconds_per_call": [round(x, 4) for x in calls], "median": columns / sorted(calls)[1],
            "what": "the SAME batch as `value`, host to host: NumPy inputs (raw: tau, omega, 33 moments, f, mu0, I0, phi0) -> "
                    "NumPy u [C,32,21,3], u0, fluxes, one call: input checks, plan creation, H2D of the raw inputs, delta-M "
                    f"scaling / rescaling on the device, windowed solve + evaluation ({window} columns per window), D2H through "
                    "pinned staging overlapped with the next window; result arrays preallocated by the caller; the loop has opted in to the "
                    "library's large-block pool (pydisort_amd.pooled(): off by default); best of 3 calls (every call's time listed)"}


# ---------------------------------------------------------------------------------------------------------
# rank layout
# ---------------------------------------------------------------------------------------------------------
def shard_columns(rank, world, columns_per_gpu, total_columns=0):
    """Column shard of a rank: global column indices [first, first + count).  Weak scaling: every rank gets
    columns_per_gpu; strong scaling (total_columns > 0): total_columns // world each (equal counts for the all-gather)."""
    if total_columns > 0:
        per = total_columns // world
        return rank * per, per
    return rank * columns_per_gpu, columns_per_gpu


def verification_columns(count, rank, per_rank=4):
    """Which columns of a rank's shard the gathered results are checked on: the first, the last and seeded interior ones
    (different windows of the shard)."""
    rng = np.random.default_rng([2024, rank, count])
    picks = {0, count - 1} | {int(i) for i in rng.integers(0, count, size=max(0, per_rank - 2))}
    for i in range(count):  # (tiny shards: fill up deterministically)
        if len(picks) >= min(per_rank, count):
            break
        picks.add(i)
    return sorted(picks)


def shard_config(rank, world, columns_per_gpu, total_columns):
    """The synthetic inputs of a rank's shard, exactly as that rank generates them (deterministic in the shard)."""
    from pydisort_amd import synthetic
    first, count = shard_columns(rank, world, columns_per_gpu, total_columns)
    cfg = synthetic.cfg4_columns_block(count, first=first) if total_columns > 0 else synthetic.cfg4_columns(count, first=first)
    return cfg, first, count


def prepare_cfg4(cfg):
    from pydisort_amd._prepare import prepare_columns
    C, N = cfg["tau_arr"].shape[0], NQUAD // 2
    return prepare_columns(cfg["tau_arr"], cfg["omega_arr"], NQUAD, cfg["Leg_coeffs_all"], cfg["mu0"], cfg["I0"],
                           cfg["phi0"], NQUAD, NQUAD, None, None,  # no Dirichlet sources: nothing to allocate or upload
                           cfg["f_arr"], np.zeros((C, L, 0)), np.zeros((C, 0, N, N)), np.zeros((C, 0, N)))


def verify_gathered(plan, device, world, columns_per_gpu, total_columns, per_rank=4):
    """The gathered arrays against the truth, on a rank that holds them: for EVERY rank's shard, regenerate its inputs, solve
    `per_rank` of its columns here on their own (a small one-window plan) and compare them bit for bit with that rank's
    slot of the gathered u and fluxes.  Catches wrong offsets, rank order, stale or torn slots.  Returns (ok, ranks, detail)."""
    from pydisort_amd._engine import Plan
    detail = []
    ok = True
    for r in range(world):
        cfg, _, count = shard_config(r, world, columns_per_gpu, total_columns)
        idx = verification_columns(count, r, per_rank)
        sub = {k: (v[idx] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == count else v) for k, v in cfg.items()}
        small = Plan(prepare_cfg4(sub), device=device)
        try:
            small.set_eval_points(np.concatenate((np.zeros((len(idx), 1)), sub["tau_arr"]), axis=1), np.array([0.0, np.pi / 2, np.pi]))
            small.run()
            want = small.fetch()
        finally:
            small.close()
        bad = 0
        for j, i in enumerate(idx):
            gu, gf = plan.fetch_gathered_columns(r, i, 1)
            same = (np.array_equal(gu[0], want["u"][j]) and np.array_equal(gf[0, 0], want["flux_up"][j])
                    and np.array_equal(gf[1, 0], want["flux_down_diffuse"][j]) and np.array_equal(gf[2, 0], want["flux_down_direct"][j]))
            bad += 0 if same else 1
        ok = ok and bad == 0
        detail.append({"rank": r, "columns": idx, "mismatches": bad})
    return ok, world, detail


def reduce_max_seconds(ctl, seconds):
    """Max over ranks of a host-side duration through the control plane (pydisort_amd/_control.py: sockets, no PyTorch)."""
    return float(ctl.allreduce(float(seconds), "max"))


def all_ranks_ok(ctl, ok):
    """True iff every rank reports ok (MIN over the control plane)."""
    return ctl.all_ok(bool(ok))


# Where a rank is, for the message of a run that is stopped from outside (time-out) or by its own deadline: a hang is reported
# with the phase and the Python stack of every rank, not as a silent kill.
_PHASE = {"name": "start", "since": time.time(), "t0": time.time()}


def phase(name):
    _PHASE.update(name=name, since=time.time())
    d = os.environ.get("RTD_BENCH_RUN_DIR")
    if d:
        try:
            with open(os.path.join(d, f"phase_{os.environ.get('RANK', '0')}"), "w") as f:
                f.write(f"{name} (entered {time.time() - _PHASE['t0']:.1f} s after the rank started)")
        except OSError:
            pass


def arm_rank_deadline(rank, seconds):
    """Every rank ends itself -- with its phase and the stack of every thread on stderr -- `seconds` after it started, whatever
    launched it: the driver's own limit (1 800 s) must never be what stops a hung run.  SIGUSR1 dumps the stacks without
    ending the rank (bench.py's launcher sends it before it stops the ranks)."""
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, file=sys.stderr, all_threads=True)

    def expired():
        print(f"[bench] rank {rank}: still in phase '{_PHASE['name']}' (for {time.time() - _PHASE['since']:.0f} s) when the run's "
              f"deadline of {seconds:.0f} s expired; stacks follow", file=sys.stderr)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        sys.stderr.flush()
        os._exit(EXIT_TIMEOUT)

    t = threading.Timer(seconds, expired)
    t.daemon = True
    t.start()
    return t


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


DEFAULT_TIMEOUT = 1500.0  # seconds; below the driver's 1 800 s, so that a hang ends HERE, with a diagnosis


def spawn_ranks(n, argv, timeout=None):
    """Start n fresh rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, one GPU each), wait for
    all of them, relay rank 0's JSON line.  Any rank failing (or the timeout) ends the others and the run with a
    non-zero exit code, after every rank's phase has been printed and every live rank has dumped its stacks.  The parent
    never touches a GPU."""
    import signal
    import tempfile
    timeout = float(os.environ.get("RTD_BENCH_TIMEOUT", str(DEFAULT_TIMEOUT))) if timeout is None else timeout
    port = _free_port()
    run_dir = tempfile.mkdtemp(prefix="rtd_bench_run_")
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RTD_BENCH_RUN_DIR=run_dir, RTD_BENCH_TIMEOUT=str(timeout + 30.0))  # (the launcher's limit comes first)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0 = time.time()
    code = 0

    def report_state():
        for r, p in enumerate(procs):
            try:
                with open(os.path.join(run_dir, f"phase_{r}")) as f:
                    where = f.read()
            except OSError:
                where = "no phase recorded"
            st = p.poll()
            print(f"[bench]   rank {r}: {'running' if st is None else f'exited with status {st}'}; last phase: {where}", file=sys.stderr)
        for p in procs:  # live ranks dump the stacks of all their threads (faulthandler, also from inside a C call)
            if p.poll() is None:
                try:
                    p.send_signal(signal.SIGUSR1)
                except OSError:
                    pass
        time.sleep(1.0)

    while True:
        states = [p.poll() for p in procs]
        bad = [s for s in states if s not in (None, 0)]
        if bad:
            code = bad[0] if bad[0] > 0 else 1
            print(f"[bench] a rank exited with status {bad[0]} after {time.time() - t0:.0f} s: stopping the run", file=sys.stderr)
            report_state()
            break
        if all(s == 0 for s in states):
            break
        if time.time() - t0 > timeout:
            code = EXIT_TIMEOUT
            print(f"[bench] ranks still running after {timeout:.0f} s: stopping the run", file=sys.stderr)
            report_state()
            break
        time.sleep(0.1)
    for p in procs:  # exact PIDs only
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=5)
    import shutil
    shutil.rmtree(run_dir, ignore_errors=True)
    if code:
        return code
    text = (out0[0] if out0 else b"").decode()
    line = next((ln for ln in reversed(text.splitlines()) if ln.startswith("{")), None)
    if line is None:
        print("[bench] rank 0 printed no result line", file=sys.stderr)
        return 1
    res = json.loads(line)
    if res.get("n_gpus") != n:
        print(f"[bench] result line reports n_gpus = {res.get('n_gpus')}, expected {n}", file=sys.stderr)
        return EXIT_RANKS
    print(line)
    return 0


# ---------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------
def run_rank(a, rank, world, local):
    # RCCL (its version banner) writes to the C-level stdout: file descriptor 1
    # points at stderr for the whole run and the result line goes out through a private duplicate of the real stdout, so that
    # the JSON line is the only thing this program writes there
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    stub = os.environ.get("RTD_BENCH_STUB") == "1"  # CPU test of the launch / control plane: no GPU, no librtd
    if stub and os.environ.get("RTD_BENCH_STUB_FAIL_RANK") == str(rank):
        sys.exit(7)  # test hook: a rank that dies before it joins
    if stub and os.environ.get("RTD_BENCH_STUB_HANG_RANK") == str(rank):
        phase("stub hang")  # test hook: a rank that never joins
        time.sleep(3600)
    cpu = None
    filled = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not stub:
        cpu = cpu_baseline_subprocess()
    live = None
    if rank == 0 and world == 1 and not a.no_extras and not a.no_live_traffic and not stub:
        live = live_traffic(a.columns)  # before this process touches the GPU (the profiler runs a child of its own)
        filled = filled_counters() if live else None

    first, C = shard_columns(rank, world, a.columns, a.total_columns)
    strong = a.total_columns > 0
    multi = world > 1 or a.force_dist
    arm_rank_deadline(rank, float(os.environ.get("RTD_BENCH_TIMEOUT", str(DEFAULT_TIMEOUT))))
    plan = None
    cfg = None
    mine = {"rank": rank, "local_rank": local, "pid": os.getpid(), "columns": C}  # this rank's own timings: where a poor curve comes from
    shared_gpu = bool(os.environ.get("RTD_RCCL_STUB")) and multi
    dev = local
    transport = None
    if not stub:
        phase("generate inputs")
        from pydisort_amd import synthetic
        from pydisort_amd import _engine
        from pydisort_amd._engine import Plan
        from pydisort_amd._prepare import prepare_columns
        ndev = _engine.device_count()  # hipGetDeviceCount: does not initialise a device
        if shared_gpu:
            # RTD_RCCL_STUB (tests only): every rank process on device 0 over the tests' stand-in transport -- RCCL refuses two
            # ranks on one GPU, and this is how the rank > 0 code of the data plane executes on a one-GPU box.  NOT a scaling run.
            dev = 0
            mine["device"] = 0
        elif ndev <= local:
            print(f"[bench] rank {rank}: LOCAL_RANK {local} but only {ndev} HIP device(s) visible: one rank per GPU is the "
                  "contract (check ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES and --gpus)", file=sys.stderr)
            sys.stderr.flush()
            os._exit(EXIT_RANKS)
        t0 = time.perf_counter()
        cfg, _, _ = shard_config(rank, world, a.columns, a.total_columns)
        mine["input_generation_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        prep = prepare_cfg4(cfg)
        mine["host_preparation_s"] = time.perf_counter() - t0
        if multi:
            Plan.comm_preload()  # RCCL from the ROCm install, bound to librtd's HIP runtime (the only one in this process)
        phase("create plan, upload inputs")
        t0 = time.perf_counter()
        plan = Plan(prep, device=dev, work_columns=a.columns)  # uploads: inputs now resident in HBM
        tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
        plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
        plan.synchronize()
        mine["plan_creation_and_upload_s"] = time.perf_counter() - t0

    ctl = None
    gather = None
    gather_calls = {}
    chosen = "none"
    collective = "none (single rank)"
    rccl_says = None
    watchdog = None
    if multi:
        # control plane: barriers, status and the max-over-ranks of the time -- sockets between the rank processes of this
        # node (pydisort_amd/_control.py), no PyTorch: librtd's HIP runtime and RCCL are the only GPU libraries of a rank
        phase("join the control plane")
        from pydisort_amd import _control
        t0 = time.perf_counter()
        try:
            ctl = _control.ControlPlane(rank, world)
        except _control.ControlError as e:
            print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(EXIT_RANKS)
        mine["control_plane_join_s"] = time.perf_counter() - t0
        if not stub:
            # data plane: RCCL all-gather of u + fluxes inside librtd.  A rank that cannot bootstrap or hangs ends the
            # whole run: the plan is never touched again after a failure, and a watchdog ends a rank stuck in RCCL.
            limit = float(os.environ.get("RTD_RCCL_TIMEOUT", "300"))

            def expired():
                print(f"[bench] rank {rank}: RCCL bootstrap / first gather still pending after {limit:.0f} s (phase '{_PHASE['name']}')", file=sys.stderr)
                import faulthandler
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                os._exit(EXIT_RCCL)

            watchdog = threading.Timer(limit, expired)
            watchdog.daemon = True
            watchdog.start()
            phase("RCCL unique id")
            uid, err = None, None
            if rank == 0:
                try:
                    uid = Plan.comm_unique_id()
                except Exception as e:
                    err = e
            uid = ctl.broadcast_bytes(uid, src=0)
            ok = uid is not None
            gather_calls = {"all": plan.allgather_results, "root": lambda: plan.gather_results(0), "none": None}
            if ok:
                try:
                    phase("ncclCommInitRank")
                    t0 = time.perf_counter()
                    plan.comm_init(uid, rank, world)
                    mine["rccl_comm_init_s"] = time.perf_counter() - t0
                    rccl_says = plan.comm_size()  # ncclCommCount / ncclCommUserRank / ncclCommCuDevice: RCCL's own statement
                    transport = plan.comm_transport()  # "rccl", or "stub:..." under RTD_RCCL_STUB
                    mine["transport"] = transport
                    if shared_gpu != transport.startswith("stub"):
                        raise RuntimeError(f"transport is {transport!r} with RTD_RCCL_STUB {'set' if shared_gpu else 'unset'}")
                    mine["rccl_nranks"], mine["rccl_rank"], mine["rccl_device"] = rccl_says
                    if rccl_says[0] != world or rccl_says[1] != rank:
                        raise RuntimeError(f"RCCL reports rank {rccl_says[1]} of {rccl_says[0]}, launched as rank {rank} of {world}")
                    t0 = time.perf_counter()
                    for mode in (("all", "root") if a.gather == "auto" else (a.gather,)):  # every collective the run may use, once
                        phase(f"first step + first collective ({mode})")
                        plan.run()
                        if gather_calls[mode]:
                            gather_calls[mode]()
                        plan.synchronize()
                    mine["first_collectives_s"] = time.perf_counter() - t0
                except Exception as e:
                    ok, err = False, e
            if not ok:
                print(f"[bench] rank {rank}: RCCL data plane failed: {err!r}", file=sys.stderr)
            phase("all ranks report their RCCL bootstrap")
            everyone = all_ranks_ok(ctl, ok)
            watchdog.cancel()
            if not everyone:
                sys.stderr.flush()
                os._exit(EXIT_RCCL)  # no clean-up through a communicator that may be half-built

    def barrier():
        if plan is not None:
            plan.synchronize()
        if ctl is not None:
            ctl.barrier()

    own = {}  # this rank's own seconds of the last timed region (the line reports the max over the ranks)

    def timed(nsteps, gather_fn, fresh, label=None):
        """nsteps steps between barriers + device synchronisation on both sides -> seconds, max over the ranks.
        fresh: every step treats the resident inputs as new (the per-column Legendre tables at -mu0 and the beam attenuations
        are recomputed: what the reference does in every call, _solve_for_gen_and_part_sols.py:96-109)."""
        if label:
            phase(label)
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            if plan is not None:
                if fresh:
                    plan.invalidate_tables()
                plan.run()
                if gather_fn:
                    gather_fn()
            else:
                time.sleep(0.001)
        if plan is not None:
            plan.synchronize()
        el = time.perf_counter() - t0
        barrier()
        own["seconds"] = el
        return reduce_max_seconds(ctl, el) if ctl is not None else el

    # which collective the timed region uses.  --gather auto (the default): ncclAllGather to every rank unless it costs the
    # step >= 10 % more than gathering on rank 0 alone (round 6: 3 % let two runs of one curve report different collectives) (SURVEY 8(e) allows root-only "if only rank 0 needs results"); decided
    # from max-over-ranks times of short trial regions, so every rank decides alike.
    gather_rates = {}
    probe = max(2, min(a.steps, 3))
    if multi and not stub:
        chosen = a.gather
        if a.gather == "auto":
            trial = {}
            for mode in ("all", "root"):
                timed(1, gather_calls[mode], True)
                trial[mode] = timed(probe, gather_calls[mode], True)
            chosen = "root" if trial["all"] > 1.10 * trial["root"] else "all"
        gather = gather_calls[chosen]
        collective = {
            "all": f"rccl ncclAllGather of u + fluxes per step, nranks = {world}, on its own stream (overlaps the next step)",
            "root": f"rccl ncclSend/ncclRecv of u + fluxes to rank 0 per step, nranks = {world}, on its own stream",
            "none": f"rccl communicator of {world} ranks initialised, no data-path collective (--gather none)"}[chosen]
        if a.gather == "auto":
            collective += " [--gather auto: chosen from trial regions, all-gather unless >= 10 % slower than root-only]"

    timed(a.warmup, gather, True, "warm-up steps") if a.warmup > 0 else barrier()
    elapsed = timed(a.steps, gather, True, "timed region")          # THE timed region: exactly --steps steps, fresh inputs every step
    mine["ms_per_step_own"] = 1e3 * own["seconds"] / a.steps
    elapsed_cached = timed(a.steps, gather, False, "timed region (cached tables)")  # the same on repeated inputs (tables kept from run to run)
    mine["ms_per_step_own_cached_tables"] = 1e3 * own["seconds"] / a.steps
    joined = world
    if ctl is not None:
        total_cols = int(ctl.allreduce(int(C), "sum"))
        joined = int(ctl.allreduce(1, "sum"))
    else:
        total_cols = C
    if joined != world:
        print(f"[bench] only {joined} of {world} ranks took part", file=sys.stderr)
        os._exit(EXIT_RANKS)

    # N > 1 (or --force-dist): the gathered arrays are CHECKED before any number is reported -- every rank that holds them
    # compares >= 4 columns of every rank's slot with a local solve of the same columns, bit for bit -- and the compute-only and
    # per-collective rates of the same run go into the line, so that one scaling run separates compute scaling from the
    # cost of the collective.
    verified = None
    if multi and not stub:
        holds = chosen == "all" or (chosen == "root" and rank == 0)
        ok, detail = True, None
        if chosen != "none":
            timed(1, gather, True, "gather to be verified")  # (a fresh gather of a known step; its results are what is checked)
            phase("verify the gathered arrays against local solves")
            if holds:
                try:
                    ok, nver, detail = verify_gathered(plan, dev, world, a.columns, a.total_columns)
                except Exception as e:
                    ok, detail = False, repr(e)
                if not ok:
                    print(f"[bench] rank {rank}: gathered results differ from a local solve of the same columns: {detail}", file=sys.stderr)
            if not all_ranks_ok(ctl, ok):
                sys.stderr.flush()
                os._exit(EXIT_VERIFY)
            verified = {"gather_verified": True, "ranks_verified": world, "columns_per_rank": 4,
                        "verified_on": "every rank" if chosen == "all" else "rank 0", "how": "bit-for-bit against a local one-window solve of "
                        "the same regenerated columns (u and the three fluxes)", "detail": detail}
        for mode in ("none", "all", "root"):
            if mode == chosen:
                gather_rates[mode] = total_cols * a.steps / elapsed
            elif mode in gather_calls and (a.gather == "auto" or mode == "none"):
                timed(1, gather_calls[mode], True, f"rate with --gather {mode}")
                gather_rates[mode] = total_cols * probe / timed(probe, gather_calls[mode], True)
                mine[f"ms_per_step_own_gather_{mode}"] = 1e3 * own["seconds"] / probe

    # per-kernel HIP-event times from a separate short pass (events + a stream sync per window would otherwise sit
    # inside the timed region; the timed region above is the free-running pipeline)
    stage, sweeps = None, None
    per_rank = ctl.gather(mine) if ctl is not None else [mine]
    phase("HIP-event pass, extras")
    if plan is not None and rank == 0:
        plan.enable_timing(True)
        plan.timing(reset=True)
        for _ in range(max(2, min(a.steps, 5))):
            plan.run()
        stage = plan.timing(reset=True)
        plan.enable_timing(False)
        sweeps = plan.max_sweeps()

    extras = {}
    if rank == 0 and world == 1 and not a.no_extras and not stub:
        plan.close()  # the extras build their own plans: give the arena back first
        plan = None
        extras = extra_measurements(dev, cfg if strong else None, a.columns, live, filled)
    if rank == 0:
        value = total_cols * a.steps / elapsed
        out = {
            "metric": "column-solves/sec (32 streams, 20 layers)", "value": value, "unit": "column-solves/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
            "timed_seconds": elapsed,
            "value_means": "fresh inputs every step: inputs resident in HBM, nothing kept from step to step -- the per-column "
                           "Legendre tables at -mu0 and the beam attenuations are recomputed like everything else (the reference "
                           "computes them in every call)",
            "value_cached_tables": total_cols * a.steps / elapsed_cached,
            "value_cached_tables_means": "the same steps on REPEATED inputs: a plan keeps those per-column tables while its inputs are "
                                         "unchanged (rtd_tables_mu0_kernel runs once; a serving optimisation, not the headline)",
            "data": "synthetic",
            "config": {"workload": "cfg4: synthetic Henyey-Greenstein, 20 layers, 32 streams, 32 Fourier modes, "
                                   "delta-M on, beam source, u at 21 interfaces x 3 azimuths + fluxes",
                       "columns_per_gpu_per_step": C, "global_columns_per_step": total_cols,
                       "columns_per_window": a.columns, "ranks_joined": joined,
                       "rccl_nranks": rccl_says[0] if rccl_says else None,
                       "total_columns": a.total_columns if strong else None,
                       "parallelism": f"column-sharded x{world}", "collective": collective,
                       "max_jacobi_sweeps": sweeps},
        }
        if verified is not None:
            out["config"].update(gather_verified=verified["gather_verified"], ranks_verified=verified["ranks_verified"])
            out["gather_verification"] = verified
        elif multi and not stub:
            out["config"].update(gather_verified=None, ranks_verified=0)  # --gather none: nothing is gathered
        if multi and not stub:
            out["transport"] = transport  # what carried the collectives: "rccl" -- or the tests' stand-in
            if shared_gpu:
                out["devices_used"] = 1
                out["not_a_rate"] = True
                out["transport_note"] = ("RTD_RCCL_STUB: every rank process ran on device 0 over the tests' stand-in for RCCL (tests/stub/"
                                         "rccl_stub.cpp: hipIpc / shared memory, host-synchronous).  This line proves that the N-rank data "
                                         "plane EXECUTES and that what it gathers is right; `value` is N processes sharing one GPU through a "
                                         "synchronous transport and must not be quoted as a rate or a scaling point.")
        if multi:
            out["control_plane"] = "sockets between the rank processes (pydisort_amd/_control.py); torch imported: " + str("torch" in sys.modules)
            out["per_rank"] = per_rank  # every rank's own seconds: input generation, plan creation + upload, RCCL bootstrap, its own ms per step
            if rccl_says:
                out["rccl"] = {"nranks": rccl_says[0], "rank_of_this_line": rccl_says[1], "device": rccl_says[2],
                               "source": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice on the plan's communicator (rtd_comm_size); "
                                         "every rank's own answer is in per_rank"}
        if gather_rates:
            out["gather_rates"] = {"unit": "column-solves/sec, whole job", "chosen": chosen, "compute_only": gather_rates.get("none"),
                                   "allgather": gather_rates.get("all"), "root_only": gather_rates.get("root"),
                                   "what": "the same run with no data-path collective, with ncclAllGather to every rank and with "
                                           "ncclSend/ncclRecv to rank 0: compute scaling and the cost of the collective, separately"}
        if stage is not None:
            fl = algorithmic_flops()
            nwin = max(1, -(-C // a.columns))
            fused_bc = stage["iface"][0] < 0.05 * stage["sweep"][0]  # NQuad = 32: one fused kernel, timed in the sweep slot
            names = {"eigen": "rtd_eigen_kernel<16, 2>",
                     "bc": "rtd_bc_mfma_kernel" if fused_bc else "rtd_bc_tile_kernel<1>"}
            cols_per_launch = C / nwin  # average over the windows of a step (the last one may be short)
            roof, ms = roofline_of(stage, fl, cols_per_launch, names)
            tkey = {"rtd_eigen_kernel<16, 2>": "rtd_eigen_kernel", "rtd_bc_mfma_kernel": "rtd_bc_mfma_kernel"}.get(roof["kernel"], "rtd_sweep_kernel")
            live_main = None
            if live:  # the headline config's kernels among everything the profiler passes saw
                live_main = {k: live[n] for k, n in (("rtd_eigen_kernel", "rtd_eigen_kernel<16,2>"), ("rtd_bc_mfma_kernel", "rtd_bc_mfma_kernel"),
                                                     ("rtd_fourier_kernel", "rtd_fourier_kernel<16>")) if n in live}
            roof["traffic"] = live_main.get(tkey) if live_main and tkey in live_main else measured_traffic(tkey, a.columns)
            roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation (child process, "
                                      "32 windows, windows one after the other)" if live_main and tkey in live_main else
                                      "profiles/r06_pmc_traffic.json (committed passes)")
            if live_main:
                roof["traffic_all_kernels_per_window"] = float(sum(live_main.values()))
                ok_ = roof.get("other_kernel")
                if ok_ and ok_.get("kernel") in live_main and ok_.get("ms_per_launch"):
                    # the second kernel of the step against ITS roofline: it moves its bytes at a rate near what the memory system gives
                    # (MI355X_MICROARCH.md: 6.29 TB/s attainable of the 8 TB/s peak) while its FP64 units are a quarter busy
                    rate = live_main[ok_["kernel"]] / (ok_["ms_per_launch"] * 1e-3) / 1e12
                    ok_.update(hbm_bytes_per_launch=live_main[ok_["kernel"]], hbm_TB_per_s=rate, hbm_frac_of_8_TB_per_s=rate / 8.0,
                               bound="hbm + latency of its dependent chains: see DESIGN.md section 4")
            roof["launches_per_step"] = nwin
            roof["whole_path_tflops"] = fl["total"] * value / world / 1e12
            roof["whole_path_frac"] = fl["total"] * value / world / 1e12 / FP64_PEAK_TFLOPS
            roof["note"] = ("FP64 path (SURVEY 8(d): compute-bound, not HBM-bound); the dominant kernel issues FP64 vector "
                            "instructions (the matrix pipe has the same FP64 peak): peak = MI355X FP64 vector = matrix peak; "
                            "achieved = algorithmic FLOPs of the kernel x columns per launch / its HIP-event duration (separate "
                            "timing pass after the timed region, windows one after the other; the timed region itself runs the "
                            "eigen kernel of window w + 1 beside the boundary-condition kernel of window w on two streams); "
                            "traffic = HBM bytes per launch of that kernel: see traffic_source")
            out["roofline"] = roof
            ev = north_star_evidence(a.columns, elapsed / a.steps / max(nwin, 1), live_main)
            if ev:
                out["measured_hbm_and_mfma"] = ev
        out["cpu_baseline"] = cpu
        if not multi:
            out["multi_gpu"] = {
                "this_line": "one GPU; the scaling curve (N = 1, 2, 4, 8 over RCCL / xGMI) is the driver's to measure: python -m torch.distributed.run "
                             "--nproc-per-node N bench.py --gpus N (or bench.py --gpus N alone: it starts the ranks itself)",
                "n_rank_data_plane_executed": "with 2, 4 and 8 rank processes on ONE GPU over the tests' stand-in transport (tests/stub/rccl_stub.cpp, "
                                              "RTD_RCCL_STUB): tests/test_gpu_multi_gpu.py -- all four partitions, bench.py --gpus 8 and the driver's "
                                              "torch.distributed.run command line end to end, every gathered slot verified bit for bit; evidence "
                                              "profiles/r06_bench_stub_8ranks.json (transport = stub, not_a_rate = true: never a throughput)",
                "over_rccl": "one-rank communicator on every box (tests/test_gpu_distributed.py); two-rank tests where two GPUs are visible"}
            if not a.no_extras and not stub:
                out["multi_gpu"]["n_rank_check"] = n_rank_check()
        out.update(extras)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    phase("leave")
    if ctl is not None:
        ctl.barrier()
        ctl.close()
    if plan is not None:
        plan.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--columns", type=int, default=256,
                    help="columns per GPU per step (weak scaling); with --total-columns: columns per window")
    ap.add_argument("--total-columns", type=int, default=100_000,
                    help="strong scaling (default, BASELINE's literal batch: 100000): this many columns in total per step, "
                         "split over the GPUs; 0 = weak scaling, --columns per GPU per step")
    ap.add_argument("--gather", choices=("auto", "all", "root", "none"), default="auto",
                    help="N > 1: results of a step to every rank (ncclAllGather), to rank 0 only (ncclSend/ncclRecv), or nowhere; "
                         "auto (default): all-gather unless trial regions show it >= 10 %% slower than root-only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the parity, only_flux and end-to-end legs (profiling runs: every kernel launch is then the workload)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the multi-rank code path even with one rank")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the committed PMC passes instead of two rocprofv3 passes of a child run (~20 s)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child-filled", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.pmc_child:  # child of live_traffic(), under rocprofv3: the workload only
        pmc_child(a.columns)
        return
    if a.pmc_child_filled:  # child of filled_counters()
        pmc_child_filled()
        return
    if a.cpu_baseline_only:  # child of cpu_baseline_subprocess(): no GPU, one JSON line
        res = cpu_baseline()
        res["_samples"] = [(int(f), u.tolist()) for f, u in _ORACLE_SAMPLES]
        print(json.dumps(res))
        return
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))
        if a.force_dist:  # one rank, but through the same rendezvous as N ranks
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE = {world}: launch one rank per GPU "
              f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...) or let bench.py "
              "start the ranks itself by running it without RANK/WORLD_SIZE in the environment", file=sys.stderr)
        sys.exit(EXIT_RANKS)
    run_rank(a, rank, world, local)


if __name__ == "__main__":
    main()
