"""Hardware-facing readiness of the multi-GPU path on the one GPU a test box has: bench.py through the SAME code N ranks
take -- socket control plane (no PyTorch), RCCL from the ROCm install, communicator bootstrap under the watchdog, the data-path collective on
its own stream overlapped with the next step -- with a communicator of one rank (`--force-dist`), as a fresh child process
like the driver's.  (N > 1 itself: world-size-2 control-plane tests in test_distributed_cpu.py; the driver's SCALE run.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=900)


@pytest.mark.parametrize("gather,prefix", [("all", "rccl ncclAllGather"), ("root", "rccl ncclSend/ncclRecv"), ("none", "rccl communicator")])
def test_bench_force_dist_runs_the_rccl_path_with_one_rank(gather, prefix):
    r = _bench(["--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                "--total-columns", "6000", "--gather", gather])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # the JSON line is the only thing on stdout (RCCL banners go to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["ranks_joined"] == 1
    assert out["config"]["collective"].startswith(prefix), out["config"]["collective"]
    assert out["scaling"] == "strong" and out["config"]["global_columns_per_step"] == 6000
    assert out["config"]["columns_per_window"] == 256 and out["roofline"]["launches_per_step"] == 24
    assert out["value"] > 1e4 and 0.05 < out["roofline"]["frac"] < 1.0
    assert out["config"]["rccl_nranks"] == 1 and out["per_rank"][0]["rccl_comm_init_s"] > 0      # ncclCommCount's answer, not the caller's argument
    assert out["control_plane"].endswith("torch imported: False")                               # north_star: no PyTorch


def test_gathered_results_on_the_root_equal_the_ranks_own():
    """rtd_comm_gather_results (ncclSend / ncclRecv to one rank) with a one-rank communicator: the gathered arrays are the
    rank's own results, bit for bit, like the all-gather's."""
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    import pydisort_amd
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    cfg = synthetic.cfg4_columns(8, L=5, NQuad=8)
    Plan.comm_preload()
    _, sol = pydisort_amd.pydisort_batch(**cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((8, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, 2.0]))
    plan.comm_init(Plan.comm_unique_id(), 0, 1)
    plan.run()
    plan.gather_results(0)
    plan.run()            # the next step overlaps the gather (its evaluation waits for it)
    plan.synchronize()
    gu, gf = plan.fetch_gathered_results()
    res = plan.fetch()
    assert np.array_equal(gu, res["u"]) and np.array_equal(gf[0, 0], res["flux_up"])
    assert np.array_equal(gf[0, 1], res["flux_down_diffuse"]) and np.array_equal(gf[0, 2], res["flux_down_direct"])
    with pytest.raises(RuntimeError):
        plan.gather_results(1)  # root outside the communicator
    plan.close()


def test_bench_refuses_a_rank_without_a_gpu_of_its_own():
    """One rank per GPU is the contract: a rank whose LOCAL_RANK has no device exits 2 with a message instead of sharing
    GPU 0 (which would report N x the throughput of one GPU as "scaling")."""
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--total-columns", "64"],
               RANK="0", LOCAL_RANK="63", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "HIP device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
