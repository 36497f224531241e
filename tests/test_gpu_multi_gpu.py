"""N > 1: every partition of SURVEY 8(e), the gathered / reduced / stitched results against a single-rank solve.

* over RCCL, TWO ranks, one process per GPU -- only where >= 2 HIP devices are visible (the driver's 8-GPU node; skipped on the
  builder's one-GPU boxes);
* over the tests' STAND-IN transport (tests/stub/rccl_stub.cpp, selected by RTD_RCCL_STUB), 2 and 8 rank processes that
  share device 0 -- on every box.  RCCL refuses a second rank on a device, so this is what executes the rank > 0 code of
  rtd_comm_* (slot offsets, the root's receive loop, the mode all-reduce, the layer stitch) on one GPU.  It proves the data
  plane's logic, not RCCL's transport, and nothing measured through it is a rate;
* the SAME worker with one rank over RCCL on every box (process start, id hand-over, collectives, comparison)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
MODES = ("allgather", "root", "modes", "layers")
STUB = os.path.join(ROOT, "tests", "stub", "librccl_stub.so")


def _devices():
    from pydisort_amd import _engine
    return _engine.device_count()


def _clean_env(stub=False, **over):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RTD_RCCL_STUB")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if stub:
        if not os.path.exists(STUB):  # (__graft_entry__.build() makes it; a box that got the sources only builds it here: hipcc is in the image)
            subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stub", "build_stub.py")], check=True, timeout=600)
        assert os.path.exists(STUB), f"{STUB} not built (python tests/stub/build_stub.py; __graft_entry__.build() does it)"
        env["RTD_RCCL_STUB"] = STUB
        env.setdefault("RCCL_STUB_TIMEOUT_S", "240")
    env.update(over)
    return env


def _run(mode, world, same_device=False, stub=False, **env_over):
    env = _clean_env(stub, **env_over)
    with tempfile.TemporaryDirectory(prefix="rtd_dist_") as d:
        # every rank writes to files of its own: with pipes drained one process at a time, a chatty rank (NCCL_DEBUG=INFO,
        # RTD_DEBUG) fills its 64 KB pipe while another is waited for, blocks in write() and hangs the collective
        logs = [(open(os.path.join(d, f"rank_{r}.out"), "w+"), open(os.path.join(d, f"rank_{r}.err"), "w+")) for r in range(world)]
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--rank", str(r),
                                   "--world", str(world), "--dir", d] + (["--device", "0"] if same_device else []),
                                  env=env, stdout=logs[r][0], stderr=logs[r][1], text=True) for r in range(world)]
        outs = []
        try:
            deadline = time.monotonic() + 600
            for p in procs:
                p.wait(timeout=max(1.0, deadline - time.monotonic()))
        finally:
            for p in procs:  # exact PIDs only
                if p.poll() is None:
                    p.kill()
                    p.wait()
            for fo, fe in logs:
                texts = []
                for f in (fo, fe):
                    f.flush()
                    f.seek(0)
                    texts.append(f.read())
                    f.close()
                outs.append(tuple(texts))
        res = []
        for r, p in enumerate(procs):
            path = os.path.join(d, f"result_{r}.json")
            assert os.path.exists(path), f"rank {r} (exit {p.returncode}) left no result: {outs[r][1][-2000:]}"
            res.append(json.load(open(path)))
        return procs, res, outs


@pytest.mark.parametrize("mode", MODES)
def test_every_partition_with_one_rank(mode):
    """The worker with a communicator of one rank: process start, id hand-over, collectives, comparison -- on any box."""
    procs, res, outs = _run(mode, 1)
    assert procs[0].returncode == 0 and res[0]["ok"], (res, outs[0][1][-2000:])
    assert res[0]["checks"]["transport"] == "rccl"


@pytest.mark.parametrize("mode", MODES)
def test_every_partition_with_two_ranks_on_two_gpus(mode):
    """Two ranks, two GPUs: column shards + ncclAllGather (every rank checks the gathered arrays of BOTH ranks against one
    plan's solve of the union, bit for bit), column shards + ncclSend/ncclRecv to the root (the root checks; the other rank
    must be refused a fetch), Fourier-mode shards + ncclAllReduce against the unsharded solve, layer shards + the ONE
    all-gather that stitches the boundary-condition system against the one-piece solve."""
    if _devices() < 2:
        pytest.skip("needs >= 2 HIP devices (runs on the multi-GPU node)")
    procs, res, outs = _run(mode, 2)
    for r in range(2):
        assert procs[r].returncode == 0 and res[r]["ok"], (r, res[r], outs[r][1][-2000:])
    if mode == "allgather":
        assert all(res[r]["checks"]["q32"]["bit_equal"] and res[r]["checks"]["q8"]["bit_equal"] for r in range(2))
    if mode == "root":
        assert res[0]["checks"]["q32"]["bit_equal"] and res[1]["checks"]["q32"]["non_root_fetch_refused"]


@pytest.mark.parametrize("world", (2, 8))
@pytest.mark.parametrize("mode", MODES)
def test_every_partition_over_the_stub_transport_on_one_gpu(mode, world):
    """2 and 8 rank processes on device 0 over the stand-in transport: the same worker, the same comparisons as the two-GPU
    RCCL test -- every rank > 0 slot offset, the root's receive loop over 7 senders, an all-reduce of 8 mode shards, a
    boundary-condition system stitched from 8 layer shards."""
    procs, res, outs = _run(mode, world, same_device=True, stub=True)
    for r in range(world):
        assert procs[r].returncode == 0 and res[r]["ok"], (r, res[r], outs[r][1][-2000:])
        assert res[r]["checks"]["transport"].startswith("stub:"), res[r]["checks"]
    if mode == "allgather":
        assert all(res[r]["checks"]["q32"]["bit_equal"] and res[r]["checks"]["q8"]["bit_equal"] for r in range(world))
        assert all(res[r]["checks"]["q32_rccl_says"] == [world, r, 0] for r in range(world))
    if mode == "root":
        assert res[0]["checks"]["q32"]["bit_equal"] and res[0]["checks"]["q8"]["bit_equal"]
        assert all(res[r]["checks"]["q32"]["non_root_fetch_refused"] for r in range(1, world))


def test_stub_transport_through_shared_memory_staging():
    """The stand-in's second transport (host staging through /dev/shm instead of hipIpc): the all-gather and the root gather
    with three ranks -- what the tests fall back to on a box whose driver does not hand out IPC handles."""
    for mode in ("allgather", "root", "modes"):
        procs, res, outs = _run(mode, 3, same_device=True, stub=True, RCCL_STUB_TRANSPORT="shm")
        for r in range(3):
            assert procs[r].returncode == 0 and res[r]["ok"], (mode, r, res[r], outs[r][1][-2000:])
            assert res[r]["checks"]["transport"] == "stub:shm"


def test_bench_with_eight_ranks_over_the_stub_transport():
    """bench.py --gpus 8 end to end on ONE GPU (RTD_RCCL_STUB): 8 rank processes, socket control plane, communicator of 8,
    --gather auto (both collectives tried), every rank verifies all 8 slots of what it gathered bit for bit, per-rank records.
    The line says transport = stub and not_a_rate: it is evidence that the N-rank path executes, never a throughput."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-extras", "--total-columns", "8192"], env=_clean_env(True), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["transport"].startswith("stub:") and out["not_a_rate"] is True and out["devices_used"] == 1
    assert out["config"]["gather_verified"] is True and out["config"]["ranks_verified"] == 8 and out["config"]["rccl_nranks"] == 8
    assert [d["rank"] for d in out["gather_verification"]["detail"]] == list(range(8))
    assert all(d["mismatches"] == 0 and len(d["columns"]) >= 4 for d in out["gather_verification"]["detail"])
    assert [p["rank"] for p in out["per_rank"]] == list(range(8))
    assert all(p["rccl_nranks"] == 8 and p["rccl_rank"] == p["rank"] and p["rccl_device"] == 0 and p["columns"] == 1024 for p in out["per_rank"])
    g = out["gather_rates"]
    assert g["compute_only"] > 0 and g["allgather"] > 0 and g["root_only"] > 0 and g["chosen"] in ("all", "root")
    assert out["control_plane"].endswith("torch imported: False")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_stub_8ranks.json"), "w") as f:
        f.write(json.dumps(out) + "\n")
    # --gather auto may settle on the root-only collective (then rank 0 alone holds and verifies the arrays): the all-gather
    # form explicitly, where EVERY one of the 8 ranks checks all 8 slots of its own copy
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-extras", "--total-columns", "4096", "--gather", "all"], env=_clean_env(True), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["transport"].startswith("stub:") and out["gather_verification"]["verified_on"] == "every rank"
    assert out["config"]["ranks_verified"] == 8 and out["config"]["collective"].startswith("rccl ncclAllGather")


def test_bench_under_the_drivers_own_launcher_with_four_ranks_over_the_stub_transport():
    """The driver's N > 1 command line, verbatim -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with the GPU data plane underneath (4 rank processes on
    device 0 over the stand-in transport): the launcher is PyTorch's, its store holds MASTER_PORT, the ranks meet over the socket
    control plane without importing torch, build a communicator of 4, gather and verify."""
    pytest.importorskip("torch")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras", "--total-columns", "4096"], env=_clean_env(True), capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["steps"] == 2 and out["warmup"] == 1 and out["transport"].startswith("stub:") and out["not_a_rate"] is True
    assert out["config"]["rccl_nranks"] == 4 and out["config"]["gather_verified"] is True and out["config"]["ranks_verified"] == 4
    assert out["config"]["global_columns_per_step"] == 4096 and [p["rank"] for p in out["per_rank"]] == [0, 1, 2, 3]
    assert out["control_plane"].endswith("torch imported: False")


def test_bench_verifies_what_it_gathers_before_it_reports():
    """bench.py --gpus 2: the N > 1 line carries gather_verified / ranks_verified = 2 and the compute-only, all-gather and
    root-only rates of the same run (>= 2 devices); a one-rank --force-dist run carries the same fields on any box."""
    n = 2 if _devices() >= 2 else 1
    args = ["--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--total-columns", "3000"]
    if n == 1:
        args.append("--force-dist")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == n and out["config"]["gather_verified"] is True and out["config"]["ranks_verified"] == n
    assert all(d["mismatches"] == 0 and len(d["columns"]) >= 4 for d in out["gather_verification"]["detail"])
    g = out["gather_rates"]
    assert g["compute_only"] > 0 and g["allgather"] > 0 and g["root_only"] > 0 and g["chosen"] in ("all", "root")
    assert out["value_cached_tables"] > 0
    # RCCL's own statement (ncclCommCount on every rank's communicator), the torch-free control plane, every rank's own timings
    assert out["config"]["rccl_nranks"] == n and out["rccl"]["nranks"] == n
    assert [p["rank"] for p in out["per_rank"]] == list(range(n))
    assert all(p["rccl_nranks"] == n and p["rccl_rank"] == p["rank"] and p["rccl_device"] == p["local_rank"] for p in out["per_rank"])
    assert all(p["ms_per_step_own"] > 0 and p["input_generation_s"] > 0 and p["plan_creation_and_upload_s"] > 0 for p in out["per_rank"])
    assert out["control_plane"].endswith("torch imported: False")
    assert out["transport"] == "rccl" and "not_a_rate" not in out
