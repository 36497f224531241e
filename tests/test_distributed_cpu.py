"""N > 1 paths of bench.py on CPU (world_size 2, no PyTorch anywhere: the control plane is pydisort_amd/_control.py).

* the control plane two ranks use: disjoint column shards, the max-over-ranks time, the all-ranks status flag, the
  broadcast that carries the RCCL unique id, the gather of the per-rank records; a rank that leaves or falls out of step is
  NAMED by the others instead of hanging them;
* the launcher: `bench.py --gpus 2` itself starts two fresh rank processes, both join the rendezvous and contribute
  to the result line (a no-GPU stub stands in for the device work) -- also with `torch` made un-importable, and under the
  environment torch.distributed.run gives its workers; a rank that dies, a run that overruns its time limit, or a
  WORLD_SIZE that does not match --gpus ends the run with a non-zero exit code and says where every rank was.
(The RCCL collectives themselves: tests/test_gpu_multi_gpu.py with two ranks where two GPUs exist, one rank elsewhere.)"""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, key, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    import bench
    from pydisort_amd import _control, synthetic
    ctl = _control.ControlPlane(rank, world, key=key, timeout=60)
    first, C = bench.shard_columns(rank, world, 3)
    sfirst, sC = bench.shard_columns(rank, world, 3, total_columns=11)
    cfg = synthetic.cfg4_columns(C, first=first, L=4, NQuad=8)
    uid = ctl.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
    nothing = ctl.broadcast_bytes(None)              # rank 0 had nothing to send: None everywhere, no hang
    elapsed = bench.reduce_max_seconds(ctl, 1.0 + rank)
    all_ok = bench.all_ranks_ok(ctl, True)
    one_bad = bench.all_ranks_ok(ctl, rank != 1)
    total = ctl.allreduce(C, "sum")
    recs = ctl.gather({"rank": rank, "x": 10 * rank})
    ctl.barrier()
    q.put((rank, first, C, cfg["tau_arr"].sum(), uid == bytes(range(128)), elapsed, all_ok, one_bad, sfirst, sC, total, recs, nothing,
           "torch" in sys.modules))
    ctl.close()


def test_two_rank_sharding_and_timing_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"rtd-test-{os.getpid()}-{_free_port()}"
    procs = [ctx.Process(target=_worker, args=(r, world, key, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, c0, s0, n0, e0, a0, b0, sf0, sc0, t0, g0, x0, th0), (r1, f1, c1, s1, n1, e1, a1, b1, sf1, sc1, t1, g1, x1, th1) = out
    assert (f0, c0, f1, c1) == (0, 3, 3, 3)           # disjoint, contiguous shards
    assert (sf0, sc0, sf1, sc1) == (0, 5, 5, 5)       # strong scaling: equal shares of the total
    assert n0 and n1                                  # unique id reached every rank, byte for byte
    assert x0 is None and x1 is None
    assert e0 == e1 == 2.0                            # max over ranks
    assert a0 and a1 and not b0 and not b1            # one failing rank is seen by every rank
    assert t0 == t1 == 6
    assert g0 == [{"rank": 0, "x": 0}, {"rank": 1, "x": 10}] and g1 is None   # per-rank records, in rank order, on rank 0
    assert not th0 and not th1                        # the control plane never imports torch
    sys.path[:0] = [os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import synthetic
    whole = synthetic.cfg4_columns(6, L=4, NQuad=8)["tau_arr"]
    assert np.isclose(s0, whole[:3].sum()) and np.isclose(s1, whole[3:].sum())  # union == global batch


def _leaver(rank, world, key, q, how):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import _control
    ctl = _control.ControlPlane(rank, world, key=key, timeout=20)
    ctl.barrier()
    try:
        if rank == 1 and how == "leaves":
            os._exit(0)                               # dies between two collectives
        if rank == 1 and how == "out_of_step":
            ctl.allreduce(1.0, "max")                 # rank 0 is in a barrier
        else:
            ctl.barrier()
        q.put((rank, "no error"))
    except _control.ControlError as e:
        q.put((rank, str(e)))


@pytest.mark.parametrize("how,expect", [("leaves", "rank 1 left"), ("out_of_step", "rank 1 is at allreduce_max #2, rank 0 at barrier #2")])
def test_a_rank_that_leaves_or_falls_out_of_step_is_named(how, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"rtd-test-{os.getpid()}-{_free_port()}"
    procs = [ctx.Process(target=_leaver, args=(r, 2, key, q, how)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(1 if how == "leaves" else 2))
    for p in procs:
        p.join(timeout=30)
    assert expect in got[0], got


def _stray_then_join(key, q):
    """A local process that is not a rank of the run: connects and says nothing, connects and says nonsense, connects as a rank
    of a different world -- then the real rank 1 joins."""
    import socket
    import struct
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import _control
    addr = _control._address(key)
    deadline = time.time() + 20
    strays = []
    while time.time() < deadline:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            s.connect(addr)
            strays.append(s)
            break
        except (FileNotFoundError, ConnectionRefusedError):
            s.close()
            time.sleep(0.02)
    silent = strays[0]                                        # never says hello
    junk = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    junk.connect(addr)
    junk.sendall(struct.pack("<I", 5) + b"\xff\xfenot")       # not JSON
    other = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    other.connect(addr)
    _control._send(other, {"rank": 1, "world": 7})            # a rank of some other run
    ctl = _control.ControlPlane(1, 2, key=key, timeout=30)
    q.put(("joined", ctl.allreduce(5, "sum")))
    ctl.close()
    for s in (silent, junk, other):
        s.close()


def test_stray_connections_do_not_stall_or_abort_the_join():
    """round-5 advice: rank 0 used to block on the first connection's hello and to abort the run on a malformed one.  A silent
    connection, a garbage one and a hello of another world are dropped (the silent one after 5 s); the real rank joins."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import _control
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"rtd-test-{os.getpid()}-{_free_port()}"
    p = ctx.Process(target=_stray_then_join, args=(key, q))
    p.start()
    t0 = time.time()
    ctl = _control.ControlPlane(0, 2, key=key, timeout=30)
    assert ctl.allreduce(3, "sum") == 8
    assert q.get(timeout=30) == ("joined", 8)
    ctl.close()
    p.join(timeout=30)
    assert p.exitcode == 0 and time.time() - t0 < 25


def test_two_launches_without_a_master_port_do_not_share_a_rendezvous():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import _control
    a = _control.rendezvous_key({"RTD_BENCH_RUN_DIR": "/tmp/rtd_bench_run_aaa"})
    b = _control.rendezvous_key({"RTD_BENCH_RUN_DIR": "/tmp/rtd_bench_run_bbb"})
    assert a != b and f"ppid{os.getppid()}" in a
    assert _control.rendezvous_key({"MASTER_PORT": "29500"}) == _control.rendezvous_key({"MASTER_PORT": "29500"})


def test_ranks_that_never_join_are_listed():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import _control
    with pytest.raises(_control.ControlError, match=r"ranks \[1, 2\] of 3 never joined"):
        _control.ControlPlane(0, 3, key=f"rtd-test-{os.getpid()}-{_free_port()}", timeout=1.0)


def _bench(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTD_BENCH_STUB="1", RTD_BENCH_TIMEOUT="240")
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_2_starts_two_ranks_that_both_contribute():
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--columns", "3", "--total-columns", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                             # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_joined"] == 2
    assert out["config"]["global_columns_per_step"] == 6 and out["config"]["columns_per_gpu_per_step"] == 3
    assert out["scaling"] == "weak" and out["steps"] == 2 and out["warmup"] == 1
    assert out["value"] > 0 and out["higher_is_better"] is True


def test_bench_strong_scaling_splits_the_total():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--total-columns", "10", "--columns", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["global_columns_per_step"] == 10
    assert out["config"]["columns_per_gpu_per_step"] == 5 and out["n_gpus"] == 2


def test_bench_fails_loudly_when_a_rank_dies():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--columns", "3", "--total-columns", "0"], RTD_BENCH_STUB_FAIL_RANK="1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # no result line from a broken run


def test_bench_defaults_to_baselines_literal_batch():
    """No flags: BASELINE's 10^5 columns, strong scaling -- 50 000 per rank at two ranks, windows of 256."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["global_columns_per_step"] == 100_000
    assert out["config"]["columns_per_gpu_per_step"] == 50_000 and out["config"]["columns_per_window"] == 256


def test_bench_runs_without_torch():
    """north_star: "no PyTorch".  `torch` is made un-importable for the launcher and both ranks (a package of that name that
    raises on import comes first on sys.path); the run must not notice."""
    with tempfile.TemporaryDirectory() as d:
        os.mkdir(os.path.join(d, "torch"))
        with open(os.path.join(d, "torch", "__init__.py"), "w") as f:
            f.write("raise ImportError('torch is not available in this test')\n")
        r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--columns", "3", "--total-columns", "0"],
                   PYTHONPATH=d + os.pathsep + os.environ.get("PYTHONPATH", ""))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_joined"] == 2
    assert out["control_plane"].endswith("torch imported: False")
    assert [p["rank"] for p in out["per_rank"]] == [0, 1] and all("ms_per_step_own" in p for p in out["per_rank"])


def test_bench_under_the_environment_of_torch_distributed_run():
    """The driver launches N > 1 as `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N`: the workers get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (+ TORCHELASTIC_RUN_ID), and the
    launcher's own store LISTENS on MASTER_PORT -- the control plane must meet without binding it."""
    port = _free_port()
    busy = socket.socket()
    busy.bind(("127.0.0.1", port))
    busy.listen(1)                                    # the port is taken, as it is under torch.distributed.run
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTD_BENCH_STUB="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="none")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                               "--total-columns", "10", "--columns", "4"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    busy.close()
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_columns_per_step"] == 10 and out["config"]["ranks_joined"] == 2


def test_bench_under_torch_distributed_run_itself():
    """The driver's own N > 1 command line, with the no-GPU stub: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` (the launcher is PyTorch's; the ranks do not import it)."""
    pytest.importorskip("torch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTD_BENCH_STUB="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--total-columns", "10", "--columns", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_joined"] == 2 and out["control_plane"].endswith("torch imported: False")


def test_bench_time_limit_reports_every_ranks_phase():
    """A run that overruns RTD_BENCH_TIMEOUT is stopped by bench.py itself (default 1 500 s: under the driver's 1 800 s), with the
    phase every rank was in."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--columns", "3", "--total-columns", "0"], RTD_BENCH_STUB_HANG_RANK="1",
               RTD_BENCH_TIMEOUT="6")
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "rank 1: running; last phase: stub hang" in r.stderr and "rank 0: running; last phase: join the control plane" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_default_time_limit_is_below_the_drivers():
    sys.path[:0] = [ROOT]
    import bench
    assert bench.DEFAULT_TIMEOUT < 1800


def test_bench_refuses_world_size_mismatch():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
