"""N > 1 paths of bench.py on CPU (gloo, world_size 2).

* the control plane two ranks use: disjoint column shards, the max-over-ranks time, the all-ranks status flag and the
  broadcast that carries the RCCL unique id;
* the launcher: `bench.py --gpus 2` itself starts two fresh rank processes, both join the rendezvous and contribute
  to the result line (a no-GPU stub stands in for the device work); a rank that dies, or a WORLD_SIZE that does not
  match --gpus, ends the run with a non-zero exit code.
(The RCCL all-gather itself is covered on the GPU box with a one-rank communicator.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
    import bench
    from pydisort_amd import synthetic
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    first, C = bench.shard_columns(rank, world, 3)
    sfirst, sC = bench.shard_columns(rank, world, 3, total_columns=11)
    cfg = synthetic.cfg4_columns(C, first=first, L=4, NQuad=8)
    uid = [b"x" * 128 if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    elapsed = bench.reduce_max_seconds(dist, 1.0 + rank)
    all_ok = bench.all_ranks_ok(dist, True)
    one_bad = bench.all_ranks_ok(dist, rank != 1)
    dist.barrier()
    q.put((rank, first, C, cfg["tau_arr"].sum(), len(uid[0]), elapsed, all_ok, one_bad, sfirst, sC))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, c0, s0, n0, e0, a0, b0, sf0, sc0), (r1, f1, c1, s1, n1, e1, a1, b1, sf1, sc1) = out
    assert (f0, c0, f1, c1) == (0, 3, 3, 3)           # disjoint, contiguous shards
    assert (sf0, sc0, sf1, sc1) == (0, 5, 5, 5)       # strong scaling: equal shares of the total
    assert n0 == n1 == 128                            # unique id reached every rank
    assert e0 == e1 == 2.0                            # max over ranks
    assert a0 and a1 and not b0 and not b1            # one failing rank is seen by every rank
    sys.path[:0] = [os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import synthetic
    whole = synthetic.cfg4_columns(6, L=4, NQuad=8)["tau_arr"]
    assert np.isclose(s0, whole[:3].sum()) and np.isclose(s1, whole[3:].sum())  # union == global batch


def _bench(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTD_BENCH_STUB="1", RTD_BENCH_TIMEOUT="240", **env_over)
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


def test_bench_refuses_world_size_mismatch():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
