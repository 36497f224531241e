"""N > 1 control path of bench.py on CPU: two gloo ranks shard the synthetic columns without overlap,
agree on the max-over-ranks time, and exchange the RCCL unique-id placeholder through the same
broadcast the real run uses.  (The RCCL all-gather itself is covered on the GPU box with one rank.)"""
import os
import socket
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
    cfg = synthetic.cfg4_columns(C, first=first, L=4, NQuad=8)
    uid = [b"x" * 128 if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    elapsed = bench.reduce_max_seconds(dist, 1.0 + rank)
    dist.barrier()
    q.put((rank, first, C, cfg["tau_arr"].sum(), len(uid[0]), elapsed))
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
    (r0, f0, c0, s0, n0, e0), (r1, f1, c1, s1, n1, e1) = out
    assert (f0, c0, f1, c1) == (0, 3, 3, 3)           # disjoint, contiguous shards
    assert n0 == n1 == 128                            # unique id reached every rank
    assert e0 == e1 == 2.0                            # max over ranks
    sys.path[:0] = [os.path.join(ROOT, "pythonic-disort_amd")]
    from pydisort_amd import synthetic
    whole = synthetic.cfg4_columns(6, L=4, NQuad=8)["tau_arr"]
    assert np.isclose(s0, whole[:3].sum()) and np.isclose(s1, whole[3:].sum())  # union == global batch
