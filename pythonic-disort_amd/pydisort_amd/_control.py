"""Control plane of a one-node, one-process-per-GPU run: barrier, max / min / sum over the ranks, a broadcast (the 128-byte
RCCL unique id) and a gather of small records to rank 0 -- standard library only (no PyTorch, no MPI).

The DATA plane is RCCL inside librtd.so (include/rtd.h: rtd_comm_*); this module only carries a few bytes per step between
the rank processes of one node, as SURVEY section 8(e) asks ("ranks shard columns with no exchange during the solve").  The
reference has no multi-process form (its loops _solve_for_gen_and_part_sols.py:88-91, _solve_for_coeffs.py:110-111 are the
independent units that are sharded).

Topology: a star.  Rank 0 listens on a Unix-domain socket in the abstract namespace (no file, gone with the process) whose name
is derived from MASTER_PORT (+ the launcher's run id): every launcher that sets RANK / WORLD_SIZE / MASTER_PORT for its ranks
-- bench.py's own, torch.distributed.run -- gives them a rendezvous without this module ever binding the launcher's TCP port
(torch.distributed.run keeps its own store there).  Every collective carries a sequence number and its name: ranks that fall
out of step, leave, or never arrive end the run with a message that says who and where, never with a silent hang.
"""
import json
import os
import select
import socket
import struct
import time


class ControlError(RuntimeError):
    """A rank left, never joined, or fell out of step; the message names the rank(s) and the collective."""


def rendezvous_key(environ=None):
    """The name ranks of one launch share: MASTER_PORT (unique per launch on a node) + the elastic run id when there is one."""
    e = os.environ if environ is None else environ
    # a per-launch nonce on top: bench.py's launcher gives its ranks a run directory of their own (RTD_BENCH_RUN_DIR); without a
    # MASTER_PORT the launcher's pid stands in (the ranks of one launch share a parent), so two launches on a node never meet
    nonce = os.path.basename(e.get("RTD_BENCH_RUN_DIR", "")) or "none"
    port = e.get("MASTER_PORT") or f"ppid{os.getppid()}"
    return f"rtd-ctl-{e.get('MASTER_ADDR', '127.0.0.1')}-{port}-{e.get('TORCHELASTIC_RUN_ID', 'none')}-{nonce}"


def _address(key):
    d = os.environ.get("RTD_CTL_DIR")  # a directory for a file-system socket where abstract names are not available
    return os.path.join(d, key + ".sock") if d else "\0" + key


def _send(sock, obj):
    raw = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(raw)) + raw)


def _recv_exact(sock, n, deadline, who):
    buf = b""
    while len(buf) < n:
        left = deadline - time.monotonic()
        if left <= 0:
            raise TimeoutError(who)
        r, _, _ = select.select([sock], [], [], min(left, 1.0))
        if not r:
            continue
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError(who)
        buf += chunk
    return buf


def _recv(sock, deadline, who):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4, deadline, who))
    return json.loads(_recv_exact(sock, n, deadline, who).decode())


class ControlPlane:
    """rank / world as the launcher set them; `timeout` bounds every wait (joining and each collective)."""

    def __init__(self, rank, world, key=None, timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.timeout = float(os.environ.get("RTD_CTL_TIMEOUT", "600")) if timeout is None else float(timeout)
        self.seq = 0
        self.peers = {}     # rank 0: {rank: socket}
        self.sock = None    # other ranks: the socket to rank 0
        self._listener = None
        if self.world == 1:
            return
        addr = _address(key or rendezvous_key())
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            ls = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            if not addr.startswith("\0") and os.path.exists(addr):
                os.unlink(addr)
            ls.bind(addr)
            ls.listen(self.world)
            self._listener = ls
            try:
                while len(self.peers) < self.world - 1:
                    left = deadline - time.monotonic()
                    missing = sorted(set(range(1, self.world)) - set(self.peers))
                    if left <= 0:
                        raise ControlError(f"control plane: ranks {missing} of {self.world} never joined within {self.timeout:.0f} s")
                    r, _, _ = select.select([ls], [], [], min(left, 1.0))
                    if not r:
                        continue
                    c, _ = ls.accept()
                    # a connection that says nothing for 5 s, hangs up, or says something that is not a rank of this run (any local
                    # process can connect to the socket) is dropped; the join goes on and ends on its own deadline, naming who is missing
                    try:
                        hello = _recv(c, min(deadline, time.monotonic() + 5.0), "hello")
                        ok = isinstance(hello, dict) and hello.get("world") == self.world and isinstance(hello.get("rank"), int) \
                            and 0 < hello["rank"] < self.world and hello["rank"] not in self.peers
                    except (TimeoutError, ConnectionError, ValueError, struct.error, UnicodeDecodeError):
                        ok = False
                    if not ok:
                        c.close()
                        continue
                    self.peers[hello["rank"]] = c
                for c in self.peers.values():
                    _send(c, {"joined": self.world})
            except BaseException:
                for c in self.peers.values():
                    c.close()
                self.peers = {}
                ls.close()
                self._listener = None
                raise
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(addr)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise ControlError(f"control plane: rank {self.rank} found no rank 0 to join within {self.timeout:.0f} s")
                    time.sleep(0.02)
            _send(s, {"rank": self.rank, "world": self.world})
            try:
                ack = _recv(s, deadline, "join")
            except (TimeoutError, ConnectionError):
                raise ControlError(f"control plane: rank {self.rank}: rank 0 closed or stalled while the ranks were joining")
            if ack.get("joined") != self.world:
                raise ControlError(f"control plane: rank 0 runs {ack.get('joined')} ranks, this rank was started as one of {self.world}")
            self.sock = s

    # -- the one primitive: everybody hands a value to rank 0, rank 0 answers everybody -------------------------------
    def _exchange(self, name, value, combine):
        self.seq += 1
        if self.world == 1:
            out = combine([value])
            return out.value if isinstance(out, _RootOnly) else out
        deadline = time.monotonic() + self.timeout
        tag = {"seq": self.seq, "op": name}
        if self.rank == 0:
            vals = {0: value}
            waiting = dict(self.peers)
            while waiting:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise ControlError(f"control plane: ranks {sorted(waiting)} did not reach {name} #{self.seq} within {self.timeout:.0f} s")
                ready, _, _ = select.select(list(waiting.values()), [], [], min(left, 1.0))
                for c in ready:
                    r = next(k for k, v in waiting.items() if v is c)
                    try:
                        msg = _recv(c, deadline, f"rank {r}")
                    except ConnectionError:
                        raise ControlError(f"control plane: rank {r} left before {name} #{self.seq}")
                    except TimeoutError:
                        raise ControlError(f"control plane: rank {r} stalled inside {name} #{self.seq}")
                    if msg.get("seq") != self.seq or msg.get("op") != name:
                        raise ControlError(f"control plane: rank {r} is at {msg.get('op')} #{msg.get('seq')}, rank 0 at {name} #{self.seq}")
                    vals[r] = msg.get("v")
                    del waiting[r]
            out = combine([vals[r] for r in range(self.world)])
            for r, c in self.peers.items():
                try:
                    _send(c, dict(tag, v=out if not isinstance(out, _RootOnly) else None))
                except OSError:
                    raise ControlError(f"control plane: rank {r} left during {name} #{self.seq}")
            return out.value if isinstance(out, _RootOnly) else out
        try:
            _send(self.sock, dict(tag, v=value))
            msg = _recv(self.sock, deadline, "rank 0")
        except (ConnectionError, OSError):
            raise ControlError(f"control plane: rank 0 left before answering {name} #{self.seq} of rank {self.rank}")
        except TimeoutError:
            raise ControlError(f"control plane: rank {self.rank} had no answer to {name} #{self.seq} within {self.timeout:.0f} s")
        if msg.get("seq") != self.seq or msg.get("op") != name:
            raise ControlError(f"control plane: rank 0 answered {msg.get('op')} #{msg.get('seq')} to {name} #{self.seq} of rank {self.rank}")
        return msg.get("v")

    def barrier(self):
        self._exchange("barrier", None, lambda v: None)

    def allreduce(self, value, op="max"):
        """max / min / sum of one number over the ranks, on every rank."""
        fn = {"max": max, "min": min, "sum": sum}[op]
        return self._exchange("allreduce_" + op, value, fn)

    def all_ok(self, ok):
        """True iff every rank says ok."""
        return self.allreduce(1 if ok else 0, "min") == 1

    def broadcast_bytes(self, data, src=0):
        """`data` (bytes or None) of rank `src` on every rank; None stays None (the sender had nothing to send)."""
        mine = data.hex() if (self.rank == src and data is not None) else None
        got = self._exchange("broadcast", mine, lambda v: v[src])
        return None if got is None else bytes.fromhex(got)

    def gather(self, record):
        """JSON-able `record` of every rank, as a list in rank order on rank 0 (None elsewhere)."""
        return self._exchange("gather", record, lambda v: _RootOnly(list(v)))

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock else []) + ([self._listener] if self._listener else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock, self._listener = {}, None, None


class _RootOnly:
    """Result of a collective that only rank 0 keeps (the others are answered with None)."""

    def __init__(self, value):
        self.value = value
