"""Control plane for one-process-per-GPU runs (bench.py, multi-GPU training) -- standard library only.

The data path is RCCL inside libyue_hip.so (yue_comm_init / yue_bpr_epoch).  This module only ships the
128-byte RCCL id, synchronises and reduces a few scalars between the ranks of ONE node.  It reads what the
launcher (`python -m torch.distributed.run`, or any launcher that exports the same variables) puts into the
environment -- RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT -- and speaks a tiny length-prefixed
protocol over TCP: a star with rank 0 in the middle.  No torch, no gloo: a process that uses the library never
loads a second HIP / RCCL runtime.

Rendezvous: the launcher's own store already listens on MASTER_PORT, so rank 0 binds the first free port of
MASTER_PORT+1 .. MASTER_PORT+64 and every other rank walks the same ports until a listener answers the
handshake (a token derived from the job's environment) -- a foreign service on one of the ports is skipped.
"""
import hashlib
import math
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b'YUE1'
_PORT_SPAN = 64
_TIMEOUT_S = 300.0


def env_rank():
    """(rank, world, local_rank) from the launcher's environment; (0, 1, 0) when run directly."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0')))


def _job_token(world):
    key = '|'.join([os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ.get('MASTER_PORT', '29500'), str(world),
                    os.environ.get('TORCHELASTIC_RUN_ID', ''), os.environ.get('YUE_JOB_ID', '')])
    return hashlib.sha256(key.encode()).digest()[:16]


def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError('control plane: peer closed the connection')
        buf.extend(chunk)
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class ControlPlane(object):
    """rank / world / local_rank + broadcast_bytes, barrier, reduce_max, allreduce_sum over a TCP star."""

    def __init__(self):
        self.rank, self.world, self.local_rank = env_rank()
        self._peers = []          # rank 0: sockets of ranks 1..world-1 (index rank-1)
        self._up = None           # other ranks: socket to rank 0
        if self.world > 1:
            self._connect()

    # -- rendezvous ---------------------------------------------------------------------------
    def _connect(self):
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        port0 = int(os.environ.get('MASTER_PORT', '29500')) + 1
        token = _job_token(self.world)
        deadline = time.time() + _TIMEOUT_S
        if self.rank == 0:
            srv = None
            for port in range(port0, port0 + _PORT_SPAN):
                try:
                    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    s.bind((addr if addr not in ('localhost',) else '127.0.0.1', port))
                    s.listen(self.world + 8)
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise RuntimeError('control plane: no free port in %d..%d' % (port0, port0 + _PORT_SPAN - 1))
            srv.settimeout(1.0)
            peers = {}
            while len(peers) < self.world - 1:
                if time.time() > deadline:
                    raise RuntimeError('control plane: ranks %s never connected' % sorted(set(range(1, self.world)) - set(peers)))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    conn.settimeout(10.0)
                    hello = _recv_exact(conn, 4 + 16 + 4)
                    if hello[:4] != _MAGIC or hello[4:20] != token:
                        conn.close()
                        continue
                    (r,) = struct.unpack('<i', hello[20:24])
                    if not (0 < r < self.world) or r in peers:
                        conn.close()
                        continue
                    conn.sendall(_MAGIC + token)
                    conn.settimeout(_TIMEOUT_S)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    peers[r] = conn
                except (OSError, ConnectionError, struct.error):
                    conn.close()
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while True:
                for port in range(port0, port0 + _PORT_SPAN):
                    try:
                        s = socket.create_connection((addr, port), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        s.settimeout(5.0)
                        s.sendall(_MAGIC + token + struct.pack('<i', self.rank))
                        if _recv_exact(s, 20) == _MAGIC + token:
                            s.settimeout(_TIMEOUT_S)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._up = s
                            return
                    except (OSError, ConnectionError):
                        pass
                    s.close()
                if time.time() > deadline:
                    raise RuntimeError('control plane: rank %d found no rank 0 on ports %d..%d of %s' % (self.rank, port0, port0 + _PORT_SPAN - 1, addr))
                time.sleep(0.05)

    # -- primitives ---------------------------------------------------------------------------
    def _gather(self, payload):
        """rank 0 gets [payload of rank 0, 1, ...]; the others get None."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [_recv(s) for s in self._peers]
        _send(self._up, payload)
        return None

    def _scatter_same(self, payload):
        """every rank gets rank 0's payload."""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for s in self._peers:
                _send(s, payload)
            return payload
        return _recv(self._up)

    def broadcast_bytes(self, payload, src=0):
        """payload is given on `src` (None elsewhere); every rank gets the bytes."""
        parts = self._gather(payload if self.rank == src else b'')
        return self._scatter_same(parts[src] if self.rank == 0 else None)

    def barrier(self):
        self._gather(b'')
        self._scatter_same(b'')

    def reduce_max(self, x):
        parts = self._gather(struct.pack('<d', float(x)))
        out = self._scatter_same(struct.pack('<d', max(struct.unpack('<d', p)[0] for p in parts)) if self.rank == 0 else None)
        return struct.unpack('<d', out)[0]

    def gather_floats(self, x):
        """every rank's value, in rank order, on every rank (bench.py: per-rank step times beside their maximum)"""
        parts = self._gather(struct.pack('<d', float(x)))
        out = self._scatter_same(b''.join(parts) if self.rank == 0 else None)
        return list(struct.unpack('<%dd' % (len(out) // 8), out))

    def allreduce_sum(self, array):
        """In-place float32/float64 numpy sum over ranks, added in rank order (CPU; the executable spec in tests)."""
        if self.world == 1:
            return array
        parts = self._gather(np.ascontiguousarray(array).tobytes())
        total = None
        if self.rank == 0:
            acc = np.frombuffer(parts[0], dtype=array.dtype).copy()
            for p in parts[1:]:
                acc += np.frombuffer(p, dtype=array.dtype)
            total = acc.tobytes()
        array[...] = np.frombuffer(self._scatter_same(total), dtype=array.dtype).reshape(array.shape)
        return array

    def close(self):
        for s in self._peers:
            s.close()
        self._peers = []
        if self._up is not None:
            self._up.close()
            self._up = None


def user_block_width(round_events, events_total, m, world):
    """Users per round on a communicator: the same number on every rank, derived from job-wide
    counts only (mirror of the rule in csrc/yue_hip.hip: yue_bpr_epoch)."""
    per_user = float(events_total) / world / m
    return max(1, int(math.floor(round_events / max(per_user, 1e-9) + 0.5)))


def epoch_round_ptr(ev_ptr, round_events, events_total=None, world=1):
    """Round boundaries (event offsets) of yue_bpr_epoch for this rank's events: blocks of whole users,
    user_block_width users each (the last block takes the rest)."""
    m = len(ev_ptr) - 1
    total = int(ev_ptr[-1]) if events_total is None else events_total
    ub = user_block_width(round_events, total, m, world)
    return [int(ev_ptr[u]) for u in range(0, m, ub)] + [int(ev_ptr[-1])]


def epoch_block_plan(m, k, round_events, events_total, world):
    """The host-side schedule of yue_bpr_epoch (csrc/yue_hip.hip): user-block width, blocks per apply / all-reduce
    group (at least ~8 MB of user-factor differences per collective), and the groups as (first user, one past the
    last user, float32 elements all-reduced).  The same on every rank by construction."""
    ub = user_block_width(round_events, events_total, m, world)
    group = max(1, (8 << 20) // max(1, ub * k * 4))
    n_blocks = (m + ub - 1) // ub
    groups = []
    for b0 in range(0, n_blocks, group):
        u0, u1 = b0 * ub, min(m, (b0 + group) * ub)
        groups.append((u0, u1, (u1 - u0) * k))
    return {'user_block': ub, 'blocks_per_group': group, 'n_blocks': n_blocks, 'groups': groups}


class stdout_to_stderr(object):
    """RCCL prints a version banner on STDOUT when a communicator is created; a program whose stdout is a protocol
    (bench.py: one JSON line) wraps the communicator set-up in this: file descriptor 1 points at stderr meanwhile."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._keep = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self._keep, 1)
        os.close(self._keep)
        return False


def attach_device(dev, cp):
    """Join `dev` (a yue_amd._shim.Device) to the job's RCCL communicator."""
    if cp.world > 1:
        from ._shim import comm_unique_id
        with stdout_to_stderr():
            ident = cp.broadcast_bytes(comm_unique_id() if cp.rank == 0 else None)
            dev.comm_init(ident, cp.rank, cp.world)
