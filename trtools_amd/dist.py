"""Multi-GPU sharding of the locus axis (one process per GPU).

Loci are independent, so the call set is cut into contiguous locus shards, one
per rank, and every rank runs the unchanged single-GPU path on its shard.  The
only data that crosses ranks:

* dumpSTR's ``sample_info`` (per-sample numcalls / totaldp / per-filter counts)
  and ``loc_info`` counters are sums over ALL loci  -> all-reduce (sum);
* the per-locus result rows (statSTR table rows, locus filter bits) are needed by
  the rank that writes the output                          -> all-gather in rank order.

``RcclComm`` runs both on device buffers through libtrk (RCCL over xGMI);
``SocketGroup`` does the same on host arrays over plain TCP (rendezvous, barrier,
the small reductions of the sharded command lines).  No torch in the package: the
gloo tests bring their own communicator (tests/torch_comm.py)."""
import struct

import numpy as np


def locus_shard(n_loci, rank, world):
    """Contiguous, balanced shard [start, stop) of rank ``rank`` (output order = input order)."""
    base, rem = divmod(int(n_loci), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class SocketGroup:
    """Process group over plain TCP sockets -- rendezvous, barrier and small host-array collectives without torch.
    Rank 0 listens on (addr, port) and is the hub: a collective is a gather of every rank's payload to rank 0 and a
    broadcast of the result.  What crosses it is small (the 128-byte RCCL id, a few timings, dumpSTR's per-sample
    counters: < 1 MB); bulk data goes through RCCL (``RcclComm``).  Same surface as ``RcclComm`` (and the tests' gloo
    communicator) so the sharded command lines run on any of them."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0):
        import os
        import socket
        import struct
        import time
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(port) if port is not None else int(os.environ.get('MASTER_PORT', '29500')) + 17
        self._struct = struct
        self._peers = {}          # rank 0: rank -> socket
        self._hub = None          # other ranks: socket to rank 0
        if self.world <= 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            while len(self._peers) < self.world - 1:
                c, _a = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(None)
                r = struct.unpack('<q', self._recv_exact(c, 8))[0]
                if not (0 < r < self.world) or r in self._peers:
                    c.close()
                    raise OSError("rendezvous: unexpected rank %d" % r)
                self._peers[r] = c
            srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=10)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.1)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(None)
            c.sendall(struct.pack('<q', self.rank))
            self._hub = c

    @staticmethod
    def _recv_exact(c, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(min(n - len(buf), 1 << 20))
            if not chunk:
                raise OSError("peer closed the connection")
            buf += chunk
        return bytes(buf)

    def _send(self, c, payload):
        c.sendall(self._struct.pack('<q', len(payload)) + payload)

    def _recv(self, c):
        n = self._struct.unpack('<q', self._recv_exact(c, 8))[0]
        return self._recv_exact(c, n)

    def _gather(self, payload):
        """rank 0: list of every rank's payload in rank order; other ranks: None (payload sent)."""
        if self.world <= 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [self._recv(self._peers[r]) for r in range(1, self.world)]
        self._send(self._hub, payload)
        return None

    def _bcast(self, payload):
        if self.world <= 1:
            return payload
        if self.rank == 0:
            for r in range(1, self.world):
                self._send(self._peers[r], payload)
            return payload
        return self._recv(self._hub)

    def broadcast_bytes(self, payload):
        """``payload`` of rank 0 on every rank."""
        return self._bcast(payload if self.rank == 0 else b'')

    def barrier(self):
        self._gather(b'')
        self._bcast(b'')

    def allgather_bytes(self, arr):
        mine = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1).tobytes()
        parts = self._gather(mine)
        blob = self._bcast(pack_frames(parts) if parts is not None else b'')
        return [np.frombuffer(p, dtype=np.uint8) for p in unpack_frames(blob)]

    def _allreduce(self, arr, dtype, fold):
        a = np.ascontiguousarray(arr, dtype=dtype)
        parts = self._gather(a.tobytes())
        if parts is not None:
            acc = np.frombuffer(parts[0], dtype=dtype).copy()
            for p in parts[1:]:
                acc = fold(acc, np.frombuffer(p, dtype=dtype))
            parts = acc.tobytes()
        return np.frombuffer(self._bcast(parts or b''), dtype=dtype).reshape(a.shape).copy()

    def allreduce_sum_i64(self, arr):
        return self._allreduce(arr, np.int64, np.add)

    def allreduce_max_f64(self, arr):
        return self._allreduce(arr, np.float64, np.maximum)

    def close(self):
        for c in list(self._peers.values()) + ([self._hub] if self._hub else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._hub = {}, None


def pack_frames(parts):
    """A list of byte strings as one blob: count, the sizes (int64 each), then the raw bytes.  Plain framing --
    nothing a peer sends is ever unpickled."""
    parts = [bytes(p) for p in parts]
    return struct.pack('<q', len(parts)) + struct.pack('<%dq' % len(parts), *[len(p) for p in parts]) + b''.join(parts)


def unpack_frames(blob):
    blob = bytes(blob)
    if len(blob) < 8:
        raise ValueError("truncated frame list")
    n = struct.unpack_from('<q', blob, 0)[0]
    if n < 0 or 8 + 8 * n > len(blob):
        raise ValueError("bad frame count %d" % n)
    sizes = struct.unpack_from('<%dq' % n, blob, 8)
    at, out = 8 + 8 * n, []
    for sz in sizes:
        if sz < 0 or at + sz > len(blob):
            raise ValueError("bad frame size %d" % sz)
        out.append(blob[at:at + sz])
        at += sz
    if at != len(blob):
        raise ValueError("trailing bytes after the frames")
    return out


class RcclComm:
    """Collectives on DeviceArrays through libtrk (trk_allreduce_sum_i64 / trk_allgather)."""

    def __init__(self, engine, rank, world):
        self.eng, self.rank, self.world = engine, rank, world

    def allreduce_sum_i64(self, arr):
        d = self.eng.upload(np.ascontiguousarray(arr, dtype=np.int64))
        self.eng.allreduce_sum_i64(d)
        out = d.get()
        d.free()
        return out

    def allgather_bytes(self, arr):
        sizes = self.allreduce_sum_i64(np.eye(self.world, dtype=np.int64)[self.rank] * arr.size)
        m = int(sizes.max())
        pad = np.zeros(m, dtype=np.uint8)
        pad[:arr.size] = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
        send = self.eng.upload(pad)
        recv = self.eng.empty((self.world, m), np.uint8)
        self.eng.allgather(send, recv)
        out = recv.get()
        send.free()
        recv.free()
        return [out[r, :int(sizes[r])] for r in range(self.world)]


def reduce_sample_info(sample_info, comm):
    """Cohort-wide ``sample_info`` from per-shard ones (dumpSTR.py:1251-1259 semantics:
    integer counters add; totaldp adds and stays nan once any shard poisoned it).  totaldp is a float sum
    (ExpansionHunter's LC depth is a Float field): every rank's float64 partial sums are gathered and added on the
    host in rank order -- the order the single-process run meets the records in -- never cast to integers."""
    keys = list(sample_info.keys())
    ints = np.stack([np.asarray(sample_info[k], dtype=np.int64) for k in keys if k != 'totaldp'])
    td = np.asarray(sample_info['totaldp'], dtype=np.float64)
    poisoned = np.isnan(td)
    red = comm.allreduce_sum_i64(np.concatenate([ints, poisoned.astype(np.int64)[None, :]]))
    parts = comm.allgather_bytes(np.frombuffer(np.where(poisoned, 0.0, td).astype(np.float64).tobytes(), dtype=np.uint8))
    total = np.zeros(td.shape[0], dtype=np.float64)
    for p in parts:
        total = total + np.frombuffer(p.tobytes(), dtype=np.float64)
    out = type(sample_info)()
    it = iter(red[:len(keys) - 1])
    for k in keys:
        if k == 'totaldp':
            total[red[-1] > 0] = np.nan
            out[k] = total
        else:
            out[k] = next(it)
    return out


def reduce_loc_info(loc_info, comm):
    keys = list(loc_info.keys())
    red = comm.allreduce_sum_i64(np.array([int(loc_info[k]) for k in keys], dtype=np.int64))
    out = type(loc_info)()
    for k, v in zip(keys, red):
        out[k] = int(v)
    return out


def gather_rows(rows_bytes, comm):
    """Concatenate per-shard output (bytes) in rank order == locus order."""
    parts = comm.allgather_bytes(np.frombuffer(rows_bytes, dtype=np.uint8))
    return b''.join(p.tobytes() for p in parts)


# ---------------------------------------------------------------------------------------
# locus-sharded command-line runs (statSTR / dumpSTR under a one-process-per-GPU launcher)
# ---------------------------------------------------------------------------------------
_comm = None


def set_comm(comm):
    """Install the communicator the sharded CLIs use (tests: tests/torch_comm.TorchComm over gloo)."""
    global _comm
    old = _comm
    _comm = comm
    return old


def _exchange_id(uid, rank, world, addr, port):
    """Rank 0 hands the 128-byte RCCL id to every other rank over plain TCP (no torch needed)."""
    g = SocketGroup(rank, world, addr, port)
    try:
        return g.broadcast_bytes(uid if rank == 0 else b'')
    finally:
        g.close()


def get_comm():
    """(rank, world, comm).  world == 1 -> comm is None.  With WORLD_SIZE > 1 in the environment
    (torchrun / any launcher exporting RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) an RCCL
    communicator is created inside libtrk on this process's GPU."""
    import os
    global _comm
    if _comm is not None:
        return _comm.rank, _comm.world, _comm
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world <= 1:
        return 0, 1, None
    from . import runtime
    eng = runtime.get_compute().eng
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500')) + 17
    uid = _exchange_id(eng.comm_unique_id() if rank == 0 else None, rank, world, addr, port)
    eng.comm_init(rank, world, uid)
    _comm = RcclComm(eng, rank, world)
    return rank, world, _comm


def merge_parts(parts, comm):
    """``parts``: this rank's list of (batch_index, bytes).  Returns, on every rank, the
    concatenation of all ranks' parts in batch order (== record order of the input)."""
    mine = pack_frames([struct.pack('<q', int(i)) + bytes(p) for i, p in parts])
    blobs = comm.allgather_bytes(np.frombuffer(mine, dtype=np.uint8))
    allparts = []
    for b in blobs:
        for fr in unpack_frames(b.tobytes()):
            allparts.append((struct.unpack_from('<q', fr, 0)[0], fr[8:]))
    allparts.sort(key=lambda t: t[0])
    return b''.join(p for _, p in allparts)
