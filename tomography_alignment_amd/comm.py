"""
Multi-GPU communicator: RCCL over xGMI through the C-ABI (tomo_comm_* / tomo_allreduce_*), one process
per GPU.  Replaces the mpi4py COMM_WORLD object the reference's recon/*_mpi.py take as first argument
(recon/sirt_mpi.py:12,38-39: Get_size / Get_rank; :68,103 Allreduce(SUM); :110 scalar allreduce).

Bootstrap: rank 0 creates the ncclUniqueId and serves it over a loopback TCP socket next to the launcher's
MASTER_PORT; the other ranks of the node connect and fetch it.  Every exchange carries the launch key (the
launcher's port, run id and pid, or the nonce bench.py's own launcher exports), so a rank can only ever
receive the id of ITS launch: a listener left behind by a crashed earlier launch answers "not yours" and the
rank moves on to the next candidate port.  (An earlier version passed the id through a file in /tmp, which a
crashed launch with the same key could leave behind for the next one to read.)  Launch contract:
`python -m torch.distributed.run --nproc-per-node N ...` -- or `python bench.py --gpus N`, which spawns the
ranks itself -- exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; this module only reads those variables,
it does not import torch.
"""
import ctypes
import os
import socket
import struct
import time

import numpy as np

try:
    from . import _lib
except ImportError:      # package directory itself on sys.path
    import _lib

N_CANDIDATE_PORTS = 8


def rendezvous_key():
    """Identifies one launch: an explicit nonce (TOMO_RDV_KEY, set by bench.py's launcher) or the launcher's port and run id
    plus its pid (all ranks of a node share the parent)."""
    k = os.environ.get("TOMO_RDV_KEY")
    if k:
        return k
    return "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())


def candidate_ports(base=None):
    """Loopback ports the id server may sit on: TOMO_RDV_PORT or MASTER_PORT + 1, and the next few (first free wins)."""
    if base is None:
        base = int(os.environ.get("TOMO_RDV_PORT") or (int(os.environ.get("MASTER_PORT", "29500")) + 1))
    return [1024 + (int(base) + i - 1024) % (65536 - 1024) for i in range(N_CANDIDATE_PORTS)]


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed")
        buf += chunk
    return buf


def exchange_from_rank0(rank, size, make_payload, timeout=300.0, key=None, port=None):
    """Rank 0 calls make_payload() and serves the bytes on 127.0.0.1 until the size - 1 other ranks of THIS launch (same
    key) have fetched them; every other rank connects (retrying until `timeout`) and returns them.  Returns the payload on
    every rank."""
    key_b = (key or rendezvous_key()).encode()
    ports = candidate_ports(port)
    if rank == 0:
        payload = make_payload()
        if size <= 1:
            return payload
        srv = None
        for p in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                s.bind(("127.0.0.1", p))
                s.listen(64)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise _lib.TomoError("RCCL id rendezvous: none of the loopback ports %s is free" % ports)
        served, deadline = 0, time.time() + timeout
        try:
            while served < size - 1:
                srv.settimeout(max(0.05, deadline - time.time()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise _lib.TomoError("RCCL id rendezvous: %d of %d ranks fetched the id within %.0f s" % (served, size - 1, timeout))
                try:
                    conn.settimeout(5.0)
                    n = struct.unpack("<I", _recv_exact(conn, 4))[0]
                    theirs = _recv_exact(conn, n) if n < 4096 else b""
                    if theirs == key_b:
                        conn.sendall(b"OK" + struct.pack("<I", len(payload)) + payload)
                        _recv_exact(conn, 1)               # the rank confirms it holds the bytes
                        served += 1
                    else:
                        conn.sendall(b"NO")                # somebody else's launch
                except (OSError, ConnectionError, struct.error):
                    pass
                finally:
                    conn.close()
        finally:
            srv.close()
        return payload
    deadline = time.time() + timeout
    while True:
        for p in ports:
            try:
                c = socket.create_connection(("127.0.0.1", p), timeout=1.0)
            except OSError:
                continue
            try:
                c.settimeout(10.0)
                c.sendall(struct.pack("<I", len(key_b)) + key_b)
                if _recv_exact(c, 2) == b"OK":
                    n = struct.unpack("<I", _recv_exact(c, 4))[0]
                    payload = _recv_exact(c, n)
                    c.sendall(b"!")
                    return payload
            except (OSError, ConnectionError, struct.error):
                pass
            finally:
                c.close()
        if time.time() > deadline:
            raise _lib.TomoError("timed out waiting for rank 0's RCCL id on 127.0.0.1 ports %s (key %s)" % (ports, key_b.decode()))
        time.sleep(0.05)


class SingleComm(object):
    """World of one: every collective is the identity (the unsharded solvers use this)."""
    size = 1
    rank = 0

    def Get_size(self):
        return 1

    def Get_rank(self):
        return 0

    def allreduce_sum_(self, buf):
        return buf

    def allreduce_scalar(self, v):
        return v

    def allreduce_max(self, v):
        return v

    def allreduce_array(self, a):
        return a

    def barrier(self):
        pass


class RcclComm(object):

    device_scalars = True      # HipBackend.acc_fetch(allreduce=True) sums device accumulators over THIS communicator (tomo_acc_fetch)

    def __init__(self, ctx, rank, size, id_bytes):
        self.ctx = ctx
        self.rank = int(rank)
        self.size = int(size)
        buf = ctypes.create_string_buffer(bytes(id_bytes), _lib.COMM_ID_BYTES)
        ctx.check(ctx.lib.tomo_comm_init(ctx.handle, buf, self.size, self.rank))

    # mpi4py-flavoured accessors so reference-style call sites read the same
    def Get_size(self):
        return self.size

    def Get_rank(self):
        return self.rank

    @staticmethod
    def unique_id(lib=None):
        lib = lib or _lib.load()
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = lib.tomo_comm_get_unique_id(buf)
        if rc != 0:
            raise _lib.TomoError("tomo_comm_get_unique_id failed: %s" % (lib.tomo_last_error(None) or b"").decode())
        return buf.raw

    @classmethod
    def from_env(cls, ctx=None, timeout=300.0):
        rank = int(os.environ.get("RANK", "0"))
        size = int(os.environ.get("WORLD_SIZE", "1"))
        if ctx is None:
            ctx = _lib.Context(int(os.environ.get("LOCAL_RANK", str(rank))))
        if size == 1:
            c = SingleComm()
            c.ctx = ctx
            return c
        # one node (the launch contract is --nnodes=1): let RCCL bootstrap over loopback instead of probing NICs
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        uid = exchange_from_rank0(rank, size, lambda: cls.unique_id(ctx.lib), timeout=timeout)
        comm = cls(ctx, rank, size, uid)
        comm.barrier()                  # every rank has joined the communicator
        return comm

    def allreduce_sum_(self, buf):
        """In-place sum of a float32 DeviceArray across ranks (recon/sirt_mpi.py:103)."""
        self.ctx.check(self.ctx.lib.tomo_allreduce_sum_f32(self.ctx.handle, buf.ptr, buf.size))
        return buf

    def allreduce_sum_async(self, buf):
        """Start an in-place sum of `buf` on the communication stream (after everything queued on the compute stream);
        `join()` makes the compute stream wait for all of them."""
        self.ctx.check(self.ctx.lib.tomo_allreduce_sum_f32_async(self.ctx.handle, buf.ptr, buf.size))
        return buf

    def join(self):
        self.ctx.check(self.ctx.lib.tomo_comm_join(self.ctx.handle))

    def wait_next(self):
        """The compute stream waits for the oldest allreduce_sum_async it has not waited for yet (issue order)."""
        self.ctx.check(self.ctx.lib.tomo_comm_wait_next(self.ctx.handle))

    def reduce_scatter_sum_async(self, buf, n_per_rank):
        """Start an in-place reduce-scatter of buf[0 : size * n_per_rank] on the communication stream: afterwards (wait_next) this
        rank's piece buf[rank * n_per_rank : (rank + 1) * n_per_rank] holds the sum over the ranks; the other pieces are undefined."""
        assert buf.size >= self.size * n_per_rank
        self.ctx.check(self.ctx.lib.tomo_reduce_scatter_sum_f32_async(self.ctx.handle, buf.ptr, int(n_per_rank)))
        return buf

    def allgather_async(self, buf, n_per_rank):
        """Start an in-place all-gather of buf[0 : size * n_per_rank] (this rank contributes its piece); wait_next_gather()."""
        assert buf.size >= self.size * n_per_rank
        self.ctx.check(self.ctx.lib.tomo_allgather_f32_async(self.ctx.handle, buf.ptr, int(n_per_rank)))
        return buf

    def wait_next_gather(self):
        """The compute stream waits for the oldest allgather_async it has not waited for yet."""
        self.ctx.check(self.ctx.lib.tomo_comm_wait_next_gather(self.ctx.handle))

    def allreduce_scalar(self, v):
        a = np.array([v], np.float64)
        self.ctx.check(self.ctx.lib.tomo_allreduce_sum_f64_host(self.ctx.handle, _lib.dptr(a), 1))
        return float(a[0])

    def allreduce_max(self, v):
        a = np.array([v], np.float64)
        self.ctx.check(self.ctx.lib.tomo_allreduce_max_f64_host(self.ctx.handle, _lib.dptr(a), 1))
        return float(a[0])

    def allreduce_array(self, a):
        """Element-wise sum of a small float64 host array across ranks (in place)."""
        flat = np.ascontiguousarray(a, np.float64).ravel()
        self.ctx.check(self.ctx.lib.tomo_allreduce_sum_f64_host(self.ctx.handle, _lib.dptr(flat), flat.size))
        a[...] = flat.reshape(a.shape)
        return a

    def barrier(self):
        self.allreduce_scalar(0.0)

    def close(self):
        self.ctx.check(self.ctx.lib.tomo_comm_destroy(self.ctx.handle))
