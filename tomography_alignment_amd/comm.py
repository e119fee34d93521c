"""
Multi-GPU communicator: RCCL over xGMI through the C-ABI (tomo_comm_* / tomo_allreduce_*), one process
per GPU.  Replaces the mpi4py COMM_WORLD object the reference's recon/*_mpi.py take as first argument
(recon/sirt_mpi.py:12,38-39: Get_size / Get_rank; :68,103 Allreduce(SUM); :110 scalar allreduce).

Bootstrap: rank 0 creates the ncclUniqueId and publishes it through a file keyed by the launcher
(MASTER_PORT + the launcher's pid); the other ranks of the node poll for it.  Launch contract:
`python -m torch.distributed.run --nproc-per-node N ...` exports RANK / LOCAL_RANK / WORLD_SIZE; this
module only reads those variables -- it does not import torch.
"""
import ctypes
import os
import tempfile
import time

import numpy as np

try:
    from . import _lib
except ImportError:      # package directory itself on sys.path
    import _lib


def rendezvous_key():
    """Identifies one launch: the launcher's port and run id plus its pid (all ranks of a node share the parent)."""
    return "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())


def exchange_from_rank0(rank, size, make_payload, timeout=300.0, directory=None, key=None):
    """Rank 0 calls make_payload() and publishes the bytes through a file; every other rank polls for it and reads
    it.  The file is written atomically (tmp + rename) and left for the ranks to read; stale files of earlier launches
    cannot collide because the key carries the launcher's pid.  Returns the payload on every rank."""
    directory = directory or tempfile.gettempdir()
    path = os.path.join(directory, "tomo_rccl_%s.id" % (key or rendezvous_key()))
    if rank == 0:
        payload = make_payload()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)
        return payload
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise _lib.TomoError("timed out waiting for the RCCL id file %s" % path)
        time.sleep(0.02)
    with open(path, "rb") as f:
        return f.read()


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
        comm.barrier()                  # every rank has joined the communicator, hence has read the id
        if rank == 0:
            try:
                os.remove(os.path.join(tempfile.gettempdir(), "tomo_rccl_%s.id" % rendezvous_key()))
            except OSError:
                pass
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
