"""CPU stand-ins used ONLY by the tests: an oracle-backed object with the method set of
tomography_alignment_amd.backend.HipBackend (so the solvers' control flow can run without a GPU) and a
torch.distributed (gloo) communicator with the method set of tomography_alignment_amd.comm.RcclComm.
Nothing in the product imports this module."""
import numpy as np

from oracle import oracle as orc


class Buf(object):
    """Host 'device buffer': same tiny surface as _lib.DeviceArray."""

    def __init__(self, a, copy=True):
        self.a = np.ascontiguousarray(a, np.float32).ravel()
        if copy:
            self.a = self.a.copy()
        self.size = self.a.size

    def view(self, offset, n):
        return Buf(self.a[offset:offset + n], copy=False)

    def zero_(self):
        self.a[:] = 0
        return self

    def download(self):
        return self.a.copy()

    def upload(self, host):
        self.a[:] = np.asarray(host, np.float32).ravel()
        return self

    def copy_from(self, other):
        self.a[:] = other.a
        return self


class OracleBackend(object):
    name = "oracle(test)"

    TILE_W = 16          # x width of a tile column of the product's tile kernels (tomo_adjoint_xslab_info)

    def __init__(self, geometry, declines_tiles=False):
        self.declines_tiles = declines_tiles       # stand-in for an angle block holding a pose the tile kernels decline
        self.geometry = geometry
        self.n_vox = int(np.prod(geometry.vox_shape))
        self.n_det = int(np.prod(geometry.det_shape))
        self.og = orc.Geo(geometry.n_proj, np.asarray(geometry.vox_shape), np.ones(3), np.asarray(geometry.det_shape), np.ones(2),
                          step_size=geometry.step_size)
        self.calls = {"forward": 0, "adjoint": 0, "cost_grad": 0, "proj_grad": 0}
        self.n_uploaded = 0        # float32 values that crossed "PCIe" (tests of what a rank uploads)

    def upload(self, host):
        self.n_uploaded += int(np.size(host))
        return Buf(host)

    def download(self, buf):
        return buf.download()

    def zeros(self, n):
        return Buf(np.zeros(int(n), np.float32))

    empty = zeros

    def copy(self, dst, src):
        dst.copy_from(src)

    def is_buffer(self, x):
        return isinstance(x, Buf)

    def _kw(self, poses):
        self.og.cor_shift = np.stack([poses[:, 6], np.zeros(len(poses)), np.zeros(len(poses))], axis=1)
        return dict(phi=poses[:, 0], alpha=poses[:, 1], beta=poses[:, 2], xyz_shift=poses[:, 3:6])

    def forward(self, poses, vol, out):
        self.calls["forward"] += 1
        out.a[:] = orc.forward(self.og, vol.a, **self._kw(poses)).astype(np.float32).ravel()
        return out

    def adjoint(self, poses, proj, out, accumulate=False):
        self.calls["adjoint"] += 1
        r = orc.adjoint(self.og, proj.a, **self._kw(poses)).astype(np.float32)
        out.a[:] = out.a + r if accumulate else r
        return out

    # ---- x-slab forms (tomo_adjoint_xslab / tomo_forward_xslab): tile columns of width 16 on a grid that starts at x = -1.
    # Emulated by voxel ranges: columns [xt0, xt1) finalise -- and, here, read -- the voxels x in [16 xt0 - 1, 16 xt1 - 1) (the last
    # column up to nx).  The slabs of a partition of the columns then sum to the whole operator, like the product's.
    def xslab_info(self):
        nx = int(self.geometry.vox_shape[0])
        return (nx + 1 + self.TILE_W - 1) // self.TILE_W, self.TILE_W

    def _xrange(self, xt0, xt1):
        nx = int(self.geometry.vox_shape[0])
        n_xt, w = self.xslab_info()
        lo = 0 if xt0 <= 0 else min(nx, max(0, w * xt0 - 1))
        hi = nx if xt1 >= n_xt else min(nx, max(0, w * xt1 - 1))
        plane = self.n_vox // nx
        return lo * plane, max(lo, hi) * plane

    def _check_tiles(self):
        if self.declines_tiles:
            raise RuntimeError("stand-in: these poses do not take the tile kernels")

    def tiles_take(self, poses, proj, vol):
        return not self.declines_tiles

    def adjoint_xslab(self, poses, proj, out, xt0, xt1, same_sinogram=False):
        self._check_tiles()
        if xt1 <= xt0:
            return
        self.calls["adjoint"] += 1
        r = orc.adjoint(self.og, proj.a, **self._kw(poses)).astype(np.float32)
        a, b = self._xrange(xt0, xt1)
        out.a[a:b] += r[a:b]

    def forward_xslab(self, poses, vol, out, xt0, xt1):
        self._check_tiles()
        if xt1 <= xt0:
            return
        self.calls["forward"] += 1
        a, b = self._xrange(xt0, xt1)
        part = np.zeros_like(vol.a)
        part[a:b] = vol.a[a:b]
        out.a += orc.forward(self.og, part, **self._kw(poses)).astype(np.float32).ravel()

    def update_acc(self, rec, bp, v, positivity=False, gt=None, first=True):
        if first:
            self._acc = 0.0
        e = self.update(rec, bp, v, positivity, gt)
        if e is not None:
            self._acc += e

    def update_acc_fetch(self):
        return self._acc

    def triplets(self, pose):
        p = pose[0]
        return orc.forward_sparse(self.og, p[1], p[2], p[0], p[3:6], np.array([p[6], 0., 0.]))

    def proj_grad(self, pose, vol, proj_out, grad_out, row_order=0):
        self.calls["proj_grad"] += 1
        p, g = orc.projection_gradient(self.og, vol.a, pose[0, 1], pose[0, 2], pose[0, 0], pose[0, 3:6], np.array([pose[0, 6], 0, 0]))
        if row_order == 1:
            g = g[[0, 1, 2, 4, 5, 3]]
        proj_out.a[:] = p
        grad_out.a[:] = g.ravel()

    def cost_grad(self, poses, vol, b, resid=None, rows=None):
        n = poses.shape[0]
        cost, g6 = np.zeros(n), np.zeros((n, 6))
        table = b.a.reshape(-1, self.n_det)
        for i in range(n):
            self.calls["cost_grad"] += 1
            p, g = orc.projection_gradient(self.og, vol.a, poses[i, 1], poses[i, 2], poses[i, 0], poses[i, 3:6], np.array([poses[i, 6], 0, 0]))
            res = table[i if rows is None else int(rows[i])].astype(np.float64) - p
            cost[i] = 0.5 * np.dot(res, res)
            g6[i] = np.dot(-g.astype(np.float64), res)
        return cost, g6

    def fill(self, buf, value):
        buf.a[:] = value

    def recip_guard(self, buf, thresh=None):
        bad = (buf.a == 0.) if thresh is None else (buf.a < thresh)
        with np.errstate(divide="ignore"):
            r = 1. / buf.a
        r[bad] = 0.
        buf.a[:] = r

    def residual_scale(self, b, ax, w, out):
        r = b.a - ax.a
        out.a[:] = r if w is None else w.a * r
        return float(np.dot(r.astype(np.float64), r.astype(np.float64)))

    def update(self, rec, bp, v, positivity=False, gt=None):
        rec.a += bp.a if v is None else bp.a * v.a
        if positivity:
            rec.a[rec.a < 0.] = 0.
        if gt is None:
            return None
        e = (gt.a - rec.a).astype(np.float64)
        return float(np.dot(e, e))

    def axpy(self, y, x, a):
        y.a += np.float32(a) * x.a

    def xpay(self, y, x, a):
        y.a[:] = x.a + np.float32(a) * y.a

    def sub(self, out, a, b):
        out.a[:] = a.a - b.a

    def mul(self, y, x):
        y.a *= x.a

    def dot(self, a, b):
        return float(np.dot(a.a.astype(np.float64), b.a.astype(np.float64)))

    def diff_sumsq(self, a, b):
        e = (a.a - b.a).astype(np.float64)
        return float(np.dot(e, e))

    # device accumulators (tomo_acc_*): local sums; summing over ranks is the communicator's job here
    def acc_zero(self, slot0, n=1):
        if not hasattr(self, "_accs"):
            self._accs = np.zeros(16)
        self._accs[slot0:slot0 + n] = 0.0

    def dot_acc(self, a, b, slot, diff=False):
        if not hasattr(self, "_accs"):
            self._accs = np.zeros(16)
        self._accs[slot] += self.diff_sumsq(a, b) if diff else self.dot(a, b)

    def acc_fetch(self, slot0, n=1, allreduce=False):
        assert not allreduce, "the stand-in backend has no communicator of its own"
        return self._accs[slot0:slot0 + n].copy()

    def sync(self):
        pass


class GlooComm(object):
    """torch.distributed (gloo) with the RcclComm method set, on the host buffers above."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.ctx = None
        self.n_vol_allreduce = 0
        self.n_slab_allreduce = 0
        self.n_wait = 0
        self.slab_sizes = []

    def Get_size(self):
        return self.size

    def Get_rank(self):
        return self.rank

    def allreduce_sum_(self, buf):
        import torch
        t = torch.from_numpy(buf.a)
        self.dist.all_reduce(t)
        self.n_vol_allreduce += 1
        return buf

    # the asynchronous forms: gloo on host buffers is synchronous, so "start" does the all-reduce and the waits are no-ops; what
    # the tests see is the SEQUENCE of collectives (gloo fails on mismatched sizes / hangs on mismatched counts) and the counts
    def allreduce_sum_async(self, buf):
        import torch
        t = torch.from_numpy(buf.a)
        if t.numel():
            self.dist.all_reduce(t)
        self.n_slab_allreduce += 1
        self.slab_sizes.append(int(buf.size))
        return buf

    def wait_next(self):
        self.n_wait += 1

    # reduce-scatter / all-gather (round 4).  The reduce-scatter POISONS the pieces it does not own (RCCL leaves them undefined): a
    # solver that read another rank's piece before the all-gather would turn its result into NaN.
    def reduce_scatter_sum_async(self, buf, n_per_rank):
        import torch
        n = int(n_per_rank)
        seg = buf.a[:n * self.size]
        t = torch.from_numpy(seg)
        if t.numel():
            self.dist.all_reduce(t)
        mine = seg[self.rank * n:(self.rank + 1) * n].copy()
        if self.size > 1:
            seg[:] = np.nan
        seg[self.rank * n:(self.rank + 1) * n] = mine
        self.n_reduce_scatter = getattr(self, "n_reduce_scatter", 0) + 1
        self.slab_sizes.append(n * self.size)
        return buf

    def allgather_async(self, buf, n_per_rank):
        import torch
        n = int(n_per_rank)
        seg = buf.a[:n * self.size]
        if n:
            parts = [torch.empty(n, dtype=torch.float32) for _ in range(self.size)]
            self.dist.all_gather(parts, torch.from_numpy(seg[self.rank * n:(self.rank + 1) * n].copy()))
            for q, part in enumerate(parts):
                seg[q * n:(q + 1) * n] = part.numpy()
        self.n_allgather = getattr(self, "n_allgather", 0) + 1
        return buf

    def wait_next_gather(self):
        self.n_wait_gather = getattr(self, "n_wait_gather", 0) + 1

    def join(self):
        pass

    def allreduce_scalar(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t)
        return float(t[0])

    def allreduce_max(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def allreduce_array(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, np.float64))
        self.dist.all_reduce(t)
        a[...] = t.numpy()
        return a

    def barrier(self):
        self.dist.barrier()
