"""
ctypes binding of libtomo_hip.so (include/tomo.h) -- the only native surface of this package.

There is NO CPU fallback: if the HIP library is missing or no MI355X is usable, every entry point
raises.  (The reference binds its native code the same way: `from src import ray_wt_grad`,
utilities/ray_voxel_utilities.py:3.)
"""
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Multi-process GPU work on this platform needs dmabuf IPC (RCCL's peer buffers); the variable is read when the HIP runtime
# initialises, i.e. at the first call into the library, so it is set -- if the launcher has not -- before the library loads.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
LIB_PATH = os.environ.get("TOMO_HIP_LIB") or os.path.join(_HERE, "libtomo_hip.so")   # override: development builds only
POSE_STRIDE = 7
GEOM_WIDE_ROWS = 1        # TOMO_GEOM_WIDE_ROWS of include/tomo.h
COMM_ID_BYTES = 128

_c_i64 = ctypes.c_int64
_c_vp = ctypes.c_void_p
_c_dp = ctypes.POINTER(ctypes.c_double)


class TomoGeom(ctypes.Structure):
    """struct tomo_geom of include/tomo.h."""
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32),
                ("ndx", ctypes.c_int32), ("ndz", ctypes.c_int32),
                ("vox_origin", ctypes.c_double * 3), ("vox_pitch", ctypes.c_double * 3),
                ("det_x0", ctypes.c_double), ("det_z0", ctypes.c_double),
                ("det_dx", ctypes.c_double), ("det_dz", ctypes.c_double),
                ("src_y", ctypes.c_double), ("det_y", ctypes.c_double), ("step", ctypes.c_double)]


# every symbol include/tomo.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "tomo_abi_version": (ctypes.c_int, []),
    "tomo_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "tomo_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_c_vp)]),
    "tomo_ctx_destroy": (ctypes.c_int, [_c_vp]),
    "tomo_last_error": (ctypes.c_char_p, [_c_vp]),
    "tomo_device_name": (ctypes.c_int, [_c_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "tomo_malloc": (ctypes.c_int, [_c_vp, ctypes.c_size_t, ctypes.POINTER(_c_vp)]),
    "tomo_free": (ctypes.c_int, [_c_vp, _c_vp]),
    "tomo_memcpy_h2d": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_size_t]),
    "tomo_memcpy_d2h": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_size_t]),
    "tomo_memcpy_d2d": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_size_t]),
    "tomo_memset0": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_size_t]),
    "tomo_sync": (ctypes.c_int, [_c_vp]),
    "tomo_ctx_make_current": (ctypes.c_int, [_c_vp]),
    "tomo_set_option": (ctypes.c_int, [_c_vp, ctypes.c_char_p, ctypes.c_int]),
    "tomo_ctx_set_cu_mask": (ctypes.c_int, [_c_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]),
    "tomo_check_geometry": (ctypes.c_int, [ctypes.POINTER(TomoGeom), ctypes.POINTER(ctypes.c_int)]),
    "tomo_set_geometry": (ctypes.c_int, [_c_vp, ctypes.POINTER(TomoGeom)]),
    "tomo_forward": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp]),
    "tomo_adjoint": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp, ctypes.c_int]),
    "tomo_adjoint_xslab_info": (ctypes.c_int, [_c_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "tomo_adjoint_xslab": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int]),
    "tomo_forward_xslab": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int]),
    "tomo_backproject_voxel": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp]),
    "tomo_proj_grad": (ctypes.c_int, [_c_vp, _c_dp, _c_vp, _c_vp, _c_vp, ctypes.c_int]),
    "tomo_cost_grad": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp, _c_dp, _c_dp, _c_vp]),
    "tomo_cost_grad_rows": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, _c_vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, _c_dp, _c_dp,
                                           _c_vp]),
    "tomo_triplets": (ctypes.c_int, [_c_vp, _c_dp, _c_i64, _c_vp, _c_vp, _c_vp, ctypes.POINTER(_c_i64)]),
    "tomo_vox_splat": (ctypes.c_int, [_c_vp, _c_dp, _c_dp, _c_vp, _c_vp, _c_vp]),
    "tomo_vox_triplets": (ctypes.c_int, [_c_vp, _c_dp, _c_dp, _c_vp, _c_vp]),
    "tomo_phantom_ellipsoids": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, ctypes.c_int]),
    "tomo_vec_recip_guard": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_float, ctypes.c_int]),
    "tomo_vec_fill": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_float]),
    "tomo_vec_residual_scale": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_dp]),
    "tomo_vec_update": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, _c_vp, _c_dp]),
    "tomo_vec_update_acc": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, _c_vp, ctypes.c_int]),
    "tomo_vec_update_acc_fetch": (ctypes.c_int, [_c_vp, _c_dp]),
    "tomo_vec_axpy": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_float, _c_i64]),
    "tomo_vec_xpay": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_float, _c_i64]),
    "tomo_vec_sub": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64]),
    "tomo_vec_mul": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64]),
    "tomo_vec_dot": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_dp]),
    "tomo_vec_diff_sumsq": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_dp]),
    "tomo_acc_zero": (ctypes.c_int, [_c_vp, ctypes.c_int, ctypes.c_int]),
    "tomo_vec_dot_acc": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_int, ctypes.c_int]),
    "tomo_acc_fetch": (ctypes.c_int, [_c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp]),
    "tomo_vec_soft_threshold": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_float]),
    "tomo_tv_denoise_fista": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                             ctypes.c_int, ctypes.POINTER(ctypes.c_int), _c_dp]),
    "tomo_tv_norm_3d": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp]),
    "tomo_release_workspace": (ctypes.c_int, [_c_vp]),
    "tomo_csr_assemble": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int, _c_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    "tomo_csr_fetch": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp]),
    "tomo_trilinear_ray_interp": (ctypes.c_int, [_c_vp, _c_vp, _c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]),
    "tomo_trilinear_ray_sparse": (ctypes.c_int, [_c_vp, _c_vp, _c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_vp, _c_vp, _c_dp, _c_vp]),
    "tomo_bilinear_vox_interp": (ctypes.c_int, [_c_vp, ctypes.c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int, _c_vp, _c_vp, _c_vp]),
    "tomo_bilinear_sparse": (ctypes.c_int, [_c_vp, ctypes.c_int, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, ctypes.c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "tomo_comm_get_unique_id": (ctypes.c_int, [_c_vp]),
    "tomo_comm_init": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, ctypes.c_int]),
    "tomo_comm_destroy": (ctypes.c_int, [_c_vp]),
    "tomo_allreduce_sum_f32": (ctypes.c_int, [_c_vp, _c_vp, _c_i64]),
    "tomo_allreduce_sum_f32_async": (ctypes.c_int, [_c_vp, _c_vp, _c_i64]),
    "tomo_comm_join": (ctypes.c_int, [_c_vp]),
    "tomo_comm_wait_next": (ctypes.c_int, [_c_vp]),
    "tomo_reduce_scatter_sum_f32_async": (ctypes.c_int, [_c_vp, _c_vp, _c_i64]),
    "tomo_allgather_f32_async": (ctypes.c_int, [_c_vp, _c_vp, _c_i64]),
    "tomo_comm_wait_next_gather": (ctypes.c_int, [_c_vp]),
    "tomo_allreduce_sum_f64_host": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int]),
    "tomo_allreduce_max_f64_host": (ctypes.c_int, [_c_vp, _c_dp, ctypes.c_int]),
    "tomo_timer_start": (ctypes.c_int, [_c_vp]),
    "tomo_timer_stop": (ctypes.c_int, [_c_vp, ctypes.POINTER(ctypes.c_float)]),
    "tomo_profile_enable": (ctypes.c_int, [_c_vp, ctypes.c_int]),
    "tomo_profile_reset": (ctypes.c_int, [_c_vp]),
    "tomo_profile_get": (ctypes.c_int, [_c_vp, ctypes.c_char_p, ctypes.POINTER(_c_i64), _c_dp]),
}

_lib = None
_lock = threading.Lock()


class TomoError(RuntimeError):
    pass


def load():
    """Load libtomo_hip.so and bind every symbol; raises TomoError (never falls back) on failure."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise TomoError("libtomo_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "or `make -C tomography_alignment_amd/csrc`; there is no CPU fallback" % LIB_PATH)
            try:
                lib = ctypes.CDLL(LIB_PATH)
            except OSError as e:
                raise TomoError("cannot load %s: %s" % (LIB_PATH, e))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)          # AttributeError if include/tomo.h and the .so disagree
                fn.restype = res
                fn.argtypes = args
            if lib.tomo_abi_version() != 1:
                raise TomoError("libtomo_hip.so ABI version mismatch")
            _lib = lib
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    return x


LIVE_CONTEXTS = weakref.WeakSet()      # every open Context of this process (tests close what a test leaves behind, tests/conftest.py)


class Context(object):
    """One tomo_ctx: one GPU, one HIP stream.  All device memory of the package hangs off it: close() frees the DeviceArrays still
    alive on it, then the context's own streams, events and workspaces."""

    def __init__(self, device=None):
        self._arrays = weakref.WeakSet()
        self.lib = load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        n = ctypes.c_int(0)
        rc = self.lib.tomo_device_count(ctypes.byref(n))
        if rc != 0 or n.value < 1:
            raise TomoError("no HIP device visible (rc=%d: %s); this package has no CPU path"
                            % (rc, (self.lib.tomo_last_error(None) or b"").decode()))
        h = _c_vp()
        self._h = None
        self.check(self.lib.tomo_ctx_create(int(device) % n.value, ctypes.byref(h)), None)
        self._h = h
        self.device = int(device) % n.value
        self._geom_key = None
        LIVE_CONTEXTS.add(self)

    def check(self, rc, h="self"):
        if rc != 0:
            hh = self._h if h == "self" else h
            msg = self.lib.tomo_last_error(hh) or b""
            raise TomoError("libtomo_hip error %d: %s" % (rc, msg.decode(errors="replace")))

    def close(self):
        if getattr(self, "_h", None) is not None:
            for a in list(getattr(self, "_arrays", ())):      # buffers that outlive the context would never be freed (free() needs the handle)
                a.free()
            self.lib.tomo_ctx_destroy(self._h)
            self._h = None
            LIVE_CONTEXTS.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if self._h is None:
            raise TomoError("context closed")
        return self._h

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self.check(self.lib.tomo_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def sync(self):
        self.check(self.lib.tomo_sync(self.handle))

    def set_option(self, key, value):
        self.check(self.lib.tomo_set_option(self.handle, key.encode(), int(value)))

    def set_cu_mask(self, cus=None):
        """Restrict the compute stream to the CUs in `cus` (an iterable of CU indices; None: unrestricted).  As measured on this runtime only
        LEADING RANGES (range(k)) take effect, and the masked stream is a blocking one (include/tomo.h: tomo_ctx_set_cu_mask)."""
        if cus is None:
            self.check(self.lib.tomo_ctx_set_cu_mask(self.handle, None, 0))
            return
        cus = sorted(set(int(c) for c in cus))
        if not cus or cus[0] < 0:
            raise ValueError("set_cu_mask: give at least one CU index >= 0 (None restores the unrestricted stream)")
        words = (max(cus) // 32 + 1) if cus else 1
        m = (ctypes.c_uint32 * words)()
        for c in cus:
            m[c // 32] |= 1 << (c % 32)
        self.check(self.lib.tomo_ctx_set_cu_mask(self.handle, m, words))

    def make_current(self):
        """Bind the CALLING thread to this context's GPU (HIP's current device is per thread; a new thread starts on device 0)."""
        self.check(self.lib.tomo_ctx_make_current(self.handle))

    # ---- memory
    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray(self, shape, dtype)
        self.check(self.lib.tomo_memset0(self.handle, a.ptr, a.nbytes))
        return a

    def to_device(self, host, dtype=np.float32):
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray(self, host.shape, dtype)
        a.upload(host)
        return a

    # ---- geometry
    def set_geometry(self, geometry):
        """geometry: anything exposing the attributes of utilities/geometry.py `Geometry`."""
        g = geom_struct(geometry)
        key = bytes(g)
        if key != self._geom_key:
            self.check(self.lib.tomo_set_geometry(self.handle, ctypes.byref(g)))
            self._geom_key = key

    # ---- timing
    def timer_start(self):
        self.check(self.lib.tomo_timer_start(self.handle))

    def timer_stop(self):
        ms = ctypes.c_float(0)
        self.check(self.lib.tomo_timer_stop(self.handle, ctypes.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self.check(self.lib.tomo_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        self.check(self.lib.tomo_profile_reset(self.handle))

    def profile_get(self, kernel):
        n, ms = _c_i64(0), ctypes.c_double(0)
        self.check(self.lib.tomo_profile_get(self.handle, kernel.encode(), ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value


class DeviceArray(object):
    """A typed HBM buffer owned by a Context (tomo_malloc / tomo_free)."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if len(self.shape) else 1
        self.nbytes = self.size * self.dtype.itemsize
        p = _c_vp()
        ctx.check(ctx.lib.tomo_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p
        self._owner = True
        ctx._arrays.add(self)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError("upload: size mismatch %d != %d" % (host.size, self.size))
        self.ctx.check(self.ctx.lib.tomo_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data_as(_c_vp), self.nbytes))
        return self

    def download(self, out=None):
        if out is None:
            out = np.empty(self.shape, self.dtype)
        if not (out.flags["C_CONTIGUOUS"] and out.dtype == self.dtype and out.size == self.size):
            raise ValueError("download: need a C-contiguous array of matching dtype/size")
        self.ctx.check(self.ctx.lib.tomo_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(_c_vp), self.ptr, self.nbytes))
        return out

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError("copy_from: size mismatch")
        self.ctx.check(self.ctx.lib.tomo_memcpy_d2d(self.ctx.handle, self.ptr, other.ptr, self.nbytes))
        return self

    def zero_(self):
        self.ctx.check(self.ctx.lib.tomo_memset0(self.ctx.handle, self.ptr, self.nbytes))
        return self

    def view(self, offset_elems, n_elems):
        """Non-owning window [offset, offset+n) of this buffer."""
        if offset_elems < 0 or offset_elems + n_elems > self.size:
            raise ValueError("view out of range")
        v = object.__new__(DeviceArray)
        v.ctx, v.shape, v.dtype, v.size = self.ctx, (int(n_elems),), self.dtype, int(n_elems)
        v.nbytes = v.size * self.dtype.itemsize
        v.ptr = _c_vp(self.ptr.value + offset_elems * self.dtype.itemsize)
        v._owner = False
        v._base = self
        return v

    def free(self):
        if getattr(self, "_owner", False) and self.ptr is not None and self.ctx._h is not None:
            self.ctx.lib.tomo_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


KERNEL_SOURCE_FILES = ("Makefile", "kernels_grad.hip.h", "kernels_ray.hip.h", "kernels_tile.hip.h", "kernels_tile_flat.hip.h", "kernels_tile_gather.hip.h",
                       "tomo_ctx.h", "tomo_project.hip", "tomo_raycore.h")


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources of the projector and gradient kernels and of the code that configures and launches them
    (KERNEL_SOURCE_FILES in csrc/: the kernel headers, tomo_raycore.h, tomo_project.hip, the context layout, the build flags), in name order.
    The committed PMC counters (profiles/sq_counters.json, profiles/pmc_traffic.json) hold exactly those kernels and carry the hash of the
    sources they were taken on; bench.py refuses them when these have changed since (counts of an old kernel must never be
    divided by the time of a new one).  Round 6: the files that define none of those kernels (tomo_ctx.hip: context, vector lines,
    collectives; tomo_f2py.hip, tomo_csr.hip, tomo_reg.hip) are no longer part of it -- an edit there does not make the counters stale."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in sorted(KERNEL_SOURCE_FILES):
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def geom_struct(geometry):
    """Fill struct tomo_geom from a Geometry-like object (utilities/geometry.py:14-47,77-105)."""
    g = TomoGeom()
    g.nx, g.ny, g.nz = (int(v) for v in geometry.vox_shape)
    g.ndx, g.ndz = (int(v) for v in geometry.det_shape)
    for a in range(3):
        g.vox_origin[a] = float(geometry.vox_origin[a])
        g.vox_pitch[a] = float(np.asarray(geometry.vox_pix, dtype=np.float64).ravel()[a])
    src = np.asarray(geometry.source_centers)
    det = np.asarray(geometry.det_centers)
    ndz = g.ndz
    g.det_x0 = float(src[0, 0])
    g.det_z0 = float(src[2, 0])
    g.det_dx = float(src[0, ndz] - src[0, 0]) if g.ndx > 1 else float(np.asarray(geometry.det_pix).ravel()[0])
    g.det_dz = float(src[2, 1] - src[2, 0]) if g.ndz > 1 else float(np.asarray(geometry.det_pix).ravel()[1])
    g.src_y = float(src[1, 0])
    g.det_y = float(det[1, 0])
    g.step = float(geometry.step_size)
    return g


def poses_array(phi, alpha, beta, xyz_shift, cor_shift):
    """(n,7) float64 rows phi,alpha,beta,tx,ty,tz,cor_x; cor_shift is (n,3) or (3,) -- only its x
    component acts (utilities/ray_voxel_utilities.py:72-73)."""
    phi = np.atleast_1d(np.asarray(phi, dtype=np.float64))
    n = phi.size
    out = np.zeros((n, POSE_STRIDE), np.float64)
    out[:, 0] = phi
    out[:, 1] = np.broadcast_to(np.atleast_1d(np.asarray(alpha, np.float64)), (n,))
    out[:, 2] = np.broadcast_to(np.atleast_1d(np.asarray(beta, np.float64)), (n,))
    out[:, 3:6] = np.asarray(xyz_shift, np.float64).reshape(-1, 3) if np.size(xyz_shift) else 0.0
    cor = np.asarray(cor_shift, np.float64)
    out[:, 6] = cor.reshape(-1, 3)[:, 0] if cor.ndim == 2 else cor.ravel()[0]
    return np.ascontiguousarray(out)


def dptr(a):
    return a.ctypes.data_as(_c_dp)
