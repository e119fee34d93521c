"""
Device-resident counterpart of the reference's utilities/tv_denoise.py: the total-variation proximal step of TV-FISTA
(`denoise_fista`, :98-170, called from recon/regularized.py:93) and `tv_norm_3d` (:62-64, recon/regularized.py:107), with the
reference's signatures.  Volumes are float32 [nx][ny][nz]; a numpy array goes up and the result comes back, a DeviceArray stays
in HBM (what a regularised solver iterating on the GPU passes).  The 2-D variants and the anisotropic norm of the reference
file are not on the tomography path (the solvers call the 3-D forms only) and are not provided.
"""
import numpy as np

try:
    from .. import _lib
except ImportError:      # imported as top-level `utilities`
    import _lib

_ctx = None


def _context(ctx=None):
    """The caller's context, or a module-level one created on first use and closed by close_context() / at interpreter exit."""
    global _ctx
    if ctx is not None:
        return ctx
    if _ctx is not None and _ctx._h is None:      # closed behind our back: open another
        _ctx = None
    if _ctx is None:
        import atexit
        _ctx = _lib.Context()          # raises without a GPU: no CPU fallback
        atexit.register(close_context)
    return _ctx


def close_context():
    """Destroy the module-level context (its stream, staging buffers and TV workspace)."""
    global _ctx
    if _ctx is not None:
        c, _ctx = _ctx, None
        c.close()


def release_workspace(ctx=None):
    """Hand the TV workspace (7 volumes, kept in the context between calls) back to the device: call it when the regularised
    solver is done and the same process goes on to something large (tomo_release_workspace)."""
    c = ctx if ctx is not None else _ctx
    if c is not None:
        c.check(c.lib.tomo_release_workspace(c.handle))


def denoise_fista(im, weight=50, niter=200, eps=1.e-5, check_gap_frequency=3, ctx=None, return_info=False):
    """argmin_res 0.5*||im - res||^2 + weight*TV(res) (isotropic TV, FISTA on the dual) -- utilities/tv_denoise.py:98-170.
    Returns the reference's `new`: the iterate of the last dual-gap check -- as FLOAT32 for numpy input of any dtype (the
    device computes in float32; the reference computes in the input's dtype)."""
    import ctypes
    if isinstance(im, _lib.DeviceArray):
        if len(im.shape) != 3:
            raise ValueError("denoise_fista: a DeviceArray must carry its 3-D shape")
        c = im.ctx
        d_im, shape, host = im, im.shape, False
    else:
        a = np.ascontiguousarray(im, np.float32)
        if a.ndim != 3:
            raise ValueError("denoise_fista: 3-D volumes only (the reference's 2-D branch is not on the tomography path)")
        c = _context(ctx)
        d_im, shape, host = c.to_device(a), a.shape, True
    d_out = c.empty(shape)
    it, gap = ctypes.c_int(0), ctypes.c_double(0)
    c.check(c.lib.tomo_tv_denoise_fista(c.handle, d_im.ptr, d_out.ptr, int(shape[0]), int(shape[1]), int(shape[2]), float(weight), int(niter),
                                        float(eps), int(check_gap_frequency), ctypes.byref(it), ctypes.byref(gap)))
    out = d_out.download() if host else d_out
    return (out, it.value, gap.value) if return_info else out


def tv_norm_3d(x, ctx=None):
    """||gradient(x)||_2 -- utilities/tv_denoise.py:62-64."""
    import ctypes
    if isinstance(x, _lib.DeviceArray):
        c, d_x, shape = x.ctx, x, x.shape
    else:
        a = np.ascontiguousarray(x, np.float32)
        c = _context(ctx)
        d_x, shape = c.to_device(a), a.shape
    v = ctypes.c_double(0)
    c.check(c.lib.tomo_tv_norm_3d(c.handle, d_x.ptr, int(shape[0]), int(shape[1]), int(shape[2]), ctypes.byref(v)))
    return v.value
