"""
The vector kernel of the reference's recon/regularized.py that sits between two projector applications: `soft_thresholding`
(:433-440, the proximal step of run_lasso_ista :278, of its backtracking line search :321 and of run_lasso_fista :375), with the
reference's signature.  The regularised solver DRIVERS (Tikhonov / LASSO / TV-FISTA loops, their plotting) are out of scope
(SURVEY section 2); the TV proximal step they call lives in utilities/tv_denoise.py.
"""
import numpy as np

try:
    from .. import _lib
    from ..utilities import tv_denoise as _tv
except ImportError:      # imported as top-level `recon`
    import _lib
    from utilities import tv_denoise as _tv


def soft_thresholding(x, _lambda, ctx=None):
    """x - l where x > l, x + l where x < -l, else 0.  numpy in -> numpy float32 out (same shape); DeviceArray in -> a new
    DeviceArray.  The device computes in float32: a float64 input comes back as FLOAT32, not as float32 values labelled
    float64 (the reference computes in the input's dtype)."""
    if isinstance(x, _lib.DeviceArray):
        c = x.ctx
        out = c.empty(x.shape)
        c.check(c.lib.tomo_vec_soft_threshold(c.handle, out.ptr, x.ptr, x.size, float(_lambda)))
        return out
    a = np.asarray(x)
    c = _tv._context(ctx)
    d = c.to_device(a.ravel())
    c.check(c.lib.tomo_vec_soft_threshold(c.handle, d.ptr, d.ptr, d.size, float(_lambda)))
    return d.download().reshape(a.shape)
