"""GPU parity of the regularised solvers' vector kernels through the C-ABI: soft threshold
(recon/regularized.py:433) and the TV proximal step (utilities/tv_denoise.py:98-170) against the reference's own outputs
(golden G9) and the oracle."""
import numpy as np
import pytest

from conftest import golden, rel_max

pytestmark = pytest.mark.gpu


def test_soft_threshold_vs_reference_golden():
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.recon import regularized
    g = golden("g9_regularized")
    out = regularized.soft_thresholding(g["st_x"], float(g["st_lambda"]))
    assert out.dtype == np.float32 and np.array_equal(out, g["st_out"])                 # bit-exact: one subtraction per element
    ctx = _lib.Context()
    d = regularized.soft_thresholding(ctx.to_device(g["st_x"].reshape(64, 64)), float(g["st_lambda"]))
    assert isinstance(d, _lib.DeviceArray) and np.array_equal(d.download().ravel(), g["st_out"])
    assert np.array_equal(regularized.soft_thresholding(np.zeros(0, np.float32), 1.0), np.zeros(0, np.float32))
    # float64 in -> float32 out: the device computes in float32 and says so, never float32 values labelled float64
    o64 = regularized.soft_thresholding(g["st_x"].astype(np.float64), float(g["st_lambda"]))
    assert o64.dtype == np.float32 and np.array_equal(o64, g["st_out"])


def test_tv_denoise_fista_vs_reference_golden(capsys):
    from oracle import oracle as orc
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.utilities import tv_denoise
    g = golden("g9_regularized")
    im = g["tv_im"]
    assert np.isclose(tv_denoise.tv_norm_3d(im), float(g["tv_norm"]), rtol=1e-6)
    cases = {"a": dict(weight=0.2, niter=20, eps=0.0, check_gap_frequency=3), "b": dict(weight=0.05, niter=200, eps=1.e-3, check_gap_frequency=3),
             "c": dict(weight=0.5, niter=1, eps=0.0, check_gap_frequency=1), "d": dict(weight=0.5, niter=0)}
    want_iters = {"a": 20, "b": 6, "c": 1, "d": 0}
    errs = {}
    for tag, kw in cases.items():
        out, it, gap = tv_denoise.denoise_fista(im, return_info=True, **kw)
        _, _, want_gap = orc.tv_denoise_fista(im, return_info=True, **kw)
        errs[tag] = rel_max(out, g["tv_" + tag])
        assert out.shape == im.shape and out.dtype == np.float32 and it == want_iters[tag], (tag, it)
        assert errs[tag] < 1e-5 and (want_gap == 0 or abs(gap - want_gap) < 1e-5 * max(abs(want_gap), 1e-3)), (tag, errs[tag], gap, want_gap)
    with capsys.disabled():
        print("\n[TV-FISTA 16x12x20] rel-max vs the reference's outputs: " + ", ".join("%s %.1e" % kv for kv in errs.items()))
    # a larger, ragged volume (z not a multiple of the work-group width) resident on the device, against the oracle
    rng = np.random.default_rng(3)
    shape = (40, 33, 300)
    vol = np.zeros(shape, np.float32)
    vol[8:30, 5:25, 40:260] = 1.0
    noisy = (vol + 0.2 * rng.standard_normal(shape)).astype(np.float32)
    ctx = _lib.Context()
    d = ctx.to_device(noisy)
    got = tv_denoise.denoise_fista(d, weight=0.3, niter=12, eps=0.0)
    want = orc.tv_denoise_fista(noisy, weight=0.3, niter=12, eps=0.0)
    assert isinstance(got, _lib.DeviceArray) and rel_max(got.download(), want) < 1e-5
    assert np.isclose(tv_denoise.tv_norm_3d(d), orc.tv_norm_3d(noisy), rtol=1e-6)
    assert tv_denoise.tv_norm_3d(got) < 0.6 * tv_denoise.tv_norm_3d(d)                  # it does denoise
    with pytest.raises(_lib.TomoError):
        tv_denoise.denoise_fista(np.zeros((4, 1, 4), np.float32), weight=1.0)            # the reference's div needs >= 2 per axis
    # float64 in -> float32 out; a second call of another size reuses / regrows the context's workspace; the module
    # context can be closed explicitly and is re-created on demand
    o64 = tv_denoise.denoise_fista(im.astype(np.float64), **cases["a"])
    assert o64.dtype == np.float32 and rel_max(o64, g["tv_a"]) < 1e-5
    tv_denoise.close_context()
    assert tv_denoise._ctx is None and rel_max(tv_denoise.denoise_fista(im, **cases["a"]), g["tv_a"]) < 1e-5


def test_tv_workspace_can_be_released():
    """the TV-FISTA workspace (7 volumes, kept in the context between calls) is handed back by tomo_release_workspace
    (utilities/tv_denoise.py::release_workspace) and re-allocated by the next call that needs it; results unchanged."""
    from tomography_alignment_amd import _lib
    from tomography_alignment_amd.utilities import tv_denoise
    rng = np.random.default_rng(3)
    im = rng.uniform(0, 1, (24, 20, 28)).astype(np.float32)
    ctx = _lib.Context()
    a = tv_denoise.denoise_fista(im, weight=0.2, niter=12, ctx=ctx)
    tv_denoise.release_workspace(ctx)
    tv_denoise.release_workspace(ctx)                       # idempotent
    b = tv_denoise.denoise_fista(im, weight=0.2, niter=12, ctx=ctx)
    assert np.array_equal(a, b)
    ctx.close()
