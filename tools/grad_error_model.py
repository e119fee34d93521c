#!/usr/bin/env python3
"""Where does float32 cost the gradient kernels their accuracy?  A CPU model (numpy, no GPU) of one pose of
tests/test_gpu_fuzz.py::test_gradient_kernels_agree_on_random_geometry -- by default the geometry of the round-3 soak failure
(seed 81, case 0, pose 1: volume 59 x 71 x 61, detector 21 x 5, step 1.0, tilt 0.5 deg; gpurun_out/r3_soak_final.log:108).

Per ray it evaluates S0 = sum_j grad_j, S1 = sum_j sf_j grad_j (the seven sums of csrc/kernels_grad.hip.h; semantics
src/ray_wt_grad.f90:136-220) in float64 and under three separate float32 models, everything else float64:
  positions : sample positions as the round-3 kernels form them (float64 block anchor + float32 in-block offset
              x = fma(jj, (float)d, f0), tomo_raycore.h::tomo_block_anchor), lerps and sums exact;
  fixed     : sample positions in 32.32 fixed point, re-anchored per block (the round-4 kernels), lerps and sums exact;
  lerps     : exact positions, the lerps in float32 in the order of k_proj_grad (z, y, x), sums exact;
  sums      : exact positions and lerps, per-block float32 accumulation of the seven sums.
and prints, per gradient row group, the worst error of each model in units of the test's metric (max |error| / row-group maximum)
next to the conditioning of the sums (sum_j |grad_j| against |sum_j grad_j|, in-volume samples per ray)."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

JB = 32
f32 = np.float32


def fuzz_cases(seed):
    """The random draws of the test, in the test's order: yields (k, case) for the cases the test runs."""
    rng = np.random.default_rng(7000 + seed)
    for k in range(6):
        shape = tuple(int(v) for v in rng.integers(16, 72, 3))
        ndet = (int(rng.integers(5, 80)), int(rng.integers(3, 140)))
        step = float(rng.choice([1.0, 1.0, 0.5, 1.3]))
        n = int(rng.integers(1, 4))
        phi = rng.uniform(0, np.pi, n)
        tilt = np.deg2rad(rng.choice([0.0, 0.5, 2.0, 6.0]))
        alpha, beta = rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n)
        xyz = rng.uniform(-4, 4, (n, 3))
        cor = np.zeros((n, 3))
        cor[:, 0] = rng.uniform(-1, 1, n)
        ii, jj, kk = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
        fr, ph = rng.uniform(0.05, 0.35, 3), rng.uniform(0, 6.28, 3)
        x = (0.6 + 0.4 * np.cos(fr[0] * ii + ph[0]) * np.cos(fr[1] * jj + ph[1]) * np.cos(fr[2] * kk + ph[2])).astype(np.float32)
        og = orc.Geo(n, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor, step_size=step)
        want_p = np.zeros((n, ndet[0] * ndet[1]))
        for i in range(n):
            want_p[i], _ = orc.projection_gradient(og, x, alpha[i], beta[i], phi[i], xyz[i], cor[i], precision=np.float64)
        if np.max(np.abs(want_p)) == 0:
            continue
        rng.standard_normal(want_p.shape)
        yield k, (shape, ndet, step, phi, alpha, beta, xyz, cor, x, np.rad2deg(tilt))


def corners(vol_pad, cx, cy, cz):
    """The 8 corner values (halo 2) of cells (cx, cy, cz): arrays [2][2][2] of shape cx.shape."""
    return [[[vol_pad[cx + a + 2, cy + b + 2, cz + c + 2] for c in (0, 1)] for b in (0, 1)] for a in (0, 1)]


def grad_terms(v, wx, wy, wz, dt, offset=False):
    """value and spatial gradient of the trilinear interpolant, lerps in dtype `dt` in k_proj_grad's order (z, then y, then x).
    offset: the lerps run on the corners MINUS corner 000 (round 4: rounding then scales with the local differences, not the values)."""
    v = [[[v[a][b][c].astype(dt) for c in (0, 1)] for b in (0, 1)] for a in (0, 1)]
    if offset:
        v0 = v[0][0][0]
        v = [[[(v[a][b][c] - v0).astype(dt) for c in (0, 1)] for b in (0, 1)] for a in (0, 1)]
    wx, wy, wz = wx.astype(dt), wy.astype(dt), wz.astype(dt)

    def fma(a, b, c):        # one rounding, as the hardware's fused multiply-add
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(dt)
    d00, d01, d10, d11 = v[0][0][1] - v[0][0][0], v[0][1][1] - v[0][1][0], v[1][0][1] - v[1][0][0], v[1][1][1] - v[1][1][0]
    c00, c01, c10, c11 = fma(wz, d00, v[0][0][0]), fma(wz, d01, v[0][1][0]), fma(wz, d10, v[1][0][0]), fma(wz, d11, v[1][1][0])
    dz0, dz1 = fma(wy, d01 - d00, d00), fma(wy, d11 - d10, d10)
    gz = fma(wx, dz1 - dz0, dz0)
    dy0, dy1 = c01 - c00, c11 - c10
    e0, e1 = fma(wy, dy0, c00), fma(wy, dy1, c10)
    gy = fma(wx, dy1 - dy0, dy0)
    gx = e1 - e0
    val = fma(wx, gx, e0)
    if offset:
        val = (val + v0).astype(dt)
    return val, gx, gy, gz


def analyse(seed, k, case, pose, verbose):
    shape, ndet, step, phi, alpha, beta, xyz, cor, x, tilt = case
    n_poses = len(phi)
    og = orc.Geo(n_poses, np.array(shape), np.ones(3), np.array(ndet), np.ones(2), cor_shift=cor, step_size=step)
    i = pose
    say = print if verbose else (lambda *a: None)
    say("seed %d case %d pose %d: volume %s detector %s step %.2f tilt +-%.1f deg; phi %.4f alpha %.5f beta %.5f t %s"
          % (seed, k, i, shape, ndet, step, tilt, phi[i], alpha[i], beta[i], np.round(xyz[i], 3)))
    p0, rhat, n, r_len0, src, det = orc.ray_setup(og, alpha[i], beta[i], phi[i], xyz[i], cor[i])
    ray_vec = (det - src)[:, 0]
    der = orc.derivative_ray_points(src, ray_vec, alpha[i], beta[i], phi[i], xyz[i])       # (9, 3, n_rays)
    want_p, want_g = orc.projection_gradient(og, x, alpha[i], beta[i], phi[i], xyz[i], cor[i], precision=np.float64)
    n_det = ndet[0] * ndet[1]
    nx, ny, nz = shape
    vp = np.zeros((nx + 4, ny + 4, nz + 4), np.float64)
    vp[2:-2, 2:-2, 2:-2] = x
    d = step * rhat                                   # (3, n_rays)
    j = np.arange(n)
    pos = p0[:, :, None] + j[None, None, :] * d[:, :, None]        # exact (float64) sample positions (3, n_rays, n)
    inside = np.all((pos >= -1) & (pos < np.array([nx, ny, nz])[:, None, None]), axis=0)
    sf = j * step / r_len0

    def sums(cell, w, dt_lerp, dt_sum, offset=False, jbs=JB):
        cx, cy, cz = (np.clip(cell[a], -2, [nx, ny, nz][a]) for a in range(3))
        v = corners(vp, cx, cy, cz)
        val, gx, gy, gz = grad_terms(v, w[0], w[1], w[2], dt_lerp, offset)
        m = inside
        out = np.zeros((7, n_det))
        mag = np.zeros((7, n_det))
        terms = [val, gx, gy, gz, sf * gx, sf * gy, sf * gz]
        for k, t in enumerate(terms):
            t = np.where(m, t.astype(np.float64), 0.0)
            mag[k] = np.abs(t).sum(axis=1)
            if dt_sum == np.float64:
                out[k] = t.sum(axis=1)
            else:       # float32 sums of `jbs` samples, added in float32 into the block's sum (JB samples from the ray's first in-volume
                        # sample), the blocks' sums added in float64 -- vectorised over rays (a ray's in-volume samples are contiguous)
                first = np.argmax(m, axis=1)
                nb = -(-int(m.sum(axis=1).max()) // JB)
                cols = first[:, None] + np.arange(nb * JB)[None, :]
                ok = cols < t.shape[1]
                ts = np.where(ok, np.take_along_axis(t, np.minimum(cols, t.shape[1] - 1), axis=1), 0.0).astype(np.float32)
                ts = ts.reshape(n_det, nb, JB // jbs, jbs)
                a = np.zeros(ts.shape[:3], np.float32)
                for q in range(jbs):
                    a = (a + ts[..., q]).astype(np.float32)
                A = np.zeros(ts.shape[:2], np.float32)
                for q in range(JB // jbs):
                    A = (A + a[..., q]).astype(np.float32)
                out[k] = A.astype(np.float64).sum(axis=1)
        return out, mag

    def to_grad(S):
        """the per-ray 9x3 Jacobian applied to the sums (kernels_grad.hip.h::grad_finish; src/ray_wt_grad.f90:136-149)"""
        g = np.zeros((6, n_det))
        for k in range(3):
            g[k] = sum(der[k, a] * S[1 + a] for a in range(3))
        for k in range(3, 6):
            g[k] = sum(der[k, a] * S[1 + a] + der[k + 3, a] * S[4 + a] for a in range(3))
        return g

    cell = np.floor(pos).astype(np.int64)
    w = pos - cell
    S_exact, mag = sums(cell, w, np.float64, np.float64)
    g_exact = to_grad(S_exact)
    say("model check: float64 model vs oracle: value %.1e gradient %.1e (of the row-group maximum)"
          % (np.max(np.abs(S_exact[0] - want_p)) / np.max(np.abs(want_p)),
             max(np.max(np.abs(g_exact[:3] - want_g[:3])) / np.max(np.abs(want_g[:3])), np.max(np.abs(g_exact[3:] - want_g[3:])) / np.max(np.abs(want_g[3:])))))

    # positions as the kernels form them: blocks of JB samples from the ray's first in-volume sample (vectorised over rays)
    def kernel_positions(fixed):
        first = np.argmax(inside, axis=1)
        nb = -(-int(inside.sum(axis=1).max()) // JB)
        jb0 = first[:, None] + JB * np.arange(nb)[None, :]                      # (n_det, nb) first sample of each block
        t = np.arange(JB)
        c2, w2 = cell.copy(), w.copy()
        cols = (jb0[:, :, None] + t[None, None, :]).reshape(n_det, -1)
        ok = cols < n
        for a in range(3):
            sblk = p0[a][:, None] + jb0 * d[a][:, None]                        # tomo_block_anchor
            f = np.floor(sblk + 0.5 * (JB - 1) * d[a][:, None])
            if fixed:
                q0 = np.floor((sblk - f) * 4294967296.0 + 0.5).astype(np.int64)
                dq = np.floor(d[a] * 4294967296.0 + 0.5).astype(np.int64)
                q = q0[:, :, None] + t[None, None, :].astype(np.int64) * dq[:, None, None]
                ca = (q >> 32) + f[:, :, None].astype(np.int64)
                wa = ((q & 0xffffffff).astype(np.float32) * f32(2.0 ** -32)).astype(np.float64)     # v_cvt_f32_u32 + scale
            else:
                f0 = (sblk - f).astype(np.float32)
                xf = (t[None, None, :].astype(np.float64) * d[a].astype(np.float32).astype(np.float64)[:, None, None] + f0.astype(np.float64)[:, :, None]).astype(np.float32)
                fl = np.floor(xf)
                ca = fl.astype(np.int64) + f[:, :, None].astype(np.int64)
                wa = (xf - fl).astype(np.float64)
            ca, wa = ca.reshape(n_det, -1), wa.reshape(n_det, -1)
            for r in range(n_det):
                c2[a, r, cols[r][ok[r]]] = ca[r][ok[r]]
                w2[a, r, cols[r][ok[r]]] = wa[r][ok[r]]
        return c2, w2

    results = {}
    c2, w2 = kernel_positions(False)
    moved = np.abs((c2 + w2) - pos)[:, inside]
    say("float32 in-block positions: max |dp| %.2e voxel, mean %.2e; cells that differ: %d" % (moved.max(), moved.mean(), int(np.sum((c2 != cell)[:, inside]))))
    results["positions (float32 offsets)"] = sums(c2, w2, np.float64, np.float64)[0]
    c3, w3 = kernel_positions(True)
    moved = np.abs((c3 + w3) - pos)[:, inside]
    say("32.32 fixed-point positions: max |dp| %.2e voxel, mean %.2e; cells that differ: %d" % (moved.max(), moved.mean(), int(np.sum((c3 != cell)[:, inside]))))
    results["positions (32.32 fixed)"] = sums(c3, w3, np.float64, np.float64)[0]
    wf = w.astype(np.float32).astype(np.float64)
    if verbose:
        results["lerps (float32, exact positions)"] = sums(cell, wf, np.float32, np.float64)[0]
        results["sums (float32 per block)"] = sums(cell, w, np.float64, np.float32)[0]
        results["lerps on corner differences"] = sums(cell, wf, np.float32, np.float64, offset=True)[0]
        for b in (8, 4, 2):
            results["sums (float32 per %d samples)" % b] = sums(cell, w, np.float64, np.float32, jbs=b)[0]
        results["fixed + diff lerps + sums per 4"] = sums(c3, w3, np.float32, np.float32, offset=True, jbs=4)[0]
        results["fixed + diff lerps + sums per 8"] = sums(c3, w3, np.float32, np.float32, offset=True, jbs=8)[0]
    results["round 3 kernels (f32 offsets, lerps on values, sums per 32)"] = sums(c2, w2, np.float32, np.float32)[0]
    results["round 4 kernels (f32 offsets, lerps on differences, sums per 8)"] = sums(c2, w2, np.float32, np.float32, offset=True, jbs=8)[0]

    gmax = np.array([np.max(np.abs(want_g[:3]))] * 3 + [np.max(np.abs(want_g[3:]))] * 3)
    face = orc.ray_face_distance(og, alpha[i], beta[i], phi[i], xyz[i], cor[i])
    well = face >= 2e-5
    say("rays %d, in-volume samples per ray %d..%d, rays within 2e-5 voxel of a face %d; row-group maxima: translations %.4g, angles %.4g"
        % (n_det, inside.sum(axis=1).min(), inside.sum(axis=1).max(), int(np.sum(~well)), gmax[0], gmax[3]))
    say("conditioning: sum_j|grad_j| / max_rays|sum_j grad_j| per axis: %s" % np.round(mag[1:4].max(axis=1) / np.abs(S_exact[1:4]).max(axis=1), 1))
    say("%-64s %10s %10s %10s   (max |error| / row-group maximum; the test's metric, bar 1e-5)" % ("float32 model", "value", "transl.", "angles"))
    out = {}
    for name, S in results.items():
        g = to_grad(S)
        ev = np.max(np.abs(S[0] - S_exact[0])) / np.max(np.abs(S_exact[0]))
        et = np.max(np.abs(g[:3] - g_exact[:3])[:, well], initial=0.0) / gmax[0]
        ea = np.max(np.abs(g[3:] - g_exact[3:])[:, well], initial=0.0) / gmax[3]
        say("%-64s %10.2e %10.2e %10.2e" % (name, ev, et, ea))
        out[name] = (ev, et, ea)
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--soak":        # every case and pose of seeds [a, b): the two composite models only
        a, b = int(sys.argv[2]), int(sys.argv[3])
        worst = {}
        for seed in range(a, b):
            for k, case in fuzz_cases(seed):
                for pose in range(len(case[3])):
                    if np.prod(case[1]) * 2 * case[0][1] / case[2] > 4e6:      # keep the model's arrays small: skip the largest detectors
                        continue
                    res = analyse(seed, k, case, pose, False)
                    for name, (ev, et, ea) in res.items():
                        if name.startswith("round"):
                            e = max(et, ea)
                            if e > worst.get(name, (0,))[0]:
                                worst[name] = (e, seed, k, pose)
                            if e > 5e-6:
                                print("seed %d case %d pose %d %s: %.2e" % (seed, k, pose, name[:15], e), flush=True)
        for name, (e, seed, k, pose) in worst.items():
            print("worst over seeds %d..%d: %-64s %.2e (seed %d case %d pose %d)" % (a, b - 1, name, e, seed, k, pose))
        return
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 81
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    pose = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for k, case in fuzz_cases(seed):
        if k == want:
            analyse(seed, k, case, pose, True)


if __name__ == "__main__":
    main()
