"""
Drop-in for the reference's utilities/alignment_functions.py:7-485: `AlignmentUtilities`, the eleven
`cost_* / gradient_*` parameter-subset pairs, the two finite-difference checkers and
`gradient_descent`, with the reference's signatures and return values.

What is different underneath: one evaluation is ONE fused GPU kernel (tomo_cost_grad: projection +
6-DoF Jacobian + residual + the 7 reductions) instead of a Fortran call followed by a 6 x n_det host
GEMV.  Like the reference (utilities/alignment_functions.py:151-188) every call re-reads `rec`; when the
caller pins the volume (`proj_obj.pin_volume(rec)`: "I will not modify it") it stays in HBM and the last
evaluation is memoised, so that an optimiser's `fun(x)` followed by `jac(x)` at the same point costs a
single launch.  The vector-returning modes (`return_vector=True`) go
through `projection_gradient` exactly like the reference.

Row order of the pose Jacobian everywhere: tx, ty, tz, phi, alpha, beta
(utilities/ray_voxel_utilities.py:39-49).
"""
import numpy as np

try:                                   # scipy >= 1.8 moved the module (reference imports the old path, :4)
    from scipy.optimize.linesearch import line_search_armijo, line_search_wolfe1
except ImportError:                    # pragma: no cover - depends on the installed scipy
    from scipy.optimize._linesearch import line_search_armijo, line_search_wolfe1

_ROW = {"x": 0, "y": 1, "z": 2, "p": 3, "a": 4, "b": 5}


class AlignmentUtilities(object):
    """One measured projection `proj` against a projector object (reference :7-37)."""

    def __init__(self, proj, proj_obj, geometry):
        self.proj = proj
        self.proj_obj = proj_obj
        self.proj_mask = proj > 0
        self.geometry = geometry
        self._b_dev = None
        self._memo_key = None
        self._memo_val = None

    def _cor(self):
        return np.asarray(self.geometry.cor_shift, np.float64).reshape(-1)[:3]

    def cost(self, rec, angles, translations):
        """residual vector b - proj(p)   (reference :16-25)."""
        phi, alpha, beta = angles
        this_proj, _ = self.proj_obj.projection_gradient(rec=rec, alpha=alpha, beta=beta, phi=phi, xyz_shift=translations,
                                                         cor_shift=self.geometry.cor_shift)
        return np.asarray(self.proj).ravel() - this_proj

    def gradient(self, rec, angles, translations):
        """(residual, -dproj/dp)   (reference :27-37)."""
        phi, alpha, beta = angles
        this_proj, this_grad = self.proj_obj.projection_gradient(rec=rec, alpha=alpha, beta=beta, phi=phi,
                                                                 xyz_shift=translations, cor_shift=self.geometry.cor_shift)
        residual = np.asarray(self.proj).ravel() - this_proj
        this_grad *= -1
        return residual, this_grad

    def cost_and_gradient(self, rec, angles, translations):
        """Fused evaluation: (0.5*||b - proj||^2, J.residual[6]) with J = -dproj/dp -- the numbers the
        scalar modes of cost_* / gradient_* reduce to (reference :124,146)."""
        phi, alpha, beta = (float(v) for v in angles)
        t = np.asarray(translations, np.float64).reshape(3)
        po = self.proj_obj
        # the memo (an optimiser's fun(x) followed by jac(x) at the same point = one launch) is only sound while the volume
        # is pinned, i.e. the caller has promised not to change it; an unpinned `rec` is re-read on every call like the
        # reference does (utilities/projection_operators.py:112-122)
        memo_ok = hasattr(po, "volume_is_pinned") and po.volume_is_pinned(rec)
        vol = po.set_volume(rec)
        key = (phi, alpha, beta, t[0], t[1], t[2], po._vol_gen) if memo_ok else None
        if key is not None and key == self._memo_key:
            return self._memo_val
        be = po.backend
        if self._b_dev is None:
            self._b_dev = be.upload(np.asarray(self.proj, np.float32).ravel())
        pose = po.pose_row(alpha, beta, phi, t, self._cor())
        cost, g6 = po.pinned_call(be.cost_grad, pose, vol, self._b_dev)
        self._memo_key, self._memo_val = key, (float(cost[0]), g6[0].copy())
        return self._memo_val


# -------------------------------------------------------------------------------------------------
# parameter-subset pairs.  `letters` names which pose components the parameter vector carries, in
# the reference's order (always ascending Jacobian row): e.g. "xzab" = (tx, tz, alpha, beta).
# -------------------------------------------------------------------------------------------------
def _pose_from(parameters, letters, angles_in, xyz_in):
    pose = np.array([xyz_in[0], xyz_in[1], xyz_in[2], angles_in[0], angles_in[1], angles_in[2]], dtype=np.float64)
    for k, ch in enumerate(letters):
        pose[_ROW[ch]] += parameters[k]
    return pose[3:6].copy(), pose[0:3].copy()      # (phi, alpha, beta), (tx, ty, tz)


def _make_pair(letters):
    rows = [_ROW[ch] for ch in letters]

    def cost(parameters, align_obj, rec, angles_in, xyz_in, scale_factor=None, return_vector=False):
        angles, translations = _pose_from(parameters, letters, angles_in, xyz_in)
        if return_vector:
            return align_obj.cost(rec, angles, translations)
        if hasattr(align_obj, "cost_and_gradient"):
            return align_obj.cost_and_gradient(rec, angles, translations)[0]
        return 0.5 * np.linalg.norm(align_obj.cost(rec, angles, translations)) ** 2

    def gradient(parameters, align_obj, rec, angles_in, xyz_in, scale_factor=None, return_vector=False):
        angles, translations = _pose_from(parameters, letters, angles_in, xyz_in)
        scale = np.ones(len(rows)) if scale_factor is None else np.asarray(scale_factor, np.float64)
        if return_vector or not hasattr(align_obj, "cost_and_gradient"):
            residual, s = align_obj.gradient(rec, angles, translations)
            s = s[rows] * scale[:, np.newaxis]
            return s.T if return_vector else np.dot(s, residual)
        return align_obj.cost_and_gradient(rec, angles, translations)[1][rows] * scale

    cost.__name__ = "cost_" + letters
    gradient.__name__ = "gradient_" + letters
    cost.__doc__ = "0.5*||b - proj||^2 over parameters (%s); reference utilities/alignment_functions.py:113-460." % ",".join(letters)
    gradient.__doc__ = "d cost / d(%s); reference utilities/alignment_functions.py:127-485." % ",".join(letters)
    return cost, gradient


cost_xzpab, gradient_xzpab = _make_pair("xzpab")     # reference :113-148
cost_xzab, gradient_xzab = _make_pair("xzab")        # :151-188
cost_xz, gradient_xz = _make_pair("xz")              # :191-222
cost_x, gradient_x = _make_pair("x")                 # :244-275
cost_z, gradient_z = _make_pair("z")                 # :278-309
cost_ab, gradient_ab = _make_pair("ab")              # :312-345
cost_a, gradient_a = _make_pair("a")                 # :348-383
cost_b, gradient_b = _make_pair("b")                 # :386-421
cost_xzb, gradient_xzb = _make_pair("xzb")           # :448-485


def _half_sq(v):
    return 0.5 * np.linalg.norm(v) ** 2


def gradient_xz_fd(parameters, align_obj, rec, angles_in, xyz_in, scale_factor=None, return_vector=False):
    """Central finite differences in (tx, tz); eps 1e-4 / 1e-4 (reference :225-241)."""
    translations = np.array([xyz_in[0] + parameters[0], xyz_in[1], xyz_in[2] + parameters[1]])
    eps = np.array([1.e-4, 1.e-3, 1.e-4])
    grad = np.zeros(3)
    for i in range(3):
        e = np.zeros(3)
        e[i] = eps[i]
        grad[i] = (_half_sq(align_obj.cost(rec, angles_in, translations + e)) -
                   _half_sq(align_obj.cost(rec, angles_in, translations - e))) / (2 * eps[i])
    return np.array([grad[0], grad[2]])


def gradient_ab_fd(parameters, align_obj, rec, angles_in, xyz_in, scale_factor=None, return_vector=False):
    """Central finite differences in (alpha, beta) about angles_in (reference :424-445; like the
    reference, the perturbation is applied to angles_in, not angles_in + parameters)."""
    eps = np.array([1.e-3, 1.e-4, 1.e-4])
    grad = np.zeros(3)
    for i in range(3):
        e = np.zeros(3)
        e[i] = eps[i]
        grad[i] = (_half_sq(align_obj.cost(rec, angles_in + e, xyz_in)) -
                   _half_sq(align_obj.cost(rec, angles_in - e, xyz_in))) / (2 * eps[i])
    return grad[1:]


def gradient_descent(x, cost_function, gradient_function, args=(), options={}):
    """Steepest descent with scipy's Armijo / Wolfe line searches and the reference's fall-backs
    (utilities/alignment_functions.py:40-110).  Returns (x, f, stop): stop 1 = relative cost change
    <= eps, 2 = line search gave up, 0 = maxiter."""
    n_itmax = options.get('maxiter', 100)
    step_search = options.get('step_search', 'armijo')
    eps = options.get('eps', 1.e-6)
    verbose = options.get('verbose', False)
    align_obj, rec, angles_in, xyz_in, scale_factor = args

    cost = np.zeros(n_itmax + 1)
    f = cost_function(x, align_obj, rec, angles_in, xyz_in, scale_factor=scale_factor, return_vector=False)
    fp = gradient_function(x, align_obj, rec, angles_in, xyz_in, scale_factor=scale_factor, return_vector=False)
    cost[0] = f
    stop, it, ls_counter, alpha = 0, 0, 0, 0.0
    while not stop and it < n_itmax:
        if verbose:
            print(it, f, alpha, fp, x)
        search_dir = -fp
        if step_search == 'armijo':
            alpha, _, f_new = line_search_armijo(cost_function, x, search_dir, fp, cost[it], alpha0=1.0,
                                                 args=(align_obj, rec, angles_in, xyz_in, None))
        elif step_search == 'wolfe':
            alpha, _, _, f_new, f_old, fp_new = line_search_wolfe1(cost_function, gradient_function, x, search_dir,
                                                                   gfk=fp, amax=1.e-3, amin=1.e-12,
                                                                   args=(align_obj, rec, angles_in, xyz_in, None, None))
        if alpha is None:
            print('%s line search failed' % (step_search))
            ls_counter += 1
            ls_success = False
            alpha = 1.0
            while not ls_success and alpha > 1.e-15:      # brute-force shrink by 10 (reference :84-91)
                alpha = alpha / 10
                f_new = cost_function(x - alpha * fp, align_obj, rec, angles_in, xyz_in, scale_factor, False)
                ls_success = f_new < cost[it]
            if not ls_success or ls_counter >= 2:
                stop = 2
                it += 1
                if verbose:
                    print('either linesearch failed or two successive brute linesearch iterations')
        else:
            x = x - alpha * fp
            it += 1
            f = cost_function(x, align_obj, rec, angles_in, xyz_in, scale_factor, False)
            fp = gradient_function(x, align_obj, rec, angles_in, xyz_in, scale_factor, False)
            cost[it] = f
            if np.abs(cost[it] - cost[it - 1]) / max(cost[it], cost[it - 1], 1.0) <= eps:
                stop = 1
    return x, f, stop
