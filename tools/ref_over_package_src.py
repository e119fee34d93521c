#!/usr/bin/env python3
"""INTEGRATION.md level 2 1/2, checked: ALL of the reference's Python, only its f2py package `src` replaced (VERDICT r5 missing 3).

Runs in the AUTHORING container only (needs /root/reference; nothing here travels to the GPU box):
  * `src`, `src.ray_wt_grad`, `src.vox_wt_grad` are this package's twins (tomography_alignment_amd/src/*.py), put where the reference
    looks them up (`sys.modules`, the three lines INTEGRATION.md gives);
  * `utilities.projection_operators` -- whose lines 5-8 import utilities.ray_voxel_utilities (`from src import ray_wt_grad`) AND
    utilities.voxel_utilities (`from src import vox_wt_grad`: the import that failed until round 6) -- `utilities.geometry`, `recon.sirt`
    and `utilities.alignment_functions` are imported from /root/reference UNMODIFIED (the two numpy/scipy shims of make_golden.py);
  * every reference module must have resolved to /root/reference, both `src` modules to this package;
  * the reference's own callers then reach the twins with arguments the twins accept: utilities.voxel_utilities.forward_sparse /
    forward_proj_grad and utilities.ray_voxel_utilities.forward_sparse run their numpy and enter the twin, which -- no GPU here, no CPU
    fallback -- raises this package's TomoError (a ValueError from the twins' argument checks would come first and fail this script).
What the twins RETURN for those arguments is checked on the GPU against the f2py module's own outputs for the same arrays
(tests/golden/g13, tests/test_gpu_parity.py::test_vox_wt_grad_twin_vs_reference_golden; G3 / G1 b for ray_wt_grad)."""
import inspect
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("REF", "/root/reference")
if not os.path.isdir(os.path.join(REF, "utilities")):
    raise SystemExit("needs the reference tree (authoring container only)")

import numpy as np  # noqa: E402
import numpy.lib._index_tricks_impl as _it  # noqa: E402
np.lib.index_tricks = _it                                    # utilities/generate_phantom.py:173 (numpy >= 2)
import scipy.optimize  # noqa: E402
import scipy.optimize._linesearch as _ls  # noqa: E402
_m = types.ModuleType("scipy.optimize.linesearch")          # utilities/alignment_functions.py:4 (scipy moved the module)
_m.line_search_armijo = _ls.line_search_armijo
_m.line_search_wolfe1 = _ls.line_search_wolfe1
sys.modules["scipy.optimize.linesearch"] = _m
scipy.optimize.linesearch = _m

# ---- the three lines of INTEGRATION.md level 2 1/2
sys.path.insert(0, ROOT)
import tomography_alignment_amd.src as amd_src  # noqa: E402
from tomography_alignment_amd.src import ray_wt_grad as amd_ray, vox_wt_grad as amd_vox  # noqa: E402
sys.modules.update({"src": amd_src, "src.ray_wt_grad": amd_ray, "src.vox_wt_grad": amd_vox})
from tomography_alignment_amd import _lib  # noqa: E402

sys.path.insert(0, REF)
from utilities import projection_operators, ray_voxel_utilities, voxel_utilities, geometry, alignment_functions  # noqa: E402
from recon import sirt  # noqa: E402

ok = True
for mod, rel in ((projection_operators, "utilities/projection_operators.py"), (ray_voxel_utilities, "utilities/ray_voxel_utilities.py"),
                 (voxel_utilities, "utilities/voxel_utilities.py"), (geometry, "utilities/geometry.py"),
                 (alignment_functions, "utilities/alignment_functions.py"), (sirt, "recon/sirt.py")):
    here = os.path.realpath(inspect.getsourcefile(mod)) == os.path.realpath(os.path.join(REF, rel))
    print("  %-36s imported from the reference, unmodified: %s" % (rel, here))
    ok &= here
for name, mod, twin in (("ray_voxel_utilities.ray_wt_grad", ray_voxel_utilities.ray_wt_grad, amd_ray), ("voxel_utilities.vox_wt_grad", voxel_utilities.vox_wt_grad, amd_vox)):
    print("  %-36s is this package's twin: %s (%s)" % (name, mod is twin, os.path.relpath(inspect.getsourcefile(mod), ROOT)))
    ok &= mod is twin

import ctypes  # noqa: E402
n = ctypes.c_int(0)
have_gpu = _lib.load().tomo_device_count(ctypes.byref(n)) == 0 and n.value > 0
N = 8
geo = geometry.Geometry(1, np.array([N, N, N]), np.ones(3), np.array([N, N]), np.ones(2))
geo.cor_shift = np.array([0.25, 0.0, -0.5])
rec = np.random.default_rng(0).uniform(0, 1, (N, N, N)).astype(np.float32)
calls = (("utilities/voxel_utilities.py::forward_sparse", lambda: voxel_utilities.forward_sparse(geo, 0.01, -0.02, 0.7, np.array([0.5, 0.0, -0.3]))),
         ("utilities/voxel_utilities.py::forward_proj_grad", lambda: voxel_utilities.forward_proj_grad(geo, 0.01, -0.02, 0.7, np.array([0.5, 0.0, -0.3]), rec)),
         ("utilities/ray_voxel_utilities.py::forward_sparse", lambda: ray_voxel_utilities.forward_sparse(__import__("copy").deepcopy(geo), 0.01, -0.02, 0.7, np.array([0.5, 0.0, -0.3]))))
for name, call in calls:
    try:
        out = call()
        good = have_gpu and out is not None
        print("  %-52s ran on the GPU through the twin: %s" % (name, good))
    except _lib.TomoError as e:
        good = not have_gpu
        print("  %-52s reached the twin; no GPU here -> TomoError (%s): %s" % (name, str(e)[:60], good))
    ok &= good
print("RESULT: %s" % ("ok -- the reference's utilities/ and recon/ import and call through tomography_alignment_amd/src" if ok else "MISMATCH"))
sys.exit(0 if ok else 1)
