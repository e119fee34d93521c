#!/usr/bin/env python3
"""Run tests/test_gpu_fuzz.py's two cross-checks (tile kernels, gradient kernels) over many seeds (development aid):
python tools/fuzz_soak.py [first] [last] [tile|grad|both] -- keeps going after a failed seed and lists the failures at the end."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as f      # noqa: E402

a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 40)
which = sys.argv[3] if len(sys.argv) > 3 else "both"
fns = [fn for key, fn in (("tile", f.test_tile_kernels_agree_with_ray_driven_kernels), ("grad", f.test_gradient_kernels_agree_on_random_geometry)) if which in (key, "both")]
bad = []
for seed in range(a, b):
    for fn in fns:
        try:
            fn(seed)
        except AssertionError as e:
            bad.append((seed, fn.__name__, str(e)[:300]))
            print("FAILED seed", seed, fn.__name__, str(e)[:300], flush=True)
print("seeds %d..%d: %d failures" % (a, b - 1, len(bad)))
for x in bad:
    print(x)
