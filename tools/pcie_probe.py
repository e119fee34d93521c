import sys, time
sys.path.insert(0, '.')
import numpy as np
from tomography_alignment_amd import _lib
ctx = _lib.Context(0)
n = 1024 ** 3
x = np.ones(n, np.float32)
d = ctx.to_device(x)
for rep in range(3):
    t0 = time.perf_counter(); d.upload(x); ctx.sync(); t1 = time.perf_counter(); y = d.download(); t2 = time.perf_counter()
    print("4 GiB pageable: H2D %.1f GB/s, D2H (into a fresh array) %.1f GB/s" % (4.295 / (t1 - t0), 4.295 / (t2 - t1)))
out = np.empty(n, np.float32)
t0 = time.perf_counter(); d.download(out); t1 = time.perf_counter()
print("D2H into an existing array %.1f GB/s" % (4.295 / (t1 - t0)))
