"""Development aid: bench.py on another build of the library (TOMO_AB_LIB=path), e.g. the ablation builds of tools/gpu_r3l.sh.
Results of such a run are measurements of the ablated build, never a bench line."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tomography_alignment_amd import _lib  # noqa: E402

if os.environ.get("TOMO_AB_LIB"):
    _lib.LIB_PATH = os.environ["TOMO_AB_LIB"]
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
