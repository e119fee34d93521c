"""Stands where the reference's `src` package (its two f2py extension modules) stands: `from src import ray_wt_grad`
(utilities/ray_voxel_utilities.py:3) resolves here when the package directory is first on sys.path."""
