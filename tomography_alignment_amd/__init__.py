"""
tomography_alignment_amd -- MI355X-native drop-in for the ray-driven projection hot path of
pandekan/tomography_alignment.

Host code is plain Python/NumPy mirroring the reference's API surface
(`utilities.projection_operators.ProjectionMatrix`, `utilities.alignment_functions`,
`recon.sirt.SIRT`, `recon.cgls.CGLS`); all compute goes through the C-ABI of
libtomo_hip.so (include/tomo.h) -- hand-written HIP kernels for gfx950.  There is no CPU path.
"""
__version__ = "0.1.0"
