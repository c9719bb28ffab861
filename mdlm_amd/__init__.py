"""mdlm_amd: MI355X-native implementation of MDLM's denoise-and-sample hot path.

Hand-written HIP kernels (gfx950) behind a C-ABI (`include/mdlm_hip.h`,
`libmdlm_hip.so`), with a host-side mirror of the reference's `Diffusion` /
`DIT` interface.  There is no CPU fallback.
"""
__version__ = '0.1.0'
