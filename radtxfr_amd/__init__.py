"""radtxfr_amd -- MI355X-native engine for the line-by-line LWIR radiative-transfer hot path of
westi024/RadTxfr (Voigt line-sum -> layer optical depth -> Schwarzschild up/down-welling + Planck ->
at-sensor radiance -> MAKO ILS), behind the reference's own Python call signatures:

    from radtxfr_amd import radiative_transfer as rt, hapi, ILS_MAKO

Compute lives in libradtxfr_hip.so (hand-written HIP for gfx950, C ABI in include/radtxfr_hip.h);
PyTorch-ROCm provides device memory, streams and torch.distributed. No CPU fallback.
"""
__version__ = "0.1.0"
