"""ccnet_amd -- MI355X-native ControlNet-conditioned latent-diffusion sampling (the ccnet hot path).

Host side mirrors the reference's L2 surface; all compute goes through the
hand-written gfx950 kernels of libccnet_hip.so (include/ccnet_hip.h).
"""
__version__ = "0.1.0"
