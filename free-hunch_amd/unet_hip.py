"""HIP execution of the UNet step list.  Bring-up state: delegates to the PyTorch-ROCm ops while the gfx950
conv / GroupNorm / attention kernels land one by one ("UNet kernels")."""
from .unet import _TorchOps


class HipOps(_TorchOps):
    pass
