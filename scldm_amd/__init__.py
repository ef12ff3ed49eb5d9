"""scldm_amd - MI355X (gfx950) implementation of the scLDM latent-diffusion hot path.

Same Python API as the reference for this path (`scldm.nnets.DiT`, `scldm.transport.*`, ...),
backed by hand-written HIP kernels in libscldm_hip.so (C ABI: include/scldm_hip.h).
No CPU fallback exists: using a network without the built library or without a GPU raises.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (ctypes binding; the library itself is loaded on first use)
from .nnets import Decoder, DiT, Encoder  # noqa: F401
from .vae import TransformerVAE  # noqa: F401
from .transport import Sampler, Transport, create_transport  # noqa: F401
