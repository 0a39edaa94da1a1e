"""megacrn_amd - MI355X-native MegaCRN encoder/decoder hot path (HIP kernels behind the reference's
nn.Module surface).  Importing this package loads libmegacrn_hip.so and fails loudly if it is missing."""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is not built)
_lib.set_precision(_lib.default_precision())
from .modules import AGCN, AGCRNCell, ADCRNN_Encoder, ADCRNN_Decoder, MegaCRN, print_params  # noqa: F401

__all__ = ["AGCN", "AGCRNCell", "ADCRNN_Encoder", "ADCRNN_Decoder", "MegaCRN", "print_params"]
