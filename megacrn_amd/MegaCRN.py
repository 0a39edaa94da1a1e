"""Import shim: ``from MegaCRN import MegaCRN`` (model/traintest_MegaCRN.py:15) works unchanged when
this package directory is first on ``sys.path``; the classes live in ``megacrn_amd.modules``."""
from .modules import (AGCN, AGCRNCell, ADCRNN_Encoder, ADCRNN_Decoder, MegaCRN,  # noqa: F401
                      print_params)
