"""Import shim: ``from MegaCRN import MegaCRN`` (model/traintest_MegaCRN.py:15) works unchanged when
this package directory is first on ``sys.path``; the classes live in ``megacrn_amd.modules``.

Two ways this file gets imported:
  * ``import megacrn_amd.MegaCRN``  - as a sub-module of the package (relative import works);
  * ``from MegaCRN import MegaCRN`` - as a TOP-LEVEL module, because ``.../megacrn_amd`` itself is on
    ``sys.path`` (the reference trainer's spelling).  There is no parent package then, so the package is
    imported by its absolute name after making its parent directory importable.
"""
import os as _os
import sys as _sys

if __package__:
    from .modules import (AGCN, AGCRNCell, ADCRNN_Encoder, ADCRNN_Decoder, MegaCRN,  # noqa: F401
                          print_params)
else:
    _parent = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _parent not in _sys.path:
        _sys.path.append(_parent)
    from megacrn_amd.modules import (AGCN, AGCRNCell, ADCRNN_Encoder, ADCRNN_Decoder, MegaCRN,  # noqa: F401
                                     print_params)

__all__ = ["AGCN", "AGCRNCell", "ADCRNN_Encoder", "ADCRNN_Decoder", "MegaCRN", "print_params"]

if __package__:
    # `import megacrn_amd.MegaCRN` rebinds the package attribute `megacrn_amd.MegaCRN` from the class to this
    # module (that is what the import system does for sub-modules).  Keep `megacrn_amd.MegaCRN(...)` working
    # afterwards: calling the module constructs the class.
    import types as _types

    class _CallableShim(_types.ModuleType):
        def __call__(self, *args, **kwargs):
            return MegaCRN(*args, **kwargs)

    _sys.modules[__name__].__class__ = _CallableShim
