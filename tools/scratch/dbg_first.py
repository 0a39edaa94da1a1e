import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
variant = sys.argv[1]
if variant == "torch_first":
    import torch
import megacrn_amd
from megacrn_amd._lib import lib, set_precision
from tools.gemm_probe import run
set_precision("bf16x3")
if variant == "setdebug":
    lib.mcrn_set_debug(0)
try:
    print(variant, run(13248, 128, int(sys.argv[2]), 0, 0, 3, reps=5))
except Exception as e:
    print(variant, "FAILED", str(e)[-80:])
