"""ctypes binding of libmegacrn_hip.so (C ABI declared in include/megacrn_hip.h).

There is no CPU / eager fallback: if the library is missing, importing this module raises, and
every wrapper raises ``RuntimeError`` when the C entry point returns non-zero.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  MUST precede loading the .so: PyTorch bundles its own libamdhip64; loading ours first
#                     would bind the system HIP runtime and leave two runtimes in the process (launches then fail
#                     with "no ROCm-capable device").  With torch loaded, the soname resolves to the same runtime.

_HERE = os.path.dirname(os.path.abspath(__file__))
# MEGACRN_LIB: measurement-only build variants (csrc/Makefile `timeline` / `ablate`); default is the shipped library
LIB_PATH = os.environ.get("MEGACRN_LIB") or os.path.join(_HERE, "libmegacrn_hip.so")

EXPORTS = [
    "mcrn_last_error", "mcrn_version", "mcrn_build_id", "mcrn_last_launch_count", "mcrn_launch_histogram",
    "mcrn_model_workspace_bytes", "mcrn_model_forward", "mcrn_model_backward",
    "mcrn_supports_workspace_bytes", "mcrn_supports_forward", "mcrn_supports_backward",
    "mcrn_agcn_workspace_bytes", "mcrn_agcn_forward", "mcrn_agcn_backward",
    "mcrn_cell_workspace_bytes", "mcrn_cell_forward", "mcrn_cell_backward",
    "mcrn_memory_workspace_bytes", "mcrn_memory_forward", "mcrn_memory_backward",
    "mcrn_flat_clip_adam", "mcrn_loss_fwd_bwd", "mcrn_eval_metrics", "mcrn_gemm_f32", "mcrn_prof_begin", "mcrn_prof_end", "mcrn_prof_clock_mhz", "mcrn_set_gemm_cfg", "mcrn_set_debug", "mcrn_model_autotune", "mcrn_autotune_entries", "mcrn_autotune_clear", "mcrn_autotune_export", "mcrn_autotune_import", "mcrn_set_precision", "mcrn_get_precision", "mcrn_set_side_stream",
]


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "N", "T_in", "T_out", "input_dim", "output_dim", "ycov_dim",
                                       "H", "mem_num", "mem_dim", "cheb_k", "precision")]


PARAM_FIELDS = ("Memory", "Wq", "We1", "We2", "enc_gate_w", "enc_gate_b", "enc_update_w", "enc_update_b",
                "dec_gate_w", "dec_gate_b", "dec_update_w", "dec_update_b", "proj_w", "proj_b")

# state_dict key of each field (num_layers == 1), SURVEY.md 8(b)
PARAM_KEYS = ("memory.Memory", "memory.Wq", "memory.We1", "memory.We2",
              "encoder.dcrnn_cells.0.gate.weights", "encoder.dcrnn_cells.0.gate.bias",
              "encoder.dcrnn_cells.0.update.weights", "encoder.dcrnn_cells.0.update.bias",
              "decoder.dcrnn_cells.0.gate.weights", "decoder.dcrnn_cells.0.gate.bias",
              "decoder.dcrnn_cells.0.update.weights", "decoder.dcrnn_cells.0.update.bias",
              "proj.0.weight", "proj.0.bias")


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


class Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C megacrn_amd/csrc`.  megacrn_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols {missing}")
    vp, i, f, sz, ll = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong
    lib.mcrn_last_error.restype = C.c_char_p
    lib.mcrn_version.restype = i
    lib.mcrn_build_id.restype = C.c_char_p
    lib.mcrn_launch_histogram.restype = C.c_longlong
    lib.mcrn_launch_histogram.argtypes = [C.c_char_p, C.c_longlong, C.c_int]
    lib.mcrn_last_launch_count.restype = i
    lib.mcrn_model_workspace_bytes.restype = sz
    lib.mcrn_model_workspace_bytes.argtypes = [C.POINTER(Dims)]
    lib.mcrn_model_forward.restype = i
    lib.mcrn_model_forward.argtypes = [C.POINTER(Dims), C.POINTER(Params), vp, vp, vp, C.POINTER(C.c_int),
                                       vp, sz, vp, vp, vp, vp, vp, vp]
    lib.mcrn_model_backward.restype = i
    lib.mcrn_model_backward.argtypes = [C.POINTER(Dims), C.POINTER(Params), C.POINTER(C.c_int), vp, vp, vp, vp,
                                        vp, vp, sz, C.POINTER(Grads), vp]
    lib.mcrn_supports_workspace_bytes.restype = sz
    lib.mcrn_supports_workspace_bytes.argtypes = [i, i, i]
    lib.mcrn_supports_forward.restype = i
    lib.mcrn_supports_forward.argtypes = [i, i, i, vp, vp, vp, vp, sz, vp, vp, vp]
    lib.mcrn_supports_backward.restype = i
    lib.mcrn_supports_backward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    lib.mcrn_agcn_workspace_bytes.restype = sz
    lib.mcrn_agcn_workspace_bytes.argtypes = [i] * 5
    lib.mcrn_agcn_forward.restype = i
    lib.mcrn_agcn_forward.argtypes = [i] * 5 + [vp] * 5 + [vp, sz, vp, vp]
    lib.mcrn_agcn_backward.restype = i
    lib.mcrn_agcn_backward.argtypes = [i] * 5 + [vp] * 4 + [vp, sz] + [vp] * 5 + [vp]
    lib.mcrn_cell_workspace_bytes.restype = sz
    lib.mcrn_cell_workspace_bytes.argtypes = [i] * 5
    lib.mcrn_cell_forward.restype = i
    lib.mcrn_cell_forward.argtypes = [i] * 5 + [vp] * 8 + [vp, sz, vp, vp]
    lib.mcrn_cell_backward.restype = i
    lib.mcrn_cell_backward.argtypes = [i] * 5 + [vp] * 5 + [vp, sz] + [vp] * 8 + [vp]
    lib.mcrn_memory_workspace_bytes.restype = sz
    lib.mcrn_memory_workspace_bytes.argtypes = [i] * 5
    lib.mcrn_memory_forward.restype = i
    lib.mcrn_memory_forward.argtypes = [i] * 5 + [vp] * 3 + [vp, sz] + [vp] * 5 + [vp]
    lib.mcrn_memory_backward.restype = i
    lib.mcrn_memory_backward.argtypes = [i] * 5 + [vp] * 7 + [vp, sz] + [vp] * 3 + [vp]
    lib.mcrn_flat_clip_adam.restype = i
    lib.mcrn_flat_clip_adam.argtypes = [vp, vp, vp, vp, ll, f, f, f, f, i, f, f, vp, vp, vp]
    lib.mcrn_loss_fwd_bwd.restype = i
    lib.mcrn_loss_fwd_bwd.argtypes = [i] * 5 + [vp] * 5 + [f] * 5 + [vp] * 4 + [vp]
    lib.mcrn_eval_metrics.restype = i
    lib.mcrn_eval_metrics.argtypes = [i] * 5 + [vp] * 5 + [f] * 5 + [C.POINTER(C.c_int), i, vp, vp, vp]
    lib.mcrn_gemm_f32.restype = i
    lib.mcrn_gemm_f32.argtypes = [i, i, i, i, i, vp, vp, vp, f, f, i, vp, vp]
    lib.mcrn_prof_begin.restype = i
    lib.mcrn_prof_begin.argtypes = [i]
    lib.mcrn_set_precision.restype = i
    lib.mcrn_set_precision.argtypes = [i]
    lib.mcrn_get_precision.restype = i
    lib.mcrn_set_side_stream.restype = i
    lib.mcrn_set_side_stream.argtypes = [i]
    if os.environ.get("MCRN_SIDE_STREAM") == "0":        # measurement / debugging: everything on the caller's stream
        lib.mcrn_set_side_stream(0)
    lib.mcrn_model_autotune.restype = i
    lib.mcrn_model_autotune.argtypes = [C.POINTER(Dims), vp, sz, vp]
    lib.mcrn_autotune_entries.restype = i
    lib.mcrn_autotune_clear.restype = i
    lib.mcrn_autotune_export.restype = ll
    lib.mcrn_autotune_export.argtypes = [C.POINTER(C.c_int), ll]
    lib.mcrn_autotune_import.restype = i
    lib.mcrn_autotune_import.argtypes = [C.POINTER(C.c_int), ll]
    lib.mcrn_set_debug.restype = i
    lib.mcrn_set_debug.argtypes = [i]
    lib.mcrn_set_gemm_cfg.restype = i
    lib.mcrn_set_gemm_cfg.argtypes = [i]
    lib.mcrn_prof_end.restype = i
    lib.mcrn_prof_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]
    lib.mcrn_prof_clock_mhz.restype = i
    lib.mcrn_prof_clock_mhz.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    return lib


lib = _load()

F32, BF16X3, BF16 = 0, 1, 2
PRECISIONS = {"f32": F32, "bf16x3": BF16X3, "bf16": BF16}


def build_id() -> str:
    """16 hex digits identifying the sources this library was built from (csrc/Makefile); measured artefacts carry it."""
    return lib.mcrn_build_id().decode()


def launch_histogram(reset: bool = False) -> dict:
    """{kernel family: launches since the last reset} - which kernels the library's plan really chose (debug / tests)."""
    n = lib.mcrn_launch_histogram(None, 0, 0)
    buf = C.create_string_buffer(int(n))
    lib.mcrn_launch_histogram(buf, n, 1 if reset else 0)
    return {k: int(v) for k, v in (line.split("=") for line in buf.value.decode().splitlines() if line)}


def default_precision() -> int:
    """MEGACRN_PRECISION=f32|bf16x3|bf16 (default bf16x3: fp32-equivalent split on the bf16 matrix cores; bf16:
    bf16-resident propagation for large graphs, own tolerance)."""
    return PRECISIONS[os.environ.get("MEGACRN_PRECISION", "bf16x3")]


def set_precision(name_or_id) -> None:
    v = PRECISIONS[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
    check(lib.mcrn_set_precision(v), "mcrn_set_precision")


def autotune_export() -> list:
    """The tile table chosen by mcrn_model_autotune as a list of ints (see include/megacrn_hip.h)."""
    n = lib.mcrn_autotune_export(None, 0)
    buf = (C.c_int * max(n, 1))()
    lib.mcrn_autotune_export(buf, n)
    return list(buf[:n])


def autotune_import(words) -> None:
    arr = (C.c_int * max(len(words), 1))(*words)
    check(lib.mcrn_autotune_import(arr, len(words)), "mcrn_autotune_import")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib.mcrn_last_error().decode()}")
