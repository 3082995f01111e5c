"""ctypes binding of the C ABI declared in ``include/lightglue_amd.h``.

The shared library ``liblightglue_amd.so`` is built in-tree by ``lightglue_amd/csrc/Makefile``
(``__graft_entry__.build()``).  There is NO fallback: if the library is missing or fails to load,
``load()`` raises, and so does every product entry point that needs it.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# LIGHTGLUE_AMD_LIB overrides the library file (A/B timing of two builds in one process tree); it must still be
# a build of this package's csrc/ — there is no other implementation to fall back to.
_LIB_PATH = Path(os.environ.get("LIGHTGLUE_AMD_LIB") or Path(__file__).resolve().parent / "liblightglue_amd.so")

LG_PREC = {"fp32": 0, "bf16": 1, "fp16": 2, "f16x3": 4}   # include/lightglue_amd.h LG_PREC_* (3 was split-bf16, removed in round 3)
LG_OK, LG_ERR_INVALID, LG_ERR_HIP, LG_ERR_STATE, LG_ERR_RANGE, LG_ERR_DEVICE = 0, 1, 2, 3, 4, 5
LG_FLAG_NO_PRUNING, LG_FLAG_EXT, LG_FLAG_CHECK_FINITE = 1, 2, 4

# every symbol include/lightglue_amd.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "lg_last_error", "lg_version", "lg_engine_create", "lg_engine_destroy", "lg_engine_set_weight",
    "lg_engine_finalize_weights", "lg_engine_reserve", "lg_engine_forward", "lg_unpack_wire", "lg_engine_set_option",
    "lg_engine_debug_stop_after", "lg_engine_debug_read", "lg_engine_debug_caps",
    "lg_profile_num_classes", "lg_profile_class_name", "lg_engine_profile_enable", "lg_engine_profile_read",
    "lg_sp_sample_descriptors", "lg_sp_detect_workspace_bytes", "lg_sp_detect",
    "lg_sp_pack_conv_weight", "lg_sp_encode_workspace_bytes", "lg_sp_encode", "lg_sp_pack_conv_weight_split", "lg_sp_encode_split", "lg_debug_mfma_sustained",
)


class LgConfig(C.Structure):
    _fields_ = [
        ("input_dim", C.c_int32), ("descriptor_dim", C.c_int32), ("n_layers", C.c_int32), ("num_heads", C.c_int32),
        ("add_scale_ori", C.c_int32),
        ("depth_confidence", C.c_double), ("width_confidence", C.c_double), ("filter_threshold", C.c_double),
        ("pruning_min_kpts", C.c_int32), ("precision", C.c_int32), ("attn_precision", C.c_int32),
    ]


_fp = C.c_void_p  # device pointers travel as integers


class LgForwardIO(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n0", C.c_int32), ("n1", C.c_int32), ("flags", C.c_uint32),
        ("kpts0", _fp), ("kpts1", _fp), ("desc0", _fp), ("desc1", _fp), ("size0", _fp), ("size1", _fp),
        ("scales0", _fp), ("oris0", _fp), ("scales1", _fp), ("oris1", _fp),
        ("matches0", _fp), ("matches1", _fp), ("scores0", _fp), ("scores1", _fp), ("stop", _fp),
        ("prune0", _fp), ("prune1", _fp), ("matches", _fp), ("match_scores", _fp), ("n_matches", _fp),
        ("num0", _fp), ("num1", _fp), ("log_assignment", _fp),
        # round-5 extension (read by the engine only when flags & LG_FLAG_EXT)
        ("matches0_i64", _fp), ("matches1_i64", _fp), ("matches_i64", _fp), ("stop_i64", _fp),
        ("prune0_i64", _fp), ("prune1_i64", _fp), ("prune0_f32", _fp), ("prune1_f32", _fp),
        ("wire", _fp), ("wire_stride", C.c_int64), ("status", _fp),
    ]


class LgUnpackIO(C.Structure):
    """lg_unpack_io (include/lightglue_amd.h): the receiving side of the pair-sharded wire row"""
    _fields_ = [
        ("wire", _fp), ("wire_stride", C.c_int64), ("rows", C.c_int32),
        ("n0", C.c_int32), ("n1", C.c_int32), ("with_prune", C.c_int32), ("pairs_out", C.c_int32),
        ("order", _fp),
        ("matches0", _fp), ("matches1", _fp), ("stop", _fp),
        ("scores0", _fp), ("scores1", _fp),
        ("prune0_i64", _fp), ("prune1_i64", _fp), ("prune0_f32", _fp), ("prune1_f32", _fp),
        ("matches", _fp), ("match_scores", _fp),
        ("info", _fp),
    ]


def wire_width(n0: int, n1: int) -> int:
    """LG_WIRE_WIDTH: int32 elements of one lg_forward_io.wire row"""
    return 3 * n0 + 3 * n1 + 2


class LightGlueAmdError(RuntimeError):
    pass


_lib = None


def library_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise LightGlueAmdError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {_LIB_PATH.parent / 'csrc'}`; lightglue_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(os.fspath(_LIB_PATH))
    lib.lg_last_error.restype = C.c_char_p
    lib.lg_version.restype = C.c_char_p
    lib.lg_engine_create.argtypes = [C.POINTER(LgConfig), C.POINTER(C.c_void_p)]
    lib.lg_engine_destroy.argtypes = [C.c_void_p]
    lib.lg_engine_destroy.restype = None
    lib.lg_engine_set_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    lib.lg_engine_finalize_weights.argtypes = [C.c_void_p]
    lib.lg_engine_reserve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.lg_engine_forward.argtypes = [C.c_void_p, C.POINTER(LgForwardIO), C.c_void_p]
    lib.lg_unpack_wire.argtypes = [C.POINTER(LgUnpackIO), C.c_void_p]
    lib.lg_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.lg_engine_debug_stop_after.argtypes = [C.c_void_p, C.c_int32]
    lib.lg_engine_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.lg_engine_debug_caps.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.lg_profile_num_classes.restype = C.c_int32
    lib.lg_profile_class_name.restype = C.c_char_p
    lib.lg_profile_class_name.argtypes = [C.c_int32]
    lib.lg_engine_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.lg_engine_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]
    lib.lg_sp_sample_descriptors.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lg_sp_detect_workspace_bytes.argtypes = [C.c_int32] * 4
    lib.lg_sp_detect_workspace_bytes.restype = C.c_int64
    lib.lg_sp_detect.argtypes = [C.c_void_p] + [C.c_int32] * 5 + [C.c_float] + [C.c_int32] * 3 + [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
    lib.lg_sp_pack_conv_weight.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lg_sp_encode_workspace_bytes.argtypes = [C.c_int32] * 3
    lib.lg_sp_encode_workspace_bytes.restype = C.c_int64
    lib.lg_sp_encode.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lg_sp_pack_conv_weight_split.argtypes = lib.lg_sp_pack_conv_weight.argtypes
    lib.lg_sp_encode_split.argtypes = lib.lg_sp_encode.argtypes
    lib.lg_debug_mfma_sustained.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map C error codes to the exception types the reference raises."""
    if rc == LG_OK:
        return
    msg = load().lg_last_error().decode()
    if rc == LG_ERR_INVALID:
        raise AssertionError(msg)
    raise LightGlueAmdError(f"lightglue_amd error {rc}: {msg}")
