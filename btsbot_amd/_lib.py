"""ctypes binding of libbtsbot_hip.so (include/btsbot_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BTSBOT_AMD_LIB: another build of the same library, for A/B timing of kernel variants -- tools/build_variant.sh)
LIB_PATH = os.environ.get("BTSBOT_AMD_LIB") or os.path.join(_HERE, "libbtsbot_hip.so")
ABI_VERSION = 1

OK = 0
WIRING = {"mm_ConvNeXt": 0, "ConvNeXt": 1, "frozen_fusion": 2, "um_nn": 3, "mm_MaxViT": 4,
          "MaxViT": 5, "frozen_fusion_MaxViT": 6}
PRECISION = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2,
             "float16": 2, "fp8": 3, "f16x2": 4}

# every symbol include/btsbot_hip.h declares (tests/test_abi.py checks the export list)
SYMBOLS = [
    "btsbot_last_error", "btsbot_abi_version", "btsbot_create", "btsbot_destroy",
    "btsbot_param_count", "btsbot_param_floats", "btsbot_param_info_at", "btsbot_pack_params",
    "btsbot_pack_params_train",
    "btsbot_workspace_bytes", "btsbot_reserve", "btsbot_forward", "btsbot_set_debug",
    "btsbot_read_tap", "btsbot_bce_fwd_bwd", "btsbot_adamw_step",
    "btsbot_set_profile", "btsbot_profile_categories", "btsbot_profile_category_name",
    "btsbot_profile_collect",
    "btsbot_op_gemm", "btsbot_op_dwconv_ln", "btsbot_op_stem", "btsbot_op_ln_patch",
    "btsbot_reserve_train", "btsbot_forward_train", "btsbot_backward", "btsbot_debug_stamps",
    "btsbot_grad_buckets", "btsbot_wait_grad_bucket", "btsbot_allreduce_grads", "btsbot_use_workspace", "btsbot_set_option",
    "btsbot_augment", "btsbot_eval_metrics", "btsbot_prep_triplets",
]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("wiring", C.c_int32), ("precision", C.c_int32),
        ("depths", C.c_int32 * 4), ("dims", C.c_int32 * 4), ("image_size", C.c_int32),
        ("head_norm", C.c_int32), ("n_meta", C.c_int32),
        ("meta_fc1", C.c_int32), ("meta_fc2", C.c_int32),
        ("comb_fc1", C.c_int32), ("comb_fc2", C.c_int32),
        ("meta_dropout", C.c_float), ("comb_dropout", C.c_float),
    ]


class ParamInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 96), ("offset", C.c_int64), ("numel", C.c_int64),
        ("ndim", C.c_int32), ("shape", C.c_int32 * 4), ("is_buffer", C.c_int32),
    ]


class BtsbotHipError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load the shared library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise BtsbotHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C btsbot_amd/csrc`.  btsbot_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.btsbot_last_error.restype = C.c_char_p
    L.btsbot_last_error.argtypes = []
    L.btsbot_abi_version.restype = i32
    L.btsbot_abi_version.argtypes = []
    L.btsbot_create.restype = i32
    L.btsbot_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.btsbot_destroy.restype = i32
    L.btsbot_destroy.argtypes = [vp]
    L.btsbot_param_count.restype = i32
    L.btsbot_param_count.argtypes = [vp]
    L.btsbot_param_floats.restype = i64
    L.btsbot_param_floats.argtypes = [vp]
    L.btsbot_param_info_at.restype = i32
    L.btsbot_param_info_at.argtypes = [vp, i32, C.POINTER(ParamInfo)]
    L.btsbot_pack_params.restype = i32
    L.btsbot_pack_params.argtypes = [vp, vp, vp]
    L.btsbot_pack_params_train.restype = i32
    L.btsbot_pack_params_train.argtypes = [vp, vp, vp]
    L.btsbot_workspace_bytes.restype = i64
    L.btsbot_workspace_bytes.argtypes = [vp, i32]
    L.btsbot_reserve.restype = i32
    L.btsbot_reserve.argtypes = [vp, i32]
    L.btsbot_forward.restype = i32
    L.btsbot_forward.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.c_uint64, vp]
    L.btsbot_set_debug.restype = i32
    L.btsbot_set_debug.argtypes = [vp, i32]
    L.btsbot_read_tap.restype = i64
    L.btsbot_read_tap.argtypes = [vp, C.c_char_p, vp, i64, vp]
    L.btsbot_set_profile.restype = i32
    L.btsbot_set_profile.argtypes = [vp, i32]
    L.btsbot_profile_categories.restype = i32
    L.btsbot_profile_categories.argtypes = []
    L.btsbot_profile_category_name.restype = C.c_char_p
    L.btsbot_profile_category_name.argtypes = [i32]
    L.btsbot_profile_collect.restype = i32
    L.btsbot_profile_collect.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    L.btsbot_op_gemm.restype = i32
    L.btsbot_op_gemm.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.btsbot_op_dwconv_ln.restype = i32
    L.btsbot_op_dwconv_ln.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.btsbot_op_stem.restype = i32
    L.btsbot_op_stem.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.btsbot_op_ln_patch.restype = i32
    L.btsbot_op_ln_patch.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, vp]
    L.btsbot_debug_stamps.restype = i32
    L.btsbot_debug_stamps.argtypes = [vp, vp]
    L.btsbot_reserve_train.restype = i32
    L.btsbot_reserve_train.argtypes = [vp, i32, i32]
    L.btsbot_forward_train.restype = i32
    L.btsbot_forward_train.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp]
    L.btsbot_backward.restype = i32
    L.btsbot_backward.argtypes = [vp, vp, vp, i32, i32, vp]
    L.btsbot_grad_buckets.restype = i32
    L.btsbot_grad_buckets.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64)]
    L.btsbot_wait_grad_bucket.restype = i32
    L.btsbot_wait_grad_bucket.argtypes = [vp, i32, vp]
    L.btsbot_set_option.restype = i32
    L.btsbot_set_option.argtypes = [vp, C.c_char_p, i32]
    L.btsbot_use_workspace.restype = i32
    L.btsbot_use_workspace.argtypes = [vp, i32, vp, i64]
    L.btsbot_allreduce_grads.restype = i32
    L.btsbot_allreduce_grads.argtypes = [vp, vp, vp, i32, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), vp]
    L.btsbot_bce_fwd_bwd.restype = i32
    L.btsbot_bce_fwd_bwd.argtypes = [vp, vp, f32, i32, i32, vp, vp, vp]
    L.btsbot_adamw_step.restype = i32
    L.btsbot_adamw_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]
    L.btsbot_augment.restype = i32
    L.btsbot_augment.argtypes = [vp, vp, vp, vp, i32, vp]
    L.btsbot_prep_triplets.restype = i32
    L.btsbot_prep_triplets.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.btsbot_eval_metrics.restype = i32
    L.btsbot_eval_metrics.argtypes = [vp, vp, f32, i64, vp, vp]
    if L.btsbot_abi_version() != ABI_VERSION:
        raise BtsbotHipError(f"ABI mismatch: library {L.btsbot_abi_version()} vs binding {ABI_VERSION}")
    _lib = L
    return L


def check(status: int, what: str) -> int:
    if status < 0:
        msg = lib().btsbot_last_error().decode("utf-8", "replace")
        raise BtsbotHipError(f"{what} failed ({status}): {msg}")
    return status


def make_config(wiring: str, precision: str, depths, dims, head_norm: bool, n_meta: int,
                meta_fc1: int, meta_fc2: int, comb_fc1: int, comb_fc2: int,
                meta_dropout: float, comb_dropout: float, image_size: int = 63) -> Config:
    if precision not in PRECISION:
        raise ValueError(f"precision must be one of {sorted(set(PRECISION))}, got {precision!r}")
    cfg = Config()
    cfg.abi_version = ABI_VERSION
    cfg.wiring = WIRING[wiring]
    cfg.precision = PRECISION[precision]
    cfg.depths = (C.c_int32 * 4)(*depths)
    cfg.dims = (C.c_int32 * 4)(*dims)
    cfg.image_size = image_size
    cfg.head_norm = int(head_norm)
    cfg.n_meta = n_meta
    cfg.meta_fc1, cfg.meta_fc2 = meta_fc1, meta_fc2
    cfg.comb_fc1, cfg.comb_fc2 = comb_fc1, comb_fc2
    cfg.meta_dropout, cfg.comb_dropout = meta_dropout, comb_dropout
    return cfg


class Handle:
    """Owns one btsbot_handle."""

    def __init__(self, cfg: Config):
        self._h = C.c_void_p()
        check(lib().btsbot_create(C.byref(cfg), C.byref(self._h)), "btsbot_create")

    @property
    def ptr(self):
        return self._h

    def params(self):
        L = lib()
        out = []
        info = ParamInfo()
        for i in range(L.btsbot_param_count(self._h)):
            check(L.btsbot_param_info_at(self._h, i, C.byref(info)), "btsbot_param_info_at")
            shape = tuple(info.shape[k] for k in range(info.ndim))
            out.append((info.name.decode(), int(info.offset), int(info.numel), shape,
                        bool(info.is_buffer)))
        return out

    def param_floats(self) -> int:
        return int(lib().btsbot_param_floats(self._h))

    def close(self):
        if self._h:
            lib().btsbot_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
