"""ctypes binding of libcaco_hip.so (include/caco_hip.h).

This is the stub a maintainer of the reference would add to call the MI355X path (see
INTEGRATION.md).  There is deliberately NO fallback: if the shared library is missing or a call
fails, a RuntimeError / ValueError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.path.join(HERE, "libcaco_hip.so")
# CACO_LIB_PATH (another build of the library: an A/B variant, the simulator build of the test suite) is honoured ONLY together with
# CACO_ALLOW_VARIANT_LIB=1 - load() refuses it otherwise, so that nothing can stand in for the product by accident.
LIB_PATH = os.environ.get("CACO_LIB_PATH") or PRODUCT_LIB_PATH
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "caco_hip.h")

CACO_OK = 0
CACO_ERR_INVALID = -1
DTYPE_F32 = 0
DTYPE_BF16 = 1


class CacoConfigC(C.Structure):
    _fields_ = [
        ("audio_hidden", C.c_int32), ("audio_layers", C.c_int32), ("audio_heads", C.c_int32),
        ("audio_intermediate", C.c_int32), ("patch_size", C.c_int32), ("num_freq_patches", C.c_int32),
        ("audio_ln_eps", C.c_float),
        ("text_vocab", C.c_int32), ("text_hidden", C.c_int32), ("text_layers", C.c_int32), ("text_heads", C.c_int32),
        ("text_intermediate", C.c_int32), ("text_max_pos", C.c_int32), ("text_type_vocab", C.c_int32),
        ("text_ln_eps", C.c_float),
        ("projection_size", C.c_int32), ("pool_heads", C.c_int32), ("logit_scale", C.c_float),
        ("has_audio", C.c_int32), ("has_text", C.c_int32), ("mae_decoder_layers", C.c_int32),
        ("caption_decoder_layers", C.c_int32),
    ]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

_SIGNATURES = {
    "caco_version": (C.c_char_p, []),
    "caco_last_error": (C.c_char_p, []),
    "caco_config_size": (_i32, []),
    "caco_default_config": (None, [C.POINTER(CacoConfigC)]),
    "caco_create": (C.c_int, [C.POINTER(CacoConfigC), C.POINTER(_vp)]),
    "caco_destroy": (None, [_vp]),
    "caco_load_tensor": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32]),
    "caco_finalize_weights": (C.c_int, [_vp]),
    "caco_set_logit_scale": (C.c_int, [_vp, _f32]),
    "caco_get_logit_scale": (_f32, [_vp]),
    "caco_mel_num_frames": (_i64, [_i64]),
    "caco_mel_spectrogram": (C.c_int, [_vp, _i32, _i64, _f32, _f32, _vp, _vp]),
    "caco_mel_patches": (C.c_int, [_vp, _i32, _i64, _i32, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "caco_mel_patches_lens": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "caco_audio_forward": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "caco_text_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "caco_encode_audio": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "caco_encode_audio_ex": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _i32, _vp]),
    "caco_encode_text": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "caco_similarity": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _f32, _vp, _i32, _vp]),
    "caco_similarity_ld": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _f32, _vp, _i32, _vp]),
    "caco_l2_normalize": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "caco_topk": (C.c_int, [_vp, _i32, _i32, _i64, _i64, _i32, _vp, _vp, _vp]),
    "caco_token_group_mean": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "caco_mae_forward": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "caco_workspace_bytes": (_i64, [_vp]),
    "caco_set_gemm_tile": (_i32, [_i32]),
    "caco_set_switch": (C.c_int, [C.c_char_p, _i32]),
    "caco_get_switch": (_i32, [C.c_char_p]),
    "caco_set_ln_fold": (_i32, [_i32]),
    "caco_model_set_ln_fold": (_i32, [_vp, _i32]),
    "caco_profile_enable": (C.c_int, [_i32]),
    "caco_profile_report": (_i64, [C.c_char_p, _i64]),
    "caco_op_gemm_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "caco_op_gemm_bf16_strided": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
    "caco_op_gemm_bf16_f32out": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "caco_op_layernorm": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp]),
    "caco_op_attention": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "caco_op_attention_qkv": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "caco_decode_begin": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "caco_decode_step": (C.c_int, [_vp, _vp, _vp, _vp]),
    "caco_decode_end": (None, [_vp]),
    "caco_decoder_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def declared_symbols(header: str = HEADER_PATH) -> List[str]:
    """Every function name include/caco_hip.h declares."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(caco_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Load the library and bind argument types.  Raises if it is absent: no CPU fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if os.path.realpath(LIB_PATH) != os.path.realpath(PRODUCT_LIB_PATH) and os.environ.get("CACO_ALLOW_VARIANT_LIB") != "1":
        raise RuntimeError(
            f"CACO_LIB_PATH={LIB_PATH} is not the product library {PRODUCT_LIB_PATH}: set CACO_ALLOW_VARIANT_LIB=1 to run an A/B "
            "variant or the simulator build on purpose")
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (build it with `python -m cacophony_amd.build`); "
            "cacophony_amd has no CPU fallback")
    # torch must load ITS bundled HIP runtime (libamdhip64.so.7) first so that this library binds to the same
    # copy by SONAME; loading /opt/rocm's copy first leaves the process with two runtimes and no visible device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    product = os.path.realpath(LIB_PATH) == os.path.realpath(PRODUCT_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)      # AttributeError here = header/library mismatch: fail loudly
        except AttributeError:
            # an older build of the library as an A/B arm (tools/build_r2_arm.sh: commit cccbeef predates the switch API): the
            # missing entry point answers "unknown switch" / CACO_ERR_INVALID.  Never for the product library.
            if product or name not in ("caco_set_switch", "caco_get_switch"):
                raise
            setattr(lib, name, (lambda *a: CACO_ERR_INVALID) if name == "caco_set_switch" else (lambda *a: -2 ** 31))
            continue
        fn.restype = res
        fn.argtypes = args
    # caco_default_config memsets sizeof(caco_config) bytes: a stale struct on either side would overflow
    if lib.caco_config_size() != C.sizeof(CacoConfigC):
        raise RuntimeError(f"caco_config is {lib.caco_config_size()} bytes in {LIB_PATH} but {C.sizeof(CacoConfigC)} in this binding")
    _lib = lib
    return lib


class switch:
    """Context manager for tests / A-B runs: `with switch("CACO_POS_FUSE", 1): ...` sets a run-time switch of the library
    (caco_set_switch) and restores its previous value on exit."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        lib = load()
        self.prev = lib.caco_get_switch(self.name)
        check(lib.caco_set_switch(self.name, self.value), "caco_set_switch")
        return self

    def __exit__(self, *exc):
        # a restore the library refuses would leave this block's value in force for everything that runs later: raise, unless an
        # exception is already on its way out of the block
        status = load().caco_set_switch(self.name, self.prev)
        if exc[0] is None:
            check(status, f"caco_set_switch restore {self.name.decode()}={self.prev}")
        return False


def last_error() -> str:
    return load().caco_last_error().decode("utf-8", "replace")


def check(status: int, what: str = "") -> None:
    if status == CACO_OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if status == CACO_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(f"{msg} (status {status})")
