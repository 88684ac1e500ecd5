"""Test-side access to the wavesim build of the kernels (tools/wavesim): the SAME csrc/*.hip sources compiled for x86
against a functional model of the gfx950 constructs, exporting the SAME C ABI (include/caco_hip.h) with host pointers
in place of device pointers.  CPU test infrastructure only - the product package never loads it.

    sim = simlib.load()                       # builds tools/wavesim/libcaco_sim.so on first use (~40 s, cached)
    simlib.check(sim.caco_op_layernorm(...))  # same entry points, same signatures as cacophony_amd._lib
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Mapping, Optional

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from cacophony_amd import _lib  # noqa: E402  (signatures and the config struct only; nothing is loaded from it)

_sim: Optional[C.CDLL] = None


def available() -> bool:
    sys.path.insert(0, os.path.join(REPO, "tools", "wavesim"))
    import build_sim
    return os.path.exists(build_sim.CXX)


def load() -> C.CDLL:
    global _sim
    if _sim is not None:
        return _sim
    sys.path.insert(0, os.path.join(REPO, "tools", "wavesim"))
    import build_sim
    # CACO_SIM_LIB: a sanitizer build of the same library (tools/wavesim/tsan_check.py --pytest preloads the runtime)
    path = os.environ.get("CACO_SIM_LIB") or build_sim.build(verbose=False)
    lib = C.CDLL(path)
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    assert lib.caco_config_size() == C.sizeof(_lib.CacoConfigC)
    _sim = lib
    return lib


SWITCHES = ("CACO_PINGPONG", "CACO_POS_FUSE", "CACO_POOL_FUSE", "CACO_ATTN_SMALL", "CACO_ATTN_ROWS", "CACO_W_NGROUP",
            "CACO_W8_MIN_TILES", "CACO_W4H_MAX_TILES")
_SWITCH_DEFAULTS = {"CACO_ATTN_ROWS": 64, "CACO_W_NGROUP": -1, "CACO_W8_MIN_TILES": 128}


def sync_switches(lib=None) -> None:
    """Push the CACO_* switch variables of os.environ into the library (caco_set_switch): the library reads its environment
    once per switch, so a driver script that flips os.environ between cases calls this after every change.  A variable that
    is absent means the switch's built-in default."""
    lib = lib or load()
    for name in SWITCHES:
        v = os.environ.get(name)
        value = int(v) if v not in (None, "") else _SWITCH_DEFAULTS.get(name, 0)
        check(lib.caco_set_switch(name.encode(), value), f"caco_set_switch {name}={value}", lib)      # a refused value must not pass silently


def check(status: int, what: str = "", lib=None) -> None:
    if status == 0:
        return
    msg = (lib or load()).caco_last_error().decode("utf-8", "replace")
    if status == _lib.CACO_ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (status {status})")


def ptr(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, np.ndarray):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(t.data_ptr())


class SimModel:
    """The construction / weight-loading sequence of cacophony_amd.model._HipModel against the simulator library, plus thin
    wrappers of the forward entry points on torch CPU tensors."""

    def __init__(self, audio_config, text_config, caco_config, mae_decoder_layers: int = 0, caption_decoder_layers: int = 0, lib=None):
        self.lib = lib if lib is not None else load()          # lib: a variant build of the simulator library (tests/test_wavesim.py)
        self.a, self.t, self.c = audio_config, text_config, caco_config
        cfg = _lib.CacoConfigC()
        self.lib.caco_default_config(C.byref(cfg))
        a, t = audio_config, text_config
        cfg.has_audio, cfg.has_text = int(a is not None), int(t is not None)
        if a is not None:
            cfg.audio_hidden, cfg.audio_layers, cfg.audio_heads = a.hidden_size, a.num_layers, a.num_heads
            cfg.audio_intermediate, cfg.patch_size, cfg.num_freq_patches = a.intermediate_size, a.patch_size, a.num_freq_patches
            cfg.audio_ln_eps = a.layer_norm_eps
        if t is not None:
            cfg.text_vocab, cfg.text_hidden, cfg.text_layers = t.vocab_size, t.hidden_size, t.num_hidden_layers
            cfg.text_heads, cfg.text_intermediate = t.num_attention_heads, t.intermediate_size
            cfg.text_max_pos, cfg.text_type_vocab, cfg.text_ln_eps = t.max_position_embeddings, t.type_vocab_size, t.layer_norm_eps
        cfg.projection_size, cfg.pool_heads = caco_config.projection_size, caco_config.num_attention_pool_heads
        cfg.logit_scale = caco_config.logit_scale_init_value
        cfg.mae_decoder_layers = mae_decoder_layers
        cfg.caption_decoder_layers = caption_decoder_layers
        self.cfg = cfg
        self.h = C.c_void_p()
        check(self.lib.caco_create(C.byref(cfg), C.byref(self.h)), "caco_create")

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            self.lib.caco_destroy(h)
            self.h = None

    def load_state_dict(self, state: Mapping[str, object]) -> "SimModel":
        for name, value in state.items():
            arr = value.detach().cpu().numpy() if torch.is_tensor(value) else np.asarray(value)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(self.lib.caco_load_tensor(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim), name)
        check(self.lib.caco_finalize_weights(self.h), "finalize")
        return self

    def set_ln_fold(self, mode: int) -> int:
        return int(self.lib.caco_model_set_ln_fold(self.h, int(mode)))

    def audio_forward(self, patches, tinds, finds, mask, normalize=False):
        patches = torch.as_tensor(patches).contiguous()
        dt = _lib.DTYPE_BF16 if patches.dtype == torch.bfloat16 else _lib.DTYPE_F32
        if dt == _lib.DTYPE_F32:
            patches = patches.float()
        f = lambda x: torch.as_tensor(x).float().contiguous()
        tinds, finds, mask = f(tinds), f(finds), f(mask)
        B, S = mask.shape
        emb = torch.empty(B, self.cfg.projection_size)
        hid = torch.empty(B, S, self.cfg.audio_hidden)
        check(self.lib.caco_audio_forward(self.h, ptr(patches), dt, ptr(tinds), ptr(finds), ptr(mask), B, S, int(normalize),
                                          ptr(emb), ptr(hid), None), "audio_forward")
        return emb, hid

    def text_forward(self, ids, mask, normalize=False, position_ids=None):
        ids = torch.as_tensor(ids).long().contiguous()
        mask = torch.as_tensor(mask).long().contiguous()
        pos = None if position_ids is None else torch.as_tensor(position_ids).long().contiguous()
        B, T = ids.shape
        emb = torch.empty(B, self.cfg.projection_size)
        hid = torch.empty(B, T, self.cfg.text_hidden)
        check(self.lib.caco_text_forward(self.h, ptr(ids), ptr(mask), ptr(pos), B, T, int(normalize), ptr(emb), ptr(hid), None),
              "text_forward")
        return emb, hid

    def encode_audio(self, wav, max_patches=None, lengths=None):
        wav = torch.as_tensor(wav).float().contiguous()
        B, n = wav.shape
        if max_patches is None:
            max_patches = max(8, n * 8 // 160 // 16)    # patches_seq_len rule, eval_caco_torch.py:573,607-612
        lens = None if lengths is None else torch.as_tensor(lengths).long().contiguous()
        emb = torch.empty(B, self.cfg.projection_size)
        check(self.lib.caco_encode_audio_ex(self.h, ptr(wav), ptr(lens), B, n, int(max_patches), ptr(emb), 0, None), "encode_audio")
        return emb

    def encode_text(self, ids, mask):
        ids = torch.as_tensor(ids).long().contiguous()
        mask = torch.as_tensor(mask).long().contiguous()
        B, T = ids.shape
        emb = torch.empty(B, self.cfg.projection_size)
        check(self.lib.caco_encode_text(self.h, ptr(ids), ptr(mask), B, T, ptr(emb), 0, None), "encode_text")
        return emb
