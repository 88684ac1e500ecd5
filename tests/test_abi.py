"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every symbol
include/caco_hip.h declares, the ctypes binding covers the same set, and the product path fails loudly
(no CPU fallback) when asked to compute without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from cacophony_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/caco_hip.h but not exported"
    assert set(names) == set(_lib._SIGNATURES), "ctypes binding and header disagree"
    assert lib.caco_version().startswith(b"cacophony_amd")


def test_default_config_matches_reference_defaults():
    lib = _lib.load()
    cfg = _lib.CacoConfigC()
    lib.caco_default_config(C.byref(cfg))
    # create_caco_model(), src/caco_torch/caco.py:264-317
    assert (cfg.audio_hidden, cfg.audio_layers, cfg.audio_heads, cfg.audio_intermediate) == (768, 12, 8, 3072)
    assert (cfg.patch_size, cfg.num_freq_patches) == (256, 8)
    assert (cfg.text_vocab, cfg.text_hidden, cfg.text_layers, cfg.text_heads, cfg.text_max_pos) == (50265, 768, 12, 12, 514)
    assert (cfg.projection_size, cfg.pool_heads) == (768, 2)
    assert abs(cfg.logit_scale - 2.6592) < 1e-6 and abs(cfg.audio_ln_eps - 1e-5) < 1e-12


def test_host_side_helpers_without_gpu():
    lib = _lib.load()
    assert lib.caco_mel_num_frames(160000) == 1000          # eval_caco_torch.py:66-72
    assert lib.caco_mel_num_frames(12345) == 78
    assert lib.caco_set_gemm_tile(0) in (128, 256)


def test_argument_errors_are_status_codes_not_crashes():
    lib = _lib.load()
    assert lib.caco_create(None, None) == _lib.CACO_ERR_INVALID
    assert b"null" in lib.caco_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.caco_similarity(None, 1, None, 1, 768, 1.0, None, 1, None), "similarity")


@pytest.mark.skipif(torch.cuda.is_available(), reason="exercises the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from cacophony_amd import frontend
    from cacophony_amd.model import create_caco_model
    with pytest.raises(RuntimeError, match="no CPU path"):
        create_caco_model()
    with pytest.raises(RuntimeError, match="no CPU path"):
        frontend.compute_mel_spectrogram(np.zeros(16000, np.float32))


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cacophony_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f"{f} reaches into oracle/"


def test_spectrogram_to_patches_host_mirror_matches_oracle():
    from cacophony_amd.frontend import spectrogram_to_patches
    from oracle import caco_oracle as O
    rng = np.random.default_rng(0)
    for frames, max_p in ((1000, 500), (300, 500), (300, 100), (43, 16), (15, 8)):
        spec = rng.standard_normal((frames, 128)).astype(np.float32)
        a, b = spectrogram_to_patches(spec, 16, 16, max_p), O.spectrogram_to_patches(spec, 16, 16, max_p)
        for k in b:
            np.testing.assert_array_equal(a[k], b[k])


def test_integration_doc_struct_matches_header_and_binding():
    """INTEGRATION.md section 2 shows the ctypes struct a maintainer would copy: its field list must be the header's
    (round 1 shipped it one field short: caco_default_config memsets sizeof(caco_config) -> 4-byte overflow)."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "caco_hip.h")).read()
    body = re.search(r"typedef struct caco_config \{(.*?)\} caco_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    header_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        header_fields += [(n.strip(), typ) for n in names.split(",")]
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    snippet = doc[doc.index("class caco_config(C.Structure):"):doc.index("lib.caco_config_size.restype")]
    ns = {"C": C}
    exec(snippet, ns)
    doc_fields = [(n, "float" if t is C.c_float else "int32_t") for n, t in ns["caco_config"]._fields_]
    assert doc_fields == header_fields
    assert [(n, "float" if t is C.c_float else "int32_t") for n, t in _lib.CacoConfigC._fields_] == header_fields
    lib = _lib.load()
    assert lib.caco_config_size() == C.sizeof(ns["caco_config"]) == C.sizeof(_lib.CacoConfigC) == 4 * len(header_fields)


def test_run_time_switches_read_their_environment_once_then_only_the_api(tmp_path):
    """caco_set_switch / caco_get_switch (host-only, no GPU needed): defaults, round trip, the context manager restores,
    unknown names are status codes, and the environment is the INITIAL value only - a later change of the variable is not
    seen (no launch path calls getenv), a later caco_set_switch is."""
    import subprocess
    import sys
    lib = _lib.load()
    assert lib.caco_get_switch(b"CACO_NO_SUCH_SWITCH") == -2 ** 31
    assert lib.caco_set_switch(b"CACO_NO_SUCH_SWITCH", 1) == _lib.CACO_ERR_INVALID and b"unknown switch" in lib.caco_last_error()
    assert lib.caco_set_switch(None, 1) == _lib.CACO_ERR_INVALID
    prev = lib.caco_get_switch(b"CACO_POS_FUSE")
    with _lib.switch("CACO_POS_FUSE", 1 - prev):
        assert lib.caco_get_switch(b"CACO_POS_FUSE") == 1 - prev
    assert lib.caco_get_switch(b"CACO_POS_FUSE") == prev
    # values are validated per switch; the "environment not read yet" sentinel can never be stored (it would re-arm getenv)
    for name, bad in ((b"CACO_POS_FUSE", -2 ** 31), (b"CACO_W_NGROUP", -2 ** 31), (b"CACO_W_NGROUP", -2), (b"CACO_ATTN_ROWS", 48),
                      (b"CACO_W8_MIN_TILES", -1), (b"CACO_PINGPONG", 2)):
        before = lib.caco_get_switch(name)
        assert lib.caco_set_switch(name, bad) == _lib.CACO_ERR_INVALID and b"out of range" in lib.caco_last_error(), (name, bad)
        assert lib.caco_get_switch(name) == before
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "from cacophony_amd import _lib\n"
        "lib = _lib.load()\n"
        "g = lambda n: lib.caco_get_switch(n.encode())\n"
        "first = [g(n) for n in ('CACO_PINGPONG', 'CACO_W_NGROUP', 'CACO_ATTN_ROWS', 'CACO_W8_MIN_TILES', 'CACO_POOL_FUSE')]\n"
        "os.environ['CACO_PINGPONG'] = '0'; os.environ['CACO_POOL_FUSE'] = '1'\n"      # after first use: must not be seen
        "later = [g('CACO_PINGPONG'), g('CACO_POOL_FUSE')]\n"
        "lib.caco_set_switch(b'CACO_PINGPONG', 0)\n"
        "print(first, later, g('CACO_PINGPONG'))\n" % root)
    env = dict(os.environ, CACO_PINGPONG="1", CACO_W_NGROUP="2")
    for k in ("CACO_ATTN_ROWS", "CACO_W8_MIN_TILES", "CACO_POOL_FUSE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "[1, 2, 64, 128, 0] [1, 0] 0"
    # the environment goes through the same per-switch ranges as caco_set_switch: an out-of-range initial value is replaced by the
    # default (with a line on stderr) instead of reaching a launch path, and the context manager can then restore what it found
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "from cacophony_amd import _lib\n"
        "lib = _lib.load()\n"
        "g = lambda n: lib.caco_get_switch(n.encode())\n"
        "first = [g(n) for n in ('CACO_PINGPONG', 'CACO_ATTN_ROWS', 'CACO_W8_MIN_TILES', 'CACO_W_NGROUP', 'CACO_POS_FUSE')]\n"
        "with _lib.switch('CACO_PINGPONG', 1):\n"
        "    inside = g('CACO_PINGPONG')\n"
        "print(first, inside, g('CACO_PINGPONG'))\n" % root)
    env = dict(os.environ, CACO_PINGPONG="2", CACO_ATTN_ROWS="48", CACO_W8_MIN_TILES="-5", CACO_W_NGROUP="5000", CACO_POS_FUSE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "[0, 64, 128, -1, 1] 1 0"
    for name in ("CACO_PINGPONG=2", "CACO_ATTN_ROWS=48", "CACO_W8_MIN_TILES=-5", "CACO_W_NGROUP=5000"):
        assert f"{name} is out of range" in out.stderr, out.stderr[-1000:]
    assert "CACO_POS_FUSE" not in out.stderr


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """include/caco_hip.h compiled as C99 (-pedantic, warnings as errors) and linked against the library from a C program that
    calls the host-only entry points: the boundary is a C ABI, not a C++ one."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include "caco_hip.h"\n'
        "int main(void) {\n"
        "  caco_config cfg;\n"
        "  caco_default_config(&cfg);\n"
        '  printf("%d %d %d %d %d\\n", (int)caco_config_size(), (int)sizeof(cfg), (int)cfg.audio_hidden, (int)cfg.text_vocab,\n'
        '         (int)caco_get_switch("CACO_ATTN_ROWS"));\n'
        "  if (caco_create(NULL, NULL) == 0) return 2;            /* a null config is a status code, not a crash */\n"
        '  printf("%s\\n", caco_last_error());\n'
        "  return caco_mel_num_frames(160000) == 1000 ? 0 : 3;\n"
        "}\n")
    exe = tmp_path / "abi"
    libdir = os.path.join(root, "cacophony_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        os.path.join(libdir, "libcaco_hip.so"), f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ)
    env.pop("CACO_ATTN_ROWS", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert r.stdout.splitlines()[0] == "88 88 768 50265 64" and "null" in r.stdout.splitlines()[1]
