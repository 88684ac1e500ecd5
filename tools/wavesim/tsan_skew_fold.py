import os, subprocess, sys, re
import os; HERE=os.path.dirname(os.path.abspath(__file__)); REPO=os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import build_sim
lib = build_sim.build(tsan=True, defines=("-DW8_F32_SKEW",), tag="skew", verbose=False)
rt = subprocess.run([build_sim.CXX, "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
code = r'''
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, REPO)
from dataclasses import replace
from tests import simlib
from cacophony_amd import _lib, config as Cf, synth
lib = C.CDLL(LIB)
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
a, t, cc = Cf.tiny_configs(2); a = replace(a, num_layers=1)
state = synth.make_caco_state(*Cf.tiny_configs(2))
wav = torch.from_numpy(synth.make_waveforms(8, start=40))
lib.caco_set_switch(b"CACO_W8_MIN_TILES", 1)
for fold in (1,):
    m = simlib.SimModel(a, None, cc, lib=lib).load_state_dict({k: v for k, v in state.items() if k.startswith(("audio_", "logit_scale")) and ".layers.1." not in k})
    m.set_ln_fold(fold)
    e = m.encode_audio(wav).numpy()
    print("fold", fold, "finite", bool(np.isfinite(e).all()), float(np.abs(e).max()))
print("DRIVER DONE")
'''
env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4", WAVESIM_THREADS="3", OMP_NUM_THREADS="1")
r = subprocess.run([sys.executable, "-c", f"REPO = {REPO!r}\nLIB = {lib!r}\n" + code], env=env, capture_output=True, text=True)
print(r.stdout[-600:]); print("exit", r.returncode)
warn = re.findall(r"WARNING: ThreadSanitizer: ([^\n(]+)", r.stderr)
from collections import Counter
print(Counter(warn))
i = r.stderr.find("WARNING: ThreadSanitizer")
print(r.stderr[i:i+2500] if i >= 0 else r.stderr[-1500:])
