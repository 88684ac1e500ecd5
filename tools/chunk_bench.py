#!/usr/bin/env python3
"""Does the audio tower run faster when the 256 clips go through it in sub-batches whose activations fit the 256 MiB
Infinity Cache?  Same work, same stream: 1 x 256, 2 x 128, 4 x 64, 8 x 32 clips.   python tools/chunk_bench.py"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from cacophony_amd import config as Cfg, synth
from cacophony_amd.model import create_caco_model
import bench
dev = torch.device("cuda:0")
state = synth.make_caco_state(Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config())
model = create_caco_model(device=dev).load_state_dict(state)
wav, ids, mask = bench._make_inputs(256, 0, dev)
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for chunks in (1, 2, 4, 8):
    c = 256 // chunks
    parts = [wav[i * c:(i + 1) * c].contiguous() for i in range(chunks)]
    def run():
        for p in parts:
            model.encode_audio(p, 500)
    print("audio tower, %d x %3d clips: %.2f ms" % (chunks, c, timeit(run)), flush=True)
