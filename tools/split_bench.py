import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from cacophony_amd import config as Cfg, synth
from cacophony_amd.model import create_caco_model, similarity
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
print("audio only      %.2f ms" % timeit(lambda: model.encode_audio(wav, 500)))
print("text only       %.2f ms" % timeit(lambda: model.encode_text(ids, mask)))
print("pairs (2 strm)  %.2f ms" % timeit(lambda: model.encode_pairs(wav, ids, mask, 500)))
def serial():
    model.encode_audio(wav, 500); model.encode_text(ids, mask)
print("serial a+t      %.2f ms" % timeit(serial))
