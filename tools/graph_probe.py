import sys, time, torch
sys.path.insert(0, "/root/repo")
from cacophony_amd import config as Cfg, synth
from cacophony_amd.model import create_caco_model, similarity
import bench
dev = torch.device("cuda:0")
state = synth.make_caco_state(Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config())
model = create_caco_model(device=dev).load_state_dict(state)
wav, ids, mask = bench._make_inputs(256, 0, dev)
sim = torch.empty(256, 256, device=dev)
def step():
    ea, et = model.encode_pairs(wav, ids, mask, 500)
    return similarity(ea, et, 1.0, out=sim)
for _ in range(3): step()
torch.cuda.synchronize()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=s):
        out = step()
    torch.cuda.synchronize()
    ref = sim.clone()
    sim.zero_()
    g.replay(); torch.cuda.synchronize()
    print("graph replay equal:", torch.equal(sim, ref))
    print("graph ms/step", timeit(g.replay))
    print("eager ms/step", timeit(step))
except Exception as e:
    print("capture failed:", repr(e)[:500])
