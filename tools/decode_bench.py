#!/usr/bin/env python3
"""Caption decoding throughput at the full configuration (12 + 12 layers, 4-layer decoder, vocabulary 50265):
key / value-cached steps (caco_decode_step) vs the reference-style full-prefix recompute (get_decoder_logits).
   python tools/decode_bench.py [--batch 256] [--steps 32]"""
import argparse, os, sys, time
from dataclasses import replace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cacophony_amd import captioning, config as Cfg, frontend, synth
from cacophony_amd.model import CACO

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=32)
    a = ap.parse_args()
    ac, tc, cc = Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config()
    dc = replace(tc, num_hidden_layers=4)
    model = CACO(ac, tc, cc, decoder_config=dc, device="cuda:0").load_state_dict(synth.make_caco_state(ac, tc, cc, decoder_cfg=dc))
    wav = torch.from_numpy(synth.make_waveforms(a.batch)).cuda()
    ab = frontend.mel_patches_device(wav, 500, torch.bfloat16)
    _, ah = model.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"])
    tok = torch.zeros(a.batch, dtype=torch.long, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = captioning.CaptionDecodeState(model, ah, ab["audio_mask"], a.steps + 1)
    torch.cuda.synchronize(); t_begin = time.perf_counter() - t0
    st.step(tok); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        lg = st.step(tok)
        tok = lg.argmax(-1)
    torch.cuda.synchronize(); t_c = (time.perf_counter() - t0) / a.steps
    st.close()
    print(f"cached : begin {t_begin*1e3:.1f} ms, {t_c*1e3:.2f} ms per step of {a.batch} clips = {a.batch / t_c:.0f} tokens/s")
    ids = torch.zeros(a.batch, 1, dtype=torch.long, device="cuda")
    n = min(a.steps, 16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        lg = model.get_decoder_logits(ah, ab["audio_mask"], ids, torch.ones_like(ids))[:, -1]
        ids = torch.cat([ids, lg.argmax(-1)[:, None]], 1)
    torch.cuda.synchronize(); t_u = (time.perf_counter() - t0) / n
    print(f"prefix : {t_u*1e3:.2f} ms per step (mean over the first {n} positions) = {a.batch / t_u:.0f} tokens/s")

if __name__ == "__main__":
    main()
