#!/usr/bin/env python3
"""Interleaved A/B of the library's run-time switches inside ONE process (GPU box).

    python tools/ab_switches.py [--reps 5] [--steps 10] [--out gpurun_out/r4_v0/ab_switches.json] [--batch 256] [--layers 12]

The model, the inputs and the bench step (bench.make_step: both towers + similarity, text tower on its side stream) are built
once.  Each repetition walks the configurations in a rotated order - default, every switch alone, the combinations listed
below - and times `steps` steps of each between two device synchronisations, so that every configuration sees the same box,
the same clock / thermal state and the same neighbours.  Per-stage times of one single-stream pass per configuration come from
the library's own HIP-event recorder (caco_profile_*).

Output: one table (mean / min / max ms per step, delta against the default measured in the same repetition) and the decision
by the rule of DESIGN.md section 9: FLIP when the configuration beats the default in EVERY repetition by more than --margin
ms (default 0.3 = the box-to-box spread of one commit), DELETE when it loses in every repetition by that margin, KEEP-OFF
otherwise.  The same run re-checks that every configuration's similarity matrix agrees with the default's (max |diff|).

Replaces the bench.py-per-switch loop of tools/gpu_session.sh `ab` (a fresh process, model build and warm-up per switch:
~40 s each, not interleaved).  Needs caco_set_switch (round 4): no environment variable is involved after start-up.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

# name -> {switch: value}; "ln_fold" is the per-model LayerNorm-fold mode (caco_model_set_ln_fold), not a switch
CONFIGS = [
    ("default", {}),
    ("ngroup_one", {"CACO_W_NGROUP": 0}),
    ("pingpong", {"CACO_PINGPONG": 1}),
    ("pos_fuse", {"CACO_POS_FUSE": 1}),
    ("pool_fuse", {"CACO_POOL_FUSE": 1}),
    ("attn_small", {"CACO_ATTN_SMALL": 1}),
    ("text_w4h", {"CACO_W4H_MAX_TILES": 128}),
    ("text_n768_128", {"CACO_W8_MIN_TILES": 200}),
    ("attn_rows32", {"CACO_ATTN_ROWS": 32}),
    ("ln_fold", {"ln_fold": 1}),
    ("fusions", {"CACO_POS_FUSE": 1, "CACO_POOL_FUSE": 1, "CACO_ATTN_SMALL": 1}),
    ("fusions+pingpong", {"CACO_POS_FUSE": 1, "CACO_POOL_FUSE": 1, "CACO_ATTN_SMALL": 1, "CACO_PINGPONG": 1}),
]
STAGES = ("audio.gemm_fc1", "audio.gemm_qkv", "audio.gemm_fc2", "audio.gemm_out", "audio.attention", "audio.ln", "audio.pos_embed",
          "audio.patch_embed", "audio.pool", "text.attention", "text.gemm_out", "text.gemm_fc2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.3, help="ms a configuration must win / lose by in every repetition")
    ap.add_argument("--batch", type=int, default=bench.B_PER_GPU)
    ap.add_argument("--layers", type=int, default=0, help="0 = the full 12 + 12-layer model; n = an n-layer model (simulator dry runs)")
    ap.add_argument("--only", default="", help="comma-separated configuration names (default: all)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    from cacophony_amd import _lib, config as Cfg, synth
    from cacophony_amd.dist import gather_packed
    from cacophony_amd.model import CACO, similarity

    lib = _lib.load()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    if args.layers:
        a, t, cc = Cfg.tiny_configs(args.layers)
    else:
        a, t, cc = Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config()
    model = CACO(a, t, cc, device=device).load_state_dict(synth.make_caco_state(a, t, cc))
    wav, ids, mask = bench._make_inputs(args.batch, 0, device)
    sim_out = torch.empty(args.batch, args.batch, dtype=torch.float32, device=device)
    step = bench.make_step(model, wav, ids, mask, sim_out, similarity, gather_packed)
    sync = torch.cuda.synchronize

    names = [n for n, _ in CONFIGS]
    only = [x for x in args.only.split(",") if x]
    configs = [(n, c) for n, c in CONFIGS if not only or n in only or n == "default"]
    defaults = {n: int(lib.caco_get_switch(n.encode())) for n in bench.SWITCH_NAMES}
    fold0 = int(lib.caco_model_set_ln_fold(model._handle, -99))

    def apply(cfg):
        for n, v in defaults.items():
            _lib.check(lib.caco_set_switch(n.encode(), int(cfg.get(n, v))), "caco_set_switch")
        lib.caco_model_set_ln_fold(model._handle, int(cfg.get("ln_fold", fold0)))

    def timed(n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        sync()
        return (time.perf_counter() - t0) / n * 1e3

    # warm-up of every configuration (arena growth, kernel attributes), reference output, per-stage profile
    ref, checks, stages = None, {}, {}
    for name, cfg in configs:
        apply(cfg)
        timed(args.warmup)
        out = sim_out.clone()
        if ref is None:
            ref = out
        checks[name] = {"finite": bool(torch.isfinite(out).all().item()), "max_abs_diff_vs_default": float((out - ref).abs().max().item())}
        lib.caco_profile_enable(1)
        ea = model.encode_audio(wav, bench.SEQ)
        et = model.encode_text(ids, mask, check_ids=False)
        similarity(ea, et, 1.0, out=sim_out)
        sync()
        buf = C.create_string_buffer(1 << 16)
        lib.caco_profile_report(buf, len(buf))
        lib.caco_profile_enable(0)
        prof = json.loads(buf.value.decode())
        stages[name] = {k: round(prof[k]["ms"], 4) for k in STAGES if k in prof}
        stages[name]["sum_all_stages"] = round(sum(v["ms"] for v in prof.values()), 3)

    times = {n: [] for n, _ in configs}
    for rep in range(args.reps):
        order = configs[rep % len(configs):] + configs[:rep % len(configs)]      # rotated: nobody always follows the same neighbour
        for name, cfg in order:
            apply(cfg)
            timed(1)
            times[name].append(timed(args.steps))
    apply({})

    base = np.array(times["default"])
    rows = []
    for name, cfg in configs:
        tms = np.array(times[name])
        d = tms - base
        if name == "default":
            verdict = "-"
        elif not checks[name]["finite"] or checks[name]["max_abs_diff_vs_default"] > 2e-3:
            verdict = "BROKEN (output differs)"
        elif (d < -args.margin).all():
            verdict = "FLIP"
        elif (d > args.margin).all():
            verdict = "DELETE"
        else:
            verdict = "KEEP-OFF (inside the margin)"
        rows.append({"config": name, "switches": cfg, "ms_mean": round(float(tms.mean()), 3), "ms_min": round(float(tms.min()), 3),
                     "ms_max": round(float(tms.max()), 3), "delta_mean": round(float(d.mean()), 3), "delta_min": round(float(d.min()), 3),
                     "delta_max": round(float(d.max()), 3), "verdict": verdict, **checks[name], "stages_ms": stages[name],
                     # a fusion that removes one small launch (pos_fuse: <= 0.14 ms, pool_fuse: <= 0.10 ms) cannot clear the step margin by
                     # construction: its evidence is the single-stream stage sum (one pass, HIP events), reported next to the verdict
                     "delta_stage_sum": round(stages[name]["sum_all_stages"] - stages["default"]["sum_all_stages"], 3)})
    print(f"{'config':<18} {'ms mean':>8} {'min':>8} {'max':>8} {'d mean':>8} {'d min':>8} {'d max':>8} {'d stages':>9}  {'|dsim|':>8}  verdict")
    for r in rows:
        print(f"{r['config']:<18} {r['ms_mean']:8.3f} {r['ms_min']:8.3f} {r['ms_max']:8.3f} {r['delta_mean']:+8.3f} {r['delta_min']:+8.3f} "
              f"{r['delta_max']:+8.3f} {r['delta_stage_sum']:+9.3f}  {r['max_abs_diff_vs_default']:8.1e}  {r['verdict']}")
    print("\nper-stage ms of one single-stream pass (library HIP events):")
    keys = [k for k in STAGES if any(k in stages[n] for n, _ in configs)] + ["sum_all_stages"]
    print(f"{'config':<18} " + " ".join(f"{(k[0] + '.' + k.split('.')[-1].replace('gemm_', ''))[:9]:>9}" for k in keys))
    for name, _ in configs:
        print(f"{name:<18} " + " ".join(f"{stages[name].get(k, float('nan')):9.3f}" for k in keys))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump({"reps": args.reps, "steps": args.steps, "batch": args.batch, "margin_ms": args.margin, "rows": rows,
                   "times_ms": times, "known_configs": names}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
