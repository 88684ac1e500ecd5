#!/usr/bin/env python3
"""Interleaved A/B of compile-time variant libraries inside ONE process (GPU box).

    python tools/ab_variants.py [--reps 5] [--steps 10] [--out gpurun_out/r6_v0/ab_variants.json] default skew classic ...

Every name is `default` (cacophony_amd/libcaco_hip.so) or a library under cacophony_amd/_variants/libcaco_hip_<name>.so
(tools/build_variants.sh); `<name>+fold` runs that library's model in the LayerNorm-folded form (caco_model_set_ln_fold).  All of them are dlopen'ed side by side - each has its own process-global state, they share one HIP
runtime - and one model per library is created from the SAME seeded state dict; the bench step (bench.make_step) of each is timed
in a rotated order, `steps` steps per visit, so that every variant sees the same box, clock / thermal state and neighbours.
Per-stage times come from each library's own HIP-event recorder.  Verdict as in tools/ab_switches.py: WIN when the variant beats the
default in EVERY repetition by more than --margin ms, LOSE when it loses in every repetition by that margin, else inside the margin.
Each variant's similarity matrix is compared with the default's.

(Round 1-3 ran one bench.py process per variant and repetition: ~35 s each, not interleaved.)  The product never
loads a variant library; this tool is the only place several are in one process.
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

STAGES = ("audio.gemm_fc1", "audio.gemm_qkv", "audio.gemm_fc2", "audio.gemm_out", "audio.attention", "audio.ln", "audio.patch_embed",
          "text.gemm_fc1", "text.gemm_fc2", "text.attention")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["default"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=bench.B_PER_GPU)
    ap.add_argument("--layers", type=int, default=0, help="0 = the full 12 + 12-layer model; n = an n-layer model (dry runs)")
    ap.add_argument("--lib-pattern", default=os.path.join(REPO, "cacophony_amd", "_variants", "libcaco_hip_{name}.so"),
                    help="where a variant's library lives (simulator dry run: tools/wavesim/libcaco_sim_{name}.so)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    names = list(dict.fromkeys(["default"] + [n for n in args.names if n != "default"]))

    from cacophony_amd import _lib, config as Cfg, synth
    from cacophony_amd.dist import gather_packed
    from cacophony_amd.model import CACO, similarity

    default_lib = _lib.load()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    if args.layers:
        a, t, cc = Cfg.tiny_configs(args.layers)
    else:
        a, t, cc = Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config()
    state = synth.make_caco_state(a, t, cc)
    wav, ids, mask = bench._make_inputs(args.batch, 0, device)
    sync = torch.cuda.synchronize

    libs, models, steps, outs = {}, {}, {}, {}
    for name in names:
        base, _, opt = name.partition("+")          # "skew+fold": library `skew`, its model in the LayerNorm-folded form
        if base == "default":
            lib = default_lib
        else:
            path = args.lib_pattern.format(name=base)
            if not os.path.exists(path):
                print(f"skip {name}: {path} not built (tools/build_variants.sh)")
                continue
            lib = C.CDLL(path)
            for fn, (res, argt) in _lib._SIGNATURES.items():
                try:
                    f = getattr(lib, fn)
                except AttributeError:          # the round-2 arm (tools/build_r2_arm.sh) predates the switch API
                    if fn not in ("caco_set_switch", "caco_get_switch"):
                        raise
                    continue
                f.restype, f.argtypes = res, argt
        # a CACO object keeps the library it was created with (model.py: self._lib = _lib.load()): swap the binding's singleton
        # around the construction only
        saved, _lib._lib = _lib._lib, lib
        try:
            models[name] = CACO(a, t, cc, device=device).load_state_dict(state)
        finally:
            _lib._lib = saved
        if opt == "fold":
            assert int(lib.caco_model_set_ln_fold(models[name]._handle, 1)) == 1
        elif opt:
            raise SystemExit(f"unknown option in variant name {name!r} (only +fold)")
        libs[name] = lib
        outs[name] = torch.empty(args.batch, args.batch, dtype=torch.float32, device=device)
        steps[name] = bench.make_step(models[name], wav, ids, mask, outs[name], similarity, gather_packed)
    names = [n for n in names if n in libs]

    def timed(name, n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            steps[name]()
        sync()
        return (time.perf_counter() - t0) / n * 1e3

    checks, stages = {}, {}
    for name in names:
        timed(name, args.warmup)
        checks[name] = {"finite": bool(torch.isfinite(outs[name]).all().item()),
                        "max_abs_diff_vs_default": float((outs[name] - outs["default"]).abs().max().item())}
        lib, m = libs[name], models[name]
        lib.caco_profile_enable(1)
        ea = m.encode_audio(wav, bench.SEQ)
        et = m.encode_text(ids, mask, check_ids=False)
        similarity(ea, et, 1.0, out=outs[name])
        sync()
        buf = C.create_string_buffer(1 << 16)
        lib.caco_profile_report(buf, len(buf))
        lib.caco_profile_enable(0)
        prof = json.loads(buf.value.decode())
        stages[name] = {k: round(prof[k]["ms"], 4) for k in STAGES if k in prof}
        stages[name]["sum_all_stages"] = round(sum(v["ms"] for v in prof.values()), 3)

    times = {n: [] for n in names}
    for rep in range(args.reps):
        order = names[rep % len(names):] + names[:rep % len(names)]
        for name in order:
            timed(name, 1)
            times[name].append(timed(name, args.steps))

    base = np.array(times["default"])
    rows = []
    for name in names:
        tms = np.array(times[name])
        d = tms - base
        if name == "default":
            verdict = "-"
        elif not checks[name]["finite"] or checks[name]["max_abs_diff_vs_default"] > 2e-3:
            verdict = "BROKEN (output differs)"
        elif (d < -args.margin).all():
            verdict = "WIN"
        elif (d > args.margin).all():
            verdict = "LOSE"
        else:
            verdict = "inside the margin"
        rows.append({"variant": name, "ms_mean": round(float(tms.mean()), 3), "ms_min": round(float(tms.min()), 3), "ms_max": round(float(tms.max()), 3),
                     "delta_mean": round(float(d.mean()), 3), "delta_min": round(float(d.min()), 3), "delta_max": round(float(d.max()), 3),
                     "verdict": verdict, **checks[name], "stages_ms": stages[name]})
    print(f"{'variant':<18} {'ms mean':>8} {'min':>8} {'max':>8} {'d mean':>8} {'d min':>8} {'d max':>8}  {'|dsim|':>8}  verdict")
    for r in rows:
        print(f"{r['variant']:<18} {r['ms_mean']:8.3f} {r['ms_min']:8.3f} {r['ms_max']:8.3f} {r['delta_mean']:+8.3f} {r['delta_min']:+8.3f} "
              f"{r['delta_max']:+8.3f}  {r['max_abs_diff_vs_default']:8.1e}  {r['verdict']}")
    print("\nper-stage ms of one single-stream pass (each library's own HIP events):")
    keys = [k for k in STAGES if any(k in stages[n] for n in names)] + ["sum_all_stages"]
    print(f"{'variant':<18} " + " ".join(f"{(k[0] + '.' + k.split('.')[-1].replace('gemm_', ''))[:9]:>9}" for k in keys))
    for name in names:
        print(f"{name:<18} " + " ".join(f"{stages[name].get(k, float('nan')):9.3f}" for k in keys))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump({"reps": args.reps, "steps": args.steps, "batch": args.batch, "margin_ms": args.margin, "rows": rows, "times_ms": times},
                  open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
