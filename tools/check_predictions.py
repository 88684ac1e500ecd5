#!/usr/bin/env python3
"""Predictions stated before any measurement (profiles/r4_cpu/epilogue_budget.txt, csrc/gemm_w8_skew.inc, tools/build_variants.sh)
against what a GPU session measured.  Reads tools/ab_variants.py's JSON (per-stage times of every library in one process) and,
optionally, the truth session's bench line; prints one row per prediction with HELD / MISSED and the measured per-launch time.

    python tools/check_predictions.py gpurun_out/r6_v0/ab_variants.json [gpurun_out/r6_v0/bench.json]

No GPU needed (it only reads the records); exits 0 whatever the outcome - it is a report, not a test."""
import json
import sys

LAUNCHES = 12          # audio layers: every audio.* GEMM / attention stage is 12 launches per pass
# (variant, stage, what was predicted, lower bound ms per launch, upper bound ms per launch, source)
PREDICTIONS = [
    ("default", "audio.gemm_qkv", "round 4: peeled first K-tile + packed bias epilogue, 0.38 -> ~0.33 ms", 0.0, 0.345, "profiles/r4_cpu/epilogue_budget.txt"),
    ("default", "audio.gemm_fc1", "round 4: peeled first K-tile, 0.585 -> ~0.545 ms", 0.0, 0.56, "profiles/r4_cpu/epilogue_budget.txt"),
    ("skew", "audio.gemm_out", "round 5: epilogue under the K-loop, circular panel list, 0.23 -> <= 0.20 ms", 0.0, 0.20, "csrc/gemm_w8_skew.inc"),
    ("skew", "audio.gemm_fc2", "round 5: epilogue under the K-loop, circular panel list, 0.54 -> <= 0.41 ms", 0.0, 0.41, "csrc/gemm_w8_skew.inc"),
    ("skew_lin", "audio.gemm_out", "round 5: linear panel list (8 extra K-tiles per launch), 0.23 -> <= 0.21 ms", 0.0, 0.21, "csrc/gemm_w8_skew.inc"),
    ("skew_lin", "audio.gemm_fc2", "round 5: linear panel list (43 extra K-tiles per launch), 0.54 -> <= 0.46 ms", 0.0, 0.46, "csrc/gemm_w8_skew.inc"),
    ("attn_lean", "audio.attention", "round 5: no spill, epilogue on register pairs, 0.264 -> 0.255-0.262 ms", 0.0, 0.262, "tools/build_variants.sh"),
    ("attn_lean_k2", "audio.attention", "round 5: attn_lean + K fragment reads pinned 2 steps ahead, 0.264 -> 0.245-0.262 ms", 0.0, 0.262, "tools/build_variants.sh"),
]
# step-level predictions: (variant, delta vs default in ms: lower, upper, text)
STEP = [
    ("classic", +0.2, +1.1, "round 4: the default (peeled / packed) beats `classic` by 0.2 .. 1.1 ms per step"),
    ("skew", -2.2, -1.8, "round 5: skew (circular) beats the default by 1.8 .. 2.2 ms per step (if the power cap returns cycles as time)"),
    ("skew_lin", -1.3, -1.0, "round 5: skew_lin beats the default by 1.0 .. 1.3 ms per step"),
    ("default+fold", -0.3, +0.8, "round 2 measured: the LayerNorm-folded stack on the serialized epilogues is a wash (-2.2 ms of LN passes, +2.7 ms of epilogues)"),
    ("skew+fold", -3.7, -2.3, "round 5: with the fold producers under the K-loop the folded stack wins another 0.5 .. 1.5 ms over skew alone"),
    ("wb", -0.3, +0.3, "round 6: default cache policy on every epilogue store / residual load of the persistent GEMM: inside the box's spread"),
    ("hints", -0.3, +0.3, "round 6: nt on the GEMM operands, attention and LayerNorm output stores: inside the box's spread (W with nt may lose)"),
]


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 0
    ab = json.load(open(sys.argv[1]))
    rows = {r["variant"]: r for r in ab["rows"]}
    print(f"{'variant':<10} {'stage':<16} {'per launch':>10} {'default':>9}  outcome  prediction")
    for var, stage, text, lo, hi, src in PREDICTIONS:
        if var not in rows or stage not in rows[var]["stages_ms"]:
            print(f"{var:<10} {stage:<16} {'-':>10} {'-':>9}  NOT RUN  {text}")
            continue
        ms = rows[var]["stages_ms"][stage] / LAUNCHES
        base = rows["default"]["stages_ms"].get(stage, float("nan")) / LAUNCHES
        ok = lo <= ms <= hi
        print(f"{var:<10} {stage:<16} {ms:10.4f} {base:9.4f}  {'HELD   ' if ok else 'MISSED '}  {text}  [{src}]")
    print()
    for var, lo, hi, text in STEP:
        if var not in rows:
            print(f"{var:<10} step delta      -  NOT RUN  {text}")
            continue
        d = rows[var]["delta_mean"]
        print(f"{var:<10} step delta {d:+7.3f} ms ({rows[var]['verdict']})  {'HELD   ' if lo <= d <= hi else 'MISSED '}  {text}")
    if len(sys.argv) > 2:
        b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        r = b.get("roofline") or {}
        print(f"\nbench line: {b.get('ms_per_step')} ms/step, {b.get('value')} {b.get('unit')}; fc1 {r.get('avg_launch_ms')} ms = frac {r.get('frac')} "
              f"(targets of the round-5 verdict: step <= 26.0 ms, fc1 frac >= 0.45); library {b.get('config', {}).get('lib_path')}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
