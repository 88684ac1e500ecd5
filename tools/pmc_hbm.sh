#!/bin/bash
# HBM traffic of one GEMM launch from the TCC fabric counters, calibrated on a streaming kernel of known size.
#   bash tools/pmc_hbm.sh <tag> <only> <tile>
# FETCH_SIZE and WRITE_SIZE are collected in separate passes (TCC slot limit); the calibration run measures what the
# counters report for a 1 GiB write / 1 GiB read stream (MI355X_MICROARCH.md: FETCH_SIZE = 1/2 of a wide coalesced
# read on gfx950; WRITE_SIZE uncalibrated), and the GEMM's counters are scaled by those factors.
TAG=${1:-hbm}; ONLY=${2:-fc1}; TILE=${3:-256}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
one() { # name counter cmd...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/raw_$name -o p -- "$@" > $OUT/$name.log 2>&1
  local f=$(find $OUT/raw_$name -name "*counter_collection.csv" | head -1)
  python tools/summarize_pmc.py "$f" > $OUT/$name.csv
  rm -rf $OUT/raw_$name
}
[ -x tools/probes/mall_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/mall_probe.hip -o tools/probes/mall_probe 2>/dev/null
one cal_fetch FETCH_SIZE tools/probes/mall_probe
one cal_write WRITE_SIZE tools/probes/mall_probe
one gemm_fetch FETCH_SIZE python tools/gemm_bench.py --iters 3 --only $ONLY --tile $TILE
one gemm_write WRITE_SIZE python tools/gemm_bench.py --iters 3 --only $ONLY --tile $TILE
python - "$OUT" "$ONLY" "${CACO_W_NGROUP:--1}" <<'PY'
import csv, json, sys
out, only, w_ngroup = sys.argv[1], sys.argv[2], int(sys.argv[3])
def rows(name):
    return list(csv.DictReader(open(f"{out}/{name}.csv")))
def val(name, key, ctr):
    r = [x for x in rows(name) if key in x["kernel"]]
    return float(r[0][ctr]), int(r[0]["dispatches"])
# mall_probe: rd / wr kernels stream sizes 16 MB .. 2 GiB, 12 launches each; mean bytes per launch:
sizes = [16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048]
mean_bytes = sum(sizes) / len(sizes) * (1 << 20)
f_rd, _ = val("cal_fetch", "rd(", "FETCH_SIZE")
w_wr, _ = val("cal_write", "wr(", "WRITE_SIZE")
kf, kw = mean_bytes / f_rd, mean_bytes / w_wr          # bytes per counter unit
gf, n = val("gemm_fetch", "gemm_bf16", "FETCH_SIZE")
gw, _ = val("gemm_write", "gemm_bf16", "WRITE_SIZE")
res = {"shape": only, "fetch_counter": gf, "write_counter": gw, "bytes_per_fetch_unit": kf, "bytes_per_write_unit": kw,
       "hbm_read_bytes": gf * kf, "hbm_write_bytes": gw * kw, "hbm_bytes": gf * kf + gw * kw, "dispatches": n,
       "w_ngroup": w_ngroup}     # the tile order measured (switch CACO_W_NGROUP: -1 = groups by shape, 0 = one group): bench.py only quotes a matching file
json.dump(res, open(f"{out}/hbm_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
