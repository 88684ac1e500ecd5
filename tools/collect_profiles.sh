#!/bin/bash
# Everything profiles/<tag>/ holds, in one GPU-box call:  bash tools/collect_profiles.sh r2_v3
TAG=${1:-r2_v3}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err
bash tools/profile_bench.sh $TAG --no-extra-configs > /dev/null 2>&1
cp gpurun_out/prof_$TAG/bench.json $O/bench_under_rocprof.json; cp gpurun_out/prof_$TAG/kernel_stats_summary.csv $O/
bash tools/pmc_run.sh ${TAG}_fc1 gemm_bf16_w8 -- python tools/gemm_bench.py --iters 3 --only fc1 > /dev/null 2>&1
for f in sq1 sq2 sq3 tcc1 tcc2; do cat gpurun_out/${TAG}_fc1/$f.csv gpurun_out/${TAG}_fc1/$f.dur > $O/pmc_fc1_$f.csv; done
bash tools/pmc_run.sh ${TAG}_blaslt Cijk -- python tools/blaslt_gemm.py --only fc1 > /dev/null 2>&1
for f in sq1 sq2 sq3 tcc1 tcc2; do cat gpurun_out/${TAG}_blaslt/$f.csv gpurun_out/${TAG}_blaslt/$f.dur > $O/pmc_hipblaslt_fc1_$f.csv; done
bash tools/pmc_hbm.sh ${TAG}_hbm fc1 256 > /dev/null 2>&1; cp gpurun_out/${TAG}_hbm/hbm_traffic.json $O/
{ echo "# tools/power_probe.py on one MI355X (socket power cap 1400 W): same instruction stream on random vs zero operands";
  python tools/power_probe.py --only fc1,qkv,fc2 --seconds 3; python tools/power_probe.py --stages --seconds 3; python tools/power_probe.py --attn --seconds 3; } 2>&1 | grep -v amdgpu.ids > $O/power_probe.txt
{ echo "# tools/power_probe.py --blaslt: the vendor library (torch.matmul -> hipBLASLt, no bias / activation / residual) on the encoder shapes";
  python tools/power_probe.py --blaslt --only fc1,qkv,fc2,sq8k --seconds 3; } 2>&1 | grep -v amdgpu.ids > $O/blaslt_calib.txt
{ echo "# tools/mfma_power.py: register-resident MFMA loops (no LDS, no memory)"; python tools/mfma_power.py --seconds 3; } 2>&1 | grep -v amdgpu.ids > $O/mfma_power.txt
{ echo "# tools/mall_power.py: device copies, working set X (read X/2, write X/2)"; python tools/mall_power.py; } 2>&1 | grep -v amdgpu.ids > $O/mall_power.txt
ls -la $O
