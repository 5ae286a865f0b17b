#!/bin/bash
# Round 5 evidence run (GPU box, repo root): kernel trace + FETCH / WRITE + SQ passes for the three bench configurations that
# BASELINE.json names on one GPU -- float64 records (headline), float32 records with fp64 arithmetic (configs[2]), k12 + admittance
# (configs[4]) -- then the throughput sweep over every reachable layout.  Summaries land under gpurun_out/ ; the ones to keep are
# copied into profiles/ by hand.
set -x
T=${1:-r05a}
bash tools/gpu_profile.sh ${T}_f64 "double, 25, false, irlosc::TopoDualUr5" "osc_row16_f64_n25_k13+tree" > gpurun_out/profile_${T}_f64.log 2>&1
bash tools/gpu_pmc_sq.sh ${T}_f64 "osc_row16_kernel" --no-from-q --no-end-to-end > gpurun_out/sq_${T}_f64.log 2>&1
bash tools/gpu_profile.sh ${T}_mixed "float, 25, false, irlosc::TopoDualUr5" "osc_row16_f32in_f64_n25_k13+tree" --dtype mixed > gpurun_out/profile_${T}_mixed.log 2>&1
bash tools/gpu_pmc_sq.sh ${T}_mixed "osc_row16_kernel" --dtype mixed --no-from-q --no-end-to-end > gpurun_out/sq_${T}_mixed.log 2>&1
bash tools/gpu_profile.sh ${T}_k12 "double, 25, false, irlosc::TopoDualUr5" "osc_row16_f64_n25_k12+tree" --layout k12_admit > gpurun_out/profile_${T}_k12.log 2>&1
bash tools/gpu_pmc_sq.sh ${T}_k12 "osc_row16_kernel" --layout k12_admit --no-from-q --no-end-to-end > gpurun_out/sq_${T}_k12.log 2>&1
# the raw rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back): the summaries above are what is kept
find gpurun_out -maxdepth 1 -type d \( -name "prof_*" -o -name "pmc*" \) | xargs rm -rf
python tools/layout_sweep.py --out gpurun_out/layout_sweep_${T}.json > gpurun_out/layout_sweep_${T}.txt 2>&1
tail -40 gpurun_out/layout_sweep_${T}.txt
