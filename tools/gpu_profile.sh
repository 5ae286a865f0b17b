#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_profile.sh <tag>
# kernel trace + stats, then the two PMC passes for HBM traffic (separate runs, see MI355X_MICROARCH.md)
set -x
tag=${1:-rXX}
export TMPDIR=/tmp
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary"
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o prof -- $B > gpurun_out/bench_prof_$tag.log 2>&1
grep -a "^{\"metric\"" gpurun_out/bench_prof_$tag.log | tail -1 > gpurun_out/bench_under_rocprof_$tag.json
python tools/rocprof_summary.py $(find gpurun_out/prof_$tag -name "*.db" | head -1) > gpurun_out/kernel_stats_$tag.txt 2>&1; cat gpurun_out/kernel_stats_$tag.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$tag -o pmc -- $B > gpurun_out/pmc_fetch_$tag.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$tag -o pmc -- $B > gpurun_out/pmc_write_$tag.log 2>&1
python tools/pmc_dump.py $(find gpurun_out/pmc_fetch_$tag -name "*.db" | head -1) osc_ > gpurun_out/pmc_fetch_size_$tag.txt; cat gpurun_out/pmc_fetch_size_$tag.txt
python tools/pmc_dump.py $(find gpurun_out/pmc_write_$tag -name "*.db" | head -1) osc_ > gpurun_out/pmc_write_size_$tag.txt; cat gpurun_out/pmc_write_size_$tag.txt
python tools/pmc_traffic.py $(find gpurun_out/pmc_fetch_$tag -name "*.db" | head -1) $(find gpurun_out/pmc_write_$tag -name "*.db" | head -1) osc_group_kernel_f32 osc_group_f32_n25_k13 gpurun_out/hbm_traffic_$tag.json
