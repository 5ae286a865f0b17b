#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_profile_frontend.sh <tag>
# Kernel trace of the rigid-body front end alone (tools/fe_bench.py), then FETCH_SIZE / WRITE_SIZE and SQ counters in
# separate counter-only passes.  Output under gpurun_out/ ; copy what is to be kept into profiles/.
tag=${1:-rXX}
export TMPDIR=/tmp
B="python tools/fe_bench.py --batches 65536 --reps 12"
rm -rf gpurun_out/fe_prof_$tag gpurun_out/fe_pmc*_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/fe_prof_$tag -o prof -- $B > gpurun_out/fe_prof_$tag.log 2>&1
db=$(find gpurun_out/fe_prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && timeout 60 python tools/rocprof_summary.py "$db" > gpurun_out/fe_kernel_stats_$tag.txt 2>&1
cat gpurun_out/fe_kernel_stats_$tag.txt
(for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"; do
   d=gpurun_out/fe_pmc_$(echo $c | cut -d' ' -f1)_$tag
   timeout 300 rocprofv3 --pmc $c -d $d -o pmc -- $B > $d.log 2>&1
   db=$(find $d -name "*.db" | head -1)
   if [ -n "$db" ]; then timeout 60 python tools/pmc_dump.py "$db" osc_frontend; else echo "no db for $c"; fi
 done) > gpurun_out/fe_pmc_$tag.txt
cat gpurun_out/fe_pmc_$tag.txt
