#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_pmc_sq.sh <tag> [kernel-substring] [extra bench flags]
# SQ occupancy / issue counters of the dominant kernel, two PMC passes (counters only: no trace domains).
tag=${1:-rXX}
kern=${2:-osc_group}
shift; shift
export TMPDIR=/tmp
B="python bench.py --steps 160 --warmup 16 --preroll 160 --sustained-steps 0 --no-cpu-baseline --no-secondary $*"
rm -rf gpurun_out/pmcA_$tag gpurun_out/pmcB_$tag
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmcA_$tag -o pmc -- $B > gpurun_out/pmcA_$tag.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmcB_$tag -o pmc -- $B > gpurun_out/pmcB_$tag.log 2>&1
(for d in gpurun_out/pmcA_$tag gpurun_out/pmcB_$tag; do
   db=$(find $d -name "*.db" | head -1)
   if [ -n "$db" ]; then python tools/pmc_dump.py "$db" "$kern"; else echo "no db under $d"; fi
 done) > gpurun_out/pmc_sq_$tag.txt
cat gpurun_out/pmc_sq_$tag.txt
