#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_pmc_fromq.sh <tag> [fromq_bench flags]
# SQ occupancy / issue counters of the kernels of the path from joint coordinates (tools/fromq_bench.py), two PMC passes
# (counters only: no trace domains); run once as is (fused) and once with IRLOSC_FUSED=0 for the path through dense records.
tag=${1:-rXX}
shift
export TMPDIR=/tmp
B="python tools/fromq_bench.py --steps 16 --reps 1 $*"
rm -rf gpurun_out/pmcA_$tag gpurun_out/pmcB_$tag
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmcA_$tag -o pmc -- $B > gpurun_out/pmcA_$tag.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmcB_$tag -o pmc -- $B > gpurun_out/pmcB_$tag.log 2>&1
(for d in gpurun_out/pmcA_$tag gpurun_out/pmcB_$tag; do
   db=$(find $d -name "*.db" | head -1)
   if [ -n "$db" ]; then python tools/pmc_dump.py "$db" "osc_"; else echo "no db under $d"; fi
 done) > gpurun_out/pmc_fromq_$tag.txt
cat gpurun_out/pmc_fromq_$tag.txt
