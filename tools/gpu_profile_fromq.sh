#!/bin/bash
# usage (GPU box, repo root): bash tools/gpu_profile_fromq.sh <tag> [fromq_bench flags]
# The path from joint coordinates (tools/fromq_bench.py; IRLOSC_FUSED=0 for the one through dense records): kernel trace +
# stats, then FETCH_SIZE / WRITE_SIZE in their own passes (counters only), per kernel.  Everything under gpurun_out/.
tag=${1:-rXX}
shift
export TMPDIR=/tmp
B="python tools/fromq_bench.py --steps 64 --reps 2 $*"
rm -rf gpurun_out/fq_prof_$tag gpurun_out/fq_fetch_$tag gpurun_out/fq_write_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/fq_prof_$tag -o prof -- $B > gpurun_out/fq_prof_$tag.log 2>&1
db=$(find gpurun_out/fq_prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > gpurun_out/fromq_kernel_stats_$tag.txt 2>&1
cut -c1-200 gpurun_out/fromq_kernel_stats_$tag.txt; tail -4 gpurun_out/fq_prof_$tag.log
timeout 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/fq_fetch_$tag -o pmc -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/fq_write_$tag -o pmc -- $B > /dev/null 2>&1
fdb=$(find gpurun_out/fq_fetch_$tag -name "*.db" | head -1); wdb=$(find gpurun_out/fq_write_$tag -name "*.db" | head -1)
if [ -n "$fdb" ] && [ -n "$wdb" ]; then
  (python tools/pmc_dump.py "$fdb" osc_; python tools/pmc_dump.py "$wdb" osc_) > gpurun_out/fromq_pmc_fetch_write_$tag.txt
  cat gpurun_out/fromq_pmc_fetch_write_$tag.txt
  if grep -q osc_lane_kernel gpurun_out/fromq_kernel_stats_$tag.txt; then      # round 6: the OSC step one lane per robot + its eigen pass
    python tools/pmc_traffic.py "$fdb" "$wdb" osc_frontend_lane_compact "fromq_lane:walk" gpurun_out/hbm_traffic_fq_$tag.json "$db" 65536 8
    python tools/pmc_traffic.py "$fdb" "$wdb" osc_lane_kernel "fromq_lane:osc_lane" gpurun_out/hbm_traffic_fq_$tag.json "$db" 65536 8
    python tools/pmc_traffic.py "$fdb" "$wdb" osc_lane_eigen_kernel "fromq_lane:eigen_pass" gpurun_out/hbm_traffic_fq_$tag.json "$db" 65536 8
    # (osc_lane_eigen16_kernel, the form for thin lists, is launched too and returns at once on this workload: see the kernel stats)
  else
    python tools/pmc_traffic.py "$fdb" "$wdb" osc_frontend_lane_compact "osc_frontend_lane_compact_dual_ur5" gpurun_out/hbm_traffic_fq_$tag.json "$db" 65536 8
    python tools/pmc_traffic.py "$fdb" "$wdb" "25, true" "osc_row16_f64_n25_k13_fromq" gpurun_out/hbm_traffic_fq_$tag.json "$db" 65536 8
  fi
fi
