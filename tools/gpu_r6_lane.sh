#!/bin/bash
# round 6: first contact of the lane-per-robot OSC step with the GPU -- fused tests, then A/B timing against the row16 FROMQ kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 __graft_entry__.py > gpurun_out/r6_build.log 2>&1 || { tail -20 gpurun_out/r6_build.log; exit 1; }
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -q -m gpu -k "fused or from_q or reachable or walk" -x 2>&1 | tail -25 | tee gpurun_out/r6_lane_tests.log
for lay in k13 k12_admit k7; do
  for lane in 0 1; do
    echo "== layout $lay IRLOSC_LANE=$lane" | tee -a gpurun_out/r6_lane_ab.log
    IRLOSC_LANE=$lane timeout 300 python3 tools/fromq_bench.py --layout $lay --steps 64 --reps 3 2>&1 | tail -5 | tee -a gpurun_out/r6_lane_ab.log
  done
done
