#!/bin/bash
# round 6: the lane-per-robot OSC step on the GPU -- fused tests, then A/B timing against the row16 FROMQ kernel and between prefetch depths
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 __graft_entry__.py > gpurun_out/r6_build.log 2>&1 || { tail -20 gpurun_out/r6_build.log; exit 1; }
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -q -m gpu -k "fused or from_q or reachable or walk" -x 2>&1 | tail -8 | tee gpurun_out/r6_lane_tests.log
: > gpurun_out/r6_lane_ab.log
for v in $VARIANTS; do
  echo "== k13 variant $v" | tee -a gpurun_out/r6_lane_ab.log
  IRLOSC_LIB=tools/_exp/libirlosc_$v.so timeout 300 python3 tools/fromq_bench.py --layout k13 --steps 64 --reps 3 2>&1 | tail -4 | cut -c1-150 | tee -a gpurun_out/r6_lane_ab.log
done
for lay in k13 k12_admit k7; do
  for lane in 0 1; do
    echo "== layout $lay IRLOSC_LANE=$lane" | tee -a gpurun_out/r6_lane_ab.log
    IRLOSC_LANE=$lane timeout 300 python3 tools/fromq_bench.py --layout $lay --steps 64 --reps 3 2>&1 | tail -4 | cut -c1-150 | tee -a gpurun_out/r6_lane_ab.log
  done
done
