#!/bin/bash
# round 6: the device-side choice between the two forms of the eigen pass (IRLOSC_LANE_EIG_MIN = flagged robots per step from which the
# lane form takes a step): 0 = always the lane form, 1000000 = always the row16 form, default 3000; fused path, several batch sizes
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r6_eigmin_${TAG:-x}.log; : > $out
for rep in 1 2; do
  for B in ${SIZES:-1024 4096 16384 32768 65536}; do
    for lay in ${LAYOUTS:-k13 k12_admit}; do
      for m in ${MINS:-default 0 1000000}; do
        if [ "$m" = default ]; then unset IRLOSC_LANE_EIG_MIN; else export IRLOSC_LANE_EIG_MIN=$m; fi
        r=$(timeout 300 python3 tools/fromq_bench.py --layout $lay --batch $B --steps ${STEPS:-4000} --reps 2 2>&1 | grep -a "M steps" | tail -n 1 | awk '{print $(NF-2)}')
        echo "$rep B=$B $lay min=$m $r" | tee -a $out
      done
    done
  done
done
