#!/bin/bash
# round 6: fused path at several batch sizes, product against variant libraries (alternating), one box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r6_ab_sizes_${TAG:-x}.log; : > $out
for rep in 1 2; do
  for B in ${SIZES:-1024 4096 16384 32768}; do
    for lay in ${LAYOUTS:-k13 k12_admit}; do
      for v in product $VARIANTS; do
        lib=irl_control_amd/libirlosc.so; [ "$v" != product ] && lib=tools/_exp/libirlosc_$v.so
        r=$(IRLOSC_LIB=$lib timeout 300 python3 tools/fromq_bench.py --layout $lay --batch $B --steps ${STEPS:-4000} --reps 2 2>&1 | grep -a "M steps" | tail -n 1 | awk '{print $(NF-2)}')
        echo "$rep B=$B $lay $v $r" | tee -a $out
      done
    done
  done
done
