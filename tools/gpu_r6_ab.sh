#!/bin/bash
# round 6: alternating A/B of variant libraries on the fused path (VARIANTS="name …" under tools/_exp/; "" = the product), one box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r6_ab_${TAG:-x}.log; : > $out
for rep in 1 2; do
  for lay in ${LAYOUTS:-k13 k12_admit k7}; do
    for v in product $VARIANTS; do
      lib=irl_control_amd/libirlosc.so; [ "$v" != product ] && lib=tools/_exp/libirlosc_$v.so
      echo "== rep $rep layout $lay variant $v" >> $out
      IRLOSC_LIB=$lib timeout 300 python3 tools/fromq_bench.py --layout $lay --steps ${STEPS:-4000} --reps 2 2>&1 | tail -n 3 | cut -c1-150 >> $out
    done
  done
done
grep -a "==\|M steps" $out | paste - - - | awk '{print $3,$5,$7,$(NF-2)}' | sort | tee gpurun_out/r6_ab_${TAG:-x}_summary.txt
