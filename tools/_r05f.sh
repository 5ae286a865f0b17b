export TMPDIR=/tmp
for v in smallside nostore; do
  export IRLOSC_LIB=tools/_exp/libirlosc_$v.so
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v_$v -o prof -- python tools/fromq_bench.py --steps 32 --reps 1 > /dev/null 2>&1
  db=$(find gpurun_out/prof_v_$v -name "*.db" | head -1); echo $v; python tools/rocprof_summary.py "$db" 2>&1 | grep -i "compact" | cut -c1-140
  rm -rf gpurun_out/prof_v_$v
done
