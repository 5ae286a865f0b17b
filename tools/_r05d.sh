set -x
python -m pytest tests -q -m gpu -x -k "from_q or fused or frontend or smoke" 2>&1 | tail -6 > gpurun_out/r05d_tests.log
cat gpurun_out/r05d_tests.log
python tools/fromq_bench.py --steps 400 > gpurun_out/r05d_fromq_s.txt 2>&1
IRLOSC_WALK=general python tools/fromq_bench.py --steps 400 > gpurun_out/r05d_fromq_general.txt 2>&1
tail -5 gpurun_out/r05d_fromq_s.txt gpurun_out/r05d_fromq_general.txt
python tools/fused_sweep.py --seeds 6 > gpurun_out/r05d_fused_sweep.txt 2>&1
tail -12 gpurun_out/r05d_fused_sweep.txt
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05d_fq -o prof -- python tools/fromq_bench.py --steps 400 --reps 2 > gpurun_out/r05d_fromq_prof.log 2>&1
db=$(find gpurun_out/prof_r05d_fq -name "*.db" | head -1); python tools/rocprof_summary.py "$db" > gpurun_out/r05d_fromq_kernel_stats.txt 2>&1; head -12 gpurun_out/r05d_fromq_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/prof_r05d_fq
