python -m pytest tests -q -m gpu -x -k "from_q or fused or frontend or smoke or structural" 2>&1 | grep -E "passed|failed" | tail -2
python tools/fromq_bench.py --steps 400 2>&1 | tail -3
IRLOSC_WALK=general python tools/fromq_bench.py --steps 400 2>&1 | tail -3 | head -2
python tools/fused_sweep.py --seeds 2 2>&1 | tail -1 | cut -c1-220
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_p -o prof -- python tools/fromq_bench.py --steps 400 --reps 2 > /dev/null 2>&1
db=$(find gpurun_out/prof_p -name "*.db" | head -1); python tools/rocprof_summary.py "$db" 2>&1 | head -6 | cut -c1-150
rm -rf gpurun_out/prof_p
