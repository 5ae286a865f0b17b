#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_profile.sh <tag> <kernel-substring> <hbm_traffic.json key> [bench flags]
# kernel trace + stats, then the two PMC passes for HBM traffic (separate runs, counters only: MI355X_MICROARCH.md, HBM).
# Everything lands under gpurun_out/ ; copy the summaries to keep into profiles/.
tag=${1:-rXX}; kern=${2:-25, false, irlosc::TopoDualUr5}; key=${3:-osc_row16_f64_n25_k13+tree}
# (defaults: the tree form of the fp64 row16 kernel, what `python bench.py` runs; for `--workload synthetic` pass "25, false, void" and
#  osc_row16_f64_n25_k13)
shift; shift; shift
export TMPDIR=/tmp
# The bench command itself at its default length (1 000 untimed + 200 warm-up + 2 000 timed steps = 400 trains): a short command
# (round 3: 168 steps) is over before the clocks have settled and its per-dispatch average is not the steady state.
B="python bench.py --sustained-steps 8000 --no-cpu-baseline --no-secondary --no-from-q --no-end-to-end $*"      # (the sustained leg as in rounds 4-5: 100 000 steps are 12 500 traced dispatches)
BP="python bench.py --steps 256 --warmup 64 --preroll 480 --sustained-steps 0 --no-cpu-baseline --no-secondary --no-from-q --no-end-to-end $*"   # PMC passes (slow under counters)
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o prof -- $B > gpurun_out/bench_prof_$tag.log 2>&1
grep -a "^{\"metric\"" gpurun_out/bench_prof_$tag.log | tail -1 > gpurun_out/bench_under_rocprof_$tag.json
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > gpurun_out/kernel_stats_$tag.txt 2>&1
cat gpurun_out/kernel_stats_$tag.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$tag -o pmc -- $BP > gpurun_out/pmc_fetch_$tag.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$tag -o pmc -- $BP > gpurun_out/pmc_write_$tag.log 2>&1
fdb=$(find gpurun_out/pmc_fetch_$tag -name "*.db" | head -1); wdb=$(find gpurun_out/pmc_write_$tag -name "*.db" | head -1)
if [ -n "$fdb" ] && [ -n "$wdb" ]; then
  python tools/pmc_dump.py "$fdb" osc_ > gpurun_out/pmc_fetch_size_$tag.txt; cat gpurun_out/pmc_fetch_size_$tag.txt
  python tools/pmc_dump.py "$wdb" osc_ > gpurun_out/pmc_write_size_$tag.txt; cat gpurun_out/pmc_write_size_$tag.txt
  python tools/pmc_traffic.py "$fdb" "$wdb" "$kern" "$key" gpurun_out/hbm_traffic_$tag.json "$db" ${INSTANCES:-65536} ${SPL:-8}
  python tools/rocprof_gaps.py "$db" 200 > gpurun_out/kernel_gaps_$tag.txt; cat gpurun_out/kernel_gaps_$tag.txt
fi
