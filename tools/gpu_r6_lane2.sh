#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python3 __graft_entry__.py > gpurun_out/r6_build.log 2>&1 || { tail -20 gpurun_out/r6_build.log; exit 1; }
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -q -m gpu -k "fused or from_q or reachable or walk" -x 2>&1 | tail -5
echo "== phase stamps (lane kernel)"; IRLOSC_PHASE_LANE=1 IRLOSC_LIB=tools/_exp/libirlosc_stamps.so timeout 300 python3 tools/phase_timing.py f64 k13 65536 fromq 2>&1 | grep -a "phase timing" | tail -2
for v in "" jt2; do
  echo "== k13 variant '$v'"
  lib=irl_control_amd/libirlosc.so; [ -n "$v" ] && lib=tools/_exp/libirlosc_$v.so
  IRLOSC_LIB=$lib timeout 300 python3 tools/fromq_bench.py --layout k13 --steps 64 --reps 3 2>&1 | tail -3 | cut -c1-150
done
bash tools/gpu_profile_fromq.sh r06d_lane --layout k13 2>&1 | grep "osc_lane\|compact_kernel\|task_rows" | cut -c1-130
