#!/bin/bash
# Round 5 wide parity sweeps on the library as committed (GPU box, repo root): the row16 kernels -- exact and KMAX-padded, dense and tree
# form, float64 and float32 records -- against the generic kernel's Jacobi solution (tools/parity_sweep.py), and the fused path against
# the path through dense records (tools/fused_sweep.py).  Totals only are kept.
out=gpurun_out/r05_parity_sweep_wide.txt; : > $out
run() { echo "## parity_sweep.py $*" >> $out; python tools/parity_sweep.py "$@" 2>&1 | grep -E "TOTAL|over 1e-05, [1-9]" | tail -3 >> $out; }
run --seeds 96 --physical
run --seeds 48 --physical --stress
run --seeds 48 --stress
run --seeds 32 --mode mixed --stress
run --seeds 32 --mode mixed --physical
run --seeds 24 --layout k12_admit --stress
run --seeds 24 --layout k12_admit --physical
run --seeds 16 --layout k7 --stress
run --seeds 16 --layout k6 --physical --stress
for lay in r6 r3 br4 br7 rl8 rl9_admit rlb10 rlbr10 rlb11_branch_b brl14 rlb16; do
  run --seeds 8 --layout $lay --stress
  run --seeds 8 --layout $lay --physical
done
echo "## fused_sweep.py --seeds 40" >> $out
python tools/fused_sweep.py --seeds 40 2>&1 | tail -1 >> $out
cat $out
