#!/bin/bash
# Round 6 final evidence run (GPU box, repo root): the whole GPU suite, then every profile / sweep / bench line that DESIGN.md and
# profiles/ quote, on ONE box with the library as committed.  Summaries under gpurun_out/ (raw rocprofv3 databases are deleted).
T=${1:-r06}
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${T}_gpu_all.log; grep -E "passed|failed|error" gpurun_out/${T}_gpu_all.log
python bench.py --sustained-steps 8000 --no-from-q --no-end-to-end > gpurun_out/${T}_bench_before_the_trace.json 2> /dev/null      # (untraced, minutes before the trace: boxes drift)
bash tools/gpu_profile.sh ${T}_f64 "double, 25, false, irlosc::TopoDualUr5" "osc_row16_f64_n25_k13+tree" > gpurun_out/profile_${T}_f64.log 2>&1
bash tools/gpu_pmc_sq.sh ${T}_f64 "osc_row16_kernel" --no-from-q --no-end-to-end > gpurun_out/sq_${T}_f64.log 2>&1
bash tools/gpu_profile.sh ${T}_mixed "float, 25, false, irlosc::TopoDualUr5" "osc_row16_f32in_f64_n25_k13+tree" --dtype mixed > gpurun_out/profile_${T}_mixed.log 2>&1
bash tools/gpu_pmc_sq.sh ${T}_mixed "osc_row16_kernel" --dtype mixed --no-from-q --no-end-to-end > gpurun_out/sq_${T}_mixed.log 2>&1
bash tools/gpu_profile.sh ${T}_k12 "double, 25, false, irlosc::TopoDualUr5" "osc_row16_f64_n25_k12+tree" --layout k12_admit > gpurun_out/profile_${T}_k12.log 2>&1
bash tools/gpu_pmc_sq.sh ${T}_k12 "osc_row16_kernel" --layout k12_admit --no-from-q --no-end-to-end > gpurun_out/sq_${T}_k12.log 2>&1
find gpurun_out -maxdepth 1 -type d \( -name "prof_*" -o -name "pmc*" \) | xargs rm -rf
bash tools/gpu_profile_fromq.sh ${T} > gpurun_out/profile_${T}_fromq.log 2>&1
bash tools/gpu_pmc_fromq.sh ${T} > /dev/null 2>&1
find gpurun_out -maxdepth 1 -type d \( -name "fq_*" -o -name "pmc*" \) | xargs rm -rf
IRLOSC_FQ_OVERLAP=0 bash tools/gpu_profile_fromq.sh ${T}s > gpurun_out/profile_${T}s_fromq.log 2>&1      # every train on ONE stream: each kernel's own cost
IRLOSC_FQ_OVERLAP=0 bash tools/gpu_pmc_fromq.sh ${T}s > /dev/null 2>&1
(echo "# IRLOSC_FQ_OVERLAP=0 (every train on one stream)"; IRLOSC_FQ_OVERLAP=0 python tools/fromq_bench.py --steps 8000 --reps 2; echo "# default (consecutive trains on two banks / streams)"; python tools/fromq_bench.py --steps 8000 --reps 2) > gpurun_out/${T}_fromq_one_vs_two_streams.txt 2>&1
TAG=${T} bash tools/gpu_r6_eigmin.sh > /dev/null 2>&1
find gpurun_out -maxdepth 1 -type d \( -name "fq_*" -o -name "pmc*" \) | xargs rm -rf
python tools/train_timing.py --from-q --out gpurun_out/${T}_train_timing_fromq.json > gpurun_out/${T}_train_timing_fromq.log 2>&1
python tools/train_timing.py --out gpurun_out/${T}_train_timing.json > gpurun_out/${T}_train_timing.log 2>&1
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> /dev/null
python bench.py --dtype mixed > gpurun_out/${T}_bench_mixed.json 2> /dev/null
python bench.py --layout k12_admit > gpurun_out/${T}_bench_k12_admit.json 2> /dev/null
python bench.py --layout k7 > gpurun_out/${T}_bench_k7.json 2> /dev/null
python bench.py --batch 4096 > gpurun_out/${T}_bench_b4096.json 2> /dev/null
python bench.py --batch 32768 > gpurun_out/${T}_bench_b32768.json 2> /dev/null
python tools/layout_sweep.py --out gpurun_out/layout_sweep_${T}.json > gpurun_out/layout_sweep_${T}.txt 2>&1
python tools/fused_sweep.py --seeds 6 > gpurun_out/${T}_fused_sweep.txt 2>&1
IRLOSC_LANE=0 python tools/fromq_bench.py --steps 128 --reps 3 > gpurun_out/${T}_fromq_ab.txt 2>&1      # the row16 FROMQ kernel behind the walk (round 5's path) ...
python tools/fromq_bench.py --steps 128 --reps 3 >> gpurun_out/${T}_fromq_ab.txt 2>&1                    # ... against the lane-per-robot step, same box
for l in k12_admit k7; do IRLOSC_LANE=0 python tools/fromq_bench.py --layout $l --steps 128 --reps 3 >> gpurun_out/${T}_fromq_ab.txt 2>&1; python tools/fromq_bench.py --layout $l --steps 128 --reps 3 >> gpurun_out/${T}_fromq_ab.txt 2>&1; done
python tools/parity_sweep.py --seeds 8 > gpurun_out/${T}_parity_sweep.txt 2>&1
python tools/parity_sweep.py --seeds 8 --stress >> gpurun_out/${T}_parity_sweep.txt 2>&1
python tools/parity_sweep.py --seeds 8 --physical >> gpurun_out/${T}_parity_sweep.txt 2>&1
python tools/parity_sweep.py --seeds 4 --stress --layout k12_admit >> gpurun_out/${T}_parity_sweep.txt 2>&1
for f in before_the_trace default driver mixed k12_admit k7 b4096 b32768; do python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$f', '%.4g' % d['value'], '%.5f' % d['ms_per_step'], 'frac %.3f' % r['frac'], 'span', r.get('untraced_kernel_span_us'), 'sclk', r.get('sclk_mhz'), 'sustained', c.get('sustained_value'), 'from_q', c.get('from_q_value'), 'parity', c.get('parity_max_rel_err'), c.get('parity_n_over_tol'))
"; done
grep TOTAL gpurun_out/${T}_parity_sweep.txt gpurun_out/${T}_fused_sweep.txt | cut -c1-200
