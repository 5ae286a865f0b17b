set -x
python -m pytest tests/test_gpu_layouts.py tests/test_gpu_parity.py -q -m gpu -x -k "layout or padding or admit or wrench or k12 or full_size" 2>&1 | tail -6 > gpurun_out/r05c_tests.log
bash tools/gpu_profile.sh r05c_k12 "double, 25, false, irlosc::TopoDualUr5" "osc_row16_f64_n25_k12+tree" --layout k12_admit > gpurun_out/profile_r05c_k12.log 2>&1
bash tools/gpu_pmc_sq.sh r05c_k12 "osc_row16_kernel" --layout k12_admit --no-from-q --no-end-to-end > gpurun_out/sq_r05c_k12.log 2>&1
find gpurun_out -maxdepth 1 -type d \( -name "prof_*" -o -name "pmc*" \) | xargs rm -rf
python bench.py > gpurun_out/r05c_bench_default.json 2> gpurun_out/r05c_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05c_bench_driver.json 2> gpurun_out/r05c_bench_driver.err
python bench.py --dtype mixed > gpurun_out/r05c_bench_mixed.json 2> gpurun_out/r05c_bench_mixed.err
python bench.py --layout k12_admit > gpurun_out/r05c_bench_k12_admit.json 2> gpurun_out/r05c_bench_k12_admit.err
cat gpurun_out/r05c_tests.log
for f in default driver mixed k12_admit; do python -c "
import json
d=json.loads(open('gpurun_out/r05c_bench_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_from_kernel_span'), d['config'].get('sustained_value'), d['config'].get('from_q_value'), d['config'].get('parity_max_rel_err'))
"; done
