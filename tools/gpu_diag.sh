set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/torchrun1.log 2>&1; tail -2 gpurun_out/torchrun1.log
IRLOSC_PHASE_TIMING=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/phase.log 2>&1; grep -a "phase timing" gpurun_out/phase.log | tail -3
rm -rf gpurun_out/pmcA gpurun_out/pmcB
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmcA -o pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmcA.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmcB -o pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmcB.log 2>&1
find gpurun_out/pmcA gpurun_out/pmcB -name "*.db" | while read f; do python tools/pmc_dump.py $f osc_group; done
