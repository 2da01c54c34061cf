#!/bin/bash
# PMC profile of the scan forward kernel at the bench shape (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc/trace -o scan -- python $R/tools/bench_kernels.py --batch 192 --iters 10 > $R/gpurun_out/pmc/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/pmc/pmc1 -o scan -- python $R/tools/bench_kernels.py --batch 192 --iters 3 > $R/gpurun_out/pmc/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc/pmc2 -o scan -- python $R/tools/bench_kernels.py --batch 192 --iters 3 > $R/gpurun_out/pmc/pmc2.log 2>&1
ls -R $R/gpurun_out/pmc | head -40
