#!/bin/bash
# Round-2 profile set (run on the GPU box; summaries are copied into profiles/ by hand afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (top kernels)
#   2. PMC passes of the two scan kernels at the bench shape (tools/prof_kernel.sh)
#   3. HBM traffic of the scan kernels (separate FETCH_SIZE / WRITE_SIZE passes, tools/prof_traffic.sh)
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh --no-extras > gpurun_out/prof_bench_r02.log 2>&1
bash tools/prof_kernel.sh bwd_r02 scan_bwd_kernel --dtype bf16 --batch 1536 --only scan_idx > gpurun_out/pmc_bwd_r02.txt 2>&1
bash tools/prof_kernel.sh fwd_r02 scan_fwd_kernel --dtype bf16 --batch 1536 --only scan_idx > gpurun_out/pmc_fwd_r02.txt 2>&1
bash tools/prof_kernel.sh fwd32_r02 scan_fwd_kernel --dtype fp32 --batch 768 --only scan_fwd > gpurun_out/pmc_fwd32_r02.txt 2>&1
bash tools/prof_traffic.sh > gpurun_out/traffic_r02.log 2>&1
tail -3 gpurun_out/prof_bench_r02.log
