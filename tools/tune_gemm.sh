#!/bin/bash
# Record hipBLASLt/rocBLAS solutions for GEMM shapes the packaged table does not know yet (run on the GPU box, ~1 min)
# and merge them into diffma-diffusion-mamba_amd/tuned/gemm_gfx950.csv:   tools/tune_gemm.sh [bench.py args]
T=diffma-diffusion-mamba_amd/tuned/gemm_gfx950.csv
mkdir -p gpurun_out; rm -f gpurun_out/gemm_tuning_rank0.csv
python bench.py --gemm-tuning tune --cpu-steps 0 --steps 3 --warmup 2 "$@" > gpurun_out/tune_train.log 2>&1
tail -1 gpurun_out/tune_train.log | cut -c1-200
if [ -f gpurun_out/gemm_tuning_rank0.csv ]; then
  grep -v "^Validator" gpurun_out/gemm_tuning_rank0.csv | while read -r line; do
    key=$(echo "$line" | cut -d, -f1,2)
    grep -q "^$key," $T || echo "$line" >> $T
  done
fi
cp $T gpurun_out/gemm_gfx950.merged.csv; wc -l $T
