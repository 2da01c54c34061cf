#!/bin/bash
# round-6 GPU pass 3: dm_ssd_bwd with transposing LDS reads instead of selector MFMAs vs the previous build; its tests; PMC of it
mkdir -p gpurun_out/r06
L=diffma-diffusion-mamba_amd/csrc
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "ssd or mamba2" -p no:cacheprovider > gpurun_out/r06/ssd_tests2.txt 2>&1
tail -5 gpurun_out/r06/ssd_tests2.txt
for r in 1 2; do for lib in libdiffma_hip.so lib_ssdold.so; do echo "== $lib"; DIFFMA_HIP_LIB=$PWD/$L/$lib python tools/bench_ssd.py 256 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06/ssd_ab2.txt 2>&1
cat gpurun_out/r06/ssd_ab2.txt
bash tools/pmc_cmd.sh r06_m2_ssd "%ssd_bwd%|%ssd_fwd_k%" python $PWD/tools/bench_ssd.py 256 > /dev/null 2>&1
cat gpurun_out/r06_m2_ssd_pmc.txt | head -40
