#!/bin/bash
# A/B of fp32 forward-scan builds on one box: tools/ab_fwd32.sh  (run on the GPU box; builds the variants there)
cd $GRAFT_REPO_ROOT
bash tools/exp_lib.sh w4 "-DDM_FWD_F32_WAVES=4 -DDM_FAST_BUILD" scan_fwd_f32 > /dev/null 2>&1
bash tools/exp_lib.sh w3 "-DDM_FWD_F32_WAVES=3 -DDM_FAST_BUILD" scan_fwd_f32 > /dev/null 2>&1
C=diffma-diffusion-mamba_amd/csrc
for r in 1 2; do
 for L in libdiffma_hip.so lib_w3.so lib_w4.so; do
  echo "== $L"; DIFFMA_HIP_LIB=$PWD/$C/$L python tools/bench_kernels.py --dtype fp32 --batch 768 --only scan_fwd --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-160
 done
done
