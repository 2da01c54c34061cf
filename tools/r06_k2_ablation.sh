#!/bin/bash
# K2 ablations on the round-6 kernel (results WRONG by design, timing only): which part of the dB / dC path costs what
L=diffma-diffusion-mamba_amd/csrc
for r in 1 2; do
for lib in libdiffma_hip.so lib_k2e128.so lib_k2e32.so lib_k2e34.so lib_k2e1.so; do
  echo "== $lib"; DIFFMA_HIP_LIB=$PWD/$L/$lib KB_BATCH=1536 tools/kb.sh scan_hoist 2>&1 | grep kernel_only
done; done
