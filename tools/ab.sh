#!/bin/bash
# A/B two kernel builds on the same GPU box: tools/ab.sh <kernels> libA.so libB.so   (alternates A B A B)
K=${1:-scan_bwd,scan_idx}; A=$2; B=$3
for r in 1 2; do
  for L in "$A" "$B"; do echo "== $L"; DIFFMA_HIP_LIB=$PWD/$L tools/kb.sh $K 2>&1 | grep -v amdgpu.ids; done
done
