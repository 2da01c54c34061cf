#!/bin/bash
# K4x slab form, the FAILING row-table layout of round 4 rebuilt (csrc/conv_xproj.hip, -DDM_K4X_REPRO=v) and soaked: which single
# change around the multiply cures it?  Build (in the build container):
#   for v in 1 2 3 4 5 6; do hipcc <Makefile flags> -DDM_K4X_REPRO=$v -c conv_xproj.hip -o cx$v.o; hipcc -shared ... -o tools/ubench/build/libk4x_v$v.so; done
# Run (GPU box): bash tools/ubench/k4x_repro.sh   -> gpurun_out/k4x_repro.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=gpurun_out/k4x_repro.txt; : > $OUT
for v in ${VARIANTS:-0 6 1 2 3 4 5 7 8 9 10}; do
  if [ $v = 0 ]; then LIB=""; else LIB=$GRAFT_REPO_ROOT/tools/ubench/build/libk4x_v$v.so; fi
  echo "== variant $v" >> $OUT
  DIFFMA_HIP_LIB=$LIB SLAB=1 ITERS=${ITERS:-3000} timeout 600 python tools/dbg_k4x2.py 2>&1 | tail -4 >> $OUT
done
cat $OUT
