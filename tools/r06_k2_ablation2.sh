L=diffma-diffusion-mamba_amd/csrc
for r in 1 2; do
for lib in libdiffma_hip.so lib_k2e16.so lib_k2e4.so lib_k2e64.so lib_k2e8.so lib_k2e2.so lib_k2e76.so; do
  echo "== $lib"; DIFFMA_HIP_LIB=$PWD/$L/$lib KB_BATCH=1536 tools/kb.sh scan_hoist 2>&1 | grep kernel_only
done; done
