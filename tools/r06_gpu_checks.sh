#!/bin/bash
# round-6 GPU pass 1: K2 halves-in-sequence against the whole-channel build (same box), its tests, the suite's long poles
mkdir -p gpurun_out/r06
export DIFFMA_TEST_REPORT_DIR=$PWD/gpurun_out/r06
L=diffma-diffusion-mamba_amd/csrc
( KB_BATCH=1536 tools/ab.sh scan_hoist,scan_idx $L/libdiffma_hip.so $L/lib_k2old.so ) > gpurun_out/r06/k2_ab.txt 2>&1
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "scan" --durations=15 -p no:cacheprovider > gpurun_out/r06/scan_tests.txt 2>&1
tail -25 gpurun_out/r06/scan_tests.txt
python -m pytest tests/test_model_gpu.py tests/test_drivers_gpu.py -m gpu -q -x --durations=15 -p no:cacheprovider -k "full_size or bench_dispatch or sample_main or training_step_gradients" > gpurun_out/r06/model_tests.txt 2>&1
tail -25 gpurun_out/r06/model_tests.txt
cat gpurun_out/r06/k2_ab.txt
