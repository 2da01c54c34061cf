#!/bin/bash
# developer aid: find which earlier test makes test_diffma_mamba2_forward_matches_reference produce NaN in bf16
T=tests/test_model_gpu.py::test_diffma_mamba2_forward_matches_reference
run() { echo "== $*"; timeout 600 python -m pytest -q -p no:cacheprovider "$@" 2>&1 | tail -3; }
run $T
run tests/test_drivers_gpu.py $T -k "sample_main or mamba2_forward"
run tests/test_drivers_gpu.py $T -k "train_main or mamba2_forward"
run tests/test_drivers_gpu.py $T -k "sampling_loops or ct_encoder or adamw or mamba2_forward"
run tests/test_kernels_gpu.py $T
run tests/test_model_gpu.py
